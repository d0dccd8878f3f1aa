for cfg in "4 4" "6 8" "8 8" "3 4"; do set -- $cfg
GPU_MAX_HW_QUEUES=$2 BENCH_REFINE_STREAMS=$1 BENCH_REFINE_NPROBES=8 BENCH_REFINE_KFS=464 timeout 600 python bench.py --no-encode --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['at_recall_095']
print('streams $1 queues $2: headline', d['ms_per_step'], 'refine', a['ms_per_step'], a['recall_at_10'])"
done
