timeout 900 python -m pytest tests/test_ivfpq_gpu.py -m gpu -q -k "large_k or candidates or sq8 or merge or refine" 2>&1 | tail -2
python tools/candidates_bench.py 2>/dev/null | tail -1
CAND_KC=2000 python tools/candidates_bench.py 2>/dev/null | tail -1
python bench.py --no-encode --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step']); a=d['at_recall_095']; print(a['qps'], a['ms_per_step'], a['recall_at_10'], a['k_factor_rf'])"
