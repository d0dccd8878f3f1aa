#!/bin/bash
# the cfg4 headline step with the exact list pruning in its two forms (MI_SCAN_PRUNE_MODE 1 = two phases, by P1; 2 = the early stop
# inside the scan kernel); each bench line also times the exhaustive step (MI_SCAN_PRUNE=0) in the same process: GPU box
for cfg in ${PRUNE_CFGS:-0:0 1:1 1:2}; do set -- ${cfg/:/ }
MI_SCAN_PRUNE_MODE=$1 MI_SCAN_PRUNE_P1=$2 timeout 600 python bench.py --no-encode --no-cpu-baseline --no-refine-point 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']; r=f['exhaustive_launch']; p=d['pruning']; x=d['exhaustive_scan']
print('mode $1 P1 $2: %.1f queries/s  %.4f ms/step (exhaustive %.1f, %.4f ms, its scan launch %.4f ms)  recall %.4f  pruned scan launch %.4f ms  scanned beyond it %.4f of the groups  host_io %.1f' % (
  d['value'], d['ms_per_step'], x['queries_per_s'], x['ms_per_step'], r['avg_launch_ms'], d['recall_at_10'], f['avg_launch_ms'],
  p.get('scanned_fraction', p.get('second_launch_fraction')), d['host_io']['queries_per_s']))"
done
