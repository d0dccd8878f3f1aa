timeout 300 python -m pytest tests/test_ivfpq_gpu.py -x -q -k "exact_list_pruning or search_matches_oracle" 2>&1 | tail -1
for st in ${PRUNE_STREAMS:-2 3 4}; do
timeout 600 python bench.py --no-encode --no-cpu-baseline --no-refine-point --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']; r=f['exhaustive_launch']; p=d['pruning']; x=d['exhaustive_scan']
print('streams $st: %.1f queries/s  %.4f ms/step (exhaustive %.1f, %.4f ms)  pruned scan launch %.4f ms %.2f GB frac %.3f  scanned %.4f  host_io %.1f  traffic %s' % (
  d['value'], d['ms_per_step'], x['queries_per_s'], x['ms_per_step'], f['avg_launch_ms'], f['bytes_per_launch']/1e9, f['frac'], p['scanned_fraction'], d['host_io']['queries_per_s'], r['traffic']))"
done
