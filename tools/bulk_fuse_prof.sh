#!/bin/bash
# per-kernel times of the many-token encode with / without the fused epilogues (GPU box): summaries -> gpurun_out/r04_fuse/
out=$GRAFT_REPO_ROOT/gpurun_out/r04_fuse
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
E="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 4 --warmup 1 --no-cpu-baseline"
for v in fused plain "$@"; do
  case $v in fused) e="X=1";; plain) e="MI_NO_BULK_FUSE=1";; norm_only) e="MI_NO_ROPE_FUSE=1";; rope_only) e="MI_NO_NORM_FUSE=1";; esac
  env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_enc_$v -o r -- $E > $out/enc_$v.json 2> $out/enc_$v.err
  cp $(find /tmp/p_enc_$v -name "*kernel_stats.csv" | head -1) $out/enc_${v}_kernel_stats.csv
  echo "== $v"; head -9 $out/enc_${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
