#!/bin/bash
# the many-token encode with and without the fused RMSNorm / rotary epilogues (csrc/encoder_kernels.h GemmArgs::ssq_out / rope_cs),
# alternating so that the box's thermal drift does not pick the winner: GPU box
mkdir -p gpurun_out/r04
for v in fused plain fused plain norm_only rope_only fused plain; do
  case $v in fused) e="X=1";; plain) e="MI_NO_BULK_FUSE=1";; norm_only) e="MI_NO_ROPE_FUSE=1";; rope_only) e="MI_NO_NORM_FUSE=1";; esac
  env $e timeout 600 python bench.py --workload encode --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r04/enc_$v.json 2> gpurun_out/r04/enc_$v.log
  python - <<PY
import json
o=json.load(open("gpurun_out/r04/enc_$v.json"))
print("$v", o["value"], o["ms_per_step"], o["roofline"]["frac"])
PY
done
