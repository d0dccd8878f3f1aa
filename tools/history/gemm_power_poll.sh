#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# clocks / power while the slab GEMM and hipBLASLt alternate in 2-second blocks on one shape (GPU box)
export M=${M:-32768} SECS=2 SHAPES=${SHAPES:-down}
( for i in $(seq 1 70); do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.15; done ) > gpurun_out/r03_power_poll_$SHAPES.txt &
P=$!
python tools/gemm_sustained.py 2>&1 | grep -v amdgpu.ids
wait $P
head -3 gpurun_out/r03_power_poll_$SHAPES.txt | cut -c1-400
