#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a one-GPU box: N ranks share GPU 0 over gloo (BENCH_REHEARSAL=1).
# Control flow only -- never a measurement.  usage: tools/rehearse_multirank.sh N [bench.py args...]
N=$1; shift
BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29600 + N)) bench.py --gpus $N "$@"
