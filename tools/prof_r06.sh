#!/bin/bash
# usage: tools/prof_r06.sh [parts = "cfg4 b1 mid enc"]   (GPU box, from the repo root) -- round-6 rocprofv3 evidence -> gpurun_out/r06_prof/
#   cfg4  the default line's search half (207 M, incl. the recall-0.95 point): --kernel-trace --stats; FETCH_SIZE / WRITE_SIZE passes of
#         scan / re-rank / selection with the refine point's (nprobe, k_factor) sweep pinned to the timed shape (BENCH_REFINE_NPROBES /
#         BENCH_REFINE_KFS), so that a kernel's mean is over launches of ONE shape
#   b1    the query-time encoder (one 31-token query): stats + FETCH_SIZE of its kernels
#   mid   the few-hundred-token encoder (bench.py's 16 queries, 570 tokens): stats + FETCH_SIZE + WRITE_SIZE of every kernel of the pass
#   mid256  the same for bench.py's 256 queries (8 097 tokens)
#   enc   encode (cfg3): stats, a plain run, FETCH_SIZE / WRITE_SIZE of its GEMM kernels
# Counters always in their own passes with --kernel-trace only.  tools/pmc_json_r06.py turns the summaries into the stamped
# profiles/r06_*_pmc.json files bench.py reads for `roofline.traffic`.
parts=${1:-"cfg4 b1 mid enc"}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() { # tag, command...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o r -- "$@" > $out/${tag}_under_stats.out 2> $out/${tag}_stats.err
  cp $(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
}
pmc() { # tag, counter, regex, command...
  local tag=$1 c=$2 re=$3; shift 3
  rm -rf /tmp/p_${tag}_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$re" --output-format csv -d /tmp/p_${tag}_$c -o r -- "$@" > $out/${tag}_under_$c.out 2> $out/${tag}_$c.err
  python $R/tools/pmc_summarize.py $(find /tmp/p_${tag}_$c -name "*counter_collection.csv" | head -1) "" > $out/${tag}_$c.txt
}
for part in $parts; do case $part in
cfg4)
  B="python $R/bench.py --no-encode --no-cpu-baseline --streams 1"
  # the step as it runs by default (exact list pruning on; bench.py times the exhaustive step beside it): its kernels
  export BENCH_NO_EXHAUSTIVE=1    # (only the pruned step's launches of the scan kernel in these three passes)
  stats cfg4_pruned $B --no-refine-point --no-recall
  for c in FETCH_SIZE WRITE_SIZE; do pmc cfg4_pruned $c "scan_kernel" $B --no-refine-point --no-recall --steps 10; done
  unset BENCH_NO_EXHAUSTIVE
  # everything below on the EXHAUSTIVE scan (MI_SCAN_PRUNE=0) -- the launch the scan kernel's roofline is quoted on: with the
  # pruning on, the timed step's launches of the same kernel read a sixteenth of the lists and would share every mean
  export MI_SCAN_PRUNE=0
  stats cfg4 $B
  # the refine point the run above chose (held-out batch): the counter passes pin the sweep to it
  KF=$(python -c "import json,sys; d=json.loads([l for l in open('$out/cfg4_under_stats.out') if l.startswith('{')][-1]); a=d['at_recall_095']; print(a['nprobe'], a['k_factor_rf'])")
  set -- $KF
  export BENCH_REFINE_NPROBES=$1 BENCH_REFINE_KFS=$2
  echo "refine point: nprobe $1 k_factor $2"
  for c in FETCH_SIZE WRITE_SIZE; do pmc cfg4 $c "scan_kernel|rerank_sq8|select_pairs" $B --no-recall --steps 10; done
  unset BENCH_REFINE_NPROBES BENCH_REFINE_KFS MI_SCAN_PRUNE ;;
b1)
  Q="python $R/tools/encode_b1.py 31 40 1"
  stats b1 $Q; timeout 300 $Q > $out/b1_plain.out 2>/dev/null
  pmc b1 FETCH_SIZE "few_" $Q ;;
mid)
  export ENC_NQ=16 ENC_REPS=30
  Q="python $R/tools/encode_mid_prof.py"
  stats mid $Q; timeout 300 $Q > $out/mid_plain.out 2>/dev/null
  export ENC_REPS=5
  for c in FETCH_SIZE WRITE_SIZE; do pmc mid $c "mienc" $Q; done
  unset ENC_NQ ENC_REPS ;;
mid256)
  export ENC_NQ=256 ENC_REPS=10
  Q="python $R/tools/encode_mid_prof.py"
  stats mid256 $Q; timeout 300 $Q > $out/mid256_plain.out 2>/dev/null
  export ENC_REPS=3
  for c in FETCH_SIZE WRITE_SIZE; do pmc mid256 $c "mienc" $Q; done
  unset ENC_NQ ENC_REPS ;;
enc)
  E="python $R/bench.py --workload encode --steps 4 --warmup 1 --no-cpu-baseline --no-full"
  stats encode $E; cp $out/encode_under_stats.out $out/encode_under_stats.json
  timeout 600 $E > $out/encode_plain.json 2> $out/encode_plain.err
  for c in FETCH_SIZE WRITE_SIZE; do pmc encode_gemm $c "gemm_bf16_(ring|slab)" $E; done ;;
esac; done
cd $R
for f in cfg4 b1 mid mid256 encode; do [ -f $out/${f}_kernel_stats.csv ] && { echo "== $f"; head -9 $out/${f}_kernel_stats.csv | cut -c1-160; }; done
grep -h "ms per\|per encode" $out/*_plain.out $out/*_under_stats.out 2>/dev/null
