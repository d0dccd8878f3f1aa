#!/bin/bash
# usage (GPU box): tools/prof_r05_mid.sh <tag>  -- the few-hundred-token encoder regime (cfg5 batch 16 / 64):
# rocprofv3 kernel stats, then FETCH_SIZE and WRITE_SIZE in their own passes (kernel-trace only beside --pmc)
tag=${1:-r05_mid}
R=$GRAFT_REPO_ROOT
for nq in ${NQS:-16 64}; do
  ENC_NQ=$nq ENC_REPS=30 bash $R/tools/prof_cmd.sh ${tag}_nq${nq}_stats python $R/tools/encode_mid_prof.py
  grep "ms per forward" $R/gpurun_out/${tag}_nq${nq}_stats/cmd.log
  for c in FETCH_SIZE WRITE_SIZE; do
    ENC_NQ=$nq ENC_REPS=5 bash $R/tools/pmc_cmd.sh ${tag}_nq${nq}_$c $c python $R/tools/encode_mid_prof.py > /dev/null
    f=$(find $R/gpurun_out/${tag}_nq${nq}_$c -name "*counter_collection.csv" | head -1)
    python $R/tools/pmc_summarize.py $f mienc > $R/gpurun_out/${tag}_nq${nq}_$c.txt
    find $R/gpurun_out/${tag}_nq${nq}_$c -name "*.csv" -size +4M -delete
  done
done
