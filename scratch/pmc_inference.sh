#!/bin/bash
# usage: scratch/pmc_inference.sh <tag>   (on the GPU box): where do k_spmm's row gathers come from in SAGE.inference?
# Three separate rocprofv3 --pmc passes (the guide's rule: counters in their own runs, --kernel-trace only; the program directly
# after --): FETCH_SIZE (fabric / HBM side reads; x2 on gfx950), WRITE_SIZE, and the L2's hit / miss / request counts.
set -o pipefail
T=$1; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 bench.py --mode inference --steps 1 --warmup 1 > $O/pmc_$tag.log 2>&1 || { tail -5 $O/pmc_$tag.log; exit 1; }
done
python scratch/pmc_inference.py $O > $O/inference_pmc.json && cat $O/inference_pmc.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum
