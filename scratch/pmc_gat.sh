#!/bin/bash
# usage: scratch/pmc_gat.sh <tag>   (on the GPU box): HBM bytes per launch of the GATv2 step's kernels -- two rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, the program directly after --) of bench.py --model gat, reduced by scratch/pmc_traffic.py
set -o pipefail
T=$1; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --model gat --steps 10 --warmup 4 --no-roofline --cpu-baseline-steps 0 > $O/pmc_f.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --model gat --steps 10 --warmup 4 --no-roofline --cpu-baseline-steps 0 > $O/pmc_w.log 2>&1 || exit 1
python scratch/pmc_traffic.py $O/pmc_f $O/pmc_w $O/gat_pmc_traffic.json > $O/gat_pmc_summary.txt
rm -rf $O/pmc_f $O/pmc_w
grep -i "gat" $O/gat_pmc_summary.txt | head -20
