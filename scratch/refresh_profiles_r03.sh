#!/bin/bash
# usage: scratch/refresh_profiles_r03.sh <tag>   (on the GPU box; writes gpurun_out/<tag>/*, to be copied into profiles/)
# default bench + rocprof kernel stats / step timeline + the two --pmc passes (FETCH_SIZE, WRITE_SIZE) + inference bench
set -o pipefail
T=$1; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[1] bench default"; timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "[2] kernel stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 20 --cpu-baseline-steps 0 --no-roofline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
python scratch/timeline3.py $O/prof > $O/step_timeline.txt 2>&1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats_pipelined.csv
echo "[3] pmc fetch"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 10 --warmup 4 --no-roofline --cpu-baseline-steps 0 > $O/pmc_f.log 2>&1 || exit 1
echo "[4] pmc write"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 10 --warmup 4 --no-roofline --cpu-baseline-steps 0 > $O/pmc_w.log 2>&1 || exit 1
python scratch/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json > $O/pmc_summary.txt
echo "[5] inference"; timeout -k 10 300 python bench.py --mode inference --steps 5 --warmup 1 > $O/bench_inference.json 2> $O/bench_inference.err || exit 1
rm -rf $O/prof $O/pmc_f $O/pmc_w
ls $O; cat $O/pmc_summary.txt | head -40
