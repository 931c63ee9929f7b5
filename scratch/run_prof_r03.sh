#!/bin/bash
# usage: scratch/run_prof_r03.sh <tag> [bench.py args...]   (on the GPU box): rocprofv3 kernel stats + one-step timeline of a bench run
set -o pipefail
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 20 --cpu-baseline-steps 0 --no-roofline "$@" > $O/bench_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
python scratch/timeline3.py $O/prof > $O/step_timeline.txt 2>&1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/prof
tail -2 $O/step_timeline.txt
