#!/bin/bash
# usage: scratch/pmc_shards.sh <tag>   (on the GPU box): HBM bytes per launch of the static sharded step's kernels -- two rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, the program directly after --) of bench.py --dist shards launched kernel by
# kernel on one stream (counter collection serialises kernels: the flag-ordered two-stream graphs cannot run under it), reduced by
# scratch/pmc_traffic.py
set -o pipefail
T=$1; O=gpurun_out/$T; mkdir -p $O
export BLISS_SHARD_GRAPH=0 BLISS_SHARD_PIPELINE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --dist shards --steps 10 --warmup 4 --no-roofline > $O/pmc_f.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --dist shards --steps 10 --warmup 4 --no-roofline > $O/pmc_w.log 2>&1 || exit 1
python scratch/pmc_traffic.py $O/pmc_f $O/pmc_w $O/shards_pmc.json > $O/shards_pmc_summary.txt
rm -rf $O/pmc_f $O/pmc_w
grep -i "k_sd_\|k_bin_scatter\|k_tile\|k_cross" $O/shards_pmc_summary.txt | head -20
