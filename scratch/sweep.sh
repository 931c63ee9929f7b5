#!/bin/bash
# the other bench configurations through the same loop (one line each)
run() { timeout -k 10 300 python bench.py --cpu-baseline-steps 0 --no-roofline "$@" 2>gpurun_out/sweep_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms')" || { echo "FAILED: $*"; tail -5 gpurun_out/sweep_err.log; }; }
run --config cora
run --config pubmed
run --config yelp
run --model gat
run --sampler poisson-ladies
BLISS_PIPELINE_FLAGS=0 run --steps 200
run --no-pipeline
run --steps 101
