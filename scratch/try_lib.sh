#!/bin/bash
# usage: scratch/try_lib.sh lib.so  -- parity subset + bench with a variant library
cp bliss_gnn_amd/libbliss_gnn.so /tmp/base.so; cp $1 bliss_gnn_amd/libbliss_gnn.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or oracle or static_shape or pipelined_two_step" 2>&1 | tail -4
timeout -k 10 300 python bench.py --steps 400 --warmup 40 --cpu-baseline-steps 0 --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
try:
    o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(o['ms_per_step'],4), o['step_ms_percentiles'])
except Exception as e: print('bench failed', e)"
cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so
