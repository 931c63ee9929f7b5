#!/bin/bash
# usage: scratch/ab_env.sh <kernel-substring> "ENV=val ..." "ENV=val ..." ...  (two rounds each; "-" = no extra env)
K=$1; shift
for round in 1 2; do for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  env $ee timeout -k 10 300 python bench.py --steps 400 --warmup 40 --cpu-baseline-steps 0 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
ak=o['roofline'].get('all_kernels',{})
print('[$e]', round(o['ms_per_step'],4), 'median', round(o['step_ms_percentiles']['median'],4), {k:v['avg_launch_us'] for k,v in ak.items() if any(x in k for x in '$K'.split(','))})"
done; done
