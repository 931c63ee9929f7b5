#!/bin/bash
# usage: scratch/ab_env800.sh "ENV=val ..." ...  (800-step windows, two rounds each; "-" = no extra env)
for round in 1 2; do for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  env $ee timeout -k 10 300 python bench.py --steps 800 --warmup 40 --cpu-baseline-steps 0 --no-roofline 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', round(o['ms_per_step'],4), o['step_ms_percentiles'])"
done; done
