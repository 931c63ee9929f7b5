#!/bin/bash
# usage: ENVV="A=1" scratch/ab_libs_env.sh lib1 lib2 ... ("base" = in-tree), 800-step windows, two rounds
cp bliss_gnn_amd/libbliss_gnn.so /tmp/base.so
for round in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so; else cp $lib bliss_gnn_amd/libbliss_gnn.so; fi
  env $ENVV timeout -k 10 300 python bench.py --steps 800 --warmup 40 --cpu-baseline-steps 0 --no-roofline 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ENVV $lib', round(o['ms_per_step'],4), o['step_ms_percentiles'])"
done; done
cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so
