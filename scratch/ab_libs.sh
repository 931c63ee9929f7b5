#!/bin/bash
# usage: scratch/ab_libs.sh <kernel-substring> lib1.so lib2.so ...   ("base" = the in-tree build); two rounds each
K=$1; shift
cp bliss_gnn_amd/libbliss_gnn.so /tmp/base.so
for round in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so; else cp $lib bliss_gnn_amd/libbliss_gnn.so; fi
  timeout -k 10 300 python bench.py --steps 400 --warmup 40 --cpu-baseline-steps 0 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
ak=o['roofline'].get('all_kernels',{})
print('$lib', round(o['ms_per_step'],4), 'median', round(o['step_ms_percentiles']['median'],4), {k:v['avg_launch_us'] for k,v in ak.items() if '$K' in k})"
done; done
cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so
