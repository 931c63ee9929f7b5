#!/bin/bash
# usage: scratch/other_configs_r03.sh <tag>   (on the GPU box): one FULL bench line (roofline object included) per other
# configuration, by the same bench.py -> gpurun_out/<tag>/bench_<name>.json, plus a one-line summary file
T=$1; O=gpurun_out/$T; mkdir -p $O
one() {
  name=$1; shift
  timeout -k 10 400 python bench.py --cpu-baseline-steps 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "FAILED: $name"; tail -3 $O/bench_$name.err; return; }
  python -c "
import json
o=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
r=o.get('roofline') or {}
print('$name', round(o['value'],1), o['unit'], round(o['ms_per_step'],4), 'ms/step', o.get('step_ms_percentiles'), '| roofline', r.get('kernel'), round(r.get('frac',0),4))" | tee -a $O/other_configs.txt
}
one cora --config cora
one pubmed --config pubmed
one yelp --config yelp
one gat --model gat --steps 200 --warmup 40
one poisson_ladies --sampler poisson-ladies
one force_dist --force-dist --steps 200 --warmup 40
