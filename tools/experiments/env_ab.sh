#!/bin/bash
# usage: tools/experiments/env_ab.sh VAR v1 v2 ...  -- the in-tree library with VAR set to each value in turn
VAR=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    export $VAR=$v
    python bench.py --only-main --steps 200 --warmup 20 $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$VAR=$v', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], {k:round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
