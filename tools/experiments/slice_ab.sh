#!/bin/bash
# the general-mesh legs (fan slice) with and without the visiting order
for m in mesh-general mesh-soup; do
  for v in 0 1; do
    MCL_VISIT=$v python bench.py --only-main --map $m --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$m visit=$v', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], {k:round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
