#!/bin/bash
# the general-mesh legs (fan slice): every particle on its own / groups of spatial neighbours
for m in mesh-general mesh-soup; do
  for v in 0 1; do
    MCL_SLICE_GROUP=$v python bench.py --only-main --map $m --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$m group=$v', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], {k:round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()}, d['roofline']['mbes_path']['groups_deferred_to_general_kernel'])"
  done
done
