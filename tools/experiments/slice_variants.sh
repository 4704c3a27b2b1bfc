#!/bin/bash
# the general-mesh legs with tools/experiments/lib<NAME>.so for every NAME given, against the in-tree library (cur)
for rep in 1 2; do
for v in cur "$@"; do
  if [ $v = cur ]; then unset MCL_LIB; else export MCL_LIB=$PWD/tools/experiments/lib$v.so; fi
  for m in mesh-general mesh-soup; do
  python bench.py --only-main --map $m --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v $m', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], d['roofline']['mbes_path'].get('groups_deferred_to_general_kernel'))"
  done
done
done
