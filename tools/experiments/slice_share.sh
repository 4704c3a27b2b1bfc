#!/bin/bash
# share of the fan slice's group kernel spent in the beam runs / in the whole per-triangle body: libNORUN.so / libNOBODY.so
# (-DSLICE_EXP_NORUN / -DSLICE_EXP_NOBODY builds: wrong results, timing only) against the in-tree library
for v in ${VARIANTS:-cur NORUN NOBODY NOB}; do
  if [ $v = cur ]; then unset MCL_LIB; else export MCL_LIB=$PWD/tools/experiments/lib$v.so; fi
  for m in mesh-general mesh-soup; do
  python bench.py --only-main --map $m --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v $m', d['ms_per_step'], 'main_us', d['roofline']['launch_us'])"
  done
done
