#!/bin/bash
# prev (tools/experiments/libprev.so) vs the in-tree library on the headline leg, interleaved
for rep in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export MCL_LIB=$PWD/tools/experiments/libprev.so; else unset MCL_LIB; fi
    python bench.py --only-main --steps 200 --warmup 20 $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], {k:round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
