#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mbes.py -q -m gpu -x 2>&1 | tail -3
for m in mesh grid; do
  python bench.py --map $m --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m', 'steps/s', d['value'], 'mbes_ms', d['kernels']['update_mbes']['avg_ms'], 'err', d['pose_error_m'])"
done
