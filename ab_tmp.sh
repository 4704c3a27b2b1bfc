#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mbes.py tests/test_gpu_edge_cases.py -q -m gpu -x > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -5
for extra in "" "--mesh-general"; do
  python bench.py --map mesh $extra --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('mesh $extra', 'steps/s', d['value'], 'mbes_ms', d['kernels']['update_mbes']['avg_ms'], d['roofline'].get('valu_insts_per_ray'))"
done
