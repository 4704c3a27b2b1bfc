#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mbes.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -5
for m in mesh grid mesh grid; do
  python bench.py --map $m --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$m', 'steps/s', d['value'], 'mbes_ms', d['kernels']['update_mbes']['avg_ms'], 'rmse', d['pose_rmse_m'])"
done
