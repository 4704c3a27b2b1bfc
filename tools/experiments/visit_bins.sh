#!/bin/bash
# same-box sweep of the visiting order's bin layout (MCL_VISIT_BINS) on the headline leg
tools/kstats.sh 2>&1 | grep -E "k_predict_pose|k_resample_gather|k_visit_scan|k_mbes_sweep|k_cdf|k_quant|k_mbes_cast"
for b in "$@"; do
  export MCL_VISIT_BINS=$b
  python bench.py --only-main --steps 200 --warmup 20 $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('bins $b', d['ms_per_step'], 'main_us', d['roofline']['launch_us'], {k:round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})"
done
