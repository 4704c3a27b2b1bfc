#!/bin/bash
# tools/experiments/slice_phases.sh -- on the GPU box: where the group slice's time goes on the two soups -- the library as built,
# -DSLICE_EXP_NOBODY (phase B tests every staged triangle but builds no segment) and -DSLICE_EXP_NOB (no phase B at all:
# geometry + staging only).  Prints ms per step of `bench.py --map <m> --only-main` for each.  Round 6, regular / irregular soup:
# staging 0.73 / 1.03 ms, the cut tests 0.37 / 0.60, segments + beam runs 1.50 / 1.21 of 2.60 / 2.84 per launch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
F="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F -DSLICE_EXP_NOBODY=1 -o /tmp/libmcl_nobody.so $R/smarc_navigation_amd/csrc/mcl_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
/opt/rocm/bin/hipcc $F -DSLICE_EXP_NOB=1 -o /tmp/libmcl_nob.so $R/smarc_navigation_amd/csrc/mcl_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for m in mesh-general mesh-soup; do
  for v in $R/smarc_navigation_amd/libmcl_hip.so /tmp/libmcl_nobody.so /tmp/libmcl_nob.so; do
    MCL_LIB=$v python3 $R/bench.py --map $m --only-main --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m', '$(basename $v)', d['ms_per_step'], d['kernels'].get('mbes_main'))"
  done
done
