#!/bin/bash
# tools/experiments/reasons_pytest.sh <pytest args> -- on the GPU box: run GPU tests on a -DSWEEP_REASONS build with the
# sweep's per-reason counters of every update on stderr (mcl_sweep.h: SWEEP_NOTE codes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
V=/tmp/libmcl_reasons.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical \
  -fno-slp-vectorize -DSWEEP_REASONS=1 -o $V $R/smarc_navigation_amd/csrc/mcl_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
cd $R && MCL_LIB=$V MCL_DEBUG_WORK=1 python3 -m pytest "$@" -s -q 2>&1 | grep -E "declined|handed|passed|failed|fuzz"
