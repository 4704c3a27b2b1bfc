#!/bin/bash
# tools/experiments/tin_hole_reasons.sh [Y first last] -- on the GPU box: the step-by-step trace over a data gap
# (tin_hole_trace.py) on a -DSWEEP_REASONS build, with the sweep's per-reason counters of every update beside it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
V=/tmp/libmcl_reasons.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical \
  -fno-slp-vectorize -DSWEEP_REASONS=1 -o $V $R/smarc_navigation_amd/csrc/mcl_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
MCL_LIB=$V MCL_DEBUG_WORK=1 python3 $R/tools/experiments/tin_hole_trace.py "$@" 2>&1 | grep -E "^step|declined"
