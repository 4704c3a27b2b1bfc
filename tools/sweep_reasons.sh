#!/bin/bash
# tools/sweep_reasons.sh [bench.py args] -- run on the GPU box (gpurun): WHY the fan sweep declines particles.
# Builds a -DSWEEP_REASONS variant of the library (every SWEEP_FAIL / SWEEP_NOTE site counts its code in a 16-word
# table, mcl_sweep.h) and runs a short bench with MCL_DEBUG_WORK=1: one line per update on stderr,
#   [mbes] sweep handed over H of N particles
#   [mbes] declined sides by reason: 1:.. 5:.. 10:..     (1 position / tilt / footprint, 5 no nadir hit, 6-8 degenerate
#                                                          start, 9 under a grid, 10 border re-crossing, 11 step limit,
#                                                          12 seabed above the horizon)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=/tmp/libmcl_reasons.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical \
  -fno-slp-vectorize -DSWEEP_REASONS=1 -o $V $R/smarc_navigation_amd/csrc/mcl_api.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
MCL_LIB=$V MCL_DEBUG_WORK=1 python3 $R/bench.py --only-main --steps 3 --warmup 1 "$@" 2>&1 >/dev/null | grep '^\[mbes\]' | tail -6
