#!/usr/bin/env python3
"""Step by step over a data gap: the irregular TIN of the bench with the triangles around (1, Y) missing, one line per step --
ms of the step (synchronised), particles the sweep handed over / cast by the fan slice / by the ray traversal, distance of the
truth from the gap.  tools/experiments/tin_hole_trace.py [Y] [first] [last]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

y = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
last = int(sys.argv[3]) if len(sys.argv) > 3 else 110
m = bench.punch_hole(bench.build_map('mesh-tin'), 1.0, y)
P, B = 1048576, 512
stream = synth.odom_stream(last)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
e = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(e, m)
e.init_particles()
for k in range(last):
    e.sync()
    t0 = time.perf_counter()
    e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    ms = 1e3 * (time.perf_counter() - t0)
    if k >= first:
        t = stream['truth'][k]
        print('step %3d  %.3f ms  handed %7d  slice %7d  traversal %7d  truth (%.2f, %.2f) gap at %.2f m, into the gap %d beams' % (
            (k, ms, e.mbes_last_path()[1]) + e.mbes_last_handover() + (t[0], t[1], float(np.hypot(t[0] - 1.0, t[1] - y)), int((ranges[k] >= bench.R_MAX).sum()))), flush=True)
