#!/usr/bin/env python3
"""Is a long run of fused steps fed fast enough by the host?  Enqueue time per step against GPU time per step
(tools/experiments/host_feed.py [steps])."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
P, B = 1048576, 512
m = bench.build_map('mesh')
stream = synth.odom_stream(steps + 50)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
e = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(e, m)
e.init_particles()


def step(k):
    e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, bench.SIGMA, bench.R_MAX)


for k in range(50):
    step(k)
e.sync()
for n in (50, 200, 1000, steps):
    t0 = time.perf_counter()
    for k in range(50, 50 + n):
        step(k)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print('%5d steps: enqueue %.4f ms/step, until synchronised %.4f ms/step' % (n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n), flush=True)
