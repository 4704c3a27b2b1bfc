#!/usr/bin/env python3
"""Soak of the hole-crossing TIN walk (round 6): the irregular TIN of the bench with GAPS of random size (0.4 .. 2.5 m, one per
~ 6 x 6 m of map: ~ 14 000 of them) under and around the track, STEPS fused steps at 1 048 576 particles x 512 beams.  For the first CHECK
steps the same filter runs beside it with the rims NOT linked (MCL_TIN_RIMS=0: every slice that reaches a gap hands its
particle to the fan slice -- independent code that is exact across gaps): the two mean poses must agree within 5 mm.  Then
every 50 steps: finite mean within a metre of the truth, how many particles the sweep handed over, ms per step.
tools/experiments/soak_tin_holes.py [STEPS] [CHECK]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
check = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P, B = 1048576, 512
m = bench.punch_gaps(bench.build_map('mesh-tin'))
print(m['desc'], flush=True)
stream = synth.odom_stream(steps)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
print('pings: %.1f beams of %d look into a gap on average' % (float((ranges >= bench.R_MAX).sum(axis=1).mean()), B), flush=True)
a = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(a, m)
os.environ['MCL_TIN_RIMS'] = '0'
b = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(b, m)
del os.environ['MCL_TIN_RIMS']
a.init_particles()
b.init_particles()
worst, handed, t0 = 0.0, 0, time.perf_counter()
for k in range(steps):
    od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
    a.step_mbes(*od, ranges[k], ba, bench.SIGMA, bench.R_MAX)
    if k < check:
        b.step_mbes(*od, ranges[k], ba, bench.SIGMA, bench.R_MAX)
        ma, mb = a.last_mean_cov()[0], b.last_mean_cov()[0]
        print('step %d: rims %s / no rims %s handed over; |mean difference| %.2e m' % (k + 1, a.mbes_last_path()[1], b.mbes_last_path()[1], np.abs(ma - mb)[:3].max()), flush=True)
        assert np.abs(ma - mb).max() < 5e-3, (k, ma, mb)
        if k == check - 1:
            b.close()
            a.sync()
            t0 = time.perf_counter()
    if k >= check and (k % 50 == 49 or k == steps - 1):
        a.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 50.0
        t = stream['truth'][k]
        mean, _, c9 = a.last_mean_cov()
        err = float(np.hypot(mean[0] - t[0], mean[1] - t[1]))
        worst = max(worst, err)
        path = a.mbes_last_path()
        handed += path[1]
        assert path[0] == 1, path
        assert np.all(np.isfinite(mean)) and np.all(np.isfinite(c9)), (k, mean)
        assert err < 1.0, (k, err)
        print('step %d: %.3f ms per step, mean error %.3f m, sigma %.3f x %.3f m, handed over %d (slice %d, traversal %d)' % (
            (k + 1, ms, err, np.sqrt(c9[0]), np.sqrt(c9[4]), path[1]) + a.mbes_last_handover()), flush=True)
        t0 = time.perf_counter()
print('soak ok: %d steps on the TIN with gaps, worst mean error %.3f m, particles handed over at the sampled steps: %d' % (steps, worst, handed))
