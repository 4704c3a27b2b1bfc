#!/usr/bin/env python3
"""Soak of the half-edge TIN sweep (round 6): STEPS fused steps at 1 048 576 particles x 512 beams on the irregular TIN handed
over in RANDOM order, the same filter on the TIN as generated beside it for the first CHECK steps (mean pose within 5 mm, the bound BASELINE.md 4 sets against the oracle filter: the
two tables hold the same surface; a nadir on a shared edge may start in either triangle), then every 100 steps: the sweep
cast every particle (no hand-over), the mean is finite and within a metre of the truth.  On the way the track leaves the
converged regime nowhere, crosses ~60 m of map and turns twice.  tools/experiments/soak_tin.py [STEPS] [CHECK]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P, B = 1048576, 512
m = bench.build_map('mesh-tin-shuffled')
m0 = bench.build_map('mesh-tin')
stream = synth.odom_stream(steps)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m0, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
a = engine.Engine(P, seed=5, **bench.COV)
b = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(a, m)
bench.attach_map(b, m0)
a.init_particles()
b.init_particles()
worst, handed = 0.0, 0
for k in range(steps):
    od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
    a.step_mbes(*od, ranges[k], ba, bench.SIGMA, bench.R_MAX)
    if k < check:
        b.step_mbes(*od, ranges[k], ba, bench.SIGMA, bench.R_MAX)
        ma, mb = a.last_mean_cov()[0], b.last_mean_cov()[0]
        assert np.abs(ma - mb).max() < 5e-3, (k, ma, mb)   # (a nadir on a shared edge starts in either triangle: last-bit differences in a few log-likelihoods move a few resampling decisions)
    if k % 100 == 99 or k == steps - 1:
        t = stream['truth'][k]
        mean, _, c9 = a.last_mean_cov()
        err = float(np.hypot(mean[0] - t[0], mean[1] - t[1]))
        worst = max(worst, err)
        path = a.mbes_last_path()
        handed += path[1]
        assert path[0] == 1, path
        assert np.all(np.isfinite(mean)) and np.all(np.isfinite(c9)), (k, mean)
        assert err < 1.0, (k, err)
        print('step %d: mean error %.3f m, sigma %.3f x %.3f m, path %s' % (k + 1, err, np.sqrt(c9[0]), np.sqrt(c9[4]), path), flush=True)
print('soak ok: %d steps on the shuffled TIN, worst mean error %.3f m, particles handed over at the sampled steps: %d' % (steps, worst, handed))
