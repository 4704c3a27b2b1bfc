#!/usr/bin/env python3
"""Soak of BASELINE config 5's fused step (visiting order + landmark k-NN on top of the fan sweep): STEPS steps at 524 288
particles, the separate calls beside it on a second engine for the first CHECK steps (bitwise), then mean error / finiteness
every 100 steps.  tools/experiments/soak_config5.py [STEPS] [CHECK]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 60
P, B = 524288, 512
m = bench.build_map('mesh')
stream = synth.odom_stream(steps)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
lm = synth.landmark_map(4096, (-64.0, -354.0, 643.0, 353.0))
rs = np.random.RandomState(8)
a = engine.Engine(P, seed=5, **bench.COV)
b = engine.Engine(P, seed=5, **bench.COV)
for e in (a, b):
    bench.attach_map(e, m)
    e.set_landmarks(lm)
    e.init_particles()
worst = 0.0
for k in range(steps):
    t = stream['truth'][k]
    T = synth.rigid_matrix(*t)
    near = lm[np.argsort(np.sum((lm[:, :2] - t[:2]) ** 2, axis=1))[:16]]
    det = (near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(16, 3)
    if k % 7 == 3:
        det[rs.randint(16)] = np.nan
    od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
    a.step_mbes_landmarks(*od, ranges[k], ba, bench.SIGMA, bench.R_MAX, det, 0.3, k=4, gate=11.345)
    if k < check:
        b.predict(*od)
        b.update_mbes(ranges[k], ba, bench.SIGMA, bench.R_MAX)
        b.update_landmarks(det, 0.3, k=4, gate=11.345, accumulate=True)
        b.resample()
        assert np.array_equal(a.get_particles(), b.get_particles()), k
        assert np.array_equal(a.last_indices(), b.last_indices()), k
    if k % 100 == 99 or k == steps - 1:
        mean, _, c9 = a.last_mean_cov()
        err = float(np.hypot(mean[0] - t[0], mean[1] - t[1]))
        worst = max(worst, err)
        path = a.mbes_last_path()
        assert np.all(np.isfinite(mean)) and np.all(np.isfinite(c9)), (k, mean)
        assert err < 1.0, (k, err)
        print('step %d: mean error %.3f m, sigma %.3f x %.3f m, path %s' % (k + 1, err, np.sqrt(c9[0]), np.sqrt(c9[4]), path), flush=True)
print('soak ok: %d steps, worst mean error %.3f m' % (steps, worst))
