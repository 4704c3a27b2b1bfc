# tools/visit_order_experiment.py -- run on the GPU box: how much faster is the MBES update when the 64 particles of a sweep
# wavefront are spatial neighbours?  The headline filter runs 120 steps; then its cloud is written back with the slots
# permuted on the HOST -- random, sorted by a coarse (x, y, yaw) key over mean +- 4 sigma (order inside a bin random),
# sorted only inside chunks of consecutive slots -- and the same update is timed on each.  ("particle order":
# the finding, and what producing such an order on the device cost in round 4.)
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import bench
from smarc_navigation_amd import engine, synth
m = bench.build_map('mesh')
B, P = 512, 1 << 20
stream = synth.odom_stream(200)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
e = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(e, m)
e.init_particles()
for k in range(120):
    e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, bench.SIGMA, bench.R_MAX)
e.sync()
st = e.get_particles()
print('cloud: sigma x %.3f m, y %.3f m, yaw %.4f rad' % (st[0].std(), st[1].std(), st[5].std()))
def time_update(tag):
    for rep in range(3):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    t0 = time.perf_counter()
    for rep in range(20):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    print('%-58s update_mbes %.1f us' % (tag, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
time_update('slot order as the resampling left it')
rs = np.random.RandomState(1)
e.set_particles(st[:, rs.permutation(P)]); time_update('random permutation of the slots')
def q(v, nb):
    lo, w = v.mean() - 4 * v.std(), 8 * v.std() / nb
    return np.clip(np.floor((v - lo) / w).astype(np.int64), 0, nb - 1)
for nb in (16, 32):
    key = (q(st[0], nb) * nb + q(st[1], nb)) * nb + q(st[5], nb)
    for chunk in (4096, 16384, 65536, P):
        o = np.arange(P)
        for c0 in range(0, P, chunk):
            sl = slice(c0, c0 + chunk)
            o[sl] = c0 + np.lexsort((rs.rand(chunk), key[sl]))
        e.set_particles(st[:, o])
        time_update('%2d^3 bins of (x, y, yaw), sorted inside chunks of %7d' % (nb, chunk))
key = q(st[0], 64) * 64 + q(st[1], 64)
e.set_particles(st[:, np.lexsort((rs.rand(P), key))]); time_update('64^2 bins of (x, y) only')
