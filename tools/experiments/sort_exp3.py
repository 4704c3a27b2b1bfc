import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import bench
from smarc_navigation_amd import engine, synth
os.environ['MCL_VISIT_ORDER'] = '0'
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
def time_update(tag):
    for rep in range(3):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    t0 = time.perf_counter()
    for rep in range(20):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    print(tag, 'update_mbes %.1f us' % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
time_update('slot order                      ')
rs = np.random.RandomState(1)
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
        e.set_particles(st[:, o]); time_update('bins %2d^3, sorted inside chunks of %7d' % (nb, chunk))
# proc noise effect: keys from a state perturbed like one predict step
print('process cov', bench.COV.get('process_cov'))
