# experiment: how much faster is the sweep when neighbouring slots hold spatial neighbours?
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
print('cloud std', st[0].std(), st[1].std(), st[5].std())
def time_update(tag):
    e.timing_enable(True) if hasattr(e, 'timing_enable') else None
    for rep in range(3):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    t0 = time.perf_counter()
    for rep in range(20):
        e.update_mbes(ranges[120], ba, bench.SIGMA, bench.R_MAX)
    e.sync()
    print(tag, 'update_mbes %.1f us' % ((time.perf_counter() - t0) / 20 * 1e6), e.mbes_last_path())
time_update('slot order after resampling ')
rs = np.random.RandomState(1)
p = rs.permutation(P)
e.set_particles(st[:, p]); time_update('random order                ')
# sort by yaw bin then x then y (coarse bins so that neighbours agree in all three)
def key(st, nb):
    q = lambda v: np.clip(((v - v.min()) / (np.ptp(v) + 1e-12) * nb).astype(np.int64), 0, nb - 1)
    return (q(st[5]) * nb + q(st[0])) * nb + q(st[1])
for nb in (16, 64, 256):
    o = np.argsort(key(st, nb), kind='stable')
    e.set_particles(st[:, o]); time_update('sorted, %3d bins per axis    ' % nb)
