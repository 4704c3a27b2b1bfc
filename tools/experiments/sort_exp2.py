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
def spread(v, bits):
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out
for name, key in (
    ('lex yaw16 x32 y64', lambda: (q(st[5], 16) * 32 + q(st[0], 32)) * 64 + q(st[1], 64)),
    ('lex yaw32 x32 y32', lambda: (q(st[5], 32) * 32 + q(st[0], 32)) * 32 + q(st[1], 32)),
    ('lex x32 y32 yaw32', lambda: (q(st[0], 32) * 32 + q(st[1], 32)) * 32 + q(st[5], 32)),
    ('lex yaw8 x64 y64 ', lambda: (q(st[5], 8) * 64 + q(st[0], 64)) * 64 + q(st[1], 64)),
    ('morton 5-5-5      ', lambda: spread(q(st[0], 32), 5) | (spread(q(st[1], 32), 5) << 1) | (spread(q(st[5], 32), 5) << 2)),
    ('morton 6-6-6      ', lambda: spread(q(st[0], 64), 6) | (spread(q(st[1], 64), 6) << 1) | (spread(q(st[5], 64), 6) << 2)),
    ('lex yaw64 only    ', lambda: q(st[5], 64)),
    ('lex x64 y64 only  ', lambda: q(st[0], 64) * 64 + q(st[1], 64)),
):
    k = key()
    o = np.lexsort((rs.rand(P), k))   # random order inside a bin
    e.set_particles(st[:, o]); time_update('%s bins used %6d' % (name, len(np.unique(k))))
