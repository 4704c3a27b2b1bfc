import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import bench
from smarc_navigation_amd import engine, synth
m = bench.build_map('mesh')
B, P = 512, 1 << 20
stream = synth.odom_stream(400)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
def q(v, nb):
    lo, w = v.mean() - 4 * v.std(), 8 * v.std() / nb
    return np.clip(np.floor((v - lo) / w).astype(np.int64), 0, nb - 1)
def ancestors(idx):
    # per-slot ancestor after keep / lost / dupes (auv_pf.py:183-198)
    n = idx.size
    first = np.ones(n, bool); first[1:] = idx[1:] != idx[:-1]
    keep = np.zeros(n, bool); keep[idx[first]] = True
    lost = np.nonzero(~keep)[0]
    dupes = idx[~first]
    anc = np.arange(n); anc[lost] = dupes[:lost.size]
    return anc
def run(sort_at, total, track_from):
    e = engine.Engine(P, seed=5, **bench.COV)
    bench.attach_map(e, m)
    e.init_particles()
    order = None
    for k in range(total):
        if k == sort_at:
            st = e.get_particles()
            key = (q(st[0], 32) * 32 + q(st[1], 32)) * 32 + q(st[5], 32)
            order = np.argsort(key, kind='stable')
        if k == track_from and order is None:
            order = np.arange(P)
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, bench.SIGMA, bench.R_MAX)
        if order is not None:
            anc = ancestors(e.last_indices().astype(np.int64))
            pos = np.empty(P, np.int64); pos[order] = np.arange(P)
            order = np.argsort(pos[anc], kind='stable')   # children take their ancestor's place
    e.sync()
    st = e.get_particles()
    if order is None:
        key = (q(st[0], 32) * 32 + q(st[1], 32)) * 32 + q(st[5], 32)
        order = np.argsort(key, kind='stable')
    res = []
    for name, o in (('slot order', np.arange(P)), ('tracked order', order)):
        e.set_particles(st[:, o])
        for rep in range(3): e.update_mbes(ranges[total], ba, bench.SIGMA, bench.R_MAX)
        e.sync(); t0 = time.perf_counter()
        for rep in range(20): e.update_mbes(ranges[total], ba, bench.SIGMA, bench.R_MAX)
        e.sync(); res.append('%s %.1f us' % (name, (time.perf_counter() - t0) / 20 * 1e6))
    e.close()
    return ' | '.join(res)
for T in (0, 1, 4, 16, 48):
    print('sorted at step 120, order tracked through %2d resamplings:' % T, run(120, 120 + T, 10**9), flush=True)
print('never sorted, order tracked from step 0 through 168 resamplings:', run(10**9, 168, 0), flush=True)
