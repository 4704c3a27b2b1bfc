"""Shared test helpers: golden loading and the scenario replay driver.

The replay driver re-enacts what oracle/ref_harness/gen_golden.py did to the reference node
(odom_callback per sample, update+resample per GPS fix, loc_loop per publish tick) against a
`backend` object, regenerating the reference's RNG draws from the recorded seed
(numpy legacy RandomState stream, consumption order SURVEY.md A.1)."""
import os

import numpy as np

from smarc_navigation_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def host_threads():
    """Host threads worth giving the oracle's OpenMP loops: the affinity mask capped by the cgroup CPU quota (the GPU
    box shows 256 hardware threads but grants 16 CPUs of time; oversubscribing them is slower)."""
    cores = len(os.sched_getaffinity(0))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(-(-int(quota) // int(period)))))
    except (IOError, ValueError):
        pass
    return cores


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


class OracleBackend(object):
    """Reference-exact CPU oracle (fp64, reference resamplers)."""

    def __init__(self, g, resampler):
        from oracle import oracle as orc
        self.orc = orc
        self.n = int(g['n'])
        self.m2o = g['m2o']
        self.meas_std = float(g['meas_std'])
        self.motion_cov = g['motion_cov']
        self.res_cov = g['res_cov']
        self.resampler = resampler
        self.soa = np.zeros((6, self.n))
        self.last_indices = None
        self.last_w_raw = None
        self.last_w_norm = None

    def init(self, init_cov, normals):
        self.orc.add_noise(self.soa, init_cov, normals)

    def predict(self, v, wz, q, z, dt, normals):
        self.orc.predict(self.soa, v, wz, q, z, dt, self.motion_cov, normals)

    def update_resample(self, gx, gy, rs):
        w_raw, _ = self.orc.gps_weights(self.soa, self.m2o, gx, gy, self.meas_std)
        self.last_w_raw = w_raw + 1e-200
        w = self.orc.normalise_ref(w_raw)
        self.last_w_norm = w
        if self.resampler == 'systematic':
            idx, rc = self.orc.systematic_ref(w, rs.random_sample())
            assert rc == 0
        else:
            k = self.orc.residual_k(w)
            idx, k2 = self.orc.residual_ref(w, rs.random_sample(self.n - k))
            assert k == k2
        self.last_indices = idx
        lost, dupes = self.orc.lost_dupes(idx)
        self.orc.reassign(self.soa, lost, dupes)
        self.orc.add_noise(self.soa, self.res_cov, rs.randn(self.n, 6))

    def state(self):
        return self.orc.from_soa(self.soa)

    def mean_cov(self):
        return self.orc.mean_cov(self.soa)


def replay(g, backend, check=None):
    """Runs the golden scenario on `backend`; returns dict of produced observables."""
    n, n_steps = int(g['n']), int(g['n_steps'])
    stream = synth.odom_stream(n_steps)
    rs = np.random.RandomState(int(g['seed']))
    backend.init(g['init_cov'], rs.randn(n, 6))
    out = dict(init_state=backend.state(), ckpt_states=[], mean=[], yaw=[], cov=[], post_update_states=[],
               indices=[], weights_raw=[], weights_norm=[])
    fix_idx = list(g['fix_idx'])
    fix_xy = g['fix_xy_map']
    pub_every = int(g['pub_every'])
    fp = 0
    old_t = stream['t0']
    for k in range(n_steps):
        t = stream['stamp'][k]
        backend.predict(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], t - old_t,
                        rs.randn(n, 6))
        old_t = t
        if fp < len(fix_idx) and fix_idx[fp] == k:
            backend.update_resample(fix_xy[fp][0], fix_xy[fp][1], rs)
            out['post_update_states'].append(backend.state())
            out['indices'].append(backend.last_indices)
            out['weights_raw'].append(backend.last_w_raw)
            out['weights_norm'].append(backend.last_w_norm)
            fp += 1
        if (k + 1) % pub_every == 0 or k == n_steps - 1:
            m, y, c = backend.mean_cov()
            out['mean'].append(m)
            out['yaw'].append(y)
            out['cov'].append(c)
        if (k + 1) % 25 == 0 or k == n_steps - 1:
            out['ckpt_states'].append(backend.state())
    return out
