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


def outliers_explained(orc, omap, soa, ba, got, ref, r_max, m2o=None, off=None, tol=1e-3, delta=1e-3, label=''):
    """The a15 tolerance contract where it is relaxed (fp32 expected range within 1e-3 m of the fp64 oracle).
    A beam that grazes a crest or a triangle edge is ill-conditioned: the last bit of fp32 decides between the crest and
    the shadow behind it, and the range jumps by metres.  Tests therefore tolerate a bounded NUMBER of rays beyond the
    tolerance -- and this function bounds HOW FAR off such a ray may be: every ray further than `tol` from the oracle
    must be, within `tol`, an answer the fp64 oracle itself gives when the sensor moves by `delta` (1 mm) along one of the
    six axis directions (or lie between those answers when they span a continuous branch, grazing incidence without a
    hit / miss flip).  Asserts it ray by ray; returns (number of outliers, their largest deviation from the unperturbed
    oracle, the largest jump between neighbouring beams of the oracle's own profile at an outlier).  Only the particles
    that own an outlier are cast again."""
    m2o = np.identity(4) if m2o is None else m2o
    off = [0.0] * 6 if off is None else off
    err = np.abs(got - ref)
    bad = err > tol
    if not bad.any():
        return 0, 0.0, 0.0
    rows = np.unique(np.nonzero(bad)[0])
    sub = np.ascontiguousarray(soa[:, rows])
    cands = [ref[rows]]
    for axis in range(3):
        for sgn in (1.0, -1.0):
            s = sub.copy()
            s[axis] += sgn * delta
            _, ex = orc.mbes_update(s, m2o, off, omap, ba, None, 0.2, r_max)
            cands.append(ex)
    cands = np.stack(cands)                                    # 7 x rows x B
    g = got[rows]
    lo, hi = cands.min(axis=0), cands.max(axis=0)
    # candidates closer than 10 cm to one another span a continuous branch (grazing incidence: d range / d shift of 30
    # and more, without a hit / miss flip): any value between them is an oracle answer for a shift below delta.  The
    # seven candidates are clustered by that rule and the GPU value measured against the nearest cluster's interval
    cs = np.sort(cands, axis=0)
    cid = np.concatenate([np.zeros((1,) + cs.shape[1:], int), np.cumsum(np.diff(cs, axis=0) >= 0.1, axis=0)])
    dev = np.full(g.shape, np.inf)
    for k in range(cs.shape[0]):
        mk = cid == k
        if not mk.any():
            break
        clo = np.where(mk, cs, np.inf).min(axis=0)
        chi = np.where(mk, cs, -np.inf).max(axis=0)
        dev = np.minimum(dev, np.where(np.isfinite(clo), np.abs(g - np.clip(g, clo, chi)), np.inf))
    b = bad[rows]
    # the local shadow jump of the oracle's own range profile: an outlier sits where neighbouring beams differ by more
    # than the outlier is off, or where the ray's own answer moves under the 1 mm shift
    prof = ref[rows]
    jump = np.zeros_like(prof)
    if prof.shape[1] > 1:
        d = np.abs(np.diff(prof, axis=1))
        jump[:, :-1] = d
        jump[:, 1:] = np.maximum(jump[:, 1:], d)
    jump = np.maximum(jump, hi - lo)
    worst = np.unravel_index(np.argmax(np.where(b, dev, -1.0)), dev.shape)
    assert dev[b].max() <= tol, '%s ray (particle %d, beam %d): GPU %.5f m, oracle %.5f m, oracle under 1 mm shifts %s' % (
        label, rows[worst[0]], worst[1], g[worst], prof[worst], np.array2string(cands[(slice(None),) + worst], precision=4))
    unexplained = b & (err[rows] > jump + tol)
    assert not unexplained.any(), '%s: %d outlier rays are further off than the local jump of the oracle profile' % (label, int(unexplained.sum()))
    print('%s: %d of %d rays beyond %.0e m, the worst off by %.3e m; all are oracle answers under a 1 mm shift of the sensor '
          '(local jump of the oracle profile there: up to %.2f m)' % (label, int(bad.sum()), err.size, tol, err[bad].max(), jump[b].max()))
    return int(bad.sum()), float(err[bad].max()), float(jump[b].max())


def lw_outliers_explained(orc, omap, soa, ba, ranges, sigma, r_max, lw_got, lw_ref, m2o=None, off=None, delta=1e-3, label='',
                          select=None):
    """The log-likelihood side of the same contract (include/mcl.h, mcl_update_mbes: |d| <= 1e-2 or 2e-4 |lw|): a particle
    outside it must own a ray that flips between a crest and its shadow, i.e. its log-likelihood must lie inside the
    interval the fp64 oracle spans when the sensor moves by `delta` (1 mm) along the six axis directions -- widened by
    the usual tolerance.  Asserts it for every such particle; returns their number."""
    m2o = np.identity(4) if m2o is None else m2o
    off = [0.0] * 6 if off is None else off
    d = np.abs(lw_got - lw_ref)
    out = ~((d <= 1e-2) | (d <= 2e-4 * np.abs(lw_ref))) if select is None else np.asarray(select, bool)
    if not out.any():
        return 0
    rows = np.nonzero(out)[0]
    sub = np.ascontiguousarray(soa[:, rows])
    cands = [lw_ref[rows]]
    for axis in range(3):
        for sgn in (1.0, -1.0):
            s = sub.copy()
            s[axis] += sgn * delta
            lw_c, _ = orc.mbes_update(s, m2o, off, omap, ba, ranges, sigma, r_max)
            cands.append(lw_c)
    cands = np.stack(cands)
    lo, hi = cands.min(axis=0), cands.max(axis=0)
    slack = 1e-2 + 2e-4 * np.maximum(np.abs(lo), np.abs(hi))
    g = lw_got[rows]
    inside = (g >= lo - slack) & (g <= hi + slack)
    assert inside.all(), '%s: particle %d: log-likelihood %.4f outside the oracle interval [%.4f, %.4f] under 1 mm shifts' % (
        label, rows[np.argmin(inside)], g[np.argmin(inside)], lo[np.argmin(inside)], hi[np.argmin(inside)])
    print('%s: %d particles beyond the log-likelihood tolerance (worst |d| %.3e), each inside the oracle interval under a 1 mm '
          'shift of the sensor (widest interval %.2f)' % (label, rows.size, d[out].max(), (hi - lo).max()))
    return int(rows.size)


LIVE_SPAN = 30.0   # a particle's fixed-point weight q = floor(exp(lw - max) 2^s) is non-zero only for lw > max - s ln 2 (s = 63 - ceil(log2 N) = 43 at N = 2^20 -> 29.8); 30 covers every N >= 2^20


def live_particle_contract(orc, omap, soa, ba, ranges, sigma, r_max, lw_got, lw_ref, lw_max, m2o=None, off=None, label='',
                           allow=None):
    """The a15 contract WHERE IT BITES (include/mcl.h beside mcl_update_mbes): the relative arm of the
    log-likelihood tolerance (2e-4 |lw|) may only ever matter for particles that cannot receive offspring.  For every
    checked particle that is LIVE -- log-likelihood within LIVE_SPAN of the cloud's maximum `lw_max`, in the GPU's or in
    the oracle's arithmetic: its fixed-point weight is non-zero -- the absolute bound |d lw| <= 1e-2 must
    hold on its own.  The only exception is a particle that owns a ray grazing a crest (the last bit of fp32 decides
    between the crest and its shadow): it must lie inside the oracle's own interval under 1 mm shifts of the sensor, and
    there may be at most `allow` of them (default: 1 per 2 000 live particles, at least 1).  Prints the measured maximum
    on the live set; returns (live particles, max |d| among them, exceptions)."""
    d = np.abs(lw_got - lw_ref)
    live = (lw_got >= lw_max - LIVE_SPAN) | (lw_ref >= lw_max - LIVE_SPAN)
    n_live = int(live.sum())
    if n_live == 0:
        print('%s: no live particle (lw >= max - %.0f) among the %d checked' % (label, LIVE_SPAN, d.size))
        return 0, 0.0, 0
    viol = live & ~(d <= 1e-2)
    within = d[live & ~viol]
    print('%s: %d live particles (lw >= max - %.0f) of %d checked: max |dlw| %.3e on the live set (bound 1e-2)%s' % (
        label, n_live, LIVE_SPAN, d.size, within.max() if within.size else 0.0,
        '' if not viol.any() else '; %d beyond it (worst %.3e) -- must own a grazing ray' % (int(viol.sum()), d[viol].max())))
    n_exc = 0
    if viol.any():
        allow = max(1, n_live // 2000) if allow is None else allow
        assert viol.sum() <= allow, '%s: %d live particles beyond |dlw| <= 1e-2 (allowed %d)' % (label, int(viol.sum()), allow)
        n_exc = lw_outliers_explained(orc, omap, soa, ba, ranges, sigma, r_max, lw_got, lw_ref, m2o=m2o, off=off,
                                      label=label + ' (live)', select=viol)
    return n_live, float(within.max()) if within.size else 0.0, n_exc


def live_picks(lw, count, seed=0):
    """Indices of up to `count` LIVE particles of a cloud (lw >= max - LIVE_SPAN): the best ones and a random draw of the rest."""
    live = np.flatnonzero(lw >= np.max(lw) - LIVE_SPAN)
    if live.size <= count:
        return live
    best = live[np.argsort(lw[live])[-(count // 4):]]
    rest = np.setdiff1d(live, best)
    return np.concatenate([best, np.random.RandomState(seed).choice(rest, count - best.size, replace=False)])
