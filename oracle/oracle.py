"""ctypes binding of the CPU oracle (oracle/libmcl_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package smarc_navigation_amd never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libmcl_oracle.so')

_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_u32p = np.ctypeslib.ndpointer(np.uint32, flags='C_CONTIGUOUS')
_u64p = np.ctypeslib.ndpointer(np.uint64, flags='C_CONTIGUOUS')


def build(force=False):
    src = os.path.join(_HERE, 'mcl_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _SO


class _Grid(C.Structure):
    _fields_ = [('nx', C.c_int), ('ny', C.c_int), ('ox', C.c_double), ('oy', C.c_double),
                ('res', C.c_double), ('z', C.c_void_p)]


def _load():
    # MCL_ORACLE_LIB: another build of the same source (tests/test_host_sanitizers.py: the ASan / UBSan build)
    alt = os.environ.get('MCL_ORACLE_LIB')
    if not alt:
        build()
    L = C.CDLL(alt or _SO)
    d, i, i64, u64, u32, vp = C.c_double, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p
    sig = {
        'orc_euler_from_quat': (None, [_f64p, _f64p]),
        'orc_quat_from_euler': (None, [d, d, d, _f64p]),
        'orc_matrix_from_tf': (None, [_f64p, _f64p, _f64p]),
        'orc_wrap_pi': (d, [d]),
        'orc_add_noise': (None, [i, _f64p, _f64p, _f64p]),
        'orc_predict': (None, [i, _f64p, _f64p, d, _f64p, d, d, _f64p, vp]),
        'orc_gps_weights': (None, [i, _f64p, _f64p, d, d, d, vp, vp]),
        'orc_numpy_pairwise_sum': (d, [vp, i64, i64]),
        'orc_normalise_ref': (None, [i, _f64p]),
        'orc_systematic_ref': (i, [i, _f64p, d, _i32p]),
        'orc_stratified_ref': (i, [i, _f64p, _f64p, _i32p]),
        'orc_multinomial_ref': (i, [i, _f64p, _f64p, _i32p]),
        'orc_naive_ref': (i, [i, _f64p, d, _i32p]),
        'orc_residual_ref': (i, [i, _f64p, _f64p, _i32p]),
        'orc_residual_k': (i, [i, _f64p]),
        'orc_lost_dupes': (i, [i, _i32p, _i32p, _i32p]),
        'orc_reassign': (None, [i, _f64p, i, _i32p, _i32p]),
        'orc_mean_cov': (None, [i, _f64p, _f64p, _f64p, _f64p]),
        'orc_det_exp': (d, [d]),
        'orc_fixed_weights': (u64, [i, _f64p, i, i64, _u64p, vp]),
        'orc_fixed_weights_m': (u64, [i, _f64p, i, i64, d, _u64p, vp]),
        'orc_weight_exponent': (i64, [d]),
        'orc_quantise_log_weight': (u64, [d, i64, i]),
        'orc_systematic_ncum': (None, [i, _u64p, u64, u64, i64, u64, _u32p]),
        'orc_indices_from_ncum': (None, [i64, _u32p, i64, i64, _i32p]),
        'orc_philox4x32': (None, [u32, u32, u32, u32, u32, u32, _u32p]),
        'orc_set_threads': (C.c_int, [C.c_int]),
        'orc_native_normals': (None, [i, i64, u64, u32, u32, _f64p]),
        'orc_native_u53': (u64, [u64, u32]),
        'orc_ray_grid': (d, [C.POINTER(_Grid), _f64p, _f64p, d]),
        'orc_ray_mesh_brute': (d, [_f32p, _u32p, i64, _f64p, _f64p, d]),
        'orc_mesh_build': (vp, [_f32p, i64, _u32p, i64]),
        'orc_mesh_free': (None, [vp]),
        'orc_ray_mesh': (d, [vp, _f64p, _f64p, d]),
        'orc_landmark_update': (None, [i, _f64p, _f64p, _f64p, _f64p, i64, _f64p, i, d, i, d, _f64p]),
        'orc_assign_dense': (d, [i, i, _f64p, _i32p]),
        'orc_landmark_assign_update': (None, [i, _f64p, _f64p, _f64p, _f64p, i64, _f64p, i, d, i, d, d, _f64p, vp]),
        'orc_landmark_update_maha': (None, [i, _f64p, _f64p, _f64p, _f64p, vp, i64, _f64p, i, d, vp, i, d, _f64p]),
        'orc_landmark_assign_update_maha': (None, [i, _f64p, _f64p, _f64p, _f64p, vp, i64, _f64p, i, d, vp, i, d, d, _f64p, vp, vp]),
        'orc_gridmap_add_pings': (None, [i, i, d, d, d, vp, vp, i64, _f64p, _f32p, _f32p, i, d, _f64p, _f64p, vp]),
        'orc_gridmap_finalize': (i64, [i, i, vp, vp, i, _f32p]),
        'orc_mbes_update': (None, [i, _f64p, _f64p, _f64p, i, vp, _f32p, vp, i, d, d, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


_L = _load()


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


# ---- state helpers: tests use (n, 6) row arrays; the oracle uses SoA (6, n)
def to_soa(poses):
    return np.ascontiguousarray(np.asarray(poses, dtype=np.float64).T)


def from_soa(soa):
    return np.ascontiguousarray(soa.T)


def euler_from_quat(q):
    out = np.zeros(3)
    _L.orc_euler_from_quat(_c(q), out)
    return out


def quat_from_euler(r, p, y):
    out = np.zeros(4)
    _L.orc_quat_from_euler(r, p, y, out)
    return out


def matrix_from_tf(t, q):
    out = np.zeros(16)
    _L.orc_matrix_from_tf(_c(t), _c(q), out)
    return out.reshape(4, 4)


def wrap_pi(a):
    return _L.orc_wrap_pi(float(a))


def add_noise(soa, cov, normals):
    _L.orc_add_noise(soa.shape[1], soa, _c(cov), _c(normals))


def predict(soa, v, wz, q, z, dt, pcov, normals=None):
    nz = None if normals is None else _c(normals)
    _L.orc_predict(soa.shape[1], soa, _c(v), float(wz), _c(q), float(z), float(dt), _c(pcov),
                   None if nz is None else nz.ctypes.data)


def gps_weights(soa, m2o, gx, gy, sigma):
    n = soa.shape[1]
    w, lw = np.zeros(n), np.zeros(n)
    _L.orc_gps_weights(n, soa, _c(m2o).reshape(-1), float(gx), float(gy), float(sigma),
                       w.ctypes.data, lw.ctypes.data)
    return w, lw


def numpy_sum(a, stride=1):
    a = _c(a)
    return _L.orc_numpy_pairwise_sum(a.ctypes.data, a.size // stride, stride)


def normalise_ref(w_raw):
    w = _c(w_raw).copy()
    _L.orc_normalise_ref(w.size, w)
    return w


def systematic_ref(w, u):
    w = _c(w)
    idx = np.zeros(w.size, np.int32)
    rc = _L.orc_systematic_ref(w.size, w, float(u), idx)
    return idx, rc


def stratified_ref(w, u):
    w = _c(w)
    idx = np.zeros(w.size, np.int32)
    rc = _L.orc_stratified_ref(w.size, w, _c(u), idx)
    return idx, rc


def multinomial_ref(w, u):
    w = _c(w)
    idx = np.zeros(w.size, np.int32)
    rc = _L.orc_multinomial_ref(w.size, w, _c(u), idx)
    return idx, rc


def naive_ref(w, u01):
    w = _c(w)
    idx = np.zeros(w.size, np.int32)
    rc = _L.orc_naive_ref(w.size, w, float(u01), idx)
    return idx, rc


def residual_k(w):
    w = _c(w)
    return _L.orc_residual_k(w.size, w)


def residual_ref(w, u):
    w = _c(w)
    idx = np.zeros(w.size, np.int32)
    u = _c(u) if len(u) else np.zeros(1)
    k = _L.orc_residual_ref(w.size, w, u, idx)
    return idx, k


def lost_dupes(idx):
    idx = _c(idx, np.int32)
    lost = np.zeros(idx.size, np.int32)
    dupes = np.zeros(idx.size, np.int32)
    nl = _L.orc_lost_dupes(idx.size, idx, lost, dupes)
    return lost[:nl].copy(), dupes[:nl].copy()


def reassign(soa, lost, dupes):
    _L.orc_reassign(soa.shape[1], soa, len(lost), _c(lost, np.int32), _c(dupes, np.int32))


def mean_cov(soa):
    mean, yaw, cov = np.zeros(6), np.zeros(1), np.zeros(9)
    _L.orc_mean_cov(soa.shape[1], soa, mean, yaw, cov)
    return mean, float(yaw[0]), cov


def det_exp(x):
    return _L.orc_det_exp(float(x))


def fixed_weights(lw, mode, n_global=None):
    lw = _c(lw)
    q = np.zeros(lw.size, np.uint64)
    wl = np.zeros(lw.size)
    tot = _L.orc_fixed_weights(lw.size, lw, int(mode), int(n_global or lw.size), q, wl.ctypes.data)
    return q, int(tot), wl


def fixed_weights_shard(lw, mode, n_global, m_lw_global):
    lw = _c(lw)
    q = np.zeros(lw.size, np.uint64)
    tot = _L.orc_fixed_weights_m(lw.size, lw, int(mode), int(n_global), float(m_lw_global), q, None)
    return q, int(tot)


def weight_exponent(m_lw):
    """The integer exponent K log-likelihood weights are quantised relative to (mcl_device.h: weight_exponent)."""
    return int(_L.orc_weight_exponent(float(m_lw)))


def fixed_weights_own_exponent(lw, n_global):
    """A shard of a cloud spread over several processes: its log-likelihood weights at the exponent of its OWN maximum
    (what mcl_resample.h: k_quantise_tiles leaves when it fills a shard record).  Returns (q, K_r); the cloud's weights
    are q >> (max_r K_r - K_r)."""
    lw = _c(lw)
    q, _ = fixed_weights_shard(lw, 1, n_global, float(np.max(lw)) if lw.size else -np.inf)
    return q, weight_exponent(float(np.max(lw)) if lw.size else -np.inf)


def systematic_ncum(q, u53, c_offset=0, total=None, n_global=None):
    q = _c(q, np.uint64)
    if total is None:
        total = int(q.sum(dtype=np.uint64))
    out = np.zeros(q.size, np.uint32)
    _L.orc_systematic_ncum(q.size, q, int(c_offset), int(total), int(n_global or q.size), int(u53), out)
    return out


def indices_from_ncum(ncum, i0=0, cnt=None):
    ncum = _c(ncum, np.uint32)
    cnt = ncum.size - i0 if cnt is None else cnt
    idx = np.zeros(cnt, np.int32)
    _L.orc_indices_from_ncum(ncum.size, ncum, int(i0), int(cnt), idx)
    return idx


def systematic_fixed(lw, mode, u53):
    """Full fixed-point systematic resample of one shard-less filter: indices + ncum."""
    q, tot, _ = fixed_weights(lw, mode)
    ncum = systematic_ncum(q, u53, 0, tot, q.size)
    return indices_from_ncum(ncum), ncum, q


def u_to_u53(u):
    return int(np.floor(float(u) * 9007199254740992.0))


def philox(c, k):
    out = np.zeros(4, np.uint32)
    _L.orc_philox4x32(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return out


def set_threads(nthreads):
    """Host threads for the particle-parallel loops (0 = leave as is); returns the count in effect."""
    return int(_L.orc_set_threads(int(nthreads)))


def native_normals(n, gid0, seed, purpose, step):
    out = np.zeros((n, 6))
    _L.orc_native_normals(n, int(gid0), int(seed), int(purpose), int(step), out)
    return out


def native_u53(seed, step):
    return int(_L.orc_native_u53(int(seed), int(step)))


class Grid(object):
    def __init__(self, z, origin, res):
        self.z = _c(z, np.float32)
        self.s = _Grid(self.z.shape[0], self.z.shape[1], float(origin[0]), float(origin[1]), float(res),
                       self.z.ctypes.data)

    def ray(self, o, d, r_max):
        return _L.orc_ray_grid(C.byref(self.s), _c(o), _c(d), float(r_max))

    def _map(self):
        return 0, C.addressof(self.s)


class Mesh(object):
    def __init__(self, verts, tris):
        self.verts = _c(verts, np.float32)
        self.tris = _c(tris, np.uint32)
        self.h = _L.orc_mesh_build(self.verts, self.verts.shape[0], self.tris, self.tris.shape[0])

    def __del__(self):
        if getattr(self, 'h', None):
            _L.orc_mesh_free(self.h)
            self.h = None

    def ray(self, o, d, r_max):
        return _L.orc_ray_mesh(self.h, _c(o), _c(d), float(r_max))

    def ray_brute(self, o, d, r_max):
        return _L.orc_ray_mesh_brute(self.verts, self.tris, self.tris.shape[0], _c(o), _c(d), float(r_max))

    def _map(self):
        return 1, self.h


def mbes_update(soa, m2o, sensor_off, amap, beam_angles, ranges, sigma, r_max, want_expected=True):
    n = soa.shape[1]
    ba = _c(beam_angles, np.float32)
    B = ba.size
    lw = np.zeros(n)
    ex = np.zeros((n, B)) if want_expected else None
    kind, ptr = amap._map()
    rg = None if ranges is None else _c(ranges, np.float32)
    _L.orc_mbes_update(n, soa, _c(m2o).reshape(-1), _c(sensor_off), kind, ptr, ba,
                       None if rg is None else rg.ctypes.data, B, float(sigma), float(r_max),
                       lw.ctypes.data, None if ex is None else ex.ctypes.data)
    return lw, ex


def landmark_update(soa, m2o, sensor_off, landmarks, det, sigma, k=1, gate=11.345):
    lm, dt = _c(landmarks), _c(det)
    lw = np.zeros(soa.shape[1])
    _L.orc_landmark_update(soa.shape[1], soa, _c(m2o).reshape(-1), _c(sensor_off), lm, lm.shape[0], dt,
                           dt.shape[0], float(sigma), int(k), float(gate), lw)
    return lw


def landmark_update_maha(soa, m2o, sensor_off, landmarks, det, sigma, k=1, gate=11.345, lmcov=None, Q6=None):
    """k-NN update with the reference's Mahalanobis distance (sensor-frame innovation, S = R^T Sigma_j R + Q)"""
    soa, lm, dt = _c(soa), _c(landmarks), _c(det)
    cov = None if lmcov is None else _c(lmcov)
    q = None if Q6 is None else _c(Q6)
    lw = np.zeros(soa.shape[1])
    _L.orc_landmark_update_maha(soa.shape[1], soa, _c(m2o).reshape(-1), _c(sensor_off), lm,
                                cov.ctypes.data if cov is not None else None, lm.shape[0], dt, dt.shape[0], float(sigma),
                                q.ctypes.data if q is not None else None, int(k), float(gate), lw)
    return lw


def landmark_assign_update_maha(soa, m2o, sensor_off, landmarks, det, sigma, k_cand, gate, new_mh_dist, lmcov=None, Q6=None,
                                want_assign=False, want_table=False):
    """global assignment on the reference's Mahalanobis table; want_table: also the (n_lm + D_valid) x D_valid table of
    particle 0 in the reference's own layout (ekf_slam_core.cpp:248-281), for its own Munkres"""
    soa = _c(soa)
    lm = _c(landmarks).reshape(-1, 3)
    dt = _c(det).reshape(-1, 3)
    cov = None if lmcov is None else _c(lmcov)
    q = None if Q6 is None else _c(Q6)
    lw = np.zeros(soa.shape[1])
    asg = np.zeros((soa.shape[1], dt.shape[0]), dtype=np.int32) if want_assign else None
    nv = int(np.sum(~np.isnan(dt).any(axis=1)))
    tab = np.zeros((lm.shape[0] + nv, nv)) if want_table else None
    _L.orc_landmark_assign_update_maha(soa.shape[1], soa, _c(m2o).reshape(-1), _c(sensor_off), lm,
                                       cov.ctypes.data if cov is not None else None, lm.shape[0], dt, dt.shape[0],
                                       float(sigma), q.ctypes.data if q is not None else None, int(k_cand), float(gate),
                                       float(new_mh_dist), lw, asg.ctypes.data if asg is not None else None,
                                       tab.ctypes.data if tab is not None else None)
    out = [lw]
    if want_assign:
        out.append(asg)
    if want_table:
        out.append(tab)
    return out[0] if len(out) == 1 else tuple(out)


def assign_dense(cost):
    """Optimal assignment of every ROW of cost[n, m] (n <= m) to a distinct column; returns
    (col_of_row[n], total)."""
    cost = _c(cost)
    n, m = cost.shape
    col = np.zeros(n, dtype=np.int32)
    total = _L.orc_assign_dense(n, m, cost.reshape(-1), col)
    return col, float(total)


def landmark_assign_update(soa, m2o, sensor_off, landmarks, det, sigma, k_cand, gate, new_mh_dist, want_assign=False):
    soa = _c(soa)
    lm = _c(landmarks).reshape(-1, 3)
    dt = _c(det).reshape(-1, 3)
    lw = np.zeros(soa.shape[1])
    asg = np.zeros((soa.shape[1], dt.shape[0]), dtype=np.int32) if want_assign else None
    _L.orc_landmark_assign_update(soa.shape[1], soa, _c(m2o).reshape(-1), _c(sensor_off), lm, lm.shape[0], dt,
                                  dt.shape[0], float(sigma), int(k_cand), float(gate), float(new_mh_dist), lw,
                                  asg.ctypes.data if asg is not None else None)
    return (lw, asg) if want_assign else lw


_REF_MUNKRES = os.path.join(_HERE, '_ref', 'libref_munkres.so')


def ref_munkres(cost):
    """The REFERENCE's Munkres<double> (oracle/_ref, built from /root/reference/auv_ekf_slam/utils/munkres):
    cost[rows, cols] as ekf_slam_core.cpp builds it (rows = landmarks, cols = measurements);
    returns row_of_col[cols].  None if the reference build is not present."""
    if not os.path.exists(_REF_MUNKRES):
        return None
    lib = C.CDLL(_REF_MUNKRES)
    cost = _c(cost)
    r, c = cost.shape
    out = np.zeros(c, dtype=np.int32)
    lib.ref_munkres_solve.argtypes = [C.c_int, C.c_int, _f64p, _i32p]
    lib.ref_munkres_solve(r, c, cost.reshape(-1), out)
    return out


class GridMapBuilder(object):
    """Self-oracle of include/mcl_map.h (fixed-point accumulators, order-free)."""

    def __init__(self, nx, ny, origin, res):
        self.nx, self.ny, self.origin, self.res = int(nx), int(ny), (float(origin[0]), float(origin[1])), float(res)
        self.sum = np.zeros(self.nx * self.ny, dtype=np.int64)
        self.cnt = np.zeros(self.nx * self.ny, dtype=np.uint32)

    def add_pings(self, poses6, ranges, beam_angles, r_max, m2o=None, sensor_off=None, want_points=False):
        poses6 = _c(poses6).reshape(-1, 6)
        ranges = np.ascontiguousarray(ranges, dtype=np.float32).reshape(poses6.shape[0], -1)
        ba = np.ascontiguousarray(beam_angles, dtype=np.float32)
        m2o = _c(np.identity(4) if m2o is None else m2o).reshape(-1)
        so = _c([0.0] * 6 if sensor_off is None else sensor_off)
        pts = np.zeros((poses6.shape[0], ba.size, 3)) if want_points else None
        _L.orc_gridmap_add_pings(self.nx, self.ny, self.origin[0], self.origin[1], self.res, self.sum.ctypes.data,
                                 self.cnt.ctypes.data, poses6.shape[0], poses6.reshape(-1), ranges.reshape(-1), ba,
                                 ba.size, float(r_max), m2o, so, pts.ctypes.data if pts is not None else None)
        return pts

    def finalize(self, fill_passes=0):
        z = np.zeros(self.nx * self.ny, dtype=np.float32)
        empty = _L.orc_gridmap_finalize(self.nx, self.ny, self.sum.ctypes.data, self.cnt.ctypes.data, int(fill_passes), z)
        return z.reshape(self.nx, self.ny), int(empty)
