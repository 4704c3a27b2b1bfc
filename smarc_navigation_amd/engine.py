"""Thin Python wrapper over the C ABI handle (include/mcl.h).  numpy in / numpy out; all compute
runs in libmcl_hip.so on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, Odom, Timing, MclError  # noqa: F401

SYSTEMATIC, RESIDUAL, STRATIFIED, MULTINOMIAL, NAIVE = 0, 1, 2, 3, 4
RNG_NATIVE, RNG_REPLAY = 0, 1
WEIGHT_LINEAR_FLOOR, WEIGHT_LOG_SHIFT, WEIGHT_LINEAR = 0, 1, 2


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def make_odom(v, wz, q, z, stamp=0.0):
    o = Odom()
    o.stamp = float(stamp)
    o.v[:] = [float(x) for x in v]
    o.w_z = float(wz)
    o.q[:] = [float(x) for x in q]
    o.z = float(z)
    return o


class Engine(object):
    """One shard of the particle filter on one GPU."""

    def __init__(self, n_particles, init_cov=(0,) * 6, process_cov=(0,) * 6, resample_cov=(0,) * 6,
                 meas_std=1.0, m2o=None, seed=0, rng_mode=RNG_NATIVE, resample_scheme=SYSTEMATIC,
                 device=0, rank=0, world=1, n_global=0, global_offset=0):
        self.lib = _lib.load()
        cfg = Config()
        cfg.n_particles = int(n_particles)
        cfg.n_global = int(n_global)
        cfg.global_offset = int(global_offset)
        cfg.device, cfg.rank, cfg.world = int(device), int(rank), int(world)
        cfg.resample_scheme, cfg.rng_mode, cfg.comm_mode = int(resample_scheme), int(rng_mode), 0
        cfg.seed = int(seed)
        cfg.init_cov[:] = [float(x) for x in init_cov]
        cfg.process_cov[:] = [float(x) for x in process_cov]
        cfg.resample_cov[:] = [float(x) for x in resample_cov]
        cfg.meas_std = float(meas_std)
        m = np.identity(4) if m2o is None else np.asarray(m2o, dtype=np.float64)
        cfg.m2o[:] = [float(x) for x in m.reshape(-1)]
        self.n = int(n_particles)
        self.n_global = int(n_global) if n_global else self.n
        self.h = C.c_void_p()
        _lib.check(self.lib.mcl_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.mcl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        _lib.check(st, self.h)

    # ---- particle lifecycle
    def init_particles(self, normals=None):
        nz = _f64(normals)
        self._ck(self.lib.mcl_init_particles(self.h, _ptr(nz)))

    def predict(self, v, wz, q, z, dt, normals=None, stamp=0.0):
        nz = _f64(normals)
        od = make_odom(v, wz, q, z, stamp)
        self._ck(self.lib.mcl_predict(self.h, C.byref(od), float(dt), _ptr(nz)))

    def update_gps(self, gx, gy):
        self._ck(self.lib.mcl_update_gps(self.h, float(gx), float(gy)))

    def set_map_grid(self, z, origin, res):
        z = _f32(z)
        self._ck(self.lib.mcl_set_map_grid(self.h, _ptr(z), z.shape[0], z.shape[1], float(origin[0]),
                                           float(origin[1]), float(res)))

    def set_map_mesh(self, verts, tris, heightfield=False, general=False, unstructured=False):
        v = _f32(verts)
        t = np.ascontiguousarray(tris, dtype=np.uint32)
        self._ck(self.lib.mcl_set_map_mesh_ex(self.h, _ptr(v), v.shape[0], _ptr(t), t.shape[0],
                                              (1 if heightfield else 0) | (2 if general else 0) |
                                              (4 if unstructured else 0)))

    def update_mbes(self, ranges, beam_angles, sigma, r_max, sensor_offset=None):
        r, a, so = _f32(ranges), _f32(beam_angles), _f64(sensor_offset)
        self._ck(self.lib.mcl_update_mbes(self.h, _ptr(r), _ptr(a), a.size, float(sigma), float(r_max), _ptr(so)))

    def mbes_expected(self, first, count, beam_angles, r_max, sensor_offset=None):
        a, so = _f32(beam_angles), _f64(sensor_offset)
        out = np.zeros((count, a.size), np.float32)
        self._ck(self.lib.mcl_mbes_expected(self.h, int(first), int(count), _ptr(a), a.size, float(r_max),
                                            _ptr(so), _ptr(out)))
        return out

    def set_landmarks(self, xyz):
        a = _f64(xyz)
        self._ck(self.lib.mcl_set_landmarks(self.h, _ptr(a), a.shape[0]))

    def set_landmark_noise(self, cov6=None, Q6=None):
        """Mahalanobis association: per-landmark covariance (n x 6: xx xy xz yy yz zz, map frame) and / or the
        sensor-frame measurement covariance Q (6); both None = isotropic sigma again."""
        c, q = _f64(cov6), _f64(Q6)
        self._ck(self.lib.mcl_set_landmark_noise(self.h, _ptr(c), _ptr(q)))

    def update_landmarks(self, det_xyz, sigma, k=1, gate=11.345, sensor_offset=None, accumulate=False):
        d, so = _f64(det_xyz), _f64(sensor_offset)
        self._ck(self.lib.mcl_update_landmarks(self.h, _ptr(d), d.shape[0], float(sigma), int(k), float(gate),
                                               _ptr(so), 1 if accumulate else 0))

    def update_landmarks_assign(self, det_xyz, sigma, k_cand=8, gate=11.345, new_mh_dist=11.345, sensor_offset=None,
                                accumulate=False, n_keep=0):
        """Landmark update with a global (Hungarian) assignment per particle; returns the assignment of
        the first n_keep particles (n_keep x n_det int32) or None."""
        d, so = _f64(det_xyz), _f64(sensor_offset)
        out = np.zeros((int(n_keep), d.shape[0]), dtype=np.int32) if n_keep else None
        self._ck(self.lib.mcl_update_landmarks_assign(self.h, _ptr(d), d.shape[0], float(sigma), int(k_cand), float(gate),
                                                      float(new_mh_dist), _ptr(so), 1 if accumulate else 0,
                                                      out.ctypes.data if out is not None else None, int(n_keep)))
        return out

    def resample(self, uniforms=None, normals=None):
        u = None if uniforms is None else _f64(np.atleast_1d(uniforms))
        nz = _f64(normals)
        self._ck(self.lib.mcl_resample(self.h, _ptr(u), 0 if u is None else u.size, _ptr(nz)))

    def resample_prepare(self):
        k = C.c_int64(0)
        self._ck(self.lib.mcl_resample_prepare(self.h, C.byref(k)))
        return int(k.value)

    def mean_cov(self):
        mean, yaw, cov = np.zeros(6), np.zeros(1), np.zeros(9)
        self._ck(self.lib.mcl_mean_cov(self.h, _ptr(mean), _ptr(yaw), _ptr(cov)))
        return mean, float(yaw[0]), cov

    def mean_cov_async(self):
        """queue mean/cov on the stream without waiting; read it later with last_mean_cov / mean_history"""
        self._ck(self.lib.mcl_mean_cov_async(self.h))

    def last_mean_cov(self):
        mean, yaw, cov = np.zeros(6), np.zeros(1), np.zeros(9)
        self._ck(self.lib.mcl_last_mean_cov(self.h, _ptr(mean), _ptr(yaw), _ptr(cov)))
        return mean, float(yaw[0]), cov

    def mean_history(self, last_k):
        out = np.zeros((int(last_k), 6))
        self._ck(self.lib.mcl_mean_history(self.h, int(last_k), _ptr(out)))
        return out

    def poses(self):
        out = np.zeros((self.n, 7))
        self._ck(self.lib.mcl_get_poses(self.h, _ptr(out)))
        return out

    # ---- state access
    def get_particles(self, weights=False):
        soa = np.zeros((6, self.n))
        w = np.zeros(self.n) if weights else None
        self._ck(self.lib.mcl_get_particles(self.h, _ptr(soa), _ptr(w)))
        return (soa, w) if weights else soa

    def set_particles(self, soa):
        s = _f64(soa)
        assert s.shape == (6, self.n)
        self._ck(self.lib.mcl_set_particles(self.h, _ptr(s)))

    def get_log_weights(self):
        lw = np.zeros(self.n)
        self._ck(self.lib.mcl_get_log_weights(self.h, _ptr(lw)))
        return lw

    def set_log_weights(self, lw, mode=WEIGHT_LOG_SHIFT):
        a = _f64(lw)
        self._ck(self.lib.mcl_set_log_weights(self.h, _ptr(a), int(mode)))

    def last_indices(self):
        idx = np.zeros(self.n, np.int32)
        self._ck(self.lib.mcl_get_last_indices(self.h, _ptr(idx)))
        return idx

    def last_offspring_cdf(self):
        c = np.zeros(self.n_global, np.uint32)
        self._ck(self.lib.mcl_get_last_offspring_cdf(self.h, _ptr(c)))
        return c

    def fixed_weights(self):
        q = np.zeros(self.n, np.uint64)
        t = C.c_uint64(0)
        self._ck(self.lib.mcl_get_fixed_weights(self.h, _ptr(q), C.addressof(t)))
        return q, int(t.value)

    # ---- fused asynchronous step
    def step_mbes(self, v, wz, q, z, dt, ranges, beam_angles, sigma, r_max, sensor_offset=None):
        od = make_odom(v, wz, q, z)
        r, a, so = _f32(ranges), _f32(beam_angles), _f64(sensor_offset)
        self._ck(self.lib.mcl_step_mbes(self.h, C.byref(od), float(dt), _ptr(r), _ptr(a), a.size, float(sigma),
                                        float(r_max), _ptr(so)))

    def step_mbes_landmarks(self, v, wz, q, z, dt, ranges, beam_angles, sigma, r_max, det_xyz, lm_sigma, k=1, gate=11.345,
                            sensor_offset=None, lm_sensor_offset=None):
        """step_mbes with the landmark observation of the ping accumulated onto the MBES log-likelihood (BASELINE
        config 5): mcl_step_mbes_landmarks"""
        od = make_odom(v, wz, q, z)
        r, a, so = _f32(ranges), _f32(beam_angles), _f64(sensor_offset)
        d, lso = _f64(det_xyz), _f64(lm_sensor_offset)
        self._ck(self.lib.mcl_step_mbes_landmarks(self.h, C.byref(od), float(dt), _ptr(r), _ptr(a), a.size, float(sigma),
                                                  float(r_max), _ptr(so), _ptr(d), d.shape[0], float(lm_sigma), int(k),
                                                  float(gate), _ptr(lso)))

    def sync(self):
        self._ck(self.lib.mcl_sync(self.h))

    # ---- multi-GPU
    def comm_init(self, uid_bytes, overlap=True):
        self._ck(self.lib.mcl_comm_init_ex(self.h, uid_bytes, 0 if overlap else 1))

    def comm_ranks(self):
        """(ranks RCCL connected = all-reduce(sum) of 1, overlapped-gather communicator present)"""
        r, o = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mcl_comm_ranks(self.h, C.byref(r), C.byref(o)))
        return int(r.value), bool(o.value)

    def comm_selftest(self, timeout_ms=20000):
        self._ck(self.lib.mcl_comm_selftest(self.h, int(timeout_ms)))

    def comm_shutdown(self, abort=False):
        self._ck(self.lib.mcl_comm_shutdown(self.h, 1 if abort else 0))

    def exchange_stats(self, reset=False):
        """(particle states sent to peers, lost slots filled) by the sharded resamples since the last reset"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.mcl_exchange_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    def exchange_ops(self, reset=False):
        """(point-to-point operations issued, exchanges) by the sharded resamples since the last reset"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.mcl_exchange_ops(self.h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    # ---- instrumentation
    def timing_enable(self, on=True):
        self._ck(self.lib.mcl_timing_enable(self.h, 1 if on else 0))

    def mbes_last_path(self):
        """(path, handed_over, deferred_groups) of the last MBES update: path 1 = fan sweep, 0 = ray traversal."""
        p, ho, dg = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.mcl_mbes_last_path(self.h, C.byref(p), C.byref(ho), C.byref(dg)))
        return int(p.value), int(ho.value), int(dg.value)

    def mbes_last_handover(self):
        """(by_slice, by_traversal): who cast the particles the last update's sweep handed over (TIN with holes: the fan slice first)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.mcl_mbes_last_handover(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def mbes_visit_order(self):
        """(slots, sorted): slots[p] = state slot of the particle the last fused step's sweep visited at position p."""
        slots = np.empty(self.n, dtype=np.uint32)
        srt = C.c_int32(0)
        self._ck(self.lib.mcl_mbes_visit_order(self.h, _ptr(slots), C.byref(srt)))
        return slots, bool(srt.value)

    def timing_get(self):
        t = Timing()
        self._ck(self.lib.mcl_timing_get(self.h, C.byref(t)))
        return {name: (t.ms[k], t.launches[k]) for k, name in enumerate(_lib.MCL_K_NAMES)}


def comm_unique_id():
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    _lib.check(lib.mcl_comm_unique_id(buf))
    return buf.raw


def group_resample(engines, uniforms=None, normals_per_shard=None):
    lib = _lib.load()
    ns = len(engines)
    hs = (C.c_void_p * ns)(*[e.h for e in engines])
    u = None if uniforms is None else _f64(np.atleast_1d(uniforms))
    keep = None
    nzp = None
    if normals_per_shard is not None:
        keep = [_f64(a) for a in normals_per_shard]
        nzp = (C.c_void_p * ns)(*[a.ctypes.data for a in keep])
    _lib.check(lib.mcl_group_resample(hs, ns, _ptr(u), 0 if u is None else u.size, nzp), engines[0].h)


def group_step_mbes(engines, v, wz, q, z, dt, ranges, beam_angles, sigma, r_max, sensor_offset=None):
    """one fused step of a LOCAL group of shards (mcl_group_step_mbes); read the result with
    engines[0].last_mean_cov()"""
    lib = _lib.load()
    ns = len(engines)
    hs = (C.c_void_p * ns)(*[e.h for e in engines])
    od = make_odom(v, wz, q, z)
    r, a, so = _f32(ranges), _f32(beam_angles), _f64(sensor_offset)
    _lib.check(lib.mcl_group_step_mbes(hs, ns, C.byref(od), float(dt), _ptr(r), _ptr(a), a.size, float(sigma),
                                       float(r_max), _ptr(so)), engines[0].h)


def group_step_mbes_landmarks(engines, v, wz, q, z, dt, ranges, beam_angles, sigma, r_max, det_xyz, lm_sigma, k=1,
                              gate=11.345, sensor_offset=None, lm_sensor_offset=None):
    """group_step_mbes with the landmark observation of the ping on top (mcl_group_step_mbes_landmarks)"""
    lib = _lib.load()
    ns = len(engines)
    hs = (C.c_void_p * ns)(*[e.h for e in engines])
    od = make_odom(v, wz, q, z)
    r, a, so = _f32(ranges), _f32(beam_angles), _f64(sensor_offset)
    d, lso = _f64(det_xyz), _f64(lm_sensor_offset)
    _lib.check(lib.mcl_group_step_mbes_landmarks(hs, ns, C.byref(od), float(dt), _ptr(r), _ptr(a), a.size, float(sigma),
                                                 float(r_max), _ptr(so), _ptr(d), d.shape[0], float(lm_sigma), int(k),
                                                 float(gate), _ptr(lso)), engines[0].h)


def group_mean_cov(engines):
    lib = _lib.load()
    ns = len(engines)
    hs = (C.c_void_p * ns)(*[e.h for e in engines])
    mean, yaw, cov = np.zeros(6), np.zeros(1), np.zeros(9)
    _lib.check(lib.mcl_group_mean_cov(hs, ns, _ptr(mean), _ptr(yaw), _ptr(cov)), engines[0].h)
    return mean, float(yaw[0]), cov


def resample_indices(weights, uniforms, scheme=SYSTEMATIC, device=0):
    """resampling.py free-function form on the GPU (fixed-point CDF)."""
    lib = _lib.load()
    w = _f64(weights)
    u = _f64(np.atleast_1d(uniforms))
    out = np.zeros(w.size, np.int32)
    _lib.check(lib.mcl_resample_indices(int(scheme), _ptr(w), w.size, _ptr(u), u.size, int(device), _ptr(out)))
    return out
