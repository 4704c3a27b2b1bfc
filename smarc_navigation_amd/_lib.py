"""ctypes loader for libmcl_hip.so (the C ABI in include/mcl.h, mcl_dr.h and mcl_map.h).

Fails loudly when the shared library is missing: there is no Python/CPU fallback for the hot
path.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C smarc_navigation_amd/csrc`."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('MCL_LIB', os.path.join(_HERE, 'libmcl_hip.so'))  # MCL_LIB: kernel-variant A/B runs

MCL_K_NAMES = ['predict', 'update_gps', 'update_mbes', 'normalise', 'scan', 'resample', 'mean_cov', 'noise', 'comm',
               'mbes_main', 'comm_records', 'pack', 'comm_p2p', 'comm_moments', 'update_landmarks']


class MclError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, 'mcl status %d: %s' % (status, msg))
        self.status = status


class Config(C.Structure):
    _fields_ = [('n_particles', C.c_int64), ('n_global', C.c_int64), ('global_offset', C.c_int64),
                ('device', C.c_int32), ('rank', C.c_int32), ('world', C.c_int32),
                ('resample_scheme', C.c_int32), ('rng_mode', C.c_int32), ('comm_mode', C.c_int32),
                ('seed', C.c_uint64), ('init_cov', C.c_double * 6), ('process_cov', C.c_double * 6),
                ('resample_cov', C.c_double * 6), ('meas_std', C.c_double), ('m2o', C.c_double * 16)]


class Odom(C.Structure):
    _fields_ = [('stamp', C.c_double), ('v', C.c_double * 3), ('w_z', C.c_double), ('q', C.c_double * 4),
                ('z', C.c_double)]


class DrConfig(C.Structure):
    _fields_ = [('dvl_period', C.c_double), ('dr_period', C.c_double)]


class DrOdom(C.Structure):
    _fields_ = [('published', C.c_int32), ('used_dvl', C.c_int32), ('t_now', C.c_double), ('pos', C.c_double * 3),
                ('q', C.c_double * 4), ('rpy', C.c_double * 3), ('lin_vel', C.c_double * 3),
                ('ang_vel', C.c_double * 3)]


class Timing(C.Structure):
    _fields_ = [('ms', C.c_double * len(MCL_K_NAMES)), ('launches', C.c_int64 * len(MCL_K_NAMES))]


# every symbol include/mcl.h, mcl_dr.h and mcl_map.h declare: name -> (restype, argtypes)
_vp, _i32, _i64, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = {
    'mcl_abi_version': (C.c_int, []),
    'mcl_status_string': (C.c_char_p, [C.c_int]),
    'mcl_last_error': (C.c_char_p, [_vp]),
    'mcl_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'mcl_matrix_from_tf': (C.c_int, [_vp, _vp, _vp]),
    'mcl_create': (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    'mcl_destroy': (C.c_int, [_vp]),
    'mcl_init_particles': (C.c_int, [_vp, _vp]),
    'mcl_predict': (C.c_int, [_vp, C.POINTER(Odom), _d, _vp]),
    'mcl_update_gps': (C.c_int, [_vp, _d, _d]),
    'mcl_set_map_grid': (C.c_int, [_vp, _vp, _i32, _i32, _d, _d, _d]),
    'mcl_set_map_mesh': (C.c_int, [_vp, _vp, _i64, _vp, _i64]),
    'mcl_set_map_mesh_ex': (C.c_int, [_vp, _vp, _i64, _vp, _i64, C.c_uint32]),
    'mcl_update_mbes': (C.c_int, [_vp, _vp, _vp, _i32, _d, _d, _vp]),
    'mcl_mbes_expected': (C.c_int, [_vp, _i64, _i64, _vp, _i32, _d, _vp, _vp]),
    'mcl_set_landmarks': (C.c_int, [_vp, _vp, _i64]),
    'mcl_set_landmark_noise': (C.c_int, [_vp, _vp, _vp]),
    'mcl_update_landmarks': (C.c_int, [_vp, _vp, _i32, _d, _i32, _d, _vp, _i32]),
    'mcl_update_landmarks_assign': (C.c_int, [_vp, _vp, _i32, _d, _i32, _d, _d, _vp, _i32, _vp, _i64]),
    'mcl_resample': (C.c_int, [_vp, _vp, _i64, _vp]),
    'mcl_resample_prepare': (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    'mcl_mean_cov': (C.c_int, [_vp, _vp, _vp, _vp]),
    'mcl_get_poses': (C.c_int, [_vp, _vp]),
    'mcl_get_particles': (C.c_int, [_vp, _vp, _vp]),
    'mcl_set_particles': (C.c_int, [_vp, _vp]),
    'mcl_get_log_weights': (C.c_int, [_vp, _vp]),
    'mcl_set_log_weights': (C.c_int, [_vp, _vp, _i32]),
    'mcl_get_last_indices': (C.c_int, [_vp, _vp]),
    'mcl_get_last_offspring_cdf': (C.c_int, [_vp, _vp]),
    'mcl_get_fixed_weights': (C.c_int, [_vp, _vp, _vp]),
    'mcl_step_mbes': (C.c_int, [_vp, C.POINTER(Odom), _d, _vp, _vp, _i32, _d, _d, _vp]),
    'mcl_step_mbes_landmarks': (C.c_int, [_vp, C.POINTER(Odom), _d, _vp, _vp, _i32, _d, _d, _vp, _vp, _i32, _d, _i32, _d, _vp]),
    'mcl_sync': (C.c_int, [_vp]),
    'mcl_last_mean_cov': (C.c_int, [_vp, _vp, _vp, _vp]),
    'mcl_mean_history': (C.c_int, [_vp, _i64, _vp]),
    'mcl_resample_indices': (C.c_int, [_i32, _vp, _i64, _vp, _i64, _i32, _vp]),
    'mcl_comm_unique_id': (C.c_int, [C.c_char_p]),
    'mcl_comm_init': (C.c_int, [_vp, C.c_char_p]),
    'mcl_comm_init_ex': (C.c_int, [_vp, C.c_char_p, C.c_uint32]),
    'mcl_comm_ranks': (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    'mcl_comm_selftest': (C.c_int, [_vp, _i32]),
    'mcl_comm_shutdown': (C.c_int, [_vp, _i32]),
    'mcl_mean_cov_async': (C.c_int, [_vp]),
    'mcl_group_resample': (C.c_int, [C.POINTER(_vp), _i32, _vp, _i64, C.POINTER(_vp)]),
    'mcl_group_mean_cov': (C.c_int, [C.POINTER(_vp), _i32, _vp, _vp, _vp]),
    'mcl_group_step_mbes': (C.c_int, [C.POINTER(_vp), _i32, C.POINTER(Odom), _d, _vp, _vp, _i32, _d, _d, _vp]),
    'mcl_group_step_mbes_landmarks': (C.c_int, [C.POINTER(_vp), _i32, C.POINTER(Odom), _d, _vp, _vp, _i32, _d, _d, _vp,
                                                _vp, _i32, _d, _i32, _d, _vp]),
    'mcl_exchange_stats': (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i32]),
    'mcl_exchange_ops': (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i32]),
    'mcl_exchange_plan': (C.c_int, [_i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'mcl_timing_enable': (C.c_int, [_vp, _i32]),
    'mcl_timing_get': (C.c_int, [_vp, C.POINTER(Timing)]),
    'mcl_mbes_last_path': (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'mcl_mbes_last_handover': (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'mcl_mbes_visit_order': (C.c_int, [_vp, _vp, C.POINTER(C.c_int32)]),
    # include/mcl_dr.h: the dead-reckoning integrator (host only)
    'mcl_dr_create': (C.c_int, [C.POINTER(DrConfig), C.POINTER(_vp)]),
    'mcl_dr_destroy': (None, [_vp]),
    'mcl_dr_heading': (C.c_int, [_vp, _vp]),
    'mcl_dr_gps': (C.c_int, [_vp, _d, _d, C.c_int, _vp, C.POINTER(C.c_int), _vp, _vp]),
    'mcl_dr_imu': (C.c_int, [_vp, _d, _vp, _vp]),
    'mcl_dr_dvl': (C.c_int, [_vp, _d, _vp]),
    'mcl_dr_depth': (C.c_int, [_vp, _d]),
    'mcl_dr_thrust_cmd': (C.c_int, [_vp, _d]),
    'mcl_dr_thrust': (C.c_int, [_vp, _d, _d]),
    'mcl_dr_tick': (C.c_int, [_vp, C.POINTER(DrOdom)]),
    'mcl_dr_to_odom': (C.c_int, [C.POINTER(DrOdom), _d, C.POINTER(Odom)]),
    # include/mcl_map.h: the bathymetry map builder
    'mcl_gridmap_create': (C.c_int, [_i32, _i32, _d, _d, _d, _i32, C.POINTER(_vp)]),
    'mcl_gridmap_destroy': (None, [_vp]),
    'mcl_gridmap_last_error': (C.c_char_p, [_vp]),
    'mcl_gridmap_clear': (C.c_int, [_vp]),
    'mcl_gridmap_add_pings': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _d, _vp, _vp, _vp]),
    'mcl_gridmap_finalize': (C.c_int, [_vp, _i32, _vp, C.POINTER(C.c_int64), _vp]),
}

_lib = None


def load():
    """Load libmcl_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError('libmcl_hip.so not built (%s): the MCL hot path has no fallback; run '
                          '__graft_entry__.build()' % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mcl_abi_version() != 4:
        raise ImportError('libmcl_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(status, handle=None):
    if status != 0:
        lib = load()
        msg = lib.mcl_last_error(handle)
        raise MclError(status, (msg or b'').decode() or lib.mcl_status_string(status).decode())
