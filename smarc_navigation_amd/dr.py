"""Dead-reckoning integrator: the host-side producer of the particle filter's Odometry input
(include/mcl_dr.h; replaces sam_dead_reckoning/scripts/dr_node.py's VehicleDR callbacks).

`VehicleDR` mirrors the reference node's callback names (stim_cb, sbg_cb, gps_cb, dvl_cb, depth_cb,
thrust_cb, thrust_cmd_cb, dr_timer) over plain values instead of ROS messages; `replay_events`
feeds it a time-ordered event table (synth.raw_sensor_events format) and returns what every timer
tick published."""
import ctypes as C

import numpy as np

from . import _lib, synth


def _v(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size != n:
        raise ValueError('expected %d values' % n)
    return a


class VehicleDR(object):
    def __init__(self, dvl_period=0.2, dr_period=0.02):
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.DrConfig(dvl_period, dr_period)
        _lib.check(self._L.mcl_dr_create(C.byref(cfg), C.byref(self._h)))
        self.m2o = None  # (translation[3], quaternion[4]) once gps_cb fixed the map -> odom transform

    def close(self):
        if self._h:
            self._L.mcl_dr_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def sbg_cb(self, q):
        q = _v(q, 4)
        _lib.check(self._L.mcl_dr_heading(self._h, q.ctypes.data))

    def gps_cb(self, gx_map, gy_map, pressure_tf=None):
        """Fix already in the map frame; pressure_tf = translation base_link -> pressure_link or None."""
        t, q = np.zeros(3), np.zeros(4)
        flag = C.c_int(0)
        b2p = _v(pressure_tf, 3) if pressure_tf is not None else None
        _lib.check(self._L.mcl_dr_gps(self._h, float(gx_map), float(gy_map), 1 if b2p is not None else 0,
                                      b2p.ctypes.data if b2p is not None else None, C.byref(flag),
                                      t.ctypes.data, q.ctypes.data))
        if flag.value:
            self.m2o = (t, q)
        return bool(flag.value)

    def stim_cb(self, stamp, q, ang_vel):
        q, w = _v(q, 4), _v(ang_vel, 3)
        _lib.check(self._L.mcl_dr_imu(self._h, float(stamp), q.ctypes.data, w.ctypes.data))

    def dvl_cb(self, stamp, vel):
        v = _v(vel, 3)
        _lib.check(self._L.mcl_dr_dvl(self._h, float(stamp), v.ctypes.data))

    def depth_cb(self, z):
        _lib.check(self._L.mcl_dr_depth(self._h, float(z)))

    def thrust_cmd_cb(self, thruster_horizontal_radians):
        _lib.check(self._L.mcl_dr_thrust_cmd(self._h, float(thruster_horizontal_radians)))

    def thrust_cb(self, rpm1, rpm2):
        _lib.check(self._L.mcl_dr_thrust(self._h, float(rpm1), float(rpm2)))

    def dr_timer(self):
        """One timer tick; returns the DrOdom it published (published == 0: nothing yet)."""
        o = _lib.DrOdom()
        _lib.check(self._L.mcl_dr_tick(self._h, C.byref(o)))
        return o

    def to_odom(self, dr_odom, stamp):
        """The mcl_odom (fields auv_pf.odom_callback consumes) of a published tick."""
        od = _lib.Odom()
        _lib.check(self._L.mcl_dr_to_odom(C.byref(dr_odom), float(stamp), C.byref(od)))
        return od


def replay_events(t, kind, data, gps_map=None, pressure_tf=None, dvl_period=0.2, dr_period=0.02):
    """Drive a VehicleDR with an event table.  gps_map[k] = the k-th fix in the map frame (default:
    the event's own x, y).  Returns (ticks[n_ticks, 15], m2o[7]) with tick columns
    published, x, y, z, qx, qy, qz, qw, vx, vy, vz, wx, wy, wz, t_now (NaN where not published)."""
    dr = VehicleDR(dvl_period, dr_period)
    ticks = []
    m2o = np.full(7, np.nan)
    ig = 0
    for ti, ki, d in zip(t, kind, data):
        if ki == synth.EV_IMU:
            dr.stim_cb(ti, d[0:4], d[4:7])
        elif ki == synth.EV_HEADING:
            dr.sbg_cb(d[0:4])
        elif ki == synth.EV_GPS:
            g = gps_map[ig] if gps_map is not None else d[0:2]
            ig += 1
            if dr.gps_cb(g[0], g[1], pressure_tf):
                m2o[:] = np.concatenate(dr.m2o)
        elif ki == synth.EV_DVL:
            dr.dvl_cb(ti, d[0:3])
        elif ki == synth.EV_DEPTH:
            dr.depth_cb(d[0])
        elif ki == synth.EV_THRUST:
            dr.thrust_cb(d[0], d[1])
        elif ki == synth.EV_THRUST_CMD:
            dr.thrust_cmd_cb(d[0])
        elif ki == synth.EV_TICK:
            o = dr.dr_timer()
            row = np.full(15, np.nan)
            row[0] = 0.0
            if o.published:
                row[:] = [1.0] + list(o.pos) + list(o.q) + list(o.lin_vel) + list(o.ang_vel) + [o.t_now]
            ticks.append(row)
    dr.close()
    return np.array(ticks), m2o
