"""CPU restatement of the reference dead-reckoning node -- TEST INFRASTRUCTURE ONLY (never imported
by the product path).  Follows sam_dead_reckoning/scripts/dr_node.py (VehicleDR) and sam_mm.py (SAM)
callback by callback with numpy in the reference's own operation order; pinned to the golden fixtures
tests/golden/dr_*.npz that the reference itself produced (oracle/ref_harness/gen_golden_dr.py)."""
import math

import numpy as np

EV_IMU, EV_HEADING, EV_GPS, EV_DVL, EV_DEPTH, EV_THRUST, EV_THRUST_CMD, EV_TICK = range(8)


def _quat_matrix3(q):  # tf.transformations.quaternion_matrix
    q = np.array(q, dtype=np.float64)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        return np.identity(3)
    q = q * math.sqrt(2.0 / nq)
    o = np.outer(q, q)
    return np.array([[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                     [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                     [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]])


def euler_from_quaternion(q):  # static xyz (dr_node.py:10)
    M = _quat_matrix3(q)
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > np.finfo(float).eps * 4.0:
        return math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
    return math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0


def quaternion_from_euler(ai, aj, ak):
    ci, si = math.cos(ai / 2.0), math.sin(ai / 2.0)
    cj, sj = math.cos(aj / 2.0), math.sin(aj / 2.0)
    ck, sk = math.cos(ak / 2.0), math.sin(ak / 2.0)
    return np.array([cj * (si * ck) - sj * (ci * sk), cj * (si * sk) + sj * (ci * ck),
                     cj * (ci * sk) - sj * (si * ck), cj * (ci * ck) + sj * (si * sk)])


def sam_motion(control):  # sam_mm.py:30-120
    rpm, dr = control
    m, Izz, x_g, y_g, KT = 15.4, 1.6202, 0.4, 0.0, 0.3
    dr = dr * -1.0
    M = np.array([[m, 0.0, -m * y_g], [0, m, m * x_g], [-m * y_g, m * x_g, Izz]])
    F_T = KT * rpm
    tauc = np.array([F_T * np.cos(dr), -F_T * np.sin(dr), 0.0])
    return np.linalg.inv(M).dot(tauc)


def full_rotation(roll, pitch, yaw):  # dr_node.py:257-270 (third row of rot_y as written there)
    rz = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
    ry = np.array([[np.cos(pitch), 0.0, np.sin(pitch)], [0.0, 1.0, 0.0], [-np.sin(pitch), np.cos(pitch), 0.0]])
    rx = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(roll), -np.sin(roll)], [0.0, np.sin(roll), np.cos(roll)]])
    return np.matmul(rz, np.matmul(ry, rx))


def replay_events(t, kind, data, gps_map=None, pressure_tf=None, dvl_period=0.2, dr_period=0.02):
    init_heading = init_m2o = init_stim = dvl_on = depth_meas = False
    gps_registered = True
    init_quat = None
    pos_t, rot_t, vel_rot = np.zeros(3), np.zeros(3), np.zeros(3)
    t_stim_prev = t_dvl_prev = t_now = 0.0
    dvl = np.zeros(3)
    b2p = np.zeros(3)
    base_depth = 0.0
    u = [0.0, 0.0]
    thrust_cmd = 0.0
    lim = 7 * np.pi / 180
    ticks, m2o, ig = [], np.full(7, np.nan), 0
    for ti, ki, d in zip(t, kind, data):
        if ki == EV_IMU:        # stim_cb :273-302
            if init_stim and init_m2o:
                e = euler_from_quaternion(d[0:4])
                vel_rot = np.array(d[4:7])
                rot_t = rot_t + vel_rot * (ti - t_stim_prev)
                t_stim_prev = ti
                rot_t[0], rot_t[1] = e[0], e[1]
            else:
                t_stim_prev = ti
                init_stim = True
        elif ki == EV_HEADING:  # sbg_cb
            init_quat = d[0:4].copy()
            init_heading = True
        elif ki == EV_GPS:      # gps_cb :108-161
            g = gps_map[ig] if gps_map is not None else d[0:2]
            ig += 1
            if gps_registered:
                if init_heading:
                    e = euler_from_quaternion(init_quat)
                    m2o[:] = np.concatenate([[g[0], g[1], 0.0], quaternion_from_euler(0.0, 0.0, e[2])])
                    init_m2o = True
                    gps_registered = False
                if pressure_tf is not None:
                    b2p = np.array(pressure_tf, dtype=np.float64)
                    depth_meas = True
        elif ki == EV_DVL:      # dvl_cb :305-336
            dvl = d[0:3].copy()
            if not dvl_on:
                t_now = ti
                dvl_on = True
            t_dvl_prev = ti
        elif ki == EV_DEPTH:
            if depth_meas:
                base_depth = d[0] + b2p[0] * np.sin(rot_t[1])
        elif ki == EV_THRUST:
            u = [d[0] + d[1], float(np.clip(-thrust_cmd, -lim, lim))]
        elif ki == EV_THRUST_CMD:
            thrust_cmd = d[0]
        elif ki == EV_TICK:     # dr_timer :165-236
            row = np.full(15, np.nan)
            row[0] = 0.0
            if init_m2o and init_stim:
                pose = np.concatenate([pos_t, rot_t])
                lin = np.zeros(3)
                if dvl_on:
                    R = full_rotation(pose[3], pose[4], pose[5])
                    if t_now - t_dvl_prev < dvl_period and abs(dvl[1]) < 0.2 and abs(dvl[0]) < 1.5 and dvl[0] > -0.1:
                        lin = dvl.copy()
                    else:
                        acc = sam_motion(u)[0:3]
                        lin = np.array([acc[0], -acc[1], 0.0]) * dr_period
                    step = np.matmul(R, lin * dr_period)
                    pose[0:2] += step[0:2]
                pose[2] = base_depth
                q = quaternion_from_euler(pose[3], pose[4], pose[5])
                t_now += dr_period
                pos_t = pose[0:3].copy()
                row[:] = np.concatenate([[1.0], pose[0:3], q, lin, vel_rot, [t_now]])
            ticks.append(row)
    return np.array(ticks), m2o
