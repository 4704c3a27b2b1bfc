#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own
auv_particle_filter modules (read-only, from /root/reference) behind ROS stand-ins.

TEST INFRASTRUCTURE.  Runs only in the development container (the reference does not exist
on the GPU box); the committed .npz files are pure data: inputs, RNG seeds and the outputs
the reference produced.  Re-run:  python oracle/ref_harness/gen_golden.py

RNG: the reference never seeds numpy's legacy global RandomState; the harness seeds it
(np.random.seed(k)) so tests regenerate the identical draw stream from the seed alone
(consumption order: SURVEY.md Appendix A.1).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/auv_particle_filter/scripts'
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import rospy  # noqa: E402  (stub)
import tf  # noqa: E402  (stub)
from nav_msgs.msg import Odometry  # noqa: E402
import resampling as ref_resampling  # noqa: E402  (reference)
import auv_particle as ref_particle  # noqa: E402  (reference)
import auv_pf as ref_pf  # noqa: E402  (reference)

from smarc_navigation_amd import synth  # noqa: E402

import manifest  # noqa: E402  (this directory: where to write, and the fixture hashes)

OUT = manifest.golden_dir()
WRITTEN = []


def make_node(n, init_cov, motion_cov, res_cov, meas_std, m2o, utm2map):
    """Build the reference node object without running its blocking __init__
    (auv_pf.py:25-119 waits on tf and spins); attribute set = what __init__ creates."""
    pf = object.__new__(ref_pf.auv_pf)
    pf.pc = n
    pf.map_frame, pf.base_frame, pf.utm_frame, pf.odom_frame = 'map', 'base_link', 'utm', 'sam/odom'
    pf.listener = tf.TransformListener()
    pf.listener.utm2map = utm2map
    pf.res_noise_cov = list(res_cov)
    pf.poses = ref_pf.PoseArray()
    pf.poses.header.frame_id = pf.odom_frame
    pf.pf_pub = rospy.Publisher('/particle_poses', ref_pf.PoseArray)
    pf.loc_pose = Odometry()
    pf.loc_pose.header.frame_id = pf.odom_frame
    pf.loc_pose.child_frame_id = pf.base_frame
    pf.loc_pub = rospy.Publisher('/average_pose', Odometry)
    pf.loc_tf = tf.TransformBroadcaster()
    pf.m2o_mat = m2o
    pf.particles = np.empty(n, dtype=object)
    for i in range(n):
        pf.particles[i] = ref_particle.Particle(n, i, m2o, init_cov=list(init_cov), meas_std=meas_std,
                                                process_cov=list(motion_cov))
    pf.time = 0.0
    pf.old_time = 0.0
    pf.diving = True
    return pf


def odom_msg(stream, k):
    m = Odometry()
    m.header.stamp = rospy.Time(stream['stamp'][k])
    m.twist.twist.linear.x, m.twist.twist.linear.y, m.twist.twist.linear.z = stream['v'][k]
    m.twist.twist.angular.z = stream['wz'][k]
    (m.pose.pose.orientation.x, m.pose.pose.orientation.y,
     m.pose.pose.orientation.z, m.pose.pose.orientation.w) = stream['q'][k]
    m.pose.pose.position.z = stream['z'][k]
    return m


def gps_msg(x, y):
    m = Odometry()
    m.pose.pose.position.x, m.pose.pose.position.y = x, y
    return m


def poses_of(pf):
    return np.array([np.asarray(p.p_pose, dtype=np.float64) for p in pf.particles])


def run_scenario(name, n, n_steps, seed, init_cov, motion_cov, res_cov, meas_std, gps_every,
                 resampler, pub_every=5):
    """Drive the reference node with a synthetic stream.  GPS gating is specified away as in
    SURVEY A.10: an update+resample runs on every fix (diving == False)."""
    stream = synth.odom_stream(n_steps)
    m2o = synth.rigid_matrix(12.5, -7.25, 0.0, 0.0, 0.0, 0.3)
    utm2map = synth.rigid_matrix(-1000.0, -2000.0, 0.0, 0.0, 0.0, 0.0)
    map2utm = np.linalg.inv(utm2map)
    fix_idx, fix_xy = synth.gps_fixes(stream['truth'], m2o, every=gps_every if gps_every else n_steps + 1)
    if resampler == 'systematic':
        ref_pf.residual_resample = ref_resampling.systematic_resample
    else:
        ref_pf.residual_resample = ref_resampling.residual_resample

    # capture what resample() computed: indices + uniforms are observable through the resampler
    cap = {'indices': [], 'weights_raw': [], 'weights_norm': []}
    orig = ref_pf.residual_resample

    def spy(weights):
        cap['weights_norm'].append(np.array(weights, copy=True))
        idx = orig(weights)
        cap['indices'].append(np.array(idx, dtype=np.int32))
        return idx
    ref_pf.residual_resample = spy

    np.random.seed(seed)
    pf = make_node(n, init_cov, motion_cov, res_cov, meas_std, m2o, utm2map)
    pf.old_time = stream['t0']
    pf.time = stream['t0']
    init_state = poses_of(pf)
    ckpt_steps, ckpt_states = [], []
    pub_steps, means, covs, quats, tf_trans, posearray0 = [], [], [], [], [], []
    post_update_states = []
    fix_ptr = 0
    for k in range(n_steps):
        pf.odom_callback(odom_msg(stream, k))
        if fix_ptr < fix_idx.size and fix_idx[fix_ptr] == k:
            gx, gy = fix_xy[fix_ptr]
            utm = map2utm.dot(np.array([gx, gy, 0.0, 1.0]))
            pf.diving = False
            # the reference's gate (auv_pf.py:126) is only true mid-odom-callback; open it
            pf.time = pf.old_time + 1.0
            w = pf.update(gps_msg(utm[0], utm[1]))
            cap['weights_raw'].append(np.array(w, copy=True))
            pf.resample(w)
            pf.time = pf.old_time
            post_update_states.append(poses_of(pf))
            fix_ptr += 1
        if (k + 1) % pub_every == 0 or k == n_steps - 1:
            pf.loc_loop(None)
            lp = pf.loc_pose
            pub_steps.append(k)
            means.append([lp.pose.pose.position.x, lp.pose.pose.position.y, lp.pose.pose.position.z])
            o = lp.pose.pose.orientation
            quats.append([o.x, o.y, o.z, o.w])
            covs.append(list(lp.pose.covariance))
            tf_trans.append(pf.loc_tf.sent[-1][0])
        if (k + 1) % 25 == 0 or k == n_steps - 1:
            ckpt_steps.append(k)
            ckpt_states.append(poses_of(pf))
    # last PoseArray (positions + quaternions of every particle)
    pa = pf.poses.poses
    posearray = np.array([[p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y,
                           p.orientation.z, p.orientation.w] for p in pa])
    out = dict(
        n=n, n_steps=n_steps, seed=seed, init_cov=np.array(init_cov), motion_cov=np.array(motion_cov),
        res_cov=np.array(res_cov), meas_std=meas_std, m2o=m2o, resampler=resampler, pub_every=pub_every,
        gps_every=gps_every, fix_idx=fix_idx[:fix_ptr], fix_xy_map=fix_xy[:fix_ptr],
        init_state=init_state, ckpt_steps=np.array(ckpt_steps), ckpt_states=np.array(ckpt_states),
        pub_steps=np.array(pub_steps), mean_xyz=np.array(means), quat=np.array(quats),
        cov36=np.array(covs), tf_trans=np.array(tf_trans), posearray_last=posearray,
        post_update_states=np.array(post_update_states),
        weights_raw=np.array(cap['weights_raw']), weights_norm=np.array(cap['weights_norm']),
        indices=np.array(cap['indices']))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    WRITTEN.append(name + '.npz')
    ref_pf.residual_resample = ref_resampling.residual_resample
    print('%-28s n=%d steps=%d fixes=%d final mean=(%.4f, %.4f)' % (name, n, n_steps, fix_ptr,
                                                                   means[-1][0], means[-1][1]))


def weight_shapes(n, rs):
    """A few weight vectors per size: flat, random, peaked (GPS-like), degenerate with zeros."""
    out = {}
    out['flat'] = np.full(n, 1.0 / n)
    w = rs.rand(n)
    out['random'] = w / w.sum()
    d = rs.randn(n) * 3.0
    w = np.exp(-0.5 * d * d) + 1e-200
    out['peaked'] = w / w.sum()
    w = rs.rand(n)
    w[rs.rand(n) < 0.5] = 0.0
    if w.sum() == 0.0:
        w[0] = 1.0
    out['zeros'] = w / w.sum()
    w = np.full(n, 1e-200)
    w[n // 3] = 1.0
    out['single'] = w / w.sum()
    return out


def gen_resampling():
    """Known-answer vectors for every resampler in the reference's resampling.py."""
    rs = np.random.RandomState(1234)
    out = {}
    cases = []
    for n in (1, 2, 7, 64, 128, 1000, 4096, 65536):
        for shape, w in weight_shapes(n, rs).items():
            if n > 4096 and shape not in ('random', 'peaked'):
                continue  # keep the fixture small
            tag = 'n%d_%s' % (n, shape)
            seed = int(rs.randint(1, 2 ** 31 - 1))
            out[tag + '_w'] = w
            out[tag + '_seed'] = seed
            for fn in ('systematic_resample', 'stratified_resample', 'multinomial_resample',
                       'residual_resample', 'naive_resample'):
                if fn == 'residual_resample' and shape in ('single',) and n > 1:
                    # k = sum(floor(N w)) == 1 -> sum(residual) ~ 0 -> inf/nan in the reference (A.6)
                    continue
                if fn in ('residual_resample', 'naive_resample') and n > 4096:
                    continue
                np.random.seed(seed)
                # resampling.py binds `random` at import: reseeding the global state is enough
                with np.errstate(all='ignore'):
                    try:
                        idx = np.asarray(getattr(ref_resampling, fn)(w.copy()), dtype=np.int64)
                    except IndexError:
                        continue
                out[tag + '_' + fn] = idx.astype(np.int32)
            cases.append(tag)
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'resampling_kat.npz'), **out)
    WRITTEN.append('resampling_kat.npz')
    print('resampling_kat: %d cases' % len(cases))


def gen_particle_kat():
    """Per-function known answers: motion_pred, compute_weight, fullRotation, add_noise,
    matrix_from_tf, euler helpers (auv_particle.py)."""
    rs = np.random.RandomState(77)
    m2o = synth.rigid_matrix(3.0, -4.0, 0.5, 0.01, -0.02, 1.1)
    n = 64
    out = {'m2o': m2o}
    # motion_pred
    pose0 = rs.randn(n, 6) * np.array([50, 50, 5, 0.2, 0.2, 3.0])
    v = rs.randn(n, 3) * np.array([1.5, 0.3, 0.2])
    wz = rs.randn(n) * 0.2
    rpy = rs.randn(n, 3) * np.array([0.3, 0.3, 3.0])
    q = synth.quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    zz = rs.randn(n) * 3 - 5
    dts = np.abs(rs.randn(n)) * 0.05 + 0.001
    pcov = np.abs(rs.randn(n, 6)) * np.array([1e-2, 1e-2, 1e-3, 1e-4, 1e-4, 1e-3])
    seeds = rs.randint(1, 2 ** 31 - 1, size=n)
    pose1 = np.zeros((n, 6))
    for i in range(n):
        np.random.seed(int(seeds[i]))
        p = ref_particle.Particle(1, 0, m2o, process_cov=list(pcov[i]))
        p.p_pose = pose0[i].copy()
        stream = dict(stamp=[0.0], v=[v[i]], wz=[wz[i]], q=[q[i]], z=[zz[i]])
        np.random.seed(int(seeds[i]))
        p.motion_pred(odom_msg(stream, 0), float(dts[i]))
        pose1[i] = p.p_pose
    out.update(mp_pose0=pose0, mp_v=v, mp_wz=wz, mp_q=q, mp_z=zz, mp_dt=dts, mp_pcov=pcov,
               mp_seeds=seeds, mp_pose1=pose1)
    # compute_weight
    gps = rs.randn(n, 2) * 30
    stds = np.abs(rs.randn(n)) * 2 + 0.05
    ww = np.zeros(n)
    pmap = np.zeros((n, 3))
    from geometry_msgs.msg import PointStamped
    for i in range(n):
        p = ref_particle.Particle(1, 0, m2o, meas_std=float(stds[i]))
        p.p_pose = pose0[i].copy()
        g = PointStamped()
        # keep some fixes near the particle so weights are not all underflowed
        pm = m2o.dot(np.array([pose0[i, 0], pose0[i, 1], pose0[i, 2], 1.0]))
        if i % 2 == 0:
            gps[i] = pm[:2] + rs.randn(2) * stds[i] * 2
        g.point.x, g.point.y = gps[i]
        p.compute_weight(g)
        ww[i] = p.w
        pmap[i] = p.p
    out.update(cw_gps=gps, cw_std=stds, cw_w=ww, cw_pmap=pmap)
    # fullRotation rows 0-1 (row 2 is malformed in the reference and unused)
    rots = np.zeros((n, 3, 3))
    p = ref_particle.Particle(1, 0, m2o)
    for i in range(n):
        rots[i] = p.fullRotation(rpy[i, 0], rpy[i, 1], rpy[i, 2])
    out.update(fr_rpy=rpy, fr_R=rots)
    # add_noise
    cov = np.array([0.1, 0.2, 0.0, 1e-3, 0.0, 1e-4])
    np.random.seed(5)
    p = ref_particle.Particle(1, 0, m2o)
    p.p_pose = pose0[0].copy()
    np.random.seed(6)
    p.add_noise(list(cov))
    out.update(an_cov=cov, an_seed=6, an_pose0=pose0[0], an_pose1=np.asarray(p.p_pose))
    # euler_from_quaternion as used by motion_pred (stub == restated published algorithm)
    from tf.transformations import euler_from_quaternion, quaternion_from_euler
    out['eq_q'] = q
    out['eq_rpy'] = np.array([euler_from_quaternion(q[i]) for i in range(n)])
    out['qe_q'] = np.array([quaternion_from_euler(*rpy[i]) for i in range(n)])
    # matrix_from_tf

    class _T(object):
        _type = 'geometry_msgs/Transform'
    t = _T()
    t.translation = types.SimpleNamespace(x=1.5, y=-2.5, z=0.25)
    t.rotation = types.SimpleNamespace(x=q[3, 0], y=q[3, 1], z=q[3, 2], w=q[3, 3])
    out['mt_in'] = np.array([1.5, -2.5, 0.25, q[3, 0], q[3, 1], q[3, 2], q[3, 3]])
    out['mt_M'] = ref_particle.matrix_from_tf(t)
    np.savez_compressed(os.path.join(OUT, 'particle_kat.npz'), **out)
    WRITTEN.append('particle_kat.npz')
    print('particle_kat written')


def main():
    os.makedirs(OUT, exist_ok=True)
    launch_init = [0.1, 0.1, 0.0, 0.0, 0.0, 0.0]
    launch_motion = [0.0, 0.0, 0.0, 0.0, 0.0, 1e-12]
    launch_res = [1.0, 1.0, 0.0, 0.0, 0.0, 1e-4]
    motion2 = [1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6]
    full_cov = [1e-4, 2e-4, 1e-5, 1e-6, 2e-6, 1e-6]
    # config 1: predict only, launch defaults and the second motion covariance
    run_scenario('traj_predict_launch', 128, 1000, 0, launch_init, launch_motion, launch_res, 1.0, 0, 'residual')
    run_scenario('traj_predict_motion2', 128, 1000, 1, launch_init, motion2, launch_res, 1.0, 0, 'residual')
    # node as written (residual resample, auv_pf.py:182)
    run_scenario('traj_gps_residual', 128, 1000, 2, launch_init, motion2, launch_res, 1.0, 50, 'residual')
    # same node with the reference's own systematic_resample (resampling.py:135) swapped in
    run_scenario('traj_gps_systematic', 128, 1000, 3, launch_init, motion2, launch_res, 1.0, 50, 'systematic')
    run_scenario('traj_gps_systematic_n1000', 1000, 300, 4, launch_init, motion2,
                 [0.5, 0.5, 1e-3, 1e-5, 1e-5, 1e-4], 2.0, 25, 'systematic')
    run_scenario('traj_gps_systematic_fullcov', 7, 200, 5, [0.1, 0.2, 0.05, 0.01, 0.02, 0.03], full_cov,
                 [0.5, 0.5, 1e-3, 1e-5, 1e-5, 1e-4], 0.7, 20, 'systematic')
    gen_resampling()
    gen_particle_kat()
    manifest.record(OUT, WRITTEN, 'oracle/ref_harness/gen_golden.py', needs_reference=True)


if __name__ == '__main__':
    main()
