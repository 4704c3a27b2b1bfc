"""Host-side mirror of the reference node class `auv_pf` (auv_particle_filter/scripts/auv_pf.py):
same class name, method names, parameter names/defaults and message fields -- the per-particle
Python loops are replaced by calls through the C ABI (libmcl_hip.so).  ROS-free core: transport
(publishers, tf) is injected; `ros_node.py` plugs in rospy when ROS is installed.

Divergences from the reference, all documented in INTEGRATION.md:
  * one owner lock around the handle (the reference's three rospy threads are unlocked);
  * the GPS gate `self.time > self.old_time` (auv_pf.py:126, only true mid-callback) is "every fix
    while not diving";
  * default resampling scheme on the GPU is systematic (resampling.py:135); pass
    resample_scheme='residual' for the node's literal behaviour (auv_pf.py:182).
"""
import math
import threading

import numpy as np

from . import engine as _engine
from . import msgs as _msgs

_SCHEMES = {'systematic': _engine.SYSTEMATIC, 'residual': _engine.RESIDUAL,
            'stratified': _engine.STRATIFIED, 'multinomial': _engine.MULTINOMIAL}

DEFAULT_PARAMS = {
    # auv_pf.py:27-31
    'particle_count': 10, 'map_frame': 'map', 'base_frame': 'base_link', 'utm_frame': 'utm',
    'odom_frame': 'sam/odom',
    # auv_pf.py:39 (+ launch defaults auv_pf.launch:17-20 for the three strings, which have no code default)
    'measurement_std': 0.01,
    'motion_covariance': '[0.0000, 0.0000, 0.0, 0.0, 0.0, 0.000000000001]',
    'init_covariance': '[0.1, 0.1, 0.0, 0.0, 0.0, 0.0]',
    'resampling_noise_covariance': '[1., 1., 0.0, 0.0, 0.0, 0.0001]',
    # topics (auv_pf.py:64,71,101,106,110)
    'particle_poses_topic': '/particle_poses', 'odom_corrected_topic': '/average_pose',
    'aux_dive': '/dive', 'gps_odom_topic': '/gps', 'odom_topic': 'odom',
    # new, behaviour-preserving defaults
    'resample_scheme': 'systematic', 'seed': 0, 'device': 0,
    'mbes_topic': '/mbes_scan', 'mbes_std': 0.2, 'mbes_sensor_offset': '[0.0, 0.0, 0.0, 0.0, 0.0, 0.0]',
    # the bathymetric map the MBES update casts against: an .npz with `z` (nx, ny) + `origin` (2) + `res` (height
    # grid, map frame) or with `verts` (nv, 3) + `tris` (nt, 3); an ASCII .ply triangle mesh; a .mclgrid height grid
    # (save_mclgrid: the format the roscpp node reads too).  '' = no map: pings ignored
    'map_grid_file': '', 'map_mesh_file': '',
    # MBES pings as sensor_msgs/PointCloud2 (what mbes_mapper's receptor makes of a LaserScan,
    # mbes_receptor.cpp:126-165): '' = not subscribed.  `mbes_points_frame`: 'base' -- points in base_frame, as
    # transformLaserScanToPointCloud(base_frame_, ...) leaves them -- or 'sensor'
    'mbes_pointcloud_topic': '', 'mbes_points_frame': 'base', 'mbes_range_max': 100.0,
    # BASELINE config 5: landmark detections with per-particle k-NN data association.  `landmark_map_file`: the
    # feature map -- the .yaml the reference's map provider serves (auv_ekf_localization/scripts/map_provider_node.py:
    # 35-56: a list of models with position {x, y, z}, those below `rocks_depth` kept), an .npz with `landmarks`
    # (n, 3), or a text file of x y z rows; map frame.  '' = no landmark map: detections ignored.
    # `lm_detect_topic`: geometry_msgs/PoseArray of detections in base_frame, positions only, as the MBES receptors
    # publish them (mbes_toy_processor/src/toy_mbes_receptor.cpp:68-110; consumer auv_ekf_slam/src/ekf_slam.cpp:41).
    # The receptor stamps a detection message with its ping's stamp and publishes it AFTER processing the ping, so in a
    # live system it reaches the node when that ping's update and resampling are done: it is then a measurement update
    # of its own, followed by the resampling (detections and ranges are independent measurements of the same pose).  A
    # stream that delivers the detections first (a bag replayed by topic) has them wait for the ping with THEIR stamp
    # (equal within `landmark_sync_tol` seconds: the stamps are copies of one another), whose likelihood they join
    # before the resampling.  Without a bathymetric map a detection message is always an update of its own.  Detections
    # older than `landmark_max_age` seconds of the filter's clock (the latest odometry stamp) are dropped: their
    # base_frame positions describe a pose the cloud has long left.
    # NOTE (live order): a detection message that follows its ping is its OWN update + resampling, so with a receptor
    # running the filter resamples -- and adds `resampling_noise_covariance` -- TWICE per ping (once for the ranges, once
    # for the detections); choose the resampling noise for that rate.  `landmark_late` = 'drop' ignores detections that
    # arrive after their ping instead (one resampling per ping; detections then only count when they come ahead of
    # their ping and join its likelihood).
    'landmark_map_file': '', 'rocks_depth': float('inf'), 'lm_detect_topic': '/landmarks_detected',
    'landmark_std': 0.3, 'landmark_k': 1, 'landmark_gate': 11.345, 'landmark_sync_tol': 1e-3, 'landmark_max_age': 0.5,
    'landmark_late': 'update',
}


def parse_cov_string(cov_string):
    """The reference's ad-hoc parser (auv_pf.py:40-44): strip brackets, split on ', '."""
    cov_string = cov_string.replace('[', '')
    cov_string = cov_string.replace(']', '')
    cov_list = list(cov_string.split(", "))
    return list(map(float, cov_list))


def quaternion_from_euler(roll, pitch, yaw):
    """tf.transformations.quaternion_from_euler, axes 'sxyz' (auv_pf.py:233)."""
    cr, sr = math.cos(roll / 2.0), math.sin(roll / 2.0)
    cp, sp = math.cos(pitch / 2.0), math.sin(pitch / 2.0)
    cy, sy = math.cos(yaw / 2.0), math.sin(yaw / 2.0)
    return [cp * (sr * cy) - sp * (cr * sy), cp * (sr * sy) + sp * (cr * cy),
            cp * (cr * sy) - sp * (sr * cy), cp * (cr * cy) + sp * (sr * sy)]


def load_map_file(path):
    """('grid', z, origin, res) or ('mesh', verts, tris) from an .npz (keys above), an ASCII .ply triangle mesh or a
    .mclgrid height grid."""
    if path.lower().endswith('.npz'):
        with np.load(path, allow_pickle=False) as f:
            if 'z' in f.files:
                return ('grid', np.asarray(f['z'], np.float32), tuple(float(v) for v in f['origin']), float(f['res']))
            return ('mesh', np.asarray(f['verts'], np.float32), np.asarray(f['tris'], np.uint32))
    if path.lower().endswith('.ply'):
        with open(path, 'r') as f:
            if f.readline().strip() != 'ply':
                raise ValueError('%s: not a PLY file' % path)
            nv = nf = 0
            for line in f:
                t = line.split()
                if t[:2] == ['format', 'ascii']:
                    continue
                if t and t[0] == 'format':
                    raise ValueError('%s: only ASCII PLY is read' % path)
                if t[:2] == ['element', 'vertex']:
                    nv = int(t[2])
                if t[:2] == ['element', 'face']:
                    nf = int(t[2])
                if t == ['end_header']:
                    break
            verts = np.array([[float(v) for v in f.readline().split()[:3]] for _ in range(nv)], np.float32)
            tris = []
            for _ in range(nf):
                t = [int(v) for v in f.readline().split()]
                for k in range(2, t[0]):   # fan-triangulate polygons
                    tris.append((t[1], t[k], t[k + 1]))
        return ('mesh', verts, np.array(tris, np.uint32))
    if path.lower().endswith('.mclgrid'):   # text header + raw float32 (what the roscpp node reads: pf_core.hpp)
        with open(path, 'rb') as f:
            head = f.readline().split()
            if len(head) != 6 or head[0] != b'mclgrid':
                raise ValueError('%s: bad mclgrid header' % path)
            nx, ny = int(head[1]), int(head[2])
            z = np.frombuffer(f.read(4 * nx * ny), dtype='<f4')
            if z.size != nx * ny:
                raise ValueError('%s: truncated height array' % path)
        return ('grid', z.reshape(nx, ny).astype(np.float32), (float(head[3]), float(head[4])), float(head[5]))
    raise ValueError('map file %s: expected .npz, .ply or .mclgrid' % path)


def load_landmark_file(path, rocks_depth=float('inf')):
    """(n, 3) float64 landmark positions in the map frame.  .yaml / .yml: the Gazebo model list of the reference's map
    provider (map_provider_node.py:43-52: the first top-level entry is a list of models with position {x, y, z};
    models with z < rocks_depth are kept -- the provider's own filter; its default keeps the ones below -90 m, the
    default here keeps all); .npz: key `landmarks`; anything else: whitespace-separated x y z rows."""
    low = path.lower()
    if low.endswith('.npz'):
        with np.load(path, allow_pickle=False) as f:
            pts = np.asarray(f['landmarks'], np.float64).reshape(-1, 3)
    elif low.endswith('.yaml') or low.endswith('.yml'):
        import yaml
        with open(path, 'r') as f:
            data = yaml.safe_load(f)
        models = list(data.values())[0] if isinstance(data, dict) else data
        pts = np.array([[float(m['position']['x']), float(m['position']['y']), float(m['position']['z'])]
                        for m in models], np.float64).reshape(-1, 3)
    else:
        pts = np.loadtxt(path, dtype=np.float64, ndmin=2)[:, :3]
    keep = pts[:, 2] < float(rocks_depth)
    return np.ascontiguousarray(pts[keep])


def save_mclgrid(path, z, origin, res):
    """Height grid as `mclgrid nx ny origin_x origin_y res\\n` + nx * ny little-endian float32, z[ix * ny + iy]."""
    z = np.ascontiguousarray(z, dtype='<f4')
    with open(path, 'wb') as f:
        f.write(('mclgrid %d %d %r %r %r\n' % (z.shape[0], z.shape[1], float(origin[0]), float(origin[1]), float(res))).encode())
        f.write(z.tobytes())


def matrix_from_tf(translation, rotation):
    """auv_particle.py:110-125: 4x4 from a tf translation (x,y,z) and quaternion (x,y,z,w)."""
    q = np.array(rotation[:4], dtype=np.float64)
    nq = float(np.dot(q, q))
    m = np.identity(4)
    if nq >= np.finfo(float).eps * 4.0:
        q *= math.sqrt(2.0 / nq)
        o = np.outer(q, q)
        m[:3, :3] = [[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                     [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                     [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]]
    m[:3, 3] = translation[:3]
    return m


def _rigid(tx, ty, tz, roll, pitch, yaw):
    """T(t) R(static-xyz rpy): the pose of the sensor in base_link (`mbes_sensor_offset`)."""
    cr, sr, cp, sp, cy, sy = (math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw),
                              math.sin(yaw))
    m = np.identity(4)
    m[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    m[:3, 3] = (tx, ty, tz)
    return m


class RecordingTransport(object):
    """Default transport: keeps what the node would have published / broadcast."""

    def __init__(self, utm2map=None):
        self.utm2map = np.identity(4) if utm2map is None else np.asarray(utm2map, dtype=np.float64)
        self.particle_poses, self.odom_corrected, self.tf = [], [], []

    def transformPoint(self, frame, pt):  # tf.TransformListener.transformPoint (auv_pf.py:151)
        v = self.utm2map.dot(np.array([pt.point.x, pt.point.y, pt.point.z, 1.0]))
        out = _msgs.PointStamped()
        out.header.frame_id = frame
        out.point.x, out.point.y, out.point.z = float(v[0]), float(v[1]), float(v[2])
        return out

    def publish_poses(self, msg):
        self.particle_poses.append(msg)

    def publish_odom(self, msg):
        self.odom_corrected.append(msg)

    def sendTransform(self, trans, rot, stamp, child, parent):
        self.tf.append((list(trans), list(rot), stamp, child, parent))

    def now(self):
        return _msgs.Time(0.0)


class auv_pf(object):

    def __init__(self, params=None, m2o_mat=None, transport=None, rng_mode=_engine.RNG_NATIVE):
        p = dict(DEFAULT_PARAMS)
        p.update(params or {})
        self.params = p
        self.pc = int(p['particle_count'])
        self.map_frame, self.base_frame = p['map_frame'], p['base_frame']
        self.utm_frame, self.odom_frame = p['utm_frame'], p['odom_frame']
        meas_std = float(p['measurement_std'])
        motion_cov = parse_cov_string(p['motion_covariance'])
        init_cov = parse_cov_string(p['init_covariance'])
        self.res_noise_cov = parse_cov_string(p['resampling_noise_covariance'])
        self.mbes_std = float(p['mbes_std'])
        self.mbes_sensor_offset = parse_cov_string(p['mbes_sensor_offset'])
        self.transport = transport if transport is not None else RecordingTransport()
        self.m2o_mat = np.identity(4) if m2o_mat is None else np.asarray(m2o_mat, dtype=np.float64)
        self.lock = threading.Lock()
        # the particle list of auv_pf.py:89-94 lives in HBM
        self.particles = _engine.Engine(self.pc, init_cov=init_cov, process_cov=motion_cov,
                                        resample_cov=self.res_noise_cov, meas_std=meas_std, m2o=self.m2o_mat,
                                        seed=int(p['seed']), rng_mode=rng_mode,
                                        resample_scheme=_SCHEMES[p['resample_scheme']], device=int(p['device']))
        self._replay = None  # REPLAY mode: object with randn(n,6) / random_sample(k)
        if rng_mode == _engine.RNG_NATIVE:
            self.particles.init_particles()
        self.poses = _msgs.PoseArray()
        self.poses.header.frame_id = self.odom_frame
        self.loc_pose = _msgs.Odometry()
        self.loc_pose.header.frame_id = self.odom_frame
        self.loc_pose.child_frame_id = self.base_frame
        self.cov = np.zeros((3, 3))
        self.time = 0.0
        self.old_time = 0.0
        self.diving = True  # auv_pf.py:103
        self.odom_latest = None
        self.has_map = False
        self.mbes_range_max = float(p['mbes_range_max'])
        self.mbes_points_frame = p['mbes_points_frame']
        for key in ('map_grid_file', 'map_mesh_file'):   # the node's own map parameters
            if p[key]:
                self.load_map(p[key])
        # landmark detections (config 5)
        self.has_landmarks = False
        self.landmark_std, self.landmark_k = float(p['landmark_std']), int(p['landmark_k'])
        self.landmark_gate, self.landmark_sync_tol = float(p['landmark_gate']), float(p['landmark_sync_tol'])
        self.landmark_max_age = float(p['landmark_max_age'])
        self.landmark_late = str(p['landmark_late'])
        if self.landmark_late not in ('update', 'drop'):
            raise ValueError("landmark_late must be 'update' or 'drop', not %r" % self.landmark_late)
        self._pending_det = None   # (stamp, (n_det, 3) detections in base_frame) that arrived AHEAD of their ping
        self._last_ping_stamp = None
        if p['landmark_map_file']:
            self.set_landmarks(load_landmark_file(p['landmark_map_file'], float(p['rocks_depth'])))

    # ---- REPLAY-mode RNG source (parity runs): rs must offer randn(n, 6) and random_sample(k)
    def set_replay_source(self, rs):
        self._replay = rs
        self.particles.init_particles(rs.randn(self.pc, 6))

    def start_timing(self, stamp):
        """auv_pf.py:96-98: `Start timing now`."""
        self.time = float(stamp)
        self.old_time = float(stamp)

    # ---- callbacks, same names as the reference
    def dive_cb(self, dive_msg):
        self.diving = dive_msg.data

    def odom_callback(self, odom_msg):
        with self.lock:
            self.time = odom_msg.header.stamp.to_sec()
            self.odom_latest = odom_msg
            if self.old_time and self.time > self.old_time:
                self.predict(odom_msg)
            self.old_time = self.time

    def predict(self, odom_t):
        dt = self.time - self.old_time
        tw, po = odom_t.twist.twist, odom_t.pose.pose
        nz = self._replay.randn(self.pc, 6) if self._replay is not None else None
        self.particles.predict([tw.linear.x, tw.linear.y, tw.linear.z], tw.angular.z,
                               [po.orientation.x, po.orientation.y, po.orientation.z, po.orientation.w],
                               po.position.z, dt, nz, stamp=self.time)

    def gps_odom_cb(self, gps_odom):
        with self.lock:
            if self.old_time and not self.diving:
                weights = self.update(gps_odom)
                self.resample(weights)

    def update(self, gps_odom):
        goal_point = _msgs.PointStamped()
        goal_point.header.frame_id = self.utm_frame
        goal_point.point.x = gps_odom.pose.pose.position.x
        goal_point.point.y = gps_odom.pose.pose.position.y
        goal_point.point.z = 0.
        gps_map = self.transport.transformPoint(self.map_frame, goal_point)  # once, not per particle
        self.particles.update_gps(gps_map.point.x, gps_map.point.y)
        return self.particles  # the weights stay in HBM; resample() consumes them there

    def resample(self, weights):
        if self._replay is not None:
            need = self.particles.resample_prepare()
            u = self._replay.random_sample(need) if need != 1 else self._replay.random_sample()
            self.particles.resample(u, self._replay.randn(self.pc, 6))
        else:
            self.particles.resample()

    def reassign_poses(self, lost, dupes):
        """Folded into resample() on the device (auv_pf.py:195-198 semantics)."""
        return None

    # ---- MBES (north_star): map + ping callback
    def set_map_grid(self, z, origin, res):
        with self.lock:
            self.particles.set_map_grid(z, origin, res)
            self.has_map = True

    def set_map_mesh(self, verts, tris):
        with self.lock:
            self.particles.set_map_mesh(verts, tris)
            self.has_map = True

    def load_map(self, path):
        m = load_map_file(path)
        if m[0] == 'grid':
            self.set_map_grid(m[1], m[2], m[3])
        else:
            self.set_map_mesh(m[1], m[2])

    # ---- landmark detections (BASELINE config 5)
    def set_landmarks(self, xyz):
        xyz = np.ascontiguousarray(np.asarray(xyz, np.float64).reshape(-1, 3))
        if xyz.shape[0] == 0:
            raise ValueError('landmark map: no landmarks (all filtered by rocks_depth?)')
        with self.lock:
            self.particles.set_landmarks(xyz)
            self.has_landmarks = True

    def lm_detect_cb(self, lm_msg):
        """geometry_msgs/PoseArray of detections in base_frame (toy_mbes_receptor.cpp:75-105: positions only, stamped
        with the ping's stamp, published after the receptor has processed the ping).  The usual order -- the ping with
        this stamp (or a later one) has already been through mbes_cb --: a measurement update of its own followed by
        the resampling, like a GPS fix.  Ahead of its ping: held until mbes_cb sees the ping with the same stamp, whose
        likelihood it joins (mcl_update_landmarks(accumulate = 1)).  Never applied to another ping's likelihood."""
        det = np.array([[p.position.x, p.position.y, p.position.z] for p in lm_msg.poses], np.float64).reshape(-1, 3)
        if det.shape[0] == 0:
            return
        with self.lock:
            if not (self.old_time and self.has_landmarks):
                return
            stamp = self._stamp_of(lm_msg)
            if self.time - stamp > self.landmark_max_age:
                return
            if self.has_map and (self._last_ping_stamp is None or stamp > self._last_ping_stamp + self.landmark_sync_tol):
                self._pending_det = (stamp, det)
                return
            if self.landmark_late == 'drop' and self.has_map:
                return   # (after its ping, and the node is asked for ONE resampling per ping)
            self.particles.update_landmarks(det, self.landmark_std, k=self.landmark_k, gate=self.landmark_gate,
                                            accumulate=False)
            self.resample(self.particles)

    def _stamp_of(self, msg):
        """header.stamp in seconds; a message without one (hand-made test messages) counts as `now`."""
        stamp = getattr(getattr(msg, 'header', None), 'stamp', None)
        return float(stamp.to_sec()) if stamp is not None else float(self.time)

    def _accumulate_pending_detections(self, ping_stamp):
        """Called under the lock right after an MBES update: detections that arrived ahead of THIS ping onto its
        likelihood; held detections of an earlier ping (which never came) are dropped, of a later one kept."""
        ping_stamp = float(ping_stamp)
        self._last_ping_stamp = ping_stamp
        if self._pending_det is None:
            return
        stamp, det = self._pending_det
        if stamp > ping_stamp + self.landmark_sync_tol:
            return   # (still ahead: its ping is yet to come)
        self._pending_det = None
        if stamp < ping_stamp - self.landmark_sync_tol:
            return   # (its ping never arrived: dropped, never applied to another ping)
        self.particles.update_landmarks(det, self.landmark_std, k=self.landmark_k, gate=self.landmark_gate,
                                        accumulate=True)

    def mbes_pc_cb(self, cloud):
        """One ping as a sensor_msgs/PointCloud2 (or msgs.PointCloud2): every point is a beam's hit.  In the sensor
        frame beam b looks along (0, sin a_b, -cos a_b) (include/mcl.h), so a point gives a_b = atan2(y, -z) and the
        range |p|; points in base_frame (mbes_receptor.cpp:138: the receptor projects the scan into base_frame) are
        taken back through the sensor offset first.  The beams are handed over in ascending angle."""
        pts = _msgs.pointcloud2_xyz(cloud)
        if self.mbes_points_frame != 'sensor':
            o = self.mbes_sensor_offset
            T = _rigid(o[0], o[1], o[2], o[3], o[4], o[5])
            pts = (pts - T[:3, 3]).dot(T[:3, :3])   # R^T (p - t), row-wise
        if pts.shape[0] == 0:
            return
        ang = np.arctan2(pts[:, 1], -pts[:, 2])
        rng = np.sqrt(np.sum(pts * pts, axis=1))
        order = np.argsort(ang, kind='stable')
        with self.lock:
            if not (self.old_time and self.has_map):
                return
            self.particles.update_mbes(rng[order].astype(np.float32), ang[order].astype(np.float32), self.mbes_std,
                                       self.mbes_range_max, self.mbes_sensor_offset)
            self._accumulate_pending_detections(self._stamp_of(cloud))
            self.resample(self.particles)

    def mbes_cb(self, scan):
        with self.lock:
            if not (self.old_time and self.has_map):
                return
            n = len(scan.ranges)
            angles = scan.angle_min + scan.angle_increment * np.arange(n)
            self.particles.update_mbes(np.asarray(scan.ranges, dtype=np.float32), angles.astype(np.float32),
                                       self.mbes_std, float(scan.range_max), self.mbes_sensor_offset)
            self._accumulate_pending_detections(self._stamp_of(scan))
            self.resample(self.particles)

    # ---- publishing, auv_pf.py:218-285
    def update_loc_pose(self, pose_list=None):
        mean, yaw, cov9 = self.particles.mean_cov()
        lp = self.loc_pose
        lp.pose.pose.position.x, lp.pose.pose.position.y, lp.pose.pose.position.z = mean[0], mean[1], mean[2]
        quat_t = quaternion_from_euler(mean[3], mean[4], yaw)
        lp.pose.pose.orientation = _msgs.Quaternion(*quat_t)
        lp.header.stamp = self.transport.now()
        self.cov = np.array(cov9).reshape(3, 3)
        lp.pose.covariance = [0.] * 36
        for i in range(3):
            for j in range(3):
                lp.pose.covariance[i * 3 + j] = float(self.cov[i, j])
        self.transport.publish_odom(lp)
        self.transport.sendTransform([mean[0], mean[1], 0.], quat_t, self.transport.now(), self.base_frame,
                                     self.odom_frame)
        return mean, yaw, cov9

    def loc_loop(self, event=None):
        with self.lock:
            self.poses.data = self.particles.poses()  # (n, 7): position + quaternion_from_euler per particle
            self.poses.header.stamp = self.transport.now()
            self.update_loc_pose()
            self.transport.publish_poses(self.poses)
