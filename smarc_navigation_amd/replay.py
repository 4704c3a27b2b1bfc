#!/usr/bin/env python3
"""Recorded-stream replay + evaluation.

Replays a recorded input stream (own .npz format, no rosbag dependency) through the node class
(smarc_navigation_amd/auv_pf.py: the same callbacks the rospy wrapper registers) and scores the
result the way the reference's visual_tools.py does: time-synchronised (GPS, DR, PF) samples accumulated per
callback (odom_cb, visual_tools.py:27-37,80-110), the two error series it plots (:127,:135) and its shutdown
summary (finish_hld :61-76: path length of the three tracks and the norm of each track's final position),
plus the RMSE between the PF mean pose and a reference track (the "pose RMSE vs ref" of the metric).

Stream file (np.savez): stamp[n], v[n,3], wz[n], q[n,4], z[n]  (odometry, /sam/dr/odom);
optional gps_idx[k], gps_xy_utm[k,2]; optional mbes_idx[m], mbes_ranges[m,B], mbes_angles[B],
mbes_range_max; optional dr_xyz[n,3] (dead-reckoning track), truth_xyz[n,3]; t0.
A raw-sensor file (ev_t, ev_kind, ev_data as synth.raw_sensor_events makes them; optional gps_map,
pressure_tf) is first run through the dead-reckoning integrator (--raw).

    python -m smarc_navigation_amd.replay stream.npz --particles 65536 [--map-grid map.npz] [--out traj.csv]

rosbag input (BASELINE config 1 is "rosbag replay"): `--bag` reads a ROS 1 bag (format 2.0, rosbag_io.py: no ROS
installation needed) and runs its messages IN RECORDED ORDER through the node's callbacks -- what `rosbag play` into
the live node does (the reference reads its bags with rosbag.Bag(...).read_messages(),
auv_ekf_localization/rosbags/rosbag_handler.py:8-19):

    python -m smarc_navigation_amd.replay run.bag --bag --odom-topic /sam/dr/odom --gps-topic /sam/dr/gps \
        --mbes-topic /sam/mbes_scan --particles 128
"""
import argparse
import json

import numpy as np


def track_metrics(vec):
    """visual_tools.py:61-76 for one 3 x n track: summed segment lengths and |last position|."""
    vec = np.asarray(vec, dtype=np.float64)
    dist = 0.
    for i in range(1, vec.shape[1]):
        dist += np.linalg.norm(vec[:, i] - vec[:, i - 1])
    final = float(np.linalg.norm(vec[:, -1])) if vec.shape[1] else 0.0
    return float(dist), final


class ApproximateTimeSync(object):
    """The time synchroniser visual_tools.py:27-37 relies on: ros_comm's message_filters
    ApproximateTimeSynchronizer (Python implementation, ROS melodic / noetic 1.14-1.16; a third-party
    dependency, not part of the reference repository), restated from its published algorithm:
    every topic keeps the last `queue_size` messages by stamp; when a message arrives, the other topics'
    stamps within `slop` of it are collected, sorted by distance, and the first combination (itertools
    product order: nearest candidates first) whose span is < slop and whose messages are all still queued
    is delivered and removed.  add(topic_index, stamp, msg) returns the delivered tuple or None."""

    def __init__(self, n_topics, queue_size, slop):
        self.queues = [dict() for _ in range(n_topics)]
        self.queue_size, self.slop = int(queue_size), float(slop)

    def add(self, index, stamp, msg):
        import itertools
        q = self.queues[index]
        q[stamp] = msg
        while len(q) > self.queue_size:
            del q[min(q)]
        others = self.queues[:index] + self.queues[index + 1:]
        cands = []
        for oq in others:
            ts = sorted(((s, abs(s - stamp)) for s in oq if abs(s - stamp) <= self.slop), key=lambda x: x[1])
            if not ts:
                return None
            cands.append([s for s, _ in ts])
        for vv in itertools.product(*cands):
            vv = list(vv)
            vv.insert(index, stamp)
            if (max(vv) - min(vv)) < self.slop and all(t in qq for qq, t in zip(self.queues, vv)):
                out = tuple(qq[t] for qq, t in zip(self.queues, vv))
                for qq, t in zip(self.queues, vv):
                    del qq[t]
                return out
        return None


class DRStats(object):
    """Mirror of visual_tools.py's DRStatsVisualization (the reference's evaluation node): GPS fix (utm),
    dead-reckoning odometry and PF odometry, time-synchronised (queue 20, slop 20 s, :27-37), accumulated per
    sample (odom_cb :80-110: the fix is transformed utm -> odom frame with z = 0; the three 3 x k arrays START
    with a zero column, :48-50), scored at shutdown (finish_hld :61-76) and as the two error series the node
    plots (visualize :127,:135).  `utm2odom`: 4x4, or None while the transform is unavailable (the triple
    is then dropped like the node drops it, :108-109)."""

    def __init__(self, queue_size=20, slop=20.0):
        self.sync = ApproximateTimeSync(3, queue_size, slop)
        self.filter_cnt = 1
        self.gps_odom_vec = np.zeros((3, 1))
        self.dr_odom_vec = np.zeros((3, 1))
        self.pf_odom_vec = np.zeros((3, 1))
        self.utm2odom = None

    def odom_cb(self, gps_xyz_utm, dr_xyz, pf_xyz):
        if self.utm2odom is None:
            return False
        g = np.asarray(self.utm2odom, dtype=np.float64).dot([gps_xyz_utm[0], gps_xyz_utm[1], 0.0, 1.0])[:3]
        self.gps_odom_vec = np.hstack((self.gps_odom_vec, g.reshape((3, 1))))
        self.dr_odom_vec = np.hstack((self.dr_odom_vec, np.asarray(dr_xyz, dtype=np.float64).reshape((3, 1))))
        self.pf_odom_vec = np.hstack((self.pf_odom_vec, np.asarray(pf_xyz, dtype=np.float64).reshape((3, 1))))
        self.filter_cnt += 1
        return True

    def add(self, topic, stamp, xyz):
        """topic: 0 gps (utm), 1 dead reckoning, 2 particle filter; runs odom_cb on every synchronised triple"""
        hit = self.sync.add(topic, float(stamp), np.asarray(xyz, dtype=np.float64))
        return self.odom_cb(*hit) if hit is not None else False

    def error_series(self):
        """|gps - pf| and |gps - dr| per accumulated sample (visual_tools.py:127,:135)"""
        return (np.linalg.norm(self.gps_odom_vec - self.pf_odom_vec, axis=0),
                np.linalg.norm(self.gps_odom_vec - self.dr_odom_vec, axis=0))

    def finish_hld(self):
        """the six numbers the node prints at shutdown (visual_tools.py:61-76)"""
        out = {}
        for name, vec in (('GPS', self.gps_odom_vec), ('DR', self.dr_odom_vec), ('PF', self.pf_odom_vec)):
            out[name + ' distance'], out[name + ' final error'] = track_metrics(vec)
        return out


def pose_rmse(est_xy, ref_xy):
    d = np.asarray(est_xy, dtype=np.float64) - np.asarray(ref_xy, dtype=np.float64)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))


def odom_stream_from_raw(t, kind, data, gps_map=None, pressure_tf=None, dvl_period=0.2, dr_period=0.02):
    """Raw IMU / DVL / depth / thruster events (synth.raw_sensor_events format) -> the odometry stream
    `replay` consumes, through the dead-reckoning integrator (dr.py): one sample
    per published timer tick, stamped with the tick's event time.  Also returns the map -> odom
    transform the integrator fixed from the first usable GPS fix (4x4) for the filter's m2o."""
    from . import dr, synth
    from . import auv_pf as node
    ticks, m2o = dr.replay_events(t, kind, data, gps_map=gps_map, pressure_tf=pressure_tf, dvl_period=dvl_period,
                                  dr_period=dr_period)
    tick_t = np.asarray(t)[np.asarray(kind) == synth.EV_TICK]
    pub = ticks[:, 0] > 0
    stream = dict(stamp=tick_t[pub], v=ticks[pub, 8:11], wz=ticks[pub, 13], q=ticks[pub, 4:8], z=ticks[pub, 3],
                  dr_xyz=ticks[pub, 1:4])
    if pub.any():
        stream['t0'] = float(stream['stamp'][0]) - dr_period
    m2o_mat = None if np.isnan(m2o).any() else node.matrix_from_tf(m2o[0:3], m2o[3:7])
    return stream, m2o_mat


def replay(stream, params=None, m2o=None, utm2map=None, grid=None, mesh=None, publish_every=5):
    """Drive the node with a recorded stream; returns dict(pf_xyz[n_pub,3], pub_idx, summary)."""
    from . import auv_pf as node
    from . import msgs
    tr = node.RecordingTransport(utm2map)
    pf = node.auv_pf(params or {}, m2o_mat=m2o, transport=tr)
    if grid is not None:
        pf.set_map_grid(grid['z'], grid['origin'], float(grid['res']))
    if mesh is not None:
        pf.set_map_mesh(mesh['verts'], mesh['tris'])
    n = len(stream['stamp'])
    pf.start_timing(float(stream['t0']) if 't0' in stream else float(stream['stamp'][0]) - 0.02)
    gps_at = {int(k): j for j, k in enumerate(stream['gps_idx'])} if 'gps_idx' in stream else {}
    mbes_at = {int(k): j for j, k in enumerate(stream['mbes_idx'])} if 'mbes_idx' in stream else {}
    pub_idx, pf_xyz = [], []
    stats = DRStats() if ('gps_idx' in stream and 'dr_xyz' in stream) else None
    if stats is not None:
        # visual_tools transforms the fix into the odom frame: (map <- odom)^-1 (map <- utm)
        u2m = np.identity(4) if utm2map is None else np.asarray(utm2map, dtype=np.float64)
        stats.utm2odom = np.linalg.inv(np.identity(4) if m2o is None else np.asarray(m2o, dtype=np.float64)).dot(u2m)
    for k in range(n):
        pf.odom_callback(msgs.odometry_from_stream(stream, k))
        if stats is not None:
            stats.add(1, stream['stamp'][k], stream['dr_xyz'][k])
            if k in gps_at:
                stats.add(0, stream['stamp'][k], list(stream['gps_xy_utm'][gps_at[k]]) + [0.0])
        if k in gps_at:
            g = msgs.Odometry()
            g.pose.pose.position.x = float(stream['gps_xy_utm'][gps_at[k]][0])
            g.pose.pose.position.y = float(stream['gps_xy_utm'][gps_at[k]][1])
            pf.dive_cb(msgs.Bool(False))
            pf.gps_odom_cb(g)
        if k in mbes_at:
            ang = np.asarray(stream['mbes_angles'], dtype=np.float64)
            scan = msgs.LaserScan(stream['mbes_ranges'][mbes_at[k]], float(ang[0]),
                                  float(ang[1] - ang[0]) if ang.size > 1 else 0.0,
                                  float(stream['mbes_range_max']) if 'mbes_range_max' in stream else 100.0)
            pf.mbes_cb(scan)
        if (k + 1) % publish_every == 0 or k == n - 1:
            pf.loc_loop(None)
            p = tr.odom_corrected[-1].pose.pose.position
            pub_idx.append(k)
            pf_xyz.append([p.x, p.y, p.z])
            if stats is not None:
                stats.add(2, stream['stamp'][k], [p.x, p.y, p.z])
    pf_xyz = np.array(pf_xyz)
    summary = {}
    summary['pf_distance'], summary['pf_final'] = track_metrics(pf_xyz.T)
    for name in ('dr_xyz', 'truth_xyz'):
        if name in stream:
            ref = np.asarray(stream[name])[pub_idx]
            tag = name.split('_')[0]
            summary[tag + '_distance'], summary[tag + '_final'] = track_metrics(ref.T)
            summary['pf_rmse_vs_' + tag] = pose_rmse(pf_xyz[:, :2], ref[:, :2])
    out = dict(pf_xyz=pf_xyz, pub_idx=np.array(pub_idx), summary=summary)
    if stats is not None:
        out['stats'] = stats
        out['err_gps_pf'], out['err_gps_dr'] = stats.error_series()
        summary['visual_tools'] = stats.finish_hld()
    return out


def stream_to_bag(path, stream, odom_topic='/sam/dr/odom', gps_topic='/sam/dr/gps', mbes_topic='/sam/mbes_scan',
                  dive_topic='/dive', compression='none'):
    """A stream file's content as a rosbag 2.0 file (rosbag_io.write_bag): one nav_msgs/Odometry per sample, the GPS
    fixes (nav_msgs/Odometry in utm, preceded once by std_msgs/Bool false on the dive topic -- the node ignores fixes
    while `diving`, auv_pf.py:103,126) and the pings (sensor_msgs/LaserScan) right after the sample they belong to."""
    from . import msgs, rosbag_io
    out = []
    n = len(stream['stamp'])
    gps_at = {int(k): j for j, k in enumerate(stream['gps_idx'])} if 'gps_idx' in stream else {}
    mbes_at = {int(k): j for j, k in enumerate(stream['mbes_idx'])} if 'mbes_idx' in stream else {}
    surfaced = False
    for k in range(n):
        t = float(stream['stamp'][k])
        od = msgs.odometry_from_stream(stream, k)
        od.header.frame_id, od.child_frame_id = 'sam/odom', 'sam/base_link'
        out.append((odom_topic, 'nav_msgs/Odometry', od, t))
        if k in gps_at:
            if not surfaced:
                out.append((dive_topic, 'std_msgs/Bool', msgs.Bool(False), t))
                surfaced = True
            g = msgs.Odometry()
            g.header.stamp, g.header.frame_id = msgs.Time(t), 'utm'
            g.pose.pose.position.x = float(stream['gps_xy_utm'][gps_at[k]][0])
            g.pose.pose.position.y = float(stream['gps_xy_utm'][gps_at[k]][1])
            out.append((gps_topic, 'nav_msgs/Odometry', g, t))
        if k in mbes_at:
            ang = np.asarray(stream['mbes_angles'], dtype=np.float64)
            scan = msgs.LaserScan(np.asarray(stream['mbes_ranges'][mbes_at[k]], np.float32), float(ang[0]),
                                  float(ang[1] - ang[0]) if ang.size > 1 else 0.0,
                                  float(stream['mbes_range_max']) if 'mbes_range_max' in stream else 100.0)
            scan.header.stamp, scan.header.frame_id = msgs.Time(t), 'sam/mbes_link'
            out.append((mbes_topic, 'sensor_msgs/LaserScan', scan, t))
    rosbag_io.write_bag(path, out, compression=compression)
    return len(out)


def replay_bag(path, params=None, m2o=None, utm2map=None, grid=None, mesh=None, odom_topic='/sam/dr/odom',
               gps_topic='/sam/dr/gps', mbes_topic='/sam/mbes_scan', mbes_cloud_topic=None, dive_topic='/dive',
               lm_detect_topic=None, publish_period=0.1, t0=None):
    """BASELINE config 1, literally: a recorded ROS 1 bag's messages in recorded order through the node's callbacks
    (odom_callback, gps_odom_cb, dive_cb, mbes_cb, mbes_pc_cb, lm_detect_cb), loc_loop every `publish_period` seconds
    of bag time (auv_pf.py:114: a 10 Hz timer).  Returns dict(pf_xyz[k,3], pf_stamp[k], counts{topic: messages},
    summary); no ROS installation is needed (rosbag_io.Bag)."""
    from . import auv_pf as node
    from . import rosbag_io
    tr = node.RecordingTransport(utm2map)
    pf = node.auv_pf(params or {}, m2o_mat=m2o, transport=tr)
    if grid is not None:
        pf.set_map_grid(grid['z'], grid['origin'], float(grid['res']))
    if mesh is not None:
        pf.set_map_mesh(mesh['verts'], mesh['tris'])
    route = {odom_topic: pf.odom_callback, gps_topic: pf.gps_odom_cb, dive_topic: pf.dive_cb, mbes_topic: pf.mbes_cb}
    if mbes_cloud_topic:
        route[mbes_cloud_topic] = pf.mbes_pc_cb
    if lm_detect_topic:
        route[lm_detect_topic] = pf.lm_detect_cb
    counts, pf_xyz, pf_stamp = {}, [], []
    started, next_pub, last_t = False, None, None
    for topic, msg, t in rosbag_io.Bag(path).read_messages(topics=list(route)):
        if not started:
            # "Start timing now" (auv_pf.py:96-98): the node is up before the first message
            pf.start_timing(float(t0) if t0 is not None else t - 0.02)
            next_pub, started = t + publish_period, True
        while t >= next_pub:   # the 10 Hz timer, in bag time
            pf.loc_loop(None)
            p = tr.odom_corrected[-1].pose.pose.position
            pf_xyz.append([p.x, p.y, p.z])
            pf_stamp.append(next_pub)
            next_pub += publish_period
        route[topic](msg)
        counts[topic] = counts.get(topic, 0) + 1
        last_t = t
    if started:
        pf.loc_loop(None)
        p = tr.odom_corrected[-1].pose.pose.position
        pf_xyz.append([p.x, p.y, p.z])
        pf_stamp.append(last_t)
    pf_xyz = np.array(pf_xyz).reshape(-1, 3)
    summary = {'messages': counts}
    if len(pf_xyz):
        summary['pf_distance'], summary['pf_final'] = track_metrics(pf_xyz.T)
    return dict(pf_xyz=pf_xyz, pf_stamp=np.array(pf_stamp), counts=counts, summary=summary, node=pf)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('stream')
    ap.add_argument('--particles', type=int, default=4096)
    ap.add_argument('--map-grid', help='npz with z, origin, res')
    ap.add_argument('--out', help='CSV of the published mean pose')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--raw', action='store_true', help='the file holds raw sensor events: integrate them first')
    ap.add_argument('--bag', action='store_true', help='the file is a ROS 1 bag (format 2.0): replay its messages in recorded order')
    ap.add_argument('--odom-topic', default='/sam/dr/odom')
    ap.add_argument('--gps-topic', default='/sam/dr/gps')
    ap.add_argument('--mbes-topic', default='/sam/mbes_scan')
    ap.add_argument('--dive-topic', default='/dive')
    a = ap.parse_args(argv)
    if a.bag:
        grid = dict(np.load(a.map_grid)) if a.map_grid else None
        res = replay_bag(a.stream, dict(particle_count=a.particles, seed=a.seed), grid=grid, odom_topic=a.odom_topic,
                         gps_topic=a.gps_topic, mbes_topic=a.mbes_topic, dive_topic=a.dive_topic)
        if a.out:
            np.savetxt(a.out, np.column_stack([res['pf_stamp'], res['pf_xyz']]), delimiter=',', header='stamp,x,y,z')
        print(json.dumps(res['summary']))
        return
    stream = dict(np.load(a.stream, allow_pickle=False))
    m2o = None
    if a.raw:
        ptf = stream.get('pressure_tf')
        if ptf is not None and np.isnan(ptf).any():
            ptf = None
        stream, m2o = odom_stream_from_raw(stream['ev_t'], stream['ev_kind'], stream['ev_data'],
                                           gps_map=stream.get('gps_map'), pressure_tf=ptf)
    grid = dict(np.load(a.map_grid)) if a.map_grid else None
    res = replay(stream, dict(particle_count=a.particles, seed=a.seed), m2o=m2o, grid=grid)
    if a.out:
        np.savetxt(a.out, np.column_stack([res['pub_idx'], res['pf_xyz']]), delimiter=',', header='step,x,y,z')
    print(json.dumps(res['summary']))


if __name__ == '__main__':
    main()
