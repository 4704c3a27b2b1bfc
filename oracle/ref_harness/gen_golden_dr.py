#!/usr/bin/env python3
"""Golden fixtures for the dead-reckoning integrator, made by IMPORTING the
reference's sam_dead_reckoning/scripts/dr_node.py (+ sam_mm.py) read-only behind ROS stand-ins and
driving its callbacks with the event stream of smarc_navigation_amd.synth.raw_sensor_events.

TEST INFRASTRUCTURE; runs only in the development container.  The committed .npz files hold data
only: the events and every Odometry / tf the reference published.
Re-run:  python oracle/ref_harness/gen_golden_dr.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/sam_dead_reckoning/scripts'
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import rospy  # noqa: E402  (stub)
from geometry_msgs.msg import PoseWithCovarianceStamped  # noqa: E402
from nav_msgs.msg import Odometry  # noqa: E402
from sensor_msgs.msg import Imu  # noqa: E402
from smarc_msgs.msg import DVL, ThrusterFeedback  # noqa: E402
from sam_msgs.msg import ThrusterAngles  # noqa: E402
import dr_node as ref_dr  # noqa: E402  (reference)

from smarc_navigation_amd import synth  # noqa: E402

import manifest  # noqa: E402  (this directory: where to write, and the fixture hashes)

OUT = manifest.golden_dir()


def run(scenario, pressure_tf, dvl_period, dr_period, utm2map):
    rospy.set_params({'dvl_period': dvl_period, 'dr_period': dr_period})
    node = ref_dr.VehicleDR()  # the stand-in rospy.spin() returns at once
    node.listener.utm2map = utm2map
    if pressure_tf is not None:
        node.listener.frames[(node.base_frame, node.press_frame)] = (list(pressure_tf), [0.0, 0.0, 0.0, 1.0])
    t, kind, data = synth.raw_sensor_events(scenario=scenario)
    ticks = []
    m2o = np.full(7, np.nan)
    sink = io.StringIO()
    for ti, ki, d in zip(t, kind, data):
        rospy.Time._now = float(ti)
        if ki == synth.EV_IMU:
            m = Imu()
            m.header.stamp = rospy.Time(float(ti))
            m.orientation.x, m.orientation.y, m.orientation.z, m.orientation.w = d[0:4]
            m.angular_velocity.x, m.angular_velocity.y, m.angular_velocity.z = d[4:7]
            node.stim_cb(m)
        elif ki == synth.EV_HEADING:
            m = Imu()
            m.orientation.x, m.orientation.y, m.orientation.z, m.orientation.w = d[0:4]
            node.sbg_cb(m)
        elif ki == synth.EV_GPS:
            if node.gps_sub.registered:
                m = Odometry()
                m.pose.pose.position.x, m.pose.pose.position.y = d[0], d[1]
                n0 = len(node.static_tf_bc.sent)
                node.gps_cb(m)
                if len(node.static_tf_bc.sent) > n0:
                    ts = node.static_tf_bc.sent[-1].transform
                    m2o[:] = [ts.translation.x, ts.translation.y, ts.translation.z,
                              ts.rotation.x, ts.rotation.y, ts.rotation.z, ts.rotation.w]
        elif ki == synth.EV_DVL:
            m = DVL()
            m.header.stamp = rospy.Time(float(ti))
            m.velocity.x, m.velocity.y, m.velocity.z = d[0:3]
            node.dvl_cb(m)
        elif ki == synth.EV_DEPTH:
            m = PoseWithCovarianceStamped()
            m.pose.pose.position.z = d[0]
            node.depth_cb(m)
        elif ki == synth.EV_THRUST:
            a, b = ThrusterFeedback(), ThrusterFeedback()
            a.rpm.rpm, b.rpm.rpm = int(d[0]), int(d[1])
            node.thrust_cb(a, b)
        elif ki == synth.EV_THRUST_CMD:
            m = ThrusterAngles()
            m.thruster_horizontal_radians = d[0]
            node.thrust_cmd_cb(m)
        elif ki == synth.EV_TICK:
            n0 = len(node.pub_odom.sent)
            with contextlib.redirect_stdout(sink):  # the reference prints the velocity every tick
                node.dr_timer(None)
            row = np.full(15, np.nan)
            row[0] = 0.0
            if len(node.pub_odom.sent) > n0:
                o = node.pub_odom.sent[-1]
                p, q = o.pose.pose.position, o.pose.pose.orientation
                lv, av = o.twist.twist.linear, o.twist.twist.angular
                row[:] = [1.0, p.x, p.y, p.z, q.x, q.y, q.z, q.w, lv.x, lv.y, lv.z, av.x, av.y, av.z, node.t_now]
            ticks.append(row)
    ticks = np.array(ticks)
    # utm -> map is applied by tf in the node; the fixture carries the map-frame fixes the node saw
    gps_map = np.array([utm2map.dot([d[0], d[1], 0.0, 1.0])[:2] for d in data[kind == synth.EV_GPS]])
    return dict(scenario=scenario, ev_t=t, ev_kind=kind, ev_data=data, ticks=ticks, m2o=m2o,
                pressure_tf=np.array(pressure_tf if pressure_tf is not None else [np.nan] * 3),
                dvl_period=dvl_period, dr_period=dr_period, gps_map=gps_map, utm2map=utm2map)


def main():
    utm2map = synth.rigid_matrix(-3.0, 4.5, 0.0, 0.0, 0.0, 0.4)
    a = run('auv', (0.35, 0.0, 0.06), 0.2, 0.02, utm2map)
    b = run('surface', None, 0.2, 0.02, utm2map)
    for name, g in (('dr_auv', a), ('dr_surface', b)):
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **g)
        tk = g['ticks']
        pub = tk[:, 0] > 0
        print(name, 'ticks', len(tk), 'published', int(pub.sum()), 'final xy', tk[pub][-1, 1:3], 'm2o', g['m2o'])
    manifest.record(OUT, ['dr_auv.npz', 'dr_surface.npz'], 'oracle/ref_harness/gen_golden_dr.py', needs_reference=True)


if __name__ == '__main__':
    main()
