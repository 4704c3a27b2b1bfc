#!/usr/bin/env python
"""rospy wrapper: node `auv_pf`, same private parameters, topics, frames and message types as
auv_particle_filter/scripts/auv_pf.py; the numerics run in libmcl_hip.so.  Beyond the reference:
`~map_grid_file` / `~map_mesh_file` load the bathymetric map, `~mbes_topic` (sensor_msgs/LaserScan) and
`~mbes_pointcloud_topic` (sensor_msgs/PointCloud2, the form mbes_mapper's receptor gives a ping) feed the MBES update;
`~landmark_map_file` + `~lm_detect_topic` (geometry_msgs/PoseArray, toy_mbes_receptor.cpp:68-110) the landmark k-NN
update of BASELINE config 5.

ROS is not installed in the build container: the module imports whatever `rospy` / `tf` / `tf2_ros` / `*_msgs` are on
the path -- a ROS 1 installation, or the stand-ins of tests/ros_stubs that tests/test_ros_node_stub.py drives main()
with, end to end.  rosrun auv_particle_filter_hip ros_node.py, or auv_pf.launch."""
import sys

import numpy as np

from smarc_navigation_amd import auv_pf as _node

try:  # pragma: no cover - needs a ROS installation
    import rospy
    import tf
    import tf2_ros
    from geometry_msgs.msg import Pose, PoseArray, Quaternion
    from nav_msgs.msg import Odometry
    from sensor_msgs.msg import LaserScan, PointCloud2
    from std_msgs.msg import Bool
    HAVE_ROS = True
except ImportError:
    HAVE_ROS = False


class RosTransport(object):
    """Publishers / tf of auv_pf.py:62-74 behind the transport interface of the mirror class."""

    def __init__(self, params, map_frame, max_poses):
        self.pf_pub = rospy.Publisher(params['particle_poses_topic'], PoseArray, queue_size=10)
        self.loc_pub = rospy.Publisher(params['odom_corrected_topic'], Odometry, queue_size=100)
        self.loc_tf = tf.TransformBroadcaster()
        self.listener = tf.TransformListener()
        self.map_frame = map_frame
        self.max_poses = max_poses

    def transformPoint(self, frame, pt):
        from geometry_msgs.msg import PointStamped
        g = PointStamped()
        g.header.frame_id = pt.header.frame_id
        g.header.stamp = rospy.Time(0)
        g.point.x, g.point.y, g.point.z = pt.point.x, pt.point.y, pt.point.z
        return self.listener.transformPoint(frame, g)

    def publish_poses(self, msg):
        out = PoseArray()
        out.header.frame_id = msg.header.frame_id
        out.header.stamp = rospy.Time.now()
        data = msg.data
        stride = max(1, int(np.ceil(data.shape[0] / float(self.max_poses))))
        for row in data[::stride]:
            p = Pose()
            p.position.x, p.position.y, p.position.z = row[0], row[1], row[2]
            p.orientation = Quaternion(row[3], row[4], row[5], row[6])
            out.poses.append(p)
        self.pf_pub.publish(out)

    def publish_odom(self, m):
        out = Odometry()
        out.header.frame_id, out.child_frame_id = m.header.frame_id, m.child_frame_id
        out.header.stamp = rospy.Time.now()
        out.pose.pose.position.x = m.pose.pose.position.x
        out.pose.pose.position.y = m.pose.pose.position.y
        out.pose.pose.position.z = m.pose.pose.position.z
        o = m.pose.pose.orientation
        out.pose.pose.orientation = Quaternion(o.x, o.y, o.z, o.w)
        out.pose.covariance = list(m.pose.covariance)
        self.loc_pub.publish(out)

    def sendTransform(self, trans, rot, stamp, child, parent):
        self.loc_tf.sendTransform(trans, rot, rospy.Time.now(), child, parent)

    def now(self):
        return rospy.Time.now()


def main():
    if not HAVE_ROS:
        sys.stderr.write('ros_node.py: rospy/tf not importable; this wrapper needs a ROS 1 environment\n')
        return 2
    rospy.init_node('auv_pf', disable_signals=False)
    params = {}
    for key, default in _node.DEFAULT_PARAMS.items():
        params[key] = rospy.get_param('~' + key, default)
    # Transforms from auv_2_ros (auv_pf.py:76-87): map <- odom, wait up to 60 s, exit quietly on failure
    buf = tf2_ros.Buffer()
    tf2_ros.TransformListener(buf)
    try:
        rospy.loginfo("Waiting for transforms")
        t = buf.lookup_transform(params['map_frame'], params['odom_frame'], rospy.Time(0), rospy.Duration(60)).transform
        m2o = _node.matrix_from_tf((t.translation.x, t.translation.y, t.translation.z),
                                   (t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w))
        rospy.loginfo("PF: got transform %s to %s" % (params['map_frame'], params['odom_frame']))
    except Exception:
        rospy.logerr("PF: Could not lookup transform %s to %s" % (params['map_frame'], params['odom_frame']))
        return 1
    transport = RosTransport(params, params['map_frame'], int(rospy.get_param('~max_published_poses', 5000)))
    try:
        pf = _node.auv_pf(params, m2o_mat=m2o, transport=transport)   # (loads ~map_grid_file / ~map_mesh_file)
    except (IOError, OSError, ValueError, KeyError, TypeError) as ex:
        rospy.logerr("PF: could not load the map: %s" % ex)
        return 1
    if not pf.has_map:
        rospy.logwarn("PF: no ~map_grid_file / ~map_mesh_file: MBES pings will be ignored (GPS updates only)")
    pf.start_timing(rospy.Time.now().to_sec())
    rospy.Subscriber(params['aux_dive'], Bool, pf.dive_cb, queue_size=100)
    rospy.Subscriber(params['gps_odom_topic'], Odometry, pf.gps_odom_cb, queue_size=100)
    rospy.Subscriber(params['mbes_topic'], LaserScan, pf.mbes_cb, queue_size=10)
    if params['mbes_pointcloud_topic']:
        rospy.Subscriber(params['mbes_pointcloud_topic'], PointCloud2, pf.mbes_pc_cb, queue_size=10)
    if pf.has_landmarks:   # config 5: detections of the MBES receptors (toy_mbes_receptor.cpp:39 publishes them)
        rospy.Subscriber(params['lm_detect_topic'], PoseArray, pf.lm_detect_cb, queue_size=10)
    rospy.Subscriber(params['odom_topic'], Odometry, pf.odom_callback, queue_size=100)
    rospy.Timer(rospy.Duration(0.1), pf.loc_loop)
    rospy.loginfo("Particle filter class successfully created")
    rospy.spin()
    return 0


if __name__ == '__main__':
    sys.exit(main())
