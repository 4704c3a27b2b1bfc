"""Plain-Python message shapes with the field names of the ROS messages the node exchanges
(nav_msgs/Odometry, geometry_msgs/PoseArray, sensor_msgs/LaserScan, std_msgs/Bool).  Used when
rospy is absent (tests, replay tool); with ROS installed the real messages duck-type the same."""


class Time(object):
    def __init__(self, secs=0.0):
        self.secs = float(secs)

    def to_sec(self):
        return self.secs


class Header(object):
    def __init__(self, frame_id='', stamp=None):
        self.frame_id = frame_id
        self.stamp = stamp if stamp is not None else Time(0.0)


class Vector3(object):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


Point = Vector3


class Quaternion(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Pose(object):
    def __init__(self):
        self.position = Point()
        self.orientation = Quaternion()


class PoseWithCovariance(object):
    def __init__(self):
        self.pose = Pose()
        self.covariance = [0.0] * 36


class Twist(object):
    def __init__(self):
        self.linear = Vector3()
        self.angular = Vector3()


class TwistWithCovariance(object):
    def __init__(self):
        self.twist = Twist()
        self.covariance = [0.0] * 36


class Odometry(object):
    def __init__(self):
        self.header = Header()
        self.child_frame_id = ''
        self.pose = PoseWithCovariance()
        self.twist = TwistWithCovariance()


class PoseArray(object):
    def __init__(self):
        self.header = Header()
        self.poses = []
        self.data = None  # (n, 7) numpy block x,y,z,qx,qy,qz,qw -- the bulk payload


class PointStamped(object):
    def __init__(self):
        self.header = Header()
        self.point = Point()


class Bool(object):
    def __init__(self, data=False):
        self.data = data


class LaserScan(object):
    """MBES ping as the legacy front-end publishes it (mbes_processors/mbes_toy_processor/src/
    toy_mbes_manipulator.cpp:69-73: angle_min + k*angle_increment, ranges[k])."""

    def __init__(self, ranges=(), angle_min=0.0, angle_increment=0.0, range_max=100.0):
        self.header = Header()
        self.ranges = list(ranges)
        self.angle_min = angle_min
        self.angle_increment = angle_increment
        self.range_max = range_max


def odometry_from_stream(stream, k):
    """Odometry message k of a synth.odom_stream (what dr_node publishes on /sam/dr/odom)."""
    m = Odometry()
    m.header.stamp = Time(stream['stamp'][k])
    m.twist.twist.linear.x, m.twist.twist.linear.y, m.twist.twist.linear.z = (float(x) for x in stream['v'][k])
    m.twist.twist.angular.z = float(stream['wz'][k])
    q = stream['q'][k]
    m.pose.pose.orientation = Quaternion(float(q[0]), float(q[1]), float(q[2]), float(q[3]))
    m.pose.pose.position.z = float(stream['z'][k])
    return m
