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


class PointField(object):
    """sensor_msgs/PointField: datatype 7 = FLOAT32, 8 = FLOAT64."""
    FLOAT32, FLOAT64 = 7, 8

    def __init__(self, name='', offset=0, datatype=7, count=1):
        self.name, self.offset, self.datatype, self.count = name, offset, datatype, count


class PointCloud2(object):
    """sensor_msgs/PointCloud2 as the MBES front-ends of the reference stack publish a ping
    (mbes_processors/mbes_mapper/src/mbes_receptor.cpp:126-165: one LaserScan projected into base_frame)."""

    def __init__(self):
        self.header = Header()
        self.height, self.width = 1, 0
        self.fields = []
        self.is_bigendian = False
        self.point_step, self.row_step = 0, 0
        self.data = b''
        self.is_dense = True


def pointcloud2_from_xyz(xyz, frame_id='', stamp=None, dtype='f4'):
    """Pack an (n, 3) array the way pcl::toROSMsg / laser_geometry do: x, y, z as FLOAT32 at offsets 0, 4, 8 in
    16-byte points (dtype 'f8': FLOAT64 at 0, 8, 16 in 24-byte points)."""
    import numpy as np
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    m = PointCloud2()
    m.header = Header(frame_id, stamp)
    m.width = xyz.shape[0]
    size = 4 if dtype == 'f4' else 8
    m.point_step = 16 if dtype == 'f4' else 24
    m.row_step = m.point_step * m.width
    kind = PointField.FLOAT32 if dtype == 'f4' else PointField.FLOAT64
    m.fields = [PointField(n, k * size, kind, 1) for k, n in enumerate('xyz')]
    buf = np.zeros((m.width, m.point_step), np.uint8)
    for k in range(3):
        buf[:, k * size:(k + 1) * size] = xyz[:, k].astype('<' + dtype).view(np.uint8).reshape(-1, size)
    m.data = buf.tobytes()
    return m


def pointcloud2_xyz(msg):
    """(n, 3) float64 x, y, z of a sensor_msgs/PointCloud2 (real message or the class above), NaN points dropped.
    Reads the byte layout the message declares (fields / point_step / row_step / endianness) -- no dependency on
    sensor_msgs.point_cloud2."""
    import numpy as np
    off, typ = {}, {}
    for f in msg.fields:
        if f.name in ('x', 'y', 'z'):
            off[f.name], typ[f.name] = int(f.offset), int(f.datatype)
    if set(off) != set('xyz'):
        raise ValueError('PointCloud2 without x / y / z fields')
    n = int(msg.width) * int(msg.height)
    raw = np.frombuffer(bytes(msg.data), dtype=np.uint8)
    step, row = int(msg.point_step), int(msg.row_step) or int(msg.point_step) * int(msg.width)
    if int(msg.height) > 1 and row != step * int(msg.width):   # padded rows
        raw = raw.reshape(int(msg.height), row)[:, :step * int(msg.width)].reshape(-1)
    pts = raw[:n * step].reshape(n, step)
    out = np.zeros((n, 3))
    for k, name in enumerate('xyz'):
        if typ[name] not in (PointField.FLOAT32, PointField.FLOAT64):
            raise ValueError('PointCloud2 field %s is not FLOAT32 / FLOAT64' % name)
        size = 4 if typ[name] == PointField.FLOAT32 else 8
        dt = ('>' if msg.is_bigendian else '<') + ('f4' if size == 4 else 'f8')
        out[:, k] = np.ascontiguousarray(pts[:, off[name]:off[name] + size]).view(dt).reshape(-1)
    return out[np.all(np.isfinite(out), axis=1)]


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
