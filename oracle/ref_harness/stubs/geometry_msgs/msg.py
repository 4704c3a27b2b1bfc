"""geometry_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""


class _Header(object):
    def __init__(self):
        self.frame_id = ''
        self.stamp = None


class Point(object):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class Vector3(Point):
    pass


class Quaternion(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Pose(object):
    def __init__(self):
        self.position = Point()
        self.orientation = Quaternion()


class PoseArray(object):
    def __init__(self):
        self.header = _Header()
        self.poses = []


class PointStamped(object):
    def __init__(self):
        self.header = _Header()
        self.point = Point()


class PoseWithCovariance(object):
    def __init__(self):
        self.pose = Pose()
        self.covariance = [0.0] * 36


class PoseWithCovarianceStamped(object):
    def __init__(self):
        self.header = _Header()
        self.pose = PoseWithCovariance()


class Twist(object):
    def __init__(self):
        self.linear = Vector3()
        self.angular = Vector3()


class TwistWithCovariance(object):
    def __init__(self):
        self.twist = Twist()
        self.covariance = [0.0] * 36


class Transform(object):
    def __init__(self):
        self.translation = Vector3()
        self.rotation = Quaternion()


class TransformStamped(object):
    def __init__(self):
        self.header = _Header()
        self.child_frame_id = ''
        self.transform = Transform()
