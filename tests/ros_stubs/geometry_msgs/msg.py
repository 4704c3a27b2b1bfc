"""geometry_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""


class Header(object):
    def __init__(self):
        self.frame_id, self.stamp = '', None


class Point(object):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


Vector3 = Point


class Quaternion(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Pose(object):
    def __init__(self):
        self.position, self.orientation = Point(), Quaternion()


class PoseArray(object):
    def __init__(self):
        self.header, self.poses = Header(), []


class PointStamped(object):
    def __init__(self):
        self.header, self.point = Header(), Point()


class PoseWithCovariance(object):
    def __init__(self):
        self.pose, self.covariance = Pose(), [0.0] * 36


class Twist(object):
    def __init__(self):
        self.linear, self.angular = Vector3(), Vector3()


class TwistWithCovariance(object):
    def __init__(self):
        self.twist, self.covariance = Twist(), [0.0] * 36
