"""nav_msgs stand-in (TEST INFRASTRUCTURE ONLY)."""
from geometry_msgs.msg import _Header, PoseWithCovariance, TwistWithCovariance


class Odometry(object):
    def __init__(self):
        self.header = _Header()
        self.child_frame_id = ''
        self.pose = PoseWithCovariance()
        self.twist = TwistWithCovariance()
