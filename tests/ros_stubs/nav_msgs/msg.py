"""nav_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""
from geometry_msgs.msg import Header, PoseWithCovariance, TwistWithCovariance


class Odometry(object):
    def __init__(self):
        self.header, self.child_frame_id = Header(), ''
        self.pose, self.twist = PoseWithCovariance(), TwistWithCovariance()
