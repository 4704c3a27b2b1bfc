"""sensor_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""
from geometry_msgs.msg import _Header, Quaternion, Vector3


class Imu(object):
    def __init__(self):
        self.header = _Header()
        self.orientation = Quaternion()
        self.angular_velocity = Vector3()
        self.linear_acceleration = Vector3()
