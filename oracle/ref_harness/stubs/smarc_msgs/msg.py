"""smarc_msgs stand-ins (TEST INFRASTRUCTURE ONLY): the fields dr_node.py reads."""
from geometry_msgs.msg import _Header, Vector3


class DVL(object):
    def __init__(self):
        self.header = _Header()
        self.velocity = Vector3()


class _Rpm(object):
    def __init__(self):
        self.rpm = 0


class ThrusterFeedback(object):
    def __init__(self):
        self.header = _Header()
        self.rpm = _Rpm()
