"""sam_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""


class ThrusterAngles(object):
    def __init__(self):
        self.thruster_vertical_radians = 0.0
        self.thruster_horizontal_radians = 0.0
