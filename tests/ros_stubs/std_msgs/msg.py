"""std_msgs stand-ins (TEST INFRASTRUCTURE ONLY)."""


class Bool(object):
    def __init__(self, data=False):
        self.data = data
