"""tf2_ros stand-in (TEST INFRASTRUCTURE ONLY)."""


class Buffer(object):
    def lookup_transform(self, *a, **k):
        raise RuntimeError("no tf in harness")


class TransformListener(object):
    def __init__(self, buf):
        pass


class StaticTransformBroadcaster(object):
    def __init__(self):
        self.sent = []

    def sendTransform(self, ts):
        self.sent.append(ts)
