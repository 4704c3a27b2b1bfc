"""tf stand-in (TEST INFRASTRUCTURE ONLY)."""
from . import transformations  # noqa: F401


class LookupException(Exception):
    pass


class ConnectivityException(Exception):
    pass


class ExtrapolationException(Exception):
    pass


class TransformListener(object):
    """transformPoint applies a fixed rigid utm->map transform set by the harness."""

    def __init__(self):
        self.utm2map = None  # 4x4
        self.frames = {}     # (target, source) -> (translation, quaternion) for lookupTransform

    def lookupTransform(self, target, source, stamp):
        if (target, source) not in self.frames:
            raise LookupException()
        return self.frames[(target, source)]

    def transformPoint(self, frame, pt):
        import numpy as np
        from geometry_msgs.msg import PointStamped
        if self.utm2map is None:
            raise LookupException()
        v = self.utm2map.dot(np.array([pt.point.x, pt.point.y, pt.point.z, 1.0]))
        out = PointStamped()
        out.header.frame_id = frame
        out.point.x, out.point.y, out.point.z = v[0], v[1], v[2]
        return out


class TransformBroadcaster(object):
    def __init__(self):
        self.sent = []

    def sendTransform(self, trans, rot, stamp, child, parent):
        self.sent.append((list(trans), list(rot), child, parent))
