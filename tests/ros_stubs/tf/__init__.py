"""tf stand-in (TEST INFRASTRUCTURE ONLY)."""
import numpy as np

utm2map = np.identity(4)
broadcasts = []


class LookupException(Exception):
    pass


class TransformListener(object):
    def transformPoint(self, frame, pt):
        from geometry_msgs.msg import PointStamped
        v = utm2map.dot(np.array([pt.point.x, pt.point.y, pt.point.z, 1.0]))
        out = PointStamped()
        out.header.frame_id = frame
        out.point.x, out.point.y, out.point.z = float(v[0]), float(v[1]), float(v[2])
        return out


class TransformBroadcaster(object):
    def sendTransform(self, trans, rot, stamp, child, parent):
        broadcasts.append((list(trans), list(rot), child, parent))
