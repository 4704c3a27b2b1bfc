"""tf2_ros stand-in (TEST INFRASTRUCTURE ONLY): lookup_transform answers from a table the test fills."""
transforms = {}   # (target, source) -> ((x, y, z), (qx, qy, qz, qw))


class _V(object):
    pass


class Buffer(object):
    def lookup_transform(self, target, source, stamp, timeout=None):
        if (target, source) not in transforms:
            raise RuntimeError('no transform %s <- %s' % (target, source))
        t, q = transforms[(target, source)]
        ts = _V()
        ts.transform = _V()
        ts.transform.translation = _V()
        ts.transform.rotation = _V()
        ts.transform.translation.x, ts.transform.translation.y, ts.transform.translation.z = t
        ts.transform.rotation.x, ts.transform.rotation.y, ts.transform.rotation.z, ts.transform.rotation.w = q
        return ts


class TransformListener(object):
    def __init__(self, buf):
        self.buf = buf
