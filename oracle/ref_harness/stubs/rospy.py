"""Minimal rospy stand-in so the reference's auv_particle_filter modules import in a
ROS-less container.  TEST INFRASTRUCTURE ONLY (used by oracle/ref_harness/gen_golden.py);
nothing here is shipped or imported by the product path."""
import time as _time

_params = {}


class ROSInterruptException(Exception):
    pass


class Duration(object):
    def __init__(self, secs=0.0):
        self.secs = float(secs)


class Time(object):
    _now = 0.0

    def __init__(self, secs=0.0):
        self.secs = float(secs)

    def to_sec(self):
        return self.secs

    @staticmethod
    def now():
        return Time(Time._now)


def get_param(name, default=None):
    key = name.lstrip('~')
    if key in _params:
        return _params[key]
    if default is None:
        raise KeyError(name)
    return default


def set_params(d):
    _params.clear()
    _params.update(d)


def loginfo(*a, **k):
    pass


logwarn = logerr = logdebug = loginfo


class Publisher(object):
    def __init__(self, topic, typ, queue_size=1):
        self.topic = topic
        self.sent = []

    def publish(self, msg):
        self.sent.append(msg)


class Subscriber(object):
    def __init__(self, topic, typ, cb, queue_size=1):
        self.topic, self.cb = topic, cb
        self.registered = True

    def unregister(self):
        self.registered = False


class Timer(object):
    def __init__(self, period, cb):
        self.period, self.cb = period, cb


def init_node(*a, **k):
    pass


def spin():
    pass
