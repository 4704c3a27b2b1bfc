"""rospy stand-in for tests/test_ros_node_stub.py (TEST INFRASTRUCTURE ONLY): a parameter server, publishers that
record, subscribers / timers that register themselves so the test can deliver messages, a settable clock."""
_params = {}
publishers, subscribers, timers, log = {}, {}, [], []


def reset(params=None):
    _params.clear()
    _params.update(params or {})
    publishers.clear()
    subscribers.clear()
    del timers[:]
    del log[:]
    Time._now = 0.0


class ROSInterruptException(Exception):
    pass


class Duration(object):
    def __init__(self, secs=0.0):
        self.secs = float(secs)

    def to_sec(self):
        return self.secs


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


def _logger(level):
    def f(msg, *a):
        log.append((level, msg % a if a else msg))
    return f


loginfo, logwarn, logerr, logdebug = _logger('info'), _logger('warn'), _logger('err'), _logger('debug')


class Publisher(object):
    def __init__(self, topic, typ, queue_size=1):
        self.topic, self.typ, self.sent = topic, typ, []
        publishers[topic] = self

    def publish(self, msg):
        self.sent.append(msg)


class Subscriber(object):
    def __init__(self, topic, typ, cb, queue_size=1):
        self.topic, self.typ, self.cb = topic, typ, cb
        subscribers[topic] = self


class Timer(object):
    def __init__(self, period, cb):
        self.period, self.cb = period, cb
        timers.append(self)


def init_node(name, **k):
    log.append(('init_node', name))


def spin():
    return
