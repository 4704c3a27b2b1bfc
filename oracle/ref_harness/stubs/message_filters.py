"""message_filters stand-in (TEST INFRASTRUCTURE ONLY): the harness calls the callbacks itself."""


class Subscriber(object):
    def __init__(self, topic, typ):
        self.topic = topic


class ApproximateTimeSynchronizer(object):
    def __init__(self, subs, queue_size, slop=0.1, allow_headerless=False):
        self.cb = None

    def registerCallback(self, cb):
        self.cb = cb
