class Empty(object):
    pass


class SetBool(object):
    pass


class SetBoolRequest(object):
    pass
