class Empty(object):
    pass
