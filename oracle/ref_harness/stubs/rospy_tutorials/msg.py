class Floats(object):
    pass
