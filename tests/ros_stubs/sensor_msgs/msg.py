"""sensor_msgs stand-ins (TEST INFRASTRUCTURE ONLY): the byte layout of PointCloud2 is the real one."""
from geometry_msgs.msg import Header


class LaserScan(object):
    def __init__(self):
        self.header = Header()
        self.angle_min = self.angle_max = self.angle_increment = 0.0
        self.time_increment = self.scan_time = 0.0
        self.range_min, self.range_max = 0.0, 100.0
        self.ranges, self.intensities = [], []


class PointField(object):
    INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 = 1, 2, 3, 4, 5, 6, 7, 8

    def __init__(self, name='', offset=0, datatype=7, count=1):
        self.name, self.offset, self.datatype, self.count = name, offset, datatype, count


class PointCloud2(object):
    def __init__(self):
        self.header = Header()
        self.height, self.width = 1, 0
        self.fields = []
        self.is_bigendian = False
        self.point_step = self.row_step = 0
        self.data = b''
        self.is_dense = True
