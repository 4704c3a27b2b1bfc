"""ROS-free reader (and minimal writer) of rosbag format 2.0 files, so that BASELINE config 1 -- "rosbag replay" -- is
literally that: a recorded bag's messages, in recorded order, through the node's callbacks (replay.replay_bag), with no
ROS installation.  The reference reads its bags with `rosbag.Bag(...).read_messages()`
(auv_ekf_localization/rosbags/rosbag_handler.py:8-19); `Bag(path).read_messages(topics)` here yields the same
(topic, msg, t) triples for the message types the particle-filter node exchanges.

Format (ROS wiki "Bags/Format/2.0", restated): the line `#ROSBAG V2.0`, then records
`<header_len u32><header><data_len u32><data>`, little endian; a header is a sequence of `<field_len u32><name>=<value>`;
`op` (1 byte) tells the record: 0x03 bag header, 0x05 chunk (`compression` = none | bz2 | lz4, `size` = uncompressed
bytes; its data are connection and message records), 0x07 connection (`conn`, `topic`; data = a header of `type`,
`md5sum`, `message_definition`, ...), 0x02 message (`conn`, `time` = secs u32 + nsecs u32; data = the serialised
message), 0x04 index data, 0x06 chunk info.  The reader walks the file front to back and needs no index (an unindexed
or truncated bag -- a run that was killed -- is read as far as it goes).  Compression: none and bz2 (Python's bz2);
lz4 chunks raise BagError (no lz4 module in this environment).

Message (de)serialisation: ROS 1 wire format (little endian; string = u32 length + bytes; T[] = u32 count + items;
T[n] = items; time = u32 secs + u32 nsecs) for nav_msgs/Odometry, sensor_msgs/LaserScan, sensor_msgs/PointCloud2,
geometry_msgs/PoseArray, std_msgs/Bool, into the plain message shapes of msgs.py; other types come back as RawMessage.
The writer exists for tests and for exporting synthetic streams (write_bag): uncompressed or bz2 chunks, a connection
record per topic carrying the full message definition, chunk-info and index records so that ROS's own tools accept the
file."""
import bz2
import struct

import numpy as np

from . import msgs as _msgs

MAGIC = b'#ROSBAG V2.0\n'
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07


class BagError(ValueError):
    pass


class RawMessage(object):
    """A message of a type this module does not decode: its connection's type name and its bytes."""

    def __init__(self, type_name, data):
        self.type_name, self.data = type_name, data


# ------------------------------------------------------------------ wire format of the message types
class _R(object):
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, fmt):
        v = struct.unpack_from('<' + fmt, self.b, self.o)
        self.o += struct.calcsize('<' + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        n = self.take('I')
        s = self.b[self.o:self.o + n]
        if len(s) != n:
            raise BagError('truncated string')
        self.o += n
        return s.decode('utf-8', 'replace')

    def array(self, dtype, n):
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o)
        self.o += a.nbytes
        return a

    def header(self):
        h = _msgs.Header()
        h.seq = self.take('I')
        secs, nsecs = self.take('II')
        h.stamp = _msgs.Time(secs + 1e-9 * nsecs)
        h.frame_id = self.string()
        return h


def _w_string(s):
    b = s.encode('utf-8')
    return struct.pack('<I', len(b)) + b


def _w_time(t):
    secs = int(np.floor(t))
    nsecs = int(round((t - secs) * 1e9))
    if nsecs >= 1000000000:
        secs, nsecs = secs + 1, nsecs - 1000000000
    return struct.pack('<II', secs, nsecs)


def _w_header(h):
    return struct.pack('<I', int(getattr(h, 'seq', 0))) + _w_time(h.stamp.to_sec()) + _w_string(h.frame_id)


def _read_pose(r, pose):
    pose.position.x, pose.position.y, pose.position.z = r.take('ddd')
    q = r.take('dddd')
    pose.orientation = _msgs.Quaternion(*q)


def _dec_odometry(b):
    r = _R(b)
    m = _msgs.Odometry()
    m.header = r.header()
    m.child_frame_id = r.string()
    _read_pose(r, m.pose.pose)
    m.pose.covariance = list(r.array('<f8', 36))
    tw = m.twist.twist
    tw.linear.x, tw.linear.y, tw.linear.z, tw.angular.x, tw.angular.y, tw.angular.z = r.take('dddddd')
    m.twist.covariance = list(r.array('<f8', 36))
    return m


def _enc_odometry(m):
    p, tw = m.pose.pose, m.twist.twist
    return (_w_header(m.header) + _w_string(m.child_frame_id) +
            struct.pack('<7d', p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z,
                        p.orientation.w) + np.asarray(m.pose.covariance, '<f8').tobytes() +
            struct.pack('<6d', tw.linear.x, tw.linear.y, tw.linear.z, tw.angular.x, tw.angular.y, tw.angular.z) +
            np.asarray(m.twist.covariance, '<f8').tobytes())


def _dec_laserscan(b):
    r = _R(b)
    h = r.header()
    amin, amax, ainc, tinc, stime, rmin, rmax = r.take('fffffff')
    ranges = r.array('<f4', r.take('I'))
    inten = r.array('<f4', r.take('I'))
    m = _msgs.LaserScan(ranges, amin, ainc, rmax)
    m.ranges = np.array(ranges)          # (kept as an array: a ping has hundreds of beams)
    m.header, m.angle_max, m.time_increment, m.scan_time, m.range_min, m.intensities = h, amax, tinc, stime, rmin, np.array(inten)
    return m


def _enc_laserscan(m):
    ranges = np.asarray(m.ranges, '<f4')
    inten = np.asarray(getattr(m, 'intensities', ()), '<f4')
    amax = getattr(m, 'angle_max', m.angle_min + m.angle_increment * max(ranges.size - 1, 0))
    return (_w_header(m.header) + struct.pack('<7f', m.angle_min, amax, m.angle_increment, getattr(m, 'time_increment', 0.0),
                                               getattr(m, 'scan_time', 0.0), getattr(m, 'range_min', 0.0), m.range_max) +
            struct.pack('<I', ranges.size) + ranges.tobytes() + struct.pack('<I', inten.size) + inten.tobytes())


def _dec_bool(b):
    return _msgs.Bool(bool(b[0]))


def _enc_bool(m):
    return struct.pack('<B', 1 if m.data else 0)


def _dec_posearray(b):
    r = _R(b)
    m = _msgs.PoseArray()
    m.header = r.header()
    n = r.take('I')
    block = r.array('<f8', 7 * n).reshape(n, 7)
    m.data = np.array(block)
    for row in block:
        p = _msgs.Pose()
        p.position.x, p.position.y, p.position.z = (float(v) for v in row[:3])
        p.orientation = _msgs.Quaternion(*(float(v) for v in row[3:]))
        m.poses.append(p)
    return m


def _enc_posearray(m):
    rows = [[p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w]
            for p in m.poses]
    return _w_header(m.header) + struct.pack('<I', len(rows)) + np.asarray(rows, '<f8').reshape(-1).tobytes()


def _dec_pointcloud2(b):
    r = _R(b)
    m = _msgs.PointCloud2()
    m.header = r.header()
    m.height, m.width = r.take('II')
    m.fields = []
    for _ in range(r.take('I')):
        name = r.string()
        off, dt, cnt = r.take('IBI')
        m.fields.append(_msgs.PointField(name, off, dt, cnt))
    m.is_bigendian = bool(r.take('B'))
    m.point_step, m.row_step = r.take('II')
    n = r.take('I')
    m.data = bytes(r.b[r.o:r.o + n])
    r.o += n
    m.is_dense = bool(r.take('B'))
    return m


def _enc_pointcloud2(m):
    out = _w_header(m.header) + struct.pack('<II', m.height, m.width) + struct.pack('<I', len(m.fields))
    for f in m.fields:
        out += _w_string(f.name) + struct.pack('<IBI', f.offset, f.datatype, f.count)
    data = bytes(m.data)
    return (out + struct.pack('<B', 1 if getattr(m, 'is_bigendian', False) else 0) + struct.pack('<II', m.point_step, m.row_step) +
            struct.pack('<I', len(data)) + data + struct.pack('<B', 1 if getattr(m, 'is_dense', True) else 0))



# ---- message definitions, as rosbag stores them in a connection header (`message_definition`: the .msg text followed by
# the definitions of every message it depends on, separated by a line of 80 '=' and "MSG: <type>") -- rosbag's Python
# reader and rqt_bag generate their classes from this text, so a bag without it is unreadable for ROS's own tools.  The
# standard definitions of std_msgs / geometry_msgs / nav_msgs / sensor_msgs (ROS 1, comments shortened);
# tests/test_rosbag_io.py recomputes every md5sum of TYPES from them with genmsg's rule.
_SEP = '=' * 80 + '\n'
MSG_TEXT = {
    'std_msgs/Header': 'uint32 seq\ntime stamp\nstring frame_id\n',
    'std_msgs/Bool': 'bool data\n',
    'geometry_msgs/Point': 'float64 x\nfloat64 y\nfloat64 z\n',
    'geometry_msgs/Quaternion': 'float64 x\nfloat64 y\nfloat64 z\nfloat64 w\n',
    'geometry_msgs/Vector3': 'float64 x\nfloat64 y\nfloat64 z\n',
    'geometry_msgs/Pose': 'Point position\nQuaternion orientation\n',
    'geometry_msgs/Twist': 'Vector3  linear\nVector3  angular\n',
    'geometry_msgs/PoseWithCovariance': '# row-major 6x6 covariance of (x, y, z, rotation about X, Y, Z)\nPose pose\nfloat64[36] covariance\n',
    'geometry_msgs/TwistWithCovariance': '# row-major 6x6 covariance of (x, y, z, rotation about X, Y, Z)\nTwist twist\nfloat64[36] covariance\n',
    'geometry_msgs/PoseArray': '# An array of poses with a header for global reference.\nHeader header\nPose[] poses\n',
    'nav_msgs/Odometry': '# An estimate of a position and velocity in free space: pose in header.frame_id, twist in child_frame_id\n'
                         'Header header\nstring child_frame_id\ngeometry_msgs/PoseWithCovariance pose\n'
                         'geometry_msgs/TwistWithCovariance twist\n',
    'sensor_msgs/LaserScan': '# Single scan from a planar laser range-finder\nHeader header\nfloat32 angle_min\nfloat32 angle_max\n'
                             'float32 angle_increment\nfloat32 time_increment\nfloat32 scan_time\nfloat32 range_min\n'
                             'float32 range_max\nfloat32[] ranges\nfloat32[] intensities\n',
    'sensor_msgs/PointField': 'uint8 INT8    = 1\nuint8 UINT8   = 2\nuint8 INT16   = 3\nuint8 UINT16  = 4\nuint8 INT32   = 5\n'
                              'uint8 UINT32  = 6\nuint8 FLOAT32 = 7\nuint8 FLOAT64 = 8\nstring name\nuint32 offset\nuint8  datatype\n'
                              'uint32 count\n',
    'sensor_msgs/PointCloud2': '# N-dimensional points; layout described by `fields`\nHeader header\nuint32 height\nuint32 width\n'
                               'PointField[] fields\nbool    is_bigendian\nuint32  point_step\nuint32  row_step\nuint8[] data\n'
                               'bool is_dense\n',
}
# dependencies in the order gendeps --cat lists them (depth first, each once)
MSG_DEPS = {
    'std_msgs/Bool': [],
    'geometry_msgs/PoseArray': ['std_msgs/Header', 'geometry_msgs/Pose', 'geometry_msgs/Point', 'geometry_msgs/Quaternion'],
    'nav_msgs/Odometry': ['std_msgs/Header', 'geometry_msgs/PoseWithCovariance', 'geometry_msgs/Pose', 'geometry_msgs/Point',
                          'geometry_msgs/Quaternion', 'geometry_msgs/TwistWithCovariance', 'geometry_msgs/Twist', 'geometry_msgs/Vector3'],
    'sensor_msgs/LaserScan': ['std_msgs/Header'],
    'sensor_msgs/PointCloud2': ['std_msgs/Header', 'sensor_msgs/PointField'],
}


def message_definition(type_name):
    """The full `message_definition` text of a connection header for one of the types in TYPES."""
    return MSG_TEXT[type_name] + ''.join('\n' + _SEP + 'MSG: %s\n' % d + MSG_TEXT[d] for d in MSG_DEPS[type_name])


# type name -> (md5sum of the ROS message definition, decoder, encoder)
TYPES = {
    'nav_msgs/Odometry': ('cd5e73d190d741a2f92e81eda573aca7', _dec_odometry, _enc_odometry),
    'sensor_msgs/LaserScan': ('90c7ef2dc6895d81024acba2ac42f369', _dec_laserscan, _enc_laserscan),
    'sensor_msgs/PointCloud2': ('1158d486dd51d683ce2f1be655c3c181', _dec_pointcloud2, _enc_pointcloud2),
    'geometry_msgs/PoseArray': ('916c28c5764443f268b296bb671b9d97', _dec_posearray, _enc_posearray),
    'std_msgs/Bool': ('8b94c1b53db61fb6aed406028ad6332a', _dec_bool, _enc_bool),
}


# ------------------------------------------------------------------ records
def _parse_fields(b):
    b = bytes(b)   # (a record's header or a connection record's data: small; may arrive as a view of the file)
    out, o = {}, 0
    while o < len(b):
        if o + 4 > len(b):
            raise BagError('truncated record header')
        n, = struct.unpack_from('<I', b, o)
        o += 4
        f = b[o:o + n]
        if len(f) != n or b'=' not in f:
            raise BagError('bad header field')
        o += n
        k, v = f.split(b'=', 1)
        out[k.decode('ascii')] = v
    return out


def _records(buf, start=0):
    """(fields, data) of every complete record in buf; stops quietly at a truncated tail."""
    o = start
    while o + 4 <= len(buf):
        hl, = struct.unpack_from('<I', buf, o)
        if o + 4 + hl + 4 > len(buf):
            return
        head = _parse_fields(bytes(buf[o + 4:o + 4 + hl]))
        dl, = struct.unpack_from('<I', buf, o + 4 + hl)
        d0 = o + 8 + hl
        if d0 + dl > len(buf):
            return
        yield head, buf[d0:d0 + dl]   # (buf a memoryview: a view, not a copy)
        o = d0 + dl


def _field(name, value):
    f = name.encode('ascii') + b'=' + value
    return struct.pack('<I', len(f)) + f


def _record(fields, data):
    h = b''.join(_field(k, v) for k, v in fields)
    return struct.pack('<I', len(h)) + h + struct.pack('<I', len(data)) + data


class Bag(object):
    """`for topic, msg, t in Bag(path).read_messages(topics=None)`: the messages ordered by their record (receive) time
    like rosbag.Bag.read_messages -- which merges the connections' index entries by time, what the reference's
    rosbag_handler.py:8-19 iterates --, messages of equal time in file order; `by_time=False` yields plain file order (a
    chunk is a contiguous run; rosbag writes in receive order, so the two differ only for bags merged or re-indexed by
    tools).  t = the record's receive time in seconds.  `connections`: topic -> type name, filled as the file is read
    (connection records precede their messages)."""

    def __init__(self, path):
        with open(path, 'rb') as f:
            self.buf = f.read()
        if not self.buf.startswith(MAGIC):
            raise BagError('%s: not a rosbag 2.0 file' % path)
        self.connections = {}

    def _conn(self, head, data, conns):
        cid, = struct.unpack('<I', head['conn'])
        ch = _parse_fields(data)
        topic = head['topic'].decode('utf-8')
        conns[cid] = (topic, ch.get('type', b'').decode('ascii'))
        self.connections[topic] = conns[cid][1]

    def read_messages(self, topics=None, raw=False, by_time=True):
        conns = {}
        want = None if topics is None else set(topics)

        def inner(records):
            for head, data in records:
                op = head['op'][0]
                if op == OP_CONNECTION:
                    self._conn(head, data, conns)
                elif op == OP_MSG:
                    cid, = struct.unpack('<I', head['conn'])
                    if cid not in conns:
                        continue
                    topic, typ = conns[cid]
                    if want is not None and topic not in want:
                        continue
                    secs, nsecs = struct.unpack('<II', head['time'])
                    yield topic, typ, data, secs + 1e-9 * nsecs
                elif op == OP_CHUNK:
                    comp = head.get('compression', b'none')
                    if comp == b'none':
                        body = data
                    elif comp == b'bz2':
                        body = memoryview(bz2.decompress(data))
                    else:
                        raise BagError('chunk compression %r is not supported (none and bz2 are)' % comp.decode('ascii', 'replace'))
                    for item in inner(_records(body)):
                        yield item

        def decode(topic, typ, data, t):
            dec = TYPES.get(typ)
            return topic, (dec[1](bytes(data)) if dec and not raw else RawMessage(typ, bytes(data))), t

        view = memoryview(self.buf)
        if not by_time:
            for item in inner(_records(view, len(MAGIC))):
                yield decode(*item)
            return
        # ordered by time WITHOUT decoding first: the first pass keeps (topic, type, a VIEW of the record's
        # bytes, time) -- a few dozen bytes per message on top of the file itself (a bz2 chunk's decompressed body is
        # kept while its views are alive) --, messages are decoded one at a time as they are yielded
        items = list(inner(_records(view, len(MAGIC))))
        for k in sorted(range(len(items)), key=lambda j: (items[j][3], j)):   # (stable: equal stamps keep the file's order)
            yield decode(*items[k])


def write_bag(path, messages, compression='none', chunk_messages=64):
    """messages: iterable of (topic, type name, msg, t) in the order they are to be recorded.  Writes a complete
    rosbag 2.0 file: bag header (padded to 4096 bytes), chunks of `chunk_messages` messages with their index records,
    then the connection and chunk-info records the bag header's index_pos points at."""
    messages = list(messages)
    chunk_messages = max(int(chunk_messages), 1)
    conn_id, conn_recs = {}, {}
    for topic, typ, _, _ in messages:
        if topic not in conn_id:
            if typ not in TYPES:
                raise BagError('write_bag: no encoder for %s' % typ)
            cid = len(conn_id)
            conn_id[topic] = cid
            ch = _field('topic', topic.encode()) + _field('type', typ.encode()) + _field('md5sum', TYPES[typ][0].encode()) + \
                _field('message_definition', message_definition(typ).encode())
            conn_recs[cid] = _record([('op', bytes([OP_CONNECTION])), ('conn', struct.pack('<I', cid)), ('topic', topic.encode())], ch)
    body = bytearray()
    chunk_infos = []
    for c0 in range(0, len(messages), chunk_messages):
        part = messages[c0:c0 + chunk_messages]
        chunk, seen, index = bytearray(), set(), {}
        for topic, typ, msg, t in part:
            cid = conn_id[topic]
            if cid not in seen:   # a chunk carries the connection records of the messages in it
                seen.add(cid)
                chunk += conn_recs[cid]
            index.setdefault(cid, []).append((t, len(chunk)))
            chunk += _record([('op', bytes([OP_MSG])), ('conn', struct.pack('<I', cid)), ('time', _w_time(t))], TYPES[typ][2](msg))
        raw = bytes(chunk)
        data = bz2.compress(raw) if compression == 'bz2' else raw
        pos = len(MAGIC) + 4096 + len(body)
        body += _record([('op', bytes([OP_CHUNK])), ('compression', compression.encode()), ('size', struct.pack('<I', len(raw)))], data)
        for cid, entries in index.items():
            idx = b''.join(_w_time(t) + struct.pack('<I', off) for t, off in entries)
            body += _record([('op', bytes([OP_INDEX])), ('ver', struct.pack('<I', 1)), ('conn', struct.pack('<I', cid)),
                             ('count', struct.pack('<I', len(entries)))], idx)
        ts = [t for _, _, _, t in part]
        counts = b''.join(struct.pack('<II', cid, len(e)) for cid, e in index.items())
        chunk_infos.append(_record([('op', bytes([OP_CHUNK_INFO])), ('ver', struct.pack('<I', 1)), ('chunk_pos', struct.pack('<Q', pos)),
                                    ('start_time', _w_time(min(ts))), ('end_time', _w_time(max(ts))),
                                    ('count', struct.pack('<I', len(index)))], counts))
    index_pos = len(MAGIC) + 4096 + len(body)
    head_fields = [('op', bytes([OP_BAG_HEADER])), ('index_pos', struct.pack('<Q', index_pos)),
                   ('conn_count', struct.pack('<I', len(conn_id))), ('chunk_count', struct.pack('<I', len(chunk_infos)))]
    h = b''.join(_field(k, v) for k, v in head_fields)
    pad = 4096 - (4 + len(h) + 4)
    with open(path, 'wb') as f:
        f.write(MAGIC)
        f.write(struct.pack('<I', len(h)) + h + struct.pack('<I', pad) + b' ' * pad)
        f.write(bytes(body))
        for cid in sorted(conn_recs):
            f.write(conn_recs[cid])
        for rec in chunk_infos:
            f.write(rec)
