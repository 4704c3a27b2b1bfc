"""rosbag format 2.0 without ROS (smarc_navigation_amd/rosbag_io.py) and BASELINE config 1 as a literal rosbag replay
(replay.replay_bag): a tiny synthetic bag written by the test -- uncompressed and bz2 chunks -- is read back message
for message, a truncated bag is read as far as it goes, and its messages drive the node's callbacks in recorded order
(CPU: a recording engine pins the plumbing; GPU: the same bag through the real engine equals the .npz replay).
The reference reads its bags with rosbag.Bag(...).read_messages() (auv_ekf_localization/rosbags/rosbag_handler.py:8-19)."""
import struct

import numpy as np
import pytest

from smarc_navigation_amd import msgs, replay, rosbag_io, synth


def _stream(n=60):
    s = synth.odom_stream(n)
    st = dict(stamp=s['stamp'], v=s['v'], wz=s['wz'], q=s['q'], z=s['z'], t0=s['t0'])
    st['gps_idx'] = np.array([20, 45])
    st['gps_xy_utm'] = np.array([[1000.5, 2000.25], [1001.5, 2000.75]])
    ang = synth.beam_angles(16)
    st['mbes_idx'] = np.array([10, 30, 50])
    st['mbes_angles'] = ang
    st['mbes_ranges'] = (20.0 / np.cos(ang))[None, :].repeat(3, axis=0).astype(np.float32) + np.arange(3, dtype=np.float32)[:, None]
    st['mbes_range_max'] = 80.0
    return st


@pytest.mark.parametrize('compression', ['none', 'bz2'])
def test_bag_round_trip_message_for_message(tmp_path, compression):
    st = _stream()
    path = str(tmp_path / 'run.bag')
    n_msgs = replay.stream_to_bag(path, st, compression=compression)
    raw = open(path, 'rb').read()
    assert raw.startswith(b'#ROSBAG V2.0\n')
    # the bag header record is padded to 4096 bytes and points at the index section (connection + chunk-info records)
    hl, = struct.unpack_from('<I', raw, 13)
    dl, = struct.unpack_from('<I', raw, 13 + 4 + hl)
    assert 4 + hl + 4 + dl == 4096
    bag = rosbag_io.Bag(path)
    got = list(bag.read_messages())
    assert len(got) == n_msgs == 60 + 2 + 1 + 3
    assert bag.connections == {'/sam/dr/odom': 'nav_msgs/Odometry', '/sam/dr/gps': 'nav_msgs/Odometry', '/dive': 'std_msgs/Bool',
                               '/sam/mbes_scan': 'sensor_msgs/LaserScan'}
    assert [t for _, _, t in got] == sorted(t for _, _, t in got)          # recorded order = time order here
    odoms = [m for topic, m, _ in got if topic == '/sam/dr/odom']
    for k in (0, 17, 59):
        ref = msgs.odometry_from_stream(st, k)
        m = odoms[k]
        assert abs(m.header.stamp.to_sec() - st['stamp'][k]) < 1e-9 and m.header.frame_id == 'sam/odom'
        assert (m.twist.twist.linear.x, m.twist.twist.angular.z, m.pose.pose.position.z) == (
            ref.twist.twist.linear.x, ref.twist.twist.angular.z, ref.pose.pose.position.z)
        assert (m.pose.pose.orientation.x, m.pose.pose.orientation.w) == (ref.pose.pose.orientation.x, ref.pose.pose.orientation.w)
    scans = [m for topic, m, _ in got if topic == '/sam/mbes_scan']
    assert len(scans) == 3 and np.array_equal(scans[1].ranges, st['mbes_ranges'][1]) and scans[1].range_max == 80.0
    assert abs(scans[0].angle_min - float(st['mbes_angles'][0])) < 1e-7
    fixes = [m for topic, m, _ in got if topic == '/sam/dr/gps']
    assert (fixes[1].pose.pose.position.x, fixes[1].pose.pose.position.y) == (1001.5, 2000.75)
    assert [m.data for topic, m, _ in got if topic == '/dive'] == [False]
    # a topic filter, and raw access for types the module does not decode
    assert len(list(bag.read_messages(topics=['/sam/mbes_scan']))) == 3
    r = next(iter(bag.read_messages(topics=['/dive'], raw=True)))[1]
    assert isinstance(r, rosbag_io.RawMessage) and r.type_name == 'std_msgs/Bool' and r.data == b'\x00'


def test_posearray_and_pointcloud2_round_trip(tmp_path):
    pa = msgs.PoseArray()
    pa.header.frame_id, pa.header.stamp = 'sam/base_link', msgs.Time(12.5)
    for row in ((1.0, 2.0, -15.0), (0.5, -3.0, -14.0)):
        p = msgs.Pose()
        p.position.x, p.position.y, p.position.z = row
        pa.poses.append(p)
    xyz = np.random.RandomState(1).randn(32, 3)
    pc = msgs.pointcloud2_from_xyz(xyz, 'sam/base_link', msgs.Time(12.5))
    path = str(tmp_path / 'det.bag')
    rosbag_io.write_bag(path, [('/sam/mbes_detections', 'geometry_msgs/PoseArray', pa, 12.5),
                               ('/sam/mbes_cloud', 'sensor_msgs/PointCloud2', pc, 12.5)])
    got = {topic: m for topic, m, _ in rosbag_io.Bag(path).read_messages()}
    d = got['/sam/mbes_detections']
    assert [(p.position.x, p.position.y, p.position.z) for p in d.poses] == [(1.0, 2.0, -15.0), (0.5, -3.0, -14.0)]
    assert d.poses[0].orientation.w == 1.0 and d.header.frame_id == 'sam/base_link'
    np.testing.assert_array_equal(msgs.pointcloud2_xyz(got['/sam/mbes_cloud']), xyz.astype(np.float32).astype(np.float64))


def test_truncated_and_foreign_files(tmp_path):
    st = _stream(30)
    st.pop('gps_idx'), st.pop('mbes_idx')
    path = str(tmp_path / 'run.bag')
    replay.stream_to_bag(path, st)
    raw = open(path, 'rb').read()
    cut = str(tmp_path / 'cut.bag')
    open(cut, 'wb').write(raw[:13 + 4096 + 40])                # a recording that was killed inside its first chunk
    assert list(rosbag_io.Bag(cut).read_messages()) == []       # ... is read as far as it goes: no complete record
    with pytest.raises(rosbag_io.BagError):
        rosbag_io.Bag(str(tmp_path / 'x.bag')) if open(str(tmp_path / 'x.bag'), 'wb').write(b'not a bag') else None
    # an lz4 chunk is refused, not misread
    bad = raw.replace(b'compression=none', b'compression=lz44')
    open(cut, 'wb').write(bad)
    with pytest.raises(rosbag_io.BagError):
        list(rosbag_io.Bag(cut).read_messages())


class _FakeEngine(object):
    calls = []

    def __init__(self, n, **kw):
        self.n = n
        _FakeEngine.calls.append(('create', n, kw))

    def __getattr__(self, name):
        def f(*a, **k):
            _FakeEngine.calls.append((name, a, k))
            if name == 'mean_cov':
                return np.arange(6.0), 0.25, np.arange(9.0)
            if name == 'poses':
                return np.zeros((self.n, 7))
            return None
        return f


def test_config1_rosbag_replay_plumbing_on_cpu(tmp_path, monkeypatch):
    """128 particles, a recorded bag, no GPU: which ABI call every recorded message becomes, in recorded order."""
    from smarc_navigation_amd import engine as eng
    monkeypatch.setattr(eng, 'Engine', _FakeEngine)
    _FakeEngine.calls = []
    st = _stream()
    path = str(tmp_path / 'run.bag')
    replay.stream_to_bag(path, st, compression='bz2')
    z = synth.bathymetry_grid(64, 64, 1.0, (-32.0, -32.0), seed=1)
    res = replay.replay_bag(path, dict(particle_count=128), grid=dict(z=z, origin=(-32.0, -32.0), res=1.0),
                            utm2map=np.array([[1.0, 0, 0, -1000.0], [0, 1.0, 0, -2000.0], [0, 0, 1, 0], [0, 0, 0, 1]]))
    assert res['counts'] == {'/sam/dr/odom': 60, '/sam/dr/gps': 2, '/dive': 1, '/sam/mbes_scan': 3}
    names = [c[0] for c in _FakeEngine.calls]
    assert _FakeEngine.calls[0][:2] == ('create', 128)
    assert names.count('predict') == 60 and names.count('update_mbes') == 3 and names.count('update_gps') == 2
    # dt of every predict from the recorded stamps (auv_pf.py:204-205)
    dts = [c[1][4] for c in _FakeEngine.calls if c[0] == 'predict']
    np.testing.assert_allclose(dts, 0.02, atol=1e-6)
    # recorded order: the first ping comes after the 11th odometry sample, the first fix (utm -> map once) after the 21st
    first_ping = names.index('update_mbes')
    assert names[:first_ping].count('predict') == 11
    first_fix = names.index('update_gps')
    assert names[:first_fix].count('predict') == 21 and _FakeEngine.calls[first_fix][1] == (0.5, 0.25)
    np.testing.assert_array_equal(_FakeEngine.calls[first_ping][1][0], st['mbes_ranges'][0])
    # the 10 Hz timer in bag time: 60 samples at 50 Hz = 1.2 s -> 11 ticks + the final one
    assert len(res['pf_xyz']) == 12 and np.all(np.diff(res['pf_stamp'][:-1]) > 0.0999)


@pytest.mark.gpu
def test_rosbag_replay_equals_the_stream_replay_on_the_gpu(tmp_path):
    """The same recorded inputs as a bag and as the .npz stream: the node publishes the same final pose."""
    st = _stream(100)
    st['gps_idx'], st['gps_xy_utm'] = np.array([40, 80]), np.array([[0.8, 0.0], [1.6, 0.05]])
    st['mbes_idx'] = np.array([10, 30, 50])
    origin = (-32.0, -32.0)
    z = synth.bathymetry_grid(64, 64, 1.0, origin, seed=1)
    grid = dict(z=z, origin=origin, res=1.0)
    params = dict(particle_count=128, seed=4, init_covariance='[0.5, 0.5, 0.0, 0.0, 0.0, 0.01]',
                  motion_covariance='[0.001, 0.001, 0.0, 0.0, 0.0, 0.00001]', measurement_std=1.0)
    path = str(tmp_path / 'run.bag')
    replay.stream_to_bag(path, st)
    a = replay.replay_bag(path, params, grid=grid, t0=float(st['t0']))
    b = replay.replay(st, params, grid=grid, publish_every=1000)
    assert a['counts']['/sam/dr/odom'] == 100
    np.testing.assert_array_equal(a['pf_xyz'][-1], b['pf_xyz'][-1])


def _genmsg_md5(rosbag_io, type_name, cache):
    """genmsg.gentools.compute_md5: comments and blank lines dropped, constants first, a field of a message type written
    as `<md5 of that type> <name>` (arrays of messages lose their brackets), builtin fields as `<type> <name>`."""
    import hashlib
    import re
    if type_name in cache:
        return cache[type_name]
    builtin = {'bool', 'int8', 'uint8', 'int16', 'uint16', 'int32', 'uint32', 'int64', 'uint64', 'float32', 'float64', 'string',
               'time', 'duration', 'char', 'byte'}
    pkg = type_name.split('/')[0]
    consts, fields = [], []
    for line in rosbag_io.MSG_TEXT[type_name].split('\n'):
        line = line.split('#')[0].strip()
        if not line:
            continue
        if '=' in line:
            t, rest = line.split(None, 1)
            name, val = [x.strip() for x in rest.split('=', 1)]
            consts.append('%s %s=%s' % (t, name, val))
            continue
        t, name = line.split()
        base = re.sub(r'\[.*\]$', '', t)
        if base in builtin:
            fields.append('%s %s' % (t, name))
        else:
            full = 'std_msgs/Header' if base == 'Header' else (base if '/' in base else pkg + '/' + base)
            fields.append('%s %s' % (_genmsg_md5(rosbag_io, full, cache), name))
    cache[type_name] = hashlib.md5('\n'.join(consts + fields).encode()).hexdigest()
    return cache[type_name]


def test_embedded_message_definitions_have_the_md5sums_the_bag_declares():
    """write_bag stores the full message definition in every connection header (rosbag's Python reader and rqt_bag
    generate their classes from it): the md5sum of each, recomputed by genmsg's rule from the embedded text, is the one in
    TYPES -- the published checksums of the ROS 1 standard messages."""
    cache = {}
    assert _genmsg_md5(rosbag_io, 'std_msgs/Header', cache) == '2176decaecbce78abc3b96ef049fabed'
    for typ, (md5, _, _) in rosbag_io.TYPES.items():
        assert _genmsg_md5(rosbag_io, typ, cache) == md5, typ
        text = rosbag_io.message_definition(typ)
        for dep in rosbag_io.MSG_DEPS[typ]:
            assert ('\n' + '=' * 80 + '\nMSG: %s\n' % dep) in text


def test_connection_headers_carry_the_definition_and_messages_come_back_in_time_order(tmp_path):
    od = [msgs.Odometry() for _ in range(3)]
    for k, o in enumerate(od):
        o.header.stamp = msgs.Time(10.0 + k)
    b = msgs.Bool(True)
    # recorded out of time order (a merged bag): odom 12, bool 10.5, odom 10, odom 11
    path = str(tmp_path / 'o.bag')
    rosbag_io.write_bag(path, [('/o', 'nav_msgs/Odometry', od[2], 12.0), ('/d', 'std_msgs/Bool', b, 10.5),
                         ('/o', 'nav_msgs/Odometry', od[0], 10.0), ('/o', 'nav_msgs/Odometry', od[1], 11.0)], chunk_messages=0)
    raw = open(path, 'rb').read()
    assert b'MSG: geometry_msgs/PoseWithCovariance' in raw and b'definition omitted' not in raw
    bag = rosbag_io.Bag(path)
    assert [round(t, 3) for _, _, t in bag.read_messages()] == [10.0, 10.5, 11.0, 12.0]            # rosbag's order
    assert [round(t, 3) for _, _, t in bag.read_messages(by_time=False)] == [12.0, 10.5, 10.0, 11.0]   # the file's
