"""Bathymetry map builder (include/mcl_map.h).  CPU: the oracle's definition on analytic cases.
GPU: kernels vs oracle (point cloud to rounding, accumulators exact), order independence (bitwise), and
the survey -> map -> localise round trip through the ray-cast."""
import numpy as np
import pytest

from smarc_navigation_amd import synth


def _survey(n_pings, B, seed=0, span=40.0):
    rs = np.random.RandomState(seed)
    t = np.linspace(0.0, 1.0, n_pings)
    poses = np.zeros((n_pings, 6))
    poses[:, 0] = -span / 2 + span * t
    poses[:, 1] = 6.0 * np.sin(6.0 * t)
    poses[:, 2] = -2.0 + 0.2 * np.sin(9.0 * t)
    poses[:, 3] = 0.03 * rs.randn(n_pings)
    poses[:, 4] = 0.03 * rs.randn(n_pings)
    poses[:, 5] = 0.3 * np.cos(5.0 * t)
    return poses, synth.beam_angles(B)


def test_oracle_flat_seabed_and_fill():
    from oracle import oracle as orc
    poses = np.zeros((1, 6))
    poses[0, 2] = -1.0
    ba = np.array([-np.pi / 4, 0.0, np.pi / 4], dtype=np.float32)
    depth = 10.0
    ranges = (depth / np.cos(ba)).astype(np.float32)[None, :]
    b = orc.GridMapBuilder(32, 32, (-16.0, -16.0), 1.0)
    pts = b.add_pings(poses, ranges, ba, 100.0, want_points=True)
    # beam b looks along (0, sin a, -cos a): across-track y = depth tan a, z = -1 - depth
    np.testing.assert_allclose(pts[0, :, 0], 0.0, atol=1e-12)
    np.testing.assert_allclose(pts[0, :, 1], depth * np.tan(ba.astype(np.float64)), rtol=1e-6)
    np.testing.assert_allclose(pts[0, :, 2], -11.0, rtol=1e-6)
    z, empty = b.finalize(0)
    assert empty == 32 * 32 - 3 and np.isfinite(z[16, 16]) and abs(z[16, 16] + 11.0) < 1e-5
    assert np.isfinite(z[16, 26]) and np.isfinite(z[16, 6])
    # invalid ranges are skipped
    b2 = orc.GridMapBuilder(32, 32, (-16.0, -16.0), 1.0)
    p2 = b2.add_pings(poses, np.array([[np.nan, -1.0, 200.0]], np.float32), ba, 100.0, want_points=True)
    assert np.isnan(p2).all() and b2.cnt.sum() == 0
    # hole filling spreads outward one ring per sweep and keeps measured nodes
    z1, e1 = b.finalize(1)
    assert e1 < empty and z1[16, 16] == z[16, 16] and np.isfinite(z1[15, 15])
    zf, ef = b.finalize(64)
    assert ef == 0 and np.all(np.abs(zf + 11.0) < 1e-4)


@pytest.mark.gpu
def test_gpu_matches_oracle_and_is_order_free():
    from oracle import oracle as orc
    from smarc_navigation_amd import gridmap
    rs = np.random.RandomState(1)
    n_pings, B = 400, 128
    poses, ba = _survey(n_pings, B)
    ranges = (18.0 + rs.rand(n_pings, B) * 4.0).astype(np.float32)
    ranges[5, 7] = np.nan
    ranges[9, :4] = -1.0
    ranges[11, 3] = 500.0
    m2o = synth.rigid_matrix(0.5, -0.25, 0.0, 0.0, 0.0, 0.1)
    off = [0.2, 0.0, -0.1, 0.01, -0.02, 0.03]
    nx, ny, origin, res = 96, 128, (-30.0, -50.0), 0.75
    g = gridmap.GridMapBuilder(nx, ny, origin, res)
    pts = g.add_pings(poses, ranges, ba, 60.0, m2o=m2o, sensor_offset=off, want_points=True)
    z, empty, cnt = g.finalize(0, want_counts=True)
    o = orc.GridMapBuilder(nx, ny, origin, res)
    opts = o.add_pings(poses, ranges, ba, 60.0, m2o=m2o, sensor_off=off, want_points=True)
    oz, oempty = o.finalize(0)
    assert np.array_equal(np.isnan(pts), np.isnan(opts))
    ok = ~np.isnan(opts)
    np.testing.assert_allclose(pts[ok], opts[ok], rtol=0, atol=1e-11)
    # accumulators: integer arithmetic, so identical unless a point sits within an ulp of a node boundary
    ocnt = o.cnt.reshape(nx, ny)
    assert (cnt != ocnt).sum() <= 2
    same = cnt == ocnt
    assert empty == oempty or abs(empty - oempty) <= 2
    both = same & (ocnt > 0)
    np.testing.assert_allclose(z[both], oz[both], rtol=0, atol=2e-6)
    assert both.sum() > 1500
    # hole filling follows the same definition
    zf, ef = g.finalize(3)
    ozf, oef = o.finalize(3)
    m = np.isfinite(ozf) & np.isfinite(zf)
    assert abs(ef - oef) <= 4 and m.sum() > both.sum()
    np.testing.assert_allclose(zf[m], ozf[m], rtol=0, atol=1e-4)
    # order independence: the same pings in two batches, reversed -> bitwise the same map
    g.clear()
    g.add_pings(poses[200:][::-1], ranges[200:][::-1], ba, 60.0, m2o=m2o, sensor_offset=off)
    g.add_pings(poses[:200][::-1], ranges[:200][::-1], ba, 60.0, m2o=m2o, sensor_offset=off)
    z2, empty2, cnt2 = g.finalize(0, want_counts=True)
    assert np.array_equal(cnt2, cnt) and np.array_equal(z2.view(np.uint32), z.view(np.uint32)) and empty2 == empty
    g.close()


@pytest.mark.gpu
def test_survey_map_localise_round_trip():
    """Ray-cast pings from a truth grid, rebuild the grid from them, localise in the rebuilt map."""
    from smarc_navigation_amd import engine as eng, gridmap
    origin, res = (-64.0, -64.0), 1.0
    truth_map = synth.bathymetry_grid(128, 128, res, origin, seed=3)
    B = 256
    ba = synth.beam_angles(B)
    # lawn-mower survey: 8 lines along x, 3 m apart in y... dense enough that every node in the box is hit
    lines = []
    for k, y in enumerate(np.arange(-30.0, 30.1, 4.0)):
        xs = np.arange(-30.0, 30.1, 0.5)
        if k % 2:
            xs = xs[::-1]
        for x in xs:
            lines.append([x, y, -2.0, 0.0, 0.0, 0.0 if k % 2 == 0 else np.pi])
    poses = np.array(lines)
    e = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    e.set_map_grid(truth_map, origin, res)
    ranges = np.zeros((len(poses), B), np.float32)
    for i, p in enumerate(poses):
        e.set_particles(p[:, None].copy())
        ranges[i] = e.mbes_expected(0, 1, ba, 80.0)[0]
    e.close()
    g = gridmap.GridMapBuilder(128, 128, origin, res)
    g.add_pings(poses, ranges, ba, 79.9)
    z, empty, cnt = g.finalize(0, want_counts=True)
    box = (slice(64 - 25, 64 + 26), slice(64 - 25, 64 + 26))
    assert np.isfinite(z[box]).all() and cnt[box].min() >= 1
    # a node's mean is taken over hits within half a cell of it: equal to the truth up to the local slope
    err = z[box] - truth_map[box]
    assert np.sqrt(np.mean(err ** 2)) < 0.08 and np.abs(err).max() < 0.4
    zf, ef = g.finalize(200)
    assert ef == 0
    g.close()
    # localise in the rebuilt map: particles displaced by 1 m converge onto the pose the pings were taken from
    f = eng.Engine(20000, seed=3, init_cov=[1.0, 1.0, 0.0, 0.0, 0.0, 0.0], resample_cov=[0.003, 0.003, 0, 0, 0, 0])
    f.set_map_grid(zf, origin, res)
    f.init_particles()
    st = f.get_particles()
    st[0] += 5.0
    st[1] += 2.0
    st[2] = -2.0
    f.set_particles(st)
    k = int(np.argmin(np.abs(poses[:, 0] - 5.0) + np.abs(poses[:, 1] - 2.0)))
    for _ in range(5):
        f.update_mbes(ranges[k], ba, 0.2, 80.0)
        f.resample()
    mean, _, _ = f.mean_cov()
    assert abs(mean[0] - poses[k, 0]) < 0.25 and abs(mean[1] - poses[k, 1]) < 0.25
    f.close()


def test_fuse_swath_follows_pclfuser_and_pcd_round_trip(tmp_path):
    """Host restatement of mbes_receptor.cpp:64-107: middle ping's pose = submap frame, every ping moved by
    T_submap<-map * T_map<-base_t, cloud stored with the submap pose as VIEWPOINT.  No GPU."""
    from smarc_navigation_amd import gridmap, synth
    rs = np.random.RandomState(3)
    n = 5
    poses = np.column_stack([np.linspace(0, 4, n), 0.3 * rs.randn(n), -2 + 0.1 * rs.randn(n), 0.02 * rs.randn(n),
                             0.03 * rs.randn(n), 0.2 + 0.05 * rs.randn(n)])
    pts_map = [rs.randn(7, 3) * [1, 10, 0.2] + [p[0], p[1], -20.0] for p in poses]     # truth in the map
    pts_base = [(pm - synth.rigid_matrix(*p)[:3, 3]).dot(synth.rigid_matrix(*p)[:3, :3]) for pm, p in zip(pts_map, poses)]
    sm = gridmap.fuse_swath(pts_base, poses, index=3)
    assert sm['frame_id'] == 'submap_3_frame' and sm['points'].shape == (35, 3)
    T = sm['T_map_submap']
    np.testing.assert_allclose(T, synth.rigid_matrix(*poses[2]), atol=1e-12)            # (5 - 1) / 2 = ping 2
    back = sm['points'].astype(np.float64).dot(T[:3, :3].T) + T[:3, 3]
    np.testing.assert_allclose(back, np.concatenate(pts_map), atol=2e-5)                 # float32 cloud
    # quaternion of the stored orientation reproduces the rotation
    x, y, z, w = sm['quat']
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    np.testing.assert_allclose(R, T[:3, :3], atol=1e-12)
    path = str(tmp_path / 'submap_3_frame.pdc')
    gridmap.save_pcd_ascii(path, sm)
    rd = gridmap.load_pcd_ascii(path)
    assert rd['n'] == 35
    np.testing.assert_array_equal(rd['points'], sm['points'])
    np.testing.assert_allclose(rd['origin'], sm['origin'], rtol=1e-8)
    np.testing.assert_allclose(rd['quat'], sm['quat'], rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
def test_submap_builder_windows_pings_on_the_gpu(tmp_path):
    """N-ping windowing with the points produced by the GPU kernel in the submap frame == the host pclFuser
    restatement on the same pings."""
    from smarc_navigation_amd import gridmap, synth
    B, meas = 64, 4
    ba = synth.beam_angles(B)
    stream = synth.odom_stream(10 * 50)
    poses = stream['truth'][49::50][:10]
    rs = np.random.RandomState(2)
    ranges = (18.0 / np.cos(ba)[None, :] + 0.05 * rs.randn(10, B)).astype(np.float32)
    ranges[3, 5] = 0.0
    sb = gridmap.SubmapBuilder(meas, ba, 80.0)
    got = [sb.add_ping(p, r) for p, r in zip(poses, ranges)]
    assert [g is not None for g in got] == [False, False, False, True, False, False, False, True, False, False]
    assert len(sb.submaps) == 2
    for k, sm in enumerate(sb.submaps):
        sl = slice(k * meas, (k + 1) * meas)
        # the same pings as points in their own base frames (beam b looks along (0, sin a, -cos a))
        pts_base = []
        for r in ranges[sl]:
            ok = r > 0
            pts_base.append(np.column_stack([np.zeros(B), np.sin(ba) * r, -np.cos(ba) * r])[ok])
        ref = gridmap.fuse_swath(pts_base, poses[sl], index=k)
        assert sm['frame_id'] == ref['frame_id'] and sm['points'].shape == ref['points'].shape
        np.testing.assert_allclose(sm['points'], ref['points'], atol=2e-5)
        np.testing.assert_allclose(sm['origin'], ref['origin'], atol=1e-12)
    paths = sb.save(str(tmp_path))
    assert [p.split('/')[-1] for p in paths] == ['submap_0_frame.pdc', 'submap_1_frame.pdc']
    assert gridmap.load_pcd_ascii(paths[1])['n'] == sb.submaps[1]['points'].shape[0]
