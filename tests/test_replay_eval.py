"""Replay / evaluation tool: the track metrics restate visual_tools.py:61-76 (CPU test); the replay
end-to-end run needs the GPU."""
import numpy as np
import pytest

from smarc_navigation_amd import replay as rp
from smarc_navigation_amd import synth


def test_track_metrics_match_visual_tools_formula():
    rs = np.random.RandomState(0)
    vec = np.cumsum(rs.randn(3, 50), axis=1)
    dist, final = rp.track_metrics(vec)
    ref = sum(np.linalg.norm(vec[:, i] - vec[:, i - 1]) for i in range(1, vec.shape[1]))
    assert dist == pytest.approx(ref, rel=1e-15)
    assert final == pytest.approx(np.linalg.norm(vec[:, -1]), rel=1e-15)
    assert rp.pose_rmse([[0, 0], [3, 4]], [[0, 0], [0, 0]]) == pytest.approx(np.sqrt(12.5))


@pytest.mark.gpu
def test_replay_with_mbes_pings_tracks_the_truth():
    from smarc_navigation_amd import engine
    n_steps, B = 150, 96
    st = synth.odom_stream(n_steps)
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    e = engine.Engine(1, rng_mode=engine.RNG_REPLAY)
    e.set_map_grid(z, origin, 1.0)
    idx = np.arange(4, n_steps, 5)
    ranges = np.zeros((idx.size, B), np.float32)
    rs = np.random.RandomState(4)
    for j, k in enumerate(idx):
        e.set_particles(st['truth'][k][:, None].copy())
        ranges[j] = e.mbes_expected(0, 1, ba, 80.0)[0] + 0.1 * rs.randn(B)
    # a dead-reckoning track that drifts away from the truth
    dr = st['truth'][:, :3].copy()
    dr[:, 0] += 0.02 * np.arange(n_steps)
    stream = dict(stamp=st['stamp'], v=st['v'], wz=st['wz'], q=st['q'], z=st['z'], t0=st['t0'],
                  mbes_idx=idx, mbes_ranges=ranges, mbes_angles=ba, mbes_range_max=80.0,
                  truth_xyz=st['truth'][:, :3], dr_xyz=dr)
    out = rp.replay(stream, dict(particle_count=16384, init_covariance='[1.0, 1.0, 0.0, 0.0, 0.0, 0.01]',
                                 motion_covariance='[0.0001, 0.0001, 0.0, 0.0, 0.0, 0.000001]',
                                 resampling_noise_covariance='[0.01, 0.01, 0.0, 0.0, 0.0, 0.00001]', seed=2),
                    grid=dict(z=z, origin=origin, res=1.0))
    s = out['summary']
    print(s)
    assert s['pf_rmse_vs_truth'] < 0.35
    assert abs(s['pf_distance'] - s['truth_distance']) < 1.0


def test_raw_events_become_the_odometry_stream_the_filter_consumes():
    """Host-only: raw IMU/DVL/depth events -> dead-reckoning integrator -> replay stream."""
    t, k, d = synth.raw_sensor_events(duration=12.0, seed=4)
    stream, m2o = rp.odom_stream_from_raw(t, k, d, pressure_tf=(0.3, 0.0, 0.05))
    n = len(stream['stamp'])
    assert n > 500 and stream['v'].shape == (n, 3) and stream['q'].shape == (n, 4)
    assert np.all(np.diff(stream['stamp']) > 0)
    assert m2o.shape == (4, 4) and m2o[0, 3] == 12.5 and m2o[1, 3] == -7.25
    # DVL ticks carry the body-frame DVL velocity, the depth is the pressure reading
    assert np.median(stream['v'][:, 0]) == pytest.approx(1.0, abs=0.1)
    assert stream['z'].min() < -1.9
    np.testing.assert_allclose(np.linalg.norm(stream['q'], axis=1), 1.0, atol=1e-12)


@pytest.mark.gpu
def test_raw_sensor_replay_without_fixes_follows_dead_reckoning():
    """Raw events -> DR -> particle filter; with no GPS/MBES correction and small process noise the
    filter's mean pose must stay on the dead-reckoning track (same motion model)."""
    t, k, d = synth.raw_sensor_events(duration=20.0, seed=6)
    stream, m2o = rp.odom_stream_from_raw(t, k, d, pressure_tf=(0.3, 0.0, 0.05))
    out = rp.replay(stream, dict(particle_count=8192, init_covariance='[0.0, 0.0, 0.0, 0.0, 0.0, 0.0]',
                                 motion_covariance='[0.000001, 0.000001, 0.0, 0.0, 0.0, 0.0]',
                                 resampling_noise_covariance='[0.0, 0.0, 0.0, 0.0, 0.0, 0.0]', seed=3), m2o=m2o)
    s = out['summary']
    print(s)
    # the filter integrates the yaw rate once per odometry sample (50 Hz), the integrator once per IMU
    # sample (100 Hz): the tracks agree to discretisation, so compare path length and depth
    assert abs(s['pf_distance'] - s['dr_distance']) < 0.05 * s['dr_distance'] + 0.1
    assert np.allclose(out['pf_xyz'][:, 2], stream['dr_xyz'][out['pub_idx'], 2], atol=1e-9)


def test_drstats_mirror_matches_the_reference_node():
    """DRStats against visual_tools.py itself (golden: oracle/ref_harness/gen_golden_stats.py ran the
    reference's odom_cb / finish_hld on synchronised triples, three of them without tf)."""
    from tests import helpers
    from smarc_navigation_amd import replay
    g = helpers.load('visual_tools_stats')
    st = replay.DRStats()
    dropped = set(int(k) for k in g['dropped'])
    for k in range(g['dr'].shape[0]):
        st.utm2odom = None if k in dropped else g['utm2odom']
        assert st.odom_cb(g['gps_utm'][k], g['dr'][k], g['pf'][k]) == (k not in dropped)
    assert st.filter_cnt == int(g['filter_cnt'])
    np.testing.assert_allclose(st.gps_odom_vec, g['gps_odom_vec'], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(st.dr_odom_vec, g['dr_odom_vec'])
    np.testing.assert_array_equal(st.pf_odom_vec, g['pf_odom_vec'])
    e_pf, e_dr = st.error_series()
    np.testing.assert_allclose(e_pf, g['err_pf'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(e_dr, g['err_dr'], rtol=0, atol=1e-9)
    fin = st.finish_hld()
    for name, val in zip(g['printed_names'], g['printed_values']):
        assert abs(fin[str(name)] - float(val)) <= 1e-9 * max(1.0, abs(float(val))), name


def test_approximate_time_sync_follows_message_filters():
    from smarc_navigation_amd.replay import ApproximateTimeSync
    s = ApproximateTimeSync(3, queue_size=3, slop=0.5)
    assert s.add(0, 10.0, 'g10') is None
    assert s.add(1, 10.1, 'd10') is None
    assert s.add(2, 10.2, 'p10') == ('g10', 'd10', 'p10')     # span 0.2 < slop
    assert s.add(2, 10.25, 'p10b') is None                       # its partners were consumed
    assert s.add(0, 11.0, 'g11') is None
    assert s.add(1, 11.6, 'd11') is None                         # 0.6 from g11: outside the slop of every gps message
    assert s.add(1, 11.2, 'd11b') is None                        # pf still missing within slop (10.25 is 0.95 away)
    assert s.add(2, 11.3, 'p11') == ('g11', 'd11b', 'p11')      # nearest partners first: d11b (0.1) before d11 (0.3)
    # queue_size: the oldest stamps fall out
    for k in range(5):
        s.add(0, 20.0 + k, 'g%d' % k)
    assert sorted(s.queues[0]) == [22.0, 23.0, 24.0]
    # span test is strict (< slop)
    t = ApproximateTimeSync(2, 5, 1.0)
    t.add(0, 1.0, 'a')
    assert t.add(1, 2.0, 'b') is None
    assert t.add(1, 1.9, 'c') == ('a', 'c')
