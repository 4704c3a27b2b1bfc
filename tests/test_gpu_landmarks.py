"""Landmark k-NN data-association update (BASELINE config 5) vs the brute-force fp64 self-oracle.
PARITY UNPINNED vs the reference: auv_particle_filter has no landmark model."""
import numpy as np
import pytest

from smarc_navigation_amd import synth

pytestmark = pytest.mark.gpu


def _scene(n, n_lm=4096, n_det=16, seed=6):
    rs = np.random.RandomState(seed)
    lm = np.stack([rs.uniform(-64, 448, n_lm), rs.uniform(-256, 256, n_lm), rs.uniform(-24, -16, n_lm)], axis=1)
    soa = rs.randn(6, n) * np.array([1.0, 1.0, 0.2, 0.02, 0.02, 0.05])[:, None]
    soa[0] += 100.0
    soa[1] += 20.0
    soa[2] += -2.0
    truth = np.array([100.0, 20.0, -2.0, 0.0, 0.0, 0.0])
    # detections = the landmarks nearest to the truth sensor, seen from the truth pose, + noise
    d2 = (lm[:, 0] - truth[0]) ** 2 + (lm[:, 1] - truth[1]) ** 2
    near = np.argsort(d2)[:n_det]
    det = lm[near] - truth[:3] + 0.05 * rs.randn(n_det, 3)  # truth has identity rotation
    return lm, soa, det


@pytest.mark.parametrize('k', [1, 2, 4])
def test_landmark_update_matches_bruteforce_oracle(k):
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    n = 3000
    lm, soa, det = _scene(n)
    det[3] = np.nan  # an invalid detection is skipped
    m2o = synth.rigid_matrix(0.5, -0.5, 0.0, 0.0, 0.0, 0.02)
    off = [0.1, 0.0, -0.2, 0.0, 0.01, 0.0]
    e = eng.Engine(n, m2o=m2o, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    e.update_landmarks(det, 0.5, k=k, gate=11.345, sensor_offset=off)
    lw = e.get_log_weights()
    ref = orc.landmark_update(soa, m2o, off, lm, det, 0.5, k, 11.345)
    np.testing.assert_allclose(lw, ref, rtol=1e-11, atol=1e-9)
    assert np.std(ref) > 1.0  # the update discriminates between particles


def test_landmark_dense_map_exercises_knn_mixture():
    """Many landmarks inside one gate: the k-nearest bookkeeping matters."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    rs = np.random.RandomState(2)
    n = 500
    lm = np.stack([rs.uniform(-5, 5, 3000), rs.uniform(-5, 5, 3000), rs.uniform(-1, 1, 3000)], axis=1)
    soa = np.zeros((6, n))
    soa[0] = rs.uniform(-2, 2, n)
    soa[1] = rs.uniform(-2, 2, n)
    soa[5] = rs.uniform(-3, 3, n)
    det = rs.uniform(-2, 2, size=(20, 3))  # more than 16 detections: lanes loop
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    for k in (1, 3):
        e.update_landmarks(det, 0.3, k=k, gate=9.0)
        ref = orc.landmark_update(soa, np.identity(4), [0] * 6, lm, det, 0.3, k, 9.0)
        np.testing.assert_allclose(e.get_log_weights(), ref, rtol=1e-11, atol=1e-9)


def test_landmarks_accumulate_onto_mbes_and_filter_converges():
    from smarc_navigation_amd import engine as eng
    n = 20000
    lm, soa, det = _scene(n)
    origin = (-64.0, -256.0)
    z = synth.bathymetry_grid(512, 512, 1.0, origin, seed=3)
    ba = synth.beam_angles(64)
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_map_grid(z, origin, 1.0)
    e.set_landmarks(lm)
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    one.set_map_grid(z, origin, 1.0)
    one.set_particles(np.array([[100.0], [20.0], [-2.0], [0.0], [0.0], [0.0]]))
    ranges = one.mbes_expected(0, 1, ba, 80.0)[0]
    e.update_mbes(ranges, ba, 0.2, 80.0)
    lw_mbes = e.get_log_weights()
    e.update_landmarks(det, 0.5, k=2, accumulate=True)
    lw_both = e.get_log_weights()
    e.update_landmarks(det, 0.5, k=2, accumulate=False)
    lw_lm = e.get_log_weights()
    np.testing.assert_allclose(lw_both, lw_mbes + lw_lm, rtol=1e-12, atol=1e-9)
    e.set_log_weights(lw_both, eng.WEIGHT_LOG_SHIFT)
    e.resample(0.37, np.zeros((n, 6)))
    mean, _, _ = e.mean_cov()
    assert np.hypot(mean[0] - 100.0, mean[1] - 20.0) < 0.15


def test_update_landmarks_needs_a_feature_map():
    from smarc_navigation_amd import engine as eng
    e = eng.Engine(16)
    e.init_particles()
    with pytest.raises(eng.MclError) as ei:
        e.update_landmarks(np.zeros((2, 3)), 0.5)
    assert ei.value.status == -5


@pytest.mark.parametrize('mode,n', [('iso', 70001), ('maha', 70001), ('iso', 3001)])
def test_fused_landmark_step_on_a_grid_equals_the_separate_calls(mode, n):
    """mcl_step_mbes_landmarks on a height grid at an odd particle count, isotropic and Mahalanobis, a sensor offset on
    both sensors, including a dt = 0 step (the predict does not run: z, roll, pitch are read from the state) and a
    ping without a single valid detection -- against the separate calls, bit for bit.  (70 001 particles: the fan sweep,
    whose staged beam-table copy carries the detections; 3 001: the ray traversal, where the landmark launch copies them.)"""
    from smarc_navigation_amd import engine as eng
    z = synth.bathymetry_grid(256, 256, 1.0, (-64.0, -128.0), seed=3)
    lm = synth.landmark_map(2048, (-60.0, -120.0, 180.0, 120.0))
    stream = synth.odom_stream(5)
    ba = synth.beam_angles(64)
    rs = np.random.RandomState(1)
    cov = dict(init_cov=[1.0, 1.0, 0.0, 0.0, 0.0, 0.02], process_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5],
               resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])
    off, lm_off = [0.2, 0.0, -0.1, 0.0, 0.02, 0.0], [0.5, 0.1, 0.0, 0.0, 0.0, 0.03]
    a = eng.Engine(n, seed=9, **cov)
    b = eng.Engine(n, seed=9, **cov)
    for e in (a, b):
        e.set_map_grid(z, (-64.0, -128.0), 1.0)
        e.set_landmarks(lm)
        if mode == 'maha':
            cov6 = np.tile([0.04, 0.0, 0.0, 0.09, 0.0, 0.01], (len(lm), 1))
            e.set_landmark_noise(cov6, [0.09, 0.01, 0.0, 0.09, 0.0, 0.04])
        e.init_particles()
    for k in range(5):
        t = stream['truth'][k]
        T = synth.rigid_matrix(*t)
        near = lm[np.argsort(np.sum((lm[:, :2] - t[:2]) ** 2, axis=1))[:12]]
        det = (near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(12, 3)
        if k == 3:
            det[:] = np.nan
        ranges = (20.0 + rs.rand(64)).astype(np.float32)
        dt = 0.0 if k == 2 else stream['dt']
        od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], dt)
        a.predict(*od)
        a.update_mbes(ranges, ba, 0.3, 100.0, sensor_offset=off)
        a.update_landmarks(det, 0.3, k=3, gate=11.345, sensor_offset=lm_off, accumulate=True)
        lw_a = a.get_log_weights()
        a.resample()
        b.step_mbes_landmarks(*od, ranges, ba, 0.3, 100.0, det, 0.3, k=3, gate=11.345, sensor_offset=off,
                              lm_sensor_offset=lm_off)
        assert np.array_equal(b.get_log_weights(), lw_a), k
        assert np.array_equal(b.last_indices(), a.last_indices()), k
        assert np.array_equal(b.get_particles(), a.get_particles()), k
        np.testing.assert_allclose(b.last_mean_cov()[0], a.mean_cov()[0], rtol=0, atol=1e-10)


def test_fused_landmark_step_argument_and_state_errors():
    from smarc_navigation_amd import engine as eng
    e = eng.Engine(256, seed=1)
    z = np.full((32, 32), -20.0)
    e.set_map_grid(z, (-16.0, -16.0), 1.0)
    e.init_particles()
    ba = synth.beam_angles(16)
    r = np.full(16, 20.0, np.float32)
    od = ([1.0, 0, 0], 0.0, [0, 0, 0, 1.0], -2.0, 0.1)
    with pytest.raises(eng.MclError, match='no feature map'):
        e.step_mbes_landmarks(*od, r, ba, 0.2, 100.0, np.zeros((2, 3)), 0.3)
    e.set_landmarks(np.zeros((4, 3)))
    with pytest.raises(eng.MclError, match='bad argument'):
        e.step_mbes_landmarks(*od, r, ba, 0.2, 100.0, np.zeros((2, 3)), 0.3, k=9)
    before = e.get_particles()
    with pytest.raises(eng.MclError):
        e.step_mbes_landmarks(*od, r, ba, -1.0, 100.0, np.zeros((2, 3)), 0.3)
    assert np.array_equal(e.get_particles(), before)   # nothing ran
    e.step_mbes_landmarks(*od, r, ba, 0.2, 100.0, np.zeros((2, 3)), 0.3)
    assert np.all(np.isfinite(e.last_mean_cov()[0]))


@pytest.mark.parametrize('seed', range(10))
def test_landmark_update_fuzz_against_bruteforce(seed):
    """Random feature maps (1 .. 3 000 landmarks, clustered or spread), gates, 1 .. 40 detections, k = 1 .. 4, isotropic and
    Mahalanobis, poses inside, at the border of, and far outside the landmark grid (the clamped outer ring of the
    neighbourhood table), a NaN pose and an infinite one: every particle against the brute-force oracle."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    rs = np.random.RandomState(100 + seed)
    n = int(rs.choice([1, 63, 64, 65, 700, 2049]))
    n_lm = int(rs.choice([1, 2, 17, 400, 3000]))
    span = float(rs.choice([2.0, 30.0, 400.0]))
    lm = np.stack([rs.uniform(-span, span, n_lm), rs.uniform(-span, span, n_lm), rs.uniform(-3, 3, n_lm)], axis=1)
    if seed % 3 == 0:   # clustered: many landmarks inside one gate
        lm[:, :2] = lm[:, :2] * 0.02 + rs.uniform(-span, span, 2)
    n_det = int(rs.choice([1, 5, 16, 17, 40]))
    k = int(rs.randint(1, 5))
    sigma = float(rs.choice([0.1, 0.5, 2.0]))
    gate = float(rs.choice([3.0, 11.345, 40.0]))
    soa = np.zeros((6, n))
    centre = lm[rs.randint(n_lm)]
    soa[0] = centre[0] + rs.randn(n) * sigma * 3
    soa[1] = centre[1] + rs.randn(n) * sigma * 3
    soa[2] = centre[2] + rs.randn(n) * 0.3
    soa[3:5] = rs.randn(2, n) * 0.05
    soa[5] = rs.uniform(-3.1, 3.1, n)
    far = rs.rand(n) < 0.2   # outside the grid, by a little and by a lot
    soa[0, far] += rs.choice([-1.0, 1.0], far.sum()) * rs.choice([span, 3 * span + 50.0, 1e7], far.sum())
    soa[1, far] += rs.choice([-1.0, 1.0], far.sum()) * rs.choice([0.0, span, 1e9], far.sum())
    if n > 2:
        soa[0, 1] = np.nan
        soa[1, 2] = np.inf
    det = rs.randn(n_det, 3) * np.array([2.0, 2.0, 0.5])
    if n_det > 2:
        det[1] = np.nan
    m2o = synth.rigid_matrix(*(rs.randn(3) * 2.0), 0.0, 0.0, rs.uniform(-3, 3))
    off = [0.3, -0.1, 0.2, 0.0, 0.03, 0.1]
    e = eng.Engine(n, m2o=m2o, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    if seed % 2:
        cov6 = np.abs(rs.randn(n_lm, 6)) * np.array([0.05, 0.0, 0.0, 0.05, 0.0, 0.02])
        Q6 = [sigma ** 2, 0.1 * sigma ** 2, 0.0, 1.5 * sigma ** 2, 0.0, 0.5 * sigma ** 2]
        e.set_landmark_noise(cov6, Q6)
        ref = orc.landmark_update_maha(soa, m2o, off, lm, det, sigma, k, gate, lmcov=cov6, Q6=Q6)
    else:
        ref = orc.landmark_update(soa, m2o, off, lm, det, sigma, k, gate)
    e.update_landmarks(det, sigma, k=k, gate=gate, sensor_offset=off)
    lw = e.get_log_weights()
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(lw), fin)
    np.testing.assert_allclose(lw[fin], ref[fin], rtol=1e-10, atol=1e-8)
