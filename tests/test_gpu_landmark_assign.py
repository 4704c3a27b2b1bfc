"""Landmark update with a global (Hungarian) assignment per particle (include/mcl.h
mcl_update_landmarks_assign; table as auv_ekf_slam/src/ekf_slam_core.cpp:172-312) vs the oracle: a dense
table over ALL landmarks solved by orc_assign_dense, which tests/test_oracle_assign.py pins to the
reference's own Munkres.  The particle-filter adaptation itself is this build's definition (parity
unpinned), the optimal total is unique so the log-weights compare to rounding."""
import numpy as np
import pytest

from smarc_navigation_amd import synth

pytestmark = pytest.mark.gpu


def _run(n, lm, soa, det, sigma, k_cand, gate, new_mh, m2o=None, off=None, keep=0):
    from smarc_navigation_amd import engine as eng
    e = eng.Engine(n, m2o=m2o, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    asg = e.update_landmarks_assign(det, sigma, k_cand=k_cand, gate=gate, new_mh_dist=new_mh, sensor_offset=off,
                                    n_keep=keep)
    lw = e.get_log_weights()
    e.close()
    return lw, asg


def test_sparse_map_matches_dense_oracle():
    from oracle import oracle as orc
    rs = np.random.RandomState(6)
    n, n_lm = 2000, 4096
    lm = np.stack([rs.uniform(-64, 448, n_lm), rs.uniform(-256, 256, n_lm), rs.uniform(-24, -16, n_lm)], axis=1)
    soa = rs.randn(6, n) * np.array([1.0, 1.0, 0.2, 0.02, 0.02, 0.05])[:, None]
    soa[0] += 100.0
    soa[1] += 20.0
    soa[2] += -2.0
    d2 = (lm[:, 0] - 100.0) ** 2 + (lm[:, 1] - 20.0) ** 2
    near = np.argsort(d2)[:16]
    det = lm[near] - np.array([100.0, 20.0, -2.0]) + 0.05 * rs.randn(16, 3)
    det[3] = np.nan
    det[9] = det[8] + 0.02          # two detections of the same landmark: a conflict for the assignment
    m2o = synth.rigid_matrix(0.5, -0.5, 0.0, 0.0, 0.0, 0.02)
    off = [0.1, 0.0, -0.2, 0.0, 0.01, 0.0]
    lw, asg = _run(n, lm, soa, det, 0.5, 8, 11.345, 9.0, m2o, off, keep=n)
    ref, rasg = orc.landmark_assign_update(soa, m2o, off, lm, det, 0.5, 8, 11.345, 9.0, want_assign=True)
    np.testing.assert_allclose(lw, ref, rtol=1e-11, atol=1e-9)
    assert np.std(ref) > 1.0
    # a landmark explains at most one detection; invalid detections are flagged; the cost-unique cases agree
    assert (asg[:, 3] == -2).all()
    for i in range(0, n, 37):
        used = asg[i][asg[i] >= 0]
        assert used.size == np.unique(used).size
    assert (asg == rasg).mean() > 0.999
    both = (asg[:, 8] >= 0) & (asg[:, 9] >= 0)
    assert not np.any(asg[both, 8] == asg[both, 9])


@pytest.mark.parametrize('k_cand,new_mh', [(8, 11.345), (3, 6.0), (1, 2.0)])
def test_dense_map_with_many_conflicts(k_cand, new_mh):
    """Landmark density high enough that detections compete for the same landmarks all the time."""
    from oracle import oracle as orc
    rs = np.random.RandomState(3)
    n = 700
    lm = np.stack([rs.uniform(-6, 6, 220), rs.uniform(-6, 6, 220), rs.uniform(-1, 1, 220)], axis=1)
    soa = np.zeros((6, n))
    soa[0] = rs.uniform(-2, 2, n)
    soa[1] = rs.uniform(-2, 2, n)
    soa[5] = rs.uniform(-3, 3, n)
    det = rs.uniform(-2.5, 2.5, size=(16, 3))
    det[:, 2] *= 0.3
    det[5] = det[4] + 0.05
    det[6] = det[4] - 0.05
    lw, asg = _run(n, lm, soa, det, 0.6, k_cand, 11.345, new_mh, keep=64)
    ref, rasg = orc.landmark_assign_update(soa, np.identity(4), [0] * 6, lm, det, 0.6, k_cand, 11.345, new_mh,
                                           want_assign=True)
    np.testing.assert_allclose(lw, ref, rtol=1e-11, atol=1e-9)
    # conflicts really happen: the independent nearest-neighbour choice would reuse landmarks
    nn = orc.landmark_update(soa, np.identity(4), [0] * 6, lm, det, 0.6, 1, 11.345)
    assert np.mean(np.abs(nn - ref) > 1e-6) > 0.5
    for i in range(64):
        used = asg[i][asg[i] >= 0]
        assert used.size == np.unique(used).size


def test_edge_cases_and_accumulate():
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    rs = np.random.RandomState(9)
    n = 130   # not a multiple of the 8 particles per block
    lm = np.stack([rs.uniform(-10, 10, 50), rs.uniform(-10, 10, 50), rs.uniform(-6, -4, 50)], axis=1)
    soa = rs.randn(6, n) * np.array([0.5, 0.5, 0.1, 0.01, 0.01, 0.02])[:, None]
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    # all detections invalid -> log-weight 0 for everyone
    det = np.full((4, 3), np.nan)
    e.update_landmarks_assign(det, 0.5)
    assert np.all(e.get_log_weights() == 0.0)
    # nothing inside any gate -> every detection takes the new-landmark hypothesis
    det = np.array([[500.0, 0.0, 0.0], [0.0, 500.0, 0.0]])
    asg = e.update_landmarks_assign(det, 0.5, new_mh_dist=7.0, n_keep=n)
    assert (asg == -1).all()
    lognorm = 1.5 * np.log(2 * np.pi) + 3 * np.log(0.5)
    np.testing.assert_allclose(e.get_log_weights(), -0.5 * 14.0 - 2 * lognorm, rtol=1e-14)
    # a single detection
    det = lm[7:8] + 0.1
    e.update_landmarks_assign(det, 0.5)
    ref = orc.landmark_assign_update(soa, np.identity(4), [0] * 6, lm, det, 0.5, 8, 11.345, 11.345)
    np.testing.assert_allclose(e.get_log_weights(), ref, rtol=1e-11, atol=1e-9)
    # accumulate onto an earlier update of the same ping
    e.update_landmarks(det, 0.5, k=1)
    base = e.get_log_weights().copy()
    e.update_landmarks_assign(det, 0.5, accumulate=True)
    np.testing.assert_allclose(e.get_log_weights(), base + ref, rtol=1e-11, atol=1e-9)
    # argument checks
    with pytest.raises(eng.MclError):
        e.update_landmarks_assign(np.zeros((17, 3)), 0.5)
    with pytest.raises(eng.MclError):
        e.update_landmarks_assign(det, 0.5, k_cand=9)
    e.close()


def test_full_filter_step_with_assignment_localises():
    """Predict-free loop: assignment update + resample pulls a displaced cloud onto the truth."""
    from smarc_navigation_amd import engine as eng
    rs = np.random.RandomState(12)
    n = 20000
    lm = np.stack([rs.uniform(-40, 40, 400), rs.uniform(-40, 40, 400), rs.uniform(-12, -8, 400)], axis=1)
    truth = np.array([3.0, -2.0, 0.0])
    e = eng.Engine(n, seed=4, init_cov=[4.0, 4.0, 0.0, 0.0, 0.0, 0.0], resample_cov=[0.01, 0.01, 0, 0, 0, 0])
    e.init_particles()
    e.set_landmarks(lm)
    d2 = np.sum((lm[:, :2] - truth[:2]) ** 2, axis=1)
    near = np.argsort(d2)[:12]
    for it in range(6):
        det = lm[near] - truth + 0.05 * rs.randn(12, 3)
        e.update_landmarks_assign(det, 0.3, new_mh_dist=11.345)
        e.resample()
    mean, _, _ = e.mean_cov()
    assert abs(mean[0] - truth[0]) < 0.1 and abs(mean[1] - truth[1]) < 0.1
    e.close()


def _maha_scene(seed, n=3000, n_lm=400, n_det=10, dense=False):
    rs = np.random.RandomState(seed)
    span = 6.0 if dense else 25.0
    lm = np.column_stack([rs.uniform(-span, span, n_lm), rs.uniform(-span, span, n_lm), -20 + rs.randn(n_lm)])
    cov = np.zeros((n_lm, 6))
    for j in range(n_lm):
        A = rs.randn(3, 3) * np.array([0.4, 0.2, 0.1])
        S = A.dot(A.T) + 0.01 * np.identity(3)
        cov[j] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
    Aq = rs.randn(3, 3) * 0.15
    Q = Aq.dot(Aq.T) + 0.02 * np.identity(3)
    Q6 = np.array([Q[0, 0], Q[0, 1], Q[0, 2], Q[1, 1], Q[1, 2], Q[2, 2]])
    soa = rs.randn(6, n) * np.array([1.0, 1.0, 0.1, 0.03, 0.03, 0.2])[:, None]
    soa[2] -= 2.0
    m2o = synth.rigid_matrix(0.5, -0.25, 0.0, 0.0, 0.0, 0.1)
    off = [0.2, 0.0, -0.1, 0.0, 0.02, 0.05]
    near = np.argsort(np.sum(lm[:, :2] ** 2, axis=1))[:n_det]
    det = (lm[near] - [0.5, -0.25, -2.0]) + 0.3 * rs.randn(n_det, 3)
    det[1] = np.nan
    return soa, m2o, off, lm, cov, Q6, det


@pytest.mark.parametrize('with_cov,with_q', [(True, True), (False, True), (True, False)])
def test_mahalanobis_knn_and_assignment_match_oracle(with_cov, with_q):
    """mcl_set_landmark_noise: per-landmark covariance and sensor-frame Q.  The kernels evaluate the distance
    in the map frame with cofactors, the oracle the reference's way in the sensor frame with Gaussian
    elimination: the same numbers to 1e-9."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    soa, m2o, off, lm, cov, Q6, det = _maha_scene(5)
    e = eng.Engine(soa.shape[1], m2o=m2o, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    e.set_landmark_noise(cov if with_cov else None, Q6 if with_q else None)
    kw = dict(lmcov=cov if with_cov else None, Q6=Q6 if with_q else None)
    for k in (1, 3):
        e.update_landmarks(det, 0.3, k=k, gate=11.345, sensor_offset=off)
        ref = orc.landmark_update_maha(soa, m2o, off, lm, det, 0.3, k, 11.345, **kw)
        np.testing.assert_allclose(e.get_log_weights(), ref, rtol=1e-9, atol=1e-9)
    asg = e.update_landmarks_assign(det, 0.3, k_cand=8, gate=11.345, new_mh_dist=9.0, sensor_offset=off, n_keep=200)
    ref, ref_asg = orc.landmark_assign_update_maha(soa, m2o, off, lm, det, 0.3, 8, 11.345, 9.0, want_assign=True, **kw)
    np.testing.assert_allclose(e.get_log_weights(), ref, rtol=1e-9, atol=1e-9)
    same = np.mean(asg == ref_asg[:200])
    assert same > 0.98  # ties between equal-cost optima may be broken differently
    # back to the isotropic distance
    e.set_landmark_noise(None, None)
    e.update_landmarks(det, 0.3, k=2, gate=11.345, sensor_offset=off)
    np.testing.assert_allclose(e.get_log_weights(), orc.landmark_update(soa, m2o, off, lm, det, 0.3, 2, 11.345), rtol=1e-9, atol=1e-9)


def test_dense_cluster_more_gated_landmarks_than_candidates():
    """with more than k_cand landmarks inside the gate of a detection the sparse graph keeps only the
    k_cand nearest; that truncation is part of the definition (mcl.h) and the oracle applies the same rule, so
    the GPU must still reproduce it exactly on a map dense enough that every detection has > 8 gated landmarks."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    soa, m2o, off, lm, cov, Q6, det = _maha_scene(7, n=1500, n_lm=600, dense=True)
    e = eng.Engine(soa.shape[1], m2o=m2o, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_landmarks(lm)
    e.set_landmark_noise(cov, Q6)
    # how many landmarks are gated per detection (particle 0): the premise of the test
    _, _, tab = orc.landmark_assign_update_maha(soa[:, :1].copy(), m2o, off, lm, det, 0.3, 600, 11.345, 9.0, lmcov=cov, Q6=Q6,
                                                want_assign=True, want_table=True)
    gated = (tab[:600] < 10000.0).sum(axis=0)
    print('gated landmarks per detection:', gated)
    assert gated.max() > 8
    e.update_landmarks_assign(det, 0.3, k_cand=8, gate=11.345, new_mh_dist=9.0, sensor_offset=off)
    ref = orc.landmark_assign_update_maha(soa, m2o, off, lm, det, 0.3, 8, 11.345, 9.0, lmcov=cov, Q6=Q6)
    np.testing.assert_allclose(e.get_log_weights(), ref, rtol=1e-9, atol=1e-9)
