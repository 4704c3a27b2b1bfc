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
