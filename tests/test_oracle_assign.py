"""The oracle's assignment solver against the REFERENCE's own Munkres (oracle/_ref, compiled from
/root/reference/auv_ekf_slam/utils/munkres where it lies) on correspondence tables built the way
auv_ekf_slam/src/ekf_slam_core.cpp:172-296 builds them, and the assignment-based landmark update's
basic properties.  CPU only."""
import itertools

import numpy as np
import pytest

from oracle import oracle as orc


def _table(rs, n_lm, n_det, gate=11.34, new_mh=9.0, p_in=0.3):
    """rows = landmarks + one new-landmark row per measurement, cols = measurements (the reference layout)."""
    t = np.full((n_lm + n_det, n_det), 10000.0)
    inside = rs.rand(n_lm, n_det) < p_in
    t[:n_lm][inside] = rs.rand(int(inside.sum())) * gate
    for d in range(n_det):
        t[n_lm + d, d] = new_mh
    return t


def test_dense_solver_is_optimal_on_small_tables():
    rs = np.random.RandomState(0)
    for _ in range(40):
        n, m = rs.randint(1, 5), rs.randint(5, 8)
        c = rs.rand(n, m) * 10
        col, total = orc.assign_dense(c)
        best = min(sum(c[r, p[r]] for r in range(n)) for p in itertools.permutations(range(m), n))
        assert total == pytest.approx(best, rel=1e-14)
        assert len(set(col.tolist())) == n
        assert total == pytest.approx(sum(c[r, col[r]] for r in range(n)), rel=1e-14)


def test_dense_solver_matches_reference_munkres():
    rs = np.random.RandomState(1)
    probe = orc.ref_munkres(np.zeros((2, 2)))
    if probe is None:
        pytest.skip('oracle/_ref/libref_munkres.so not built (needs /root/reference)')
    n_cases = 0
    for n_lm, n_det in [(3, 2), (8, 4), (20, 8), (40, 16), (64, 16), (5, 7)]:
        for rep in range(12):
            t = _table(rs, n_lm, n_det, p_in=[0.1, 0.3, 0.7][rep % 3])
            row_of_col = orc.ref_munkres(t)
            assert (row_of_col >= 0).all() and len(set(row_of_col.tolist())) == n_det
            ref_total = sum(t[row_of_col[c], c] for c in range(n_det))
            col, total = orc.assign_dense(t.T.copy())      # oracle layout: rows = measurements
            assert total == pytest.approx(ref_total, rel=1e-12), (n_lm, n_det, rep)
            assert ref_total < 10000.0                       # never an "infinite" pair: each measurement owns a row
            n_cases += 1
    assert n_cases == 72


def test_assignment_update_properties():
    rs = np.random.RandomState(2)
    lm = np.column_stack([rs.uniform(-20, 20, 60), rs.uniform(-20, 20, 60), rs.uniform(-12, -8, 60)])
    soa = np.zeros((6, 32))
    soa[0], soa[1] = rs.randn(32) * 0.5, rs.randn(32) * 0.5
    soa[5] = rs.randn(32) * 0.05
    # detections = a few landmarks seen from the origin, two of them the SAME landmark (a conflict), one NaN
    pick = [3, 7, 7, 11, 20]
    det = lm[pick] + rs.randn(5, 3) * 0.1
    det = np.vstack([det, [np.nan, 0, 0], [100.0, 100.0, 0.0]])
    sigma, gate, new_mh = 0.5, 11.34, 11.34
    lw, asg = orc.landmark_assign_update(soa, np.identity(4), [0] * 6, lm, det, sigma, 8, gate, new_mh, want_assign=True)
    lognorm = 1.5 * np.log(2 * np.pi) + 3 * np.log(sigma)
    for i in range(32):
        a = asg[i]
        assert a[5] == -2                         # NaN detection skipped
        assert a[6] == -1                         # far detection -> new-landmark hypothesis
        used = [x for x in a if x >= 0]
        assert len(used) == len(set(used))        # a landmark explains at most one detection
        assert not (a[1] == 7 and a[2] == 7)
        assert lw[i] <= -6 * lognorm + 1e-12
    # with no conflicts the assignment equals independent nearest-neighbour association
    det2 = lm[[3, 11, 20]] + rs.randn(3, 3) * 0.05
    lw2, asg2 = orc.landmark_assign_update(soa, np.identity(4), [0] * 6, lm, det2, sigma, 8, gate, new_mh, want_assign=True)
    lw_nn = orc.landmark_update(soa, np.identity(4), [0] * 6, lm, det2, sigma, 1, gate)
    near = np.all(asg2 >= 0, axis=1)
    assert near.sum() > 10
    np.testing.assert_allclose(lw2[near], lw_nn[near], rtol=0, atol=1e-9)


def _maha_scene(seed, n_lm=120, n_det=7, n=6):
    rs = np.random.RandomState(seed)
    lm = np.column_stack([rs.uniform(-15, 15, n_lm), rs.uniform(-15, 15, n_lm), -20 + rs.randn(n_lm)])
    # per-landmark covariances: random SPD, anisotropic (map frame), 6 unique entries xx xy xz yy yz zz
    cov = np.zeros((n_lm, 6))
    for j in range(n_lm):
        A = rs.randn(3, 3) * np.array([0.4, 0.2, 0.1])
        S = A.dot(A.T) + 0.01 * np.identity(3)
        cov[j] = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
    Aq = rs.randn(3, 3) * 0.15
    Q = Aq.dot(Aq.T) + 0.02 * np.identity(3)
    Q6 = np.array([Q[0, 0], Q[0, 1], Q[0, 2], Q[1, 1], Q[1, 2], Q[2, 2]])
    soa = np.zeros((6, n))
    soa[0], soa[1] = rs.randn(n), rs.randn(n)
    soa[2] = -2.0
    soa[3], soa[4], soa[5] = 0.05 * rs.randn(n), 0.05 * rs.randn(n), 0.3 * rs.randn(n)
    m2o = np.identity(4)
    m2o[:3, 3] = [0.5, -0.25, 0.0]
    off = [0.2, 0.0, -0.1, 0.0, 0.02, 0.05]
    near = np.argsort(np.sum(lm[:, :2] ** 2, axis=1))[:n_det]
    det = lm[near] - [0.5, -0.25, -2.0] + 0.3 * rs.randn(n_det, 3)
    det[2] = np.nan
    return soa, m2o, off, lm, cov, Q6, det


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_mahalanobis_table_end_to_end_against_reference_munkres(seed):
    """The dense correspondence table built the reference's way (sensor-frame innovation, S = R^T Sigma_j R + Q,
    d_m if < gate else 10000, one new-landmark row per detection: ekf_slam_core.cpp:135-178,248-281) handed to
    the REFERENCE's own Munkres (oracle/_ref): its optimal total is the total behind the oracle's log-weight."""
    soa, m2o, off, lm, cov, Q6, det = _maha_scene(seed)
    gate, new_mh = 11.345, 9.0
    lw, asg, tab = orc.landmark_assign_update_maha(soa[:, :1].copy(), m2o, off, lm, det, 0.3, 8, gate, new_mh, lmcov=cov,
                                                   Q6=Q6, want_assign=True, want_table=True)
    row_of_col = orc.ref_munkres(tab)
    if row_of_col is None:
        pytest.skip('oracle/_ref not built (reference sources absent)')
    total_ref = sum(tab[row_of_col[c], c] for c in range(tab.shape[1]))
    nv = tab.shape[1]
    Q = np.array([[Q6[0], Q6[1], Q6[2]], [Q6[1], Q6[3], Q6[4]], [Q6[2], Q6[4], Q6[5]]])
    lognorm = 1.5 * np.log(2 * np.pi) + 0.5 * np.log(np.linalg.det(Q))
    assert abs((-0.5 * total_ref - nv * lognorm) - lw[0]) <= 1e-9
    # and the table entries are the textbook Mahalanobis distances (numpy, written out again here)
    from smarc_navigation_amd import synth
    Ms = m2o.dot(synth.rigid_matrix(*soa[:, 0])).dot(synth.rigid_matrix(*off))
    R, o = Ms[:3, :3], Ms[:3, 3]
    valid = [d for d in range(det.shape[0]) if not np.isnan(det[d]).any()]
    for c, d in enumerate(valid):
        for j in (0, 17, 63):
            C = np.array([[cov[j, 0], cov[j, 1], cov[j, 2]], [cov[j, 1], cov[j, 3], cov[j, 4]], [cov[j, 2], cov[j, 4], cov[j, 5]]])
            nu = det[d] - R.T.dot(lm[j] - o)
            dm = nu.dot(np.linalg.solve(R.T.dot(C).dot(R) + Q, nu))
            assert abs(tab[j, c] - (dm if dm < gate else 10000.0)) <= 1e-9 * max(1.0, dm) or (dm < gate and tab[j, c] == 10000.0)
