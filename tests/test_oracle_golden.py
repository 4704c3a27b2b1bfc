"""CPU tests: the oracle (oracle/mcl_oracle.c) against the golden vectors produced by the
reference's own Python (tests/golden, generator oracle/ref_harness/gen_golden.py)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import oracle as orc
from tests import helpers


def test_transform_helpers_vs_scipy():
    rs = np.random.RandomState(3)
    for _ in range(200):
        rpy = rs.uniform(-1, 1, 3) * np.array([np.pi, np.pi / 2 * 0.98, np.pi])
        q = orc.quat_from_euler(*rpy)
        q_sp = Rotation.from_euler('xyz', rpy).as_quat()
        if np.dot(q, q_sp) < 0:
            q_sp = -q_sp
        np.testing.assert_allclose(q, q_sp, atol=1e-14)
        np.testing.assert_allclose(orc.euler_from_quat(q), rpy, atol=1e-12)
        M = orc.matrix_from_tf([1, 2, 3], q)
        np.testing.assert_allclose(M[:3, :3], Rotation.from_euler('xyz', rpy).as_matrix(), atol=1e-14)
        np.testing.assert_allclose(M[:3, 3], [1, 2, 3])


def test_wrap_matches_python_floored_mod():
    for a in [0.0, 3.0, -3.0, np.pi, -np.pi, 7.5, -7.5, 100.0, -100.0, 1e-300, 2 * np.pi, -2 * np.pi]:
        assert orc.wrap_pi(a) == (np.float64(a) + np.pi) % (2 * np.pi) - np.pi


def test_numpy_sum_restatement_is_bit_exact():
    rs = np.random.RandomState(0)
    for n in [1, 5, 8, 9, 100, 128, 129, 1000, 8192, 8193, 20000, 65536, 300001]:
        a = rs.rand(n) * rs.choice([1e-3, 1.0, 1e3])
        assert orc.numpy_sum(a) == a.sum()
        m = rs.rand(n, 6)
        col = np.ascontiguousarray(m[:, 5])
        assert orc.numpy_sum(col) / n == np.mean(m[:, 5])


def test_particle_kat():
    g = helpers.load('particle_kat')
    n = g['mp_pose0'].shape[0]
    for i in range(n):
        soa = orc.to_soa(g['mp_pose0'][i:i + 1])
        nz = np.random.RandomState(int(g['mp_seeds'][i])).randn(1, 6)
        orc.predict(soa, g['mp_v'][i], g['mp_wz'][i], g['mp_q'][i], g['mp_z'][i], g['mp_dt'][i], g['mp_pcov'][i], nz)
        np.testing.assert_allclose(orc.from_soa(soa)[0], g['mp_pose1'][i], rtol=0, atol=1e-12)
    # compute_weight (scipy pdf) vs closed form
    for i in range(n):
        soa = orc.to_soa(g['mp_pose0'][i:i + 1])
        w, lw = orc.gps_weights(soa, g['m2o'], g['cw_gps'][i, 0], g['cw_gps'][i, 1], g['cw_std'][i])
        ref = g['cw_w'][i]
        if ref > 1e-300:
            assert abs(w[0] - ref) <= 1e-12 * ref
            assert abs(np.exp(lw[0]) - ref) <= 1e-11 * ref
        else:
            assert w[0] <= 1e-300
    # euler / quaternion helpers
    for i in range(n):
        np.testing.assert_allclose(orc.euler_from_quat(g['eq_q'][i]), g['eq_rpy'][i], atol=1e-15)
        np.testing.assert_allclose(orc.quat_from_euler(*g['fr_rpy'][i]), g['qe_q'][i], atol=1e-15)
    # add_noise
    soa = orc.to_soa(g['an_pose0'][None, :])
    orc.add_noise(soa, g['an_cov'], np.random.RandomState(int(g['an_seed'])).randn(1, 6))
    np.testing.assert_allclose(orc.from_soa(soa)[0], g['an_pose1'], atol=1e-15)
    # matrix_from_tf
    np.testing.assert_allclose(orc.matrix_from_tf(g['mt_in'][:3], g['mt_in'][3:]), g['mt_M'], atol=1e-15)


def test_fullrotation_rows_used_by_predict():
    """Rows 0-1 of the reference's (malformed) fullRotation equal the proper Rz*Ry*Rx."""
    g = helpers.load('particle_kat')
    for rpy, R in zip(g['fr_rpy'], g['fr_R']):
        Rp = Rotation.from_euler('xyz', rpy).as_matrix()
        np.testing.assert_allclose(R[:2], Rp[:2], atol=1e-14)


def _uniforms(seed, count):
    return np.random.RandomState(seed).random_sample(count)


def test_resampling_kat_all_schemes():
    g = helpers.load('resampling_kat')
    n_checked = 0
    for tag in g['cases']:
        w = g[tag + '_w']
        seed = int(g[tag + '_seed'])
        n = w.size
        key = tag + '_systematic_resample'
        if key in g:
            idx, rc = orc.systematic_ref(w, _uniforms(seed, 1)[0])
            assert rc == 0 and np.array_equal(idx, g[key]), tag
            n_checked += 1
        key = tag + '_stratified_resample'
        if key in g:
            idx, rc = orc.stratified_ref(w, _uniforms(seed, n))
            assert rc == 0 and np.array_equal(idx, g[key]), tag
            n_checked += 1
        key = tag + '_multinomial_resample'
        if key in g:
            idx, rc = orc.multinomial_ref(w, _uniforms(seed, n))
            assert np.array_equal(idx, g[key]), tag
            n_checked += 1
        key = tag + '_residual_resample'
        if key in g:
            k = orc.residual_k(w)
            idx, k2 = orc.residual_ref(w, _uniforms(seed, n - k))
            assert k == k2 and np.array_equal(idx, g[key]), tag
            n_checked += 1
        key = tag + '_naive_resample'
        if key in g:
            idx, rc = orc.naive_ref(w, _uniforms(seed, 1)[0])
            assert rc == 0 and np.array_equal(idx, g[key]), tag
            n_checked += 1
    assert n_checked > 120


def test_lost_dupes_matches_reference_list_semantics():
    rs = np.random.RandomState(5)
    for n in [1, 2, 10, 200]:
        for _ in range(20):
            idx = rs.randint(0, n, size=n)
            if rs.rand() < 0.5:
                idx = np.sort(idx)
            keep = list(set(idx.tolist()))
            lost = [i for i in range(n) if i not in keep]
            dupes = idx.tolist()
            for i in keep:
                dupes.remove(i)
            lo, du = orc.lost_dupes(idx)
            assert lo.tolist() == lost and du.tolist() == dupes


@pytest.mark.parametrize('name,resampler', [
    ('traj_predict_launch', 'residual'),
    ('traj_predict_motion2', 'residual'),
    ('traj_gps_residual', 'residual'),
    ('traj_gps_systematic', 'systematic'),
    ('traj_gps_systematic_n1000', 'systematic'),
    ('traj_gps_systematic_fullcov', 'systematic'),
])
def test_trajectory_replay_matches_reference(name, resampler):
    g = helpers.load(name)
    out = helpers.replay(g, helpers.OracleBackend(g, resampler))
    np.testing.assert_allclose(out['init_state'], g['init_state'], rtol=0, atol=1e-15)
    # indices / weights at every GPS fix
    for k in range(len(g['fix_idx'])):
        assert np.array_equal(out['indices'][k], g['indices'][k]), (name, k)
        np.testing.assert_allclose(out['weights_raw'][k], g['weights_raw'][k], rtol=1e-11, atol=0)
        np.testing.assert_allclose(out['weights_norm'][k], g['weights_norm'][k], rtol=1e-11, atol=0)
        np.testing.assert_allclose(out['post_update_states'][k], g['post_update_states'][k], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(out['ckpt_states']), g['ckpt_states'], rtol=0, atol=1e-9)
    # published mean pose, yaw (through the quaternion), covariance layout
    mean = np.array(out['mean'])
    np.testing.assert_allclose(mean[:, :3], g['mean_xyz'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(mean[:, :2], g['tf_trans'][:, :2], rtol=0, atol=1e-9)
    assert np.all(g['tf_trans'][:, 2] == 0.0)
    for k in range(mean.shape[0]):
        q = orc.quat_from_euler(mean[k, 3], mean[k, 4], out['yaw'][k])
        np.testing.assert_allclose(q, g['quat'][k], rtol=0, atol=1e-9)
        cov36 = g['cov36'][k]
        np.testing.assert_allclose(out['cov'][k], cov36[:9], rtol=1e-9, atol=1e-15)
        assert np.all(cov36[9:] == 0.0)
