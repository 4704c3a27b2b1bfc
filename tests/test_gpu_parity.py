"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on identical inputs and against the golden vectors from the reference."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope='module')
def eng():
    from smarc_navigation_amd import engine
    return engine


class GpuBackend(object):
    """Same driver interface as helpers.OracleBackend, but every op runs in libmcl_hip.so."""

    def __init__(self, g, engine):
        self.n = int(g['n'])
        self.e = engine.Engine(self.n, init_cov=g['init_cov'], process_cov=g['motion_cov'],
                               resample_cov=g['res_cov'], meas_std=float(g['meas_std']), m2o=g['m2o'],
                               rng_mode=engine.RNG_REPLAY)
        self.last_indices = self.last_w_raw = self.last_w_norm = None

    def init(self, init_cov, normals):
        self.e.init_particles(normals)

    def predict(self, v, wz, q, z, dt, normals):
        self.e.predict(v, wz, q, z, dt, normals)

    def update_resample(self, gx, gy, rs):
        self.e.update_gps(gx, gy)
        self.last_w_raw = np.exp(self.e.get_log_weights()) + 1e-200
        self.e.resample(rs.random_sample(), rs.randn(self.n, 6))
        self.last_indices = self.e.last_indices()
        q, tot = self.e.fixed_weights()
        self.last_w_norm = q.astype(np.float64) / float(tot)

    def state(self):
        return np.ascontiguousarray(self.e.get_particles().T)

    def mean_cov(self):
        return self.e.mean_cov()


@pytest.mark.parametrize('name', ['traj_predict_launch', 'traj_predict_motion2', 'traj_gps_systematic',
                                  'traj_gps_systematic_n1000', 'traj_gps_systematic_fullcov'])
def test_trajectory_replay_vs_reference_golden(name, eng, orc):
    """Whole-filter parity with the reference node on recorded inputs + replayed RNG draws:
    states 1e-9, normalised weights rel 1e-11, resample indices exact."""
    g = helpers.load(name)
    out = helpers.replay(g, GpuBackend(g, eng))
    np.testing.assert_allclose(out['init_state'], g['init_state'], rtol=0, atol=1e-15)
    for k in range(len(g['fix_idx'])):
        assert np.array_equal(out['indices'][k], g['indices'][k]), (name, k)
        np.testing.assert_allclose(out['weights_raw'][k], g['weights_raw'][k], rtol=1e-10, atol=0)
        np.testing.assert_allclose(out['weights_norm'][k], g['weights_norm'][k], rtol=1e-10, atol=1e-15)
        np.testing.assert_allclose(out['post_update_states'][k], g['post_update_states'][k], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(out['ckpt_states']), g['ckpt_states'], rtol=0, atol=1e-9)
    mean = np.array(out['mean'])
    np.testing.assert_allclose(mean[:, :3], g['mean_xyz'], rtol=0, atol=1e-9)
    for k in range(mean.shape[0]):
        q = orc.quat_from_euler(mean[k, 3], mean[k, 4], out['yaw'][k])
        np.testing.assert_allclose(q, g['quat'][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(out['cov'][k], g['cov36'][k][:9], rtol=1e-9, atol=1e-15)


class GpuResidualBackend(GpuBackend):
    """The node as written: residual_resample (auv_pf.py:182), literal GPU restatement."""

    def __init__(self, g, engine):
        self.n = int(g['n'])
        self.e = engine.Engine(self.n, init_cov=g['init_cov'], process_cov=g['motion_cov'],
                               resample_cov=g['res_cov'], meas_std=float(g['meas_std']), m2o=g['m2o'],
                               rng_mode=engine.RNG_REPLAY, resample_scheme=engine.RESIDUAL)
        self.last_indices = self.last_w_raw = self.last_w_norm = None

    def update_resample(self, gx, gy, rs):
        self.e.update_gps(gx, gy)
        self.last_w_raw = np.exp(self.e.get_log_weights()) + 1e-200
        need = self.e.resample_prepare()          # N - k uniforms, like resampling.py:74
        self.e.resample(rs.random_sample(need), rs.randn(self.n, 6))
        self.last_indices = self.e.last_indices()


def test_trajectory_replay_residual_node_as_written(eng):
    """auv_pf.py as written calls residual_resample: the GPU compatibility mode reproduces the
    reference trajectory, indices exact, on the golden N=128 run."""
    g = helpers.load('traj_gps_residual')
    out = helpers.replay(g, GpuResidualBackend(g, eng))
    for k in range(len(g['fix_idx'])):
        assert np.array_equal(out['indices'][k], g['indices'][k]), k
        np.testing.assert_allclose(out['post_update_states'][k], g['post_update_states'][k], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(out['ckpt_states']), g['ckpt_states'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(out['mean'])[:, :3], g['mean_xyz'], rtol=0, atol=1e-9)


def test_all_resampler_kats_vs_reference(eng):
    """stratified / multinomial / residual / naive golden vectors (resampling.py) through the GPU."""
    g = helpers.load('resampling_kat')
    counts, worst, total = {}, {}, {}
    for tag in g['cases']:
        w = g[tag + '_w']
        n = w.size
        seed = int(g[tag + '_seed'])
        for name, scheme in (('stratified_resample', eng.STRATIFIED), ('multinomial_resample', eng.MULTINOMIAL),
                             ('residual_resample', eng.RESIDUAL), ('naive_resample', eng.NAIVE)):
            key = tag + '_' + name
            if key not in g:
                continue
            u = np.random.RandomState(seed).random_sample(n)  # a prefix is what the scheme consumes
            idx = eng.resample_indices(w, u, scheme=scheme)
            ref = g[key]
            if name == 'residual_resample' or n <= 4096:
                assert np.array_equal(idx, ref), key
            else:  # N = 65536: allow the reference's own fp64 cumsum rounding
                miss = int(np.count_nonzero(idx != ref))
                worst[name] = max(worst.get(name, 0), miss)
                total[name] = total.get(name, 0) + miss
                assert miss <= 2, key
            counts[name] = counts.get(name, 0) + 1
    assert min(counts.values()) >= 20, counts
    # what the bound of 2 actually costs: the observed mismatches against the reference's fp64 indices
    print('N = 65 536 cases: index mismatches vs the reference, worst case per scheme %r, summed over all cases %r' % (worst, total))
    assert sum(total.values()) <= 8, total


@pytest.mark.parametrize('scheme', ['STRATIFIED', 'MULTINOMIAL', 'RESIDUAL'])
def test_alt_resampler_reassign_matches_reference_semantics(scheme, eng, orc):
    """keep/lost/dupes for an UNSORTED ancestor vector (auv_pf.py:183-198) on the GPU == oracle."""
    n = 5000
    rs = np.random.RandomState(3)
    lw = -0.5 * (rs.randn(n) * 2.0) ** 2
    soa = rs.randn(6, n)
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY, resample_scheme=getattr(eng, scheme))
    e.set_particles(soa)
    e.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
    need = e.resample_prepare()
    assert need == (n if scheme != 'RESIDUAL' else need) and 0 <= need <= n
    e.resample(rs.random_sample(need), np.zeros((n, 6)))
    idx = e.last_indices()
    assert idx.min() >= 0 and idx.max() < n
    lost, dupes = orc.lost_dupes(idx)
    ref = soa.copy()
    orc.reassign(ref, lost, dupes)
    assert np.array_equal(e.get_particles(), ref)
    # native RNG path runs too
    e2 = eng.Engine(n, resample_scheme=getattr(eng, scheme), seed=4)
    e2.set_particles(soa)
    e2.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
    e2.resample()
    i2 = e2.last_indices()
    assert np.bincount(i2, minlength=n).sum() == n


def test_systematic_kat_vs_reference(eng):
    """resampling.py:systematic_resample golden vectors through mcl_resample_indices."""
    g = helpers.load('resampling_kat')
    n_cases = 0
    for tag in g['cases']:
        key = tag + '_systematic_resample'
        if key not in g:
            continue
        w = g[tag + '_w']
        u = np.random.RandomState(int(g[tag + '_seed'])).random_sample(1)[0]
        idx = eng.resample_indices(w, u)
        assert np.array_equal(idx, g[key]), tag
        n_cases += 1
    assert n_cases >= 30


@pytest.mark.parametrize('n', [1, 2, 7, 64, 1000, 2048, 2049, 4096, 65536, 1048576 + 3])
@pytest.mark.parametrize('mode', [0, 1])
def test_fixed_point_resample_bit_exact_vs_oracle(n, mode, eng, orc):
    """q, offspring CDF, ancestor indices and the reassigned state are bit-identical to the
    oracle's integer spec for any n (ragged tile edges included)."""
    rs = np.random.RandomState(n * 2 + mode)
    lw = -0.5 * (rs.randn(n) * 3.0) ** 2 + (2.0 if mode == 0 else -300.0)
    if n > 10:
        lw[rs.randint(0, n, size=n // 10)] = -1e4  # underflowing particles
    u = rs.random_sample()
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    soa = rs.randn(6, n)
    e.set_particles(soa)
    e.set_log_weights(lw, mode)
    e.resample(u, np.zeros((n, 6)))
    q_ref, tot_ref, _ = orc.fixed_weights(lw, mode)
    q, tot = e.fixed_weights()
    assert tot == tot_ref and np.array_equal(q, q_ref)
    ncum_ref = orc.systematic_ncum(q_ref, orc.u_to_u53(u), 0, tot_ref, n)
    assert np.array_equal(e.last_offspring_cdf(), ncum_ref)
    idx_ref = orc.indices_from_ncum(ncum_ref)
    idx = e.last_indices()
    assert np.array_equal(idx, idx_ref)
    assert int(ncum_ref[-1]) == n and np.all(np.diff(idx.astype(np.int64)) >= 0)
    # keep/lost/dupes reassign (auv_pf.py:183-198), zero noise -> exact copy semantics
    lost, dupes = orc.lost_dupes(idx_ref)
    ref = soa.copy()
    orc.reassign(ref, lost, dupes)
    assert np.array_equal(e.get_particles(), ref)
    e.close()


def test_fixed_point_matches_reference_fp64_indices(eng, orc):
    """The integer CDF reproduces the reference's fp64 sequential-cumsum indices (differences are
    possible only when a position falls within ~N*2^-53 of a CDF edge; report the count)."""
    rs = np.random.RandomState(11)
    mism = 0
    for n in [128, 4096, 65536, 1 << 20]:
        w = rs.rand(n) ** 4
        w /= w.sum()
        u = rs.random_sample()
        ref, rc = orc.systematic_ref(w, u)
        idx = eng.resample_indices(w, u)
        d = int(np.count_nonzero(idx != ref))
        mism += d
        assert d <= 2, (n, d)
    print('fixed-point vs fp64 reference index mismatches over 4 sizes:', mism)


def test_predict_native_matches_oracle(eng, orc):
    n = 5000
    pc = [1e-3, 2e-3, 0, 0, 0, 1e-4]
    e = eng.Engine(n, init_cov=[1, 2, 0.1, 0.01, 0.02, 0.3], process_cov=pc, seed=0xABCDEF0123)
    e.init_particles()
    s0 = e.get_particles()
    ref0 = np.zeros((6, n))
    orc.add_noise(ref0, [1, 2, 0.1, 0.01, 0.02, 0.3], orc.native_normals(n, 0, 0xABCDEF0123, 0, 0))
    np.testing.assert_allclose(s0, ref0, rtol=0, atol=1e-13)
    q = orc.quat_from_euler(0.05, -0.03, 1.0)
    ref = s0.copy()
    for step in range(3):
        e.predict([1.2, 0.1, -0.05], 0.07, q, -3.5, 0.02)
        orc.predict(ref, [1.2, 0.1, -0.05], 0.07, q, -3.5, 0.02, pc, orc.native_normals(n, 0, 0xABCDEF0123, 1, step))
    np.testing.assert_allclose(e.get_particles(), ref, rtol=0, atol=1e-12)


def test_native_normals_statistics(eng):
    n = 1 << 20
    e = eng.Engine(n, init_cov=[1, 1, 1, 1, 1, 1], seed=7)
    e.init_particles()
    s = e.get_particles()
    assert np.all(np.abs(s.mean(axis=1)) < 5e-3)
    assert np.all(np.abs(s.var(axis=1) - 1.0) < 1e-2)
    c = np.corrcoef(s)
    assert np.all(np.abs(c - np.eye(6)) < 5e-3)


def test_gps_weights_match_oracle(eng, orc):
    n = 4097
    rs = np.random.RandomState(2)
    from smarc_navigation_amd import synth
    m2o = synth.rigid_matrix(3, -4, 0.5, 0.01, -0.02, 1.1)
    e = eng.Engine(n, meas_std=1.7, m2o=m2o, rng_mode=eng.RNG_REPLAY)
    soa = rs.randn(6, n) * np.array([[5], [5], [1], [0.1], [0.1], [1]])
    e.set_particles(soa)
    e.update_gps(1.5, -2.0)
    w, lw = orc.gps_weights(soa, m2o, 1.5, -2.0, 1.7)
    np.testing.assert_allclose(e.get_log_weights(), lw, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(np.exp(e.get_log_weights()), w, rtol=1e-12)


def test_mean_cov_and_poses_match_oracle(eng, orc):
    n = 100003
    rs = np.random.RandomState(4)
    soa = rs.randn(6, n) * np.array([[30], [20], [2], [0.2], [0.2], [4]]) + np.array([[100], [-50], [-5], [0], [0], [0]])
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    mean, yaw, cov = e.mean_cov()
    m_ref, y_ref, c_ref = orc.mean_cov(soa)
    np.testing.assert_allclose(mean, m_ref, rtol=1e-12, atol=1e-12)
    assert abs(yaw - y_ref) < 1e-12
    np.testing.assert_allclose(cov, c_ref, rtol=1e-10, atol=1e-12)
    assert cov[6] == 0.0 and cov[7] == 0.0 and cov[3] == cov[1]
    poses = e.poses()
    np.testing.assert_array_equal(poses[:, :3], soa[:3].T)
    for i in [0, 1, 777, n - 1]:
        np.testing.assert_allclose(poses[i, 3:], orc.quat_from_euler(soa[3, i], soa[4, i], soa[5, i]), atol=1e-14)


@pytest.mark.parametrize('exchange', ['p2p', 'allgather'])
@pytest.mark.parametrize('shards', [2, 4, 8])
def test_sharded_equals_unsharded_bitwise(shards, exchange, eng, monkeypatch):
    """particle shards + exchange steps give bit-identical states and indices -- with the O(n)-per-rank
    exchange (every shard expands its own CDF slice, only surplus copies that fill a PEER's lost slots move) and with
    the all-gather exchange of rounds 1-2 (MCL_EXCHANGE=allgather)."""
    if exchange == 'allgather':
        monkeypatch.setenv('MCL_EXCHANGE', 'allgather')
    else:
        monkeypatch.delenv('MCL_EXCHANGE', raising=False)
    n = 8192 * shards
    cov = dict(init_cov=[2, 2, 0, 0, 0, 0.05], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0, 0, 0, 1e-4], meas_std=2.0, seed=99)
    one = eng.Engine(n, **cov)
    many = [eng.Engine(n // shards, rank=r, world=shards, n_global=n, global_offset=r * (n // shards), **cov)
            for r in range(shards)]
    for e in [one] + many:
        e.init_particles()
    from oracle import oracle as orc
    q = orc.quat_from_euler(0.01, 0.02, 0.3)
    for step in range(3):
        for e in [one] + many:
            e.predict([1.0, 0.05, 0.0], 0.02, q, -2.0, 0.02)
            e.update_gps(0.1 * step, -0.05 * step)
        one.resample()
        eng.group_resample(many)
        full = one.get_particles()
        parts = np.concatenate([e.get_particles() for e in many], axis=1)
        assert np.array_equal(full, parts), step
        assert np.array_equal(one.last_indices(), np.concatenate([e.last_indices() for e in many]))
        assert np.array_equal(one.last_offspring_cdf(), many[0].last_offspring_cdf())
        assert np.array_equal(one.last_offspring_cdf(), many[-1].last_offspring_cdf())
    if exchange == 'p2p':
        sent = sum(e.exchange_stats()[0] for e in many)
        lost = sum(e.exchange_stats()[1] for e in many)
        print('%d shards: %d of %d copied particles crossed a shard border' % (shards, sent, lost))
        assert 0 < sent < lost   # something moved between shards, most copies stayed at home
    m1 = one.mean_cov()
    m2 = eng.group_mean_cov(many)
    np.testing.assert_allclose(m1[0], m2[0], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(m1[2], m2[2], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('where', ['one_shard', 'last_particle', 'every_other_shard'])
def test_sharded_exchange_when_the_weight_sits_in_few_places(where, eng, monkeypatch):
    """The O(n) exchange's extremes: all the weight in one shard (it ships copies to every other shard, far more than
    the send buffer's initial capacity: the grow-and-pack-again path), in ONE particle (every slot but one is lost,
    one ancestor owns every surplus copy), or in alternate shards (shards with no lost slot at all next to shards
    with no survivor).  Bit for bit the unsharded filter, twice in a row (the buffers are reused)."""
    monkeypatch.delenv('MCL_EXCHANGE', raising=False)
    shards, n = 4, 4 * 8192
    nl = n // shards
    cov = dict(resample_cov=[0.01, 0.01, 0.0025, 1e-4, 1e-4, 1e-4], seed=31)
    rs = np.random.RandomState(5)
    soa = rs.randn(6, n)
    one = eng.Engine(n, **cov)
    many = [eng.Engine(nl, rank=r, world=shards, n_global=n, global_offset=r * nl, **cov) for r in range(shards)]
    for rep in range(2):
        lw = np.full(n, -800.0)
        if where == 'one_shard':
            lw[2 * nl:2 * nl + 100] = -0.5 * rs.randn(100) ** 2
        elif where == 'last_particle':
            lw[n - 1] = 0.0
        else:
            lw[0:nl] = -0.5 * rs.randn(nl) ** 2
            lw[2 * nl:3 * nl] = -0.5 * rs.randn(nl) ** 2
        one.set_particles(soa)
        one.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
        one.resample()
        for r, e in enumerate(many):
            sl = slice(r * nl, (r + 1) * nl)
            e.set_particles(np.ascontiguousarray(soa[:, sl]))
            e.set_log_weights(lw[sl], eng.WEIGHT_LOG_SHIFT)
        eng.group_resample(many)
        assert np.array_equal(one.last_indices(), np.concatenate([e.last_indices() for e in many])), (where, rep)
        got = np.concatenate([e.get_particles() for e in many], axis=1)
        assert np.array_equal(one.get_particles(), got), (where, rep)
        soa = got + 0.0
    sent = [e.exchange_stats()[0] for e in many]
    if where == 'one_shard':
        assert sent[2] > 2 * 3 * nl * 0.9 and sent[0] == sent[1] == sent[3] == 0   # shard 2 fed all the others, twice
    if where == 'last_particle':
        assert sent[3] == 2 * 3 * nl and sum(sent[:3]) == 0


def test_rccl_paths_with_one_rank_match_plain_filter(eng, monkeypatch):
    """MCL_FORCE_COMM=1 builds a 1-rank RCCL communicator: all-reduce / all-gather calls, the second
    communicator and the overlapped state all-gather of mcl_step_mbes run for real, and must not
    change a single bit of the result."""
    from smarc_navigation_amd import synth
    from oracle import oracle as orc
    n, B = 16384, 64
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    cov = dict(init_cov=[1, 1, 0, 0, 0, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0, 0, 0, 1e-4], seed=21)
    q = orc.quat_from_euler(0.0, 0.0, 0.1)
    ranges = np.full(B, 21.0, np.float32)
    results = []
    for force, no_overlap, exchange in (('0', '0', 'p2p'), ('1', '0', 'p2p'), ('1', '0', 'allgather'), ('1', '1', 'allgather')):
        monkeypatch.setenv('MCL_FORCE_COMM', force)
        monkeypatch.setenv('MCL_NO_OVERLAP', no_overlap)
        monkeypatch.setenv('MCL_EXCHANGE', exchange)
        e = eng.Engine(n, **cov)
        e.comm_init(eng.comm_unique_id())
        # the communicator really exists (an all-reduce(sum) of 1 counts its ranks), the second one only
        # with the all-gather exchange and the overlap on; the deadline self-test runs the step's collective pattern
        ranks, overlap = e.comm_ranks()
        assert ranks == 1 and overlap == (force == '1' and no_overlap == '0' and exchange == 'allgather')
        e.comm_selftest(20000)
        e.set_map_grid(z, origin, 1.0)
        e.init_particles()
        for k in range(4):
            e.step_mbes([1.0, 0.0, 0.0], 0.02, q, -2.0, 0.02, ranges, ba, 0.5, 80.0)
        e.sync()
        results.append((e.get_particles(), e.last_mean_cov()[0]))
        e.close()
    for st, mean in results[1:]:
        assert np.array_equal(st, results[0][0])
        np.testing.assert_allclose(mean, results[0][1], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize('fused', [True, False])
def test_state_exchange_leaves_out_the_components_predict_made_uniform(fused, eng, monkeypatch):
    """After motion_pred every particle holds the odometry's depth, roll and pitch, so the pre-resample state
    exchange only gathers x, y, yaw (24 B instead of 48 B per particle of the global cloud) and the gather kernel
    substitutes the three constants -- bit for bit the plain filter, with resampling noise on ALL six components,
    a rolled and pitched vehicle, and a resample that does NOT follow a predict (set_particles: full exchange)."""
    from smarc_navigation_amd import synth
    from oracle import oracle as orc
    n, B = 16384, 64
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    cov = dict(init_cov=[1, 1, 0.04, 0.0025, 0.0025, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0.0025, 1e-4, 1e-4, 1e-4], seed=22)
    q = orc.quat_from_euler(0.03, -0.05, 0.1)
    ranges = np.full(B, 21.0, np.float32)
    results = []
    for force, no_overlap, exchange in (('0', '0', 'p2p'), ('1', '0', 'p2p'), ('1', '0', 'allgather'), ('1', '1', 'allgather')):
        monkeypatch.setenv('MCL_FORCE_COMM', force)
        monkeypatch.setenv('MCL_NO_OVERLAP', no_overlap)
        monkeypatch.setenv('MCL_EXCHANGE', exchange)
        e = eng.Engine(n, **cov)
        e.comm_init(eng.comm_unique_id())
        e.set_map_grid(z, origin, 1.0)
        e.init_particles()
        for k in range(3):
            if fused:
                e.step_mbes([1.0, 0.0, 0.0], 0.02, q, -2.0 - 0.1 * k, 0.02, ranges, ba, 0.5, 80.0)
            else:
                e.predict([1.0, 0.0, 0.0], 0.02, q, -2.0 - 0.1 * k, 0.02)
                e.update_mbes(ranges, ba, 0.5, 80.0)
                e.resample()
        st1 = e.get_particles()
        # a resample that does not follow a predict: nothing is uniform, everything is exchanged
        soa = st1.copy()
        soa[2] += np.linspace(0.0, 1.0, n)
        e.set_particles(soa)
        e.update_mbes(ranges, ba, 0.5, 80.0)
        e.resample()
        results.append((st1, e.get_particles()))
        e.close()
    for st1, st2 in results[1:]:
        assert np.array_equal(st1, results[0][0])
        assert np.array_equal(st2, results[0][1])
    assert np.ptp(results[0][1][2]) > 0.2   # the second resample really carried distinct depths


def test_comm_shutdown_and_reinit_without_overlap(eng, monkeypatch):
    """The fall-back bench.py takes when the overlap self-test fails: abort both communicators,
    re-initialise under a fresh id with MCL_COMM_NO_OVERLAP, same results.  Also: a step that fails after
    the overlapped gather was started (no map) must not leave a stale gather behind."""
    from smarc_navigation_amd import synth
    from oracle import oracle as orc
    monkeypatch.setenv('MCL_FORCE_COMM', '1')
    monkeypatch.setenv('MCL_EXCHANGE', 'allgather')   # (the exchange that has a second communicator to lose)
    n, B = 8192, 64
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    cov = dict(init_cov=[1, 1, 0, 0, 0, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0, 0, 0, 1e-4], seed=22)
    q = orc.quat_from_euler(0.0, 0.0, 0.1)
    ranges = np.full(B, 21.0, np.float32)
    out = []
    for reinit in (False, True):
        e = eng.Engine(n, **cov)
        e.comm_init(eng.comm_unique_id())
        if reinit:
            e.comm_shutdown(abort=True)
            assert e.comm_ranks() == (1, False)
            e.comm_init(eng.comm_unique_id(), overlap=False)
            assert e.comm_ranks() == (1, False)
        e.init_particles()
        with pytest.raises(eng.MclError):  # no map yet: refused before anything is queued
            e.step_mbes([1.0, 0.0, 0.0], 0.02, q, -2.0, 0.02, ranges, ba, 0.5, 80.0)
        e.set_map_grid(z, origin, 1.0)
        for k in range(3):
            e.step_mbes([1.0, 0.0, 0.0], 0.02, q, -2.0, 0.02, ranges, ba, 0.5, 80.0)
        e.sync()
        out.append(e.get_particles())
        e.close()
    assert np.array_equal(out[0], out[1])


def test_rccl_single_rank_world1_smoke(eng):
    """The RCCL entry points work (world == 1 is a no-op communicator)."""
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    e = eng.Engine(1024)
    e.comm_init(uid)
    e.init_particles()
    e.update_gps(0, 0)
    e.resample()


def test_fused_step_equals_separate_calls_bitwise(eng):
    """mcl_step_mbes (predict + pose records in one kernel, max lw from the cast epilogue, the sums of
    update_loc_pose inside the resample gather) against the same step made of the separate ABI calls:
    particles bit for bit, mean / covariance to rounding (one-pass shifted sums vs two passes)."""
    from smarc_navigation_amd import synth
    from oracle import oracle as orc
    n, B = 30000, 96
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    cov = dict(init_cov=[1, 1, 0, 0, 0, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0, 0, 0, 1e-4], seed=31)
    m2o = synth.rigid_matrix(0.5, -1.0, 0.0, 0.0, 0.0, 0.1)
    off = [0.2, 0.0, -0.1, 0.0, 0.01, 0.02]
    stream = synth.odom_stream(5)
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY, m2o=m2o)
    one.set_map_grid(z, origin, 1.0)
    a, b = eng.Engine(n, m2o=m2o, **cov), eng.Engine(n, m2o=m2o, **cov)
    for e in (a, b):
        e.set_map_grid(z, origin, 1.0)
        e.init_particles()
    for k in range(4):
        one.set_particles(stream['truth'][k][:, None].copy())
        ranges = one.mbes_expected(0, 1, ba, 80.0, off)[0]
        a.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges, ba, 0.3, 80.0, off)
        ma = a.last_mean_cov()
        b.predict(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
        b.update_mbes(ranges, ba, 0.3, 80.0, off)
        b.resample()
        mb = b.mean_cov()
        assert np.array_equal(a.get_particles(), b.get_particles()), k
        assert np.array_equal(a.last_indices(), b.last_indices())
        np.testing.assert_allclose(ma[0], mb[0], rtol=0, atol=1e-12)
        assert abs(ma[1] - mb[1]) <= 1e-12
        np.testing.assert_allclose(ma[2], mb[2], rtol=1e-9, atol=1e-15)
        # and both are the oracle's mean/cov of that state
        m6, yaw, c9 = orc.mean_cov(a.get_particles())
        np.testing.assert_allclose(ma[0], m6, rtol=0, atol=1e-10)
        np.testing.assert_allclose(ma[2], c9, rtol=1e-8, atol=1e-15)


def test_fused_step_that_fails_after_predict_leaves_a_complete_state(eng, monkeypatch):
    """The fused step's predict kernel does not store z, roll, pitch (the gather of the same call substitutes them);
    a step that leaves between the two must store them itself: the state then equals a plain mcl_predict's."""
    from smarc_navigation_amd import synth
    n, B = 20000, 64
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=3)
    ba = synth.beam_angles(B)
    cov = dict(init_cov=[1, 1, 0, 0, 0, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.01, 0, 0, 0, 1e-4], seed=31)
    stream = synth.odom_stream(2)
    ranges = np.full(B, 30.0, np.float32)
    b = eng.Engine(n, **cov)
    monkeypatch.setenv('MCL_FAULT_INJECT', 'step_after_predict')
    a = eng.Engine(n, **cov)
    shards = [eng.Engine(n // 2, rank=r, world=2, n_global=n, global_offset=r * (n // 2), **cov) for r in range(2)]
    monkeypatch.delenv('MCL_FAULT_INJECT')
    for e in [a, b] + shards:
        e.set_map_grid(z, origin, 1.0)
        e.init_particles()
    args = (stream['v'][0], stream['wz'][0], stream['q'][0], stream['z'][0], stream['dt'])
    with pytest.raises(eng.MclError, match='injected fault'):
        a.step_mbes(*args, ranges, ba, 0.3, 80.0)
    with pytest.raises(eng.MclError, match='injected fault'):
        eng.group_step_mbes(shards, *args, ranges, ba, 0.3, 80.0)
    b.predict(*args)
    want = b.get_particles()
    assert np.array_equal(a.get_particles(), want)
    assert np.array_equal(np.concatenate([e.get_particles() for e in shards], axis=1)[:, :n // 2], want[:, :n // 2])
    assert np.all(want[2] == stream['z'][0])
    # and the handle goes on working: the update + resample of that state
    a.update_mbes(ranges, ba, 0.3, 80.0)
    a.resample()
    b.update_mbes(ranges, ba, 0.3, 80.0)
    b.resample()
    assert np.array_equal(a.get_particles(), b.get_particles())


@pytest.mark.parametrize('n,heavy,where', [(65536, 150, 'first_tile'), (65536, 1, 'one'), (300000, 40, 'spread'),
                                           (2048 * 3 + 5, 2, 'tile_edges')])
def test_expansion_of_heavy_ancestors_matches_list_semantics(eng, n, heavy, where):
    """Ancestors with many copies: more than EXP_HEAVY surplus copies go through the block-cooperative list,
    more than EXP_LIST of them in one tile overflow it; one particle holding all the weight; heavy
    ancestors sitting on tile borders.  keep/lost/dupes must equal the reference's list semantics."""
    from oracle import oracle as orc
    rs = np.random.RandomState(5)
    lw = np.full(n, -800.0)
    if where == 'first_tile':
        idx = np.arange(heavy) * 3
    elif where == 'one':
        idx = np.array([n // 2 + 17])
    elif where == 'tile_edges':
        idx = np.array([2047, 2048])
    else:
        idx = np.sort(rs.choice(n, heavy, replace=False))
    lw[idx] = rs.uniform(-1.0, 0.0, idx.size)
    soa = rs.randn(6, n)
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
    u = 0.37
    e.resample(uniforms=[u], normals=np.zeros((n, 6)))
    ref_idx, _, _ = orc.systematic_fixed(lw, 1, orc.u_to_u53(u))
    assert np.array_equal(e.last_indices(), ref_idx)
    lost, dupes = orc.lost_dupes(ref_idx)
    assert lost.size >= n - idx.size - 1
    ref = soa.copy()
    orc.reassign(ref, lost, dupes)
    assert np.array_equal(e.get_particles(), ref)


def test_native_normals_are_bit_identical_to_the_oracle(eng, orc):
    """The Box-Muller draws use deterministic log / sincos (fixed fma sequences, mcl_device.h == mcl_oracle.c):
    init noise (all six components), predict noise and resample noise agree with the oracle BIT FOR BIT."""
    n = 50000
    cov = dict(init_cov=[1.0, 4.0, 0.25, 0.0625, 0.015625, 16.0], process_cov=[1e-3, 2e-3, 0, 0, 0, 1e-5],
               resample_cov=[0.01, 0.02, 0.03, 1e-4, 2e-4, 1e-4], seed=77)
    e = eng.Engine(n, **cov)
    e.init_particles()
    ref = np.zeros((6, n))
    orc.add_noise(ref, cov['init_cov'], orc.native_normals(n, 0, 77, 0, 0))
    assert np.array_equal(e.get_particles(), ref)  # sqrt(cov) are powers of two: state = draw * 2^k exactly
    # resample noise on a state that survives unchanged (uniform weights: every slot keeps its particle)
    e.set_log_weights(np.zeros(n), eng.WEIGHT_LOG_SHIFT)
    e.resample()
    assert np.array_equal(e.last_indices(), np.arange(n))
    z = orc.native_normals(n, 0, 77, 2, 0)
    # (the kernel adds the noise with one fma, the oracle with a multiply and an add: last-bit difference; the
    #  DRAWS themselves are bit-identical -- recover them from the state and compare where sqrt(cov) is a power of two)
    got = e.get_particles()
    orc.add_noise(ref, cov['resample_cov'], z)
    np.testing.assert_allclose(got, ref, rtol=4e-16, atol=2e-16)
