"""BASELINE configs 4 and 5 at FULL size on one GPU: 4 194 304 particles as 8 MCL_COMM_LOCAL shards of 524 288 --
phase for phase what 8 ranks under RCCL execute (fused predict + fan sweep, max / totals / hand-over records, the
O(n)-per-rank dupes exchange, gather + moments) -- against the unsharded 4 M filter, bit for bit, with spot checks of
the log-likelihoods against the fp64 oracle.  Plus the metric's own configuration (1 M x 512, fused step) as a
trajectory against the oracle with the SAME Philox draws, within the bound stated in BASELINE.md 4."""
import numpy as np
import pytest

from smarc_navigation_amd import synth

pytestmark = pytest.mark.gpu

SHARDS, NS, B = 8, 524288, 512
N = SHARDS * NS
COV = dict(init_cov=[2.0, 2.0, 0.0, 0.0, 0.0, 0.05], process_cov=[1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6],
           resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])
SIGMA, R_MAX = 0.2, 100.0
# stated bound on the deviation of the filter's mean pose from the oracle's (same draws), per step: the two filters
# weigh with fp32 / fp64 ray-casts, so resampling decisions differ on a handful of particles near CDF edges
TRAJ_TOL_XY_M, TRAJ_TOL_YAW_RAD = 5e-3, 1e-3


def _mesh():
    from oracle import oracle as orc
    origin = (-64.0, -354.0)
    z = synth.bathymetry_grid(708, 708, 1.0, origin, seed=3)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    return verts, tris, orc.Mesh(verts, tris)


def _ranges(eng, verts, tris, truth, ba, seed):
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    one.set_map_mesh(verts, tris)
    rs = np.random.RandomState(seed)
    out = np.zeros((len(truth), ba.size), np.float32)
    for k in range(len(truth)):
        one.set_particles(truth[k][:, None].copy())
        out[k] = one.mbes_expected(0, 1, ba, R_MAX)[0] + SIGMA * rs.randn(ba.size)
    one.close()
    return out


def _lw_ok(lw, lw_ref):
    d = np.abs(lw - lw_ref)
    return (d <= 1e-2) | (d <= 2e-4 * np.abs(lw_ref))


@pytest.mark.parametrize('exchange', ['p2p', 'allgather'])
def test_config4_eight_local_shards_of_524288_fused_steps_bitwise(exchange, monkeypatch):
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    monkeypatch.setenv('MCL_EXCHANGE', exchange)
    monkeypatch.delenv('MCL_SWEEP', raising=False)
    verts, tris, omap = _mesh()
    steps = 3 if exchange == 'p2p' else 2
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    ranges = _ranges(eng, verts, tris, stream['truth'], ba, 4)
    one = eng.Engine(N, seed=5, **COV)
    many = [eng.Engine(NS, rank=r, world=SHARDS, n_global=N, global_offset=r * NS, seed=5, **COV) for r in range(SHARDS)]
    for e in [one] + many:
        e.set_map_mesh(verts, tris)
        e.init_particles()
    pick = np.random.RandomState(3).choice(N, 4096, replace=False)
    for k in range(steps):
        args = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, SIGMA, R_MAX)
        one.step_mbes(*args)
        eng.group_step_mbes(many, *args)
        assert one.mbes_last_path()[0] == 1 and all(e.mbes_last_path()[0] == 1 for e in many)   # the fan sweep
        lw1 = one.get_log_weights()
        lwm = np.concatenate([e.get_log_weights() for e in many])
        assert np.array_equal(lw1, lwm), k
        idx1 = one.last_indices()
        idxm = np.concatenate([e.last_indices() for e in many])
        assert np.array_equal(idx1, idxm), k
        st1 = one.get_particles()
        stm = np.concatenate([e.get_particles() for e in many], axis=1)
        assert np.array_equal(st1, stm), k
        m1, y1, c1 = one.last_mean_cov()
        mm, ym, cm = many[0].last_mean_cov()
        np.testing.assert_allclose(mm, m1, rtol=0, atol=1e-10)
        np.testing.assert_allclose(cm, c1, rtol=1e-9, atol=1e-12)
        if k == 0:
            # spot check against the oracle: the pre-resample state of step 0 is init + predict, reproducible on the CPU
            soa = np.zeros((6, N))
            orc.add_noise(soa, COV['init_cov'], orc.native_normals(N, 0, 5, 0, 0))
            orc.predict(soa, stream['v'][0], stream['wz'][0], stream['q'][0], stream['z'][0], stream['dt'],
                        COV['process_cov'], orc.native_normals(N, 0, 5, 1, 0))
            sub = np.ascontiguousarray(soa[:, pick])
            lw_ref, _ = orc.mbes_update(sub, np.identity(4), [0] * 6, omap, ba, ranges[0], SIGMA, R_MAX)
            ok = _lw_ok(lw1[pick], lw_ref)
            print('config 4 (%s): 4096 of %d log-likelihoods vs oracle: max |d| %.3e, outside tolerance %d' % (
                exchange, N, np.abs(lw1[pick] - lw_ref).max(), int((~ok).sum())))
            assert (~ok).sum() <= 8
            from tests.helpers import lw_outliers_explained
            lw_outliers_explained(orc, omap, sub, ba, ranges[0], SIGMA, R_MAX, lw1[pick], lw_ref, label='config 4')
            # the contract where it bites: every particle that can receive offspring within |d lw| <= 1e-2 ABSOLUTE
            from tests.helpers import live_particle_contract, live_picks
            live = live_picks(lw1, 2048, seed=8)
            lsub = np.ascontiguousarray(soa[:, live])
            lw_live, _ = orc.mbes_update(lsub, np.identity(4), [0] * 6, omap, ba, ranges[0], SIGMA, R_MAX)
            n_live, _, _ = live_particle_contract(orc, omap, lsub, ba, ranges[0], SIGMA, R_MAX, lw1[live], lw_live, float(lw1.max()),
                                                  label='config 4 (4 M x 512)')
            assert n_live >= 16
            # indices are the exact systematic resample of the GPU's own log-weights
            ref_idx, _, _ = orc.systematic_fixed(lw1, 1, orc.native_u53(5, 0))
            assert np.array_equal(idx1, ref_idx)
    if exchange == 'p2p':
        sent = sum(e.exchange_stats()[0] for e in many)
        lost = sum(e.exchange_stats()[1] for e in many)
        print('config 4: %d of %d copied particles crossed a shard border (%.2f %% of the cloud per step)' % (
            sent, lost, 100.0 * sent / (steps * N)))
        assert sent < lost
        # one send and one receive per peer and exchange at most (a copy travels as one record)
        for e in many:
            ops, rounds = e.exchange_ops()
            assert rounds == steps and ops <= 2 * (len(many) - 1) * rounds, (ops, rounds)


def test_config5_eight_local_shards_with_landmark_knn_bitwise():
    """config 4 + per-particle landmark k-NN association: 16 detections x 4 096 landmarks, k = 4, chi-square gate
    11.345, accumulated onto the MBES log-likelihood of the same ping."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    verts, tris, omap = _mesh()
    steps = 2
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    ranges = _ranges(eng, verts, tris, stream['truth'], ba, 4)
    lm = synth.landmark_map(4096, (-64.0, -354.0, 643.0, 353.0))
    rs = np.random.RandomState(8)
    dets = []
    for k in range(steps):
        t = stream['truth'][k]
        T = synth.rigid_matrix(*t)
        near = lm[np.argsort(np.sum((lm[:, :2] - t[:2]) ** 2, axis=1))[:16]]
        dets.append((near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(16, 3))
    one = eng.Engine(N, seed=5, **COV)
    many = [eng.Engine(NS, rank=r, world=SHARDS, n_global=N, global_offset=r * NS, seed=5, **COV) for r in range(SHARDS)]
    for e in [one] + many:
        e.set_map_mesh(verts, tris)
        e.set_landmarks(lm)
        e.init_particles()
    pick = np.random.RandomState(3).choice(N, 2048, replace=False)
    for k in range(steps):
        for e in [one] + many:
            e.predict(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
            e.update_mbes(ranges[k], ba, SIGMA, R_MAX)
        lw_mbes = one.get_log_weights()
        pre = one.get_particles()
        for e in [one] + many:
            e.update_landmarks(dets[k], 0.3, k=4, gate=11.345, accumulate=True)
        lw1 = one.get_log_weights()
        assert np.array_equal(lw1, np.concatenate([e.get_log_weights() for e in many])), k
        # the landmark term against the brute-force oracle (all 4 096 landmarks per detection) on a sample
        sub = np.ascontiguousarray(pre[:, pick])
        ref = orc.landmark_update(sub, np.identity(4), [0] * 6, lm, dets[k], 0.3, 4, 11.345)
        np.testing.assert_allclose((lw1 - lw_mbes)[pick], ref, rtol=1e-9, atol=1e-7)
        assert np.std(ref) > 0.5   # the detections discriminate between particles
        one.resample()
        eng.group_resample(many)
        assert np.array_equal(one.last_indices(), np.concatenate([e.last_indices() for e in many])), k
        assert np.array_equal(one.get_particles(), np.concatenate([e.get_particles() for e in many], axis=1)), k
    np.testing.assert_allclose(eng.group_mean_cov(many)[0], one.mean_cov()[0], rtol=0, atol=1e-10)


def test_config5_fused_step_equals_the_separate_calls_on_one_gpu_and_on_eight_shards():
    """mcl_step_mbes_landmarks (one launch sequence: predict + pose records, fan sweep, landmark k-NN on top, resample
    + moments) against mcl_predict + mcl_update_mbes + mcl_update_landmarks(accumulate) + mcl_resample, and against
    mcl_group_step_mbes_landmarks over 8 shards of 524 288: log-weights, ancestor indices and particles bit for bit."""
    from smarc_navigation_amd import engine as eng
    verts, tris, _ = _mesh()
    steps = 3
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    ranges = _ranges(eng, verts, tris, stream['truth'], ba, 4)
    lm = synth.landmark_map(4096, (-64.0, -354.0, 643.0, 353.0))
    rs = np.random.RandomState(8)
    dets = []
    for k in range(steps):
        t = stream['truth'][k]
        T = synth.rigid_matrix(*t)
        near = lm[np.argsort(np.sum((lm[:, :2] - t[:2]) ** 2, axis=1))[:16]]
        dets.append((near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(16, 3))
    sep = eng.Engine(N, seed=5, **COV)
    one = eng.Engine(N, seed=5, **COV)
    many = [eng.Engine(NS, rank=r, world=SHARDS, n_global=N, global_offset=r * NS, seed=5, **COV) for r in range(SHARDS)]
    for e in [sep, one] + many:
        e.set_map_mesh(verts, tris)
        e.set_landmarks(lm)
        e.init_particles()
    for k in range(steps):
        od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
        sep.predict(*od)
        sep.update_mbes(ranges[k], ba, SIGMA, R_MAX)
        lw_mbes = sep.get_log_weights()
        sep.update_landmarks(dets[k], 0.3, k=4, gate=11.345, accumulate=True)
        lw_sep = sep.get_log_weights()
        assert np.std(lw_sep - lw_mbes) > 0.5   # the landmark term is there and discriminates
        sep.resample()
        args = od + (ranges[k], ba, SIGMA, R_MAX, dets[k], 0.3)
        one.step_mbes_landmarks(*args, k=4, gate=11.345)
        eng.group_step_mbes_landmarks(many, *args, k=4, gate=11.345)
        assert one.mbes_last_path()[0] == 1 and all(e.mbes_last_path()[0] == 1 for e in many)   # the fan sweep
        lw1 = one.get_log_weights()
        assert np.array_equal(lw1, lw_sep), k
        assert np.array_equal(lw1, np.concatenate([e.get_log_weights() for e in many])), k
        idx1 = one.last_indices()
        assert np.array_equal(idx1, sep.last_indices()), k
        assert np.array_equal(idx1, np.concatenate([e.last_indices() for e in many])), k
        st1 = one.get_particles()
        assert np.array_equal(st1, sep.get_particles()), k
        assert np.array_equal(st1, np.concatenate([e.get_particles() for e in many], axis=1)), k
        m1, y1, c1 = one.last_mean_cov()
        ms, ys, cs = sep.mean_cov()
        mm, ym, cm = many[0].last_mean_cov()
        np.testing.assert_allclose(m1, ms, rtol=0, atol=1e-10)
        np.testing.assert_allclose(c1, cs, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(mm, m1, rtol=0, atol=1e-10)
        np.testing.assert_allclose(cm, c1, rtol=1e-9, atol=1e-12)
    # the landmark kernel has its own timing region (ABI 4)
    one.timing_enable(True)
    one.step_mbes_landmarks(*args, k=4, gate=11.345)
    one.sync()
    tim = one.timing_get()
    one.timing_enable(False)
    assert tim['update_landmarks'][1] == 1 and tim['update_landmarks'][0] > 0.0
    assert tim['normalise'][0] < 0.05   # no max-lw reduction pass: the kernel left the maximum in the slots (ms)


def test_metric_config_fused_step_trajectory_vs_oracle():
    """1 048 576 particles x 512 beams on the 999 698-triangle mesh, mcl_step_mbes x 5 against the oracle filter
    with the same Philox draws (the 'pose RMSE vs ref' of the metric at the metric's size).  Bound: BASELINE.md 4."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    from tests.helpers import host_threads
    orc.set_threads(host_threads())
    n, steps = 1 << 20, 5
    verts, tris, omap = _mesh()
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    ranges = _ranges(eng, verts, tris, stream['truth'], ba, 4)
    e = eng.Engine(n, seed=5, **COV)
    e.set_map_mesh(verts, tris)
    e.init_particles()
    for k in range(steps):
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, SIGMA, R_MAX)
    e.sync()
    assert e.mbes_last_path()[:2] == (1, 0)
    got = e.mean_history(steps)
    soa = np.zeros((6, n))
    orc.add_noise(soa, COV['init_cov'], orc.native_normals(n, 0, 5, 0, 0))
    ref = np.zeros((steps, 6))
    for k in range(steps):
        orc.predict(soa, stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                    COV['process_cov'], orc.native_normals(n, 0, 5, 1, k))
        lw, _ = orc.mbes_update(soa, np.identity(4), [0] * 6, omap, ba, ranges[k], SIGMA, R_MAX, want_expected=False)
        idx, _, _ = orc.systematic_fixed(lw, 1, orc.native_u53(5, k))
        lost, dupes = orc.lost_dupes(idx)
        orc.reassign(soa, lost, dupes)
        orc.add_noise(soa, COV['resample_cov'], orc.native_normals(n, 0, 5, 2, k))
        ref[k] = orc.mean_cov(soa)[0]
    dxy = np.hypot(got[:, 0] - ref[:, 0], got[:, 1] - ref[:, 1])
    dyaw = np.abs(got[:, 5] - ref[:, 5])
    print('1 M x 512 fused step vs oracle, %d steps: mean-pose deviation max %.3e m (rms %.3e m), yaw max %.3e rad' % (
        steps, dxy.max(), np.sqrt(np.mean(dxy ** 2)), dyaw.max()))
    assert dxy.max() <= TRAJ_TOL_XY_M and dyaw.max() <= TRAJ_TOL_YAW_RAD
    np.testing.assert_allclose(got[:, 2:5], ref[:, 2:5], rtol=0, atol=1e-9)   # z, roll, pitch: the odometry's


def test_grid_fused_step_trajectory_vs_oracle():
    """BASELINE config 2's map kind at size: 262 144 particles x 256 beams on the 708 x 708 height GRID (bilinear
    patches: the cell-walk sweep), mcl_step_mbes x 4 against the oracle filter with the same Philox draws.  Same bound
    as the mesh."""
    from smarc_navigation_amd import engine as eng
    from oracle import oracle as orc
    from tests.helpers import host_threads
    orc.set_threads(host_threads())
    n, steps, Bg = 1 << 18, 4, 256
    origin = (-64.0, -354.0)
    z = synth.bathymetry_grid(708, 708, 1.0, origin, seed=3)
    omap = orc.Grid(z, origin, 1.0)
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(Bg)
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    one.set_map_grid(z, origin, 1.0)
    rs = np.random.RandomState(4)
    ranges = np.zeros((steps, Bg), np.float32)
    for k in range(steps):
        one.set_particles(stream['truth'][k][:, None].copy())
        ranges[k] = one.mbes_expected(0, 1, ba, R_MAX)[0] + SIGMA * rs.randn(Bg)
    one.close()
    e = eng.Engine(n, seed=5, **COV)
    e.set_map_grid(z, origin, 1.0)
    e.init_particles()
    for k in range(steps):
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba, SIGMA, R_MAX)
    e.sync()
    assert e.mbes_last_path()[:2] == (1, 0)
    got = e.mean_history(steps)
    soa = np.zeros((6, n))
    orc.add_noise(soa, COV['init_cov'], orc.native_normals(n, 0, 5, 0, 0))
    ref = np.zeros((steps, 6))
    for k in range(steps):
        orc.predict(soa, stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                    COV['process_cov'], orc.native_normals(n, 0, 5, 1, k))
        lw, _ = orc.mbes_update(soa, np.identity(4), [0] * 6, omap, ba, ranges[k], SIGMA, R_MAX, want_expected=False)
        idx, _, _ = orc.systematic_fixed(lw, 1, orc.native_u53(5, k))
        lost, dupes = orc.lost_dupes(idx)
        orc.reassign(soa, lost, dupes)
        orc.add_noise(soa, COV['resample_cov'], orc.native_normals(n, 0, 5, 2, k))
        ref[k] = orc.mean_cov(soa)[0]
    dxy = np.hypot(got[:, 0] - ref[:, 0], got[:, 1] - ref[:, 1])
    dyaw = np.abs(got[:, 5] - ref[:, 5])
    print('262 144 x 256 fused step on the height grid vs oracle, %d steps: mean-pose deviation max %.3e m, yaw max %.3e rad' % (
        steps, dxy.max(), dyaw.max()))
    assert dxy.max() <= TRAJ_TOL_XY_M and dyaw.max() <= TRAJ_TOL_YAW_RAD
