"""Edge cases of the GPU path (empty / ragged / degenerate inputs, maximum sizes),
each checked against the oracle's definition."""
import numpy as np
import pytest

from smarc_navigation_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from smarc_navigation_amd import engine
    return engine


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle
    return oracle


def _grid_engine(eng, n, nx=96, ny=96):
    origin = (-48.0, -48.0)
    z = synth.bathymetry_grid(nx, ny, 1.0, origin, seed=2)
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_map_grid(z, origin, 1.0)
    return e, z, origin


def test_all_beams_invalid_gives_uniform_weights(eng, orc):
    n = 1000
    e, z, origin = _grid_engine(eng, n)
    soa = np.zeros((6, n))
    soa[2] = -2.0
    e.set_particles(soa)
    ba = synth.beam_angles(32)
    ranges = np.full(32, np.nan, np.float32)
    ranges[::2] = 0.0
    e.update_mbes(ranges, ba, 0.2, 80.0)
    assert np.all(e.get_log_weights() == 0.0)
    e.resample(0.25, np.zeros((n, 6)))
    assert np.array_equal(e.last_indices(), np.arange(n))  # uniform weights: every particle survives


def test_rmax_shorter_than_depth_and_single_particle_single_beam(eng, orc):
    e, z, origin = _grid_engine(eng, 1)
    soa = np.array([[0.5], [0.25], [-2.0], [0.0], [0.0], [0.3]])
    e.set_particles(soa)
    ba = np.zeros(1, np.float32)
    got = e.mbes_expected(0, 1, ba, 5.0)  # the seabed is ~18 m away
    assert got[0, 0] == 5.0
    _, ref = orc.mbes_update(soa, np.identity(4), [0] * 6, orc.Grid(z, origin, 1.0), ba, None, 0.2, 200.0)
    got = e.mbes_expected(0, 1, ba, 200.0)
    assert abs(got[0, 0] - ref[0, 0]) < 1e-3
    e.update_mbes(np.array([ref[0, 0]], np.float32), ba, 0.2, 200.0)
    e.resample(0.9, np.zeros((1, 6)))
    assert e.last_indices()[0] == 0


def test_particles_off_the_map_and_below_the_seabed(eng, orc):
    n = 64
    e, z, origin = _grid_engine(eng, n)
    rs = np.random.RandomState(0)
    soa = np.zeros((6, n))
    soa[0] = rs.uniform(-300, 300, n)   # most are far off the 96 m map
    soa[1] = rs.uniform(-300, 300, n)
    soa[2] = -2.0
    soa[2, :8] = -40.0                  # under the seabed
    soa[0, :8] = rs.uniform(-10, 10, 8)
    soa[1, :8] = rs.uniform(-10, 10, 8)
    e.set_particles(soa)
    ba = synth.beam_angles(48, 0.9)
    got = e.mbes_expected(0, n, ba, 70.0)
    _, ref = orc.mbes_update(soa, np.identity(4), [0] * 6, orc.Grid(z, origin, 1.0), ba, None, 0.2, 70.0)
    assert np.all(got[:8] == 0.0) and np.all(ref[:8] == 0.0)  # origin below the surface -> range 0
    err = np.abs(got - ref)
    assert (err > 1e-3).mean() < 0.01, 'map-border grazing rays only'
    assert np.count_nonzero(ref == 70.0) > 1000


@pytest.mark.parametrize('case', ['all_minus_inf', 'one_survivor', 'nan_weights', 'huge_spread'])
def test_degenerate_weight_vectors(case, eng, orc):
    n = 4099
    rs = np.random.RandomState(1)
    lw = -rs.rand(n) * 10
    if case == 'all_minus_inf':
        lw[:] = -np.inf
    elif case == 'one_survivor':
        lw[:] = -1e6
        lw[1234] = 0.0
    elif case == 'nan_weights':
        lw[::3] = np.nan
    else:
        lw = -rs.rand(n) * 1400.0  # exp() underflows for most
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    soa = rs.randn(6, n)
    e.set_particles(soa)
    e.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
    e.resample(0.5, np.zeros((n, 6)))
    idx = e.last_indices()
    ref, _, _ = orc.systematic_fixed(lw, 1, orc.u_to_u53(0.5))
    assert np.array_equal(idx, ref)
    if case == 'all_minus_inf':
        assert np.array_equal(idx, np.arange(n))
    if case == 'one_survivor':
        assert np.all(idx == 1234)
        assert np.all(e.get_particles() == soa[:, 1234:1235])
    if case == 'nan_weights':
        assert not np.any(np.isin(idx, np.arange(0, n, 3)))


def test_four_million_particles_resample_and_moments(eng, orc):
    """Largest single-GPU shard of BASELINE config 4 (4 M): integer CDF vs oracle, bit-exact."""
    n = 1 << 22
    rs = np.random.RandomState(5)
    lw = -0.5 * rs.randn(n) ** 2 * 9.0
    e = eng.Engine(n, seed=11)
    e.init_particles()
    e.set_log_weights(lw, eng.WEIGHT_LOG_SHIFT)
    e.resample()
    idx = e.last_indices()
    ref, ncum, _ = orc.systematic_fixed(lw, 1, orc.native_u53(11, 0))
    assert np.array_equal(idx, ref)
    assert np.array_equal(e.last_offspring_cdf(), ncum)
    mean, yaw, cov = e.mean_cov()
    assert np.all(np.isfinite(mean)) and np.all(np.isfinite(cov))


def test_predict_nonpositive_dt_is_a_noop_and_bad_arguments_fail(eng):
    e = eng.Engine(128, init_cov=[1, 1, 0, 0, 0, 0.1], seed=1)
    e.init_particles()
    s0 = e.get_particles()
    e.predict([1, 0, 0], 0.1, [0, 0, 0, 1], -1.0, 0.0)
    e.predict([1, 0, 0], 0.1, [0, 0, 0, 1], -1.0, -0.5)
    assert np.array_equal(e.get_particles(), s0)  # auv_pf.py:205 gate
    with pytest.raises(eng.MclError):
        e.update_mbes(np.ones(4, np.float32), np.zeros(4, np.float32), -1.0, 50.0)  # sigma <= 0
    with pytest.raises(eng.MclError):
        e.resample()  # no weights yet
    with pytest.raises(eng.MclError):
        eng.Engine(0)
    with pytest.raises(eng.MclError):
        eng.Engine(16, device=99)
    r = eng.Engine(16, rng_mode=eng.RNG_REPLAY)
    with pytest.raises(eng.MclError):
        r.init_particles()  # REPLAY needs the draws


def test_beam_counts_not_multiple_of_64_and_unsorted_angles(eng, orc):
    n = 20
    e, z, origin = _grid_engine(eng, n, 160, 160)
    e2, _, _ = _grid_engine(eng, n, 160, 160)
    rs = np.random.RandomState(3)
    soa = rs.randn(6, n) * np.array([2, 2, 0.2, 0.05, 0.05, 3.0])[:, None]
    soa[2] -= 2.0
    for B in (1, 2, 63, 65, 127, 700):
        ba = rs.uniform(-1.0, 1.0, B).astype(np.float32)  # unsorted
        for ee, org in ((e, origin),):
            ee.set_particles(soa)
            ee.set_map_grid(synth.bathymetry_grid(160, 160, 1.0, (-80.0, -80.0), seed=2), (-80.0, -80.0), 1.0)
            got = ee.mbes_expected(0, n, ba, 90.0)
            _, ref = orc.mbes_update(soa, np.identity(4), [0] * 6,
                                     orc.Grid(synth.bathymetry_grid(160, 160, 1.0, (-80.0, -80.0), seed=2), (-80.0, -80.0), 1.0),
                                     ba, None, 0.2, 90.0)
            assert np.abs(got - ref).max() <= 1e-3, B
