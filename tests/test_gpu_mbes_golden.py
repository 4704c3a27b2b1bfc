"""GPU parity of the MBES / mesh / landmark updates against the INDEPENDENT golden vectors
(oracle/ref_harness/gen_golden_mbes.py: dense sampling + brentq, brute-force Moller-Trumbore over every
triangle, all-pairs landmark distances -- written separately from both the kernels and mcl_oracle.c).
Tolerances: expected range |d| <= 1e-3 m on every unambiguous ray, log-likelihood
|d| <= 1e-2 or 2e-4 relative; landmarks (fp64 on the GPU) 1e-9."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from smarc_navigation_amd import engine
    return engine


def _engine_with_map(eng, g, general=False):
    n = g['poses'].shape[0]
    e = eng.Engine(n, m2o=g['m2o'], rng_mode=eng.RNG_REPLAY)
    e.set_particles(np.ascontiguousarray(g['poses'].T))
    if str(g['kind']) == 'grid':
        e.set_map_grid(g['z'], tuple(g['origin']), float(g['res']))
    else:
        e.set_map_mesh(g['verts'], g['tris'], general=general)
    return e, n


@pytest.mark.parametrize('name,general,sweep', [('mbes_grid_interior', False, False), ('mbes_grid_rough', False, False),
                                                ('mbes_grid_border', False, False), ('mbes_mesh_regular', False, False),
                                                ('mbes_mesh_regular', True, False), ('mbes_mesh_tin', False, False),
                                                # triangle soups: the fan slice (mcl_slice.h) / the ray traversal over records
                                                ('mbes_mesh_regular', 'slice', False), ('mbes_mesh_tin', 'slice', False),
                                                ('mbes_tin_sweep', 'slice', False), ('mbes_mesh_sweep', 'slice', False),
                                                # the fan sweep (mcl_sweep.h), forced for these small pose sets
                                                ('mbes_mesh_regular', False, True), ('mbes_grid_interior', False, True),
                                                ('mbes_grid_rough', False, True), ('mbes_grid_border', False, True),
                                                # ... and on maps wide enough for the whole swath (gen_golden_mbes.py --sweep)
                                                ('mbes_mesh_sweep', False, True), ('mbes_mesh_sweep_d2', False, True),
                                                ('mbes_grid_sweep', False, True), ('mbes_grid_sweep_rough', False, True),
                                                ('mbes_tin_sweep', False, True), ('mbes_tin_sweep', False, False),
                                                ('mbes_mesh_sweep', False, False), ('mbes_grid_sweep_rough', False, False)])
def test_gpu_expected_ranges_and_loglik_match_independent_golden(name, general, sweep, eng, monkeypatch):
    g = helpers.load(name)
    monkeypatch.setenv('MCL_SWEEP', '1' if sweep else '0')
    monkeypatch.setenv('MCL_SLICE', '1' if general == 'slice' else '0')
    e, n = _engine_with_map(eng, g, bool(general))
    r_max, sigma = float(g['r_max']), float(g['sigma'])
    got = e.mbes_expected(0, n, g['beam_angles'], r_max, g['sensor_offset'])
    ok = g['ok']
    err = np.abs(got - g['expected'])
    print('%s%s: max |range error| %.3e m over %d unambiguous rays (%d at r_max)' % (
        name, ' (general path)' if general else '', err[ok].max(), int(ok.sum()), int((g['expected'] >= r_max)[ok].sum())))
    assert err[ok].max() <= 1e-3
    path, handed, _ = e.mbes_last_path()
    assert path == (1 if sweep else (2 if general == 'slice' else 0))
    if general == 'slice':
        print('   fan slice: %d of %d poses handed to the general kernel' % (handed, n))
    if sweep:
        print('   fan sweep: %d of %d poses handed to the traversal kernels' % (handed, n))
        # (small maps / rough terrain: the sweep declines, the result is checked either way; the *_sweep maps are
        #  wide enough for every pose, the rough one may lose the more tilted ones to the slope bound)
        assert handed < n or name not in ('mbes_grid_interior', 'mbes_grid_sweep_rough')
        assert handed == 0 or name not in ('mbes_mesh_sweep', 'mbes_mesh_sweep_d2', 'mbes_grid_sweep', 'mbes_tin_sweep')
    e.update_mbes(g['ranges'], g['beam_angles'], sigma, r_max, g['sensor_offset'])
    lw = e.get_log_weights()
    fin = np.isfinite(g['lw'])
    d = np.abs(lw - g['lw'])[fin]
    print('   max |dlw| %.3e (|lw| up to %.0f)' % (d.max(), np.abs(g['lw'][fin]).max()))
    assert np.all((d <= 1e-2) | (d <= 2e-4 * np.abs(g['lw'][fin])))


@pytest.mark.parametrize('k', [1, 2, 4])
def test_gpu_landmark_knn_matches_independent_golden(k, eng):
    g = helpers.load('landmarks_knn')
    n = g['poses'].shape[0]
    e = eng.Engine(n, m2o=g['m2o'], rng_mode=eng.RNG_REPLAY)
    e.set_particles(np.ascontiguousarray(g['poses'].T))
    e.set_landmarks(g['landmarks'])
    e.update_landmarks(g['det'], float(g['sigma']), k=k, gate=float(g['gate']), sensor_offset=g['sensor_offset'])
    np.testing.assert_allclose(e.get_log_weights(), g['lw_k%d' % k], rtol=1e-9, atol=1e-9)
