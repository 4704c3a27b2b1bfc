"""The fp64 C oracle's MBES / mesh / landmark functions against the INDEPENDENT golden vectors of
oracle/ref_harness/gen_golden_mbes.py (dense sampling + brentq on the bilinear surface, brute-force
Moller-Trumbore over every triangle, all-pairs landmark distances: no code shared with mcl_oracle.c).
The reference itself has no MBES model, so this is the strongest pin these rows can get:
two implementations written separately from the definition in DESIGN.md section 5 agree."""
import numpy as np
import pytest

from tests import helpers

GRID_CASES = ['mbes_grid_interior', 'mbes_grid_rough', 'mbes_grid_border', 'mbes_grid_sweep', 'mbes_grid_sweep_rough']
MESH_CASES = ['mbes_mesh_regular', 'mbes_mesh_tin', 'mbes_mesh_sweep', 'mbes_mesh_sweep_d2', 'mbes_tin_sweep']


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle
    return oracle


def _map(orc, g):
    if str(g['kind']) == 'grid':
        return orc.Grid(g['z'], tuple(g['origin']), float(g['res']))
    return orc.Mesh(g['verts'], g['tris'])


@pytest.mark.parametrize('name', GRID_CASES + MESH_CASES)
def test_oracle_expected_ranges_and_loglik_match_independent_golden(name, orc):
    g = helpers.load(name)
    soa = np.ascontiguousarray(g['poses'].T)
    r_max, sigma = float(g['r_max']), float(g['sigma'])
    lw, exp = orc.mbes_update(soa, g['m2o'], g['sensor_offset'], _map(orc, g), g['beam_angles'], g['ranges'], sigma, r_max)
    ok = g['ok']
    err = np.abs(exp - g['expected'])[ok]
    print('%s: %d rays, max |range error| %.3e m, %d at r_max' % (name, err.size, err.max(), int((g['expected'] >= r_max).sum())))
    assert ok.mean() > 0.97
    assert err.max() <= 1e-7   # two fp64 implementations of the same definition
    fin = np.isfinite(g['lw'])
    assert fin.sum() >= 0.9 * fin.size
    np.testing.assert_allclose(lw[fin], g['lw'][fin], rtol=1e-9, atol=1e-6)
    # hits and misses are the same rays
    assert np.array_equal((exp >= r_max)[ok], (g['expected'] >= r_max)[ok])


@pytest.mark.parametrize('k', [1, 2, 4])
def test_oracle_landmark_knn_matches_independent_golden(k, orc):
    g = helpers.load('landmarks_knn')
    soa = np.ascontiguousarray(g['poses'].T)
    lw = orc.landmark_update(soa, g['m2o'], g['sensor_offset'], g['landmarks'], g['det'], float(g['sigma']), k, float(g['gate']))
    np.testing.assert_allclose(lw, g['lw_k%d' % k], rtol=1e-10, atol=1e-9)
