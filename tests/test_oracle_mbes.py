"""CPU tests of the MBES self-oracle (fp64): analytic cases and brute-force cross-checks.
PARITY UNPINNED vs the reference (no MBES model there)."""
import numpy as np

from oracle import oracle as orc
from smarc_navigation_amd import synth


def test_flat_grid_analytic():
    z = np.full((40, 50), -25.0, np.float32)
    g = orc.Grid(z, (-20.0, -25.0), 1.0)
    for th in np.linspace(-0.8, 0.8, 9):
        d = np.array([0.0, np.sin(th), -np.cos(th)])
        r = g.ray(np.array([0.3, 0.2, -5.0]), d, 200.0)
        assert abs(r - 20.0 / np.cos(th)) < 1e-9


def test_tilted_plane_grid_analytic():
    nx = ny = 64
    x = -32.0 + np.arange(nx)
    y = -32.0 + np.arange(ny)
    z = (-20.0 + 0.1 * x[:, None] - 0.05 * y[None, :]).astype(np.float32)
    g = orc.Grid(z, (-32.0, -32.0), 1.0)
    o = np.array([1.5, -2.25, -3.0])
    rs = np.random.RandomState(0)
    for _ in range(50):
        d = rs.randn(3)
        d[2] = -abs(d[2]) - 0.5
        d /= np.linalg.norm(d)
        # plane: z = -20 + 0.1 x - 0.05 y
        t = (-20.0 + 0.1 * o[0] - 0.05 * o[1] - o[2]) / (d[2] - 0.1 * d[0] + 0.05 * d[1])
        r = g.ray(o, d, 500.0)
        p = o + t * d
        if abs(p[0]) < 31 and abs(p[1]) < 31:
            assert abs(r - t) < 1e-5  # float32 node heights


def test_grid_origin_below_surface_and_misses():
    z = np.full((10, 10), -5.0, np.float32)
    g = orc.Grid(z, (0.0, 0.0), 1.0)
    assert g.ray(np.array([4.0, 4.0, -6.0]), np.array([0.0, 0.0, -1.0]), 50.0) == 0.0
    assert g.ray(np.array([4.0, 4.0, -1.0]), np.array([0.0, 0.0, 1.0]), 50.0) == 50.0      # looks up
    assert g.ray(np.array([40.0, 4.0, -1.0]), np.array([0.0, 0.0, -1.0]), 50.0) == 50.0    # off the map
    r = g.ray(np.array([-3.0, 4.5, -1.0]), np.array([0.6, 0.0, -0.8]), 50.0)                # enters from outside
    assert abs(r - 5.0) < 1e-12


def test_mesh_accelerated_equals_brute_force():
    origin = (-16.0, -12.0)
    z = synth.bathymetry_grid(33, 25, 1.0, origin, seed=2)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    m = orc.Mesh(verts, tris)
    rs = np.random.RandomState(1)
    nhit = 0
    for _ in range(300):
        o = np.array([rs.uniform(-14, 14), rs.uniform(-10, 10), -2.0])
        d = rs.randn(3)
        d[2] = -abs(d[2]) - 1.5
        d /= np.linalg.norm(d)
        a, b = m.ray(o, d, 100.0), m.ray_brute(o, d, 100.0)
        assert abs(a - b) < 1e-9
        nhit += a < 100.0
    assert nhit > 150


def test_single_triangle_analytic():
    verts = np.array([[0, 0, -10], [10, 0, -10], [0, 10, -10]], np.float32)
    tris = np.array([[0, 1, 2]], np.uint32)
    m = orc.Mesh(verts, tris)
    assert abs(m.ray(np.array([2.0, 2.0, 0.0]), np.array([0.0, 0.0, -1.0]), 50.0) - 10.0) < 1e-12
    assert m.ray(np.array([8.0, 8.0, 0.0]), np.array([0.0, 0.0, -1.0]), 50.0) == 50.0
    d = np.array([0.6, 0.0, -0.8])
    assert abs(m.ray(np.array([0.0, 1.0, 0.0]), d, 50.0) - 12.5) < 1e-12


def test_mbes_update_likelihood_formula():
    z = np.full((20, 20), -12.0, np.float32)
    g = orc.Grid(z, (-10.0, -10.0), 1.0)
    soa = np.zeros((6, 1))
    soa[2] = -2.0
    ba = np.array([-0.3, 0.0, 0.3], np.float32)
    ranges = np.array([10.5, -1.0, 11.0], np.float32)
    lw, ex = orc.mbes_update(soa, np.identity(4), [0] * 6, g, ba, ranges, 0.5, 100.0)
    e0, e2 = 10.0 / np.cos(np.float64(ba[0])), 10.0 / np.cos(np.float64(ba[2]))
    want = -0.5 * (((10.5 - e0) / 0.5) ** 2 + ((11.0 - e2) / 0.5) ** 2) - 2 * np.log(0.5 * np.sqrt(2 * np.pi))
    assert abs(lw[0] - want) < 1e-9
    np.testing.assert_allclose(ex[0], [e0, 10.0, e2], atol=1e-9)


def test_fixed_point_spec_properties():
    rs = np.random.RandomState(0)
    for n in [1, 3, 1000, 5000]:
        lw = -rs.rand(n) * 50
        u = rs.random_sample()
        idx, ncum, q = orc.systematic_fixed(lw, 1, orc.u_to_u53(u))
        assert ncum[-1] == n and np.all(np.diff(ncum.astype(np.int64)) >= 0)
        w = np.exp(lw - lw.max())
        ref, _ = orc.systematic_ref(w / w.sum(), u)
        assert np.count_nonzero(ref != idx) == 0
        # sharded CDF == unsharded CDF
        if n >= 1000:
            h = n // 2
            tot = int(q.sum(dtype=np.uint64))
            a = orc.systematic_ncum(q[:h], orc.u_to_u53(u), 0, tot, n)
            b = orc.systematic_ncum(q[h:], orc.u_to_u53(u), int(q[:h].sum(dtype=np.uint64)), tot, n)
            assert np.array_equal(np.concatenate([a, b]), ncum)


def test_det_exp_accuracy():
    xs = np.concatenate([np.linspace(-700, 5, 2001), -np.logspace(-12, 2, 200)])
    got = np.array([orc.det_exp(x) for x in xs])
    ref = np.exp(xs)
    assert np.max(np.abs(got - ref) / ref) < 4e-16
    assert orc.det_exp(0.0) == 1.0 and orc.det_exp(-1e9) == 0.0 and orc.det_exp(float('-inf')) == 0.0
