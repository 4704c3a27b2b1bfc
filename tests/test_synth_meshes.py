"""The synthetic survey meshes of the bench and the GPU tests (smarc_navigation_amd/synth.py, bench.py: no GPU needed): an
irregular TIN, the same handed over in random order, with data gaps, with a ragged outline -- what the generators promise
(the GPU tests and the bench legs build on it): same surface after a shuffle, one edge-connected piece after mesh_ragged,
gaps only inside, every edge shared by at most two triangles."""
import numpy as np

from smarc_navigation_amd import synth


def _tin(nx=60, ny=50, seed=3):
    origin = (-30.0, -25.0)
    z = synth.bathymetry_grid(nx, ny, 1.0, origin, seed=seed)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=seed)
    return origin, verts, tris


def _edge_counts(tris):
    t = tris.astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    e.sort(axis=1)
    _, cnt = np.unique(e[:, 0] * (t.max() + 1) + e[:, 1], return_counts=True)
    return cnt


def test_shuffle_keeps_the_surface():
    _, verts, tris = _tin()
    v2, t2 = synth.mesh_shuffle(verts, tris, seed=5)
    a = np.sort(np.sort(verts[tris.astype(np.int64)].reshape(len(tris), 9), axis=1), axis=0)
    b = np.sort(np.sort(v2[t2.astype(np.int64)].reshape(len(t2), 9), axis=1), axis=0)
    assert np.array_equal(a, b)          # the same triangles (as sets of coordinates)
    assert not np.array_equal(tris, t2)


def test_ragged_outline_is_one_piece_with_bays():
    origin, verts, tris = _tin()
    t = synth.mesh_ragged(verts, tris, seed=2, band=2.0, bays=4, bay_width=(2.0, 4.0), bay_depth=(6.0, 15.0), keep=(0.0, 0.0))
    assert 0.6 * len(tris) < len(t) < len(tris)
    assert _edge_counts(t).max() == 2    # still a manifold with boundary
    # one edge-connected piece: a flood over shared edges reaches every triangle
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    tt = t.astype(np.int64)
    e = np.concatenate([tt[:, [0, 1]], tt[:, [1, 2]], tt[:, [2, 0]]])
    e.sort(axis=1)
    key = e[:, 0] * (verts.shape[0] + 1) + e[:, 1]
    order = np.argsort(key, kind='stable')
    same = key[order][1:] == key[order][:-1]
    a, b = order[:-1][same] % len(tt), order[1:][same] % len(tt)
    n, _ = connected_components(coo_matrix((np.ones(a.size), (a, b)), shape=(len(tt), len(tt))), directed=False)
    assert n == 1
    # the kept point is still on the mesh, the outline is no longer the bounding box all around
    c = verts[tt].mean(axis=1)
    assert np.hypot(c[:, 0], c[:, 1]).min() < 1.0
    boundary = _edge_counts(t) == 1
    assert boundary.sum() > 2 * (60 + 50 - 2)     # longer than the rectangle's outline: sawtooth and bays


def test_punch_hole_and_gaps_remove_interior_triangles_only():
    import bench
    origin, verts, tris = _tin(90, 80)
    m = dict(verts=verts, tris=tris, origin=(-30.0, -25.0), desc='tin')
    one = bench.punch_hole(m, 5.0, 3.0, radius=1.2)
    gone = len(tris) - len(one['tris'])
    assert 2 <= gone <= 16 and 'missing' in one['desc']
    many = bench.punch_gaps(m, tile=6.0, seed=1)
    assert 0.05 * len(tris) < len(tris) - len(many['tris']) < 0.3 * len(tris)
    # the outer ring of tiles is intact: the outline is still the rectangle's
    c = verts[many['tris'].astype(np.int64)].mean(axis=1)
    x0, y0 = m['origin']
    ring = (c[:, 0] < x0 + 6.0) | (c[:, 1] < y0 + 6.0)
    c_all = verts[tris.astype(np.int64)].mean(axis=1)
    assert ring.sum() == ((c_all[:, 0] < x0 + 6.0) | (c_all[:, 1] < y0 + 6.0)).sum()
