"""GPU parity tests for the fan slice (smarc_navigation_amd/csrc/mcl_slice.h): the MBES update over ARBITRARY triangle
meshes -- soups with overhangs, vertical faces, floating sheets, holes -- without a march per ray.  Every case is checked
against the fp64 oracle (oracle/mcl_oracle.c: brute-force Moller-Trumbore over every triangle) within SURVEY 8(d)'s
1e-3 m, and against the ray traversal over triangle records (MCL_SLICE=0: an independent fp32 algorithm);
mcl_mbes_last_path says which kernels really ran (2 = the slice) and how many particles it handed over."""
import numpy as np
import pytest

from smarc_navigation_amd import synth
from tests.helpers import outliers_explained

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope='module')
def eng():
    from smarc_navigation_amd import engine
    return engine


def _cloud(n, seed, spread, centre):
    rs = np.random.RandomState(seed)
    soa = rs.randn(6, n) * np.array(spread)[:, None]
    for k in range(3):
        soa[k] += centre[k]
    return soa


def _soup(seed=3, nt=400):
    """Random triangles at several depths (overhangs everywhere) over a flat floor of two big triangles."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-25, 25, size=(nt, 2))
    zc = rs.uniform(-30, -10, size=nt)
    verts = np.zeros((nt * 3 + 4, 3), np.float32)
    for k in range(nt):
        for v in range(3):
            verts[3 * k + v, :2] = c[k] + rs.uniform(-3, 3, size=2)
            verts[3 * k + v, 2] = zc[k] + rs.uniform(-1.5, 1.5)
    tris = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
    verts[-4:] = [[-40, -40, -35], [40, -40, -35], [40, 40, -35], [-40, 40, -35]]
    b = nt * 3
    tris = np.vstack([tris, np.array([[b, b + 1, b + 2], [b, b + 2, b + 3]], np.uint32)])
    return verts, tris


def _terrain_with_wall_and_deck(seed=8):
    """A triangulated terrain + a vertical wall standing on it + a deck floating 6 m above it: not a height field."""
    origin = (-60.0, -60.0)
    z = synth.bathymetry_grid(120, 120, 1.0, origin, seed=seed)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    nv = verts.shape[0]
    wall = np.array([[10.0, -20.0, -30.0], [10.0, 20.0, -30.0], [10.0, 20.0, -8.0], [10.0, -20.0, -8.0]], np.float32)
    deck = np.array([[-25.0, -15.0, -12.0], [-5.0, -15.0, -12.5], [-5.0, 15.0, -12.0], [-25.0, 15.0, -11.5]], np.float32)
    verts = np.vstack([verts, wall, deck]).astype(np.float32)
    extra = np.array([[nv, nv + 1, nv + 2], [nv, nv + 2, nv + 3], [nv + 4, nv + 5, nv + 6], [nv + 4, nv + 6, nv + 7]], np.uint32)
    return verts, np.vstack([tris, extra])


def _check(eng, orc, verts, tris, soa, B, r_max, monkeypatch, general=True, max_bad=0, off=None, half_swath=np.pi / 3,
           min_handed=0, max_handed=0):
    ba = synth.beam_angles(B, half_swath)
    mesh = orc.Mesh(verts, tris)
    n = soa.shape[1]
    _, ref = orc.mbes_update(soa, np.identity(4), off or [0] * 6, mesh, ba, None, 0.2, r_max)
    out = {}
    for slice_on in ('1', '0'):
        monkeypatch.setenv('MCL_SLICE', slice_on)
        e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
        e.set_particles(soa)
        e.set_map_mesh(verts, tris, general=general)
        got = e.mbes_expected(0, n, ba, r_max, off)
        path = e.mbes_last_path()
        rs = np.random.RandomState(1)
        ranges = (ref[0] + 0.2 * rs.randn(B)).astype(np.float32)
        if B > 8:
            ranges[::7] = 0.0
            ranges[3] = np.nan
        e.update_mbes(ranges, ba, 0.2, r_max, off)
        lw = e.get_log_weights()
        e.close()
        out[slice_on] = (got, lw, path)
    got, lw, path = out['1']
    assert path[0] == 2 and out['0'][2][0] == 0, (path, out['0'][2])
    assert min_handed <= path[1] <= max_handed, path
    err = np.abs(got - ref)
    bad = int((err > 1e-3).sum())
    print('slice: max |expected range error| %.3e m over %d rays (%d beyond 1e-3), handed over %d of %d; vs traversal %.3e' % (
        err.max(), err.size, bad, path[1], n, np.abs(got - out['0'][0]).max()))
    assert bad <= max_bad, np.sort(err.ravel())[-5:]
    outliers_explained(orc, mesh, soa, ba, got, ref, r_max, off=off, label='slice')
    lw_ref, _ = orc.mbes_update(soa, np.identity(4), off or [0] * 6, mesh, ba, ranges, 0.2, r_max)
    rel = np.abs(lw - lw_ref) / np.maximum(1.0, np.abs(lw_ref))
    if max_bad == 0:
        assert rel.max() <= 2e-4, rel.max()
    return got, ref


def test_slice_on_a_soup_with_overhangs_the_nearest_hit_wins(eng, orc, monkeypatch):
    verts, tris = _soup()
    soa = _cloud(96, 4, (4.0, 4.0, 0.3, 0.05, 0.05, 3.0), (0.0, 0.0, -2.0))
    got, ref = _check(eng, orc, verts, tris, soa, 256, 80.0, monkeypatch, general=False, max_bad=4)
    assert (ref < 30.0).mean() > 0.2 and (ref > 30.0).mean() > 0.2   # both the floating triangles and the floor are hit


def test_slice_with_a_vertical_wall_and_a_floating_deck(eng, orc, monkeypatch):
    verts, tris = _terrain_with_wall_and_deck()
    soa = _cloud(128, 5, (6.0, 6.0, 0.3, 0.04, 0.04, 3.0), (0.0, 0.0, -2.0))
    _check(eng, orc, verts, tris, soa, 200, 70.0, monkeypatch, general=False, max_bad=3,
           off=[0.3, -0.1, -0.2, 0.01, -0.02, 0.05])


def test_slice_steep_sheet_that_passes_over_the_sensor(eng, orc, monkeypatch):
    """A large steep sheet (a cliff face, a wreck's side) rising from 10 m below the sensor on its starboard side to 10 m
    above it on its port side: in the fan plane a segment from (s, t) = (5, 10) to (-1, -10), whose hidden end lies on the
    OTHER side of the nadir than the part the beams see (it crosses the sensor's horizon at s = +2 and covers every
    tangent from 0.5 upward).  ADVICE r4: the beam run was taken from the hidden end's own sign and the sheet was missed."""
    floor = [[-60, -60, -35], [60, -60, -35.5], [60, 60, -35], [-60, 60, -34.5]]
    # sensor near (0, 0, -15) heading along +x: starboard = -y.  The sheet: y = -5 at z = -25 up to y = +1 at z = -5
    sheet = [[-30, -5, -25], [30, -5, -25], [30, 1, -5], [-30, 1, -5]]
    # ... and its mirror image on the port side, further out, so that both signs of the hidden end are exercised
    sheet2 = [[-30, 9, -25], [30, 9, -25], [30, 3, -5], [-30, 3, -5]]
    verts = np.array(floor + sheet + sheet2, np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]], np.uint32)
    soa = _cloud(96, 12, (1.0, 0.4, 0.5, 0.05, 0.05, 0.3), (0.0, 1.0, -15.0))
    got, ref = _check(eng, orc, verts, tris, soa, 255, 80.0, monkeypatch, general=False, max_bad=3, half_swath=1.3)
    assert (ref < 12.0).mean() > 0.25          # a good part of every fan ends on a sheet, metres from the sensor
    assert (ref > 15.0).mean() > 0.05          # ... the beams between the sheets reach the floor


@pytest.mark.parametrize('B', [512, 33, 2, 1])
def test_slice_on_a_tin_cast_as_a_soup_vs_oracle(B, eng, orc, monkeypatch):
    origin = (-90.0, -80.0)
    z = synth.bathymetry_grid(200, 180, 1.0, origin, seed=8)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7)
    soa = _cloud(64 if B > 2 else 7, 3, (4.0, 4.0, 0.3, 0.06, 0.06, 3.0), (5.0, 8.0, -2.0))
    _check(eng, orc, verts, tris, soa, B, 80.0, monkeypatch)


def test_slice_hands_strongly_rolled_fans_to_the_general_kernel(eng, orc, monkeypatch):
    origin = (-90.0, -80.0)
    z = synth.bathymetry_grid(200, 180, 1.0, origin, seed=9)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    n = 96
    soa = _cloud(n, 6, (3.0, 3.0, 0.2, 0.0, 0.0, 3.0), (0.0, 0.0, -3.0))
    soa[3] = np.where(np.arange(n) % 3 == 0, 1.2, 0.1)     # every third vehicle rolled by 69 degrees
    soa[4] = np.where(np.arange(n) % 3 == 1, 0.4, 0.02)    # ... or pitched by 23 (still sliced)
    _check(eng, orc, verts, tris, soa, 128, 90.0, monkeypatch, max_bad=4, min_handed=n // 3, max_handed=n // 3)


def test_slice_sensors_off_the_map_and_under_the_mesh(eng, orc, monkeypatch):
    origin = (-40.0, -40.0)
    z = synth.bathymetry_grid(80, 80, 1.0, origin, seed=5)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    n = 120
    soa = _cloud(n, 7, (30.0, 30.0, 0.3, 0.05, 0.05, 3.0), (0.0, 0.0, -2.0))   # a third of them off the 80 m map
    soa[2, ::10] = -40.0                                                       # under the seabed: sees it from below
    got, ref = _check(eng, orc, verts, tris, soa, 128, 60.0, monkeypatch, max_bad=6, half_swath=1.3)
    assert (ref == 60.0).mean() > 0.1 and (ref < 60.0).mean() > 0.3


def test_fused_steps_on_a_soup_take_the_slice_and_track_the_truth(eng, orc, monkeypatch):
    """predict + slice update + resample on the irregular TIN cast as a soup: the cloud converges onto the track."""
    monkeypatch.delenv('MCL_SLICE', raising=False)
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=3)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7)
    n, B = 16384, 128
    steps = 12
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    one.set_map_mesh(verts, tris)
    e = eng.Engine(n, seed=3, init_cov=[4.0, 4.0, 0, 0, 0, 0.02], process_cov=[1e-4, 1e-4, 0, 0, 0, 1e-6],
                   resample_cov=[1e-2, 1e-2, 0, 0, 0, 1e-5])
    e.set_map_mesh(verts, tris, general=True)
    e.init_particles()
    rs = np.random.RandomState(2)
    for k in range(steps):
        one.set_particles(stream['truth'][k][:, None].copy())
        ranges = (one.mbes_expected(0, 1, ba, 100.0)[0] + 0.2 * rs.randn(B)).astype(np.float32)
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges, ba, 0.2, 100.0)
        assert e.mbes_last_path()[0] == 2
    mean, yaw, cov = e.last_mean_cov()
    err = np.hypot(mean[0] - stream['truth'][steps - 1][0], mean[1] - stream['truth'][steps - 1][1])
    print('soup filter: mean error %.3f m after %d steps, sigma %.3f x %.3f' % (err, steps, np.sqrt(cov[0]), np.sqrt(cov[4])))
    assert err < 0.5


@pytest.mark.parametrize('seed', range(12))
def test_slice_fuzz_against_the_oracle(seed, eng, orc, monkeypatch):
    """Random soups (triangle sizes from 0.5 to 12 m, depths from -35 to -8 m, a floor under them), random attitudes up to
    35 degrees of roll / pitch, sensors inside, over the edge of and well off the soup's extent, odd beam counts and
    swaths: every ray within 1e-3 m of the brute-force oracle except a handful that graze an edge (counted, bounded)."""
    rs = np.random.RandomState(100 + seed)
    nt = int(rs.choice([40, 200, 800]))
    size = float(rs.choice([0.5, 3.0, 12.0]))
    ext = float(rs.choice([15.0, 40.0]))
    c = rs.uniform(-ext, ext, size=(nt, 2))
    zc = rs.uniform(-35.0, -8.0, size=nt)
    verts = np.zeros((nt * 3 + 4, 3), np.float32)
    for k in range(nt):
        for v in range(3):
            verts[3 * k + v, :2] = c[k] + rs.uniform(-size, size, size=2)
            verts[3 * k + v, 2] = zc[k] + rs.uniform(-0.3 * size, 0.3 * size)
    tris = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
    e2 = ext + 15.0
    verts[-4:] = [[-e2, -e2, -40.0], [e2, -e2, -40.5], [e2, e2, -40.0], [-e2, e2, -39.5]]
    b = nt * 3
    tris = np.vstack([tris, np.array([[b, b + 1, b + 2], [b, b + 2, b + 3]], np.uint32)])
    n = 48
    soa = rs.randn(6, n) * np.array([0.6 * ext, 0.6 * ext, 1.0, 0.25, 0.25, 3.0])[:, None]
    soa[2] -= 3.0
    soa[:2, ::6] *= 2.5                                   # every sixth sensor far out, some beyond the floor's edge
    soa[3:5] = np.clip(soa[3:5], -0.6, 0.6)
    B = int(rs.choice([1, 7, 64, 255]))
    half = float(rs.choice([0.6, 1.05, 1.4]))
    ba = synth.beam_angles(B, half)
    r_max = float(rs.choice([45.0, 90.0]))
    mesh = orc.Mesh(verts, tris)
    _, ref = orc.mbes_update(soa, np.identity(4), [0] * 6, mesh, ba, None, 0.2, r_max)
    monkeypatch.setenv('MCL_SLICE', '1')
    e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
    e.set_particles(soa)
    e.set_map_mesh(verts, tris)
    got = e.mbes_expected(0, n, ba, r_max)
    path = e.mbes_last_path()
    e.close()
    assert path[0] == 2, path
    err = np.abs(got - ref)
    bad = int((err > 1e-3).sum())
    print('fuzz %d: %d triangles of ~%.1f m, %d beams, handed over %d of %d; max err %.2e, %d of %d rays beyond 1e-3' % (
        seed, nt, size, B, path[1], n, err.max(), bad, err.size))
    assert bad <= max(3, err.size // 2000), np.sort(err.ravel())[-6:]
    outliers_explained(orc, mesh, soa, ba, got, ref, r_max, label='slice fuzz %d' % seed)


@pytest.mark.parametrize('kind', ['tin', 'regular'])
def test_slice_over_groups_of_neighbours_equals_the_per_particle_slice_bitwise(kind, eng, monkeypatch):
    """With the visiting order 36 consecutive pose records are spatial neighbours: k_mbes_slice_group builds ONE candidate
    triangle list per group (every test widened by the group's spread), stages each triangle's vertices once in LDS and
    casts the members with the per-particle kernel's own arithmetic.  A member's result is the minimum over the triangles
    ITS plane cuts, and the staged set contains them all: the filter is bit for bit the one the per-particle kernel
    runs -- log-weights, indices, states --, on the irregular TIN cast as a soup (a triangle lies in several cells: staged
    once) and on the regular mesh, while most groups really take the group path."""
    monkeypatch.delenv('MCL_SLICE', raising=False)
    monkeypatch.setenv('MCL_VISIT', '1')
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=3)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7) if kind == 'tin' else synth.mesh_from_grid(z, 1.0, origin)
    n, B, steps = 65536, 200, 5
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY)
    one.set_map_mesh(verts, tris)
    rs = np.random.RandomState(2)
    pings = []
    for k in range(steps):
        one.set_particles(stream['truth'][k][:, None].copy())
        pings.append((one.mbes_expected(0, 1, ba, 100.0)[0] + 0.2 * rs.randn(B)).astype(np.float32))
    one.close()
    res = {}
    for group in ('1', '0'):
        monkeypatch.setenv('MCL_SLICE_GROUP', group)
        e = eng.Engine(n, seed=3, init_cov=[4.0, 4.0, 0, 0, 0, 0.02], process_cov=[1e-4, 1e-4, 0, 0, 0, 1e-6],
                       resample_cov=[1e-2, 1e-2, 0, 0, 0, 1e-5])
        e.set_map_mesh(verts, tris, general=True)
        e.init_particles()
        out = []
        for k in range(steps):
            e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], pings[k], ba, 0.2, 100.0)
            path = e.mbes_last_path()
            assert path[0] == 2, path
            out.append((e.get_log_weights(), e.last_indices(), e.get_particles(), path))
        e.close()
        res[group] = out
    ngroups = (n + 59) // 60   # (mcl_slice.h: SLICE_G)
    loose = [o[3][2] for o in res['1']]
    print('%s: groups left to the per-particle kernel per step: %r of %d' % (kind, loose, ngroups))
    assert loose[0] == -1                         # the first step has no visiting order: no groups
    assert all(0 <= v < ngroups // 4 for v in loose[2:]), loose
    assert all(o[3][2] == -1 for o in res['0'])
    for k, (x, y) in enumerate(zip(res['1'], res['0'])):
        assert np.array_equal(x[0], y[0]), (k, int((x[0] != y[0]).sum()))
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), k
        assert x[3][1] == y[3][1]


def test_slice_beam_tables_too_long_for_the_group_kernel_fall_back_instead_of_failing(eng, monkeypatch):
    """the group kernel stages the ping's beam table in LDS beside the triangles; at 2 040 - 2 048 beams
    (the sweep / slice accept up to 2 048) its dynamic LDS exceeds what hipFuncSetAttribute grants, and from ~1 990 its
    static __shared__ pushes the launch over 160 KiB.  The host now decides BEFORE asking the runtime: such a ping is cast
    by the per-particle kernel -- the fused step succeeds and equals the MCL_SLICE_GROUP=0 filter bit for bit -- while a
    table that fits still takes the group kernel."""
    monkeypatch.delenv('MCL_SLICE', raising=False)
    monkeypatch.setenv('MCL_VISIT', '1')
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=3)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7)
    n, steps = 8192, 3
    stream = synth.odom_stream(steps)
    for B, grouped in ((2048, False), (2000, False), (1500, True)):
        ba = synth.beam_angles(B)
        ranges = (22.0 / np.cos(ba)).astype(np.float32)
        res = {}
        for group in ('1', '0'):
            monkeypatch.setenv('MCL_SLICE_GROUP', group)
            e = eng.Engine(n, seed=3, init_cov=[1.0, 1.0, 0, 0, 0, 0.01], process_cov=[1e-4, 1e-4, 0, 0, 0, 1e-6],
                           resample_cov=[1e-2, 1e-2, 0, 0, 0, 1e-5])
            e.set_map_mesh(verts, tris, general=True)
            e.init_particles()
            out = []
            for k in range(steps):
                e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges, ba, 2.0, 100.0)
                path = e.mbes_last_path()
                assert path[0] == 2, path
                out.append((e.get_log_weights(), e.last_indices(), path[2]))
            e.close()
            res[group] = out
        loose = [o[2] for o in res['1']]
        print('%d beams: groups left to the per-particle kernel per step %r' % (B, loose))
        assert (loose[-1] >= 0) == grouped, (B, loose)   # -1: the group kernel did not run
        for x, y in zip(res['1'], res['0']):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), B


def test_slice_groups_on_a_dispersed_cloud_and_an_odd_particle_count(eng, monkeypatch):
    """A cloud a metre wide at 12 particles per bin (groups of neighbours whose planes differ by more than the group kernel
    accepts over a 40 m fan: left to the per-particle kernel, by its list; the bitwise test above is the mostly-tight
    case), 50 001 particles (the last group is partial), strongly tilted vehicles among
    them (declined by both kernels: the general kernel casts them, by the record's slot): still bit for bit the
    per-particle filter."""
    monkeypatch.delenv('MCL_SLICE', raising=False)
    monkeypatch.setenv('MCL_VISIT', '1')
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=5)
    verts, tris = synth.mesh_tin(z, 1.0, origin, seed=9)
    n, B, steps = 50001, 96, 4
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    res = {}
    for group in ('1', '0'):
        monkeypatch.setenv('MCL_SLICE_GROUP', group)
        # (a flat likelihood -- sigma 30 m -- keeps the cloud as wide as the resampling noise makes it: ~1 m, 0.01 rad)
        e = eng.Engine(n, seed=4, init_cov=[4.0, 4.0, 0, 0, 0, 0.001], process_cov=[1e-2, 1e-2, 0, 0, 0, 1e-6],
                       resample_cov=[0.25, 0.25, 0, 0, 0, 1e-4])
        e.set_map_mesh(verts, tris, general=True)
        e.init_particles()
        out = []
        for k in range(steps):
            q = __import__('oracle.oracle', fromlist=['x']).quat_from_euler(1.2 if k == 2 else 0.02, 0.01, 0.0)   # step 2: rolled by 69 degrees
            e.step_mbes(stream['v'][k], stream['wz'][k], q, stream['z'][k], stream['dt'],
                        (22.0 / np.cos(ba)).astype(np.float32), ba, 30.0, 100.0)
            out.append((e.get_log_weights(), e.last_indices(), e.get_particles(), e.mbes_last_path()))
        e.close()
        res[group] = out
    ngroups = (n + 59) // 60   # (mcl_slice.h: SLICE_G)
    loose = [o[3][2] for o in res['1']]
    handed = [o[3][1] for o in res['1']]
    print('dispersed cloud: groups left to the per-particle kernel per step %r of %d, handed to the general kernel %r' % (loose, ngroups, handed))
    assert loose[0] == -1 and min(loose[1:]) > ngroups // 2      # most groups are not tight: cast from the list ...
    assert handed[2] == n                                         # ... and the rolled step is declined altogether
    for k, (x, y) in enumerate(zip(res['1'], res['0'])):
        assert np.array_equal(x[0], y[0]), (k, int((x[0] != y[0]).sum()))
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), k
        assert x[3][1] == y[3][1]
