"""The spatial visiting order of the fan sweep (mcl_kernels.h: VisitArgs; DESIGN.md 5 "particle order").

The fused step's gather bins every particle it writes by (x, y, yaw), k_visit_scan turns the per-workgroup counts into
positions, and the NEXT predict writes the pose records in that order, so that the 64 lanes of a sweep wavefront are
spatial neighbours.  Only the order in which particles are CAST changes -- state slots, RNG keys and keep / lost / dupes
(auv_pf.py:183-198) do not, and by the determinism rule no log-likelihood depends on it:

  * with and without the order the filter is the same, bit for bit (log-weights, indices, states, mean / cov), on every
    sweep surface, for particle counts that fill the gather's grid unevenly and for more particles per thread than the
    gather parks in LDS;
  * the order is a permutation of the slots, and a wave's 64 particles are far closer together than 64 random ones;
  * particles the sweep hands to the general kernel (border) arrive under the right slot.
"""
import numpy as np
import pytest

from smarc_navigation_amd import synth

pytestmark = pytest.mark.gpu

ORIGIN = (-64.0, -354.0)
COV = dict(init_cov=[2.0, 2.0, 0.0, 0.0, 0.0, 0.05], process_cov=[1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6],
           resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])


@pytest.fixture(scope='module')
def eng():
    from smarc_navigation_amd import engine
    return engine


@pytest.fixture(scope='module')
def terrain():
    return synth.bathymetry_grid(708, 708, 1.0, ORIGIN, seed=3)


def _set_map(e, z, kind):
    if kind == 'grid':
        e.set_map_grid(z, ORIGIN, 1.0)
    elif kind == 'mesh':
        e.set_map_mesh(*synth.mesh_from_grid(z, 1.0, ORIGIN))
    else:
        e.set_map_mesh(*synth.mesh_tin(z, 1.0, ORIGIN, seed=7))


def _run(eng, z, kind, n, steps, visit, monkeypatch, cov=COV, m2o=None, sigma=0.2, keep=False):
    monkeypatch.setenv('MCL_VISIT', visit)
    monkeypatch.setenv('MCL_SWEEP', '1')
    kw = dict(m2o=m2o) if m2o is not None else {}
    e = eng.Engine(n, seed=5, **cov, **kw)
    _set_map(e, z, kind)
    e.init_particles()
    stream = synth.odom_stream(steps)
    B = 128
    ba = synth.beam_angles(B)
    rs = np.random.RandomState(4)
    out = []
    for k in range(steps):
        ranges = (21.0 / np.cos(ba) + 0.05 * rs.randn(B)).astype(np.float32)
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges, ba, sigma, 100.0)
        path = e.mbes_last_path()
        assert path[0] == 1, path
        slots, srt = e.mbes_visit_order()
        mean, yaw, cov9 = e.mean_cov()
        out.append(dict(lw=e.get_log_weights(), idx=e.last_indices(), st=e.get_particles(), mean=mean, yaw=yaw, cov=cov9,
                        slots=slots, sorted=srt, handed=path[1]))
    if keep:
        return out, e
    e.close()
    return out


@pytest.mark.parametrize('kind,n', [('mesh', 65536), ('grid', 40000), ('tin', 50001), ('mesh', 300000)])
def test_filter_is_bitwise_the_same_with_and_without_the_visiting_order(kind, n, eng, terrain, monkeypatch):
    a = _run(eng, terrain, kind, n, 4, '1', monkeypatch)
    b = _run(eng, terrain, kind, n, 4, '0', monkeypatch)
    assert not a[0]['sorted'] and all(s['sorted'] for s in a[1:])       # (the first step has no gather behind it)
    assert not any(s['sorted'] for s in b)
    for k, (x, y) in enumerate(zip(a, b)):
        for key in ('lw', 'idx', 'st', 'mean', 'yaw', 'cov'):
            assert np.array_equal(x[key], y[key]), (k, key)
        assert x['handed'] == y['handed']
        assert np.array_equal(np.sort(x['slots']), np.arange(n, dtype=np.uint32)), k   # a permutation of the slots
        assert np.array_equal(y['slots'], np.arange(n, dtype=np.uint32))


def test_a_wave_of_the_sweep_holds_spatial_neighbours(eng, terrain, monkeypatch):
    n = 262144
    out, e = _run(eng, terrain, 'mesh', n, 12, '1', monkeypatch, keep=True)
    # the order of step k was prepared from the state step k - 1 left: compare with THAT state
    slots, st = out[-1]['slots'], out[-2]['st']
    e.close()
    pos = st[:, slots.astype(np.int64)]
    sd = np.array([st[0].std(), st[1].std(), st[5].std()])
    def spread(p):
        g = p[[0, 1, 5]].reshape(3, -1, 64)
        return (g.std(axis=2) / sd[:, None]).mean(axis=1)
    in_order, in_slots = spread(pos), spread(st)
    print('spread of (x, y, yaw) inside a wave / spread of the cloud: visiting order %r, slot order %r' % (in_order.round(3), in_slots.round(3)))
    assert (in_slots > 0.8).all()           # slot order: a wave is a random sample of the cloud
    # visiting order: x is the major key (one bin: a third of a sigma wide or finer); at 64 particles per bin a wave still
    # straddles a few (y, yaw) bins, the bins lag a contracting cloud by a step and the key anticipates the next predict's
    # noise -- the VOLUME a wave's particles span is what shrinks, by an order of magnitude
    assert in_order[0] < 0.5
    assert np.prod(in_order) < 0.2 * np.prod(in_slots)


def test_more_particles_per_thread_than_the_gather_parks_in_lds(eng, terrain, monkeypatch):
    """1 310 720 particles: five per gather thread, the fifth is binned from the state the gather wrote."""
    n = 5 * 262144
    a = _run(eng, terrain, 'mesh', n, 3, '1', monkeypatch)
    b = _run(eng, terrain, 'mesh', n, 3, '0', monkeypatch)
    assert a[-1]['sorted']
    for k, (x, y) in enumerate(zip(a, b)):
        for key in ('lw', 'idx', 'st', 'mean', 'yaw', 'cov'):
            assert np.array_equal(x[key], y[key]), (k, key)
        assert np.array_equal(np.sort(x['slots']), np.arange(n, dtype=np.uint32)), k


def test_hand_overs_keep_their_slot_under_the_visiting_order(eng, terrain, monkeypatch):
    """A cloud born across the map border: the sweep hands hundreds of particles to the general kernel in every step; their
    log-likelihoods land in the slots they belong to."""
    cov = dict(init_cov=[900.0, 900.0, 0.0, 0.0, 0.0, 0.5], process_cov=[1e-2, 1e-2, 0.0, 0.0, 0.0, 1e-4],
               resample_cov=[1.0, 1.0, 0.0, 0.0, 0.0, 1e-3])
    m2o = synth.rigid_matrix(ORIGIN[0] + 20.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    monkeypatch.setenv('MCL_VISIT_BINS', '16,16,16')
    a = _run(eng, terrain, 'mesh', 65536, 4, '1', monkeypatch, cov=cov, m2o=m2o, sigma=50.0)
    b = _run(eng, terrain, 'mesh', 65536, 4, '0', monkeypatch, cov=cov, m2o=m2o, sigma=50.0)
    assert a[-1]['sorted'] and min(s['handed'] for s in a[1:]) > 50, [s['handed'] for s in a]
    for k, (x, y) in enumerate(zip(a, b)):
        for key in ('lw', 'idx', 'st', 'mean', 'yaw', 'cov'):
            assert np.array_equal(x[key], y[key]), (k, key)
        assert x['handed'] == y['handed']


def test_separate_calls_between_fused_steps_fall_back_to_slot_order(eng, terrain, monkeypatch):
    """set_particles / a resample that is not preceded by a predict invalidate the prepared order; a dt <= 0 step has its
    records written by the stand-alone pose kernel -- in the prepared order, like the node's separate calls -- but (no
    predict in front of its gather: z, roll, pitch are not uniform, the plain gather kernel) prepares none for the step
    after it."""
    monkeypatch.setenv('MCL_VISIT', '1')
    monkeypatch.setenv('MCL_SWEEP', '1')
    n = 32768
    e = eng.Engine(n, seed=5, **COV)
    _set_map(e, terrain, 'mesh')
    e.init_particles()
    stream = synth.odom_stream(5)
    ba = synth.beam_angles(64)
    ranges = (21.0 / np.cos(ba)).astype(np.float32)
    def step(k, dt=None):
        e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'] if dt is None else dt, ranges, ba, 0.2, 100.0)
        return e.mbes_visit_order()[1]
    assert step(0) is False
    assert step(1) is True
    assert step(2, dt=0.0) is True            # no predict: the pose kernel of the update writes the records, in the prepared order
    assert step(3) is False
    assert step(4) is True
    e.set_particles(e.get_particles())
    assert step(0) is False
    assert step(1) is True
    e.update_mbes(ranges, ba, 0.2, 100.0)
    e.resample()                              # no predict since the last gather: the plain gather kernel prepares nothing
    assert step(2) is False
    e.close()


def test_separate_calls_take_the_visiting_order_too(eng, terrain, monkeypatch):
    """The node's call sequence -- mcl_predict per odometry message, mcl_update_mbes + mcl_resample per ping -- prepares the
    order in the plain resample (the stash kernel, its sums unused) and scatters the records in the stand-alone pose kernel:
    bit for bit the same filter as without the order, the sweep's records in sorted order from the second ping on."""
    n, steps = 65536, 4
    stream = synth.odom_stream(steps)
    B = 128
    ba = synth.beam_angles(B)
    res = {}
    for visit in ('1', '0'):
        monkeypatch.setenv('MCL_VISIT', visit)
        monkeypatch.setenv('MCL_SWEEP', '1')
        e = eng.Engine(n, seed=5, **COV)
        _set_map(e, terrain, 'mesh')
        e.init_particles()
        rs = np.random.RandomState(4)
        out = []
        for k in range(steps):
            for sub in range(3):   # three odometry messages per ping
                e.predict(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'] / 3.0)
            ranges = (21.0 / np.cos(ba) + 0.05 * rs.randn(B)).astype(np.float32)
            e.update_mbes(ranges, ba, 0.2, 100.0)
            slots, srt = e.mbes_visit_order()
            lw = e.get_log_weights()
            e.resample()
            out.append(dict(lw=lw, idx=e.last_indices(), st=e.get_particles(), mc=e.mean_cov(), slots=slots, sorted=srt))
        e.close()
        res[visit] = out
    assert [s['sorted'] for s in res['1']] == [False, True, True, True]
    assert not any(s['sorted'] for s in res['0'])
    for k, (x, y) in enumerate(zip(res['1'], res['0'])):
        assert np.array_equal(x['lw'], y['lw']) and np.array_equal(x['idx'], y['idx']) and np.array_equal(x['st'], y['st']), k
        for u, v in zip(x['mc'], y['mc']):
            assert np.array_equal(np.asarray(u), np.asarray(v)), k
        assert np.array_equal(np.sort(x['slots']), np.arange(n, dtype=np.uint32))
