"""Dead-reckoning integrator (include/mcl_dr.h, host-only code in libmcl_hip.so) against the golden
fixtures produced by the reference's own dr_node.py / sam_mm.py (oracle/ref_harness/gen_golden_dr.py),
and against the pure-Python restatement in oracle/dr_oracle.py on further seeds.  No GPU needed."""
import os

import numpy as np
import pytest

from smarc_navigation_amd import dr, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    ptf = g['pressure_tf']
    return g, (None if np.isnan(ptf).any() else ptf)


@pytest.mark.parametrize('name', ['dr_auv', 'dr_surface'])
def test_events_regenerate_from_seed(name):
    g, _ = _load(name)
    t, k, d = synth.raw_sensor_events(scenario=str(g['scenario']))
    assert np.array_equal(t, g['ev_t']) and np.array_equal(k, g['ev_kind']) and np.array_equal(d, g['ev_data'])


@pytest.mark.parametrize('name', ['dr_auv', 'dr_surface'])
def test_library_matches_reference_node(name):
    g, ptf = _load(name)
    ticks, m2o = dr.replay_events(g['ev_t'], g['ev_kind'], g['ev_data'], gps_map=g['gps_map'], pressure_tf=ptf,
                                  dvl_period=float(g['dvl_period']), dr_period=float(g['dr_period']))
    ref = g['ticks']
    assert ticks.shape == ref.shape
    assert np.array_equal(ticks[:, 0], ref[:, 0])            # which ticks publish: exact
    pub = ref[:, 0] > 0
    assert pub.sum() > 1800
    # 2000 ticks of accumulation; the node inverts its mass matrix with LAPACK, the library in closed form
    np.testing.assert_allclose(ticks[pub, 1:], ref[pub, 1:], rtol=0, atol=1e-11)
    np.testing.assert_allclose(m2o, g['m2o'], rtol=0, atol=1e-14)


@pytest.mark.parametrize('name', ['dr_auv', 'dr_surface'])
def test_oracle_matches_reference_node(name):
    from oracle import dr_oracle
    g, ptf = _load(name)
    ticks, m2o = dr_oracle.replay_events(g['ev_t'], g['ev_kind'], g['ev_data'], gps_map=g['gps_map'], pressure_tf=ptf,
                                         dvl_period=float(g['dvl_period']), dr_period=float(g['dr_period']))
    ref = g['ticks']
    assert np.array_equal(ticks[:, 0], ref[:, 0])
    pub = ref[:, 0] > 0
    np.testing.assert_allclose(ticks[pub, 1:], ref[pub, 1:], rtol=0, atol=1e-11)
    np.testing.assert_allclose(m2o, g['m2o'], rtol=0, atol=1e-14)


@pytest.mark.parametrize('seed,scenario', [(1, 'auv'), (2, 'surface'), (3, 'auv')])
def test_library_matches_oracle_on_other_streams(seed, scenario):
    from oracle import dr_oracle
    t, k, d = synth.raw_sensor_events(duration=25.0, seed=seed, scenario=scenario)
    ptf = (0.2, 0.0, 0.1) if scenario == 'auv' else None
    a, ma = dr.replay_events(t, k, d, pressure_tf=ptf, dvl_period=0.25, dr_period=0.02)
    b, mb = dr_oracle.replay_events(t, k, d, pressure_tf=ptf, dvl_period=0.25, dr_period=0.02)
    assert np.array_equal(a[:, 0], b[:, 0])
    pub = a[:, 0] > 0
    np.testing.assert_allclose(a[pub, 1:], b[pub, 1:], rtol=0, atol=1e-11)
    np.testing.assert_allclose(ma, mb, rtol=0, atol=1e-14)


def test_gates_and_flags():
    v = dr.VehicleDR(0.2, 0.02)
    assert v.dr_timer().published == 0                     # nothing initialised
    assert not v.gps_cb(1.0, 2.0)                          # no heading yet: the fix is not used
    v.sbg_cb(synth.quat_from_rpy(0.0, 0.0, 0.5))
    assert v.gps_cb(1.0, 2.0, (0.3, 0.0, 0.0))
    assert not v.gps_cb(5.0, 5.0)                          # subscriber unregistered after initialisation
    np.testing.assert_allclose(v.m2o[0], [1.0, 2.0, 0.0])
    v.stim_cb(10.0, synth.quat_from_rpy(0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    assert v.dr_timer().published == 1                     # first IMU sample arms the integrator
    v.dvl_cb(10.0, (1.0, 0.0, 0.0))
    o = v.dr_timer()
    assert o.used_dvl == 1 and o.lin_vel[0] == 1.0
    v.dvl_cb(10.02, (1.6, 0.0, 0.0))                       # |vx| gate -> thrust model (zero rpm -> zero velocity)
    o = v.dr_timer()
    assert o.used_dvl == 0 and o.lin_vel[0] == 0.0
    od = v.to_odom(o, 10.04)
    assert od.stamp == 10.04 and od.q[3] == o.q[3]
    v.close()
