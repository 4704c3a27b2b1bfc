"""The assembly merge loop of the fan sweep (mcl_sweep.h: sweep_merge_asm -- lattice walk, TIN; the height grid's cell walk
runs the same scenes: its beam loop is the compiler's in both builds since the round-5 assembly version of it, bit-equal
but 7 % slower, was dropped) against the compiler's build of the same loop: a second library built with -DSWEEP_MERGE_CXX=1 (csrc/Makefile: libmcl_hip_cxxmerge.so, every
kernel takes the C++ loop), the same clouds through both, log-weights BIT FOR BIT.  Collapsed and wide clouds (lanes of a
wave at the same / at different beams), a cloud hanging over the map border (slices that end there, hand-overs to the
general kernel), odd beam counts, invalid beams, the lattice mesh and the irregular TIN.

A build-vs-build A/B, not a parity test: the file sorts LAST (it sat in front of the reference-pinned
parity tests and a one-ulp difference under `pytest -x` cut 140 of them off).  Since round 4 every particle's result is
independent of the hand-over order (tests/test_gpu_determinism.py), so a difference here can only be the loop itself."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'smarc_navigation_amd', 'csrc')

_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from smarc_navigation_amd import engine as eng, synth
out = {}
origin = (-64.0, -354.0)
z = synth.bathymetry_grid(708, 708, 1.0, origin, seed=3)
for kind in ('mesh', 'tin', 'grid'):
    if kind == 'mesh':
        verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    elif kind == 'tin':
        verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7)
    for n, spread, B, x0 in ((131072, 0.05, 512, 100.0), (65536, 30.0, 301, 100.0), (60000, 300.0, 128, 100.0), (20000, 1.0, 511, -40.0)):
        rs = np.random.RandomState(3)
        soa = rs.randn(6, n) * np.array([spread, spread, 0.3, 0.03, 0.03, 1.0])[:, None]
        soa[0] += x0
        soa[2] -= 5.0
        e = eng.Engine(n, rng_mode=eng.RNG_REPLAY)
        if kind == 'grid':
            e.set_map_grid(z, origin, 1.0)
        else:
            e.set_map_mesh(verts, tris)
        e.set_particles(soa)
        ba = synth.beam_angles(B)
        ranges = (25.0 / np.cos(ba) + 0.1 * rs.randn(B)).astype(np.float32)
        ranges[::9] = 0.0
        e.update_mbes(ranges, ba, 0.2, 100.0)
        path = e.mbes_last_path()
        assert path[0] == 1, path
        out['%%s_%%d_%%g_%%d' %% (kind, n, spread, B)] = e.get_log_weights()
        out['path_%%s_%%d' %% (kind, n)] = np.array(path)
        e.close()
    # fused steps: straight after a predict the library can PROVE that no beam reaches the seabed beyond r_max and the
    # assembly loop leaves the clamp out (mcl_host_update.h: sweep_noclamp); the C++ loop always clamps
    # (r_max 100 m: every beam reaches the map's lowest point well inside it; 44 m: the outer beams do not -- the proof
    #  fails, the clamp stays and binds on real beams; 30 m: most of the fan is beyond it)
    n, B = 131072, 512
    stream = synth.odom_stream(12)
    ba = synth.beam_angles(B)
    for r_max in (100.0, 44.0, 30.0):
        e = eng.Engine(n, seed=5, init_cov=[0.5, 0.5, 0, 0, 0, 0.01], process_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5],
                       resample_cov=[1e-3, 1e-3, 0, 0, 0, 1e-5])
        if kind == 'grid':
            e.set_map_grid(z, origin, 1.0)
        else:
            e.set_map_mesh(verts, tris)
        e.init_particles()
        rs = np.random.RandomState(11)
        for k in range(12):
            ranges = np.minimum(20.0 / np.cos(ba) + 0.3 * rs.randn(B), r_max - 0.5).astype(np.float32)
            ranges[5::17] = 0.0
            e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges, ba, 0.2, r_max)
        e.sync()
        assert e.mbes_last_path()[0] == 1
        out['steps_%%s_%%g' %% (kind, r_max)] = e.get_particles()
        out['hist_%%s_%%g' %% (kind, r_max)] = e.mean_history(12)
        e.close()
np.savez(sys.argv[1], **out)
'''


def test_assembly_merge_loop_equals_the_compilers_bit_for_bit(tmp_path):
    variant = os.path.join(ROOT, 'smarc_navigation_amd', 'libmcl_hip_cxxmerge.so')   # (built by __graft_entry__.build())
    if not os.path.exists(variant):
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        if not os.path.exists(hipcc):
            pytest.skip('no prebuilt variant and no hipcc on this box')
        subprocess.check_call(['make', '-C', CSRC, '../libmcl_hip_cxxmerge.so', 'HIPCC=' + hipcc])
    child = str(tmp_path / 'child.py')
    with open(child, 'w') as f:
        f.write(_CHILD % {'root': ROOT})
    res = {}
    for name, lib in (('asm', None), ('cxx', variant)):
        env = dict(os.environ)
        env.pop('MCL_LIB', None)
        if lib:
            env['MCL_LIB'] = lib
        out = str(tmp_path / (name + '.npz'))
        env['MCL_DEBUG_WORK'] = '1'
        p = subprocess.run([sys.executable, child, out], env=env, stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = np.load(out)
        # the fused steps really ran without the clamp (and the set_particles clouds, whose depths differ, with it)
        assert 'proved idle: skipped' in p.stderr and 'r_max kept' in p.stderr, p.stderr[-2000:]
    keys = [k for k in res['asm'].files if not k.startswith('path_')]
    assert len(keys) == 30
    handed = 0
    for k in res['asm'].files:
        a, c = res['asm'][k], res['cxx'][k]
        if k.startswith('path_'):
            assert np.array_equal(a, c), k
            handed += int(a[1])
            continue
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, c), '%s: %d of %d log-weights differ (max %.3e)' % (k, (a != c).sum(), a.size, np.abs(a - c).max())
    assert handed > 1000   # the border clouds really went through the hand-over list and the general kernel
