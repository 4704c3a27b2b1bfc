"""Host-side sanitizers (SURVEY 5 "race detection / sanitizers"; VERDICT r4 missing 2).  GPU AddressSanitizer is not
available on this pool, and the reference is racy by construction (auv_pf.py:126,202-211,264-285: three rospy threads
enter the filter unlocked) -- so everything of the host side that needs no device is built with plain g++ under
AddressSanitizer + UBSan and ThreadSanitizer (`make -C smarc_navigation_amd/csrc host-asan host-tsan`) and run here:

  * the device-free arithmetic of the library (mcl_host_pure.h: transfer plan of the resample exchange, matrix_from_tf,
    euler_from_quat, host Philox; mcl_dr_impl.h: every callback of the dead-reckoning integrator; pf_core.hpp's
    parsers) under random and hostile inputs; round 6: mcl_halfedge.h -- the half-edge table of the TIN sweep built from
    random jittered meshes in random order and mixed windings, its invariants checked record by record (Morton
    permutation, every triangle counter-clockwise, next_a / next_b name the half-edges that run the shared edges the
    other way round, border codes) and WALKED on the CPU by the kernel's own rule from random points along random
    vertical planes to the map border (every exit edge cut by the plane, s never decreases); meshes with three faces on
    an edge or a folded pair refused;
  * the roscpp node, compiled UNCHANGED against the stand-in ROS, over a recording engine whose state is unsynchronised
    like the real handle's, with odometry / GPS + dive / pings + detections / timer arriving from four threads at once:
    ThreadSanitizer must see no race (the node's mutex is what the reference lacks), ASan no overrun of any buffer
    handed across the ABI;
  * the C oracle (test infrastructure) rebuilt with ASan + UBSan and its golden tests re-run against that build."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, 'build', 'host_san')


@pytest.fixture(scope='module')
def built():
    if shutil.which('g++') is None or shutil.which('make') is None:
        pytest.skip('no g++ / make')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'smarc_navigation_amd', 'csrc'), 'host-asan', 'host-tsan'],
                          stdout=subprocess.DEVNULL)
    return SAN


@pytest.fixture(scope='module')
def scene(tmp_path_factory):
    from smarc_navigation_amd import auv_pf, synth
    d = tmp_path_factory.mktemp('san')
    z = synth.bathymetry_grid(64, 64, 1.0, (-32.0, -32.0), seed=1)
    mpath, lpath = str(d / 'map.mclgrid'), str(d / 'rocks.txt')
    auv_pf.save_mclgrid(mpath, z, (-32.0, -32.0), 1.0)
    with open(lpath, 'w') as f:
        f.write('3.0 4.0 -18.0\n-2.0 -6.0 -17.5\n')
    return mpath, lpath


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=e, timeout=timeout, cwd=ROOT)
    for marker in ('AddressSanitizer', 'ThreadSanitizer', 'runtime error', 'LeakSanitizer'):
        assert marker not in p.stdout, p.stdout[-4000:]
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def test_device_free_host_arithmetic_under_asan_and_ubsan(built):
    out = _run([os.path.join(built, 'host_pure_asan')])
    assert 'host_pure_driver: ok' in out


def test_node_callbacks_from_four_threads_are_race_free_under_tsan(built, scene):
    out = _run([os.path.join(built, 'node_tsan')] + list(scene), env={'TSAN_OPTIONS': 'halt_on_error=1'})
    assert 'threads_scenario: ok, 400 + 400 publications from four threads' in out


def test_node_hands_the_abi_exactly_the_buffers_it_promises_under_asan(built, scene):
    out = _run([os.path.join(built, 'node_asan')] + list(scene))
    assert 'threads_scenario: ok' in out


def test_oracle_golden_tests_pass_on_its_asan_ubsan_build(built):
    """The checker itself: oracle/mcl_oracle.c with -fsanitize=address,undefined, the golden / KAT tests re-run in a child
    Python with libasan preloaded (leak checking off: CPython does not free at exit)."""
    gcc = shutil.which('gcc')
    asan = subprocess.check_output([gcc, '-print-file-name=libasan.so'], universal_newlines=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('no libasan.so beside gcc')
    env = {'LD_PRELOAD': asan, 'ASAN_OPTIONS': 'detect_leaks=0', 'MCL_ORACLE_LIB': os.path.join(built, 'libmcl_oracle_asan.so'),
           'OMP_NUM_THREADS': '2'}
    out = _run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider', 'tests/test_oracle_golden.py',
                'tests/test_oracle_mbes.py', 'tests/test_oracle_mbes_golden.py'], env=env, timeout=900)
    assert ' passed' in out and 'failed' not in out, out[-2000:]
