"""A pure-C program linked against libmcl_hip.so (examples/mcl_c_example.c) gives bit-identical
numbers to the ctypes path: the C ABI is the boundary, Python is only one of its clients."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_matches_ctypes_path(tmp_path):
    exe = str(tmp_path / 'mcl_c_example')
    libdir = os.path.join(ROOT, 'smarc_navigation_amd')
    subprocess.check_call(['gcc', '-std=c11', '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'mcl_c_example.c'), '-L' + libdir, '-lmcl_hip',
                           '-Wl,-rpath,' + libdir, '-lm', '-o', exe])
    out = subprocess.check_output([exe]).decode().strip().splitlines()[-1]
    c_vals = np.array([float(x) for x in out.split()])
    from smarc_navigation_amd import engine
    m2o = np.identity(4)
    c, s = np.cos(0.3), np.sin(0.3)
    m2o[:2, :2] = [[c, -s], [s, c]]
    m2o[:3, 3] = [2.0, -1.0, 0.0]
    e = engine.Engine(4096, init_cov=[0.5, 0.5, 0, 0, 0, 0.01], process_cov=[1e-4, 1e-4, 0, 0, 0, 1e-6],
                      resample_cov=[0.01, 0.01, 0, 0, 0, 1e-5], meas_std=1.5, m2o=m2o, seed=42)
    e.init_particles()
    for k in range(25):
        e.predict([1.0, 0.05, 0.0], 0.02, [0.0, 0.0, 0.0, 1.0], -2.0, 0.02, stamp=100.0 + 0.02 * (k + 1))
    e.update_gps(2.6, -0.8)
    e.resample()
    mean, yaw, cov = e.mean_cov()
    py_vals = np.array([mean[0], mean[1], mean[2], yaw, cov[0], cov[1], cov[4]])
    # the rotation part of m2o comes from mcl_matrix_from_tf in C and from cos/sin here: allow that ulp
    np.testing.assert_allclose(c_vals, py_vals, rtol=1e-12, atol=1e-12)
