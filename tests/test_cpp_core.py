"""The roscpp node's ROS-free core (ros/auv_particle_filter_hip/include/auv_particle_filter_hip/pf_core.hpp) -- C++ host
code over the C ABI.  CPU: it compiles and links against libmcl_hip.so with plain g++ (the roscpp glue around it,
src/auv_pf_node.cpp, is compiled against stand-in ROS headers in tests/test_roscpp_node_stub.py).  GPU: examples/pf_core_example.cpp fed a map
file, odometry, a LaserScan ping, the same ping as points in base_frame and a GPS fix publishes what the Python
mirror publishes for the same inputs (same seed: same Philox draws)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'pf_core_example')
    subprocess.check_call([gxx, '-std=c++14', '-Wall', '-Werror', '-Wno-comment', '-I' + os.path.join(ROOT, 'include'),
                           '-I' + os.path.join(ROOT, 'ros', 'auv_particle_filter_hip', 'include'),
                           os.path.join(ROOT, 'examples', 'pf_core_example.cpp'),
                           '-L' + os.path.join(ROOT, 'smarc_navigation_amd'), '-lmcl_hip',
                           '-Wl,-rpath,' + os.path.join(ROOT, 'smarc_navigation_amd'), '-o', exe])
    return exe


def test_cpp_core_compiles_and_links_against_the_c_abi(tmp_path):
    exe = _build(tmp_path)
    assert os.access(exe, os.X_OK)
    # without its two arguments it prints the usage and exits 2 -- before anything touches a GPU
    assert subprocess.call([exe], stderr=subprocess.DEVNULL) == 2


def test_mclgrid_files_round_trip(tmp_path):
    from smarc_navigation_amd import auv_pf, synth
    z = synth.bathymetry_grid(9, 7, 0.5, (1.5, -2.0), seed=3)
    path = str(tmp_path / 'm.mclgrid')
    auv_pf.save_mclgrid(path, z, (1.5, -2.0), 0.5)
    kind, z2, origin, res = auv_pf.load_map_file(path)
    assert kind == 'grid' and np.array_equal(z2, z.astype(np.float32)) and origin == (1.5, -2.0) and res == 0.5


@pytest.mark.gpu
def test_cpp_core_publishes_what_the_python_mirror_publishes(tmp_path):
    from smarc_navigation_amd import auv_pf, engine as eng, msgs, synth
    exe = _build(tmp_path)
    origin = (-64.0, -64.0)
    z = synth.bathymetry_grid(128, 128, 1.0, origin, seed=1)
    mpath = str(tmp_path / 'map.mclgrid')
    auv_pf.save_mclgrid(mpath, z, origin, 1.0)
    B = 96
    angles = np.linspace(-1.0, 1.0, B)
    off = [0.3, 0.0, -0.1, 0.0, 0.05, 0.0]
    m2o = auv_pf.matrix_from_tf((0.5, -0.5, 0.0), (0.0, 0.0, 0.0, 1.0))
    one = eng.Engine(1, rng_mode=eng.RNG_REPLAY, m2o=m2o)
    one.set_map_grid(z, origin, 1.0)
    one.set_particles(np.array([[0.0], [0.0], [-2.0], [0.0], [0.0], [0.0]]))
    ranges = one.mbes_expected(0, 1, angles.astype(np.float32), 80.0, off)[0]
    rpath = str(tmp_path / 'ranges.txt')
    np.savetxt(rpath, ranges, fmt='%.9g')
    ranges = np.loadtxt(rpath).astype(np.float32)   # exactly what the C++ side parses
    out = subprocess.check_output([exe, mpath, rpath], universal_newlines=True).split()
    got = np.array([float(v) for v in out[:8]])
    assert int(out[8]) == 4096 * 7
    # ---- the Python mirror, same inputs
    params = {'particle_count': 4096, 'seed': 11, 'init_covariance': '[0.5, 0.5, 0.0, 0.0, 0.0, 0.01]',
              'motion_covariance': '[0.001, 0.001, 0.0, 0.0, 0.0, 0.00001]',
              'resampling_noise_covariance': '[0.01, 0.01, 0.0, 0.0, 0.0, 0.0001]', 'measurement_std': 1.0,
              'mbes_sensor_offset': '[0.3, 0.0, -0.1, 0.0, 0.05, 0.0]', 'map_grid_file': mpath}
    pf = auv_pf.auv_pf(params, m2o_mat=m2o)
    pf.start_timing(100.0)
    ainc = float(np.float32(2.0 / (B - 1)))   # (float32, as a sensor_msgs/LaserScan carries it)
    scan = msgs.LaserScan(ranges, -1.0, ainc, 80.0)
    a = -1.0 + ainc * np.arange(B)
    pts_sensor = np.stack([np.zeros(B), ranges * np.sin(a), -ranges * np.cos(a)], axis=1)
    T = auv_pf._rigid(*off)
    pc = msgs.pointcloud2_from_xyz((pts_sensor.dot(T[:3, :3].T) + T[:3, 3])[::-1], 'base')
    for k in range(3):
        om = msgs.Odometry()
        om.header.stamp = msgs.Time(100.02 + 0.02 * k)
        om.twist.twist.linear.x, om.twist.twist.angular.z = 1.0, 0.05
        om.pose.pose.position.z = -2.0
        pf.odom_callback(om)
        if k == 1:
            pf.mbes_cb(scan)
        if k == 2:
            pf.mbes_pc_cb(pc)
    pf.dive_cb(msgs.Bool(False))
    g = msgs.Odometry()
    pf.transport.utm2map = np.identity(4)
    g.pose.pose.position.x, g.pose.pose.position.y = 0.6, -0.4
    pf.gps_odom_cb(g)
    mean, yaw, cov9 = pf.update_loc_pose()
    quat = auv_pf.quaternion_from_euler(mean[3], mean[4], yaw)
    ref = np.array([mean[0], mean[1], mean[2], yaw, cov9[0], cov9[1], cov9[4], quat[2]])
    # the odometry and GPS paths are bit-identical; the two front-ends form beam angles in double (numpy) / float
    # (C++) arithmetic, so the pings differ in the last bits of a few angles
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    assert np.hypot(got[0], got[1]) < 1.0
