"""The roscpp node (ros/auv_particle_filter_hip/src/auv_pf_node.cpp), compiled UNCHANGED with plain g++ against
stand-in roscpp / tf2_ros / message headers (tests/ros_stubs_cpp: the real signatures over an in-process "master") and
run in one process: parameters -> tf lookup -> map file -> publishers / subscribers / timer -> three odometry messages,
a LaserScan ping, the same ping as a PointCloud2, /dive, a GPS fix, one timer tick -> what it publishes.

CPU (`-m "not gpu"`): it compiles with -Wall -Werror and links against libmcl_hip.so; a failed map <- odom lookup ends
the node with an error before anything touches a GPU (auv_pf.py:84-87).
GPU (`-m gpu`): its publications equal, number for number, those of the ROS-free core driven directly
(examples/pf_core_example.cpp, itself checked against the Python mirror in tests/test_cpp_core.py)."""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ros', 'auv_particle_filter_hip')


def _gxx():
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    return gxx


def _flags():
    return ['-std=c++14', '-Wall', '-Werror', '-Wno-comment', '-I' + os.path.join(ROOT, 'include'),
            '-I' + os.path.join(PKG, 'include')]


def _link():
    lib = os.path.join(ROOT, 'smarc_navigation_amd')
    return ['-L' + lib, '-lmcl_hip', '-Wl,-rpath,' + lib]


def _build_node(tmp_path):
    exe = str(tmp_path / 'auv_pf_node')
    stubs = os.path.join(ROOT, 'tests', 'ros_stubs_cpp')
    subprocess.check_call([_gxx()] + _flags() + ['-I' + stubs, os.path.join(PKG, 'src', 'auv_pf_node.cpp'),
                                                 os.path.join(stubs, 'scenario.cpp')] + _link() + ['-o', exe])
    return exe


def _build_example(tmp_path):
    exe = str(tmp_path / 'pf_core_example')
    subprocess.check_call([_gxx()] + _flags() + [os.path.join(ROOT, 'examples', 'pf_core_example.cpp')] + _link() + ['-o', exe])
    return exe


def test_roscpp_node_compiles_against_the_stand_in_ros_and_fails_cleanly_without_tf(tmp_path):
    exe = _build_node(tmp_path)
    assert subprocess.call([exe], stderr=subprocess.DEVNULL) == 2          # the scenario's usage message
    (tmp_path / 'ranges.txt').write_text('10.0\n10.5\n11.0\n')
    p = subprocess.run([exe, str(tmp_path / 'missing.mclgrid'), str(tmp_path / 'ranges.txt'), 'no-tf'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert p.returncode == 1 and 'Could not lookup transform map to sam/odom' in p.stderr and p.stdout == ''


def _landmark_yaml(path):
    """A landmark map in the form the reference's map provider parses (map_provider_node.py:43-52: a Gazebo model list)."""
    rows = [(3.4, 3.6, -18.0), (-1.6, -6.4, -17.5), (20.0, 20.0, -19.0), (5.0, -5.0, -95.0)]
    with open(path, 'w') as f:
        f.write('models:\n')
        for k, (x, y, z) in enumerate(rows):
            f.write('- name: rock_%d\n  position:\n    x: %r\n    y: %r\n    z: %r\n  orientation: {x: 0, y: 0, z: 0, w: 1}\n' % (k, x, y, z))
    return rows


def test_landmark_yaml_is_read_alike_by_the_cpp_core_and_the_python_mirror(tmp_path):
    """map_provider_node.py:35-56: every model's position, the ones below rocks_depth kept."""
    from smarc_navigation_amd import auv_pf
    path = str(tmp_path / 'map.yaml')
    rows = _landmark_yaml(path)
    np.testing.assert_allclose(auv_pf.load_landmark_file(path), np.array(rows))
    np.testing.assert_allclose(auv_pf.load_landmark_file(path, rocks_depth=-90.0), np.array(rows[3:]))
    src = str(tmp_path / 'lm.cpp')
    with open(src, 'w') as f:
        f.write('#include "auv_particle_filter_hip/pf_core.hpp"\nint main(int c, char** v) { std::vector<double> x; std::string e;\n'
                '  if (!auv_pf_hip::load_landmark_file(v[1], std::atof(v[2]), x, e)) { std::puts(e.c_str()); return 1; }\n'
                '  for (double d : x) std::printf("%.17g\\n", d);\n  return 0; }\n')
    exe = str(tmp_path / 'lm')
    subprocess.check_call([_gxx()] + _flags() + [src, '-o', exe])
    got = np.array(subprocess.check_output([exe, path, '1e300'], universal_newlines=True).split(), float).reshape(-1, 3)
    np.testing.assert_array_equal(got, np.array(rows))
    got = np.array(subprocess.check_output([exe, path, '-90'], universal_newlines=True).split(), float).reshape(-1, 3)
    np.testing.assert_array_equal(got, np.array(rows[3:]))
    assert subprocess.call([exe, path, '-1000'], stdout=subprocess.DEVNULL) == 1   # nothing left: an error, not an empty map


@pytest.mark.gpu
@pytest.mark.parametrize('landmarks', [False, True])
def test_roscpp_node_publishes_what_the_core_publishes(landmarks, tmp_path):
    from smarc_navigation_amd import auv_pf, engine as eng, synth
    node, example = _build_node(tmp_path), _build_example(tmp_path)
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
    one.close()
    rpath = str(tmp_path / 'ranges.txt')
    np.savetxt(rpath, ranges, fmt='%.9g')
    extra = []
    if landmarks:   # BASELINE config 5 through the node: a landmark map + one detection message with the first ping
        lpath = str(tmp_path / 'landmarks.yaml')
        _landmark_yaml(lpath)
        extra = [lpath]
        plain = subprocess.check_output([example, mpath, rpath], universal_newlines=True).split()
    ref = subprocess.check_output([example, mpath, rpath] + extra, universal_newlines=True).split()
    if landmarks:
        assert ref[:2] != plain[:2]   # the detections moved the estimate: they were used
    out = subprocess.run([node, mpath, rpath] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split('\n')
    got = lines[0].split()
    # mean x y z, covariance xx xy yy, quaternion z, number of pose values: the same calls, the same numbers
    for k in (0, 1, 2, 4, 5, 6, 7, 8):
        assert got[k] == ref[k], (k, got, ref)
    assert abs(float(got[3]) - math.cos(0.5 * float(ref[3]))) < 1e-15      # orientation.w of a level vehicle
    assert math.hypot(float(got[0]), float(got[1])) < 1.0                   # the pings and the fix were used
    # frames, tf odom -> base with z = 0 (auv_pf.py:256-260), topics, queue sizes, the 10 Hz timer, three spinner threads
    assert lines[1] == 'frames sam/odom sam/base_link sam/odom | tf sam/odom sam/base_link %s %s 0' % (got[0], got[1])
    assert lines[2] == ('subs /dive /sam/dr/gps /sam/dr/odom /sam/mbes_cloud%s /sam/mbes_scan | pubs /sam/dr/odom_corrected:100 '
                        '/sam/dr/particle_poses:10 | timers 0.100 | spinner 3 | node auv_pf' % (' /sam/mbes_detections' if landmarks else ''))
    assert 'Particle filter class successfully created' in out.stderr
