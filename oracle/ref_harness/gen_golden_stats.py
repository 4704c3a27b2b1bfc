#!/usr/bin/env python3
"""Golden fixture for the evaluation tool: the reference's own
auv_particle_filter/scripts/visual_tools.py (DRStatsVisualization.odom_cb :80-110, finish_hld :61-76 and the
two error series visualize() plots :127,:135) IMPORTED read-only behind ROS stand-ins and fed synchronised
(gps, dr, pf) triples.  The time synchroniser itself is ros_comm's message_filters (not part of the reference);
the stand-in hands the triples over directly.  TEST INFRASTRUCTURE; the .npz holds data only.
Re-run:  python oracle/ref_harness/gen_golden_stats.py"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, 'stubs'))
sys.path.insert(0, '/root/reference/auv_particle_filter/scripts')
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import rospy  # noqa: E402  (stub)
nm = types.ModuleType('rospy.numpy_msg')
nm.numpy_msg = lambda t: t
sys.modules['rospy.numpy_msg'] = nm
import tf  # noqa: E402  (stub)
from nav_msgs.msg import Odometry  # noqa: E402
import visual_tools as ref_vt  # noqa: E402  (reference)

from smarc_navigation_amd import synth  # noqa: E402


def odom(x, y, z):
    m = Odometry()
    m.pose.pose.position.x, m.pose.pose.position.y, m.pose.pose.position.z = float(x), float(y), float(z)
    return m


def main():
    rs = np.random.RandomState(17)
    n = 60
    stream = synth.odom_stream(n * 50)
    truth = stream['truth'][49::50, :3]
    utm2odom = synth.rigid_matrix(-651200.0, -6524300.0, 0.0, 0.0, 0.0, 0.0)
    gps_utm = np.column_stack([truth[:, 0] + 651200.0 + rs.randn(n), truth[:, 1] + 6524300.0 + rs.randn(n), np.zeros(n)])
    dr = truth + np.cumsum(0.02 * rs.randn(n, 3), axis=0)          # drifting dead reckoning
    pf = truth + 0.1 * rs.randn(n, 3)
    vt = object.__new__(ref_vt.DRStatsVisualization)              # __init__ subscribes and blocks on rospy
    vt.listener = tf.TransformListener()
    vt.filter_cnt = 1
    vt.gps_odom_vec = np.zeros((3, 1))
    vt.dr_odom_vec = np.zeros((3, 1))
    vt.pf_odom_vec = np.zeros((3, 1))
    dropped = []
    for k in range(n):
        # tf not available for three samples: the callback logs and drops the triple (visual_tools.py:108-109)
        vt.listener.utm2map = None if k in (0, 1, 20) else utm2odom
        if vt.listener.utm2map is None:
            dropped.append(k)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            vt.odom_cb(odom(*gps_utm[k]), odom(*dr[k]), odom(*pf[k]))
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        vt.finish_hld()
    printed = {}
    for line in out.getvalue().strip().splitlines():
        name, val = line.rsplit(' ', 1)
        printed[name.strip()] = float(val)
    err_pf = np.linalg.norm(vt.gps_odom_vec - vt.pf_odom_vec, axis=0)   # visual_tools.py:127
    err_dr = np.linalg.norm(vt.gps_odom_vec - vt.dr_odom_vec, axis=0)   # visual_tools.py:135
    import manifest  # (this directory: where to write, and the fixture hashes)
    out_dir = manifest.golden_dir()
    np.savez_compressed(os.path.join(out_dir, 'visual_tools_stats.npz'), gps_utm=gps_utm, dr=dr, pf=pf,
                        utm2odom=utm2odom, dropped=np.array(dropped), gps_odom_vec=vt.gps_odom_vec, dr_odom_vec=vt.dr_odom_vec,
                        pf_odom_vec=vt.pf_odom_vec, filter_cnt=vt.filter_cnt, err_pf=err_pf, err_dr=err_dr,
                        printed_names=np.array(sorted(printed)), printed_values=np.array([printed[k] for k in sorted(printed)]))
    manifest.record(out_dir, ['visual_tools_stats.npz'], 'oracle/ref_harness/gen_golden_stats.py', needs_reference=True)
    print(printed, vt.filter_cnt, vt.gps_odom_vec.shape)


if __name__ == '__main__':
    main()
