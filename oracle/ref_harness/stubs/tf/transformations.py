"""Own restatement of the handful of tf.transformations helpers the reference PF calls
(auv_particle.py:8-10, auv_pf.py:17), default axes 'sxyz', quaternion order (x, y, z, w).
The real library (ROS geometry/tf) is absent from this image; these follow the published
math of static-xyz Euler angles and are cross-checked against scipy.spatial.transform in
tests/test_oracle_golden.py.  TEST INFRASTRUCTURE ONLY."""
import math
import numpy as np

_EPS = np.finfo(float).eps * 4.0


def identity_matrix():
    return np.identity(4)


def translation_matrix(direction):
    m = np.identity(4)
    m[:3, 3] = direction[:3]
    return m


def translation_from_matrix(matrix):
    return np.array(matrix, copy=False)[:3, 3].copy()


def quaternion_from_euler(ai, aj, ak, axes='sxyz'):
    assert axes == 'sxyz'
    hr, hp, hy = ai / 2.0, aj / 2.0, ak / 2.0
    cr, sr = math.cos(hr), math.sin(hr)
    cp, sp = math.cos(hp), math.sin(hp)
    cy, sy = math.cos(hy), math.sin(hy)
    q = np.empty(4)
    q[0] = cp * (sr * cy) - sp * (cr * sy)
    q[1] = cp * (sr * sy) + sp * (cr * cy)
    q[2] = cp * (cr * sy) - sp * (sr * cy)
    q[3] = cp * (cr * cy) + sp * (sr * sy)
    return q


def quaternion_matrix(quaternion):
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < _EPS:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array((
        (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
        (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
        (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
        (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)


def euler_from_matrix(matrix, axes='sxyz'):
    assert axes == 'sxyz'
    M = np.array(matrix, dtype=np.float64, copy=False)[:3, :3]
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > _EPS:
        ax = math.atan2(M[2, 1], M[2, 2])
        ay = math.atan2(-M[2, 0], cy)
        az = math.atan2(M[1, 0], M[0, 0])
    else:
        ax = math.atan2(-M[1, 2], M[1, 1])
        ay = math.atan2(-M[2, 0], cy)
        az = 0.0
    return ax, ay, az


def euler_from_quaternion(quaternion, axes='sxyz'):
    return euler_from_matrix(quaternion_matrix(quaternion), axes)


def quaternion_from_matrix(matrix):
    M = np.array(matrix, dtype=np.float64, copy=False)[:4, :4]
    t = np.trace(M)
    q = np.empty(4)
    if t > M[3, 3]:
        q[3] = t
        q[2] = M[1, 0] - M[0, 1]
        q[1] = M[0, 2] - M[2, 0]
        q[0] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    return q


def rotation_matrix(angle, direction, point=None):
    d = np.array(direction[:3], dtype=np.float64)
    d /= np.linalg.norm(d)
    c, s = math.cos(angle), math.sin(angle)
    R = np.diag([c, c, c]) + np.outer(d, d) * (1.0 - c)
    d *= s
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    M = np.identity(4)
    M[:3, :3] = R
    return M


def quaternion_multiply(quaternion1, quaternion0):
    x0, y0, z0, w0 = quaternion0
    x1, y1, z1, w1 = quaternion1
    return np.array([x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                     -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
                     -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0])
