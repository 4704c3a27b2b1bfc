#!/usr/bin/env python3
"""Independent golden vectors for the MBES / mesh / landmark part of the path.

TEST INFRASTRUCTURE.  The reference has no MBES measurement model, so these rows cannot be
pinned to it.  What this script provides instead is a SECOND, independent statement of the definition in
DESIGN.md section 5, written in plain numpy/scipy from that text alone -- it imports nothing from
oracle/mcl_oracle.c, oracle/oracle.py or the kernels and shares no traversal, no root formula and no
acceleration structure with them:

  * height grid:   f(t) = z_ray(t) - bilinear_height(x(t), y(t)) is SAMPLED densely along the ray
                   (5 mm steps); the first sample with f <= 0 brackets the hit and
                   scipy.optimize.brentq refines it.  No cell walk, no patch quadratic.
  * triangle mesh: Moller-Trumbore against EVERY triangle of the mesh (vectorised), smallest t wins.
                   No cell grid, no plane-form records.
  * landmarks:     all pairwise distances, sort, gate, log-sum-exp.
  * beam log-likelihood and the sensor pose chain are written out again here.

Rays that graze the surface closer than the sampling can resolve are flagged in `ok` (False) and skipped
by the tests: a dense sample is not an exact root finder, and saying so is better than a loose tolerance.

Nearest reference analogues of the model (for the reader, not used here): landmark -> sensor frame
auv_ekf_slam/src/correspondence_obj_mbes.cpp:26-35; Gaussian likelihood
auv_ekf_localization/src/correspondence_obj.cpp:80-97; LaserScan beam geometry
mbes_processors/mbes_toy_processor/src/toy_mbes_manipulator.cpp:69-73.

Re-run:  python oracle/ref_harness/gen_golden_mbes.py [case ...]    (default: every case; writes tests/golden/<case>.npz and its
         hash into tests/golden/MANIFEST.json; MCL_GOLDEN_OUT=dir writes there instead -- the regeneration check)
"""
import os
import sys

import numpy as np
from scipy.optimize import brentq

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from smarc_navigation_amd import synth  # noqa: E402  (terrain and beam-angle DATA only)

import manifest  # noqa: E402  (this directory: where to write, and the fixture hashes)

OUT = manifest.golden_dir()
WRITTEN = []
DT = 0.005        # coarse sampling step along a ray, metres
G_COARSE = 0.01   # = L * DT / 2 with L = 4 >= |df/dt|: no crossing hides between two samples above it
FINE = 1e-4       # fine sampling step inside the runs of coarse samples below G_COARSE
CLEAR = 1e-3      # a ray passing this close above the surface without crossing is "ambiguous"


# ----------------------------------------------------------------------------- geometry, written out
def rot(roll, pitch, yaw):
    """Static-xyz Euler angles: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz.dot(Ry).dot(Rx)


def homog(p6):
    M = np.identity(4)
    M[:3, :3] = rot(p6[3], p6[4], p6[5])
    M[:3, 3] = p6[:3]
    return M


def sensor_in_map(m2o, pose6, off6):
    """map <- sensor = (map <- odom) (odom <- base) (base <- sensor)."""
    return m2o.dot(homog(pose6)).dot(homog(off6))


def beam_dirs_sensor(angles):
    a = np.asarray(angles, dtype=np.float64)
    return np.stack([np.zeros_like(a), np.sin(a), -np.cos(a)], axis=1)


# ----------------------------------------------------------------------------- height grid by sampling
def bilinear(z, origin, res, x, y):
    """Height of the bilinear surface through the nodes; NaN outside the node lattice."""
    nx, ny = z.shape
    u, v = (x - origin[0]) / res, (y - origin[1]) / res
    inside = (u >= 0) & (u <= nx - 1) & (v >= 0) & (v <= ny - 1)
    i = np.clip(np.floor(u).astype(int), 0, nx - 2)
    j = np.clip(np.floor(v).astype(int), 0, ny - 2)
    a, b = u - i, v - j
    h = z[i, j] * (1 - a) * (1 - b) + z[i + 1, j] * a * (1 - b) + z[i, j + 1] * (1 - a) * b + z[i + 1, j + 1] * a * b
    return np.where(inside, h, np.nan)


def grid_range(z, origin, res, o, d, r_max):
    """(range, unambiguous) of one ray against the bilinear surface.

    Coarse pass: f sampled every DT.  Two neighbouring samples that are both above G_COARSE cannot hide a
    crossing (|f'| <= L = 4 on these terrains, and L * DT / 2 = G_COARSE), so only the runs of samples with
    f <= G_COARSE are looked at again, at FINE = 0.1 mm; the first fine sample with f <= 0 brackets the hit
    for brentq.  The answer is flagged ambiguous when the ray passes within CLEAR = 1 mm of the surface
    without crossing it before the reported hit (or at all): the expected range is discontinuous there."""
    zz = z.astype(np.float64)

    def F(t):
        return (o[2] + t * d[2]) - bilinear(zz, origin, res, o[0] + t * d[0], o[1] + t * d[1])

    t = np.unique(np.minimum(np.arange(0.0, r_max + DT, DT), r_max))
    f = F(t)
    if np.isnan(f[0]):
        return r_max, False  # sensor off the map: not part of these fixtures
    out = np.flatnonzero(np.isnan(f))
    end = out[0] if out.size else f.size  # samples [0, end) are over the map
    leaves = end < f.size
    t, f = t[:end], f[:end]
    if f[0] <= 0.0:
        return 0.0, True  # the sensor itself is at or below the seabed
    near = f <= G_COARSE
    if leaves:
        near[-1] = True  # the sliver between the last sample over the map and the border
    idx = np.flatnonzero(near)
    unamb = True
    if idx.size:
        runs = np.split(idx, np.flatnonzero(np.diff(idx) > 1) + 1)
        for run in runs:
            lo = t[max(run[0] - 1, 0)]
            hi = min(t[min(run[-1] + 1, end - 1)] + (DT if (leaves and run[-1] == end - 1) else 0.0), r_max)
            tf = np.arange(lo, hi + FINE, FINE)
            tf = tf[tf <= r_max]
            ff = F(tf)
            bad = np.flatnonzero(np.isnan(ff))
            if bad.size:
                tf, ff = tf[:bad[0]], ff[:bad[0]]
            if ff.size == 0:
                continue
            below = np.flatnonzero(ff <= 0.0)
            stop = below[0] if below.size else ff.size
            # local minima of the clearance before the crossing (the final monotone approach has none)
            seg = ff[:stop]
            if seg.size >= 3:
                mins = (seg[1:-1] < seg[:-2]) & (seg[1:-1] <= seg[2:])
                if np.any(seg[1:-1][mins] < CLEAR):
                    unamb = False
            if not below.size and seg.size and (seg[-1] < CLEAR) and (seg.size < 2 or seg[-1] < seg[-2]) and bad.size:
                unamb = False  # leaves the map while skimming the surface
            if below.size:
                k = below[0]
                if k == 0:
                    return float(tf[0]), unamb
                root = brentq(lambda s_: float(F(np.array(s_))), tf[k - 1], tf[k], xtol=1e-13, rtol=1e-15)
                return float(root), unamb
    return r_max, unamb


# ----------------------------------------------------------------------------- mesh by brute force
def mesh_range(verts, tris, o, d, r_max):
    """Moller-Trumbore of one ray against every triangle; (range, unambiguous)."""
    v0 = verts[tris[:, 0]].astype(np.float64)
    e1 = verts[tris[:, 1]].astype(np.float64) - v0
    e2 = verts[tris[:, 2]].astype(np.float64) - v0
    p = np.cross(np.broadcast_to(d, e2.shape), e2)
    det = np.sum(e1 * p, axis=1)
    good = np.abs(det) > 1e-14
    inv = np.where(good, 1.0 / np.where(good, det, 1.0), 0.0)
    s = o[None, :] - v0
    u = np.sum(s * p, axis=1) * inv
    q = np.cross(s, e1)
    v = np.sum(np.broadcast_to(d, q.shape) * q, axis=1) * inv
    t = np.sum(e2 * q, axis=1) * inv
    hit = good & (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0) & (t <= r_max)
    if not hit.any():
        # near misses along an edge (|bary| tiny outside) would make the answer depend on rounding
        near = good & (u >= -1e-9) & (v >= -1e-9) & (u + v <= 1 + 1e-9) & (t >= 0) & (t <= r_max)
        return r_max, not near.any()
    return float(np.min(t[hit])), True


# ----------------------------------------------------------------------------- likelihoods
def beam_loglik(ranges, expected, sigma):
    r = np.asarray(ranges, dtype=np.float64)
    valid = r > 0  # NaN fails the comparison
    res = (r[valid] - expected[valid]) / sigma
    return -0.5 * np.sum(res * res) - np.count_nonzero(valid) * np.log(sigma * np.sqrt(2.0 * np.pi))


def landmark_loglik(m2o, pose6, off6, landmarks, det, sigma, k, gate):
    M = sensor_in_map(m2o, pose6, off6)
    total, used = 0.0, 0
    for zd in det:
        if np.any(np.isnan(zd)):
            continue
        used += 1
        p = M[:3, :3].dot(zd) + M[:3, 3]
        maha = np.sum((landmarks - p[None, :]) ** 2, axis=1) / (sigma * sigma)
        near = np.sort(maha[maha < gate])[:k]
        total += np.log(np.sum(np.exp(-0.5 * near))) if near.size else -0.5 * gate
    return total - used * (1.5 * np.log(2.0 * np.pi) + 3.0 * np.log(sigma))


# ----------------------------------------------------------------------------- scenes
def poses_over(rs, n, centre, spread):
    p = rs.randn(n, 6) * np.asarray(spread)[None, :]
    p[:, :3] += np.asarray(centre)[None, :]
    return p


def cast_all(kind, amap, m2o, poses, off6, angles, r_max):
    dirs = beam_dirs_sensor(angles)
    exp = np.zeros((len(poses), len(angles)))
    ok = np.zeros(exp.shape, dtype=bool)
    for i, p6 in enumerate(poses):
        M = sensor_in_map(m2o, p6, off6)
        o = M[:3, 3]
        for b, ds in enumerate(dirs):
            d = M[:3, :3].dot(ds)
            if kind == 'grid':
                exp[i, b], ok[i, b] = grid_range(amap['z'], amap['origin'], amap['res'], o, d, r_max)
            else:
                exp[i, b], ok[i, b] = mesh_range(amap['verts'], amap['tris'], o, d, r_max)
    return exp, ok


def make_case(name, kind, amap, m2o, poses, off6, n_beams, half_swath, sigma, r_max, seed):
    rs = np.random.RandomState(seed)
    angles = synth.beam_angles(n_beams, half_swath)
    exp, ok = cast_all(kind, amap, m2o, poses, off6, angles, r_max)
    ranges = (exp[0] + sigma * rs.randn(n_beams)).astype(np.float32)
    ranges[::9] = 0.0          # invalid beams are skipped
    ranges[4] = np.nan
    # log-likelihood of every particle whose beams are all unambiguous (NaN otherwise)
    lw = np.array([beam_loglik(ranges, exp[i], sigma) if ok[i].all() else np.nan for i in range(len(poses))])
    d = dict(kind=kind, m2o=m2o, poses=poses, sensor_offset=np.asarray(off6, dtype=np.float64),
             beam_angles=angles, sigma=sigma, r_max=r_max, expected=exp, ok=ok, ranges=ranges, lw=lw)
    if kind == 'grid':
        d.update(z=amap['z'], origin=np.asarray(amap['origin'], dtype=np.float64), res=amap['res'])
    else:
        d.update(verts=amap['verts'], tris=amap['tris'])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    WRITTEN.append(name + '.npz')
    print('%-22s rays %5d  unambiguous %5d  r_max %4d  particles with lw %d/%d' % (
        name, exp.size, int(ok.sum()), int((exp >= r_max).sum()), int(np.isfinite(lw).sum()), len(poses)))


# ----------------------------------------------------------------------------- the cases
# Every case is self-contained: its poses come from ITS OWN seeded stream (pose_seed), its range noise from another
# (noise_seed) -- no case depends on which cases ran before it (round 5: one shared stream per group; a case inserted in
# the middle moved the poses of every case after it, and a fixture that was not regenerated no longer followed from
# this script -- VERDICT r5 weak 1).  tests/test_golden_manifest.py re-runs a subset into a scratch directory on every CPU
# run and compares bit for bit; `python oracle/ref_harness/manifest.py --regen-check` re-runs them all.
M2O = synth.rigid_matrix(1.5, -2.0, 0.3, 0.0, 0.0, 0.25)
OFF = [0.3, -0.1, -0.2, 0.01, -0.02, 0.05]
IDENT = np.identity(4)
ZERO = [0.0] * 6


def _grid96():
    return synth.bathymetry_grid(96, 96, 1.0, (-48.0, -48.0), seed=21)


def _grid48():
    return synth.bathymetry_grid(48, 48, 1.0, (-24.0, -24.0), seed=23)


def _grid144():
    return synth.bathymetry_grid(144, 144, 1.0, (-72.0, -72.0), seed=31)


def case_grid_interior():
    # grid, converged cloud in the interior, sensor offset + map<-odom transform
    rs = np.random.RandomState(101)
    make_case('mbes_grid_interior', 'grid', dict(z=_grid96(), origin=(-48.0, -48.0), res=1.0), M2O,
              poses_over(rs, 24, (2.0, -3.0, -2.0), (2.0, 2.0, 0.3, 0.05, 0.05, 3.0)), OFF, 64, np.pi / 3, 0.2, 80.0, 1)


def case_grid_rough():
    # grid, rough terrain (x4 relief), wide swath and large roll/pitch: grazing rays, occlusion
    rs = np.random.RandomState(102)
    zr = (-20.0 + 4.0 * (_grid96().astype(np.float64) + 20.0)).astype(np.float32)
    make_case('mbes_grid_rough', 'grid', dict(z=zr, origin=(-48.0, -48.0), res=1.0), IDENT,
              poses_over(rs, 24, (0.0, 0.0, -2.0), (6.0, 6.0, 0.5, 0.35, 0.2, 3.0)), ZERO, 96, 1.2, 0.2, 100.0, 2)


def case_grid_border():
    # grid, cloud straddling the map border: rays leave the map (r_max), coarse resolution 2 m
    rs = np.random.RandomState(103)
    z2 = synth.bathymetry_grid(64, 64, 2.0, (-30.0, -64.0), seed=22)
    make_case('mbes_grid_border', 'grid', dict(z=z2, origin=(-30.0, -64.0), res=2.0), IDENT,
              poses_over(rs, 24, (-22.0, 40.0, -3.0), (3.0, 8.0, 0.3, 0.1, 0.1, 3.0)), ZERO, 64, 1.3, 0.3, 70.0, 3)


def case_mesh_regular():
    # mesh: the same kind of terrain triangulated (regular lattice, one diagonal), interior
    rs = np.random.RandomState(104)
    verts, tris = synth.mesh_from_grid(_grid48(), 1.0, (-24.0, -24.0))
    make_case('mbes_mesh_regular', 'mesh', dict(verts=verts, tris=tris), M2O,
              poses_over(rs, 16, (0.0, -2.0, -2.0), (2.0, 2.0, 0.3, 0.05, 0.05, 3.0)), OFF, 48, np.pi / 3, 0.2, 80.0, 4)


def case_mesh_tin():
    # mesh: irregular TIN (jittered vertices, random diagonals) with steeper relief, wide swath, border exits
    rs = np.random.RandomState(105)
    zt = (-20.0 + 3.0 * (_grid48().astype(np.float64) + 20.0)).astype(np.float32)
    vt, tt = synth.mesh_tin(zt, 1.0, (-24.0, -24.0), seed=5)
    make_case('mbes_mesh_tin', 'mesh', dict(verts=vt, tris=tt), IDENT,
              poses_over(rs, 16, (8.0, 6.0, -2.0), (5.0, 5.0, 0.4, 0.3, 0.15, 3.0)), ZERO, 48, 1.2, 0.2, 60.0, 5)


def case_landmarks_knn():
    # landmark k-NN association
    rs = np.random.RandomState(106)
    lm = synth.landmark_map(300, (-40.0, -40.0, 40.0, 40.0), (-22.0, -16.0), seed=6)
    poses = poses_over(rs, 32, (0.0, 0.0, -2.0), (1.5, 1.5, 0.2, 0.03, 0.03, 0.2))
    truth = np.array([0.2, -0.1, -2.0, 0.0, 0.0, 0.05])
    Mt = sensor_in_map(M2O, truth, OFF)
    order = np.argsort(np.sum((lm[:, :2] - Mt[:2, 3][None, :]) ** 2, axis=1))
    det = (lm[order[:12]] - Mt[:3, 3][None, :]).dot(Mt[:3, :3]) + 0.1 * rs.randn(12, 3)  # R^T (l - o)
    det[5] = np.nan                       # an invalid detection
    det[9] += (30.0, 30.0, 0.0)           # one far from every landmark: outside the gate for all particles
    out = {}
    for k in (1, 2, 4):
        out['lw_k%d' % k] = np.array([landmark_loglik(M2O, p6, OFF, lm, det, 0.4, k, 11.345) for p6 in poses])
    np.savez_compressed(os.path.join(OUT, 'landmarks_knn.npz'), m2o=M2O, poses=poses, sensor_offset=np.asarray(OFF),
                        landmarks=lm, det=det, sigma=0.4, gate=11.345, **out)
    WRITTEN.append('landmarks_knn.npz')
    print('landmarks_knn          particles %d  detections %d  landmarks %d' % (len(poses), len(det), len(lm)))


# cases sized for the fan sweep (smarc_navigation_amd/csrc/mcl_sweep.h): maps wide enough for the whole swath, so that
# the sweep really casts these poses instead of handing them to the traversal kernels
def _sweep_mesh(name, diag, pose_seed):
    rs = np.random.RandomState(pose_seed)
    verts, tris = synth.mesh_from_grid(_grid144(), 1.0, (-72.0, -72.0), diagonal=diag)
    make_case(name, 'mesh', dict(verts=verts, tris=tris), M2O,
              poses_over(rs, 12, (1.0, -2.0, -2.0), (2.5, 2.5, 0.3, 0.06, 0.06, 3.0)), OFF, 64, np.pi / 3, 0.2, 80.0, 7)


def case_mesh_sweep():
    _sweep_mesh('mbes_mesh_sweep', '00-11', 201)


def case_mesh_sweep_d2():
    _sweep_mesh('mbes_mesh_sweep_d2', '10-01', 202)


def case_grid_sweep():
    rs = np.random.RandomState(203)
    make_case('mbes_grid_sweep', 'grid', dict(z=_grid144(), origin=(-72.0, -72.0), res=1.0), M2O,
              poses_over(rs, 12, (-1.0, 2.0, -2.0), (2.5, 2.5, 0.3, 0.06, 0.06, 3.0)), OFF, 64, np.pi / 3, 0.2, 80.0, 8)


def case_tin_sweep():
    # an irregular TIN over the same terrain (jittered vertices, random diagonals): the sweep walks it by adjacency
    rs = np.random.RandomState(204)
    vt, tt = synth.mesh_tin(_grid144(), 1.0, (-72.0, -72.0), seed=9)
    make_case('mbes_tin_sweep', 'mesh', dict(verts=vt, tris=tt), M2O,
              poses_over(rs, 12, (0.0, 1.0, -2.0), (2.5, 2.5, 0.3, 0.06, 0.06, 3.0)), OFF, 64, np.pi / 3, 0.2, 80.0, 10)


def case_grid_sweep_rough():
    # stronger relief (x3): twisted patches, steeper triangles, shadows; moderate tilt so the slope bound holds
    rs = np.random.RandomState(205)
    zr = (-20.0 + 3.0 * (_grid144().astype(np.float64) + 20.0)).astype(np.float32)
    make_case('mbes_grid_sweep_rough', 'grid', dict(z=zr, origin=(-72.0, -72.0), res=1.0), IDENT,
              poses_over(rs, 12, (0.0, 0.0, -1.0), (3.0, 3.0, 0.4, 0.03, 0.03, 3.0)), ZERO, 72, 1.15, 0.2, 90.0, 9)


CASES = {'mbes_grid_interior': case_grid_interior, 'mbes_grid_rough': case_grid_rough, 'mbes_grid_border': case_grid_border,
         'mbes_mesh_regular': case_mesh_regular, 'mbes_mesh_tin': case_mesh_tin, 'landmarks_knn': case_landmarks_knn,
         'mbes_mesh_sweep': case_mesh_sweep, 'mbes_mesh_sweep_d2': case_mesh_sweep_d2, 'mbes_grid_sweep': case_grid_sweep,
         'mbes_tin_sweep': case_tin_sweep, 'mbes_grid_sweep_rough': case_grid_sweep_rough}


def main():
    names = [a for a in sys.argv[1:] if not a.startswith('--')] or sorted(CASES)
    for n in names:
        if n not in CASES:
            sys.exit('unknown case %s (cases: %s)' % (n, ', '.join(sorted(CASES))))
    for n in names:
        CASES[n]()
    manifest.record(OUT, WRITTEN, 'oracle/ref_harness/gen_golden_mbes.py', needs_reference=False)


if __name__ == '__main__':
    main()
