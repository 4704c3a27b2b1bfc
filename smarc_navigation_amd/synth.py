"""Synthetic input streams for tests and bench (numpy only, no GPU, no reference code).

Follows SURVEY.md section 8(d): the odometry stream is what sam_dead_reckoning's dr_node
publishes (body-frame DVL twist, IMU yaw rate, roll/pitch quaternion, pressure depth;
reference contract: sam_dead_reckoning/scripts/dr_node.py:165-246), the GPS fixes are
truth + N(0, sigma), the bathymetry is a smooth swell plus fBm noise.
"""
import math
import numpy as np


def quat_from_rpy(roll, pitch, yaw):
    """Static-xyz Euler -> quaternion (x, y, z, w); vectorised."""
    hr, hp, hy = np.asarray(roll) / 2.0, np.asarray(pitch) / 2.0, np.asarray(yaw) / 2.0
    cr, sr = np.cos(hr), np.sin(hr)
    cp, sp = np.cos(hp), np.sin(hp)
    cy, sy = np.cos(hy), np.sin(hy)
    return np.stack([sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy], axis=-1)


def wrap_pi(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def odom_stream(n_steps=3000, dt=0.02, t0=100.0, x0=0.0, y0=0.0, yaw0=0.0, z_mean=-2.0):
    """Lawn-mower-ish track.  Returns dict of arrays:
    stamp[n], v[n,3] (body frame), wz[n], q[n,4], z[n], rpy[n,3], truth[n,6] (pose in the
    odom frame after integrating sample k with the noise-free motion model)."""
    k = np.arange(n_steps)
    t = k * dt
    stamp = t0 + (k + 1) * dt
    v = np.stack([1.0 + 0.05 * np.sin(0.1 * t), 0.02 * np.sin(0.3 * t), np.zeros(n_steps)], axis=1)
    # piecewise yaw rate: straight legs with +-0.1 rad/s turns
    phase = (t % 40.0)
    leg = (np.floor(t / 40.0).astype(int)) % 2
    wz = np.where(phase > 30.0, np.where(leg == 0, 0.1, -0.1), 0.0)
    roll = 0.02 * np.sin(0.5 * t)
    pitch = 0.02 * np.cos(0.5 * t)
    z = z_mean - 0.5 * np.sin(0.05 * t)
    truth = np.zeros((n_steps, 6))
    x, y, yaw = x0, y0, yaw0
    dr_yaw = np.zeros(n_steps)
    for i in range(n_steps):
        yaw = float(wrap_pi(yaw + wz[i] * dt))
        cr, sr = math.cos(roll[i]), math.sin(roll[i])
        cp, sp = math.cos(pitch[i]), math.sin(pitch[i])
        cy, sy = math.cos(yaw), math.sin(yaw)
        vx, vy, vz = v[i] * dt
        x += cy * cp * vx + (cy * sp * sr - sy * cr) * vy + (cy * sp * cr + sy * sr) * vz
        y += sy * cp * vx + (sy * sp * sr + cy * cr) * vy + (sy * sp * cr - cy * sr) * vz
        truth[i] = (x, y, z[i], roll[i], pitch[i], yaw)
        dr_yaw[i] = yaw
    q = quat_from_rpy(roll, pitch, dr_yaw)
    return dict(stamp=stamp, dt=dt, t0=t0, v=v, wz=wz, q=q, z=z,
                rpy=np.stack([roll, pitch, dr_yaw], axis=1), truth=truth)


def rigid_matrix(tx, ty, tz, roll, pitch, yaw):
    """4x4 homogeneous transform T(t) * R(static-xyz rpy)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    m = np.identity(4)
    m[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    m[:3, 3] = (tx, ty, tz)
    return m


def gps_fixes(truth, m2o, every=50, sigma=1.0, seed=2):
    """GPS fixes in the MAP frame: m2o * truth_xy + N(0, sigma).  Returns (step_idx, xy)."""
    rs = np.random.RandomState(seed)
    idx = np.arange(every - 1, truth.shape[0], every)
    p = np.concatenate([truth[idx, :3], np.ones((idx.size, 1))], axis=1).dot(m2o.T)[:, :2]
    return idx, p + sigma * rs.randn(idx.size, 2)


def _value_noise(nx, ny, cells, rs):
    """Bilinear-interpolated lattice noise with `cells` lattice cells across the grid."""
    lat = rs.rand(cells + 2, cells + 2) * 2.0 - 1.0
    gx = np.linspace(0.0, cells, nx, endpoint=False)
    gy = np.linspace(0.0, cells, ny, endpoint=False)
    ix, iy = gx.astype(int), gy.astype(int)
    fx, fy = gx - ix, gy - iy
    fx = fx * fx * (3 - 2 * fx)
    fy = fy * fy * (3 - 2 * fy)
    a = lat[np.ix_(ix, iy)]
    b = lat[np.ix_(ix + 1, iy)]
    c = lat[np.ix_(ix, iy + 1)]
    d = lat[np.ix_(ix + 1, iy + 1)]
    FX, FY = fx[:, None], fy[None, :]
    return a * (1 - FX) * (1 - FY) + b * FX * (1 - FY) + c * (1 - FX) * FY + d * FX * FY


def bathymetry_grid(nx=512, ny=512, res=1.0, origin=(-64.0, -256.0), seed=3, depth=-20.0,
                    swell=3.0, fbm_amp=0.5):
    """Height grid z[ix, iy] (fp32, C order, x-major): depth + swell*sin(x/17)cos(y/23) + fBm.
    Node (ix, iy) sits at (origin_x + ix*res, origin_y + iy*res) in the map frame."""
    rs = np.random.RandomState(seed)
    x = origin[0] + res * np.arange(nx)
    y = origin[1] + res * np.arange(ny)
    z = depth + swell * np.sin(x[:, None] / 17.0) * np.cos(y[None, :] / 23.0)
    amp, cells, tot = 1.0, 8, 0.0
    noise = np.zeros((nx, ny))
    for _ in range(5):
        noise += amp * _value_noise(nx, ny, cells, rs)
        tot += amp
        amp *= 0.5
        cells *= 2
    z = z + fbm_amp * noise / tot * 2.0
    return np.ascontiguousarray(z, dtype=np.float32)


def mesh_from_grid(z, res, origin, diagonal='00-11'):
    """Triangulate a height grid: verts[nv,3] fp32, tris[nt,3] uint32; 2 triangles per cell,
    split along the (ix,iy)-(ix+1,iy+1) diagonal ('00-11') or the (ix+1,iy)-(ix,iy+1) one ('10-01')."""
    nx, ny = z.shape
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing='ij')
    verts = np.stack([origin[0] + res * ix, origin[1] + res * iy, z], axis=-1)
    verts = verts.reshape(-1, 3).astype(np.float32)
    v00 = (ix[:-1, :-1] * ny + iy[:-1, :-1]).reshape(-1)
    v10 = v00 + ny
    v01 = v00 + 1
    v11 = v00 + ny + 1
    if diagonal == '10-01':
        tris = np.concatenate([np.stack([v00, v10, v01], axis=1),
                               np.stack([v10, v11, v01], axis=1)], axis=0).astype(np.uint32)
    else:
        tris = np.concatenate([np.stack([v00, v10, v11], axis=1),
                               np.stack([v00, v11, v01], axis=1)], axis=0).astype(np.uint32)
    return verts, tris


def mesh_tin(z, res, origin, seed=7, jitter=0.25):
    """An irregular TIN over the same terrain: every interior node of the height grid is moved in xy by
    up to +-jitter*res (uniform; border nodes stay, so the footprint is unchanged), its height is the
    bilinear height of the grid at the new position, and every cell is split along a random diagonal.
    jitter <= 0.25 keeps every quad convex, so the triangles tile the plane without overlap.  The
    vertices no longer sit on a lattice: this is NOT a triangulated regular grid."""
    nx, ny = z.shape
    rs = np.random.RandomState(seed)
    ix, iy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64), indexing='ij')
    ju = (rs.rand(nx, ny) * 2.0 - 1.0) * jitter
    jv = (rs.rand(nx, ny) * 2.0 - 1.0) * jitter
    ju[0, :] = ju[-1, :] = 0.0
    ju[:, 0] = ju[:, -1] = 0.0
    jv[0, :] = jv[-1, :] = 0.0
    jv[:, 0] = jv[:, -1] = 0.0
    u, v = ix + ju, iy + jv
    i0 = np.clip(np.floor(u).astype(int), 0, nx - 2)
    j0 = np.clip(np.floor(v).astype(int), 0, ny - 2)
    fu, fv = u - i0, v - j0
    zd = z.astype(np.float64)
    h = (zd[i0, j0] * (1 - fu) * (1 - fv) + zd[i0 + 1, j0] * fu * (1 - fv) +
         zd[i0, j0 + 1] * (1 - fu) * fv + zd[i0 + 1, j0 + 1] * fu * fv)
    verts = np.stack([origin[0] + res * u, origin[1] + res * v, h], axis=-1).reshape(-1, 3).astype(np.float32)
    cx, cy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing='ij')
    v00 = (cx * ny + cy).reshape(-1)
    v10, v01, v11 = v00 + ny, v00 + 1, v00 + ny + 1
    d = rs.rand(v00.size) < 0.5
    t1 = np.where(d[:, None], np.stack([v00, v10, v11], axis=1), np.stack([v00, v10, v01], axis=1))
    t2 = np.where(d[:, None], np.stack([v00, v11, v01], axis=1), np.stack([v10, v11, v01], axis=1))
    tris = np.concatenate([t1, t2], axis=0).astype(np.uint32)
    return verts, tris


def mesh_shuffle(verts, tris, seed=9):
    """The same surface handed over the way a mesh file from a survey tool may hold it: vertices renumbered by a
    random permutation, triangles in random order, each triangle's corners rotated at random and every second
    one's winding reversed.  Geometry unchanged."""
    rs = np.random.RandomState(seed)
    nv, nt = verts.shape[0], tris.shape[0]
    pv = rs.permutation(nv)                 # new position of old vertex i: where[i]
    where = np.empty(nv, dtype=np.int64)
    where[pv] = np.arange(nv)
    v2 = np.ascontiguousarray(verts[pv])
    t2 = where[tris.astype(np.int64)]
    t2 = t2[rs.permutation(nt)]
    rot = rs.randint(0, 3, nt)
    idx = (np.arange(3)[None, :] + rot[:, None]) % 3
    t2 = np.take_along_axis(t2, idx, axis=1)
    flip = rs.rand(nt) < 0.5
    t2[flip] = t2[flip][:, ::-1]
    return v2, np.ascontiguousarray(t2.astype(np.uint32))


def mesh_ragged(verts, tris, seed=13, band=3.0, p_gone=0.45, bays=6, bay_width=(2.0, 6.0), bay_depth=(10.0, 40.0), keep=None):
    """The same surface with the OUTLINE of a real survey: the triangles within `band` metres of the bounding box are removed
    at random (a sawtooth border), `bays` rectangular bays are cut in from the sides, and of what is left the largest
    edge-connected piece is kept (the scraps of a ragged border would lie outside the outline).  keep = (x, y): a bay is
    re-drawn until it stays a bay's width clear of that point.  Returns the triangle array (the vertices stay)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rs = np.random.RandomState(seed)
    t = tris.astype(np.int64)
    c = verts[t].mean(axis=1)
    x0, y0, x1, y1 = verts[:, 0].min(), verts[:, 1].min(), verts[:, 0].max(), verts[:, 1].max()
    edge = np.minimum(np.minimum(c[:, 0] - x0, x1 - c[:, 0]), np.minimum(c[:, 1] - y0, y1 - c[:, 1]))
    gone = (edge < band) & (rs.rand(len(t)) < p_gone)
    for _ in range(bays):
        for _try in range(20):
            w = rs.uniform(*bay_width)
            dep = rs.uniform(*bay_depth)
            side = rs.randint(4)
            if side < 2:
                y = rs.uniform(y0 + 0.1 * (y1 - y0), y1 - 0.1 * (y1 - y0))
                box = (x0 - 1, x0 + dep, y - w / 2, y + w / 2) if side == 0 else (x1 - dep, x1 + 1, y - w / 2, y + w / 2)
            else:
                x = rs.uniform(x0 + 0.1 * (x1 - x0), x1 - 0.1 * (x1 - x0))
                box = (x - w / 2, x + w / 2, y0 - 1, y0 + dep) if side == 2 else (x - w / 2, x + w / 2, y1 - dep, y1 + 1)
            if keep is None or not (box[0] - w < keep[0] < box[1] + w and box[2] - w < keep[1] < box[3] + w):
                break
        gone |= (c[:, 0] > box[0]) & (c[:, 0] < box[1]) & (c[:, 1] > box[2]) & (c[:, 1] < box[3])
    t = t[~gone]
    # triangles that share an edge: sort the 3 n undirected edges, equal neighbours in the order are pairs
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    e.sort(axis=1)
    key = e[:, 0] * (verts.shape[0] + 1) + e[:, 1]
    order = np.argsort(key, kind='stable')
    same = key[order][1:] == key[order][:-1]
    a, b = order[:-1][same] % len(t), order[1:][same] % len(t)
    ncomp, lab = connected_components(coo_matrix((np.ones(a.size), (a, b)), shape=(len(t), len(t))), directed=False)
    big = np.argmax(np.bincount(lab))
    return np.ascontiguousarray(t[lab == big].astype(np.uint32))


def landmark_map(n=4096, extent=(-64.0, -256.0, 448.0, 256.0), z_range=(-24.0, -16.0), seed=6):
    """Feature map of BASELINE config 5: n landmarks uniform over the map (seed 6)."""
    rs = np.random.RandomState(seed)
    x = extent[0] + (extent[2] - extent[0]) * rs.rand(n)
    y = extent[1] + (extent[3] - extent[1]) * rs.rand(n)
    zz = z_range[0] + (z_range[1] - z_range[0]) * rs.rand(n)
    return np.stack([x, y, zz], axis=1)


def beam_angles(n_beams, half_swath=math.pi / 3):
    """LaserScan-style fan: angle_min=-half_swath, equal increments, inclusive of +half_swath
    (mbes_processors/mbes_toy_processor/src/toy_mbes_manipulator.cpp:69-73 geometry)."""
    if n_beams == 1:
        return np.zeros(1, dtype=np.float32)
    return np.linspace(-half_swath, half_swath, n_beams).astype(np.float32)


# ---------------------------------------------------------------- raw sensor events for the DR integrator
EV_IMU, EV_HEADING, EV_GPS, EV_DVL, EV_DEPTH, EV_THRUST, EV_THRUST_CMD, EV_TICK = range(8)


def raw_sensor_events(duration=40.0, seed=11, scenario='auv', t0=50.0):
    """Time-ordered raw sensor events for the dead-reckoning integrator (the callbacks of
    sam_dead_reckoning/scripts/dr_node.py: stim_cb, sbg_cb, gps_cb, dvl_cb, depth_cb, thrust_cb,
    thrust_cmd_cb, dr_timer).  Returns (t[n], kind[n] int32, data[n, 7]) with data columns:
      EV_IMU: qx qy qz qw wx wy wz     EV_HEADING: qx qy qz qw     EV_GPS: x y (map frame)
      EV_DVL: vx vy vz                 EV_DEPTH: z                 EV_THRUST: rpm1 rpm2
      EV_THRUST_CMD: horizontal_radians                            EV_TICK: (none)
    scenario 'auv': pressure sensor present, DVL from t0+1 s with outliers that trip the plausibility
    gates and a 3 s dropout; 'surface': heading arrives after the first GPS fix, DVL starts late."""
    rs = np.random.RandomState(seed)
    ev = []

    def add(t, kind, vals=()):
        row = np.zeros(7)
        row[:len(vals)] = vals
        ev.append((t, kind, row))

    surface = scenario == 'surface'
    # ticks at 50 Hz, IMU at 100 Hz (offset so stamps never tie with ticks)
    for k in range(int(duration / 0.02)):
        add(t0 + 0.02 * k + 0.0101, EV_TICK)
    for k in range(int(duration / 0.01)):
        t = t0 + 0.01 * k + 0.0033
        tt = t - t0
        roll, pitch = 0.03 * math.sin(0.7 * tt), 0.05 * math.cos(0.4 * tt)
        yaw = 0.3 + 0.2 * math.sin(0.15 * tt)
        q = quat_from_rpy(roll, pitch, yaw)
        w = (0.021 * math.cos(0.7 * tt), -0.02 * math.sin(0.4 * tt),
             0.03 * math.cos(0.15 * tt) + 0.002 * rs.randn())
        add(t, EV_IMU, (q[0], q[1], q[2], q[3], w[0], w[1], w[2]))
    add(t0 + (2.5 if surface else 0.0507), EV_HEADING, tuple(quat_from_rpy(0.01, -0.02, 0.9)))
    for k in range(5):
        add(t0 + 0.5017 + 1.0 * k, EV_GPS, (12.5 + 0.1 * k, -7.25 - 0.1 * k))
    dvl_start = 6.0 if surface else 1.0
    k = 0
    t = t0 + dvl_start + 0.0049
    while t < t0 + duration:
        tt = t - t0
        if not (20.0 <= tt < 23.0):  # dropout: the motion model takes over
            v = [1.0 + 0.05 * math.sin(0.1 * tt) + 0.01 * rs.randn(), 0.02 * math.sin(0.3 * tt) + 0.005 * rs.randn(),
                 0.01 * rs.randn()]
            if k % 17 == 5:
                v[1] = 0.25      # |vy| gate
            if k % 23 == 7:
                v[0] = 1.7       # |vx| gate
            if k % 29 == 11:
                v[0] = -0.2      # reverse gate
            add(t, EV_DVL, v)
        k += 1
        t += 0.2 + (0.013 if k % 7 == 0 else 0.0)  # an occasional late message trips the age gate
    for k in range(int(duration / 0.1)):
        tt = 0.1 * k + 0.0071
        add(t0 + tt, EV_DEPTH, (-2.0 - 0.5 * math.sin(0.05 * tt) + 0.01 * rs.randn(),))
        add(t0 + tt + 0.0013, EV_THRUST, (float(int(400 + 50 * math.sin(0.2 * tt))), float(int(380 + 40 * math.cos(0.1 * tt)))))
    for k in range(int(duration / 0.5)):
        tt = 0.5 * k + 0.0191
        add(t0 + tt, EV_THRUST_CMD, (0.2 * math.sin(0.3 * tt),))  # beyond the 7 degree clip at the peaks
    ev.sort(key=lambda e: (e[0], e[1]))
    return (np.array([e[0] for e in ev]), np.array([e[1] for e in ev], dtype=np.int32),
            np.stack([e[2] for e in ev]))
