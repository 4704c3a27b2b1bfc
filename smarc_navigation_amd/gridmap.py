"""Bathymetry map builder (include/mcl_map.h): pings at known poses -> swath point cloud -> the
height grid `Engine.set_map_grid` consumes.  GPU only (no fallback)."""
import ctypes as C

import numpy as np

from . import _lib


class GridMapBuilder(object):
    def __init__(self, nx, ny, origin, res, device=0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.nx, self.ny, self.origin, self.res = int(nx), int(ny), (float(origin[0]), float(origin[1])), float(res)
        rc = self._L.mcl_gridmap_create(self.nx, self.ny, self.origin[0], self.origin[1], self.res, int(device),
                                        C.byref(self._h))
        if rc != 0:
            raise _lib.MclError(rc, (self._L.mcl_gridmap_last_error(None) or b'').decode())

    def _ck(self, rc):
        if rc != 0:
            raise _lib.MclError(rc, (self._L.mcl_gridmap_last_error(self._h) or b'').decode())

    def close(self):
        if self._h:
            self._L.mcl_gridmap_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def clear(self):
        self._ck(self._L.mcl_gridmap_clear(self._h))

    def add_pings(self, poses6, ranges, beam_angles, r_max, m2o=None, sensor_offset=None, want_points=False):
        """poses6[n, 6] (x, y, z, roll, pitch, yaw), ranges[n, B]; returns the point cloud [n, B, 3] (map frame,
        NaN for skipped beams) when want_points."""
        poses6 = np.ascontiguousarray(poses6, dtype=np.float64).reshape(-1, 6)
        ba = np.ascontiguousarray(beam_angles, dtype=np.float32)
        ranges = np.ascontiguousarray(ranges, dtype=np.float32).reshape(poses6.shape[0], ba.size)
        m = None if m2o is None else np.ascontiguousarray(m2o, dtype=np.float64).reshape(16)
        so = None if sensor_offset is None else np.ascontiguousarray(sensor_offset, dtype=np.float64).reshape(6)
        pts = np.zeros((poses6.shape[0], ba.size, 3)) if want_points else None
        self._ck(self._L.mcl_gridmap_add_pings(self._h, poses6.ctypes.data, poses6.shape[0], ranges.ctypes.data,
                                               ba.ctypes.data, ba.size, float(r_max),
                                               m.ctypes.data if m is not None else None,
                                               so.ctypes.data if so is not None else None,
                                               pts.ctypes.data if pts is not None else None))
        return pts

    def finalize(self, fill_passes=0, want_counts=False):
        """Returns (z[nx, ny] float32 with NaN where empty, n_empty[, counts])."""
        z = np.zeros((self.nx, self.ny), dtype=np.float32)
        cnt = np.zeros((self.nx, self.ny), dtype=np.uint32) if want_counts else None
        ne = C.c_int64(0)
        self._ck(self._L.mcl_gridmap_finalize(self._h, int(fill_passes), z.ctypes.data, C.byref(ne),
                                              cnt.ctypes.data if cnt is not None else None))
        return (z, int(ne.value), cnt) if want_counts else (z, int(ne.value))


# ---------------------------------------------------------------------------------------------------------
# Submap windowing of mbes_processors/mbes_mapper (mbes_receptor.cpp): every `meas_size` pings form one
# swath; pclFuser (:64-107) takes the map -> base pose of the MIDDLE ping ((meas_size - 1) / 2) as the
# submap frame, moves every ping's points (stored in the base frame of their own time, MBESLaserCB :126-165)
# into it with  T_submap<-map * (T_base_t<-map)^-1,  concatenates them, stores the submap's pose in the map as
# the cloud's sensor origin / orientation, publishes it as "submap_<k>_frame" and dumps it as ASCII PCD
# ("./submap_<k>_frame.pdc" -- the node's own spelling of the extension).
def _pose_matrix(p6):
    from . import synth
    return synth.rigid_matrix(*[float(v) for v in p6])


def _quat_from_matrix(R):
    """rotation matrix -> (x, y, z, w), w >= 0"""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q if q[3] >= 0 else -q


def fuse_swath(points_base, poses_map_base, index=0):
    """pclFuser on the host: points_base[k] = (m_k x 3) points of ping k in ITS base frame, poses_map_base[k] = the
    vehicle pose (x, y, z, roll, pitch, yaw) in the map at that ping.  Returns the submap dict."""
    n = len(points_base)
    mid = (n - 1) // 2
    T_map_sub = _pose_matrix(poses_map_base[mid])      # map <- base at the middle ping = pose of the submap
    T_sub_map = np.linalg.inv(T_map_sub)
    out = []
    for pts, p6 in zip(points_base, poses_map_base):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
        T = T_sub_map.dot(_pose_matrix(p6))            # submap <- map <- base_t
        out.append(pts.dot(T[:3, :3].T) + T[:3, 3])
    cloud = np.concatenate(out, axis=0) if out else np.zeros((0, 3))
    cloud = cloud[~np.isnan(cloud).any(axis=1)]
    return dict(index=int(index), frame_id='submap_%d_frame' % index, points=cloud.astype(np.float32),
                origin=T_map_sub[:3, 3].copy(), quat=_quat_from_matrix(T_map_sub[:3, :3]), T_map_submap=T_map_sub)


def save_pcd_ascii(path, submap):
    """what pcl::io::savePCDFileASCII writes for an xyz cloud with a sensor pose (VIEWPOINT tx ty tz qw qx qy qz)"""
    pts = np.asarray(submap['points'], dtype=np.float32)
    o, q = submap['origin'], submap['quat']
    with open(path, 'w') as f:
        f.write('# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n'
                'COUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT %.9g %.9g %.9g %.9g %.9g %.9g %.9g\nPOINTS %d\nDATA ascii\n'
                % (pts.shape[0], o[0], o[1], o[2], q[3], q[0], q[1], q[2], pts.shape[0]))
        for p in pts:
            f.write('%.9g %.9g %.9g\n' % (p[0], p[1], p[2]))


def load_pcd_ascii(path):
    with open(path) as f:
        lines = f.read().splitlines()
    k = lines.index('DATA ascii')
    hdr = {l.split(' ', 1)[0]: l.split(' ', 1)[1] for l in lines[:k] if not l.startswith('#')}
    vp = [float(v) for v in hdr['VIEWPOINT'].split()]
    pts = np.array([[float(v) for v in l.split()] for l in lines[k + 1:] if l.strip()], dtype=np.float32).reshape(-1, 3)
    return dict(points=pts, origin=np.array(vp[:3]), quat=np.array([vp[4], vp[5], vp[6], vp[3]]), n=int(hdr['POINTS']))


class SubmapBuilder(object):
    """The N-ping windowing of MBESReceptor (meas_size pings per swath) over LaserScan-style pings: ranges +
    beam angles + the vehicle pose in the map.  The points of a swath are produced on the GPU directly in the
    submap frame (mcl_gridmap_add_pings with map<-odom replaced by submap<-map: one thread per beam)."""

    def __init__(self, meas_size, beam_angles, r_max, sensor_offset=None, device=0):
        self.meas_size = int(meas_size)
        self.beam_angles = np.ascontiguousarray(beam_angles, dtype=np.float32)
        self.r_max, self.sensor_offset = float(r_max), sensor_offset
        self._g = GridMapBuilder(2, 2, (0.0, 0.0), 1.0, device=device)  # only its point output is used
        self._poses, self._ranges = [], []
        self.submaps = []

    def add_ping(self, pose6_map_base, ranges):
        """returns the fused submap when this ping completes a swath (mbes_receptor.cpp:160-163), else None"""
        self._poses.append(np.asarray(pose6_map_base, dtype=np.float64))
        self._ranges.append(np.asarray(ranges, dtype=np.float32))
        if len(self._poses) < self.meas_size:
            return None
        poses = np.stack(self._poses)
        T_map_sub = _pose_matrix(poses[(self.meas_size - 1) // 2])
        pts = self._g.add_pings(poses, np.stack(self._ranges), self.beam_angles, self.r_max,
                                m2o=np.linalg.inv(T_map_sub), sensor_offset=self.sensor_offset, want_points=True)
        self._g.clear()
        cloud = pts.reshape(-1, 3)
        cloud = cloud[~np.isnan(cloud).any(axis=1)]
        k = len(self.submaps)
        sm = dict(index=k, frame_id='submap_%d_frame' % k, points=cloud.astype(np.float32), origin=T_map_sub[:3, 3].copy(),
                  quat=_quat_from_matrix(T_map_sub[:3, :3]), T_map_submap=T_map_sub)
        self.submaps.append(sm)
        self._poses, self._ranges = [], []
        return sm

    def save(self, directory='.'):
        import os
        paths = []
        for sm in self.submaps:
            paths.append(os.path.join(directory, 'submap_%d_frame.pdc' % sm['index']))
            save_pcd_ascii(paths[-1], sm)
        return paths
