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
