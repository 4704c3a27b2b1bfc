/* mcl_map.h -- C ABI of the bathymetry map builder: multibeam pings taken at known poses are turned
 * into the swath point cloud and rasterised into the height grid that mcl_set_map_grid consumes
 * (second half).
 *
 * Reference analogue: mbes_processors/mbes_mapper/src/mbes_receptor.cpp -- MBESLaserCB :126-165 turns a
 * LaserScan into points in the base frame and stores them with the map->base pose, pclFuser :64-107
 * transforms every ping of a swath into one frame and concatenates them.  The build does that
 * transform on the GPU (one thread per beam; the concatenated cloud is the optional points_out) and
 * adds what the ray-cast needs: a regular grid of mean depths.  The rasterisation has no reference
 * counterpart (parity unpinned); it is defined so that it is bit-reproducible:
 *   node (i, j) = nearest grid node of the point's (x, y);  the node accumulates round(z * 2^20) in a
 *   64-bit integer and a count (integer atomics: the order of the pings does not matter);
 *   z(i, j) = sum / 2^20 / count;  empty nodes are NaN, optionally filled by `fill_passes` Jacobi
 *   sweeps (an empty node takes the mean of its non-empty 8-neighbours of the previous sweep).
 * Beam geometry, sensor pose and validity rules are those of mcl_update_mbes (mcl.h): beam b looks along
 * (0, sin a_b, -cos a_b) in the sensor frame, sensor pose = m2o * T(xyz) R(rpy) * T_off R_off, ranges
 * <= 0, NaN or >= r_max are skipped.  Conventions as in mcl.h (int status, one owner thread). */
#ifndef MCL_MAP_H
#define MCL_MAP_H
#include "mcl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcl_gridmap mcl_gridmap;

/* nx x ny nodes, node (i, j) at (ox + i res, oy + j res) -- the layout of mcl_set_map_grid */
int mcl_gridmap_create(int32_t nx, int32_t ny, double ox, double oy, double res, int32_t device, mcl_gridmap** out);
void mcl_gridmap_destroy(mcl_gridmap* g);
const char* mcl_gridmap_last_error(const mcl_gridmap* g);
int mcl_gridmap_clear(mcl_gridmap* g);

/* poses6: n_pings x (x, y, z, roll, pitch, yaw) in the odom frame (what the filter estimates);
 * ranges: n_pings x n_beams.  m2o (row-major 4x4) and sensor_offset may be NULL (identity / zero).
 * points_out (optional, host): n_pings x n_beams x 3 doubles in the map frame, NaN for skipped beams. */
int mcl_gridmap_add_pings(mcl_gridmap* g, const double* poses6, int64_t n_pings, const float* ranges,
                          const float* beam_angles, int32_t n_beams, double r_max, const double m2o[16],
                          const double sensor_offset[6], double* points_out);

/* z_out: nx * ny floats (z[i * ny + j]); n_empty (optional): nodes still NaN after filling;
 * counts_out (optional): nx * ny hit counts. */
int mcl_gridmap_finalize(mcl_gridmap* g, int32_t fill_passes, float* z_out, int64_t* n_empty, uint32_t* counts_out);

#ifdef __cplusplus
}
#endif
#endif
