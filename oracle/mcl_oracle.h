/* mcl_oracle.h -- CPU restatement (plain C, IEEE fp64) of the auv_particle_filter hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (smarc_navigation_amd/, include/)
 * never links, imports or calls it.
 *
 * Parity status:
 *   - predict / add_noise / GPS weight / normalise / all five resamplers / keep-lost-dupes
 *     reassign / mean+cov: PINNED against golden vectors produced by importing the reference's
 *     own Python (tests/golden/*.npz, generator oracle/ref_harness/gen_golden.py).
 *   - MBES grid / mesh ray-cast + beam log-likelihood: PARITY UNPINNED.  The reference's
 *     auv_particle_filter has no MBES model; these functions are the build's own
 *     fp64 definition ("self-oracle"), checked only against analytic cases.
 *
 * State layout everywhere: SoA, state[c*n + i], c = 0..5 = x, y, z, roll, pitch, yaw (odom frame).
 * All citations are relative to /root/reference/auv_particle_filter/scripts/.
 */
#ifndef MCL_ORACLE_H
#define MCL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry helpers (tf.transformations 'sxyz', published algorithm; auv_particle.py:8-10) */
void orc_euler_from_quat(const double q[4], double rpy[3]);
void orc_quat_from_euler(double roll, double pitch, double yaw, double q[4]);
void orc_matrix_from_tf(const double t[3], const double q[4], double M[16]); /* auv_particle.py:110-125 */
double orc_wrap_pi(double a);                                                 /* auv_particle.py:48 */

/* ---- a2: Particle.add_noise (auv_particle.py:32-36). normals: n x 6, particle-major */
int orc_set_threads(int nthreads);
void orc_add_noise(int n, double* state, const double cov[6], const double* normals);
/* ---- a4/a5: Particle.motion_pred + fullRotation (auv_particle.py:38-97). normals n x 6 or NULL */
void orc_predict(int n, double* state, const double v[3], double wz, const double q[4], double z,
                 double dt, const double pcov[6], const double* normals);
/* ---- a7: get_p_pose + compute_weight (auv_particle.py:72-106), closed form of the 2-D pdf.
 * w_raw (may be NULL) = pdf (no +1e-200); lw (may be NULL) = log pdf */
void orc_gps_weights(int n, const double* state, const double m2o[16], double gx, double gy,
                     double sigma, double* w_raw, double* lw);
/* ---- a6/a8: weights += 1e-200 (auv_pf.py:165); weights /= weights.sum() (auv_pf.py:172) */
double orc_numpy_pairwise_sum(const double* a, int64_t n, int64_t stride);
void orc_normalise_ref(int n, double* w);

/* ---- a9-a11: resampling.py, reference-exact fp64.  Return 0, or -1 if the reference would have
 * raised IndexError (index clamped to n-1 in that case).  uniforms = draws in reference order. */
int orc_systematic_ref(int n, const double* w, double u, int32_t* idx);          /* :135-168 */
int orc_stratified_ref(int n, const double* w, const double* u, int32_t* idx);   /* :80-114  */
int orc_multinomial_ref(int n, const double* w, const double* u, int32_t* idx);  /* :171-194 */
int orc_naive_ref(int n, const double* w, double u01, int32_t* idx);             /* :116-131 */
/* residual: returns k = number of deterministic copies (uniforms consumed = n-k), or -1 */
int orc_residual_ref(int n, const double* w, const double* u, int32_t* idx);     /* :27-76   */
int orc_residual_k(int n, const double* w);

/* ---- a12: keep/lost/dupes + reassign (auv_pf.py:183-198).  Works for ANY index vector
 * (sorted or not).  lost/dupes sized n; returns count. */
int orc_lost_dupes(int n, const int32_t* idx, int32_t* lost, int32_t* dupes);
void orc_reassign(int n, double* state, int n_lost, const int32_t* lost, const int32_t* dupes);

/* ---- a13: update_loc_pose (auv_pf.py:218-260): mean6 (sequential row accumulation), yaw =
 * pairwise mean of wrapped yaws, cov written like pose.covariance[i*3+j] (first 9 entries) */
void orc_mean_cov(int n, const double* state, double mean6[6], double* yaw_mean, double cov9[9]);

/* ==== Fixed-point weight / systematic-resample SPEC (what the HIP path must match bit-exactly;
 * DESIGN.md "Resampling arithmetic").  Deterministic, order-free (integer sums). */
double orc_det_exp(double x);
/* mode 0 (GPS/reference): w = det_exp(lw) + 1e-200, q_i = floor(w_i / mw * 2^s), mw = weight of the max-lw particle,
 * s = 63 - ceil_log2(n_global);  mode 1 (MBES/log domain): q_i = floor(exp(lw_i) 2^(s - K)), K = the integer exponent of
 * the largest log-likelihood (orc_weight_exponent) -- a shard that only knows its own maximum quantises at its own
 * exponent and shifts (orc_quantise_log_weight; mcl_device.h: quantise_log_weight).
 * Multi-shard: pass m_lw_global via lw of all shards (the caller concatenates).  Returns T = sum q. */
uint64_t orc_fixed_weights(int n, const double* lw, int mode, int64_t n_global, uint64_t* q,
                           double* w_lin);
double orc_max(int n, const double* v);
int64_t orc_weight_exponent(double m_lw);
uint64_t orc_quantise_log_weight(double lw, int64_t K, int s);
uint64_t orc_fixed_weights_m(int n, const double* lw, int mode, int64_t n_global, double m_lw_global,
                             uint64_t* q, double* w_lin);
/* ncum[j] = #{ i in [0,N) : (U53 + i*2^53) * T < C_j * N * 2^53 },  C = inclusive scan of q
 * (+ c_offset), N = n_global, T = total.  ncum is this shard's slice. */
void orc_systematic_ncum(int n, const uint64_t* q, uint64_t c_offset, uint64_t total, int64_t n_global,
                         uint64_t u53, uint32_t* ncum);
/* idx[i] = min{ j : ncum[j] > i } for i in [i0, i0+cnt) over the GLOBAL ncum[0..N) */
void orc_indices_from_ncum(int64_t n_global, const uint32_t* ncum, int64_t i0, int64_t cnt, int32_t* idx);

/* ---- Philox4x32-10 + Box-Muller (native RNG mode; DESIGN.md "RNG") */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]);
/* purpose: 0 init, 1 predict, 2 resample noise; writes n x 6 normals (unused slots = 0) */
void orc_native_normals(int n, int64_t gid0, uint64_t seed, uint32_t purpose, uint32_t step, double* normals);
uint64_t orc_native_u53(uint64_t seed, uint32_t step);

/* ==== MBES (SELF-ORACLE, parity unpinned) */
typedef struct {
  int nx, ny;
  double ox, oy, res;
  const float* z; /* z[ix*ny + iy] */
} orc_grid;
/* expected range of one ray vs bilinear height field; r_max if no hit */
double orc_ray_grid(const orc_grid* g, const double o[3], const double d[3], double r_max);
/* brute-force closest hit ray vs triangle soup (Moller-Trumbore, two-sided) */
double orc_ray_mesh_brute(const float* verts, const uint32_t* tris, int64_t nt, const double o[3],
                          const double d[3], double r_max);
/* accelerated mesh: opaque uniform-grid binning built once */
typedef struct orc_mesh orc_mesh;
orc_mesh* orc_mesh_build(const float* verts, int64_t nv, const uint32_t* tris, int64_t nt);
void orc_mesh_free(orc_mesh* m);
double orc_ray_mesh(const orc_mesh* m, const double o[3], const double d[3], double r_max);
/* per-particle sensor rays -> expected ranges (n x B) and log-weights (n).  map_kind 0 grid, 1 mesh.
 * ranges: measured, B floats (<=0 or NaN = invalid beam).  exp_out may be NULL. */
void orc_mbes_update(int n, const double* state, const double m2o[16], const double sensor_off[6],
                     int map_kind, const void* map, const float* beam_angles, const float* ranges, int B,
                     double sigma, double r_max, double* lw, double* exp_out);
/* landmark update with k-NN association (SELF-ORACLE, brute force over all landmarks) */
void orc_landmark_update(int n, const double* state, const double m2o[16], const double sensor_off[6],
                         const double* lm, int64_t n_lm, const double* det, int n_det, double sigma, int k,
                         double gate, double* lw);
/* dense rectangular assignment (n rows assigned, m >= n columns); pinned to the reference Munkres via oracle/_ref */
double orc_assign_dense(int n, int m, const double* cost, int* col_of_row);
/* landmark update with a global assignment per particle (table as auv_ekf_slam/src/ekf_slam_core.cpp:172-312) */
void orc_landmark_assign_update(int n, const double* state, const double m2o[16], const double sensor_off[6],
                                const double* lm, int64_t n_lm, const double* det, int n_det, double sigma,
                                int k_cand, double gate, double new_mh_dist, double* lw, int* assign_out);
/* deterministic log / sin-cos(2 pi u) of the NATIVE Box-Muller draws (== mcl_device.h det_log / det_sincos2pi) */
double orc_det_log(double x);
void orc_det_sincos2pi(double u, double* sn, double* cs);
/* Mahalanobis association the reference's way (sensor-frame innovation, S = R^T Sigma_j R + Q; ekf_slam_core.cpp:135-178) */
void orc_landmark_update_maha(int n, const double* state, const double m2o[16], const double sensor_off[6], const double* lm,
                              const double* lmcov, int64_t n_lm, const double* det, int n_det, double sigma, const double* Q6,
                              int k, double gate, double* lw);
void orc_landmark_assign_update_maha(int n, const double* state, const double m2o[16], const double sensor_off[6],
                                     const double* lm, const double* lmcov, int64_t n_lm, const double* det, int n_det,
                                     double sigma, const double* Q6, int k_cand, double gate, double new_mh_dist, double* lw,
                                     int* assign_out, double* table_out);
/* bathymetry map builder (self-oracle of include/mcl_map.h) */
void orc_gridmap_add_pings(int nx, int ny, double ox, double oy, double res, int64_t* sum, uint32_t* cnt,
                           int64_t n_pings, const double* poses6, const float* ranges, const float* beam_angles,
                           int B, double r_max, const double m2o[16], const double sensor_off[6], double* points_out);
int64_t orc_gridmap_finalize(int nx, int ny, const int64_t* sum, const uint32_t* cnt, int fill_passes, float* z_out);
#ifdef __cplusplus
}
#endif
#endif
