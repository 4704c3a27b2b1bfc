/* mcl.h -- C ABI of the MI355X-native Monte-Carlo localization engine (libmcl_hip.so).
 *
 * Drop-in boundary for the hot path of smarc_navigation's auv_particle_filter node.  The
 * reference has no FFI of its own (its hot path is Python inside the rospy node); this ABI sits
 * at the seam between the node's ROS plumbing (auv_pf.py) and its numerics (auv_particle.py,
 * resampling.py).  Each entry point cites the reference code it replaces, relative to
 * /root/reference/auv_particle_filter/scripts/.
 *
 * Conventions: extern "C"; opaque handle; every call returns an int status (MCL_OK == 0,
 * negative = error, mcl_last_error() gives the text); no exceptions cross the boundary; caller
 * owns every host buffer, the library owns every device buffer; one handle is used by one
 * thread at a time (the reference's three unlocked rospy threads are specified away,
 * SURVEY.md A.10).  All particle math is IEEE fp64 like the reference; state is SoA in HBM:
 * six arrays x, y, z, roll, pitch, yaw of n doubles (odom frame).  There is NO CPU fallback:
 * without a gfx950 device mcl_create() fails with MCL_ERR_NO_DEVICE.
 */
#ifndef MCL_H
#define MCL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCL_ABI_VERSION 4   /* 2: mcl_timing gained MCL_K_MBES_MAIN; 3: the exchange's phases (MCL_K_COMM_RECORDS ... MCL_K_COMM_MOMENTS);
                               4: MCL_K_UPDATE_LANDMARKS, mcl_step_mbes_landmarks */

typedef struct mcl_handle mcl_handle;

enum mcl_status {
  MCL_OK = 0,
  MCL_ERR_INVALID = -1,     /* bad argument */
  MCL_ERR_NO_DEVICE = -2,   /* no HIP device / not gfx950 */
  MCL_ERR_HIP = -3,         /* a HIP runtime call failed */
  MCL_ERR_UNSUPPORTED = -4, /* valid request this build does not implement */
  MCL_ERR_STATE = -5,       /* call order (e.g. update_mbes before set_map_*) */
  MCL_ERR_COMM = -6,        /* RCCL failure */
  MCL_ERR_ALLOC = -7
};

enum mcl_resample_scheme {
  MCL_RESAMPLE_SYSTEMATIC = 0, /* resampling.py:135-168 -- the GPU production scheme */
  MCL_RESAMPLE_RESIDUAL = 1,   /* resampling.py:27-76   -- what auv_pf.py:182 calls */
  MCL_RESAMPLE_STRATIFIED = 2, /* resampling.py:80-114 */
  MCL_RESAMPLE_MULTINOMIAL = 3,/* resampling.py:171-194 */
  MCL_RESAMPLE_NAIVE = 4       /* resampling.py:116-131 -- systematic positions, `>` instead of `<`; sharded like it */
};

enum mcl_rng_mode {
  MCL_RNG_NATIVE = 0, /* Philox4x32-10 keyed by (seed, step, GLOBAL particle id): grid/GPU-count invariant */
  MCL_RNG_REPLAY = 1  /* caller passes the normals/uniforms (parity with numpy's MT19937 stream) */
};

enum mcl_comm_mode {
  MCL_COMM_NONE = 0,  /* single shard */
  MCL_COMM_RCCL = 1,  /* one process per GPU, RCCL over xGMI (mcl_comm_init) */
  MCL_COMM_LOCAL = 2  /* several shards in ONE process (same or different GPUs); test harness */
};

enum mcl_weight_mode {
  MCL_WEIGHT_LINEAR_FLOOR = 0, /* w = exp(lw) + 1e-200   (auv_pf.py:165, GPS update) */
  MCL_WEIGHT_LOG_SHIFT = 1,    /* w = exp(lw) 2^-K, K = rint(max lw x log2 e) + 1: relative to the power of two next above the largest
                                * weight -- in (2^-3/2, 2^-1/2] of it -- so that a shard can form its weights before the cloud's maximum is
                                * known (MBES update, log domain; rounds 1-5: exp(lw - max lw)) */
  MCL_WEIGHT_LINEAR = 2        /* the values ARE linear weights (mcl_resample_indices) */
};

typedef struct mcl_config {
  int64_t n_particles;   /* particles owned by THIS shard (auv_pf.py:27 `particle_count`) */
  int64_t n_global;      /* total over all shards; 0 -> n_particles */
  int64_t global_offset; /* global id of this shard's particle 0 */
  int32_t device;        /* HIP device ordinal */
  int32_t rank, world;   /* shard index / count (world 0 or 1 -> single shard) */
  int32_t resample_scheme;
  int32_t rng_mode;
  int32_t comm_mode;
  uint64_t seed;
  double init_cov[6];     /* auv_pf.py:46-50  `init_covariance`             order x,y,z,roll,pitch,yaw */
  double process_cov[6];  /* auv_pf.py:40-44  `motion_covariance` */
  double resample_cov[6]; /* auv_pf.py:52-56  `resampling_noise_covariance` */
  double meas_std;        /* auv_pf.py:39     `measurement_std` (GPS) */
  double m2o[16];         /* row-major 4x4 map<-odom, matrix_from_tf (auv_particle.py:110-125) */
} mcl_config;

/* The fields of nav_msgs/Odometry the filter consumes (auv_particle.py:45-70). */
typedef struct mcl_odom {
  double stamp; /* header.stamp.to_sec() */
  double v[3];  /* twist.twist.linear, body frame */
  double w_z;   /* twist.twist.angular.z */
  double q[4];  /* pose.pose.orientation x,y,z,w */
  double z;     /* pose.pose.position.z */
} mcl_odom;

/* Per-kernel device time accumulated with HIP events on the handle's stream while timing is on. */
enum mcl_kernel_id {
  MCL_K_PREDICT = 0,
  MCL_K_UPDATE_GPS = 1,
  MCL_K_UPDATE_MBES = 2,
  MCL_K_NORMALISE = 3, /* max-reduce + exp + fixed-point quantise */
  MCL_K_SCAN = 4,      /* u64 prefix scan + offspring counts */
  MCL_K_RESAMPLE = 5,  /* lost-slot scan + reassign gather + noise */
  MCL_K_MEAN_COV = 6,
  MCL_K_NOISE = 7,
  MCL_K_COMM = 8,
  MCL_K_MBES_MAIN = 9, /* the ONE dominant launch of an MBES update (first sweep pass, or the fast traversal kernel):
                          nested inside MCL_K_UPDATE_MBES, whose region also holds the memset, the pose kernel and the
                          (usually empty) hand-over launches */
  /* the O(n)-per-rank resample exchange by phase; MCL_K_COMM keeps the all-gather scheme's transfers */
  MCL_K_COMM_RECORDS = 10, /* all-reduce of the max-lw slots, all-gathers of the shard totals and hand-over records */
  MCL_K_PACK = 11,         /* k_pack_dupes: surplus copies into per-copy records */
  MCL_K_COMM_P2P = 12,     /* the ONE group of ncclSend / ncclRecv (at most one each per peer) */
  MCL_K_COMM_MOMENTS = 13, /* all-reduce of the mean / covariance sums */
  MCL_K_UPDATE_LANDMARKS = 14, /* k_landmark_update (until ABI 3 counted under MCL_K_UPDATE_MBES) */
  MCL_K_COUNT = 15
};
typedef struct mcl_timing {
  double ms[MCL_K_COUNT];       /* summed device milliseconds */
  int64_t launches[MCL_K_COUNT]; /* number of timed regions */
} mcl_timing;

/* ---- library ------------------------------------------------------------------------- */
int mcl_abi_version(void);
const char* mcl_status_string(int status);
const char* mcl_last_error(const mcl_handle* h); /* h may be NULL: last create() error */
int mcl_device_count(int* count);
/* a14: matrix_from_tf (auv_particle.py:110-125): 4x4 row-major map<-odom from a tf translation
 * (x,y,z) and quaternion (x,y,z,w); host-side helper for filling mcl_config.m2o */
int mcl_matrix_from_tf(const double translation[3], const double quaternion[4], double m16[16]);

/* ---- lifetime: replaces auv_pf.__init__'s particle list (auv_pf.py:89-94) ---------------- */
int mcl_create(const mcl_config* cfg, mcl_handle** out);
int mcl_destroy(mcl_handle* h);
/* Particle.__init__: pose = 0 + add_noise(init_cov) (auv_particle.py:24,30).
 * replay_normals: n x 6 doubles, particle-major (REPLAY mode) or NULL (NATIVE). */
int mcl_init_particles(mcl_handle* h, const double* replay_normals);

/* ---- a3/a4/a5: auv_pf.predict + Particle.motion_pred (auv_pf.py:213-216, auv_particle.py:38-97)
 * dt <= 0 is a no-op like the reference's `time > old_time` gate (auv_pf.py:205). */
int mcl_predict(mcl_handle* h, const mcl_odom* odom, double dt, const double* replay_normals);

/* ---- a6/a7: auv_pf.update + Particle.compute_weight (auv_pf.py:135-167, auv_particle.py:100-106)
 * (gx, gy) = the GPS fix already transformed utm->map (the reference repeats that transform per
 * particle with identical result, auv_pf.py:151).  Leaves log-weights on the device. */
int mcl_update_gps(mcl_handle* h, double gx_map, double gy_map);

/* ---- a15: MBES measurement update (no reference symbol; north_star).  Map in the MAP frame. */
int mcl_set_map_grid(mcl_handle* h, const float* z, int32_t nx, int32_t ny, double origin_x,
                     double origin_y, double res); /* z[ix*ny + iy] at (ox + ix*res, oy + iy*res) */
int mcl_set_map_mesh(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt);
/* flags: MCL_MESH_HEIGHTFIELD = the caller declares the mesh single-valued in z over (x,y) (what a
 * bathymetric surface is).  It enables neighbour-chained ray starts; a mesh with
 * overhangs must NOT set it -- results would be wrong for occluded beams. */
#define MCL_MESH_HEIGHTFIELD 1u
/* MCL_MESH_GENERAL: cast this mesh with the general triangle-record traversal only -- no structured-mesh fast
 * path even when the mesh is detected to be a triangulated regular height grid, and no fan sweep by adjacency
 * (testing / A-B runs of the kernels that serve arbitrary triangle soups) */
#define MCL_MESH_GENERAL 2u
/* MCL_MESH_UNSTRUCTURED: no structured-mesh detection, but the fan sweep by triangle adjacency is still used when
 * mesh_build proves the mesh a single-valued height field (A-B runs of the adjacency walk on a regular mesh) */
#define MCL_MESH_UNSTRUCTURED 4u
int mcl_set_map_mesh_ex(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt,
                        uint32_t flags);
/* ranges[b] <= 0 or NaN marks an invalid beam; beam b looks along (0, sin a_b, -cos a_b) in the
 * sensor frame; sensor_offset = x,y,z,roll,pitch,yaw of the sensor in base_link (NULL = zeros).
 * Precision contract (state in fp64, ray-cast in fp32 -- SURVEY 8(d) allows it with a stated tolerance): against the
 * fp64 definition (oracle/mcl_oracle.c) an expected range is within 1e-3 m and a log-likelihood within
 *     |d lw| <= 1e-2   or   |d lw| <= 2e-4 |lw|
 * -- the relative arm is this build's addition to SURVEY's absolute 1e-2: the sum of 512 squared residuals of a particle
 * metres off the truth reaches |lw| ~ 2e4, where fp32 residuals alone are worth 1e-2 (measured: 1.2e-2).  WHERE IT BITES
 * the absolute bound holds on its own: every particle that can receive offspring -- log-likelihood within 30 of the
 * cloud's maximum, i.e. a non-zero fixed-point weight (q = floor(exp(lw) 2^(s - K)), s = 43 at 2^20 particles: exp(-29.8) =
 * 2^-43) -- is within |d lw| <= 1e-2 of the fp64 definition (measured at 1 M x 512: 1.1e-3 on the lattice mesh, the
 * height grid, the irregular TIN and the triangle soup; tests/helpers.py: live_particle_contract, asserted in
 * test_gpu_fullsize.py, test_gpu_config45.py and the driver's smoke()).  The relative arm only ever excuses particles whose
 * weight is zero.  Exceptions
 * are the rays that graze a crest or an edge: fp32's last bit decides between the crest and the shadow behind it.
 * Their NUMBER is bounded by the tests (a few per 10^4 rays on rough terrain) and so is their SIZE: each is, within
 * 1e-3 m, an answer the fp64 definition itself gives when the sensor moves by 1 mm (tests/helpers.py:
 * outliers_explained, lw_outliers_explained). */
int mcl_update_mbes(mcl_handle* h, const float* ranges, const float* beam_angles, int32_t n_beams,
                    double sigma, double r_max, const double sensor_offset[6]);
/* expected ranges of particles [first, first+count) x n_beams (host floats); parity/diagnostics */
int mcl_mbes_expected(mcl_handle* h, int64_t first, int64_t count, const float* beam_angles,
                      int32_t n_beams, double r_max, const double sensor_offset[6], float* out);

/* ---- landmark update with k-nearest-neighbour data association (BASELINE config 5; nearest
 * reference analogues: auv_ekf_localization/src/ekf_localization.cpp:479-524 max-likelihood
 * landmark, auv_ekf_slam/src/ekf_slam.cpp:100-103 chi-square gate).  Landmarks in the MAP frame
 * (n x 3 doubles); detections in the SENSOR frame (n_det x 3 doubles, NaN row = invalid);
 * 1 <= k <= 4; gate = chi-square threshold on |p - l|^2 / sigma^2; accumulate != 0 adds the
 * log-likelihood to the one an earlier update left (e.g. MBES ranges + landmarks of one ping). */
int mcl_set_landmarks(mcl_handle* h, const double* xyz, int64_t n_landmarks);
/* Mahalanobis association as in the reference's EKF-SLAM (auv_ekf_slam/src/ekf_slam_core.cpp:160-178,
 * correspondence_obj_mbes.cpp:26-35,110-120): innovation nu = z - R^T (l - o) in the SENSOR frame,
 * S = H Sigma H^T + Q, d_m = nu^T S^-1 nu gated against `gate`.  A particle is a pose hypothesis, so of the EKF's
 * covariance only the landmark's own 3x3 block Sigma_j remains and H is R^T: S = R^T Sigma_j R + Q.
 * cov6: n_landmarks x 6 (xx xy xz yy yz zz, MAP frame) or NULL = exact map; Q6: measurement covariance in the
 * sensor frame or NULL = sigma^2 I of the update call.  Both NULL switches back to the isotropic distance.
 * Affects mcl_update_landmarks (k-NN: each term weighted by 1/sqrt(det S)) and mcl_update_landmarks_assign
 * (table entries d_m as in the reference). */
int mcl_set_landmark_noise(mcl_handle* h, const double* cov6, const double Q6[6]);
int mcl_update_landmarks(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k,
                         double gate, const double sensor_offset[6], int32_t accumulate);
/* Landmark update with a GLOBAL assignment per particle: the correspondence table
 * of the reference's batch association (auv_ekf_slam/src/ekf_slam_core.cpp:172-178: distance if < gate,
 * else 10000; :269-281: one new-landmark hypothesis per detection at cost new_mh_dist; :298-312: Munkres
 * assignment) solved exactly for every particle; lw = -1/2 * optimal total - n_valid * lognorm.  A
 * landmark explains at most one detection.  n_det <= 16; 1 <= k_cand <= 8 = nearest gated landmarks
 * kept as candidates of a detection: when MORE than k_cand landmarks lie inside a detection's gate the farther
 * ones are dropped before the assignment, so on maps that dense the optimum can differ from the one over the
 * full table (part of the definition: the oracle applies the same rule; tests/test_gpu_landmark_assign.py
 * ::test_dense_cluster...).  With mcl_set_landmark_noise the table entries are the reference's Mahalanobis
 * distances.  assign_out (optional, host): n_keep x n_det int32 for the first
 * n_keep particles -- landmark index, -1 = new-landmark hypothesis, -2 = invalid (NaN) detection. */
int mcl_update_landmarks_assign(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k_cand,
                                double gate, double new_mh_dist, const double sensor_offset[6], int32_t accumulate,
                                int32_t* assign_out, int64_t n_keep);

/* ---- a8-a12 + a2: auv_pf.resample (auv_pf.py:169-198): normalise, resample, keep/lost/dupes
 * reassign, add_noise(resampling_noise_covariance).
 * uniforms: REPLAY: scheme-dependent draws in reference order (systematic: 1); NATIVE: NULL.
 * replay_normals: n x 6 post-resample noise draws (REPLAY) or NULL. */
int mcl_resample(mcl_handle* h, const double* uniforms, int64_t n_uniforms, const double* replay_normals);
/* number of uniforms the next mcl_resample consumes in REPLAY mode: systematic 1, stratified and
 * multinomial n, residual n - sum(floor(n w)) (resampling.py:74; computed here from the weights) */
int mcl_resample_prepare(mcl_handle* h, int64_t* n_uniforms);

/* ---- a13: loc_loop/update_loc_pose (auv_pf.py:218-285) */
int mcl_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]);
/* the same evaluation queued on the handle's stream without waiting for it: the result lands in the
 * pinned ring that mcl_last_mean_cov / mcl_mean_history read */
int mcl_mean_cov_async(mcl_handle* h);
/* PoseArray payload: n x 7 doubles (x,y,z,qx,qy,qz,qw), quaternion_from_euler per particle */
int mcl_get_poses(mcl_handle* h, double* pose7);

/* ---- state access (checkpoint / tests).  soa: 6 x n doubles; w: n normalised weights or NULL */
int mcl_get_particles(mcl_handle* h, double* soa, double* w);
int mcl_set_particles(mcl_handle* h, const double* soa);
int mcl_get_log_weights(mcl_handle* h, double* lw);
int mcl_set_log_weights(mcl_handle* h, const double* lw, int32_t weight_mode);
int mcl_get_last_indices(mcl_handle* h, int32_t* idx);      /* FilterPy-style ancestor indices, n */
int mcl_get_last_offspring_cdf(mcl_handle* h, uint32_t* ncum); /* n_global cumulative offspring counts */
/* the fixed-point weights of the last resample, q_i = floor(w_i 2^s), s = 63 - ceil(log2 n_global), this shard's n of them, and
 * their total over ALL shards: the integers every resampling decision was taken on (identical for any tiling / sharding) */
int mcl_get_fixed_weights(mcl_handle* h, uint64_t* q, uint64_t* total);

/* ---- one fused filter step, fully asynchronous on the handle's stream (bench / production):
 * predict -> MBES update -> normalise -> resample(+noise) -> mean/cov.  NATIVE rng only. */
int mcl_step_mbes(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges,
                  const float* beam_angles, int32_t n_beams, double sigma, double r_max,
                  const double sensor_offset[6]);
/* The same step with the landmark observation of the ping on top (BASELINE config 5): after the MBES update,
 * mcl_update_landmarks(det_xyz, n_det, lm_sigma, k, gate, lm_sensor_offset, accumulate = 1), then the resample and
 * mean / cov -- the same particles, weights and moments, bit for bit, as mcl_predict + mcl_update_mbes +
 * mcl_update_landmarks(accumulate) + mcl_resample leave, without the passes the separate calls need in between
 * (z / roll / pitch stores, the pose kernel, the max-lw reduction, a second moments pass). */
int mcl_step_mbes_landmarks(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges,
                            const float* beam_angles, int32_t n_beams, double sigma, double r_max,
                            const double sensor_offset[6], const double* det_xyz, int32_t n_det, double lm_sigma,
                            int32_t k, double gate, const double lm_sensor_offset[6]);
int mcl_sync(mcl_handle* h);
/* mean/cov computed by the last mcl_step_mbes (syncs the stream).  One process per GPU (mcl_comm_init): the sums of a fused
 * step travel with the NEXT step's shard records instead of an all-reduce of their own, so right after a step they may still
 * lie shard by shard -- mcl_last_mean_cov and mcl_mean_history then complete them with one all-reduce: COLLECTIVE calls in
 * that case, to be made by every rank (like mcl_get_last_indices after the O(n) exchange).  MCL_MOMENTS_RIDE=0 (read at the
 * first sharded resample) restores the all-reduce after every step. */
int mcl_last_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]);
/* mean poses (6 doubles each, oldest first) of the last `last_k` mean/cov evaluations (<= 4096 are
 * kept in a pinned ring): lets a caller score a whole asynchronous run without per-step syncs */
int mcl_mean_history(mcl_handle* h, int64_t last_k, double* mean6_out);

/* ---- resampling.py as free functions on the GPU (fixed-point CDF):
 * weights need not be normalised; out: n int32 ancestor indices. */
int mcl_resample_indices(int32_t scheme, const double* weights, int64_t n, const double* uniforms,
                         int64_t n_uniforms, int32_t device, int32_t* out);

/* ---- multi-GPU: particles shard by contiguous global id; RCCL all-reduce (max log-w),
 * all-gather (shard totals, offspring CDF, ancestor states).
 * PRECONDITION: every rank is given the SAME inputs per step (odometry message, ping, dt).  motion_pred leaves the
 * odometry's depth, roll and pitch on every particle (auv_particle.py:55-57,70), and a resample that directly follows
 * a predict does not ship those three components between shards: a copied particle takes them from the receiving
 * rank's own odometry.  Ranks fed different odometry would silently disagree on z, roll, pitch of copied particles. */
int mcl_comm_unique_id(char id[128]);
int mcl_comm_init(mcl_handle* h, const char id[128]); /* MCL_COMM_RCCL: collective over all ranks */
/* flags: MCL_COMM_NO_OVERLAP = no second communicator; the pre-resample state all-gather then runs
 * in line on the handle's stream instead of under the measurement update (also: env MCL_NO_OVERLAP=1) */
#define MCL_COMM_NO_OVERLAP 1u
int mcl_comm_init_ex(mcl_handle* h, const char id[128], uint32_t flags);
/* ranks = ncclAllReduce(sum) of 1 over the communicator (1 without one): the number of ranks RCCL really
 * connected; overlap (optional) = 1 if the second communicator for the overlapped gather exists */
int mcl_comm_ranks(mcl_handle* h, int32_t* ranks, int32_t* overlap);
/* Runs the collective pattern of one mcl_step_mbes (state all-gather on the second communicator while the
 * first one reduces and gathers) three times and waits for it with a deadline.  On a timeout BOTH
 * communicators are aborted (ncclCommAbort) and MCL_ERR_COMM is returned -- the caller can then
 * re-initialise with MCL_COMM_NO_OVERLAP under a fresh unique id, or exit: it never hangs. */
int mcl_comm_selftest(mcl_handle* h, int32_t timeout_ms);
/* destroy (abort = 0, after draining the streams) or abort the communicators; the handle stays usable */
int mcl_comm_shutdown(mcl_handle* h, int32_t abort);
/* MCL_COMM_LOCAL: all shards live in this process; the exchange steps are device copies.  The
 * sharded algorithm (and therefore every result bit) is the one the RCCL path runs. */
int mcl_group_resample(mcl_handle** shards, int32_t n_shards, const double* uniforms, int64_t n_uniforms,
                       const double* const* replay_normals /* per shard, or NULL */);
int mcl_group_mean_cov(mcl_handle** shards, int32_t n_shards, double mean6[6], double* yaw_mean,
                       double cov9[9]);
/* mcl_step_mbes over a LOCAL group: every shard runs the fused predict + MBES update, then the sharded resample
 * (+ moments) -- phase for phase what N ranks under RCCL execute, so a 4 M-particle BASELINE config 4 step can be
 * checked bit for bit against the unsharded filter on one GPU.  The result is read with mcl_last_mean_cov(shards[0]). */
int mcl_group_step_mbes(mcl_handle** shards, int32_t n_shards, const mcl_odom* odom, double dt, const float* ranges,
                        const float* beam_angles, int32_t n_beams, double sigma, double r_max,
                        const double sensor_offset[6]);
/* ... and mcl_step_mbes_landmarks over a LOCAL group (BASELINE config 5's step, shard for shard) */
int mcl_group_step_mbes_landmarks(mcl_handle** shards, int32_t n_shards, const mcl_odom* odom, double dt,
                                  const float* ranges, const float* beam_angles, int32_t n_beams, double sigma,
                                  double r_max, const double sensor_offset[6], const double* det_xyz, int32_t n_det,
                                  double lm_sigma, int32_t k, double gate, const double lm_sensor_offset[6]);
/* Resample exchange between shards.  Default: O(n) per rank -- every shard expands its own slice of the
 * offspring CDF, the shards all-gather two integers each (lost slots L_r, surplus copies S_r), and rank q sends rank r
 * exactly the surplus copies whose positions in the global dupes order fall into r's lost ranks (ncclSend / ncclRecv in
 * one group; one host synchronisation per resample to learn the sizes).  MCL_EXCHANGE=allgather (environment, read in
 * mcl_create) selects the all-gather of CDF + state of rounds 1-2.  Results are bit-identical.
 * After an O(n) exchange a shard holds only its own slice of the offspring CDF: mcl_get_last_indices and
 * mcl_get_last_offspring_cdf all-gather it on demand -- under RCCL that is a COLLECTIVE call (every rank makes it).
 * mcl_exchange_stats: particle states this shard sent to peers and lost slots it filled, summed since the last reset. */
int mcl_exchange_stats(mcl_handle* h, int64_t* states_sent, int64_t* lost_slots, int32_t reset);
/* Point-to-point operations (ncclSend + ncclRecv; device copies in a LOCAL group) this shard issued in `resamples`
 * exchanges since the last reset.  A surplus copy travels as ONE record of its non-uniform components (x, y, yaw right
 * after a predict) and the copies a peer needs are contiguous, so an exchange is at most 2 (world - 1) operations. */
int mcl_exchange_ops(mcl_handle* h, int64_t* p2p_ops, int64_t* resamples, int32_t reset);
/* The transfer plan of that exchange as pure host arithmetic (no device, no handle): given every shard's lost-slot
 * and surplus-copy counts, what `rank` sends to / receives from each peer r -- send_off / send_cnt: a range of rank's
 * packed surplus list; recv_off / recv_cnt: a range of rank's lost ranks; entry [rank] is the part that stays at home.
 * MCL_ERR_INVALID unless the counts add up (sum lost == sum surplus).  The library sizes its ncclSend / ncclRecv calls
 * with exactly this function; tests/test_exchange_plan.py checks its properties for random worlds on CPU. */
int mcl_exchange_plan(int32_t world, const uint32_t* lost, const uint32_t* surplus, int32_t rank, uint32_t* send_off,
                      uint32_t* send_cnt, uint32_t* recv_off, uint32_t* recv_cnt);

/* ---- instrumentation */
int mcl_timing_enable(mcl_handle* h, int32_t on);
int mcl_timing_get(mcl_handle* h, mcl_timing* out); /* syncs, returns and resets the accumulators */
/* Which kernels cast the last MBES update (syncs): path = 1 fan sweep (mcl_sweep.h: height grids, regularly triangulated
 * meshes, height-field TINs; ascending beam angles), 2 fan slice (mcl_slice.h: every other triangle mesh, ascending beam
 * angles), 0 ray traversal; handed_over = particles the sweep / slice passed on to the general kernel (paths 1, 2),
 * deferred_groups = groups of eight the fast traversal passed on to the general one (path 0); path 2: groups of 60 spatial
 * neighbours the group form of the slice left to the per-particle kernel, or -1 when every particle was cast on its own
 * (no visiting order: DESIGN.md 5).  Any pointer may be NULL. */
int mcl_mbes_last_path(mcl_handle* h, int32_t* path, int64_t* handed_over, int64_t* deferred_groups);
/* Who cast the particles the last MBES update's first stage handed over (syncs).  On a height-field TIN with HOLES (an
 * edge with no triangle behind it inside the bounding box: data gaps, a ragged outline) the adjacency walk goes on through
 * empty space wherever mesh_build could link the rim (DESIGN.md 5 "Holes and the outline"); where it could not (a hole
 * with an island, MCL_TIN_RIMS=0) a slice that reaches the edge ends the walk, and the fan slice (mcl_slice.h: any soup,
 * exact across gaps) casts those particles before the ray traversal is asked: by_slice = particles it cast,
 * by_traversal = particles that reached the general kernel (on every other map: all of handed_over).
 * MCL_HANDOVER_SLICE=0 sends them all to the traversal (A/B).  Any pointer may be NULL. */
int mcl_mbes_last_handover(mcl_handle* h, int64_t* by_slice, int64_t* by_traversal);
/* The order in which the last MBES update's fan sweep (or group slice) visited the particles (syncs): slots[p] = state
 * slot of the particle at position p (a wavefront of the sweep casts 64 consecutive positions); *sorted = 1 when the
 * positions follow the spatial order the previous resample prepared -- in a fused step, or in mcl_resample right after
 * mcl_predict + mcl_update_mbes, the node's call sequence -- (bins of x, y, yaw: DESIGN.md 5 "particle order"), 0 when they are
 * the slots themselves (slots[] is then the identity).  Only the visiting order ever changes: state slots, RNG keys and
 * keep / lost / dupes (auv_pf.py:183-198) do not, and no log-likelihood depends on it.  MCL_VISIT=0 switches the
 * spatial order off, MCL_VISIT=1 forces it for shards of any size (default: >= 393 216 particles).  slots: n entries. */
int mcl_mbes_visit_order(mcl_handle* h, uint32_t* slots, int32_t* sorted);

#ifdef __cplusplus
}
#endif
#endif /* MCL_H */
