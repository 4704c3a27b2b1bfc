// mcl_api.hip -- host side of libmcl_hip.so: the C ABI declared in include/mcl.h.
// C++ host code that owns the device buffers, orders kernels on one HIP stream per handle and
// runs the shard-exchange steps over RCCL (one process per GPU) or device copies (LOCAL group).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcl.h"
#include "mcl_kernels.h"
#include "mcl_mbes.h"
#include "mcl_sweep.h"
#include "mcl_mesh.h"
#include "mcl_resample.h"
#include "mcl_resample_alt.h"
#include "mcl_landmarks.h"

#define MEAN_RING 4096
#define RING_STRIDE 20  // doubles per mean/cov result: 16 payload + [16] format tag
// control block layout (bytes)
#define CTRL_SLOTS 0                       // MCL_MAX_SLOTS u64
#define CTRL_WORK (8 * MCL_MAX_SLOTS)      // int: groups deferred by the fast MBES kernel
#define CTRL_DEFER (CTRL_WORK + 4)         // int: particles the first sweep pass declined
#define CTRL_DEFER2 (CTRL_WORK + 8)        // int: particles the bounds-checked second pass handed to the traversal kernels
#define CTRL_T_QUANT (CTRL_WORK + 12)      // u32 tickets, self-resetting
#define CTRL_T_EXPAND (CTRL_WORK + 16)
#define CTRL_T_GATHER (CTRL_WORK + 20)
#define CTRL_BYTES 1024

namespace {

thread_local std::string g_create_err;

struct TimedRegion {
  hipEvent_t a, b;
  int k;
  bool open;
};

}  // namespace

struct mcl_handle {
  mcl_config cfg;
  long long n = 0, ng = 0, goff = 0;
  int rank = 0, world = 1;
  int device = 0;
  hipStream_t stream = nullptr;
  // particle state: two ping-pong SoA buffers of 6*n doubles; multi-shard: a global copy
  double* state[2] = {nullptr, nullptr};
  int cur = 0;
  double* state_glob = nullptr;  // 6*ng (world > 1)
  double* lw = nullptr;          // n log-weights
  double* wnorm = nullptr;       // n (lazily)
  u64* q = nullptr;              // n fixed-point weights
  u32* ncum = nullptr;           // ng offspring CDF (global)
  u32* zcum = nullptr;           // ng scratch (generic keep/lost/dupes of the explicit-index schemes)
  u32* zr = nullptr;             // n: rank of a lost slot among the lost slots, or ZR_SURVIVOR
  u32* dupes32 = nullptr;        // ng: dupes[k] = ancestor copied into the k-th lost slot
  u64* desc = nullptr;           // ntiles_glob look-back descriptors
  unsigned char* ctrl = nullptr; // control block: max-lw slots | MBES work counter | kernel tickets (CTRL_* offsets)
  u32 epoch = 0;                 // look-back epoch (one per k_cdf_expand launch)
  bool max_valid = false;        // the slots hold max lw of the current log-weights
  bool pose_ready = false;       // pose_dev already holds the records of the current state (fused predict)
  u64* tile64 = nullptr;
  u32* tile32 = nullptr;
  long long ntiles_loc = 0, ntiles_glob = 0;
  double* part = nullptr;     // reduction partials [7][MCL_MAX_GRID]
  double* scal = nullptr;     // device scalars: [0] max lw, [8..14] sums7, [16..21] cov6
  u64* totals = nullptr;      // device, world entries (+1 scratch)
  int* idx = nullptr;         // n (lazily)
  double* replay_dev = nullptr;
  double* pose7 = nullptr;
  double* host_pin_dev = nullptr;  // device-side address of host_pin (kernels write results into the ring directly)
  bool moments_direct = false;
  double* host_pin = nullptr;  // pinned ring: MEAN_RING entries of 16 doubles (sums7, pad, cov-sums6, pad2)
  long long mean_count = 0;     // number of mean/cov results produced so far
  // MBES
  float2* beam_sc = nullptr;
  float* ranges_dev = nullptr;
  const float* ranges_ptr = nullptr;  // where the ranges of this update are on the device (ranges_dev, or beside the sweep's beam table)
  bool ranges_pending = false;
  float* exp_dev = nullptr;
  MbesPose* pose_dev = nullptr;
  MbesGroup* mbes_groups = nullptr;  // one record per group of MBES_WAVES particles
  // visiting order for dispersed clouds: Morton keys, radix sort (rocPRIM), permutation
  u32 *sort_keys = nullptr, *sort_keys_out = nullptr, *sort_idx = nullptr, *mbes_perm = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  // pinned ring of 4 slots x 4 ints, one slot per MBES update: [0] groups the natural-order classification deferred,
  // [1] particles the sweep handed to the traversal kernels, [2] particles its first pass declined.  An update reads
  // the slot of the update TWO before it, after waiting for that update's event (long since complete when the host
  // runs ahead): the visiting-order and grid-size decisions are a function of the filter's history, never of timing.
  int* work_host = nullptr;
  hipEvent_t ev_upd[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long upd_seq = 0;
  int env_sort = -1;            // MCL_SORT_VISITS=0/1 forces the decision (tests, A/B)
  int* mbes_worklist = nullptr;  // ngroups + 1 ints; [ngroups] is the counter
  int* lm_worklist = nullptr;    // n + 1 ints; [n] is the counter (landmark assignment: particles with clashes)
  // alternative resamplers (lazily allocated)
  u64* cq = nullptr;       // inclusive scan of q
  u64* u53 = nullptr;      // uniforms as 53-bit integers
  u32 *cnt = nullptr, *first = nullptr, *flags = nullptr, *fcum = nullptr, *copies = nullptr, *ccum = nullptr;
  int* dupes = nullptr;
  double *cs = nullptr, *chunk = nullptr, *uni_dev = nullptr;
  long long residual_k = -1;  // copies count cached by mcl_resample_prepare
  bool idx_explicit = false;  // last resample produced idx[] directly (non-systematic)
  size_t exp_cap = 0;
  int beams_cap = 0;
  std::vector<float> beam_cache;  // last uploaded angles
  int beam_lo = -1, beam_hi = -1;  // extreme-angle beams (footprint shortcut)
  bool beams_sorted = false;
  // fan sweep (mcl_sweep.h)
  std::vector<float> ranges_host;   // last uploaded ranges (the sweep's beam table is built from them)
  bool sweep_angles_ok = false;     // ascending, finite, |a| <= 85 degrees
  int b_split = 0;
  float4* sweep_beams = nullptr;   // the table of the CURRENT update: one of sweep_buf[2]
  float* sweep_tail = nullptr;
  int sweep_cap = 0;
  // the table travels on its own stream into alternating device buffers, so the 12 KiB copy of ping k + 1 overlaps
  // the kernels of ping k instead of standing between two steps (4 us of copy + its launch gaps)
  float4* sweep_buf[2] = {nullptr, nullptr};
  float* sweep_stage[2] = {nullptr, nullptr};   // pinned staging, one per buffer
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  bool stage_used[2] = {false, false};
  int sweep_sel = 0;
  hipStream_t copy_stream = nullptr;
  u32* defer_idx = nullptr;
  u32* defer_idx2 = nullptr;        // what the bounds-checked second pass hands on
  int env_sweep = -1;               // MCL_SWEEP=0/1 forces the decision (tests, A/B)
  int env_nsub = 0;                 // MCL_SWEEP_NSUB=1/2/4 forces the lanes per particle side (A/B)
  bool sweep_now = false;           // decided by the first launch_mbes call of an update
  bool sweep_two_pass = false;      // lattice maps: a bounds-checked second pass precedes the traversal kernels
  int sweep_nvalid = 0;
  float* grid = nullptr;
  int gnx = 0, gny = 0;
  double gox = 0, goy = 0, gres = 1;
  float gzmin = 0, gzmax = 0;
  double gslope_max = 0;       // steepest patch gradient of the height grid (the fan sweep's tilt bound)
  MeshDev* mesh = nullptr;
  LandmarkDev* landmarks = nullptr;
  double* det_dev = nullptr;
  int det_cap = 0;
  int map_kind = -1;  // 0 grid, 1 mesh
  bool mesh_heightfield = false;
  bool force_general_mesh = false;  // MCL_MESH_GENERAL / MCL_MESH_UNSTRUCTURED: no structured-mesh fast path
  bool mesh_no_sweep = false;       // MCL_MESH_GENERAL: triangle-record traversal only (no adjacency sweep either)
  // bookkeeping
  int weight_mode = 0;
  bool have_lw = false, have_cdf = false, have_meancov = false;
  uint32_t step_predict = 0, step_resample = 0;
  bool timing = false;
  std::vector<TimedRegion> regions;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  mcl_timing tacc;
  ncclComm_t comm = nullptr;
  // overlap of the pre-resample state all-gather with the measurement update (second communicator,
  // second stream); falls back to an in-line gather when the split is unavailable
  ncclComm_t comm2 = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_state_ready = nullptr, ev_gather_done = nullptr;
  bool gather_inflight = false;
  // z, roll, pitch of every particle are the odometry's right after motion_pred: the exchange leaves them out
  bool uni_valid = false;       // true from a predict until the state is written by anything else
  double uni_val[3] = {0, 0, 0};
  unsigned gather_uni_mask = 0; // components the last state exchange skipped (phase_gather substitutes uni_val)
  bool fault_step = false;      // MCL_FAULT_INJECT=step_after_predict (tests): the fused step fails after its predict
  bool uni_deferred = false;    // fused step in flight: the predict kernel did NOT store z, roll, pitch (the gather of
                                // the same call substitutes them; materialise_uniform() on any other way out)
  // O(n)-per-rank resample exchange: hand-over records {L | S << 32, x0, y0, z0} of every shard,
  // surplus copies packed for the peers, copies received for this shard's lost slots
  bool exch_allgather = false;   // MCL_EXCHANGE=allgather: the all-gather exchange of rounds 1-2 instead
  u64* lsx = nullptr;            // device, world x 4 words
  u64* lsx_host = nullptr;       // pinned, world x 4 words + the sequence word k_publish_ls writes last
  u64* lsx_host_dev = nullptr;   // its device-side address
  u64 ls_seq = 0;
  double* xsend = nullptr;       // 6 x xsend_cap
  size_t xsend_cap = 0;
  double* xrecv = nullptr;       // 6 x n
  std::vector<u32> ex_L, ex_S;   // per shard, filled by exchange_ls
  std::vector<u32> ex_Lpre, ex_Spre;
  unsigned long long ex_sent = 0, ex_lost = 0;  // particle states sent to peers / lost slots, summed over the resamples
  bool cdf_global = false;       // ncum holds the GLOBAL offspring CDF (else only this shard's slice)
  std::vector<mcl_handle*> group;  // LOCAL group this shard was last resampled in (lazy CDF all-gather)
  // environment switches, read once in mcl_create (never on the per-measurement path)
  bool env_debug_work = false, env_force_comm = false, env_no_overlap = false;
  // pinned staging so that asynchronous uploads never read caller-owned pageable memory after the call returns
  struct PinSlot {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;  // recorded after the async copy out of this slot
    bool used = false;
  } pin_ring[8];
  unsigned pin_next = 0;
  int* asg_dev = nullptr;  // landmark assignment output (cached, grown on demand)
  size_t asg_cap = 0;
  std::string err;
};

namespace {

#define HIPCHK(h, call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (h)->err = buf_;                                                                         \
      return MCL_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)
#define NCCLCHK(h, call)                                                                       \
  do {                                                                                         \
    ncclResult_t e_ = (call);                                                                  \
    if (e_ != ncclSuccess) {                                                                   \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(e_), __FILE__, __LINE__); \
      (h)->err = buf_;                                                                         \
      return MCL_ERR_COMM;                                                                     \
    }                                                                                          \
  } while (0)
#define RET_IF(x)           \
  do {                      \
    int rc_ = (x);          \
    if (rc_ != MCL_OK) return rc_; \
  } while (0)

int fail(mcl_handle* h, int code, const char* msg) {
  if (h) h->err = msg;
  return code;
}

int grid_for(long long n, int block = MCL_BLOCK) {
  long long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > MCL_MAX_GRID) g = MCL_MAX_GRID;
  return (int)g;
}
int grid_tiles(long long ntiles) {
  if (ntiles < 1) ntiles = 1;
  return (int)(ntiles > MCL_MAX_GRID ? MCL_MAX_GRID : ntiles);
}

StatePtrs state_ptrs(double* base, long long n) {
  StatePtrs s;
  for (int c = 0; c < 6; ++c) s.c[c] = base + (size_t)c * n;
  return s;
}

void t_begin(mcl_handle* h, int k) {
  if (!h->timing) return;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!h->ev_pool.empty()) {
    ev = h->ev_pool.back();
    h->ev_pool.pop_back();
  } else {
    // timing-only events: no system-scope fence when they are recorded (a default event releases / acquires at system
    // scope -- a cache write-back and invalidate around every timed region, which made the regions ~20 % longer than
    // the kernels inside them are under rocprofv3)
    if (hipEventCreateWithFlags(&ev.first, hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreate(&ev.first);
    if (hipEventCreateWithFlags(&ev.second, hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreate(&ev.second);
  }
  (void)hipEventRecord(ev.first, h->stream);
  h->regions.push_back(TimedRegion{ev.first, ev.second, k, true});
}
// closes the innermost open region (regions nest: MCL_K_MBES_MAIN inside MCL_K_UPDATE_MBES)
void t_end(mcl_handle* h) {
  if (!h->timing) return;
  for (size_t r = h->regions.size(); r-- > 0;)
    if (h->regions[r].open) {
      h->regions[r].open = false;
      (void)hipEventRecord(h->regions[r].b, h->stream);
      return;
    }
}
void t_collect(mcl_handle* h) {
  for (auto& r : h->regions) {
    float ms = 0.f;
    if (r.open) (void)hipEventRecord(r.b, h->stream);  // (an error return left it open)
    (void)hipEventSynchronize(r.b);
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    h->tacc.ms[r.k] += ms;
    h->tacc.launches[r.k] += 1;
    h->ev_pool.push_back({r.a, r.b});
  }
  h->regions.clear();
}

// euler_from_quaternion(q,'sxyz') -- tf.transformations' published algorithm (auv_particle.py:50)
void euler_from_quat(const double qin[4], double rpy[3]) {
  double nq = qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3];
  double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (nq >= 2.220446049250313e-16 * 4.0) {
    double s = std::sqrt(2.0 / nq);
    double q[4] = {qin[0] * s, qin[1] * s, qin[2] * s, qin[3] * s};
    double o[4][4];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) o[a][b] = q[a] * q[b];
    M[0] = 1.0 - o[1][1] - o[2][2];
    M[1] = o[0][1] - o[2][3];
    M[2] = o[0][2] + o[1][3];
    M[3] = o[0][1] + o[2][3];
    M[4] = 1.0 - o[0][0] - o[2][2];
    M[5] = o[1][2] - o[0][3];
    M[6] = o[0][2] - o[1][3];
    M[7] = o[1][2] + o[0][3];
    M[8] = 1.0 - o[0][0] - o[1][1];
  }
  double cy = std::sqrt(M[0] * M[0] + M[3] * M[3]);
  if (cy > 2.220446049250313e-16 * 4.0) {
    rpy[0] = std::atan2(M[7], M[8]);
    rpy[1] = std::atan2(-M[6], cy);
    rpy[2] = std::atan2(M[3], M[0]);
  } else {
    rpy[0] = std::atan2(-M[5], M[4]);
    rpy[1] = std::atan2(-M[6], cy);
    rpy[2] = 0.0;
  }
}

void philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0;
  o[1] = c1;
  o[2] = c2;
  o[3] = c3;
}
uint64_t native_u53(uint64_t seed, uint32_t step) {
  uint32_t o[4];
  philox_host(0xFFFFFFFFu, 0u, step, 3u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return ((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6);
}

int ceil_log2(long long n) {
  int l = 0;
  while ((1ll << l) < n) ++l;
  return l;
}

NoiseArgs noise_args(const mcl_handle* h, const double cov[6], uint32_t purpose, uint32_t step) {
  NoiseArgs a;
  for (int c = 0; c < 6; ++c) a.sq[c] = std::sqrt(cov[c]);
  a.k0 = (uint32_t)h->cfg.seed;
  a.k1 = (uint32_t)(h->cfg.seed >> 32);
  a.step = step;
  a.purpose = purpose;
  a.gid0 = h->goff;
  return a;
}

// Host -> device upload that honours "the caller owns every host buffer" (include/mcl.h): when the call
// returns the caller may overwrite `src`.  Small payloads (ranges, detections, uniforms) are copied into a
// ring of pinned slots and travel asynchronously; large ones (REPLAY normals: the parity path, not the
// production path) are copied synchronously.
int upload(mcl_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return MCL_OK;
  if (bytes > (1u << 20)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
  }
  mcl_handle::PinSlot& sl = h->pin_ring[h->pin_next++ % 8u];
  if (sl.used) HIPCHK(h, hipEventSynchronize(sl.ev));
  if (sl.cap < bytes) {
    if (sl.p) (void)hipHostFree(sl.p);
    sl.p = nullptr;
    sl.cap = 0;
    size_t cap = 4096;
    while (cap < bytes) cap <<= 1;
    HIPCHK(h, hipHostMalloc(&sl.p, cap, hipHostMallocDefault));
    sl.cap = cap;
  }
  if (!sl.ev) HIPCHK(h, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  memcpy(sl.p, src, bytes);
  HIPCHK(h, hipMemcpyAsync(dst, sl.p, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(sl.ev, h->stream));
  sl.used = true;
  return MCL_OK;
}

int upload_replay(mcl_handle* h, const double* normals) {
  if (!h->replay_dev) HIPCHK(h, hipMalloc(&h->replay_dev, sizeof(double) * 6 * (size_t)h->n));
  return upload(h, h->replay_dev, normals, sizeof(double) * 6 * (size_t)h->n);
}

int set_device(mcl_handle* h) {
  HIPCHK(h, hipSetDevice(h->device));
  return MCL_OK;
}

// ------------------------------------------------------------------------------------------
// resample pipeline, written over a set of shards so that the RCCL path (one shard per process)
// and the LOCAL test group (several shards in one process) execute the same phases.
// ------------------------------------------------------------------------------------------
u64* ctrl_slots(mcl_handle* h) { return (u64*)(h->ctrl + CTRL_SLOTS); }
u32* ctrl_u32(mcl_handle* h, int off) { return (u32*)(h->ctrl + off); }

// max lw into the slots (unless the update kernel that wrote lw already did it)
int ensure_max_slots(mcl_handle* h) {
  if (h->max_valid) return MCL_OK;
  HIPCHK(h, hipMemsetAsync(h->ctrl + CTRL_SLOTS, 0, 8 * MCL_MAX_SLOTS, h->stream));
  k_max_slots<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->lw, h->n, ctrl_slots(h));
  HIPCHK(h, hipGetLastError());
  h->max_valid = true;
  return MCL_OK;
}
// scal[0] = local max lw (the value the shard exchange reduces)
int phase_local_max(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_NORMALISE);
  RET_IF(ensure_max_slots(h));
  k_max_finish<<<1, 64, 0, h->stream>>>(ctrl_slots(h), h->scal);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

int exchange_max(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      NCCLCHK(h, ncclAllReduce(h->scal, h->scal, 1, ncclDouble, ncclMax, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  double m = -INFINITY;
  for (int s = 0; s < ns; ++s) {
    double v;
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(&v, sh[s]->scal, sizeof(double), hipMemcpyDeviceToHost, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    if (v > m) m = v;
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->scal, &m, sizeof(double), hipMemcpyHostToDevice, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}

// fixed-point weights, tile sums, exclusive tile offsets and the shard total in ONE launch.
// from_slots: single shard -- the kernel reads the maximum straight from the slots (no k_max_finish)
// fused_next: k_cdf_expand<true> follows and adds the tile sums up itself (no tile scan launch)
int phase_quantise(mcl_handle* h, bool from_slots, bool fused_next = false) {
  RET_IF(set_device(h));
  const double scale = std::ldexp(1.0, 63 - ceil_log2(h->ng));
  t_begin(h, MCL_K_NORMALISE);
  if (from_slots) RET_IF(ensure_max_slots(h));
  QuantArgs a;
  a.lw = h->lw;
  a.n = h->n;
  a.slots = from_slots ? ctrl_slots(h) : nullptr;
  a.m_lw = h->scal;
  a.mode = h->weight_mode;
  a.scale = scale;
  a.q = h->q;
  a.tile_sum = h->tile64;
  k_quantise_tiles<<<(unsigned)h->ntiles_loc, MCL_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  if (!fused_next) {
    // sharded / explicit-position schemes: exclusive tile offsets and the shard total as separate arrays
    t_begin(h, MCL_K_SCAN);
    k_scan_tile_sums<u64><<<1, 1024, 0, h->stream>>>(h->tile64, h->ntiles_loc, h->totals + h->rank);
    t_end(h);
  }
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

int exchange_totals(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      NCCLCHK(h, ncclAllGather(h->totals + h->rank, h->totals, 1, ncclUint64, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  std::vector<u64> t(ns);
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(&t[s], sh[s]->totals + s, sizeof(u64), hipMemcpyDeviceToHost, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->totals, t.data(), sizeof(u64) * ns, hipMemcpyHostToDevice, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}

int phase_cdf(mcl_handle* h, uint64_t u53) {
  RET_IF(set_device(h));
  CdfArgs a;
  a.totals = h->totals;
  a.rank = h->rank;
  a.world = h->world;
  a.n_global = (u64)h->ng;
  a.u53 = u53;
  t_begin(h, MCL_K_SCAN);
  k_offspring_cdf<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->tile64, a,
                                                                          h->ncum + h->goff);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// all-gather of the offspring CDF slices and of the pre-resample state (north_star: "all-gather
// before resampling"); shards are contiguous and equal-sized
int exchange_cdf_state(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      if (h->gather_inflight) {
        // the state went out right after predict and travelled under the measurement update
        NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
        h->gather_inflight = false;
      } else {
        h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
        NCCLCHK(h, ncclGroupStart());
        NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
        for (int c = 0; c < 6; ++c)
          if (!((h->gather_uni_mask >> c) & 1u))
            NCCLCHK(h, ncclAllGather(h->state[h->cur] + (size_t)c * h->n, h->state_glob + (size_t)c * h->ng,
                                     (size_t)h->n, ncclDouble, h->comm, h->stream));
        NCCLCHK(h, ncclGroupEnd());
      }
      t_end(h);
    }
    return MCL_OK;
  }
  for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  for (int d = 0; d < ns; ++d) {
    mcl_handle* D = sh[d];
    D->gather_uni_mask = D->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
    RET_IF(set_device(D));
    for (int s = 0; s < ns; ++s) {
      mcl_handle* S = sh[s];
      if (s != d)
        HIPCHK(D, hipMemcpyAsync(D->ncum + S->goff, S->ncum + S->goff, sizeof(u32) * (size_t)S->n,
                                 hipMemcpyDefault, D->stream));
      for (int c = 0; c < 6; ++c)
        if (!((D->gather_uni_mask >> c) & 1u))
          HIPCHK(D, hipMemcpyAsync(D->state_glob + (size_t)c * D->ng + S->goff, S->state[S->cur] + (size_t)c * S->n,
                                   sizeof(double) * (size_t)S->n, hipMemcpyDefault, D->stream));
    }
    HIPCHK(D, hipStreamSynchronize(D->stream));
  }
  return MCL_OK;
}

// The pre-resample state is final once predict has run (updates only read it): send it on the
// second communicator/stream so the 48 B x N_global all-gather overlaps the ray-cast.
// an overlapped gather that will not be consumed (error return, state overwritten by the caller):
// let it finish, then forget it, so the next resample gathers the state it actually resamples
int cancel_state_gather(mcl_handle* h) {
  if (!h->gather_inflight) return MCL_OK;
  h->gather_inflight = false;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
  return MCL_OK;
}

int start_state_gather(mcl_handle* h) {
  if (!h->comm2 || !h->state_glob) return MCL_OK;
  HIPCHK(h, hipEventRecord(h->ev_state_ready, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_state_ready, 0));
  // 24 B instead of 48 B per particle of the GLOBAL cloud when z, roll, pitch are the odometry's on every particle
  h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;
  NCCLCHK(h, ncclGroupStart());
  for (int c = 0; c < 6; ++c)
    if (!((h->gather_uni_mask >> c) & 1u))
      NCCLCHK(h, ncclAllGather(h->state[h->cur] + (size_t)c * h->n, h->state_glob + (size_t)c * h->ng, (size_t)h->n,
                               ncclDouble, h->comm2, h->comm_stream));
  NCCLCHK(h, ncclGroupEnd());
  HIPCHK(h, hipEventRecord(h->ev_gather_done, h->comm_stream));
  h->gather_inflight = true;
  return MCL_OK;
}

// lost-slot ranks + dupes list.  fused_cdf: single shard, the offspring CDF is computed in the same pass
int phase_expand(mcl_handle* h, bool fused_cdf, uint64_t u53) {
  RET_IF(set_device(h));
  ExpandArgs a;
  memset(&a, 0, sizeof a);
  a.q = h->q;
  a.tile_sum = h->tile64;
  a.n_fine = h->ntiles_loc;
  a.n_global_u = (u64)h->ng;
  a.u53 = u53;
  a.total_out = h->totals + h->rank;
  a.ncum = h->ncum;
  a.n = h->ng;
  a.own0 = h->goff;
  a.own_n = h->n;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.desc = h->desc;
  a.ticket = ctrl_u32(h, CTRL_T_EXPAND);
  a.epoch = ++h->epoch;
  const unsigned grid = (unsigned)((h->ng + RS_TILE - 1) / RS_TILE);
  t_begin(h, fused_cdf ? MCL_K_SCAN : MCL_K_RESAMPLE);
  if (fused_cdf)
    k_cdf_expand<true><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  else
    k_cdf_expand<false><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// ---- O(n)-per-rank exchange -------------------------------------------------------------------------------
int alloc_lsx(mcl_handle* h) {
  if (h->lsx) return MCL_OK;
  HIPCHK(h, hipMalloc(&h->lsx, sizeof(u64) * 4 * (size_t)h->world));
  HIPCHK(h, hipHostMalloc(&h->lsx_host, sizeof(u64) * (4 * (size_t)h->world + 1), hipHostMallocMapped | hipHostMallocCoherent));  // (fine-grained: the host polls it while kernels run)
  memset(h->lsx_host, 0, sizeof(u64) * (4 * (size_t)h->world + 1));
  if (hipHostGetDevicePointer((void**)&h->lsx_host_dev, h->lsx_host, 0) != hipSuccess) h->lsx_host_dev = nullptr;
  return MCL_OK;
}
int launch_pack(mcl_handle* h, u64 publish_seq = 0);
// CDF, lost ranks and dupes list of THIS shard only (k_cdf_expand<true> over the shard, global weight offsets from the
// all-gathered totals); leaves the shard's hand-over record in lsx[rank]
int phase_expand_local(mcl_handle* h, uint64_t u53) {
  RET_IF(set_device(h));
  RET_IF(alloc_lsx(h));
  ExpandArgs a;
  memset(&a, 0, sizeof a);
  a.q = h->q;
  a.tile_sum = h->tile64;  // exclusive offsets (k_scan_tile_sums)
  a.n_fine = h->ntiles_loc;
  a.n_global_u = (u64)h->ng;
  a.u53 = u53;
  a.total_out = h->totals + h->world;  // (unused in this mode)
  a.ncum = h->ncum + h->goff;
  a.n = h->n;
  a.own0 = 0;
  a.own_n = h->n;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.desc = h->desc;
  a.ticket = ctrl_u32(h, CTRL_T_EXPAND);
  a.epoch = ++h->epoch;
  a.totals = h->totals;
  a.rank = h->rank;
  a.world = h->world;
  a.ls_out = h->lsx + 4 * (size_t)h->rank;
  for (int c = 0; c < 3; ++c) a.p0[c] = h->state[h->cur] + (size_t)c * h->n;
  const unsigned grid = (unsigned)((h->n + RS_TILE - 1) / RS_TILE);
  t_begin(h, MCL_K_SCAN);
  k_cdf_expand<true><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cdf_global = h->world == 1;
  return MCL_OK;
}

// every shard learns {L, S} of every shard (and the position of global particle 0); the host needs them to size the
// point-to-point transfers: ONE stream synchronisation per resample
int exchange_ls(mcl_handle** sh, int ns) {
  const int world = sh[0]->world;
  if (ns == 1) {
    mcl_handle* h = sh[0];
    t_begin(h, MCL_K_COMM);
    if (h->comm && world > 1)
      NCCLCHK(h, ncclAllGather(h->lsx + 4 * (size_t)h->rank, h->lsx, 4, ncclUint64, h->comm, h->stream));
    if (h->lsx_host_dev) {
      // the records travel to pinned memory by a kernel that writes a sequence word last; the pack kernel (sized on
      // the device from the same records) is queued behind it BEFORE the host starts to wait, so the GPU keeps
      // working while the host wakes up; the host spins on the word instead of synchronising the stream
      const u64 seq = ++h->ls_seq;
      t_end(h);
      RET_IF(launch_pack(h, seq));   // (its first workgroup publishes the records before it packs)
      volatile u64* flag = h->lsx_host + 4 * (size_t)world;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xfffu) == 0u) {
          if (hipStreamQuery(h->stream) == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
            return fail(h, MCL_ERR_HIP, "resample exchange: the hand-over records never arrived");
          if (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > 60)
            return fail(h, MCL_ERR_COMM, "resample exchange: timed out waiting for the hand-over records");
        }
        __builtin_ia32_pause();
      }
    } else {
      HIPCHK(h, hipMemcpyAsync(h->lsx_host, h->lsx, sizeof(u64) * 4 * (size_t)world, hipMemcpyDeviceToHost, h->stream));
      t_end(h);
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  } else {
    for (int s = 0; s < ns; ++s) {
      mcl_handle* h = sh[s];
      RET_IF(set_device(h));
      HIPCHK(h, hipMemcpyAsync(h->lsx_host + 4 * (size_t)s, h->lsx + 4 * (size_t)s, sizeof(u64) * 4, hipMemcpyDeviceToHost, h->stream));
    }
    for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    for (int d = 0; d < ns; ++d) {
      for (int s = 0; s < ns; ++s)
        if (s != d) memcpy(sh[d]->lsx_host + 4 * (size_t)s, sh[s]->lsx_host + 4 * (size_t)s, sizeof(u64) * 4);
      RET_IF(set_device(sh[d]));
      HIPCHK(sh[d], hipMemcpyAsync(sh[d]->lsx, sh[d]->lsx_host, sizeof(u64) * 4 * (size_t)ns, hipMemcpyHostToDevice, sh[d]->stream));
    }
  }
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    h->ex_L.assign(world, 0u);
    h->ex_S.assign(world, 0u);
    h->ex_Lpre.assign(world + 1, 0u);
    h->ex_Spre.assign(world + 1, 0u);
    for (int r = 0; r < world; ++r) {
      h->ex_L[r] = (u32)(h->lsx_host[4 * (size_t)r] & 0xffffffffull);
      h->ex_S[r] = (u32)(h->lsx_host[4 * (size_t)r] >> 32);
      h->ex_Lpre[r + 1] = h->ex_Lpre[r] + h->ex_L[r];
      h->ex_Spre[r + 1] = h->ex_Spre[r] + h->ex_S[r];
    }
    if (h->ex_Lpre[world] != h->ex_Spre[world])
      return fail(h, MCL_ERR_COMM, "resample exchange: lost slots and surplus copies of the shards do not add up (ranks fed different inputs?)");
  }
  return MCL_OK;
}

// surplus copies into the send buffer (own lost slots: straight into the receive buffer).  The kernel takes its sizes
// from the records on the device, so it can be queued before the host has read them (exchange_ls)
int launch_pack(mcl_handle* h, u64 publish_seq) {
  RET_IF(set_device(h));
  if (!h->xrecv) HIPCHK(h, hipMalloc(&h->xrecv, sizeof(double) * 6 * (size_t)h->n));
  if (!h->xsend) {
    // a shard's surplus is statistically a few per cent of its slots; n / 4 entries to start with, grown on demand
    const size_t cap = std::max<size_t>((size_t)h->n / 4, 4096);
    HIPCHK(h, hipMalloc(&h->xsend, sizeof(double) * 6 * cap));
    h->xsend_cap = cap;
  }
  h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
  PackArgs a;
  a.src = state_ptrs(h->state[h->cur], h->n);
  a.dupes = h->dupes32;
  a.lsx = h->lsx;
  a.rank = h->rank;
  a.world = h->world;
  a.cap = (u32)std::min<size_t>(h->xsend_cap, 0xffffffffull);
  a.uni_mask = h->gather_uni_mask;
  a.send = state_ptrs(h->xsend, (long long)h->xsend_cap);
  a.recv = state_ptrs(h->xrecv, h->n);
  a.host_words = publish_seq ? h->lsx_host_dev : nullptr;
  a.host_seq = publish_seq ? h->lsx_host_dev + 4 * (size_t)h->world : nullptr;
  a.seq = publish_seq;
  t_begin(h, MCL_K_RESAMPLE);
  k_pack_dupes<<<(unsigned)std::min<long long>(grid_for(h->n), 512), MCL_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
// after the host has read the records: the rare shard whose surplus exceeds the send buffer grows it and packs again;
// LOCAL groups pack here in the first place
int phase_pack(mcl_handle* h, bool already_packed) {
  RET_IF(set_device(h));
  const u32 S = h->ex_S[h->rank];
  if (already_packed && (size_t)S <= h->xsend_cap) return MCL_OK;
  if ((size_t)S > h->xsend_cap) {
    if (h->xsend) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      (void)hipFree(h->xsend);
      h->xsend = nullptr;
    }
    const size_t cap = std::max<size_t>((size_t)S + (size_t)S / 4, 4096);
    HIPCHK(h, hipMalloc(&h->xsend, sizeof(double) * 6 * cap));
    h->xsend_cap = cap;
  }
  return launch_pack(h);
}

// the range of global dupes positions that shard `from` holds and shard `to` needs: [lo, hi).  Lpre / Spre: exclusive
// prefix sums of the shards' lost-slot and surplus-copy counts (world + 1 entries).  Pure host arithmetic: also what
// mcl_exchange_plan exposes, so the plan is property-tested without a GPU (tests/test_exchange_plan.py).
void plan_range(const u32* Lpre, const u32* Spre, int from, int to, u32& lo, u32& hi) {
  lo = std::max(Spre[from], Lpre[to]);
  hi = std::min(Spre[from + 1], Lpre[to + 1]);
  if (hi < lo) hi = lo;
}
void ex_range(const mcl_handle* h, int from, int to, u32& lo, u32& hi) {
  plan_range(h->ex_Lpre.data(), h->ex_Spre.data(), from, to, lo, hi);
}

int exchange_dupes(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    h->ex_lost += h->ex_L[h->rank];
    if (!h->comm || h->world == 1) return MCL_OK;
    const int q = h->rank;
    t_begin(h, MCL_K_COMM);
    NCCLCHK(h, ncclGroupStart());
    for (int r = 0; r < h->world; ++r) {
      if (r == q) continue;
      u32 lo, hi;
      ex_range(h, q, r, lo, hi);  // what I hold and r needs
      if (hi > lo) {
        h->ex_sent += hi - lo;
        for (int c = 0; c < 6; ++c)
          if (!((h->gather_uni_mask >> c) & 1u))
            NCCLCHK(h, ncclSend(h->xsend + (size_t)c * h->xsend_cap + (lo - h->ex_Spre[q]), (size_t)(hi - lo), ncclDouble, r, h->comm, h->stream));
      }
      ex_range(h, r, q, lo, hi);  // what r holds and I need
      if (hi > lo)
        for (int c = 0; c < 6; ++c)
          if (!((h->gather_uni_mask >> c) & 1u))
            NCCLCHK(h, ncclRecv(h->xrecv + (size_t)c * h->n + (lo - h->ex_Lpre[q]), (size_t)(hi - lo), ncclDouble, r, h->comm, h->stream));
    }
    NCCLCHK(h, ncclGroupEnd());
    t_end(h);
    return MCL_OK;
  }
  // LOCAL group: the same ranges as device copies (after every shard has packed)
  for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  for (int d = 0; d < ns; ++d) {
    mcl_handle* D = sh[d];
    RET_IF(set_device(D));
    D->ex_lost += D->ex_L[d];
    for (int s = 0; s < ns; ++s) {
      if (s == d) continue;
      mcl_handle* S = sh[s];
      u32 lo, hi;
      ex_range(D, s, d, lo, hi);
      if (hi <= lo) continue;
      S->ex_sent += hi - lo;
      for (int c = 0; c < 6; ++c)
        if (!((D->gather_uni_mask >> c) & 1u))
          HIPCHK(D, hipMemcpyAsync(D->xrecv + (size_t)c * D->n + (lo - D->ex_Lpre[d]),
                                   S->xsend + (size_t)c * S->xsend_cap + (lo - S->ex_Spre[s]),
                                   sizeof(double) * (size_t)(hi - lo), hipMemcpyDefault, D->stream));
    }
  }
  for (int d = 0; d < ns; ++d) HIPCHK(sh[d], hipStreamSynchronize(sh[d]->stream));
  return MCL_OK;
}

// the global offspring CDF on demand (mcl_get_last_indices / mcl_get_last_offspring_cdf after an O(n) exchange, which
// leaves only the shard's own slice): RCCL -- a COLLECTIVE all-gather, every rank must make the call; LOCAL group --
// copies from the peers' slices
int ensure_global_cdf(mcl_handle* h) {
  if (h->cdf_global || h->world == 1) return MCL_OK;
  RET_IF(set_device(h));
  if (h->comm) {
    NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
  } else if ((int)h->group.size() == h->world) {
    for (mcl_handle* S : h->group) {
      if (S == h) continue;
      HIPCHK(S, hipStreamSynchronize(S->stream));
      HIPCHK(h, hipMemcpyAsync(h->ncum + S->goff, S->ncum + S->goff, sizeof(u32) * (size_t)S->n, hipMemcpyDefault, h->stream));
    }
  } else {
    return fail(h, MCL_ERR_STATE, "the global offspring CDF needs the communicator or the LOCAL group of the last resample");
  }
  h->cdf_global = true;
  return MCL_OK;
}

// reassign gather + resampling noise (+ the sums of update_loc_pose of the new state when with_moments)
int phase_gather(mcl_handle* h, const double* replay_normals, bool with_moments) {
  RET_IF(set_device(h));
  if (replay_normals) RET_IF(upload_replay(h, replay_normals));
  GatherArgs a;
  const bool multi = h->world > 1 || h->comm;
  const bool p2p = multi && !h->exch_allgather;
  a.src = (multi && !p2p) ? state_ptrs(h->state_glob, h->ng) : state_ptrs(h->state[h->cur], h->n);
  a.recv_mode = p2p ? 1 : 0;
  a.recv = p2p ? state_ptrs(h->xrecv, h->n) : a.src;
  a.shift_dev = p2p ? (const double*)(h->lsx + 1) : nullptr;
  a.dst = state_ptrs(h->state[h->cur ^ 1], h->n);
  a.n = h->n;
  a.goff = h->goff;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.nz = noise_args(h, h->cfg.resample_cov, 2u, h->step_resample);
  a.add_noise = 1;
  a.part = h->part;
  a.sums_out = h->scal + 32;
  // (single shard: the gather reads straight from the pre-resample state; after a predict z, roll, pitch are the same
  //  three numbers on every particle, so they are substituted instead of read -- bit-identical, 24 B x N less traffic)
  a.uni_mask = multi ? h->gather_uni_mask : (h->uni_valid ? 0x1cu : 0u);
  for (int c = 0; c < 6; ++c) a.uni[c] = (c >= 2 && c <= 4) ? h->uni_val[c - 2] : 0.0;
  if (with_moments && !multi && h->host_pin_dev) {
    // single shard: the last block writes the sums straight into the pinned ring entry (no copy command);
    // the host reads it only after synchronising the stream
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 1.0;
    a.sums_out = h->host_pin_dev + RING_STRIDE * (h->mean_count % MEAN_RING);
    h->moments_direct = true;
  } else {
    h->moments_direct = false;
  }
  a.ticket = ctrl_u32(h, CTRL_T_GATHER);
  const double* rp = replay_normals ? h->replay_dev : nullptr;
  t_begin(h, MCL_K_RESAMPLE);
  // one particle per thread up to 256 blocks (= 256 tickets), grid-stride beyond
  long long gg = (h->n + RS_BLOCK - 1) / RS_BLOCK;
  gg = gg < 1 ? 1 : (gg > GATHER_MAX_GRID ? GATHER_MAX_GRID : gg);
  if (with_moments)
    k_resample_gather<true><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  else
    k_resample_gather<false><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cur ^= 1;
  h->uni_valid = false;  // (the new state carries resampling noise)
  h->uni_deferred = false;
  h->gather_uni_mask = 0u;
  h->step_resample++;
  h->have_cdf = true;
  h->have_lw = false;
  h->pose_ready = false;
  return MCL_OK;
}


// ------------------------------------------------------------------------------------------
// stratified / multinomial / residual (single shard): explicit ancestor vector + generic reassign
// ------------------------------------------------------------------------------------------
template <class T>
int lazy_alloc(mcl_handle* h, T** p, size_t count) {
  if (!*p) HIPCHK(h, hipMalloc(p, sizeof(T) * count));
  return MCL_OK;
}
int alt_alloc(mcl_handle* h) {
  const size_t n = (size_t)h->n;
  RET_IF(lazy_alloc(h, &h->cq, n));
  RET_IF(lazy_alloc(h, &h->u53, n));
  RET_IF(lazy_alloc(h, &h->cnt, n));
  RET_IF(lazy_alloc(h, &h->first, n));
  RET_IF(lazy_alloc(h, &h->flags, n));
  RET_IF(lazy_alloc(h, &h->fcum, n));
  RET_IF(lazy_alloc(h, &h->copies, n));
  RET_IF(lazy_alloc(h, &h->ccum, n));
  RET_IF(lazy_alloc(h, &h->dupes, n));
  RET_IF(lazy_alloc(h, &h->cs, n));
  RET_IF(lazy_alloc(h, &h->chunk, n / 8192 + 2));
  RET_IF(lazy_alloc(h, &h->uni_dev, n));
  RET_IF(lazy_alloc(h, &h->wnorm, n));
  RET_IF(lazy_alloc(h, &h->idx, n));
  return MCL_OK;
}
// inclusive u32 scan of `in` into `out` (n local); uses tile32 as scratch
int scan_u32(mcl_handle* h, const u32* in, u32* out) {
  k_u32_tile_sums<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(in, h->n, h->tile32);
  k_scan_tile_sums<u32><<<1, 1024, 0, h->stream>>>(h->tile32, h->ntiles_loc, h->tile32 + h->ntiles_glob);
  k_u32_scan<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(in, h->n, h->tile32, out);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
// residual: normalise like auv_pf.py:172 (numpy's summation order), copies = floor(N w), k = sum
int residual_prepare(mcl_handle* h) {
  if (h->residual_k >= 0) return MCL_OK;
  RET_IF(alt_alloc(h));
  RET_IF(phase_local_max(h));
  const long long nchunks = (h->n + 8191) / 8192;
  t_begin(h, MCL_K_NORMALISE);
  k_linear_weights<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->lw, h->n, h->scal, h->weight_mode, h->wnorm);
  if (h->weight_mode == MCL_WEIGHT_LINEAR) {
    // free-function form (resampling.py): the caller's weights are used as they are, no renormalisation
    const double one = 1.0;
    HIPCHK(h, hipMemcpyAsync(h->scal + 1, &one, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  } else {
    k_np_chunk_sums<<<(unsigned)((nchunks + 63) / 64), 64, 0, h->stream>>>(h->wnorm, h->n, h->chunk);
    k_np_sum_final<<<1, 64, 0, h->stream>>>(h->chunk, nchunks, h->scal + 1);
  }
  k_residual_copies<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->wnorm, h->n, h->scal + 1, h->copies);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  RET_IF(scan_u32(h, h->copies, h->ccum));
  u32 k = 0;
  HIPCHK(h, hipMemcpyAsync(&k, h->ccum + (h->n - 1), sizeof(u32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->residual_k = k > (u32)h->n ? h->n : (long long)k;
  return MCL_OK;
}
long long uniforms_needed(mcl_handle* h, int* rc) {
  *rc = MCL_OK;
  switch (h->cfg.resample_scheme) {
    case MCL_RESAMPLE_SYSTEMATIC:
    case MCL_RESAMPLE_NAIVE: return 1;
    case MCL_RESAMPLE_STRATIFIED:
    case MCL_RESAMPLE_MULTINOMIAL: return h->n;
    case MCL_RESAMPLE_RESIDUAL:
      *rc = residual_prepare(h);
      return *rc == MCL_OK ? h->n - h->residual_k : 0;
  }
  *rc = MCL_ERR_INVALID;
  return 0;
}
int make_uniforms(mcl_handle* h, const double* uniforms, long long nu, long long need) {
  if (need <= 0) return MCL_OK;
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!uniforms || nu < need) return fail(h, MCL_ERR_INVALID, "resample: not enough replay uniforms for this scheme");
    RET_IF(upload(h, h->uni_dev, uniforms, sizeof(double) * (size_t)need));
    rp = h->uni_dev;
  }
  k_make_u53<<<grid_for(need), MCL_BLOCK, 0, h->stream>>>(rp, need, (u32)h->cfg.seed, (u32)(h->cfg.seed >> 32),
                                                        h->step_resample, h->u53);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int alt_indices(mcl_handle* h, const double* uniforms, long long nu) {
  RET_IF(set_device(h));
  RET_IF(alt_alloc(h));
  const int scheme = h->cfg.resample_scheme;
  if (scheme == MCL_RESAMPLE_RESIDUAL) {
    RET_IF(residual_prepare(h));
    const long long k = h->residual_k, need = h->n - k;
    RET_IF(make_uniforms(h, uniforms, nu, need));
    t_begin(h, MCL_K_SCAN);
    if (k > 0) k_residual_head<<<grid_for(k), MCL_BLOCK, 0, h->stream>>>(h->ccum, h->n, k, h->idx);
    if (need > 0) {
      k_residual_cumsum<<<1, 64, 0, h->stream>>>(h->wnorm, h->copies, h->n, h->cs);
      k_residual_searchsorted<<<1, 64, 0, h->stream>>>(h->cs, h->n, h->u53, need, h->idx + k);
    }
    t_end(h);
  } else {
    RET_IF(phase_quantise(h, true));
    RET_IF(make_uniforms(h, uniforms, nu, h->n));
    t_begin(h, MCL_K_SCAN);
    k_u64_scan<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->tile64, h->cq);
    if (scheme == MCL_RESAMPLE_STRATIFIED)
      k_stratified_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->cq, h->u53, h->n, h->idx);
    else
      k_multinomial_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->cq, h->u53, h->n, h->idx);
    t_end(h);
  }
  HIPCHK(h, hipGetLastError());
  h->idx_explicit = true;
  h->have_cdf = false;
  return MCL_OK;
}
int run_resample_alt(mcl_handle* h, const double* uniforms, long long nu, const double* replay_normals) {
  h->uni_valid = false;  // (single shard only: no exchange; the new state carries resampling noise)
  RET_IF(alt_indices(h, uniforms, nu));
  // keep/lost/dupes for an arbitrary ancestor vector (auv_pf.py:183-198) + noise
  if (replay_normals) RET_IF(upload_replay(h, replay_normals));
  t_begin(h, MCL_K_RESAMPLE);
  HIPCHK(h, hipMemsetAsync(h->cnt, 0, sizeof(u32) * (size_t)h->n, h->stream));
  HIPCHK(h, hipMemsetAsync(h->first, 0xff, sizeof(u32) * (size_t)h->n, h->stream));
  k_idx_hist<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->n, h->cnt, h->first);
  k_flags<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->cnt, h->first, h->n, 0, h->flags);
  RET_IF(scan_u32(h, h->flags, h->zcum));
  k_flags<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->cnt, h->first, h->n, 1, h->flags);
  RET_IF(scan_u32(h, h->flags, h->fcum));
  k_compact_dupes<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->flags, h->fcum, h->n, h->dupes);
  ReassignIdxArgs a;
  a.src = state_ptrs(h->state[h->cur], h->n);
  a.dst = state_ptrs(h->state[h->cur ^ 1], h->n);
  a.n = h->n;
  a.nz = noise_args(h, h->cfg.resample_cov, 2u, h->step_resample);
  k_reassign_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(a, h->cnt, h->zcum, h->dupes,
                                                             replay_normals ? h->replay_dev : nullptr);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cur ^= 1;
  h->step_resample++;
  h->have_cdf = false;
  h->idx_explicit = true;
  h->have_lw = false;
  h->residual_k = -1;
  return MCL_OK;
}

int run_resample(mcl_handle** sh, int ns, const double* uniforms, long long nu,
                 const double* const* replay_normals, bool with_moments = false) {
  mcl_handle* h0 = sh[0];
  for (int s = 0; s < ns; ++s) {
    if (!sh[s]->have_lw) return fail(sh[s], MCL_ERR_STATE, "resample: no weights (call an update first)");
    if (sh[s]->cfg.resample_scheme != MCL_RESAMPLE_SYSTEMATIC && sh[s]->cfg.resample_scheme != MCL_RESAMPLE_NAIVE) {
      if (ns > 1 || sh[s]->world > 1)
        return fail(sh[s], MCL_ERR_UNSUPPORTED, "resample: only the systematic scheme is sharded across GPUs");
      return run_resample_alt(sh[s], uniforms, nu, replay_normals ? replay_normals[0] : nullptr);
    }
  }
  uint64_t u53;
  if (h0->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!uniforms || nu < 1) return fail(h0, MCL_ERR_INVALID, "resample: REPLAY mode needs 1 uniform");
    if (!(uniforms[0] >= 0.0 && uniforms[0] < 1.0)) return fail(h0, MCL_ERR_INVALID, "resample: u not in [0,1)");
    u53 = (uint64_t)std::floor(uniforms[0] * 9007199254740992.0);
  } else {
    u53 = native_u53(h0->cfg.seed, h0->step_resample);
  }
  if (h0->cfg.resample_scheme == MCL_RESAMPLE_NAIVE) u53 |= MCL_U53_NAIVE;  // ">=" at the CDF edges (mcl_device.h)
  // one shard (and few enough tiles for every k_cdf_expand block to add their sums up itself):
  // max from the slots -> quantise -> CDF + expansion -> gather, three launches
  const bool single = ns == 1 && h0->world == 1 && !h0->comm && h0->ntiles_loc <= 8192;
  if (single) {
    h0->cdf_global = true;
    h0->group.clear();
    RET_IF(phase_quantise(h0, true, true));
    RET_IF(phase_expand(h0, true, u53));
    return phase_gather(h0, (replay_normals && h0->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[0] : nullptr,
                        with_moments);
  }
  if (ns == 1) {
    // one process per GPU: the 64 max-lw slots the update kernel filled are all-reduced as they are (ordered u64
    // keys: the maximum of the keys is the key of the maximum) and the quantise kernel reads them -- no k_max_finish
    h0->group.clear();
    RET_IF(set_device(h0));
    t_begin(h0, MCL_K_NORMALISE);
    RET_IF(ensure_max_slots(h0));
    t_end(h0);
    if (h0->comm && h0->world > 1) {
      t_begin(h0, MCL_K_COMM);
      NCCLCHK(h0, ncclAllReduce(ctrl_slots(h0), ctrl_slots(h0), MCL_MAX_SLOTS, ncclUint64, ncclMax, h0->comm, h0->stream));
      t_end(h0);
    }
    RET_IF(phase_quantise(h0, true));
  } else {
    for (int s = 0; s < ns; ++s) {
      sh[s]->group.assign(sh, sh + ns);
      RET_IF(phase_local_max(sh[s]));
    }
    RET_IF(exchange_max(sh, ns));
    for (int s = 0; s < ns; ++s) RET_IF(phase_quantise(sh[s], false));
  }
  RET_IF(exchange_totals(sh, ns));
  if (!h0->exch_allgather) {
    // O(n) per rank: every shard expands its OWN slice, the shards exchange two integers each, and only
    // the surplus copies whose global positions fall into a peer's lost ranks cross a link
    for (int s = 0; s < ns; ++s) RET_IF(phase_expand_local(sh[s], u53));
    RET_IF(exchange_ls(sh, ns));
    for (int s = 0; s < ns; ++s) RET_IF(phase_pack(sh[s], ns == 1 && sh[s]->lsx_host_dev != nullptr));
    RET_IF(exchange_dupes(sh, ns));
    for (int s = 0; s < ns; ++s)
      RET_IF(phase_gather(sh[s], (replay_normals && sh[s]->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[s] : nullptr,
                          with_moments));
    return MCL_OK;
  }
  for (int s = 0; s < ns; ++s) RET_IF(phase_cdf(sh[s], u53));
  RET_IF(exchange_cdf_state(sh, ns));
  for (int s = 0; s < ns; ++s) sh[s]->cdf_global = true;
  for (int s = 0; s < ns; ++s) RET_IF(phase_expand(sh[s], false, 0));
  for (int s = 0; s < ns; ++s)
    RET_IF(phase_gather(sh[s], (replay_normals && sh[s]->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[s] : nullptr,
                        with_moments));
  return MCL_OK;
}

// ------------------------------------------------------------------------------------------ mean/cov
int phase_mean_partial(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_MEAN_COV);
  const int g = grid_for(h->n);
  k_mean_partial<<<g, MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->part);
  k_sum_final<<<7, MCL_BLOCK, 0, h->stream>>>(h->part, g, h->scal + 8);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int phase_cov_partial(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_MEAN_COV);
  const int g = grid_for(h->n);
  k_cov_partial<<<g, MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->scal + 8,
                                                1.0 / (double)h->ng, h->part);
  k_sum_final<<<6, MCL_BLOCK, 0, h->stream>>>(h->part, g, h->scal + 16);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int exchange_sums(mcl_handle** sh, int ns, int off, int cnt) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      NCCLCHK(h, ncclAllReduce(h->scal + off, h->scal + off, cnt, ncclDouble, ncclSum, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  std::vector<double> acc(cnt, 0.0), tmp(cnt);
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(tmp.data(), sh[s]->scal + off, sizeof(double) * cnt, hipMemcpyDeviceToHost,
                                 sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    for (int k = 0; k < cnt; ++k) acc[k] += tmp[k];
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->scal + off, acc.data(), sizeof(double) * cnt, hipMemcpyHostToDevice,
                                 sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}
int run_mean_cov_async(mcl_handle** sh, int ns) {
  for (int s = 0; s < ns; ++s) RET_IF(phase_mean_partial(sh[s]));
  RET_IF(exchange_sums(sh, ns, 8, 7));
  for (int s = 0; s < ns; ++s) RET_IF(phase_cov_partial(sh[s]));
  RET_IF(exchange_sums(sh, ns, 16, 6));
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    RET_IF(set_device(h));
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 0.0;  // format 0: [0..6] sums of the 6 components + wrapped yaw, [8..13] centred second-moment sums
    HIPCHK(h, hipMemcpyAsync(slot, h->scal + 8, sizeof(double) * 14, hipMemcpyDeviceToHost, h->stream));
    h->mean_count++;
    h->have_meancov = true;
  }
  return MCL_OK;
}
// the sums k_resample_gather<true> left in scal[32..47]: reduce over the shards, queue the copy to the ring
int collect_fused_moments(mcl_handle** sh, int ns) {
  if (ns == 1 && sh[0]->moments_direct) {
    sh[0]->mean_count++;
    sh[0]->have_meancov = true;
    return MCL_OK;
  }
  RET_IF(exchange_sums(sh, ns, 32, MOM_COUNT));
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    RET_IF(set_device(h));
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 1.0;  // format 1: 13 sums about the shift in [13..15] (mcl_resample.h, k_resample_gather)
    t_begin(h, MCL_K_MEAN_COV);
    HIPCHK(h, hipMemcpyAsync(slot, h->scal + 32, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
    t_end(h);
    h->mean_count++;
    h->have_meancov = true;
  }
  return MCL_OK;
}
// ring entry -> mean pose, arithmetic mean of the wrapped yaw, covariance as auv_pf.py:238-252 lays it out
void finish_mean_cov(const mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9], long long which = -1) {
  const double N = (double)h->ng;
  if (which < 0) which = h->mean_count - 1;
  const double* p = h->host_pin + RING_STRIDE * (which % MEAN_RING);
  double c[6];
  if (p[16] == 0.0) {
    for (int k = 0; k < 6; ++k) mean6[k] = p[k] / N;
    for (int k = 0; k < 6; ++k) c[k] = p[8 + k] / N;
  } else {
    // d = x - shift:  mean = shift + sum(d)/N ;  cov_ab = sum(d_a d_b)/N - (sum d_a / N)(sum d_b / N)
    const double m0 = p[0] / N, m1 = p[1] / N, m2 = p[2] / N;
    mean6[0] = p[13] + m0;
    mean6[1] = p[14] + m1;
    mean6[2] = p[15] + m2;
    for (int k = 3; k < 6; ++k) mean6[k] = p[k] / N;
    c[0] = p[7] / N - m0 * m0;
    c[1] = p[8] / N - m1 * m1;
    c[2] = p[9] / N - m2 * m2;
    c[3] = p[10] / N - m0 * m1;
    c[4] = p[11] / N - m0 * m2;
    c[5] = p[12] / N - m1 * m2;
  }
  if (yaw_mean) *yaw_mean = p[6] / N;
  cov9[0] = c[0];
  cov9[1] = c[3];
  cov9[2] = c[4];
  cov9[3] = c[3];  // only [1,0] mirrored (auv_pf.py:246)
  cov9[4] = c[1];
  cov9[5] = c[5];
  cov9[6] = 0.0;
  cov9[7] = 0.0;
  cov9[8] = c[2];
}

// ------------------------------------------------------------------------------------------ MBES
int upload_beams(mcl_handle* h, const float* ranges, const float* beam_angles, int B) {
  if (B > h->beams_cap) {
    if (h->beam_sc) (void)hipFree(h->beam_sc);
    if (h->ranges_dev) (void)hipFree(h->ranges_dev);
    HIPCHK(h, hipMalloc(&h->beam_sc, sizeof(float2) * (size_t)B));
    HIPCHK(h, hipMalloc(&h->ranges_dev, sizeof(float) * (size_t)B));
    h->beams_cap = B;
    h->beam_cache.clear();
  }
  if ((int)h->beam_cache.size() != B || memcmp(h->beam_cache.data(), beam_angles, sizeof(float) * B) != 0) {
    std::vector<float2> sc(B);
    for (int b = 0; b < B; ++b) {
      sc[b].x = (float)std::sin((double)beam_angles[b]);
      sc[b].y = (float)std::cos((double)beam_angles[b]);
    }
    RET_IF(upload(h, h->beam_sc, sc.data(), sizeof(float2) * (size_t)B));
    h->beam_cache.assign(beam_angles, beam_angles + B);
    int lo = 0, hi = 0;
    bool finite = true;
    for (int b = 0; b < B; ++b) {
      if (!(beam_angles[b] == beam_angles[b])) finite = false;
      if (beam_angles[b] < beam_angles[lo]) lo = b;
      if (beam_angles[b] > beam_angles[hi]) hi = b;
    }
    const bool ok = finite && (double)beam_angles[hi] - (double)beam_angles[lo] < 3.0;  // span < pi
    h->beam_lo = ok ? lo : -1;
    h->beam_hi = ok ? hi : -1;
    bool asc = finite;
    for (int b = 1; b < B && asc; ++b) asc = beam_angles[b] >= beam_angles[b - 1];
    h->beams_sorted = asc;
    // the fan sweep walks outward from the nadir on either side: ascending angles, all within 85 degrees of it
    h->sweep_angles_ok = asc && B <= 2048 && beam_angles[0] >= -1.4835f && beam_angles[B - 1] <= 1.4835f;
    h->b_split = 0;
    while (h->b_split < B && beam_angles[h->b_split] < 0.f) ++h->b_split;
  }
  // (the ranges travel with the first launch_mbes of the update: in one copy with the sweep's beam table, or alone)
  if (ranges) {
    h->ranges_host.assign(ranges, ranges + B);
    h->ranges_pending = true;
  } else {
    h->ranges_host.clear();
    h->ranges_pending = false;
  }
  return MCL_OK;
}

// Lanes per particle side of the fan sweep: a small cloud splits a side's beams over 2 or 4 lanes (mcl_sweep.h SUB:
// each resolves its own run of >= 16 beams, starting at the hit of the run's first beam); the GLOBAL particle count
// decides, so every shard sums in the same order.
// (measured, round 3 with the grid's cell walk, the whole fused step in ms -- traversal | sweep with 1 / 2 / 4 lanes per
//  side, 256 beams:
//    grid   4 096: 0.086 | 0.095 0.090 0.083     mesh   4 096: 0.079 | 0.097 0.092 0.075
//    grid   8 192: 0.089 | 0.093 0.085 0.081     mesh   8 192: 0.089 | 0.094 0.090 0.082
//    grid  32 768: 0.158 | 0.094 0.089 0.087     mesh  32 768: 0.157 | 0.095 0.093 0.088
//    grid  65 536: 0.154 | 0.100 0.098 0.101     mesh  65 536: 0.153 | 0.099 0.100 0.098
//    grid 131 072: 0.218 | 0.120 0.125 0.133     mesh 131 072: 0.218 | 0.116 0.122 0.128
//  below 4 096 the traversal wins (128 particles: 0.059 against 0.067); a later run pays one slanted traversal for its
//  start; worth it while the chip is not full)
int sweep_lanes_per_side(const mcl_handle* h, bool with_ranges, int B) {
  int nsub = 1;
  if (with_ranges) {
    if (h->map_kind == 0)
      nsub = h->ng < 49152 ? 4 : (h->ng < 98304 ? 2 : 1);
    else if (h->mesh && h->mesh->heights && !h->force_general_mesh)
      nsub = h->ng < 49152 ? 4 : 1;
    if (h->env_nsub) nsub = h->env_nsub;
    while (nsub > 1 && B / (2 * nsub) < 16) nsub >>= 1;
  }
  return nsub;
}

// beam table of the fan sweep: side-signed tangent, secant, measured range, weight; and per beam the sum of the
// squared normalised residuals against r_max over the beams from it to the end of its side (mcl_sweep.h)
int upload_sweep_beams(mcl_handle* h, bool with_ranges, int B, double sigma, double r_max, int nsub) {
  // one device block, one copy per update: B records | B tail sums | B measured ranges (for the traversal kernels
  // that take the hand-overs)
  const size_t blk_floats = (size_t)B * 7 + 4;   // B records | B tail sums | B measured ranges | first / second tangent of either side | B tail sums per run
  if (B > h->sweep_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    for (int k = 0; k < 2; ++k) {
      if (h->sweep_buf[k]) (void)hipFree(h->sweep_buf[k]);
      if (h->sweep_stage[k]) (void)hipHostFree(h->sweep_stage[k]);
      h->sweep_buf[k] = nullptr;
      h->sweep_stage[k] = nullptr;
      HIPCHK(h, hipMalloc(&h->sweep_buf[k], sizeof(float) * blk_floats));
      HIPCHK(h, hipHostMalloc(&h->sweep_stage[k], sizeof(float) * blk_floats, hipHostMallocDefault));
      if (!h->ev_stage[k]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_stage[k], hipEventDisableTiming));
      h->stage_used[k] = false;
    }
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    h->sweep_cap = B;
  }
  const int sel = (h->sweep_sel ^= 1);
  h->sweep_beams = h->sweep_buf[sel];
  h->sweep_tail = (float*)(h->sweep_beams + B);
  if (h->stage_used[sel]) HIPCHK(h, hipEventSynchronize(h->ev_stage[sel]));   // (two updates old: long complete)
  struct Blk {   // (the table is built straight into the pinned staging buffer)
    float* p;
    float* data() { return p; }
    float& operator[](size_t k) { return p[k]; }
  } blk{h->sweep_stage[sel]};
  float4* tb = (float4*)blk.data();
  float* tail = blk.data() + (size_t)B * 4;
  float* rng = tail + B;
  for (int b = 0; b < B; ++b) tail[b] = 0.f;
  const float rmaxf = (float)r_max;
  int nvalid = 0;
  for (int b = 0; b < B; ++b) {
    const double ang = (double)h->beam_cache[b];
    const float rm = (with_ranges && (int)h->ranges_host.size() == B) ? h->ranges_host[b] : 0.f;
    const bool valid = rm > 0.f;  // NaN fails the test (as in the cast kernels)
    nvalid += valid ? 1 : 0;
    // the residual's constants: (range_b - r) w with r = min(t / cos a, r_max) is max(z w - t (w / cos a), (z - r_max) w);
    // an invalid beam has all three zero.  Expected-range calls (no measured ranges) keep 1 / cos a in .y
    const double sec = 1.0 / std::cos(ang);
    // .x: the side-signed tangent of the beam SWEEP_TAN_AHEAD places further out on this beam's side (mcl_sweep.h)
    {
      const int nb = b < h->b_split ? b - SWEEP_TAN_AHEAD : b + SWEEP_TAN_AHEAD;
      tb[b].x = (nb < 0 || nb >= B) ? INFINITY : (float)(std::tan((double)h->beam_cache[nb]) * (b < h->b_split ? -1.0 : 1.0));
    }
    tb[b].y = with_ranges ? (valid ? (float)(sec / sigma) : 0.f) : (float)sec;
    tb[b].z = valid ? (float)((double)rm / sigma) : 0.f;
    tb[b].w = valid ? (float)(((double)rm - (double)rmaxf) / sigma) : 0.f;
    rng[b] = rm;
  }
  auto miss = [&](int b) { return tb[b].w * tb[b].w; };
  float run = 0.f;
  for (int b = B - 1; b >= h->b_split; --b) tail[b] = (run += miss(b));
  run = 0.f;
  for (int b = 0; b < h->b_split; ++b) tail[b] = (run += miss(b));
  // the same sums per RUN of a side's beams (sub-fans: a lane accounts for its own run only; taking them as
  // differences of the side's sums lost digits when r_max is short and the sums are large)
  float* tail_run = blk.data() + (size_t)B * 6 + 4;
  for (int side = 0; side < 2; ++side) {
    const int nb = side ? h->b_split : B - h->b_split;
    const int per = nsub > 1 ? std::max((nb + nsub - 1) / nsub, 2) : std::max(nb, 1);
    for (int first = 0; first < nb; first += per) {
      const int last = std::min(first + per, nb);
      float acc = 0.f;
      for (int k = last - 1; k >= first; --k) {   // k-th beam of the side, counted outward from the nadir
        const int b = side ? h->b_split - 1 - k : h->b_split + k;
        tail_run[b] = (acc += miss(b));
      }
    }
  }
  h->sweep_nvalid = nvalid;
  for (int k = 0; k < 2; ++k) {
    const int bp = h->b_split + k, bm = h->b_split - 1 - k;
    blk[(size_t)B * 6 + 2 * k] = bp < B ? (float)std::tan((double)h->beam_cache[bp]) : INFINITY;
    blk[(size_t)B * 6 + 2 * k + 1] = bm >= 0 ? (float)(-std::tan((double)h->beam_cache[bm])) : INFINITY;
  }
  // the device buffer was last read by the update two before this one; the copy waits for the event of the update
  // just before (later on the same stream, so certainly enough -- whatever an error path did to the alternation): it
  // then runs under that step's normalise / scan / gather kernels.  The compute stream waits for the copy.
  if (h->ev_upd[0] && h->upd_seq >= 1) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_upd[(h->upd_seq - 1) & 3], 0));
  HIPCHK(h, hipMemcpyAsync(h->sweep_beams, blk.data(), sizeof(float) * blk_floats, hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(h, hipEventRecord(h->ev_stage[sel], h->copy_stream));
  h->stage_used[sel] = true;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_stage[sel], 0));
  h->ranges_ptr = h->sweep_tail + B;
  h->ranges_pending = false;
  return MCL_OK;
}

void rot_rpy(double roll, double pitch, double yaw, double R[9]) {
  double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
  double cy = std::cos(yaw), sy = std::sin(yaw);
  R[0] = cy * cp;
  R[1] = cy * sp * sr - sy * cr;
  R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp;
  R[4] = sy * sp * sr + cy * cr;
  R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;
  R[7] = cp * sr;
  R[8] = cp * cr;
}

// Morton visiting order of the particles' pose records (k_mbes_keys): h->mbes_perm
int sort_visiting_order(mcl_handle* h, const MbesArgs& a) {
  const size_t n = (size_t)h->n;
  if (!h->sort_keys) {
    HIPCHK(h, hipMalloc(&h->sort_keys, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->sort_keys_out, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->sort_idx, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->mbes_perm, sizeof(u32) * n));
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, h->sort_tmp_bytes, h->sort_keys, h->sort_keys_out, h->sort_idx,
                                        h->mbes_perm, n, 0, 24, h->stream));
    HIPCHK(h, hipMalloc(&h->sort_tmp, h->sort_tmp_bytes));
  }
  k_mbes_keys<<<grid_for(h->n), 256, 0, h->stream>>>(a, h->sort_keys, h->sort_idx);
  // stable LSD radix sort of (key, slot) pairs: the visiting order is deterministic
  HIPCHK(h, rocprim::radix_sort_pairs(h->sort_tmp, h->sort_tmp_bytes, h->sort_keys, h->sort_keys_out, h->sort_idx,
                                      h->mbes_perm, n, 0, 24, h->stream));
  return MCL_OK;
}

int launch_mbes(mcl_handle* h, bool with_ranges, int B, double sigma, double r_max, const double sensor_offset[6],
                double* lw_out, float* exp_out, long long exp_first, long long exp_count, bool pose_done = false,
                MbesArgs* args_only = nullptr) {
  if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, "update_mbes: no map (call mcl_set_map_grid/mesh first)");
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = sensor_offset ? sensor_offset : zero6;
  MbesArgs a;
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int k = 0; k < 12; ++k) a.m2o[k] = h->cfg.m2o[k];
  for (int k = 0; k < 3; ++k) a.off_t[k] = so[k];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.beam_sc = h->beam_sc;
  a.ranges = nullptr;  // (set below, once the ranges are on the device)
  a.n_beams = B;
  a.sorted = h->beams_sorted ? 1 : 0;
  a.b_lo = h->beam_lo;
  a.b_hi = h->beam_hi;
  a.inv_sigma = (float)(1.0 / sigma);
  a.r_max = (float)r_max;
  a.lognorm = std::log(sigma * std::sqrt(2.0 * MCL_PI));
  a.lw = lw_out;
  a.exp_out = exp_out;
  a.exp_first = exp_first;
  a.exp_count = exp_count;
  if (!h->pose_dev) HIPCHK(h, hipMalloc(&h->pose_dev, sizeof(MbesPose) * (size_t)h->n));
  a.pose = h->pose_dev;
  memset(&a.mesh, 0, sizeof a.mesh);
  a.stats = nullptr;
  a.cells = 0;
  a.perm = nullptr;
  a.diag_mode = 0;
  a.sweep_beams = nullptr;
  a.sweep_tail = nullptr;
  a.b_split = 0;
  a.sweep_nvalid = 0;
  a.sweep_nsub = 1;
  a.sweep_tan0 = nullptr;
  a.sweep_tail_run = nullptr;
  a.sweep_c2z_min = 2.f;
  a.sweep_slope = 0.f;
  a.defer_idx = nullptr;
  a.defer_count = (int*)(h->ctrl + CTRL_DEFER);
  a.n_dev = nullptr;
  a.host_count = nullptr;
#ifdef MBES_STATS
  {
    static unsigned long long* g_stats = nullptr;
    if (!g_stats) {
      hipMalloc(&g_stats, 32);
      hipMemset(g_stats, 0, 32);
    }
    unsigned long long hs[4];
    hipMemcpy(hs, g_stats, 32, hipMemcpyDeviceToHost);
    if (hs[2]) fprintf(stderr, "[mbes stats] rays %llu steps/ray %.2f tests/ray %.2f retries/ray %.4f\n", hs[2], (double)hs[0] / hs[2], (double)hs[1] / hs[2], (double)hs[3] / hs[2]);
    hipMemset(g_stats, 0, 32);
    a.stats = g_stats;
  }
#endif
  if (h->map_kind == 0) {
    a.grid = h->grid;
    a.nx = h->gnx;
    a.ny = h->gny;
    a.ox = h->gox;
    a.oy = h->goy;
    a.inv_res = 1.0 / h->gres;
    a.res = (float)h->gres;
    a.zmin_map = h->gzmin;
    a.zmax_map = h->gzmax;
  } else {
    const MeshDev* m = h->mesh;
    a.mesh = mesh_args(m);
    a.grid = m->heights;  // non-null: structured mesh (triangulated regular height grid)
    a.nx = m->gx + 1;
    a.ny = m->gy + 1;
    a.ox = m->x0;
    a.oy = m->y0;
    a.inv_res = 1.0 / m->cs;
    a.res = (float)m->cs;
    a.zmin_map = m->zmin;
    a.zmax_map = m->zmax;
    a.diag_mode = m->diag_mode;
    a.cells = (m->heights && !h->force_general_mesh) ? 0 : 1;
  }
  const long long ngroups = (h->n + MBES_WAVES - 1) / MBES_WAVES;
  const int grid = (int)(ngroups < 4096 ? ngroups : 4096);
  if (!h->mbes_worklist) HIPCHK(h, hipMalloc(&h->mbes_worklist, sizeof(int) * (size_t)(ngroups + 1)));
  if (!h->mbes_groups) HIPCHK(h, hipMalloc(&h->mbes_groups, sizeof(MbesGroup) * (size_t)ngroups));
  a.worklist = h->mbes_worklist;
  a.groups = h->mbes_groups;
  a.work_count = (int*)(h->ctrl + CTRL_WORK);
  // the cast kernels leave max lw in the control block's slots: the normalisation needs no reduction pass
  a.max_slots = (with_ranges && lw_out == h->lw) ? ctrl_slots(h) : nullptr;
  // height grids and structured meshes: the pose kernel classifies the groups, k_mbes_fast casts the
  // eligible ones, k_mbes_cast<.,.,1> the worklist; triangle-record meshes keep the two-mode kernel
  const bool lean = true;  // every map kind: the pose kernel classifies the groups
  const bool structured = h->map_kind == 1 && h->mesh->heights && !h->force_general_mesh;
  // ---- fan sweep (mcl_sweep.h): regularly triangulated meshes, ascending beam angles.  The fan plane may lean
  // from the vertical only as far as the steepest triangle allows (tan(tilt) * slope < 1, with a margin).
  if (!pose_done) {
    // (below ~8 k particles even four lanes per side cannot fill the chip and the wave-per-particle traversal is
    //  faster -- measured at 128 ... 262 144 particles x 256 / 512 beams; MCL_SWEEP=1 forces it)
    const bool lattice = h->map_kind == 0 || (structured && (a.diag_mode == 1 || a.diag_mode == 2));
    const long long sweep_min_n = h->env_sweep == 1 ? 1 : (lattice ? 8192 : 16384);   // (adjacency sweep: one lane per side only)
    // a height-field TIN with adjacency -- also a triangulated height grid whose cells are split along mixed diagonals
    // (tin_ok: mesh_build has PROVEN the mesh single-valued over (x, y) -- adjacency, fold and pairwise overlap tests)
    const bool tin = h->map_kind == 1 && h->mesh->tin_ok && !h->mesh_no_sweep && (!structured || a.diag_mode == 0);
    bool sweep = ((structured && (a.diag_mode == 1 || a.diag_mode == 2)) || h->map_kind == 0 || tin) && h->sweep_angles_ok && h->env_sweep != 0 &&
                 h->ng >= sweep_min_n &&  // (the GLOBAL count: every shard of a cloud takes the same path, results do not depend on the GPU count)
                 h->n < (1ll << 31) && (long long)a.nx * a.ny < (1ll << 30) && a.ny < (1 << 21);
    h->sweep_now = sweep;
    if (sweep) {
      RET_IF(upload_sweep_beams(h, with_ranges, B, sigma, r_max, sweep_lanes_per_side(h, with_ranges, B)));
    } else if (with_ranges && h->ranges_pending) {
      RET_IF(upload(h, h->ranges_dev, h->ranges_host.data(), sizeof(float) * (size_t)B));
      h->ranges_ptr = h->ranges_dev;
      h->ranges_pending = false;
    }
  }
  a.ranges = with_ranges ? h->ranges_ptr : nullptr;
  const bool sweep = h->sweep_now;
  if (sweep) {
    if (!h->defer_idx) HIPCHK(h, hipMalloc(&h->defer_idx, sizeof(u32) * (size_t)h->n));
    if (!h->defer_idx2) HIPCHK(h, hipMalloc(&h->defer_idx2, sizeof(u32) * (size_t)h->n));
    a.sweep_beams = h->sweep_beams;
    a.sweep_tail = h->sweep_tail;
    a.b_split = h->b_split;
    a.sweep_nvalid = h->sweep_nvalid;
    a.sweep_tan0 = h->sweep_tail + 2 * (size_t)B;
    a.sweep_tail_run = h->sweep_tail + 2 * (size_t)B + 4;
    // (grids: 0.45 -- below 0.5 the plane function cannot change sign around a cell's four corners, mcl_sweep.h)
    const double slope_max = h->map_kind == 0 ? h->gslope_max : h->mesh->slope_max;
    const double tan_lim = std::min(std::tan(35.0 * MCL_PI / 180.0), (h->map_kind == 0 ? 0.45 : 0.8) / std::max(slope_max, 1e-9));
    a.sweep_c2z_min = (float)(1.0 / std::sqrt(1.0 + tan_lim * tan_lim));
    a.sweep_slope = (float)slope_max;
    a.defer_idx = h->defer_idx;
  }
  if (args_only) {
    *args_only = a;
    return MCL_OK;
  }
  // ---- counters of the update two before this one (deterministic lag, see mcl_handle::work_host)
  if (!h->work_host) {
    HIPCHK(h, hipHostMalloc(&h->work_host, 64, hipHostMallocDefault));
    memset(h->work_host, 0, 64);
    for (int k = 0; k < 4; ++k) HIPCHK(h, hipEventCreateWithFlags(&h->ev_upd[k], hipEventDisableTiming));
  }
  static const int wh_zero[4] = {0, 0, 0, 0};
  const int* wh_prev = wh_zero;
  if (h->upd_seq >= 2) {
    HIPCHK(h, hipEventSynchronize(h->ev_upd[(h->upd_seq - 2) & 3]));
    wh_prev = h->work_host + 4 * ((h->upd_seq - 2) & 3);
  }
  int* wh_cur = h->work_host + 4 * (h->upd_seq & 3);  // (its last user, four updates ago, finished before the event above)
  wh_cur[0] = wh_cur[1] = wh_cur[2] = wh_cur[3] = 0;
  struct SeqGuard {  // whatever path returns: this update's kernels are behind its event
    mcl_handle* h;
    ~SeqGuard() {
      (void)hipEventRecord(h->ev_upd[h->upd_seq & 3], h->stream);
      h->upd_seq++;
    }
  } seq_guard{h};
  t_begin(h, MCL_K_UPDATE_MBES);
  if (!pose_done) {
    // (the fused predict has already reset the control block and written poses, group records and worklist)
    if (a.max_slots)
      HIPCHK(h, hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));  // slots + work and hand-over counters (one aligned fill)
    else
      HIPCHK(h, hipMemsetAsync(a.work_count, 0, 3 * sizeof(int), h->stream));
    if (lean && !sweep)
      k_mbes_pose<true><<<grid_for(h->n), 256, 0, h->stream>>>(a);
    else
      k_mbes_pose<false><<<grid_for(h->n), 256, 0, h->stream>>>(a);
  }
  if (a.max_slots) h->max_valid = true;
  a.perm = nullptr;
  if (sweep) {
    // The hand-over count of the sweep two updates ago (see work_host).  When it was large (a cloud on the map
    // border, a fan too tilted for the terrain) the particles are visited in Morton order: the hand-over list
    // inherits it wave by wave, so the groups of eight the cast kernels form from it share tiles.
    const bool sort_now = h->env_sort >= 0 ? h->env_sort == 1 : (long long)wh_prev[1] * 16 > h->n;
    if (sort_now && h->n > MBES_WAVES) {
      RET_IF(sort_visiting_order(h, a));
      a.perm = h->mbes_perm;
    }
    const int nsub = sweep_lanes_per_side(h, with_ranges, B);
    a.sweep_nsub = nsub;
    const int sthreads = nsub == 4 ? 512 : SWEEP_THREADS;
    const int per_block = sthreads / 64 / (2 * nsub) * 64;
    // (expected ranges of a few particles: only their lanes are launched)
    const long long n_part = (!with_ranges && !a.perm) ? std::max<long long>(std::min<long long>(exp_count, h->n - exp_first), 1) : h->n;
    const int sgrid = (int)((n_part + per_block - 1) / per_block);
    const size_t lds = (size_t)(B + 2) * sizeof(float4) + (size_t)(B + 4) * sizeof(float);
    // What the first pass declines goes through a second, bounds-checked pass (lattice maps: a slice that leaves the
    // map ends there), and what that one declines is cast the old way, in the order of its hand-over list: group
    // records and worklist (k_mbes_classify), the fast kernel, the general kernel.  All of them read the length of
    // their list on the device.
    const bool lattice = h->map_kind == 0 || (structured && a.diag_mode != 0);
    h->sweep_two_pass = lattice;
    MbesArgs c = a;   // second pass (always one lane per side: it sees few particles)
    c.sweep_nsub = 1;
    c.perm = h->defer_idx;
    c.n_dev = a.defer_count;
    c.defer_idx = h->defer_idx2;
    c.defer_count = (int*)(h->ctrl + CTRL_DEFER2);
    c.host_count = wh_cur + 2;
    MbesArgs d = a;   // traversal kernels
    d.perm = lattice ? h->defer_idx2 : h->defer_idx;
    d.n_dev = lattice ? c.defer_count : a.defer_count;
    d.host_count = wh_cur + 1;  // (pinned: the classify kernel stores the count there, no copy on the stream)
    // (their loops are grid-stride: the grids only set the parallelism.  After an update that handed nothing over
    //  they are launched small -- three empty 2048-workgroup launches cost 15 us, 2.5 % of the update)
    const bool few = wh_prev[1] == 0, few2 = wh_prev[2] == 0;
    const int cgrid = (int)std::min<long long>(grid_for(h->n), few ? 32 : 1024);
    const int fgrid = (int)std::min<long long>(ngroups, few ? 64 : 2048);
    const int dgrid = (int)std::min<long long>(ngroups, few ? 64 : 512);
    const int s2grid = (int)std::min<long long>((h->n + SWEEP_THREADS / 2 - 1) / (SWEEP_THREADS / 2), few2 ? 32 : 4096);
#define LAUNCH_SWEEP(SURFV, MAPV)                                                        \
  do {                                                                                   \
    if (with_ranges) {                                                                   \
      t_begin(h, MCL_K_MBES_MAIN);                                                       \
      if (nsub > 1)                                                                      \
        k_mbes_sweep<SURFV, false, false, true><<<sgrid, sthreads, lds, h->stream>>>(a); \
      else                                                                               \
        k_mbes_sweep<SURFV, false, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);  \
      t_end(h);                                                                          \
      k_mbes_sweep<SURFV, false, true><<<s2grid, SWEEP_THREADS, lds, h->stream>>>(c);    \
      k_mbes_classify<<<cgrid, 256, 0, h->stream>>>(d);                                  \
      k_mbes_fast<SURFV, false><<<fgrid, MBES_THREADS, 0, h->stream>>>(d);               \
      k_mbes_cast<MAPV, false, 1><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);             \
    } else {                                                                             \
      k_mbes_sweep<SURFV, true, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);     \
      k_mbes_sweep<SURFV, true, true><<<s2grid, SWEEP_THREADS, lds, h->stream>>>(c);     \
      k_mbes_classify<<<cgrid, 256, 0, h->stream>>>(d);                                  \
      k_mbes_fast<SURFV, true><<<fgrid, MBES_THREADS, 0, h->stream>>>(d);                \
      k_mbes_cast<MAPV, true, 1><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);              \
    }                                                                                    \
  } while (0)
#define LAUNCH_SWEEP_TIN(SURFV, MAPV)                                                    \
  do {                                                                                   \
    if (with_ranges) {                                                                   \
      t_begin(h, MCL_K_MBES_MAIN);                                                       \
      if (nsub > 1)                                                                      \
        k_mbes_sweep<5, false, false, true><<<sgrid, sthreads, lds, h->stream>>>(a);     \
      else                                                                               \
        k_mbes_sweep<5, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);             \
      t_end(h);                                                                          \
      k_mbes_classify<<<cgrid, 256, 0, h->stream>>>(d);                                  \
      k_mbes_fast<SURFV, false><<<fgrid, MBES_THREADS, 0, h->stream>>>(d);               \
      k_mbes_cast<MAPV, false, 1><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);             \
    } else {                                                                             \
      k_mbes_sweep<5, true><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);                \
      k_mbes_classify<<<cgrid, 256, 0, h->stream>>>(d);                                  \
      k_mbes_fast<SURFV, true><<<fgrid, MBES_THREADS, 0, h->stream>>>(d);                \
      k_mbes_cast<MAPV, true, 1><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);              \
    }                                                                                    \
  } while (0)
    if (h->map_kind == 0)
      LAUNCH_SWEEP(0, 0);
    else if (!structured)
      LAUNCH_SWEEP_TIN(4, 1);   // hand-overs: triangle records
    else if (a.diag_mode == 0)
      LAUNCH_SWEEP_TIN(1, 2);   // hand-overs: node heights with the per-cell diagonal bit
    else if (a.diag_mode == 1)
      LAUNCH_SWEEP(2, 2);
    else
      LAUNCH_SWEEP(3, 2);
#undef LAUNCH_SWEEP
#undef LAUNCH_SWEEP_TIN
    if (h->env_debug_work) {
      int cnt = 0;
      (void)hipMemcpyAsync(&cnt, a.defer_count, sizeof(int), hipMemcpyDeviceToHost, h->stream);
      (void)hipStreamSynchronize(h->stream);
      fprintf(stderr, "[mbes] sweep handed over %d of %lld particles\n", cnt, (long long)h->n);
    }
    t_end(h);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
  }
  if (lean) {
    // Dispersed cloud?  The natural-order classification has just counted the groups without a common tile.
    // That count travels to the host asynchronously and is read one call late (no synchronisation): when the
    // previous update deferred more than 1/16 of its groups, this one visits the particles in Morton order.
    const bool sort_now = h->env_sort >= 0 ? h->env_sort == 1 : (long long)wh_prev[0] * 16 > ngroups;
    HIPCHK(h, hipMemcpyAsync(wh_cur, a.work_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (sort_now && h->n > MBES_WAVES) {
      RET_IF(sort_visiting_order(h, a));
      a.perm = h->mbes_perm;
      HIPCHK(h, hipMemsetAsync(a.work_count, 0, sizeof(int), h->stream));
      k_mbes_classify<<<grid_for(h->n), 256, 0, h->stream>>>(a);
    }
  }
  const int ggrid = (int)(ngroups < 512 ? ngroups : 512);
#define LAUNCH_LEAN(SURFV, MAPV)                                                   \
  do {                                                                             \
    if (with_ranges) {                                                             \
      t_begin(h, MCL_K_MBES_MAIN);                                                 \
      k_mbes_fast<SURFV, false><<<grid, MBES_THREADS, 0, h->stream>>>(a);          \
      t_end(h);                                                                    \
      k_mbes_cast<MAPV, false, 1><<<ggrid, MBES_THREADS, 0, h->stream>>>(a);       \
    } else {                                                                       \
      k_mbes_fast<SURFV, true><<<grid, MBES_THREADS, 0, h->stream>>>(a);           \
      k_mbes_cast<MAPV, true, 1><<<ggrid, MBES_THREADS, 0, h->stream>>>(a);        \
    }                                                                              \
  } while (0)
  if (h->map_kind == 0) {
    LAUNCH_LEAN(0, 0);
  } else if (structured) {
    if (a.diag_mode == 1)
      LAUNCH_LEAN(2, 2);
    else if (a.diag_mode == 2)
      LAUNCH_LEAN(3, 2);
    else
      LAUNCH_LEAN(1, 2);
  } else {
    LAUNCH_LEAN(4, 1);  // triangle records: cell-word tiles
  }
#undef LAUNCH_LEAN
  if (h->env_debug_work) {  // diagnostics: how many groups the fast kernel deferred
    int cnt = 0;
    (void)hipMemcpyAsync(&cnt, a.work_count, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    (void)hipStreamSynchronize(h->stream);
    fprintf(stderr, "[mbes] deferred %d of %lld groups\n", cnt, ngroups);
    if (lean && cnt > 0) {
      std::vector<MbesGroup> g((size_t)ngroups);
      (void)hipMemcpy(g.data(), h->mbes_groups, sizeof(MbesGroup) * (size_t)ngroups, hipMemcpyDeviceToHost);
      long long why[32] = {0}, area = 0, na = 0;
      for (const MbesGroup& G : g)
        if (!G.fast) {
          why[G.why & 31]++;
          area += (long long)G.tw * G.th;
          ++na;
        }
      fprintf(stderr, "[mbes] why:");
      for (int k = 0; k < 32; ++k)
        if (why[k]) fprintf(stderr, " %d:%lld", k, why[k]);
      fprintf(stderr, "  mean window of deferred groups %lld nodes\n", na ? area / na : 0);
    }
  }
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// a fused step that leaves before its gather: store the z, roll, pitch its predict kernel did not
int materialise_uniform(mcl_handle* h) {
  if (!h->uni_deferred) return MCL_OK;
  h->uni_deferred = false;
  RET_IF(set_device(h));
  k_fill_uniform<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->uni_val[0],
                                                             h->uni_val[1], h->uni_val[2]);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// pose_for: when given (fused step, NATIVE rng) the kernel also writes the MBES pose records of the new
// state; *pose_written tells the caller whether it did (a dt <= 0 step leaves the state untouched)
int do_predict(mcl_handle* h, const mcl_odom* od, double dt, const double* replay_normals,
               const MbesArgs* pose_for = nullptr, bool* pose_written = nullptr, bool defer_uniform = false) {
  if (pose_written) *pose_written = false;
  if (!(dt > 0.0)) return MCL_OK;  // auv_pf.py:205 gate
  double rpy[3];
  euler_from_quat(od->q, rpy);
  const double roll = rpy[0], pitch = rpy[1];
  const double cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll), sr = std::sin(roll);
  // M1 = Ry' Rx with the reference's Ry' (auv_particle.py:90-92); rows 0,1 only
  const double M1r0[3] = {cp, sp * sr, sp * cr};
  const double M1r1[3] = {0.0, cr, -sr};
  const double vdt[3] = {od->v[0] * dt, od->v[1] * dt, od->v[2] * dt};
  PredictArgs a;
  a.m0 = M1r0[0] * vdt[0] + M1r0[1] * vdt[1] + M1r0[2] * vdt[2];
  a.m1 = M1r1[0] * vdt[0] + M1r1[1] * vdt[1] + M1r1[2] * vdt[2];
  a.wzdt = od->w_z * dt;
  a.z = od->z;
  a.roll = roll;
  a.pitch = pitch;
  a.nz = noise_args(h, h->cfg.process_cov, 1u, h->step_predict);
  a.zero_ptr = nullptr;
  a.zero_words = 0;
  a.skip_uniform = 0;
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (replay_normals) {
      RET_IF(upload_replay(h, replay_normals));
      rp = h->replay_dev;
    } else {
      for (int c = 0; c < 6; ++c) a.nz.sq[c] = 0.0;  // REPLAY without draws: noise-free
      rp = nullptr;
    }
  }
  t_begin(h, MCL_K_PREDICT);
  if (h->cfg.rng_mode == MCL_RNG_REPLAY && !rp) {
    // noise-free: feed zeros through the native branch with sq = 0
  }
  if (pose_for && !rp && h->cfg.rng_mode == MCL_RNG_NATIVE) {
    const bool lean = !pose_for->sweep_beams;  // the fan sweep needs no group records
    a.skip_uniform = defer_uniform ? 1 : 0;
    h->uni_deferred = defer_uniform;
    if (lean) {
      // reset the slots + work counter first: the kernel appends the deferred groups to the worklist
      // (the whole block: ONE aligned fill; the kernel tickets in it are zero between launches anyway)
      HIPCHK(h, hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));
    } else {
      a.zero_ptr = (unsigned long long*)h->ctrl;   // the kernel's first workgroup zeroes it: no memset launch
      a.zero_words = CTRL_BYTES / 8;
    }
    if (lean)
      k_predict_pose<true><<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, *pose_for);
    else
      k_predict_pose<false><<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, *pose_for);
    if (pose_written) *pose_written = true;
  } else {
    k_predict<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, rp);
  }
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->step_predict++;
  // every particle now holds the odometry's depth, roll and pitch (both kernels store these three constants)
  h->uni_valid = true;
  h->uni_val[0] = a.z;
  h->uni_val[1] = a.roll;
  h->uni_val[2] = a.pitch;
  return MCL_OK;
}

}  // namespace

// dead-reckoning integrator (host only; uses euler_from_quat above)
#include "mcl_dr_impl.h"
// bathymetry map builder (uses rot_rpy above)
#include "mcl_gridmap.h"

// ============================================================================================ C ABI
extern "C" {

int mcl_abi_version(void) { return MCL_ABI_VERSION; }

const char* mcl_status_string(int s) {
  switch (s) {
    case MCL_OK: return "ok";
    case MCL_ERR_INVALID: return "invalid argument";
    case MCL_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case MCL_ERR_HIP: return "HIP runtime error";
    case MCL_ERR_UNSUPPORTED: return "unsupported";
    case MCL_ERR_STATE: return "bad call order";
    case MCL_ERR_COMM: return "RCCL error";
    case MCL_ERR_ALLOC: return "allocation failed";
  }
  return "unknown status";
}

const char* mcl_last_error(const mcl_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int mcl_matrix_from_tf(const double translation[3], const double quaternion[4], double m16[16]) {
  if (!translation || !quaternion || !m16) return MCL_ERR_INVALID;
  // quaternion_matrix (tf.transformations): scale by sqrt(2/|q|^2), outer product
  const double* qi = quaternion;
  const double nq = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (nq >= 2.220446049250313e-16 * 4.0) {
    const double s = std::sqrt(2.0 / nq);
    const double q[4] = {qi[0] * s, qi[1] * s, qi[2] * s, qi[3] * s};
    double o[4][4];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) o[a][b] = q[a] * q[b];
    R[0] = 1.0 - o[1][1] - o[2][2];
    R[1] = o[0][1] - o[2][3];
    R[2] = o[0][2] + o[1][3];
    R[3] = o[0][1] + o[2][3];
    R[4] = 1.0 - o[0][0] - o[2][2];
    R[5] = o[1][2] - o[0][3];
    R[6] = o[0][2] - o[1][3];
    R[7] = o[1][2] + o[0][3];
    R[8] = 1.0 - o[0][0] - o[1][1];
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m16[r * 4 + c] = R[r * 3 + c];
    m16[r * 4 + 3] = translation[r];
  }
  m16[12] = m16[13] = m16[14] = 0.0;
  m16[15] = 1.0;
  return MCL_OK;
}

int mcl_device_count(int* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  if (count) *count = n;
  return MCL_OK;
}

int mcl_create(const mcl_config* cfg, mcl_handle** out) {
  if (!cfg || !out) {
    g_create_err = "mcl_create: null argument";
    return MCL_ERR_INVALID;
  }
  *out = nullptr;
  if (cfg->n_particles < 1 || cfg->n_particles > 0x7fffffffll) {
    g_create_err = "mcl_create: n_particles out of range";
    return MCL_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_err = "mcl_create: no HIP device visible (this library has no CPU fallback)";
    return MCL_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    g_create_err = "mcl_create: device ordinal out of range";
    return MCL_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  memset(&prop, 0, sizeof prop);
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_err = std::string("mcl_create: device is not gfx950 (found ") + prop.gcnArchName + ")";
    return MCL_ERR_NO_DEVICE;
  }
  mcl_handle* h = new mcl_handle();
  h->cfg = *cfg;
  h->n = cfg->n_particles;
  h->world = cfg->world > 1 ? cfg->world : 1;
  h->rank = h->world > 1 ? cfg->rank : 0;
  h->ng = cfg->n_global > 0 ? cfg->n_global : cfg->n_particles;
  h->goff = h->world > 1 ? cfg->global_offset : 0;
  h->device = cfg->device;
  memset(&h->tacc, 0, sizeof h->tacc);
  {
    auto on = [](const char* name) {
      const char* v = getenv(name);
      return v && v[0] == '1';
    };
    h->env_debug_work = getenv("MCL_DEBUG_WORK") != nullptr;
    if (const char* sv = getenv("MCL_SORT_VISITS")) h->env_sort = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_SWEEP")) h->env_sweep = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_SWEEP_NSUB")) h->env_nsub = (sv[0] == '2' || sv[0] == '4') ? sv[0] - '0' : 1;
    h->env_force_comm = on("MCL_FORCE_COMM");
    if (const char* ex = getenv("MCL_EXCHANGE")) h->exch_allgather = strcmp(ex, "allgather") == 0;
    if (const char* fi = getenv("MCL_FAULT_INJECT")) h->fault_step = strcmp(fi, "step_after_predict") == 0;
    h->env_no_overlap = on("MCL_NO_OVERLAP");
  }
  if (h->ng > 0xffffffffll || h->goff + h->n > h->ng || h->rank >= h->world) {
    g_create_err = "mcl_create: inconsistent shard geometry";
    delete h;
    return MCL_ERR_INVALID;
  }
  if (h->world > 1 && (h->ng != h->n * h->world || h->goff != h->n * h->rank)) {
    g_create_err = "mcl_create: shards must be equal-sized contiguous blocks (n_global = world * n_particles)";
    delete h;
    return MCL_ERR_INVALID;
  }
#define CREATE_CHK(call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      g_create_err = std::string(#call " failed: ") + hipGetErrorString(e_); \
      mcl_destroy(h);                                                      \
      return e_ == hipErrorOutOfMemory ? MCL_ERR_ALLOC : MCL_ERR_HIP;      \
    }                                                                      \
  } while (0)
  CREATE_CHK(hipSetDevice(h->device));
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t n = (size_t)h->n, ng = (size_t)h->ng;
  CREATE_CHK(hipMalloc(&h->state[0], sizeof(double) * 6 * n));
  CREATE_CHK(hipMalloc(&h->state[1], sizeof(double) * 6 * n));
  CREATE_CHK(hipMemsetAsync(h->state[0], 0, sizeof(double) * 6 * n, h->stream));
  CREATE_CHK(hipMemsetAsync(h->state[1], 0, sizeof(double) * 6 * n, h->stream));
  if (h->world > 1 && h->exch_allgather) CREATE_CHK(hipMalloc(&h->state_glob, sizeof(double) * 6 * ng));
  CREATE_CHK(hipMalloc(&h->lw, sizeof(double) * n));
  CREATE_CHK(hipMalloc(&h->q, sizeof(u64) * n));
  CREATE_CHK(hipMalloc(&h->ncum, sizeof(u32) * ng));
  CREATE_CHK(hipMalloc(&h->zcum, sizeof(u32) * ng));
  h->ntiles_loc = (h->n + MCL_SCAN_TILE - 1) / MCL_SCAN_TILE;
  h->ntiles_glob = (h->ng + MCL_SCAN_TILE - 1) / MCL_SCAN_TILE;
  CREATE_CHK(hipMalloc(&h->tile64, sizeof(u64) * (size_t)(h->ntiles_loc + 1)));
  CREATE_CHK(hipMalloc(&h->tile32, sizeof(u32) * (size_t)(h->ntiles_glob + 1)));
  CREATE_CHK(hipMalloc(&h->part, sizeof(double) * MOM_COUNT * MCL_MAX_GRID));
  CREATE_CHK(hipMalloc(&h->scal, sizeof(double) * 64));
  CREATE_CHK(hipMemsetAsync(h->scal, 0, sizeof(double) * 64, h->stream));
  CREATE_CHK(hipMalloc(&h->zr, sizeof(u32) * n));
  CREATE_CHK(hipMalloc(&h->dupes32, sizeof(u32) * ng));
  CREATE_CHK(hipMalloc(&h->desc, sizeof(u64) * (size_t)(h->ntiles_glob + 1)));
  CREATE_CHK(hipMemsetAsync(h->desc, 0, sizeof(u64) * (size_t)(h->ntiles_glob + 1), h->stream));
  CREATE_CHK(hipMalloc(&h->ctrl, CTRL_BYTES));
  CREATE_CHK(hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));
  CREATE_CHK(hipMalloc(&h->totals, sizeof(u64) * (size_t)(h->world + 1)));
  CREATE_CHK(hipMemsetAsync(h->totals, 0, sizeof(u64) * (size_t)(h->world + 1), h->stream));
  CREATE_CHK(hipHostMalloc(&h->host_pin, sizeof(double) * RING_STRIDE * MEAN_RING, hipHostMallocDefault));
  if (hipHostGetDevicePointer((void**)&h->host_pin_dev, h->host_pin, 0) != hipSuccess) h->host_pin_dev = nullptr;
  CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
  *out = h;
  return MCL_OK;
}

int mcl_destroy(mcl_handle* h) {
  if (!h) return MCL_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  t_collect(h);
  for (auto& e : h->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (h->comm2) ncclCommDestroy(h->comm2);
  if (h->comm) ncclCommDestroy(h->comm);
  h->comm2 = h->comm = nullptr;
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->ev_state_ready) (void)hipEventDestroy(h->ev_state_ready);
  if (h->ev_gather_done) (void)hipEventDestroy(h->ev_gather_done);
  void* bufs[] = {h->state[0], h->state[1], h->state_glob, h->lw, h->wnorm, h->q, h->ncum, h->zcum, h->zr, h->dupes32, h->desc, h->ctrl,
                  h->tile64, h->tile32, h->part, h->scal, h->totals, h->idx, h->replay_dev, h->pose7,
                  h->beam_sc, h->ranges_dev, h->exp_dev, h->grid, h->pose_dev, h->mbes_worklist, h->mbes_groups, h->sort_keys, h->sort_keys_out, h->sort_idx, h->mbes_perm, h->sort_tmp, h->sweep_buf[0], h->sweep_buf[1], h->defer_idx, h->defer_idx2, h->lm_worklist, h->cq, h->u53, h->cnt, h->first,
                  h->flags, h->fcum, h->copies, h->ccum, h->dupes, h->cs, h->chunk, h->uni_dev, h->lsx, h->xsend, h->xrecv};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (h->mesh) mesh_free(h->mesh);
  if (h->landmarks) landmarks_free(h->landmarks);
  if (h->det_dev) (void)hipFree(h->det_dev);
  if (h->host_pin) (void)hipHostFree(h->host_pin);
  if (h->lsx_host) (void)hipHostFree(h->lsx_host);
  if (h->work_host) (void)hipHostFree(h->work_host);
  if (h->copy_stream) {
    (void)hipStreamSynchronize(h->copy_stream);
    (void)hipStreamDestroy(h->copy_stream);
  }
  for (int k = 0; k < 2; ++k) {
    if (h->sweep_stage[k]) (void)hipHostFree(h->sweep_stage[k]);
    if (h->ev_stage[k]) (void)hipEventDestroy(h->ev_stage[k]);
  }
  for (auto& e : h->ev_upd)
    if (e) (void)hipEventDestroy(e);
  for (auto& sl : h->pin_ring) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.p) (void)hipHostFree(sl.p);
  }
  if (h->asg_dev) (void)hipFree(h->asg_dev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MCL_OK;
}

int mcl_init_particles(mcl_handle* h, const double* replay_normals) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!replay_normals) return fail(h, MCL_ERR_INVALID, "init_particles: REPLAY mode needs n x 6 normals");
    RET_IF(upload_replay(h, replay_normals));
    rp = h->replay_dev;
  }
  RET_IF(cancel_state_gather(h));
  h->uni_valid = false;
  NoiseArgs a = noise_args(h, h->cfg.init_cov, 0u, 0u);
  t_begin(h, MCL_K_NOISE);
  k_add_noise<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, rp, 1);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->step_predict = 0;
  h->step_resample = 0;
  h->have_lw = h->have_cdf = false;
  return MCL_OK;
}

int mcl_predict(mcl_handle* h, const mcl_odom* odom, double dt, const double* replay_normals) {
  if (!h || !odom) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  RET_IF(cancel_state_gather(h));
  return do_predict(h, odom, dt, replay_normals);
}

int mcl_update_gps(mcl_handle* h, double gx_map, double gy_map) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!(h->cfg.meas_std > 0.0)) return fail(h, MCL_ERR_INVALID, "update_gps: meas_std must be > 0");
  GpsArgs a;
  for (int k = 0; k < 4; ++k) {
    a.r0[k] = h->cfg.m2o[k];
    a.r1[k] = h->cfg.m2o[4 + k];
  }
  const double s2 = h->cfg.meas_std * h->cfg.meas_std;
  a.gx = gx_map;
  a.gy = gy_map;
  a.inv_s2 = 1.0 / s2;
  a.lognorm = std::log(2.0 * MCL_PI * s2);
  t_begin(h, MCL_K_UPDATE_GPS);
  k_gps_logw<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, h->lw);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->weight_mode = MCL_WEIGHT_LINEAR_FLOOR;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_set_map_grid(mcl_handle* h, const float* z, int32_t nx, int32_t ny, double ox, double oy, double res) {
  if (!h || !z || nx < 2 || ny < 2 || !(res > 0.0)) return fail(h, MCL_ERR_INVALID, "set_map_grid: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->grid) (void)hipFree(h->grid);
  h->grid = nullptr;
  const size_t cnt = (size_t)nx * (size_t)ny;
  HIPCHK(h, hipMalloc(&h->grid, sizeof(float) * cnt));
  HIPCHK(h, hipMemcpy(h->grid, z, sizeof(float) * cnt, hipMemcpyHostToDevice));
  float mn = z[0], mx = z[0];
  for (size_t k = 1; k < cnt; ++k) {
    if (z[k] < mn) mn = z[k];
    if (z[k] > mx) mx = z[k];
  }
  h->gnx = nx;
  h->gny = ny;
  h->gox = ox;
  h->goy = oy;
  h->gres = res;
  h->gzmin = mn;
  h->gzmax = mx;
  {
    // steepest gradient of a bilinear patch: its x slope lies between those of the cell's two x edges, its y slope
    // between those of the two y edges
    double g2 = 0.0;
    for (int ix = 0; ix + 1 < nx; ++ix)
      for (int iy = 0; iy + 1 < ny; ++iy) {
        const size_t k = (size_t)ix * ny + iy;
        const double h00 = z[k], h01 = z[k + 1], h10 = z[k + ny], h11 = z[k + ny + 1];
        const double ax = std::max(std::fabs(h10 - h00), std::fabs(h11 - h01));
        const double ay = std::max(std::fabs(h01 - h00), std::fabs(h11 - h10));
        g2 = std::max(g2, ax * ax + ay * ay);
      }
    h->gslope_max = std::sqrt(g2) / res;
  }
  h->map_kind = 0;
  return MCL_OK;
}

int mcl_set_map_mesh(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt) {
  return mcl_set_map_mesh_ex(h, verts, nv, tris, nt, 0u);
}

int mcl_set_map_mesh_ex(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt,
                        uint32_t flags) {
  if (!h || !verts || !tris || nv < 3 || nt < 1) return fail(h, MCL_ERR_INVALID, "set_map_mesh: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->mesh) mesh_free(h->mesh);
  h->mesh = nullptr;
  std::string err;
  int rc = mesh_build(verts, nv, tris, nt, &h->mesh, &err);
  if (rc != MCL_OK) {
    h->err = err;
    return rc;
  }
  h->map_kind = 1;
  h->mesh_heightfield = (flags & MCL_MESH_HEIGHTFIELD) != 0;
  h->force_general_mesh = (flags & (MCL_MESH_GENERAL | MCL_MESH_UNSTRUCTURED)) != 0;
  h->mesh_no_sweep = (flags & MCL_MESH_GENERAL) != 0;
  if (h->mesh_heightfield && h->mesh->n_vertical > 0) {
    h->err = "set_map_mesh: MCL_MESH_HEIGHTFIELD declared but the mesh has vertical faces";
    h->mesh_heightfield = false;
    return MCL_ERR_INVALID;
  }
  return MCL_OK;
}

int mcl_update_mbes(mcl_handle* h, const float* ranges, const float* beam_angles, int32_t B, double sigma,
                    double r_max, const double sensor_offset[6]) {
  if (!h || !ranges || !beam_angles || B < 1 || !(sigma > 0.0) || !(r_max > 0.0))
    return fail(h, MCL_ERR_INVALID, "update_mbes: bad argument");
  RET_IF(set_device(h));
  RET_IF(upload_beams(h, ranges, beam_angles, B));
  RET_IF(launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0));
  h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_mbes_expected(mcl_handle* h, int64_t first, int64_t count, const float* beam_angles, int32_t B,
                      double r_max, const double sensor_offset[6], float* out) {
  if (!h || !beam_angles || !out || B < 1 || first < 0 || count < 1 || first + count > h->n)
    return fail(h, MCL_ERR_INVALID, "mbes_expected: bad argument");
  RET_IF(set_device(h));
  RET_IF(upload_beams(h, nullptr, beam_angles, B));
  const size_t need = (size_t)count * (size_t)B;
  if (need > h->exp_cap) {
    if (h->exp_dev) (void)hipFree(h->exp_dev);
    h->exp_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->exp_dev, sizeof(float) * need));
    h->exp_cap = need;
  }
  RET_IF(launch_mbes(h, false, B, 1.0, r_max, sensor_offset, nullptr, h->exp_dev, first, count));
  HIPCHK(h, hipMemcpyAsync(out, h->exp_dev, sizeof(float) * need, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_landmarks(mcl_handle* h, const double* xyz, int64_t n_landmarks) {
  if (!h || !xyz || n_landmarks < 1) return fail(h, MCL_ERR_INVALID, "set_landmarks: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->landmarks) landmarks_free(h->landmarks);
  h->landmarks = new LandmarkDev();
  h->landmarks->host_xyz.assign(xyz, xyz + 3 * n_landmarks);
  return MCL_OK;
}

namespace {
// largest eigenvalue bound of a symmetric 3x3 (xx xy xz yy yz zz): Gershgorin
double sym3_lam_bound(const double* s) {
  const double r0 = s[0] + std::fabs(s[1]) + std::fabs(s[2]), r1 = s[3] + std::fabs(s[1]) + std::fabs(s[4]),
               r2 = s[5] + std::fabs(s[2]) + std::fabs(s[4]);
  return std::max(r0, std::max(r1, r2));
}
double sym3_det(const double* s) {
  return s[0] * (s[3] * s[5] - s[4] * s[4]) - s[1] * (s[1] * s[5] - s[4] * s[2]) + s[2] * (s[1] * s[4] - s[3] * s[2]);
}
// measurement covariance of this update: the Q of mcl_set_landmark_noise, else sigma^2 I
void landmark_q(const LandmarkDev* L, double sigma, double Q[6]) {
  if (L->have_q) {
    for (int k = 0; k < 6; ++k) Q[k] = L->Q[k];
  } else {
    Q[0] = Q[3] = Q[5] = sigma * sigma;
    Q[1] = Q[2] = Q[4] = 0.0;
  }
}
// every landmark inside the gate lies within this distance of the detection: d^2 >= |nu|^2 / lambda_max(S)
double landmark_gate_radius(const LandmarkDev* L, double sigma, double gate) {
  if (!L->maha) return sigma * std::sqrt(gate);
  double Q[6];
  landmark_q(L, sigma, Q);
  return std::sqrt(gate * (L->lam_cov_max + sym3_lam_bound(Q)));
}
void landmark_noise_args(const LandmarkDev* L, double sigma, LandmarkArgs& a) {
  a.maha = L->maha ? 1 : 0;
  a.lmcov = L->lmcov;
  landmark_q(L, sigma, a.Q);
  a.logdet_q = std::log(sym3_det(a.Q));
  a.lognorm = 1.5 * std::log(2.0 * MCL_PI) + 0.5 * a.logdet_q;  // isotropic: 3/2 log 2pi + 3 log sigma
}
}  // namespace

int mcl_set_landmark_noise(mcl_handle* h, const double* cov6, const double Q6[6]) {
  if (!h) return MCL_ERR_INVALID;
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, "set_landmark_noise: no feature map (call mcl_set_landmarks first)");
  RET_IF(set_device(h));
  LandmarkDev* L = h->landmarks;
  const size_t n = L->host_xyz.size() / 3;
  L->host_cov.clear();
  L->lam_cov_max = 0.0;
  if (cov6) {
    for (size_t i = 0; i < n; ++i) {
      const double* s = cov6 + 6 * i;
      if (!(s[0] >= 0.0 && s[3] >= 0.0 && s[5] >= 0.0) || !(sym3_det(s) >= 0.0))
        return fail(h, MCL_ERR_INVALID, "set_landmark_noise: landmark covariance not positive semi-definite");
      L->lam_cov_max = std::max(L->lam_cov_max, sym3_lam_bound(s));
    }
    L->host_cov.assign(cov6, cov6 + 6 * n);
  }
  L->have_q = Q6 != nullptr;
  if (Q6) {
    if (!(sym3_det(Q6) > 0.0) || !(Q6[0] > 0.0)) return fail(h, MCL_ERR_INVALID, "set_landmark_noise: Q must be positive definite");
    for (int k = 0; k < 6; ++k) L->Q[k] = Q6[k];
  }
  L->maha = cov6 != nullptr || Q6 != nullptr;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  L->built_for = -1.0;  // the covariances travel with the next grid build
  return MCL_OK;
}

int mcl_update_landmarks(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k, double gate,
                         const double sensor_offset[6], int32_t accumulate) {
  if (!h || !det_xyz || n_det < 1 || !(sigma > 0.0) || k < 1 || k > LM_MAX_K || !(gate > 0.0))
    return fail(h, MCL_ERR_INVALID, "update_landmarks: bad argument (1 <= k <= 4)");
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, "update_landmarks: no feature map (call mcl_set_landmarks first)");
  if (accumulate && !h->have_lw) return fail(h, MCL_ERR_STATE, "update_landmarks: nothing to accumulate onto");
  RET_IF(set_device(h));
  {
    // the cell grid depends on the gate radius only: rebuild (and drain the stream first -- the old
    // arrays may still be read by a kernel in flight) only when it changes
    const double radius = landmark_gate_radius(h->landmarks, sigma, gate);
    if (!(h->landmarks->built_for == radius && h->landmarks->lm)) {
      std::string err;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      int rc = landmarks_build(h->landmarks, radius, &err);
      if (rc != MCL_OK) {
        h->err = err;
        return rc;
      }
    }
  }
  if (n_det > h->det_cap) {
    if (h->det_dev) (void)hipFree(h->det_dev);
    h->det_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->det_dev, sizeof(double) * 3 * (size_t)n_det));
    h->det_cap = n_det;
  }
  RET_IF(upload(h, h->det_dev, det_xyz, sizeof(double) * 3 * (size_t)n_det));
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = sensor_offset ? sensor_offset : zero6;
  LandmarkArgs a;
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int q = 0; q < 12; ++q) a.m2o[q] = h->cfg.m2o[q];
  for (int q = 0; q < 3; ++q) a.off_t[q] = so[q];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.det = h->det_dev;
  a.n_det = n_det;
  a.lm = h->landmarks->lm;
  a.cell_start = h->landmarks->cell_start;
  a.gx = h->landmarks->gx;
  a.gy = h->landmarks->gy;
  a.x0 = h->landmarks->x0;
  a.y0 = h->landmarks->y0;
  a.inv_cs = 1.0 / h->landmarks->cs;
  a.inv_s2 = 1.0 / (sigma * sigma);
  a.gate = gate;
  landmark_noise_args(h->landmarks, sigma, a);
  a.k = k;
  a.accumulate = accumulate ? 1 : 0;
  a.lw = h->lw;
  t_begin(h, MCL_K_UPDATE_MBES);
  long long blocks = (h->n + 255) / 256;   // (a wave per 64 particles, four waves per workgroup)
  if (blocks > 16384) blocks = 16384;
  if (a.maha)
    k_landmark_update<true><<<(unsigned)blocks, 256, 0, h->stream>>>(a);
  else
    k_landmark_update<false><<<(unsigned)blocks, 256, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  if (!accumulate) h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_update_landmarks_assign(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k_cand,
                                double gate, double new_mh_dist, const double sensor_offset[6], int32_t accumulate,
                                int32_t* assign_out, int64_t n_keep) {
  if (!h || !det_xyz || n_det < 1 || n_det > LM_SUB || !(sigma > 0.0) || k_cand < 1 || k_cand > LA_KC || !(gate > 0.0) ||
      !(new_mh_dist >= 0.0) || n_keep < 0 || (n_keep > 0 && !assign_out))
    return fail(h, MCL_ERR_INVALID, "update_landmarks_assign: bad argument (n_det <= 16, 1 <= k_cand <= 8)");
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, "update_landmarks_assign: no feature map (call mcl_set_landmarks first)");
  if (accumulate && !h->have_lw) return fail(h, MCL_ERR_STATE, "update_landmarks_assign: nothing to accumulate onto");
  RET_IF(set_device(h));
  {
    // the cell grid depends on the gate radius only: rebuild (and drain the stream first -- the old
    // arrays may still be read by a kernel in flight) only when it changes
    const double radius = landmark_gate_radius(h->landmarks, sigma, gate);
    if (!(h->landmarks->built_for == radius && h->landmarks->lm)) {
      std::string err;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      int rc = landmarks_build(h->landmarks, radius, &err);
      if (rc != MCL_OK) {
        h->err = err;
        return rc;
      }
    }
  }
  if (n_det > h->det_cap) {
    if (h->det_dev) (void)hipFree(h->det_dev);
    h->det_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->det_dev, sizeof(double) * 3 * (size_t)n_det));
    h->det_cap = n_det;
  }
  RET_IF(upload(h, h->det_dev, det_xyz, sizeof(double) * 3 * (size_t)n_det));
  if (n_keep > h->n) n_keep = h->n;
  int* asg_dev = nullptr;
  if (n_keep > 0) {
    const size_t need = (size_t)n_keep * (size_t)n_det;
    if (need > h->asg_cap) {
      if (h->asg_dev) (void)hipFree(h->asg_dev);
      h->asg_dev = nullptr;
      h->asg_cap = 0;
      HIPCHK(h, hipMalloc(&h->asg_dev, sizeof(int) * need));
      h->asg_cap = need;
    }
    asg_dev = h->asg_dev;
  }
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = sensor_offset ? sensor_offset : zero6;
  LandmarkAssignArgs aa;
  LandmarkArgs& a = aa.base;
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int q = 0; q < 12; ++q) a.m2o[q] = h->cfg.m2o[q];
  for (int q = 0; q < 3; ++q) a.off_t[q] = so[q];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.det = h->det_dev;
  a.n_det = n_det;
  a.lm = h->landmarks->lm;
  a.cell_start = h->landmarks->cell_start;
  a.gx = h->landmarks->gx;
  a.gy = h->landmarks->gy;
  a.x0 = h->landmarks->x0;
  a.y0 = h->landmarks->y0;
  a.inv_cs = 1.0 / h->landmarks->cs;
  a.inv_s2 = 1.0 / (sigma * sigma);
  a.gate = gate;
  landmark_noise_args(h->landmarks, sigma, a);
  a.k = k_cand;
  a.accumulate = accumulate ? 1 : 0;
  a.lw = h->lw;
  aa.orig = h->landmarks->orig;
  aa.new_mh = new_mh_dist;
  aa.k_cand = k_cand;
  aa.assign_out = asg_dev;
  aa.n_keep = n_keep;
  hipError_t le = hipSuccess;
  if (!h->lm_worklist) le = hipMalloc(&h->lm_worklist, sizeof(int) * ((size_t)h->n + 1));
  if (le == hipSuccess) {
    aa.worklist = h->lm_worklist;
    aa.work_count = h->lm_worklist + h->n;
    le = hipMemsetAsync(aa.work_count, 0, sizeof(int), h->stream);
  }
  if (le == hipSuccess) {
    t_begin(h, MCL_K_UPDATE_MBES);
    long long blocks = (h->n + LA_PER_BLOCK - 1) / LA_PER_BLOCK;
    if (blocks > 32768) blocks = 32768;
    // every particle: conflict-free answer or worklist entry; then the solver over the worklist (its grid
    // strides over the device-side count)
    k_landmark_assign<false><<<(unsigned)blocks, LA_PER_BLOCK * LM_SUB, 0, h->stream>>>(aa);
    k_landmark_assign<true><<<(unsigned)std::min<long long>(blocks, 2048), LA_PER_BLOCK * LM_SUB, 0, h->stream>>>(aa);
    t_end(h);
    le = hipGetLastError();
  }
  if (le == hipSuccess && n_keep > 0)
    le = hipMemcpyAsync(assign_out, asg_dev, sizeof(int) * (size_t)n_keep * n_det, hipMemcpyDeviceToHost, h->stream);
  if (le == hipSuccess && n_keep > 0) le = hipStreamSynchronize(h->stream);
  HIPCHK(h, le);
  if (!accumulate) h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_resample(mcl_handle* h, const double* uniforms, int64_t n_uniforms, const double* replay_normals) {
  if (!h) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm)
    return fail(h, MCL_ERR_STATE, "resample: multi-shard handle needs mcl_comm_init or mcl_group_resample");
  if (h->cfg.rng_mode == MCL_RNG_REPLAY && !replay_normals)
    return fail(h, MCL_ERR_INVALID, "resample: REPLAY mode needs n x 6 normals");
  const double* rn[1] = {replay_normals};
  return run_resample(&h, 1, uniforms, n_uniforms, rn);
}

int mcl_resample_prepare(mcl_handle* h, int64_t* n_uniforms) {
  if (!h || !n_uniforms) return MCL_ERR_INVALID;
  if (!h->have_lw) return fail(h, MCL_ERR_STATE, "resample_prepare: no weights (call an update first)");
  RET_IF(set_device(h));
  int rc;
  *n_uniforms = uniforms_needed(h, &rc);
  return rc;
}

int mcl_group_resample(mcl_handle** shards, int32_t ns, const double* uniforms, int64_t n_uniforms,
                       const double* const* replay_normals) {
  if (!shards || ns < 1) return MCL_ERR_INVALID;
  for (int s = 0; s < ns; ++s)
    if (!shards[s] || shards[s]->world != ns || shards[s]->rank != s)
      return fail(shards[0], MCL_ERR_INVALID, "group_resample: shards must be ranks 0..n-1 of one world");
  return run_resample(shards, ns, uniforms, n_uniforms, replay_normals);
}

int mcl_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!h || !mean6 || !cov9) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, "mean_cov: multi-shard handle needs a communicator");
  RET_IF(run_mean_cov_async(&h, 1));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  finish_mean_cov(h, mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_mean_cov_async(mcl_handle* h) {
  if (!h) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, "mean_cov: multi-shard handle needs a communicator");
  return run_mean_cov_async(&h, 1);
}

int mcl_group_mean_cov(mcl_handle** shards, int32_t ns, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!shards || ns < 1 || !mean6 || !cov9) return MCL_ERR_INVALID;
  RET_IF(run_mean_cov_async(shards, ns));
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(shards[s]));
    HIPCHK(shards[s], hipStreamSynchronize(shards[s]->stream));
  }
  finish_mean_cov(shards[0], mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_last_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!h || !mean6 || !cov9) return MCL_ERR_INVALID;
  if (!h->have_meancov) return fail(h, MCL_ERR_STATE, "last_mean_cov: nothing computed yet");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  finish_mean_cov(h, mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_mean_history(mcl_handle* h, int64_t last_k, double* mean6_out) {
  if (!h || !mean6_out || last_k < 1) return MCL_ERR_INVALID;
  if (last_k > h->mean_count || last_k > MEAN_RING) return fail(h, MCL_ERR_INVALID, "mean_history: not that many results kept");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (long long k = 0; k < last_k; ++k) {
    double yaw, cov9[9];
    finish_mean_cov(h, mean6_out + 6 * k, &yaw, cov9, h->mean_count - last_k + k);
  }
  return MCL_OK;
}

int mcl_get_poses(mcl_handle* h, double* pose7) {
  if (!h || !pose7) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!h->pose7) HIPCHK(h, hipMalloc(&h->pose7, sizeof(double) * 7 * (size_t)h->n));
  k_poses<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->pose7);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(pose7, h->pose7, sizeof(double) * 7 * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_particles(mcl_handle* h, double* soa, double* w) {
  if (!h || !soa) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(soa, h->state[h->cur], sizeof(double) * 6 * (size_t)h->n, hipMemcpyDeviceToHost,
                           h->stream));
  if (w) {
    if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_particles: weights exist only after a resample");
    if (!h->wnorm) HIPCHK(h, hipMalloc(&h->wnorm, sizeof(double) * (size_t)h->n));
    k_normalised_weights<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->totals, h->world, h->wnorm);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(w, h->wnorm, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_particles(mcl_handle* h, const double* soa) {
  if (!h || !soa) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  RET_IF(cancel_state_gather(h));
  h->uni_valid = false;
  HIPCHK(h, hipMemcpyAsync(h->state[h->cur], soa, sizeof(double) * 6 * (size_t)h->n, hipMemcpyHostToDevice,
                           h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_log_weights(mcl_handle* h, double* lw) {
  if (!h || !lw) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(lw, h->lw, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_log_weights(mcl_handle* h, const double* lw, int32_t weight_mode) {
  if (!h || !lw || weight_mode < 0 || weight_mode > 2) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(h->lw, lw, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->weight_mode = weight_mode;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_get_last_indices(mcl_handle* h, int32_t* idx) {
  if (!h || !idx) return MCL_ERR_INVALID;
  if (!h->have_cdf && !h->idx_explicit) return fail(h, MCL_ERR_STATE, "get_last_indices: no resample yet");
  RET_IF(set_device(h));
  if (!h->idx) HIPCHK(h, hipMalloc(&h->idx, sizeof(int) * (size_t)h->n));
  if (!h->idx_explicit) RET_IF(ensure_global_cdf(h));
  if (!h->idx_explicit) {
    k_indices<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->ncum, h->ng, h->goff, h->n, h->idx);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipMemcpyAsync(idx, h->idx, sizeof(int) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_last_offspring_cdf(mcl_handle* h, uint32_t* ncum) {
  if (!h || !ncum) return MCL_ERR_INVALID;
  if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_last_offspring_cdf: no resample yet");
  RET_IF(set_device(h));
  RET_IF(ensure_global_cdf(h));
  HIPCHK(h, hipMemcpyAsync(ncum, h->ncum, sizeof(u32) * (size_t)h->ng, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_fixed_weights(mcl_handle* h, uint64_t* q, uint64_t* total) {
  if (!h || !q) return MCL_ERR_INVALID;
  if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_fixed_weights: no resample yet");
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(q, h->q, sizeof(u64) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  std::vector<u64> t(h->world);
  HIPCHK(h, hipMemcpyAsync(t.data(), h->totals, sizeof(u64) * (size_t)h->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (total) {
    u64 T = 0;
    for (u64 v : t) T += v;
    *total = T;
  }
  return MCL_OK;
}

int mcl_step_mbes(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges, const float* beam_angles,
                  int32_t B, double sigma, double r_max, const double sensor_offset[6]) {
  if (!h || !odom || !ranges || !beam_angles) return MCL_ERR_INVALID;
  if (h->cfg.rng_mode != MCL_RNG_NATIVE) return fail(h, MCL_ERR_INVALID, "step_mbes: NATIVE rng only");
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, "step_mbes: multi-shard handle needs mcl_comm_init");
  if (B < 1 || !(sigma > 0.0) || !(r_max > 0.0)) return fail(h, MCL_ERR_INVALID, "step_mbes: bad argument");
  if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, "step_mbes: no map (call mcl_set_map_grid/mesh first)");
  RET_IF(set_device(h));
  // predict writes the MBES pose records of the new state in the same pass (the map and sensor offset are known here)
  // (the beam table first: the group classification in that kernel follows the two extreme beams)
  RET_IF(upload_beams(h, ranges, beam_angles, B));
  MbesArgs pa;
  RET_IF(launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, false, &pa));
  bool pose_done = false;
  const bool sys = h->cfg.resample_scheme == MCL_RESAMPLE_SYSTEMATIC || h->cfg.resample_scheme == MCL_RESAMPLE_NAIVE;
  // (systematic scheme: the gather of this call substitutes z, roll, pitch -- the predict kernel does not store them)
  RET_IF(do_predict(h, odom, dt, nullptr, &pa, &pose_done, sys));
  int rc_u = h->fault_step ? fail(h, MCL_ERR_STATE, "step_mbes: injected fault after predict") : start_state_gather(h);
  if (rc_u == MCL_OK) rc_u = launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, pose_done);
  if (rc_u != MCL_OK) {
    const std::string keep = h->err;
    (void)cancel_state_gather(h);
    (void)materialise_uniform(h);
    h->err = keep;
    return rc_u;
  }
  h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->residual_k = -1;
  // resample; the gather pass also accumulates the sums of update_loc_pose of the new state
  rc_u = run_resample(&h, 1, nullptr, 0, nullptr, sys);
  if (rc_u != MCL_OK) {
    const std::string keep = h->err;
    (void)materialise_uniform(h);
    h->err = keep;
    return rc_u;
  }
  if (sys)
    RET_IF(collect_fused_moments(&h, 1));
  else
    RET_IF(run_mean_cov_async(&h, 1));
  return MCL_OK;
}

int mcl_group_step_mbes(mcl_handle** shards, int32_t ns, const mcl_odom* odom, double dt, const float* ranges,
                        const float* beam_angles, int32_t B, double sigma, double r_max, const double sensor_offset[6]) {
  if (!shards || ns < 1 || !odom || !ranges || !beam_angles) return MCL_ERR_INVALID;
  for (int s = 0; s < ns; ++s)
    if (!shards[s] || shards[s]->world != ns || shards[s]->rank != s)
      return fail(shards[0], MCL_ERR_INVALID, "group_step_mbes: shards must be ranks 0..n-1 of one world");
  if (B < 1 || !(sigma > 0.0) || !(r_max > 0.0)) return fail(shards[0], MCL_ERR_INVALID, "group_step_mbes: bad argument");
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = shards[s];
    if (h->cfg.rng_mode != MCL_RNG_NATIVE) return fail(h, MCL_ERR_INVALID, "group_step_mbes: NATIVE rng only");
    if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, "group_step_mbes: no map (call mcl_set_map_grid/mesh first)");
    if (h->cfg.resample_scheme != MCL_RESAMPLE_SYSTEMATIC && h->cfg.resample_scheme != MCL_RESAMPLE_NAIVE)
      return fail(h, MCL_ERR_UNSUPPORTED, "group_step_mbes: only the systematic scheme is sharded");
    RET_IF(set_device(h));
    // the same fused front half as mcl_step_mbes: predict writes the pose records, the sweep leaves max lw in the slots
    RET_IF(upload_beams(h, ranges, beam_angles, B));
    MbesArgs pa;
    RET_IF(launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, false, &pa));
    bool pose_done = false;
    int rc = do_predict(h, odom, dt, nullptr, &pa, &pose_done, true);
    if (rc == MCL_OK && h->fault_step) rc = fail(h, MCL_ERR_STATE, "group_step_mbes: injected fault after predict");
    if (rc == MCL_OK) rc = launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, pose_done);
    if (rc != MCL_OK) {
      const std::string keep = h->err;
      for (int t = 0; t <= s; ++t) (void)materialise_uniform(shards[t]);
      h->err = keep;
      return rc;
    }
    h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
    h->have_lw = true;
    h->residual_k = -1;
  }
  const int rc = run_resample(shards, ns, nullptr, 0, nullptr, true);
  if (rc != MCL_OK) {
    const std::string keep = shards[0]->err;
    for (int t = 0; t < ns; ++t) (void)materialise_uniform(shards[t]);
    shards[0]->err = keep;
    return rc;
  }
  return collect_fused_moments(shards, ns);
}

int mcl_exchange_plan(int32_t world, const uint32_t* lost, const uint32_t* surplus, int32_t rank, uint32_t* send_off,
                      uint32_t* send_cnt, uint32_t* recv_off, uint32_t* recv_cnt) {
  if (world < 1 || !lost || !surplus || rank < 0 || rank >= world || !send_off || !send_cnt || !recv_off || !recv_cnt)
    return MCL_ERR_INVALID;
  std::vector<u32> Lpre((size_t)world + 1, 0u), Spre((size_t)world + 1, 0u);
  unsigned long long tl = 0, ts = 0;
  for (int r = 0; r < world; ++r) {
    tl += lost[r];
    ts += surplus[r];
    if (tl > 0xffffffffull || ts > 0xffffffffull) return MCL_ERR_INVALID;
    Lpre[r + 1] = (u32)tl;
    Spre[r + 1] = (u32)ts;
  }
  if (tl != ts) return MCL_ERR_INVALID;   // every lost slot takes exactly one surplus copy
  for (int r = 0; r < world; ++r) {
    u32 lo, hi;
    plan_range(Lpre.data(), Spre.data(), rank, r, lo, hi);   // what `rank` holds and r needs
    send_off[r] = lo - Spre[rank];
    send_cnt[r] = hi - lo;
    plan_range(Lpre.data(), Spre.data(), r, rank, lo, hi);   // what r holds and `rank` needs
    recv_off[r] = lo - Lpre[rank];
    recv_cnt[r] = hi - lo;
  }
  return MCL_OK;
}

int mcl_exchange_stats(mcl_handle* h, int64_t* states_sent, int64_t* lost_slots, int32_t reset) {
  if (!h) return MCL_ERR_INVALID;
  if (states_sent) *states_sent = (int64_t)h->ex_sent;
  if (lost_slots) *lost_slots = (int64_t)h->ex_lost;
  if (reset) h->ex_sent = h->ex_lost = 0;
  return MCL_OK;
}

int mcl_sync(mcl_handle* h) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_resample_indices(int32_t scheme, const double* weights, int64_t n, const double* uniforms,
                         int64_t n_uniforms, int32_t device, int32_t* out) {
  if (!weights || !out || n < 1) return MCL_ERR_INVALID;
  mcl_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_particles = n;
  cfg.device = device;
  cfg.resample_scheme = scheme;
  cfg.rng_mode = MCL_RNG_REPLAY;
  mcl_handle* h = nullptr;
  int rc = mcl_create(&cfg, &h);
  if (rc != MCL_OK) return rc;
  rc = mcl_set_log_weights(h, weights, MCL_WEIGHT_LINEAR);
  if (rc == MCL_OK) {
    if (scheme != MCL_RESAMPLE_SYSTEMATIC && scheme != MCL_RESAMPLE_NAIVE) {
      int64_t need = 0;
      rc = mcl_resample_prepare(h, &need);
      if (rc == MCL_OK) rc = alt_indices(h, uniforms, n_uniforms);
      if (rc == MCL_OK) rc = mcl_get_last_indices(h, out);
      if (rc != MCL_OK) g_create_err = h->err;
    } else if (!uniforms || n_uniforms < 1 || !(uniforms[0] >= 0.0 && uniforms[0] < 1.0)) {
      g_create_err = "resample_indices: systematic needs one uniform in [0,1)";
      rc = MCL_ERR_INVALID;
    } else {
      uint64_t u53 = (uint64_t)std::floor(uniforms[0] * 9007199254740992.0);
      if (scheme == MCL_RESAMPLE_NAIVE) u53 |= MCL_U53_NAIVE;
      rc = phase_quantise(h, true);
      if (rc == MCL_OK) rc = phase_cdf(h, u53);
      if (rc == MCL_OK) {
        h->have_cdf = true;
        h->cdf_global = true;
        rc = mcl_get_last_indices(h, out);
      }
      if (rc != MCL_OK) g_create_err = h->err;
    }
  } else {
    g_create_err = h->err;
  }
  mcl_destroy(h);
  return rc;
}

int mcl_comm_unique_id(char id[128]) {
  if (!id) return MCL_ERR_INVALID;
  ncclUniqueId uid;
  static_assert(sizeof(ncclUniqueId) <= 128, "unique id size");
  if (ncclGetUniqueId(&uid) != ncclSuccess) {
    g_create_err = "ncclGetUniqueId failed";
    return MCL_ERR_COMM;
  }
  memset(id, 0, 128);
  memcpy(id, &uid, sizeof uid);
  return MCL_OK;
}

namespace {
// wait for an event with a deadline; 0 = done, 1 = timed out, negative = HIP error
int wait_event_ms(hipEvent_t ev, int timeout_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return -1;
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
      return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
void comm_teardown(mcl_handle* h, bool abort) {
  if (h->comm2) (void)(abort ? ncclCommAbort(h->comm2) : ncclCommDestroy(h->comm2));
  if (h->comm) (void)(abort ? ncclCommAbort(h->comm) : ncclCommDestroy(h->comm));
  h->comm2 = nullptr;
  h->comm = nullptr;
  h->gather_inflight = false;
}
}  // namespace

int mcl_comm_init_ex(mcl_handle* h, const char id[128], uint32_t flags) {
  if (!h || !id) return MCL_ERR_INVALID;
  if (h->world < 2 && !h->env_force_comm) return MCL_OK;  // MCL_FORCE_COMM=1: test hook, 1-rank communicator
  if (h->comm) return fail(h, MCL_ERR_STATE, "comm_init: communicator exists (mcl_comm_shutdown first)");
  RET_IF(set_device(h));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  NCCLCHK(h, ncclCommInitRank(&h->comm, h->world, uid, h->rank));
  if (h->exch_allgather && !h->state_glob) HIPCHK(h, hipMalloc(&h->state_glob, sizeof(double) * 6 * (size_t)h->ng));
  // second communicator + stream for the overlapped state all-gather (MCL_EXCHANGE=allgather only: the O(n)
  // exchange ships a few per cent of a shard and has nothing worth hiding); optional
  const bool overlap = h->exch_allgather && !(flags & MCL_COMM_NO_OVERLAP) && !h->env_no_overlap;
  if (overlap && ncclCommSplit(h->comm, 0, h->rank, &h->comm2, nullptr) == ncclSuccess && h->comm2) {
    if (!h->comm_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (!h->ev_state_ready) HIPCHK(h, hipEventCreateWithFlags(&h->ev_state_ready, hipEventDisableTiming));
    if (!h->ev_gather_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather_done, hipEventDisableTiming));
  } else {
    h->comm2 = nullptr;
  }
  return MCL_OK;
}

int mcl_comm_init(mcl_handle* h, const char id[128]) { return mcl_comm_init_ex(h, id, 0u); }

int mcl_comm_ranks(mcl_handle* h, int32_t* ranks, int32_t* overlap) {
  if (!h || !ranks) return MCL_ERR_INVALID;
  if (overlap) *overlap = h->comm2 ? 1 : 0;
  if (!h->comm) {
    *ranks = 1;
    return MCL_OK;
  }
  RET_IF(set_device(h));
  // every rank contributes 1: the sum is the number of ranks RCCL really connected
  int* d = (int*)(h->totals + h->world);  // scratch word behind the shard totals
  const int one = 1;
  HIPCHK(h, hipMemcpyAsync(d, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  NCCLCHK(h, ncclAllReduce(d, d, 1, ncclInt32, ncclSum, h->comm, h->stream));
  int got = 0;
  HIPCHK(h, hipMemcpyAsync(&got, d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *ranks = got;
  return MCL_OK;
}

int mcl_comm_selftest(mcl_handle* h, int32_t timeout_ms) {
  if (!h) return MCL_ERR_INVALID;
  if (!h->comm) return MCL_OK;
  if (timeout_ms < 1) timeout_ms = 1;
  RET_IF(set_device(h));
  // the exact concurrency pattern of mcl_step_mbes: the 6-array state all-gather on the second
  // communicator/stream while the first communicator runs its all-reduce + all-gathers, three rounds
  hipEvent_t done = nullptr;
  HIPCHK(h, hipEventCreateWithFlags(&done, hipEventDisableTiming));
  int rc = MCL_OK;
  for (int round = 0; round < 3 && rc == MCL_OK; ++round) {
    rc = start_state_gather(h);
    if (rc != MCL_OK) break;
    ncclResult_t e = ncclAllReduce(h->scal + 24, h->scal + 24, 1, ncclDouble, ncclMax, h->comm, h->stream);
    if (e == ncclSuccess) e = ncclAllGather(h->totals + h->rank, h->totals, 1, ncclUint64, h->comm, h->stream);
    if (e == ncclSuccess && h->exch_allgather)
      e = ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream);
    if (e == ncclSuccess && !h->exch_allgather && h->world > 1) {
      // the O(n) exchange's pattern: the hand-over records all-gathered, then grouped point-to-point transfers
      // (here: one word to the next rank, one from the previous)
      RET_IF(alloc_lsx(h));
      e = ncclAllGather(h->lsx + 4 * (size_t)h->rank, h->lsx, 4, ncclUint64, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclGroupStart();
      if (e == ncclSuccess) e = ncclSend(h->totals + h->rank, 1, ncclUint64, (h->rank + 1) % h->world, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclRecv(h->totals + h->world, 1, ncclUint64, (h->rank + h->world - 1) % h->world, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclGroupEnd();
    }
    if (e != ncclSuccess) {
      h->err = std::string("comm_selftest: ") + ncclGetErrorString(e);
      rc = MCL_ERR_COMM;
      break;
    }
    if (h->gather_inflight) {
      (void)hipStreamWaitEvent(h->stream, h->ev_gather_done, 0);
      h->gather_inflight = false;
    }
    (void)hipEventRecord(done, h->stream);
    const int w = wait_event_ms(done, timeout_ms);
    if (w != 0) {
      h->err = w > 0 ? "comm_selftest: collectives did not complete before the deadline (communicators aborted)"
                     : "comm_selftest: HIP error while waiting";
      comm_teardown(h, true);
      rc = MCL_ERR_COMM;
    }
  }
  (void)hipEventDestroy(done);
  return rc;
}

int mcl_comm_shutdown(mcl_handle* h, int32_t abort) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!abort && h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!abort && h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  comm_teardown(h, abort != 0);
  return MCL_OK;
}

int mcl_mbes_last_path(mcl_handle* h, int32_t* path, int64_t* handed_over, int64_t* deferred_groups) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int cnt[3] = {0, 0, 0};  // deferred groups, declined by the first sweep pass, declined by the second
  HIPCHK(h, hipMemcpy(cnt, h->ctrl + CTRL_WORK, sizeof cnt, hipMemcpyDeviceToHost));
  if (path) *path = h->sweep_now ? 1 : 0;
  if (handed_over) *handed_over = h->sweep_now ? (h->sweep_two_pass ? cnt[2] : cnt[1]) : 0;
  if (deferred_groups) *deferred_groups = cnt[0];
  return MCL_OK;
}

int mcl_timing_enable(mcl_handle* h, int32_t on) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  t_collect(h);
  h->timing = on != 0;
  return MCL_OK;
}

int mcl_timing_get(mcl_handle* h, mcl_timing* out) {
  if (!h || !out) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  t_collect(h);
  *out = h->tacc;
  memset(&h->tacc, 0, sizeof h->tacc);
  return MCL_OK;
}

}  // extern "C"
