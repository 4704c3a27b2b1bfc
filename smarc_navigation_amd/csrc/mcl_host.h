// mcl_host.h -- host side of libmcl_hip.so, part 1: the handle (device buffers, streams, communicators, caches) and
// the helpers every other part uses (error macros, launch geometry, timing regions, Philox on the host, uploads).
// One translation unit: mcl_api.hip includes mcl_host.h, mcl_host_resample.h, mcl_host_moments.h, mcl_host_update.h
// in this order and then defines the C ABI.
#pragma once
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
#include "mcl_host_pure.h"
#include "mcl_kernels.h"
#include "mcl_mbes.h"
#include "mcl_sweep.h"
#include "mcl_slice.h"
#include "mcl_mesh.h"
#include "mcl_resample.h"
#include "mcl_resample_alt.h"
#include "mcl_landmarks.h"

#define MEAN_RING 4096
#define RING_STRIDE 20  // doubles per mean/cov result: 16 payload + [16] format tag
// control block layout (bytes)
#define CTRL_SLOTS 0                       // MCL_MAX_SLOTS u64
#define CTRL_WORK (8 * MCL_MAX_SLOTS)      // int: groups deferred by the fast MBES kernel
#define CTRL_DEFER (CTRL_WORK + 4)         // int: particles the fan sweep handed to the general kernel
#define CTRL_LOOSE (CTRL_WORK + 8)         // int: groups the fan slice's group kernel left to the per-particle kernel
#define CTRL_T_QUANT (CTRL_WORK + 12)      // u32 tickets, self-resetting
#define CTRL_T_EXPAND (CTRL_WORK + 16)
#define CTRL_T_GATHER (CTRL_WORK + 20)
#define CTRL_T_VISIT (CTRL_WORK + 24)
#define CTRL_DEFER2 (CTRL_WORK + 28)       // int: particles the fan slice, as the TIN sweep's hand-over kernel, handed on to the general kernel
#define CTRL_SLOTS2 1024                   // a second set of MCL_MAX_SLOTS u64: max lw after the fused landmark update
#define CTRL_BYTES 2048

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
  int slot_set = 0;              // ... which set: 0 = CTRL_SLOTS (MBES kernels, k_max_slots), 1 = CTRL_SLOTS2 (fused landmark update)
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
  // [1] particles the sweep handed to the general kernel.  An update reads
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
  u32* defer_idx = nullptr;         // particles the sweep hands to k_mbes_cast<., ., 2>
  u32* defer2_idx = nullptr;        // TIN with holes: what the fan slice -- the sweep's hand-over kernel there -- declines in turn
  int env_handover_slice = -1;      // MCL_HANDOVER_SLICE=0: the ray traversal takes the TIN sweep's hand-overs even on meshes with holes (A/B)
  unsigned* reasons_dev = nullptr;  // -DSWEEP_REASONS builds: why the sweep declined (16 counters)
  // spatial visiting order of the sweep (mcl_kernels.h: VisitArgs): prepared by the fused step's gather, used by the
  // next fused predict
  u32 *visit_okey = nullptr, *visit_base = nullptr;
  unsigned short* visit_cnt = nullptr;
  u64* visit_desc = nullptr;
  u32 visit_epoch = 0;
  VisitPar* visit_par = nullptr;    // two entries: read / written alternately
  unsigned visit_flip = 0;
  bool gather_attr_set = false;     // k_resample_gather<true, true, true> may use its 112 KiB of dynamic LDS
  bool visit_ready = false;         // okey / hist / binbase describe the slots of the current state
  bool pose_visit = false;          // pose_dev lies in visiting order (written by the last fused predict)
  int env_visit = -1;               // MCL_VISIT=0/1 forces the decision (tests, A/B)
  int visit_nb[3] = {16, 32, 8};    // MCL_VISIT_BINS=x,y,yaw  (measured at 1 M x 512, sweep us: 8,8,8 277; 16,16,8 268; 16,16,16 265; 16,32,8 263)
  float visit_range = 2.5f;         // MCL_VISIT_RANGE: bins span mean +- range * sigma  (headline sweep us -- 4: 260.5, 3: 257.2, 2.5: 257.1; the tempered filter -- 4: 300, 3: 294, 2: 288)
  long long visit_min_n = 393216;   // MCL_VISIT_MIN_N: smaller shards are visited in slot order (the order costs ~20 us per step whatever
                                    // the size -- a launch, a second gather pass, scattered record stores --: measured worth +5 us at
                                    // 524 288 x 512, -9 us at 65 536 x 256)
  int env_sweep = -1;               // MCL_SWEEP=0/1 forces the decision (tests, A/B)
  int env_nsub = 0;                 // MCL_SWEEP_NSUB=1/2/4 forces the lanes per particle side (A/B)
  bool sweep_now = false;           // decided by the first launch_mbes call of an update
  bool slice_now = false;           // ... the fan slice (mcl_slice.h) casts it
  bool handover_slice_now = false;  // the last update's sweep hand-overs went through the fan slice first (TIN with holes)
  int env_slice = -1;               // MCL_SLICE=0 keeps the ray traversal on triangle soups (tests, A/B)
  int env_slice_group = -1;         // MCL_SLICE_GROUP=0: the fan slice casts every particle on its own (no shared candidate lists)
  u32* slice_loose = nullptr;       // k_mbes_slice_group: the groups of SLICE_G pose records it left to k_mbes_slice
  bool slice_attr_set = false;
  bool slice_group_ran = false;     // the last sliced update went through k_mbes_slice_group first
  size_t slice_attr_bytes = 0;
  int sweep_nvalid = 0;
  float* grid = nullptr;
  float* grid_pad = nullptr;   // the same heights inside a one-node ring of NaNs (fan sweep: MbesArgs::grid_pad)
  int gnx = 0, gny = 0;
  double gox = 0, goy = 0, gres = 1;
  float gzmin = 0, gzmax = 0;
  double gslope_max = 0;       // steepest patch gradient of the height grid (the fan sweep's tilt bound)
  MeshDev* mesh = nullptr;
  LandmarkDev* landmarks = nullptr;
  double* det_dev = nullptr;
  int det_cap = 0;
  // fused landmark step: the detections ride in the per-ping beam table's staged copy (its own stream, under the previous
  // step's kernels) instead of a copy on the compute stream in front of the predict
  const double* det_ride = nullptr;      // host detections waiting for the next table upload (3 x det_ride_n doubles)
  int det_ride_n = 0, det_ride_cap = 0;  // ... their count; room for them behind either table buffer
  const double* det_ride_dev = nullptr;  // where that upload put them (device), until the landmark kernel is launched
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
  // one collective for "maximum, then totals" (mcl_resample.h: k_quantise_tiles' records, k_shift_scan)
  u64* shrec = nullptr;          // device: world x SHREC_WORDS all-gathered shard records, then the shift word
  unsigned short* tile_bits = nullptr;   // device: ntiles_loc x 64 bit counts of this shard's tiles
  const u64* qshift_cur = nullptr;       // the shift the weights in `q` are read with (nullptr: none) -- set by every resample
  bool shrec_dirty = false;              // a launch that accumulates into the record is queued and k_shift_scan (which zeroes it) is not
  // the fused step's moments ride with the NEXT step's records instead of an all-reduce of their own:
  bool moments_ride = false;             // this gather writes its 13 sums into the shard's record (set by phase_gather)
  bool mom_pending = false;              // a ring entry is reserved whose sums still lie, per shard, in the records
  long long mom_pending_entry = -1;      // ... which one (index into the result ring, before the modulo)
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
  unsigned long long ex_ops = 0, ex_rounds = 0; // point-to-point operations (sends + receives) issued, likewise; exchanges
  int ex_nship = 6;                             // doubles per exchanged copy (3 right after a predict: x, y, yaw)
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
int fail(mcl_handle* h, int code, const std::string& msg) { return fail(h, code, msg.c_str()); }

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

}  // namespace
