// mcl_mbes.h -- MBES measurement update by RAY TRAVERSAL: per-particle, per-beam ray-cast against the bathymetric
// map -> expected ranges -> Gaussian log-likelihood (north_star; no reference symbol).  Since round 2 the
// default on regular meshes, height grids and height-field TINs is the fan sweep (mcl_sweep.h); these kernels cast
// what the sweep does not cover (triangle soups with vertical faces / folds / non-manifold edges, unsorted beam tables,
// small clouds) and the particles it hands over.  Shared by both: the pose records, MbesArgs, cast_clear.
//
//   k_mbes_pose / k_predict_pose : one thread per particle, fp64 -> sensor pose record (origin in map cell units,
//                 two columns of R_map_sensor) in HBM (48 B/particle); <true>: also the tile window and fast-path
//                 eligibility of every group of MBES_WAVES particles (classify_group).
//   k_mbes_fast : one WAVEFRONT per particle, lanes = consecutive beams (a fan is coherent: neighbouring lanes walk
//                 neighbouring cells), MBES_WAVES particles per workgroup share one LDS tile (heights; cell words for
//                 triangle records); the ray follows the surface's clearance cell by cell (cast_clear).
//   k_mbes_cast : the general kernel, one wavefront per particle on the map in global memory (L2) -- MODE 1: the groups
//                 k_mbes_fast left on the worklist (window clipped by the map border or larger than LDS), MODE 2: the
//                 particles the fan sweep handed over (a.perm[0 .. *a.n_dev)).
//
// DETERMINISM RULE (round 4): the log-likelihood of a particle is a pure function of its pose record, the beam table
// and the map -- never of the particles it happens to be grouped with, of the grid size, of a tile origin or of the
// order of a hand-over list.  Hence: (i) ray coordinates are relative to the PARTICLE's own cell (own_fan: integer cell
// + fraction in [0, 1)), tiles and arrays are addressed through a base pointer shifted to that cell; (ii) the water
// column is skipped down to the MAP's highest point, not a tile's; (iii) which algorithm casts a particle is decided
// by the particle alone (own_fan::simple: its own fan footprint lies inside the map -> the clearance walk, from LDS or
// from global memory alike; else the clipped general march).  A sharded filter therefore equals the unsharded one bit
// for bit on every path, and a hand-over list may be built with atomics.
// Bound: VALU issue + LDS reads; compulsory HBM traffic is only 48 B + 8 B per particle plus the map tile reads
// (L2-resident).
#pragma once
#include <hip/hip_fp16.h>

#include "mcl_device.h"
#include "mcl_kernels.h"

#ifndef MBES_WAVES
#define MBES_WAVES 8                         // particles per workgroup
#endif
#ifndef MBES_MIN_WAVES_PER_SIMD
#define MBES_MIN_WAVES_PER_SIMD 8            // grid / structured mesh: <= 64 VGPRs, 4 workgroups (32 waves) per CU
#endif
#ifndef MBES_MIN_WAVES_MESH
#define MBES_MIN_WAVES_MESH 6                // general triangle records: <= 80 VGPRs, 3 workgroups per CU
#endif
#define MBES_THREADS (MBES_WAVES * 64)
#ifndef MBES_TILE_FLOATS
#define MBES_TILE_FLOATS 8192                // 32 KiB tile in LDS
#endif
#ifndef MBES_DIR_BINS
#define MBES_DIR_BINS 16                     // fan-direction bins of the Morton visiting order (<= 16: 4 key bits)
#endif
#ifndef MBES_DIR_BINS
#define MBES_DIR_BINS 16                     // fan-direction bins of the Morton visiting order (<= 16: 4 key bits)
#endif
#ifndef MBES_TILE_MARGIN
#define MBES_TILE_MARGIN 0                   // nodes staged around a group's window (A/B: a margin re-uses tiles more often but raises the tile maximum the rays start from -- measured slower)
#endif

struct MbesPose {   // 48 B
  double um, vm;    // sensor origin in GLOBAL cell units (fp64: precise before the tile shift)
  float oz;
  float c1[3], c2[3];  // columns 1, 2 of R_map_sensor:  D_b = sin a_b * c1 - cos a_b * c2
  u32 slot;            // the particle's state slot: records may lie in VISITING order (mcl_kernels.h: VisitArgs), log-weights never
};

// Tile window of one group of MBES_WAVES consecutive particles, decided once per group by the pose kernel
// (one thread per particle, 8-lane shuffles) instead of by every wave of the cast kernel.
struct MbesGroup {   // 32 B
  int tx0, ty0, tw, th;  // window origin and size: nodes (grid / structured mesh)
  int fast;              // 1: k_mbes_fast casts this group; 0: it is on the worklist of the general kernel
  int why;               // diagnostics (MCL_DEBUG_WORK): 1 a fan that is not simple (own_fan), 4 window larger than the LDS tile
  int pad[2];
};

struct MeshArgs {
  const float4* tri;       // 3 float4 per (cell, triangle) record, plane form (mcl_mesh.h)
  const float4* tri_mt;    // Moller-Trumbore records for near-vertical triangles (or nullptr)
  const u32* cell_start;   // gx*gy + 1
  const uint2* cell_info;  // x = half2(zmin, zmax) conservative, y = start | count << 27
  int gx, gy;
  float cs;
  const uint4* tin_he;     // fan sweep over a TIN: one 32-byte record per half-edge {x, y, z of the opposite vertex, next_a | next_b, -, -, -} (mcl_mesh.h)
  u32 tin_he_bytes;        // ... its size (read through a raw buffer)
  const u32* cell_rim;     // ... per cell: first rim record of the linked hole that may lie under a sensor there (0xffffffff none, 0xfffffffe several), or nullptr
  u32 tin_outline;         // ... first rim record of the outline when it is linked, else 0
  u32 tin_nhe;             // ... its half-edge records (3 x triangles); behind them the rim records of the holes the walk crosses (mcl_halfedge.h)
  const float4* cell_tri;  // fan slice (mcl_slice.h): the three vertices (x, y, z, -), map frame, of every (cell, triangle) record, indexed like `tri`
  double x0, y0;           // map-frame position of cell (0, 0)'s corner
};

struct MbesArgs {
  const double* st[6];  // x,y,z,roll,pitch,yaw (odom frame)
  long long n;
  double m2o[12];       // rows 0..2 of map<-odom
  double off_t[3];      // sensor offset translation in base_link
  double off_R[9];      // sensor offset rotation
  MbesPose* pose;       // n records (scratch, written by k_mbes_pose)
  const float2* beam_sc;  // (sin a_b, cos a_b)
  const float* ranges;    // measured
  int n_beams;
  int sorted;             // 1: beam angles ascend with the beam index
  int b_lo, b_hi;         // indices of the extreme beam angles (span < pi), or -1: scan every beam for the footprint
  const float* grid;      // z[ix*ny + iy]
  const float* grid_pad;  // fan sweep (lattice maps): the same heights inside a one-node ring of NaNs, (nx + 2) x nyp, node (i, j) at [(i + 1) * nyp + j + 1]; the ring's payload says which border: 1 = an x side, 2 = a y side, 3 = a corner
  int nyp;                // its pitch: ny + 2
  int nx, ny;             // grid: nodes; mesh: cells + 1
  double ox, oy, inv_res;
  float res;
  float zmin_map, zmax_map;
  int cells;              // 1: the LDS tile holds per-CELL words (triangle-record meshes), windows are cell ranges
  int diag_mode;          // structured mesh: 1 = every cell split along 00-11, 2 = along 10-01, 0 = per-cell bit
  float inv_sigma, r_max;
  double lognorm;         // log(sigma sqrt(2 pi))
  double* lw;             // out: log-likelihood per particle
  u64* max_slots;         // out: running maximum of lw (ordered keys, MCL_MAX_SLOTS words), or nullptr
  float* exp_out;         // out (EXPECT_ONLY): expected ranges [(i-exp_first)*B + b]
  long long exp_first, exp_count;
  MeshArgs mesh;
  const u32* perm;        // visiting order: wave j of the cast kernels works on particle perm[j] (nullptr: j)
  struct MbesGroup* groups;  // one record per group of MBES_WAVES particles (written by the pose kernels)
  int* worklist;          // group ids deferred by the fast kernel (capacity = number of groups)
  int* work_count;        // device counter, zeroed before every fast launch
  unsigned long long* stats;  // MBES_STATS builds: steps, exact tests, rays, retries
  // fan sweep (mcl_sweep.h): regularly triangulated meshes, ascending beam angles
  const float4* sweep_beams;  // per beam: side-signed tan of the NEXT beam of its side (+inf beyond the last), w / cos a, z w, (z - r_max) w  (z measured range, w = 1 / sigma; invalid beam: zeros; expected-range calls: .y = 1 / cos a)
  const float* sweep_tail_run;   // the same sums per run of a side's beams (SUB kernels stage these instead)
  const float* sweep_tan0;    // 4 floats behind the table: side-signed tan of the first beam of the + / - side, then of the second one (+inf: no such beam) -- staged in LDS with the table (as kernel arguments, selected by side, clang copied them to scratch)
  const float* sweep_tail;    // per beam: sum of ((range - r_max) * weight)^2 over this beam and the ones beyond it on its side
  int b_split;                // first beam with a >= 0: beams [b_split, B) sweep outward on the + side, [0, b_split) on the - side
  int sweep_nvalid;           // beams with a valid measured range
  int sweep_nsub;             // lanes per particle side (1, 2 or 4: small clouds split a side's beams over several lanes)
  int sweep_noclamp;          // 1: no beam of this ping can meet the seabed beyond r_max (proved on the host): the merge loop skips the clamp
  float sweep_c2z_min;        // cos of the largest fan-plane tilt the sweep accepts (terrain slope bound, mcl_host_update.h)
  float sweep_slope;          // the map's steepest slope |grad h| (second pass: may a slice end at the map border?)
  u32* defer_idx;             // particles the sweep hands over: the visiting order (perm) of the cast kernels that follow it
  int* defer_count;           // device counter, zeroed with the control block
  const int* n_dev;           // when set, the classify / cast kernels visit *n_dev entries of perm instead of a.n
  int* host_count;            // pinned host word (or nullptr): k_mbes_cast<.,.,2> leaves the hand-over count there
  int slice;                  // 1: this update is cast by the fan slice (mcl_slice.h)
  const u32* in_list;         // fan slice as the HAND-OVER kernel of the TIN sweep (round 6): it casts the *in_count pose records in_list names
  const int* in_count;        //   (the sweep's hand-over list; nullptr: all a.n records) and hands what it declines on through defer_idx / defer_count
  // visiting order (mcl_kernels.h: VisitArgs) for the stand-alone pose kernel: when set, k_mbes_pose<false> stores the record
  // of slot i at its sorted position (the fused step's predict kernel does the same from its own arguments)
  const u32* visit_okey;
  const u32* visit_base;
  int visit_nb;
  u32* slice_loose;           // fan slice over groups of spatial neighbours (k_mbes_slice_group): the groups of SLICE_G records it leaves to k_mbes_slice (nullptr: k_mbes_slice casts everything)
  int* slice_loose_count;     // ... their number (device counter in the control block, zeroed with it)
  unsigned* reasons;          // SWEEP_REASONS builds: 16 counters, why the sweep declined a particle side (or nullptr)
};
__device__ __forceinline__ long long mbes_count(const MbesArgs& a) { return a.n_dev ? (long long)*a.n_dev : a.n; }

#ifdef MBES_STATS
#define STAT_INC(v) (++(v))
#else
#define STAT_INC(v) ((void)0)
#endif
struct RayStats {
  int steps, tests, rays, retries;
};
__device__ __forceinline__ float uniform_f32(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double uniform_f64(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// ------------------------------------------------------------------ pose record
// sensor pose in the map = m2o * T(x,y,z) R(rpy) * T_off R_off, origin in map cell units
struct PoseXform {
  double m2o[12];
  double off_t[3];
  double off_R[9];
  double ox, oy, inv_res;
};
__device__ __forceinline__ MbesPose make_pose(const PoseXform& T, double x, double y, double z, double sr, double cr,
                                              double sp, double cp, double sy, double cy) {
  const double Rp[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                        -sp,     cp * sr,                cp * cr};
  double Rmp[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rmp[r * 3 + c] = T.m2o[r * 4 + 0] * Rp[c] + T.m2o[r * 4 + 1] * Rp[3 + c] + T.m2o[r * 4 + 2] * Rp[6 + c];
  double o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = (T.m2o[r * 4 + 0] * x + T.m2o[r * 4 + 1] * y + T.m2o[r * 4 + 2] * z + T.m2o[r * 4 + 3]) +
           (Rmp[r * 3 + 0] * T.off_t[0] + Rmp[r * 3 + 1] * T.off_t[1] + Rmp[r * 3 + 2] * T.off_t[2]);
  MbesPose P;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    P.c1[r] = (float)(Rmp[r * 3 + 0] * T.off_R[1] + Rmp[r * 3 + 1] * T.off_R[4] + Rmp[r * 3 + 2] * T.off_R[7]);
    P.c2[r] = (float)(Rmp[r * 3 + 0] * T.off_R[2] + Rmp[r * 3 + 1] * T.off_R[5] + Rmp[r * 3 + 2] * T.off_R[8]);
  }
  P.um = (o[0] - T.ox) * T.inv_res;
  P.vm = (o[1] - T.oy) * T.inv_res;
  P.oz = (float)o[2];
  P.slot = 0u;
  return P;
}
__device__ __forceinline__ PoseXform pose_xform(const MbesArgs& a) {
  PoseXform T;
#pragma unroll
  for (int k = 0; k < 12; ++k) T.m2o[k] = a.m2o[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) T.off_t[k] = a.off_t[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) T.off_R[k] = a.off_R[k];
  T.ox = a.ox;
  T.oy = a.oy;
  T.inv_res = a.inv_res;
  return T;
}
// ---- what a particle's own fan needs (the determinism rule above): its cell, the fraction inside it, and the node
// (cell) window its fan covers.  The footprint of a fan is bounded by its two extreme-angle beams followed down to
// z_min(map) (planar fan: the end points of all beams at that depth are collinear and ordered by angle).  `simple`:
// both of them point downward and reach z_min inside r_max, and the window -- one node of margin -- is not clipped by
// the map border: the clearance walk (cast_clear / cast_fast) then needs no bounds test, and the sensor lies inside
// the window with a cell of margin by construction.
struct OwnFan {
  int I0, J0;              // the sensor's cell (floor of its position in cell units)
  float ul, vl;            // position inside that cell, [0, 1)
  int wx0, wy0, wx1, wy1;  // window: nodes (grid / structured mesh) or cells (a.cells), inclusive
  bool sane;               // finite position within +-1e9 cells
  bool simple;
};
__device__ __forceinline__ OwnFan own_fan(const MbesArgs& a, const MbesPose& P) {
  OwnFan F;
  F.sane = fabs(P.um) < 1e9 && fabs(P.vm) < 1e9;  // (NaN: false)
  const double fu = F.sane ? floor(P.um) : 0.0, fv = F.sane ? floor(P.vm) : 0.0;
  F.I0 = (int)fu;
  F.J0 = (int)fv;
  F.ul = F.sane ? (float)(P.um - fu) : 0.f;
  F.vl = F.sane ? (float)(P.vm - fv) : 0.f;
  const float inv_res = (float)a.inv_res;
  float umin = F.ul, umax = F.ul, vmin = F.vl, vmax = F.vl;
  bool simple = F.sane && a.b_lo >= 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float2 sc = a.beam_sc[k == 0 ? max(a.b_lo, 0) : max(a.b_hi, 0)];
    const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
    const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
    const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
    const float t_end = fmaxf((a.zmin_map - P.oz) * __builtin_amdgcn_rcpf(dz), 0.f);
    simple = simple & (dz < -1e-4f) & (t_end <= a.r_max);
    const float ue = F.ul + t_end * dx * inv_res, ve = F.vl + t_end * dy * inv_res;
    umin = fminf(umin, ue);
    umax = fmaxf(umax, ue);
    vmin = fminf(vmin, ve);
    vmax = fmaxf(vmax, ve);
  }
  // (a fan that is not simple has no finite footprint: its window is never used)
  simple = simple & (fmaxf(fmaxf(-umin, umax), fmaxf(-vmin, vmax)) < 1e6f);
  const int cz = a.cells ? 1 : 0;
  F.wx0 = F.I0 + (simple ? (int)floorf(umin) : 0) - 1;
  F.wy0 = F.J0 + (simple ? (int)floorf(vmin) : 0) - 1;
  F.wx1 = F.I0 + (simple ? (int)floorf(umax) : 0) + 2 - cz;
  F.wy1 = F.J0 + (simple ? (int)floorf(vmax) : 0) + 2 - cz;
  const int lim_x = a.nx - 1 - cz, lim_y = a.ny - 1 - cz;
  F.simple = simple && F.wx0 >= 0 && F.wy0 >= 0 && F.wx1 <= lim_x && F.wy1 <= lim_y;
  return F;
}

// Group classification for the fast cast kernel (height grids, structured meshes, cell words of triangle records).
// Called by every lane of a wave with the pose record of "its" particle (lanes 8k..8k+7 = one group).  A group is FAST
// when every fan in it is simple (own_fan) and their common window fits the LDS tile; everything else goes on the
// worklist of the general kernel -- which casts a simple particle with the same arithmetic from global memory, so the
// grouping decides where the heights are read from and nothing else.
__device__ __forceinline__ void classify_group(const MbesArgs& a, const MbesPose& P, bool valid, long long i) {
  const int lane = threadIdx.x & 63;
  const OwnFan F = own_fan(a, P);
  const int BIG = 0x3fffffff;
  int x0 = (valid && F.simple) ? F.wx0 : BIG, y0 = (valid && F.simple) ? F.wy0 : BIG;
  int x1 = (valid && F.simple) ? F.wx1 : -BIG, y1 = (valid && F.simple) ? F.wy1 : -BIG;
#pragma unroll
  for (int o = 1; o < MBES_WAVES; o <<= 1) {
    x0 = min(x0, __shfl_xor(x0, o, 64));
    y0 = min(y0, __shfl_xor(y0, o, 64));
    x1 = max(x1, __shfl_xor(x1, o, 64));
    y1 = max(y1, __shfl_xor(y1, o, 64));
  }
  const int cz = a.cells ? 1 : 0;
  const int tw = x1 - x0 + 1, th = y1 - y0 + 1;
  const bool fits = x0 <= x1 && tw >= 2 - cz && th >= 2 - cz && (long long)tw * th <= (cz ? (MBES_TILE_FLOATS * 3) / 4 : MBES_TILE_FLOATS);
  const unsigned long long okm = __ballot(!valid || F.simple);
  const unsigned gm = (1u << MBES_WAVES) - 1u;
  const unsigned grp_bits = (unsigned)(okm >> (lane & ~(MBES_WAVES - 1))) & gm;
  const bool fast = fits && grp_bits == gm;
  const bool leader = (lane & (MBES_WAVES - 1)) == 0 && valid;
  if (leader) {
    MbesGroup G;
    G.tx0 = x0;
    G.ty0 = y0;
    G.tw = tw;
    G.th = th;
    G.fast = fast ? 1 : 0;
    G.why = (grp_bits != gm ? 1 : 0) | (fits ? 0 : 4);   // diagnostics: 1 a fan that is not simple, 4 window larger than the LDS tile
    G.pad[0] = G.pad[1] = 0;
    a.groups[i / MBES_WAVES] = G;
  }
  // one atomic per wave: the leaders of its deferred groups take consecutive worklist slots (the ORDER of the
  // worklist only schedules the general kernel: results do not depend on it)
  const unsigned long long dm = __ballot(leader && !fast);
  if (dm) {
    int base = 0;
    if (lane == 0) base = atomicAdd(a.work_count, (int)__popcll(dm));
    base = __builtin_amdgcn_readfirstlane(base);
    if (leader && !fast) a.worklist[base + (int)__popcll(dm & ((1ull << lane) - 1ull))] = (int)(i / MBES_WAVES);
  }
}

// stand-alone pose kernel (mcl_update_mbes / mcl_mbes_expected on an arbitrary state)
template <bool CLASSIFY>
__global__ void __launch_bounds__(256) k_mbes_pose(MbesArgs a) {
  const PoseXform T = pose_xform(a);
  const long long n_pad = (a.n + 63) & ~63ll;  // whole waves take part in the group shuffles
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_pad;
       i += (long long)gridDim.x * blockDim.x) {
    const bool valid = i < a.n;
    MbesPose P;
    P.um = P.vm = 0.0;
    P.oz = 0.f;
    if (valid) {
      double sr, cr, sp, cp, sy, cy;
      sincos(a.st[3][i], &sr, &cr);
      sincos(a.st[4][i], &sp, &cp);
      sincos(a.st[5][i], &sy, &cy);
      P = make_pose(T, a.st[0][i], a.st[1][i], a.st[2][i], sr, cr, sp, cp, sy, cy);
      P.slot = (u32)i;
      u32 pos = (u32)i;
      if (!CLASSIFY && a.visit_okey) {   // the visiting order the last resample prepared (separate predict / update / resample calls)
        const u32 ok = a.visit_okey[i], key = ok & ((1u << VISIT_KEY_BITS) - 1u);
        const u32 owner = ((u32)i >> VISIT_OWNER_SHIFT) & VISIT_OWNER_MASK;
        pos = a.visit_base[(size_t)owner * a.visit_nb + key] + (ok >> VISIT_KEY_BITS);
      }
      a.pose[pos] = P;
    }
    if (CLASSIFY) classify_group(a, P, valid, i);
  }
}
// motion_pred (mcl_kernels.h:k_predict) and the pose record of the state it has just written, in one pass
// (the fused step: the measurement update that follows would re-read all six components).  Same arithmetic
// as k_predict followed by k_mbes_pose, bit for bit: roll and pitch are the odometry's for every particle.
template <bool CLASSIFY>
__global__ void __launch_bounds__(MCL_BLOCK) k_predict_pose(StatePtrs s, long long n, PredictArgs a, MbesArgs m) {
  const PoseXform T = pose_xform(m);
  const long long n_pad = (n + 63) & ~63ll;
  if (!CLASSIFY && a.zero_ptr && blockIdx.x == 0)   // (this kernel does not touch the block; the update after it does)
    for (int k = threadIdx.x; k < a.zero_words; k += blockDim.x) a.zero_ptr[k] = 0ull;
  // where a slot's record goes: its place in the visiting order the last gather prepared -- okey[slot] -> base[owner][key]
  // + rank, two DEPENDENT loads.  (Round 5, first version: both were waited for at the top of every iteration, two exposed
  // memory latencies per particle with three waves per SIMD to hide them: +7 us.)  The key word of the NEXT iteration is
  // loaded during this one's arithmetic, the table word is requested at the top and only added at the store.
  const bool visit = !CLASSIFY && a.visit_okey != nullptr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  u32 ok_next = (visit && i < n) ? a.visit_okey[i] : 0u;
  double sr, cr, sp, cp;
  sincos(a.roll, &sr, &cr);
  sincos(a.pitch, &sp, &cp);
  for (; i < n_pad; i += stride) {
    const bool valid = i < n;
    MbesPose P;
    P.um = P.vm = 0.0;
    P.oz = 0.f;
    if (valid) {
      const u32 ok = ok_next;
      u32 bv = 0u;
      if (visit) {
        const u32 key = ok & ((1u << VISIT_KEY_BITS) - 1u), owner = ((u32)i >> VISIT_OWNER_SHIFT) & VISIT_OWNER_MASK;
        bv = a.visit_base[(size_t)owner * a.visit_nb + key];
        if (i + stride < n) ok_next = a.visit_okey[i + stride];
      }
      // (the yaw too -- the first state word the arithmetic needs: requested before the ~500 instructions of Philox /
      //  Box-Muller, not 90 instructions before its use; x and y as well would cost the third wave per SIMD, 170 VGPRs)
      const double w0 = s.c[5][i];
      u32x4 o = philox4x32((u32)(a.nz.gid0 + i), 0u, a.nz.step, 1u, a.nz.k0, a.nz.k1);
      double n0, n1, n5, unused;
      box_muller(o.x, o.y, n0, n1);
      box_muller(o.z, o.w, n5, unused);
      const double yaw_t = wrap_pi(w0 + a.wzdt + a.nz.sq[5] * n5);
      double sy, cy;
      sincos(yaw_t, &sy, &cy);
      const double x = s.c[0][i] + ((cy * a.m0 - sy * a.m1) + a.nz.sq[0] * n0);
      const double y = s.c[1][i] + ((sy * a.m0 + cy * a.m1) + a.nz.sq[1] * n1);
      s.c[0][i] = x;
      s.c[1][i] = y;
      if (!a.skip_uniform) {
        s.c[2][i] = a.z;
        s.c[3][i] = a.roll;
        s.c[4][i] = a.pitch;
      }
      s.c[5][i] = yaw_t;
      P = make_pose(T, x, y, a.z, sr, cr, sp, cp, sy, cy);
      P.slot = (u32)i;
      m.pose[visit ? bv + (ok >> VISIT_KEY_BITS) : (u32)i] = P;
    }
    if (CLASSIFY) classify_group(m, P, valid, i);
  }
}

// ---- spatially coherent visiting order (dispersed clouds: global localisation, before the filter converges)
// A group of MBES_WAVES particles shares one LDS tile only if their fans are neighbours.  When the natural
// (slot) order leaves many groups without a common tile, the particles are VISITED in the order of a Morton
// key of their map cell under a bin of the fan direction; state slots are untouched, so keep/lost/dupes and
// every RNG draw (keyed by the slot's global id) are unaffected.
__device__ __forceinline__ u32 morton_spread10(u32 v) {
  v &= 0x3ffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__global__ void __launch_bounds__(256) k_mbes_keys(MbesArgs a, u32* __restrict__ keys, u32* __restrict__ idx) {
  // finest block size for which the map fits 1024 x 1024 blocks (1 cell up to 1023 cells a side)
  int sh = 0;
  while (((a.nx > a.ny ? a.nx : a.ny) >> sh) > 1023) ++sh;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    const MbesPose P = a.pose[i];
    u32 key = 0xffffffu;  // off the map (or NaN): visited last
    if (P.um >= 0.0 && P.um < (double)a.nx && P.vm >= 0.0 && P.vm < (double)a.ny) {
      // MBES_DIR_BINS bins of the fan's direction in the map (a line: modulo pi) above the Morton code of the
      // cell, so that a group's fans are parallel as well as close: their common window stays a narrow strip
      const float ang = atan2f(P.c1[1], P.c1[0]);  // (-pi, pi]
      int bin = (int)floorf((ang < 0.f ? ang + 3.14159265f : ang) * ((float)MBES_DIR_BINS / 3.14159265f));
      bin = bin < 0 ? 0 : (bin > MBES_DIR_BINS - 1 ? MBES_DIR_BINS - 1 : bin);
      key = ((u32)bin << 20) | morton_spread10((u32)P.um >> sh) | (morton_spread10((u32)P.vm >> sh) << 1);
    }
    keys[i] = key;
    idx[i] = (u32)i;
  }
}
// group records and worklist for the visiting order a.perm (same decision as in the pose kernels)
__global__ void __launch_bounds__(256) k_mbes_classify(MbesArgs a) {
  const long long n = mbes_count(a);
  if (a.host_count && blockIdx.x == 0 && threadIdx.x == 0) *a.host_count = (int)n;  // (read by the host one update late)
  const long long n_pad = (n + 63) & ~63ll;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n_pad;
       j += (long long)gridDim.x * blockDim.x) {
    const bool valid = j < n;
    MbesPose P;
    P.um = P.vm = 0.0;
    P.oz = 0.f;
    if (valid) P = a.pose[a.perm[j]];
    classify_group(a, P, valid, j);
  }
}

// ------------------------------------------------------------------ per-cell exact tests (triangle records)
// nearest hit of the ray with the records [s, e) of one cell; ray origin relative to that cell's
// corner (metres).  Accepts t in [0, t_hi].  Plane-form records: t from the plane equation, then
// the barycentrics of the hit point in the xy projection (~26 VALU per triangle); near-vertical
// triangles (flag) fall back to Moller-Trumbore on the second record array.
__device__ __forceinline__ float records_hit(const MeshArgs& ma, u32 s, u32 e, float olx, float oly, float olz,
                                             float dx, float dy, float dz, float t_hi) {
  float best = __builtin_inff();
  const float EPS = 2e-5f;
  for (u32 k = s; k < e; ++k) {
    const float4 r0 = ma.tri[3 * (size_t)k], r1 = ma.tri[3 * (size_t)k + 1];
    const float2 r2 = *(const float2*)&ma.tri[3 * (size_t)k + 2];
    float t, u, v;
    if (r0.w == 0.f) {
      const float den = fmaf(r0.x, dx, fmaf(r0.y, dy, dz));
      t = (r0.z - fmaf(r0.x, olx, fmaf(r0.y, oly, olz))) * fast_rcp(den);
      const float hx = fmaf(t, dx, olx) - r1.x, hy = fmaf(t, dy, oly) - r1.y;
      u = fmaf(r1.z, hx, r1.w * hy);
      v = fmaf(r2.x, hx, r2.y * hy);
    } else {
      const float4 v0 = ma.tri_mt[3 * (size_t)k], e1 = ma.tri_mt[3 * (size_t)k + 1], e2 = ma.tri_mt[3 * (size_t)k + 2];
      const float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
      const float det = e1.x * px + e1.y * py + e1.z * pz;
      if (fabsf(det) < 1e-20f) continue;
      const float inv = fast_rcp(det);
      const float sx = olx - v0.x, sy = oly - v0.y, sz = olz - v0.z;
      u = (sx * px + sy * py + sz * pz) * inv;
      const float qx = sy * e1.z - sz * e1.y, qy = sz * e1.x - sx * e1.z, qz = sx * e1.y - sy * e1.x;
      v = (dx * qx + dy * qy + dz * qz) * inv;
      t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
    }
    if (u >= -EPS && v >= -EPS && u + v <= 1.f + EPS && t >= 0.f && t <= t_hi && t < best) best = t;
  }
  return best;
}
__device__ __forceinline__ float cell_triangles_hit(const MeshArgs& ma, int gix, int giy, float olx, float oly,
                                                    float olz, float dx, float dy, float dz, float t_hi) {
  const size_t c = (size_t)gix * ma.gy + giy;
  return records_hit(ma, ma.cell_start[c], ma.cell_start[c + 1], olx, oly, olz, dx, dy, dz, t_hi);
}
// unpack a cell_info word pair
__device__ __forceinline__ void cell_zrange(u32 hz, float& zlo, float& zhi) {
  zlo = __half2float(__ushort_as_half((unsigned short)(hz & 0xffffu)));
  zhi = __half2float(__ushort_as_half((unsigned short)(hz >> 16)));
}

// ------------------------------------------------------------------ the ray traversal (triangle records)
// The clipped general march over the WHOLE cell grid of a triangle-record mesh in global memory (L2): sensors off the
// map or at its border, fans that reach beyond it, windows larger than LDS.  (Lattice maps -- height grids, structured
// meshes -- take cast_clear<SURF, true> on the NaN-ringed height array instead.)
// `map`: (zmin, zmax | record range) words per cell, pitch in cells, cw x ch cells.  Coordinates are relative to the
// particle's own cell (bi, bj) -- u0, v0 in [0, 1), cell indices ix, iy relative to it: the arithmetic does not depend
// on where the map's origin is.
__device__ __forceinline__ float cast_ray(const uint2* __restrict__ map, int pitch, const MbesArgs& a, int bi, int bj,
                                          int cw, int ch, float u0, float v0, float oz, float du, float dv, float dx,
                                          float dy, float dz, float t_lo, float r_max, RayStats& rs) {
  STAT_INC(rs.rays);
  float t0 = t_lo, t1 = r_max;
  const float inv_du = du != 0.f ? fast_rcp(du) : 0.f, inv_dv = dv != 0.f ? fast_rcp(dv) : 0.f;
  // clip to the map: [-bi, cw - bi] x [-bj, ch - bj] in these coordinates (small integers: exact in fp32)
  const int ix_lo = -bi, ix_hi = cw - 1 - bi, iy_lo = -bj, iy_hi = ch - 1 - bj;
  const float ux0 = (float)ix_lo, ux1 = (float)(cw - bi), vy0 = (float)iy_lo, vy1 = (float)(ch - bj);
  if (du == 0.f) {
    if (u0 < ux0 || u0 > ux1) return r_max;
  } else {
    const float ta = (ux0 - u0) * inv_du, tb = (ux1 - u0) * inv_du;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  if (dv == 0.f) {
    if (v0 < vy0 || v0 > vy1) return r_max;
  } else {
    const float ta = (vy0 - v0) * inv_dv, tb = (vy1 - v0) * inv_dv;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  if (!(t0 <= t1)) return r_max;
  const float pu = u0 + t0 * du, pv = v0 + t0 * dv;
  int ix = min(max((int)floorf(pu), ix_lo), ix_hi);
  int iy = min(max((int)floorf(pv), iy_lo), iy_hi);
  if (du < 0.f && ix > ix_lo && (float)ix >= pu) --ix;
  if (dv < 0.f && iy > iy_lo && (float)iy >= pv) --iy;
  const float INF = __builtin_inff();
  const int sx = du > 0.f ? 1 : -1, sy = dv > 0.f ? 1 : -1;
  const float dtx = fabsf(inv_du), dty = fabsf(inv_dv);
  float tnx = du != 0.f ? ((float)(ix + (du > 0.f ? 1 : 0)) - u0) * inv_du : INF;
  float tny = dv != 0.f ? ((float)(iy + (dv > 0.f ? 1 : 0)) - v0) * inv_dv : INF;
  float z_in = oz + t0 * dz;
  float result = r_max;
  int guard = cw + ch + 4;
  for (;;) {
    // ---- phase 1: walk cells until one might contain the surface (z-range reject test only)
    bool cand = false;
    float t_out = t1, z_out = z_in;
    while (guard > 0) {
      --guard;
      STAT_INC(rs.steps);
      t_out = fminf(fminf(tnx, tny), t1);
      z_out = oz + t_out * dz;
      const uint2 ci = map[(size_t)(ix + bi) * pitch + (iy + bj)];
      float czlo, czhi;
      cell_zrange(ci.x, czlo, czhi);
      cand = fminf(z_in, z_out) <= czhi + 1e-4f && fmaxf(z_in, z_out) >= czlo - 1e-4f;
      if (cand) break;
      if (t_out >= t1) break;
      const bool stepx = tnx <= tny;
      ix += stepx ? sx : 0;
      iy += stepx ? 0 : sy;
      if ((unsigned)(ix + bi) >= (unsigned)cw || (unsigned)(iy + bj) >= (unsigned)ch) {  // fp32 slop at the map edge
        guard = 0;
        break;
      }
      tnx += stepx ? dtx : 0.f;
      tny += stepx ? 0.f : dty;
      z_in = z_out;
    }
    if (!cand) break;
    // ---- phase 2: exact test in the candidate cell (lanes reconverge here)
    STAT_INC(rs.tests);
    const float t = cell_triangles_hit(a.mesh, ix + bi, iy + bj, (u0 - (float)ix) * a.mesh.cs,
                                       (v0 - (float)iy) * a.mesh.cs, oz, dx, dy, dz, t_out + 1e-4f);
    if (t < INF) {
      result = fminf(t, r_max);
      break;
    }
    if (t_out >= t1 || guard <= 0) break;
    const bool stepx = tnx <= tny;
    ix += stepx ? sx : 0;
    iy += stepx ? 0 : sy;
    if ((unsigned)(ix + bi) >= (unsigned)cw || (unsigned)(iy + bj) >= (unsigned)ch) break;
    tnx += stepx ? dtx : 0.f;
    tny += stepx ? 0.f : dty;
    z_in = z_out;
  }
  return result;
}

// ------------------------------------------------------------------ fast traversal (LDS tile)
// Preconditions (own_fan::simple): the fan's window was not clipped by the map border.  The footprint
// construction then guarantees the ray stays inside it until it is below every node (t1), so no window
// clipping and no per-step bounds tests are needed.  ~35 VALU of set-up, ~16 per cell, ~45 per
// exact patch test (both roots of the patch quadratic at once: crossing and grazing cases).
// `tile` points at the word of the particle's OWN cell (tx0, ty0) -- in an LDS tile or in the global array alike --,
// pitch th; u0, v0 in [0, 1) and the cell indices are relative to that cell (the determinism rule).
template <int MAP>
__device__ __forceinline__ float cast_fast(const float* __restrict__ tile, int th, const MbesArgs& a, int tx0,
                                           int ty0, float u0, float v0, float oz, float du, float dv, float dz,
                                           float zmax, float r_max, RayStats& rs) {
  STAT_INC(rs.rays);
  const float INF = __builtin_inff();
  const float rdz = fast_rcp(dz);
  float t_lo = 0.f, t1 = r_max;
  if (dz < 0.f) {
    if (oz > zmax) t_lo = fmaxf((zmax - oz) * rdz - 1e-3f, 0.f);  // skip the water column above the tile
    t1 = fminf(r_max, (a.zmin_map - oz) * rdz + 1e-2f);           // below every node beyond this
  }
  const float pu = fmaf(t_lo, du, u0), pv = fmaf(t_lo, dv, v0);
  const float fu = floorf(pu), fv = floorf(pv);
  int ix = (int)fu, iy = (int)fv;
  const float adu = fabsf(fast_rcp(du)), adv = fabsf(fast_rcp(dv));  // +inf for an axis-parallel ray
  // distance to the next cell border along each axis.  An axis-parallel ray that starts exactly on a
  // grid line gives 0 * inf = NaN: minNum(NaN, inf) = inf turns it into "never crosses" (a NaN here would
  // fail every `tnx <= tny` test and march the other axis forever)
  float tnx = fminf(fmaf(du > 0.f ? (fu + 1.f) - pu : pu - fu, adu, t_lo), INF);
  float tny = fminf(fmaf(dv > 0.f ? (fv + 1.f) - pv : pv - fv, adv, t_lo), INF);
  const int sx = du > 0.f ? 1 : -1, sy = dv > 0.f ? 1 : -1;
  const int dax = sx * th;
  int addr = ix * th + iy;
  float t_in = t_lo, z_in = fmaf(t_lo, dz, oz);
  bool first = !(t_lo > 0.f);  // only a march that starts at the sensor can begin under the seabed
  for (int guard = 0; guard < 4096; ++guard) {
    float t_out, z_out;
    bool cand;
    float h00 = 0.f, h10 = 0.f, h01 = 0.f, h11 = 0.f;
    uint2 ci = make_uint2(0u, 0u);
    // ---- phase 1: cells whose LDS bound the ray does not reach are skipped
    for (;;) {
      STAT_INC(rs.steps);
      t_out = fminf(fminf(tnx, tny), t1);
      z_out = fmaf(t_out, dz, oz);
      const float zlo = fminf(z_in, z_out);
      if (MAP != 1) {
        const float* p = tile + addr;
        h00 = p[0];
        h01 = p[1];
        h10 = p[th];
        h11 = p[th + 1];
        cand = zlo <= fmaxf(fmaxf(h00, h10), fmaxf(h01, h11));
      } else {
        ci = ((const uint2*)tile)[addr];
        float czlo, czhi;
        cell_zrange(ci.x, czlo, czhi);
        cand = zlo <= czhi + 1e-4f && fmaxf(z_in, z_out) >= czlo - 1e-4f;
      }
      if (cand || t_out >= t1) break;
      const bool stepx = tnx <= tny;
      addr += stepx ? dax : sy;
      ix += stepx ? sx : 0;
      iy += stepx ? 0 : sy;
      tnx += stepx ? adu : 0.f;
      tny += stepx ? 0.f : adv;
      t_in = t_out;
      z_in = z_out;
    }
    if (!cand) return r_max;
    // ---- phase 2: exact test (lanes reconverge here)
    STAT_INC(rs.tests);
    if (MAP == 0) {
      const float uc = u0 - (float)ix, vc = v0 - (float)iy;
      const float B = h10 - h00, C = h01 - h00, D = (h00 - h10) - (h01 - h11);
      const float c0 = oz - (h00 + B * uc + C * vc + D * uc * vc);
      const float c1 = dz - (B * du + C * dv + D * (uc * dv + vc * du));
      const float c2 = -D * du * dv;
      if (first) {
        if (c0 + t_in * (c1 + t_in * c2) <= 0.f) return t_in;  // sensor at or below the seabed
        first = false;
      }
      const float disc = fmaf(c1, c1, -4.f * c2 * c0);
      if (disc >= 0.f) {
        const float sq = fast_sqrt(disc);
        const float qv = -0.5f * (c1 + (c1 >= 0.f ? sq : -sq));
        const float r1 = c0 * fast_rcp(qv), r2 = qv * fast_rcp(c2);  // r2 = inf/NaN on a planar patch
        const float lo = t_in - 1e-4f, hi = t_out + 1e-4f;
        const float g1 = (r1 >= lo && r1 <= hi) ? r1 : INF, g2 = (r2 >= lo && r2 <= hi) ? r2 : INF;
        const float root = fminf(g1, g2);
        if (root < INF) return fminf(fminf(fmaxf(root, t_in), t_out), r_max);
      }
    } else if (MAP == 2) {
      // structured mesh: the cell's two triangles are two planes through its corner heights; bit 0
      // of h00 selects the diagonal.  00-11 split: A = (00,10,11) where v <= u, B = (00,11,01) where
      // v >= u.  10-01 split: C = (00,10,01) where u+v <= 1, D = (10,11,01) where u+v >= 1.
      first = false;
      const bool d1 = (__float_as_uint(h00) & 1u) != 0u;
      const float uc = u0 - (float)ix, vc = v0 - (float)iy;
      const float a1 = h10 - h00, b1 = d1 ? h01 - h00 : h11 - h10;           // first triangle (A or C)
      const float a2 = h11 - h01, b2 = d1 ? h11 - h10 : h01 - h00;           // second triangle (B or D)
      const float c2 = d1 ? (h10 - h11) + h01 : h00;
      const float lo = t_in - 1e-4f, hi = t_out + 1e-4f;
      const float t1p = ((h00 + a1 * uc + b1 * vc) - oz) * fast_rcp(dz - a1 * du - b1 * dv);
      const float t2p = ((c2 + a2 * uc + b2 * vc) - oz) * fast_rcp(dz - a2 * du - b2 * dv);
      // side of the diagonal at each candidate point: s = v - u (00-11) or u + v - 1 (10-01)
      const float su = d1 ? 1.f : -1.f, s0 = d1 ? -1.f : 0.f;
      const float s1 = (vc + t1p * dv) + su * (uc + t1p * du) + s0;
      const float s2 = (vc + t2p * dv) + su * (uc + t2p * du) + s0;
      const float g1 = (t1p >= lo && t1p <= hi && s1 <= 2e-5f) ? t1p : INF;
      const float g2 = (t2p >= lo && t2p <= hi && s2 >= -2e-5f) ? t2p : INF;
      const float root = fminf(g1, g2);
      if (root < INF) return fminf(fmaxf(root, 0.f), r_max);
    } else {
      first = false;
      // the record range rides in the LDS tile word (start | count << 27): no dependent global load
      const u32 rs0 = ci.y & 0x7ffffffu;
      u32 cnt = ci.y >> 27;
      if (cnt == 31u) {
        if ((unsigned)(tx0 + ix) >= (unsigned)a.mesh.gx || (unsigned)(ty0 + iy) >= (unsigned)a.mesh.gy) return r_max;
        const size_t c = (size_t)(tx0 + ix) * a.mesh.gy + (ty0 + iy);
        cnt = a.mesh.cell_start[c + 1] - a.mesh.cell_start[c];
      }
      const float t = records_hit(a.mesh, rs0, rs0 + cnt, (u0 - (float)ix) * a.mesh.cs, (v0 - (float)iy) * a.mesh.cs, oz,
                                  du * a.res, dv * a.res, dz, t_out + 1e-4f);
      if (t < INF) return fminf(t, r_max);
    }
    if (t_out >= t1) return r_max;
    const bool stepx = tnx <= tny;
    addr += stepx ? dax : sy;
    ix += stepx ? sx : 0;
    iy += stepx ? 0 : sy;
    tnx += stepx ? adu : 0.f;
    tny += stepx ? 0.f : adv;
    t_in = t_out;
    z_in = z_out;
  }
  return r_max;
}

// ------------------------------------------------------------------ clearance traversal (LDS tile or global array)
// Same preconditions and conventions as cast_fast: `tile` points at the node (I0, J0) of the particle's own cell, u0, v0
// are the fractions inside it, zmax is the MAP's highest point.  ONE loop, no separate exact test: the ray is followed cell by cell in
// its own "forward" frame (a', b' grow along the ray in both axes, F00 is the corner it enters by, F11 the
// one it leaves by), carrying the clearance g = z_ray - h at the cell border it has just crossed.  Heights
// along cell EDGES are linear for bilinear patches and for either triangulation, so g at the exit point is
// two corner reads and a lerp, and g at the entry of the next cell is the same number (continuity).
//   SURF 1..3 (a cell = two triangles): along the ray h is piecewise linear with one kink where the ray
//     crosses the cell's diagonal, so the first hit is the first sign change of g over
//     {entry, diagonal crossing, exit} and the range follows by linear interpolation -- exact.
//     SURF 1: the diagonal of a cell is the LSB of its (0,0) corner height (mixed meshes);
//     SURF 2 / 3: every cell is split along 00-11 / 10-01 (no bit to read, heights untouched).
//   SURF 0 (bilinear patch): g is a quadratic in t between the borders, g(tau) = g_in + B tau + K tau^2 with
//     K = -(twist) cu cv dt^2: a sign change at the exit, or both ends above and a real root inside (grazing).
// ~41 VALU + 4 LDS reads per cell for the triangulated surfaces, against 25 per cell + 76 per exact
// two-plane test before; the wave executes max-over-lanes CELLS only, there is no second divergent phase.
// RING (the general kernel, lattice maps): `tile` is the height array inside its one-node ring of NaNs
//   (MbesArgs::grid_pad) and nothing is assumed about the fan: the ray may start outside the map -- it is over the
//   map's rectangle for t in [t_min, t_exit], computed by the caller; the start cell is clamped to the map's cells [fu_lo, fu_hi] x [fv_lo, fv_hi] (relative to
//   the particle's cell) -- and may leave it: the first cell beyond the border has a NaN exit corner, the clearance
//   there is NaN and the march ends with r_max (a mesh has no side walls; a ray from inside a grid never re-enters
//   it).  A grid is solid below its surface: a ray that enters the map from the side under the seabed hits at its
//   entry.  For a fan that IS simple (t_min = 0, every cell inside the map) the arithmetic is that of the plain
//   version, bit for bit.
template <int SURF, bool RING = false>
__device__ __forceinline__ float cast_clear(const float* __restrict__ tile, int th, const MbesArgs& a, float u0, float v0,
                                            float oz, float du, float dv, float dz, float zmax, float r_max,
                                            float t_min = 0.f, float t_exit = 0.f, float fu_lo = 0.f, float fu_hi = 0.f,
                                            float fv_lo = 0.f, float fv_hi = 0.f) {
  const float INF = __builtin_inff();
  const float rdz = fast_rcp(dz);
  float t_lo = 0.f, t1 = r_max;
  if (dz < 0.f) {
    if (oz > zmax) t_lo = fmaxf((zmax - oz) * rdz - 1e-3f, 0.f);  // skip the water column above the map
    t1 = fminf(r_max, (a.zmin_map - oz) * rdz + 1e-2f);           // below every node beyond this
  }
  if (RING) {
    t_lo = fmaxf(t_lo, t_min);
    if (!(t_lo <= t_exit)) return r_max;   // the ray is over the map's edge before it comes down to the highest node
  }
  const bool px = du > 0.f, py = dv > 0.f;
  const float cu = fabsf(du), cv = fabsf(dv);                        // cells per metre along each axis
  // metres per cell; an axis-parallel ray gets a huge FINITE value: 0 * 1e30 = 0, so nothing below can turn into
  // a NaN (0 * inf marched the other axis forever) and a vertical ray simply "leaves" its cell at
  // t ~ 1e30, where the linear interpolation of the clearance still returns g_in / |dz|
  const float iu = fminf(fabsf(fast_rcp(du)), 1e30f), iv = fminf(fabsf(fast_rcp(dv)), 1e30f);
  float fu = floorf(fmaf(t_lo, du, u0)), fv = floorf(fmaf(t_lo, dv, v0));
  if (RING) {  // (rounding at the map's edge must not put the start cell into the ring)
    fu = fminf(fmaxf(fu, fu_lo), fu_hi);
    fv = fminf(fmaxf(fv, fv_lo), fv_hi);
  }
  // forward-frame coordinates of the ray at parameter t inside the current cell: a' = A0 + t cu, b' = B0 + t cv
  float A0 = px ? u0 - fu : (fu + 1.f) - u0;
  float B0 = py ? v0 - fv : (fv + 1.f) - v0;
  // parameter at which the ray reaches a' = 1 / b' = 1
  float tnx = (1.f - A0) * iu, tny = (1.f - B0) * iv;
  // LDS BYTE offsets: F00 = entry corner, F10 / F01 one node on along x / y, F11 the exit corner
  const char* base = (const char*)tile;
  const int oxb = (px ? th : -th) * 4, oyb = py ? 4 : -4, oxyb = oxb + oyb;
  int a00 = (__mul24((int)fu + (px ? 0 : 1), th) + (int)fv + (py ? 0 : 1)) * 4;  // tile indices fit 24 bits
#define TILE_AT(off) (*(const float*)(base + (off)))
  // diagonal of a cell in the forward frame: mirroring ONE axis turns a 00-11 split into a 10-01 split
  const bool flip = px != py;
  bool fd = SURF == 2 ? flip : !flip;       // true: the cell is split along F10-F01 (SURF 1: re-read per cell)
  const int kcb = (px ? 0 : -th * 4) + (py ? 0 : -4);  // SURF 1: the canonical (0,0) corner relative to F00
  if (SURF == 1) fd = ((__float_as_uint(TILE_AT(a00 + kcb)) & 1u) != 0u) != flip;
  // The march starts on the border by which the ray ENTERED the start cell (the later of its two previous
  // crossings): the clearance there is an edge lerp like every other one.  Only when that point lies behind
  // the sensor (the sensor is inside the start cell) the surface is evaluated under the start point itself.
  float t_in, g_in;
  {
    const float tpx = tnx - iu, tpy = tny - iv;
    const bool via_x = tpx >= tpy;
    const float t_s = via_x ? tpx : tpy;
    if (t_s >= 0.f) {
      // entered through a' = 0 (via_x: nodes F00 .. F01, coordinate b') or b' = 0 (nodes F00 .. F10, coordinate a')
      const float F00 = TILE_AT(a00), Fn = TILE_AT(a00 + (via_x ? oyb : oxb));
      const float f = fmaf(t_s, via_x ? cv : cu, via_x ? B0 : A0);
      t_in = t_s;
      g_in = fmaf(t_s, dz, oz) - fmaf(f, Fn - F00, F00);
      if (RING && SURF == 0 && t_min > 0.f && g_in <= 0.f) return fminf(t_in, r_max);  // into the map from the side, under the seabed
    } else {
      const float F00 = TILE_AT(a00), F01 = TILE_AT(a00 + oyb), F10 = TILE_AT(a00 + oxb), F11 = TILE_AT(a00 + oxyb);
      const float as = fmaf(t_lo, cu, A0), bs = fmaf(t_lo, cv, B0);
      float h;
      if (SURF == 0) {
        const float h0 = fmaf(as, F10 - F00, F00), h1 = fmaf(as, F11 - F01, F01);
        h = fmaf(bs, h1 - h0, h0);
      } else {
        // fd false: triangles (F00,F10,F11) below the diagonal b' = a', (F00,F01,F11) above it
        // fd true : triangles (F00,F10,F01) where a' + b' <= 1, (F10,F11,F01) beyond
        const bool lower = (fd & (as + bs <= 1.f)) | (!fd & (bs <= as));
        const float sa = lower ? F10 - F00 : F11 - F01;
        const float sb = (lower == fd) ? F01 - F00 : F11 - F10;
        const float hc = (fd & !lower) ? (F10 + F01) - F11 : F00;
        h = fmaf(bs, sb, fmaf(as, sa, hc));
      }
      t_in = t_lo;
      g_in = fmaf(t_lo, dz, oz) - h;
      if (SURF == 0 && !(t_lo > 0.f) && g_in <= 0.f) return 0.f;  // the sensor itself is at or below the seabed
    }
  }
  // per-ray constants of the diagonal crossing  t_d = (ck - A0 + sB B0) * inv_sel:
  //   fd: a' + b' = 1 -> (1 - A0 - B0) / (cu + cv)      !fd: a' = b' -> (B0 - A0) / (cu - cv)
  // and of the two nodes the diagonal joins (P0 at a' = 0 ... P1 at a' = 1 along it)
  float ck = fd ? 1.f : 0.f, sB = fd ? -1.f : 1.f, inv_sel = fast_rcp(fd ? cu + cv : cu - cv);
  int kP0 = fd ? oyb : 0, kP1 = fd ? oxb : oxyb;
  const float cucv = cu * cv;
  float t_out, g_out, t_d = 0.f, g_d = 0.f, kq = 0.f;
  bool in = false, hit;
  for (;;) {
    const bool stepx = tnx <= tny;
    const int aE = a00 + (stepx ? oxb : oyb);  // the node next to F00 on the edge the ray leaves by = next cell's F00
    // always to the cell's exit edge (never cut at t1: the clearance is evaluated ON the edge); a hit beyond
    // r_max is discarded at the end, and the cell that contains t1 still lies inside the tile
    t_out = stepx ? tnx : tny;
    const float f = fmaf(t_out, stepx ? cv : cu, stepx ? B0 : A0);  // position along that edge, it ends in F11
    const float F11 = TILE_AT(a00 + oxyb);
    if (SURF == 0) {
      const float F00 = TILE_AT(a00), F01 = TILE_AT(a00 + oyb), F10 = TILE_AT(a00 + oxb);
      const float E0 = stepx ? F10 : F01;
      g_out = fmaf(t_out, dz, oz) - fmaf(f, F11 - E0, E0);
      const float dt = t_out - t_in;
      kq = -(((F11 - F01) - (F10 - F00)) * cucv) * (dt * dt);
      // a sign change at the exit, or both ends above and a real root inside (K > 0: the patch bulges up to the ray).
      // With B = g_out - g_in - K < 0 the smaller root is 2 g_in / (sqrt(disc) - B) and sqrt(disc) <= -B, so it
      // can only be < 1 if g_out < K: that one compare keeps the grazing test out of the common path
      // (the branch is wave-uniform: taken only when some lane is that close to the surface)
      hit = (g_out > 0.f) != (g_in > 0.f);
      const bool maybe = (!hit) & (g_out < kq);
      if (__builtin_amdgcn_ballot_w64(maybe) != 0ull) {
        const float Bq = (g_out - g_in) - kq;
        const float disc = fmaf(Bq, Bq, -4.f * kq * g_in);
        if (maybe & (Bq < 0.f) & (disc >= 0.f)) hit = 2.f * g_in < fast_sqrt(disc) - Bq;
      }
    } else {
      if (SURF == 1) {
        fd = ((__float_as_uint(TILE_AT(a00 + kcb)) & 1u) != 0u) != flip;
        ck = fd ? 1.f : 0.f;
        sB = fd ? -1.f : 1.f;
        inv_sel = fast_rcp(fd ? cu + cv : cu - cv);
        kP0 = fd ? oyb : 0;
        kP1 = fd ? oxb : oxyb;
      }
      const float E0 = TILE_AT(aE), P0 = TILE_AT(a00 + kP0), P1 = TILE_AT(a00 + kP1);
      g_out = fmaf(t_out, dz, oz) - fmaf(f, F11 - E0, E0);
      t_d = fmaf(sB, B0, ck - A0) * inv_sel;
      in = (t_d > t_in) & (t_d < t_out);  // NaN / inf: the ray does not cross the diagonal inside the cell
      g_d = fmaf(t_d, dz, oz) - fmaf(fmaf(t_d, cu, A0), P1 - P0, P0);
      const bool c_i = g_in > 0.f, c_d = g_d > 0.f, c_o = g_out > 0.f;
      const bool x1 = in & (c_d != c_i);  // sign change on the first piece
      hit = x1 | ((c_o != c_i) != x1);    // or on the last one: c_o != (in ? c_d : c_i), as lane-mask XORs
    }
    if (RING ? (hit | !(t_out < t1) | (g_out != g_out)) : (hit | !(t_out < t1))) break;
    a00 = aE;
    A0 -= stepx ? 1.f : 0.f;
    B0 -= stepx ? 0.f : 1.f;
    tnx += stepx ? iu : 0.f;
    tny += stepx ? 0.f : iv;
    t_in = t_out;
    g_in = g_out;
  }
#undef TILE_AT
  if (RING && g_out != g_out) return r_max;   // left the map
  if (!hit) return r_max;
  float root;
  if (SURF == 0) {
    // smallest root in [0, 1] of K tau^2 + B tau + g_in = 0
    const float Bq = (g_out - g_in) - kq;
    const float disc = fmaxf(fmaf(Bq, Bq, -4.f * kq * g_in), 0.f);
    const float sq = fast_sqrt(disc);
    const float qv = -0.5f * (Bq + (Bq >= 0.f ? sq : -sq));
    const float r1 = g_in * fast_rcp(qv), r2 = qv * fast_rcp(kq);  // r2 = inf / NaN on a planar patch
    const float c1 = (r1 >= -1e-4f && r1 <= 1.0001f) ? r1 : INF, c2 = (r2 >= -1e-4f && r2 <= 1.0001f) ? r2 : INF;
    float tau = fminf(c1, c2);
    if (!(tau < INF)) tau = g_in * fast_rcp(g_in - g_out);  // rounding pushed both just outside: the chord
    root = fmaf(fminf(fmaxf(tau, 0.f), 1.f), t_out - t_in, t_in);
  } else {
    // the piece of the cell in which g changes sign: [t_in, t_d], [t_d, t_out] or the whole cell
    const bool first = in & ((g_d > 0.f) != (g_in > 0.f));
    const bool from_in = first | !in;
    const float ta = from_in ? t_in : t_d, ga = from_in ? g_in : g_d;
    const float tb = first ? t_d : t_out, gb = first ? g_d : g_out;
    root = fmaf(ga * fast_rcp(ga - gb), tb - ta, ta);
    root = fminf(fmaxf(root, ta), tb);
  }
  return root <= r_max ? fmaxf(root, 0.f) : r_max;
}

// ------------------------------------------------------------------ the general cast kernel
// One wavefront per particle, lanes = consecutive beams, the map in global memory (L2); no LDS, no barriers.
// MODE 1: the groups k_mbes_fast left on a.worklist (window larger than LDS, a fan that is not simple).
// MODE 2: the particles a.perm[0 .. *a.n_dev) -- the fan sweep's hand-over list, in any order.
// Per particle (own_fan): a simple fan is cast by the clearance walk -- the arithmetic of k_mbes_fast, from the global
// array instead of an LDS tile, bit for bit the same result --, everything else by the clipped general march.
template <int MAP, bool EXPECT_ONLY, int MODE>
__global__ void __launch_bounds__(MBES_THREADS, MAP == 1 ? MBES_MIN_WAVES_MESH - 1 : MBES_MIN_WAVES_MESH) k_mbes_cast(MbesArgs a) {
  static_assert(MODE == 1 || MODE == 2, "MODE 1: worklist of groups, MODE 2: list of particles");
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n_eff = mbes_count(a);
  if (MODE == 2 && a.host_count && blockIdx.x == 0 && threadIdx.x == 0) *a.host_count = (int)n_eff;  // (read by the host two updates later)
  const float inv_res = (float)a.inv_res;
  const u32* perm = a.perm;
  const long long nwork = MODE == 1 ? (long long)*a.work_count : (n_eff + MBES_WAVES - 1) / MBES_WAVES;
  double wmax = -__builtin_inf();  // lane 0: largest log-likelihood this wave has written
  for (long long it = blockIdx.x; it < nwork; it += gridDim.x) {
    const long long grp = MODE == 1 ? (long long)a.worklist[it] : it;
    const long long j = grp * MBES_WAVES + w;  // position in the visiting order
    if (j >= n_eff) continue;
    const long long ip = perm ? (long long)perm[j] : j;  // the particle's pose record
    MbesPose P;
    long long i;   // its state slot (== ip unless the records lie in visiting order)
    {
      // the record is wave-uniform: pin it in SGPRs
      const MbesPose Pv = a.pose[ip];
      i = (long long)(u32)__builtin_amdgcn_readfirstlane((int)Pv.slot);
      P.um = uniform_f64(Pv.um);
      P.vm = uniform_f64(Pv.vm);
      P.oz = uniform_f32(Pv.oz);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        P.c1[r] = uniform_f32(Pv.c1[r]);
        P.c2[r] = uniform_f32(Pv.c2[r]);
      }
    }
    const OwnFan F = own_fan(a, P);
    float acc = 0.f;
    int nvalid = 0;
    RayStats rs = {0, 0, 0, 0};
    // lattice maps: the map's cells and nodes relative to the particle's own cell (small integers: exact in fp32)
    const float cu_lo = (float)(-F.I0), cu_hi = (float)(a.nx - 2 - F.I0), cv_lo = (float)(-F.J0), cv_hi = (float)(a.ny - 2 - F.J0);
    const float* gpp = a.grid_pad + ((long long)(F.I0 + 1) * a.nyp + (F.J0 + 1));   // node (I0, J0) inside the ring (only ever dereferenced at map cells)
    for (int b = lane; b < a.n_beams; b += 64) {
      const float2 sc = a.beam_sc[b];
      const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
      const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
      const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
      float e;
      if (!F.sane) {
        e = a.r_max;  // (NaN / absurd position: every beam misses)
      } else if (MAP != 1) {
        // where the ray enters the map's rectangle (t_min = 0 for a sensor over the map), if at all
        const float du = dx * inv_res, dv = dy * inv_res;
        float t0 = 0.f, t1 = a.r_max;
        bool miss = false;
        if (du == 0.f) {
          miss = F.ul < cu_lo || F.ul > cu_hi + 1.f;
        } else {
          const float r = fast_rcp(du), ta = (cu_lo - F.ul) * r, tb = (cu_hi + 1.f - F.ul) * r;
          t0 = fmaxf(t0, fminf(ta, tb));
          t1 = fminf(t1, fmaxf(ta, tb));
        }
        if (dv == 0.f) {
          miss = miss || F.vl < cv_lo || F.vl > cv_hi + 1.f;
        } else {
          const float r = fast_rcp(dv), ta = (cv_lo - F.vl) * r, tb = (cv_hi + 1.f - F.vl) * r;
          t0 = fmaxf(t0, fminf(ta, tb));
          t1 = fminf(t1, fmaxf(ta, tb));
        }
        if (miss || !(t0 <= t1)) {
          e = a.r_max;
        } else if (MAP == 0) {
          e = cast_clear<0, true>(gpp, a.nyp, a, F.ul, F.vl, P.oz, du, dv, dz, a.zmax_map, a.r_max, t0, t1, cu_lo, cu_hi, cv_lo, cv_hi);
        } else {
          e = a.diag_mode == 1 ? cast_clear<2, true>(gpp, a.nyp, a, F.ul, F.vl, P.oz, du, dv, dz, a.zmax_map, a.r_max, t0, t1, cu_lo, cu_hi, cv_lo, cv_hi)
            : a.diag_mode == 2 ? cast_clear<3, true>(gpp, a.nyp, a, F.ul, F.vl, P.oz, du, dv, dz, a.zmax_map, a.r_max, t0, t1, cu_lo, cu_hi, cv_lo, cv_hi)
                               : cast_clear<1, true>(gpp, a.nyp, a, F.ul, F.vl, P.oz, du, dv, dz, a.zmax_map, a.r_max, t0, t1, cu_lo, cu_hi, cv_lo, cv_hi);
        }
      } else if (F.simple) {
        e = cast_fast<1>((const float*)(a.mesh.cell_info + ((size_t)F.I0 * a.mesh.gy + F.J0)), a.mesh.gy, a, F.I0, F.J0, F.ul, F.vl,
                         P.oz, dx * inv_res, dy * inv_res, dz, a.zmax_map, a.r_max, rs);
      } else {
        float t_lo = 0.f;  // skip the water column above the map's highest point
        if (dz < 0.f && P.oz > a.zmax_map) t_lo = fmaxf((a.zmax_map - P.oz) * fast_rcp(dz) - 1e-3f, 0.f);
        e = cast_ray(a.mesh.cell_info, a.mesh.gy, a, F.I0, F.J0, a.mesh.gx, a.mesh.gy, F.ul, F.vl, P.oz, dx * inv_res,
                     dy * inv_res, dx, dy, dz, t_lo, a.r_max, rs);
      }
      if (EXPECT_ONLY) {
        if (i >= a.exp_first && i < a.exp_first + a.exp_count)
          a.exp_out[(size_t)(i - a.exp_first) * a.n_beams + b] = e;
      } else {
        const float rm = a.ranges[b];
        if (rm > 0.f) {  // NaN fails the test
          const float d = (rm - e) * a.inv_sigma;
          acc += d * d;
          ++nvalid;
        }
      }
    }
#ifdef MBES_STATS
    if (a.stats) {
      const int s0 = wave_sum(rs.steps), s1 = wave_sum(rs.tests), s2 = wave_sum(rs.rays), s3 = wave_sum(rs.retries);
      if (lane == 0) {
        atomicAdd(&a.stats[0], (unsigned long long)s0);
        atomicAdd(&a.stats[1], (unsigned long long)s1);
        atomicAdd(&a.stats[2], (unsigned long long)s2);
        atomicAdd(&a.stats[3], (unsigned long long)s3);
      }
    }
#endif
    if (!EXPECT_ONLY) {
      const double accd = wave_sum((double)acc);
      const int nv = wave_sum(nvalid);
      if (lane == 0) {
        const double v = -0.5 * accd - (double)nv * a.lognorm;
        a.lw[i] = v;
        wmax = v > wmax ? v : wmax;  // NaN never wins
      }
    }
  }
  // the normalisation needs max lw: one atomic per wave on an order-preserving key, spread over the slots
  if (!EXPECT_ONLY && lane == 0 && a.max_slots && wmax > -__builtin_inf())
    atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * MBES_WAVES + w) & (MCL_MAX_SLOTS - 1)], ordered_key(wmax));
}

// ------------------------------------------------------------------ the fast cast kernel
// Height grids (MAP 0) and structured meshes (MAP 2).  One wavefront per particle, lanes = consecutive
// beams, MBES_WAVES particles per workgroup share one LDS tile.  Everything that is decided per GROUP (tile
// window, eligibility) was decided by the pose kernel (classify_group) and arrives as one 32-byte record
// through scalar loads; what is left per group is the staging of the tile and its maximum height.
// SURF: the surface cast_clear follows -- 0 bilinear grid, 1 triangulated with a per-cell diagonal bit,
// 2 / 3 triangulated with every cell split along 00-11 / 10-01.
// SURF 4: arbitrary triangle soups -- the tile holds the 8-byte cell words (conservative z-range, record range)
// of mcl_mesh.h, the traversal is cast_fast<1> (cell march on the z-ranges + plane-form triangle records from L2).
template <int SURF, bool EXPECT_ONLY>
__global__ void __launch_bounds__(MBES_THREADS, SURF == 4 ? MBES_MIN_WAVES_MESH : MBES_MIN_WAVES_PER_SIMD) k_mbes_fast(MbesArgs a) {
  constexpr bool CELLS = SURF == 4;
  constexpr int TILE_WORDS = CELLS ? (MBES_TILE_FLOATS * 3) / 2 : MBES_TILE_FLOATS;  // 48 KiB of cell words / 32 KiB of heights
  constexpr int TILE_CAP = CELLS ? TILE_WORDS / 2 : TILE_WORDS;
  __shared__ __attribute__((aligned(16))) float tile[TILE_WORDS];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n_eff = mbes_count(a);
  const long long ngroups = (n_eff + MBES_WAVES - 1) / MBES_WAVES;
  const float inv_res = (float)a.inv_res;
  double wmax = -__builtin_inf();  // lane 0: largest log-likelihood this wave has written
  // the staged tile: origin and size (cw == 0: none yet).  Consecutive groups of a converged cloud have (nearly) the
  // same window, so the tile is re-used for as long as the next group's window lies inside it -- no staging, no
  // barrier, the waves of the workgroup drift freely.  (Which tile a particle is cast from does not change its result.)
  int cx0 = 0, cy0 = 0, cw = 0, ch = 0;
  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const MbesGroup G = a.groups[grp];  // uniform address: scalar loads
    if (!G.fast) continue;              // on the general kernel's worklist
    const long long j = grp * MBES_WAVES + w;  // position in the visiting order
    const bool valid = j < n_eff;
    const long long i = (valid && a.perm) ? (long long)a.perm[j] : j;  // the particle
    const bool inside = G.tx0 >= cx0 && G.ty0 >= cy0 && G.tx0 + G.tw <= cx0 + cw && G.ty0 + G.th <= cy0 + ch;
    if (!inside) {  // uniform over the workgroup (G, c* are the same in every wave)
      // window + margin, clipped to the map; without the margin if that does not fit the LDS tile
      int mg = MBES_TILE_MARGIN;
      int ex0, ey0, ew, eh;
      for (;;) {
        ex0 = max(G.tx0 - mg, 0);
        ey0 = max(G.ty0 - mg, 0);
        ew = min(G.tx0 + G.tw + mg, a.nx - (CELLS ? 1 : 0)) - ex0;
        eh = min(G.ty0 + G.th + mg, a.ny - (CELLS ? 1 : 0)) - ey0;
        if (mg == 0 || ew * eh <= TILE_CAP) break;
        mg >>= 1;
      }
      cx0 = ex0;
      cy0 = ey0;
      cw = ew;
      ch = eh;
      __syncthreads();  // every wave is done with the previous tile
      // rows to waves, columns to lanes (coalesced along iy)
      if (!CELLS) {
        for (int ix = w; ix < cw; ix += MBES_WAVES) {
          const float* src = a.grid + (size_t)(cx0 + ix) * a.ny + cy0;
          for (int iy = lane; iy < ch; iy += 64) tile[ix * ch + iy] = src[iy];
        }
      } else {
        uint2* t2 = (uint2*)tile;
        for (int ix = w; ix < cw; ix += MBES_WAVES) {
          const uint2* src = a.mesh.cell_info + (size_t)(cx0 + ix) * a.mesh.gy + cy0;
          for (int iy = lane; iy < ch; iy += 64) t2[ix * ch + iy] = src[iy];
        }
      }
      __syncthreads();
    }
    const int th = ch;
    if (!valid) continue;
    // the pose record is wave-uniform (scalar loads); coordinates relative to the particle's own cell, the tile
    // addressed from that cell's word (the determinism rule at the top of this file)
    const MbesPose P = a.pose[i];
    const OwnFan F = own_fan(a, P);
    const int toff = (F.I0 - cx0) * th + (F.J0 - cy0);
    float acc = 0.f;
    int nvalid = 0;
    for (int b = lane; b < a.n_beams; b += 64) {
      const float2 sc = a.beam_sc[b];
      const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
      const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
      const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
      float e;
      if (CELLS) {
        RayStats rs = {0, 0, 0, 0};
        e = cast_fast<1>((const float*)((const uint2*)tile + toff), th, a, F.I0, F.J0, F.ul, F.vl, P.oz, dx * inv_res, dy * inv_res, dz,
                         a.zmax_map, a.r_max, rs);
      } else {
        e = cast_clear<(CELLS ? 0 : SURF)>(tile + toff, th, a, F.ul, F.vl, P.oz, dx * inv_res, dy * inv_res, dz, a.zmax_map, a.r_max);
      }
      if (EXPECT_ONLY) {
        if (i >= a.exp_first && i < a.exp_first + a.exp_count)
          a.exp_out[(size_t)(i - a.exp_first) * a.n_beams + b] = e;
      } else {
        const float rm = a.ranges[b];
        if (rm > 0.f) {  // NaN fails the test
          const float d = (rm - e) * a.inv_sigma;
          acc += d * d;
          ++nvalid;
        }
      }
    }
    if (!EXPECT_ONLY) {
      const double accd = wave_sum((double)acc);
      const int nv = wave_sum(nvalid);
      if (lane == 0) {
        const double v = -0.5 * accd - (double)nv * a.lognorm;
        a.lw[i] = v;
        wmax = v > wmax ? v : wmax;  // NaN never wins
      }
    }
  }
  // the normalisation needs max lw: one atomic per wave on an order-preserving key, spread over the slots
  if (!EXPECT_ONLY && lane == 0 && a.max_slots && wmax > -__builtin_inf())
    atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * MBES_WAVES + w) & (MCL_MAX_SLOTS - 1)], ordered_key(wmax));
}
