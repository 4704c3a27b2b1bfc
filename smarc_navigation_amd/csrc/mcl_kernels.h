// mcl_kernels.h -- streaming kernels of the particle filter hot path (gfx950, wave64, fp64 SoA).
// K1 predict, K2a GPS weight, K3 normalise/quantise, K4 u64 scan + offspring CDF, K5 lost-slot
// scan + reassign gather + noise, K6 mean/cov, PoseArray export.  All HBM-streaming; no MFMA.
#pragma once
#include "mcl_device.h"

#define MCL_BLOCK 256
#define MCL_SCAN_ITEMS 4
#define MCL_SCAN_TILE (MCL_BLOCK * MCL_SCAN_ITEMS)
#define MCL_MAX_GRID 2048

struct StatePtrs {
  double* c[6];  // x, y, z, roll, pitch, yaw
};
struct NoiseArgs {
  double sq[6];   // sqrt(cov)
  u32 k0, k1;     // Philox key (seed)
  u32 step, purpose;
  long long gid0;  // global id of local particle 0
};

// ------------------------------------------------------------------ spatial visiting order of the fan sweep
// ("particle order").  The sweep runs one LANE per particle side, so a wave is as fast as its 64 particles
// are alike; slots carry no spatial order (resampling leaves survivors where they are).  Only the VISITING order is
// changed: the resample gather bins every particle it writes by (x, y, yaw) -- bins over mean +- VISIT_RANGE sigma of
// the cloud one step earlier, extrapolated by one step of its drift -- and counts, per workgroup, how many of its
// particles fell into each bin, handing every particle its rank inside (workgroup, bin); k_visit_scan turns the
// count matrix into start positions; the predict kernel of the NEXT step writes the pose record of slot i to position
// base[workgroup of i][key] + rank (an exact counting sort: a bijection onto [0, n)) with the slot
// in the record.  State slots, RNG keys (global ids) and keep / lost / dupes are untouched (auv_pf.py:183-198), and
// by the determinism rule (mcl_mbes.h) the order in which particles are cast changes no log-likelihood.
#define VISIT_MAX_BINS 4096
#define VISIT_KEY_BITS 12
#define VISIT_OWNER_SHIFT 10   // log2(RS_BLOCK): the gather workgroup of slot i is (i >> 10) % its grid
#define VISIT_OWNER_MASK 255   // GATHER_MAX_GRID - 1 (a smaller grid covers every slot with its first pass: no wrap)
struct VisitPar {    // written by the gather's last block for the next gather
  double mean[3];    // x, y, wrapped yaw of the cloud the bins were derived from
  double lo[3];      // lower edge of bin 0
  float inv[3];      // bins per unit
  int valid;
};
struct VisitArgs {
  u32* okey;             // per slot: key | rank << VISIT_KEY_BITS        (nullptr: no visiting order)
  unsigned short* cnt;   // [workgroup][nb]: particles of a gather workgroup per bin (<= 65 535: the host checks)
  u32* base;             // [workgroup][nb]: (k_visit_scan) position of that workgroup's first particle of the bin
  u64* desc;             // nb / 64 look-back descriptors of k_visit_scan: epoch << 32 | particles in the workgroup's 64 bins
  u32 epoch;             // differs from launch to launch: stale descriptors are never valid
  const VisitPar* par_in;
  VisitPar* par_out;
  int nbx, nby, nbw, nb; // bins per dimension, nb = nbx * nby * nbw <= VISIT_MAX_BINS, a multiple of 64
  float range;           // bins span mean +- range * sigma
  // the process noise the NEXT predict will add (Philox is counter-based: its draws are known now) -- the key is taken
  // from where the particle will be cast, not from where the gather leaves it: sqrt of the x, y, yaw process
  // covariances and the predict's step counter (a guess: any other next call only blurs the bins)
  float psq[3];
  u32 pstep;
};
__device__ __forceinline__ u32 load_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 6 normals for particle gid (purpose 0 init / 2 resample-noise): two Philox blocks
// Counter-based: a draw depends only on (gid, block, step, purpose, seed), so a pair whose two
// covariances are zero is simply not evaluated (its products with sqrt(cov) = 0 are zero anyway);
// the launch covariances leave (z, roll) out -- one fp64 log/sqrt/sincos less per particle.
__device__ __forceinline__ void native_normals6(long long gid, const NoiseArgs& a, double z[6]) {
  const bool p0 = a.sq[0] != 0.0 || a.sq[1] != 0.0, p1 = a.sq[2] != 0.0 || a.sq[3] != 0.0;
  const bool p2 = a.sq[4] != 0.0 || a.sq[5] != 0.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) z[c] = 0.0;
  if (p0 || p1) {
    const u32x4 o = philox4x32((u32)gid, 0u, a.step, a.purpose, a.k0, a.k1);
    if (p0) box_muller(o.x, o.y, z[0], z[1]);
    if (p1) box_muller(o.z, o.w, z[2], z[3]);
  }
  if (p2) {
    const u32x4 o = philox4x32((u32)gid, 1u, a.step, a.purpose, a.k0, a.k1);
    box_muller(o.x, o.y, z[4], z[5]);
  }
}

// ------------------------------------------------------------------ a2: Particle.add_noise
// (auv_particle.py:32-36).  replay: n x 6 particle-major normals or nullptr (Philox).
__global__ void __launch_bounds__(MCL_BLOCK) k_add_noise(StatePtrs s, long long n, NoiseArgs a,
                                                         const double* __restrict__ replay, int zero_first) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    double z[6];
    if (replay) {
#pragma unroll
      for (int c = 0; c < 6; ++c) z[c] = replay[i * 6 + c];
    } else {
      native_normals6(a.gid0 + i, a, z);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double base = zero_first ? 0.0 : s.c[c][i];
      s.c[c][i] = base + a.sq[c] * z[c];
    }
  }
}

// ------------------------------------------------------------------ a4/a5: Particle.motion_pred
// (auv_particle.py:38-70).  Per-step constants are hoisted to the host (the reference recomputes
// euler_from_quaternion and the Ry'*Rx product per particle with identical results):
//   m = (Ry' Rx)(v dt)  ->  step = Rz(yaw_t) m ; rows 0-1 only (row 2 of fullRotation is unused).
struct PredictArgs {
  double m0, m1;      // rows 0,1 of (Ry' Rx) (v dt)
  double wzdt;        // w_z * dt
  double z, roll, pitch;
  NoiseArgs nz;
  // fused step, sweep path: the predict kernel's first workgroup also zeroes the control block (max-lw slots, hand-over
  // counters) for the update that follows -- a hipMemsetAsync of 524 B is TWO fill kernels and a launch gap, 15 us
  unsigned long long* zero_ptr;
  int zero_words;
  // fused step: z, roll and pitch are the odometry's on EVERY particle after motion_pred (auv_particle.py:55-57,70) and
  // the resample gather of the same call substitutes them -- they are not stored here and not read there (48 B x N of
  // HBM traffic per step); the host fills them in if the step fails between the two kernels
  int skip_uniform;
  // visiting order (see VisitArgs): when set, the pose record of slot i goes to its sorted position
  const u32* visit_okey;
  const u32* visit_base;
  int visit_nb;
};
// z, roll, pitch of every particle := the three constants (the deferred stores of a fused step that did not reach
// its gather)
__global__ void __launch_bounds__(MCL_BLOCK) k_fill_uniform(StatePtrs s, long long n, double z, double roll, double pitch) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    s.c[2][i] = z;
    s.c[3][i] = roll;
    s.c[4][i] = pitch;
  }
}
__global__ void __launch_bounds__(MCL_BLOCK) k_predict(StatePtrs s, long long n, PredictArgs a,
                                                       const double* __restrict__ replay) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    double n0, n1, n5;
    if (replay) {
      n0 = replay[i * 6 + 0];
      n1 = replay[i * 6 + 1];
      n5 = replay[i * 6 + 5];
    } else {
      u32x4 o = philox4x32((u32)(a.nz.gid0 + i), 0u, a.nz.step, 1u, a.nz.k0, a.nz.k1);
      double unused;
      box_muller(o.x, o.y, n0, n1);
      box_muller(o.z, o.w, n5, unused);
    }
    const double yaw_t = wrap_pi(s.c[5][i] + a.wzdt + a.nz.sq[5] * n5);
    double sy, cy;
    sincos(yaw_t, &sy, &cy);
    s.c[0][i] += (cy * a.m0 - sy * a.m1) + a.nz.sq[0] * n0;
    s.c[1][i] += (sy * a.m0 + cy * a.m1) + a.nz.sq[1] * n1;
    s.c[2][i] = a.z;      // depth read directly (auv_particle.py:70)
    s.c[3][i] = a.roll;   // roll/pitch read directly (auv_particle.py:55-57)
    s.c[4][i] = a.pitch;
    s.c[5][i] = yaw_t;
  }
}

// ------------------------------------------------------------------ a7: compute_weight (GPS)
// (auv_particle.py:72-106): p_map = m2o [x y z 1]; log N2(gps; p_map_xy, sigma^2 I)
struct GpsArgs {
  double r0[4], r1[4];  // rows 0,1 of m2o
  double gx, gy, inv_s2, lognorm;
};
__global__ void __launch_bounds__(MCL_BLOCK) k_gps_logw(StatePtrs s, long long n, GpsArgs a,
                                                        double* __restrict__ lw) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double x = s.c[0][i], y = s.c[1][i], z = s.c[2][i];
    const double px = a.r0[0] * x + a.r0[1] * y + a.r0[2] * z + a.r0[3];
    const double py = a.r1[0] * x + a.r1[1] * y + a.r1[2] * z + a.r1[3];
    const double dx = a.gx - px, dy = a.gy - py;
    lw[i] = -0.5 * ((dx * dx + dy * dy) * a.inv_s2) - a.lognorm;
  }
}

// ------------------------------------------------------------------ K3: max log-weight
__global__ void __launch_bounds__(MCL_BLOCK) k_max_partial(const double* __restrict__ v, long long n,
                                                           double* __restrict__ part) {
  __shared__ double sh[16];
  double m = -__builtin_inf();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    double x = v[i];
    m = (x > m) ? x : m;  // NaN never wins
  }
  m = block_max(m, sh, -__builtin_inf());
  if (threadIdx.x == 0) part[blockIdx.x] = m;
}
__global__ void __launch_bounds__(1024) k_max_final(const double* __restrict__ part, int np,
                                                    double* __restrict__ out) {
  __shared__ double sh[16];
  double m = -__builtin_inf();
  for (int i = threadIdx.x; i < np; i += blockDim.x) m = part[i] > m ? part[i] : m;
  m = block_max(m, sh, -__builtin_inf());
  if (threadIdx.x == 0) out[0] = m;
}

// ------------------------------------------------------------------ K3/K4: fixed-point weights
// q_i = floor(w_i / mw * 2^s), w_i = det_exp(lw_i) + 1e-200 (mode 0: GPS, auv_pf.py:165), mw = weight of the max-lw
// particle; mode 1 (log-likelihoods): floor(exp(lw_i) 2^(s - K)), K the integer exponent of the maximum
// (mcl_device.h: quantise_log_weight).  Integer sums are exact and order-free, so the CDF is identical for any grid
// shape / GPU count.
__device__ __forceinline__ u64 quantise_weight(double lw, double m_lw, int mode, double scale, int s) {
  double w, mw;
  if (mode == 0) {
    w = det_exp(lw) + 1.e-200;
    mw = det_exp(m_lw) + 1.e-200;
  } else if (mode == 1) {
    return quantise_log_weight(lw, weight_exponent(m_lw), s);
  } else {  // mode 2: `lw` already holds linear weights, m_lw their maximum
    w = lw > 0.0 ? lw : 0.0;
    mw = m_lw;
  }
  return (u64)((w / mw) * scale);
}

// pass 2: exclusive scan of the tile sums in one block (T = u64 or u32); total -> total_out[0]
template <class T>
__global__ void __launch_bounds__(1024) k_scan_tile_sums(T* __restrict__ tile_sum, long long ntiles,
                                                         T* __restrict__ total_out) {
  __shared__ T sh[16];
  __shared__ T carry_sh;
  if (threadIdx.x == 0) carry_sh = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (long long base = 0; base < ntiles; base += blockDim.x) {
    long long i = base + threadIdx.x;
    T v = i < ntiles ? tile_sum[i] : T(0);
    T incl = wave_scan_incl(v);
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    T woff = T(0);
    for (int k = 0; k < w; ++k) woff += sh[k];
    T carry = carry_sh;
    if (i < ntiles) tile_sum[i] = carry + woff + incl - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_sh = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) total_out[0] = carry_sh;
}

// block-wide inclusive scan of one tile held as ITEMS per thread in BLOCKED order
// (thread t owns items t*ITEMS .. t*ITEMS+ITEMS-1)
template <class T>
__device__ __forceinline__ void tile_scan_blocked(T (&v)[MCL_SCAN_ITEMS], T* sh /*16*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 1; k < MCL_SCAN_ITEMS; ++k) v[k] += v[k - 1];
  T tot = v[MCL_SCAN_ITEMS - 1];
  T incl = wave_scan_incl(tot);
  __syncthreads();
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  T off = incl - tot;
  for (int k = 0; k < w; ++k) off += sh[k];
#pragma unroll
  for (int k = 0; k < MCL_SCAN_ITEMS; ++k) v[k] += off;
}

// pass 3: C_j = inclusive scan of q (+ shard offset); offspring CDF
//   ncum[j] = #{ i in [0,N) : (U + i 2^53) T < C_j N 2^53 } = floor(C_j N / T) + [rem 2^53 > U T]
// (systematic_resample, resampling.py:154-168, in exact integer arithmetic)
struct CdfArgs {
  const u64* totals;   // per-shard totals, `world` entries (device)
  int rank, world;
  u64 n_global;
  u64 u53;
  const u64* shift;    // q is at the shard's own exponent: the cloud's weights are q >> shift[0] (nullptr: q as it is)
};
__global__ void __launch_bounds__(MCL_BLOCK) k_offspring_cdf(const u64* __restrict__ q, long long n,
                                                             const u64* __restrict__ tile_off, CdfArgs a,
                                                             u32* __restrict__ ncum) {
  __shared__ u64 sh[16];
  u64 shard_off = 0, T = 0;
  const u32 qshift = a.shift ? (u32)a.shift[0] : 0u;
  for (int r = 0; r < a.world; ++r) {
    u64 t = a.totals[r];
    if (r < a.rank) shard_off += t;
    T += t;
  }
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE + (long long)threadIdx.x * MCL_SCAN_ITEMS;
    u64 v[MCL_SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) v[k] = (base + k < n) ? shift_weight(q[base + k], qshift) : 0ull;
    tile_scan_blocked(v, sh);
    const u64 off = shard_off + tile_off[tile];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
      if (base + k < n) {
        u64 quo, rem;
        muldiv_u64(v[k] + off, a.n_global, T, quo, rem);
        ncum[base + k] = (u32)quo + (shl53_gt_mul(rem, a.u53, T) ? 1u : 0u);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ K5: lost-slot ranks
// zflag_j = [offspring_j == 0]; zcum = inclusive scan over the GLOBAL particle range
__device__ __forceinline__ u32 offspring(const u32* __restrict__ ncum, long long j) {
  return ncum[j] - (j > 0 ? ncum[j - 1] : 0u);
}
__global__ void __launch_bounds__(MCL_BLOCK) k_zero_tile_sums(const u32* __restrict__ ncum, long long n,
                                                              u32* __restrict__ tile_sum) {
  __shared__ u32 sh[16];
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE;
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
      long long j = base + (long long)k * MCL_BLOCK + threadIdx.x;
      if (j < n) acc += offspring(ncum, j) == 0u ? 1u : 0u;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) tile_sum[tile] = acc;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(MCL_BLOCK) k_zero_scan(const u32* __restrict__ ncum, long long n,
                                                         const u32* __restrict__ tile_off,
                                                         u32* __restrict__ zcum) {
  __shared__ u32 sh[16];
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE + (long long)threadIdx.x * MCL_SCAN_ITEMS;
    u32 v[MCL_SCAN_ITEMS];
    u32 prev = (base > 0 && base - 1 < n) ? ncum[base - 1] : 0u;
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
      u32 cur = (base + k < n) ? ncum[base + k] : prev;
      v[k] = (base + k < n && cur == prev) ? 1u : 0u;
      prev = cur;
    }
    tile_scan_blocked(v, sh);
    const u32 off = tile_off[tile];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k)
      if (base + k < n) zcum[base + k] = v[k] + off;
    __syncthreads();
  }
}

// a12 + a2: keep/lost/dupes reassign (auv_pf.py:183-198) then add_noise (auv_pf.py:191-192).
// Survivors stay in their slot; the k-th lost slot (ascending) takes the k-th entry of the dupes
// list = ancestor j with  d_{j-1} <= k < d_j,  d_j = ncum_j - (j+1) + zcum_j  (cumulative surplus
// copies).  src reads the pre-resample state (global copy), dst is this shard's slice.
struct ReassignArgs {
  StatePtrs src;       // n_global particles (pre-resample), global indexing
  StatePtrs dst;       // n local particles
  long long n, n_global, goff;
  NoiseArgs nz;
  int add_noise;
};
__global__ void __launch_bounds__(MCL_BLOCK) k_reassign_noise(ReassignArgs a, const u32* __restrict__ ncum,
                                                              const u32* __restrict__ zcum,
                                                              const double* __restrict__ replay) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long g = a.goff + i;
    long long src = g;
    if (offspring(ncum, g) == 0u) {
      const u32 k = zcum[g] - 1u;
      long long lo = 0, hi = a.n_global;  // first j with d_j > k
      while (lo < hi) {
        long long mid = lo + ((hi - lo) >> 1);
        u32 d = ncum[mid] - (u32)(mid + 1) + zcum[mid];
        if (d > k)
          hi = mid;
        else
          lo = mid + 1;
      }
      src = lo < a.n_global ? lo : a.n_global - 1;
    }
    double z[6] = {0, 0, 0, 0, 0, 0};
    if (a.add_noise) {
      if (replay) {
#pragma unroll
        for (int c = 0; c < 6; ++c) z[c] = replay[i * 6 + c];
      } else {
        native_normals6(g, a.nz, z);
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) a.dst.c[c][i] = a.src.c[c][src] + a.nz.sq[c] * z[c];
  }
}

// FilterPy-style ancestor indices: idx_i = min{ j : ncum_j > i }
__global__ void __launch_bounds__(MCL_BLOCK) k_indices(const u32* __restrict__ ncum, long long n_global,
                                                       long long i0, long long cnt, int* __restrict__ idx) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < cnt;
       t += (long long)gridDim.x * blockDim.x) {
    const u32 i = (u32)(i0 + t);
    long long lo = 0, hi = n_global;
    while (lo < hi) {
      long long mid = lo + ((hi - lo) >> 1);
      if (ncum[mid] > i)
        hi = mid;
      else
        lo = mid + 1;
    }
    idx[t] = (int)(lo < n_global ? lo : n_global - 1);
  }
}

// gather by explicit indices (used by the non-systematic schemes): dst[i] = src[idx[i]]
__global__ void __launch_bounds__(MCL_BLOCK) k_normalised_weights(const u64* __restrict__ q, long long n,
                                                                  const u64* __restrict__ totals, int world,
                                                                  const u64* __restrict__ shift, double* __restrict__ w) {
  u64 T = 0;
  for (int r = 0; r < world; ++r) T += totals[r];
  const double inv = 1.0 / (double)T;
  const u32 d = shift ? (u32)shift[0] : 0u;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    w[i] = (double)shift_weight(q[i], d) * inv;
}

// ------------------------------------------------------------------ K6: mean / covariance
// (auv_pf.py:218-252).  pass 1: sums of the 6 components + wrapped yaw; pass 2: centred moments.
__global__ void __launch_bounds__(MCL_BLOCK) k_mean_partial(StatePtrs s, long long n,
                                                            double* __restrict__ part /*[7][grid]*/) {
  __shared__ double sh[16];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] += s.c[c][i];
    acc[6] += wrap_pi(s.c[5][i]);
  }
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    double r = block_sum(acc[c], sh);
    if (threadIdx.x == 0) part[(size_t)c * gridDim.x + blockIdx.x] = r;
  }
}
__global__ void __launch_bounds__(MCL_BLOCK) k_cov_partial(StatePtrs s, long long n,
                                                           const double* __restrict__ sums, double inv_n,
                                                           double* __restrict__ part /*[6][grid]*/) {
  __shared__ double sh[16];
  const double mx = sums[0] * inv_n, my = sums[1] * inv_n, mz = sums[2] * inv_n;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double dx = s.c[0][i] - mx, dy = s.c[1][i] - my, dz = s.c[2][i] - mz;
    acc[0] += dx * dx;
    acc[1] += dy * dy;
    acc[2] += dz * dz;
    acc[3] += dx * dy;
    acc[4] += dx * dz;
    acc[5] += dy * dz;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double r = block_sum(acc[c], sh);
    if (threadIdx.x == 0) part[(size_t)c * gridDim.x + blockIdx.x] = r;
  }
}
// out[c] = sum_b part[c][b]  (one block per component)
__global__ void __launch_bounds__(MCL_BLOCK) k_sum_final(const double* __restrict__ part, int np,
                                                         double* __restrict__ out) {
  __shared__ double sh[16];
  const double* p = part + (size_t)blockIdx.x * np;
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc += p[i];
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// PoseArray payload (auv_pf.py:266-277): quaternion_from_euler per particle
__global__ void __launch_bounds__(MCL_BLOCK) k_poses(StatePtrs s, long long n, double* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    double sr, cr, sp, cp, sy, cy;
    sincos(s.c[3][i] * 0.5, &sr, &cr);
    sincos(s.c[4][i] * 0.5, &sp, &cp);
    sincos(s.c[5][i] * 0.5, &sy, &cy);
    double* o = out + i * 7;
    o[0] = s.c[0][i];
    o[1] = s.c[1][i];
    o[2] = s.c[2][i];
    o[3] = cp * (sr * cy) - sp * (cr * sy);
    o[4] = cp * (sr * sy) + sp * (cr * cy);
    o[5] = cp * (cr * sy) - sp * (sr * cy);
    o[6] = cp * (cr * cy) + sp * (sr * sy);
  }
}
