// mcl_device.h -- device-side helpers shared by the MCL kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define MCL_WAVE 64
// Single target.  Beyond wave64 / DPP / the sweep's assembly, the inter-workgroup hand-offs of mcl_resample.h rest on
// gfx950's cache behaviour and not on the HIP memory model alone: partial sums and look-back descriptors are published
// by ONE write-through (sc1) store, drained (s_waitcnt vmcnt(0)) before the relaxed ticket, and read by L1-bypassing
// (sc1) loads -- no release / acquire fence (an agent-scope acquire is an L1 invalidate, ~1.7 us in a 30 us kernel;
// MI355X_MICROARCH.md "inter-workgroup visibility").  Another architecture must not inherit that silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmcl_hip.so's kernels are written for gfx950 (MI355X) only: compile with --offload-arch=gfx950"
#endif
#define MCL_PI 3.14159265358979323846

typedef unsigned long long u64;
typedef unsigned int u32;

// ---------------------------------------------------------------- wave / block reductions
// Through the data-parallel-primitive lane moves (row_shr 1 / 2 / 4 / 8 inside rows of 16 lanes, then row_bcast 15 and
// 31 across rows) a wave's inclusive scan takes six moves and six additions and no LDS round trip; a __shfl of a 64-bit
// value is two ds_bpermute and ~100 cycles of latency per step, and the short kernels of the resampling chain (tile
// scans, 13 moment sums) spent microseconds there.  Lanes without a source add 0.
template <int CTRL, int ROW_MASK, class T>
__device__ __forceinline__ T dpp_move(T v) {
  static_assert(sizeof(T) == 8 || sizeof(T) == 4, "32- or 64-bit values");
  if constexpr (sizeof(T) == 8) {
    union { T t; int w[2]; } a, r;
    a.t = v;
    r.w[0] = __builtin_amdgcn_update_dpp(0, a.w[0], CTRL, ROW_MASK, 0xf, false);
    r.w[1] = __builtin_amdgcn_update_dpp(0, a.w[1], CTRL, ROW_MASK, 0xf, false);
    return r.t;
  } else {
    union { T t; int w; } a, r;
    a.t = v;
    r.w = __builtin_amdgcn_update_dpp(0, a.w, CTRL, ROW_MASK, 0xf, false);
    return r.t;
  }
}
template <class T>
__device__ __forceinline__ T readlane63(T v) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int w[2]; } a, r;
    a.t = v;
    r.w[0] = __builtin_amdgcn_readlane(a.w[0], 63);
    r.w[1] = __builtin_amdgcn_readlane(a.w[1], 63);
    return r.t;
  } else {
    union { T t; int w; } a, r;
    a.t = v;
    r.w = __builtin_amdgcn_readlane(a.w, 63);
    return r.t;
  }
}
// inclusive scan across the 64 lanes of a wave (additions in a fixed order)
template <class T>
__device__ __forceinline__ T wave_scan_incl_dpp(T v) {
  v += dpp_move<0x111, 0xf>(v);   // row_shr:1
  v += dpp_move<0x112, 0xf>(v);   // row_shr:2
  v += dpp_move<0x114, 0xf>(v);   // row_shr:4
  v += dpp_move<0x118, 0xf>(v);   // row_shr:8   -> prefix sums inside every row of 16
  v += dpp_move<0x142, 0xa>(v);   // row_bcast:15: the sums of rows 0 and 2 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);   // row_bcast:31: the sum of rows 0 - 1 into rows 2 and 3
  return v;
}
// the wave's sum in EVERY lane; the additions in the scan's order (not wave_sum's tree: use one or the other where
// the last bit of a floating-point sum is compared)
template <class T>
__device__ __forceinline__ T wave_sum_dpp(T v) { return readlane63(wave_scan_incl_dpp(v)); }
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
  if constexpr (std::is_integral<T>::value) {
    return wave_sum_dpp(v);   // (integers: any order gives the same sum)
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, MCL_WAVE);
    return v;  // valid in lane 0
  }
}
template <class T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_down(v, o, MCL_WAVE);
    v = w > v ? w : v;
  }
  return v;
}
template <class T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_down(v, o, MCL_WAVE);
    v = w < v ? w : v;
  }
  return v;
}
// inclusive scan across the 64 lanes of a wave
template <class T>
__device__ __forceinline__ T wave_scan_incl(T v) {
  if constexpr (std::is_integral<T>::value) {
    return wave_scan_incl_dpp(v);
  } else {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      T w = __shfl_up(v, o, MCL_WAVE);
      if (lane >= o) v += w;
    }
    return v;
  }
}

// Block sum for blockDim.x <= 1024; result valid in thread 0.  `sh` must hold 16 T.
template <class T>
__device__ __forceinline__ T block_sum(T v, T* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  T r = T(0);
  if (w == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    r = lane < nw ? sh[lane] : T(0);
    r = wave_sum(r);
  }
  return r;
}
template <class T>
__device__ __forceinline__ T block_max(T v, T* sh, T lowest) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  T r = lowest;
  if (w == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    r = lane < nw ? sh[lane] : lowest;
    r = wave_max(r);
  }
  return r;
}

// ---------------------------------------------------------------- order-preserving keys, agent-scope words
// double -> u64 with the same ordering (NaN excluded by the callers); key 0 is below every value and
// stands for "no value yet" (decodes to -inf).  Lets a max over doubles be an integer atomicMax.
__device__ __forceinline__ u64 ordered_key(double x) {
  const u64 b = (u64)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_value(u64 k) {
  if (k == 0ull) return -__builtin_inf();
  const u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
#define MCL_MAX_SLOTS 64
// relaxed agent-scope accesses (global_load/store ... sc1): L2-served, never a stale per-CU L1 line.
// One naturally aligned 8-byte word written by ONE store is its own hand-off granule (value + tag):
// no fence is needed to read it from another CU (MI355X_MICROARCH.md, inter-workgroup visibility).
// -DMCL_FENCED=1 (csrc/Makefile: libmcl_hip_fenced.so, test infrastructure): the same hand-offs by the HIP memory model's
// book -- release stores, acquire loads, a device-scope fence on either side of every ticket -- whatever they cost.
// tests/test_gpu_zz_fenced.py runs the filter through both builds, bit for bit: if a driver, a partition mode or a
// compiler ever breaks what the fence-free form rests on, the two part ways there instead of in the field.
#ifndef MCL_FENCED
#define MCL_FENCED 0
#endif
#if MCL_FENCED
#define MCL_ORDER_LOAD __ATOMIC_ACQUIRE
#define MCL_ORDER_STORE __ATOMIC_RELEASE
#define MCL_TICKET_FENCE() __threadfence()
#else
#define MCL_ORDER_LOAD __ATOMIC_RELAXED
#define MCL_ORDER_STORE __ATOMIC_RELAXED
#define MCL_TICKET_FENCE() ((void)0)
#endif
__device__ __forceinline__ u64 load_agent(const u64* p) {
  return __hip_atomic_load(p, MCL_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(u64* p, u64 v) {
  __hip_atomic_store(p, v, MCL_ORDER_STORE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
  return __hip_atomic_load(p, MCL_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(double* p, double v) {
  __hip_atomic_store(p, v, MCL_ORDER_STORE, __HIP_MEMORY_SCOPE_AGENT);
}
// max over the slots an update kernel filled (one wave; every lane gets the result)
__device__ __forceinline__ double max_from_slots(const u64* slots) {
  u64 k = slots[threadIdx.x & 63];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 w = __shfl_xor(k, o, MCL_WAVE);
    k = w > k ? w : k;
  }
  return ordered_value(k);
}

// ---------------------------------------------------------------- scalar math
// (a + pi) % (2 pi) - pi with Python's floored modulo (auv_particle.py:48, auv_pf.py:229)
__device__ __forceinline__ double wrap_pi(double a) {
  const double b = 2.0 * MCL_PI;
  double s = a + MCL_PI;
  // fmod(s, b) == s exactly for 0 <= s < b -- a yaw that has not just crossed +-pi: skip ocml's iterative fp64 fmod
  // (the branch is taken by whole waves almost always; the result is bit for bit the same either way)
  if (s >= 0.0 && s < b) return s - MCL_PI;
  double m = fmod(s, b);
  if (m != 0.0) {
    if (m < 0.0) m += b;
  } else {
    m = 0.0;
  }
  return m - MCL_PI;
}

// Deterministic exp (DESIGN.md "Resampling arithmetic"): only fma/mul/rint/bit ops, the same
// sequence as oracle/mcl_oracle.c:orc_det_exp, so CPU and GPU agree bit for bit.
__device__ __forceinline__ double det_exp(double x) {
#pragma clang fp contract(off)
  if (!(x >= -700.0)) return 0.0;
  if (x > 709.0) return __builtin_inf();
  const double LOG2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double k = __builtin_rint(x * LOG2E);
  double r = __builtin_fma(-k, LN2_HI, x);
  r = __builtin_fma(-k, LN2_LO, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  long long ki = (long long)k;
  u64 bits = (u64)(ki + 1023) << 52;
  return p * __longlong_as_double((long long)bits);
}

// exp(x) = p 2^k, p in [2^-1/2, 2^1/2]: det_exp's own sequence without the final scaling, for any finite x (no
// under- / overflow: the exponent stays an integer).  NaN, +-inf and |x| > 1e11 (no log-likelihood): p = 0.
__device__ __forceinline__ double det_exp_parts(double x, long long& ki) {
#pragma clang fp contract(off)
  ki = 0;
  if (!(x >= -1.0e11) || !(x <= 1.0e11)) return 0.0;
  const double LOG2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double k = __builtin_rint(x * LOG2E);
  double r = __builtin_fma(-k, LN2_HI, x);
  r = __builtin_fma(-k, LN2_LO, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  ki = (long long)k;
  return p;
}
// Fixed-point weights of log-likelihoods: q_i = floor(exp(lw_i) 2^(s - K)) with the INTEGER
// exponent K = rint(m log2 e) + 1 of the cloud's largest log-likelihood m -- a power of two, not exp(m) itself, so
// that a shard which only knows its OWN maximum m_r <= m can quantise at K_r <= K and the cloud's weights are those
// SHIFTED RIGHT by K - K_r, exactly: floor(floor(x) / 2^d) = floor(x / 2^d).  (rounds 1-5: q_i = floor(det_exp(lw_i -
// m) 2^s), which needed the global maximum BEFORE the first weight could be formed: one more collective per step.)
// exp(lw_i - K ln 2) <= 2^-1/2: the largest weight lies in (2^-3/2, 2^-1/2] of the scale instead of at 1 -- at most
// a bit and a half of the 43 bits a million particles leave.  Identical in oracle/mcl_oracle.c.
#define MCL_K_NONE (-(1ll << 40))   // no finite log-likelihood (m = -inf): uniform weights, as before
__device__ __forceinline__ long long weight_exponent(double m_lw) {
#pragma clang fp contract(off)
  if (m_lw == -__builtin_inf()) return MCL_K_NONE;
  if (!(m_lw >= -1.0e11)) return -(1ll << 38);   // (also NaN)
  if (!(m_lw <= 1.0e11)) return 1ll << 38;
  return (long long)__builtin_rint(m_lw * 1.44269504088896338700e+00) + 1;
}
__device__ __forceinline__ u64 quantise_log_weight(double lw, long long K, int s) {
  if (K == MCL_K_NONE) return 1ull << s;
  long long ki;
  const double p = det_exp_parts(lw, ki);
  const long long e = ki - K + (long long)s;
  if (p == 0.0 || e < -1000 || e > 62) return 0ull;   // (e <= s - 1 for every lw <= m: rint is monotone)
  return (u64)(p * __longlong_as_double((long long)((u64)(e + 1023) << 52)));
}
// a shard's weights (quantised at its own exponent) at the cloud's: shifted right by the difference of the exponents
__device__ __forceinline__ u64 shift_weight(u64 q, u32 d) { return d >= 64u ? 0ull : q >> d; }

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
struct u32x4 {
  u32 x, y, z, w;
};
__device__ __forceinline__ u32x4 philox4x32(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply per product: v_mad_u64_u32 instead of a v_mul_hi_u32 / v_mul_lo_u32 pair -- both
    //  quarter rate, so half the multiplier time of the ten rounds)
    const u64 p0 = (u64)0xD2511F53u * (u64)c0, p1 = (u64)0xCD9E8D57u * (u64)c2;
    const u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
    u32 n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}
// Deterministic log and sin/cos for the Box-Muller draws: fixed sequences of fma / mul / div / rint / bit ops,
// written identically in oracle/mcl_oracle.c (orc_det_log, orc_det_sincos2pi), so the NATIVE normals agree
// bit for bit between CPU and GPU -- and cost a fraction of ocml's log / sincos (whose large-argument paths
// and special cases these draws never need).  Accuracy: < 3e-16 relative (log), < 2e-16 absolute (sin, cos).
__device__ __forceinline__ double det_log(double x) {  // x a positive normal number
#pragma clang fp contract(off)
  const u64 bits = (u64)__double_as_longlong(x);
  long long e = (long long)(bits >> 52) - 1023;
  double m = __longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));  // [1, 2)
  if (m > 1.41421356237309514547) {
    m *= 0.5;
    e += 1;
  }
  const double s = (m - 1.0) / (m + 1.0);  // |s| <= 0.1716
  const double s2 = s * s;
  double p = 1.0 / 23.0;                   // atanh series: log m = 2 s (1 + s2/3 + s2^2/5 + ...)
  p = __builtin_fma(p, s2, 1.0 / 21.0);
  p = __builtin_fma(p, s2, 1.0 / 19.0);
  p = __builtin_fma(p, s2, 1.0 / 17.0);
  p = __builtin_fma(p, s2, 1.0 / 15.0);
  p = __builtin_fma(p, s2, 1.0 / 13.0);
  p = __builtin_fma(p, s2, 1.0 / 11.0);
  p = __builtin_fma(p, s2, 1.0 / 9.0);
  p = __builtin_fma(p, s2, 1.0 / 7.0);
  p = __builtin_fma(p, s2, 1.0 / 5.0);
  p = __builtin_fma(p, s2, 1.0 / 3.0);
  p = __builtin_fma(p, s2, 1.0);
  const double lm = 2.0 * s * p;
  const double ed = (double)e;
  return __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, lm));
}
// sin and cos of 2 pi u for u in [0, 1]
__device__ __forceinline__ void det_sincos2pi(double u, double& sn, double& cs) {
#pragma clang fp contract(off)
  const double k = __builtin_rint(u * 4.0);          // quarter turns, 0 .. 4
  const double r = __builtin_fma(-k, 0.25, u);       // exact: |r| <= 1/8
  const double th = r * 6.28318530717958647693;      // |th| <= pi/4
  const double t2 = th * th;
  double ps = -1.0 / 355687428096000.0;              // sin: th (1 - t2/3! + ... - t2^8/17!)
  ps = __builtin_fma(ps, t2, 1.0 / 1307674368000.0);
  ps = __builtin_fma(ps, t2, -1.0 / 6227020800.0);
  ps = __builtin_fma(ps, t2, 1.0 / 39916800.0);
  ps = __builtin_fma(ps, t2, -1.0 / 362880.0);
  ps = __builtin_fma(ps, t2, 1.0 / 5040.0);
  ps = __builtin_fma(ps, t2, -1.0 / 120.0);
  ps = __builtin_fma(ps, t2, 1.0 / 6.0);
  const double s0 = __builtin_fma(-(th * t2), ps, th);
  double pc = 1.0 / 6402373705728000.0;              // cos: 1 - t2/2! + ... + t2^9/18!
  pc = __builtin_fma(pc, t2, -1.0 / 20922789888000.0);
  pc = __builtin_fma(pc, t2, 1.0 / 87178291200.0);
  pc = __builtin_fma(pc, t2, -1.0 / 479001600.0);
  pc = __builtin_fma(pc, t2, 1.0 / 3628800.0);
  pc = __builtin_fma(pc, t2, -1.0 / 40320.0);
  pc = __builtin_fma(pc, t2, 1.0 / 720.0);
  pc = __builtin_fma(pc, t2, -1.0 / 24.0);
  pc = __builtin_fma(pc, t2, 0.5);
  const double c0 = __builtin_fma(-t2, pc, 1.0);
  const int q = (int)k & 3;
  sn = q == 0 ? s0 : (q == 1 ? c0 : (q == 2 ? -s0 : -c0));
  cs = q == 0 ? c0 : (q == 1 ? -s0 : (q == 2 ? -c0 : s0));
}
__device__ __forceinline__ void box_muller(u32 a, u32 b, double& n0, double& n1) {
  double u1 = ((double)a + 0.5) * (1.0 / 4294967296.0);
  double u2 = ((double)b + 0.5) * (1.0 / 4294967296.0);
  double r = sqrt(-2.0 * det_log(u1));   // IEEE sqrt: correctly rounded on both sides
  double s, c;
  det_sincos2pi(u2, s, c);
  n0 = r * c;
  n1 = r * s;
}

// ---------------------------------------------------------------- 128-bit helpers
// floor(C * N / T) and remainder for C < 2^64, N < 2^32, T >= 1 (quotient <= N when C <= T)
__device__ __forceinline__ void muldiv_u64(u64 C, u64 N, u64 T, u64& quo, u64& rem) {
  const u64 a_lo = C * N, a_hi = __umul64hi(C, N);
  // estimate from doubles: relative error ~2^-52, quotient < 2^33 -> off by at most 1
  double est = ((double)C * (double)N) / (double)T;
  u64 a = (u64)est;
  // R = A - a*T as a signed 128-bit value
  u64 p_lo = a * T, p_hi = __umul64hi(a, T);
  u64 r_lo = a_lo - p_lo;
  u64 r_hi = a_hi - p_hi - (a_lo < p_lo ? 1ull : 0ull);
  // bring R into [0, T)
  for (int it = 0; it < 4; ++it) {
    if ((long long)r_hi < 0) {  // R < 0 : a too big
      --a;
      u64 t = r_lo + T;
      r_hi += (t < r_lo) ? 1ull : 0ull;
      r_lo = t;
    } else if (r_hi != 0 || r_lo >= T) {  // R >= T : a too small
      ++a;
      u64 t = r_lo - T;
      r_hi -= (r_lo < T) ? 1ull : 0ull;
      r_lo = t;
    } else {
      break;
    }
  }
  quo = a;
  rem = r_lo;
}
// The same with the estimate's factor N / T (as a double) formed once by the caller: a block that divides thousands of
// numbers by the same total pays one fp64 division, not one per number (the division and two of the three u64 -> f64
// conversions were a third of the routine).  The estimate may be off by one more unit in the last place -- the
// correction below brings the remainder into [0, T) either way, so quotient and remainder are EXACT and identical.
__device__ __forceinline__ void muldiv_u64(u64 C, u64 N, u64 T, double n_over_t, u64& quo, u64& rem) {
  const u64 a_lo = C * N, a_hi = __umul64hi(C, N);
  u64 a = (u64)((double)C * n_over_t);
  u64 p_lo = a * T, p_hi = __umul64hi(a, T);
  u64 r_lo = a_lo - p_lo;
  u64 r_hi = a_hi - p_hi - (a_lo < p_lo ? 1ull : 0ull);
  for (int it = 0; it < 6; ++it) {
    if ((long long)r_hi < 0) {  // R < 0 : a too big
      --a;
      u64 t = r_lo + T;
      r_hi += (t < r_lo) ? 1ull : 0ull;
      r_lo = t;
    } else if (r_hi != 0 || r_lo >= T) {  // R >= T : a too small
      ++a;
      u64 t = r_lo - T;
      r_hi -= (r_lo < T) ? 1ull : 0ull;
      r_lo = t;
    } else {
      break;
    }
  }
  quo = a;
  rem = r_lo;
}
// (r << 53) > U * T   for r < T <= 2^64-1, U < 2^53.  Bit 63 of U selects ">=" instead of ">": that is the
// only difference between naive_resample (resampling.py:116-131: `while resample_id > cdf[ind]`, i.e. the
// first j with cdf_j >= pos) and systematic_resample (:161-167: the first j with pos < cs_j).
#define MCL_U53_NAIVE (1ull << 63)
__device__ __forceinline__ bool shl53_gt_mul(u64 r, u64 U, u64 T) {
  const bool or_equal = (U >> 63) != 0ull;
  U &= ~MCL_U53_NAIVE;
  const u64 l_hi = r >> 11, l_lo = r << 53;
  const u64 m_lo = U * T, m_hi = __umul64hi(U, T);
  return l_hi > m_hi || (l_hi == m_hi && (or_equal ? l_lo >= m_lo : l_lo > m_lo));
}
