// mcl_resample_alt.h -- the reference's other resamplers on the GPU (resampling.py):
//   stratified  :80-114, multinomial :171-194  -- parallel, on the exact integer CDF
//   residual    :27-76  -- what auv_pf.py:182 actually calls.  Reproduced LITERALLY, including
//                FilterPy's sign-flipped residual and numpy's carried-bounds searchsorted on the
//                resulting non-monotone cumsum.  Those two steps are sequential by
//                construction, so they run on one lane: a reference-compatibility mode for the
//                particle counts the node itself can run (it is O(N^2) in keep/lost/dupes), not
//                the production scheme.
// All three produce an explicit ancestor vector idx[]; the generic keep/lost/dupes reassign for an
// arbitrary (unsorted) idx follows auv_pf.py:183-198 with atomics + scans.
#pragma once
#include "mcl_kernels.h"

// ------------------------------------------------------------------ uniforms
// U53_i = floor(u_i * 2^53) from replayed doubles, or Philox (purpose 4, counter = draw index)
__global__ void __launch_bounds__(MCL_BLOCK) k_make_u53(const double* __restrict__ replay, long long n,
                                                        u32 k0, u32 k1, u32 step, u64* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    if (replay) {
      out[i] = (u64)(replay[i] * 9007199254740992.0);
    } else {
      u32x4 o = philox4x32((u32)i, 0u, step, 4u, k0, k1);
      out[i] = ((u64)(o.x >> 5) << 26) | (u64)(o.y >> 6);
    }
  }
}

// C = inclusive scan of q (+ tile offsets); needed by the explicit-position schemes
__global__ void __launch_bounds__(MCL_BLOCK) k_u64_scan(const u64* __restrict__ q, long long n,
                                                        const u64* __restrict__ tile_off, u64* __restrict__ c) {
  __shared__ u64 sh[16];
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE + (long long)threadIdx.x * MCL_SCAN_ITEMS;
    u64 v[MCL_SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) v[k] = (base + k < n) ? q[base + k] : 0ull;
    tile_scan_blocked(v, sh);
    const u64 off = tile_off[tile];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k)
      if (base + k < n) c[base + k] = v[k] + off;
    __syncthreads();
  }
}

// 192-bit helpers: L = (U + i 2^53) * T ;  R_j = (C_j * N) << 53
struct u192 {
  u64 w0, w1, w2;
};
__device__ __forceinline__ bool lt192(const u192& a, const u192& b) {
  if (a.w2 != b.w2) return a.w2 < b.w2;
  if (a.w1 != b.w1) return a.w1 < b.w1;
  return a.w0 < b.w0;
}
__device__ __forceinline__ u192 pos_times_total(u64 U, u64 i, u64 T) {
  const u64 a_lo = (i << 53) | U, a_hi = i >> 11;
  u192 r;
  r.w0 = a_lo * T;
  const u64 p0_hi = __umul64hi(a_lo, T);
  const u64 p1_lo = a_hi * T, p1_hi = __umul64hi(a_hi, T);
  r.w1 = p0_hi + p1_lo;
  r.w2 = p1_hi + (r.w1 < p0_hi ? 1ull : 0ull);
  return r;
}
__device__ __forceinline__ u192 cdf_times_n_shl53(u64 C, u64 N) {
  const u64 m_lo = C * N, m_hi = __umul64hi(C, N);
  u192 r;
  r.w0 = m_lo << 53;
  r.w1 = (m_lo >> 11) | (m_hi << 53);
  r.w2 = m_hi >> 11;
  return r;
}

// stratified: idx_i = min{ j : (U_i + i 2^53) T < C_j N 2^53 }   (pos_i = (u_i + i)/N < cs_j)
__global__ void __launch_bounds__(MCL_BLOCK) k_stratified_idx(const u64* __restrict__ c, const u64* __restrict__ u53,
                                                              long long n, int* __restrict__ idx) {
  const u64 T = c[n - 1];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const u192 L = pos_times_total(u53[i], (u64)i, T);
    long long lo = 0, hi = n;
    while (lo < hi) {
      const long long mid = lo + ((hi - lo) >> 1);
      if (lt192(L, cdf_times_n_shl53(c[mid], (u64)n)))
        hi = mid;
      else
        lo = mid + 1;
    }
    idx[i] = (int)(lo < n ? lo : n - 1);
  }
}

// multinomial: idx_i = searchsorted(cs, u_i, 'left') = min{ j : C_j 2^53 >= U_i T }
__global__ void __launch_bounds__(MCL_BLOCK) k_multinomial_idx(const u64* __restrict__ c, const u64* __restrict__ u53,
                                                               long long n, int* __restrict__ idx) {
  const u64 T = c[n - 1];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const u64 U = u53[i];
    const u64 m_lo = U * T, m_hi = __umul64hi(U, T);
    long long lo = 0, hi = n;
    while (lo < hi) {
      const long long mid = lo + ((hi - lo) >> 1);
      const u64 cj = c[mid];
      const u64 l_hi = cj >> 11, l_lo = cj << 53;
      const bool less = l_hi < m_hi || (l_hi == m_hi && l_lo < m_lo);  // cs_j < u
      if (less)
        lo = mid + 1;
      else
        hi = mid;
    }
    idx[i] = (int)(lo < n ? lo : n - 1);
  }
}

// ------------------------------------------------------------------ generic keep/lost/dupes
// cnt[v] = #occurrences, first[v] = smallest position i with idx_i == v
__global__ void __launch_bounds__(MCL_BLOCK) k_idx_hist(const int* __restrict__ idx, long long n,
                                                        u32* __restrict__ cnt, u32* __restrict__ first) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = idx[i];
    atomicAdd(&cnt[v], 1u);
    atomicMin(&first[v], (u32)i);
  }
}
// flags: mode 0 -> [cnt_i == 0] (lost slots); mode 1 -> [first[idx_i] != i] (dupes entries)
__global__ void __launch_bounds__(MCL_BLOCK) k_flags(const int* __restrict__ idx, const u32* __restrict__ cnt,
                                                     const u32* __restrict__ first, long long n, int mode,
                                                     u32* __restrict__ flags) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    flags[i] = mode == 0 ? (cnt[i] == 0u ? 1u : 0u) : (first[idx[i]] != (u32)i ? 1u : 0u);
}
__global__ void __launch_bounds__(MCL_BLOCK) k_u32_tile_sums(const u32* __restrict__ in, long long n,
                                                             u32* __restrict__ tile_sum) {
  __shared__ u32 sh[16];
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE;
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
      long long j = base + (long long)k * MCL_BLOCK + threadIdx.x;
      if (j < n) acc += in[j];
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) tile_sum[tile] = acc;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(MCL_BLOCK) k_u32_scan(const u32* __restrict__ in, long long n,
                                                        const u32* __restrict__ tile_off, u32* __restrict__ out) {
  __shared__ u32 sh[16];
  for (long long tile = blockIdx.x; tile * MCL_SCAN_TILE < n; tile += gridDim.x) {
    const long long base = tile * MCL_SCAN_TILE + (long long)threadIdx.x * MCL_SCAN_ITEMS;
    u32 v[MCL_SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
    tile_scan_blocked(v, sh);
    const u32 off = tile_off[tile];
#pragma unroll
    for (int k = 0; k < MCL_SCAN_ITEMS; ++k)
      if (base + k < n) out[base + k] = v[k] + off;
    __syncthreads();
  }
}
// dupes[rank] = idx_i for the flagged entries (order of idx preserved: auv_pf.py:185-187)
__global__ void __launch_bounds__(MCL_BLOCK) k_compact_dupes(const int* __restrict__ idx, const u32* __restrict__ flags,
                                                             const u32* __restrict__ fcum, long long n,
                                                             int* __restrict__ dupes) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    if (flags[i]) dupes[fcum[i] - 1u] = idx[i];
}
struct ReassignIdxArgs {
  StatePtrs src, dst;
  long long n;
  NoiseArgs nz;
};
__global__ void __launch_bounds__(MCL_BLOCK) k_reassign_idx(ReassignIdxArgs a, const u32* __restrict__ cnt,
                                                            const u32* __restrict__ zcum, const int* __restrict__ dupes,
                                                            const double* __restrict__ replay) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    long long src = i;
    if (cnt[i] == 0u) src = dupes[zcum[i] - 1u];
    double z[6];
    if (replay) {
#pragma unroll
      for (int c = 0; c < 6; ++c) z[c] = replay[i * 6 + c];
    } else {
      native_normals6(a.nz.gid0 + i, a.nz, z);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) a.dst.c[c][i] = a.src.c[c][src] + a.nz.sq[c] * z[c];
  }
}

// ------------------------------------------------------------------ residual (literal restatement)
// linear weights as the reference holds them: w = exp(lw) + 1e-200  /  exp(lw - max)
__global__ void __launch_bounds__(MCL_BLOCK) k_linear_weights(const double* __restrict__ lw, long long n,
                                                              const double* __restrict__ m_lw, int mode,
                                                              double* __restrict__ w) {
  const double m = m_lw[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    w[i] = mode == 0 ? exp(lw[i]) + 1.e-200 : (mode == 1 ? exp(lw[i] - m) : lw[i]);
}

// numpy's pairwise sum of one contiguous chunk (loops_utils.h pairwise_sum), recursion depth <= 7
__device__ double np_pairwise(const double* a, long long n) {
  if (n < 8) {
    double r = 0.0;
    for (long long i = 0; i < n; ++i) r += a[i];
    return r;
  } else if (n <= 128) {
    double r[8];
    long long i;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
  }
}
// numpy add.reduce: 8192-element chunks, each pairwise-summed (one thread per chunk) ...
__global__ void __launch_bounds__(64) k_np_chunk_sums(const double* __restrict__ w, long long n,
                                                      double* __restrict__ chunk) {
#pragma clang fp contract(off)
  const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long base = c * 8192;
  if (base < n) chunk[c] = np_pairwise(w + base, (n - base) < 8192 ? (n - base) : 8192);
}
// ... accumulated left to right; then weights /= sum (auv_pf.py:172)
__global__ void __launch_bounds__(64) k_np_sum_final(const double* __restrict__ chunk, long long nchunks,
                                                     double* __restrict__ out) {
#pragma clang fp contract(off)
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double r = 0.0;
    for (long long c = 0; c < nchunks; ++c) r += chunk[c];
    out[0] = r;
  }
}
// w /= S ; copies_i = floor(N w_i) (resampling.py:61)
__global__ void __launch_bounds__(MCL_BLOCK) k_residual_copies(double* __restrict__ w, long long n,
                                                               const double* __restrict__ S, u32* __restrict__ copies) {
#pragma clang fp contract(off)
  const double s = S[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double wi = w[i] / s;
    w[i] = wi;
    const double c = floor((double)n * wi);
    copies[i] = c > 0.0 ? (c < 4294967295.0 ? (u32)c : 0xffffffffu) : 0u;
  }
}
// head of the index vector: i repeated copies_i times (resampling.py:63-66); ccum = inclusive scan
__global__ void __launch_bounds__(MCL_BLOCK) k_residual_head(const u32* __restrict__ ccum, long long n, long long k,
                                                             int* __restrict__ idx) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < k && t < n;
       t += (long long)gridDim.x * blockDim.x) {
    long long lo = 0, hi = n;  // first i with ccum_i > t
    while (lo < hi) {
      const long long mid = lo + ((hi - lo) >> 1);
      if (ccum[mid] > (u32)t)
        hi = mid;
      else
        lo = mid + 1;
    }
    idx[t] = (int)lo;
  }
}
// residual = w - copies (sic); residual /= builtins.sum(residual); cumsum; cs[-1] = 1
// (resampling.py:70-73).  Sequential left-to-right fp64 sums: one lane.
__global__ void __launch_bounds__(64) k_residual_cumsum(const double* __restrict__ w, const u32* __restrict__ copies,
                                                        long long n, double* __restrict__ cs) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (long long i = 0; i < n; ++i) s += w[i] - (double)copies[i];
  double acc = 0.0;
  for (long long i = 0; i < n; ++i) {
    acc += (w[i] - (double)copies[i]) / s;
    cs[i] = acc;
  }
  cs[n - 1] = 1.0;
}
// np.searchsorted(cs, keys) exactly as numpy's binsearch<left> runs it: bounds carried from key to
// key (matters because cs is NOT monotone here).  Sequential: one lane.
__global__ void __launch_bounds__(64) k_residual_searchsorted(const double* __restrict__ cs, long long n,
                                                              const u64* __restrict__ u53, long long nkeys,
                                                              int* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || nkeys <= 0) return;
  long long min_idx = 0, max_idx = n;
  double last = (double)u53[0] * (1.0 / 9007199254740992.0);
  for (long long t = 0; t < nkeys; ++t) {
    const double kv = (double)u53[t] * (1.0 / 9007199254740992.0);
    if (last < kv) {
      max_idx = n;
    } else {
      min_idx = 0;
      max_idx = (max_idx < n) ? (max_idx + 1) : n;
    }
    last = kv;
    while (min_idx < max_idx) {
      const long long mid = min_idx + ((max_idx - min_idx) >> 1);
      if (cs[mid] < kv)
        min_idx = mid + 1;
      else
        max_idx = mid;
    }
    out[t] = (int)min_idx;
  }
}
