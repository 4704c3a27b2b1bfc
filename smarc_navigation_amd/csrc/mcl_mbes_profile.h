// mcl_mbes_profile.h -- profile-marching MBES update for STRUCTURED meshes (triangulated regular
// height grids, mcl_mesh.h) -- the beams of one ping lie in ONE plane, so instead of casting 512
// independent 3-D rays per particle the wavefront
//   1. builds the exact intersection polyline of the fan plane with the triangulated surface
//      (lanes = grid-line strips along the swath; per strip at most four vertices in a fixed order:
//      the crossing of the strip's near grid line, the diagonal of the first cell, the grid line
//      between the two cells, the diagonal of the second cell), expressed in in-plane coordinates
//      (a along the across-track axis, b downward), ordered by a, in LDS;
//   2. prefix-minimises cot(vertex) = b/|a| outward from nadir on both sides (on a height field the
//      first hit of a beam at angle t is the first vertex outward with prefix-min <= cot t: beams
//      further out can only hit further out, occlusion included);
//   3. answers each beam with a binary search on that array and one 2-D line/segment intersection.
// ~0.8k VALU wave-instructions per particle instead of ~2.1k for the per-ray traversal, uniform
// control flow.  Exact for piecewise-planar surfaces; groups that are not eligible (tile clipped or
// too large, fan plane tilted too far from vertical, a strip that does not resolve, too many
// vertices) are appended to the worklist and handled by k_mbes_cast<2,*,1>.
#pragma once
#include "mcl_mbes.h"

#define PROF_MAXV 192          // polyline vertices kept per particle
#define PROF_MIN_WAVES 6       // 50 KiB LDS per workgroup -> 3 workgroups per CU

struct ProfGeom {   // per-particle plane constants in tile-local, major/minor axis terms (wave-uniform)
  float uM, um;     // sensor origin: major / minor coordinate (cells)
  float oz;
  float kM, km, kz; // signed distance of a node: d = kM (M - uM) + km (m - um) + kz (h - oz)
  float c1M, c1m, c1z, c2M, c2m, c2z;  // in-plane axes in (major, minor, z) metres
  float slope;      // d(minor)/d(major) of the straight ground trace
  float shift;      // minor offset of the trace at the reference depth: -kz (href - oz) / km
  int sM;           // strip direction along the major axis so that `a` increases with the strip index
  int strideM, stridem;  // LDS strides of the major / minor axis
  int nM, nm;       // nodes along major / minor
};

// crossing of the fan plane with grid line M = m (a line of nodes along the minor axis).
// Returns false if no sign change is found next to the straight-trace guess.
__device__ __forceinline__ bool prof_line_cross(const float* __restrict__ tile, const ProfGeom& g, int m, float& y,
                                                float& h) {
  if (m < 0 || m >= g.nM) return false;
  const float yg = g.um + g.shift + ((float)m - g.uM) * g.slope;  // d = 0 at the reference depth
  const int jg = (int)floorf(yg);
  if (jg < 1 || jg + 2 >= g.nm) return false;
  const float* col = tile + m * g.strideM;
  const float dm = g.kM * ((float)m - g.uM);
  float hh[4], dd[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hh[q] = col[(jg - 1 + q) * g.stridem];
    dd[q] = dm + g.km * ((float)(jg - 1 + q) - g.um) + g.kz * (hh[q] - g.oz);
  }
  // sign change in (jg, jg+1) preferred, then its neighbours; d == 0 counts as positive
  const bool s0 = dd[0] >= 0.f, s1 = dd[1] >= 0.f, s2 = dd[2] >= 0.f, s3 = dd[3] >= 0.f;
  int q;
  if (s1 != s2)
    q = 1;
  else if (s0 != s1)
    q = 0;
  else if (s2 != s3)
    q = 2;
  else
    return false;
  const float dA = q == 0 ? dd[0] : (q == 1 ? dd[1] : dd[2]), dB = q == 0 ? dd[1] : (q == 1 ? dd[2] : dd[3]);
  const float hA = q == 0 ? hh[0] : (q == 1 ? hh[1] : hh[2]), hB = q == 0 ? hh[1] : (q == 1 ? hh[2] : hh[3]);
  const float f = dA * fast_rcp(dA - dB);
  y = (float)(jg - 1 + q) + f;
  h = hA + f * (hB - hA);
  return true;
}

// in-plane coordinates of a surface point given in (major, minor, height)
__device__ __forceinline__ float2 prof_ab(const ProfGeom& g, float pM, float pm, float h, float res) {
  const float dM = (pM - g.uM) * res, dmn = (pm - g.um) * res, dz = h - g.oz;
  return make_float2(dM * g.c1M + dmn * g.c1m + dz * g.c1z, -(dM * g.c2M + dmn * g.c2m + dz * g.c2z));
}

// crossing of the plane with the diagonal of cell (cM, cj); returns false if the diagonal is not crossed
__device__ __forceinline__ bool prof_diag_cross(const float* __restrict__ tile, const ProfGeom& g, int cM, int cj,
                                                float res, float2& ab) {
  const float* p = tile + cM * g.strideM + cj * g.stridem;
  const float h00 = p[0], h10 = p[g.strideM], h01 = p[g.stridem], h11 = p[g.strideM + g.stridem];
  const bool d1 = (__float_as_uint(h00) & 1u) != 0u;  // 0: (0,0)-(1,1), 1: (1,0)-(0,1); symmetric under transposition
  // endpoints P -> Q in (major, minor) cell offsets
  const float pMo = d1 ? 1.f : 0.f, qMo = d1 ? 0.f : 1.f;  // P = (pMo, 0), Q = (qMo, 1)
  const float hP = d1 ? h10 : h00, hQ = d1 ? h01 : h11;
  const float dP = g.kM * ((float)cM + pMo - g.uM) + g.km * ((float)cj - g.um) + g.kz * (hP - g.oz);
  const float dQ = g.kM * ((float)cM + qMo - g.uM) + g.km * ((float)cj + 1.f - g.um) + g.kz * (hQ - g.oz);
  if ((dP >= 0.f) == (dQ >= 0.f)) return false;
  const float f = dP * fast_rcp(dP - dQ);
  ab = prof_ab(g, (float)cM + pMo + f * (qMo - pMo), (float)cj + f, hP + f * (hQ - hP), res);
  return true;
}

template <bool EXPECT_ONLY>
__global__ void __launch_bounds__(MBES_THREADS, PROF_MIN_WAVES) k_mbes_profile(MbesArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[MBES_TILE_FLOATS];
  __shared__ float2 poly[MBES_WAVES][PROF_MAXV];
  __shared__ float pmin[MBES_WAVES][PROF_MAXV];
  __shared__ float red[5][MBES_WAVES];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ngroups = (a.n + MBES_WAVES - 1) / MBES_WAVES;
  const float inv_res = (float)a.inv_res;
  const float INF = __builtin_inff();

  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long i = grp * MBES_WAVES + w;
    const bool valid = i < a.n;
    MbesPose P;
    if (valid) {
      const MbesPose Pv = a.pose[i];
      P.um = uniform_f64(Pv.um);
      P.vm = uniform_f64(Pv.vm);
      P.oz = uniform_f32(Pv.oz);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        P.c1[r] = uniform_f32(Pv.c1[r]);
        P.c2[r] = uniform_f32(Pv.c2[r]);
      }
    }
    // ---- footprint of the fan (global cell units) from its two extreme beams
    float umin = INF, umax = -INF, vmin = INF, vmax = -INF;
    bool simple = valid && a.b_lo >= 0;
    if (valid) {
      umin = umax = (float)P.um;
      vmin = vmax = (float)P.vm;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float2 sc = a.beam_sc[k == 0 ? max(a.b_lo, 0) : max(a.b_hi, 0)];
        const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
        const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
        const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
        const float t_end = fmaxf((a.zmin_map - P.oz) * fast_rcp(dz), 0.f);
        simple = simple && dz < -1e-4f && t_end <= a.r_max;
        const float ue = (float)P.um + t_end * dx * inv_res, ve = (float)P.vm + t_end * dy * inv_res;
        umin = fminf(umin, ue);
        umax = fmaxf(umax, ue);
        vmin = fminf(vmin, ve);
        vmax = fmaxf(vmax, ve);
      }
    }
    __syncthreads();  // previous group's LDS fully consumed
    if (lane == 0) {
      red[0][w] = umin;
      red[1][w] = umax;
      red[2][w] = vmin;
      red[3][w] = vmax;
      red[4][w] = (!valid || simple) ? 1.f : 0.f;
    }
    __syncthreads();
    float a0 = red[0][lane & (MBES_WAVES - 1)], a1 = red[1][lane & (MBES_WAVES - 1)];
    float b0 = red[2][lane & (MBES_WAVES - 1)], b1 = red[3][lane & (MBES_WAVES - 1)];
    float okf = red[4][lane & (MBES_WAVES - 1)];
#pragma unroll
    for (int o = MBES_WAVES / 2; o > 0; o >>= 1) {
      a0 = fminf(a0, __shfl_xor(a0, o, 64));
      a1 = fmaxf(a1, __shfl_xor(a1, o, 64));
      b0 = fminf(b0, __shfl_xor(b0, o, 64));
      b1 = fmaxf(b1, __shfl_xor(b1, o, 64));
      okf = fminf(okf, __shfl_xor(okf, o, 64));
    }
    // node window with two cells of margin (the polyline is built one strip beyond the footprint)
    const int wx0 = (int)floorf(a0) - 3, wy0 = (int)floorf(b0) - 3;
    const int wx1 = (int)floorf(a1) + 4, wy1 = (int)floorf(b1) + 4;
    const bool have = a0 <= a1;
    int tx0 = max(wx0, 0), ty0 = max(wy0, 0);
    const int tx1 = min(wx1, a.nx - 1), ty1 = min(wy1, a.ny - 1);
    int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
    const bool clipped = wx0 < 0 || wy0 < 0 || wx1 > a.nx - 1 || wy1 > a.ny - 1;
    bool eligible = have && okf > 0.5f && !clipped && tw >= 2 && th >= 2 && (long long)tw * th <= MBES_TILE_FLOATS;
    tx0 = __builtin_amdgcn_readfirstlane(tx0);
    ty0 = __builtin_amdgcn_readfirstlane(ty0);
    tw = __builtin_amdgcn_readfirstlane(tw);
    th = __builtin_amdgcn_readfirstlane(th);
    eligible = __builtin_amdgcn_readfirstlane(eligible ? 1 : 0) != 0;
    if (!eligible) {  // block-uniform
      if (have && threadIdx.x == 0) a.worklist[atomicAdd(a.work_count, 1)] = (int)grp;
      if (!have && valid) {  // cannot happen for a valid particle; keep the outputs defined
        if (!EXPECT_ONLY && lane == 0) a.lw[i] = 0.0;
      }
      continue;
    }
    // ---- stage the height tile
    for (int ix = w; ix < tw; ix += MBES_WAVES) {
      const float* src = a.grid + (size_t)(tx0 + ix) * a.ny + ty0;
      for (int iy = lane; iy < th; iy += 64) tile[ix * th + iy] = src[iy];
    }
    __syncthreads();

    // ---- per-particle plane geometry (wave-uniform)
    bool fail = false;
    int nv = 0, k0 = 0;
    ProfGeom g;
    if (valid) {
      // plane normal n = c1 x c2 (metres); node distance d = n . (p - O)
      const float nx = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1];
      const float ny = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2];
      const float nz = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
      // ground trace direction (-ny, nx); major axis = the larger component
      const bool xmajor = fabsf(ny) >= fabsf(nx);
      const float u0 = (float)(P.um - (double)tx0), v0 = (float)(P.vm - (double)ty0);
      g.oz = P.oz;
      g.kz = nz;
      if (xmajor) {
        g.uM = u0;
        g.um = v0;
        g.kM = nx * a.res;
        g.km = ny * a.res;
        g.c1M = P.c1[0];
        g.c1m = P.c1[1];
        g.c2M = P.c2[0];
        g.c2m = P.c2[1];
        g.slope = -nx * fast_rcp(ny);  // along the trace: d(y)/d(x) = nx / -ny
        g.strideM = th;
        g.stridem = 1;
        g.nM = tw;
        g.nm = th;
      } else {
        g.uM = v0;
        g.um = u0;
        g.kM = ny * a.res;
        g.km = nx * a.res;
        g.c1M = P.c1[1];
        g.c1m = P.c1[0];
        g.c2M = P.c2[1];
        g.c2m = P.c2[0];
        g.slope = -ny * fast_rcp(nx);
        g.strideM = 1;
        g.stridem = th;
        g.nM = th;
        g.nm = tw;
      }
      g.c1z = P.c1[2];
      g.c2z = P.c2[2];
      g.shift = -g.kz * (0.5f * (a.zmin_map + a.zmax_map) - g.oz) * fast_rcp(g.km);
      // `a` grows with the major coordinate when the across-track axis c1 points along +major
      const float c1_along = g.c1M + g.c1m * g.slope;
      g.sM = c1_along >= 0.f ? 1 : -1;
      // the plane must be close to vertical and cut the minor axis steeply enough for unique crossings
      const float nh = sqrtf(nx * nx + ny * ny);
      if (!(fabsf(nz) <= 0.30f * nh) || !(fabsf(g.km) >= 0.5f * a.res * nh)) fail = true;
      // ---- strips: grid lines from one footprint end to the other (+1 line of margin each side)
      const float Mlo = xmajor ? umin - (float)tx0 : vmin - (float)ty0;
      const float Mhi = xmajor ? umax - (float)tx0 : vmax - (float)ty0;
      // grid lines floor(Mlo) .. floor(Mhi)+1 bracket every hit (the footprint is taken at depth z_min)
      const int m_first = g.sM > 0 ? (int)floorf(Mlo) : (int)floorf(Mhi) + 1;
      const int nstrips = (int)floorf(Mhi) - (int)floorf(Mlo) + 1;
      int base = 0;
      for (int s0 = 0; s0 < nstrips && !fail; s0 += 64) {
        const int s = s0 + lane;
        const bool live = s < nstrips;
        const int m = m_first + g.sM * s;   // near line of this strip; far line = m + sM
        float y0 = 0.f, h0 = 0.f, y1 = 0.f, h1 = 0.f;
        bool ok = true;
        // lane s needs the crossings of lines m and m + sM: the latter is lane s+1's near line
        bool ok0 = false;
        if (s <= nstrips) ok0 = prof_line_cross(tile, g, m, y0, h0);
        y1 = __shfl_down(y0, 1, 64);
        h1 = __shfl_down(h0, 1, 64);
        bool ok1 = __shfl_down(ok0 ? 1 : 0, 1, 64) != 0;
        if (lane == 63 && live) ok1 = prof_line_cross(tile, g, m + g.sM, y1, h1);
        ok = ok0 && ok1;
        // up to four vertices in a fixed order: near-line crossing, first cell's diagonal, the grid
        // line between the two cells, second cell's diagonal (named registers: no runtime indexing)
        float2 v0 = make_float2(0.f, 0.f), vD0 = v0, vL = v0, vD1 = v0;
        int f1 = 0, f2 = 0, f3 = 0;
        int cnt = 0;
        if (live && ok) {
          const int cM = min(m, m + g.sM);
          const float dirm = y1 >= y0 ? 1.f : -1.f;
          const int jc0 = (int)floorf(y0 + 1e-5f * dirm), jc1 = (int)floorf(y1 - 1e-5f * dirm);
          if (abs(jc1 - jc0) > 1 || jc0 < 0 || jc1 < 0 || jc0 + 1 >= g.nm || jc1 + 1 >= g.nm) {
            ok = false;
          } else {
            v0 = prof_ab(g, (float)m, y0, h0, a.res);
            f1 = prof_diag_cross(tile, g, cM, jc0, a.res, vD0) ? 1 : 0;
            if (jc1 != jc0) {
              // the grid line between the two cells: edge (cM, jl) - (cM+1, jl)
              const int jl = max(jc0, jc1);
              const float hA = tile[cM * g.strideM + jl * g.stridem], hB = tile[(cM + 1) * g.strideM + jl * g.stridem];
              const float dmn = g.km * ((float)jl - g.um);
              const float dA = g.kM * ((float)cM - g.uM) + dmn + g.kz * (hA - g.oz);
              const float dB = g.kM * ((float)cM + 1.f - g.uM) + dmn + g.kz * (hB - g.oz);
              if ((dA >= 0.f) == (dB >= 0.f)) {
                ok = false;
              } else {
                const float f = dA * fast_rcp(dA - dB);
                vL = prof_ab(g, (float)cM + f, (float)jl, hA + f * (hB - hA), a.res);
                f2 = 1;
                f3 = prof_diag_cross(tile, g, cM, jc1, a.res, vD1) ? 1 : 0;
              }
            }
            cnt = 1 + f1 + f2 + f3;
          }
        }
        if (live && !ok) cnt = -1;
        // any failed strip fails the particle
        const bool any_bad = __builtin_amdgcn_readfirstlane(__any(cnt < 0) ? 1 : 0) != 0;
        if (any_bad) {
          fail = true;
          break;
        }
        // travelling toward -major reverses the order inside a strip relative to increasing major, but the
        // strips themselves are visited in the direction of increasing `a`, and so are the vertices.
        int off = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(off, o, 64);
          if (lane >= o) off += v;
        }
        const int total = __shfl(off, 63, 64);
        off -= cnt;
        if (base + total > PROF_MAXV) {
          fail = true;
          break;
        }
        if (cnt > 0) {
          float2* dst = &poly[w][base + off];
          dst[0] = v0;
          if (f1) dst[1] = vD0;
          if (f2) dst[1 + f1] = vL;
          if (f3) dst[1 + f1 + f2] = vD1;
        }
        base += total;
      }
      nv = base;
      if (nv < 2) fail = true;
    }
    // ---- group vote: one failing particle defers the whole workgroup
    __syncthreads();
    if (lane == 0) red[4][w] = fail ? 0.f : 1.f;
    __syncthreads();
    float okg = red[4][lane & (MBES_WAVES - 1)];
#pragma unroll
    for (int o = MBES_WAVES / 2; o > 0; o >>= 1) okg = fminf(okg, __shfl_xor(okg, o, 64));
    if (!(okg > 0.5f)) {
      if (threadIdx.x == 0) a.worklist[atomicAdd(a.work_count, 1)] = (int)grp;
      continue;
    }
    if (!valid) continue;

    // ---- nadir split and outward prefix-min of cot = b / |a|
    {
      int c = 0;
      for (int k = lane; k < nv; k += 64) c += poly[w][k].x <= 0.f ? 1 : 0;
      k0 = wave_sum(c);
      k0 = __shfl(k0, 0, 64);
      // right side: k0 .. nv-1 forward ; left side: k0-1 .. 0 backward
      float carry = INF;
      for (int c0 = k0; c0 < nv; c0 += 64) {
        const int k = c0 + lane;
        float v = INF;
        if (k < nv) {
          const float2 p = poly[w][k];
          v = p.y * fast_rcp(fmaxf(fabsf(p.x), 1e-12f));
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float u = __shfl_up(v, o, 64);
          if (lane >= o) v = fminf(v, u);
        }
        v = fminf(v, carry);
        if (k < nv) pmin[w][k] = v;
        carry = __shfl(v, 63, 64);
      }
      carry = INF;
      for (int c0 = k0 - 1; c0 >= 0; c0 -= 64) {
        const int k = c0 - lane;
        float v = INF;
        if (k >= 0) {
          const float2 p = poly[w][k];
          v = p.y * fast_rcp(fmaxf(fabsf(p.x), 1e-12f));
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float u = __shfl_up(v, o, 64);
          if (lane >= o) v = fminf(v, u);
        }
        v = fminf(v, carry);
        if (k >= 0) pmin[w][k] = v;
        carry = __shfl(v, 63, 64);
      }
    }
    // (LDS writes of this wave are visible to itself: same wave, in-order LDS)

    // ---- beams: lane l owns the K consecutive beams [l K, (l+1) K).  With ascending beam angles the hit
    // vertex moves monotonically from beam to beam on each side of nadir, so after one binary search the
    // index is walked; unsorted angle lists fall back to a search per beam.
    float acc = 0.f;
    int nvalid = 0;
    const int K = (a.n_beams + 63) >> 6;
    int rr = -1, rl = -1;  // cursors: right k in [k0, nv], left r = k0-1-k in [0, k0]
    for (int kk = 0; kk < K; ++kk) {
      const int b = lane * K + kk;
      if (b >= a.n_beams) break;
      const float2 sc = a.beam_sc[b];
      float e = a.r_max;
      if (sc.y > 0.f) {
        const float target = sc.y * fast_rcp(fmaxf(fabsf(sc.x), 1e-12f));  // cot of the beam angle
        int k = -1, kin = -1;
        if (sc.x >= 0.f) {
          if (rr < 0 || !a.sorted) {  // first k in [k0, nv) with pmin[k] <= target
            int lo = k0, hi = nv;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (pmin[w][mid] <= target)
                hi = mid;
              else
                lo = mid + 1;
            }
            rr = lo;
          } else {  // angles ascending: the target shrinks, the index can only grow
            while (rr < nv && pmin[w][rr] > target) ++rr;
          }
          if (rr < nv) {
            k = rr;
            kin = rr - 1;
          }
        } else {
          if (rl < 0 || !a.sorted) {  // first r in [0, k0) with pmin[k0-1-r] <= target
            int lo = 0, hi = k0;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (pmin[w][k0 - 1 - mid] <= target)
                hi = mid;
              else
                lo = mid + 1;
            }
            rl = lo;
          } else {  // angles ascending on the left: the target grows, the index can only shrink
            while (rl > 0 && pmin[w][k0 - rl] <= target) --rl;
          }
          if (rl < k0) {
            k = k0 - 1 - rl;
            kin = k + 1;
          }
        }
        if (k >= 0 && kin >= 0 && kin < nv) {
          const float2 q0 = poly[w][kin], q1 = poly[w][k];
          const float ea = q1.x - q0.x, eb = q1.y - q0.y;
          const float den = sc.x * eb - sc.y * ea;
          const float t = (q0.x * eb - q0.y * ea) * fast_rcp(den);
          if (t >= 0.f) e = fminf(t, a.r_max);
        }
      }
      if (EXPECT_ONLY) {
        if (i >= a.exp_first && i < a.exp_first + a.exp_count)
          a.exp_out[(size_t)(i - a.exp_first) * a.n_beams + b] = e;
      } else {
        const float rm = a.ranges[b];
        if (rm > 0.f) {
          const float d = (rm - e) * a.inv_sigma;
          acc += d * d;
          ++nvalid;
        }
      }
    }
    if (!EXPECT_ONLY) {
      const double accd = wave_sum((double)acc);
      const int nvs = wave_sum(nvalid);
      if (lane == 0) a.lw[i] = -0.5 * accd - (double)nvs * a.lognorm;
    }
  }
}
