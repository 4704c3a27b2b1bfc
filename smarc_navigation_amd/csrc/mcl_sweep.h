// mcl_sweep.h -- MBES update on a regularly triangulated height mesh WITHOUT a traversal per ray: the fan sweep.
//
// The beams of one ping lie in one plane through the sensor (D_b = sin a_b c1 - cos a_b c2, mcl_mbes.h), so all
// 512 expected ranges of a particle are intersections of rays from ONE point with ONE curve: the slice of the
// seabed by the fan plane.  On a triangulated surface that slice is a polyline whose vertices are the points
// where the plane crosses triangle edges.  In the plane's own coordinates (s along c1, t along -c2: beam b is
// the half line s = t tan a_b) the first hit of beam b is the first polyline segment, walking outward from the
// nadir, whose far end has s/t >= tan a_b -- PROVIDED the slice is a graph over s, which holds when the plane
// is closer to vertical than the steepest triangle is to horizontal (tan(tilt) * max slope < 1, checked per
// particle against the map's slope bound).  Crossings of one beam are then ordered by s, and s grows along the
// walk, so the first one met is the nearest.
//
// One LANE per (particle, side of the nadir): it finds the nadir hit with the ordinary clearance traversal
// (cast_clear on the height array in global memory: a near-vertical ray, one to three cells), then walks the
// slice triangle by triangle -- on a lattice triangulation the triangle across edge (A, B) from (A, B, C) has
// the third node A + B - C, so a step is one height load, one plane evaluation and a handful of selects --
// and merges the ascending beam table against the polyline as it goes: every beam is resolved by one
// 2-D segment intersection (~19 VALU) instead of a cell-by-cell march (~200 VALU in k_mbes_fast).  No LDS
// tile, no groups: heights come through L1/L2 (the walks of a converged cloud share their lines).
//
// Anything the sweep cannot prove simple -- fan too tilted for the slope bound, footprint not inside the map,
// no nadir hit inside r_max, a degenerate start triangle -- is handed over, per PARTICLE: the hand-over list is the
// visiting order of an ordinary k_mbes_classify / k_mbes_fast / k_mbes_cast pass that reads its length on the device.
#pragma once
#include "mcl_mbes.h"

#ifndef SWEEP_THREADS
#define SWEEP_THREADS 256
#endif
#ifndef SWEEP_MIN_WAVES
#define SWEEP_MIN_WAVES 8   // waves per SIMD the register budget is held to (<= 64 VGPRs)
#endif

#ifdef SWEEP_DEBUG
#define SWEEP_FAIL(code)                                      \
  do {                                                        \
    if (EXPECT_ONLY && side == 1) exp_row[19] = (float)(code); \
    return false;                                             \
  } while (0)
#else
#define SWEEP_FAIL(code) return false
#endif

struct SweepNode {
  int P;        // lattice coordinates relative to the sensor's cell, packed i * 65536 + j (j signed)
  float d;      // signed distance to the fan plane (scaled)
  float s, t;   // in-plane coordinates, s mirrored so that it grows outward on this lane's side
};

// SURF 2: every cell split along 00-11; SURF 3: along 10-01.
// Returns false when the particle has to go to the general kernel.  acc: sum over this side's beams of
// ((range - expected) * weight)^2; EXPECT_ONLY: expected ranges to exp_row[b] instead.
template <int SURF, bool EXPECT_ONLY>
__device__ __forceinline__ bool sweep_side(const MbesArgs& a, const MbesPose& P, const float4* __restrict__ sbeam,
                                           const float* __restrict__ stail, int side, float* __restrict__ exp_row,
                                           float& acc_out) {
  acc_out = 0.f;
  const int nx = a.nx, ny = a.ny, B = a.n_beams;
  // every test before the first map access feeds ONE verdict (`pre`), tested once
  bool pre = P.um >= 1.0 && P.um < (double)(nx - 2) && P.vm >= 1.0 && P.vm < (double)(ny - 2);  // (NaN: false)
  const float c2z = P.c2[2];
  pre = pre & (c2z >= a.sweep_c2z_min);  // else the fan plane is too far from vertical for the terrain's slopes
  const double fum = pre ? floor(P.um) : 1.0, fvm = pre ? floor(P.vm) : 1.0;
  const int I0 = (int)fum, J0 = (int)fvm;
  const float ul = (float)(P.um - fum), vl = (float)(P.vm - fvm);
  const float res = a.res, inv_res = (float)a.inv_res, oz = P.oz;
  const float sg = side ? -1.f : 1.f;
  // beams of this side, outward from the nadir
  int ptr = side ? a.b_split - 1 : a.b_split;
  const int pstep = side ? -1 : 1, pend = side ? -1 : B;
  const bool none = ptr == pend;
  // ---- how far out can the walk go?  The outermost beam of the side is below every node once it reaches z_min
  float s_stop = none ? 0.f : a.r_max;  // (a side without beams only takes part in the nadir cast)
  if (!none) {
    const float2 sc = a.beam_sc[side ? 0 : B - 1];
    const float dz_e = sc.x * P.c1[2] - sc.y * c2z;
    if (dz_e < -1e-4f) s_stop = fminf(s_stop, (a.zmin_map - oz) * fast_rcp(dz_e) * fabsf(sc.x));
  }
  s_stop += 2.f * res;
  // ---- footprint of everything the lane may touch: X = O + s (+-c1) - t c2 with s in [-2 res, s_stop + 2 res] and
  // t between the values at which such a point can lie on the surface (z in [z_min, z_max]); 3 nodes of margin
  {
    const float rc = fast_rcp(c2z);
    const float sl = fabsf(P.c1[2]) * (s_stop + 2.f * res);
    const float t_hi = ((oz - a.zmin_map) + sl) * rc + res, t_lo = fminf(((oz - a.zmax_map) - sl) * rc - res, 0.f);
    const float s_lo = -2.f * res, s_hi = s_stop + 2.f * res;
    const float ax = sg * P.c1[0] * inv_res, ay = sg * P.c1[1] * inv_res, bx = -P.c2[0] * inv_res, by = -P.c2[1] * inv_res;
    const float ux0 = fminf(s_lo * ax, s_hi * ax) + fminf(t_lo * bx, t_hi * bx);
    const float ux1 = fmaxf(s_lo * ax, s_hi * ax) + fmaxf(t_lo * bx, t_hi * bx);
    const float vy0 = fminf(s_lo * ay, s_hi * ay) + fminf(t_lo * by, t_hi * by);
    const float vy1 = fmaxf(s_lo * ay, s_hi * ay) + fmaxf(t_lo * by, t_hi * by);
    const float fi0 = (float)I0, fj0 = (float)J0;
    pre = pre & (fi0 + ux0 >= 3.f) & (fi0 + ux1 <= (float)(nx - 5)) & (fj0 + vy0 >= 3.f) & (fj0 + vy1 <= (float)(ny - 5));
    pre = pre & (s_hi * fmaxf(fabsf(ax), fabsf(ay)) + fmaxf(-t_lo, t_hi) * fmaxf(fabsf(bx), fabsf(by)) < 30000.f);  // packed coordinates
  }
  if (!pre) SWEEP_FAIL(1);
  const float* __restrict__ gp = a.grid + ((size_t)I0 * ny + J0);  // node (I0, J0); every access below is inside the footprint
  const float* __restrict__ grid = a.grid;
  const int g0 = I0 * ny + J0, g_hi = nx * ny - 1;  // (maps below 2^31 nodes: checked on the host)
  // ---- nadir hit: the ordinary clearance traversal on the global height array
  const float r0 = cast_clear<SURF>(gp, ny, a, ul, vl, oz, -P.c2[0] * inv_res, -P.c2[1] * inv_res, -c2z, a.zmax_map, a.r_max);
  if (!(r0 < a.r_max)) SWEEP_FAIL(5);
  if (none) return true;
  // ---- plane and in-plane coordinates as affine functions of (i, j, h): lattice coordinates relative to (I0, J0)
  const float nx_ = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny_ = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
              nz_ = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
  const float pu = nx_ * res, pv = ny_ * res, pz = nz_, p0 = -(pu * ul + pv * vl + pz * oz);
  const float su = sg * P.c1[0] * res, sv = sg * P.c1[1] * res, sz = sg * P.c1[2], s0 = -(su * ul + sv * vl + sz * oz);
  const float tu = -P.c2[0] * res, tv = -P.c2[1] * res, tz = -c2z, t0 = -(tu * ul + tv * vl + tz * oz);
  auto node = [&](int Pk) {
    SweepNode N;
    N.P = Pk;
    const int j = __builtin_amdgcn_sbfe(Pk, 0, 16), i = (Pk - j) >> 16;
    const float h = gp[i * ny + j];
    const float fi = (float)i, fj = (float)j;
    N.d = fmaf(pu, fi, fmaf(pv, fj, fmaf(pz, h, p0)));
    N.s = fmaf(su, fi, fmaf(sv, fj, fmaf(sz, h, s0)));
    N.t = fmaf(tu, fi, fmaf(tv, fj, fmaf(tz, h, t0)));
    return N;
  };
  // ---- the triangle under the nadir hit
  SweepNode A, Bn;
  int C;
  float s_prev, t_prev, s_cur, t_cur;
#ifdef SWEEP_DEBUG
  float dbg[6] = {0, 0, 0, 0, 0, 0};
#endif
  {
    const float uh = fmaf(r0, -P.c2[0] * inv_res, ul), vh = fmaf(r0, -P.c2[1] * inv_res, vl);
    const float cfi = floorf(uh), cfj = floorf(vh);
    const float fu = uh - cfi, fv = vh - cfj;
    const int c00 = (int)cfi * 65536 + (int)cfj;
    int k0, k1, k2;
    if (SURF == 2) {
      const bool lower = fv <= fu;  // (00, 10, 11) : (00, 11, 01)
      k0 = c00;
      k1 = lower ? c00 + 65536 : c00 + 65537;
      k2 = lower ? c00 + 65537 : c00 + 1;
    } else {
      const bool lower = fu + fv <= 1.f;  // (00, 10, 01) : (10, 11, 01)
      k0 = lower ? c00 : c00 + 65536;
      k1 = lower ? c00 + 65536 : c00 + 65537;
      k2 = c00 + 1;
    }
    const SweepNode N0 = node(k0), N1 = node(k1), N2 = node(k2);
    const bool p0b = N0.d > 0.f, p1b = N1.d > 0.f, p2b = N2.d > 0.f;
    if (p0b == p1b && p1b == p2b) SWEEP_FAIL(6);  // the plane misses the triangle (rounding at its border)
    // the node alone on its side of the plane, and the two edges the plane crosses
    const int L = (p0b != p1b && p0b != p2b) ? 0 : ((p1b != p0b && p1b != p2b) ? 1 : 2);
    const SweepNode NL = L == 0 ? N0 : (L == 1 ? N1 : N2);
    const SweepNode NM = L == 0 ? N1 : N0;
    const SweepNode NN = L == 2 ? N1 : N2;
    const float lm = NL.d * fast_rcp(NL.d - NM.d), ln = NL.d * fast_rcp(NL.d - NN.d);
    const float sm = fmaf(lm, NM.s - NL.s, NL.s), tm = fmaf(lm, NM.t - NL.t, NL.t);
    const float sn = fmaf(ln, NN.s - NL.s, NL.s), tn = fmaf(ln, NN.t - NL.t, NL.t);
    if (!(sm != sn)) SWEEP_FAIL(7);  // the plane only touches the triangle at a node (or NaN)
    const bool far_m = sm > sn;     // this side walks out through the edge whose crossing lies further out
    const SweepNode NF = far_m ? NM : NN;
    const bool pl = NL.d > 0.f;
    A = pl ? NF : NL;   // A: d <= 0, Bn: d > 0
    Bn = pl ? NL : NF;
#ifdef SWEEP_DEBUG
    dbg[0] = sm; dbg[1] = tm; dbg[2] = sn; dbg[3] = tn; dbg[4] = (float)L; dbg[5] = NL.d;
#endif
    C = far_m ? NN.P : NM.P;
    s_cur = far_m ? sm : sn;
    t_cur = far_m ? tm : tn;
    s_prev = far_m ? sn : sm;
    t_prev = far_m ? tn : tm;
    if (!(t_cur > 0.f)) SWEEP_FAIL(8);
  }
  // ---- walk outward, merging the beam table against the polyline
  float acc = 0.f;
  bool ok = true;
  const int max_steps = (int)(3.f * (s_stop + 4.f * res) * inv_res) + 16;
  int step = 0;
  float4 bm = sbeam[ptr];  // the next beam to resolve stays in registers across segments: a vertex it passes beyond costs no LDS read
  for (;;) {
    // the third node of the triangle across (A, Bn): its height load is in flight while the beams are resolved
    const int Nk = (int)((unsigned)A.P + (unsigned)Bn.P - (unsigned)C);
    const int nj = __builtin_amdgcn_sbfe(Nk, 0, 16), ni = (Nk - nj) >> 16;
    // (the footprint test keeps a sane walk inside the map; the clamp keeps a NaN-driven one from reading outside it)
    const float hN = grid[(unsigned)min(max(g0 + ni * ny + nj, 0), g_hi)];
    const float dts = t_cur - t_prev;
    for (;;) {
      // (no `ptr != pend` test: the record beyond the last beam has tan a = +inf and t_cur > 0, so e_cur = -inf)
      const float e_cur = fmaf(-bm.x, t_cur, s_cur);
      if (!(e_cur >= 0.f)) break;  // the beam passes beyond this vertex
      const float e_prev = fmaf(-bm.x, t_prev, s_prev);
      // crossing of the half line s = t tan a with the segment: e changes sign (<= 0 at prev, >= 0 at cur)
      const float lam = fmaxf(fminf(e_prev * fast_rcp(e_prev - e_cur), 1.f), 0.f);
      const float tau = fmaf(lam, dts, t_prev);
      const float r = fminf(tau * bm.y, a.r_max);  // range = t / cos a; beyond r_max (or NaN): r_max
      if (EXPECT_ONLY) {
        exp_row[ptr] = r;
      } else {
        const float dd = (bm.z - r) * bm.w;
        acc = fmaf(dd, dd, acc);
      }
      ptr += pstep;
      bm = sbeam[ptr];  // (one sentinel record on either end of the table)
    }
    if (ptr == pend) break;
    if (s_cur > s_stop) break;  // every beam left misses inside r_max (tail below)
    if (++step > max_steps) {
      ok = false;
      break;
    }
    const float fi = (float)ni, fj = (float)nj;
    const float dN = fmaf(pu, fi, fmaf(pv, fj, fmaf(pz, hN, p0)));
    const float sN = fmaf(su, fi, fmaf(sv, fj, fmaf(sz, hN, s0)));
    const float tN = fmaf(tu, fi, fmaf(tv, fj, fmaf(tz, hN, t0)));
    const bool pos = dN > 0.f;
    C = pos ? Bn.P : A.P;
    A.P = pos ? A.P : Nk;
    A.d = pos ? A.d : dN;
    A.s = pos ? A.s : sN;
    A.t = pos ? A.t : tN;
    Bn.P = pos ? Nk : Bn.P;
    Bn.d = pos ? dN : Bn.d;
    Bn.s = pos ? sN : Bn.s;
    Bn.t = pos ? tN : Bn.t;
    const float lam = A.d * fast_rcp(A.d - Bn.d);
    s_prev = s_cur;
    t_prev = t_cur;
    s_cur = fmaf(lam, Bn.s - A.s, A.s);
    t_cur = fmaf(lam, Bn.t - A.t, A.t);
    if (!(t_cur > 0.f)) {  // the seabed rises above the sensor's own horizon: not for the sweep (see the sentinel records)
      ok = false;
      break;
    }
  }
  if (ok && ptr != pend) {
    if (EXPECT_ONLY) {
      for (; ptr != pend; ptr += pstep) exp_row[ptr] = a.r_max;
    } else {
      acc += stail[ptr];
    }
  }
#ifdef SWEEP_DEBUG
  if (EXPECT_ONLY && side == 1) {
    exp_row[0] = r0; exp_row[1] = s_prev; exp_row[2] = t_prev; exp_row[3] = s_cur; exp_row[4] = t_cur;
    exp_row[5] = A.d; exp_row[6] = Bn.d; exp_row[7] = (float)step; exp_row[8] = (float)ptr; exp_row[9] = s_stop;
    exp_row[10] = (float)max_steps; exp_row[11] = ok ? 1.f : 0.f; exp_row[12] = dbg[0]; exp_row[13] = dbg[1];
    exp_row[14] = dbg[2]; exp_row[15] = dbg[3]; exp_row[16] = dbg[4]; exp_row[17] = dbg[5];
  }
#endif
  acc_out = acc;
  return ok;
}

template <int SURF, bool EXPECT_ONLY>
__global__ void __launch_bounds__(SWEEP_THREADS, SWEEP_MIN_WAVES) k_mbes_sweep(MbesArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sweep_lds[];
  float4* sbeam = (float4*)sweep_lds + 1;  // records -1 and n_beams exist (read, never used)
  float* stail = (float*)(sbeam + a.n_beams + 1);
  for (int b = threadIdx.x; b < a.n_beams; b += SWEEP_THREADS) {
    sbeam[b] = a.sweep_beams[b];
    stail[b] = a.sweep_tail[b];
  }
  if (threadIdx.x == 0) sbeam[-1] = sbeam[a.n_beams] = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);  // "never reached"
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long gid = blockIdx.x * (long long)SWEEP_THREADS + threadIdx.x;
  // position in the visiting order; lanes 2k, 2k+1 = the two sides of one particle.  (Expected ranges in the natural
  // order: the grid only covers the particles asked for.)
  const long long j = (gid >> 1) + ((EXPECT_ONLY && !a.perm) ? a.exp_first : 0);
  const int side = (int)(gid & 1);
  const bool valid = j < a.n;
  const long long i = (valid && a.perm) ? (long long)a.perm[j] : j;
  bool ok = true, work = valid;
  float* exp_row = nullptr;
  if (EXPECT_ONLY) {
    work = valid && i >= a.exp_first && i < a.exp_first + a.exp_count;
    if (work) exp_row = a.exp_out + (size_t)(i - a.exp_first) * a.n_beams;
  }
  float acc = 0.f;
  if (work) {
    const MbesPose P = a.pose[i];
    ok = sweep_side<SURF, EXPECT_ONLY>(a, P, sbeam, stail, side, exp_row, acc);
  }
#ifdef SWEEP_PRINTF
  if (EXPECT_ONLY && work) printf("sweep i=%lld side=%d ok=%d acc=%g\n", i, side, (int)ok, (double)acc);
#endif
  // both sides of a particle agree on its fate (the exchange is NOT under `ok &&`: every lane takes part in it)
  const int ok_i = ok ? 1 : 0;
  const int ok_other = __shfl_xor(ok_i, 1, 64);
  const bool ok2 = (ok_i != 0) & (ok_other != 0);
#ifdef SWEEP_DEBUG
  if (EXPECT_ONLY && work && side == 0) {
    exp_row[20] = ok ? 1.f : 0.f;
    exp_row[21] = (float)ok_other;
  }
#endif
  const double acc2 = (double)acc + (double)__shfl_xor(acc, 1, 64);
#ifdef SWEEP_DEBUG2
  if (EXPECT_ONLY && work) exp_row[side ? a.n_beams - 1 : a.n_beams - 2] = 500.f + 10.f * ok_i + ok_other;
#endif
  const bool writer = work && side == 0;
  double v = -__builtin_inf();
  if (writer && ok2 && !EXPECT_ONLY) {
    v = -0.5 * acc2 - (double)a.sweep_nvalid * a.lognorm;
    a.lw[i] = v;
  }
  // hand-overs: one atomic per wave
  const unsigned long long dm = __ballot(writer && !ok2);
  if (dm) {
    int base = 0;
    if (lane == 0) base = atomicAdd(a.defer_count, (int)__popcll(dm));
    base = __builtin_amdgcn_readfirstlane(base);
    if (writer && !ok2) a.defer_idx[base + (int)__popcll(dm & ((1ull << lane) - 1ull))] = (u32)i;
  }
  if (!EXPECT_ONLY && a.max_slots) {
    // the normalisation needs max lw: one atomic per wave on an order-preserving key
    if (!(v == v)) v = -__builtin_inf();  // NaN never wins
    const double m = wave_max(v);
    if (lane == 0 && m > -__builtin_inf())
      atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * (SWEEP_THREADS / 64) + (threadIdx.x >> 6)) & (MCL_MAX_SLOTS - 1)],
                ordered_key(m));
  }
}
