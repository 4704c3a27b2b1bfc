// mcl_slice.h -- MBES update over an ARBITRARY triangle mesh without a march per ray: the fan slice.
//
// The beams of one ping lie in one plane through the sensor (mcl_sweep.h).  On a height field the fan sweep walks the
// slice of the seabed by that plane from the nadir outward; a triangle soup -- vertical faces, overhangs, wrecks
// floating over the seabed, non-manifold edges: everything mesh_build cannot prove a single-valued height field -- has
// no such order.  But the slice is still only a few hundred SEGMENTS (plane x triangle), and a beam's expected range is
// the nearest crossing of its half line with any of them:
//
//   one WAVEFRONT per particle;
//   1. lanes = columns of the mesh's cell grid along the fan plane's trace: every lane enumerates the two to four cells
//      of its column the plane can cut inside the map's depth range (their cell words loaded as one batch), tests each
//      cell's box (its own z-range) against the plane, and appends the triangle records of the cells that are cut to
//      the wave's list in LDS;
//   2. lanes = triangles of that list (balanced to one iteration; columns that miss the map or hold one cell no longer
//      idle): first the cheap question -- does the plane separate its vertices? --, the list compacted in place to the
//      ones it does, then the expensive part with every lane busy.  Per triangle: the three
//      vertices (map-frame coordinates in the record: a vertex has the same bits in every record it appears in), their
//      distance to the plane and in-plane coordinates (s along c1, t along -c2: beam b is the half
//      line s = t tan a_b, t > 0); a triangle the plane separates gives one segment -- its end points interpolated
//      from the vertex BELOW the plane to the one ABOVE it, so that two triangles sharing an edge compute the same
//      point bit for bit: the slice is watertight;
//   3. per segment: the run of beams whose tangent lies between those of its end points (lower bound from a bucket
//      table over the ascending tangent table in LDS), and for each of them the crossing -> range = t / cos a -> an LDS
//      atomicMin on the beam's slot (positive floats order like their bit patterns);
//   4. the wave's lanes then take the beams back (b = lane, lane + 64, ...): residual, sum, log-likelihood -- the same
//      epilogue as the traversal kernels.
//
// Work per particle: ~150-600 triangle tests + ~1 000 beam x segment pairs, against 512 rays x ~35 cells x (z-range
// test + triangle records) for the traversal (k_mbes_fast<4>): measured at 1 048 576 x 512 on one MI355X 4.8 -> 4.0 ms
// per update on the regular 1 M-triangle mesh cast as a soup, 21.3 -> 6.7 ms on the irregular TIN cast as a soup
// (bench.py extra.mesh_general / mesh_soup_irregular).  The kernel is VALU-issue-bound: 2 600 / 4 000 wave
// instructions per particle at lane utilisation 0.49 / 0.58 (rocprofv3 SQ counters, tools/pmc_kernel.sh); what is
// left is the divergence of the beam loop (a segment under the sensor covers a dozen beams, one at the swath's edge
// one or two) and phase 1 itself -- a chain of dependent global loads with next to no arithmetic, 40 - 60 % of the
// kernel's time (ablation: the launch without phases 2 - 4 takes 2.5 of the 4.1 / 6.7 ms).
//
// Exact by construction (two-sided triangles, no side walls: the oracle's definition), deterministic (a minimum does
// not depend on the order of its operands), a function of the particle alone (the determinism rule of mcl_mbes.h).
// Declined -- and handed to k_mbes_cast<1, ., 2> through the sweep's hand-over list --: fans whose plane is closer to
// horizontal than 60 degrees or whose across-track axis points more than 60 degrees out of the horizontal (the column
// enumeration assumes a near-vertical plane with a near-horizontal trace), NaN poses.  Beam tables: ascending, within
// 85 degrees of the nadir (the host checks: sweep_angles_ok), like the sweep.
#pragma once
#include "mcl_mbes.h"

#ifndef SLICE_WAVES
#define SLICE_WAVES 4   // particles per workgroup
#endif
#ifndef SLICE_G
#define SLICE_G 60      // particles per group of k_mbes_slice_group (at the end of this file)
#endif
#define SLICE_THREADS (SLICE_WAVES * 64)
#ifndef SLICE_LIST
#define SLICE_LIST 511    // triangle records a wave's list holds per chunk of columns (2 KiB with its counter); more: the general kernel
#endif
#define SLICE_ROWS 4      // cell words of one column loaded per batch
#ifndef SLICE_GRID
#define SLICE_GRID 32768   // workgroups of a launch: a wave takes every (4 x SLICE_GRID)-th particle.  (measured at 1 M x 512, regular mesh
                           //  as a soup / irregular TIN, ms: one particle per wave 4.12 / 6.71; 2 048 workgroups 4.57 / 7.90; 8 192: 3.80 / 6.44;
                           //  32 768: 3.68 / 6.21; 65 536: 3.72 / 6.26 -- the tables a workgroup builds are shared by 8 particles per wave.
                           //  Requesting the NEXT chunk's cell words before this chunk's triangles are worked on was built and
                           //  measured too: 95 VGPRs, 5 waves per SIMD instead of 7 -- 4.26 / 7.12, reverted.)
#endif
#define SLICE_LUT 512     // tangent buckets of the beam look-up table (1 KiB per workgroup)
#ifndef SLICE_COLS
#define SLICE_COLS 32     // columns per chunk (<= 64: one per lane).  (measured, 1 M x 512, regular mesh as a soup / irregular TIN, ms:
                          //  4 waves, list 1023, 64 columns 4.81 / 7.10; 8 waves, 767, 64: 4.44 / 6.54; 8, 511, 32: 4.11 / 6.90;
                          //  4, 511, 32: 3.95 / 6.69 -- 20 KiB of LDS per workgroup, 8 waves per SIMD)
#endif

// v_min_u32 on an LDS word: ranges are positive floats, their order is the order of their bit patterns
__device__ __forceinline__ void lds_min_range(unsigned* slot, float range) {
  __hip_atomic_fetch_min(slot, __float_as_uint(range), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One beam of a segment's run, WITHOUT branches: the crossing of the half line s = t T with the segment A -> B (e = s - T t
// changes sign) as a range into the beam's LDS minimum; a beam the segment does not cross, a crossing behind the sensor or
// beyond r_max stores +inf, which a minimum ignores (the slots start at r_max).  (Until late in round 5 the three tests
// were `continue`s: the compiler built a nest of exec-mask regions around them, ~25 scalar and ~25 vector instructions
// and five branches per beam; the arithmetic is unchanged.)
__device__ __forceinline__ void slice_beam(unsigned* slot, float T, float sec_b, float sA, float tA, float sB, float tB,
                                           float dts, float r_max) {
  const float eA = fmaf(-T, tA, sA), eB = fmaf(-T, tB, sB);
  const bool miss = (eA > 0.f) == (eB > 0.f) && eA != 0.f && eB != 0.f;
  const float den = eA - eB;
  float lam = den != 0.f ? eA * __builtin_amdgcn_rcpf(den) : 0.f;
  lam = fminf(fmaxf(lam, 0.f), 1.f);
  const float tau = fmaf(lam, dts, tA);
  const float range = tau * sec_b;
  const bool ok = !miss & (tau > 0.f) & (range < r_max);   // (NaN anywhere: not ok)
  __hip_atomic_fetch_min(slot, ok ? __float_as_uint(range) : 0x7f800000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool EXPECT_ONLY>
__global__ void __launch_bounds__(SLICE_THREADS, 4) k_mbes_slice(MbesArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char slice_lds[];
  const int B = a.n_beams;
  float2* tsb = (float2*)slice_lds;                  // B: (ascending tangent, 1 / cos a) -- one ds_read_b64 per beam
  unsigned* rng_all = (unsigned*)(tsb + B);          // SLICE_WAVES x B: nearest crossing per beam (float bits)
  unsigned* tl_all = rng_all + (size_t)SLICE_WAVES * B;   // SLICE_WAVES x (SLICE_LIST + 1): the wave's triangle list, its length
  unsigned short* lut = (unsigned short*)(tl_all + (size_t)SLICE_WAVES * (SLICE_LIST + 1));   // SLICE_LUT: first beam at or beyond a tangent bucket's lower end
  if (!EXPECT_ONLY && a.in_list) {
    // the TIN sweep's hand-over kernel: mostly there is nothing on the list -- a workgroup without work leaves before it
    // builds the tables (the host sizes the grid by the count of two updates ago; a cloud that runs into a ragged
    // outline all at once still finds workgroups enough)
    const long long n0 = a.slice_loose ? (long long)*a.slice_loose_count * SLICE_G : (long long)*a.in_count;
    if (a.host_count && blockIdx.x == 0 && threadIdx.x == 0) *a.host_count = *a.in_count;   // (the sweep's hand-over count: sizes these launches two updates on)
    if ((long long)blockIdx.x * SLICE_WAVES >= n0) return;
  }
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float2 sc = a.beam_sc[b];
    const float sec = __builtin_amdgcn_rcpf(sc.y);
    tsb[b] = make_float2(sc.x * sec, sec);
  }
  __syncthreads();
  // "first beam with tan >= T" without a bisection per segment: SLICE_LUT uniform buckets over the table's tangent
  // range, each holding the first beam at or beyond its lower end (built here by bisection, once per workgroup); a
  // query reads its bucket and scans forward -- one or two beams for a table that is uniform in angle
  const float lut_lo = tsb[0].x, lut_w = fmaxf((tsb[B - 1].x - tsb[0].x) * (1.f / SLICE_LUT), 1e-12f), lut_iw = 1.f / lut_w;
  for (int k = threadIdx.x; k < SLICE_LUT; k += blockDim.x) {
    const float T = lut_lo + (float)k * lut_w;
    int lo = 0, hi = B;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (tsb[mid].x < T) lo = mid + 1; else hi = mid;
    }
    lut[k] = (unsigned short)lo;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* rng = rng_all + (size_t)w * B;
  unsigned* tlist = tl_all + (size_t)w * (SLICE_LIST + 1);
  unsigned* tcount = tlist + SLICE_LIST;
  const MeshArgs& ma = a.mesh;
  const float cs = ma.cs, ics = 1.f / cs;
  const unsigned rmax_bits = __float_as_uint(a.r_max);
  const float tan_lo = tsb[0].x, tan_hi = tsb[B - 1].x;
  const float2 sc_lo = a.beam_sc[0], sc_hi = a.beam_sc[B - 1];
  double wmax = -__builtin_inf();
  // (behind k_mbes_slice_group: only the members of the groups it left, by its list -- the list's length is read here)
  const bool listed = !EXPECT_ONLY && a.slice_loose != nullptr;
  // (as the TIN sweep's hand-over kernel: positions run over the sweep's list, in_list[p] is the pose record)
  const long long n_in = a.in_list ? (long long)*a.in_count : a.n;
  const long long n_it = listed ? (long long)*a.slice_loose_count * SLICE_G : n_in;
  for (long long it = (long long)blockIdx.x * SLICE_WAVES + w; it < n_it; it += (long long)gridDim.x * SLICE_WAVES) {
    const long long p_in = listed ? (long long)a.slice_loose[it / SLICE_G] * SLICE_G + it % SLICE_G : it;
    if (p_in >= n_in) continue;
    const long long i = a.in_list ? (long long)a.in_list[p_in] : p_in;
    if (EXPECT_ONLY && (i < a.exp_first || i >= a.exp_first + a.exp_count)) continue;
    MbesPose P;
    u32 slot;   // the particle's state slot: the records may lie in visiting order (mcl_kernels.h: VisitArgs)
    {
      const MbesPose Pv = a.pose[i];   // wave-uniform: scalar registers
      slot = EXPECT_ONLY ? (u32)i : (u32)__builtin_amdgcn_readfirstlane((int)Pv.slot);
      P.um = uniform_f64(Pv.um);
      P.vm = uniform_f64(Pv.vm);
      P.oz = uniform_f32(Pv.oz);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        P.c1[r] = uniform_f32(Pv.c1[r]);
        P.c2[r] = uniform_f32(Pv.c2[r]);
      }
    }
    const float c2z = P.c2[2];
    const float h1 = P.c1[0] * P.c1[0] + P.c1[1] * P.c1[1];   // squared horizontal part of the across-track axis
    const bool sane = fabs(P.um) < 1e9 && fabs(P.vm) < 1e9;
    if (!(sane && c2z >= 0.5f && h1 >= 0.25f)) {   // (NaN: declined) -> the general kernel
      if (lane == 0) a.defer_idx[atomicAdd(a.defer_count, 1)] = (u32)i;
      continue;
    }
    for (int b = lane; b < B; b += 64) rng[b] = rmax_bits;
    // the sensor in the map frame, split into an fp32 part and its sub-ulp rest (vertices are fp32, the sensor is not)
    const double Ox = ma.x0 + P.um * (double)cs, Oy = ma.y0 + P.vm * (double)cs;
    const float Oxf = (float)Ox, Oyf = (float)Oy, dOx = (float)(Ox - (double)Oxf), dOy = (float)(Oy - (double)Oyf);
    const float oz = P.oz;
    const float nx = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
                nz = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
    // ---- how far out the fan can meet the map: the outermost beam of either side down to z_min, or r_max
    float s_pos = 0.f, s_neg = 0.f;
    {
      const float2 hi = sc_hi, lo = sc_lo;
      const float rate_hi = hi.y * c2z - hi.x * P.c1[2], rate_lo = lo.y * c2z - lo.x * P.c1[2];   // descent per metre of range
      const float rho_hi = rate_hi > 1e-4f ? fminf(a.r_max, fmaxf(oz - a.zmin_map, 0.f) * __builtin_amdgcn_rcpf(rate_hi)) : a.r_max;
      const float rho_lo = rate_lo > 1e-4f ? fminf(a.r_max, fmaxf(oz - a.zmin_map, 0.f) * __builtin_amdgcn_rcpf(rate_lo)) : a.r_max;
      s_pos = fmaxf(hi.x, 0.f) * rho_hi + cs;
      s_neg = fmaxf(-lo.x, 0.f) * rho_lo + cs;
    }
    // depth range along -c2 inside which a point of the plane can lie on the map (z in [z_min, z_max])
    const float s_abs = fmaxf(s_pos, s_neg);
    const float rc = __builtin_amdgcn_rcpf(c2z);
    const float t_b = fminf(((oz - a.zmin_map) + fabsf(P.c1[2]) * s_abs) * rc, a.r_max) + cs;
    const float t_a = fmaxf(((oz - a.zmax_map) - fabsf(P.c1[2]) * s_abs) * rc - cs, 0.f);
    // ---- columns of cells along the trace.  Major axis M: the larger horizontal component of c1; a point of the
    // plane is O + s c1 - t c2, its major coordinate m = s c1m - t c2m, its minor one q = s c1q - t c2q
    const bool major_x = fabsf(P.c1[0]) >= fabsf(P.c1[1]);
    const float c1m = major_x ? P.c1[0] : P.c1[1], c1q = major_x ? P.c1[1] : P.c1[0];
    const float c2m = major_x ? P.c2[0] : P.c2[1], c2q = major_x ? P.c2[1] : P.c2[0];
    const double pm_d = major_x ? P.um : P.vm, pq_d = major_x ? P.vm : P.um;   // sensor position, cell units
    const int gm = major_x ? ma.gx : ma.gy, gq = major_x ? ma.gy : ma.gx;
    const int Im = (int)floor(pm_d), Iq = (int)floor(pq_d);
    const float fm = (float)(pm_d - (double)Im), fq = (float)(pq_d - (double)Iq);   // fraction inside the sensor's cell
    // range of the major coordinate (metres from the sensor) over s in [-s_neg, s_pos], t in [t_a, t_b]
    const float m0 = fminf(-s_neg * c1m, s_pos * c1m) + fminf(-t_a * c2m, -t_b * c2m);
    const float m1 = fmaxf(-s_neg * c1m, s_pos * c1m) + fmaxf(-t_a * c2m, -t_b * c2m);
    int k_lo = Im + (int)floorf(fm + m0 * ics), k_hi = Im + (int)floorf(fm + m1 * ics);
    k_lo = max(k_lo, 0);
    k_hi = min(k_hi, gm - 1);
    const float rmq = c1q * __builtin_amdgcn_rcpf(c1m);   // |.| <= 1
    const float kq = c2m * rmq - c2q;                      // minor coordinate: q = m rmq + t kq
    // The columns are taken in chunks of at most 64 (one per lane), sized evenly; per chunk three phases share the
    // wave's list in LDS.
    const int ncols = k_hi - k_lo + 1;
    const int nchunks = (ncols + SLICE_COLS - 1) / SLICE_COLS;
    int per_chunk = nchunks > 0 ? (ncols + nchunks - 1) / nchunks : 1;
    bool overflow = false;
    for (int cbase = k_lo; cbase <= k_hi;) {
      // ---- phase A: lanes = columns.  The triangle records of the cells the plane cuts go onto the list, one word per
      // record (the order of the list is the order of the LDS atomics -- a minimum does not care).  The cell words of a
      // column's rows are loaded as one batch: ONE memory latency per chunk.
      if (lane == 0) *tcount = 0u;
      const int col = cbase + lane;
      const bool col_ok = lane < per_chunk && col <= k_hi;
      // this column's slab of the major coordinate (metres from the sensor), and the minor range of the plane over it
      const float ma0 = ((float)(col - Im) - fm) * cs, ma1 = ma0 + cs;
      const float q0 = fminf(ma0 * rmq, ma1 * rmq) + fminf(t_a * kq, t_b * kq);
      const float q1 = fmaxf(ma0 * rmq, ma1 * rmq) + fmaxf(t_a * kq, t_b * kq);
      int r_lo = Iq + (int)floorf(fq + q0 * ics), r_hi = Iq + (int)floorf(fq + q1 * ics);
      r_lo = max(r_lo, 0);
      r_hi = col_ok ? min(r_hi, gq - 1) : r_lo - 1;
      for (int rb = r_lo; __builtin_amdgcn_ballot_w64(rb <= r_hi) != 0ull; rb += SLICE_ROWS) {
        uint2 info[SLICE_ROWS];
#pragma unroll
        for (int r = 0; r < SLICE_ROWS; ++r) {
          const int row = rb + r;
          const size_t c = major_x ? (size_t)col * ma.gy + row : (size_t)row * ma.gy + col;
          info[r] = row <= r_hi ? ma.cell_info[c] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < SLICE_ROWS; ++r) {
          const int row = rb + r;
          u32 cnt = info[r].y >> 27;
          if (cnt == 0u) continue;
          const int ci = major_x ? col : row, cj = major_x ? row : col;
          const u32 rs = info[r].y & 0x7ffffffu;
          // the cell's box against the plane (its own z-range, conservative half floats)
          float zlo, zhi;
          cell_zrange(info[r].x, zlo, zhi);
          const float I0x = (float)(ci - (major_x ? Im : Iq)) - (major_x ? fm : fq), I0y = (float)(cj - (major_x ? Iq : Im)) - (major_x ? fq : fm);
          const float cxm = (I0x + 0.5f) * cs, cym = (I0y + 0.5f) * cs, czm = 0.5f * (zlo + zhi) - oz;
          const float dist = nx * cxm + ny * cym + nz * czm;
          const float ext = 0.5f * cs * (fabsf(nx) + fabsf(ny)) + (0.5f * (zhi - zlo) + 1e-3f) * fabsf(nz) + 1e-4f;
          if (!(fabsf(dist) <= ext)) continue;
          if (cnt == 31u) cnt = ma.cell_start[(size_t)ci * ma.gy + cj + 1] - rs;   // "31 or more": the exact count
          const u32 pos = __hip_atomic_fetch_add(tcount, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          for (u32 k = 0; k < cnt && pos + k < SLICE_LIST; ++k) tlist[pos + k] = rs + k;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const u32 ntri = (u32)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned*)tcount);
      if (ntri > SLICE_LIST) {   // more records than the list holds (wave-uniform): the same columns again in halves
        if (per_chunk > 1) {
          per_chunk = (per_chunk + 1) >> 1;
          continue;
        }
        overflow = true;   // (one column alone: a pile-up of records) -> the general kernel casts this particle
        break;
      }
      // ---- phase B1: lanes = triangles of the list.  Which of them does the plane separate?  The list is compacted
      // in place (ballot + prefix count: every lane has read its entry before any lane writes), so that the expensive
      // phase runs with all lanes busy
      u32 nseg = 0;
      for (u32 base = 0; base < ntri; base += 64) {
        const u32 it = base + lane;
        const u32 k = it < ntri ? tlist[it] : 0u;
        bool cut = false;
        if (it < ntri) {
          const float4 v0 = ma.cell_tri[3 * (size_t)k], v1 = ma.cell_tri[3 * (size_t)k + 1], v2 = ma.cell_tri[3 * (size_t)k + 2];
          const float d0 = fmaf(nx, (v0.x - Oxf) - dOx, fmaf(ny, (v0.y - Oyf) - dOy, nz * (v0.z - oz)));
          const float d1 = fmaf(nx, (v1.x - Oxf) - dOx, fmaf(ny, (v1.y - Oyf) - dOy, nz * (v1.z - oz)));
          const float d2 = fmaf(nx, (v2.x - Oxf) - dOx, fmaf(ny, (v2.y - Oyf) - dOy, nz * (v2.z - oz)));
          const bool p0 = d0 > 0.f, p1 = d1 > 0.f, p2 = d2 > 0.f;
          cut = !(p0 == p1 && p1 == p2);   // (NaN vertices: all false, not cut)
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cut);
        if (cut) tlist[nseg + (u32)__popcll(m & ((1ull << lane) - 1ull))] = k;
        nseg += (u32)__popcll(m);
      }
      // ---- phase B2: lanes = triangles the plane separates
      for (u32 it = lane; it < nseg; it += 64) {
      const size_t k = tlist[it];
      const float4 c0 = ma.cell_tri[3 * k], c1v = ma.cell_tri[3 * k + 1], c2v = ma.cell_tri[3 * k + 2];
      // vertices relative to the sensor (v - O with O = Of + dO: exact up to one ulp of a <= 100 m difference)
      const float x0 = (c0.x - Oxf) - dOx, y0 = (c0.y - Oyf) - dOy, z0 = c0.z - oz;
      const float x1 = (c1v.x - Oxf) - dOx, y1 = (c1v.y - Oyf) - dOy, z1 = c1v.z - oz;
      const float x2 = (c2v.x - Oxf) - dOx, y2 = (c2v.y - Oyf) - dOy, z2 = c2v.z - oz;
      const float d0 = fmaf(nx, x0, fmaf(ny, y0, nz * z0)), d1 = fmaf(nx, x1, fmaf(ny, y1, nz * z1)),
                  d2 = fmaf(nx, x2, fmaf(ny, y2, nz * z2));
      const bool p0 = d0 > 0.f, p1 = d1 > 0.f, p2 = d2 > 0.f;
      // the vertex alone on its side, and the two others: crossings on the edges (L, M) and (L, N)
      const int L = (p0 != p1 && p0 != p2) ? 0 : ((p1 != p0 && p1 != p2) ? 1 : 2);
      const float xl = L == 0 ? x0 : (L == 1 ? x1 : x2), yl = L == 0 ? y0 : (L == 1 ? y1 : y2), zl = L == 0 ? z0 : (L == 1 ? z1 : z2);
      const float xm = L == 0 ? x1 : x0, ym = L == 0 ? y1 : y0, zm = L == 0 ? z1 : z0;
      const float xn = L == 2 ? x1 : x2, yn = L == 2 ? y1 : y2, zn = L == 2 ? z1 : z2;
      const float dl = L == 0 ? d0 : (L == 1 ? d1 : d2), dm = L == 0 ? d1 : d0, dn = L == 2 ? d1 : d2;
      // in-plane coordinates of the three
      const float sl = fmaf(P.c1[0], xl, fmaf(P.c1[1], yl, P.c1[2] * zl)), tl = -fmaf(P.c2[0], xl, fmaf(P.c2[1], yl, c2z * zl));
      const float sm = fmaf(P.c1[0], xm, fmaf(P.c1[1], ym, P.c1[2] * zm)), tm = -fmaf(P.c2[0], xm, fmaf(P.c2[1], ym, c2z * zm));
      const float sn = fmaf(P.c1[0], xn, fmaf(P.c1[1], yn, P.c1[2] * zn)), tn = -fmaf(P.c2[0], xn, fmaf(P.c2[1], yn, c2z * zn));
      // crossing of an edge, always from its vertex below the plane (d <= 0) to the one above: the same operands in
      // the same order from either triangle on the edge
      const bool lpos = dl > 0.f;
      float sA, tA, sB, tB;
      {
        const float db = lpos ? dm : dl, da = lpos ? dl : dm;            // below, above
        const float sbv = lpos ? sm : sl, sav = lpos ? sl : sm, tbv = lpos ? tm : tl, tav = lpos ? tl : tm;
        const float lam = db * __builtin_amdgcn_rcpf(db - da);
        sA = fmaf(lam, sav - sbv, sbv);
        tA = fmaf(lam, tav - tbv, tbv);
      }
      {
        const float db = lpos ? dn : dl, da = lpos ? dl : dn;
        const float sbv = lpos ? sn : sl, sav = lpos ? sl : sn, tbv = lpos ? tn : tl, tav = lpos ? tl : tn;
        const float lam = db * __builtin_amdgcn_rcpf(db - da);
        sB = fmaf(lam, sav - sbv, sbv);
        tB = fmaf(lam, tav - tbv, tbv);
      }
      if (!(tA > 0.f) && !(tB > 0.f)) continue;   // behind the sensor's own horizon (or NaN)
      // tangents of the end points as seen from the sensor.  An end point behind the horizon (t <= 0): the visible part
      // of the segment runs from the other end to where it crosses t = 0, at s0 = (sA tB - sB tA) / (tB - tA), and the
      // tangent s / t along it tends to +-infinity with the sign of s0 -- NOT of the hidden end point's own s (a steep
      // sheet from (s, t) = (-1, -10) to (5, 10) crosses the horizon at s0 = +2 and covers the tangents [0.5, +inf))
      const float INF = __builtin_inff();
      const float cross = sA * tB - sB * tA;   // sign of s0 when A is the hidden end, of -s0 when B is
      const float TA = tA > 0.f ? sA * __builtin_amdgcn_rcpf(tA) : (cross > 0.f ? INF : -INF);
      const float TB = tB > 0.f ? sB * __builtin_amdgcn_rcpf(tB) : (cross < 0.f ? INF : -INF);
      const float T_lo = fminf(TA, TB), T_hi = fmaxf(TA, TB);
      if (T_hi < tan_lo || T_lo > tan_hi) continue;
      // first beam with tan >= T_lo (minus a hair: rounding of the quotient).  The bucket's lower end is <= T_first, so
      // its first beam is at or before the one wanted: scan forward (rounding of the bucket index: one bucket back)
      const float T_first = T_lo - 1e-6f * fmaxf(1.f, fabsf(T_lo)), T_last = T_hi + 1e-6f * fmaxf(1.f, fabsf(T_hi));
      int lo = (int)lut[min(max((int)((T_first - lut_lo) * lut_iw) - 1, 0), SLICE_LUT - 1)];
      while (lo < B && tsb[lo].x < T_first) ++lo;
      const float dts = tB - tA;
      // (the beam's tangent and secant are requested one beam ahead: the loop was a chain of two dependent LDS reads per
      //  beam -- ~20 cycles per instruction at 4 - 7 waves per SIMD)
      float2 ts = tsb[min(lo, B - 1)];
      for (int b = lo; b < B; ++b) {
        const float T = ts.x, sec_b = ts.y;
        if (T > T_last) break;
        ts = tsb[min(b + 1, B - 1)];
        slice_beam(&rng[b], T, sec_b, sA, tA, sB, tB, dts, a.r_max);
      }
      }
      cbase += per_chunk;
    }
    if (overflow) {   // (wave-uniform)
      if (lane == 0) a.defer_idx[atomicAdd(a.defer_count, 1)] = (u32)i;
      continue;
    }
    // (the LDS operations of one wave complete in order: no barrier between its lanes' minima and the reads below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float acc = 0.f;
    int nvalid = 0;
    for (int b = lane; b < B; b += 64) {
      const float e = __uint_as_float(rng[b]);
      if (EXPECT_ONLY) {
        a.exp_out[(size_t)(i - a.exp_first) * B + b] = e;
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
        a.lw[slot] = v;
        wmax = v > wmax ? v : wmax;  // NaN never wins
      }
    }
  }
  if (!EXPECT_ONLY && lane == 0 && a.max_slots && wmax > -__builtin_inf())
    atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * SLICE_WAVES + w) & (MCL_MAX_SLOTS - 1)], ordered_key(wmax));
}

// ------------------------------------------------------------------ the fan slice over GROUPS of spatial neighbours
// With the visiting order (mcl_kernels.h: VisitArgs) 32 consecutive pose records are particles from one small bin of
// (x, y, yaw): their fan planes cut almost the same cells and triangles.  k_mbes_slice spends 40 - 60 % of its time in
// phase A -- a chain of dependent global loads per particle -- and fetches every triangle's vertices twice per particle
// from L2; on an irregular mesh it also meets every triangle once per cell it overlaps (~2 records per triangle).  Here a
// workgroup takes a GROUP of SLICE_G records:
//   G0  wave 0, one lane per particle: the fan geometry of each, the group's spread about its first valid member
//       (dO = largest distance between sensors, dn = largest difference of plane normals) and the union of the fans'
//       extents.  A group that is not tight (dO + dn x reach > SLICE_G_DELTA cells) is left to k_mbes_slice;
//   GA  all waves, lanes = columns of the cell grid along the reference member's trace, as in k_mbes_slice but ONCE per
//       group and with every test widened by that spread: a cell passes if the REFERENCE plane comes within
//       ext + dO + dn x (distance + box radius) of its box -- a superset of the cells any member's plane cuts;
//       the records of the cells that pass are DEDUPLICATED by source triangle (an LDS hash set) and their vertices staged
//       in LDS: 36 B per unique triangle, fetched once per group;
//   B   every wave then casts its members one after another against the staged triangles with k_mbes_slice's own
//       arithmetic, expression for expression (cut test, watertight segment end points, beam run, LDS minima): a member's
//       result is the minimum over the triangles ITS plane cuts, and the staged set contains all of them -- the same bits
//       as k_mbes_slice gives, whatever the group (tests/test_gpu_slice.py compares the two kernels bit for bit).
// Groups that overflow the staging area (SLICE_G_TRIS unique triangles) or are not tight are appended to the list
// a.slice_loose; k_mbes_slice, launched behind this kernel with that list, casts exactly their members.
#ifndef SLICE_G
#define SLICE_G 60            // particles per group: five per wave, one lane each in phase G0 (<= 64).  Per step, regular / irregular
                              // soup, 1 M x 512: 24: 3.23 / 3.90 ms, 36: 3.12 / 3.77, 48: 3.05 / 3.70, 60: 3.03 / 3.67 -- the group phases are
                              // paid once per group and neighbours in the visiting order stay tight (113 of 17 477 groups left)
#endif
#define SLICE_G_WAVES 12
#define SLICE_G_THREADS (SLICE_G_WAVES * 64)
#ifndef SLICE_G_TRIS
#define SLICE_G_TRIS 896      // unique triangles a group may stage (31.5 KB)
#endif
#define SLICE_G_HASH 2048     // slots of the de-duplication set (a power of two > 2 x SLICE_G_TRIS)
#define SLICE_G_RSLOTS 8      // phase GA1: row slots per column (threads per column of the trace)
#define SLICE_G_QUEUE 128     // phase B: cut triangles waiting per wave (u16; 12 x 256 B inside the idle hash set)
#ifndef SLICE_G_DELTA
#define SLICE_G_DELTA 1.5f    // a group is tight if its members' planes stay within this many cells of the reference's over the fan
#endif
__global__ void __launch_bounds__(SLICE_G_THREADS, 6) k_mbes_slice_group(MbesArgs a) {   // (12 waves x 2 workgroups per CU = 6 waves per SIMD: <= 85 VGPRs, <= 80 KB of LDS)
  extern __shared__ __attribute__((aligned(16))) unsigned char slice_lds[];
  const int B = a.n_beams;
  float2* tsb = (float2*)slice_lds;                  // B: (ascending tangent, 1 / cos a)
  unsigned* rng_all = (unsigned*)(tsb + B);          // SLICE_G_WAVES x B
  float* verts = (float*)(rng_all + (size_t)SLICE_G_WAVES * B);          // SLICE_G_TRIS x 9
  unsigned* hset = (unsigned*)(verts + (size_t)SLICE_G_TRIS * 9);        // SLICE_G_HASH
  unsigned short* lut = (unsigned short*)(hset + SLICE_G_HASH);          // SLICE_LUT
  float* rmeas = (float*)(lut + SLICE_LUT);                              // B: the ping's measured ranges
  __shared__ MbesPose g_pose[SLICE_G];   // the members' records (G0 has them in registers: phase B reads them from here)
  __shared__ u32 g_rec[SLICE_G];         // ... and which pose record each is (its position, or the entry of the hand-over list it came from)
  __shared__ float g_ref[24];     // the reference member's geometry and the group's widened extents
  __shared__ int g_int[8];        // k_lo, k_hi, Im, Iq, major_x, tight, ntri
  __shared__ unsigned g_ntri, g_ncand;
  if (a.in_list && (long long)blockIdx.x * SLICE_G >= (long long)*a.in_count) return;   // (hand-over kernel: nothing for this workgroup -- before the tables)
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float2 sc = a.beam_sc[b];
    const float sec = __builtin_amdgcn_rcpf(sc.y);
    tsb[b] = make_float2(sc.x * sec, sec);
    rmeas[b] = a.ranges[b];
  }
  __syncthreads();
  const float lut_lo = tsb[0].x, lut_w = fmaxf((tsb[B - 1].x - tsb[0].x) * (1.f / SLICE_LUT), 1e-12f), lut_iw = 1.f / lut_w;
  for (int k = threadIdx.x; k < SLICE_LUT; k += blockDim.x) {
    const float T = lut_lo + (float)k * lut_w;
    int lo = 0, hi = B;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (tsb[mid].x < T) lo = mid + 1; else hi = mid;
    }
    lut[k] = (unsigned short)lo;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* rng = rng_all + (size_t)w * B;
  const MeshArgs& ma = a.mesh;
  const float cs = ma.cs, ics = 1.f / cs;
  const unsigned rmax_bits = __float_as_uint(a.r_max);
  const float tan_lo = tsb[0].x, tan_hi = tsb[B - 1].x;
  const float2 sc_lo = a.beam_sc[0], sc_hi = a.beam_sc[B - 1];
  double wmax = -__builtin_inf();
  const long long n_in = a.in_list ? (long long)*a.in_count : a.n;   // (hand-over kernel of the TIN sweep: the sweep's list)
  const long long ngroups = (n_in + SLICE_G - 1) / SLICE_G;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long long i0 = g * SLICE_G;
    __syncthreads();   // (the previous group's staging area and flags are free)
    for (int k = threadIdx.x; k < SLICE_G_HASH; k += blockDim.x) hset[k] = 0xffffffffu;
    if (threadIdx.x == 0) g_ntri = g_ncand = 0u;
    // ---- G0: wave 0, one lane per member
    if (w == 0) {
      const long long p_in = i0 + lane;
      const bool in = lane < SLICE_G && p_in < n_in;
      MbesPose P;
      P.um = P.vm = 0.0;
      P.oz = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) P.c1[r] = P.c2[r] = 0.f;
      if (in) {
        const long long i = a.in_list ? (long long)a.in_list[p_in] : p_in;
        P = a.pose[i];
        g_pose[lane] = P;
        g_rec[lane] = (u32)i;
      }
      const float c2z = P.c2[2];
      const float h1 = P.c1[0] * P.c1[0] + P.c1[1] * P.c1[1];
      const bool ok = in && fabs(P.um) < 1e9 && fabs(P.vm) < 1e9 && c2z >= 0.5f && h1 >= 0.25f;   // (k_mbes_slice's own test)
      const unsigned long long okm = __ballot(ok);
      const int ref = okm ? __ffsll((long long)okm) - 1 : 0;
      // the member's fan extents (k_mbes_slice: s_pos, s_neg, t_a, t_b)
      const float oz = P.oz;
      float s_pos = 0.f, s_neg = 0.f;
      {
        const float rate_hi = sc_hi.y * c2z - sc_hi.x * P.c1[2], rate_lo = sc_lo.y * c2z - sc_lo.x * P.c1[2];
        const float rho_hi = rate_hi > 1e-4f ? fminf(a.r_max, fmaxf(oz - a.zmin_map, 0.f) * __builtin_amdgcn_rcpf(rate_hi)) : a.r_max;
        const float rho_lo = rate_lo > 1e-4f ? fminf(a.r_max, fmaxf(oz - a.zmin_map, 0.f) * __builtin_amdgcn_rcpf(rate_lo)) : a.r_max;
        s_pos = fmaxf(sc_hi.x, 0.f) * rho_hi + cs;
        s_neg = fmaxf(-sc_lo.x, 0.f) * rho_lo + cs;
      }
      const float s_abs = fmaxf(s_pos, s_neg);
      const float rc = __builtin_amdgcn_rcpf(fmaxf(c2z, 0.5f));
      const float t_b = fminf(((oz - a.zmin_map) + fabsf(P.c1[2]) * s_abs) * rc, a.r_max) + cs;
      const float t_a = fmaxf(((oz - a.zmax_map) - fabsf(P.c1[2]) * s_abs) * rc - cs, 0.f);
      const float nx = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
                  nz = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
      // the reference member's values in every lane
      const double um_r = __shfl(P.um, ref, 64), vm_r = __shfl(P.vm, ref, 64);
      const float oz_r = __shfl(oz, ref, 64), nx_r = __shfl(nx, ref, 64), ny_r = __shfl(ny, ref, 64), nz_r = __shfl(nz, ref, 64);
      // spread about the reference: sensor distance (metres) and normal difference
      const float dux = (float)((P.um - um_r) * (double)cs), duy = (float)((P.vm - vm_r) * (double)cs), duz = oz - oz_r;
      float dO = ok ? sqrtf(dux * dux + duy * duy + duz * duz) : 0.f;
      float dn = ok ? sqrtf((nx - nx_r) * (nx - nx_r) + (ny - ny_r) * (ny - ny_r) + (nz - nz_r) * (nz - nz_r)) : 0.f;
      float sp = ok ? s_pos : 0.f, sn = ok ? s_neg : 0.f, ta = ok ? t_a : 1e30f, tb = ok ? t_b : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        dO = fmaxf(dO, __shfl_xor(dO, o, 64));
        dn = fmaxf(dn, __shfl_xor(dn, o, 64));
        sp = fmaxf(sp, __shfl_xor(sp, o, 64));
        sn = fmaxf(sn, __shfl_xor(sn, o, 64));
        ta = fminf(ta, __shfl_xor(ta, o, 64));
        tb = fmaxf(tb, __shfl_xor(tb, o, 64));
      }
      // (dn x reach: the furthest point of the union's fan region from the reference sensor; 1.25: the members' in-plane
      //  axes differ too -- their (s, t) frames are turned against the reference's by at most dn)
      const float reach = sqrtf(fmaxf(sp, sn) * fmaxf(sp, sn) + tb * tb);
      const float delta = 1.25f * (dO + dn * reach) + 1e-3f;
      if (lane == ref) {
        g_ref[0] = P.c1[0]; g_ref[1] = P.c1[1]; g_ref[2] = P.c1[2];
        g_ref[3] = P.c2[0]; g_ref[4] = P.c2[1]; g_ref[5] = P.c2[2];
        g_ref[6] = oz; g_ref[7] = nx; g_ref[8] = ny; g_ref[9] = nz;
        g_ref[10] = sp + delta; g_ref[11] = sn + delta;
        g_ref[12] = fmaxf(ta - delta, 0.f); g_ref[13] = tb + delta;
        g_ref[14] = dO; g_ref[15] = dn; g_ref[16] = delta;
        const int Iu = (int)floor(P.um), Iv = (int)floor(P.vm);
        g_int[0] = Iu; g_int[1] = Iv;
        g_ref[17] = (float)(P.um - (double)Iu); g_ref[18] = (float)(P.vm - (double)Iv);
        {   // the reference sensor in the map frame, fp32 part + rest (like every member's below)
          const double Ox = ma.x0 + P.um * (double)cs, Oy = ma.y0 + P.vm * (double)cs;
          const float Oxf = (float)Ox, Oyf = (float)Oy;
          g_ref[19] = Oxf; g_ref[20] = (float)(Ox - (double)Oxf); g_ref[21] = Oyf; g_ref[22] = (float)(Oy - (double)Oyf);
        }
        g_int[5] = (okm != 0ull && delta <= SLICE_G_DELTA * cs) ? 1 : 0;
      }
    }
    __syncthreads();
    bool tight = g_int[5] != 0;
    if (tight) {
      // ---- GA: the reference member's columns, every test widened by the group's spread; all waves, a chunk of 64
      // columns per wave at a time
      const float c1x = g_ref[0], c1y = g_ref[1], c2x = g_ref[3], c2y = g_ref[4], c2zr = g_ref[5];
      const float ozr = g_ref[6], nxr = g_ref[7], nyr = g_ref[8], nzr = g_ref[9];
      const float s_pos = g_ref[10], s_neg = g_ref[11], t_a = g_ref[12], t_b = g_ref[13];
      const float dO = g_ref[14], dn = g_ref[15], delta = g_ref[16];
      const float rOxf = g_ref[19], rdOx = g_ref[20], rOyf = g_ref[21], rdOy = g_ref[22];
      (void)c2zr;
      const bool major_x = fabsf(c1x) >= fabsf(c1y);
      const float c1m = major_x ? c1x : c1y, c1q = major_x ? c1y : c1x;
      const float c2m = major_x ? c2x : c2y, c2q = major_x ? c2y : c2x;
      const int gm = major_x ? ma.gx : ma.gy, gq = major_x ? ma.gy : ma.gx;
      const int Im = major_x ? g_int[0] : g_int[1], Iq = major_x ? g_int[1] : g_int[0];
      const float fm = major_x ? g_ref[17] : g_ref[18], fq = major_x ? g_ref[18] : g_ref[17];
      const float m0 = fminf(-s_neg * c1m, s_pos * c1m) + fminf(-t_a * c2m, -t_b * c2m) - delta;
      const float m1 = fmaxf(-s_neg * c1m, s_pos * c1m) + fmaxf(-t_a * c2m, -t_b * c2m) + delta;
      int k_lo = Im + (int)floorf(fm + m0 * ics), k_hi = Im + (int)floorf(fm + m1 * ics);
      k_lo = max(k_lo, 0);
      k_hi = min(k_hi, gm - 1);
      const float rmq = c1q * __builtin_amdgcn_rcpf(c1m);
      const float kq = c2m * rmq - c2q;
      // GA1: a thread per (column, row slot) -- 96 columns x 8 row slots per pass, so the cell words of the whole trace are
      // ONE round of loads over all twelve waves (until late in round 5: lanes = columns, two or three waves busy, a
      // column's rows in batches of four); the records of the cells that pass go onto a candidate list in LDS
      unsigned* clist = rng_all;   // (the members' range slots are not in use yet: their area holds the list)
      const u32 CL = (u32)(SLICE_G_WAVES * B);   // capacity in words
      for (int cbase = k_lo; cbase <= k_hi; cbase += SLICE_G_THREADS / SLICE_G_RSLOTS) {
        const int col = cbase + (int)(threadIdx.x / SLICE_G_RSLOTS);
        const bool col_ok = col <= k_hi;
        const float ma0 = ((float)(col - Im) - fm) * cs, ma1 = ma0 + cs;
        const float q0 = fminf(ma0 * rmq, ma1 * rmq) + fminf(t_a * kq, t_b * kq) - delta;
        const float q1 = fmaxf(ma0 * rmq, ma1 * rmq) + fmaxf(t_a * kq, t_b * kq) + delta;
        int r_lo = Iq + (int)floorf(fq + q0 * ics), r_hi = Iq + (int)floorf(fq + q1 * ics);
        r_lo = max(r_lo, 0);
        r_hi = col_ok ? min(r_hi, gq - 1) : r_lo - 1;
        for (int row = r_lo + (int)(threadIdx.x % SLICE_G_RSLOTS); row <= r_hi; row += SLICE_G_RSLOTS) {
          const size_t c = major_x ? (size_t)col * ma.gy + row : (size_t)row * ma.gy + col;
          const uint2 info = ma.cell_info[c];
          u32 cnt = info.y >> 27;
          if (cnt == 0u) continue;
          const int ci = major_x ? col : row, cj = major_x ? row : col;
          const u32 rs = info.y & 0x7ffffffu;
          float zlo, zhi;
          cell_zrange(info.x, zlo, zhi);
          const float I0x = (float)(ci - g_int[0]) - g_ref[17], I0y = (float)(cj - g_int[1]) - g_ref[18];
          const float cxm = (I0x + 0.5f) * cs, cym = (I0y + 0.5f) * cs, czm = 0.5f * (zlo + zhi) - ozr;
          const float dist = nxr * cxm + nyr * cym + nzr * czm;
          const float hz = 0.5f * (zhi - zlo);
          const float ext = 0.5f * cs * (fabsf(nxr) + fabsf(nyr)) + (hz + 1e-3f) * fabsf(nzr) + 1e-4f;
          // a member's plane: |n_p . (c - O_p)| <= |n_ref . (c - O_ref)| + dn |c - O_ref| + dO, and its own extent is
          // within dn x (box radius) of the reference's
          const float rad = sqrtf(0.5f * cs * cs + hz * hz);
          const float wide = dO + dn * (sqrtf(cxm * cxm + cym * cym + czm * czm) + 2.f * rad) + 1e-3f;
          if (!(fabsf(dist) <= ext + wide)) continue;
          if (cnt == 31u) cnt = ma.cell_start[(size_t)ci * ma.gy + cj + 1] - rs;
          const u32 pos = atomicAdd(&g_ncand, cnt);
          for (u32 k = 0; k < cnt && pos + k < CL; ++k) clist[pos + k] = rs + k;
        }
      }
      __syncthreads();
      const u32 ncand = g_ncand;
      if (ncand > CL) {
        if (threadIdx.x == 0) g_ntri = SLICE_G_TRIS + 1u;   // (too many candidates: the group is left to k_mbes_slice)
      } else {
        // GA2: lanes = candidate records: the three vertices in one batch of loads; can ANY member's plane separate
        // them?  Not if all three lie on one side of the reference plane by more than a member's plane can differ from it
        // there (dO + dn x distance).  The survivors are de-duplicated by source triangle and staged
        for (u32 k = threadIdx.x; k < ncand; k += SLICE_G_THREADS) {
          const size_t r = clist[k];
          const float4 v0 = ma.cell_tri[3 * r], v1 = ma.cell_tri[3 * r + 1], v2 = ma.cell_tri[3 * r + 2];
          {
            const float ax0 = (v0.x - rOxf) - rdOx, ay0 = (v0.y - rOyf) - rdOy, az0 = v0.z - ozr;
            const float ax1 = (v1.x - rOxf) - rdOx, ay1 = (v1.y - rOyf) - rdOy, az1 = v1.z - ozr;
            const float ax2 = (v2.x - rOxf) - rdOx, ay2 = (v2.y - rOyf) - rdOy, az2 = v2.z - ozr;
            const float e0 = fmaf(nxr, ax0, fmaf(nyr, ay0, nzr * az0)), e1 = fmaf(nxr, ax1, fmaf(nyr, ay1, nzr * az1)),
                        e2 = fmaf(nxr, ax2, fmaf(nyr, ay2, nzr * az2));
            const float w0 = dO + dn * sqrtf(ax0 * ax0 + ay0 * ay0 + az0 * az0) + 1e-3f;
            const float w1 = dO + dn * sqrtf(ax1 * ax1 + ay1 * ay1 + az1 * az1) + 1e-3f;
            const float w2 = dO + dn * sqrtf(ax2 * ax2 + ay2 * ay2 + az2 * az2) + 1e-3f;
            if ((e0 > w0 && e1 > w1 && e2 > w2) || (e0 < -w0 && e1 < -w1 && e2 < -w2)) continue;
          }
          const u32 tid = __float_as_uint(v0.w);
          // first record of this triangle in the group?  (open addressing; the set is twice the staging capacity)
          u32 hslot = (tid * 2654435761u) >> (32 - 11);
          bool fresh = false;
          for (int probe = 0; probe < SLICE_G_HASH; ++probe) {
            const u32 prev = atomicCAS(&hset[hslot], 0xffffffffu, tid);
            if (prev == 0xffffffffu) {
              fresh = true;
              break;
            }
            if (prev == tid) break;
            hslot = (hslot + 1u) & (SLICE_G_HASH - 1);
          }
          if (!fresh) continue;
          const u32 slot = atomicAdd(&g_ntri, 1u);
          if (slot >= SLICE_G_TRIS) continue;   // (overflow: the group is handed to k_mbes_slice below)
          float* vp = verts + 9 * (size_t)slot;
          vp[0] = v0.x; vp[1] = v0.y; vp[2] = v0.z;
          vp[3] = v1.x; vp[4] = v1.y; vp[5] = v1.z;
          vp[6] = v2.x; vp[7] = v2.y; vp[8] = v2.z;
        }
      }
    }
    __syncthreads();
    const u32 ntri = g_ntri;
    tight = tight && ntri <= SLICE_G_TRIS;
    if (!tight) {   // left to k_mbes_slice: onto its list (order irrelevant)
      if (threadIdx.x == 0) a.slice_loose[atomicAdd(a.slice_loose_count, 1)] = (u32)g;
      continue;
    }
    // ---- B: every wave casts its members against the staged triangles (k_mbes_slice's phases B1, B2 and 4)
    for (int mbr = w; mbr < SLICE_G; mbr += SLICE_G_WAVES) {
      if (i0 + mbr >= n_in) break;
      const long long i = (long long)g_rec[mbr];
#ifdef SLICE_EXP_NOB
      if (lane == 0) a.lw[a.pose[i].slot] = 0.0;
      if (i >= 0) continue;
#endif
      MbesPose P;
      u32 slot;
      {
        const MbesPose Pv = g_pose[mbr];   // wave-uniform: scalar registers
        slot = (u32)__builtin_amdgcn_readfirstlane((int)Pv.slot);
        P.um = uniform_f64(Pv.um);
        P.vm = uniform_f64(Pv.vm);
        P.oz = uniform_f32(Pv.oz);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          P.c1[r] = uniform_f32(Pv.c1[r]);
          P.c2[r] = uniform_f32(Pv.c2[r]);
        }
      }
      const float c2z = P.c2[2];
      const float h1 = P.c1[0] * P.c1[0] + P.c1[1] * P.c1[1];
      const bool sane = fabs(P.um) < 1e9 && fabs(P.vm) < 1e9;
      if (!(sane && c2z >= 0.5f && h1 >= 0.25f)) {   // (NaN: declined) -> the general kernel, by the record's position
        if (lane == 0) a.defer_idx[atomicAdd(a.defer_count, 1)] = (u32)i;
        continue;
      }
      for (int b = lane; b < B; b += 64) rng[b] = rmax_bits;
      const double Ox = ma.x0 + P.um * (double)cs, Oy = ma.y0 + P.vm * (double)cs;
      const float Oxf = (float)Ox, Oyf = (float)Oy, dOx = (float)(Ox - (double)Oxf), dOy = (float)(Oy - (double)Oyf);
      const float oz = P.oz;
      const float nx = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
                  nz = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
      // one segment per staged triangle this member's plane separates (k_mbes_slice's arithmetic, expression for
      // expression).  The staged set is the union over the group: a member's own plane cuts well under half of it, so
      // the triangles are TESTED 64 at a time (nine LDS words, three dot products) and the ones that are cut queue up in
      // LDS; whenever 64 are waiting, the wave builds their segments and walks their beam runs with every lane busy
      // (rounds before: every lane carried its triangle through the whole body or idled -- lane utilisation 0.43).
      auto cast_tri = [&](u32 it) {
        const float* vp = verts + 9 * (size_t)it;
        const float x0 = (vp[0] - Oxf) - dOx, y0 = (vp[1] - Oyf) - dOy, z0 = vp[2] - oz;
        const float x1 = (vp[3] - Oxf) - dOx, y1 = (vp[4] - Oyf) - dOy, z1 = vp[5] - oz;
        const float x2 = (vp[6] - Oxf) - dOx, y2 = (vp[7] - Oyf) - dOy, z2 = vp[8] - oz;
        const float d0 = fmaf(nx, x0, fmaf(ny, y0, nz * z0)), d1 = fmaf(nx, x1, fmaf(ny, y1, nz * z1)),
                    d2 = fmaf(nx, x2, fmaf(ny, y2, nz * z2));
        const bool p0 = d0 > 0.f, p1 = d1 > 0.f, p2 = d2 > 0.f;
        if (p0 == p1 && p1 == p2) return;   // (not separated by this member's plane; NaN vertices: all false)
        const int L = (p0 != p1 && p0 != p2) ? 0 : ((p1 != p0 && p1 != p2) ? 1 : 2);
        const float xl = L == 0 ? x0 : (L == 1 ? x1 : x2), yl = L == 0 ? y0 : (L == 1 ? y1 : y2), zl = L == 0 ? z0 : (L == 1 ? z1 : z2);
        const float xm = L == 0 ? x1 : x0, ym = L == 0 ? y1 : y0, zm = L == 0 ? z1 : z0;
        const float xn = L == 2 ? x1 : x2, yn = L == 2 ? y1 : y2, zn = L == 2 ? z1 : z2;
        const float dl = L == 0 ? d0 : (L == 1 ? d1 : d2), dm = L == 0 ? d1 : d0, dnn = L == 2 ? d1 : d2;
        const float sl = fmaf(P.c1[0], xl, fmaf(P.c1[1], yl, P.c1[2] * zl)), tl = -fmaf(P.c2[0], xl, fmaf(P.c2[1], yl, c2z * zl));
        const float sm = fmaf(P.c1[0], xm, fmaf(P.c1[1], ym, P.c1[2] * zm)), tm = -fmaf(P.c2[0], xm, fmaf(P.c2[1], ym, c2z * zm));
        const float sn = fmaf(P.c1[0], xn, fmaf(P.c1[1], yn, P.c1[2] * zn)), tn = -fmaf(P.c2[0], xn, fmaf(P.c2[1], yn, c2z * zn));
        const bool lpos = dl > 0.f;
        float sA, tA, sB, tB;
        {
          const float db = lpos ? dm : dl, da = lpos ? dl : dm;
          const float sbv = lpos ? sm : sl, sav = lpos ? sl : sm, tbv = lpos ? tm : tl, tav = lpos ? tl : tm;
          const float lam = db * __builtin_amdgcn_rcpf(db - da);
          sA = fmaf(lam, sav - sbv, sbv);
          tA = fmaf(lam, tav - tbv, tbv);
        }
        {
          const float db = lpos ? dnn : dl, da = lpos ? dl : dnn;
          const float sbv = lpos ? sn : sl, sav = lpos ? sl : sn, tbv = lpos ? tn : tl, tav = lpos ? tl : tn;
          const float lam = db * __builtin_amdgcn_rcpf(db - da);
          sB = fmaf(lam, sav - sbv, sbv);
          tB = fmaf(lam, tav - tbv, tbv);
        }
        if (!(tA > 0.f) && !(tB > 0.f)) return;
        const float INF = __builtin_inff();
        const float cross = sA * tB - sB * tA;
        const float TA = tA > 0.f ? sA * __builtin_amdgcn_rcpf(tA) : (cross > 0.f ? INF : -INF);
        const float TB = tB > 0.f ? sB * __builtin_amdgcn_rcpf(tB) : (cross < 0.f ? INF : -INF);
        const float T_lo = fminf(TA, TB), T_hi = fmaxf(TA, TB);
        if (T_hi < tan_lo || T_lo > tan_hi) return;
        const float T_first = T_lo - 1e-6f * fmaxf(1.f, fabsf(T_lo)), T_last = T_hi + 1e-6f * fmaxf(1.f, fabsf(T_hi));
        int lo = (int)lut[min(max((int)((T_first - lut_lo) * lut_iw) - 1, 0), SLICE_LUT - 1)];
        while (lo < B && tsb[lo].x < T_first) ++lo;
        const float dts = tB - tA;
        float2 ts = tsb[min(lo, B - 1)];   // (requested one beam ahead, as in k_mbes_slice)
        for (int b = lo; b < B; ++b) {
          const float T = ts.x, sec_b = ts.y;
#ifdef SLICE_EXP_NORUN
          if (b > lo) break;
#endif
          if (T > T_last) break;
          ts = tsb[min(b + 1, B - 1)];
          slice_beam(&rng[b], T, sec_b, sA, tA, sB, tB, dts, a.r_max);
        }
      };
      unsigned short* queue = (unsigned short*)hset + (size_t)w * SLICE_G_QUEUE;   // (the hash set is idle in this phase)
      u32 qn = 0;   // wave-uniform
      for (u32 base = 0; base < ntri || qn > 0u; base += 64) {
        const u32 it = base + lane;
        bool cut = false;
        if (it < ntri) {
          const float* vp = verts + 9 * (size_t)it;
          const float x0 = (vp[0] - Oxf) - dOx, y0 = (vp[1] - Oyf) - dOy, z0 = vp[2] - oz;
          const float x1 = (vp[3] - Oxf) - dOx, y1 = (vp[4] - Oyf) - dOy, z1 = vp[5] - oz;
          const float x2 = (vp[6] - Oxf) - dOx, y2 = (vp[7] - Oyf) - dOy, z2 = vp[8] - oz;
          const float d0 = fmaf(nx, x0, fmaf(ny, y0, nz * z0)), d1 = fmaf(nx, x1, fmaf(ny, y1, nz * z1)),
                      d2 = fmaf(nx, x2, fmaf(ny, y2, nz * z2));
          const bool p0 = d0 > 0.f, p1 = d1 > 0.f, p2 = d2 > 0.f;
          cut = !(p0 == p1 && p1 == p2);
        }
        const unsigned long long cm = __ballot(cut);
        if (cut) queue[qn + (u32)__builtin_amdgcn_mbcnt_hi((u32)(cm >> 32), __builtin_amdgcn_mbcnt_lo((u32)cm, 0u))] = (unsigned short)it;
        qn += (u32)__popcll(cm);
        // 64 waiting, or the last triangles of the group: one pass of the body (ONE call site: the body is long)
        if (qn >= 64u || (base + 64u >= ntri && qn > 0u)) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const u32 take = min(qn, 64u);
          qn -= take;
          const bool act = (u32)lane < take;
          const u32 t = queue[qn + (act ? (u32)lane : 0u)];
          __builtin_amdgcn_wave_barrier();   // (read before the next pass writes behind qn)
#ifndef SLICE_EXP_NOBODY
          if (act) cast_tri(t);
#else
          if (act && t == 0xffffu) cast_tri(t);
#endif
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      float acc = 0.f;
      int nvalid = 0;
      for (int b = lane; b < B; b += 64) {
        const float e = __uint_as_float(rng[b]);
        const float rm = rmeas[b];
        if (rm > 0.f) {  // NaN fails the test
          const float d = (rm - e) * a.inv_sigma;
          acc += d * d;
          ++nvalid;
        }
      }
      const double accd = wave_sum((double)acc);
      const int nv = wave_sum(nvalid);
      if (lane == 0) {
        const double v = -0.5 * accd - (double)nv * a.lognorm;
        a.lw[slot] = v;
        wmax = v > wmax ? v : wmax;  // NaN never wins
      }
    }
  }
  if (lane == 0 && a.max_slots && wmax > -__builtin_inf())
    atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * SLICE_G_WAVES + w) & (MCL_MAX_SLOTS - 1)], ordered_key(wmax));
}
