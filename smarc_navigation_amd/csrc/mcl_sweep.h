// mcl_sweep.h -- MBES update WITHOUT a traversal per ray: the fan sweep.  Three surfaces: regularly triangulated
// height meshes (lattice walk, SURF 2 / 3), height grids with bilinear patches (SURF 0) and arbitrary height-field
// TINs (adjacency walk, SURF 5: sweep_side_tin below).  The idea, for a triangulated surface:
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
// 2-D segment intersection (~10 VALU) instead of a cell-by-cell march (~200 VALU in k_mbes_fast).  No LDS
// tile, no groups: heights come through L1/L2 (the walks of a converged cloud share their lines).
//
// The map border (lattice maps, round 4: ONE pass): the sweep reads a copy of the height array inside a one-node ring
// of NaNs (MbesArgs::grid_pad).  A walk that steps off the map loads a NaN, its next vertex becomes NaN and the test
// that ends a walk anyway (`t > 0`) fails -- no bounds test per step.  Only then, outside the loop, the lane asks
// whether the slice may END there (the beams left return r_max: a mesh has no side walls, a ray from inside a grid
// never re-enters it); the NaN's payload says which border it was.  Rounds 2-3 ran a second, bounds-checked launch
// over the particles whose footprint was not inside the map.
//
// Anything the sweep cannot prove simple -- fan too tilted for the slope bound, the nadir ray's footprint not inside
// the map, no nadir hit inside r_max, a degenerate start triangle, a slice that may come back over the border -- is
// handed over, per PARTICLE, to k_mbes_cast<., ., 2> (one wavefront per particle, reads the list's length on the
// device).  The ORDER of that list is arbitrary (atomics): every particle's result is a function of the particle alone.
#pragma once
#include "mcl_halfedge.h"
#include "mcl_mbes.h"

#ifndef SWEEP_THREADS
#define SWEEP_THREADS 256
#endif
#ifndef SWEEP_MIN_WAVES_GRID
#define SWEEP_MIN_WAVES_GRID 6   // height grids carry the conic of the current cell as well (<= 80 VGPRs; measured: 4, 5 and 6 waves run alike -- the kernel is bound by VALU issue, not by latency)
#endif
#ifndef SWEEP_MIN_WAVES_TIN
#define SWEEP_MIN_WAVES_TIN 6
#endif
#define SWEEP_TIN_GAPS 12   // gaps a side of a fan crosses by their rims before the particle is handed over (mcl_sweep.h: sweep_side_tin<.., HOLES>)
#ifndef SWEEP_MIN_WAVES
#define SWEEP_MIN_WAVES 8   // waves per SIMD the register budget is held to (<= 64 VGPRs)
#endif

// why a lane declined its particle: 1 position / tilt / footprint, 5 no nadir hit inside r_max, 6-8 degenerate start,
// 9 sensor under a grid's surface, 10 border the slice may re-cross, 11 step limit, 12 seabed above the horizon,
// 13 TIN: hole without rim records or ragged outline, 14 TIN: no way on from a rim (or more than SWEEP_TIN_GAPS gaps on one
// side), 15 TIN: the slice dips under a beam that looked into a gap, 4 TIN: nadir in a gap / off the mesh.
// -DSWEEP_REASONS counts them in MbesArgs::reasons[code] (tools/sweep_reasons.py).
#ifdef SWEEP_REASONS
#define SWEEP_NOTE(code) do { if (a.reasons) atomicAdd(&a.reasons[(code) & 15], 1u); } while (0)
#else
#define SWEEP_NOTE(code) ((void)0)
#endif
#define SWEEP_FAIL(code) do { SWEEP_NOTE(code); return false; } while (0)
// -DSWEEP_TIMELINE (tools/sweep_timeline.py): every wave of the last sweep launch leaves {start, reached the barrier, end}
// on the 100 MHz device clock and its hardware slot -- how full the machine is over the launch, where the tail begins
#ifdef SWEEP_TIMELINE
#define SWEEP_TL_WAVES 65536
__device__ unsigned long long g_sweep_tl[6 * SWEEP_TL_WAVES];
#endif
#ifndef SWEEP_SCHED_BARRIER
#define SWEEP_SCHED_BARRIER 1   // (the grid's inner loop: pins the wait for a beam record to its first use)
#endif
#ifndef SWEEP_MERGE_CXX
#define SWEEP_MERGE_CXX 0   // 1: every kernel takes the compiler's merge loop (csrc/Makefile builds that variant, tests/test_gpu_zz_merge_asm.py compares the two bit for bit)
#endif

// v_max_f32 as the hardware does it (IEEE maxNum: a NaN operand loses).  fmaxf() adds a canonicalising v_max(x, x) in
// front of every operand that comes out of memory -- one more instruction in a 12-instruction loop.
__device__ __forceinline__ float hw_max(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// The merge loop of the lattice sweep in assembly (gfx950): the beams of ONE segment (sp, tp) -> (sc, tc).  The arithmetic
// is the C++ loop's, instruction for instruction (sweep_side keeps that loop for the sub-fan and expected-range kernels,
// and tests/test_gpu_zz_merge_asm.py compares a build that takes it everywhere, bit for bit):
//   D = dss - T dts;  tau = med3(num / D, tp, tc)   (seg_tau below: dss = sc - sp, num = tp dss - sp dts per segment);
//   dd = max(z w - tau (w / cos a), (z - r_max) w);  acc += dd^2;  next record;  e_cur = sc - T' tc
// 8 VALU per beam (9 with the clamp to r_max) + one ds_read_b128 -- rounds 4-5: 9 / 10 with lam = e_prev / (e_prev - e_cur)
// clamped to [0, 1] and tau = tp + lam dts (the compiler's version of that loop: 14 -- it rotates a tangent queue through
// three registers per beam, and its own unrolling fetches two records into different registers and copies them back).
// * Record b of the beam table carries the tangent of the NEXT beam of its side (mcl_host_update.h:
//   upload_sweep_beams; rounds 2-3: of the beam after that, a two-deep queue in registers).  The loop is unrolled
//   by two and alternates between TWO record tuples, A = v[60:63] and B = v[56:59]: the tangent of the beam pending in one
//   half is the .x of the record the other half has just used, read where it lies -- no copy into a queue register.  A
//   tuple is reloaded right after the one instruction that still needs its .x.
// * Two independent chains per half: the pending beam's crossing and residual, and the NEXT beam's test against the
//   vertex (e_cur alternates between two registers): a wave has something to issue while a result is in flight.
// * The statement keeps NO state but acc and bp: it fetches the pending beam's record and its tangent itself -- the
//   tangent is the .x of the record one place back towards the nadir, bp + SWEEP_PREV_OFF on either side, read straight
//   into B.x (k_mbes_sweep stages the table with a record in front of either side's first beam for this) -- and forms
//   the test that at least one lane has a beam on this segment.  (As an in / out operand the tangent cost a move on
//   entry and two under two exec masks on exit; the record as an operand pinned to a tuple four moves around EVERY
//   statement: the register allocator will not leave a value in a physical register between two asm statements.)  The tuples are plain clobbers;
//   their fields are named v56 .. v63 in the text -- inline assembly has no way to name the parts of a tuple operand, and
//   four ds_read_b32 into free-standing registers cost an 8-way bank conflict each as soon as the lanes of a wave stand
//   at different beams (+ 20 % on the sigma = 50 m cloud).
// * A lane leaves the loop (its exec bit is cleared) when its pending beam passes beyond the vertex (e_cur < 0 or NaN:
//   the sentinel records end every table).  On exit no fetch is in flight and bp belongs to the pending beam.
// * The table is walked by the ds_read's immediate offset: one pointer add per TWO beams.  An immediate cannot be
//   negative and the two sides walk in opposite directions, so bp is kept BELOW the pending record: by 64 bytes on
//   side 0 (ascending: pending at bp + 64, the next two at + 80, + 96, the one before at + 48), by 32 on side 1
//   (descending: pending at + 32, the next two at + 16, + 0, the one before at + 48).
// * ONE statement for both sides, a scalar branch between two copies: two statements under `if (side)` made the
//   compiler copy every in / out operand before and after them -- seven moves per walk step.
// (hazards: the v_rcp result is first read three instructions later; SALU reads of VCC after v_cmp and VALU after a
//  write of EXEC are interlocked.)
typedef float sweep_rec __attribute__((ext_vector_type(4)));   // {side-signed tan of a beam further out, w / cos a, z w, (z - r_max) w}
#define SWEEP_BIAS0 64u      // bp = address of the pending record - bias (side 0 / side 1)
#define SWEEP_BIAS1 32u
#define SWEEP_PREV_OFF 48u   // ... and the record one place back towards the nadir is at bp + 48 on either side
// first half: the pending beam's tangent in TAN, its record in A = v[60:63] (.x: the next beam's tangent); fetches B
#define SWEEP_MERGE_HALF1(TAN, OFF1, CLAMP)                                                     \
      "v_fma_f32 %[d], -" TAN ", %[dts], %[dss]\n\t"                                     \
      "s_waitcnt lgkmcnt(0)\n\t"                                                         \
      "v_fma_f32 %[e2], -v60, %[tc], %[sc]\n\t"                                          \
      "ds_read_b128 v[56:59], %[bp]" OFF1 "\n\t"                                         \
      "v_rcp_f32 %[d], %[d]\n\t"                                                         \
      "v_cmp_le_f32 vcc, 0, %[e2]\n\t"                                                   \
      "s_andn2_b64 vcc, exec, vcc\n\t"                                                   \
      "s_or_b64 %[odd], %[odd], vcc\n\t"                                                 \
      "v_mul_f32 %[ep], %[num], %[d]\n\t"                                                \
      "v_med3_f32 %[ep], %[ep], %[tp], %[tc]\n\t"                                        \
      "v_fma_f32 %[ep], -%[ep], v61, v62\n\t"                                            \
      CLAMP("v_max_f32 %[ep], %[ep], v63\n\t")                                           \
      "v_fmac_f32 %[acc], %[ep], %[ep]\n\t"                                              \
      "s_andn2_b64 exec, exec, vcc\n\t"
// second half: the pending beam's tangent in A.x = v60, its record in B = v[56:59]; fetches A, moves bp by two records
#define SWEEP_MERGE_HALF2(OFF2, STEP2, CLAMP)                                                   \
      "v_fma_f32 %[d], -v60, %[dts], %[dss]\n\t"                                         \
      "s_waitcnt lgkmcnt(0)\n\t"                                                         \
      "v_fma_f32 %[ec], -v56, %[tc], %[sc]\n\t"                                          \
      "ds_read_b128 v[60:63], %[bp]" OFF2 "\n\t"                                         \
      STEP2 "\n\t"                                                                       \
      "v_rcp_f32 %[d], %[d]\n\t"                                                         \
      "v_cmp_le_f32 vcc, 0, %[ec]\n\t"                                                   \
      "s_nop 0\n\t"                                                                      \
      "v_mul_f32 %[ep], %[num], %[d]\n\t"                                                \
      "v_med3_f32 %[ep], %[ep], %[tp], %[tc]\n\t"                                        \
      "v_fma_f32 %[ep], -%[ep], v57, v58\n\t"                                            \
      CLAMP("v_max_f32 %[ep], %[ep], v59\n\t")                                           \
      "v_fmac_f32 %[acc], %[ep], %[ep]\n\t"                                              \
      "s_and_b64 exec, exec, vcc\n\t"
#define SWEEP_CLAMP_ON(x) x
#define SWEEP_CLAMP_OFF(x)
#define SWEEP_MERGE_ASM_TEXT(OFFP, OFF0, OFF1, OFF2, STEP2, STEP1, L1, L9, CLAMP)                \
      "s_mov_b64 %[sav], exec\n\t"                                                       \
      "s_mov_b64 %[odd], 0\n\t"                                                          \
      "ds_read_b32 v56, %[bp]" OFFP "\n\t"                                               \
      "ds_read_b128 v[60:63], %[bp]" OFF0 "\n\t"                                         \
      "s_waitcnt lgkmcnt(0)\n\t"                                                         \
      "v_fma_f32 %[ec], -v56, %[tc], %[sc]\n\t"                                          \
      "v_cmp_le_f32 vcc, 0, %[ec]\n\t"                                                   \
      "s_and_b64 exec, exec, vcc\n\t"                                                    \
      "s_cbranch_execz " L9 "f\n"                                                         \
      L1 ":\n\t"                                                                         \
      SWEEP_MERGE_HALF1("v56", OFF1, CLAMP)                                                     \
      "s_cbranch_execz " L9 "f\n\t"                                                      \
      SWEEP_MERGE_HALF2(OFF2, STEP2, CLAMP)                                                     \
      "s_cbranch_execnz " L1 "b\n"                                                        \
      L9 ":\n\t"                                                                         \
      "s_mov_b64 exec, %[odd]\n\t"                                                       \
      STEP1 "\n\t"                                                                       \
      "s_mov_b64 exec, %[sav]\n\t"                                                       \
      "s_waitcnt lgkmcnt(0)"
// sel = side + 2 * noclamp (wave-uniform).  noclamp: the host has proved for this launch that no beam can reach the seabed
// beyond r_max (mcl_host_update.h: launch_mbes -- depth, roll and pitch are the odometry's on every particle and the
// map has a lowest point), so max(residual, (z - r_max) w) is the residual: one instruction per beam less.
#define SWEEP_MERGE_SIDE0(L1, L9, CLAMP) \
  SWEEP_MERGE_ASM_TEXT(" offset:48", " offset:64", " offset:80", " offset:96", "v_add_u32 %[bp], 32, %[bp]", "v_add_u32 %[bp], 16, %[bp]", L1, L9, CLAMP)
#define SWEEP_MERGE_SIDE1(L1, L9, CLAMP) \
  SWEEP_MERGE_ASM_TEXT(" offset:48", " offset:32", " offset:16", "", "v_subrev_u32 %[bp], 32, %[bp]", "v_subrev_u32 %[bp], 16, %[bp]", L1, L9, CLAMP)
__device__ __forceinline__ void sweep_merge_asm(int sel, float& acc, unsigned& bp, float dss, float num, float tp, float sc,
                                                float tc, float dts) {
  float ep, d, ec, e2;
  unsigned long long sav, odd;
  asm volatile("s_cmp_lt_u32 %[sel], 2\n\t"
               "s_cbranch_scc0 52f\n\t"
               "s_cmp_lg_u32 %[sel], 0\n\t"
               "s_cbranch_scc1 51f\n\t"
               SWEEP_MERGE_SIDE0("11", "19", SWEEP_CLAMP_ON) "\n\t"
               "s_branch 60f\n"
               "51:\n\t"
               SWEEP_MERGE_SIDE1("21", "29", SWEEP_CLAMP_ON) "\n\t"
               "s_branch 60f\n"
               "52:\n\t"
               "s_cmp_lg_u32 %[sel], 2\n\t"
               "s_cbranch_scc1 53f\n\t"
               SWEEP_MERGE_SIDE0("31", "39", SWEEP_CLAMP_OFF) "\n\t"
               "s_branch 60f\n"
               "53:\n\t"
               SWEEP_MERGE_SIDE1("41", "49", SWEEP_CLAMP_OFF) "\n"
               "60:"
               : [acc] "+v"(acc), [bp] "+v"(bp),
                 [ep] "=&v"(ep), [d] "=&v"(d), [ec] "=&v"(ec), [e2] "=&v"(e2), [sav] "=&s"(sav), [odd] "=&s"(odd)
               : [dss] "v"(dss), [num] "v"(num), [tp] "v"(tp), [sc] "v"(sc), [tc] "v"(tc), [dts] "v"(dts), [sel] "s"(sel)
               : "vcc", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// The crossing of the half line s = T t with the segment (sp, tp) -> (sc, tc), as its t: with ds = sc - sp, dt = tc - tp,
//   tau = (tp ds - sp dt) / (ds - T dt) = num / D,   num a constant of the segment, D one fma per beam
// -- taken between the segment's ends by v_med3_f32 (a beam that runs along its segment: D -> 0, any t of the segment is
// right; D = 0, num = 0: NaN -> med3 returns the smaller end).  Rounds 2-5 formed lam = e_prev / (e_prev - e_cur) clamped to
// [0, 1] and tau = tp + lam dt: one instruction more per beam.  The same three instructions in sweep_merge_asm.
__device__ __forceinline__ float seg_tau(float num, float D, float tp, float tc) {
  return __builtin_amdgcn_fmed3f(num * __builtin_amdgcn_rcpf(D), tp, tc);
}

struct SweepNode {
  int P;        // lattice coordinates relative to the sensor's cell, packed i * 65536 + j (j signed)
  float d;      // signed distance to the fan plane (scaled)
  float s, t;   // in-plane coordinates, s mirrored so that it grows outward on this lane's side
};

// SURF 2: every cell split along 00-11; SURF 3: along 10-01.  (Height grids, SURF 0: sweep_side_grid below.)
// Returns false when the particle has to go to the general kernel.  acc: sum over this side's beams of
// ((range - expected) * weight)^2; EXPECT_ONLY: expected ranges to exp_row[b] instead.
// The map border: only the nadir ray (cast without bounds tests) has to stay inside the map; a slice that leaves the
//   map ENDS there (the NaN ring, top of this file): the beams left return r_max.  That conclusion needs the slice to
//   cross the border line once: the border's trace in the fan plane must be steeper than the slice can be, which the
//   fan's tilt, the map's steepest slope and the angle between fan and border decide (sweep_border_final); a level
//   vehicle always qualifies, a tilted fan slanting along the border over steep terrain goes to the traversal kernel.
// SUB (sub-fans, small clouds): the beams of a side are split over `nsub` lanes; each resolves only its own run of
//   beams [sub * per, (sub + 1) * per) and starts its walk where the first of them meets the seabed (see the start of
//   the walk below) -- the walk is a chain of dependent loads, so at 65 536 particles one lane per side leaves the
//   chip three quarters empty and every lane waiting; four lanes per side fill it, each with a quarter of the walk.
// May a slice that has just stepped off the map end there?  h: the NaN it loaded (payload 1: an x side, 2: a y side,
// 3: a corner).  In plane coordinates the border x = x_b is the line s = s_L + k t with |k| <= sin(tilt) / |c1x|, the
// slice is t = f(s) with |f'| <= (slope + sin(tilt)) / (cos(tilt) - slope sin(tilt)); they meet once if |k f'| < 1
// (0.9 here).  c1x, c1y: the map components of the fan's across-track unit vector (sign irrelevant).
__device__ __forceinline__ bool sweep_border_final(const MbesArgs& a, float h, float c2z, float c1x, float c1y) {
  const unsigned side = __float_as_uint(h) & 3u;
  const float sb = fast_sqrt(fmaxf(1.f - c2z * c2z, 0.f));
  const float lhs = sb * (a.sweep_slope + sb), rhs = 0.9f * (c2z - a.sweep_slope * sb);
  return (side == 1u || side == 2u) & (lhs < rhs * fabsf(side == 1u ? c1x : c1y));
}

template <int SURF, bool EXPECT_ONLY, bool SUB = false>
__device__ __forceinline__ bool sweep_side(const MbesArgs& a, const MbesPose& P, const float4* __restrict__ sbeam,
                                           const float* __restrict__ stail, int side, int sub, int nsub,
                                           float* __restrict__ exp_row, float& acc_out) {
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
  // beams of this side (of this lane's run of them), outward from the nadir
  const int nb_side = side ? a.b_split : B - a.b_split;
  const int per = SUB ? max((nb_side + nsub - 1) / nsub, 2) : nb_side;   // (>= 2: a later run finds its first two tangents in the records of the two beams before it)
  const int first = SUB ? min(sub * per, nb_side) : 0, last = SUB ? min(first + per, nb_side) : nb_side;
  int ptr = side ? a.b_split - 1 - first : a.b_split + first;
  const int pstep = side ? -1 : 1, pend = side ? a.b_split - 1 - last : a.b_split + last;
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
    const float s_hi = s_stop + 2.f * res;
    const float ax = sg * P.c1[0] * inv_res, ay = sg * P.c1[1] * inv_res, bx = -P.c2[0] * inv_res, by = -P.c2[1] * inv_res;
    const float fi0 = (float)I0, fj0 = (float)J0;
    // only the nadir ray (cast below without bounds tests) has to stay inside: t in [t_lo, t_hi] along -c2
    const float n0 = fminf(t_lo * bx, t_hi * bx), n1 = fmaxf(t_lo * bx, t_hi * bx);
    const float m0 = fminf(t_lo * by, t_hi * by), m1 = fmaxf(t_lo * by, t_hi * by);
    pre = pre & (fi0 + n0 >= 3.f) & (fi0 + n1 <= (float)(nx - 5)) & (fj0 + m0 >= 3.f) & (fj0 + m1 <= (float)(ny - 5));
    pre = pre & (s_hi * fmaxf(fabsf(ax), fabsf(ay)) + fmaxf(-t_lo, t_hi) * fmaxf(fabsf(bx), fabsf(by)) < 30000.f);  // packed coordinates
  }
  if (!pre) SWEEP_FAIL(1);
  // the height array inside its ring of NaNs: pitch nyp = ny + 2, node (i, j) at [(i + 1) * nyp + j + 1]
  const int nyp = a.nyp;
  const int g0i = (I0 + 1) * nyp + (J0 + 1);  // (maps below 2^30 nodes: checked on the host)
  const float* __restrict__ gp = a.grid_pad + (size_t)g0i;  // node (I0, J0)
  // ... as a raw buffer (stride 0, num_records in bytes): out-of-range reads return 0
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.grid_pad, 0, (nx + 2) * nyp * 4, 0x00020000);
  const int ny4 = nyp * 4, g0b = g0i * 4;
  // ---- start of the walk: the nadir hit, by the ordinary clearance traversal on the global height array.  A later
  // run of a side's beams (SUB, first > 0) starts where ITS first beam meets the seabed instead -- the same traversal
  // along that beam, a few dozen cells -- and walks on from there: under the tilt bound the slice is a graph over s, so
  // everything the sweep's argument needs holds from any exact hit outward.  If that beam has no hit inside r_max the
  // lane falls back to walking out from the nadir (it resolves nothing on the way: its first beam is still pending).
  float dxs = -P.c2[0], dys = -P.c2[1], dzs = -c2z, r0 = 0.f;
  bool own_start = false;
  if (SUB && first > 0 && !none) {
    const float2 sc = a.beam_sc[ptr];
    dxs = sc.x * P.c1[0] - sc.y * P.c2[0];
    dys = sc.x * P.c1[1] - sc.y * P.c2[1];
    dzs = sc.x * P.c1[2] - sc.y * c2z;
    // like the nadir ray, the start ray is cast without bounds tests: down to z_min (or r_max) it has to stay inside
    // the map with three nodes of margin -- else this lane walks out from the nadir like the first run
    const float te = dzs < -1e-4f ? fminf(a.r_max, (a.zmin_map - oz) * fast_rcp(dzs) + res) : a.r_max;
    const float ex = te * dxs * inv_res, ey = te * dys * inv_res;
    own_start = ((float)I0 + fminf(ex, 0.f) >= 3.f) & ((float)I0 + fmaxf(ex, 0.f) <= (float)(nx - 5)) &
                ((float)J0 + fminf(ey, 0.f) >= 3.f) & ((float)J0 + fmaxf(ey, 0.f) <= (float)(ny - 5));
    if (!own_start) {
      dxs = -P.c2[0];
      dys = -P.c2[1];
      dzs = -c2z;
    }
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    r0 = cast_clear<SURF>(gp, nyp, a, ul, vl, oz, dxs * inv_res, dys * inv_res, dzs, a.zmax_map, a.r_max);
    if (!SUB || !own_start || ((r0 < a.r_max) & (r0 > 0.f))) break;
    dxs = -P.c2[0];
    dys = -P.c2[1];
    dzs = -c2z;
    own_start = false;
  }
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
    const float h = gp[i * nyp + j];
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
  {
    const float uh = fmaf(r0, dxs * inv_res, ul), vh = fmaf(r0, dys * inv_res, vl);
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
    const bool p0b = __float_as_int(N0.d) >= 0, p1b = __float_as_int(N1.d) >= 0, p2b = __float_as_int(N2.d) >= 0;   // (sides of the plane by the sign bit, like the walk)
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
    const bool pl = __float_as_int(NL.d) >= 0;
    A = pl ? NF : NL;   // (along the walk A is the node found last and Bn the other end of the edge the slice leaves
    Bn = pl ? NL : NF;  //  through: plane functions of opposite sign bits)
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
  // the next beam to resolve stays in registers across segments (a vertex it passes beyond costs no LDS read), the one
  // after it is already on its way from LDS: the table is walked by pointer, one add per beam
  // (record b carries the tangent of the NEXT beam of its side in .x: the decision to leave the merge loop never waits
  //  for the record that has just been requested)
  // (by its LDS byte address: the merge loop of the main kernels is assembly, sweep_merge_asm above)
  // (the assembly loop walks the table through immediate offsets, which cannot be negative: every table address is
  //  kept low by the side's bias -- sweep_merge_asm -- and (bp - sb_off) >> 4 is still the beam)
  const unsigned sb_off = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)sbeam -
                          ((!EXPECT_ONLY && !SUB && !SWEEP_MERGE_CXX) ? (side ? SWEEP_BIAS1 : SWEEP_BIAS0) : 0u);
  unsigned bp = sb_off + (unsigned)(ptr * 16);
  const unsigned bp_end = sb_off + (unsigned)(pend * 16);
  const int msel = side + 2 * a.sweep_noclamp;
  const int pstep16 = pstep * 16;
  float tcur = stail[a.n_beams + side];   // tan of the pending beam (side-signed): all the state the C++ loop keeps
  if (SUB && first > 0) tcur = sbeam[ptr - pstep].x;   // (a later run of the side: in the record one beam back)
  sweep_rec bm;   // the pending beam's record
  {
    const float4 r = sbeam[ptr];
    bm.x = r.x;
    bm.y = r.y;
    bm.z = r.z;
    bm.w = r.w;
  }
  // one step of the walk: resolve the beams of the segment (sp, tp) -> (sc, tc), then cross into the next triangle.
  // Returns true when the walk is over (all beams resolved, stop distance, map border, failure).
  // (EXITS: the three tests that end a walk normally are made every SECOND step -- a step too many finds its beams
  //  beyond the stop distance, at r_max, or none; the test for a NaN / a horizon stays in every step)
  const auto walk_step = [&](float& sp, float& tp, float& sc, float& tc, const int step, auto EXITS) -> bool {
    // the third node of the triangle across (A, Bn): its height load is in flight while the beams are resolved
    const int Nk = (int)((unsigned)A.P + (unsigned)Bn.P - (unsigned)C);
    const int nj = __builtin_amdgcn_sbfe(Nk, 0, 16), ni = (Nk - nj) >> 16;
    // (the footprint test keeps a sane walk inside the map; a NaN-driven one is stopped by the buffer's own range check:
    //  a raw buffer load beyond num_records returns 0 -- no clamp, no 64-bit address arithmetic.  |ni| < 30000 and
    //  4 ny < 2^23 -- checked on the host --: the full-rate 24-bit multiply)
    const float hN = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, __mul24(ni, ny4) + ((nj << 2) + g0b), 0, 0));
    const float dts = tc - tp;
    const float dss = sc - sp, num = fmaf(tp, dss, -(sp * dts));   // (the segment's constants of seg_tau)
    {
      // (no end-of-table test: the record beyond the last beam has tan a = +inf and tc > 0, so e_cur = -inf.  The
      //  loop is rotated: e_cur of the NEXT beam is formed at the end of the body, one compare decides)
      float e_cur = fmaf(-tcur, tc, sc);
      if (!EXPECT_ONLY && !SUB && !SWEEP_MERGE_CXX) {
        sweep_merge_asm(msel, acc, bp, dss, num, tp, sc, tc, dts);   // (msel = side + 2 noclamp: wave-uniform)
      } else {
        // one beam on the segment (prev -> cur): the crossing of the half line s = t tan a with the chord (e changes
        // sign: <= 0 at prev, >= 0 at cur); then on to the next beam of the table
        while (e_cur >= 0.f && (!SUB || bp != bp_end)) {   // until the pending beam passes beyond this vertex
          const float tau = seg_tau(num, fmaf(-tcur, dts, dss), tp, tc);
          // range = t / cos a, beyond r_max (or NaN): r_max.  The table carries the residual's constants
          // (mcl_host_update.h: upload_sweep_beams): (range_b - r) w = max(z w - t (w / cos a), (z - r_max) w)
          if (EXPECT_ONLY) {
            exp_row[(int)(bp - sb_off) >> 4] = fminf(tau * bm.y, a.r_max);
          } else {
            const float dd = hw_max(fmaf(-tau, bm.y, bm.z), bm.w);
            acc = fmaf(dd, dd, acc);
          }
          tcur = bm.x;
          bp += pstep16;
          const float4 r = sbeam[(int)(bp - sb_off) >> 4];
          bm.x = r.x;
          bm.y = r.y;
          bm.z = r.z;
          bm.w = r.w;
          e_cur = fmaf(-tcur, tc, sc);
        }
      }
    }
    if (decltype(EXITS)::value) {
      if (bp == bp_end) return true;
      if (sc > s_stop) return true;  // every beam left misses inside r_max (tail below)
      if (step > max_steps) {   // (step: the wave's own count -- every lane still walking has taken as many)
        SWEEP_NOTE(11);
        ok = false;   // (the node found last has a height: the test after the loop hands the particle over)
        return true;
      }
    }
    const float fi = (float)ni, fj = (float)nj;
    const float dN = fmaf(pu, fi, fmaf(pv, fj, fmaf(pz, hN, p0)));
    // the new node replaces the one on ITS side of the plane; the next edge joins it to the one that stays.  The new
    // node always becomes A, the one that stays moves to Bn only when it was A: five selects and two moves per step
    // (nine selects with fixed roles "A below, Bn above"); the new node's s and t are formed in A's registers, after
    // the selects have read them
    // (sides by the SIGN BIT of the plane function: one xor and one compare, no state)
    const bool keep_a = (__float_as_int(dN) ^ __float_as_int(A.d)) < 0;
    C = keep_a ? Bn.P : A.P;
    Bn.P = keep_a ? A.P : Bn.P;
    Bn.d = keep_a ? A.d : Bn.d;
    Bn.s = keep_a ? A.s : Bn.s;
    Bn.t = keep_a ? A.t : Bn.t;
    A.P = Nk;
    A.d = dN;
    A.s = fmaf(su, fi, fmaf(sv, fj, fmaf(sz, hN, s0)));
    A.t = fmaf(tu, fi, fmaf(tv, fj, fmaf(tz, hN, t0)));
    const float lam = A.d * fast_rcp(A.d - Bn.d);
    const float s_new = fmaf(lam, Bn.s - A.s, A.s), t_new = fmaf(lam, Bn.t - A.t, A.t);
    // every crossing is a vertex of the slice.  The new vertex takes the place of the one before last and the CALLER
    // swaps the roles (the walk loop is unrolled by two): no register shuffling per step
    sp = s_new;
    tp = t_new;
    if (!(t_new > 0.f)) {
      // the seabed rises above the sensor's own horizon (see the sentinel records) -- or the height was a NaN: the walk
      // has stepped off the map (decided after the loop: nothing here but the exit)
      ok = false;
      return true;
    }
    return false;
  };
  for (int step = 1;; step += 2) {
    if (walk_step(s_prev, t_prev, s_cur, t_cur, step, std::true_type())) break;
    if (walk_step(s_cur, t_cur, s_prev, t_prev, step + 1, std::false_type())) break;
  }
  if (!ok) {
    // a NaN height: the slice ends at the map border -- final if it cannot come back (the beams left get r_max through
    // the tail below).  (hN: the height of the node this lane found last, A -- loaded again here rather than kept alive
    // across the loop; su = +-c1x res: the walk keeps no other copy)
    const int lj = __builtin_amdgcn_sbfe(A.P, 0, 16), li = (A.P - lj) >> 16;
    const float hN = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, __mul24(li, ny4) + ((lj << 2) + g0b), 0, 0));
    ok = (hN != hN) && sweep_border_final(a, hN, c2z, su * inv_res, sv * inv_res);
    if (!ok) SWEEP_NOTE(hN != hN ? 10 : 12);
  }
  if (ok && bp != bp_end) {
    ptr = (int)(bp - sb_off) >> 4;
    if (EXPECT_ONLY) {
      for (; ptr != pend; ptr += pstep) exp_row[ptr] = a.r_max;
    } else {
      acc += stail[ptr];   // (SUB: the staged tail sums end with this lane's run)
    }
  }
  acc_out = acc;
  return ok;
}

// ------------------------------------------------------------------ height GRIDS (SURF 0): the cell walk
// The same sweep, cell by cell instead of triangle by triangle.  Along cell edges a bilinear patch is linear, so the
// points where the fan plane crosses CELL edges are exact; inside a cell the plane function is bilinear and -- unless
// its four corner signs alternate, which the tilt bound for grids excludes (tan(tilt) * slope < 0.45) -- its zero set
// joins the two edges whose end points differ in sign.  State: the edge (A, B) the slice leaves the current cell
// through (A: plane function <= 0, B: > 0; e = B - A and n = the step across that edge into the next cell, both as
// packed lattice vectors: the handedness of (e, n) never changes along a walk).  One step: the two far corners C = A + n
// and D = B + n of the next cell decide its exit edge -- (A, C) if C is positive, (D, B) if both are not, (C, D)
// otherwise --, and between the entry and the exit crossing the slice is an arc of a conic inside ONE cell: a beam whose
// angle the two crossings bracket meets the surface in that cell, at the root of the bilinear-patch quadratic of the
// oracle (orc_ray_grid) at which the clearance turns negative (the first hit is always an entry); a beam that passes
// just beyond the far crossing may still graze the arc -- bounded by the patch's twist -- and is then tested against the
// same quadratic, the root accepted if its point lies in the cell.
// (until round 3 the grid was walked along the 00-11 triangulation of the node values like a lattice mesh, passing
//  over the crossings of the auxiliary diagonals: two steps per cell, of which the lanes of a wave took the beam-
//  resolving one at different times -- lane utilisation 0.51, 2.8 x the instructions of the mesh walk.)
template <bool EXPECT_ONLY, bool SUB = false>
__device__ __forceinline__ bool sweep_side_grid(const MbesArgs& a, const MbesPose& P, const float4* __restrict__ sbeam,
                                                const float* __restrict__ stail, int side, int sub, int nsub,
                                                float* __restrict__ exp_row, float& acc_out) {
  acc_out = 0.f;
  const int nx = a.nx, ny = a.ny, B = a.n_beams;
  bool pre = P.um >= 1.0 && P.um < (double)(nx - 2) && P.vm >= 1.0 && P.vm < (double)(ny - 2);  // (NaN: false)
  const float c2z = P.c2[2];
  pre = pre & (c2z >= a.sweep_c2z_min);
  const double fum = pre ? floor(P.um) : 1.0, fvm = pre ? floor(P.vm) : 1.0;
  const int I0 = (int)fum, J0 = (int)fvm;
  const float ul = (float)(P.um - fum), vl = (float)(P.vm - fvm);
  const float res = a.res, inv_res = (float)a.inv_res, oz = P.oz;
  const float sg = side ? -1.f : 1.f;
  const int nb_side = side ? a.b_split : B - a.b_split;
  const int per = SUB ? max((nb_side + nsub - 1) / nsub, 2) : nb_side;
  const int first = SUB ? min(sub * per, nb_side) : 0, last = SUB ? min(first + per, nb_side) : nb_side;
  int ptr = side ? a.b_split - 1 - first : a.b_split + first;
  const int pstep = side ? -1 : 1, pend = side ? a.b_split - 1 - last : a.b_split + last;
  const bool none = ptr == pend;
  float s_stop = none ? 0.f : a.r_max;
  if (!none) {
    const float2 sc = a.beam_sc[side ? 0 : B - 1];
    const float dz_e = sc.x * P.c1[2] - sc.y * c2z;
    if (dz_e < -1e-4f) s_stop = fminf(s_stop, (a.zmin_map - oz) * fast_rcp(dz_e) * fabsf(sc.x));
  }
  s_stop += 2.f * res;
  // (s, t) -> cells: u = ul + ax s + bx t, v = vl + ay s + by t
  const float ax = sg * P.c1[0] * inv_res, ay = sg * P.c1[1] * inv_res, bx = -P.c2[0] * inv_res, by = -P.c2[1] * inv_res;
  {  // footprint (see sweep_side)
    const float rc = fast_rcp(c2z);
    const float sl = fabsf(P.c1[2]) * (s_stop + 2.f * res);
    const float t_hi = ((oz - a.zmin_map) + sl) * rc + res, t_lo = fminf(((oz - a.zmax_map) - sl) * rc - res, 0.f);
    const float s_hi = s_stop + 2.f * res;
    const float fi0 = (float)I0, fj0 = (float)J0;
    // only the start ray (cast below without bounds tests) has to stay inside the map (see sweep_side)
    const float n0 = fminf(t_lo * bx, t_hi * bx), n1 = fmaxf(t_lo * bx, t_hi * bx);
    const float m0 = fminf(t_lo * by, t_hi * by), m1 = fmaxf(t_lo * by, t_hi * by);
    pre = pre & (fi0 + n0 >= 3.f) & (fi0 + n1 <= (float)(nx - 5)) & (fj0 + m0 >= 3.f) & (fj0 + m1 <= (float)(ny - 5));
    pre = pre & (s_hi * fmaxf(fabsf(ax), fabsf(ay)) + fmaxf(-t_lo, t_hi) * fmaxf(fabsf(bx), fabsf(by)) < 30000.f);
  }
  if (!pre) SWEEP_FAIL(1);
  // the height array inside its ring of NaNs (see sweep_side)
  const int nyp = a.nyp;
  const int g0i = (I0 + 1) * nyp + (J0 + 1);
  const float* __restrict__ gp = a.grid_pad + (size_t)g0i;
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.grid_pad, 0, (nx + 2) * nyp * 4, 0x00020000);
  const int ny4 = nyp * 4, g0b = g0i * 4;
  // ---- start of the walk (see sweep_side): the nadir hit, or the hit of the first beam of a later run
  float dxs = -P.c2[0], dys = -P.c2[1], dzs = -c2z, r0 = 0.f;
  bool own_start = false;
  if (SUB && first > 0 && !none) {
    const float2 sc = a.beam_sc[ptr];
    dxs = sc.x * P.c1[0] - sc.y * P.c2[0];
    dys = sc.x * P.c1[1] - sc.y * P.c2[1];
    dzs = sc.x * P.c1[2] - sc.y * c2z;
    // like the nadir ray, the start ray is cast without bounds tests: down to z_min (or r_max) it has to stay inside
    // the map with three nodes of margin -- else this lane walks out from the nadir like the first run
    const float te = dzs < -1e-4f ? fminf(a.r_max, (a.zmin_map - oz) * fast_rcp(dzs) + res) : a.r_max;
    const float ex = te * dxs * inv_res, ey = te * dys * inv_res;
    own_start = ((float)I0 + fminf(ex, 0.f) >= 3.f) & ((float)I0 + fmaxf(ex, 0.f) <= (float)(nx - 5)) &
                ((float)J0 + fminf(ey, 0.f) >= 3.f) & ((float)J0 + fmaxf(ey, 0.f) <= (float)(ny - 5));
    if (!own_start) {
      dxs = -P.c2[0];
      dys = -P.c2[1];
      dzs = -c2z;
    }
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    r0 = cast_clear<0>(gp, nyp, a, ul, vl, oz, dxs * inv_res, dys * inv_res, dzs, a.zmax_map, a.r_max);
    if (!SUB || !own_start || ((r0 < a.r_max) & (r0 > 0.f))) break;
    dxs = -P.c2[0];
    dys = -P.c2[1];
    dzs = -c2z;
    own_start = false;
  }
  if (!(r0 < a.r_max)) SWEEP_FAIL(5);
  if (!(r0 > 0.f)) SWEEP_FAIL(9);  // the sensor is at or below the seabed (a grid is solid underneath)
  if (none) return true;
  // ---- plane and in-plane coordinates as affine functions of (i, j, h): lattice coordinates relative to (I0, J0)
  const float nx_ = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny_ = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
              nz_ = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
  const float pu = nx_ * res, pv = ny_ * res, pz = nz_, p0 = -(pu * ul + pv * vl + pz * oz);
  const float su = sg * P.c1[0] * res, sv = sg * P.c1[1] * res, sz = sg * P.c1[2], s0 = -(su * ul + sv * vl + sz * oz);
  const float tu = -P.c2[0] * res, tv = -P.c2[1] * res, tz = -c2z, t0 = -(tu * ul + tv * vl + tz * oz);
  // ---- the cell under the start hit: its two crossings, the outer one is where this side leaves it
  int PA, e, n;                       // packed lattice: node A, the edge vector A -> B, the step into the next cell
  float dA, sA, tA, dB, sB, tB;       // plane function and in-plane coordinates of the edge's nodes
  float sp, tp, sc, tc;               // the arc of the current cell: entry and exit crossing
  float hp00, hp01, hp10, hp11;       // corner heights of the current cell
  {
    const float uh = fmaf(r0, dxs * inv_res, ul), vh = fmaf(r0, dys * inv_res, vl);
    const float cfi = floorf(uh), cfj = floorf(vh);
    const int ci = (int)cfi, cj = (int)cfj;
    const int c00 = ci * 65536 + cj;
    const float* cp = gp + (ci * nyp + cj);
    hp00 = cp[0];
    hp01 = cp[1];
    hp10 = cp[nyp];
    hp11 = cp[nyp + 1];
    struct Nd { float d, s, t; };
    const auto nd = [&](float fi, float fj, float h) {
      Nd N;
      N.d = fmaf(pu, fi, fmaf(pv, fj, fmaf(pz, h, p0)));
      N.s = fmaf(su, fi, fmaf(sv, fj, fmaf(sz, h, s0)));
      N.t = fmaf(tu, fi, fmaf(tv, fj, fmaf(tz, h, t0)));
      return N;
    };
    const Nd N00 = nd(cfi, cfj, hp00), N10 = nd(cfi + 1.f, cfj, hp10), N01 = nd(cfi, cfj + 1.f, hp01),
             N11 = nd(cfi + 1.f, cfj + 1.f, hp11);
    // edges in the order 00-10 (outward -j), 10-11 (+i), 01-11 (+j), 00-01 (-i)
    const float NEG = -__builtin_inff();
    float se[4], te[4];
    int cnt = 0;
    const auto cross = [&](const Nd& X, const Nd& Y, int k) {
      const bool c = (X.d > 0.f) != (Y.d > 0.f);
      const float lam = X.d * fast_rcp(X.d - Y.d);
      se[k] = c ? fmaf(lam, Y.s - X.s, X.s) : NEG;
      te[k] = fmaf(lam, Y.t - X.t, X.t);
      cnt += c ? 1 : 0;
    };
    cross(N00, N10, 0);
    cross(N10, N11, 1);
    cross(N01, N11, 2);
    cross(N00, N01, 3);
    if (cnt != 2) SWEEP_FAIL(6);  // the plane misses the cell (rounding at its border), touches a corner, or the signs alternate
    int kf = 0;
    sc = se[0];
    tc = te[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (se[k] > sc) {
        kf = k;
        sc = se[k];
        tc = te[k];
      }
    sp = NEG;
    tp = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k != kf && se[k] > sp) {
        sp = se[k];
        tp = te[k];
      }
    if (!(sp < sc)) SWEEP_FAIL(7);  // (NaN, or the two crossings coincide at a node)
    if (!(tc > 0.f)) SWEEP_FAIL(8);
    const Nd X = (kf == 0 || kf == 3) ? N00 : (kf == 1 ? N10 : N01);
    const Nd Y = kf == 0 ? N10 : (kf == 3 ? N01 : N11);
    const int PX = (kf == 0 || kf == 3) ? c00 : (kf == 1 ? c00 + 65536 : c00 + 1);
    const int PY = kf == 0 ? c00 + 65536 : (kf == 3 ? c00 + 1 : c00 + 65537);
    const bool xpos = X.d > 0.f;
    PA = xpos ? PY : PX;
    e = xpos ? PX - PY : PY - PX;
    n = kf == 0 ? -1 : (kf == 1 ? 65536 : (kf == 2 ? 1 : -65536));
    dA = xpos ? Y.d : X.d;
    sA = xpos ? Y.s : X.s;
    tA = xpos ? Y.t : X.t;
    dB = xpos ? X.d : Y.d;
    sB = xpos ? X.s : Y.s;
    tB = xpos ? X.t : Y.t;
  }
  // ---- walk outward, merging the beam table against the arcs
  float acc = 0.f;
  bool ok = true;
  const int max_steps = (int)(3.f * (s_stop + 4.f * res) * inv_res) + 16;
  int step = 0;
  const float4* bp = sbeam + ptr;
  const float4* const bp_end = sbeam + pend;
  // (record b carries the tangent of the NEXT beam of its side: the pending tangent is all the state the loop keeps;
  //  the two-deep queue of rounds 2-3 cost two moves per beam)
  float tcur = stail[a.n_beams + side];       // tan of the pending beam (side-signed)
  if (SUB && first > 0) tcur = bp[-pstep].x;
  float4 bm = bp[0];
  const float rc2z = fast_rcp(c2z);
  const float axay2 = 2.f * (ax * ay), axby2 = 2.f * fmaf(ax, by, ay * bx), bxby2 = 2.f * (bx * by);
  float hC = 0.f, hD = 0.f;   // the far corners loaded last (read after the loop: why did the walk end?)
  for (;;) {
    // the far corners of the cell across (A, B): their heights are in flight while this cell's beams are resolved
    // (the footprint test keeps a sane walk inside the map; a NaN-driven one is stopped by the buffer's own range check)
    const int PC = PA + n, PD = PC + e;
    const int cj = __builtin_amdgcn_sbfe(PC, 0, 16), ci = (PC - cj) >> 16;
    const int dj = __builtin_amdgcn_sbfe(PD, 0, 16), di = (PD - dj) >> 16;
    hC = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, __mul24(ci, ny4) + ((cj << 2) + g0b), 0, 0));
    hD = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, __mul24(di, ny4) + ((dj << 2) + g0b), 0, 0));
    // ---- the conic of the current cell (the one on the -n side of the edge): in the fan plane the clearance
    // z - h(u, v) over it is G = g0 + g1 s + g2 t + g3 s^2 + g4 s t + g5 t^2 (u, v, z affine in (s, t), h bilinear), and
    // along beam s = t tan a the quadratic g0 + (g1 T + g2) t + (g3 T^2 + g4 T + g5) t^2 -- orc_ray_grid's, with tau = t
    const int emin = min(e, 0);
    const int Pm = PA + emin - max(n, 0);
    const int j0 = __builtin_amdgcn_sbfe(Pm, 0, 16), i0 = (Pm - j0) >> 16;
    const float pB = hp10 - hp00, pC = hp01 - hp00, pD = (hp00 - hp10) - (hp01 - hp11);
    const float uc = ul - (float)i0, vc = vl - (float)j0;
    const float g0 = oz - fmaf(pD * uc, vc, fmaf(pC, vc, fmaf(pB, uc, hp00)));
    const float g1 = sz - fmaf(pD, fmaf(uc, ay, vc * ax), fmaf(pC, ay, pB * ax));
    const float g2 = tz - fmaf(pD, fmaf(uc, by, vc * bx), fmaf(pC, by, pB * bx));
    const float g3n = pD * axay2, g4n = pD * axby2, g5n = pD * bxby2;   // -2 g3, -2 g4, -2 g5
    // how far (in e = s - t tan a, per unit tan a) the arc can bulge beyond its chord: along the chord the clearance is
    // -twist * du * dv * l (1 - l) <= |twist du dv| / 4, and moving along -c2 changes the clearance at a rate of at
    // least c2z (1 - slope tan(tilt)) >= 0.55 c2z
    const float ds = sc - sp, dts = tc - tp;
    const float kb = 0.46f * fabsf(pD * fmaf(ax, ds, bx * dts) * fmaf(ay, ds, by * dts)) * rc2z + 1e-6f;
    // the corner heights of the NEXT cell
    {
      const int Pn = PA + emin + min(n, 0);
      const int j1 = __builtin_amdgcn_sbfe(Pn, 0, 16), i1 = (Pn - j1) >> 16;
      const int ob = __mul24(i1, ny4) + ((j1 << 2) + g0b);
      hp00 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, ob, 0, 0));
      hp01 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, ob + 4, 0, 0));
      hp10 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, ob, ny4, 0));
      hp11 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, ob + 4, ny4, 0));
    }
    // ---- the beams of this arc.  (No end-of-table test: the record beyond the last beam has tan a = +inf and tc > 0,
    // so e_cur = -inf.  The loop is rotated: e_cur of the NEXT beam is formed at the end of the body.)
    float e_cur = fmaf(-tcur, tc, sc);
    // The root at which the clearance turns negative, in the form that does not cancel: with
    // w = q1 + sgn(q1) sqrt(q1^2 - 4 q2 g0):  q1 < 0 (the clearance falls along the beam from the start: all but
    // strongly twisted, distant patches): -2 g0 / w;  q1 >= 0: w / (-2 q2)
    const float nG0 = -2.f * g0;
    const auto root = [&](float& disc) {
      const float q1 = fmaf(g1, tcur, g2), q2n = fmaf(fmaf(g3n, tcur, g4n), tcur, g5n);
      disc = fmaf(q2n, -nG0, q1 * q1);
      const float sq = fast_sqrt(fabsf(disc));   // (negative by rounding only: a tangent beam)
      float tau = nG0 * fast_rcp(q1 - sq);
      if (__builtin_amdgcn_ballot_w64(q1 >= 0.f) != 0ull) {   // (wave-uniform, rare: a real branch)
        asm volatile("; q1 >= 0");
        if (q1 >= 0.f) tau = (q1 + sq) * fast_rcp(q2n);
      }
      return tau;
    };
    // range = t / cos a, beyond r_max (or NaN): r_max.  The table carries the residual's constants (mcl_host_update.h:
    // upload_sweep_beams): (range_b - r) w = max(z w - t (w / cos a), (z - r_max) w); then on to the next beam
    const auto take = [&](float tau) {
      if (EXPECT_ONLY) {
        exp_row[bp - sbeam] = fminf(tau * bm.y, a.r_max);
      } else {
        const float dd = hw_max(fmaf(-tau, bm.y, bm.z), bm.w);
        acc = fmaf(dd, dd, acc);
      }
      tcur = bm.x;
      bp += pstep;
      bm = bp[0];
      e_cur = fmaf(-tcur, tc, sc);
    };
    for (;;) {
      while (e_cur >= 0.f && (!SUB || bp != bp_end)) {   // the beams the two crossings bracket
        float disc;
        const float tau = root(disc);
#if SWEEP_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // (the record's fields are first needed below: the wait for it belongs here)
#endif
        take(tau);
      }
      if (SUB && bp == bp_end) break;   // (the beams beyond belong to the next lane of this side)
      // the pending beam passes beyond the exit crossing: by more than the arc can bulge?  (sentinel: NaN)
      const float e_prev = fmaf(-tcur, tp, sp);
      if (!(hw_max(e_cur, e_prev) + tcur * kb >= 0.f)) break;
      // it may graze the arc: only if the root's point lies in this cell
      float disc;
      const float tau = root(disc);
      const float du = fmaf(ax, tcur, bx), dv = fmaf(ay, tcur, by);  // cells per unit t along the beam
      const float EPS = 2e-4f;
      const bool in = (disc >= 0.f) & (tau > 0.f) & (fabsf(fmaf(du, tau, uc) - 0.5f) <= 0.5f + EPS) &
                      (fabsf(fmaf(dv, tau, vc) - 0.5f) <= 0.5f + EPS);
      if (!in) break;
      take(tau);
    }
    if (bp == bp_end) break;
    if (sc > s_stop) break;  // every beam left misses inside r_max (tail below)
    if (++step > max_steps) {
      SWEEP_NOTE(11);
      ok = false;
      break;
    }
    // ---- across the edge: the exit edge of the next cell
    const float fci = (float)ci, fcj = (float)cj, fdi = (float)di, fdj = (float)dj;
    const float dC = fmaf(pu, fci, fmaf(pv, fcj, fmaf(pz, hC, p0)));
    const float sC = fmaf(su, fci, fmaf(sv, fcj, fmaf(sz, hC, s0)));
    const float tC = fmaf(tu, fci, fmaf(tv, fcj, fmaf(tz, hC, t0)));
    const float dD = fmaf(pu, fdi, fmaf(pv, fdj, fmaf(pz, hD, p0)));
    const float sD = fmaf(su, fdi, fmaf(sv, fdj, fmaf(sz, hD, s0)));
    const float tD = fmaf(tu, fdi, fmaf(tv, fdj, fmaf(tz, hD, t0)));
    const bool viaA = dC > 0.f;               // exit (A, C): A stays, C takes B's place
    const bool viaB = !viaA & !(dD > 0.f);    // exit (D, B): D takes A's place, B stays; else (C, D)
    PA = viaA ? PA : (viaB ? PD : PC);
    dA = viaA ? dA : (viaB ? dD : dC);
    sA = viaA ? sA : (viaB ? sD : sC);
    tA = viaA ? tA : (viaB ? tD : tC);
    dB = viaA ? dC : (viaB ? dB : dD);
    sB = viaA ? sC : (viaB ? sB : sD);
    tB = viaA ? tC : (viaB ? tB : tD);
    const int e_old = e;
    e = viaA ? n : (viaB ? -n : e);
    n = viaA ? -e_old : (viaB ? e_old : n);
    const float lam = dA * fast_rcp(dA - dB);
    sp = sc;
    tp = tc;
    sc = fmaf(lam, sB - sA, sA);
    tc = fmaf(lam, tB - tA, tA);
    if (!(tc > 0.f)) {  // the seabed rises above the sensor's own horizon -- or a NaN corner (decided below)
      ok = false;
      break;
    }
  }
  if (!ok && step <= max_steps) {
    // a NaN corner: the next cell is off the map, the slice ends at the border -- final if it cannot come back (see
    // sweep_side; C and D lie beyond the same border line)
    const float hB = hC != hC ? hC : hD;
    ok = (hB != hB) && sweep_border_final(a, hB, c2z, su * inv_res, sv * inv_res);
    if (!ok) SWEEP_NOTE(hB != hB ? 10 : 12);
  }
  if (ok && bp != bp_end) {
    ptr = (int)(bp - sbeam);
    if (EXPECT_ONLY) {
      for (; ptr != pend; ptr += pstep) exp_row[ptr] = a.r_max;
    } else {
      acc += stail[ptr];   // (SUB: the staged tail sums end with this lane's run)
    }
  }
  acc_out = acc;
  return ok;
}

// ------------------------------------------------------------------ arbitrary height-field TINs (SURF 5)
// The same sweep without the lattice: the triangle across an edge comes from the half-edge table built by
// mesh_build (mcl_mesh.h: one 32-byte record per half-edge, triangles in Morton order of their centroids): entering
// a triangle through half-edge h, record h holds the vertex the slice meets next (x, y, z) and the two half-edges it
// can leave through.  Per step: that ONE record (rounds 3-5: a triangle record, then the vertex it named -- two
// dependent loads and a dozen selects on vertex ids),
// plane function and in-plane coordinates from map-frame coordinates (the sensor position, fp64, is subtracted as an
// fp32 part and its sub-ulp rest: vertices are fp32 inputs, the sensor is not).  The walk ends at a mesh border: the map's outer border (the
// beams left return r_max, under the rule of the second pass above) or a hole / ragged outline (hand-over).
// The start triangle: the (cell, triangle) records of mcl_mesh.h carry their source triangle's index in a spare word.
__device__ __forceinline__ float tin_nadir(const MbesArgs& a, int I0, int J0, float ul, float vl, float oz, float dx, float dy,
                                           float dz, u32& tri_id) {
  // the near-vertical ray O + t (dx, dy, dz), cell by cell through the cell grid (cells of a.mesh.cs metres)
  const MeshArgs& ma = a.mesh;
  const float cs = ma.cs, ics = 1.f / cs;
  const float du = dx * ics, dv = dy * ics;
  const float rdz = fast_rcp(dz);
  float t = oz > a.zmax_map ? fmaxf((a.zmax_map - oz) * rdz - 1e-3f, 0.f) : 0.f;
  const float t1 = fminf(a.r_max, (a.zmin_map - oz) * rdz + 1e-2f);
  const float pu = fmaf(t, du, ul), pv = fmaf(t, dv, vl);
  int ci = (int)floorf(pu), cj = (int)floorf(pv);
  const int sx = du > 0.f ? 1 : -1, sy = dv > 0.f ? 1 : -1;
  const float adu = fminf(fabsf(fast_rcp(du)), 1e30f), adv = fminf(fabsf(fast_rcp(dv)), 1e30f);
  float tnx = t + (du > 0.f ? (float)(ci + 1) - pu : pu - (float)ci) * adu;
  float tny = t + (dv > 0.f ? (float)(cj + 1) - pv : pv - (float)cj) * adv;
  tri_id = 0xffffffffu;
  for (int guard = 0; guard < 64; ++guard) {
    const float t_out = fminf(fminf(tnx, tny), t1);
    const int gi = I0 + ci, gj = J0 + cj;
    if ((unsigned)gi >= (unsigned)ma.gx || (unsigned)gj >= (unsigned)ma.gy) break;  // the nadir ray leaves the map: no start
    const size_t c = (size_t)gi * ma.gy + gj;
    const u32 rs = ma.cell_start[c], re = ma.cell_start[c + 1];
    const float olx = (ul - (float)ci) * cs, oly = (vl - (float)cj) * cs;
    float best = __builtin_inff();
    for (u32 k = rs; k < re; ++k) {
      const float4 r0 = ma.tri[3 * (size_t)k], r1 = ma.tri[3 * (size_t)k + 1], r2 = ma.tri[3 * (size_t)k + 2];
      const float den = fmaf(r0.x, dx, fmaf(r0.y, dy, dz));
      const float th = (r0.z - fmaf(r0.x, olx, fmaf(r0.y, oly, oz))) * fast_rcp(den);
      const float hx = fmaf(th, dx, olx) - r1.x, hy = fmaf(th, dy, oly) - r1.y;
      const float bu = fmaf(r1.z, hx, r1.w * hy), bv = fmaf(r2.x, hx, r2.y * hy);
      const float EPS = 2e-5f;
      if (bu >= -EPS && bv >= -EPS && bu + bv <= 1.f + EPS && th >= 0.f && th <= t_out + 1e-4f && th < best) {
        best = th;
        tri_id = __float_as_uint(r2.z);
      }
    }
    if (best < __builtin_inff()) return fminf(best, a.r_max);
    if (!(t_out < t1)) break;
    const bool stepx = tnx <= tny;
    ci += stepx ? sx : 0;
    cj += stepx ? 0 : sy;
    tnx += stepx ? adu : 0.f;
    tny += stepx ? 0.f : adv;
  }
  return a.r_max;
}

template <bool EXPECT_ONLY, bool SUB = false, bool HOLES = false>
__device__ __forceinline__ bool sweep_side_tin(const MbesArgs& a, const MbesPose& P, const float4* __restrict__ sbeam,
                                               const float* __restrict__ stail, int side, int sub, int nsub,
                                               float* __restrict__ exp_row, float& acc_out) {
  acc_out = 0.f;
  const MeshArgs& ma = a.mesh;
  const int nx = a.nx, ny = a.ny, B = a.n_beams;  // (cells + 1 of the cell grid: the mesh's bounding box)
  bool pre = P.um >= 0.0 && P.um < (double)(nx - 1) && P.vm >= 0.0 && P.vm < (double)(ny - 1);  // (NaN: false)
  const float c2z = P.c2[2];
  // (HOLES, the outline linked: a sensor beyond the bounding box -- the vehicle has left the map and looks back in -- is no
  //  reason to decline: its walk starts where the fan plane meets the outline, below)
  const bool off_box = HOLES && !pre && ma.tin_outline != 0u && fabs(P.um) < 1e6 && fabs(P.vm) < 1e6 && c2z >= a.sweep_c2z_min;
  pre = pre & (c2z >= a.sweep_c2z_min);
  const double fum = pre ? floor(P.um) : 1.0, fvm = pre ? floor(P.vm) : 1.0;
  const int I0 = (int)fum, J0 = (int)fvm;
  const float ul = (float)(P.um - fum), vl = (float)(P.vm - fvm);
  const float res = a.res, inv_res = (float)a.inv_res, oz = P.oz;
  const float sg = side ? -1.f : 1.f;
  const int nb_side = side ? a.b_split : B - a.b_split;
  const int per = SUB ? max((nb_side + nsub - 1) / nsub, 2) : nb_side;   // (>= 2: a later run finds its first two tangents in the records of the two beams before it)
  const int first = SUB ? min(sub * per, nb_side) : 0, last = SUB ? min(first + per, nb_side) : nb_side;
  int ptr = side ? a.b_split - 1 - first : a.b_split + first;
  const int pstep = side ? -1 : 1, pend = side ? a.b_split - 1 - last : a.b_split + last;
  const bool none = ptr == pend;
  float s_stop = none ? 0.f : a.r_max;
  if (!none) {
    const float2 sc = a.beam_sc[side ? 0 : B - 1];
    const float dz_e = sc.x * P.c1[2] - sc.y * c2z;
    if (dz_e < -1e-4f) s_stop = fminf(s_stop, (a.zmin_map - oz) * fast_rcp(dz_e) * fabsf(sc.x));
  }
  s_stop += 2.f * res;
  // (no footprint test: the walk goes from triangle to triangle through the adjacency table and ends at the mesh's
  //  border -- at the OUTER border of a rectangular map for good, under the rule of sweep_side's second pass)
  if (!pre && !off_box) SWEEP_FAIL(1);
  u32 T = 0xffffffffu;
  const float r0 = off_box ? a.r_max : tin_nadir(a, I0, J0, ul, vl, oz, -P.c2[0], -P.c2[1], -c2z, T);
  // (HOLES: no triangle under the sensor may mean that its nadir ray goes through a linked hole, or past the linked
  //  outline -- the walk then starts at the rim, below)
  const bool in_gap = HOLES && T == 0xffffffffu && (ma.cell_rim != nullptr || ma.tin_outline != 0u);
  if ((!(r0 < a.r_max) || T == 0xffffffffu) && !in_gap) SWEEP_FAIL(4);
  if (none && !in_gap) return true;
  // plane and in-plane coordinates from (x - Ox, y - Oy, z - Oz) in metres
  const double Ox = ma.x0 + P.um * (double)ma.cs, Oy = ma.y0 + P.vm * (double)ma.cs;
  const float Oxf = (float)Ox, Oyf = (float)Oy, dOx = (float)(Ox - (double)Oxf), dOy = (float)(Oy - (double)Oyf);
  const float nx_ = P.c1[1] * P.c2[2] - P.c1[2] * P.c2[1], ny_ = P.c1[2] * P.c2[0] - P.c1[0] * P.c2[2],
              nz_ = P.c1[0] * P.c2[1] - P.c1[1] * P.c2[0];
  const float su = sg * P.c1[0], sv = sg * P.c1[1], sz = sg * P.c1[2];
  const float tu = -P.c2[0], tv = -P.c2[1], tz = -c2z;
  struct TinNode {
    float d, s, t;
  };
  // (member-wise: a select between whole structs goes through scratch)
  const auto sel = [](bool c, const TinNode& x, const TinNode& y) {
    TinNode r;
    r.d = c ? x.d : y.d;
    r.s = c ? x.s : y.s;
    r.t = c ? x.t : y.t;
    return r;
  };
  auto node_of = [&](const uint4 v) {
    // v - O with O = Of + dO split once per lane: the difference of two fp32 numbers a swath apart is exact (or off by
    // one ulp of a <= 100 m difference), the sub-ulp rest of the sensor position follows -- no fp64 per node
    const float rx = (__uint_as_float(v.x) - Oxf) - dOx, ry = (__uint_as_float(v.y) - Oyf) - dOy, rz = __uint_as_float(v.z) - oz;
    TinNode N;
    N.d = fmaf(nx_, rx, fmaf(ny_, ry, nz_ * rz));
    N.s = fmaf(su, rx, fmaf(sv, ry, sz * rz));
    N.t = fmaf(tu, rx, fmaf(tv, ry, tz * rz));
    return N;
  };
  // the half-edge table (mcl_mesh.h: MeshDev::tin_he) as a raw buffer: record h at byte 32 h; a border code
  // (h >= 0xfffffff0) is beyond num_records whatever the shift leaves of it and reads zeros -- no select, no 64-bit
  // address arithmetic
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ma.tin_he, 0, (int)ma.tin_he_bytes, 0x00020000);
  const auto he_xyzn = [&](u32 h) {   // {x, y, z of the vertex opposite half-edge h, next_a}
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (int)(h << 5), 0, 0));
  };
  const auto he_nb = [&](u32 h) {     // next_b
    return (u32)__builtin_amdgcn_raw_buffer_load_b32(hrsrc, (int)(h << 5) + 16, 0, 0);
  };
  // Along the rim of a linked hole or of the outline (mcl_halfedge.h): `cnt` edges from the one at place `pos` of the rim
  // records rbase .. rbase + rlen - 1, read by index (loads that do not wait for one another).  Of the edges the fan plane
  // cuts beyond s_min -- not the edge `skip` the slice came through, not an edge of the outline on the bounding box (word 3,
  // top bit: no way in through it) -- the nearest is kept in r: its rim record, its two ends (the interior half-edge runs
  // CA -> CB), the cut (cs, ct); r.cuts counts them.
  struct RimCut {
    u32 best, cuts;
    float cs;
  };
  const auto rim_edges = [&](RimCut& r, const u32 rbase, const u32 rlen, u32 pos, const u32 cnt, const float s_min, const bool strict, const u32 skip, const bool box_too) {
    pos -= pos >= rlen ? rlen : 0u;
    u32 cur = rbase + pos;
    uint4 qc = he_xyzn(cur);
    TinNode Nc = node_of(qc);
    for (u32 g = 0; g < cnt; ++g) {
      pos += 1u;
      pos -= pos >= rlen ? rlen : 0u;
      const u32 nxt = rbase + pos;
      const uint4 qn = he_xyzn(nxt);
      const TinNode Nn = node_of(qn);
      if ((__float_as_int(Nc.d) ^ __float_as_int(Nn.d)) < 0) {
        const float lam = Nc.d * fast_rcp(Nc.d - Nn.d);
        const float sx = fmaf(lam, Nn.s - Nc.s, Nc.s);
        const bool beyond = (strict ? sx > s_min : sx >= s_min) && cur != skip && (box_too || (int)qc.w >= 0);
        r.cuts += beyond ? 1u : 0u;
        const bool take = beyond && sx < r.cs;
        r.cs = take ? sx : r.cs;
        r.best = take ? cur : r.best;
      }
      cur = nxt;
      qc = qn;
      Nc = Nn;
    }
  };
  // ... the whole rim: edge by edge, or -- a long rim: chunk records from `cbase`, each the sphere around RIM_CHUNK consecutive
  // edges -- only the chunks whose sphere the plane cuts and that do not lie entirely before s_min
  const auto rim_cut = [&](const u32 rbase, const u32 rlen, const u32 cbase, const float s_min, const bool strict, const u32 skip, const bool box_too) {
    RimCut r;
    r.best = 0xffffffffu;
    r.cuts = 0u;
    r.cs = __builtin_inff();
    if (cbase == 0u) {
      rim_edges(r, rbase, rlen, 0u, rlen, s_min, strict, skip, box_too);
    } else {
      // (two levels: behind the rim's chunk records one record per RIM_CHUNK of them -- a scan of every chunk of a
      //  3 000-edge outline is 180 loads that wait for one another's branch: 1.3 ms per launch at 1 M particles)
      const u32 nch = (rlen + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK;
      const u32 nsup = (nch + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK;
      const auto cuts_sphere = [&](const u32 rec) {
        const uint4 q = he_xyzn(rec);
        const TinNode N = node_of(q);
        const float R = __uint_as_float(q.w);
        return fabsf(N.d) <= R && N.s + R >= s_min;
      };
      for (u32 g = 0; g < nsup; ++g) {
        if (!cuts_sphere(cbase + nch + g)) continue;
        const u32 c1 = min(g * halfedge::RIM_CHUNK + halfedge::RIM_CHUNK, nch);
        for (u32 c = g * halfedge::RIM_CHUNK; c < c1; ++c) {
          if (!cuts_sphere(cbase + c)) continue;
          const u32 p0 = c * halfedge::RIM_CHUNK;
          rim_edges(r, rbase, rlen, p0, min((u32)halfedge::RIM_CHUNK, rlen - p0), s_min, strict, skip, box_too);
        }
      }
    }
    return r;
  };
  // ... and the cut itself, once more for the edge that was kept (the search carries three registers, not ten, beside the
  // walk's state: the same expressions, the same bits): its ends -> (A, Bn), the cut -> (cs, ct)
  const auto rim_enter = [&](const u32 best, const u32 rbase, const u32 rlen, TinNode& CA, TinNode& CB, float& cs, float& ct) {
    u32 nx2 = best + 1u;
    nx2 = nx2 >= rbase + rlen ? rbase : nx2;
    CA = node_of(he_xyzn(best));
    CB = node_of(he_xyzn(nx2));
    const float lam = CA.d * fast_rcp(CA.d - CB.d);
    cs = fmaf(lam, CB.s - CA.s, CA.s);
    ct = fmaf(lam, CB.t - CA.t, CA.t);
  };
  TinNode A, Bn;
  u32 nb;    // the half-edge through which the slice enters the next triangle (or a border code)
  bool ao;   // is A the ORIGIN of that half-edge (in the next triangle's own counter-clockwise order)?
  float s_prev, t_prev, s_cur, t_cur;
  bool blind = false;   // (HOLES) a sensor beyond the outline whose fan plane meets no mesh on this side: every beam misses
  if (HOLES && in_gap) {
    // the nadir ray found no triangle: does it go through a linked hole?  The sensor's cell names the candidate (mcl_mesh.h:
    // cell_rim); the fan plane cuts its rim an even number of times, and the ray -- s = 0 -- runs between two cuts, through
    // the gap, exactly when an ODD number of them lies on this side's s > 0.  Then the nearest is where this side's slice
    // meets the mesh, and the beams up to its tangent look into the gap.  (Both sides of a particle test the same rim:
    // they agree, up to a cut at s = 0 to rounding -- one side then declines and the particle is handed over.)
    // No hole named for the cell (or a sensor beyond the bounding box) and the OUTLINE linked: is the sensor beyond it?  Then
    // an EVEN number of cuts lies on this side -- none: this side of the fan sees no seabed at all; else the nearest is where
    // its slice runs onto the mesh (through an edge on the bounding box as well: word 3's mark only bars the way BACK in).
    u32 rb = (!off_box && ma.cell_rim != nullptr) ? ma.cell_rim[(size_t)I0 * ma.gy + J0] : 0xffffffffu;
    const bool beyond = rb == 0xffffffffu;
    if (beyond) rb = ma.tin_outline;
    if (rb >= 0xfffffffeu || rb == 0u) SWEEP_FAIL(4);
    const uint4 rw = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (int)(rb << 5) + 16, 0, 0));   // {half-edge, first record, edges, chunks | outline}
    const RimCut rc = rim_cut(rb, rw.z, rw.w & 0x7fffffffu, 0.f, true, 0xffffffffu, beyond);
    if (((rc.cuts & 1u) != 0u) == beyond) SWEEP_FAIL(4);   // (inside the outline and no triangle: a hole that is not linked; outside the named hole)
    if (none) return true;
    blind = rc.cuts == 0u;
    s_cur = 0.f;
    t_cur = 1.f;
    A.d = A.s = A.t = 0.f;
    Bn = A;
    if (!blind) rim_enter(rc.best, rb, rw.z, A, Bn, s_cur, t_cur);
    if (!(t_cur > 0.f)) SWEEP_FAIL(4);
    ao = true;
    nb = blind ? 0xfffffff0u : he_nb(rc.best);
    s_prev = s_cur;   // (the walk's first segment is the point on the rim: the beams up to it are taken by the gap, below)
    t_prev = t_cur;
  } else {
    // triangle T through its three records: 3 T + e holds the vertex opposite edge e, v_e+2, and -- next_a -- the
    // half-edge on the far side of edge e + 2.  So vertex j comes from record (j + 1) % 3 and the far side of edge j
    // from the same record
    const uint4 q0 = he_xyzn(3u * T), q1 = he_xyzn(3u * T + 1u), q2 = he_xyzn(3u * T + 2u);
    const TinNode N0 = node_of(q1), N1 = node_of(q2), N2 = node_of(q0);
    const u32 f0 = q1.w, f1 = q2.w, f2 = q0.w;   // far side of edge 0 = (v0, v1), 1 = (v1, v2), 2 = (v2, v0)
    const bool p0b = __float_as_int(N0.d) >= 0, p1b = __float_as_int(N1.d) >= 0, p2b = __float_as_int(N2.d) >= 0;   // (sides of the plane by the sign bit, like the walk)
    if (p0b == p1b && p1b == p2b) SWEEP_FAIL(6);
    const int L = (p0b != p1b && p0b != p2b) ? 0 : ((p1b != p0b && p1b != p2b) ? 1 : 2);
    // local vertices L, L+1, L+2; the plane crosses edge L (vL, vL+1) and edge L+2 (vL+2, vL)
    const TinNode NL = sel(L == 0, N0, sel(L == 1, N1, N2));
    const TinNode NM = sel(L == 0, N1, sel(L == 1, N2, N0));  // vL+1
    const TinNode NN = sel(L == 0, N2, sel(L == 1, N0, N1));  // vL+2
    const u32 nbM = L == 0 ? f0 : (L == 1 ? f1 : f2);  // across edge L
    const u32 nbN = L == 0 ? f2 : (L == 1 ? f0 : f1);  // across edge L+2
    const float lm = NL.d * fast_rcp(NL.d - NM.d), ln = NL.d * fast_rcp(NL.d - NN.d);
    const float sm = fmaf(lm, NM.s - NL.s, NL.s), tm = fmaf(lm, NM.t - NL.t, NL.t);
    const float sn = fmaf(ln, NN.s - NL.s, NL.s), tn = fmaf(ln, NN.t - NL.t, NL.t);
    if (!(sm != sn)) SWEEP_FAIL(7);
    const bool far_m = sm > sn;
    const TinNode NF = sel(far_m, NM, NN);
    const bool pl = __float_as_int(NL.d) >= 0;
    A = sel(pl, NF, NL);   // (A, Bn: plane functions of opposite sign bits; along the walk A is the vertex found last)
    Bn = sel(pl, NL, NF);
    nb = far_m ? nbM : nbN;
    // the neighbour runs the shared edge the other way round (both counter-clockwise): out through edge L = (vL, vL+1)
    // its half-edge starts at vL+1 = NF, out through edge L+2 = (vL+2, vL) at vL = NL
    ao = far_m == pl;
    s_cur = far_m ? sm : sn;
    t_cur = far_m ? tm : tn;
    s_prev = far_m ? sn : sm;
    t_prev = far_m ? tn : tm;
    if (!(t_cur > 0.f)) SWEEP_FAIL(8);
  }
  float acc = 0.f;
  bool ok = true;
  const int max_steps = (int)(6.f * (s_stop + 4.f * res) * inv_res) + 64;  // (triangles may be much smaller than a cell)
  // (the table is walked by LDS byte address: the merge loop of the main kernel is sweep_merge_asm, as in sweep_side)
  // (the assembly loop walks the table through immediate offsets, which cannot be negative: every table address is
  //  kept low by the side's bias -- sweep_merge_asm -- and (bp - sb_off) >> 4 is still the beam)
  const unsigned sb_off = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)sbeam -
                          ((!EXPECT_ONLY && !SUB && !SWEEP_MERGE_CXX) ? (side ? SWEEP_BIAS1 : SWEEP_BIAS0) : 0u);
  unsigned bp = sb_off + (unsigned)(ptr * 16);
  const unsigned bp_end = sb_off + (unsigned)(pend * 16);
  const int msel = side + 2 * a.sweep_noclamp;
  const int pstep16 = pstep * 16;
  float tcur = stail[a.n_beams + side];
  if (SUB && first > 0) tcur = sbeam[ptr - pstep].x;
  sweep_rec bm;   // the pending beam's record
  {
    const float4 r = sbeam[ptr];
    bm.x = r.x;
    bm.y = r.y;
    bm.z = r.z;
    bm.w = r.w;
  }
  float gap_tan = -__builtin_inff();   // (HOLES) the largest tangent of a gap some beam of this lane looked into
  int gaps = 0;                        // (HOLES) gaps crossed on this side
  // the beams of the segment (dss, num, tp) -> (sc, tc): sweep_merge_asm, or -- expected ranges, runs of a side's beams, the
  // compiler-built variant -- the same loop in C++
  const auto merge = [&](const int sel_, const float dss, const float num, const float tp, const float sc, const float tc, const float dts) {
    if (!EXPECT_ONLY && !SUB && !SWEEP_MERGE_CXX) {
      sweep_merge_asm(sel_, acc, bp, dss, num, tp, sc, tc, dts);   // (sel = side + 2 noclamp: wave-uniform)
    } else {
      float e_cur = fmaf(-tcur, tc, sc);
      while (e_cur >= 0.f && (!SUB || bp != bp_end)) {
        const float tau = seg_tau(num, fmaf(-tcur, dts, dss), tp, tc);
        if (EXPECT_ONLY) {
          exp_row[(int)(bp - sb_off) >> 4] = fminf(tau * bm.y, a.r_max);
        } else {
          const float dd = hw_max(fmaf(-tau, bm.y, bm.z), bm.w);
          acc = fmaf(dd, dd, acc);
        }
        tcur = bm.x;
        bp += pstep16;
        const float4 r = sbeam[(int)(bp - sb_off) >> 4];
        bm.x = r.x;
        bm.y = r.y;
        bm.z = r.z;
        bm.w = r.w;
        e_cur = fmaf(-tcur, tc, sc);
      }
    }
  };
  // (HOLES) the beams that look into a gap whose far rim the slice meets at (xs, xt) -- tangents up to xs / xt, by the merge's
  // own test against that point scaled by a power of two (the next segment starts there: no beam between the two tests) --
  // hit nothing: each takes its clamp value (z - r_max) w, what the tail sums hold for the beams beyond the end of a walk.
  // By the merge itself, on a segment so far away that every crossing lies beyond r_max (always the statement WITH the
  // clamp).  Such a beam runs on UNDER the seabed beyond the hole: should the slice ever dip below it again -- a later vertex
  // at a smaller tangent -- it would come up against the seabed from below, which the merge cannot know: gap_tan
  // remembers the tangent, and the walk's t > 0 test carries the comparison.
  const auto gap_beams = [&](const float xs, const float xt) {
    const float K = 0x1p60f;
    const unsigned bp_in = bp;
    merge(msel & 1, 1.f, 1.f, xt * K, xs * K, xt * K, 0.f);
    gap_tan = bp != bp_in ? fmaxf(gap_tan, xs * fast_rcp(xt)) : gap_tan;
  };
  if (HOLES && in_gap && !blind) gap_beams(s_cur, t_cur);   // (the nadir ray goes through a gap: the beams from the nadir to the rim)
  // into the triangle behind half-edge nb, whose record is (hq, hb): the new vertex replaces the one on ITS side of the
  // plane (sides by the sign bit of the plane function) and always takes the role of A; the one that stays moves to Bn only
  // when it was A (three selects).  The slice leaves through the edge that joins the new vertex to the one that stays: the
  // entered half-edge runs a -> b, the new vertex N is opposite; a stays -> out through (N, a), whose far side is next_a
  // and ends in N = A; b stays -> out through (b, N), next_b, which starts in N = A.  The cut of that edge -> (sp, tp).
  // Returns true when the walk is over (not for the sweep).
  const auto cross = [&](const uint4 hq, const u32 hb, float& sp, float& tp) -> bool {
    const float rx = (__uint_as_float(hq.x) - Oxf) - dOx, ry = (__uint_as_float(hq.y) - Oyf) - dOy, rz = __uint_as_float(hq.z) - oz;
    const float dN = fmaf(nx_, rx, fmaf(ny_, ry, nz_ * rz));
    const bool keep_a = (__float_as_int(dN) ^ __float_as_int(A.d)) < 0;
    const bool stays_a = keep_a == ao;   // the vertex that stays is the half-edge's origin
    nb = stays_a ? hq.w : hb;
    ao = !stays_a;
    Bn.d = keep_a ? A.d : Bn.d;
    Bn.s = keep_a ? A.s : Bn.s;
    Bn.t = keep_a ? A.t : Bn.t;
    A.d = dN;
    A.s = fmaf(su, rx, fmaf(sv, ry, sz * rz));
    A.t = fmaf(tu, rx, fmaf(tv, ry, tz * rz));
    const float lam = A.d * fast_rcp(A.d - Bn.d);
    sp = fmaf(lam, Bn.s - A.s, A.s);
    tp = fmaf(lam, Bn.t - A.t, A.t);
    // (HOLES: ... and not at a smaller tangent than a gap a beam looked into -- gap_tan = -inf until then: the fma is
    //  + inf for t > 0 and NaN or - inf otherwise, and v_min returns the number)
    if (!((HOLES ? fminf(tp, fmaf(-gap_tan, tp, sp)) : tp) > 0.f)) {
      SWEEP_NOTE(tp > 0.f ? 15 : 12);
      ok = false;
      return true;
    }
    return false;
  };
  // one step of the walk: resolve the beams of the segment (sp, tp) -> (sc, tc), then cross into the neighbour.  The new
  // vertex of the slice takes the place of the one before last and the CALLER swaps the roles (the loop is unrolled by
  // two).  Returns true when the walk is over.  EXITS: as in sweep_side, the tests that end a walk normally run every
  // second step; the step count is the wave's.
  const auto walk_step = [&](float& sp, float& tp, float& sc, float& tc, const int step, auto EXITS) -> bool {
    // the entered half-edge's record is in flight while the beams are resolved: the vertex the slice meets next and
    // the two half-edges it can leave through -- ONE dependent load per step
    const uint4 hq = he_xyzn(nb);
    const u32 hb = he_nb(nb);
    const float dts = tc - tp;
    const float dss = sc - sp, num = fmaf(tp, dss, -(sp * dts));   // (the segment's constants of seg_tau)
    merge(msel, dss, num, tp, sc, tc, dts);
    if (decltype(EXITS)::value) {
      if (bp == bp_end) return true;
      if (sc > s_stop) return true;
    }
    if (HOLES ? nb >= ma.tin_nhe : nb >= 0xfffffff0u) {
      if (!HOLES || nb >= 0xfffffff0u) {
        // the slice runs off the mesh.  Through the map's outer border: final if it cannot come back (same bound as in
        // sweep_side's second pass); through a ragged outline or a hole without rim records: not for the sweep
        const float sb = fast_sqrt(fmaxf(1.f - c2z * c2z, 0.f));
        const float lhs = sb * (a.sweep_slope + sb), rhs = 0.9f * (c2z - a.sweep_slope * sb);
        ok = (nb != 0xffffffffu) & (lhs < rhs * fabsf(nb == 0xfffffff0u ? P.c1[0] : P.c1[1]));
        if (!ok) SWEEP_NOTE(nb == 0xffffffffu ? 13 : 10);
        return true;  // (ok: the beams left get r_max through the tail below)
      }
      // A LINKED RIM (mcl_halfedge.h: link_holes -- a hole's, or the outline's): nb names the rim record of the edge the slice
      // has just reached.  Along the rim: of the edges the fan plane cuts, the nearest one further out is where the slice
      // meets the mesh again (nothing lies in the space a linked rim bounds).
      const u32 k0 = nb;
      const uint4 rw = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (int)(k0 << 5) + 16, 0, 0));   // {half-edge, first record, edges, chunks | outline}
      const RimCut rc = rim_cut(rw.y, rw.z, rw.w & 0x7fffffffu, sc, false, k0, false);   // (every edge of the rim but the one reached)
      const u32 best = rc.best;
      if (best == 0xffffffffu && (int)rw.w < 0) return true;   // beyond the OUTLINE and no way back in: nothing lies further out -- the beams left get r_max through the tail below
      // (a slice through a rim vertex can find the two cuts there in either order and go back and forth between the gap and
      //  a sliver: a side crosses SWEEP_TIN_GAPS gaps at most)
      if (best == 0xffffffffu || ++gaps > SWEEP_TIN_GAPS) {
        SWEEP_NOTE(14);
        ok = false;
        return true;
      }
      // on from the far rim: in through the interior half-edge of that edge, which runs from its origin (-> A) to its end (-> Bn)
      rim_enter(best, rw.y, rw.z, A, Bn, sc, tc);
      if (!(tc > 0.f)) {
        SWEEP_NOTE(14);
        ok = false;
        return true;
      }
      gap_beams(sc, tc);   // the beams that look into the gap miss
      if (decltype(EXITS)::value && bp == bp_end) return true;
      ao = true;
      // (the step ends HERE, through this path's own copy of the crossing: state that changed in this branch alone and
      //  went on into the common code cost the loop eight register copies on every step -- 14 % of the kernel)
      nb = he_nb(best);
      if (decltype(EXITS)::value && step > max_steps) {
        SWEEP_NOTE(11);
        ok = false;
        return true;
      }
      return cross(he_xyzn(nb), he_nb(nb), sp, tp);
    }
    if (decltype(EXITS)::value && step > max_steps) {
      SWEEP_NOTE(11);
      ok = false;
      return true;
    }
    return cross(hq, hb, sp, tp);
  };
  if (!(HOLES && blind)) {
    for (int step = 1;; step += 2) {
      if (walk_step(s_prev, t_prev, s_cur, t_cur, step, std::true_type())) break;
      if (walk_step(s_cur, t_cur, s_prev, t_prev, step + 1, std::false_type())) break;
    }
  }
  if (ok && bp != bp_end) {
    ptr = (int)(bp - sb_off) >> 4;
    if (EXPECT_ONLY) {
      for (; ptr != pend; ptr += pstep) exp_row[ptr] = a.r_max;
    } else {
      acc += stail[ptr];
    }
  }
  acc_out = acc;
  return ok;
}

// Work layout: a WAVE holds 64 particles' lanes of ONE side (and one run of its beams) -- with `nsub` runs per side,
// waves 2 nsub g ... 2 nsub g + 2 nsub - 1 of a workgroup are the (side, run) combinations of the same 64 particles.
// The lanes of a wave then walk alike whenever the cloud is coherent, and the beam table is read by broadcast (with the
// two sides of a particle in neighbouring lanes, round 2, every ds_read hit two distant records: 2.2e7 bank-conflict
// cycles per launch, now 3e5).  The lanes of a particle meet through LDS.
#define SWEEP_MAX_WAVES 8   // SUB kernels: up to 4 runs per side x 2 sides

// one (particle, side, run): cast; the (+ side, run 0) lane then combines the verdicts and sums in a fixed order, writes
// lw or hands the particle over.  j0: the workgroup's first particle.
template <int SURF, bool EXPECT_ONLY, bool SUB>
__device__ __forceinline__ double sweep_lane(const MbesArgs& a, long long j0, long long n, const float4* sbeam,
                                             const float* stail, float* xacc, int* xok) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: scalar registers)
  const int nsub = SUB ? a.sweep_nsub : 1, combos = 2 * nsub;
  const int group = w / combos, combo = w - group * combos;
  const int side = combo & 1, sub = combo >> 1, pl = group * 64 + lane;   // particle within the workgroup
  // (expected ranges: the grid only covers the particles asked for)
  const long long i = j0 + pl + (EXPECT_ONLY ? a.exp_first : 0);
  const bool valid = i < n;
  bool ok = true, work = valid;
  float* exp_row = nullptr;
  if (EXPECT_ONLY) {
    work = valid && i < a.exp_first + a.exp_count;
    if (work) exp_row = a.exp_out + (size_t)(i - a.exp_first) * a.n_beams;
  }
  float acc = 0.f;
  u32 slot = (u32)i;   // where the log-likelihood goes: the record's state slot (records may lie in visiting order)
  const float4* sside = sbeam + (side ? 0 : 2);   // (this side's records by beam index: k_mbes_sweep's table layout)
  if (work) {
    const MbesPose P = a.pose[i];
    slot = EXPECT_ONLY ? (u32)i : P.slot;
    if (SURF == 5 || SURF == 6)   // (6: a TIN with rim records -- holes the walk crosses)
      ok = sweep_side_tin<EXPECT_ONLY, SUB, SURF == 6>(a, P, sside, stail, side, sub, nsub, exp_row, acc);
    else if (SURF == 0)
      ok = sweep_side_grid<EXPECT_ONLY, SUB>(a, P, sside, stail, side, sub, nsub, exp_row, acc);
    else
      ok = sweep_side<(SURF == 3 ? 3 : 2), EXPECT_ONLY, SUB>(a, P, sside, stail, side, sub, nsub, exp_row, acc);
  }
  // the lanes of a particle agree on its fate: every lane but the first leaves its verdict and sum in LDS
  if (combo) {
    xacc[w * 64 + lane] = acc;
    xok[w * 64 + lane] = ok ? 1 : 0;
  }
#ifdef SWEEP_TIMELINE
  if (lane == 0 && blockIdx.x * (blockDim.x >> 6) + w < SWEEP_TL_WAVES)
    g_sweep_tl[6 * (blockIdx.x * (blockDim.x >> 6) + w) + 1] = wall_clock64();
#endif
  __syncthreads();
  double v = -__builtin_inf();
  if (!combo) {
    bool ok2 = ok;
    double acc2 = (double)acc;
    for (int c = 1; c < combos; ++c) {   // fixed order: the sum does not depend on which wave finished first
      ok2 = ok2 & (xok[(w + c) * 64 + lane] != 0);
      acc2 += (double)xacc[(w + c) * 64 + lane];
    }
    if (work && ok2 && !EXPECT_ONLY) {
      v = -0.5 * acc2 - (double)a.sweep_nvalid * a.lognorm;
      a.lw[slot] = v;
    }
    // hand-overs: one atomic per wave.  (The list's order is the waves' finishing order: k_mbes_cast<., ., 2> casts
    // every entry with arithmetic that depends on the particle alone -- mcl_mbes.h, the determinism rule.)
    const unsigned long long dm = __ballot(work && !ok2);
    if (dm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(a.defer_count, (int)__popcll(dm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (work && !ok2) a.defer_idx[base + (int)__popcll(dm & ((1ull << lane) - 1ull))] = (u32)i;
    }
  }
  return v == v ? v : -__builtin_inf();  // NaN never wins the maximum
}

// (register budgets: the lattice walk 8 waves / SIMD (64 VGPRs), grids and TINs 6; the sub-fan kernel over a grid 5 -- it
//  carries the conic AND the start ray's footprint test, and spilled 8 B per lane at 6: small clouds, latency-bound anyway)
template <int SURF, bool EXPECT_ONLY, bool SUB = false>
__global__ void __launch_bounds__(SUB ? 64 * SWEEP_MAX_WAVES : SWEEP_THREADS, SURF == 0 ? (SUB ? SWEEP_MIN_WAVES_GRID - 1 : SWEEP_MIN_WAVES_GRID) : (SURF == 6 && !SUB && !EXPECT_ONLY ? 8 : (SURF == 5 || SURF == 6 ? SWEEP_MIN_WAVES_TIN : SWEEP_MIN_WAVES))) k_mbes_sweep(MbesArgs a) {
#ifdef SWEEP_TIMELINE
  const unsigned long long tl0 = wall_clock64(), tc0 = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char sweep_lds[];
  __shared__ float xacc[64 * (SUB ? SWEEP_MAX_WAVES : SWEEP_THREADS / 64)];
  __shared__ int xok[64 * (SUB ? SWEEP_MAX_WAVES : SWEEP_THREADS / 64)];
  // the table in LDS: four spare records | -1 (sentinel) | side 1's beams 0 .. b_split - 1 | a record in front of either
  // side's first beam (its .x: that beam's tangent, read by sweep_merge_asm) | side 0's beams | sentinel | tail sums.
  // Side 1 indexes it by beam from sbeam, side 0 from sbeam + 2 (sweep_lane); the merge statement reads up to three
  // records beyond either end (never used)
  float4* sbeam = (float4*)sweep_lds + 4;
  float* stail = (float*)(sbeam + a.n_beams + 3);
  for (int b = threadIdx.x; b < a.n_beams; b += blockDim.x) {
    sbeam[b + (b >= a.b_split ? 2 : 0)] = a.sweep_beams[b];
    stail[b] = SUB ? a.sweep_tail_run[b] : a.sweep_tail[b];   // (SUB: tail sums that end with the lane's own run)
  }
  if (threadIdx.x < 4) stail[a.n_beams + threadIdx.x] = a.sweep_tan0[threadIdx.x];   // (first / second tangent of either side)
  if (threadIdx.x == 0) sbeam[-1] = sbeam[a.n_beams + 2] = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);  // "never reached"
  if (threadIdx.x == 1) sbeam[a.b_split] = make_float4(a.sweep_tan0[1], 0.f, 0.f, 0.f);
  if (threadIdx.x == 2) sbeam[a.b_split + 1] = make_float4(a.sweep_tan0[0], 0.f, 0.f, 0.f);
  __syncthreads();
  // particles per workgroup: its waves divided by the (side, run) combinations of a particle; one lane per
  // (particle, side, run)
  const int per_block = (int)(blockDim.x >> 6) / (2 * (SUB ? a.sweep_nsub : 1)) * 64;
  const double vmax = sweep_lane<SURF, EXPECT_ONLY, SUB>(a, blockIdx.x * (long long)per_block, a.n, sbeam, stail, xacc, xok);
  if (!EXPECT_ONLY && a.max_slots) {
    // the normalisation needs max lw: one atomic per wave that wrote log-likelihoods, on an order-preserving key
    const double m = wave_max(vmax);
    if ((threadIdx.x & 63) == 0 && m > -__builtin_inf())
      atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (MCL_MAX_SLOTS - 1)],
                ordered_key(m));
  }
#ifdef SWEEP_TIMELINE
  {
    const unsigned wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && wv < SWEEP_TL_WAVES) {
      g_sweep_tl[6 * wv] = tl0;
      g_sweep_tl[6 * wv + 2] = wall_clock64();
      g_sweep_tl[6 * wv + 4] = tc0;                           // shader clock (s_memtime): the clock the launch ran at
      g_sweep_tl[6 * wv + 5] = __builtin_readcyclecounter();
      g_sweep_tl[6 * wv + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4)             // HW_ID
                               | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // XCC_ID
    }
  }
#endif
}
