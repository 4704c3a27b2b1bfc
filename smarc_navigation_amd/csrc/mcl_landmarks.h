// mcl_landmarks.h -- landmark measurement update with k-nearest-neighbour data association
// (BASELINE config 5; no reference symbol in auv_particle_filter).  Nearest reference analogues:
// landmark -> sensor-frame model auv_ekf_slam/src/correspondence_obj_mbes.cpp:26-35, chi-square
// gate auv_ekf_slam/src/ekf_slam.cpp:100-103, max-likelihood nearest landmark
// auv_ekf_localization/src/ekf_localization.cpp:479-524, Gaussian likelihood
// auv_ekf_localization/src/correspondence_obj.cpp:80-97.
//
// Definition (same in oracle/mcl_oracle.c:orc_landmark_update): a detection z_d (sensor frame) is
// mapped into the map frame with the particle's sensor pose; maha_j = |p_d - l_j|^2 / sigma^2; over
// the k nearest landmarks with maha_j <= gate:  lw_d = log sum_j exp(-maha_j/2), or -gate/2 if none;
// lw = sum_d lw_d - D (3/2 log 2pi + 3 log sigma).
// Landmarks are binned in a uniform xy grid with cell >= gate radius, so the 3x3 neighbourhood
// holds every landmark inside the gate: exact, ~9 cell probes per (particle, detection).
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mcl.h"
#include "mcl_device.h"

#define LM_MAX_K 4
#define LM_SUB 16  // lanes per particle (one detection per lane)

struct LandmarkArgs {
  const double* st[6];
  long long n;
  double m2o[12];
  double off_t[3];
  double off_R[9];
  const double* det;      // D x 3 (sensor frame)
  int n_det;
  const double* lm;       // landmarks in CELL order, 3 doubles each
  const u32* cell_start;  // gx*gy + 1
  // the 3 x 3 neighbourhood of every query cell, flattened (k_landmark_update): nb_cell[(cx + 2) * (gy + 4) + cy + 2] =
  // {first entry, count}; an entry = {x, y, z, slot (bits of the 4th double)} in the order the three row ranges are visited
  const uint2* nb_cell;
  const double4* nb_list;
  int gx, gy;
  double x0, y0, inv_cs;
  double inv_s2, gate, lognorm;  // lognorm = 3/2 log(2 pi) + 3 log(sigma)  (Mahalanobis mode: + 1/2 log det Q)
  // Mahalanobis mode (mcl_set_landmark_noise): per-landmark position covariance (cell order, 6 doubles: xx xy xz
  // yy yz zz, or nullptr = none) and the sensor-frame measurement covariance Q
  int maha;
  const double* lmcov;
  double Q[6];
  double logdet_q;               // log det Q
  int k;
  int accumulate;         // add to lw instead of overwriting
  double* lw;
  // fused step (mcl_step_mbes_landmarks): the predict kernel of the same call has not stored z, roll, pitch (they are
  // the odometry's on every particle: bit c of uni_mask set = component c is uni[c - 2]); the maximum of the
  // accumulated lw goes to max_slots (ordered keys, MCL_MAX_SLOTS words) so that no k_max_slots pass follows
  unsigned uni_mask;
  double uni[3];
  u64* max_slots;
};

// Cost of pairing a detection (already mapped to p = o + R z in the map frame) with landmark slot e, and
// log det of its innovation covariance.
//   isotropic:   |p - l|^2 / sigma^2,  S = sigma^2 I
//   Mahalanobis: the reference's d_m = nu^T S^-1 nu with nu = z - R^T (l - o) in the SENSOR frame and
//                S = H Sigma H^T + Q (auv_ekf_slam/src/ekf_slam_core.cpp:160-162, correspondence_obj_mbes.cpp:
//                26-35,110-120).  A particle is a pose hypothesis, so the robot block of the EKF covariance is
//                gone and H is the landmark block R^T: S = R^T Sigma_j R + Q.  Evaluated in the map frame, where
//                it is the same number:  (p - l)^T (Sigma_j + R Q R^T)^-1 (p - l).   Qm = R Q R^T per particle.
__device__ __forceinline__ void landmark_qm(const LandmarkArgs& a, const double Rs[9], double Qm[6]) {
  // T = R Q (Q symmetric: xx xy xz yy yz zz), Qm = T R^T
  const double q[9] = {a.Q[0], a.Q[1], a.Q[2], a.Q[1], a.Q[3], a.Q[4], a.Q[2], a.Q[4], a.Q[5]};
  double T[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = Rs[r * 3] * q[c] + Rs[r * 3 + 1] * q[3 + c] + Rs[r * 3 + 2] * q[6 + c];
  const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int r = ij[k][0], c = ij[k][1];
    Qm[k] = T[r * 3] * Rs[c * 3] + T[r * 3 + 1] * Rs[c * 3 + 1] + T[r * 3 + 2] * Rs[c * 3 + 2];
  }
}
template <bool MAHA>
__device__ __forceinline__ double landmark_pair_cost_d(const LandmarkArgs& a, const double Qm[6], double dx, double dy,
                                                       double dz, u32 e, double* logdet) {
  if (!MAHA) {
    *logdet = 0.0;  // constant: folded into lognorm
    return (dx * dx + dy * dy + dz * dz) * a.inv_s2;
  }
  double s[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) s[k] = Qm[k] + (a.lmcov ? a.lmcov[6 * (size_t)e + k] : 0.0);
  // symmetric 3x3 inverse by cofactors
  const double c00 = s[3] * s[5] - s[4] * s[4], c01 = s[2] * s[4] - s[1] * s[5], c02 = s[1] * s[4] - s[2] * s[3];
  const double det = s[0] * c00 + s[1] * c01 + s[2] * c02;
  const double c11 = s[0] * s[5] - s[2] * s[2], c12 = s[1] * s[2] - s[0] * s[4], c22 = s[0] * s[3] - s[1] * s[1];
  const double quad = dx * (c00 * dx + c01 * dy + c02 * dz) + dy * (c01 * dx + c11 * dy + c12 * dz) +
                      dz * (c02 * dx + c12 * dy + c22 * dz);
  *logdet = log(det) - a.logdet_q;  // relative to Q's: lognorm carries 1/2 log det Q
  return quad / det;
}

template <bool MAHA>
__device__ __forceinline__ double landmark_pair_cost(const LandmarkArgs& a, const double Qm[6], double px, double py,
                                                     double pz, u32 e, double* logdet) {
  const double dx = px - a.lm[3 * (size_t)e], dy = py - a.lm[3 * (size_t)e + 1], dz = pz - a.lm[3 * (size_t)e + 2];
  return landmark_pair_cost_d<MAHA>(a, Qm, dx, dy, dz, e, logdet);
}
__device__ __forceinline__ double landmark_pair_cost(const LandmarkArgs& a, const double Qm[6], double px, double py,
                                                     double pz, u32 e, double* logdet) {
  return a.maha ? landmark_pair_cost<true>(a, Qm, px, py, pz, e, logdet) : landmark_pair_cost<false>(a, Qm, px, py, pz, e, logdet);
}

// sensor pose of particle i in the map frame (fp64): M = m2o * T(x,y,z) R(rpy) * T_off R_off
__device__ __forceinline__ void landmark_sensor_pose(const LandmarkArgs& a, long long i, double Rs[9], double o[3]) {
  const bool uni = a.uni_mask == 0x1cu;   // (all three or none: do_predict's skip_uniform)
  const double x = a.st[0][i], y = a.st[1][i], z = uni ? a.uni[0] : a.st[2][i];
  double sr, cr, sp, cp, sy, cy;
  sincos(uni ? a.uni[1] : a.st[3][i], &sr, &cr);
  sincos(uni ? a.uni[2] : a.st[4][i], &sp, &cp);
  sincos(a.st[5][i], &sy, &cy);
  const double Rp[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                        -sp,     cp * sr,                cp * cr};
  double Rmp[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rmp[r * 3 + c] = a.m2o[r * 4 + 0] * Rp[c] + a.m2o[r * 4 + 1] * Rp[3 + c] + a.m2o[r * 4 + 2] * Rp[6 + c];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rs[r * 3 + c] = Rmp[r * 3 + 0] * a.off_R[c] + Rmp[r * 3 + 1] * a.off_R[3 + c] + Rmp[r * 3 + 2] * a.off_R[6 + c];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = (a.m2o[r * 4 + 0] * x + a.m2o[r * 4 + 1] * y + a.m2o[r * 4 + 2] * z + a.m2o[r * 4 + 3]) +
           (Rmp[r * 3 + 0] * a.off_t[0] + Rmp[r * 3 + 1] * a.off_t[1] + Rmp[r * 3 + 2] * a.off_t[2]);
}

// Work layout (round 5): a lane owns a particle and walks the detections of the ping one after the other; the pose
// (three fp64 sincos, two 3x3 products, R Q R^T in Mahalanobis mode) stays in registers.  The lanes of a wave ask
// about the SAME detection from neighbouring poses, so in a converged or spatially ordered cloud they read the same
// table entry and the same landmarks (one cache line per wave instead of sixteen), and the detection is a wave-
// uniform operand.  (Rounds 3-4: 4 particles x 16 detections per wave pass, poses handed over through LDS, a
// butterfly sum per particle: 2.4 x the instructions, lane utilisation 0.65, every lane on its own cache lines.)
// One table load gives the flattened 3 x 3 neighbourhood of the query cell (rounds 1-4: six cell_start loads and
// three loops); the entry of the NEXT detection is requested before the landmarks of this one are visited, so a
// detection costs one memory round trip, not two.  The sum over the detections runs in their order, like the oracle's.
template <bool MAHA>
__global__ void __launch_bounds__(256) k_landmark_update(LandmarkArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ey = a.gy + 4;
  const double cx_hi = (double)(a.gx + 1), cy_hi = (double)(a.gy + 1);
  // the detections through the scalar cache (constant address space: written by a copy before this launch)
  typedef const double __attribute__((address_space(4))) * cdp;
  const cdp det = (cdp)(unsigned long long)a.det;
  double wmax = -__builtin_inf();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    double Rs[9], o[3], Qm[6] = {0, 0, 0, 0, 0, 0};
    landmark_sensor_pose(a, i, Rs, o);
    if (MAHA) landmark_qm(a, Rs, Qm);
    // detection d seen from this pose, and the table entry of its cell (cells beyond the one-cell border of the grid
    // are clamped onto the empty outer ring of the table; NaN coordinates land there too)
    auto query = [&](int d, double& px, double& py, double& pz, uint2& ent, bool& ok) {
      const double zx = det[3 * d], zy = det[3 * d + 1], zz = det[3 * d + 2];   // (scalar loads: d is wave-uniform)
      ok = zx == zx && zy == zy && zz == zz;   // NaN = invalid detection
      px = o[0] + Rs[0] * zx + Rs[1] * zy + Rs[2] * zz;
      py = o[1] + Rs[3] * zx + Rs[4] * zy + Rs[5] * zz;
      pz = o[2] + Rs[6] * zx + Rs[7] * zy + Rs[8] * zz;
      const double fx = fmin(fmax(floor((px - a.x0) * a.inv_cs), -2.0), cx_hi);
      const double fy = fmin(fmax(floor((py - a.y0) * a.inv_cs), -2.0), cy_hi);
      ent = a.nb_cell[(size_t)((int)fx + 2) * ey + ((int)fy + 2)];
    };
    double acc = 0.0;
    int nvalid = 0;
    double px, py, pz;
    uint2 ent;
    bool ok;
    query(0, px, py, pz, ent, ok);
    for (int d = 0; d < a.n_det; ++d) {
      const double cpx = px, cpy = py, cpz = pz;
      const uint2 cur = ent;
      const bool cok = ok;
      if (d + 1 < a.n_det) query(d + 1, px, py, pz, ent, ok);
      if (!cok) continue;   // (wave-uniform)
      double best[LM_MAX_K], bld[LM_MAX_K];
#pragma unroll
      for (int q = 0; q < LM_MAX_K; ++q) {
        best[q] = __builtin_inf();
        bld[q] = 0.0;
      }
      for (u32 e = cur.x; e < cur.x + cur.y; ++e) {
        const double4 l = a.nb_list[e];
        double ld;
        double m = landmark_pair_cost_d<MAHA>(a, Qm, cpx - l.x, cpy - l.y, cpz - l.z, (u32)__double_as_longlong(l.w), &ld);
        if (m <= a.gate) {
          // insert into the sorted k-best list
#pragma unroll
          for (int q = 0; q < LM_MAX_K; ++q) {
            if (m < best[q]) {
              const double t = best[q], tl = bld[q];
              best[q] = m;
              bld[q] = ld;
              m = t;
              ld = tl;
            }
          }
        }
      }
      double lwd;
      if (best[0] == __builtin_inf()) {
        lwd = -0.5 * a.gate;
      } else {
        // log sum_k exp(-d_k^2 / 2) / sqrt(det S_k / det Q) over the k nearest inside the gate, anchored at the nearest
        const double e0 = -0.5 * (best[0] + bld[0]);
        lwd = e0;   // (one neighbour in the gate: exp(0) = 1, log(1) = 0 -- the same number without the calls)
        if (a.k > 1 && best[1] != __builtin_inf()) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < LM_MAX_K; ++q)
            if (q < a.k && best[q] != __builtin_inf()) s += exp(-0.5 * (best[q] + bld[q]) - e0);
          lwd = e0 + log(s);
        }
      }
      acc += lwd;
      ++nvalid;
    }
    double v = acc - (double)nvalid * a.lognorm;
    if (a.accumulate) v += a.lw[i];
    a.lw[i] = v;
    wmax = (v > wmax) ? v : wmax;   // NaN never wins (k_max_slots)
  }
  if (a.max_slots) {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, o2, 64));
    if (lane == 0 && wmax > -__builtin_inf())
      atomicMax((unsigned long long*)&a.max_slots[(blockIdx.x * 4 + w) & (MCL_MAX_SLOTS - 1)], ordered_key(wmax));
  }
}


// ------------------------------------------------------------------ global (Hungarian) assignment
// SURVEY 8(f) rank 4.  Per particle the correspondence table of the reference's batch association
// (auv_ekf_slam/src/ekf_slam_core.cpp:172-178: Mahalanobis distance if < gate else 10000; :269-281:
// one new-landmark row per detection at cost new_mh_dist; :298-312: Munkres) is solved exactly:
// lw = -1/2 (optimal total) - D_valid lognorm.  Every detection owns a private new-landmark column,
// so the optimum never uses a 10000 entry and the table reduces to the gated pairs: a sparse
// bipartite graph with <= k_cand + 1 edges per detection (k_cand <= 8 nearest gated landmarks).
//   phase 1 (16 lanes per particle, one detection each): gated candidates from the landmark cell
//            grid, sorted by distance;
//   phase 2 (same lanes): shared landmarks get one column slot (first occurrence);
//   phase 3 (lane 0 of the group): shortest-augmenting-path assignment with potentials on the sparse
//            graph -- typically one relaxation per detection, conflicts cost a few more;
//   phase 4 (16 lanes): each detection reads its matched edge, 16-lane sum.
// Oracle: dense table over ALL landmarks + dense solver (orc_landmark_assign_update), the solver
// itself pinned to the reference's Munkres.
#define LA_KC 8
#define LA_PER_BLOCK 8                    // particles per 128-thread block
#define LA_COLS (LM_SUB + LM_SUB * LA_KC) // 16 private columns + 128 first-occurrence slots
#define LA_START LA_COLS                  // virtual start column

struct LandmarkAssignArgs {
  LandmarkArgs base;
  const u32* orig;       // cell-ordered slot -> caller's landmark index
  double new_mh;
  int k_cand;
  int* assign_out;       // optional: first n_keep particles x n_det (landmark index, -1 new, -2 invalid)
  long long n_keep;
  int* worklist;         // particles whose cheapest edges clash (filled by the fast kernel)
  int* work_count;
};

struct LaSharedFast {   // 560 B per particle: the fast kernel only exchanges candidate ids and choices
  u32 cand_id[LM_SUB][LA_KC];
  unsigned char ncand[LM_SUB];   // 255 = invalid detection
  unsigned char choice[LM_SUB];
};

struct LaSharedFull {
  double cand_cost[LM_SUB][LA_KC];
  double v[LA_COLS + 1];
  double minv[LA_COLS + 1];
  double u[LM_SUB];
  u32 cand_id[LM_SUB][LA_KC];
  unsigned char cand_col[LM_SUB][LA_KC];
  unsigned char ncand[LM_SUB];
  unsigned char p[LA_COLS + 1];     // row + 1 matched to the column, 0 = free
  unsigned char way[LA_COLS + 1];
  unsigned char flag[LA_COLS + 1];  // bit 0 touched, bit 1 used
  unsigned char list[LA_COLS + 1];  // touched columns of the current search
};

// FULL = false: phases 1, 2, 2b, 4 -- every particle; a particle whose detections' cheapest edges clash
//               is appended to the worklist instead of being answered.
// FULL = true : the worklist only, with the augmenting-path solver (phase 3).
template <bool FULL>
__global__ void __launch_bounds__(LA_PER_BLOCK * LM_SUB) k_landmark_assign(LandmarkAssignArgs aa) {
  using Shared = typename std::conditional<FULL, LaSharedFull, LaSharedFast>::type;
  __shared__ Shared sh[LA_PER_BLOCK];
  const LandmarkArgs& a = aa.base;
  const int sub = threadIdx.x & (LM_SUB - 1);
  const int g = threadIdx.x / LM_SUB;
  Shared& S = sh[g];
  const double INF = __builtin_inf();
  const long long total = FULL ? (long long)*aa.work_count : a.n;
  const long long nblk = (total + LA_PER_BLOCK - 1) / LA_PER_BLOCK;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // block-uniform trip count
    const long long slot = blk * LA_PER_BLOCK + g;
    const bool live = slot < total;
    const long long i = live ? (FULL ? (long long)aa.worklist[slot] : slot) : 0;
    // ---- phase 1: candidates of detection `sub`, nearest first
    double best[LA_KC];
    u32 bid[LA_KC];
#pragma unroll
    for (int q = 0; q < LA_KC; ++q) {
      best[q] = INF;
      bid[q] = 0xffffffffu;
    }
    bool valid = false;
    if (live && sub < a.n_det) {
      const double zx = a.det[3 * sub], zy = a.det[3 * sub + 1], zz = a.det[3 * sub + 2];
      valid = zx == zx && zy == zy && zz == zz;
      if (valid) {
        double Rs[9], o[3];
        landmark_sensor_pose(a, i, Rs, o);
        const double px = o[0] + Rs[0] * zx + Rs[1] * zy + Rs[2] * zz;
        const double py = o[1] + Rs[3] * zx + Rs[4] * zy + Rs[5] * zz;
        const double pz = o[2] + Rs[6] * zx + Rs[7] * zy + Rs[8] * zz;
        double Qm[6] = {0, 0, 0, 0, 0, 0};
        if (a.maha) landmark_qm(a, Rs, Qm);
        const int cx = (int)floor((px - a.x0) * a.inv_cs), cyi = (int)floor((py - a.y0) * a.inv_cs);
        // (the three cells of a grid row are neighbours in the cell-ordered landmark array: one range per row)
        const int iy0 = max(cyi - 1, 0), iy1 = min(cyi + 1, a.gy - 1);
        for (int ix = max(cx - 1, 0); ix <= min(cx + 1, a.gx - 1) && iy0 <= iy1; ++ix)
          {
            const size_t c = (size_t)ix * a.gy;
            for (u32 e = a.cell_start[c + iy0], e1 = a.cell_start[c + iy1 + 1]; e < e1; ++e) {
              double ld_unused;
              double m = landmark_pair_cost(a, Qm, px, py, pz, e, &ld_unused);
              if (m < a.gate) {  // strict, ekf_slam_core.cpp:173
                u32 id = e;
#pragma unroll
                for (int q = 0; q < LA_KC; ++q) {
                  if (m < best[q]) {
                    const double t = best[q];
                    const u32 ti = bid[q];
                    best[q] = m;
                    bid[q] = id;
                    m = t;
                    id = ti;
                  }
                }
              }
            }
          }
      }
    }
    int nc = 0;
#pragma unroll
    for (int q = 0; q < LA_KC; ++q) nc += (q < aa.k_cand && best[q] != INF) ? 1 : 0;
    __syncthreads();  // previous iteration's shared state fully consumed
#pragma unroll
    for (int q = 0; q < LA_KC; ++q) {
      S.cand_id[sub][q] = q < nc ? bid[q] : 0xffffffffu;
      if constexpr (FULL) S.cand_cost[sub][q] = best[q];
    }
    S.ncand[sub] = valid ? (unsigned char)nc : (unsigned char)255;
    if constexpr (FULL) {
      S.u[sub] = 0.0;
      for (int c = sub; c <= LA_COLS; c += LM_SUB) {
        S.v[c] = 0.0;
        S.p[c] = 0;
        S.flag[c] = 0;
      }
    }
    __syncthreads();
    // ---- phase 2: column slot of every candidate = its first occurrence over (row, rank); a shared
    // landmark gets ONE column.  col[] stays in registers; the cheapest edge is found on the way.
    int col[LA_KC];
    double cmin = aa.new_mh;
    int choice = valid ? sub : -1, choice_q = -1;
#pragma unroll
    for (int q = 0; q < LA_KC; ++q) {
      col[q] = LM_SUB + sub * LA_KC + q;
      if (q < nc) {
        const u32 id = bid[q];
        bool found = false;
        for (int r = 0; r < sub && !found; ++r) {
          const int ncr = S.ncand[r] == 255 ? 0 : S.ncand[r];
          for (int t = 0; t < ncr; ++t)
            if (S.cand_id[r][t] == id) {
              col[q] = LM_SUB + r * LA_KC + t;
              found = true;
              break;
            }
        }
        if (valid && best[q] < cmin) {
          cmin = best[q];
          choice = col[q];
          choice_q = q;
        }
        if constexpr (FULL) S.cand_col[sub][q] = (unsigned char)col[q];
      }
    }
    double cost = 0.0;
    int nvalid = 0, asg = -2;
    if constexpr (!FULL) {
      // ---- phase 2b: if every detection's cheapest edge leads to a different column, those edges ARE
      // the optimum (the sum of the row minima is a lower bound): the usual case on a sparse map
      S.choice[sub] = (unsigned char)(choice < 0 ? 255 : choice);
      __syncthreads();
      int clash = 0;
      if (choice >= 0)
        for (int r = 0; r < LM_SUB; ++r) clash |= (r != sub && S.choice[r] == (unsigned char)choice) ? 1 : 0;
#pragma unroll
      for (int o2 = LM_SUB / 2; o2 > 0; o2 >>= 1) clash |= __shfl_xor(clash, o2, 64);
      if (clash) {
        if (sub == 0 && live) aa.worklist[atomicAdd(aa.work_count, 1)] = (int)i;
        continue;  // uniform over the 16 lanes of the particle; the block-level barriers stay matched
                   // because every thread still executes the same number of loop iterations
      }
      if (valid) {
        nvalid = 1;
        cost = cmin;
        asg = choice_q < 0 ? -1 : (aa.assign_out ? (int)aa.orig[bid[choice_q]] : 0);
      }
    } else {
      __syncthreads();
      // ---- phase 3: sparse shortest-augmenting-path assignment, one lane per particle
      if (sub == 0 && live) {
        for (int i0row = 0; i0row < LM_SUB; ++i0row) {
          if (S.ncand[i0row] == 255) continue;
          int ntouched = 0;
          int j0 = LA_START;
          S.p[LA_START] = (unsigned char)(i0row + 1);
          S.flag[LA_START] = 2;
          for (;;) {
            const int r = S.p[j0] - 1;
            const double ur = S.u[r];
            // relax the edges of row r: private new-landmark column, then its candidates
            const int ne = 1 + S.ncand[r];
            for (int e = 0; e < ne; ++e) {
              const int j = e == 0 ? r : S.cand_col[r][e - 1];
              const double c = e == 0 ? aa.new_mh : S.cand_cost[r][e - 1];
              const unsigned char f = S.flag[j];
              if (f & 2) continue;
              const double cur = c - ur - S.v[j];
              if (!(f & 1)) {
                S.flag[j] = f | 1;
                S.list[ntouched++] = (unsigned char)j;
                S.minv[j] = INF;
              }
              if (cur < S.minv[j]) {
                S.minv[j] = cur;
                S.way[j] = (unsigned char)j0;
              }
            }
            double delta = INF;
            int j1 = -1;
            for (int t = 0; t < ntouched; ++t) {
              const int j = S.list[t];
              if (!(S.flag[j] & 2) && S.minv[j] < delta) {
                delta = S.minv[j];
                j1 = j;
              }
            }
            // potentials: rows of the used columns (incl. the start) go up, used columns go down
            S.u[i0row] += delta;
            for (int t = 0; t < ntouched; ++t) {
              const int j = S.list[t];
              if (S.flag[j] & 2) {
                S.u[S.p[j] - 1] += delta;
                S.v[j] -= delta;
              } else {
                S.minv[j] -= delta;
              }
            }
            j0 = j1;
            S.flag[j0] |= 2;
            if (S.p[j0] == 0) break;
          }
          // augment along the alternating path back to the start
          while (j0 != LA_START) {
            const int j1 = S.way[j0];
            S.p[j0] = S.p[j1];
            j0 = j1;
          }
          for (int t = 0; t < ntouched; ++t) S.flag[S.list[t]] = 0;
        }
      }
      __syncthreads();
      // ---- phase 4: every detection reads its matched edge
      if (live && valid) {
        nvalid = 1;
        asg = -1;
        cost = aa.new_mh;
        if (S.p[sub] != sub + 1) {
#pragma unroll
          for (int q = 0; q < LA_KC; ++q)
            if (q < nc && S.p[col[q]] == sub + 1) {
              cost = best[q];
              asg = aa.assign_out ? (int)aa.orig[bid[q]] : 0;
            }
        }
      }
    }
    if (aa.assign_out && live && i < aa.n_keep && sub < a.n_det) aa.assign_out[i * a.n_det + sub] = asg;
#pragma unroll
    for (int o2 = LM_SUB / 2; o2 > 0; o2 >>= 1) {
      cost += __shfl_xor(cost, o2, 64);
      nvalid += __shfl_xor(nvalid, o2, 64);
    }
    if (sub == 0 && live) {
      const double val = -0.5 * cost - (double)nvalid * a.lognorm;
      a.lw[i] = a.accumulate ? a.lw[i] + val : val;
    }
  }
}

struct LandmarkDev {
  std::vector<double> host_xyz;  // as given
  double* lm = nullptr;
  u32* cell_start = nullptr;
  u32* orig = nullptr;      // cell-ordered slot -> index in the caller's landmark list
  uint2* nb_cell = nullptr;   // (gx + 4) x (gy + 4): flattened 3 x 3 neighbourhood of a query cell (k_landmark_update)
  double4* nb_list = nullptr;
  int gx = 0, gy = 0;
  double x0 = 0, y0 = 0, cs = 0;
  double built_for = -1.0;  // gate radius the grid was built for
  // Mahalanobis mode (mcl_set_landmark_noise)
  std::vector<double> host_cov;  // n x 6 as given, or empty
  double* lmcov = nullptr;       // cell order
  bool maha = false, have_q = false;
  double Q[6] = {0, 0, 0, 0, 0, 0};
  double lam_cov_max = 0.0;      // largest eigenvalue over the landmark covariances (bound: Gershgorin)
};

inline void landmarks_free(LandmarkDev* L) {
  if (!L) return;
  if (L->lm) (void)hipFree(L->lm);
  if (L->cell_start) (void)hipFree(L->cell_start);
  if (L->orig) (void)hipFree(L->orig);
  if (L->nb_cell) (void)hipFree(L->nb_cell);
  if (L->nb_list) (void)hipFree(L->nb_list);
  if (L->lmcov) (void)hipFree(L->lmcov);
  delete L;
}

// (re)build the xy cell grid for a gate radius r (cells >= r, at most 2048 x 2048)
inline int landmarks_build(LandmarkDev* L, double r, std::string* err) {
  if (L->built_for == r && L->lm) return MCL_OK;
  const size_t n = L->host_xyz.size() / 3;
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (size_t i = 0; i < n; ++i) {
    xmin = std::min(xmin, L->host_xyz[3 * i]);
    xmax = std::max(xmax, L->host_xyz[3 * i]);
    ymin = std::min(ymin, L->host_xyz[3 * i + 1]);
    ymax = std::max(ymax, L->host_xyz[3 * i + 1]);
  }
  double cs = std::max(r, 1e-6);
  cs = std::max(cs, std::max(xmax - xmin, ymax - ymin) / 2048.0);
  L->cs = cs;
  // one cell of padding so that a query just outside the bbox still sees its 3x3 neighbourhood
  L->x0 = xmin - cs;
  L->y0 = ymin - cs;
  L->gx = (int)std::floor((xmax - L->x0) / cs) + 2;
  L->gy = (int)std::floor((ymax - L->y0) / cs) + 2;
  const size_t nc = (size_t)L->gx * L->gy;
  std::vector<u32> start(nc + 1, 0u), fill(nc, 0u);
  auto cell = [&](size_t i) {
    int a = (int)std::floor((L->host_xyz[3 * i] - L->x0) / cs), b = (int)std::floor((L->host_xyz[3 * i + 1] - L->y0) / cs);
    a = std::min(std::max(a, 0), L->gx - 1);
    b = std::min(std::max(b, 0), L->gy - 1);
    return (size_t)a * L->gy + b;
  };
  for (size_t i = 0; i < n; ++i) start[cell(i) + 1]++;
  for (size_t c = 0; c < nc; ++c) start[c + 1] += start[c];
  std::vector<double> sorted(3 * std::max<size_t>(n, 1));
  std::vector<u32> orig(std::max<size_t>(n, 1));
  for (size_t i = 0; i < n; ++i) {
    const size_t c = cell(i), r2 = start[c] + fill[c]++;
    orig[r2] = (u32)i;
    sorted[3 * r2] = L->host_xyz[3 * i];
    sorted[3 * r2 + 1] = L->host_xyz[3 * i + 1];
    sorted[3 * r2 + 2] = L->host_xyz[3 * i + 2];
  }
  if (L->lm) (void)hipFree(L->lm);
  if (L->cell_start) (void)hipFree(L->cell_start);
  if (L->orig) (void)hipFree(L->orig);
  L->lm = nullptr;
  L->cell_start = nullptr;
  L->orig = nullptr;
  if (hipMalloc(&L->lm, sizeof(double) * sorted.size()) != hipSuccess ||
      hipMalloc(&L->orig, sizeof(u32) * orig.size()) != hipSuccess ||
      hipMalloc(&L->cell_start, sizeof(u32) * (nc + 1)) != hipSuccess) {
    *err = "update_landmarks: device allocation failed";
    return MCL_ERR_ALLOC;
  }
  if (hipMemcpy(L->lm, sorted.data(), sizeof(double) * sorted.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(L->orig, orig.data(), sizeof(u32) * orig.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(L->cell_start, start.data(), sizeof(u32) * (nc + 1), hipMemcpyHostToDevice) != hipSuccess) {
    *err = "update_landmarks: upload failed";
    return MCL_ERR_HIP;
  }
  // the neighbourhood table: for every query cell (-2 .. gx + 1) x (-2 .. gy + 1) the landmarks of the cells
  // (cx - 1 .. cx + 1) x (cy - 1 .. cy + 1) that exist, column by column and in cell order within a column -- the
  // order in which the three row ranges of cell_start are walked (the assignment kernels still walk those)
  {
    const int ex = L->gx + 4, ey = L->gy + 4;
    std::vector<uint2> nb((size_t)ex * ey);
    std::vector<double4> list;
    list.reserve(9 * n + 1);
    for (int qx = -2; qx <= L->gx + 1; ++qx)
      for (int qy = -2; qy <= L->gy + 1; ++qy) {
        const int iy0 = std::max(qy - 1, 0), iy1 = std::min(qy + 1, L->gy - 1);
        const size_t first = list.size();
        for (int k = 0; k < 3 && iy0 <= iy1; ++k) {
          const int ix = qx - 1 + k;
          if (ix < 0 || ix >= L->gx) continue;
          for (u32 e = start[(size_t)ix * L->gy + iy0]; e < start[(size_t)ix * L->gy + iy1 + 1]; ++e) {
            double4 v;
            v.x = sorted[3 * (size_t)e];
            v.y = sorted[3 * (size_t)e + 1];
            v.z = sorted[3 * (size_t)e + 2];
            const long long bits = (long long)e;
            memcpy(&v.w, &bits, sizeof v.w);
            list.push_back(v);
          }
        }
        nb[(size_t)(qx + 2) * ey + (qy + 2)] = make_uint2((u32)first, (u32)(list.size() - first));
      }
    if (list.empty()) list.push_back(double4{0, 0, 0, 0});
    if (L->nb_cell) (void)hipFree(L->nb_cell);
    if (L->nb_list) (void)hipFree(L->nb_list);
    L->nb_cell = nullptr;
    L->nb_list = nullptr;
    if (hipMalloc(&L->nb_cell, sizeof(uint2) * nb.size()) != hipSuccess ||
        hipMalloc(&L->nb_list, sizeof(double4) * list.size()) != hipSuccess) {
      *err = "update_landmarks: device allocation failed";
      return MCL_ERR_ALLOC;
    }
    if (hipMemcpy(L->nb_cell, nb.data(), sizeof(uint2) * nb.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(L->nb_list, list.data(), sizeof(double4) * list.size(), hipMemcpyHostToDevice) != hipSuccess) {
      *err = "update_landmarks: upload failed";
      return MCL_ERR_HIP;
    }
  }
  if (L->lmcov) (void)hipFree(L->lmcov);
  L->lmcov = nullptr;
  if (!L->host_cov.empty()) {
    std::vector<double> sc(6 * std::max<size_t>(n, 1));
    for (size_t r2 = 0; r2 < n; ++r2)
      for (int k = 0; k < 6; ++k) sc[6 * r2 + k] = L->host_cov[6 * (size_t)orig[r2] + k];
    if (hipMalloc(&L->lmcov, sizeof(double) * sc.size()) != hipSuccess ||
        hipMemcpy(L->lmcov, sc.data(), sizeof(double) * sc.size(), hipMemcpyHostToDevice) != hipSuccess) {
      *err = "update_landmarks: covariance upload failed";
      return MCL_ERR_ALLOC;
    }
  }
  L->built_for = r;
  return MCL_OK;
}
