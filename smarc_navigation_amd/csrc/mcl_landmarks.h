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
  int gx, gy;
  double x0, y0, inv_cs;
  double inv_s2, gate, lognorm;  // lognorm = 3/2 log(2 pi) + 3 log(sigma)
  int k;
  int accumulate;         // add to lw instead of overwriting
  double* lw;
};

__global__ void __launch_bounds__(256) k_landmark_update(LandmarkArgs a) {
  const int sub = threadIdx.x & (LM_SUB - 1);
  const long long per_block = blockDim.x / LM_SUB;
  for (long long i = blockIdx.x * per_block + threadIdx.x / LM_SUB; i < a.n; i += (long long)gridDim.x * per_block) {
    // sensor pose in the map frame (fp64): M = m2o * T(x,y,z) R(rpy) * T_off R_off
    const double x = a.st[0][i], y = a.st[1][i], z = a.st[2][i];
    double sr, cr, sp, cp, sy, cy;
    sincos(a.st[3][i], &sr, &cr);
    sincos(a.st[4][i], &sp, &cp);
    sincos(a.st[5][i], &sy, &cy);
    const double Rp[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                          sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                          -sp,     cp * sr,                cp * cr};
    double Rmp[9], Rs[9], o[3];
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
    double acc = 0.0;
    int nvalid = 0;
    for (int d = sub; d < a.n_det; d += LM_SUB) {
      const double zx = a.det[3 * d], zy = a.det[3 * d + 1], zz = a.det[3 * d + 2];
      if (!(zx == zx && zy == zy && zz == zz)) continue;  // NaN = invalid detection
      const double px = o[0] + Rs[0] * zx + Rs[1] * zy + Rs[2] * zz;
      const double py = o[1] + Rs[3] * zx + Rs[4] * zy + Rs[5] * zz;
      const double pz = o[2] + Rs[6] * zx + Rs[7] * zy + Rs[8] * zz;
      double best[LM_MAX_K];
#pragma unroll
      for (int q = 0; q < LM_MAX_K; ++q) best[q] = __builtin_inf();
      const int cx = (int)floor((px - a.x0) * a.inv_cs), cyi = (int)floor((py - a.y0) * a.inv_cs);
      for (int ix = max(cx - 1, 0); ix <= min(cx + 1, a.gx - 1); ++ix)
        for (int iy = max(cyi - 1, 0); iy <= min(cyi + 1, a.gy - 1); ++iy) {
          const size_t c = (size_t)ix * a.gy + iy;
          for (u32 e = a.cell_start[c]; e < a.cell_start[c + 1]; ++e) {
            const double dx = px - a.lm[3 * (size_t)e], dy = py - a.lm[3 * (size_t)e + 1], dz = pz - a.lm[3 * (size_t)e + 2];
            double m = (dx * dx + dy * dy + dz * dz) * a.inv_s2;
            if (m <= a.gate) {
              // insert into the sorted k-best list
#pragma unroll
              for (int q = 0; q < LM_MAX_K; ++q) {
                if (m < best[q]) {
                  const double t = best[q];
                  best[q] = m;
                  m = t;
                }
              }
            }
          }
        }
      double lwd;
      if (best[0] == __builtin_inf()) {
        lwd = -0.5 * a.gate;
      } else {
        // log-sum-exp over the k nearest inside the gate, anchored at the nearest
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < LM_MAX_K; ++q)
          if (q < a.k && best[q] != __builtin_inf()) s += exp(-0.5 * (best[q] - best[0]));
        lwd = -0.5 * best[0] + log(s);
      }
      acc += lwd;
      ++nvalid;
    }
    // reduce over the LM_SUB lanes of this particle
#pragma unroll
    for (int o2 = LM_SUB / 2; o2 > 0; o2 >>= 1) {
      acc += __shfl_xor(acc, o2, 64);
      nvalid += __shfl_xor(nvalid, o2, 64);
    }
    if (sub == 0) {
      const double v = acc - (double)nvalid * a.lognorm;
      a.lw[i] = a.accumulate ? a.lw[i] + v : v;
    }
  }
}

struct LandmarkDev {
  std::vector<double> host_xyz;  // as given
  double* lm = nullptr;
  u32* cell_start = nullptr;
  int gx = 0, gy = 0;
  double x0 = 0, y0 = 0, cs = 0;
  double built_for = -1.0;  // gate radius the grid was built for
};

inline void landmarks_free(LandmarkDev* L) {
  if (!L) return;
  if (L->lm) (void)hipFree(L->lm);
  if (L->cell_start) (void)hipFree(L->cell_start);
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
  for (size_t i = 0; i < n; ++i) {
    const size_t c = cell(i), r2 = start[c] + fill[c]++;
    sorted[3 * r2] = L->host_xyz[3 * i];
    sorted[3 * r2 + 1] = L->host_xyz[3 * i + 1];
    sorted[3 * r2 + 2] = L->host_xyz[3 * i + 2];
  }
  if (L->lm) (void)hipFree(L->lm);
  if (L->cell_start) (void)hipFree(L->cell_start);
  L->lm = nullptr;
  L->cell_start = nullptr;
  if (hipMalloc(&L->lm, sizeof(double) * sorted.size()) != hipSuccess ||
      hipMalloc(&L->cell_start, sizeof(u32) * (nc + 1)) != hipSuccess) {
    *err = "update_landmarks: device allocation failed";
    return MCL_ERR_ALLOC;
  }
  if (hipMemcpy(L->lm, sorted.data(), sizeof(double) * sorted.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(L->cell_start, start.data(), sizeof(u32) * (nc + 1), hipMemcpyHostToDevice) != hipSuccess) {
    *err = "update_landmarks: upload failed";
    return MCL_ERR_HIP;
  }
  L->built_for = r;
  return MCL_OK;
}
