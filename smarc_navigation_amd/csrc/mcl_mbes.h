// mcl_mbes.h -- MBES measurement update: per-particle, per-beam ray-cast against the bathymetric
// map -> expected ranges -> Gaussian log-likelihood (north_star; no reference symbol).
//
// Mapping: one wavefront per particle, lanes = consecutive beams (coherent fan: neighbouring lanes
// walk neighbouring cells).  A workgroup of P waves stages the height tile that bounds the fans of
// its P particles into LDS once, then every ray marches the tile from LDS in fp32, tile-local
// coordinates.  Not HBM-bound: per step the compulsory HBM traffic is 48 B/particle + the map once.
#pragma once
#include "mcl_device.h"

#define MBES_WAVES 16                        // particles per workgroup
#define MBES_THREADS (MBES_WAVES * 64)
#define MBES_TILE_FLOATS 12288               // 48 KiB height tile in LDS

struct MbesArgs {
  const double* st[6];  // x,y,z,roll,pitch,yaw (odom frame)
  long long n;
  double m2o[12];       // rows 0..2 of map<-odom
  double off_t[3];      // sensor offset translation in base_link
  double off_R[9];      // sensor offset rotation
  const float2* beam_sc;  // (sin a_b, cos a_b)
  const float* ranges;    // measured (nullptr -> expected-only call)
  int n_beams;
  const float* grid;      // z[ix*ny + iy]
  int nx, ny;
  double ox, oy, inv_res;
  float res;
  float zmin_map, zmax_map;
  float inv_sigma, r_max;
  double lognorm;         // log(sigma sqrt(2 pi))
  double* lw;             // out: log-likelihood per particle (may be nullptr)
  float* exp_out;         // out: expected ranges [(i-exp_first)*B + b] (may be nullptr)
  long long exp_first, exp_count;
};

struct HeightLDS {
  const float* t;
  int th;  // tile nodes in y (row pitch)
  __device__ __forceinline__ void corners(int ix, int iy, float& h00, float& h10, float& h01, float& h11) const {
    const float* p = t + ix * th + iy;
    h00 = p[0];
    h01 = p[1];
    h10 = p[th];
    h11 = p[th + 1];
  }
};
struct HeightGlobal {
  const float* g;
  int ny;
  __device__ __forceinline__ void corners(int ix, int iy, float& h00, float& h10, float& h01, float& h11) const {
    const float* p = g + (size_t)ix * ny + iy;
    h00 = p[0];
    h01 = p[1];
    h10 = p[ny];
    h11 = p[ny + 1];
  }
};

// First intersection of the ray (u0 + t du, v0 + t dv, oz + t dz), u/v in CELL units of a cw x ch
// cell domain, t in metres, with the bilinear height field.  Returns r_max when there is none.
// Same definition as oracle/mcl_oracle.c:orc_ray_grid (fp64) -- here fp32, tile-local.
template <class H>
__device__ __forceinline__ float march_heightfield(const H& hm, int cw, int ch, float u0, float v0, float oz,
                                                   float du, float dv, float dz, float t_lo, float r_max) {
  float t0 = t_lo, t1 = r_max;
  // clip to the domain [0,cw] x [0,ch]
  if (du == 0.f) {
    if (u0 < 0.f || u0 > (float)cw) return r_max;
  } else {
    float inv = 1.f / du;
    float ta = (0.f - u0) * inv, tb = ((float)cw - u0) * inv;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  if (dv == 0.f) {
    if (v0 < 0.f || v0 > (float)ch) return r_max;
  } else {
    float inv = 1.f / dv;
    float ta = (0.f - v0) * inv, tb = ((float)ch - v0) * inv;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  if (!(t0 <= t1)) return r_max;
  const float pu = u0 + t0 * du, pv = v0 + t0 * dv;
  int ix = min(max((int)floorf(pu), 0), cw - 1);
  int iy = min(max((int)floorf(pv), 0), ch - 1);
  if (du < 0.f && ix > 0 && (float)ix >= pu) --ix;
  if (dv < 0.f && iy > 0 && (float)iy >= pv) --iy;
  const int sx = du > 0.f ? 1 : -1, sy = dv > 0.f ? 1 : -1;
  const float inv_du = du != 0.f ? 1.f / du : 0.f, inv_dv = dv != 0.f ? 1.f / dv : 0.f;
  const float INF = __builtin_inff();
  float t_in = t0;
  float z_in = oz + t_in * dz;
  bool first = true;
  const int max_steps = cw + ch + 4;
  for (int step = 0; step < max_steps; ++step) {
    const float tnx = du != 0.f ? ((float)(ix + (du > 0.f ? 1 : 0)) - u0) * inv_du : INF;
    const float tny = dv != 0.f ? ((float)(iy + (dv > 0.f ? 1 : 0)) - v0) * inv_dv : INF;
    float t_out = fminf(fminf(tnx, tny), t1);
    float h00, h10, h01, h11;
    hm.corners(ix, iy, h00, h10, h01, h11);
    const float hmax = fmaxf(fmaxf(h00, h10), fmaxf(h01, h11));
    const float z_out = oz + t_out * dz;
    if (fminf(z_in, z_out) <= hmax || first) {
      const float uc = u0 - (float)ix, vc = v0 - (float)iy;
      const float B = h10 - h00, C = h01 - h00, D = (h00 - h10) - (h01 - h11);
      const float c0 = oz - (h00 + B * uc + C * vc + D * uc * vc);
      const float c1 = dz - (B * du + C * dv + D * (uc * dv + vc * du));
      const float c2 = -D * du * dv;
      if (first) {
        const float f0 = c0 + t_in * (c1 + t_in * c2);
        if (f0 <= 0.f) return t_in;  // origin / map entry at or below the seabed
        first = false;
      }
      const float f_out = c0 + t_out * (c1 + t_out * c2);
      bool hit = f_out <= 0.f;
      float hi_t = t_out;
      if (!hit && c2 != 0.f) {  // grazing: both ends above, dips below in between
        const float tv = -0.5f * c1 / c2;
        if (tv > t_in && tv < t_out && c0 + tv * (c1 + tv * c2) < 0.f) {
          hit = true;
          hi_t = tv;
        }
      }
      if (hit) {
        // smallest root in [t_in, hi_t]; f(t_in) > 0 >= f(hi_t)
        float root;
        if (c2 == 0.f) {
          root = -c0 / c1;
        } else {
          const float disc = fmaxf(c1 * c1 - 4.f * c2 * c0, 0.f);
          const float sq = sqrtf(disc);
          const float qv = -0.5f * (c1 + (c1 >= 0.f ? sq : -sq));
          const float r1 = qv != 0.f ? c0 / qv : 0.f;
          const float r2 = qv / c2;
          const float ra = fminf(r1, r2), rb = fmaxf(r1, r2);
          root = (ra >= t_in - 1e-3f && ra <= hi_t + 1e-3f) ? ra : rb;
        }
        root = fminf(fmaxf(root, t_in), hi_t);
        return fminf(root, r_max);
      }
    }
    if (t_out >= t1) return r_max;
    if (tnx <= tny)
      ix += sx;
    else
      iy += sy;
    if (ix < 0 || iy < 0 || ix >= cw || iy >= ch) return r_max;
    t_in = t_out;
    z_in = z_out;
  }
  return r_max;
}

struct MbesParticle {  // per-particle pose constants shared through LDS
  double um, vm;       // sensor origin in GLOBAL cell units (double: precise before tile shift)
  float oz;
  float c1[3], c2[3];  // columns 1, 2 of R_map_sensor:  D_b = sin a_b * c1 - cos a_b * c2
  int valid;
};

__device__ __forceinline__ void mbes_pose(const MbesArgs& a, long long i, MbesParticle& P) {
  const double x = a.st[0][i], y = a.st[1][i], z = a.st[2][i];
  double sr, cr, sp, cp, sy, cy;
  sincos(a.st[3][i], &sr, &cr);
  sincos(a.st[4][i], &sp, &cp);
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
  double o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = (a.m2o[r * 4 + 0] * x + a.m2o[r * 4 + 1] * y + a.m2o[r * 4 + 2] * z + a.m2o[r * 4 + 3]) +
           (Rmp[r * 3 + 0] * a.off_t[0] + Rmp[r * 3 + 1] * a.off_t[1] + Rmp[r * 3 + 2] * a.off_t[2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    P.c1[r] = (float)(Rmp[r * 3 + 0] * a.off_R[1] + Rmp[r * 3 + 1] * a.off_R[4] + Rmp[r * 3 + 2] * a.off_R[7]);
    P.c2[r] = (float)(Rmp[r * 3 + 0] * a.off_R[2] + Rmp[r * 3 + 1] * a.off_R[5] + Rmp[r * 3 + 2] * a.off_R[8]);
  }
  P.um = (o[0] - a.ox) * a.inv_res;
  P.vm = (o[1] - a.oy) * a.inv_res;
  P.oz = (float)o[2];
  P.valid = 1;
}

template <bool EXPECT_ONLY>
__global__ void __launch_bounds__(MBES_THREADS) k_mbes_grid(MbesArgs a) {
  __shared__ float tile[MBES_TILE_FLOATS];
  __shared__ MbesParticle sp[MBES_WAVES];
  __shared__ float red[4][MBES_WAVES];  // umin, umax, vmin, vmax per wave ; later zmax
  __shared__ int tinfo[6];              // tx0, ty0, tw, th, use_lds
  __shared__ float tzmax;

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long ngroups = (a.n + MBES_WAVES - 1) / MBES_WAVES;
  const float inv_res = (float)a.inv_res;

  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long i = grp * MBES_WAVES + w;
    __syncthreads();  // previous group's tile / sp fully consumed
    if (threadIdx.x < MBES_WAVES) {
      long long ip = grp * MBES_WAVES + threadIdx.x;
      if (ip < a.n)
        mbes_pose(a, ip, sp[threadIdx.x]);
      else
        sp[threadIdx.x].valid = 0;
    }
    __syncthreads();
    const MbesParticle P = sp[w];
    // ---- footprint of this wave's fan (global cell units) -> block bbox
    float umin = (float)P.um, umax = umin, vmin = (float)P.vm, vmax = vmin;
    if (P.valid) {
      for (int b = lane; b < a.n_beams; b += 64) {
        const float2 sc = a.beam_sc[b];
        const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
        const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
        const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
        float t_end = a.r_max;
        if (dz < -1e-6f) t_end = fminf(t_end, fmaxf((a.zmin_map - P.oz) / dz, 0.f));
        const float ue = (float)P.um + t_end * dx * inv_res, ve = (float)P.vm + t_end * dy * inv_res;
        umin = fminf(umin, ue);
        umax = fmaxf(umax, ue);
        vmin = fminf(vmin, ve);
        vmax = fmaxf(vmax, ve);
      }
    }
    umin = wave_min(umin);
    umax = wave_max(umax);
    vmin = wave_min(vmin);
    vmax = wave_max(vmax);
    if (lane == 0) {
      const bool ok = P.valid != 0;
      red[0][w] = ok ? umin : __builtin_inff();
      red[1][w] = ok ? umax : -__builtin_inff();
      red[2][w] = ok ? vmin : __builtin_inff();
      red[3][w] = ok ? vmax : -__builtin_inff();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a0 = red[0][0], a1 = red[1][0], b0 = red[2][0], b1 = red[3][0];
      for (int k = 1; k < MBES_WAVES; ++k) {
        a0 = fminf(a0, red[0][k]);
        a1 = fmaxf(a1, red[1][k]);
        b0 = fminf(b0, red[2][k]);
        b1 = fmaxf(b1, red[3][k]);
      }
      // node range [tx0, tx1] clipped to the map, one cell of margin for fp32 slop
      int tx0 = max((int)floorf(a0) - 1, 0), tx1 = min((int)floorf(a1) + 2, a.nx - 1);
      int ty0 = max((int)floorf(b0) - 1, 0), ty1 = min((int)floorf(b1) + 2, a.ny - 1);
      int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
      int use = (tw >= 2 && th >= 2 && (long long)tw * th <= MBES_TILE_FLOATS) ? 1 : 0;
      if (tw < 2 || th < 2) use = -1;  // fans entirely off the map
      tinfo[0] = tx0;
      tinfo[1] = ty0;
      tinfo[2] = tw;
      tinfo[3] = th;
      tinfo[4] = use;
    }
    __syncthreads();
    const int tx0 = tinfo[0], ty0 = tinfo[1], tw = tinfo[2], th = tinfo[3], use = tinfo[4];
    float zmax = a.zmax_map;
    if (use == 1) {
      // ---- stage the tile (coalesced along iy) and find its max height
      float m = -__builtin_inff();
      const int cells = tw * th;
      for (int k = threadIdx.x; k < cells; k += MBES_THREADS) {
        const int ix = k / th, iy = k - ix * th;
        const float h = a.grid[(size_t)(tx0 + ix) * a.ny + (ty0 + iy)];
        tile[k] = h;
        m = fmaxf(m, h);
      }
      m = wave_max(m);
      if (lane == 0) red[0][w] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        float mm = red[0][0];
        for (int k = 1; k < MBES_WAVES; ++k) mm = fmaxf(mm, red[0][k]);
        tzmax = mm;
      }
      __syncthreads();
      zmax = tzmax;
    }
    if (!P.valid) continue;
    // ---- march this particle's beams
    float acc = 0.f;
    int nvalid = 0;
    const float u0 = use == 1 ? (float)(P.um - (double)tx0) : (float)P.um;
    const float v0 = use == 1 ? (float)(P.vm - (double)ty0) : (float)P.vm;
    for (int b = lane; b < a.n_beams; b += 64) {
      const float2 sc = a.beam_sc[b];
      const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
      const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
      const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
      float t_lo = 0.f;  // skip the water column above the tile's highest node
      if (dz < 0.f && P.oz > zmax) t_lo = fmaxf((zmax - P.oz) / dz - 1e-3f, 0.f);
      float e;
      if (use == 1) {
        HeightLDS hm{tile, th};
        e = march_heightfield(hm, tw - 1, th - 1, u0, v0, P.oz, dx * inv_res, dy * inv_res, dz, t_lo, a.r_max);
      } else if (use == 0) {
        HeightGlobal hm{a.grid, a.ny};
        e = march_heightfield(hm, a.nx - 1, a.ny - 1, u0, v0, P.oz, dx * inv_res, dy * inv_res, dz, t_lo, a.r_max);
      } else {
        e = a.r_max;
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
      double accd = wave_sum((double)acc);
      int nv = wave_sum(nvalid);
      if (lane == 0) a.lw[i] = -0.5 * accd - (double)nv * a.lognorm;
    }
  }
}
