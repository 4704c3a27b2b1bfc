// mcl_mesh.h -- triangle-mesh bathymetry: host-side acceleration-structure build + ray-cast kernel.
//
// Structure (built once per map on the host, resident in HBM): a uniform xy cell grid over the
// mesh; per cell a contiguous run of triangle records (v0 RELATIVE TO THE CELL CORNER, e1, e2 as
// three float4 -> cell-local fp32 arithmetic keeps full precision on a 700 m map) and the z-range
// of the cell's triangles.  The kernel is the grid kernel's twin: one wavefront per particle, lanes
// = consecutive beams, the z-range tile that bounds the workgroup's fans staged in LDS; a ray walks
// the cells (2-D DDA), culls on the LDS z-range and runs Moller-Trumbore only on cells whose
// z-range it actually enters (triangle records come from L2).
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/mcl.h"
#include "mcl_mbes.h"

struct MeshDev {
  float4* tri = nullptr;       // 3 float4 per (cell, triangle) record
  u32* cell_start = nullptr;   // gx*gy + 1
  float2* cell_z = nullptr;    // (zmin, zmax) per cell; empty = (+inf, -inf)
  int gx = 0, gy = 0;
  double x0 = 0, y0 = 0, cs = 1;
  float zmin = 0, zmax = 0;
  size_t n_records = 0;
};

inline void mesh_free(MeshDev* m) {
  if (!m) return;
  if (m->tri) (void)hipFree(m->tri);
  if (m->cell_start) (void)hipFree(m->cell_start);
  if (m->cell_z) (void)hipFree(m->cell_z);
  delete m;
}

inline int mesh_build(const float* verts, int64_t nv, const uint32_t* tris, int64_t nt, MeshDev** out,
                      std::string* err) {
  *out = nullptr;
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  float zmin = INFINITY, zmax = -INFINITY;
  for (int64_t i = 0; i < nv; ++i) {
    xmin = std::min(xmin, (double)verts[3 * i]);
    xmax = std::max(xmax, (double)verts[3 * i]);
    ymin = std::min(ymin, (double)verts[3 * i + 1]);
    ymax = std::max(ymax, (double)verts[3 * i + 1]);
    zmin = std::min(zmin, verts[3 * i + 2]);
    zmax = std::max(zmax, verts[3 * i + 2]);
  }
  for (int64_t k = 0; k < 3 * nt; ++k)
    if (tris[k] >= (uint64_t)nv) {
      *err = "set_map_mesh: triangle index out of range";
      return MCL_ERR_INVALID;
    }
  const double area = (xmax - xmin) * (ymax - ymin);
  double cs = std::sqrt(area / (double)std::max<int64_t>(nt / 2, 1));
  if (!(cs > 0.0)) cs = 1.0;
  while (((xmax - xmin) / cs + 1.0) * ((ymax - ymin) / cs + 1.0) > 6.4e7) cs *= 2.0;
  MeshDev* m = new MeshDev();
  m->cs = cs;
  m->x0 = xmin;
  m->y0 = ymin;
  // cells cover [x0, x0 + gx*cs]; a vertex exactly on the upper border belongs to the last cell
  m->gx = std::max(1, (int)std::ceil((xmax - xmin) / cs - 1e-9));
  m->gy = std::max(1, (int)std::ceil((ymax - ymin) / cs - 1e-9));
  m->zmin = zmin;
  m->zmax = zmax;
  const size_t nc = (size_t)m->gx * (size_t)m->gy;
  auto cell_range = [&](int64_t k, int& a0, int& a1, int& b0, int& b1) {
    double txmin = INFINITY, txmax = -INFINITY, tymin = INFINITY, tymax = -INFINITY;
    for (int c = 0; c < 3; ++c) {
      const float* v = verts + 3 * (size_t)tris[3 * k + c];
      txmin = std::min(txmin, (double)v[0]);
      txmax = std::max(txmax, (double)v[0]);
      tymin = std::min(tymin, (double)v[1]);
      tymax = std::max(tymax, (double)v[1]);
    }
    const double e = 1e-7;  // a triangle that only touches a cell border is not binned beyond it
    a0 = (int)std::floor((txmin - m->x0) / cs + e);
    a1 = (int)std::floor((txmax - m->x0) / cs - e);
    b0 = (int)std::floor((tymin - m->y0) / cs + e);
    b1 = (int)std::floor((tymax - m->y0) / cs - e);
    a0 = std::min(std::max(a0, 0), m->gx - 1);
    b0 = std::min(std::max(b0, 0), m->gy - 1);
    a1 = std::min(std::max(a1, a0), m->gx - 1);
    b1 = std::min(std::max(b1, b0), m->gy - 1);
  };
  std::vector<u32> start(nc + 1, 0u);
  for (int64_t k = 0; k < nt; ++k) {
    int a0, a1, b0, b1;
    cell_range(k, a0, a1, b0, b1);
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b) start[(size_t)a * m->gy + b + 1]++;
  }
  for (size_t c = 0; c < nc; ++c) {
    if ((uint64_t)start[c + 1] + start[c] > 0xffffffffull) {
      *err = "set_map_mesh: too many (cell, triangle) records";
      delete m;
      return MCL_ERR_UNSUPPORTED;
    }
    start[c + 1] += start[c];
  }
  m->n_records = start[nc];
  std::vector<float4> rec(3 * std::max<size_t>(m->n_records, 1));
  std::vector<float2> cz(nc, make_float2(INFINITY, -INFINITY));
  std::vector<u32> fill(nc, 0u);
  for (int64_t k = 0; k < nt; ++k) {
    int a0, a1, b0, b1;
    cell_range(k, a0, a1, b0, b1);
    const float* v0 = verts + 3 * (size_t)tris[3 * k];
    const float* v1 = verts + 3 * (size_t)tris[3 * k + 1];
    const float* v2 = verts + 3 * (size_t)tris[3 * k + 2];
    const float tz0 = std::min(v0[2], std::min(v1[2], v2[2])), tz1 = std::max(v0[2], std::max(v1[2], v2[2]));
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b) {
        const size_t c = (size_t)a * m->gy + b;
        const size_t r = 3 * ((size_t)start[c] + fill[c]++);
        const double cx = m->x0 + a * cs, cy = m->y0 + b * cs;
        rec[r + 0] = make_float4((float)(v0[0] - cx), (float)(v0[1] - cy), v0[2], 0.f);
        rec[r + 1] = make_float4((float)((double)v1[0] - v0[0]), (float)((double)v1[1] - v0[1]),
                                 (float)((double)v1[2] - v0[2]), 0.f);
        rec[r + 2] = make_float4((float)((double)v2[0] - v0[0]), (float)((double)v2[1] - v0[1]),
                                 (float)((double)v2[2] - v0[2]), 0.f);
        cz[c].x = std::min(cz[c].x, tz0);
        cz[c].y = std::max(cz[c].y, tz1);
      }
  }
  hipError_t e1 = hipMalloc(&m->tri, sizeof(float4) * rec.size());
  hipError_t e2 = hipMalloc(&m->cell_start, sizeof(u32) * (nc + 1));
  hipError_t e3 = hipMalloc(&m->cell_z, sizeof(float2) * nc);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    *err = "set_map_mesh: device allocation failed";
    mesh_free(m);
    return MCL_ERR_ALLOC;
  }
  if (hipMemcpy(m->tri, rec.data(), sizeof(float4) * rec.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->cell_start, start.data(), sizeof(u32) * (nc + 1), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->cell_z, cz.data(), sizeof(float2) * nc, hipMemcpyHostToDevice) != hipSuccess) {
    *err = "set_map_mesh: upload failed";
    mesh_free(m);
    return MCL_ERR_HIP;
  }
  *out = m;
  return MCL_OK;
}

struct MeshArgs {
  const float4* tri;
  const u32* cell_start;
  const float2* cell_z;
  int gx, gy;
  float cs;
};

struct CellZLDS {
  const float2* t;
  int th;
  __device__ __forceinline__ float2 get(int ix, int iy) const { return t[ix * th + iy]; }
};
struct CellZGlobal {
  const float2* g;
  int gy;
  __device__ __forceinline__ float2 get(int ix, int iy) const { return g[(size_t)ix * gy + iy]; }
};

// nearest hit of the ray with the triangles binned in cell (gix, giy); ray origin given relative to
// that cell's corner.  Accepts t in [0, t_hi].
__device__ __forceinline__ float cell_triangles_hit(const MeshArgs& ma, int gix, int giy, float olx, float oly,
                                                    float olz, float dx, float dy, float dz, float t_hi) {
  const size_t c = (size_t)gix * ma.gy + giy;
  const u32 s = ma.cell_start[c], e = ma.cell_start[c + 1];
  float best = __builtin_inff();
  const float EPS = 2e-5f;
  for (u32 k = s; k < e; ++k) {
    const float4 v0 = ma.tri[3 * (size_t)k], e1 = ma.tri[3 * (size_t)k + 1], e2 = ma.tri[3 * (size_t)k + 2];
    const float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
    const float det = e1.x * px + e1.y * py + e1.z * pz;
    if (fabsf(det) < 1e-20f) continue;
    const float inv = 1.f / det;
    const float sx = olx - v0.x, sy = oly - v0.y, sz = olz - v0.z;
    const float u = (sx * px + sy * py + sz * pz) * inv;
    const float qx = sy * e1.z - sz * e1.y, qy = sz * e1.x - sx * e1.z, qz = sx * e1.y - sy * e1.x;
    const float v = (dx * qx + dy * qy + dz * qz) * inv;
    const float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
    if (u >= -EPS && v >= -EPS && u + v <= 1.f + EPS && t >= 0.f && t <= t_hi && t < best) best = t;
  }
  return best;
}

// 2-D DDA over a cw x ch cell window whose cell (0,0) is global cell (tx0, ty0); u/v in cell units.
template <class Z>
__device__ __forceinline__ float march_mesh(const Z& zc, const MeshArgs& ma, int tx0, int ty0, int cw, int ch,
                                            float u0, float v0, float oz, float du, float dv, float dx, float dy,
                                            float dz, float t_lo, float r_max) {
  float t0 = t_lo, t1 = r_max;
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
  float z_in = oz + t0 * dz;
  const int max_steps = cw + ch + 4;
  for (int step = 0; step < max_steps; ++step) {
    const float tnx = du != 0.f ? ((float)(ix + (du > 0.f ? 1 : 0)) - u0) * inv_du : INF;
    const float tny = dv != 0.f ? ((float)(iy + (dv > 0.f ? 1 : 0)) - v0) * inv_dv : INF;
    const float t_out = fminf(fminf(tnx, tny), t1);
    const float z_out = oz + t_out * dz;
    const float2 zr = zc.get(ix, iy);
    if (fminf(z_in, z_out) <= zr.y + 1e-4f && fmaxf(z_in, z_out) >= zr.x - 1e-4f) {
      const float t = cell_triangles_hit(ma, tx0 + ix, ty0 + iy, (u0 - (float)ix) * ma.cs, (v0 - (float)iy) * ma.cs,
                                         oz, dx, dy, dz, t_out + 1e-4f);
      if (t < INF) return fminf(t, r_max);
    }
    if (t_out >= t1) return r_max;
    if (tnx <= tny)
      ix += sx;
    else
      iy += sy;
    if (ix < 0 || iy < 0 || ix >= cw || iy >= ch) return r_max;
    z_in = z_out;
  }
  return r_max;
}

#define MESH_TILE_CELLS (MBES_TILE_FLOATS / 2)

template <bool EXPECT_ONLY>
__global__ void __launch_bounds__(MBES_THREADS) k_mbes_mesh(MbesArgs a, MeshArgs ma) {
  __shared__ float2 tile[MESH_TILE_CELLS];
  __shared__ MbesParticle sp[MBES_WAVES];
  __shared__ float red[4][MBES_WAVES];
  __shared__ int tinfo[6];
  __shared__ float tzmax;

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long ngroups = (a.n + MBES_WAVES - 1) / MBES_WAVES;
  const float inv_res = (float)a.inv_res;

  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long i = grp * MBES_WAVES + w;
    __syncthreads();
    if (threadIdx.x < MBES_WAVES) {
      long long ip = grp * MBES_WAVES + threadIdx.x;
      if (ip < a.n)
        mbes_pose(a, ip, sp[threadIdx.x]);
      else
        sp[threadIdx.x].valid = 0;
    }
    __syncthreads();
    const MbesParticle P = sp[w];
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
      // CELL range [tx0, tx1] clipped to the grid, one cell of margin
      int tx0 = max((int)floorf(a0) - 1, 0), tx1 = min((int)floorf(a1) + 1, ma.gx - 1);
      int ty0 = max((int)floorf(b0) - 1, 0), ty1 = min((int)floorf(b1) + 1, ma.gy - 1);
      int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
      int use = (tw >= 1 && th >= 1 && (long long)tw * th <= MESH_TILE_CELLS) ? 1 : 0;
      if (tw < 1 || th < 1) use = -1;
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
      float m = -__builtin_inff();
      const int cells = tw * th;
      for (int k = threadIdx.x; k < cells; k += MBES_THREADS) {
        const int ix = k / th, iy = k - ix * th;
        const float2 zr = ma.cell_z[(size_t)(tx0 + ix) * ma.gy + (ty0 + iy)];
        tile[k] = zr;
        m = fmaxf(m, zr.y);
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
    float acc = 0.f;
    int nvalid = 0;
    const float u0 = use == 1 ? (float)(P.um - (double)tx0) : (float)P.um;
    const float v0 = use == 1 ? (float)(P.vm - (double)ty0) : (float)P.vm;
    for (int b = lane; b < a.n_beams; b += 64) {
      const float2 sc = a.beam_sc[b];
      const float dx = sc.x * P.c1[0] - sc.y * P.c2[0];
      const float dy = sc.x * P.c1[1] - sc.y * P.c2[1];
      const float dz = sc.x * P.c1[2] - sc.y * P.c2[2];
      float t_lo = 0.f;
      if (dz < 0.f && P.oz > zmax) t_lo = fmaxf((zmax - P.oz) / dz - 1e-3f, 0.f);
      float e;
      if (use == 1) {
        CellZLDS zc{tile, th};
        e = march_mesh(zc, ma, tx0, ty0, tw, th, u0, v0, P.oz, dx * inv_res, dy * inv_res, dx, dy, dz, t_lo, a.r_max);
      } else if (use == 0) {
        CellZGlobal zc{ma.cell_z, ma.gy};
        e = march_mesh(zc, ma, 0, 0, ma.gx, ma.gy, u0, v0, P.oz, dx * inv_res, dy * inv_res, dx, dy, dz, t_lo, a.r_max);
      } else {
        e = a.r_max;
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
      double accd = wave_sum((double)acc);
      int nv = wave_sum(nvalid);
      if (lane == 0) a.lw[i] = -0.5 * accd - (double)nv * a.lognorm;
    }
  }
}

inline int mesh_launch(MeshDev* m, MbesArgs a, bool expect_only, hipStream_t stream) {
  if (!m) return MCL_ERR_STATE;
  MeshArgs ma;
  ma.tri = m->tri;
  ma.cell_start = m->cell_start;
  ma.cell_z = m->cell_z;
  ma.gx = m->gx;
  ma.gy = m->gy;
  ma.cs = (float)m->cs;
  a.grid = nullptr;
  a.nx = m->gx + 1;
  a.ny = m->gy + 1;
  a.ox = m->x0;
  a.oy = m->y0;
  a.inv_res = 1.0 / m->cs;
  a.res = (float)m->cs;
  a.zmin_map = m->zmin;
  a.zmax_map = m->zmax;
  long long ngroups = (a.n + MBES_WAVES - 1) / MBES_WAVES;
  int grid = (int)(ngroups < 65535 ? ngroups : 65535);
  if (expect_only)
    k_mbes_mesh<true><<<grid, MBES_THREADS, 0, stream>>>(a, ma);
  else
    k_mbes_mesh<false><<<grid, MBES_THREADS, 0, stream>>>(a, ma);
  return MCL_OK;
}
