// mcl_gridmap.h -- bathymetry map builder (include/mcl_map.h): pings at known poses -> swath point
// cloud -> height grid.  One thread per (ping, beam): fp64 pose composition (a few hundred flops per
// ping, shared by its beams through LDS), integer atomics into the node accumulators (exact, order-free).
// HBM-bound scatter: 4 B range in, optional 24 B point out, two atomics per valid beam.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "../../include/mcl_map.h"
#include "mcl_device.h"

#define GM_FIX 1048576.0  // 2^20: depth quantum of the accumulators (about 1e-6 m)

struct mcl_gridmap {
  int device = 0;
  int nx = 0, ny = 0;
  double ox = 0, oy = 0, res = 1;
  long long* sum = nullptr;   // nx*ny fixed-point depth sums
  u32* cnt = nullptr;         // nx*ny hit counts
  float* za = nullptr;        // finalize ping-pong
  float* zb = nullptr;
  u32* empty_cnt = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
};

struct GmPingArgs {
  const double* poses;   // n_pings x 6
  const float* ranges;   // n_pings x B
  const float2* beam_sc; // (sin a, cos a)
  long long n_pings;
  int B;
  float r_max;
  double m2o[12];
  double off_t[3], off_R[9];
  double ox, oy, inv_res;
  int nx, ny;
  long long* sum;
  u32* cnt;
  double* points;        // optional n_pings x B x 3
};

// one workgroup per ping: thread 0 composes the sensor pose (fp64), the beams read it from LDS
__global__ void __launch_bounds__(256) k_gm_add_pings(GmPingArgs a) {
#pragma clang fp contract(off)  // same operation order as the oracle: products and sums rounded separately
  __shared__ double sp[12];  // origin (3) + R_map_sensor (9)
  for (long long p = blockIdx.x; p < a.n_pings; p += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const double* ps = a.poses + 6 * p;
      double sr, cr, spi, cp, sy, cy;
      sincos(ps[3], &sr, &cr);
      sincos(ps[4], &spi, &cp);
      sincos(ps[5], &sy, &cy);
      const double Rp[9] = {cy * cp, cy * spi * sr - sy * cr, cy * spi * cr + sy * sr,
                            sy * cp, sy * spi * sr + cy * cr, sy * spi * cr - cy * sr,
                            -spi,    cp * sr,                 cp * cr};
      double Rmp[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          Rmp[r * 3 + c] = a.m2o[r * 4 + 0] * Rp[c] + a.m2o[r * 4 + 1] * Rp[3 + c] + a.m2o[r * 4 + 2] * Rp[6 + c];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          sp[3 + r * 3 + c] = Rmp[r * 3 + 0] * a.off_R[c] + Rmp[r * 3 + 1] * a.off_R[3 + c] + Rmp[r * 3 + 2] * a.off_R[6 + c];
      for (int r = 0; r < 3; ++r)
        sp[r] = (a.m2o[r * 4 + 0] * ps[0] + a.m2o[r * 4 + 1] * ps[1] + a.m2o[r * 4 + 2] * ps[2] + a.m2o[r * 4 + 3]) +
                (Rmp[r * 3 + 0] * a.off_t[0] + Rmp[r * 3 + 1] * a.off_t[1] + Rmp[r * 3 + 2] * a.off_t[2]);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
      const float r = a.ranges[(size_t)p * a.B + b];
      const bool ok = r > 0.f && r < a.r_max;  // NaN fails both
      double x = __builtin_nan(""), y = x, z = x;
      if (ok) {
        const float2 sc = a.beam_sc[b];
        // beam direction in the sensor frame (0, sin a, -cos a) scaled by the range
        const double dy = (double)r * (double)sc.x, dz = -(double)r * (double)sc.y;
        x = sp[0] + sp[3 + 1] * dy + sp[3 + 2] * dz;
        y = sp[1] + sp[3 + 4] * dy + sp[3 + 5] * dz;
        z = sp[2] + sp[3 + 7] * dy + sp[3 + 8] * dz;
        const double fi = floor((x - a.ox) * a.inv_res + 0.5), fj = floor((y - a.oy) * a.inv_res + 0.5);
        if (fi >= 0.0 && fj >= 0.0 && fi < (double)a.nx && fj < (double)a.ny) {
          const size_t node = (size_t)fi * a.ny + (size_t)fj;
          atomicAdd((unsigned long long*)&a.sum[node], (unsigned long long)(long long)llrint(z * GM_FIX));
          atomicAdd(&a.cnt[node], 1u);
        }
      }
      if (a.points) {
        double* o = a.points + ((size_t)p * a.B + b) * 3;
        o[0] = x;
        o[1] = y;
        o[2] = z;
      }
    }
  }
}

__global__ void k_gm_mean(const long long* __restrict__ sum, const u32* __restrict__ cnt, long long n, float* __restrict__ z,
                          u32* __restrict__ empty) {
  u32 e = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const u32 c = cnt[i];
    z[i] = c ? (float)(((double)sum[i] / GM_FIX) / (double)c) : __builtin_nanf("");
    e += c ? 0u : 1u;
  }
  e = wave_sum((int)e);
  if ((threadIdx.x & 63) == 0 && e) atomicAdd(empty, e);
}

// one Jacobi sweep: an empty node takes the mean of its non-empty 8-neighbours (fixed summation order)
__global__ void k_gm_fill(const float* __restrict__ zin, float* __restrict__ zout, int nx, int ny, u32* __restrict__ empty) {
  const long long n = (long long)nx * ny;
  u32 e = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = zin[i];
    if (v != v) {
      const int ix = (int)(i / ny), iy = (int)(i % ny);
      float s = 0.f;
      int k = 0;
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
          const int jx = ix + dx, jy = iy + dy;
          if ((dx || dy) && jx >= 0 && jy >= 0 && jx < nx && jy < ny) {
            const float w = zin[(size_t)jx * ny + jy];
            if (w == w) {
              s += w;
              ++k;
            }
          }
        }
      if (k) v = s / (float)k;
      e += k ? 0u : 1u;
    }
    zout[i] = v;
  }
  e = wave_sum((int)e);
  if ((threadIdx.x & 63) == 0 && e) atomicAdd(empty, e);
}

namespace gm_detail {
inline int fail(mcl_gridmap* g, int code, const char* msg) {
  if (g) g->err = msg;
  return code;
}
#define GMCHK(g, call)                                                                                \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      (g)->err = std::string(#call) + " failed: " + hipGetErrorString(e_);                            \
      return MCL_ERR_HIP;                                                                             \
    }                                                                                                 \
  } while (0)
static std::string g_gm_create_err;
}  // namespace gm_detail

extern "C" {

int mcl_gridmap_create(int32_t nx, int32_t ny, double ox, double oy, double res, int32_t device, mcl_gridmap** out) {
  if (!out || nx < 2 || ny < 2 || !(res > 0.0) || (long long)nx * ny > (1ll << 31)) return MCL_ERR_INVALID;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) {
    gm_detail::g_gm_create_err = "gridmap_create: no such HIP device (the map builder has no CPU fallback)";
    return MCL_ERR_NO_DEVICE;
  }
  mcl_gridmap* g = new mcl_gridmap();
  g->device = device;
  g->nx = nx;
  g->ny = ny;
  g->ox = ox;
  g->oy = oy;
  g->res = res;
  const size_t n = (size_t)nx * ny;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess ||
      hipMalloc(&g->sum, n * sizeof(long long)) != hipSuccess || hipMalloc(&g->cnt, n * sizeof(u32)) != hipSuccess ||
      hipMalloc(&g->za, n * sizeof(float)) != hipSuccess || hipMalloc(&g->zb, n * sizeof(float)) != hipSuccess ||
      hipMalloc(&g->empty_cnt, sizeof(u32)) != hipSuccess) {
    gm_detail::g_gm_create_err = "gridmap_create: device allocation failed";
    mcl_gridmap_destroy(g);
    return MCL_ERR_ALLOC;
  }
  *out = g;
  return mcl_gridmap_clear(g);
}

void mcl_gridmap_destroy(mcl_gridmap* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  void* bufs[] = {g->sum, g->cnt, g->za, g->zb, g->empty_cnt};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

const char* mcl_gridmap_last_error(const mcl_gridmap* g) { return g ? g->err.c_str() : gm_detail::g_gm_create_err.c_str(); }

int mcl_gridmap_clear(mcl_gridmap* g) {
  if (!g) return MCL_ERR_INVALID;
  GMCHK(g, hipSetDevice(g->device));
  const size_t n = (size_t)g->nx * g->ny;
  GMCHK(g, hipMemsetAsync(g->sum, 0, n * sizeof(long long), g->stream));
  GMCHK(g, hipMemsetAsync(g->cnt, 0, n * sizeof(u32), g->stream));
  GMCHK(g, hipStreamSynchronize(g->stream));
  return MCL_OK;
}

int mcl_gridmap_add_pings(mcl_gridmap* g, const double* poses6, int64_t n_pings, const float* ranges,
                          const float* beam_angles, int32_t n_beams, double r_max, const double m2o[16],
                          const double sensor_offset[6], double* points_out) {
  if (!g) return MCL_ERR_INVALID;
  if (!poses6 || !ranges || !beam_angles || n_pings < 1 || n_beams < 1 || !(r_max > 0.0))
    return gm_detail::fail(g, MCL_ERR_INVALID, "gridmap_add_pings: bad argument");
  GMCHK(g, hipSetDevice(g->device));
  const size_t nb = (size_t)n_pings * n_beams;
  double* poses_d = nullptr;
  float* ranges_d = nullptr;
  float2* sc_d = nullptr;
  double* pts_d = nullptr;
  std::vector<float2> sc((size_t)n_beams);
  for (int b = 0; b < n_beams; ++b) sc[b] = make_float2((float)std::sin((double)beam_angles[b]), (float)std::cos((double)beam_angles[b]));
  int rc = MCL_OK;
  hipError_t e = hipMalloc(&poses_d, sizeof(double) * 6 * (size_t)n_pings);
  if (e == hipSuccess) e = hipMalloc(&ranges_d, sizeof(float) * nb);
  if (e == hipSuccess) e = hipMalloc(&sc_d, sizeof(float2) * (size_t)n_beams);
  if (e == hipSuccess && points_out) e = hipMalloc(&pts_d, sizeof(double) * 3 * nb);
  if (e != hipSuccess) {
    g->err = "gridmap_add_pings: device allocation failed";
    rc = MCL_ERR_ALLOC;
  }
  if (rc == MCL_OK) {
    e = hipMemcpyAsync(poses_d, poses6, sizeof(double) * 6 * (size_t)n_pings, hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ranges_d, ranges, sizeof(float) * nb, hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sc_d, sc.data(), sizeof(float2) * (size_t)n_beams, hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) {
      static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      static const double zero6[6] = {0, 0, 0, 0, 0, 0};
      const double* M = m2o ? m2o : ident;
      const double* so = sensor_offset ? sensor_offset : zero6;
      GmPingArgs a;
      a.poses = poses_d;
      a.ranges = ranges_d;
      a.beam_sc = sc_d;
      a.n_pings = n_pings;
      a.B = n_beams;
      a.r_max = (float)r_max;
      for (int q = 0; q < 12; ++q) a.m2o[q] = M[q];
      for (int q = 0; q < 3; ++q) a.off_t[q] = so[q];
      rot_rpy(so[3], so[4], so[5], a.off_R);
      a.ox = g->ox;
      a.oy = g->oy;
      a.inv_res = 1.0 / g->res;
      a.nx = g->nx;
      a.ny = g->ny;
      a.sum = g->sum;
      a.cnt = g->cnt;
      a.points = pts_d;
      const unsigned blocks = (unsigned)std::min<long long>(n_pings, 65535);
      k_gm_add_pings<<<blocks, 256, 0, g->stream>>>(a);
      e = hipGetLastError();
    }
    if (e == hipSuccess && points_out) e = hipMemcpyAsync(points_out, pts_d, sizeof(double) * 3 * nb, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) {
      g->err = std::string("gridmap_add_pings: ") + hipGetErrorString(e);
      rc = MCL_ERR_HIP;
    }
  }
  if (poses_d) (void)hipFree(poses_d);
  if (ranges_d) (void)hipFree(ranges_d);
  if (sc_d) (void)hipFree(sc_d);
  if (pts_d) (void)hipFree(pts_d);
  return rc;
}

int mcl_gridmap_finalize(mcl_gridmap* g, int32_t fill_passes, float* z_out, int64_t* n_empty, uint32_t* counts_out) {
  if (!g) return MCL_ERR_INVALID;
  if (!z_out || fill_passes < 0) return gm_detail::fail(g, MCL_ERR_INVALID, "gridmap_finalize: bad argument");
  GMCHK(g, hipSetDevice(g->device));
  const long long n = (long long)g->nx * g->ny;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  GMCHK(g, hipMemsetAsync(g->empty_cnt, 0, sizeof(u32), g->stream));
  k_gm_mean<<<blocks, 256, 0, g->stream>>>(g->sum, g->cnt, n, g->za, g->empty_cnt);
  float* cur = g->za;
  float* nxt = g->zb;
  u32 empty = 0;
  GMCHK(g, hipMemcpyAsync(&empty, g->empty_cnt, sizeof(u32), hipMemcpyDeviceToHost, g->stream));
  GMCHK(g, hipStreamSynchronize(g->stream));
  for (int p = 0; p < fill_passes && empty > 0; ++p) {
    GMCHK(g, hipMemsetAsync(g->empty_cnt, 0, sizeof(u32), g->stream));
    k_gm_fill<<<blocks, 256, 0, g->stream>>>(cur, nxt, g->nx, g->ny, g->empty_cnt);
    GMCHK(g, hipMemcpyAsync(&empty, g->empty_cnt, sizeof(u32), hipMemcpyDeviceToHost, g->stream));
    GMCHK(g, hipStreamSynchronize(g->stream));
    std::swap(cur, nxt);
  }
  GMCHK(g, hipGetLastError());
  GMCHK(g, hipMemcpyAsync(z_out, cur, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, g->stream));
  if (counts_out) GMCHK(g, hipMemcpyAsync(counts_out, g->cnt, sizeof(u32) * (size_t)n, hipMemcpyDeviceToHost, g->stream));
  GMCHK(g, hipStreamSynchronize(g->stream));
  if (n_empty) *n_empty = (int64_t)empty;
  return MCL_OK;
}

}  // extern "C"
