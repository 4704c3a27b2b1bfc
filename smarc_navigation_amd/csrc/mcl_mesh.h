// mcl_mesh.h -- triangle-mesh bathymetry: host-side build of the acceleration structure that
// k_mbes_cast<1,*> (mcl_mbes.h) traverses.
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
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mcl.h"
#include "mcl_halfedge.h"
#include "mcl_mbes.h"

struct MeshDev {
  float4* tri = nullptr;       // 3 float4 per (cell, triangle) record, plane form (see mesh_build)
  float4* tri_mt = nullptr;    // Moller-Trumbore form (v0, e1, e2), only if some triangle is near-vertical
  u32* cell_start = nullptr;   // gx*gy + 1
  uint2* cell_info = nullptr;  // per cell: x = half2(zmin rounded down, zmax rounded up), y = start | count << 27
  int gx = 0, gy = 0;
  double x0 = 0, y0 = 0, cs = 1;
  float zmin = 0, zmax = 0;
  size_t n_records = 0;
  // structured mesh (a triangulated regular height grid): node heights, diagonal bit in the LSB
  float* heights = nullptr;  // (gx+1)*(gy+1), or nullptr if the mesh is not structured
  float* heights_pad = nullptr;  // the same inside a one-node ring of NaNs, (gx+3)*(gy+3) (fan sweep: MbesArgs::grid_pad)
  size_t n_vertical = 0;  // triangles whose xy projection is degenerate (cannot be a height field)
  int diag_mode = 0;      // structured: 1 = every cell split along 00-11, 2 = along 10-01, 0 = mixed (LSB per cell)
  double slope_max = 0;   // steepest triangle, |grad h| (the fan sweep's tilt bound, mcl_sweep.h)
  // triangle adjacency for the fan sweep over an arbitrary height-field TIN (mcl_sweep.h: sweep_side_tin): ONE 32-byte
  // record per half-edge h = 3 T + e (T: the triangle's number in Morton order of its xy centroid, e: its edge
  // (v_e, v_e+1), every triangle taken counter-clockwise in xy) -- what a walk that ENTERS T through that edge needs:
  //   {x, y, z of the vertex opposite the edge (v_e+2),
  //    next_a: the half-edge on the far side of edge e+2 = (v_e+2, v_e), next_b: of edge e+1 = (v_e+1, v_e+2), -, -, -}
  // (0xffffffff: hole / ragged border, 0xfffffff0 / 0xfffffff1: the map's outer x / y border).  One dependent load per
  // step of the walk, no vertex ids: which of the two exits a slice takes is decided by the side of the plane the new
  // vertex lies on (rounds 3-5: a 32-byte triangle record and then the vertex it named, 16 bytes, a second dependent
  // load; input order -- the walk's neighbours megabytes apart when the caller's triangles come in no spatial order).
  // tin_ok: every edge has at most two triangles and their third vertices lie on opposite sides of it in the xy
  // projection (no fold: the mesh is a height field), no vertical triangle, no two triangles overlapping in xy.
  uint4* tin_he = nullptr;
  size_t tin_he_bytes = 0;
  bool tin_ok = false;
  size_t tin_nhe = 0;       // half-edge records (3 x triangles); behind them rim records, behind those chunk records (mcl_halfedge.h)
  size_t tin_outline = 0;   // first rim record of the OUTLINE when it is linked (a sensor beyond it starts its walk there), else 0
  size_t tin_rims = 0;      // rim records behind the 3 nt half-edge records: the edges of the holes the walk crosses by itself (mcl_halfedge.h: link_holes)
  u32* cell_rim = nullptr;   // per cell of the cell grid: the first rim record of the linked hole whose bounding box reaches into the cell (0xffffffff: none, 0xfffffffe: more than one) -- where a walk starts whose nadir ray falls into a gap
  bool tin_holes = false;   // some edge of the TIN has no triangle on its far side and does not lie on the bounding box: a hole or a ragged outline (walks that reach it hand their particle over)
  // fan slice over an arbitrary triangle soup (mcl_slice.h): per (cell, triangle) record the three vertices of its source
  // triangle in MAP-FRAME coordinates, 3 float4 {x, y, z, -}, same indexing as `tri`.  (Absolute, not cell-relative: a
  // vertex has the same bits in every record it appears in, so the slices of two triangles that share an edge meet in
  // one point.  No index indirection: the walk over a cell's records is one dependent load, not two.)
  float4* cell_tri = nullptr;
};

// The height array of a lattice map inside a one-node ring of quiet NaNs whose payload names the border: 1 = beyond an
// x side (i = -1 or nx), 2 = beyond a y side, 3 = a corner.  The fan sweep (mcl_sweep.h) walks node by node: stepping
// off the map it loads a NaN, which ends the walk through the test that ends it anyway -- no bounds test per step.
inline hipError_t upload_padded_heights(const float* z, int nx, int ny, float** out) {
  const size_t nyp = (size_t)ny + 2, cnt = ((size_t)nx + 2) * nyp;
  std::vector<float> pad(cnt);
  auto nanp = [](uint32_t payload) {
    const uint32_t b = 0x7fc00000u | payload;
    float f;
    memcpy(&f, &b, 4);
    return f;
  };
  for (size_t i = 0; i < (size_t)nx + 2; ++i)
    for (size_t j = 0; j < nyp; ++j) {
      const bool ox = i == 0 || i == (size_t)nx + 1, oy = j == 0 || j == nyp - 1;
      pad[i * nyp + j] = (ox || oy) ? nanp((ox ? 1u : 0u) | (oy ? 2u : 0u)) : z[(i - 1) * (size_t)ny + (j - 1)];
    }
  *out = nullptr;
  hipError_t e = hipMalloc(out, sizeof(float) * cnt);
  if (e != hipSuccess) return e;
  return hipMemcpy(*out, pad.data(), sizeof(float) * cnt, hipMemcpyHostToDevice);
}

inline void mesh_free(MeshDev* m) {
  if (!m) return;
  if (m->tri) (void)hipFree(m->tri);
  if (m->cell_start) (void)hipFree(m->cell_start);
  if (m->cell_info) (void)hipFree(m->cell_info);
  if (m->tri_mt) (void)hipFree(m->tri_mt);
  if (m->heights) (void)hipFree(m->heights);
  if (m->heights_pad) (void)hipFree(m->heights_pad);
  if (m->tin_he) (void)hipFree(m->tin_he);
  if (m->cell_rim) (void)hipFree(m->cell_rim);
  if (m->cell_tri) (void)hipFree(m->cell_tri);
  delete m;
}

// float -> IEEE half bits with directed rounding (dir < 0: toward -inf, dir > 0: toward +inf)
inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal
      float f = (float)man * 5.9604644775390625e-08f;  // 2^-24
      memcpy(&bits, &f, 4);
      bits |= sign;
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &bits, 4);
  return out;
}
inline uint16_t float_to_half_dir(float x, int dir) {
  if (x != x) return 0x7e00u;
  if (x > 65504.f) return dir > 0 ? 0x7c00u : 0x7bffu;    // +inf / max finite
  if (x < -65504.f) return dir < 0 ? 0xfc00u : 0xfbffu;
  // round to nearest first (via truncation of the magnitude), then fix the direction
  uint32_t b;
  memcpy(&b, &x, 4);
  const uint32_t sign = (b >> 16) & 0x8000u;
  const float ax = std::fabs(x);
  uint16_t h;
  if (ax < 6.103515625e-05f) {  // subnormal half range
    h = (uint16_t)(ax * 16777216.0f);  // floor(ax / 2^-24)
  } else {
    uint32_t ab;
    memcpy(&ab, &ax, 4);
    const uint32_t exp = ((ab >> 23) & 0xffu) - 112u, man = (ab >> 13) & 0x3ffu;
    h = (uint16_t)((exp << 10) | man);  // magnitude truncated toward zero
  }
  h |= (uint16_t)sign;
  // h now is x rounded toward zero; step outward if the direction requires it
  float back = half_bits_to_float(h);
  if (dir > 0 && back < x) h = (h & 0x8000u) ? (uint16_t)(h - 1) : (uint16_t)(h + 1);
  if (dir < 0 && back > x) h = (h & 0x8000u) ? (uint16_t)(h + 1) : (uint16_t)(h - 1);
  if (dir > 0 && x > 0.f && back < x && (h & 0x7fffu) == 0x7c00u) h = 0x7c00u;
  // crossing zero: rounding a tiny negative toward +inf gives -0 -> fine; tiny positive toward -inf gives +0
  if (dir > 0 && x < 0.f && back < x && h == 0x7fffu) h = 0x8000u;
  return h;
}

inline int mesh_build(const float* verts, int64_t nv, const uint32_t* tris, int64_t nt, bool want_slice, MeshDev** out,
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
  // exact binning: a triangle is recorded in a cell only if its xy projection really overlaps the cell
  // (separating-axis test: the cell's axes are covered by cell_range, the triangle's three edge normals here).
  // On an irregular TIN the bbox of a triangle touches up to 3 x 3 cells but the triangle itself only 2-3:
  // ~10 -> ~4 records per cell.
  auto overlaps = [&](int64_t k, int a, int b) {
    const double e = 1e-7 * cs;
    const double bx0 = m->x0 + a * cs + e, bx1 = m->x0 + (a + 1) * cs - e, by0 = m->y0 + b * cs + e, by1 = m->y0 + (b + 1) * cs - e;
    double px[3], py[3];
    for (int c = 0; c < 3; ++c) {
      px[c] = verts[3 * (size_t)tris[3 * k + c]];
      py[c] = verts[3 * (size_t)tris[3 * k + c] + 1];
    }
    for (int c = 0; c < 3; ++c) {
      const int d = (c + 1) % 3, o = (c + 2) % 3;
      const double nx = -(py[d] - py[c]), ny = px[d] - px[c];  // normal of edge c -> d
      double side = nx * (px[o] - px[c]) + ny * (py[o] - py[c]);  // the third vertex is on this side
      if (side == 0.0) continue;                                   // degenerate projection: bbox binning decides
      const double sgn = side > 0.0 ? 1.0 : -1.0;
      // the box corner farthest towards the triangle's side
      const double cxm = (nx * sgn > 0.0) ? bx1 : bx0, cym = (ny * sgn > 0.0) ? by1 : by0;
      if (sgn * (nx * (cxm - px[c]) + ny * (cym - py[c])) < 0.0) return false;  // the whole box is outside this edge
    }
    return true;
  };
  std::vector<u32> start(nc + 1, 0u);
  for (int64_t k = 0; k < nt; ++k) {
    int a0, a1, b0, b1;
    cell_range(k, a0, a1, b0, b1);
    bool any = false;
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b)
        if (overlaps(k, a, b)) {
          start[(size_t)a * m->gy + b + 1]++;
          any = true;
        }
    if (!any) start[(size_t)a0 * m->gy + b0 + 1]++;  // (a sliver thinner than the tolerance: keep it somewhere)
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
  if (m->n_records >= (1ull << 27)) {
    *err = "set_map_mesh: more than 2^27 (cell, triangle) records";
    delete m;
    return MCL_ERR_UNSUPPORTED;
  }
  std::vector<float4> rec(3 * std::max<size_t>(m->n_records, 1)), rec_mt;
  std::vector<float2> cz(nc, make_float2(INFINITY, -INFINITY));
  std::vector<u32> fill(nc, 0u);
  std::vector<u32> rec_tri(std::max<size_t>(m->n_records, 1), 0u);  // source triangle of every record (overlap test below)
  bool any_vertical = false;
  for (int64_t k = 0; k < nt && !any_vertical; ++k) {
    const float* v0 = verts + 3 * (size_t)tris[3 * k];
    const float* v1 = verts + 3 * (size_t)tris[3 * k + 1];
    const float* v2 = verts + 3 * (size_t)tris[3 * k + 2];
    const double ax = (double)v1[0] - v0[0], ay = (double)v1[1] - v0[1], az = (double)v1[2] - v0[2];
    const double bx = (double)v2[0] - v0[0], by = (double)v2[1] - v0[1], bz = (double)v2[2] - v0[2];
    const double nz = ax * by - ay * bx, nx = ay * bz - az * by, ny = az * bx - ax * bz;
    if (std::fabs(nz) <= 1e-4 * std::sqrt(nx * nx + ny * ny + nz * nz)) any_vertical = true;
  }
  if (any_vertical) rec_mt.resize(rec.size());
  // ---- locality pass for the adjacency walk (mcl_sweep.h: sweep_side_tin): triangles renumbered by the Morton code of
  // their xy centroid (16 bits per axis over the bounding box; ties by input order), so that the records a walk loads
  // one after the other -- and the walks of a wave's 64 neighbouring particles -- lie in the cache lines their
  // neighbours have just touched, whatever order the caller's arrays come in.  (96 nt < 2^31: the table is read
  // through a raw buffer with 32-bit byte offsets.)
  const bool want_tin = !any_vertical && nt > 0 && nt < (1ll << 31) / 96 && nv < (1ll << 31);
  std::vector<u32> new_of_old;
  if (want_tin) halfedge::morton_order(verts, tris, nt, xmin, xmax, ymin, ymax, new_of_old);
  for (int64_t k = 0; k < nt; ++k) {
    int a0, a1, b0, b1;
    cell_range(k, a0, a1, b0, b1);
    const float* v0 = verts + 3 * (size_t)tris[3 * k];
    const float* v1 = verts + 3 * (size_t)tris[3 * k + 1];
    const float* v2 = verts + 3 * (size_t)tris[3 * k + 2];
    const float tz0 = std::min(v0[2], std::min(v1[2], v2[2])), tz1 = std::max(v0[2], std::max(v1[2], v2[2]));
    const double ax = (double)v1[0] - v0[0], ay = (double)v1[1] - v0[1], az = (double)v1[2] - v0[2];
    const double bx = (double)v2[0] - v0[0], by = (double)v2[1] - v0[1], bz = (double)v2[2] - v0[2];
    const double nz = ax * by - ay * bx, nx = ay * bz - az * by, ny = az * bx - ax * bz;
    const double nlen = std::sqrt(nx * nx + ny * ny + nz * nz);
    const bool vertical = std::fabs(nz) <= 1e-4 * nlen;
    if (vertical && (tz1 - tz0) > 1e-6) m->n_vertical++;
    bool any_cell = false;
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b) any_cell = any_cell || overlaps(k, a, b);
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b) {
        if (any_cell ? !overlaps(k, a, b) : !(a == a0 && b == b0)) continue;
        const size_t c = (size_t)a * m->gy + b;
        const size_t r = 3 * ((size_t)start[c] + fill[c]++);
        rec_tri[r / 3] = (u32)k;
        const double cx = m->x0 + a * cs, cy = m->y0 + b * cs;
        const double lx = v0[0] - cx, ly = v0[1] - cy;  // v0 relative to the cell corner
        if (!vertical) {
          // plane z = pd - px*x - py*y (cell-local x, y); barycentrics (u, v) = M (x - lx, y - ly)
          const double px = nx / nz, py = ny / nz, pd = (double)v0[2] + px * lx + py * ly;
          const double det = ax * by - ay * bx;
          rec[r + 0] = make_float4((float)px, (float)py, (float)pd, 0.f);
          rec[r + 1] = make_float4((float)lx, (float)ly, (float)(by / det), (float)(-bx / det));
          float tid;  // (the source triangle's index rides in the record's spare word: sweep_side_tin's start)
          const u32 k32 = want_tin ? new_of_old[(size_t)k] : (u32)k;   // (its number in the half-edge table's order)
          memcpy(&tid, &k32, 4);
          rec[r + 2] = make_float4((float)(-ay / det), (float)(ax / det), tid, 0.f);
        } else {
          rec[r + 0] = make_float4(0.f, 0.f, 0.f, 1.f);  // flag: use the Moller-Trumbore record
          rec[r + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
          rec[r + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (any_vertical) {
          rec_mt[r + 0] = make_float4((float)lx, (float)ly, v0[2], 0.f);
          rec_mt[r + 1] = make_float4((float)ax, (float)ay, (float)az, 0.f);
          rec_mt[r + 2] = make_float4((float)bx, (float)by, (float)bz, 0.f);
        }
        cz[c].x = std::min(cz[c].x, tz0);
        cz[c].y = std::max(cz[c].y, tz1);
      }
  }
  std::vector<uint2> info(nc);
  for (size_t c = 0; c < nc; ++c) {
    const u32 cnt = start[c + 1] - start[c];
    const uint16_t hlo = float_to_half_dir(cz[c].x, -1), hhi = float_to_half_dir(cz[c].y, +1);
    info[c].x = (u32)hlo | ((u32)hhi << 16);
    info[c].y = start[c] | (std::min(cnt, 31u) << 27);  // count 31 = "31 or more": read cell_start
  }
  hipError_t e1 = hipMalloc(&m->tri, sizeof(float4) * rec.size());
  hipError_t e2 = hipMalloc(&m->cell_start, sizeof(u32) * (nc + 1));
  hipError_t e3 = hipMalloc(&m->cell_info, sizeof(uint2) * nc);
  hipError_t e4 = any_vertical ? hipMalloc(&m->tri_mt, sizeof(float4) * rec_mt.size()) : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    *err = "set_map_mesh: device allocation failed";
    mesh_free(m);
    return MCL_ERR_ALLOC;
  }
  if (hipMemcpy(m->tri, rec.data(), sizeof(float4) * rec.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->cell_start, start.data(), sizeof(u32) * (nc + 1), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->cell_info, info.data(), sizeof(uint2) * nc, hipMemcpyHostToDevice) != hipSuccess ||
      (any_vertical &&
       hipMemcpy(m->tri_mt, rec_mt.data(), sizeof(float4) * rec_mt.size(), hipMemcpyHostToDevice) != hipSuccess)) {
    *err = "set_map_mesh: upload failed";
    mesh_free(m);
    return MCL_ERR_HIP;
  }
  // ---- triangle adjacency (fan sweep over a TIN): edge -> the (at most two) triangles on it
  if (want_tin) {
    // (the local tests, the adjacency and the table itself: mcl_halfedge.h -- device-free, also compiled and walked on the
    //  CPU under the host sanitizers)
    std::vector<u32> twin;
    std::vector<unsigned char> ccw;
    double g2 = 0.0;
    bool ok = halfedge::adjacency(verts, tris, nt, twin, ccw, g2);
    // ---- GLOBAL single-valuedness.  The edge tests above are local: two sheets that overlap in xy without sharing an
    // edge (a seabed and a wreck floating above it) pass them, and a walk by adjacency would never meet the upper
    // sheet.  Two triangles whose xy projections overlap with positive area are both recorded in some common cell
    // (exact binning above), so it is enough to test the pairs of every cell: separating-axis test on the six edge
    // normals; projections that merely touch (shared edge or vertex) count as separated.
    if (ok) {
      const double eps = 1e-9 * cs * cs;
      auto proj_overlap = [&](u32 ka, u32 kb) {
        double P[2][3][2];
        const u32 kk[2] = {ka, kb};
        for (int t = 0; t < 2; ++t)
          for (int c = 0; c < 3; ++c) {
            P[t][c][0] = verts[3 * (size_t)tris[3 * (size_t)kk[t] + c]];
            P[t][c][1] = verts[3 * (size_t)tris[3 * (size_t)kk[t] + c] + 1];
          }
        for (int t = 0; t < 2; ++t)
          for (int c = 0; c < 3; ++c) {
            const int d = (c + 1) % 3;
            double ax = -(P[t][d][1] - P[t][c][1]), ay = P[t][d][0] - P[t][c][0];
            const double len = std::sqrt(ax * ax + ay * ay);
            if (!(len > 0.0)) continue;
            ax /= len;
            ay /= len;
            double lo[2], hi[2];
            for (int u = 0; u < 2; ++u) {
              lo[u] = INFINITY;
              hi[u] = -INFINITY;
              for (int q = 0; q < 3; ++q) {
                const double pr = ax * P[u][q][0] + ay * P[u][q][1];
                lo[u] = std::min(lo[u], pr);
                hi[u] = std::max(hi[u], pr);
              }
            }
            // (eps is an area; the axis is a unit vector, so compare lengths against eps / cs)
            if (hi[0] <= lo[1] + eps / cs || hi[1] <= lo[0] + eps / cs) return false;  // separated (or touching)
          }
        return true;
      };
      for (size_t c = 0; c < nc && ok; ++c) {
        const u32 rs = start[c], re = start[c + 1];
        if (re - rs > 256u) {  // pathological pile-up in one cell: not worth proving anything
          ok = false;
          break;
        }
        for (u32 i = rs; i < re && ok; ++i)
          for (u32 j = i + 1; j < re && ok; ++j)
            if (rec_tri[i] != rec_tri[j] && proj_overlap(rec_tri[i], rec_tri[j])) ok = false;
      }
    }
    if (ok) {
      std::vector<halfedge::Rec> he;
      halfedge::build_table(verts, tris, nt, twin, ccw, new_of_old, xmin, xmax, ymin, ymax, he);
      static_assert(sizeof(halfedge::Rec) == 2 * sizeof(uint4), "the device reads a half-edge record as two 16-byte words");
      const bool link = !(getenv("MCL_TIN_RIMS") && atoi(getenv("MCL_TIN_RIMS")) == 0);   // (0: no hole is crossed -- A/B, tests)
      const bool box_outline = getenv("MCL_TIN_BOX_OUTLINE") && atoi(getenv("MCL_TIN_BOX_OUTLINE")) == 1;   // (mcl_halfedge.h: link an outline that lies on the bounding box all around as well -- INTEGRATION.md 4a)
      const halfedge::Links links = link ? halfedge::link_holes(he, nt, box_outline) : halfedge::Links();
      m->tin_rims = links.nrim;
      m->tin_nhe = 3 * (size_t)nt;
      m->tin_outline = links.outline ? links.outline_base : 0;
      // (does a walk ever meet an edge with nothing behind it that is not on the bounding box -- a rim record or the HOLE
      //  code?  Only such meshes stage their hand-overs through the fan slice, mcl_host_update.h; an outline on the box all
      //  around, linked for sensors beyond it, is not one)
      m->tin_holes = false;
      for (size_t q = 0; q < 3 * (size_t)nt && !m->tin_holes; ++q)
        m->tin_holes = (he[q].next_a >= 3 * (size_t)nt && he[q].next_a < 0xfffffff0u) || he[q].next_a == halfedge::HOLE ||
                       (he[q].next_b >= 3 * (size_t)nt && he[q].next_b < 0xfffffff0u) || he[q].next_b == halfedge::HOLE;
      std::vector<uint32_t> cell_rim;
      if (m->tin_rims) {
        // which hole may lie under a sensor: every cell the hole's bounding box reaches into names the hole's first rim
        // record (the walk itself decides by parity whether its nadir ray goes through THAT hole, mcl_sweep.h)
        cell_rim.assign(nc, 0xffffffffu);
        for (size_t k = 3 * (size_t)nt; k < 3 * (size_t)nt + links.nrim; k += he[k].pad1) {
          if (he[k].pad2 & halfedge::RIM_EXTERIOR) continue;   // (the outline: a sensor beyond it is off the mesh)
          float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f;
          for (size_t q = k; q < k + he[k].pad1; ++q) {
            float x, y;
            memcpy(&x, &he[q].x, 4);
            memcpy(&y, &he[q].y, 4);
            bx0 = std::min(bx0, x), bx1 = std::max(bx1, x), by0 = std::min(by0, y), by1 = std::max(by1, y);
          }
          const int i0 = std::max(0, (int)std::floor((bx0 - m->x0) / m->cs) - 1), i1 = std::min(m->gx - 1, (int)std::floor((bx1 - m->x0) / m->cs) + 1);
          const int j0 = std::max(0, (int)std::floor((by0 - m->y0) / m->cs) - 1), j1 = std::min(m->gy - 1, (int)std::floor((by1 - m->y0) / m->cs) + 1);
          for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j) {
              uint32_t& c = cell_rim[(size_t)i * m->gy + j];
              c = c == 0xffffffffu ? (uint32_t)k : 0xfffffffeu;
            }
        }
      }
      m->tin_he_bytes = sizeof(halfedge::Rec) * he.size();
      if (hipMalloc(&m->tin_he, m->tin_he_bytes) == hipSuccess &&
          hipMemcpy(m->tin_he, he.data(), m->tin_he_bytes, hipMemcpyHostToDevice) == hipSuccess &&
          (cell_rim.empty() || (hipMalloc(&m->cell_rim, 4 * cell_rim.size()) == hipSuccess &&
                                hipMemcpy(m->cell_rim, cell_rim.data(), 4 * cell_rim.size(), hipMemcpyHostToDevice) == hipSuccess))) {
        m->tin_ok = true;
        m->slope_max = std::sqrt(g2);
      } else {
        (void)hipGetLastError();
      }
    }
  }
  // ---- structured-mesh detection: is this exactly a regular height grid with every cell split into
  // two triangles along one of its diagonals?  Then node heights + one diagonal bit per cell describe
  // it completely and the cast kernel can test the two planes from LDS (k_mbes_cast<2,*>).
  {
    const size_t nnx = (size_t)m->gx + 1, nny = (size_t)m->gy + 1;
    bool ok = (size_t)nt == 2 * nc && (size_t)nv == nnx * nny;
    std::vector<float> hts;
    std::vector<int> node_of;
    if (ok) {
      hts.assign(nnx * nny, NAN);
      node_of.assign((size_t)nv, -1);
      for (int64_t i = 0; i < nv && ok; ++i) {
        const double fx = ((double)verts[3 * i] - m->x0) / cs, fy = ((double)verts[3 * i + 1] - m->y0) / cs;
        const double rx = std::round(fx), ry = std::round(fy);
        if (std::fabs(fx - rx) > 1e-4 || std::fabs(fy - ry) > 1e-4 || rx < 0 || ry < 0 || rx >= nnx || ry >= nny) {
          ok = false;
          break;
        }
        const size_t node = (size_t)rx * nny + (size_t)ry;
        if (hts[node] == hts[node]) ok = false;  // two vertices on one node
        hts[node] = verts[3 * i + 2];
        node_of[i] = (int)node;
      }
    }
    std::vector<unsigned char> diag, seen;
    if (ok) {
      diag.assign(nc, 0);
      seen.assign(nc, 0);  // bit 0: lower/first triangle seen, bit 1: the other
      for (int64_t k = 0; k < nt && ok; ++k) {
        int ixs[3], iys[3];
        for (int c = 0; c < 3; ++c) {
          const int node = node_of[tris[3 * k + c]];
          ixs[c] = node / (int)nny;
          iys[c] = node % (int)nny;
        }
        const int cx = std::min(ixs[0], std::min(ixs[1], ixs[2])), cy = std::min(iys[0], std::min(iys[1], iys[2]));
        if (cx >= m->gx || cy >= m->gy) {
          ok = false;
          break;
        }
        int mask = 0;  // which corners: bit (dx + 2*dy)
        for (int c = 0; c < 3; ++c) {
          const int dx = ixs[c] - cx, dy = iys[c] - cy;
          if (dx > 1 || dy > 1) ok = false;
          mask |= 1 << (dx + 2 * dy);
        }
        if (!ok) break;
        // corners: 1 = (0,0), 2 = (1,0), 4 = (0,1), 8 = (1,1)
        const size_t c = (size_t)cx * m->gy + cy;
        int d, half;
        if (mask == (1 | 2 | 8)) { d = 0; half = 1; }        // 00,10,11  (v <= u)
        else if (mask == (1 | 8 | 4)) { d = 0; half = 2; }   // 00,11,01  (v >= u)
        else if (mask == (1 | 2 | 4)) { d = 1; half = 1; }   // 00,10,01  (u + v <= 1)
        else if (mask == (2 | 8 | 4)) { d = 1; half = 2; }   // 10,11,01  (u + v >= 1)
        else { ok = false; break; }
        if (seen[c] && diag[c] != d) ok = false;
        if (seen[c] & half) ok = false;
        diag[c] = (unsigned char)d;
        seen[c] |= (unsigned char)half;
      }
      for (size_t c = 0; c < nc && ok; ++c)
        if (seen[c] != 3) ok = false;
    }
    if (ok) {
      size_t n1 = 0;
      for (size_t c = 0; c < nc; ++c) n1 += diag[c];
      m->diag_mode = n1 == 0 ? 1 : (n1 == nc ? 2 : 0);
      if (m->diag_mode == 0)  // mixed: the cell's diagonal rides in the LSB of its (0,0) corner height (a 1-ulp change)
        for (size_t ix = 0; ix < nnx; ++ix)
          for (size_t iy = 0; iy < nny; ++iy) {
            u32 b;
            memcpy(&b, &hts[ix * nny + iy], 4);
            const u32 dbit = (ix < (size_t)m->gx && iy < (size_t)m->gy) ? diag[ix * m->gy + iy] : 0u;
            b = (b & ~1u) | dbit;
            memcpy(&hts[ix * nny + iy], &b, 4);
          }
      // steepest triangle: the two triangles of a cell take their x slope from one x edge and their y slope from
      // one y edge of the cell, in the four combinations either triangulation can produce
      {
        double g2 = 0.0;
        for (size_t ix = 0; ix + 1 < nnx; ++ix)
          for (size_t iy = 0; iy + 1 < nny; ++iy) {
            const double h00 = hts[ix * nny + iy], h10 = hts[(ix + 1) * nny + iy], h01 = hts[ix * nny + iy + 1],
                         h11 = hts[(ix + 1) * nny + iy + 1];
            const double ax = std::max(std::fabs(h10 - h00), std::fabs(h11 - h01));
            const double ay = std::max(std::fabs(h01 - h00), std::fabs(h11 - h10));
            g2 = std::max(g2, ax * ax + ay * ay);
          }
        m->slope_max = std::sqrt(g2) / cs;
      }
      if (hipMalloc(&m->heights, sizeof(float) * hts.size()) != hipSuccess ||
          hipMemcpy(m->heights, hts.data(), sizeof(float) * hts.size(), hipMemcpyHostToDevice) != hipSuccess ||
          upload_padded_heights(hts.data(), (int)nnx, (int)nny, &m->heights_pad) != hipSuccess) {
        *err = "set_map_mesh: device allocation failed";
        mesh_free(m);
        return MCL_ERR_ALLOC;
      }
    }
  }
  // ---- fan slice (mcl_slice.h): the three vertices of every record's source triangle, map frame -- 48 B per (cell,
  // triangle) record, > 100 MB for a million triangles.  Only the fan slice reads them, and it never casts a structured
  // mesh (the lattice sweep / the node-height traversal do) unless the caller forces the general path: built for
  // everything else.  It is an optional accelerator: if its allocation fails the map still loads and the ray
  // traversal casts what the slice would have (cell_tri == nullptr disables the slice, mcl_host_update.h).
  if (!m->heights || want_slice) {
    std::vector<float4> ct(3 * std::max<size_t>(m->n_records, 1));
    for (size_t r = 0; r < m->n_records; ++r) {
      const u32 k = rec_tri[r];
      for (int c = 0; c < 3; ++c) {
        const float* v = verts + 3 * (size_t)tris[3 * (size_t)k + c];
        ct[3 * r + c] = make_float4(v[0], v[1], v[2], 0.f);
      }
      memcpy(&ct[3 * r].w, &k, 4);   // the source triangle's index in the first vertex's spare word (k_mbes_slice_group: a triangle recorded in several cells is staged once)
    }
    if (hipMalloc(&m->cell_tri, sizeof(float4) * ct.size()) != hipSuccess) {
      (void)hipGetLastError();
      m->cell_tri = nullptr;
    } else if (hipMemcpy(m->cell_tri, ct.data(), sizeof(float4) * ct.size(), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(m->cell_tri);
      m->cell_tri = nullptr;
    }
  }
  *out = m;
  return MCL_OK;
}

inline MeshArgs mesh_args(const MeshDev* m) {
  MeshArgs ma;
  ma.tri = m->tri;
  ma.tri_mt = m->tri_mt;
  ma.cell_start = m->cell_start;
  ma.cell_info = m->cell_info;
  ma.gx = m->gx;
  ma.gy = m->gy;
  ma.cs = (float)m->cs;
  ma.tin_he = m->tin_he;
  ma.tin_he_bytes = (u32)m->tin_he_bytes;
  ma.cell_rim = m->cell_rim;
  ma.tin_nhe = (u32)m->tin_nhe;
  ma.tin_outline = (u32)m->tin_outline;
  ma.cell_tri = m->cell_tri;
  ma.x0 = m->x0;
  ma.y0 = m->y0;
  return ma;
}
