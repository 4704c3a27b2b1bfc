// mcl_halfedge.h -- the half-edge table of the fan sweep over an arbitrary height-field TIN (mcl_sweep.h: sweep_side_tin),
// built on the host.  No HIP header: mcl_mesh.h (mesh_build) includes it, and `make host-asan` compiles it with plain g++
// under AddressSanitizer / UBSan, where tests/host_san/host_pure_driver.cpp checks the table's invariants and WALKS it
// on the CPU by the kernel's own rule on random meshes handed over in random order and mixed windings.
//
// One 32-byte record per half-edge h = 3 T + e (T: the triangle's number in Morton order of its xy centroid, e: its edge
// (v_e, v_e+1), every triangle taken COUNTER-CLOCKWISE in xy) -- what a walk that ENTERS T through that edge needs:
//   word 0 .. 2: x, y, z of the vertex opposite the edge (v_e+2)
//   word 3 (next_a): the half-edge on the far side of edge e+2 = (v_e+2, v_e)
//   word 4 (next_b): the half-edge on the far side of edge e+1 = (v_e+1, v_e+2)
//   word 5 .. 7: unused
// (0xffffffff: hole / ragged border, 0xfffffff0 / 0xfffffff1: the map's outer x / y border.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace halfedge {

struct Rec {   // 32 bytes, the device reads it as two 16-byte words
  uint32_t x, y, z, next_a, next_b, pad0, pad1, pad2;
};
constexpr uint32_t HOLE = 0xffffffffu, BORDER_X = 0xfffffff0u, BORDER_Y = 0xfffffff1u;

// triangles renumbered by the Morton code of their xy centroid (16 bits per axis over the bounding box; ties by input
// order): new_of_old[k] = place of input triangle k in the table
inline void morton_order(const float* verts, const uint32_t* tris, int64_t nt, double xmin, double xmax, double ymin,
                         double ymax, std::vector<uint32_t>& new_of_old) {
  auto spread = [](uint32_t v) {   // 16 bits -> every second bit of 32
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
  };
  const double qx = 65535.0 / std::max(xmax - xmin, 1e-30), qy = 65535.0 / std::max(ymax - ymin, 1e-30);
  std::vector<uint64_t> key((size_t)nt);
  for (int64_t k = 0; k < nt; ++k) {
    double cx = 0.0, cy = 0.0;
    for (int c = 0; c < 3; ++c) {
      cx += verts[3 * (size_t)tris[3 * k + c]];
      cy += verts[3 * (size_t)tris[3 * k + c] + 1];
    }
    const uint32_t ix = (uint32_t)std::min(65535.0, std::max(0.0, (cx / 3.0 - xmin) * qx));
    const uint32_t iy = (uint32_t)std::min(65535.0, std::max(0.0, (cy / 3.0 - ymin) * qy));
    key[(size_t)k] = ((uint64_t)(spread(ix) | (spread(iy) << 1)) << 32) | (uint64_t)k;
  }
  std::sort(key.begin(), key.end());
  new_of_old.resize((size_t)nt);
  for (int64_t r = 0; r < nt; ++r) new_of_old[(size_t)(key[(size_t)r] & 0xffffffffull)] = (uint32_t)r;
}

// The LOCAL tests and the adjacency: every edge has at most two triangles, their third vertices lie on opposite sides of
// it in the xy projection (no fold), no degenerate or vertical triangle.  twin[3 k + e] = 3 k2 + e2 (input numbering,
// input winding) or HOLE; ccw[k]: the triangle's xy projection is counter-clockwise as given; g2: the steepest
// triangle's squared slope.  Returns false when the mesh cannot be walked.
inline bool adjacency(const float* verts, const uint32_t* tris, int64_t nt, std::vector<uint32_t>& twin,
                      std::vector<unsigned char>& ccw, double& g2) {
  bool ok = true;
  std::unordered_map<uint64_t, int64_t> edge_first;  // undirected edge -> 3 * triangle + local edge of the first owner
  edge_first.reserve((size_t)nt * 2);
  twin.assign(3 * (size_t)nt, HOLE);
  ccw.assign((size_t)nt, 1);
  g2 = 0.0;
  for (int64_t k = 0; k < nt && ok; ++k) {
    const uint32_t v[3] = {tris[3 * k], tris[3 * k + 1], tris[3 * k + 2]};
    if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) ok = false;
    // slope of the triangle's plane
    const float* p0 = verts + 3 * (size_t)v[0];
    const float* p1 = verts + 3 * (size_t)v[1];
    const float* p2 = verts + 3 * (size_t)v[2];
    const double ax = (double)p1[0] - p0[0], ay = (double)p1[1] - p0[1], az = (double)p1[2] - p0[2];
    const double bx = (double)p2[0] - p0[0], by = (double)p2[1] - p0[1], bz = (double)p2[2] - p0[2];
    const double nz = ax * by - ay * bx, nx = ay * bz - az * by, ny = az * bx - ax * bz;
    if (nz == 0.0) ok = false; else g2 = std::max(g2, (nx * nx + ny * ny) / (nz * nz));
    ccw[(size_t)k] = nz > 0.0;
  }
  for (int64_t k = 0; k < nt && ok; ++k)
    for (int e = 0; e < 3 && ok; ++e) {
      const uint32_t a = tris[3 * k + e], b = tris[3 * k + (e + 1) % 3];
      const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = edge_first.find(key);
      if (it == edge_first.end()) {
        edge_first.emplace(key, 3 * k + e);
      } else if (it->second < 0) {
        ok = false;  // a third triangle on this edge
      } else {
        const int64_t k2 = it->second / 3;
        const int e2 = (int)(it->second % 3);
        // the two third vertices must lie on opposite sides of the edge in the xy projection (no fold)
        const float* pa = verts + 3 * (size_t)a;
        const float* pb = verts + 3 * (size_t)b;
        const float* pc = verts + 3 * (size_t)tris[3 * k + (e + 2) % 3];
        const float* pd = verts + 3 * (size_t)tris[3 * k2 + (e2 + 2) % 3];
        const double ex = (double)pb[0] - pa[0], ey = (double)pb[1] - pa[1];
        const double sc = ex * ((double)pc[1] - pa[1]) - ey * ((double)pc[0] - pa[0]);
        const double sd = ex * ((double)pd[1] - pa[1]) - ey * ((double)pd[0] - pa[0]);
        if (!(sc * sd < 0.0)) ok = false;
        twin[3 * (size_t)k + e] = (uint32_t)(3 * k2 + e2);
        twin[3 * (size_t)k2 + e2] = (uint32_t)(3 * k + e);
        it->second = -1;
      }
    }
  return ok;
}

// The table itself, from the adjacency and the order.  A clockwise input triangle (i0, i1, i2) is taken as (i0, i2, i1):
// its table edge j is its input edge 2 - j reversed, its table vertex j its input vertex (3 - j) % 3.  An edge without a
// second triangle: on the OUTER border of a rectangular map (both ends on the same side of the bounding box: BORDER_X /
// BORDER_Y -- a slice that leaves there cannot come back, mcl_sweep.h) or anywhere else (a hole, a ragged outline: HOLE).
inline void build_table(const float* verts, const uint32_t* tris, int64_t nt, const std::vector<uint32_t>& twin,
                        const std::vector<unsigned char>& ccw, const std::vector<uint32_t>& new_of_old, double xmin,
                        double xmax, double ymin, double ymax, std::vector<Rec>& he) {
  const double eb = 1e-6 * std::max(1.0, std::max(xmax - xmin, ymax - ymin));
  auto far_side = [&](int64_t k, int e) -> uint32_t {   // (input triangle k, INPUT edge e) -> half-edge in the table's numbering
    const uint32_t t = twin[3 * (size_t)k + e];
    if (t == HOLE) {
      const float* pa = verts + 3 * (size_t)tris[3 * k + e];
      const float* pb = verts + 3 * (size_t)tris[3 * k + (e + 1) % 3];
      const bool on_x = (std::fabs(pa[0] - xmin) <= eb && std::fabs(pb[0] - xmin) <= eb) ||
                        (std::fabs(pa[0] - xmax) <= eb && std::fabs(pb[0] - xmax) <= eb);
      const bool on_y = (std::fabs(pa[1] - ymin) <= eb && std::fabs(pb[1] - ymin) <= eb) ||
                        (std::fabs(pa[1] - ymax) <= eb && std::fabs(pb[1] - ymax) <= eb);
      return on_x ? BORDER_X : (on_y ? BORDER_Y : HOLE);
    }
    const uint32_t k2 = t / 3u, e2 = t % 3u;
    return 3u * new_of_old[k2] + (ccw[k2] ? e2 : 2u - e2);
  };
  he.assign(3 * (size_t)nt, Rec{0, 0, 0, HOLE, HOLE, 0, 0, 0});
  for (int64_t k = 0; k < nt; ++k) {
    const bool c = ccw[(size_t)k] != 0;
    for (int j = 0; j < 3; ++j) {   // table edge j of this triangle
      const auto tv = [&](int q) { return tris[3 * k + (c ? q % 3 : (3 - q % 3) % 3)]; };   // table vertex q -> vertex id
      const auto te = [&](int q) { return c ? q % 3 : 2 - q % 3; };                          // table edge q -> input edge
      const float* po = verts + 3 * (size_t)tv(j + 2);
      Rec r{0, 0, 0, far_side(k, te(j + 2)), far_side(k, te(j + 1)), 0, 0, 0};
      std::memcpy(&r.x, po, 4);
      std::memcpy(&r.y, po + 1, 4);
      std::memcpy(&r.z, po + 2, 4);
      he[3 * (size_t)new_of_old[(size_t)k] + j] = r;
    }
  }
}

}  // namespace halfedge
