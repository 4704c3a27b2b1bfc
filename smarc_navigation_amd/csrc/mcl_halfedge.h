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
//
// HOLES THE WALK CAN CROSS (link_holes): behind the 3 nt half-edge records follow RIM records, one per edge of every small
// closed hole -- a data gap of a survey: a boundary loop that runs clockwise around empty space, touches the map's outer
// border nowhere and has no piece of mesh inside.  The far side of such an edge names its rim record (index >= 3 nt) instead
// of HOLE:
//   word 0 .. 2: x, y, z of the edge's ORIGIN a (the interior half-edge runs a -> b, the hole on its right)
//   word 3: the rim record of the next edge around the hole (it starts at b)
//   word 4: the interior half-edge itself -- the walk re-enters the mesh THROUGH it
//   word 5, 6: the first rim record of this hole and the number of its edges (the records of a hole lie together, in rim order:
//              the walk reads them by index -- loads that do not wait for one another)
//   word 7: for a rim of more than RIM_CHUNK_MIN edges the first of its CHUNK records (behind all rim records: {centre x, y, z
//           and radius of a sphere around RIM_CHUNK consecutive edges}, and behind a rim's chunk records one such record per
//           RIM_CHUNK of them, the sphere around theirs -- the walk tests the fan plane against the spheres, two levels, and
//           goes through the edges of the chunks it cuts); top bit: the rim is the mesh's OUTLINE
// The ragged OUTLINE of a survey (every outline edge that is not on the bounding box) is linked the same way when no other piece
// of mesh lies outside it: a slice that leaves through it either finds a cut further out -- a bay of the outline: it walks on
// from there -- or none: then nothing lies beyond, and the beams left miss.  Its edges on the bounding box keep their border
// codes and are marked in word 3 (top bit): no way back in through them.
// A slice that reaches the hole goes around its rim once, finds the nearest edge further out that the fan plane cuts,
// lets the beams that look into the gap miss, and walks on from there (mcl_sweep.h: sweep_side_tin, SURF 6).
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

// Rim records for the holes a walk can cross (see the top of the file).  Boundary loops are traced through the table itself
// (the far side of edge j of triangle T is next_a of record 3 T + (j + 1) % 3 and next_b of record 3 T + (j + 2) % 3; its
// origin is the vertex of record 3 T + (j + 1) % 3): from a boundary half-edge a -> b the next one around the same empty
// face starts at b and is found by turning about b through the triangles of its fan -- which pairs the edges properly
// where two fans touch in a vertex each keeps to itself (pinched holes come out as ONE loop, a triangle that hangs on a rim
// by a vertex as a piece of its own).  With the mesh on the left a loop runs CLOCKWISE around a hole and counter-
// clockwise around a piece of mesh; crossing a hole by its own rim alone is exact only if nothing else lies in it, so
// a hole that contains a piece of mesh (an island) is not linked.  Returns what was appended; he keeps its first 3 nt
// records in place.
constexpr int RIM_CHUNK = 16;       // a long rim is searched by chunks of so many consecutive edges, each inside a bounding sphere
constexpr int RIM_CHUNK_MIN = 48;   // ... rims of up to so many edges are searched edge by edge
constexpr uint32_t RIM_ON_BOX = 0x80000000u;   // word 3 of a rim record: this edge of the OUTLINE lies on the bounding box (no way back in through it)
constexpr uint32_t RIM_EXTERIOR = 0x80000000u; // word 7 of a rim record: the rim is the mesh's outline, the empty space beyond it unbounded
struct Links {
  size_t nrim = 0, nchunk = 0;   // rim records behind the 3 nt half-edge records, chunk records behind those
  size_t holes = 0;              // linked holes
  bool outline = false;          // the outline is linked as well
  size_t outline_base = 0;       // ... its first rim record
};
inline Links link_holes(std::vector<Rec>& he, int64_t nt, bool box_outline = false) {
  Links out;
  const size_t nhe = 3 * (size_t)nt;
  if (he.size() != nhe) return out;
  auto far_of = [&](size_t T, int j) -> uint32_t& { return he[3 * T + (size_t)((j + 1) % 3)].next_a; };
  auto far_of_b = [&](size_t T, int j) -> uint32_t& { return he[3 * T + (size_t)((j + 2) % 3)].next_b; };
  auto origin = [&](size_t T, int j) -> const Rec& { return he[3 * T + (size_t)((j + 1) % 3)]; };
  auto fl = [](uint32_t w) {
    float f;
    std::memcpy(&f, &w, 4);
    return (double)f;
  };
  std::vector<unsigned char> seen(nhe, 0);
  struct Loop {
    size_t first, len;
    double area2;
    float bx0, bx1, by0, by1;
    bool on_box, ragged;   // has an edge on the bounding box / an edge that is not
  };
  std::vector<uint32_t> edges;   // the boundary half-edges (3 T + j), loop after loop
  std::vector<Loop> holes, pieces;   // clockwise loops (holes); counter-clockwise loops (outlines of pieces of mesh)
  for (size_t h0 = 0; h0 < nhe; ++h0) {
    if (seen[h0] || far_of(h0 / 3, (int)(h0 % 3)) < 0xfffffff0u) continue;
    Loop L{edges.size(), 0, 0.0, 3e38f, -3e38f, 3e38f, -3e38f, false, false};
    size_t h = h0;
    for (size_t guard = 0;; ++guard) {
      if (guard > nhe || seen[h]) return out;   // (cannot happen on a table that passed adjacency())
      seen[h] = 1;
      edges.push_back((uint32_t)h);
      const size_t T = h / 3;
      const int j = (int)(h % 3);
      (far_of(T, j) != HOLE ? L.on_box : L.ragged) = true;
      const Rec& a = origin(T, j);
      const Rec& b = origin(T, (j + 1) % 3);
      L.area2 += fl(a.x) * fl(b.y) - fl(b.x) * fl(a.y);
      L.bx0 = std::min(L.bx0, (float)fl(a.x)), L.bx1 = std::max(L.bx1, (float)fl(a.x));
      L.by0 = std::min(L.by0, (float)fl(a.y)), L.by1 = std::max(L.by1, (float)fl(a.y));
      // the next boundary edge around this face: turn about b
      size_t g = 3 * T + (size_t)((j + 1) % 3);
      for (size_t turn = 0; far_of(g / 3, (int)(g % 3)) < 0xfffffff0u; ++turn) {
        if (turn > nhe) return out;
        const uint32_t t = far_of(g / 3, (int)(g % 3));   // runs (end of g) -> b in the neighbour
        g = 3 * (size_t)(t / 3u) + (size_t)((t % 3u + 1u) % 3u);
      }
      h = g;
      if (h == h0) break;
    }
    L.len = edges.size() - L.first;
    if (L.area2 > 0.0)
      pieces.push_back(L);
    else if (!L.on_box)
      holes.push_back(L);
  }
  if (pieces.empty()) return out;
  auto inside = [&](const Loop& H, double px, double py) {   // crossing number against the loop's polygon
    if (px < H.bx0 || px > H.bx1 || py < H.by0 || py > H.by1) return false;
    bool in = false;
    for (size_t e = 0; e < H.len; ++e) {
      const uint32_t hh = edges[H.first + e];
      const Rec& a = origin(hh / 3, (int)(hh % 3));
      const Rec& b = origin(hh / 3, (int)((hh % 3 + 1) % 3));
      const double ax = fl(a.x), ay = fl(a.y), bx = fl(b.x), by = fl(b.y);
      if ((ay > py) != (by > py) && px < ax + (py - ay) * (bx - ax) / (by - ay)) in = !in;
    }
    return in;
  };
  // a piece of mesh other than the one with the largest outline may lie INSIDE a hole (an island; a triangle that hangs on
  // the rim by one vertex): that hole is not linked -- or OUTSIDE the largest outline: then the outline is not.  One point
  // of the piece, the centroid of the triangle of its first boundary edge, against the polygons.
  size_t big = 0;
  for (size_t q = 1; q < pieces.size(); ++q)
    if (pieces[q].area2 > pieces[big].area2) big = q;
  // (the point-in-polygon tests below are pieces x (holes + outline edges): a mesh shattered into tens of thousands of scraps
  //  would spend minutes here -- it gets no rims, its walks hand over as before)
  if ((double)(pieces.size() - 1) * (double)(holes.size() + pieces[big].len) > 4e8) return out;
  bool outline_ok = pieces[big].ragged || box_outline;   // (an outline that lies on the bounding box all around needs no records for a walk that LEAVES: the border codes say it all -- box_outline: records all the same, for sensors beyond the box that look back in)
  {
    std::vector<unsigned char> bad(holes.size(), 0);
    for (size_t q = 0; q < pieces.size(); ++q) {
      if (q == big) continue;
      const size_t T = edges[pieces[q].first] / 3;
      double px = 0.0, py = 0.0;
      for (int j = 0; j < 3; ++j) px += fl(origin(T, j).x) / 3.0, py += fl(origin(T, j).y) / 3.0;
      if (!inside(pieces[big], px, py)) outline_ok = false;
      for (size_t k = 0; k < holes.size(); ++k)
        if (!bad[k] && inside(holes[k], px, py)) bad[k] = 1;
    }
    size_t w = 0;
    for (size_t k = 0; k < holes.size(); ++k)
      if (!bad[k]) holes[w++] = holes[k];
    holes.resize(w);
  }
  std::vector<Loop> rims(holes);
  if (outline_ok) rims.push_back(pieces[big]);
  if (rims.empty()) return out;
  size_t nrim = 0, nchunk = 0;
  for (const Loop& L : rims) {
    nrim += L.len;
    if (L.len > (size_t)RIM_CHUNK_MIN) {
      const size_t nch = (L.len + RIM_CHUNK - 1) / RIM_CHUNK;
      nchunk += nch + (nch + RIM_CHUNK - 1) / RIM_CHUNK;
    }
  }
  if ((nhe + nrim + nchunk) * sizeof(Rec) >= (size_t)1 << 31) return out;   // (the device addresses the table by 32-bit byte offsets)
  he.reserve(nhe + nrim + nchunk);
  size_t cbase = nhe + nrim;   // chunk records follow ALL rim records
  std::vector<Rec> chunks;
  for (size_t r = 0; r < rims.size(); ++r) {
    const Loop& L = rims[r];
    const bool exterior = outline_ok && r + 1 == rims.size();
    const size_t base = he.size();
    if (exterior) out.outline_base = base;
    const bool chunked = L.len > (size_t)RIM_CHUNK_MIN;
    const uint32_t w7 = (chunked ? (uint32_t)(cbase + chunks.size()) : 0u) | (exterior ? RIM_EXTERIOR : 0u);
    for (size_t q = 0; q < L.len; ++q) {
      const uint32_t h = edges[L.first + q];
      const size_t T = h / 3;
      const int j = (int)(h % 3);
      const Rec a = origin(T, j);
      const bool on_box = far_of(T, j) != HOLE;
      he.push_back(Rec{a.x, a.y, a.z, (uint32_t)(base + (q + 1) % L.len) | (on_box ? RIM_ON_BOX : 0u), h, (uint32_t)base, (uint32_t)L.len, w7});
      if (!on_box) {   // (an edge on the bounding box keeps its border code: a slice that leaves through it ends under the old rule)
        far_of(T, j) = (uint32_t)(base + q);
        far_of_b(T, j) = (uint32_t)(base + q);
      }
    }
    if (chunked)
      for (size_t c0 = 0; c0 < L.len; c0 += RIM_CHUNK) {
        // the sphere around vertices c0 .. c0 + RIM_CHUNK (the ends of the chunk's edges), a millimetre wider
        const size_t n = std::min<size_t>(RIM_CHUNK, L.len - c0) + 1;
        double cx = 0, cy = 0, cz = 0, rr = 0;
        for (size_t q = 0; q < n; ++q) {
          const Rec& v = he[base + (c0 + q) % L.len];
          cx += fl(v.x) / n, cy += fl(v.y) / n, cz += fl(v.z) / n;
        }
        const float fcx = (float)cx, fcy = (float)cy, fcz = (float)cz;
        for (size_t q = 0; q < n; ++q) {
          const Rec& v = he[base + (c0 + q) % L.len];
          const double dx = fl(v.x) - fcx, dy = fl(v.y) - fcy, dz = fl(v.z) - fcz;
          rr = std::max(rr, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        const float fr = (float)(rr * (1.0 + 1e-5) + 1e-3);
        Rec c{0, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(&c.x, &fcx, 4);
        std::memcpy(&c.y, &fcy, 4);
        std::memcpy(&c.z, &fcz, 4);
        std::memcpy(&c.next_a, &fr, 4);
        chunks.push_back(c);
      }
    if (chunked) {
      // ... and behind a rim's chunk records one record per RIM_CHUNK of THEM: the sphere around their spheres
      const size_t c_first = chunks.size() - (L.len + RIM_CHUNK - 1) / RIM_CHUNK, nch = chunks.size() - c_first;
      for (size_t g0 = 0; g0 < nch; g0 += RIM_CHUNK) {
        const size_t n = std::min<size_t>(RIM_CHUNK, nch - g0);
        double cx = 0, cy = 0, cz = 0, rr = 0;
        for (size_t q = 0; q < n; ++q) cx += fl(chunks[c_first + g0 + q].x) / n, cy += fl(chunks[c_first + g0 + q].y) / n, cz += fl(chunks[c_first + g0 + q].z) / n;
        const float fcx = (float)cx, fcy = (float)cy, fcz = (float)cz;
        for (size_t q = 0; q < n; ++q) {
          const Rec& v = chunks[c_first + g0 + q];
          const double dx = fl(v.x) - fcx, dy = fl(v.y) - fcy, dz = fl(v.z) - fcz;
          rr = std::max(rr, std::sqrt(dx * dx + dy * dy + dz * dz) + fl(v.next_a));
        }
        const float fr = (float)(rr * (1.0 + 1e-5) + 1e-3);
        Rec c{0, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(&c.x, &fcx, 4);
        std::memcpy(&c.y, &fcy, 4);
        std::memcpy(&c.z, &fcz, 4);
        std::memcpy(&c.next_a, &fr, 4);
        chunks.push_back(c);
      }
    }
  }
  he.insert(he.end(), chunks.begin(), chunks.end());
  out.nrim = nrim;
  out.nchunk = chunks.size();
  out.holes = holes.size();
  out.outline = outline_ok;
  return out;
}

}  // namespace halfedge
