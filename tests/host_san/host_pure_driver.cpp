// host_pure_driver.cpp -- TEST INFRASTRUCTURE: the device-free host arithmetic of libmcl_hip.so under
// AddressSanitizer + UBSan (make -C smarc_navigation_amd/csrc host-asan): mcl_host_pure.h (transfer plan of the resample
// exchange, matrix_from_tf, euler_from_quat, Philox on the host), mcl_dr_impl.h (the dead-reckoning integrator, every
// callback of sam_dead_reckoning/scripts/dr_node.py) and pf_core.hpp's parsers, driven with random and hostile inputs.
// Round 6: mcl_halfedge.h -- the half-edge table of the TIN sweep -- built from random jittered meshes handed over in random
// order with mixed windings, its invariants checked record by record, and WALKED on the CPU by the kernel's own rule
// (mcl_sweep.h: sweep_side_tin -- the new vertex replaces the end on its side of the plane; which of next_a / next_b is
// taken follows from that and from one bit of state) from every triangle a random vertical plane cuts to the mesh border.
// Exit code 0 and no sanitizer report = pass; the properties checked here are the ones tests/test_exchange_plan.py and
// tests/test_dr_golden.py check through the real library.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>

#include "../../smarc_navigation_amd/csrc/mcl_host_pure.h"
#include "../../smarc_navigation_amd/csrc/mcl_halfedge.h"
#include "../../smarc_navigation_amd/csrc/mcl_dr_impl.h"
#include "auv_particle_filter_hip/pf_core.hpp"

#define CHECK(c)                                                            \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)


// ---- a jittered TIN over a (nx x ny)-node grid, every cell split along a random diagonal (synth.mesh_tin), then shuffled:
// vertices renumbered, triangles permuted, corners rotated, half of the windings reversed (synth.mesh_shuffle)
static void random_tin(std::mt19937_64& rng, int nx, int ny, std::vector<float>& verts, std::vector<uint32_t>& tris) {
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<float> v0((size_t)nx * ny * 3);
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) {
      const bool inner = i > 0 && j > 0 && i < nx - 1 && j < ny - 1;
      float* p = &v0[3 * ((size_t)i * ny + j)];
      p[0] = (float)(i + (inner ? 0.25 * U(rng) : 0.0)) * 1.5f - 7.f;
      p[1] = (float)(j + (inner ? 0.25 * U(rng) : 0.0)) * 1.5f + 3.f;
      p[2] = (float)(-20.0 + 2.0 * std::sin(0.4 * i) * std::cos(0.3 * j) + 0.3 * U(rng));
    }
  std::vector<uint32_t> t0;
  for (int i = 0; i + 1 < nx; ++i)
    for (int j = 0; j + 1 < ny; ++j) {
      const uint32_t a = (uint32_t)(i * ny + j), b = a + (uint32_t)ny, c = a + 1u, d = b + 1u;   // 00, 10, 01, 11
      if (rng() & 1u) { t0.insert(t0.end(), {a, b, d}); t0.insert(t0.end(), {a, d, c}); }
      else { t0.insert(t0.end(), {a, b, c}); t0.insert(t0.end(), {b, d, c}); }
    }
  const size_t nv = (size_t)nx * ny, nt = t0.size() / 3;
  std::vector<uint32_t> perm(nv), where(nv), torder(nt);
  for (size_t k = 0; k < nv; ++k) perm[k] = (uint32_t)k;
  for (size_t k = 0; k < nt; ++k) torder[k] = (uint32_t)k;
  std::shuffle(perm.begin(), perm.end(), rng);
  std::shuffle(torder.begin(), torder.end(), rng);
  for (size_t k = 0; k < nv; ++k) where[perm[k]] = (uint32_t)k;
  verts.resize(3 * nv);
  for (size_t k = 0; k < nv; ++k)
    for (int c = 0; c < 3; ++c) verts[3 * k + c] = v0[3 * (size_t)perm[k] + c];
  tris.resize(3 * nt);
  for (size_t k = 0; k < nt; ++k) {
    uint32_t v[3] = {where[t0[3 * (size_t)torder[k]]], where[t0[3 * (size_t)torder[k] + 1]], where[t0[3 * (size_t)torder[k] + 2]]};
    const int rot = (int)(rng() % 3);
    uint32_t w[3] = {v[rot], v[(rot + 1) % 3], v[(rot + 2) % 3]};
    if (rng() & 1u) std::swap(w[1], w[2]);   // winding reversed
    for (int c = 0; c < 3; ++c) tris[3 * k + c] = w[c];
  }
}

static int check_halfedge_tables(std::mt19937_64& rng) {
  using halfedge::Rec;
  for (int trial = 0; trial < 40; ++trial) {
    const int nx = 3 + (int)(rng() % 14), ny = 3 + (int)(rng() % 11);
    std::vector<float> verts;
    std::vector<uint32_t> tris;
    random_tin(rng, nx, ny, verts, tris);
    const int64_t nt = (int64_t)tris.size() / 3;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (size_t i = 0; i < verts.size() / 3; ++i) {
      xmin = std::min(xmin, (double)verts[3 * i]); xmax = std::max(xmax, (double)verts[3 * i]);
      ymin = std::min(ymin, (double)verts[3 * i + 1]); ymax = std::max(ymax, (double)verts[3 * i + 1]);
    }
    std::vector<uint32_t> new_of_old, twin;
    std::vector<unsigned char> ccw;
    std::vector<Rec> he;
    double g2 = 0.0;
    halfedge::morton_order(verts.data(), tris.data(), nt, xmin, xmax, ymin, ymax, new_of_old);
    {   // a permutation
      std::vector<unsigned char> seen((size_t)nt, 0);
      for (int64_t k = 0; k < nt; ++k) {
        CHECK(new_of_old[(size_t)k] < (uint32_t)nt && !seen[new_of_old[(size_t)k]]);
        seen[new_of_old[(size_t)k]] = 1;
      }
    }
    CHECK(halfedge::adjacency(verts.data(), tris.data(), nt, twin, ccw, g2));
    halfedge::build_table(verts.data(), tris.data(), nt, twin, ccw, new_of_old, xmin, xmax, ymin, ymax, he);
    CHECK((int64_t)he.size() == 3 * nt);
    auto vert = [&](uint32_t T, int j, float out[3]) {   // table vertex j of table triangle T: the record of edge j + 1 holds it
      const Rec& r = he[3 * (size_t)T + (size_t)((j + 1) % 3)];
      std::memcpy(out, &r.x, 12);
    };
    size_t borders = 0;
    for (uint32_t T = 0; T < (uint32_t)nt; ++T) {
      float v[3][3];
      for (int j = 0; j < 3; ++j) vert(T, j, v[j]);
      // counter-clockwise in xy
      const double area = ((double)v[1][0] - v[0][0]) * ((double)v[2][1] - v[0][1]) - ((double)v[1][1] - v[0][1]) * ((double)v[2][0] - v[0][0]);
      CHECK(area > 0.0);
      for (int e = 0; e < 3; ++e) {
        const Rec& r = he[3 * (size_t)T + e];
        // next_a: across edge e + 2 = (v_e+2, v_e): the far half-edge runs v_e -> v_e+2; next_b: across (v_e+1, v_e+2): v_e+2 -> v_e+1
        const uint32_t nxt[2] = {r.next_a, r.next_b};
        const int from[2] = {e % 3, (e + 2) % 3}, to[2] = {(e + 2) % 3, (e + 1) % 3};
        for (int s = 0; s < 2; ++s) {
          if (nxt[s] >= 0xfffffff0u) {
            CHECK(nxt[s] == halfedge::BORDER_X || nxt[s] == halfedge::BORDER_Y);   // (this mesh has no holes)
            ++borders;
            continue;
          }
          CHECK(nxt[s] < 3u * (uint32_t)nt);
          const uint32_t T2 = nxt[s] / 3u;
          const int e2 = (int)(nxt[s] % 3u);
          CHECK(T2 != T);
          float a[3], b[3];
          vert(T2, e2, a);
          vert(T2, (e2 + 1) % 3, b);
          CHECK(std::memcmp(a, v[from[s]], 12) == 0 && std::memcmp(b, v[to[s]], 12) == 0);
        }
      }
    }
    CHECK(borders == (size_t)(2 * (nx - 1) + 2 * (ny - 1)) * 2);   // (every border edge is named by two records of its triangle)
    // ---- walk: a vertical plane through a random interior point, outward on both sides from the triangle under it, by the
    // kernel's rule.  Every step must cross an edge whose ends lie on opposite sides of the plane, s must not decrease
    // (the slice of a height field by a vertical plane is a graph over s), and the walk must end at the OUTER border.
    std::uniform_real_distribution<double> U01(0.0, 1.0);
    long total_steps = 0;
    for (int w = 0; w < 20; ++w) {
      const double px = xmin + (0.2 + 0.6 * U01(rng)) * (xmax - xmin), py = ymin + (0.2 + 0.6 * U01(rng)) * (ymax - ymin);
      const double ang = 6.283185307179586 * U01(rng);
      const double c1x = std::cos(ang), c1y = std::sin(ang);      // across-track axis (s); plane normal = (-c1y, c1x, 0)
      auto d_of = [&](const float* p) { return -c1y * ((double)p[0] - px) + c1x * ((double)p[1] - py); };
      auto s_of = [&](const float* p) { return c1x * ((double)p[0] - px) + c1y * ((double)p[1] - py); };
      // the triangle that contains (px, py)
      int64_t T0 = -1;
      for (uint32_t T = 0; T < (uint32_t)nt && T0 < 0; ++T) {
        float v[3][3];
        for (int j = 0; j < 3; ++j) vert(T, j, v[j]);
        bool in = true;
        for (int j = 0; j < 3 && in; ++j) {
          const float* a = v[j];
          const float* b = v[(j + 1) % 3];
          in = ((double)b[0] - a[0]) * (py - a[1]) - ((double)b[1] - a[1]) * (px - a[0]) >= 0.0;
        }
        if (in) T0 = T;
      }
      CHECK(T0 >= 0);
      for (int side = 0; side < 2; ++side) {
        const double sg = side ? -1.0 : 1.0;
        float v[3][3];
        double d[3];
        for (int j = 0; j < 3; ++j) {
          vert((uint32_t)T0, j, v[j]);
          d[j] = d_of(v[j]);
        }
        const bool p0 = d[0] >= 0, p1 = d[1] >= 0, p2 = d[2] >= 0;
        if (p0 == p1 && p1 == p2) continue;   // (the plane grazes a vertex: the kernel declines too)
        const int L = (p0 != p1 && p0 != p2) ? 0 : ((p1 != p0 && p1 != p2) ? 1 : 2);
        const int M = (L + 1) % 3, N = (L + 2) % 3;
        // crossings on edge L = (vL, vL+1) and edge L+2 = (vL+2, vL); this side leaves through the one further out
        const double lm = d[L] / (d[L] - d[M]), ln = d[L] / (d[L] - d[N]);
        const double sm = sg * (s_of(v[L]) + lm * (s_of(v[M]) - s_of(v[L]))), sn = sg * (s_of(v[L]) + ln * (s_of(v[N]) - s_of(v[L])));
        if (sm == sn) continue;
        const bool far_m = sm > sn;
        const bool pl = d[L] >= 0;
        // far side of edge j of T0: next_a of record (j + 1) % 3
        const uint32_t fM = he[3 * (size_t)T0 + (size_t)((L + 1) % 3)].next_a, fN = he[3 * (size_t)T0 + (size_t)((L + 2 + 1) % 3)].next_a;
        uint32_t nb = far_m ? fM : fN;
        double Ad, Bd, As, Bs;   // A: found last (NF when pl), B: the other end of the exit edge
        {
          const int F = far_m ? M : N;
          Ad = pl ? d[F] : d[L]; Bd = pl ? d[L] : d[F];
          As = pl ? sg * s_of(v[F]) : sg * s_of(v[L]); Bs = pl ? sg * s_of(v[L]) : sg * s_of(v[F]);
        }
        bool ao = far_m == pl;
        double s_prev = far_m ? sm : sn;
        int steps = 0;
        while (nb < 0xfffffff0u) {
          CHECK(++steps <= 3 * nt);
          CHECK((Ad >= 0) != (Bd >= 0));                 // the exit edge is cut by the plane
          const Rec& r = he[nb];
          float pN[3];
          std::memcpy(pN, &r.x, 12);
          const double dN = d_of(pN), sN = sg * s_of(pN);
          const bool keep_a = (dN >= 0) != (Ad >= 0);    // the new vertex replaces the end on ITS side
          const bool stays_a = keep_a == ao;
          nb = stays_a ? r.next_a : r.next_b;
          ao = !stays_a;
          if (keep_a) { Bd = Ad; Bs = As; }
          Ad = dN; As = sN;
          const double lam = Ad / (Ad - Bd);
          const double s_new = As + lam * (Bs - As);
          CHECK(s_new >= s_prev - 1e-9);                  // outward, never back
          s_prev = s_new;
        }
        CHECK(nb == halfedge::BORDER_X || nb == halfedge::BORDER_Y);
        total_steps += steps;
      }
    }
    CHECK(total_steps >= 20);   // (the walks did cross triangles)
    // ---- meshes the walk must refuse: a triangle listed twice (three faces on an edge), a folded pair
    {
      std::vector<uint32_t> dup(tris);
      dup.insert(dup.end(), tris.begin(), tris.begin() + 3);
      CHECK(!halfedge::adjacency(verts.data(), dup.data(), nt + 1, twin, ccw, g2));
      std::vector<float> fv = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0.2f, 0.2f, 1};   // two triangles on edge (0,1), third vertices on the SAME side
      std::vector<uint32_t> ft = {0, 1, 2, 1, 0, 3};
      CHECK(!halfedge::adjacency(fv.data(), ft.data(), 2, twin, ccw, g2));
    }
  }
  return 0;
}

// ---- holes and outlines the walk crosses (mcl_halfedge.h: link_holes): random TINs with gaps punched into them -- discs of
// triangles removed around interior points (sometimes merged, sometimes large: a rim long enough for chunk records), a ring
// (an island inside a hole), and RAGGED OUTLINES: border triangles removed at random and bays cut deep into the mesh, the
// largest edge-connected piece kept.  Checked: the rim records (loops in rim order around empty space, naming the interior
// half-edge that names them; box edges of the outline flagged and left with their border code; every vertex of a chunk
// inside its sphere); and WALKS by the kernel's rule (sweep_side_tin<.., HOLES>): at a rim the nearest cut further out --
// found through the chunk spheres exactly as by brute force over every edge --, NO triangle between the two cuts, s never
// decreasing; beyond the outline with no cut left, no triangle anywhere further along the line; an island's hole never linked.
#ifndef RIM_TRIALS
#define RIM_TRIALS 96   // (-DRIM_TRIALS=5000: a longer hunt, by hand)
#endif
static int check_hole_rims(std::mt19937_64& rng) {
  using halfedge::Rec;
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  int linked_meshes = 0, refused = 0, outline_meshes = 0, chunked_meshes = 0;
  long crossings = 0, bays = 0, finals = 0, outside_starts = 0, blinds = 0;
  for (int trial = 0; trial < RIM_TRIALS; ++trial) {
    const int kind = trial % 8;   // 0-2 discs, 3 a disc at the outline, 4 ring (island), 5-6 ragged outline + bays (+ discs), 7 a large hole
    const bool large = kind == 7, ragged = kind == 5 || kind == 6;
    const int nx = (large || ragged ? 22 : 9) + (int)(rng() % 10), ny = (large || ragged ? 20 : 9) + (int)(rng() % 10);
    std::vector<float> verts;
    std::vector<uint32_t> all, tris;
    random_tin(rng, nx, ny, verts, all);
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (size_t i = 0; i < verts.size() / 3; ++i) {
      xmin = std::min(xmin, (double)verts[3 * i]); xmax = std::max(xmax, (double)verts[3 * i]);
      ymin = std::min(ymin, (double)verts[3 * i + 1]); ymax = std::max(ymax, (double)verts[3 * i + 1]);
    }
    const int ndisc = large ? 1 : 1 + (int)(rng() % 3);
    double cx[3], cy[3], rad[3];
    for (int q = 0; q < ndisc; ++q) {
      cx[q] = xmin + (0.3 + 0.4 * U01(rng)) * (xmax - xmin);
      cy[q] = ymin + (0.3 + 0.4 * U01(rng)) * (ymax - ymin);
      rad[q] = large ? 6.5 + 1.5 * U01(rng) : 0.8 + 1.2 * U01(rng);
    }
    if (kind == 3) cx[0] = xmin + 0.3;
    // bays: rectangles cut in from a side (ragged kinds)
    double bay[3][4];
    const int nbay = ragged ? 1 + (int)(rng() % 3) : 0;
    for (int q = 0; q < nbay; ++q) {
      const double w = 1.6 + 2.0 * U01(rng), dep = 4.0 + 6.0 * U01(rng);
      if (rng() & 1u) {   // from the left or right side
        const double y = ymin + (0.2 + 0.6 * U01(rng)) * (ymax - ymin);
        const bool left = rng() & 1u;
        bay[q][0] = left ? xmin - 1 : xmax - dep; bay[q][1] = left ? xmin + dep : xmax + 1; bay[q][2] = y - w / 2; bay[q][3] = y + w / 2;
      } else {
        const double x = xmin + (0.2 + 0.6 * U01(rng)) * (xmax - xmin);
        const bool low = rng() & 1u;
        bay[q][0] = x - w / 2; bay[q][1] = x + w / 2; bay[q][2] = low ? ymin - 1 : ymax - dep; bay[q][3] = low ? ymin + dep : ymax + 1;
      }
    }
    for (size_t k = 0; k < all.size() / 3; ++k) {
      double mx = 0, my = 0;
      for (int c = 0; c < 3; ++c) { mx += verts[3 * (size_t)all[3 * k + c]] / 3.0; my += verts[3 * (size_t)all[3 * k + c] + 1] / 3.0; }
      bool gone = false;
      for (int q = 0; q < ndisc; ++q) {
        const double r = std::hypot(mx - cx[q], my - cy[q]);
        gone |= (kind == 4 && q == 0) ? (r > 1.2 && r < 3.0) : (ragged && q > 0 ? false : r < rad[q]);
      }
      for (int q = 0; q < nbay; ++q) gone |= mx > bay[q][0] && mx < bay[q][1] && my > bay[q][2] && my < bay[q][3];
      if (ragged) {
        const double e = std::min(std::min(mx - xmin, xmax - mx), std::min(my - ymin, ymax - my));
        gone |= e < 1.7 && U01(rng) < 0.45;
      }
      if (!gone) tris.insert(tris.end(), all.begin() + 3 * (long)k, all.begin() + 3 * (long)k + 3);
    }
    if (ragged) {
      // keep the largest edge-connected piece (the scraps a ragged border leaves would lie OUTSIDE the outline)
      const size_t n = tris.size() / 3;
      std::unordered_map<uint64_t, std::vector<uint32_t>> by_edge;
      for (size_t k = 0; k < n; ++k)
        for (int e = 0; e < 3; ++e) {
          const uint32_t a = tris[3 * k + e], b = tris[3 * k + (e + 1) % 3];
          by_edge[a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a].push_back((uint32_t)k);
        }
      std::vector<int> comp(n, -1);
      int ncomp = 0, bestc = 0;
      size_t bestn = 0;
      for (size_t k0 = 0; k0 < n; ++k0) {
        if (comp[k0] >= 0) continue;
        std::vector<uint32_t> stack{(uint32_t)k0};
        comp[k0] = ncomp;
        size_t cnt = 0;
        while (!stack.empty()) {
          const uint32_t k = stack.back();
          stack.pop_back();
          ++cnt;
          for (int e = 0; e < 3; ++e) {
            const uint32_t a = tris[3 * (size_t)k + e], b = tris[3 * (size_t)k + (e + 1) % 3];
            for (uint32_t k2 : by_edge[a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a])
              if (comp[k2] < 0) { comp[k2] = ncomp; stack.push_back(k2); }
          }
        }
        if (cnt > bestn) { bestn = cnt; bestc = ncomp; }
        ++ncomp;
      }
      std::vector<uint32_t> kept;
      for (size_t k = 0; k < n; ++k)
        if (comp[k] == bestc) kept.insert(kept.end(), tris.begin() + 3 * (long)k, tris.begin() + 3 * (long)k + 3);
      tris.swap(kept);
      // (the bounding box is the kept piece's own)
      xmin = ymin = 1e300; xmax = ymax = -1e300;
      for (uint32_t v : tris) {
        xmin = std::min(xmin, (double)verts[3 * (size_t)v]); xmax = std::max(xmax, (double)verts[3 * (size_t)v]);
        ymin = std::min(ymin, (double)verts[3 * (size_t)v + 1]); ymax = std::max(ymax, (double)verts[3 * (size_t)v + 1]);
      }
    }
    const int64_t nt = (int64_t)tris.size() / 3;
    std::vector<uint32_t> new_of_old, twin;
    std::vector<unsigned char> ccw;
    std::vector<Rec> he;
    double g2 = 0.0;
    halfedge::morton_order(verts.data(), tris.data(), nt, xmin, xmax, ymin, ymax, new_of_old);
    CHECK(halfedge::adjacency(verts.data(), tris.data(), nt, twin, ccw, g2));
    halfedge::build_table(verts.data(), tris.data(), nt, twin, ccw, new_of_old, xmin, xmax, ymin, ymax, he);
    const size_t nhe = 3 * (size_t)nt;
    const halfedge::Links links = halfedge::link_holes(he, nt);
    const size_t nrim = links.nrim;
    CHECK(he.size() == nhe + nrim + links.nchunk);
    auto xyz = [&](const Rec& r, float out[3]) { std::memcpy(out, &r.x, 12); };
    auto vert = [&](uint32_t T, int j, float out[3]) { xyz(he[3 * (size_t)T + (size_t)((j + 1) % 3)], out); };
    auto inside_some_triangle = [&](double px, double py) {
      for (uint32_t T = 0; T < (uint32_t)nt; ++T) {
        float v[3][3];
        for (int j = 0; j < 3; ++j) vert(T, j, v[j]);
        bool in = true;
        for (int j = 0; j < 3 && in; ++j)
          in = ((double)v[(j + 1) % 3][0] - v[j][0]) * (py - v[j][1]) - ((double)v[(j + 1) % 3][1] - v[j][1]) * (px - v[j][0]) > 1e-9;
        if (in) return true;
      }
      return false;
    };
    if (kind == 4 && inside_some_triangle(cx[0], cy[0])) {
      // an island inside a hole: THAT hole may not be linked (others of the same mesh may): the island's point lies inside
      // no linked hole's rim polygon
      for (size_t k = nhe; k < nhe + nrim; k += he[k].pad1) {
        if (he[k].pad2 & halfedge::RIM_EXTERIOR) continue;
        bool in = false;
        for (size_t q = k; q < k + he[k].pad1; ++q) {
          float a[3], b[3];
          xyz(he[q], a);
          xyz(he[he[q].next_a & 0x7fffffffu], b);
          if ((a[1] > cy[0]) != (b[1] > cy[0]) && cx[0] < a[0] + (cy[0] - a[1]) * ((double)b[0] - a[0]) / ((double)b[1] - a[1])) in = !in;
        }
        CHECK(!in);
      }
      ++refused;
    }
    if (nrim == 0) continue;
    ++linked_meshes;
    outline_meshes += links.outline ? 1 : 0;
    chunked_meshes += links.nchunk ? 1 : 0;
    // ---- the records
    for (size_t k = nhe; k < nhe + nrim; ++k) {
      const Rec& r = he[k];
      const uint32_t nxt = r.next_a & 0x7fffffffu;
      const bool on_box = (r.next_a & halfedge::RIM_ON_BOX) != 0, ext = (r.pad2 & halfedge::RIM_EXTERIOR) != 0;
      CHECK(nxt >= nhe && nxt < nhe + nrim && r.next_b < nhe);
      CHECK(r.pad0 >= nhe && r.pad1 >= 3 && k >= r.pad0 && k < (size_t)r.pad0 + r.pad1);             // the rim's records lie together ...
      CHECK(nxt == r.pad0 + (uint32_t)((k - r.pad0 + 1) % r.pad1) && he[nxt].pad0 == r.pad0 && he[nxt].pad1 == r.pad1 && he[nxt].pad2 == r.pad2);   // ... in rim order
      CHECK(!on_box || ext);                                                                          // only the outline has edges on the box
      const uint32_t h = r.next_b, T = h / 3u;
      const int j = (int)(h % 3u);
      const uint32_t far_a = he[3 * (size_t)T + (size_t)((j + 1) % 3)].next_a, far_b = he[3 * (size_t)T + (size_t)((j + 2) % 3)].next_b;
      CHECK(far_a == far_b);
      if (on_box) CHECK(far_a == halfedge::BORDER_X || far_a == halfedge::BORDER_Y); else CHECK(far_a == (uint32_t)k);
      float a[3], b[3], ra[3], rb[3];
      vert(T, j, a);
      vert(T, (j + 1) % 3, b);
      xyz(r, ra);
      xyz(he[nxt], rb);
      CHECK(std::memcmp(a, ra, 12) == 0 && std::memcmp(b, rb, 12) == 0);   // the edge a -> b, the next rim edge starts at b
      // empty space on the RIGHT of a -> b: a point just right of the edge's middle lies in no triangle
      const double ex = (double)b[0] - a[0], ey = (double)b[1] - a[1], el = std::hypot(ex, ey);
      CHECK(!inside_some_triangle(0.5 * (a[0] + b[0]) + 1e-3 * ey / el, 0.5 * (a[1] + b[1]) - 1e-3 * ex / el));
      const uint32_t cb = r.pad2 & 0x7fffffffu;
      CHECK((cb != 0) == (r.pad1 > (uint32_t)halfedge::RIM_CHUNK_MIN));
      if (cb) {   // the vertex lies in the sphere of its chunk (and, as the end of the last edge before it, of the one before)
        CHECK(cb >= nhe + nrim && cb + (r.pad1 + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK <= he.size());
        {   // ... and the chunk's sphere in the sphere of its group of chunks
          const size_t nch2 = (r.pad1 + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK, ch = (k - r.pad0) / halfedge::RIM_CHUNK;
          CHECK(cb + nch2 + ch / halfedge::RIM_CHUNK < he.size());
          float c0v[3], c1v[3], R0, R1;
          xyz(he[cb + ch], c0v);
          xyz(he[cb + nch2 + ch / halfedge::RIM_CHUNK], c1v);
          std::memcpy(&R0, &he[cb + ch].next_a, 4);
          std::memcpy(&R1, &he[cb + nch2 + ch / halfedge::RIM_CHUNK].next_a, 4);
          CHECK(std::sqrt(((double)c0v[0] - c1v[0]) * ((double)c0v[0] - c1v[0]) + ((double)c0v[1] - c1v[1]) * ((double)c0v[1] - c1v[1]) + ((double)c0v[2] - c1v[2]) * ((double)c0v[2] - c1v[2])) + R0 <= R1);
        }
        const size_t pos = k - r.pad0, nch = (r.pad1 + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK;
        const size_t c1 = pos / halfedge::RIM_CHUNK, c0 = pos % halfedge::RIM_CHUNK == 0 ? (c1 + nch - 1) % nch : c1;
        for (size_t c : {c0, c1}) {
          float cc[3], R;
          xyz(he[cb + c], cc);
          std::memcpy(&R, &he[cb + c].next_a, 4);
          CHECK(std::sqrt(((double)ra[0] - cc[0]) * ((double)ra[0] - cc[0]) + ((double)ra[1] - cc[1]) * ((double)ra[1] - cc[1]) + ((double)ra[2] - cc[2]) * ((double)ra[2] - cc[2])) <= R);
        }
      }
    }
    // no rim code on an edge of a loop that was not linked
    // ---- walks
    int final_checks = 0;
    for (int w = 0; w < 24; ++w) {
      double px, py;
      if (ragged && (w & 3) == 3) {   // around the outline, beyond the bounding box as well
        const double e = -4.0 + 7.0 * U01(rng);   // metres inside (+) / outside (-) a random side
        const int sd = (int)(rng() % 4);
        px = sd == 0 ? xmin + e : (sd == 1 ? xmax - e : xmin + U01(rng) * (xmax - xmin));
        py = sd == 2 ? ymin + e : (sd == 3 ? ymax - e : ymin + U01(rng) * (ymax - ymin));
      } else if (ragged && (w & 1)) {   // near the outline
        px = xmin + (0.1 + 0.8 * U01(rng)) * (xmax - xmin);
        py = ymin + (0.1 + 0.8 * U01(rng)) * (ymax - ymin);
      } else {
        const int q0 = (int)(rng() % (unsigned)ndisc);
        const double reach = large ? 18.0 : 3.0;
        px = cx[q0] + reach * (U01(rng) - 0.5);
        py = cy[q0] + reach * (U01(rng) - 0.5);
      }
      const double ang = 6.283185307179586 * U01(rng);
      const double c1x = std::cos(ang), c1y = std::sin(ang);
      auto d_of = [&](const float* p) { return -c1y * ((double)p[0] - px) + c1x * ((double)p[1] - py); };
      auto s_of = [&](const float* p) { return c1x * ((double)p[0] - px) + c1y * ((double)p[1] - py); };
      int64_t T0 = -1;
      for (uint32_t T = 0; T < (uint32_t)nt && T0 < 0; ++T) {
        float v[3][3];
        for (int j = 0; j < 3; ++j) vert(T, j, v[j]);
        bool in = true;
        for (int j = 0; j < 3 && in; ++j)
          in = ((double)v[(j + 1) % 3][0] - v[j][0]) * (py - v[j][1]) - ((double)v[(j + 1) % 3][1] - v[j][1]) * (px - v[j][0]) >= 0.0;
        if (in) T0 = T;
      }
      if (T0 < 0 && !links.outline) continue;   // (the point fell into a gap or off the mesh)
      // the nearest cut of a rim beyond s_min (strictly, or not), not through the edge `k0`, through an edge on the box only if
      // box_too -- by brute force over every edge, and through the two levels of chunk spheres as the kernel goes: the same edge
      struct RimHit { uint32_t best; int cuts; double bs, Ad, Bd, As, Bs; };
      auto rim_search = [&](double sg, uint32_t k0, uint32_t rbase, uint32_t rlen, uint32_t cb, double s_min, bool strict, bool box_too, RimHit& hit) -> int {
        uint32_t best[2] = {0xffffffffu, 0xffffffffu};
        double bs[2] = {1e300, 1e300};
        hit.cuts = 0;
        hit.Ad = hit.Bd = hit.As = hit.Bs = 0;
        for (int pass = 0; pass < (cb ? 2 : 1); ++pass) {
          for (uint32_t e = 0; e < rlen; ++e) {
            const uint32_t cur = rbase + e, nxt = rbase + (e + 1) % rlen;
            if (cur == k0 || (!box_too && (he[cur].next_a & halfedge::RIM_ON_BOX))) continue;
            if (pass == 1) {   // is this edge's chunk visited?  (two levels: its chunk's sphere and the sphere around 16 chunks)
              const uint32_t nch = (rlen + halfedge::RIM_CHUNK - 1) / halfedge::RIM_CHUNK, ch = e / halfedge::RIM_CHUNK;
              bool visit = true;
              for (const uint32_t rec : {cb + nch + ch / halfedge::RIM_CHUNK, cb + ch}) {
                float cc[3], R;
                xyz(he[rec], cc);
                std::memcpy(&R, &he[rec].next_a, 4);
                visit = visit && std::fabs(d_of(cc)) <= R && sg * s_of(cc) + R >= s_min;
              }
              if (!visit) continue;
            }
            float pc[3], pn[3];
            xyz(he[cur], pc);
            xyz(he[nxt], pn);
            const double dc = d_of(pc), dn = d_of(pn), sc2 = sg * s_of(pc), sn2 = sg * s_of(pn);
            if ((dc >= 0) == (dn >= 0)) continue;
            const double lam = dc / (dc - dn), sx = sc2 + lam * (sn2 - sc2);
            if (!(strict ? sx > s_min : sx >= s_min)) continue;
            if (pass == 0) ++hit.cuts;
            if (sx < bs[pass]) {
              bs[pass] = sx; best[pass] = cur;
              if (pass == 0) { hit.Ad = dc; hit.Bd = dn; hit.As = sc2; hit.Bs = sn2; }
            }
          }
        }
        if (cb) CHECK(best[1] == best[0]);
        hit.best = best[0];
        hit.bs = bs[0];
        return 0;
      };
      for (int side = 0; side < 2; ++side) {
        const double sg = side ? -1.0 : 1.0;
        uint32_t nb;
        double Ad, Bd, As, Bs, s_prev;
        bool ao;
        if (T0 < 0) {
          // no triangle under the point and the outline linked: beyond it?  Then an even number of its edges -- box edges too --
          // is cut on this side: none: nothing to see; else the walk starts at the nearest
          const uint32_t ob = (uint32_t)links.outline_base;
          RimHit hit;
          if (rim_search(sg, 0xffffffffu, ob, he[ob].pad1, he[ob].pad2 & 0x7fffffffu, 0.0, true, true, hit)) return 1;
          if (hit.cuts & 1) continue;   // (inside the outline: a hole -- the kernel asks the cell grid, or hands over)
          if (hit.cuts == 0) {
            if (final_checks < 12) {
              ++final_checks;
              for (double sx = 0.05; sx < 80.0; sx += 0.2) CHECK(!inside_some_triangle(px + sg * sx * c1x, py + sg * sx * c1y));
              ++blinds;
            }
            continue;
          }
          for (int m = 1; m < 12; ++m) CHECK(!inside_some_triangle(px + sg * hit.bs * m / 12.0 * c1x, py + sg * hit.bs * m / 12.0 * c1y));
          ++outside_starts;
          s_prev = hit.bs;
          Ad = hit.Ad; Bd = hit.Bd; As = hit.As; Bs = hit.Bs;
          ao = true;
          nb = he[hit.best].next_b;
          CHECK(nb < nhe);
        } else {
        float v[3][3];
        double d[3];
        for (int j = 0; j < 3; ++j) {
          vert((uint32_t)T0, j, v[j]);
          d[j] = d_of(v[j]);
        }
        const bool p0 = d[0] >= 0, p1 = d[1] >= 0, p2 = d[2] >= 0;
        if (p0 == p1 && p1 == p2) continue;
        const int L = (p0 != p1 && p0 != p2) ? 0 : ((p1 != p0 && p1 != p2) ? 1 : 2);
        const int M = (L + 1) % 3, N = (L + 2) % 3;
        const double lm = d[L] / (d[L] - d[M]), ln = d[L] / (d[L] - d[N]);
        const double sm = sg * (s_of(v[L]) + lm * (s_of(v[M]) - s_of(v[L]))), sn = sg * (s_of(v[L]) + ln * (s_of(v[N]) - s_of(v[L])));
        if (sm == sn) continue;
        const bool far_m = sm > sn;
        const bool pl = d[L] >= 0;
        const uint32_t fM = he[3 * (size_t)T0 + (size_t)((L + 1) % 3)].next_a, fN = he[3 * (size_t)T0 + (size_t)((L + 2 + 1) % 3)].next_a;
        nb = far_m ? fM : fN;
        {
          const int F = far_m ? M : N;
          Ad = pl ? d[F] : d[L]; Bd = pl ? d[L] : d[F];
          As = pl ? sg * s_of(v[F]) : sg * s_of(v[L]); Bs = pl ? sg * s_of(v[L]) : sg * s_of(v[F]);
        }
        ao = far_m == pl;
        s_prev = far_m ? sm : sn;
        }
        int steps = 0;
        bool ended = false;
        while (nb < 0xfffffff0u && !ended) {
          CHECK(++steps <= 3 * nt + 64);
          if (nb >= nhe) {
            CHECK(nb < nhe + nrim);
            // a rim: the nearest cut further out, not through the edge reached, not through an edge on the box
            const uint32_t k0 = nb, rbase = he[k0].pad0, rlen = he[k0].pad1, cb = he[k0].pad2 & 0x7fffffffu;
            const bool ext = (he[k0].pad2 & halfedge::RIM_EXTERIOR) != 0;
            RimHit hit;
            if (rim_search(sg, k0, rbase, rlen, cb, s_prev, false, false, hit)) return 1;
            const uint32_t best[1] = {hit.best};
            const double bs[1] = {hit.bs}, bAd = hit.Ad, bBd = hit.Bd, bAs = hit.As, bBs = hit.Bs;
            if (best[0] == 0xffffffffu) {
              if (!ext) break;   // (a cut through a rim vertex, to rounding: the kernel hands over)
              // beyond the outline for good: no triangle further along the line
              if (final_checks < 8) {
                ++final_checks;
                for (double sx = s_prev + 0.05; sx < 80.0; sx += 0.2) CHECK(!inside_some_triangle(px + sg * sx * c1x, py + sg * sx * c1y));
                ++finals;
              }
              ended = true;
              break;
            }
            for (int m = 1; m < 8; ++m) {
              const double sm2 = s_prev + (bs[0] - s_prev) * m / 8.0;
              CHECK(!inside_some_triangle(px + sg * sm2 * c1x, py + sg * sm2 * c1y));
            }
            (ext ? bays : crossings) += 1;
            s_prev = bs[0];
            Ad = bAd; Bd = bBd; As = bAs; Bs = bBs;
            ao = true;
            nb = he[best[0]].next_b;
            CHECK(nb < nhe);
          }
          CHECK((Ad >= 0) != (Bd >= 0));
          const Rec& r = he[nb];
          float pN[3];
          xyz(r, pN);
          const double dN = d_of(pN), sN = sg * s_of(pN);
          const bool keep_a = (dN >= 0) != (Ad >= 0);
          const bool stays_a = keep_a == ao;
          nb = stays_a ? r.next_a : r.next_b;
          ao = !stays_a;
          if (keep_a) { Bd = Ad; Bs = As; }
          Ad = dN; As = sN;
          const double lam = Ad / (Ad - Bd);
          const double s_new = As + lam * (Bs - As);
          CHECK(s_new >= s_prev - 1e-9);
          s_prev = s_new;
        }
      }
    }
  }
  std::fprintf(stderr, "rims: %d meshes linked (%d with their outline, %d with chunk records), %ld holes crossed, %ld bays of an outline crossed, %ld exits beyond an outline verified empty, %ld walks started beyond an outline, %ld sides beyond one that see nothing, %d islands refused\n",
               linked_meshes, outline_meshes, chunked_meshes, crossings, bays, finals, outside_starts, blinds, refused);
  CHECK(linked_meshes >= 40 && crossings >= 200 && refused >= 4 && outline_meshes >= 12 && chunked_meshes >= 12 && bays >= 10 && finals >= 50 && outside_starts >= 30 && blinds >= 10);
  return 0;
}

int main() {
  std::mt19937_64 rng(12345);
  if (check_halfedge_tables(rng) != 0) return 1;
  if (check_hole_rims(rng) != 0) return 1;
  // ---- transfer plan: for random worlds, what q sends r is what r receives from q, and every lost slot is filled once
  for (int trial = 0; trial < 2000; ++trial) {
    const int world = 1 + (int)(rng() % 9);
    std::vector<uint32_t> lost(world), surplus(world);
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) {
      lost[r] = (uint32_t)(rng() % (trial % 7 == 0 ? 3 : 100000));
      total += lost[r];
    }
    uint64_t left = total;
    for (int r = 0; r < world; ++r) {
      surplus[r] = r == world - 1 ? (uint32_t)left : (uint32_t)(left ? rng() % (left + 1) : 0);
      left -= surplus[r];
    }
    std::vector<std::vector<uint32_t>> so(world), sc(world), ro(world), rc(world);
    for (int q = 0; q < world; ++q) {
      so[q].resize(world); sc[q].resize(world); ro[q].resize(world); rc[q].resize(world);
      CHECK(exchange_plan_impl(world, lost.data(), surplus.data(), q, so[q].data(), sc[q].data(), ro[q].data(), rc[q].data()) == MCL_OK);
    }
    for (int q = 0; q < world; ++q) {
      uint64_t sent = 0, got = 0;
      for (int r = 0; r < world; ++r) {
        CHECK(sc[q][r] == rc[r][q]);
        if (sc[q][r]) CHECK((uint64_t)so[q][r] + sc[q][r] <= surplus[q]);   // (an empty range's offset means nothing)
        if (rc[q][r]) CHECK((uint64_t)ro[q][r] + rc[q][r] <= lost[q]);
        sent += sc[q][r];
        got += rc[q][r];
      }
      CHECK(sent == surplus[q] && got == lost[q]);
    }
    uint32_t bad = surplus[0] + 1;   // counts that do not add up are refused, not planned
    std::swap(bad, surplus[0]);
    CHECK(exchange_plan_impl(world, lost.data(), surplus.data(), 0, so[0].data(), sc[0].data(), ro[0].data(), rc[0].data()) == MCL_ERR_INVALID);
  }
  CHECK(exchange_plan_impl(0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr) == MCL_ERR_INVALID);
  // ---- matrix_from_tf / euler_from_quat: rotation part orthonormal for any quaternion, identity for a zero one
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int trial = 0; trial < 5000; ++trial) {
    double q[4] = {U(rng), U(rng), U(rng), U(rng)}, t[3] = {U(rng) * 1e3, U(rng) * 1e3, U(rng) * 10}, m[16], rpy[3];
    if (trial == 0) q[0] = q[1] = q[2] = q[3] = 0.0;
    if (trial == 1) { q[0] = 1e-200; q[1] = q[2] = q[3] = 0.0; }
    CHECK(matrix_from_tf_impl(t, q, m) == MCL_OK);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double d = 0.0;
        for (int k = 0; k < 3; ++k) d += m[a * 4 + k] * m[b * 4 + k];
        CHECK(std::fabs(d - (a == b ? 1.0 : 0.0)) < 1e-12);
      }
    CHECK(m[3] == t[0] && m[7] == t[1] && m[11] == t[2] && m[15] == 1.0);
    euler_from_quat(q, rpy);
    CHECK(rpy[0] == rpy[0] && rpy[1] == rpy[1] && rpy[2] == rpy[2]);
  }
  CHECK(matrix_from_tf_impl(nullptr, nullptr, nullptr) == MCL_ERR_INVALID);
  CHECK(native_u53(7, 3) < (1ull << 53) && ceil_log2(1) == 0 && ceil_log2(1048577) == 21);
  // ---- the dead-reckoning integrator: a few thousand messages in a plausible and then in a hostile order
  {
    mcl_dr_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.dvl_period = 0.1;
    cfg.dr_period = 0.02;
    mcl_dr* d = nullptr;
    CHECK(mcl_dr_create(&cfg, &d) == MCL_OK && d);
    const double q0[4] = {0, 0, 0.1, 0.99};
    int has = 0;
    mcl_dr_odom tick;
    mcl_odom od;
    CHECK(mcl_dr_tick(d, &tick) == MCL_OK && tick.published == 0);   // nothing known yet: nothing published (dr_node.py:167)
    mcl_dr_heading(d, q0);
    const double b2p[3] = {0.1, 0.0, -0.05};
    double m2o_t[3], m2o_q[4];
    mcl_dr_gps(d, 10.0, -4.0, 1, b2p, &has, m2o_t, m2o_q);
    CHECK(has == 1);
    for (int k = 0; k < 4000; ++k) {
      const double stamp = 100.0 + 0.005 * k;
      const double qi[4] = {0.01 * std::sin(0.01 * k), 0.01 * std::cos(0.013 * k), std::sin(0.001 * k), std::cos(0.001 * k)};
      const double w[3] = {0.0, 0.0, 0.05};
      (void)mcl_dr_imu(d, stamp, qi, w);
      if (k % 20 == 0) {
        const double v[3] = {1.0 + U(rng) * (k % 400 == 0 ? 50.0 : 0.05), U(rng) * 0.02, 0.0};   // (every 20th a wild one: the gates)
        (void)mcl_dr_dvl(d, stamp, v);
      }
      if (k % 50 == 0) (void)mcl_dr_depth(d, 2.0 + 0.1 * std::sin(0.01 * k));
      if (k % 97 == 0) {
        (void)mcl_dr_thrust_cmd(d, 0.05 * U(rng));
        (void)mcl_dr_thrust(d, 800.0 + 100.0 * U(rng), 800.0);
      }
      if (k % 4 == 0) {
        CHECK(mcl_dr_tick(d, &tick) == MCL_OK);
        if (tick.published) CHECK(mcl_dr_to_odom(&tick, stamp, &od) == MCL_OK && od.stamp == stamp);
      }
    }
    CHECK(tick.published == 1 && tick.pos[0] == tick.pos[0]);
    const double nanq[4] = {NAN, NAN, NAN, NAN};
    (void)mcl_dr_imu(d, 50.0, nanq, nanq);            // time going backwards, NaNs: must not crash or index out of range
    (void)mcl_dr_tick(d, &tick);
    mcl_dr_destroy(d);
  }
  // ---- the node core's parsers on hostile text
  {
    double c[6];
    CHECK(auv_pf_hip::parse_cov_string("[1., 2., 0.0, 0.0, 0.0, 0.0001]", c) && c[1] == 2.0 && c[5] == 0.0001);
    CHECK(!auv_pf_hip::parse_cov_string("", c) && !auv_pf_hip::parse_cov_string("[1, 2, 3]", c) && !auv_pf_hip::parse_cov_string("[a, b, c, d, e, f]", c));
    CHECK(!auv_pf_hip::parse_cov_string("[, , , , , ]", c));
    auv_pf_hip::MapFile m;
    std::string err;
    CHECK(!auv_pf_hip::load_map_file("/nonexistent/map.ply", m, err) && !err.empty());
    std::vector<double> lm;
    CHECK(!auv_pf_hip::load_landmark_file("/nonexistent/rocks.yaml", 1e300, lm, err));
  }
  std::puts("host_pure_driver: ok");
  return 0;
}
