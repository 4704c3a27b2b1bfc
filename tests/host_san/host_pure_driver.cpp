// host_pure_driver.cpp -- TEST INFRASTRUCTURE: the device-free host arithmetic of libmcl_hip.so under
// AddressSanitizer + UBSan (make -C smarc_navigation_amd/csrc host-asan): mcl_host_pure.h (transfer plan of the resample
// exchange, matrix_from_tf, euler_from_quat, Philox on the host), mcl_dr_impl.h (the dead-reckoning integrator, every
// callback of sam_dead_reckoning/scripts/dr_node.py) and pf_core.hpp's parsers, driven with random and hostile inputs.
// Exit code 0 and no sanitizer report = pass; the properties checked here are the ones tests/test_exchange_plan.py and
// tests/test_dr_golden.py check through the real library.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../smarc_navigation_amd/csrc/mcl_host_pure.h"
#include "../../smarc_navigation_amd/csrc/mcl_dr_impl.h"
#include "auv_particle_filter_hip/pf_core.hpp"

#define CHECK(c)                                                            \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  std::mt19937_64 rng(12345);
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
