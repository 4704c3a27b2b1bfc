// mcl_host_pure.h -- the host arithmetic of libmcl_hip.so that touches no device: tf.transformations' Euler /
// quaternion formulas, Philox on the host, the resample exchange's transfer plan, matrix_from_tf.  No HIP header: the
// translation unit mcl_api.hip includes it through mcl_host.h, and `make host-asan` compiles it -- with mcl_dr_impl.h and
// the node's core -- under AddressSanitizer / UBSan / ThreadSanitizer with plain g++ (the reference is racy
// by construction, auv_pf.py:126,202-211,264-285; GPU sanitizers are not available on this pool).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mcl.h"

namespace {

typedef unsigned int u32_host;

// euler_from_quaternion(q,'sxyz') -- tf.transformations' published algorithm (auv_particle.py:50)
void euler_from_quat(const double qin[4], double rpy[3]) {
  double nq = qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3];
  double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (nq >= 2.220446049250313e-16 * 4.0) {
    double s = std::sqrt(2.0 / nq);
    double q[4] = {qin[0] * s, qin[1] * s, qin[2] * s, qin[3] * s};
    double o[4][4];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) o[a][b] = q[a] * q[b];
    M[0] = 1.0 - o[1][1] - o[2][2];
    M[1] = o[0][1] - o[2][3];
    M[2] = o[0][2] + o[1][3];
    M[3] = o[0][1] + o[2][3];
    M[4] = 1.0 - o[0][0] - o[2][2];
    M[5] = o[1][2] - o[0][3];
    M[6] = o[0][2] - o[1][3];
    M[7] = o[1][2] + o[0][3];
    M[8] = 1.0 - o[0][0] - o[1][1];
  }
  double cy = std::sqrt(M[0] * M[0] + M[3] * M[3]);
  if (cy > 2.220446049250313e-16 * 4.0) {
    rpy[0] = std::atan2(M[7], M[8]);
    rpy[1] = std::atan2(-M[6], cy);
    rpy[2] = std::atan2(M[3], M[0]);
  } else {
    rpy[0] = std::atan2(-M[5], M[4]);
    rpy[1] = std::atan2(-M[6], cy);
    rpy[2] = 0.0;
  }
}

void philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0;
  o[1] = c1;
  o[2] = c2;
  o[3] = c3;
}
uint64_t native_u53(uint64_t seed, uint32_t step) {
  uint32_t o[4];
  philox_host(0xFFFFFFFFu, 0u, step, 3u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return ((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6);
}

int ceil_log2(long long n) {
  int l = 0;
  while ((1ll << l) < n) ++l;
  return l;
}


// the range of global dupes positions that shard `from` holds and shard `to` needs: [lo, hi).  Lpre / Spre: exclusive
// prefix sums of the shards' lost-slot and surplus-copy counts (world + 1 entries).  Pure host arithmetic: also what
// mcl_exchange_plan exposes, so the plan is property-tested without a GPU (tests/test_exchange_plan.py).
void plan_range(const u32_host* Lpre, const u32_host* Spre, int from, int to, u32_host& lo, u32_host& hi) {
  lo = std::max(Spre[from], Lpre[to]);
  hi = std::min(Spre[from + 1], Lpre[to + 1]);
  if (hi < lo) hi = lo;
}

// mcl_exchange_plan (include/mcl.h): what `rank` sends to / receives from each peer
int exchange_plan_impl(int32_t world, const uint32_t* lost, const uint32_t* surplus, int32_t rank, uint32_t* send_off,
                       uint32_t* send_cnt, uint32_t* recv_off, uint32_t* recv_cnt) {
  if (world < 1 || !lost || !surplus || rank < 0 || rank >= world || !send_off || !send_cnt || !recv_off || !recv_cnt)
    return MCL_ERR_INVALID;
  std::vector<u32_host> Lpre((size_t)world + 1, 0u), Spre((size_t)world + 1, 0u);
  unsigned long long tl = 0, ts = 0;
  for (int r = 0; r < world; ++r) {
    tl += lost[r];
    ts += surplus[r];
    if (tl > 0xffffffffull || ts > 0xffffffffull) return MCL_ERR_INVALID;
    Lpre[r + 1] = (u32_host)tl;
    Spre[r + 1] = (u32_host)ts;
  }
  if (tl != ts) return MCL_ERR_INVALID;   // every lost slot takes exactly one surplus copy
  for (int r = 0; r < world; ++r) {
    u32_host lo, hi;
    plan_range(Lpre.data(), Spre.data(), rank, r, lo, hi);   // what `rank` holds and r needs
    send_off[r] = lo - Spre[rank];
    send_cnt[r] = hi - lo;
    plan_range(Lpre.data(), Spre.data(), r, rank, lo, hi);   // what r holds and `rank` needs
    recv_off[r] = lo - Lpre[rank];
    recv_cnt[r] = hi - lo;
  }
  return MCL_OK;
}

// Particle.matrix_from_tf (auv_particle.py:110-125): 4 x 4 from translation + quaternion (quaternion_matrix of
// tf.transformations: scale by sqrt(2 / |q|^2), outer product)
int matrix_from_tf_impl(const double translation[3], const double quaternion[4], double m16[16]) {
  if (!translation || !quaternion || !m16) return MCL_ERR_INVALID;
  const double* qi = quaternion;
  const double nq = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (nq >= 2.220446049250313e-16 * 4.0) {
    const double s = std::sqrt(2.0 / nq);
    const double q[4] = {qi[0] * s, qi[1] * s, qi[2] * s, qi[3] * s};
    double o[4][4];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) o[a][b] = q[a] * q[b];
    R[0] = 1.0 - o[1][1] - o[2][2];
    R[1] = o[0][1] - o[2][3];
    R[2] = o[0][2] + o[1][3];
    R[3] = o[0][1] + o[2][3];
    R[4] = 1.0 - o[0][0] - o[2][2];
    R[5] = o[1][2] - o[0][3];
    R[6] = o[0][2] - o[1][3];
    R[7] = o[1][2] + o[0][3];
    R[8] = 1.0 - o[0][0] - o[1][1];
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m16[r * 4 + c] = R[r * 3 + c];
    m16[r * 4 + 3] = translation[r];
  }
  m16[12] = m16[13] = m16[14] = 0.0;
  m16[15] = 1.0;
  return MCL_OK;
}

}  // namespace
