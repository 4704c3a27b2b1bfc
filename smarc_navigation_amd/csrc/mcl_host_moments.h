// mcl_host_moments.h -- host side, part 3: mean / covariance of the cloud (two-pass kernels, the sums fused into the
// resample gather, the pinned result ring).
#pragma once
#include "mcl_host_resample.h"

namespace {

// ------------------------------------------------------------------------------------------ mean/cov
int phase_mean_partial(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_MEAN_COV);
  const int g = grid_for(h->n);
  k_mean_partial<<<g, MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->part);
  k_sum_final<<<7, MCL_BLOCK, 0, h->stream>>>(h->part, g, h->scal + 8);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int phase_cov_partial(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_MEAN_COV);
  const int g = grid_for(h->n);
  k_cov_partial<<<g, MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->scal + 8,
                                                1.0 / (double)h->ng, h->part);
  k_sum_final<<<6, MCL_BLOCK, 0, h->stream>>>(h->part, g, h->scal + 16);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int exchange_sums(mcl_handle** sh, int ns, int off, int cnt) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM_MOMENTS);
      NCCLCHK(h, ncclAllReduce(h->scal + off, h->scal + off, cnt, ncclDouble, ncclSum, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  std::vector<double> acc(cnt, 0.0), tmp(cnt);
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(tmp.data(), sh[s]->scal + off, sizeof(double) * cnt, hipMemcpyDeviceToHost,
                                 sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    for (int k = 0; k < cnt; ++k) acc[k] += tmp[k];
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->scal + off, acc.data(), sizeof(double) * cnt, hipMemcpyHostToDevice,
                                 sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}
int run_mean_cov_async(mcl_handle** sh, int ns) {
  for (int s = 0; s < ns; ++s) RET_IF(phase_mean_partial(sh[s]));
  RET_IF(exchange_sums(sh, ns, 8, 7));
  for (int s = 0; s < ns; ++s) RET_IF(phase_cov_partial(sh[s]));
  RET_IF(exchange_sums(sh, ns, 16, 6));
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    RET_IF(set_device(h));
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 0.0;  // format 0: [0..6] sums of the 6 components + wrapped yaw, [8..13] centred second-moment sums
    HIPCHK(h, hipMemcpyAsync(slot, h->scal + 8, sizeof(double) * 14, hipMemcpyDeviceToHost, h->stream));
    h->mean_count++;
    h->have_meancov = true;
  }
  return MCL_OK;
}
// A fused step's moments that still lie, shard by shard, in the records (they ride with the next step's all-gather):
// somebody wants them NOW -- a reader, a shutdown, a step that takes another path.  One all-reduce, like rounds 1-5 did
// after every step.  In a multi-process run this makes the reader a COLLECTIVE call (include/mcl.h).
int flush_pending_moments(mcl_handle* h) {
  if (!h->mom_pending) return MCL_OK;
  h->mom_pending = false;
  double* slot = h->host_pin + RING_STRIDE * (h->mom_pending_entry % MEAN_RING);
  if (!h->comm || !h->shrec) {   // (the communicator is gone: the entry cannot be completed)
    for (int k = 0; k < 16; ++k) slot[k] = NAN;
    return MCL_OK;
  }
  RET_IF(set_device(h));
  const double* mine = (const double*)(h->shrec + (size_t)h->rank * SHREC_WORDS + SHREC_MOM);
  t_begin(h, MCL_K_COMM_MOMENTS);
  HIPCHK(h, hipMemcpyAsync(h->scal + 32 + MOM_COUNT, mine + MOM_COUNT, sizeof(double) * 3, hipMemcpyDeviceToDevice, h->stream));
  NCCLCHK(h, ncclAllReduce(mine, h->scal + 32, MOM_COUNT, ncclDouble, ncclSum, h->comm, h->stream));
  t_end(h);
  t_begin(h, MCL_K_MEAN_COV);
  HIPCHK(h, hipMemcpyAsync(slot, h->scal + 32, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
  t_end(h);
  return MCL_OK;
}

// the sums k_resample_gather<true> left in scal[32..47]: reduce over the shards, queue the copy to the ring
int collect_fused_moments(mcl_handle** sh, int ns) {
  if (ns == 1 && sh[0]->moments_direct) {
    sh[0]->mean_count++;
    sh[0]->have_meancov = true;
    return MCL_OK;
  }
  if (ns == 1 && sh[0]->moments_ride) {
    // no collective now: the sums wait in the record; the ring entry is reserved, k_shift_scan of the next step (or a
    // flush) fills it
    mcl_handle* h = sh[0];
    RET_IF(flush_pending_moments(h));   // (an older entry that no records exchange came for)
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 1.0;
    h->mom_pending = true;
    h->mom_pending_entry = h->mean_count;
    h->mean_count++;
    h->have_meancov = true;
    return MCL_OK;
  }
  RET_IF(exchange_sums(sh, ns, 32, MOM_COUNT));
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    RET_IF(set_device(h));
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 1.0;  // format 1: 13 sums about the shift in [13..15] (mcl_resample.h, k_resample_gather)
    t_begin(h, MCL_K_MEAN_COV);
    HIPCHK(h, hipMemcpyAsync(slot, h->scal + 32, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
    t_end(h);
    h->mean_count++;
    h->have_meancov = true;
  }
  return MCL_OK;
}
// ring entry -> mean pose, arithmetic mean of the wrapped yaw, covariance as auv_pf.py:238-252 lays it out
void finish_mean_cov(const mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9], long long which = -1) {
  const double N = (double)h->ng;
  if (which < 0) which = h->mean_count - 1;
  const double* p = h->host_pin + RING_STRIDE * (which % MEAN_RING);
  double c[6];
  if (p[16] == 0.0) {
    for (int k = 0; k < 6; ++k) mean6[k] = p[k] / N;
    for (int k = 0; k < 6; ++k) c[k] = p[8 + k] / N;
  } else {
    // d = x - shift:  mean = shift + sum(d)/N ;  cov_ab = sum(d_a d_b)/N - (sum d_a / N)(sum d_b / N)
    const double m0 = p[0] / N, m1 = p[1] / N, m2 = p[2] / N;
    mean6[0] = p[13] + m0;
    mean6[1] = p[14] + m1;
    mean6[2] = p[15] + m2;
    for (int k = 3; k < 6; ++k) mean6[k] = p[k] / N;
    c[0] = p[7] / N - m0 * m0;
    c[1] = p[8] / N - m1 * m1;
    c[2] = p[9] / N - m2 * m2;
    c[3] = p[10] / N - m0 * m1;
    c[4] = p[11] / N - m0 * m2;
    c[5] = p[12] / N - m1 * m2;
  }
  if (yaw_mean) *yaw_mean = p[6] / N;
  cov9[0] = c[0];
  cov9[1] = c[3];
  cov9[2] = c[4];
  cov9[3] = c[3];  // only [1,0] mirrored (auv_pf.py:246)
  cov9[4] = c[1];
  cov9[5] = c[5];
  cov9[6] = 0.0;
  cov9[7] = 0.0;
  cov9[8] = c[2];
}

}  // namespace
