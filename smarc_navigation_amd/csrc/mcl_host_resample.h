// mcl_host_resample.h -- host side, part 2: the resample pipeline as phases over a set of shards (max -> quantise ->
// totals -> CDF / expansion -> exchange -> gather), the O(n)-per-rank exchange, the alternative schemes.
#pragma once
#include "mcl_host.h"

namespace {

// ------------------------------------------------------------------------------------------
// resample pipeline, written over a set of shards so that the RCCL path (one shard per process)
// and the LOCAL test group (several shards in one process) execute the same phases.
// ------------------------------------------------------------------------------------------
u64* ctrl_slots(mcl_handle* h) { return (u64*)(h->ctrl + (h->slot_set ? CTRL_SLOTS2 : CTRL_SLOTS)); }
u32* ctrl_u32(mcl_handle* h, int off) { return (u32*)(h->ctrl + off); }

// max lw into the slots (unless the update kernel that wrote lw already did it)
int ensure_max_slots(mcl_handle* h) {
  if (h->max_valid) return MCL_OK;
  h->slot_set = 0;
  HIPCHK(h, hipMemsetAsync(h->ctrl + CTRL_SLOTS, 0, 8 * MCL_MAX_SLOTS, h->stream));
  k_max_slots<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->lw, h->n, ctrl_slots(h));
  HIPCHK(h, hipGetLastError());
  h->max_valid = true;
  return MCL_OK;
}
// scal[0] = local max lw (the value the shard exchange reduces)
int phase_local_max(mcl_handle* h) {
  RET_IF(set_device(h));
  t_begin(h, MCL_K_NORMALISE);
  RET_IF(ensure_max_slots(h));
  k_max_finish<<<1, 64, 0, h->stream>>>(ctrl_slots(h), h->scal);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

int exchange_max(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      NCCLCHK(h, ncclAllReduce(h->scal, h->scal, 1, ncclDouble, ncclMax, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  double m = -INFINITY;
  for (int s = 0; s < ns; ++s) {
    double v;
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(&v, sh[s]->scal, sizeof(double), hipMemcpyDeviceToHost, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    if (v > m) m = v;
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->scal, &m, sizeof(double), hipMemcpyHostToDevice, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}

// fixed-point weights, tile sums, exclusive tile offsets and the shard total in ONE launch.
// from_slots: single shard -- the kernel reads the maximum straight from the slots (no k_max_finish)
// fused_next: k_cdf_expand<true> follows and adds the tile sums up itself (no tile scan launch)
int phase_quantise(mcl_handle* h, bool from_slots, bool fused_next = false) {
  RET_IF(set_device(h));
  const double scale = std::ldexp(1.0, 63 - ceil_log2(h->ng));
  t_begin(h, MCL_K_NORMALISE);
  if (from_slots) RET_IF(ensure_max_slots(h));
  QuantArgs a;
  a.lw = h->lw;
  a.n = h->n;
  a.slots = from_slots ? ctrl_slots(h) : nullptr;
  a.m_lw = h->scal;
  a.mode = h->weight_mode;
  a.scale = scale;
  a.s = 63 - ceil_log2(h->ng);
  a.q = h->q;
  a.tile_sum = h->tile64;
  a.rec = nullptr;
  a.tile_bits = nullptr;
  h->qshift_cur = nullptr;
  k_quantise_tiles<<<(unsigned)h->ntiles_loc, MCL_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  if (!fused_next) {
    // sharded / explicit-position schemes: exclusive tile offsets and the shard total as separate arrays
    t_begin(h, MCL_K_SCAN);
    k_scan_tile_sums<u64><<<1, 1024, 0, h->stream>>>(h->tile64, h->ntiles_loc, h->totals + h->rank);
    t_end(h);
  }
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// ---- one shard per process: "maximum, then totals" in ONE collective.  The shard quantises at the
// exponent of its OWN maximum and leaves, beside the weights, how many of them have each bit set; the all-gather of
// those records (exponent + 8 x 64 partial bit counts: 4 KiB per rank) tells every rank the cloud's exponent, every shard's shift and -- exactly -- every
// shard's total at that shift; k_shift_scan turns them into the totals / tile offsets / shift the CDF kernels read.
// Weight modes other than log-likelihoods (GPS: linear weights relative to the maximum's) keep the two-step form.
bool one_collective(const mcl_handle* h) { return h->comm && h->weight_mode == MCL_WEIGHT_LOG_SHIFT; }
int alloc_shrec(mcl_handle* h) {
  if (h->shrec) return MCL_OK;
  const size_t words = (size_t)h->world * SHREC_WORDS + 1;
  HIPCHK(h, hipMalloc(&h->shrec, sizeof(u64) * words));
  HIPCHK(h, hipMemsetAsync(h->shrec, 0, sizeof(u64) * words, h->stream));
  HIPCHK(h, hipMalloc(&h->tile_bits, sizeof(unsigned short) * 64 * (size_t)h->ntiles_loc));
  return MCL_OK;
}
int phase_quantise_shard(mcl_handle* h) {
  RET_IF(set_device(h));
  RET_IF(alloc_shrec(h));
  u64* mine = h->shrec + (size_t)h->rank * SHREC_WORDS;
  if (h->shrec_dirty)   // (an earlier resample failed between its quantise launch and its k_shift_scan)
    HIPCHK(h, hipMemsetAsync(mine, 0, sizeof(u64) * SHREC_WORDS, h->stream));
  t_begin(h, MCL_K_NORMALISE);
  RET_IF(ensure_max_slots(h));
  QuantArgs a;
  a.lw = h->lw;
  a.n = h->n;
  a.slots = ctrl_slots(h);   // the SHARD's maximum
  a.m_lw = h->scal;
  a.mode = h->weight_mode;
  a.scale = std::ldexp(1.0, 63 - ceil_log2(h->ng));
  a.s = 63 - ceil_log2(h->ng);
  a.q = h->q;
  a.tile_sum = h->tile64;
  a.rec = mine;
  a.tile_bits = h->tile_bits;
  h->shrec_dirty = true;
  k_quantise_shard<<<(unsigned)h->ntiles_loc, MCL_SCAN_TILE, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int exchange_shard_records(mcl_handle* h) {
  t_begin(h, MCL_K_COMM_RECORDS);
  NCCLCHK(h, ncclAllGather(h->shrec + (size_t)h->rank * SHREC_WORDS, h->shrec, SHREC_WORDS, ncclUint64, h->comm, h->stream));
  t_end(h);
  return MCL_OK;
}
int phase_shift_scan(mcl_handle* h) {
  RET_IF(set_device(h));
  ShiftArgs a;
  a.recs = h->shrec;
  a.rank = h->rank;
  a.world = h->world;
  a.tile_bits = h->tile_bits;
  a.ntiles = h->ntiles_loc;
  a.tile_off = h->tile64;
  a.totals = h->totals;
  a.shift_out = h->shrec + (size_t)h->world * SHREC_WORDS;
  // the last fused step's moments came with the records: their sum goes straight into the ring entry reserved for them
  a.mom_out = (h->mom_pending && h->host_pin_dev) ? h->host_pin_dev + RING_STRIDE * (h->mom_pending_entry % MEAN_RING) : nullptr;
  t_begin(h, MCL_K_SCAN);
  k_shift_scan<<<1, 1024, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  if (a.mom_out) h->mom_pending = false;
  h->shrec_dirty = false;
  h->qshift_cur = a.shift_out;
  return MCL_OK;
}

int exchange_totals(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM_RECORDS);
      NCCLCHK(h, ncclAllGather(h->totals + h->rank, h->totals, 1, ncclUint64, h->comm, h->stream));
      t_end(h);
    }
    return MCL_OK;
  }
  std::vector<u64> t(ns);
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(&t[s], sh[s]->totals + s, sizeof(u64), hipMemcpyDeviceToHost, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(sh[s]));
    HIPCHK(sh[s], hipMemcpyAsync(sh[s]->totals, t.data(), sizeof(u64) * ns, hipMemcpyHostToDevice, sh[s]->stream));
    HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  }
  return MCL_OK;
}

int phase_cdf(mcl_handle* h, uint64_t u53) {
  RET_IF(set_device(h));
  CdfArgs a;
  a.totals = h->totals;
  a.rank = h->rank;
  a.world = h->world;
  a.n_global = (u64)h->ng;
  a.u53 = u53;
  a.shift = h->qshift_cur;
  t_begin(h, MCL_K_SCAN);
  k_offspring_cdf<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->tile64, a,
                                                                          h->ncum + h->goff);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// all-gather of the offspring CDF slices and of the pre-resample state (north_star: "all-gather
// before resampling"); shards are contiguous and equal-sized
int exchange_cdf_state(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    if (h->comm) {
      t_begin(h, MCL_K_COMM);
      if (h->gather_inflight) {
        // the state went out right after predict and travelled under the measurement update
        NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
        h->gather_inflight = false;
      } else {
        h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
        NCCLCHK(h, ncclGroupStart());
        NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
        for (int c = 0; c < 6; ++c)
          if (!((h->gather_uni_mask >> c) & 1u))
            NCCLCHK(h, ncclAllGather(h->state[h->cur] + (size_t)c * h->n, h->state_glob + (size_t)c * h->ng,
                                     (size_t)h->n, ncclDouble, h->comm, h->stream));
        NCCLCHK(h, ncclGroupEnd());
      }
      t_end(h);
    }
    return MCL_OK;
  }
  for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  for (int d = 0; d < ns; ++d) {
    mcl_handle* D = sh[d];
    D->gather_uni_mask = D->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
    RET_IF(set_device(D));
    for (int s = 0; s < ns; ++s) {
      mcl_handle* S = sh[s];
      if (s != d)
        HIPCHK(D, hipMemcpyAsync(D->ncum + S->goff, S->ncum + S->goff, sizeof(u32) * (size_t)S->n,
                                 hipMemcpyDefault, D->stream));
      for (int c = 0; c < 6; ++c)
        if (!((D->gather_uni_mask >> c) & 1u))
          HIPCHK(D, hipMemcpyAsync(D->state_glob + (size_t)c * D->ng + S->goff, S->state[S->cur] + (size_t)c * S->n,
                                   sizeof(double) * (size_t)S->n, hipMemcpyDefault, D->stream));
    }
    HIPCHK(D, hipStreamSynchronize(D->stream));
  }
  return MCL_OK;
}

// The pre-resample state is final once predict has run (updates only read it): send it on the
// second communicator/stream so the 48 B x N_global all-gather overlaps the ray-cast.
// an overlapped gather that will not be consumed (error return, state overwritten by the caller):
// let it finish, then forget it, so the next resample gathers the state it actually resamples
int cancel_state_gather(mcl_handle* h) {
  if (!h->gather_inflight) return MCL_OK;
  h->gather_inflight = false;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
  return MCL_OK;
}

int start_state_gather(mcl_handle* h) {
  if (!h->comm2 || !h->state_glob) return MCL_OK;
  HIPCHK(h, hipEventRecord(h->ev_state_ready, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_state_ready, 0));
  // 24 B instead of 48 B per particle of the GLOBAL cloud when z, roll, pitch are the odometry's on every particle
  h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;
  NCCLCHK(h, ncclGroupStart());
  for (int c = 0; c < 6; ++c)
    if (!((h->gather_uni_mask >> c) & 1u))
      NCCLCHK(h, ncclAllGather(h->state[h->cur] + (size_t)c * h->n, h->state_glob + (size_t)c * h->ng, (size_t)h->n,
                               ncclDouble, h->comm2, h->comm_stream));
  NCCLCHK(h, ncclGroupEnd());
  HIPCHK(h, hipEventRecord(h->ev_gather_done, h->comm_stream));
  h->gather_inflight = true;
  return MCL_OK;
}

// lost-slot ranks + dupes list.  fused_cdf: single shard, the offspring CDF is computed in the same pass
int phase_expand(mcl_handle* h, bool fused_cdf, uint64_t u53) {
  RET_IF(set_device(h));
  ExpandArgs a;
  memset(&a, 0, sizeof a);
  a.q = h->q;
  a.tile_sum = h->tile64;
  a.n_fine = h->ntiles_loc;
  a.n_global_u = (u64)h->ng;
  a.u53 = u53;
  a.total_out = h->totals + h->rank;
  a.ncum = h->ncum;
  a.n = h->ng;
  a.own0 = h->goff;
  a.own_n = h->n;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.desc = h->desc;
  a.ticket = ctrl_u32(h, CTRL_T_EXPAND);
  a.epoch = ++h->epoch;
  const unsigned grid = (unsigned)((h->ng + RS_TILE - 1) / RS_TILE);
  t_begin(h, fused_cdf ? MCL_K_SCAN : MCL_K_RESAMPLE);
  if (fused_cdf)
    k_cdf_expand<true><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  else
    k_cdf_expand<false><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// ---- O(n)-per-rank exchange -------------------------------------------------------------------------------
int alloc_lsx(mcl_handle* h) {
  if (h->lsx) return MCL_OK;
  HIPCHK(h, hipMalloc(&h->lsx, sizeof(u64) * 4 * (size_t)h->world));
  HIPCHK(h, hipHostMalloc(&h->lsx_host, sizeof(u64) * (4 * (size_t)h->world + 1), hipHostMallocMapped | hipHostMallocCoherent));  // (fine-grained: the host polls it while kernels run)
  memset(h->lsx_host, 0, sizeof(u64) * (4 * (size_t)h->world + 1));
  if (hipHostGetDevicePointer((void**)&h->lsx_host_dev, h->lsx_host, 0) != hipSuccess) h->lsx_host_dev = nullptr;
  return MCL_OK;
}
int launch_pack(mcl_handle* h, u64 publish_seq = 0);
// CDF, lost ranks and dupes list of THIS shard only (k_cdf_expand<true> over the shard, global weight offsets from the
// all-gathered totals); leaves the shard's hand-over record in lsx[rank]
int phase_expand_local(mcl_handle* h, uint64_t u53) {
  RET_IF(set_device(h));
  RET_IF(alloc_lsx(h));
  ExpandArgs a;
  memset(&a, 0, sizeof a);
  a.q = h->q;
  a.tile_sum = h->tile64;  // exclusive offsets (k_scan_tile_sums)
  a.n_fine = h->ntiles_loc;
  a.n_global_u = (u64)h->ng;
  a.u53 = u53;
  a.total_out = h->totals + h->world;  // (unused in this mode)
  a.ncum = h->ncum + h->goff;
  a.n = h->n;
  a.own0 = 0;
  a.own_n = h->n;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.desc = h->desc;
  a.ticket = ctrl_u32(h, CTRL_T_EXPAND);
  a.epoch = ++h->epoch;
  a.totals = h->totals;
  a.rank = h->rank;
  a.world = h->world;
  a.ls_out = h->lsx + 4 * (size_t)h->rank;
  a.shift = h->qshift_cur;
  for (int c = 0; c < 3; ++c) a.p0[c] = h->state[h->cur] + (size_t)c * h->n;
  // (after a fused predict the state's z words are not written yet -- the moments' shift takes the uniform
  //  depth itself, like the unsharded gather does, so sharded and unsharded means agree to the last bit)
  a.p0_z_uniform = h->uni_valid ? 1 : 0;
  a.p0_z = h->uni_val[0];
  const unsigned grid = (unsigned)((h->n + RS_TILE - 1) / RS_TILE);
  t_begin(h, MCL_K_SCAN);
  k_cdf_expand<true><<<grid, RS_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cdf_global = h->world == 1;
  return MCL_OK;
}

// every shard learns {L, S} of every shard (and the position of global particle 0); the host needs them to size the
// point-to-point transfers: ONE stream synchronisation per resample
int exchange_ls(mcl_handle** sh, int ns) {
  const int world = sh[0]->world;
  if (ns == 1) {
    mcl_handle* h = sh[0];
    t_begin(h, MCL_K_COMM_RECORDS);
    if (h->comm && world > 1)
      NCCLCHK(h, ncclAllGather(h->lsx + 4 * (size_t)h->rank, h->lsx, 4, ncclUint64, h->comm, h->stream));
    if (h->lsx_host_dev) {
      // the records travel to pinned memory by a kernel that writes a sequence word last; the pack kernel (sized on
      // the device from the same records) is queued behind it BEFORE the host starts to wait, so the GPU keeps
      // working while the host wakes up; the host spins on the word instead of synchronising the stream
      const u64 seq = ++h->ls_seq;
      t_end(h);
      RET_IF(launch_pack(h, seq));   // (its first workgroup publishes the records before it packs)
      volatile u64* flag = h->lsx_host + 4 * (size_t)world;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xfffu) == 0u) {
          if (hipStreamQuery(h->stream) == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
            return fail(h, MCL_ERR_HIP, "resample exchange: the hand-over records never arrived");
          if (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > 60)
            return fail(h, MCL_ERR_COMM, "resample exchange: timed out waiting for the hand-over records");
        }
        __builtin_ia32_pause();
      }
    } else {
      HIPCHK(h, hipMemcpyAsync(h->lsx_host, h->lsx, sizeof(u64) * 4 * (size_t)world, hipMemcpyDeviceToHost, h->stream));
      t_end(h);
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  } else {
    for (int s = 0; s < ns; ++s) {
      mcl_handle* h = sh[s];
      RET_IF(set_device(h));
      HIPCHK(h, hipMemcpyAsync(h->lsx_host + 4 * (size_t)s, h->lsx + 4 * (size_t)s, sizeof(u64) * 4, hipMemcpyDeviceToHost, h->stream));
    }
    for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
    for (int d = 0; d < ns; ++d) {
      for (int s = 0; s < ns; ++s)
        if (s != d) memcpy(sh[d]->lsx_host + 4 * (size_t)s, sh[s]->lsx_host + 4 * (size_t)s, sizeof(u64) * 4);
      RET_IF(set_device(sh[d]));
      HIPCHK(sh[d], hipMemcpyAsync(sh[d]->lsx, sh[d]->lsx_host, sizeof(u64) * 4 * (size_t)ns, hipMemcpyHostToDevice, sh[d]->stream));
    }
  }
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = sh[s];
    h->ex_L.assign(world, 0u);
    h->ex_S.assign(world, 0u);
    h->ex_Lpre.assign(world + 1, 0u);
    h->ex_Spre.assign(world + 1, 0u);
    for (int r = 0; r < world; ++r) {
      h->ex_L[r] = (u32)(h->lsx_host[4 * (size_t)r] & 0xffffffffull);
      h->ex_S[r] = (u32)(h->lsx_host[4 * (size_t)r] >> 32);
      h->ex_Lpre[r + 1] = h->ex_Lpre[r] + h->ex_L[r];
      h->ex_Spre[r + 1] = h->ex_Spre[r] + h->ex_S[r];
    }
    if (h->ex_Lpre[world] != h->ex_Spre[world])
      return fail(h, MCL_ERR_COMM, "resample exchange: lost slots and surplus copies of the shards do not add up (ranks fed different inputs?)");
  }
  return MCL_OK;
}

// surplus copies into the send buffer (own lost slots: straight into the receive buffer).  The kernel takes its sizes
// from the records on the device, so it can be queued before the host has read them (exchange_ls)
int launch_pack(mcl_handle* h, u64 publish_seq) {
  RET_IF(set_device(h));
  if (!h->xrecv) HIPCHK(h, hipMalloc(&h->xrecv, sizeof(double) * 6 * (size_t)h->n));
  if (!h->xsend) {
    // a shard's surplus is statistically a few per cent of its slots; n / 4 entries to start with, grown on demand
    const size_t cap = std::max<size_t>((size_t)h->n / 4, 4096);
    HIPCHK(h, hipMalloc(&h->xsend, sizeof(double) * 6 * cap));
    h->xsend_cap = cap;
  }
  h->gather_uni_mask = h->uni_valid ? 0x1cu : 0u;  // z, roll, pitch: the same on every particle of every shard
  PackArgs a;
  a.src = state_ptrs(h->state[h->cur], h->n);
  a.dupes = h->dupes32;
  a.lsx = h->lsx;
  a.rank = h->rank;
  a.world = h->world;
  a.cap = (u32)std::min<size_t>(h->xsend_cap, 0xffffffffull);
  a.nship = 0;
  for (int c = 0; c < 6; ++c) {
    a.ship[c] = 0;
    if (!((h->gather_uni_mask >> c) & 1u)) a.ship[a.nship++] = c;
  }
  h->ex_nship = a.nship;
  a.send = h->xsend;
  a.recv = h->xrecv;
  a.host_words = publish_seq ? h->lsx_host_dev : nullptr;
  a.host_seq = publish_seq ? h->lsx_host_dev + 4 * (size_t)h->world : nullptr;
  a.seq = publish_seq;
  t_begin(h, MCL_K_PACK);
  k_pack_dupes<<<(unsigned)std::min<long long>(grid_for(h->n), 512), MCL_BLOCK, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
// after the host has read the records: the rare shard whose surplus exceeds the send buffer grows it and packs again;
// LOCAL groups pack here in the first place
int phase_pack(mcl_handle* h, bool already_packed) {
  RET_IF(set_device(h));
  const u32 S = h->ex_S[h->rank];
  if (already_packed && (size_t)S <= h->xsend_cap) return MCL_OK;
  if ((size_t)S > h->xsend_cap) {
    if (h->xsend) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      (void)hipFree(h->xsend);
      h->xsend = nullptr;
    }
    const size_t cap = std::max<size_t>((size_t)S + (size_t)S / 4, 4096);
    HIPCHK(h, hipMalloc(&h->xsend, sizeof(double) * 6 * cap));
    h->xsend_cap = cap;
  }
  return launch_pack(h);
}

void ex_range(const mcl_handle* h, int from, int to, u32& lo, u32& hi) {
  plan_range(h->ex_Lpre.data(), h->ex_Spre.data(), from, to, lo, hi);
}

int exchange_dupes(mcl_handle** sh, int ns) {
  if (ns == 1) {
    mcl_handle* h = sh[0];
    h->ex_lost += h->ex_L[h->rank];
    ++h->ex_rounds;
    if (!h->comm || h->world == 1) return MCL_OK;
    const int q = h->rank;
    const size_t K = (size_t)h->ex_nship;
    t_begin(h, MCL_K_COMM_P2P);
    NCCLCHK(h, ncclGroupStart());
    for (int r = 0; r < h->world; ++r) {
      if (r == q) continue;
      u32 lo, hi;
      ex_range(h, q, r, lo, hi);  // what I hold and r needs
      // (ONE send and ONE receive per peer: a copy is one record of K doubles, the range a peer needs is contiguous)
      if (hi > lo) {
        h->ex_sent += hi - lo;
        NCCLCHK(h, ncclSend(h->xsend + (size_t)(lo - h->ex_Spre[q]) * K, (size_t)(hi - lo) * K, ncclDouble, r, h->comm, h->stream));
        ++h->ex_ops;
      }
      ex_range(h, r, q, lo, hi);  // what r holds and I need
      if (hi > lo) {
        NCCLCHK(h, ncclRecv(h->xrecv + (size_t)(lo - h->ex_Lpre[q]) * K, (size_t)(hi - lo) * K, ncclDouble, r, h->comm, h->stream));
        ++h->ex_ops;
      }
    }
    NCCLCHK(h, ncclGroupEnd());
    t_end(h);
    return MCL_OK;
  }
  // LOCAL group: the same ranges as device copies (after every shard has packed)
  for (int s = 0; s < ns; ++s) HIPCHK(sh[s], hipStreamSynchronize(sh[s]->stream));
  for (int d = 0; d < ns; ++d) {
    mcl_handle* D = sh[d];
    RET_IF(set_device(D));
    D->ex_lost += D->ex_L[d];
    ++D->ex_rounds;
    for (int s = 0; s < ns; ++s) {
      if (s == d) continue;
      mcl_handle* S = sh[s];
      u32 lo, hi;
      ex_range(D, s, d, lo, hi);
      if (hi <= lo) continue;
      S->ex_sent += hi - lo;
      ++S->ex_ops;
      ++D->ex_ops;
      const size_t K = (size_t)D->ex_nship;
      HIPCHK(D, hipMemcpyAsync(D->xrecv + (size_t)(lo - D->ex_Lpre[d]) * K, S->xsend + (size_t)(lo - S->ex_Spre[s]) * K,
                               sizeof(double) * (size_t)(hi - lo) * K, hipMemcpyDefault, D->stream));
    }
  }
  for (int d = 0; d < ns; ++d) HIPCHK(sh[d], hipStreamSynchronize(sh[d]->stream));
  return MCL_OK;
}

// the global offspring CDF on demand (mcl_get_last_indices / mcl_get_last_offspring_cdf after an O(n) exchange, which
// leaves only the shard's own slice): RCCL -- a COLLECTIVE all-gather, every rank must make the call; LOCAL group --
// copies from the peers' slices
int ensure_global_cdf(mcl_handle* h) {
  if (h->cdf_global || h->world == 1) return MCL_OK;
  RET_IF(set_device(h));
  if (h->comm) {
    NCCLCHK(h, ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream));
  } else if ((int)h->group.size() == h->world) {
    for (mcl_handle* S : h->group) {
      if (S == h) continue;
      HIPCHK(S, hipStreamSynchronize(S->stream));
      HIPCHK(h, hipMemcpyAsync(h->ncum + S->goff, S->ncum + S->goff, sizeof(u32) * (size_t)S->n, hipMemcpyDefault, h->stream));
    }
  } else {
    return fail(h, MCL_ERR_STATE, "the global offspring CDF needs the communicator or the LOCAL group of the last resample");
  }
  h->cdf_global = true;
  return MCL_OK;
}

// reassign gather + resampling noise (+ the sums of update_loc_pose of the new state when with_moments)
int phase_gather(mcl_handle* h, const double* replay_normals, bool with_moments) {
  RET_IF(set_device(h));
  if (replay_normals) RET_IF(upload_replay(h, replay_normals));
  GatherArgs a;
  const bool multi = h->world > 1 || h->comm;
  const bool p2p = multi && !h->exch_allgather;
  a.src = (multi && !p2p) ? state_ptrs(h->state_glob, h->ng) : state_ptrs(h->state[h->cur], h->n);
  a.recv_mode = p2p ? 1 : 0;
  a.recv_stride = 1;
  a.recv = a.src;
  if (p2p) {   // records of ex_nship doubles (k_pack_dupes): component c is word j of its copy's record
    a.recv_stride = h->ex_nship;
    int j = 0;
    for (int c = 0; c < 6; ++c) {
      a.recv.c[c] = h->xrecv + j;
      if (!((h->gather_uni_mask >> c) & 1u)) ++j;
    }
  }
  a.shift_dev = p2p ? (const double*)(h->lsx + 1) : nullptr;
  a.dst = state_ptrs(h->state[h->cur ^ 1], h->n);
  a.n = h->n;
  a.goff = h->goff;
  a.zr = h->zr;
  a.dupes = h->dupes32;
  a.nz = noise_args(h, h->cfg.resample_cov, 2u, h->step_resample);
  a.add_noise = 1;
  a.part = h->part;
  a.sums_out = h->scal + 32;
  // one process per GPU, log-likelihood weights: the sums go into the shard's record and travel with the NEXT step's
  // all-gather (collect_fused_moments reserves their ring entry; a reader that comes first flushes them by an all-reduce)
  static const bool ride_env = !(getenv("MCL_MOMENTS_RIDE") && atoi(getenv("MCL_MOMENTS_RIDE")) == 0);   // (A/B switch)
  h->moments_ride = with_moments && ride_env && one_collective(h) && h->shrec && h->host_pin_dev && h->qshift_cur != nullptr;
  if (h->moments_ride) a.sums_out = (double*)(h->shrec + (size_t)h->rank * SHREC_WORDS + SHREC_MOM);
  // (single shard: the gather reads straight from the pre-resample state; after a predict z, roll, pitch are the same
  //  three numbers on every particle, so they are substituted instead of read -- bit-identical, 24 B x N less traffic)
  a.uni_mask = multi ? h->gather_uni_mask : (h->uni_valid ? 0x1cu : 0u);
  for (int c = 0; c < 6; ++c) a.uni[c] = (c >= 2 && c <= 4) ? h->uni_val[c - 2] : 0.0;
  if (with_moments && !multi && h->host_pin_dev) {
    // single shard: the last block writes the sums straight into the pinned ring entry (no copy command);
    // the host reads it only after synchronising the stream
    double* slot = h->host_pin + RING_STRIDE * (h->mean_count % MEAN_RING);
    slot[16] = 1.0;
    a.sums_out = h->host_pin_dev + RING_STRIDE * (h->mean_count % MEAN_RING);
    h->moments_direct = true;
  } else {
    h->moments_direct = false;
  }
  a.ticket = ctrl_u32(h, CTRL_T_GATHER);
  const double* rp = replay_normals ? h->replay_dev : nullptr;
  // the visiting order of the NEXT fan sweep / group slice (mcl_kernels.h: VisitArgs)
  memset(&a.visit, 0, sizeof a.visit);
  h->visit_ready = false;
  // (the stash kernel -- mcl_resample.h -- takes the sums in a second pass: the fused step with resampling noise on
  //  x, y, yaw only; it also prepares the visiting order)
  const bool uni = a.uni_mask == 0x1cu;   // z, roll, pitch substituted: the lean kernel
  // Is a visiting order wanted for the next update?  (the sweep gains 17 % of 0.3 ms from it: shards of >= visit_min_n
  // particles; the slice 30 - 45 % of milliseconds -- its group kernel needs the order --: from 8 192 particles)
  const bool by_size = h->sweep_now ? h->n >= h->visit_min_n : h->n >= 8192;
  const bool order_ok = !rp && (h->sweep_now || (h->slice_now && h->env_slice_group != 0)) && h->env_visit != 0 &&
                        (h->env_visit == 1 || by_size) && h->n <= 63ll * GATHER_MAX_GRID * RS_BLOCK;   // (16-bit counts per (workgroup, bin): a workgroup takes whole chunks of RS_BLOCK slots -- at most 63 of them, 64 512 particles, all of which land in ONE bin on a first gather -- ADVICE r5)
  // (also for a plain mcl_resample right after predict + update -- the node's call sequence --: the sums are then a
  //  by-product nobody reads, the visiting order is what the stash kernel is taken for)
  const bool want_order = !with_moments && order_ok && h->world == 1 && !h->comm;
  const bool stash = (with_moments || want_order) && uni && a.nz.sq[2] == 0.0 && a.nz.sq[3] == 0.0 && a.nz.sq[4] == 0.0;
  const bool visit = stash && order_ok;
  if (visit) {
    const int nb = h->visit_nb[0] * h->visit_nb[1] * h->visit_nb[2];
    if (!h->visit_okey) {
      HIPCHK(h, hipMalloc(&h->visit_okey, sizeof(u32) * (size_t)h->n));
      HIPCHK(h, hipMalloc(&h->visit_base, sizeof(u32) * (size_t)GATHER_MAX_GRID * VISIT_MAX_BINS));
      HIPCHK(h, hipMalloc(&h->visit_cnt, sizeof(unsigned short) * (size_t)GATHER_MAX_GRID * VISIT_MAX_BINS));
      HIPCHK(h, hipMalloc(&h->visit_desc, sizeof(u64) * (VISIT_MAX_BINS / 64)));
      HIPCHK(h, hipMemsetAsync(h->visit_desc, 0, sizeof(u64) * (VISIT_MAX_BINS / 64), h->stream));
      HIPCHK(h, hipMalloc(&h->visit_par, sizeof(VisitPar) * 2));
      HIPCHK(h, hipMemsetAsync(h->visit_par, 0, sizeof(VisitPar) * 2, h->stream));
    }
    a.visit.okey = h->visit_okey;
    a.visit.cnt = h->visit_cnt;
    a.visit.base = h->visit_base;
    a.visit.desc = h->visit_desc;
    a.visit.epoch = ++h->visit_epoch;
    a.visit.par_in = h->visit_par + (h->visit_flip & 1u);
    a.visit.par_out = h->visit_par + ((h->visit_flip & 1u) ^ 1u);
    h->visit_flip ^= 1u;
    a.visit.nbx = h->visit_nb[0];
    a.visit.nby = h->visit_nb[1];
    a.visit.nbw = h->visit_nb[2];
    a.visit.nb = nb;
    a.visit.range = h->visit_range;
    a.visit.psq[0] = (float)std::sqrt(h->cfg.process_cov[0]);
    a.visit.psq[1] = (float)std::sqrt(h->cfg.process_cov[1]);
    a.visit.psq[2] = (float)std::sqrt(h->cfg.process_cov[5]);
    a.visit.pstep = h->step_predict;
    {
      static const bool preview = !(getenv("MCL_VISIT_NOISE") && atoi(getenv("MCL_VISIT_NOISE")) == 0);   // (A/B switch)
      if (!preview) a.visit.psq[0] = a.visit.psq[1] = a.visit.psq[2] = 0.f;
    }
  }
  t_begin(h, MCL_K_RESAMPLE);
  // one particle per thread up to 256 blocks (= 256 tickets), grid-stride beyond
  long long gg = (h->n + RS_BLOCK - 1) / RS_BLOCK;
  gg = gg < 1 ? 1 : (gg > GATHER_MAX_GRID ? GATHER_MAX_GRID : gg);
  if (stash) {
    if (!h->gather_attr_set) {   // (more than 64 KiB of dynamic LDS has to be asked for, once per device)
      HIPCHK(h, hipFuncSetAttribute((const void*)k_resample_gather<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GATHER_STASH_LDS));
      h->gather_attr_set = true;
    }
    k_resample_gather<true, true, true><<<(unsigned)gg, RS_BLOCK, GATHER_STASH_LDS, h->stream>>>(a, rp);
  } else if (with_moments && uni)
    k_resample_gather<true, true><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  else if (with_moments)
    k_resample_gather<true><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  else if (uni)
    k_resample_gather<false, true><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  else
    k_resample_gather<false><<<(unsigned)gg, RS_BLOCK, 0, h->stream>>>(a, rp);
  if (visit) {
    k_visit_scan<<<(unsigned)(a.visit.nb / 64), 1024, 0, h->stream>>>(a.visit, (int)gg, ctrl_u32(h, CTRL_T_VISIT));
    h->visit_ready = true;
  }
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cur ^= 1;
  h->uni_valid = false;  // (the new state carries resampling noise)
  h->uni_deferred = false;
  h->gather_uni_mask = 0u;
  h->step_resample++;
  h->have_cdf = true;
  h->have_lw = false;
  h->pose_ready = false;
  return MCL_OK;
}


// ------------------------------------------------------------------------------------------
// stratified / multinomial / residual (single shard): explicit ancestor vector + generic reassign
// ------------------------------------------------------------------------------------------
template <class T>
int lazy_alloc(mcl_handle* h, T** p, size_t count) {
  if (!*p) HIPCHK(h, hipMalloc(p, sizeof(T) * count));
  return MCL_OK;
}
int alt_alloc(mcl_handle* h) {
  const size_t n = (size_t)h->n;
  RET_IF(lazy_alloc(h, &h->cq, n));
  RET_IF(lazy_alloc(h, &h->u53, n));
  RET_IF(lazy_alloc(h, &h->cnt, n));
  RET_IF(lazy_alloc(h, &h->first, n));
  RET_IF(lazy_alloc(h, &h->flags, n));
  RET_IF(lazy_alloc(h, &h->fcum, n));
  RET_IF(lazy_alloc(h, &h->copies, n));
  RET_IF(lazy_alloc(h, &h->ccum, n));
  RET_IF(lazy_alloc(h, &h->dupes, n));
  RET_IF(lazy_alloc(h, &h->cs, n));
  RET_IF(lazy_alloc(h, &h->chunk, n / 8192 + 2));
  RET_IF(lazy_alloc(h, &h->uni_dev, n));
  RET_IF(lazy_alloc(h, &h->wnorm, n));
  RET_IF(lazy_alloc(h, &h->idx, n));
  return MCL_OK;
}
// inclusive u32 scan of `in` into `out` (n local); uses tile32 as scratch
int scan_u32(mcl_handle* h, const u32* in, u32* out) {
  k_u32_tile_sums<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(in, h->n, h->tile32);
  k_scan_tile_sums<u32><<<1, 1024, 0, h->stream>>>(h->tile32, h->ntiles_loc, h->tile32 + h->ntiles_glob);
  k_u32_scan<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(in, h->n, h->tile32, out);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
// residual: normalise like auv_pf.py:172 (numpy's summation order), copies = floor(N w), k = sum
int residual_prepare(mcl_handle* h) {
  if (h->residual_k >= 0) return MCL_OK;
  RET_IF(alt_alloc(h));
  RET_IF(phase_local_max(h));
  const long long nchunks = (h->n + 8191) / 8192;
  t_begin(h, MCL_K_NORMALISE);
  k_linear_weights<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->lw, h->n, h->scal, h->weight_mode, h->wnorm);
  if (h->weight_mode == MCL_WEIGHT_LINEAR) {
    // free-function form (resampling.py): the caller's weights are used as they are, no renormalisation
    const double one = 1.0;
    HIPCHK(h, hipMemcpyAsync(h->scal + 1, &one, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  } else {
    k_np_chunk_sums<<<(unsigned)((nchunks + 63) / 64), 64, 0, h->stream>>>(h->wnorm, h->n, h->chunk);
    k_np_sum_final<<<1, 64, 0, h->stream>>>(h->chunk, nchunks, h->scal + 1);
  }
  k_residual_copies<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->wnorm, h->n, h->scal + 1, h->copies);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  RET_IF(scan_u32(h, h->copies, h->ccum));
  u32 k = 0;
  HIPCHK(h, hipMemcpyAsync(&k, h->ccum + (h->n - 1), sizeof(u32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->residual_k = k > (u32)h->n ? h->n : (long long)k;
  return MCL_OK;
}
long long uniforms_needed(mcl_handle* h, int* rc) {
  *rc = MCL_OK;
  switch (h->cfg.resample_scheme) {
    case MCL_RESAMPLE_SYSTEMATIC:
    case MCL_RESAMPLE_NAIVE: return 1;
    case MCL_RESAMPLE_STRATIFIED:
    case MCL_RESAMPLE_MULTINOMIAL: return h->n;
    case MCL_RESAMPLE_RESIDUAL:
      *rc = residual_prepare(h);
      return *rc == MCL_OK ? h->n - h->residual_k : 0;
  }
  *rc = MCL_ERR_INVALID;
  return 0;
}
int make_uniforms(mcl_handle* h, const double* uniforms, long long nu, long long need) {
  if (need <= 0) return MCL_OK;
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!uniforms || nu < need) return fail(h, MCL_ERR_INVALID, "resample: not enough replay uniforms for this scheme");
    RET_IF(upload(h, h->uni_dev, uniforms, sizeof(double) * (size_t)need));
    rp = h->uni_dev;
  }
  k_make_u53<<<grid_for(need), MCL_BLOCK, 0, h->stream>>>(rp, need, (u32)h->cfg.seed, (u32)(h->cfg.seed >> 32),
                                                        h->step_resample, h->u53);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}
int alt_indices(mcl_handle* h, const double* uniforms, long long nu) {
  RET_IF(set_device(h));
  RET_IF(alt_alloc(h));
  const int scheme = h->cfg.resample_scheme;
  if (scheme == MCL_RESAMPLE_RESIDUAL) {
    RET_IF(residual_prepare(h));
    const long long k = h->residual_k, need = h->n - k;
    RET_IF(make_uniforms(h, uniforms, nu, need));
    t_begin(h, MCL_K_SCAN);
    if (k > 0) k_residual_head<<<grid_for(k), MCL_BLOCK, 0, h->stream>>>(h->ccum, h->n, k, h->idx);
    if (need > 0) {
      k_residual_cumsum<<<1, 64, 0, h->stream>>>(h->wnorm, h->copies, h->n, h->cs);
      k_residual_searchsorted<<<1, 64, 0, h->stream>>>(h->cs, h->n, h->u53, need, h->idx + k);
    }
    t_end(h);
  } else {
    RET_IF(phase_quantise(h, true));
    RET_IF(make_uniforms(h, uniforms, nu, h->n));
    t_begin(h, MCL_K_SCAN);
    k_u64_scan<<<grid_tiles(h->ntiles_loc), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->tile64, h->cq);
    if (scheme == MCL_RESAMPLE_STRATIFIED)
      k_stratified_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->cq, h->u53, h->n, h->idx);
    else
      k_multinomial_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->cq, h->u53, h->n, h->idx);
    t_end(h);
  }
  HIPCHK(h, hipGetLastError());
  h->idx_explicit = true;
  h->have_cdf = false;
  return MCL_OK;
}
int run_resample_alt(mcl_handle* h, const double* uniforms, long long nu, const double* replay_normals) {
  h->uni_valid = false;
  h->visit_ready = false;  // (single shard only: no exchange; the new state carries resampling noise)
  RET_IF(alt_indices(h, uniforms, nu));
  // keep/lost/dupes for an arbitrary ancestor vector (auv_pf.py:183-198) + noise
  if (replay_normals) RET_IF(upload_replay(h, replay_normals));
  t_begin(h, MCL_K_RESAMPLE);
  HIPCHK(h, hipMemsetAsync(h->cnt, 0, sizeof(u32) * (size_t)h->n, h->stream));
  HIPCHK(h, hipMemsetAsync(h->first, 0xff, sizeof(u32) * (size_t)h->n, h->stream));
  k_idx_hist<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->n, h->cnt, h->first);
  k_flags<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->cnt, h->first, h->n, 0, h->flags);
  RET_IF(scan_u32(h, h->flags, h->zcum));
  k_flags<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->cnt, h->first, h->n, 1, h->flags);
  RET_IF(scan_u32(h, h->flags, h->fcum));
  k_compact_dupes<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->idx, h->flags, h->fcum, h->n, h->dupes);
  ReassignIdxArgs a;
  a.src = state_ptrs(h->state[h->cur], h->n);
  a.dst = state_ptrs(h->state[h->cur ^ 1], h->n);
  a.n = h->n;
  a.nz = noise_args(h, h->cfg.resample_cov, 2u, h->step_resample);
  k_reassign_idx<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(a, h->cnt, h->zcum, h->dupes,
                                                             replay_normals ? h->replay_dev : nullptr);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->cur ^= 1;
  h->step_resample++;
  h->have_cdf = false;
  h->idx_explicit = true;
  h->have_lw = false;
  h->residual_k = -1;
  return MCL_OK;
}

int run_resample(mcl_handle** sh, int ns, const double* uniforms, long long nu,
                 const double* const* replay_normals, bool with_moments = false) {
  mcl_handle* h0 = sh[0];
  for (int s = 0; s < ns; ++s) {
    if (!sh[s]->have_lw) return fail(sh[s], MCL_ERR_STATE, "resample: no weights (call an update first)");
    if (sh[s]->cfg.resample_scheme != MCL_RESAMPLE_SYSTEMATIC && sh[s]->cfg.resample_scheme != MCL_RESAMPLE_NAIVE) {
      if (ns > 1 || sh[s]->world > 1)
        return fail(sh[s], MCL_ERR_UNSUPPORTED, "resample: only the systematic scheme is sharded across GPUs");
      return run_resample_alt(sh[s], uniforms, nu, replay_normals ? replay_normals[0] : nullptr);
    }
  }
  uint64_t u53;
  if (h0->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!uniforms || nu < 1) return fail(h0, MCL_ERR_INVALID, "resample: REPLAY mode needs 1 uniform");
    if (!(uniforms[0] >= 0.0 && uniforms[0] < 1.0)) return fail(h0, MCL_ERR_INVALID, "resample: u not in [0,1)");
    u53 = (uint64_t)std::floor(uniforms[0] * 9007199254740992.0);
  } else {
    u53 = native_u53(h0->cfg.seed, h0->step_resample);
  }
  if (h0->cfg.resample_scheme == MCL_RESAMPLE_NAIVE) u53 |= MCL_U53_NAIVE;  // ">=" at the CDF edges (mcl_device.h)
  // one shard (and few enough tiles for every k_cdf_expand block to add their sums up itself):
  // max from the slots -> quantise -> CDF + expansion -> gather, three launches
  const bool single = ns == 1 && h0->world == 1 && !h0->comm && h0->ntiles_loc <= 8192;
  if (single) {
    h0->cdf_global = true;
    h0->group.clear();
    RET_IF(phase_quantise(h0, true, true));
    RET_IF(phase_expand(h0, true, u53));
    return phase_gather(h0, (replay_normals && h0->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[0] : nullptr,
                        with_moments);
  }
  if (ns == 1 && one_collective(h0)) {
    // one process per GPU, log-likelihood weights: quantise at the shard's own exponent, ONE all-gather of the shards'
    // records, shift (rounds 1-5: all-reduce of the maximum, quantise, all-gather of the totals -- two dependent
    // collectives)
    h0->group.clear();
    RET_IF(phase_quantise_shard(h0));
    RET_IF(exchange_shard_records(h0));
    RET_IF(phase_shift_scan(h0));
  } else if (ns == 1) {
    // one process per GPU: the 64 max-lw slots the update kernel filled are all-reduced as they are (ordered u64
    // keys: the maximum of the keys is the key of the maximum) and the quantise kernel reads them -- no k_max_finish
    h0->group.clear();
    RET_IF(set_device(h0));
    t_begin(h0, MCL_K_NORMALISE);
    RET_IF(ensure_max_slots(h0));
    t_end(h0);
    if (h0->comm && h0->world > 1) {
      t_begin(h0, MCL_K_COMM_RECORDS);
      NCCLCHK(h0, ncclAllReduce(ctrl_slots(h0), ctrl_slots(h0), MCL_MAX_SLOTS, ncclUint64, ncclMax, h0->comm, h0->stream));
      t_end(h0);
    }
    RET_IF(phase_quantise(h0, true));
  } else {
    for (int s = 0; s < ns; ++s) {
      sh[s]->group.assign(sh, sh + ns);
      RET_IF(phase_local_max(sh[s]));
    }
    RET_IF(exchange_max(sh, ns));
    for (int s = 0; s < ns; ++s) RET_IF(phase_quantise(sh[s], false));
  }
  if (!(ns == 1 && one_collective(h0))) RET_IF(exchange_totals(sh, ns));
  if (!h0->exch_allgather) {
    // O(n) per rank: every shard expands its OWN slice, the shards exchange two integers each, and only
    // the surplus copies whose global positions fall into a peer's lost ranks cross a link
    for (int s = 0; s < ns; ++s) RET_IF(phase_expand_local(sh[s], u53));
    RET_IF(exchange_ls(sh, ns));
    for (int s = 0; s < ns; ++s) RET_IF(phase_pack(sh[s], ns == 1 && sh[s]->lsx_host_dev != nullptr));
    RET_IF(exchange_dupes(sh, ns));
    for (int s = 0; s < ns; ++s)
      RET_IF(phase_gather(sh[s], (replay_normals && sh[s]->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[s] : nullptr,
                          with_moments));
    return MCL_OK;
  }
  for (int s = 0; s < ns; ++s) RET_IF(phase_cdf(sh[s], u53));
  RET_IF(exchange_cdf_state(sh, ns));
  for (int s = 0; s < ns; ++s) sh[s]->cdf_global = true;
  for (int s = 0; s < ns; ++s) RET_IF(phase_expand(sh[s], false, 0));
  for (int s = 0; s < ns; ++s)
    RET_IF(phase_gather(sh[s], (replay_normals && sh[s]->cfg.rng_mode == MCL_RNG_REPLAY) ? replay_normals[s] : nullptr,
                        with_moments));
  return MCL_OK;
}

}  // namespace
