// mcl_host_update.h -- host side, part 4: beam tables, the MBES update (fan sweep / traversal stages) and the predict
// launch of the fused step.
#pragma once
#include "mcl_host_moments.h"

namespace {

// ------------------------------------------------------------------------------------------ MBES
int upload_beams(mcl_handle* h, const float* ranges, const float* beam_angles, int B) {
  h->det_ride = nullptr;   // (detections of a fused landmark step that failed before its table upload: the caller's buffer is gone)
  h->det_ride_dev = nullptr;
  if (B > h->beams_cap) {
    if (h->beam_sc) (void)hipFree(h->beam_sc);
    if (h->ranges_dev) (void)hipFree(h->ranges_dev);
    HIPCHK(h, hipMalloc(&h->beam_sc, sizeof(float2) * (size_t)B));
    HIPCHK(h, hipMalloc(&h->ranges_dev, sizeof(float) * (size_t)B));
    h->beams_cap = B;
    h->beam_cache.clear();
  }
  if ((int)h->beam_cache.size() != B || memcmp(h->beam_cache.data(), beam_angles, sizeof(float) * B) != 0) {
    std::vector<float2> sc(B);
    for (int b = 0; b < B; ++b) {
      sc[b].x = (float)std::sin((double)beam_angles[b]);
      sc[b].y = (float)std::cos((double)beam_angles[b]);
    }
    RET_IF(upload(h, h->beam_sc, sc.data(), sizeof(float2) * (size_t)B));
    h->beam_cache.assign(beam_angles, beam_angles + B);
    int lo = 0, hi = 0;
    bool finite = true;
    for (int b = 0; b < B; ++b) {
      if (!(beam_angles[b] == beam_angles[b])) finite = false;
      if (beam_angles[b] < beam_angles[lo]) lo = b;
      if (beam_angles[b] > beam_angles[hi]) hi = b;
    }
    const bool ok = finite && (double)beam_angles[hi] - (double)beam_angles[lo] < 3.0;  // span < pi
    h->beam_lo = ok ? lo : -1;
    h->beam_hi = ok ? hi : -1;
    bool asc = finite;
    for (int b = 1; b < B && asc; ++b) asc = beam_angles[b] >= beam_angles[b - 1];
    h->beams_sorted = asc;
    // the fan sweep walks outward from the nadir on either side: ascending angles, all within 85 degrees of it
    h->sweep_angles_ok = asc && B <= 2048 && beam_angles[0] >= -1.4835f && beam_angles[B - 1] <= 1.4835f;
    h->b_split = 0;
    while (h->b_split < B && beam_angles[h->b_split] < 0.f) ++h->b_split;
  }
  // (the ranges travel with the first launch_mbes of the update: in one copy with the sweep's beam table, or alone)
  if (ranges) {
    h->ranges_host.assign(ranges, ranges + B);
    h->ranges_pending = true;
  } else {
    h->ranges_host.clear();
    h->ranges_pending = false;
  }
  return MCL_OK;
}

// Lanes per particle side of the fan sweep: a small cloud splits a side's beams over 2 or 4 lanes (mcl_sweep.h SUB:
// each resolves its own run of >= 16 beams, starting at the hit of the run's first beam); the GLOBAL particle count
// decides, so every shard sums in the same order.
// (measured, round 3 with the grid's cell walk, the whole fused step in ms -- traversal | sweep with 1 / 2 / 4 lanes per
//  side, 256 beams:
//    grid   4 096: 0.086 | 0.095 0.090 0.083     mesh   4 096: 0.079 | 0.097 0.092 0.075
//    grid   8 192: 0.089 | 0.093 0.085 0.081     mesh   8 192: 0.089 | 0.094 0.090 0.082
//    grid  32 768: 0.158 | 0.094 0.089 0.087     mesh  32 768: 0.157 | 0.095 0.093 0.088
//    grid  65 536: 0.154 | 0.100 0.098 0.101     mesh  65 536: 0.153 | 0.099 0.100 0.098
//    grid 131 072: 0.218 | 0.120 0.125 0.133     mesh 131 072: 0.218 | 0.116 0.122 0.128
//  below 4 096 the traversal wins (128 particles: 0.059 against 0.067); a later run pays one slanted traversal for its
//  start; worth it while the chip is not full)
int sweep_lanes_per_side(const mcl_handle* h, bool with_ranges, int B) {
  int nsub = 1;
  if (with_ranges) {
    if (h->map_kind == 0)
      nsub = h->ng < 49152 ? 4 : (h->ng < 98304 ? 2 : 1);
    else if (h->mesh && h->mesh->heights && !h->force_general_mesh)
      nsub = h->ng < 49152 ? 4 : 1;
    if (h->env_nsub) nsub = h->env_nsub;
    while (nsub > 1 && B / (2 * nsub) < 16) nsub >>= 1;
  }
  return nsub;
}

// beam table of the fan sweep: side-signed tangent, secant, measured range, weight; and per beam the sum of the
// squared normalised residuals against r_max over the beams from it to the end of its side (mcl_sweep.h)
int upload_sweep_beams(mcl_handle* h, bool with_ranges, int B, double sigma, double r_max, int nsub) {
  // one device block, one copy per update: B records | B tail sums | B measured ranges (for the traversal kernels
  // that take the hand-overs)
  const size_t tab_floats = ((size_t)B * 7 + 4 + 1) & ~(size_t)1;   // B records | B tail sums | B measured ranges | first / second tangent of either side | B tail sums per run (| pad to 8 bytes)
  const bool grow_ride = h->det_ride && h->det_ride_n > h->det_ride_cap;
  // (the new capacities are COMMITTED only when both buffers of both halves exist -- ADVICE r5: a failed allocation
  //  used to leave null buffers behind capacities that said there was room, and the next fused landmark step built its
  //  table through a null pinned pointer)
  const int ride_cap = grow_ride ? std::max(h->det_ride_n, 16) : h->det_ride_cap;
  const size_t blk_floats = tab_floats + 6 * (size_t)ride_cap;   // | the ping's landmark detections (fp64), when they ride along
  if (B > h->sweep_cap || grow_ride) {
    const size_t cap_b = (size_t)std::max(B, h->sweep_cap);
    const size_t alloc_floats = ((cap_b * 7 + 4 + 1) & ~(size_t)1) + 6 * (size_t)ride_cap;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    for (int k = 0; k < 2; ++k) {
      if (h->sweep_buf[k]) (void)hipFree(h->sweep_buf[k]);
      if (h->sweep_stage[k]) (void)hipHostFree(h->sweep_stage[k]);
      h->sweep_buf[k] = nullptr;
      h->sweep_stage[k] = nullptr;
      h->stage_used[k] = false;
    }
    h->sweep_cap = 0;       // (nothing is allocated from here until every allocation below has succeeded)
    h->det_ride_cap = 0;
    for (int k = 0; k < 2; ++k) {
      HIPCHK(h, hipMalloc(&h->sweep_buf[k], sizeof(float) * alloc_floats));
      HIPCHK(h, hipHostMalloc(&h->sweep_stage[k], sizeof(float) * alloc_floats, hipHostMallocDefault));
      if (!h->ev_stage[k]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_stage[k], hipEventDisableTiming));
    }
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    h->sweep_cap = (int)cap_b;
    h->det_ride_cap = ride_cap;
  }
  const int sel = (h->sweep_sel ^= 1);
  h->sweep_beams = h->sweep_buf[sel];
  h->sweep_tail = (float*)(h->sweep_beams + B);
  if (h->stage_used[sel]) HIPCHK(h, hipEventSynchronize(h->ev_stage[sel]));   // (two updates old: long complete)
  struct Blk {   // (the table is built straight into the pinned staging buffer)
    float* p;
    float* data() { return p; }
    float& operator[](size_t k) { return p[k]; }
  } blk{h->sweep_stage[sel]};
  float4* tb = (float4*)blk.data();
  float* tail = blk.data() + (size_t)B * 4;
  float* rng = tail + B;
  for (int b = 0; b < B; ++b) tail[b] = 0.f;
  h->det_ride_dev = nullptr;
  if (h->det_ride) {
    memcpy(blk.data() + tab_floats, h->det_ride, sizeof(double) * 3 * (size_t)h->det_ride_n);
    h->det_ride_dev = (const double*)((const float*)h->sweep_buf[sel] + tab_floats);
    h->det_ride = nullptr;
  }
  const float rmaxf = (float)r_max;
  int nvalid = 0;
  for (int b = 0; b < B; ++b) {
    const double ang = (double)h->beam_cache[b];
    const float rm = (with_ranges && (int)h->ranges_host.size() == B) ? h->ranges_host[b] : 0.f;
    const bool valid = rm > 0.f;  // NaN fails the test (as in the cast kernels)
    nvalid += valid ? 1 : 0;
    // the residual's constants: (range_b - r) w with r = min(t / cos a, r_max) is max(z w - t (w / cos a), (z - r_max) w);
    // an invalid beam has all three zero.  Expected-range calls (no measured ranges) keep 1 / cos a in .y
    const double sec = 1.0 / std::cos(ang);
    // .x: the side-signed tangent of the NEXT beam of this beam's side (+inf beyond the last): a merge loop reads the
    // tangent it needs next where it lies, in the record it has just used (mcl_sweep.h)
    {
      const int nb = b < h->b_split ? b - 1 : b + 1;
      tb[b].x = (nb < 0 || nb >= B) ? INFINITY : (float)(std::tan((double)h->beam_cache[nb]) * (b < h->b_split ? -1.0 : 1.0));
    }
    tb[b].y = with_ranges ? (valid ? (float)(sec / sigma) : 0.f) : (float)sec;
    tb[b].z = valid ? (float)((double)rm / sigma) : 0.f;
    tb[b].w = valid ? (float)(((double)rm - (double)rmaxf) / sigma) : 0.f;
    rng[b] = rm;
  }
  auto miss = [&](int b) { return tb[b].w * tb[b].w; };
  float run = 0.f;
  for (int b = B - 1; b >= h->b_split; --b) tail[b] = (run += miss(b));
  run = 0.f;
  for (int b = 0; b < h->b_split; ++b) tail[b] = (run += miss(b));
  // the same sums per RUN of a side's beams (sub-fans: a lane accounts for its own run only; taking them as
  // differences of the side's sums lost digits when r_max is short and the sums are large)
  float* tail_run = blk.data() + (size_t)B * 6 + 4;
  for (int side = 0; side < 2; ++side) {
    const int nb = side ? h->b_split : B - h->b_split;
    const int per = nsub > 1 ? std::max((nb + nsub - 1) / nsub, 2) : std::max(nb, 1);
    for (int first = 0; first < nb; first += per) {
      const int last = std::min(first + per, nb);
      float acc = 0.f;
      for (int k = last - 1; k >= first; --k) {   // k-th beam of the side, counted outward from the nadir
        const int b = side ? h->b_split - 1 - k : h->b_split + k;
        tail_run[b] = (acc += miss(b));
      }
    }
  }
  h->sweep_nvalid = nvalid;
  for (int k = 0; k < 2; ++k) {
    const int bp = h->b_split + k, bm = h->b_split - 1 - k;
    blk[(size_t)B * 6 + 2 * k] = bp < B ? (float)std::tan((double)h->beam_cache[bp]) : INFINITY;
    blk[(size_t)B * 6 + 2 * k + 1] = bm >= 0 ? (float)(-std::tan((double)h->beam_cache[bm])) : INFINITY;
  }
  // the device buffer was last read by the update two before this one; the copy waits for the event of the update
  // just before (later on the same stream, so certainly enough -- whatever an error path did to the alternation): it
  // then runs under that step's normalise / scan / gather kernels.  The compute stream waits for the copy.
  if (h->ev_upd[0] && h->upd_seq >= 1) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_upd[(h->upd_seq - 1) & 3], 0));
  HIPCHK(h, hipMemcpyAsync(h->sweep_beams, blk.data(), sizeof(float) * (h->det_ride_dev ? blk_floats : tab_floats), hipMemcpyHostToDevice,
                           h->copy_stream));
  HIPCHK(h, hipEventRecord(h->ev_stage[sel], h->copy_stream));
  h->stage_used[sel] = true;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_stage[sel], 0));
  h->ranges_ptr = h->sweep_tail + B;
  h->ranges_pending = false;
  return MCL_OK;
}

void rot_rpy(double roll, double pitch, double yaw, double R[9]) {
  double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
  double cy = std::cos(yaw), sy = std::sin(yaw);
  R[0] = cy * cp;
  R[1] = cy * sp * sr - sy * cr;
  R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp;
  R[4] = sy * sp * sr + cy * cr;
  R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;
  R[7] = cp * sr;
  R[8] = cp * cr;
}

// Morton visiting order of the particles' pose records (k_mbes_keys): h->mbes_perm
int sort_visiting_order(mcl_handle* h, const MbesArgs& a) {
  const size_t n = (size_t)h->n;
  if (!h->sort_keys) {
    HIPCHK(h, hipMalloc(&h->sort_keys, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->sort_keys_out, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->sort_idx, sizeof(u32) * n));
    HIPCHK(h, hipMalloc(&h->mbes_perm, sizeof(u32) * n));
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, h->sort_tmp_bytes, h->sort_keys, h->sort_keys_out, h->sort_idx,
                                        h->mbes_perm, n, 0, 24, h->stream));
    HIPCHK(h, hipMalloc(&h->sort_tmp, h->sort_tmp_bytes));
  }
  k_mbes_keys<<<grid_for(h->n), 256, 0, h->stream>>>(a, h->sort_keys, h->sort_idx);
  // stable LSD radix sort of (key, slot) pairs: the visiting order is deterministic
  HIPCHK(h, rocprim::radix_sort_pairs(h->sort_tmp, h->sort_tmp_bytes, h->sort_keys, h->sort_keys_out, h->sort_idx,
                                      h->mbes_perm, n, 0, 24, h->stream));
  return MCL_OK;
}

int launch_mbes(mcl_handle* h, bool with_ranges, int B, double sigma, double r_max, const double sensor_offset[6],
                double* lw_out, float* exp_out, long long exp_first, long long exp_count, bool pose_done = false,
                MbesArgs* args_only = nullptr) {
  if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, "update_mbes: no map (call mcl_set_map_grid/mesh first)");
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = sensor_offset ? sensor_offset : zero6;
  MbesArgs a;
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int k = 0; k < 12; ++k) a.m2o[k] = h->cfg.m2o[k];
  for (int k = 0; k < 3; ++k) a.off_t[k] = so[k];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.beam_sc = h->beam_sc;
  a.ranges = nullptr;  // (set below, once the ranges are on the device)
  a.n_beams = B;
  a.sorted = h->beams_sorted ? 1 : 0;
  a.b_lo = h->beam_lo;
  a.b_hi = h->beam_hi;
  a.inv_sigma = (float)(1.0 / sigma);
  a.r_max = (float)r_max;
  a.lognorm = std::log(sigma * std::sqrt(2.0 * MCL_PI));
  a.lw = lw_out;
  a.exp_out = exp_out;
  a.exp_first = exp_first;
  a.exp_count = exp_count;
  if (!h->pose_dev) HIPCHK(h, hipMalloc(&h->pose_dev, sizeof(MbesPose) * (size_t)h->n));
  a.pose = h->pose_dev;
  memset(&a.mesh, 0, sizeof a.mesh);
  a.stats = nullptr;
  a.cells = 0;
  a.perm = nullptr;
  a.diag_mode = 0;
  a.sweep_beams = nullptr;
  a.sweep_tail = nullptr;
  a.b_split = 0;
  a.sweep_nvalid = 0;
  a.sweep_nsub = 1;
  a.sweep_noclamp = 0;
  a.sweep_tan0 = nullptr;
  a.sweep_tail_run = nullptr;
  a.sweep_c2z_min = 2.f;
  a.sweep_slope = 0.f;
  a.defer_idx = nullptr;
  a.defer_count = (int*)(h->ctrl + CTRL_DEFER);
  a.n_dev = nullptr;
  a.host_count = nullptr;
  a.reasons = nullptr;
  a.slice = 0;
  a.visit_okey = a.visit_base = nullptr;
  a.visit_nb = 0;
  a.slice_loose = nullptr;
  a.slice_loose_count = nullptr;
  a.in_list = nullptr;
  a.in_count = nullptr;
  a.grid_pad = nullptr;
  a.nyp = 0;
#ifdef SWEEP_REASONS
  if (!h->reasons_dev) {
    HIPCHK(h, hipMalloc(&h->reasons_dev, 64));
    HIPCHK(h, hipMemset(h->reasons_dev, 0, 64));
  }
  a.reasons = h->reasons_dev;
#endif
#ifdef MBES_STATS
  {
    static unsigned long long* g_stats = nullptr;
    if (!g_stats) {
      hipMalloc(&g_stats, 32);
      hipMemset(g_stats, 0, 32);
    }
    unsigned long long hs[4];
    hipMemcpy(hs, g_stats, 32, hipMemcpyDeviceToHost);
    if (hs[2]) fprintf(stderr, "[mbes stats] rays %llu steps/ray %.2f tests/ray %.2f retries/ray %.4f\n", hs[2], (double)hs[0] / hs[2], (double)hs[1] / hs[2], (double)hs[3] / hs[2]);
    hipMemset(g_stats, 0, 32);
    a.stats = g_stats;
  }
#endif
  if (h->map_kind == 0) {
    a.grid = h->grid;
    a.grid_pad = h->grid_pad;
    a.nyp = h->gny + 2;
    a.nx = h->gnx;
    a.ny = h->gny;
    a.ox = h->gox;
    a.oy = h->goy;
    a.inv_res = 1.0 / h->gres;
    a.res = (float)h->gres;
    a.zmin_map = h->gzmin;
    a.zmax_map = h->gzmax;
  } else {
    const MeshDev* m = h->mesh;
    a.mesh = mesh_args(m);
    a.grid = m->heights;  // non-null: structured mesh (triangulated regular height grid)
    a.grid_pad = m->heights_pad;
    a.nyp = m->gy + 3;
    a.nx = m->gx + 1;
    a.ny = m->gy + 1;
    a.ox = m->x0;
    a.oy = m->y0;
    a.inv_res = 1.0 / m->cs;
    a.res = (float)m->cs;
    a.zmin_map = m->zmin;
    a.zmax_map = m->zmax;
    a.diag_mode = m->diag_mode;
    a.cells = (m->heights && !h->force_general_mesh) ? 0 : 1;
  }
  const long long ngroups = (h->n + MBES_WAVES - 1) / MBES_WAVES;
  const int grid = (int)(ngroups < 4096 ? ngroups : 4096);
  if (!h->mbes_worklist) HIPCHK(h, hipMalloc(&h->mbes_worklist, sizeof(int) * (size_t)(ngroups + 1)));
  if (!h->mbes_groups) HIPCHK(h, hipMalloc(&h->mbes_groups, sizeof(MbesGroup) * (size_t)ngroups));
  a.worklist = h->mbes_worklist;
  a.groups = h->mbes_groups;
  a.work_count = (int*)(h->ctrl + CTRL_WORK);
  // the cast kernels leave max lw in the control block's slots: the normalisation needs no reduction pass
  a.max_slots = (with_ranges && lw_out == h->lw) ? (u64*)(h->ctrl + CTRL_SLOTS) : nullptr;
  // height grids and structured meshes: the pose kernel classifies the groups, k_mbes_fast casts the
  // eligible ones, k_mbes_cast<.,.,1> the worklist; triangle-record meshes keep the two-mode kernel
  const bool lean = true;  // every map kind: the pose kernel classifies the groups
  const bool structured = h->map_kind == 1 && h->mesh->heights && !h->force_general_mesh;
  // ---- fan sweep (mcl_sweep.h): regularly triangulated meshes, ascending beam angles.  The fan plane may lean
  // from the vertical only as far as the steepest triangle allows (tan(tilt) * slope < 1, with a margin).
  if (!pose_done) {
    // (below ~8 k particles even four lanes per side cannot fill the chip and the wave-per-particle traversal is
    //  faster -- measured at 128 ... 262 144 particles x 256 / 512 beams; MCL_SWEEP=1 forces it)
    const bool lattice = h->map_kind == 0 || (structured && (a.diag_mode == 1 || a.diag_mode == 2));
    const long long sweep_min_n = h->env_sweep == 1 ? 1 : (lattice ? 8192 : 16384);   // (adjacency sweep: one lane per side only)
    // a height-field TIN with adjacency -- also a triangulated height grid whose cells are split along mixed diagonals
    // (tin_ok: mesh_build has PROVEN the mesh single-valued over (x, y) -- adjacency, fold and pairwise overlap tests)
    const bool tin = h->map_kind == 1 && h->mesh->tin_ok && !h->mesh_no_sweep && (!structured || a.diag_mode == 0);
    bool sweep = ((structured && (a.diag_mode == 1 || a.diag_mode == 2)) || h->map_kind == 0 || tin) && h->sweep_angles_ok && h->env_sweep != 0 &&
                 h->ng >= sweep_min_n &&  // (the GLOBAL count: every shard of a cloud takes the same path, results do not depend on the GPU count)
                 h->n < (1ll << 31) && (long long)(a.nx + 2) * (a.ny + 2) < (1ll << 30) && a.ny < (1 << 21);   // (the ringed height array as ONE raw buffer: < 2^32 bytes)
    h->sweep_now = sweep;
    // ---- fan slice (mcl_slice.h): every other triangle mesh -- soups, meshes mesh_build could not prove a height
    // field, MCL_MESH_GENERAL -- with an ascending beam table; MCL_SLICE=0 keeps the ray traversal (tests, A/B)
    h->slice_now = !sweep && h->map_kind == 1 && !structured && h->mesh->cell_tri && h->sweep_angles_ok && h->env_slice != 0 &&
                   h->n < (1ll << 31);
    if (sweep) {
      const int nsub_up = sweep_lanes_per_side(h, with_ranges, B);
      RET_IF(upload_sweep_beams(h, with_ranges, B, sigma, r_max, nsub_up));
    } else if (with_ranges && h->ranges_pending) {
      RET_IF(upload(h, h->ranges_dev, h->ranges_host.data(), sizeof(float) * (size_t)B));
      h->ranges_ptr = h->ranges_dev;
      h->ranges_pending = false;
    }
  }
  a.ranges = with_ranges ? h->ranges_ptr : nullptr;
  const bool sweep = h->sweep_now;
  if (sweep) {
    if (!h->defer_idx) HIPCHK(h, hipMalloc(&h->defer_idx, sizeof(u32) * (size_t)h->n));
    a.sweep_beams = h->sweep_beams;
    a.sweep_tail = h->sweep_tail;
    a.b_split = h->b_split;
    a.sweep_nvalid = h->sweep_nvalid;
    a.sweep_tan0 = h->sweep_tail + 2 * (size_t)B;
    a.sweep_tail_run = h->sweep_tail + 2 * (size_t)B + 4;
    // (grids: 0.45 -- below 0.5 the plane function cannot change sign around a cell's four corners, mcl_sweep.h)
    const double slope_max = h->map_kind == 0 ? h->gslope_max : h->mesh->slope_max;
    const double tan_lim = std::min(std::tan(35.0 * MCL_PI / 180.0), (h->map_kind == 0 ? 0.45 : 0.8) / std::max(slope_max, 1e-9));
    a.sweep_c2z_min = (float)(1.0 / std::sqrt(1.0 + tan_lim * tan_lim));
    a.sweep_slope = (float)slope_max;
    a.defer_idx = h->defer_idx;
  }
  const bool slice = h->slice_now;
  if (slice) {
    if (!h->defer_idx) HIPCHK(h, hipMalloc(&h->defer_idx, sizeof(u32) * (size_t)h->n));
    a.defer_idx = h->defer_idx;
    a.slice = 1;
  }
  if (args_only) {
    *args_only = a;
    return MCL_OK;
  }
  // ---- counters of the update two before this one (deterministic lag, see mcl_handle::work_host)
  if (!h->work_host) {
    HIPCHK(h, hipHostMalloc(&h->work_host, 64, hipHostMallocDefault));
    memset(h->work_host, 0, 64);
    for (int k = 0; k < 4; ++k) HIPCHK(h, hipEventCreateWithFlags(&h->ev_upd[k], hipEventDisableTiming));
  }
  static const int wh_zero[4] = {0, 0, 0, 0};
  const int* wh_prev = wh_zero;
  if (h->upd_seq >= 2) {
    HIPCHK(h, hipEventSynchronize(h->ev_upd[(h->upd_seq - 2) & 3]));
    wh_prev = h->work_host + 4 * ((h->upd_seq - 2) & 3);
  }
  int* wh_cur = h->work_host + 4 * (h->upd_seq & 3);  // (its last user, four updates ago, finished before the event above)
  wh_cur[0] = wh_cur[1] = wh_cur[2] = wh_cur[3] = 0;
  struct SeqGuard {  // whatever path returns: this update's kernels are behind its event
    mcl_handle* h;
    ~SeqGuard() {
      (void)hipEventRecord(h->ev_upd[h->upd_seq & 3], h->stream);
      h->upd_seq++;
    }
  } seq_guard{h};
  t_begin(h, MCL_K_UPDATE_MBES);
  if (!pose_done) {
    // (the pose kernel below writes the records in slot order -- or, when the last resample of separate calls prepared a
    //  visiting order for these very slots and a sweep / group slice is about to read them, in that order)
    h->pose_visit = false;
    if (with_ranges && h->visit_ready && (sweep || (slice && h->env_slice_group != 0))) {
      a.visit_okey = h->visit_okey;
      a.visit_base = h->visit_base;
      a.visit_nb = h->visit_nb[0] * h->visit_nb[1] * h->visit_nb[2];
      h->pose_visit = true;
    }
    // (the fused predict has already reset the control block and written poses, group records and worklist)
    if (a.max_slots)
      HIPCHK(h, hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));  // slots + work and hand-over counters (one aligned fill)
    else
      HIPCHK(h, hipMemsetAsync(a.work_count, 0, CTRL_DEFER2 + 4 - CTRL_WORK, h->stream));   // (work, hand-over and loose-group counters; the tickets between them are zero between launches anyway)
    if (lean && !sweep && !slice)
      k_mbes_pose<true><<<grid_for(h->n), 256, 0, h->stream>>>(a);
    else
      k_mbes_pose<false><<<grid_for(h->n), 256, 0, h->stream>>>(a);
  }
  if (a.max_slots) {
    h->max_valid = true;
    h->slot_set = 0;
  }
  a.perm = nullptr;
  if (sweep) {
    const int nsub = sweep_lanes_per_side(h, with_ranges, B);
    a.sweep_nsub = nsub;
    // May the merge loop skip the clamp of the expected range to r_max?  Straight after a predict every particle holds
    // the odometry's depth, roll and pitch; if the map frame is not tilted against the odometry frame the sensor's depth
    // and the vertical component of every beam are then the same for the whole cloud (yaw turns about the vertical),
    // and a beam cannot travel further than down to the map's lowest point: if that is inside r_max for every valid beam
    // of the ping, max(residual, (z - r_max) w) is the residual.
    a.sweep_noclamp = 0;
    {
      static const bool allow = !(getenv("MCL_SWEEP_NOCLAMP") && atoi(getenv("MCL_SWEEP_NOCLAMP")) == 0);
      if (allow && with_ranges && h->uni_valid && a.m2o[8] == 0.0 && a.m2o[9] == 0.0 && (int)h->ranges_host.size() == B) {
        const double roll = h->uni_val[1], pitch = h->uni_val[2], k = a.m2o[10];
        const double zr[3] = {-std::sin(pitch), std::cos(pitch) * std::sin(roll), std::cos(pitch) * std::cos(roll)};
        const double c1z = k * (zr[0] * a.off_R[1] + zr[1] * a.off_R[4] + zr[2] * a.off_R[7]);
        const double c2z = k * (zr[0] * a.off_R[2] + zr[1] * a.off_R[5] + zr[2] * a.off_R[8]);
        const double oz = a.m2o[11] + k * (h->uni_val[0] + zr[0] * a.off_t[0] + zr[1] * a.off_t[1] + zr[2] * a.off_t[2]);
        bool inside = true;
        for (int b = 0; b < B && inside; ++b) {
          if (!(h->ranges_host[b] > 0.f)) continue;
          const double ang = (double)h->beam_cache[b];
          const double dz = std::sin(ang) * c1z - std::cos(ang) * c2z;   // vertical component of the beam's direction
          const double far = dz < -1e-3 ? ((double)a.zmin_map - oz) / dz : -1.0;
          inside = far >= 0.0 && far <= r_max * (1.0 - 1e-3);
        }
        a.sweep_noclamp = inside ? 1 : 0;
      }
    }
    const int sthreads = nsub == 4 ? 512 : SWEEP_THREADS;
    const int per_block = sthreads / 64 / (2 * nsub) * 64;
    // (expected ranges of a few particles: only their lanes are launched)
    const long long n_part = !with_ranges ? std::max<long long>(std::min<long long>(exp_count, h->n - exp_first), 1) : h->n;
    const int sgrid = (int)((n_part + per_block - 1) / per_block);
    const size_t lds = (size_t)(B + 7) * sizeof(float4) + (size_t)(B + 4) * sizeof(float);   // (k_mbes_sweep: 4 spare records, two sentinels, one record in front of either side)
    // What the sweep declines (mcl_sweep.h: tilt, position, no nadir hit, a border the slice may re-cross) is cast by
    // the general kernel, one wavefront per particle, in the order of the hand-over list -- whose length it reads on
    // the device.  TWO launches per update (rounds 2-3: five -- a bounds-checked second sweep pass, classify, fast, cast).
    MbesArgs d = a;
    d.perm = h->defer_idx;
    d.n_dev = a.defer_count;
    d.host_count = wh_cur + 1;  // (pinned: the kernel stores the count there, no copy on the stream)
    // (its loop is grid-stride: the grid only sets the parallelism.  After an update that handed nothing over it is
    //  launched small)
    const bool few = wh_prev[1] == 0;
    const int dgrid = (int)std::min<long long>(ngroups, few ? 64 : 4096);
#define LAUNCH_SWEEP(SURFV, MAPV)                                                      \
  do {                                                                                 \
    if (with_ranges) {                                                                 \
      t_begin(h, MCL_K_MBES_MAIN);                                                     \
      if (nsub > 1)                                                                    \
        k_mbes_sweep<SURFV, false, true><<<sgrid, sthreads, lds, h->stream>>>(a);      \
      else                                                                             \
        k_mbes_sweep<SURFV, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);       \
      t_end(h);                                                                        \
      k_mbes_cast<MAPV, false, 2><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);           \
    } else {                                                                           \
      k_mbes_sweep<SURFV, true><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);          \
      k_mbes_cast<MAPV, true, 2><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);            \
    }                                                                                  \
  } while (0)
    // A TIN with HOLES (data gaps, a ragged outline): a slice that runs into one ends the walk and its particle is handed
    // over -- with a converged cloud every particle of the ping at once, for as long as the gap lies under the swath.  The
    // ray traversal casts a million such particles in 21 ms (48 x the sweep); the fan slice (mcl_slice.h: any soup, exact
    // across holes) in ~3: on meshes that HAVE holes the hand-over list goes through the slice first -- groups of
    // SLICE_G consecutive entries (a wave of the sweep appends its particles together: spatial neighbours) share one
    // staged triangle list, k_mbes_slice takes the groups that are not tight -- and only what the slice declines too
    // (fans further than 60 degrees from the vertical) reaches the general kernel, through a second list.  Which
    // kernel casts a particle is still decided by the particle (and the map) alone: the determinism rule holds.
    const bool slice_handover = with_ranges && nsub == 1 && h->map_kind == 1 && !structured && h->mesh->tin_holes && h->mesh->cell_tri &&
                                h->env_slice != 0 && h->env_handover_slice != 0;
    if (with_ranges) h->handover_slice_now = slice_handover;
    if (slice_handover) {
      if (!h->defer2_idx) HIPCHK(h, hipMalloc(&h->defer2_idx, sizeof(u32) * (size_t)h->n));
      const long long ngr = (h->n + SLICE_G - 1) / SLICE_G;
      if (!h->slice_loose) HIPCHK(h, hipMalloc(&h->slice_loose, sizeof(u32) * (size_t)ngr));
      t_begin(h, MCL_K_MBES_MAIN);
      if (h->mesh->tin_rims)
        k_mbes_sweep<6, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);
      else
        k_mbes_sweep<5, false><<<sgrid, SWEEP_THREADS, lds, h->stream>>>(a);
      t_end(h);
      MbesArgs s2 = a;
      s2.slice = 1;
      s2.in_list = h->defer_idx;
      s2.in_count = a.defer_count;
      s2.defer_idx = h->defer2_idx;
      s2.defer_count = (int*)(h->ctrl + CTRL_DEFER2);
      s2.slice_loose = nullptr;
      s2.host_count = wh_cur + 2;   // (pinned: how many particles the SWEEP handed over -- the cast below reports what the slice handed on)
      // (both kernels stride over their lists and every workgroup of them builds the ping's beam tables first -- 0.5 ms for
      //  4 096 workgroups that then find a handful of particles: the grids follow the count of two updates ago, doubled,
      //  and never fall below 512 workgroups -- those without work leave at once, mcl_slice.h -- so that the first two
      //  updates of a cloud that runs into a ragged outline all at once are not cast by 64)
      const long long h_prev = 2ll * std::max(wh_prev[2], 0);
      const size_t lds_s = ((size_t)B * (2 + SLICE_WAVES) + (size_t)SLICE_WAVES * (SLICE_LIST + 1)) * sizeof(float) + SLICE_LUT * sizeof(unsigned short);
      const size_t lds_g = (size_t)B * (3 + SLICE_G_WAVES) * sizeof(float) + (size_t)SLICE_G_TRIS * 9 * sizeof(float) +
                           SLICE_G_HASH * sizeof(unsigned) + SLICE_LUT * sizeof(unsigned short);
      // (the group kernel pays off on long lists; group and per-particle slice give the same bits -- tests/test_gpu_slice.py --,
      //  so a short list, by the count of two updates ago, goes to the per-particle kernel alone: one launch less on every
      //  update that hands nothing over)
      const bool fits = lds_g + 4096 <= (size_t)160 * 1024 && h->env_slice_group != 0 && h_prev >= 8192;
      const unsigned hgrid = (unsigned)std::min<long long>(std::max<long long>((h_prev + SLICE_G - 1) / SLICE_G, 512), 4096);
      const unsigned pgrid = (unsigned)std::min<long long>(std::max<long long>((h_prev + SLICE_WAVES - 1) / SLICE_WAVES, 512), 2048);
      if (fits) {
        if (!h->slice_attr_set || lds_g > h->slice_attr_bytes) {
          HIPCHK(h, hipFuncSetAttribute((const void*)k_mbes_slice_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));
          h->slice_attr_set = true;
          h->slice_attr_bytes = lds_g;
        }
        s2.slice_loose = h->slice_loose;
        s2.slice_loose_count = (int*)(h->ctrl + CTRL_LOOSE);
        k_mbes_slice_group<<<(unsigned)std::min<long long>(ngr, hgrid), SLICE_G_THREADS, lds_g, h->stream>>>(s2);
      }
      k_mbes_slice<false><<<pgrid, SLICE_THREADS, lds_s, h->stream>>>(s2);
      d.perm = h->defer2_idx;
      d.n_dev = s2.defer_count;
      k_mbes_cast<1, false, 2><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);
    } else if (h->map_kind == 0)
      LAUNCH_SWEEP(0, 0);
    else if (!structured && h->mesh->tin_rims)
      LAUNCH_SWEEP(6, 1);   // TIN by adjacency, holes crossed by their rim records (expected ranges, runs of beams: the staged hand-over above took the rest)
    else if (!structured)
      LAUNCH_SWEEP(5, 1);   // TIN by adjacency; hand-overs: triangle records
    else if (a.diag_mode == 0)
      LAUNCH_SWEEP(5, 2);   // mixed diagonals: the adjacency walk; hand-overs: node heights with the per-cell diagonal bit
    else if (a.diag_mode == 1)
      LAUNCH_SWEEP(2, 2);
    else
      LAUNCH_SWEEP(3, 2);
#undef LAUNCH_SWEEP
    if (h->env_debug_work) {
      int cnt = 0;
      (void)hipMemcpyAsync(&cnt, a.defer_count, sizeof(int), hipMemcpyDeviceToHost, h->stream);
      (void)hipStreamSynchronize(h->stream);
      fprintf(stderr, "[mbes] sweep handed over %d of %lld particles (clamp to r_max %s)\n", cnt, (long long)h->n,
              a.sweep_noclamp ? "proved idle: skipped" : "kept");
#ifdef SWEEP_REASONS
      // why (mcl_sweep.h: SWEEP_FAIL / SWEEP_NOTE codes, per particle SIDE): -DSWEEP_REASONS builds only
      unsigned why[16];
      (void)hipMemcpy(why, h->reasons_dev, sizeof why, hipMemcpyDeviceToHost);
      (void)hipMemset(h->reasons_dev, 0, sizeof why);
      fprintf(stderr, "[mbes] declined sides by reason:");
      for (int k = 0; k < 16; ++k)
        if (why[k]) fprintf(stderr, " %d:%u", k, why[k]);
      fprintf(stderr, "\n");
#endif
    }
    t_end(h);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
  }
  if (slice) {
    // one wavefront per particle; what it declines (fans far from vertical) goes to the general kernel over the
    // triangle records, through the same hand-over list as the sweep's
    MbesArgs d = a;
    d.perm = h->defer_idx;
    d.n_dev = a.defer_count;
    d.host_count = wh_cur + 1;
    const long long n_part = !with_ranges ? std::max<long long>(std::min<long long>(exp_count, h->n - exp_first), 1) : h->n;
    // a capped grid: the beam tables and the tangent buckets a workgroup builds in LDS are shared by 8 particles per wave
    // at 1 M instead of ONE (mcl_slice.h: SLICE_GRID, measured)
    const int sgrid = (int)std::min<long long>((n_part + SLICE_WAVES - 1) / SLICE_WAVES, SLICE_GRID);
    const size_t lds = ((size_t)B * (2 + SLICE_WAVES) + (size_t)SLICE_WAVES * (SLICE_LIST + 1)) * sizeof(float) + SLICE_LUT * sizeof(unsigned short);
    const int dgrid = (int)std::min<long long>(ngroups, wh_prev[1] == 0 ? 64 : 4096);
    if (with_ranges) {
      t_begin(h, MCL_K_MBES_MAIN);
      // records in visiting order: groups of SLICE_G consecutive ones are spatial neighbours and share ONE candidate
      // triangle list, staged in LDS (k_mbes_slice_group); what it leaves -- groups that are not tight, that overflow
      // the staging area -- k_mbes_slice casts, by the list the group kernel leaves
      a.slice_loose = nullptr;
      h->slice_group_ran = false;
      if (pose_done && h->pose_visit && h->env_slice_group != 0) {
        const long long ngr = (h->n + SLICE_G - 1) / SLICE_G;
        if (!h->slice_loose) HIPCHK(h, hipMalloc(&h->slice_loose, sizeof(u32) * (size_t)ngr));
        a.slice_loose = h->slice_loose;
        a.slice_loose_count = (int*)(h->ctrl + CTRL_LOOSE);   // (zeroed with the control block by this step's predict)
        const size_t lds_g = (size_t)B * (3 + SLICE_G_WAVES) * sizeof(float) + (size_t)SLICE_G_TRIS * 9 * sizeof(float) +
                             SLICE_G_HASH * sizeof(unsigned) + SLICE_LUT * sizeof(unsigned short);
        // does the group kernel's staging layout fit this ping's beam table?  Decided BEFORE anything is asked of the
        // runtime (hipFuncSetAttribute itself fails beyond the device's limit, and the kernel's static
        // __shared__ -- member poses, reference plane, counters: a conservative 4 KiB -- counts against the same 160 KiB)
        const bool fits = lds_g + 4096 <= (size_t)160 * 1024;
        if (fits && (!h->slice_attr_set || lds_g > h->slice_attr_bytes)) {   // (more than 64 KiB of dynamic LDS has to be asked for)
          HIPCHK(h, hipFuncSetAttribute((const void*)k_mbes_slice_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));
          h->slice_attr_set = true;
          h->slice_attr_bytes = lds_g;
        }
        if (fits) {
          k_mbes_slice_group<<<(unsigned)std::min<long long>(ngr, 4096), SLICE_G_THREADS, lds_g, h->stream>>>(a);
          h->slice_group_ran = true;
        } else
          a.slice_loose = nullptr;   // (a beam table too long for the staging layout: the per-particle kernel casts everything)
      }
      // (behind the group kernel it casts the few groups on its list: a small grid -- every workgroup builds the beam tables)
      k_mbes_slice<false><<<a.slice_loose ? std::min(sgrid, 2048) : sgrid, SLICE_THREADS, lds, h->stream>>>(a);
      t_end(h);
      k_mbes_cast<1, false, 2><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);
    } else {
      k_mbes_slice<true><<<sgrid, SLICE_THREADS, lds, h->stream>>>(a);
      k_mbes_cast<1, true, 2><<<dgrid, MBES_THREADS, 0, h->stream>>>(d);
    }
    t_end(h);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
  }
  if (lean) {
    // Dispersed cloud?  The natural-order classification has just counted the groups without a common tile.
    // That count travels to the host asynchronously and is read one call late (no synchronisation): when the
    // previous update deferred more than 1/16 of its groups, this one visits the particles in Morton order.
    const bool sort_now = h->env_sort >= 0 ? h->env_sort == 1 : (long long)wh_prev[0] * 16 > ngroups;
    HIPCHK(h, hipMemcpyAsync(wh_cur, a.work_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (sort_now && h->n > MBES_WAVES) {
      RET_IF(sort_visiting_order(h, a));
      a.perm = h->mbes_perm;
      HIPCHK(h, hipMemsetAsync(a.work_count, 0, sizeof(int), h->stream));
      k_mbes_classify<<<grid_for(h->n), 256, 0, h->stream>>>(a);
    }
  }
  const int ggrid = (int)(ngroups < 512 ? ngroups : 512);
#define LAUNCH_LEAN(SURFV, MAPV)                                                   \
  do {                                                                             \
    if (with_ranges) {                                                             \
      t_begin(h, MCL_K_MBES_MAIN);                                                 \
      k_mbes_fast<SURFV, false><<<grid, MBES_THREADS, 0, h->stream>>>(a);          \
      t_end(h);                                                                    \
      k_mbes_cast<MAPV, false, 1><<<ggrid, MBES_THREADS, 0, h->stream>>>(a);       \
    } else {                                                                       \
      k_mbes_fast<SURFV, true><<<grid, MBES_THREADS, 0, h->stream>>>(a);           \
      k_mbes_cast<MAPV, true, 1><<<ggrid, MBES_THREADS, 0, h->stream>>>(a);        \
    }                                                                              \
  } while (0)
  if (h->map_kind == 0) {
    LAUNCH_LEAN(0, 0);
  } else if (structured) {
    if (a.diag_mode == 1)
      LAUNCH_LEAN(2, 2);
    else if (a.diag_mode == 2)
      LAUNCH_LEAN(3, 2);
    else
      LAUNCH_LEAN(1, 2);
  } else {
    LAUNCH_LEAN(4, 1);  // triangle records: cell-word tiles
  }
#undef LAUNCH_LEAN
  if (h->env_debug_work) {  // diagnostics: how many groups the fast kernel deferred
    int cnt = 0;
    (void)hipMemcpyAsync(&cnt, a.work_count, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    (void)hipStreamSynchronize(h->stream);
    fprintf(stderr, "[mbes] deferred %d of %lld groups\n", cnt, ngroups);
    if (lean && cnt > 0) {
      std::vector<MbesGroup> g((size_t)ngroups);
      (void)hipMemcpy(g.data(), h->mbes_groups, sizeof(MbesGroup) * (size_t)ngroups, hipMemcpyDeviceToHost);
      long long why[32] = {0}, area = 0, na = 0;
      for (const MbesGroup& G : g)
        if (!G.fast) {
          why[G.why & 31]++;
          area += (long long)G.tw * G.th;
          ++na;
        }
      fprintf(stderr, "[mbes] why:");
      for (int k = 0; k < 32; ++k)
        if (why[k]) fprintf(stderr, " %d:%lld", k, why[k]);
      fprintf(stderr, "  mean window of deferred groups %lld nodes\n", na ? area / na : 0);
    }
  }
  t_end(h);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// a fused step that leaves before its gather: store the z, roll, pitch its predict kernel did not
int materialise_uniform(mcl_handle* h) {
  if (!h->uni_deferred) return MCL_OK;
  h->uni_deferred = false;
  RET_IF(set_device(h));
  k_fill_uniform<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->uni_val[0],
                                                             h->uni_val[1], h->uni_val[2]);
  HIPCHK(h, hipGetLastError());
  return MCL_OK;
}

// pose_for: when given (fused step, NATIVE rng) the kernel also writes the MBES pose records of the new
// state; *pose_written tells the caller whether it did (a dt <= 0 step leaves the state untouched)
int do_predict(mcl_handle* h, const mcl_odom* od, double dt, const double* replay_normals,
               const MbesArgs* pose_for = nullptr, bool* pose_written = nullptr, bool defer_uniform = false) {
  if (pose_written) *pose_written = false;
  if (!(dt > 0.0)) return MCL_OK;  // auv_pf.py:205 gate
  double rpy[3];
  euler_from_quat(od->q, rpy);
  const double roll = rpy[0], pitch = rpy[1];
  const double cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll), sr = std::sin(roll);
  // M1 = Ry' Rx with the reference's Ry' (auv_particle.py:90-92); rows 0,1 only
  const double M1r0[3] = {cp, sp * sr, sp * cr};
  const double M1r1[3] = {0.0, cr, -sr};
  const double vdt[3] = {od->v[0] * dt, od->v[1] * dt, od->v[2] * dt};
  PredictArgs a;
  a.m0 = M1r0[0] * vdt[0] + M1r0[1] * vdt[1] + M1r0[2] * vdt[2];
  a.m1 = M1r1[0] * vdt[0] + M1r1[1] * vdt[1] + M1r1[2] * vdt[2];
  a.wzdt = od->w_z * dt;
  a.z = od->z;
  a.roll = roll;
  a.pitch = pitch;
  a.nz = noise_args(h, h->cfg.process_cov, 1u, h->step_predict);
  a.zero_ptr = nullptr;
  a.zero_words = 0;
  a.skip_uniform = 0;
  a.visit_okey = a.visit_base = nullptr;
  a.visit_nb = 0;
  h->pose_visit = false;
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (replay_normals) {
      RET_IF(upload_replay(h, replay_normals));
      rp = h->replay_dev;
    } else {
      for (int c = 0; c < 6; ++c) a.nz.sq[c] = 0.0;  // REPLAY without draws: noise-free
      rp = nullptr;
    }
  }
  t_begin(h, MCL_K_PREDICT);
  if (h->cfg.rng_mode == MCL_RNG_REPLAY && !rp) {
    // noise-free: feed zeros through the native branch with sq = 0
  }
  if (pose_for && !rp && h->cfg.rng_mode == MCL_RNG_NATIVE) {
    const bool lean = !pose_for->sweep_beams && !pose_for->slice;  // the fan sweep and the fan slice need no group records
    a.skip_uniform = defer_uniform ? 1 : 0;
    h->uni_deferred = defer_uniform;
    if (lean) {
      // reset the slots + work counter first: the kernel appends the deferred groups to the worklist
      // (the whole block: ONE aligned fill; the kernel tickets in it are zero between launches anyway)
      HIPCHK(h, hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));
    } else {
      a.zero_ptr = (unsigned long long*)h->ctrl;   // the kernel's first workgroup zeroes it: no memset launch
      a.zero_words = CTRL_BYTES / 8;
      if ((pose_for->sweep_beams || (pose_for->slice && h->env_slice_group != 0)) && h->visit_ready) {
        // the fan sweep visits the particles in the spatial order the last gather prepared: the records go to their
        // sorted positions (the sweep writes log-likelihoods by the slot in the record)
        a.visit_okey = h->visit_okey;
        a.visit_base = h->visit_base;
        a.visit_nb = h->visit_nb[0] * h->visit_nb[1] * h->visit_nb[2];
        h->pose_visit = true;
      }
    }
    if (lean)
      k_predict_pose<true><<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, *pose_for);
    else
      k_predict_pose<false><<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, *pose_for);
    if (pose_written) *pose_written = true;
  } else {
    k_predict<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, rp);
  }
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->step_predict++;
  // every particle now holds the odometry's depth, roll and pitch (both kernels store these three constants)
  h->uni_valid = true;
  h->uni_val[0] = a.z;
  h->uni_val[1] = a.roll;
  h->uni_val[2] = a.pitch;
  return MCL_OK;
}

}  // namespace
