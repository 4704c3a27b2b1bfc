// mcl_resample.h -- the systematic resample as three streaming kernels (gfx950, wave64):
//
//   k_quantise_tiles   lw -> fixed-point weights q + per-tile sums; the block that finishes LAST scans the
//                      tile sums (exclusive offsets + total) -- no separate scan launch.
//   k_cdf_expand       q -> offspring CDF ncum (exact integer arithmetic); zero-offspring
//                      flags and their global ranks by a single-pass DECOUPLED LOOK-BACK scan over the
//                      tiles; every lost slot learns its rank, every ancestor with surplus copies writes
//                      its index into the dupes list at the ranks it owns.  keep/lost/dupes of
//                      auv_pf.py:183-190 without a search: dupes[k] is written, not looked up.
//   k_resample_gather  dst[i] = src[i] (survivor) or src[dupes[rank_i]] (lost slot) + add_noise
//                      (auv_pf.py:191-198), and -- for the fused step -- the 13 sums of
//                      update_loc_pose (auv_pf.py:218-252) of the NEW state in the same pass, taken about
//                      a shift (the pre-resample state of global particle 0) so that the centred moments
//                      keep full precision; the last block adds the per-block partials in block order.
//
// Inter-block hand-offs follow MI355X_MICROARCH.md ("inter-workgroup visibility"): a tile descriptor is ONE
// naturally aligned 8-byte word {epoch:30 | status:2 | value:32} written by ONE relaxed agent-scope store
// and read by relaxed agent-scope loads (sc1: served by L2, never a stale L1 line), so it needs no fence;
// the "last block" of the gather publishes its 13 partial sums with one write-through store, drains it
// (s_waitcnt vmcnt(0)) and draws a ticket; the block that draws the last one acquires and adds them up.
// Tiles are handed out by a ticket counter, so a block only ever waits for tiles that are already running;
// blocks are 1024 threads so that 1 M particles need only 256 tickets (returning atomics on one word
// serialise in L2 at ~88 per microsecond).
#pragma once
#include "mcl_kernels.h"

// ------------------------------------------------------------------ max log-weight into slots
__global__ void __launch_bounds__(MCL_BLOCK) k_max_slots(const double* __restrict__ v, long long n,
                                                         u64* __restrict__ slots) {
  __shared__ double sh[16];
  double m = -__builtin_inf();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double x = v[i];
    m = (x > m) ? x : m;  // NaN never wins
  }
  m = block_max(m, sh, -__builtin_inf());
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&slots[blockIdx.x & (MCL_MAX_SLOTS - 1)], ordered_key(m));
}
__global__ void __launch_bounds__(64) k_max_finish(const u64* __restrict__ slots, double* __restrict__ out) {
  const double m = max_from_slots(slots);
  if (threadIdx.x == 0) out[0] = m;
}

// ------------------------------------------------------------------ K1: quantise + tile sums
// One returning atomic on ONE word costs ~11 ns (the L2 serialises them: ~88 per microsecond), so a
// "last block done" ticket per 256-thread block would cost more than this whole kernel at 1 M particles.
// K1 therefore only writes one sum per tile of MCL_SCAN_TILE particles; whoever needs a prefix of them
// (k_cdf_expand, one 1024-thread block per 4 tiles) adds the few hundred sums up itself.
struct QuantArgs {
  const double* lw;
  long long n;
  const u64* slots;    // max-log-weight slots filled by the update kernel, or nullptr: read m_lw[0]
  const double* m_lw;  // the (all-reduced) maximum
  int mode;
  double scale;
  int s;               // log2 of scale
  u64* q;
  u64* tile_sum;   // gridDim.x entries
  // one shard of a cloud spread over several processes: the maximum in `slots` is the SHARD's, the weights
  // are taken at the shard's own exponent, and the kernel leaves what the other shards need to put them on the
  // cloud's exponent WITHOUT another pass: the number of weights with bit b set, b = 0 .. 63 -- the shard's total at
  // any right shift d is sum_{b >= d} count_b 2^(b - d) -- per tile (tile_bits, 64 u16 each: the tile offsets at the
  // shift that turns out to apply) and for the shard (rec[1 + b], accumulated by atomics: zero before the launch,
  // zeroed again by k_shift_scan); rec[0] = the shard's exponent (biased).  nullptr: none of this.
  u64* rec;
  unsigned short* tile_bits;
};
#define SHREC_COPIES 8                       // the bit counts are accumulated in 8 copies (workgroup & 7): 512 workgroups' atomics on 64 words of four cache lines cost 10 us, on 512 words one
#define SHREC_MOM (1 + 64 * SHREC_COPIES)    // ... then the 13 sums of update_loc_pose (+ 3 shift words) the shard's LAST fused step left: they ride with this step's record
#define SHREC_WORDS (SHREC_MOM + 16)         // a shard's record: exponent, 8 x 64 bit counts (to be added up), 16 moment words
#define SHREC_BIAS (1ll << 41)
__global__ void __launch_bounds__(MCL_BLOCK) k_quantise_tiles(QuantArgs a) {
  __shared__ u64 sh[16];
  const long long tile = blockIdx.x;  // the grid is exactly the number of tiles
  const long long base = tile * MCL_SCAN_TILE;
  // (the log-weights first: their loads are in flight while the maximum is read from the 64 slots -- two memory
  //  latencies in a 9 us kernel were one after the other)
  double lwv[MCL_SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
    const long long i = base + (long long)k * MCL_BLOCK + threadIdx.x;
    lwv[k] = i < a.n ? a.lw[i] : 0.0;
  }
  const double m = a.slots ? max_from_slots(a.slots) : a.m_lw[0];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < MCL_SCAN_ITEMS; ++k) {
    const long long i = base + (long long)k * MCL_BLOCK + threadIdx.x;
    if (i < a.n) {
      const u64 qi = quantise_weight(lwv[k], m, a.mode, a.scale, a.s);
      a.q[i] = qi;
      acc += qi;
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) a.tile_sum[tile] = acc;
}

// The same for ONE SHARD of a cloud spread over several processes (QuantArgs::rec): weights at the shard's own exponent
// and the bit counts.  One weight per thread, 16 waves per tile: the counting is 63 ballots per wave (walk the word from
// its top bit down: a shift and a sign test per bit), lane b keeps bit b's count -- a few hundred instructions per wave,
// which need the chip's waves side by side, not four tiles' worth in a row per wave (the first version, inside
// k_quantise_tiles with four weights per thread: + 13 us at 524 288 particles, latency of two waves per SIMD).
__global__ void __launch_bounds__(MCL_SCAN_TILE) k_quantise_shard(QuantArgs a) {
  __shared__ u32 bc[64];
  const long long tile = blockIdx.x;
  const long long i = tile * MCL_SCAN_TILE + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const double lw = i < a.n ? a.lw[i] : 0.0;
  if (threadIdx.x < 64) bc[threadIdx.x] = 0u;
  const double m = max_from_slots(a.slots);
  u64 q = 0ull;
  if (i < a.n) {
    q = quantise_weight(lw, m, a.mode, a.scale, a.s);
    a.q[i] = q;
  }
  u64 any = q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, MCL_WAVE);
  const int top = any ? 63 - __builtin_clzll(any) : -1;   // (wave-uniform)
  u32 mine = 0u;
  if (top >= 0) {
    u64 x = q << (63 - top);   // bit `top` in the top position
    for (int b = top; b >= 0; --b) {
      const u32 c = (u32)__popcll(__ballot((long long)x < 0));
      mine = lane == b ? c : mine;
      x <<= 1;
    }
  }
  __syncthreads();   // (bc zeroed)
  if (mine) atomicAdd(&bc[lane], mine);
  __syncthreads();
  if (threadIdx.x < 64) {
    const u32 c = bc[threadIdx.x];
    a.tile_bits[tile * 64 + threadIdx.x] = (unsigned short)c;   // (<= 1 024 weights per tile)
    if (c) atomicAdd((unsigned long long*)&a.rec[1 + 64 * (tile & (SHREC_COPIES - 1)) + threadIdx.x], (unsigned long long)c);
  }
  if (tile == 0 && threadIdx.x == 0) a.rec[0] = (u64)(weight_exponent(m) + SHREC_BIAS);
}

// After the all-gather of the shards' records (ONE collective where rounds 1-5 had an all-reduce of the maximum and,
// dependent on it, an all-gather of the totals): the cloud's exponent K = max K_r, every shard's shift d_r = K - K_r and
// total at that shift, this shard's shift (-> shift_out, read by k_cdf_expand / k_offspring_cdf: q >> d on the fly) and
// its exclusive tile offsets at that shift from the per-tile bit counts; its record's accumulators are zeroed for the
// next resample.  One block.
struct ShiftArgs {
  u64* recs;           // world x SHREC_WORDS (all-gathered)
  int rank, world;
  const unsigned short* tile_bits;
  long long ntiles;
  u64* tile_off;       // out: exclusive offsets of this shard's tiles at the cloud's exponent
  u64* totals;         // out: world totals at the cloud's exponent
  u64* shift_out;      // out: this shard's shift (0 .. 64)
  double* mom_out;     // the PREVIOUS fused step's moments ride in the records (words SHREC_MOM ...): their sum over the shards, in rank order, goes here (16 doubles of the pinned result ring); nullptr: none pending
};
__global__ void __launch_bounds__(1024) k_shift_scan(ShiftArgs a) {
  __shared__ u64 sh[16];
  __shared__ u64 carry_sh;
  __shared__ u32 d_sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u64 K = 0ull;
  for (int r = 0; r < a.world; ++r) {
    const u64 k = a.recs[(size_t)r * SHREC_WORDS];
    K = k > K ? k : K;
  }
  // a wave per shard: lane b holds count_b 2^(b - d), the shard's total at its shift is their sum
  for (int r = w; r < a.world; r += 16) {
    const u64 dd = K - a.recs[(size_t)r * SHREC_WORDS];
    const u32 d = dd > 64ull ? 64u : (u32)dd;
    u64 c = 0ull;
#pragma unroll
    for (int k = 0; k < SHREC_COPIES; ++k) c += a.recs[(size_t)r * SHREC_WORDS + 1 + 64 * k + lane];
    const u64 t = wave_sum((u32)lane >= d ? c << ((u32)lane - d) : 0ull);
    if (lane == 0) {
      a.totals[r] = t;
      if (r == a.rank) {
        d_sh = d;
        a.shift_out[0] = (u64)d;
      }
    }
  }
  if (a.mom_out && threadIdx.x >= 1024 - 16) {
    // (the last 16 threads: 13 sums added in rank order -- every rank gets the same bits --, the shift words are the
    //  same on every rank: this shard's own)
    const int c = threadIdx.x - (1024 - 16);
    double v = 0.0;
    if (c < 13)
      for (int r = 0; r < a.world; ++r) v += __longlong_as_double((long long)a.recs[(size_t)r * SHREC_WORDS + SHREC_MOM + c]);
    else
      v = __longlong_as_double((long long)a.recs[(size_t)a.rank * SHREC_WORDS + SHREC_MOM + c]);
    a.mom_out[c] = v;
  }
  if (threadIdx.x == 0) carry_sh = 0ull;
  __syncthreads();
  const u32 d = d_sh;
  // two threads per tile, 32 bit counts each (four 16-byte loads in flight at once), 512 tiles per round
  const int half = threadIdx.x & 1;
  for (long long base = 0; base < a.ntiles; base += blockDim.x / 2) {
    const long long i = base + (threadIdx.x >> 1);
    u64 v = 0ull;
    if (i < a.ntiles) {
      const uint4* tb = (const uint4*)(a.tile_bits + i * 64 + half * 32);
      uint4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = tb[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 wds[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 b0 = (u32)(half * 32 + j * 8 + k * 2);
          const u64 c0 = wds[k] & 0xffffu, c1 = wds[k] >> 16;
          v += b0 >= d ? c0 << (b0 - d) : 0ull;
          v += b0 + 1u >= d ? c1 << (b0 + 1u - d) : 0ull;
        }
      }
    }
    v += __shfl_xor(v, 1, MCL_WAVE);   // the tile's sum in both of its threads
    const u64 mine = half ? 0ull : v;  // ... counted once
    const u64 incl = wave_scan_incl(mine);
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    u64 woff = 0ull;
    for (int k = 0; k < w; ++k) woff += sh[k];
    const u64 carry = carry_sh;
    if (i < a.ntiles && !half) a.tile_off[i] = carry + woff + incl - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_sh = carry + woff + incl;
    __syncthreads();
  }
  // (every read of this shard's own record is behind the barriers above)
  if (threadIdx.x < 64 * SHREC_COPIES) a.recs[(size_t)a.rank * SHREC_WORDS + 1 + threadIdx.x] = 0ull;
}

// ------------------------------------------------------------------ decoupled look-back (u32 sums)
#define DESC_AGG 1ull
#define DESC_PREFIX 2ull
__device__ __forceinline__ u32 wave_sum_all(u32 v) { return wave_sum_dpp(v); }
// Called by ONE full wave.  Publishes this tile's aggregate, returns the sum of all earlier tiles.
__device__ __forceinline__ u32 scan_lookback(u64* desc, long long tile, u32 agg, u32 epoch) {
  const int lane = threadIdx.x & 63;
  const u64 ep = (u64)(epoch & 0x3fffffffu);
  const u64 tag = ep << 34;
  if (tile == 0) {
    if (lane == 0) store_agent(&desc[0], tag | (DESC_PREFIX << 32) | (u64)agg);
    return 0u;
  }
  if (lane == 0) store_agent(&desc[tile], tag | (DESC_AGG << 32) | (u64)agg);
  u32 excl = 0u;
  long long p = tile - 1;  // nearest predecessor not yet accounted for; lane l looks at p - l
  for (;;) {
    const long long idx = p - lane;
    const u64 d = idx >= 0 ? load_agent(&desc[idx]) : (tag | (DESC_PREFIX << 32));  // before tile 0: prefix 0
    const u32 st = (u32)(d >> 32) & 3u;
    const bool valid = (d >> 34) == ep && st != 0u;
    const unsigned long long vmask = __ballot(valid), pmask = __ballot(valid && st == (u32)DESC_PREFIX);
    const int first_invalid = (~vmask) ? __ffsll((long long)~vmask) - 1 : 64;
    const int first_pref = pmask ? __ffsll((long long)pmask) - 1 : 64;
    if (first_pref < first_invalid) {  // an inclusive prefix within the valid run: done
      excl += wave_sum_all(lane <= first_pref ? (u32)d : 0u);
      break;
    }
    if (first_invalid == 0) {  // the nearest predecessor has not published yet
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    excl += wave_sum_all(lane < first_invalid ? (u32)d : 0u);
    p -= first_invalid;
  }
  if (lane == 0) store_agent(&desc[tile], tag | (DESC_PREFIX << 32) | (u64)(excl + agg));
  return excl;
}

// ------------------------------------------------------------------ K2: CDF + lost ranks + dupes list
#define RS_BLOCK 1024                       // 16 waves: one block per CU, <= 256 tickets at 1 M particles
#define RS_ITEMS MCL_SCAN_ITEMS
#define RS_TILE (RS_BLOCK * RS_ITEMS)
#define RS_FINE (RS_TILE / MCL_SCAN_TILE)   // K1 tiles per K2 tile
#define EXP_HEAVY 24   // an ancestor with more surplus copies than this is expanded by the whole block
#define EXP_LIST 192
#define ZR_SURVIVOR 0xffffffffu
struct ExpandArgs {
  const u64* q;          // FROM_Q: fixed-point weights of this (single) shard
  const u64* tile_sum;   // FROM_Q: the sums k_quantise_tiles wrote, n_fine entries
  long long n_fine;
  u64 n_global_u, u53;   // FROM_Q
  u64* total_out;        // FROM_Q: block 0 leaves the total weight here
  u32* ncum;             // FROM_Q: out; else in: the all-gathered global offspring CDF
  long long n;           // elements scanned (always the GLOBAL particle count)
  long long own0, own_n; // global index range of the slots this shard owns
  u32* zr;               // own_n: rank among the lost slots, or ZR_SURVIVOR
  u32* dupes;            // n: dupes[k] = ancestor whose copy the k-th lost slot receives
  u64* desc;             // one descriptor per tile
  u32* ticket;           // zero before the launch, left zero
  u32 epoch;             // differs from launch to launch: stale descriptors are never valid
  // FROM_Q over ONE SHARD of a sharded cloud (the O(n)-per-rank exchange): `totals` = the all-gathered
  // shard totals (the weight before this shard and the global total come from them), `tile_sum` then holds the
  // EXCLUSIVE tile offsets k_scan_tile_sums left; ranks (zr) and the dupes list are LOCAL to the shard -- positions
  // in the shard's own list of lost slots / surplus copies -- and the block of the last tile leaves in `ls_out`
  // {L = lost slots, S = surplus copies} of the shard and the position of its particle 0 (the moments' shift)
  const u64* totals;
  int rank, world;
  u64* ls_out;           // 4 words: L | S << 32, then x, y, z of local particle 0 as doubles
  const double* p0[3];
  const u64* shift;      // FROM_Q over one shard of several processes: q is at the shard's own exponent, the cloud's weights are q >> shift[0] (k_shift_scan); nullptr: q as it is
  int p0_z_uniform;      // the fused step has not stored z yet (k_predict_pose, skip_uniform): it is the odometry's ...
  double p0_z;           // ... depth on every particle -- the shift's z component is this value, not a stale state word
};
template <bool FROM_Q>
__global__ void __launch_bounds__(RS_BLOCK) k_cdf_expand(ExpandArgs a) {
  __shared__ u64 sh64[16];
  __shared__ u64 shb[16];
  __shared__ u32 sh32[16];
  __shared__ u32 snc[RS_TILE];
  __shared__ u32 heavy[EXP_LIST][3];
  __shared__ u32 tile_sh, zex_sh, agg_sh, heavy_n;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    MCL_TICKET_FENCE();
    const u32 t = atomicAdd(a.ticket, 1u);
    MCL_TICKET_FENCE();
    tile_sh = t;
    heavy_n = 0u;
    if (t == gridDim.x - 1u) atomicExch(a.ticket, 0u);  // every ticket of this launch has been drawn
  }
  __syncthreads();
  const long long tile = tile_sh;
  const long long base = tile * RS_TILE + (long long)tid * RS_ITEMS;
  u32 nc[RS_ITEMS];
  u32 prev_tile = 0u;  // offspring CDF just before this tile
  u32 nc_start = 0u;   // ... and just before the scanned range (non-zero only for a later shard of a sharded cloud)
  if (FROM_Q) {
    const long long fine0 = tile * RS_FINE;
    // (this tile's weights first: in flight while every block adds the tile sums up)
    u64 v[RS_ITEMS];
    const u32 qshift = a.shift ? (u32)a.shift[0] : 0u;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) v[k] = (base + k < a.n) ? shift_weight(a.q[base + k], qshift) : 0ull;
    u64 off = 0ull, T = 0ull;
    if (a.totals) {
      // one shard of several: the weight before the shard and the global total from the all-gathered shard totals
      u64 shard_off = 0ull;
      for (int r = 0; r < a.world; ++r) {
        const u64 t = a.totals[r];
        shard_off += r < a.rank ? t : 0ull;
        T += t;
      }
      off = shard_off + a.tile_sum[fine0];
      if (shard_off != 0ull) {
        u64 quo, rem;
        muldiv_u64(shard_off, a.n_global_u, T, quo, rem);
        nc_start = (u32)quo + (shl53_gt_mul(rem, a.u53, T) ? 1u : 0u);
      }
    } else {
      // weight before this tile and total weight: every block adds the K1 tile sums up itself
      u64 pre = 0ull, tot = 0ull;
      for (long long i = tid; i < a.n_fine; i += RS_BLOCK) {
        const u64 v = a.tile_sum[i];
        tot += v;
        pre += i < fine0 ? v : 0ull;
      }
      pre = wave_sum(pre);
      tot = wave_sum(tot);
      if (lane == 0) {
        sh64[w] = pre;
        shb[w] = tot;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < RS_BLOCK / 64; ++k) {
        off += sh64[k];
        T += shb[k];
      }
      __syncthreads();  // sh64 is reused by the scan below
      if (blockIdx.x == 0 && tid == 0) a.total_out[0] = T;
    }
    tile_scan_blocked(v, sh64);
    const double n_over_t = (double)a.n_global_u / (double)T;   // (one fp64 division per thread, not one per weight)
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
      nc[k] = 0u;
      if (base + k < a.n) {
        u64 quo, rem;
        muldiv_u64(v[k] + off, a.n_global_u, T, n_over_t, quo, rem);
        nc[k] = (u32)quo + (shl53_gt_mul(rem, a.u53, T) ? 1u : 0u);
        a.ncum[base + k] = nc[k];
      }
    }
    prev_tile = nc_start;
    if (off != 0ull && w == 0) {   // (only thread 0 looks at it)
      u64 quo, rem;
      muldiv_u64(off, a.n_global_u, T, n_over_t, quo, rem);
      prev_tile = (u32)quo + (shl53_gt_mul(rem, a.u53, T) ? 1u : 0u);
    }
  } else {
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) nc[k] = (base + k < a.n) ? a.ncum[base + k] : 0u;
    if (tile > 0) prev_tile = a.ncum[tile * RS_TILE - 1];
  }
  // ---- offspring counts need the left neighbour: through LDS
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) snc[tid * RS_ITEMS + k] = nc[k];
  __syncthreads();
  const u32 prev = tid == 0 ? prev_tile : snc[tid * RS_ITEMS - 1];
  u32 z[RS_ITEMS], zs[RS_ITEMS];
  {
    u32 cp = prev;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
      const bool in = base + k < a.n;
      z[k] = (in && nc[k] == cp) ? 1u : 0u;
      zs[k] = z[k];
      if (in) cp = nc[k];
    }
  }
  tile_scan_blocked(zs, sh32);  // inclusive, within the tile
  if (tid == RS_BLOCK - 1) agg_sh = zs[RS_ITEMS - 1];
  __syncthreads();
  if (w == 0) {
    const u32 zex = scan_lookback(a.desc, tile, agg_sh, a.epoch);
    if (tid == 0) zex_sh = zex;
  }
  __syncthreads();
  const u32 zex = zex_sh;
  if (FROM_Q && a.ls_out && tile == (long long)gridDim.x - 1 && tid == 0) {
    // the shard's hand-over record: lost slots, surplus copies (= offspring of the shard - slots + lost), particle 0
    const u32 L = zex + agg_sh;
    const u32 S = (snc[(a.n - 1) - tile * RS_TILE] - nc_start) - (u32)a.n + L;
    a.ls_out[0] = (u64)L | ((u64)S << 32);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      a.ls_out[1 + c] = (u64)__double_as_longlong((c == 2 && a.p0_z_uniform) ? a.p0_z : a.p0[c][0]);
  }
  // ---- ranks of the lost slots; dupes entries of the ancestors with surplus copies
  {
    u32 cp = prev;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
      const long long j = base + k;
      if (j < a.n) {
        const u32 zc = zex + zs[k];  // zero-offspring slots in [0, j]
        const u32 c = nc[k] - cp;
        cp = nc[k];
        if (j >= a.own0 && j < a.own0 + a.own_n) a.zr[j - a.own0] = z[k] ? zc - 1u : ZR_SURVIVOR;
        if (c > 1u) {
          // cumulative surplus before j:  E_{j-1} = ncum_{j-1} - j + zcum_{j-1}   (mod 2^32, the result is >= 0)
          const u32 s = c - 1u, e0 = ((nc[k] - c) - nc_start) - (u32)j + zc;
          if (s <= EXP_HEAVY) {
            for (u32 r = 0; r < s; ++r) a.dupes[e0 + r] = (u32)j;
          } else {
            const u32 slot = atomicAdd(&heavy_n, 1u);
            if (slot < EXP_LIST) {
              heavy[slot][0] = (u32)j;
              heavy[slot][1] = e0;
              heavy[slot][2] = s;
            } else {
              for (u32 r = 0; r < s; ++r) a.dupes[e0 + r] = (u32)j;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const u32 hn = heavy_n < EXP_LIST ? heavy_n : EXP_LIST;
  for (u32 h = 0; h < hn; ++h) {
    const u32 j = heavy[h][0], e0 = heavy[h][1], s = heavy[h][2];
    for (u32 r = tid; r < s; r += RS_BLOCK) a.dupes[e0 + r] = j;
  }
}

// ------------------------------------------------------------------ O(n) exchange: pack the surplus copies
// Surplus copy p of this shard (p-th entry of its local dupes list) has position Spre + p in the GLOBAL dupes order;
// the lost slot of global rank k takes the copy at position k.  Copies whose position falls into this shard's own
// lost ranks [Lpre, Lpre + L) go straight into its receive buffer, the others into the send buffer, from which the
// host sends each peer the contiguous range that intersects the peer's lost ranks.
struct PackArgs {
  StatePtrs src;       // this shard's pre-resample state
  const u32* dupes;    // local ancestor index of every surplus copy
  const u64* lsx;      // hand-over records of every shard (device): word 0 of record r = L_r | S_r << 32
  int rank, world;
  u32 cap;             // capacity of the send buffer (copies): copies beyond it are dropped, the host sees S > cap in
                       // the same records, grows the buffer and packs again
  // A copy travels as ONE record of `nship` doubles -- the components that are not the same value on every particle of
  // the cloud (x, y, yaw right after a predict; all six otherwise), ship[j] = component of word j -- so that the copies a
  // peer needs, contiguous in the global dupes order, are ONE contiguous block: one ncclSend / ncclRecv per peer
  int nship;
  int ship[6];
  double* send;        // [cap][nship]
  double* recv;        // [L][nship]
  // when set, workgroup 0 first copies the records to pinned host memory and then writes the sequence word the host
  // spins on (system-scope release: the records are visible before it) -- a stream synchronisation costs the host tens
  // of microseconds of wake-up, this a few
  u64* host_words;
  u64* host_seq;
  u64 seq;
};
// (sizes come from the records ON THE DEVICE: the kernel is queued before the host has seen them)
__global__ void __launch_bounds__(MCL_BLOCK) k_pack_dupes(PackArgs a) {
  if (a.host_words && blockIdx.x == 0) {
    for (int k = threadIdx.x; k < 4 * a.world; k += MCL_BLOCK) a.host_words[k] = a.lsx[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  u32 Spre = 0u, Lpre = 0u, S = 0u, L = 0u;
  for (int r = 0; r <= a.rank; ++r) {
    const u64 w = a.lsx[4 * (size_t)r];
    const u32 l = (u32)(w & 0xffffffffull), sc = (u32)(w >> 32);
    if (r < a.rank) {
      Lpre += l;
      Spre += sc;
    } else {
      L = l;
      S = sc;
    }
  }
  const u32 Sc = S < a.cap ? S : a.cap;
  for (u32 p = blockIdx.x * MCL_BLOCK + threadIdx.x; p < Sc; p += gridDim.x * MCL_BLOCK) {
    const u32 anc = a.dupes[p];
    const u32 g = Spre + p;
    const bool self = g >= Lpre && g - Lpre < L;
    double* rec = self ? a.recv + (size_t)(g - Lpre) * a.nship : a.send + (size_t)p * a.nship;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < a.nship) rec[j] = a.src.c[a.ship[j]][anc];
  }
}
// ------------------------------------------------------------------ K3: gather + noise (+ fused moments)
#define MOM_COUNT 13   // sum d(x,y,z), sum roll, pitch, yaw, sum wrap(yaw), sum dxx dyy dzz dxy dxz dyz
#define GATHER_MAX_GRID 256
static_assert(RS_BLOCK == (1 << VISIT_OWNER_SHIFT) && GATHER_MAX_GRID == VISIT_OWNER_MASK + 1,
              "k_predict_pose finds the gather workgroup of a slot as (slot >> VISIT_OWNER_SHIFT) & VISIT_OWNER_MASK");
struct GatherArgs {
  StatePtrs src;  // pre-resample state, GLOBAL indexing (state_glob when sharded)
  StatePtrs dst;  // this shard's slice of the new state
  long long n, goff;
  const u32* zr;
  const u32* dupes;
  NoiseArgs nz;
  int add_noise;
  double* part;      // MOMENTS: [MOM_COUNT][gridDim.x] per-block partial sums
  double* sums_out;  // MOMENTS: 13 sums followed by the 3 shifts
  u32* ticket;       // MOMENTS: zero before the launch, left zero
  // components the exchange did not gather because every particle of the cloud holds the same value (z, roll,
  // pitch straight after motion_pred: auv_particle.py:55-57,70): bit c set -> src.c[c] is not read, uni[c] is the value
  unsigned uni_mask;
  double uni[6];
  // O(n) exchange: src is this shard's OWN pre-resample state (local indexing), zr holds LOCAL lost ranks and a lost
  // slot reads its copy's record: recv.c[c] points at component c's word of record 0, records are recv_stride doubles
  // apart (k_pack_dupes); shift_dev = the position of global particle 0 (every rank has it from the hand-over
  // records).  recv_mode 0: the look-up through the dupes list into src (single shard / all-gathered state).
  int recv_mode;
  int recv_stride;
  StatePtrs recv;
  const double* shift_dev;
  // MOMENTS: the visiting order of the NEXT fan sweep (mcl_kernels.h: VisitArgs); visit.okey == nullptr: none
  VisitArgs visit;
};
// UNI: z, roll, pitch are the odometry's on every particle (a.uni_mask == 0x1c: the resample right after a predict --
// every fused step): three state components are kernel arguments, not loads, and the three that are left fit the
// register budget while in flight across the arithmetic (the generic kernel, with six, loads them after it)
// STASH (MOMENTS and UNI, resampling noise on x, y, yaw only -- every fused step of the launch files' covariances): the
// loop does not accumulate.  It parks x, y, yaw of the particle it has just written in LDS (fp64, the first GATHER_STASH
// particles of a thread = 1 M particles per shard; beyond, the second pass re-reads the state) and the 13 sums -- the
// same additions in the same order -- are taken in a second pass, when the Philox / Box-Muller registers are dead:
// the loop runs without its 26 accumulator registers (round 4: 128 VGPRs, the limit of a 16-wave workgroup, and
// anything added to it spilled), and the second pass has room for the visiting order of the next fan sweep
// (mcl_kernels.h: VisitArgs): key from (x, y, wrapped yaw) relative to the previous cloud's bins, rank inside
// (workgroup, bin) from an LDS counter.
#define GATHER_STASH 4
#define GATHER_STASH_LDS (3 * GATHER_STASH * RS_BLOCK * sizeof(double) + VISIT_MAX_BINS * sizeof(u32))
// Box-Muller in fast fp32 (v_log_f32, v_sin_f32 / v_cos_f32 take turns): the visiting order only needs the coming process
// noise to a fraction of a bin
__device__ __forceinline__ void box_muller_fast(u32 ua, u32 ub, float& n0, float& n1) {
  const float u1 = ((float)(ua >> 8) + 0.5f) * (1.f / 16777216.f), u2 = (float)(ub >> 8) * (1.f / 16777216.f);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), log2 in hardware
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ void moments_add(double (&acc)[MOM_COUNT], const double (&v)[6], const double (&shift)[3], double& wy) {
  const double dx = v[0] - shift[0], dy = v[1] - shift[1], dz = v[2] - shift[2];
  acc[0] += dx;
  acc[1] += dy;
  acc[2] += dz;
  acc[3] += v[3];
  acc[4] += v[4];
  acc[5] += v[5];
  wy = wrap_pi(v[5]);
  acc[6] += wy;
  acc[7] += dx * dx;
  acc[8] += dy * dy;
  acc[9] += dz * dz;
  acc[10] += dx * dy;
  acc[11] += dx * dz;
  acc[12] += dy * dz;
}
template <bool MOMENTS, bool UNI = false, bool STASH = false>
__global__ void __launch_bounds__(RS_BLOCK) k_resample_gather(GatherArgs a, const double* __restrict__ replay) {
  static_assert(!STASH || (MOMENTS && UNI), "the stash variant is the fused step's kernel");
  __shared__ double red[MOM_COUNT + 1][RS_BLOCK / 64];
  __shared__ u32 last_sh;
  extern __shared__ __attribute__((aligned(16))) unsigned char gather_lds[];   // STASH: GATHER_STASH_LDS bytes
  double* const stash = (double*)gather_lds;                                                 // [GATHER_STASH][3][RS_BLOCK]
  u32* const vhist = (u32*)(gather_lds + 3 * GATHER_STASH * RS_BLOCK * sizeof(double));      // [VISIT_MAX_BINS]
  double acc[MOM_COUNT];
  double shift[3] = {0.0, 0.0, 0.0};
  const bool visit = STASH && a.visit.okey != nullptr;
  if (MOMENTS) {
#pragma unroll
    for (int c = 0; c < MOM_COUNT; ++c) acc[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c)  // a member of the cloud, the same on every shard
      shift[c] = a.recv_mode ? a.shift_dev[c] : (((a.uni_mask >> c) & 1u) ? a.uni[c] : a.src.c[c][0]);
  }
  if (visit)   // (read and written only after the barrier behind the loop)
    for (int k = threadIdx.x; k < a.visit.nb; k += RS_BLOCK) vhist[k] = 0u;
  // The gather of a resampled cloud is a chain of three dependent, scattered loads per particle (zr -> dupes -> state)
  // in front of ~500 instructions of Philox / Box-Muller arithmetic that need none of them, and with four waves per
  // SIMD nothing else hides the chain (round 3: 70 % of the wave cycles waiting, rocprofv3 SQ_WAIT_ANY).
  // The loop is therefore software-pipelined by hand: the state loads of THIS particle and the index load of the NEXT
  // one are issued before the arithmetic, the next particle's ancestor index after it.  (A batch of two particles per
  // iteration spilled 68 B per lane: 13 fp64 accumulators leave no room for a second state.)
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  u32 r = i < a.n ? a.zr[i] : ZR_SURVIVOR;
  long long src = 0;
  bool from_recv = false;
  if (i < a.n) {
    const bool surv = r == ZR_SURVIVOR;
    src = a.recv_mode ? (surv ? i : (long long)r * a.recv_stride) : (surv ? a.goff + i : (long long)a.dupes[r]);
    from_recv = a.recv_mode && !surv;
  }
  int it = 0;
  for (; i < a.n; i += stride) {
    const long long g = a.goff + i;
    double v[6];
    if (UNI) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {   // this particle's state: in flight during the arithmetic below
        const double* from = from_recv ? a.recv.c[c] : a.src.c[c];
        v[c] = (c >= 2 && c <= 4) ? a.uni[c] : from[src];
      }
    }
    const long long in = i + stride;
    const u32 rn = in < a.n ? a.zr[in] : ZR_SURVIVOR;   // the next particle's slot record: likewise
    double z[6] = {0, 0, 0, 0, 0, 0};
    if (a.add_noise) {
      if (replay) {
#pragma unroll
        for (int c = 0; c < 6; ++c) z[c] = replay[i * 6 + c];
      } else {
        native_normals6(g, a.nz, z);
      }
    }
    // ... and its ancestor (rn has arrived by now): in flight during the stores and the moments
    const bool survn = rn == ZR_SURVIVOR;
    const long long srcn = a.recv_mode ? (survn ? in : (long long)rn * a.recv_stride) : (survn ? a.goff + in : (long long)a.dupes[rn]);
    if (!UNI) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double* from = from_recv ? a.recv.c[c] : a.src.c[c];
        v[c] = ((a.uni_mask >> c) & 1u) ? a.uni[c] : from[src];
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      v[c] = v[c] + a.nz.sq[c] * z[c];
      a.dst.c[c][i] = v[c];
    }
    if (STASH) {
      if (it < GATHER_STASH) {
        stash[(it * 3 + 0) * RS_BLOCK + threadIdx.x] = v[0];
        stash[(it * 3 + 1) * RS_BLOCK + threadIdx.x] = v[1];
        stash[(it * 3 + 2) * RS_BLOCK + threadIdx.x] = v[5];
      }
      ++it;
    } else if (MOMENTS) {
      double wy;
      moments_add(acc, v, shift, wy);
    }
    src = srcn;
    from_recv = a.recv_mode && !survn;
  }
  if (!MOMENTS) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float vacc = 0.f;
  if (STASH) {
    // ---- second pass: the 13 sums (z, roll, pitch are the three constants: their noise covariances are zero) and the
    // bins of the visiting order.  A record that is not valid yet is all zeros: every particle lands in bin 0 -- still
    // a bijection -- and the bins follow a step later.
    VisitPar vp;
    float ox = 0.f, oy = 0.f, ow = 0.f;
    if (visit) {
      vp = *a.visit.par_in;
      ox = (float)(vp.lo[0] - shift[0]) * vp.inv[0];
      oy = (float)(vp.lo[1] - shift[1]) * vp.inv[1];
      ow = (float)(vp.lo[2] - vp.mean[2]) * vp.inv[2];
      __syncthreads();   // the counters are zero
    }
    int k = 0;
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < a.n; j += stride, ++k) {
      double v[6];
      if (k < GATHER_STASH) {
        v[0] = stash[(k * 3 + 0) * RS_BLOCK + threadIdx.x];
        v[1] = stash[(k * 3 + 1) * RS_BLOCK + threadIdx.x];
        v[5] = stash[(k * 3 + 2) * RS_BLOCK + threadIdx.x];
      } else {
        v[0] = a.dst.c[0][j];
        v[1] = a.dst.c[1][j];
        v[5] = a.dst.c[5][j];
      }
      v[2] = a.uni[2];
      v[3] = a.uni[3];
      v[4] = a.uni[4];
      double wy;
      moments_add(acc, v, shift, wy);
      if (visit) {
        const float dw = (float)(wy - vp.mean[2]);
        vacc = fmaf(dw, dw, vacc);   // (the yaw spread the NEXT bins need: the 14th partial sum; fp32 is plenty)
        // where the next predict's noise will put the particle (auv_particle.py:40,47,65: the same Philox block as
        // k_predict_pose, the normals in fast fp32) -- the common motion moves every particle alike
        const u32x4 o = philox4x32((u32)(a.goff + j), 0u, a.visit.pstep, 1u, a.nz.k0, a.nz.k1);
        float n0, n1, n5, unused;
        box_muller_fast(o.x, o.y, n0, n1);
        box_muller_fast(o.z, o.w, n5, unused);
        const float px = fmaf(a.visit.psq[0], n0, (float)(v[0] - shift[0])), py = fmaf(a.visit.psq[1], n1, (float)(v[1] - shift[1]));
        const float pw = fmaf(a.visit.psq[2], n5, dw);
        // (fmaxf / min: a NaN coordinate lands in bin 0)
        const int kx = min((int)fmaxf(fmaf(px, vp.inv[0], -ox), 0.f), a.visit.nbx - 1);
        const int ky = min((int)fmaxf(fmaf(py, vp.inv[1], -oy), 0.f), a.visit.nby - 1);
        const int kw = min((int)fmaxf(fmaf(pw, vp.inv[2], -ow), 0.f), a.visit.nbw - 1);
        const u32 key = (u32)((kx * a.visit.nby + ky) * a.visit.nbw + kw);
        const u32 rank = atomicAdd(&vhist[key], 1u);   // LDS: the particle's place among this workgroup's members of the bin
        a.visit.okey[j] = key | (rank << VISIT_KEY_BITS);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MOM_COUNT; ++c) {
    const double s = wave_sum_dpp(acc[c]);   // (13 shuffle trees through LDS were half of this kernel's 9 us tail)
    if (lane == 0) red[c][w] = s;
  }
  if (visit) {
    const float s = wave_sum_dpp(vacc);
    if (lane == 0) red[MOM_COUNT][w] = (double)s;
  }
  __syncthreads();
  if (visit) {   // this workgroup's row of the count matrix, two 16-bit counts per word (plain stores: read by the NEXT launch, k_visit_scan)
    u32* row = (u32*)(a.visit.cnt + (size_t)blockIdx.x * a.visit.nb);
    for (int k = threadIdx.x; 2 * k < a.visit.nb; k += RS_BLOCK) row[k] = vhist[2 * k] | (vhist[2 * k + 1] << 16);
  }
  if (w == 0) {
    // wave 0: lanes 0..12 publish this block's 13 partial sums (ONE write-through store instruction), drain it,
    // then lane 0 draws the ticket -- no release fence: only these words are read inside the launch
    if (lane < MOM_COUNT + (visit ? 1 : 0)) {
      double s = 0.0;
      for (int k = 0; k < RS_BLOCK / 64; ++k) s += red[lane][k];
      store_agent(&a.part[(size_t)lane * gridDim.x + blockIdx.x], s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MCL_TICKET_FENCE();
    if (lane == 0) last_sh = (atomicAdd(a.ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
    MCL_TICKET_FENCE();
  }
  __syncthreads();
  if (!last_sh) return;
  // ---- the last block adds the partials in a fixed order (the result does not depend on which block is last):
  // wave c takes component c, lane l the blocks l, l + 64, ...; then the fixed shuffle tree
  // (no acquire fence -- an L1 invalidate, ~1.7 us --: the partials were stored write-through and drained before
  //  their block's ticket, and are read here by agent-scope loads, issued after the ticket's value has come back)
  if (w < MOM_COUNT + (visit ? 1 : 0)) {
    double s = 0.0;
    for (unsigned b = lane; b < gridDim.x; b += 64) s += load_agent(&a.part[(size_t)w * gridDim.x + b]);
    s = wave_sum_dpp(s);
    if (lane == 0) {
      if (w < MOM_COUNT) a.sums_out[w] = s;
      red[w][0] = s;
    }
  }
  if (threadIdx.x < 3) a.sums_out[MOM_COUNT + threadIdx.x] = shift[threadIdx.x];
  if (threadIdx.x == 0) *a.ticket = 0u;
  if (!visit) return;
  // ---- bins of the next gather: mean +- range * sigma of THIS cloud, moved on by the drift since the previous one
  __syncthreads();
  if (threadIdx.x == 0) {
    const VisitPar vp = *a.visit.par_in;
    const double inv_n = 1.0 / (double)a.n;
    const double m[3] = {red[0][0] * inv_n, red[1][0] * inv_n, red[6][0] * inv_n};   // about the shift / plain
    const double dm = m[2] - vp.mean[2];
    const double var[3] = {red[7][0] * inv_n - m[0] * m[0], red[8][0] * inv_n - m[1] * m[1], red[MOM_COUNT][0] * inv_n - dm * dm};
    const double mean[3] = {shift[0] + m[0], shift[1] + m[1], m[2]};
    const int nbin[3] = {a.visit.nbx, a.visit.nby, a.visit.nbw};
    VisitPar o;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // (var = E[(v - ref)^2] - (E[v] - ref)^2 for any reference: the yaw's is the previous mean, x and y's the shift)
      const double sd = sqrt(var[c] > 1e-18 ? var[c] : 1e-18);
      const double drift = vp.valid ? mean[c] - vp.mean[c] : 0.0;
      o.mean[c] = mean[c];
      o.lo[c] = mean[c] + drift - (double)a.visit.range * sd;
      o.inv[c] = (float)((double)nbin[c] / (2.0 * (double)a.visit.range * sd));
      ok = ok && mean[c] == mean[c] && sd == sd && sd < 1e30;
    }
    o.valid = ok ? 1 : 0;
    if (!ok) {   // (a cloud with NaN moments: start over from "no bins yet")
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o.mean[c] = o.lo[c] = 0.0;
        o.inv[c] = 0.f;
      }
    }
    *a.visit.par_out = o;
  }
}

// ------------------------------------------------------------------ visiting order: counts -> positions
// cnt[G][nb] (one row per gather workgroup) -> base[G][nb]: position of the workgroup's first particle of the bin =
// particles in earlier bins + particles of earlier workgroups in the same bin.  One workgroup per 64 bins (at most 64
// of them: all resident), wave r takes a run of ceil(G / 16) rows; the particles in earlier bins come from a look-back
// over the other workgroups' descriptors -- ONE wave reads all of them at once --, so there is no second pass and no
// "last block".
__global__ void __launch_bounds__(1024) k_visit_scan(VisitArgs a, int G, u32* ticket) {
  __shared__ u32 sh[16][64];
  __shared__ u32 binbase[64];
  __shared__ u32 slice_sh;
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  // (slices are handed out by a ticket, like k_cdf_expand's tiles: a workgroup only ever waits for slices that are
  //  already running, whatever else shares the chip -- the shards of a LOCAL group launch their scans side by side)
  if (threadIdx.x == 0) {
    MCL_TICKET_FENCE();
    const u32 t = atomicAdd(ticket, 1u);
    MCL_TICKET_FENCE();
    slice_sh = t;
    if (t == gridDim.x - 1u) atomicExch(ticket, 0u);   // every ticket of this launch has been drawn
  }
  __syncthreads();
  const u32 slice = slice_sh;
  const int bin = (int)slice * 64 + lane;
  const int rpg = (G + 15) >> 4;
  u32 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = r * rpg + k;
    v[k] = (k < rpg && row < G) ? (u32)a.cnt[(size_t)row * a.nb + bin] : 0u;
  }
  u32 t = 0u;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u32 x = v[k];
    v[k] = t;
    t += x;
  }
  sh[r][lane] = t;
  __syncthreads();
  u32 off = 0u, tot = 0u;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u32 x = sh[k][lane];
    off += k < r ? x : 0u;
    tot += x;
  }
  if (r == 0) {
    // this workgroup's 64 bin totals -> its descriptor; the particles of the workgroups before it
    const u32 incl = wave_scan_incl_dpp(tot);
    const u32 mine = readlane63(incl);
    const u64 tag = (u64)a.epoch << 32;
    if (lane == 0) store_agent(&a.desc[slice], tag | (u64)mine);
    u32 before = 0u;
    for (;;) {
      const u64 d = lane < (int)slice ? load_agent(&a.desc[lane]) : tag;
      if (__ballot((d >> 32) == (u64)a.epoch) == ~0ull) {
        before = wave_sum_dpp(lane < (int)slice ? (u32)d : 0u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    binbase[lane] = before + incl - tot;
  }
  __syncthreads();
  const u32 bb = binbase[lane] + off;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = r * rpg + k;
    if (k < rpg && row < G) a.base[(size_t)row * a.nb + bin] = bb + v[k];
  }
}
