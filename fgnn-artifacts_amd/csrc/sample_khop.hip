// sample_khop.hip -- fixed-fanout uniform neighbour sampling without replacement for gfx950.
//
// Replaces GPUSampleKHop0 (reference samgraph/common/cuda/cuda_sampling_khop0.cu:41-253) and
// GPUSampleKHop2 (cuda_sampling_khop2.cu:41-252).  Results are bit-identical to
// oracle/fgnn_oracle.c (Philox mode), including khop2's in-place mutation of the CSR row.
//
// MI355X design (not the reference's thread-per-seed + pad + count + compact pipeline):
//  * the number of edges a seed emits, min(deg, fanout), is known from indptr alone, so the output
//    offsets are computed BEFORE sampling (count kernel -> one-workgroup scan) and the sampler
//    writes the compacted COO directly: no padded temporaries, no compaction pass;
//  * inside the sampler a workgroup owns S consecutive seeds.  Phase A (one lane per long row)
//    resolves WHICH CSR positions are emitted -- for khop2 by simulating the partial Fisher-Yates
//    on positions only (LDS-resident swap log, no memory traffic), for khop0 by a wave-parallel
//    reservoir (LDS atomicMax of the winning j per slot).  Phase B is output-slot-parallel: lane p
//    handles output edge p, so the writes of out_src/out_dst are fully coalesced and the CSR reads of
//    short rows are contiguous; every lane has independent loads in flight.  Phase C (khop2) applies
//    the row mutation with the values phase B already fetched;
//  * sizes may live on the device (d_num_input) so layers chain without a host round trip.
#include <cstdlib>

#include "fgnn_device.h"

namespace fgnn {

namespace {

// ---------------------------------------------------------------------------------------------
// count: c_i = min(deg(input[i]), fanout); one partial sum per workgroup of S seeds
template <int S>
__global__ __launch_bounds__(S) void khop_count_kernel_s(const uint32_t *__restrict__ indptr,
                                                         const uint32_t *__restrict__ input, size_t num_input,
                                                         const uint32_t *d_num_input, size_t cap, uint32_t fanout,
                                                         uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t sh[S / kWave];
  const size_t n = resolve_count(num_input, d_num_input, cap);
  const size_t i = (size_t)blockIdx.x * S + threadIdx.x;
  uint32_t c = 0;
  if (i < n) {
    const uint32_t rid = input[i];
    const uint32_t len = indptr[rid + 1] - indptr[rid];
    c = len < fanout ? len : fanout;
  }
  uint32_t tot;
  (void)block_exclusive_scan<S / kWave>(c, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// locate the seed that owns output slot p: largest k with lo[k] <= p   (lo has S+1 entries)
template <int S>
__device__ __forceinline__ int owner_of_slot(const uint32_t *lo, uint32_t p) {
  int a = 0, b = S;  // invariant: lo[a] <= p < lo[b]
#pragma unroll
  for (int step = S; step > 1; step >>= 1) {
    const int mid = (a + b) >> 1;
    if (lo[mid] <= p) a = mid; else b = mid;
  }
  return a;
}

// ---------------------------------------------------------------------------------------------
// KHOP2 == false : reservoir sampling (khop0.cu:41-90), CSR untouched
// KHOP2 == true  : partial Fisher-Yates in place (khop2.cu:41-89)
// S seeds per workgroup, S threads.  Dynamic LDS: KHOP2 ? 3*F*S words : F*S words.
constexpr uint32_t kWriteBack = 0x80000000u;  // flag on a logged swap position: this step's value survives in the row

struct FuseArgs {               // dedup insert fused into phase B (engine path)
  HtView t;                     // t.table == null: not fused
  uint32_t *pos;                // bucket of every emitted edge (disp != null: the insert's outcome instead)
  uint32_t *disp;               // non-null: resolving insert (fgnn_device.h), the batch's last fill
  uint32_t *d_num_items;        // the table's {count, count before the running fill}
  BatchStart start;             // start.n2o != null: this is the first launch of a batch
};

// khop0 on hub rows.  The reservoir draws one number per row ELEMENT (khop0.cu:41-90); a row of 10^5..10^6 entries
// (R-MAT hubs sit in nearly every frontier) walked by the one workgroup that owns its seed was the kernel's critical
// path.  Rows beyond kSplitRow entries are therefore drawn by the WHOLE chip before the sampler runs: a scan lists
// them (hub index per seed), khop0_hub_kernel strides all workgroups over each listed row's Philox blocks and keeps,
// per output slot, the largest winning element in a small global table; the sampler copies those winners instead of
// drawing.  Same draws, same winners (max is order-free): bit-identical.
constexpr uint32_t kSplitRow = 16384;
struct HubSplit {
  uint32_t *of_seed;   // [cap] hub index of seed i, or kEmpty; null: no split (scratch too small, or khop2)
  uint32_t *count;     // [1] hubs claimed (may exceed cap_hubs: the surplus stays with the sampler)
  uint32_t *list;      // [cap_hubs][3] seed position, row offset, row length
  uint32_t *win;       // [cap_hubs][F] slot winners, initialised to the slot's own position
  uint32_t cap_hubs;
};

__global__ __launch_bounds__(256) void khop0_hub_scan_kernel(const uint32_t *__restrict__ indptr,
                                                             const uint32_t *__restrict__ input, size_t num_input,
                                                             const uint32_t *d_num_input, size_t cap, uint32_t F,
                                                             HubSplit hub) {
  const size_t n = resolve_count(num_input, d_num_input, cap);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  uint32_t h = FGNN_EMPTY_KEY;
  if (i < n) {
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid], len = indptr[rid + 1] - off;
    if (len > kSplitRow && len > F) {
      h = atomicAdd(hub.count, 1u);
      if (h < hub.cap_hubs) {
        hub.list[3 * h] = (uint32_t)i;
        hub.list[3 * h + 1] = off;
        hub.list[3 * h + 2] = len;
        for (uint32_t q = 0; q < F; ++q) hub.win[(size_t)h * F + q] = q;
      } else {
        h = FGNN_EMPTY_KEY;
      }
    }
  }
  hub.of_seed[i] = h;
}

__global__ __launch_bounds__(256) void khop0_hub_kernel(HubSplit hub, uint32_t F, uint64_t seed, uint64_t batch_key,
                                                        uint32_t tag) {
  // The Philox blocks of ALL listed rows form one flat index space (a first version walked the rows one after the other:
  // a dependent list read and a nearly empty grid per short hub, 0.26 -> 0.53 ms per step): every workgroup builds the
  // prefix of the rows' block counts in LDS, a lane finds the row of each of its blocks by binary search there.
  __shared__ uint32_t sh_pref[1024 + 1];
  __shared__ uint32_t sh_scan[256 / kWave];
  const uint32_t claimed = *hub.count;
  const uint32_t nh = claimed < hub.cap_hubs ? claimed : hub.cap_hubs;  // <= 1024
  if (nh == 0) return;
  const uint32_t first_blk = F >> 2;
  uint32_t running = 0;
  for (uint32_t h0 = 0; h0 < nh; h0 += 256) {
    const uint32_t h = h0 + threadIdx.x;
    uint32_t nblk = 0;
    if (h < nh) {
      const uint32_t klen = hub.list[3 * h + 2];
      const uint32_t last_blk = (klen + 3) >> 2;  // blocks [first_blk, last_blk) hold elements F .. klen - 1
      nblk = last_blk > first_blk ? last_blk - first_blk : 0;
    }
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<256 / kWave>(nblk, sh_scan, &tot);
    if (h < nh) sh_pref[h] = running + ex;
    running += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) sh_pref[nh] = running;
  __syncthreads();
  const uint32_t total = sh_pref[nh];
  const uint32_t lanes = gridDim.x * 256u;
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < total; w += lanes) {
    uint32_t a = 0, b = nh;  // sh_pref[a] <= w < sh_pref[b]
    while (b - a > 1) {
      const uint32_t mid = (a + b) >> 1;
      if (sh_pref[mid] <= w) a = mid; else b = mid;
    }
    const uint32_t item = hub.list[3 * a], klen = hub.list[3 * a + 2];
    const uint32_t jb = first_blk + (w - sh_pref[a]);
    uint32_t *win = hub.win + (size_t)a * F;
    const u32x4 blk = philox_block(seed, batch_key, tag, item, jb);
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t j = (jb << 2) + u;
      if (j >= F && j < klen) {
        const uint32_t kk = pick_word(blk, u) % (j + 1);
        if (kk < F) atomicMax(&win[kk], j);
      }
    }
  }
}

// S seeds per workgroup handled by T >= S threads: the per-seed phases (0, A, C) use the first S threads, the
// per-edge phase B uses all T, so a small frontier still gives every CU several waves to overlap latencies.
template <int S, int T, bool KHOP2, int FMAX>
__global__ __launch_bounds__(T) void khop_sample_kernel(const uint32_t *__restrict__ indptr, uint32_t *indices,
                                                        const uint32_t *__restrict__ input, size_t num_input,
                                                        const uint32_t *d_num_input, size_t cap, uint32_t F,
                                                        const uint32_t *__restrict__ block_offsets,
                                                        uint32_t *__restrict__ out_src, uint32_t *__restrict__ out_dst,
                                                        int src_mode, uint64_t seed, uint64_t batch_key,
                                                        uint32_t tag, FuseArgs fuse, ScanWs scan, size_t *d_num_out,
                                                        HubSplit hub FGNN_ABLATE_PARAM) {
  constexpr int NW = T / kWave;
  static_assert(T >= S && T % kWave == 0, "threads per workgroup");
  extern __shared__ uint32_t dyn[];
  __shared__ uint32_t sh_scan[NW];
  __shared__ uint32_t sh_off[S], sh_len[S], sh_rid[S], sh_lo[S + 1];
  __shared__ uint32_t sh_hub[KHOP2 ? 1 : S];  // khop0: hub index of the seed (row drawn by khop0_hub_kernel), or kEmpty

  uint32_t *sh_o = dyn;                         // [F][S] khop2: origin position, then fetched value
  uint32_t *sh_s = dyn + (size_t)F * S;         // [F][S] khop2: swap-log position s_j
  uint32_t *sh_w = dyn + (size_t)2 * F * S;     // [F][S] khop2: origin written to s_j, then its value

  const int tid = threadIdx.x;
  const size_t n = resolve_count(num_input, d_num_input, cap);
  // this kernel is pass 1 of the dedup fill: the later passes want the item count from before the fill
  if (fuse.t.table && blockIdx.x == 0 && tid == 0) {
    if (fuse.start.n2o) {  // first launch of a batch: the seeds ARE the table's first n items
      fuse.d_num_items[0] = (uint32_t)n;
      fuse.d_num_items[1] = (uint32_t)n;
      fgnn_batch_meta *m = fuse.start.meta;
      if (m) {  // header of the batch summary; this launch's own num_edge entry is written by its last tile
        m->key = fuse.start.key;
        for (uint32_t l = 0; l < FGNN_MAX_LAYERS; ++l) {
          if (l != fuse.start.layer) m->num_edge[l] = 0;
          m->num_src[l] = 0;
          m->num_dst[l] = 0;
        }
        m->num_layers = fuse.start.num_layers;
        m->num_input = (uint32_t)n;
        m->num_output = (uint32_t)n;
        m->num_miss = 0;
        m->num_cache = 0;
        m->overflow = 0;
        m->t_start = wall_clock64();
        m->t_sampled = 0;
        m->t_closed = 0;
      }
    } else {
      fuse.d_num_items[1] = fuse.d_num_items[0];
    }
  }
  // single-pass mode (scan.desc != null): the output offset comes from the prefix over the earlier workgroups'
  // edge counts -- no count kernel, no scan kernel
  const bool single_pass = scan.desc != nullptr;
  __shared__ uint32_t sh_tile[2];
  const uint32_t tile = scan_take_tile(scan, sh_tile);  // blockIdx.x unless the ticket A/B switch is on
  const size_t first = (size_t)tile * S;
  const uint32_t last_tile = n ? (uint32_t)((n - 1) / S) : 0u;  // tiles beyond it have no seeds and nobody waits for them
  if (tile > last_tile) return;  // whole workgroup exits together
  phase_mark(scan, tile, 0);
  // the per-seed phases (0, A, C) run on ONE S-lane group of the workgroup; which one rotates with the tile id so
  // that the ALU-heavy swap simulation of the workgroups sharing a CU does not pile up on the same SIMD
  const int k = tid % S;
  const bool seed_lane = (uint32_t)(tid / S) == tile % (uint32_t)(T / S);
  const size_t i = first + k;
  uint32_t rid = 0, off = 0, len = 0;
  if (seed_lane && i < n) {
    rid = input[i];
    off = indptr[rid];
    len = indptr[rid + 1] - off;
    if (fuse.t.table && fuse.start.n2o) {  // FillWithUnique: seed i gets local id i (min-insert: order does not matter)
      (void)ht_insert_min(fuse.t, rid, (uint32_t)i);
      fuse.start.n2o[i] = rid;
      if (fuse.start.items_copy) fuse.start.items_copy[i] = rid;
    }
  }
  const uint32_t c = len < F ? len : F;
  uint32_t total;
  const uint32_t lo = block_exclusive_scan<NW>(c, sh_scan, &total);
  if (seed_lane) {
    sh_off[k] = off;
    sh_len[k] = len;
    sh_rid[k] = rid;
    sh_lo[k] = lo;
    if (!KHOP2) sh_hub[k] = (hub.of_seed && i < n) ? hub.of_seed[i] : FGNN_EMPTY_KEY;
  }
  if (tid == 0) sh_lo[S] = total;
  // single pass: the tile's edge count is known here, long before its offset is needed (after phase A)
  if (single_pass) scan_publish_aggregate(scan, tile, total);
  const bool big = seed_lane && len > F;

  // ---- phase A: which CSR positions does a long row emit? ---------------------------------
  if (KHOP2) {
    // A.1 (all T threads): the draws.  sel_j = philox(item, j) % (len - j) depends only on the row length, so the
    // T/S lane groups split the Philox blocks of every long row between them and leave sel_j in the swap-log slots.
    // This takes the Philox rounds and the 32-bit modulos off the one wave that runs the serial recurrence below.
    __syncthreads();  // sh_len of every seed is visible
    {
      constexpr uint32_t G = T / S;
      const uint32_t klen = sh_len[k];
      if (klen > F && !(ablate & 1u)) {
        const uint32_t item = (uint32_t)(first + k);
        for (uint32_t blk_id = (uint32_t)tid / S; blk_id * 4 < F; blk_id += G) {
          const u32x4 blk = philox_block(seed, batch_key, tag, item, blk_id);
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t j = blk_id * 4 + u;
            if (j < F) sh_s[j * S + k] = pick_word(blk, u) % (klen - j);
          }
        }
      }
    }
    __syncthreads();
    if (big && (ablate & 1u)) {  // profiling only: skip the swap simulation
      for (uint32_t j = 0; j < F; ++j) { sh_s[j * S + k] = j; sh_w[j * S + k] = j; sh_o[j * S + k] = j; }
    } else if (big) {
      // A.2 (one lane per long row): simulate `for j: emit A[sel_j]; swap(A[sel_j], A[len-1-j])` on POSITIONS.
      // Content of a position p at step j = origin written by the last earlier step i with
      // s_i == p, else p itself (a consumed tail position len-1-i is never touched again).
      //   o_j = content(sel_j)      -> emitted origin
      //   w_j = content(len-1-j)    -> origin that moves into position sel_j
      if (FMAX > 0) {
        // fanout <= FMAX: the swap log lives in registers (fully unrolled, static indices); the LDS
        // copies below are write-only here -- no LDS round trip inside the O(F^2) recurrence.
        // Write-back flag (bit 31 of the logged position; row offsets stay below 2^31 by the host check):
        // position s_j receives a value iff it is not a consumed tail slot and no later step writes it again;
        // "a later step hits the same position" is the very comparison the recurrence makes for o_j.
        uint32_t rs[FMAX > 0 ? FMAX : 1], rw[FMAX > 0 ? FMAX : 1];
        uint32_t keep = 0xffffffffu;  // bit q: step q's write to s_q is still the last one
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
          if ((uint32_t)j < F) {
            const uint32_t sel = sh_s[j * S + k];
            const uint32_t t = len - 1 - (uint32_t)j;
            uint32_t o = sel, w = t;
#pragma unroll
            for (int q = 0; q < j; ++q) {
              const bool hit = rs[q] == sel;
              o = hit ? rw[q] : o;
              keep = hit ? keep & ~(1u << q) : keep;
              w = rs[q] == t ? rw[q] : w;
            }
            rs[j] = sel;
            rw[j] = w;
            sh_w[j * S + k] = w;
            sh_o[j * S + k] = o;
          }
        }
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
          if ((uint32_t)j < F) {
            const bool wb = rs[j] < len - F && ((keep >> j) & 1u);
            sh_s[j * S + k] = rs[j] | (wb ? kWriteBack : 0u);
          }
        }
      } else {
        for (uint32_t j = 0; j < F; ++j) {
          const uint32_t sel = sh_s[j * S + k];
          const uint32_t t = len - 1 - j;
          uint32_t o = sel, w = t;
          for (uint32_t q = 0; q < j; ++q) {
            const uint32_t sq = sh_s[q * S + k];
            const uint32_t wq = sh_w[q * S + k];
            if (sq == sel) o = wq;
            if (sq == t) w = wq;
          }
          sh_w[j * S + k] = w;
          sh_o[j * S + k] = o;
        }
        for (uint32_t j = 0; j < F; ++j) {
          const uint32_t sj = sh_s[j * S + k];
          bool wb = sj < len - F;
          for (uint32_t q = j + 1; q < F && wb; ++q) wb = (sh_s[q * S + k] & ~kWriteBack) != sj;
          if (wb) sh_s[j * S + k] = sj | kWriteBack;
        }
      }
    }
  } else {
    // reservoir: slot k ends up with A[max{j >= F : draw_j % (j+1) == k}], or A[k] if none.
    if (seed_lane)
      for (uint32_t q = 0; q < F; ++q) sh_o[q * S + k] = q;
    __syncthreads();
    // One draw per row ELEMENT beyond the fanout (khop0.cu:41-90): rows can be millions long (R-MAT hubs), and the
    // reference walks each with ONE thread.  Rows up to kHugeRow elements: a wave per row; longer ones: the whole
    // workgroup per row, one after the other (a hub in a tile used to keep one wave busy for the others' idle time).
    constexpr uint32_t kHugeRow = 4096;
    auto reservoir = [&](int row, uint32_t klen, uint32_t first_lane, uint32_t lanes) {
      const uint32_t item = (uint32_t)(first + row);
      // a lane handles 4 consecutive j (one Philox block): j4 = 4*(F/4 + lane + lanes*it)
      for (uint32_t jb = (F >> 2) + first_lane; (jb << 2) < klen; jb += lanes) {
        const u32x4 blk = philox_block(seed, batch_key, tag, item, jb);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t j = (jb << 2) + u;
          if (j >= F && j < klen) {
            const uint32_t kk = pick_word(blk, u) % (j + 1);
            if (kk < F) atomicMax(&sh_o[kk * S + row], j);
          }
        }
      }
    };
    for (int row = wave_id(); row < S; row += NW) {
      const uint32_t klen = sh_len[row];
      if (klen > F && klen <= kHugeRow) reservoir(row, klen, (uint32_t)lane_id(), kWave);  // wave-uniform
    }
    for (int row = 0; row < S; ++row) {
      const uint32_t klen = sh_len[row];
      if (klen <= kHugeRow) continue;  // workgroup-uniform
      const uint32_t h = sh_hub[row];
      if (h == FGNN_EMPTY_KEY) {
        reservoir(row, klen, (uint32_t)tid, T);
      } else {  // drawn by the whole chip beforehand: the winners are in the hub table
        for (uint32_t q = tid; q < F; q += T) sh_o[q * S + row] = hub.win[(size_t)h * F + q];
      }
    }
  }
  __syncthreads();
  phase_mark(scan, tile, 1);

  // ---- phase B: one lane per output edge, 4 edges per lane in flight ---------------------------
  size_t base;
  if (single_pass) {
    // a waiter that outlasts the poll budget recomputes a missing tile's edge count itself (forward progress without
    // assumptions about dispatch, fgnn_device.h): min(deg, F) over the tile's seeds, from immutable inputs
    base = scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
      uint32_t cm = 0;
      const size_t im = (size_t)m * S + tid;
      if (tid < S && im < n) {
        const uint32_t r = input[im];
        const uint32_t l = indptr[r + 1] - indptr[r];
        cm = l < F ? l : F;
      }
      uint32_t tot_m;
      (void)block_exclusive_scan<NW>(cm, sh_scan, &tot_m);
      return tot_m;
    });
    phase_mark(scan, tile, 2);
    if (tile == last_tile && tid == 0 && d_num_out) *d_num_out = base + total;
  } else {
    base = block_offsets[tile];
  }
  constexpr int UB = 4;
  for (uint32_t p0 = tid; p0 < total; p0 += T * UB) {
    uint32_t v[UB], wv[UB], slot[UB], srcv[UB];
    bool live[UB], bigv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t p = p0 + u * T;
      live[u] = p < total;
      bigv[u] = false;
      slot[u] = 0;
      srcv[u] = 0;
      v[u] = 0;
      wv[u] = 0;
      if (live[u]) {
        const int k = owner_of_slot<S>(sh_lo, p);
        const uint32_t j = p - sh_lo[k];
        const uint32_t koff = sh_off[k];
        bigv[u] = sh_len[k] > F;
        slot[u] = j * S + k;
        const uint32_t pos = bigv[u] ? sh_o[slot[u]] : j;
        srcv[u] = src_mode == FGNN_SRC_LOCAL ? (uint32_t)(first + k) : sh_rid[k];
        if (!(ablate & 8u)) {
          v[u] = indices[koff + pos];
          if (KHOP2 && bigv[u]) wv[u] = indices[koff + sh_w[slot[u]]];
        }
      }
    }
    // FillWithDuplicates pass 1 right here: the neighbour ids are in registers, the edge indices are known
    uint32_t bucket[UB];
    if (fuse.t.table && !(ablate & 2u)) {
      uint32_t ival[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) ival[u] = fuse.t.pend | (uint32_t)(base + p0 + u * T);
      if (fuse.disp) ht_insert_resolve_batch<UB>(fuse.t, v, ival, live, fuse.disp, bucket);
      else ht_insert_min_batch<UB>(fuse.t, v, ival, live, bucket);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (live[u]) {
        const uint32_t p = p0 + u * T;
        out_dst[base + p] = v[u];
        out_src[base + p] = srcv[u];
        if (fuse.t.table) fuse.pos[base + p] = (ablate & 2u) ? (fuse.disp ? FGNN_EMPTY_KEY : kNoBucket) : bucket[u];
        if (KHOP2 && bigv[u]) {
          sh_o[slot[u]] = v[u];   // value that lands in the consumed tail slot len-1-j
          sh_w[slot[u]] = wv[u];  // value that lands in position s_j
        }
      }
    }
  }

  // ---- phase C (khop2): apply the swaps to the CSR row -- one lane per (row, step) like phase B, so the F tail
  // stores of a row are consecutive addresses in consecutive lanes (1-2 lines per row instead of F separate
  // partial-line stores from one lane) -------------------------------------------------------------------------
  phase_mark(scan, tile, 3);
  if (KHOP2 && !(ablate & 4u)) {
    __syncthreads();  // every read of the old row contents above has been consumed
    for (uint32_t p = tid; p < total; p += T) {
      const int kk = owner_of_slot<S>(sh_lo, p);
      const uint32_t klen = sh_len[kk];
      if (klen > F) {
        const uint32_t j = p - sh_lo[kk];
        const uint32_t koff = sh_off[kk];
        const uint32_t sl = j * S + kk;
        indices[koff + klen - 1 - j] = sh_o[sl];  // the emitted value moves to the consumed tail slot
        const uint32_t sj = sh_s[sl];
        if (sj & kWriteBack) indices[koff + (sj & ~kWriteBack)] = sh_w[sl];
      }
    }
  }
  phase_mark(scan, tile, 4);
}

template <bool KHOP2>
int launch_khop(const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                const uint32_t *d_num_input, size_t cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws,
                size_t ws_bytes, hipStream_t stream, fgnn_hashtable *fuse_ht = nullptr,
                ScanWsHost *scan_host = nullptr, const BatchStart *start = nullptr, bool resolve = false) {
  if (fanout == 0 || fanout > 0x7fffffffu) return FGNN_EINVAL;
  if (!d_num_input) cap = num_input;
  if (cap == 0) {
    if (d_num_out) FGNN_HIP_CHECK(hipMemsetAsync(d_num_out, 0, sizeof(size_t), stream));
    return FGNN_OK;
  }
  if (cap > 0xffffffffull) return FGNN_EINVAL;
  const uint32_t F = (uint32_t)fanout;
  const uint32_t tag = ((KHOP2 ? FGNN_KHOP2 : FGNN_KHOP0) << 8) | (layer & 0xffu);
  const size_t words_per_seed = (KHOP2 ? 3u : 1u) * (size_t)F;
  // seeds per workgroup: as many as fit in ~120 KiB of LDS (160 KiB per CU on gfx950)
  // small workgroups for small frontiers: the kernel is latency-bound, more resident waves hide more of it
  int S = 256;
  while (S > 64 && (words_per_seed * S * 4 > 120 * 1024 || cap / S < 2048)) S >>= 1;
  if (const int v = tune_int("FGNN_KHOP_S", 0))  // profiling build only
    if (v == 64 || v == 128 || v == 256) S = v;
  if (words_per_seed * S * 4 > 150 * 1024) return FGNN_EINVAL;  // fanout > ~200 (khop2) unsupported
  const size_t nb = div_up(cap, (size_t)S);
#ifdef FGNN_PROFILING
  const uint32_t ablate = (uint32_t)tune_int("FGNN_KHOP_ABLATE", 0);  // tools/khop_ablate.py; results are wrong when set
#endif
  FuseArgs fuse{HtView{nullptr, 0, 0, 1, 0, 0}, nullptr, nullptr, nullptr,
                BatchStart{nullptr, nullptr, nullptr, 0, 0, 0}};
  uint32_t *sums = static_cast<uint32_t *>(ws);
  if (fuse_ht) {
    // ws = pos[cap*F] (consumed by the dedup passes) | dedup sums | ... ; this kernel's offsets go at the very end
    if (ws_bytes < (cap * fanout + nb + 8) * sizeof(uint32_t) + fgnn_scratch_bytes(cap * fanout)) return FGNN_ENOSPC;
    if (cap * fanout > fuse_ht->max_fill_items) return FGNN_EINVAL;  // pending indices must fit the value field
    fuse.t = ht_view(fuse_ht);
    fuse.pos = static_cast<uint32_t *>(ws);
    fuse.d_num_items = fuse_ht->d_num_items;
    if (resolve) {
      if (!fuse_ht->disp) return FGNN_EINVAL;
      fuse.disp = fuse_ht->disp;
    }
    if (start) fuse.start = *start;
    sums = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + ws_bytes) - (nb + 4);
  } else if (ws_bytes < (nb + 1) * sizeof(uint32_t)) {
    return FGNN_ENOSPC;
  }
  const size_t lds = words_per_seed * S * sizeof(uint32_t);
  // khop0: rows beyond kSplitRow entries are drawn by the whole chip first (HubSplit), when the scratch has room for
  // the hub tables: they sit right below the sampler's block offsets (fused: behind the dedup's part of the scratch)
  HubSplit hub{nullptr, nullptr, nullptr, nullptr, 0};
  if (!KHOP2 && !(ablate & 16u)) {
    const size_t used_front = fuse_ht ? (cap * fanout + div_up(cap * fanout, (size_t)64) + 72) : 0;  // pos[] + dedup sums
    const size_t words_total = ws_bytes / sizeof(uint32_t);
    const size_t tail = fuse_ht ? nb + 4 : 0;            // fused: sampler sums at the very end
    const size_t head = fuse_ht ? used_front : nb + 8;   // plain: sampler sums at the start
    if (words_total > head + tail + cap + 16) {
      const size_t room = words_total - head - tail - cap - 16;
      size_t hubs = room / (F + 3);
      if (hubs > 1024) hubs = 1024;
      if (hubs >= 8) {
        uint32_t *base = static_cast<uint32_t *>(ws) + head;
        hub.of_seed = base;
        hub.count = base + cap;
        hub.list = base + cap + 8;
        hub.win = hub.list + 3 * hubs;
        hub.cap_hubs = (uint32_t)hubs;
        FGNN_HIP_CHECK(hipMemsetAsync(hub.count, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(khop0_hub_scan_kernel, dim3(div_up(cap, (size_t)256)), dim3(256), 0, stream, indptr, input,
                           num_input, d_num_input, cap, F, hub);
        hipLaunchKernelGGL(khop0_hub_kernel, dim3((unsigned)device_cu_count() * 4), dim3(256), 0, stream, hub, F, seed,
                           batch_key, tag);
      }
    }
  }
  ScanWs scan{nullptr, nullptr, 0, 0, nullptr, nullptr, 0, kScanHelpAfterPolls, nullptr};
  bool want_scan = scan_host && nb <= scan_host->ws.max_tiles;

#define FGNN_LAUNCH_KHOP2(SS, FM)                                                                              \
  do {                                                                                                         \
    static bool attr_done = false;                                                                             \
    if (!attr_done) {                                                                                          \
      FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&khop_sample_kernel<SS, 256, KHOP2, FM>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));             \
      attr_done = true;                                                                                        \
    }                                                                                                          \
    /* single pass (no count kernel, no scan kernel) for grids of up to 1536 workgroups (every tile sums all its    */ \
    /* predecessors' aggregates: O(tiles^2) descriptor reads); a waiter that outlasts its poll budget recomputes    */ \
    /* the missing aggregates itself, so residency and dispatch order are not assumed                              */ \
    if (want_scan && nb <= 1536) scan = scan_host->next(0, nb);                                            \
    if (!scan.desc) {                                                                                          \
      scan.log = phase_log_base();                                                                             \
      hipLaunchKernelGGL((khop_count_kernel_s<SS>), dim3(nb), dim3(SS), 0, stream, indptr, input, num_input,   \
                         d_num_input, cap, F, sums);                                                           \
      if (launch_scan_block_sums(sums, nb, d_num_out, nullptr, nullptr, nullptr, stream, d_num_input, SS) !=   \
          FGNN_OK)                                                                                             \
        return FGNN_EHIP;                                                                                      \
    }                                                                                                          \
    hipLaunchKernelGGL((khop_sample_kernel<SS, 256, KHOP2, FM>), dim3(nb), dim3(256), lds, stream, indptr, indices, \
                       input, num_input, d_num_input, cap, F, sums, out_src, out_dst, src_mode, seed,          \
                       batch_key, tag, fuse, scan, d_num_out, hub FGNN_ABLATE_ARG(ablate));                    \
  } while (0)
#define FGNN_LAUNCH_KHOP(SS)                                                                                   \
  do {                                                                                                         \
    if (!KHOP2) FGNN_LAUNCH_KHOP2(SS, 0);                                                                      \
    else if (F <= 8) FGNN_LAUNCH_KHOP2(SS, 8);                                                                 \
    else if (F <= 16) FGNN_LAUNCH_KHOP2(SS, 16);                                                               \
    else if (F <= 32) FGNN_LAUNCH_KHOP2(SS, 32);                                                               \
    else FGNN_LAUNCH_KHOP2(SS, 0);                                                                             \
  } while (0)

  if (S == 256) FGNN_LAUNCH_KHOP(256);
  else if (S == 128) FGNN_LAUNCH_KHOP(128);
  else FGNN_LAUNCH_KHOP(64);
#undef FGNN_LAUNCH_KHOP
#undef FGNN_LAUNCH_KHOP2
  return launch_status(__func__);
}

}  // namespace

int sample_khop_fused(bool khop2, const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                      const uint32_t *d_num_input, size_t cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                      size_t *d_num_out, uint64_t seed, uint64_t batch_key, uint32_t layer, fgnn_hashtable *ht,
                      void *ws, size_t ws_bytes, void *stream, ScanWsHost *scan, const BatchStart *start, bool resolve) {
  if (!ht) return FGNN_EINVAL;
  if (start && (d_num_input || !start->n2o)) return FGNN_EINVAL;  // the first launch takes the seeds with a host count
  auto st = static_cast<hipStream_t>(stream);
  return khop2 ? launch_khop<true>(indptr, indices, input, num_input, d_num_input, cap, fanout, out_src, out_dst,
                                   d_num_out, FGNN_SRC_LOCAL, seed, batch_key, layer, ws, ws_bytes, st, ht, scan, start,
                                   resolve)
               : launch_khop<false>(indptr, indices, input, num_input, d_num_input, cap, fanout, out_src, out_dst,
                                    d_num_out, FGNN_SRC_LOCAL, seed, batch_key, layer, ws, ws_bytes, st, ht, scan, start,
                                    resolve);
}

// the same sampler without the fused insert, but with the slot's look-back descriptors (one launch): the engine's
// split last layer -- khop2's CSR-order chain then ends with this kernel, the dedup insert runs behind it
int sample_khop_plain(bool khop2, const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                      const uint32_t *d_num_input, size_t cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                      size_t *d_num_out, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream, ScanWsHost *scan) {
  auto st = static_cast<hipStream_t>(stream);
  return khop2 ? launch_khop<true>(indptr, indices, input, num_input, d_num_input, cap, fanout, out_src, out_dst,
                                   d_num_out, FGNN_SRC_LOCAL, seed, batch_key, layer, ws, ws_bytes, st, nullptr, scan)
               : launch_khop<false>(indptr, indices, input, num_input, d_num_input, cap, fanout, out_src, out_dst,
                                    d_num_out, FGNN_SRC_LOCAL, seed, batch_key, layer, ws, ws_bytes, st, nullptr, scan);
}

}  // namespace fgnn

extern "C" int fgnn_sample_khop0(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                 size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                 uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed,
                                 uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes, void *stream) {
  return fgnn::launch_khop<false>(indptr, const_cast<uint32_t *>(indices), input, num_input, d_num_input,
                                  num_input_cap, fanout, out_src, out_dst, d_num_out, src_mode, seed, batch_key, layer,
                                  ws, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int fgnn_sample_khop2(const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                                 const uint32_t *d_num_input, size_t num_input_cap, size_t fanout, uint32_t *out_src,
                                 uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                                 uint32_t layer, void *ws, size_t ws_bytes, void *stream) {
  return fgnn::launch_khop<true>(indptr, indices, input, num_input, d_num_input, num_input_cap, fanout, out_src,
                                 out_dst, d_num_out, src_mode, seed, batch_key, layer, ws, ws_bytes,
                                 static_cast<hipStream_t>(stream));
}
