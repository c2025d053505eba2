// hashtable_partition.hip -- the batch's LAST dedup fill, hash-partitioned and deduplicated in LDS.
//
// Replaces pass 1 of FillWithDuplicates (generate_hashmap_duplicates, reference
// samgraph/common/cuda/cuda_hashtable.cu:176-211: one probe + CAS into the global table per edge) for the fill after
// which nothing looks the table up again -- layer 0, 96 % of a GraphSAGE [25,10] batch's edges.  Same outcome per item
// as ht_insert_resolve (fgnn_device.h): the key's final value -- pend|i (item i is the first occurrence), pend|j (a
// duplicate of the earlier item j), or the local id a node got in an earlier layer.
//
// Why: the global table is 64 MiB, an edge's probe is a random 8-byte access that moves a 64-byte line, a new key's CAS
// another (memory-side): 46.7 MB of fabric traffic for 5.9 MB of algorithmic bytes on the papers100M shape
// (profiles/r03_c_pmc_traffic.json), at the chip's random-request rate.  Duplicates are global (hub nodes reached from
// everywhere: only 0.26 % of a layer's edges repeat inside the tile a sampler workgroup emits,
// profiles/r02_b_dup_locality.txt), so a per-workgroup LDS table in front of the global one merges nothing.  What
// works is making the table itself local: keys are split by the top bits of their hash into bins of 1-2 K items, a
// bin's items are brought together by one streaming pass, and ONE workgroup dedups a bin in a 64 KiB LDS table -- the
// per-block hash buckets are the whole table, the global one is not touched at all:
//   part_hist_kernel     items -> counts[workgroup][bin]      (LDS histogram per workgroup; plain stores of its row)
//   part_scatter_kernel  items -> (key, value) pairs by bin   (every workgroup sums the matrix's columns itself: bin
//                                                               bases and its own runs, no atomics, no cursor, a
//                                                               layout that is a pure function of the input)
//   part_dedup_kernel    bin -> LDS table (64-bit CAS / min, as the global protocol) -> outcome of every item
// The nodes the table already knows (seeds and earlier layers' nodes: n2o[0 .. num_items)) go through the same
// partition with value = their local id, so "already known" needs no global lookup either.  The table's `Reset` is a
// generation bump anyway; this fill leaves the global table untouched.
// Traffic: items read twice (2 x 1.5 MB), pairs written and read (2 x 3 MB), every item marked "first occurrence" in
// item order by the scatter pass (1.5 MB, coalesced) and an outcome scattered only to the quarter of the items that are
// not -- against 371 K random probes + 290 K memory-side CAS.  PMC per papers100M batch: 1.0 + 6.8 + 4.1 MB for the three
// launches against 46.7 MB for the insert (profiles/r04_e_pmc_traffic.json); interleaved A/B: sampler-side stage 0.0899 ->
// 0.0690 ms per batch (profiles/r04_a_ab_partition_vs_global.txt).
//
// The same holds for EVERY fill of a batch whose samplers do not insert into the table themselves (weighted, random
// walk, khop1, ...: `table_free`, hashtable_fill_duplicates_ex): all their fills are partitioned, the table is never
// read.
//
// Bins whose distinct keys do not fit the LDS table (a graph whose ids collide in the hash's top bits; never seen on
// the R-MAT or power-law shapes) fall back, per bin, to the global table with the ordinary probes: a key's items all
// sit in one bin, i.e. in one workgroup, so a workgroup barrier orders "all inserted" before "read back".
#include <atomic>

#include "fgnn_device.h"

namespace fgnn {
namespace {

constexpr int kPartThreads = 1024;                // hist / scatter workgroup: 16 waves, 4 items per thread and tile
constexpr int kPartIPT = 4;
constexpr int kPartTile = kPartThreads * kPartIPT;  // 4096 items per tile
constexpr int kPartMaxBlocks = 128;               // hist / scatter workgroups (rows of the count matrix)
constexpr int kPartDedupThreads = 512;
constexpr int kPartRegPairs = 8;                  // pairs a dedup lane keeps in registers: bins up to 4096 items
constexpr uint32_t kPartLdsSlots = 8192;          // 64 KiB of 8-byte buckets: two workgroups per CU
constexpr size_t kPartLdsBytes = kPartLdsSlots * sizeof(unsigned long long) + 16;  // + two counters
// items per bin aimed at (the bin count is the next power of two: bins hold 1024 .. 2048 items, ~0.8 of them distinct:
// LDS load <= 0.2).  2048 rather than 1536: a papers100M batch (~393 K items) then always gets 256 bins; at 1536 the
// batches straddled the 256 / 512 boundary and the 512-bin ones cost the whole path 1 % (two builds alternating)
constexpr uint32_t kPartPerBin = 2048;
constexpr uint32_t kPartMinLog2 = 5, kPartMaxLog2 = 11;
constexpr uint32_t kPartHash = 0x9E3779B1u;

struct PartView {
  uint2 *pairs;          // [max_fill_items + max_items] (key, value) grouped by bin
  uint32_t *counts;      // [kPartMaxBlocks][max_bins] items of bin b in workgroup k's tiles (written by the histogram)
  uint32_t *bin_base;    // [max_bins + 1]  written by the scatter kernel
  uint32_t max_log2;     // log2(max_bins) <= 11
  uint32_t lds_limit;    // distinct keys a bin may hold in LDS before it falls back to the global table
};

// bins for `total` items: a power of two, the same in all three kernels (they derive it from the same device counts)
__device__ __forceinline__ uint32_t part_log2(uint32_t total, uint32_t max_log2) {
  const uint32_t want = (total + kPartPerBin - 1) / kPartPerBin;
  uint32_t lg = want <= 1u ? 0u : 32u - (uint32_t)__clz(want - 1u);
  lg = lg < kPartMinLog2 ? kPartMinLog2 : lg;
  return lg > max_log2 ? max_log2 : lg;
}
// workgroups of the histogram / scatter launches that have a tile (both launches walk the tiles with the same stride,
// so workgroup k sees the same items in both)
__device__ __forceinline__ uint32_t part_blocks(uint32_t total, uint32_t launched) {
  const uint32_t tiles = (total + kPartTile - 1) / kPartTile;
  return tiles < launched ? tiles : launched;
}

// item t of the fill's flat item space: the K known nodes first (value = local id), then the n new items
// (value = pend | item index)
__device__ __forceinline__ void part_item(uint32_t t, uint32_t K, const uint32_t *__restrict__ n2o,
                                          const uint32_t *__restrict__ items, uint32_t pend, uint32_t *key,
                                          uint32_t *val) {
  if (t < K) {
    *key = n2o[t];
    *val = t;
  } else {
    *key = items[t - K];
    *val = pend | (t - K);
  }
}

// counts[k][b] = items of bin b among workgroup k's tiles.  Plain stores of the whole row (zeros included): nothing to
// clear between fills, no atomics, and the scatter's offsets are a pure function of the matrix (deterministic layout).
__global__ __launch_bounds__(kPartThreads) void part_hist_kernel(const uint32_t *__restrict__ items, size_t n_host,
                                                                 const size_t *d_n, size_t cap,
                                                                 const uint32_t *__restrict__ n2o,
                                                                 uint32_t *d_num_items, uint32_t pend, PartView p,
                                                                 uint32_t own_blocks, FixTail fix) {
  if (blockIdx.x >= own_blocks) {  // an earlier fill's remap fix-up riding along (FixTail, fgnn_device.h)
    run_fix_tail(fix, own_blocks);
    return;
  }
  __shared__ uint32_t hist[1u << kPartMaxLog2];
  const uint32_t n = (uint32_t)resolve_count64(n_host, d_n, cap);
  const uint32_t K = d_num_items[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) d_num_items[1] = K;  // count before this fill (pass 1's duty)
  const uint32_t total = n + K;
  const uint32_t lg = part_log2(total, p.max_log2), B = 1u << lg;
  if (blockIdx.x >= part_blocks(total, own_blocks)) return;  // no tile: the scatter does not read this row
  for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) hist[b] = 0;
  __syncthreads();
  for (uint32_t t0 = blockIdx.x * kPartTile; t0 < total; t0 += own_blocks * kPartTile) {
    uint32_t key[kPartIPT], val;
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r) {
      const uint32_t t = t0 + r * kPartThreads + threadIdx.x;
      key[r] = FGNN_EMPTY_KEY;
      if (t < total) part_item(t, K, n2o, items, pend, &key[r], &val);
    }
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r)
      if (t0 + r * kPartThreads + threadIdx.x < total) atomicAdd(&hist[(key[r] * kPartHash) >> (32 - lg)], 1u);
  }
  __syncthreads();
  uint32_t *row = p.counts + ((size_t)blockIdx.x << lg);
  for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) row[b] = hist[b];
}

__global__ __launch_bounds__(kPartThreads) void part_scatter_kernel(const uint32_t *__restrict__ items, size_t n_host,
                                                                    const size_t *d_n, size_t cap,
                                                                    const uint32_t *__restrict__ n2o,
                                                                    const uint32_t *d_num_items, uint32_t pend,
                                                                    PartView p, uint32_t *__restrict__ pos) {
  constexpr int NW = kPartThreads / kWave;
  __shared__ uint32_t off[1u << kPartMaxLog2], cnt[1u << kPartMaxLog2], lbase[1u << kPartMaxLog2];
  __shared__ uint2 stage[kPartTile];
  __shared__ uint32_t sh[NW];
  const uint32_t n = (uint32_t)resolve_count64(n_host, d_n, cap);
  const uint32_t K = d_num_items[1];  // == [0] until the count+assign pass; [1] was written by the histogram launch
  const uint32_t total = n + K;
  const uint32_t lg = part_log2(total, p.max_log2), B = 1u << lg;
  const uint32_t nblk = part_blocks(total, gridDim.x);
  if (blockIdx.x >= nblk) return;
  // the first tile's items: their loads go out together with the matrix loads below (one memory round trip)
  uint32_t key[kPartIPT], val[kPartIPT];
  uint32_t t0 = blockIdx.x * kPartTile;
#pragma unroll
  for (int r = 0; r < kPartIPT; ++r) {
    const uint32_t t = t0 + r * kPartThreads + threadIdx.x;
    key[r] = FGNN_EMPTY_KEY;
    val[r] = 0;
    if (t < total) part_item(t, K, n2o, items, pend, &key[r], &val[r]);
    // most items turn out to be first occurrences (76 % on the papers100M shape): every item is marked as one here,
    // in item order (coalesced), and the dedup kernel scatters an outcome only to those that are not
    if (t < total && t >= K) pos[t - K] = kPartIsOwner;
  }
  // where this workgroup's run of bin b starts = items of bins below b (all workgroups) + items of bin b in the
  // workgroups before this one: column sums over the count matrix (nblk rows of B cells).  A lane takes one bin (two
  // when B = 2048) and every G-th row, G = lanes per bin: its loads are independent, a wave reads 64 consecutive cells
  // of a row, the partial sums stay in registers and meet in LDS with one add per lane.
  for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
    off[b] = 0;  // bin b in the workgroups before this one
    cnt[b] = 0;  // bin b in all workgroups
  }
  __syncthreads();
  {
    const uint32_t bc = B < (uint32_t)kPartThreads ? B : (uint32_t)kPartThreads;  // bins side by side
    const uint32_t G = kPartThreads / bc, g = threadIdx.x / bc, b0 = threadIdx.x % bc;
    for (uint32_t b = b0; b < B; b += kPartThreads) {  // one trip, two when B = 2048
      uint32_t all = 0, before = 0;
      for (uint32_t k0 = g; k0 < nblk; k0 += 8 * G) {
        uint32_t c[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
          const uint32_t k = k0 + u * G;
          c[u] = k < nblk ? p.counts[((size_t)k << lg) + b] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
          all += c[u];
          before += (k0 + u * G < blockIdx.x) ? c[u] : 0u;
        }
      }
      if (all) atomicAdd(&cnt[b], all);
      if (before) atomicAdd(&off[b], before);
    }
    __syncthreads();
    // thread j owns bins j*per .. j*per+per-1 (consecutive, so that the block scan of the per-thread totals gives the
    // bin bases)
    const uint32_t per = (B + kPartThreads - 1) / kPartThreads;  // 1 or 2
    uint32_t tot_b[2] = {0, 0};
#pragma unroll
    for (uint32_t q = 0; q < 2; ++q) {
      const uint32_t b = threadIdx.x * per + q;
      if (q < per && b < B) tot_b[q] = cnt[b];
    }
    uint32_t tot;
    uint32_t run = block_exclusive_scan<NW>(tot_b[0] + tot_b[1], sh, &tot);
#pragma unroll
    for (uint32_t q = 0; q < 2; ++q) {
      const uint32_t b = threadIdx.x * per + q;
      if (q < per && b < B) {
        off[b] += run;
        if (blockIdx.x == 0) p.bin_base[b] = run;
        run += tot_b[q];
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.bin_base[B] = tot;
  }
  // Per tile: rank every item inside its bin (LDS add), order the tile by bin in an LDS stage, write it out in that
  // order -- consecutive lanes then store consecutive pairs of a bin's run (a wave covers a handful of runs) instead of
  // 64 scattered 8-byte stores per wave instruction.
  const uint32_t per = (B + kPartThreads - 1) / kPartThreads;
  for (;;) {
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) cnt[b] = 0;
    __syncthreads();  // off[] complete (first round) / advanced (later rounds), cnt[] zero
    uint32_t where[kPartIPT];  // bin << 16 | rank inside the tile's run of the bin
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r) {
      where[r] = 0;
      if (t0 + r * kPartThreads + threadIdx.x < total) {
        const uint32_t b = (key[r] * kPartHash) >> (32 - lg);
        where[r] = (b << 16) | atomicAdd(&cnt[b], 1u);  // rank < 4096
      }
    }
    __syncthreads();
    {  // lbase[b] = items of the tile in bins below b
      uint32_t c2[2] = {0, 0};
#pragma unroll
      for (uint32_t q = 0; q < 2; ++q) {
        const uint32_t b = threadIdx.x * per + q;
        if (q < per && b < B) c2[q] = cnt[b];
      }
      uint32_t tot;
      uint32_t run = block_exclusive_scan<NW>(c2[0] + c2[1], sh, &tot);
#pragma unroll
      for (uint32_t q = 0; q < 2; ++q) {
        const uint32_t b = threadIdx.x * per + q;
        if (q < per && b < B) {
          lbase[b] = run;
          run += c2[q];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r)
      if (t0 + r * kPartThreads + threadIdx.x < total)
        stage[lbase[where[r] >> 16] + (where[r] & 0xffffu)] = make_uint2(key[r], val[r]);
    __syncthreads();
    const uint32_t in_tile = total - t0 < (uint32_t)kPartTile ? total - t0 : (uint32_t)kPartTile;
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r) {
      const uint32_t i = r * kPartThreads + threadIdx.x;
      if (i < in_tile) {
        const uint2 kv = stage[i];
        const uint32_t b = (kv.x * kPartHash) >> (32 - lg);
        p.pairs[off[b] + (i - lbase[b])] = kv;
      }
    }
    t0 += gridDim.x * kPartTile;
    if (t0 >= total) break;  // uniform
    __syncthreads();  // every off[] / lbase[] / stage[] read done
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) off[b] += cnt[b];
#pragma unroll
    for (int r = 0; r < kPartIPT; ++r) {
      const uint32_t t = t0 + r * kPartThreads + threadIdx.x;
      key[r] = FGNN_EMPTY_KEY;
      val[r] = 0;
      if (t < total) part_item(t, K, n2o, items, pend, &key[r], &val[r]);
      if (t < total && t >= K) pos[t - K] = kPartIsOwner;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t part_slot(uint32_t key, uint32_t lg) {
  // the hash bits right below the bin's
  return ((key * kPartHash) << lg) >> (32 - 13);
}

// minimum value per key, the global protocol (ht_insert_min) on LDS atomics; true if the key was new
__device__ __forceinline__ bool part_lds_insert(unsigned long long *tab, uint32_t lg, uint2 kv, uint32_t *overflow) {
  const unsigned long long mine = ((unsigned long long)kv.x << 32) | kv.y;
  uint32_t h = part_slot(kv.x, lg);
  for (uint32_t probes = 0; probes < kPartLdsSlots; ++probes) {
    unsigned long long cur = tab[h];
    if (cur == kEmpty64) {
      cur = atomicCAS(&tab[h], kEmpty64, mine);
      if (cur == kEmpty64) return true;
    }
    if ((uint32_t)(cur >> 32) == kv.x) {
      if ((uint32_t)cur > kv.y) atomicMin(&tab[h], mine);
      return false;
    }
    h = (h + 1) & (kPartLdsSlots - 1);
    // somebody has found the table full: the bin goes through the global table whatever this item finds -- do not walk
    // the other 8 K slots for every remaining item (a bin of 100 K items would cost ~10^9 LDS probes before it fell back)
    if ((probes & 31u) == 31u && *(volatile uint32_t *)overflow) return false;
  }
  *overflow = 1u;  // table full (benign race: every writer stores 1)
  return false;
}
__device__ __forceinline__ uint32_t part_lds_find(const unsigned long long *tab, uint32_t lg, uint32_t key) {
  uint32_t h = part_slot(key, lg);
  unsigned long long cur = tab[h];
  while ((uint32_t)(cur >> 32) != key) {  // present by construction
    h = (h + 1) & (kPartLdsSlots - 1);
    cur = tab[h];
  }
  return (uint32_t)cur;
}

__global__ __launch_bounds__(kPartDedupThreads) void part_dedup_kernel(PartView p, HtView t, size_t n_host,
                                                                       const size_t *d_n, size_t cap,
                                                                       const uint32_t *d_num_items,
                                                                       uint32_t *__restrict__ pos) {
  static_assert(kPartLdsSlots == (1u << 13), "part_slot takes 13 bits");
  extern __shared__ unsigned long long part_lds[];  // kPartLdsBytes: the table, then two counters
  unsigned long long *tab = part_lds;
  uint32_t &sh_distinct = reinterpret_cast<uint32_t *>(part_lds + kPartLdsSlots)[0];
  uint32_t &sh_overflow = reinterpret_cast<uint32_t *>(part_lds + kPartLdsSlots)[1];
  const uint32_t n = (uint32_t)resolve_count64(n_host, d_n, cap);
  const uint32_t total = n + d_num_items[1];
  const uint32_t lg = part_log2(total, p.max_log2), B = 1u << lg;
  for (uint32_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint32_t beg = p.bin_base[b], cnt = p.bin_base[b + 1] - beg;
    // bins of up to kPartRegPairs x 512 items (all of them unless a hub node piles its duplicates into one): a lane's
    // pairs stay in registers between the insert and the read-back
    const bool in_regs = cnt <= (uint32_t)kPartRegPairs * kPartDedupThreads;
    uint2 kv[kPartRegPairs];
    if (in_regs) {
#pragma unroll
      for (int r = 0; r < kPartRegPairs; ++r) {
        const uint32_t q = r * kPartDedupThreads + threadIdx.x;
        kv[r] = q < cnt ? p.pairs[beg + q] : make_uint2(FGNN_EMPTY_KEY, 0u);
      }
    }
    for (uint32_t q = threadIdx.x; q < kPartLdsSlots; q += kPartDedupThreads) tab[q] = kEmpty64;
    if (threadIdx.x == 0) {
      sh_distinct = 0;
      sh_overflow = 0;
    }
    __syncthreads();
    uint32_t newkeys = 0;
    // more items than four tables' worth: a bin that full (fills beyond 2048 bins x 2048 items, or a run of ids that
    // collide in the hash's top bits) almost surely overflows -- straight to the global table, no LDS pass
    const bool hopeless = cnt > 4u * kPartLdsSlots;
    if (hopeless) {
      if (threadIdx.x == 0) sh_overflow = 1u;
    } else if (in_regs) {
#pragma unroll
      for (int r = 0; r < kPartRegPairs; ++r)
        if (r * kPartDedupThreads + threadIdx.x < cnt) newkeys += part_lds_insert(tab, lg, kv[r], &sh_overflow) ? 1u : 0u;
    } else {
      for (uint32_t q = threadIdx.x; q < cnt && !*(volatile uint32_t *)&sh_overflow; q += kPartDedupThreads)
        newkeys += part_lds_insert(tab, lg, p.pairs[beg + q], &sh_overflow) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) newkeys += __shfl_xor(newkeys, d, kWave);
    if (lane_id() == 0 && newkeys) atomicAdd(&sh_distinct, newkeys);
    __syncthreads();
    const bool lds_ok = sh_overflow == 0 && sh_distinct <= p.lds_limit;
    if (lds_ok) {
      if (in_regs) {
#pragma unroll
        for (int r = 0; r < kPartRegPairs; ++r)
          if (r * kPartDedupThreads + threadIdx.x < cnt && (kv[r].y & t.pend)) {  // (a known node: nobody asks for its outcome)
            const uint32_t v = part_lds_find(tab, lg, kv[r].x);
            if (v != kv[r].y) pos[kv[r].y & (t.pend - 1u)] = v;  // first occurrences keep the scatter kernel's mark
          }
      } else {
        for (uint32_t q = threadIdx.x; q < cnt; q += kPartDedupThreads) {
          const uint2 x = p.pairs[beg + q];
          if (x.y & t.pend) {
            const uint32_t v = part_lds_find(tab, lg, x.x);
            if (v != x.y) pos[x.y & (t.pend - 1u)] = v;
          }
        }
      }
    } else {
      // this bin's keys through the global table (generation-tagged, ht_insert_min), the known nodes with their local
      // ids too: a batch whose fills are all partitioned has never put them there (and what an earlier fill's
      // fall-back left pending in the table loses against the local id: min)
      for (uint32_t q = threadIdx.x; q < cnt; q += kPartDedupThreads) {
        const uint2 x = p.pairs[beg + q];
        (void)ht_insert_min(t, x.x, x.y);
      }
      __threadfence();
      __syncthreads();  // every item of these keys is this workgroup's: all inserted before any is read back
      for (uint32_t q = threadIdx.x; q < cnt; q += kPartDedupThreads) {
        const uint2 x = p.pairs[beg + q];
        if (!(x.y & t.pend)) continue;
        uint32_t h = hash_slot(x.x, t.shift, t.mask), v = FGNN_EMPTY_KEY;
        for (uint32_t probes = 0; probes <= t.mask; ++probes) {
          const unsigned long long cur = __hip_atomic_load(&t.table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!ht_live(t, cur)) break;
          if ((uint32_t)(cur >> 32) == x.x) {
            v = ht_value(t, cur);
            break;
          }
          h = (h + 1) & t.mask;
        }
        if (v != x.y) pos[x.y & (t.pend - 1u)] = v;
      }
    }
    __syncthreads();  // the table is wiped for the next bin
  }
}

std::atomic<int> g_part_lds_limit{-1};
constexpr int kMaxAttrDevices = 64;

}  // namespace

struct PartWs {
  uint2 *pairs = nullptr;
  uint32_t *bins = nullptr;  // counts[kPartMaxBlocks][max_bins] | base[max_bins + 1]
  uint32_t max_log2 = 0;
  size_t max_bins = 0;
  size_t pairs_cap = 0;
};

PartWs *partition_create(size_t max_items, size_t max_fill_items) {
  auto *w = new PartWs();
  w->pairs_cap = max_items + max_fill_items;
  uint32_t lg = kPartMinLog2;
  while (lg < kPartMaxLog2 && ((size_t)kPartPerBin << lg) < w->pairs_cap) ++lg;
  w->max_log2 = lg;
  w->max_bins = (size_t)1 << lg;
  const size_t bin_bytes = ((size_t)kPartMaxBlocks * w->max_bins + w->max_bins + 1) * sizeof(uint32_t);
  if (hipMalloc(&w->pairs, w->pairs_cap * sizeof(uint2)) != hipSuccess ||
      hipMalloc(&w->bins, bin_bytes) != hipSuccess || hipMemset(w->bins, 0, bin_bytes) != hipSuccess) {
    partition_destroy(w);
    (void)hipGetLastError();
    return nullptr;
  }
  return w;
}

void partition_destroy(PartWs *w) {
  if (!w) return;
  if (w->pairs) (void)hipFree(w->pairs);
  if (w->bins) (void)hipFree(w->bins);
  delete w;
}

bool partition_fits(const PartWs *w, const fgnn_hashtable *ht, size_t cap) {
  return w && ht && cap > 0 && cap + ht->max_items <= w->pairs_cap && cap + ht->max_items < 0x7fffffffull;
}

int partition_fill(PartWs *w, const fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                   const size_t *d_num_items, size_t cap, uint32_t *pos, hipStream_t s, const FixTail &carry) {
  const HtView tv = ht_view(ht);
  PartView p{w->pairs, w->bins, w->bins + (size_t)kPartMaxBlocks * w->max_bins, w->max_log2,
             g_part_lds_limit.load(std::memory_order_relaxed) >= 0
                 ? (uint32_t)g_part_lds_limit.load(std::memory_order_relaxed) : kPartLdsSlots * 3u / 4u};
  // tiles are walked with a stride: the count matrix has one row per workgroup
  const size_t tiles = div_up(cap + ht->max_items, (size_t)kPartTile);
  const size_t blocks = tiles < (size_t)kPartMaxBlocks ? tiles : (size_t)kPartMaxBlocks;
  const size_t bins_grid = w->max_bins < (size_t)device_cu_count() * 2 ? w->max_bins : (size_t)device_cu_count() * 2;
  // (the fix-up tail's workgroups run with this launch's 1024 threads: run_fix_tail strides by blockDim)
  hipLaunchKernelGGL(part_hist_kernel, dim3((unsigned)(blocks + carry.blocks)), dim3(kPartThreads), 0, s, items,
                     num_items, d_num_items, cap, ht->n2o, ht->d_num_items, tv.pend, p, (uint32_t)blocks, carry);
  hipLaunchKernelGGL(part_scatter_kernel, dim3((unsigned)blocks), dim3(kPartThreads), 0, s, items, num_items,
                     d_num_items, cap, ht->n2o, ht->d_num_items, tv.pend, p, pos);
  // more than the 64 KiB a kernel gets without asking (160 KiB per CU on gfx950).  The attribute belongs to the
  // function ON A DEVICE: asked once per device (a process that samples on a second GPU launches there too)
  static std::atomic<bool> attr_done[kMaxAttrDevices];
  int dev = 0;
  FGNN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxAttrDevices || !attr_done[dev].load(std::memory_order_acquire)) {
    FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&part_dedup_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartLdsBytes));
    if (dev >= 0 && dev < kMaxAttrDevices) attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(part_dedup_kernel, dim3((unsigned)bins_grid), dim3(kPartDedupThreads), kPartLdsBytes, s, p, tv,
                     num_items, d_num_items, cap, ht->d_num_items, pos);
  return launch_status(__func__);
}

}  // namespace fgnn

extern "C" void fgnn_debug_set_partition_lds_limit(int distinct_keys) {
  fgnn::g_part_lds_limit.store(distinct_keys, std::memory_order_relaxed);
}
