// hashtable.hip -- frontier dedup / compaction / remap for gfx950.
//
// Replaces OrderedHashTable (reference samgraph/common/cuda/cuda_hashtable.{h,cu}) and the
// dst half of GPUMapEdges (cuda_mapping.cu:31-81).  Observable behaviour == oracle
// fgnn_oracle_ht_* : new nodes get local ids in order of FIRST occurrence (a legal outcome of the
// reference's CAS race, and what its CPU twin CPUHashTable2 produces with one thread).
//
// MI355X design
//  * bucket = ONE 64-bit word {key:hi32, value:lo32} instead of the reference's 16-byte
//    {key,local,index,version}: a whole insert is a single 8-byte CAS, duplicates resolve to the
//    minimum index with one 8-byte atomicMin (the key half is identical, so u64 min == value min);
//    half the table footprint keeps the table inside the 256 MiB Infinity Cache;
//  * value = [generation | pending flag | index]: pending|i while item i is the pending first occurrence of a new
//    key, a local id once assigned.  A bucket whose generation is not the table's current one is EMPTY, so
//    Reset() (cuda_hashtable.cu:714-723, a 137 MB memset per batch in the reference) is a generation bump; the
//    table is really wiped only when the generation counter wraps (every 2^(32 - index bits - 1) - 1 batches:
//    511 for GraphSAGE [25,10] at batch 8000).  A free bucket is claimed by CAS(expected = the word just read);
//  * device-scope atomics on CDNA4 execute at the memory side (~20 G scattered atomics/s chip-wide),
//    so the insert first READS the bucket and only issues an atomic when it can change something:
//    a key that is already present with a smaller value costs one 8-byte load, no atomic.
//    Stale reads (per-XCD L2s are not coherent inside a launch) can only show an older state, which
//    at worst causes a redundant atomic, never a wrong decision;
//  * the insert remembers each item's bucket in `pos[]`, so the later passes (owner flag, local-id
//    assignment, remap) read the bucket directly instead of re-probing (the reference re-probes in
//    count_hashmap, compact_hashmap and map_edge_ids);
//  * owner ranks come from wave ballots + a one-workgroup scan of per-workgroup counts.
#include <cstdlib>

#include "fgnn_device.h"


namespace fgnn {
namespace {

__device__ __forceinline__ uint32_t ht_find(const HtView &t, uint32_t id, uint32_t *bucket) {
  uint32_t h = hash_slot(id, t.shift, t.mask);
  // the key is present by contract; bound the probe anyway so a violated contract cannot hang the GPU
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    const unsigned long long cur = t.table[h];
    if (!ht_live(t, cur)) break;
    if ((uint32_t)(cur >> 32) == id) { *bucket = h; return ht_value(t, cur); }
    h = (h + 1) & t.mask;
  }
  *bucket = 0;
  return FGNN_EMPTY_KEY;
}

// FillWithUnique: item i -> local id base + i
__global__ __launch_bounds__(kBlock) void ht_fill_unique_kernel(HtView t, const uint32_t *__restrict__ items,
                                                                size_t n, uint32_t *__restrict__ n2o,
                                                                uint32_t *d_num_items, size_t max_items) {
  const uint32_t base = d_num_items[0];
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && base + i < max_items) {
    const uint32_t id = items[i];
    (void)ht_insert_min(t, id, (uint32_t)(base + i));
    n2o[base + i] = id;
  }
}

// first fill of a batch on a freshly reset table: item i -> local id i; also copies the items (the
// batch's output_nodes), initialises the batch summary and sets the item count -- one launch instead
// of memset + fill + advance + copy
__global__ __launch_bounds__(kBlock) void ht_start_batch_kernel(HtView t, const uint32_t *__restrict__ items,
                                                                size_t n, uint32_t *__restrict__ n2o,
                                                                uint32_t *__restrict__ items_copy,
                                                                uint32_t *d_num_items, fgnn_batch_meta *meta,
                                                                uint64_t key, uint32_t num_layers) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) {
    d_num_items[0] = (uint32_t)n;
    d_num_items[1] = 0;
    if (meta) {
      fgnn_batch_meta m;
      memset(&m, 0, sizeof(m));
      m.key = key;
      m.num_layers = num_layers;
      m.num_output = (uint32_t)n;
      m.num_input = (uint32_t)n;
      m.t_start = wall_clock64();
      *meta = m;
    }
  }
  if (i < n) {
    const uint32_t id = items[i];
    (void)ht_insert_min(t, id, (uint32_t)i);
    n2o[i] = id;
    if (items_copy) items_copy[i] = id;
  }
}

// The real wipe behind Reset (only when the generation counter wraps): all buckets to the never-used pattern,
// counts zero.  Non-temporal 16-byte stores: the wiped
// table is not read again before the next batch has gone through the whole chain, and 64 MiB of ordinary stores
// would sit as dirty lines in the Infinity Cache and be evicted by -- i.e. slow down -- the random reads of the
// kernels that follow (measured: 500 K cold 4-byte reads take 21 us instead of 8 us right after a large write).
__global__ __launch_bounds__(kBlock) void ht_wipe_kernel(unsigned long long *table, size_t capacity,
                                                         uint32_t *d_num_items) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 *t = reinterpret_cast<v4 *>(table);
  const size_t n16 = capacity / 2;  // capacity is a power of two >= 1024
  const v4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock)
    __builtin_nontemporal_store(ones, &t[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d_num_items[0] = 0;
    d_num_items[1] = 0;
  }
}

__global__ void ht_advance_kernel(uint32_t *d_num_items, uint32_t add) { d_num_items[0] += add; }

// pass 1: insert every item with value PENDING|i; remember its bucket.  One item per lane: issuing a lane's inserts four
// at a time (ht_insert_*_batch) was measured on the layer-0 fill of the papers100M shape and changes nothing -- 371 K
// probe reads + 290 K CAS at the chip's random-access rate ARE the kernel's 27-30 us
template <int IPT>
__global__ __launch_bounds__(kBlock) void ht_insert_kernel(HtView t, const uint32_t *__restrict__ items, size_t n_host,
                                                           const size_t *d_n, size_t cap,
                                                           uint32_t *__restrict__ pos, uint32_t *d_num_items,
                                                           uint32_t *disp, uint32_t own_blocks, FixTail fix) {
  if (blockIdx.x >= own_blocks) {  // an earlier fill's remap fix-up riding along (FixTail, fgnn_device.h)
    run_fix_tail(fix, own_blocks);
    return;
  }
  const size_t n = resolve_count64(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
  if (blockIdx.x == 0 && threadIdx.x == 0) d_num_items[1] = d_num_items[0];  // count before this fill
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    if (i < n)  // disp != null: resolving insert, pos[] takes the outcome (last fill of a batch)
      pos[i] = disp ? ht_insert_resolve(t, items[i], t.pend | (uint32_t)i, disp)
                    : ht_insert_min(t, items[i], t.pend | (uint32_t)i);
  }
}

// pass 2: owner(i) <=> bucket value == PENDING|i ; per-workgroup owner counts; flag kept in pos bit 31
template <int IPT>
__global__ __launch_bounds__(kBlock) void ht_count_kernel(HtView t, size_t n_host, const size_t *d_n, size_t cap,
                                                          uint32_t *__restrict__ pos,
                                                          uint32_t *__restrict__ block_sums,
                                                          uint32_t *d_num_items) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t n = resolve_count64(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
  if (blockIdx.x == 0 && threadIdx.x == 0) d_num_items[1] = d_num_items[0];  // count before this fill
  uint32_t cnt = 0;
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    if (i < n) {
      const uint32_t b = pos[i];
      const bool owner = b != kNoBucket && ht_value(t, t.table[b]) == (t.pend | (uint32_t)i);
      if (owner) { pos[i] = b | kPosOwner; ++cnt; }
    }
  }
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(cnt, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// pass 3: owners take local id = old_num_items + rank (rank in item order) and append to N2O
template <int IPT>
__global__ __launch_bounds__(kBlock) void ht_assign_kernel(HtView t, const uint32_t *__restrict__ items,
                                                           size_t n_host,
                                                           const size_t *d_n, size_t cap,
                                                           const uint32_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ block_offsets,
                                                           const uint32_t *d_num_items,
                                                           uint32_t *__restrict__ n2o, size_t max_items,
                                                           LayerSummary summary) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t n = resolve_count64(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (summary.num_dst) *summary.num_dst = d_num_items[1];
    if (summary.num_src) *summary.num_src = d_num_items[0];
    if (summary.num_total) *summary.num_total = d_num_items[0];
  }
  uint32_t running = d_num_items[1] + block_offsets[blockIdx.x];
  // item order inside the tile is r-major: i = tile0 + r*kBlock + tid
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    uint32_t b = 0;
    bool owner = false;
    if (i < n) {
      b = pos[i];
      owner = (b & kPosOwner) != 0;
    }
    uint32_t tot;
    const uint32_t rank = block_exclusive_rank<kWavesPerBlock>(owner, sh, &tot);
    if (owner) {
      const uint32_t local = running + rank;
      if (local < max_items) {
        // low half of the little-endian 64-bit bucket = value
        reinterpret_cast<uint32_t *>(&t.table[b & ~kPosOwner])[0] = t.gen_base | local;
        n2o[local] = items[i];
      }
    }
    running += tot;
  }
}

// passes 2 + 3 (+ most of 4) in ONE launch: a workgroup owns a contiguous chunk of `rounds` x 256 items
// (rounds from the device-side item count, so the whole grid shares the work evenly), counts its owners, gets
// the number of owners before it by a decoupled look-back over the earlier workgroups (fgnn_device.h) and
// assigns local ids -- every bucket is read ONCE instead of in a count kernel and again in an assign kernel,
// and the scan kernel between them is gone.  The remap is resolved on the spot for owners and for keys that
// already had a local id before this fill; only duplicates WITHIN the fill (value still PENDING|other) are
// left for ht_map_fix_kernel.  d_num_items[1] must hold the item count before the fill (set by pass 1).
__global__ __launch_bounds__(kBlock) void ht_count_assign_kernel(HtView t, const uint32_t *__restrict__ items,
                                                                 size_t n_host,
                                                                 const size_t *d_n, size_t cap,
                                                                 const uint32_t *__restrict__ pos,
                                                                 uint32_t *d_num_items, uint32_t *__restrict__ n2o,
                                                                 size_t max_items, LayerSummary summary,
                                                                 uint32_t *mapped, ScanWs scan, bool final_fill,
                                                                 const uint32_t *__restrict__ disp, bool exact) {
  // disp != null (implies final_fill): pos[] holds the OUTCOMES of a resolving insert (fgnn_device.h), not buckets --
  // the value the bucket read below would have returned, up to take-overs noted in disp[]: no table access at all.
  // exact (implies final_fill, disp == null): pos[] holds the FINAL outcomes (the partitioned fill,
  // hashtable_partition.hip): nothing to look up anywhere
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t sh_tile[2];
  const uint32_t n = (uint32_t)resolve_count64(n_host, d_n, cap);  // cap < 2^31 (host check)
  const uint32_t per_round = kBlock * gridDim.x;
  const uint32_t rounds = n ? (n - 1) / per_round + 1 : 1u;  // <= 32 by the host's grid choice
  const uint32_t chunk = rounds * kBlock;
  const uint32_t ntiles = n ? (n - 1) / chunk + 1 : 1u;
  const uint32_t tile = scan_take_tile(scan, sh_tile);
  if (tile >= ntiles) return;  // whole workgroup; nobody looks back at an unused tile
  phase_mark(scan, tile, 0);
  const uint32_t old = d_num_items[1];
  // owners among the items of tile `tl` (bit r of *mask: this thread's item of round r is one).  own = true: this
  // workgroup's tile -- the remap entries of non-owners are written on the way.  own = false: another tile's count,
  // recomputed by a waiter that helps (scan_prefix_help): same reads, no writes.
  auto count_chunk = [&](uint32_t tl, bool own, uint32_t *mask) -> uint32_t {
    const size_t c0 = (size_t)tl * chunk;
    uint32_t om = 0, cn = 0;
    // four rounds at a time: their loads are independent, keep them all in flight
    for (uint32_t r0 = 0; r0 < rounds; r0 += 4) {
      uint32_t bk[4], v[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = c0 + (size_t)(r0 + u) * kBlock + threadIdx.x;
        ok[u] = r0 + u < rounds && i < n;
        bk[u] = ok[u] ? pos[i] : kNoBucket;
      }
      if (exact) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t i = c0 + (size_t)(r0 + u) * kBlock + threadIdx.x;
          v[u] = !ok[u] ? FGNN_EMPTY_KEY : bk[u] == kPartIsOwner ? (t.pend | (uint32_t)i) : bk[u];
          bk[u] = 0;  // "has a bucket"
        }
      } else if (disp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t i = c0 + (size_t)(r0 + u) * kBlock + threadIdx.x;
          v[u] = ok[u] ? bk[u] : FGNN_EMPTY_KEY;
          if (ok[u] && v[u] == (t.pend | (uint32_t)i)) {  // held the key when it inserted: still?
            const uint32_t note = disp[i] ^ t.gen_base;   // this generation's note: pend|item that took the key over
            if ((note >> t.vp1) == 0u && (note & t.pend)) v[u] = note;
          }
          bk[u] = 0;  // "has a bucket"
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // a helper (own == false) reads buckets another workgroup may be rewriting right now (pend|i -> local id):
          // agent-scope atomic accesses on both sides make that a defined race whose either outcome is handled (see
          // the release / acquire note below), instead of plain loads the compiler may hoist or an XCD's L2 may serve
          // stale in the wrong direction
          if (bk[u] == kNoBucket) v[u] = FGNN_EMPTY_KEY;
          else if (own) v[u] = ht_value(t, t.table[bk[u]]);
          else v[u] = ht_value(t, __hip_atomic_load(&t.table[bk[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          const size_t i = c0 + (size_t)(r0 + u) * kBlock + threadIdx.x;
          if (bk[u] != kNoBucket && v[u] == (t.pend | (uint32_t)i)) {
            om |= 1u << (r0 + u);
            ++cn;
          } else if (own && mapped) {
            mapped[i] = v[u];  // final local id, or pend|owner (fixed up later), or EMPTY (no bucket)
          }
        }
      }
    }
    *mask = om;
    return cn;
  };
  const size_t chunk0 = (size_t)tile * chunk;
  uint32_t owner_mask = 0;
  const uint32_t cnt = count_chunk(tile, true, &owner_mask);
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(cnt, sh, &tot);
  phase_mark(scan, tile, 1);
  scan_publish_aggregate(scan, tile, tot);
  // Buckets of this tile's owners change (pend|i -> local id) further down, i.e. only after the aggregate has been
  // published: a helper that recounts this tile either still sees the pending values or -- release here, acquire +
  // descriptor re-check there -- finds the published word and takes that
  if (!final_fill && threadIdx.x == 0) __atomic_thread_fence(__ATOMIC_RELEASE);
  const uint32_t before = scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
    uint32_t mask_m, tot_m;
    const uint32_t cm = count_chunk(m, false, &mask_m);
    (void)block_exclusive_scan<kWavesPerBlock>(cm, sh, &tot_m);
    return tot_m;
  }, /*acquire_recheck=*/!final_fill);
  phase_mark(scan, tile, 2);
  if (tile == ntiles - 1 && threadIdx.x == 0) {
    const uint32_t now = old + before + tot;
    d_num_items[0] = now;
    if (summary.num_dst) *summary.num_dst = old;
    if (summary.num_src) *summary.num_src = now;
    if (summary.num_total) *summary.num_total = now;
  }
  uint32_t running = old + before;
  for (uint32_t r = 0; r < rounds; ++r) {
    const size_t i = chunk0 + (size_t)r * kBlock + threadIdx.x;
    const bool owner = (owner_mask >> r) & 1u;
    uint32_t t2;
    const uint32_t rank = block_exclusive_rank<kWavesPerBlock>(owner, sh, &t2);
    if (owner) {
      const uint32_t local = running + rank;
      if (local < max_items) {
        // low half of the bucket = value.  final_fill: nothing looks these keys up again before the table is reset
        // (last layer of a batch; the fix-up below takes a duplicate's id from its owner's remap entry instead)
        if (!final_fill)
          __hip_atomic_store(reinterpret_cast<uint32_t *>(&t.table[pos[i]]), t.gen_base | local, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        n2o[local] = items[i];
      }
      if (mapped) mapped[i] = local < max_items ? local : FGNN_EMPTY_KEY;
    }
    running += t2;
  }
  phase_mark(scan, tile, 3);
}

// the remap entries ht_count_assign_kernel could not resolve: duplicates inside the fill whose owner had not
// been numbered yet.  The unresolved entry holds pend|owner's item index, and the owner's own remap entry (written
// by the previous kernel) is its local id: one read of a small array instead of a 64 MiB table probe.
__global__ __launch_bounds__(kBlock) void ht_map_fix_kernel(FixTail fix) { run_fix_tail(fix, 0); }

// pass 4: mapped[i] = local id of items[i] (bucket known)
template <int IPT>
__global__ __launch_bounds__(kBlock) void ht_map_pos_kernel(HtView t, size_t n_host, const size_t *d_n, size_t cap,
                                                            const uint32_t *__restrict__ pos,
                                                            uint32_t *__restrict__ mapped) {
  const size_t n = resolve_count64(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    if (i < n) {
      const uint32_t b = pos[i] & ~kPosOwner;
      mapped[i] = b != kNoBucket ? ht_value(t, t.table[b]) : FGNN_EMPTY_KEY;
    }
  }
}

// GPUMapEdges for ids without a remembered bucket
template <int IPT>
__global__ __launch_bounds__(kBlock) void ht_map_probe_kernel(HtView t, const uint32_t *__restrict__ items,
                                                              size_t n_host,
                                                              const size_t *d_n, size_t cap,
                                                              uint32_t *__restrict__ mapped) {
  const size_t n = resolve_count64(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    if (i < n) {
      uint32_t b;
      mapped[i] = ht_find(t, items[i], &b);
    }
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" fgnn_hashtable *fgnn_hashtable_create(size_t max_items, int *h_err) {
  // fills of any size up to 2^30 items: one generation only, every Reset wipes the table
  return fgnn_hashtable_create_ex(max_items, (size_t(1) << 30) - 1, h_err);
}

extern "C" fgnn_hashtable *fgnn_hashtable_create_ex(size_t max_items, size_t max_fill_items, int *h_err) {
  auto fail = [&](int code) -> fgnn_hashtable * {
    if (h_err) *h_err = code;
    return nullptr;
  };
  if (max_items == 0 || max_items >= (size_t(1) << 30) || max_fill_items >= (size_t(1) << 30)) return fail(FGNN_EINVAL);
  size_t cap = 1024;
  uint32_t lg = 10;
  while (cap < 2 * max_items) { cap <<= 1; ++lg; }
  auto *ht = new fgnn_hashtable();
  ht->capacity = cap;
  ht->max_items = max_items;
  ht->shift = 32 - lg;
  ht->table = nullptr;
  ht->n2o = nullptr;
  ht->n2o_owned = nullptr;
  ht->d_num_items = nullptr;
  // value field: the largest index is a local id (< max_items) or a pending item index (< max_fill_items)
  const size_t max_index = (max_items > max_fill_items ? max_items : max_fill_items);
  uint32_t vbits = 1;
  while ((size_t(1) << vbits) <= max_index) ++vbits;
  ht->vp1 = vbits + 1;                                 // + the pending flag; <= 31
  ht->gen_limit = (1u << (32 - ht->vp1)) - 1u;         // >= 1; the all-ones generation is the wiped pattern
  ht->gen = 0;
  ht->max_fill_items = max_fill_items;
  ht->disp = nullptr;
  ht->part = nullptr;
  ht->scan = new ScanWsHost();
  if (ht->scan->create(4096) != FGNN_OK ||
      hipMalloc(&ht->table, cap * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&ht->n2o_owned, max_items * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&ht->d_num_items, 2 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(ht->table, 0xFF, cap * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ht->d_num_items, 0, 2 * sizeof(uint32_t)) != hipSuccess) {
    fgnn_hashtable_destroy(ht);
    return fail(FGNN_EHIP);
  }
  ht->n2o = ht->n2o_owned;
  if (max_fill_items <= (size_t(1) << 24)) {  // the batch driver's tables: notes of the resolving insert
    const size_t bytes = (max_fill_items + 1) * sizeof(uint32_t);
    if (hipMalloc(&ht->disp, bytes) != hipSuccess || hipMemset(ht->disp, 0, bytes) != hipSuccess) {
      fgnn_hashtable_destroy(ht);
      return fail(FGNN_EHIP);
    }
    // null: the last fill goes through the global table too (FGNN_HT_PARTITION=0: A/B switch of the profiling build,
    // read per table so that one process can hold both kinds)
    if (tune_int("FGNN_HT_PARTITION", 1) != 0) ht->part = partition_create(max_items, max_fill_items);
  }
  if (h_err) *h_err = FGNN_OK;
  return ht;
}

extern "C" int fgnn_hashtable_set_n2o(fgnn_hashtable *ht, uint32_t *storage) {
  if (!ht) return FGNN_EINVAL;
  ht->n2o = storage ? storage : ht->n2o_owned;
  return FGNN_OK;
}

extern "C" int fgnn_hashtable_start_batch(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                          uint32_t *items_copy, fgnn_batch_meta *d_meta, uint64_t key,
                                          uint32_t num_layers, void *stream) {
  if (!ht || (!items && num_items) || num_items > ht->max_items) return FGNN_EINVAL;
  const size_t nb = num_items ? div_up(num_items, kBlock) : 1;
  hipLaunchKernelGGL(ht_start_batch_kernel, dim3(nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream), ht_view(ht),
                     items, num_items, ht->n2o, items_copy, ht->d_num_items, d_meta, key, num_layers);
  return launch_status(__func__);
}

extern "C" void fgnn_hashtable_destroy(fgnn_hashtable *ht) {
  if (!ht) return;
  if (ht->table) (void)hipFree(ht->table);
  if (ht->n2o_owned) (void)hipFree(ht->n2o_owned);
  if (ht->d_num_items) (void)hipFree(ht->d_num_items);
  if (ht->disp) (void)hipFree(ht->disp);
  partition_destroy(ht->part);
  if (ht->scan) {
    ht->scan->destroy();
    delete ht->scan;
  }
  delete ht;
}

extern "C" size_t fgnn_hashtable_capacity(const fgnn_hashtable *ht) { return ht ? ht->capacity : 0; }
extern "C" const uint32_t *fgnn_hashtable_n2o(const fgnn_hashtable *ht) { return ht ? ht->n2o : nullptr; }
extern "C" const uint32_t *fgnn_hashtable_d_num_items(const fgnn_hashtable *ht) {
  return ht ? ht->d_num_items : nullptr;
}

// Reset: every bucket reads as empty afterwards.  Normally a generation bump (kernels launched from now on carry
// the new generation); the table is physically wiped when the generations are used up.  zero_counts: also clear the
// item counts (the batch driver does not need that, its first kernel of a batch sets them).
int fgnn::hashtable_next_generation(fgnn_hashtable *ht, void *stream, bool zero_counts) {
  auto s = static_cast<hipStream_t>(stream);
  if (ht->gen + 1 < ht->gen_limit) {
    ++ht->gen;
    if (zero_counts) FGNN_HIP_CHECK(hipMemsetAsync(ht->d_num_items, 0, 2 * sizeof(uint32_t), s));
    return FGNN_OK;
  }
  ht->gen = 0;
  // notes of earlier cycles must not read as this cycle's
  if (ht->disp) FGNN_HIP_CHECK(hipMemsetAsync(ht->disp, 0, (ht->max_fill_items + 1) * sizeof(uint32_t), s));
  size_t blocks = div_up(ht->capacity / 2, (size_t)kBlock * 4);
  const size_t max_blocks = (size_t)device_cu_count() * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(ht_wipe_kernel, dim3(blocks), dim3(kBlock), 0, s, ht->table, ht->capacity, ht->d_num_items);
  return launch_status(__func__);
}

extern "C" int fgnn_hashtable_reset(fgnn_hashtable *ht, void *stream) {
  if (!ht) return FGNN_EINVAL;
  return fgnn::hashtable_next_generation(ht, stream, true);
}

extern "C" int fgnn_hashtable_fill_unique(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                          void *stream) {
  if (!ht || (!items && num_items)) return FGNN_EINVAL;
  if (num_items == 0) return FGNN_OK;
  if (num_items > ht->max_items) return FGNN_EINVAL;
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ht_fill_unique_kernel, dim3(div_up(num_items, kBlock)), dim3(kBlock), 0, s, ht_view(ht), items,
                     num_items, ht->n2o, ht->d_num_items, ht->max_items);
  hipLaunchKernelGGL(ht_advance_kernel, dim3(1), dim3(1), 0, s, ht->d_num_items, (uint32_t)num_items);
  return launch_status(__func__);
}

extern "C" int fgnn_hashtable_fill_duplicates(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                              const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped,
                                              void *ws, size_t ws_bytes, void *stream) {
  return fgnn::hashtable_fill_duplicates_ex(ht, items, num_items, d_num_items, num_items_cap, mapped, ws, ws_bytes,
                                            stream, fgnn::LayerSummary{nullptr, nullptr, nullptr}, false, nullptr, false);
}

// grid of the one-launch count+assign path, 0 if the fill is too large for it: the grid must be resident at once
// (prefix over the lower-numbered workgroups) and a chunk at most 32 rounds
static size_t count_assign_grid(size_t cap, const fgnn::ScanWsHost *scan) {
  static int per_cu = -1;
  if (per_cu < 0 &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ht_count_assign_kernel, kBlock, 0) != hipSuccess)
    per_cu = 0;
  const size_t nb1 = div_up(cap, (size_t)kBlock);
  size_t grid = (size_t)per_cu * device_cu_count() * 3 / 4;
  if (grid > scan->ws.max_tiles) grid = scan->ws.max_tiles;
  if (grid > nb1) grid = nb1;
  return grid > 0 && div_up(cap, grid * kBlock) <= 32 ? grid : 0;
}

bool fgnn::hashtable_can_resolve(const fgnn_hashtable *ht, size_t cap) {
  static const bool enabled = tune_int("FGNN_HT_RESOLVE", 1) != 0;  // A/B switch, profiling build only
  return enabled && ht && ht->disp && ht->scan && cap > 0 && cap <= ht->max_fill_items && count_assign_grid(cap, ht->scan) > 0;
}

// can a fill of `cap` items take the partitioned, table-free path (hashtable_partition.hip)?  The batch driver asks for
// EVERY layer of a batch before it promises `table_free` to any of them: a fill that fell back to the global insert
// would dedup against a table the earlier, partitioned fills never wrote their new nodes to
bool fgnn::hashtable_can_partition(const fgnn_hashtable *ht, size_t cap) {
  return ht && ht->scan && cap > 0 && cap <= ht->max_fill_items && partition_fits(ht->part, ht, cap) &&
         count_assign_grid(cap, ht->scan) > 0;
}

int fgnn::hashtable_fill_duplicates_ex(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                       const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped, void *ws,
                                       size_t ws_bytes, void *stream, LayerSummary summary, bool already_inserted,
                                       ScanWsHost *scan, bool final_fill, bool resolved, FixTail *owed_fix,
                                       const FixTail *carry_fix, bool table_free) {
  if (!ht) return FGNN_EINVAL;
  if (owed_fix) *owed_fix = no_fix_tail();  // mapped != null on return: the caller owes this fill's fix-up
  FixTail carry = carry_fix && carry_fix->mapped ? *carry_fix : no_fix_tail();
  // resolved: pos[] holds insert outcomes (sample_khop_fused(..., resolve = true)); only the one-launch path reads them
  if (resolved && !(already_inserted && final_fill && mapped && ht->disp)) return FGNN_EINVAL;
  if (!scan) scan = ht->scan;  // hashtable_can_resolve looks at ht->scan: the descriptors used must be those
  size_t cap = d_num_items ? num_items_cap : num_items;
  if (cap == 0) return carry.mapped ? hashtable_map_fix(carry, stream) : FGNN_OK;
  if (!items || cap > ht->max_fill_items) return FGNN_EINVAL;  // pending indices must fit the value field
  auto s = static_cast<hipStream_t>(stream);
  const HtView tv = ht_view(ht);
  // latency-bound at mini-batch sizes: one item per lane (8x more waves in flight) unless the input is huge
  const int ipt = cap <= (4u << 20) ? 1 : kItemsPerThread;
  const size_t nb = div_up(cap, (size_t)kBlock * ipt);
  // scratch layout: pos[cap] | block_sums[nb + 1]
  const size_t need = (cap + nb + 2) * sizeof(uint32_t);
  if (ws_bytes < need) return FGNN_ENOSPC;
  uint32_t *pos = static_cast<uint32_t *>(ws);
  uint32_t *sums = pos + cap;
#define FGNN_HT(KERNEL, ...)                                                                  \
  do {                                                                                        \
    if (ipt == 1) hipLaunchKernelGGL((KERNEL<1>), dim3(nb), dim3(kBlock), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<kItemsPerThread>), dim3(nb), dim3(kBlock), 0, s, __VA_ARGS__); \
  } while (0)
  // already_inserted: the sampler kernel inserted each edge as it produced it and left the buckets in pos[]
  bool exact = false;
  // table_free is the caller's promise for the WHOLE batch (hashtable_can_partition held for every fill): a table-free
  // fill that cannot be partitioned here would silently dedup against an incomplete table
  if (table_free && !final_fill && !(mapped && scan == ht->scan && hashtable_can_partition(ht, cap))) return FGNN_EINVAL;
  if (!already_inserted && (final_fill || table_free) && mapped && scan == ht->scan &&
      partition_fits(ht->part, ht, cap) && count_assign_grid(cap, scan) > 0) {
    // the batch's last fill (or any fill of a batch that never reads the table): partitioned by hash, deduplicated in
    // LDS, the global table untouched
    const int rc = partition_fill(ht->part, ht, items, num_items, d_num_items, cap, pos, s, carry);
    if (rc != FGNN_OK) return rc;
    exact = true;
  } else if (!already_inserted) {
    // the last fill of a batch done here (samplers that do not insert themselves) resolves too
    resolved = final_fill && mapped && scan == ht->scan && hashtable_can_resolve(ht, cap);
    uint32_t *const disp = resolved ? ht->disp : nullptr;
    const unsigned grid = (unsigned)(nb + carry.blocks);
    if (ipt == 1)
      hipLaunchKernelGGL((ht_insert_kernel<1>), dim3(grid), dim3(kBlock), 0, s, tv, items, num_items, d_num_items, cap,
                         pos, ht->d_num_items, disp, (uint32_t)nb, carry);
    else
      hipLaunchKernelGGL((ht_insert_kernel<kItemsPerThread>), dim3(grid), dim3(kBlock), 0, s, tv, items, num_items,
                         d_num_items, cap, pos, ht->d_num_items, disp, (uint32_t)nb, carry);
  } else if (carry.mapped) {
    const int rc = hashtable_map_fix(carry, stream);
    if (rc != FGNN_OK) return rc;
  }
  if (!scan) scan = ht->scan;
  if (scan) {
    const size_t grid = count_assign_grid(cap, scan);
    if (grid > 0) {
      hipLaunchKernelGGL(ht_count_assign_kernel, dim3(grid), dim3(kBlock), 0, s, tv, items, num_items,
                         d_num_items, cap, pos, ht->d_num_items, ht->n2o, ht->max_items, summary, mapped,
                         scan->next(1, grid), (final_fill || exact) && mapped != nullptr,
                         resolved && !exact ? ht->disp : nullptr, exact);
      if (mapped) {
        const FixTail fix{mapped, d_num_items, num_items, cap, tv.pend, fix_tail_blocks(cap)};
        if (owed_fix) *owed_fix = fix;
        else return hashtable_map_fix(fix, stream);
      }
      return launch_status(__func__);
    }
  }
  if (resolved || exact) return FGNN_EINVAL;  // (both paths made sure of the one-launch grid before they started)
  // d_num_items[1] keeps the old count (set by the count kernel) for pass 3; d_num_items[0] advances in the scan
  FGNN_HT(ht_count_kernel, tv, num_items, d_num_items, cap, pos, sums, ht->d_num_items);
  if (launch_scan_block_sums(sums, nb, nullptr, nullptr, ht->d_num_items + 1, ht->d_num_items, s, nullptr,
                             (uint32_t)(kBlock * ipt), d_num_items) != FGNN_OK)
    return FGNN_EHIP;
  FGNN_HT(ht_assign_kernel, tv, items, num_items, d_num_items, cap, pos, sums, ht->d_num_items, ht->n2o,
          ht->max_items, summary);
  if (mapped) FGNN_HT(ht_map_pos_kernel, tv, num_items, d_num_items, cap, pos, mapped);
#undef FGNN_HT
  return launch_status(__func__);
}

int fgnn::hashtable_map_fix(const FixTail &fix, void *stream) {
  if (!fix.mapped || fix.blocks == 0) return FGNN_EINVAL;
  hipLaunchKernelGGL(ht_map_fix_kernel, dim3(fix.blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), fix);
  return launch_status(__func__);
}

extern "C" int fgnn_hashtable_map(const fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                  const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped, void *stream) {
  if (!ht) return FGNN_EINVAL;
  size_t cap = d_num_items ? num_items_cap : num_items;
  if (cap == 0) return FGNN_OK;
  if (!items || !mapped) return FGNN_EINVAL;
  hipLaunchKernelGGL((ht_map_probe_kernel<1>), dim3(div_up(cap, kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), ht_view(ht), items, num_items, d_num_items, cap, mapped);
  return launch_status(__func__);
}
