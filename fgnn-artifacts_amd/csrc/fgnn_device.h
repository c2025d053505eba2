// fgnn_device.h -- device-side building blocks shared by the gfx950 kernels: Philox draws,
// wave64 ballot/prefix compaction, block scans.  CDNA4 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "fgnn_hip.h"

namespace fgnn { struct ScanWsHost; struct PartWs; }

// OrderedHashTable state (hashtable.hip); the samplers insert into it directly (fused path)
struct fgnn_hashtable {
  unsigned long long *table;  // capacity buckets of {key:hi32, value:lo32}
  uint32_t *n2o;              // max_items; where new nodes are appended (owned, or a batch's input_nodes buffer)
  uint32_t *n2o_owned;
  uint32_t *d_num_items;      // [2]: current count, count before the running fill
  size_t capacity;            // power of two
  size_t max_items;
  uint32_t shift;             // 32 - log2(capacity)
  fgnn::ScanWsHost *scan;     // look-back descriptors of the single-pass count+assign kernel
  // bucket value = [generation : 32 - vp1 bits][pending : 1][index : vp1 - 1 bits].  A bucket whose generation is not
  // the current one is EMPTY: Reset() is a generation bump, the table is only really wiped when the counter wraps.
  uint32_t vp1;               // bits of pending flag + index (<= 31)
  uint32_t gen;               // current generation, 0 .. gen_limit - 1
  uint32_t gen_limit;         // (1 << (32 - vp1)) - 1: the all-ones generation marks never-used / wiped buckets
  size_t max_fill_items;      // largest fill (pending index) the value field can hold
  // "you have been replaced" notes of the resolving insert (ht_insert_resolve): disp[j] = generation | pending | i says
  // that item i took the key over from pending item j.  Generation-tagged like the buckets, zeroed with the table's
  // wipe.  Only allocated for tables made with a modest max_fill_items (the batch driver's); null otherwise.
  uint32_t *disp;
  // the batch's last fill, hash-partitioned and deduplicated in LDS (hashtable_partition.hip); null: not available
  fgnn::PartWs *part;
};

namespace fgnn {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kItemsPerThread = 8;   // edge-parallel kernels: 2048 items per workgroup
constexpr int kTile = kBlock * kItemsPerThread;

// records "<what>: <hip error string>" for fgnn_last_error(); defined in capi.hip
void set_last_error(const char *what, hipError_t e);

#define FGNN_HIP_CHECK(expr)                      \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) {                       \
      ::fgnn::set_last_error(#expr, _e);          \
      return FGNN_EHIP;                           \
    }                                             \
  } while (0)

// after a batch of hipLaunchKernelGGL calls
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(what, e);
    return FGNN_EHIP;
  }
  return FGNN_OK;
}

// ---- build flavours --------------------------------------------------------------------------------------------
// The shipped library (libfgnn_hip.so) reads NO tuning or ablation switch from the environment: a stray variable must
// not change -- let alone corrupt -- a training run.  The A/B and ablation knobs the tools/ scripts use exist only in
// the profiling build (make prof -> lib/libfgnn_hip_prof.so, -DFGNN_PROFILING); there `tune_int` reads FGNN_* variables
// and the kernels take an `ablate` mask.  In the release build `tune_int` is the default and `ablate` a constant 0, so
// the compiler drops the branches.
#ifdef FGNN_PROFILING
inline int tune_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
#define FGNN_ABLATE_PARAM , uint32_t ablate
#define FGNN_ABLATE_ARG(x) , (uint32_t)(x)
#else
inline int tune_int(const char *, int dflt) { return dflt; }
constexpr uint32_t ablate = 0u;
#define FGNN_ABLATE_PARAM
#define FGNN_ABLATE_ARG(x)
#endif

// ---- Philox4x32-10, addressed exactly like oracle/fgnn_oracle.c:fgnn_philox_draw ------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox_block(uint64_t seed, uint64_t batch_key, uint32_t tag,
                                              uint32_t item, uint32_t block) {
  uint32_t c0 = block, c1 = item, c2 = tag, c3 = (uint32_t)batch_key;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(batch_key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ uint32_t pick_word(const u32x4 &b, uint32_t j) {
  const uint32_t lo = (j & 1u) ? b.y : b.x;
  const uint32_t hi = (j & 1u) ? b.w : b.z;
  return (j & 2u) ? hi : lo;
}

__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t batch_key, uint32_t tag,
                                               uint32_t item, uint32_t j) {
  return pick_word(philox_block(seed, batch_key, tag, item, j >> 2), j);
}

// ---- dedup hash table probes (see hashtable.hip for the design) -----------------------------------
constexpr unsigned long long kEmpty64 = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kPosOwner = 0x80000000u;  // flag on a remembered bucket index (three-kernel path): item owns its key
constexpr uint32_t kNoBucket = 0x7FFFFFFFu;  // capacity <= 2^31, so never a real bucket index

// what the kernels need of a table: the current generation is baked in at launch time
struct HtView {
  unsigned long long *table;
  uint32_t shift, mask;
  uint32_t vp1;       // bits of pending + index
  uint32_t gen_base;  // generation << vp1
  uint32_t pend;      // pending flag = 1 << (vp1 - 1): value = pend | i while item i is a pending first occurrence
};
inline HtView ht_view(const fgnn_hashtable *ht) {
  return HtView{ht->table, ht->shift, (uint32_t)(ht->capacity - 1), ht->vp1, ht->gen << ht->vp1, 1u << (ht->vp1 - 1)};
}

__device__ __forceinline__ uint32_t hash_slot(uint32_t id, uint32_t shift, uint32_t mask) {
  return ((id * 0x9E3779B1u) >> shift) & mask;
}
// does this bucket word belong to the view's generation?  (stale generations and the wiped pattern read as empty)
__device__ __forceinline__ bool ht_live(const HtView &t, unsigned long long word) {
  return (((uint32_t)word ^ t.gen_base) >> t.vp1) == 0u;
}
// value field without the generation
__device__ __forceinline__ uint32_t ht_value(const HtView &t, unsigned long long word) {
  return (uint32_t)word ^ t.gen_base;
}

// Inserts (id, value) keeping the minimum value per key.  `value` = pend|index or a local id.  Returns the bucket.
__device__ __forceinline__ uint32_t ht_insert_min(const HtView &t, uint32_t id, uint32_t value) {
  const uint32_t mine_v = t.gen_base | value;
  const unsigned long long mine = ((unsigned long long)id << 32) | mine_v;
  uint32_t h = hash_slot(id, t.shift, t.mask);
  // load factor <= 0.5 by construction; the bound only keeps a violated contract (more distinct
  // keys than max_items) from hanging the GPU
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    unsigned long long cur = t.table[h];
    for (int tries = 0; tries < 4 && !ht_live(t, cur); ++tries) {  // free for this generation: claim it
      const unsigned long long old = atomicCAS(&t.table[h], cur, mine);
      if (old == cur) return h;
      cur = old;  // the word we read was out of date: look at what is really there
    }
    if (ht_live(t, cur) && (uint32_t)(cur >> 32) == id) {
      if ((uint32_t)cur > mine_v) atomicMin(&t.table[h], mine);
      return h;
    }
    h = (h + 1) & t.mask;
  }
  return kNoBucket;
}

// ---- resolving insert (last fill of a batch) -----------------------------------------------------------------------
// ht_insert_min leaves the outcome in the table and the later passes read it back: one random 8-byte read per item in
// ht_count_assign_kernel.  The resolving insert RETURNS the outcome instead -- the key's value as this insert left or
// found it: pend|i (item i holds the key), pend|j with j < i (duplicate of a pending item), a local id (the node was
// known before this fill), or EMPTY (table full) -- so that the next pass reads a sequential array.  What an insert
// cannot know is whether a LATER insert with a smaller index takes the key over; that insert's atomicMin returns the
// value it replaced, i.e. it knows exactly whom it replaced, and leaves that item a note in disp[].  A duplicate that
// was pointed at a holder which is replaced afterwards reaches the final owner by following the notes
// (ht_map_fix_kernel).  Stale reads (non-coherent L2s) only ever show an older = larger value: at worst a redundant
// atomicMin or one more hop in that chain.
__device__ __forceinline__ uint32_t ht_lower_resolve(const HtView &t, uint32_t h, uint32_t id, uint32_t seen_v,
                                                     uint32_t value, uint32_t *disp) {
  if (seen_v < value) return seen_v;  // a local id, or an earlier pending item
  const unsigned long long mine = ((unsigned long long)id << 32) | t.gen_base | value;
  const uint32_t old_v = ht_value(t, atomicMin(&t.table[h], mine));
  if (old_v < value) return old_v;    // somebody smaller got there between the read and the atomic
  disp[old_v & (t.pend - 1u)] = t.gen_base | value;  // old_v = pend|j: item j is told who replaced it
  return value;
}

__device__ __forceinline__ uint32_t ht_insert_resolve(const HtView &t, uint32_t id, uint32_t value, uint32_t *disp) {
  const unsigned long long mine = ((unsigned long long)id << 32) | t.gen_base | value;
  uint32_t h = hash_slot(id, t.shift, t.mask);
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    unsigned long long cur = t.table[h];
    for (int tries = 0; tries < 4 && !ht_live(t, cur); ++tries) {
      const unsigned long long old = atomicCAS(&t.table[h], cur, mine);
      if (old == cur) return value;
      cur = old;
    }
    if (ht_live(t, cur) && (uint32_t)(cur >> 32) == id) return ht_lower_resolve(t, h, id, ht_value(t, cur), value, disp);
    h = (h + 1) & t.mask;
  }
  return FGNN_EMPTY_KEY;
}

template <int UB>
__device__ __forceinline__ void ht_insert_resolve_batch(const HtView &t, const uint32_t (&key)[UB],
                                                        const uint32_t (&val)[UB], const bool (&live)[UB],
                                                        uint32_t *disp, uint32_t (&outcome)[UB]) {
  uint32_t h[UB];
  unsigned long long cur[UB], old[UB];
  bool tried[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    h[u] = hash_slot(key[u], t.shift, t.mask);
    cur[u] = live[u] ? t.table[h[u]] : 0ull;
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    old[u] = cur[u];
    tried[u] = live[u] && !ht_live(t, cur[u]);
    if (tried[u]) old[u] = atomicCAS(&t.table[h[u]], cur[u], ((unsigned long long)key[u] << 32) | t.gen_base | val[u]);
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    if (!live[u]) continue;
    if (tried[u] && old[u] == cur[u]) outcome[u] = val[u];
    else if (ht_live(t, old[u]) && (uint32_t)(old[u] >> 32) == key[u])
      outcome[u] = ht_lower_resolve(t, h[u], key[u], ht_value(t, old[u]), val[u], disp);
    else outcome[u] = ht_insert_resolve(t, key[u], val[u], disp);
  }
}

// UB independent inserts with their memory operations overlapped: all probe loads are issued first, then all
// CAS, and only keys that collide with a different key fall back to the sequential probe loop.  A lane that
// inserts its UB edges one after the other pays UB dependent (load + CAS) round trips to the memory-side
// atomic unit; here it pays about one.
template <int UB>
__device__ __forceinline__ void ht_insert_min_batch(const HtView &t, const uint32_t (&key)[UB],
                                                    const uint32_t (&val)[UB], const bool (&live)[UB],
                                                    uint32_t (&bucket)[UB]) {
  uint32_t h[UB];
  unsigned long long cur[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    h[u] = hash_slot(key[u], t.shift, t.mask);
    cur[u] = live[u] ? t.table[h[u]] : 0ull;
  }
  unsigned long long old[UB];
  bool tried[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    old[u] = cur[u];
    tried[u] = live[u] && !ht_live(t, cur[u]);
    if (tried[u]) old[u] = atomicCAS(&t.table[h[u]], cur[u], ((unsigned long long)key[u] << 32) | t.gen_base | val[u]);
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    if (!live[u]) continue;
    const uint32_t mine_v = t.gen_base | val[u];
    if (tried[u] && old[u] == cur[u]) {  // our CAS installed the key
      bucket[u] = h[u];
    } else if (ht_live(t, old[u]) && (uint32_t)(old[u] >> 32) == key[u]) {  // key already there: keep the minimum
      if ((uint32_t)old[u] > mine_v) atomicMin(&t.table[h[u]], ((unsigned long long)key[u] << 32) | mine_v);
      bucket[u] = h[u];
    } else {  // slot taken by another key, or our view of it was out of date: ordinary probing from here
      bucket[u] = ht_insert_min(t, key[u], val[u]);
    }
  }
}

// ---- wave64 / block primitives ----------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// rank of this lane among the lanes whose predicate is set, and the wave total (ballot + popcount)
__device__ __forceinline__ uint32_t wave_rank(bool pred, uint32_t *total) {
  const unsigned long long m = __ballot(pred);
  *total = (uint32_t)__popcll(m);
  return (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
}

// inclusive wave scan by shuffles
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, kWave);
    if (lane_id() >= d) v += t;
  }
  return v;
}

// exclusive block scan of one value per thread (NW waves per block); *block_total gets the sum.
// `sh` needs NW words.  Contains two barriers.
template <int NW = kWavesPerBlock>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *sh, uint32_t *block_total) {
  const uint32_t inc = wave_inclusive_scan(v);
  if (lane_id() == kWave - 1) sh[wave_id()] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t t = sh[w];
    if (w < wave_id()) base += t;
    tot += t;
  }
  __syncthreads();
  *block_total = tot;
  return base + inc - v;
}

// exclusive block scan of a 0/1 flag per thread using ballots (cheaper than the shuffle scan)
template <int NW = kWavesPerBlock>
__device__ __forceinline__ uint32_t block_exclusive_rank(bool pred, uint32_t *sh, uint32_t *block_total) {
  uint32_t wt;
  const uint32_t r = wave_rank(pred, &wt);
  if (lane_id() == 0) sh[wave_id()] = wt;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t t = sh[w];
    if (w < wave_id()) base += t;
    tot += t;
  }
  __syncthreads();
  *block_total = tot;
  return base + r;
}

// ---- single-pass prefix across workgroups ------------------------------------------------------------------
// Replaces "per-workgroup count kernel -> scan kernel -> consumer kernel" by ONE kernel: workgroup `tile`
// (= blockIdx.x) publishes its aggregate and sums the aggregates its predecessors published.  Each descriptor is ONE
// 64-bit word {generation|status : value}, written and polled with relaxed agent-scope atomics: flag and
// payload travel together, so no fence is needed (MI355X guide, inter-workgroup hand-off, single granule).
// The generation (one per launch) makes stale descriptors of earlier launches read as "not ready": no reset
// between launches.
// Forward progress (the three kernels of the papers100M chain: k-hop sampler, dedup count+assign, cache split):
// tile = blockIdx.x, and a workgroup that has polled a predecessor's descriptor for longer than any healthy launch
// takes stops waiting and HELPS -- it recomputes the missing tile's aggregate itself (scan_prefix_help: an aggregate is
// a pure function of the launch's read-only inputs), publishes it on the missing tile's behalf and moves on.  So a
// resident workgroup always terminates, whatever the dispatch order, however few workgroups are resident and whatever
// other streams or processes put on the GPU meanwhile; the fast path costs nothing extra.  (Round 1 relied on in-order
// dispatch per XCD plus a residency cap computed for an idle GPU and flagged the batch after a seconds-long spin.)
// Tried first and measured: drawing the tile as a start-order TICKET (one atomicAdd per workgroup at entry) gives the
// same guarantee by construction, but same-address device-scope atomics complete at ~29 ns each on MI355X, i.e. 900
// tiles serialise for 26 us: layer-0 sampler 50.6 -> 76.8 us, count+assign 10.0 -> 26.9, cache split 16.6 -> 31.5,
// whole step 0.134 -> 0.198 ms (profiles/r02_ticket_ab.txt).  Tickets remain for hash_dedup_kernel only (<= 350 tiles
// on a 270 us kernel; its aggregate is the outcome of a rejection loop that a helper could not redo without the LDS
// the waiting tile's own selections occupy).
// Spins there stay bounded as a last line of defence: on timeout an error word is set (the batch summary's `overflow`
// in the batch driver), the prefix is wrong, the kernel terminates.
struct ScanWs {
  unsigned long long *desc;  // [max_tiles]
  uint32_t *error;           // [1] set to 1 if a spin timed out
  uint32_t gen;              // this launch's generation (1 .. 2^30-1)
  uint32_t max_tiles;
  unsigned long long *log;   // diagnostics (fgnn_debug_phase_log): [tile][8] wall-clock stamps per phase, or null
  uint32_t *ticket;          // [1] running ticket counter (null: tile = blockIdx.x)
  uint32_t ticket_base;      // counter value when this launch draws its first ticket
  uint32_t help_after;       // scan_prefix_help: polls of one descriptor before the waiter starts helping
  unsigned long long *helps; // diagnostics: aggregates recomputed by helpers, process-wide (fgnn_debug_scan_helps), or null
};

// tile of this workgroup = order in which it started among the workgroups of its launch (w.ticket != null), so that it
// only ever waits for workgroups that are already running.  EVERY workgroup of the grid must call this exactly once,
// before any early exit (the host advances the base by the grid size).  The counter is never reset: launches that share
// descriptors are stream-ordered and the host knows every grid size, so launch i's tickets are
// [base_i, base_i + grid_i) modulo 2^32.  `sh` = one LDS word; contains two barriers.
__device__ __forceinline__ uint32_t scan_take_tile(const ScanWs &w, uint32_t *sh) {
  if (!w.ticket) return blockIdx.x;
  if (threadIdx.x == 0)
    *sh = __hip_atomic_fetch_add(w.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - w.ticket_base;
  __syncthreads();
  const uint32_t t = *sh;
  __syncthreads();
  return t;
}

constexpr uint32_t kPhaseLogTiles = 4096, kPhaseLogKinds = 4;
// phase timestamps of the single-pass kernels (100 MHz wall clock); a no-op unless a log buffer is installed
__device__ __forceinline__ void phase_mark(const ScanWs &w, uint32_t tile, int phase) {
  if (w.log && threadIdx.x == 0 && tile < kPhaseLogTiles) w.log[(size_t)tile * 8 + phase] = wall_clock64();
}
unsigned long long *phase_log_base();  // capi.hip
unsigned long long *scan_help_counter();  // capi.hip: device word counting helped tiles (allocated on first use)
int scan_help_after_override();           // capi.hip: -1 unless fgnn_debug_set_scan_help_after() was called
// Where a timed-out cross-workgroup wait is reported for launches made by this host thread: the batch driver points
// it at the batch summary's `overflow` word (so the host sees it with the batch), otherwise the descriptors' own word.
uint32_t *&scan_error_sink();           // capi.hip (thread-local)
struct ScanErrorSink {
  uint32_t *prev;
  explicit ScanErrorSink(uint32_t *p) : prev(scan_error_sink()) { scan_error_sink() = p; }
  ~ScanErrorSink() { scan_error_sink() = prev; }
};

constexpr uint32_t kScanAggregate = 1u;

__device__ __forceinline__ void scan_publish(const ScanWs &w, uint32_t tile, uint32_t status, uint32_t value) {
  const unsigned long long word = ((unsigned long long)((w.gen << 2) | status) << 32) | value;
  __hip_atomic_store(&w.desc[tile], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of `aggregate` over tiles [0, tile); call from all threads (`sh` = one LDS word).
// Grids here are at most a few thousand tiles and start together, so a chained look-back (walk back until a
// tile that already knows its inclusive prefix) degenerates into tile/64 dependent round trips.  Instead every
// tile publishes only its aggregate and sums ALL its predecessors' aggregates itself: thread j polls tiles
// j, j + blockDim, ... (coalesced 8-byte loads), i.e. ONE memory round trip after the slowest predecessor has
// published, no dependence between tiles.  O(tiles^2) 8-byte loads in total -- ~1 M for 1500 tiles, noise.
// the two halves are also usable apart: publish as soon as the aggregate is known, sum the predecessors later
// (by then they have usually all published: no waiting)
__device__ __forceinline__ void scan_publish_aggregate(const ScanWs &w, uint32_t tile, uint32_t aggregate) {
  if (threadIdx.x == 0) scan_publish(w, tile, kScanAggregate, aggregate);
}

__device__ __forceinline__ uint32_t scan_prefix(const ScanWs &w, uint32_t tile, uint32_t *sh) {
  if (threadIdx.x == 0) *sh = 0;
  __syncthreads();
  uint32_t part = 0;
  for (uint32_t j = threadIdx.x; j < tile; j += blockDim.x) {
    uint32_t spins = 0;
    for (;;) {
      const unsigned long long word = __hip_atomic_load(&w.desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t tag = (uint32_t)(word >> 32);
      if ((tag >> 2) == w.gen && (tag & 3u) != 0) {
        part += (uint32_t)word;
        break;
      }
      if (++spins > (1u << 22)) {  // seconds: terminate with a wrong prefix rather than hang the GPU
        if (w.error) *w.error = 1u;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, kWave);
  if (lane_id() == 0 && part) atomicAdd(sh, part);
  __syncthreads();
  const uint32_t r = *sh;
  __syncthreads();
  return r;
}

__device__ __forceinline__ uint32_t scan_lookback(const ScanWs &w, uint32_t tile, uint32_t aggregate, uint32_t *sh) {
  scan_publish_aggregate(w, tile, aggregate);
  return scan_prefix(w, tile, sh);
}

// polls of one descriptor before a waiter starts helping: ~1 us per poll (load round trip + sleep) -- an order of
// magnitude above the longest wait of a healthy launch (predecessors publish within microseconds of starting), so the
// helping path only runs when predecessors are not resident
constexpr uint32_t kScanHelpAfterPolls = 192;

__device__ __forceinline__ bool scan_desc_ready(const ScanWs &w, uint32_t j, uint32_t *value) {
  const unsigned long long word = __hip_atomic_load(&w.desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t tag = (uint32_t)(word >> 32);
  *value = (uint32_t)word;
  return (tag >> 2) == w.gen && (tag & 3u) != 0;
}

// scan_prefix that cannot starve.  `agg(m)` recomputes the aggregate of tile m: called by ALL threads of the workgroup
// together (it may contain barriers), returns the value to every thread, must not disturb LDS the caller still needs,
// and must be a pure function of inputs that do not change while the kernel runs -- or of inputs that only change
// AFTER tile m has published (then `acquire_recheck` makes the published value win, see ht_count_assign_kernel).
// `sh` = two LDS words.  Fast path (every predecessor publishes within the poll budget): identical to scan_prefix.
template <typename AggFn>
__device__ __forceinline__ uint32_t scan_prefix_help(const ScanWs &w, uint32_t tile, uint32_t *sh, AggFn agg,
                                                     bool acquire_recheck = false) {
  if (threadIdx.x == 0) {
    sh[0] = 0;
    sh[1] = 0;
  }
  __syncthreads();
  uint32_t part = 0;
  bool gave_up = false;
  for (uint32_t j = threadIdx.x; j < tile && !gave_up; j += blockDim.x) {
    uint32_t spins = 0, v;
    for (;;) {
      if (scan_desc_ready(w, j, &v)) {
        part += v;
        break;
      }
      if (++spins > w.help_after) {
        gave_up = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (gave_up) sh[1] = 1u;  // benign race: every writer stores 1
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, kWave);
  if (lane_id() == 0 && part) atomicAdd(&sh[0], part);
  __syncthreads();
  uint32_t r = sh[0];
  const bool help = sh[1] != 0;
  __syncthreads();
  if (!help) return r;
  // ---- helping path (rare): start over; whatever is published is taken, whatever is not is recomputed here --------
  __shared__ uint32_t s_missing[256];
  __shared__ uint32_t s_nmiss, s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  part = 0;
  for (uint32_t base = 0; base < tile; base += 256) {
    if (threadIdx.x == 0) s_nmiss = 0;
    __syncthreads();
    const uint32_t j = base + threadIdx.x;
    if (threadIdx.x < 256 && j < tile) {
      uint32_t v;
      if (scan_desc_ready(w, j, &v)) part += v;
      else s_missing[atomicAdd(&s_nmiss, 1u)] = j;
    }
    __syncthreads();
    const uint32_t nm = s_nmiss;
    for (uint32_t q = 0; q < nm; ++q) {
      const uint32_t m = s_missing[q];  // uniform across the workgroup
      uint32_t a = agg(m);
      if (threadIdx.x == 0) {
        uint32_t v;
        if (acquire_recheck) __atomic_thread_fence(__ATOMIC_ACQUIRE);  // inputs read above, descriptor read below
        if (scan_desc_ready(w, m, &v)) a = v;                // the tile itself got there meanwhile: its word rules
        else scan_publish(w, m, kScanAggregate, a);          // same value the tile will store: spares other waiters
        part += a;
        if (w.helps) atomicAdd(w.helps, 1ull);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, kWave);
  if (lane_id() == 0 && part) atomicAdd(&s_sum, part);
  __syncthreads();
  r = s_sum;
  __syncthreads();
  return r;
}

__device__ __forceinline__ size_t resolve_count(size_t n_host, const uint32_t *d_n, size_t cap) {
  size_t n = d_n ? (size_t)(*d_n) : n_host;
  return n < cap ? n : cap;
}
__device__ __forceinline__ size_t resolve_count64(size_t n_host, const size_t *d_n, size_t cap) {
  size_t n = d_n ? *d_n : n_host;
  return n < cap ? n : cap;
}

// Exclusive scan of `n` block sums by ONE workgroup of 1024 threads, in place (defined in
// scan.hip).  total -> *total64 and *total32 (either may be null).  If accum != null the kernel also
// does *accum_out = *accum + total (used to advance the hash table's item count on the device).
// d_items32 / d_items64 (optional): device count of the items the sums were computed from, items_per_sum
// of them per entry -- lets the scan skip the capacity padding.
int launch_scan_block_sums(uint32_t *sums, size_t n, size_t *total64, uint32_t *total32,
                           const uint32_t *accum, uint32_t *accum_out, hipStream_t stream,
                           const uint32_t *d_items32 = nullptr, uint32_t items_per_sum = 0,
                           const size_t *d_items64 = nullptr);

inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// Stable sort of n (key, value) uint32 pairs by key, ascending (defined in scan.hip): one launch up to 8 k pairs, five
// up to 65 k, twelve beyond -- grids of independent workgroups.  keys_alt/vals_alt: n words each; ws:
// sort_pairs_ws_words(n) words.  The sorted pairs are in *sorted_keys / *sorted_vals afterwards -- one of the two buffer
// pairs, the other is clobbered.
size_t sort_pairs_ws_words(size_t n);
int launch_sort_pairs_u32(uint32_t *keys, uint32_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                          uint32_t *ws, hipStream_t stream, uint32_t **sorted_keys, uint32_t **sorted_vals);

// compute units of the current device (256 on MI355X)
inline int device_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// host side of ScanWs: owns the descriptors and hands out generations
struct ScanWsHost {
  ScanWs ws{nullptr, nullptr, 0, 0, nullptr, nullptr, 0, kScanHelpAfterPolls, nullptr};
  int create(size_t max_tiles);
  void destroy();
  // descriptor view for the next launch of `grid` workgroups (all of which call scan_take_tile)
  // kind: which kernel family the launch belongs to (0 sampler, 1 dedup count+assign, 2 cache split) -- selects
  // the section of the diagnostic phase log
  // use_ticket: the kernel draws its tiles with scan_take_tile (hash_dedup_kernel); otherwise tile = blockIdx.x and
  // waiters help (scan_prefix_help)
  // A ticketed launch that the runtime refused never drew its tickets: the caller hands them back (unnext_tickets).
  void unnext_tickets(size_t grid) { ws.ticket_base -= (uint32_t)grid; }
  ScanWs next(uint32_t kind, size_t grid, bool use_ticket = false) {
    if (ws.gen >= 0x3FFFFFFEu) {
      // generations are about to repeat (once per 2^30 launches): descriptors stamped during the previous cycle
      // must not read as fresh, so start the new cycle from zeroed descriptors, fenced against everything in flight
      (void)hipDeviceSynchronize();
      (void)hipMemset(ws.desc, 0, (size_t)ws.max_tiles * sizeof(unsigned long long));
      (void)hipDeviceSynchronize();
      ws.gen = 0;
    }
    ws.gen += 1u;
    ScanWs v = ws;
    // (the A/B switch that put tickets into EVERY single-pass kernel, profiles/r02_ticket_ab.txt, is gone: kernels that
    // take tile = blockIdx.x never draw from the counter, so advancing the base for them left base and counter out of
    // step for the next ticketed launch on the slot)
    // fgnn_debug_set_scan_help_after(polls): 0 makes every wait that is not satisfied at once take the helping path
    // (tests/test_coresidency_gpu.py)
    const int help_after = scan_help_after_override();
    if (help_after >= 0) v.help_after = (uint32_t)help_after;
    v.helps = scan_help_counter();
    if (use_ticket) ws.ticket_base += (uint32_t)grid;  // wraps with the 32-bit device counter
    else v.ticket = nullptr;
    if (uint32_t *sink = scan_error_sink()) v.error = sink;
    unsigned long long *log = phase_log_base();
    v.log = log ? log + (size_t)(kind % kPhaseLogKinds) * kPhaseLogTiles * 8 : nullptr;
    return v;
  }
};

// ---- remap fix-up as a tail of another launch ---------------------------------------------------------------------
// The entries of a layer's remapped edge list that ht_count_assign_kernel could not resolve (duplicates inside the fill:
// pend|owner's item index) are resolved from the owner's own entry -- ht_map_fix_kernel's job, a few microseconds of
// work that used to cost a launch per layer.  In the batch driver it rides along as extra workgroups of a launch that
// follows anyway (the next fill's first dedup kernel, the cache split): `blocks` workgroups behind the host kernel's
// own grid walk the list with a stride.  mapped == null: no tail.
struct FixTail {
  uint32_t *mapped;
  const size_t *d_n;   // device count (null: n_host)
  size_t n_host, cap;
  uint32_t pend;       // the table's pending flag (HtView::pend)
  uint32_t blocks;     // workgroups of kBlock threads appended to the host kernel's grid
};
inline FixTail no_fix_tail() { return FixTail{nullptr, nullptr, 0, 0, 0, 0}; }
// one hop, except after a resolving insert: the item pointed at may itself have lost the key later and then points on
// (its entry may be mid-update by its own lane -- either state leads to the owner); every hop leads to a strictly
// smaller item index (a key is only ever taken over by an earlier item), so the walk ends at the owner
__device__ __forceinline__ void map_fix_item(uint32_t *mapped, uint32_t pend, size_t i) {
  uint32_t m = mapped[i];
  if ((m & pend) && m != FGNN_EMPTY_KEY) {
    uint32_t at = (uint32_t)i;
    while ((m & pend) && m != FGNN_EMPTY_KEY) {
      const uint32_t j = m & (pend - 1u);
      if (j >= at) break;  // cannot happen with consistent notes; never loop on garbage
      at = j;
      m = mapped[j];
    }
    mapped[i] = m;
  }
}
// called by the workgroups blockIdx.x >= first_block of a host kernel launched with first_block + tail.blocks workgroups
__device__ __forceinline__ void run_fix_tail(const FixTail &t, uint32_t first_block) {
  size_t n = t.d_n ? *t.d_n : t.n_host;
  if (n > t.cap) n = t.cap;
  const size_t stride = (size_t)t.blocks * blockDim.x;
  for (size_t i = (size_t)(blockIdx.x - first_block) * blockDim.x + threadIdx.x; i < n; i += stride)
    map_fix_item(t.mapped, t.pend, i);
}
inline uint32_t fix_tail_blocks(size_t cap) {
  const size_t b = (cap + kBlock - 1) / kBlock;
  return (uint32_t)(b < 512 ? b : 512);
}

struct ScanWsHost;
// where the dedup's last pass leaves the sizes of the layer it just closed (all device pointers, nullable)
struct LayerSummary {
  uint32_t *num_dst;    // #items in the table before the fill  (TrainGraph::num_dst)
  uint32_t *num_src;    // #items after the fill                (TrainGraph::num_src)
  uint32_t *num_total;  // same value again (the batch's running num_input)
};
int hashtable_fill_duplicates_ex(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                 const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped, void *ws,
                                 size_t ws_bytes, void *stream, LayerSummary summary, bool already_inserted,
                                 ScanWsHost *scan, bool final_fill = false, bool resolved = false,
                                 FixTail *owed_fix = nullptr, const FixTail *carry_fix = nullptr,
                                 bool table_free = false);
// table_free: nothing after this fill reads the global table (no fused sampler insert, no fgnn_hashtable_map, and every
// later fill of the batch is table_free or final too) -- the fill may then go through the partitioned path although it
// is not the batch's last (the batch driver's samplers that do not insert themselves: every layer).  The promise is
// per BATCH: the caller checks hashtable_can_partition for every layer's capacity first (a non-final table_free fill
// that cannot be partitioned is refused, FGNN_EINVAL)
bool hashtable_can_partition(const fgnn_hashtable *ht, size_t cap);
// owed_fix (non-null: the caller takes the remap fix-up of THIS fill over; filled in, mapped == null if none is owed):
// the fix-up only rewrites `mapped` entries from other `mapped` entries -- nothing of the next layer's sampling reads
// it, so the batch driver lets it ride on a later launch (FixTail) instead of giving it a launch of its own.
// carry_fix (non-null, mapped != null): an EARLIER fill's owed fix-up; appended to this fill's insert launch when the
// call launches one (already_inserted == false), launched on its own otherwise -- handled either way.
int hashtable_map_fix(const FixTail &fix, void *stream);  // an owed fix-up as a launch of its own
// hashtable_partition.hip: pass 1 of a batch's LAST fill without the global table -- keys partitioned by hash, every
// bin deduplicated by one workgroup in LDS; pos[i] receives the exact outcome of item i (kPartIsOwner: first occurrence,
// pend|j: duplicate of item j, else the local id the node already had).  Also sets d_num_items[1] like every pass 1.
// pos[i] == kPartIsOwner stands for pend|i (item i is a first occurrence: the common case is marked in item order by the
// scatter pass, only the others are scattered by the dedup pass)
constexpr uint32_t kPartIsOwner = 0xFFFFFFFEu;
PartWs *partition_create(size_t max_items, size_t max_fill_items);
void partition_destroy(PartWs *w);
bool partition_fits(const PartWs *w, const fgnn_hashtable *ht, size_t cap);
int partition_fill(PartWs *w, const fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                   const size_t *d_num_items, size_t cap, uint32_t *pos, hipStream_t s, const FixTail &carry);
// can the last fill of `cap` items go through the resolving insert (disp[] allocated, one-launch count+assign)?
bool hashtable_can_resolve(const fgnn_hashtable *ht, size_t cap);
// Reset as the batch driver uses it: generation bump (wipe only on wrap), optionally without touching the counts
int hashtable_next_generation(fgnn_hashtable *ht, void *stream, bool zero_counts);
// small jobs of a batch that ride along with its feature gather launch (gather_rows16_kernel): the label rows of the
// seeds and the copy of the batch summary into pinned host memory
struct GatherTail {
  void *label_out;              // null: no label job
  const void *label_src;
  const uint32_t *label_index;  // output_nodes
  uint32_t num_label, label_esz;
  uint32_t *meta_dst;           // null: no summary copy (else host-mapped, 4-byte words)
  const uint32_t *meta_src;
  uint32_t meta_words;          // <= 256
  // null, or u64[2 * kGatherStampBlocks + 1] in pinned host memory: workgroup b posts its start / end clock (100 MHz) at
  // [2b], [2b + 1], workgroup 0 the grid size at [2 * kGatherStampBlocks] -- the launch's own duration without events
  unsigned long long *stamps = nullptr;
};
constexpr uint32_t kGatherStampBlocks = 4096;  // (the gather's grid is at most 4-6 workgroups per CU)
// can this gather carry a tail (16-byte row path, one launch)?
bool gather_takes_tail(const void *out, const void *src, size_t n_cap, size_t dim, int dtype);
int gather_rows_ex(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index, size_t n,
                   const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, uint32_t src_row_mask, void *stream,
                   const GatherTail *tail, size_t host_grid = 0, size_t wg_per_cu = 0);
// The trainer-side extraction of one batch as ONE launch (extract_fused_kernel, cache_gather.hip): CombineMissData with
// the host row fetch fused in + CombineCacheData + the label rows + the summary copy + (engine) the word copies that
// take a message's arrays out of its queue slot.  link_wgs > 0: the first link_wgs workgroups pull the miss rows (their
// source is host memory behind the host link) while the others stream the hit rows from the HBM cache; 0: every
// workgroup takes its share of both lists (miss source in HBM).
struct ExtractJob {
  void *out;
  const void *miss_rows, *cache_rows;              // full table (indexed by miss_src & miss_mask) / cache (by cache_src)
  const uint32_t *miss_src, *miss_dst, *cache_src, *cache_dst;
  size_t num_miss, num_cache;                      // host counts, used when d_counts == null
  const uint32_t *d_counts;                        // device: {num_miss, num_cache}
  size_t cap;                                      // capacity of either list under device counts
  size_t dim;
  int dtype;
  uint32_t miss_mask;
  GatherTail tail;
  const fgnn_copy_segment *segs;                   // host array of word copies (null: none)
  int num_segs;
  size_t link_wgs;                                 // workgroups of the link band (0: no band)
  size_t wg_per_cu;                                // persistent workgroups per CU of the HBM band (0: default 4)
  unsigned long long *stamps;                      // null, or u64[2 * grid]: every workgroup's start / end wall clock
};
// can the batch take the one-launch path (16-byte row chunks, 32-bit chunk index)?
bool extract_can_fuse(const ExtractJob &j);
int extract_fused(const ExtractJob &j, void *stream, size_t *grid_out = nullptr);
size_t extract_fused_grid(const ExtractJob &j, size_t *link = nullptr);  // workgroups of that launch (of its link band)
bool pointer_is_host(const void *p);             // host memory the GPU reads over the host link?
// workgroups of a host-source gather on a GPU that also runs the sampling chain (see gather_rows_ex)
constexpr size_t kSharedGpuHostGrid = 64;
// persistent workgroups per CU of the batch driver's HBM feature gather (the stateless entry points keep 4, which is
// what the kernel alone likes best: 59 us against 61.5 for a papers100M batch).  With other batches' sampling chains
// beside it, 3 leaves them the wave slots and memory requests they need: whole path 0.1067 -> 0.1027 ms per batch at a
// gather that takes 85.6 instead of 82.4 us (twitter / uk shapes -1 % / -2 %; 2 per CU: 0.1055 and a 113 us gather) --
// profiles/r04_h_gather_wg_per_cu.txt
constexpr size_t kSharedGpuGatherWgPerCu = 3;
// fgnn_get_miss_cache_index with look-back descriptors for the one-launch path (scan == null: three launches)
// carry_fix: an owed remap fix-up; rides on the one-launch split, launched on its own on the three-launch path
int get_miss_cache_index_ex(const uint32_t *table, const uint32_t *nodes, size_t num_nodes,
                            const uint32_t *d_num_nodes, size_t num_nodes_cap, uint32_t *miss_src, uint32_t *miss_dst,
                            uint32_t *cache_src, uint32_t *cache_dst, uint32_t *d_counts, void *ws, size_t ws_bytes,
                            void *stream, ScanWsHost *scan, unsigned long long *stamp = nullptr,
                            const FixTail *carry_fix = nullptr);
// fgnn_sample_weighted_khop_hash_dedup with the slot's look-back descriptors (scan == null: descriptors in ws)
int sample_hash_dedup(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                      const uint32_t *alias_table, const uint32_t *input, size_t num_input, const uint32_t *d_num_input,
                      size_t num_input_cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out,
                      int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream, ScanWsHost *scan_host);
// 5-ary search trees over the long rows of a prefix-sum table (prefix_tree.hip): tree_off[row] = node index of the
// row's root in `pool` (4 floats per node), FGNN_EMPTY_KEY for rows without a tree (short, or not non-decreasing)
constexpr uint32_t kPrefixTreeMinLen = 64;
struct PrefixTreeView {
  const uint32_t *tree_off;  // null: no trees
  const float *pool;
  // one 16-byte record per node, {row offset, row length, tree root (or FGNN_EMPTY_KEY), bits of the row's last prefix
  // value}: what a draw needs before its search, from ONE line instead of three (indptr pair, tree_off, the row's
  // last entry) and one round trip earlier.  null: not built
  const uint4 *rec;
};
struct PrefixTreeHost;
PrefixTreeHost *prefix_tree_build(const uint32_t *indptr, const float *prefix, size_t num_node);  // synchronous
void prefix_tree_destroy(PrefixTreeHost *t);
PrefixTreeView prefix_tree_view(const PrefixTreeHost *t);
void prefix_tree_stats(const PrefixTreeHost *t, size_t out[3]);  // long rows, rows refused (not monotone), pool bytes

// What the FIRST sampler launch of a batch does on top of sampling, so that no separate start-of-batch kernel is
// needed: FillWithUnique(seeds) (seed i -> local id i), copy of the seeds (the batch's output_nodes), item counts and
// the batch summary's header.  n2o == null: not the first launch.
struct BatchStart {
  uint32_t *n2o;
  uint32_t *items_copy;
  fgnn_batch_meta *meta;
  uint64_t key;
  uint32_t num_layers;
  uint32_t layer;  // this launch's layer: its num_edge entry is written by the launch itself, not by the header init
};
// the three with-replacement samplers (khop1 / weighted_khop / weighted_khop_prefix by sample_type; table_f = prob or
// prefix table) for callers that know the number of graph nodes, guarantee unique seeds and own a RankWs: the seed
// order then comes from a bitmap over the id space instead of a radix sort (sample_weighted.hip) -- no library call.
// rank == null: as the C entry points (scan.hip's sort orders the seeds).  Scratch: fgnn_weighted_scratch_bytes.
struct RankWs {
  uint32_t *bitmap;   // rank_ws_bytes(num_node) bytes, ALL ZERO between calls (the call restores that): bitmap | pre | sums
  ScanWsHost *scan;   // look-back descriptors for the two single-pass launches
};
size_t rank_ws_bytes(size_t num_node);
int sample_with_replacement_ex(int sample_type, const uint32_t *indptr, const uint32_t *indices, const float *table_f,
                               const uint32_t *alias, const uint32_t *input, size_t num_input,
                               const uint32_t *d_num_input, size_t num_input_cap, size_t fanout, uint32_t *out_src,
                               uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                               uint32_t layer, void *ws, size_t ws_bytes, void *stream, size_t num_node,
                               const RankWs *rank, PrefixTreeView tree = PrefixTreeView{nullptr, nullptr, nullptr});
// fgnn_sample_random_walk with look-back descriptors: the edge offsets and the compacted output come from one launch
// (scan == null: per-workgroup sums -> scan -> emit, as the C entry point)
int sample_random_walk_ex(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input, size_t num_input,
                          const uint32_t *d_num_input, size_t num_input_cap, size_t walk_len, double restart_prob,
                          size_t num_walks, size_t K, uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data,
                          size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws,
                          size_t ws_bytes, void *stream, ScanWsHost *scan);
// k-hop sampling with the dedup insert fused into the sampler (the engine's path): as fgnn_sample_khop0/2
// with FGNN_SRC_LOCAL, and every emitted edge e is inserted into `ht` with value PENDING|e; its bucket goes
// to ws[e] (the pos[] array hashtable_fill_duplicates_ex(already_inserted = true) expects at ws).
int sample_khop_fused(bool khop2, const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                      const uint32_t *d_num_input, size_t cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                      size_t *d_num_out, uint64_t seed, uint64_t batch_key, uint32_t layer, fgnn_hashtable *ht,
                      void *ws, size_t ws_bytes, void *stream, ScanWsHost *scan, const BatchStart *start = nullptr,
                      bool resolve = false);
int sample_khop_plain(bool khop2, const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t num_input,
                      const uint32_t *d_num_input, size_t cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                      size_t *d_num_out, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream, ScanWsHost *scan);
// resolve: the fill is the batch's last (hashtable_fill_duplicates_ex(..., final_fill, resolved = true) must follow):
// ws[e] receives the insert's OUTCOME (ht_insert_resolve) instead of the bucket.  Needs ht->disp.

}  // namespace fgnn
