// sample_weighted.hip -- weighted neighbour sampling with replacement via per-row prefix sums.
//
// Replaces GPUSampleWeightedKHopPrefix (reference samgraph/common/cuda/
// cuda_sampling_weighted_khop_prefix.cu:41-255): every (seed, slot) draws x = U(0,1] * rowsum and
// binary-searches the row's inclusive prefix-sum table; the reference then radix-sorts ALL
// num_input*fanout (src,dst) pairs by src and drops ADJACENT duplicates.  Bit-identical to oracle
// fgnn_oracle_sample_weighted_khop_prefix.
//
// MI355X design: seeds are unique, so the stable sort of the pairs by src is just the seeds' groups
// reordered by seed id -- only the num_input seeds are sorted (the radix sort of scan.hip, fanout x fewer
// keys than the reference sorts), the draws are one-lane-per-draw (all 64 lanes busy, independent
// binary searches in flight), adjacent-duplicate removal is a per-seed count known before the
// compaction, and the offsets come from a scan in sorted-seed order.
// Where the number of graph nodes is known (batch driver) the seeds are not sorted at all: they are unique node ids, so
// a seed's position in id order is the number of seed bits below its own in a bitmap over the id space -- set bits,
// popcount prefix per word, one lookup per seed -- ~5 small launches instead of four radix passes
// (183 -> ~80 us for 1.3 M seeds on the twitter shape), and independent of how the ids are distributed.
#include <cstring>

#include "fgnn_device.h"

namespace fgnn {
namespace {

__device__ __forceinline__ float uniform_float(uint32_t x) {
  return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1], 24 bits, as the oracle
}

// Position of the first entry >= x in a non-decreasing row with a 5-ary tree (prefix_tree.hip): what the reference's
// binary search (weighted_khop_prefix.cu:66-86) returns on such a row, in ceil(log5 len) 16-byte look-ups.
// x <= the row's last entry.
__device__ __forceinline__ uint32_t tree_search(uint32_t len, const float *pool, uint32_t root, float x) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 *nodes = reinterpret_cast<const f4 *>(pool);
  uint32_t T = 0;
  for (uint32_t m = len; m > 1; m = (m + 4u) / 5u) ++T;  // smallest T with 5^T >= len
  uint32_t idx = 0, base = root;
  for (uint32_t l = T; l >= 1; --l) {
    const f4 nd = nodes[base + idx];  // +inf marks an empty child: never below x
    idx = idx * 5u + (uint32_t)(nd.x < x) + (uint32_t)(nd.y < x) + (uint32_t)(nd.z < x) + (uint32_t)(nd.w < x);
    uint32_t n = len;  // nodes of level l = ceil(len / 5^l)
    for (uint32_t k = 0; k < l; ++k) n = (n + 4u) / 5u;
    base += n;
  }
  return idx;  // level 1's separators are the row's own entries: the descent ends at the position
}

// the draw of (seed i, slot j).  MODE 0: per-row prefix sums + binary search (weighted_khop_prefix.cu:41-92),
// 1: uniform with replacement (khop1.cu:42-72), 2: alias method, alias table = node ids (weighted_khop.cu:41-76)
template <int MODE>
__device__ __forceinline__ uint32_t weighted_pick(const uint32_t *__restrict__ indices, const float *__restrict__ prefix,
                                                  const uint32_t *__restrict__ alias, uint32_t off, uint32_t len,
                                                  uint32_t i, uint32_t j, uint64_t seed, uint64_t batch_key,
                                                  uint32_t tag, const float *tree_pool = nullptr,
                                                  uint32_t tree_root = FGNN_EMPTY_KEY, const float *up_known = nullptr) {
  if (len == 0) return FGNN_EMPTY_KEY;
  if (MODE == 1) return indices[off + philox_u32(seed, batch_key, tag, i, j) % len];
  if (MODE == 2) {
    const u32x4 blk = philox_block(seed, batch_key, tag, i, j >> 1);  // draws 2j, 2j+1 share a block
    const uint32_t r0 = (j & 1u) ? blk.z : blk.x;
    const uint32_t r1 = (j & 1u) ? blk.w : blk.y;
    const uint32_t k = r0 % len;
    return uniform_float(r1) < prefix[off + k] ? indices[off + k] : alias[off + k];
  }
  const float up = up_known ? *up_known : prefix[off + len - 1];  // (the node record carries the row's last entry)
  const float x = uniform_float(philox_u32(seed, batch_key, tag, i, j)) * up;
  if (tree_root != FGNN_EMPTY_KEY) return indices[off + tree_search(len, tree_pool, tree_root, x)];
  if (x <= prefix[off]) return indices[off];
  size_t lo = off, hi = (size_t)off + len - 1;
  while (hi - lo >= 2) {
    const size_t mid = (lo + hi) >> 1;
    if (prefix[mid] >= x) hi = mid; else lo = mid;
  }
  return indices[hi];
}

// one lane per (seed, slot)
template <int MODE>
__global__ __launch_bounds__(kBlock) void weighted_draw_kernel(const uint32_t *__restrict__ indptr,
                                                               const uint32_t *__restrict__ indices,
                                                               const float *__restrict__ prefix,
                                                               const uint32_t *__restrict__ alias,
                                                               const uint32_t *__restrict__ input, size_t n_host,
                                                               const uint32_t *d_n, size_t cap, uint32_t F,
                                                               uint32_t *__restrict__ tmp_dst, uint64_t seed,
                                                               uint64_t batch_key, uint32_t tag, PrefixTreeView tree) {
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t total = n * F;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const size_t i = t / F;
    const uint32_t j = (uint32_t)(t - i * F);
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    const uint32_t root = (MODE == 0 && tree.tree_off && len > kPrefixTreeMinLen) ? tree.tree_off[rid] : FGNN_EMPTY_KEY;
    tmp_dst[t] = weighted_pick<MODE>(indices, prefix, alias, off, len, (uint32_t)i, j, seed, batch_key, tag, tree.pool,
                                     root);
  }
}

// weighted_draw_kernel + weighted_count_kernel in one launch for fan-outs up to 64: a seed's F draws sit in F consecutive
// lanes of ONE wavefront (64 / F seeds per wave), so "is this draw equal to the seed's next one" is a lane shuffle and
// the per-seed count a ballot -- no second pass over the draws.  Also covers the padding up to `cap` like the count
// kernel (keys / counts / order of unused seed slots).
template <int MODE>
__global__ __launch_bounds__(kBlock) void weighted_draw_count_kernel(
    const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices, const float *__restrict__ prefix,
    const uint32_t *__restrict__ alias, const uint32_t *__restrict__ input, size_t n_host, const uint32_t *d_n,
    size_t cap, uint32_t F, uint32_t *__restrict__ tmp_dst, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
    uint32_t *__restrict__ cnt, uint32_t *bitmap, uint32_t *__restrict__ order, uint64_t seed, uint64_t batch_key,
    uint32_t tag, PrefixTreeView tree) {
  const size_t n = resolve_count(n_host, d_n, cap);
  const uint32_t G = (uint32_t)kWave / F;               // seeds per wave (F <= 64)
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t g = lane / F, j = lane - g * F;
  const bool lane_used = g < G;
  const unsigned long long gmask = lane_used ? (((F == 64u) ? ~0ull : ((1ull << F) - 1ull)) << (g * F)) : 0ull;
  const size_t waves = (size_t)gridDim.x * kWavesPerBlock;
  for (size_t w = (size_t)blockIdx.x * kWavesPerBlock + wave_id(); w * G < cap; w += waves) {
    const size_t i = w * G + g;
    const bool seed_here = lane_used && i < cap;
    uint32_t rid = FGNN_EMPTY_KEY, off = 0, len = 0, root = FGNN_EMPTY_KEY;
    float up = 0.f;
    const bool rec = MODE == 0 && tree.rec != nullptr;
    if (seed_here && i < n) {
      rid = input[i];
      if (rec) {  // row bounds, tree root and row sum from ONE line (three lines and one more round trip without)
        const uint4 rc = tree.rec[rid];
        off = rc.x;
        len = rc.y;
        root = rc.z;
        up = __uint_as_float(rc.w);
      } else {
        off = indptr[rid];
        len = indptr[rid + 1] - off;
        // (the root's index is fetched with the row bounds: no extra round trip; the F lanes of a seed share the address)
        if (MODE == 0 && tree.tree_off) root = tree.tree_off[rid];
        if (len <= kPrefixTreeMinLen) root = FGNN_EMPTY_KEY;
      }
    }
    uint32_t pick = FGNN_EMPTY_KEY;
    if (seed_here && i < n) {
      pick = weighted_pick<MODE>(indices, prefix, alias, off, len, (uint32_t)i, j, seed, batch_key, tag, tree.pool, root,
                                 rec ? &up : nullptr);
      tmp_dst[i * F + j] = pick;
    }
    // a draw is dropped when it equals the seed's NEXT draw; the last one is always kept (count_edge, prefix.cu:94-112)
    const uint32_t nxt = __shfl_down(pick, 1, kWave);
    const bool keep = seed_here && i < n && len != 0 && (j + 1 == F || pick != nxt);
    const uint32_t c = (uint32_t)__popcll(__ballot(keep) & gmask);
    if (seed_here && j == 0) {
      const uint32_t key = (i < n && len != 0) ? rid : FGNN_EMPTY_KEY;
      keys[i] = key;
      vals[i] = (uint32_t)i;
      cnt[i] = c;
      if (bitmap) {
        order[i] = FGNN_EMPTY_KEY;
        if (key != FGNN_EMPTY_KEY) atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
      }
    }
  }
}

// one lane per seed: sort key (seed id, kEmptyKey for empty rows / padding) and #edges kept after
// removing a draw equal to the NEXT draw of the same seed (count_edge, prefix.cu:94-112)
__global__ __launch_bounds__(kBlock) void weighted_count_kernel(const uint32_t *__restrict__ indptr,
                                                                const uint32_t *__restrict__ input, size_t n_host,
                                                                const uint32_t *d_n, size_t cap, uint32_t F,
                                                                const uint32_t *__restrict__ tmp_dst,
                                                                uint32_t *__restrict__ keys,
                                                                uint32_t *__restrict__ vals,
                                                                uint32_t *__restrict__ cnt, uint32_t *bitmap,
                                                                uint32_t *__restrict__ order) {
  // bitmap != null (batch driver): this launch is also the first step of the seed ranking -- one bit per seed id
  // into a bitmap that is all zero on entry, order[] starts as "no seed at this rank"
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  uint32_t key = FGNN_EMPTY_KEY, c = 0;
  if (i < n) {
    const uint32_t rid = input[i];
    if (indptr[rid + 1] != indptr[rid]) {
      key = rid;
      const uint32_t *d = tmp_dst + i * F;
      uint32_t prev = d[0];
      for (uint32_t j = 1; j < F; ++j) {
        const uint32_t cur = d[j];
        c += (prev != cur);
        prev = cur;
      }
      c += 1;  // the last draw of a seed is always kept
    }
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
  cnt[i] = c;
  if (bitmap) {
    order[i] = FGNN_EMPTY_KEY;
    if (key != FGNN_EMPTY_KEY) atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
  }
}

// per-workgroup sums of cnt in sorted-seed order
__global__ __launch_bounds__(kBlock) void weighted_sorted_sums_kernel(const uint32_t *__restrict__ order,
                                                                      const uint32_t *__restrict__ cnt, size_t cap,
                                                                      uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t oi = r < cap ? order[r] : FGNN_EMPTY_KEY;  // EMPTY: no seed at this rank (bitmap path)
  const uint32_t c = oi != FGNN_EMPTY_KEY ? cnt[oi] : 0u;
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(c, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void weighted_emit_kernel(const uint32_t *__restrict__ input,
                                                               const uint32_t *__restrict__ order,
                                                               const uint32_t *__restrict__ cnt, size_t cap,
                                                               uint32_t F, const uint32_t *__restrict__ tmp_dst,
                                                               const uint32_t *__restrict__ block_offsets,
                                                               uint32_t *__restrict__ out_src,
                                                               uint32_t *__restrict__ out_dst, int src_mode) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t i = 0, c = 0;
  if (r < cap) {
    i = order[r];
    c = i != FGNN_EMPTY_KEY ? cnt[i] : 0u;
  }
  uint32_t tot;
  const uint32_t lo = block_exclusive_scan<kWavesPerBlock>(c, sh, &tot);
  if (c == 0) return;
  size_t w = (size_t)block_offsets[blockIdx.x] + lo;
  const uint32_t src = src_mode == FGNN_SRC_LOCAL ? i : input[i];
  const uint32_t *d = tmp_dst + (size_t)i * F;
  uint32_t cur = d[0];
  for (uint32_t j = 0; j < F; ++j) {
    const uint32_t nxt = (j + 1 < F) ? d[j + 1] : ~cur;  // ~cur != cur: last one always kept
    if (cur != nxt) {
      out_src[w] = src;
      out_dst[w] = cur;
      ++w;
    }
    cur = nxt;
  }
}

// ---- seed order without a sort (unique node ids, known id range) -------------------------------------------------
constexpr int kWordsPerThread = 16;
constexpr int kWordsPerBlock = kBlock * kWordsPerThread;  // bitmap words one workgroup of the popcount passes covers

constexpr size_t kSinglePassTiles = 1536;  // grids whose workgroups sum all their predecessors' aggregates (fgnn_device.h)

// back to all zero (only where the single-pass emit kernel, which does this on the way, cannot be used)
__global__ __launch_bounds__(kBlock) void rank_clear_kernel(const uint32_t *__restrict__ keys, size_t cap,
                                                            uint32_t *bitmap) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const uint32_t key = keys[i];
  if (key != FGNN_EMPTY_KEY) bitmap[key >> 5] = 0u;
}

// mode 0: block_sums[b] = set bits in the workgroup's words; mode 1: pre[w] = set bits before word w (block_sums
// holds the scanned workgroup offsets)
template <int MODE>
__global__ __launch_bounds__(kBlock) void rank_popcount_kernel(const uint32_t *__restrict__ bitmap, size_t words,
                                                               uint32_t *__restrict__ block_sums,
                                                               uint32_t *__restrict__ pre) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t w0 = (size_t)blockIdx.x * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
  uint32_t local[kWordsPerThread];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kWordsPerThread; ++k) {
    local[k] = w0 + k < words ? (uint32_t)__popc(bitmap[w0 + k]) : 0u;
    sum += local[k];
  }
  uint32_t tot;
  const uint32_t ex = block_exclusive_scan<kWavesPerBlock>(sum, sh, &tot);
  if (MODE == 0) {
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
  } else {
    uint32_t run = block_sums[blockIdx.x] + ex;
#pragma unroll
    for (int k = 0; k < kWordsPerThread; ++k) {
      if (w0 + k < words) pre[w0 + k] = run;
      run += local[k];
    }
  }
}

// order[rank of seed i in id order] = i
__global__ __launch_bounds__(kBlock) void rank_scatter_kernel(const uint32_t *__restrict__ keys, size_t cap,
                                                              const uint32_t *__restrict__ bitmap,
                                                              const uint32_t *__restrict__ pre,
                                                              uint32_t *__restrict__ order) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const uint32_t key = keys[i];
  if (key == FGNN_EMPTY_KEY) return;
  const uint32_t w = key >> 5, b = key & 31u;
  // a rank beyond the seed count means the bitmap held bits that are not this call's (a failed earlier call on the
  // slot): never write outside order[]
  const size_t r = (size_t)pre[w] + (uint32_t)__popc(bitmap[w] & ((1u << b) - 1u));
  if (r < cap) order[r] = (uint32_t)i;
}

// pre[w] = set bits before word w in ONE launch: a workgroup popcounts its kWordsPerBlock words, publishes the sum and
// adds up what the workgroups before it published (fgnn_device.h, single-pass prefix; a waiter that outlasts its
// poll budget recounts the missing tile from the bitmap, which nothing writes while this kernel runs).
// Word order inside a tile is (round, lane, k): four rounds of 256 lanes x 4 words, so that every load and store is
// 16 bytes per lane, consecutive across lanes (the first version gave each lane 16 consecutive words: 64 cache lines
// per wave instruction, 18 us for twitter's 5 MB bitmap).
struct alignas(16) u32v4 { uint32_t x, y, z, w; };
constexpr int kRankRounds = kWordsPerBlock / (kBlock * 4);  // 4

__global__ __launch_bounds__(kBlock) void rank_prefix_kernel(const uint32_t *__restrict__ bitmap, size_t words,
                                                             uint32_t *__restrict__ pre, ScanWs scan) {
  __shared__ uint32_t sh[kRankRounds][kWavesPerBlock];
  __shared__ uint32_t sh_scan[kWavesPerBlock];
  __shared__ uint32_t sh_tile[2];
  const uint32_t tile = blockIdx.x;
  // popcounts of this lane's 4 words in each round, packed one per byte (<= 32 each); words is padded to a multiple of
  // 4 by the allocation (rank_ws_bytes), so whole 16-byte groups are either inside or outside
  auto load_tile = [&](uint32_t tl, uint32_t *packed) -> uint32_t {
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < kRankRounds; ++r) {
      const size_t w0 = (size_t)tl * kWordsPerBlock + (size_t)r * (kBlock * 4) + (size_t)threadIdx.x * 4;
      uint32_t pk = 0;
      if (w0 < words) {
        const u32v4 v = *reinterpret_cast<const u32v4 *>(bitmap + w0);
        pk = (uint32_t)__popc(v.x) | ((uint32_t)__popc(v.y) << 8) | ((uint32_t)__popc(v.z) << 16) |
             ((uint32_t)__popc(v.w) << 24);
      }
      if (packed) packed[r] = pk;
      sum += (pk & 0xffu) + ((pk >> 8) & 0xffu) + ((pk >> 16) & 0xffu) + (pk >> 24);
    }
    return sum;
  };
  uint32_t packed[kRankRounds];
  const uint32_t mine = load_tile(tile, packed);
  // per-round inclusive scans inside the wave; wave totals per round to LDS; one barrier
  uint32_t inc[kRankRounds], tr[kRankRounds];
#pragma unroll
  for (int r = 0; r < kRankRounds; ++r) {
    const uint32_t pk = packed[r];
    tr[r] = (pk & 0xffu) + ((pk >> 8) & 0xffu) + ((pk >> 16) & 0xffu) + (pk >> 24);
    inc[r] = wave_inclusive_scan(tr[r]);
    if (lane_id() == kWave - 1) sh[r][wave_id()] = inc[r];
  }
  __syncthreads();
  uint32_t tot = 0, base[kRankRounds];
#pragma unroll
  for (int r = 0; r < kRankRounds; ++r) {
    base[r] = tot;
#pragma unroll
    for (int wv = 0; wv < kWavesPerBlock; ++wv) {
      const uint32_t t = sh[r][wv];
      if (wv < wave_id()) base[r] += t;
      tot += t;
    }
    base[r] += inc[r] - tr[r];
  }
  (void)mine;
  scan_publish_aggregate(scan, tile, tot);
  const uint32_t before = scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
    uint32_t tot_m;
    (void)block_exclusive_scan<kWavesPerBlock>(load_tile(m, nullptr), sh_scan, &tot_m);
    return tot_m;
  });
#pragma unroll
  for (int r = 0; r < kRankRounds; ++r) {
    const size_t w0 = (size_t)tile * kWordsPerBlock + (size_t)r * (kBlock * 4) + (size_t)threadIdx.x * 4;
    if (w0 < words) {
      const uint32_t pk = packed[r];
      u32v4 o;
      o.x = before + base[r];
      o.y = o.x + (pk & 0xffu);
      o.z = o.y + ((pk >> 8) & 0xffu);
      o.w = o.z + ((pk >> 16) & 0xffu);
      *reinterpret_cast<u32v4 *>(pre + w0) = o;
    }
  }
}

// weighted_sorted_sums_kernel -> scan -> weighted_emit_kernel in ONE launch: a workgroup owns 256 x IPT consecutive ranks
// (IPT consecutive ranks per lane), its output offset is the prefix over the earlier workgroups' edge counts.  Also
// restores the ranking bitmap to all zero for the next call (every seed clears its own word; the ranking is done).
template <int IPT>
__global__ __launch_bounds__(kBlock) void weighted_emit_sp_kernel(const uint32_t *__restrict__ input,
                                                                  const uint32_t *__restrict__ order,
                                                                  const uint32_t *__restrict__ cnt,
                                                                  const uint32_t *__restrict__ keys, size_t cap,
                                                                  uint32_t F, const uint32_t *__restrict__ tmp_dst,
                                                                  uint32_t *__restrict__ out_src,
                                                                  uint32_t *__restrict__ out_dst, int src_mode,
                                                                  ScanWs scan, size_t *d_num_out, uint32_t *bitmap) {
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t sh_tile[2];
  const uint32_t tile = blockIdx.x;
  auto tile_sum = [&](uint32_t tl, uint32_t *oi, uint32_t *c) -> uint32_t {
    const size_t r0 = ((size_t)tl * kBlock + threadIdx.x) * IPT;
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const uint32_t o = r0 + q < cap ? order[r0 + q] : FGNN_EMPTY_KEY;  // EMPTY: no seed at this rank
      const uint32_t cc = o != FGNN_EMPTY_KEY ? cnt[o] : 0u;
      if (oi) {
        oi[q] = o;
        c[q] = cc;
      }
      sum += cc;
    }
    return sum;
  };
  uint32_t oi[IPT], c[IPT];
  const uint32_t sum = tile_sum(tile, oi, c);
  uint32_t tot;
  const uint32_t lo = block_exclusive_scan<kWavesPerBlock>(sum, sh, &tot);
  scan_publish_aggregate(scan, tile, tot);
  const uint32_t before = scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
    uint32_t tot_m;
    (void)block_exclusive_scan<kWavesPerBlock>(tile_sum(m, nullptr, nullptr), sh, &tot_m);
    return tot_m;
  });
  if (tile == gridDim.x - 1 && threadIdx.x == 0 && d_num_out) *d_num_out = (size_t)before + tot;
  size_t w = (size_t)before + lo;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    if (c[q] == 0) continue;
    const uint32_t i = oi[q];
    if (bitmap) bitmap[keys[i] >> 5] = 0u;  // c != 0 <=> the seed has a key, i.e. a bit in the bitmap
    const uint32_t src = src_mode == FGNN_SRC_LOCAL ? i : input[i];
    const uint32_t *d = tmp_dst + (size_t)i * F;
    uint32_t cur = d[0];
    for (uint32_t j = 0; j < F; ++j) {
      const uint32_t nxt = (j + 1 < F) ? d[j + 1] : ~cur;  // ~cur != cur: last one always kept
      if (cur != nxt) {
        out_src[w] = src;
        out_dst[w] = cur;
        ++w;
      }
      cur = nxt;
    }
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

// scratch layout for cap seeds, fanout F (all uint32 unless noted):
//   tmp_dst[cap*F] | keys[cap] | vals[cap] | keys_out[cap] | order[cap] | cnt[cap] | sums[nb+1] | sort workspace
// bitmap words of the seed ranking, padded to whole 16-byte groups (rank_prefix_kernel moves four words per lane)
static size_t rank_words(size_t num_node) { return (fgnn::div_up(num_node, (size_t)32) + 3) & ~(size_t)3; }

size_t fgnn::rank_ws_bytes(size_t num_node) {
  const size_t words = rank_words(num_node);
  return (2 * words + fgnn::div_up(words, (size_t)fgnn::kWordsPerBlock) + 16) * sizeof(uint32_t);  // bitmap | pre | sums
}

extern "C" size_t fgnn_weighted_scratch_bytes(size_t num_input_cap, size_t fanout) {
  const size_t nb = div_up(num_input_cap, kBlock);
  return (num_input_cap * fanout + 5 * num_input_cap + nb + 8 + fgnn::sort_pairs_ws_words(num_input_cap)) *
             sizeof(uint32_t) + 256;
}

namespace fgnn {
namespace {

int launch_with_replacement(int mode, int sample_type, const uint32_t *indptr, const uint32_t *indices,
                            const float *table_f, const uint32_t *alias, const uint32_t *input, size_t num_input,
                            const uint32_t *d_num_input, size_t num_input_cap, size_t fanout, uint32_t *out_src,
                            uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                            uint32_t layer, void *ws, size_t ws_bytes, void *stream, size_t num_node = 0,
                            const RankWs *rank = nullptr, PrefixTreeView tree = PrefixTreeView{nullptr, nullptr, nullptr}) {
  // rank != null: the caller guarantees unique seeds below num_node (the batch driver's frontier) and owns an all-zero
  // bitmap over the id space: the seeds are ordered by counting bits, nothing is sorted and no library is called
  auto st = static_cast<hipStream_t>(stream);
  size_t cap = d_num_input ? num_input_cap : num_input;
  if (fanout == 0 || fanout > 0xffffu) return FGNN_EINVAL;
  if (cap == 0) {
    if (d_num_out) FGNN_HIP_CHECK(hipMemsetAsync(d_num_out, 0, sizeof(size_t), st));
    return FGNN_OK;
  }
  if (!indptr || !indices || !input || cap * fanout >= 0x7fffffffull) return FGNN_EINVAL;
  if ((mode != 1 && !table_f) || (mode == 2 && !alias)) return FGNN_EINVAL;
  if (ws_bytes < fgnn_weighted_scratch_bytes(cap, fanout)) return FGNN_ENOSPC;
  if (rank && (!rank->bitmap || !rank->scan || num_node == 0)) return FGNN_EINVAL;
  const uint32_t F = (uint32_t)fanout;
  const uint32_t tag = ((uint32_t)sample_type << 8) | (layer & 0xffu);
  const size_t nb = div_up(cap, kBlock);
  uint32_t *tmp_dst = static_cast<uint32_t *>(ws);
  uint32_t *keys = tmp_dst + cap * F;
  uint32_t *vals = keys + cap;
  uint32_t *keys_out = vals + cap;
  uint32_t *order = keys_out + cap;
  uint32_t *cnt = order + cap;
  uint32_t *sums = cnt + cap;
  uint32_t *sort_ws = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(sums + nb + 8) + 255) & ~uintptr_t(255));

  uint32_t *bitmap = rank ? rank->bitmap : nullptr;
  if (F <= (uint32_t)kWave) {
    // draws + per-seed counts + ranking bits in one launch
    const size_t seeds_per_wg = (size_t)kWavesPerBlock * ((size_t)kWave / F);
    size_t blocks = div_up(cap, seeds_per_wg);
    if (blocks > 256 * 32) blocks = 256 * 32;
#define FGNN_DRAWC(M)                                                                                            \
  hipLaunchKernelGGL((weighted_draw_count_kernel<M>), dim3(blocks), dim3(kBlock), 0, st, indptr, indices, table_f, \
                     alias, input, num_input, d_num_input, cap, F, tmp_dst, keys, vals, cnt, bitmap, order, seed,    \
                     batch_key, tag, tree)
    if (mode == 1) FGNN_DRAWC(1);
    else if (mode == 2) FGNN_DRAWC(2);
    else FGNN_DRAWC(0);
#undef FGNN_DRAWC
  } else {
    size_t blocks = div_up(cap * F, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
#define FGNN_DRAW(M)                                                                                              \
  hipLaunchKernelGGL((weighted_draw_kernel<M>), dim3(blocks), dim3(kBlock), 0, st, indptr, indices, table_f, alias, \
                     input, num_input, d_num_input, cap, F, tmp_dst, seed, batch_key, tag, tree)
    if (mode == 1) FGNN_DRAW(1);
    else if (mode == 2) FGNN_DRAW(2);
    else FGNN_DRAW(0);
#undef FGNN_DRAW
    hipLaunchKernelGGL(weighted_count_kernel, dim3(nb), dim3(kBlock), 0, st, indptr, input, num_input, d_num_input, cap,
                       F, tmp_dst, keys, vals, cnt, bitmap, order);
  }
  if (rank) {
    // order by counting bits below each seed id (see the file header): 5 launches per layer in all
    const size_t words = rank_words(num_node);
    const size_t nb2 = div_up(words, (size_t)kWordsPerBlock);
    uint32_t *pre = bitmap + words;
    if (nb2 <= kSinglePassTiles && nb2 <= rank->scan->ws.max_tiles) {
      hipLaunchKernelGGL(rank_prefix_kernel, dim3(nb2), dim3(kBlock), 0, st, bitmap, words, pre,
                         rank->scan->next(0, nb2));
    } else {  // graphs beyond ~200 M nodes: per-workgroup sums -> scan -> prefixes
      uint32_t *sums2 = pre + words;
      hipLaunchKernelGGL((rank_popcount_kernel<0>), dim3(nb2), dim3(kBlock), 0, st, bitmap, words, sums2, pre);
      if (launch_scan_block_sums(sums2, nb2, nullptr, nullptr, nullptr, nullptr, st) != FGNN_OK) {
        (void)hipMemsetAsync(bitmap, 0, words * sizeof(uint32_t), st);  // the contract: all zero between calls
        return FGNN_EHIP;
      }
      hipLaunchKernelGGL((rank_popcount_kernel<1>), dim3(nb2), dim3(kBlock), 0, st, bitmap, words, sums2, pre);
    }
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(nb), dim3(kBlock), 0, st, keys, cap, bitmap, pre, order);
    const int ipt = nb <= kSinglePassTiles ? 1 : nb <= 4 * kSinglePassTiles ? 4 : nb <= 16 * kSinglePassTiles ? 16 : 0;
    const size_t grid = ipt ? div_up(cap, (size_t)kBlock * ipt) : 0;
    if (ipt && grid <= rank->scan->ws.max_tiles) {
#define FGNN_EMIT(I)                                                                                                  \
  hipLaunchKernelGGL((weighted_emit_sp_kernel<I>), dim3(grid), dim3(kBlock), 0, st, input, order, cnt, keys, cap, F, \
                     tmp_dst, out_src, out_dst, src_mode, rank->scan->next(0, grid), d_num_out, bitmap)
      if (ipt == 1) FGNN_EMIT(1);
      else if (ipt == 4) FGNN_EMIT(4);
      else FGNN_EMIT(16);
#undef FGNN_EMIT
      const int rc = launch_status(__func__);
      // the emit kernel is what clears the bits it consumed: if a launch of this call was refused, wipe them here
      if (rc != FGNN_OK) (void)hipMemsetAsync(bitmap, 0, words * sizeof(uint32_t), st);
      return rc;
    }
    hipLaunchKernelGGL(rank_clear_kernel, dim3(nb), dim3(kBlock), 0, st, keys, cap, bitmap);
  } else {
    // seeds of unknown range: sorted with their positions; `order` = the positions in seed-id order afterwards
    if (launch_sort_pairs_u32(keys, keys_out, vals, order, cap, sort_ws, st, nullptr, &order) != FGNN_OK) return FGNN_EHIP;
  }
  hipLaunchKernelGGL(weighted_sorted_sums_kernel, dim3(nb), dim3(kBlock), 0, st, order, cnt, cap, sums);
  if (launch_scan_block_sums(sums, nb, d_num_out, nullptr, nullptr, nullptr, st) != FGNN_OK) return FGNN_EHIP;
  hipLaunchKernelGGL(weighted_emit_kernel, dim3(nb), dim3(kBlock), 0, st, input, order, cnt, cap, F, tmp_dst, sums,
                     out_src, out_dst, src_mode);
  return launch_status(__func__);
}

}  // namespace
}  // namespace fgnn

int fgnn::sample_with_replacement_ex(int sample_type, const uint32_t *indptr, const uint32_t *indices,
                                     const float *table_f, const uint32_t *alias, const uint32_t *input,
                                     size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                     uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                                     uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                                     void *stream, size_t num_node, const RankWs *rank, PrefixTreeView tree) {
  const int mode = sample_type == FGNN_KHOP1 ? 1 : sample_type == FGNN_WEIGHTED_KHOP ? 2 : 0;
  return launch_with_replacement(mode, sample_type, indptr, indices, table_f, alias, input, num_input, d_num_input,
                                 num_input_cap, fanout, out_src, out_dst, d_num_out, src_mode, seed, batch_key, layer,
                                 ws, ws_bytes, stream, num_node, rank, mode == 0 ? tree : PrefixTreeView{nullptr, nullptr, nullptr});
}

extern "C" int fgnn_sample_weighted_khop_prefix(const uint32_t *indptr, const uint32_t *indices, const float *prefix,
                                                const uint32_t *input, size_t num_input,
                                                const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                                uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                                                uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws,
                                                size_t ws_bytes, void *stream) {
  return fgnn::launch_with_replacement(0, FGNN_WEIGHTED_KHOP_PREFIX, indptr, indices, prefix, nullptr, input, num_input,
                                       d_num_input, num_input_cap, fanout, out_src, out_dst, d_num_out, src_mode, seed,
                                       batch_key, layer, ws, ws_bytes, stream);
}

extern "C" int fgnn_sample_khop1(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                 size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                 uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed,
                                 uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes, void *stream) {
  return fgnn::launch_with_replacement(1, FGNN_KHOP1, indptr, indices, nullptr, nullptr, input, num_input, d_num_input,
                                       num_input_cap, fanout, out_src, out_dst, d_num_out, src_mode, seed, batch_key,
                                       layer, ws, ws_bytes, stream);
}

extern "C" int fgnn_sample_weighted_khop(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                                         const uint32_t *alias_table, const uint32_t *input, size_t num_input,
                                         const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                         uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                                         uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                                         void *stream) {
  return fgnn::launch_with_replacement(2, FGNN_WEIGHTED_KHOP, indptr, indices, prob_table, alias_table, input,
                                       num_input, d_num_input, num_input_cap, fanout, out_src, out_dst, d_num_out,
                                       src_mode, seed, batch_key, layer, ws, ws_bytes, stream);
}
