// scan.hip -- single-workgroup exclusive scan of per-block partial sums (<= a few 100k entries); the single-pass scan's
// workspace; a stable radix sort of uint32 pairs.
// Replaces cub::DeviceScan::ExclusiveSum at the reference's call sites
// (cuda_sampling_khop2.cu:221-229, cuda_hashtable.cu:757-768, cuda_cache.cu:193-205).
#include "fgnn_device.h"

namespace fgnn {

constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(uint32_t *sums, size_t n, size_t *total64,
                                                                    uint32_t *total32, const uint32_t *accum,
                                                                    uint32_t *accum_out, const uint32_t *d_items32,
                                                                    const size_t *d_items64, uint32_t items_per_sum) {
  __shared__ uint32_t sh[kScanBlock / kWave];
  // grids are sized by capacity; only the first ceil(items / items_per_sum) sums can be non-zero
  if (d_items32 || d_items64) {
    const size_t items = d_items32 ? (size_t)*d_items32 : *d_items64;
    const size_t live = (items + items_per_sum - 1) / items_per_sum;
    if (live < n) n = live;
  }
  uint32_t carry = 0;
  for (size_t base = 0; base < n; base += kScanBlock) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = i < n ? sums[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<kScanBlock / kWave>(v, sh, &tot);
    if (i < n) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    if (total64) *total64 = carry;
    if (total32) *total32 = carry;
    if (accum_out) *accum_out = (accum ? *accum : 0u) + carry;
  }
}

int ScanWsHost::create(size_t max_tiles) {
  if (max_tiles == 0) max_tiles = 1;
  if (hipMalloc(&ws.desc, max_tiles * sizeof(unsigned long long)) != hipSuccess) return FGNN_EHIP;
  if (hipMalloc(&ws.error, 2 * sizeof(uint32_t)) != hipSuccess) return FGNN_EHIP;  // {error, ticket counter}
  ws.ticket = ws.error + 1;
  ws.ticket_base = 0;
  ws.max_tiles = (uint32_t)max_tiles;
  ws.gen = 0;
  // generation 0 never matches a launch (generations start at 1)
  if (hipMemset(ws.desc, 0, max_tiles * sizeof(unsigned long long)) != hipSuccess) return FGNN_EHIP;
  if (hipMemset(ws.error, 0, 2 * sizeof(uint32_t)) != hipSuccess) return FGNN_EHIP;
  return FGNN_OK;
}

void ScanWsHost::destroy() {
  if (ws.desc) (void)hipFree(ws.desc);
  if (ws.error) (void)hipFree(ws.error);
  ws.desc = nullptr;
  ws.error = nullptr;
  ws.ticket = nullptr;
}

int launch_scan_block_sums(uint32_t *sums, size_t n, size_t *total64, uint32_t *total32, const uint32_t *accum,
                           uint32_t *accum_out, hipStream_t stream, const uint32_t *d_items32, uint32_t items_per_sum,
                           const size_t *d_items64) {
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(kScanBlock), 0, stream, sums, n, total64, total32, accum,
                     accum_out, d_items32, d_items64, items_per_sum ? items_per_sum : 1u);
  return launch_status(__func__);
}

// ---- stable LSD radix sort of (key, value) uint32 pairs ----------------------------------------------------------
// Replaces cub::DeviceRadixSort::SortPairs where the reference orders a layer's seeds
// (cuda_sampling_weighted_khop_prefix.cu:200-215) for callers that cannot give the id range (the stateless sampler entry
// points; the batch driver ranks its seeds through a bitmap and sorts nothing).  Four passes of 8 bits; per pass: tile
// histograms (digit-major) -> one workgroup per digit scans its row of tile counts -> tiles scatter with a stable
// in-tile rank (ballot matching inside a wave, wave counts through LDS).  Every step is a grid of independent
// workgroups: no look-back, nothing to wait for.
// The launches are most of the time of a small sort, so: up to 65 k pairs a pass is ONE launch over tiles of 256 pairs
// (the scatter sums the few tile counts itself and counts the next pass's digits as it places the pairs: 5 launches),
// and up to kCountSortMax pairs (a mini-batch's seeds) the order is found by counting -- a pair's position is the number
// of pairs that sort before it -- in one launch.
// Whole stateless weighted call, us, this sort | rocPRIM's: 8000 seeds 35 | 42, 22500: 57 | 53, 200 k: 94 | 100,
// 1.3 M: 347 | 388 (tools/probe/stateless_weighted_time.py).
constexpr int kSortBlock = 256;
constexpr int kSortWaves = kSortBlock / kWave;
constexpr int kSortRounds = 8;  // a tile = 2048 pairs ...
constexpr int kSortTile = kSortBlock * kSortRounds;
constexpr int kSortRadix = 256;
constexpr size_t kFusedTiles = 256;  // ... except in the fused passes (<= 65536 pairs): 256 pairs, more workgroups
static_assert(kSortRadix == kSortBlock, "one thread per digit in the histogram and offset steps");

template <int ROUNDS>
__global__ __launch_bounds__(kSortBlock) void sort_hist_kernel(const uint32_t *__restrict__ keys, size_t n,
                                                              uint32_t shift, uint32_t *__restrict__ hist,
                                                              size_t tiles, uint32_t *__restrict__ later,
                                                              int num_later) {
  __shared__ uint32_t h[kSortRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t t0 = (size_t)blockIdx.x * (ROUNDS * kSortBlock);
  uint32_t k[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // all loads in flight before the first count
    const size_t i = t0 + (size_t)r * kSortBlock + threadIdx.x;
    k[r] = i < n ? keys[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const size_t i = t0 + (size_t)r * kSortBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(k[r] >> shift) & (kSortRadix - 1)], 1u);
  }
  __syncthreads();
  if (num_later == 0) {
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];  // digit-major: a digit's row is scanned next
    return;
  }
  // fused passes: tile-major (the scatter's threads read the rows of the tiles before theirs side by side), and the
  // histograms the scatter steps will count into start at zero -- this tile's row of each
  const size_t at = (size_t)blockIdx.x * kSortRadix + threadIdx.x;
  hist[at] = h[threadIdx.x];
  for (int p = 0; p < num_later; ++p) later[(size_t)p * kSortRadix * tiles + at] = 0u;
}

// workgroup d: exclusive scan of digit d's tile counts in place, the digit's total to totals[d]
__global__ __launch_bounds__(kSortBlock) void sort_rowscan_kernel(uint32_t *__restrict__ hist, size_t tiles,
                                                                 uint32_t *__restrict__ totals) {
  __shared__ uint32_t sh[kSortWaves];
  uint32_t *row = hist + (size_t)blockIdx.x * tiles;
  uint32_t carry = 0;
  for (size_t base = 0; base < tiles; base += kSortBlock) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = i < tiles ? row[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<kSortWaves>(v, sh, &tot);
    if (i < tiles) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// FUSED (<= kFusedTiles tiles of 256): the tile sums its digits' rows itself instead of reading a scanned histogram,
// and counts the NEXT pass's digit of every pair into the histogram of the tile the pair lands in: one launch per pass.
template <bool FUSED, int ROUNDS>
__global__ __launch_bounds__(kSortBlock) void sort_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                 const uint32_t *__restrict__ vals_in,
                                                                 uint32_t *__restrict__ keys_out,
                                                                 uint32_t *__restrict__ vals_out, size_t n,
                                                                 uint32_t shift, const uint32_t *__restrict__ hist,
                                                                 size_t tiles,
                                                                 const uint32_t *__restrict__ totals,
                                                                 uint32_t *__restrict__ next_hist) {
  __shared__ uint32_t base[kSortRadix];  // where this tile's next item of digit d goes
  __shared__ uint32_t wcnt[kSortWaves][kSortRadix];
  __shared__ uint32_t sh[kSortWaves];
  uint32_t row_total, row_before;
  if (FUSED) {
    row_total = row_before = 0;
#pragma unroll 8
    for (uint32_t t = 0; t < (uint32_t)tiles; ++t) {  // <= kFusedTiles rows of 1 KB, tile-major
      const uint32_t c = hist[(size_t)t * kSortRadix + threadIdx.x];
      row_total += c;
      row_before += t < blockIdx.x ? c : 0u;
    }
  } else {
    row_total = totals[threadIdx.x];
    row_before = hist[(size_t)threadIdx.x * tiles + blockIdx.x];
  }
  uint32_t tot;
  const uint32_t below = block_exclusive_scan<kSortWaves>(row_total, sh, &tot);  // items of smaller digits
  base[threadIdx.x] = below + row_before;
  const size_t t0 = (size_t)blockIdx.x * (ROUNDS * kSortBlock);
  const uint32_t wave = wave_id(), lane = lane_id();
  uint32_t ks[ROUNDS], vs[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // the tile's pairs are fetched at once: one round trip, not one per round
    const size_t i = t0 + (size_t)r * kSortBlock + threadIdx.x;
    ks[r] = i < n ? keys_in[i] : 0u;
    vs[r] = i < n ? vals_in[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const size_t r0 = t0 + (size_t)r * kSortBlock;
    if (r0 >= n) break;  // the same for the whole workgroup
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();  // (also orders base[] of the previous round / the prologue before its readers)
    const size_t i = r0 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = ks[r];
    const uint32_t d = (key >> shift) & (kSortRadix - 1);
    // lanes of this wave holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t off = base[d] + rank;
      for (uint32_t w = 0; w < wave; ++w) off += wcnt[w][d];
      keys_out[off] = key;
      vals_out[off] = vs[r];
      if (FUSED && next_hist) {
        const uint32_t next_digit = (key >> (shift + 8u)) & (kSortRadix - 1);
        atomicAdd(&next_hist[(size_t)(off / (ROUNDS * kSortBlock)) * kSortRadix + next_digit], 1u);
      }
    }
    __syncthreads();
    uint32_t add = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) add += wcnt[w][threadIdx.x];
    base[threadIdx.x] += add;
  }
}

// ---- the same order for few pairs: position = #{j : (key[j], j) < (key[i], i)} -----------------------------------
constexpr size_t kCountSortMax = 8192;  // n^2 comparisons: a few us here, ~50 us at 22 k (the fused passes take ~25)
constexpr int kCountLanes = 16;          // lanes that share one pair, each counting a 16th of the others
constexpr int kCountPairs = kSortBlock / kCountLanes;
constexpr int kCountTile = 2048;

__global__ __launch_bounds__(kSortBlock) void sort_by_counting_kernel(const uint32_t *__restrict__ keys,
                                                                     const uint32_t *__restrict__ vals,
                                                                     uint32_t *__restrict__ keys_out,
                                                                     uint32_t *__restrict__ vals_out, uint32_t n) {
  __shared__ uint32_t tile[kCountTile];
  const uint32_t s = threadIdx.x % kCountLanes;
  const uint32_t i0 = blockIdx.x * kCountPairs, i = i0 + threadIdx.x / kCountLanes;
  const bool have = i < n;
  const uint32_t my = have ? keys[i] : 0u;
  uint32_t before = 0;
  for (uint32_t base = 0; base < n; base += kCountTile) {
    const uint32_t m = n - base < (uint32_t)kCountTile ? n - base : (uint32_t)kCountTile;
    for (uint32_t t = threadIdx.x; t < m; t += kSortBlock) tile[t] = keys[base + t];
    __syncthreads();
    if (base + m <= i0) {  // every j of the tile is below this workgroup's pairs: ties sort before
      for (uint32_t j = s; j < m; j += kCountLanes) before += tile[j] <= my;
    } else if (base >= i0 + kCountPairs) {  // every j above: ties sort after
      for (uint32_t j = s; j < m; j += kCountLanes) before += tile[j] < my;
    } else {
      for (uint32_t j = s; j < m; j += kCountLanes) {
        const uint32_t k = tile[j];
        before += (base + j < i) ? (k <= my) : (k < my);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int off = kCountLanes / 2; off >= 1; off >>= 1) before += __shfl_xor(before, off, kWave);
  if (have && s == 0) {
    keys_out[before] = my;
    vals_out[before] = vals[i];
  }
}

size_t sort_pairs_ws_words(size_t n) {
  if (n == 0) n = 1;
  const size_t small = div_up(n, (size_t)kSortBlock);
  if (small <= kFusedTiles) return (size_t)kSortRadix * small * 4;
  return (size_t)kSortRadix * div_up(n, (size_t)kSortTile) + kSortRadix;
}

int launch_sort_pairs_u32(uint32_t *keys, uint32_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                          uint32_t *ws, hipStream_t stream, uint32_t **sorted_keys, uint32_t **sorted_vals) {
  if (sorted_keys) *sorted_keys = keys;
  if (sorted_vals) *sorted_vals = vals;
  if (n == 0) return FGNN_OK;
  if (!keys || !keys_alt || !vals || !vals_alt || !ws || n >= 0xffffffffull) return FGNN_EINVAL;
  if (n <= kCountSortMax) {
    hipLaunchKernelGGL(sort_by_counting_kernel, dim3(div_up(n, (size_t)kCountPairs)), dim3(kSortBlock), 0, stream, keys,
                       vals, keys_alt, vals_alt, (uint32_t)n);
    if (sorted_keys) *sorted_keys = keys_alt;
    if (sorted_vals) *sorted_vals = vals_alt;
    return launch_status(__func__);
  }
  if (div_up(n, (size_t)kSortBlock) <= kFusedTiles) {  // one histogram per pass, all four prepared by the first launch
    const size_t small = div_up(n, (size_t)kSortBlock), hw = (size_t)kSortRadix * small;
    hipLaunchKernelGGL((sort_hist_kernel<1>), dim3(small), dim3(kSortBlock), 0, stream, keys, n, 0u, ws, small, ws + hw,
                       3);
    for (uint32_t pass = 0; pass < 4; ++pass) {
      hipLaunchKernelGGL((sort_scatter_kernel<true, 1>), dim3(small), dim3(kSortBlock), 0, stream, keys, vals, keys_alt,
                         vals_alt, n, 8u * pass, ws + pass * hw, small, (const uint32_t *)nullptr,
                         pass < 3 ? ws + (pass + 1) * hw : (uint32_t *)nullptr);
      uint32_t *t = keys; keys = keys_alt; keys_alt = t;
      t = vals; vals = vals_alt; vals_alt = t;
    }
    return launch_status(__func__);
  }
  const size_t tiles = div_up(n, (size_t)kSortTile);
  uint32_t *hist = ws, *totals = ws + (size_t)kSortRadix * tiles;
  for (uint32_t pass = 0; pass < 4; ++pass) {  // an even number of passes: the result is back in keys / vals
    const uint32_t shift = 8u * pass;
    hipLaunchKernelGGL((sort_hist_kernel<kSortRounds>), dim3(tiles), dim3(kSortBlock), 0, stream, keys, n, shift, hist,
                       tiles, (uint32_t *)nullptr, 0);
    hipLaunchKernelGGL(sort_rowscan_kernel, dim3(kSortRadix), dim3(kSortBlock), 0, stream, hist, tiles, totals);
    hipLaunchKernelGGL((sort_scatter_kernel<false, kSortRounds>), dim3(tiles), dim3(kSortBlock), 0, stream, keys, vals,
                       keys_alt, vals_alt, n, shift, hist, tiles, totals, (uint32_t *)nullptr);
    uint32_t *t = keys; keys = keys_alt; keys_alt = t;
    t = vals; vals = vals_alt; vals_alt = t;
  }
  return launch_status(__func__);
}

}  // namespace fgnn
