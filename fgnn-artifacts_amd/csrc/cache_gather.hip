// cache_gather.hip -- cache hit/miss index split and row gathers for gfx950.
//
// Replaces GetMissCacheIndex (reference samgraph/common/cuda/cuda_cache.cu:33-234, duplicated in
// cuda_cache_manager_device.cu:38-163,266-337), GPUExtract (cuda_extraction.cu:30-117) and
// CombineMissData / CombineCacheData (cuda_cache_manager_device.cu:165-210,339-442;
// dist/dist_cache_manager_device.cu:38-183).
//
// MI355X design
//  * split: the reference reads table[nodes[i]] five times per node over three kernels and two
//    device scans.  Here: one kernel reads it once, keeps the slot in scratch and emits per-workgroup
//    miss counts; a one-workgroup scan; one kernel writes BOTH output lists (the hit position follows
//    from the miss rank: hits_before = items_before - misses_before), ranks by wave ballots;
//  * gathers: rows are moved as 16-byte chunks by a flat chunk index (chunk c of row r), so every
//    lane of a wave is busy for any feature width (D=100 -> 25 chunks, D=128 -> 32, D=256 -> 64) and
//    several independent 16-byte loads are in flight per lane; a scalar path covers rows whose byte
//    length is not a multiple of 16 (labels: dim 1 x 8 B).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fgnn_device.h"

namespace fgnn {
namespace {

template <int IPT>
__global__ __launch_bounds__(kBlock) void cache_count_kernel(const uint32_t *__restrict__ table,
                                                             const uint32_t *__restrict__ nodes, size_t n_host,
                                                             const uint32_t *d_n, size_t cap,
                                                             uint32_t *__restrict__ slot,
                                                             uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
  uint32_t miss = 0;
#pragma unroll
  for (int r = 0; r < IPT; ++r) {
    const size_t i = tile0 + (size_t)r * kBlock + threadIdx.x;
    if (i < n) {
      const uint32_t s = table[nodes[i]];
      slot[i] = s;
      miss += (s == FGNN_EMPTY_KEY);
    }
  }
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(miss, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

template <int IPT>
__global__ __launch_bounds__(kBlock) void cache_split_kernel(const uint32_t *__restrict__ nodes, size_t n_host,
                                                             const uint32_t *d_n, size_t cap,
                                                             const uint32_t *__restrict__ slot,
                                                             const uint32_t *__restrict__ block_offsets,
                                                             const uint32_t *__restrict__ d_total_miss,
                                                             uint32_t *__restrict__ miss_src,
                                                             uint32_t *__restrict__ miss_dst,
                                                             uint32_t *__restrict__ cache_src,
                                                             uint32_t *__restrict__ cache_dst,
                                                             uint32_t *__restrict__ d_counts) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t tile0 = (size_t)blockIdx.x * (kBlock * IPT);
  uint32_t miss_before = block_offsets[blockIdx.x];
  for (int r = 0; r < IPT; ++r) {
    const size_t row0 = tile0 + (size_t)r * kBlock;
    const size_t i = row0 + threadIdx.x;
    uint32_t s = 0;
    bool is_miss = false;
    if (i < n) {
      s = slot[i];
      is_miss = (s == FGNN_EMPTY_KEY);
    }
    uint32_t tot;
    const uint32_t mrank = block_exclusive_rank<kWavesPerBlock>(is_miss, sh, &tot);
    if (i < n) {
      if (is_miss) {
        const uint32_t p = miss_before + mrank;
        miss_dst[p] = (uint32_t)i;
        miss_src[p] = nodes[i];
      } else {
        // hits before i = items before i - misses before i
        const uint32_t p = (uint32_t)(row0 - miss_before) + (threadIdx.x - mrank);
        cache_dst[p] = (uint32_t)i;
        cache_src[p] = s;
      }
    }
    miss_before += tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t m = *d_total_miss;
    d_counts[0] = m;
    d_counts[1] = (uint32_t)n - m;
  }
}

// count + scan + split in ONE launch (batch driver path): a workgroup owns `rounds` x 256 consecutive nodes,
// looks every node up once, learns the number of misses before its chunk by a decoupled look-back over the
// earlier workgroups and writes both lists.  Same output as cache_count_kernel -> scan -> cache_split_kernel.
__global__ __launch_bounds__(kBlock) void cache_split_fused_kernel(const uint32_t *__restrict__ table,
                                                                   const uint32_t *__restrict__ nodes, size_t n_host,
                                                                   const uint32_t *d_n, size_t cap,
                                                                   uint32_t *__restrict__ slot,
                                                                   uint32_t *__restrict__ miss_src,
                                                                   uint32_t *__restrict__ miss_dst,
                                                                   uint32_t *__restrict__ cache_src,
                                                                   uint32_t *__restrict__ cache_dst,
                                                                   uint32_t *__restrict__ d_counts, ScanWs scan,
                                                                   unsigned long long *stamp, uint32_t own_blocks,
                                                                   FixTail fix FGNN_ABLATE_PARAM) {
  if (blockIdx.x >= own_blocks) {  // the last fill's remap fix-up riding along (FixTail, fgnn_device.h)
    run_fix_tail(fix, own_blocks);
    return;
  }
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t sh_tile[2];
  if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = wall_clock64();  // fgnn_batch_meta::t_sampled
  const uint32_t n = (uint32_t)resolve_count(n_host, d_n, cap);  // cap < 2^32 (host check)
  const uint32_t per_round = kBlock * own_blocks;
  const uint32_t rounds = n ? (n - 1) / per_round + 1 : 1u;  // <= 32 by the host's grid choice
  const uint32_t chunk = rounds * kBlock;
  const uint32_t ntiles = n ? (n - 1) / chunk + 1 : 1u;
  const uint32_t tile = scan_take_tile(scan, sh_tile);
  if (tile >= ntiles) return;
  phase_mark(scan, tile, 0);
  const size_t chunk0 = (size_t)tile * chunk;
  // misses among the nodes of tile `tl` (bit r of *mask: this thread's node of round r is one).  own = true: this
  // workgroup's tile -- the looked-up slots are kept for the split below.  own = false: another tile's count,
  // recomputed by a waiter that helps (scan_prefix_help): same reads, no writes.
  auto count_chunk = [&](uint32_t tl, bool own, uint32_t *mask) -> uint32_t {
    const size_t c0 = (size_t)tl * chunk;
    uint32_t mm = 0, cn = 0;
    // four rounds at a time: their loads are independent, keep them all in flight
    for (uint32_t r0 = 0; r0 < rounds; r0 += 4) {
      uint32_t nd[4], sv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = c0 + (size_t)(r0 + u) * kBlock + threadIdx.x;
        ok[u] = r0 + u < rounds && i < n;
        nd[u] = ok[u] ? nodes[i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = ok[u] && !(ablate & 1u) ? table[nd[u]] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          if (own && !(ablate & 2u)) slot[c0 + (size_t)(r0 + u) * kBlock + threadIdx.x] = sv[u];
          if (sv[u] == FGNN_EMPTY_KEY) {
            mm |= 1u << (r0 + u);
            ++cn;
          }
        }
      }
    }
    *mask = mm;
    return cn;
  };
  uint32_t miss_mask = 0;
  const uint32_t cnt = count_chunk(tile, true, &miss_mask);
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(cnt, sh, &tot);
  phase_mark(scan, tile, 1);
  scan_publish_aggregate(scan, tile, tot);
  uint32_t miss_before = (ablate & 8u) ? 0u : scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
    uint32_t mask_m, tot_m;
    const uint32_t cm = count_chunk(m, false, &mask_m);
    (void)block_exclusive_scan<kWavesPerBlock>(cm, sh, &tot_m);
    return tot_m;
  });
  phase_mark(scan, tile, 2);
  if (tile == ntiles - 1 && threadIdx.x == 0) {
    d_counts[0] = miss_before + tot;
    d_counts[1] = (uint32_t)n - (miss_before + tot);
  }
  for (uint32_t r = 0; r < rounds; ++r) {
    const size_t row0 = chunk0 + (size_t)r * kBlock;
    const size_t i = row0 + threadIdx.x;
    const bool is_miss = (miss_mask >> r) & 1u;
    uint32_t t2;
    const uint32_t mrank = block_exclusive_rank<kWavesPerBlock>(is_miss, sh, &t2);
    if (i < n && !(ablate & 4u)) {
      if (is_miss) {
        const uint32_t p = miss_before + mrank;
        miss_dst[p] = (uint32_t)i;
        miss_src[p] = nodes[i];
      } else {
        const uint32_t p = (uint32_t)(row0 - miss_before) + (threadIdx.x - mrank);
        cache_dst[p] = (uint32_t)i;
        cache_src[p] = slot[i];
      }
    }
    miss_before += t2;
  }
  phase_mark(scan, tile, 3);
}

struct alignas(16) chunk16 { uint32_t a, b, c, d; };

// flat 16-byte-chunk gather; chunks_per_row = row_bytes / 16.  A workgroup walks tiles of
// kBlock*UNROLL consecutive chunks (chunk c of the output = chunk c%cpr of row c/cpr).  The grid is capped
// at a few workgroups per CU: enough loads in flight to saturate HBM while leaving wave slots for the
// latency-bound sampling kernels of the next batch that run concurrently on another stream.
// CPR > 0 fixes chunks-per-row at compile time (the index division becomes a multiply/shift).
// `tail` (batch driver): the batch's label rows and its 128-byte summary ride along with the feature gather instead of
// taking a launch each (a 4 us element gather and a 4 us copyBuffer behind 6 us event gaps, per batch)
template <int UNROLL, int CPR, bool NT, bool NTS>
__global__ __launch_bounds__(kBlock) void gather_rows16_kernel(chunk16 *__restrict__ out,
                                                               const chunk16 *__restrict__ src,
                                                               const uint32_t *__restrict__ src_index,
                                                               const uint32_t *__restrict__ dst_index, size_t n_host,
                                                               const uint32_t *d_n, size_t cap, uint32_t cpr_rt,
                                                               uint32_t src_mask, GatherTail tail) {
  const uint32_t cpr = CPR ? (uint32_t)CPR : cpr_rt;
  const bool stamped = tail.stamps != nullptr && blockIdx.x < kGatherStampBlocks;
  if (stamped && threadIdx.x == 0) {
    tail.stamps[2 * blockIdx.x] = wall_clock64();
    if (blockIdx.x == 0) tail.stamps[2 * kGatherStampBlocks] = gridDim.x;
  }
  if (tail.label_out) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < tail.num_label; i += stride) {
      const size_t r = tail.label_index[i];
      switch (tail.label_esz) {
        case 1: static_cast<uint8_t *>(tail.label_out)[i] = static_cast<const uint8_t *>(tail.label_src)[r]; break;
        case 2: static_cast<uint16_t *>(tail.label_out)[i] = static_cast<const uint16_t *>(tail.label_src)[r]; break;
        case 4: static_cast<uint32_t *>(tail.label_out)[i] = static_cast<const uint32_t *>(tail.label_src)[r]; break;
        default:
          static_cast<unsigned long long *>(tail.label_out)[i] = static_cast<const unsigned long long *>(tail.label_src)[r];
      }
    }
  }
  // the summary is final before this launch starts (every kernel that writes it is earlier in the stream)
  if (tail.meta_dst && blockIdx.x == gridDim.x - 1 && threadIdx.x < tail.meta_words)
    tail.meta_dst[threadIdx.x] = tail.meta_src[threadIdx.x];
  const uint32_t n = (uint32_t)resolve_count(n_host, d_n, cap);
  const uint32_t total = n * cpr;  // host guarantees cap * cpr < 2^32
  constexpr uint32_t tile = kBlock * UNROLL;
  const uint32_t full = total / tile * tile;
  uint32_t tile0 = blockIdx.x * tile;
  for (; tile0 < full; tile0 += gridDim.x * tile) {
    chunk16 v[UNROLL];
    size_t dsts[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t cc = tile0 + u * kBlock + threadIdx.x;
      const uint32_t row = cc / cpr;
      const uint32_t col = cc - row * cpr;
      const size_t srow = (src_index ? src_index[row] : row) & src_mask;
      const size_t drow = dst_index ? dst_index[row] : row;
      const chunk16 *sp = src + srow * cpr + col;
      if (NT) {
        v[u].a = __builtin_nontemporal_load(&sp->a);
        v[u].b = __builtin_nontemporal_load(&sp->b);
        v[u].c = __builtin_nontemporal_load(&sp->c);
        v[u].d = __builtin_nontemporal_load(&sp->d);
      } else {
        v[u] = *sp;
      }
      dsts[u] = drow * cpr + col;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NTS) {
        chunk16 *dp = out + dsts[u];
        __builtin_nontemporal_store(v[u].a, &dp->a);
        __builtin_nontemporal_store(v[u].b, &dp->b);
        __builtin_nontemporal_store(v[u].c, &dp->c);
        __builtin_nontemporal_store(v[u].d, &dp->d);
      } else {
        out[dsts[u]] = v[u];
      }
    }
  }
  // ragged last tile (handled by the workgroup whose turn it is)
  if (tile0 == full && full < total) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t cc = full + u * kBlock + threadIdx.x;
      if (cc < total) {
        const uint32_t row = cc / cpr;
        const uint32_t col = cc - row * cpr;
        const size_t srow = (src_index ? src_index[row] : row) & src_mask;
        const size_t drow = dst_index ? dst_index[row] : row;
        out[drow * cpr + col] = src[srow * cpr + col];
      }
    }
  }
  if (stamped) {
    __syncthreads();
    if (threadIdx.x == 0) tail.stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

// ---- the one-launch trainer-side extraction (ExtractJob, fgnn_device.h) ----------------------------------------------
// One band of workgroups `bid` of `nb` walking a list's rows as flat 16-byte chunks: the loop of gather_rows16_kernel
// with both index arrays present, non-temporal loads and stores.
template <int UNROLL, int CPR>
__device__ __forceinline__ void gather_band(chunk16 *__restrict__ out, const chunk16 *__restrict__ src,
                                            const uint32_t *__restrict__ src_index,
                                            const uint32_t *__restrict__ dst_index, uint32_t n, uint32_t cpr_rt,
                                            uint32_t src_mask, uint32_t bid, uint32_t nb) {
  const uint32_t cpr = CPR ? (uint32_t)CPR : cpr_rt;
  const uint32_t total = n * cpr;  // host guarantees cap * cpr < 2^32
  constexpr uint32_t tile = kBlock * UNROLL;
  const uint32_t full = total / tile * tile;
  uint32_t tile0 = bid * tile;
  for (; tile0 < full; tile0 += nb * tile) {
    chunk16 v[UNROLL];
    size_t dsts[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t cc = tile0 + u * kBlock + threadIdx.x;
      const uint32_t row = cc / cpr;
      const uint32_t col = cc - row * cpr;
      const size_t srow = src_index[row] & src_mask;
      const chunk16 *sp = src + srow * cpr + col;
      v[u].a = __builtin_nontemporal_load(&sp->a);
      v[u].b = __builtin_nontemporal_load(&sp->b);
      v[u].c = __builtin_nontemporal_load(&sp->c);
      v[u].d = __builtin_nontemporal_load(&sp->d);
      dsts[u] = (size_t)dst_index[row] * cpr + col;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      chunk16 *dp = out + dsts[u];
      __builtin_nontemporal_store(v[u].a, &dp->a);
      __builtin_nontemporal_store(v[u].b, &dp->b);
      __builtin_nontemporal_store(v[u].c, &dp->c);
      __builtin_nontemporal_store(v[u].d, &dp->d);
    }
  }
  if (tile0 == full && full < total) {  // ragged last tile: the workgroup whose turn it is
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t cc = full + u * kBlock + threadIdx.x;
      if (cc < total) {
        const uint32_t row = cc / cpr;
        const uint32_t col = cc - row * cpr;
        out[(size_t)dst_index[row] * cpr + col] = src[(size_t)(src_index[row] & src_mask) * cpr + col];
      }
    }
  }
}

struct FusedArgs {
  chunk16 *out;
  const chunk16 *miss_rows, *cache_rows;
  const uint32_t *miss_src, *miss_dst, *cache_src, *cache_dst;
  size_t num_miss, num_cache;
  const uint32_t *d_counts;
  size_t cap;
  uint32_t cpr, miss_mask;
  uint32_t link_wgs;  // workgroups [0, link_wgs) are the link band
  GatherTail tail;
  unsigned long long *stamps;
  // the word copies (arrays of a received message leaving its queue slot): flat index space over all segments
  int num_segs;
  uint32_t *seg_dst[FGNN_MAX_COPY_SEGMENTS];
  const uint32_t *seg_src[FGNN_MAX_COPY_SEGMENTS];
  uint32_t seg_begin[FGNN_MAX_COPY_SEGMENTS + 1];
};

// UL / UH: independent 16-byte loads in flight per lane in the link band / the HBM band
template <int UL, int UH, int CPR>
__global__ __launch_bounds__(kBlock) void extract_fused_kernel(const FusedArgs a) {
  __shared__ uint32_t *s_dst[FGNN_MAX_COPY_SEGMENTS];
  __shared__ const uint32_t *s_src[FGNN_MAX_COPY_SEGMENTS];
  __shared__ uint32_t s_begin[FGNN_MAX_COPY_SEGMENTS + 1];
  const uint32_t n_miss = (uint32_t)resolve_count(a.num_miss, a.d_counts, a.cap);
  const uint32_t n_cache = (uint32_t)resolve_count(a.num_cache, a.d_counts ? a.d_counts + 1 : nullptr, a.cap);
  // the link band is small on purpose: 16 workgroups x 4 loads per lane keep 256 KB of host reads in flight, twice what
  // the link needs at its ~2 us latency; more of them only queue in front of everything else the chip has outstanding
  // (profiles/r06_a_link_band_sweep.txt: 56 GB/s with 16 workgroups, 49 with 256 or with a launch of its own).  Packing
  // the band onto one XCD (workgroups 0, 8, 16, ...) was measured too: no gain, the queueing is not per XCD
  const bool in_link = blockIdx.x < a.link_wgs;
  const uint32_t bid = in_link ? blockIdx.x : blockIdx.x - a.link_wgs;
  // stamps are indexed by band position: [0, link_wgs) the link band, then the HBM band
  const uint32_t stamp_at = in_link ? bid : a.link_wgs + bid;
  if (a.stamps && threadIdx.x == 0) a.stamps[2 * stamp_at] = wall_clock64();
  if (in_link) {
    // link band: rows come over the host link, a few workgroups with several loads in flight each keep it full
    gather_band<UL, CPR>(a.out, a.miss_rows, a.miss_src, a.miss_dst, n_miss, a.cpr, a.miss_mask, bid, a.link_wgs);
  } else {
    const uint32_t nb = gridDim.x - a.link_wgs;
    const GatherTail &tail = a.tail;
    if (tail.label_out) {
      const uint32_t stride = nb * kBlock;
      for (uint32_t i = bid * kBlock + threadIdx.x; i < tail.num_label; i += stride) {
        const size_t r = tail.label_index[i];
        switch (tail.label_esz) {
          case 1: static_cast<uint8_t *>(tail.label_out)[i] = static_cast<const uint8_t *>(tail.label_src)[r]; break;
          case 2: static_cast<uint16_t *>(tail.label_out)[i] = static_cast<const uint16_t *>(tail.label_src)[r]; break;
          case 4: static_cast<uint32_t *>(tail.label_out)[i] = static_cast<const uint32_t *>(tail.label_src)[r]; break;
          default:
            static_cast<unsigned long long *>(tail.label_out)[i] = static_cast<const unsigned long long *>(tail.label_src)[r];
        }
      }
    }
    // the summary is final before this launch starts (every kernel that writes it is earlier in the stream)
    if (tail.meta_dst && bid == nb - 1 && threadIdx.x < tail.meta_words) tail.meta_dst[threadIdx.x] = tail.meta_src[threadIdx.x];
    if (a.num_segs) {
      // the table goes to LDS with constant indices into the kernel arguments (a variable index would make the
      // compiler copy them to scratch memory)
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < FGNN_MAX_COPY_SEGMENTS; ++k) {
          s_dst[k] = a.seg_dst[k];
          s_src[k] = a.seg_src[k];
          s_begin[k] = a.seg_begin[k];
        }
        s_begin[FGNN_MAX_COPY_SEGMENTS] = a.seg_begin[FGNN_MAX_COPY_SEGMENTS];
      }
      __syncthreads();
      const int ns = a.num_segs;
      const uint32_t total = s_begin[ns];
      const uint32_t stride = nb * kBlock;
      constexpr int U = 8;
      // (i0 + u * stride stays below 2^32: the host refuses copies of 2^31 words and more)
      for (uint32_t i0 = bid * kBlock + threadIdx.x; i0 < total; i0 += U * stride) {
        uint32_t v[U];
        uint32_t *d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = i0 + u * stride;
          d[u] = nullptr;
          v[u] = 0;
          if (i < total && i >= i0) {
            int s = 0;
            while (s + 1 < ns && i >= s_begin[s + 1]) ++s;
            const uint32_t off = i - s_begin[s];
            d[u] = s_dst[s] + off;
            v[u] = s_src[s][off];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (d[u]) *d[u] = v[u];
      }
    }
    if (a.link_wgs == 0)  // miss rows in HBM too: every workgroup takes its share of both lists
      gather_band<UH, CPR>(a.out, a.miss_rows, a.miss_src, a.miss_dst, n_miss, a.cpr, a.miss_mask, bid, nb);
    gather_band<UH, CPR>(a.out, a.cache_rows, a.cache_src, a.cache_dst, n_cache, a.cpr, 0xFFFFFFFFu, bid, nb);
  }
  if (a.stamps) {
    __syncthreads();
    if (threadIdx.x == 0) a.stamps[2 * stamp_at + 1] = wall_clock64();
  }
}

// generic element gather (element = 1, 2, 4 or 8 bytes)
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_rows_elem_kernel(T *__restrict__ out, const T *__restrict__ src,
                                                                  const uint32_t *__restrict__ src_index,
                                                                  const uint32_t *__restrict__ dst_index,
                                                                  size_t n_host, const uint32_t *d_n, size_t cap,
                                                                  size_t dim, uint32_t src_mask) {
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t total = n * dim;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    const size_t row = e / dim;
    const size_t col = e - row * dim;
    const size_t srow = (src_index ? src_index[row] : row) & src_mask;
    const size_t drow = dst_index ? dst_index[row] : row;
    out[drow * dim + col] = src[srow * dim + col];
  }
}

size_t dtype_bytes(int dtype) {
  switch (dtype) {
    case FGNN_I8: case FGNN_U8: return 1;
    case FGNN_F16: return 2;
    case FGNN_F32: case FGNN_I32: return 4;
    case FGNN_F64: case FGNN_I64: return 8;
    default: return 0;
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" int fgnn_get_miss_cache_index(const uint32_t *table, const uint32_t *nodes, size_t num_nodes,
                                         const uint32_t *d_num_nodes, size_t num_nodes_cap, uint32_t *miss_src,
                                         uint32_t *miss_dst, uint32_t *cache_src, uint32_t *cache_dst,
                                         uint32_t *d_counts, void *ws, size_t ws_bytes, void *stream) {
  return fgnn::get_miss_cache_index_ex(table, nodes, num_nodes, d_num_nodes, num_nodes_cap, miss_src, miss_dst,
                                       cache_src, cache_dst, d_counts, ws, ws_bytes, stream, nullptr, nullptr);
}

int fgnn::get_miss_cache_index_ex(const uint32_t *table, const uint32_t *nodes, size_t num_nodes,
                                  const uint32_t *d_num_nodes, size_t num_nodes_cap, uint32_t *miss_src,
                                  uint32_t *miss_dst, uint32_t *cache_src, uint32_t *cache_dst, uint32_t *d_counts,
                                  void *ws, size_t ws_bytes, void *stream, ScanWsHost *scan,
                                  unsigned long long *stamp, const FixTail *carry_fix) {
  auto s = static_cast<hipStream_t>(stream);
  size_t cap = d_num_nodes ? num_nodes_cap : num_nodes;
  if (!d_counts) return FGNN_EINVAL;
  const FixTail carry = carry_fix && carry_fix->mapped ? *carry_fix : no_fix_tail();
  if (cap == 0) {
    FGNN_HIP_CHECK(hipMemsetAsync(d_counts, 0, 2 * sizeof(uint32_t), s));
    return carry.mapped ? hashtable_map_fix(carry, stream) : FGNN_OK;
  }
  if (!table || !nodes || !miss_src || !miss_dst || !cache_src || !cache_dst || cap > 0xffffffffull)
    return FGNN_EINVAL;
  const int ipt = cap <= (4u << 20) ? 1 : kItemsPerThread;
  const size_t nb = div_up(cap, (size_t)kBlock * ipt);
  // scratch: slot[cap] | block_sums[nb] | total_miss[1]
  if (ws_bytes < (cap + nb + 2) * sizeof(uint32_t)) return FGNN_ENOSPC;
  uint32_t *slot = static_cast<uint32_t *>(ws);
  uint32_t *sums = slot + cap;
  uint32_t *total = sums + nb;
  if (scan) {
    // single-pass path: grid resident at once (prefix over the lower-numbered workgroups), a chunk at most 32 rounds
    static int per_cu = -1;
    if (per_cu < 0 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cache_split_fused_kernel, kBlock, 0) != hipSuccess)
      per_cu = 0;
    const size_t nb1 = div_up(cap, (size_t)kBlock);
    size_t grid = (size_t)per_cu * device_cu_count() * 3 / 4;
    if (grid > scan->ws.max_tiles) grid = scan->ws.max_tiles;
    if (grid > nb1) grid = nb1;
    if (grid > 0 && div_up(cap, grid * kBlock) <= 32) {
#ifdef FGNN_PROFILING
      const uint32_t ablate = (uint32_t)tune_int("FGNN_SPLIT_ABLATE", 0);  // results are wrong when set
      if (const int g = tune_int("FGNN_SPLIT_GRID", 0)) grid = (size_t)g < grid ? (size_t)g : grid;
#endif
      hipLaunchKernelGGL(cache_split_fused_kernel, dim3(grid + carry.blocks), dim3(kBlock), 0, s, table, nodes,
                         num_nodes, d_num_nodes, cap, slot, miss_src, miss_dst, cache_src, cache_dst, d_counts,
                         scan->next(2, grid), stamp, (uint32_t)grid, carry FGNN_ABLATE_ARG(ablate));
      return launch_status(__func__);
    }
  }
  if (carry.mapped) {
    const int rc = hashtable_map_fix(carry, stream);
    if (rc != FGNN_OK) return rc;
  }
  if (ipt == 1)
    hipLaunchKernelGGL((cache_count_kernel<1>), dim3(nb), dim3(kBlock), 0, s, table, nodes, num_nodes, d_num_nodes, cap,
                       slot, sums);
  else
    hipLaunchKernelGGL((cache_count_kernel<kItemsPerThread>), dim3(nb), dim3(kBlock), 0, s, table, nodes, num_nodes,
                       d_num_nodes, cap, slot, sums);
  if (launch_scan_block_sums(sums, nb, nullptr, total, nullptr, nullptr, s, d_num_nodes, (uint32_t)(kBlock * ipt)) !=
      FGNN_OK)
    return FGNN_EHIP;
  if (ipt == 1)
    hipLaunchKernelGGL((cache_split_kernel<1>), dim3(nb), dim3(kBlock), 0, s, nodes, num_nodes, d_num_nodes, cap, slot,
                       sums, total, miss_src, miss_dst, cache_src, cache_dst, d_counts);
  else
    hipLaunchKernelGGL((cache_split_kernel<kItemsPerThread>), dim3(nb), dim3(kBlock), 0, s, nodes, num_nodes,
                       d_num_nodes, cap, slot, sums, total, miss_src, miss_dst, cache_src, cache_dst, d_counts);
  return launch_status(__func__);
}

extern "C" int fgnn_gather_rows(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index,
                                size_t n, const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, void *stream) {
  return fgnn_gather_rows_masked(out, src, src_index, dst_index, n, d_n, n_cap, dim, dtype, 0xFFFFFFFFu, stream);
}

extern "C" int fgnn_gather_rows_masked(void *out, const void *src, const uint32_t *src_index,
                                       const uint32_t *dst_index, size_t n, const uint32_t *d_n, size_t n_cap,
                                       size_t dim, int dtype, uint32_t src_row_mask, void *stream) {
  return fgnn::gather_rows_ex(out, src, src_index, dst_index, n, d_n, n_cap, dim, dtype, src_row_mask, stream, nullptr);
}

extern "C" int fgnn_gather_rows_shared(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index,
                                       size_t n, const uint32_t *d_n, size_t n_cap, size_t dim, int dtype,
                                       uint32_t src_row_mask, int shared_gpu, void *stream) {
  return fgnn::gather_rows_ex(out, src, src_index, dst_index, n, d_n, n_cap, dim, dtype, src_row_mask, stream, nullptr,
                              shared_gpu ? fgnn::kSharedGpuHostGrid : 0);
}

bool fgnn::gather_takes_tail(const void *out, const void *src, size_t n_cap, size_t dim, int dtype) {
  const size_t row_bytes = dim * dtype_bytes(dtype);
  return row_bytes && row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(src) % 16 == 0 && n_cap > 0 && n_cap * (row_bytes / 16) < 0xffffffffull;
}

// is `p` host memory the GPU reads over the host link?  Asked per launch that could be one (a registered table may be
// freed and its address reused for device memory: no caching by pointer value); a microsecond next to a launch whose
// rows cross the host link
bool fgnn::pointer_is_host(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) == hipSuccess) return a.type == hipMemoryTypeHost;
  (void)hipGetLastError();
  return false;
}

int fgnn::gather_rows_ex(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index, size_t n,
                         const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, uint32_t src_row_mask,
                         void *stream, const GatherTail *tail_in, size_t host_grid, size_t wg_per_cu_in) {
  auto s = static_cast<hipStream_t>(stream);
  const GatherTail tail = tail_in ? *tail_in : GatherTail{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
  if (tail_in && !gather_takes_tail(out, src, d_n ? n_cap : n, dim, dtype)) return FGNN_EINVAL;
  const size_t esz = dtype_bytes(dtype);
  size_t cap = d_n ? n_cap : n;
  if (esz == 0 || dim == 0) return FGNN_EINVAL;
  if (cap == 0) return FGNN_OK;
  if (!out || !src) return FGNN_EINVAL;
  const size_t row_bytes = dim * esz;
  const bool vec = (row_bytes % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(src) % 16 == 0) && row_bytes / 16 <= 0xffffffffull;
  if (vec) {
    const uint32_t cpr = (uint32_t)(row_bytes / 16);
    if (cap * cpr >= 0xffffffffull) {
      // the kernel indexes 16-byte chunks with 32 bits (64 GiB of rows per launch): larger host-sized gathers (e.g.
      // filling a whole-table feature cache) go in slices; device-sized ones of that size are not supported
      if (d_n || (!src_index && src_row_mask != 0xFFFFFFFFu)) return FGNN_EINVAL;
      const size_t rows_per = (size_t(1) << 31) / cpr;
      for (size_t r0 = 0; r0 < n; r0 += rows_per) {
        const size_t m = n - r0 < rows_per ? n - r0 : rows_per;
        const int rc = fgnn_gather_rows_masked(
            dst_index ? out : static_cast<char *>(out) + r0 * row_bytes,
            src_index ? src : static_cast<const char *>(src) + r0 * row_bytes, src_index ? src_index + r0 : nullptr,
            dst_index ? dst_index + r0 : nullptr, m, nullptr, m, dim, dtype, src_row_mask, stream);  // (no tail here)
        if (rc != FGNN_OK) return rc;
      }
      return FGNN_OK;
    }
    const size_t total = cap * cpr;
    // Rows read from HOST memory (registered / pinned: miss rows) come over the host link at ~50 GB/s, and the grid of
    // such a launch is a trade: sized like an HBM gather (1024 persistent workgroups, ~1 M loads in flight) it keeps
    // the link busiest -- a GPU that only extracts (an arch5 trainer, four batches in flight) pulls 50.6 GB/s against
    // 47.1 with 64 workgroups (profiles/r03_trainer_sweep.txt) -- but those slow reads sit in the memory pipeline in
    // front of every other kernel's misses: on a GPU that ALSO samples, the next batch's sampling chain ran 2-6x slower
    // beside a miss gather (profiles/r03_extract_timeline.txt) and 64 workgroups gave the whole leg 0.334 ms per batch
    // instead of 0.390 (profiles/r03_extract_sweep.txt).  The caller says which GPU it is on: `host_grid` workgroups
    // for host-source launches, 0 = the HBM-sized grid.  (The sweeps' knobs -- FGNN_GATHER_HOST_WGS / _UNROLL /
    // _WG_PER_CU / _NT / _NTS -- exist in the profiling build only.)
    const size_t host_wgs = (size_t)tune_int("FGNN_GATHER_HOST_WGS", (int)host_grid);  // 0: no special case
    const bool host_src = host_wgs != 0 && pointer_is_host(src);
    const int unroll = tune_int("FGNN_GATHER_UNROLL", 4);
    const size_t wg_per_cu = (size_t)tune_int("FGNN_GATHER_WG_PER_CU", wg_per_cu_in ? (int)wg_per_cu_in : 4),
                 cus = (size_t)device_cu_count();
    // non-temporal loads: gathered rows are touched once; measured 6.4 TB/s vs 4.9 TB/s with default-policy
    // loads (profiles/r01_gather_sweep.csv)
    const bool nt = tune_int("FGNN_GATHER_NT", 1) != 0;
    // non-temporal stores too: 256 MB of gathered rows left as dirty lines in the Infinity Cache slow down the
    // cold random reads of the next batch's sampling chain (whole step 0.228 -> 0.207 ms)
    const bool nts = tune_int("FGNN_GATHER_NTS", 1) != 0;
#define FGNN_GATHER3(U, C, N)                                                                                    \
  do {                                                                                                           \
    size_t blocks = div_up(total, (size_t)kBlock * U);                                                           \
    if (blocks > cus * wg_per_cu) blocks = cus * wg_per_cu;                                                      \
    if (host_src && host_wgs && blocks > host_wgs) blocks = host_wgs;                                            \
    if (nts) hipLaunchKernelGGL((gather_rows16_kernel<U, C, N, true>), dim3(blocks), dim3(kBlock), 0, s,          \
                       static_cast<chunk16 *>(out), static_cast<const chunk16 *>(src), src_index, dst_index, n,  \
                       d_n, cap, cpr, src_row_mask, tail);                                                       \
    else hipLaunchKernelGGL((gather_rows16_kernel<U, C, N, false>), dim3(blocks), dim3(kBlock), 0, s,             \
                       static_cast<chunk16 *>(out), static_cast<const chunk16 *>(src), src_index, dst_index, n,  \
                       d_n, cap, cpr, src_row_mask, tail);                                                       \
  } while (0)
#define FGNN_GATHER2(U, C) do { if (nt) FGNN_GATHER3(U, C, true); else FGNN_GATHER3(U, C, false); } while (0)
#define FGNN_GATHER(U)                                  \
  do {                                                  \
    if (cpr == 32) FGNN_GATHER2(U, 32);                 \
    else if (cpr == 64) FGNN_GATHER2(U, 64);            \
    else if (cpr == 25) FGNN_GATHER2(U, 25);            \
    else FGNN_GATHER2(U, 0);                            \
  } while (0)
    if (unroll == 8) FGNN_GATHER(8);
    else if (unroll == 2) FGNN_GATHER(2);
    else FGNN_GATHER(4);
#undef FGNN_GATHER
#undef FGNN_GATHER2
#undef FGNN_GATHER3
  } else {
    const size_t total = cap * dim;
    size_t blocks = div_up(total, kBlock);
    if (blocks > (size_t)device_cu_count() * 16) blocks = (size_t)device_cu_count() * 16;
#define FGNN_ELEM(T)                                                                                              \
  hipLaunchKernelGGL((gather_rows_elem_kernel<T>), dim3(blocks), dim3(kBlock), 0, s, static_cast<T *>(out),       \
                     static_cast<const T *>(src), src_index, dst_index, n, d_n, cap, dim, src_row_mask)
    switch (esz) {
      case 1: FGNN_ELEM(uint8_t); break;
      case 2: FGNN_ELEM(uint16_t); break;
      case 4: FGNN_ELEM(uint32_t); break;
      default: FGNN_ELEM(unsigned long long); break;
    }
#undef FGNN_ELEM
  }
  return launch_status(__func__);
}

// ---- one-launch extraction: host side ----------------------------------------------------------------------------------
namespace {

struct FusedPlan {
  uint32_t cpr = 0;
  size_t link = 0, hbm = 0;  // workgroups per band
  int ul = 4;
};

// false: the job cannot take the one-launch path (row width / alignment / size)
bool fused_plan(const ExtractJob &j, FusedPlan *p) {
  const size_t esz = dtype_bytes(j.dtype);
  const size_t row_bytes = j.dim * esz;
  if (!row_bytes || row_bytes % 16 || row_bytes / 16 > 0xffffffffull) return false;
  const size_t cap_m = j.d_counts ? j.cap : j.num_miss, cap_c = j.d_counts ? j.cap : j.num_cache;
  if ((cap_m && (!j.miss_rows || !j.miss_src || !j.miss_dst)) || (cap_c && (!j.cache_rows || !j.cache_src || !j.cache_dst)))
    return false;
  if ((cap_m || cap_c) && !j.out) return false;
  for (const void *q : {(const void *)j.out, j.miss_rows, j.cache_rows})
    if (reinterpret_cast<uintptr_t>(q) % 16) return false;
  const size_t cpr = row_bytes / 16;
  if (cap_m * cpr >= 0xffffffffull || cap_c * cpr >= 0xffffffffull) return false;
  if (j.num_segs < 0 || j.num_segs > FGNN_MAX_COPY_SEGMENTS || (j.num_segs && !j.segs)) return false;
  size_t words = 0;
  for (int k = 0; k < j.num_segs; ++k) words += j.segs[k].words;
  if (words >= (size_t(1) << 31)) return false;
  p->cpr = (uint32_t)cpr;
  p->ul = tune_int("FGNN_FUSED_LINK_UNROLL", 4) == 8 ? 8 : 4;
  const size_t cus = (size_t)device_cu_count();
  const size_t per_cu = (size_t)tune_int("FGNN_GATHER_WG_PER_CU", j.wg_per_cu ? (int)j.wg_per_cu : 4);
  const size_t link_wgs = (size_t)tune_int("FGNN_FUSED_LINK_WGS", (int)j.link_wgs);
  p->link = cap_m && link_wgs ? std::min(link_wgs, div_up(cap_m * cpr, (size_t)kBlock * p->ul)) : 0;
  size_t chunks = cap_c * cpr + (p->link ? 0 : cap_m * cpr);
  size_t hbm = div_up(chunks, (size_t)kBlock * 4);
  hbm = std::max(hbm, div_up(words, (size_t)kBlock * 8));
  if (j.tail.label_out) hbm = std::max(hbm, div_up((size_t)j.tail.num_label, (size_t)kBlock));
  if (j.tail.meta_dst) hbm = std::max<size_t>(hbm, 1);
  p->hbm = std::min(hbm, cus * per_cu);
  return true;
}

}  // namespace

bool fgnn::extract_can_fuse(const ExtractJob &j) {
  FusedPlan p;
  return fused_plan(j, &p);
}

size_t fgnn::extract_fused_grid(const ExtractJob &j, size_t *link) {
  FusedPlan p;
  if (!fused_plan(j, &p)) return 0;
  if (link) *link = p.link;
  return p.link + p.hbm;
}

int fgnn::extract_fused(const ExtractJob &j, void *stream, size_t *grid_out) {
  FusedPlan p;
  if (!fused_plan(j, &p)) return FGNN_EINVAL;
  const size_t grid = p.link + p.hbm;
  if (grid_out) *grid_out = grid;
  if (grid == 0) return FGNN_OK;
  FusedArgs a;
  memset(static_cast<void *>(&a), 0, sizeof(a));
  a.out = static_cast<chunk16 *>(j.out);
  a.miss_rows = static_cast<const chunk16 *>(j.miss_rows);
  a.cache_rows = static_cast<const chunk16 *>(j.cache_rows);
  a.miss_src = j.miss_src; a.miss_dst = j.miss_dst; a.cache_src = j.cache_src; a.cache_dst = j.cache_dst;
  a.num_miss = j.num_miss; a.num_cache = j.num_cache;
  a.d_counts = j.d_counts;
  a.cap = j.d_counts ? j.cap : std::max(j.num_miss, j.num_cache);  // (the kernel clamps either count to it)
  a.cpr = p.cpr;
  a.miss_mask = j.miss_mask;
  a.link_wgs = (uint32_t)p.link;
  a.tail = j.tail;
  a.stamps = j.stamps;
  for (int k = 0; k < j.num_segs; ++k) {
    if (!j.segs[k].words) continue;
    if (!j.segs[k].dst || !j.segs[k].src) return FGNN_EINVAL;
    a.seg_dst[a.num_segs] = j.segs[k].dst;
    a.seg_src[a.num_segs] = j.segs[k].src;
    a.seg_begin[a.num_segs + 1] = a.seg_begin[a.num_segs] + (uint32_t)j.segs[k].words;
    ++a.num_segs;
  }
  auto s = static_cast<hipStream_t>(stream);
#define FGNN_FUSED2(UL, C) \
  hipLaunchKernelGGL((extract_fused_kernel<UL, 4, C>), dim3((unsigned)grid), dim3(kBlock), 0, s, a)
#define FGNN_FUSED(C) do { if (p.ul == 4) FGNN_FUSED2(4, C); else FGNN_FUSED2(8, C); } while (0)
  if (p.cpr == 32) FGNN_FUSED(32);
  else if (p.cpr == 64) FGNN_FUSED(64);
  else if (p.cpr == 25) FGNN_FUSED(25);
  else FGNN_FUSED(0);
#undef FGNN_FUSED
#undef FGNN_FUSED2
  return launch_status(__func__);
}

namespace {
ExtractJob job_from_c(const fgnn_extract_job *c) {
  ExtractJob j;
  memset(static_cast<void *>(&j), 0, sizeof(j));
  j.out = c->out;
  j.miss_rows = c->miss_rows; j.cache_rows = c->cache_rows;
  j.miss_src = c->miss_src; j.miss_dst = c->miss_dst; j.cache_src = c->cache_src; j.cache_dst = c->cache_dst;
  j.num_miss = c->num_miss; j.num_cache = c->num_cache;
  j.d_counts = c->d_counts;
  j.cap = c->cap;
  j.dim = c->dim;
  j.dtype = c->dtype;
  j.miss_mask = c->miss_row_mask;
  if (c->label_out && c->num_label) {
    j.tail.label_out = c->label_out;
    j.tail.label_src = c->label_src;
    j.tail.label_index = c->label_index;
    j.tail.num_label = (uint32_t)c->num_label;
    j.tail.label_esz = (uint32_t)dtype_bytes(c->label_dtype);
  }
  j.segs = c->segs;
  j.num_segs = c->num_segs;
  j.link_wgs = c->link_workgroups > 0 ? (size_t)c->link_workgroups : 0;
  j.stamps = c->stamps;
  return j;
}
bool job_ok(const fgnn_extract_job *c) {
  if (!c || c->num_label > 0xffffffffull) return false;
  if (c->label_out && c->num_label && (!c->label_src || !c->label_index || dtype_bytes(c->label_dtype) == 0)) return false;
  return true;
}
}  // namespace

extern "C" int fgnn_extract_fused(const fgnn_extract_job *c, void *stream) {
  if (!job_ok(c)) return FGNN_EINVAL;
  return fgnn::extract_fused(job_from_c(c), stream, nullptr);
}

extern "C" size_t fgnn_extract_fused_link_grid(const fgnn_extract_job *c) {
  size_t link = 0;
  return job_ok(c) && fgnn::extract_fused_grid(job_from_c(c), &link) ? link : 0;
}

extern "C" size_t fgnn_extract_fused_grid(const fgnn_extract_job *c) {
  return job_ok(c) ? fgnn::extract_fused_grid(job_from_c(c)) : 0;
}

// ---- dynamic cache index (the arch4 prototype): GPUDynamicCacheManager::ReplaceCacheGPU -----------------------------
namespace fgnn {
namespace {
// hashtable_reset_nodes / hashtable_insert_nodes (cuda_cache_manager_device.cu:212-246): insert == false writes
// EMPTY at the listed nodes, insert == true writes each node's position in the list
__global__ __launch_bounds__(kBlock) void cache_table_scatter_kernel(uint32_t *table, const uint32_t *nodes, size_t n,
                                                                     const uint32_t *d_n, size_t cap, bool insert) {
  const size_t count = resolve_count(n, d_n, cap);
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride)
    table[nodes[i]] = insert ? (uint32_t)i : FGNN_EMPTY_KEY;
}
}  // namespace
}  // namespace fgnn

extern "C" int fgnn_cache_table_replace(uint32_t *table, const uint32_t *old_nodes, size_t num_old,
                                        const uint32_t *new_nodes, size_t num_new, void *stream) {
  if (!table || (num_old && !old_nodes) || (num_new && !new_nodes)) return FGNN_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t most = (size_t)device_cu_count() * 8;
  if (num_old) {
    const size_t blocks = std::min(div_up(num_old, kBlock), most);
    hipLaunchKernelGGL(cache_table_scatter_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, table, old_nodes,
                       num_old, (const uint32_t *)nullptr, num_old, false);
  }
  if (num_new) {
    const size_t blocks = std::min(div_up(num_new, kBlock), most);
    hipLaunchKernelGGL(cache_table_scatter_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, table, new_nodes,
                       num_new, (const uint32_t *)nullptr, num_new, true);
  }
  return launch_status(__func__);
}
