// neighbourhood.hip -- whole-neighbourhood expansion (no sampling) of a node list.
//
// Reference: GPUExtractNeighbour (cuda/cuda_extract_neighbour.cu:41-169: count_edge -> cub ExclusiveSum over
// `block * grid + 1` size_t entries -> compact_edge, one THREAD copying whole rows serially, four stream
// synchronisations and two pool allocations) and its consumer DoGPUSampleAllNeighbour (cuda/cuda_loops.cc:500-571:
// per layer GPUExtractNeighbour of every node seen so far + FillWithDupMutable), which is the "sampler" of the static
// pre-sampling cache policy (cuda/pre_sampler.cc:69-71).
//
// Two entry points:
//  * fgnn_extract_neighbour: the reference's L1 function.  Rows are copied by the whole workgroup, one lane per OUTPUT
//    element (coalesced stores, contiguous loads inside a row): the owner row of an element is found by binary search
//    in the tile's 256 row offsets kept in LDS, so a 100 000-neighbour row costs the same per element as a 3-neighbour one.
//  * fgnn_neighbourhood_expand: one level of the closed-neighbourhood search with the dedup done by a direct-map stamp
//    array over the node ids instead of the hash table -- with 288 GB of HBM a u32 per graph node (444 MB for
//    papers100M) is cheap, the expansion has no capacity limit (the reference's table is sized for SAMPLED frontiers and
//    overflows on whole neighbourhoods of a power-law graph), and nothing needs a reset between batches: batch b stamps
//    with b.  Only newly reached nodes are appended, so every level expands just the previous level's frontier.
#include "fgnn_device.h"

namespace fgnn {
namespace {

constexpr int kRowsPerTile = kBlock;  // one row per thread in the offset phase

__global__ __launch_bounds__(kBlock) void nbr_tile_sums_kernel(const uint32_t *indptr, const uint32_t *input,
                                                               size_t num_input, const uint32_t *d_num_input,
                                                               size_t cap, uint32_t *sums) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t n = resolve_count(num_input, d_num_input, cap);
  const size_t i = (size_t)blockIdx.x * kRowsPerTile + threadIdx.x;
  uint32_t deg = 0;
  if (i < n) {
    const uint32_t v = input[i];
    deg = indptr[v + 1] - indptr[v];
  }
  uint32_t tot;
  (void)block_exclusive_scan(deg, sh, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void nbr_emit_kernel(const uint32_t *indptr, const uint32_t *indices,
                                                          const uint32_t *input, size_t num_input,
                                                          const uint32_t *d_num_input, size_t cap,
                                                          const uint32_t *tile_offset, uint32_t *out, size_t out_cap) {
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t row_off[kRowsPerTile];    // exclusive offsets of the tile's rows inside the tile's output
  __shared__ uint32_t row_start[kRowsPerTile];  // indptr[row]
  const size_t n = resolve_count(num_input, d_num_input, cap);
  const size_t i = (size_t)blockIdx.x * kRowsPerTile + threadIdx.x;
  uint32_t start = 0, deg = 0;
  if (i < n) {
    const uint32_t v = input[i];
    start = indptr[v];
    deg = indptr[v + 1] - start;
  }
  uint32_t tot;
  const uint32_t off = block_exclusive_scan(deg, sh, &tot);
  row_off[threadIdx.x] = off;
  row_start[threadIdx.x] = start;
  __syncthreads();
  const size_t base = tile_offset[blockIdx.x];
  for (uint32_t p = threadIdx.x; p < tot; p += kBlock) {
    // last row whose offset is <= p (empty rows share their successor's offset and are skipped by "last")
    uint32_t lo = 0, hi = kRowsPerTile - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (row_off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const size_t o = base + p;
    if (o < out_cap) out[o] = indices[(size_t)row_start[lo] + (p - row_off[lo])];
  }
}

// FillWithUnique of the seeds (cuda_loops.cc:512-517): stamped and counted BEFORE any expansion starts, so that a seed
// reached as somebody's neighbour is never appended to the next level as well
__global__ __launch_bounds__(kBlock) void nbr_mark_kernel(const uint32_t *nodes, size_t num_nodes,
                                                          const uint32_t *d_num_nodes, size_t cap, uint32_t *stamp,
                                                          uint32_t mark, uint32_t *freq) {
  const size_t n = resolve_count(num_nodes, d_num_nodes, cap);
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t v = nodes[i];
    if (atomicExch(&stamp[v], mark) != mark && freq) atomicAdd(&freq[v], 1u);
  }
}

// one level: 16 lanes per frontier node, 4 nodes per wave, 16 per workgroup and iteration
constexpr int kGroup = 16;
constexpr int kGroupsPerBlock = kBlock / kGroup;

__global__ __launch_bounds__(kBlock) void nbr_expand_kernel(const uint32_t *indptr, const uint32_t *indices,
                                                            const uint32_t *frontier, size_t num_frontier,
                                                            const uint32_t *d_num_frontier, size_t cap,
                                                            uint32_t *stamp, uint32_t mark, uint32_t *freq,
                                                            uint32_t *next, size_t next_cap, uint32_t *d_num_next) {
  const size_t n = resolve_count(num_frontier, d_num_frontier, cap);
  const int g = threadIdx.x / kGroup, gl = threadIdx.x % kGroup;
  const size_t stride = (size_t)gridDim.x * kGroupsPerBlock;
  // the loop bound is the same for every lane of a wave (ballots below): iterate on the wave's first group
  const size_t first = (size_t)blockIdx.x * kGroupsPerBlock + (size_t)(g & ~3);
  for (size_t it = first; it < n; it += stride) {
    const size_t i = it + (size_t)(g & 3);
    uint32_t start = 0, deg = 0;
    if (i < n) {
      const uint32_t v = frontier[i];
      start = indptr[v];
      deg = indptr[v + 1] - start;
    }
    // rows of the wave's four groups advance together, 16 elements each per step
    uint32_t longest = deg;
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, 16, kWave));
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, 32, kWave));
    for (uint32_t e = gl; e - gl < longest; e += kGroup) {
      bool fresh = false;
      uint32_t u = 0;
      if (e < deg) {
        u = indices[(size_t)start + e];
        // read before atomic: most neighbours of a later level are already stamped
        fresh = stamp[u] != mark && atomicExch(&stamp[u], mark) != mark;
      }
      uint32_t total;
      const uint32_t r = wave_rank(fresh, &total);
      uint32_t at = 0;
      if (total) {
        if (lane_id() == 0) at = atomicAdd(d_num_next, total);
        at = (uint32_t)__shfl((int)at, 0, kWave);
      }
      if (fresh) {
        if (freq) atomicAdd(&freq[u], 1u);
        if ((size_t)at + r < next_cap) next[at + r] = u;
      }
    }
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" size_t fgnn_extract_neighbour_scratch_bytes(size_t num_input_cap) {
  return (div_up(num_input_cap ? num_input_cap : 1, kRowsPerTile) + 1) * sizeof(uint32_t);
}

extern "C" int fgnn_extract_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                      size_t num_input, const uint32_t *d_num_input, size_t num_input_cap,
                                      uint32_t *out, size_t out_cap, size_t *d_num_out, void *ws, size_t ws_bytes,
                                      void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!d_num_out || (!d_num_input && num_input > num_input_cap)) return FGNN_EINVAL;
  const size_t tiles = div_up(num_input_cap, kRowsPerTile);
  if (tiles == 0) {  // an empty list may come with null pointers
    FGNN_HIP_CHECK(hipMemsetAsync(d_num_out, 0, sizeof(size_t), st));
    return FGNN_OK;
  }
  if (!indptr || !indices || !input || !ws || (!out && out_cap)) return FGNN_EINVAL;
  if (ws_bytes < fgnn_extract_neighbour_scratch_bytes(num_input_cap)) return FGNN_ENOSPC;
  if (tiles > 0x7FFFFFFFull) return FGNN_EINVAL;
  uint32_t *sums = static_cast<uint32_t *>(ws);
  hipLaunchKernelGGL(nbr_tile_sums_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, indptr, input, num_input,
                     d_num_input, num_input_cap, sums);
  // offsets are 32-bit: the caller's out_cap bounds what is written, *d_num_out is exact below 2^32 neighbours
  int rc = launch_scan_block_sums(sums, tiles, d_num_out, nullptr, nullptr, nullptr, st);
  if (rc != FGNN_OK) return rc;
  if (out_cap)
    hipLaunchKernelGGL(nbr_emit_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, indptr, indices, input, num_input,
                       d_num_input, num_input_cap, sums, out, out_cap);
  return launch_status(__func__);
}

extern "C" int fgnn_neighbourhood_expand(const uint32_t *indptr, const uint32_t *indices, const uint32_t *frontier,
                                         size_t num_frontier, const uint32_t *d_num_frontier, size_t frontier_cap,
                                         uint32_t *stamp, uint32_t mark, uint32_t *freq, uint32_t *next,
                                         size_t next_cap, uint32_t *d_num_next, int mark_frontier, void *stream) {
  if (!d_num_frontier && num_frontier > frontier_cap) return FGNN_EINVAL;
  if (frontier_cap == 0 || (!d_num_frontier && num_frontier == 0)) return FGNN_OK;
  if (!indptr || !indices || !frontier || !stamp || !d_num_next || (!next && next_cap)) return FGNN_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the frontier's size lives on the device: a grid for the capacity, bounded at 16 workgroups per CU
  size_t blocks = div_up(d_num_frontier ? frontier_cap : num_frontier, kGroupsPerBlock);
  const size_t most = (size_t)device_cu_count() * 16;
  if (blocks > most) blocks = most;
  if (blocks == 0) blocks = 1;
  if (mark_frontier) {
    size_t mb = div_up(d_num_frontier ? frontier_cap : num_frontier, kBlock);
    if (mb > most) mb = most;
    hipLaunchKernelGGL(nbr_mark_kernel, dim3((unsigned)mb), dim3(kBlock), 0, st, frontier, num_frontier,
                       d_num_frontier, frontier_cap, stamp, mark, freq);
  }
  hipLaunchKernelGGL(nbr_expand_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, indptr, indices, frontier,
                     num_frontier, d_num_frontier, frontier_cap, stamp, mark, freq, next, next_cap, d_num_next);
  return launch_status(__func__);
}
