// presample.hip -- pre-sampling cache policy on the GPU (init-time, not on the per-batch path).
// Reference dist/pre_sampler.cc:75-162 (twin cuda/pre_sampler.cc:57-142): after every presample
// batch the input nodes are copied to the host, an OpenMP loop bumps a 64-bit (freq<<32|node)
// table, and __gnu_parallel::sort orders it descending => rank = frequency desc, node id desc on
// ties.  Here the frequency table lives in HBM (one atomicAdd per input node, no host copy per
// batch) and the 64-bit keys are sorted by rocPRIM; the resulting rank list is identical because
// the keys are unique.  The direct-map cache table of SampleCacheTableInit (dist_engine.cc:193-229)
// is a scatter of the first num_cached rank entries.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fgnn_device.h"

namespace fgnn {
namespace {

__global__ void freq_count_kernel(uint32_t *freq, const uint32_t *nodes, size_t n_host, const uint32_t *d_n,
                                  size_t cap) {
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&freq[nodes[i]], 1u);
}

__global__ void make_keys_kernel(const uint32_t *freq, unsigned long long *keys, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    keys[i] = ((unsigned long long)freq[i] << 32) | (unsigned long long)i;
}

__global__ void low_words_kernel(const unsigned long long *keys, uint32_t *out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint32_t)keys[i];
}

__global__ void cache_table_fill_kernel(uint32_t *table, size_t num_node, const uint32_t *rank, size_t num_cached) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_cached; i += stride) {
    const uint32_t node = rank[i];
    if (node < num_node) table[node] = (uint32_t)i;
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" int fgnn_presample_count(uint32_t *d_freq, const uint32_t *d_nodes, size_t num_nodes,
                                    const uint32_t *d_num_nodes, size_t num_nodes_cap, void *stream) {
  const size_t cap = d_num_nodes ? num_nodes_cap : num_nodes;
  if (cap == 0) return FGNN_OK;
  if (!d_freq || !d_nodes) return FGNN_EINVAL;
  size_t blocks = div_up(cap, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(freq_count_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_freq, d_nodes,
                     num_nodes, d_num_nodes, cap);
  return launch_status(__func__);
}

extern "C" size_t fgnn_presample_rank_scratch_bytes(size_t num_node) {
  size_t temp = 0;
  unsigned long long *k = nullptr;
  if (rocprim::radix_sort_keys_desc(nullptr, temp, k, k, num_node, 0, 64, nullptr) != hipSuccess) return 0;
  return 2 * num_node * sizeof(unsigned long long) + ((temp + 255) & ~size_t(255)) + 256;
}

extern "C" int fgnn_presample_rank(const uint32_t *d_freq, size_t num_node, uint32_t *d_rank, void *ws,
                                   size_t ws_bytes, void *stream) {
  if (num_node == 0) return FGNN_OK;
  if (!d_freq || !d_rank || !ws) return FGNN_EINVAL;
  if (ws_bytes < fgnn_presample_rank_scratch_bytes(num_node)) return FGNN_ENOSPC;
  auto st = static_cast<hipStream_t>(stream);
  auto *keys = static_cast<unsigned long long *>(ws);
  auto *keys_out = keys + num_node;
  void *temp = keys_out + num_node;
  size_t temp_bytes = ws_bytes - 2 * num_node * sizeof(unsigned long long);
  hipLaunchKernelGGL(make_keys_kernel, dim3(4096), dim3(256), 0, st, d_freq, keys, num_node);
  FGNN_HIP_CHECK(rocprim::radix_sort_keys_desc(temp, temp_bytes, keys, keys_out, num_node, 0, 64, st));
  hipLaunchKernelGGL(low_words_kernel, dim3(4096), dim3(256), 0, st, keys_out, d_rank, num_node);
  return launch_status(__func__);
}

extern "C" int fgnn_cache_table_build(uint32_t *d_table, size_t num_node, const uint32_t *d_rank, size_t num_cached,
                                      void *stream) {
  if (!d_table || (num_cached && !d_rank) || num_cached > num_node) return FGNN_EINVAL;
  auto st = static_cast<hipStream_t>(stream);
  FGNN_HIP_CHECK(hipMemsetAsync(d_table, 0xFF, num_node * sizeof(uint32_t), st));
  if (num_cached) {
    size_t blocks = div_up(num_cached, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cache_table_fill_kernel, dim3(blocks), dim3(256), 0, st, d_table, num_node, d_rank, num_cached);
  }
  return launch_status(__func__);
}
