// presample.hip -- pre-sampling cache policy on the GPU.
// Reference dist/pre_sampler.cc:75-162 (twin cuda/pre_sampler.cc:57-142): after every presample
// batch the input nodes are copied to the host, an OpenMP loop bumps a 64-bit (freq<<32|node)
// table, and __gnu_parallel::sort orders it descending => rank = frequency desc, node id desc on
// ties.  Here the frequency table lives in HBM (one atomicAdd per input node, no host copy per
// batch) and the 64-bit keys are sorted by rocPRIM; the resulting rank list is identical because
// the keys are unique.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "eng_common.h"

namespace sam {
namespace {

__global__ void freq_count_kernel(uint32_t *freq, const uint32_t *nodes, const uint32_t *d_n, size_t cap) {
  size_t n = *d_n;
  if (n > cap) n = cap;
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

}  // namespace

void PresampleCount(uint32_t *d_freq, const uint32_t *d_nodes, const uint32_t *d_n, size_t cap, hipStream_t st) {
  size_t blocks = RoundUpDiv(cap, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(freq_count_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, d_freq, d_nodes, d_n, cap);
}

// d_freq[num_node] -> h_rank[num_node] (host): sort desc of (freq << 32 | node)
void PresampleRank(const uint32_t *d_freq, size_t num_node, uint32_t *h_rank, hipStream_t st) {
  unsigned long long *keys = nullptr, *keys_out = nullptr;
  uint32_t *d_rank = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  SAM_HIP(hipMalloc(&keys, num_node * sizeof(unsigned long long)));
  SAM_HIP(hipMalloc(&keys_out, num_node * sizeof(unsigned long long)));
  SAM_HIP(hipMalloc(&d_rank, num_node * sizeof(uint32_t)));
  hipLaunchKernelGGL(make_keys_kernel, dim3(4096), dim3(256), 0, st, d_freq, keys, num_node);
  SAM_HIP(rocprim::radix_sort_keys_desc(nullptr, temp_bytes, keys, keys_out, num_node, 0, 64, st));
  SAM_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
  SAM_HIP(rocprim::radix_sort_keys_desc(temp, temp_bytes, keys, keys_out, num_node, 0, 64, st));
  hipLaunchKernelGGL(low_words_kernel, dim3(4096), dim3(256), 0, st, keys_out, d_rank, num_node);
  SAM_HIP(hipMemcpyAsync(h_rank, d_rank, num_node * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SAM_HIP(hipStreamSynchronize(st));
  (void)hipFree(keys);
  (void)hipFree(keys_out);
  (void)hipFree(d_rank);
  (void)hipFree(temp);
}

}  // namespace sam
