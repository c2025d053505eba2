// presample.hip -- engine-side glue of the pre-sampling cache policy (dist/pre_sampler.cc:75-162): the counting and
// ranking kernels live in the kernel library (csrc/presample.hip, fgnn_presample_*); the rank list goes to the
// shared host array every sampler and trainer reads (dist_engine.cc:115-127).
#include "eng_common.h"

namespace sam {

void PresampleCount(uint32_t *d_freq, const uint32_t *d_nodes, const uint32_t *d_n, size_t cap, hipStream_t st) {
  SAM_FGNN(fgnn_presample_count(d_freq, d_nodes, 0, d_n, cap, st));
}

// d_freq[num_node] -> h_rank[num_node] (host): sort desc of (freq << 32 | node)
void PresampleRank(const uint32_t *d_freq, size_t num_node, uint32_t *h_rank, hipStream_t st) {
  uint32_t *d_rank = nullptr;
  void *ws = nullptr;
  const size_t ws_bytes = fgnn_presample_rank_scratch_bytes(num_node);
  SAM_HIP(hipMalloc(&d_rank, num_node * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&ws, ws_bytes));
  SAM_FGNN(fgnn_presample_rank(d_freq, num_node, d_rank, ws, ws_bytes, st));
  SAM_HIP(hipMemcpyAsync(h_rank, d_rank, num_node * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SAM_HIP(hipStreamSynchronize(st));
  (void)hipFree(d_rank);
  (void)hipFree(ws);
}

}  // namespace sam
