// sanity_check.hip -- the reference's only runtime net on the path: every batch of seeds the shuffler hands out holds
// valid ids, and no id is handed out twice within an epoch.
// Replaces GPUSanityCheckList + GPUBatchSanityCheck (reference samgraph/common/cuda/cuda_sanity_check.cu:28-88,
// called from dist/dist_shuffler.cc:169-176 and cuda/cuda_shuffler.cc:144-151 under SAMGRAPH_SANITY_CHECK).
//
// MI355X design: ONE launch for both checks; the per-epoch "seen" map is a bitmap (1 bit per node: 13.9 MB at
// papers100M instead of the reference's 444 MB u32 map, so it stays in the Infinity Cache and its per-epoch clear is
// 32x cheaper) marked with atomicOr, which also catches a duplicate INSIDE one batch whichever lane gets there first
// (the reference reads then writes non-atomically and can miss two lanes racing on the same id).  The reference
// assert()s inside the kernel, which takes the whole GPU context down; here the kernel reports through a device word
// {bit 0: invalid id, bit 1: duplicate, bit 2: id beyond the map} and the caller decides (the engine aborts like the
// reference's CHECK, eng_shuffler.cc).
#include "fgnn_device.h"

namespace fgnn {
namespace {

__global__ __launch_bounds__(kBlock) void sanity_check_kernel(uint32_t *__restrict__ seen_bits, size_t num_node,
                                                              const uint32_t *__restrict__ input, size_t n,
                                                              uint32_t invalid_val, uint32_t *__restrict__ flags) {
  uint32_t bad = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const uint32_t id = input[i];
    if (id == invalid_val) {
      bad |= 1u;
    } else if (seen_bits) {
      if (id >= num_node) {
        bad |= 4u;
      } else {
        const uint32_t bit = 1u << (id & 31u);
        if (atomicOr(&seen_bits[id >> 5], bit) & bit) bad |= 2u;
      }
    }
  }
  // one atomic per wave that saw anything (the common case issues none)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bad |= __shfl_xor(bad, d, kWave);
  if (bad && lane_id() == 0) atomicOr(flags, bad);
}

}  // namespace
}  // namespace fgnn

extern "C" size_t fgnn_sanity_map_bytes(size_t num_node) { return fgnn::div_up(num_node, 32) * sizeof(uint32_t); }

extern "C" int fgnn_sanity_check_batch(uint32_t *seen_bits, size_t num_node, const uint32_t *input, size_t num_input,
                                       uint32_t invalid_val, uint32_t *d_flags, void *stream) {
  if (!d_flags || (!input && num_input)) return FGNN_EINVAL;
  if (num_input == 0) return FGNN_OK;
  size_t blocks = fgnn::div_up(num_input, (size_t)fgnn::kBlock);
  const size_t max_blocks = (size_t)fgnn::device_cu_count() * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(fgnn::sanity_check_kernel, dim3(blocks), dim3(fgnn::kBlock), 0, static_cast<hipStream_t>(stream),
                     seen_bits, num_node, input, num_input, invalid_val, d_flags);
  return fgnn::launch_status(__func__);
}
