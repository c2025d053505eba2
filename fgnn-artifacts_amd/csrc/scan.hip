// scan.hip -- single-workgroup exclusive scan of per-block partial sums (<= a few 100k entries).
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

}  // namespace fgnn
