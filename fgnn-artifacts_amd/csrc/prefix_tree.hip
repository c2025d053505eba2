// prefix_tree.hip -- 5-ary search trees over the long rows of a per-row prefix-sum table (init time).
//
// The weighted sampler draws x = U(0,1] * rowsum and looks for the row position whose prefix interval holds x by
// binary search (reference samgraph/common/cuda/cuda_sampling_weighted_khop_prefix.cu:66-86): ceil(log2 deg) DEPENDENT
// 4-byte reads per draw, each a different cache line of the row -- ~20 on twitter's hub rows, and hub rows are where a
// frontier's draws concentrate (weighted_draw_count_kernel: 165 of a twitter-shaped batch's 656 us in round 3).
// What that search returns on a non-decreasing row is the FIRST position whose prefix is >= x (invariant
// prefix[lo] < x <= prefix[hi], ends with hi = lo + 1) -- a property of the row and x alone, whatever the probe order.
//
// What a probe costs on this GPU is a tag look-up in the CU's vector cache per (lane, line) -- a wave instruction whose
// 64 lanes touch 64 lines keeps that cache busy for 64 cycles whatever the width of the load -- and a round trip.  A
// wider node therefore only pays if ONE load instruction fetches it: a first version with 64-byte nodes (16
// separators, four 16-byte loads per node and a 16-entry scan of the row at the end) made twice the look-ups of the
// binary search and was SLOWER (twitter shape, interleaved A/B: sampler-side stage 0.351 -> 0.396 ms per batch,
// profiles/r04_b_tree16_ab.txt).  The node here is what one lane gets from one load: 16 bytes = 4 separators = a 5-way
// decision, log2(5) = 2.3 bits per look-up and round trip instead of 1.
//
// Rows longer than kPrefixTreeMinLen get levels l = T .. 1 (5^T >= len): node j of level l covers row positions
// [j 5^l, (j+1) 5^l) and holds the LAST prefix value of its first four children (+inf where a child is empty; the
// fifth child is implied: x <= the row's last value).  Level 1's separators are row entries themselves, so the descent
// ends at the position -- the row itself is not read.  Levels sit top first in one pool of 16-byte nodes,
// tree_off[row] = index of the row's root.  ceil(log5 len) look-ups: 4 for 300 entries, 6 for 4 096, 7 for 65 536, 9 for
// a million, against 9, 12, 16 and 20.
// Rows whose prefix sums are NOT non-decreasing (a table that was not built by a sequential sum) get no tree: there
// the binary search's answer depends on its probe order, and the sampler keeps the reference's search for them --
// results stay bit-identical to the reference for ANY table.  Memory: ~1 float per entry of a long row + 4 bytes per
// node id (twitter shape: < 6 GB + 0.17 GB of 288 GB).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fgnn_device.h"

namespace fgnn {
namespace {

__device__ __forceinline__ uint32_t tree_levels(uint32_t len) {  // smallest T with 5^T >= len
  uint32_t T = 0;
  for (uint32_t m = len; m > 1; m = (m + 4u) / 5u) ++T;
  return T;
}
// nodes of a row's tree: level l has ceil(len / 5^l) of them (the root level one)
__device__ __forceinline__ uint32_t tree_nodes(uint32_t len) {
  uint32_t nodes = 0;
  for (uint32_t m = len; m > 1;) {
    m = (m + 4u) / 5u;
    nodes += m;
  }
  return nodes;
}

__global__ __launch_bounds__(kBlock) void tree_size_kernel(const uint32_t *__restrict__ indptr, size_t num_node,
                                                           uint32_t *__restrict__ sizes, uint32_t *long_rows,
                                                           uint32_t *num_long) {
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= num_node) return;
  const uint32_t len = indptr[r + 1] - indptr[r];
  uint32_t sz = 0;
  if (len > kPrefixTreeMinLen) {
    sz = tree_nodes(len);
    long_rows[atomicAdd(num_long, 1u)] = (uint32_t)r;
  }
  sizes[r] = sz;
}

// one wavefront per long row: check that the row is non-decreasing, then gather its separators
__global__ __launch_bounds__(kBlock) void tree_fill_kernel(const uint32_t *__restrict__ indptr,
                                                           const float *__restrict__ prefix,
                                                           const uint32_t *__restrict__ long_rows,
                                                           const uint32_t *num_long, uint32_t *tree_off, float *pool,
                                                           uint32_t *num_refused) {
  const uint32_t nl = *num_long;
  const uint32_t waves = gridDim.x * kWavesPerBlock;
  for (uint32_t q = blockIdx.x * kWavesPerBlock + wave_id(); q < nl; q += waves) {
    const uint32_t r = long_rows[q];
    const uint32_t off = indptr[r], len = indptr[r + 1] - off;
    const float *row = prefix + off;
    bool ok = true;
    for (uint32_t i = lane_id(); i + 1 < len; i += kWave) ok = ok && row[i] <= row[i + 1];  // (NaN: refused)
    if (__ballot(!ok) != 0ull) {
      if (lane_id() == 0) {
        tree_off[r] = FGNN_EMPTY_KEY;
        atomicAdd(num_refused, 1u);
      }
      continue;
    }
    const uint32_t T = tree_levels(len);
    float *node = pool + (size_t)tree_off[r] * 4;
    for (uint32_t l = T; l >= 1; --l) {
      unsigned long long child = 1;  // 5^(l-1): positions under one child of a level-l node
      for (uint32_t k = 1; k < l; ++k) child *= 5ull;
      uint32_t n = len;              // ceil(len / 5^l)
      for (uint32_t k = 0; k < l; ++k) n = (n + 4u) / 5u;
      for (uint32_t e = lane_id(); e < 4u * n; e += kWave) {
        const unsigned long long start = (unsigned long long)(e >> 2) * child * 5ull + (unsigned long long)(e & 3u) * child;
        float v = __builtin_inff();  // an empty child
        if (start < len) {
          const unsigned long long end = start + child;  // last position of the child, clipped to the row
          v = row[end < len ? (uint32_t)(end - 1ull) : len - 1u];
        }
        node[e] = v;
      }
      node += 4u * n;
    }
  }
}

// rows without a tree read EMPTY: mark the short ones (the long ones hold their root's node index from the scan)
__global__ __launch_bounds__(kBlock) void tree_mark_short_kernel(const uint32_t *__restrict__ sizes, size_t num_node,
                                                                 uint32_t *tree_off) {
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < num_node && sizes[r] == 0) tree_off[r] = FGNN_EMPTY_KEY;
}

// the per-node records (PrefixTreeView::rec), once the trees' roots are final
__global__ __launch_bounds__(kBlock) void tree_node_rec_kernel(const uint32_t *__restrict__ indptr,
                                                               const float *__restrict__ prefix,
                                                               const uint32_t *__restrict__ tree_off, size_t num_node,
                                                               uint4 *__restrict__ rec) {
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= num_node) return;
  const uint32_t off = indptr[r], len = indptr[r + 1] - off;
  const uint32_t root = len > kPrefixTreeMinLen ? tree_off[r] : FGNN_EMPTY_KEY;
  rec[r] = uint4{off, len, root, len ? __float_as_uint(prefix[off + len - 1]) : 0u};
}

}  // namespace

struct PrefixTreeHost {
  uint32_t *tree_off = nullptr;
  float *pool = nullptr;
  uint4 *rec = nullptr;
  size_t nodes = 0, long_rows = 0, refused = 0;
};

void prefix_tree_destroy(PrefixTreeHost *t) {
  if (!t) return;
  if (t->tree_off) (void)hipFree(t->tree_off);
  if (t->pool) (void)hipFree(t->pool);
  if (t->rec) (void)hipFree(t->rec);
  delete t;
}

PrefixTreeView prefix_tree_view(const PrefixTreeHost *t) {
  return t ? PrefixTreeView{t->tree_off, t->pool, t->rec} : PrefixTreeView{nullptr, nullptr, nullptr};
}

// synchronous (init time); null if there is nothing to build or memory is short -- the sampler then searches every row
// like the reference
PrefixTreeHost *prefix_tree_build(const uint32_t *indptr, const float *prefix, size_t num_node) {
  if (!indptr || !prefix || num_node == 0 || num_node >= 0xffffffffull) return nullptr;
  auto *t = new PrefixTreeHost();
  uint32_t *sizes = nullptr, *long_rows = nullptr, *counters = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  auto fail = [&]() -> PrefixTreeHost * {
    (void)hipGetLastError();
    if (sizes) (void)hipFree(sizes);
    if (long_rows) (void)hipFree(long_rows);
    if (counters) (void)hipFree(counters);
    if (temp) (void)hipFree(temp);
    prefix_tree_destroy(t);
    return nullptr;
  };
  const size_t nb = div_up(num_node, (size_t)kBlock);
  if (hipMalloc(&t->tree_off, num_node * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&sizes, num_node * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&long_rows, num_node * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&counters, 2 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(counters, 0, 2 * sizeof(uint32_t)) != hipSuccess)
    return fail();
  hipLaunchKernelGGL(tree_size_kernel, dim3(nb), dim3(kBlock), 0, 0, indptr, num_node, sizes, long_rows, counters);
  if (rocprim::exclusive_scan(nullptr, temp_bytes, sizes, t->tree_off, 0u, num_node, rocprim::plus<uint32_t>(), 0) !=
          hipSuccess ||
      hipMalloc(&temp, temp_bytes ? temp_bytes : 16) != hipSuccess ||
      rocprim::exclusive_scan(temp, temp_bytes, sizes, t->tree_off, 0u, num_node, rocprim::plus<uint32_t>(), 0) !=
          hipSuccess)
    return fail();
  uint32_t last_off = 0, last_size = 0, h_counters[2] = {0, 0};
  if (hipMemcpy(&last_off, t->tree_off + num_node - 1, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(&last_size, sizes + num_node - 1, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(h_counters, counters, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
    return fail();
  t->nodes = (size_t)last_off + last_size;
  t->long_rows = h_counters[0];
  if (t->nodes == 0) return fail();  // no long row: nothing to gain
  if (t->nodes >= 0xffffffffull || hipMalloc(&t->pool, t->nodes * 4 * sizeof(float)) != hipSuccess) return fail();
  hipLaunchKernelGGL(tree_mark_short_kernel, dim3(nb), dim3(kBlock), 0, 0, sizes, num_node, t->tree_off);
  size_t blocks = div_up(t->long_rows, (size_t)kWavesPerBlock);
  if (blocks > (size_t)device_cu_count() * 16) blocks = (size_t)device_cu_count() * 16;
  hipLaunchKernelGGL(tree_fill_kernel, dim3(blocks), dim3(kBlock), 0, 0, indptr, prefix, long_rows, counters,
                     t->tree_off, t->pool, counters + 1);
  if (hipMemcpy(h_counters, counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return fail();
  t->refused = h_counters[1];
  // 16 bytes per node (twitter shape: 0.67 GB); without the memory the draws read indptr / tree_off / the row as before
  // (FGNN_PREFIX_REC=0, profiling build: A/B)
  if (tune_int("FGNN_PREFIX_REC", 1) != 0 && hipMalloc(&t->rec, num_node * sizeof(uint4)) == hipSuccess) {
    hipLaunchKernelGGL(tree_node_rec_kernel, dim3(nb), dim3(kBlock), 0, 0, indptr, prefix, t->tree_off, num_node, t->rec);
    if (hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(t->rec);
      t->rec = nullptr;
    }
  } else {
    (void)hipGetLastError();
    t->rec = nullptr;
  }
  (void)hipFree(sizes);
  (void)hipFree(long_rows);
  (void)hipFree(counters);
  (void)hipFree(temp);
  return t;
}

void prefix_tree_stats(const PrefixTreeHost *t, size_t out[3]) {
  out[0] = t ? t->long_rows : 0;
  out[1] = t ? t->refused : 0;
  out[2] = t ? t->nodes * 16 : 0;
}

}  // namespace fgnn
