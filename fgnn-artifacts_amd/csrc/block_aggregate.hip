// block_aggregate.hip -- neighbourhood aggregation over a sampled block (the message-passing step of the DGL-free
// SAGEConv / GraphConv / PinSAGE layers in examples/models.py):
//     out[col[e], :] += w[e] * h[row[e], :]        for every edge e of the block
// Replaces torch's `h[row]` (materialises an E x D tensor) + `index_add_` (one float atomic per element): rows are
// read once straight from h, products are accumulated in registers while consecutive edges share their destination
// (the samplers emit edges seed-major, so a destination's edges are contiguous) and only segment boundaries touch
// memory, with hardware float atomics (a segment may continue in the next wave's chunk).  The same kernel with row
// and col swapped is the backward pass (grad_h[row[e]] += w[e] * grad_out[col[e]]).  Sums in fp32; the order of the
// additions is not fixed, results agree with the torch reference to rounding (tests/test_hip_parity.py, rtol 1e-4).
#include "fgnn_device.h"

namespace fgnn {
namespace {

constexpr int kEdgesPerWave = 32;

// Lane l owns dims l, l + 64, ... (P of them per pass over the edges): every load and every float atomic of a wave
// instruction covers 64 consecutive floats.  All lanes run the edge loop (the shuffles need them), memory operations
// are masked by `l + 64 p < dim`.
template <int P>
__global__ __launch_bounds__(kBlock) void block_aggregate_kernel(const uint32_t *__restrict__ src_idx,
                                                                 const uint32_t *__restrict__ dst_idx,
                                                                 const float *__restrict__ w,
                                                                 const float *__restrict__ h, float *out,
                                                                 size_t num_edge, uint32_t dim, uint32_t out_ld,
                                                                 float *deg) {
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + wave_id();
  const size_t e0 = wave * kEdgesPerWave;
  if (e0 >= num_edge) return;  // wave-uniform
  const size_t e1 = e0 + kEdgesPerWave < num_edge ? e0 + kEdgesPerWave : num_edge;
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t cnt = (uint32_t)(e1 - e0);
  // the chunk's edges: lane i holds edge e0 + i; the loop below broadcasts them with wave shuffles instead of
  // issuing a dependent (index -> row) load chain per edge
  uint32_t my_src = 0, my_dst = 0;
  float my_w = 1.0f;
  if (lane < cnt) {
    my_src = src_idx[e0 + lane];
    my_dst = dst_idx[e0 + lane];
    if (w) my_w = w[e0 + lane];
  }
  // in-degrees on the way (deg != null): the first lane of every run of equal destinations adds the run's length --
  // one float atomic per (wave, destination) instead of an index_add_ pass of its own over the edges
  if (deg) {
    const uint32_t prev = __shfl_up(my_dst, 1, kWave);
    const bool head = lane < cnt && (lane == 0 || prev != my_dst);
    const unsigned long long heads = __ballot(head);
    if (head) {
      const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
      const uint32_t next = later ? (uint32_t)__builtin_ctzll(later) : cnt;
      unsafeAtomicAdd(&deg[my_dst], (float)(next - lane));
    }
  }
  for (uint32_t dbase = 0; dbase < dim; dbase += kWave * P) {  // one trip for dim <= 64 P
    bool act[P];
    float acc[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      act[q] = dbase + lane + kWave * q < dim;
      acc[q] = 0.0f;
    }
    uint32_t cur = __shfl(my_dst, 0, kWave);
    constexpr uint32_t UN = 4;  // edges whose row loads are in flight together
    for (uint32_t i0 = 0; i0 < cnt; i0 += UN) {
      float v[UN][P];
      uint32_t c[UN];
      float we[UN];
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) {
        const uint32_t i = i0 + u < cnt ? i0 + u : cnt - 1;  // clamp: the extra loads are discarded below
        const uint32_t sidx = __shfl(my_src, (int)i, kWave);
        c[u] = __shfl(my_dst, (int)i, kWave);
        we[u] = __shfl(my_w, (int)i, kWave);
        const float *hp = h + (size_t)sidx * dim + dbase + lane;
#pragma unroll
        for (int q = 0; q < P; ++q) v[u][q] = act[q] ? hp[kWave * q] : 0.0f;
      }
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) {
        if (i0 + u >= cnt) break;  // wave-uniform
        if (c[u] != cur) {         // wave-uniform: segment boundary, flush
          float *op = out + (size_t)cur * out_ld + dbase + lane;
#pragma unroll
          for (int q = 0; q < P; ++q) {
            if (act[q]) unsafeAtomicAdd(op + kWave * q, acc[q]);
            acc[q] = 0.0f;
          }
          cur = c[u];
        }
#pragma unroll
        for (int q = 0; q < P; ++q) acc[q] += we[u] * v[u][q];
      }
    }
    float *op = out + (size_t)cur * out_ld + dbase + lane;
#pragma unroll
    for (int q = 0; q < P; ++q)
      if (act[q]) unsafeAtomicAdd(op + kWave * q, acc[q]);
  }
}

}  // namespace
}  // namespace fgnn

extern "C" int fgnn_block_aggregate_ex(const uint32_t *src_index, const uint32_t *dst_index, const float *edge_weight,
                                       size_t num_edge, const float *h, size_t dim, float *out, size_t out_ld,
                                       float *in_degree, void *stream) {
  using namespace fgnn;
  if (num_edge == 0) return FGNN_OK;
  if (!src_index || !dst_index || !h || !out || dim == 0 || dim > 0xffffffffull || out_ld < dim || out_ld > 0xffffffffull)
    return FGNN_EINVAL;
  auto st = static_cast<hipStream_t>(stream);
  const size_t waves = div_up(num_edge, (size_t)kEdgesPerWave);
  const size_t blocks = div_up(waves, (size_t)kWavesPerBlock);
#define FGNN_AGG(PP)                                                                                               \
  hipLaunchKernelGGL((block_aggregate_kernel<PP>), dim3(blocks), dim3(kBlock), 0, st, src_index, dst_index,        \
                     edge_weight, h, out, num_edge, (uint32_t)dim, (uint32_t)out_ld, in_degree)
  if (dim <= 64) FGNN_AGG(1);
  else if (dim <= 128) FGNN_AGG(2);
  else FGNN_AGG(4);
#undef FGNN_AGG
  return launch_status(__func__);
}

extern "C" int fgnn_block_aggregate(const uint32_t *src_index, const uint32_t *dst_index, const float *edge_weight,
                                    size_t num_edge, const float *h, size_t dim, float *out, void *stream) {
  return fgnn_block_aggregate_ex(src_index, dst_index, edge_weight, num_edge, h, dim, out, dim, nullptr, stream);
}
