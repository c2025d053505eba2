// train_ops.hip -- the small fused pieces of a GraphSAGE training step on the blocks the engine returns (consumer side
// of the path; reference: the DGL / PyTorch ops of example/samgraph/multi_gpu/train_graphsage.py:
// 24-51,300-330 -- SAGEConv('mean'), ReLU, Dropout, CrossEntropyLoss, Adam).
//
// Why kernels for elementwise work: the step is replayed as a captured HIP graph (examples/graphed_step.py) and a graph
// node costs the GPU 15-20 us on this runtime whatever it does (profiles/r04_c_train_graph_vs_eager.txt) -- the ~38
// nodes of round 4's step took 0.58 ms for ~0.35 ms of kernels.  Each entry point below replaces two to four torch ops
// (= nodes) by one launch, with the same fp32 arithmetic per element:
//   fgnn_sage_finish_z      clamp_(deg, 1) + reciprocal_ + z[:, din:] *= inv + z[:, :din] = h[:num_dst]
//   fgnn_sage_grad_prep     zeros(gh) + gh[:num_dst] += gz[:, :din] + gagg = gz[:, din:] * inv
//   fgnn_relu_dropout       F.relu + nn.Dropout (forward: Philox mask from a device-side step counter; backward from y)
//   fgnn_softmax_xent       log_softmax + nll_loss(mean) forward AND the gradient of the logits, deterministic sum
//   fgnn_adam_step          torch.optim.Adam's update for up to 8 tensors in one launch, step count on the device
#include "fgnn_device.h"

namespace fgnn {
namespace {

// z[:, :din] = h[:num_dst], z[:, din:] *= inv, inv = 1 / max(deg, 1) (also stored: the backward pass scales with it)
__global__ __launch_bounds__(kBlock) void sage_finish_z_kernel(float *__restrict__ z, uint32_t ld,
                                                               const float *__restrict__ h, uint32_t h_ld,
                                                               const float *__restrict__ deg, float *__restrict__ inv_out,
                                                               uint32_t num_dst, uint32_t din) {
  // one thread per (row, 4 columns of the left half and the matching 4 of the right half)
  const uint32_t q = din / 4;
  const size_t total = (size_t)num_dst * q;
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
    const uint32_t i = (uint32_t)(t / q), c = (uint32_t)(t - (size_t)i * q) * 4;
    const float d = deg[i];
    const float inv = 1.0f / (d < 1.0f ? 1.0f : d);  // clamp(min=1).reciprocal()
    if (c == 0) inv_out[i] = inv;
    f4 *zl = reinterpret_cast<f4 *>(z + (size_t)i * ld + c);
    f4 *zr = reinterpret_cast<f4 *>(z + (size_t)i * ld + din + c);
    *zl = *reinterpret_cast<const f4 *>(h + (size_t)i * h_ld + c);
    f4 v = *zr;
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    *zr = v;
  }
}

__global__ __launch_bounds__(kBlock) void sage_grad_prep_kernel(const float *__restrict__ gz, uint32_t gz_ld,
                                                                const float *__restrict__ inv, float *__restrict__ gh,
                                                                float *__restrict__ gagg, uint32_t num_dst,
                                                                uint32_t num_src, uint32_t din) {
  const uint32_t q = din / 4;
  const size_t total = (size_t)num_src * q;
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
    const uint32_t i = (uint32_t)(t / q), c = (uint32_t)(t - (size_t)i * q) * 4;
    f4 l = {0.f, 0.f, 0.f, 0.f};
    if (i < num_dst) {
      l = *reinterpret_cast<const f4 *>(gz + (size_t)i * gz_ld + c);
      f4 r = *reinterpret_cast<const f4 *>(gz + (size_t)i * gz_ld + din + c);
      const float s = inv[i];
      r.x *= s; r.y *= s; r.z *= s; r.w *= s;
      *reinterpret_cast<f4 *>(gagg + (size_t)i * din + c) = r;
    }
    *reinterpret_cast<f4 *>(gh + (size_t)i * din + c) = l;
  }
}

// y = relu(x) * keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox keyed (seed, *d_step, layer tag, element / 4).
// *d_step is the optimizer's device-side step count (advanced by fgnn_adam_step once per training step: a captured
// graph replays this launch, so the count cannot be a launch argument); null: step 0.  The backward pass needs no
// random numbers: y > 0 exactly where the element was kept and positive.
__global__ __launch_bounds__(kBlock) void relu_dropout_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              size_t n4, float p, float scale, uint64_t seed,
                                                              const unsigned long long *__restrict__ d_step,
                                                              uint32_t layer_tag) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const unsigned long long step = d_step ? *d_step : 0ull;
  // p as a 32-bit threshold: keep iff r >= p * 2^32
  const uint32_t thr = p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (uint32_t)((double)p * 4294967296.0));
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (size_t)gridDim.x * kBlock) {
    const f4 v = reinterpret_cast<const f4 *>(x)[t];
    const u32x4 r = philox_block(seed, step, layer_tag, (uint32_t)t, (uint32_t)(t >> 32));
    f4 o;
    o.x = (v.x > 0.f && r.x >= thr) ? v.x * scale : 0.f;
    o.y = (v.y > 0.f && r.y >= thr) ? v.y * scale : 0.f;
    o.z = (v.z > 0.f && r.z >= thr) ? v.z * scale : 0.f;
    o.w = (v.w > 0.f && r.w >= thr) ? v.w * scale : 0.f;
    reinterpret_cast<f4 *>(y)[t] = o;
  }
}

__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(const float *__restrict__ y,
                                                                  const float *__restrict__ gy, float *__restrict__ gx,
                                                                  size_t n4, float scale) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (size_t)gridDim.x * kBlock) {
    const f4 v = reinterpret_cast<const f4 *>(y)[t], g = reinterpret_cast<const f4 *>(gy)[t];
    f4 o;
    o.x = v.x > 0.f ? g.x * scale : 0.f;
    o.y = v.y > 0.f ? g.y * scale : 0.f;
    o.z = v.z > 0.f ? g.z * scale : 0.f;
    o.w = v.w > 0.f ? g.w * scale : 0.f;
    reinterpret_cast<f4 *>(gx)[t] = o;
  }
}

// one wavefront per row: max, log-sum-exp, loss_i = lse - x[label]; dlogits = (softmax - onehot) / n.  The mean of the
// row losses is summed in a FIXED order -- rows of a workgroup in LDS, the workgroups' partial sums by the workgroup that
// arrives last (eight independent loads per thread, then a tree) -- so the value does not depend on the order the
// workgroups run in.
__global__ __launch_bounds__(kBlock) void softmax_xent_kernel(const float *__restrict__ logits, uint32_t ld,
                                                              const long long *__restrict__ labels, uint32_t n,
                                                              uint32_t C, float *__restrict__ partial,
                                                              float *__restrict__ loss, float *__restrict__ dlogits,
                                                              uint32_t dl_ld, uint32_t *arrive) {
  const uint32_t lane = lane_id();
  __shared__ float wl[kWavesPerBlock];
  // rows whose label is outside [0, C) are IGNORED like torch's ignore_index rows (default -100): no loss, a zero
  // gradient row, and the mean runs over the other rows only.  Every workgroup counts them itself (n labels out of L2,
  // ~1 us) -- the gradient's scale must be known before the first row is written
  __shared__ uint32_t s_valid[kWavesPerBlock];
  {
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
      const long long lab = labels[i];
      cnt += (lab >= 0 && lab < (long long)C) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, kWave);
    if (lane == 0) s_valid[wave_id()] = cnt;
    __syncthreads();
  }
  uint32_t valid = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) valid += s_valid[w];
  const float inv_n = 1.0f / (float)(valid ? valid : 1u);
  float my = 0.f;
  // (a wave walks rows with the grid's stride: at most kXentBlocks arrivals at the counter below -- same-address
  // atomics complete one per ~29 ns on this GPU, a workgroup per four rows made the launch 56 us for 8 000 rows)
  for (uint32_t row = blockIdx.x * kWavesPerBlock + wave_id(); row < n; row += gridDim.x * kWavesPerBlock) {
    const float *x = logits + (size_t)row * ld;
    const long long lab = labels[row];
    float *g = dlogits + (size_t)row * dl_ld;
    if (lab < 0 || lab >= (long long)C) {  // ignored row
      for (uint32_t c = lane; c < C; c += kWave) g[c] = 0.f;
      continue;
    }
    float lse;
    if (C <= 4u * kWave) {  // the row in registers: one pass over memory (class counts of the reference's datasets: 41-172)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = lane + (uint32_t)u * kWave < C ? x[lane + (uint32_t)u * kWave] : -__builtin_inff();
      float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, kWave));
      float e[4], s = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        e[u] = lane + (uint32_t)u * kWave < C ? expf(v[u] - m) : 0.f;
        s += e[u];
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, kWave);
      lse = m + logf(s);
      const float inv_s = 1.0f / s;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t c = lane + (uint32_t)u * kWave;
        if (c < C) g[c] = (e[u] * inv_s - ((long long)c == lab ? 1.f : 0.f)) * inv_n;
      }
    } else {
      float m = -__builtin_inff();
      for (uint32_t c = lane; c < C; c += kWave) m = fmaxf(m, x[c]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, kWave));
      float s = 0.f;
      for (uint32_t c = lane; c < C; c += kWave) s += expf(x[c] - m);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, kWave);
      lse = m + logf(s);
      const float inv_s = 1.0f / s;
      for (uint32_t c = lane; c < C; c += kWave) g[c] = (expf(x[c] - m) * inv_s - ((long long)c == lab ? 1.f : 0.f)) * inv_n;
    }
    my += lse - x[lab];
  }
  if (lane == 0) wl[wave_id()] = my;
  __shared__ uint32_t last;
  __syncthreads();
  if (threadIdx.x == 0) {
    float p = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) p += wl[w];
    __hip_atomic_store(&partial[blockIdx.x], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();  // this workgroup's partial sum before the arrival
    last = atomicAdd(arrive, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  __shared__ float part[kBlock];
  float acc = 0.f;
  for (uint32_t b0 = 0; b0 < gridDim.x; b0 += 8u * kBlock) {  // (one trip: at most kXentBlocks partial sums)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t b = b0 + (uint32_t)u * kBlock + threadIdx.x;
      v[u] = b < gridDim.x ? __hip_atomic_load(&partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *loss = part[0] * inv_n;  // mean over the rows that count (0 when none does; torch: nan)
    *arrive = 0;  // ready for the next launch (stream-ordered)
  }
}

struct AdamTensors {
  float *p[8];
  const float *g[8];
  float *m[8], *v[8];
  unsigned long long n[8];
  int count;
};

__global__ __launch_bounds__(kBlock) void adam_kernel(AdamTensors t, float lr, float b1, float b2, float eps,
                                                      float weight_decay, unsigned long long *d_step) {
  // the step count lives on the device (a captured graph replays this launch): everyone reads the OLD value, the last
  // workgroup to leave advances it (stream order makes the new value visible to the next launch)
  const unsigned long long step = *d_step + 1ull;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  const float step_size = lr / bc1, rsq_bc2 = 1.0f / sqrtf(bc2);
  for (int k = 0; k < t.count; ++k) {
    float *p = t.p[k], *m = t.m[k], *v = t.v[k];
    const float *g = t.g[k];
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < t.n[k];
         i += (unsigned long long)gridDim.x * kBlock) {
      float gi = g[i];
      const float pi = p[i];
      if (weight_decay != 0.f) gi += weight_decay * pi;
      const float mi = b1 * m[i] + (1.0f - b1) * gi;
      const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      p[i] = pi - step_size * (mi / (sqrtf(vi) * rsq_bc2 + eps));
    }
  }
  __shared__ uint32_t last;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int *arrive = reinterpret_cast<unsigned int *>(d_step + 1);
    last = atomicAdd(arrive, 1u) == gridDim.x - 1 ? 1u : 0u;
    if (last) {
      *arrive = 0;
      *d_step = step;
    }
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" int fgnn_sage_finish_z(float *z, size_t ld, const float *h, size_t h_ld, const float *deg, float *inv,
                                  size_t num_dst, size_t din, void *stream) {
  if (num_dst == 0) return FGNN_OK;
  if (!z || !h || !deg || !inv || din == 0 || din % 4 || ld < 2 * din || ld % 4 || h_ld < din || h_ld % 4 ||
      num_dst > 0xffffffffull || ld > 0xffffffffull || h_ld > 0xffffffffull ||
      reinterpret_cast<uintptr_t>(z) % 16 || reinterpret_cast<uintptr_t>(h) % 16)
    return FGNN_EINVAL;
  auto st = static_cast<hipStream_t>(stream);
  const size_t total = num_dst * (din / 4);
  size_t blocks = div_up(total, (size_t)kBlock);
  const size_t most = (size_t)device_cu_count() * 16;
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(sage_finish_z_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, z, (uint32_t)ld, h,
                     (uint32_t)h_ld, deg, inv, (uint32_t)num_dst, (uint32_t)din);
  return launch_status(__func__);
}

extern "C" int fgnn_sage_grad_prep(const float *gz, size_t gz_ld, const float *inv, float *gh, float *gagg,
                                   size_t num_dst, size_t num_src, size_t din, void *stream) {
  if (num_src == 0) return FGNN_OK;
  if (!gz || !inv || !gh || !gagg || din == 0 || din % 4 || gz_ld < 2 * din || gz_ld % 4 || num_dst > num_src ||
      num_src > 0xffffffffull || gz_ld > 0xffffffffull || reinterpret_cast<uintptr_t>(gz) % 16 ||
      reinterpret_cast<uintptr_t>(gh) % 16 || reinterpret_cast<uintptr_t>(gagg) % 16)
    return FGNN_EINVAL;
  const size_t total = num_src * (din / 4);
  size_t blocks = div_up(total, (size_t)kBlock);
  const size_t most = (size_t)device_cu_count() * 16;
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(sage_grad_prep_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), gz,
                     (uint32_t)gz_ld, inv, gh, gagg, (uint32_t)num_dst, (uint32_t)num_src, (uint32_t)din);
  return launch_status(__func__);
}

extern "C" int fgnn_relu_dropout(const float *x, float *y, size_t n, float p, uint64_t seed,
                                 const unsigned long long *d_step, uint32_t layer_tag, void *stream) {
  if (n == 0) return FGNN_OK;
  if (!x || !y || n % 4 || !(p >= 0.f && p < 1.f) || reinterpret_cast<uintptr_t>(x) % 16 ||
      reinterpret_cast<uintptr_t>(y) % 16)
    return FGNN_EINVAL;
  size_t blocks = div_up(n / 4, (size_t)kBlock);
  const size_t most = (size_t)device_cu_count() * 16;
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(relu_dropout_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, y,
                     n / 4, p, 1.0f / (1.0f - p), seed, d_step, layer_tag);
  return launch_status(__func__);
}

extern "C" int fgnn_relu_dropout_backward(const float *y, const float *gy, float *gx, size_t n, float p, void *stream) {
  if (n == 0) return FGNN_OK;
  if (!y || !gy || !gx || n % 4 || !(p >= 0.f && p < 1.f) || reinterpret_cast<uintptr_t>(y) % 16 ||
      reinterpret_cast<uintptr_t>(gy) % 16 || reinterpret_cast<uintptr_t>(gx) % 16)
    return FGNN_EINVAL;
  size_t blocks = div_up(n / 4, (size_t)kBlock);
  const size_t most = (size_t)device_cu_count() * 16;
  if (blocks > most) blocks = most;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), y,
                     gy, gx, n / 4, 1.0f / (1.0f - p));
  return launch_status(__func__);
}

static constexpr size_t kXentBlocks = 512;
extern "C" size_t fgnn_softmax_xent_scratch_bytes(size_t n) {
  (void)n;
  return (kXentBlocks + 4) * sizeof(float);  // arrival counter | a partial sum per workgroup
}

extern "C" int fgnn_softmax_xent(const float *logits, size_t ld, const long long *labels, size_t n, size_t num_class,
                                 float *loss, float *dlogits, size_t dl_ld, void *ws, size_t ws_bytes, void *stream) {
  if (!logits || !labels || !loss || !dlogits || !ws || n == 0 || num_class == 0 || ld < num_class || dl_ld < num_class ||
      n > 0xffffffffull || ld > 0xffffffffull || dl_ld > 0xffffffffull || ws_bytes < fgnn_softmax_xent_scratch_bytes(n))
    return FGNN_EINVAL;
  // scratch: arrival counter (zero between launches: the caller zeroes it once, the kernel restores it) | partial sums
  uint32_t *arrive = static_cast<uint32_t *>(ws);
  float *row_loss = static_cast<float *>(ws) + 4;
  size_t blocks = div_up(n, (size_t)kWavesPerBlock);
  if (blocks > kXentBlocks) blocks = kXentBlocks;  // (the grid is a function of n alone: the sum's order is fixed)
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), logits, (uint32_t)ld, labels, (uint32_t)n, (uint32_t)num_class,
                     row_loss, loss, dlogits, (uint32_t)dl_ld, arrive);
  return launch_status(__func__);
}

extern "C" int fgnn_adam_step(float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const size_t *numel, int count, float lr, float beta1,
                              float beta2, float eps, float weight_decay, unsigned long long *d_step, void *stream) {
  if (count <= 0 || count > 8 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !d_step) return FGNN_EINVAL;
  AdamTensors t;
  size_t most = 0;
  for (int k = 0; k < count; ++k) {
    if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]) return FGNN_EINVAL;
    t.p[k] = params[k];
    t.g[k] = grads[k];
    t.m[k] = exp_avg[k];
    t.v[k] = exp_avg_sq[k];
    t.n[k] = numel[k];
    if (numel[k] > most) most = numel[k];
  }
  t.count = count;
  size_t blocks = div_up(most ? most : 1, (size_t)kBlock);
  const size_t cap = (size_t)device_cu_count() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), t, lr, beta1,
                     beta2, eps, weight_decay, d_step);
  return launch_status(__func__);
}
