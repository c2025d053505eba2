// capi.hip -- small entry points of libfgnn_hip.so that are not tied to one kernel file.
#include <atomic>

#include "fgnn_device.h"

#include <cstdio>

namespace fgnn {
static thread_local char g_last_error[512] = "";
void set_last_error(const char *what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}
uint32_t *&scan_error_sink() {
  static thread_local uint32_t *sink = nullptr;
  return sink;
}
static unsigned long long *g_phase_log = nullptr;
unsigned long long *phase_log_base() { return g_phase_log; }
// one device word per (process, device 0..15) counting the aggregates that waiting workgroups recomputed for tiles that
// had not published in time (scan_prefix_help); never freed
unsigned long long *scan_help_counter() {
  static unsigned long long *g_helps[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_helps[dev]) {
    unsigned long long *p = nullptr;
    if (hipMalloc(&p, sizeof(*p)) != hipSuccess || hipMemset(p, 0, sizeof(*p)) != hipSuccess) return nullptr;
    g_helps[dev] = p;
  }
  return g_helps[dev];
}
static std::atomic<int> g_scan_help_after{-1};
int scan_help_after_override() { return g_scan_help_after.load(std::memory_order_relaxed); }
}  // namespace fgnn

extern "C" void fgnn_debug_set_scan_help_after(int polls) {
  fgnn::g_scan_help_after.store(polls, std::memory_order_relaxed);
}

extern "C" int fgnn_debug_sort_pairs(uint32_t *d_keys, uint32_t *d_vals, size_t n, void *stream) {
  if (n == 0) return FGNN_OK;
  if (!d_keys || !d_vals) return FGNN_EINVAL;
  auto st = static_cast<hipStream_t>(stream);
  uint32_t *tmp = nullptr;
  const size_t words = 2 * n + fgnn::sort_pairs_ws_words(n);
  if (hipMalloc(&tmp, words * sizeof(uint32_t)) != hipSuccess) return FGNN_EHIP;
  uint32_t *sk = nullptr, *sv = nullptr;
  int rc = fgnn::launch_sort_pairs_u32(d_keys, tmp, d_vals, tmp + n, n, tmp + 2 * n, st, &sk, &sv);
  if (rc == FGNN_OK && sk != d_keys) {  // (the small-n paths leave the result in the alternate buffers)
    if (hipMemcpyAsync(d_keys, sk, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_vals, sv, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
      rc = FGNN_EHIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = FGNN_EHIP;
  (void)hipFree(tmp);
  return rc;
}

extern "C" size_t fgnn_debug_phase_log_bytes(void) {
  return (size_t)fgnn::kPhaseLogKinds * fgnn::kPhaseLogTiles * 8 * sizeof(unsigned long long);
}
extern "C" void fgnn_debug_phase_log(unsigned long long *d_buf) { fgnn::g_phase_log = d_buf; }

// a foreign tenant for the co-residency stress test: `workgroups` x 256 threads that hold their wave slots for `usec`
// microseconds of the 100 MHz wall clock and do nothing else (bounded: every wave exits by itself)
__global__ __launch_bounds__(256) void debug_occupy_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int fgnn_debug_occupy(size_t workgroups, unsigned usec, void *stream) {
  if (workgroups == 0 || workgroups > (1u << 20) || usec > 2000000u) return FGNN_EINVAL;
  hipLaunchKernelGGL(debug_occupy_kernel, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (unsigned long long)usec * 100ull);
  return fgnn::launch_status(__func__);
}

// What the memory system sustains for the sampling chain's access pattern, measured where and when the caller runs
// (bench.py's roofline_sample): `num_items` INDEPENDENT random 4-byte reads from a caller's array (large: the CSR),
// four in flight per lane, nothing else.  The same probe as tools/probe/rand_probe.hip's read4.
__device__ __forceinline__ unsigned long long probe_mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__global__ __launch_bounds__(256) void random_read_probe_kernel(const uint32_t *__restrict__ a, unsigned long long n_elems,
                                                                unsigned long long n_items, unsigned long long salt,
                                                                uint32_t *sink) {
  constexpr int K = 4;
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (unsigned long long base = i * K; base < n_items; base += (unsigned long long)gridDim.x * 256 * K) {
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = a[probe_mix(base + k + salt) % n_elems];
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
  }
  if (acc == 0x12345677u) sink[0] = acc;  // (keeps the loads alive; practically never true)
}
extern "C" int fgnn_debug_random_reads(const uint32_t *array, size_t num_elems, size_t num_items, uint64_t salt,
                                       uint32_t *d_sink, void *stream) {
  if (!array || !d_sink || num_elems == 0 || num_items == 0) return FGNN_EINVAL;
  size_t grid = (num_items + 1023) / 1024;
  if (grid > (1u << 20)) grid = 1u << 20;
  hipLaunchKernelGGL(random_read_probe_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), array,
                     (unsigned long long)num_elems, (unsigned long long)num_items, (unsigned long long)salt, d_sink);
  return fgnn::launch_status(__func__);
}

extern "C" unsigned long long fgnn_debug_scan_helps(void) {
  unsigned long long v = 0, *p = fgnn::scan_help_counter();
  if (!p || hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return v;
}

extern "C" const char *fgnn_last_error(void) { return fgnn::g_last_error; }

extern "C" const char *fgnn_version(void) { return "fgnn-hip 0.1 (gfx950)"; }

extern "C" int fgnn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Largest layout any call needs: one uint32 per item + one per workgroup + slack.
extern "C" size_t fgnn_scratch_bytes(size_t n_cap) {
  return (n_cap + fgnn::div_up(n_cap, 64) + 64) * sizeof(uint32_t);  // items + one sum per >=64-item workgroup
}
