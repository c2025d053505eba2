// rand_probe.hip -- measures what the MI355X memory system sustains for the access patterns of the sampling chain:
// independent random 4-byte reads (CSR picks, cache-table lookups), random 8-byte reads and 8-byte CAS (dedup
// table), as a function of the array size (L2 / Infinity Cache / HBM / TLB reach) and of loads in flight per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/rand_probe tools/probe/rand_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int K, typename T>
__global__ __launch_bounds__(256) void read_kernel(const T *a, uint64_t n_elems, uint64_t n_items, uint64_t salt, T *sink) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  T acc = 0;
  for (uint64_t base = i * K; base < n_items; base += (uint64_t)gridDim.x * 256 * K) {
    T v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = a[mix(base + k + salt) % n_elems];
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
  }
  if (acc == (T)0x1234567) sink[0] = acc;
}

template <int K>
__global__ __launch_bounds__(256) void cas_kernel(unsigned long long *a, uint64_t n_elems, uint64_t n_items, uint64_t salt) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (uint64_t base = i * K; base < n_items; base += (uint64_t)gridDim.x * 256 * K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint64_t h = mix(base + k + salt);
      atomicCAS(&a[h % n_elems], ~0ull, h);
    }
  }
}

template <typename F>
static float time_us(F f, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const uint64_t max_bytes = 32ull << 30;
  void *buf;
  CK(hipMalloc(&buf, max_bytes));
  CK(hipMemset(buf, 0xff, max_bytes));
  uint32_t *sink;
  CK(hipMalloc(&sink, 64));
  const uint64_t sizes[] = {16ull << 20, 64ull << 20, 512ull << 20, 8ull << 30, 32ull << 30};
  const uint64_t items[] = {500000, 4000000};
  uint64_t salt = 1;
  printf("pattern,array_MiB,items,K,grid,us,G_per_s\n");
  for (uint64_t sz : sizes)
    for (uint64_t n : items) {
      for (int K : {1, 4, 8}) {
        const int grid = (int)((n + 256ull * K - 1) / (256ull * K));
        float us = 0;
        auto run4 = [&] {
          salt += n;
          if (K == 1) hipLaunchKernelGGL((read_kernel<1, uint32_t>), dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, sz / 4, n, salt, sink);
          if (K == 4) hipLaunchKernelGGL((read_kernel<4, uint32_t>), dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, sz / 4, n, salt, sink);
          if (K == 8) hipLaunchKernelGGL((read_kernel<8, uint32_t>), dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, sz / 4, n, salt, sink);
        };
        us = time_us(run4);
        printf("read4,%llu,%llu,%d,%d,%.1f,%.2f\n", (unsigned long long)(sz >> 20), (unsigned long long)n, K, grid, us, n / us * 1e-3);
        auto run8 = [&] {
          salt += n;
          if (K == 1) hipLaunchKernelGGL((read_kernel<1, uint64_t>), dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, sz / 8, n, salt, (uint64_t *)sink);
          if (K == 4) hipLaunchKernelGGL((read_kernel<4, uint64_t>), dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, sz / 8, n, salt, (uint64_t *)sink);
          if (K == 8) hipLaunchKernelGGL((read_kernel<8, uint64_t>), dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, sz / 8, n, salt, (uint64_t *)sink);
        };
        us = time_us(run8);
        printf("read8,%llu,%llu,%d,%d,%.1f,%.2f\n", (unsigned long long)(sz >> 20), (unsigned long long)n, K, grid, us, n / us * 1e-3);
        if (sz <= (512ull << 20)) {
          auto runc = [&] {
            salt += n;
            if (K == 1) hipLaunchKernelGGL((cas_kernel<1>), dim3(grid), dim3(256), 0, 0, (unsigned long long *)buf, sz / 8, n, salt);
            if (K == 4) hipLaunchKernelGGL((cas_kernel<4>), dim3(grid), dim3(256), 0, 0, (unsigned long long *)buf, sz / 8, n, salt);
            if (K == 8) hipLaunchKernelGGL((cas_kernel<8>), dim3(grid), dim3(256), 0, 0, (unsigned long long *)buf, sz / 8, n, salt);
          };
          us = time_us(runc, 3);
          printf("cas8,%llu,%llu,%d,%d,%.1f,%.2f\n", (unsigned long long)(sz >> 20), (unsigned long long)n, K, grid, us, n / us * 1e-3);
        }
      }
    }
  // cold vs warm: the same 500 K random 4-byte reads from a 444 MB table (its own allocation, like the cache
  // table) with and without other kernels touching tens of GB in between (what the sampling chain does)
  {
    uint32_t *tab;
    const uint64_t tab_elems = 111059956;
    CK(hipMalloc(&tab, tab_elems * 4));
    CK(hipMemset(tab, 0xff, tab_elems * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e30f, sum = 0;
      const int reps = 8;
      for (int r = 0; r < reps; ++r) {
        salt += 500000;
        if (mode == 1)  // thrash: 4 M random reads over 32 GB
          hipLaunchKernelGGL((read_kernel<4, uint32_t>), dim3(3907), dim3(256), 0, 0, (const uint32_t *)buf, max_bytes / 4, 4000000ull, salt * 7, sink);
        if (mode == 2)  // thrash: stream-write 1 GB
          CK(hipMemsetAsync(buf, 0xff, 1ull << 30, 0));
        if (mode == 3) {  // both
          hipLaunchKernelGGL((read_kernel<4, uint32_t>), dim3(3907), dim3(256), 0, 0, (const uint32_t *)buf, max_bytes / 4, 4000000ull, salt * 7, sink);
          CK(hipMemsetAsync(buf, 0xff, 1ull << 30, 0));
        }
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((read_kernel<1, uint32_t>), dim3(1954), dim3(256), 0, 0, (const uint32_t *)tab, tab_elems, 500000ull, salt, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        sum += ms;
      }
      printf("table444MB_500K_mode%d(0 back-to-back,1 after 4M random reads over 32GB,2 after 1GB memset,3 both),%.1f,%.1f\n", mode, best * 1e3f, sum / reps * 1e3f);
    }
  }
  // empty-kernel floor
  auto nop = [&] { hipLaunchKernelGGL((read_kernel<1, uint32_t>), dim3(1), dim3(256), 0, 0, (const uint32_t *)buf, 1024, 0, 0, sink); };
  printf("empty,0,0,0,1,%.1f,0\n", time_us(nop));
  return 0;
}
