// sample_random_walk.hip -- restart random walks + per-seed visit-frequency top-K (PinSAGE).
//
// Replaces GPUSampleRandomWalk (reference samgraph/common/cuda/cuda_sampling_random_walk.cu:43-161)
// and FrequencyHashmap::GetTopK (cuda_frequency_hashmap.cu:1143-1367: a global {key,count,index}
// hash table with one 512-bucket segment per seed, 11 timed steps, a 64-bit radix sort and two
// device scans).  Bit-identical to oracle fgnn_oracle_sample_random_walk; ties between equal visit
// counts are broken by first visit position (a legal outcome of the reference's CAS race).
//
// MI355X design: a seed's whole multiset of visits is tiny (num_walks * walk_len <= a few hundred
// entries), so the frequency count and the top-K selection never leave the CU: a group of W lanes of a
// wavefront owns one seed (W = num_walks rounded up to a power of two: 16 seeds per wave at 4 walks, 2 at 25,
// one at > 32), lanes walk in parallel (lane = walk), the visited ids go to LDS, each lane then ranks one
// visit entry against the others by broadcast LDS reads -- no global hash table, no sort, no
// atomics, no reset pass.  Edge offsets come from per-seed counts + one scan; the padded top-K
// records are compacted by a second small kernel.
#include <cstdlib>

#include "fgnn_device.h"

namespace fgnn {
namespace {

constexpr int kRwWaves = 4;  // waves per workgroup

__device__ __forceinline__ double uniform_double(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// lanes a seed gets: the smallest power of two >= num_walks (64 when num_walks is larger: lanes then loop over walks)
inline uint32_t rw_group_width(size_t num_walks) {
  uint32_t w = 1;
  while (w < num_walks && w < (uint32_t)kWave) w <<= 1;
  return w;
}

// A wavefront owns G = 64 / W seeds, W = rw_group_width(num_walks): PinSAGE's reference default of 4 walks per seed
// (multi_gpu/train_pinsage.py:130-134) puts 16 seeds into a wave instead of idling 60 of its 64 lanes, 25 walks put 2.
// Dynamic LDS: kRwWaves * G * 2P words (P = num_walks * walk_len): per seed its visited ids + representative counts.
__global__ __launch_bounds__(kRwWaves *kWave) void random_walk_topk_kernel(
    const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ input,
    size_t n_host, const uint32_t *d_n, size_t cap, uint32_t walk_len, double restart_prob, uint32_t num_walks,
    uint32_t K, uint32_t W, uint32_t *__restrict__ pad_dst, uint32_t *__restrict__ pad_cnt,
    uint32_t *__restrict__ seed_cnt, uint64_t seed, uint64_t batch_key, uint32_t tag) {
  extern __shared__ uint32_t dyn[];
  const uint32_t P = num_walks * walk_len;
  const uint32_t G = (uint32_t)kWave / W;
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t g = lane / W, sl = lane % W;  // seed of the wave, lane within the seed's group
  uint32_t *visited = dyn + ((size_t)wave_id() * G + g) * 2 * P;
  uint32_t *repcnt = visited + P;
  const size_t n = resolve_count(n_host, d_n, cap);
  const size_t i = ((size_t)blockIdx.x * kRwWaves + wave_id()) * G + g;
  const bool valid = i < n;
  const unsigned long long gmask = (W == (uint32_t)kWave ? ~0ull : ((1ull << W) - 1ull)) << (g * W);
  const uint32_t start = valid ? input[i] : FGNN_EMPTY_KEY;

  // ---- walks: one lane per walk (cuda_sampling_random_walk.cu:57-101) ----
  for (uint32_t walk = sl; walk < num_walks; walk += W) {
    uint32_t node = start;
    for (uint32_t step = 0; step < walk_len; ++step) {
      const uint32_t pos = step * num_walks + walk;
      uint32_t v = FGNN_EMPTY_KEY;
      if (node != FGNN_EMPTY_KEY) {
        const uint32_t off = indptr[node];
        const uint32_t len = indptr[node + 1] - off;
        if (len == 0) {
          node = FGNN_EMPTY_KEY;
        } else {
          const uint32_t d = walk * walk_len + step;
          const u32x4 blk = philox_block(seed, batch_key, tag, (uint32_t)i, d >> 1);  // draws 2d, 2d+1
          const uint32_t r0 = (d & 1u) ? blk.z : blk.x;
          const uint32_t r1 = (d & 1u) ? blk.w : blk.y;
          node = indices[off + r0 % len];
          v = node;
          if (uniform_double(r1) < restart_prob) node = FGNN_EMPTY_KEY;
        }
      }
      visited[pos] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  // ---- compaction: walks end early (restart, dead ends), so about half of the P slots hold no visit; the two all-pairs
  //      phases below cost O(m^2) on the m real visits instead of O(P^2).  Stable (first-visit order is the tie-break of
  //      the top-K), in place: a round reads its W slots before it writes, and writes land at or below what it read ----
  uint32_t m = 0;
  for (uint32_t p0 = 0; p0 < P; p0 += W) {
    const uint32_t p = p0 + sl;
    const uint32_t v = p < P ? visited[p] : FGNN_EMPTY_KEY;
    const unsigned long long live = __ballot(v != FGNN_EMPTY_KEY) & gmask;
    if (v != FGNN_EMPTY_KEY) visited[m + (uint32_t)__popcll(live & ((1ull << lane) - 1ull))] = v;
    m += (uint32_t)__popcll(live);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  // ---- frequency: one lane per visit entry; repcnt[p] = visit count if p is the FIRST visit of its
  //      destination (its representative), else 0 ----
  uint32_t kept = 0;
  for (uint32_t p0 = 0; p0 < m; p0 += W) {
    const uint32_t p = p0 + sl;
    uint32_t c = 0;
    if (p < m) {
      const uint32_t me = visited[p];
      bool rep = true;
      for (uint32_t q = 0; q < m; ++q) {
        const bool eq = visited[q] == me;
        c += eq;
        if (eq && q < p) rep = false;
      }
      if (!rep) c = 0;
      repcnt[p] = c;
    }
    kept += (uint32_t)__popcll(__ballot(c != 0) & gmask);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  // ---- top-K: rank of a representative by (count desc, first position asc) ----
  for (uint32_t p0 = 0; p0 < m; p0 += W) {
    const uint32_t p = p0 + sl;
    if (p >= m || !valid) continue;
    const uint32_t count = repcnt[p];
    if (count == 0) continue;
    uint32_t rank = 0;
    for (uint32_t q = 0; q < m; ++q) {
      const uint32_t qc = repcnt[q];
      rank += (qc > count) || (qc == count && q < p);
    }
    if (rank < K) {
      pad_dst[i * K + rank] = visited[p];
      pad_cnt[i * K + rank] = count;
    }
  }
  if (sl == 0 && i < cap) seed_cnt[i] = valid ? (kept < K ? kept : K) : 0u;
}

__global__ __launch_bounds__(kBlock) void rw_sums_kernel(const uint32_t *__restrict__ seed_cnt, size_t cap,
                                                         uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t c = i < cap ? seed_cnt[i] : 0u;
  uint32_t tot;
  (void)block_exclusive_scan<kWavesPerBlock>(c, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// compact_output_revised (cuda_frequency_hashmap.cu:644-676)
__global__ __launch_bounds__(kBlock) void rw_emit_kernel(const uint32_t *__restrict__ input,
                                                         const uint32_t *__restrict__ seed_cnt, size_t cap, uint32_t K,
                                                         const uint32_t *__restrict__ pad_dst,
                                                         const uint32_t *__restrict__ pad_cnt,
                                                         const uint32_t *__restrict__ block_offsets,
                                                         uint32_t *__restrict__ out_src, uint32_t *__restrict__ out_dst,
                                                         uint32_t *__restrict__ out_data, int src_mode) {
  __shared__ uint32_t sh[kWavesPerBlock];
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t c = i < cap ? seed_cnt[i] : 0u;
  uint32_t tot;
  const uint32_t lo = block_exclusive_scan<kWavesPerBlock>(c, sh, &tot);
  if (c == 0) return;
  const size_t w = (size_t)block_offsets[blockIdx.x] + lo;
  const uint32_t src = src_mode == FGNN_SRC_LOCAL ? (uint32_t)i : input[i];
  for (uint32_t k = 0; k < c; ++k) {
    out_src[w + k] = src;
    out_dst[w + k] = pad_dst[i * K + k];
    out_data[w + k] = pad_cnt[i * K + k];
  }
}

// rw_sums_kernel -> scan -> rw_emit_kernel in ONE launch (batch driver): a workgroup owns 256 x IPT consecutive seeds, its
// output offset is the prefix over the earlier workgroups' edge counts (fgnn_device.h, single-pass prefix with helping:
// a missing tile's count is re-summed from seed_cnt, which nothing writes while this kernel runs)
template <int IPT>
__global__ __launch_bounds__(kBlock) void rw_emit_sp_kernel(const uint32_t *__restrict__ input,
                                                            const uint32_t *__restrict__ seed_cnt, size_t cap, uint32_t K,
                                                            const uint32_t *__restrict__ pad_dst,
                                                            const uint32_t *__restrict__ pad_cnt,
                                                            uint32_t *__restrict__ out_src, uint32_t *__restrict__ out_dst,
                                                            uint32_t *__restrict__ out_data, int src_mode, ScanWs scan,
                                                            size_t *d_num_out) {
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t sh_tile[2];
  const uint32_t tile = blockIdx.x;
  auto tile_sum = [&](uint32_t tl, uint32_t *c) -> uint32_t {
    const size_t i0 = ((size_t)tl * kBlock + threadIdx.x) * IPT;
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const uint32_t cc = i0 + q < cap ? seed_cnt[i0 + q] : 0u;
      if (c) c[q] = cc;
      sum += cc;
    }
    return sum;
  };
  uint32_t c[IPT];
  const uint32_t sum = tile_sum(tile, c);
  uint32_t tot;
  const uint32_t lo = block_exclusive_scan<kWavesPerBlock>(sum, sh, &tot);
  scan_publish_aggregate(scan, tile, tot);
  const uint32_t before = scan_prefix_help(scan, tile, sh_tile, [&](uint32_t m) -> uint32_t {
    uint32_t tot_m;
    (void)block_exclusive_scan<kWavesPerBlock>(tile_sum(m, nullptr), sh, &tot_m);
    return tot_m;
  });
  if (tile == gridDim.x - 1 && threadIdx.x == 0 && d_num_out) *d_num_out = (size_t)before + tot;
  size_t w = (size_t)before + lo;
  const size_t i0 = ((size_t)tile * kBlock + threadIdx.x) * IPT;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const size_t i = i0 + q;
    const uint32_t src = src_mode == FGNN_SRC_LOCAL ? (uint32_t)i : (c[q] ? input[i] : 0u);
    for (uint32_t k = 0; k < c[q]; ++k) {
      out_src[w + k] = src;
      out_dst[w + k] = pad_dst[i * K + k];
      out_data[w + k] = pad_cnt[i * K + k];
    }
    w += c[q];
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

// scratch: pad_dst[cap*K] | pad_cnt[cap*K] | seed_cnt[cap] | sums[nb+1]
extern "C" size_t fgnn_random_walk_scratch_bytes(size_t num_input_cap, size_t K) {
  return (2 * num_input_cap * K + num_input_cap + div_up(num_input_cap, kBlock) + 8) * sizeof(uint32_t);
}

extern "C" int fgnn_sample_random_walk(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                       size_t num_input, const uint32_t *d_num_input, size_t num_input_cap,
                                       size_t walk_len, double restart_prob, size_t num_walks, size_t K,
                                       uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data, size_t *d_num_out,
                                       int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws,
                                       size_t ws_bytes, void *stream) {
  return fgnn::sample_random_walk_ex(indptr, indices, input, num_input, d_num_input, num_input_cap, walk_len,
                                     restart_prob, num_walks, K, out_src, out_dst, out_data, d_num_out, src_mode, seed,
                                     batch_key, layer, ws, ws_bytes, stream, nullptr);
}

int fgnn::sample_random_walk_ex(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input, size_t num_input,
                                const uint32_t *d_num_input, size_t num_input_cap, size_t walk_len, double restart_prob,
                                size_t num_walks, size_t K, uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data,
                                size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer,
                                void *ws, size_t ws_bytes, void *stream, ScanWsHost *scan) {
  auto st = static_cast<hipStream_t>(stream);
  size_t cap = d_num_input ? num_input_cap : num_input;
  if (walk_len == 0 || num_walks == 0 || K == 0) return FGNN_EINVAL;
  const size_t P = walk_len * num_walks;
  if (P > 4096 || K > 0xffff) return FGNN_EINVAL;  // seeds per workgroup * 2P words of LDS (<= 128 KiB)
  if (cap == 0) {
    if (d_num_out) FGNN_HIP_CHECK(hipMemsetAsync(d_num_out, 0, sizeof(size_t), st));
    return FGNN_OK;
  }
  if (!indptr || !indices || !input || !out_src || !out_dst || !out_data || cap * K >= 0x7fffffffull)
    return FGNN_EINVAL;
  if (ws_bytes < fgnn_random_walk_scratch_bytes(cap, K)) return FGNN_ENOSPC;
  const uint32_t tag = ((uint32_t)FGNN_RANDOM_WALK << 8) | (layer & 0xffu);
  uint32_t *pad_dst = static_cast<uint32_t *>(ws);
  uint32_t *pad_cnt = pad_dst + cap * K;
  uint32_t *seed_cnt = pad_cnt + cap * K;
  uint32_t *sums = seed_cnt + cap;
  const size_t nb = div_up(cap, kBlock);
  uint32_t W = rw_group_width(num_walks);
  while (W < (uint32_t)kWave && (size_t)kRwWaves * (kWave / W) * 2 * P * sizeof(uint32_t) > 128 * 1024) W <<= 1;  // long walks
  const size_t seeds_per_wg = (size_t)kRwWaves * (kWave / W);
  const size_t lds = seeds_per_wg * 2 * P * sizeof(uint32_t);
  static bool attr_done = false;
  if (!attr_done) {
    FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&random_walk_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(random_walk_topk_kernel, dim3(div_up(cap, seeds_per_wg)), dim3(kRwWaves * kWave), lds, st, indptr,
                     indices, input, num_input, d_num_input, cap, (uint32_t)walk_len, restart_prob, (uint32_t)num_walks,
                     (uint32_t)K, W, pad_dst, pad_cnt, seed_cnt, seed, batch_key, tag);
  if (scan) {  // batch driver: counts -> offsets -> compacted COO in one launch
    constexpr size_t kSinglePassTiles = 1536;
    const int ipt = nb <= kSinglePassTiles ? 1 : nb <= 4 * kSinglePassTiles ? 4 : nb <= 16 * kSinglePassTiles ? 16 : 0;
    const size_t grid = ipt ? div_up(cap, (size_t)kBlock * ipt) : 0;
    if (ipt && grid <= scan->ws.max_tiles) {
#define FGNN_RW_EMIT(I)                                                                                              \
  hipLaunchKernelGGL((rw_emit_sp_kernel<I>), dim3(grid), dim3(kBlock), 0, st, input, seed_cnt, cap, (uint32_t)K, pad_dst, \
                     pad_cnt, out_src, out_dst, out_data, src_mode, scan->next(0, grid), d_num_out)
      if (ipt == 1) FGNN_RW_EMIT(1);
      else if (ipt == 4) FGNN_RW_EMIT(4);
      else FGNN_RW_EMIT(16);
#undef FGNN_RW_EMIT
      return launch_status(__func__);
    }
  }
  hipLaunchKernelGGL(rw_sums_kernel, dim3(nb), dim3(kBlock), 0, st, seed_cnt, cap, sums);
  if (launch_scan_block_sums(sums, nb, d_num_out, nullptr, nullptr, nullptr, st) != FGNN_OK) return FGNN_EHIP;
  hipLaunchKernelGGL(rw_emit_kernel, dim3(nb), dim3(kBlock), 0, st, input, seed_cnt, cap, (uint32_t)K, pad_dst, pad_cnt,
                     sums, out_src, out_dst, out_data, src_mode);
  return launch_status(__func__);
}
