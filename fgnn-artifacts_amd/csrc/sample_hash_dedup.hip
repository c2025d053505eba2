// sample_hash_dedup.hip -- weighted sampling WITHOUT repeated values: alias-method draws, a draw whose value the
// seed has already selected is rejected, until `fanout` distinct neighbours are found.
//
// Replaces GPUSampleWeightedKHopHashDedup (reference samgraph/common/cuda/
// cuda_sampling_weighted_khop_hash_dedup.cu:41-282: one thread per seed with a 100-word private table in scratch
// memory, padded output, count_edge, DeviceScan, compact_edge).  Bit-identical to oracle
// fgnn_oracle_sample_weighted_khop_hash_dedup, including its termination rule (the reference spins forever on a
// row with fewer than `fanout` distinct selectable values).
//
// MI355X design: ONE launch.  A workgroup owns 256 consecutive seeds, one lane per seed; the selected values of a
// seed live in LDS (column `lane` of sel[fanout][256], conflict-free), where the rejection test scans them and
// from where they are emitted -- no padded temporaries, no count / scan / compact kernels: the output offset of a
// workgroup comes from the cross-workgroup prefix of fgnn_device.h (a workgroup waits only for lower-numbered ones,
// which were dispatched before it: safe next to other streams' kernels).  Frontiers of more than kMaxScanTiles
// workgroups take two launches of the same kernel instead (count, one-workgroup scan, emit).
#include "fgnn_device.h"

namespace fgnn {
namespace {

constexpr uint32_t kMaxScanTiles = 1536;  // cap of every waiting grid (fgnn_device.h): 192 workgroups per XCD
constexpr uint32_t kMaxFanout = 50;  // the reference's per-thread table has 50 slots (hash_dedup.cu:43,72)
__host__ __device__ constexpr uint32_t max_attempts(uint32_t fanout) { return 64u * fanout; }

__device__ __forceinline__ float uniform_float(uint32_t x) {
  return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1], 24 bits, as the oracle
}

__global__ __launch_bounds__(kBlock) void hash_dedup_kernel(const uint32_t *__restrict__ indptr,
                                                            const uint32_t *__restrict__ indices,
                                                            const float *__restrict__ prob,
                                                            const uint32_t *__restrict__ alias,
                                                            const uint32_t *__restrict__ input, size_t n_host,
                                                            const uint32_t *d_n, size_t cap, uint32_t F,
                                                            uint32_t *__restrict__ out_src,
                                                            uint32_t *__restrict__ out_dst, int src_mode,
                                                            uint64_t seed, uint64_t batch_key, uint32_t tag,
                                                            ScanWs scan, size_t *d_num_out, int mode,
                                                            uint32_t *__restrict__ block_sums) {
  // mode 0: single pass (offsets by look-back); 1: count only (block_sums[tile] = edges of the tile);
  // 2: emit with the scanned block_sums as offsets
  extern __shared__ uint32_t sel[];  // [F][kBlock]
  __shared__ uint32_t sh[kWavesPerBlock];
  __shared__ uint32_t sh_tile;
  const uint32_t n = (uint32_t)resolve_count(n_host, d_n, cap);
  const uint32_t ntiles = n ? (n - 1) / kBlock + 1 : 1u;
  const int tid = threadIdx.x;
  const uint32_t tile = mode == 0 ? scan_take_tile(scan, &sh_tile) : blockIdx.x;
  if (tile >= ntiles) {
    if (mode == 1 && tid == 0) block_sums[tile] = 0;
    return;
  }
  {
    const uint32_t i = tile * kBlock + tid;
    uint32_t rid = 0, c = 0;
    if (i < n) {
      rid = input[i];
      const uint32_t off = indptr[rid];
      const uint32_t len = indptr[rid + 1] - off;
      if (len <= F) {
        for (uint32_t j = 0; j < len; ++j) sel[j * kBlock + tid] = indices[off + j];
        c = len;
      } else {
        u32x4 blk{0, 0, 0, 0};
        const uint32_t limit = max_attempts(F);
        for (uint32_t a = 0; a < limit && c < F; ++a) {
          if ((a & 1u) == 0) blk = philox_block(seed, batch_key, tag, i, a >> 1);  // draws 2a, 2a+1
          const uint32_t k = ((a & 1u) ? blk.z : blk.x) % len;
          const float r = uniform_float((a & 1u) ? blk.w : blk.y);
          uint32_t v = indices[off + k];
          if (r > prob[off + k]) v = alias[off + k];
          bool seen = false;
          for (uint32_t q = 0; q < c; ++q) seen |= sel[q * kBlock + tid] == v;
          if (!seen) {
            sel[c * kBlock + tid] = v;
            ++c;
          }
        }
      }
    }
    uint32_t total;
    const uint32_t lo = block_exclusive_scan<kWavesPerBlock>(c, sh, &total);
    if (mode == 1) {
      if (tid == 0) block_sums[tile] = total;
      return;
    }
    const size_t base = mode == 0 ? (size_t)scan_lookback(scan, tile, total, &sh_tile) : (size_t)block_sums[tile];
    if (mode == 0 && tile == ntiles - 1 && tid == 0 && d_num_out) *d_num_out = base + total;
    const uint32_t src = src_mode == FGNN_SRC_LOCAL ? i : rid;
    for (uint32_t j = 0; j < c; ++j) {
      out_src[base + lo + j] = src;
      out_dst[base + lo + j] = sel[j * kBlock + tid];
    }
  }
}

}  // namespace

int sample_hash_dedup(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                      const uint32_t *alias_table, const uint32_t *input, size_t num_input, const uint32_t *d_num_input,
                      size_t num_input_cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out,
                      int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream, ScanWsHost *scan_host) {
  auto st = static_cast<hipStream_t>(stream);
  size_t cap = d_num_input ? num_input_cap : num_input;
  if (fanout == 0 || fanout > kMaxFanout) return FGNN_EINVAL;
  if (cap == 0) {
    if (d_num_out) FGNN_HIP_CHECK(hipMemsetAsync(d_num_out, 0, sizeof(size_t), st));
    return FGNN_OK;
  }
  if (!indptr || !indices || !prob_table || !alias_table || !input || cap * fanout >= 0x7fffffffull) return FGNN_EINVAL;
  const uint32_t F = (uint32_t)fanout;
  const uint32_t tag = ((uint32_t)FGNN_WEIGHTED_KHOP_HASH_DEDUP << 8) | (layer & 0xffu);
  const size_t lds = (size_t)F * kBlock * sizeof(uint32_t);
  const size_t nb = div_up(cap, (size_t)kBlock);
  ScanWs scan{nullptr, nullptr, 0, 0, nullptr, nullptr, 0, kScanHelpAfterPolls, nullptr};
  uint32_t *sums = nullptr;
  int mode = 0;
  if (nb > kMaxScanTiles) {
    if (ws_bytes < (nb + 2) * sizeof(uint32_t)) return FGNN_ENOSPC;
    sums = static_cast<uint32_t *>(ws);
    mode = 1;
  } else if (scan_host && nb <= scan_host->ws.max_tiles) {
    scan = scan_host->next(0, nb, /*use_ticket=*/true);
  } else {
    // stateless entry point: descriptors and ticket counter in the caller's scratch, zeroed, generation 1
    if (ws_bytes < (nb + 1) * sizeof(unsigned long long)) return FGNN_ENOSPC;
    FGNN_HIP_CHECK(hipMemsetAsync(ws, 0, (nb + 1) * sizeof(unsigned long long), st));
    scan.desc = static_cast<unsigned long long *>(ws);
    scan.gen = 1;
    scan.max_tiles = (uint32_t)nb;
    scan.ticket = reinterpret_cast<uint32_t *>(scan.desc + nb);
    scan.ticket_base = 0;
  }
#define FGNN_HD(MODE)                                                                                              \
  hipLaunchKernelGGL(hash_dedup_kernel, dim3(nb), dim3(kBlock), lds, st, indptr, indices, prob_table, alias_table, \
                     input, num_input, d_num_input, cap, F, out_src, out_dst, src_mode, seed, batch_key, tag, scan, \
                     d_num_out, MODE, sums)
  if (mode == 0) {
    FGNN_HD(0);
    if (scan_host && scan.ticket && scan.desc == scan_host->ws.desc) {
      // a launch the runtime refused drew no tickets: hand them back, or the slot's next ticketed launch would compute
      // its tiles against a base the device counter never reached
      const int rc = launch_status(__func__);
      if (rc != FGNN_OK) scan_host->unnext_tickets(nb);
      return rc;
    }
  } else {
    FGNN_HD(1);
    if (launch_scan_block_sums(sums, nb, d_num_out, nullptr, nullptr, nullptr, st) != FGNN_OK) return FGNN_EHIP;
    FGNN_HD(2);
  }
#undef FGNN_HD
  return launch_status(__func__);
}

}  // namespace fgnn

extern "C" size_t fgnn_hash_dedup_scratch_bytes(size_t num_input_cap) {
  return (fgnn::div_up(num_input_cap, (size_t)fgnn::kBlock) + 2) * sizeof(unsigned long long);
}

extern "C" int fgnn_sample_weighted_khop_hash_dedup(const uint32_t *indptr, const uint32_t *indices,
                                                    const float *prob_table, const uint32_t *alias_table,
                                                    const uint32_t *input, size_t num_input,
                                                    const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                                    uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out,
                                                    int src_mode, uint64_t seed, uint64_t batch_key, uint32_t layer,
                                                    void *ws, size_t ws_bytes, void *stream) {
  return fgnn::sample_hash_dedup(indptr, indices, prob_table, alias_table, input, num_input, d_num_input,
                                 num_input_cap, fanout, out_src, out_dst, d_num_out, src_mode, seed, batch_key, layer,
                                 ws, ws_bytes, stream, nullptr);
}
