// pack.hip -- serialises one sampled batch into a queue slot with a single kernel.
// Wire format == reference task_queue.cc:68-88,154-227:
//   TransData{have_data,num_layer,key,input_size,output_size,num_miss}
//   [input_nodes] output_nodes [miss_src miss_dst] [cache_src cache_dst]
//   { GraphData{num_src,num_dst,num_edge} row col [data] } x num_layer
// All sizes are read from the device-side batch summary; the slot is host memory mapped into the
// GPU's address space, so the stores travel over the host link while the sampler keeps going.
// With a device-ring slot (eng_queue.h) the arrays are written there instead -- same layout, same
// offsets -- and only the headers (TransData, GraphData) go to the host slot, where the receiver's
// CPU parses them.
#include "eng_queue.h"

namespace sam {
namespace {

constexpr int kPackBlock = 256;

// A message is a dozen arrays of very different sizes.  Copying them one after the other makes every lane wait for one
// load -> store round trip per array (16-19 us for a 5.5 MB papers100M message, whatever the grid); instead the arrays
// form ONE flat index space: a lane finds the array of each of its words in a small table in LDS and keeps eight
// independent loads in flight.  Arrays start at arbitrary 4-byte offsets of the message, so words it is.
constexpr int kMaxCopySegments = 2 + 4 + 3 * FGNN_MAX_LAYERS;
struct CopyTable {
  uint32_t *dst[kMaxCopySegments];
  const uint32_t *src[kMaxCopySegments];
  size_t begin[kMaxCopySegments + 1];  // flat index of each segment's first word; begin[n] = total
  int n;
};

__device__ __forceinline__ void table_add(CopyTable &t, uint32_t *dst, const uint32_t *src, size_t words) {
  if (words == 0) return;
  t.dst[t.n] = dst;
  t.src[t.n] = src;
  t.begin[t.n + 1] = t.begin[t.n] + words;
  ++t.n;
}

// all threads, after the table is complete and a barrier
__device__ __forceinline__ void copy_flat(const CopyTable &t) {
  const size_t total = t.begin[t.n];
  const size_t stride = (size_t)gridDim.x * kPackBlock;
  constexpr int U = 8;
  for (size_t i0 = (size_t)blockIdx.x * kPackBlock + threadIdx.x; i0 < total; i0 += U * stride) {
    uint32_t v[U];
    uint32_t *d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      d[u] = nullptr;
      v[u] = 0;
      if (i < total) {
        int s = 0;
        while (i >= t.begin[s + 1]) ++s;  // <= a dozen entries
        const size_t off = i - t.begin[s];
        d[u] = t.dst[s] + off;
        v[u] = t.src[s][off];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (d[u]) *d[u] = v[u];
  }
}

__global__ __launch_bounds__(kPackBlock) void pack_kernel(PackArgs a) {
  __shared__ CopyTable tab;
  // header words (TransData, one GraphData per layer) and where they go, as word offsets from the start of a slot: they
  // are staged here by one lane and stored by ONE wave instruction per destination (the host slot is memory across the
  // host link)
  constexpr int kHdrWords = (sizeof(TransData) + FGNN_MAX_LAYERS * sizeof(GraphData)) / 4;
  __shared__ uint32_t hw[kHdrWords];
  __shared__ uint32_t hoff[kHdrWords];
  __shared__ int hn;
  // The batch summary is read field by field through the pointer, by one lane: a by-value copy indexed with the layer
  // number lives in scratch memory (168 bytes per lane), and a launch that needs scratch costs ~15 us of set-up on top
  // of its work -- that, not the copy, was this kernel's duration (16-19 us for a 5.5 MB message)
  const fgnn_batch_meta *m = a.d_meta;
  // this kernel closes the batch: its start time goes into the summary (t_closed) ...
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    const_cast<fgnn_batch_meta *>(m)->t_closed = wall_clock64();
  if (threadIdx.x == 0) {
    const uint32_t num_layers = m->num_layers, num_input = m->num_input, num_output = m->num_output, num_miss = m->num_miss;
    // total size first: an oversized message is flagged, never written past the slot
    const size_t num_cache = (size_t)num_input - num_miss;
    size_t words = num_output;
    if (a.ship_input) words += num_input;
    if (a.ship_cache_index) words += 2 * (size_t)num_miss + 2 * num_cache;
    size_t bytes = sizeof(TransData) + words * sizeof(uint32_t);
    for (uint32_t l = 0; l < num_layers; ++l)
      bytes += sizeof(GraphData) + (size_t)m->num_edge[l] * (a.have_data ? 3 : 2) * sizeof(uint32_t);
    const bool fits = bytes <= a.slot_bytes && m->overflow == 0;
    const uint64_t t64[4] = {m->key, num_input, num_output, num_miss};
    hw[0] = a.have_data != 0;                            // bool + padding
    hw[1] = (uint32_t)(fits ? (int)num_layers : -1);     // -1: the receiver aborts (CHECK_LE at task_queue.cc:162)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hw[2 + 2 * k] = (uint32_t)t64[k];
      hw[3 + 2 * k] = (uint32_t)(t64[k] >> 32);
    }
    for (int k = 0; k < 10; ++k) hoff[k] = (uint32_t)k;
    int n = 10;
    // arrays go to the payload slot (device ring) or, without one, to the host slot; same layout and offsets
    uint32_t *base = a.payload ? static_cast<uint32_t *>(a.payload) : static_cast<uint32_t *>(a.slot);
    uint32_t *p = base + sizeof(TransData) / 4;
    tab.n = 0;
    tab.begin[0] = 0;
    if (fits) {
      if (a.ship_input) { table_add(tab, p, a.input_nodes, num_input); p += num_input; }
      table_add(tab, p, a.output_nodes, num_output);
      p += num_output;
      if (a.ship_cache_index) {
        table_add(tab, p, a.cidx[0], num_miss); p += num_miss;
        table_add(tab, p, a.cidx[1], num_miss); p += num_miss;
        table_add(tab, p, a.cidx[2], num_cache); p += num_cache;
        table_add(tab, p, a.cidx[3], num_cache); p += num_cache;
      }
      // GraphData headers hold size_t fields: the payload before them is a multiple of 4 bytes only, so
      // they are written as 32-bit halves (the reference writes them through a misaligned pointer)
      for (uint32_t l = 0; l < num_layers; ++l) {
        const size_t ne = m->num_edge[l];
        const uint64_t v[3] = {m->num_src[l], m->num_dst[l], ne};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          hw[n] = (uint32_t)v[k];
          hoff[n++] = (uint32_t)(p - base) + 2 * k;
          hw[n] = (uint32_t)(v[k] >> 32);
          hoff[n++] = (uint32_t)(p - base) + 2 * k + 1;
        }
        p += sizeof(GraphData) / sizeof(uint32_t);
        table_add(tab, p, a.row[l], ne); p += ne;
        table_add(tab, p, a.col[l], ne); p += ne;
        if (a.have_data) { table_add(tab, p, a.data[l], ne); p += ne; }
      }
    }
    hn = n;
    if (a.msg_words) *a.msg_words = fits ? (uint32_t)(p - base) : 0u;  // SAMGRAPH_HANDOFF_CHECK: the checksum kernel's length
  }
  __syncthreads();
  // ... and the summary goes to the host from here (every other kernel that writes it is earlier in the stream; the
  // stamp above is this workgroup's own store, ordered by the barrier): no copy command behind this launch
  if (a.h_meta && blockIdx.x == gridDim.x - 1 && threadIdx.x < sizeof(fgnn_batch_meta) / 4)
    a.h_meta[threadIdx.x] = reinterpret_cast<const volatile uint32_t *>(m)[threadIdx.x];
  if (blockIdx.x == 0 && (int)threadIdx.x < hn) {
    // the host slot always gets the headers (the receiver's CPU parses them there); the device-ring slot gets them
    // too: it is a complete message (copied back to the host slot as it is when a receiver cannot map the ring)
    static_cast<uint32_t *>(a.slot)[hoff[threadIdx.x]] = hw[threadIdx.x];
    if (a.payload) static_cast<uint32_t *>(a.payload)[hoff[threadIdx.x]] = hw[threadIdx.x];
  }
  copy_flat(tab);
}

__global__ __launch_bounds__(kPackBlock) void unpack_kernel(UnpackArgs a) {
  __shared__ CopyTable tab;
  if (threadIdx.x == 0) {
    tab.n = 0;
    tab.begin[0] = 0;
    for (int k = 0; k < a.num_segments; ++k) table_add(tab, a.seg[k].dst, a.seg[k].src, a.seg[k].words);
  }
  __syncthreads();
  copy_flat(tab);
}

// one workgroup: sum of word[i] * (2 i + 1) mod 2^64 -- a dropped, repeated, shifted or swapped word changes it
constexpr int kSumBlock = 1024;
__global__ __launch_bounds__(kSumBlock) void message_checksum_kernel(uint32_t *msg, const uint32_t *d_words,
                                                                     size_t words_host, int verify, uint32_t *d_result) {
  __shared__ unsigned long long part[kSumBlock / 64];
  const size_t words = d_words ? *d_words : words_host;
  unsigned long long s = 0;
  for (size_t i = threadIdx.x; i < words; i += kSumBlock) s += (unsigned long long)msg[i] * (2ull * i + 1ull);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kSumBlock / 64; ++w) t += part[w];
    if (!verify) {
      if (words) {
        msg[words] = (uint32_t)t;
        msg[words + 1] = (uint32_t)(t >> 32);
      }
    } else if (words == 0 || msg[words] != (uint32_t)t || msg[words + 1] != (uint32_t)(t >> 32)) {
      atomicOr(d_result, 1u);
    }
  }
}

}  // namespace

int LaunchMessageChecksum(uint32_t *msg, const uint32_t *d_words, size_t words, int verify, uint32_t *d_result,
                          hipStream_t stream) {
  hipLaunchKernelGGL(message_checksum_kernel, dim3(1), dim3(kSumBlock), 0, stream, msg, d_words, words, verify, d_result);
  return hipGetLastError() == hipSuccess ? FGNN_OK : FGNN_EHIP;
}

int LaunchUnpack(const UnpackArgs &a, hipStream_t stream) {
  size_t words = 0;
  for (int k = 0; k < a.num_segments; ++k) words += a.seg[k].words;
  if (words == 0) return FGNN_OK;
  size_t blocks = (words + kPackBlock * 4 - 1) / (kPackBlock * 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(kPackBlock), 0, stream, a);
  return hipGetLastError() == hipSuccess ? FGNN_OK : FGNN_EHIP;
}

int LaunchPack(const PackArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(kPackBlock), 0, stream, a);
  return hipGetLastError() == hipSuccess ? FGNN_OK : FGNN_EHIP;
}

}  // namespace sam
