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

__device__ __forceinline__ void copy_words(uint32_t *dst, const uint32_t *src, size_t n) {
  const size_t stride = (size_t)gridDim.x * kPackBlock;
  for (size_t i = (size_t)blockIdx.x * kPackBlock + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(kPackBlock) void pack_kernel(PackArgs a) {
  const fgnn_batch_meta m = *a.d_meta;
  TransData *hdr = static_cast<TransData *>(a.slot);
  // total size first: an oversized message is flagged, never written past the slot
  const size_t num_cache = (size_t)m.num_input - m.num_miss;
  size_t words = m.num_output;
  if (a.ship_input) words += m.num_input;
  if (a.ship_cache_index) words += 2 * (size_t)m.num_miss + 2 * num_cache;
  size_t bytes = sizeof(TransData) + words * sizeof(uint32_t);
  for (uint32_t l = 0; l < m.num_layers; ++l)
    bytes += sizeof(GraphData) + m.num_edge[l] * (a.have_data ? 3 : 2) * sizeof(uint32_t);
  const bool fits = bytes <= a.slot_bytes && m.overflow == 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the device-ring slot gets the headers too: it is a complete message (copied back to the host slot as it is
    // when a receiver cannot map the ring)
    for (TransData *h : {hdr, static_cast<TransData *>(a.payload)}) {
      if (!h) continue;
      h->have_data = a.have_data != 0;
      h->num_layer = fits ? (int)m.num_layers : -1;  // -1: the receiver aborts (CHECK_LE at task_queue.cc:162)
      h->key = m.key;
      h->input_size = m.num_input;
      h->output_size = m.num_output;
      h->num_miss = m.num_miss;
    }
  }
  if (!fits) return;
  // arrays go to the payload slot; `hp` walks the host slot in step for the GraphData headers
  const bool split = a.payload != nullptr;
  uint32_t *p = split ? static_cast<TransData *>(a.payload)->data : hdr->data;
  uint32_t *hp = hdr->data;
  const uint32_t *p_begin = p;
  if (a.ship_input) { copy_words(p, a.input_nodes, m.num_input); p += m.num_input; }
  copy_words(p, a.output_nodes, m.num_output);
  p += m.num_output;
  if (a.ship_cache_index) {
    if (m.num_miss) {
      copy_words(p, a.cidx[0], m.num_miss); p += m.num_miss;
      copy_words(p, a.cidx[1], m.num_miss); p += m.num_miss;
    }
    if (num_cache) {
      copy_words(p, a.cidx[2], num_cache); p += num_cache;
      copy_words(p, a.cidx[3], num_cache); p += num_cache;
    }
  }
  // GraphData headers hold size_t fields: the payload before them is a multiple of 4 bytes only, so
  // they are written as two 32-bit halves (the reference writes them through a misaligned pointer)
  for (uint32_t l = 0; l < m.num_layers; ++l) {
    const size_t ne = m.num_edge[l];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const uint64_t v[3] = {m.num_src[l], m.num_dst[l], ne};
      uint32_t *h = hp + (p - p_begin);  // the same offset in the host slot
      for (int k = 0; k < 3; ++k) {
        h[2 * k] = p[2 * k] = (uint32_t)v[k];
        h[2 * k + 1] = p[2 * k + 1] = (uint32_t)(v[k] >> 32);
      }
    }
    p += sizeof(GraphData) / sizeof(uint32_t);
    copy_words(p, a.row[l], ne); p += ne;
    copy_words(p, a.col[l], ne); p += ne;
    if (a.have_data) { copy_words(p, a.data[l], ne); p += ne; }
  }
}

__global__ __launch_bounds__(kPackBlock) void unpack_kernel(UnpackArgs a) {
  for (int k = 0; k < a.num_segments; ++k) copy_words(a.seg[k].dst, a.seg[k].src, a.seg[k].words);
}

}  // namespace

int LaunchUnpack(const UnpackArgs &a, hipStream_t stream) {
  size_t words = 0;
  for (int k = 0; k < a.num_segments; ++k) words += a.seg[k].words;
  if (words == 0) return FGNN_OK;
  size_t blocks = (words + kPackBlock * 4 - 1) / (kPackBlock * 4);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(kPackBlock), 0, stream, a);
  return hipGetLastError() == hipSuccess ? FGNN_OK : FGNN_EHIP;
}

int LaunchPack(const PackArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(pack_kernel, dim3(256), dim3(kPackBlock), 0, stream, a);
  return hipGetLastError() == hipSuccess ? FGNN_OK : FGNN_EHIP;
}

}  // namespace sam
