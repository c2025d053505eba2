// eng_dynamic.cc -- arch4 with the dynamic feature cache (the reference's prototype, `_cache_policy = dynamic_cache`).
//
// Reference: cuda_loops_arch4.cc:56-97,136-187 (sample sub-loop + cache copy sub-loop), DoGPUSampleDyCache
// (cuda_loops.cc:269-498), DoDynamicCacheFeatureCopy (cuda_loops.cc:1124-1289), GPUDynamicCacheManager
// (cuda_cache_manager_host.cc:171-207, cuda_cache_manager_device.cu:212-246,444-515,632-708).
//
// What the prototype does, per batch:
//  * sampling as usual down to layer 1; after layer 1's dedup the hash table additionally takes ALL neighbours of every
//    node seen so far (GPUExtractNeighbour + FillWithDupMutable): that node list -- a superset of whatever layer 0 can
//    sample -- is the batch's input_nodes, known before layer 0 has been sampled, so the feature extraction can start
//    one layer early; layer 0 is sampled afterwards without inserting anything, and its edges are mapped through the
//    table at the end;
//  * the feature cache is the PREVIOUS batch's feature tensor: a direct-map table node -> row of that tensor splits
//    input_nodes into hits and misses (GetMissCacheIndex), misses come from host memory, hits from the previous
//    tensor, and the table is then re-pointed at the batch just extracted (ReplaceCacheGPU).
// The reference hands the batch to its copy thread after layer 1 and samples layer 0 meanwhile; here one thread walks
// the same steps in order (the result is the same; the overlap is left to the caller's pipelining of whole batches).
// Order of the table's new nodes: first occurrence in the neighbour list, which fgnn_extract_neighbour emits in input
// order (the reference's tile-internal order permutes the local ids of layer 0's new nodes, nothing else).
#include <algorithm>
#include <cstring>

#include "eng_engine.h"

namespace sam {

namespace {

// grows, never shrinks: the path is not allocation-free like the batch driver (nor is the reference's)
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  template <typename T>
  T *Reserve(size_t count) {
    const size_t bytes = count * sizeof(T);
    if (bytes > cap) {
      if (p) (void)hipFree(p);
      cap = bytes + bytes / 4 + 256;
      SAM_HIP(hipMalloc(&p, cap));
    }
    return static_cast<T *>(p);
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

std::shared_ptr<void> Pooled(DevicePool &pool, size_t bytes) {
  void *p = pool.Alloc(bytes ? bytes : 16);
  return std::shared_ptr<void>(p, [&pool](void *q) { pool.Free(q); });
}

}  // namespace

struct DynamicCache {
  fgnn_hashtable *ht = nullptr;
  uint32_t *table = nullptr;  // sampler device: node -> row of prev_feat, or EMPTY
  DevBuf src[FGNN_MAX_LAYERS], dst[FGNN_MAX_LAYERS], mapped[FGNN_MAX_LAYERS], nbrs, ws, nbr_ws, idx[4], counters;
  DevicePool sampler_pool;    // per-batch arrays that stay on the sampler device (input / output nodes)
  std::shared_ptr<void> prev_nodes, prev_feat;
  size_t num_prev = 0;
  ~DynamicCache() {
    prev_nodes.reset();
    prev_feat.reset();
    if (ht) fgnn_hashtable_destroy(ht);
    if (table) (void)hipFree(table);
  }
};

void DynamicCacheDeleter::operator()(DynamicCache *p) const { delete p; }

void Engine::InitDynamicCache() {
  // cuda_engine.cc:170-178: the manager exists iff the policy is kDynamicCache and no static cache is configured
  SAM_CHECK(!RC().UseGPUCache()) << "dynamic_cache takes no cache_percentage: the cache is the previous batch";
  SAM_CHECK(RC().fanout.size() >= 2) << "dynamic_cache needs at least two layers (the prefetch list is built after "
                                        "layer 1, cuda_loops.cc:400-421)";
  switch (RC().sample_type) {  // cuda_loops.cc:333-357: everything else is CHECK(0)
    case kKHop0: case kKHop1: case kWeightedKHop: break;
    default: SAM_FATAL << "dynamic_cache supports khop0, khop1 and weighted_khop sampling only";
  }
  SAM_HIP(hipSetDevice(device_));
  dyn_.reset(new DynamicCache());
  int err = 0;
  // whole neighbourhoods go into the table: it is sized for the graph, not for sampled frontiers
  dyn_->ht = fgnn_hashtable_create(ds_.num_node, &err);
  SAM_CHECK(dyn_->ht) << "fgnn_hashtable_create failed: " << err << " " << fgnn_last_error();
  SAM_HIP(hipMalloc(&dyn_->table, ds_.num_node * sizeof(uint32_t)));
  SAM_HIP(hipMemset(dyn_->table, 0xFF, ds_.num_node * sizeof(uint32_t)));
}

void Engine::SampleOnceDynamic() {
  while (pool_->Full()) {
    if (shutdown_) return;
    std::this_thread::sleep_for(std::chrono::microseconds(1));
  }
  DynamicCache &dc = *dyn_;
  SAM_HIP(hipSetDevice(device_));
  Timer t0;
  const uint32_t *d_batch = nullptr;
  size_t bsize = 0;
  if (!shuffler_->GetBatch(&d_batch, &bsize)) {
    std::this_thread::sleep_for(std::chrono::microseconds(1));
    return;
  }
  const uint64_t key = BatchKey(shuffler_->Epoch(), shuffler_->Step());
  const double shuffle_time = t0.Passed();
  Timer t_sample;
  const size_t L = RC().fanout.size();
  auto b = std::make_shared<GraphBatch>();
  b->key = key;
  b->num_layer = (int)L;

  // ---- DoGPUSampleDyCache ----------------------------------------------------------------------------------------
  SAM_FGNN(fgnn_hashtable_reset(dc.ht, stream_));
  SAM_FGNN(fgnn_hashtable_fill_unique(dc.ht, d_batch, bsize, stream_));
  const uint32_t *unique = fgnn_hashtable_n2o(dc.ht);
  const uint32_t *d_num_unique = fgnn_hashtable_d_num_items(dc.ht);
  size_t *d_num_out = dc.counters.Reserve<size_t>(2);
  auto read_u32 = [&](const uint32_t *d) {
    uint32_t v = 0;
    SAM_HIP(hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, stream_));
    SAM_HIP(hipStreamSynchronize(stream_));
    return (size_t)v;
  };
  auto read_u64 = [&](const size_t *d) {
    size_t v = 0;
    SAM_HIP(hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, stream_));
    SAM_HIP(hipStreamSynchronize(stream_));
    return v;
  };
  size_t num_input = bsize, num_edge[FGNN_MAX_LAYERS] = {0}, total_edges = 0, num_all = 0;
  for (int i = (int)L - 1; i >= 0; --i) {
    const size_t fanout = RC().fanout[i], cap = std::max<size_t>(num_input * fanout, 1);
    uint32_t *out_src = dc.src[i].Reserve<uint32_t>(cap), *out_dst = dc.dst[i].Reserve<uint32_t>(cap);
    uint32_t *mapped = dc.mapped[i].Reserve<uint32_t>(cap);
    // the seed's position in `unique` is its local id: the src half of GPUMapEdges (cuda_loops.cc:461-476) is free
    switch (RC().sample_type) {
      case kKHop0: {
        const size_t wb = fgnn_scratch_bytes(num_input);
        SAM_FGNN(fgnn_sample_khop0(d_indptr_, d_indices_, unique, num_input, nullptr, num_input, fanout, out_src,
                                   out_dst, d_num_out, FGNN_SRC_LOCAL, RC().seed, key, (uint32_t)i,
                                   dc.ws.Reserve<char>(wb), wb, stream_));
        break;
      }
      case kKHop1: {
        const size_t wb = fgnn_weighted_scratch_bytes(num_input, fanout);
        SAM_FGNN(fgnn_sample_khop1(d_indptr_, d_indices_, unique, num_input, nullptr, num_input, fanout, out_src,
                                   out_dst, d_num_out, FGNN_SRC_LOCAL, RC().seed, key, (uint32_t)i,
                                   dc.ws.Reserve<char>(wb), wb, stream_));
        break;
      }
      default: {
        const size_t wb = fgnn_weighted_scratch_bytes(num_input, fanout);
        SAM_FGNN(fgnn_sample_weighted_khop(d_indptr_, d_indices_, d_prob_, d_alias_, unique, num_input, nullptr,
                                           num_input, fanout, out_src, out_dst, d_num_out, FGNN_SRC_LOCAL, RC().seed,
                                           key, (uint32_t)i, dc.ws.Reserve<char>(wb), wb, stream_));
        break;
      }
    }
    const size_t num_samples = read_u64(d_num_out);  // cuda_loops.cc:360-363
    SAM_CHECK_LE(num_samples, cap);
    size_t num_unique;
    if (i == 0) {
      // every possible neighbour is in the table already (:395-399); the dst half of GPUMapEdges (:461-476)
      num_unique = num_all;
      SAM_FGNN(fgnn_hashtable_map(dc.ht, out_dst, num_samples, nullptr, std::max<size_t>(num_samples, 1), mapped,
                                  stream_));
    } else {
      const size_t wb = fgnn_scratch_bytes(std::max<size_t>(num_samples, 1));
      SAM_FGNN(fgnn_hashtable_fill_duplicates(dc.ht, out_dst, num_samples, nullptr, std::max<size_t>(num_samples, 1),
                                              mapped, dc.ws.Reserve<char>(wb), wb, stream_));
      num_unique = read_u32(d_num_unique);
      if (i == 1) {
        // :400-421: all neighbours of everything seen so far -> the batch's input nodes
        size_t *d_nn = d_num_out + 1;
        const size_t eb = fgnn_extract_neighbour_scratch_bytes(num_unique);
        // scratch of its own: the emit launch below is still reading it when the dedup's scratch is (re)sized
        char *ews = dc.nbr_ws.Reserve<char>(eb);
        SAM_FGNN(fgnn_extract_neighbour(d_indptr_, d_indices_, unique, num_unique, nullptr, num_unique, nullptr, 0, d_nn,
                                        ews, eb, stream_));
        const size_t num_nbrs = read_u64(d_nn);
        uint32_t *nbrs = dc.nbrs.Reserve<uint32_t>(std::max<size_t>(num_nbrs, 1));
        SAM_FGNN(fgnn_extract_neighbour(d_indptr_, d_indices_, unique, num_unique, nullptr, num_unique, nbrs, num_nbrs,
                                        d_nn, ews, eb, stream_));
        const size_t fb = fgnn_scratch_bytes(std::max<size_t>(num_nbrs, 1));
        SAM_FGNN(fgnn_hashtable_fill_duplicates(dc.ht, nbrs, num_nbrs, nullptr, std::max<size_t>(num_nbrs, 1), nullptr,
                                                dc.ws.Reserve<char>(fb), fb, stream_));
        num_all = read_u32(d_num_unique);
      }
    }
    b->graphs[i].num_src = num_unique;
    b->graphs[i].num_dst = num_input;
    b->graphs[i].num_edge = num_edge[i] = num_samples;
    total_edges += num_samples;
    num_input = num_unique;
  }
  // node lists stay on the sampler device (adapter.cc:150-192)
  auto in_nodes = Pooled(dc.sampler_pool, num_all * sizeof(uint32_t));
  auto out_nodes = Pooled(dc.sampler_pool, bsize * sizeof(uint32_t));
  SAM_HIP(hipMemcpyAsync(in_nodes.get(), unique, num_all * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  SAM_HIP(hipMemcpyAsync(out_nodes.get(), d_batch, bsize * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  const double sample_time = t_sample.Passed();

  // ---- DoDynamicCacheFeatureCopy: index split on the sampler GPU ------------------------------------------------------
  Timer t_copy;
  uint32_t *idx[4];
  for (int k = 0; k < 4; ++k) idx[k] = dc.idx[k].Reserve<uint32_t>(std::max<size_t>(num_all, 1));
  uint32_t *d_counts = reinterpret_cast<uint32_t *>(d_num_out);  // both layer counters have been read
  const size_t sb = fgnn_scratch_bytes(std::max<size_t>(num_all, 1));
  SAM_FGNN(fgnn_get_miss_cache_index(dc.table, static_cast<const uint32_t *>(in_nodes.get()), num_all, nullptr,
                                     std::max<size_t>(num_all, 1), idx[0], idx[1], idx[2], idx[3], d_counts,
                                     dc.ws.Reserve<char>(sb), sb, stream_));
  uint32_t counts[2] = {0, 0};
  SAM_HIP(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, stream_));
  SAM_HIP(hipStreamSynchronize(stream_));
  const size_t num_miss = counts[0], num_cache = counts[1];
  SAM_CHECK_EQ(num_miss + num_cache, num_all);  // cuda_loops.cc:1172

  // ---- the trainer GPU: graph copy (DoGraphCopy), index copy, combine, labels ----------------------------------------
  SAM_HIP(hipSetDevice(tdevice_));
  auto to_trainer = [&](const void *src, size_t bytes) {
    auto d = Pooled(dev_pool_, bytes);
    if (bytes) SAM_HIP(hipMemcpyAsync(d.get(), src, bytes, hipMemcpyDefault, tstream_));
    b->shared.push_back(d);
    return static_cast<uint32_t *>(d.get());
  };
  size_t graph_bytes = 0;
  for (size_t l = 0; l < L; ++l) {
    b->graphs[l].row = to_trainer(dc.mapped[l].p, num_edge[l] * sizeof(uint32_t));
    b->graphs[l].col = to_trainer(dc.src[l].p, num_edge[l] * sizeof(uint32_t));
    graph_bytes += num_edge[l] * 8;
  }
  const uint32_t *t_idx[4] = {to_trainer(idx[0], num_miss * 4), to_trainer(idx[1], num_miss * 4),
                              to_trainer(idx[2], num_cache * 4), to_trainer(idx[3], num_cache * 4)};
  const uint32_t *t_out = to_trainer(out_nodes.get(), bsize * sizeof(uint32_t));
  const size_t row_bytes = ds_.feat_dim * 4;
  auto feat = Pooled(dev_pool_, num_all * row_bytes);
  if (num_miss)   // ExtractMissData + copy + CombineMissData (:1222-1251), the host fetch done by the gather itself
    SAM_FGNN(fgnn_gather_rows_masked(feat.get(), dev_host_feat_, t_idx[0], t_idx[1], num_miss, nullptr, num_miss,
                                     ds_.feat_dim, FGNN_F32, FeatRowMask(), tstream_));
  if (num_cache)  // CombineCacheData (:1253-1259): rows of the previous batch's tensor
    SAM_FGNN(fgnn_gather_rows(feat.get(), dc.prev_feat.get(), t_idx[2], t_idx[3], num_cache, nullptr, num_cache,
                              ds_.feat_dim, FGNN_F32, tstream_));
  auto lab = Pooled(dev_pool_, bsize * 8);
  SAM_FGNN(fgnn_gather_rows(lab.get(), d_label_, t_out, nullptr, bsize, nullptr, std::max<size_t>(bsize, 1), 1, FGNN_I64,
                            tstream_));
  SAM_HIP(hipStreamSynchronize(tstream_));

  // ---- ReplaceCacheGPU (:1261-1265): this batch is the next one's cache ----------------------------------------------
  SAM_HIP(hipSetDevice(device_));
  SAM_FGNN(fgnn_cache_table_replace(dc.table, static_cast<const uint32_t *>(dc.prev_nodes.get()), dc.num_prev,
                                    static_cast<const uint32_t *>(in_nodes.get()), num_all, stream_));
  SAM_HIP(hipStreamSynchronize(stream_));
  dc.prev_nodes = in_nodes;
  dc.num_prev = num_all;
  dc.prev_feat = feat;

  b->feat = feat.get();
  b->feat_rows = num_all;
  b->label = lab.get();
  b->input_nodes = static_cast<const uint32_t *>(in_nodes.get());
  b->output_nodes = static_cast<const uint32_t *>(out_nodes.get());
  b->num_input = num_all;
  b->num_output = bsize;
  b->input_device = b->output_device = device_;
  b->device = tdevice_;
  b->shared.push_back(feat);
  b->shared.push_back(lab);
  b->shared.push_back(in_nodes);
  b->shared.push_back(out_nodes);
  pool_->Submit(b);

  const double copy_time = t_copy.Passed();
  auto &P = Profiler::Get();
  P.LogStep(key, kLogL1NumSample, (double)total_edges);
  P.LogStep(key, kLogL1NumNode, (double)num_all);
  P.LogStep(key, kLogL1SampleTime, shuffle_time + sample_time);
  P.LogStep(key, kLogL2ShuffleTime, shuffle_time);
  P.LogStep(key, kLogL2CoreSampleTime, sample_time);
  P.LogStep(key, kLogL1CopyTime, copy_time);
  P.LogStep(key, kLogL2CacheCopyTime, copy_time);
  P.LogStep(key, kLogL1FeatureBytes, (double)num_all * row_bytes);
  P.LogStep(key, kLogL1MissBytes, (double)num_miss * row_bytes);
  P.LogStep(key, kLogL1LabelBytes, (double)bsize * 8);
  P.LogStep(key, kLogL1GraphBytes, (double)graph_bytes);
  P.LogEpochAdd(key, kLogEpochSampleTime, shuffle_time + sample_time);
  P.LogEpochAdd(key, kLogEpochCopyTime, copy_time);
  P.LogEpochAdd(key, kLogEpochFeatureBytes, (double)num_all * row_bytes);
  P.LogEpochAdd(key, kLogEpochMissBytes, (double)num_miss * row_bytes);
}

}  // namespace sam
