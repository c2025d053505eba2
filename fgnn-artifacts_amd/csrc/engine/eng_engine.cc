#include "eng_engine.h"

#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstring>

namespace sam {

// ------------------------------------------------------------------------------------------------
// DevicePool / GraphPool

DevicePool::~DevicePool() {
  for (auto &kv : free_) (void)hipFree(kv.second);
}

void *DevicePool::Alloc(size_t bytes) {
  if (bytes == 0) bytes = 256;
  // round to 256 B below 1 MiB, to 1 MiB above: similar sizes recur every step
  const size_t gran = bytes < (1u << 20) ? 256 : (1u << 20);
  bytes = (bytes + gran - 1) / gran * gran;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = free_.lower_bound(bytes);
    if (it != free_.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {
      void *p = it->second;
      live_[p] = it->first;
      free_.erase(it);
      return p;
    }
  }
  void *p = nullptr;
  SAM_HIP(hipMalloc(&p, bytes));
  std::lock_guard<std::mutex> lk(mu_);
  live_[p] = bytes;
  return p;
}

void DevicePool::Free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(mu_);
  auto it = live_.find(p);
  SAM_CHECK(it != live_.end()) << "DevicePool::Free of an unknown pointer";
  free_.emplace(it->second, p);
  live_.erase(it);
}

std::shared_ptr<GraphBatch> GraphPool::Get() {
  while (true) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!q_.empty()) {
        auto b = q_.front();
        q_.pop();
        return b;
      }
      if (stop_) return nullptr;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(1));
  }
}

void GraphPool::Submit(std::shared_ptr<GraphBatch> b) {
  std::lock_guard<std::mutex> lk(mu_);
  if (stop_) return;  // shutdown while the extractor still had a batch in its hands: nobody will ask for it
  q_.push(std::move(b));
}

bool GraphPool::Full() {
  std::lock_guard<std::mutex> lk(mu_);
  return q_.size() >= max_size_;
}

// ------------------------------------------------------------------------------------------------

Engine &Engine::Get() {
  static Engine e;
  return e;
}

static void *DeviceVisible(void *host) {
  void *dev = nullptr;
  SAM_HIP(hipHostGetDevicePointer(&dev, host, 0));
  return dev;
}

void Engine::Init() {
  if (data_initialized_) return;
  SAM_CHECK(RC().is_configured);
  Timer t;
  ds_.Load(RC());
  Profiler::Get().LogInit(kLogInitL2LoadDataset, t.Passed());
  const bool aligned = RC().run_arch == kArch6 || RC().run_arch == kArch7;
  num_step_ = aligned ? Shuffler::AlignedNumStep(ds_.num_train, RC().batch_size, RC().num_worker)
                      : RoundUpDiv(ds_.num_train, RC().batch_size);
  Profiler::Get().Resize(RC().num_epoch, num_step_);
  data_initialized_ = true;
  if (RC().run_arch == kArch1) {
    InitArch1();
  } else if (RC().run_arch == kArch2 || RC().run_arch == kArch3 || RC().run_arch == kArch4) {
    InitInProcess();
  } else if (RC().run_arch == kArch7) {
    InitArch7();
  } else {
    SAM_CHECK(RC().run_arch == kArch5 || RC().run_arch == kArch6);
    // shared queue + sampler barrier, created BEFORE fork (dist_engine.cc:129-153); an arch6 worker's queue only joins
    // the two halves of that worker, so each creates its own after the fork
    Timer tq;
    if (RC().run_arch == kArch5) CreateQueue();
    SharedRegion bp = SharedCreate(sizeof(pthread_barrier_t));
    sampler_barrier_ = static_cast<pthread_barrier_t *>(bp.ptr);
    if (bp.creator) {
      pthread_barrierattr_t attr;
      pthread_barrierattr_init(&attr);
      pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
      pthread_barrier_init(sampler_barrier_, &attr, (unsigned)RC().num_sample_worker);
      SharedPublish(bp.ptr);
    }
    Profiler::Get().LogInit(kLogInitL2DistQueue, tq.Passed());
  }
  Profiler::Get().LogInit(kLogInitL1Common, t.Passed());
}

void Engine::CreateQueue() {
  const bool have_data = RC().sample_type == kRandomWalk;
  size_t slot = MaxMessageBytes(RC().batch_size, RC().fanout.data(), RC().fanout.size(), have_data);
  size_t slots = RC().mq_budget_bytes / slot;
  if (slots > kMaxSlots) slots = kMaxSlots;
  // sampler and extractor in ONE process (arch2-4, arch6): the reference joins them by a TaskQueue of max_sampling_jobs
  // tasks (cuda_engine.cc:142), not by the 170-slot shared-memory queue of arch5 -- and a sampler that may run 170
  // batches ahead parks their payloads in pinned host memory once its few HBM slots are taken (RunConfig::
  // DeviceRingSlots), over the link the extractor's miss rows need.  The extractor keeps kExtractDepth messages in
  // flight and the sampler two: below that depth the two halves would wait for each other's slots
  if (RC().run_arch != kArch5) slots = std::min(slots, std::max<size_t>(RC().max_sampling_jobs, kExtractDepth + 4));
  if (slots < 2) slots = 2;
  mq_ = new MemoryQueue(slot, slots);
}

// batch buffers of the message-producing sampler loop (arch5 sampler process, in-process archs): `slots` buffers rotating
// over `streams` HIP streams -- the chains of consecutive batches overlap on the GPU (fgnn_sampler_sample orders what has
// to stay ordered with events); one batch alone cannot fill the chip.  SAMGRAPH_SAMPLER_STREAMS / _SLOTS override.
void Engine::CreateSamplerSlots(size_t streams, size_t slots) {
  const char *e_slots = getenv("SAMGRAPH_SAMPLER_SLOTS"), *e_streams = getenv("SAMGRAPH_SAMPLER_STREAMS");
  const size_t n_streams = e_streams && atoi(e_streams) > 0 ? (size_t)atoi(e_streams) : streams;
  slots_.resize(e_slots && atoi(e_slots) > 0 ? (size_t)atoi(e_slots) : e_streams ? 3 * n_streams : slots);
  for (size_t i = 0; i < slots_.size(); ++i) {
    Slot &s = slots_[i];
    int err = 0;
    s.fb = fgnn_batch_create(sampler_, 0, FGNN_F32, FGNN_I64, 0, &err);
    SAM_CHECK(s.fb) << "fgnn_batch_create failed: " << err << " " << fgnn_last_error();
    if (i < n_streams) {
      SAM_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    } else {
      s.st = slots_[i % n_streams].st;
      s.owns_st = false;
    }
    SAM_HIP(hipEventCreate(&s.e0));
    SAM_HIP(hipEventCreate(&s.e1));
    SAM_HIP(hipEventCreate(&s.e2));
    if (s.owns_st) shuffler_->TrackStream(s.st);  // batches in flight read the epoch's seed array
  }
}

void Engine::UploadTopology(int device) {
  SAM_HIP(hipSetDevice(device));
  device_ = device;
  SAM_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  SAM_HIP(hipMalloc(&d_indptr_, ds_.indptr.bytes));
  SAM_HIP(hipMalloc(&d_indices_, ds_.indices.bytes ? ds_.indices.bytes : 4));
  SAM_HIP(hipMemcpy(d_indptr_, ds_.indptr.ptr, ds_.indptr.bytes, hipMemcpyHostToDevice));
  if (ds_.indices.bytes) SAM_HIP(hipMemcpy(d_indices_, ds_.indices.ptr, ds_.indices.bytes, hipMemcpyHostToDevice));
  if (RC().sample_type == kWeightedKHopPrefix) {
    SAM_HIP(hipMalloc(&d_prefix_, ds_.prob_prefix.bytes ? ds_.prob_prefix.bytes : 4));
    SAM_HIP(hipMemcpy(d_prefix_, ds_.prob_prefix.ptr, ds_.prob_prefix.bytes, hipMemcpyHostToDevice));
  }
  if (RC().sample_type == kWeightedKHop || RC().sample_type == kWeightedKHopHashDedup) {
    SAM_HIP(hipMalloc(&d_prob_, ds_.prob_table.bytes ? ds_.prob_table.bytes : 4));
    SAM_HIP(hipMalloc(&d_alias_, ds_.alias_table.bytes ? ds_.alias_table.bytes : 4));
    SAM_HIP(hipMemcpy(d_prob_, ds_.prob_table.ptr, ds_.prob_table.bytes, hipMemcpyHostToDevice));
    SAM_HIP(hipMemcpy(d_alias_, ds_.alias_table.ptr, ds_.alias_table.bytes, hipMemcpyHostToDevice));
  }
}

void Engine::CreateSampler() {
  fgnn_sampler_config c;
  memset(&c, 0, sizeof(c));
  c.indptr = d_indptr_;
  c.indices = d_indices_;
  c.prob_prefix = d_prefix_;
  c.prob_table = d_prob_;
  c.alias_table = d_alias_;
  c.num_node = ds_.num_node;
  c.sample_type = RC().sample_type;
  c.num_layers = RC().fanout.size();
  for (size_t i = 0; i < c.num_layers; ++i) c.fanout[i] = RC().fanout[i];
  c.max_batch_size = RC().batch_size;
  c.seed = RC().seed;
  c.walk_len = RC().random_walk_length;
  c.num_walks = RC().num_random_walk;
  c.restart_prob = RC().random_walk_restart_prob;
  int err = 0;
  sampler_ = fgnn_sampler_create(&c, &err);
  SAM_CHECK(sampler_) << "fgnn_sampler_create failed: " << err << " " << fgnn_last_error();
}

void Engine::ReleaseBatch(GraphBatch *b) {
  if (!b) return;
  for (void *p : b->pooled) dev_pool_.Free(p);
  b->pooled.clear();
  for (void *p : b->host_owned) free(p);
  b->host_owned.clear();
  b->shared.clear();
  if (b->fb) {
    for (auto &s : slots_)
      if (s.fb == b->fb) s.busy = false;
    b->fb = nullptr;
  }
}

// ------------------------------------------------------------------------------------------------
// arch1: one GPU samples and extracts (cuda_engine.cc:64-196, cuda_loops_arch1.cc:44-80)

void Engine::InitArch1() { InitSingleGPU(true); }

// arch7 (the reference's "SGNN-DGL" baseline, cuda_engine.cc:102-112,329-332, cuda_loops_arch7.cc:54-84): every worker
// process runs its own engine on its own GPU over an equal share of the train set and only SAMPLES; the script
// gathers features itself (samgraph.torch.load_subtensor), so batches carry neither features nor labels.
void Engine::InitArch7() {
  SAM_CHECK(!RC().UseGPUCache()) << "arch7 has no feature cache (cuda_engine.cc:331)";
  InitSingleGPU(false);
}

void Engine::InitSingleGPU(bool extract) {
  SAM_CHECK(RC().sampler_ctx.IsGPU() && RC().trainer_ctx.IsGPU())
      << "arch1 / arch7 need cuda contexts: the sampling path has no CPU fallback";
  SAM_CHECK_EQ(RC().sampler_ctx.device_id, RC().trainer_ctx.device_id);
  Timer t;
  UploadTopology(RC().sampler_ctx.device_id);
  if (extract) {
    SAM_HIP(hipMalloc(&d_feat_, ds_.feat.bytes));
    SAM_HIP(hipMemcpy(d_feat_, ds_.feat.ptr, ds_.feat.bytes, hipMemcpyHostToDevice));
    SAM_HIP(hipMalloc(&d_label_, ds_.label.bytes));
    SAM_HIP(hipMemcpy(d_label_, ds_.label.ptr, ds_.label.bytes, hipMemcpyHostToDevice));
  }
  CreateSampler();
  const bool aligned = RC().run_arch == kArch7;
  shuffler_.reset(new Shuffler(static_cast<const uint32_t *>(ds_.train_set.ptr), ds_.num_train, RC().num_epoch,
                               RC().batch_size, aligned ? (int)RC().worker_id : 0, aligned ? (int)RC().num_worker : 1,
                               stream_, aligned));
  // the aligned split pads the set with repeated ids and the reference's aligned shuffler has no check either
  if (RC().option_sanity_check && !aligned) shuffler_->EnableSanityCheck(ds_.num_node);
  pool_.reset(new GraphPool(RC().max_copying_jobs));
  slots_.resize(RC().max_copying_jobs + 2);
  for (auto &s : slots_) {
    int err = 0;
    s.fb = fgnn_batch_create(sampler_, extract ? ds_.feat_dim : 0, FGNN_F32, FGNN_I64, 0, &err);
    SAM_CHECK(s.fb) << "fgnn_batch_create failed: " << err << " " << fgnn_last_error();
    if (extract) SAM_FGNN(fgnn_batch_set_feat_row_mask(s.fb, FeatRowMask()));
    SAM_HIP(hipEventCreate(&s.e0));
    SAM_HIP(hipEventCreate(&s.e1));
    SAM_HIP(hipEventCreate(&s.e2));
  }
  Profiler::Get().LogInit(kLogInitL1Sampler, t.Passed());
  initialized_ = true;
}

// ------------------------------------------------------------------------------------------------
// arch2 / arch3 / arch4: one process, sampler on sampler_ctx, copy + extraction + training on trainer_ctx (the same GPU
// in arch2) -- cuda_engine.cc:64-196, cuda_loops_arch{2,3,4}.cc.  Built from the arch5 halves: the sampler half
// serialises each batch into an in-process pinned ring, the extractor half rebuilds it on the trainer GPU and
// gathers the features there (HBM cache rows + miss rows straight from registered host memory).
void Engine::InitInProcess() {
  SAM_CHECK(RC().sampler_ctx.IsGPU() && RC().trainer_ctx.IsGPU())
      << "arch2-4 need cuda contexts: the sampling path has no CPU fallback";
  // the dynamic cache only exists in arch4's loops (cuda_loops_arch4.cc:218-240); arch2 / arch3 ignore the policy
  const bool dynamic = RC().cache_policy == kDynamicCache && RC().run_arch == kArch4;
  if (dynamic) SAM_CHECK(!RC().UseGPUCache()) << "dynamic_cache takes no cache_percentage (cuda_engine.cc:148-178)";
  Timer t;
  if (!dynamic) CreateQueue();
  // sampler half
  UploadTopology(RC().sampler_ctx.device_id);
  if (!dynamic) {
    mq_->PinMemory();
    if (mq_->CreateDeviceRing(0, (uint32_t)RC().DeviceRingSlots())) ring_id_ = 0;
    CreateSampler();
  }
  shuffler_.reset(new Shuffler(static_cast<const uint32_t *>(ds_.train_set.ptr), ds_.num_train, RC().num_epoch,
                               RC().batch_size, 0, 1, stream_));
  if (RC().option_sanity_check) shuffler_->EnableSanityCheck(ds_.num_node);
  // the dynamic-cache loop (eng_dynamic.cc) hands batches over in process, no messages
  if (!dynamic) CreateSamplerSlots(kInProcessSamplerStreams, kInProcessSamplerSlots);
  if (RC().UseGPUCache()) {
    Timer tp;
    if (RC().cache_policy == kCacheByPreSample || RC().cache_policy == kCacheByPreSampleStatic) PreSample();
    Profiler::Get().LogInit(kLogInitL2Presample, tp.Passed());
    Timer tc;
    BuildCacheTable();
    Profiler::Get().LogInit(kLogInitL2BuildCache, tc.Passed());
  }
  Profiler::Get().LogInit(kLogInitL1Sampler, t.Passed());
  // extractor / trainer half
  Timer tt;
  SAM_HIP(hipSetDevice(RC().trainer_ctx.device_id));
  tdevice_ = RC().trainer_ctx.device_id;
  SAM_HIP(hipStreamCreateWithFlags(&tstream_, hipStreamNonBlocking));
  for (auto &x : xctx_) {
    SAM_HIP(hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking));
    for (auto &e : x.ev) SAM_HIP(hipEventCreate(&e));
  }
  SAM_HIP(hipHostRegister(ds_.feat.ptr, ds_.feat.bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  dev_host_feat_ = DeviceVisible(ds_.feat.ptr);
  SAM_HIP(hipMalloc(&d_label_, ds_.label.bytes));
  SAM_HIP(hipMemcpy(d_label_, ds_.label.ptr, ds_.label.bytes, hipMemcpyHostToDevice));
  if (RC().UseGPUCache()) BuildTrainerCache();
  pool_.reset(new GraphPool(RC().max_copying_jobs));
  if (dynamic) InitDynamicCache();
  Profiler::Get().LogInit(kLogInitL1Trainer, tt.Passed());
  initialized_ = true;
}

void Engine::SampleOnceArch1() {
  if (pool_->Full()) {
    std::this_thread::sleep_for(std::chrono::microseconds(1));
    return;
  }
  Timer t0;
  const uint32_t *d_batch = nullptr;
  size_t bsize = 0;
  if (!shuffler_->GetBatch(&d_batch, &bsize)) {
    std::this_thread::sleep_for(std::chrono::microseconds(1));
    return;
  }
  const uint64_t key = BatchKey(shuffler_->Epoch(), shuffler_->Step());
  const double shuffle_time = t0.Passed();
  Slot *s = nullptr;
  for (size_t k = 0; k < slots_.size() && !s; ++k) {
    Slot &c = slots_[(next_slot_ + k) % slots_.size()];
    if (!c.busy) {
      s = &c;
      next_slot_ = (next_slot_ + k + 1) % slots_.size();
    }
  }
  SAM_CHECK(s) << "no free batch buffer: the trainer holds more than max_copying_jobs batches";
  SAM_HIP(hipEventRecord(s->e0, stream_));
  SAM_FGNN(fgnn_sampler_sample(sampler_, d_batch, bsize, key, s->fb, stream_));
  SAM_HIP(hipEventRecord(s->e1, stream_));
  const bool extract = d_feat_ != nullptr;  // arch7 hands the sampled blocks over without features
  if (extract) SAM_FGNN(fgnn_batch_extract(s->fb, d_feat_, d_label_, stream_));
  SAM_HIP(hipEventRecord(s->e2, stream_));
  SAM_FGNN(fgnn_batch_finish(s->fb, stream_));
  fgnn_batch_meta m;
  SAM_FGNN(fgnn_batch_wait(s->fb, &m));
  SAM_CHECK_EQ(m.overflow, 0u);
  s->busy = true;

  auto b = std::make_shared<GraphBatch>();
  b->key = key;
  b->num_layer = (int)m.num_layers;
  for (uint32_t l = 0; l < m.num_layers; ++l) {
    b->graphs[l].row = fgnn_batch_row(s->fb, l);
    b->graphs[l].col = fgnn_batch_col(s->fb, l);
    b->graphs[l].data = fgnn_batch_data(s->fb, l);
    b->graphs[l].num_src = m.num_src[l];
    b->graphs[l].num_dst = m.num_dst[l];
    b->graphs[l].num_edge = m.num_edge[l];
  }
  if (extract) {
    b->feat = fgnn_batch_feat(s->fb);
    b->feat_rows = m.num_input;
    b->label = fgnn_batch_label(s->fb);
  }
  b->input_nodes = fgnn_batch_input_nodes(s->fb);
  b->output_nodes = fgnn_batch_output_nodes(s->fb);
  b->num_input = m.num_input;
  b->num_output = m.num_output;
  b->input_device = b->output_device = b->device = device_;
  b->fb = s->fb;
  pool_->Submit(b);

  float ms_sample = 0, ms_extract = 0;
  (void)hipEventElapsedTime(&ms_sample, s->e0, s->e1);
  (void)hipEventElapsedTime(&ms_extract, s->e1, s->e2);
  auto &P = Profiler::Get();
  size_t edges = 0;
  for (uint32_t l = 0; l < m.num_layers; ++l) edges += m.num_edge[l];
  P.LogStep(key, kLogL1NumSample, (double)edges);
  P.LogStep(key, kLogL1NumNode, (double)m.num_input);
  P.LogStep(key, kLogL1SampleTime, shuffle_time + ms_sample * 1e-3);
  P.LogStep(key, kLogL1CopyTime, ms_extract * 1e-3);
  P.LogStep(key, kLogL2ShuffleTime, shuffle_time);
  P.LogStep(key, kLogL2CoreSampleTime, ms_sample * 1e-3);
  P.LogStep(key, kLogL2ExtractTime, ms_extract * 1e-3);
  if (extract) {
    P.LogStep(key, kLogL1FeatureBytes, (double)m.num_input * ds_.feat_dim * 4);
    P.LogStep(key, kLogL1LabelBytes, (double)m.num_output * 8);
  }
  P.LogEpochAdd(key, kLogEpochSampleTime, shuffle_time + ms_sample * 1e-3);
  P.LogEpochAdd(key, kLogEpochCopyTime, ms_extract * 1e-3);
  P.LogEpochAdd(key, kLogEpochSampleTotalTime, t0.Passed());
}

// ------------------------------------------------------------------------------------------------
// arch5 sampler process (dist_engine.cc:231-364, dist_loops_arch5.cc:60-156)

void Engine::SampleInit(int worker_id, Context ctx) {
  if (initialized_) return;
  SAM_CHECK(data_initialized_) << "samgraph_data_init must run before fork";
  SAM_CHECK(ctx.IsGPU()) << "sampler context must be cuda:N (no CPU sampling path in this build)";
  Timer t;
  dist_type_ = DistType::Sample;
  worker_id_ = worker_id;
  if (const char *e = getenv("SAMGRAPH_HANDOFF_CHECK")) handoff_check_left_ = (size_t)strtoull(e, nullptr, 10);
  const bool arch6 = RC().run_arch == kArch6;
  if (arch6) CreateQueue();  // joins this worker's sampler half and extractor half only
  UploadTopology(ctx.device_id);
  mq_->PinMemory();
  {
    const int ring = arch6 ? 0 : worker_id;  // an arch6 queue is private to its worker
    if (ring < kMaxRings && mq_->CreateDeviceRing(ring, (uint32_t)RC().DeviceRingSlots())) ring_id_ = ring;
  }
  CreateSampler();
  // arch6: equal shares of the padded train set (DistAlignedShuffler, dist_engine.cc:276-281)
  shuffler_.reset(new Shuffler(static_cast<const uint32_t *>(ds_.train_set.ptr), ds_.num_train, RC().num_epoch,
                               RC().batch_size, worker_id, (int)RC().num_sample_worker, stream_, arch6));
  if (RC().option_sanity_check && !arch6) shuffler_->EnableSanityCheck(ds_.num_node);
  pool_.reset(new GraphPool(RC().max_copying_jobs));
  // Batches between sample_once (enqueued) and the publisher thread (completed + published): 6 batch buffers over 2
  // streams.  The chains of consecutive batches overlap on the GPU (fgnn_sampler_sample orders what has to stay ordered
  // with events; one batch alone cannot fill the chip); a batch buffer is reused in stream order, so the host may run
  // further ahead instead of waiting for a publication every call (with 3 buffers the sampler measured 142 us per
  // papers100M batch, 60 of them waiting here, the GPU idle 21 % of the time).  Three streams, nine buffers: with round
  // 3's kernels a third batch in flight only slowed khop2's order chain down (profiles/r03_sampler_streams_sweep.txt:
  // 81.2 us per batch with 2 / 6, 85.4 with 3 / 6); since round 4's partitioned dedup the chain leaves room -- 151
  // batches in 14.7-14.9 ms with 3 / 9 against 15.6-15.8 with 2 / 6, 15.7 with 4 / 12 (profiles/r05_m_sampler_streams_sweep.txt)
  CreateSamplerSlots(3, 9);
  if (RC().UseGPUCache()) {
    Timer tp;
    if (RC().cache_policy == kCacheByPreSample || RC().cache_policy == kCacheByPreSampleStatic) {
      // every worker stops here, not only the one that would have pre-sampled (dist/pre_sampler.cc:87-88)
      SAM_CHECK(RC().cache_policy != kCacheByPreSampleStatic)
          << "kCacheByPreSampleStatic is not implemented in DistEngine now!";
      if (worker_id == 0) PreSample();
      int rc = pthread_barrier_wait(sampler_barrier_);
      SAM_CHECK(rc == 0 || rc == PTHREAD_BARRIER_SERIAL_THREAD);
    }
    Profiler::Get().LogInit(kLogInitL2Presample, tp.Passed());
    Timer tc;
    BuildCacheTable();
    Profiler::Get().LogInit(kLogInitL2BuildCache, tc.Passed());
  }
  Profiler::Get().LogInit(kLogInitL1Sampler, t.Passed());
  initialized_ = true;
}

void Engine::PreSample() {
  if (RC().cache_policy == kCacheByPreSampleStatic) {
    // the multi-process engine has none (dist/pre_sampler.cc:87-88)
    SAM_CHECK(RC().run_arch != kArch5 && RC().run_arch != kArch6)
        << "kCacheByPreSampleStatic is not implemented in DistEngine now!";
    PreSampleStatic();
    return;
  }
  // frequency ranking over presample_epoch epochs of the sampling path (dist/pre_sampler.cc:75-162)
  const int epochs = RC().presample_epoch > 0 ? RC().presample_epoch : 1;
  Shuffler sh(static_cast<const uint32_t *>(ds_.train_set.ptr), ds_.num_train, epochs, RC().batch_size, 0, 1, stream_);
  uint32_t *d_freq = nullptr;
  SAM_HIP(hipMalloc(&d_freq, ds_.num_node * sizeof(uint32_t)));
  SAM_HIP(hipMemsetAsync(d_freq, 0, ds_.num_node * sizeof(uint32_t), stream_));
  fgnn_batch *fb = slots_[0].fb;
  const uint32_t *d_batch;
  size_t bsize;
  const size_t cap = fgnn_sampler_max_nodes(sampler_);
  while (sh.GetBatch(&d_batch, &bsize)) {
    // a key space of its own: the presample draws must not replay the training epochs' draws
    const uint64_t key = (1ull << 63) | (sh.Epoch() * sh.NumStep() + sh.Step());
    SAM_FGNN(fgnn_sampler_sample(sampler_, d_batch, bsize, key, fb, stream_));
    PresampleCount(d_freq, fgnn_batch_input_nodes(fb), &fgnn_batch_device_meta(fb)->num_input, cap, stream_);
  }
  PresampleRank(d_freq, ds_.num_node, ds_.ranking_nodes, stream_);
  (void)hipFree(d_freq);
}

void Engine::PreSampleStatic() {
  // cuda/pre_sampler.cc:57-109 with DoGPUSampleAllNeighbour (cuda_loops.cc:500-571) as the "sampler": the access
  // frequency of a node = the number of batches whose closed L-hop neighbourhood holds it -- no random draw anywhere.
  // The neighbourhoods are grown level by level against a stamp array over the node ids (fgnn_neighbourhood_expand);
  // batch b stamps with b, so nothing is reset between batches and no size is read back by the host.
  const int epochs = RC().presample_epoch > 0 ? RC().presample_epoch : 1;
  const size_t layers = RC().fanout.size();  // GetFanout().size(): num_layer copies of num_neighbor for random walks
  Shuffler sh(static_cast<const uint32_t *>(ds_.train_set.ptr), ds_.num_train, epochs, RC().batch_size, 0, 1, stream_);
  const size_t n = ds_.num_node;
  uint32_t *d_freq = nullptr, *d_stamp = nullptr, *d_front[2] = {nullptr, nullptr}, *d_count = nullptr;
  SAM_HIP(hipMalloc(&d_freq, n * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_stamp, n * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_front[0], n * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_front[1], n * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_count, (layers + 1) * sizeof(uint32_t)));
  SAM_HIP(hipMemsetAsync(d_freq, 0, n * sizeof(uint32_t), stream_));
  SAM_HIP(hipMemsetAsync(d_stamp, 0, n * sizeof(uint32_t), stream_));
  const uint32_t *d_batch;
  size_t bsize;
  uint32_t mark = 0;
  while (sh.GetBatch(&d_batch, &bsize)) {
    ++mark;  // < 2^32 batches: stamps never repeat
    SAM_HIP(hipMemsetAsync(d_count, 0, (layers + 1) * sizeof(uint32_t), stream_));
    for (size_t l = 0; l < layers; ++l) {
      const uint32_t *front = l == 0 ? d_batch : d_front[(l - 1) & 1];
      SAM_FGNN(fgnn_neighbourhood_expand(d_indptr_, d_indices_, front, l == 0 ? bsize : 0, l == 0 ? nullptr : d_count + l,
                                         l == 0 ? bsize : n, d_stamp, mark, d_freq, d_front[l & 1], n, d_count + l + 1,
                                         l == 0 ? 1 : 0, stream_));
    }
  }
  PresampleRank(d_freq, n, ds_.ranking_nodes, stream_);
  (void)hipFree(d_freq);
  (void)hipFree(d_stamp);
  (void)hipFree(d_front[0]);
  (void)hipFree(d_front[1]);
  (void)hipFree(d_count);
}

void Engine::BuildCacheTable() {
  // direct-map table: table[ranking_nodes[i]] = i for i < num_cached (dist_engine.cc:193-229)
  num_cached_ = (size_t)(ds_.num_node * RC().cache_percentage);
  std::vector<uint32_t> table(ds_.num_node, FGNN_EMPTY_KEY);
  for (size_t i = 0; i < num_cached_; ++i) table[ds_.ranking_nodes[i]] = (uint32_t)i;
  SAM_HIP(hipMalloc(&d_cache_table_, ds_.num_node * sizeof(uint32_t)));
  SAM_HIP(hipMemcpy(d_cache_table_, table.data(), ds_.num_node * sizeof(uint32_t), hipMemcpyHostToDevice));
}

void Engine::PublishPending() {
  FlushOwedTail();
  std::unique_lock<std::mutex> lk(pub_mu_);
  pub_cv_.wait(lk, [&] { return pub_q_.empty(); });
}

void Engine::PublisherLoop() {
  SAM_HIP(hipSetDevice(device_));
  for (;;) {
    int cur = -1;
    bool overdue = false;
    {
      std::unique_lock<std::mutex> lk(pub_mu_);
      auto ready = [&] { return pub_stop_ || !pub_q_.empty(); };
      const uint64_t since = tail_since_us_.load(std::memory_order_acquire);
      if (since == 0) {
        // (woken by notify_all when a tail becomes owed: the predicate then also holds for "re-evaluate the wait")
        pub_cv_.wait(lk, [&] { return ready() || tail_since_us_.load(std::memory_order_acquire) != 0; });
      } else {
        // a batch's tail is owed to the next sample_once: if none comes (the end of a script's loop), finish it here
        const uint64_t now = Timer::NowMicro();
        const uint64_t left = now >= since + kOwedTailUs ? 0 : since + kOwedTailUs - now;
        if (!pub_cv_.wait_for(lk, std::chrono::microseconds(left + 1), ready)) overdue = true;
      }
      if (!pub_q_.empty()) {
        cur = pub_q_.front();
        overdue = false;
      } else if (pub_stop_) {
        return;
      }
    }
    if (cur < 0) {
      if (overdue && tail_since_us_.load(std::memory_order_acquire) != 0 &&
          Timer::NowMicro() >= tail_since_us_.load(std::memory_order_acquire) + kOwedTailUs && enq_mu_.try_lock()) {
        // (try_lock: a sample_once that is running right now finishes the tail itself)
        if (tail_owed_ >= 0 && Timer::NowMicro() >= tail_since_us_.load(std::memory_order_acquire) + kOwedTailUs)
          FinishBatch(tail_owed_);
        enq_mu_.unlock();
      }
      continue;
    }
    PublishSlot(cur);
    {
      std::lock_guard<std::mutex> lk(pub_mu_);
      pub_q_.pop_front();
      slots_[cur].pending = false;
    }
    pub_cv_.notify_all();
  }
}

void Engine::PublishSlot(int slot) {
  Slot &s = slots_[slot];
  fgnn_batch_meta m;
  Timer t_wait;
  SAM_FGNN(fgnn_batch_wait(s.fb, &m));  // the pack kernel is ordered before the summary copy
  sstat_.pub_wait += t_wait.Passed();
  Timer t_pub;
  SAM_CHECK_EQ(m.overflow, 0u) << "batch exceeded its buffers";
  mq_->SimpleSend(s.mq_key);
  // sample / cache-index times from the kernels' own time stamps in the summary (10 ns ticks): three event records and
  // their bubbles per batch less on the sampler's streams
  float ms_sample = 0, ms_index = 0, ms_send = 0;
  const uint64_t t_end_sample = m.t_sampled ? m.t_sampled : m.t_closed;
  if (m.t_start && t_end_sample > m.t_start) ms_sample = (float)((t_end_sample - m.t_start) * 1e-5);
  if (m.t_sampled && m.t_closed > m.t_sampled) ms_index = (float)((m.t_closed - m.t_sampled) * 1e-5);
  auto &P = Profiler::Get();
  size_t edges = 0;
  for (uint32_t l = 0; l < m.num_layers; ++l) edges += m.num_edge[l];
  const uint64_t key = s.key;
  const double total = s.started.Passed();
  ms_send = (float)(total * 1e3) - ms_sample - ms_index;
  P.LogStep(key, kLogL1NumSample, (double)edges);
  P.LogStep(key, kLogL1NumNode, (double)m.num_input);
  P.LogStep(key, kLogL1SampleTime, ms_sample * 1e-3);
  P.LogStep(key, kLogL2CoreSampleTime, ms_sample * 1e-3);
  P.LogStep(key, kLogL3CacheGetIndexTime, ms_index * 1e-3);
  P.LogStep(key, kLogL1SendTime, ms_send * 1e-3);
  P.LogEpochAdd(key, kLogEpochSampleTime, ms_sample * 1e-3);
  P.LogEpochAdd(key, kLogEpochSampleGetCacheMissIndexTime, ms_index * 1e-3);
  P.LogEpochAdd(key, kLogEpochSampleSendTime, ms_send * 1e-3);
  P.LogEpochAdd(key, kLogEpochSampleTotalTime, total);
  if (s.started_us) {  // the engine's own events of the Chrome trace (profiler.h:142-165): sample = enqueue .. published
    const uint64_t now = Timer::NowMicro();
    P.TraceStep(key, kL1Event_Sample, s.started_us, true);
    P.TraceStep(key, kL1Event_Sample, now, false);
    P.TraceStep(key, kL2Event_Sample_Core, s.started_us, true);
    P.TraceStep(key, kL2Event_Sample_Core, s.started_us + (uint64_t)(ms_sample * 1e3), false);
  }
  sstat_.pub_rest += t_pub.Passed();
}

// One call = one batch ENQUEUED, in two halves (fgnn_sampler_sample_begin / _end): this call enqueues the sampling CHAIN
// of batch k -- for khop2 the kernels that rewrite CSR rows and therefore run in batch order on the GPU -- and only then
// the TAIL of batch k - 1 (last dedup fill, cache-index split, message pack, hand-over to the publisher thread).  With
// whole batches enqueued one after the other, the first sampler launch of batch k sat behind ~7 launches and the queue
// calls of batch k - 1's tail in this thread's order: 30 of a sampler GPU's 100 us per batch were the chain waiting for
// the host (profiles/r06_a_pipeline_stages.txt).  The batch whose tail is owed is finished by the next call, by
// whoever asks for published results (PublishPending), or -- when no further call comes: the end of a script's loop --
// by the publisher thread after kOwedTailUs.
void Engine::SampleOnceArch5() {
  SAM_HIP(hipSetDevice(device_));
  const uint32_t *d_batch = nullptr;
  size_t bsize = 0;
  if (!shuffler_->GetBatch(&d_batch, &bsize)) SAM_FATAL << "null task from DoShuffle!";
  const uint64_t key = BatchKey(shuffler_->Epoch(), shuffler_->Step());
  const int cur = (int)(next_slot_++ % slots_.size());
  Slot &s = slots_[cur];
  Timer t_wait;
  {  // the slot's previous batch (slots_.size() batches ago) must have been published
    std::unique_lock<std::mutex> lk(pub_mu_);
    pub_cv_.wait(lk, [&] { return !s.pending; });
    if (!publish_thread_.joinable()) publish_thread_ = std::thread([this] { PublisherLoop(); });
  }
  sstat_.slot_wait += t_wait.Passed();
  std::lock_guard<std::mutex> enq(enq_mu_);  // (the publisher thread finishes an overdue tail under the same lock)
  Timer t_enq;
  s.started = Timer();
  s.started_us = RC().option_dump_trace ? Timer::NowMicro() : 0;
  s.key = key;
  SAM_FGNN(fgnn_sampler_sample_begin(sampler_, d_batch, bsize, key, s.fb, s.st, &s.seq));
  if (tail_owed_ >= 0) FinishBatch(tail_owed_);
  // Only a sampler PROCESS owes tails.  Where sampler and extractor share a process (arch2-4, arch6) the thread that
  // asks for published results (samgraph_get_log_*, PublishPending) may be the one that consumes the batches: finishing
  // an owed tail there can wait for a free slot of the in-process ring, which only that thread's consumption frees
  if (RC().run_arch == kArch5 && dist_type_ == DistType::Sample) {
    tail_owed_ = cur;
    {  // (under the publisher's mutex: it either sees the value when it checks its wait condition or is already waiting
       // when the notification comes -- a lost wake-up would leave the last batch of a script's loop unpublished)
      std::lock_guard<std::mutex> lk(pub_mu_);
      tail_since_us_.store(Timer::NowMicro(), std::memory_order_release);
    }
    pub_cv_.notify_all();  // the publisher thread now waits with a time limit
  } else {
    FinishBatch(cur);
  }
  sstat_.enqueue += t_enq.Passed();
  ++sstat_.n;
  // (the reference drains its pipeline at the end of an epoch because the next call reshuffles the seed array in place,
  // dist_loops_arch5.cc:131-137; the shuffler here keeps the previous epoch's device array alive instead)
}

// the batch in slot `cur` has its chain enqueued: tail, cache-index split, message pack, hand-over.  enq_mu_ is held
void Engine::FinishBatch(int cur) {
  Slot &s = slots_[cur];
  tail_owed_ = -1;
  tail_since_us_.store(0, std::memory_order_release);
  const bool use_cache = RC().UseGPUCache();
  SAM_FGNN(fgnn_sampler_sample_end(sampler_, s.seq, s.fb, use_cache ? d_cache_table_ : nullptr, s.st));
  // serialise straight into a queue slot (MessageTaskQueue::Send, task_queue.cc:378-386)
  void *slot = mq_->GetPtr(&s.mq_key);
  if (!slot) return;  // this process is shutting down (MemoryQueue::Close): the batch is dropped, Shutdown syncs its stream
  PackArgs a;
  memset(&a, 0, sizeof(a));
  a.d_meta = fgnn_batch_device_meta(s.fb);
  a.input_nodes = fgnn_batch_input_nodes(s.fb);
  a.output_nodes = fgnn_batch_output_nodes(s.fb);
  for (int k = 0; k < 4; ++k) a.cidx[k] = fgnn_batch_cache_index_ptr(s.fb, k);
  for (size_t l = 0; l < RC().fanout.size(); ++l) {
    a.row[l] = fgnn_batch_row(s.fb, (int)l);
    a.col[l] = fgnn_batch_col(s.fb, (int)l);
    a.data[l] = fgnn_batch_data(s.fb, (int)l);
  }
  a.ship_input = (!use_cache || RC().have_switcher) ? 1 : 0;
  a.ship_cache_index = use_cache ? 1 : 0;
  a.have_data = RC().sample_type == kRandomWalk ? 1 : 0;
  a.slot = DeviceVisible(slot);
  // null: no device ring, or none of its slots is free (counted per sampler either way: samgraph_ext_queue_stats)
  a.payload = mq_->ClaimDeviceSlot(ring_id_ >= 0 ? ring_id_ : (RC().run_arch == kArch5 ? worker_id_ : 0), s.mq_key);
  a.slot_bytes = mq_->SlotBytes();
  a.h_meta = reinterpret_cast<uint32_t *>(fgnn_batch_host_meta(s.fb));
  const bool check = handoff_check_left_ > 0;
  if (check) {
    --handoff_check_left_;
    if (!s.d_msg_words) SAM_HIP(hipMalloc(&s.d_msg_words, sizeof(uint32_t)));
    a.msg_words = s.d_msg_words;
  }
  mq_->MarkChecked(s.mq_key, worker_id_, check);
  SAM_FGNN(LaunchPack(a, s.st));
  if (check) {  // checksum of the packed words, appended behind them in the slot the payload went to
    uint32_t *where = static_cast<uint32_t *>(a.payload ? a.payload : a.slot);
    SAM_FGNN(LaunchMessageChecksum(where, s.d_msg_words, 0, 0, nullptr, s.st));
  }
  SAM_FGNN(fgnn_batch_meta_copied(s.fb));  // by the pack kernel
  SAM_FGNN(fgnn_batch_finish(s.fb, s.st));
  {  // hand the batch to the publisher thread: it waits for the GPU work and publishes in this order
    std::lock_guard<std::mutex> lk(pub_mu_);
    s.pending = true;
    pub_q_.push_back(cur);
  }
  pub_cv_.notify_all();
}

// a batch whose chain is enqueued and whose tail is not: finish it now (callers that need everything published)
void Engine::FlushOwedTail() {
  // nothing owed (always the case where sampler and extractor share a process): do not even wait for the enqueueing
  // thread -- it may be waiting for a free ring slot that only the CALLER's consumption frees
  if (tail_since_us_.load(std::memory_order_acquire) == 0) return;
  std::lock_guard<std::mutex> enq(enq_mu_);
  if (tail_owed_ >= 0) {
    SAM_HIP(hipSetDevice(device_));
    FinishBatch(tail_owed_);
  }
}

// ------------------------------------------------------------------------------------------------
// arch5 trainer process (dist_engine.cc:366-482, dist_loops_arch5.cc:158-285, dist_loops.cc:713-929)

void Engine::TrainInit(int worker_id, Context ctx, DistType type) {
  (void)worker_id;
  // arch6 ("SGNN", dist_engine.cc:231-482 with kArch6): a worker calls sample_init AND train_init in the same process
  // on the same GPU; the second call adds the extractor half
  const bool second_half = RC().run_arch == kArch6 && dist_type_ == DistType::Sample && !tstream_;
  if (initialized_ && !second_half) return;
  SAM_CHECK(data_initialized_) << "samgraph_data_init must run before fork";
  SAM_CHECK(ctx.IsGPU()) << "trainer context must be cuda:N";
  if (RC().run_arch == kArch6) {
    SAM_CHECK(second_half) << "arch6: samgraph_sample_init must precede samgraph_train_init in each worker";
    SAM_CHECK_EQ(ctx.device_id, device_) << "arch6 samples and trains on the same GPU";
    SAM_CHECK(type == DistType::Extract);
  }
  Timer t;
  if (!second_half) dist_type_ = type;
  SAM_HIP(hipSetDevice(ctx.device_id));
  tdevice_ = ctx.device_id;
  SAM_HIP(hipStreamCreateWithFlags(&tstream_, hipStreamNonBlocking));
  for (auto &x : xctx_) {
    SAM_HIP(hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking));
    for (auto &e : x.ev) SAM_HIP(hipEventCreate(&e));
  }
  if (!second_half) mq_->PinMemory();
  // the host feature table becomes GPU-readable: miss rows are fetched by the gather kernel itself
  // (replaces the OpenMP ExtractMissData + H2D copy, cuda_cache_manager_host.cc:38-56)
  SAM_HIP(hipHostRegister(ds_.feat.ptr, ds_.feat.bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  dev_host_feat_ = DeviceVisible(ds_.feat.ptr);
  SAM_HIP(hipMalloc(&d_label_, ds_.label.bytes));
  SAM_HIP(hipMemcpy(d_label_, ds_.label.ptr, ds_.label.bytes, hipMemcpyHostToDevice));
  if (RC().UseGPUCache()) {
    Timer tc;
    BuildTrainerCache();
    Profiler::Get().LogInit(kLogInitL2BuildCache, tc.Passed());
  }
  pool_.reset(new GraphPool(RC().max_copying_jobs));
  Profiler::Get().LogInit(kLogInitL1Trainer, t.Passed());
  initialized_ = true;
}

void Engine::BuildTrainerCache() {
  // DistCacheManager (dist_cache_manager_host.cc:60-119): top cache_percentage*N rows of the ranking
  num_cached_ = (size_t)(ds_.num_node * RC().cache_percentage);
  const size_t row_bytes = ds_.feat_dim * 4;
  SAM_HIP(hipMalloc(&d_cache_rows_, num_cached_ ? num_cached_ * row_bytes : 16));
  if (num_cached_) {
    uint32_t *d_rank = nullptr;
    SAM_HIP(hipMalloc(&d_rank, num_cached_ * sizeof(uint32_t)));
    SAM_HIP(hipMemcpy(d_rank, ds_.ranking_nodes, num_cached_ * sizeof(uint32_t), hipMemcpyHostToDevice));
    SAM_FGNN(fgnn_gather_rows_masked(d_cache_rows_, dev_host_feat_, d_rank, nullptr, num_cached_, nullptr, num_cached_,
                                     ds_.feat_dim, FGNN_F32, FeatRowMask(), tstream_));
    SAM_HIP(hipStreamSynchronize(tstream_));
    (void)hipFree(d_rank);
  }
  if (dist_type_ == DistType::Switch) BuildCacheTable();  // the switcher splits hits/misses itself
}

// One batch of the trainer-side loop in two halves, so that StartExtract's thread can keep several batches in flight
// (each on its own stream): TrainerIssue receives a message and enqueues its copies and gathers, TrainerComplete waits
// for them, hands the queue slot back and submits the batch.  The reference does both in one step with a sync in
// between every copy (dist_loops.cc:713-929); with one batch at a time the host link idles while the next message is
// parsed and its ~10 copies are enqueued, and the GPU idles while the host waits.
void Engine::TrainerOnce() {
  SAM_HIP(hipSetDevice(tdevice_));
  while (pool_->Full() && !shutdown_) std::this_thread::sleep_for(std::chrono::microseconds(1));
  if (shutdown_) return;
  TrainerIssue(xctx_[0], nullptr, 0);
  if (xctx_[0].b) TrainerComplete(xctx_[0]);
}

// The extractor's loop (StartExtract's thread of an arch5 trainer, samgraph_start's copy thread of arch2 / arch3):
// up to kExtractDepth batches in flight.  A further message is taken only when one is PUBLISHED (TryRecv never waits):
// a trainer that blocked for a message while holding unreleased queue slots could wait for a sampler that is itself
// waiting for one of those slots (few slots: large fan-outs under SAMGRAPH_MQ_BYTES, many trainers).  With nothing in
// flight the thread holds no slot and may block like the reference's loop does.
void Engine::ExtractLoop(size_t count) {
  SAM_HIP(hipSetDevice(tdevice_));
  size_t issued = 0;
  int head = 0, inflight = 0;
  while ((issued < count || inflight) && !shutdown_) {
    if (issued < count && inflight < kExtractDepth) {
      const void *msg = nullptr;
      size_t key = 0;
      if (inflight == 0 || mq_->TryRecv(&msg, &key)) {
        ExtractCtx &x = xctx_[(head + inflight) % kExtractDepth];
        TrainerIssue(x, msg, key);
        if (!x.b) break;  // the queue was closed under a blocking receive: this process is shutting down
        ++inflight;
        ++issued;
        continue;
      }
    }
    TrainerComplete(xctx_[head]);
    head = (head + 1) % kExtractDepth;
    --inflight;
  }
  // leaving with batches still in flight (shutdown, or the queue closed under a blocking receive): their queue slots go
  // back -- a sampler of another process that wraps around to an unreleased slot would wait for it forever
  for (; inflight; --inflight, head = (head + 1) % kExtractDepth) {
    ExtractCtx &x = xctx_[head];
    if (!x.b) continue;
    (void)hipStreamSynchronize(x.st);  // the launches read the slot in place
    mq_->Release(x.mq_key);
    ReleaseBatch(x.b.get());
    x.b.reset();
  }
}

// `msg` null: block for the next message (the caller holds no queue slot); else the message TryRecv handed out
void Engine::TrainerIssue(ExtractCtx &x, const void *taken, size_t taken_key) {
  hipStream_t tstream_ = x.st;  // everything of this batch goes to the context's stream
  hipEvent_t *te_ = x.ev;
  const uint64_t recv_us = RC().option_dump_trace ? Timer::NowMicro() : 0;
  Timer t_recv;
  size_t mq_key = taken_key;
  const char *msg = static_cast<const char *>(taken ? taken : mq_->Recv(&mq_key));
  if (!msg) {  // MemoryQueue::Close
    x.b.reset();
    return;
  }
  const double recv_time = t_recv.Passed();
  Timer t_copy;
  TransData hdr;
  memcpy(&hdr, msg, sizeof(hdr));
  SAM_CHECK(hdr.num_layer >= 0)
      << "Size of batch topology exceeds memory queue capability. Raise SAMGRAPH_MQ_BYTES";  // task_queue.cc:162
  SAM_CHECK_LE((size_t)hdr.num_layer, (size_t)FGNN_MAX_LAYERS);
  const bool use_cache = RC().UseGPUCache();
  const bool ship_input = !use_cache || RC().have_switcher;
  // headers are parsed from the host slot (`p`); the arrays are copied from the payload slot (`q`), which is the host
  // slot itself or a slot of the sampler's HBM ring at the same offsets
  bool payload_on_device = false;
  const char *payload = static_cast<const char *>(mq_->Payload(mq_key, msg, &payload_on_device));
  const uint32_t *p = reinterpret_cast<const uint32_t *>(msg + sizeof(TransData));
  const ptrdiff_t to_payload = payload - msg;
  auto b = std::make_shared<GraphBatch>();
  x.b = b;
  x.mq_key = mq_key;
  b->key = hdr.key;
  b->num_layer = hdr.num_layer;
  b->num_input = hdr.input_size;
  b->num_output = hdr.output_size;
  b->device = tdevice_;

  // Arrays that outlive the queue slot (ids, COO) go into ONE pooled buffer, copied by ONE kernel; the cache index
  // arrays are used in place by the gathers below (the slot is released only after they have run).  `in_place`
  // translates a position in the host slot into the device-visible address of the same position in the payload slot.
  auto in_place = [&](const uint32_t *src) -> const uint32_t * {
    const char *q = reinterpret_cast<const char *>(src) + to_payload;
    return reinterpret_cast<const uint32_t *>(payload_on_device ? q : static_cast<const char *>(mq_->DeviceVisiblePtr(q)));
  };
  constexpr size_t kAlignWords = 64;  // 256-byte aligned sub-arrays
  auto padded = [&](size_t n) { return (n + kAlignWords - 1) / kAlignWords * kAlignWords; };
  size_t keep_words = padded(hdr.output_size) + (ship_input ? padded(hdr.input_size) : 0);
  {
    const uint32_t *q = p + (ship_input ? hdr.input_size : 0) + hdr.output_size;
    if (use_cache) q += 2 * hdr.input_size;  // miss + cache index pairs
    for (int l = 0; l < hdr.num_layer; ++l) {
      uint64_t g[3];
      memcpy(g, q, sizeof(g));
      keep_words += (hdr.have_data ? 3 : 2) * padded(g[2]);
      q += sizeof(GraphData) / sizeof(uint32_t) + g[2] * (hdr.have_data ? 3 : 2);
    }
  }
  uint32_t *keep = static_cast<uint32_t *>(dev_pool_.Alloc(keep_words * sizeof(uint32_t)));
  b->pooled.push_back(keep);
  UnpackArgs ua;
  ua.num_segments = 0;
  auto to_device = [&](const uint32_t *src, size_t n) -> uint32_t * {
    uint32_t *d = keep;
    keep += padded(n);
    if (n) {
      auto &sg = ua.seg[ua.num_segments++];
      sg.dst = d;
      sg.src = in_place(src);
      sg.words = n;
    }
    return d;
  };

  uint32_t *d_input = nullptr;
  if (ship_input) {
    d_input = to_device(p, hdr.input_size);
    p += hdr.input_size;
    b->input_nodes = d_input;
    b->input_device = tdevice_;
  }
  const uint32_t *slot_output = p;
  uint32_t *d_output = to_device(p, hdr.output_size);
  p += hdr.output_size;
  b->output_nodes = d_output;
  b->output_device = tdevice_;

  const size_t num_miss = hdr.num_miss, num_cache = hdr.input_size - hdr.num_miss;
  const uint32_t *d_cidx[4] = {nullptr, nullptr, nullptr, nullptr};
  if (use_cache) {
    if (num_miss) {
      d_cidx[0] = in_place(p); p += num_miss;
      d_cidx[1] = in_place(p); p += num_miss;
    }
    if (num_cache) {
      d_cidx[2] = in_place(p); p += num_cache;
      d_cidx[3] = in_place(p); p += num_cache;
    }
  }
  size_t graph_bytes = 0;
  for (int l = 0; l < hdr.num_layer; ++l) {
    uint64_t g[3];
    memcpy(g, p, sizeof(g));  // GraphData header, possibly 4-byte aligned
    p += sizeof(GraphData) / sizeof(uint32_t);
    TrainGraphView &v = b->graphs[l];
    v.num_src = g[0];
    v.num_dst = g[1];
    v.num_edge = g[2];
    v.row = to_device(p, v.num_edge); p += v.num_edge;
    v.col = to_device(p, v.num_edge); p += v.num_edge;
    if (hdr.have_data) { v.data = to_device(p, v.num_edge); p += v.num_edge; }
    graph_bytes += v.num_edge * (hdr.have_data ? 12 : 8);
  }
  SAM_CHECK_LE((size_t)((const char *)p - msg), mq_->SlotBytes());
  if (mq_->IsChecked(mq_key)) {
    // the sender's checksum, recomputed through the very address the arrays are read from (the sampler's HBM slot
    // mapped over xGMI, or the pinned host slot) BEFORE anything trusts the payload: a mapping that reads garbage must
    // neither pass as a slow but valid run nor feed garbage indices to the gathers below (warm-up messages only: the
    // synchronisation costs nothing that is measured)
    if (!x.d_check) SAM_HIP(hipMalloc(&x.d_check, sizeof(uint32_t)));
    SAM_HIP(hipMemsetAsync(x.d_check, 0, sizeof(uint32_t), tstream_));
    const char *base = payload_on_device ? payload : static_cast<const char *>(mq_->DeviceVisiblePtr(msg));
    SAM_FGNN(LaunchMessageChecksum(reinterpret_cast<uint32_t *>(const_cast<char *>(base)), nullptr,
                                   (size_t)((const char *)p - msg) / sizeof(uint32_t), 1, x.d_check, tstream_));
    uint32_t bad = 0;
    SAM_HIP(hipMemcpyAsync(&bad, x.d_check, sizeof(bad), hipMemcpyDeviceToHost, tstream_));
    SAM_HIP(hipStreamSynchronize(tstream_));
    mq_->CountCheck(mq_key, bad == 0);
    SAM_CHECK(bad == 0) << "hand-off check: message " << mq_key << " (batch key " << hdr.key
                        << ") does not verify on the receiving GPU -- the payload read through the "
                        << "mapped slot differs from what the sampler packed";
  }
  // features (DoCacheFeatureCopy / DoSwitchCacheFeatureCopy / DoCPUFeatureExtract+DoFeatureCopy)
  const size_t row_bytes = ds_.feat_dim * 4;
  void *d_feat = dev_pool_.Alloc(hdr.input_size * row_bytes);
  b->pooled.push_back(d_feat);
  b->feat = d_feat;
  b->feat_rows = hdr.input_size;
  void *d_lab = dev_pool_.Alloc(hdr.output_size * 8);
  b->pooled.push_back(d_lab);
  b->label = d_lab;
  size_t miss_rows = num_miss;
  bool timed_gathers = false, unpacked = false, labelled = false;
  x.stamp_grid = 0;
  if (use_cache && dist_type_ != DistType::Switch) {
    // the whole batch in ONE launch: a band of workgroups pulls the miss rows over the host link
    // (CombineMissData with ExtractMissData's fetch fused in) while the rest of the grid streams the hit rows out of
    // the HBM cache (CombineCacheData), gathers the labels (DoCPULabelExtractAndCopy, dist_loops.cc:886-929) and copies
    // the arrays that outlive the queue slot out of it (ParseData); the reference: ~10 copies, an OpenMP gather and two
    // kernels with a sync after each (dist_loops.cc:713-929)
    fgnn_copy_segment segs[UnpackArgs::kMaxSegments];
    for (int k = 0; k < ua.num_segments; ++k) segs[k] = fgnn_copy_segment{ua.seg[k].dst, ua.seg[k].src, ua.seg[k].words};
    fgnn_extract_job j;
    memset(&j, 0, sizeof(j));
    j.out = d_feat;
    j.miss_rows = dev_host_feat_;
    j.cache_rows = d_cache_rows_;
    j.miss_src = d_cidx[0]; j.miss_dst = d_cidx[1]; j.cache_src = d_cidx[2]; j.cache_dst = d_cidx[3];
    j.num_miss = num_miss;
    j.num_cache = num_cache;
    j.dim = ds_.feat_dim;
    j.dtype = FGNN_F32;
    j.miss_row_mask = FeatRowMask();
    j.label_out = d_lab;
    j.label_src = d_label_;
    j.label_index = in_place(slot_output);  // read where the message has them: the copy above is this launch's own
    j.num_label = hdr.output_size;
    j.label_dtype = FGNN_I64;
    j.segs = segs;
    j.num_segs = ua.num_segments;
    j.link_workgroups = ExtractLinkWgs();
    const size_t grid = fgnn_extract_fused_grid(&j);
    if (grid) {
      // per-band durations (kLogL3CacheCombine{Miss,Cache}Time): every workgroup posts its start / end clock into
      // pinned host memory -- no event records around the launch, no copy behind it
      if (grid > x.stamp_cap) {
        if (x.h_stamps) (void)hipHostFree(x.h_stamps);
        x.stamp_cap = grid + 256;
        SAM_HIP(hipHostMalloc(reinterpret_cast<void **>(&x.h_stamps), 2 * x.stamp_cap * sizeof(unsigned long long),
                              hipHostMallocDefault));
      }
      j.stamps = x.h_stamps;
      SAM_FGNN(fgnn_extract_fused(&j, tstream_));
      x.stamp_grid = grid;
      x.stamp_link = fgnn_extract_fused_link_grid(&j);  // (fewer than asked for when the miss rows are few)
      timed_gathers = unpacked = labelled = true;
    }
  }
  if (!unpacked) SAM_FGNN(LaunchUnpack(ua, tstream_));
  if (unpacked) {
    // everything went with the one launch above
  } else if (!use_cache) {
    SAM_FGNN(fgnn_gather_rows_shared(d_feat, dev_host_feat_, d_input, nullptr, hdr.input_size, nullptr, hdr.input_size,
                                     ds_.feat_dim, FGNN_F32, FeatRowMask(), ExtractorSharesGpu(), tstream_));
    miss_rows = hdr.input_size;
  } else if (dist_type_ == DistType::Switch) {
    // own (smaller) cache: split on this GPU with device-side counts
    const size_t n = hdr.input_size;
    uint32_t *idx[4];
    for (auto &q : idx) { q = static_cast<uint32_t *>(dev_pool_.Alloc(n * sizeof(uint32_t))); b->pooled.push_back(q); }
    const size_t ws_bytes = fgnn_scratch_bytes(n);
    void *ws = dev_pool_.Alloc(ws_bytes + 16);
    b->pooled.push_back(ws);
    uint32_t *d_counts = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + ws_bytes);
    SAM_FGNN(fgnn_get_miss_cache_index(d_cache_table_, d_input, n, nullptr, n, idx[0], idx[1], idx[2], idx[3], d_counts,
                                       ws, ws_bytes, tstream_));
    fgnn_extract_job j;
    memset(&j, 0, sizeof(j));
    j.out = d_feat;
    j.miss_rows = dev_host_feat_;
    j.cache_rows = d_cache_rows_;
    j.miss_src = idx[0]; j.miss_dst = idx[1]; j.cache_src = idx[2]; j.cache_dst = idx[3];
    j.d_counts = d_counts;
    j.cap = n;
    j.dim = ds_.feat_dim;
    j.dtype = FGNN_F32;
    j.miss_row_mask = FeatRowMask();
    j.link_workgroups = FGNN_LINK_WGS_SHARED;  // a switcher shares its GPU with a sampler
    if (fgnn_extract_fused_grid(&j)) {
      SAM_FGNN(fgnn_extract_fused(&j, tstream_));
    } else {
      SAM_FGNN(fgnn_gather_rows_shared(d_feat, dev_host_feat_, idx[0], idx[1], 0, d_counts, n, ds_.feat_dim, FGNN_F32,
                                       FeatRowMask(), ExtractorSharesGpu(), tstream_));
      SAM_FGNN(fgnn_gather_rows(d_feat, d_cache_rows_, idx[2], idx[3], 0, d_counts + 1, n, ds_.feat_dim, FGNN_F32,
                                tstream_));
    }
  } else {
    // rows that are not whole 16-byte chunks: one launch per list, bracketed by events
    SAM_HIP(hipEventRecord(te_[0], tstream_));
    if (num_miss)   // CombineMissData with the host fetch fused in
      SAM_FGNN(fgnn_gather_rows_shared(d_feat, dev_host_feat_, d_cidx[0], d_cidx[1], num_miss, nullptr, num_miss,
                                       ds_.feat_dim, FGNN_F32, FeatRowMask(), ExtractorSharesGpu(), tstream_));
    SAM_HIP(hipEventRecord(te_[1], tstream_));
    if (num_cache)  // CombineCacheData
      SAM_FGNN(fgnn_gather_rows(d_feat, d_cache_rows_, d_cidx[2], d_cidx[3], num_cache, nullptr, num_cache,
                                ds_.feat_dim, FGNN_F32, tstream_));
    SAM_HIP(hipEventRecord(te_[2], tstream_));
    timed_gathers = true;
  }
  // labels (DoCPULabelExtractAndCopy, dist_loops.cc:886-929) -- gathered on the GPU from the HBM copy
  if (!labelled)
    SAM_FGNN(fgnn_gather_rows(d_lab, d_label_, d_output, nullptr, hdr.output_size, nullptr, hdr.output_size, 1, FGNN_I64,
                              tstream_));
  xstat_.recv += recv_time;
  xstat_.issue += t_copy.Passed();
  ++xstat_.n;
  x.recv_time = recv_time;
  x.recv_us = recv_us;
  x.t_copy = t_copy;
  x.timed_gathers = timed_gathers;
  x.miss_rows = miss_rows;
  x.graph_bytes = graph_bytes;
  x.input_size = hdr.input_size;
  x.output_size = hdr.output_size;
}

void Engine::TrainerComplete(ExtractCtx &x) {
  hipStream_t tstream_ = x.st;
  hipEvent_t *te_ = x.ev;
  std::shared_ptr<GraphBatch> b = std::move(x.b);
  const size_t mq_key = x.mq_key, miss_rows = x.miss_rows, graph_bytes = x.graph_bytes;
  const double recv_time = x.recv_time;
  const Timer &t_copy = x.t_copy;
  const bool timed_gathers = x.timed_gathers;
  const size_t row_bytes = ds_.feat_dim * 4;
  struct { size_t input_size, output_size; } hdr{x.input_size, x.output_size};
  Timer t_wait;
  while (pool_->Full() && !shutdown_) std::this_thread::sleep_for(std::chrono::microseconds(1));
  xstat_.pool_wait += t_wait.Passed();
  Timer t_sync;
  SAM_HIP(hipStreamSynchronize(tstream_));
  xstat_.sync += t_sync.Passed();
  mq_->Release(mq_key);
  pool_->Submit(b);

  const double copy_time = t_copy.Passed();
  auto &P = Profiler::Get();
  P.LogStep(b->key, kLogL1RecvTime, recv_time);
  P.LogStep(b->key, kLogL1CopyTime, recv_time + copy_time);
  P.LogStep(b->key, kLogL2CacheCopyTime, copy_time);
  if (timed_gathers) {  // device time of the two gathers (the reference times them on the host around syncs)
    float ms_miss = 0, ms_cache = 0;
    if (x.stamp_grid) {
      // one launch: a band's time = first start .. last end over its workgroups (100 MHz device wall clock)
      auto span = [&](size_t lo, size_t hi) -> float {
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t k = lo; k < hi; ++k) {
          t0 = std::min(t0, x.h_stamps[2 * k]);
          t1 = std::max(t1, x.h_stamps[2 * k + 1]);
        }
        return hi > lo && t1 >= t0 ? (float)((double)(t1 - t0) * 1e-5) : 0.0f;
      };
      ms_miss = span(0, x.stamp_link);
      ms_cache = span(x.stamp_link, x.stamp_grid);
    } else {
      (void)hipEventElapsedTime(&ms_miss, te_[0], te_[1]);
      (void)hipEventElapsedTime(&ms_cache, te_[1], te_[2]);
    }
    P.LogStep(b->key, kLogL3CacheCombineMissTime, ms_miss * 1e-3);
    P.LogStep(b->key, kLogL3CacheCombineCacheTime, ms_cache * 1e-3);
  }
  P.LogStep(b->key, kLogL1FeatureBytes, (double)hdr.input_size * row_bytes);
  P.LogStep(b->key, kLogL1MissBytes, (double)miss_rows * row_bytes);
  P.LogStep(b->key, kLogL1LabelBytes, (double)hdr.output_size * 8);
  P.LogStep(b->key, kLogL1GraphBytes, (double)graph_bytes);
  P.LogEpochAdd(b->key, kLogEpochCopyTime, recv_time + copy_time);
  P.LogEpochAdd(b->key, kLogEpochFeatureBytes, (double)hdr.input_size * row_bytes);
  P.LogEpochAdd(b->key, kLogEpochMissBytes, (double)miss_rows * row_bytes);
  if (x.recv_us) {  // copy = message received .. batch handed to the graph pool
    const uint64_t now = Timer::NowMicro();
    P.TraceStep(b->key, kL1Event_Copy, x.recv_us, true);
    P.TraceStep(b->key, kL1Event_Copy, now, false);
    P.TraceStep(b->key, RC().UseGPUCache() ? kL2Event_Copy_CacheCopy : kL2Event_Copy_Extract,
                x.recv_us + (uint64_t)(recv_time * 1e6), true);
    P.TraceStep(b->key, RC().UseGPUCache() ? kL2Event_Copy_CacheCopy : kL2Event_Copy_Extract, now, false);
  }
}

void Engine::StartExtract(int count) {
  if (RC().run_arch == kArch6) {
    // one background thread running both halves per batch (SampleCopySubSloop, dist_loops_arch6.cc:196-207,218-227);
    // the reference loops until shutdown, here `count` bounds it when positive
    SAM_CHECK(initialized_ && tstream_);
    if (extract_thread_.joinable()) extract_thread_.join();
    const size_t total = count > 0 ? (size_t)count : RC().num_epoch * shuffler_->NumLocalStep();
    extract_thread_ = std::thread([this, total]() {
      for (size_t i = 0; i < total && !shutdown_; ++i) {
        SampleOnceArch5();
        PublishPending();
        TrainerOnce();
      }
    });
    return;
  }
  SAM_CHECK(initialized_ && (dist_type_ == DistType::Extract || dist_type_ == DistType::Switch));
  if (extract_thread_.joinable()) extract_thread_.join();
  extract_thread_ = std::thread([this, count]() { ExtractLoop(count > 0 ? (size_t)count : 0); });
}

void Engine::RunSampleOnce() {
  SAM_CHECK(initialized_);
  if (RC().run_arch == kArch1 || RC().run_arch == kArch7) {
    SampleOnceArch1();
  } else if (RC().run_arch == kArch2 || RC().run_arch == kArch3 || RC().run_arch == kArch4 ||
             RC().run_arch == kArch6) {
    // both sub-loops once, like RunArch3LoopsOnce (cuda_loops_arch3.cc:198-205) / RunArch6LoopsOnce
    // (dist_loops_arch6.cc:209-216): sample + publish, then copy/extract
    SAM_CHECK(!sample_thread_.joinable()) << "samgraph_sample_once after samgraph_start";
    SAM_CHECK(tstream_) << "arch6: samgraph_train_init has not run in this worker";
    if (dyn_) {  // RunArch4LoopsOnce with UseDynamicGPUCache (cuda_loops_arch4.cc:216-223)
      SampleOnceDynamic();
      return;
    }
    SampleOnceArch5();
    PublishPending();
    TrainerOnce();
  } else if (dist_type_ == DistType::Sample) {
    SampleOnceArch5();
  } else {
    TrainerOnce();  // trainer without --pipeline runs one copy iteration inline (dist_loops_arch5.cc:278-285)
  }
}

uint64_t Engine::GetNextBatch() {
  SAM_CHECK(initialized_ && pool_);
  if (current_) {
    ReleaseBatch(current_.get());
    current_.reset();
  }
  auto b = pool_->Get();
  SAM_CHECK(b) << "graph pool stopped";
  current_ = b;
  return b->key;
}

void Engine::Start() {
  if (!(RC().run_arch == kArch2 || RC().run_arch == kArch3 || RC().run_arch == kArch4)) return;
  SAM_CHECK(initialized_);
  if (sample_thread_.joinable()) return;
  // the sampler thread and the copy/extract thread of cuda_loops_arch3.cc:178-196; each handles every batch of the run
  const size_t total = RC().num_epoch * num_step_;
  if (dyn_) {  // one thread walks sample -> cache copy per batch (see eng_dynamic.cc on the reference's two)
    sample_thread_ = std::thread([this, total]() {
      for (size_t i = 0; i < total && !shutdown_; ++i) SampleOnceDynamic();
    });
    return;
  }
  sample_thread_ = std::thread([this, total]() {
    for (size_t i = 0; i < total && !shutdown_; ++i) SampleOnceArch5();
    if (!shutdown_) PublishPending();
  });
  // the copy thread keeps several batches in flight, like an arch5 trainer's (one at a time leaves the host link idle
  // while the next message is parsed: 0.48 against 0.27 ms per batch on a 1 M-node graph)
  extract_thread_ = std::thread([this, total]() { ExtractLoop(total); });
}

void Engine::Shutdown() {
  shutdown_ = true;
  if (pool_) pool_->Stop();
  // threads of THIS process that are blocked on the queue (samgraph_start's loops stopped before their last batch: the
  // sampler waits for a free slot, the extractor for a message; an arch5 trainer's StartExtract thread waiting for a
  // message that no sampler will send any more) give up.  A receiver that gives up holds no claim (MemoryQueue::Recv)
  // and hands back the slots of the batches it still had in flight (ExtractLoop)
  if (mq_ && (sample_thread_.joinable() || extract_thread_.joinable())) mq_->Close();
  if (sample_thread_.joinable()) sample_thread_.join();
  if (extract_thread_.joinable()) extract_thread_.join();
  if (publish_thread_.joinable()) {  // publishes what is still in flight, then stops
    FlushOwedTail();
    {
      std::lock_guard<std::mutex> lk(pub_mu_);
      pub_stop_ = true;
    }
    pub_cv_.notify_all();
    publish_thread_.join();
  }
  // the shuffler's helper thread synchronises the batch streams before it recycles a seed array (eng_shuffler.cc,
  // Prepare): it must be done before any of those streams is destroyed below -- a run that stops shortly after an epoch
  // boundary would otherwise hand it a dead handle
  if (shuffler_) shuffler_->Quiesce();
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (auto &sl : slots_)
    if (sl.st) (void)hipStreamSynchronize(sl.st);
  if (tstream_) (void)hipStreamSynchronize(tstream_);
  if (sstat_.n)
    SAM_LOG(kInfo) << "sampler: " << sstat_.n << " batches; per batch: waiting for the slot's previous batch to be "
                   << "published " << sstat_.slot_wait / sstat_.n * 1e3 << " ms, enqueueing (host) "
                   << sstat_.enqueue / sstat_.n * 1e3 << " ms; publisher thread: waiting for the GPU "
                   << sstat_.pub_wait / sstat_.n * 1e3 << " ms, publishing + logging " << sstat_.pub_rest / sstat_.n * 1e3
                   << " ms";
  if (xstat_.n)
    SAM_LOG(kInfo) << "extraction thread: " << xstat_.n << " batches; per batch: waiting for a message "
                   << xstat_.recv / xstat_.n * 1e3 << " ms, parsing + enqueueing copies and gathers "
                   << xstat_.issue / xstat_.n * 1e3 << " ms, waiting for room in the graph pool "
                   << xstat_.pool_wait / xstat_.n * 1e3 << " ms, waiting for the GPU " << xstat_.sync / xstat_.n * 1e3
                   << " ms";
  for (auto &x : xctx_) {
    if (x.st) (void)hipStreamSynchronize(x.st);
    for (auto &e : x.ev)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (x.st) { (void)hipStreamDestroy(x.st); x.st = nullptr; }
    if (x.d_check) { (void)hipFree(x.d_check); x.d_check = nullptr; }
    if (x.h_stamps) { (void)hipHostFree(x.h_stamps); x.h_stamps = nullptr; x.stamp_cap = 0; }
    x.b.reset();
  }
  if (mq_ && ring_id_ >= 0) {
    // messages published into this sampler's HBM ring must stay readable until their receivers have copied them
    // (the in-process engines read their own ring and their threads have been joined: nothing to wait for)
    // SAMGRAPH_DEVICE_RING_DRAIN_S: tools that run a sampler with nobody reading (tools/sampler_timeline.py)
    const char *e_drain = getenv("SAMGRAPH_DEVICE_RING_DRAIN_S");
    mq_->DrainDeviceRing(ring_id_, RC().run_arch != kArch5 ? 0.0 : e_drain ? atof(e_drain) : 120.0);
    ring_id_ = -1;
  }
  if (current_) {
    ReleaseBatch(current_.get());
    current_.reset();
  }
  if (dyn_) {  // queued batches share buffers with the cache state: drop them before the pools they return to
    pool_.reset();
    dyn_.reset();
  }
  for (auto &s : slots_) {
    if (s.fb) fgnn_batch_destroy(s.fb);
    if (s.e0) (void)hipEventDestroy(s.e0);
    if (s.e1) (void)hipEventDestroy(s.e1);
    if (s.e2) (void)hipEventDestroy(s.e2);
    if (s.st && s.owns_st) (void)hipStreamDestroy(s.st);
    if (s.d_msg_words) (void)hipFree(s.d_msg_words);
  }
  slots_.clear();
  if (sampler_) {
    fgnn_sampler_destroy(sampler_);
    sampler_ = nullptr;
  }
  shuffler_.reset();
  initialized_ = false;
}

}  // namespace sam
