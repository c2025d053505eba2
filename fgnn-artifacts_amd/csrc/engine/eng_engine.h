// eng_engine.h -- the process-wide engine singleton behind the samgraph_* C ABI.
// Reference: engine.{h,cc} (base), cuda/cuda_engine.cc + cuda/cuda_loops_arch1.cc (arch1),
// dist/dist_engine.cc + dist/dist_loops.cc + dist/dist_loops_arch5.cc (arch5, FGNN), dist/dist_loops_arch6.cc (arch6),
// cuda/cuda_loops_arch7.cc (arch7),
// graph_pool.cc, workspace_pool.cc.
#pragma once
#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "eng_config.h"
#include "eng_dataset.h"
#include "eng_profiler.h"
#include "eng_queue.h"
#include "eng_shuffler.h"

namespace sam {

// size-bucketed device allocator (WorkspacePool, workspace_pool.cc): batches come and go every
// step, hipMalloc/hipFree must not be on that path
class DevicePool {
 public:
  ~DevicePool();
  void *Alloc(size_t bytes);
  void Free(void *p);

 private:
  std::mutex mu_;
  std::multimap<size_t, void *> free_;
  std::map<void *, size_t> live_;
};

struct TrainGraphView {  // TrainGraph, common.h:186-193
  const uint32_t *row = nullptr, *col = nullptr, *data = nullptr;
  size_t num_src = 0, num_dst = 0, num_edge = 0;
};

struct GraphBatch {      // Task, common.h:205-222 (trainer-side view)
  uint64_t key = 0;
  int num_layer = 0;
  TrainGraphView graphs[FGNN_MAX_LAYERS];
  const void *feat = nullptr;
  size_t feat_rows = 0;
  const void *label = nullptr;
  const uint32_t *input_nodes = nullptr;   // device (sampler ctx) or host; may be null
  const uint32_t *output_nodes = nullptr;
  size_t num_input = 0, num_output = 0;
  int input_device = -1, output_device = -1, device = -1;  // -1 = host memory
  // ownership
  fgnn_batch *fb = nullptr;           // arch1: pooled sampler-side buffers
  std::vector<void *> pooled;         // arch5 trainer: DevicePool allocations
  std::vector<void *> host_owned;     // malloc'd host arrays
  std::vector<std::shared_ptr<void>> shared;  // arch4 dynamic cache: buffers the engine may keep using as the cache
};

class GraphPool {        // graph_pool.cc:31-62
 public:
  explicit GraphPool(size_t max_size) : max_size_(max_size ? max_size : 1) {}
  std::shared_ptr<GraphBatch> Get();
  void Submit(std::shared_ptr<GraphBatch> b);
  bool Full();
  void Stop() { stop_ = true; }

 private:
  std::mutex mu_;
  std::queue<std::shared_ptr<GraphBatch>> q_;
  size_t max_size_;
  std::atomic<bool> stop_{false};
};

enum class DistType { Default, Sample, Extract, Switch };

struct DynamicCache;  // eng_dynamic.cc
struct DynamicCacheDeleter { void operator()(DynamicCache *p) const; };

class Engine {
 public:
  static Engine &Get();

  void Init();  // samgraph_init (arch1) / samgraph_data_init (arch5)
  void SampleInit(int worker_id, Context ctx);
  void TrainInit(int worker_id, Context ctx, DistType type);
  void Start();  // samgraph_start: background sampler + extractor threads (arch2-4); nothing to do in arch1 / arch5
  void Shutdown();
  void RunSampleOnce();
  void StartExtract(int count);
  uint64_t GetNextBatch();
  std::shared_ptr<GraphBatch> Current() { return current_; }

  bool Initialized() const { return initialized_; }
  size_t NumEpoch() const { return RC().num_epoch; }
  size_t NumStep() const { return num_step_; }
  size_t NumLocalStep() const { return shuffler_ ? shuffler_->NumLocalStep() : 0; }
  uint64_t BatchKey(uint64_t epoch, uint64_t step) const { return epoch * num_step_ + step; }
  Dataset &Data() { return ds_; }
  void ForwardBarrier() { outer_counter_++; }
  bool QueueStats(int ring, uint64_t out[6]) const { return mq_ && mq_->RingStats(ring, out); }
  bool RingMapping(int ring, int64_t out[3]) const { return mq_ && mq_->RingMapping(ring, out); }
  // the parent saw a child die (samgraph_wait_one_child): peers blocked on the queue give up instead of hanging
  void AbortQueue() { if (mq_) mq_->Abort(); }
  // sample_once only enqueues a batch; its profiler values appear when the publisher thread has published it.  The
  // profiler getters / reports of the C ABI call this first: everything enqueued so far is published (and logged).
  void SyncPublished() { if (publish_thread_.joinable()) PublishPending(); }
  // SAMGRAPH_EMPTY_FEAT=k: the feature table holds 2^k rows, node ids are masked before indexing it (the reference's
  // mock extraction, cpu_extraction.cc:47-62)
  // does the extractor's GPU also run this process's sampling chain (arch2-4 with both contexts on one device, arch6)?
  // its host-source gathers then stay small (fgnn_gather_rows_shared); an arch5 trainer has its GPU to itself.
  // SAMGRAPH_EXTRACT_SHARED_GPU=0/1 overrides (A/B runs; two arch5 ranks that share one development GPU)
  int ExtractorSharesGpu() const {
    static const int forced = [] { const char *e = getenv("SAMGRAPH_EXTRACT_SHARED_GPU"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced;
    return sampler_ != nullptr && device_ == tdevice_;
  }
  // workgroups of the one-launch extraction's link band (SAMGRAPH_EXTRACT_LINK_WGS=n fixes it: A/B runs)
  int ExtractLinkWgs() const {
    static const int forced = [] { const char *e = getenv("SAMGRAPH_EXTRACT_LINK_WGS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    return ExtractorSharesGpu() ? FGNN_LINK_WGS_SHARED : FGNN_LINK_WGS_DEDICATED;
  }
  uint32_t FeatRowMask() const {
    return RC().option_empty_feat ? (uint32_t)((1ull << RC().option_empty_feat) - 1) : 0xFFFFFFFFu;
  }

 private:
  // shared
  void UploadTopology(int device);
  void CreateSampler();
  void ReleaseBatch(GraphBatch *b);
  // arch1
  void InitArch1();
  void InitArch7();
  void InitSingleGPU(bool extract);
  void SampleOnceArch1();
  // arch2 / arch3 / arch4: sampler and extractor in ONE process (cuda_loops_arch{2,3,4}.cc), the arch5 halves
  // joined by an in-process queue
  void InitInProcess();
  void CreateQueue();
  void CreateSamplerSlots(size_t streams, size_t slots);
  static constexpr size_t kInProcessSamplerStreams = 2, kInProcessSamplerSlots = 2;
  // arch5 sampler
  void PreSample();
  void PreSampleStatic();
  void BuildCacheTable();
  void SampleOnceArch5();
  void PublishPending();          // returns once every batch enqueued so far has been published
  void PublishSlot(int slot);     // publisher thread: wait for the batch in `slot`, publish it, log its times
  void PublisherLoop();
  // arch4 with the dynamic cache prototype (eng_dynamic.cc)
  void InitDynamicCache();
  void SampleOnceDynamic();
  // arch5 trainer
  void TrainerOnce();
  void ExtractLoop(size_t count);
  struct ExtractCtx;
  void FinishBatch(int slot);
  void FlushOwedTail();
  void TrainerIssue(ExtractCtx &x, const void *taken, size_t taken_key);
  void TrainerComplete(ExtractCtx &x);
  void BuildTrainerCache();

  Dataset ds_;
  bool initialized_ = false, data_initialized_ = false;
  size_t num_step_ = 0;
  DistType dist_type_ = DistType::Default;
  int device_ = -1;             // sampler-side device (arch1: the only one)
  hipStream_t stream_ = nullptr;
  int tdevice_ = -1;            // trainer-side device (arch5 trainer process, arch2-4 extractor)
  hipStream_t tstream_ = nullptr;
  // trainer: a batch being received / extracted (eng_engine.cc: TrainerIssue / TrainerComplete)
  static constexpr int kExtractDepth = 4;
  struct ExtractCtx {
    hipStream_t st = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // brackets of the miss-row and cached-row gathers
    std::shared_ptr<GraphBatch> b;
    size_t mq_key = 0, miss_rows = 0, graph_bytes = 0, input_size = 0, output_size = 0;
    double recv_time = 0;
    uint64_t recv_us = 0;     // SAMGRAPH_DUMP_TRACE: wall clock when the receive began
    Timer t_copy;
    bool timed_gathers = false;
    uint32_t *d_check = nullptr;  // SAMGRAPH_HANDOFF_CHECK: result word of the verification kernel
    // one-launch extraction: per-workgroup start / end clocks (pinned host memory), workgroups of the launch and of
    // its link band
    unsigned long long *h_stamps = nullptr;
    size_t stamp_cap = 0, stamp_grid = 0, stamp_link = 0;
  };
  ExtractCtx xctx_[kExtractDepth];
  // where the extraction thread's time goes (reported at shutdown, log level info)
  struct { double recv = 0, issue = 0, pool_wait = 0, sync = 0; size_t n = 0; } xstat_;
  // the same for an arch5 sampler's sample_once calls
  struct { double slot_wait = 0, enqueue = 0, pub_wait = 0, pub_rest = 0; size_t n = 0; } sstat_;

  // device copies
  uint32_t *d_indptr_ = nullptr, *d_indices_ = nullptr;
  float *d_prefix_ = nullptr, *d_prob_ = nullptr;
  uint32_t *d_alias_ = nullptr;
  void *d_feat_ = nullptr;    // arch1: full table in HBM
  void *d_label_ = nullptr;
  uint32_t *d_cache_table_ = nullptr;  // direct-map table u32[num_node]
  void *d_cache_rows_ = nullptr;       // trainer: cached feature rows
  void *dev_host_feat_ = nullptr;      // device-visible address of the registered host feature table
  size_t num_cached_ = 0;

  fgnn_sampler *sampler_ = nullptr;
  std::unique_ptr<Shuffler> shuffler_;
  std::unique_ptr<GraphPool> pool_;
  std::shared_ptr<GraphBatch> current_;
  DevicePool dev_pool_;
  std::unique_ptr<DynamicCache, DynamicCacheDeleter> dyn_;  // declared after dev_pool_: destroyed before it

  // sampler-side batch buffers
  struct Slot {
    fgnn_batch *fb = nullptr;
    bool busy = false;           // arch1: handed to the trainer
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    hipStream_t st = nullptr;    // arch2-6: the slot's stream (arch5 sampler: shared by the slots i, i + streams, ...)
    bool owns_st = true;
    // arch5: message being written
    bool pending = false;
    size_t mq_key = 0;
    uint64_t key = 0;
    Timer started;
    uint64_t started_us = 0;  // SAMGRAPH_DUMP_TRACE: wall clock of the sample_once call
    uint32_t *d_msg_words = nullptr;  // SAMGRAPH_HANDOFF_CHECK: message length left by the pack kernel
    uint64_t seq = 0;                 // the batch's number in the kernel-level sampler (sample_begin -> sample_end)
  };
  std::vector<Slot> slots_;
  size_t next_slot_ = 0;
  // arch5 sampler: batches whose GPU work is enqueued, oldest first.  The publisher thread waits for each one's
  // completion and publishes it, so sample_once returns after ENQUEUEING a batch and no batch is ever held back until
  // the next call (the reference's PIPELINE branch keeps the newest batch unpublished until the next sample_once or
  // the end of the epoch, dist_loops_arch5.cc:108-146)
  std::deque<int> pub_q_;
  std::mutex pub_mu_;
  std::condition_variable pub_cv_;
  std::thread publish_thread_;
  bool pub_stop_ = false;
  // the batch whose sampling chain is enqueued and whose tail is owed to the next sample_once (eng_engine.cc:
  // SampleOnceArch5); enq_mu_ serialises the enqueueing thread and the publisher thread's finish of an overdue tail
  std::mutex enq_mu_;
  int tail_owed_ = -1;                          // slot index, under enq_mu_
  std::atomic<uint64_t> tail_since_us_{0};      // wall clock (us) since when; 0: none
  static constexpr uint64_t kOwedTailUs = 300;  // ~3 batch intervals

  // arch5 shared state (created before fork)
  MemoryQueue *mq_ = nullptr;
  int ring_id_ = -1;  // this sampler's HBM message ring (eng_queue.h), -1: none
  pthread_barrier_t *sampler_barrier_ = nullptr;
  std::thread extract_thread_, sample_thread_;
  std::atomic<bool> shutdown_{false};
  std::atomic<size_t> outer_counter_{0};
  // SAMGRAPH_HANDOFF_CHECK=n: this sampler appends a checksum to its first n messages; a receiver verifies every
  // message that carries one through the address it reads the payload from, and aborts on a mismatch
  size_t handoff_check_left_ = 0;
  int worker_id_ = 0;
};

// presample.hip
void PresampleCount(uint32_t *d_freq, const uint32_t *d_nodes, const uint32_t *d_n, size_t cap, hipStream_t st);
void PresampleRank(const uint32_t *d_freq, size_t num_node, uint32_t *h_rank, hipStream_t st);

}  // namespace sam
