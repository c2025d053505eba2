// eng_config.h -- RunConfig (reference run_config.h:31-94, operation.cc:45-169, run_config.cc:78-101)
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "eng_common.h"

namespace sam {

enum RunArch { kArch0 = 0, kArch1, kArch2, kArch3, kArch4, kArch5, kArch6, kArch7 };
enum SampleType { kKHop0 = 0, kKHop1, kWeightedKHop, kRandomWalk, kWeightedKHopPrefix, kKHop2, kWeightedKHopHashDedup };
enum CachePolicy {
  kCacheByDegree = 0, kCacheByHeuristic, kCacheByPreSample, kCacheByDegreeHop, kCacheByPreSampleStatic,
  kCacheByFakeOptimal, kDynamicCache, kCacheByRandom
};

struct RunConfig {
  std::unordered_map<std::string, std::string> raw;
  std::string dataset_path;
  int run_arch = kArch1;
  int sample_type = kKHop2;
  size_t batch_size = 0, num_epoch = 0;
  Context sampler_ctx, trainer_ctx;
  int cache_policy = kCacheByPreSample;
  double cache_percentage = 0.0;
  size_t max_sampling_jobs = 1, max_copying_jobs = 1;
  std::vector<size_t> fanout;
  size_t random_walk_length = 0, num_random_walk = 0, num_neighbor = 0, num_layer = 0;
  double random_walk_restart_prob = 0.0;
  size_t num_sample_worker = 1, num_train_worker = 1;
  size_t worker_id = 0, num_worker = 1;  // arch6 (num_worker), arch7 (both)
  bool have_switcher = false;
  int barriered_epoch = 0, presample_epoch = 0;
  int omp_thread_num = 1;
  uint64_t seed = 0x5A4D47;
  bool is_configured = false;
  // environment (run_config.cc:78-101)
  int profile_level = 0;
  bool option_dump_trace = false, option_sanity_check = false;
  size_t option_empty_feat = 0;
  size_t mq_budget_bytes = 8ull << 30;  // SAMGRAPH_MQ_BYTES: total size of the shared queue
  // SAMGRAPH_DEVICE_RING_SLOTS: message slots per sampler in its HBM (eng_queue.h).  Unset: as many as the queue has
  // slots -- max(max_sampling_jobs, 8) where sampler and extractor share a process (arch2-4, arch6: a plain device
  // pointer; CreateDeviceRing clamps to the queue's size), for arch5 as many as the host ring has slots
  // (170 x ~46 MB = 7.8 GB of the 288 at [25,10] x 8000; mapped by the trainers with hipIpc, a trainer that cannot map
  // it makes the whole job fall back to the host ring): whenever the trainers are the slower side the queue fills up,
  // and with fewer device slots than queue slots the surplus messages would travel through pinned host memory -- over
  // the same host link the trainers' miss rows need
  long device_ring_slots = -1;
  size_t DeviceRingSlots() const {
    if (device_ring_slots >= 0) return (size_t)device_ring_slots;
    return 170;  // clamped to the queue's slots (MemoryQueue::CreateDeviceRing)
  }

  bool UseGPUCache() const { return cache_percentage > 0 && run_arch != kArch1; }  // run_config.h:84-86
  void Parse(const char **keys, const char **vals, size_t n);
};

RunConfig &RC();

}  // namespace sam
