#include "eng_config.h"

#include <cstring>

namespace sam {

int MinLogLevel() {
  static int lvl = [] {
    const char *e = getenv("SAMGRAPH_LOG_LEVEL");
    if (!e) return (int)kWarning;
    std::string s(e);
    if (s == "trace") return (int)kTrace;
    if (s == "debug") return (int)kDebug;
    if (s == "info") return (int)kInfo;
    if (s == "warn" || s == "warning") return (int)kWarning;
    if (s == "error") return (int)kError;
    return (int)kWarning;
  }();
  return lvl;
}

Context::Context(const std::string &name) {
  const size_t d = name.find(':');
  SAM_CHECK(d != std::string::npos) << "bad context string " << name;
  const std::string dev = name.substr(0, d);
  device_id = std::stoi(name.substr(d + 1));
  if (dev == "cpu") device_type = kCPU;
  else if (dev == "cuda") device_type = kGPU;
  else if (dev == "mmap") device_type = kMMAP;
  else SAM_FATAL << "bad context string " << name;
}

std::string Context::Str() const {
  const char *n[] = {"cpu", "mmap", "cuda"};
  return std::string(n[device_type]) + ":" + std::to_string(device_id);
}

RunConfig &RC() {
  static RunConfig rc;
  return rc;
}

static bool EnvOn(const char *k) {
  const char *e = getenv(k);
  return e && (!strcmp(e, "ON") || !strcmp(e, "1"));
}

void RunConfig::Parse(const char **keys, const char **vals, size_t n) {
  SAM_CHECK(!is_configured);
  for (size_t i = 0; i < n; ++i) raw[keys[i]] = vals[i];
  auto need = [&](const char *k) -> const std::string & {
    SAM_CHECK(raw.count(k)) << "missing config key " << k;
    return raw[k];
  };
  dataset_path = need("dataset_path");
  run_arch = std::stoi(need("_arch"));
  sample_type = std::stoi(need("_sample_type"));
  batch_size = std::stoull(need("batch_size"));
  num_epoch = std::stoull(need("num_epoch"));
  cache_policy = std::stoi(need("_cache_policy"));
  cache_percentage = std::stod(need("cache_percentage"));
  max_sampling_jobs = std::stoull(need("max_sampling_jobs"));
  max_copying_jobs = std::stoull(need("max_copying_jobs"));
  omp_thread_num = std::stoi(need("omp_thread_num"));
  switch (run_arch) {
    case kArch1: case kArch2: case kArch3: case kArch4:
      sampler_ctx = Context(need("sampler_ctx"));
      trainer_ctx = Context(need("trainer_ctx"));
      break;
    case kArch5:
      num_sample_worker = std::stoull(need("num_sample_worker"));
      num_train_worker = std::stoull(need("num_train_worker"));
      have_switcher = raw.count("have_switcher") ? std::stoi(raw["have_switcher"]) != 0 : false;
      break;
    case kArch6:  // operation.cc:104-108
      num_worker = std::stoull(need("num_worker"));
      num_sample_worker = num_train_worker = num_worker;
      break;
    case kArch7:  // operation.cc:109-118
      worker_id = std::stoull(need("worker_id"));
      num_worker = std::stoull(need("num_worker"));
      sampler_ctx = Context(need("sampler_ctx"));
      trainer_ctx = Context(need("trainer_ctx"));
      SAM_CHECK(num_worker > 0 && worker_id < num_worker) << "bad worker_id / num_worker";
      break;
    default:
      SAM_FATAL << "run arch " << run_arch << " is not built (supported: arch1-arch7; arch0 is the reference's CPU "
                   "sampling mode)";
  }
  if (sample_type != kRandomWalk) {
    const size_t nf = std::stoull(need("num_fanout"));
    std::stringstream ss(need("fanout"));
    for (size_t i = 0; i < nf; ++i) {
      size_t f = 0;
      ss >> f;
      SAM_CHECK(f > 0) << "bad fanout list";
      fanout.push_back(f);
    }
  } else {
    random_walk_length = std::stoull(need("random_walk_length"));
    random_walk_restart_prob = std::stod(need("random_walk_restart_prob"));
    num_random_walk = std::stoull(need("num_random_walk"));
    num_neighbor = std::stoull(need("num_neighbor"));
    num_layer = std::stoull(need("num_layer"));
    fanout.assign(num_layer, num_neighbor);
  }
  SAM_CHECK(!fanout.empty() && fanout.size() <= FGNN_MAX_LAYERS);
  barriered_epoch = raw.count("barriered_epoch") ? std::stoi(raw["barriered_epoch"]) : 0;
  presample_epoch = raw.count("presample_epoch") ? std::stoi(raw["presample_epoch"]) : 0;
  if (raw.count("seed")) seed = std::stoull(raw["seed"], nullptr, 0);
  // environment
  if (const char *e = getenv("SAMGRAPH_PROFILE_LEVEL")) profile_level = atoi(e);
  option_dump_trace = EnvOn("SAMGRAPH_DUMP_TRACE");
  option_sanity_check = EnvOn("SAMGRAPH_SANITY_CHECK");
  if (const char *e = getenv("SAMGRAPH_EMPTY_FEAT")) option_empty_feat = strtoull(e, nullptr, 10);
  if (const char *e = getenv("SAMGRAPH_MQ_BYTES")) mq_budget_bytes = strtoull(e, nullptr, 10);
  if (const char *e = getenv("SAMGRAPH_DEVICE_RING_SLOTS")) device_ring_slots = strtol(e, nullptr, 10);
  switch (sample_type) {
    case kKHop0: case kKHop1: case kKHop2: case kWeightedKHop: case kWeightedKHopPrefix: case kRandomWalk: break;
    case kWeightedKHopHashDedup:
      // the reference's per-thread table has 50 slots (cuda_sampling_weighted_khop_hash_dedup.cu:43-72); with a
      // larger fanout its sampler never returns
      for (size_t f : fanout) SAM_CHECK(f <= 50) << "weighted_khop_hash_dedup: fanout " << f << " > 50";
      break;
    default: SAM_FATAL << "unknown sample type " << sample_type;
  }
  is_configured = true;
}

}  // namespace sam
