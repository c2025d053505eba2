// eng_operation.cc -- the samgraph_* C ABI (include/samgraph.h; reference operation.{h,cc},
// torch/adapter.{h,cc}).
#include <cstdlib>
#include <signal.h>
#include <sys/resource.h>
#include <sys/wait.h>

#include "eng_engine.h"
#include "samgraph.h"
#include "samgraph_ext.h"

using namespace sam;

namespace {
std::shared_ptr<GraphBatch> CurrentChecked(uint64_t key) {
  auto b = Engine::Get().Current();
  SAM_CHECK(b) << "no current batch: call samgraph_get_next_batch first";
  SAM_CHECK_EQ(b->key, key);  // adapter.cc:52
  return b;
}
}  // namespace

extern "C" {

void samgraph_config(const char **config_keys, const char **config_values, const size_t num_config_items) {
  // An arch5 sampler keeps three batch streams busy next to the engine's own stream and the null stream, and the HIP
  // runtime multiplexes a process's streams onto 4 hardware queues by default: two of the batch streams then share a
  // queue and their chains take turns instead of overlapping (one sampler GPU: 142 -> 120 us per papers100M batch,
  // GPU idle 21 % -> 7 %).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the
  // process -- after this call in the reference's call order (config before any init); a value set by the user wins.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  // the samplers' HBM message rings go to the trainer processes through hipIpc handles: on hosts whose driver only
  // supports dmabuf IPC the export fails ("invalid argument") unless the legacy mode is off, and the hand-off would
  // fall back to the pinned host ring.  Same rule: read at the runtime's first call, a value set by the user wins
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  // every slot of a sampler's HBM ring is its own allocation and its own IPC handle (eng_queue.h): a trainer of an
  // S-sampler job imports up to S x 170 buffers.  On the development boxes' driver an imported slot keeps no
  // descriptor open (13-25 open files per rank with every ring mapped, bench.py's pipeline.open_files); a driver
  // whose dmabuf import does must not run into the usual soft limit of 1024 (the hard limit is the user's to set)
  {
    struct rlimit rl;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < 65536 && rl.rlim_cur < rl.rlim_max) {
      rl.rlim_cur = rl.rlim_max < 65536 ? rl.rlim_max : 65536;
      (void)setrlimit(RLIMIT_NOFILE, &rl);
    }
  }
  RC().Parse(config_keys, config_values, num_config_items);
}

void samgraph_init(void) {
  SAM_CHECK(RC().is_configured);
  SAM_CHECK((RC().run_arch >= kArch1 && RC().run_arch <= kArch4) || RC().run_arch == kArch7)
      << "samgraph_init is the single-process entry (arch1-4, arch7)";
  Engine::Get().Init();
}

void samgraph_start(void) { Engine::Get().Start(); }
void samgraph_shutdown(void) { Engine::Get().Shutdown(); }

size_t samgraph_num_epoch(void) { return Engine::Get().NumEpoch(); }
size_t samgraph_steps_per_epoch(void) { return Engine::Get().NumStep(); }
size_t samgraph_num_class(void) { return Engine::Get().Data().num_class; }
size_t samgraph_feat_dim(void) { return Engine::Get().Data().feat_dim; }

uint64_t samgraph_get_next_batch(void) { return Engine::Get().GetNextBatch(); }
void samgraph_sample_once(void) { Engine::Get().RunSampleOnce(); }

size_t samgraph_get_graph_num_src(uint64_t key, int graph_id) { return CurrentChecked(key)->graphs[graph_id].num_src; }
size_t samgraph_get_graph_num_dst(uint64_t key, int graph_id) { return CurrentChecked(key)->graphs[graph_id].num_dst; }
size_t samgraph_get_graph_num_edge(uint64_t key, int graph_id) {
  return CurrentChecked(key)->graphs[graph_id].num_edge;
}

void samgraph_log_step(uint64_t epoch, uint64_t step, int item, double val) {
  SAM_CHECK_LT(item, kNumLogStepItems);
  Profiler::Get().LogStep(Engine::Get().BatchKey(epoch, step), item, val);
}
void samgraph_log_step_add(uint64_t epoch, uint64_t step, int item, double val) {
  SAM_CHECK_LT(item, kNumLogStepItems);
  Profiler::Get().LogStepAdd(Engine::Get().BatchKey(epoch, step), item, val);
}
void samgraph_log_epoch_add(uint64_t epoch, int item, double val) {
  SAM_CHECK_LT(item, kNumLogEpochItems);
  Profiler::Get().LogEpochAdd(Engine::Get().BatchKey(epoch, 0), item, val);
}
double samgraph_get_log_init_value(int item) {
  SAM_CHECK_LT(item, kNumLogInitItems);
  return Profiler::Get().GetLogInitValue(item);
}
double samgraph_get_log_step_value(uint64_t epoch, uint64_t step, int item) {
  SAM_CHECK_LT(item, kNumLogStepItems);
  Engine::Get().SyncPublished();
  return Profiler::Get().GetLogStepValue(Engine::Get().BatchKey(epoch, step), item);
}
double samgraph_get_log_epoch_value(uint64_t epoch, int item) {
  SAM_CHECK_LT(item, kNumLogEpochItems);
  Engine::Get().SyncPublished();
  return Profiler::Get().GetLogEpochValue(epoch, item);
}

void samgraph_report_init(void) { Profiler::Get().ReportInit(); }
void samgraph_report_step(uint64_t epoch, uint64_t step) {
  Engine::Get().SyncPublished();
  Profiler::Get().ReportStep(epoch, step);
}
void samgraph_report_step_average(uint64_t epoch, uint64_t step) {
  Engine::Get().SyncPublished();
  Profiler::Get().ReportStepAverage(epoch, step);
}
void samgraph_report_epoch(uint64_t epoch) {
  Engine::Get().SyncPublished();
  Profiler::Get().ReportEpoch(epoch);
}
void samgraph_report_epoch_average(uint64_t epoch) {
  Engine::Get().SyncPublished();
  Profiler::Get().ReportEpochAverage(epoch);
}
void samgraph_report_node_access(void) {}  // SAMGRAPH_LOG_NODE_ACCESS analysis is an offline study (profiler.cc:568-866)

void samgraph_trace_step_begin(uint64_t key, int item, uint64_t ts) { Profiler::Get().TraceStep(key, item, ts, true); }
void samgraph_trace_step_end(uint64_t key, int item, uint64_t ts) { Profiler::Get().TraceStep(key, item, ts, false); }
void samgraph_trace_step_begin_now(uint64_t key, int item) {
  Profiler::Get().TraceStep(key, item, Timer::NowMicro(), true);
}
void samgraph_trace_step_end_now(uint64_t key, int item) {
  Profiler::Get().TraceStep(key, item, Timer::NowMicro(), false);
}
void samgraph_dump_trace(void) { Profiler::Get().DumpTrace(); }

void samgraph_forward_barrier(void) { Engine::Get().ForwardBarrier(); }

// include/samgraph_ext.h
int samgraph_ext_queue_stats(int ring, uint64_t out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
  return Engine::Get().QueueStats(ring, out) ? 0 : -1;
}

int samgraph_ext_ring_mapping(int ring, int64_t out[3]) {
  out[0] = 0;
  out[1] = out[2] = -1;
  return Engine::Get().RingMapping(ring, out) ? 0 : -1;
}

void samgraph_data_init(void) {
  SAM_CHECK(RC().is_configured);
  SAM_CHECK(RC().run_arch == kArch5 || RC().run_arch == kArch6)
      << "samgraph_data_init is the multi-process entry (arch5, arch6)";
  Engine::Get().Init();
}
void samgraph_sample_init(int worker_id, const char *ctx) {
  SAM_CHECK(RC().is_configured);
  Engine::Get().SampleInit(worker_id, Context(std::string(ctx)));
}
void samgraph_train_init(int worker_id, const char *ctx) {
  SAM_CHECK(RC().is_configured);
  Engine::Get().TrainInit(worker_id, Context(std::string(ctx)), DistType::Extract);
}
void samgraph_extract_start(int count) { Engine::Get().StartExtract(count); }
void samgraph_switch_init(int worker_id, const char *ctx, double cache_percentage) {
  RC().cache_percentage = cache_percentage;  // operation.cc:363
  SAM_CHECK(RC().is_configured);
  Engine::Get().TrainInit(worker_id, Context(std::string(ctx)), DistType::Switch);
}
size_t samgraph_num_local_step(void) { return Engine::Get().NumLocalStep(); }

int samgraph_wait_one_child(void) {  // operation.cc:374-385
  int child_stat = 0;
  pid_t pid = waitpid(-1, &child_stat, 0);
  if (WEXITSTATUS(child_stat) != 0) {
    SAM_LOG(kError) << "detect a terminated child " << pid << ", status is " << WEXITSTATUS(child_stat);
    Engine::Get().AbortQueue();
    return 1;
  } else if (WIFSIGNALED(child_stat) && (WTERMSIG(child_stat) == SIGABRT)) {
    SAM_LOG(kError) << "detect an aborted child " << pid;
    Engine::Get().AbortQueue();
    return 1;
  }
  return 0;
}

// ---- tensor getters (adapter.cc:48-192) ------------------------------------------------------------

const void *samgraph_torch_get_graph_feat_ptr(uint64_t key, size_t *num_rows, size_t *dim, int *dtype, int *device) {
  auto b = CurrentChecked(key);
  *num_rows = b->feat_rows;
  *dim = Engine::Get().Data().feat_dim;
  *dtype = FGNN_F32;
  *device = b->device;
  return b->feat;
}
const void *samgraph_torch_get_graph_label_ptr(uint64_t key, size_t *num, int *dtype, int *device) {
  auto b = CurrentChecked(key);
  *num = b->num_output;
  *dtype = FGNN_I64;
  *device = b->device;
  return b->label;
}
const uint32_t *samgraph_torch_get_graph_row_ptr(uint64_t key, int layer, size_t *num, int *device) {
  auto b = CurrentChecked(key);
  SAM_CHECK(layer >= 0 && layer < b->num_layer);
  *num = b->graphs[layer].num_edge;
  *device = b->device;
  return b->graphs[layer].row;
}
const uint32_t *samgraph_torch_get_graph_col_ptr(uint64_t key, int layer, size_t *num, int *device) {
  auto b = CurrentChecked(key);
  SAM_CHECK(layer >= 0 && layer < b->num_layer);
  *num = b->graphs[layer].num_edge;
  *device = b->device;
  return b->graphs[layer].col;
}
const uint32_t *samgraph_torch_get_graph_data_ptr(uint64_t key, int layer, size_t *num, int *device) {
  auto b = CurrentChecked(key);
  SAM_CHECK(layer >= 0 && layer < b->num_layer);
  *num = b->graphs[layer].data ? b->graphs[layer].num_edge : 0;
  *device = b->device;
  return b->graphs[layer].data;
}
const uint32_t *samgraph_torch_get_graph_input_nodes_ptr(uint64_t key, size_t *num, int *device) {
  auto b = CurrentChecked(key);
  *num = b->input_nodes ? b->num_input : 0;
  *device = b->input_device;
  return b->input_nodes;
}
const uint32_t *samgraph_torch_get_graph_output_nodes_ptr(uint64_t key, size_t *num, int *device) {
  auto b = CurrentChecked(key);
  *num = b->num_output;
  *device = b->output_device;
  return b->output_nodes;
}
const void *samgraph_torch_get_dataset_feat_ptr(size_t *num_rows, size_t *dim, int *dtype) {
  auto &d = Engine::Get().Data();
  *num_rows = d.feat_rows;
  *dim = d.feat_dim;
  *dtype = FGNN_F32;
  return d.feat.ptr;
}
const void *samgraph_torch_get_dataset_label_ptr(size_t *num, int *dtype) {
  auto &d = Engine::Get().Data();
  *num = d.num_node;
  *dtype = FGNN_I64;
  return d.label.ptr;
}

}  // extern "C"
