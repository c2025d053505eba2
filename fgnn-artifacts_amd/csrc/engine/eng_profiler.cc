#include "eng_profiler.h"

#include <strings.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace sam {

Profiler &Profiler::Get() {
  static Profiler p;
  return p;
}

void Profiler::Resize(size_t num_epoch, size_t num_step) {
  num_epoch_ = num_epoch ? num_epoch : 1;
  num_step_ = num_step ? num_step : 1;
  for (auto &v : step_) v.assign(num_epoch_ * num_step_, 0.0);
  for (auto &v : epoch_) v.assign(num_epoch_, 0.0);
}

void Profiler::LogStep(uint64_t key, int item, double v) {
  if (item >= 0 && item < kNumLogStepItems && key < step_[item].size()) step_[item][key] = v;
}
void Profiler::LogStepAdd(uint64_t key, int item, double v) {
  if (item >= 0 && item < kNumLogStepItems && key < step_[item].size()) step_[item][key] += v;
}
void Profiler::LogEpochAdd(uint64_t key, int item, double v) {
  const uint64_t e = key / num_step_;
  if (item >= 0 && item < kNumLogEpochItems && e < epoch_[item].size()) epoch_[item][e] += v;
}
double Profiler::GetLogStepValue(uint64_t key, int item) const {
  return (item >= 0 && item < kNumLogStepItems && key < step_[item].size()) ? step_[item][key] : 0.0;
}
double Profiler::GetLogEpochValue(uint64_t epoch, int item) const {
  return (item >= 0 && item < kNumLogEpochItems && epoch < epoch_[item].size()) ? epoch_[item][epoch] : 0.0;
}

// ---- reports (profiler.cc:371-557) -------------------------------------------------------------------------------
// The reference prints fixed tables per SAMGRAPH_PROFILE_LEVEL (1: L1 items, 2: + L2, 3: + L3).  Same gating and the
// same item names here, from one name table per item family; an item this engine has no separate step for (its fused
// kernels do in one launch what the reference times as "khop count edge", "walk topk step 1-11", ...) reads 0.
namespace {

const char *const kInitNames[kNumLogInitItems] = {
    "L1Common", "L1Sampler", "L1Trainer", "L2LoadDataset", "L2DistQueue", "L2Presample", "L2InternalState", "L2BuildCache",
    "L3LoadDatasetMMap", "L3LoadDatasetCopy", "L3DistQueueAlloc", "L3DistQueuePin", "L3DistQueuePush", "L3PresampleInit",
    "L3PresampleSample", "L3PresampleCopy", "L3PresampleCount", "L3PresampleSort", "L3PresampleReset",
    "L3PresampleGetRank", "L3InternalStateCreateCtx", "L3InternalStateCreateStream"};
const char *const kStepNames[kNumLogStepItems] = {
    "L1NumSample", "L1NumNode", "L1SampleTime", "L1SendTime", "L1RecvTime", "L1CopyTime", "L1ConvertTime", "L1TrainTime",
    "L1FeatureBytes", "L1LabelBytes", "L1IdBytes", "L1GraphBytes", "L1MissBytes", "L1PrefetchAdvanced",
    "L1GetNeighbourTime", "L2ShuffleTime", "L2LastLayerTime", "L2LastLayerSize", "L2CoreSampleTime", "L2IdRemapTime",
    "L2GraphCopyTime", "L2IdCopyTime", "L2ExtractTime", "L2FeatCopyTime", "L2CacheCopyTime", "L3KHopSampleCooTime",
    "L3KHopSampleSortCooTime", "L3KHopSampleCountEdgeTime", "L3KHopSampleCompactEdgesTime", "L3RandomWalkSampleCooTime",
    "L3RandomWalkTopKTime", "L3RandomWalkTopKStep1Time", "L3RandomWalkTopKStep2Time", "L3RandomWalkTopKStep3Time",
    "L3RandomWalkTopKStep4Time", "L3RandomWalkTopKStep5Time", "L3RandomWalkTopKStep6Time", "L3RandomWalkTopKStep7Time",
    "L3RandomWalkTopKStep8Time", "L3RandomWalkTopKStep9Time", "L3RandomWalkTopKStep10Time", "L3RandomWalkTopKStep11Time",
    "L3RemapFillUniqueTime", "L3RemapPopulateTime", "L3RemapMapNodeTime", "L3RemapMapEdgeTime", "L3CacheGetIndexTime",
    "L3CacheCopyIndexTime", "L3CacheExtractMissTime", "L3CacheCopyMissTime", "L3CacheCombineMissTime",
    "L3CacheCombineCacheTime"};
const char *const kEpochNames[kNumLogEpochItems] = {"SampleTime", "SampleGetCacheMissIndexTime", "SampleSendTime",
                                                    "SampleTotalTime", "CopyTime", "ConvertTime", "TrainTime",
                                                    "TotalTime", "FeatureBytes", "MissBytes"};
const char *const kTraceNames[kNumTraceItems] = {
    "kL0Event_Train_Step", "kL1Event_Sample", "kL2Event_Sample_Shuffle", "kL2Event_Sample_Core",
    "kL2Event_Sample_IdRemap", "kL1Event_Copy", "kL2Event_Copy_Id", "kL2Event_Copy_Graph", "kL2Event_Copy_Extract",
    "kL2Event_Copy_FeatCopy", "kL2Event_Copy_CacheCopy", "kL3Event_Copy_CacheCopy_GetIndex",
    "kL3Event_Copy_CacheCopy_CopyIndex", "kL3Event_Copy_CacheCopy_ExtractMiss", "kL3Event_Copy_CacheCopy_CopyMiss",
    "kL3Event_Copy_CacheCopy_CombineMiss", "kL3Event_Copy_CacheCopy_CombineCache", "kL1Event_Convert", "kL1Event_Train"};

int ProfileLevel() {
  const char *e = getenv("SAMGRAPH_PROFILE_LEVEL");
  const int v = e ? atoi(e) : 0;
  return v < 0 ? 0 : v > 3 ? 3 : v;
}

std::string Readable(const char *name, double v) {
  char buf[64];
  const std::string n(name);
  const bool bytes = n.size() > 5 && n.compare(n.size() - 5, 5, "Bytes") == 0;
  const bool time = n.size() > 4 && n.compare(n.size() - 4, 4, "Time") == 0;
  if (bytes) {  // ToReadableSize (common.cc)
    const char *unit[] = {"Bytes", "KB", "MB", "GB", "TB"};
    int u = 0;
    while (v >= 1024.0 && u < 4) { v /= 1024.0; ++u; }
    snprintf(buf, sizeof(buf), "%.2f %s", v, unit[u]);
  } else if (time) {
    snprintf(buf, sizeof(buf), "%.4lf", v);
  } else {
    snprintf(buf, sizeof(buf), "%.0lf", v);
  }
  return buf;
}

// one "[<title> Profiler Level l ...]" block per level up to the configured one, four items per line
void PrintLevels(const char *title, const char *where, const char *const *names, int count, const double *vals,
                 int max_level) {
  for (int level = 1; level <= max_level; ++level) {
    bool any = false;
    int col = 0;
    for (int i = 0; i < count; ++i) {
      if (names[i][0] != 'L' || names[i][1] != '0' + level) continue;
      if (!any) printf("    [%s Profiler Level %d%s]\n", title, level, where);
      any = true;
      printf("%s%s %s", col == 0 ? "        L" : " | ", col == 0 ? std::to_string(level).c_str() : "",
             (std::string(names[i] + 2) + " " + Readable(names[i], vals[i])).c_str());
      if (++col == 4) { printf("\n"); col = 0; }
    }
    if (col) printf("\n");
  }
}

}  // namespace

void Profiler::ReportInit() const {
  PrintLevels("Init", "", kInitNames, kNumLogInitItems, init_, std::max(1, ProfileLevel()));
}

void Profiler::ReportStep(uint64_t epoch, uint64_t step) const {
  const uint64_t key = epoch * num_step_ + step;
  double v[kNumLogStepItems];
  for (int i = 0; i < kNumLogStepItems; ++i) v[i] = GetLogStepValue(key, i);
  char where[64];
  snprintf(where, sizeof(where), " E%lu S%lu", (unsigned long)epoch, (unsigned long)step);
  PrintLevels("Step(profile)", where, kStepNames, kNumLogStepItems, v, std::max(1, ProfileLevel()));
}

void Profiler::ReportStepAverage(uint64_t epoch, uint64_t step) const {
  const uint64_t n = epoch * num_step_ + step + 1;
  double v[kNumLogStepItems] = {0};
  for (int i = 0; i < kNumLogStepItems; ++i) {
    for (uint64_t k = 0; k < n && k < step_[i].size(); ++k) v[i] += step_[i][k];
    v[i] /= (double)n;
  }
  char where[64];
  snprintf(where, sizeof(where), " E%lu S%lu", (unsigned long)epoch, (unsigned long)step);
  PrintLevels("Step(average)", where, kStepNames, kNumLogStepItems, v, std::max(1, ProfileLevel()));
}

static void PrintEpoch(const char *title, uint64_t epoch, const double *v) {
  printf("    [%s E%lu]\n       ", title, (unsigned long)epoch);
  for (int i = 0; i < kNumLogEpochItems; ++i)
    printf(" %s %s%s", kEpochNames[i], Readable(kEpochNames[i], v[i]).c_str(), i + 1 < kNumLogEpochItems ? " |" : "\n");
}

void Profiler::ReportEpoch(uint64_t epoch) const {
  double v[kNumLogEpochItems];
  for (int i = 0; i < kNumLogEpochItems; ++i) v[i] = GetLogEpochValue(epoch, i);
  PrintEpoch("Epoch(profile)", epoch, v);
}

void Profiler::ReportEpochAverage(uint64_t epoch) const {
  double v[kNumLogEpochItems] = {0};
  for (int i = 0; i < kNumLogEpochItems; ++i) {
    for (uint64_t e = 0; e <= epoch && e < epoch_[i].size(); ++e) v[i] += epoch_[i][e];
    v[i] /= (double)(epoch + 1);
  }
  PrintEpoch("Epoch(average)", epoch, v);
}

void Profiler::TraceStep(uint64_t key, int item, uint64_t ts, bool begin) {
  if (item < 0 || item >= kNumTraceItems) return;
  std::lock_guard<std::mutex> lk(trace_mu_);
  if (begin) {
    traces_.push_back({key, item, ts, 0});
  } else {
    for (auto it = traces_.rbegin(); it != traces_.rend(); ++it)
      if (it->key == key && it->item == item) { it->end = ts; break; }
  }
}

// Chrome trace-event JSON (profiler.cc:286-364): one "B" and one "E" record per finished event, named
// "<trace item>-<batch key>", thread lanes 0 (whole step) / 1 (sample) / 2 (copy) / 3 (convert, train).  Only when
// SAMGRAPH_DUMP_TRACE is set, like the reference; the reference writes to stderr -- so does this unless the variable
// holds a path (anything but 1 / ON / true), in which case the JSON goes to that file.
void Profiler::DumpTrace() const {
  const char *e = getenv("SAMGRAPH_DUMP_TRACE");
  if (!e || !*e || !strcmp(e, "0") || !strcasecmp(e, "off") || !strcasecmp(e, "false")) return;
  const bool to_file = strcmp(e, "1") && strcasecmp(e, "on") && strcasecmp(e, "true");
  FILE *f = to_file ? fopen(e, "w") : stderr;
  if (!f) {
    fprintf(stderr, "samgraph_dump_trace: cannot write %s\n", e);
    return;
  }
  std::lock_guard<std::mutex> lk(trace_mu_);
  fprintf(f, "[\n");
  bool first = true;
  for (int item = 0; item < kNumTraceItems; ++item) {
    const int tid = item < 1 ? 0 : item < 5 ? 1 : item < 17 ? 2 : 3;
    for (const Trace &t : traces_) {
      if (t.item != item || t.begin == 0) continue;
      if (t.end == 0) {
        fprintf(stderr, "samgraph_dump_trace: an event without end (%s-%lu)\n", kTraceNames[item], (unsigned long)t.key);
        continue;
      }
      for (int ph = 0; ph < 2; ++ph) {
        fprintf(f, "%s{\"name\":\"%s-%lu\",\"ph\":\"%s\",\"pid\":0,\"tid\":%d,\"ts\":%lu,\"cat\":\"\",\"id\":0}\n",
                first ? "" : ",", kTraceNames[item], (unsigned long)t.key, ph == 0 ? "B" : "E", tid,
                (unsigned long)(ph == 0 ? t.begin : t.end));
        first = false;
      }
    }
  }
  fprintf(f, "]\n");
  if (to_file) fclose(f);
}

}  // namespace sam
