// eng_profiler.h -- Profiler API surface (reference profiler.h:30-165, profiler.cc:371-557): dense
// tables [item][epoch*num_step+step]; the getters and item numbering are part of the Python contract.
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>

namespace sam {

constexpr int kNumLogInitItems = 22;
constexpr int kNumLogStepItems = 52;
constexpr int kNumLogEpochItems = 10;
constexpr int kNumTraceItems = 19;

// the items the engine itself fills (numbering == samgraph/common/__init__.py)
enum {
  kLogInitL1Common = 0, kLogInitL1Sampler = 1, kLogInitL1Trainer = 2, kLogInitL2LoadDataset = 3,
  kLogInitL2DistQueue = 4, kLogInitL2Presample = 5, kLogInitL2InternalState = 6, kLogInitL2BuildCache = 7,
};
enum {
  kLogL1NumSample = 0, kLogL1NumNode = 1, kLogL1SampleTime = 2, kLogL1SendTime = 3, kLogL1RecvTime = 4,
  kLogL1CopyTime = 5, kLogL1FeatureBytes = 8, kLogL1LabelBytes = 9, kLogL1IdBytes = 10, kLogL1GraphBytes = 11,
  kLogL1MissBytes = 12, kLogL2ShuffleTime = 15, kLogL2LastLayerSize = 17, kLogL2CoreSampleTime = 18,
  kLogL2ExtractTime = 22, kLogL2CacheCopyTime = 24, kLogL3CacheGetIndexTime = 46, kLogL3CacheCombineMissTime = 50,
  kLogL3CacheCombineCacheTime = 51,
};
enum {
  kLogEpochSampleTime = 0, kLogEpochSampleGetCacheMissIndexTime = 1, kLogEpochSampleSendTime = 2,
  kLogEpochSampleTotalTime = 3, kLogEpochCopyTime = 4, kLogEpochFeatureBytes = 8, kLogEpochMissBytes = 9,
};

enum {  // trace items (profiler.h:142-165)
  kL0Event_Train_Step = 0, kL1Event_Sample = 1, kL2Event_Sample_Shuffle = 2, kL2Event_Sample_Core = 3,
  kL2Event_Sample_IdRemap = 4, kL1Event_Copy = 5, kL2Event_Copy_Id = 6, kL2Event_Copy_Graph = 7,
  kL2Event_Copy_Extract = 8, kL2Event_Copy_FeatCopy = 9, kL2Event_Copy_CacheCopy = 10, kL1Event_Convert = 17,
  kL1Event_Train = 18,
};

class Profiler {
 public:
  static Profiler &Get();
  void Resize(size_t num_epoch, size_t num_step);
  void LogInit(int item, double v) { init_[item] = v; }
  void LogInitAdd(int item, double v) { init_[item] += v; }
  void LogStep(uint64_t key, int item, double v);
  void LogStepAdd(uint64_t key, int item, double v);
  void LogEpochAdd(uint64_t key, int item, double v);
  double GetLogInitValue(int item) const { return init_[item]; }
  double GetLogStepValue(uint64_t key, int item) const;
  double GetLogEpochValue(uint64_t epoch, int item) const;
  void ReportInit() const;
  void ReportStep(uint64_t epoch, uint64_t step) const;
  void ReportStepAverage(uint64_t epoch, uint64_t step) const;
  void ReportEpoch(uint64_t epoch) const;
  void ReportEpochAverage(uint64_t epoch) const;
  void TraceStep(uint64_t key, int item, uint64_t ts, bool begin);
  void DumpTrace() const;

 private:
  size_t num_step_ = 1, num_epoch_ = 1;
  double init_[kNumLogInitItems] = {0};
  std::vector<double> step_[kNumLogStepItems];
  std::vector<double> epoch_[kNumLogEpochItems];
  struct Trace { uint64_t key; int item; uint64_t begin, end; };
  std::vector<Trace> traces_;
  mutable std::mutex trace_mu_;  // the engine's threads and the script log events concurrently
};

}  // namespace sam
