#include "eng_profiler.h"

#include <cstdio>

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

void Profiler::ReportInit() const {
  printf("    [Init Profiler Level 1]\n        L1  common %.4lf | sampler %.4lf | trainer %.4lf\n"
         "        L2  load dataset %.4lf | dist queue %.4lf | presample %.4lf | internal state %.4lf | "
         "build cache %.4lf\n",
         init_[0], init_[1], init_[2], init_[3], init_[4], init_[5], init_[6], init_[7]);
}

void Profiler::ReportStep(uint64_t epoch, uint64_t step) const {
  const uint64_t key = epoch * num_step_ + step;
  printf("    [Step(profile) E%lu S%lu]\n        L1  sample %.4lf | send %.4lf | recv %.4lf | copy %.4lf | "
         "num node %.0lf | num sample %.0lf\n",
         (unsigned long)epoch, (unsigned long)step, GetLogStepValue(key, kLogL1SampleTime),
         GetLogStepValue(key, kLogL1SendTime), GetLogStepValue(key, kLogL1RecvTime),
         GetLogStepValue(key, kLogL1CopyTime), GetLogStepValue(key, kLogL1NumNode),
         GetLogStepValue(key, kLogL1NumSample));
}

void Profiler::ReportStepAverage(uint64_t epoch, uint64_t step) const {
  const uint64_t n = epoch * num_step_ + step + 1;
  double s[kNumLogStepItems] = {0};
  for (int i = 0; i < kNumLogStepItems; ++i)
    for (uint64_t k = 0; k < n && k < step_[i].size(); ++k) s[i] += step_[i][k];
  printf("    [Step(average) E%lu S%lu]\n        L1  sample %.4lf | send %.4lf | recv %.4lf | copy %.4lf | "
         "num node %.0lf | num sample %.0lf\n",
         (unsigned long)epoch, (unsigned long)step, s[kLogL1SampleTime] / n, s[kLogL1SendTime] / n,
         s[kLogL1RecvTime] / n, s[kLogL1CopyTime] / n, s[kLogL1NumNode] / n, s[kLogL1NumSample] / n);
}

void Profiler::ReportEpoch(uint64_t epoch) const {
  printf("    [Epoch(profile) E%lu]\n        sample %.4lf | get cache miss index %.4lf | send %.4lf | "
         "sample total %.4lf | copy %.4lf\n",
         (unsigned long)epoch, GetLogEpochValue(epoch, 0), GetLogEpochValue(epoch, 1), GetLogEpochValue(epoch, 2),
         GetLogEpochValue(epoch, 3), GetLogEpochValue(epoch, 4));
}

void Profiler::ReportEpochAverage(uint64_t epoch) const {
  double s[kNumLogEpochItems] = {0};
  for (int i = 0; i < kNumLogEpochItems; ++i)
    for (uint64_t e = 0; e <= epoch && e < epoch_[i].size(); ++e) s[i] += epoch_[i][e];
  const double n = (double)(epoch + 1);
  printf("    [Epoch(average) E%lu]\n        sample %.4lf | get cache miss index %.4lf | send %.4lf | "
         "sample total %.4lf | copy %.4lf\n",
         (unsigned long)epoch, s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n);
}

void Profiler::TraceStep(uint64_t key, int item, uint64_t ts, bool begin) {
  if (item < 0 || item >= kNumTraceItems) return;
  if (begin) {
    traces_.push_back({key, item, ts, ts});
  } else {
    for (auto it = traces_.rbegin(); it != traces_.rend(); ++it)
      if (it->key == key && it->item == item) { it->end = ts; break; }
  }
}

void Profiler::DumpTrace() const {
  // Chrome trace-event JSON (profiler.cc:286-364)
  fprintf(stderr, "[\n");
  for (size_t i = 0; i < traces_.size(); ++i)
    fprintf(stderr, "{\"name\":\"item%d\",\"cat\":\"samgraph\",\"ph\":\"X\",\"pid\":0,\"tid\":%d,\"ts\":%lu,\"dur\":%lu,"
                    "\"args\":{\"key\":%lu}}%s\n",
            traces_[i].item, traces_[i].item, (unsigned long)traces_[i].begin,
            (unsigned long)(traces_[i].end - traces_[i].begin), (unsigned long)traces_[i].key,
            i + 1 < traces_.size() ? "," : "");
  fprintf(stderr, "]\n");
}

}  // namespace sam
