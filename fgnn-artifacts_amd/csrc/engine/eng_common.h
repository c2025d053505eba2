// eng_common.h -- logging / CHECK (reference logging.h:32-77: a failed CHECK prints file:line and
// abort()s), device contexts (common.cc:45-64), wall-clock timer (timer.h).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "fgnn_hip.h"

namespace sam {

enum LogLevel { kTrace = 0, kDebug, kInfo, kWarning, kError, kFatal };
int MinLogLevel();  // SAMGRAPH_LOG_LEVEL (logging.cc:73), default warning

class LogMessage {
 public:
  LogMessage(const char *file, int line, int level, bool fatal) : level_(level), fatal_(fatal) {
    static const char *names[] = {"TRACE", "DEBUG", "INFO", "WARNING", "ERROR", "FATAL"};
    ss_ << "[" << names[level] << "] " << file << ":" << line << ": ";
  }
  ~LogMessage() {
    if (fatal_ || level_ >= MinLogLevel()) {
      ss_ << "\n";
      fputs(ss_.str().c_str(), stderr);
      fflush(stderr);
    }
    if (fatal_) abort();
  }
  std::ostream &stream() { return ss_; }

 private:
  std::ostringstream ss_;
  int level_;
  bool fatal_;
};

#define SAM_LOG(level) ::sam::LogMessage(__FILE__, __LINE__, ::sam::level, false).stream()
#define SAM_FATAL ::sam::LogMessage(__FILE__, __LINE__, ::sam::kFatal, true).stream()
#define SAM_CHECK(cond) \
  if (!(cond)) ::sam::LogMessage(__FILE__, __LINE__, ::sam::kFatal, true).stream() << "Check failed: " #cond " "
#define SAM_CHECK_EQ(a, b) SAM_CHECK((a) == (b)) << "(" << (a) << " vs " << (b) << ") "
#define SAM_CHECK_LE(a, b) SAM_CHECK((a) <= (b)) << "(" << (a) << " vs " << (b) << ") "
#define SAM_CHECK_LT(a, b) SAM_CHECK((a) < (b)) << "(" << (a) << " vs " << (b) << ") "
#define SAM_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) SAM_FATAL << "HIP error: " #expr ": " << hipGetErrorString(_e);          \
  } while (0)
#define SAM_FGNN(expr)                                                                             \
  do {                                                                                             \
    int _rc = (expr);                                                                              \
    if (_rc != FGNN_OK) SAM_FATAL << #expr " failed with " << _rc << " " << fgnn_last_error();     \
  } while (0)

enum DeviceType { kCPU = 0, kMMAP = 1, kGPU = 2 };  // common.h:48

struct Context {
  int device_type = kCPU;
  int device_id = 0;
  Context() = default;
  Context(int t, int i) : device_type(t), device_id(i) {}
  explicit Context(const std::string &name);  // "cpu:0" | "cuda:1" | "mmap:0"
  bool IsGPU() const { return device_type == kGPU; }
  std::string Str() const;
};

class Timer {
 public:
  Timer() : t0_(std::chrono::steady_clock::now()) {}
  double Passed() const {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
  }
  static uint64_t NowMicro() {
    return std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
  }

 private:
  std::chrono::steady_clock::time_point t0_;
};

inline size_t RoundUpDiv(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace sam
