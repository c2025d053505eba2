// capi.hip -- small entry points of libfgnn_hip.so that are not tied to one kernel file.
#include "fgnn_device.h"

#include <cstdio>

namespace fgnn {
static thread_local char g_last_error[512] = "";
void set_last_error(const char *what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}
uint32_t *&scan_error_sink() {
  static thread_local uint32_t *sink = nullptr;
  return sink;
}
static unsigned long long *g_phase_log = nullptr;
unsigned long long *phase_log_base() { return g_phase_log; }
}  // namespace fgnn

extern "C" size_t fgnn_debug_phase_log_bytes(void) {
  return (size_t)fgnn::kPhaseLogKinds * fgnn::kPhaseLogTiles * 8 * sizeof(unsigned long long);
}
extern "C" void fgnn_debug_phase_log(unsigned long long *d_buf) { fgnn::g_phase_log = d_buf; }

extern "C" const char *fgnn_last_error(void) { return fgnn::g_last_error; }

extern "C" const char *fgnn_version(void) { return "fgnn-hip 0.1 (gfx950)"; }

extern "C" int fgnn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Largest layout any call needs: one uint32 per item + one per workgroup + slack.
extern "C" size_t fgnn_scratch_bytes(size_t n_cap) {
  return (n_cap + fgnn::div_up(n_cap, 64) + 64) * sizeof(uint32_t);  // items + one sum per >=64-item workgroup
}
