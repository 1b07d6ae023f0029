// Error plumbing shared by all launchers (see launch.h).
#include "launch.h"

#include <cstdarg>
#include <cstdio>
#include <vector>

namespace tfx {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

const char* last_error() { return g_err; }

// ---- launch-level profiler ------------------------------------------------------------------------------
namespace {
struct Rec { hipEvent_t a, b; double flops; };
struct Prof {
  bool on = false;
  std::vector<Rec> recs[3];   // 0 bf16 GEMM, 1 attention, 2 fp8 GEMM
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
} g_prof;
}  // namespace

void prof_enable(int on) { g_prof.on = on != 0; }
bool prof_on() { return g_prof.on; }
bool prof_on(hipStream_t st) {
  if (!g_prof.on) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;   // events recorded while capturing cannot be timed
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
  return cs == hipStreamCaptureStatusNone;
}
void prof_begin(int kind, double flops, hipStream_t st) {
  Rec r{g_prof.get(), g_prof.get(), flops};
  hipEventRecord(r.a, st);
  g_prof.recs[kind].push_back(r);
}
void prof_end(int kind, hipStream_t st) { hipEventRecord(g_prof.recs[kind].back().b, st); }
int prof_collect(int kind, double* total_ms, double* total_flops, int* launches) {
  double ms = 0, fl = 0;
  int n = 0;
  for (Rec& r : g_prof.recs[kind]) {
    if (hipEventSynchronize(r.b) != hipSuccess) return fail("prof_collect: event sync failed");
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return fail("prof_collect: elapsed time failed");
    ms += t; fl += r.flops; ++n;
    g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b);
  }
  g_prof.recs[kind].clear();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return 0;
}

}  // namespace tfx
