// Error reporting, ABI version and the built-in kernel timer of libspcl_hip.so (host only).
#include <cxxabi.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/spcl_hip.h"

namespace spcl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- hipFuncSetAttribute with its result kept: a kernel that needs more dynamic LDS than the default 64 KB asks for it once;
// if the runtime refuses, the launch behind it fails with an unspecific "invalid value".  The refusal is remembered here
// and SPCL_LAUNCH_CHECK reports IT (results were discarded).
static thread_local char g_attr_err[256] = "";
void func_lds_limit(const void* fn, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    snprintf(g_attr_err, sizeof(g_attr_err), "hipFuncSetAttribute(%s, max dynamic LDS = %d bytes): %s", what, bytes,
             hipGetErrorString(e));
    (void)hipGetLastError();
  }
}
const char* take_attr_error() {
  if (!g_attr_err[0]) return nullptr;
  static thread_local char out[256];
  memcpy(out, g_attr_err, sizeof(out));
  g_attr_err[0] = 0;
  return out;
}

// ---- one-shot weight-gradient tail capture (see spcl_wgrad_tail_capture in the header)
static thread_local spcl_wgrad_tail* g_tail_slot = nullptr;
spcl_wgrad_tail* take_tail_capture() {
  spcl_wgrad_tail* s = g_tail_slot;
  g_tail_slot = nullptr;
  return s;
}
void arm_tail_capture(spcl_wgrad_tail* slot) { g_tail_slot = slot; }

// ---- kernel timer
struct ProfRecord {
  const void* fn;
  hipStream_t st;
  hipEvent_t e0, e1;
  double bytes, flops;
};
bool g_prof_on = false;
static std::vector<ProfRecord> g_records;
static double g_next_bytes = 0.0, g_next_flops = 0.0;

void prof_cost(double bytes, double flops) {
  if (!g_prof_on) return;
  g_next_bytes = bytes;
  g_next_flops = flops;
}
void prof_begin(const void* fn, hipStream_t st) {
  ProfRecord r;
  r.fn = fn;
  r.st = st;
  r.bytes = g_next_bytes;
  r.flops = g_next_flops;
  g_next_bytes = g_next_flops = 0.0;
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, st);
  g_records.push_back(r);
}
void prof_end(hipStream_t st) {
  if (!g_records.empty()) (void)hipEventRecord(g_records.back().e1, st);
}
static void prof_clear() {
  for (auto& r : g_records) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_records.clear();
}
}  // namespace spcl

extern "C" int spcl_abi_version(void) { return SPCL_ABI_VERSION; }
extern "C" const char* spcl_last_error(void) { return spcl::g_err; }

extern "C" int spcl_profile_enable(int on) {
  spcl::prof_clear();
  spcl::g_prof_on = on != 0;
  return SPCL_OK;
}
extern "C" int spcl_profile_count(void) { return (int)spcl::g_records.size(); }
extern "C" int spcl_profile_get(int i, char* name, int name_cap, float* usec, double* bytes, double* flops) {
  if (i < 0 || i >= (int)spcl::g_records.size() || !name || name_cap < 2 || !usec || !bytes || !flops) {
    spcl::set_error("profile_get: bad index / null pointer");
    return SPCL_EINVAL;
  }
  const spcl::ProfRecord& r = spcl::g_records[i];
  if (hipEventSynchronize(r.e1) != hipSuccess) {
    spcl::set_error("profile_get: event not recorded (profiling inside a graph capture?)");
    return SPCL_ELAUNCH;
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, r.e0, r.e1);
  *usec = ms * 1e3f;
  *bytes = r.bytes;
  *flops = r.flops;
  const char* mangled = hipKernelNameRefByPtr(r.fn, r.st);
  std::string nm = mangled ? mangled : "?";
  if (mangled) {
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    if (status == 0 && dem) {
      nm = dem;
      const size_t paren = nm.rfind('(');  // drop the argument list and the leading "void "
      if (paren != std::string::npos) nm.resize(paren);
      if (nm.rfind("void ", 0) == 0) nm.erase(0, 5);
    }
    free(dem);
  }
  snprintf(name, (size_t)name_cap, "%s", nm.c_str());
  return SPCL_OK;
}

extern "C" int spcl_wgrad_tail_capture(spcl_wgrad_tail* slot) {
  if (slot) {
    memset(slot, 0, sizeof(*slot));
    slot->kind = -1;
  }
  spcl::arm_tail_capture(slot);
  return SPCL_OK;
}
