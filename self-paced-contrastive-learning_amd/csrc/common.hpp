// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libspcl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/spcl_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 = 4 VGPRs (MFMA 16x16x32 operand)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef uint16_t bf16_t;

namespace spcl {

void set_error(const char* fmt, ...);

#define SPCL_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      spcl::set_error(__VA_ARGS__);      \
      return SPCL_EINVAL;                \
    }                                    \
  } while (0)

void func_lds_limit(const void* fn, int bytes, const char* what);  // hipFuncSetAttribute, result remembered
const char* take_attr_error();

#define SPCL_LAUNCH_CHECK(name)                                                 \
  do {                                                                          \
    if (const char* a_ = spcl::take_attr_error()) {                             \
      (void)hipGetLastError();                                                  \
      spcl::set_error("%s: %s", name, a_);                                      \
      return SPCL_ELAUNCH;                                                      \
    }                                                                           \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      spcl::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
      return SPCL_ELAUNCH;                                                      \
    }                                                                           \
  } while (0)

// ---- built-in kernel timer (spcl_profile_* of the C ABI): when enabled, every launch of the library is bracketed by
// HIP events on the launch stream and remembered with its kernel symbol and (where the caller states them) its
// algorithmic bytes / FLOPs.  Disabled (the default) it costs one predictable branch per launch.
extern bool g_prof_on;
void prof_begin(const void* kernel_fn, hipStream_t st);
void prof_end(hipStream_t st);
void prof_cost(double bytes, double flops);  // cost of the NEXT launch (ignored when profiling is off)
struct ProfScope {
  hipStream_t st;
  bool on;
  ProfScope(const void* fn, hipStream_t s) : st(s), on(g_prof_on) { if (on) prof_begin(fn, s); }
  ~ProfScope() { if (on) prof_end(st); }
};
#define SPCL_LAUNCH(kernel, grid, block, lds, st, ...)                   \
  do {                                                                   \
    spcl::ProfScope prof_scope_((const void*)(kernel), (st));            \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);       \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even via the hardware convert (keeps NaN a NaN, MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = SPCL_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = SPCL_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// butterfly sum over the 64 lanes of a wave (every lane gets the total; fixed order -> deterministic)
// (the four steps inside a 16-lane row by DPP moves -- quad swaps, half mirror, mirror: no trips through the LDS crossbar --,
// the two across rows by shuffles)
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
// the same inside one 16-lane row (every lane of the row gets the row's total): doubles move as two 32-bit halves
template <int CTRL> __device__ __forceinline__ double dpp_move_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double row16_sum_f64(double v) {
  v += dpp_move_f64<0xB1>(v);
  v += dpp_move_f64<0x4E>(v);
  v += dpp_move_f64<0x141>(v);
  v += dpp_move_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}

// one-shot capture of a weight gradient's pending final sum (spcl_wgrad_tail_capture): the producer that finds a slot
// fills it, skips its own reduction launch and clears the capture
spcl_wgrad_tail* take_tail_capture();

// wgrad.hip: final sum of a weight gradient's split partial slabs
void launch_wgrad_reduce(const float* partial, int nsplit, int nblk_ci, int nblk_co, int CIB, int COB, int Cin, int Cout,
                         float* dw_oihw, hipStream_t st);

// Tuning / ablation knobs read from the environment exist in LAB builds only (-DSPCL_LAB=1: tools/diag/mk_variant.sh,
// SPCL_BUILD_DEFS): the shipped library compiles every one of them to its default -- nothing in the process environment can
// change what a production kernel computes.
#ifndef SPCL_LAB
#define SPCL_LAB 0
#endif
inline int lab_env(const char* name, int dflt) {
#if SPCL_LAB
  const char* e = getenv(name);
  return e != nullptr ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

inline bool lab_flag(const char* name) {  // "is the variable set at all" (the switches that are on by their mere presence)
#if SPCL_LAB
  return getenv(name) != nullptr;
#else
  (void)name;
  return false;
#endif
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

}  // namespace spcl
