// Fused RAdam step on ONE flat fp32 parameter (the optimizer step of the pre-train iteration,
// contrastyou/trainer/base.py:62 -> RAdam; the build follows torch.optim.RAdam).
// torch's foreach implementation is ~40 elementwise launches over the 1.3 M-element flat parameter (~270 us per
// step); this is one 1-thread "tick" (step counter + the scalar coefficients, in double) and one streaming kernel
// (p, g, m, v read once, p, m, v written once: 28 bytes per element).
#include <string.h>
#include "common.hpp"

namespace spcl {

struct ScalarAdds {
  const float* src[8];
  float* dst[8];
  float count[8];
  int k;
};

// beta^t by repeated squaring: IEEE multiplications only, so that the host (optim.py radam_coefficients, the staged steps)
// and this kernel produce the same bits -- a replayed step and an eager one then update with identical coefficients
__device__ __forceinline__ double ipow(double b, int64_t e) {
#pragma clang fp contract(off)
  double r = 1.0;
  while (e > 0) {
    if (e & 1) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}

// coef[0] = lr / (1 - beta1^t);  coef[1] = rect * sqrt(1 - beta2^t) when rho_t > 5 else 0;  coef[2] = rho_t > 5
// Threads 1 .. k of the same (one-wave) launch perform the step's meter updates (spcl_radam_step_scalars): the running
// means of the host-side meters are one more few-microsecond launch per step otherwise.
__global__ __launch_bounds__(64) void radam_tick_kernel(int64_t* step, const float* lr, double beta1, double beta2,
                                                        float* coef, ScalarAdds a) {
#pragma clang fp contract(off)  // (IEEE operations one by one, as the host computes them: optim.py radam_coefficients)
  if (threadIdx.x > 0) {
    const int i = threadIdx.x - 1;
    if (i < a.k) {
      a.dst[i][0] = fmaf(a.count[i], a.src[i][0], a.dst[i][0]);
      a.dst[i][1] += a.count[i];
    }
    return;
  }
  const int64_t t = step[0] + 1;
  step[0] = t;
  const double b1t = ipow(beta1, t), b2t = ipow(beta2, t);
  const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
  const double rho_inf = 2.0 / (1.0 - beta2) - 1.0;
  const double rho_t = rho_inf - 2.0 * (double)t * b2t / bc2;
  coef[0] = (float)((double)lr[0] / bc1);
  if (rho_t > 5.0) {
    const double rect = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
    coef[1] = (float)(rect * sqrt(bc2));
    coef[2] = 1.f;
  } else {
    coef[1] = 0.f;
    coef[2] = 0.f;
  }
}

// STAGED (spcl_radam_apply_staged): the coefficients were computed on the host and travel with the step's staged bytes
// (coef[3] = the step count t they belong to): no coefficient launch; workgroup 0 records t in the device counter and its
// threads 1 .. k perform the meter updates.
template <bool STAGED>
__global__ __launch_bounds__(256) void radam_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                          size_t n, const float* __restrict__ coef, float omb1,
                                                          float beta2, float omb2, float eps, float wd, float gscale,
                                                          int64_t* step, ScalarAdds a) {
  const float c_m = coef[0], c_u = coef[1];
  const bool rect = coef[2] != 0.f;
  if (STAGED && blockIdx.x == 0) {
    if (threadIdx.x == 0) step[0] = (int64_t)coef[3];
    else if ((int)threadIdx.x <= a.k) {
      const int j = threadIdx.x - 1;
      a.dst[j][0] = fmaf(a.count[j], a.src[j][0], a.dst[j][0]);
      a.dst[j][1] += a.count[j];
    }
  }
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg = fmaf(wd, pp, gg * gscale);            // gscale: 1 / world of the data-parallel mean (1.f: exact identity)
    mm = fmaf(omb1, gg - mm, mm);              // lerp_(grad, 1 - beta1)
    vv = fmaf(vv, beta2, omb2 * gg * gg);      // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float u = rect ? c_u / (sqrtf(vv) + eps) : 1.f;
    pp = fmaf(-c_m * mm, u, pp);
  };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      upd(pe, gg[e], me, ve);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    ((f32x4*)p)[i] = pp;
    ((f32x4*)m)[i] = mm;
    ((f32x4*)v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {  // tail of n % 4 elements
    const size_t i = 4 * n4 + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}

}  // namespace spcl

using namespace spcl;

// running means of the host-side meters: dst = [sum, count];  sum += count_i * src,  count += count_i
__global__ void accumulate_scalars_kernel(ScalarAdds a) {
  const int i = threadIdx.x;
  if (i < a.k) {
    a.dst[i][0] = fmaf(a.count[i], a.src[i][0], a.dst[i][0]);
    a.dst[i][1] += a.count[i];
  }
}

static int fill_scalar_adds(ScalarAdds& a, int k, const void* const* src, void* const* dst, const float* count,
                            const char* who) {
  a.k = k;
  for (int i = 0; i < k; ++i) {
    SPCL_CHECK_ARG(src[i] && dst[i], "%s: null pointer", who);
    for (int j = 0; j < i; ++j) SPCL_CHECK_ARG(dst[j] != dst[i], "%s: a destination appears twice", who);
    a.src[i] = (const float*)src[i];
    a.dst[i] = (float*)dst[i];
    a.count[i] = count[i];
  }
  return SPCL_OK;
}

extern "C" int spcl_accumulate_scalars(int k, const void* const* src, void* const* dst, const float* count,
                                       void* stream) {
  SPCL_CHECK_ARG(k >= 1 && k <= 8 && src && dst && count, "accumulate_scalars: 1 <= k <= 8 pairs per call");
  ScalarAdds a;
  if (int rc = fill_scalar_adds(a, k, src, dst, count, "accumulate_scalars")) return rc;
  SPCL_LAUNCH(accumulate_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  SPCL_LAUNCH_CHECK("accumulate_scalars");
  return SPCL_OK;
}

// Per-step host inputs of a captured step (label vectors, flip flags, scalars): the bytes travel AS KERNEL ARGUMENTS
// (copied at launch time: no pinned host buffer whose lifetime the caller would have to guard, no DMA engine) and one
// tiny launch writes them to their persistent device block, in stream order ahead of the replay that reads them.
constexpr int STAGE_WORDS = 896;  // 3 584 bytes per launch (kernel arguments are limited to 4 KB)
struct StageWords {
  uint32_t w[STAGE_WORDS];
};
__global__ __launch_bounds__(256) void stage_bytes_kernel(uint32_t* __restrict__ dst, int nwords, StageWords s) {
  for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = s.w[i];
}

extern "C" int spcl_stage_bytes(void* dst, const void* host_src, size_t nbytes, void* stream) {
  SPCL_CHECK_ARG(dst && host_src, "stage_bytes: null pointer");
  SPCL_CHECK_ARG(nbytes % 4 == 0 && (uintptr_t)dst % 4 == 0, "stage_bytes: size and destination must be multiples of 4");
  const uint32_t* src = (const uint32_t*)host_src;
  uint32_t* d = (uint32_t*)dst;
  size_t left = nbytes / 4;
  while (left > 0) {
    const int n = left > (size_t)STAGE_WORDS ? STAGE_WORDS : (int)left;
    StageWords s;
    memcpy(s.w, src, (size_t)n * 4);
    SPCL_LAUNCH(stage_bytes_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d, n, s);
    src += n; d += n; left -= n;
  }
  SPCL_LAUNCH_CHECK("stage_bytes");
  return SPCL_OK;
}

// Two device-to-device copies in ONE launch (the fine-tune step's image and label map into the captured step's persistent
// input buffers: two copy launches of ~5 us each otherwise, whatever their size).  16-byte granules, grid-stride.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ __launch_bounds__(256) void copy_pair_kernel(u32x4* __restrict__ da, const u32x4* __restrict__ sa, size_t na,
                                                        u32x4* __restrict__ db, const u32x4* __restrict__ sb, size_t nb) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) da[i] = sa[i];
    else db[i - na] = sb[i - na];
  }
}

extern "C" int spcl_copy_pair(void* dst_a, const void* src_a, size_t bytes_a, void* dst_b, const void* src_b, size_t bytes_b,
                              void* stream) {
  SPCL_CHECK_ARG(dst_a && src_a && dst_b && src_b, "copy_pair: null pointer");
  SPCL_CHECK_ARG(bytes_a % 16 == 0 && bytes_b % 16 == 0 &&
                     ((uintptr_t)dst_a | (uintptr_t)src_a | (uintptr_t)dst_b | (uintptr_t)src_b) % 16 == 0,
                 "copy_pair: sizes and addresses must be multiples of 16 bytes");
  const size_t n = (bytes_a + bytes_b) / 16;
  if (n == 0) return SPCL_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  SPCL_LAUNCH(copy_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)dst_a, (const u32x4*)src_a,
              bytes_a / 16, (u32x4*)dst_b, (const u32x4*)src_b, bytes_b / 16);
  SPCL_LAUNCH_CHECK("copy_pair");
  return SPCL_OK;
}

extern "C" int spcl_radam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                               int64_t* step, const float* lr, double beta1, double beta2, double eps,
                               double weight_decay, float* coef, void* stream) {
  return spcl_radam_step_scalars(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, coef, 0,
                                 nullptr, nullptr, nullptr, stream);
}

extern "C" int spcl_radam_step_scalars(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                       int64_t* step, const float* lr, double beta1, double beta2, double eps,
                                       double weight_decay, float* coef, int k, const void* const* src,
                                       void* const* dst, const float* count, void* stream) {
  return spcl_radam_step_scaled(param, grad, 1.0, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, coef,
                                k, src, dst, count, stream);
}

extern "C" int spcl_radam_step_scaled(float* param, const float* grad, double grad_scale, float* exp_avg,
                                      float* exp_avg_sq, size_t n, int64_t* step, const float* lr, double beta1,
                                      double beta2, double eps, double weight_decay, float* coef, int k,
                                      const void* const* src, void* const* dst, const float* count, void* stream) {
  SPCL_CHECK_ARG(grad_scale > 0.0 && grad_scale <= 1.0, "radam_step: grad_scale in (0, 1]");
  SPCL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step && lr && coef, "radam_step: null pointer");
  SPCL_CHECK_ARG(k >= 0 && k <= 8 && (k == 0 || (src && dst && count)), "radam_step: 0 <= k <= 8 scalar adds");
  ScalarAdds adds;
  adds.k = 0;
  if (k > 0)
    if (int rc = fill_scalar_adds(adds, k, src, dst, count, "radam_step")) return rc;
  SPCL_CHECK_ARG(n > 0, "radam_step: empty parameter");
  SPCL_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                 "radam_step: buffers must be 16-byte aligned");
  SPCL_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "radam_step: betas");
  hipStream_t st = (hipStream_t)stream;
  SPCL_LAUNCH(radam_tick_kernel, dim3(1), dim3(k > 0 ? 64 : 1), 0, st, step, lr, beta1, beta2, coef, adds);
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  SPCL_LAUNCH(radam_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n4,
                     n, (const float*)coef, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)grad_scale, (int64_t*)nullptr, adds);
  SPCL_LAUNCH_CHECK("radam_step");
  return SPCL_OK;
}

extern "C" int spcl_radam_apply_staged(float* param, const float* grad, double grad_scale, float* exp_avg,
                                       float* exp_avg_sq, size_t n, int64_t* step, const float* coef, double beta1,
                                       double beta2, double eps, double weight_decay, int k, const void* const* src,
                                       void* const* dst, const float* count, void* stream) {
  SPCL_CHECK_ARG(grad_scale > 0.0 && grad_scale <= 1.0, "radam_apply_staged: grad_scale in (0, 1]");
  SPCL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step && coef, "radam_apply_staged: null pointer");
  SPCL_CHECK_ARG(k >= 0 && k <= 8 && (k == 0 || (src && dst && count)), "radam_apply_staged: 0 <= k <= 8 scalar adds");
  ScalarAdds adds;
  adds.k = 0;
  if (k > 0)
    if (int rc = fill_scalar_adds(adds, k, src, dst, count, "radam_apply_staged")) return rc;
  SPCL_CHECK_ARG(n > 0, "radam_apply_staged: empty parameter");
  SPCL_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)coef) % 16 == 0,
                 "radam_apply_staged: buffers must be 16-byte aligned");
  SPCL_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "radam_apply_staged: betas");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  SPCL_LAUNCH(radam_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
              exp_avg_sq, n4, n, coef, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
              (float)weight_decay, (float)grad_scale, step, adds);
  SPCL_LAUNCH_CHECK("radam_apply_staged");
  return SPCL_OK;
}
