// Projector head for gfx950: global average pool -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise, fwd + bwd.
// Replaces contrastyou/projectors/heads.py:9-25,78-92 and nn.py:8-15,29-36,56-58 (K8-K10 + their backward).
// Sizes are tiny (N=64 rows, 256x256 weights): every kernel is launch/latency-bound, so they are plain
// coalesced FMA kernels with fixed (deterministic) reduction order; fp32 throughout.
#include "common.hpp"

namespace spcl {

constexpr float kLeaky = 0.01f;

// pooled[n][c] = mean_hw feat[n][hw][c].  Workgroup = (image, 64 channels): lanes = channels (coalesced), the 4 waves
// take interleaved pixels and are combined through LDS in fixed order.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ feat, int HW, int C, int Cs,
                                                      float* __restrict__ pooled) {
  __shared__ float red[4][64];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
    const T* p = feat + (size_t)n * HW * Cs + c;
#pragma unroll 8
    for (int i = wave; i < HW; i += 4) s += Elem<T>::load(p + (size_t)i * Cs);
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C)
    pooled[(size_t)n * C + c] = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) / (float)HW;
}

// y[n][o] = sum_k act(x[n][k]) * W[o][k] + b[o];  one wave per output column o; its weight row lives in registers and
// the rows n are processed RB at a time with all their loads issued together (a row-at-a-time loop was a chain of
// eight ~1 us global-load latencies: 12.5 us for a 64x256x256 product).  act = leaky (x is a saved pre-activation)
// when LEAKY_IN.
template <bool LEAKY_IN>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ b, int N, int K, int O,
                                                         float* __restrict__ y) {
  constexpr int RB = 8, KMAX = 8;  // K <= 512
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= O) return;
  const float bias = b[o];
  float wr[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) wr[i] = lane + 64 * i < K ? W[(size_t)o * K + lane + 64 * i] : 0.f;
  for (int n0 = blockIdx.y * RB; n0 < N; n0 += gridDim.y * RB) {
    float xv[RB][KMAX];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        xv[r][i] = (n0 + r < N && lane + 64 * i < K) ? x[(size_t)(n0 + r) * K + lane + 64 * i] : 0.f;
    float s[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        float v = xv[r][i];
        if (LEAKY_IN) v = v > 0.f ? v : kLeaky * v;
        s[r] = fmaf(v, wr[i], s[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) s[r] = wave_sum(s[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (n0 + r < N) y[(size_t)(n0 + r) * O + o] = s[r] + bias;
    }
  }
}

// z = o / max(||o||, 1e-12)   (F.normalize p=2 dim=1)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ o, int N, int O,
                                                         float* __restrict__ z) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < O; k += 64) {
    float v = o[(size_t)n * O + k];
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int k = lane; k < O; k += 64) z[(size_t)n * O + k] = o[(size_t)n * O + k] * inv;
}

// do = (dz - z (z.dz)) / max(||o||,eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ o, const float* __restrict__ dz,
                                                         int N, int O, float* __restrict__ d_o) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f, dot = 0.f;
  for (int k = lane; k < O; k += 64) {
    float v = o[(size_t)n * O + k];
    s = fmaf(v, v, s);
    dot = fmaf(v, dz[(size_t)n * O + k], dot);
  }
  s = wave_sum(s);
  dot = wave_sum(dot);
  const float nrm = sqrtf(s);
  if (nrm > 1e-12f) {
    const float inv = 1.f / nrm;
    const float zd = dot * inv;  // z . dz
    for (int k = lane; k < O; k += 64) {
      float zk = o[(size_t)n * O + k] * inv;
      d_o[(size_t)n * O + k] = (dz[(size_t)n * O + k] - zk * zd) * inv;
    }
  } else {  // clamp branch of F.normalize: denominator is the constant eps
    for (int k = lane; k < O; k += 64) d_o[(size_t)n * O + k] = dz[(size_t)n * O + k] * 1e12f;
  }
}

// dW[o][k] = sum_n g[n][o] * act(x[n][k]);  db[o] = sum_n g[n][o]      (thread per (o,k), n sequential)
template <bool LEAKY_IN>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           int N, int K, int O, float* __restrict__ dW,
                                                           float* __restrict__ db) {
  const int o = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f, sb = 0.f;
#pragma unroll 8
  for (int n = 0; n < N; ++n) {
    const float gv = g[(size_t)n * O + o];
    float xv = x[(size_t)n * K + k];
    if (LEAKY_IN) xv = xv > 0.f ? xv : kLeaky * xv;
    s = fmaf(gv, xv, s);
    sb += gv;
  }
  dW[(size_t)o * K + k] = s;
  if (k == 0) db[o] = sb;
}

// dx[n][k] = (sum_o g[n][o] W[o][k]) * (LEAKY_OUT ? leaky'(pre[n][k]) : 1).  Workgroup = (row n, 64 columns k): the
// 4 waves take interleaved o and are combined through LDS in fixed order.
template <bool LEAKY_OUT>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                           const float* __restrict__ pre, int N, int K, int O,
                                                           float* __restrict__ dx) {
  __shared__ float red[4][64];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (k < K) {
#pragma unroll 8
    for (int o = wave; o < O; o += 4) s = fmaf(g[(size_t)n * O + o], W[(size_t)o * K + k], s);
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && k < K) {
    float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (LEAKY_OUT) v *= pre[(size_t)n * K + k] > 0.f ? 1.f : kLeaky;
    dx[(size_t)n * K + k] = v;
  }
}

// dfeat[n][hw][c] = dpooled[n][c] / HW  (0 in the channel padding)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dpooled, int HW, int C, int Cs,
                                                          T* __restrict__ dfeat, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % Cs);
  const size_t n = idx / ((size_t)HW * Cs);
  const float v = c < C ? dpooled[n * C + c] / (float)HW : 0.f;
  Elem<T>::store(dfeat + idx, v);
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_proj_forward(const void* feat, int dtype, int N, int HW, int C, int Cs, const float* w1,
                                 const float* b1, const float* w2, const float* b2, int hid, int out_dim,
                                 int normalize, float* pooled, float* pre, float* o, float* z, void* stream) {
  SPCL_CHECK_ARG(feat && w1 && b1 && pooled && o && z, "proj_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0, "proj_forward: bad shape");
  SPCL_CHECK_ARG(hid == 0 || (w2 && b2 && pre), "proj_forward: mlp head needs w2/b2/pre");
  SPCL_CHECK_ARG(C <= 512 && hid <= 512, "proj_forward: at most 512 input / hidden features (C=%d, hid=%d)", C, hid);
  hipStream_t st = (hipStream_t)stream;
  dim3 pg(cdiv(C, 64), N);
  if (dtype == SPCL_F32)
    SPCL_LAUNCH(avgpool_kernel<float>, pg, dim3(256), 0, st, (const float*)feat, HW, C, Cs, pooled);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH(avgpool_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t*)feat, HW, C, Cs, pooled);
  else {
    set_error("proj_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  const int ny = (N + 7) / 8 < 8 ? (N + 7) / 8 : 8;  // 8 rows per block iteration
  if (hid > 0) {
    SPCL_LAUNCH(linear_fwd_kernel<false>, dim3(cdiv(hid, 4), ny), dim3(256), 0, st, (const float*)pooled, w1,
                       b1, N, C, hid, pre);
    SPCL_LAUNCH(linear_fwd_kernel<true>, dim3(cdiv(out_dim, 4), ny), dim3(256), 0, st, (const float*)pre, w2,
                       b2, N, hid, out_dim, o);
  } else {
    SPCL_LAUNCH(linear_fwd_kernel<false>, dim3(cdiv(out_dim, 4), ny), dim3(256), 0, st, (const float*)pooled,
                       w1, b1, N, C, out_dim, o);
  }
  if (normalize)
    SPCL_LAUNCH(l2norm_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, (const float*)o, N, out_dim, z);
  else
    (void)hipMemcpyAsync(z, o, (size_t)N * out_dim * sizeof(float), hipMemcpyDeviceToDevice, st);
  SPCL_LAUNCH_CHECK("proj_forward");
  return SPCL_OK;
}

extern "C" int spcl_proj_backward(const float* dz, int dtype, int N, int HW, int C, int Cs, const float* w1,
                                  const float* w2, int hid, int out_dim, int normalize, const float* pooled,
                                  const float* pre, const float* o, float* dw1, float* db1, float* dw2, float* db2,
                                  float* scratch, void* dfeat, void* stream) {
  SPCL_CHECK_ARG(dz && w1 && pooled && o && dw1 && db1 && scratch, "proj_backward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0, "proj_backward: bad shape");
  SPCL_CHECK_ARG(hid == 0 || (w2 && pre && dw2 && db2), "proj_backward: mlp head needs w2/pre/dw2/db2");
  hipStream_t st = (hipStream_t)stream;
  float* d_o = scratch;                                 // [N,out]
  float* dpre = scratch + (size_t)N * out_dim;          // [N,hid]
  float* dpool = dpre + (size_t)N * (hid > 0 ? hid : 0);  // [N,C]
  const float* go = dz;
  if (normalize) {
    SPCL_LAUNCH(l2norm_bwd_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, o, dz, N, out_dim, d_o);
    go = d_o;
  }
  if (hid > 0) {
    SPCL_LAUNCH(linear_wgrad_kernel<true>, dim3(cdiv(hid, 256), out_dim), dim3(256), 0, st, go, pre, N, hid,
                       out_dim, dw2, db2);
    SPCL_LAUNCH(linear_dgrad_kernel<true>, dim3(cdiv(hid, 64), N), dim3(256), 0, st, go, w2, pre, N, hid,
                       out_dim, dpre);
    SPCL_LAUNCH(linear_wgrad_kernel<false>, dim3(cdiv(C, 256), hid), dim3(256), 0, st, (const float*)dpre,
                       pooled, N, C, hid, dw1, db1);
    if (dfeat)
      SPCL_LAUNCH(linear_dgrad_kernel<false>, dim3(cdiv(C, 64), N), dim3(256), 0, st, (const float*)dpre, w1,
                         (const float*)nullptr, N, C, hid, dpool);
  } else {
    SPCL_LAUNCH(linear_wgrad_kernel<false>, dim3(cdiv(C, 256), out_dim), dim3(256), 0, st, go, pooled, N, C,
                       out_dim, dw1, db1);
    if (dfeat)
      SPCL_LAUNCH(linear_dgrad_kernel<false>, dim3(cdiv(C, 64), N), dim3(256), 0, st, go, w1,
                         (const float*)nullptr, N, C, out_dim, dpool);
  }
  if (dfeat) {
    const size_t total = (size_t)N * HW * Cs;
    dim3 g((unsigned)((total + 255) / 256));
    if (dtype == SPCL_F32)
      SPCL_LAUNCH(avgpool_bwd_kernel<float>, g, dim3(256), 0, st, (const float*)dpool, HW, C, Cs,
                         (float*)dfeat, total);
    else if (dtype == SPCL_BF16)
      SPCL_LAUNCH(avgpool_bwd_kernel<bf16_t>, g, dim3(256), 0, st, (const float*)dpool, HW, C, Cs,
                         (bf16_t*)dfeat, total);
    else {
      set_error("proj_backward: dtype %d", dtype);
      return SPCL_EINVAL;
    }
  }
  SPCL_LAUNCH_CHECK("proj_backward");
  return SPCL_OK;
}
