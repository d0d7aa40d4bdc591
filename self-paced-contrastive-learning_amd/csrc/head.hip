// Segmentation head and fine-tune / evaluation arithmetic on gfx950:
//   * 1x1 convolution with bias, semi_seg/arch/unet.py:147 `_Deconv_1x1` (forward :229) + its backward;
//   * softmax over classes (`logits.softmax(1)`, semi_seg/epochers/new_epocher.py:86,271) + backward;
//   * deepclustering2.loss.KL_div(reduction="mean") as called there (restated: mean over positions of
//     sum_c -t log((p+eps)/(t+eps))) + backward w.r.t. the probabilities;
//   * class2one_hot (:84,270), arg-max over classes (`.max(1)[1]`, :89,282) and the per-sample per-class intersection /
//     union counts of contrastyou/meters/general_dice_meter.py:131-160.
// All of them are one-pixel-per-thread HBM streams over NHWC data: activations [N,H,W,CS] (dtype), class maps
// [N,H,W,K] f32 (K <= 16), labels [N,H,W] int64.  Reductions are per-workgroup partials + a fixed-order second stage
// (float) or integer atomics (counts): deterministic.
#include "common.hpp"

namespace spcl {

constexpr int HEAD_MAX_K = 16;
constexpr int HEAD_MAX_C = 256;
constexpr int HEAD_RED_WG = 1024;

// ------------------------------------------------------------------------------------------------ 1x1 conv + bias
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// 16-byte groups of a pixel's channels (CS is a multiple of 16, so a pixel is a whole number of groups)
template <typename T> struct VecIO;
template <> struct VecIO<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
    const u32x4 r = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(r[i] << 16);
      v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    *(u32x4*)p = r;
  }
};
template <> struct VecIO<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* v) {
    const f32x4 r = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = r[i];
  }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};

// KB = compile-time bound of the class count (4 / 8 / 16) so that the per-class accumulators live in registers.
// The weights sit in LDS as [KB][CS] with zeros in the channel padding and the unused classes: no bounds tests inside.
// IN_BN: x is the RAW output of the last 3x3 convolution and the input of the 1x1 convolution is relu(scale x + shift),
// rounded to T as the activation writer would have stored it (bnrelu_fwd_lin_kernel: the same FMA, the same conversion) --
// the decoder's last BatchNorm + ReLU without a tensor of its own (spcl_conv1x1_forward_bn).
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T, int KB, bool IN_BN = false>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const T* __restrict__ x, size_t npix, int C, int CS, int K,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out,
                                                          const float* __restrict__ in_scale = nullptr,
                                                          const float* __restrict__ in_shift = nullptr) {
  constexpr int VN = VecIO<T>::N;
  __shared__ float ws[KB * HEAD_MAX_C + KB];
  __shared__ float bnc[IN_BN ? 2 * HEAD_MAX_C : 1];
  for (int i = threadIdx.x; i < KB * CS; i += 256) {
    const int k = i / CS, c = i - k * CS;
    ws[i] = (k < K && c < C) ? w[k * C + c] : 0.f;
  }
  if (threadIdx.x < KB) ws[KB * CS + threadIdx.x] = threadIdx.x < K ? b[threadIdx.x] : 0.f;
  if (IN_BN)
    for (int c = threadIdx.x; c < CS; c += 256) { bnc[c] = in_scale[c]; bnc[HEAD_MAX_C + c] = in_shift[c]; }
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    float acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = ws[KB * CS + k];
    const T* px = x + p * CS;
    for (int c0 = 0; c0 < CS; c0 += VN) {
      float xv[VN];
      VecIO<T>::load(px + c0, xv);
      if (IN_BN) {
#pragma unroll
        for (int e = 0; e < VN; ++e) xv[e] = round_as<T>(fmaxf(fmaf(bnc[c0 + e], xv[e], bnc[HEAD_MAX_C + c0 + e]), 0.f));
      }
#pragma unroll
      for (int e = 0; e < VN; ++e)
#pragma unroll
        for (int k = 0; k < KB; ++k) acc[k] = fmaf(xv[e], ws[k * CS + c0 + e], acc[k]);
    }
    if (K == KB && KB % 4 == 0) {
#pragma unroll
      for (int k = 0; k < KB; k += 4) *(f32x4*)(out + p * K + k) = (f32x4){acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (k < K) out[p * K + k] = acc[k];
    }
  }
}

// dX[p][c] = sum_k dO[p][k] w[k][c]; per-workgroup partials of dW[k][c] = sum_p dO[p][k] x[p][c], db[k] = sum_p dO[p][k].
// Channels are walked in groups of CG = 64 / KB so that the KB x CG accumulators stay in registers; a pixel's group is
// read and written as 16-byte vectors.
// IN_BN (spcl_conv1x1_backward_bn): x is the raw 3x3-convolution output y; the 1x1 convolution's input is recomputed as in
// the forward, dx is the gradient w.r.t. that ACTIVATION, and the workgroup also leaves the partial sums of the BatchNorm's
// backward -- rows[wg][0][c] = sum dz, rows[wg][1][c] = sum dz (y - mean), dz = dx (as stored) where scale y + shift > 0 --
// in the form the convolution kernels' dgrad epilogues leave them (bnrelu_bwd_reduce_lin_kernel's arithmetic per pixel).
template <typename T, int KB, bool IN_BN = false>
__global__ __launch_bounds__(256) void conv1x1_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dout,
                                                          size_t npix, int C, int CS, int K,
                                                          const float* __restrict__ w, T* __restrict__ dx,
                                                          float* __restrict__ partial /* [grid][K][C+1] */,
                                                          const float* __restrict__ in_scale = nullptr,
                                                          const float* __restrict__ in_shift = nullptr,
                                                          const float* __restrict__ in_mean = nullptr,
                                                          float* __restrict__ rows = nullptr /* [grid][2][CS] */) {
  constexpr int CG = 64 / KB, VN = VecIO<T>::N;
  static_assert(CG % 4 == 0, "channel group");
  __shared__ float ws[KB * HEAD_MAX_C];
  __shared__ float red[4][KB * (CG + 1)];
  __shared__ float red2[IN_BN ? 4 : 1][2 * CG];
  for (int i = threadIdx.x; i < KB * CS; i += 256) {
    const int k = i / CS, c = i - k * CS;
    ws[i] = (k < K && c < C) ? w[k * C + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mypart = partial + (size_t)blockIdx.x * K * (C + 1);
  for (int c0 = 0; c0 < CS; c0 += CG) {
    float aw[KB][CG], ab[KB];
    float s1[IN_BN ? CG : 1], s2[IN_BN ? CG : 1];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      ab[k] = 0.f;
#pragma unroll
      for (int c = 0; c < CG; ++c) aw[k][c] = 0.f;
    }
    if (IN_BN) {
#pragma unroll
      for (int c = 0; c < CG; ++c) s1[c] = s2[c] = 0.f;
    }
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
      float g[KB];
      if (K == KB && KB % 4 == 0) {
#pragma unroll
        for (int k = 0; k < KB; k += 4) {
          const f32x4 r = *(const f32x4*)(dout + p * K + k);
          g[k] = r[0]; g[k + 1] = r[1]; g[k + 2] = r[2]; g[k + 3] = r[3];
        }
      } else {
#pragma unroll
        for (int k = 0; k < KB; ++k) g[k] = k < K ? dout[p * K + k] : 0.f;
      }
      float xv[CG], d[CG];
      if (CG >= VN) {
#pragma unroll
        for (int c = 0; c < CG; c += VN) VecIO<T>::load(x + p * CS + c0 + c, xv + c);
      } else {  // bf16 with 16 classes: 4 channels = half a vector
#pragma unroll
        for (int c = 0; c < CG; ++c) xv[c] = Elem<T>::load(x + p * CS + c0 + c);
      }
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        // (IN_BN: the coefficients are wave-uniform loads straight from global memory -- scalar registers; as LDS reads the
        // compiler hoisted all 48 into vector registers and the kernel fell to one wave per SIMD, 49 -> 101 us)
        float xin = xv[c], z = 0.f;
        if (IN_BN) {
          z = fmaf(in_scale[c0 + c], xv[c], in_shift[c0 + c]);
          xin = round_as<T>(fmaxf(z, 0.f));
        }
        float dd = 0.f;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          aw[k][c] = fmaf(g[k], xin, aw[k][c]);
          dd = fmaf(g[k], ws[k * CS + c0 + c], dd);  // zero weights in the channel padding: exact zeros there
        }
        d[c] = dd;
        if (IN_BN) {
          const float dz = z > 0.f ? round_as<T>(dd) : 0.f;
          s1[c] += dz;
          s2[c] = fmaf(dz, xv[c] - in_mean[c0 + c], s2[c]);
        }
      }
      if (CG >= VN) {
#pragma unroll
        for (int c = 0; c < CG; c += VN) VecIO<T>::store(dx + p * CS + c0 + c, d + c);
      } else {
#pragma unroll
        for (int c = 0; c < CG; ++c) Elem<T>::store(dx + p * CS + c0 + c, d[c]);
      }
      if (c0 == 0) {
#pragma unroll
        for (int k = 0; k < KB; ++k) ab[k] += g[k];
      }
    }
    // wave butterfly, then the 4 waves in fixed order
#pragma unroll
    for (int k = 0; k < KB; ++k) {
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const float s = wave_sum(aw[k][c]);
        if (lane == 0) red[wave][k * (CG + 1) + c] = s;
      }
      const float sb = wave_sum(ab[k]);
      if (lane == 0) red[wave][k * (CG + 1) + CG] = sb;
    }
    if (IN_BN) {
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const float t1 = wave_sum(s1[c]), t2 = wave_sum(s2[c]);
        if (lane == 0) { red2[wave][c] = t1; red2[wave][CG + c] = t2; }
      }
    }
    __syncthreads();
    if (IN_BN && threadIdx.x < 2 * CG) {
      const int which = threadIdx.x / CG, c = threadIdx.x - which * CG;
      if (c0 + c < CS)
        rows[((size_t)blockIdx.x * 2 + which) * CS + c0 + c] =
            (red2[0][threadIdx.x] + red2[1][threadIdx.x]) + (red2[2][threadIdx.x] + red2[3][threadIdx.x]);
    }
    for (int i = threadIdx.x; i < KB * (CG + 1); i += 256) {
      const int k = i / (CG + 1), c = i - k * (CG + 1);
      const float s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
      if (k < K) {
        if (c < CG) {
          if (c0 + c < C) mypart[k * (C + 1) + c0 + c] = s;
        } else if (c0 == 0) {
          mypart[k * (C + 1) + C] = s;
        }
      }
    }
    __syncthreads();
  }
}

template <typename T>
static void launch_conv1x1_fwd_bn(int grid, hipStream_t st, const void* x, size_t npix, int C, int CS, int K, const float* w,
                                  const float* b, float* out, const float* sc, const float* sh) {
  if (K <= 4) SPCL_LAUNCH((conv1x1_fwd_kernel<T, 4, true>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out, sc, sh);
  else if (K <= 8) SPCL_LAUNCH((conv1x1_fwd_kernel<T, 8, true>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out, sc, sh);
  else SPCL_LAUNCH((conv1x1_fwd_kernel<T, 16, true>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out, sc, sh);
}
template <typename T>
static void launch_conv1x1_bwd_bn(int grid, hipStream_t st, const void* x, const float* dout, size_t npix, int C, int CS,
                                  int K, const float* w, void* dx, float* ws, const float* sc, const float* sh,
                                  const float* mu, float* rows) {
  if (K <= 4) SPCL_LAUNCH((conv1x1_bwd_kernel<T, 4, true>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws, sc, sh, mu, rows);
  else if (K <= 8) SPCL_LAUNCH((conv1x1_bwd_kernel<T, 8, true>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws, sc, sh, mu, rows);
  else SPCL_LAUNCH((conv1x1_bwd_kernel<T, 16, true>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws, sc, sh, mu, rows);
}
template <typename T>
static void launch_conv1x1_fwd(int grid, hipStream_t st, const void* x, size_t npix, int C, int CS, int K, const float* w,
                               const float* b, float* out) {
  if (K <= 4) SPCL_LAUNCH((conv1x1_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out);
  else if (K <= 8) SPCL_LAUNCH((conv1x1_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out);
  else SPCL_LAUNCH((conv1x1_fwd_kernel<T, 16>), dim3(grid), dim3(256), 0, st, (const T*)x, npix, C, CS, K, w, b, out);
}
template <typename T>
static void launch_conv1x1_bwd(int grid, hipStream_t st, const void* x, const float* dout, size_t npix, int C, int CS,
                               int K, const float* w, void* dx, float* ws) {
  if (K <= 4) SPCL_LAUNCH((conv1x1_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws);
  else if (K <= 8) SPCL_LAUNCH((conv1x1_bwd_kernel<T, 8>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws);
  else SPCL_LAUNCH((conv1x1_bwd_kernel<T, 16>), dim3(grid), dim3(256), 0, st, (const T*)x, dout, npix, C, CS, K, w, (T*)dx, ws);
}

// out[i] = scale * sum over workgroup partials: one wave per output, fixed-order butterfly
__global__ __launch_bounds__(256) void head_partial_sum_kernel(const float* __restrict__ partial, int nwg, int n,
                                                               float scale, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
#pragma unroll 4
  for (int w = lane; w < nwg; w += 64) s += partial[(size_t)w * n + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = s * scale;
}

// dW[k][c], db[k] from the [nwg][K][C+1] partial rows: one wave per output, lanes stride the rows, fixed-order butterfly
__global__ __launch_bounds__(256) void conv1x1_finish_kernel(const float* __restrict__ partial, int nwg, int K, int C,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), n = K * (C + 1);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
#pragma unroll 4
  for (int w = lane; w < nwg; w += 64) s += partial[(size_t)w * n + i];
  s = wave_sum(s);
  if (lane == 0) {
    const int k = i / (C + 1), c = i - k * (C + 1);
    if (c < C) dw[k * C + c] = s;
    else db[k] = s;
  }
}

// ------------------------------------------------------------------------------------------------ softmax over K
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ logits, size_t npix, int K,
                                                          float* __restrict__ prob) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    float v[HEAD_MAX_K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) { v[k] = logits[p * K + k]; m = fmaxf(m, v[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) { v[k] = expf(v[k] - m); s += v[k]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) prob[p * K + k] = v[k] * inv;
  }
}

// dlogits = p * (dp - sum_k dp_k p_k)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ dprob,
                                                          size_t npix, int K, float* __restrict__ dlogits) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    float pv[HEAD_MAX_K], dv[HEAD_MAX_K], dot = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) { pv[k] = prob[p * K + k]; dv[k] = dprob[p * K + k]; dot = fmaf(pv[k], dv[k], dot); }
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) dlogits[p * K + k] = pv[k] * (dv[k] - dot);
  }
}

// ------------------------------------------------------------------------------------------------ KL_div
// partial[wg] = sum over the workgroup's positions of sum_k -t log((p+eps)/(t+eps))
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ prob, const float* __restrict__ target,
                                                     size_t npix, int K, float eps, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) {
        const float t = target[p * K + k];
        s -= t * logf((prob[p * K + k] + eps) / (t + eps));
      }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dprob = gscale[0] * (-t / (p + eps)) / npix
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ target,
                                                     size_t n, float eps, float inv_m,
                                                     const float* __restrict__ gscale, float* __restrict__ dprob) {
  const float gs = gscale[0] * inv_m;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dprob[i] = -gs * target[i] / (prob[i] + eps);
}

// ------------------------------------------------------------------------------------------------ labels
__global__ __launch_bounds__(256) void one_hot_kernel(const int64_t* __restrict__ labels, size_t npix, int K,
                                                      float* __restrict__ out) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const int64_t l = labels[p];
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) out[p * K + k] = l == k ? 1.f : 0.f;
  }
}

// first maximum over the K classes (torch.max(1)[1])
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, size_t npix, int K,
                                                     int64_t* __restrict__ out) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    float m = logits[p * K];
    int best = 0;
    for (int k = 1; k < K; ++k) {
      const float v = logits[p * K + k];
      if (v > m) { m = v; best = k; }
    }
    out[p] = best;
  }
}

// inter[n][c] += [pred==c && tgt==c], uni[n][c] += [pred==c] + [tgt==c]; integer atomics (exact, order-free)
__global__ __launch_bounds__(256) void dice_counts_kernel(const int64_t* __restrict__ pred,
                                                          const int64_t* __restrict__ target, int per_sample, int C,
                                                          unsigned long long* __restrict__ inter,
                                                          unsigned long long* __restrict__ uni) {
  __shared__ unsigned int si[64], su[64];
  const int n = blockIdx.y;
  if (threadIdx.x < 64) si[threadIdx.x] = su[threadIdx.x] = 0u;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per_sample; i += gridDim.x * 256) {
    const int64_t pv = pred[(size_t)n * per_sample + i], tv = target[(size_t)n * per_sample + i];
    if (pv >= 0 && pv < C) atomicAdd(&su[pv], 1u);
    if (tv >= 0 && tv < C) atomicAdd(&su[tv], 1u);
    if (pv == tv && pv >= 0 && pv < C) atomicAdd(&si[pv], 1u);
  }
  __syncthreads();
  if (threadIdx.x < C) {
    if (si[threadIdx.x]) atomicAdd(&inter[(size_t)n * C + threadIdx.x], (unsigned long long)si[threadIdx.x]);
    if (su[threadIdx.x]) atomicAdd(&uni[(size_t)n * C + threadIdx.x], (unsigned long long)su[threadIdx.x]);
  }
}

// ------------------------------------------------------------------------------------------------ fused supervised loss
// The fine-tune criterion as the reference composes it (semi_seg/epochers/new_epocher.py:268-282):
//   onehot = class2one_hot(target, C);  loss = KL_div(logits.softmax(1), onehot);  Dice counts of logits.max(1)[1] vs target
// in ONE pass over the class map instead of seven (softmax, one-hot, KL forward, arg-max, Dice counts, and -- for a unit
// upstream gradient -- KL backward and softmax backward): per pixel the same arithmetic in the same order as the separate
// kernels (softmax_fwd_kernel, kl_fwd_kernel with t = one-hot: the terms with t == 0 add exact zeros; kl_bwd_kernel +
// softmax_bwd_kernel with grad_loss == 1), so a pixel's probabilities, loss term and gradient are bit-identical to theirs.
// Grid (gx, B): blockIdx.y = sample, as dice_counts_kernel.  partial[wg] = the workgroup's loss sum.
__global__ __launch_bounds__(256) void sup_loss_fwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, int per_sample, int K,
                                                           float eps, float inv_m, float* __restrict__ partial,
                                                           float* __restrict__ dlogits, int C,
                                                           unsigned* __restrict__ cnt /* [workgroup][2][HEAD_MAX_K] */) {
  __shared__ float red[4];
  __shared__ unsigned int si[64], su[64];
  const int n = blockIdx.y;
  if (threadIdx.x < 64) si[threadIdx.x] = su[threadIdx.x] = 0u;
  __syncthreads();
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per_sample; i += gridDim.x * 256) {
    const size_t p = (size_t)n * per_sample + i;
    const int64_t l = labels[p];
    float v[HEAD_MAX_K];
    float m = -INFINITY;
    int best = 0;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) {
        v[k] = logits[p * K + k];
        m = fmaxf(m, v[k]);
        if (k > 0 && v[k] > v[best]) best = k;  // first maximum (torch.max(1)[1])
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) { v[k] = expf(v[k] - m); sum += v[k]; }
    const float inv = 1.f / sum;
    float pl = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) { v[k] = v[k] * inv; pl = (l == k) ? v[k] : pl; }
    const bool lab_ok = l >= 0 && l < K;
    float dpl = 0.f, dot = 0.f;
    if (lab_ok) {
      s -= 1.f * logf((pl + eps) / (1.f + eps));
      dpl = -(1.f * inv_m) * 1.f / (pl + eps);  // kl_bwd_kernel with gscale == 1: -gs * t / (p + eps)
      dot = fmaf(pl, dpl, 0.f);                // softmax_bwd_kernel's dot: the one non-zero term
    }
#pragma unroll
    for (int k = 0; k < HEAD_MAX_K; ++k)
      if (k < K) dlogits[p * K + k] = v[k] * (((lab_ok && l == k) ? dpl : -0.f) - dot);
    if (best < C) atomicAdd(&su[best], 1u);
    if (l >= 0 && l < C) atomicAdd(&su[l], 1u);
    if (l == best && best < C) atomicAdd(&si[best], 1u);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  // the workgroup's Dice counts as a row of the workspace (sup_loss_finish_kernel adds the rows of an image up in fixed
  // order: no atomics on the result, which therefore needs no zero fill in front of the launch)
  if (threadIdx.x < C) {
    unsigned* row = cnt + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * HEAD_MAX_K;
    row[threadIdx.x] = si[threadIdx.x];
    row[HEAD_MAX_K + threadIdx.x] = su[threadIdx.x];
  }
}

// K == C == 4 (the ACDC fine-tune head): the pixel's logits / gradient as one 16-byte access, the three Dice counters as
// 16-bit fields of two 64-bit registers per thread (a thread sees < 2^15 pixels: the host checks), summed over the wave at the
// end -- the LDS atomics of the general kernel serialise 64 lanes on four addresses three times per pixel (37.9 -> 16 us at
// 32 x 224^2).  The per-pixel arithmetic is the general kernel's, statement for statement (bit-identical gradient).
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void sup_loss_fwd4_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ labels, int per_sample, float eps,
                                                            float inv_m, float* __restrict__ partial,
                                                            float* __restrict__ dlogits,
                                                            unsigned* __restrict__ cnt /* [workgroup][2][HEAD_MAX_K] */) {
  constexpr int K = 4;
  __shared__ float red[4];
  __shared__ unsigned int si[4][4], su[4][4];
  const int n = blockIdx.y;
  float s = 0.f;
  unsigned long long cu = 0ull, ci = 0ull;
#pragma unroll 2
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per_sample; i += gridDim.x * 256) {
    const size_t p = (size_t)n * per_sample + i;
    const int64_t l = labels[p];
    const f32x4 raw = *(const f32x4*)(logits + p * K);
    float v[K];
    float m = -INFINITY;
    int best = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      v[k] = raw[k];
      m = fmaxf(m, v[k]);
    }
#pragma unroll
    for (int k = 1; k < K; ++k)
      if (v[k] > v[best]) best = k;  // first maximum (torch.max(1)[1])
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - m); sum += v[k]; }
    const float inv = 1.f / sum;
    float pl = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = v[k] * inv; pl = (l == k) ? v[k] : pl; }
    const bool lab_ok = l >= 0 && l < K;
    float dpl = 0.f, dot = 0.f;
    if (lab_ok) {
      s -= 1.f * logf((pl + eps) / (1.f + eps));
      dpl = -(1.f * inv_m) * 1.f / (pl + eps);
      dot = fmaf(pl, dpl, 0.f);
    }
    f32x4 out;
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = v[k] * (((lab_ok && l == k) ? dpl : -0.f) - dot);
    *(f32x4*)(dlogits + p * K) = out;
    cu += 1ull << (16 * best);
    if (lab_ok) cu += 1ull << (16 * (int)l);
    if (l == best) ci += 1ull << (16 * best);
  }
  s = wave_sum(s);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const unsigned tu = wave_sum_u32((unsigned)(cu >> (16 * c)) & 0xffffu);
    const unsigned ti = wave_sum_u32((unsigned)(ci >> (16 * c)) & 0xffffu);
    if (lane == 0) { su[wave][c] = tu; si[wave][c] = ti; }
  }
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  if (threadIdx.x < K) {
    const unsigned ti = si[0][threadIdx.x] + si[1][threadIdx.x] + si[2][threadIdx.x] + si[3][threadIdx.x];
    const unsigned tu = su[0][threadIdx.x] + su[1][threadIdx.x] + su[2][threadIdx.x] + su[3][threadIdx.x];
    unsigned* row = cnt + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * HEAD_MAX_K;
    row[threadIdx.x] = ti;
    row[HEAD_MAX_K + threadIdx.x] = tu;
  }
}

// the finish of spcl_sup_loss_forward: workgroup 0 = the loss (head_partial_sum_kernel's wave 0: same order, same bits), the
// others one thread per (image, which, class): the image's gx count rows added in index order -> inter / union (plain stores)
__global__ __launch_bounds__(256) void sup_loss_finish_kernel(const float* __restrict__ partial, int nwg, float scale,
                                                              float* __restrict__ loss, const unsigned* __restrict__ cnt,
                                                              int gx, int B, int C, unsigned long long* __restrict__ inter,
                                                              unsigned long long* __restrict__ uni) {
  if (blockIdx.x == 0) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    float s = 0.f;
#pragma unroll 4
    for (int w = lane; w < nwg; w += 64) s += partial[w];
    s = wave_sum(s);
    if (lane == 0) loss[0] = s * scale;
    return;
  }
  const int o = (blockIdx.x - 1) * 256 + threadIdx.x;
  if (o >= B * 2 * C) return;
  const int n = o / (2 * C), r = o - n * 2 * C, which = r / C, c = r - which * C;
  unsigned long long t = 0ull;
  for (int bx = 0; bx < gx; ++bx) t += cnt[(size_t)(n * gx + bx) * 2 * HEAD_MAX_K + which * HEAD_MAX_K + c];
  (which ? uni : inter)[(size_t)n * C + c] = t;
}

static int head_grid(size_t n, int cap) {
  size_t g = (n + 255) / 256;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_conv1x1_forward(const void* x, int dtype, size_t npix, int C, int CS, int K, const float* w,
                                    const float* b, float* out, void* stream) {
  SPCL_CHECK_ARG(x && w && b && out, "conv1x1_forward: null pointer");
  SPCL_CHECK_ARG(npix > 0 && C > 0 && C <= CS && C <= HEAD_MAX_C && K > 0 && K <= HEAD_MAX_K, "conv1x1_forward: shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = head_grid(npix, 4096);
  prof_cost((double)npix * (CS * (dtype == SPCL_F32 ? 4.0 : 2.0) + K * 4.0), 2.0 * npix * C * K);
  if (dtype == SPCL_F32) launch_conv1x1_fwd<float>(grid, st, x, npix, C, CS, K, w, b, out);
  else if (dtype == SPCL_BF16) launch_conv1x1_fwd<bf16_t>(grid, st, x, npix, C, CS, K, w, b, out);
  else {
    set_error("conv1x1_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv1x1_forward");
  return SPCL_OK;
}

extern "C" size_t spcl_conv1x1_bwd_workspace_bytes(int C, int K) {
  return (size_t)HEAD_RED_WG * K * (C + 1) * sizeof(float);
}

extern "C" int spcl_conv1x1_backward(const void* x, const float* dout, int dtype, size_t npix, int C, int CS, int K,
                                     const float* w, void* dx, float* dw, float* db, float* ws, void* stream) {
  SPCL_CHECK_ARG(x && dout && w && dx && dw && db && ws, "conv1x1_backward: null pointer");
  SPCL_CHECK_ARG(npix > 0 && C > 0 && C <= CS && C <= HEAD_MAX_C && K > 0 && K <= HEAD_MAX_K && CS % 16 == 0,
                 "conv1x1_backward: shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = head_grid(npix, HEAD_RED_WG);
  prof_cost((double)npix * (2.0 * CS * (dtype == SPCL_F32 ? 4.0 : 2.0) + K * 4.0), 4.0 * npix * C * K);
  if (dtype == SPCL_F32) launch_conv1x1_bwd<float>(grid, st, x, dout, npix, C, CS, K, w, dx, ws);
  else if (dtype == SPCL_BF16) launch_conv1x1_bwd<bf16_t>(grid, st, x, dout, npix, C, CS, K, w, dx, ws);
  else {
    set_error("conv1x1_backward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH(conv1x1_finish_kernel, dim3(cdiv(K * (C + 1), 4)), dim3(256), 0, st, (const float*)ws, grid, K, C, dw,
              db);
  SPCL_LAUNCH_CHECK("conv1x1_backward");
  return SPCL_OK;
}

// The decoder's last BatchNorm + ReLU folded into the 1x1 convolution on both sides (unet.py:82 -> :229): y is the raw output
// of the last 3x3 convolution, the class map is conv1x1(relu(scale y + shift)) -- the activation tensor is never written.
extern "C" int spcl_conv1x1_forward_bn(const void* y, int dtype, size_t npix, int C, int CS, int K, const float* scale,
                                       const float* shift, const float* w, const float* b, float* out, void* stream) {
  SPCL_CHECK_ARG(y && scale && shift && w && b && out, "conv1x1_forward_bn: null pointer");
  SPCL_CHECK_ARG(npix > 0 && C > 0 && C <= CS && CS <= HEAD_MAX_C && K > 0 && K <= HEAD_MAX_K, "conv1x1_forward_bn: shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = head_grid(npix, 4096);
  prof_cost((double)npix * (CS * (dtype == SPCL_F32 ? 4.0 : 2.0) + K * 4.0), 2.0 * npix * C * K);
  if (dtype == SPCL_F32) launch_conv1x1_fwd_bn<float>(grid, st, y, npix, C, CS, K, w, b, out, scale, shift);
  else if (dtype == SPCL_BF16) launch_conv1x1_fwd_bn<bf16_t>(grid, st, y, npix, C, CS, K, w, b, out, scale, shift);
  else {
    set_error("conv1x1_forward_bn: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv1x1_forward_bn");
  return SPCL_OK;
}

extern "C" int spcl_conv1x1_bwd_rows(size_t npix) { return head_grid(npix, HEAD_RED_WG); }

// ... and its backward: dact = gradient w.r.t. relu(scale y + shift) (what spcl_bnrelu_backward_rows applies), dw / db as
// spcl_conv1x1_backward, rows [spcl_conv1x1_bwd_rows(npix)][2][CS] = that BatchNorm's backward partial sums (sum dz,
// sum dz (y - mean)): the separate reduction pass over (y, dact) disappears.
extern "C" int spcl_conv1x1_backward_bn(const void* y, const float* dout, int dtype, size_t npix, int C, int CS, int K,
                                        const float* scale, const float* shift, const float* mean, const float* w,
                                        void* dact, float* dw, float* db, float* ws, float* rows, void* stream) {
  SPCL_CHECK_ARG(y && dout && scale && shift && mean && w && dact && dw && db && ws && rows, "conv1x1_backward_bn: null pointer");
  SPCL_CHECK_ARG(npix > 0 && C > 0 && C <= CS && CS <= HEAD_MAX_C && K > 0 && K <= HEAD_MAX_K && CS % 16 == 0,
                 "conv1x1_backward_bn: shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = head_grid(npix, HEAD_RED_WG);
  prof_cost((double)npix * (2.0 * CS * (dtype == SPCL_F32 ? 4.0 : 2.0) + K * 4.0), 4.0 * npix * C * K);
  if (dtype == SPCL_F32) launch_conv1x1_bwd_bn<float>(grid, st, y, dout, npix, C, CS, K, w, dact, ws, scale, shift, mean, rows);
  else if (dtype == SPCL_BF16) launch_conv1x1_bwd_bn<bf16_t>(grid, st, y, dout, npix, C, CS, K, w, dact, ws, scale, shift, mean, rows);
  else {
    set_error("conv1x1_backward_bn: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH(conv1x1_finish_kernel, dim3(cdiv(K * (C + 1), 4)), dim3(256), 0, st, (const float*)ws, grid, K, C, dw,
              db);
  SPCL_LAUNCH_CHECK("conv1x1_backward_bn");
  return SPCL_OK;
}

extern "C" int spcl_softmax_forward(const float* logits, size_t npix, int K, float* prob, void* stream) {
  SPCL_CHECK_ARG(logits && prob && npix > 0 && K > 0 && K <= HEAD_MAX_K, "softmax_forward: bad args");
  prof_cost((double)npix * K * 8.0, 0.0);
  SPCL_LAUNCH(softmax_fwd_kernel, dim3(head_grid(npix, 4096)), dim3(256), 0, (hipStream_t)stream, logits, npix, K, prob);
  SPCL_LAUNCH_CHECK("softmax_forward");
  return SPCL_OK;
}

extern "C" int spcl_softmax_backward(const float* prob, const float* dprob, size_t npix, int K, float* dlogits,
                                     void* stream) {
  SPCL_CHECK_ARG(prob && dprob && dlogits && npix > 0 && K > 0 && K <= HEAD_MAX_K, "softmax_backward: bad args");
  prof_cost((double)npix * K * 12.0, 0.0);
  SPCL_LAUNCH(softmax_bwd_kernel, dim3(head_grid(npix, 4096)), dim3(256), 0, (hipStream_t)stream, prob, dprob, npix, K,
              dlogits);
  SPCL_LAUNCH_CHECK("softmax_backward");
  return SPCL_OK;
}

// (HEAD_RED_WG partial sums; behind them spcl_sup_loss_forward's per-workgroup Dice count rows [HEAD_RED_WG][2][HEAD_MAX_K] u32)
extern "C" size_t spcl_kl_workspace_bytes(void) { return (size_t)HEAD_RED_WG * (1 + 2 * HEAD_MAX_K) * sizeof(float); }

extern "C" int spcl_kl_div_forward(const float* prob, const float* target, size_t npix, int K, float eps, float* ws,
                                   float* loss, void* stream) {
  SPCL_CHECK_ARG(prob && target && ws && loss && npix > 0 && K > 0 && K <= HEAD_MAX_K, "kl_div_forward: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int grid = head_grid(npix, HEAD_RED_WG);
  prof_cost((double)npix * K * 8.0, 0.0);
  SPCL_LAUNCH(kl_fwd_kernel, dim3(grid), dim3(256), 0, st, prob, target, npix, K, eps, ws);
  SPCL_LAUNCH(head_partial_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, grid, 1, 1.f / (float)npix, loss);
  SPCL_LAUNCH_CHECK("kl_div_forward");
  return SPCL_OK;
}

extern "C" int spcl_kl_div_backward(const float* prob, const float* target, size_t npix, int K, float eps,
                                    const float* grad_loss, float* dprob, void* stream) {
  SPCL_CHECK_ARG(prob && target && grad_loss && dprob && npix > 0 && K > 0, "kl_div_backward: bad args");
  prof_cost((double)npix * K * 12.0, 0.0);
  SPCL_LAUNCH(kl_bwd_kernel, dim3(head_grid(npix * K, 4096)), dim3(256), 0, (hipStream_t)stream, prob, target, npix * K,
              eps, 1.f / (float)npix, grad_loss, dprob);
  SPCL_LAUNCH_CHECK("kl_div_backward");
  return SPCL_OK;
}

extern "C" int spcl_one_hot(const int64_t* labels, size_t npix, int K, float* out, void* stream) {
  SPCL_CHECK_ARG(labels && out && npix > 0 && K > 0 && K <= HEAD_MAX_K, "one_hot: bad args");
  SPCL_LAUNCH(one_hot_kernel, dim3(head_grid(npix, 4096)), dim3(256), 0, (hipStream_t)stream, labels, npix, K, out);
  SPCL_LAUNCH_CHECK("one_hot");
  return SPCL_OK;
}

extern "C" int spcl_argmax_classes(const float* logits, size_t npix, int K, int64_t* out, void* stream) {
  SPCL_CHECK_ARG(logits && out && npix > 0 && K > 0, "argmax_classes: bad args");
  SPCL_LAUNCH(argmax_kernel, dim3(head_grid(npix, 4096)), dim3(256), 0, (hipStream_t)stream, logits, npix, K, out);
  SPCL_LAUNCH_CHECK("argmax_classes");
  return SPCL_OK;
}

extern "C" int spcl_sup_loss_forward(const float* logits, const int64_t* labels, int B, int per_sample, int K, float eps,
                                     float* ws, float* loss, float* dlogits_unit, int64_t* inter_zeroed,
                                     int64_t* union_zeroed, void* stream) {
  SPCL_CHECK_ARG(logits && labels && ws && loss && dlogits_unit && inter_zeroed && union_zeroed,
                 "sup_loss_forward: null pointer");
  SPCL_CHECK_ARG(B > 0 && per_sample > 0 && K > 0 && K <= HEAD_MAX_K, "sup_loss_forward: bad shape (K <= %d)", HEAD_MAX_K);
  SPCL_CHECK_ARG(B <= HEAD_RED_WG, "sup_loss_forward: at most %d samples per call (one partial sum per workgroup in ws)",
                 HEAD_RED_WG);
  hipStream_t st = (hipStream_t)stream;
  int gx = (per_sample + 255) / 256;
  const int cap = HEAD_RED_WG / B > 1 ? HEAD_RED_WG / B : 1;  // (the workspace holds HEAD_RED_WG partial sums)
  if (gx > cap) gx = cap;
  if (gx > 64) gx = 64;
  const size_t npix = (size_t)B * per_sample;
  unsigned* cnt = (unsigned*)(ws + HEAD_RED_WG);  // count rows of the gx * B <= HEAD_RED_WG workgroups
  prof_cost((double)npix * (K * 8.0 + 8.0), 0.0);
  static const bool no_k4 = lab_env("SPCL_SUP_LOSS_K4", 1) == 0;  // A/B switch
  if (K == 4 && !no_k4 && (per_sample + gx * 256 - 1) / (gx * 256) < 32768 && (uintptr_t)logits % 16 == 0 &&
      (uintptr_t)dlogits_unit % 16 == 0)
    SPCL_LAUNCH(sup_loss_fwd4_kernel, dim3(gx, B), dim3(256), 0, st, logits, labels, per_sample, eps, 1.f / (float)npix, ws,
                dlogits_unit, cnt);
  else
    SPCL_LAUNCH(sup_loss_fwd_kernel, dim3(gx, B), dim3(256), 0, st, logits, labels, per_sample, K, eps, 1.f / (float)npix, ws,
                dlogits_unit, K, cnt);
  SPCL_LAUNCH(sup_loss_finish_kernel, dim3(1 + (B * 2 * K + 255) / 256), dim3(256), 0, st, (const float*)ws, gx * B,
              1.f / (float)npix, loss, (const unsigned*)cnt, gx, B, K, (unsigned long long*)inter_zeroed,
              (unsigned long long*)union_zeroed);
  SPCL_LAUNCH_CHECK("sup_loss_forward");
  return SPCL_OK;
}

extern "C" int spcl_dice_counts(const int64_t* pred, const int64_t* target, int B, int per_sample, int C,
                                int64_t* inter_zeroed, int64_t* union_zeroed, void* stream) {
  SPCL_CHECK_ARG(pred && target && inter_zeroed && union_zeroed, "dice_counts: null pointer");
  SPCL_CHECK_ARG(B > 0 && per_sample > 0 && C > 0 && C <= 64, "dice_counts: bad shape");
  int gx = (per_sample + 255) / 256;
  if (gx > 64) gx = 64;
  SPCL_LAUNCH(dice_counts_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, pred, target, per_sample, C,
              (unsigned long long*)inter_zeroed, (unsigned long long*)union_zeroed);
  SPCL_LAUNCH_CHECK("dice_counts");
  return SPCL_OK;
}
