// BatchNorm2d (train/eval) + ReLU + 2x2 max-pool glue kernels for the NHWC encoder on gfx950.
// Replaces nn.BatchNorm2d(momentum) / nn.ReLU / nn.MaxPool2d(2,2) of semi_seg/arch/unet.py:73-77,118-121 and their
// autograd backward.  All of these are HBM-bound streaming kernels: 16-byte vector loads/stores per lane, per-channel
// reductions as per-workgroup partials + a fixed-order second stage (deterministic, no float atomics).
#include "common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
template <typename T> struct Chunk;
template <> struct Chunk<float> { static constexpr int EPC = 4; };
template <> struct Chunk<bf16_t> { static constexpr int EPC = 8; };

template <typename T> __device__ __forceinline__ void unpack(u32x4 raw, float* v);
template <> __device__ __forceinline__ void unpack<float>(u32x4 raw, float* v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(raw[e]);
}
template <> __device__ __forceinline__ void unpack<bf16_t>(u32x4 raw, float* v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = __uint_as_float(raw[e] << 16);
    v[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack(const float* v);
template <> __device__ __forceinline__ u32x4 pack<float>(const float* v) {
  return (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
}
template <> __device__ __forceinline__ u32x4 pack<bf16_t>(const float* v) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
  return o;
}

// ------------------------------------------------------------------------------------------------ statistics
// The conv epilogue leaves one row [3][CS] = (count, mean, M2) per pixel tile (f32, exact within the tile).  They are
// combined in DOUBLE as raw moments (n, sum x = n mean, sum x^2 = M2 + n mean^2) -- at 53 bits the final
// var = E[x^2] - mean^2 loses nothing that matters (relative error ~1e-16 (1 + mean^2/var)) and the inner loop is
// three FMAs with no division -- in a fixed order (deterministic): a workgroup = 16 channels x TL tile lanes; a lane
// walks its rows serially, the TL lanes of a channel are folded by a tree in LDS.  Up to BN_DIRECT_TILES rows one
// launch does everything; beyond that a first launch reduces groups of BN_GROUP_TILES rows to double-precision
// partial rows (n, sum x, sum x^2; kept behind the tile rows of the same buffer, see spcl_bn_stats_elems) and a
// second launch finishes from those.
constexpr int BN_GROUP_TILES = 256;
constexpr int BN_DIRECT_TILES = 2048;
__host__ __device__ inline int bn_groups(int ntiles) {
  return ntiles <= BN_DIRECT_TILES ? 0 : (ntiles + BN_GROUP_TILES - 1) / BN_GROUP_TILES;
}

struct BnFinalArgs {
  const float* gamma;
  const float* beta;
  float momentum, eps;
  float* running_mean;
  float* running_var;
  int64_t* nbt;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
};

// SRC = float: tile rows (count, mean, M2);  double: partial rows (n, sum x, sum x^2).
// FINAL: write the BatchNorm coefficients, else one partial row per blockIdx.y.
template <typename SRC, bool FINAL>
__global__ __launch_bounds__(1024) void bn_reduce_kernel(const SRC* __restrict__ rows, int nrows, int rows_per_group,
                                                         int C, int CS, double* __restrict__ partial, BnFinalArgs f) {
  __shared__ double red[3][64][16];
  const int TL = blockDim.x >> 4;
  const int c16 = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + c16;
  const int r0 = blockIdx.y * rows_per_group, r1 = min(nrows, r0 + rows_per_group);
  double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int r = r0 + tl; r < r1; r += TL) {
    const SRC* p = rows + (size_t)r * 3 * CS + c;
    const double a = (double)p[0], b = (double)p[CS], q = (double)p[2 * CS];
    if (sizeof(SRC) == 4) {
      const double ab = a * b;
      n += a;
      s1 += ab;
      s2 += fma(ab, b, q);
    } else {
      n += a;
      s1 += b;
      s2 += q;
    }
  }
  red[0][tl][c16] = n; red[1][tl][c16] = s1; red[2][tl][c16] = s2;
  __syncthreads();
  for (int o = TL >> 1; o > 0; o >>= 1) {
    if (tl < o) {
      n += red[0][tl + o][c16]; s1 += red[1][tl + o][c16]; s2 += red[2][tl + o][c16];
      red[0][tl][c16] = n; red[1][tl][c16] = s1; red[2][tl][c16] = s2;
    }
    __syncthreads();
  }
  if (tl != 0) return;
  if (!FINAL) {
    double* q = partial + (size_t)blockIdx.y * 3 * CS + c;
    q[0] = n; q[CS] = s1; q[2 * CS] = s2;
    return;
  }
  if (c >= C) {  // channel padding
    f.mean[c] = 0.f; f.invstd[c] = 0.f; f.scale[c] = 0.f; f.shift[c] = 0.f;
    return;
  }
  const double mu = s1 / n;
  const double m2 = fmax(s2 - s1 * mu, 0.0);
  const double var = m2 / n;
  const float is = 1.0f / sqrtf((float)var + f.eps);
  const float sc = f.gamma[c] * is;
  f.mean[c] = (float)mu;
  f.invstd[c] = is;
  f.scale[c] = sc;
  f.shift[c] = f.beta[c] - (float)mu * sc;
  if (f.running_mean != nullptr) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mu;
  if (f.running_var != nullptr) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
  }
  if (f.nbt != nullptr && c == 0) f.nbt[0] += 1;
}

__global__ __launch_bounds__(256) void bn_eval_affine_kernel(int C, int CS, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ rm, const float* __restrict__ rv,
                                                             float eps, float* mean, float* invstd, float* scale,
                                                             float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CS) return;
  if (c >= C) { mean[c] = 0.f; invstd[c] = 0.f; scale[c] = 0.f; shift[c] = 0.f; return; }
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * is;
  mean[c] = rm[c];
  invstd[c] = is;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// ------------------------------------------------------------------------------------------------ forward
// thread = (output position, 16-byte channel chunk).  POOL: position = 2x2 window, else a single pixel.
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bnrelu_fwd_kernel(const T* __restrict__ y, int N, int H, int W, int CS,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ act,
                                                         T* __restrict__ pool) {
  constexpr int EPC = Chunk<T>::EPC;
  const int CPC = CS / EPC;
  // POOL: positions are 2x2 windows on the CEIL grid; torch.max_pool2d floors, so a window cut by an odd edge
  // produces no pooled value but its pixels still get their activation
  const int PH = POOL ? (H + 1) / 2 : H, PW = POOL ? (W + 1) / 2 : W;
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * PH * PW * CPC;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CPC);
    const size_t pos = idx / CPC;
    const int ox = (int)(pos % PW), oy = (int)((pos / PW) % PH), n = (int)(pos / ((size_t)PW * PH));
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      *(f32x4*)&sc[e] = *(const f32x4*)(scale + cc * EPC + e);
      *(f32x4*)&sh[e] = *(const f32x4*)(shift + cc * EPC + e);
    }
    if (POOL) {
      float mx[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) mx[e] = 0.f;  // activations are >= 0
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
        if (yy < H && xx < W) {
          const size_t off = (((size_t)n * H + yy) * W + xx) * CS + cc * EPC;
          float v[EPC];
          unpack<T>(*(const u32x4*)(y + off), v);
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            v[e] = fmaxf(fmaf(sc[e], v[e], sh[e]), 0.f);
            mx[e] = fmaxf(mx[e], v[e]);
          }
          if (act != nullptr) *(u32x4*)(act + off) = pack<T>(v);
        }
      }
      if (pool != nullptr && oy < OH && ox < OW)
        *(u32x4*)(pool + (((size_t)n * OH + oy) * OW + ox) * CS + cc * EPC) = pack<T>(mx);
    } else {
      const size_t off = (((size_t)n * H + oy) * W + ox) * CS + cc * EPC;
      float v[EPC];
      unpack<T>(*(const u32x4*)(y + off), v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(sc[e], v[e], sh[e]), 0.f);
      *(u32x4*)(act + off) = pack<T>(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// One position = one 2x2 window (POOL) or one pixel, one 16-byte channel chunk.  The raw packed loads stay in
// registers and the math runs channel by channel (element-outer), which keeps the live state small (occupancy):
//   z = scale*y+shift;  g = dact (+ dpool at the window arg-max: first max in scan order like torch.max_pool2d)
//   dz = g * [z > 0]
template <typename T> struct Word;  // one 32-bit word of a packed chunk
template <> struct Word<float> {
  static constexpr int EPW = 1;
  static __device__ __forceinline__ float get(uint32_t w, int) { return __uint_as_float(w); }
  static __device__ __forceinline__ uint32_t make(const float* v) { return __float_as_uint(v[0]); }
};
template <> struct Word<bf16_t> {
  static constexpr int EPW = 2;
  static __device__ __forceinline__ float get(uint32_t w, int h) {
    return __uint_as_float(h ? (w & 0xffff0000u) : (w << 16));
  }
  static __device__ __forceinline__ uint32_t make(const float* v) {
    return (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  }
};

template <typename T, bool POOL, int K>
struct BwdRaw {
  u32x4 ry[K], rg[K], rdp;
  size_t off[K];
  bool valid[K], has_g, complete;
  __device__ __forceinline__ void load(const T* y, const T* dact, const T* dpool, int n, int oy, int ox, int H, int W,
                                       int CS, int cc) {
    constexpr int EPC = Chunk<T>::EPC;
    has_g = dact != nullptr;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int yy = POOL ? 2 * oy + (k >> 1) : oy, xx = POOL ? 2 * ox + (k & 1) : ox;
      valid[k] = yy < H && xx < W;
      off[k] = (((size_t)n * H + yy) * W + xx) * CS + cc * EPC;
      ry[k] = (u32x4){0u, 0u, 0u, 0u};
      rg[k] = (u32x4){0u, 0u, 0u, 0u};
      if (valid[k]) {
        ry[k] = *(const u32x4*)(y + off[k]);
        if (has_g) rg[k] = *(const u32x4*)(dact + off[k]);
      }
    }
    complete = POOL && dpool != nullptr && oy < H / 2 && ox < W / 2;  // floor semantics of max_pool2d
    rdp = (u32x4){0u, 0u, 0u, 0u};
    if (complete) rdp = *(const u32x4*)(dpool + (((size_t)n * (H / 2) + oy) * (W / 2) + ox) * CS + cc * EPC);
  }
  // dz of channel element (wi, h) for the K pixels; yk = the raw y values
  __device__ __forceinline__ void elem(int wi, int h, float sc, float sh, float* yk, float* dz) const {
    float z[K];
    int best = 0;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      yk[k] = Word<T>::get(ry[k][wi], h);
      z[k] = valid[k] ? fmaf(sc, yk[k], sh) : -1.f;
      const float a = fmaxf(z[k], 0.f);
      if (k == 0) m = a;
      else if (a > m) { m = a; best = k; }
    }
    const float dp = complete ? Word<T>::get(rdp[wi], h) : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float g = Word<T>::get(rg[k][wi], h);
      if (POOL) g += (k == best) ? dp : 0.f;
      dz[k] = z[k] > 0.f ? g : 0.f;
    }
  }
};

constexpr int BWD_MAX_WG = 512;

template <typename T, bool POOL>
__global__ __launch_bounds__(256, 4) void bnrelu_bwd_reduce_kernel(const T* __restrict__ y, const T* __restrict__ dact,
                                                                const T* __restrict__ dpool, int N, int H, int W,
                                                                int CS, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                float* __restrict__ partial /* [grid][2][CS] */) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  constexpr int K = POOL ? 4 : 1;
  __shared__ float red[256][2 * EPC + 1];
  const int CPC = CS / EPC;          // chunks per pixel
  const int PL = 256 / CPC;          // position lanes per workgroup
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  const int OH = POOL ? (H + 1) / 2 : H, OW = POOL ? (W + 1) / 2 : W;  // ceil grid: every pixel is visited once
  const size_t npos = (size_t)N * OH * OW;
  float sc[EPC], sh[EPC], mu[EPC], is[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e += 4) {
    *(f32x4*)&sc[e] = *(const f32x4*)(scale + cc * EPC + e);
    *(f32x4*)&sh[e] = *(const f32x4*)(shift + cc * EPC + e);
    *(f32x4*)&mu[e] = *(const f32x4*)(mean + cc * EPC + e);
    *(f32x4*)&is[e] = *(const f32x4*)(invstd + cc * EPC + e);
  }
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL) {
    for (size_t pos = (size_t)blockIdx.x * PL + pl; pos < npos; pos += (size_t)gridDim.x * PL) {
      const int ox = (int)(pos % OW), oy = (int)((pos / OW) % OH), n = (int)(pos / ((size_t)OW * OH));
      BwdRaw<T, POOL, K> b;
      b.load(y, dact, dpool, n, oy, ox, H, W, CS, cc);
#pragma unroll
      for (int wi = 0; wi < 4; ++wi)
#pragma unroll
        for (int h = 0; h < EPW; ++h) {
          const int e = wi * EPW + h;
          float yk[K], dz[K];
          b.elem(wi, h, sc[e], sh[e], yk, dz);
#pragma unroll
          for (int k = 0; k < K; ++k) {  // invalid pixels carry dz == 0
            s1[e] += dz[k];
            s2[e] = fmaf(dz[k], (yk[k] - mu[e]) * is[e], s2[e]);
          }
        }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[threadIdx.x][e] = s1[e];
    red[threadIdx.x][EPC + e] = s2[e];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * CS; o += 256) {
    const int which = o / CS, c = o - which * CS;
    const int ccc = c / EPC, e = c - ccc * EPC;
    float s = 0.f;
    for (int p = 0; p < PL; ++p) s += red[p * CPC + ccc][which * EPC + e];
    partial[((size_t)blockIdx.x * 2 + which) * CS + c] = s;
  }
}

// dbeta = sum dz, dgamma = sum dz*yhat.  One wave per channel: lanes stride over the workgroup partials, then a
// fixed-order butterfly -> deterministic.  For the apply pass the BN-backward is folded to  dy = scale*dz + A*y + B:
//   training: A = -scale*invstd*dgamma/M,  B = -scale*dbeta/M - A*mean;   eval: A = B = 0
__global__ __launch_bounds__(256) void bnrelu_bwd_fin_kernel(const float* __restrict__ partial, int nwg, int C, int CS,
                                                             float M, int training, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ ab) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= CS) return;
  float s1 = 0.f, s2 = 0.f;
  for (int w = lane; w < nwg; w += 64) {
    s1 += partial[((size_t)w * 2 + 0) * CS + c];
    s2 += partial[((size_t)w * 2 + 1) * CS + c];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    if (c < C) {
      dbeta[c] = s1;
      dgamma[c] = s2;
    }
    float A = 0.f, B = 0.f;
    if (training) {
      A = -scale[c] * invstd[c] * (s2 / M);
      B = -scale[c] * (s1 / M) - A * mean[c];
    }
    ab[c] = A;
    ab[CS + c] = B;
  }
}

template <typename T, bool POOL>
__global__ __launch_bounds__(256, 3) void bnrelu_bwd_apply_kernel(const T* __restrict__ y, const T* __restrict__ dact,
                                                               const T* __restrict__ dpool, int N, int H, int W,
                                                               int CS, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ ab, T* __restrict__ dy) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  constexpr int K = POOL ? 4 : 1;
  const int CPC = CS / EPC;
  const int OH = POOL ? (H + 1) / 2 : H, OW = POOL ? (W + 1) / 2 : W;
  const size_t total = (size_t)N * OH * OW * CPC;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % CPC);
    const size_t pos = idx / CPC;
    const int ox = (int)(pos % OW), oy = (int)((pos / OW) % OH), n = (int)(pos / ((size_t)OW * OH));
    float sc[EPC], sh[EPC], A[EPC], B[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      *(f32x4*)&sc[e] = *(const f32x4*)(scale + cc * EPC + e);
      *(f32x4*)&sh[e] = *(const f32x4*)(shift + cc * EPC + e);
      *(f32x4*)&A[e] = *(const f32x4*)(ab + cc * EPC + e);
      *(f32x4*)&B[e] = *(const f32x4*)(ab + CS + cc * EPC + e);
    }
    BwdRaw<T, POOL, K> b;
    b.load(y, dact, dpool, n, oy, ox, H, W, CS, cc);
    u32x4 out[K];
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
      float o[K][EPW];
#pragma unroll
      for (int h = 0; h < EPW; ++h) {
        const int e = wi * EPW + h;
        float yk[K], dz[K];
        b.elem(wi, h, sc[e], sh[e], yk, dz);
#pragma unroll
        for (int k = 0; k < K; ++k) o[k][h] = fmaf(sc[e], dz[k], fmaf(A[e], yk[k], B[e]));
      }
#pragma unroll
      for (int k = 0; k < K; ++k) out[k][wi] = Word<T>::make(o[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (b.valid[k]) *(u32x4*)(dy + b.off[k]) = out[k];
  }
}

static int stream_grid(size_t total_threads) {
  size_t g = (total_threads + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
static int bnrelu_bwd_launch(const void* y, const void* dact, const void* dpool, int N, int H, int W, int C, int CS,
                             const float* mean, const float* invstd, const float* scale, const float* shift,
                             int training, float* ws, float* dgamma, float* dbeta, void* dy, hipStream_t st) {
  constexpr int EPC = Chunk<T>::EPC;
  const bool pool = dpool != nullptr;
  const int OH = pool ? (H + 1) / 2 : H, OW = pool ? (W + 1) / 2 : W;
  const size_t npos = (size_t)N * OH * OW;
  const int CPC = CS / EPC;
  const int PL = 256 / CPC;
  int nwg = (int)((npos + PL - 1) / PL);
  if (nwg > BWD_MAX_WG) nwg = BWD_MAX_WG;
  float* partial = ws;                          // [nwg][2][CS]
  float* ab = ws + (size_t)BWD_MAX_WG * 2 * CS;  // [2][CS]: folded BN-backward coefficients
  const float M = (float)((size_t)N * H * W);
  if (pool) {
    hipLaunchKernelGGL((bnrelu_bwd_reduce_kernel<T, true>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, partial);
  } else {
    hipLaunchKernelGGL((bnrelu_bwd_reduce_kernel<T, false>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, partial);
  }
  hipLaunchKernelGGL(bnrelu_bwd_fin_kernel, dim3(cdiv(CS, 4)), dim3(256), 0, st, (const float*)partial, nwg, C, CS,
                     M, training, mean, invstd, scale, dgamma, dbeta, ab);
  const int grid = stream_grid(npos * CPC);
  if (pool) {
    hipLaunchKernelGGL((bnrelu_bwd_apply_kernel<T, true>), dim3(grid), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, scale, shift, (const float*)ab, (T*)dy);
  } else {
    hipLaunchKernelGGL((bnrelu_bwd_apply_kernel<T, false>), dim3(grid), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, scale, shift, (const float*)ab, (T*)dy);
  }
  return 0;
}

}  // namespace spcl

using namespace spcl;

extern "C" size_t spcl_bn_stats_elems(int ntiles, int CS) {
  return ((size_t)ntiles + 2 * (size_t)bn_groups(ntiles)) * 3 * CS;  // tile rows + double-precision partial rows
}

extern "C" int spcl_bn_finalize(float* stats, int ntiles, int C, int CS, const float* gamma, const float* beta,
                                float momentum, float eps, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                                void* stream) {
  SPCL_CHECK_ARG(stats && gamma && beta && mean && invstd && scale && shift, "bn_finalize: null pointer");
  SPCL_CHECK_ARG(ntiles > 0 && C > 0 && CS >= C && CS % 16 == 0, "bn_finalize: bad shape");
  hipStream_t st = (hipStream_t)stream;
  BnFinalArgs f{gamma, beta, momentum, eps, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift};
  const int groups = bn_groups(ntiles);
  if (groups == 0) {
    hipLaunchKernelGGL((bn_reduce_kernel<float, true>), dim3(CS / 16, 1), dim3(ntiles > 256 ? 1024 : 256), 0, st, stats,
                       ntiles, ntiles, C, CS, (double*)nullptr, f);
  } else {
    double* partial = (double*)(stats + (size_t)ntiles * 3 * CS);  // spcl_bn_stats_elems reserves it
    hipLaunchKernelGGL((bn_reduce_kernel<float, false>), dim3(CS / 16, groups), dim3(256), 0, st, stats, ntiles,
                       BN_GROUP_TILES, C, CS, partial, f);
    hipLaunchKernelGGL((bn_reduce_kernel<double, true>), dim3(CS / 16, 1), dim3(groups > 64 ? 1024 : 256), 0, st,
                       (const double*)partial, groups, groups, C, CS, (double*)nullptr, f);
  }
  SPCL_LAUNCH_CHECK("bn_finalize");
  return SPCL_OK;
}

extern "C" int spcl_bn_eval_affine(int C, int CS, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                   float* shift, void* stream) {
  SPCL_CHECK_ARG(gamma && beta && running_mean && running_var && mean && invstd && scale && shift,
                 "bn_eval_affine: null pointer");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(CS, 256)), dim3(256), 0, (hipStream_t)stream, C, CS, gamma, beta,
                     running_mean, running_var, eps, mean, invstd, scale, shift);
  SPCL_LAUNCH_CHECK("bn_eval_affine");
  return SPCL_OK;
}

extern "C" int spcl_bnrelu_pool_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                                        const float* shift, void* act_out, void* pool_out, void* stream) {
  SPCL_CHECK_ARG(y && scale && shift && (act_out || pool_out), "bnrelu_pool_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0, "bnrelu_pool_forward: bad shape");
  SPCL_CHECK_ARG(!pool_out || (H >= 2 && W >= 2), "bnrelu_pool_forward: 2x2 pooling needs H,W >= 2");
  hipStream_t st = (hipStream_t)stream;
  if (dtype != SPCL_F32 && dtype != SPCL_BF16) {
    set_error("bnrelu_pool_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  const int epc = dtype == SPCL_F32 ? 4 : 8;
  const size_t npos = pool_out ? (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) : (size_t)N * H * W;
  const int grid = stream_grid(npos * (CS / epc));
  if (dtype == SPCL_F32) {
    if (pool_out)
      hipLaunchKernelGGL((bnrelu_fwd_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)y, N, H, W, CS,
                         scale, shift, (float*)act_out, (float*)pool_out);
    else
      hipLaunchKernelGGL((bnrelu_fwd_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)y, N, H, W, CS,
                         scale, shift, (float*)act_out, (float*)pool_out);
  } else {
    if (pool_out)
      hipLaunchKernelGGL((bnrelu_fwd_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)y, N, H, W, CS,
                         scale, shift, (bf16_t*)act_out, (bf16_t*)pool_out);
    else
      hipLaunchKernelGGL((bnrelu_fwd_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)y, N, H, W,
                         CS, scale, shift, (bf16_t*)act_out, (bf16_t*)pool_out);
  }
  SPCL_LAUNCH_CHECK("bnrelu_pool_forward");
  return SPCL_OK;
}

extern "C" size_t spcl_bnrelu_bwd_workspace_bytes(int N, int H, int W, int CS) {
  (void)N; (void)H; (void)W;
  return ((size_t)BWD_MAX_WG * 2 * CS + 2 * (size_t)CS) * sizeof(float);
}

extern "C" int spcl_bnrelu_pool_backward(const void* y, const void* dact, const void* dpool, int dtype, int N, int H,
                                         int W, int C, int CS, const float* mean, const float* invstd,
                                         const float* scale, const float* shift, int training, float* ws,
                                         float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && (dact || dpool) && mean && invstd && scale && shift && ws && dgamma && dbeta && dy,
                 "bnrelu_pool_backward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 1024,
                 "bnrelu_pool_backward: bad shape");
  SPCL_CHECK_ARG(!dpool || (H >= 2 && W >= 2), "bnrelu_pool_backward: 2x2 pooling needs H,W >= 2");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                             dy, st);
  else if (dtype == SPCL_BF16)
    bnrelu_bwd_launch<bf16_t>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                              dy, st);
  else {
    set_error("bnrelu_pool_backward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_pool_backward");
  return SPCL_OK;
}
