// BatchNorm2d (train/eval) + ReLU + 2x2 max-pool glue kernels for the NHWC encoder on gfx950.
// Replaces nn.BatchNorm2d(momentum) / nn.ReLU / nn.MaxPool2d(2,2) of semi_seg/arch/unet.py:73-77,118-121 and their
// autograd backward.  All of these are HBM-bound streaming kernels: 16-byte vector loads/stores per lane, per-channel
// reductions as per-workgroup partials + a fixed-order second stage (deterministic, no float atomics).
#include <mutex>
#include <type_traits>
#include "bn_acc.hpp"
#include "common.hpp"
#include "image_acorr.hpp"

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
constexpr int BN_DIRECT_TILES = 1024;
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

__device__ __forceinline__ void store_agent(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// SRC = float: tile rows (count, mean, M2);  double: partial rows (n, sum x, sum x^2).
// FINAL: write the BatchNorm coefficients, else one partial row per blockIdx.y.
// CW channels per workgroup (16: 64-byte row segments; 4 workgroups of 4 channels were tried for the 64-channel /
// 2048-tile layers and were no faster: the single launch is a latency chain, which is why 1024 tiles already go two-level)
// Latency, not work, is what these launches cost (a trivial dependent launch replays in 1.5 us; this kernel took 5-9): the row
// lanes are folded by two wave shuffles and ONE barrier (it was a log2(64)-round LDS tree: -0.5 .. -1.4 us per launch, ten
// launches per pre-train step), and the channel's gamma / beta / running statistics are four plain loads issued together
// behind the reduction.  Requesting them FIRST (-DSPCL_BN_PREFETCH=1) looked like one round trip less and measured +4 us per
// launch: vector-memory results return in order, and those four lines (parameters the optimizer wrote, buffers last touched
// a step ago) are slower to arrive than the rows the producing convolution has just left -- everything queues behind them.
struct BnChan { float gamma, beta, rm, rv; };
// (four PLAIN loads at clamped / substituted addresses, issued together: inside `if`s -- a null running-statistics pointer, the
// channel padding -- every one of them was waited for at its join: four dependent ~1.7 us round trips per launch)
__device__ __forceinline__ BnChan bn_prefetch_channel(int c, int C, const BnFinalArgs& f) {
  const int cc = c < C ? c : C - 1;
  const float* rm = f.running_mean != nullptr ? f.running_mean : f.gamma;
  const float* rv = f.running_var != nullptr ? f.running_var : f.gamma;
  BnChan ch;
  ch.gamma = f.gamma[cc];
  ch.beta = f.beta[cc];
  ch.rm = rm[cc];
  ch.rv = rv[cc];
  return ch;
}
__device__ __forceinline__ void bn_final_channel(int c, int C, double n, double s1, double s2, const BnFinalArgs& f,
                                                 const BnChan& ch);
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m, 64);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the row lanes tl of one channel (threads tid = tl * CW + c16, CW = 16: a wave holds four row lanes of each channel):
// two shuffles inside the wave, the waves' values through LDS, thread (tl 0, c16) adds them in wave order.  Fixed order.
template <int CW>
__device__ __forceinline__ void bn_fold_lanes(double& n, double& s1, double& s2, double (*red)[1024 / 64][CW]) {
  static_assert(CW == 16, "four row lanes per wave");
  const int c16 = threadIdx.x % CW, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  n += shfl_xor_f64(n, 16); s1 += shfl_xor_f64(s1, 16); s2 += shfl_xor_f64(s2, 16);
  n += shfl_xor_f64(n, 32); s1 += shfl_xor_f64(s1, 32); s2 += shfl_xor_f64(s2, 32);
  if ((threadIdx.x & 63) < CW) { red[0][wave][c16] = n; red[1][wave][c16] = s1; red[2][wave][c16] = s2; }
  __syncthreads();
  if (threadIdx.x < CW) {
    n = red[0][0][c16]; s1 = red[1][0][c16]; s2 = red[2][0][c16];
    for (int w = 1; w < nw; ++w) { n += red[0][w][c16]; s1 += red[1][w][c16]; s2 += red[2][w][c16]; }
  }
}

template <typename SRC, bool FINAL, int CW>
__global__ __launch_bounds__(1024) void bn_reduce_kernel(const SRC* __restrict__ rows, int nrows, int rows_per_group,
                                                         int C, int CS, double* __restrict__ partial, BnFinalArgs f,
                                                         unsigned* tickets = nullptr) {
  __shared__ double red[3][1024 / 64][CW];
  __shared__ unsigned s_ticket;
  const int TL = blockDim.x / CW;
  const int c16 = threadIdx.x % CW, tl = threadIdx.x / CW;
  const int c = blockIdx.x * CW + c16;
  const bool finishes = FINAL || tickets != nullptr;  // this launch writes the coefficients (some workgroup of it)
  BnChan ch = {0.f, 0.f, 0.f, 0.f};
#ifndef SPCL_BN_PREFETCH
#define SPCL_BN_PREFETCH 0
#endif
  if (SPCL_BN_PREFETCH && finishes) ch = bn_prefetch_channel(c, C, f);  // (every thread: no divergent region around the loads)
  const int r0 = blockIdx.y * rows_per_group, r1 = min(nrows, r0 + rows_per_group);
  double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 8
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
  bn_fold_lanes<CW>(n, s1, s2, red);
  if (!FINAL && tickets != nullptr) {
    // Both levels in ONE launch: the group's row leaves with agent-scope stores (written through: no L2 write-back fence),
    // the workgroup takes a ticket once they are acknowledged, and the LAST group of this channel block folds all the rows
    // (agent-scope loads: another XCD's L2 never held them) and finishes.  Fixed order -> deterministic; the ticket resets
    // itself for the next launch.
    if (tl == 0) {
      double* q = partial + (size_t)blockIdx.y * 3 * CS + c;
      store_agent(q, n); store_agent(q + CS, s1); store_agent(q + 2 * CS, s2);
      __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    if (threadIdx.x == 0)
      s_ticket = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.y - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    n = 0.0; s1 = 0.0; s2 = 0.0;
    for (int r = tl; r < (int)gridDim.y; r += TL) {
      const double* p = partial + (size_t)r * 3 * CS + c;
      n += load_agent(p); s1 += load_agent(p + CS); s2 += load_agent(p + 2 * CS);
    }
    __syncthreads();
    bn_fold_lanes<CW>(n, s1, s2, red);
    if (tl == 0) bn_final_channel(c, C, n, s1, s2, f, SPCL_BN_PREFETCH ? ch : bn_prefetch_channel(c, C, f));
    return;
  }
  if (tl != 0) return;
  if (!FINAL) {
    double* q = partial + (size_t)blockIdx.y * 3 * CS + c;
    q[0] = n; q[CS] = s1; q[2 * CS] = s2;
    return;
  }
  bn_final_channel(c, C, n, s1, s2, f, SPCL_BN_PREFETCH ? ch : bn_prefetch_channel(c, C, f));
}

__device__ __forceinline__ void bn_final_channel(int c, int C, double n, double s1, double s2, const BnFinalArgs& f,
                                                 const BnChan& ch) {
  if (c >= C) {  // channel padding
    f.mean[c] = 0.f; f.invstd[c] = 0.f; f.scale[c] = 0.f; f.shift[c] = 0.f;
    return;
  }
  const double mu = s1 / n;
  const double m2 = fmax(s2 - s1 * mu, 0.0);
  const double var = m2 / n;
  const float is = 1.0f / sqrtf((float)var + f.eps);
  const float sc = ch.gamma * is;
  f.mean[c] = (float)mu;
  f.invstd[c] = is;
  f.scale[c] = sc;
  f.shift[c] = ch.beta - (float)mu * sc;
  if (f.running_mean != nullptr) f.running_mean[c] = (1.f - f.momentum) * ch.rm + f.momentum * (float)mu;
  if (f.running_var != nullptr) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    f.running_var[c] = (1.f - f.momentum) * ch.rv + f.momentum * (float)unbiased;
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

// ... for up to SPCL_BN_EVAL_MAX BatchNorms in ONE launch: an eval-mode forward pass of the full UNet asked for its 22 layers'
// coefficients one 5 us launch at a time -- a quarter of the GPU time of an 8-slice validation batch (round 6).
struct BnEvalItems {
  spcl_bn_eval_item it[SPCL_BN_EVAL_MAX];
  int blk_end[SPCL_BN_EVAL_MAX];
};
__global__ __launch_bounds__(256) void bn_eval_affine_multi_kernel(BnEvalItems p, int n) {
  int i = 0;
  for (int k = 0; k < n; ++k) i += (int)blockIdx.x >= p.blk_end[k] ? 1 : 0;
  if (i >= n) return;
  const spcl_bn_eval_item& q = p.it[i];
  const int c = ((int)blockIdx.x - (i > 0 ? p.blk_end[i - 1] : 0)) * 256 + threadIdx.x;
  if (c >= q.CS) return;
  float* mean = q.st;
  float* invstd = q.st + q.CS;
  float* scale = q.st + 2 * q.CS;
  float* shift = q.st + 3 * q.CS;
  if (c >= q.C) { mean[c] = 0.f; invstd[c] = 0.f; scale[c] = 0.f; shift[c] = 0.f; return; }
  const float is = 1.0f / sqrtf(q.running_var[c] + q.eps);
  const float sc = q.gamma[c] * is;
  mean[c] = q.running_mean[c];
  invstd[c] = is;
  scale[c] = sc;
  shift[c] = q.beta[c] - q.running_mean[c] * sc;
}

// ------------------------------------------------------------------------------------------------ streaming kernels
// BN-apply + ReLU (+ 2x2 max-pool) forward and backward are HBM streams.  Common thread geometry: a thread owns ONE
// 16-byte channel chunk `cc` for the whole launch -- its per-channel coefficients stay in registers -- and walks the
// positions pl, pl + PL, ... (PL = position lanes per workgroup).  No integer division in any loop (the image / row
// split is wave-uniform scalar work per ROW), several independent 16-byte loads in flight per thread.
//   non-pool: positions are pixels, the tensors are walked linearly;
//   pool:     positions are 2x2 windows on the CEIL grid, walked row by row (torch.max_pool2d floors: a window cut by
//             an odd edge produces no pooled value, but its pixels still get their activation / gradient).
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

template <int EPC> __device__ __forceinline__ void load_coef(float* dst, const float* src, int cc) {
#pragma unroll
  for (int e = 0; e < EPC; e += 4) *(f32x4*)&dst[e] = *(const f32x4*)(src + cc * EPC + e);
}

// Coefficients from a fixed-point accumulator block (bn_acc.hpp) in the prologue of a 256-thread streaming kernel, CS <= 256:
// thread c derives channel c's pair, LDS hands every thread the eight of its chunk; the launch's first workgroup also leaves
// what the finalize launch used to leave (forward: mean / invstd / scale / shift + running statistics; backward: dgamma,
// dbeta).  Must be called by ALL threads of the workgroup (a barrier inside).
template <int EPC>
__device__ __forceinline__ void acc_coef_fwd(const BnAccFwd& bn, int cc, float* sc, float* sh) {
  __shared__ float cl[2][256];
  const int c = threadIdx.x;
  if (c < bn.CS) {
    BnAccFwdRaw raw;
    raw.load(bn, c);
    float a, b;
    bn_acc_fwd_channel(bn, raw, c, a, b, blockIdx.x == 0);
    cl[0][c] = a;
    cl[1][c] = b;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sc[e] = cl[0][cc * EPC + e];
    sh[e] = cl[1][cc * EPC + e];
  }
}
template <int EPC>
__device__ __forceinline__ void acc_coef_bwd(const BnAccBwd& bw, int cc, float* A, float* B) {
  __shared__ float cl[2][256];
  const int c = threadIdx.x;
  if (c < bw.CS) {
    BnAccBwdRaw raw;
    raw.load(bw, c);
    float a, b;
    bn_acc_bwd_channel(bw, raw, c, a, b, blockIdx.x == 0);
    cl[0][c] = a;
    cl[1][c] = b;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    A[e] = cl[0][cc * EPC + e];
    B[e] = cl[1][cc * EPC + e];
  }
}

constexpr int STREAM_UNROLL = 2;
constexpr int IMG_WGRAD_WG = 2048;  // upper bound of the workgroups of the fused image-wgrad pass

// ---- forward, no pooling: act = relu(scale*y + shift)
template <typename T>
__global__ __launch_bounds__(256, 6) void bnrelu_fwd_lin_kernel(const T* __restrict__ y, size_t npix, int CS,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ act,
                                                             int AS /* elements between two pixels of act (>= CS) */,
                                                             BnAccFwd bn = BnAccFwd{}) {
  constexpr int EPC = Chunk<T>::EPC, U = STREAM_UNROLL;
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float sc[EPC], sh[EPC];
  if (bn.acc != nullptr) acc_coef_fwd<EPC>(bn, cc, sc, sh);  // (wave-uniform; before any thread leaves: a barrier inside)
  if (pl >= PL) return;
  if (bn.acc == nullptr) {
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
  }
  const size_t stride = (size_t)gridDim.x * PL;
  for (size_t p = (size_t)blockIdx.x * PL + pl; p < npix; p += U * stride) {
    u32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p + u * stride < npix) r[u] = *(const u32x4*)(y + (p + u * stride) * CS + cc * EPC);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p + u * stride >= npix) break;
      float v[EPC];
      unpack<T>(r[u], v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(sc[e], v[e], sh[e]), 0.f);
      *(u32x4*)(act + (p + u * stride) * AS + cc * EPC) = pack<T>(v);
    }
  }
}

// ---- forward, no pooling, with the GLOBAL AVERAGE of the activation as a side output: gap[n][c] = mean over the image's pixels
// of act as stored (rounded to T) -- what the projector's AdaptiveAvgPool2d((1, 1)) (contrastyou/projectors/heads.py:78-92,
// nn.py:56-58) would compute from the tensor this launch writes, without a launch of its own reading it back.  One workgroup
// per image (the small maps at the encoder's end: 14 x 14 x 256 is 100 KB); the pixel lanes of a channel chunk meet in LDS in
// fixed order.
template <typename T>
__global__ __launch_bounds__(256) void bnrelu_fwd_gap_kernel(const T* __restrict__ y, int HW, int C, int CS, int CW,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ act,
                                                            float* __restrict__ gap, BnAccFwd bn) {
  // workgroup = (image blockIdx.x, channels [c0, c0 + CW) with c0 = blockIdx.y * CW): 256 / (CW / EPC) pixel lanes
  constexpr int EPC = Chunk<T>::EPC, U = 4;
  __shared__ float part[256 * EPC];  // [pixel lane][CW] (PL * CW <= 256 * EPC)
  __shared__ float cl[2][256];
  const int c0 = blockIdx.y * CW;
  const int CPC = CW / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  const bool live = pl < PL;
  float sc[EPC], sh[EPC];
  if (bn.acc != nullptr) {
    // the window's coefficients from the accumulator block; the window's workgroup of image 0 writes them (and the running
    // statistics) where the finalize launch used to
    const int c = threadIdx.x;
    if (c < CW) {
      BnAccFwdRaw raw;
      raw.load(bn, c0 + c);
      float a_, b_;
      bn_acc_fwd_channel(bn, raw, c0 + c, a_, b_, blockIdx.x == 0);
      cl[0][c] = a_;
      cl[1][c] = b_;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sc[e] = cl[0][cc * EPC + e];
      sh[e] = cl[1][cc * EPC + e];
    }
  } else if (live) {
    load_coef<EPC>(sc, scale + c0, cc);
    load_coef<EPC>(sh, shift + c0, cc);
  }
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = 0.f;
  if (live) {
    const T* yb = y + (size_t)blockIdx.x * HW * CS + c0 + cc * EPC;
    T* ab = act + (size_t)blockIdx.x * HW * CS + c0 + cc * EPC;
    for (int p = pl; p < HW; p += U * PL) {
      u32x4 r[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u * PL < HW) r[u] = *(const u32x4*)(yb + (size_t)(p + u * PL) * CS);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (p + u * PL >= HW) break;
        float v[EPC];
        unpack<T>(r[u], v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(sc[e], v[e], sh[e]), 0.f);
        const u32x4 pk = pack<T>(v);
        *(u32x4*)(ab + (size_t)(p + u * PL) * CS) = pk;
        unpack<T>(pk, v);  // the mean is that of the STORED values
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] += v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) part[pl * CW + cc * EPC + e] = s[e];
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < CW && c0 + c < C) {
    float t = 0.f;
    for (int q = 0; q < PL; ++q) t += part[q * CW + c];
    gap[(size_t)blockIdx.x * C + c0 + c] = t / (float)HW;
  }
}

// ---- forward, no pooling, output written 2x2-replicated: up[n, 2h + a, 2w + b, :] = relu(scale*y[n, h, w, :] + shift) -- the
// activation of a block whose only consumer is the decoder's nn.Upsample(scale_factor=2) (semi_seg/arch/unet.py:89, nearest):
// the upsampled tensor is written directly, the low-resolution activation and the separate upsampling launch are skipped
template <typename T>
__global__ __launch_bounds__(256, 6) void bnrelu_fwd_up2_kernel(const T* __restrict__ y, unsigned nrows /* N*H */, int W,
                                                             int CS, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ up) {
  constexpr int EPC = Chunk<T>::EPC;
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  if (pl >= PL) return;
  float sc[EPC], sh[EPC];
  load_coef<EPC>(sc, scale, cc);
  load_coef<EPC>(sh, shift, cc);
  for (unsigned r = blockIdx.x; r < nrows; r += gridDim.x) {  // r = n*H + h -> output rows 2r, 2r + 1 (wave-uniform)
    const T* yr = y + (size_t)r * W * CS + cc * EPC;
    T* o0 = up + (size_t)(2 * r) * (2 * W) * CS + cc * EPC;
    for (int w = pl; w < W; w += 2 * PL) {
      u32x4 rr[2];
      const bool second = w + PL < W;
      rr[0] = *(const u32x4*)(yr + (size_t)w * CS);
      if (second) rr[1] = *(const u32x4*)(yr + (size_t)(w + PL) * CS);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !second) break;
        float v[EPC];
        unpack<T>(rr[u], v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fmaf(sc[e], v[e], sh[e]), 0.f);
        const u32x4 pv = pack<T>(v);
        T* o = o0 + (size_t)(2 * (w + u * PL)) * CS;
        *(u32x4*)o = pv;
        *(u32x4*)(o + CS) = pv;
        *(u32x4*)(o + (size_t)2 * W * CS) = pv;
        *(u32x4*)(o + (size_t)2 * W * CS + CS) = pv;
      }
    }
  }
}

// one 2x2 window of one channel chunk: raw loads (issued together), then the math
template <typename T>
struct Window {
  u32x4 ry[4];
  bool valid[4];
  size_t off[4];
  // (n, oy) wave-uniform; ox per thread
  // FAST: H and W even -- every window is complete (the validity flags fold away: no per-load / per-store branches)
  template <bool FAST = false>
  __device__ __forceinline__ void load(const T* y, size_t row0, int oy, int ox, int H, int W, int CS, int cc) {
    constexpr int EPC = Chunk<T>::EPC;
    const bool y1 = FAST || 2 * oy + 1 < H, x1 = FAST || 2 * ox + 1 < W;
    const size_t o00 = (row0 * W + 2 * ox) * CS + cc * EPC;  // row0 = n*H + 2*oy
    off[0] = o00; off[1] = o00 + CS; off[2] = o00 + (size_t)W * CS; off[3] = off[2] + CS;
    valid[0] = true; valid[1] = x1; valid[2] = y1; valid[3] = x1 && y1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ry[k] = (u32x4){0u, 0u, 0u, 0u};
      if (valid[k]) ry[k] = *(const u32x4*)(y + off[k]);
    }
  }
};

// ---- forward with 2x2 max-pool (act optional)
template <typename T, bool FAST>
__global__ __launch_bounds__(256, 4) void bnrelu_fwd_pool_kernel(const T* __restrict__ y, int N, int H, int W, int CS,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ act,
                                                              T* __restrict__ pool, int AS, BnAccFwd bn = BnAccFwd{}) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float sc[EPC], sh[EPC];
  if (bn.acc != nullptr) acc_coef_fwd<EPC>(bn, cc, sc, sh);  // (wave-uniform; before any thread leaves: a barrier inside)
  if (pl >= PL) return;
  const int PH = (H + 1) / 2, PW = (W + 1) / 2, OH = H / 2, OW = W / 2;
  if (bn.acc == nullptr) {
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
  }
  const int rows = N * PH;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / PH, oy = r - n * PH;  // wave-uniform
    const size_t row0 = (size_t)n * H + 2 * oy;
    for (int ox = pl; ox < PW; ox += 2 * PL) {
      Window<T> w[2];
      const bool second = ox + PL < PW;
      w[0].template load<FAST>(y, row0, oy, ox, H, W, CS, cc);
      if (second) w[1].template load<FAST>(y, row0, oy, ox + PL, H, W, CS, cc);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !second) break;
        u32x4 mxw, aw[4];
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
          float mx[EPW], a[4][EPW];
#pragma unroll
          for (int h = 0; h < EPW; ++h) {
            const int e = wi * EPW + h;
            mx[h] = 0.f;  // activations are >= 0
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              a[k][h] = fmaxf(fmaf(sc[e], Word<T>::get(w[u].ry[k][wi], h), sh[e]), 0.f);
              if (w[u].valid[k]) mx[h] = fmaxf(mx[h], a[k][h]);
            }
          }
          mxw[wi] = Word<T>::make(mx);
#pragma unroll
          for (int k = 0; k < 4; ++k) aw[k][wi] = Word<T>::make(a[k]);
        }
        const int oxu = ox + u * PL;
        if (act != nullptr) {
          // (act may be a channel slice of a wider tensor -- the skip half of a decoder concatenation: AS elements per pixel)
          const size_t a00 = (row0 * W + 2 * oxu) * AS + cc * EPC;
          const size_t aoff[4] = {a00, a00 + AS, a00 + (size_t)W * AS, a00 + (size_t)W * AS + AS};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (w[u].valid[k]) *(u32x4*)(act + aoff[k]) = aw[k];
        }
        if (pool != nullptr && (FAST || (oy < OH && oxu < OW)))
          *(u32x4*)(pool + (((size_t)n * OH + oy) * OW + oxu) * CS + cc * EPC) = mxw;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
//   z = scale*y + shift;  g = dact (+ dpool at the window arg-max: first max in scan order like torch.max_pool2d);
//   dz = g * [z > 0];  dbeta = sum dz, dgamma = sum dz*yhat;
//   training: dy = scale*(dz - dbeta/M - yhat*dgamma/M), folded to dy = scale*dz + A*y + B;  eval: dy = scale*dz.
// Pass 1 (reduce) leaves per-workgroup partials, `fin` sums them in fixed order (deterministic), pass 2 applies.
constexpr int BWD_MAX_WG = 2048;
constexpr int STREAM_MAX_WG = 2048;  // long-lived workgroups: launching a wave costs more than an iteration

template <typename T, int EPC>
__device__ __forceinline__ void wg_reduce_partials(const float* s1, const float* s2, int CS, int CPC, int PL,
                                                   float* __restrict__ partial, float (*red)[2 * EPC + 1],
                                                   long long* __restrict__ acc = nullptr) {
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[threadIdx.x][e] = s1[e];
    red[threadIdx.x][EPC + e] = s2[e];
  }
  __syncthreads();
  if (acc != nullptr) {
    // thread o -> word o of the block's row: channel o / 4, sum (o / 2) % 2, limb o % 2 (the two limb threads of a sum both
    // form it -- the same fixed-order LDS walk -- and each adds its limb: consecutive lanes, consecutive words)
    for (int o = threadIdx.x; o < 4 * CS; o += 256) {
      const int c = o >> 2, which = (o >> 1) & 1, limb = o & 1;
      const int ccc = c / EPC, e = c - ccc * EPC;
      float s = 0.f;
      for (int p = 0; p < PL; ++p) s += red[p * CPC + ccc][which * EPC + e];
      bn_acc_add_word(acc, CS, (int)(blockIdx.x & (BN_ACC_REPLICAS - 1)), c, which, limb, s);
    }
    return;
  }
  for (int o = threadIdx.x; o < 2 * CS; o += 256) {
    const int which = o / CS, c = o - which * CS;
    const int ccc = c / EPC, e = c - ccc * EPC;
    float s = 0.f;
    for (int p = 0; p < PL; ++p) s += red[p * CPC + ccc][which * EPC + e];
    partial[((size_t)blockIdx.x * 2 + which) * CS + c] = s;
  }
}

// ---- pass 1, no pooling (linear)
template <typename T>
__global__ __launch_bounds__(256, 5) void bnrelu_bwd_reduce_lin_kernel(const T* __restrict__ y, const T* __restrict__ g,
                                                                    size_t npix, int CS,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    float* __restrict__ partial /* [grid][2][CS] */,
                                                                    int GS /* elements between two pixels of g */,
                                                                    long long* __restrict__ acc = nullptr) {
  // acc != null (here and in the other reduction passes): the workgroup's sums are ADDED to that fixed-point block (bn_acc.hpp;
  // sum dz (y - mean) NOT scaled by invstd: the consumer does) instead of leaving a partial row for the finalize launch
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW, U = STREAM_UNROLL;
  __shared__ float red[256][2 * EPC + 1];
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL) {
    float sc[EPC], sh[EPC], mu[EPC], is[EPC];
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
    load_coef<EPC>(mu, mean, cc);
    load_coef<EPC>(is, invstd, cc);
    const size_t stride = (size_t)gridDim.x * PL;
    for (size_t p = (size_t)blockIdx.x * PL + pl; p < npix; p += U * stride) {
      u32x4 ry[U], rg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ry[u] = rg[u] = (u32x4){0u, 0u, 0u, 0u};  // a zero gradient contributes nothing
        if (p + u * stride < npix) {
          ry[u] = *(const u32x4*)(y + (p + u * stride) * CS + cc * EPC);
          rg[u] = *(const u32x4*)(g + (p + u * stride) * GS + cc * EPC);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int wi = 0; wi < 4; ++wi)
#pragma unroll
          for (int h = 0; h < EPW; ++h) {
            const int e = wi * EPW + h;
            const float yv = Word<T>::get(ry[u][wi], h);
            const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Word<T>::get(rg[u][wi], h) : 0.f;
            s1[e] += dz;
            s2[e] = fmaf(dz, yv - mu[e], s2[e]);
          }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) s2[e] *= acc != nullptr ? 1.f : is[e];  // sum dz*yhat = invstd * sum dz*(y - mean)
  }
  wg_reduce_partials<T, EPC>(s1, s2, CS, CPC, PL, partial, red, acc);
}

// ---- pass 1 when the gradient arrives at TWICE the resolution (the activation went through nn.Upsample(scale_factor=2),
// unet.py:89): g[p] = the sum of the 2 x 2 gradients (spcl_upsample2x_backward's arithmetic: (a + b) + (c + d) in f32, rounded
// to T) is formed here, written for the apply pass, and folded into the BatchNorm sums in the same sweep -- the separate
// upsample2x_bwd pass and one read of its output disappear.  Thread geometry and summation order of
// bnrelu_bwd_reduce_lin_kernel (bit-identical sums).
template <typename T>
__global__ __launch_bounds__(256, 3) void bnrelu_bwd_reduce_up2_kernel(const T* __restrict__ y, const T* __restrict__ du,
                                                                    T* __restrict__ g, size_t npix, int W, int CS,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    float* __restrict__ partial /* [grid][2][CS] */,
                                                                    long long* __restrict__ acc = nullptr) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW, U = 2;  // (five 16-byte loads per position in flight: 2 x 5 of them)
  __shared__ float red[256][2 * EPC + 1];
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL) {
    float sc[EPC], sh[EPC], mu[EPC], is[EPC];
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
    load_coef<EPC>(mu, mean, cc);
    load_coef<EPC>(is, invstd, cc);
    const size_t stride = (size_t)gridDim.x * PL;
    const size_t rowstep = (size_t)2 * W * CS;  // elements between two rows of the fine gradient
    for (size_t p = (size_t)blockIdx.x * PL + pl; p < npix; p += U * stride) {
      u32x4 ry[U], ra[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t q = p + u * stride;
        ry[u] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 4; ++k) ra[u][k] = (u32x4){0u, 0u, 0u, 0u};
        if (q < npix) {
          const unsigned r = (unsigned)q / (unsigned)W, w = (unsigned)q - r * (unsigned)W;  // r = n H + h: fine rows 2 r, 2 r + 1 (32-bit: the host checks)
          const T* s = du + ((size_t)(2 * r) * (size_t)(2 * W) + 2 * w) * CS + cc * EPC;
          ry[u] = *(const u32x4*)(y + q * CS + cc * EPC);
          ra[u][0] = *(const u32x4*)s;
          ra[u][1] = *(const u32x4*)(s + CS);
          ra[u][2] = *(const u32x4*)(s + rowstep);
          ra[u][3] = *(const u32x4*)(s + rowstep + CS);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        u32x4 go;
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
          float gv[EPW];
#pragma unroll
          for (int h = 0; h < EPW; ++h) {
            const float acc = (Word<T>::get(ra[u][0][wi], h) + Word<T>::get(ra[u][1][wi], h)) +
                              (Word<T>::get(ra[u][2][wi], h) + Word<T>::get(ra[u][3][wi], h));
            gv[h] = acc;
          }
          go[wi] = Word<T>::make(gv);
#pragma unroll
          for (int h = 0; h < EPW; ++h) {
            const int e = wi * EPW + h;
            const float yv = Word<T>::get(ry[u][wi], h);
            const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Word<T>::get(go[wi], h) : 0.f;  // g as STORED
            s1[e] += dz;
            s2[e] = fmaf(dz, yv - mu[e], s2[e]);
          }
        }
        if (p + u * stride < npix) *(u32x4*)(g + (p + u * stride) * CS + cc * EPC) = go;
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) s2[e] *= acc != nullptr ? 1.f : is[e];
  }
  wg_reduce_partials<T, EPC>(s1, s2, CS, CPC, PL, partial, red, acc);
}

// ---- the same two passes for a BROADCAST gradient: g[n][p][c] = gb[n][c] for every pixel p of image n -- the gradient of a
// global average pool (projectors/heads.py:9-18: AdaptiveAvgPool2d((1, 1))), which the projector's backward then hands
// over as [N][CS] values instead of writing N x HW x CS of them.  Workgroup = (image, pixel split): the thread's gradient
// chunk is loaded once.
template <typename T>
__global__ __launch_bounds__(256, 5) void bnrelu_bwd_reduce_bcast_kernel(const T* __restrict__ y, const T* __restrict__ gb,
                                                                      int HW, int CS, int SPLIT,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      float* __restrict__ partial /* [grid][2][CS] */,
                                                                      long long* __restrict__ acc = nullptr) {
  // acc != null: the workgroup's sums are ADDED to that fixed-point block (bn_acc.hpp; sum dz (y - mean) NOT yet scaled by
  // invstd: the consumer does) instead of leaving a partial row for the finalize launch
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  __shared__ float red[256][2 * EPC + 1];
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  const int n = blockIdx.x / SPLIT, sp = blockIdx.x - n * SPLIT;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL) {
    float sc[EPC], sh[EPC], mu[EPC], is[EPC];
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
    load_coef<EPC>(mu, mean, cc);
    load_coef<EPC>(is, invstd, cc);
    const u32x4 rg = *(const u32x4*)(gb + (size_t)n * CS + cc * EPC);
    const T* yn = y + (size_t)n * HW * CS + cc * EPC;
    for (int p = sp * PL + pl; p < HW; p += SPLIT * PL) {
      const u32x4 ry = *(const u32x4*)(yn + (size_t)p * CS);
#pragma unroll
      for (int wi = 0; wi < 4; ++wi)
#pragma unroll
        for (int h = 0; h < EPW; ++h) {
          const int e = wi * EPW + h;
          const float yv = Word<T>::get(ry[wi], h);
          const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Word<T>::get(rg[wi], h) : 0.f;
          s1[e] += dz;
          s2[e] = fmaf(dz, yv - mu[e], s2[e]);
        }
    }
    if (acc == nullptr) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) s2[e] *= is[e];
    }
  }
  wg_reduce_partials<T, EPC>(s1, s2, CS, CPC, PL, partial, red, acc);
}

template <typename T>
__global__ __launch_bounds__(256, 6) void bnrelu_bwd_apply_bcast_kernel(const T* __restrict__ y, const T* __restrict__ gb,
                                                                     int HW, int CS, int SPLIT,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ ab, T* __restrict__ dy,
                                                                     BnAccBwd bw = BnAccBwd{}) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float sc[EPC], sh[EPC], A[EPC], B[EPC];
  if (bw.acc != nullptr) acc_coef_bwd<EPC>(bw, cc, A, B);  // (before any thread leaves: a barrier inside)
  if (pl >= PL) return;
  const int n = blockIdx.x / SPLIT, sp = blockIdx.x - n * SPLIT;
  load_coef<EPC>(sc, scale, cc);
  load_coef<EPC>(sh, shift, cc);
  if (bw.acc == nullptr) {
    load_coef<EPC>(A, ab, cc);
    load_coef<EPC>(B, ab + CS, cc);
  }
  const u32x4 rg = *(const u32x4*)(gb + (size_t)n * CS + cc * EPC);
  const size_t base = (size_t)n * HW * CS + cc * EPC;
  for (int p = sp * PL + pl; p < HW; p += SPLIT * PL) {
    const u32x4 ry = *(const u32x4*)(y + base + (size_t)p * CS);
    u32x4 out;
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
      float o[EPW];
#pragma unroll
      for (int h = 0; h < EPW; ++h) {
        const int e = wi * EPW + h;
        const float yv = Word<T>::get(ry[wi], h);
        const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Word<T>::get(rg[wi], h) : 0.f;
        o[h] = fmaf(sc[e], dz, fmaf(A[e], yv, B[e]));
      }
      out[wi] = Word<T>::make(o);
    }
    *(u32x4*)(dy + base + (size_t)p * CS) = out;
  }
}

// the window's dz for channel element (wi, h): pooled gradient routed to the first maximum, ReLU gate
template <typename T>
__device__ __forceinline__ void window_dz(const Window<T>& w, const u32x4* rg, bool has_g, u32x4 rdp, bool complete,
                                          int wi, int h, float sc, float sh, float* yk, float* dz) {
  float z[4];
  int best = 0;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    yk[k] = Word<T>::get(w.ry[k][wi], h);
    z[k] = w.valid[k] ? fmaf(sc, yk[k], sh) : -1.f;
    const float a = fmaxf(z[k], 0.f);
    if (k == 0) m = a;
    else if (a > m) { m = a; best = k; }
  }
  const float dp = complete ? Word<T>::get(rdp[wi], h) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float g = has_g ? Word<T>::get(rg[k][wi], h) : 0.f;
    g += (k == best) ? dp : 0.f;
    dz[k] = z[k] > 0.f ? g : 0.f;
  }
}

#ifndef SPCL_BWD_POOL_MAXW
#define SPCL_BWD_POOL_MAXW 0
#endif
#if SPCL_BWD_POOL_MAXW
#define BWD_POOL_WPE __attribute__((amdgpu_waves_per_eu(3, SPCL_BWD_POOL_MAXW)))
#else
#define BWD_POOL_WPE
#endif
// ---- pass 1 / pass 2 with pooling.  APPLY == false: partial sums;  true: dy
// FAST: H and W even, pooled gradient only (dact == null, dpool given) -- the training step's case: the run-time flags
// of the general form are scalar / exec branches around every load and store of the loop
// ACC: the apply pass derives A, B from a fixed-point accumulator block in its prologue (bn_acc.hpp) -- its own instantiation:
// the prologue's registers would take the plain form from five to four waves per SIMD (97 registers for 93)
template <typename T, bool APPLY, bool FAST, bool ACC = false>
__global__ __launch_bounds__(256, 3) BWD_POOL_WPE void bnrelu_bwd_pool_kernel(const T* __restrict__ y, const T* __restrict__ dact,
                                                              const T* __restrict__ dpool, int N, int H, int W, int CS,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ ab, float* __restrict__ partial,
                                                              T* __restrict__ dy, int GS = 0 /* dact pixel stride */,
                                                              BnAccBwd bw = BnAccBwd{}) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW;
  __shared__ float red[APPLY ? 1 : 256][2 * EPC + 1];
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  const int PH = (H + 1) / 2, PW = (W + 1) / 2, OH = H / 2, OW = W / 2;
  const bool has_g = !FAST && dact != nullptr;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
  float c0[EPC], c1[EPC];  // reduce: c0 = mean, c1 = invstd;  apply: c0 = A, c1 = B
  constexpr bool from_acc = APPLY && ACC;  // (apply pass: A, B derived from a fixed-point block, bn_acc.hpp)
  if (from_acc) acc_coef_bwd<EPC>(bw, cc, c0, c1);
  if (pl < PL) {
    float sc[EPC], sh[EPC];
    load_coef<EPC>(sc, scale, cc);
    load_coef<EPC>(sh, shift, cc);
    if (!from_acc) {
      load_coef<EPC>(c0, APPLY ? ab : mean, cc);
      load_coef<EPC>(c1, APPLY ? ab + CS : invstd, cc);
    }
    const int rows = N * PH;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const int n = r / PH, oy = r - n * PH;  // wave-uniform
      const size_t row0 = (size_t)n * H + 2 * oy;
      for (int ox = pl; ox < PW; ox += PL) {
        Window<T> w;
        w.template load<FAST>(y, row0, oy, ox, H, W, CS, cc);
        u32x4 rg[4], rdp = {0u, 0u, 0u, 0u};
        if (has_g) {
          const size_t g00 = (row0 * W + 2 * ox) * GS + cc * EPC;
          const size_t goff[4] = {g00, g00 + GS, g00 + (size_t)W * GS, g00 + (size_t)W * GS + GS};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            rg[k] = (u32x4){0u, 0u, 0u, 0u};
            if (w.valid[k]) rg[k] = *(const u32x4*)(dact + goff[k]);
          }
        }
        const bool complete = FAST || (dpool != nullptr && oy < OH && ox < OW);  // floor semantics of max_pool2d
        if (complete) rdp = *(const u32x4*)(dpool + (((size_t)n * OH + oy) * OW + ox) * CS + cc * EPC);
        // (all five requests of the window before any arithmetic: with the shorter FAST body the scheduler sank the pooled
        // gradient's load to its use behind the waits for the window -- a second, dependent round trip per iteration, + 2.5 us
        // on the 224^2 launch)
        if (APPLY && FAST) __builtin_amdgcn_sched_barrier(0);
        u32x4 out[4];
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
          float o[4][EPW];
#pragma unroll
          for (int h = 0; h < EPW; ++h) {
            const int e = wi * EPW + h;
            if (!APPLY && !has_g) {
              // pooled gradient only: dz is non-zero at the window's first maximum alone -> track (z, y) of it
              float zb = -1.f, yb = 0.f;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float yv = Word<T>::get(w.ry[k][wi], h);
                const float z = w.valid[k] ? fmaf(sc[e], yv, sh[e]) : -1.f;
                if (k == 0 || z > zb) { zb = z; yb = yv; }
              }
              const float dzb = (complete && zb > 0.f) ? Word<T>::get(rdp[wi], h) : 0.f;
              s1[e] += dzb;
              s2[e] = fmaf(dzb, yb - c0[e], s2[e]);
              continue;
            }
            if (APPLY && FAST) {
              // the training step's case, written for the instruction count (this kernel is the step's largest symbol: 4
              // launches, 79 us, its vector port 63 % busy beside 0.69 of the HBM peak): whole windows, pooled gradient only.
              // The first maximum of relu(z) in scan order as LANE MASKS -- z_k > (running maximum, >= 0) for k = 1 .. 3,
              // combined by scalar and-nots -- instead of a tracked index and four index compares: 30 vector instructions
              // per (window, channel) for 44.  Same values: dz_k = dp at that pixel if its z > 0, else 0; o = fma(sc, dz, X).
              float yk[4], z[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                yk[k] = Word<T>::get(w.ry[k][wi], h);
                z[k] = fmaf(sc[e], yk[k], sh[e]);
              }
              float m = fmaxf(z[0], 0.f);
              const bool s1 = z[1] > m;
              m = fmaxf(m, z[1]);
              const bool s2 = z[2] > m;
              m = fmaxf(m, z[2]);
              const bool s3 = z[3] > m;
              const bool b[4] = {z[0] > 0.f && !s1 && !s2 && !s3, s1 && !s2 && !s3, s2 && !s3, s3};
              const float dp = Word<T>::get(rdp[wi], h);
#pragma unroll
              for (int k = 0; k < 4; ++k) o[k][h] = fmaf(sc[e], b[k] ? dp : 0.f, fmaf(c0[e], yk[k], c1[e]));
              continue;
            }
            float yk[4], dz[4];
            window_dz<T>(w, rg, has_g, rdp, complete, wi, h, sc[e], sh[e], yk, dz);
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // invalid pixels carry dz == 0
              if (APPLY) o[k][h] = fmaf(sc[e], dz[k], fmaf(c0[e], yk[k], c1[e]));
              else {
                s1[e] += dz[k];
                s2[e] = fmaf(dz[k], yk[k] - c0[e], s2[e]);
              }
            }
          }
          if (APPLY) {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k][wi] = Word<T>::make(o[k]);
          }
        }
        if (APPLY) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (w.valid[k]) *(u32x4*)(dy + w.off[k]) = out[k];
        }
      }
    }
    if (!APPLY) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) s2[e] *= bw.acc != nullptr ? 1.f : c1[e];
    }
  }
  if (!APPLY) wg_reduce_partials<T, EPC>(s1, s2, CS, CPC, PL, partial, red, const_cast<long long*>(bw.acc));  // (reduction pass: bw.acc = the block to fill)
}

// dbeta = sum dz, dgamma = sum dz*yhat.  One workgroup per channel: threads stride over the workgroup partials, then
// a fixed-order butterfly and a fixed-order sum of the 4 waves -> deterministic.  Folded coefficients for pass 2:
//   training: A = -scale*invstd*dgamma/M,  B = -scale*dbeta/M - A*mean;   eval: A = B = 0
// `rs`: sub-rows per partial row (2; 11 with the image sums below).
// image3 (the first conv of a one-channel-image block; replaces the fused BN-backward + weight-gradient pass over y and g):
// with dy = scale dz + A y + B the weight gradient is
//   dW[c][t] = scale[c] S1[c][t] + A[c] sum_p y[p][c] img[p + t] + B[c] sum_p img[p + t]
// S1 = sum_p dz img are sub-rows 2 .. 10 (left by the dgrad that produced dz, conv_fast.hip MODE 4); y is LINEAR in the
// image, y[p][c] = sum_t' W[c][t'] img[p + t'], so the second sum is (W R)[c][t] with R[t'][t] = sum_p img[p + t'] img[p + t],
// the 9 x 9 autocorrelation of the zero-padded image batch (image_autocorr_kernel: 45 + 9 sums, 13 MB read) -- no pass
// over the activations at all.  W and img as the forward convolution saw them (bf16-rounded); the sums in double.
struct Image3Args {
  const float* acorr;   // [nacorr][64]: 45 upper-triangle R sums, then 9 image sums per partial row; null = off
  int nacorr;
  const float* w_oihw;  // [C][1][3][3] f32 master weights
  float* dw;            // [C][9]
};
__device__ __forceinline__ int acorr_index(int a, int b) {  // upper triangle of the symmetric 9 x 9 matrix, a <= b
  return a * 9 - a * (a - 1) / 2 + (b - a);
}
__global__ __launch_bounds__(256) void bnrelu_bwd_fin_kernel(const float* __restrict__ partial, int nwg, int C, int CS,
                                                             float M, int training, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ ab, float* __restrict__ zero_fill,
                                                             int nzero, int s2_centered, int rs, Image3Args im,
                                                             int transposed = 0) {
  __shared__ float red[11][4];
  __shared__ double racc4[4][64];
  __shared__ double racc[54];
  __shared__ float coef[2];
  // image3 launches may run THREE workgroups per channel (grid 3 CS), each with three of the nine tap sums (and, all of them,
  // the two BatchNorm sums the coefficients come from): 5 of the 11 sub-row walks per workgroup instead of 11 on 16 CUs.  The
  // walk is instantiated for 9 and for 3 taps with a wave-uniform first tap (no load sits behind a branch of its own).
  const int c = blockIdx.x % CS, grp = blockIdx.x / CS;
  const bool split = gridDim.x == 3u * (unsigned)CS;
  const int t0 = split ? 3 * grp : 0, ntap = split ? 3 : 9;
  if (blockIdx.x == 0)  // scratch the next launch wants zeroed (the image-wgrad pass's zero row): saves a memset launch
    for (int z = threadIdx.x; z < nzero; z += 256) zero_fill[z] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the channel's coefficients are requested WITH the partials (one memory round trip instead of a second, dependent
  // one after the reduction: this kernel is pure latency, ten launches per step)
  const float c_invstd = invstd[c], c_scale = scale[c], c_mean = mean[c];
  const bool img3 = im.acorr != nullptr;
  __shared__ float wsh[9];
  if (img3 && threadIdx.x < 9 && c < C)  // the channel's nine weights as the forward MFMA saw them, requested up front
    wsh[threadIdx.x] = bf16_to_f32(f32_to_bf16(im.w_oihw[c * 9 + threadIdx.x]));
  float s1 = 0.f, s2 = 0.f;
  float st[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) st[t] = 0.f;
  auto walk = [&](auto ntap_c) {
    constexpr int NTAP = decltype(ntap_c)::value;  // 0: no tap sums (not the image3 path)
    if (transposed) {  // [sub-row][channel][row]: consecutive threads read consecutive floats (the row-major walk below costs
                       // one cache line per lane and load: 11 k line requests per workgroup on the image3 path)
#pragma unroll 4
      for (int w = threadIdx.x; w < nwg; w += 256) {
        s1 += partial[((size_t)0 * CS + c) * nwg + w];
        s2 += partial[((size_t)1 * CS + c) * nwg + w];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) st[t] += partial[((size_t)(2 + t0 + t) * CS + c) * nwg + w];
      }
    } else {
#pragma unroll 4
      for (int w = threadIdx.x; w < nwg; w += 256) {
        s1 += partial[((size_t)w * rs + 0) * CS + c];
        s2 += partial[((size_t)w * rs + 1) * CS + c];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) st[t] += partial[((size_t)w * rs + 2 + t0 + t) * CS + c];
      }
    }
  };
  if (!img3) walk(std::integral_constant<int, 0>{});
  else if (split) walk(std::integral_constant<int, 3>{});
  else walk(std::integral_constant<int, 9>{});
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  if (img3) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {  // (st[t] holds tap t0 + t; the entries beyond ntap are zero and land on unused rows)
      st[t] = wave_sum(st[t]);
      if (lane == 0 && t < ntap) red[2 + t0 + t][wave] = st[t];
    }
    {  // the autocorrelation / image sums over the partial rows of image_autocorr_kernel, in double: wave p takes rows
       // p, p + 4, ... (eight loads in flight per lane), the four waves are added in fixed order below
      double a = 0.0;
#pragma unroll 8
      for (int w = wave; w < im.nacorr; w += 4) a += (double)im.acorr[(size_t)w * 64 + lane];
      racc4[wave][lane] = a;
    }
  }
  __syncthreads();
  if (img3 && threadIdx.x < 54)
    racc[threadIdx.x] = (racc4[0][threadIdx.x] + racc4[1][threadIdx.x]) + (racc4[2][threadIdx.x] + racc4[3][threadIdx.x]);
  if (threadIdx.x == 0) {
    s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (s2_centered) s2 *= c_invstd;  // rows held sum dz (y - mean): dgamma = invstd * that
    if (c < C && grp == 0) {
      dbeta[c] = s1;
      dgamma[c] = s2;
    }
    float A = 0.f, B = 0.f;
    if (training) {
      A = -c_scale * c_invstd * (s2 / M);
      B = -c_scale * (s1 / M) - A * c_mean;
    }
    if (grp == 0) {
      ab[c] = A;
      ab[CS + c] = B;
    }
    coef[0] = A;
    coef[1] = B;
  }
  if (!img3) return;
  __syncthreads();
  __shared__ double wrp[9][9];
  if (threadIdx.x < 81) {  // (W R)[c][t] = sum_u W[c][u] R[u][t]: the 81 products side by side, nine fixed-order sums below
    const int t = threadIdx.x / 9, u = threadIdx.x - 9 * t;
    wrp[t][u] = (double)wsh[u] * racc[u <= t ? acorr_index(u, t) : acorr_index(t, u)];
  }
  __syncthreads();
  if ((int)threadIdx.x >= t0 && (int)threadIdx.x < t0 + ntap && c < C) {
    const int t = threadIdx.x;
    const double S1 = (double)((red[2 + t][0] + red[2 + t][1]) + (red[2 + t][2] + red[2 + t][3]));
    double wr = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) wr += wrp[t][u];
    im.dw[c * 9 + t] = (float)((double)c_scale * S1 + (double)coef[0] * wr + (double)coef[1] * racc[45 + t]);
  }
}

__global__ __launch_bounds__(256) void image_autocorr_kernel(const float* __restrict__ img, int H, int W,
                                                             float* __restrict__ out) {
  image_autocorr_body(img, H, W, out, blockIdx.x);  // (image_acorr.hpp)
}

// rows [nrows][2][CS] of per-tile partial sums (the dgrad epilogue of conv_fast.hip, MODE 2) folded G at a time, so that
// the per-channel final sum never walks more than BWD_MAX_WG rows
constexpr int ACORR_FOLD = 16;
__global__ __launch_bounds__(256) void bwd_rows_group_kernel(const float* __restrict__ rows, int nrows, int G, int CS,
                                                             float* __restrict__ out, int rs /* sub-rows: 2 or 11 */,
                                                             int nwg = 0x7fffffff,
                                                             const float* __restrict__ acorr_in = nullptr,
                                                             int nacorr = 0, float* __restrict__ acorr_out = nullptr,
                                                             int transposed = 0) {
  if ((int)blockIdx.x >= nwg) {  // image3: the autocorrelation's per-band rows folded to ACORR_FOLD rows in the same launch
    const int j = blockIdx.x - nwg, k = threadIdx.x;
    if (k < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int w = j; w < nacorr; w += ACORR_FOLD) s += acorr_in[(size_t)w * 64 + k];
      acorr_out[(size_t)j * 64 + k] = s;
    }
    return;
  }
  const int w = blockIdx.x;
  for (int o = threadIdx.x; o < rs * CS; o += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < G; ++k) {
      const int r = w * G + k;
      if (r < nrows) s += rows[(size_t)r * rs * CS + o];
    }
    if (transposed) out[(size_t)o * (nwg == 0x7fffffff ? (int)gridDim.x : nwg) + w] = s;
    else out[(size_t)w * rs * CS + o] = s;
  }
}

// the same rows ADDED to a fixed-point accumulator block (bn_acc.hpp) instead: G rows per workgroup (few workgroups: the adds
// to one address serialise), then the apply pass derives its coefficients from the block -- no finalize launch.  Thread o ->
// word o of the block's row (channel o / 4, sum (o / 2) % 2, limb o % 2; the two limb threads of a sum both form it).
__global__ __launch_bounds__(256) void bwd_rows_acc_kernel(const float* __restrict__ rows, int nrows, int G, int CS,
                                                           long long* __restrict__ acc) {
  // the 256 / (4 CS) thread groups of a word split the workgroup's G rows among them and meet in LDS (fixed order)
  __shared__ float part[256];
  const int w = blockIdx.x, NW4 = 4 * CS;
  const int NP = NW4 >= 256 ? 1 : 256 / NW4;
  for (int o0 = 0; o0 < NW4; o0 += 256) {
    const int o = o0 + (int)(threadIdx.x % (NW4 < 256 ? NW4 : 256)), p = NW4 < 256 ? (int)threadIdx.x / NW4 : 0;
    const int c = o >> 2, which = (o >> 1) & 1, limb = o & 1;
    float s = 0.f;
    if (o < NW4 && p < NP) {
#pragma unroll 8
      for (int k = p; k < G; k += NP) {
        const int r = w * G + k;
        if (r < nrows) s += rows[((size_t)r * 2 + which) * CS + c];
      }
    }
    if (NP > 1) {
      part[threadIdx.x] = s;
      __syncthreads();
      if (p == 0 && o < NW4) {
        for (int q = 1; q < NP; ++q) s += part[q * NW4 + o];
      }
      __syncthreads();
    }
    if (p == 0 && o < NW4) bn_acc_add_word(acc, CS, w & (BN_ACC_REPLICAS - 1), c, which, limb, s);
  }
}

// ---- pass 2, no pooling (linear): dy = scale*dz + A*y + B
template <typename T>
__global__ __launch_bounds__(256, 6) void bnrelu_bwd_apply_lin_kernel(const T* __restrict__ y, const T* __restrict__ g,
                                                                   size_t npix, int CS,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ ab, T* __restrict__ dy,
                                                                   int GS, BnAccBwd bw = BnAccBwd{}) {
  constexpr int EPC = Chunk<T>::EPC, EPW = Word<T>::EPW, U = STREAM_UNROLL;
  const int CPC = CS / EPC, PL = 256 / CPC;
  const int cc = threadIdx.x % CPC, pl = threadIdx.x / CPC;
  float sc[EPC], sh[EPC], A[EPC], B[EPC];
  if (bw.acc != nullptr) acc_coef_bwd<EPC>(bw, cc, A, B);  // (before any thread leaves: a barrier inside)
  if (pl >= PL) return;
  load_coef<EPC>(sc, scale, cc);
  load_coef<EPC>(sh, shift, cc);
  if (bw.acc == nullptr) {
    load_coef<EPC>(A, ab, cc);
    load_coef<EPC>(B, ab + CS, cc);
  }
  const size_t stride = (size_t)gridDim.x * PL;
  for (size_t p = (size_t)blockIdx.x * PL + pl; p < npix; p += U * stride) {
    u32x4 ry[U], rg[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p + u * stride < npix) {
        ry[u] = *(const u32x4*)(y + (p + u * stride) * CS + cc * EPC);
        rg[u] = *(const u32x4*)(g + (p + u * stride) * GS + cc * EPC);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p + u * stride >= npix) break;
      u32x4 out;
#pragma unroll
      for (int wi = 0; wi < 4; ++wi) {
        float o[EPW];
#pragma unroll
        for (int h = 0; h < EPW; ++h) {
          const int e = wi * EPW + h;
          const float yv = Word<T>::get(ry[u][wi], h);
          const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Word<T>::get(rg[u][wi], h) : 0.f;
          o[h] = fmaf(sc[e], dz, fmaf(A[e], yv, B[e]));
        }
        out[wi] = Word<T>::make(o);
      }
      *(u32x4*)(dy + (p + u * stride) * CS + cc * EPC) = out;
    }
  }
}

// ---- pass 2 of the FIRST conv of a one-channel image block, fused with that conv's weight gradient.  The layer's dy
// feeds nothing but dW (the image needs no gradient), so it is formed in registers and never stored:
//   dW[co][0][ky][kx] = sum_p dy[p][co] * x[p + (ky-1, kx-1)]      (zero outside the image)
// A lane owns FOUR channels of a pixel (9 x 4 f32 accumulators: small enough for 5-6 waves per SIMD, which is what
// hides the load latency of this stream) and walks the image row by row (the image / row split is wave-uniform scalar
// work).  The nine image values come from clamped addresses times 0/1 factors, so that no load sits in a divergent
// branch.  Per-workgroup partial rows [9][CS] in a fixed order (deterministic), finished by image_wgrad_final_kernel.
template <typename T> struct Quad;  // four consecutive channels of one pixel
template <> struct Quad<float> {
  typedef u32x4 raw_t;
  static __device__ __forceinline__ float get(raw_t r, int e) { return __uint_as_float(r[e]); }
};
template <> struct Quad<bf16_t> {
  typedef __attribute__((ext_vector_type(2))) uint32_t raw_t;
  static __device__ __forceinline__ float get(raw_t r, int e) {
    return __uint_as_float((e & 1) ? (r[e >> 1] & 0xffff0000u) : (r[e >> 1] << 16));
  }
};

template <typename T>
__global__ __launch_bounds__(256, 5) void bnrelu_bwd_image_wgrad_kernel(const T* __restrict__ y, const T* __restrict__ g,
                                                                     const float* __restrict__ img, int N, int H,
                                                                     int W, int CS, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ ab,
                                                                     const float* __restrict__ zrow /* W zeros */,
                                                                     float* __restrict__ wpart /* [grid][9][CS] */) {
  typedef typename Quad<T>::raw_t raw_t;
  constexpr int U = 2;
  extern __shared__ float redw[];  // [4 waves][9][CS]
  const int QPP = CS / 4, PL = 256 / QPP;  // quads per pixel: a power of two <= 64 (checked by the launcher)
  const int q = threadIdx.x % QPP, pl = threadIdx.x / QPP;
  float acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
  float sc[4], sh[4], A[4], B[4];
  load_coef<4>(sc, scale, q);
  load_coef<4>(sh, shift, q);
  load_coef<4>(A, ab, q);
  load_coef<4>(B, ab + CS, q);
  const int rows = N * H;
  const unsigned lane_off = q * 4 * sizeof(T), pix_bytes = CS * sizeof(T);
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / H, h = r - n * H;  // wave-uniform
    // the three image rows of the taps; past the top / bottom edge a row of zeros (scalar select, no lane work)
    const float* xr[3];
    xr[1] = img + (size_t)r * W;
    xr[0] = h > 0 ? xr[1] - W : zrow;
    xr[2] = h + 1 < H ? xr[1] + W : zrow;
    const char* yr = (const char*)(y + (size_t)r * W * CS);
    const char* gr = (const char*)(g + (size_t)r * W * CS);
    for (int ox0 = pl; ox0 < W; ox0 += U * PL) {
      raw_t ry[U], rg[U];
      float xv[U][9];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ox = ox0 + u * PL;
        ok[u] = ox < W;
        const int c1 = min(ox, W - 1), c0 = max(c1 - 1, 0), c2 = min(c1 + 1, W - 1);
        const unsigned off = (unsigned)c1 * pix_bytes + lane_off;
        ry[u] = *(const raw_t*)(yr + off);
        rg[u] = *(const raw_t*)(gr + off);
        const bool left = ox > 0, right = ox + 1 < W;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const char* xk = (const char*)xr[k];  // uniform base + 32-bit lane offset
          const float a0 = *(const float*)(xk + (unsigned)(c0 * 4)), a1 = *(const float*)(xk + (unsigned)(c1 * 4)),
                      a2 = *(const float*)(xk + (unsigned)(c2 * 4));
          xv[u][3 * k + 0] = left ? a0 : 0.f;
          xv[u][3 * k + 1] = a1;
          xv[u][3 * k + 2] = right ? a2 : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yv = Quad<T>::get(ry[u], e);
          const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? Quad<T>::get(rg[u], e) : 0.f;
          float dyv = fmaf(sc[e], dz, fmaf(A[e], yv, B[e]));
          dyv = ok[u] ? dyv : 0.f;  // lanes past the row end
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t][e] = fmaf(dyv, xv[u][t], acc[t][e]);
        }
    }
  }
  // lanes of one quad sit QPP apart: butterfly over the wave (all 36 values per step, so that the shuffles pipeline),
  // then a fixed-order sum of the 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o >= QPP; o >>= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] += __shfl_xor(acc[t][e], o, 64);
  }
  if (lane < QPP) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) redw[(wave * 9 + t) * CS + lane * 4 + e] = acc[t][e];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 9 * CS; o += 256)
    wpart[(size_t)blockIdx.x * 9 * CS + o] =
        (redw[o] + redw[9 * CS + o]) + (redw[2 * 9 * CS + o] + redw[3 * 9 * CS + o]);
}

// dw[co][0][t] = sum over the workgroup rows of wpart[wg][t][co]; one workgroup per tap, fixed order
__global__ __launch_bounds__(256) void image_wgrad_final_kernel(const float* __restrict__ wpart, int nwg, int C, int CS,
                                                                float* __restrict__ dw) {
  __shared__ float red[256];
  const int t = blockIdx.x;
  const int RL = 256 / CS;  // row lanes (CS <= 256)
  const int c = threadIdx.x % CS, rl = threadIdx.x / CS;
  float s = 0.f;
  if (rl < RL) {
#pragma unroll 8
    for (int w = rl; w < nwg; w += RL) s += wpart[((size_t)w * 9 + t) * CS + c];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < CS && threadIdx.x < C) {
    float tot = 0.f;
    for (int k = 0; k < RL; ++k) tot += red[k * CS + threadIdx.x];
    dw[threadIdx.x * 9 + t] = tot;
  }
}

// enough workgroups to fill the chip several times over, never more than the positions need
static int stream_grid(size_t positions, int PL, int unroll, int cap) {
  size_t g = (positions + (size_t)PL * unroll - 1) / ((size_t)PL * unroll);
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// Pixel strides (in elements) of the activation written by the forward / of the activation gradient read by the backward when
// they are channel slices of a wider tensor (the `_strided` entry points set them around their call; 0 = dense, CS).
static thread_local int tl_act_stride = 0, tl_dact_stride = 0;
static thread_local const void* tl_up2_src = nullptr;  // spcl_bnrelu_backward_up2: the fine gradient; `dact` is then WRITTEN
static thread_local bool tl_acc_fill = false;  // spcl_bnrelu_backward_fill_acc: this call's reduction pass fills the block

template <typename T>
static int bnrelu_fwd_launch(const void* y, int N, int H, int W, int CS, const float* scale, const float* shift,
                             void* act, void* pool, hipStream_t st, const BnAccFwd* bnp = nullptr) {
  constexpr int EPC = Chunk<T>::EPC;
  const BnAccFwd bn = bnp != nullptr ? *bnp : BnAccFwd{};
  // (coefficients derived in every workgroup's prologue: fewer, longer-lived workgroups amortise it)
  static const int acc_wg = lab_env("SPCL_ACC_STREAM_WG", 768);
  const int STREAM_MAX_WG = bnp != nullptr ? acc_wg : spcl::STREAM_MAX_WG;
  const int PL = 256 / (CS / EPC);
  const int AS = tl_act_stride > 0 ? tl_act_stride : CS;
  const double tb = (double)N * H * W * CS * sizeof(T);  // bytes of one full-resolution tensor
  prof_cost(tb * (1.0 + (act != nullptr ? 1.0 : 0.0) + (pool != nullptr ? 0.25 : 0.0)), 0.0);
  if (pool != nullptr) {
    const int rows = N * ((H + 1) / 2);
    if (H % 2 == 0 && W % 2 == 0) {
      SPCL_LAUNCH((bnrelu_fwd_pool_kernel<T, true>), dim3(rows < STREAM_MAX_WG ? rows : STREAM_MAX_WG), dim3(256), 0, st, (const T*)y, N, H,
                       W, CS, scale, shift, (T*)act, (T*)pool, AS, bn);
    } else {
      SPCL_LAUNCH((bnrelu_fwd_pool_kernel<T, false>), dim3(rows < STREAM_MAX_WG ? rows : STREAM_MAX_WG), dim3(256), 0, st, (const T*)y, N, H,
                       W, CS, scale, shift, (T*)act, (T*)pool, AS, bn);
    }
  } else {
    const size_t npix = (size_t)N * H * W;
    SPCL_LAUNCH((bnrelu_fwd_lin_kernel<T>), dim3(stream_grid(npix, PL, STREAM_UNROLL, STREAM_MAX_WG)), dim3(256), 0, st,
                       (const T*)y, npix, CS, scale, shift, (T*)act, AS, bn);
  }
  return 0;
}

template <typename T>
static int bnrelu_bwd_launch(const void* y, const void* dact, const void* dpool, int N, int H, int W, int C, int CS,
                             const float* mean, const float* invstd, const float* scale, const float* shift,
                             int training, float* ws, float* dgamma, float* dbeta, void* dy, hipStream_t st,
                             const float* img = nullptr, float* dw = nullptr, const float* rows = nullptr,
                             int nrows = 0, bool bcast = false /* dact is [N][CS], one value per image and channel */,
                             long long* acc = nullptr /* the sums as a fixed-point block (bn_acc.hpp): no finalize launch */) {
  constexpr int EPC = Chunk<T>::EPC;
  // acc: either FILLED already by the dgrad that produced dact / dpool (lin / pool apply), or -- bcast -- filled by this
  // call's reduction pass; the apply kernel derives A, B in its prologue, its first workgroup writes dgamma / dbeta
  const BnAccBwd bw = acc != nullptr ? BnAccBwd{acc, mean /* = st[0]: mean, invstd, scale, shift are [4][CS] */, dgamma, dbeta,
                                                (float)((size_t)N * H * W), training, C, CS}
                                     : BnAccBwd{};
  static const int acc_wg = lab_env("SPCL_ACC_STREAM_WG", 768);
  const int STREAM_MAX_WG = acc != nullptr ? acc_wg : spcl::STREAM_MAX_WG;  // (see bnrelu_fwd_launch)
  const bool pool = dpool != nullptr;
  const bool fill = acc != nullptr && tl_acc_fill;  // the reduction pass below ADDS to the block (<= 768 workgroups: ~100 adds per address)
  const int RED_MAX_WG = fill ? STREAM_MAX_WG : BWD_MAX_WG;
  const int GS = tl_dact_stride > 0 ? tl_dact_stride : CS;
  constexpr int rs = 2;  // sub-rows of a partial row (the image3 path has eleven: spcl_bnrelu_backward_rows_image3)
  const Image3Args im3{nullptr, 0, nullptr, nullptr};
  spcl_wgrad_tail* tail = img != nullptr ? take_tail_capture() : nullptr;
  const int PL = 256 / (CS / EPC);
  const size_t npix = (size_t)N * H * W;
  const int prows = N * ((H + 1) / 2);
  float* partial = ws;                           // [nwg][2][CS]  (ws may be null with `acc`: neither is touched then)
  float* ab = ws != nullptr ? ws + (size_t)BWD_MAX_WG * 2 * CS : nullptr;  // [2][CS]: folded BN-backward coefficients
  const float M = (float)npix;
  int nwg, bsplit = 1;
  const double tb = (double)npix * CS * sizeof(T);  // bytes of one full-resolution tensor
  const double gb = (dact != nullptr ? tb : 0.0) + (pool ? 0.25 * tb : 0.0);
  prof_cost(tb + gb, 0.0);
  const float* fin_src = partial;
  int fin_transposed = 0;
  if (acc != nullptr && !bcast && !fill) {
    nwg = 0;  // (the block is complete: nothing to reduce, nothing to finalize)
  } else if (rows != nullptr && fill) {  // ... and go into the block: <= 512 workgroups add, the apply pass derives
    const int G = (nrows + 511) / 512;
    nwg = (nrows + G - 1) / G;
    SPCL_LAUNCH(bwd_rows_acc_kernel, dim3(nwg), dim3(256), 0, st, rows, nrows, G, CS, acc);
  } else if (rows != nullptr) {  // the partial sums came with the dgrad that produced dact (per conv tile, centred s2)
    static const int fin_rows = lab_env("SPCL_BWD_FIN_MAX_ROWS", BWD_MAX_WG);
    if (nrows <= fin_rows) {
      fin_src = rows;
      nwg = nrows;
    } else {
      const int G = (nrows + BWD_MAX_WG - 1) / BWD_MAX_WG;
      nwg = (nrows + G - 1) / G;
      // (folded rows written transposed, [sub-row][channel][row]: the final kernel's loads are then coalesced)
      SPCL_LAUNCH(bwd_rows_group_kernel, dim3(nwg), dim3(256), 0, st, rows, nrows, G, CS, partial, rs, nwg,
                  (const float*)nullptr, 0, (float*)nullptr, 1);
      fin_transposed = 1;
    }
  } else if (pool) {
    nwg = prows < RED_MAX_WG ? prows : RED_MAX_WG;
    const BnAccBwd rbw = fill ? bw : BnAccBwd{};
    if (H % 2 == 0 && W % 2 == 0 && dact == nullptr && dpool != nullptr) {
      SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, false, true>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)nullptr, partial,
                       (T*)nullptr, 0, rbw);
    } else {
      SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, false, false>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact,
                       (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)nullptr, partial,
                       (T*)nullptr, GS, rbw);
    }
  } else if (bcast) {
    // pixel splits per image: every workgroup leaves a row of 2 CS partial sums that bnrelu_bwd_fin_kernel walks one cache
    // line per lane -- one pass of 8 pixels per workgroup (25 splits of a 14 x 14 map: 1 600 rows) made that kernel 9.5 us
    // of the step; SPCL_BCAST_SPLIT_MAX splits -> N x that many rows
    static const int env_split = lab_env("SPCL_BCAST_SPLIT_MAX", 4);
    bsplit = (H * W + PL - 1) / PL;
    if (env_split > 0 && bsplit > env_split) bsplit = env_split;
    while (bsplit > 1 && N * bsplit > BWD_MAX_WG) bsplit = (bsplit + 1) / 2;
    nwg = N * bsplit;
    prof_cost(tb, 0.0);
    SPCL_LAUNCH((bnrelu_bwd_reduce_bcast_kernel<T>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact, H * W, CS,
                bsplit, mean, invstd, scale, shift, partial, acc);
  } else if (tl_up2_src != nullptr) {
    nwg = stream_grid(npix, PL, STREAM_UNROLL, RED_MAX_WG);
    prof_cost(tb * 6.0, 0.0);
    SPCL_LAUNCH((bnrelu_bwd_reduce_up2_kernel<T>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)tl_up2_src, (T*)dact,
                npix, W, CS, mean, invstd, scale, shift, partial, fill ? acc : (long long*)nullptr);
  } else {
    nwg = stream_grid(npix, PL, STREAM_UNROLL, RED_MAX_WG);
    SPCL_LAUNCH((bnrelu_bwd_reduce_lin_kernel<T>), dim3(nwg), dim3(256), 0, st, (const T*)y, (const T*)dact, npix,
                       CS, mean, invstd, scale, shift, partial, GS, fill ? acc : (long long*)nullptr);
  }
  float* zrow = ab != nullptr ? ab + 2 * CS + (size_t)IMG_WGRAD_WG * 9 * CS : nullptr;  // [W] zeros (image-wgrad pass only, see below)
  if (acc == nullptr)
    SPCL_LAUNCH(bnrelu_bwd_fin_kernel, dim3(CS), dim3(256), 0, st, fin_src, nwg, C, CS,
                       M, training, mean, invstd, scale, dgamma, dbeta, ab, img != nullptr ? zrow : (float*)nullptr,
                       img != nullptr ? W : 0, rows != nullptr ? 1 : 0, rs, im3, fin_transposed);
  prof_cost(2.0 * tb + gb, 0.0);
  if (pool) {
    if (H % 2 == 0 && W % 2 == 0 && dact == nullptr && dpool != nullptr) {
      if (acc != nullptr)
        SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, true, true, true>), dim3(prows < STREAM_MAX_WG ? prows : STREAM_MAX_WG), dim3(256), 0, st,
                    (const T*)y, (const T*)dact, (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)ab,
                    (float*)nullptr, (T*)dy, 0, bw);
      else
        SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, true, true>), dim3(prows < STREAM_MAX_WG ? prows : STREAM_MAX_WG), dim3(256), 0, st, (const T*)y,
                       (const T*)dact, (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)ab,
                       (float*)nullptr, (T*)dy);
    } else {
      if (acc != nullptr)
        SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, true, false, true>), dim3(prows < STREAM_MAX_WG ? prows : STREAM_MAX_WG), dim3(256), 0, st,
                    (const T*)y, (const T*)dact, (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)ab,
                    (float*)nullptr, (T*)dy, GS, bw);
      else
        SPCL_LAUNCH((bnrelu_bwd_pool_kernel<T, true, false>), dim3(prows < STREAM_MAX_WG ? prows : STREAM_MAX_WG), dim3(256), 0, st, (const T*)y,
                       (const T*)dact, (const T*)dpool, N, H, W, CS, mean, invstd, scale, shift, (const float*)ab,
                       (float*)nullptr, (T*)dy, GS);
    }
  } else if (img != nullptr) {  // first conv of a one-channel image block: dy is consumed in registers by its dW
    float* wpart = ab + 2 * CS;  // [IMG_WGRAD_WG][9][CS];  zrow = the image row above / below the image
    static const int want = lab_env("SPCL_IMGWG_WG", 1280);  // 5 resident per CU
    const int cap = want < IMG_WGRAD_WG ? (want > 0 ? want : 1) : IMG_WGRAD_WG;
    const int g = N * H < cap ? N * H : cap;
    prof_cost(2.0 * tb + (double)npix * 4, 2.0 * 9 * npix * C);
    SPCL_LAUNCH((bnrelu_bwd_image_wgrad_kernel<T>), dim3(g), dim3(256), 4 * 9 * CS * sizeof(float), st, (const T*)y,
                       (const T*)dact, img, N, H, W, CS, scale, shift, (const float*)ab, (const float*)zrow, wpart);
    if (tail != nullptr) {  // the nine taps' final sums ride in the batched weight-gradient reduction
      tail->partial = wpart; tail->dw = dw; tail->kind = 1; tail->nsplit = g; tail->nblk_ci = tail->nblk_co = 1;
      tail->CIB = 1; tail->COB = CS; tail->Cin = 1; tail->Cout = C;
    } else {
      SPCL_LAUNCH(image_wgrad_final_kernel, dim3(9), dim3(256), 0, st, (const float*)wpart, g, C, CS, dw);
    }
  } else if (bcast) {
    prof_cost(2.0 * tb, 0.0);
    SPCL_LAUNCH((bnrelu_bwd_apply_bcast_kernel<T>), dim3(N * bsplit), dim3(256), 0, st, (const T*)y, (const T*)dact, H * W,
                CS, bsplit, scale, shift, (const float*)ab, (T*)dy, bw);
  } else {
    SPCL_LAUNCH((bnrelu_bwd_apply_lin_kernel<T>), dim3(stream_grid(npix, PL, STREAM_UNROLL, STREAM_MAX_WG)), dim3(256), 0,
                       st, (const T*)y, (const T*)dact, npix, CS, scale, shift, (const float*)ab, (T*)dy, GS, bw);
  }
  return 0;
}

}  // namespace spcl

using namespace spcl;

// Self-resetting tickets of the one-launch two-level reduction (one per 16-channel block).
//  * One zeroed buffer PER (device, stream): launches on one stream are serialised, so two reductions can never interleave
//    their ticket counts; launches on different streams (two models, a second epocher) get different words (the
//    buffer used to be per device only).  Created under a mutex at the first call on that stream OUTSIDE a capture (an
//    allocation is not capturable; the eager warm-up steps of stepgraph.py run on the capture stream first); a stream first
//    seen during a capture, or more than BN_TICKET_STREAMS streams, fall back to the two launches.
//  * Memory-ordering assumption, stated: the hand-off is the form MI355X_MICROARCH.md measures as valid on gfx950 / ROCm 7.2
//    ("Workgroup dispatch, XCD placement & inter-workgroup visibility", table of sc1 hand-offs, first row) and not an
//    architectural guarantee -- every handed-off byte leaves by an 8-byte agent-scope (sc1, written-through) store, the
//    storing wave waits for their acknowledgement (s_waitcnt 0), a workgroup barrier, ONE lane takes a relaxed agent-scope
//    ticket, the workgroup whose ticket is the last loads every row with agent-scope (sc1) loads after a barrier behind
//    the returned add.  No L2 write-back fence (a release would cost ~1.7-6.5 us per workgroup, more than the launch it
//    saves).  SPCL_BN_ONE_LAUNCH=0 keeps the two launches (the A/B switch and the fallback on other parts).
constexpr int BN_TICKETS = 64;
constexpr int BN_TICKET_STREAMS = 32;
static unsigned* bn_tickets(hipStream_t st) {
  static const bool on = !(lab_env("SPCL_BN_ONE_LAUNCH", 1) == 0);
  if (!on) return nullptr;
  struct Slot { int dev; hipStream_t st; unsigned* buf; };
  static Slot slots[BN_TICKET_STREAMS];
  static int nslots = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nslots; ++i)
    if (slots[i].dev == dev && slots[i].st == st) return slots[i].buf;
  if (nslots == BN_TICKET_STREAMS) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
  unsigned* p = nullptr;
  if (hipMalloc(&p, BN_TICKETS * sizeof(unsigned)) != hipSuccess) return nullptr;
  if (hipMemsetAsync(p, 0, BN_TICKETS * sizeof(unsigned), st) != hipSuccess) {  // in stream order ahead of its only users
    (void)hipFree(p);
    return nullptr;
  }
  slots[nslots++] = Slot{dev, st, p};
  return p;
}

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
    SPCL_LAUNCH((bn_reduce_kernel<float, true, 16>), dim3(CS / 16, 1), dim3(ntiles > 256 ? 1024 : 256), 0, st, stats,
                       ntiles, ntiles, C, CS, (double*)nullptr, f, (unsigned*)nullptr);
  } else {
    double* partial = (double*)(stats + (size_t)ntiles * 3 * CS);  // spcl_bn_stats_elems reserves it
    unsigned* tickets = bn_tickets(st);
    if (tickets != nullptr && CS / 16 <= BN_TICKETS) {
      SPCL_LAUNCH((bn_reduce_kernel<float, false, 16>), dim3(CS / 16, groups), dim3(256), 0, st, stats, ntiles,
                  BN_GROUP_TILES, C, CS, partial, f, tickets);
    } else {
      SPCL_LAUNCH((bn_reduce_kernel<float, false, 16>), dim3(CS / 16, groups), dim3(256), 0, st, stats, ntiles,
                  BN_GROUP_TILES, C, CS, partial, f, (unsigned*)nullptr);
      SPCL_LAUNCH((bn_reduce_kernel<double, true, 16>), dim3(CS / 16, 1), dim3(groups > 64 ? 1024 : 256), 0, st,
                  (const double*)partial, groups, groups, C, CS, (double*)nullptr, f, (unsigned*)nullptr);
    }
  }
  SPCL_LAUNCH_CHECK("bn_finalize");
  return SPCL_OK;
}

extern "C" int spcl_bn_eval_affine(int C, int CS, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                   float* shift, void* stream) {
  SPCL_CHECK_ARG(gamma && beta && running_mean && running_var && mean && invstd && scale && shift,
                 "bn_eval_affine: null pointer");
  SPCL_LAUNCH(bn_eval_affine_kernel, dim3(cdiv(CS, 256)), dim3(256), 0, (hipStream_t)stream, C, CS, gamma, beta,
                     running_mean, running_var, eps, mean, invstd, scale, shift);
  SPCL_LAUNCH_CHECK("bn_eval_affine");
  return SPCL_OK;
}

extern "C" int spcl_bn_eval_affine_multi(const spcl_bn_eval_item* items, int n, void* stream) {
  SPCL_CHECK_ARG(items && n >= 1 && n <= SPCL_BN_EVAL_MAX, "bn_eval_affine_multi: 1 <= n <= %d layers", SPCL_BN_EVAL_MAX);
  BnEvalItems p;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    SPCL_CHECK_ARG(items[i].gamma && items[i].beta && items[i].running_mean && items[i].running_var && items[i].st,
                   "bn_eval_affine_multi: null pointer (layer %d)", i);
    SPCL_CHECK_ARG(items[i].C > 0 && items[i].CS >= items[i].C, "bn_eval_affine_multi: bad shape (layer %d)", i);
    p.it[i] = items[i];
    blocks += cdiv(items[i].CS, 256);
    p.blk_end[i] = blocks;
  }
  SPCL_LAUNCH(bn_eval_affine_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n);
  SPCL_LAUNCH_CHECK("bn_eval_affine_multi");
  return SPCL_OK;
}

extern "C" int spcl_bnrelu_pool_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                                        const float* shift, void* act_out, void* pool_out, void* stream) {
  SPCL_CHECK_ARG(y && scale && shift && (act_out || pool_out), "bnrelu_pool_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0 && CS <= 1024, "bnrelu_pool_forward: bad shape");
  SPCL_CHECK_ARG(!pool_out || (H >= 2 && W >= 2), "bnrelu_pool_forward: 2x2 pooling needs H,W >= 2");
  hipStream_t st = (hipStream_t)stream;
  if (dtype != SPCL_F32 && dtype != SPCL_BF16) {
    set_error("bnrelu_pool_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  if (dtype == SPCL_F32) bnrelu_fwd_launch<float>(y, N, H, W, CS, scale, shift, act_out, pool_out, st);
  else bnrelu_fwd_launch<bf16_t>(y, N, H, W, CS, scale, shift, act_out, pool_out, st);
  SPCL_LAUNCH_CHECK("bnrelu_pool_forward");
  return SPCL_OK;
}

// BN-apply + ReLU with the activation's global average per (image, channel) as a side output (bnrelu_fwd_gap_kernel): the
// coefficients from scale / shift, or -- bn != NULL -- derived from an accumulator block as in spcl_bnrelu_pool_forward_acc.
// gap_out [N][C] f32.  Small maps (H W <= 4096) of CS <= 256 channels (C <= 256 threads finish the means).
extern "C" int spcl_bnrelu_gap_supported(int dtype, int H, int W, int C, int CS) {
  return (dtype == SPCL_F32 || dtype == SPCL_BF16) && H > 0 && W > 0 && (long)H * W <= 4096 && C > 0 && C <= CS && CS <= 256 &&
         CS % 16 == 0;
}

extern "C" int spcl_bnrelu_gap_forward(const void* y, int dtype, int N, int H, int W, int C, int CS, const float* scale,
                                       const float* shift, const spcl_bn_acc* bn, void* act_out, float* gap_out,
                                       void* stream) {
  SPCL_CHECK_ARG(y && act_out && gap_out && (bn || (scale && shift)), "bnrelu_gap_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && spcl_bnrelu_gap_supported(dtype, H, W, C, CS), "bnrelu_gap_forward: unsupported shape / dtype");
  BnAccFwd f{};
  if (bn != nullptr) {
    SPCL_CHECK_ARG(bn->acc && bn->gamma && bn->beta && bn->st && bn->CS == CS && bn->C > 0 && bn->C <= CS && bn->count >= 1.f,
                   "bnrelu_gap_forward: bad accumulator description");
    f = BnAccFwd{bn->acc, bn->gamma, bn->beta, bn->running_mean, bn->running_var, bn->num_batches_tracked, bn->st,
                 bn->momentum, bn->eps, bn->count, bn->C, bn->CS, 1.0 / (double)bn->count};
  }
  hipStream_t st = (hipStream_t)stream;
  const int CW = CS % 64 == 0 ? 64 : CS;  // channels per workgroup: 64-channel windows (128 bytes of a bf16 pixel) where they tile
  if (dtype == SPCL_F32)
    SPCL_LAUNCH((bnrelu_fwd_gap_kernel<float>), dim3(N, CS / CW), dim3(256), 0, st, (const float*)y, H * W, C, CS, CW, scale,
                shift, (float*)act_out, gap_out, f);
  else
    SPCL_LAUNCH((bnrelu_fwd_gap_kernel<bf16_t>), dim3(N, CS / CW), dim3(256), 0, st, (const bf16_t*)y, H * W, C, CS, CW, scale,
                shift, (bf16_t*)act_out, gap_out, f);
  SPCL_LAUNCH_CHECK("bnrelu_gap_forward");
  return SPCL_OK;
}

// BN-apply + ReLU (+ 2x2 max-pool) with scale / shift DERIVED from a fixed-point accumulator block (bn_acc.hpp) in the kernel's
// prologue: what spcl_bn_finalize + spcl_bnrelu_pool_forward do in two launches, in one.  The launch's first workgroup writes
// bn->st (mean, invstd, scale, shift: what backward / a later eval needs) and updates the running statistics.  CS <= 256.
extern "C" int spcl_bnrelu_pool_forward_acc(const void* y, int dtype, int N, int H, int W, int CS, const spcl_bn_acc* bn,
                                            void* act_out, void* pool_out, void* stream) {
  SPCL_CHECK_ARG(y && bn && (act_out || pool_out), "bnrelu_pool_forward_acc: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0 && CS <= 256, "bnrelu_pool_forward_acc: bad shape (CS <= 256)");
  SPCL_CHECK_ARG(!pool_out || (H >= 2 && W >= 2), "bnrelu_pool_forward_acc: 2x2 pooling needs H,W >= 2");
  SPCL_CHECK_ARG(bn->acc && bn->gamma && bn->beta && bn->st && bn->CS == CS && bn->C > 0 && bn->C <= CS && bn->count >= 1.f,
                 "bnrelu_pool_forward_acc: bad accumulator description");
  hipStream_t st = (hipStream_t)stream;
  const BnAccFwd f{bn->acc, bn->gamma, bn->beta, bn->running_mean, bn->running_var, bn->num_batches_tracked, bn->st,
                   bn->momentum, bn->eps, bn->count, bn->C, bn->CS, 1.0 / (double)bn->count};
  if (dtype == SPCL_F32) bnrelu_fwd_launch<float>(y, N, H, W, CS, nullptr, nullptr, act_out, pool_out, st, &f);
  else if (dtype == SPCL_BF16) bnrelu_fwd_launch<bf16_t>(y, N, H, W, CS, nullptr, nullptr, act_out, pool_out, st, &f);
  else {
    set_error("bnrelu_pool_forward_acc: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_pool_forward_acc");
  return SPCL_OK;
}

// BN + ReLU (+ max-pool) BACKWARD apply pass with its coefficients derived from a fixed-point accumulator block: no reduction
// pass over the tensors where the dgrad that produced the incoming gradient filled the block (spcl_conv3x3_dgrad_bnstats_acc /
// _poolstats_acc), and no finalize launch in any case.  Exactly one of
//   dact    [N][H][W][CS]      gradient w.r.t. the activation; the block is already filled
//   dpool   [N][H/2][W/2][CS]  gradient w.r.t. the pooled activation (H, W even); the block is already filled
//   dact_nc [N][CS]            one value per (image, channel) (global average pool); THIS call's reduction pass fills the block
// st = the forward's [4][CS] (mean, invstd, scale, shift).  The first workgroup of the apply launch writes dgamma / dbeta.
extern "C" int spcl_bnrelu_backward_acc(const void* y, const void* dact, const void* dpool, const void* dact_nc, int dtype,
                                        int N, int H, int W, int C, int CS, const float* st4, int training, long long* acc,
                                        float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && st4 && acc && dgamma && dbeta && dy, "bnrelu_backward_acc: null pointer");
  SPCL_CHECK_ARG((dact != nullptr) + (dpool != nullptr) + (dact_nc != nullptr) == 1,
                 "bnrelu_backward_acc: exactly one of dact, dpool, dact_nc");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256, "bnrelu_backward_acc: bad shape (CS <= 256)");
  SPCL_CHECK_ARG(!dpool || (H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0), "bnrelu_backward_acc: pooled form needs even H, W");
  SPCL_CHECK_ARG(dtype == SPCL_BF16 || dtype == SPCL_F32, "bnrelu_backward_acc: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const float *mean = st4, *invstd = st4 + CS, *scale = st4 + 2 * CS, *shift = st4 + 3 * CS;
  const void* g = dact_nc != nullptr ? dact_nc : dact;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, g, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                             nullptr, nullptr, nullptr, 0, dact_nc != nullptr, acc);
  else
    bnrelu_bwd_launch<bf16_t>(y, g, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                              nullptr, nullptr, nullptr, 0, dact_nc != nullptr, acc);
  SPCL_LAUNCH_CHECK("bnrelu_backward_acc");
  return SPCL_OK;
}

extern "C" int spcl_bnrelu_up2_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                                       const float* shift, void* up_out, void* stream) {
  SPCL_CHECK_ARG(y && scale && shift && up_out, "bnrelu_up2_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0 && CS <= 1024, "bnrelu_up2_forward: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nrows = (unsigned)N * (unsigned)H;
  const unsigned grid = nrows < (unsigned)STREAM_MAX_WG ? nrows : (unsigned)STREAM_MAX_WG;
  const double tb = (double)N * H * W * CS * (dtype == SPCL_F32 ? 4.0 : 2.0);
  prof_cost(5.0 * tb, 0.0);
  if (dtype == SPCL_F32)
    SPCL_LAUNCH((bnrelu_fwd_up2_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)y, nrows, W, CS, scale, shift,
                (float*)up_out);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH((bnrelu_fwd_up2_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)y, nrows, W, CS, scale, shift,
                (bf16_t*)up_out);
  else {
    set_error("bnrelu_up2_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_up2_forward");
  return SPCL_OK;
}

extern "C" int spcl_bnrelu_pool_forward_strided(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                                                const float* shift, void* act_out, int act_stride, void* pool_out,
                                                void* stream) {
  SPCL_CHECK_ARG(act_out && act_stride >= CS && act_stride % 8 == 0,
                 "bnrelu_pool_forward_strided: act_out with a pixel stride >= CS, a multiple of 8 elements");
  tl_act_stride = act_stride;
  const int rc = spcl_bnrelu_pool_forward(y, dtype, N, H, W, CS, scale, shift, act_out, pool_out, stream);
  tl_act_stride = 0;
  return rc;
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

extern "C" int spcl_bnrelu_pool_backward_strided(const void* y, const void* dact, int dact_stride, const void* dpool,
                                                 int dtype, int N, int H, int W, int C, int CS, const float* mean,
                                                 const float* invstd, const float* scale, const float* shift,
                                                 int training, float* ws, float* dgamma, float* dbeta, void* dy,
                                                 void* stream) {
  SPCL_CHECK_ARG(dact && dact_stride >= CS && dact_stride % 8 == 0,
                 "bnrelu_pool_backward_strided: dact with a pixel stride >= CS, a multiple of 8 elements");
  tl_dact_stride = dact_stride;
  const int rc = spcl_bnrelu_pool_backward(y, dact, dpool, dtype, N, H, W, C, CS, mean, invstd, scale, shift, training, ws,
                                           dgamma, dbeta, dy, stream);
  tl_dact_stride = 0;
  return rc;
}

// BN + ReLU backward of a block whose activation went through nn.Upsample(scale_factor=2) (unet.py:89): d_up is the gradient
// w.r.t. the UPSAMPLED activation, [N][2H][2W][CS]; dact [N][H][W][CS] is scratch the call fills with the 2 x 2 sums
// (= spcl_upsample2x_backward's output) on its way to the BatchNorm sums.  Results as spcl_bnrelu_pool_backward(y, dact, NULL).
extern "C" int spcl_bnrelu_backward_up2(const void* y, const void* d_up, void* dact, int dtype, int N, int H, int W, int C,
                                        int CS, const float* mean, const float* invstd, const float* scale,
                                        const float* shift, int training, float* ws, float* dgamma, float* dbeta, void* dy,
                                        void* stream) {
  SPCL_CHECK_ARG(d_up && dact, "bnrelu_backward_up2: null pointer");
  SPCL_CHECK_ARG((size_t)N * H * W * 4 < 0xffffffffull, "bnrelu_backward_up2: too many pixels");
  tl_up2_src = d_up;
  const int rc = spcl_bnrelu_pool_backward(y, dact, nullptr, dtype, N, H, W, C, CS, mean, invstd, scale, shift, training, ws,
                                           dgamma, dbeta, dy, stream);
  tl_up2_src = nullptr;
  return rc;
}

// BN + ReLU (+ max-pool) backward whose reduction pass ADDS its sums to a (zeroed) fixed-point accumulator block and whose
// apply pass derives the coefficients from it: two launches, no finalize launch in between -- for gradients that did not
// come with their sums (spcl_bnrelu_backward_acc covers those that did): an activation with several consumers (the decoder's
// skip connections), the up-convolutions, a gradient that arrives as a channel slice (dact_stride > 0) or at twice the
// resolution (d_up non-NULL: dact is then scratch the call fills with the 2 x 2 sums).  st4 = the forward's [4][CS].
extern "C" int spcl_bnrelu_backward_fill_acc(const void* y, void* dact, int dact_stride, const void* dpool, const void* d_up,
                                             int dtype, int N, int H, int W, int C, int CS, const float* st4, int training,
                                             long long* acc, float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && st4 && acc && dgamma && dbeta && dy && (dact || dpool), "bnrelu_backward_fill_acc: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256, "bnrelu_backward_fill_acc: bad shape (CS <= 256)");
  SPCL_CHECK_ARG(dtype == SPCL_BF16 || dtype == SPCL_F32, "bnrelu_backward_fill_acc: dtype %d", dtype);
  SPCL_CHECK_ARG(!dpool || (H >= 2 && W >= 2), "bnrelu_backward_fill_acc: 2x2 pooling needs H,W >= 2");
  SPCL_CHECK_ARG(dact_stride == 0 || (dact && dact_stride >= CS && dact_stride % 8 == 0 && !d_up),
                 "bnrelu_backward_fill_acc: dact with a pixel stride >= CS, a multiple of 8 elements");
  SPCL_CHECK_ARG(!d_up || (dact && !dpool && (size_t)N * H * W * 4 < 0xffffffffull), "bnrelu_backward_fill_acc: the x2 form takes d_up + scratch dact");
  hipStream_t st = (hipStream_t)stream;
  const float *mean = st4, *invstd = st4 + CS, *scale = st4 + 2 * CS, *shift = st4 + 3 * CS;
  tl_acc_fill = true;
  tl_dact_stride = dact_stride;
  tl_up2_src = d_up;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                             nullptr, nullptr, nullptr, 0, false, acc);
  else
    bnrelu_bwd_launch<bf16_t>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                              nullptr, nullptr, nullptr, 0, false, acc);
  tl_acc_fill = false;
  tl_dact_stride = 0;
  tl_up2_src = nullptr;
  SPCL_LAUNCH_CHECK("bnrelu_backward_fill_acc");
  return SPCL_OK;
}

// ... and for a gradient whose sums came as per-tile ROWS [nrows][2][CS] (sum dz, sum dz (y - mean): the dgrad epilogues of
// conv_fast.hip at tile counts beyond the blocks' own limit): one small launch adds the rows to the zeroed block, the apply
// pass (dact: linear; dpool: 2 x 2 max-pool, even H, W) derives -- spcl_bnrelu_backward_rows / spcl_bnrelu_pool_backward_rows
// without their finalize launch.  bf16 / f32, CS <= 256.
extern "C" int spcl_bnrelu_backward_rows_acc(const void* y, const void* dact, const void* dpool, const float* rows, int nrows,
                                             int dtype, int N, int H, int W, int C, int CS, const float* st4, int training,
                                             long long* acc, float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && rows && st4 && acc && dgamma && dbeta && dy && ((dact != nullptr) != (dpool != nullptr)),
                 "bnrelu_backward_rows_acc: null pointer / exactly one of dact, dpool");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256 && nrows > 0,
                 "bnrelu_backward_rows_acc: bad shape (CS <= 256)");
  SPCL_CHECK_ARG(!dpool || (H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0), "bnrelu_backward_rows_acc: pooled form needs even H, W");
  SPCL_CHECK_ARG(dtype == SPCL_BF16 || dtype == SPCL_F32, "bnrelu_backward_rows_acc: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const float *mean = st4, *invstd = st4 + CS, *scale = st4 + 2 * CS, *shift = st4 + 3 * CS;
  tl_acc_fill = true;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                             nullptr, nullptr, rows, nrows, false, acc);
  else
    bnrelu_bwd_launch<bf16_t>(y, dact, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, nullptr, dgamma, dbeta, dy, st,
                              nullptr, nullptr, rows, nrows, false, acc);
  tl_acc_fill = false;
  SPCL_LAUNCH_CHECK("bnrelu_backward_rows_acc");
  return SPCL_OK;
}

// BN + ReLU backward for a gradient that is the same for every pixel of an image: dact_nc [N][CS] of dtype (what
// spcl_proj_heads_backward_pooled leaves: the gradient of the global average pool in front of the projector,
// contrastyou/projectors/heads.py:9-18).  Same arithmetic as spcl_bnrelu_pool_backward on the expanded tensor.
extern "C" int spcl_bnrelu_backward_bcast(const void* y, const void* dact_nc, int dtype, int N, int H, int W, int C, int CS,
                                          const float* mean, const float* invstd, const float* scale, const float* shift,
                                          int training, float* ws, float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && dact_nc && mean && invstd && scale && shift && ws && dgamma && dbeta && dy,
                 "bnrelu_backward_bcast: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 1024, "bnrelu_backward_bcast: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact_nc, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                             dy, st, nullptr, nullptr, nullptr, 0, true);
  else if (dtype == SPCL_BF16)
    bnrelu_bwd_launch<bf16_t>(y, dact_nc, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                              dy, st, nullptr, nullptr, nullptr, 0, true);
  else {
    set_error("bnrelu_backward_bcast: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_backward_bcast");
  return SPCL_OK;
}

extern "C" size_t spcl_bnrelu_image_wgrad_workspace_bytes(int N, int H, int W, int CS) {
  return spcl_bnrelu_bwd_workspace_bytes(N, H, W, CS) + ((size_t)IMG_WGRAD_WG * 9 * CS + (size_t)W) * sizeof(float);
}

extern "C" int spcl_bnrelu_backward_image_wgrad(const void* y, const void* dact, const float* image, int dtype, int N,
                                                int H, int W, int C, int CS, const float* mean, const float* invstd,
                                                const float* scale, const float* shift, int training, float* ws,
                                                float* dgamma, float* dbeta, float* dw, void* stream) {
  SPCL_CHECK_ARG(y && dact && image && mean && invstd && scale && shift && ws && dgamma && dbeta && dw,
                 "bnrelu_backward_image_wgrad: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256 && (CS & (CS - 1)) == 0,
                 "bnrelu_backward_image_wgrad: bad shape (CS must be a power of two in 16..256)");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                             nullptr, st, image, dw);
  else if (dtype == SPCL_BF16)
    bnrelu_bwd_launch<bf16_t>(y, dact, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma,
                              dbeta, nullptr, st, image, dw);
  else {
    set_error("bnrelu_backward_image_wgrad: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_backward_image_wgrad");
  return SPCL_OK;
}


extern "C" int spcl_bnrelu_backward_rows(const void* y, const void* dact, const float* image, const float* rows, int nrows,
                                         int dtype, int N, int H, int W, int C, int CS, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, int training,
                                         float* ws, float* dgamma, float* dbeta, void* dy, float* dw, void* stream) {
  SPCL_CHECK_ARG(y && dact && rows && mean && invstd && scale && shift && ws && dgamma && dbeta,
                 "bnrelu_backward_rows: null pointer");
  SPCL_CHECK_ARG((image != nullptr) == (dw != nullptr) && (image != nullptr) != (dy != nullptr),
                 "bnrelu_backward_rows: either (image, dw) for the fused first-layer weight gradient or dy");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 1024 && nrows > 0,
                 "bnrelu_backward_rows: bad shape");
  SPCL_CHECK_ARG(image == nullptr || (CS <= 256 && (CS & (CS - 1)) == 0), "bnrelu_backward_rows: image path needs CS "
                 "a power of two in 16..256");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SPCL_F32)
    bnrelu_bwd_launch<float>(y, dact, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta, dy,
                             st, image, dw, rows, nrows);
  else if (dtype == SPCL_BF16)
    bnrelu_bwd_launch<bf16_t>(y, dact, nullptr, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                              dy, st, image, dw, rows, nrows);
  else {
    set_error("bnrelu_backward_rows: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_backward_rows");
  return SPCL_OK;
}

// ---- image3: BN + ReLU backward of the FIRST conv of a one-channel-image block and that conv's weight gradient, finished
// from the eleven-row tiles of spcl_conv3x3_dgrad_bnstats_image and the image autocorrelation: no pass over y / g at all
// (see bnrelu_bwd_fin_kernel).  Replaces spcl_bnrelu_backward_rows(image != NULL) where that dgrad kernel exists.
extern "C" int spcl_image_autocorr_rows(int N, int H, int W) {  // per-band partial rows (the buffer holds 16 more in front)
  (void)W;
  return N * image_autocorr_bands(H);
}

extern "C" int spcl_image_autocorr(const float* image, int N, int H, int W, float* out, void* stream) {
  SPCL_CHECK_ARG(image && out && N > 0 && H > 0 && W > 0 && W <= ACORR_MAXW, "image_autocorr: bad arguments (W <= %d)",
                 ACORR_MAXW);
  const int rows = spcl_image_autocorr_rows(N, H, W);
  prof_cost((double)N * H * W * 4.0, 2.0 * 54.0 * N * H * W);
  SPCL_LAUNCH(image_autocorr_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, image, H, W, out);
  SPCL_LAUNCH_CHECK("image_autocorr");
  return SPCL_OK;
}

extern "C" size_t spcl_bnrelu_image3_workspace_bytes(int CS) {
  return ((size_t)BWD_MAX_WG * 11 * CS + 2 * (size_t)CS + ACORR_FOLD * 64) * sizeof(float);
}

extern "C" int spcl_bnrelu_backward_rows_image3(const float* rows11, int nrows, const float* acorr, int nacorr,
                                                const float* w_oihw, int N, int H, int W, int C, int CS,
                                                const float* mean, const float* invstd, const float* scale,
                                                int training, float* ws, float* dgamma, float* dbeta, float* dw,
                                                void* stream) {
  SPCL_CHECK_ARG(rows11 && acorr && w_oihw && mean && invstd && scale && ws && dgamma && dbeta && dw,
                 "bnrelu_backward_rows_image3: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256 && nrows > 0 && nacorr > 0,
                 "bnrelu_backward_rows_image3: bad shape");
  hipStream_t st = (hipStream_t)stream;
  float* partial = ws;                             // [nwg][11][CS]
  float* ab = ws + (size_t)BWD_MAX_WG * 11 * CS;   // [2][CS]: the folded coefficients (kept for symmetry with the other paths)
  float* afold = ab + 2 * (size_t)CS;              // [ACORR_FOLD][64]
  const float* fin_src = rows11;
  int nwg = nrows, transposed = 0;
  constexpr int IMG3_ROWS = 1024;  // the final kernel walks eleven strided sums per channel: four rows per thread
  if (nrows > IMG3_ROWS) {
    const int G = (nrows + IMG3_ROWS - 1) / IMG3_ROWS;
    nwg = (nrows + G - 1) / G;
    prof_cost((double)nrows * 11 * CS * 4.0, 0.0);
    // ... and, in the same launch, the autocorrelation's per-band rows folded to ACORR_FOLD
    const bool fold = nacorr > ACORR_FOLD;
    SPCL_LAUNCH(bwd_rows_group_kernel, dim3(nwg + (fold ? ACORR_FOLD : 0)), dim3(256), 0, st, rows11, nrows, G, CS, partial,
                11, nwg, acorr, nacorr, afold, 1);
    fin_src = partial;
    transposed = 1;
    if (fold) {
      acorr = afold;
      nacorr = ACORR_FOLD;
    }
  }
  const Image3Args im{acorr, nacorr, w_oihw, dw};
  SPCL_LAUNCH(bnrelu_bwd_fin_kernel, dim3(3 * CS), dim3(256), 0, st, fin_src, nwg, C, CS, (float)((size_t)N * H * W), training,
              mean, invstd, scale, dgamma, dbeta, ab, (float*)nullptr, 0, 1, 11, im, transposed);
  SPCL_LAUNCH_CHECK("bnrelu_backward_rows_image3");
  return SPCL_OK;
}

// ... from ONE row set per workgroup of spcl_conv16_bwd_fused, already in the final kernel's layout
// ([sub-row 11][channel CS][workgroup]) and with the autocorrelation rows already folded: the final kernel alone
extern "C" int spcl_bnrelu_backward_wgrows_image3(const float* wg_rows, int nwg, const float* acorr, int nacorr,
                                                  const float* w_oihw, int N, int H, int W, int C, int CS,
                                                  const float* mean, const float* invstd, const float* scale,
                                                  int training, float* ws, float* dgamma, float* dbeta, float* dw,
                                                  void* stream) {
  SPCL_CHECK_ARG(wg_rows && acorr && w_oihw && mean && invstd && scale && ws && dgamma && dbeta && dw,
                 "bnrelu_backward_wgrows_image3: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 256 && nwg > 0 && nacorr > 0 &&
                     nacorr <= 64,
                 "bnrelu_backward_wgrows_image3: bad shape");
  hipStream_t st = (hipStream_t)stream;
  float* ab = ws + (size_t)BWD_MAX_WG * 11 * CS;
  const Image3Args im{acorr, nacorr, w_oihw, dw};
  prof_cost((double)nwg * 11 * CS * 4.0, 0.0);
  SPCL_LAUNCH(bnrelu_bwd_fin_kernel, dim3(3 * CS), dim3(256), 0, st, wg_rows, nwg, C, CS, (float)((size_t)N * H * W), training,
              mean, invstd, scale, dgamma, dbeta, ab, (float*)nullptr, 0, 1, 11, im, 1);
  SPCL_LAUNCH_CHECK("bnrelu_backward_wgrows_image3");
  return SPCL_OK;
}

// BN + ReLU + 2x2 max-pool backward finished from per-tile rows that came with the dgrad producing dpool
// (spcl_conv3x3_dgrad_poolstats): final reduction of the rows -> coefficients -> the apply pass.  No `dact` here: a block
// whose activation is also consumed directly (skip connection, hook tap) keeps the two-pass path.
extern "C" int spcl_bnrelu_pool_backward_rows(const void* y, const void* dpool, const float* rows, int nrows, int dtype,
                                              int N, int H, int W, int C, int CS, const float* mean, const float* invstd,
                                              const float* scale, const float* shift, int training, float* ws,
                                              float* dgamma, float* dbeta, void* dy, void* stream) {
  SPCL_CHECK_ARG(y && dpool && rows && mean && invstd && scale && shift && ws && dgamma && dbeta && dy,
                 "bnrelu_pool_backward_rows: null pointer");
  SPCL_CHECK_ARG(N > 0 && H >= 2 && W >= 2 && C > 0 && CS >= C && CS % 16 == 0 && CS <= 1024 && nrows > 0,
                 "bnrelu_pool_backward_rows: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SPCL_BF16)
    bnrelu_bwd_launch<bf16_t>(y, nullptr, dpool, N, H, W, C, CS, mean, invstd, scale, shift, training, ws, dgamma, dbeta,
                              dy, st, nullptr, nullptr, rows, nrows);
  else {
    set_error("bnrelu_pool_backward_rows: dtype %d (bf16 only: the rows come from the specialised dgrad kernels)", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("bnrelu_pool_backward_rows");
  return SPCL_OK;
}

