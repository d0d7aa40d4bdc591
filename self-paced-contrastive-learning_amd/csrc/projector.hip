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
// Up to PROJ_MAX_HEADS projector heads of identical shape run as ONE launch per layer: blockIdx.z picks the head, the
// pointer tables (passed by value) its tensors.  A single head is the table of length one.
constexpr int PROJ_MAX_HEADS = 4;
struct CPtrs { const float* p[PROJ_MAX_HEADS]; };
struct MPtrs { float* p[PROJ_MAX_HEADS]; };

template <bool LEAKY_IN>
__global__ __launch_bounds__(256) void linear_fwd_kernel(CPtrs xs, CPtrs Ws, CPtrs bs, int N, int K, int O, MPtrs ys) {
  constexpr int RB = 8, KMAX = 8;  // K <= 512
  const float* __restrict__ x = xs.p[blockIdx.z];
  const float* __restrict__ W = Ws.p[blockIdx.z];
  const float* __restrict__ b = bs.p[blockIdx.z];
  float* __restrict__ y = ys.p[blockIdx.z];
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
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(CPtrs os, int N, int O, MPtrs zs) {
  const float* __restrict__ o = os.p[blockIdx.z];
  float* __restrict__ z = zs.p[blockIdx.z];
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
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(CPtrs os, CPtrs dzs, int N, int O, MPtrs d_os) {
  const float* __restrict__ o = os.p[blockIdx.z];
  const float* __restrict__ dz = dzs.p[blockIdx.z];
  float* __restrict__ d_o = d_os.p[blockIdx.z];
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
__device__ __forceinline__ void linear_wgrad_body(const CPtrs& gs, const CPtrs& xs, int N, int K, int O, const MPtrs& dWs,
                                                  const MPtrs& dbs, int bx, int o, int z) {
  const float* __restrict__ g = gs.p[z];
  const float* __restrict__ x = xs.p[z];
  float* __restrict__ dW = dWs.p[z];
  float* __restrict__ db = dbs.p[z];
  const int k = bx * 256 + threadIdx.x;
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
template <bool LEAKY_IN>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(CPtrs gs, CPtrs xs, int N, int K, int O, MPtrs dWs, MPtrs dbs) {
  linear_wgrad_body<LEAKY_IN>(gs, xs, N, K, O, dWs, dbs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dx[n][k] = (sum_o g[n][o] W[o][k]) * (LEAKY_OUT ? leaky'(pre[n][k]) : 1).  Workgroup = (row n, 64 columns k): the
// 4 waves take interleaved o and are combined through LDS in fixed order.
// NSUM > 0: the heads share the input (the pooled feature): ONE output, the sum over the NSUM heads in index order.
// `nc` != null: ALSO the pooled feature's gradient divided by HW in the feature's storage type, [N][K] -- the gradient of
// the global average pool as the ONE value per (image, channel) it is; the consumer (bn.hip, spcl_bnrelu_backward_bcast)
// reads it as such and the N x HW x K broadcast tensor is never written.
struct PooledOut {
  void* nc;
  int dtype, HW;
};
template <bool LEAKY_OUT>
__device__ __forceinline__ void linear_dgrad_body(const CPtrs& gs, const CPtrs& Ws, const CPtrs& pres, int N, int K, int O,
                                                  const MPtrs& dxs, int nsum, int bx, int n, int z, float (*red)[64],
                                                  PooledOut po = PooledOut{nullptr, 0, 1}) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = bx * 64 + lane;
  const int h0 = nsum > 0 ? 0 : z, h1 = nsum > 0 ? nsum : z + 1;
  float s = 0.f;
  if (k < K) {
    for (int h = h0; h < h1; ++h) {
      const float* __restrict__ g = gs.p[h];
      const float* __restrict__ W = Ws.p[h];
#pragma unroll 8
      for (int o = wave; o < O; o += 4) s = fmaf(g[(size_t)n * O + o], W[(size_t)o * K + k], s);
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && k < K) {
    float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (LEAKY_OUT) v *= pres.p[h0][(size_t)n * K + k] > 0.f ? 1.f : kLeaky;
    dxs.p[h0][(size_t)n * K + k] = v;
    if (po.nc != nullptr) {  // == avgpool_bwd_kernel's value, once per (image, channel)
      const float pv = v / (float)po.HW;
      if (po.dtype == SPCL_BF16) ((bf16_t*)po.nc)[(size_t)n * K + k] = f32_to_bf16(pv);
      else ((float*)po.nc)[(size_t)n * K + k] = pv;
    }
  }
}
// One layer's weight gradient AND input gradient in one launch (they read the same g and are independent; each is a
// few-microsecond launch of its own otherwise): blocks [0, nwb) are the weight-gradient grid (bx fastest, then o), the rest
// the input-gradient grid (bx fastest, then n); blockIdx.z = head.  With nsum > 0 the input-gradient part runs for z = 0 only.
template <bool LEAKY>
__global__ __launch_bounds__(256) void linear_bwd_pair_kernel(CPtrs gs, CPtrs xs, CPtrs Ws, CPtrs pres, int N, int K, int O,
                                                              MPtrs dWs, MPtrs dbs, MPtrs dxs, int nsum, int gxw, int nwb,
                                                              int gxd, PooledOut po = PooledOut{nullptr, 0, 1}) {
  __shared__ float red[4][64];
  const int b = blockIdx.x;
  if (b < nwb) {
    const int o = b / gxw;
    linear_wgrad_body<LEAKY>(gs, xs, N, K, O, dWs, dbs, b - o * gxw, o, blockIdx.z);
  } else {
    if (nsum > 0 && blockIdx.z > 0) return;
    const int r = b - nwb, n = r / gxd;
    linear_dgrad_body<LEAKY>(gs, Ws, pres, N, K, O, dxs, nsum, r - n * gxd, n, blockIdx.z, red, po);
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

// ---- nn.AdaptiveAvgPool2d / nn.AdaptiveMaxPool2d((OH, OW)) on an NHWC tensor (projectors/nn.py:56-64; the pooling of
// ProjectionHead(pool_name="adaptive_max") and of DenseProjectionHead, projectors/heads.py:96-120).  Window of output
// (oy, ox) = rows floor(oy H / OH) .. ceil((oy + 1) H / OH) - 1, same for columns (torch's rule).  One thread per output
// element; max keeps the FIRST maximum's flat input index for the backward (torch's tie rule in scan order).
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const T* __restrict__ x, int H, int W, int C, int Cs,
                                                                int OH, int OW, float* __restrict__ out,
                                                                int* __restrict__ arg, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  size_t r = idx / C;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const size_t n = r / OH;
  const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
  const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
  float acc = MAX ? -INFINITY : 0.f;
  int best = 0;
  for (int y = y0; y < y1; ++y)
    for (int xx = x0; xx < x1; ++xx) {
      const float v = Elem<T>::load(x + ((n * H + y) * W + xx) * (size_t)Cs + c);
      if (MAX) {
        if (v > acc || (y == y0 && xx == x0)) {
          acc = v;
          best = y * W + xx;
        }
      } else {
        acc += v;
      }
    }
  out[idx] = MAX ? acc : acc / (float)((y1 - y0) * (x1 - x0));
  if (MAX) arg[idx] = best;
}

// dx[n][y][x][c] = sum over the windows (oy, ox) that contain (y, x) of dout / window size (avg) or of dout where the
// window's arg-max is (y, x) (max).  One thread per INPUT element walks the (at most 2 x 2 ... few) covering windows:
// no atomics, deterministic.
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg,
                                                                int H, int W, int C, int Cs, int OH, int OW,
                                                                T* __restrict__ dx, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % Cs);
  size_t r = idx / Cs;
  const int xx = (int)(r % W);
  r /= W;
  const int y = (int)(r % H);
  const size_t n = r / H;
  float g = 0.f;
  if (c < C) {
    // windows containing row y: oy with floor(oy H / OH) <= y < ceil((oy + 1) H / OH)
    int oy_lo = (int)(((long)y * OH) / H), ox_lo = (int)(((long)xx * OW) / W);
    while (oy_lo > 0 && ((oy_lo) * H + OH - 1) / OH > y) --oy_lo;   // previous window still reaches y
    while (ox_lo > 0 && ((ox_lo) * W + OW - 1) / OW > xx) --ox_lo;
    for (int oy = oy_lo; oy < OH && (oy * H) / OH <= y; ++oy) {
      const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
      if (y < y0 || y >= y1) continue;
      for (int ox = ox_lo; ox < OW && (ox * W) / OW <= xx; ++ox) {
        const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
        if (xx < x0 || xx >= x1) continue;
        const size_t o = ((n * OH + oy) * OW + ox) * (size_t)C + c;
        if (MAX) g += arg[o] == y * W + xx ? dout[o] : 0.f;
        else g += dout[o] / (float)((y1 - y0) * (x1 - x0));
      }
    }
  }
  Elem<T>::store(dx + idx, g);
}

// Adaptive AVERAGE pooling of an f32 map, one wave per output window, a lane = four consecutive channels (round 6: the dense
// projector pools its 256-channel hidden activation -- 3.1 GB at Up_conv2 (60 maps of 224^2) -- and the one-thread-per-output
// kernel above kept one 4-byte load in flight per thread: a fifth of the rate of this one).  Pixels in the same order, one add per pixel: the old kernel's sums bit for
// bit; four pixels' loads are issued before the first add.  C % 4 == 0.
template <typename T>
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float ap_load4(const T* p) {
  typedef __attribute__((ext_vector_type(4))) float v4f;
  if (sizeof(T) == 4) return *(const v4f*)p;
  const uint2 raw = *(const uint2*)p;
  v4f v;
  v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
  v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void adaptive_avgpool_fwd_win_kernel(const T* __restrict__ x, int H, int W, int C, int OH,
                                                                       int OW, float* __restrict__ out, size_t nwin) {
  typedef __attribute__((ext_vector_type(4))) float v4f;
  const int lane = threadIdx.x & 63;
  const size_t win = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (win >= nwin) return;
  const int ox = (int)(win % OW);
  const size_t r = win / OW;
  const int oy = (int)(r % OH);
  const size_t n = r / OH;
  const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
  const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
  const float size = (float)((y1 - y0) * (x1 - x0));
  for (int c0 = 4 * lane; c0 < C; c0 += 256) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int y = y0; y < y1; ++y) {
      const T* row = x + ((n * H + y) * W) * (size_t)C + c0;
      int xx = x0;
      for (; xx + 4 <= x1; xx += 4) {
        const v4f a = ap_load4<T>(row + (size_t)xx * C), b = ap_load4<T>(row + (size_t)(xx + 1) * C);
        const v4f c = ap_load4<T>(row + (size_t)(xx + 2) * C), d = ap_load4<T>(row + (size_t)(xx + 3) * C);
        acc += a; acc += b; acc += c; acc += d;
      }
      for (; xx < x1; ++xx) acc += ap_load4<T>(row + (size_t)xx * C);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] / size;
    *(v4f*)(out + win * (size_t)C + c0) = acc;
  }
}

// The same gradient, one WAVE per input pixel (round 6): which windows cover (y, x) does not depend on the channel -- with
// one thread per element every element paid the integer divisions of the window search (1.6 ms for the 753 000 x 256
// gradient of a dense projection: the longest launch of a decoder pre-training step).  Here the search is wave-uniform and
// a lane owns four consecutive channels (16-byte loads of dout, one 16- or 8-byte store).  Same windows in the same order,
// same divisions: the old kernel's values bit for bit.  C and Cs multiples of 4.
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_pix_kernel(const float* __restrict__ dout, const int* __restrict__ arg,
                                                                    int H, int W, int C, int Cs, int OH, int OW,
                                                                    T* __restrict__ dx, size_t nseg, int segs_per_row,
                                                                    const T* __restrict__ act = nullptr) {
  // act (optional, [npix][Cs] f32): the pooled tensor was LeakyReLU(pre) and `act` holds it -- the gradient leaves multiplied
  // by LeakyReLU'(pre), whose sign `act` carries
  // A wave owns a SEGMENT of APB_SEG consecutive pixels of one image row: the row's windows (oy range) are found once per
  // segment, the column windows per pixel (the search is a few integer divisions: per pixel and wave they were what a
  // 3-million-pixel launch spent most of its time on)
  typedef __attribute__((ext_vector_type(4))) float v4f;
  typedef __attribute__((ext_vector_type(4))) int v4i;
  constexpr int APB_SEG = 8;
  const int lane = threadIdx.x & 63;
  const size_t seg = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int sx = (int)(seg % segs_per_row);
  const size_t r = seg / segs_per_row;
  const int y = (int)(r % H);
  const size_t n = r / H;
  int oy_lo = (int)(((long)y * OH) / H);
  while (oy_lo > 0 && ((oy_lo) * H + OH - 1) / OH > y) --oy_lo;   // previous window still reaches y
  const int xe = min(W, (sx + 1) * APB_SEG);
  for (int xx = sx * APB_SEG; xx < xe; ++xx) {
    const size_t pix = (n * H + y) * (size_t)W + xx;
    int ox_lo = (int)(((long)xx * OW) / W);
    while (ox_lo > 0 && ((ox_lo) * W + OW - 1) / OW > xx) --ox_lo;
    for (int c0 = 4 * lane; c0 < Cs; c0 += 256) {
      v4f g = {0.f, 0.f, 0.f, 0.f};
      if (c0 < C) {
        for (int oy = oy_lo; oy < OH && (oy * H) / OH <= y; ++oy) {
          const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
          if (y < y0 || y >= y1) continue;
          for (int ox = ox_lo; ox < OW && (ox * W) / OW <= xx; ++ox) {
            const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
            if (xx < x0 || xx >= x1) continue;
            const size_t o = ((n * OH + oy) * OW + ox) * (size_t)C + c0;
            const v4f d = *(const v4f*)(dout + o);
            if (MAX) {
              const v4i a = *(const v4i*)(arg + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] += a[e] == y * W + xx ? d[e] : 0.f;
            } else {
              const float size = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] += d[e] / size;
            }
          }
        }
      }
      if (act != nullptr) {
        const v4f a = ap_load4<T>(act + pix * (size_t)Cs + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= a[e] > 0.f ? 1.f : kLeaky;
      }
      T* o = dx + pix * (size_t)Cs + c0;
      if (sizeof(T) == 4) {
        *(v4f*)o = g;
      } else {
        uint2 w;
        w.x = (unsigned)f32_to_bf16(g[0]) | ((unsigned)f32_to_bf16(g[1]) << 16);
        w.y = (unsigned)f32_to_bf16(g[2]) | ((unsigned)f32_to_bf16(g[3]) << 16);
        *(uint2*)o = w;
      }
    }
  }
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_adaptive_pool2d_forward(const void* x, int dtype, int N, int H, int W, int C, int Cs, int OH, int OW,
                                            int mode, float* out, int* argmax, void* stream) {
  SPCL_CHECK_ARG(x && out, "adaptive_pool2d_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Cs >= C && OH > 0 && OW > 0, "adaptive_pool2d_forward: bad shape");
  SPCL_CHECK_ARG(mode == 0 || (mode == 1 && argmax), "adaptive_pool2d_forward: mode %d (0 avg, 1 max + argmax)", mode);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * OH * OW * C;
  dim3 g((unsigned)((total + 255) / 256));
  if (mode == 0 && OH == 1 && OW == 1 && (dtype == SPCL_F32 || dtype == SPCL_BF16)) {
    // global average (every ProjectionHead, and the pooling several hooks share): the coalesced kernel of the projector
    // path (4 waves x 64 channels per image: 5 us at 128 x 16 x 16 x 256; the one-thread-per-output kernel below walks
    // H x W strided 2-byte loads per thread: 72 us)
    dim3 pg(cdiv(C, 64), N);
    if (dtype == SPCL_F32) SPCL_LAUNCH(avgpool_kernel<float>, pg, dim3(256), 0, st, (const float*)x, H * W, C, Cs, out);
    else SPCL_LAUNCH(avgpool_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t*)x, H * W, C, Cs, out);
    SPCL_LAUNCH_CHECK("adaptive_pool2d_forward");
    return SPCL_OK;
  }
  if ((dtype == SPCL_F32 || dtype == SPCL_BF16) && mode == 0 && C % 4 == 0 && Cs == C && (uintptr_t)x % 16 == 0 &&
      (uintptr_t)out % 16 == 0) {
    const size_t nwin = (size_t)N * OH * OW;
    if (dtype == SPCL_F32)
      SPCL_LAUNCH(adaptive_avgpool_fwd_win_kernel<float>, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, st, (const float*)x, H, W, C,
                  OH, OW, out, nwin);
    else
      SPCL_LAUNCH(adaptive_avgpool_fwd_win_kernel<bf16_t>, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, H, W,
                  C, OH, OW, out, nwin);
    SPCL_LAUNCH_CHECK("adaptive_pool2d_forward");
    return SPCL_OK;
  }
  if (dtype == SPCL_F32 && mode == 0)
    SPCL_LAUNCH((adaptive_pool_fwd_kernel<float, false>), g, dim3(256), 0, st, (const float*)x, H, W, C, Cs, OH, OW, out, argmax, total);
  else if (dtype == SPCL_F32)
    SPCL_LAUNCH((adaptive_pool_fwd_kernel<float, true>), g, dim3(256), 0, st, (const float*)x, H, W, C, Cs, OH, OW, out, argmax, total);
  else if (dtype == SPCL_BF16 && mode == 0)
    SPCL_LAUNCH((adaptive_pool_fwd_kernel<bf16_t, false>), g, dim3(256), 0, st, (const bf16_t*)x, H, W, C, Cs, OH, OW, out, argmax, total);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH((adaptive_pool_fwd_kernel<bf16_t, true>), g, dim3(256), 0, st, (const bf16_t*)x, H, W, C, Cs, OH, OW, out, argmax, total);
  else {
    set_error("adaptive_pool2d_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("adaptive_pool2d_forward");
  return SPCL_OK;
}

extern "C" int spcl_adaptive_pool2d_backward(const float* dout, const int* argmax, int dtype, int N, int H, int W, int C,
                                             int Cs, int OH, int OW, int mode, void* dx, void* stream) {
  SPCL_CHECK_ARG(dout && dx, "adaptive_pool2d_backward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Cs >= C && OH > 0 && OW > 0, "adaptive_pool2d_backward: bad shape");
  SPCL_CHECK_ARG(mode == 0 || (mode == 1 && argmax), "adaptive_pool2d_backward: mode %d", mode);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W * Cs;
  dim3 g((unsigned)((total + 255) / 256));
  if (mode == 0 && OH == 1 && OW == 1 && (dtype == SPCL_F32 || dtype == SPCL_BF16)) {  // global average: one broadcast
    if (dtype == SPCL_F32) SPCL_LAUNCH(avgpool_bwd_kernel<float>, g, dim3(256), 0, st, dout, H * W, C, Cs, (float*)dx, total);
    else SPCL_LAUNCH(avgpool_bwd_kernel<bf16_t>, g, dim3(256), 0, st, dout, H * W, C, Cs, (bf16_t*)dx, total);
    SPCL_LAUNCH_CHECK("adaptive_pool2d_backward");
    return SPCL_OK;
  }
  if (C % 4 == 0 && Cs % 4 == 0 && (dtype == SPCL_F32 || dtype == SPCL_BF16) && ((uintptr_t)dout % 16 == 0) &&
      ((uintptr_t)dx % 16 == 0) && (argmax == nullptr || (uintptr_t)argmax % 16 == 0)) {
    const int spr = (W + 7) / 8;  // segments of eight pixels per image row (APB_SEG)
    const size_t nseg = (size_t)N * H * spr;
    dim3 pg((unsigned)((nseg + 3) / 4));
    if (dtype == SPCL_F32 && mode == 0)
      SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<float, false>), pg, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (float*)dx, nseg, spr);
    else if (dtype == SPCL_F32)
      SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<float, true>), pg, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (float*)dx, nseg, spr);
    else if (mode == 0)
      SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<bf16_t, false>), pg, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (bf16_t*)dx, nseg, spr);
    else
      SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<bf16_t, true>), pg, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (bf16_t*)dx, nseg, spr);
    SPCL_LAUNCH_CHECK("adaptive_pool2d_backward");
    return SPCL_OK;
  }
  if (dtype == SPCL_F32 && mode == 0)
    SPCL_LAUNCH((adaptive_pool_bwd_kernel<float, false>), g, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (float*)dx, total);
  else if (dtype == SPCL_F32)
    SPCL_LAUNCH((adaptive_pool_bwd_kernel<float, true>), g, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (float*)dx, total);
  else if (dtype == SPCL_BF16 && mode == 0)
    SPCL_LAUNCH((adaptive_pool_bwd_kernel<bf16_t, false>), g, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (bf16_t*)dx, total);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH((adaptive_pool_bwd_kernel<bf16_t, true>), g, dim3(256), 0, st, dout, argmax, H, W, C, Cs, OH, OW, (bf16_t*)dx, total);
  else {
    set_error("adaptive_pool2d_backward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("adaptive_pool2d_backward");
  return SPCL_OK;
}

static CPtrs c1(const float* p) { CPtrs t{}; t.p[0] = p; return t; }
static MPtrs m1(float* p) { MPtrs t{}; t.p[0] = p; return t; }

extern "C" int spcl_l2norm_rows_forward(const float* x, size_t rows, int O, float* z, void* stream) {
  SPCL_CHECK_ARG(x && z && rows > 0 && O > 0, "l2norm_rows_forward: bad argument");
  SPCL_LAUNCH(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c1(x), (int)rows, O, m1(z));
  SPCL_LAUNCH_CHECK("l2norm_rows_forward");
  return SPCL_OK;
}

extern "C" int spcl_l2norm_rows_backward(const float* x, const float* dz, size_t rows, int O, float* dx, void* stream) {
  SPCL_CHECK_ARG(x && dz && dx && rows > 0 && O > 0, "l2norm_rows_backward: bad argument");
  SPCL_LAUNCH(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c1(x), c1(dz), (int)rows, O, m1(dx));
  SPCL_LAUNCH_CHECK("l2norm_rows_backward");
  return SPCL_OK;
}

// K heads of one shape on the same feature: average pool once, then one launch per layer for all heads
static int proj_heads_forward(int K, const void* feat, int dtype, int N, int HW, int C, int Cs, const float* const* w1,
                              const float* const* b1, const float* const* w2, const float* const* b2, int hid, int out_dim,
                              int normalize, float* pooled, float* const* pre, float* const* o, float* const* z,
                              hipStream_t st, const char* who) {
  dim3 pg(cdiv(C, 64), N);
  if (feat == nullptr) {
    // `pooled` was filled by the producer of the feature map (spcl_bnrelu_gap_forward): nothing to read back
  } else if (dtype == SPCL_F32)
    SPCL_LAUNCH(avgpool_kernel<float>, pg, dim3(256), 0, st, (const float*)feat, HW, C, Cs, pooled);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH(avgpool_kernel<bf16_t>, pg, dim3(256), 0, st, (const bf16_t*)feat, HW, C, Cs, pooled);
  else {
    set_error("%s: dtype %d", who, dtype);
    return SPCL_EINVAL;
  }
  CPtrs P{}, W1{}, B1{}, W2{}, B2{}, PRE{}, O{};
  MPtrs PREm{}, Om{}, Zm{};
  for (int k = 0; k < K; ++k) {
    P.p[k] = pooled; W1.p[k] = w1[k]; B1.p[k] = b1[k];
    if (hid > 0) { W2.p[k] = w2[k]; B2.p[k] = b2[k]; PRE.p[k] = pre[k]; PREm.p[k] = pre[k]; }
    O.p[k] = o[k]; Om.p[k] = o[k]; Zm.p[k] = z[k];
  }
  const int ny = (N + 7) / 8 < 8 ? (N + 7) / 8 : 8;  // 8 rows per block iteration
  if (hid > 0) {
    SPCL_LAUNCH(linear_fwd_kernel<false>, dim3(cdiv(hid, 4), ny, K), dim3(256), 0, st, P, W1, B1, N, C, hid, PREm);
    SPCL_LAUNCH(linear_fwd_kernel<true>, dim3(cdiv(out_dim, 4), ny, K), dim3(256), 0, st, PRE, W2, B2, N, hid, out_dim, Om);
  } else {
    SPCL_LAUNCH(linear_fwd_kernel<false>, dim3(cdiv(out_dim, 4), ny, K), dim3(256), 0, st, P, W1, B1, N, C, out_dim, Om);
  }
  if (normalize)
    SPCL_LAUNCH(l2norm_fwd_kernel, dim3(cdiv(N, 4), 1, K), dim3(256), 0, st, O, N, out_dim, Zm);
  else
    for (int k = 0; k < K; ++k)
      if (z[k] != o[k])  // (the caller may pass z == o: the rows before normalisation ARE the output then, no copy)
        (void)hipMemcpyAsync(z[k], o[k], (size_t)N * out_dim * sizeof(float), hipMemcpyDeviceToDevice, st);
  return SPCL_OK;
}

static int proj_heads_backward(int K, const float* const* dz, int dtype, int N, int HW, int C, int Cs,
                               const float* const* w1, const float* const* w2, int hid, int out_dim, int normalize,
                               const float* pooled, const float* const* pre, const float* const* o, float* const* dw1,
                               float* const* db1, float* const* dw2, float* const* db2, float* scratch, void* dfeat,
                               hipStream_t st, const char* who, bool nc_only = false /* dfeat is [N][Cs] (Cs == C) */) {
  const PooledOut po{nc_only ? dfeat : nullptr, dtype, HW};
  // scratch: [K][N,out] d_o | [K][N,hid] dpre | [N,C] dpool
  float* d_o = scratch;
  float* dpre = d_o + (size_t)K * N * out_dim;
  float* dpool = dpre + (size_t)K * N * (hid > 0 ? hid : 0);
  CPtrs DZ{}, O{}, GO{}, PRE{}, W1{}, W2{}, DPRE{}, P{};
  MPtrs DOm{}, DPREm{}, DW1{}, DB1{}, DW2{}, DB2{}, DPOOL{};
  for (int k = 0; k < K; ++k) {
    DZ.p[k] = dz[k]; O.p[k] = o[k]; W1.p[k] = w1[k]; P.p[k] = pooled;
    DOm.p[k] = d_o + (size_t)k * N * out_dim;
    GO.p[k] = normalize ? DOm.p[k] : dz[k];
    DW1.p[k] = dw1[k]; DB1.p[k] = db1[k];
    if (hid > 0) {
      PRE.p[k] = pre[k]; W2.p[k] = w2[k]; DW2.p[k] = dw2[k]; DB2.p[k] = db2[k];
      DPREm.p[k] = dpre + (size_t)k * N * hid; DPRE.p[k] = DPREm.p[k];
    }
  }
  DPOOL.p[0] = dpool;
  if (normalize) SPCL_LAUNCH(l2norm_bwd_kernel, dim3(cdiv(N, 4), 1, K), dim3(256), 0, st, O, DZ, N, out_dim, DOm);
  if (hid > 0) {
    {  // layer 2: dW2 / db2 and the hidden gradient (through the leaky ReLU) in one launch
      const int gxw = cdiv(hid, 256), nwb = gxw * out_dim, gxd = cdiv(hid, 64);
      SPCL_LAUNCH(linear_bwd_pair_kernel<true>, dim3(nwb + gxd * N, 1, K), dim3(256), 0, st, GO, PRE, W2, PRE, N, hid, out_dim,
                  DW2, DB2, DPREm, 0, gxw, nwb, gxd);
    }
    if (dfeat) {  // layer 1: dW1 / db1 and the pooled feature's gradient (summed over the heads) in one launch
      const int gxw = cdiv(C, 256), nwb = gxw * hid, gxd = cdiv(C, 64);
      SPCL_LAUNCH(linear_bwd_pair_kernel<false>, dim3(nwb + gxd * N, 1, K), dim3(256), 0, st, DPRE, P, W1, CPtrs{}, N, C, hid,
                  DW1, DB1, DPOOL, K, gxw, nwb, gxd, po);
    } else {
      SPCL_LAUNCH(linear_wgrad_kernel<false>, dim3(cdiv(C, 256), hid, K), dim3(256), 0, st, DPRE, P, N, C, hid, DW1, DB1);
    }
  } else {
    if (dfeat) {
      const int gxw = cdiv(C, 256), nwb = gxw * out_dim, gxd = cdiv(C, 64);
      SPCL_LAUNCH(linear_bwd_pair_kernel<false>, dim3(nwb + gxd * N, 1, K), dim3(256), 0, st, GO, P, W1, CPtrs{}, N, C, out_dim,
                  DW1, DB1, DPOOL, K, gxw, nwb, gxd, po);
    } else {
      SPCL_LAUNCH(linear_wgrad_kernel<false>, dim3(cdiv(C, 256), out_dim, K), dim3(256), 0, st, GO, P, N, C, out_dim, DW1, DB1);
    }
  }
  if (dfeat && !nc_only) {
    const size_t total = (size_t)N * HW * Cs;
    dim3 g((unsigned)((total + 255) / 256));
    if (dtype == SPCL_F32)
      SPCL_LAUNCH(avgpool_bwd_kernel<float>, g, dim3(256), 0, st, (const float*)dpool, HW, C, Cs, (float*)dfeat, total);
    else if (dtype == SPCL_BF16)
      SPCL_LAUNCH(avgpool_bwd_kernel<bf16_t>, g, dim3(256), 0, st, (const float*)dpool, HW, C, Cs, (bf16_t*)dfeat, total);
    else {
      set_error("%s: dtype %d", who, dtype);
      return SPCL_EINVAL;
    }
  }
  return SPCL_OK;
}

extern "C" int spcl_proj_forward(const void* feat, int dtype, int N, int HW, int C, int Cs, const float* w1,
                                 const float* b1, const float* w2, const float* b2, int hid, int out_dim,
                                 int normalize, float* pooled, float* pre, float* o, float* z, void* stream) {
  SPCL_CHECK_ARG(w1 && b1 && pooled && o && z, "proj_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0, "proj_forward: bad shape");
  SPCL_CHECK_ARG(hid == 0 || (w2 && b2 && pre), "proj_forward: mlp head needs w2/b2/pre");
  SPCL_CHECK_ARG(C <= 512 && hid <= 512, "proj_forward: at most 512 input / hidden features (C=%d, hid=%d)", C, hid);
  const int rc = proj_heads_forward(1, feat, dtype, N, HW, C, Cs, &w1, &b1, &w2, &b2, hid, out_dim, normalize, pooled, &pre,
                                    &o, &z, (hipStream_t)stream, "proj_forward");
  if (rc != SPCL_OK) return rc;
  SPCL_LAUNCH_CHECK("proj_forward");
  return SPCL_OK;
}

extern "C" int spcl_proj_backward(const float* dz, int dtype, int N, int HW, int C, int Cs, const float* w1,
                                  const float* w2, int hid, int out_dim, int normalize, const float* pooled,
                                  const float* pre, const float* o, float* dw1, float* db1, float* dw2, float* db2,
                                  float* scratch, void* dfeat, void* stream) {
  SPCL_CHECK_ARG(dz && w1 && pooled && (o || !normalize) && dw1 && db1 && scratch, "proj_backward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0, "proj_backward: bad shape");
  SPCL_CHECK_ARG(hid == 0 || (w2 && pre && dw2 && db2), "proj_backward: mlp head needs w2/pre/dw2/db2");
  const int rc = proj_heads_backward(1, &dz, dtype, N, HW, C, Cs, &w1, &w2, hid, out_dim, normalize, pooled, &pre, &o, &dw1,
                                     &db1, &dw2, &db2, scratch, dfeat, (hipStream_t)stream, "proj_backward");
  if (rc != SPCL_OK) return rc;
  SPCL_LAUNCH_CHECK("proj_backward");
  return SPCL_OK;
}

// K <= 4 heads of identical shape on the SAME feature (several meta-label hooks on one encoder tap, SURVEY row N4;
// hooks/creator.py:102-124): pointer arrays of length K (host memory), everything else as the single-head calls.
// scratch: K * N * (out_dim + hid) + N * C floats.
extern "C" int spcl_proj_heads_forward(int K, const void* feat, int dtype, int N, int HW, int C, int Cs,
                                       const float* const* w1, const float* const* b1, const float* const* w2,
                                       const float* const* b2, int hid, int out_dim, int normalize, float* pooled,
                                       float* const* pre, float* const* o, float* const* z, void* stream) {
  SPCL_CHECK_ARG(K >= 1 && K <= PROJ_MAX_HEADS, "proj_heads_forward: %d heads (1..%d)", K, PROJ_MAX_HEADS);
  SPCL_CHECK_ARG(w1 && b1 && pooled && o && z && (hid == 0 || (w2 && b2 && pre)), "proj_heads_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0 && C <= 512 && hid <= 512,
                 "proj_heads_forward: bad shape");
  for (int k = 0; k < K; ++k)
    SPCL_CHECK_ARG(w1[k] && b1[k] && o[k] && z[k] && (hid == 0 || (w2[k] && b2[k] && pre[k])),
                   "proj_heads_forward: null pointer in head %d", k);
  const int rc = proj_heads_forward(K, feat, dtype, N, HW, C, Cs, w1, b1, w2, b2, hid, out_dim, normalize, pooled, pre, o, z,
                                    (hipStream_t)stream, "proj_heads_forward");
  if (rc != SPCL_OK) return rc;
  SPCL_LAUNCH_CHECK("proj_heads_forward");
  return SPCL_OK;
}

// spcl_proj_heads_backward with the feature gradient as ONE value per (image, channel): dfeat_nc [N][Cs] of dtype, Cs == C
// (= dpooled / HW, what every pixel of that image and channel would receive).  K = 1: the single head.
extern "C" int spcl_proj_heads_backward_pooled(int K, const float* const* dz, int dtype, int N, int HW, int C, int Cs,
                                               const float* const* w1, const float* const* w2, int hid, int out_dim,
                                               int normalize, const float* pooled, const float* const* pre,
                                               const float* const* o, float* const* dw1, float* const* db1,
                                               float* const* dw2, float* const* db2, float* scratch, void* dfeat_nc,
                                               void* stream) {
  SPCL_CHECK_ARG(K >= 1 && K <= PROJ_MAX_HEADS, "proj_heads_backward_pooled: %d heads (1..%d)", K, PROJ_MAX_HEADS);
  SPCL_CHECK_ARG(dz && w1 && pooled && o && dw1 && db1 && scratch && dfeat_nc && (hid == 0 || (w2 && pre && dw2 && db2)),
                 "proj_heads_backward_pooled: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs == C && out_dim > 0 && hid >= 0 && (dtype == SPCL_F32 || dtype == SPCL_BF16),
                 "proj_heads_backward_pooled: bad shape (Cs must equal C)");
  for (int k = 0; k < K; ++k)
    SPCL_CHECK_ARG(dz[k] && w1[k] && (o[k] || !normalize) && dw1[k] && db1[k] &&
                       (hid == 0 || (w2[k] && pre[k] && dw2[k] && db2[k])),
                   "proj_heads_backward_pooled: null pointer in head %d", k);
  const int rc = proj_heads_backward(K, dz, dtype, N, HW, C, Cs, w1, w2, hid, out_dim, normalize, pooled, pre, o, dw1, db1,
                                     dw2, db2, scratch, dfeat_nc, (hipStream_t)stream, "proj_heads_backward_pooled", true);
  if (rc != SPCL_OK) return rc;
  SPCL_LAUNCH_CHECK("proj_heads_backward_pooled");
  return SPCL_OK;
}

extern "C" int spcl_proj_heads_backward(int K, const float* const* dz, int dtype, int N, int HW, int C, int Cs,
                                        const float* const* w1, const float* const* w2, int hid, int out_dim,
                                        int normalize, const float* pooled, const float* const* pre,
                                        const float* const* o, float* const* dw1, float* const* db1, float* const* dw2,
                                        float* const* db2, float* scratch, void* dfeat, void* stream) {
  SPCL_CHECK_ARG(K >= 1 && K <= PROJ_MAX_HEADS, "proj_heads_backward: %d heads (1..%d)", K, PROJ_MAX_HEADS);
  SPCL_CHECK_ARG(dz && w1 && pooled && o && dw1 && db1 && scratch && (hid == 0 || (w2 && pre && dw2 && db2)),
                 "proj_heads_backward: null pointer");
  SPCL_CHECK_ARG(N > 0 && HW > 0 && C > 0 && Cs >= C && out_dim > 0 && hid >= 0, "proj_heads_backward: bad shape");
  for (int k = 0; k < K; ++k)
    SPCL_CHECK_ARG(dz[k] && w1[k] && (o[k] || !normalize) && dw1[k] && db1[k] &&
                       (hid == 0 || (w2[k] && pre[k] && dw2[k] && db2[k])),
                   "proj_heads_backward: null pointer in head %d", k);
  const int rc = proj_heads_backward(K, dz, dtype, N, HW, C, Cs, w1, w2, hid, out_dim, normalize, pooled, pre, o, dw1, db1,
                                     dw2, db2, scratch, dfeat, (hipStream_t)stream, "proj_heads_backward");
  if (rc != SPCL_OK) return rc;
  SPCL_LAUNCH_CHECK("proj_heads_backward");
  return SPCL_OK;
}

// adaptive average pooling backward THROUGH a LeakyReLU: dx = unpool(dout) * LeakyReLU'(pre), with act = LeakyReLU(pre) [N][H][W][C]
// f32 given instead of pre (same sign).  The dense projector's pooled-hidden form (functional._PixelMlpPooledFn).  C % 4 == 0.
extern "C" int spcl_adaptive_avgpool2d_backward_act(const float* dout, const void* act, int dtype, int N, int H, int W, int C, int OH,
                                                    int OW, void* dx, void* stream) {
  SPCL_CHECK_ARG(dout && act && dx, "adaptive_avgpool2d_backward_act: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && OH > 0 && OW > 0, "adaptive_avgpool2d_backward_act: bad shape");
  SPCL_CHECK_ARG(dtype == SPCL_F32 || dtype == SPCL_BF16, "adaptive_avgpool2d_backward_act: dtype %d", dtype);
  const int spr = (W + 7) / 8;
  const size_t nseg = (size_t)N * H * spr;
  const dim3 grid((unsigned)((nseg + 3) / 4));
  if (dtype == SPCL_F32)
    SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, dout, (const int*)nullptr, H, W,
                C, C, OH, OW, (float*)dx, nseg, spr, (const float*)act);
  else
    SPCL_LAUNCH((adaptive_pool_bwd_pix_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, dout, (const int*)nullptr, H,
                W, C, C, OH, OW, (bf16_t*)dx, nseg, spr, (const bf16_t*)act);
  SPCL_LAUNCH_CHECK("adaptive_avgpool2d_backward_act");
  return SPCL_OK;
}
