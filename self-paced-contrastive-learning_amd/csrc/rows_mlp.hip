// The pixel-wise (1x1-convolution) MLP of the dense projector (contrastyou/projectors/heads.py:28-39,96-120:
// Conv2d(C, hid, 1) -> LeakyReLU(0.01) -> Conv2d(hid, out, 1) on EVERY pixel of a decoder feature map) as
// matrix products over the pixels: 753 000 rows at Up_conv3 (112^2) and 3 million at Up_conv2 (224^2) for a 60-image batch.  Rounds 2 - 5
// ran these rows through the global projector's kernels (projector.hip: one wave per output column walking all rows, built
// for 64 rows), which re-read the whole input once per four output columns: 30 / 24 / 54 / 46 ms per launch, 158 ms per
// training step at Up_conv3 and 630 ms at Up_conv2 (tools/diag/dense_step_time.py).  Here: three tiled products on the
// exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32: operands and accumulation in f32, as the reference's conv), one
// tile-staging scheme for all of them:
//   forward            Y[M][N]  = act(X)[M][K] W[N][K]^T + b[N]
//   input gradient     D[M][K]  = (G[M][N] W[N][K]) . act'(P[M][K])
//   weight gradient    dW[N][K] = sum_m G[m][N] act(X)[m][K],  db[N] = sum_m G[m][N]   (slabs of rows, fixed-order fold)
// act = LeakyReLU(0.01) applied to the operand on load (X is then the saved pre-activation), act' its derivative.
#include "common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(4))) float rm_f32x4;
constexpr float kRowsLeaky = 0.01f;
constexpr int RM_KC = 32, RM_KP = RM_KC + 4;  // k per staged chunk; LDS row pitch (16-byte aligned rows)

__device__ __forceinline__ float rm_act(float v, bool leaky) { return (!leaky || v > 0.f) ? v : kRowsLeaky * v; }

// four consecutive elements of a row as floats, or zeros when `ok` is false.  Branch-free: the load is issued from a valid
// address either way (offset 0 of the tensor) and the result selected -- with a branch per load the compiler waited for every
// load before it issued the next one (six memory round trips per k-chunk: the first version of these kernels ran at a
// tenth of their present rate).  The extents are multiples of 4 (checked by the entry points): a chunk is whole or absent.
template <typename T>
__device__ __forceinline__ rm_f32x4 rm_load4(const T* base, long off, bool ok) {
  const T* p = base + (ok ? off : 0L);
  rm_f32x4 v;
  if (sizeof(T) == 4) {
    v = *(const rm_f32x4*)p;
  } else {
    const uint2 raw = *(const uint2*)p;
    v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
    v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
  return v;
}

// One k-chunk of products: the wave's MT x NT tiles of 16 x 16 outputs from LDS tiles As[rows][RM_KP], Bs[cols][RM_KP] holding
// RM_KC values of the reduction index per row.  Lane (r16, g) reads four consecutive k of "its" row of each tile (16 s + 4 g ..
// + 3) and feeds element e of both to MFMA e: operand A and operand B walk the reduction index in the same order, which is
// all the instruction needs.  D: row 4 g + r, column r16.
template <int MT, int NT>
__device__ __forceinline__ void rm_chunk(const float* As, const float* Bs, int arow0, rm_f32x4 (&acc)[MT][NT], int r16, int g) {
#pragma unroll
  for (int s = 0; s < RM_KC / 16; ++s) {
    rm_f32x4 a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *(const rm_f32x4*)(As + (arow0 + 16 * i + r16) * RM_KP + 16 * s + 4 * g);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *(const rm_f32x4*)(Bs + (16 * j + r16) * RM_KP + 16 * s + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  }
}

// ---- forward / input gradient: a 128 x 64 output tile per workgroup, four waves of 32 rows each
// B_T = false: Bs[col][k] = W[n0 + col][k0 + k]  (W is [N][K]: the forward)
// B_T = true:  Bs[col][k] = W[k0 + k][n0 + col]  (W is [Kred][N]: the input gradient, reduction over W's rows)
template <typename TX, typename TY, bool LEAKY_IN, bool B_T, bool MASK, bool LEAKY_OUT = false, int NT = 4>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const TX* __restrict__ X, long ldx, const float* __restrict__ W, long ldw,
                                                        const float* __restrict__ bias, const float* __restrict__ P, long ldp,
                                                        int M, int K, int N, TY* __restrict__ Y, long ldy) {
  // NT: n-tiles of 16 output columns per workgroup (BN = 16, 32 or 64: the input gradient of a 16- or 32-channel map has no
  // more columns than that -- with the 64-wide tile three quarters of its products multiplied padding).
  // One buffer: the two operand tiles during the k-loop, then the 128 x BN output tile on its way out (row pitch RM_CP)
  constexpr int BN = 16 * NT, RM_CP = BN + 4, CPR = BN / 4;
  constexpr int SM = (128 + BN) * RM_KP > 128 * RM_CP ? (128 + BN) * RM_KP : 128 * RM_CP;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  float* const As = smem;
  float* const Bs = smem + 128 * RM_KP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g = lane >> 4;
  // 1-D grid, the column block fastest: the workgroups that share a row block (the same A tile, the other pieces of the same
  // output rows) are dispatched together
  const int ncb = (N + BN - 1) / BN;
  const long m0 = (long)(blockIdx.x / ncb) * 128;
  const int n0 = (int)(blockIdx.x % ncb) * BN;
  rm_f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (rm_f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += RM_KC) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // A tile: 128 rows x 8 chunks of four
      const int c = t + 256 * it, row = c >> 3, ch = c & 7;
      const long m = m0 + row;
      const int k = k0 + 4 * ch;
      rm_f32x4 v = rm_load4<TX>(X, m * ldx + k, m < M && k < K);
      if (LEAKY_IN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rm_act(v[e], true);
      }
      *(rm_f32x4*)(As + row * RM_KP + 4 * ch) = v;
    }
    if (!B_T) {
#pragma unroll
      for (int c = t; c < BN * 8; c += 256) {  // B tile: BN output columns x 8 chunks of four along k
        const int col = c >> 3, ch = c & 7;
        const int n = n0 + col, k = k0 + 4 * ch;
        const rm_f32x4 v = rm_load4<float>(W, (long)n * ldw + k, n < N && k < K);
        *(rm_f32x4*)(Bs + col * RM_KP + 4 * ch) = v;
      }
    } else {
#pragma unroll
      for (int c = t; c < RM_KC * CPR; c += 256) {  // W rows are the reduction index: four columns of one row, stored transposed
        const int kk = c / CPR, ch = c % CPR;
        const int k = k0 + kk, n = n0 + 4 * ch;
        const rm_f32x4 v = rm_load4<float>(W, (long)k * ldw + n, k < K && n < N);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(4 * ch + e) * RM_KP + kk] = v[e];
      }
    }
    __syncthreads();
    rm_chunk<2, NT>(As, Bs, wave * 32, acc, r16, g);
    __syncthreads();
  }
  // The accumulators leave through LDS: a lane holds four ROWS of one column (the instruction's D layout), stored as they
  // are that is 32 scattered 4-byte stores per lane in 64-byte pieces.  Transposed through the tile buffer a thread writes
  // four consecutive columns (16 bytes, or 8 as bf16) and neighbouring threads one whole row of the tile.
  // (the k-loop's last barrier has passed: nobody reads the operand tiles any more)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) smem[(wave * 32 + 16 * i + 4 * g + r) * RM_CP + 16 * j + r16] = acc[i][j][r];
  __syncthreads();
#pragma unroll
  for (int c = t; c < 128 * CPR; c += 256) {
    const int row = c / CPR, ch = c % CPR;
    const long m = m0 + row;
    const int n = n0 + 4 * ch;
    const bool ok = m < M && n < N;  // (N is a multiple of 4: a chunk is whole or absent)
    rm_f32x4 v = *(const rm_f32x4*)(smem + row * RM_CP + 4 * ch);
    if (bias != nullptr) v += rm_load4<float>(bias, (long)n, n < N);
    if (MASK) {
      const rm_f32x4 pv = rm_load4<float>(P, m * ldp + n, ok);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= pv[e] > 0.f ? 1.f : kRowsLeaky;  // (torch: the slope at 0 is the negative one)
    }
    if (LEAKY_OUT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rm_act(v[e], true);
    }
    if (ok) {
      TY* o = Y + m * ldy + n;
      if (sizeof(TY) == 4) {
        *(rm_f32x4*)o = v;
      } else {
        uint2 w;
        w.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        w.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *(uint2*)o = w;
      }
    }
  }
}

// ---- weight gradient: a 64 x (16 NT) tile of dW per workgroup over one slab of rows; partial [slab][N][K] (+ [slab][N] for db)
template <typename TG, typename TX, bool LEAKY_IN, int NT = 4>
__global__ __launch_bounds__(256) void rows_wgrad_kernel(const TG* __restrict__ G, long ldg, const TX* __restrict__ X, long ldx,
                                                         int M, int N, int K, int slab_rows, int ktiles,
                                                         float* __restrict__ part, float* __restrict__ part_b) {
  __shared__ __attribute__((aligned(16))) float As[64 * RM_KP];  // G^T: [n][m]
  constexpr int BK = 16 * NT;
  __shared__ __attribute__((aligned(16))) float Bs[BK * RM_KP];  // act(X)^T: [k][m]
  __shared__ float red[16][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g = lane >> 4;
  const int nt_ = blockIdx.x / ktiles, kt_ = blockIdx.x - nt_ * ktiles;
  const int n0 = nt_ * 64, k0 = kt_ * BK;
  const long mb = (long)blockIdx.y * slab_rows;
  const long me = mb + slab_rows < (long)M ? mb + slab_rows : (long)M;
  rm_f32x4 acc[1][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[0][j] = (rm_f32x4){0.f, 0.f, 0.f, 0.f};
  rm_f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // this thread's four columns n0 + 4 (t % 16) + e over its rows (db, k-tile 0 only)
  for (long mc = mb; mc < me; mc += RM_KC) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {  // 32 rows x 16 chunks of four columns, stored transposed
      const int c = t + 256 * it, mm = c >> 4, ch = c & 15;
      const long m = mc + mm;
      {
        const int n = n0 + 4 * ch;
        const rm_f32x4 v = rm_load4<TG>(G, m * ldg + n, m < me && n < N);
        bsum += v;
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(4 * ch + e) * RM_KP + mm] = v[e];
      }
      if (4 * ch < BK) {  // (a narrower tile: the first BK / 4 chunk positions only)
        const int k = k0 + 4 * ch;
        const rm_f32x4 v = rm_load4<TX>(X, m * ldx + k, m < me && k < K);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(4 * ch + e) * RM_KP + mm] = rm_act(v[e], LEAKY_IN);
      }
    }
    __syncthreads();
    rm_chunk<1, NT>(As, Bs, wave * 16, acc, r16, g);
    __syncthreads();
  }
  float* o = part + (size_t)blockIdx.y * N * K;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int k = k0 + 16 * j + r16;
    if (k >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave * 16 + 4 * g + r;
      if (n < N) o[(size_t)n * K + k] = acc[0][j][r];
    }
  }
  if (kt_ == 0 && part_b != nullptr) {  // db: the 16 threads that staged the same four columns, added in index order
#pragma unroll
    for (int e = 0; e < 4; ++e) red[t >> 4][4 * (t & 15) + e] = bsum[e];
    __syncthreads();
    if (t < 64 && n0 + t < N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) s += red[w][t];
      part_b[(size_t)blockIdx.y * N + n0 + t] = s;
    }
  }
}

__global__ __launch_bounds__(256) void rows_wgrad_fold_kernel(const float* __restrict__ part, const float* __restrict__ part_b,
                                                              int nslab, int NK, int N, float* __restrict__ dW,
                                                              float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < NK) {
    float s = 0.f;
    for (int p = 0; p < nslab; ++p) s += part[(size_t)p * NK + i];
    dW[i] = s;
  } else if (i - NK < N && db != nullptr) {
    const int n = i - NK;
    float s = 0.f;
    for (int p = 0; p < nslab; ++p) s += part_b[(size_t)p * N + n];
    db[n] = s;
  }
}

// slabs of rows of the weight gradient: enough workgroups to fill the chip several times over (a workgroup is a chain of
// load -> barrier -> multiply -> barrier per 32 rows: with 1.5 workgroups per CU -- 96 slabs, the first version -- the launch
// was latency bound at 0.4 TB/s), bounded by the bytes of partial tiles the fold then reads (at most 32 MB)
static int rows_slabs(int M, int N, int K) {
  const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
  long s = (M + 511) / 512;                                   // at least 512 rows per slab
  const long want = (2048 + tiles - 1) / tiles;               // ~2 048 workgroups
  if (s > want) s = want;
  const long cap = (32L << 20) / ((long)N * K * 4 + N * 4);    // partial bytes
  if (s > cap) s = cap;
  return s < 1 ? 1 : (int)s;
}
static int rows_slab_rows(int M, int N, int K) {
  const int s = rows_slabs(M, N, K);
  const int r = (M + s - 1) / s;
  return (r + RM_KC - 1) / RM_KC * RM_KC;
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_rows_linear_forward(const void* x, int x_dtype, long ldx, int leaky_in, const float* W, const float* bias,
                                        int M, int K, int N, float* y, void* stream) {
  SPCL_CHECK_ARG(x && W && y, "rows_linear_forward: null pointer");
  SPCL_CHECK_ARG(M > 0 && K > 0 && N > 0 && ldx >= K && K % 4 == 0 && ldx % 4 == 0,
                 "rows_linear_forward: M, K, N > 0, K and the row pitch multiples of 4 (got %d, %d, %d, %ld)", M, K, N, ldx);
  SPCL_CHECK_ARG(x_dtype == SPCL_F32 || x_dtype == SPCL_BF16, "rows_linear_forward: dtype %d", x_dtype);
  SPCL_CHECK_ARG(!(leaky_in && x_dtype != SPCL_F32), "rows_linear_forward: a saved pre-activation is f32");
  const dim3 grid((unsigned)((size_t)((M + 127) / 128) * ((N + 63) / 64)));
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == SPCL_BF16)
    SPCL_LAUNCH((rows_gemm_kernel<bf16_t, float, false, false, false>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, W, (long)K,
                bias, nullptr, 0L, M, K, N, y, (long)N);
  else if (leaky_in)
    SPCL_LAUNCH((rows_gemm_kernel<float, float, true, false, false>), grid, dim3(256), 0, st, (const float*)x, ldx, W, (long)K, bias,
                nullptr, 0L, M, K, N, y, (long)N);
  else
    SPCL_LAUNCH((rows_gemm_kernel<float, float, false, false, false>), grid, dim3(256), 0, st, (const float*)x, ldx, W, (long)K, bias,
                nullptr, 0L, M, K, N, y, (long)N);
  SPCL_LAUNCH_CHECK("rows_linear_forward");
  return SPCL_OK;
}

// forward with the activation applied to the OUTPUT: h = LeakyReLU(x W^T + b).  (The sign of h is the pre-activation's, so a
// backward pass can take LeakyReLU' from h itself: spcl_rows_linear_backward_input's `pre` may be h.)
extern "C" int spcl_rows_linear_forward_act(const void* x, int x_dtype, long ldx, const float* W, const float* bias, int M, int K,
                                            int N, void* h, int h_dtype, void* stream) {
  SPCL_CHECK_ARG(x && W && h, "rows_linear_forward_act: null pointer");
  SPCL_CHECK_ARG(M > 0 && K > 0 && N > 0 && ldx >= K && K % 4 == 0 && ldx % 4 == 0 && N % 4 == 0,
                 "rows_linear_forward_act: M, K, N > 0, K, N and the row pitch multiples of 4 (got %d, %d, %d, %ld)", M, K, N, ldx);
  SPCL_CHECK_ARG((x_dtype == SPCL_F32 || x_dtype == SPCL_BF16) && (h_dtype == SPCL_F32 || h_dtype == SPCL_BF16),
                 "rows_linear_forward_act: dtypes %d, %d", x_dtype, h_dtype);
  const dim3 grid((unsigned)((size_t)((M + 127) / 128) * ((N + 63) / 64)));
  hipStream_t st = (hipStream_t)stream;
#define RM_FWD_ACT(TX_, TY_)                                                                                                   \
  SPCL_LAUNCH((rows_gemm_kernel<TX_, TY_, false, false, false, true>), grid, dim3(256), 0, st, (const TX_*)x, ldx, W, (long)K, bias, \
              nullptr, 0L, M, K, N, (TY_*)h, (long)N)
  if (x_dtype == SPCL_BF16 && h_dtype == SPCL_BF16) RM_FWD_ACT(bf16_t, bf16_t);
  else if (x_dtype == SPCL_BF16) RM_FWD_ACT(bf16_t, float);
  else if (h_dtype == SPCL_BF16) RM_FWD_ACT(float, bf16_t);
  else RM_FWD_ACT(float, float);
#undef RM_FWD_ACT
  SPCL_LAUNCH_CHECK("rows_linear_forward_act");
  return SPCL_OK;
}

extern "C" int spcl_rows_linear_backward_input(const void* g, int g_dtype, const float* W, const float* pre, int M, int N, int K,
                                               void* dx, int dx_dtype, long lddx, void* stream) {
  SPCL_CHECK_ARG(g && W && dx, "rows_linear_backward_input: null pointer");
  SPCL_CHECK_ARG(M > 0 && K > 0 && N > 0 && lddx >= K && N % 4 == 0 && K % 4 == 0,
                 "rows_linear_backward_input: M, K, N > 0, N and K multiples of 4 (got %d, %d, %d)", M, K, N);
  SPCL_CHECK_ARG((dx_dtype == SPCL_F32 || dx_dtype == SPCL_BF16) && (g_dtype == SPCL_F32 || g_dtype == SPCL_BF16),
                 "rows_linear_backward_input: dtypes %d, %d", g_dtype, dx_dtype);
  SPCL_CHECK_ARG(!(pre != nullptr && g_dtype != SPCL_F32), "rows_linear_backward_input: the masked form takes an f32 gradient");
  // D[M][K] = G[M][N] W[N][K]: the reduction runs over W's ROWS (B_T); its "output columns" are the K inputs
  hipStream_t st = (hipStream_t)stream;
  // (the output has K columns: 16- and 32-wide tiles for the narrow maps)
#define RM_DX_NT(TG_, TY_, MASK_, NT_)                                                                                            \
  SPCL_LAUNCH((rows_gemm_kernel<TG_, TY_, false, true, MASK_, false, NT_>),                                                       \
              dim3((unsigned)((size_t)((M + 127) / 128) * ((K + 16 * NT_ - 1) / (16 * NT_)))), dim3(256), 0, st, (const TG_*)g,    \
              (long)N, W, (long)K, nullptr, pre, (long)K, M, N, K, (TY_*)dx, lddx)
#define RM_DX(TG_, TY_, MASK_)                          \
  do {                                                  \
    if (K <= 16) RM_DX_NT(TG_, TY_, MASK_, 1);          \
    else if (K <= 32) RM_DX_NT(TG_, TY_, MASK_, 2);     \
    else RM_DX_NT(TG_, TY_, MASK_, 4);                  \
  } while (0)
  if (g_dtype == SPCL_BF16) {
    if (dx_dtype == SPCL_BF16) RM_DX(bf16_t, bf16_t, false);
    else RM_DX(bf16_t, float, false);
  } else if (dx_dtype == SPCL_BF16) {
    if (pre != nullptr) RM_DX(float, bf16_t, true);
    else RM_DX(float, bf16_t, false);
  } else {
    if (pre != nullptr) RM_DX(float, float, true);
    else RM_DX(float, float, false);
  }
#undef RM_DX
#undef RM_DX_NT
  SPCL_LAUNCH_CHECK("rows_linear_backward_input");
  return SPCL_OK;
}

extern "C" size_t spcl_rows_linear_backward_weight_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)rows_slabs(M, N, K) * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int spcl_rows_linear_backward_weight(const void* g, int g_dtype, const void* x, int x_dtype, long ldx, int leaky_in,
                                                int M, int N, int K, float* ws, size_t ws_bytes, float* dW, float* db,
                                                void* stream) {
  SPCL_CHECK_ARG(g && x && ws && dW, "rows_linear_backward_weight: null pointer");
  SPCL_CHECK_ARG(M > 0 && K > 0 && N > 0 && ldx >= K && N % 4 == 0 && K % 4 == 0 && ldx % 4 == 0,
                 "rows_linear_backward_weight: M, K, N > 0, N, K and the row pitch multiples of 4 (got %d, %d, %d)", M, K, N);
  SPCL_CHECK_ARG((x_dtype == SPCL_F32 || x_dtype == SPCL_BF16) && (g_dtype == SPCL_F32 || g_dtype == SPCL_BF16),
                 "rows_linear_backward_weight: dtypes %d, %d", g_dtype, x_dtype);
  SPCL_CHECK_ARG(!(leaky_in && x_dtype != SPCL_F32), "rows_linear_backward_weight: a saved pre-activation is f32");
  SPCL_CHECK_ARG(!(g_dtype == SPCL_BF16 && leaky_in), "rows_linear_backward_weight: a bf16 gradient comes without the activation");
  SPCL_CHECK_ARG(ws_bytes >= spcl_rows_linear_backward_weight_workspace_bytes(M, N, K),
                 "rows_linear_backward_weight: workspace of %zu bytes", spcl_rows_linear_backward_weight_workspace_bytes(M, N, K));
  const int bk = K <= 16 ? 16 : (K <= 32 ? 32 : 64);  // k-tile width (a narrow map's weight gradient has few columns)
  const int ns = rows_slabs(M, N, K), sr = rows_slab_rows(M, N, K), ktiles = (K + bk - 1) / bk, ntiles = (N + 63) / 64;
  const int nslab = (M + sr - 1) / sr;  // (slabs that hold rows: the rounded slab height may leave the last ones empty)
  float* part = ws;
  float* part_b = ws + (size_t)ns * N * K;
  const dim3 grid((unsigned)(ntiles * ktiles), (unsigned)nslab);
  hipStream_t st = (hipStream_t)stream;
#define RM_WG_NT(TG_, TX_, LK_, NT_)                                                                                                  \
  SPCL_LAUNCH((rows_wgrad_kernel<TG_, TX_, LK_, NT_>), grid, dim3(256), 0, st, (const TG_*)g, (long)N, (const TX_*)x, ldx, M, N, K, sr, \
              ktiles, part, part_b)
#define RM_WG(TG_, TX_, LK_)                      \
  do {                                            \
    if (bk == 16) RM_WG_NT(TG_, TX_, LK_, 1);     \
    else if (bk == 32) RM_WG_NT(TG_, TX_, LK_, 2); \
    else RM_WG_NT(TG_, TX_, LK_, 4);              \
  } while (0)
  if (g_dtype == SPCL_BF16) {
    if (x_dtype == SPCL_BF16) RM_WG(bf16_t, bf16_t, false);
    else RM_WG(bf16_t, float, false);
  } else if (x_dtype == SPCL_BF16) RM_WG(float, bf16_t, false);
  else if (leaky_in) RM_WG(float, float, true);
  else RM_WG(float, float, false);
#undef RM_WG
#undef RM_WG_NT
  const int total = N * K + N;
  SPCL_LAUNCH(rows_wgrad_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)part,
              (const float*)part_b, nslab, N * K, N, dW, db);
  SPCL_LAUNCH_CHECK("rows_linear_backward_weight");
  return SPCL_OK;
}
