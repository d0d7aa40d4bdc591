// Weight gradient of the 3x3 same-convolution (semi_seg/arch/unet.py:72,75, autograd backward) on gfx950 MFMA.
//   dW[tap][ci][co] = sum_pixels act(x)[p + tap][ci] * dy[p][co]
// GEMM per tap: D[m=ci][n=co] += A[m][k=pixel] * B[k=pixel][n]; both operands are pixel-major NHWC tiles in LDS, so
// the MFMA operands (K = pixels) are read TRANSPOSED: bf16 uses ds_read_b64_tr_b16 (hardware transpose, 4 pixels x
// 16 channels per 16-lane group), f32 reads one dword per lane.  The 9 taps are address offsets into the shared
// input halo tile.  Work split: workgroup = (pixel split, 32x32 / 16x16 channel block); its 4 waves own different
// (tap, ci-tile) output units (no cross-wave reduction), tiles are double-buffered in LDS with the next tile's global
// loads issued before the MFMAs of the current one; workgroup partials go to a workspace and a second kernel sums
// them in fixed order (deterministic, no float atomics) into the OIHW f32 gradient.
#include <stdlib.h>
#include "common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
template <typename T> struct Chunk;
template <> struct Chunk<float> { static constexpr int EPC = 4; };
template <> struct Chunk<bf16_t> { static constexpr int EPC = 8; };

constexpr int WG_TH = 16, WG_TW = 16, WG_HW = WG_TW + 2, WG_NHALO = (WG_TH + 2) * WG_HW, WG_NPIX = WG_TH * WG_TW;

struct WgradArgs {
  const void* x;
  const void* dy;
  const float* in_scale;
  const float* in_shift;
  float* partial;
  int N, H, W, CinS, CinK, CoutS, in_mode;
  int tilesX, tilesY, ntiles, nblk_ci, nblk_co;
  int dbuf;  // 1: LDS tile image double buffered (one barrier per tile); 0: single buffer, twice the residency
};

// relu(scale*v+shift) on one 16-byte chunk (same arithmetic as conv.hip's staging so masks agree bit-for-bit)
template <typename T> __device__ __forceinline__ u32x4 wg_bnrelu_chunk(u32x4 raw, const float* sc, const float* sh);
template <> __device__ __forceinline__ u32x4 wg_bnrelu_chunk<float>(u32x4 raw, const float* s, const float* b) {
  f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(s[e], v[e], b[e]), 0.f);
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 wg_bnrelu_chunk<bf16_t>(u32x4 raw, const float* s, const float* b) {
  u32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float lo = __uint_as_float(raw[e] << 16), hi = __uint_as_float(raw[e] & 0xffff0000u);
    lo = fmaxf(fmaf(s[2 * e], lo, b[2 * e]), 0.f);
    hi = fmaxf(fmaf(s[2 * e + 1], hi, b[2 * e + 1]), 0.f);
    out[e] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
  }
  return out;
}

// Operand fragment of one k-step for a 16-channel tile, read transposed from a [pixel][channel] LDS image.
//   bf16: k-step = 32 pixels; lane (g = lane>>4) gets pixels 8g..8g+7 of its channel (lane&15)
//   f32 : k-step = 4 pixels;  lane gets pixel g of its channel
// `lane_pixel(ks, lane)` = tile pixel whose address this lane supplies; the caller turns it into a byte address
// once per k-step, every tap / channel tile is then a compile-time offset folded into the DS instruction.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 type;
  static constexpr int KPIX = 32;
  static constexpr int SECOND = 4;  // second read: 4 pixels further
  static __device__ __forceinline__ int lane_pixel(int ks, int lane) {
    return ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
  }
  static __device__ __forceinline__ int lane_chan_bytes(int lane) { return (lane & 3) * 8; }
  // a0: address of (lane pixel, lane channel group); pix_bytes: LDS bytes per pixel
  static __device__ __forceinline__ type load(unsigned a0, int off, int pix_bytes) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + off + 4 * pix_bytes));
    type r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag<float> {
  typedef float type;
  static constexpr int KPIX = 4;
  static __device__ __forceinline__ int lane_pixel(int ks, int lane) { return ks * 4 + (lane >> 4); }
  static __device__ __forceinline__ int lane_chan_bytes(int lane) { return (lane & 15) * 4; }
  static __device__ __forceinline__ type load(unsigned a0, int off, int pix_bytes) {
    return *(const float __attribute__((address_space(3)))*)(uintptr_t)(a0 + off);
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

template <typename T, int MI, int NJ>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int EPC = Chunk<T>::EPC;
  constexpr int CIB = 16 * MI, COB = 16 * NJ;
  constexpr int XS = CIB * (int)sizeof(T), DS = COB * (int)sizeof(T);  // LDS bytes per pixel
  constexpr int XCP = CIB / EPC, DCP = COB / EPC;                       // 16-byte chunks per pixel
  constexpr int X_BYTES = WG_NHALO * XS, D_BYTES = WG_NPIX * DS, BUF_BYTES = X_BYTES + D_BYTES;
  constexpr int KSTEPS = WG_NPIX / Frag<T>::KPIX;
  constexpr int NX = (WG_NHALO * XCP + 255) / 256;  // staged 16-byte chunks per thread
  constexpr int ND = (WG_NPIX * DCP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 x [x halo | dy] (double buffer)
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.y;
  const int bci = blk / a.nblk_co, bco = blk - bci * a.nblk_co;
  const int ci0 = bci * CIB, co0 = bco * COB;
  // this thread always stages the same channel chunk (256 % XCP == 0): its BN scale/shift live in registers
  const int xch = threadIdx.x % XCP, dch = threadIdx.x % DCP;
  float sc[EPC], sh[EPC];
  if (a.in_mode == 1) {
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      *(f32x4*)&sc[e] = *(const f32x4*)(a.in_scale + ci0 + xch * EPC + e);
      *(f32x4*)&sh[e] = *(const f32x4*)(a.in_shift + ci0 + xch * EPC + e);
    }
  }

  // wave w owns the output units u = w, w+4, ... (u = tap*MI + ci-tile): no cross-wave reduction, every wave walks
  // all pixel k-steps of the tile for its own units
  constexpr int NUNITS = 9 * MI, UPW = (NUNITS + 3) / 4;
  f32x4 acc[UPW][NJ];
  int uoff[UPW];  // LDS byte offset of the unit's tap shift + channel tile inside the x halo image
#pragma unroll
  for (int uu = 0; uu < UPW; ++uu) {
    int u = wave + 4 * uu;
    if (u >= NUNITS) u = NUNITS - 1;  // clamped duplicate: computed, never written
    const int tap = u / MI, m = u - tap * MI;
    const int ky = tap / 3, kx = tap - 3 * ky;
    uoff[uu] = (ky * WG_HW + kx) * XS + m * 16 * (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[uu][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  u32x4 rx[NX], rd[ND];
  unsigned xmask = 0;  // staged x chunks that are inside the image (zero padding must stay zero after BN+ReLU)
  const int tpi = a.tilesX * a.tilesY;

  // ---- global -> registers (issued one tile ahead of the MFMAs: T14 issue-early / write-late)
  auto load_tile = [&](int tile) {
    const int n = tile / tpi;
    const int trem = tile - n * tpi;
    const int ty = trem / a.tilesX, tx = trem - ty * a.tilesX;
    const int y0 = ty * WG_TH, x0 = tx * WG_TW;
    xmask = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int q = idx / XCP;
      const int hy = q / WG_HW, hx = q - hy * WG_HW;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < WG_NHALO * XCP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t pix = ((size_t)n * a.H + gy) * a.W + gx;
        xmask |= 1u << i;
        if (a.in_mode == 2) {
          const float* src = (const float*)a.x + pix * a.CinS;
          float e[EPC];
#pragma unroll
          for (int k = 0; k < EPC; ++k) {
            const int c = xch * EPC + k;
            e[k] = c < a.CinS ? src[c] : 0.f;
          }
          if (sizeof(T) == 4) {
            v = (u32x4){__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]), __float_as_uint(e[3])};
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              v[k] = (uint32_t)f32_to_bf16(e[(2 * k) % EPC]) | ((uint32_t)f32_to_bf16(e[(2 * k + 1) % EPC]) << 16);
          }
        } else {
          v = *(const u32x4*)((const T*)a.x + pix * a.CinS + ci0 + xch * EPC);
        }
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int p = idx / DCP;
      const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < WG_NPIX * DCP && gy < a.H && gx < a.W) {
        const size_t pix = ((size_t)n * a.H + gy) * a.W + gx;
        v = *(const u32x4*)((const T*)a.dy + pix * a.CoutS + co0 + dch * EPC);
      }
      rd[i] = v;
    }
  };
  // ---- registers -> LDS buffer (fused BN-apply + ReLU of the producer layer on the in-image chunks)
  auto store_tile = [&](int buf) {
    unsigned char* bx = lds + buf * BUF_BYTES;
    unsigned char* bd = bx + X_BYTES;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < WG_NHALO * XCP) {
        u32x4 v = rx[i];
        if (a.in_mode == 1 && (xmask & (1u << i))) v = wg_bnrelu_chunk<T>(v, sc, sh);
        *(u32x4*)(bx + (idx / XCP) * XS + xch * 16) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < WG_NPIX * DCP) *(u32x4*)(bd + (idx / DCP) * DS + dch * 16) = rd[i];
    }
  };

  int tile = blockIdx.x;
  int buf = 0;
  if (tile < a.ntiles) load_tile(tile);
  while (tile < a.ntiles) {
    store_tile(buf);
    __syncthreads();  // also orders this buffer's previous readers (two iterations back) before the next overwrite
    const int next = tile + gridDim.x;
    if (next < a.ntiles) load_tile(next);

    const unsigned ldx_base = lds_base + buf * BUF_BYTES, ldd_base = ldx_base + X_BYTES;
#pragma unroll 2
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int p = Frag<T>::lane_pixel(ks, lane);  // a k-step's pixels never straddle a tile row (TW = 16)
      const unsigned xa = ldx_base + (unsigned)(((p >> 4) * WG_HW + (p & 15)) * XS + Frag<T>::lane_chan_bytes(lane));
      const unsigned da = ldd_base + (unsigned)(p * DS + Frag<T>::lane_chan_bytes(lane));
      typename Frag<T>::type bf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = Frag<T>::load(da, j * 16 * (int)sizeof(T), DS);
#pragma unroll
      for (int uu = 0; uu < UPW; ++uu) {
        typename Frag<T>::type af = Frag<T>::load(xa + uoff[uu], 0, XS);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[uu][j] = Frag<T>::mfma(af, bf[j], acc[uu][j]);
      }
    }
    if (a.dbuf) buf ^= 1;
    else __syncthreads();  // single buffer: all reads of this tile are done before the next one is written
    tile = next;
  }

  // ---- every wave writes its own units of the workgroup partial.  D layout: lane holds n = co (lane&15),
  // m = ci 4g+r.  slab layout [9][CIB][COB] f32
  const int r16 = lane & 15, g = lane >> 4;
  float* out = a.partial + ((size_t)blockIdx.x * gridDim.y + blk) * (9 * CIB * COB);
#pragma unroll
  for (int uu = 0; uu < UPW; ++uu) {
    const int u = wave + 4 * uu;
    if (u < NUNITS) {
      const int tap = u / MI, m = u - tap * MI;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[(tap * CIB + m * 16 + 4 * g + r) * COB + j * 16 + r16] = acc[uu][j][r];
    }
  }
}

// dW_oihw[co][ci][tap] = sum over pixel-split partials.  Threads follow the PARTIAL layout ([tap][ci][co], co fastest:
// coalesced 256-byte wave reads); the 4 waves of a block take splits p = w, w+4, ... (8 independent loads in flight
// each) and are combined through LDS in fixed order -> deterministic.  One scattered 4-byte store per output.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int nblk_ci,
                                                            int nblk_co, int CIB, int COB, int Cin, int Cout,
                                                            float* __restrict__ dw) {
  __shared__ float red[16][64];
  const int nwaves = blockDim.x >> 6;  // 4 or 16 split lanes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slab = 9 * CIB * COB;
  const int inner = blockIdx.x * 64 + lane;
  const int blk = blockIdx.y;
  const size_t nblk = (size_t)nblk_ci * nblk_co;
  float s = 0.f;
  if (inner < slab) {
    const float* src = partial + (size_t)blk * slab + inner;
    const size_t stride = nblk * slab;
#pragma unroll 8
    for (int p = wave; p < nsplit; p += nwaves) s += src[(size_t)p * stride];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && inner < slab) {
    float v = red[0][lane];
    for (int w = 1; w < nwaves; ++w) v += red[w][lane];  // fixed order
    const int co_l = inner % COB, ci_l = (inner / COB) % CIB, tap = inner / (COB * CIB);
    const int bci = blk / nblk_co, bco = blk - bci * nblk_co;
    const int ci = bci * CIB + ci_l, co = bco * COB + co_l;
    if (ci < Cin && co < Cout) dw[((size_t)co * Cin + ci) * 9 + tap] = v;
  }
}

struct WgradPlan {
  int MI, NJ, nblk_ci, nblk_co, nsplit, ntiles, tilesX, tilesY;
  size_t partial_floats;
};
static WgradPlan wgrad_plan(int N, int H, int W, int CinK, int CoutS, int esize) {
  WgradPlan p;
  p.NJ = CoutS >= 32 ? 2 : 1;
  p.MI = CinK >= 32 ? 2 : 1;
  p.nblk_ci = CinK / (16 * p.MI);
  p.nblk_co = CoutS / (16 * p.NJ);
  p.tilesX = cdiv(W, WG_TW);
  p.tilesY = cdiv(H, WG_TH);
  p.ntiles = N * p.tilesX * p.tilesY;
  const int nblk = p.nblk_ci * p.nblk_co;
  // exactly one resident "wave" of workgroups (LDS-limited residency x 256 CUs): measured optimum -- more workgroups
  // only add partial slabs and a tail, fewer leave CUs idle (tools/bench_kernels.py wgrad sweeps)
  const size_t lds = 2 * ((size_t)WG_NHALO * 16 * p.MI + (size_t)WG_NPIX * 16 * p.NJ) * esize;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  static const int env_wgs = getenv("SPCL_WGRAD_WGS") ? atoi(getenv("SPCL_WGRAD_WGS")) : 0;
  int ns = cdiv(env_wgs > 0 ? env_wgs : 256 * per_cu, nblk);
  if (ns > p.ntiles) ns = p.ntiles;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  p.partial_floats = (size_t)ns * nblk * 9 * (16 * p.MI) * (16 * p.NJ);
  return p;
}

template <typename T, int MI, int NJ>
static void launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  constexpr int CIB = 16 * MI, COB = 16 * NJ;
  size_t lds = (a.dbuf ? 2 : 1) * ((size_t)WG_NHALO * CIB * sizeof(T) + (size_t)WG_NPIX * COB * sizeof(T));
  if (lds > 65536)
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<T, MI, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, MI, NJ>), dim3(p.nsplit, p.nblk_ci * p.nblk_co), dim3(256), lds, st, a);
}

template <typename T>
static void launch_wgrad_t(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  if (p.MI == 1 && p.NJ == 1) launch_wgrad<T, 1, 1>(a, p, st);
  else if (p.MI == 1 && p.NJ == 2) launch_wgrad<T, 1, 2>(a, p, st);
  else if (p.MI == 2 && p.NJ == 1) launch_wgrad<T, 2, 1>(a, p, st);
  else launch_wgrad<T, 2, 2>(a, p, st);
}

}  // namespace spcl

using namespace spcl;

extern "C" size_t spcl_conv_wgrad_workspace_bytes(int N, int H, int W, int CinK, int CoutS) {
  if (N <= 0 || H <= 0 || W <= 0 || CinK <= 0 || CoutS <= 0 || CinK % 16 || CoutS % 16) return 0;
  // sized for the f32 plan (its split count is >= the bf16 one)
  const size_t a = wgrad_plan(N, H, W, CinK, CoutS, 4).partial_floats, b = wgrad_plan(N, H, W, CinK, CoutS, 2).partial_floats;
  return (a > b ? a : b) * sizeof(float);
}

extern "C" int spcl_conv3x3_wgrad(const void* x, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS,
                                  int CinK, int Cout, int CoutS, int in_mode, const float* in_scale,
                                  const float* in_shift, float* partial, float* dw_oihw, void* stream) {
  SPCL_CHECK_ARG(x && dy && partial && dw_oihw, "conv3x3_wgrad: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: bad shape");
  SPCL_CHECK_ARG(CinK % 16 == 0 && CoutS % 16 == 0 && Cin <= CinK && Cout <= CoutS, "conv3x3_wgrad: channel padding");
  SPCL_CHECK_ARG(in_mode >= 0 && in_mode <= 2, "conv3x3_wgrad: in_mode %d", in_mode);
  SPCL_CHECK_ARG(in_mode != 1 || (in_scale && in_shift), "conv3x3_wgrad: in_mode 1 needs scale/shift");
  if (in_mode == 2) SPCL_CHECK_ARG(CinK == 16 && CinS >= 1 && CinS <= 16, "conv3x3_wgrad: image mode needs Cin<=16");
  else SPCL_CHECK_ARG(CinS == CinK, "conv3x3_wgrad: CinS must equal CinK");
  hipStream_t st = (hipStream_t)stream;
  WgradPlan p = wgrad_plan(N, H, W, CinK, CoutS, dtype == SPCL_F32 ? 4 : 2);
  WgradArgs a;
  a.x = x; a.dy = dy; a.in_scale = in_scale; a.in_shift = in_shift; a.partial = partial;
  a.N = N; a.H = H; a.W = W; a.CinS = CinS; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = in_mode;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.ntiles = p.ntiles; a.nblk_ci = p.nblk_ci; a.nblk_co = p.nblk_co;
  static const int env_dbuf = getenv("SPCL_WGRAD_DBUF") ? atoi(getenv("SPCL_WGRAD_DBUF")) : 1;
  a.dbuf = env_dbuf;
  if (dtype == SPCL_F32) launch_wgrad_t<float>(a, p, st);
  else if (dtype == SPCL_BF16) launch_wgrad_t<bf16_t>(a, p, st);
  else {
    set_error("conv3x3_wgrad: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  const int slab = 9 * 16 * p.MI * 16 * p.NJ;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(slab, 64), p.nblk_ci * p.nblk_co), dim3(p.nsplit >= 64 ? 1024 : 256), 0, st,
                     (const float*)partial, p.nsplit, p.nblk_ci, p.nblk_co, 16 * p.MI, 16 * p.NJ, Cin, Cout, dw_oihw);
  SPCL_LAUNCH_CHECK("conv3x3_wgrad");
  return SPCL_OK;
}
