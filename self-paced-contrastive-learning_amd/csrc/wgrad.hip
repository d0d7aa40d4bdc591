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
#include <vector>
#include "conv_common.hpp"

namespace spcl {

// (u32x4, Chunk<T>, relu_bf16x2: conv_common.hpp)
constexpr int WG_TW = 16, WG_HW = WG_TW + 2;

struct WgradArgs {
  const void* x;
  const void* x2;  // non-null: input channels [xsplit, 2 xsplit) come from this tensor, [0, xsplit) from x; both dense
  int xsplit;      // [N][H][W][xsplit] (the decoder's concatenation read in place, spcl_conv3x3_wgrad_cat)
  int x_up2;       // 1: x is [N][H / 2][W / 2][CinS] and the layer's input its nearest x2 upsample (spcl_conv3x3_wgrad_up2)
  const void* dy;
  const float* in_scale;
  const float* in_shift;
  float* partial;
  int N, H, W, CinS, CinK, CoutS, in_mode;
  int tilesX, tilesY, ntiles, nblk_ci, nblk_co;
  int dbuf;  // 1: LDS tile image double buffered (one barrier per tile); 0: single buffer, twice the residency
  int xcd_remap;  // 1: a round's tiles are dealt to the XCDs in contiguous eighths (see the tile loop)
  unsigned long long* stamps;  // debug (SPCL_WGRAD_STAMPS=1): per-workgroup cycle counts of the loop phases, else null
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
    lo = fmaf(s[2 * e], lo, b[2 * e]);
    hi = fmaf(s[2 * e + 1], hi, b[2 * e + 1]);
    const f32x2 v = {lo, hi};  // (ReLU on the rounded pair: conv_common.hpp relu_bf16x2 -- the same bits)
    out[e] = relu_bf16x2(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v)));
  }
  return out;
}

template <> __device__ __forceinline__ u32x4 wg_bnrelu_chunk<split_f32>(u32x4 raw, const float* s, const float* b) {
  return wg_bnrelu_chunk<float>(raw, s, b);
}

// Operand fragment of one k-step for a 16-channel tile, read transposed from a [row][pixel][channel] LDS image.
//   bf16: k-step = 32 pixels = 2 tile rows x 16; lane group g = lane>>4 holds 8 of them (two hardware-transposed
//         4-pixel reads): row 2ks + (g&1), columns 8(g>>1) .. +7.  The 32 lanes a ds_read_b64_tr_b16 services together
//         (g = 0,1 / 2,3) therefore sit in DIFFERENT rows, and the row pitch is padded so that they hit complementary
//         banks (MI355X_MICROARCH.md LDS table; the former same-row mapping was a 2-way conflict on every read).
//   f32 : k-step = 4 pixels of one row; lane gets pixel g of its channel (exact-f32 parity path).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 type;
  static constexpr int KPIX = 32;
  static __device__ __forceinline__ int lane_row(int lane) { return (lane >> 4) & 1; }  // + 2 ks
  static __device__ __forceinline__ int lane_col(int lane) { return 8 * (lane >> 5) + ((lane & 15) >> 2); }
  static __device__ __forceinline__ int lane_chan_bytes(int lane) { return (lane & 3) * 8; }
  static constexpr int LSZ = 2, PLANES = 1;  // bytes per element in LDS, images per operand
  static __device__ __forceinline__ type load(unsigned a0, int off, int pix_bytes, int = 0) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(uintptr_t)(a0 + off + 4 * pix_bytes));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);  // register pair concatenation, no ALU
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // row pitch padding (bytes) for a [.. x 16*tiles channels] image whose rows hold `row_bytes`
  static constexpr int row_pad(int pix_bytes, int row_bytes) {
    // 4 consecutive pixels x 32 B are read at stride pix_bytes; the partner row must land on the other banks
    return pix_bytes == 32 ? (128 - row_bytes % 256 + 256) % 256 : (32 - row_bytes % 64 + 64) % 64;
  }
};
template <> struct Frag<float> {
  typedef float type;
  static constexpr int KPIX = 4;
  static __device__ __forceinline__ int lane_row(int) { return 0; }
  static __device__ __forceinline__ int lane_col(int lane) { return lane >> 4; }  // + 4 (ks % 4)
  static __device__ __forceinline__ int lane_chan_bytes(int lane) { return (lane & 15) * 4; }
  static constexpr int LSZ = 4, PLANES = 1;
  static __device__ __forceinline__ type load(unsigned a0, int off, int, int = 0) {
    return *(const float __attribute__((address_space(3)))*)(uintptr_t)(a0 + off);
  }
  static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static constexpr int row_pad(int, int) { return 0; }
};

// f32 tensors multiplied as three bf16 pieces (conv_common.hpp split3_pair; spcl_conv_set_f32_split): THREE bf16 images per
// operand in LDS (hi / mid / lo, each laid out exactly as the bf16 path's one -- same transposed reads, same bank analysis),
// filled by the staging from the f32 chunks; six bf16 MFMAs per fragment pair, the smallest products first.
struct Frag3 { bf16x8 h, m, l; };
template <> struct Frag<split_f32> {
  typedef Frag3 type;
  static constexpr int KPIX = 32, LSZ = 2, PLANES = 3;
  static __device__ __forceinline__ int lane_row(int lane) { return Frag<bf16_t>::lane_row(lane); }
  static __device__ __forceinline__ int lane_col(int lane) { return Frag<bf16_t>::lane_col(lane); }
  static __device__ __forceinline__ int lane_chan_bytes(int lane) { return Frag<bf16_t>::lane_chan_bytes(lane); }
  static __device__ __forceinline__ type load(unsigned a0, int off, int pix_bytes, int plane_bytes) {
    type f;
    f.h = Frag<bf16_t>::load(a0, off, pix_bytes);
    f.m = Frag<bf16_t>::load(a0 + plane_bytes, off, pix_bytes);
    f.l = Frag<bf16_t>::load(a0 + 2 * plane_bytes, off, pix_bytes);
    return f;
  }
  static __device__ __forceinline__ f32x4 mfma(const type& a, const type& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
  }
  static constexpr int row_pad(int pix_bytes, int row_bytes) { return Frag<bf16_t>::row_pad(pix_bytes, row_bytes); }
};

// NWV waves per workgroup: 4 for the 16/32-channel blocks; 8 for the 64x64 block (MI = NJ = 4), whose 36 (tap, ci-tile)
// units and 23 staging chunks per thread would not fit two waves per SIMD in a 4-wave workgroup
// IM: the input mode (0 raw, 1 BN + ReLU in the staging, 2 f32 image) and the debug stamps are compile-time: as run-time
// conditions inside the unrolled staging loops they were ~100 scalar branches per tile
#ifndef SPCL_WGRAD_STAMPS_BUILD
#define SPCL_WGRAD_STAMPS_BUILD 0
#endif
// four f32 values (one 16-byte chunk) -> 8 bytes in each of the three bf16 images
__device__ __forceinline__ void store_split(unsigned char* p, int plane_bytes, u32x4 v) {
  const Split3 a = split3_pair(__uint_as_float(v[0]), __uint_as_float(v[1]));
  const Split3 b = split3_pair(__uint_as_float(v[2]), __uint_as_float(v[3]));
  *(uint2*)p = make_uint2(a.hi, b.hi);
  *(uint2*)(p + plane_bytes) = make_uint2(a.mid, b.mid);
  *(uint2*)(p + 2 * plane_bytes) = make_uint2(a.lo, b.lo);
}

template <typename T, int MI, int NJ, int TH, int NWV, int IM>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int NTHR = 64 * NWV;
  constexpr int EPC = Chunk<T>::EPC;
  constexpr int LSZ = Frag<T>::LSZ, NPL = Frag<T>::PLANES, LCH = EPC * LSZ;  // LDS: bytes per element, images, bytes per chunk
  constexpr int CIB = 16 * MI, COB = 16 * NJ;
  constexpr int XS = CIB * LSZ, DS = COB * LSZ;          // LDS bytes per pixel
  constexpr int XCP = CIB / EPC, DCP = COB / EPC;        // 16-byte (global) chunks per pixel
  constexpr int XRP = WG_HW * XS + Frag<T>::row_pad(XS, WG_HW * XS);  // LDS bytes per halo row / dy row
  constexpr int DRP = WG_TW * DS + Frag<T>::row_pad(DS, WG_TW * DS);
  constexpr int NHROWS = TH + 2;
  constexpr int XPLB = NHROWS * XRP, DPLB = TH * DRP;    // bytes of one x / dy image
  constexpr int X_BYTES = NPL * XPLB, D_BYTES = NPL * DPLB, BUF_BYTES = X_BYTES + D_BYTES;
  constexpr int KSTEPS = TH * WG_TW / Frag<T>::KPIX;
  // staging maps: a thread owns one 16-byte channel chunk and one column; it walks the rows in compile-time steps
  constexpr int XPL = NTHR / XCP, XRPI = XPL / WG_HW, NX = (NHROWS + XRPI - 1) / XRPI;   // rows per iteration
  constexpr int DPL = NTHR / DCP, DRPI = DPL / WG_TW, ND = (TH + DRPI - 1) / DRPI;
  static_assert(XRPI >= 1 && DRPI >= 1, "channel block too wide for the staging map");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 x [x halo | dy] (double buffer)
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.y;
  const int bci = blk / a.nblk_co, bco = blk - bci * a.nblk_co;
  const int ci0 = bci * CIB, co0 = bco * COB;
  const int xch = threadIdx.x % XCP, xpl = threadIdx.x / XCP;
  const int xhy0 = xpl / WG_HW, xhx = xpl - xhy0 * WG_HW;
  const bool xact = xpl < XRPI * WG_HW;
  const int dch = threadIdx.x % DCP, dpl = threadIdx.x / DCP;
  const int dry0 = dpl / WG_TW, dcol = dpl - dry0 * WG_TW;
  float sc[EPC], sh[EPC];
  // (two input tensors with IM == 1: the coefficients are the SECOND tensor's -- x is an activation already, x2 the raw
  // output of the convolution whose BatchNorm + ReLU is applied here; the first tensor's chunks are staged as they are)
  const bool bn_x2 = IM == 1 && a.x2 != nullptr;
  const bool bn_this = IM == 1 && (!bn_x2 || ci0 + xch * EPC >= a.xsplit);
  if (IM == 1) {
    const int cc = bn_x2 ? (bn_this ? ci0 + xch * EPC - a.xsplit : 0) : ci0 + xch * EPC;
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      *(f32x4*)&sc[e] = *(const f32x4*)(a.in_scale + cc + e);
      *(f32x4*)&sh[e] = *(const f32x4*)(a.in_shift + cc + e);
    }
  }

  // wave w owns the output units u = w, w+4, ... (u = tap*MI + ci-tile): no cross-wave reduction, every wave walks
  // all pixel k-steps of the tile for its own units
  constexpr int NUNITS = 9 * MI, UPW = (NUNITS + NWV - 1) / NWV;
  f32x4 acc[UPW][NJ];
  int uoff[UPW];  // LDS byte offset of the unit's tap shift + channel tile inside the x halo image
#pragma unroll
  for (int uu = 0; uu < UPW; ++uu) {
    int u = wave + NWV * uu;
    if (u >= NUNITS) u = NUNITS - 1;  // clamped duplicate: computed, never written
    const int tap = u / MI, m = u - tap * MI;
    const int ky = tap / 3, kx = tap - 3 * ky;
    uoff[uu] = ky * XRP + kx * XS + m * 16 * LSZ;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[uu][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // per-lane operand addresses of k-step 0 (later k-steps add compile-time row / column offsets)
  const unsigned xlane = (unsigned)(Frag<T>::lane_row(lane) * XRP + Frag<T>::lane_col(lane) * XS +
                                    Frag<T>::lane_chan_bytes(lane));
  const unsigned dlane = (unsigned)(Frag<T>::lane_row(lane) * DRP + Frag<T>::lane_col(lane) * DS +
                                    Frag<T>::lane_chan_bytes(lane));

  u32x4 rx[NX], rd[ND];
  unsigned xmask = 0, dmask = 0;  // staged chunks that are inside the image (the others are written as zeros)
  const int tpi = a.tilesX * a.tilesY;

  // ---- global -> registers (issued one tile ahead of the MFMAs).  Addresses = one wave-uniform 64-bit tile base + a
  // 32-bit per-thread offset fixed for the whole launch + a wave-uniform row step per iteration.
  // (two input tensors: the thread's channel chunk picks its tensor once; pixel stride = that tensor's channel count)
  const bool xtwo = IM != 2 && a.x2 != nullptr;
  const int xc0 = ci0 + xch * EPC;                              // the thread's first channel of the logical input
  const int xps = xtwo ? a.xsplit : a.CinS;                     // elements between two pixels of the tensor it reads
  const int xc1 = xtwo && xc0 >= a.xsplit ? xc0 - a.xsplit : xc0;
  const T* const xsrc = xtwo && xc0 >= a.xsplit ? (const T*)a.x2 : (const T*)a.x;
  const int xvoff = (xhy0 * a.W + xhx) * xps + (IM == 2 ? 0 : xc1);
  const int dvoff = (dry0 * a.W + dcol) * a.CoutS + co0 + dch * EPC;
  auto load_tile = [&](int tile) {
    const int n = tile / tpi;
    const int trem = tile - n * tpi;
    const int ty = trem / a.tilesX, tx = trem - ty * a.tilesX;
    const int y0 = ty * TH, x0 = tx * WG_TW;
    const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + WG_TW < a.W;
    xmask = 0;
    const int gx = x0 - 1 + xhx;
    const bool colok = interior || (gx >= 0 && gx < a.W);
    const long xorigin = (((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * xps;  // halo origin, may be outside
    // Loads are UNCONDITIONAL (out-of-image lanes read a valid dummy address and are zeroed when the tile is written
    // to LDS): a load inside a divergent branch made the compiler wait for it at the join -- ten serialised L2 round
    // trips (~2 200 cycles) per tile, measured with in-kernel stamps.
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hy = xhy0 + i * XRPI;
      const int gy = y0 - 1 + hy;
      const bool ok = xact && hy < NHROWS && colok && (interior || (gy >= 0 && gy < a.H));
      xmask |= (ok ? 1u : 0u) << i;
      const long rowoff = xorigin + (long)(i * XRPI) * a.W * xps;  // wave-uniform
      if (IM == 2) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) {
          const float* src = (const float*)a.x + rowoff + xvoff;
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
        }
        rx[i] = v;
      } else {
        const T* src = ok ? xsrc + rowoff + xvoff : xsrc + xc1;
        if (a.x_up2 && ok)  // fine halo pixel (gy, gx) = pixel (gy >> 1, gx >> 1) of the half-resolution tensor
          src = xsrc + (((long)n * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)) * a.CinS + xc1;
        rx[i] = *(const u32x4*)src;
      }
    }
    const int dgx = x0 + dcol;
    const long dorigin = (((long)n * a.H + y0) * a.W + x0) * a.CoutS;
    dmask = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int r = dry0 + i * DRPI;
      const int gy = y0 + r;
      const bool ok = r < TH && gy < a.H && dgx < a.W;
      dmask |= (ok ? 1u : 0u) << i;
      const T* src = ok ? (const T*)a.dy + dorigin + (long)(i * DRPI) * a.W * a.CoutS + dvoff
                        : (const T*)a.dy + co0 + dch * EPC;
      rd[i] = *(const u32x4*)src;
    }
  };
  // ---- registers -> LDS buffer (fused BN-apply + ReLU of the producer layer on the in-image chunks)
  auto store_tile = [&](int buf) {
    unsigned char* bx = lds + buf * BUF_BYTES;
    unsigned char* bd = bx + X_BYTES;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hy = xhy0 + i * XRPI;
      if (xact && hy < NHROWS) {
        u32x4 v = rx[i];
        if (!(xmask & (1u << i))) v = (u32x4){0u, 0u, 0u, 0u};  // zero padding stays zero (also after BN + ReLU)
        else if (IM == 1 && bn_this) v = wg_bnrelu_chunk<T>(v, sc, sh);
        if constexpr (NPL == 3) store_split(bx + hy * XRP + xhx * XS + xch * LCH, XPLB, v);
        else *(u32x4*)(bx + hy * XRP + xhx * XS + xch * LCH) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int r = dry0 + i * DRPI;
      if (r < TH) {
        const u32x4 v = (dmask & (1u << i)) ? rd[i] : (u32x4){0u, 0u, 0u, 0u};
        if constexpr (NPL == 3) store_split(bd + r * DRP + dcol * DS + dch * LCH, DPLB, v);
        else *(u32x4*)(bd + r * DRP + dcol * DS + dch * LCH) = v;
      }
    }
  };

  auto compute_tile = [&](int buf) {
    const unsigned xa0 = lds_base + buf * BUF_BYTES + xlane, da0 = lds_base + buf * BUF_BYTES + X_BYTES + dlane;
    // operand fragments are fetched one (k-step, unit) ahead of the MFMAs that consume them, so an LDS read's
    // latency hides under the previous unit's matrix work instead of stalling every pair of MFMAs
    constexpr bool K32 = Frag<T>::KPIX == 32;  // k-step = two rows of 16 pixels (else 4 pixels of one row)
    constexpr bool SKIP_DUP = NUNITS % NWV != 0;
    const bool last_real = wave + NWV * (UPW - 1) < NUNITS;  // wave-uniform: the wave's last unit exists
    auto kpos_x = [&](int ks) { return (K32 ? 2 * ks : ks / 4) * XRP + (K32 ? 0 : 4 * (ks % 4)) * XS; };
    auto kpos_d = [&](int ks) { return (K32 ? 2 * ks : ks / 4) * DRP + (K32 ? 0 : 4 * (ks % 4)) * DS; };
    typename Frag<T>::type bf[NJ], bf_next[NJ], af, af_next;
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[j] = Frag<T>::load(da0, kpos_d(0) + j * 16 * LSZ, DS, DPLB);
    af = Frag<T>::load(xa0 + uoff[0], kpos_x(0), XS, XPLB);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int uu = 0; uu < UPW; ++uu) {
        if (uu + 1 < UPW) af_next = Frag<T>::load(xa0 + uoff[uu + 1], kpos_x(ks), XS, XPLB);
        else if (ks + 1 < KSTEPS) {
          af_next = Frag<T>::load(xa0 + uoff[0], kpos_x(ks + 1), XS, XPLB);
#pragma unroll
          for (int j = 0; j < NJ; ++j) bf_next[j] = Frag<T>::load(da0, kpos_d(ks + 1) + j * 16 * LSZ, DS, DPLB);
        }
        // (the clamped duplicate unit of the waves that own one unit less -- 8 waves on 18 units computed 24 -- is worth a
        // scalar branch: bf16 step, same box, the three narrow weight gradients 19.3 / 29.9 / 20.3 -> 18.3 / 27.5 / 19.8 us)
        if (!(SKIP_DUP && uu == UPW - 1 && !last_real)) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[uu][j] = Frag<T>::mfma(af, bf[j], acc[uu][j]);
        }
        af = af_next;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = bf_next[j];
    }
  };

  // debug stamps (SPCL_WGRAD_STAMPS=1): s_memtime ticks of wave 0 per loop phase, summed over the workgroup's tiles
  unsigned long long t_store = 0, t_bar = 0, t_issue = 0, t_comp = 0, t_first = 0, t_all = 0, ntl = 0;
  const bool stamp = SPCL_WGRAD_STAMPS_BUILD && a.stamps != nullptr;
  const unsigned long long c_begin = stamp ? __builtin_amdgcn_s_memtime() : 0;
  // Round k of the grid covers tiles k G .. k G + G - 1.  Workgroups go to the 8 XCDs round-robin, so with tile = k G + b
  // neighbouring tiles (which share halo columns / rows of x) sit on different XCDs.  Optional (xcd_remap, OFF: it
  // measured slower here, unlike in the forward conv kernels): XCD x takes the x-th eighth of the round's tiles,
  // tile = k G + (b % 8) (G / 8) + b / 8 (a bijection within the round).
  const int G = (int)gridDim.x;
  const int pos = (a.xcd_remap && (G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  int tile = pos;
  int buf = 0;
  if (tile < a.ntiles) load_tile(tile);
  if (stamp) t_first = __builtin_amdgcn_s_memtime() - c_begin;
  while (tile < a.ntiles) {
    const unsigned long long c0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    store_tile(buf);
    const unsigned long long c1 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();  // also orders this buffer's previous readers (two iterations back) before the next overwrite
    const unsigned long long c2 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    const int next = tile + gridDim.x;
    if (next < a.ntiles) load_tile(next);
    const unsigned long long c3 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    compute_tile(buf);
    if (stamp) {
      t_store += c1 - c0; t_bar += c2 - c1; t_issue += c3 - c2; t_comp += __builtin_amdgcn_s_memtime() - c3;
      ++ntl;
    }
    if (a.dbuf) buf ^= 1;
    else __syncthreads();  // single buffer: all reads of this tile are done before the next one is written
    tile = next;
  }
  if (stamp) t_all = __builtin_amdgcn_s_memtime() - c_begin;

  // ---- every wave writes its own units of the workgroup partial.  D layout: lane holds n = co (lane&15),
  // m = ci 4g+r.  slab layout [9][CIB][COB] f32
  const int r16 = lane & 15, g = lane >> 4;
  float* out = a.partial + ((size_t)blockIdx.x * gridDim.y + blk) * (9 * CIB * COB);
#pragma unroll
  for (int uu = 0; uu < UPW; ++uu) {
    const int u = wave + NWV * uu;
    if (u < NUNITS) {
      const int tap = u / MI, m = u - tap * MI;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[(tap * CIB + m * 16 + 4 * g + r) * COB + j * 16 + r16] = acc[uu][j][r];
    }
  }
  if (stamp && threadIdx.x == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * gridDim.y + blk) * 8;
    o[0] = t_first; o[1] = t_store; o[2] = t_bar; o[3] = t_issue; o[4] = t_comp; o[5] = t_all;
    o[6] = __builtin_amdgcn_s_memtime() - c_begin; o[7] = ntl;
  }
}

// dW_oihw[co][ci][tap] = sum over pixel-split partials.  Threads follow the PARTIAL layout ([tap][ci][co], co fastest):
// a lane owns 4 consecutive outputs (one 16-byte load per split: 1 KiB per wave-load); the waves of a block take the
// splits p = w, w + nwaves, ... (8 independent loads in flight each) and are combined through LDS in fixed order ->
// deterministic.  Four scattered 4-byte stores per lane into the OIHW gradient.
template <int VEC>  // outputs per lane: 4 (16-byte loads) for the large slabs, 1 where a slab has too few outputs
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int nblk_ci,
                                                            int nblk_co, int CIB, int COB, int Cin, int Cout,
                                                            float* __restrict__ dw) {
  __shared__ float red[16][64 * VEC];
  const int nwaves = blockDim.x >> 6;  // 4 or 16 split lanes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slab = 9 * CIB * COB;      // a multiple of 4 (COB is a multiple of 16)
  const int inner = (blockIdx.x * 64 + lane) * VEC;
  const int blk = blockIdx.y;
  const size_t nblk = (size_t)nblk_ci * nblk_co;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  if (inner < slab) {
    const float* src = partial + (size_t)blk * slab + inner;
    const size_t stride = nblk * slab;
#pragma unroll 8
    for (int p = wave; p < nsplit; p += nwaves) {
      if (VEC == 4) {
        const f32x4 v = *(const f32x4*)(src + (size_t)p * stride);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += v[e];
      } else {
        s[0] += src[(size_t)p * stride];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[wave][lane * VEC + e] = s[e];
  __syncthreads();
  if (wave == 0 && inner < slab) {
    const int bci = blk / nblk_co, bco = blk - bci * nblk_co;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = red[0][lane * VEC + e];
      for (int w = 1; w < nwaves; ++w) v += red[w][lane * VEC + e];  // fixed order
      const int i = inner + e;
      const int co_l = i % COB, ci_l = (i / COB) % CIB, tap = i / (COB * CIB);
      const int ci = bci * CIB + ci_l, co = bco * COB + co_l;
      if (ci < Cin && co < Cout) dw[((size_t)co * Cin + ci) * 9 + tap] = v;
    }
  }
}

// LDS bytes of ONE tile image (x halo + dy), row pitches padded as in the kernel.  esize: 4 f32, 2 bf16, 6 f32 held as three
// bf16 images (Frag<split_f32>)
static size_t wgrad_lds_bytes(int MI, int NJ, int TH, int esize) {
  const int planes = esize == 6 ? 3 : 1;
  if (esize == 6) esize = 2;
  const int XS = 16 * MI * esize, DS = 16 * NJ * esize;
  const int xrp = WG_HW * XS + (esize == 2 ? Frag<bf16_t>::row_pad(XS, WG_HW * XS) : 0);
  const int drp = WG_TW * DS + (esize == 2 ? Frag<bf16_t>::row_pad(DS, WG_TW * DS) : 0);
  return planes * ((size_t)(TH + 2) * xrp + (size_t)TH * drp);
}
template <typename T> constexpr int wgrad_esize() { return Frag<T>::PLANES == 3 ? 6 : (int)sizeof(T); }

struct WgradPlan {
  int TH, MI, NJ, nblk_ci, nblk_co, nsplit, ntiles, tilesX, tilesY;
  int dbuf;  // the tile image double buffered (where two fit the CU's LDS)
  size_t partial_floats;
};
static WgradPlan wgrad_plan(int N, int H, int W, int CinK, int CoutS, int esize) {
  WgradPlan p;
  p.NJ = CoutS % 32 == 0 ? 2 : 1;  // (48 padded channels: three 16-blocks, not one and a half 32-blocks)
  p.MI = CinK % 32 == 0 ? 2 : 1;
  // (bf16 layers whose channel counts are both multiples of 64 never get here: wgrad_gemm.hip)
  p.nblk_ci = CinK / (16 * p.MI);
  p.nblk_co = CoutS / (16 * p.NJ);
  // 14-row tiles when the height divides by 14 but not by 16 (56 / 28 / 14: 12.5 % padded pixels instead of 23 %)
  static const int env_th = lab_env("SPCL_WGRAD_TH", 0);
  p.TH = env_th ? env_th : ((H % 16 != 0 && H % 14 == 0) ? 14 : 16);
  // three-image tiles (esize 6) of the 16-channel-sided blocks: 8 rows, single buffered -- three or four workgroups per CU
  // instead of one (same box: 16 -> 16 at 224^2 186 -> 118 us, 16 -> 32 at 112^2 75 -> 53; the 32 x 32 blocks measured
  // no better on 8 rows, buffered either way)
  const bool narrow3 = esize == 6 && p.MI * p.NJ < 4 && H >= 8 && !env_th;
  if (narrow3) p.TH = 8;
  p.tilesX = cdiv(W, WG_TW);
  p.tilesY = cdiv(H, p.TH);
  p.ntiles = N * p.tilesX * p.tilesY;
  const int nblk = p.nblk_ci * p.nblk_co;
  // exactly one resident "wave" of workgroups (LDS-limited residency x 256 CUs): measured optimum -- more workgroups
  // only add partial slabs and a tail, fewer leave CUs idle (tools/bench_kernels.py wgrad sweeps)
  static const int plan_dbuf = lab_env("SPCL_WGRAD_DBUF", 1);
  p.dbuf = plan_dbuf && !narrow3 && 2 * wgrad_lds_bytes(p.MI, p.NJ, p.TH, esize) <= 160 * 1024 ? 1 : 0;
  const size_t lds = (p.dbuf ? 2 : 1) * wgrad_lds_bytes(p.MI, p.NJ, p.TH, esize);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  static const int env_wgs = lab_env("SPCL_WGRAD_WGS", 0);
  int ns = cdiv(env_wgs > 0 ? env_wgs : 256 * per_cu, nblk);
  if (ns > p.ntiles) ns = p.ntiles;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  p.partial_floats = (size_t)ns * nblk * 9 * (16 * p.MI) * (16 * p.NJ);
  return p;
}

template <typename T, int MI, int NJ, int TH, int NWV, int IM>
static void launch_wgrad_im(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  const size_t lds = (a.dbuf ? 2 : 1) * wgrad_lds_bytes(MI, NJ, TH, wgrad_esize<T>());
  if (lds > 65536)
    spcl::func_lds_limit((const void*)conv3x3_wgrad_kernel<T, MI, NJ, TH, NWV, IM>, (int)lds,
                         "conv3x3_wgrad_kernel<T, MI, NJ, TH, NWV, IM>");
  SPCL_LAUNCH((conv3x3_wgrad_kernel<T, MI, NJ, TH, NWV, IM>), dim3(p.nsplit, p.nblk_ci * p.nblk_co), dim3(64 * NWV), lds, st,
              a);
}

template <typename T, int MI, int NJ, int TH, int NWV>
static void launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  if (a.in_mode == 1) launch_wgrad_im<T, MI, NJ, TH, NWV, 1>(a, p, st);
  else if (a.in_mode == 0) launch_wgrad_im<T, MI, NJ, TH, NWV, 0>(a, p, st);
  else if constexpr (MI == 1 && NWV == 4) launch_wgrad_im<T, MI, NJ, TH, NWV, 2>(a, p, st);  // (image mode: CinK == 16)
}

template <typename T, int TH>
static void launch_wgrad_th(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  // 8 waves on the 32 x 32 block (18 (tap, ci-tile) units: 3 or 2 per wave instead of 5 or 4, the staging spread over twice
  // the threads): Conv2.b + Conv3.a -8 us per step, same box.  Bit 1 of SPCL_WGRAD_W8: the same for the 16 x 32 block.
  static const int env_w8 = lab_env("SPCL_WGRAD_W8", 1);
  if (p.MI == 1 && p.NJ == 1) launch_wgrad<T, 1, 1, TH, 4>(a, p, st);
  else if (p.MI == 1 && p.NJ == 2) {
    if ((env_w8 & 2) && a.in_mode != 2) launch_wgrad<T, 1, 2, TH, 8>(a, p, st);  // (the image mode exists for 4 waves only)
    else launch_wgrad<T, 1, 2, TH, 4>(a, p, st);
  } else if (p.MI == 2 && p.NJ == 1) launch_wgrad<T, 2, 1, TH, 4>(a, p, st);
  else if (env_w8 & 1) launch_wgrad<T, 2, 2, TH, 8>(a, p, st);
  else launch_wgrad<T, 2, 2, TH, 4>(a, p, st);
}

template <typename T>
static void launch_wgrad_t(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  if constexpr (Frag<T>::PLANES == 3) {  // (the three-image tiles: 8 rows, so that two workgroups or two buffers fit a CU)
    if (p.TH == 8) return launch_wgrad_th<T, 8>(a, p, st);
  }
  if (p.TH == 14) launch_wgrad_th<T, 14>(a, p, st);
  else launch_wgrad_th<T, 16>(a, p, st);
}

// the final sum of [nsplit][nblk_ci * nblk_co][9][CIB][COB] partial slabs into the OIHW gradient (also conv16_bwd.hip)
void launch_wgrad_reduce(const float* partial, int nsplit, int nblk_ci, int nblk_co, int CIB, int COB, int Cin, int Cout,
                         float* dw_oihw, hipStream_t st) {
  const int slab = 9 * CIB * COB;
  if (slab >= 16384)
    SPCL_LAUNCH(wgrad_reduce_kernel<4>, dim3(cdiv(slab, 256), nblk_ci * nblk_co), dim3(nsplit >= 64 ? 1024 : 256), 0, st,
                partial, nsplit, nblk_ci, nblk_co, CIB, COB, Cin, Cout, dw_oihw);
  else
    SPCL_LAUNCH(wgrad_reduce_kernel<1>, dim3(cdiv(slab, 64), nblk_ci * nblk_co), dim3(nsplit >= 64 ? 1024 : 256), 0, st,
                partial, nsplit, nblk_ci, nblk_co, CIB, COB, Cin, Cout, dw_oihw);
}

}  // namespace spcl

using namespace spcl;

// the one-layer form of the batched kernel (wgrad_gemm.hip) for bf16 layers with 64-multiple channel counts
static bool wide_item(spcl_wgrad_item& it, const void* x, const void* dy, int N, int H, int W, int Cin, int CinS, int CinK,
                      int Cout, int CoutS, int in_mode, const float* in_scale, const float* in_shift, float* dw) {
  if (Cin != CinK || CinS != CinK || Cout != CoutS ||
      !spcl_conv_wgrad_batched_supported(SPCL_BF16, Cin, CinS, Cout, CoutS, in_mode))
    return false;
  it.x = x; it.x2 = nullptr; it.x_up2 = 0; it.dy = dy; it.in_scale = in_scale; it.in_shift = in_shift; it.dw_oihw = dw;
  it.N = N; it.H = H; it.W = W; it.Cin = Cin; it.CinS = CinS; it.Cout = Cout; it.CoutS = CoutS; it.in_mode = in_mode;
  return true;
}

extern "C" size_t spcl_conv_wgrad_workspace_bytes(int N, int H, int W, int CinK, int CoutS) {
  if (N <= 0 || H <= 0 || W <= 0 || CinK <= 0 || CoutS <= 0 || CinK % 16 || CoutS % 16) return 0;
  // sized for the largest of the three plans (exact f32, bf16, split f32) and for the wide-layer kernel
  size_t bytes = 0;
  for (int esize : {4, 2, 6}) {
    const size_t b = wgrad_plan(N, H, W, CinK, CoutS, esize).partial_floats * sizeof(float);
    if (b > bytes) bytes = b;
  }
  spcl_wgrad_item it;
  static float dummy;
  if (wide_item(it, &dummy, &dummy, N, H, W, CinK, CinK, CinK, CoutS, CoutS, 0, nullptr, nullptr, &dummy)) {
    const size_t w = spcl_conv_wgrad_batched_workspace_bytes(&it, 1);
    if (w > bytes) bytes = w;
  }
  return bytes;
}

static int conv3x3_wgrad_impl(const void* x, const void* x2, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS,
                              int CinK, int Cout, int CoutS, int in_mode, const float* in_scale, const float* in_shift,
                              float* partial, float* dw_oihw, void* stream, bool x_up2 = false);

extern "C" int spcl_conv3x3_wgrad(const void* x, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS,
                                  int CinK, int Cout, int CoutS, int in_mode, const float* in_scale,
                                  const float* in_shift, float* partial, float* dw_oihw, void* stream) {
  return conv3x3_wgrad_impl(x, nullptr, dy, dtype, N, H, W, Cin, CinS, CinK, Cout, CoutS, in_mode, in_scale, in_shift, partial,
                            dw_oihw, stream);
}

// ... of the convolution behind the decoder's torch.cat((skip, up), dim=1) (unet.py:194-224), its input read from the two
// tensors in place: xa = channels [0, Chalf), xb = [Chalf, 2 Chalf), both dense [N][H][W][Chalf] bf16 (Chalf 16 / 32: the
// narrow kernel above; 64 / 128 with CoutS a multiple of 64: the batched GEMM kernel, spcl_wgrad_item::x2).  xb_scale /
// xb_shift [Chalf] (Chalf 16 / 32, or NULL): xb is the RAW output of a convolution and relu(xb_scale xb + xb_shift) is what
// the concatenation holds (the up-convolution's BatchNorm + ReLU applied while staging).  Workspace as spcl_conv3x3_wgrad
// with CinK = 2 Chalf.
extern "C" int spcl_conv3x3_wgrad_cat(const void* xa, const void* xb, const void* dy, int dtype, int N, int H, int W, int Chalf,
                                      int Cout, int CoutS, const float* xb_scale, const float* xb_shift, float* partial,
                                      float* dw_oihw, void* stream) {
  SPCL_CHECK_ARG(xa && xb, "conv3x3_wgrad_cat: null pointer");
  SPCL_CHECK_ARG(dtype == SPCL_BF16 && (Chalf == 16 || Chalf == 32 || Chalf == 64 || Chalf == 128),
                 "conv3x3_wgrad_cat: bf16, Chalf 16 / 32 / 64 / 128");
  SPCL_CHECK_ARG(!((2 * Chalf) % 64 == 0 && CoutS % 64 == 0) || Chalf % 64 == 0,
                 "conv3x3_wgrad_cat: a batched-GEMM layer needs halves of whole 64-channel blocks");
  SPCL_CHECK_ARG((uintptr_t)xa % 16 == 0 && (uintptr_t)xb % 16 == 0, "conv3x3_wgrad_cat: inputs must be 16-byte aligned");
  SPCL_CHECK_ARG((xb_scale == nullptr) == (xb_shift == nullptr), "conv3x3_wgrad_cat: xb_scale and xb_shift come together");
  SPCL_CHECK_ARG(xb_scale == nullptr || Chalf <= 32, "conv3x3_wgrad_cat: the raw second tensor exists for Chalf 16 / 32 only");
  return conv3x3_wgrad_impl(xa, xb, dy, dtype, N, H, W, 2 * Chalf, 2 * Chalf, 2 * Chalf, Cout, CoutS, xb_scale ? 1 : 0,
                            xb_scale, xb_shift, partial, dw_oihw, stream);
}

// ... of the up-convolution (unet.py:89-90): its input is nn.Upsample(scale_factor=2)(x_half), read from the half-resolution
// tensor [N][H / 2][W / 2][Cin] (H x W = the convolution's size; bf16, Cin == CinS == CinK, no input transform).
extern "C" int spcl_conv3x3_wgrad_up2(const void* x_half, const void* dy, int dtype, int N, int H, int W, int Cin, int Cout,
                                      int CoutS, float* partial, float* dw_oihw, void* stream) {
  SPCL_CHECK_ARG(dtype == SPCL_BF16 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0,
                 "conv3x3_wgrad_up2: bf16, even sizes, Cin a multiple of 16");
  SPCL_CHECK_ARG((double)N * H * W / 4.0 * Cin * 2.0 < 2147483648.0, "conv3x3_wgrad_up2: tensor too large for 32-bit offsets");
  return conv3x3_wgrad_impl(x_half, nullptr, dy, dtype, N, H, W, Cin, Cin, Cin, Cout, CoutS, 0, nullptr, nullptr, partial,
                            dw_oihw, stream, true);
}

static int conv3x3_wgrad_impl(const void* x, const void* x2, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS,
                              int CinK, int Cout, int CoutS, int in_mode, const float* in_scale, const float* in_shift,
                              float* partial, float* dw_oihw, void* stream, bool x_up2) {
  SPCL_CHECK_ARG(x && dy && partial && dw_oihw, "conv3x3_wgrad: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: bad shape");
  SPCL_CHECK_ARG(CinK % 16 == 0 && CoutS % 16 == 0 && Cin <= CinK && Cout <= CoutS, "conv3x3_wgrad: channel padding");
  SPCL_CHECK_ARG(in_mode >= 0 && in_mode <= 2, "conv3x3_wgrad: in_mode %d", in_mode);
  SPCL_CHECK_ARG(in_mode != 1 || (in_scale && in_shift), "conv3x3_wgrad: in_mode 1 needs scale/shift");
  if (in_mode == 2) SPCL_CHECK_ARG(CinK == 16 && CinS >= 1 && CinS <= 16, "conv3x3_wgrad: image mode needs Cin<=16");
  else SPCL_CHECK_ARG(CinS == CinK, "conv3x3_wgrad: CinS must equal CinK");
  hipStream_t st = (hipStream_t)stream;
  spcl_wgrad_tail* tail = take_tail_capture();  // non-null: leave the final sum to spcl_conv3x3_wgrad_batched_tails
  if (dtype == SPCL_BF16) {
    spcl_wgrad_item it;
    if (wide_item(it, x, dy, N, H, W, Cin, CinS, CinK, Cout, CoutS, in_mode, in_scale, in_shift, dw_oihw)) {
      it.x2 = x2;  // (the batched kernel reads a 64-channel block from the tensor that holds it)
      it.x_up2 = x_up2 ? 1 : 0;
      return spcl_conv3x3_wgrad_batched(&it, 1, 0, partial, stream);
    }
  }
  const bool split = dtype == SPCL_F32 && conv_f32_split();
  WgradPlan p = wgrad_plan(N, H, W, CinK, CoutS, dtype == SPCL_F32 ? (split ? 6 : 4) : 2);
  WgradArgs a;
  a.x = x; a.x2 = x2; a.xsplit = x2 ? CinK / 2 : 0; a.x_up2 = x_up2 ? 1 : 0; a.dy = dy; a.in_scale = in_scale; a.in_shift = in_shift; a.partial = partial;
  a.N = N; a.H = H; a.W = W; a.CinS = CinS; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = in_mode;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.ntiles = p.ntiles; a.nblk_ci = p.nblk_ci; a.nblk_co = p.nblk_co;
  static const int env_dbuf = lab_env("SPCL_WGRAD_DBUF", 1);
  a.dbuf = env_dbuf && p.dbuf;
  static const int env_remap = lab_env("SPCL_WGRAD_XCD_REMAP", 0);  // measured: +3..12 us per step, off
  a.xcd_remap = env_remap;
  static const int env_stamps = SPCL_WGRAD_STAMPS_BUILD ? lab_env("SPCL_WGRAD_STAMPS", 0) : 0;
  static unsigned long long* stamp_buf = nullptr;
  a.stamps = nullptr;
  const size_t nwg = (size_t)p.nsplit * p.nblk_ci * p.nblk_co;
  if (env_stamps) {  // debug only: per-phase cycle counters of wave 0 of every workgroup (synchronises!)
    if (!stamp_buf) (void)hipMalloc(&stamp_buf, 8192 * 8 * sizeof(unsigned long long));
    if (nwg <= 8192) a.stamps = stamp_buf;
  }
  {
    const double px = (double)N * H * W, es = dtype == SPCL_F32 ? 4.0 : 2.0;
    prof_cost(px * ((in_mode == 2 ? CinS * 4.0 : CinK * es) + CoutS * es) + 9.0 * Cin * Cout * 4.0,
              2.0 * px * 9.0 * Cin * Cout);
  }
  if (dtype == SPCL_F32 && split) launch_wgrad_t<split_f32>(a, p, st);
  else if (dtype == SPCL_F32) launch_wgrad_t<float>(a, p, st);
  else if (dtype == SPCL_BF16) launch_wgrad_t<bf16_t>(a, p, st);
  else {
    set_error("conv3x3_wgrad: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  if (tail != nullptr) {
    tail->partial = partial; tail->dw = dw_oihw; tail->kind = 0; tail->nsplit = p.nsplit; tail->nblk_ci = p.nblk_ci;
    tail->nblk_co = p.nblk_co; tail->CIB = 16 * p.MI; tail->COB = 16 * p.NJ; tail->Cin = Cin; tail->Cout = Cout;
  } else {
    launch_wgrad_reduce(partial, p.nsplit, p.nblk_ci, p.nblk_co, 16 * p.MI, 16 * p.NJ, Cin, Cout, dw_oihw, st);
  }
  if (a.stamps) {
    std::vector<unsigned long long> h(nwg * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double s[8] = {0};
    for (size_t i = 0; i < nwg; ++i)
      for (int k = 0; k < 8; ++k) s[k] += (double)h[i * 8 + k];
    for (int k = 0; k < 8; ++k) s[k] /= (double)nwg;
    fprintf(stderr, "[wgrad stamps] %dx%d Cin%d Cout%d MI%d NJ%d TH%d wgs=%zu tiles/wg=%.1f | memtime ticks per wg: first-load "
            "issue %.0f, store %.0f, barrier %.0f, next-load issue %.0f, compute %.0f, loop total %.0f, with epilogue %.0f\n",
            H, W, Cin, Cout, p.MI, p.NJ, p.TH, nwg, s[7], s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
  }
  SPCL_LAUNCH_CHECK("conv3x3_wgrad");
  return SPCL_OK;
}
