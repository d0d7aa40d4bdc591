// Pieces shared by the generic (conv.hip) and the compile-time-specialised (conv_fast.hip) 3x3 convolution kernels.
#pragma once
#include "common.hpp"

namespace spcl {

struct BnAccFwd;
template <typename T> struct Chunk;
template <> struct Chunk<float> { static constexpr int EPC = 4; };
template <> struct Chunk<bf16_t> { static constexpr int EPC = 8; };

struct uint4_ { uint32_t x, y, z, w; };
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__host__ __device__ inline int conv_kc(int CinK) { return CinK < 64 ? CinK : 64; }
template <typename T> __host__ __device__ inline int conv_pstride(int KC) {
  int b = KC * (int)sizeof(T);
  return b == 32 ? 32 : b + 32;  // bytes per halo pixel in LDS (see tools/ bank analysis in DESIGN.md)
}
template <typename T> __host__ __device__ inline int conv_nsteps(int KC) {
  int cp = KC / Chunk<T>::EPC;
  return (9 * cp + 3) / 4;
}

struct ConvArgs {
  const void* x;
  void* y;
  const void* wp;
  float* stats;
  const float* in_scale;
  const float* in_shift;
  int N, H, W;
  int CinS;     // storage stride of x in elements (mode 2: real channel count of the f32 image)
  int CinK;     // GEMM-K channels, multiple of 16
  int CoutS;    // storage stride of y == padded output channels (multiple of 16)
  int in_mode;  // 0 raw, 1 relu(scale*x+shift), 2 f32 image with CinS (<16) channels zero-padded to 16
  int tilesX, tilesY;
  int tpw;  // tiles per workgroup (processed sequentially)
  int dbg;  // ablation bits (experiments only, 0 in production): 1 no BN transform, 2 no k-loop, 4 no stats,
            // 8 no output stores, 16 no staging loads
  // dgrad + BatchNorm-backward partial sums of the layer whose activation gradient this conv produces (fast path only)
  const void* y2 = nullptr;
  const float* scale2 = nullptr;
  const float* shift2 = nullptr;
  const float* mean2 = nullptr;
  float* rows2 = nullptr;
  // ... or, with H2 > 0, of the POOLED layer: y2 is [N][H2][W2][CoutS] at twice the resolution (H == H2 / 2), the dgrad's
  // output is the gradient of maxpool2x2(relu(bn(y2)))
  int H2 = 0, W2 = 0;
  // ... and, for the block whose first conv reads a ONE-CHANNEL f32 image (unet.py:123), the nine sums
  // sum_p dz[p][co] img[p + tap] of that conv's weight gradient as rows 2 .. 10 of rows2 ([tile][11][CoutS], MODE 4)
  const float* img2 = nullptr;
  // the input as the channel concatenation of TWO dense tensors (x: channels [0, CinK / 2), x2: the rest; each
  // [N][H][W][CinK / 2]) -- torch.cat((skip, up), 1) of the decoder read in place (fast path only, one slab: CinK <= 64)
  const void* x2 = nullptr;
  // the output as TWO dense tensors (y: channels [0, CoutS / 2), y_hi: the rest; each [N][H][W][CoutS / 2]) -- the gradient
  // of such a concatenation written as the gradients of its parts (fast path only, plain dgrad: no statistics)
  void* y_hi = nullptr;
  // x is [N][H / 2][W / 2][CinK] and the input of the convolution is its nearest-neighbour x2 upsample (unet.py:89
  // nn.Upsample(scale_factor=2) in front of the up-convolution), never materialised (fast path only, in_mode 0)
  bool x_up2 = false;
  // in_mode 2 (the one-channel image convolution) with the image's autocorrelation rows as a by-product: one row [64] per
  // tile ([N * tiles][64]: 45 + 9 sums over the tile's pixels, the layout of image_autocorr_kernel's band rows) -- fast
  // path only, image sizes that are multiples of the 14 x 14 tile
  float* acorr_rows = nullptr;
  // BatchNorm sums through fixed-point accumulator blocks (bn_acc.hpp; fast path only -- a launch that carries one of these
  // and finds no specialised kernel FAILS, it never falls back to a kernel that would ignore them):
  long long* stats_acc = nullptr;        // the output's statistics are added here instead of written as per-tile `stats` rows
  long long* rows2_acc = nullptr;        // the dgrad's BatchNorm-backward sums are added here instead of written as `rows2`
  const struct BnAccFwd* in_bn = nullptr;  // in_mode 1: scale / shift derived from this block in the prologue (HOST pointer)
};

// f32 storage computed on the bf16 matrix rate (conv.hip "SPLIT", wgrad.hip Frag<split_f32>): the element type of tensors
// whose values are multiplied as three bf16 pieces
struct split_f32 { float v; };
template <> struct Chunk<split_f32> { static constexpr int EPC = 4; };
bool conv_f32_split();  // conv.hip: spcl_conv_set_f32_split

typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_;
struct Split3 { uint32_t hi, mid, lo; };  // packed bf16 pairs
__device__ __forceinline__ Split3 split3_pair(float v0, float v1) {
  Split3 o;
  f32x2_ v = {v0, v1};
  o.hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v_));
  v[0] -= __uint_as_float(o.hi << 16);
  v[1] -= __uint_as_float(o.hi & 0xffff0000u);
  o.mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v_));
  v[0] -= __uint_as_float(o.mid << 16);
  v[1] -= __uint_as_float(o.mid & 0xffff0000u);
  o.lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v_));
  return o;
}
// ... of the eight f32 values of two 16-byte chunks: one 16-byte chunk per plane
__device__ __forceinline__ void split3_chunk(const float* e, u32x4& ph, u32x4& pm, u32x4& pl) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Split3 o = split3_pair(e[2 * k], e[2 * k + 1]);
    ph[k] = o.hi;
    pm[k] = o.mid;
    pl[k] = o.lo;
  }
}
template <typename T> __device__ __forceinline__ f32x4 mfma_chunk(u32x4 w, u32x4 x, f32x4 acc);
template <> __device__ __forceinline__ f32x4 mfma_chunk<bf16_t>(u32x4 w, u32x4 x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0,
                                                 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_chunk<float>(u32x4 w, u32x4 x, f32x4 acc) {
  f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[3], acc, 0, 0, 0);
  return acc;
}

// sum over the 16 lanes that share lane>>4 (one pixel column group), DPP only (no LDS traffic): xor 1, xor 2,
// mirror within 8, mirror within 16
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(2))) short s16x2v;

// relu of a packed bf16 pair: the maximum with 0 of its halves read as signed 16-bit integers (v_pk_max_i16)
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t packed) {
  const s16x2v z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, packed), z));
}

// relu(scale*v+shift) on one 16-byte chunk with the coefficients already in registers (packed-f32 FMA and one
// v_cvt_pk_bf16_f32 per channel pair on the bf16 path)
template <typename T> __device__ __forceinline__ u32x4 bnrelu_regs(u32x4 raw, const float* s, const float* b);
template <> __device__ __forceinline__ u32x4 bnrelu_regs<float>(u32x4 raw, const float* s, const float* b) {
  f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(s[e], v[e], b[e]), 0.f);
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 bnrelu_regs<bf16_t>(u32x4 raw, const float* s, const float* b) {
  u32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float lo = __uint_as_float(raw[e] << 16), hi = __uint_as_float(raw[e] & 0xffff0000u);
    lo = fmaf(s[2 * e], lo, b[2 * e]);  // same arithmetic as wgrad's staging: identical activations
    hi = fmaf(s[2 * e + 1], hi, b[2 * e + 1]);
    const f32x2 v = {lo, hi};
    // ReLU on the ROUNDED pair, as a signed 16-bit maximum with 0 (one v_pk_max_i16 for two v_max_f32): rounding to bf16 keeps
    // the sign, so relu-then-round and round-then-relu are the same bits (-0 and every negative value have the sign bit set
    // and become +0; a NaN stays a NaN instead of becoming 0)
    out[e] = relu_bf16x2(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v)));
  }
  return out;
}

template <typename T> __device__ __forceinline__ void store4_fast(unsigned char* p, f32x4 v);
template <> __device__ __forceinline__ void store4_fast<float>(unsigned char* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4_fast<bf16_t>(unsigned char* p, f32x4 v) {
  const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
  uint2 o;
  o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
  o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
  *(uint2*)p = o;
}

// BatchNorm partials of one tile, row [tile][3][CoutS] = (count, mean, M2) of each output channel.  `ssum`/`ssq` are
// the per-lane sums of the lane's 4 channels cb..cb+3; after row16_sum every lane of a 16-lane row holds the row's
// totals, and lanes r16 = 0, 1, 2 each write one component with one 16-byte store (one 64-byte line per row and
// component instead of 12 scattered dwords).
__device__ __forceinline__ void write_tile_stats(float* stats, int tile, int CoutS, int cb, int r16, float cnt,
                                                 f32x4 ssum, f32x4 ssq) {
  const float inv = 1.f / cnt;
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s1 = row16_sum(ssum[r]), s2 = row16_sum(ssq[r]);
    const float mean = s1 * inv;
    // M2 = sum x^2 - n mean^2, the product rounded on its own (not contracted into an FMA): a tile whose values are all
    // equal then gives exactly 0, as the rounded sum of squares minus the equally rounded product
    float prod = s1 * mean;
    asm volatile("" : "+v"(prod));  // (an opaque copy: no contraction across it)
    const float m2 = fmaxf(s2 - prod, 0.f);
    o[r] = r16 == 0 ? cnt : (r16 == 1 ? mean : m2);
  }
  if (r16 < 3) *(f32x4*)(stats + ((size_t)tile * 3 + r16) * CoutS + cb) = o;
}

// conv_fast.hip: returns true when a specialised kernel exists for this configuration and was launched
bool launch_conv_fast(const ConvArgs& a, int th, hipStream_t st, bool dry = false);
// (tools/experiments/conv_stream.hip: the persistent DMA-pipelined variant for the layers with <= 32 input channels of round
// 4 -- isolated launches 20-30 % faster, no gain inside the step where every launch carries a fused BatchNorm mode -- is a lab
// record now, not part of the library)

// conv_gemm.hip: workgroup-level GEMM kernel for bf16 layers whose channel counts are multiples of 64 with at least one
// side >= 128 (64 -> 64, Conv3.b: 896 short single-slab workgroups, stays with the per-wave kernel).  Their packed weight
// buffers carry both layouts (conv.hip packed_elems / pack_value); conv_use_gemm (conv.hip) picks the kernel per launch.
__host__ __device__ inline bool conv_gemm_channels(int CinK, int CoutS) {
  return CinK % 64 == 0 && CoutS % 64 == 0 && CinK >= 64 && CoutS >= 64 && (CinK >= 128 || CoutS >= 128);
}
bool conv_gemm_fits(int H, int W, int CinK, int CoutS);
bool conv_use_gemm(int CinK, int CoutS, int H, int W);
void conv_set_gemm(int mode);
bool launch_conv_gemm(const ConvArgs& a, hipStream_t st);
int conv_gemm_stat_rows(int N, int H, int W, int CinK, int CoutS);

}  // namespace spcl
