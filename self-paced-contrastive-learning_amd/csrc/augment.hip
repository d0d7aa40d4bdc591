// Per-sample random flips of a batch (semi_seg/epochers/new_epocher.py:112 TensorRandomFlip(axis=[1,2]) applied
// sample by sample in new_pretrain.py:57-58,64-65) as ONE streaming launch: out[n] = flip(x[n], dims(flags[n])).
// The reference loops over the samples on the host (one flip kernel each); the batched torch formulation is still
// ~10 launches (index_select / flip / index_copy per pattern).  16-byte vectors, reversed in registers for W flips.
#include <string.h>
#include "common.hpp"

namespace spcl {

// STAGED: the launch also carries the step's host-written inputs (stepgraph.StepStage: label vectors, these flip flags) as
// kernel arguments -- workgroup 0 writes them to their persistent device block (what spcl_stage_bytes does in a launch
// of its own), and every workgroup reads ITS flags from the arguments (the device block is being written meanwhile).
constexpr int FLIP_STAGE_WORDS = 896;  // 3 584 bytes
struct FlipStage {
  uint32_t w[FLIP_STAGE_WORDS];
};
template <typename T, int VEC, bool STAGED>
__device__ __forceinline__ void flip_batch_body(const T* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W,
                                                const uint8_t* __restrict__ flags, const T* __restrict__ head, int NH) {
  // head != null: `out` has NH + N samples, the first NH are a plain copy of `head` (the unflipped view of the pair)
  const int WV = W / VEC;
  const size_t total = (size_t)(N + NH) * C * H * WV;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int wv = (int)(idx % WV);
    size_t r = idx / WV;
    const int h = (int)(r % H);
    r /= H;  // r = n*C + c
    const int n = (int)(r / C);
    const bool copy = n < NH;
    const uint8_t f = copy ? (uint8_t)0 : flags[n - NH];
    const int sh = (f & 1) ? H - 1 - h : h;
    const int swv = (f & 2) ? WV - 1 - wv : wv;
    const size_t rs = copy ? r : r - (size_t)NH * C;  // sample-channel row inside its source
    const T* src = (copy ? head : x) + (rs * H + sh) * (size_t)W + (size_t)swv * VEC;
    T v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = src[e];  // contiguous: one vector load
    T* dst = out + (r * H + h) * (size_t)W + (size_t)wv * VEC;
    if (f & 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[e] = v[VEC - 1 - e];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[e] = v[e];
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void flip_batch_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int C,
                                                         int H, int W, const uint8_t* __restrict__ flags,
                                                         const T* __restrict__ head, int NH) {
  flip_batch_body<T, VEC, false>(x, out, N, C, H, W, flags, head, NH);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void flip_pair_stage_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int C,
                                                              int H, int W, const T* __restrict__ head, int NH,
                                                              uint32_t* __restrict__ stage_dst, int nwords, int flag_off,
                                                              FlipStage s) {
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nwords; i += 256) stage_dst[i] = s.w[i];
  flip_batch_body<T, VEC, true>(x, out, N, C, H, W, (const uint8_t*)s.w + flag_off, head, NH);
}

template <typename T>
static void launch_flip(const void* x, void* out, int N, int C, int H, int W, const uint8_t* flags, hipStream_t st,
                        const void* head = nullptr, int NH = 0) {
  constexpr int V = 16 / (int)sizeof(T);
  const bool vec = W % V == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)head % 16 == 0);
  const size_t total = (size_t)(N + NH) * C * H * (vec ? W / V : W);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec) SPCL_LAUNCH((flip_batch_kernel<T, V>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, N,
                              C, H, W, flags, (const T*)head, NH);
  else SPCL_LAUNCH((flip_batch_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, N, C,
                          H, W, flags, (const T*)head, NH);
}

// ---------------------------------------------------------------------------------------------------------------------
// The pre-train augmentation recipe on device (semi_seg/augment.py:6-22 `ACDCStrongTransforms.pretrain`, which the
// reference runs with PIL in DataLoader workers): RandomRotation -> RandomVerticalFlip -> RandomHorizontalFlip ->
// RandomCrop -> ColorJitter(brightness, contrast) -> ToTensor, one view per workgroup, gathered straight from the
// device-resident slice store.  Geometry is INTEGER arithmetic (bit-exact against the oracle): the rotation is a 16.16
// fixed-point matrix applied to half-pixel coordinates, sampling is nearest (torchvision's default), outside = 0.
//   params[v] = {slice, cos_q16, sin_q16, flags (1 hflip, 2 vflip, 4 contrast before brightness), top, left,
//                brightness (f32 bits), contrast (f32 bits)}
// Colour (float-tensor semantics): brightness u = clamp(b u); contrast u = clamp(c u + (1 - c) mean(u)), the mean taken
// over the view right before the contrast step -> two passes over the view's pixels inside the workgroup with a fixed-order
// block reduction between them.
__global__ __launch_bounds__(1024) void augment_views_kernel(const float* __restrict__ src, int S, int HS, int WS,
                                                             const int* __restrict__ params, float* __restrict__ out,
                                                             int OH, int OW) {
  __shared__ float red[16];
  const int v = blockIdx.x;
  const int* pr = params + v * 8;
  const int slice = pr[0], cq = pr[1], sq = pr[2], flags = pr[3], top = pr[4], left = pr[5];
  const float b = __int_as_float(pr[6]), c = __int_as_float(pr[7]);
  const float* img = src + (size_t)slice * HS * WS;
  const bool contrast_first = flags & 4;
  auto sample = [&](int p) -> float {
    const int i = p / OW, j = p - i * OW;
    int y = i + top, x = j + left;           // position in the rotated + flipped image (same size as the slice)
    if (flags & 1) x = WS - 1 - x;           // undo the horizontal flip
    if (flags & 2) y = HS - 1 - y;           // undo the vertical flip
    const int dx2 = 2 * x + 1 - WS, dy2 = 2 * y + 1 - HS;  // half-pixel units from the centre
    // inverse rotation (output -> source): [ c s ; -s c ] in 16.16; + centre; floor of (source + 0.5) by arithmetic shift
    const int sx = (cq * dx2 + sq * dy2 + 65536 * WS) >> 17;
    const int sy = (-sq * dx2 + cq * dy2 + 65536 * HS) >> 17;
    return (sx >= 0 && sx < WS && sy >= 0 && sy < HS) ? img[(size_t)sy * WS + sx] : 0.f;
  };
  const int np = OH * OW;
  float part = 0.f;
  for (int p = threadIdx.x; p < np; p += 1024) {
    float u = sample(p);
    if (!contrast_first) u = fminf(fmaxf(b * u, 0.f), 1.f);
    part += u;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) total += red[w];  // fixed order
  const float mean = total / (float)np;
  float* o = out + (size_t)v * np;
  for (int p = threadIdx.x; p < np; p += 1024) {
    float u = sample(p);
    if (!contrast_first) u = fminf(fmaxf(b * u, 0.f), 1.f);
    u = fminf(fmaxf(c * u + (1.f - c) * mean, 0.f), 1.f);
    if (contrast_first) u = fminf(fmaxf(b * u, 0.f), 1.f);
    o[p] = u;
  }
}

// The same recipe in PIL's OWN arithmetic (what the reference's DataLoader workers compute on 8-bit slices,
// semi_seg/augment.py:6-22 through torchvision -> PIL): bit-exact against tests/golden/g9_augment.npz, which PIL itself wrote.
//   * RandomRotation = Image.rotate(angle, NEAREST) = ImagingTransformAffine's fixed-point nearest path (Geometry.c
//     affine_fixed): xin = (a2 + x a0 + y a1) >> 16, yin = (a5 + x a3 + y a4) >> 16 with the six 16.16 coefficients rounded
//     from the double-precision matrix on the host exactly as PIL rounds them (semi_seg/data/augment.py pil_affine_q16);
//     0 outside.  Flips and the crop are index arithmetic in front of it (output -> cropped -> unflipped -> rotated).
//   * ColorJitter on a one-channel image = ImageEnhance.Brightness / .Contrast = Image.blend(degenerate, image, factor) on
//     8-bit values: float(in1) + factor * float(in2 - in1) in f32 (product and sum rounded separately), TRUNCATED to 8 bits
//     (clipped to [0, 255] when the factor extrapolates); Brightness blends with black, Contrast with the image's mean grey
//     level int(mean + 0.5) taken right before the contrast step.  Saturation and hue are the identity on one channel.
//   * ToTensor: value / 255 in f32.
//   params[v] = {slice, a0, a1, a2, a3, a4, a5, flags (1 hflip, 2 vflip, 4 contrast before brightness), top, left,
//                brightness (f32 bits), contrast (f32 bits)};  the store holds 8-bit grey levels as k / 255 (k = round(255 v)).
__device__ __forceinline__ int pil_blend(int in1, int in2, float alpha, bool interpolate) {
  const float t = __fadd_rn((float)in1, __fmul_rn(alpha, (float)(in2 - in1)));  // no contraction: PIL's C rounds both
  if (interpolate) return (int)t;  // 0 <= alpha <= 1: (UINT8) of a value already inside [0, 255]
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__global__ __launch_bounds__(1024) void augment_views_pil_kernel(const float* __restrict__ src, int S, int HS, int WS,
                                                                 const int* __restrict__ params, float* __restrict__ out,
                                                                 int OH, int OW) {
  __shared__ int red[16];
  const int v = blockIdx.x;
  const int* pr = params + v * 12;
  const int slice = pr[0], a0 = pr[1], a1 = pr[2], a2 = pr[3], a3 = pr[4], a4 = pr[5], a5 = pr[6], flags = pr[7];
  const int top = pr[8], left = pr[9];
  const float b = __int_as_float(pr[10]), c = __int_as_float(pr[11]);
  const bool b_in = b >= 0.f && b <= 1.f, c_in = c >= 0.f && c <= 1.f;
  const float* img = src + (size_t)slice * HS * WS;
  const bool contrast_first = flags & 4;
  auto sample = [&](int p) -> int {
    const int i = p / OW, j = p - i * OW;
    int y = i + top, x = j + left;           // position in the rotated + flipped image (same size as the slice)
    if (flags & 1) x = WS - 1 - x;           // undo the horizontal flip
    if (flags & 2) y = HS - 1 - y;           // undo the vertical flip
    const int xin = (a2 + x * a0 + y * a1) >> 16, yin = (a5 + x * a3 + y * a4) >> 16;
    if (!(xin >= 0 && xin < WS && yin >= 0 && yin < HS)) return 0;
    return (int)__fadd_rn(__fmul_rn(img[(size_t)yin * WS + xin], 255.f), 0.5f);  // the 8-bit grey level of the store
  };
  const int np = OH * OW;
  int part = 0;
  for (int p = threadIdx.x; p < np; p += 1024) {
    int u = sample(p);
    if (!contrast_first) u = pil_blend(0, u, b, b_in);
    part += u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  long long total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) total += red[w];
  const int mean = (int)((2 * total + np) / (2LL * np));  // int(sum / count + 0.5), exactly
  float* o = out + (size_t)v * np;
  for (int p = threadIdx.x; p < np; p += 1024) {
    int u = sample(p);
    if (!contrast_first) u = pil_blend(0, u, b, b_in);
    u = pil_blend(mean, u, c, c_in);
    if (contrast_first) u = pil_blend(0, u, b, b_in);
    o[p] = __fdiv_rn((float)u, 255.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The reference's OTHER recipes in PIL's arithmetic (semi_seg/augment.py:23-37 ACDC `label` / `val`, :54-75 Prostate
// `pretrain` / `label` / `val`), and the interpolation its wrapper really selects: contrastyou/augment/synchronize.py:95-103
// runs the common transform with BILINEAR on images and NEAREST on targets, so an image is rotated by
// Image.rotate(angle, BILINEAR) -- Geometry.c ImagingGenericTransform: affine_transform in doubles, bilinear_filter8 -- and
// its label map by the fixed-point nearest loop above.  One view per workgroup, params[v][28]:
//   [0] slice  [1] flags (1 hflip, 2 vflip, 4 contrast before brightness, 8 bilinear image rotation, 16 crop FIRST: the
//   rotation turns the OH x OW crop about its own centre -- the `label` recipes: RandomCrop, then RandomRotation)
//   [2] top [3] left: the crop's offsets -- in the padded, rotated + flipped image, or (crop first) in the slice
//   [4] pad (RandomCrop(padding=), zeros)  [5] brightness, [6] contrast (f32 bits; 1.0 = the identity, exactly)
//   [8..13] the 16.16 nearest coefficients  [14..25] the six doubles of the rotation matrix (bilinear)
// Both are computed on the host exactly as PIL computes them (semi_seg/data/augment.py), for the image the rotation acts
// on: the whole slice, or the crop.  `labels` / `label_out` (optional): the slice's 8-bit label map through the same
// geometry with NEAREST, written as int64 (pil_augment.ToLabel).
constexpr int RECIPE_W = 28;
// one view's geometry and samplers, built from its parameter row (shared by the one-workgroup kernel and the split pair)
struct RecipeView {
  const float* img;
  const unsigned char* lab;
  int flags, top, left, pad, HS, WS, OH, OW, wx0, wy0, ww, wh;
  int a0, a1, a2, a3, a4, a5;
  double m[6];
  float b, c;
  bool b_in, c_in, contrast_first, bilinear, crop_first, slice_ok;

  __device__ __forceinline__ RecipeView(const float* src, const unsigned char* labels, int S, int HS_, int WS_, const int* pr,
                                        int OH_, int OW_) {
    const int slice = pr[0];
    flags = pr[1]; top = pr[2]; left = pr[3]; pad = pr[4];
    b = __int_as_float(pr[5]); c = __int_as_float(pr[6]);
    a0 = pr[8]; a1 = pr[9]; a2 = pr[10]; a3 = pr[11]; a4 = pr[12]; a5 = pr[13];
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = __hiloint2double(pr[14 + 2 * k + 1], pr[14 + 2 * k]);  // (little endian: low word first)
    b_in = b >= 0.f && b <= 1.f; c_in = c >= 0.f && c <= 1.f;
    contrast_first = flags & 4; bilinear = flags & 8; crop_first = flags & 16;
    HS = HS_; WS = WS_; OH = OH_; OW = OW_;
    img = src + (size_t)slice * HS * WS;
    lab = labels != nullptr ? labels + (size_t)slice * HS * WS : nullptr;
    // the image the rotation acts on: the slice, or its crop (a window of it)
    // (crop first: RandomCrop(padding=) pads with zeros BEFORE it crops -- the window's origin is (top - pad, left - pad) of
    // the slice and window pixels outside the slice are the padding's zeros; a slice index outside the store reads as all
    // padding)
    wx0 = crop_first ? left - pad : 0; wy0 = crop_first ? top - pad : 0;
    ww = crop_first ? OW : WS; wh = crop_first ? OH : HS;
    slice_ok = slice >= 0 && slice < S;
  }
  __device__ __forceinline__ bool inside(int xc, int yc) const {
    const int ys = wy0 + yc, xs = wx0 + xc;
    return slice_ok && ys >= 0 && ys < HS && xs >= 0 && xs < WS;
  }
  __device__ __forceinline__ int level(int xc, int yc) const {  // 8-bit grey level of window pixel (xc, yc)
    if (!inside(xc, yc)) return 0;
    return (int)__fadd_rn(__fmul_rn(img[(size_t)(wy0 + yc) * WS + wx0 + xc], 255.f), 0.5f);
  }
  // (x, y): the pixel of the ROTATED window this output pixel shows, or none
  __device__ __forceinline__ bool locate(int p, int& x, int& y) const {
    const int i = p / OW, j = p - i * OW;
    if (crop_first) { x = j; y = i; return true; }
    y = i + top - pad;
    x = j + left - pad;
    if (x < 0 || x >= WS || y < 0 || y >= HS) return false;  // the zero padding of RandomCrop(padding=)
    if (flags & 1) x = WS - 1 - x;
    if (flags & 2) y = HS - 1 - y;
    return true;
  }
  __device__ __forceinline__ int sample(int p) const {
    int x, y;
    if (!locate(p, x, y)) return 0;
    if (!bilinear) {
      const int xin = (a2 + x * a0 + y * a1) >> 16, yin = (a5 + x * a3 + y * a4) >> 16;
      return (xin >= 0 && xin < ww && yin >= 0 && yin < wh) ? level(xin, yin) : 0;
    }
    // affine_transform + bilinear_filter8 (Geometry.c), every product and sum rounded on its own as the C doubles are
    const double xc = (double)x + 0.5, yc = (double)y + 0.5;
    double xin = __dadd_rn(__dadd_rn(__dmul_rn(m[0], xc), __dmul_rn(m[1], yc)), m[2]);
    double yin = __dadd_rn(__dadd_rn(__dmul_rn(m[3], xc), __dmul_rn(m[4], yc)), m[5]);
    if (xin < 0.0 || xin >= (double)ww || yin < 0.0 || yin >= (double)wh) return 0;
    xin -= 0.5;
    yin -= 0.5;
    const int xf = xin < 0.0 ? (int)floor(xin) : (int)xin, yf = yin < 0.0 ? (int)floor(yin) : (int)yin;
    const double dx = xin - (double)xf, dy = yin - (double)yf;
    const int x0 = min(max(xf, 0), ww - 1), x1 = min(max(xf + 1, 0), ww - 1), y0 = min(max(yf, 0), wh - 1);
    const double p00 = (double)level(x0, y0), p01 = (double)level(x1, y0);
    double v1 = __dadd_rn(p00, __dmul_rn(__dsub_rn(p01, p00), dx));
    double v2 = v1;
    if (yf + 1 >= 0 && yf + 1 < wh) {
      const double p10 = (double)level(x0, yf + 1), p11 = (double)level(x1, yf + 1);
      v2 = __dadd_rn(p10, __dmul_rn(__dsub_rn(p11, p10), dx));
    }
    v1 = __dadd_rn(v1, __dmul_rn(__dsub_rn(v2, v1), dy));
    return (int)v1;  // (UINT8) of a value inside [0, 255]
  }
  // the level the contrast step sees (and averages): the sample, through the brightness step unless contrast comes first
  __device__ __forceinline__ int before_contrast(int p) const {
    const int u = sample(p);
    return contrast_first ? u : pil_blend(0, u, b, b_in);
  }
  __device__ __forceinline__ float finish(int u, int mean) const {
    u = pil_blend(mean, u, c, c_in);
    if (contrast_first) u = pil_blend(0, u, b, b_in);
    return __fdiv_rn((float)u, 255.f);
  }
  __device__ __forceinline__ long long label(int p) const {  // the label map through the same geometry, NEAREST
    int x, y, lv = 0;
    if (locate(p, x, y)) {
      const int xin = (a2 + x * a0 + y * a1) >> 16, yin = (a5 + x * a3 + y * a4) >> 16;
      if (xin >= 0 && xin < ww && yin >= 0 && yin < wh && inside(xin, yin)) lv = lab[(size_t)(wy0 + yin) * WS + wx0 + xin];
    }
    return lv;
  }
};

__global__ __launch_bounds__(1024) void augment_views_recipe_kernel(const float* __restrict__ src,
                                                                    const unsigned char* __restrict__ labels, int S, int HS,
                                                                    int WS, const int* __restrict__ params,
                                                                    float* __restrict__ out, long long* __restrict__ label_out,
                                                                    int OH, int OW) {
  __shared__ int red[16];
  const int v = blockIdx.x;
  const RecipeView rv(src, labels, S, HS, WS, params + v * RECIPE_W, OH, OW);
  const int np = OH * OW;
  int part = 0;
  for (int p = threadIdx.x; p < np; p += 1024) part += rv.before_contrast(p);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  long long total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) total += red[w];
  const int mean = (int)((2 * total + np) / (2LL * np));  // int(sum / count + 0.5), exactly
  float* o = out + (size_t)v * np;
  for (int p = threadIdx.x; p < np; p += 1024) {
    o[p] = rv.finish(rv.before_contrast(p), mean);
    if (label_out != nullptr) label_out[(size_t)v * np + p] = rv.label(p);
  }
}

// The same views from a grid that fills the chip (round 6).  One workgroup per view was 60 workgroups for a pre-train batch
// -- a quarter of the CUs -- each sampling its 50 176 pixels TWICE (the contrast step needs the view's mean first): 180 us,
// the longest launch of a training step on the product's own data path.  Split: G workgroups per view (four pixels per
// thread: a pixel is a chain of dependent gathers, the launch is bound by how many are in flight) sample a chunk each ONCE,
// park the levels as bytes and leave their integer partial sum; the second launch adds a view's partials in index order
// (integers: the mean is the one-workgroup kernel's, bit for bit) and finishes the pixels.  60 views of 224^2: 33.6 + 7.4 us
// with 16 workgroups per view, see profiles/r06_experiments/NOTES.md for the final split.
constexpr int AUG_SPLIT_MAX = 64;
static int aug_split(int np) {
  const int g = (np + 1023) / 1024;
  return g < 1 ? 1 : (g > AUG_SPLIT_MAX ? AUG_SPLIT_MAX : g);
}
__global__ __launch_bounds__(256) void augment_recipe_sample_kernel(const float* __restrict__ src,
                                                                   const unsigned char* __restrict__ labels, int S, int HS,
                                                                   int WS, const int* __restrict__ params,
                                                                   unsigned char* __restrict__ levels,
                                                                   long long* __restrict__ partials,
                                                                   long long* __restrict__ label_out, int OH, int OW) {
  __shared__ int red[4];
  const int v = blockIdx.y, g = blockIdx.x, AUG_SPLIT = gridDim.x;
  const RecipeView rv(src, labels, S, HS, WS, params + v * RECIPE_W, OH, OW);
  const int np = OH * OW, chunk = (np + AUG_SPLIT - 1) / AUG_SPLIT;
  const int p0 = g * chunk, p1 = min(np, p0 + chunk);
  int part = 0;
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    const int u = rv.before_contrast(p);
    levels[(size_t)v * np + p] = (unsigned char)u;
    part += u;
    if (label_out != nullptr) label_out[(size_t)v * np + p] = rv.label(p);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) partials[v * AUG_SPLIT + g] = (long long)red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void augment_recipe_finish_kernel(const int* __restrict__ params,
                                                                   const unsigned char* __restrict__ levels,
                                                                   const long long* __restrict__ partials,
                                                                   float* __restrict__ out, int np, int AUG_SPLIT) {
  const int v = blockIdx.y;
  const int* pr = params + v * RECIPE_W;
  const int flags = pr[1];
  const float b = __int_as_float(pr[5]), c = __int_as_float(pr[6]);
  const bool b_in = b >= 0.f && b <= 1.f, c_in = c >= 0.f && c <= 1.f, contrast_first = flags & 4;
  // the view's sum: one partial per lane of the first wave (AUG_SPLIT <= 64), added by shuffles -- integers: any order is exact
  __shared__ long long tot_s;
  if (threadIdx.x < 64) {
    long long x = (int)threadIdx.x < AUG_SPLIT ? partials[v * AUG_SPLIT + threadIdx.x] : 0LL;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (threadIdx.x == 0) tot_s = x;
  }
  __syncthreads();
  const long long total = tot_s;
  const int mean = (int)((2 * total + np) / (2LL * np));  // int(sum / count + 0.5), exactly
  const size_t base = (size_t)v * np;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256) {
    int u = pil_blend(mean, (int)levels[base + p], c, c_in);
    if (contrast_first) u = pil_blend(0, u, b, b_in);
    out[base + p] = __fdiv_rn((float)u, 255.f);
  }
}

// Image.resize((OW, OH), BILINEAR) of 8-bit images (Resample.c; torchvision Resize, semi_seg/augment.py:56,71,79): one pass
// along x, one along y, each with the per-output coefficient rows the host precomputed exactly as precompute_coeffs /
// normalize_coeffs_8bpc do (kk: 22 fractional bits, bounds: first tap / tap count), an 8-bit intermediate between them.
// Applied ONCE when a store is built (the transform is deterministic), not per view.
__global__ __launch_bounds__(256) void resize_pass_kernel(const float* __restrict__ src, int n_lines, int in_len, int out_len,
                                                         int line_stride_in, int elem_stride_in, int line_stride_out,
                                                         int elem_stride_out, size_t img_stride_in, size_t img_stride_out,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                         float* __restrict__ dst) {
  const int img = blockIdx.y;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n_lines * out_len) return;
  const int line = (int)(idx / out_len), xx = (int)(idx - (size_t)line * out_len);
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const float* in = src + img * img_stride_in + (size_t)line * line_stride_in;
  long long ss = 1LL << 21;
  for (int x = 0; x < xmax; ++x) {
    const int lvl = (int)__fadd_rn(__fmul_rn(in[(size_t)(x + xmin) * elem_stride_in], 255.f), 0.5f);
    ss += (long long)lvl * kk[(size_t)xx * ksize + x];
  }
  long long q = ss >> 22;
  q = q < 0 ? 0 : (q > 255 ? 255 : q);
  dst[img * img_stride_out + (size_t)line * line_stride_out + (size_t)xx * elem_stride_out] = __fdiv_rn((float)q, 255.f);
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_augment_views_recipe(const float* src, const unsigned char* labels, int S, int HS, int WS,
                                         const int* params, int nviews, float* out, long long* label_out, int OH, int OW,
                                         int max_pad, void* stream) {
  SPCL_CHECK_ARG(src && params && out, "augment_views_recipe: null pointer");
  SPCL_CHECK_ARG((labels != nullptr) == (label_out != nullptr), "augment_views_recipe: labels come with label_out");
  SPCL_CHECK_ARG(S > 0 && HS > 0 && WS > 0 && nviews > 0 && OH > 0 && OW > 0 && max_pad >= 0 && OH <= HS + 2 * max_pad &&
                     OW <= WS + 2 * max_pad && HS <= 4096 && WS <= 4096,
                 "augment_views_recipe: bad shape (crop %dx%d of %dx%d + %d)", OH, OW, HS, WS, max_pad);
  SPCL_LAUNCH(augment_views_recipe_kernel, dim3(nviews), dim3(1024), 0, (hipStream_t)stream, src, labels, S, HS, WS, params, out,
              label_out, OH, OW);
  SPCL_LAUNCH_CHECK("augment_views_recipe");
  return SPCL_OK;
}

extern "C" size_t spcl_augment_views_recipe_workspace_bytes(int nviews, int OH, int OW) {
  if (nviews <= 0 || OH <= 0 || OW <= 0) return 0;
  return (size_t)nviews * AUG_SPLIT_MAX * sizeof(long long) + (size_t)nviews * OH * OW;
}

extern "C" int spcl_augment_views_recipe_ws(const float* src, const unsigned char* labels, int S, int HS, int WS,
                                            const int* params, int nviews, float* out, long long* label_out, int OH, int OW,
                                            int max_pad, void* workspace, size_t workspace_bytes, void* stream) {
  SPCL_CHECK_ARG(src && params && out && workspace, "augment_views_recipe_ws: null pointer");
  SPCL_CHECK_ARG((labels != nullptr) == (label_out != nullptr), "augment_views_recipe_ws: labels come with label_out");
  SPCL_CHECK_ARG(S > 0 && HS > 0 && WS > 0 && nviews > 0 && nviews <= 65535 && OH > 0 && OW > 0 && max_pad >= 0 &&
                     OH <= HS + 2 * max_pad && OW <= WS + 2 * max_pad && HS <= 4096 && WS <= 4096,
                 "augment_views_recipe_ws: bad shape (crop %dx%d of %dx%d + %d)", OH, OW, HS, WS, max_pad);
  SPCL_CHECK_ARG(workspace_bytes >= spcl_augment_views_recipe_workspace_bytes(nviews, OH, OW) && (uintptr_t)workspace % 8 == 0,
                 "augment_views_recipe_ws: workspace of %zu bytes, 8-byte aligned",
                 spcl_augment_views_recipe_workspace_bytes(nviews, OH, OW));
  long long* partials = (long long*)workspace;
  unsigned char* levels = (unsigned char*)(partials + (size_t)nviews * AUG_SPLIT_MAX);
  const int np = OH * OW, G = aug_split(np);
  hipStream_t st = (hipStream_t)stream;
  SPCL_LAUNCH(augment_recipe_sample_kernel, dim3(G, nviews), dim3(256), 0, st, src, labels, S, HS, WS, params, levels,
              partials, label_out, OH, OW);
  int fin = (np + 256 * 8 - 1) / (256 * 8);  // eight pixels per thread
  if (fin < 1) fin = 1;
  SPCL_LAUNCH(augment_recipe_finish_kernel, dim3(fin, nviews), dim3(256), 0, st, params, (const unsigned char*)levels,
              (const long long*)partials, out, np, G);
  SPCL_LAUNCH_CHECK("augment_views_recipe_ws");
  return SPCL_OK;
}

extern "C" int spcl_resize_bilinear_pil(const float* src, int S, int HS, int WS, const int* bounds_x, const int* kk_x,
                                        int ksize_x, const int* bounds_y, const int* kk_y, int ksize_y, float* tmp, float* out,
                                        int OH, int OW, void* stream) {
  SPCL_CHECK_ARG(src && bounds_x && kk_x && bounds_y && kk_y && tmp && out, "resize_bilinear_pil: null pointer");
  SPCL_CHECK_ARG(S > 0 && HS > 0 && WS > 0 && OH > 0 && OW > 0 && ksize_x > 0 && ksize_y > 0, "resize_bilinear_pil: bad shape");
  hipStream_t st = (hipStream_t)stream;
  // horizontal: [S][HS][WS] -> tmp [S][HS][OW]; vertical: tmp -> out [S][OH][OW]
  SPCL_LAUNCH(resize_pass_kernel, dim3((unsigned)(((size_t)HS * OW + 255) / 256), S), dim3(256), 0, st, src, HS, WS, OW, WS, 1, OW,
              1, (size_t)HS * WS, (size_t)HS * OW, bounds_x, kk_x, ksize_x, tmp);
  SPCL_LAUNCH(resize_pass_kernel, dim3((unsigned)(((size_t)OW * OH + 255) / 256), S), dim3(256), 0, st, (const float*)tmp, OW, HS,
              OH, 1, OW, 1, OW, (size_t)HS * OW, (size_t)OH * OW, bounds_y, kk_y, ksize_y, out);
  SPCL_LAUNCH_CHECK("resize_bilinear_pil");
  return SPCL_OK;
}

extern "C" int spcl_augment_views_pil(const float* src, int S, int HS, int WS, const int* params, int nviews, float* out,
                                      int OH, int OW, void* stream) {
  SPCL_CHECK_ARG(src && params && out, "augment_views_pil: null pointer");
  SPCL_CHECK_ARG(S > 0 && HS > 0 && WS > 0 && nviews > 0 && OH > 0 && OW > 0 && OH <= HS && OW <= WS && HS <= 4096 &&
                     WS <= 4096,
                 "augment_views_pil: bad shape (crop %dx%d of %dx%d)", OH, OW, HS, WS);
  SPCL_LAUNCH(augment_views_pil_kernel, dim3(nviews), dim3(1024), 0, (hipStream_t)stream, src, S, HS, WS, params, out, OH,
              OW);
  SPCL_LAUNCH_CHECK("augment_views_pil");
  return SPCL_OK;
}

extern "C" int spcl_augment_views(const float* src, int S, int HS, int WS, const int* params, int nviews, float* out,
                                  int OH, int OW, void* stream) {
  SPCL_CHECK_ARG(src && params && out, "augment_views: null pointer");
  SPCL_CHECK_ARG(S > 0 && HS > 0 && WS > 0 && nviews > 0 && OH > 0 && OW > 0 && OH <= HS && OW <= WS && HS <= 4096 &&
                     WS <= 4096,
                 "augment_views: bad shape (crop %dx%d of %dx%d)", OH, OW, HS, WS);
  SPCL_LAUNCH(augment_views_kernel, dim3(nviews), dim3(1024), 0, (hipStream_t)stream, src, S, HS, WS, params, out, OH, OW);
  SPCL_LAUNCH_CHECK("augment_views");
  return SPCL_OK;
}

extern "C" int spcl_flip_batch(const void* x, void* out, int elem_size, int N, int C, int H, int W,
                               const uint8_t* flags, void* stream) {
  SPCL_CHECK_ARG(x && out && flags, "flip_batch: null pointer");
  SPCL_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "flip_batch: bad shape");
  SPCL_CHECK_ARG(x != out, "flip_batch: in-place is not supported");
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 4) launch_flip<uint32_t>(x, out, N, C, H, W, flags, st);
  else if (elem_size == 2) launch_flip<uint16_t>(x, out, N, C, H, W, flags, st);
  else {
    set_error("flip_batch: elem_size %d", elem_size);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("flip_batch");
  return SPCL_OK;
}

// spcl_flip_pair with the flags taken from a block of host bytes that the same launch also writes to `stage_dst` (see
// spcl_stage_bytes): host_src[0 .. nbytes) (nbytes <= 3 584, multiple of 4), the N flag bytes at host_src + flag_off.
extern "C" int spcl_flip_pair_stage(const void* first, const void* second, void* out, int elem_size, int N, int C, int H,
                                    int W, void* stage_dst, const void* host_src, size_t nbytes, size_t flag_off,
                                    void* stream) {
  SPCL_CHECK_ARG(first && second && out && stage_dst && host_src, "flip_pair_stage: null pointer");
  SPCL_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "flip_pair_stage: bad shape");
  SPCL_CHECK_ARG(first != out && second != out, "flip_pair_stage: in-place is not supported");
  SPCL_CHECK_ARG(nbytes % 4 == 0 && nbytes <= (size_t)FLIP_STAGE_WORDS * 4 && flag_off + (size_t)N <= nbytes &&
                     (uintptr_t)stage_dst % 4 == 0,
                 "flip_pair_stage: at most %d staged bytes, the flags inside them", FLIP_STAGE_WORDS * 4);
  FlipStage s;
  memcpy(s.w, host_src, nbytes);
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto tag) {
    typedef decltype(tag) T;
    constexpr int V = 16 / (int)sizeof(T);
    const bool vec = W % V == 0 && ((uintptr_t)second % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)first % 16 == 0);
    const size_t total = (size_t)2 * N * C * H * (vec ? W / V : W);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (vec) SPCL_LAUNCH((flip_pair_stage_kernel<T, V>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)second, (T*)out,
                         N, C, H, W, (const T*)first, N, (uint32_t*)stage_dst, (int)(nbytes / 4), (int)flag_off, s);
    else SPCL_LAUNCH((flip_pair_stage_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)second, (T*)out, N,
                     C, H, W, (const T*)first, N, (uint32_t*)stage_dst, (int)(nbytes / 4), (int)flag_off, s);
  };
  if (elem_size == 4) go(uint32_t{});
  else if (elem_size == 2) go(uint16_t{});
  else {
    set_error("flip_pair_stage: elem_size %d", elem_size);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("flip_pair_stage");
  return SPCL_OK;
}

// The pre-train step's input pair in ONE launch (new_pretrain.py:57-58,93: flip view 2 per sample, then
// torch.cat([view 1, flipped view 2])): out [2N][C][H][W] = [ first | flip(second, flags) ].
extern "C" int spcl_flip_pair(const void* first, const void* second, void* out, int elem_size, int N, int C, int H, int W,
                              const uint8_t* flags, void* stream) {
  SPCL_CHECK_ARG(first && second && out && flags, "flip_pair: null pointer");
  SPCL_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "flip_pair: bad shape");
  SPCL_CHECK_ARG(first != out && second != out, "flip_pair: in-place is not supported");
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 4) launch_flip<uint32_t>(second, out, N, C, H, W, flags, st, first, N);
  else if (elem_size == 2) launch_flip<uint16_t>(second, out, N, C, H, W, flags, st, first, N);
  else {
    set_error("flip_pair: elem_size %d", elem_size);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("flip_pair");
  return SPCL_OK;
}
