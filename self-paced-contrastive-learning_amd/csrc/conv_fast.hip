// Compile-time-specialised 3x3 convolution kernels (bf16, NHWC, 14-pixel-wide tiles) for the layer shapes of the UNet
// encoder (semi_seg/arch/unet.py:67-82,123-131): same mapping, packed-weight layout and numerics as the generic kernel
// of conv.hip, but the K-chunking, tile shape, n-tiles per wave, input mode and workgroup width are template
// parameters.  What that buys (measured with tools/experiments/convlab.hip):
//   * the tile comes from a 3-D grid (no integer divisions), the halo walk has compile-time pixel steps on top of one
//     per-lane offset, all global loads of a slab are issued back to back before the first use;
//   * the k-loop is fully unrolled: tap offsets fold into the ds_read_b128 offset field, weight fragments of the
//     16-channel layers are all fetched before staging so their L2 latency hides under the activation loads;
//   * BatchNorm partials leave as three 16-byte stores per 16-lane row.
// On the 16->16 layer at 224^2 this is 47 us against 80 us for the generic kernel (pure copy of the same tiles: 42).
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include "bn_acc.hpp"
#include "conv_common.hpp"

namespace spcl {

struct FastArgs {
  const unsigned char* x;
  const unsigned char* x2;  // non-null: the second half of the input channels comes from this tensor (ConvArgs::x2)
  unsigned char* y;
  unsigned char* y_hi;      // non-null (MODE 0): output channels [CoutS / 2, CoutS) go to this tensor (ConvArgs::y_hi)
  int up2;                  // 1: x is [N][H / 2][W / 2][CinK] and the convolution's input its nearest x2 upsample (ConvArgs::x_up2)
  const u32x4* wp;
  float* stats;
  const float* in_scale;
  const float* in_shift;
  int N, H, W, CinK, CoutS, tilesX, tilesY, gy;
  unsigned long long* stamps;  // debug (SPCL_FAST_STAMPS=1): s_memtime ticks of wave 0 per phase, else null
  int lds_flip;   // bytes between the two halo images of the cross-slab pipeline (0: one image)
  int xcd_remap;  // 1: tiles of an image are dealt to the XCDs in contiguous row-major blocks (see the kernel)
  // MODE 2 (dgrad whose output g is the gradient of relu(bn(y2))): per-tile partial sums of that BatchNorm's backward
  const unsigned char* y2;  // [N][H][W][CoutS] raw conv output of the layer being differentiated
  const float* scale2;      // its BN scale / shift / mean, [CoutS]
  const float* shift2;
  const float* mean2;
  float* rows2;             // [tile][2][CoutS]: sum dz, sum dz (y2 - mean)
  // MODE 3 (dgrad whose output g is the gradient of maxpool2x2(relu(bn(y2)))): y2 is [N][H2][W2][CoutS], H = H2 / 2
  int H2, W2;
  // MODE 4 (= MODE 2 for the layer behind a one-channel f32 image): rows2 is [tile][11][CoutS]; rows 2 .. 10 hold
  // sum_p dz[p][co] img[p + tap], the data-dependent part of that layer's weight gradient (dy = scale dz + A y + B: the
  // A and B terms need no pass over the activations, bn.hip image3)
  const float* img2;
  float* acorr_rows;  // conv3x3_image_kernel<.., ACORR = true>: [tile][64] autocorrelation partial rows of the image (ConvArgs::acorr_rows)
  // BatchNorm sums as fixed-point accumulator blocks (bn_acc.hpp) instead of per-tile rows + a reduction launch:
  long long* stats_acc;  // non-null: the output's sum x, sum x^2 are ADDED here (`stats` is then ignored)
  long long* rows2_acc;  // non-null (MODE 2 / 3): sum dz, sum dz (y2 - mean) are ADDED here (`rows2` is then only a mode marker)
  BnAccFwd in_bn;        // in_bn.acc non-null (MODE 5 .. 7): scale / shift of the input's BatchNorm are DERIVED from that block in
                         // the prologue (in_scale / in_shift unused); the launch's first workgroup writes in_bn.st and the running statistics
};

// waves per SIMD the register allocator must leave room for: what the tile's LDS footprint allows anyway, at most 4
// (fewer waves than that would not be resident regardless, so the extra registers are free)
// Halo image in LDS: pixel (hy, hx) at (hy * RP + hx) * PS.  PS = the channels + 32 bytes: an EVEN number of 16-byte
// slots, so lane groups g and g + 1 of a fragment read sit on slots of different parity and 8 pixels with distinct
// index mod 8 on 8 different slots of that parity.  ds_read_b128 serves 16 lanes per LDS cycle -- lanes {0-3, 12-15} of
// one k-group with {4-11} of the next (MI355X_MICROARCH, LDS) -- so a read is conflict-free when the pixel index q of
// fragment pixel p is congruent to p mod 8.  With 16-pixel rows the 14-wide tile makes q = p + 2 py and every row wrap
// inside a lane group collides (PMC, round 2: 44 % of all LDS cycles were bank conflicts); RP = 22 gives q = p + 8 py.
#ifndef SPCL_FAST_WPE_NT1
#define SPCL_FAST_WPE_NT1 2  // (3 waves per SIMD measured 6-14 us per step slower: spills + a 6-step ring)
#endif
#ifndef SPCL_FAST_WPE_NT1_NW4
#define SPCL_FAST_WPE_NT1_NW4 3
#endif
#ifndef SPCL_FAST_WR_NT1
#define SPCL_FAST_WR_NT1 9
#endif
#ifndef SPCL_FAST_WPE_KC32
#define SPCL_FAST_WPE_KC32 2  // waves per SIMD the 32-channel-slab kernels (a slab's 9 x NT weight fragments in registers) must fit
#endif
#ifndef SPCL_FAST_YPRE_MINKC
#define SPCL_FAST_YPRE_MINKC 16
#endif
#ifndef SPCL_FAST_WIDE_STORES
#define SPCL_FAST_WIDE_STORES 1
#endif
#ifndef SPCL_FAST_DBG
#define SPCL_FAST_DBG 0  /* timing experiments only (wrong results): 1 no ring refills, 2 no fragment reads in the k-loop, 4 the 14^2 layers as two workgroups per (tile, cout block) with half of the slabs each (what a K-split pair of wave groups would run like) */
#endif
// SPCL_FAST_ROWMAP (round 4): the m-tiles lie along the TILE ROWS -- m-tile i = row i of the tile as 16 pixel columns, of
// which the last two read halo columns 16, 17 (never staged: whatever they hold only reaches D's columns 14, 15, which nobody
// uses) -- instead of 16 consecutive pixels of the 14-wide tile in linear order.  A lane's pixel column is r16 in every
// m-tile and its row is the m-tile's index: fragment bases, output offsets and the statistics' positions are ONE per-lane
// value plus compile-time / wave-uniform steps (the linear order cost a division per m-tile base, a carry walk per m-tile in
// the epilogue, validity predicates in the last m-tile), the 16 lanes of every ds_read_b128 group sit in one halo row (no
// bank conflicts whatever the row pitch: the 22-pixel pitch -- 27-40 % padding -- goes back to 18, i.e. more workgroups per
// CU where the halo image decides the occupancy).  7-row tiles keep 7 m-tiles, 14-row tiles take 14 for 13 (+ 8 % MFMAs on
// the two 224^2 layers, which are bound by vector instructions).  Same k order per output: outputs bit for bit; the
// statistics add the same terms in another lane order.
#ifndef SPCL_FAST_ROWMAP
#define SPCL_FAST_ROWMAP 1
#endif
#ifndef SPCL_FAST_RP
#define SPCL_FAST_RP (SPCL_FAST_ROWMAP ? 18 : 22)
#endif
#ifndef SPCL_FAST_RP_NARROW
#define SPCL_FAST_RP_NARROW (SPCL_FAST_ROWMAP ? 18 : 22)
#endif
constexpr int fast_pixel_stride(int KC) { return KC == 16 ? 32 : KC * 2 + 32; }
constexpr int fast_row_pitch(int KC) { return KC == 64 ? SPCL_FAST_RP : SPCL_FAST_RP_NARROW; }
constexpr int fast_lds_bytes(int KC, int TH) { return (TH + 2) * fast_row_pitch(KC) * fast_pixel_stride(KC); }
// all weight fragments of a channel slab are requested BEFORE the slab's staging (their L2 latency, ~1 us each with
// only a handful in flight otherwise, hides under the activation loads): 9 x NT (KC = 32) or 18 x NT (KC = 64)
// fragments of 4 registers; a one-wave KC = 64 workgroup also holds 18 staging chunks and would spill
constexpr bool fast_preload_slab(int KC, int NW, int NT) { return KC == 32 || (KC == 64 && NW >= 2 && NT >= 1); }
constexpr int fast_wpe(int KC, int TH, int NW, int NT, int MODE = 0) {
  const int wgs = 160 * 1024 / fast_lds_bytes(KC, TH);
  int w = (wgs * NW + 3) / 4;
  w = w > 4 ? 4 : (w < 1 ? 1 : w);
  const int acc = (SPCL_FAST_ROWMAP ? TH : (TH * 14 + 15) / 16) * NT * 4;  // accumulator registers of a wave
  if (KC == 16 && MODE >= 2 && SPCL_FAST_YPRE_MINKC <= 16 && w > 3 && acc <= 80) return 3;  // room for the y2 requests
  if (acc > 80 && w > 2) return 2;               // 13 m-tiles x 2 n-tiles: give the allocator 256 registers
  // 9-step ring of one n-tile: 36 registers.  Four-wave workgroups (every 64 -> 64 layer): three waves per SIMD = three
  // workgroups per CU since the row order freed the m-tile bases (168 registers, one harmless address spill)
  if (KC == 64 && NT == 1 && NW == 4 && w > 2) return SPCL_FAST_WPE_NT1_NW4;
  if (KC == 64 && NT == 1 && NW >= 2 && w > 2) return SPCL_FAST_WPE_NT1;
  if (KC == 32 && fast_preload_slab(KC, NW, NT) && w > SPCL_FAST_WPE_KC32) return SPCL_FAST_WPE_KC32;
  if (fast_preload_slab(KC, NW, NT) && w > 2) return 2;  // a slab's weight fragments live in registers (PRELOAD_SLAB)
  return w;
}

template <int KC, int TH, int NT, int MODE, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(fast_wpe(KC, TH, NW, NT, MODE)))) void
conv3x3_fast_kernel(FastArgs a) {
  constexpr int TW = 14, HW_ = 16, CP = KC / 8, NHALO = (TH + 2) * HW_, PS = fast_pixel_stride(KC);
  constexpr int RP = fast_row_pitch(KC);  // LDS row pitch in pixels (HW_ of them are halo pixels)
  constexpr int NTHR = 64 * NW, NCH = NHALO * CP, ITER = (NCH + NTHR - 1) / NTHR, QS = NTHR / CP;
  constexpr bool RM = SPCL_FAST_ROWMAP != 0;
  constexpr int NPIX = TH * TW, MT = RM ? TH : (NPIX + 15) / 16, NSTEPS = (9 * CP + 3) / 4;
  constexpr bool PRELOAD_W = KC == 16;  // 5 k-steps: every weight fragment of the wave lives in registers
  constexpr bool PRELOAD_SLAB = fast_preload_slab(KC, NW, NT);
  static_assert(QS >= HW_ ? QS % HW_ == 0 : HW_ % QS == 0, "staging walk needs whole/even halo rows per iteration");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g = lane >> 4;
  // phase timing of wave 0 (debug builds only, -DSPCL_FAST_STAMPS_BUILD=1 + SPCL_FAST_STAMPS=1: the eight counter registers
  // alone push the 128-register kernels into spilling)
#ifndef SPCL_FAST_STAMPS_BUILD
#define SPCL_FAST_STAMPS_BUILD 0
#endif
  const bool stamp = SPCL_FAST_STAMPS_BUILD && a.stamps != nullptr && t == 0;
  unsigned long long t_begin = 0, t_k0 = 0, t_k1 = 0, t_store = 0;
  if (stamp) t_begin = __builtin_amdgcn_s_memtime();
  // Workgroups go to the 8 XCDs round-robin in launch order (x fastest): horizontally adjacent tiles would sit on
  // different XCDs and fetch their shared halo columns from HBM twice.  When an image's tile count divides by 8 each XCD
  // gets a block of consecutive tiles in row-major order instead (whole tile rows at 224^2: both halo directions hit L2).
  int tx = blockIdx.x, ty = blockIdx.y;
  {
    const int T = a.tilesX * a.tilesY;
    if (a.xcd_remap && (T & 7) == 0) {
      const int L = ty * a.tilesX + tx;
      const int L2 = (L & 7) * (T >> 3) + (L >> 3);
      ty = L2 / a.tilesX;
      tx = L2 - ty * a.tilesX;
    }
  }
  int n = blockIdx.z, by = 0;
#if SPCL_FAST_DBG & 4  /* timing experiment (wrong results): two workgroups per (tile, cout block), each half of the slabs */
  if (KC == 64 && a.CinK >= 128 && gridDim.z == 2u * a.N * a.gy) n >>= 1;
#endif
  if (a.gy > 1) {
    // z = by * N + n, the output-channel block SLOWEST: workgroups go to the XCDs round-robin in launch order, and the gy
    // workgroups that stage the SAME halo (same image, same tile, another 64 output channels) must meet in one XCD's L2.
    // With the channel block fastest (z = n * gy + by, rounds 2 - 5) they sat tilesX * tilesY apart in launch order, i.e.
    // on gy different XCDs at 14^2 / 28^2: every halo came out of the Infinity Cache gy times.  Whole step, same box,
    // nine rounds: -3 us (Conv4 / Conv5 and their dgrads; profiles/r06_experiments/NOTES.md).
    by = n / a.N;
    n -= by * a.N;
  }
  // image sizes that are not a multiple of the tile: the last tile of a row / column is shifted back inside the image
  // (it recomputes oy rows / ox columns of its neighbour -- identical values, written twice -- and leaves them out of
  // its statistics)
  const int y0 = min(ty * TH, a.H - TH), x0 = min(tx * TW, a.W - TW);
  const int oy = ty * TH - y0, ox = tx * TW - x0;
  const int tile = (n * a.tilesY + ty) * a.tilesX + tx;
  const int ntn = a.CoutS >> 4;
  const int nt0 = (by * NW + wave) * NT;
  const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + TW < a.W;  // whole halo inside the image
#if SPCL_FAST_DBG & 4
  const int nslab = KC < 64 ? 1 : ((a.CinK >= 128 && gridDim.z == 2u * a.N * a.gy) ? a.CinK / KC / 2 : a.CinK / KC);
#else
  const int nslab = KC < 64 ? 1 : a.CinK / KC;
#endif
  const int gps = (KC < 64 ? KC : a.CinK) * 2;  // bytes per pixel of x

  constexpr bool M2 = MODE == 2 || MODE == 4;  // BN-backward sums of the layer whose activation gradient this is
  // MODE 5 / 6 / 7 = MODE 1 (relu(scale x + shift) of the input's BatchNorm in the loader) with scale / shift DERIVED in the
  // prologue from a fixed-point accumulator block (bn_acc.hpp), the block's eight replicas of a channel split over 4 / 2 / 1
  // threads (= workgroup threads / input channels).  Instantiations of their own: the prologue's registers (2 / 4 / 8 replicas
  // x 32 bytes in flight per thread) are allocated for the form present, and must not weigh on the block 1 / 2 kernels at all
  constexpr bool M1 = MODE == 1 || MODE >= 5;
  constexpr int ACC_TPC = MODE == 5 ? 4 : (MODE == 6 ? 2 : 1);
  // MODE 4: the lane's four pixels of the 16 x 16 image halo (row lane / 4, columns 4 (lane % 4) ..), zero outside the
  // image; parked in registers across the k-loop, written to LDS when the activation halo is no longer needed
  float imgv[MODE == 4 ? 4 : 1];
  if (MODE == 4) {
    const int hr = lane >> 2, hc = (lane & 3) * 4;
    const int gy = y0 - 1 + hr;
    const float* ir = a.img2 + ((size_t)n * a.H + (gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy))) * a.W;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gx = x0 - 1 + hc + e;
      const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const float xv = ir[gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx)];  // (unconditional clamped load, then select)
      imgv[e] = in ? xv : 0.f;
    }
  }

  u32x4 wall[PRELOAD_W ? NSTEPS : 1][NT];
  if (PRELOAD_W) {
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j) wall[s][j] = a.wp[(size_t)(s * ntn + nt0 + j) * 64 + lane];
  }

  // staging map: thread -> (halo pixel q0 + k QS, 16-byte channel chunk ch); QS and the halo width are powers of two,
  // so iteration k moves by a compile-time (dky, dkx)
  const int ch = t & (CP - 1), q0 = t / CP;
  const int hy0 = q0 / HW_, hx0 = q0 % HW_;
  // (two sources: a pixel's chunks [0, CP / 2) sit in x, the others in x2, each tensor with half the pixel stride -- a
  // per-thread choice of base pointer, fixed for the launch)
  // With more than one slab (CinK = 128 / 256) a whole slab lies in one tensor: a wave-uniform choice per slab instead.
  const bool two = a.x2 != nullptr;
  const bool two_chunks = two && a.CinK == KC;
  const int gps1 = two ? gps / 2 : gps;
  const int ch1 = two_chunks ? (ch & (CP / 2 - 1)) : ch;
  const long xorg = (((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * gps1;
  const unsigned char* xb = ((two_chunks && ch >= CP / 2) ? a.x2 : a.x) + xorg;  // halo origin (may be outside)
  const unsigned char* xb2 = two ? a.x2 + xorg : nullptr;
  const unsigned voff = (unsigned)((hy0 * a.W + hx0) * gps1 + ch1 * 16);
  unsigned char* const lp = lds + (hy0 * RP + hx0) * PS + ch * 16;

  // per-lane LDS base of each m-tile's pixel p = 16 i + r16 (+ the lane's k-group when a step stays inside one tap)
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (RM) {
      abase[i] = (i * RP + r16) * PS + (CP >= 4 ? g * 16 : 0);
    } else {
      int p = 16 * i + r16;
      if (p >= NPIX) p = 0;
      const int py = p / TW, px = p - py * TW;
      abase[i] = (py * RP + px) * PS + (CP >= 4 ? g * 16 : 0);
    }
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Slabs of KC input channels, software-pipelined across slabs when the launcher provides two halo images (flip != 0;
  // one wave per SIMD on the 14^2 layers, so nothing else hides a slab's load phase): the NEXT slab's halo is requested
  // before this slab's k-loop and lands in registers while the MFMAs run, and the weight fragments stream through a ring
  // of WR k-steps: a ring slot is refilled with the fragment WR steps ahead (same slab or the next one) right after its
  // use.  The last slab is peeled (no refill across its end), so that every path into the loop head has the same order
  // of outstanding loads -- halo first, then WR x NT fragments -- and the counted waits stay exact (vmcnt is in-order).
  const int flip = a.lds_flip;
  const unsigned wvo = (unsigned)(nt0 * 64 + lane) * 16u;  // the lane's byte offset inside a k-step's fragment row
  const size_t wstep = (size_t)ntn * 1024;                  // bytes per k-step
  u32x4 v[ITER];
  float ssc[8], ssh[8];
  constexpr bool STREAM_W = PRELOAD_SLAB && KC == 64;  // (the only shape with more than one slab)
  constexpr int WR = STREAM_W ? (NT == 1 ? SPCL_FAST_WR_NT1 : 9) : (PRELOAD_SLAB ? NSTEPS : 1);
  u32x4 wsl[WR][NT];
  // MODE 5 .. 7: thread (part, c) turns its share of channel c's replicas into partial integer sums -- its loads are the FIRST
  // of the workgroup, so they return first and the arithmetic runs under the halo's flight -- the parts meet in LDS, thread c
  // derives scale / shift and parks them in LDS behind the halo image(s); every thread then takes its chunk's eight from there
  // (slab 0: after the barrier below; later slabs: plain LDS reads).  The first workgroup of the launch also writes mean /
  // invstd / scale / shift and the running statistics (what the finalize launch used to leave for backward / eval).
  constexpr bool acc_in = MODE >= 5;
  float* const coef_l = (float*)(lds + fast_lds_bytes(KC, TH) * (a.lds_flip != 0 ? 2 : 1));  // [2][CinK]
  auto load_coef_lds = [&](int slab) {
    const int cc = two_chunks ? ch1 : ch;
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *(f32x4*)&ssc[e] = *(const f32x4*)(coef_l + slab * KC + cc * 8 + e);
      *(f32x4*)&ssh[e] = *(const f32x4*)(coef_l + a.CinK + slab * KC + cc * 8 + e);
    }
  };
  auto issue_halo = [&](int slab, const bool first = false) {
    if (acc_in) {
      if (!first) load_coef_lds(slab);
    } else if (M1) {
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        // (two tensors, one slab: the coefficients belong to the SECOND tensor's channels -- x is an activation already, x2
        // the raw output of the convolution whose BatchNorm + ReLU this loader applies; chunks of x keep ch1's entry unused)
        const int cc = two_chunks ? ch1 : ch;
        *(f32x4*)&ssc[e] = *(const f32x4*)(a.in_scale + slab * KC + cc * 8 + e);
        *(f32x4*)&ssh[e] = *(const f32x4*)(a.in_shift + slab * KC + cc * 8 + e);
      }
    }
    const int hslab = (a.CinK / KC) >> 1;  // slabs per tensor when the input is two tensors of whole slabs
    const unsigned char* xs = (two && !two_chunks) ? (slab >= hslab ? xb2 + (slab - hslab) * (KC * 2) : xb + slab * (KC * 2))
                                                   : xb + slab * (KC * 2);
    // (three copies of the walk, chosen once by wave-uniform branches: an interior tile -- most of them -- runs no bounds
    // arithmetic at all; with the tests inside the loop every iteration carried two uniform branches and their scalar set-up,
    // a tenth of the instructions of a 32-channel tile)
    auto walk = [&](auto up2_c, auto interior_c) {
      constexpr bool UP2 = decltype(up2_c)::value, INTERIOR = decltype(interior_c)::value;
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int dky = QS >= HW_ ? k * (QS / HW_) : k / (HW_ / QS);
        const int dkx = QS >= HW_ ? 0 : (k % (HW_ / QS)) * QS;
        const long soff = ((long)dky * a.W + dkx) * gps1;  // wave-uniform
        bool inb = (NCH % NTHR == 0) || (k * NTHR + t < NCH);
        if (!INTERIOR) {
          const int gy = y0 - 1 + hy0 + dky, gx = x0 - 1 + hx0 + dkx;
          inb = inb && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        v[k] = (u32x4){0u, 0u, 0u, 0u};
        if (UP2) {
          // the input is nn.Upsample(scale_factor=2)(x): halo pixel (gy, gx) of the fine image is pixel (gy >> 1, gx >> 1) of
          // the [N][H / 2][W / 2] tensor x -- the upsampled tensor is never written (a per-load address instead of a stride)
          const int gy = y0 - 1 + hy0 + dky, gx = x0 - 1 + hx0 + dkx;
          if (inb)
            v[k] = *(const u32x4*)(a.x + (((long)n * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)) * gps + slab * (KC * 2) +
                                   ch * 16);
        } else if (inb) v[k] = *(const u32x4*)(xs + soff + voff);
      }
    };
    if (a.up2) walk(std::true_type{}, std::false_type{});
    else if (interior) walk(std::false_type{}, std::true_type{});
    else walk(std::false_type{}, std::false_type{});
  };
  unsigned char* lpw = lp;  // staging destination / fragment bases of the current halo image
  auto slab_body = [&](const int slab, const bool refill) {
    auto stage = [&](auto interior_c) {
      constexpr bool INTERIOR = decltype(interior_c)::value;
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int dky = QS >= HW_ ? k * (QS / HW_) : k / (HW_ / QS);
        const int dkx = QS >= HW_ ? 0 : (k % (HW_ / QS)) * QS;
        const bool in_range = (NCH % NTHR == 0) || (k * NTHR + t < NCH);
        u32x4 tv = v[k];
        if (M1) {
          bool inb = true;  // zero padding applies to the ACTIVATION: outside pixels stay 0, not relu(shift)
          if (!INTERIOR) {
            const int gy = y0 - 1 + hy0 + dky, gx = x0 - 1 + hx0 + dkx;
            inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          }
          if (inb && (!two_chunks || ch >= CP / 2)) tv = bnrelu_regs<bf16_t>(tv, ssc, ssh);
        }
        if (in_range) *(u32x4*)(lpw + (dky * RP + dkx) * PS) = tv;
      }
    };
    if (M1 && !interior) stage(std::false_type{});
    else stage(std::true_type{});
    if (refill) issue_halo(slab + 1);
    if (stamp && slab == 0) t_store = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stamp && slab == 0) t_k0 = __builtin_amdgcn_s_memtime();

    // ------------ k-loop: NSTEPS x (one 16-byte x fragment per m-tile, NT MFMAs on it)
    const u32x4* wslab = a.wp + ((size_t)slab * NSTEPS * ntn + nt0) * 64 + lane;
    auto frag_off = [&](const int s) {
      if (CP >= 4) {
        const int fc0 = 4 * s, tap = fc0 / CP, c0 = fc0 % CP, ky = tap / 3, kx = tap % 3;
        return (ky * RP + kx) * PS + c0 * 16;  // compile-time: the ds_read offset field
      }
      int fc = 4 * s + g;
      if (fc >= 9 * CP) fc = 0;  // K padding: the weights there are zero, any finite x will do
      const int tap = fc / CP, c = fc % CP, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      return (ky * RP + kx) * PS + c * 16;
    };
    if (STREAM_W) {
      // refills by buffer loads: one resource descriptor, the lane's 32-bit offset, a wave-uniform running offset (a
      // 1 KiB-per-wave vector-memory request is the expensive instruction of the step: no 64-bit address arithmetic on top)
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);
      unsigned wrun = (unsigned)(((size_t)slab * NSTEPS + WR) * wstep);
      // weight ring + fragments one step ahead, every step fenced (sched_barrier(0)): left to itself the scheduler bunches
      // the refills next to their uses or keeps them all live, and either spills or waits on every load
      u32x4 xf[2][MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[0][i] = *(const u32x4*)(lds + abase[i] + frag_off(0));
#pragma unroll
      for (int s = 0; s < NSTEPS; ++s) {
        // per m-tile: its NT MFMAs, then the NEXT step's fragment of that m-tile (and, after the first m-tile, the ring
        // refill): the reads issue in the shadow of the MFMAs.  With one fence per STEP all reads of step s + 1 were
        // issued between the last MFMA of step s and the first of s + 1 -- a ~100-cycle bubble per step with one wave
        // per SIMD (k-loop of Conv5.b: 350 cycles per step for 224 of MFMA)
        u32x4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wsl[s % WR][j];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_chunk<bf16_t>(wf[j], xf[s & 1][i], acc[i][j]);
          if (s + 1 < NSTEPS && !(SPCL_FAST_DBG & 2)) xf[(s + 1) & 1][i] = *(const u32x4*)(lds + abase[i] + frag_off(s + 1));
          if (i == MT - 1 && (s + WR < NSTEPS || refill) && !(SPCL_FAST_DBG & 1)) {  // (after the step's last MFMA: wf is dead, no copies)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              wsl[s % WR][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo + j * 1024, wrun, 0));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        wrun += (unsigned)wstep;  // uniform running offset: one scalar add per step instead of NSTEPS hoisted offsets
      }
    } else {
#pragma unroll
      for (int s = 0; s < NSTEPS; ++s) {
        u32x4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wf[j] = PRELOAD_W ? wall[s][j] : (PRELOAD_SLAB ? wsl[s][j] : wslab[(size_t)(s * ntn + j) * 64]);
        const int off = frag_off(s);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const u32x4 xf = *(const u32x4*)(lds + abase[i] + off);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_chunk<bf16_t>(wf[j], xf, acc[i][j]);
          // (row order, 14-row tiles: the fragment addresses are immediates now, and the scheduler -- no address register to
          // hold it back -- requests all 14 fragments of several steps at once: 126 registers + the weight fragments spilled
          // right behind their loads, three serialised round trips at the head of every workgroup, + 14 us on Conv1.b)
          if (RM && !PRELOAD_SLAB && i % 7 == 6) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
#ifndef SPCL_ACC_DBG
#define SPCL_ACC_DBG 0  /* timing experiments only (wrong results): 1 no coefficient arithmetic, 2 no block loads either */
#endif
  const int acc_part = acc_in ? t / a.CinK : 0, acc_c = t - acc_part * a.CinK;  // (the launcher: CinK * ACC_TPC == threads)
  constexpr int ACC_RPT = BN_ACC_REPLICAS / ACC_TPC > 0 ? BN_ACC_REPLICAS / ACC_TPC : 1;  // replicas per thread
  const bool acc_has = acc_part * ACC_RPT < BN_ACC_REPLICAS;  // (fewer replicas than threads per channel: the others idle)
  BnAccPart<ACC_RPT> accp;
  BnAccFwdParams accprm;
  if (acc_in && SPCL_ACC_DBG < 2) {
    accp.load(a.in_bn.acc, a.CinK, acc_c, acc_has ? ACC_RPT * acc_part : 0);
    accprm.load(a.in_bn, acc_c);
  }
  issue_halo(0, true);
  if (PRELOAD_SLAB) {
#pragma unroll
    for (int s = 0; s < WR; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j) wsl[s][j] = a.wp[((size_t)s * ntn + nt0 + j) * 64 + lane];
  }
  if (acc_in) {
    const bool writer = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
    BnAccSums sums{0, 0, 0, 0, 0};
    if (SPCL_ACC_DBG < 2 && acc_has) sums = accp.sums();
    if (SPCL_ACC_DBG < 2) sums.flag = accp.flag;
    if (ACC_TPC > 1) {  // partial sums of the other threads of a channel, through LDS: [part - 1][channel][4]
      long long* const part_l = (long long*)(coef_l + 2 * a.CinK);
      if (acc_part > 0) {
        long long* q = part_l + ((size_t)(acc_part - 1) * a.CinK + acc_c) * 4;
        *(i64x2*)q = (i64x2){sums.h1, sums.l1};
        *(i64x2*)(q + 2) = (i64x2){sums.h2, sums.l2};
      }
      __syncthreads();
      if (acc_part == 0) {
#pragma unroll
        for (int k = 1; k < ACC_TPC; ++k) {
          const long long* q = part_l + ((size_t)(k - 1) * a.CinK + acc_c) * 4;
          const i64x2 u = *(const i64x2*)q, w = *(const i64x2*)(q + 2);
          sums.h1 += u[0]; sums.l1 += u[1]; sums.h2 += w[0]; sums.l2 += w[1];
        }
      }
    }
    if (t < a.CinK) {
      float sc_, sh_;
      if (SPCL_ACC_DBG == 0) bn_acc_fwd_channel(a.in_bn, sums, accprm, t, sc_, sh_, writer);
      else { sc_ = a.in_bn.gamma[t]; sh_ = a.in_bn.beta[t]; }
      coef_l[t] = sc_;
      coef_l[a.CinK + t] = sh_;
    }
    __syncthreads();
    load_coef_lds(0);
  }
  if (MODE == 4) {
    // the image halo goes to LDS NOW, behind the activation halo (the launcher adds 1.5 KB), as the three bf16 copies
    // IM[kx][row][col] = bf16(halo[row][col + kx]) the tap fragments are read from (so that 8 columns are one aligned read;
    // 0 beyond the halo: those columns meet dz == 0).  Its loads are the oldest in flight, so this waits for them alone
    // (stored right after they were issued, before the halo requests, the wave sat out a whole HBM round trip with nothing
    // else in flight); the lane's columns 4 q .. 4 q + 3 plus two from its right-hand neighbour make the shifted quads.
    const int q = lane & 3;
    float n0 = __shfl_down(imgv[0], 1, 64), n1 = __shfl_down(imgv[1], 1, 64);
    if (q == 3) n0 = n1 = 0.f;
    const float h6[6] = {imgv[0], imgv[1], imgv[2], imgv[3], n0, n1};
    unsigned char* imb = lds + fast_lds_bytes(KC, TH) + (lane >> 2) * 32 + q * 8;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f32x2 lo = {h6[kx], h6[kx + 1]}, hi = {h6[kx + 2], h6[kx + 3]};
      uint2 w;
      w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
      w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
      *(uint2*)(imb + kx * 512) = w;
    }
  }
  int img = 0;
#pragma unroll 1
  for (int slab = 0; slab + 1 < nslab; ++slab) {
    if (slab > 0 && flip == 0) __syncthreads();  // one halo image: every wave is done reading the previous slab
    slab_body(slab, true);
    const int d = img ? -flip : flip;
    img ^= 1;
    lpw += d;
#pragma unroll
    for (int i = 0; i < MT; ++i) abase[i] += d;
  }
  if (nslab > 1 && flip == 0) __syncthreads();
  slab_body(nslab - 1, false);
  if (stamp) t_k1 = __builtin_amdgcn_s_memtime();

  // ------------ epilogue: lane holds couts 16 (nt0 + j) + 4 g .. +3 of pixel p = 16 i + r16 (tiles are always full)
  constexpr int DPY = RM ? 1 : 16 / TW, DPX = RM ? 0 : 16 % TW;  // from m-tile to m-tile: one row on / 16 pixels on
  const int rowb = a.CoutS * 2;
  // (MODE 0 with y_hi: the output channels' upper half goes to a second dense tensor -- the gradient of a channel
  // concatenation leaves as the two gradients of its parts; an n-tile belongs to one of them whole)
  // MODE 2 with y_hi: the BatchNorm-backward sums are those of the UPPER half's layer only (y2 / scale2 / shift2 / mean2 / rows2
  // all Chalf wide: the up-convolution whose activation is the second tensor of the concatenation); the lower half is the
  // skip activation, whose block sees other gradients too and reduces for itself
  const bool twoy = (MODE == 0 || MODE == 2) && a.y_hi != nullptr;
  const bool up_only = MODE == 2 && twoy;
  const int rowo = twoy ? rowb / 2 : rowb;  // bytes between two pixels of an output tensor
  unsigned char* yj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int gnt = nt0 + j, half = ntn / 2;
    unsigned char* dst = (twoy && gnt >= half) ? a.y_hi : a.y;
    yj[j] = dst + (((size_t)n * a.H + y0) * a.W + x0) * rowo + ((twoy && gnt >= half ? gnt - half : gnt) * 16 + 4 * g) * 2;
  }
  int py = RM ? 0 : r16 / TW, px = r16 - py * TW;
  int ob = (py * a.W + px) * rowo;
  const int dob = (DPY * a.W + DPX) * rowo, wrapo = (a.W - TW) * rowo;
  f32x4 ssum[NT], ssq[NT];
  f32x4 sc2[NT], sh2[NT], mu2[NT];  // MODE 2: BN coefficients of this lane's 4 channels per n-tile
  const unsigned char* y2b = nullptr;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ssq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (M2 || MODE == 3) {
      const int gnt2 = nt0 + j, half2 = ntn / 2;
      const int cb = (up_only ? (gnt2 >= half2 ? gnt2 - half2 : 0) : gnt2) * 16 + 4 * g;
      sc2[j] = *(const f32x4*)(a.scale2 + cb);
      sh2[j] = *(const f32x4*)(a.shift2 + cb);
      mu2[j] = *(const f32x4*)(a.mean2 + cb);
    }
  }
  if (M2) y2b = a.y2 + (((size_t)n * a.H + y0) * a.W + x0) * rowb + (nt0 * 16 + 4 * g) * 2;
  const unsigned char* y2j[NT];  // per n-tile base of y2 (up_only: the half-width tensor, upper n-tiles only)
  bool st2[NT];                  // does this n-tile take part in the statistics?
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int gnt2 = nt0 + j, half2 = ntn / 2;
    st2[j] = !up_only || gnt2 >= half2;
    y2j[j] = !M2 ? nullptr
                 : (up_only ? a.y2 + (((size_t)n * a.H + y0) * a.W + x0) * rowo + ((st2[j] ? gnt2 - half2 : 0) * 16 + 4 * g) * 2
                            : y2b + j * 32);
  }
  // MODE 4 LDS map: [0, 7168) DZ[co 16][row 14][col 16] bf16 -- dz of the tile, transposed, in the space of the activation
  // halo (no longer needed; one wave per workgroup, its own k-loop reads are behind it); columns 14 / 15 zero.
  // [halo bytes, + 1536) IM[kx 3][row 16][col 16] bf16, written in the prologue.
  constexpr int M4_DZ = 0, M4_IM = fast_lds_bytes(KC, TH);
  static_assert(MODE != 4 || fast_lds_bytes(KC, TH) >= 7168, "MODE 4 scratch does not fit the halo image");
  if (MODE == 4) {
    // the two pad columns of every (channel, row): 224 dwords, 3.5 per lane
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = k * 64 + lane;
      if (e < 16 * TH) *(uint32_t*)(lds + M4_DZ + e * 32 + 28) = 0u;
    }
    __syncthreads();  // (one wave: an ordering point for the compiler only)
  }
  const bool shifted = (oy | ox) != 0;  // wave-uniform
  int pyc = py;                         // the pixel's row inside the tile, walked with px
  // MODE 2 / 3: every y2 value the statistics need is requested BEFORE the first output store (the k-loop's fragment
  // registers are free now).  Loads and stores retire in order: a load issued after a store is only usable once that
  // store is acknowledged, and one request - wait - compute round per (m-tile, n-tile) is a memory latency each (the
  // MODE 3 epilogue of the 128-workgroup Conv5.a dgrad took 24k cycles of its 54k that way).
  constexpr int NWIN = MODE == 3 ? 4 : 1;
  // ... in chunks of GM m-tiles, the next chunk's requests ahead of this chunk's stores, two chunks in flight within a
  // register budget that the kernel's occupancy target leaves
  constexpr bool YPRE = (M2 || MODE == 3) && KC >= SPCL_FAST_YPRE_MINKC;  // (128-register kernels: no room)
  constexpr int YBUD = MODE == 3 ? 64 : 56, YR1 = NT * NWIN * 2;
  constexpr int GM = YBUD / (2 * YR1) > 0 ? YBUD / (2 * YR1) : 1;
  uint2 ypre[YPRE ? MT : 1][NT][NWIN];
  int lpx = px, lpyc = pyc, lob = ob;  // walker of the requests
  auto request_chunk = [&](const int c) {
#pragma unroll
    for (int ii = 0; ii < GM; ++ii) {
      const int i = c * GM + ii;
      if (i < MT) {
        const bool ok = RM || (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);  // (row order: the unused columns are off already)
        if (ok) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (M2) {
              ypre[i][j][0] = *(const uint2*)(y2j[j] + lob);  // (a lower n-tile of the up-only form reads a valid dummy)
            } else {
              const unsigned char* wb = a.y2 +
                                        ((((size_t)n * a.H2 + 2 * (y0 + lpyc)) * a.W2 + 2 * (x0 + lpx)) * rowb) +
                                        ((nt0 + j) * 16 + 4 * g) * 2;
#pragma unroll
              for (int k = 0; k < 4; ++k)
                ypre[i][j][k] = *(const uint2*)(wb + ((size_t)(k >> 1) * a.W2 + (k & 1)) * rowb);
            }
          }
        }
        lpx += DPX;
        lpyc += DPY;
        lob += dob;
        if (!RM && lpx >= TW) {
          lpx -= TW;
          lpyc += 1;
          lob += wrapo;
        }
      }
    }
  };
  // (row order: pixel columns 14, 15 of every m-tile are nobody's -- the same lanes throughout, switched off once for the whole
  // epilogue: their requests, stores and statistics never happen)
  if (!RM || r16 < TW) {
  if (YPRE) {
    request_chunk(0);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist every later request up here too)
  }
  uint2 pk_prev[NT];
  int ob_prev = 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const bool ok = RM || (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);  // compile-time true except in the last linear m-tile
    if (YPRE && i % GM == 0 && (i / GM + 1) * GM < MT) {
      request_chunk(i / GM + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ok) {
      const float keep = (!shifted || (pyc >= oy && px >= ox)) ? 1.f : 0.f;  // 0: the neighbour tile counts this pixel
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // (MODE 4: nobody downstream may want g itself -- the image block's first conv needs only the sums below)
        if (MODE != 4 || a.y != nullptr) {
          // complete m-tiles leave in PAIRS: lanes g and g ^ 1 hold the two 8-byte halves of a pixel's 16-byte channel run,
          // one v_permlane16_swap per dword gives the even lane group its pixel of m-tile i - 1 whole and the odd one its
          // pixel of m-tile i -- half the store instructions for the same bytes (conv3x3_image_kernel below)
          // (not in the one-wave KC = 64 kernels: at their 128-register budget the parked halves spill)
          constexpr bool PAIRED = SPCL_FAST_WIDE_STORES && MODE != 4 && !(KC == 64 && NW == 1);
          const bool first = PAIRED && i % 2 == 0 && i + 1 < MT && (RM || 16 * (i + 1) + 15 < NPIX);
          const bool second = PAIRED && i % 2 == 1 && (RM || 16 * i + 15 < NPIX);
          if (first || second) {
            const f32x2 lo = {acc[i][j][0], acc[i][j][1]}, hi = {acc[i][j][2], acc[i][j][3]};
            uint2 pkc;
            pkc.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
            pkc.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
            if (first) {
              pk_prev[j] = pkc;
              ob_prev = ob;
            } else {
              const auto rx = __builtin_amdgcn_permlane16_swap(pk_prev[j].x, pkc.x, false, false);
              const auto ry = __builtin_amdgcn_permlane16_swap(pk_prev[j].y, pkc.y, false, false);
              const u32x4 v = {rx[0], ry[0], rx[1], ry[1]};
              *(u32x4*)(yj[j] + ((g & 1) ? ob - 8 : ob_prev)) = v;
            }
          } else {
            store4_fast<bf16_t>(yj[j] + ob, acc[i][j]);
          }
        }
        if (M2) {
          // dz = g [relu(bn(y2)) > 0] with g as STORED (bf16): sum dz and sum dz (y2 - mean) of the lane's 4 channels
          const uint2 yr = YPRE ? ypre[i][j][0] : *(const uint2*)(y2j[j] + ob);
          const f32x2 glo = {acc[i][j][0], acc[i][j][1]}, ghi = {acc[i][j][2], acc[i][j][3]};
          const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
          const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
          const float yv[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u),
                               __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
          const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                               __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dz = (st2[j] && fmaf(sc2[j][r], yv[r], sh2[j][r]) > 0.f) ? gv[r] * keep : 0.f;
            ssum[j][r] += dz;
            ssq[j][r] = fmaf(dz, yv[r] - mu2[j][r], ssq[j][r]);
            // MODE 4: dz (a bf16 value: exact) goes to its [co][row][col] place in LDS right away -- keeping the
            // accumulators alive for a later pass cost eleven spilled registers on the epilogue's critical path
            if (MODE == 4) *(bf16_t*)(lds + M4_DZ + (4 * g + r) * 448 + pyc * 32 + px * 2) = f32_to_bf16(dz);
          }
        } else if (MODE == 3) {
          // the 2x2 window of y2 under this pooled pixel: the gradient goes to the first maximum of relu(bn(y2)) in scan
          // order if that maximum is positive (bn.hip bnrelu_bwd_pool_kernel<T, false>, the pass this epilogue replaces)
          const f32x2 glo = {acc[i][j][0], acc[i][j][1]}, ghi = {acc[i][j][2], acc[i][j][3]};
          const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
          const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
          const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                               __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
          uint2 yr[4];
          if (YPRE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) yr[k] = ypre[i][j][k];
          } else {
            const unsigned char* wb = a.y2 + ((((size_t)n * a.H2 + 2 * (y0 + pyc)) * a.W2 + 2 * (x0 + px)) * rowb) +
                                      ((nt0 + j) * 16 + 4 * g) * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) yr[k] = *(const uint2*)(wb + ((size_t)(k >> 1) * a.W2 + (k & 1)) * rowb);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float zb = -1.f, yb = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t wd = r < 2 ? yr[k].x : yr[k].y;
              const float yv = (r & 1) ? __uint_as_float(wd & 0xffff0000u) : __uint_as_float(wd << 16);
              const float z = fmaf(sc2[j][r], yv, sh2[j][r]);
              if (k == 0 || z > zb) {
                zb = z;
                yb = yv;
              }
            }
            const float dz = zb > 0.f ? gv[r] * keep : 0.f;
            ssum[j][r] += dz;
            ssq[j][r] = fmaf(dz, yb - mu2[j][r], ssq[j][r]);
          }
        } else {
          const f32x4 av = acc[i][j] * keep;
          ssum[j] += av;
          ssq[j] += av * acc[i][j];
        }
      }
    }
    px += DPX;
    pyc += DPY;
    ob += dob;
    if (!RM && px >= TW) {
      px -= TW;
      pyc += 1;
      ob += wrapo;
    }
  }
  }
  constexpr int RS = MODE == 4 ? 11 : 2;  // rows per tile of rows2
  // (the replica of a fixed-point accumulator block this workgroup adds to: its index in dispatch order mod 8 = its XCD)
  const int replica = (int)((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & (BN_ACC_REPLICAS - 1));
  if ((MODE == 2 || MODE == 3) && a.rows2_acc != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 s1, s2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] = row16_sum(ssum[j][r]);
        s2[r] = row16_sum(ssq[j][r]);
      }
      bn_acc_add_row16(a.rows2_acc, a.CoutS, replica, (nt0 + j) * 16 + 4 * g, r16, s1, s2);
    }
  } else if (M2 || MODE == 3) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(ssum[j][r]), s2 = row16_sum(ssq[j][r]);
        o[r] = r16 == 0 ? s1 : s2;
      }
      if (up_only) {
        if (r16 < 2 && st2[j])
          *(f32x4*)(a.rows2 + ((size_t)tile * RS + r16) * (a.CoutS / 2) + (nt0 + j - ntn / 2) * 16 + 4 * g) = o;
      } else if (r16 < 2) *(f32x4*)(a.rows2 + ((size_t)tile * RS + r16) * a.CoutS + (nt0 + j) * 16 + 4 * g) = o;
    }
    if (MODE == 4) {
      // S[co][tap] = sum over the tile's pixels of dz[p][co] img[p + tap] ON THE MATRIX PIPE: D[co][tap] += A[co][k] B[k][tap]
      // with k = (tile row pair, 16 columns) -- 7 k-steps of 32.  dz goes through LDS once, transposed to [co][row][col]
      // (4 two-byte stores per pixel), and comes back as 16-byte A fragments; the B fragments are 16-byte reads of the
      // shifted image copies.  D's layout (lane: tap r16, channels 4 g ..) is the row store below: no cross-lane sums.
      // (A first version kept 36 f32 accumulators per lane and reduced them with 144 DPP row sums: +39 us on this launch.)
      static_assert(MODE != 4 || NT == 1, "image tap sums: one n-tile");
      __syncthreads();
      const int tap = r16 < 9 ? r16 : 0, tky = tap / 3, tkx = tap - 3 * tky;
      const unsigned char* pa = lds + M4_DZ + r16 * 448 + (g >> 1) * 32 + (g & 1) * 16;
      const unsigned char* pb = lds + M4_IM + tkx * 512 + ((g >> 1) + tky) * 32 + (g & 1) * 16;
      f32x4 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < TH / 2; ++ks)
        D = mfma_chunk<bf16_t>(*(const u32x4*)(pa + ks * 64), *(const u32x4*)(pb + ks * 64), D);
      if (r16 < 9) *(f32x4*)(a.rows2 + ((size_t)tile * RS + 2 + r16) * a.CoutS + nt0 * 16 + 4 * g) = D;
    }
  } else if ((MODE == 0 || M1) && a.stats_acc != nullptr) {
    // (forming the sums and issuing the adds BEFORE the output stores was measured too: no gain -- what the adds cost is their
    // serialisation at the memory side, ~18 ns per add on one address, not their place in the epilogue)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 s1, s2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] = row16_sum(ssum[j][r]);
        s2[r] = row16_sum(ssq[j][r]);
      }
      bn_acc_add_row16(a.stats_acc, a.CoutS, replica, (nt0 + j) * 16 + 4 * g, r16, s1, s2);
    }
  } else if (a.stats != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      write_tile_stats(a.stats, tile, a.CoutS, (nt0 + j) * 16 + 4 * g, r16, (float)((TH - oy) * (TW - ox)), ssum[j],
                       ssq[j]);
  }
  if (stamp) {
    __builtin_amdgcn_s_waitcnt(0);  // (the epilogue's stores issued; their acknowledgements are not awaited by the kernel)
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
    o[0] = t_store - t_begin; o[1] = t_k0 - t_store; o[2] = t_k1 - t_k0; o[3] = t_end - t_k1;
  }
}

// First layer (unet.py:123: input_dim == 1 -> 16): the f32 single-channel image convolved with 16 filters, K = 9.
// One v_mfma_f32_16x16x16_bf16 per 16 pixels with k = (ky, kx | pad): k-group g = ky holds (kx = 0, 1, 2, pad).  The
// halo tile sits in LDS as bf16 PAIRS (x[q], x[q+1]) per pixel q, so that a lane's whole B fragment
// (x[q], x[q+1], x[q+2], x[q+3]: three taps and a finite pad against a zero weight) is one ds_read2_b32.
// Weights come from the ordinary packed forward layout (kind 0, CinK = 16): three 2-byte loads per lane.
// The kernel is a pure output stream (32 bytes per pixel out, 4 in).
// EVEN: the image size is a multiple of the tile (no shifted last tiles: the `keep` factor of the statistics folds away)
// ACORR: the tile's share of the image's 9 x 9 autocorrelation (image_acorr.hpp: R[t'][t] = sum_p img0[p + t'] img0[p + t],
// and the nine sums of img0[p + t]) from the bf16 halo pairs that sit in LDS anyway -- 7 k-steps of 32 pixels (two tile rows
// of 16 columns, the last two masked), A = B = the patch matrix (one ds_read2_b32 pair per operand), a second accumulator
// against a ones operand for the image sums: 14 MFMAs and ~60 instructions per tile instead of a 13 MB pass of its own
// (image_autocorr_body in the weight-pack launch: ~15 us of the step's first launch).  EVEN sizes only.
template <int TH, bool EVEN, bool ACORR = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void conv3x3_image_kernel(FastArgs a) {
  constexpr int TW = 14, HW_ = 16, LW = HW_ + 2;  // LDS row: 16 halo pixels + 2 so that pair q+2 of the last tap exists
  constexpr bool RM = SPCL_FAST_ROWMAP != 0;  // m-tile i = tile row i, pixel column r16 (see conv3x3_fast_kernel)
  constexpr int NPIX = TH * TW, MT = RM ? TH : (NPIX + 15) / 16, NHALO = (TH + 2) * HW_;
  __shared__ uint32_t pairs[(TH + 2) * LW];
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  int tx = blockIdx.x, ty = blockIdx.y;
  const int n = blockIdx.z;
  {  // (an XCD's workgroups take neighbouring tiles: the 128-byte lines two tiles of a row share are completed in ONE L2)
    const int T = a.tilesX * a.tilesY;
    if (a.xcd_remap && (T & 7) == 0) {
      const int L = ty * a.tilesX + tx;
      const int L2 = (L & 7) * (T >> 3) + (L >> 3);
      ty = L2 / a.tilesX;
      tx = L2 - ty * a.tilesX;
    }
  }
  const int y0 = min(ty * TH, a.H - TH), x0 = min(tx * TW, a.W - TW);  // shifted last tiles, as in the kernel above
  const int oy = ty * TH - y0, ox = tx * TW - x0;
  const int tile = (n * a.tilesY + ty) * a.tilesX + tx;

  // A fragment: row = cout r16, k-group g = ky: W[cout][0][ky][0..2], 0.  Packed index of (cout, ci = 0, tap):
  // chunk fc = 2 tap -> step tap >> 1, k-group 2 (tap & 1), element 0
  s16x4 wfrag = {0, 0, 0, 0};
  if (g < 3) {
    const bf16_t* wp = (const bf16_t*)a.wp;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tap = 3 * g + kx;
      wfrag[kx] = (short)wp[(size_t)(((tap >> 1) * 64) + (2 * (tap & 1)) * 16 + r16) * 8];
    }
  }

  // stage: lane -> halo pixels q = lane + 64 k (four halo rows of 16 per step: a lane keeps its column); pair =
  // (x[q], x[q+1]) with zero outside the image -- ONE clamped load per pixel, the right-hand neighbour comes from the next
  // lane (the second load and its bounds checks were a third of the staging's instructions; the kernel is issue-bound)
  const float* img = (const float*)a.x + (size_t)n * a.H * a.W;
  {
    const int hx = r16, gx = x0 - 1 + hx;
    const bool colok = gx >= 0 && gx < a.W;
    const int gxc = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
#pragma unroll
    for (int k = 0; k < (NHALO + 63) / 64; ++k) {
      const int hy = g + 4 * k;
      const int gy = y0 - 1 + hy;
      const int gyc = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
      const float raw = img[(size_t)gyc * a.W + gxc];
      const float v0 = (colok && gy >= 0 && gy < a.H) ? raw : 0.f;
      float v1 = __shfl_down(v0, 1, 64);
      if (hx == HW_ - 1) v1 = 0.f;
      const f32x2 pv = {v0, v1};
      if (NHALO % 64 == 0 || hy < TH + 2) pairs[hy * LW + hx] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pv, bf16x2v));
    }
  }
  if (lane < 2 * (TH + 2)) pairs[(lane >> 1) * LW + HW_ + (lane & 1)] = 0u;  // the two pad pairs of each row
  __syncthreads();

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = 16 * i + r16;
    if (p >= NPIX) p = 0;
    const int py = RM ? i : p / TW, px = RM ? r16 : p - py * TW;  // (row order: columns 14, 15 read the pad pairs; unused)
    const int gg = g < 3 ? g : 0;  // k-group 3 is all padding (zero weights): read something valid
    const uint32_t* src = pairs + (py + gg) * LW + px;
    const uint32_t lo = src[0], hi = src[2];
    const uint2 xv = {lo, hi};
    acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfrag, __builtin_bit_cast(s16x4, xv),
                                                       (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }

  constexpr int DPY = RM ? 1 : 16 / TW, DPX = RM ? 0 : 16 % TW;
  const int rowb = a.CoutS * 2;
  unsigned char* yb = a.y + (((size_t)n * a.H + y0) * a.W + x0) * rowb + 4 * g * 2;
  int py = RM ? 0 : r16 / TW, px = r16 - py * TW;
  int ob = (py * a.W + px) * rowb;
  const int dob = (DPY * a.W + DPX) * rowb, wrapo = (a.W - TW) * rowb;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  const bool shifted = !EVEN && (oy | ox) != 0;
  int pyc = py;
  // Two m-tiles per 16-byte store: lanes g and g ^ 1 hold the two 8-byte halves of a pixel's 16-byte channel run, so one
  // v_permlane16_swap per dword gives the even lane group both halves of its pixel in m-tile i and the odd one both halves of
  // its pixel in m-tile i + 1 -- half the store instructions for the same bytes (the kernel is a 103 MB output stream).
  uint2 pk[MT];
  int obs[MT];
  if (!RM || r16 < TW) {  // (row order: the two unused pixel columns are off for the whole epilogue)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const bool ok = RM || (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);
    obs[i] = ob;
    {
      const f32x2 lo = {acc[i][0], acc[i][1]}, hi = {acc[i][2], acc[i][3]};
      pk[i].x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
      pk[i].y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
    }
    if (ok) {
      if (EVEN) {
        ssum += acc[i];
        ssq += acc[i] * acc[i];
      } else {
        const float keep = (!shifted || (pyc >= oy && px >= ox)) ? 1.f : 0.f;
        const f32x4 av = acc[i] * keep;
        ssum += av;
        ssq += av * acc[i];
      }
    }
    px += DPX;
    pyc += DPY;
    ob += dob;
    if (!RM && px >= TW) {
      px -= TW;
      pyc += 1;
      ob += wrapo;
    }
  }
  constexpr int NPAIR = SPCL_FAST_WIDE_STORES ? (RM ? MT / 2 : (NPIX / 16) / 2) : 0;  // pairs of COMPLETE m-tiles
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    const int i = 2 * p;
    const auto rx = __builtin_amdgcn_permlane16_swap(pk[i].x, pk[i + 1].x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(pk[i].y, pk[i + 1].y, false, false);
    const u32x4 v = {rx[0], ry[0], rx[1], ry[1]};
    *(u32x4*)(yb + ((g & 1) ? obs[i + 1] - 8 : obs[i])) = v;
  }
#pragma unroll
  for (int i = 2 * NPAIR; i < MT; ++i) {
    const bool ok = RM || (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);
    if (ok) *(uint2*)(yb + obs[i]) = pk[i];
  }
  }
  if (a.stats != nullptr)
    write_tile_stats(a.stats, tile, a.CoutS, 4 * g, r16, (float)((TH - oy) * (TW - ox)), ssum, ssq);
  if (ACORR) {
    static_assert(!ACORR || (EVEN && TH % 2 == 0), "autocorrelation rows: whole 14 x 14 tiles");
    __shared__ float dsum[2][16][16];
    const int tap = r16 < 9 ? r16 : 0, ky = tap / 3, kx = tap - 3 * ky;
    // lane (tap r16, k-group g): pixels 8 (g & 1) .. + 7 of tile row 2 ks + (g >> 1), shifted by the tap = halo pairs
    // s, s + 2, s + 4, s + 6 with s = (row + ky) LW + 8 (g & 1) + kx; pixel columns 14, 15 (the last pair of the upper
    // k-groups) are not the tile's: zero in A, so their products vanish whatever B holds
    const uint32_t* src = pairs + ((g >> 1) + ky) * LW + 8 * (g & 1) + kx;
    const bool upper = (g & 1) != 0;
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    f32x4 D = {0.f, 0.f, 0.f, 0.f}, S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < TH / 2; ++ks) {
      const uint32_t* q = src + ks * 2 * LW;
      const u32x4 fb = {q[0], q[2], q[4], q[6]};
      u32x4 fa = fb;
      fa[3] = upper ? 0u : fb[3];
      D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), D, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, ones), S, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // D[row t' = 4 g + r][column t = r16]; every column of S holds the row's image sum
      dsum[0][4 * g + r][r16] = D[r];
      dsum[1][4 * g + r][r16] = S[r];
    }
    __syncthreads();  // (one wave: an ordering point)
    const int k = lane;
    float v = 0.f;
    if (k < 45) {  // upper triangle in the order of bn.hip acorr_index: row ra = number of row starts <= k
      int ra = 0;
#pragma unroll
      for (int i = 1; i < 9; ++i) ra += k >= i * 9 - i * (i - 1) / 2 ? 1 : 0;
      const int cb = ra + (k - (ra * 9 - ra * (ra - 1) / 2));
      v = dsum[0][ra][cb];
    } else if (k < 54) {
      v = dsum[1][k - 45][0];
    }
    a.acorr_rows[(size_t)tile * 64 + k] = v;
  }
}

template <int KC, int TH, int NT, int NW>
static void launch_fast(const FastArgs& a, int mode, hipStream_t st) {
  static const int env_pipe = lab_env("SPCL_CONV_FAST_PIPE", 1);
  const bool pipe = env_pipe && KC == 64 && a.CinK > KC;  // more than one slab: two halo images
  FastArgs b = a;
  b.lds_flip = pipe ? fast_lds_bytes(KC, TH) : 0;
  const size_t lds = fast_lds_bytes(KC, TH) * (pipe ? 2 : 1) + (a.img2 != nullptr ? 1536 : 0) +
                     (a.in_bn.acc != nullptr ? (size_t)a.CinK * 8 + (size_t)64 * NW * 32 : 0);  // (+ MODE 5 .. 7: the derived
                                                                                          // scale / shift and the partial sums)
  dim3 grid(a.tilesX, a.tilesY, a.N * a.gy), block(64 * NW);
#if SPCL_FAST_DBG & 4
  if (KC == 64 && a.CinK >= 128 && (long)a.tilesX * a.tilesY * a.N * a.gy <= 256) grid.z *= 2;
#endif
  static const bool env_stamps = SPCL_FAST_STAMPS_BUILD && lab_flag("SPCL_FAST_STAMPS");
  const size_t nwg = (size_t)grid.x * grid.y * grid.z;
  b.stamps = nullptr;
  if (env_stamps) {  // debug only (synchronises)
    (void)hipMalloc(&b.stamps, nwg * 4 * sizeof(unsigned long long));
    (void)hipMemset(b.stamps, 0, nwg * 4 * sizeof(unsigned long long));
  }
  if (a.img2 != nullptr) {
    if constexpr (KC == 16 && TH == 14 && NT == 1 && NW == 1)
      SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 4, NW>), grid, block, lds, st, b);
  } else if (mode == 1 && a.in_bn.acc != nullptr) {
    if constexpr (KC == 64) {  // (threads per input channel; launch_conv_fast has checked that it is 4, 2 or 1)
      const int tpc = 64 * NW / a.CinK;
      if (tpc == 4) SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 5, NW>), grid, block, lds, st, b);
      else if (tpc == 2) SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 6, NW>), grid, block, lds, st, b);
      else SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 7, NW>), grid, block, lds, st, b);
    }
  } else if (mode == 1) SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 1, NW>), grid, block, lds, st, b);
  else if (a.rows2 != nullptr && a.H2 > 0) SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 3, NW>), grid, block, lds, st, b);
  else if (a.rows2 != nullptr) SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 2, NW>), grid, block, lds, st, b);
  else SPCL_LAUNCH((conv3x3_fast_kernel<KC, TH, NT, 0, NW>), grid, block, lds, st, b);
  if (b.stamps != nullptr) {
    std::vector<unsigned long long> h(nwg * 4);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), b.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(b.stamps);
    double s4[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < nwg; ++i)
      for (int k = 0; k < 4; ++k) s4[k] += (double)h[i * 4 + k];
    fprintf(stderr, "[conv_fast stamps] <%d,%d,%d,m%d,%d> %dx%d CinK %d CoutS %d wgs %zu | s_memtime ticks per wg: setup + first "
            "halo + LDS store %.0f, barrier %.0f, k-loops (all slabs) %.0f, epilogue %.0f\n", KC, TH, NT,
            mode == 1 ? 1 : (a.rows2 ? (a.H2 > 0 ? 3 : 2) : 0), NW, a.H, a.W, a.CinK, a.CoutS, nwg, s4[0] / nwg, s4[1] / nwg,
            s4[2] / nwg, s4[3] / nwg);
  }
}

bool launch_conv_fast(const ConvArgs& c, int th, hipStream_t st, bool dry) {
  if (c.H < th || c.W < 14) return false;
  if (c.in_mode == 2) {
    if (c.CinS != 1 || c.CoutS != 16 || th != 14) return false;
    FastArgs a;
    if (c.x2 != nullptr) return false;
    if (c.y_hi != nullptr) return false;
    if (c.x_up2) return false;
    a.x = (const unsigned char*)c.x; a.x2 = nullptr; a.y = (unsigned char*)c.y; a.y_hi = nullptr; a.wp = (const u32x4*)c.wp; a.stats = c.stats;
    a.up2 = 0;
    a.in_scale = a.in_shift = nullptr;
    a.y2 = nullptr; a.scale2 = a.shift2 = a.mean2 = nullptr; a.rows2 = nullptr; a.H2 = a.W2 = 0; a.img2 = nullptr;
    a.N = c.N; a.H = c.H; a.W = c.W; a.CinK = 16; a.CoutS = 16;
    static const int env_img_remap = lab_env("SPCL_IMAGE_XCD_REMAP", 1);
    a.tilesX = cdiv(c.W, 14); a.tilesY = cdiv(c.H, 14); a.gy = 1; a.xcd_remap = env_img_remap; a.lds_flip = 0; a.stamps = nullptr;
    a.acorr_rows = c.acorr_rows;
    a.stats_acc = a.rows2_acc = nullptr;
    a.in_bn = BnAccFwd{};
    if (c.stats_acc != nullptr || c.rows2_acc != nullptr || c.in_bn != nullptr) return false;  // (16 384 tiles: rows + reduction)
    if (c.acorr_rows != nullptr && !(c.H % 14 == 0 && c.W % 14 == 0)) return false;  // (whole tiles only)
    if (!dry) {
      if (c.acorr_rows != nullptr) SPCL_LAUNCH((conv3x3_image_kernel<14, true, true>), dim3(a.tilesX, a.tilesY, a.N), dim3(64), 0, st, a);
      else if (c.H % 14 == 0 && c.W % 14 == 0) SPCL_LAUNCH((conv3x3_image_kernel<14, true>), dim3(a.tilesX, a.tilesY, a.N), dim3(64), 0, st, a);
      else SPCL_LAUNCH((conv3x3_image_kernel<14, false>), dim3(a.tilesX, a.tilesY, a.N), dim3(64), 0, st, a);
    }
    return true;
  }
  if (c.CinS != c.CinK) return false;
  const bool wants_acc = c.stats_acc != nullptr || c.rows2_acc != nullptr || c.in_bn != nullptr;
  const int ntn = c.CoutS / 16, KC = conv_kc(c.CinK);
  static const int env_nt1 = lab_env("SPCL_CONV_FAST_NT1", 0);
  int NT = ntn >= 2 ? 2 : 1;
  if (env_nt1 == 1 && KC == 64 && ntn == 2) NT = 1;
  // 64 -> 32 channels with the pooled BatchNorm-backward sums in the epilogue (Conv3.a's dgrad): as two one-n-tile waves,
  // each with half of the epilogue's scattered loads -- 29 us against 23 + 15 for the plain one-wave kernel and the
  // separate reduction pass (whole step 1.134 -> 1.126 ms, same box; SPCL_CONV_FAST_NT1=2 switches it off)
  if (env_nt1 != 2 && KC == 64 && ntn == 2 && c.rows2 != nullptr && c.H2 > 0) NT = 1;
  // ... and every other 64 -> 32 layer too (the decoder's 112^2 level: up-convolution and cat(32, 32) -> 32): the one-wave
  // workgroup's halo image is 23 KB, six WAVES per CU; two waves per image double that (fine-tune step 2.337 -> 2.305 ms,
  // same box, three rounds)
  if (env_nt1 != 2 && KC == 64 && ntn == 2) NT = 1;
  // 64 -> 64 channels: four one-n-tile waves when the loader also applies BN+ReLU (more lanes for the transform:
  // Conv3.b forward 38 -> 30 us) and on the small images; two two-n-tile waves for the plain 56^2 dgrad
  if (KC == 64 && ntn == 4 && env_nt1 != 3 && (c.in_mode == 1 || c.H <= 28)) NT = 1;
  // ... and with more than one slab of input channels (128 -> 64: the decoder's 56^2 level), where the cross-slab pipeline
  // keeps TWO halo images (46 KB): three two-wave workgroups per CU otherwise
  static const int env_wide = lab_env("SPCL_CONV_FAST_NT1_WIDE", 1);
  if (env_wide && KC == 64 && ntn == 4 && c.CinK > 64) NT = 1;
  // ... and, re-measured in round 4, the 56^2 dgrad as well (Conv3.b's, with the BatchNorm-backward sums in its epilogue:
  // pre-train step 1.160 -> 1.155 ms, same box, three rounds; fine-tune unchanged): four one-n-tile waves for every 64 -> 64
  static const int env_all4 = lab_env("SPCL_CONV_FAST_NT1_64", 1);
  if (env_all4 && KC == 64 && ntn == 4) NT = 1;
  // few tiles (14^2 images): with two n-tiles per wave a 128-channel output is ONE workgroup per tile -- 128 workgroups for
  // Conv5.a's dgrad at N = 64, half the CUs idle; one n-tile per wave doubles the workgroups
  static const int env_fill = lab_env("SPCL_CONV_FAST_FILL", 1);
  // (round 5, same box: 512 takes Conv5 at N = 64 -- 128 tiles x 2 cout halves = 256 workgroups, one wave per SIMD -- to 512
  // one-n-tile workgroups: Conv5.a / .b forward 14.2 / 23.7 -> 13.8 / 23.4 us, Conv5.b's dgrad 23.0 -> 21.8; 1 100, which
  // takes Conv4 too, loses 4 us there)
  static const int env_fill_max = lab_env("SPCL_CONV_FAST_FILL_MAX", 512);
  if (env_fill && KC == 64 && NT == 2 && ntn >= 8 && ntn % 4 == 0 &&
      (long)c.N * cdiv(c.W, 14) * cdiv(c.H, th) * cdiv(ntn, 8) < env_fill_max)
    NT = 1;
  // experiment: the two n-tiles of a 16 / 32-channel-slab layer as two one-n-tile waves sharing the halo image
  // (1 = all of them: + 4 us per step when measured, and again after the row order.  Per launch then: Conv2.a / Conv2.b forward
  // +- 0.7 us, but Conv2.b's dgrad -- 32 -> 32 with the BatchNorm-backward sums in its epilogue, 202 registers as one wave --
  // 42.9 -> 38.6 us as two waves of 140: 2 = that form only, the default)
  static const int env_narrow = lab_env("SPCL_CONV_FAST_NARROW_NT1", 2);
  if (env_narrow == 1 && KC < 64 && ntn == 2 && th == 7) NT = 1;
  if (env_narrow == 2 && KC == 32 && ntn == 2 && th == 7 && c.rows2 != nullptr && c.H2 == 0 && c.img2 == nullptr) NT = 1;
  int nw = ntn / NT;
  if (nw > 4) nw = 4;
  if (ntn % (NT * nw) != 0) return false;
  FastArgs a;
  a.x = (const unsigned char*)c.x; a.y = (unsigned char*)c.y; a.wp = (const u32x4*)c.wp; a.stats = c.stats;
  a.x2 = (const unsigned char*)c.x2;
  a.y_hi = (unsigned char*)c.y_hi;
  a.up2 = c.x_up2 ? 1 : 0;
  // (the upsampled input: plain convolutions of one tensor, even image sizes, 32-bit byte offsets)
  if (c.x_up2 && !(c.in_mode == 0 && c.x2 == nullptr && c.rows2 == nullptr && c.img2 == nullptr && c.H % 2 == 0 &&
                   c.W % 2 == 0 && (double)c.N * c.H * c.W / 4.0 * c.CinK * 2.0 < 2147483648.0))
    return false;
  // (two output tensors: the plain dgrad, or the one with the UPPER half's BatchNorm-backward sums -- MODE 2, same-resolution y2)
  if (c.y_hi != nullptr && !(c.in_mode == 0 && (c.rows2 == nullptr || c.H2 == 0) && c.img2 == nullptr && c.stats == nullptr &&
                             ntn % 2 == 0))
    return false;
  // two input tensors: one slab whose 16-byte chunks split evenly between them (32 = 16 + 16, 64 = 32 + 32 channels)
  // ... or several slabs, half of them in each tensor (128 = 64 + 64, 256 = 128 + 128)
  if (c.x2 != nullptr && !(((c.CinK == KC && KC >= 32) || (KC == 64 && (c.CinK / 64) % 2 == 0)) && c.rows2 == nullptr &&
                           c.img2 == nullptr))
    return false;
  if (c.x2 != nullptr && c.in_mode == 1 && c.CinK != KC) return false;  // (the half-transform exists for one slab only)
  a.in_scale = c.in_scale; a.in_shift = c.in_shift;
  a.y2 = (const unsigned char*)c.y2; a.scale2 = c.scale2; a.shift2 = c.shift2; a.mean2 = c.mean2; a.rows2 = c.rows2;
  a.H2 = c.H2; a.W2 = c.W2;
  a.img2 = c.img2;
  a.acorr_rows = nullptr;
  // fixed-point accumulator blocks (bn_acc.hpp): plain one-tensor-output launches with few enough tiles (same-address adds
  // serialise at the memory side: ~18 ns each, tiles / 8 of them per address); the prologue form needs 4, 2 or 1 threads per
  // input channel
  a.stats_acc = c.stats_acc;
  a.rows2_acc = c.rows2_acc;
  a.in_bn = c.in_bn != nullptr ? *c.in_bn : BnAccFwd{};
  if (wants_acc) {
    const long tiles = (long)c.N * cdiv(c.W, 14) * cdiv(c.H, th);
    if (c.y_hi != nullptr || c.img2 != nullptr) return false;
    if (c.stats_acc != nullptr && (c.rows2 != nullptr || tiles > BN_ACC_MAX_TILES_FWD)) return false;
    if (c.rows2_acc != nullptr && (c.rows2 == nullptr || tiles > BN_ACC_MAX_TILES_BWD)) return false;  // (rows2: the mode marker)
    if (c.in_bn != nullptr && !(c.in_mode == 1 && KC == 64 && c.x2 == nullptr && c.in_bn->CS == c.CinK &&
                                (c.CinK * 4 == 64 * nw || c.CinK * 2 == 64 * nw || c.CinK == 64 * nw)))
      return false;  // (MODE 5 / 6 / 7 exist for the 64-channel-slab kernels: 4, 2 or 1 threads per input channel)
  }
  if (c.rows2 != nullptr && c.in_mode != 0) return false;
  if (c.img2 != nullptr && !(c.rows2 != nullptr && c.H2 == 0 && KC == 16 && th == 14 && ntn == 1)) return false;
  a.N = c.N; a.H = c.H; a.W = c.W; a.CinK = c.CinK; a.CoutS = c.CoutS;
  a.tilesX = cdiv(c.W, 14); a.tilesY = cdiv(c.H, th); a.gy = ntn / (NT * nw);
  static const int env_remap = lab_env("SPCL_CONV_XCD_REMAP", 1);
  a.xcd_remap = (env_remap && a.gy == 1) ? 1 : 0;  // (with gy > 1 the z index carries the channel block: see the kernel)
  // pooled BatchNorm-backward sums in the epilogue (MODE 3): measured per block against dgrad + separate reduction pass
  // (N = 64): 32 -> 16 @112^2 43 vs 49 us, 128 -> 64 @28^2 21 vs 25.5, 256 -> 128 @14^2 30.5 vs 30.6, but the one-wave
  // 64 -> 32 @56^2 kernel 53.5 vs 43 (56 scattered 8-byte loads per lane behind one wave's MFMAs): not offered there
  // (that layer takes the two-wave form above)
  if (c.rows2 != nullptr && c.H2 > 0 && KC == 64 && nw == 1) return false;
#define SPCL_FAST_CASE(KC_, TH_, NT_, NW_)                               \
  if (KC == KC_ && th == TH_ && NT == NT_ && nw == NW_) {                \
    if (!dry) launch_fast<KC_, TH_, NT_, NW_>(a, c.in_mode, st);         \
    return true;                                                         \
  }
  SPCL_FAST_CASE(16, 14, 1, 1)  // Conv1.b forward / dgrad (16 -> 16 @ 224^2)
  SPCL_FAST_CASE(16, 14, 2, 1)  // Up_conv2.a dgrad (16 -> 32 @ 224^2)
  SPCL_FAST_CASE(32, 14, 1, 1)  // Up_conv2.a forward (cat(16, 16) -> 16 @ 224^2)
  SPCL_FAST_CASE(16, 7, 2, 1)   // Conv2.a forward (16 -> 32 @ 112^2)
  SPCL_FAST_CASE(16, 7, 1, 2)   //   "   as two one-n-tile waves (SPCL_CONV_FAST_NARROW_NT1)
  SPCL_FAST_CASE(32, 7, 1, 2)   // Conv2.b   "
  SPCL_FAST_CASE(32, 7, 1, 1)   // Conv2.a dgrad (32 -> 16)
  SPCL_FAST_CASE(32, 7, 2, 1)   // Conv2.b forward / dgrad
  SPCL_FAST_CASE(32, 7, 2, 2)   // Conv3.a forward (32 -> 64 @ 56^2)
  SPCL_FAST_CASE(64, 7, 2, 1)   // Conv3.a dgrad (64 -> 32)
  SPCL_FAST_CASE(64, 7, 1, 2)   //   "   as two one-n-tile waves
  SPCL_FAST_CASE(64, 7, 2, 2)   // Conv3.b, Conv4.a dgrad
  SPCL_FAST_CASE(64, 7, 1, 4)   //   "   as four one-n-tile waves
  SPCL_FAST_CASE(64, 7, 2, 4)   // Conv4.a forward, Conv4.b, Conv5 (7x14 tiles: 2 per 14x14 image)
#undef SPCL_FAST_CASE
  return false;
}

}  // namespace spcl
