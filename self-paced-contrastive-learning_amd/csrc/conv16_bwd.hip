// Backward of the image block's SECOND convolution (semi_seg/arch/unet.py:75 behind unet.py:72-74 with input_dim == 1:
// 16 -> 16 channels at the image resolution) in ONE pass over its two inputs.  The layer's gradient dy and the first
// conv's raw output y2 used to be read twice -- once by the weight-gradient kernel (wgrad.hip: x = relu(bn(y2)) and dy),
// once by the input-gradient kernel (conv_fast.hip MODE 4: dy through the flipped filters, y2 for the BatchNorm-backward
// sums of the layer below) -- 2 x 206 MB at 64 x 224^2.  Here a workgroup of NW waves (2; 1 and 4 are built for comparison)
// per 14 x 14 tile
//   1. stages the dy halo (16 x 16 pixels, zero outside the image) in LDS and forms g = dgrad(dy) on the matrix pipe
//      (the conv_fast.hip mapping: 13 m-tiles of 16 pixels -- dealt to the waves -- x 5 k-steps of (tap, 8 channels) chunks);
//   2. loads y2 at the tile's pixels ONCE: x = relu(scale2 y2 + shift2) goes to LDS pixel-major, and
//        dW[tap][ci][co] += sum_q x[q][ci] dy[q - tap][co]                 (q over the tile, dy from the halo image)
//      runs as 7 k-steps of 32 pixels with BOTH operands read transposed (ds_read_b64_tr_b16), the taps dealt to the waves
//      -- accumulated in registers over the workgroup's tiles, one partial slab per workgroup at the end (summed by
//      wgrad_reduce_kernel or by the batched launch's tail, as the stand-alone kernel's slabs are);
//   3. dz = g [x > 0], the BatchNorm-backward sums of the layer below and the nine image tap sums of its weight gradient
//      (image3, bn.hip) as MODE 4 leaves them: rows11 [tile][11][16] (dz takes x's place in LDS, read transposed as well).
// Nothing is written per pixel.  LDS per workgroup: dy halo 11 264 B + image copies 1 536 B + x tile (later dz) 7 168 B +
// 512 B = 20 480 B.  A workgroup walks its tiles with the next tile's global requests in flight.
//
// What the measurements said (tools/diag/conv16_phases.py, in-kernel stamps): the two kernels this
// replaces are NOT bound by HBM but by instruction issue -- per tile ~1 300 vector-ALU instructions (4 cycles each on a
// SIMD), 170 KB of LDS traffic and 135 MFMAs that hardly overlap -- so reading dy and y2 once saves the duplicated staging
// work (~10 us of 104), not the 206 MB.  Traps on the way, each worth a third of the kernel's time: image loads behind
// `if (wave == 0)` went through one register with s_waitcnt vmcnt(0) after each; the first in-loop use of the BatchNorm
// coefficients carried the compiler's vmcnt(0) for their pre-loop load, i.e. a wait for the freshly issued prefetch on every
// trip; lane-invariant addresses hoisted out of the tile loop and spilled; prefetched values spilled right behind their loads.
//
// Round 4: the production kernel is conv16_bwd_rows_kernel further down (m-tiles along the tile rows, two waves, four waves per
// SIMD: 84.6 -> 72.8 us); conv16_bwd_kernel below -- linear m-tiles, 1 / 2 / 4 waves per tile -- stays as the reference the
// row-ordered form is compared with bit for bit (tools/diag/conv16_check.py) and behind SPCL_CONV16_ROWMAP=0.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_common.hpp"

namespace spcl {

struct Bwd16Args {
  const unsigned char* dy;  // [N][H][W][16] bf16
  const u32x4* wp;          // packed dgrad fragments of the 16 -> 16 filters (conv.hip kind 1): 5 k-steps x 64 lanes
  const unsigned char* y2;  // [N][H][W][16] bf16, raw output of the first conv
  const float* scale2;      // its BatchNorm scale / shift / batch mean [16]
  const float* shift2;
  const float* mean2;
  const float* img;         // [N][H][W] f32, the block's input
  float* rows11;            // [tile][11][16], or null with wg_rows
  // ... or ONE row set per WORKGROUP (its tiles summed in registers), written as the final kernel wants it,
  // [sub-row 11][channel 16][workgroup] (bn.hip bnrelu_bwd_fin_kernel, transposed): no per-tile rows, no folding launch
  float* wg_rows;
  int nwg;                  // workgroups that walk tiles (the grid's extra z-slice folds the autocorrelation rows, below)
  const float* acorr_in;    // [nacorr][64] partial rows of image_autocorr_kernel -> acorr_out [16][64] (row j: rows j, j + 16, ..)
  int nacorr;
  float* acorr_out;
  float* partial;           // [workgroup][9][16 ci][16 co]
  int N, H, W, tilesX, tilesY, ipw, xcd_remap;
  int dbg;  // ablation bits (timing experiments only, wrong results; 0 in production): 1 no dgrad MFMAs, 2 no pixel pass,
            // 4 no weight-gradient phase, 8 no dz / tap sums, 16 no global requests after the first tile
  unsigned long long* stamps;  // debug (SPCL_CONV16_STAMPS=1): s_memtime ticks of wave 0 per phase, summed over its tiles
};

constexpr int B16_TH = 14, B16_TW = 14, B16_HW = 16, B16_RP = 22, B16_PS = 32;
constexpr int B16_DY = 0, B16_DY_BYTES = (B16_TH + 2) * B16_RP * B16_PS;  // 11 264
constexpr int B16_IM = B16_DY_BYTES, B16_IM_BYTES = 1536;
constexpr int B16_XT = B16_IM + B16_IM_BYTES, B16_XT_ROW = 16 * 32, B16_XT_BYTES = B16_TH * B16_XT_ROW;
constexpr int B16_RED = B16_XT + B16_XT_BYTES;  // [wave <= 4][2][16] f32: the waves' shares of the BatchNorm sums
constexpr int B16_ACC = B16_RED + 512;           // [9 taps][16] f32: the workgroup's tap sums over its tiles (WGROWS)
constexpr int B16_TRASH = B16_ACC + 576;         // 32 bytes nobody reads (see the throw-away store before the tile loop)
constexpr int B16_LDS = B16_TRASH + 32;          // 21 088

// the ablation bits of Bwd16Args.dbg exist only in -DSPCL_CONV16_DBG_BUILD=1 builds (tools/diag/conv16_phases.py): as run-time
// conditions they were scalar branches inside the unrolled loops -- the first build had ~200 branches per tile and wave
#ifndef SPCL_CONV16_DBG_BUILD
#define SPCL_CONV16_DBG_BUILD 0
#endif
#define B16_DBG(a) (SPCL_CONV16_DBG_BUILD ? (a).dbg : 0)

__device__ __forceinline__ bf16x8 b16_tr_frag(unsigned addr) {
  // 8 pixels (k) of the lane's channel: two hardware-transposed reads of 4 pixels x 16 channels, 8 pixels apart
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)addr);
  const s16x4 hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(addr + 8 * B16_PS));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 b16_tr_frag2(unsigned addr_lo, unsigned addr_hi) {  // the two halves at their own addresses
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)addr_lo);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)addr_hi);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// SHIFTED: the image size is not a multiple of the tile (last tiles shifted back inside: pixels two tiles cover count once)
// NW: waves per tile (1, 2 or 4): m-tile i of 13 goes to wave i % NW, tap t of 9 to wave t % NW
constexpr int b16_wpe(int NW) { return NW == 1 ? 2 : 3; }
// WGROWS: row sets per workgroup (wg_rows) instead of per tile (rows11)
template <bool SHIFTED, int NW, bool WGROWS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(b16_wpe(NW)))) void conv16_bwd_kernel(Bwd16Args a) {
  constexpr int TH = B16_TH, TW = B16_TW, HW_ = B16_HW, RP = B16_RP, PS = B16_PS;
  constexpr int NPIX = TH * TW, NSTEPS = 5, NTHR = 64 * NW, ITER = 512 / NTHR, RPI = NTHR / 32;  // halo rows per staging step
  constexpr int MW = (13 + NW - 1) / NW, TPW = (9 + NW - 1) / NW;  // m-tiles / taps per wave: i = wave + NW j
  constexpr int SPX = (16 * NW) % TW, SPY = (16 * NW) / TW;         // 16 NW pixels on = SPY rows + SPX columns
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;
  const int t0 = threadIdx.x;
  const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(t0 >> 6);
  if (WGROWS && (int)blockIdx.z * (int)(gridDim.x * gridDim.y) >= a.nwg) {  // the extra z-slice (bn.hip bwd_rows_group_kernel's job)
    // all 16 folded rows are written whatever the grid: a slice of fewer than 16 workgroups (W == 14 with H = 126 .. 210:
    // 9 .. 15 tiles) walks the rows j, j + nslice, ... (rows >= tilesX * tilesY stayed uninitialised and were
    // summed into the first conv's weight gradient, d gamma and d beta)
    const int nslice = gridDim.x * gridDim.y;
    if (t0 < 64) {
      for (int j = blockIdx.y * gridDim.x + blockIdx.x; j < 16; j += nslice) {
        float sacc = 0.f;
#pragma unroll 8
        for (int w = j; w < a.nacorr; w += 16) sacc += a.acorr_in[(size_t)w * 64 + t0];
        a.acorr_out[(size_t)j * 64 + t0] = sacc;
      }
    }
    return;
  }

  int tx = blockIdx.x, ty = blockIdx.y;
  {
    const int T = a.tilesX * a.tilesY;  // (conv_fast.hip: an XCD's workgroups take neighbouring tiles)
    if (a.xcd_remap && (T & 7) == 0) {
      const int L = ty * a.tilesX + tx;
      const int L2 = (L & 7) * (T >> 3) + (L >> 3);
      ty = L2 / a.tilesX;
      tx = L2 - ty * a.tilesX;
    }
  }
  const int y0 = min(ty * TH, a.H - TH), x0 = min(tx * TW, a.W - TW);
  const int oy = ty * TH - y0, ox = tx * TW - x0;
  const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + TW < a.W;
  const bool shifted = SHIFTED && (oy | ox) != 0;
  const int rowb = 32;

  u32x4 wall[NSTEPS];
#pragma unroll
  for (int s = 0; s < NSTEPS; ++s) wall[s] = a.wp[s * 64 + (t0 & 63)];
  const f32x4 sc2 = *(const f32x4*)(a.scale2 + 4 * ((t0 & 63) >> 4)), sh2 = *(const f32x4*)(a.shift2 + 4 * ((t0 & 63) >> 4)),
              mu2 = *(const f32x4*)(a.mean2 + 4 * ((t0 & 63) >> 4));
  // zero for the whole launch: the two pad columns of the x / dz tile, pad pixels 16, 17 of every dy halo row (the shifted
  // weight-gradient fragments read them next to x == 0; LDS starts out undefined)
  if (t0 < 4 * TH) *(u32x4*)(lds + B16_XT + (t0 >> 2) * B16_XT_ROW + TW * PS + (t0 & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (t0 >= NTHR - 64)
    *(u32x4*)(lds + B16_DY + (((t0 & 63) >> 2) * RP + HW_) * PS + (t0 & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};

  if (t0 < 48) {  // copy kx = 1: column 15, kx = 2: columns 14, 15 of the 16 rows
    const int r = t0 & 15, k = t0 >> 4;
    ((bf16_t*)(lds + B16_IM))[(k == 0 ? 256 + 15 : (k == 1 ? 512 + 14 : 512 + 15)) + r * 16] = 0;
  }
  if (WGROWS)  // the per-wave BatchNorm sums and wave 0's tap sums of the workgroup's tiles live in LDS (carried in registers
               // they pushed the kernel over its budget: 204 bytes of scratch per lane)
    for (int i = t0; i < (512 + 576) / 4; i += NTHR) ((float*)(lds + B16_RED))[i] = 0.f;
  // wave w owns the taps w, w + 4, (w + 8): no cross-wave reduction of the weight gradient
  f32x4 wacc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) wacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- global requests of one tile.  issue_halo: the image halo (wave 0: 4 pixels per lane, clamped addresses -- the zero
  // padding is applied when they are used) and the dy halo (2 chunks per thread, masked lanes keep zero); issue_y2: y2 at the
  // wave's pixels (4 x 8 bytes).  Issued for tile t + 1 before / after the weight-gradient phase of tile t.  Everything
  // derived from the thread index is recomputed behind an opaque copy where it is used: left loop-invariant, the addresses
  // and flags are hoisted out of the tile loop and spilled around it.
  float imgv[256 / (64 * NW)];
  u32x4 v[ITER];
  uint2 ypre[MW];
  // image halo: the workgroup's threads share the 16 x 16 pixels (PPT consecutive columns each: one register per thread with
  // four waves -- as four registers of wave 0 alone they were the first to be spilled, right behind their loads)
  constexpr int PPT = 256 / NTHR;
  auto issue_img = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int hr = (t * PPT) >> 4, hc = (t * PPT) & 15;
    const int gy = y0 - 1 + hr;
    const float* ir = a.img + ((size_t)n * a.H + (gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy))) * a.W;
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
      const int gx = x0 - 1 + hc + e;
      imgv[e] = ir[gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx)];
    }
  };
  auto issue_halo = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int ch = t & 1, q0 = t >> 1, hy0 = q0 / HW_, hx0 = q0 % HW_;
    const unsigned voff = (unsigned)((hy0 * a.W + hx0) * 32 + ch * 16);
    const unsigned char* xb = a.dy + (((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * 32;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      const long soff = (long)(RPI * k) * a.W * 32;
      bool inb = true;
      if (!interior) {
        const int gy = y0 - 1 + hy0 + RPI * k, gx = x0 - 1 + hx0;
        inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      }
      v[k] = (u32x4){0u, 0u, 0u, 0u};
      if (inb) v[k] = *(const u32x4*)(xb + soff + voff);
    }
  };
  auto issue_y2 = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int r16 = t & 15, g = (t & 63) >> 4;
    const int p0 = 16 * wave + r16, py0 = p0 / TW, px0 = p0 - py0 * TW;
    const unsigned char* y2b = a.y2 + (((size_t)n * a.H + y0) * a.W + x0) * rowb + 4 * g * 2;
    int py = py0, px = px0;
#pragma unroll
    for (int j = 0; j < MW; ++j) {
      const bool ok = j < MW - 1 || 16 * (wave + NW * j) + r16 < NPIX;  // (only the last round can run past pixel 195)
      ypre[j] = *(const uint2*)(y2b + (ok ? (py * a.W + px) * rowb : 0));  // (a lane beyond the last pixel: unused)
      px += SPX;
      py += SPY;
      if (px >= TW) {
        px -= TW;
        py += 1;
      }
    }
  };

#ifndef SPCL_CONV16_STAMPS_BUILD
#define SPCL_CONV16_STAMPS_BUILD 0  /* -DSPCL_CONV16_STAMPS_BUILD=1 + SPCL_CONV16_STAMPS=1: the counters cost sixteen registers */
#endif
  const bool stamp = SPCL_CONV16_STAMPS_BUILD && a.stamps != nullptr && t0 == 0;
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tc = 0;
// (every phase boundary also fences the instruction scheduler: left free it hoists the next phase's LDS reads and address
// arithmetic across, runs out of registers and spills the image halo right behind its loads -- an s_waitcnt vmcnt(0) per tile)
#define B16_STAMP(k)                                      \
  __builtin_amdgcn_sched_barrier(0);                      \
  if (stamp) {                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    tph[k] += now_ - tc;                                  \
    tc = now_;                                            \
  }
  const int n_first = blockIdx.z * a.ipw;
  if (stamp) tc = __builtin_amdgcn_s_memtime();
  if (n_first < a.N) {
    issue_img(n_first);
    issue_halo(n_first);
    issue_y2(n_first);
  }
  // the filter fragments and BatchNorm coefficients have arrived BEFORE the tile loop: their first use inside it would carry
  // the compiler's wait, and on every later trip that s_waitcnt vmcnt(0) waits for the next tile's requests just issued
  // (a throw-away LDS store of a combination of them: ordinary code the compiler must wait for, on every path into the loop)
  {
    const u32x4 wx = wall[0] ^ wall[1] ^ wall[2] ^ wall[3] ^ wall[4];
    const f32x4 cx = sc2 + sh2 + mu2;
    *(u32x4*)(lds + B16_TRASH) = wx;
    *(f32x4*)(lds + B16_TRASH + 16) = cx;
  }
#pragma unroll 1
  for (int it = 0; it < a.ipw; ++it) {
    const int n = n_first + it;
    if (n >= a.N) break;
    int t = t0;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, r16 = t & 15, g = lane >> 4;
    const int p0 = 16 * wave + r16, py0 = p0 / TW, px0 = p0 - py0 * TW;
    const int tile = (n * a.tilesY + ty) * a.tilesX + tx;
    // ---- image copies IM[kx][row][col] = bf16(halo[row][col + kx]) (conv_fast.hip MODE 4), zero outside the image: a thread
    // puts each of its pixels where the three copies want it (columns 16 - kx .. 15 of copy kx stay zero, written once)
    {
      const int hr = (t * PPT) >> 4, hc = (t * PPT) & 15;
      const int gy = y0 - 1 + hr;
#pragma unroll
      for (int e = 0; e < PPT; ++e) {
        const int gx = x0 - 1 + hc + e;
        const float hv = (interior || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)) ? imgv[e] : 0.f;
        const bf16_t hb = f32_to_bf16(hv);
        bf16_t* imb = (bf16_t*)(lds + B16_IM) + hr * 16 + hc + e;
        imb[0] = hb;
        if (hc + e >= 1) imb[256 - 1] = hb;
        if (hc + e >= 2) imb[512 - 2] = hb;
      }
    }
    // ---- dy halo -> LDS
    {
      const int ch = t & 1, q0 = t >> 1, hy0 = q0 / HW_, hx0 = q0 % HW_;
      unsigned char* const lp = lds + B16_DY + (hy0 * RP + hx0) * PS + ch * 16;
#pragma unroll
      for (int k = 0; k < ITER; ++k) *(u32x4*)(lp + (RPI * k * RP) * PS) = v[k];
    }
    // ---- the next tile's image and dy requests, as soon as their registers are free: in flight for a whole tile period
    // (issued one phase before their use they were outstanding a third of the time -- 15 KB per workgroup, not enough bytes
    // in flight to stream from HBM)
    // (with one or two waves per tile the registers run out: there the requests follow the pixel pass)
    constexpr bool EARLY = NW >= 4;
    const bool more = it + 1 < a.ipw && n + 1 < a.N && !(B16_DBG(a) & 16);
    if (EARLY && more) issue_halo(n + 1);
    B16_STAMP(0)  // staging: image copies, wait for the dy halo, LDS stores, next requests
    __syncthreads();
    B16_STAMP(1)  // barrier 1

    // ---- input gradient g at the wave's m-tiles (pixels 16 (wave + 4 j) + r16): 5 k-steps
    f32x4 acc[MW];
    {
      int abase[MW];
      int py = py0, px = px0;
#pragma unroll
      for (int j = 0; j < MW; ++j) {
        const bool ok = j < MW - 1 || 16 * (wave + NW * j) + r16 < NPIX;  // (only the last round can run past pixel 195)
        abase[j] = B16_DY + (ok ? (py * RP + px) * PS : 0);
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        px += SPX;
        py += SPY;
        if (px >= TW) {
          px -= TW;
          py += 1;
        }
      }
      if (!(B16_DBG(a) & 1))
#pragma unroll
      for (int s = 0; s < NSTEPS; ++s) {
        int fc = 4 * s + g;
        if (fc >= 18) fc = 0;  // K padding: zero weights
        const int tap = fc >> 1, c = fc & 1, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int off = (ky * RP + kx) * PS + c * 16;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
          // (a wave whose last round holds no m-tile computes on the halo's first pixels and never uses the result: five
          // MFMAs cheaper than a scalar branch per (k-step, m-tile))
          const u32x4 xf = *(const u32x4*)(lds + abase[j] + off);
          acc[j] = mfma_chunk<bf16_t>(wall[s], xf, acc[j]);
        }
      }
    }

    B16_STAMP(2)  // dgrad
    // ---- one pass over the wave's pixels: x = relu(bn(y2)) -> LDS, pixel-major (a pixel another tile counts: 0);
    // dz = g [x > 0] with g as a bf16 value, parked as packed bf16; the BatchNorm-backward sums of the layer below
    uint2 dzp[MW];
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    {
      int px = px0, py = py0;
#pragma unroll
      for (int j = 0; j < MW; ++j) {
        const bool ok = j < MW - 1 || 16 * (wave + NW * j) + r16 < NPIX;  // (only the last round can run past pixel 195)
        dzp[j] = (uint2){0u, 0u};
        if (ok && !(B16_DBG(a) & 2)) {
          const bool keep = !shifted || (py >= oy && px >= ox);
          const float yv[4] = {__uint_as_float(ypre[j].x << 16), __uint_as_float(ypre[j].x & 0xffff0000u),
                               __uint_as_float(ypre[j].y << 16), __uint_as_float(ypre[j].y & 0xffff0000u)};
          const f32x2 glo = {acc[j][0], acc[j][1]}, ghi = {acc[j][2], acc[j][3]};
          const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
          const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
          const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                               __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
          float xv[4], dz[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = fmaf(sc2[r], yv[r], sh2[r]);
            xv[r] = keep ? fmaxf(z, 0.f) : 0.f;
            dz[r] = (z > 0.f && keep) ? gv[r] : 0.f;
            ssum[r] += dz[r];
            ssq[r] = fmaf(dz[r], yv[r] - mu2[r], ssq[r]);
          }
          const f32x2 lo = {xv[0], xv[1]}, hi = {xv[2], xv[3]};
          uint2 w;
          w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
          w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
          *(uint2*)(lds + B16_XT + py * B16_XT_ROW + px * PS + g * 8) = w;
          // (dz is a bf16 value or zero: its upper halves are the exact packing)
          dzp[j].x = (__float_as_uint(dz[0]) >> 16) | (__float_as_uint(dz[1]) & 0xffff0000u);
          dzp[j].y = (__float_as_uint(dz[2]) >> 16) | (__float_as_uint(dz[3]) & 0xffff0000u);
        }
        px += SPX;
        py += SPY;
        if (px >= TW) {
          px -= TW;
          py += 1;
        }
      }
    }
    {
      // the wave's share of the two sums -> LDS (per tile: wave 0 adds the shares in wave order after the barrier; WGROWS:
      // added to the wave's own running sums, combined once at the end)
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(ssum[r]), s2 = row16_sum(ssq[r]);
        o[r] = r16 == 0 ? s1 : s2;
      }
      if (r16 < 2) {
        f32x4* rp = (f32x4*)(lds + B16_RED + ((wave * 2 + r16) * 16 + 4 * g) * 4);
        *rp = WGROWS ? *rp + o : o;
      }
    }
    if (more) {  // (likewise: the y2 registers were consumed by the pass above; the image halo rides along)
      if (!EARLY) issue_halo(n + 1);
      issue_y2(n + 1);
      issue_img(n + 1);
    }
    B16_STAMP(3)  // pixel pass
    __syncthreads();
    B16_STAMP(1)
    if (!WGROWS && t < 32) {
      const float* red = (const float*)(lds + B16_RED) + t;
      float tot = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) tot += red[32 * w];
      a.rows11[(size_t)tile * 11 * 16 + t] = tot;
    }

    // ---- weight gradient of the tile: dW[tap][ci][co] += sum_q x[q][ci] dy[q + (2 - ky, 2 - kx)][co] in halo coordinates.
    // k-step ks = tile rows 2 ks, 2 ks + 1; lane group g -> row g >> 1, columns 4 (g & 1) .. + 3 and + 8 ..: the 32 lanes one
    // transposed read serves sit in ONE row, 256 contiguous bytes (no bank conflicts)
    const unsigned tr_col = (unsigned)((4 * (g & 1) + (r16 >> 2)) * PS + (r16 & 3) * 8);
    const unsigned tr_x = lds_base + B16_XT + (g >> 1) * B16_XT_ROW + tr_col;
    if (!(B16_DBG(a) & 4)) {
      unsigned ax = tr_x, ad[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int tap = min(wave + NW * j, 8), ky = tap / 3, kx = tap - 3 * ky;
        ad[j] = lds_base + B16_DY + (g >> 1) * RP * PS + tr_col + (2 - ky) * RP * PS + (2 - kx) * PS;
      }
      // (one wave per tile: rolled, or the scheduler hoists all 140 fragment reads of the nine taps and spills)
#pragma unroll NW == 1 ? 1 : TH / 2
      for (int ks = 0; ks < TH / 2; ++ks) {
        const bf16x8 af = b16_tr_frag(ax + ks * 2 * B16_XT_ROW);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          // (likewise: a wave without a tap in the last round repeats tap 8 into an accumulator nobody stores)
          const bf16x8 bf = b16_tr_frag(ad[j] + ks * 2 * RP * PS);
          wacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, wacc[j], 0, 0, 0);
        }
      }
    }
    B16_STAMP(4)  // weight gradient (+ the two row sums)
    __syncthreads();
    B16_STAMP(1)

    // ---- dz takes x's place in LDS; S[co][tap] = sum_p dz[p][co] img[p + tap] on the matrix pipe (wave 0: 7 k-steps of 32
    // pixels, same pixel order as above): A = dz read transposed, B = the lane's tap: 2 x 4 columns of the shifted image copy
    {
      int px = px0, py = py0;
#pragma unroll
      for (int j = 0; j < MW; ++j) {
        const bool ok = j < MW - 1 || 16 * (wave + NW * j) + r16 < NPIX;  // (only the last round can run past pixel 195)
        if (ok) *(uint2*)(lds + B16_XT + py * B16_XT_ROW + px * PS + g * 8) = dzp[j];
        px += SPX;
        py += SPY;
        if (px >= TW) {
          px -= TW;
          py += 1;
        }
      }
    }
    B16_STAMP(5)  // dz stores
    __syncthreads();
    B16_STAMP(1)
    if (wave == 0 && !(B16_DBG(a) & 8)) {
      const int tap = r16 < 9 ? r16 : 0, tky = tap / 3, tkx = tap - 3 * tky;
      const unsigned char* pb = lds + B16_IM + tkx * 512 + ((g >> 1) + tky) * 32 + (g & 1) * 8;
      f32x4* dp = (f32x4*)(lds + B16_ACC + (r16 * 16 + 4 * g) * 4);
      f32x4 D = (WGROWS && r16 < 9) ? *dp : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < TH / 2; ++ks) {
        const bf16x8 af = b16_tr_frag(tr_x + ks * 2 * B16_XT_ROW);
        const uint2 b0 = *(const uint2*)(pb + ks * 64), b1 = *(const uint2*)(pb + ks * 64 + 16);
        const u32x4 bq = {b0.x, b0.y, b1.x, b1.y};
        D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bq), D, 0, 0, 0);
      }
      if (WGROWS) {
        if (r16 < 9) *dp = D;
      } else if (r16 < 9) *(f32x4*)(a.rows11 + ((size_t)tile * 11 + 2 + r16) * 16 + 4 * g) = D;
    }
    B16_STAMP(6)  // tap sums
    if (NW > 1) __syncthreads();  // every wave restages the image copies: behind wave 0's reads
    // (the dy halo is free since the barrier after the weight gradient; x / dz is rewritten only after the next staging
    // barrier, which wave 0 reaches after its reads)
  }

  if (stamp) {
    unsigned long long* o = a.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = tph[k];
  }
  const size_t wg = (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (WGROWS) {
    // the workgroup's row set [11][16] -> wg_rows [sub-row][channel][workgroup]: the two BatchNorm sums from the waves' running
    // shares (wave order), the tap sums from wave 0's
    __syncthreads();
    for (int i = t0; i < 11 * 16; i += NTHR) {
      float tot;
      if (i < 32) {
        const float* red = (const float*)(lds + B16_RED) + i;
        tot = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += red[32 * w];
      } else {
        tot = ((const float*)(lds + B16_ACC))[i - 32];
      }
      a.wg_rows[(size_t)i * a.nwg + wg] = tot;  // i = sub-row * 16 + channel
    }
  }
  // ---- the workgroup's partial slab [tap][ci][co]: lane holds co = r16, ci = 4 g + r of the wave's taps
  float* out = a.partial + wg * (9 * 256);
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tap = wave + NW * j;
    if (tap < 9) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(tap * 16 + 4 * ((t0 & 63) >> 4) + r) * 16 + (t0 & 15)] = wacc[j][r];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same pass with the m-tiles laid along the TILE ROWS (round 4): m-tile = one row of the 14 x 14 tile as 16 pixel columns
// (14 real + 2 that read the zeroed pad pixels 16, 17 of the halo rows and are never used), wave w takes the rows w, w + 2,
// ... (7 each: the MFMA count per wave is the old one's -- 13 linear m-tiles dealt to two waves were 7 rounds as well).  A
// lane's pixel column is then r16 for every round and its row is wave-uniform, so everything the linear pixel order computed
// per round and per phase (row / column of pixel 16 (wave + 2 j) + r16, a carry, a validity flag -- four loops of seven
// rounds, ~150 of the ~670 vector instructions per tile and wave) becomes an immediate offset of the LDS instruction or a
// scalar addition: dgrad fragment address = lane base + (tap, chunk) offset of the lane's k-group (five dwords per k-group,
// read from a 4 x 5 table in LDS once per tile) + round * 2 rows; x / dz store address = lane base + round * 2 rows; y2
// address = scalar row base + lane offset.  The two unused pixel columns are switched off ONCE per tile for the whole pixel
// pass (they are the same lanes in every round), their y2 request is lane 13's (no read beyond the tensor's end).
// Same MFMA operands per output pixel and weight element as the linear kernel: g, x, dz and dW bit for bit; the
// BatchNorm sums add the same terms in another lane order.
// LDS of the row-mapped kernel: the dy halo with 18-pixel rows (16 + the two zeroed pad pixels; the 22-pixel pitch of the linear
// kernel kept ITS m-tiles, which cross tile rows, off each other's banks -- here every lane group of an access sits in one
// row): 19 168 B, eight workgroups = four waves per SIMD where the registers allow it
#ifndef SPCL_CONV16_ROWS_WPE
#define SPCL_CONV16_ROWS_WPE 4
#endif
// (the shifted-tile form -- image sizes that are not a multiple of 14 -- needs a few registers more for its keep masks: asked
// for four waves per SIMD it takes 128 registers and spills a prefetched halo register right behind its load; asked for three
// it comes out at 121 without a spill)
constexpr int r16_wpe(bool shifted) { return shifted && SPCL_CONV16_ROWS_WPE > 3 ? 3 : SPCL_CONV16_ROWS_WPE; }
constexpr int R16_RP = 18;
constexpr int R16_DY = 0, R16_DY_BYTES = (B16_TH + 2) * R16_RP * B16_PS;  // 9 216
constexpr int R16_IM = R16_DY_BYTES;
constexpr int R16_XT = R16_IM + B16_IM_BYTES;
constexpr int R16_RED = R16_XT + B16_XT_BYTES;
constexpr int R16_ACC = R16_RED + 512;
constexpr int R16_TRASH = R16_ACC + 576;
constexpr int R16_TAB = R16_TRASH + 32;        // [4 k-groups][8] dwords: (tap, chunk) byte offsets of the five k-steps
constexpr int B16_LDS_ROWS = R16_TAB + 128;    // 19 168

template <bool SHIFTED, bool WGROWS>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(r16_wpe(SHIFTED)))) void conv16_bwd_rows_kernel(Bwd16Args a) {
  constexpr int NW = 2;
  constexpr int TH = B16_TH, TW = B16_TW, HW_ = B16_HW, RP = R16_RP, PS = B16_PS;
  constexpr int NSTEPS = 5, NTHR = 64 * NW, ITER = 512 / NTHR, RPI = NTHR / 32;
  constexpr int MW = TH / NW, TPW = (9 + NW - 1) / NW;
  static_assert(TH % NW == 0, "rows are dealt to the waves evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;
  const int t0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
  if (WGROWS && (int)blockIdx.z * (int)(gridDim.x * gridDim.y) >= a.nwg) {  // the extra z-slice: the autocorrelation rows
    // (16 384 per-tile rows when the image convolution leaves them (spcl_conv3x3_forward_image_acorr): one column per thread
    // and one row per load was a chain of 1 024 dependent round trips -- 120 us, longer than the launch it hides in.  The
    // workgroup's 128 threads take 16 column quads x 8 row lanes with 16-byte loads, 16 in flight each; the row lanes are
    // added through LDS in lane order: fixed order, deterministic)
    const int nslice = gridDim.x * gridDim.y;
    float* fold = (float*)lds;  // [8 row lanes][64]
    const int cq = t0 & 15, rl = t0 >> 4;
    for (int j = blockIdx.y * gridDim.x + blockIdx.x; j < 16; j += nslice) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
      for (int w = j + 16 * rl; w < a.nacorr; w += 128) sacc += *(const f32x4*)(a.acorr_in + (size_t)w * 64 + 4 * cq);
      __syncthreads();
      *(f32x4*)(fold + rl * 64 + 4 * cq) = sacc;
      __syncthreads();
      if (t0 < 64) {
        float tot = fold[t0];
#pragma unroll
        for (int r = 1; r < 8; ++r) tot += fold[r * 64 + t0];
        a.acorr_out[(size_t)j * 64 + t0] = tot;
      }
    }
    return;
  }

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
  const int y0 = min(ty * TH, a.H - TH), x0 = min(tx * TW, a.W - TW);
  const int oy = ty * TH - y0, ox = tx * TW - x0;
  const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + TW < a.W;
  const bool shifted = SHIFTED && (oy | ox) != 0;

  if (t0 < 4 * TH) *(u32x4*)(lds + R16_XT + (t0 >> 2) * B16_XT_ROW + TW * PS + (t0 & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (t0 >= NTHR - 64)
    *(u32x4*)(lds + R16_DY + (((t0 & 63) >> 2) * RP + HW_) * PS + (t0 & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (t0 < 48) {
    const int r = t0 & 15, k = t0 >> 4;
    ((bf16_t*)(lds + R16_IM))[(k == 0 ? 256 + 15 : (k == 1 ? 512 + 14 : 512 + 15)) + r * 16] = 0;
  }
  if (t0 >= 64 && t0 < 64 + 20) {  // the k-step table: lane group g, k-step s -> chunk 4 s + g = (tap, 8-channel half)
    const int e = t0 - 64, gq = e / NSTEPS, s = e - gq * NSTEPS;
    int fc = 4 * s + gq;
    if (fc >= 18) fc = 0;  // K padding: zero weights
    const int tap = fc >> 1, c = fc & 1, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    ((unsigned*)(lds + R16_TAB))[gq * 8 + s] = (unsigned)((ky * RP + kx) * PS + c * 16);
  }
  if (WGROWS)
    for (int i = t0; i < (512 + 576) / 4; i += NTHR) ((float*)(lds + R16_RED))[i] = 0.f;
  f32x4 wacc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) wacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float imgv[256 / NTHR];
  u32x4 v[ITER];
  uint2 ypre[MW];
  constexpr int PPT = 256 / NTHR;
  // Every global request is a buffer load: a scalar descriptor per tensor and tile (base = the tile's first halo / tile /
  // image pixel: scalar arithmetic), ONE 32-bit lane offset per request kind and a scalar row offset per request -- as
  // global loads with 64-bit lane addresses each request cost two address registers and their additions
  auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); };
  constexpr int OOB = (int)0x80000000u;  // (>= num_records: the load returns zeros and touches no memory; 0x7ffffff0 is INSIDE 2 GiB - 1)
  auto issue_img = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int hr = (t * PPT) >> 4, hc = (t * PPT) & 15;
    const int gy = y0 - 1 + hr;
    const __amdgpu_buffer_rsrc_t rs = rsrc(a.img + (size_t)n * a.H * a.W);
    const int rowo = (gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy)) * a.W;
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
      const int gx = x0 - 1 + hc + e;
      imgv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (rowo + (gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx))) * 4, 0, 0));
    }
  };
  auto issue_halo = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int ch = t & 1, q0 = t >> 1, hy0 = q0 / HW_, hx0 = q0 % HW_;
    const int voff = (hy0 * a.W + hx0) * 32 + ch * 16;
    const __amdgpu_buffer_rsrc_t rs = rsrc(a.dy + (((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * 32);
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      bool inb = true;
      if (!interior) {
        const int gy = y0 - 1 + hy0 + RPI * k, gx = x0 - 1 + hx0;
        inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      }
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, inb ? voff : OOB, RPI * k * a.W * 32, 0));
    }
  };
  // y2 at the wave's rows: ONE lane offset (pixel column r16 -- 13 for the two unused columns -- and channel quarter g),
  // the row is a scalar offset
  auto issue_y2 = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int r16 = t & 15, g = (t & 63) >> 4;
    const int voff = min(r16, TW - 1) * 32 + g * 8;
    const __amdgpu_buffer_rsrc_t rs = rsrc(a.y2 + (((size_t)n * a.H + y0 + wave) * a.W + x0) * 32);
#pragma unroll
    for (int j = 0; j < MW; ++j)
      ypre[j] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, NW * j * a.W * 32, 0));
  };
  const __amdgpu_buffer_rsrc_t rs_wp = rsrc(a.wp), rs_sc = rsrc(a.scale2), rs_sh = rsrc(a.shift2), rs_mu = rsrc(a.mean2);

  const bool stamp = SPCL_CONV16_STAMPS_BUILD && a.stamps != nullptr && t0 == 0;
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tc = 0;
  const int n_first = blockIdx.z * a.ipw;
  if (stamp) tc = __builtin_amdgcn_s_memtime();
  if (n_first < a.N) {
    issue_img(n_first);
    issue_halo(n_first);
    issue_y2(n_first);
  }
  __syncthreads();  // the k-step table is read before the first staging barrier
  uint2 dzp[MW];
  // ---- second half of a tile: weight gradient, dz in x's place, the image tap sums
  auto second_half = [&](const int n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, r16 = t & 15, g = lane >> 4;
    const int tile = (n * a.tilesY + ty) * a.tilesX + tx;
    const unsigned xaddr = (unsigned)(R16_XT + wave * B16_XT_ROW + r16 * PS + (((g + (r16 >> 2)) & 3) << 3));
    B16_STAMP(3)
    __syncthreads();
    B16_STAMP(1)
    if (!WGROWS && t < 32) {
      const float* red = (const float*)(lds + R16_RED) + t;
      float tot = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) tot += red[32 * w];
      a.rows11[(size_t)tile * 11 * 16 + t] = tot;
    }

    // ---- weight gradient of the tile (as in the linear kernel: both operands read transposed, the taps dealt to the waves)
    const unsigned tr_col = (unsigned)((4 * (g & 1) + (r16 >> 2)) * PS + (r16 & 3) * 8);
    // (the x / dz tile's swizzle: quad r16 & 3 of pixel column 4 (g & 1) + (r16 >> 2) -- and of the column 8 further on)
    const unsigned tr_x = lds_base + R16_XT + (g >> 1) * B16_XT_ROW + (4 * (g & 1) + (r16 >> 2)) * PS + ((((r16 & 3) + (g & 1)) & 3) << 3);
    const unsigned tr_xh = (tr_x + 8 * PS) ^ 16u;  // (quad index + 2 mod 4: bit 4 of a 32-byte-aligned pixel's offset)
    {
      unsigned ad[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int tap = min(wave + NW * j, 8), ky = tap / 3, kx = tap - 3 * ky;
        ad[j] = lds_base + R16_DY + (g >> 1) * RP * PS + tr_col + (2 - ky) * RP * PS + (2 - kx) * PS;
      }
#pragma unroll
      for (int ks = 0; ks < TH / 2; ++ks) {
        const bf16x8 af = b16_tr_frag2(tr_x + ks * 2 * B16_XT_ROW, tr_xh + ks * 2 * B16_XT_ROW);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const bf16x8 bf = b16_tr_frag(ad[j] + ks * 2 * RP * PS);
          wacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, wacc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // (one k-step's twelve fragment reads in flight, not seven: registers)
      }
    }
    B16_STAMP(4)
    __syncthreads();
    B16_STAMP(1)

    // ---- dz takes x's place in LDS; the nine image tap sums on the matrix pipe (wave 0)
#pragma unroll
    for (int j = 0; j < MW; ++j) *(uint2*)(lds + xaddr + j * (NW * B16_XT_ROW)) = dzp[j];
    B16_STAMP(5)
    __syncthreads();
    B16_STAMP(1)
    if (wave == 0) {
      const int tap = r16 < 9 ? r16 : 0, tky = tap / 3, tkx = tap - 3 * tky;
      const unsigned char* pb = lds + R16_IM + tkx * 512 + ((g >> 1) + tky) * 32 + (g & 1) * 8;
      f32x4* dp = (f32x4*)(lds + R16_ACC + (r16 * 16 + 4 * g) * 4);
      f32x4 D = (WGROWS && r16 < 9) ? *dp : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < TH / 2; ++ks) {
        const bf16x8 af = b16_tr_frag2(tr_x + ks * 2 * B16_XT_ROW, tr_xh + ks * 2 * B16_XT_ROW);
        const uint2 b0 = *(const uint2*)(pb + ks * 64), b1 = *(const uint2*)(pb + ks * 64 + 16);
        const u32x4 bq = {b0.x, b0.y, b1.x, b1.y};
        D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bq), D, 0, 0, 0);
      }
      if (WGROWS) {
        if (r16 < 9) *dp = D;
      } else if (r16 < 9) *(f32x4*)(a.rows11 + ((size_t)tile * 11 + 2 + r16) * 16 + 4 * g) = D;
    }
    B16_STAMP(6)
    __syncthreads();  // every wave restages the image copies: behind wave 0's reads
  };
  // The tile loop leaves from its MIDDLE (behind the pixel pass, where the next tile's requests are issued) and the last
  // tile's second half runs behind it: with `if (more) issue` inside one body the prefetched registers met their old values
  // in a phi at the loop's end, and the copies there waited for the requests just issued (s_waitcnt vmcnt(0) + 37 v_mov per
  // tile in one build of this kernel).
  int n = n_first;
  if (n_first < a.N)
#pragma unroll 1
  for (int it = 0;; ++it, ++n) {
    int t = t0;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, r16 = t & 15, g = lane >> 4;
    // the filter fragments (5 KB, L1-resident) are requested again for every tile -- behind the tile's own requests issued a
    // tile period ago, used after everything older: twenty registers that are free outside the dgrad rows
    u32x4 wall[NSTEPS];
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s)
      wall[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, lane * 16, s * 1024, 0));
    // ... and so are the BatchNorm coefficients of the lane's channel quad.  The two unused pixel columns (and, in a shifted
    // tile, the columns another tile counts) carry scale 0, shift -1: their x = relu(-1) = 0 and their dz = 0 fall out of the
    // same arithmetic -- no mask, no switched-off lanes in the pixel pass
    f32x4 sc2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sc, g * 16, 0, 0));
    f32x4 sh2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sh, g * 16, 0, 0));
    const f32x4 mu2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_mu, g * 16, 0, 0));
    if (r16 >= TW || (SHIFTED && r16 < ox)) {
      sc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      sh2 = (f32x4){-1.f, -1.f, -1.f, -1.f};
    }
    const u32x4 tabA = *(const u32x4*)(lds + R16_TAB + g * 32);
    const unsigned tabB = *(const unsigned*)(lds + R16_TAB + g * 32 + 16);
    {
      const int hr = (t * PPT) >> 4, hc = (t * PPT) & 15;
      const int gy = y0 - 1 + hr;
#pragma unroll
      for (int e = 0; e < PPT; ++e) {
        const int gx = x0 - 1 + hc + e;
        const float hv = (interior || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)) ? imgv[e] : 0.f;
        const bf16_t hb = f32_to_bf16(hv);
        bf16_t* imb = (bf16_t*)(lds + R16_IM) + hr * 16 + hc + e;
        imb[0] = hb;
        if (hc + e >= 1) imb[256 - 1] = hb;
        if (hc + e >= 2) imb[512 - 2] = hb;
      }
    }
    {
      const int ch = t & 1, q0 = t >> 1, hy0 = q0 / HW_, hx0 = q0 % HW_;
      unsigned char* const lp = lds + R16_DY + (hy0 * RP + hx0) * PS + ch * 16;
#pragma unroll
      for (int k = 0; k < ITER; ++k) *(u32x4*)(lp + (RPI * k * RP) * PS) = v[k];
    }
    B16_STAMP(0)
    __syncthreads();
    B16_STAMP(1)

    // ---- row by row: the input gradient g of the wave's row wave + 2 j (pixel column r16, 5 k-steps) and its pixel pass
    // (x = relu(bn(y2)) -> LDS, dz = g [x > 0] parked as packed bf16, the BatchNorm-backward sums): one accumulator alive.
    // The x / dz tile is swizzled -- channel quad q of pixel column p sits at p * 32 + ((q + (p >> 2)) & 3) * 8 -- so that
    // the 16 lanes one ds_write_b64 cycle serves (one quad, 16 columns) hit 16 different bank pairs (dense: four-way
    // conflicts, a third of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT)
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    const unsigned xaddr = (unsigned)(R16_XT + wave * B16_XT_ROW + r16 * PS + (((g + (r16 >> 2)) & 3) << 3));
    {
      const unsigned lb = (unsigned)(R16_DY + (wave * RP + r16) * PS);
      const unsigned offs[NSTEPS] = {tabA[0] + lb, tabA[1] + lb, tabA[2] + lb, tabA[3] + lb, tabB + lb};
#pragma unroll
      for (int j = 0; j < MW; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
          const u32x4 xf = *(const u32x4*)(lds + offs[s] + j * (NW * RP * PS));
          acc = mfma_chunk<bf16_t>(wall[s], xf, acc);
        }
        const bool keep_y = !shifted || wave + NW * j >= oy;  // (wave-uniform; shifted tiles only)
        const float yv[4] = {__uint_as_float(ypre[j].x << 16), __uint_as_float(ypre[j].x & 0xffff0000u),
                             __uint_as_float(ypre[j].y << 16), __uint_as_float(ypre[j].y & 0xffff0000u)};
        const f32x2 glo = {acc[0], acc[1]}, ghi = {acc[2], acc[3]};
        const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
        const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
        const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                             __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
        float xv[4], dz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float z = fmaf(sc2[r], yv[r], sh2[r]);
          if (SHIFTED) z = keep_y ? z : -1.f;
          xv[r] = z;  // (ReLU on the rounded pairs below)
          dz[r] = z > 0.f ? gv[r] : 0.f;
          ssum[r] += dz[r];
          ssq[r] = fmaf(dz[r], yv[r] - mu2[r], ssq[r]);
        }
        const f32x2 lo = {xv[0], xv[1]}, hi = {xv[2], xv[3]};
        uint2 w;
        w.x = relu_bf16x2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v)));
        w.y = relu_bf16x2(__builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v)));
        *(uint2*)(lds + xaddr + j * (NW * B16_XT_ROW)) = w;
        dzp[j].x = (__float_as_uint(dz[0]) >> 16) | (__float_as_uint(dz[1]) & 0xffff0000u);
        dzp[j].y = (__float_as_uint(dz[2]) >> 16) | (__float_as_uint(dz[3]) & 0xffff0000u);
        // (packed HERE: the compiler sinks the packing to the stores behind the weight-gradient phase and keeps the four
        // f32 values of every row alive until then -- 28 registers instead of 14)
        asm volatile("" : "+v"(dzp[j].x), "+v"(dzp[j].y));
        // (left free, the scheduler hoists every row's fragment reads and spills.  Requesting row j + 1's fragments right
        // behind row j's MFMAs -- into the registers those just read, so that they land under the row's vector instructions --
        // measured no different: with four waves per SIMD the LDS latency is covered already.  Neither did a start-up stagger
        // of the workgroups by thirds of a tile period: the phases are not in lock-step)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    B16_STAMP(2)
    {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(ssum[r]), s2 = row16_sum(ssq[r]);
        o[r] = r16 == 0 ? s1 : s2;
      }
      if (r16 < 2) {
        f32x4* rp = (f32x4*)(lds + R16_RED + ((wave * 2 + r16) * 16 + 4 * g) * 4);
        *rp = WGROWS ? *rp + o : o;
      }
    }
    if (!(it + 1 < a.ipw && n + 1 < a.N)) break;
    issue_halo(n + 1);
    issue_y2(n + 1);
    issue_img(n + 1);
    second_half(n);
  }
  if (n_first < a.N) second_half(n);

  if (stamp) {
    unsigned long long* o = a.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = tph[k];
  }
  const size_t wg = (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (WGROWS) {
    __syncthreads();
    for (int i = t0; i < 11 * 16; i += NTHR) {
      float tot;
      if (i < 32) {
        const float* red = (const float*)(lds + R16_RED) + i;
        tot = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += red[32 * w];
      } else {
        tot = ((const float*)(lds + R16_ACC))[i - 32];
      }
      a.wg_rows[(size_t)i * a.nwg + wg] = tot;
    }
  }
  float* out = a.partial + wg * (9 * 256);
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tap = wave + NW * j;
    if (tap < 9) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(tap * 16 + 4 * ((t0 & 63) >> 4) + r) * 16 + (t0 & 15)] = wacc[j][r];
    }
  }
}

int conv16_bwd_rowmap() {
  // the row-mapped kernel (two waves per tile): SPCL_CONV16_ROWMAP=0 switches back to the linear pixel order
  static const int env = lab_env("SPCL_CONV16_ROWMAP", 1);
  return env;
}

int conv16_bwd_nw() {
  // measured inside the step (64 x 224^2, same box, median of single replays; separate launches 1204.7 us): one wave per
  // tile 1208, two 1194, four 1242
  static const int env = lab_env("SPCL_CONV16_NW", 2);
  return env == 1 || env == 4 ? env : 2;
}

int conv16_bwd_ipw(int N, int H, int W) {
  // one resident generation of workgroups where the batch allows it: LDS (20 KB) and registers allow 8 / 6 / 3 per CU
  const int nw = conv16_bwd_nw();
  // (the shifted-tile form is built for three waves per SIMD and comes out at 121 registers: four are resident all the same)
  const int resident = 256 * (nw == 1 ? 8 : (nw == 2 ? (conv16_bwd_rowmap() ? 2 * SPCL_CONV16_ROWS_WPE : 6) : 3));
  const long tiles = (long)cdiv(H, B16_TH) * cdiv(W, B16_TW);
  static const int env_ipw = lab_env("SPCL_CONV16_IPW", 0);
  int ipw = env_ipw > 0 ? env_ipw : (int)((tiles * N + resident - 1) / resident);
  if (ipw < 1) ipw = 1;
  if (ipw > N) ipw = N;
  return ipw;
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_conv16_bwd_fused_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_flag("SPCL_NO_CONV16_FUSED");  // A/B switch
  if (off || CinK != 16 || CoutS != 16) return 0;
  return spcl_conv_dgrad_bnstats_image_supported(dtype, N, H, W, CinK, CoutS);
}

extern "C" int spcl_conv16_bwd_fused_splits(int N, int H, int W) {
  if (N <= 0 || H < B16_TH || W < B16_TW) return 0;
  const int ipw = conv16_bwd_ipw(N, H, W);
  return cdiv(H, B16_TH) * cdiv(W, B16_TW) * cdiv(N, ipw);
}

extern "C" int spcl_conv16_bwd_fused(const void* dy, int dtype, int N, int H, int W, const void* w_packed_dgrad,
                                     const void* y2, const float* scale2, const float* shift2, const float* mean2,
                                     const float* image, float* rows11, float* partial, float* dw_oihw, int Cin, int Cout,
                                     float* wg_rows, const float* acorr, int nacorr, float* acorr16, void* stream) {
  SPCL_CHECK_ARG(dy && w_packed_dgrad && y2 && scale2 && shift2 && mean2 && image && partial && dw_oihw,
                 "conv16_bwd_fused: null pointer");
  SPCL_CHECK_ARG((rows11 != nullptr) != (wg_rows != nullptr), "conv16_bwd_fused: exactly one of rows11 / wg_rows");
  SPCL_CHECK_ARG(wg_rows == nullptr || (acorr && acorr16 && nacorr > 0), "conv16_bwd_fused: wg_rows comes with the autocorrelation rows");
  SPCL_CHECK_ARG(Cin > 0 && Cin <= 16 && Cout > 0 && Cout <= 16, "conv16_bwd_fused: channel counts");
  if (!spcl_conv16_bwd_fused_supported(dtype, N, H, W, 16, 16)) {
    set_error("conv16_bwd_fused: unsupported configuration (bf16, 16 -> 16 channels, 14 x 14 tiles)");
    return SPCL_EUNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  spcl_wgrad_tail* tail = take_tail_capture();
  Bwd16Args a;
  a.dy = (const unsigned char*)dy; a.wp = (const u32x4*)w_packed_dgrad; a.y2 = (const unsigned char*)y2;
  a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2; a.img = image; a.rows11 = rows11; a.partial = partial;
  a.wg_rows = wg_rows; a.acorr_in = acorr; a.nacorr = nacorr; a.acorr_out = acorr16;
  a.N = N; a.H = H; a.W = W; a.tilesX = cdiv(W, B16_TW); a.tilesY = cdiv(H, B16_TH);
  a.ipw = conv16_bwd_ipw(N, H, W);
  static const int env_remap = lab_env("SPCL_CONV_XCD_REMAP", 1);
  a.xcd_remap = env_remap;
  a.dbg = lab_env("SPCL_CONV16_DBG", 0);  // (read per call: experiments flip it)
  const int nz = cdiv(N, a.ipw), nsplit = a.tilesX * a.tilesY * nz;
  a.nwg = nsplit;
  const double px = (double)N * H * W;
  prof_cost(px * 32.0 * 2.0 + px * 4.0 + (double)nsplit * 9 * 256 * 4.0, 2.0 * px * 9.0 * 256 * 2.0 + 2.0 * px * 9.0 * 16);
  a.stamps = nullptr;
  const bool want_stamps = SPCL_CONV16_STAMPS_BUILD && lab_flag("SPCL_CONV16_STAMPS");  // debug only (synchronises)
  const size_t nwg = (size_t)nsplit;
  if (want_stamps) {
    (void)hipMalloc(&a.stamps, nwg * 8 * sizeof(unsigned long long));
    (void)hipMemset(a.stamps, 0, nwg * 8 * sizeof(unsigned long long));
  }
  const int nw = conv16_bwd_nw();
  const bool even = H % B16_TH == 0 && W % B16_TW == 0, wgr = wg_rows != nullptr;
  const dim3 grid(a.tilesX, a.tilesY, nz + (wgr ? 1 : 0));
#define B16_LAUNCH(SH_, NW_, WG_) SPCL_LAUNCH((conv16_bwd_kernel<SH_, NW_, WG_>), grid, dim3(64 * NW_), B16_LDS, st, a)
#define B16_CASE(NW_)                                       \
  if (nw == NW_) {                                          \
    if (even && wgr) B16_LAUNCH(false, NW_, true);          \
    else if (even) B16_LAUNCH(false, NW_, false);           \
    else if (wgr) B16_LAUNCH(true, NW_, true);              \
    else B16_LAUNCH(true, NW_, false);                      \
  }
  if (nw == 2 && conv16_bwd_rowmap()) {
#define B16_ROWS(SH_, WG_) SPCL_LAUNCH((conv16_bwd_rows_kernel<SH_, WG_>), grid, dim3(128), B16_LDS_ROWS, st, a)
    if (even && wgr) B16_ROWS(false, true);
    else if (even) B16_ROWS(false, false);
    else if (wgr) B16_ROWS(true, true);
    else B16_ROWS(true, false);
#undef B16_ROWS
  }
#if SPCL_LAB  // (the linear-m-tile kernel lost its A/B in round 4: instantiated in lab builds only -- SPCL_CONV16_ROWMAP=0 / _NW)
  else {
    B16_CASE(1) B16_CASE(2) B16_CASE(4)
  }
#endif
#undef B16_CASE
#undef B16_LAUNCH
  if (a.stamps != nullptr) {
    std::vector<unsigned long long> h(nwg * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(a.stamps);
    double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < nwg; ++i)
      for (int k = 0; k < 7; ++k) ph[k] += (double)h[i * 8 + k];
    fprintf(stderr, "[conv16_bwd stamps] %zu workgroups x %d tiles | s_memtime ticks of wave 0 per TILE: staging %.0f, barriers %.0f, "
            "dgrad %.0f, pixel pass %.0f, wgrad %.0f, dz stores %.0f, tap sums %.0f\n", nwg, a.ipw, ph[0] / nwg / a.ipw,
            ph[1] / nwg / a.ipw, ph[2] / nwg / a.ipw, ph[3] / nwg / a.ipw, ph[4] / nwg / a.ipw, ph[5] / nwg / a.ipw,
            ph[6] / nwg / a.ipw);
  }
  if (tail != nullptr) {
    tail->partial = partial; tail->dw = dw_oihw; tail->kind = 0; tail->nsplit = nsplit; tail->nblk_ci = 1;
    tail->nblk_co = 1; tail->CIB = 16; tail->COB = 16; tail->Cin = Cin; tail->Cout = Cout;
  } else {
    launch_wgrad_reduce(partial, nsplit, 1, 1, 16, 16, Cin, Cout, dw_oihw, st);
  }
  SPCL_LAUNCH_CHECK("conv16_bwd_fused");
  return SPCL_OK;
}
