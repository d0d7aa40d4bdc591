// 3x3 same-convolution (stride 1, pad 1, no bias) on NHWC activations as an im2col-free implicit GEMM on the
// gfx950 matrix cores.  Replaces nn.Conv2d(in,out,3,1,1,bias=False) of semi_seg/arch/unet.py:72,75 (forward) and
// the data-gradient half of its autograd backward (dgrad == the same kernel on 180-degree-flipped, transposed
// weights).  The producer's BatchNorm-apply + ReLU (unet.py:73-74) is fused into the input staging, and the
// statistics the following train-mode BatchNorm needs are produced in the epilogue (Chan partials per tile).
//
// Mapping (one workgroup = one TH x TW tile of output pixels of one image x a block of output channels):
//   GEMM  D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (tap, cin)  -- weights are the MFMA "A" operand so a lane
//   ends up with 4 CONSECUTIVE output channels of one pixel (8/16-byte NHWC stores, per-channel sums on 16 lanes).
//   * input halo tile (TH+2)x(TW+2) x KC channels staged once in LDS, [pixel][channel] with a padded pixel stride
//     (conflict-free ds_read_b128); the 9 taps are address offsets into it -- no im2col buffer anywhere.
//   * weight fragments are pre-packed in MFMA lane order (spcl_conv_pack_weights): one coalesced 1 KiB
//     global_load_dwordx4 per (k-step, 16-cout tile), L2-resident, no LDS.
//   * bf16: v_mfma_f32_16x16x32_bf16 (8 bf16 / lane / operand);  f32: 4x v_mfma_f32_16x16x4_f32 per 16-byte chunk
//     (exact-f32, the parity path).
#include <stdlib.h>
#include "bn_acc.hpp"
#include "conv_common.hpp"
#include "image_acorr.hpp"

namespace spcl {

// ---- f32 storage on the bf16 matrix rate (SPLIT): every f32 operand is the exact sum of three bf16 pieces
// (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): 3 x 8 mantissa bits), the product of two operands is six
// bf16 products (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; the three dropped ones are below 2^-24 of the product) summed
// in the MFMA's f32 accumulator: 6 v_mfma_f32_16x16x32_bf16 per 32 k-values instead of 8 v_mfma_f32_16x16x4_f32 at 1/16
// of the rate -- 3/8 of the matrix time, f32-grade results (tests/test_gpu_kernels.py: the f32 tolerances unchanged).  The
// input is split ONCE per halo element in the staging (three bf16 planes per LDS pixel; each element is then read by nine
// taps), the weights once per step by the pack kernel (pack_value<float>: second half of the packed buffer).
// channels per slab of the split layout (its own: the split half of a packed buffer is laid out for it)
#ifndef SPCL_SPLIT_KC_MAX
#define SPCL_SPLIT_KC_MAX 64
#endif
// ... fewer for the layers with one or two 16-channel output tiles: their workgroups are one or two waves, and with a
// 64-channel image (60 KB) two of them fill a CU's LDS -- one wave per SIMD.  Same box, fp32 step: 32 -> 64 channels' dgrad
// (K = 64, 32 outputs) 73.6 -> 58.1 us on 32-channel slabs; 16 -> 32's dgrad (K = 32, 16 outputs) 88.4 -> 70.1 on 16-channel
// slabs; 32 outputs on 16-channel slabs lose (101 -> 115): twice the barriers for little residency.
__host__ __device__ inline int split_kc(int CinK, int CoutS) {
  const int cap = CoutS <= 16 ? 16 : (CoutS <= 32 ? 32 : SPCL_SPLIT_KC_MAX);
  return CinK < cap ? CinK : cap;
}
__host__ __device__ inline int split_pstride(int KC) {  // LDS bytes per halo pixel: three bf16 planes, an ODD multiple of 32
  const int b = 3 * KC * 2;
  return (b / 32) % 2 == 1 ? b : b + 32;
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 w, u32x4 x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// One workgroup = one TH x TW pixel tile (x a block of output channels); see the file header.  Instruction budget
// (ISA audit): staging walks the halo with incremental coordinates (no divisions in the loop, no bounds
// tests on interior tiles, BN coefficients of the thread's fixed channel chunk in registers); the epilogue walks the
// pixels incrementally and accumulates the BatchNorm sums in one pass, reduced across lanes with DPP.
// SPLIT (T = float only): the k-loop of the bf16 kernel on three planes, see above; a.wp points at the split half of the weights
template <typename T, int TH, int TW, int NT, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
  static_assert(!SPLIT || sizeof(T) == 4, "the split k-loop is the f32 storage path");
  constexpr int EPC = SPLIT ? 8 : Chunk<T>::EPC;  // channels per k-chunk (SPLIT: two 16-byte loads, one 16-byte chunk per plane)
  constexpr int NPIX = TH * TW;
  constexpr int MT = (NPIX + 15) / 16;
  constexpr int HW_ = TW + 2;
  constexpr int NHALO = (TH + 2) * HW_;
  constexpr int ESZ = (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int KC = SPLIT ? split_kc(a.CinK, a.CoutS) : conv_kc(a.CinK);
  const int CP = KC / EPC;
  const int log2cp = __builtin_ctz(CP);
  const int PSTRIDE = SPLIT ? split_pstride(KC) : conv_pstride<T>(KC);
  const int nsteps = SPLIT ? conv_nsteps<bf16_t>(KC) : conv_nsteps<T>(KC);
  const int PLANE = KC * 2;  // SPLIT: bytes between the planes of a pixel
  const int nslab = a.CinK / KC;
  const int ntiles_n = a.CoutS >> 4;

  const int tpi = a.tilesX * a.tilesY;
  const int ntiles = a.N * tpi;
  const int nt0 = (blockIdx.y * nwaves + wave) * NT;
  const bool wave_active = nt0 < ntiles_n;     // uniform per wave
  const int nvalid = min(NT, ntiles_n - nt0);  // n-tiles of this wave that exist (CoutS/16 may be odd)

  // per-lane halo base address of each m-tile's pixel (p = 16 i + r16)
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = 16 * i + r16;
    if (p >= NPIX) p = 0;
    const int py = p / TW, px = p - py * TW;
    abase[i] = (py * HW_ + px) * PSTRIDE;
  }
  // staging: the thread's channel chunk is fixed (blockDim % CP == 0); its halo pixel advances by QS per iteration
  const int sch = threadIdx.x & (CP - 1);
  const int sq0 = threadIdx.x >> log2cp;
  const int QS = (int)blockDim.x >> log2cp;
  const int dhy = QS / HW_, dhx = QS - dhy * HW_;

#pragma unroll 1
  for (int rep = 0; rep < a.tpw; ++rep) {
  const int tile = blockIdx.x * a.tpw + rep;
  if (tile >= ntiles) break;
  const int n = tile / tpi;
  const int trem = tile - n * tpi;
  const int ty = trem / a.tilesX, tx = trem - ty * a.tilesX;
  const int y0 = ty * TH, x0 = tx * TW;
  const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + TW < a.W;  // whole halo inside the image

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const u32x4* wp = (const u32x4*)a.wp;

  for (int slab = 0; slab < nslab; ++slab) {
    __syncthreads();
    // ---------------- stage the halo tile of this channel slab (fused BN-apply + ReLU of the producer)
    if (a.in_mode != 2) {
      float ssc[EPC], ssh[EPC];
      if (a.in_mode == 1) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          *(f32x4*)&ssc[e] = *(const f32x4*)(a.in_scale + slab * KC + sch * EPC + e);
          *(f32x4*)&ssh[e] = *(const f32x4*)(a.in_shift + slab * KC + sch * EPC + e);
        }
      }
      // element offsets relative to the halo origin (y0-1, x0-1); dereferenced only when inside the image
      const T* xb = (const T*)a.x + (((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.CinS + slab * KC + sch * EPC;
      int hy = sq0 / HW_, hx = sq0 - hy * HW_;
      int goff = (hy * a.W + hx) * a.CinS;
      const int dgoff = (dhy * a.W + dhx) * a.CinS, wrapg = (a.W - HW_) * a.CinS;
      unsigned char* lp = lds + sq0 * PSTRIDE + sch * 16;
      const int dlp = QS * PSTRIDE;
      if constexpr (SPLIT) {
        // four halo elements per trip, their loads UNCONDITIONAL (a pixel outside the image reads the clamped one and is
        // zeroed afterwards) and issued before the first is used: one element per trip behind `if (inb)` was one exposed
        // round trip to memory per element -- the narrow layers' workgroups are ONE wave, nothing else covers it
        constexpr int U = 4;
        const float* xin = (const float*)a.x + (long)n * a.H * a.W * a.CinS + slab * KC + sch * EPC;
        for (int q = sq0; q < NHALO; q += U * QS) {
          u32x4 v0[U], v1[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            ok[u] = q + u * QS < NHALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
            const float* src = xin + (cy * a.W + cx) * a.CinS;
            v0[u] = *(const u32x4*)src;
            v1[u] = *(const u32x4*)(src + 4);
            hx += dhx;
            hy += dhy;
            if (hx >= HW_) {
              hx -= HW_;
              hy += 1;
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (q + u * QS < NHALO) {  // (the last trip is mostly past the end: whole waves skip the ~90 instructions)
              if (a.in_mode == 1) {
                v0[u] = bnrelu_regs<float>(v0[u], ssc, ssh);
                v1[u] = bnrelu_regs<float>(v1[u], ssc + 4, ssh + 4);
              }
              float e[8];
              *(f32x4*)&e[0] = __builtin_bit_cast(f32x4, v0[u]);
              *(f32x4*)&e[4] = __builtin_bit_cast(f32x4, v1[u]);
              u32x4 ph, pm, pl;
              split3_chunk(e, ph, pm, pl);
              if (!ok[u]) ph = pm = pl = (u32x4){0u, 0u, 0u, 0u};
              *(u32x4*)lp = ph;
              *(u32x4*)(lp + PLANE) = pm;
              *(u32x4*)(lp + 2 * PLANE) = pl;
            }
            lp += dlp;
          }
        }
      } else
      for (int q = sq0; q < NHALO; q += QS) {
        bool inb = true;
        if (!interior) {
          const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
          inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (inb && !(a.dbg & 16)) {
            v = *(const u32x4*)(xb + goff);
            if (a.in_mode == 1 && !(a.dbg & 1)) v = bnrelu_regs<T>(v, ssc, ssh);
          }
          *(u32x4*)lp = v;
        }
        lp += dlp;
        hx += dhx;
        hy += dhy;
        goff += dgoff;
        if (hx >= HW_) {
          hx -= HW_;
          hy += 1;
          goff += wrapg;
        }
      }
    } else if constexpr (SPLIT) {
      // the image convolution (CinS <= 16 real channels): four elements per trip, unconditional clamped loads as above
      constexpr int U = 4;
      const float* xin = (const float*)a.x + (size_t)n * a.H * a.W * a.CinS;
      for (int idx0 = threadIdx.x; idx0 < NHALO * CP; idx0 += U * (int)blockDim.x) {
        float e[U][EPC];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = idx0 + u * (int)blockDim.x;
          const int q = idx >> log2cp, ch = idx & (CP - 1);
          const int hy = q / HW_, hx = q - hy * HW_;
          const int gy = y0 + hy - 1, gx = x0 + hx - 1;
          ok[u] = idx < NHALO * CP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          const float* src = xin + (min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1)) * a.CinS;
#pragma unroll
          for (int k = 0; k < EPC; ++k) {
            e[u][k] = 0.f;
            if (k < a.CinS) {  // (uniform)
              const int c = ch * EPC + k;
              const float v = src[min(c, a.CinS - 1)];
              e[u][k] = c < a.CinS ? v : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = idx0 + u * (int)blockDim.x;
          const int q = idx >> log2cp, ch = idx & (CP - 1);
          u32x4 ph, pm, pl;
          split3_chunk(e[u], ph, pm, pl);
          if (!ok[u]) ph = pm = pl = (u32x4){0u, 0u, 0u, 0u};
          if (idx < NHALO * CP) {
            *(u32x4*)(lds + q * PSTRIDE + ch * 16) = ph;
            *(u32x4*)(lds + q * PSTRIDE + ch * 16 + PLANE) = pm;
            *(u32x4*)(lds + q * PSTRIDE + ch * 16 + 2 * PLANE) = pl;
          }
        }
      }
    } else {
      for (int idx = threadIdx.x; idx < NHALO * CP; idx += blockDim.x) {
        const int q = idx >> log2cp, ch = idx & (CP - 1);
        const int hy = q / HW_, hx = q - hy * HW_;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
          const float* src = (const float*)a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.CinS;
          float e[EPC];
#pragma unroll
          for (int k = 0; k < EPC; ++k) {
            const int c = ch * EPC + k;
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
        *(u32x4*)(lds + q * PSTRIDE + ch * 16) = v;
      }
    }
    __syncthreads();
    if (!wave_active) continue;

    // ---------------- K loop over (tap, channel-chunk) steps; 4 chunks (k-groups g) per step
    const u32x4* wslab = wp + ((size_t)slab * nsteps * ntiles_n + nt0) * 64 + lane;
    if constexpr (SPLIT) {
      // three weight planes (hi, mid, lo: `wplane` fragments apart), three input planes per LDS pixel; products from the
      // smallest to the largest
      const size_t wplane = (size_t)nslab * nsteps * ntiles_n * 64;
      u32x4 wf[3][NT], wnext[3][NT];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[pl][j] = wslab[pl * wplane + (size_t)(j < nvalid ? j : 0) * 64];
      // the input fragments run ONE m-tile ahead of the products (across the step boundary too): with one wave per SIMD --
      // the wide layers' grids -- nothing else hides the LDS latency
      auto xoff = [&](int s) {
        if (CP >= 4) {
          const int fc0 = 4 * s;
          const int tap = fc0 >> log2cp, ch0 = fc0 & (CP - 1);
          const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
          return (ky * HW_ + kx) * PSTRIDE + (ch0 + g) * 16;
        }
        int fc = 4 * s + g;
        if (fc >= 9 * CP) fc = 0;  // K padding: weights there are zero
        const int tap = fc >> log2cp, ch = fc & (CP - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        return (ky * HW_ + kx) * PSTRIDE + ch * 16;
      };
      int off = xoff(0);
      u32x4 xh = *(const u32x4*)(lds + abase[0] + off);
      u32x4 xm = *(const u32x4*)(lds + abase[0] + off + PLANE);
      u32x4 xl = *(const u32x4*)(lds + abase[0] + off + 2 * PLANE);
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        // (unconditional -- the last step re-reads its own fragments -- and fenced: behind a branch the requests sat at the END
        // of the previous trip and every step began with a wait for the L2)
        const int sn = s + 1 < nsteps ? s + 1 : s;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            wnext[pl][j] = wslab[pl * wplane + ((size_t)sn * ntiles_n + (j < nvalid ? j : 0)) * 64];
        const int offn = xoff(sn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int an = i + 1 < MT ? abase[i + 1 < MT ? i + 1 : 0] + off : abase[0] + offn;
          const u32x4 nh = *(const u32x4*)(lds + an);
          const u32x4 nm = *(const u32x4*)(lds + an + PLANE);
          const u32x4 nl = *(const u32x4*)(lds + an + 2 * PLANE);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[2][j], xh, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[0][j], xl, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[1][j], xm, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[1][j], xh, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[0][j], xm, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(wf[0][j], xh, acc[i][j]);
          xh = nh;
          xm = nm;
          xl = nl;
        }
        off = offn;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int j = 0; j < NT; ++j) wf[pl][j] = wnext[pl][j];
      }
      continue;
    }
    u32x4 wf[NT], wnext[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = wslab[(size_t)(j < nvalid ? j : 0) * 64];
#pragma unroll 1
    for (int s = 0; s < ((a.dbg & 2) ? 0 : nsteps); ++s) {
      if (s + 1 < nsteps) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wnext[j] = wslab[((size_t)(s + 1) * ntiles_n + (j < nvalid ? j : 0)) * 64];
      }
      int off;
      if (CP >= 4) {
        // the step's 4 k-groups lie in ONE tap: the tap part is wave-uniform (scalar ALU), the lane adds its chunk
        const int fc0 = 4 * s;
        const int tap = fc0 >> log2cp, ch0 = fc0 & (CP - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        off = (ky * HW_ + kx) * PSTRIDE + (ch0 + g) * 16;
      } else {
        int fc = 4 * s + g;
        if (fc >= 9 * CP) fc = 0;  // K padding: weights there are zero, any finite x will do
        const int tap = fc >> log2cp, ch = fc & (CP - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        off = (ky * HW_ + kx) * PSTRIDE + ch * 16;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 xf = *(const u32x4*)(lds + abase[i] + off);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_chunk<T>(wf[j], xf, acc[i][j]);
      }
      if (s + 1 < nsteps) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wnext[j];
      }
    }
  }
  if (!wave_active) continue;

  // ---------------- epilogue: lane holds couts 16(nt0+j)+4g..+3 of pixel p = 16 i + r16, walked incrementally
  constexpr int DPY = 16 / TW, DPX = 16 % TW;
  const bool full_tile = y0 + TH <= a.H && x0 + TW <= a.W;
  int r16e = r16;  // opaque per tile: keeps (py, px, ob) out of the registers live across the k-loop
  asm volatile("" : "+v"(r16e));
  int py = r16e / TW, px = r16e - py * TW;
  const int rowb = a.CoutS * ESZ;
  int ob = (py * a.W + px) * rowb + (nt0 * 16 + 4 * g) * ESZ;
  const int dob = (DPY * a.W + DPX) * rowb, wrapo = (a.W - TW) * rowb;
  unsigned char* yb = (unsigned char*)a.y + (((size_t)n * a.H + y0) * a.W + x0) * rowb;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ssq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    bool ok = (16 * i + 15 < NPIX) || (16 * i + r16e < NPIX);  // compile-time true except in the last m-tile
    if (!full_tile) ok = ok && (y0 + py) < a.H && (x0 + px) < a.W;
    if (ok) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j < nvalid && !(a.dbg & 8)) store4_fast<T>(yb + ob + j * 16 * ESZ, acc[i][j]);
        ssum[j] += acc[i][j];
        ssq[j] += acc[i][j] * acc[i][j];
      }
    }
    px += DPX;
    py += DPY;
    ob += dob;
    if (px >= TW) {
      px -= TW;
      py += 1;
      ob += wrapo;
    }
  }
  if (a.stats != nullptr && !(a.dbg & 4)) {
    const int vh = min(TH, a.H - y0), vw = min(TW, a.W - x0);
    const float cnt = (float)(vh * vw);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j >= nvalid) break;
      write_tile_stats(a.stats, tile, a.CoutS, (nt0 + j) * 16 + 4 * g, r16, cnt, ssum[j], ssq[j]);
    }
  }
  }  // tiles of this workgroup
}

// The image convolution in f32 storage (unet.py:123, input_dim <= 4 -> 16 channels) as a DIRECT convolution on the vector
// ALU: K = 9 CinS, so the matrix kernels above spent 16 x 9 k-values (and, split, six products each) on at most 36 real ones
// and a one-wave workgroup's staging + k-loop + epilogue chain was 135 us at 64 x 224^2 for a 13 MB read and a 206 MB write.
// Exact f32 FMAs.  One wave per pixel tile, the accumulator layout of the matrix kernels (lane = (pixel 16 i + r16, output
// channels 4 g .. 4 g + 3)): same stores, same statistics rows.  Weights from the exact-f32 half of the packed buffer
// (pack_value: W[co][ci][tap] of ci < 4 sits at packed[(tap * 64 + co) * 4 + ci] when CinK == CoutS == 16).
template <int TH, int TW>
__global__ __launch_bounds__(64) void conv3x3_image_f32_kernel(ConvArgs a) {
  constexpr int NPIX = TH * TW, MT = (NPIX + 15) / 16, HW_ = TW + 2, NHALO = (TH + 2) * HW_;
  __shared__ float halo[4][NHALO];
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  const int tpi = a.tilesX * a.tilesY;
  const int tile = blockIdx.x;
  const int n = tile / tpi;
  const int trem = tile - n * tpi;
  const int ty = trem / a.tilesX, tx = trem - ty * a.tilesX;
  const int y0 = ty * TH, x0 = tx * TW;
  // ---- halo: unconditional clamped loads, all in flight before the first is used
  {
    const float* xin = (const float*)a.x + (size_t)n * a.H * a.W * a.CinS;
    constexpr int NQ = (NHALO + 63) / 64;
    for (int c = 0; c < a.CinS; ++c) {
      float v[NQ];
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const int q = lane + 64 * k;
        const int hy = q / HW_, hx = q - hy * HW_;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const float t = xin[(min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1)) * a.CinS + c];
        v[k] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? t : 0.f;
      }
#pragma unroll
      for (int k = 0; k < NQ; ++k)
        if (lane + 64 * k < NHALO) halo[c][lane + 64 * k] = v[k];
    }
  }
  __syncthreads();
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int hb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = 16 * i + r16;
    if (p >= NPIX) p = 0;
    const int py = p / TW, px = p - py * TW;
    hb[i] = py * HW_ + px;
  }
  const float* wp = (const float*)a.wp;
  for (int c = 0; c < a.CinS; ++c) {
    float w[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) w[t][r] = wp[(t * 64 + 4 * g + r) * 4 + c];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float xv = halo[c][hb[i] + (t / 3) * HW_ + (t % 3)];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = fmaf(w[t][r], xv, acc[i][r]);
      }
    }
  }
  // ---- epilogue: the matrix kernels' (stores + Chan partials of the tile)
  constexpr int DPY = 16 / TW, DPX = 16 % TW;
  const bool full_tile = y0 + TH <= a.H && x0 + TW <= a.W;
  int py = r16 / TW, px = r16 - py * TW;
  const int rowb = a.CoutS * 4;
  int ob = (py * a.W + px) * rowb + 4 * g * 4;
  const int dob = (DPY * a.W + DPX) * rowb, wrapo = (a.W - TW) * rowb;
  unsigned char* yb = (unsigned char*)a.y + (((size_t)n * a.H + y0) * a.W + x0) * rowb;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = ssum;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    bool ok = (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);
    if (!full_tile) ok = ok && (y0 + py) < a.H && (x0 + px) < a.W;
    if (ok) {
      *(f32x4*)(yb + ob) = acc[i];
      ssum += acc[i];
      ssq += acc[i] * acc[i];
    }
    px += DPX;
    py += DPY;
    ob += dob;
    if (px >= TW) {
      px -= TW;
      py += 1;
      ob += wrapo;
    }
  }
  if (a.stats != nullptr) {
    const int vh = min(TH, a.H - y0), vw = min(TW, a.W - x0);
    write_tile_stats(a.stats, tile, a.CoutS, 4 * g, r16, (float)(vh * vw), ssum, ssq);
  }
}

// ---------------------------------------------------------------------------------------------- weight packing
// packed[slab][step][ntile][lane][EPC]: lane (r16 = cout within tile, g = k-group) holds the EPC channels of
// flattened chunk fc = 4*step + g  (tap = fc / CP, channel chunk = fc % CP) of slab `slab`.
// kind 0 (forward):  A[cout=o][k=(tap,ci)]      = W[o][ci][tap]
// kind 1 (dgrad):    A[cout=ci][k=(tap,co)]     = W[co][ci][8-tap]        (roles of Cin/Cout swapped by the caller)
// f32 buffers hold TWO layouts: the exact-f32 kernel's fragments first (f32_exact_elems floats), then the split kernel's --
// packedS[plane 3][slab][step][ntile][lane][8] bf16 pieces in the bf16 kernel's fragment order, two pieces per float slot
__host__ __device__ inline size_t f32_exact_elems(int KinK, int NoutS) {
  const int KC = conv_kc(KinK);
  return (size_t)(KinK / KC) * conv_nsteps<float>(KC) * (NoutS / 16) * 64 * 4;
}
__host__ __device__ inline size_t f32_split_plane_elems(int KinK, int NoutS) {  // bf16 pieces per plane
  const int KC = split_kc(KinK, NoutS);
  return (size_t)(KinK / KC) * conv_nsteps<bf16_t>(KC) * (NoutS / 16) * 64 * 8;
}
template <int EPC>
__device__ __forceinline__ float pack_value_epc(const float* __restrict__ w, int Cin, int Cout, int kind, int KC, int NoutS,
                                                unsigned r);
__device__ __forceinline__ uint32_t split_piece(float v, int plane) {  // bf16 bits of piece `plane` of v (split3_pair's pieces)
  const Split3 o = split3_pair(v, 0.f);
  return (plane == 0 ? o.hi : (plane == 1 ? o.mid : o.lo)) & 0xffffu;
}

template <typename T>
__device__ __forceinline__ float pack_value(const float* __restrict__ w, int Cin, int Cout, int kind, int KinK, int NoutS,
                                            size_t idx, bool gemm) {
  constexpr int EPC = Chunk<T>::EPC;
  if (sizeof(T) == 4 && idx >= f32_exact_elems(KinK, NoutS)) {
    const unsigned per = (unsigned)f32_split_plane_elems(KinK, NoutS);
    const unsigned b = 2u * (unsigned)(idx - f32_exact_elems(KinK, NoutS));  // first of the slot's two pieces
    const int plane = (int)(b / per);
    const unsigned r = b - (unsigned)plane * per;
    const uint32_t p0 = split_piece(pack_value_epc<8>(w, Cin, Cout, kind, split_kc(KinK, NoutS), NoutS, r), plane);
    const uint32_t p1 = split_piece(pack_value_epc<8>(w, Cin, Cout, kind, split_kc(KinK, NoutS), NoutS, r + 1), plane);
    return __uint_as_float(p0 | (p1 << 16));
  }
  if (gemm && idx >= (size_t)9 * KinK * NoutS) {
    // second half of a dual-layout buffer, for conv_gemm.hip: packed[slab][tap][32-channel output tile][ks][lane][8]:
    // lane (n32 = output channel in the tile, kh = lane / 32) holds input channels 64 slab + 16 ks + 8 kh .. +7 of tap
    // `tap` -- one 1 KiB piece per A fragment
    unsigned r = (unsigned)(idx - (size_t)9 * KinK * NoutS);  // (32-bit: a 64-bit division by a run-time value is ~150 instructions,
                                                            // and this function did five per element -- the pack launch was 12.9 us)
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % 4); r /= 4;
    const int cot = (int)(r % (NoutS >> 5)); r /= (NoutS >> 5);
    const int tap = (int)(r % 9); r /= 9;
    const int slab = (int)r;
    const int kch = slab * 64 + ks * 16 + (lane >> 5) * 8 + e, nch = cot * 32 + (lane & 31);
    if (kind == 0) return (kch < Cin && nch < Cout) ? w[((size_t)nch * Cin + kch) * 9 + tap] : 0.f;
    return (kch < Cout && nch < Cin) ? w[((size_t)kch * Cin + nch) * 9 + (8 - tap)] : 0.f;
  }
  return pack_value_epc<EPC>(w, Cin, Cout, kind, conv_kc(KinK), NoutS, (unsigned)idx);  // (packed buffers hold a few million elements)
}

template <int EPC>
__device__ __forceinline__ float pack_value_epc(const float* __restrict__ w, int Cin, int Cout, int kind, int KC, int NoutS,
                                                unsigned r) {  // KC: channels per slab of the layout
  const int CP = KC / EPC;
  const int nsteps = (9 * CP + 3) / 4;  // conv_nsteps
  const int ntn = NoutS >> 4;
  const int e = (int)(r % EPC); r /= EPC;
  const int lane = (int)(r % 64); r /= 64;
  const int nt = (int)(r % ntn); r /= ntn;
  const int step = (int)(r % nsteps); r /= nsteps;
  const int slab = (int)r;
  const int r16 = lane & 15, g = lane >> 4;
  const int fc = 4 * step + g;
  float v = 0.f;
  if (fc < 9 * CP) {
    const int tap = fc / CP, ch = fc % CP;
    const int kch = slab * KC + ch * EPC + e;  // GEMM-K channel
    const int nch = nt * 16 + r16;             // GEMM-N (output) channel
    if (kind == 0) {
      if (kch < Cin && nch < Cout) v = w[((size_t)nch * Cin + kch) * 9 + tap];
    } else {
      if (kch < Cout && nch < Cin) v = w[((size_t)kch * Cin + nch) * 9 + (8 - tap)];
    }
  }
  return v;
}

// ---- the same layouts 16 BYTES at a time (the step's pack launch, conv_pack_multi_body): the eight (bf16) or four (f32)
// elements of a 16-byte chunk share everything but the GEMM-K channel, so one index decode -- five divisions by run-time
// values -- and one 16-byte store serve them all (per element the launch was decode-bound: 13 us for 5 MB, 37 us in f32
// storage where every float slot of the split layout decoded twice and split two weights into all three pieces to keep one)
__device__ __forceinline__ void pack_gather(const float* __restrict__ w, int Cin, int Cout, int kind, int kch0, int nch, int tap,
                                            int n, float* v) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = 0.f;
    if (e < n) {
      const int kch = kch0 + e;
      if (kind == 0) {
        if (kch < Cin && nch < Cout) v[e] = w[((size_t)nch * Cin + kch) * 9 + tap];
      } else {
        if (kch < Cout && nch < Cin) v[e] = w[((size_t)kch * Cin + nch) * 9 + (8 - tap)];
      }
    }
  }
}
// the standard layout's chunk that starts at element r (a multiple of EPC): its EPC values (zeros in the K padding)
template <int EPC>
__device__ __forceinline__ void pack_chunk_values(const float* __restrict__ w, int Cin, int Cout, int kind, int KC, int NoutS,
                                                  unsigned r, float* v) {
  const int CP = KC / EPC;
  const int nsteps = (9 * CP + 3) / 4;
  const int ntn = NoutS >> 4;
  r /= EPC;
  const int lane = (int)(r % 64); r /= 64;
  const int nt = (int)(r % ntn); r /= ntn;
  const int step = (int)(r % nsteps); r /= nsteps;
  const int slab = (int)r;
  const int fc = 4 * step + (lane >> 4);
  if (fc < 9 * CP) {
    const int tap = fc / CP, ch = fc % CP;
    pack_gather(w, Cin, Cout, kind, slab * KC + ch * EPC, nt * 16 + (lane & 15), tap, EPC, v);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float* v) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (uint32_t)f32_to_bf16(v[2 * k]) | ((uint32_t)f32_to_bf16(v[2 * k + 1]) << 16);
  return o;
}
// bytes [16 c, 16 c + 16) of the packed buffer
template <typename T>
__device__ __forceinline__ u32x4 pack_chunk16(const float* __restrict__ w, int Cin, int Cout, int kind, int KinK, int NoutS,
                                              unsigned c, bool gemm) {
  float v[8];
  if constexpr (sizeof(T) == 4) {
    const unsigned r0 = 4u * c, exact = (unsigned)f32_exact_elems(KinK, NoutS);
    if (r0 < exact) {
      pack_chunk_values<4>(w, Cin, Cout, kind, conv_kc(KinK), NoutS, r0, v);
      return (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    }
    const unsigned per = (unsigned)f32_split_plane_elems(KinK, NoutS);
    const unsigned b0 = 2u * (r0 - exact);  // the chunk's first bf16 piece: eight pieces of ONE plane
    const int plane = (int)(b0 / per);
    pack_chunk_values<8>(w, Cin, Cout, kind, split_kc(KinK, NoutS), NoutS, b0 - (unsigned)plane * per, v);
    u32x4 ph, pm, pl;
    split3_chunk(v, ph, pm, pl);
    return plane == 0 ? ph : (plane == 1 ? pm : pl);
  } else {
    const unsigned r0 = 8u * c;
    if (gemm && r0 >= 9u * (unsigned)KinK * (unsigned)NoutS) {  // conv_gemm.hip's layout (pack_value)
      unsigned r = (r0 - 9u * (unsigned)KinK * (unsigned)NoutS) / 8;
      const int lane = (int)(r % 64); r /= 64;
      const int ks = (int)(r % 4); r /= 4;
      const int cot = (int)(r % (NoutS >> 5)); r /= (NoutS >> 5);
      const int tap = (int)(r % 9); r /= 9;
      const int slab = (int)r;
      pack_gather(w, Cin, Cout, kind, slab * 64 + ks * 16 + (lane >> 5) * 8, cot * 32 + (lane & 31), tap, 8, v);
    } else {
      pack_chunk_values<8>(w, Cin, Cout, kind, conv_kc(KinK), NoutS, r0, v);
    }
    return pack_bf16x8(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int kind,
                                                        int KinK, int NoutS, T* __restrict__ packed, size_t total,
                                                        bool gemm) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  Elem<T>::store(packed + idx, pack_value<T>(w, Cin, Cout, kind, KinK, NoutS, idx, gemm));
}

// forward (kind 0) and dgrad (kind 1) layouts of one layer in one launch
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_both_kernel(const float* __restrict__ w, int Cin, int Cout, int CinK,
                                                             int CoutS, T* __restrict__ p0, size_t t0,
                                                             T* __restrict__ p1, size_t t1, bool gemm) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < t0) Elem<T>::store(p0 + idx, pack_value<T>(w, Cin, Cout, 0, CinK, CoutS, idx, gemm));
  else if (idx < t0 + t1) Elem<T>::store(p1 + idx - t0, pack_value<T>(w, Cin, Cout, 1, CoutS, CinK, idx - t0, gemm));
}

// the two convolutions of a block (forward + dgrad layouts each) in one launch
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_block_kernel(const float* __restrict__ wa, int CinA, int CoutA,
                                                              T* __restrict__ a0, size_t ta0, T* __restrict__ a1,
                                                              size_t ta1, const float* __restrict__ wb, int CinB,
                                                              int CoutB, T* __restrict__ b0, size_t tb0,
                                                              T* __restrict__ b1, size_t tb1, bool gemm_a0,
                                                              bool gemm_a1, bool gemm_b0, bool gemm_b1) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int CinKA = (CinA + 15) / 16 * 16, CoutSA = (CoutA + 15) / 16 * 16;
  const int CinKB = (CinB + 15) / 16 * 16, CoutSB = (CoutB + 15) / 16 * 16;
  if (idx < ta0) { Elem<T>::store(a0 + idx, pack_value<T>(wa, CinA, CoutA, 0, CinKA, CoutSA, idx, gemm_a0)); return; }
  idx -= ta0;
  if (idx < ta1) { Elem<T>::store(a1 + idx, pack_value<T>(wa, CinA, CoutA, 1, CoutSA, CinKA, idx, gemm_a1)); return; }
  idx -= ta1;
  if (idx < tb0) { Elem<T>::store(b0 + idx, pack_value<T>(wb, CinB, CoutB, 0, CinKB, CoutSB, idx, gemm_b0)); return; }
  idx -= tb0;
  if (idx < tb1) Elem<T>::store(b1 + idx, pack_value<T>(wb, CinB, CoutB, 1, CoutSB, CinKB, idx, gemm_b1));
}

// forward + dgrad layouts of many layers in one launch (spcl_conv_pack_weights_multi): 2 segments per layer, every
// workgroup inside ONE segment (segments are rounded up to whole workgroups of PACK_EPB elements).  The workgroup finds its
// segment from the table of last-block indices with statically indexed compares (the table arrives with the kernel
// arguments in a few wide scalar loads; a per-thread walk over dynamically indexed argument structs was a chain of 2 x 20
// dependent scalar loads: 20 us for what five separate launches did in 27).
#ifndef SPCL_PACK_EPB
#define SPCL_PACK_EPB 2048
#endif
constexpr int PACK_EPB = SPCL_PACK_EPB, PACK_SEGS = 2 * SPCL_PACK_MULTI_MAX + 1;  // (+ 1: the zero-fill segment, kind 2)
struct PackSeg {
  const float* w;
  void* out;
  unsigned count;  // elements of this segment
  int blk_begin;   // its first workgroup
  int Cin, Cout, kind, KinK, NoutS, gemm;
};
struct PackSegs {
  int blk_end[PACK_SEGS];  // exclusive; INT_MAX beyond the last segment
  PackSeg s[PACK_SEGS];
};
template <typename T>
__device__ __forceinline__ void conv_pack_multi_body(const PackSegs& p, const int b);

// ... and, in the SAME launch, the image autocorrelation of the "image3" path (bn.hip): both are a dozen microseconds of
// work that fills a fraction of the chip, neither depends on the other, and as one launch they overlap and pay one launch
// floor (26 us back to back).  The pack workgroups come first: they are short, and every workgroup of the launch reserves
// the autocorrelation's LDS, so whichever job is dispatched first holds the CUs' slots until it retires.
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_multi_acorr_kernel(PackSegs p, const float* __restrict__ img, int H, int W,
                                                                   float* __restrict__ acorr, int npack) {
  if ((int)blockIdx.x < npack) conv_pack_multi_body<T>(p, blockIdx.x);
  else image_autocorr_body(img, H, W, acorr, blockIdx.x - npack);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(PackSegs p) {
  conv_pack_multi_body<T>(p, blockIdx.x);
}

template <typename T>
__device__ __forceinline__ void conv_pack_multi_body(const PackSegs& p, const int b) {
  int seg = 0;
#pragma unroll
  for (int k = 0; k < PACK_SEGS; ++k) seg += b >= p.blk_end[k] ? 1 : 0;
  const PackSeg g = p.s[seg];
  constexpr int EPCH = 16 / (int)sizeof(T), CHUNKS = PACK_EPB / EPCH;  // elements per 16-byte chunk, chunks per workgroup
  const unsigned c0 = (unsigned)(b - g.blk_begin) * CHUNKS;
#pragma unroll
  for (int k = 0; k < (CHUNKS + 255) / 256; ++k) {
    const unsigned c = c0 + threadIdx.x + 256 * k;
    const unsigned i = c * EPCH;  // the chunk's first element
    if (CHUNKS % 256 != 0 && threadIdx.x + 256 * k >= CHUNKS) break;
    if (i >= g.count) continue;
    if (g.kind == 2) {
      // not a weight layout at all -- a region to ZERO (the step's BatchNorm accumulator blocks, bn_acc.hpp: they must be zero
      // before the first convolution runs, and this launch is the first of every forward pass: no fill launch)
      for (int e = 0; e < EPCH && i + e < g.count; ++e) Elem<T>::store((T*)g.out + i + e, 0.f);
    } else {  // (a layout's element count is a multiple of 256: whole chunks)
      *(u32x4*)((T*)g.out + i) = pack_chunk16<T>(g.w, g.Cin, g.Cout, g.kind, g.KinK, g.NoutS, c, g.gemm != 0);
    }
  }
}

// bf16 layers the workgroup-level GEMM kernel can take (conv_gemm.hip) carry BOTH layouts, the per-wave kernels' first:
// which kernel runs is decided per launch from the image size (conv_use_gemm), the weights do not know it
template <typename T> static size_t packed_elems(int KinK, int NoutS) {
  const int KC = conv_kc(KinK);
  const size_t one = (size_t)(KinK / KC) * conv_nsteps<T>(KC) * (NoutS / 16) * 64 * Chunk<T>::EPC;
  if (sizeof(T) == 4) return f32_exact_elems(KinK, NoutS) + 3 * f32_split_plane_elems(KinK, NoutS) / 2;  // both f32 layouts
  return (sizeof(T) == 2 && conv_gemm_channels(KinK, NoutS)) ? 2 * one : one;
}

struct TileCfg { int th, tw; };
static TileCfg pick_tile(int H, int W) {
  // measured (tools/bench_kernels.py, same box A/B): below 224^2 the half-height tile (twice the waves, half the
  // accumulators per wave) is 10-35 % faster on every layer; at 224^2 the 14x14 tile wins by ~8 %.  7x7 tiles
  // (SPCL_CONV_T77_MAXH=14) beat 7x14 at 14^2 only in the generic kernel; the specialised one is 10-15 % faster on 7x14.
  // (Packing all layers' weights in one launch at the start of forward was measured too: 40 us per step SLOWER than
  // the per-layer pack right before each conv, which leaves the fragments hot in L2 for the waves that fetch them.)
  static const int th7_max_h = lab_env("SPCL_CONV_TH7_MAXH", 112);
  static const int t77_max_h = lab_env("SPCL_CONV_T77_MAXH", 0);
  if (H % 14 == 0 && W % 14 == 0 && H <= t77_max_h) return {7, 7};
  if (H % 14 == 0 && W % 14 == 0 && H <= th7_max_h) return {7, 14};
  if (H % 14 == 0 && W % 14 == 0) return {14, 14};
  // other sizes (256^2 Prostate slices, ...): still the 14-column tiles of the specialised kernels -- their last tile per
  // row / column is shifted back inside the image -- as long as the recomputed overlap stays below a quarter
  const int th = H <= (th7_max_h > 128 ? th7_max_h : 128) ? 7 : 14;  // 128^2 (the 256^2 family's second level) like 112^2
  static const int max_overlap = lab_env("SPCL_CONV_OVERLAP_PCT", 25);
  if (H >= th && W >= 14 && (long)cdiv(H, th) * th * cdiv(W, 14) * 14 * 100 <= (long)H * W * (100 + max_overlap))
    return {th, 14};
  return {16, 16};
}

// ... per layer: a bf16 layer with 32 input channels above 112^2 (the decoder's 224^2 level: cat(16, 16) -> 16 and the
// up-convolution 32 -> 16) takes the half-height tile too: its 14-row halo image is 24.5 KB -- six one-wave workgroups per
// CU -- and those launches are latency chains (fine-tune step 2.357 -> 2.338 ms, same box, four rounds;
// SPCL_CONV_TH7_KC32=0 switches back).  The 16-channel layers' image is 8 KB: they keep the 14 x 14 tile.
static TileCfg pick_tile_k(int H, int W, int CinK, int CoutS = 0) {
  TileCfg t = pick_tile(H, W);
  static const int kc32 = lab_env("SPCL_CONV_TH7_KC32", 1);
  if (kc32 && CinK == 32 && t.th == 14 && t.tw == 14 && H % 7 == 0) t.th = 7;
  // (16 -> 32 at 224^2 -- 104 accumulator registers on the 14-row tile -- measured on 7-row tiles too: no difference)
  (void)CoutS;
  return t;
}

// Which kernel takes a bf16 layer with GEMM-eligible channels: the workgroup-level GEMM kernel where the per-wave
// kernels have no specialisation for the image size (widths that are not 14-column tileable: 32^2, 16^2 of the 256^2
// family, where the generic kernel is 2-3x slower), or everywhere when forced (spcl_conv_set_gemm(1) / SPCL_CONV_GEMM=1:
// at the 14-column sizes it is at parity or slower, profiles/r02_conv_gemm_notes.md); never with 0.
static int g_gemm_mode = -2;  // -2: read the environment on first use; -1 auto; 0 never; 1 always
void conv_set_gemm(int mode) { g_gemm_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0); }
bool conv_use_gemm(int CinK, int CoutS, int H, int W) {
  if (g_gemm_mode == -2) {
    g_gemm_mode = !lab_flag("SPCL_CONV_GEMM") ? -1 : (lab_env("SPCL_CONV_GEMM", 0) > 0 ? 1 : 0);
  }
  if (g_gemm_mode == 0 || !conv_gemm_channels(CinK, CoutS) || !conv_gemm_fits(H, W, CinK, CoutS)) return false;
  return g_gemm_mode == 1 || pick_tile(H, W).tw != 14;
}

// f32 storage: 1 (default) the split-bf16 k-loop, 0 the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 matrix rate);
// the packed buffers carry both layouts, so the switch may change between any two launches (spcl_conv_set_f32_split)
static int g_f32_split = 1;
bool conv_f32_split() { return g_f32_split != 0; }

template <typename T, int TH, int TW>
static int launch_conv(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tilesX = cdiv(a.W, TW);
  a.tilesY = cdiv(a.H, TH);
  const int ntn = a.CoutS / 16;
  const int KC = conv_kc(a.CinK);
  static const int env_lds_extra = lab_env("SPCL_CONV_LDS_EXTRA", 0);
  const bool split = sizeof(T) == 4 && g_f32_split != 0;
  const size_t lds = (size_t)(TH + 2) * (TW + 2) * (split ? split_pstride(split_kc(a.CinK, a.CoutS)) : conv_pstride<T>(KC)) + env_lds_extra;
  const int tiles = a.N * a.tilesX * a.tilesY;
  static const int env_tpw = lab_env("SPCL_CONV_TPW", 0);
  a.tpw = env_tpw > 0 ? env_tpw : 1;
  static const int env_dbg = lab_env("SPCL_CONV_DBG", 0);
  a.dbg = env_dbg;
  // waves per workgroup x n-tiles per wave
  int NT = ntn >= 2 ? 2 : 1;
  // split k-loop: matrix-bound, and ONE wave per SIMD issues a 16x16x32 MFMA every ~23 cycles where two reach the pipe's 16
  // (tools/experiments/mfmalab): two n-tiles per wave only where that still leaves four-wave workgroups, two per CU
  if (split && !(ntn >= 8 && (long)tiles * cdiv(ntn, 8) >= 512)) NT = 1;
  int wn = cdiv(ntn, NT);
  if (wn > 4) wn = 4;
  const int gy = cdiv(ntn, NT * wn);
  dim3 grid(cdiv(tiles, a.tpw), gy), block(64 * wn);
  if constexpr (sizeof(T) == 4) {
    if (a.in_mode == 2 && a.CinS <= 4 && a.CinK == 16 && a.CoutS == 16) {  // (either f32 mode: exact FMAs)
      SPCL_LAUNCH((conv3x3_image_f32_kernel<TH, TW>), dim3(tiles), dim3(64), 0, st, a);
      return 0;
    }
    if (split) {
      a.wp = (const float*)a.wp + f32_exact_elems(a.CinK, a.CoutS);
      if (NT == 1) {
        if (lds > 65536) spcl::func_lds_limit((const void*)conv3x3_mfma_kernel<T, TH, TW, 1, true>, (int)lds, "conv3x3_mfma_kernel<float, TH, TW, 1, split>");
        SPCL_LAUNCH((conv3x3_mfma_kernel<T, TH, TW, 1, true>), grid, block, lds, st, a);
      } else {
        if (lds > 65536) spcl::func_lds_limit((const void*)conv3x3_mfma_kernel<T, TH, TW, 2, true>, (int)lds, "conv3x3_mfma_kernel<float, TH, TW, 2, split>");
        SPCL_LAUNCH((conv3x3_mfma_kernel<T, TH, TW, 2, true>), grid, block, lds, st, a);
      }
      return 0;
    }
  }
  if (NT == 1) {
    if (lds > 65536) spcl::func_lds_limit((const void*)conv3x3_mfma_kernel<T, TH, TW, 1>, (int)lds, "conv3x3_mfma_kernel<T, TH, TW, 1>");
    SPCL_LAUNCH((conv3x3_mfma_kernel<T, TH, TW, 1>), grid, block, lds, st, a);
  } else {
    if (lds > 65536) spcl::func_lds_limit((const void*)conv3x3_mfma_kernel<T, TH, TW, 2>, (int)lds, "conv3x3_mfma_kernel<T, TH, TW, 2>");
    SPCL_LAUNCH((conv3x3_mfma_kernel<T, TH, TW, 2>), grid, block, lds, st, a);
  }
  return 0;
}

template <typename T>
static int launch_conv_t(const ConvArgs& a, hipStream_t st) {
  if (sizeof(T) == 2 && conv_use_gemm(a.CinK, a.CoutS, a.H, a.W)) return launch_conv_gemm(a, st) ? 0 : 1;
  TileCfg t = sizeof(T) == 2 ? pick_tile_k(a.H, a.W, a.CinK, a.CoutS) : pick_tile(a.H, a.W);
  static const int no_fast = lab_env("SPCL_CONV_NO_FAST", 0);
  if (sizeof(T) == 2 && t.tw == 14 && !no_fast && launch_conv_fast(a, t.th, st)) return 0;
  if (t.th == 7 && t.tw == 7) return launch_conv<T, 7, 7>(a, st);
  if (t.th == 7) return launch_conv<T, 7, 14>(a, st);
  if (t.th == 14) return launch_conv<T, 14, 14>(a, st);
  return launch_conv<T, 16, 16>(a, st);
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_conv_num_tiles(int N, int H, int W) {
  TileCfg t = pick_tile(H, W);
  return N * cdiv(H, t.th) * cdiv(W, t.tw);
}

extern "C" void spcl_conv_set_gemm(int on) { conv_set_gemm(on); }

extern "C" void spcl_conv_set_f32_split(int on) { g_f32_split = on != 0 ? 1 : 0; }
extern "C" int spcl_conv_get_f32_split(void) { return g_f32_split; }

extern "C" int spcl_conv_stat_rows(int dtype, int N, int H, int W, int CinK, int CoutS) {
  if (dtype == SPCL_BF16 && conv_use_gemm(CinK, CoutS, H, W)) return conv_gemm_stat_rows(N, H, W, CinK, CoutS);
  if (dtype == SPCL_BF16) {
    const TileCfg t = pick_tile_k(H, W, CinK, CoutS);
    return N * cdiv(H, t.th) * cdiv(W, t.tw);
  }
  return spcl_conv_num_tiles(N, H, W);
}

extern "C" size_t spcl_conv_packed_elems(int Cin, int Cout, int kind, int dtype) {
  const int KinK = round_up(kind == 0 ? Cin : Cout, 16), NoutS = round_up(kind == 0 ? Cout : Cin, 16);
  return dtype == SPCL_F32 ? packed_elems<float>(KinK, NoutS) : packed_elems<bf16_t>(KinK, NoutS);
}

extern "C" int spcl_conv_pack_weights(const float* w_oihw, int Cin, int Cout, int kind, int dtype, void* packed,
                                      void* stream) {
  SPCL_CHECK_ARG(w_oihw && packed, "conv_pack_weights: null pointer");
  SPCL_CHECK_ARG(Cin > 0 && Cout > 0 && (kind == 0 || kind == 1), "conv_pack_weights: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int KinK = round_up(kind == 0 ? Cin : Cout, 16), NoutS = round_up(kind == 0 ? Cout : Cin, 16);
  if (dtype == SPCL_F32) {
    size_t total = packed_elems<float>(KinK, NoutS);
    SPCL_LAUNCH(conv_pack_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_oihw, Cin,
                       Cout, kind, KinK, NoutS, (float*)packed, total, false);
  } else if (dtype == SPCL_BF16) {
    size_t total = packed_elems<bf16_t>(KinK, NoutS);
    SPCL_LAUNCH(conv_pack_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_oihw, Cin,
                       Cout, kind, KinK, NoutS, (bf16_t*)packed, total, conv_gemm_channels(KinK, NoutS));
  } else {
    set_error("conv_pack_weights: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv_pack_weights");
  return SPCL_OK;
}

extern "C" int spcl_conv_pack_weights_both(const float* w_oihw, int Cin, int Cout, int dtype, void* packed_fwd,
                                           void* packed_dgrad, void* stream) {
  SPCL_CHECK_ARG(w_oihw && packed_fwd && packed_dgrad, "conv_pack_weights_both: null pointer");
  SPCL_CHECK_ARG(Cin > 0 && Cout > 0, "conv_pack_weights_both: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int CinK = round_up(Cin, 16), CoutS = round_up(Cout, 16);
  if (dtype == SPCL_F32) {
    const size_t t0 = packed_elems<float>(CinK, CoutS), t1 = packed_elems<float>(CoutS, CinK);
    SPCL_LAUNCH(conv_pack_both_kernel<float>, dim3((unsigned)((t0 + t1 + 255) / 256)), dim3(256), 0, st, w_oihw,
                       Cin, Cout, CinK, CoutS, (float*)packed_fwd, t0, (float*)packed_dgrad, t1, false);
  } else if (dtype == SPCL_BF16) {
    const size_t t0 = packed_elems<bf16_t>(CinK, CoutS), t1 = packed_elems<bf16_t>(CoutS, CinK);
    SPCL_LAUNCH(conv_pack_both_kernel<bf16_t>, dim3((unsigned)((t0 + t1 + 255) / 256)), dim3(256), 0, st,
                       w_oihw, Cin, Cout, CinK, CoutS, (bf16_t*)packed_fwd, t0, (bf16_t*)packed_dgrad, t1,
                       conv_gemm_channels(CinK, CoutS));
  } else {
    set_error("conv_pack_weights_both: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv_pack_weights_both");
  return SPCL_OK;
}

extern "C" int spcl_conv_pack_weights_block(const float* wa_oihw, int CinA, int CoutA, void* a_fwd, void* a_dgrad,
                                           const float* wb_oihw, int CinB, int CoutB, void* b_fwd, void* b_dgrad,
                                           int dtype, void* stream) {
  return spcl_conv_pack_weights_block_at(wa_oihw, CinA, CoutA, a_fwd, a_dgrad, wb_oihw, CinB, CoutB, b_fwd, b_dgrad, dtype,
                                         0, 0, stream);
}

// H, W > 0: the image size the packed weights will be used at -- the second (band-GEMM) layout of a dual-layout buffer is
// written only where spcl_conv3x3_forward would pick that kernel at this size (it reads nothing else of it); 0, 0: both.
extern "C" int spcl_conv_pack_weights_block_at(const float* wa_oihw, int CinA, int CoutA, void* a_fwd, void* a_dgrad,
                                              const float* wb_oihw, int CinB, int CoutB, void* b_fwd, void* b_dgrad,
                                              int dtype, int H, int W, void* stream) {
  SPCL_CHECK_ARG(wa_oihw && wb_oihw && a_fwd && a_dgrad && b_fwd && b_dgrad, "conv_pack_weights_block: null pointer");
  SPCL_CHECK_ARG(CinA > 0 && CoutA > 0 && CinB > 0 && CoutB > 0 && H >= 0 && W >= 0, "conv_pack_weights_block: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int ka = round_up(CinA, 16), sa = round_up(CoutA, 16), kb = round_up(CinB, 16), sb = round_up(CoutB, 16);
  if (dtype == SPCL_F32) {
    const size_t ta0 = packed_elems<float>(ka, sa), ta1 = packed_elems<float>(sa, ka);
    const size_t tb0 = packed_elems<float>(kb, sb), tb1 = packed_elems<float>(sb, kb);
    SPCL_LAUNCH(conv_pack_block_kernel<float>, dim3((unsigned)((ta0 + ta1 + tb0 + tb1 + 255) / 256)), dim3(256), 0, st,
                wa_oihw, CinA, CoutA, (float*)a_fwd, ta0, (float*)a_dgrad, ta1, wb_oihw, CinB, CoutB, (float*)b_fwd, tb0,
                (float*)b_dgrad, tb1, false, false, false, false);
  } else if (dtype == SPCL_BF16) {
    // elements to write of each buffer: both layouts, or only the per-wave kernels' (the first half)
    auto want = [&](int K, int S) { return conv_gemm_channels(K, S) && (H == 0 || W == 0 || conv_use_gemm(K, S, H, W)); };
    auto elems = [&](int K, int S, bool g) {
      const size_t all = packed_elems<bf16_t>(K, S);
      return (conv_gemm_channels(K, S) && !g) ? all / 2 : all;
    };
    const bool ga0 = want(ka, sa), ga1 = want(sa, ka), gb0 = want(kb, sb), gb1 = want(sb, kb);
    const size_t ta0 = elems(ka, sa, ga0), ta1 = elems(sa, ka, ga1), tb0 = elems(kb, sb, gb0), tb1 = elems(sb, kb, gb1);
    SPCL_LAUNCH(conv_pack_block_kernel<bf16_t>, dim3((unsigned)((ta0 + ta1 + tb0 + tb1 + 255) / 256)), dim3(256), 0, st,
                wa_oihw, CinA, CoutA, (bf16_t*)a_fwd, ta0, (bf16_t*)a_dgrad, ta1, wb_oihw, CinB, CoutB, (bf16_t*)b_fwd,
                tb0, (bf16_t*)b_dgrad, tb1, ga0, ga1, gb0, gb1);
  } else {
    set_error("conv_pack_weights_block: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv_pack_weights_block");
  return SPCL_OK;
}

template <typename T>
static int pack_multi_t(const spcl_pack_item* items, int n, hipStream_t st, const float* img = nullptr, int N = 0, int H = 0,
                        int W = 0, float* acorr = nullptr, void* zero = nullptr, size_t zero_bytes = 0) {
  PackSegs p;
  int nseg = 0, blocks = 0;
  for (int i = 0; i < n; ++i) {
    const spcl_pack_item& it = items[i];
    const int k = round_up(it.Cin, 16), s = round_up(it.Cout, 16);
    for (int kind = 0; kind < 2; ++kind) {
      const int K = kind == 0 ? k : s, S = kind == 0 ? s : k;  // GEMM-K / output channels of this layout
      bool gemm = false;
      size_t elems = packed_elems<T>(K, S);
      if (sizeof(T) == 2 && conv_gemm_channels(K, S)) {  // dual-layout buffer: the band-GEMM half only where it is used
        gemm = it.H == 0 || it.W == 0 || conv_use_gemm(K, S, it.H, it.W);
        if (!gemm) elems /= 2;
      }
      PackSeg& g = p.s[nseg];
      g.w = it.w_oihw; g.out = kind == 0 ? it.fwd : it.dgrad; g.count = (unsigned)elems; g.blk_begin = blocks;
      g.Cin = it.Cin; g.Cout = it.Cout; g.kind = kind; g.KinK = K; g.NoutS = S; g.gemm = gemm ? 1 : 0;
      blocks += (int)((elems + PACK_EPB - 1) / PACK_EPB);
      p.blk_end[nseg++] = blocks;
    }
  }
  if (zero != nullptr && zero_bytes > 0) {  // the zero-fill segment (kind 2), as elements of T
    PackSeg& g = p.s[nseg];
    g.w = nullptr; g.out = zero; g.count = (unsigned)(zero_bytes / sizeof(T)); g.blk_begin = blocks;
    g.Cin = g.Cout = g.KinK = g.NoutS = 16; g.kind = 2; g.gemm = 0;
    blocks += (int)((g.count + PACK_EPB - 1) / PACK_EPB);
    p.blk_end[nseg++] = blocks;
  }
  for (int k = nseg; k < PACK_SEGS; ++k) {
    p.blk_end[k] = 0x7fffffff;
    p.s[k] = p.s[0];
  }
  if (img != nullptr) {
    const int nacorr = N * image_autocorr_bands(H);
    prof_cost((double)N * H * W * 4.0, 2.0 * 54.0 * N * H * W);
    SPCL_LAUNCH(conv_pack_multi_acorr_kernel<T>, dim3((unsigned)(nacorr + blocks)), dim3(256), 0, st, p, img, H, W, acorr,
                blocks);
  } else {
    SPCL_LAUNCH(conv_pack_multi_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, p);
  }
  return 0;
}

extern "C" int spcl_conv_pack_weights_multi(const spcl_pack_item* items, int n, int dtype, void* stream) {
  return spcl_conv_pack_weights_multi_acorr(items, n, dtype, nullptr, 0, 0, 0, nullptr, stream);
}

extern "C" int spcl_conv_pack_weights_multi_acorr(const spcl_pack_item* items, int n, int dtype, const float* image, int N,
                                                  int H, int W, float* acorr, void* stream) {
  return spcl_conv_pack_weights_multi_zero(items, n, dtype, image, N, H, W, acorr, nullptr, 0, stream);
}

// ... and the same launch ZEROES `zero_bytes` bytes at `zero` (a multiple of 4; the step's BatchNorm accumulator blocks,
// spcl_bn_acc_elems): the pack is the first launch of a forward pass, the blocks must be zero before the first convolution
extern "C" int spcl_conv_pack_weights_multi_zero(const spcl_pack_item* items, int n, int dtype, const float* image, int N,
                                                 int H, int W, float* acorr, void* zero, size_t zero_bytes, void* stream) {
  SPCL_CHECK_ARG((zero == nullptr) == (zero_bytes == 0) && zero_bytes % 4 == 0 && zero_bytes < (1ull << 31),
                 "conv_pack_weights_multi_zero: bad zero region");
  SPCL_CHECK_ARG(items && n >= 1 && n <= SPCL_PACK_MULTI_MAX, "conv_pack_weights_multi: 1 <= n <= %d layers",
                 SPCL_PACK_MULTI_MAX);
  SPCL_CHECK_ARG(image == nullptr || (acorr && N > 0 && H > 0 && W > 0 && W <= ACORR_MAXW),
                 "conv_pack_weights_multi_acorr: bad image arguments (W <= %d)", ACORR_MAXW);
  for (int i = 0; i < n; ++i) {
    SPCL_CHECK_ARG(items[i].w_oihw && items[i].fwd && items[i].dgrad, "conv_pack_weights_multi: null pointer (layer %d)", i);
    SPCL_CHECK_ARG(items[i].Cin > 0 && items[i].Cout > 0 && items[i].H >= 0 && items[i].W >= 0,
                   "conv_pack_weights_multi: bad shape (layer %d)", i);
  }
  if (dtype == SPCL_F32) pack_multi_t<float>(items, n, (hipStream_t)stream, image, N, H, W, acorr, zero, zero_bytes);
  else if (dtype == SPCL_BF16) pack_multi_t<bf16_t>(items, n, (hipStream_t)stream, image, N, H, W, acorr, zero, zero_bytes);
  else {
    set_error("conv_pack_weights_multi: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("conv_pack_weights_multi");
  return SPCL_OK;
}

static BnAccFwd bn_acc_fwd_of(const spcl_bn_acc& b) {
  return BnAccFwd{b.acc, b.gamma, b.beta, b.running_mean, b.running_var, b.num_batches_tracked, b.st, b.momentum, b.eps,
                  b.count, b.C, b.CS, 1.0 / (double)b.count};
}

static int conv3x3_forward_impl(const void* x, int dtype, int N, int H, int W, int CinS, int CinK, int CoutS,
                                const void* w_packed, int in_mode, const float* in_scale, const float* in_shift,
                                void* y, float* stats, void* stream, const BnAccFwd* in_bn, long long* stats_acc);

extern "C" int spcl_conv3x3_forward(const void* x, int dtype, int N, int H, int W, int CinS, int CinK, int CoutS,
                                    const void* w_packed, int in_mode, const float* in_scale, const float* in_shift,
                                    void* y, float* stats, void* stream) {
  return conv3x3_forward_impl(x, dtype, N, H, W, CinS, CinK, CoutS, w_packed, in_mode, in_scale, in_shift, y, stats, stream,
                              nullptr, nullptr);
}

// The forward convolution with its BatchNorm sums going through fixed-point accumulator blocks (bn_acc.hpp) on either side:
//   in_bn != NULL:     the input is the RAW output of the previous convolution; relu(scale x + shift) of ITS BatchNorm is applied
//                      by the loader with scale / shift derived from in_bn->acc in the prologue (no finalize launch); the first
//                      workgroup writes in_bn->st and the running statistics.  in_bn == NULL: scale / shift from the arrays
//                      in_scale / in_shift when given (spcl_conv3x3_forward's in_mode 1), else the input is read as it is.
//   stats_acc != NULL: the output's sum x / sum x^2 are added to that block (zeroed by the caller) instead of per-tile rows.
// bf16 layers the specialised 14-column kernels take, at most BN_ACC_MAX_TILES tiles: spcl_conv_bn_acc_supported says which.
extern "C" int spcl_conv_bn_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int in_kind,
                                          int with_stats_acc) {
  // in_kind: 0 the input is read as it is, 1 its BatchNorm + ReLU from a block (in_bn), 2 from scale / shift arrays
  if (dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || CinK <= 0 || CoutS <= 0 || CinK % 16 || CoutS % 16) return 0;
  if (!(CinK <= 64 || CinK % 64 == 0) || (in_kind != 1 && !with_stats_acc) || in_kind < 0 || in_kind > 2) return 0;
  if (conv_use_gemm(CinK, CoutS, H, W)) return 0;
  ConvArgs a;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr; a.stats = nullptr; a.in_scale = a.in_shift = nullptr;
  a.N = N; a.H = H; a.W = W; a.CinS = CinK; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = in_kind != 0 ? 1 : 0;
  a.tilesX = a.tilesY = 0; a.tpw = 1; a.dbg = 0;
  BnAccFwd b{};
  long long dummy;
  b.acc = &dummy; b.CS = CinK; b.C = CinK;
  if (in_kind == 1) a.in_bn = &b;
  if (with_stats_acc) a.stats_acc = &dummy;
  const TileCfg t = pick_tile_k(H, W, CinK, CoutS);
  return (t.tw == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

extern "C" size_t spcl_bn_acc_elems(int CS) { return CS > 0 ? bn_acc_words(CS) : 0; }

extern "C" int spcl_conv3x3_forward_acc(const void* x, int dtype, int N, int H, int W, int CinK, int CoutS,
                                        const void* w_packed, const spcl_bn_acc* in_bn, const float* in_scale,
                                        const float* in_shift, void* y, long long* stats_acc, float* stats_rows,
                                        void* stream) {
  SPCL_CHECK_ARG(in_bn != nullptr || stats_acc != nullptr, "conv3x3_forward_acc: neither side uses an accumulator block");
  SPCL_CHECK_ARG(!(stats_acc != nullptr && stats_rows != nullptr), "conv3x3_forward_acc: statistics either as a block or as rows");
  SPCL_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr) && !(in_bn != nullptr && in_scale != nullptr),
                 "conv3x3_forward_acc: the input's coefficients either from a block or from scale / shift arrays");
  BnAccFwd b{};
  if (in_bn != nullptr) {
    SPCL_CHECK_ARG(in_bn->acc && in_bn->gamma && in_bn->beta && in_bn->st && in_bn->count >= 1.f && in_bn->C > 0 &&
                   in_bn->C <= in_bn->CS && in_bn->CS == CinK, "conv3x3_forward_acc: bad in_bn (CS must equal CinK)");
    b = bn_acc_fwd_of(*in_bn);
  }
  return conv3x3_forward_impl(x, dtype, N, H, W, CinK, CinK, CoutS, w_packed, (in_bn != nullptr || in_scale != nullptr) ? 1 : 0,
                              in_scale, in_shift, y, stats_rows, stream, in_bn != nullptr ? &b : nullptr, stats_acc);
}

static int conv3x3_forward_impl(const void* x, int dtype, int N, int H, int W, int CinS, int CinK, int CoutS,
                                const void* w_packed, int in_mode, const float* in_scale, const float* in_shift,
                                void* y, float* stats, void* stream, const BnAccFwd* in_bn, long long* stats_acc) {
  SPCL_CHECK_ARG(x && y && w_packed, "conv3x3_forward: null pointer");
  SPCL_CHECK_ARG(N > 0 && H > 0 && W > 0, "conv3x3_forward: bad shape");
  SPCL_CHECK_ARG(CinK % 16 == 0 && CoutS % 16 == 0 && CinK > 0 && CoutS > 0, "conv3x3_forward: CinK=%d CoutS=%d must "
                 "be multiples of 16", CinK, CoutS);
  SPCL_CHECK_ARG(in_mode >= 0 && in_mode <= 2, "conv3x3_forward: in_mode %d", in_mode);
  SPCL_CHECK_ARG(in_mode != 1 || (in_scale && in_shift) || in_bn != nullptr, "conv3x3_forward: in_mode 1 needs scale/shift");
  if (in_mode == 2) SPCL_CHECK_ARG(CinK == 16 && CinS >= 1 && CinS <= 16, "conv3x3_forward: image mode needs Cin<=16");
  else SPCL_CHECK_ARG(CinS == CinK, "conv3x3_forward: CinS (%d) must equal CinK (%d)", CinS, CinK);
  SPCL_CHECK_ARG(CinK <= 64 || CinK % 64 == 0, "conv3x3_forward: CinK=%d must be <=64 or a multiple of 64", CinK);
  ConvArgs a;
  a.x = x; a.y = y; a.wp = w_packed; a.stats = stats; a.in_scale = in_scale; a.in_shift = in_shift;
  a.N = N; a.H = H; a.W = W; a.CinS = CinS; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = in_mode;
  a.tilesX = a.tilesY = 0;
  a.tpw = 1;
  a.dbg = 0;
  a.in_bn = in_bn;
  a.stats_acc = stats_acc;
  hipStream_t st = (hipStream_t)stream;
  if (in_bn != nullptr || stats_acc != nullptr) {  // only the specialised kernels know the blocks: never a silent fallback
    const TileCfg t = pick_tile_k(H, W, CinK, CoutS);
    if (dtype != SPCL_BF16 || conv_use_gemm(CinK, CoutS, H, W) || t.tw != 14 || !launch_conv_fast(a, t.th, st, true)) {
      set_error("conv3x3_forward_acc: no kernel with accumulator blocks for N=%d H=%d W=%d CinK=%d CoutS=%d (ask "
                "spcl_conv_bn_acc_supported)", N, H, W, CinK, CoutS);
      return SPCL_EUNSUPPORTED;
    }
  }
  {  // algorithmic cost of the launch: x once, y once, the packed weights once
    const double px = (double)N * H * W, es = dtype == SPCL_F32 ? 4.0 : 2.0;
    const double cin = in_mode == 2 ? CinS : CinK;
    prof_cost(px * ((in_mode == 2 ? CinS * 4.0 : CinK * es) + CoutS * es) + 9.0 * CinK * CoutS * es,
              2.0 * px * 9.0 * cin * CoutS);
  }
  int rc = 0;
  if (dtype == SPCL_F32) rc = launch_conv_t<float>(a, st);
  else if (dtype == SPCL_BF16) rc = launch_conv_t<bf16_t>(a, st);
  else {
    set_error("conv3x3_forward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  if (rc != 0) {
    set_error("conv3x3_forward: no kernel for N=%d H=%d W=%d CinK=%d CoutS=%d in_mode=%d", N, H, W, CinK, CoutS, in_mode);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_forward");
  return SPCL_OK;
}

// The image convolution (semi_seg/arch/unet.py:123, 1 -> 16 channels, in_mode 2) with the image's autocorrelation partial
// rows as a by-product (ConvArgs::acorr_rows; csrc/conv_fast.hip conv3x3_image_kernel<.., ACORR>): one row [64] per tile in
// the layout of spcl_image_autocorr's band rows, so everything downstream (spcl_conv16_bwd_fused's fold,
// spcl_bnrelu_backward_*_image3) takes them unchanged.  _rows: the number of rows (tiles), 0 where the specialised kernel
// does not take the shape (bf16, one input channel, 16 output channels, whole 14 x 14 tiles).
static bool image_acorr_args(ConvArgs& a, int dtype, int N, int H, int W, int CinS, int CoutS) {
  static const bool off = lab_env("SPCL_ACORR_IN_CONV", 1) == 0;  // A/B switch
  if (off || dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || CinS != 1 || CoutS != 16) return false;
  a.N = N; a.H = H; a.W = W; a.CinS = CinS; a.CinK = 16; a.CoutS = CoutS; a.in_mode = 2;
  a.tilesX = a.tilesY = 0; a.tpw = 1; a.dbg = 0;
  const TileCfg t = pick_tile_k(H, W, 16, CoutS);
  static const int no_fast = lab_env("SPCL_CONV_NO_FAST", 0);
  return !no_fast && t.tw == 14 && t.th == 14 && H % 14 == 0 && W % 14 == 0;
}

extern "C" int spcl_conv3x3_forward_image_acorr_rows(int dtype, int N, int H, int W, int CinS, int CoutS) {
  ConvArgs a;
  float dummy = 0.f;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr; a.stats = nullptr; a.in_scale = a.in_shift = nullptr;
  if (!image_acorr_args(a, dtype, N, H, W, CinS, CoutS)) return 0;
  a.acorr_rows = &dummy;
  if (!launch_conv_fast(a, 14, nullptr, true)) return 0;
  return N * cdiv(H, 14) * cdiv(W, 14);
}

extern "C" int spcl_conv3x3_forward_image_acorr(const void* x, int dtype, int N, int H, int W, int CinS, int CoutS,
                                                const void* w_packed, void* y, float* stats, float* acorr_rows,
                                                void* stream) {
  SPCL_CHECK_ARG(x && y && w_packed && acorr_rows, "conv3x3_forward_image_acorr: null pointer");
  ConvArgs a;
  a.x = x; a.y = y; a.wp = w_packed; a.stats = stats; a.in_scale = a.in_shift = nullptr;
  if (!image_acorr_args(a, dtype, N, H, W, CinS, CoutS)) {
    set_error("conv3x3_forward_image_acorr: unsupported configuration (ask spcl_conv3x3_forward_image_acorr_rows)");
    return SPCL_EUNSUPPORTED;
  }
  a.acorr_rows = acorr_rows;
  {
    const double px = (double)N * H * W;
    prof_cost(px * (4.0 + CoutS * 2.0) + 9.0 * 16 * CoutS * 2.0 + (double)N * cdiv(H, 14) * cdiv(W, 14) * 256.0,
              2.0 * px * 9.0 * CinS * CoutS + 2.0 * px * 2.0 * 256.0);
  }
  if (!launch_conv_fast(a, 14, (hipStream_t)stream)) {
    set_error("conv3x3_forward_image_acorr: no kernel for N=%d H=%d W=%d", N, H, W);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_forward_image_acorr");
  return SPCL_OK;
}

// The decoder's torch.cat((skip, up), dim=1) -> Conv2d (unet.py:194-224) without the concatenated tensor: the convolution
// reads its input channels [0, Chalf) from xa and [Chalf, 2 Chalf) from xb (both dense [N][H][W][Chalf] bf16, no input
// transform).  Only where a specialised kernel exists (14-column tiles, 2 Chalf = 32 or 64): ask _supported first.
static bool conv_cat_args(ConvArgs& a, int dtype, int N, int H, int W, int Chalf, int CoutS) {
  static const bool off = lab_env("SPCL_CONV_CAT", 1) == 0;  // A/B switch
  if (off || dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || CoutS % 16 != 0 || CoutS <= 0 ||
      (Chalf != 16 && Chalf != 32 && Chalf != 64 && Chalf != 128))
    return false;
  // (a layer whose weight gradient the batched GEMM kernel takes reads 64-channel blocks: the halves must be whole blocks)
  if ((2 * Chalf) % 64 == 0 && CoutS % 64 == 0 && Chalf % 64 != 0) return false;
  a.in_scale = a.in_shift = nullptr;
  a.stats = nullptr;
  a.N = N; a.H = H; a.W = W; a.CinS = 2 * Chalf; a.CinK = 2 * Chalf; a.CoutS = CoutS; a.in_mode = 0;
  a.tilesX = a.tilesY = 0;
  a.tpw = 1;
  a.dbg = 0;
  return !conv_use_gemm(a.CinK, CoutS, H, W) && pick_tile(H, W).tw == 14;
}

extern "C" int spcl_conv_cat_supported(int dtype, int N, int H, int W, int Chalf, int CoutS) {
  ConvArgs a;
  if (!conv_cat_args(a, dtype, N, H, W, Chalf, CoutS)) return 0;
  static const char dummy[16] = {0};
  a.x = dummy; a.x2 = dummy; a.y = nullptr; a.wp = nullptr;
  return launch_conv_fast(a, pick_tile_k(H, W, a.CinK).th, nullptr, true) ? 1 : 0;
}

extern "C" int spcl_conv3x3_forward_cat(const void* xa, const void* xb, int dtype, int N, int H, int W, int Chalf, int CoutS,
                                        const void* w_packed, const float* xb_scale, const float* xb_shift, void* y,
                                        float* stats, void* stream) {
  SPCL_CHECK_ARG(xa && xb && y && w_packed, "conv3x3_forward_cat: null pointer");
  SPCL_CHECK_ARG((xb_scale == nullptr) == (xb_shift == nullptr), "conv3x3_forward_cat: xb_scale and xb_shift come together");
  SPCL_CHECK_ARG(xb_scale == nullptr || Chalf <= 32, "conv3x3_forward_cat: the raw second tensor exists for Chalf 16 / 32 only");
  SPCL_CHECK_ARG((uintptr_t)xa % 16 == 0 && (uintptr_t)xb % 16 == 0, "conv3x3_forward_cat: inputs must be 16-byte aligned");
  ConvArgs a;
  if (!conv_cat_args(a, dtype, N, H, W, Chalf, CoutS)) {
    set_error("conv3x3_forward_cat: unsupported configuration (bf16, Chalf 16 / 32 / 64 / 128, 14-column tiles)");
    return SPCL_EUNSUPPORTED;
  }
  a.x = xa; a.x2 = xb; a.y = y; a.wp = w_packed; a.stats = stats;
  if (xb_scale != nullptr) { a.in_mode = 1; a.in_scale = xb_scale; a.in_shift = xb_shift; }
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (2.0 * Chalf + CoutS) * 2.0 + 9.0 * 2.0 * Chalf * CoutS * 2.0, 2.0 * px * 9.0 * 2.0 * Chalf * CoutS);
  if (!launch_conv_fast(a, pick_tile_k(H, W, a.CinK).th, st)) {
    set_error("conv3x3_forward_cat: no specialised kernel for H=%d W=%d Chalf=%d CoutS=%d", H, W, Chalf, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_forward_cat");
  return SPCL_OK;
}

// nn.Upsample(scale_factor=2) -> Conv2d (the up-convolution, unet.py:89-90) without the upsampled tensor: x is the
// half-resolution activation [N][H / 2][W / 2][CinK], H x W the convolution's (fine) size; y / stats as spcl_conv3x3_forward.
static bool conv_up2_args(ConvArgs& a, int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_env("SPCL_CONV_UP2", 1) == 0;  // A/B switch
  if (off || dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || CinK % 16 != 0 || CinK <= 0 ||
      CoutS % 16 != 0 || CoutS <= 0)
    return false;
  if (!(CinK <= 64 || CinK % 64 == 0)) return false;
  a.in_scale = a.in_shift = nullptr;
  a.stats = nullptr;
  a.N = N; a.H = H; a.W = W; a.CinS = CinK; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = 0;
  a.tilesX = a.tilesY = 0;
  a.tpw = 1;
  a.dbg = 0;
  a.x_up2 = true;
  return !conv_use_gemm(CinK, CoutS, H, W) && pick_tile(H, W).tw == 14;
}

extern "C" int spcl_conv_up2_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  ConvArgs a;
  if (!conv_up2_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  static const char dummy[16] = {0};
  a.x = dummy; a.y = nullptr; a.wp = nullptr;
  return launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, nullptr, true) ? 1 : 0;
}

extern "C" int spcl_conv3x3_forward_up2(const void* x_half, int dtype, int N, int H, int W, int CinK, int CoutS,
                                        const void* w_packed, void* y, float* stats, void* stream) {
  SPCL_CHECK_ARG(x_half && y && w_packed, "conv3x3_forward_up2: null pointer");
  ConvArgs a;
  if (!conv_up2_args(a, dtype, N, H, W, CinK, CoutS)) {
    set_error("conv3x3_forward_up2: unsupported configuration (bf16, even sizes, 14-column tiles)");
    return SPCL_EUNSUPPORTED;
  }
  a.x = x_half; a.y = y; a.wp = w_packed; a.stats = stats;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (0.25 * CinK + CoutS) * 2.0 + 9.0 * CinK * CoutS * 2.0, 2.0 * px * 9.0 * CinK * CoutS);
  if (!launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, st)) {
    set_error("conv3x3_forward_up2: no specialised kernel for H=%d W=%d CinK=%d CoutS=%d", H, W, CinK, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_forward_up2");
  return SPCL_OK;
}

// ... and its input gradient as the two gradients of the concatenated tensors: the plain convolution (dgrad weights) whose
// output channels [0, CoutS / 2) go to y_lo and [CoutS / 2, CoutS) to y_hi, both dense [N][H][W][CoutS / 2] -- each producer's
// backward then reads whole pixels instead of half of every line of one interleaved tensor.
static bool conv_split_args(ConvArgs& a, int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_env("SPCL_CONV_SPLIT", 1) == 0;  // A/B switch
  if (off || dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || CinK % 16 != 0 || CinK <= 0 || CoutS % 32 != 0 || CoutS <= 0)
    return false;
  if (!(CinK <= 64 || CinK % 64 == 0)) return false;
  a.in_scale = a.in_shift = nullptr;
  a.stats = nullptr;
  a.N = N; a.H = H; a.W = W; a.CinS = CinK; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = 0;
  a.tilesX = a.tilesY = 0;
  a.tpw = 1;
  a.dbg = 0;
  return !conv_use_gemm(CinK, CoutS, H, W) && pick_tile(H, W).tw == 14;
}

extern "C" int spcl_conv_split_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  ConvArgs a;
  if (!conv_split_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  static char dummy[16] = {0};
  a.x = dummy; a.y = dummy; a.y_hi = dummy; a.wp = nullptr;
  return launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, nullptr, true) ? 1 : 0;
}

extern "C" int spcl_conv3x3_forward_split(const void* x, int dtype, int N, int H, int W, int CinK, int CoutS,
                                          const void* w_packed, void* y_lo, void* y_hi, void* stream) {
  SPCL_CHECK_ARG(x && y_lo && y_hi && w_packed, "conv3x3_forward_split: null pointer");
  SPCL_CHECK_ARG((uintptr_t)y_lo % 16 == 0 && (uintptr_t)y_hi % 16 == 0, "conv3x3_forward_split: outputs must be 16-byte aligned");
  ConvArgs a;
  if (!conv_split_args(a, dtype, N, H, W, CinK, CoutS)) {
    set_error("conv3x3_forward_split: unsupported configuration (bf16, CoutS a multiple of 32, 14-column tiles)");
    return SPCL_EUNSUPPORTED;
  }
  a.x = x; a.y = y_lo; a.y_hi = y_hi; a.wp = w_packed;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (CinK + CoutS) * 2.0 + 9.0 * CinK * CoutS * 2.0, 2.0 * px * 9.0 * CinK * CoutS);
  if (!launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, st)) {
    set_error("conv3x3_forward_split: no specialised kernel for H=%d W=%d CinK=%d CoutS=%d", H, W, CinK, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_forward_split");
  return SPCL_OK;
}

// ... and with the BatchNorm-backward partial sums of the layer behind the UPPER half (the up-convolution, unet.py:90-92: y2_hi
// [N][H][W][CoutS / 2] its raw output, scale2 / shift2 / mean2 its coefficients): rows2 [spcl_conv_stat_rows(...)][2][CoutS / 2]
// as spcl_conv3x3_dgrad_bnstats leaves them -- that layer's reduction pass over (y2, g_hi) disappears.
extern "C" int spcl_conv_split_bnstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_env("SPCL_CONV_SPLIT_BNSTATS", 1) == 0;  // A/B switch
  ConvArgs a;
  if (off || !conv_split_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  static char dummy[16] = {0};
  static float fdummy;
  a.x = dummy; a.y = dummy; a.y_hi = dummy; a.wp = nullptr; a.rows2 = &fdummy;
  return launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, nullptr, true) ? 1 : 0;
}

extern "C" int spcl_conv3x3_dgrad_split_bnstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                                const void* w_packed, void* g_lo, void* g_hi, const void* y2_hi,
                                                const float* scale2, const float* shift2, const float* mean2, float* rows2,
                                                void* stream) {
  SPCL_CHECK_ARG(dy && g_lo && g_hi && w_packed && y2_hi && scale2 && shift2 && mean2 && rows2,
                 "conv3x3_dgrad_split_bnstats: null pointer");
  SPCL_CHECK_ARG((uintptr_t)g_lo % 16 == 0 && (uintptr_t)g_hi % 16 == 0 && (uintptr_t)y2_hi % 16 == 0,
                 "conv3x3_dgrad_split_bnstats: tensors must be 16-byte aligned");
  ConvArgs a;
  if (!conv_split_args(a, dtype, N, H, W, CinK, CoutS)) {
    set_error("conv3x3_dgrad_split_bnstats: unsupported configuration");
    return SPCL_EUNSUPPORTED;
  }
  a.x = dy; a.y = g_lo; a.y_hi = g_hi; a.wp = w_packed;
  a.y2 = y2_hi; a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2; a.rows2 = rows2;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (CinK + 1.5 * CoutS) * 2.0 + 9.0 * CinK * CoutS * 2.0, 2.0 * px * 9.0 * CinK * CoutS);
  if (!launch_conv_fast(a, pick_tile_k(H, W, CinK, CoutS).th, st)) {
    set_error("conv3x3_dgrad_split_bnstats: no specialised kernel for H=%d W=%d CinK=%d CoutS=%d", H, W, CinK, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_dgrad_split_bnstats");
  return SPCL_OK;
}

// dgrad of the SECOND conv of a block fused with the per-tile partial sums of the FIRST conv's BatchNorm backward (the
// dgrad's output g is the gradient of relu(bn(y2))): saves the separate reduction pass over (y2, g).  Only where a
// specialised kernel exists (bf16, tiles of 14 columns): ask spcl_conv_dgrad_bnstats_supported first.
static bool dgrad_bnstats_args(ConvArgs& a, int dtype, int N, int H, int W, int CinK, int CoutS) {
  if (dtype != SPCL_BF16 || N <= 0 || H <= 0 || W <= 0 || CinK % 16 != 0 || CoutS % 16 != 0) return false;
  if (!(CinK <= 64 || CinK % 64 == 0)) return false;
  a.in_scale = a.in_shift = nullptr;
  a.stats = nullptr;
  a.N = N; a.H = H; a.W = W; a.CinS = CinK; a.CinK = CinK; a.CoutS = CoutS; a.in_mode = 0;
  a.tilesX = a.tilesY = 0;
  a.tpw = 1;
  a.dbg = 0;
  return true;
}

extern "C" int spcl_conv_dgrad_bnstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_flag("SPCL_NO_DGRAD_BNSTATS");  // A/B switch
  ConvArgs a;
  if (off || !dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  if (conv_use_gemm(CinK, CoutS, H, W)) return 1;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr;
  float dummy;
  a.rows2 = &dummy;
  TileCfg t = pick_tile(H, W);
  return (t.tw == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

static int dgrad_bnstats_impl(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed, void* g,
                              const void* y2, const float* scale2, const float* shift2, const float* mean2, float* rows2,
                              long long* acc, void* stream);

extern "C" int spcl_conv3x3_dgrad_bnstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                          const void* w_packed, void* g, const void* y2, const float* scale2,
                                          const float* shift2, const float* mean2, float* rows2, void* stream) {
  SPCL_CHECK_ARG(rows2, "conv3x3_dgrad_bnstats: null pointer");
  return dgrad_bnstats_impl(dy, dtype, N, H, W, CinK, CoutS, w_packed, g, y2, scale2, shift2, mean2, rows2, nullptr, stream);
}

// ... with the sums ADDED to a fixed-point accumulator block (bn_acc.hpp; spcl_bn_acc_elems(CoutS) words, zeroed by the caller)
// instead of written as per-tile rows: the BatchNorm-backward apply pass derives its coefficients from the block in its
// prologue (spcl_bnrelu_backward_acc), no reduction / finalize launch in between.  Specialised per-wave kernels only.
extern "C" int spcl_conv_dgrad_bnstats_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  ConvArgs a;
  if (!spcl_conv_dgrad_bnstats_supported(dtype, N, H, W, CinK, CoutS) || conv_use_gemm(CinK, CoutS, H, W)) return 0;
  if (!dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr;
  float dummy;
  long long dummy2;
  a.rows2 = &dummy;
  a.rows2_acc = &dummy2;
  TileCfg t = pick_tile(H, W);
  return (t.tw == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

extern "C" int spcl_conv3x3_dgrad_bnstats_acc(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                              const void* w_packed, void* g, const void* y2, const float* scale2,
                                              const float* shift2, const float* mean2, long long* acc, void* stream) {
  SPCL_CHECK_ARG(acc, "conv3x3_dgrad_bnstats_acc: null pointer");
  return dgrad_bnstats_impl(dy, dtype, N, H, W, CinK, CoutS, w_packed, g, y2, scale2, shift2, mean2, nullptr, acc, stream);
}

static int dgrad_bnstats_impl(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed, void* g,
                              const void* y2, const float* scale2, const float* shift2, const float* mean2, float* rows2,
                              long long* acc, void* stream) {
  SPCL_CHECK_ARG(dy && w_packed && g && y2 && scale2 && shift2 && mean2, "conv3x3_dgrad_bnstats: null pointer");
  ConvArgs a;
  if (!dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS) || (acc != nullptr && conv_use_gemm(CinK, CoutS, H, W))) {
    set_error("conv3x3_dgrad_bnstats: unsupported configuration");
    return SPCL_EUNSUPPORTED;
  }
  a.x = dy; a.y = g; a.wp = w_packed;
  a.y2 = y2; a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2;
  a.rows2 = acc != nullptr ? (float*)acc : rows2;  // (with a block: only the mode marker, never written)
  a.rows2_acc = acc;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (CinK + 2.0 * CoutS) * 2.0 + 9.0 * CinK * CoutS * 2.0, 2.0 * px * 9.0 * CinK * CoutS);
  TileCfg t = pick_tile(H, W);
  if (conv_use_gemm(CinK, CoutS, H, W)) {
    if (!launch_conv_gemm(a, st)) {
      set_error("conv3x3_dgrad_bnstats: H=%d W=%d CinK=%d CoutS=%d outside the gemm kernel's range", H, W, CinK, CoutS);
      return SPCL_EUNSUPPORTED;
    }
  } else if (!(t.tw == 14 && launch_conv_fast(a, t.th, st))) {
    set_error("conv3x3_dgrad_bnstats: no specialised kernel for H=%d W=%d CinK=%d CoutS=%d", H, W, CinK, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_dgrad_bnstats");
  return SPCL_OK;
}

// The same for the block behind a ONE-CHANNEL f32 image (unet.py:123, input_dim == 1): rows2 is [tile][11][CoutS] and rows
// 2 .. 10 hold sum_p dz[p][co] image[p + tap] -- the data-dependent part of the first conv's weight gradient (bn.hip,
// spcl_bnrelu_backward_rows_image3 finishes it without another pass over y2 / g).  16 -> 16 channels on 14 x 14 tiles only.
extern "C" int spcl_conv_dgrad_bnstats_image_supported(int dtype, int N, int H, int W, int CinK, int CoutS) {
  static const bool off = lab_flag("SPCL_NO_IMAGE3");  // A/B switch
  ConvArgs a;
  if (off || CinK != 16 || CoutS != 16 || !dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  if (!spcl_conv_dgrad_bnstats_supported(dtype, N, H, W, CinK, CoutS) || conv_use_gemm(CinK, CoutS, H, W)) return 0;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr;
  float dummy;
  a.rows2 = &dummy;
  a.img2 = &dummy;
  TileCfg t = pick_tile(H, W);
  return (t.tw == 14 && t.th == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

extern "C" int spcl_conv3x3_dgrad_bnstats_image(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                                const void* w_packed, void* g, const void* y2, const float* scale2,
                                                const float* shift2, const float* mean2, const float* image,
                                                float* rows11, void* stream) {
  SPCL_CHECK_ARG(dy && w_packed && y2 && scale2 && shift2 && mean2 && image && rows11,
                 "conv3x3_dgrad_bnstats_image: null pointer");  // g may be null: the gradient itself is not written
  ConvArgs a;
  if (!spcl_conv_dgrad_bnstats_image_supported(dtype, N, H, W, CinK, CoutS) ||
      !dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) {
    set_error("conv3x3_dgrad_bnstats_image: unsupported configuration (bf16, 16 -> 16 channels, 14 x 14 tiles)");
    return SPCL_EUNSUPPORTED;
  }
  a.x = dy; a.y = g; a.wp = w_packed;
  a.y2 = y2; a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2; a.rows2 = rows11; a.img2 = image;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (CinK + (g ? 2.0 : 1.0) * CoutS) * 2.0 + px * 4.0 + 9.0 * CinK * CoutS * 2.0,
            2.0 * px * 9.0 * CinK * CoutS + 2.0 * px * 9.0 * CoutS);
  TileCfg t = pick_tile(H, W);
  if (!(t.tw == 14 && launch_conv_fast(a, t.th, st))) {
    set_error("conv3x3_dgrad_bnstats_image: no specialised kernel for H=%d W=%d", H, W);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_dgrad_bnstats_image");
  return SPCL_OK;
}

// The same for a POOLED boundary: dy is the gradient of the next block's first conv output, g = its input gradient =
// d loss / d maxpool2x2(relu(bn(y2))) at H x W, y2 the raw conv output of the block before at H2 x W2 (H = H2 / 2).  The
// epilogue routes g to the window's first positive maximum and leaves the per-tile partial sums of that BatchNorm's
// backward: the separate reduction pass over y2 (bnrelu_bwd_pool_kernel<T, false>) disappears.  Per-wave kernels only.
extern "C" int spcl_conv_dgrad_poolstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int H2, int W2) {
  static const bool off = lab_flag("SPCL_NO_DGRAD_POOLSTATS");  // A/B switch
  ConvArgs a;
  if (off || H != H2 / 2 || W != W2 / 2 || !dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  if (conv_use_gemm(CinK, CoutS, H, W)) return 0;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr;
  float dummy;
  a.rows2 = &dummy;
  a.H2 = H2; a.W2 = W2;
  TileCfg t = pick_tile(H, W);
  return (t.tw == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

static int dgrad_poolstats_impl(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                                void* g, const void* y2, int H2, int W2, const float* scale2, const float* shift2,
                                const float* mean2, float* rows2, long long* acc, void* stream);

extern "C" int spcl_conv3x3_dgrad_poolstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                            const void* w_packed, void* g, const void* y2, int H2, int W2,
                                            const float* scale2, const float* shift2, const float* mean2, float* rows2,
                                            void* stream) {
  SPCL_CHECK_ARG(rows2, "conv3x3_dgrad_poolstats: null pointer");
  return dgrad_poolstats_impl(dy, dtype, N, H, W, CinK, CoutS, w_packed, g, y2, H2, W2, scale2, shift2, mean2, rows2, nullptr,
                              stream);
}

// ... with the sums added to a fixed-point accumulator block (see spcl_conv3x3_dgrad_bnstats_acc)
extern "C" int spcl_conv_dgrad_poolstats_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int H2, int W2) {
  ConvArgs a;
  if (!spcl_conv_dgrad_poolstats_supported(dtype, N, H, W, CinK, CoutS, H2, W2)) return 0;
  if (!dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS)) return 0;
  a.x = nullptr; a.y = nullptr; a.wp = nullptr;
  float dummy;
  long long dummy2;
  a.rows2 = &dummy;
  a.rows2_acc = &dummy2;
  a.H2 = H2; a.W2 = W2;
  TileCfg t = pick_tile(H, W);
  return (t.tw == 14 && launch_conv_fast(a, t.th, nullptr, true)) ? 1 : 0;
}

extern "C" int spcl_conv3x3_dgrad_poolstats_acc(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                                const void* w_packed, void* g, const void* y2, int H2, int W2,
                                                const float* scale2, const float* shift2, const float* mean2,
                                                long long* acc, void* stream) {
  SPCL_CHECK_ARG(acc, "conv3x3_dgrad_poolstats_acc: null pointer");
  return dgrad_poolstats_impl(dy, dtype, N, H, W, CinK, CoutS, w_packed, g, y2, H2, W2, scale2, shift2, mean2, nullptr, acc,
                              stream);
}

static int dgrad_poolstats_impl(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                                void* g, const void* y2, int H2, int W2, const float* scale2, const float* shift2,
                                const float* mean2, float* rows2, long long* acc, void* stream) {
  SPCL_CHECK_ARG(dy && w_packed && g && y2 && scale2 && shift2 && mean2, "conv3x3_dgrad_poolstats: null pointer");
  ConvArgs a;
  if (H != H2 / 2 || W != W2 / 2 || !dgrad_bnstats_args(a, dtype, N, H, W, CinK, CoutS) ||
      conv_use_gemm(CinK, CoutS, H, W)) {
    set_error("conv3x3_dgrad_poolstats: unsupported configuration");
    return SPCL_EUNSUPPORTED;
  }
  a.x = dy; a.y = g; a.wp = w_packed;
  a.y2 = y2; a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2;
  a.rows2 = acc != nullptr ? (float*)acc : rows2;  // (with a block: only the mode marker, never written)
  a.rows2_acc = acc;
  a.H2 = H2; a.W2 = W2;
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)N * H * W;
  prof_cost(px * (CinK + CoutS) * 2.0 + (double)N * H2 * W2 * CoutS * 2.0 + 9.0 * CinK * CoutS * 2.0, 2.0 * px * 9.0 * CinK * CoutS);
  TileCfg t = pick_tile(H, W);
  if (!(t.tw == 14 && launch_conv_fast(a, t.th, st))) {
    set_error("conv3x3_dgrad_poolstats: no specialised kernel for H=%d W=%d CinK=%d CoutS=%d", H, W, CinK, CoutS);
    return SPCL_EUNSUPPORTED;
  }
  SPCL_LAUNCH_CHECK("conv3x3_dgrad_poolstats");
  return SPCL_OK;
}

