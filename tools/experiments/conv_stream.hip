// LAB RECORD (round 4), not compiled into libspcl_hip.so: it lived in csrc/ and was asked first by launch_conv_fast behind
// SPCL_CONV_STREAM (default off).  Measured: plain 16- / 32-channel convolutions 20-30 % faster in isolation, nothing inside the
// step (profiles/r04_experiments/NOTES.md).  To revive: move back to csrc/, restore the launch_conv_stream declaration / call.
// Persistent, software-pipelined variant of the one-wave-per-tile 3x3 convolution (conv_fast.hip) for the HBM-bound
// layers of the encoder (semi_seg/arch/unet.py:67-82,123-131: Conv1.b, Conv2.a/b forward and input gradients; <= 32 input
// channels, one channel slab).
//
// What conv_fast does per tile is a serial chain -- request the halo, wait an HBM round trip (44 % of the wave's life,
// profiles/r02_conv_fast_notes.md), k-loop, epilogue -- and the 2-3 waves per SIMD that fit run it in step: the memory
// system sees bursts.  Here a wave (= workgroup) walks a RUN of consecutive tiles and the NEXT tile's halo is always in
// flight while the current one is multiplied:
//   * the halo goes global -> LDS by DMA (buffer_load ... lds, 16 bytes per lane, zeros for out-of-image pixels by the
//     buffer's range check): no staging registers, so prefetching a whole tile costs nothing in the register file (the
//     register-staged "persistent tiles" of round 2 spilled: 60-140 registers);
//   * two halo images per wave; tile k+1's DMA is issued at the top of iteration k and waited for (counted vmcnt: DMA
//     instructions per tile is a compile-time constant) at the top of iteration k+1;
//   * the LDS image is DENSE (16 halo pixels per row, pixel stride = the channels: a DMA instruction writes 64 consecutive
//     16-byte slots = one row of 32-channel pixels or two rows of 16-channel ones): 9-18 KB for both images of a wave, so
//     8+ waves per CU keep a tile each in flight.  conv_fast's padded layout (22-pixel pitch, +32 bytes per pixel: free of
//     bank conflicts) would leave room for 4; these layers are bound by HBM and instruction issue, the LDS runs at ~20 %,
//     and a two-way conflict on its fragment reads is the cheaper price (measured: see DESIGN.md section 11);
//   * weight fragments and the fused input BatchNorm's coefficients are loaded ONCE per wave, not once per tile;
//   * MODE 1 (relu(bn(x)) fused into the loader): applied in place in LDS by the lanes that requested the slots;
//   * a wave's run of tiles is contiguous, and runs are dealt so that one XCD owns a contiguous eighth of the tensor
//     (both halo directions hit its L2).
// k-loop, epilogues (BatchNorm partial rows, MODE 2 / 3 backward sums, paired 16-byte stores) are conv_fast's.
#include <stdlib.h>
#include <vector>
#include "conv_common.hpp"

namespace spcl {

struct StreamArgs {
  const unsigned char* x;
  unsigned char* y;
  const u32x4* wp;
  float* stats;
  const float* in_scale;
  const float* in_shift;
  int N, H, W, CoutS, tilesX, tilesY;
  int ntiles, tpw, nwg;  // tiles in all, tiles per workgroup (a contiguous run), workgroups
  const unsigned char* y2;
  const float* scale2;
  const float* shift2;
  const float* mean2;
  float* rows2;
  int H2, W2;
  unsigned long long* stamps;  // debug builds (-DSPCL_STREAM_STAMPS_BUILD=1 + SPCL_STREAM_STAMPS=1): ticks per phase, else null
};
#ifndef SPCL_STREAM_STAMPS_BUILD
#define SPCL_STREAM_STAMPS_BUILD 0
#endif

constexpr unsigned ST_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) int st_i32x4;

__device__ __forceinline__ st_i32x4 st_rsrc(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  st_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));  // stride 0
  r[2] = 0x7fffffff;                                                             // bytes: offsets >= 2^31 read zeros
  r[3] = 0x00020000;
  return r;
}
// lane l of the wave writes LDS bytes [lds_dst + 16 l, + 16) with the 16 bytes at buffer offset voff (zeros out of range).
// Inline asm: hipcc drains a builtin LDS-DMA with vmcnt(0) before the next LDS access (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void st_dma16(st_i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_dst)
      : "memory");
}

constexpr int st_pixel_stride(int KC) { return KC * 2; }
constexpr int st_row_pitch() { return 16; }
// one halo image: whole 1 KiB DMA pieces (the last piece of an odd row count spills half a piece)
constexpr int st_img_bytes(int KC, int TH) { return ((TH + 2) * st_row_pitch() * st_pixel_stride(KC) + 1023) / 1024 * 1024; }
constexpr int st_wpe(int KC, int TH, int NT, int MODE) {
  const int by_lds = 160 * 1024 / (2 * st_img_bytes(KC, TH));  // workgroups (= waves) per CU
  int w = (by_lds + 3) / 4;
  w = w > 4 ? 4 : (w < 1 ? 1 : w);
  const int acc = (TH * 14 + 15) / 16 * NT * 4;
  if (acc > 80 && w > 2) return 2;
  if (KC == 32 && w > 2) return 2;  // a slab's 9 x NT weight fragments live in registers
  if (w > 2) return 2;              // (168 registers: the 16-channel kernels still spill a few dwords)
  return w;
}

template <int KC, int TH, int NT, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(st_wpe(KC, TH, NT, MODE)))) void
conv3x3_stream_kernel(StreamArgs a) {
  constexpr int TW = 14, HW_ = 16, CP = KC / 8, PS = st_pixel_stride(KC), RP = st_row_pitch();
  constexpr int ROWB = RP * PS, NROW = TH + 2, IMG = st_img_bytes(KC, TH);
  constexpr int RPI = 1024 / ROWB;                   // halo rows per DMA instruction: 1 (KC 32), 2 (KC 16)
  constexpr int D = (NROW + RPI - 1) / RPI;          // DMA instructions per tile (the counted wait below)
  static_assert(ROWB * RPI == 1024 && RP == HW_, "a DMA instruction covers whole dense rows");
  constexpr int NPIX = TH * TW, MT = (NPIX + 15) / 16, NSTEPS = (9 * CP + 3) / 4;
  constexpr int gps = KC * 2;                        // bytes per pixel of x
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;

  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  const int ntn = a.CoutS >> 4;

  // ---- this wave's run of tiles.  Workgroups go to the XCDs round-robin: logical index L gives XCD x the contiguous block
  // of runs [x nwg / 8, (x + 1) nwg / 8) (speed only; any mapping is correct)
  int L = blockIdx.x;
  if ((a.nwg & 7) == 0) L = (L & 7) * (a.nwg >> 3) + (L >> 3);
  int T = L * a.tpw;
  const int T_end = min(a.ntiles, T + a.tpw);
  if (T >= T_end) return;
  const int tpi = a.tilesX * a.tilesY;
  int n = T / tpi, ty, tx;
  {
    const int r = T - n * tpi;
    ty = r / a.tilesX;
    tx = r - ty * a.tilesX;
  }

  // ---- loaded once per wave: weight fragments, the fused input BatchNorm's coefficients
  u32x4 wf_all[NSTEPS][NT];
#pragma unroll
  for (int s = 0; s < NSTEPS; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j) wf_all[s][j] = a.wp[(size_t)(s * ntn + j) * 64 + lane];

  // lane's role in every DMA instruction: row rr of the instruction's RPI rows, halo column hx, channel chunk sl
  const int d_rr = lane / (HW_ * CP), d_hx = (lane / CP) % HW_, d_sl = lane % CP;
  const unsigned d_off = (unsigned)(d_hx * gps + d_sl * 16);  // byte offset of the lane's chunk from the row's first halo pixel
  float ssc[8], ssh[8];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *(f32x4*)&ssc[e] = *(const f32x4*)(a.in_scale + d_sl * 8 + e);
      *(f32x4*)&ssh[e] = *(const f32x4*)(a.in_shift + d_sl * 8 + e);
    }
  }
  const st_i32x4 rs = st_rsrc(a.x);

  auto issue_dma = [&](int tn, int tty, int ttx, int buf) {
    const int y0 = min(tty * TH, a.H - TH), x0 = min(ttx * TW, a.W - TW);
    const int gx = x0 - 1 + d_hx;
    const bool colok = gx >= 0 && gx < a.W;
    const unsigned ldst = lds_base + buf * IMG;
    int rr = d_rr;  // (opaque per call: the D per-row offsets below are tile-invariant and would be hoisted into D registers)
    asm volatile("" : "+v"(rr));
    // (unsigned arithmetic: the origin pixel (y0 - 1, x0 - 1) may lie outside; in-image lanes wrap back to the right offset)
    const unsigned obase = (unsigned)(((tn * a.H + (y0 - 1)) * a.W + (x0 - 1)) * gps) + d_off;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int hy = d * RPI + rr;  // (RPI == 1: wave-uniform)
      const int gy = y0 - 1 + hy;
      const bool ok = colok && gy >= 0 && gy < a.H && (NROW % RPI == 0 || hy < NROW);
      const unsigned vo = ok ? obase + (unsigned)(hy * a.W * gps) : ST_OOB;
      st_dma16(rs, vo, ldst + d * 1024);
    }
  };

  // per-lane LDS offset of each m-tile's pixel p = 16 i + r16 (+ the lane's k-group when a step stays inside one tap)
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = 16 * i + r16;
    if (p >= NPIX) p = 0;
    const int py = p / TW, px = p - py * TW;
    abase[i] = (py * RP + px) * PS + (CP >= 4 ? g * 16 : 0);
  }
  int gq = g;  // (made opaque once per tile: the per-lane fragment offsets of the 16-channel kernels must not be hoisted
               //  out of the tile loop as MT x NSTEPS live addresses)
  auto frag_off = [&](const int s) {
    if (CP >= 4) {
      const int fc0 = 4 * s, tap = fc0 / CP, c0 = fc0 % CP, ky = tap / 3, kx = tap % 3;
      return (ky * RP + kx) * PS + c0 * 16;  // compile-time: the ds_read offset field
    }
    int fc = 4 * s + gq;
    if (fc >= 9 * CP) fc = 0;  // K padding: the weights there are zero, any finite x will do
    const int tap = fc / CP, c = fc % CP, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    return (ky * RP + kx) * PS + c * 16;
  };

  constexpr bool M2 = MODE == 2;
  constexpr int DPY = 16 / TW, DPX = 16 % TW;
  const int rowb = a.CoutS * 2;
  f32x4 sc2[NT], sh2[NT], mu2[NT];  // MODE 2 / 3: BN coefficients of this lane's 4 channels per n-tile (tile-invariant)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (M2 || MODE == 3) {
      const int cb = j * 16 + 4 * g;
      sc2[j] = *(const f32x4*)(a.scale2 + cb);
      sh2[j] = *(const f32x4*)(a.shift2 + cb);
      mu2[j] = *(const f32x4*)(a.mean2 + cb);
    }
  }

  int buf = 0;
  const bool stamp = SPCL_STREAM_STAMPS_BUILD && a.stamps != nullptr;
  unsigned long long tk[5] = {0, 0, 0, 0, 0}, t_a = 0, t_b = 0;
  issue_dma(n, ty, tx, 0);
#pragma unroll 1
  for (; T < T_end; ++T) {
    if (stamp) t_a = __builtin_amdgcn_s_memtime();
    // next tile's coordinates; its halo is requested NOW (its buffer was last read by the k-loop of tile T - 1)
    int nn = n, nty = ty, ntx = tx + 1;
    if (ntx == a.tilesX) {
      ntx = 0;
      if (++nty == a.tilesY) {
        nty = 0;
        ++nn;
      }
    }
    const bool more = T + 1 < T_end;  // wave-uniform
    if (more) {
      issue_dma(nn, nty, ntx, buf ^ 1);
      // vector-memory operations retire in order: all but the D youngest done == this tile's halo has landed (and the previous
      // tile's epilogue stores are acknowledged)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (stamp) {  // [0] DMA issue + wait for this tile's halo
      t_b = __builtin_amdgcn_s_memtime();
      tk[0] += t_b - t_a;
      t_a = t_b;
    }
    const int y0 = min(ty * TH, a.H - TH), x0 = min(tx * TW, a.W - TW);
    const int oy = ty * TH - y0, ox = tx * TW - x0;
    const int tile = T;
    unsigned char* const img = lds + buf * IMG;
    if (CP < 4) asm volatile("" : "+v"(gq));

    if (MODE == 1) {
      // relu(scale x + shift) in place, by the lane that requested the slot (its channel chunk is fixed: coefficients in
      // registers).  Out-of-image pixels are the ACTIVATION's zero padding: left as the zeros the DMA wrote.
      // Branch-free: every lane reads ITS slot of all D pieces back to back, transforms, writes back (an `if` per piece was
      // nine exec-masked regions, each a dependent LDS round trip: the forward kernels lost what the pipeline gained).
      const int gx = x0 - 1 + d_hx;
      const bool colok = gx >= 0 && gx < a.W;
      u32x4 tv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) tv[d] = *(const u32x4*)(img + d * 1024 + lane * 16);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int hy = d * RPI + d_rr;
        const int gy = y0 - 1 + hy;
        const bool ok = colok && gy >= 0 && gy < a.H && (NROW % RPI == 0 || hy < NROW);
        const u32x4 tr = bnrelu_regs<bf16_t>(tv[d], ssc, ssh);
        tv[d] = ok ? tr : tv[d];  // (out-of-image slots hold the zeros the DMA wrote)
      }
#pragma unroll
      for (int d = 0; d < D; ++d) *(u32x4*)(img + d * 1024 + lane * 16) = tv[d];
    }

    if (stamp) {  // [1] in-place transform
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_b = __builtin_amdgcn_s_memtime();
      tk[1] += t_b - t_a;
      t_a = t_b;
    }
    // ------------ k-loop: NSTEPS x (one 16-byte x fragment per m-tile, NT MFMAs on it)
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
      const int off = frag_off(s);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 xf = *(const u32x4*)(img + abase[i] + off);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_chunk<bf16_t>(wf_all[s][j], xf, acc[i][j]);
      }
    }

    if (stamp) {  // [2] k-loop (issue; the last MFMAs drain in the epilogue's first uses)
      asm volatile("" ::"v"(acc[MT - 1][NT - 1][0]));
      t_b = __builtin_amdgcn_s_memtime();
      tk[2] += t_b - t_a;
      t_a = t_b;
    }
    // ------------ epilogue (conv_fast.hip): lane holds couts 16 j + 4 g .. +3 of pixel p = 16 i + r16
    unsigned char* yb = a.y + (((size_t)n * a.H + y0) * a.W + x0) * rowb + (4 * g) * 2;
    int py = r16 / TW, px = r16 - py * TW;
    int ob = (py * a.W + px) * rowb;
    const int dob = (DPY * a.W + DPX) * rowb, wrapo = (a.W - TW) * rowb;
    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      ssum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ssq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* y2b = nullptr;
    if (M2) y2b = a.y2 + (((size_t)n * a.H + y0) * a.W + x0) * rowb + (4 * g) * 2;
    const bool shifted = (oy | ox) != 0;  // wave-uniform
    int pyc = py;
    constexpr int NWIN = MODE == 3 ? 4 : 1;
    constexpr bool YPRE = M2 || MODE == 3;
    constexpr int YBUD = MODE == 3 ? 64 : 56, YR1 = NT * NWIN * 2;
    constexpr int GM = YBUD / (2 * YR1) > 0 ? YBUD / (2 * YR1) : 1;
    uint2 ypre[YPRE ? MT : 1][NT][NWIN];
    int lpx = px, lpyc = pyc, lob = ob;  // walker of the requests
    auto request_chunk = [&](const int c) {
#pragma unroll
      for (int ii = 0; ii < GM; ++ii) {
        const int i = c * GM + ii;
        if (i < MT) {
          const bool ok = (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);
          if (ok) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if (M2) {
                ypre[i][j][0] = *(const uint2*)(y2b + lob + j * 32);
              } else {
                const unsigned char* wb = a.y2 +
                                          ((((size_t)n * a.H2 + 2 * (y0 + lpyc)) * a.W2 + 2 * (x0 + lpx)) * rowb) +
                                          (j * 16 + 4 * g) * 2;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  ypre[i][j][k] = *(const uint2*)(wb + ((size_t)(k >> 1) * a.W2 + (k & 1)) * rowb);
              }
            }
          }
          lpx += DPX;
          lpyc += DPY;
          lob += dob;
          if (lpx >= TW) {
            lpx -= TW;
            lpyc += 1;
            lob += wrapo;
          }
        }
      }
    };
    if (YPRE) {
      request_chunk(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint2 pk_prev[NT];
    int ob_prev = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool ok = (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);  // compile-time true except in the last m-tile
      if (YPRE && i % GM == 0 && (i / GM + 1) * GM < MT) {
        request_chunk(i / GM + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ok) {
        const float keep = (!shifted || (pyc >= oy && px >= ox)) ? 1.f : 0.f;  // 0: the neighbour tile counts this pixel
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bool first = i % 2 == 0 && i + 1 < MT && 16 * (i + 1) + 15 < NPIX;
          const bool second = i % 2 == 1 && 16 * i + 15 < NPIX;
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
              *(u32x4*)(yb + ((g & 1) ? ob - 8 : ob_prev) + j * 32) = v;
            }
          } else {
            store4_fast<bf16_t>(yb + ob + j * 32, acc[i][j]);
          }
          if (M2) {
            const uint2 yr = ypre[i][j][0];
            const f32x2 glo = {acc[i][j][0], acc[i][j][1]}, ghi = {acc[i][j][2], acc[i][j][3]};
            const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
            const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
            const float yv[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u),
                                 __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
            const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                                 __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float dz = fmaf(sc2[j][r], yv[r], sh2[j][r]) > 0.f ? gv[r] * keep : 0.f;
              ssum[j][r] += dz;
              ssq[j][r] = fmaf(dz, yv[r] - mu2[j][r], ssq[j][r]);
            }
          } else if (MODE == 3) {
            const f32x2 glo = {acc[i][j][0], acc[i][j][1]}, ghi = {acc[i][j][2], acc[i][j][3]};
            const uint32_t g0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
            const uint32_t g1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
            const float gv[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                                 __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float zb = -1.f, ybst = 0.f;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const uint32_t wd = r < 2 ? ypre[i][j][k].x : ypre[i][j][k].y;
                const float yv = (r & 1) ? __uint_as_float(wd & 0xffff0000u) : __uint_as_float(wd << 16);
                const float z = fmaf(sc2[j][r], yv, sh2[j][r]);
                if (k == 0 || z > zb) {
                  zb = z;
                  ybst = yv;
                }
              }
              const float dz = zb > 0.f ? gv[r] * keep : 0.f;
              ssum[j][r] += dz;
              ssq[j][r] = fmaf(dz, ybst - mu2[j][r], ssq[j][r]);
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
      if (px >= TW) {
        px -= TW;
        pyc += 1;
        ob += wrapo;
      }
    }
    if (M2 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = row16_sum(ssum[j][r]), s2 = row16_sum(ssq[j][r]);
          o[r] = r16 == 0 ? s1 : s2;
        }
        if (r16 < 2) *(f32x4*)(a.rows2 + ((size_t)tile * 2 + r16) * a.CoutS + j * 16 + 4 * g) = o;
      }
    } else if (a.stats != nullptr) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        write_tile_stats(a.stats, tile, a.CoutS, j * 16 + 4 * g, r16, (float)((TH - oy) * (TW - ox)), ssum[j], ssq[j]);
    }
    if (stamp) {  // [3] epilogue (stores issued, not acknowledged)
      t_b = __builtin_amdgcn_s_memtime();
      tk[3] += t_b - t_a;
      tk[4] += 1;
    }
    buf ^= 1;
    n = nn;
    ty = nty;
    tx = ntx;
  }
  if (stamp && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) a.stamps[(size_t)blockIdx.x * 5 + k] = tk[k];
  }
}

template <int KC, int TH, int NT>
static void launch_stream(const StreamArgs& a0, int mode, hipStream_t st) {
  StreamArgs a = a0;
  const int lds = 2 * st_img_bytes(KC, TH);
  static const int env_wpc = lab_env("SPCL_CONV_STREAM_WPC", 0);
  auto go = [&](auto kern, int wpe) {
    int wpc = 160 * 1024 / lds;             // resident workgroups (= waves) per CU: LDS ...
    if (wpc > 4 * wpe) wpc = 4 * wpe;       // ... and the register budget the kernel was compiled for
    if (env_wpc > 0 && env_wpc < wpc) wpc = env_wpc;
    const int slots = 256 * wpc;
    // equal runs: every wave walks ceil(ntiles / rounds-worth) tiles, the grid is what that needs (a multiple of 8 so that
    // the XCD remap is a bijection)
    int tpw = (a.ntiles + slots - 1) / slots;
    if (tpw < 1) tpw = 1;
    int nwg = (a.ntiles + tpw - 1) / tpw;
    nwg = (nwg + 7) / 8 * 8;
    a.tpw = tpw;
    a.nwg = nwg;
    func_lds_limit((const void*)kern, lds, "conv3x3_stream_kernel");
    a.stamps = nullptr;
    static const bool env_stamps = SPCL_STREAM_STAMPS_BUILD && lab_flag("SPCL_STREAM_STAMPS");
    if (env_stamps) {  // debug only (synchronises)
      (void)hipMalloc(&a.stamps, (size_t)nwg * 5 * 8);
      (void)hipMemset(a.stamps, 0, (size_t)nwg * 5 * 8);
    }
    SPCL_LAUNCH(kern, dim3(nwg), dim3(64), lds, st, a);
    if (a.stamps != nullptr) {
      std::vector<unsigned long long> h((size_t)nwg * 5);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(a.stamps);
      double s5[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 5; ++k) s5[k] += (double)h[(size_t)i * 5 + k];
      const double nt = s5[4] > 0 ? s5[4] : 1;
      fprintf(stderr, "[conv_stream stamps] <%d,%d,%d,m%d> %dx%d CoutS %d wgs %d tpw %d wpc %d | ticks per tile: DMA issue + wait "
              "%.0f, transform %.0f, k-loop %.0f, epilogue %.0f\n", KC, TH, NT, mode == 1 ? 1 : (a.rows2 ? (a.H2 > 0 ? 3 : 2) : 0),
              a.H, a.W, a.CoutS, nwg, tpw, wpc, s5[0] / nt, s5[1] / nt, s5[2] / nt, s5[3] / nt);
    }
  };
  if (mode == 1) go(conv3x3_stream_kernel<KC, TH, NT, 1>, st_wpe(KC, TH, NT, 1));
  else if (a.rows2 != nullptr && a.H2 > 0) go(conv3x3_stream_kernel<KC, TH, NT, 3>, st_wpe(KC, TH, NT, 3));
  else if (a.rows2 != nullptr) go(conv3x3_stream_kernel<KC, TH, NT, 2>, st_wpe(KC, TH, NT, 2));
  else go(conv3x3_stream_kernel<KC, TH, NT, 0>, st_wpe(KC, TH, NT, 0));
}

// conv_fast.hip's launcher asks here first: true when the streaming kernel took the launch
bool launch_conv_stream(const ConvArgs& c, int th, hipStream_t st, bool dry) {
  if (c.x2 != nullptr || c.y_hi != nullptr || c.x_up2) return false;  // (two-tensor inputs / outputs exist in conv_fast.hip only)
  // 0 (default) off; 1 every supported shape; 2 only plain 32-input-channel convolutions (no fused input BatchNorm, no
  // BatchNorm-backward sums in the epilogue).  Isolated launches of those run 20-30 % faster than conv_fast, but neither the
  // pre-train step (1.123 -> 1.138 ms with 1) nor the fine-tune step (2.451 -> 2.469 ms with 2) gains: the tile is bound by
  // the instructions it executes, not by the exposed halo latency (profiles/r04_experiments/NOTES.md)
  static const int env_on = lab_env("SPCL_CONV_STREAM", 0);
  if (!env_on) return false;
  if (env_on == 2 && !(c.CinK == 32 && c.in_mode == 0 && c.rows2 == nullptr)) return false;
  if (env_on == 3 && !(c.in_mode == 0 && c.rows2 == nullptr)) return false;  // 3: every plain convolution it has a kernel for
  if (c.H < th || c.W < 14 || c.in_mode == 2 || c.CinS != c.CinK || c.img2 != nullptr) return false;
  const int KC = c.CinK, ntn = c.CoutS / 16;
  if (KC != 16 && KC != 32) return false;
  if (ntn < 1 || ntn > 2) return false;
  if (c.rows2 != nullptr && c.in_mode != 0) return false;
  if ((double)c.N * c.H * c.W * KC * 2 >= 2147483648.0) return false;  // 32-bit buffer offsets
  StreamArgs a;
  a.x = (const unsigned char*)c.x; a.y = (unsigned char*)c.y; a.wp = (const u32x4*)c.wp; a.stats = c.stats;
  a.in_scale = c.in_scale; a.in_shift = c.in_shift;
  a.y2 = (const unsigned char*)c.y2; a.scale2 = c.scale2; a.shift2 = c.shift2; a.mean2 = c.mean2; a.rows2 = c.rows2;
  a.H2 = c.H2; a.W2 = c.W2;
  a.N = c.N; a.H = c.H; a.W = c.W; a.CoutS = c.CoutS;
  a.tilesX = cdiv(c.W, 14); a.tilesY = cdiv(c.H, th);
  a.ntiles = a.N * a.tilesX * a.tilesY;
  a.tpw = 1; a.nwg = 0; a.stamps = nullptr;
#define SPCL_STREAM_CASE(KC_, TH_, NT_)                          \
  if (KC == KC_ && th == TH_ && ntn == NT_) {                    \
    if (!dry) launch_stream<KC_, TH_, NT_>(a, c.in_mode, st);    \
    return true;                                                 \
  }
  SPCL_STREAM_CASE(16, 14, 1)  // Conv1.b forward (16 -> 16 @ 224^2)
  SPCL_STREAM_CASE(16, 7, 2)   // Conv2.a forward (16 -> 32 @ 112^2)
  SPCL_STREAM_CASE(32, 7, 2)   // Conv2.b forward / dgrad (32 -> 32 @ 112^2)
  SPCL_STREAM_CASE(32, 7, 1)   // Conv2.a dgrad (32 -> 16 @ 112^2)
  SPCL_STREAM_CASE(32, 14, 1)  // Up_conv2.a forward (cat(16, 16) -> 16 @ 224^2)
#undef SPCL_STREAM_CASE
  return false;
}

}  // namespace spcl
