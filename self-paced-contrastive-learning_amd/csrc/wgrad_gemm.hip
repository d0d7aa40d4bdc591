// Weight gradients of the 3x3 convolutions whose channel counts are multiples of 64 (semi_seg/arch/unet.py:72,75 at
// _Conv3.b ... _Conv5.b and the decoder's wide layers; autograd backward), SEVERAL LAYERS IN ONE LAUNCH, bf16 on gfx950.
//
//   dW[tap][ci][co] = sum_pixels act(x)[p + tap][ci] * dy[p][co]          (GEMM per tap, K = pixels)
//
// Why batched.  One workgroup per CU holds a [9 taps][64 ci][64 co] f32 accumulator block in registers (8 waves x 72
// accumulator VGPRs): all nine taps read the SAME staged input tile, so the operand traffic from L2 is 288 FLOP per byte.
// The price of that block is the split-K partial it must write at the end: 147 KB per workgroup, 37.7 MB per launch
// with the 256 CUs busy -- as much as a layer's inputs.  Launched per layer that was 188 MB of partials written and
// re-read per step (r01: the slabs, not the MFMA loop, bounded the kernel).  Here the weight gradients of all >=64-channel
// layers of a backward pass are queued and computed by ONE launch whose workgroups split the UNION of the layers' pixel
// ranges: still one slab per CU, i.e. 37.7 MB in total instead of per layer, and 16 pixel tiles per workgroup instead
// of 4 between slab writes.
//
// Kernel.  Tile = TH x 16 output pixels (+ halo) of one image.  dy arrives by LDS-DMA (global_load_lds_dwordx4: 8 pixels
// x 128 B per wave-instruction, whole 128-byte lines, no VGPRs) into one of three images; x goes through registers
// (16-byte loads one tile ahead, the producer layer's BatchNorm-apply + ReLU where in_mode == 1, swizzled LDS writes) into
// one of two images.  The MFMA operands have K = pixels, so they are read
// TRANSPOSED from the pixel-major images with ds_read_b64_tr_b16; the 32-byte slot of a 16-channel group inside a pixel's
// 128 bytes is XOR-swizzled with the pixel's column (slot = group ^ Fc(col)) so that the 8 pixels one transposed read
// services (two runs of 4 consecutive columns, 8 apart) cover all 64 banks exactly once at any tap shift; with LDS-DMA
// the same permutation is applied to the SOURCE address (the DMA writes lane-linear).  Consumer wave (m, h) owns ci-tile m
// and co-tiles 2h, 2h+1 for all nine taps: per 32-pixel k-step 2 dy fragments + 9 tap-shifted x fragments feed 18
// v_mfma_f32_16x16x32_bf16; four producer waves do the staging (see the kernel).  Partials leave in fragment order (1 KiB per wave-store); a second kernel sums the slabs of
// each block in fixed order (deterministic, no float atomics) into the OIHW f32 gradients.
#include <stdlib.h>
#include <type_traits>
#include <string.h>
#include "conv_common.hpp"

namespace spcl {

// the ablation bits (WbArgs.dbg) and the in-kernel stamps act only in a -DSPCL_WGRAD_GEMM_DBG_BUILD=1 build: run-time
// conditions in the tile loops are scalar branches on every trip
#ifndef SPCL_WGRAD_GEMM_DBG_BUILD
#define SPCL_WGRAD_GEMM_DBG_BUILD 0
#endif
#define WB_DBG(a) (SPCL_WGRAD_GEMM_DBG_BUILD ? (a).dbg : 0)
constexpr int WB_MAX = SPCL_WGRAD_BATCH_MAX;
constexpr int WB_SLAB = 9 * 64 * 64;  // floats per workgroup partial


struct WbItem {
  const bf16_t* x;
  const bf16_t* x2;  // non-null: channels [Cin / 2, Cin) of the input live here, [0, Cin / 2) in x (pixel stride Cin / 2 each)
  int x_up2;         // 1: x is [N][H / 2][W / 2][CinS], the layer's input its nearest x2 upsample
  const bf16_t* dy;
  const float* in_scale;
  const float* in_shift;
  float* dw;
  int N, H, W, Cin, Cout, CinS, CoutS, in_mode;
  int tilesX, tpi, ntiles;       // tiles per row, per image, in all
  int nblk_co, nblk, nsplit;     // 64-channel output blocks (ci-major), pixel splits per block
  int wg0;                       // first workgroup of the item
  int e0;                        // first workgroup of the item in the reduce kernel
};
struct WbArgs {
  WbItem it[WB_MAX];
  int n, accumulate;
  float* partial;
  unsigned long long* stamps;  // debug (SPCL_WGRAD_GEMM_STAMPS=1): per-workgroup cycle sums of the loop phases, else null
  int dbg;  // experiments only (SPCL_WGRAD_GEMM_DBG, 0 in production): 1 no staging, 2 no MFMA loop, 4 no x transform, 8 no slab write
};

// pending final sums of the narrow layers (spcl_wgrad_tail), finished by extra workgroups of the reduce kernel
struct WbTail {
  const float* partial;
  float* dw;
  int kind, nsplit, nblk_ci, nblk_co, CIB, COB, Cin, Cout;
  int u0;  // first tail unit of this entry
};
struct WbTails {
  WbTail t[SPCL_WGRAD_TAILS_MAX];
  int n;
  int first;  // number of tail units = blockIdx.x of the first unit of the wide layers (the tails are dispatched first)
};

// 32-byte slot permutation of a pixel column (see the header): distinct for columns c, c+2, c+8, c+10 of equal parity
__host__ __device__ __forceinline__ int wb_fc(int col) { return ((col >> 1) & 3) ^ (((col >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x8 wb_frag(unsigned addr_lo, unsigned addr_hi, int off) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(addr_lo + off));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)(addr_hi + off));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Raw buffer descriptor over [base, base + 2 GiB): a lane whose byte offset is >= 0x7fffffff (WB_OOB) reads zeros -- the
// zero padding of the convolution and the ragged tile edges cost one select per chunk instead of a branch and a
// 64-bit address per lane.  Wave-uniform by construction (base comes from scalar tile arithmetic).
constexpr unsigned WB_OOB = 0x80000000u;
__device__ __forceinline__ i32x4 wb_rsrc(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));  // stride 0
  r[2] = 0x7fffffff;                                                             // bytes
  r[3] = 0x00020000;
  return r;
}

// LDS-DMA of 16 bytes per lane: lane l of the wave writes LDS byte lds_dst + 16 l (M0 = wave-uniform base) with the 16 bytes
// at buffer offset voff (zeros when out of range).  Inline asm on purpose: hipcc drains a builtin LDS-DMA with
// s_waitcnt vmcnt(0) before the next LDS access it cannot prove disjoint (cdna_hip_programming.md 5.7); the producers
// order their own DMA pieces (see the schedule in the kernel).
__device__ __forceinline__ void wb_dma16(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_dst)
      : "memory");
}

// Workgroup = 12 waves: 8 CONSUMERS (wave (m, h): ci-tile m, co-tiles 2h, 2h+1, all nine taps; nothing but transposed
// LDS reads and MFMAs) and 4 PRODUCERS (one per SIMD: all staging of the next tile -- LDS-DMA issue, the register loads,
// the BatchNorm + ReLU transform and the LDS writes).  One workgroup barrier per tile; between two barriers the consumers
// run the MFMAs of tile t out of one LDS buffer while the producers fill the other with tile t+1, so the vector ALU work
// of the staging (4 VALU per MFMA when every wave did both in turn: PMC, profiles/r02_wgrad_gemm_notes.md) issues BESIDE
// the matrix pipe instead of in its own phase.
template <int TH, int RING>
__global__ __launch_bounds__(768) void wgrad_gemm_kernel(WbArgs a) {
  constexpr int XROW = 18 * 128;                       // bytes per halo row (a multiple of the 256-byte bank row)
  constexpr int XPIX = (TH + 2) * 18, XGRP = (XPIX + 7) / 8, X_BYTES = XGRP * 1024;
  constexpr int DROW = 16 * 128;
  constexpr int DPIX = TH * 16, DGRP = DPIX / 8, D_BYTES = DGRP * 1024;
  // LDS: two x images and NDB dy images (three where they fit: TH = 14 -> 159 744 B of the 163 840)
  constexpr int NDB = (2 * X_BYTES + 3 * D_BYTES <= 163840) ? 3 : 2;
  constexpr int D_BASE = 2 * X_BYTES;
  constexpr int NXI = (XGRP + 3) / 4, NDI = (DGRP + 3) / 4;  // producer staging iterations (32 pixels x 8 chunks each)
  constexpr int KSTEPS = TH / 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;

  // ---- which (layer, block, pixel split) is this workgroup's.  Workgroups are dealt round-robin over the 8 XCDs (speed
  // only, never correctness): the logical index is remapped so that one XCD gets CONSECUTIVE logical workgroups, and the
  // blocks of one pixel split are consecutive -- the 2..16 workgroups that read the same pixels of x / dy (different
  // channel blocks) then share an L2 and run in step, so each byte leaves HBM once instead of once per block.
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  int idx = 0;
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if (wg >= a.it[i].wg0) idx = i;
  const WbItem& it = a.it[idx];
  const int local = wg - it.wg0;
  const int split = local / it.nblk, blk = local - split * it.nblk;
  const int bci = blk / it.nblk_co, bco = blk - bci * it.nblk_co;
  const int t_begin = (int)((long)split * it.ntiles / it.nsplit), t_end = (int)((long)(split + 1) * it.ntiles / it.nsplit);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (wave >= 8) {
    // =============================================================================================== producers
    const int H = it.H, W = it.W, CoutS = it.CoutS;
    const bool fused = it.in_mode == 1;
    // (two input tensors: a 64-channel block lies in one of them whole -- a workgroup-uniform choice)
    const bool xtwo = it.x2 != nullptr;
    const int CinS = xtwo ? it.Cin / 2 : it.CinS;
    const bf16_t* xg = xtwo ? (bci * 64 >= CinS ? it.x2 + (bci * 64 - CinS) : it.x + bci * 64) : it.x + bci * 64;
    const bf16_t* dyg = it.dy + bco * 64;
    const int pw = wave - 8;
    const int c = tid & 7, pb = (tid - 512) >> 3;  // 16-byte chunk of a pixel, pixel within a 32-pixel staging iteration
    // per-thread staging constants (tile independent): halo / tile coordinates (packed), byte offsets from the tile's
    // origin pixel, LDS targets.  x goes through registers: the thread keeps channels 8c..8c+7 (its BN coefficients), the
    // swizzle picks the LDS position; dy goes by LDS-DMA: the lane's LDS position is fixed, the swizzle picks the SOURCE
    // channel group.
    int xyx[NXI];
    unsigned xdst[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int p = pb + 32 * i;
      const int hy = p / 18, hx = p - 18 * hy;
      xyx[i] = p < XPIX ? (hy << 8) | hx : (127 << 8);  // rounded-up last group: row 127 is outside every tile
      xdst[i] = (unsigned)(p * 128 + ((c >> 1) ^ wb_fc(hx)) * 32 + (c & 1) * 16);
    }
    const int xc2 = c * 16, xrow2 = W * CinS * 2, xcol2 = CinS * 2;  // byte offset = hy * xrow2 + hx * xcol2 + xc2
    // dy: pixel p = pb + 32 i sits in tile row (pb >> 4) + 2 i, column pb & 15 (the same for every i)
    const int dcol = pb & 15, drow0 = pb >> 4;
    const unsigned doff0 = (unsigned)(((drow0 * W + dcol) * CoutS + ((c >> 1) ^ wb_fc(dcol)) * 16 + (c & 1) * 8) * 2);
    const unsigned dstep = (unsigned)(2 * W * CoutS * 2);
    float sc[8], sh[8];
    if (fused) {
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        *(f32x4*)&sc[e] = *(const f32x4*)(it.in_scale + bci * 64 + c * 8 + e);
        *(f32x4*)&sh[e] = *(const f32x4*)(it.in_shift + bci * 64 + c * 8 + e);
      }
    }
    // tile cursors (image, tile row, tile column), advanced by one tile per use: one for the x stream, one for dy
    struct Cursor { int n, ty, tx; };
    auto cursor_at = [&](int tile) {
      Cursor cu;
      cu.n = tile / it.tpi;
      const int rem = tile - cu.n * it.tpi;
      cu.ty = rem / it.tilesX;
      cu.tx = rem - cu.ty * it.tilesX;
      return cu;
    };
    const int tilesY = it.tpi / it.tilesX;
    auto advance = [&](Cursor& cu) {
      if (++cu.tx == it.tilesX) {
        cu.tx = 0;
        if (++cu.ty == tilesY) {
          cu.ty = 0;
          ++cu.n;
        }
      }
    };
    Cursor cx = cursor_at(t_begin), cd = cx;
    // dy of the cursor's tile -> dy image `j` by LDS-DMA (zeros beyond the image edge)
    auto dma_dy = [&](int j) {
      const int y0 = cd.ty * TH, x0 = cd.tx * 16;
      const i32x4 rs = wb_rsrc(dyg + (((long)cd.n * H + y0) * W + x0) * CoutS);
      const unsigned bd = lds_base + (unsigned)(D_BASE + j * D_BYTES);
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        if (pw + 4 * i < DGRP) {  // wave-uniform
          const bool ok = drow0 + 2 * i < TH && y0 + drow0 + 2 * i < H && x0 + dcol < W;
          wb_dma16(rs, ok ? doff0 + i * dstep : WB_OOB, __builtin_amdgcn_readfirstlane(bd + (pw + 4 * i) * 1024));
        }
      }
      advance(cd);
    };
    // x of the cursor's tile -> registers (16-byte buffer loads, zeros outside the image; the validity bits travel with
    // the data because zero padding must stay zero AFTER BatchNorm + ReLU)
    u32x4 rx[NXI];
    unsigned xmask = 0;
    auto load_x = [&]() {
      const int y0 = cx.ty * TH - 1, x0 = cx.tx * 16 - 1;
      const bool up2 = it.x_up2 != 0;  // (fine pixel (gy, gx) = pixel (gy >> 1, gx >> 1) of the image's half-resolution plane)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(up2 ? xg + (long)cx.n * (H >> 1) * (W >> 1) * CinS : xg + (((long)cx.n * H + y0) * W + x0) * CinS), 0,
          0x7fffffff, 0x00020000);
      xmask = 0;
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        if (XGRP % 4 == 0 || pw + 4 * i < XGRP) {  // wave-uniform; always true where the groups divide by the 4 waves
          const int gy = y0 + (xyx[i] >> 8), gx = x0 + (xyx[i] & 255);
          const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
          xmask |= (ok ? 1u : 0u) << i;
          const unsigned off = up2 ? (unsigned)((((gy >> 1) * (W >> 1)) + (gx >> 1)) * xcol2 + xc2)
                                   : (unsigned)((xyx[i] >> 8) * xrow2 + (xyx[i] & 255) * xcol2 + xc2);
          rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : WB_OOB), 0, 0);
        }
      }
      advance(cx);
    };
    // registers -> x image `b` (the producer layer's BatchNorm-apply + ReLU where the layer has it), swizzled 16-byte writes
    auto write_x = [&](int b, auto fused_c) {
      unsigned char* bx = lds + b * X_BYTES;
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        if (XGRP % 4 == 0 || pw + 4 * i < XGRP) {
          u32x4 v = rx[i];
          if (decltype(fused_c)::value) {
            v = bnrelu_regs<bf16_t>(v, sc, sh);
            if (!(xmask & (1u << i))) v = (u32x4){0u, 0u, 0u, 0u};
          }
          *(u32x4*)(bx + xdst[i]) = v;
        }
      }
    };
    // Schedule (k = tile - t_begin; the consumers run tile k between barrier k and barrier k + 1, reading X[k & 1] and
    // D[k % NDB]).  Producer interval k, three dy images:
    //     x(k + 1) registers -> X[(k + 1) & 1]  |  request x(k + 2)  |  counted wait: everything older than those requests
    //     has arrived, in particular dy(k + 1), whose DMA went out at the end of interval k - 1  |  DMA dy(k + 2) -> the
    //     image the consumers left at barrier k  |  barrier k + 1.
    // x registers and dy pieces of LATER tiles therefore stay in flight across the barriers.  hipcc counts only its own
    // loads, so the waits it places before a register's use cover up to NDI operations more than needed -- with this
    // order those are operations of the previous interval.  Two dy images (TH = 16): dy(k + 1) is issued at the START of
    // interval k instead.
    // (Measured dead ends, profiles/r02_wgrad_gemm_notes.md: refilling each register right after its use, and a second
    // register set requested before the transform -- hipcc answers both with spills and s_waitcnt vmcnt(0) at the joins.)
    typedef std::true_type Yes;
    typedef std::false_type No;
    const int nt = t_end - t_begin;
    const bool go = !(WB_DBG(a) & 1);
    const bool fuse = fused && !(WB_DBG(a) & 4);
    unsigned long long pt_write = 0, pt_issue = 0, pt_bar = 0;
    const bool stamp = SPCL_WGRAD_GEMM_DBG_BUILD && a.stamps != nullptr;
    __builtin_amdgcn_s_setprio(3);  // four short instruction streams beside eight MFMA streams: never wait for issue
    if (nt > 0 && go) {
      if (NDB == 3) dma_dy(0);
      load_x();  // x(0)
    }
    for (int k = -1; k < nt - 1; ++k) {
      const unsigned long long s0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
      unsigned long long s1 = s0, s2 = s0;
      if (go) {
        if (NDB == 2) dma_dy((k + 1) & 1);
        if (fuse) write_x((k + 1) & 1, Yes());
        else write_x((k + 1) & 1, No());
        if (stamp) s1 = __builtin_amdgcn_s_memtime();
        if (k + 2 < nt) {
          load_x();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XGRP / 4) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (NDB == 3 && k + 2 < nt) dma_dy((k + 2) % 3);
        if (stamp) s2 = __builtin_amdgcn_s_memtime();
      }
      __syncthreads();  // barrier k + 1: tile k + 1 is staged
      if (stamp) {
        pt_write += s1 - s0; pt_issue += s2 - s1; pt_bar += __builtin_amdgcn_s_memtime() - s2;
      }
    }
    if (stamp && tid == 512) {
      unsigned long long* o = a.stamps + (size_t)blockIdx.x * 8;
      o[0] = pt_write; o[1] = pt_issue; o[2] = pt_bar; o[3] = (unsigned long long)nt;
    }
    return;
  }

  // ================================================================================================= consumers
  // per-lane operand addresses (buffer 0; k-step / tap rows are immediate offsets).  Lane (g = lane>>4, r = lane&15) of a
  // transposed read fetches channels 4(r&3)..+3 of pixel column colL + delta in tile row (g>>1) and receives channel r
  // of the 4 columns of its 16-lane group: k = 8g + j  <->  row g>>1, column 8(g&1) + j.
  const int g = lane >> 4, r16 = lane & 15;
  const int m = wave & 3, h = wave >> 2;
  const int colL = 8 * (g & 1) + (r16 >> 2);
  unsigned xa[6], da[2][2];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int delta = d < 3 ? d : d + 1;  // 4s + kx: 0,1,2,4,5,6
    const int col = colL + delta;
    xa[d] = lds_base + (g >> 1) * XROW + col * 128 + ((m ^ wb_fc(col)) * 32) + (r16 & 3) * 8;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = colL + 4 * s;
      da[s][j] = lds_base + D_BASE + (g >> 1) * DROW + col * 128 + (((2 * h + j) ^ wb_fc(col)) * 32) + (r16 & 3) * 8;
    }

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // MFMA loop of one tile: groups of 6 MFMAs (one k-step x one tap row).  The three x fragments of the NEXT group (and
  // the two dy fragments of the next k-step) are requested before the current group's MFMAs, into the other half of a
  // two-deep register ring, so an LDS read has a whole group (and the partner wave's) to come back; the scheduling
  // barriers keep hipcc from folding the ring back into load-wait-use on one register set.
  auto compute = [&](int k) {
    constexpr int NG = KSTEPS * 3;
    // RING = groups of x fragments in flight ahead of the MFMAs.  Buffer base folded into the per-lane addresses ONCE per
    // tile and hidden from the optimiser (else it re-associates base + k-step/tap offset into a scalar and spends a
    // v_add per read instead of the ds_read offset field)
    unsigned xb[6], db[2][2];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      xb[d] = xa[d] + (unsigned)((k & 1) * X_BYTES);
      asm volatile("" : "+v"(xb[d]));
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        db[s2][j] = da[s2][j] + (unsigned)((k % NDB) * D_BYTES);
        asm volatile("" : "+v"(db[s2][j]));
      }
    bf16x8 bfr[2][2], afr[RING + 1][3];
    auto load_a = [&](int gi) {
      const int ks = gi / 3, ky = gi - 3 * ks;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) afr[gi % (RING + 1)][kx] = wb_frag(xb[kx], xb[3 + kx], (2 * ks + ky) * XROW);
      if (ky == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[ks & 1][j] = wb_frag(db[0][j], db[1][j], 2 * ks * DROW);
      }
    };
#pragma unroll
    for (int gi = 0; gi < RING; ++gi) load_a(gi);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int ks = gi / 3, ky = gi - 3 * ks;
      const bool ld = gi + RING < NG;
      const bool ldb = ld && (gi + RING) % 3 == 0;
      if (ld) load_a(gi + RING);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[ky * 3 + kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[gi % (RING + 1)][kx], bfr[ks & 1][j],
                                                                        acc[ky * 3 + kx][j], 0, 0, 0);
      // issue order of the group: one transposed read (two where the dy fragments ride along) in the shadow of each MFMA
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (ld) {
          if (ldb && i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  unsigned long long ct_bar = 0, ct_comp = 0;
  const bool cstamp = SPCL_WGRAD_GEMM_DBG_BUILD && a.stamps != nullptr;
  for (int k = 0; k < t_end - t_begin; ++k) {
    const unsigned long long s0 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();  // barrier k: the producers have staged tile k (and every consumer has left tile k - 1)
    const unsigned long long s1 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
    if (!(WB_DBG(a) & 2)) compute(k);
    if (cstamp) {
      ct_bar += s1 - s0; ct_comp += __builtin_amdgcn_s_memtime() - s1;
    }
  }
  if (cstamp && tid == 0) {
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 8;
    o[4] = ct_bar; o[5] = ct_comp;
  }

  // ---- partial slab in fragment order: [tap][m][co-tile][lane][4 rows]  (1 KiB per wave-store)
  float* out = a.partial + ((size_t)it.wg0 + (size_t)blk * it.nsplit + split) * WB_SLAB;  // splits of a block adjacent
  if (WB_DBG(a) & 8) return;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(f32x4*)(out + ((((t * 4 + m) * 4 + (2 * h + j)) * 64) + lane) * 4) = acc[t][j];
}

// dW_oihw[co][ci][tap] (+)= sum over the pixel splits of a block, fixed order.  Workgroup = (block, ci-tile m, co-tile): 9
// waves, wave = tap, lane = fragment lane; a thread sums its float4 (4 consecutive ci of one co, one tap) over the splits
// with coalesced 16-byte loads (1 KiB per wave and split).  The 16 ci x 16 co x 9 taps go through LDS into OIHW order,
// where the 16 ci x 9 taps of one co are 144 consecutive floats: the gradient leaves in 576-byte runs.
// Tail units (blockIdx.x >= tl.first): kind 0 = 64 consecutive outputs of one (ci, co) block of a narrow layer's split
// partials -- 9 waves x 4 quarter-waves of 16-byte loads, combined by two shuffles and through LDS in wave order; kind 1 = one tap of the first layer's per-workgroup rows.  Fixed order -> deterministic.
__device__ __forceinline__ void wb_tail_unit(const WbTails& tl, int unit, int accumulate, float* red) {
  int idx = 0;
#pragma unroll 1
  for (int i = 1; i < tl.n; ++i)
    if (unit >= tl.t[i].u0) idx = i;
  const WbTail& t = tl.t[idx];
  const int u = unit - t.u0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (t.kind == 0) {
    // unit = 64 consecutive outputs of one block: a quarter-wave loads them as 16 x 16 bytes (two whole lines), the four
    // quarters of the nine waves take the splits p = 4 w + q, + 36, ... (8 loads in flight each).  (One output per lane and
    // one split per wave: a thread walked nsplit / 9 splits, 21 round trips for the 1 536 slabs of conv16_bwd.hip -- the
    // launch's critical path; 16 outputs per quarter-wave instead: half-used lines, slower still.)
    const int slab = 9 * t.CIB * t.COB;  // a multiple of 64 (CIB, COB multiples of 16)
    const int chunks = slab >> 6;
    const int blk = u / chunks, o16 = lane & 15, q = lane >> 4;
    const int inner0 = (u - blk * chunks) * 64;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    {
      const size_t stride = (size_t)t.nblk_ci * t.nblk_co * slab;
      const float* src = t.partial + (size_t)blk * slab + inner0 + 4 * o16;
#pragma unroll 8
      for (int p = 4 * wave + q; p < t.nsplit; p += 36) s += *(const f32x4*)(src + (size_t)p * stride);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // quarters (0 + 2), (1 + 3), then their sum: fixed order
      s[e] += __shfl_down(s[e], 32, 64);
      s[e] += __shfl_down(s[e], 16, 64);
    }
    if (lane < 16) *(f32x4*)(red + wave * 64 + 4 * lane) = s;
    __syncthreads();
    if (wave == 0) {
      const int inner = inner0 + lane;
      float v = red[lane];
#pragma unroll
      for (int w = 1; w < 9; ++w) v += red[w * 64 + lane];
      const int bci = blk / t.nblk_co, bco = blk - bci * t.nblk_co;
      const int co_l = inner % t.COB, ci_l = (inner / t.COB) % t.CIB, tap = inner / (t.COB * t.CIB);
      const int ci = bci * t.CIB + ci_l, co = bco * t.COB + co_l;
      if (ci < t.Cin && co < t.Cout) {
        float* dst = t.dw + ((size_t)co * t.Cin + ci) * 9 + tap;
        *dst = accumulate ? *dst + v : v;
      }
    }
  } else if (t.kind == 2) {
    const int slab = 9 * t.CIB * t.COB, per = slab / 2304;
    const int blk = u / per;
    const int inner0 = (u - blk * per) * 2304 + 4 * (int)threadIdx.x;
    const size_t stride = (size_t)t.nblk_ci * t.nblk_co * slab;
    const float* src = t.partial + (size_t)blk * slab + inner0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int p = 0; p < t.nsplit; ++p) s += *(const f32x4*)(src + (size_t)p * stride);
    const int bci = blk / t.nblk_co, bco = blk - bci * t.nblk_co;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int inner = inner0 + e;
      const int co_l = inner % t.COB, ci_l = (inner / t.COB) % t.CIB, tap = inner / (t.COB * t.CIB);
      const int ci = bci * t.CIB + ci_l, co = bco * t.COB + co_l;
      if (ci < t.Cin && co < t.Cout) {
        float* dst = t.dw + ((size_t)co * t.Cin + ci) * 9 + tap;
        *dst = accumulate ? *dst + s[e] : s[e];
      }
    }
  } else {
    const int CS = t.COB, RL = 576 / CS;  // CS <= 256: at least two row lanes
    const int c = threadIdx.x % CS, rl = threadIdx.x / CS;
    float s = 0.f;
    if (rl < RL) {
#pragma unroll 8
      for (int w = rl; w < t.nsplit; w += RL) s += t.partial[((size_t)w * 9 + u) * CS + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < CS && threadIdx.x < t.Cout) {
      float tot = 0.f;
      for (int k = 0; k < RL; ++k) tot += red[k * CS + threadIdx.x];
      float* dst = t.dw + threadIdx.x * 9 + u;
      *dst = accumulate ? *dst + tot : tot;
    }
  }
}

__global__ __launch_bounds__(576) void wgrad_gemm_reduce_kernel(WbArgs a, WbTails tl) {
  __shared__ __attribute__((aligned(16))) float tile[16 * 16 * 9 + 16];
  // The tail units come FIRST in dispatch order: theirs are the long chains (a unit of the first block's one-pass backward
  // walks its 2 048 slabs as 36 streams of 57 sixteen-byte loads), and dispatched behind the ~2 000 short units of the wide
  // layers they were the launch's critical path
  if ((int)blockIdx.x < tl.first) {
    wb_tail_unit(tl, blockIdx.x, a.accumulate, tile);
    return;
  }
  int idx = 0;
  const int wgu = blockIdx.x - tl.first;  // unit index over all items: e0 counts (block, m, co-tile) units here
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if (wgu >= a.it[i].e0) idx = i;
  const WbItem& it = a.it[idx];
  const int u = wgu - it.e0;
  const int blk = u >> 4, m = (u >> 2) & 3, cot = u & 3;
  const int tap = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* src = a.partial + ((size_t)it.wg0 + (size_t)blk * it.nsplit) * WB_SLAB +
                     (size_t)((((tap * 4 + m) * 4 + cot) * 64) + lane) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int p = 0; p < it.nsplit; ++p) s += *(const f32x4*)(src + (size_t)p * WB_SLAB);
  const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[(c16 * 16 + 4 * g + r) * 9 + tap] = s[r];
  __syncthreads();
  const int bci = blk / it.nblk_co, bco = blk - bci * it.nblk_co;
  const int ci0 = bci * 64 + m * 16, co0 = bco * 64 + cot * 16;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = threadIdx.x + 576 * k;  // 2304 outputs
    const int co_l = e / 144, rest = e - co_l * 144;
    float* dst = it.dw + ((size_t)(co0 + co_l) * it.Cin + ci0) * 9 + rest;
    const float v = tile[e];
    *dst = a.accumulate ? *dst + v : v;
  }
}

static int wb_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (lab_flag("SPCL_WGRAD_GEMM_WGS")) n = lab_env("SPCL_WGRAD_GEMM_WGS", n);
  }
  return n;
}

struct WbPlan {
  WbArgs args;
  int th, nwg, total_e;
};

// tile height with the fewest padded pixels over the batch; pixel splits so that every CU gets one workgroup and the
// workgroups carry about the same number of tiles
static int wb_plan(const spcl_wgrad_item* items, int n, WbPlan& pl) {
  if (n < 1 || n > WB_MAX) return SPCL_EINVAL;
  long waste[2] = {0, 0};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 2; ++k) {
      const int th = k ? 16 : 14;
      waste[k] += (long)items[i].N * cdiv(items[i].H, th) * th * cdiv(items[i].W, 16) * 16 * (items[i].Cin / 64) *
                  (items[i].Cout / 64);
    }
  pl.th = waste[1] < waste[0] ? 16 : 14;
  if (lab_flag("SPCL_WGRAD_GEMM_TH")) pl.th = lab_env("SPCL_WGRAD_GEMM_TH", 14) == 16 ? 16 : 14;
  long units = 0;
  for (int i = 0; i < n; ++i) {
    WbItem& w = pl.args.it[i];
    const spcl_wgrad_item& s = items[i];
    w.x = (const bf16_t*)s.x; w.x2 = (const bf16_t*)s.x2; w.x_up2 = s.x_up2; w.dy = (const bf16_t*)s.dy; w.in_scale = s.in_scale; w.in_shift = s.in_shift; w.dw = s.dw_oihw;
    w.N = s.N; w.H = s.H; w.W = s.W; w.Cin = s.Cin; w.Cout = s.Cout; w.CinS = s.CinS; w.CoutS = s.CoutS;
    w.in_mode = s.in_mode;
    w.tilesX = cdiv(s.W, 16);
    w.tpi = w.tilesX * cdiv(s.H, pl.th);
    w.ntiles = s.N * w.tpi;
    w.nblk_co = s.Cout / 64;
    w.nblk = (s.Cin / 64) * w.nblk_co;
    units += (long)w.nblk * w.ntiles;
  }
  const int P = wb_cu_count();
  int used = 0;
  for (int i = 0; i < n; ++i) {
    WbItem& w = pl.args.it[i];
    long ns = (long)w.ntiles * P / units;  // floor of the proportional share
    if (ns < 1) ns = 1;
    if (ns > w.ntiles) ns = w.ntiles;
    w.nsplit = (int)ns;
    used += w.nblk * w.nsplit;
  }
  for (;;) {  // hand the remaining CUs to the layers whose workgroups carry the most tiles
    int best = -1;
    double load = 0.0;
    for (int i = 0; i < n; ++i) {
      const WbItem& w = pl.args.it[i];
      if (w.nsplit < w.ntiles && used + w.nblk <= P && (double)w.ntiles / w.nsplit > load) {
        load = (double)w.ntiles / w.nsplit;
        best = i;
      }
    }
    if (best < 0) break;
    pl.args.it[best].nsplit += 1;
    used += pl.args.it[best].nblk;
  }
  int wg = 0, e = 0;
  for (int i = 0; i < n; ++i) {
    WbItem& w = pl.args.it[i];
    w.wg0 = wg;
    w.e0 = e;
    wg += w.nblk * w.nsplit;
    e += w.nblk * 16;  // reduce-kernel workgroups: (block, ci-tile, co-tile)
  }
  pl.nwg = wg;
  pl.total_e = e;
  pl.args.n = n;
  return SPCL_OK;
}

static bool wb_item_ok(const spcl_wgrad_item& s) {
  return s.x && s.dy && s.dw_oihw && s.N > 0 && s.H > 0 && s.W > 0 && s.Cin > 0 && s.Cout > 0 && s.Cin % 64 == 0 &&
         s.Cout % 64 == 0 && s.CinS >= s.Cin && s.CoutS >= s.Cout && s.CinS % 8 == 0 && s.CoutS % 8 == 0 &&
         (s.in_mode == 0 || (s.in_mode == 1 && s.in_scale && s.in_shift)) &&
         (s.x2 == nullptr || (s.in_mode == 0 && s.Cin % 128 == 0 && s.CinS == s.Cin)) &&
         (s.x_up2 == 0 || (s.in_mode == 0 && s.x2 == nullptr && s.H % 2 == 0 && s.W % 2 == 0)) &&
         (long)s.N * s.H * s.W * (s.CinS > s.CoutS ? s.CinS : s.CoutS) < (1L << 31);
}

template <int TH, int RING> static void wb_launch(const WbPlan& pl, hipStream_t st) {
  constexpr int XGRP = ((TH + 2) * 18 + 7) / 8, DGRP = TH * 2;
  constexpr int NDB = (2 * XGRP + 3 * DGRP) * 1024 <= 163840 ? 3 : 2;
  const size_t ldsb = (size_t)(2 * XGRP + NDB * DGRP) * 1024;
  static bool attr = false;
  if (!attr) {
    spcl::func_lds_limit((const void*)wgrad_gemm_kernel<TH, RING>, (int)ldsb, "wgrad_gemm_kernel<TH, RING>");
    attr = true;
  }
  SPCL_LAUNCH((wgrad_gemm_kernel<TH, RING>), dim3(pl.nwg), dim3(768), ldsb, st, pl.args);
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_conv_wgrad_batched_supported(int dtype, int Cin, int CinS, int Cout, int CoutS, int in_mode) {
  return dtype == SPCL_BF16 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0 && CinS >= Cin && CoutS >= Cout &&
         CinS % 8 == 0 && CoutS % 8 == 0 && (in_mode == 0 || in_mode == 1);
}

extern "C" size_t spcl_conv_wgrad_batched_workspace_bytes(const spcl_wgrad_item* items, int n) {
  if (!items || n < 1 || n > WB_MAX) return 0;
  for (int i = 0; i < n; ++i)
    if (!wb_item_ok(items[i])) return 0;
  WbPlan pl;
  if (wb_plan(items, n, pl) != SPCL_OK) return 0;
  return (size_t)pl.nwg * WB_SLAB * sizeof(float);
}

static bool wb_tail_ok(const spcl_wgrad_tail& t) {
  if (!t.partial || !t.dw || t.nsplit < 1) return false;
  if (t.kind == 0)
    return t.nblk_ci > 0 && t.nblk_co > 0 && t.CIB > 0 && t.COB > 0 && t.Cin > 0 && t.Cout > 0 &&
           t.Cin <= t.nblk_ci * t.CIB && t.Cout <= t.nblk_co * t.COB && (9 * t.CIB * t.COB) % 64 == 0 &&
           (uintptr_t)t.partial % 16 == 0;
  return t.kind == 1 && t.COB >= 16 && t.COB <= 256 && t.Cout > 0 && t.Cout <= t.COB;
}

extern "C" int spcl_conv3x3_wgrad_batched(const spcl_wgrad_item* items, int n, int accumulate, float* partial,
                                          void* stream) {
  SPCL_CHECK_ARG(n >= 1, "conv3x3_wgrad_batched: %d items (1..%d)", n, WB_MAX);
  return spcl_conv3x3_wgrad_batched_tails(items, n, nullptr, 0, accumulate, partial, stream);
}

extern "C" int spcl_conv3x3_wgrad_batched_tails(const spcl_wgrad_item* items, int n, const spcl_wgrad_tail* tails,
                                                int ntails, int accumulate, float* partial, void* stream) {
  SPCL_CHECK_ARG(n >= 0 && n <= WB_MAX, "conv3x3_wgrad_batched: %d items (0..%d)", n, WB_MAX);
  SPCL_CHECK_ARG(ntails >= 0 && ntails <= SPCL_WGRAD_TAILS_MAX, "conv3x3_wgrad_batched: %d tails (0..%d)", ntails,
                 SPCL_WGRAD_TAILS_MAX);
  SPCL_CHECK_ARG(n + ntails > 0, "conv3x3_wgrad_batched: nothing to do");
  SPCL_CHECK_ARG((n == 0 || (items && partial)) && (ntails == 0 || tails), "conv3x3_wgrad_batched: null pointer");
  for (int i = 0; i < n; ++i)
    SPCL_CHECK_ARG(wb_item_ok(items[i]), "conv3x3_wgrad_batched: item %d: bf16 NHWC, channel counts multiples of 64, "
                                         "in_mode 0/1 (with scale/shift)", i);
  WbTails tl;
  memset(&tl, 0, sizeof(tl));
  int tail_units = 0;
  for (int i = 0; i < ntails; ++i) {
    SPCL_CHECK_ARG(wb_tail_ok(tails[i]), "conv3x3_wgrad_batched: tail %d is not a captured weight-gradient tail", i);
    WbTail& t = tl.t[i];
    t.partial = tails[i].partial; t.dw = tails[i].dw; t.kind = tails[i].kind; t.nsplit = tails[i].nsplit;
    t.nblk_ci = tails[i].nblk_ci; t.nblk_co = tails[i].nblk_co; t.CIB = tails[i].CIB; t.COB = tails[i].COB;
    t.Cin = tails[i].Cin; t.Cout = tails[i].Cout;
    // few splits (the many-block layers: f32 storage sends every layer here): a unit of 64 outputs kept one quarter-wave of
    // its nine waves busy and the launch was 18 000 workgroups of a few loads each (87 us for 70 MB); kind 2: 2 304 outputs
    // per unit, one 16-byte column per thread, no LDS
    if (t.kind == 0 && t.nsplit <= 16 && (9 * t.CIB * t.COB) % 2304 == 0) t.kind = 2;
    t.u0 = tail_units;
    tail_units += t.kind == 0 ? t.nblk_ci * t.nblk_co * (9 * t.CIB * t.COB / 64)
                              : (t.kind == 2 ? t.nblk_ci * t.nblk_co * (9 * t.CIB * t.COB / 2304) : 9);
  }
  tl.n = ntails;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {  // tails only
    WbArgs a;
    memset(&a, 0, sizeof(a));
    a.accumulate = accumulate ? 1 : 0;
    tl.first = tail_units;
    SPCL_LAUNCH(wgrad_gemm_reduce_kernel, dim3(tail_units), dim3(576), 0, st, a, tl);
    SPCL_LAUNCH_CHECK("conv3x3_wgrad_batched_tails");
    return SPCL_OK;
  }
  WbPlan pl;
  wb_plan(items, n, pl);
  pl.args.partial = partial;
  pl.args.accumulate = accumulate ? 1 : 0;
  static const int env_dbg = lab_env("SPCL_WGRAD_GEMM_DBG", 0);
  pl.args.dbg = env_dbg;
  static const int env_stamps = lab_env("SPCL_WGRAD_GEMM_STAMPS", 0);
  static unsigned long long* stamp_buf = nullptr;
  pl.args.stamps = nullptr;
  if (env_stamps) {  // debug only (synchronises)
    if (!stamp_buf) (void)hipMalloc(&stamp_buf, 4096 * 8 * sizeof(unsigned long long));
    if (pl.nwg <= 4096) {
      (void)hipMemsetAsync(stamp_buf, 0, 4096 * 8 * sizeof(unsigned long long), st);
      pl.args.stamps = stamp_buf;
    }
  }
  double bytes = 0.0, flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const double px = (double)items[i].N * items[i].H * items[i].W;
    bytes += px * (items[i].Cin + items[i].Cout) * 2.0 + 9.0 * items[i].Cin * items[i].Cout * 4.0;
    flops += 2.0 * px * 9.0 * items[i].Cin * items[i].Cout;
  }
  prof_cost(bytes, flops);
  static const int env_ring = lab_env("SPCL_WGRAD_GEMM_RING", 2);
  if (pl.th == 16) wb_launch<16, 2>(pl, st);
  else if (env_ring == 1) wb_launch<14, 1>(pl, st);
  else if (env_ring == 3) wb_launch<14, 3>(pl, st);
  else if (env_ring == 4) wb_launch<14, 4>(pl, st);
  else wb_launch<14, 2>(pl, st);
  tl.first = tail_units;
  SPCL_LAUNCH(wgrad_gemm_reduce_kernel, dim3(pl.total_e + tail_units), dim3(576), 0, st, pl.args, tl);
  if (pl.args.stamps) {
    static unsigned long long h[4096 * 8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, pl.args.stamps, (size_t)pl.nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[8] = {0};
    for (int i = 0; i < pl.nwg; ++i)
      for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k];
    const double tiles = sum[3] > 0 ? sum[3] : 1;
    fprintf(stderr, "[wgrad_gemm stamps] %d wgs, %.1f tiles/wg | s_memtime ticks per tile: producer write x (incl. its wait) %.0f, "
            "request x + dy wait + DMA issue %.0f, barrier %.0f | consumer barrier %.0f, compute %.0f\n", pl.nwg, tiles / pl.nwg, sum[0] / tiles,
            sum[1] / tiles, sum[2] / tiles, sum[4] / tiles, sum[5] / tiles);
  }
  SPCL_LAUNCH_CHECK("conv3x3_wgrad_batched");
  return SPCL_OK;
}
