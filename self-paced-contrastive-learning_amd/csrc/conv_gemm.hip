// 3x3 convolution of the >= 64-channel layers (bf16, NHWC) as a workgroup-level implicit GEMM: Conv3.b .. Conv5.b of the
// UNet encoder (semi_seg/arch/unet.py:67-82,125-131) and the decoder's wide layers, forward and dgrad.  These layers are
// MFMA-bound (7.4 / 14.8 GFLOP each at N = 64) and small in space (56^2 .. 14^2): the per-wave kernels of conv_fast.hip
// fetch every weight fragment from L2 in every wave (C5.b: 1.2 MB per workgroup, 150 MB per launch against 6 MB of
// activations) and reach 0.4 - 0.6 PFLOP/s.  Here:
//   * a workgroup owns a BAND of an image (R rows x TWc columns, at most 224 pixels = 7 groups of 32 consecutive pixels
//     of the band in row-major order) x a block of 64 or 128 output channels;
//   * the band's halo ((R + 2) x (TWc + 2) pixels x 64 input channels) is staged once per 64-channel slab -- the next
//     slab's loads are in flight while the current one is multiplied -- and all 9 taps are address offsets into it;
//   * the weights of one (slab, tap) for the workgroup's channel block (8 or 16 KB) arrive by LDS-DMA into a double
//     buffer, pre-packed in fragment order (conv.hip pack_value: one 1 KiB piece = one A fragment), and are SHARED by
//     the workgroup's waves: weight traffic per workgroup drops by the number of pixel groups it multiplies them with;
//   * v_mfma_f32_32x32x16_bf16: A = weights (row = output channel), B = activations (column = pixel), so a lane ends
//     with 4 consecutive output channels of its pixel per accumulator quad (8-byte NHWC stores), a wave with 32 output
//     channels x up to 4 pixel groups (64 accumulator registers).
// Same contract as the other convolution kernels: optional BatchNorm + ReLU of the producer fused into the staging
// (in_mode 1), Chan partials (count, mean, M2) of the output per statistics row for the following train-mode
// BatchNorm, or (dgrad) the partial sums of the previous BatchNorm's backward (rows2).  A statistics row here is one
// (band, pixel half): spcl_conv_stat_rows tells the caller how many there are.
#include <stdlib.h>
#include <vector>
#include "conv_common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8m;

constexpr int GEMM_PS = 128;  // bytes per halo pixel in LDS: 64 channels, the eight 16-byte chunks XOR-swizzled by the
                              // pixel's index in band order (see the kernel): conflict-free ds_read_b128 at every tap

struct GemmGeom {
  int R, TWc, HWc, bandsY, tilesX, ng, gh, tps, hdb;
  int rows() const { return bandsY * tilesX * 4; }  // statistics rows per image: one per (band, pixel part)
  size_t halo_bytes() const { return (size_t)(R + 2) * HWc * GEMM_PS; }
  size_t lds_bytes() const { return (size_t)3 * tps * 8192 + (hdb ? 2 : 1) * halo_bytes(); }
};

static GemmGeom gemm_geom(int N, int H, int W, int CinK, int CoutS) {
  GemmGeom g;
  g.TWc = W;
  if (W > 64) {  // column tiles: the width with the least padding among a few
    int best = 56;
    double bu = 0.0;
    for (int tw : {64, 56, 48, 40, 32}) {
      const double u = (double)W / (cdiv(W, tw) * tw);
      if (u > bu + 1e-9) { bu = u; best = tw; }
    }
    g.TWc = best;
  }
  g.tilesX = cdiv(W, g.TWc);
  int bestR = 1;
  double bu = 0.0;
  for (int r = 1; r * g.TWc <= 256 && r <= H; ++r) {  // at most 8 groups of 32 pixels: two per pixel part
    double u = ((double)H / (cdiv(H, r) * r)) * ((double)(r * g.TWc) / (cdiv(r * g.TWc, 64) * 64));
    if (r * g.TWc <= 192 && H * g.TWc > 192) u *= 0.7;  // small bands: a workgroup's fixed costs over less work
    if (u >= bu - 1e-9) { bu = u; bestR = r; }  // ties: the taller band (fewer halo rows per pixel)
  }
  g.R = bestR;
  g.HWc = g.TWc + 2;
  g.bandsY = cdiv(H, g.R);
  g.ng = cdiv(g.R * g.TWc, 32);
  g.gh = cdiv(g.ng, 4);
  g.tps = 3;
  g.hdb = (CinK > 64 && 3 * 3 * 8192 + 2 * g.halo_bytes() <= 160 * 1024) ? 1 : 0;  // two halo images when they fit
  return g;
}

struct GemmArgs {
  const unsigned char* x;
  unsigned char* y;
  const unsigned char* wp;
  float* stats;
  const float* in_scale;
  const float* in_shift;
  const unsigned char* y2;
  const float* scale2;
  const float* shift2;
  const float* mean2;
  float* rows2;
  int N, H, W, CinK, CoutS;
  int R, TWc, HWc, bandsY, tilesX, ng, gh, hdb;
  unsigned magic_hw, magic_tw;  // ceil(2^22 / HWc), ceil(2^22 / TWc): q / d == (q * magic) >> 22 for q < 4096
  unsigned long long* stamps;  // debug (SPCL_GEMM_STAMPS=1): s_memtime ticks of thread 0 per phase, else null
};

__device__ __forceinline__ void gemm_dma16(const void* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_dst)
               : "memory");
}

// sum over the 32 lanes that share lane >> 5 (every lane of the half gets the total)
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  return v;
}

// A workgroup = 8 waves = 2 output tiles of 32 channels x 4 pixel parts: wave -> (tile wave % 2, part wave / 2).
// GH: pixel groups per wave (1 or 2; a wave with fewer real groups multiplies a dummy one).
// TPS: taps per weight stage (3: one barrier per three taps, weights requested two stages = ~2 us ahead into a ring of
//      three 24 KB images -- the layers with at most two workgroups per CU; 1: 8 KB images, two workgroups share a CU).
// MODE 0 plain input, 1 relu(scale x + shift) on the input, 2 plain input + BatchNorm-backward sums of the output.
template <int GH, int MODE>
__global__ __launch_bounds__(512) void conv3x3_gemm_kernel(GemmArgs a) {
  constexpr int NCT = 2, NTHR = 512, TPS = 3;
  constexpr int ITER = (400 * 8 + NTHR - 1) / NTHR;  // halo chunks per thread (halo <= 400 pixels, see gemm_geom)
  constexpr int WBUF = TPS * NCT * 4096;             // one stage (TPS taps of one slab) of weights
  constexpr int PPW = TPS;                           // 1 KiB weight pieces per wave and stage (TPS x 8 pieces / 8 waves)
  constexpr int NS = TPS * 4;                        // 16-channel k-steps per stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char* const halo0 = lds + 3 * WBUF;
  const unsigned lds_base = (unsigned)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)lds;

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n32 = lane & 31, kh = lane >> 5;
  const int ct = wave & 1, ph = wave >> 1;
  const bool stamp = a.stamps != nullptr && t == 0;
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t0 = stamp ? __builtin_amdgcn_s_memtime() : 0, t1;
#define GEMM_STAMP(k)                                     \
  if (stamp) {                                            \
    t1 = __builtin_amdgcn_s_memtime();                    \
    tk[k] += t1 - t0;                                     \
    t0 = t1;                                              \
  }
  const int tx = blockIdx.x % a.tilesX, band = blockIdx.x / a.tilesX, n = blockIdx.z;
  const int cot0 = blockIdx.y * NCT;                 // first 32-channel output tile of this workgroup
  const int y0 = band * a.R, x0 = tx * a.TWc;
  const int nslab = a.CinK >> 6;
  const int ncot = a.CoutS >> 5;
  const int halo_bytes = (a.R + 2) * a.HWc * GEMM_PS;
  const int nhalo8 = (a.R + 2) * a.HWc * 8;

  // ---- staging map: chunk c = t + k NTHR -> halo pixel c / 8, 16-byte channel chunk c % 8 (= t % 8 for every k)
  // (divisions by the halo / band width as multiplications: exact for the < 4096 pixel indices here)
  const int ch = t & 7;
  int goff[ITER];   // byte offset in the image, or beyond it (the buffer load returns zeros: the padding)
  int loff[ITER];   // byte offset in the halo image: pixel slot * 128 + swizzled chunk position
  unsigned smask = 0;  // bit k: inside the image; bit 16 + k: inside the halo
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (size_t)n * a.H * a.W * a.CinK * 2), 0, a.H * a.W * a.CinK * 2, 0x00020000);
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int c = t + k * NTHR;
    const int q = c >> 3;
    const int hy = (int)(((unsigned)q * a.magic_hw) >> 22), hx = q - hy * a.HWc;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool inh = c < nhalo8;
    const bool inb = inh && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    goff[k] = inb ? ((gy * a.W + gx) * a.CinK + ch * 8) * 2 : 0x7ffffff0;
    // swizzle key of a halo pixel = its index in band order (row pitch TWc, NOT the halo's TWc + 2), mod 16: the 16
    // lanes of a ds_read_b128 group hold 16 pixels that are consecutive mod 16 in band order at EVERY tap, the key's
    // bit 0 is the slot's parity (the halo pitch has the parity of TWc) and bits 1..3 pick the chunk position
    loff[k] = q * GEMM_PS + ((ch ^ (((hy * a.TWc + hx) & 15) >> 1)) << 4);
    smask |= (inb ? 1u : 0u) << k | (inh ? 1u : 0u) << (16 + k);
  }
  u32x4 v[ITER];
  auto stage_load = [&](int slab) {  // exactly ITER vector memory operations per wave (the waits below count them)
#pragma unroll
    for (int k = 0; k < ITER; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[k], slab * 128, 0);
  };
  auto stage_store = [&](int slab) {  // -> halo image slab & 1 (when there are two), else the only one
    unsigned char* const hb = halo0 + (a.hdb ? (slab & 1) * halo_bytes : 0);
    float ssc[8], ssh[8];
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        *(f32x4*)&ssc[e] = *(const f32x4*)(a.in_scale + slab * 64 + ch * 8 + e);
        *(f32x4*)&ssh[e] = *(const f32x4*)(a.in_shift + slab * 64 + ch * 8 + e);
      }
    }
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      u32x4 tv = v[k];
      // zero padding applies to the ACTIVATION: pixels outside the image stay 0, not relu(shift)
      if (MODE == 1 && (smask >> k & 1)) tv = bnrelu_regs<bf16_t>(tv, ssc, ssh);
      if (smask >> (16 + k) & 1) *(u32x4*)(hb + loff[k]) = tv;
    }
  };
  // ---- weights of stage sg (taps (sg % SPS) TPS .. + TPS - 1 of slab sg / SPS): TPS x 8 fragments of 1 KiB, PPW per
  // wave, into image sg % 3 of the ring
  const unsigned char* wsrc = a.wp + (size_t)cot0 * 4096 + lane * 16;
  auto issue_w = [&](int sg) {
    const unsigned d = lds_base + (unsigned)((sg % 3) * WBUF);
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int p = wave * PPW + u, ti = p >> 3, rest = p & 7;
      gemm_dma16(wsrc + ((size_t)(sg * TPS + ti) * ncot * 4 + rest) * 1024, __builtin_amdgcn_readfirstlane(d + p * 1024));
    }
  };

  // ---- this wave's pixel groups: slot p = 32 (g0 + gi) + n32 of the band, row-major
  const int g0 = ph * a.gh;
  int bslot[GH];          // byte offset of the pixel's halo slot at tap (0, 0)
  int bkey[GH];           // its index in band order (the swizzle key before the tap's shift)
  int pixoff[GH];         // pixel index inside the image, -1: not a pixel of the image
#pragma unroll
  for (int gi = 0; gi < GH; ++gi) {
    const int p = 32 * (g0 + gi) + n32;
    const int py = (int)(((unsigned)p * a.magic_tw) >> 22), px = p - py * a.TWc;
    const bool real = gi < a.gh && g0 + gi < a.ng && p < a.R * a.TWc && y0 + py < a.H && x0 + px < a.W;
    bslot[gi] = real ? (py * a.HWc + px) * GEMM_PS : 0;
    bkey[gi] = real ? p : 0;
    pixoff[gi] = real ? ((y0 + py) * a.W + x0 + px) : -1;
  }
  f32x16 acc[GH];
#pragma unroll
  for (int gi = 0; gi < GH; ++gi)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[gi][e] = 0.f;

  GEMM_STAMP(0)  // index arithmetic
  constexpr int SPS = 9 / TPS;  // stages per slab
  const int nsg = nslab * SPS;
  stage_load(0);
  issue_w(0);
  issue_w(1);
  stage_store(0);
  GEMM_STAMP(1)  // first halo: loads + LDS writes
  int sg = 0;
#pragma unroll 1
  for (int slab = 0; slab < nslab; ++slab) {
    const unsigned hoff = 3 * WBUF + (a.hdb ? (slab & 1) * halo_bytes : 0);  // this slab's halo image in `lds`
#pragma unroll 1
    for (int s = 0; s < SPS; ++s, ++sg) {
      // This stage's weights were requested two stages ago.  Younger than them, and allowed to be on their way still:
      // the next stage's pieces and, right after a slab's first stage, the next slab's halo loads (issued after them).
      if (sg + 1 >= nsg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (s == 1 && slab + 1 < nslab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + ITER) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      GEMM_STAMP(2)  // wait for this wave's pieces
      __syncthreads();  // ... everyone's have landed; the halo is written; ring image (sg + 2) % 3 is free
      GEMM_STAMP(4)  // barrier
      // the other halo image (free since the previous slab's last barrier) gets the next slab BEFORE new transfers are
      // requested: the compiler's wait for the halo registers then covers last stage's requests only
      if (a.hdb && s == SPS - 1 && slab + 1 < nslab) stage_store(slab + 1);
      if (sg + 2 < nsg) issue_w(sg + 2);
      if (s == 0 && slab + 1 < nslab) stage_load(slab + 1);  // in flight during this slab's taps
      const u32x4* wl = (const u32x4*)(lds + (sg % 3) * WBUF) + ct * 256 + lane;
      // Per tap and pixel group ONE address: slot + tap shift + the swizzled position of chunk kh; the slot is 128-byte
      // aligned, so the chunk 2 ks + kh of the other k-steps is that address XOR (ks << 5) -- one vector instruction
      // per fragment read (PMC on the first version: 7.4 vector instructions per MFMA, the vector port as busy as
      // the matrix pipe).
      unsigned badr[TPS][GH];
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int tap = s * TPS + tt;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int toff = (ky * a.HWc + kx) * GEMM_PS, dk = ky * a.TWc + kx;
#pragma unroll
        for (int gi = 0; gi < GH; ++gi)
          badr[tt][gi] = hoff + bslot[gi] + toff + ((kh ^ (((bkey[gi] + dk) & 15) >> 1)) << 4);
      }
      // k-steps of 16 channels, fragments two steps ahead of their MFMAs in three register sets that rotate (no
      // copies); pinned with sched_barrier: the scheduler would sink the reads next to their use
      u32x4 af[3], bf[3][GH];
      auto fetch = [&](int i, int set) {
        const int tt = i >> 2, ks = i & 3;
        af[set] = wl[tt * NCT * 256 + ks * 64];
#pragma unroll
        for (int gi = 0; gi < GH; ++gi) bf[set][gi] = *(const u32x4*)(lds + (badr[tt][gi] ^ (unsigned)(ks << 5)));
      };
      fetch(0, 0);
      fetch(1, 1);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i + 2 < NS) fetch(i + 2, (i + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < GH; ++gi)
          acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8m, af[i % 3]),
                                                            __builtin_bit_cast(bf16x8m, bf[i % 3][gi]), acc[gi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      GEMM_STAMP(3)  // issue + MFMAs of a stage
    }
    if (!a.hdb && slab + 1 < nslab) {
      __syncthreads();  // every wave is done reading this slab's halo
      stage_store(slab + 1);
      GEMM_STAMP(4)  // next halo -> LDS
    }
  }

  // ---- epilogue: acc[gi][4q + r] = output channel 32 (cot0 + ct) + 8q + 4kh + r of pixel slot 32 (g0 + gi) + n32
  const int cb = (cot0 + ct) * 32 + 4 * kh;  // + 8q
  const int rowb = a.CoutS * 2;
  unsigned char* yb = a.y + (size_t)n * a.H * a.W * rowb + cb * 2;
  f32x4 s1[4], s2[4];
  f32x4 sc2[4], sh2[4], mu2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    s1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s2[q] = s1[q];
    if (MODE == 2) {
      sc2[q] = *(const f32x4*)(a.scale2 + cb + 8 * q);
      sh2[q] = *(const f32x4*)(a.shift2 + cb + 8 * q);
      mu2[q] = *(const f32x4*)(a.mean2 + cb + 8 * q);
    }
  }
  float cnt = 0.f;
#pragma unroll
  for (int gi = 0; gi < GH; ++gi) {
    const bool real = pixoff[gi] >= 0;
    cnt += real ? 1.f : 0.f;
    if (real) {
      unsigned char* dst = yb + (size_t)pixoff[gi] * rowb;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = {acc[gi][4 * q], acc[gi][4 * q + 1], acc[gi][4 * q + 2], acc[gi][4 * q + 3]};
        store4_fast<bf16_t>(dst + 16 * q, o);
        if (MODE == 2) {
          // dz = g [relu(bn(y2)) > 0] with g as STORED (bf16): sum dz and sum dz (y2 - mean) per channel
          const uint2 yr = *(const uint2*)(a.y2 + (dst - a.y) + 16 * q);
          const f32x2 glo = {o[0], o[1]}, ghi = {o[2], o[3]};
          const uint32_t gb0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(glo, bf16x2v));
          const uint32_t gb1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ghi, bf16x2v));
          const float yv[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u),
                               __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
          const float gv[4] = {__uint_as_float(gb0 << 16), __uint_as_float(gb0 & 0xffff0000u),
                               __uint_as_float(gb1 << 16), __uint_as_float(gb1 & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dz = fmaf(sc2[q][r], yv[r], sh2[q][r]) > 0.f ? gv[r] : 0.f;
            s1[q][r] += dz;
            s2[q][r] = fmaf(dz, yv[r] - mu2[q][r], s2[q][r]);
          }
        } else {
          s1[q] += o;
          s2[q] += o * o;
        }
      }
    }
  }
  GEMM_STAMP(5)  // stores
  if (stamp) {
    unsigned long long* o = a.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 6;
    for (int k = 0; k < 6; ++k) o[k] = tk[k];
  }
#undef GEMM_STAMP
  const bool want = MODE == 2 ? a.rows2 != nullptr : a.stats != nullptr;
  if (!want) return;
  // one statistics row per (band, pixel half); every lane of a 32-lane half ends with the half's totals
  const int row = ((n * a.bandsY + band) * a.tilesX + tx) * 4 + ph;
  cnt = half32_sum(cnt);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t1, t2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      t1[r] = half32_sum(s1[q][r]);
      t2[r] = half32_sum(s2[q][r]);
    }
    if (n32 == 0) {
      if (MODE == 2) {
        float* o = a.rows2 + (size_t)row * 2 * a.CoutS + cb + 8 * q;
        *(f32x4*)o = t1;
        *(f32x4*)(o + a.CoutS) = t2;
      } else {
        float* o = a.stats + (size_t)row * 3 * a.CoutS + cb + 8 * q;
        f32x4 mean, m2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          mean[r] = t1[r] * inv;
          float prod = t1[r] * mean[r];
          asm volatile("" : "+v"(prod));  // M2 = sum x^2 - n mean^2, the product rounded on its own (conv_common.hpp)
          m2[r] = fmaxf(t2[r] - prod, 0.f);
        }
        *(f32x4*)o = (f32x4){cnt, cnt, cnt, cnt};
        *(f32x4*)(o + a.CoutS) = mean;
        *(f32x4*)(o + 2 * a.CoutS) = m2;
      }
    }
  }
}

bool conv_gemm_fits(int H, int W, int CinK, int CoutS) {
  const GemmGeom gg = gemm_geom(1, H, W, CinK, CoutS);
  return (gg.R + 2) * gg.HWc <= 400 && gg.gh <= 2 && gg.lds_bytes() <= 160 * 1024;
}

int conv_gemm_stat_rows(int N, int H, int W, int CinK, int CoutS) { return N * gemm_geom(N, H, W, CinK, CoutS).rows(); }

template <int GH>
static void launch_gemm_mode(const GemmArgs& g, int mode, dim3 grid, size_t lds, hipStream_t st) {
#define SPCL_GEMM_LAUNCH(MODE_)                                                                              \
  do {                                                                                                       \
    static bool attr = false;                                                                                \
    if (!attr) {                                                                                             \
      spcl::func_lds_limit((const void*)conv3x3_gemm_kernel<GH, MODE_>, (int)(160 * 1024), "conv3x3_gemm_kernel");                     \
      attr = true;                                                                                           \
    }                                                                                                        \
    SPCL_LAUNCH((conv3x3_gemm_kernel<GH, MODE_>), grid, dim3(512), lds, st, g);                              \
  } while (0)
  if (mode == 1) SPCL_GEMM_LAUNCH(1);
  else if (mode == 2) SPCL_GEMM_LAUNCH(2);
  else SPCL_GEMM_LAUNCH(0);
#undef SPCL_GEMM_LAUNCH
}

// returns false when the configuration is outside the kernel's range (the caller reports the error: the packed
// weights of a gemm-eligible layer are in this kernel's layout, so there is no other kernel to fall back to)
bool launch_conv_gemm(const ConvArgs& c, hipStream_t st) {
  if (!conv_gemm_channels(c.CinK, c.CoutS) || c.CinS != c.CinK || c.in_mode > 1) return false;
  if (c.rows2 != nullptr && c.in_mode != 0) return false;
  const GemmGeom gg = gemm_geom(c.N, c.H, c.W, c.CinK, c.CoutS);
  if ((gg.R + 2) * gg.HWc > 400 || gg.gh > 2 || gg.lds_bytes() > 160 * 1024) return false;
  GemmArgs g;
  g.x = (const unsigned char*)c.x; g.y = (unsigned char*)c.y; g.wp = (const unsigned char*)c.wp + (size_t)9 * c.CinK * c.CoutS * 2;  // second layout of the dual buffer
  g.stats = c.stats;
  g.in_scale = c.in_scale; g.in_shift = c.in_shift;
  g.y2 = (const unsigned char*)c.y2; g.scale2 = c.scale2; g.shift2 = c.shift2; g.mean2 = c.mean2; g.rows2 = c.rows2;
  g.N = c.N; g.H = c.H; g.W = c.W; g.CinK = c.CinK; g.CoutS = c.CoutS;
  g.R = gg.R; g.TWc = gg.TWc; g.HWc = gg.HWc; g.bandsY = gg.bandsY; g.tilesX = gg.tilesX; g.ng = gg.ng; g.gh = gg.gh;
  g.hdb = gg.hdb;
  g.magic_hw = ((1u << 22) + gg.HWc - 1) / gg.HWc;
  g.magic_tw = ((1u << 22) + gg.TWc - 1) / gg.TWc;
  const int mode = c.rows2 != nullptr ? 2 : c.in_mode;
  dim3 grid(gg.tilesX * gg.bandsY, c.CoutS / 64, c.N);
  static const bool env_stamps = lab_flag("SPCL_GEMM_STAMPS");
  const size_t nwg = (size_t)grid.x * grid.y * grid.z;
  g.stamps = nullptr;
  if (env_stamps) {  // debug only (synchronises)
    (void)hipMalloc(&g.stamps, nwg * 6 * sizeof(unsigned long long));
    (void)hipMemset(g.stamps, 0, nwg * 6 * sizeof(unsigned long long));
  }
  const size_t lds = gg.lds_bytes();
  if (gg.gh <= 1) launch_gemm_mode<1>(g, mode, grid, lds, st);
  else launch_gemm_mode<2>(g, mode, grid, lds, st);
  if (g.stamps != nullptr) {
    std::vector<unsigned long long> h(nwg * 6);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), g.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(g.stamps);
    double s6[6] = {0, 0, 0, 0, 0, 0};
    for (size_t w = 0; w < nwg; ++w)
      for (int k = 0; k < 6; ++k) s6[k] += (double)h[w * 6 + k] / nwg;
    fprintf(stderr, "[conv gemm %dx%d %d->%d mode %d] R=%d TWc=%d ng=%d tps=%d hdb=%d wgs=%zu lds=%zu | ticks per wg: setup "
                    "%.0f, first halo %.0f, weight waits %.0f, taps %.0f, barriers (+ next halos) %.0f, stores %.0f\n", c.H, c.W,
            c.CinK, c.CoutS, mode, gg.R, gg.TWc, gg.ng, gg.tps, gg.hdb, nwg, lds, s6[0], s6[1], s6[2], s6[3], s6[4], s6[5]);
  }
  return true;
}

}  // namespace spcl
