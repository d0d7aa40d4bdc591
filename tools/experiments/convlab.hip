// Experiment (not part of the product): a compile-time-specialised one-wave-per-tile 3x3 conv (bf16, NHWC) used to
// find where the production kernel's time goes.  Features are switched by template flags so that each one's cost is
// measured without runtime branches:
//   F & 1  MFMA k-loop (otherwise the staged interior is copied through: the tilecopy pattern)
//   F & 2  BatchNorm statistics in the epilogue
//   F & 4  all weight fragments preloaded before staging (KC=16/32 only)
//   F & 8  statistics as [tile][3][CoutS] written with 16-byte stores (otherwise [3][CoutS][tile], 4-byte scatter)
//   F & 16 NT=2 only: MFMA row (j, 4g+r) <-> cout 8g+4j+r so that a lane stores 8 consecutive couts (16 bytes)
// MODE 0 raw input, 1 relu(scale*x+shift) fused into staging.
// hipcc --offload-arch=gfx950 -O3 -o convlab convlab.hip && ./convlab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;

struct Args {
  const uint16_t* x;
  uint16_t* y;
  const u32x4* wp;
  const float* sc;
  const float* sh;
  float* stats;
  int N, H, W, CoutS, tilesX, tilesY;
};

template <int KC> constexpr int pstride() { return KC * 2 == 32 ? 32 : KC * 2 + 32; }

__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

__device__ __forceinline__ u32x4 bnrelu(u32x4 raw, const float* s, const float* b) {
  u32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float lo = __uint_as_float(raw[e] << 16), hi = __uint_as_float(raw[e] & 0xffff0000u);
    lo = fmaxf(fmaf(s[2 * e], lo, b[2 * e]), 0.f);
    hi = fmaxf(fmaf(s[2 * e + 1], hi, b[2 * e + 1]), 0.f);
    const f32x2 v = {lo, hi};
    out[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
  }
  return out;
}

__device__ __forceinline__ void store4(unsigned char* p, f32x4 v) {
  const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
  uint2 o;
  o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
  o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
  *(uint2*)p = o;
}

__device__ __forceinline__ void store8(unsigned char* p, f32x4 v, f32x4 w) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]}, c = {w[0], w[1]}, d = {w[2], w[3]};
  u32x4 o;
  o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2v));
  o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2v));
  o[2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(c, bf16x2v));
  o[3] = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, bf16x2v));
  *(u32x4*)p = o;
}

template <int KC, int TH, int TW, int NT, int MODE, int F, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void convlab(Args a) {
  constexpr int CP = KC / 8, HW_ = TW + 2, NHALO = (TH + 2) * HW_, PS = pstride<KC>();
  constexpr int NCH = NHALO * CP, ITER = (NCH + 63) / 64, QS = 64 / CP;
  constexpr int NPIX = TH * TW, MT = (NPIX + 15) / 16;
  constexpr int NSTEPS = (9 * CP + 3) / 4;
  static_assert(HW_ == 16, "lab: TW = 14 only");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
  const int y0 = ty * TH, x0 = tx * TW;
  const int tile = (n * a.tilesY + ty) * a.tilesX + tx;
  const int ntn = a.CoutS >> 4;
  const bool interior = y0 > 0 && x0 > 0 && y0 + TH < a.H && x0 + TW < a.W;

  // weights (tile-invariant): optionally all up front so that their latency hides under the staging loads
  u32x4 wall[(F & 4) ? NSTEPS : 1][NT];
  if (F & 4) {
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j) wall[s][j] = a.wp[(size_t)(s * ntn + j) * 64 + lane];
  }

  // ---------------- staging
  const int ch = lane & (CP - 1), q0 = lane / CP;
  const int hy0 = q0 / HW_, hx0 = q0 % HW_;
  float ssc[8], ssh[8];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *(f32x4*)&ssc[e] = *(const f32x4*)(a.sc + ch * 8 + e);
      *(f32x4*)&ssh[e] = *(const f32x4*)(a.sh + ch * 8 + e);
    }
  }
  const unsigned char* xb =
      (const unsigned char*)a.x + ((((long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * KC) * 2;  // halo origin
  const unsigned voff = (unsigned)((hy0 * a.W + hx0) * KC * 2 + ch * 16);
  u32x4 v[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    int dky, dkx;  // compile-time pixel advance of iteration k
    if (QS >= HW_) { dky = k * (QS / HW_); dkx = 0; }
    else { dky = k / (HW_ / QS); dkx = (k % (HW_ / QS)) * QS; }
    const long soff = ((long)dky * a.W + dkx) * KC * 2;
    const bool in_range = (NCH % 64 == 0) || (k * 64 + lane < NCH);
    bool inb = in_range;
    if (!interior) {
      const int gy = y0 - 1 + hy0 + dky, gx = x0 - 1 + hx0 + dkx;
      inb = inb && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    v[k] = (u32x4){0u, 0u, 0u, 0u};
    if (inb) v[k] = *(const u32x4*)(xb + soff + voff);
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    int dky, dkx;
    if (QS >= HW_) { dky = k * (QS / HW_); dkx = 0; }
    else { dky = k / (HW_ / QS); dkx = (k % (HW_ / QS)) * QS; }
    const bool in_range = (NCH % 64 == 0) || (k * 64 + lane < NCH);
    u32x4 t = v[k];
    if (MODE == 1) {
      bool inb = true;
      if (!interior) {
        const int gy = y0 - 1 + hy0 + dky, gx = x0 - 1 + hx0 + dkx;
        inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      }
      if (inb) t = bnrelu(t, ssc, ssh);
    }
    if (in_range) *(u32x4*)(lds + ((hy0 + dky) * HW_ + hx0 + dkx) * PS + ch * 16) = t;
  }
  __syncthreads();

  unsigned char* yb = (unsigned char*)a.y + (((size_t)n * a.H + y0) * a.W + x0) * a.CoutS * 2;
  const int rowb = a.CoutS * 2;

  if (!(F & 1)) {
    // copy-through: interior chunks back to global (KC == CoutS)
#pragma unroll 2
    for (int idx = lane; idx < NPIX * CP; idx += 64) {
      const int p = idx / CP, c = idx % CP;
      const int py = p / TW, px = p % TW;
      const u32x4 t = *(const u32x4*)(lds + ((py + 1) * HW_ + px + 1) * PS + c * 16);
      *(u32x4*)(yb + (py * a.W + px) * rowb + c * 16) = t;
    }
    return;
  }

  // ---------------- k-loop
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = 16 * i + r16;
    if (p >= NPIX) p = 0;
    const int py = p / TW, px = p - py * TW;
    abase[i] = (py * HW_ + px) * PS + (CP >= 4 ? g * 16 : 0);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NSTEPS; ++s) {
    u32x4 wf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = (F & 4) ? wall[s][j] : a.wp[(size_t)(s * ntn + j) * 64 + lane];
    int off;
    if (CP >= 4) {
      const int fc0 = 4 * s, tap = fc0 / CP, c0 = fc0 % CP, ky = tap / 3, kx = tap % 3;
      off = (ky * HW_ + kx) * PS + c0 * 16;  // compile-time: folds into the ds_read offset field
    } else {
      int fc = 4 * s + g;
      if (fc >= 9 * CP) fc = 0;
      const int tap = fc / CP, c = fc % CP, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      off = (ky * HW_ + kx) * PS + c * 16;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const u32x4 xf = *(const u32x4*)(lds + abase[i] + off);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]),
                                                            __builtin_bit_cast(bf16x8, xf), acc[i][j], 0, 0, 0);
    }
  }

  // ---------------- epilogue
  constexpr int DPY = 16 / TW, DPX = 16 % TW;
  const bool full_tile = y0 + TH <= a.H && x0 + TW <= a.W;
  int py = r16 / TW, px = r16 - py * TW;
  int ob = (py * a.W + px) * rowb + 8 * g;
  const int dob = (DPY * a.W + DPX) * rowb, wrapo = (a.W - TW) * rowb;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ssq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    bool ok = (16 * i + 15 < NPIX) || (16 * i + r16 < NPIX);
    if (!full_tile) ok = ok && (y0 + py) < a.H && (x0 + px) < a.W;
    if (ok) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!((F & 16) && NT == 2)) store4(yb + ob + j * 32, acc[i][j]);
        else if (j == 0) store8(yb + ob + 8 * g, acc[i][0], acc[i][NT - 1]);
        if (F & 2) {
          ssum[j] += acc[i][j];
          ssq[j] += acc[i][j] * acc[i][j];
        }
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
  if (F & 2) {
    const int ntiles = a.N * a.tilesX * a.tilesY;
    const int vh = min(TH, a.H - y0), vw = min(TW, a.W - x0);
    const float cnt = (float)(vh * vw), inv = 1.f / cnt;
    const size_t cstride = (size_t)ntiles, kstride = (size_t)a.CoutS * ntiles;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ssum[j][r] = row16_sum(ssum[j][r]);
        ssq[j][r] = row16_sum(ssq[j][r]);
      }
      const int cb = ((F & 16) && NT == 2) ? 8 * g + 4 * j : 16 * j + 4 * g;
      if (F & 8) {
        if (r16 < 3) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float mean = ssum[j][r] * inv;
            const float m2 = fmaxf(ssq[j][r] - ssum[j][r] * mean, 0.f);
            o[r] = r16 == 0 ? cnt : (r16 == 1 ? mean : m2);
          }
          *(f32x4*)(a.stats + ((size_t)tile * 3 + r16) * a.CoutS + cb) = o;
        }
      } else if (r16 == 0) {
        float* dst = a.stats + (size_t)cb * cstride + tile;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mean = ssum[j][r] * inv;
          dst[r * cstride] = cnt;
          dst[r * cstride + kstride] = mean;
          dst[r * cstride + 2 * kstride] = fmaxf(ssq[j][r] - ssum[j][r] * mean, 0.f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// packed[step][ntile][lane][8]: production layout (csrc/conv.hip pack_value, kind 0, one slab)
static std::vector<uint16_t> pack(const std::vector<float>& w, int Cin, int Cout, bool perm) {
  const int CP = Cin / 8, nsteps = (9 * CP + 3) / 4, ntn = Cout / 16;
  std::vector<uint16_t> p((size_t)nsteps * ntn * 64 * 8, 0);
  for (int s = 0; s < nsteps; ++s)
    for (int nt = 0; nt < ntn; ++nt)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int r16 = lane & 15, g = lane >> 4, fc = 4 * s + g;
          float v = 0.f;
          if (fc < 9 * CP) {
            const int tap = fc / CP, ch = fc % CP, ci = ch * 8 + e;
            const int co = perm ? (nt / 2) * 32 + 8 * (r16 >> 2) + 4 * (nt & 1) + (r16 & 3) : nt * 16 + r16;
            v = w[((size_t)co * Cin + ci) * 9 + tap];
          }
          p[(((size_t)s * ntn + nt) * 64 + lane) * 8 + e] = f2bf(v);
        }
  return p;
}

struct Bufs {
  uint16_t *x[3], *y[3];
  u32x4 *wp, *wpp;
  float *sc, *sh, *stats;
};

template <int KC, int TH, int TW, int NT, int MODE, int F, int WPE = 4>
static double run(const char* name, Bufs& b, int N, int H, int CoutS, bool check, const std::vector<uint16_t>& hx,
                  const std::vector<float>& hw, const std::vector<float>& hsc, const std::vector<float>& hsh) {
  Args a;
  a.wp = ((F & 16) && NT == 2) ? b.wpp : b.wp; a.sc = b.sc; a.sh = b.sh; a.stats = b.stats;
  a.N = N; a.H = H; a.W = H; a.CoutS = CoutS; a.tilesX = H / TW; a.tilesY = H / TH;
  const size_t lds = (size_t)(TH + 2) * (TW + 2) * pstride<KC>();
  dim3 grid(a.tilesX, a.tilesY, N);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&](int i) {
    static const int rot = getenv("ROT") ? atoi(getenv("ROT")) : 1;
    a.x = b.x[rot ? i % 3 : 0];
    a.y = b.y[rot ? i % 3 : 0];
    hipLaunchKernelGGL((convlab<KC, TH, TW, NT, MODE, F, WPE>), grid, dim3(64), lds, 0, a);
  };
  for (int i = 0; i < 3; ++i) launch(i);
  hipEventRecord(e0);
  const int it = 30;
  for (int i = 0; i < it; ++i) launch(i);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  const double bytes = (double)N * H * H * (KC + CoutS) * 2;
  printf("%-26s KC=%2d Co=%2d H=%3d %dx%d NT=%d M=%d F=%2d W=%d : %7.1f us  %6.0f GB/s", name, KC, CoutS, H, TH, TW, NT, MODE, F,
         WPE, us, bytes / us / 1e3);
  if (check && (F & 1)) {
    // spot-check 200 random outputs of buffer 0 against a host conv
    std::vector<uint16_t> hy((size_t)N * H * H * CoutS);
    hipMemcpy(hy.data(), b.y[0], hy.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    srand(7);
    for (int t = 0; t < 200; ++t) {
      const int n = rand() % N, yy = (t < 20) ? (t % 2 ? 0 : H - 1) : rand() % H, xx = (t < 40 && t >= 20) ? 0 : rand() % H,
                co = rand() % CoutS;
      double ref = 0;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int gy = yy + ky - 1, gx = xx + kx - 1;
          if (gy < 0 || gy >= H || gx < 0 || gx >= H) continue;
          for (int ci = 0; ci < KC; ++ci) {
            float xv = bf2f(hx[(((size_t)n * H + gy) * H + gx) * KC + ci]);
            if (MODE == 1) xv = bf2f(f2bf(fmaxf(fmaf(hsc[ci], xv, hsh[ci]), 0.f)));
            ref += (double)xv * bf2f(f2bf(hw[((size_t)co * KC + ci) * 9 + ky * 3 + kx]));
          }
        }
      const double got = bf2f(hy[(((size_t)n * H + yy) * H + xx) * CoutS + co]);
      maxerr = fmax(maxerr, fabs(got - ref) / (1.0 + fabs(ref)));
    }
    printf("  maxrelerr %.2e%s", maxerr, maxerr < 1e-2 ? "" : "  <<< MISMATCH");
  }
  printf("\n");
  return us;
}

template <int KC, int CO, int TH, int NT>
static void layer(int N, int H) {
  const size_t xe = (size_t)N * H * H * KC, ye = (size_t)N * H * H * CO;
  Bufs b;
  std::vector<uint16_t> hx(xe);
  srand(1);
  for (auto& v : hx) v = f2bf((float)(rand() % 2001 - 1000) / 1000.f);
  for (int i = 0; i < 3; ++i) {
    hipMalloc(&b.x[i], xe * 2);
    hipMalloc(&b.y[i], ye * 2);
    hipMemcpy(b.x[i], hx.data(), xe * 2, hipMemcpyHostToDevice);
  }
  std::vector<float> hw((size_t)CO * KC * 9), hsc(KC), hsh(KC);
  for (auto& v : hw) v = (float)(rand() % 2001 - 1000) / 4000.f;
  for (int c = 0; c < KC; ++c) { hsc[c] = 0.5f + (c % 7) * 0.1f; hsh[c] = -0.2f + (c % 5) * 0.1f; }
  auto hp = pack(hw, KC, CO, false);
  hipMalloc(&b.wp, hp.size() * 2);
  hipMemcpy(b.wp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
  hp = pack(hw, KC, CO, true);
  hipMalloc(&b.wpp, hp.size() * 2);
  hipMemcpy(b.wpp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&b.sc, KC * 4);
  hipMalloc(&b.sh, KC * 4);
  hipMemcpy(b.sc, hsc.data(), KC * 4, hipMemcpyHostToDevice);
  hipMemcpy(b.sh, hsh.data(), KC * 4, hipMemcpyHostToDevice);
  hipMalloc(&b.stats, (size_t)3 * CO * N * (H / 7) * (H / 14) * 4);
  if (KC == CO) {
    run<KC, TH, 14, NT, 0, 0, 8>("copy", b, N, H, CO, false, hx, hw, hsc, hsh);
    run<KC, TH, 14, NT, 1, 0, 8>("copy+bn", b, N, H, CO, false, hx, hw, hsc, hsh);
  }
  run<KC, TH, 14, NT, 0, 1>("mfma", b, N, H, CO, true, hx, hw, hsc, hsh);
  run<KC, TH, 14, NT, 1, 1>("mfma+bn", b, N, H, CO, true, hx, hw, hsc, hsh);
  run<KC, TH, 14, NT, 1, 3>("mfma+bn+stats", b, N, H, CO, true, hx, hw, hsc, hsh);
  run<KC, TH, 14, NT, 1, 11>("mfma+bn+stats16B", b, N, H, CO, true, hx, hw, hsc, hsh);
  if (NT == 2) {
    run<KC, TH, 14, NT, 1, 27>("mfma+bn+stats16B+st16B", b, N, H, CO, true, hx, hw, hsc, hsh);
    run<KC, TH, 14, NT, 0, 27>("dgrad +stats16B+st16B", b, N, H, CO, true, hx, hw, hsc, hsh);
  }
  if (KC <= 16) {
    run<KC, TH, 14, NT, 1, 15>("mfma+bn+stats16B+wpre", b, N, H, CO, true, hx, hw, hsc, hsh);
    run<KC, TH, 14, NT, 0, 15>("dgrad +stats16B+wpre", b, N, H, CO, true, hx, hw, hsc, hsh);
    run<KC, TH, 14, NT, 1, 15, 5>("mfma+bn+stats16B+wpre", b, N, H, CO, true, hx, hw, hsc, hsh);
  }
  for (int i = 0; i < 3; ++i) { hipFree(b.x[i]); hipFree(b.y[i]); }
  hipFree(b.wp); hipFree(b.wpp); hipFree(b.sc); hipFree(b.sh); hipFree(b.stats);
}

int main() {
  layer<16, 16, 14, 1>(64, 224);
  layer<16, 16, 7, 1>(64, 224);
  layer<32, 32, 7, 2>(64, 112);
  layer<64, 32, 7, 2>(64, 56);
  return 0;
}
