// Experiment (not part of the product): what limits the BatchNorm streaming kernels?  Read-only sum, the two-tensor
// BN-backward reduction and the three-stream apply pass at the C1 size (N=64, 224x224, 16 bf16 channels), sweeping
// unroll, grid size and occupancy bound.  Buffers rotate so that the 256 MiB infinity cache cannot hold them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void k_sum(const u32x4* __restrict__ y,
                                                                                       size_t nchunk, float* out) {
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += U * stride) {
    u32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = (i + u * stride < nchunk) ? y[i + u * stride] : (u32x4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) s += lo(r[u][w]) + hi(r[u][w]);
  }
  if (s == 12345.678f) out[0] = s;
}

template <int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void k_reduce(
    const u32x4* __restrict__ y, const u32x4* __restrict__ g, size_t nchunk, const float* __restrict__ coef,
    float* __restrict__ partial) {
  // CS = 16 bf16: 2 chunks per pixel; thread's chunk parity is fixed (stride is even)
  const int cc = threadIdx.x & 1;
  float sc[8], sh[8], mu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = coef[cc * 8 + e]; sh[e] = coef[16 + cc * 8 + e]; mu[e] = coef[32 + cc * 8 + e]; }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += U * stride) {
    u32x4 ry[U], rg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = i + u * stride < nchunk;
      ry[u] = ok ? y[i + u * stride] : (u32x4){0, 0, 0, 0};
      rg[u] = ok ? g[i + u * stride] : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = 2 * w + h;
          const float yv = h ? hi(ry[u][w]) : lo(ry[u][w]);
          const float gv = h ? hi(rg[u][w]) : lo(rg[u][w]);
          const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? gv : 0.f;
          s1[e] += dz;
          s2[e] = fmaf(dz, yv - mu[e], s2[e]);
        }
  }
  float t = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) t += s1[e] + s2[e];
  partial[(size_t)blockIdx.x * 256 + threadIdx.x] = t;
}

template <int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void k_apply(
    const u32x4* __restrict__ y, const u32x4* __restrict__ g, size_t nchunk, const float* __restrict__ coef,
    u32x4* __restrict__ dy) {
  const int cc = threadIdx.x & 1;
  float sc[8], sh[8], A[8], B[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = coef[cc * 8 + e]; sh[e] = coef[16 + cc * 8 + e]; A[e] = coef[32 + cc * 8 + e]; B[e] = coef[48 + cc * 8 + e];
  }
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += U * stride) {
    u32x4 ry[U], rg[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < nchunk) { ry[u] = y[i + u * stride]; rg[u] = g[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u * stride >= nchunk) break;
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = 2 * w + h;
          const float yv = h ? hi(ry[u][w]) : lo(ry[u][w]);
          const float gv = h ? hi(rg[u][w]) : lo(rg[u][w]);
          const float dz = fmaf(sc[e], yv, sh[e]) > 0.f ? gv : 0.f;
          v[h] = fmaf(sc[e], dz, fmaf(A[e], yv, B[e]));
        }
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        const f2 fv = {v[0], v[1]};
        o[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fv, b2));
      }
      dy[i + u * stride] = o;
    }
  }
}

static const int NBUF = 4;
static u32x4 *Y[NBUF], *G[NBUF], *D[NBUF];
static float *coef, *partial;
static size_t nchunk;

template <typename F>
static void timeit(const char* name, int grid, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(i, grid);
  hipEventRecord(e0);
  const int it = 24;
  for (int i = 0; i < it; ++i) launch(i, grid);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  printf("%-28s grid=%5d : %7.1f us  %6.0f GB/s\n", name, grid, us, bytes / us / 1e3);
}

template <int U, int WPE>
static void sweep() {
  char nm[64];
  const double tb = (double)nchunk * 16;
  for (int grid : {1024, 2048, 4096, 8192}) {
    snprintf(nm, sizeof nm, "sum    U=%d wpe=%d", U, WPE);
    timeit(nm, grid, tb, [&](int i, int g) { hipLaunchKernelGGL((k_sum<U, WPE>), dim3(g), dim3(256), 0, 0, Y[i % NBUF], nchunk, partial); });
  }
  for (int grid : {1024, 2048, 4096, 8192}) {
    snprintf(nm, sizeof nm, "reduce U=%d wpe=%d", U, WPE);
    timeit(nm, grid, 2 * tb, [&](int i, int g) { hipLaunchKernelGGL((k_reduce<U, WPE>), dim3(g), dim3(256), 0, 0, Y[i % NBUF], G[i % NBUF], nchunk, coef, partial); });
  }
  for (int grid : {1024, 2048, 4096, 8192}) {
    snprintf(nm, sizeof nm, "apply  U=%d wpe=%d", U, WPE);
    timeit(nm, grid, 3 * tb, [&](int i, int g) { hipLaunchKernelGGL((k_apply<U, WPE>), dim3(g), dim3(256), 0, 0, Y[i % NBUF], G[i % NBUF], nchunk, coef, D[i % NBUF]); });
  }
}

int main() {
  const size_t bytes = (size_t)64 * 224 * 224 * 16 * 2;
  nchunk = bytes / 16;
  for (int i = 0; i < NBUF; ++i) {
    hipMalloc(&Y[i], bytes); hipMalloc(&G[i], bytes); hipMalloc(&D[i], bytes);
    hipMemset(Y[i], 0x3c, bytes); hipMemset(G[i], 0x3d, bytes);
  }
  hipMalloc(&coef, 64 * 4);
  hipMemset(coef, 0, 64 * 4);
  hipMalloc(&partial, 8192 * 256 * 4);
  sweep<1, 8>();
  sweep<2, 8>();
  sweep<2, 5>();
  sweep<4, 4>();
  sweep<4, 8>();
  return 0;
}
