// Experiment (not part of the product): how fast can the memory system serve the conv kernel's ACCESS PATTERN with
// no compute at all?  Each workgroup reads a (TH+2)x(TW+2) halo tile of C bf16 channels (16-byte chunks, the conv
// staging map) and writes the TH x TW interior back, NHWC.  Sweeps workgroup size and tile shape.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int TH, int TW>
__global__ void tilecopy(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int N, int H, int W, int C,
                         int tilesX, int tilesY, int wpt /*waves per tile*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int HW_ = TW + 2, NHALO = (TH + 2) * HW_;
  const int CP = C / 8;
  const int nthreads = blockDim.x;
  const int tile = blockIdx.x;
  const int tpi = tilesX * tilesY;
  const int n = tile / tpi, tr = tile % tpi, ty = tr / tilesX, tx = tr % tilesX;
  const int y0 = ty * TH, x0 = tx * TW;
  u32x4 acc = {0, 0, 0, 0};
  for (int idx = threadIdx.x; idx < NHALO * CP; idx += nthreads) {
    const int q = idx / CP, ch = idx % CP;
    const int hy = q / HW_, hx = q % HW_;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    u32x4 v = {0, 0, 0, 0};
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const u32x4*)(x + (((size_t)n * H + gy) * W + gx) * C + ch * 8);
    *(u32x4*)(lds + (size_t)idx * 16) = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < TH * TW * CP; idx += nthreads) {
    const int p = idx / CP, ch = idx % CP;
    const int py = p / TW, px = p % TW;
    u32x4 v = *(const u32x4*)(lds + ((size_t)((py + 1) * HW_ + px + 1) * CP + ch) * 16);
    *(u32x4*)(y + (((size_t)n * H + y0 + py) * W + x0 + px) * C + ch * 8) = v;
  }
}

template <int TH, int TW>
static void run(const char* name, const uint16_t* x, uint16_t* y, int N, int H, int W, int C, int threads) {
  const int tilesX = W / TW, tilesY = H / TH;
  const int tiles = N * tilesX * tilesY;
  const size_t lds = (size_t)(TH + 2) * (TW + 2) * C * 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tilecopy<TH, TW>), dim3(tiles), dim3(threads), lds, 0, x, y, N, H, W, C, tilesX, tilesY, 1);
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((tilecopy<TH, TW>), dim3(tiles), dim3(threads), lds, 0, x, y, N, H, W, C, tilesX, tilesY, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  const double bytes = 2.0 * N * H * W * C * 2;
  printf("%-10s C=%3d H=%3d threads=%4d tiles=%6d lds=%6zu : %7.1f us  %6.0f GB/s\n", name, C, H, threads, tiles, lds, us,
         bytes / us / 1e3);
}

int main() {
  const int N = 64;
  struct L { int H, C; } layers[] = {{224, 16}, {112, 32}, {56, 64}};
  for (auto l : layers) {
    const size_t elems = (size_t)N * l.H * l.H * l.C;
    uint16_t *x, *y;
    hipMalloc(&x, elems * 2);
    hipMalloc(&y, elems * 2);
    hipMemset(x, 1, elems * 2);
    for (int th : {64, 128, 256}) {
      run<14, 14>("14x14", x, y, N, l.H, l.H, l.C, th);
      run<7, 14>("7x14", x, y, N, l.H, l.H, l.C, th);
      run<7, 28>("7x28", x, y, N, l.H, l.H, l.C, th);
      run<14, 28>("14x28", x, y, N, l.H, l.H, l.C, th);
      run<28, 28>("28x28", x, y, N, l.H, l.H, l.C, th);
      run<4, 56>("4x56", x, y, N, l.H, l.H, l.C, th);
    }
    hipFree(x);
    hipFree(y);
  }
  return 0;
}
