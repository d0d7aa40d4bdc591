// Experiment (not part of the product): what does "the last workgroup to finish does the final reduction" cost on
// MI355X, compared with a separate tiny reduction launch?  A streaming kernel (read 16 B/thread x U, write the same)
// leaves one row of 48 floats per workgroup; rows are reduced in two levels (groups of G rows).
//   mode 0: rows only (+ separate reduce kernels, timed together)
//   mode 1: fused, __threadfence() + atomicAdd tickets (the textbook protocol)
//   mode 2: fused, rows stored / loaded with agent-scope (sc1) accesses, s_waitcnt, relaxed atomic tickets
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int ROW = 48, G = 128;

__device__ __forceinline__ void store_row_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int per_wg,
                                                     float* rows, float* grows, float* result, unsigned* tickets) {
  __shared__ float red[4][ROW];
  __shared__ unsigned s_ticket;
  float acc = 0.f;
  const size_t base = (size_t)blockIdx.x * per_wg;
  for (int i = threadIdx.x; i < per_wg; i += 256) {
    u32x4 v = in[base + i];
    acc += __uint_as_float(v[0] & 0x3f800000u);
    v[1] += 1;
    out[base + i] = v;
  }
  // a [ROW] row per workgroup: wave sums folded (fixed order)
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < ROW) red[wave][lane] = acc + lane;
  __syncthreads();
  if (threadIdx.x < ROW) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (MODE == 2) store_row_sc1(rows + (size_t)blockIdx.x * ROW + threadIdx.x, v);
    else rows[(size_t)blockIdx.x * ROW + threadIdx.x] = v;
  }
  if (MODE == 0) return;
  const int ngroups = (gridDim.x + G - 1) / G, grp = blockIdx.x / G;
  const int gsize = min(G, (int)gridDim.x - grp * G);
  if (MODE == 1) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);  // my row stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = atomicAdd(&tickets[1 + grp], 1u);
  __syncthreads();
  if (s_ticket != (unsigned)gsize - 1) return;
  if (MODE == 1) __threadfence();
  // last of the group: fold the group's rows (threads: ROW columns x 5 row lanes, 240 active)
  {
    const int c = threadIdx.x % ROW, rl = threadIdx.x / ROW;
    float s = 0.f;
    if (rl < 5)
      for (int r = rl; r < gsize; r += 5) {
        const float* p = rows + ((size_t)grp * G + r) * ROW + c;
        s += MODE == 2 ? load_sc1(p) : *p;
      }
    __syncthreads();
    float* fold = &red[0][0];  // reuse: [5][ROW] > 4*ROW? no -> separate
    __shared__ float fold5[5][ROW];
    if (rl < 5) fold5[rl][c] = s;
    __syncthreads();
    if (threadIdx.x < ROW) {
      const float v = ((fold5[0][c] + fold5[1][c]) + (fold5[2][c] + fold5[3][c])) + fold5[4][c];
      if (MODE == 2) store_row_sc1(grows + (size_t)grp * ROW + c, v);
      else grows[(size_t)grp * ROW + c] = v;
    }
    (void)fold;
  }
  if (MODE == 1) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_ticket = atomicAdd(&tickets[0], 1u);
    tickets[1 + grp] = 0;  // re-arm for the next launch
  }
  __syncthreads();
  if (s_ticket != (unsigned)ngroups - 1) return;
  if (MODE == 1) __threadfence();
  if (threadIdx.x < ROW) {
    float s = 0.f;
    for (int g = 0; g < ngroups; ++g) {
      const float* p = grows + (size_t)g * ROW + threadIdx.x;
      s += MODE == 2 ? load_sc1(p) : *p;
    }
    result[threadIdx.x] = s;
  }
  if (threadIdx.x == 0) tickets[0] = 0;
}

__global__ __launch_bounds__(256) void reduce1(const float* rows, int nrows, float* grows) {
  __shared__ float fold5[5][ROW];
  const int grp = blockIdx.x, gsize = min(G, nrows - grp * G);
  const int c = threadIdx.x % ROW, rl = threadIdx.x / ROW;
  float s = 0.f;
  if (rl < 5)
    for (int r = rl; r < gsize; r += 5) s += rows[((size_t)grp * G + r) * ROW + c];
  if (rl < 5) fold5[rl][c] = s;
  __syncthreads();
  if (threadIdx.x < ROW) grows[(size_t)grp * ROW + c] = ((fold5[0][c] + fold5[1][c]) + (fold5[2][c] + fold5[3][c])) + fold5[4][c];
}
__global__ void reduce2(const float* grows, int ngroups, float* result) {
  if (threadIdx.x < ROW) {
    float s = 0.f;
    for (int g = 0; g < ngroups; ++g) s += grows[(size_t)g * ROW + threadIdx.x];
    result[threadIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 16384;
  const int per_wg = argc > 2 ? atoi(argv[2]) : 392;  // x16 B: 6.3 KB per workgroup ~ a 14x14x16 bf16 tile
  const size_t n = (size_t)nwg * per_wg;
  u32x4 *in[4], *out[4];
  for (int i = 0; i < 4; ++i) {
    hipMalloc(&in[i], n * 16); hipMalloc(&out[i], n * 16);
    hipMemset(in[i], 0x3f, n * 16);
  }
  float *rows, *grows, *result[3];
  unsigned* tickets;
  const int ngroups = (nwg + G - 1) / G;
  hipMalloc(&rows, (size_t)nwg * ROW * 4); hipMalloc(&grows, (size_t)ngroups * ROW * 4);
  for (int m = 0; m < 3; ++m) { hipMalloc(&result[m], ROW * 4); hipMemset(result[m], 0, ROW * 4); }
  hipMalloc(&tickets, (1 + ngroups) * 4); hipMemset(tickets, 0, (1 + ngroups) * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float h[3][ROW];
  for (int mode = 0; mode < 3; ++mode) {
    auto launch = [&](int i) {
      if (mode == 0) {
        hipLaunchKernelGGL(stream_kernel<0>, dim3(nwg), dim3(256), 0, 0, in[i % 4], out[i % 4], per_wg, rows, grows, result[0], tickets);
        hipLaunchKernelGGL(reduce1, dim3(ngroups), dim3(256), 0, 0, rows, nwg, grows);
        hipLaunchKernelGGL(reduce2, dim3(1), dim3(64), 0, 0, grows, ngroups, result[0]);
      } else if (mode == 1)
        hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(256), 0, 0, in[i % 4], out[i % 4], per_wg, rows, grows, result[1], tickets);
      else
        hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(256), 0, 0, in[i % 4], out[i % 4], per_wg, rows, grows, result[2], tickets);
    };
    for (int i = 0; i < 5; ++i) launch(i);
    hipEventRecord(e0);
    const int it = 40;
    for (int i = 0; i < it; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h[mode], result[mode], ROW * 4, hipMemcpyDeviceToHost);
    printf("mode %d: %7.2f us per iteration (%.0f GB/s)   result[0..2] = %.1f %.1f %.1f\n", mode, ms * 1e3 / it,
           2.0 * n * 16 / (ms * 1e3 / it) / 1e3, h[mode][0], h[mode][1], h[mode][2]);
  }
  // correctness stress: 300 more launches of each fused mode, every result compared with the 3-kernel one
  int bad = 0;
  for (int mode = 1; mode < 3; ++mode)
    for (int i = 0; i < 300; ++i) {
      hipMemset(result[mode], 0, ROW * 4);
      if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(256), 0, 0, in[i % 4], out[i % 4], per_wg, rows, grows, result[1], tickets);
      else hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(256), 0, 0, in[i % 4], out[i % 4], per_wg, rows, grows, result[2], tickets);
      float r[ROW];
      hipMemcpy(r, result[mode], ROW * 4, hipMemcpyDeviceToHost);
      if (memcmp(r, h[0], ROW * 4) != 0) ++bad;
    }
  printf("stress: %d mismatching launches of 600\n", bad);
  return bad != 0;
}
