// Experiment: how fast does MI355X start workgroups?  Near-empty kernels, 256 threads per workgroup, varying the
// dynamic LDS size and the register footprint; time per launch vs number of workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NV>
__global__ __launch_bounds__(256) void k(float* out) {
  extern __shared__ float lds[];
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = out[i] + threadIdx.x;  // NV live registers
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i] * v[(i + 1) % NV];
  if (s == 123.456f) { lds[threadIdx.x] = s; out[blockIdx.x] = lds[threadIdx.x ^ 1]; }
}
template <int NV>
static void run(const char* name, int wgs, int lds, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<NV>, dim3(wgs), dim3(256), lds, 0, out);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<NV>, dim3(wgs), dim3(256), lds, 0, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s wgs=%5d lds=%6d : %7.2f us per launch\n", name, wgs, lds, ms * 1e3 / 20);
}
int main() {
  float* out; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
  for (int wgs : {256, 512, 1024, 2048, 8192}) {
    run<4>("4 regs", wgs, 0, out);
    run<4>("4 regs", wgs, 32768, out);
    run<100>("100 regs", wgs, 0, out);
    run<100>("100 regs", wgs, 32768, out);
  }
  return 0;
}
