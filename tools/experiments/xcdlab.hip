// Experiment (round 4): does a chain of tiny dependent kernels run faster when every link lands on ONE XCD?
// Link = kernel of NW workgroups; workgroup w reads 4 KB that workgroup (w + 1) % NW of the PREVIOUS link wrote (so data always
// crosses workgroups), adds, writes 4 KB.  Variants: A  grid = NW (workgroups dealt round-robin over the 8 XCDs: consumer and
// producer of a block usually sit on different XCDs);  B  grid = 8 NW, only blockIdx % 8 == 0 works (all on XCD 0).
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/xcdlab tools/experiments/xcdlab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int ONE_XCD>
__global__ __launch_bounds__(256) void link(const float4* __restrict__ in, float4* __restrict__ out, int nw) {
  int w = blockIdx.x;
  if (ONE_XCD) {
    if (w & 7) return;
    w >>= 3;
  }
  const int src = (w + 1) % nw;
  float4 v = in[src * 256 + threadIdx.x];
  v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
  out[w * 256 + threadIdx.x] = v;
}

int main() {
  float4 *a, *b;
  (void)hipMalloc(&a, 1 << 20); (void)hipMalloc(&b, 1 << 20);
  (void)hipMemset(a, 0, 1 << 20); (void)hipMemset(b, 0, 1 << 20);
  hipStream_t st; (void)hipStreamCreate(&st);
  for (int nw : {1, 8, 64}) {
    for (int variant = 0; variant < 2; ++variant) {
      hipGraph_t g; hipGraphExec_t ge;
      (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 200; ++i) {
        float4* in = (i & 1) ? b : a; float4* out = (i & 1) ? a : b;
        if (variant == 0) hipLaunchKernelGGL(link<0>, dim3(nw), dim3(256), 0, st, in, out, nw);
        else hipLaunchKernelGGL(link<1>, dim3(8 * nw), dim3(256), 0, st, in, out, nw);
      }
      (void)hipStreamEndCapture(st, &g);
      (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
      (void)hipStreamSynchronize(st);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, st);
      for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, st);
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("nw %3d  %s : %6.2f us per link\n", nw, variant ? "all on one XCD (grid x 8)" : "round-robin over XCDs   ", ms * 1e3 / 2000);
    }
  }
  return 0;
}
