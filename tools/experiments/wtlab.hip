// Experiment (round 4): does a heavy-writer kernel's end-of-kernel L2 write-back (8 XCDs, private L2s) cost less when its
// output leaves by WRITE-THROUGH (sc1) stores?  Chain per iteration:  W (writes B bytes, streaming) -> tiny (one workgroup,
// depends on W) -> R (reads the B bytes, streaming).  Timed as a whole over many iterations, stores plain / nontemporal / sc1.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/wtlab tools/experiments/wtlab.hip && tools/experiments/wtlab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE>  // 0 plain, 1 nontemporal, 2 sc1 (write-through)
__global__ __launch_bounds__(256) void writer(u32x4* __restrict__ out, const u32x4* __restrict__ in, size_t n16, unsigned salt) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    u32x4 v = in[i & 65535];  // (L2-resident source: the kernel is an output stream)
    v[0] += salt;
    if (MODE == 0) out[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, &out[i]);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(i * 16), 0, 16);
  }
}
__global__ void tiny(const u32x4* __restrict__ in, unsigned* __restrict__ flag) {
  if (threadIdx.x == 0) flag[0] = in[12345][0] + 1;
}
__global__ __launch_bounds__(256) void reader(const u32x4* __restrict__ in, size_t n16, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const u32x4 v = in[i];
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, u32x4* buf, const u32x4* src, size_t bytes, unsigned* flag, int with_tiny, int with_reader) {
  const size_t n16 = bytes / 16;
  const int wgs = 8192;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto chain = [&](unsigned s) {
    hipLaunchKernelGGL((writer<MODE>), dim3(wgs), dim3(256), 0, 0, buf, src, n16, s);
    if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, buf, flag);
    if (with_reader) hipLaunchKernelGGL(reader, dim3(wgs), dim3(256), 0, 0, buf, n16, flag);
  };
  // captured as a graph of 20 chains (the training step runs from a hipGraph)
  hipStream_t st; (void)hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; ++i) {
    hipLaunchKernelGGL((writer<MODE>), dim3(wgs), dim3(256), 0, st, buf, src, n16, (unsigned)i);
    if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, buf, flag);
    if (with_reader) hipLaunchKernelGGL(reader, dim3(wgs), dim3(256), 0, st, buf, n16, flag);
  }
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-12s %6.1f MB  tiny %d reader %d : %8.2f us per chain\n", name, bytes / 1e6, with_tiny, with_reader, ms * 1e3 / (reps * 20));
  (void)chain;
}

int main() {
  u32x4 *buf, *src; unsigned* flag;
  const size_t maxb = 256u << 20;
  (void)hipMalloc(&buf, maxb); (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&flag, 64);
  (void)hipMemset(src, 1, 1 << 20); (void)hipMemset(buf, 0, maxb);
  for (size_t mb : {103, 26, 6}) {
    const size_t bytes = mb << 20;
    for (int tr = 0; tr < 3; ++tr) {
      const int wt = tr >= 1, wr = tr >= 2;
      run<0>("plain", buf, src, bytes, flag, wt, wr);
      run<1>("nontemporal", buf, src, bytes, flag, wt, wr);
      run<2>("sc1", buf, src, bytes, flag, wt, wr);
    }
  }
  return 0;
}
