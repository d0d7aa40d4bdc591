// Per-sample random flips of a batch (semi_seg/epochers/new_epocher.py:112 TensorRandomFlip(axis=[1,2]) applied
// sample by sample in new_pretrain.py:57-58,64-65) as ONE streaming launch: out[n] = flip(x[n], dims(flags[n])).
// The reference loops over the samples on the host (one flip kernel each); the batched torch formulation is still
// ~10 launches (index_select / flip / index_copy per pattern).  16-byte vectors, reversed in registers for W flips.
#include "common.hpp"

namespace spcl {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void flip_batch_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int C,
                                                         int H, int W, const uint8_t* __restrict__ flags) {
  const int WV = W / VEC;
  const size_t total = (size_t)N * C * H * WV;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int wv = (int)(idx % WV);
    size_t r = idx / WV;
    const int h = (int)(r % H);
    r /= H;  // r = n*C + c
    const int n = (int)(r / C);
    const uint8_t f = flags[n];
    const int sh = (f & 1) ? H - 1 - h : h;
    const int swv = (f & 2) ? WV - 1 - wv : wv;
    const T* src = x + (r * H + sh) * (size_t)W + (size_t)swv * VEC;
    T v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = src[e];  // contiguous: one vector load
    T* dst = out + (r * H + h) * (size_t)W + (size_t)wv * VEC;
    if (f & 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[e] = v[VEC - 1 - e];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[e] = v[e];
    }
  }
}

template <typename T>
static void launch_flip(const void* x, void* out, int N, int C, int H, int W, const uint8_t* flags, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const bool vec = W % V == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const size_t total = (size_t)N * C * H * (vec ? W / V : W);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec) SPCL_LAUNCH((flip_batch_kernel<T, V>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, N,
                              C, H, W, flags);
  else SPCL_LAUNCH((flip_batch_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, N, C,
                          H, W, flags);
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_flip_batch(const void* x, void* out, int elem_size, int N, int C, int H, int W,
                               const uint8_t* flags, void* stream) {
  SPCL_CHECK_ARG(x && out && flags, "flip_batch: null pointer");
  SPCL_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "flip_batch: bad shape");
  SPCL_CHECK_ARG(x != out, "flip_batch: in-place is not supported");
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 4) launch_flip<uint32_t>(x, out, N, C, H, W, flags, st);
  else if (elem_size == 2) launch_flip<uint16_t>(x, out, N, C, H, W, flags, st);
  else {
    set_error("flip_batch: elem_size %d", elem_size);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("flip_batch");
  return SPCL_OK;
}
