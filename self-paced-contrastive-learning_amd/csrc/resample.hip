// Decoder data movement of the full UNet (semi_seg/arch/unet.py:85-97 nn.Upsample(scale_factor=2), :194,204,214,224
// torch.cat((skip, up), dim=1)) and its autograd backward as NHWC streaming kernels, 16-byte chunks:
//   upsample2x fwd : y[n, 2h+a, 2w+b, :] = x[n, h, w, :]          (nearest; one read, four writes)
//   upsample2x bwd : dx[n, h, w, :] = sum_{a,b} dy[n, 2h+a, 2w+b, :]   (fp32 sum, rounded once)
//   concat2 / split2: out[p, :] = (a[p, :], b[p, :]) and back -- channel counts are multiples of 16, so every source
//   and destination is a whole number of 16-byte chunks.
// PyTorch's channels-last upsample ran at 0.3 TB/s and cat at 2 TB/s on these shapes (tools/prof_torch_ops.py finetune).
#include "common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                             unsigned npix_in, int W, int CPC) {
  const size_t total = (size_t)npix_in * CPC;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const unsigned p = (unsigned)(idx / CPC), c = (unsigned)(idx - (size_t)p * CPC);
    const unsigned r = p / W, w = p - r * W;  // r = n*H + h  ->  output rows 2r, 2r+1
    const u32x4 v = x[idx];
    u32x4* o = y + ((size_t)(2 * r) * (2 * W) + 2 * w) * CPC + c;
    o[0] = v;
    o[CPC] = v;
    o[(size_t)2 * W * CPC] = v;
    o[(size_t)2 * W * CPC + CPC] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx,
                                                             unsigned npix_in, int W, int CS) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int CPC = CS / EPC;
  const size_t total = (size_t)npix_in * CPC;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const unsigned p = (unsigned)(idx / CPC), c = (unsigned)(idx - (size_t)p * CPC);
    const unsigned r = p / W, w = p - r * W;
    const T* s = dy + (((size_t)(2 * r) * (2 * W) + 2 * w) * CPC + c) * EPC;
    const size_t rowstep = (size_t)2 * W * CS;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      acc[e] = (Elem<T>::load(s + e) + Elem<T>::load(s + CS + e)) +
               (Elem<T>::load(s + rowstep + e) + Elem<T>::load(s + rowstep + CS + e));
    T* d = dx + idx * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) Elem<T>::store(d + e, acc[e]);
  }
}

// split == false: out[p] = (a[p], b[p]);  split == true: (a[p], b[p]) = out[p]
template <bool SPLIT>
__global__ __launch_bounds__(256) void concat2_kernel(u32x4* __restrict__ a, u32x4* __restrict__ b,
                                                      u32x4* __restrict__ out, size_t npix, int CPA, int CPB) {
  const int CPO = CPA + CPB;
  const size_t total = npix * CPO;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t p = idx / CPO;
    const int c = (int)(idx - p * CPO);
    u32x4* src = c < CPA ? a + p * CPA + c : b + p * CPB + (c - CPA);
    if (SPLIT) *src = out[idx];
    else out[idx] = *src;
  }
}

static int rs_grid(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace spcl

using namespace spcl;

extern "C" int spcl_upsample2x_forward(const void* x, void* y, int dtype, int N, int H, int W, int CS, void* stream) {
  SPCL_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0, "upsample2x_forward: bad args");
  SPCL_CHECK_ARG(dtype == SPCL_F32 || dtype == SPCL_BF16, "upsample2x_forward: dtype %d", dtype);
  const int es = dtype == SPCL_F32 ? 4 : 2, CPC = CS * es / 16;
  const size_t npix = (size_t)N * H * W;
  SPCL_CHECK_ARG(npix < 0xffffffffull, "upsample2x_forward: too many pixels");
  prof_cost((double)npix * CS * es * 5.0, 0.0);
  SPCL_LAUNCH(upsample2x_fwd_kernel, dim3(rs_grid(npix * CPC)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x,
              (u32x4*)y, (unsigned)npix, W, CPC);
  SPCL_LAUNCH_CHECK("upsample2x_forward");
  return SPCL_OK;
}

extern "C" int spcl_upsample2x_backward(const void* dy, void* dx, int dtype, int N, int H, int W, int CS, void* stream) {
  SPCL_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && CS > 0 && CS % 16 == 0, "upsample2x_backward: bad args");
  const size_t npix = (size_t)N * H * W;  // H, W = the LOW-resolution size (that of dx)
  SPCL_CHECK_ARG(npix < 0xffffffffull, "upsample2x_backward: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  prof_cost((double)npix * CS * (dtype == SPCL_F32 ? 4.0 : 2.0) * 5.0, 0.0);
  if (dtype == SPCL_F32)
    SPCL_LAUNCH(upsample2x_bwd_kernel<float>, dim3(rs_grid(npix * (CS / 4))), dim3(256), 0, st, (const float*)dy,
                (float*)dx, (unsigned)npix, W, CS);
  else if (dtype == SPCL_BF16)
    SPCL_LAUNCH(upsample2x_bwd_kernel<bf16_t>, dim3(rs_grid(npix * (CS / 8))), dim3(256), 0, st, (const bf16_t*)dy,
                (bf16_t*)dx, (unsigned)npix, W, CS);
  else {
    set_error("upsample2x_backward: dtype %d", dtype);
    return SPCL_EINVAL;
  }
  SPCL_LAUNCH_CHECK("upsample2x_backward");
  return SPCL_OK;
}

extern "C" int spcl_concat2_channels(const void* a, const void* b, void* out, int elem_size, size_t npix, int CA, int CB,
                                     void* stream) {
  SPCL_CHECK_ARG(a && b && out && npix > 0 && CA > 0 && CB > 0, "concat2_channels: bad args");
  SPCL_CHECK_ARG((CA * elem_size) % 16 == 0 && (CB * elem_size) % 16 == 0, "concat2_channels: channel runs must be "
                 "multiples of 16 bytes");
  const int CPA = CA * elem_size / 16, CPB = CB * elem_size / 16;
  prof_cost(2.0 * npix * (CA + CB) * elem_size, 0.0);
  SPCL_LAUNCH(concat2_kernel<false>, dim3(rs_grid(npix * (CPA + CPB))), dim3(256), 0, (hipStream_t)stream, (u32x4*)a,
              (u32x4*)b, (u32x4*)out, npix, CPA, CPB);
  SPCL_LAUNCH_CHECK("concat2_channels");
  return SPCL_OK;
}

extern "C" int spcl_split2_channels(const void* in, void* a, void* b, int elem_size, size_t npix, int CA, int CB,
                                    void* stream) {
  SPCL_CHECK_ARG(a && b && in && npix > 0 && CA > 0 && CB > 0, "split2_channels: bad args");
  SPCL_CHECK_ARG((CA * elem_size) % 16 == 0 && (CB * elem_size) % 16 == 0, "split2_channels: channel runs must be "
                 "multiples of 16 bytes");
  const int CPA = CA * elem_size / 16, CPB = CB * elem_size / 16;
  prof_cost(2.0 * npix * (CA + CB) * elem_size, 0.0);
  SPCL_LAUNCH(concat2_kernel<true>, dim3(rs_grid(npix * (CPA + CPB))), dim3(256), 0, (hipStream_t)stream, (u32x4*)a,
              (u32x4*)b, (u32x4*)in, npix, CPA, CPB);
  SPCL_LAUNCH_CHECK("split2_channels");
  return SPCL_OK;
}
