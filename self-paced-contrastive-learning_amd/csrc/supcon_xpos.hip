// SupConLoss1(exclude_other_pos=True) (contrastyou/losses/contrast_loss3.py:97-100 inside :59-110) and its gradient.
//
// With P = cat(z1, z2) (2n x d, unit rows), S = P P^T / t, m = max S (detached), L = S - m, E = exp(L), pos / neg the
// off-diagonal masks, c_i = sum_j pos_ij, q_i = sum_j neg_ij, N_i = sum_j neg_ij E_ij, rho_i = q_i / (c_i + q_i) + 1e-4:
//     l_ij  = L_ij - log(E_ij + N_i / rho_i + 1e-16)                (every positive is scored against the negatives only)
//     loss  = -(1 / 2n) sum_i (1 / c_i) sum_j pos_ij l_ij
//     G_ik  = d loss / d L_ik = -(1 / (2n c_i)) [ pos_ik (1 - E_ik / D_ik) - neg_ik (E_ik / rho_i) sum_j pos_ij / D_ij ],
//     D_ij  = E_ij + N_i / rho_i + 1e-16;          d loss / d P = (G + G^T) P / t.
// This variant is not used by the hooks (INFONCEHook.init_criterion builds SupConLoss1() with the default, semi_seg/hooks/
// infonce.py:92-94): it is built for completeness of the class's signature, as plain fp32 row kernels (one workgroup per
// row of the similarity matrix, dot products on the vector ALU, G materialised in the workspace), not tuned.
#include "common.hpp"

namespace spcl {

struct XposArgs {
  const float* z1;
  const float* z2;
  const float* labels;  // [n] or null
  const float* mask;    // [n][n] or null (== 1 positive, == 0 negative)
  int n, d;
  float inv_t;
  float* ws;  // [2n][2n] G | [2n] row max | [2n] norm defect | [2n] row loss
};

__device__ __forceinline__ const float* xrow(const XposArgs& a, int i) {
  return i < a.n ? a.z1 + (size_t)i * a.d : a.z2 + (size_t)(i - a.n) * a.d;
}
// pos / neg of the pair (i, j), contrast_loss3.py:43-57,69-78: masks tiled 2 x 2, diagonal removed
__device__ __forceinline__ void xpair(const XposArgs& a, int i, int j, bool& pos, bool& neg) {
  pos = neg = false;
  if (i == j) return;
  const int u = i % a.n, v = j % a.n;
  if (a.mask) {
    const float mv = a.mask[(size_t)u * a.n + v];
    pos = mv == 1.f;
    neg = mv == 0.f;
  } else if (a.labels) {
    pos = a.labels[u] == a.labels[v];
    neg = !pos;
  } else {
    pos = u == v;
    neg = !pos;
  }
}
__device__ __forceinline__ float xdot(const float* p, const float* q, int d) {
  float s = 0.f;
  for (int e = 0; e < d; ++e) s = fmaf(p[e], q[e], s);
  return s;
}
__device__ float block_sum(float v, float* red) {  // 256 threads, fixed order
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void xpos_rowmax_kernel(XposArgs a) {
  __shared__ float red[4];
  const int n2 = 2 * a.n, i = blockIdx.x;
  const float* pi = xrow(a, i);
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < n2; j += 256) mx = fmaxf(mx, xdot(pi, xrow(a, j), a.d) * a.inv_t);
  mx = block_max(mx, red);
  if (threadIdx.x == 0) {
    a.ws[(size_t)n2 * n2 + i] = mx;
    a.ws[(size_t)n2 * n2 + n2 + i] = fabsf(sqrtf(xdot(pi, pi, a.d)) - 1.f);
  }
}

__global__ __launch_bounds__(256) void xpos_rows_kernel(XposArgs a) {
  extern __shared__ float lrow[];  // L_i. of this row
  __shared__ float red[4];
  const int n2 = 2 * a.n, i = blockIdx.x;
  const float* rowmax = a.ws + (size_t)n2 * n2;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n2; j += 256) m = fmaxf(m, rowmax[j]);
  m = block_max(m, red);
  const float* pi = xrow(a, i);
  float nsum = 0.f, cpos = 0.f, cneg = 0.f;
  for (int j = threadIdx.x; j < n2; j += 256) {
    const float l = xdot(pi, xrow(a, j), a.d) * a.inv_t - m;
    lrow[j] = l;
    bool pos, neg;
    xpair(a, i, j, pos, neg);
    if (neg) nsum += expf(l);
    cpos += pos ? 1.f : 0.f;
    cneg += neg ? 1.f : 0.f;
  }
  nsum = block_sum(nsum, red);
  cpos = block_sum(cpos, red);
  cneg = block_sum(cneg, red);
  const float rho = cneg / (cpos + cneg) + 1e-4f;
  const float nr = nsum / rho;
  float tsum = 0.f, qsum = 0.f;
  for (int j = threadIdx.x; j < n2; j += 256) {
    bool pos, neg;
    xpair(a, i, j, pos, neg);
    if (pos) {
      const float dd = expf(lrow[j]) + nr + 1e-16f;
      tsum += lrow[j] - logf(dd);
      qsum += 1.f / dd;
    }
  }
  tsum = block_sum(tsum, red);
  qsum = block_sum(qsum, red);
  const float k = -1.f / ((float)n2 * cpos);
  float* g = a.ws + (size_t)i * n2;
  for (int j = threadIdx.x; j < n2; j += 256) {
    bool pos, neg;
    xpair(a, i, j, pos, neg);
    const float e = expf(lrow[j]);
    float v = 0.f;
    if (pos) v = k * (1.f - e / (e + nr + 1e-16f));
    else if (neg) v = -k * (e / rho) * qsum;
    g[j] = v;
  }
  if (threadIdx.x == 0) a.ws[(size_t)n2 * n2 + 2 * n2 + i] = tsum / cpos;  // NaN when the row has no positive, as the reference
}

__global__ __launch_bounds__(256) void xpos_finish_kernel(XposArgs a, float* out) {
  __shared__ float red[4];
  const int n2 = 2 * a.n;
  const float* rl = a.ws + (size_t)n2 * n2 + 2 * n2;
  const float* df = a.ws + (size_t)n2 * n2 + n2;
  float s = 0.f, mx = 0.f;
  for (int j = threadIdx.x; j < n2; j += 256) {
    s += rl[j];
    mx = fmaxf(mx, df[j]);
  }
  s = block_sum(s, red);
  mx = block_max(mx, red);
  if (threadIdx.x == 0) {
    out[0] = -s / (float)n2;
    out[1] = 1.f;   // rho (no self-paced weights here)
    out[2] = 1.f / (float)n2;
    out[3] = mx;    // largest | |row| - 1 |
  }
}

// dP_i = go / t * sum_k (G_ik + G_ki) P_k
__global__ __launch_bounds__(256) void xpos_backward_kernel(XposArgs a, const float* go, float* dz1, float* dz2) {
  const int n2 = 2 * a.n, i = blockIdx.x;
  const float scale = go[0] * a.inv_t;
  float* dst = i < a.n ? dz1 + (size_t)i * a.d : dz2 + (size_t)(i - a.n) * a.d;
  for (int e = threadIdx.x; e < a.d; e += 256) {
    float s = 0.f;
    for (int k = 0; k < n2; ++k) s = fmaf(a.ws[(size_t)i * n2 + k] + a.ws[(size_t)k * n2 + i], xrow(a, k)[e], s);
    dst[e] = s * scale;
  }
}

}  // namespace spcl

using namespace spcl;

extern "C" size_t spcl_supcon_xpos_workspace_bytes(int n, int d) {
  if (n <= 0 || d <= 0 || n > 4096) return 0;  // the row of logits lives in LDS (2n floats <= 32 KB)
  const size_t n2 = 2 * (size_t)n;
  return (n2 * n2 + 3 * n2 + 16) * sizeof(float);
}

extern "C" int spcl_supcon_xpos_forward(const float* z1, const float* z2, const float* labels, const float* mask, int n,
                                        int d, float temperature, float* ws, float* out, void* stream) {
  SPCL_CHECK_ARG(z1 && z2 && ws && out, "supcon_xpos_forward: null pointer");
  SPCL_CHECK_ARG(spcl_supcon_xpos_workspace_bytes(n, d) > 0 && temperature > 0.f, "supcon_xpos_forward: n=%d d=%d", n, d);
  hipStream_t st = (hipStream_t)stream;
  XposArgs a{z1, z2, labels, mask, n, d, 1.f / temperature, ws};
  SPCL_LAUNCH(xpos_rowmax_kernel, dim3(2 * n), dim3(256), 0, st, a);
  SPCL_LAUNCH(xpos_rows_kernel, dim3(2 * n), dim3(256), (size_t)2 * n * sizeof(float), st, a);
  SPCL_LAUNCH(xpos_finish_kernel, dim3(1), dim3(256), 0, st, a, out);
  SPCL_LAUNCH_CHECK("supcon_xpos_forward");
  return SPCL_OK;
}

extern "C" int spcl_supcon_xpos_backward(const float* z1, const float* z2, int n, int d, float temperature,
                                         const float* ws, const float* grad_out, float* dz1, float* dz2, void* stream) {
  SPCL_CHECK_ARG(z1 && z2 && ws && grad_out && dz1 && dz2, "supcon_xpos_backward: null pointer");
  SPCL_CHECK_ARG(spcl_supcon_xpos_workspace_bytes(n, d) > 0 && temperature > 0.f, "supcon_xpos_backward: n=%d d=%d", n, d);
  XposArgs a{z1, z2, nullptr, nullptr, n, d, 1.f / temperature, const_cast<float*>(ws)};
  SPCL_LAUNCH(xpos_backward_kernel, dim3(2 * n), dim3(256), 0, (hipStream_t)stream, a, grad_out, dz1, dz2);
  SPCL_LAUNCH_CHECK("supcon_xpos_backward");
  return SPCL_OK;
}
