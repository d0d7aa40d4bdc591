// The 9 x 9 autocorrelation of the zero-padded one-channel image batch ("image3"), as a device function
// so that it can run as its own launch (bn.hip) or in the spare workgroups of the weight-pack launch (conv.hip).
#pragma once
#include "common.hpp"

namespace spcl {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // (identical to the kernels' own typedef)

// R[t'][t] = sum over all pixels p of all images of img0[p + t' - 1] img0[p + t - 1] (img0: the image as the forward
// convolution saw it, bf16-rounded, zero outside) for the 45 tap pairs t' <= t, and sum_p img[p + t - 1] for the 9 taps:
// one partial row [64] per workgroup (a band of rows of one image; 54 used).  Threads walk the band's pixels, nine loads
// per pixel (L1-resident neighbours), 45 + 9 FMAs; fixed-order reduction (butterfly, then the four waves).
constexpr int ACORR_BAND = 8, ACORR_MAXW = 256;  // (22 KB of LDS: seven workgroups per CU, room beside the pack job)
// On the matrix pipe: D[t'][t] += A[t'][k] B[k][t] with k = 32 consecutive pixels of a row and A == B == the patch matrix
// P[k][t] = img0[pixel k + tap t] -- one 16-byte LDS read and one v_mfma_f32_16x16x32_bf16 per wave and 32 pixels.  The
// band sits in LDS as THREE bf16 copies shifted left by kx = 0, 1, 2 (a lane's 8 consecutive pixels of tap (ky, kx) are
// then one aligned read), zero beyond the image's last column AS A PIXEL (a padded pixel must not contribute through a
// tap that reaches back inside).  Column 9 of B is all ones: D[t'][9] = sum_p img0[p + t'] (the nine image sums, on the
// bf16-rounded image: their rounding errors are zero-mean, 1e-6 relative over 3.2 M pixels).  (The first version, 54 f32
// FMAs per pixel on the vector ALU: 29-55 us for this 13 MB read.)
__device__ __forceinline__ void image_autocorr_body(const float* __restrict__ img, int H, int W,
                                                    float* __restrict__ out, const int blk /* band index */) {
  constexpr int CW = ACORR_MAXW;               // pixel columns per copy row (multiple of 32)
  constexpr int NR = ACORR_BAND + 2;           // frame rows
  // A fragment read is 16 lanes (the taps) x 4 k-groups of 16 bytes from nine different (ky, kx) rows: with power-of-two
  // pitches all nine start in the same banks (the k-loop was LDS-bandwidth bound: 15 us for a 13 MB read).  Row pitch
  // = 12 sixteen-byte slots mod 16, copy pitch = 4 slots mod 16: tap (ky, kx), k-group g starts at slot 12 ky + 4 kx + g.
  constexpr int ROWB = CW * 2 + 192;
  constexpr int COPYB = (NR * ROWB + 255) / 256 * 256 + 64;
  __shared__ __attribute__((aligned(16))) unsigned char cp[3 * COPYB];  // [kx][frame row][pixel column] bf16
  __shared__ float dsum[4][16][16];
  const int bands = (H + ACORR_BAND - 1) / ACORR_BAND;
  const int n = blk / bands, b = blk - n * bands;
  const int r0 = b * ACORR_BAND, r1 = min(H, r0 + ACORR_BAND);
  const float* base = img + (size_t)n * H * W;
  const int nr = r1 - r0 + 2;
  const int WP = (W + 31) / 32 * 32;           // pixel columns walked (zeros beyond W)
  {  // stage: thread -> (frame row, 8-pixel group); all loads of an iteration in flight, then the three shifted copies
    const int groups = WP / 8;
    constexpr int SIT = (NR * (CW / 8) + 255) / 256;  // every load of the band in flight before the first conversion
    float vv[SIT][10];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int e = it * 256 + threadIdx.x;
      const int rr = e / groups, gc = e - rr * groups;
      const int gy = r0 - 1 + rr;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int gx = 8 * gc - 1 + j;
        const bool in = e < nr * groups && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const float x = base[(size_t)(in ? gy : 0) * W + (in ? gx : 0)];  // unconditional (clamped) load, then select:
        vv[it][j] = in ? x : 0.f;                                          // a load inside the ?: became 40 branches
      }
    }
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int e = it * 256 + threadIdx.x;
      if (e >= nr * groups) break;
      const int rr = e / groups, gc = e - rr * groups;
      const float* v = vv[it];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int p0 = 8 * gc + 2 * q, p1 = p0 + 1;  // pixel columns of this pair
          const uint32_t lo = p0 < W ? f32_to_bf16(v[2 * q + kx]) : 0, hi = p1 < W ? f32_to_bf16(v[2 * q + 1 + kx]) : 0;
          w[q] = lo | (hi << 16);
        }
        *(u32x4*)(cp + (size_t)kx * COPYB + (size_t)rr * ROWB + gc * 16) = (u32x4){w[0], w[1], w[2], w[3]};
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
  const int tap = r16 < 9 ? r16 : 0, ky = tap / 3, kx = tap - 3 * ky;
  f32x4 D = {0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int yy = wave; yy < r1 - r0; yy += 4) {
    const unsigned char* rowp = cp + (size_t)kx * COPYB + (size_t)(yy + ky) * ROWB + g * 16;
    for (int x0 = 0; x0 < WP; x0 += 32) {
      const u32x4 fr = *(const u32x4*)(rowp + x0 * 2);
      // (pixels beyond W read zeros in every copy, so the ones column needs no mask of its own: A is zero there)
      D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr),
                                                  __builtin_bit_cast(bf16x8, r16 == 9 ? ones : fr), D, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) dsum[wave][4 * g + r][r16] = D[r];  // D[row t' = 4 g + r][column t = r16]
  __syncthreads();
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    float v = 0.f;
    if (k < 54) {
      int ra, cb;
      if (k < 45) {  // upper triangle in the order of acorr_index
        ra = 0;
        int rem = k;
        while (rem >= 9 - ra) { rem -= 9 - ra; ++ra; }
        cb = ra + rem;
      } else {
        ra = k - 45;
        cb = 9;
      }
      v = (dsum[0][ra][cb] + dsum[1][ra][cb]) + (dsum[2][ra][cb] + dsum[3][ra][cb]);
    }
    out[(size_t)blk * 64 + k] = v;
  }
}


__host__ __device__ inline int image_autocorr_bands(int H) { return (H + ACORR_BAND - 1) / ACORR_BAND; }

}  // namespace spcl
