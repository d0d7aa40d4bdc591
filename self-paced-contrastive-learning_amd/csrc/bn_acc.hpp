// BatchNorm statistics as EXACT fixed-point sums: what lets a producer kernel's epilogue ADD its tile's sums and the next
// launch's prologue READ the totals, with no reduction / finalize launch in between (semi_seg/arch/unet.py:73,76:
// nn.BatchNorm2d(momentum=0.1) in train mode, and its autograd backward).
//
// A value v (an f32 tile sum, exact as a double) is split into two 64-bit integers
//     hi = rint(v * 2^10)                      (multiples of 2^-10, |hi| < 2^50 for |v| < 2^40)
//     lo = rint((v - hi * 2^-10) * 2^60)       (the remainder, |lo| <= 2^49; the only rounding: 2^-61 absolute)
// and each is added to its own accumulator word with an integer atomic.  Integer addition is associative and commutative:
// whatever order the workgroups arrive in, the totals are the same bits -- the determinism of the fixed-order reductions
// they replace, without an order.  Range: 2^13 adds of |v| < 2^40 per word cannot overflow; resolution 8.7e-19 (gradient
// sums of 1e-12 still carry 20 bits).  A value outside the range (or a NaN) raises the block's flag word instead: the
// consumer then produces NaN coefficients, the loss turns NaN and the criterion raises, as the reference does
// (contrastyou/losses/contrast_loss3.py:203-204).
//
// Layout of one accumulator block (spcl_bn_acc_elems(CS) words, zeroed by the caller before the producer launch):
//     acc[replica][channel][4] = {s1 hi, s1 lo, s2 hi, s2 lo},  then BN_ACC_FLAG_WORDS flag words.
// BN_ACC_REPLICAS copies, chosen by the producer workgroup's index mod 8 (= its XCD under round-robin dispatch), bound the
// adders per address to tiles / 8 -- same-address atomics serialise at the memory side (MI355X_MICROARCH.md, "Global float
// atomics": one row for everybody is 14x slower) -- and the consumer's prologue to 8 x 32 bytes per channel.
// forward:  s1 = sum x,   s2 = sum x^2            (x = the convolution's f32 accumulators, as the per-tile rows had them)
// backward: s1 = sum dz,  s2 = sum dz (y - mean)  (dz = the gradient w.r.t. the BatchNorm's output behind the ReLU gate)
#pragma once
#include "common.hpp"

namespace spcl {

#ifndef SPCL_ACC_REPLICAS
#define SPCL_ACC_REPLICAS 8
#endif
constexpr int BN_ACC_REPLICAS = SPCL_ACC_REPLICAS;
constexpr int BN_ACC_FLAG_WORDS = 4;  // (one used; 32 bytes keep the next block 32-byte aligned)
constexpr double BN_ACC_HI = 1024.0, BN_ACC_HI_INV = 1.0 / 1024.0;
constexpr double BN_ACC_LO = 1152921504606846976.0 /* 2^60 */, BN_ACC_LO_INV = 1.0 / 1152921504606846976.0;
constexpr float BN_ACC_MAX = 1099511627776.0f;  // 2^40
// Layers with more producer tiles than this keep the per-tile rows + reduction launch.  Same-address adds serialise at the
// memory side: measured ~18 ns each (ONE replica for the 2 048 tiles of Conv3.a: 17 -> 53 us; two: 33; four: 22.6; eight: 19),
// i.e. tiles / 8 x 18 ns per address, spread over the launch.  At 2 048 tiles (Conv3 at N = 64, 224^2) the forward
// statistics cost their producers + 2.8 / + 3 us and the consumer prologues + 4 / + 1 us -- what the two reduction launches
// they replace cost (same-box A/B of the whole step with the limit at 1 024 and at 4 096: 1 045 vs 1 050 us, bench medians
// 1 053 vs 1 055: equal) -- so the limit only decides the launch count there; the dgrads' backward sums cost their producers
// + 0 .. 1.4 us at the same tile count (longer, more staggered epilogues).  Blocks 1 / 2 (8 192 / 16 384 tiles) stay on rows.
#ifndef SPCL_ACC_FWD_TILES
#define SPCL_ACC_FWD_TILES 4096
#endif
constexpr int BN_ACC_MAX_TILES_FWD = SPCL_ACC_FWD_TILES;
constexpr int BN_ACC_MAX_TILES_BWD = 4096;

__host__ __device__ inline size_t bn_acc_words(int CS) { return (size_t)BN_ACC_REPLICAS * CS * 4 + BN_ACC_FLAG_WORDS; }

// what a consumer kernel needs to turn a FORWARD block into scale / shift (and, its first workgroup, to leave mean / invstd
// / scale / shift and the running statistics where the old finalize launch left them)
struct BnAccFwd {
  const long long* acc;  // null: the kernel takes its coefficients from the scale / shift arrays as before
  const float* gamma;
  const float* beta;
  float* running_mean;  // may be null (track_running_stats off / frozen)
  float* running_var;
  long long* nbt;
  float* st;  // [4][CS]: mean, invstd, scale, shift
  float momentum, eps, count;
  int C, CS;
  double inv_count;  // 1 / count, formed on the host (a double division in every consumer workgroup's prologue otherwise)
};

// ... and a BACKWARD block into the folded coefficients of dy = scale dz + A y + B (first workgroup: dgamma, dbeta)
struct BnAccBwd {
  const long long* acc;  // null: coefficients from the `ab` array as before
  const float* st;       // [4][CS] of the forward
  float* dgamma;         // [C]
  float* dbeta;
  float count;
  int training, C, CS;
};

#ifdef __HIPCC__
__device__ __forceinline__ void bn_acc_atomic_add(long long* p, long long v) {
  // (result unused: a no-return global_atomic_add_x2, performed at the memory side)
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the two limbs of v; ok = false (and zeros) when v is outside the representable range or not a number
__device__ __forceinline__ void bn_acc_split(float v, long long& hi, long long& lo, bool& ok) {
  ok = fabsf(v) < BN_ACC_MAX;  // (false for NaN)
  const double d = ok ? (double)v : 0.0;
  const double h = rint(d * BN_ACC_HI);
  const double l = rint(fma(h, -BN_ACC_HI_INV, d) * BN_ACC_LO);
  hi = (long long)h;
  lo = (long long)l;
}

// Producer, conv epilogue form: after row16_sum every lane of a 16-lane row holds the row's totals s1[4], s2[4] of the four
// channels cb .. cb + 3.  Lane r16 of the row adds ONE word: channel r16 & 3, sum (r16 >> 2) & 1, limb r16 >> 3 -- the 16
// lanes of a row cover the 4 x 32 contiguous bytes of their channels, the four rows of a wave 512 contiguous bytes per
// 16-channel n-tile: one wave-wide atomic instruction per n-tile.
__device__ __forceinline__ void bn_acc_add_row16(long long* acc, int CS, int replica, int cb, int r16, f32x4 s1, f32x4 s2,
                                                 bool active = true) {
  const int q = r16 & 3, which = (r16 >> 2) & 1, limb = r16 >> 3;
  float v = which ? s2[0] : s1[0];
#pragma unroll
  for (int r = 1; r < 4; ++r) v = q == r ? (which ? s2[r] : s1[r]) : v;
  long long hi, lo;
  bool ok;
  bn_acc_split(v, hi, lo, ok);
  if (!active) return;
  long long* row = acc + ((size_t)replica * CS + cb + q) * 4;
  bn_acc_atomic_add(row + which * 2 + limb, limb ? lo : hi);
  if (!ok) bn_acc_atomic_add(acc + (size_t)BN_ACC_REPLICAS * CS * 4, 1);  // rare: raise the block's flag
}

// Producer, generic form: one thread adds ONE WORD -- limb `limb` of sum `which` of channel c, v = that sum.  Called with
// consecutive threads on consecutive words (thread o: c = o / 4, which = (o / 2) % 2, limb = o % 2) a wave's instruction covers
// 512 contiguous bytes; one lane per channel with scattered 8-byte words is the 17x slower shape of MI355X_MICROARCH.md's
// atomics table (measured here: + 7 us on a 256-workgroup reduction pass).
__device__ __forceinline__ void bn_acc_add_word(long long* acc, int CS, int replica, int c, int which, int limb, float v) {
  long long hi, lo;
  bool ok;
  bn_acc_split(v, hi, lo, ok);
  bn_acc_atomic_add(acc + ((size_t)replica * CS + c) * 4 + which * 2 + limb, limb ? lo : hi);
  if (!ok) bn_acc_atomic_add(acc + (size_t)BN_ACC_REPLICAS * CS * 4, 1);
}

// Consumer, in two phases so that a kernel can put its own first loads between them: `load` issues the requests of channel c
// (one memory round trip; vector-memory results return in order, so requests issued BEFORE the kernel's halo / tensor loads
// come back first), `add_to` folds them into the running integer sums.  RPT replicas per thread: 8 = the whole block; a
// workgroup with spare threads splits the replicas over 2 or 4 threads per channel (partial sums through LDS -- integers, any
// order) and holds 8 / 16 registers in flight instead of 64 (conv_fast.hip MODE 5: with 64 the prologue spilled, and a spill
// reload is a wait for EVERYTHING in flight).
typedef __attribute__((ext_vector_type(2))) long long i64x2;
struct BnAccSums {
  long long h1, l1, h2, l2, flag;
  __device__ __forceinline__ void totals(double& t1, double& t2) const {
    t1 = fma((double)l1, BN_ACC_LO_INV, (double)h1 * BN_ACC_HI_INV);
    t2 = fma((double)l2, BN_ACC_LO_INV, (double)h2 * BN_ACC_HI_INV);
    if (flag != 0) t1 = t2 = __builtin_nan("");
  }
};
template <int RPT>
struct BnAccPart {
  i64x2 a[RPT], b[RPT];
  long long flag;
  __device__ __forceinline__ void load(const long long* acc, int CS, int c, int r0) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const long long* row = acc + ((size_t)(r0 + r) * CS + c) * 4;
      a[r] = *(const i64x2*)row;
      b[r] = *(const i64x2*)(row + 2);
    }
    flag = acc[(size_t)BN_ACC_REPLICAS * CS * 4];
  }
  __device__ __forceinline__ BnAccSums sums() const {
    BnAccSums s{0, 0, 0, 0, flag};
#pragma unroll
    for (int r = 0; r < RPT; ++r) { s.h1 += a[r][0]; s.l1 += a[r][1]; s.h2 += b[r][0]; s.l2 += b[r][1]; }
    return s;
  }
};
typedef BnAccPart<BN_ACC_REPLICAS> BnAccRaw;

// forward coefficients of channel c from the totals (bn.hip bn_final_channel's arithmetic); `write`: also leave mean /
// invstd / scale / shift and update the running statistics (ONE workgroup of the consumer launch does)
struct BnAccFwdParams {  // the channel's parameters (clamped index: no divergent loads)
  float g, b, rm, rv;
  __device__ __forceinline__ void load(const BnAccFwd& f, int c) {
    const int cc = c < f.C ? c : f.C - 1;
    g = f.gamma[cc];
    b = f.beta[cc];
    rm = (f.running_mean != nullptr ? f.running_mean : f.gamma)[cc];
    rv = (f.running_var != nullptr ? f.running_var : f.gamma)[cc];
  }
};
__device__ __forceinline__ void bn_acc_fwd_channel(const BnAccFwd& f, const BnAccSums& sums, const BnAccFwdParams& in, int c,
                                                   float& scale, float& shift, bool write) {
  if (c >= f.C) {  // channel padding
    scale = 0.f;
    shift = 0.f;
    if (write) { f.st[c] = 0.f; f.st[f.CS + c] = 0.f; f.st[2 * f.CS + c] = 0.f; f.st[3 * f.CS + c] = 0.f; }
    return;
  }
  double s1, s2;
  sums.totals(s1, s2);
  const float g = in.g, b = in.b;
  const double n = (double)f.count;
  const double mu = s1 * f.inv_count;
  const double m2 = fmax(s2 - s1 * mu, 0.0);
  const double var = m2 * f.inv_count;
  float is = 1.0f / sqrtf((float)var + f.eps);
  float sc = g * is;
  scale = sc;
  shift = b - (float)mu * sc;
  if (sums.flag != 0) {  // a sum left the fixed-point range (or was a NaN): nothing derived from this block may look valid
    // (fmax above would have turned a NaN second moment into var = 0, i.e. a finite scale)
    is = sc = scale = shift = __builtin_nanf("");
  }
  if (write) {
    f.st[c] = (float)mu;
    f.st[f.CS + c] = is;
    f.st[2 * f.CS + c] = sc;
    f.st[3 * f.CS + c] = shift;
    if (f.running_mean != nullptr) f.running_mean[c] = (1.f - f.momentum) * in.rm + f.momentum * (float)mu;
    if (f.running_var != nullptr) {
      const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
      f.running_var[c] = (1.f - f.momentum) * in.rv + f.momentum * (float)unbiased;
    }
    if (f.nbt != nullptr && c == 0) f.nbt[0] += 1;
  }
}
// (one thread, the whole block: the streaming kernels' prologue)
struct BnAccFwdRaw {
  BnAccRaw raw;
  BnAccFwdParams prm;
  __device__ __forceinline__ void load(const BnAccFwd& f, int c) {
    raw.load(f.acc, f.CS, c < f.C ? c : f.C - 1, 0);
    prm.load(f, c);
  }
};
__device__ __forceinline__ void bn_acc_fwd_channel(const BnAccFwd& f, const BnAccFwdRaw& in, int c, float& scale, float& shift,
                                                   bool write) {
  bn_acc_fwd_channel(f, in.raw.sums(), in.prm, c, scale, shift, write);
}

// backward: folded coefficients of dy = scale dz + A y + B (bn.hip bnrelu_bwd_fin_kernel's arithmetic)
struct BnAccBwdRaw {
  BnAccRaw raw;
  float mean, invstd, scale;
  __device__ __forceinline__ void load(const BnAccBwd& f, int c) {
    raw.load(f.acc, f.CS, c, 0);
    mean = f.st[c];
    invstd = f.st[f.CS + c];
    scale = f.st[2 * f.CS + c];
  }
};
__device__ __forceinline__ void bn_acc_bwd_channel(const BnAccBwd& f, const BnAccBwdRaw& in, int c, float& A, float& B,
                                                   bool write) {
  double t1, t2;
  in.raw.sums().totals(t1, t2);
  const float mean = in.mean, invstd = in.invstd, scale = in.scale;
  const float s1 = (float)t1;
  const float s2 = (float)t2 * invstd;  // the block held sum dz (y - mean): dgamma = invstd * that
  A = 0.f;
  B = 0.f;
  if (f.training) {
    A = -scale * invstd * (s2 / f.count);
    B = -scale * (s1 / f.count) - A * mean;
  }
  if (write && c < f.C) {
    f.dbeta[c] = s1;
    f.dgamma[c] = s2;
  }
}
#endif  // __HIPCC__

}  // namespace spcl
