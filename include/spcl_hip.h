/* spcl_hip.h -- C ABI of libspcl_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * self-paced contrastive pre-train hot path.
 *
 * The reference (jizongFox/Self-paced-Contrastive-Learning) has NO FFI: its seam is four Python classes
 * dispatching stock ATen ops.  Each entry point below replaces the ATen op sequence of one reference
 * call site (cited per function as  path:line  relative to the reference root).  The Python mirror of the
 * reference classes (self-paced-contrastive-learning_amd/{contrastyou,semi_seg}) binds these through ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (spcl_last_error() gives the text); nothing throws
 *  - all data pointers are DEVICE pointers owned by the caller and only borrowed for the call
 *  - work is enqueued on the caller's stream (hipStream_t passed as void*); no host sync, no allocation,
 *    no internal threads -> every call is hipGraph-capturable
 *  - activations are NHWC ("channels last"), channel count padded to a multiple of 16 (pad lanes hold 0)
 *  - dtype: SPCL_F32 = 0 (f32 storage: the parity mode; convolution products are f32-grade split-bf16 by default, exact-f32
 *    MFMA on request -- spcl_conv_set_f32_split below), SPCL_BF16 = 1 (bf16 storage, f32 accumulate)
 *  - the library reads NOTHING from the environment: every switch is an argument or a setter declared here
 */
#ifndef SPCL_HIP_H
#define SPCL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPCL_F32 0
#define SPCL_BF16 1

#define SPCL_OK 0
#define SPCL_EINVAL (-1)
#define SPCL_ELAUNCH (-2)
#define SPCL_EUNSUPPORTED (-3)

/* Bumped whenever a struct layout, a limit (SPCL_*_MAX) or an entry point's meaning changes.  The library is git-ignored and
 * travels next to the sources: the Python binding (native.py) refuses a library whose version is not the header's, so that a
 * stale build fails at load time instead of running kernels on structs of another stride. */
#define SPCL_ABI_VERSION 9
int spcl_abi_version(void);
const char* spcl_last_error(void);

/* ---------------------------------------------------------------- contrastive loss -----------------------
 * Replaces contrastyou/losses/contrast_loss3.py:25-31 (exp_sim_temperature), :41-110 (SupConLoss1._forward),
 * :126-214 (SelfPacedSupConLoss._forward/_self_paced_mask) and the autograd backward of those (K11-K17).
 *
 * z1,z2   [n,d] f32 row-major (unit rows);  labels [n] f32 or NULL;  mask [n,n] f32 or NULL
 *         (labels==NULL && mask==NULL -> SimCLR identity positives, contrast_loss3.py:140-143)
 * sp_mode 0 = no self-pacing (SupConLoss1), 1 = hard, 2 = soft;  gamma = age parameter
 * ws      workspace of spcl_supcon_workspace_bytes(n,d) bytes (f32 aligned); holds the padded projections,
 *         the per-row statistics kept for backward and the column-split partials
 * Three schedules, chosen by size alone (same results within the parity tolerance, same workspace contract):
 *   2n <= 64 (the training sizes): one workgroup, one launch -- forward scalars, row statistics and dLoss/dP for a unit
 *     upstream gradient (kept in ws; spcl_supcon_backward then is one scaling launch); exact-f32 MFMA, same k order as the
 *     sweeps;
 *   64 < 2n < 1024, or an explicit `mask`: the [2n,2n] matrix is never materialised; every sweep recomputes its S tiles on the
 *     exact-f32 MFMA (bitwise an fmaf chain) -- the training sizes are launch-latency bound either way;
 *   2n >= 1024 (labels / SimCLR modes, d <= 128, 2n a multiple of 256): the forward is FUSED -- two sweeps over the
 *     similarity tiles (row sums first, then the weights that depend on them), each recomputing S on the bf16 matrix pipe from
 *     the two-term split P = Ph + Pm (S ~ Ph Ph^T + Ph Pm^T + Pm Ph^T, logit error ~1e-5); no logits matrix is written or
 *     read (SURVEY 8(d) prices this size against the materialised schedule's 205.5 MB; the fused forward's own HBM traffic is
 *     a few MB).  The backward forms H = G + G^T tile by tile the same way.  Other large shapes (d > 128, ragged 2n) keep
 *     the round-1 schedule: logits written once to ws as f32 [2n,2n], streamed by the later passes.
 * out     [8] f32: out[0]=loss out[1]=rho(downgrade ratio) out[2]=kappa(effective -dloss/drow scale)
 *                  out[3]=max|‖z‖-1| (is_normalized contract, contrast_loss3.py:20-22,154)
 */
size_t spcl_supcon_workspace_bytes(int n, int d);
int spcl_supcon_forward(const float* z1, const float* z2, const float* labels, const float* mask, int n, int d,
                        float temperature, int sp_mode, float gamma, int correct_grad, float* ws, float* out,
                        void* stream);
/* dz1,dz2 [n,d] f32 = grad_out[0] * dLoss/dz ; uses the row statistics left in ws by the forward call */
int spcl_supcon_backward(const float* labels, const float* mask, int n, int d, float temperature, int sp_mode,
                         float gamma, const float* ws_fwd, float* ws_bwd, const float* out_fwd,
                         const float* grad_out, float* dz1, float* dz2, void* stream);
size_t spcl_supcon_bwd_workspace_bytes(int n, int d);
/* 1 (and the block's offset in the FORWARD workspace, in floats, and its row pitch) when spcl_supcon_forward leaves dLoss/dP
 * for a unit upstream gradient there (the training sizes, 2n <= 64): rows [0, 2n) x [0, d) of it ARE the gradient when the
 * upstream gradient is exactly 1 -- spcl_supcon_backward would only multiply them by it.  0: this shape recomputes. */
int spcl_supcon_unit_gradient_block(int n, int d, size_t* offset_floats, int* row_pitch);
/* K (1..4) losses of ONE shape in the launches of one -- the K meta-label hooks that semi_seg/hooks/creator.py:102-124
 * puts on one feature, each with its own label vector and its own self-paced age parameter (hooks/infonce.py:133-141).
 * Head h reads z1 + h z_stride and z2 + h z_stride ([n,d] f32 each), labels + h n (NULL: SimCLR for every head), uses
 * gammas[h] (host array, read during the call), owns ws + h ws_stride floats (>= spcl_supcon_workspace_bytes(n,d) / 4)
 * and writes out + 8 h.  Same arithmetic, per head, as spcl_supcon_forward / _backward (bit-identical results).  Only the
 * training-size schedules (2n below the large-batch threshold; SPCL_EUNSUPPORTED otherwise) and no explicit mask.
 * backward: grad_out[K] (device), dz1 / dz2 + h z_stride, ws_bwd + h wsb_stride floats. */
int spcl_supcon_forward_heads(int K, const float* z1, const float* z2, size_t z_stride, const float* labels, int n,
                              int d, float temperature, int sp_mode, const float* gammas, int correct_grad, float* ws,
                              size_t ws_stride, float* out, void* stream);
/* The projector's F.normalize (contrastyou/projectors/heads.py:15-17, nn.py:29-36) and its backward inside the loss launch:
 * o1 / o2 are the K heads' rows BEFORE normalisation; the launch normalises them (z = o / max(||o||, 1e-12): the rows it
 * leaves in the workspace for the taps are the normalised ones, bit for bit those of spcl_l2norm_rows_forward), evaluates
 * the loss exactly as spcl_supcon_forward_heads does on them, and the unit-gradient block (spcl_supcon_unit_gradient_block;
 * spcl_supcon_backward(_heads) scales it) holds d loss / d o.  Training sizes only: spcl_supcon_rows_supported(n, d) != 0
 * (2n <= 64, d <= 256: the one-workgroup schedule); labels [K][n] or NULL, no explicit mask.  K = 1: strides unused. */
int spcl_supcon_rows_supported(int n, int d);
int spcl_supcon_forward_rows(int K, const float* o1, const float* o2, size_t o_stride, const float* labels, int n, int d,
                             float temperature, int sp_mode, const float* gammas, int correct_grad, float* ws,
                             size_t ws_stride, float* out, void* stream);
int spcl_supcon_backward_heads(int K, const float* labels, int n, int d, float temperature, int sp_mode,
                               const float* gammas, const float* ws_fwd, size_t ws_stride, float* ws_bwd,
                               size_t wsb_stride, const float* out_fwd, const float* grad_out, float* dz1, float* dz2,
                               size_t z_stride, void* stream);
/* lazily materialise the [2n,2n] hook taps (contrast_loss3.py:175-178,188): any pointer may be NULL */
int spcl_supcon_materialize(const float* labels, const float* mask, int n, int d, float temperature, int sp_mode,
                            float gamma, const float* ws_fwd, float* sim_logits, float* sim_exp, float* pos_mask,
                            float* neg_mask, float* sp_mask, void* stream);

/* SupConLoss1(exclude_other_pos=True) (contrast_loss3.py:97-100: every positive pair is scored against the row's
 * negatives only, their sum divided by the row's negative ratio + 1e-4) and its gradient; not used by the hooks, plain
 * fp32 row kernels.  labels / mask as above (both NULL = SimCLR); ws of spcl_supcon_xpos_workspace_bytes(n, d) (0 when
 * n > 4096) keeps d loss / d logits for the backward; out[0] = loss, out[3] = max | |row| - 1 |. */
size_t spcl_supcon_xpos_workspace_bytes(int n, int d);
int spcl_supcon_xpos_forward(const float* z1, const float* z2, const float* labels, const float* mask, int n, int d,
                             float temperature, float* ws, float* out, void* stream);
int spcl_supcon_xpos_backward(const float* z1, const float* z2, int n, int d, float temperature, const float* ws,
                              const float* grad_out, float* dz1, float* dz2, void* stream);

/* ---------------------------------------------------------------- projector ------------------------------
 * Replaces contrastyou/projectors/heads.py:9-25,78-92 + nn.py:8-15,29-36,56-58:
 * AdaptiveAvgPool2d((1,1)) -> Flatten -> Linear -> LeakyReLU(0.01) -> Linear -> F.normalize(p=2,dim=1).
 * feat [N,H,W,Cs] NHWC (dtype), first C of Cs channels used; w1 [hid,C] b1 [hid] w2 [out,hid] b2 [out] f32
 * (hid == 0 -> "linear" head: w1 is [out,C], w2/b2 ignored).  Saved for backward: pooled [N,C], pre [N,hid],
 * o [N,out] (un-normalised), all f32.  z [N,out] f32.
 * feat == NULL (spcl_proj_forward, spcl_proj_heads_forward): `pooled` is an INPUT -- the rows the producer of the feature
 * map left beside it (spcl_bnrelu_gap_forward) -- and no pooling launch runs; normalize == 0 with z == o: no copy.
 */
int spcl_proj_forward(const void* feat, int dtype, int N, int HW, int C, int Cs, const float* w1, const float* b1,
                      const float* w2, const float* b2, int hid, int out_dim, int normalize, float* pooled,
                      float* pre, float* o, float* z, void* stream);
/* dz [N,out] -> dw1,db1,dw2,db2 (f32, overwritten) and dfeat [N,H,W,Cs] (dtype, overwritten; NULL to skip) */
int spcl_proj_backward(const float* dz, int dtype, int N, int HW, int C, int Cs, const float* w1, const float* w2,
                       int hid, int out_dim, int normalize, const float* pooled, const float* pre, const float* o,
                       float* dw1, float* db1, float* dw2, float* db2, float* scratch /* [N,hid]+[N,out]+[N,C] f32 */,
                       void* dfeat, void* stream);

/* K <= 4 heads of IDENTICAL shape on the SAME feature -- several meta-label hooks on one encoder tap (hooks/creator.py:
 * 102-124 builds one INFONCEHook per contrast_on, each with its own ProjectionHead; semi_seg/hooks/infonce.py:224-231
 * projects the same tensor in each): the feature is average-pooled once and every layer of all K heads is ONE launch
 * (blockIdx.z = head); the backward sums the K input gradients in head order and broadcasts once.  The pointer
 * arguments marked [] are HOST arrays of K device pointers; the rest as spcl_proj_forward / spcl_proj_backward.
 * scratch: K * N * (out_dim + hid) + N * C floats. */
int spcl_proj_heads_forward(int K, const void* feat, int dtype, int N, int HW, int C, int Cs, const float* const* w1 /*[]*/,
                            const float* const* b1 /*[]*/, const float* const* w2 /*[]*/, const float* const* b2 /*[]*/,
                            int hid, int out_dim, int normalize, float* pooled, float* const* pre /*[]*/,
                            float* const* o /*[]*/, float* const* z /*[]*/, void* stream);
int spcl_proj_heads_backward(int K, const float* const* dz /*[]*/, int dtype, int N, int HW, int C, int Cs,
                             const float* const* w1 /*[]*/, const float* const* w2 /*[]*/, int hid, int out_dim,
                             int normalize, const float* pooled, const float* const* pre /*[]*/,
                             const float* const* o /*[]*/, float* const* dw1 /*[]*/, float* const* db1 /*[]*/,
                             float* const* dw2 /*[]*/, float* const* db2 /*[]*/, float* scratch, void* dfeat, void* stream);
/* the same with the feature gradient as ONE value per (image, channel): dfeat_nc [N][Cs] of dtype (Cs == C), = dpooled / HW,
 * what every pixel of that image and channel receives from the global average pool (heads.py:9-18).  The N x HW x Cs
 * broadcast is never written; spcl_bnrelu_backward_bcast consumes the [N][Cs] form. */
int spcl_proj_heads_backward_pooled(int K, const float* const* dz, int dtype, int N, int HW, int C, int Cs,
                                    const float* const* w1, const float* const* w2, int hid, int out_dim, int normalize,
                                    const float* pooled, const float* const* pre, const float* const* o, float* const* dw1,
                                    float* const* db1, float* const* dw2, float* const* db2, float* scratch, void* dfeat_nc,
                                    void* stream);

/* ---------------------------------------------------------------- encoder --------------------------------
 * Replaces semi_seg/arch/unet.py:67-82 (_ConvBlock: Conv2d 3x3 no bias -> BatchNorm2d -> ReLU, x2),
 * :118-121 (MaxPool2d 2x2), :156-190 (forward until Conv5) and their autograd backward (K1-K4, K17).
 */

/* weights: OIHW f32 master (state_dict layout) -> MFMA fragment-packed kernel layout (zero padded to 16-channel
 * multiples).  kind 0: forward  (K = (tap, ci), N = co);  kind 1: dgrad (K = (tap', co), N = ci, taps flipped
 * 180 degrees) -- the data gradient of a 3x3 same-conv is the same conv on these weights.
 * size in elements of `dtype`: spcl_conv_packed_elems(Cin, Cout, kind, dtype). */
size_t spcl_conv_packed_elems(int Cin, int Cout, int kind, int dtype);
int spcl_conv_pack_weights(const float* w_oihw, int Cin, int Cout, int kind, int dtype, void* packed, void* stream);
/* both layouts of one layer in one launch (forward keeps the dgrad copy for its backward) */
int spcl_conv_pack_weights_both(const float* w_oihw, int Cin, int Cout, int dtype, void* packed_fwd,
                                void* packed_dgrad, void* stream);

/* both convolutions of a Conv-BN-ReLU x2 block (forward + dgrad layouts each, sizes as spcl_conv_packed_elems) in one
 * launch, issued right before the block so that the fragments are still in L2 when its kernels fetch them */
int spcl_conv_pack_weights_block(const float* wa_oihw, int CinA, int CoutA, void* a_fwd, void* a_dgrad,
                                 const float* wb_oihw, int CinB, int CoutB, void* b_fwd, void* b_dgrad, int dtype,
                                 void* stream);
/* the same for weights that will be used at image size H x W: the band-GEMM layout of a dual-layout buffer is written only
 * where spcl_conv3x3_forward picks that kernel at this size (H = W = 0: both layouts, == spcl_conv_pack_weights_block) */
int spcl_conv_pack_weights_block_at(const float* wa_oihw, int CinA, int CoutA, void* a_fwd, void* a_dgrad,
                                    const float* wb_oihw, int CinB, int CoutB, void* b_fwd, void* b_dgrad, int dtype,
                                    int H, int W, void* stream);

/* forward + dgrad layouts of up to SPCL_PACK_MULTI_MAX layers in ONE launch (every conv weight of the UNet encoder at the
 * start of its forward pass: unet.py:123-131 are ten nn.Conv2d; one launch instead of one per block).  Per layer: the
 * OIHW master, both destination buffers (sizes as spcl_conv_packed_elems) and the image size it will be used at (as
 * spcl_conv_pack_weights_block_at: 0, 0 = both layouts of a dual-layout buffer).  items is host memory, read during the
 * call. */
#define SPCL_PACK_MULTI_MAX 24
typedef struct spcl_pack_item {
  const float* w_oihw;
  void* fwd;
  void* dgrad;
  int Cin, Cout, H, W;
} spcl_pack_item;
int spcl_conv_pack_weights_multi(const spcl_pack_item* items, int n, int dtype, void* stream);
/* the same launch also runs spcl_image_autocorr(image, N, H, W, acorr) (see "image3" below) in workgroups of its own: two
 * short, independent jobs at the start of a forward pass overlap and pay one launch.  image == NULL: the plain pack. */
int spcl_conv_pack_weights_multi_acorr(const spcl_pack_item* items, int n, int dtype, const float* image, int N, int H,
                                       int W, float* acorr, void* stream);
/* ... and zeroes zero_bytes bytes at `zero` in the same launch (the step's BatchNorm accumulator blocks, see spcl_bn_acc below:
 * they must be zero before the first convolution of the pass, and this is the pass's first launch).  zero == NULL: none. */
int spcl_conv_pack_weights_multi_zero(const spcl_pack_item* items, int n, int dtype, const float* image, int N, int H, int W,
                                      float* acorr, void* zero, size_t zero_bytes, void* stream);

/* y = conv3x3(act(x)), NHWC, implicit GEMM on MFMA.                    (unet.py:72,75; dgrad: with kind-1 weights)
 * x [N,H,W,CinS] of dtype; CinK = GEMM-K channels (multiple of 16, == CinS for in_mode 0/1).
 * in_mode 0: act = identity;  1: act = relu(in_scale[c]*x+in_shift[c])  (the producer's BatchNorm-apply + ReLU,
 * unet.py:73-74, fused into the load);  2: x is the f32 input image [N,H,W,CinS] with CinS<=16 real channels
 * (== NCHW when CinS==1), zero-padded to CinK=16 on load.
 * y [N,H,W,CoutS] raw conv output (dtype).  stats != NULL: per-tile Chan partials (count, mean, M2) per output
 * channel, one row stats[tile][3][CoutS] f32 per pixel tile (tile < spcl_conv_num_tiles(N,H,W)), for the train-mode
 * BatchNorm that follows; the buffer holds spcl_bn_stats_elems(ntiles, CoutS) floats (tile rows + the finalize
 * kernel's partial rows). */
int spcl_conv_num_tiles(int N, int H, int W);
/* Statistics rows the convolution of this configuration writes (stats [rows][3][CoutS], rows2 [rows][2][CoutS]): the
 * pixel tiles above, except where the workgroup-level GEMM kernel runs (csrc/conv_gemm.hip: bf16, channel counts
 * multiples of 64 with one side >= 128, see spcl_conv_set_gemm), which writes one row per (image band, pixel part). */
int spcl_conv_stat_rows(int dtype, int N, int H, int W, int CinK, int CoutS);
/* Which kernel takes those layers: -1 (default) the GEMM kernel only at image sizes the per-wave kernels have no
 * specialisation for (widths that do not tile by 14 columns: the 32^2 / 16^2 levels of 256^2 inputs), 1 wherever it fits,
 * 0 never.  Packed weights are valid under every setting. */
void spcl_conv_set_gemm(int mode);
/* How the f32-storage convolutions (forward, data gradient, weight gradient: nn.Conv2d of unet.py:72,75 run in torch's default
 * float32, the north_star's "within fp32 tolerance" path) multiply: 1 (default) every f32 operand as the exact sum of three
 * bf16 pieces, six v_mfma_f32_16x16x32_bf16 products per pair accumulated in f32 (the dropped terms are below 2^-24 of a
 * product: f32-grade results at 3/8 of the exact path's matrix time); 0 v_mfma_f32_16x16x4_f32 (exact f32 products, 1/16 of
 * the bf16 matrix rate).  Packed f32 weights carry both layouts: the switch may change between any two launches.  One
 * process-wide setting, read at launch time on the calling thread (not synchronised: set it before the threads that launch
 * start, as the tests and bench.py --f32-exact do).  The one-channel image convolution (in_mode 2, CoutS == 16) is a direct
 * exact-FMA kernel under both settings. */
void spcl_conv_set_f32_split(int on);
int spcl_conv_get_f32_split(void);
size_t spcl_bn_stats_elems(int ntiles, int CS);
int spcl_conv3x3_forward(const void* x, int dtype, int N, int H, int W, int CinS, int CinK, int CoutS,
                         const void* w_packed, int in_mode, const float* in_scale, const float* in_shift, void* y,
                         float* stats, void* stream);
/* The first convolution of the image block (semi_seg/arch/unet.py:123 behind :67-71, nn.Conv2d(1, 16, 3, 1, 1, bias=False)
 * on the f32 slice, in_mode 2 of spcl_conv3x3_forward) that ALSO leaves the image's autocorrelation partial rows -- what
 * spcl_image_autocorr / spcl_conv_pack_weights_multi_acorr compute in a pass of their own for the "image3" backward of that
 * convolution (spcl_bnrelu_backward_rows_image3): acorr_rows [rows][64] f32, one row per 14 x 14 tile, same layout and
 * meaning per row (45 upper-triangle sums R[t'][t] + 9 image sums over the tile's pixels), so spcl_conv16_bwd_fused's
 * fold and the image3 final kernels take them unchanged.  _rows: the number of rows, 0 where the specialised kernel does
 * not take the shape (bf16, CinS == 1, CoutS == 16, H and W multiples of 14 above 112). */
int spcl_conv3x3_forward_image_acorr_rows(int dtype, int N, int H, int W, int CinS, int CoutS);
int spcl_conv3x3_forward_image_acorr(const void* x, int dtype, int N, int H, int W, int CinS, int CoutS,
                                     const void* w_packed, void* y, float* stats, float* acorr_rows, void* stream);
/* The decoder's torch.cat((skip, up), dim=1) -> Conv2d(2 Chalf, Cout, 3, 1, 1) (unet.py:194-197, 201-204, ... the first
 * convolution of every Up_conv block) WITHOUT the concatenated tensor: input channels [0, Chalf) are read from xa, [Chalf,
 * 2 Chalf) from xb, both dense [N][H][W][Chalf] bf16; w_packed = spcl_conv_pack_weights(kind 0) of the [Cout][2 Chalf]
 * weight, y / stats as spcl_conv3x3_forward.  Only where a specialised kernel exists (Chalf 16 or 32, 14-column tiles,
 * not a layer whose weight gradient the batched GEMM takes): spcl_conv_cat_supported answers; spcl_conv3x3_wgrad_cat is
 * the matching weight gradient (workspace: spcl_conv_wgrad_workspace_bytes with CinK = 2 Chalf).  The input gradient is the
 * ordinary dgrad: one [N][H][W][2 Chalf] tensor whose channel halves the two producers' backward passes read in place
 * (spcl_bnrelu_pool_backward_strided). */
int spcl_conv_cat_supported(int dtype, int N, int H, int W, int Chalf, int CoutS);
int spcl_conv3x3_forward_cat(const void* xa, const void* xb, int dtype, int N, int H, int W, int Chalf, int CoutS,
                             const void* w_packed, const float* xb_scale, const float* xb_shift, void* y, float* stats,
                             void* stream);
/* xb_scale / xb_shift [Chalf] (both, or both NULL; Chalf 16 / 32): xb is the RAW output of the up-convolution (unet.py:90) and
 * the concatenation holds relu(xb_scale xb + xb_shift) -- that BatchNorm + ReLU (unet.py:91-92) applied while the halo is
 * staged, so the up-convolution's activation is never written either. */
int spcl_conv3x3_wgrad_cat(const void* xa, const void* xb, const void* dy, int dtype, int N, int H, int W, int Chalf,
                           int Cout, int CoutS, const float* xb_scale, const float* xb_shift, float* partial,
                           float* dw_oihw, void* stream);
/* nn.Upsample(scale_factor=2) -> Conv2d of the up-convolution (unet.py:89-90) without the upsampled tensor: x_half is the
 * half-resolution activation [N][H / 2][W / 2][CinK], H x W the convolution's size; the loaders turn a fine halo pixel
 * (gy, gx) into pixel (gy >> 1, gx >> 1) of x_half.  Forward (y / stats as spcl_conv3x3_forward; where
 * spcl_conv_up2_supported says so) and weight gradient (workspace: spcl_conv_wgrad_workspace_bytes; the batched kernel
 * takes it through spcl_wgrad_item::x_up2).  The input gradient is the ordinary dgrad followed by the 2 x 2 sum
 * (spcl_upsample2x_backward / spcl_bnrelu_backward_up2). */
int spcl_conv_up2_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_forward_up2(const void* x_half, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                             void* y, float* stats, void* stream);
int spcl_conv3x3_wgrad_up2(const void* x_half, const void* dy, int dtype, int N, int H, int W, int Cin, int Cout, int CoutS,
                           float* partial, float* dw_oihw, void* stream);
/* ... and the input gradient of that convolution as the gradients of the two concatenated tensors: the plain 3x3 convolution
 * (w_packed = the dgrad layout, kind 1) whose output channels [0, CoutS / 2) are written to y_lo and [CoutS / 2, CoutS) to
 * y_hi, both dense [N][H][W][CoutS / 2] bf16 (torch.cat's backward, unet.py:194-224, without the interleaved tensor).
 * Where spcl_conv_split_supported says so (a specialised kernel, CoutS a multiple of 32). */
int spcl_conv_split_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_forward_split(const void* x, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                               void* y_lo, void* y_hi, void* stream);
/* ... the same with the BatchNorm-backward partial sums of the layer behind the UPPER half -- the up-convolution (unet.py:90-92):
 * y2_hi [N][H][W][CoutS / 2] its raw output, scale2 / shift2 / mean2 [CoutS / 2] its coefficients; rows2
 * [spcl_conv_stat_rows(dtype, N, H, W, CinK, CoutS)][2][CoutS / 2] (sum dz, sum dz (y2 - mean)) as spcl_conv3x3_dgrad_bnstats
 * leaves them, for spcl_bnrelu_backward_rows: that layer's reduction pass over (y2, g_hi) disappears. */
int spcl_conv_split_bnstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_dgrad_split_bnstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                     const void* w_packed, void* g_lo, void* g_hi, const void* y2_hi, const float* scale2,
                                     const float* shift2, const float* mean2, float* rows2, void* stream);

/* dW[co][ci][ky][kx] (OIHW f32, overwritten) = sum_pixels act(x)[p+tap][ci] * dy[p][co]   (weight gradient of
 * unet.py:72,75).  x / in_mode / CinK as in forward; Cin, Cout = real channel counts of dW.
 * partial: workspace of spcl_conv_wgrad_workspace_bytes() (deterministic two-stage reduction, no atomics). */
size_t spcl_conv_wgrad_workspace_bytes(int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_wgrad(const void* x, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS, int CinK,
                       int Cout, int CoutS, int in_mode, const float* in_scale, const float* in_shift, float* partial,
                       float* dw_oihw, void* stream);

/* nn.AdaptiveAvgPool2d / nn.AdaptiveMaxPool2d((OH, OW)) on an NHWC tensor x [N,H,W,Cs] of dtype (C real channels) ->
 * out [N,OH,OW,C] f32 (contrastyou/projectors/nn.py:56-64: the pooling of ProjectionHead(pool_name="adaptive_max") and of
 * DenseProjectionHead, heads.py:96-120).  mode 0 average, 1 maximum (argmax [N,OH,OW,C] int32 receives the flat y * W + x of
 * the first maximum).  Backward: dx [N,H,W,Cs] of dtype, deterministic (one thread per input element, no atomics). */
int spcl_adaptive_pool2d_forward(const void* x, int dtype, int N, int H, int W, int C, int Cs, int OH, int OW, int mode,
                                 float* out, int* argmax, void* stream);
int spcl_adaptive_pool2d_backward(const float* dout, const int* argmax, int dtype, int N, int H, int W, int C, int Cs,
                                  int OH, int OW, int mode, void* dx, void* stream);

/* F.normalize(p=2, dim=-1, eps=1e-12) of `rows` rows of O floats (projectors/nn.py:29-36 on a pixel-major map) and its
 * backward. */
int spcl_l2norm_rows_forward(const float* x, size_t rows, int O, float* z, void* stream);
int spcl_l2norm_rows_backward(const float* x, const float* dz, size_t rows, int O, float* dx, void* stream);

/* The same weight gradients for SEVERAL layers in ONE launch (bf16, channel counts multiples of 64: _Conv3.b.._Conv5.b
 * and the decoder's wide layers).  The workgroups split the union of the layers' pixel ranges, so the launch writes
 * one split-K partial per CU in total instead of one per CU and layer (csrc/wgrad_gemm.hip).  x / dy / in_mode (0 or
 * 1) / in_scale / in_shift as above with CinK == Cin; dw_oihw [Cout][Cin][3][3] f32 is overwritten, or added to when
 * accumulate != 0 (a gradient bucket that was zeroed before backward and may already hold another use's gradient).
 * partial: spcl_conv_wgrad_batched_workspace_bytes(items, n) bytes.  items is host memory, read during the call. */
#define SPCL_WGRAD_BATCH_MAX 16
typedef struct spcl_wgrad_item {
  const void* x;
  const void* dy;
  const float* in_scale;
  const float* in_shift;
  float* dw_oihw;
  int N, H, W, Cin, CinS, Cout, CoutS, in_mode;
  const void* x2; /* NULL, or: input channels [Cin / 2, Cin) come from this tensor and [0, Cin / 2) from x, both dense
                     [N][H][W][Cin / 2] (the decoder's concatenation read in place; Cin a multiple of 128, in_mode 0) */
  int x_up2;      /* 1: x is [N][H / 2][W / 2][CinS] and the layer's input its nearest x2 upsample (unet.py:89), in_mode 0 */
} spcl_wgrad_item;
int spcl_conv_wgrad_batched_supported(int dtype, int Cin, int CinS, int Cout, int CoutS, int in_mode);
size_t spcl_conv_wgrad_batched_workspace_bytes(const spcl_wgrad_item* items, int n);
int spcl_conv3x3_wgrad_batched(const spcl_wgrad_item* items, int n, int accumulate, float* partial, void* stream);

/* Deferred final sums ("tails") of the weight gradients that do NOT go through the batched GEMM kernel: the narrow
 * layers' split partials of spcl_conv3x3_wgrad and the first layer's per-workgroup rows of the fused BN-backward +
 * weight-gradient pass (spcl_bnrelu_backward_image_wgrad / spcl_bnrelu_backward_rows with an image).  Each of those
 * ends in its own few-microsecond reduction launch; with a capture armed the producer skips it and describes the
 * pending sum instead, and spcl_conv3x3_wgrad_batched_tails finishes up to SPCL_WGRAD_TAILS_MAX of them in extra
 * workgroups of the batched launch's reduction kernel (fixed summation order; dw written, or added to when
 * accumulate != 0).  The producer's `partial` / `ws` buffer must stay alive until that launch has run.
 *   spcl_wgrad_tail_capture(slot): one-shot, per calling thread -- the NEXT call of one of the producers above fills
 *   *slot and leaves dw untouched (slot == NULL disarms).  A producer that takes the batched path itself (bf16, channel
 *   counts multiples of 64) ignores the capture and clears it; slot->kind stays -1 then.
 * unet.py:72,75 weight gradients; replaces nothing new in the reference, only removes launches. */
#define SPCL_WGRAD_TAILS_MAX 16
typedef struct spcl_wgrad_tail {
  const float* partial; /* kind 0: [nsplit][nblk_ci*nblk_co][9*CIB*COB]; kind 1: [nsplit][9][COB] */
  float* dw;            /* OIHW destination */
  int kind;             /* -1 empty, 0 conv split partials, 1 first-layer rows */
  int nsplit, nblk_ci, nblk_co, CIB, COB, Cin, Cout;
} spcl_wgrad_tail;
int spcl_wgrad_tail_capture(spcl_wgrad_tail* slot);
/* n may be 0 (tails only; partial may then be NULL), ntails may be 0 (== spcl_conv3x3_wgrad_batched) */
int spcl_conv3x3_wgrad_batched_tails(const spcl_wgrad_item* items, int n, const spcl_wgrad_tail* tails, int ntails,
                                     int accumulate, float* partial, void* stream);
/* train-mode BatchNorm statistics (unet.py:73,76; torch.nn.BatchNorm2d semantics): combines the conv epilogue
 * partials stats[ntiles][3][CS] (Chan, fixed order, in double; the tail of the buffer is scratch) -> mean, invstd = 1/sqrt(var_biased+eps), scale = gamma*invstd,
 * shift = beta-mean*scale (all [CS] f32, zero in the channel padding) and updates running_mean / running_var
 * (unbiased variance, momentum) and num_batches_tracked (+1) when those pointers are non-NULL. */
int spcl_bn_finalize(float* stats, int ntiles, int C, int CS, const float* gamma, const float* beta,
                     float momentum, float eps, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                     void* stream);
/* eval mode: mean/invstd/scale/shift from the running statistics */
int spcl_bn_eval_affine(int C, int CS, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* mean, float* invstd, float* scale, float* shift,
                        void* stream);
/* ... of up to SPCL_BN_EVAL_MAX BatchNorms in one launch (the eval-mode forward of the full UNet: 22 layers); st: [4][CS] =
 * mean, invstd, scale, shift of that layer, as spcl_bn_eval_affine writes them. */
#define SPCL_BN_EVAL_MAX 32
typedef struct {
  const float* gamma;
  const float* beta;
  const float* running_mean;
  const float* running_var;
  float* st;
  int C, CS;
  float eps;
} spcl_bn_eval_item;
int spcl_bn_eval_affine_multi(const spcl_bn_eval_item* items, int n, void* stream);

/* a = relu(scale*y+shift) [N,H,W,CS] (act_out, may be NULL) and/or its 2x2/2 max-pool p [N,H/2,W/2,CS]
 * (pool_out, may be NULL)                                                       -- unet.py:74,77 + :118-121 */
int spcl_bnrelu_pool_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                             const float* shift, void* act_out, void* pool_out, void* stream);
/* backward of the above through ReLU, (optional) max-pool and BatchNorm:
 *   g  = dact (+ dpool routed to the window arg-max: first maximum in scan order, as torch.max_pool2d)
 *   dz = g * [scale*y+shift > 0];  dbeta = sum dz;  dgamma = sum dz*yhat,  yhat = (y-mean)*invstd
 *   training: dy = scale*(dz - dbeta/M - yhat*dgamma/M);   eval: dy = scale*dz
 * dact / dpool may be NULL (not both).  Deterministic: per-workgroup partials + fixed-order second stage. */
size_t spcl_bnrelu_bwd_workspace_bytes(int N, int H, int W, int CS);
int spcl_bnrelu_pool_backward(const void* y, const void* dact, const void* dpool, int dtype, int N, int H, int W,
                              int C, int CS, const float* mean, const float* invstd, const float* scale,
                              const float* shift, int training, float* ws, float* dgamma, float* dbeta, void* dy,
                              void* stream);
/* The same two calls with the activation written / its gradient read as a CHANNEL SLICE of a wider NHWC tensor
 * (act_stride / dact_stride = elements between two pixels, >= CS, a multiple of 8): the decoder's
 * `torch.cat((skip, up), dim=1)` (semi_seg/arch/unet.py:194-224) then needs no copy in either direction -- both producers
 * write their half of the concatenated tensor, both consumers read their half of its gradient (round 4, row N1). */
/* BN-apply + ReLU written 2x2-REPLICATED: up_out [N][2H][2W][CS] = nearest-upsample(relu(scale y + shift)) -- the activation of
 * a block whose only consumer is the decoder's nn.Upsample(scale_factor=2) (semi_seg/arch/unet.py:89): neither the
 * low-resolution activation nor a separate upsampling pass is written (round 4, row N1). */
int spcl_bnrelu_up2_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale, const float* shift,
                            void* up_out, void* stream);
int spcl_bnrelu_pool_forward_strided(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                                     const float* shift, void* act_out, int act_stride, void* pool_out, void* stream);
int spcl_bnrelu_pool_backward_strided(const void* y, const void* dact, int dact_stride, const void* dpool, int dtype, int N,
                                      int H, int W, int C, int CS, const float* mean, const float* invstd,
                                      const float* scale, const float* shift, int training, float* ws, float* dgamma,
                                      float* dbeta, void* dy, void* stream);
/* BN + ReLU backward of a block whose activation went through nn.Upsample(scale_factor=2) (unet.py:89) before its consumer:
 * d_up = the gradient w.r.t. the UPSAMPLED activation [N][2H][2W][CS]; the 2 x 2 sums (spcl_upsample2x_backward) are formed
 * inside the BatchNorm-backward reduction pass and left in dact [N][H][W][CS] (scratch, written) for the apply pass.  Results as
 * spcl_upsample2x_backward followed by spcl_bnrelu_pool_backward(y, dact, NULL, ...), bit for bit; one pass over the fine
 * gradient and one launch less. */
int spcl_bnrelu_backward_up2(const void* y, const void* d_up, void* dact, int dtype, int N, int H, int W, int C, int CS,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int training,
                             float* ws, float* dgamma, float* dbeta, void* dy, void* stream);
/* BN + ReLU backward (no pooling) for a gradient that is the same for every pixel of an image: dact_nc [N][CS] of dtype.
 * Same arithmetic as spcl_bnrelu_pool_backward(dact = the expanded tensor, dpool = NULL). */
int spcl_bnrelu_backward_bcast(const void* y, const void* dact_nc, int dtype, int N, int H, int W, int C, int CS,
                               const float* mean, const float* invstd, const float* scale, const float* shift, int training,
                               float* ws, float* dgamma, float* dbeta, void* dy, void* stream);

/* BatchNorm+ReLU backward of the FIRST conv of a one-channel image block, fused with that conv's weight gradient
 * (`_Conv1.conv.0`/`.1`/`.2` backward, semi_seg/arch/unet.py:70-72 with input_dim = 1, + autograd).  The input image
 * needs no gradient, so this layer's dy feeds nothing but dW: it is formed in registers and never written.
 *   y, dact   [N][H][W][CS] of dtype (raw conv output, gradient of the activation);  image [N][H][W] f32
 *   dw        [C][1][3][3] f32;  dgamma, dbeta [C];  ws of spcl_bnrelu_image_wgrad_workspace_bytes
 * CS a power of two in 16..256.  Deterministic (fixed workgroup partition, fixed-order sums). */
size_t spcl_bnrelu_image_wgrad_workspace_bytes(int N, int H, int W, int CS);
int spcl_bnrelu_backward_image_wgrad(const void* y, const void* dact, const float* image, int dtype, int N, int H,
                                     int W, int C, int CS, const float* mean, const float* invstd, const float* scale,
                                     const float* shift, int training, float* ws, float* dgamma, float* dbeta,
                                     float* dw, void* stream);

/* dgrad of a block's SECOND conv fused with the per-tile partial sums of the FIRST conv's BatchNorm backward
 * (`_ConvBlock.conv.3` input gradient -> `.conv.1/.2` backward, semi_seg/arch/unet.py:70-77 + autograd): the dgrad's
 * output g is d loss / d relu(bn(y2)), so its epilogue already holds what the reduction pass over (y2, g) would read.
 *   dy [N][H][W][CinK], g [N][H][W][CoutS], y2 [N][H][W][CoutS] (raw output of the first conv), all bf16;
 *   rows2 [spcl_conv_num_tiles(N,H,W)][2][CoutS] f32: sum dz and sum dz (y2 - mean) per conv tile, dz = g [bn(y2) > 0].
 * Exists only where a specialised conv kernel does: ask spcl_conv_dgrad_bnstats_supported (1 / 0) first, else run
 * spcl_conv3x3_forward + spcl_bnrelu_pool_backward.  spcl_bnrelu_backward_rows finishes from the rows: dgamma, dbeta and
 * either dy (image == dw == NULL) or, for a one-channel image block, that layer's dW (dy == NULL), as the two entry
 * points above do after their own reduction pass.  Same workspace sizes as those. */
int spcl_conv_dgrad_bnstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_dgrad_bnstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                               void* g, const void* y2, const float* scale2, const float* shift2, const float* mean2,
                               float* rows2, void* stream);

/* ---- image3: the first conv of a block whose input is a ONE-CHANNEL f32 image (unet.py:123 with input_dim == 1, the
 * configuration every driver of the reference uses).  Its BatchNorm backward and weight gradient need no pass over the
 * activations: with dy = scale dz + A y + B (the folded BN backward), dW[c][t] = scale[c] sum_p dz[p][c] img[p + t] +
 * A[c] (W R)[c][t] + B[c] sum_p img[p + t], R = the 9 x 9 autocorrelation of the zero-padded image batch (y is linear in
 * the image).  Replaces spcl_bnrelu_backward_rows(image != NULL) + the dgrad before it where supported (bf16, 16 -> 16
 * channels, sizes tiled 14 x 14):
 *   spcl_image_autocorr          image [N][H][W] f32 (W <= 256) -> out [spcl_image_autocorr_rows(N, H, W)][64] partial rows
 *   spcl_conv3x3_dgrad_bnstats_image   spcl_conv3x3_dgrad_bnstats + rows 2 .. 10 of rows11 [tiles][11][CoutS] = the nine
 *                                sums sum_p dz[p][co] image[p + tap] (tiles = spcl_conv_stat_rows); g == NULL: the
 *                                gradient tensor is not written (image3 needs only the rows)
 *   spcl_bnrelu_backward_rows_image3   rows11 + autocorrelation + the f32 master weights [C][1][3][3] -> dgamma, dbeta, dW
 *                                (ws: spcl_bnrelu_image3_workspace_bytes(CS) bytes) */
int spcl_image_autocorr_rows(int N, int H, int W);
int spcl_image_autocorr(const float* image, int N, int H, int W, float* out, void* stream);
int spcl_conv_dgrad_bnstats_image_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_dgrad_bnstats_image(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS,
                                     const void* w_packed, void* g, const void* y2, const float* scale2,
                                     const float* shift2, const float* mean2, const float* image, float* rows11,
                                     void* stream);
/* The image block's SECOND conv (unet.py:75 at 16 -> 16 channels), its whole backward in one pass over dy and y2: the
 * weight gradient (what spcl_conv3x3_wgrad(x = y2, in_mode 1, scale2, shift2) computes; partial = the split slabs,
 * spcl_conv16_bwd_fused_splits(N, H, W) x 9 x 256 floats; the final sum goes to dw_oihw [Cout][Cin][3][3] or to a captured
 * tail, as there) AND the rows11 of spcl_conv3x3_dgrad_bnstats_image(g == NULL).  Nothing is written per pixel.
 * Same gate as spcl_conv_dgrad_bnstats_image_supported.
 * Exactly one of rows11 ([tiles][11][CoutS], as above) / wg_rows: ONE row set per workgroup (its tiles summed in
 * registers), [11][16][spcl_conv16_bwd_fused_splits] -- the layout spcl_bnrelu_backward_wgrows_image3 finishes from without
 * a folding launch; with wg_rows the same launch folds the autocorrelation's partial rows acorr [nacorr][64] to
 * acorr16 [16][64]. */
int spcl_conv16_bwd_fused_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv16_bwd_fused_splits(int N, int H, int W);
int spcl_conv16_bwd_fused(const void* dy, int dtype, int N, int H, int W, const void* w_packed_dgrad, const void* y2,
                          const float* scale2, const float* shift2, const float* mean2, const float* image,
                          float* rows11, float* partial, float* dw_oihw, int Cin, int Cout, float* wg_rows,
                          const float* acorr, int nacorr, float* acorr16, void* stream);
int spcl_bnrelu_backward_wgrows_image3(const float* wg_rows, int nwg, const float* acorr, int nacorr, const float* w_oihw,
                                       int N, int H, int W, int C, int CS, const float* mean, const float* invstd,
                                       const float* scale, int training, float* ws, float* dgamma, float* dbeta,
                                       float* dw, void* stream);
size_t spcl_bnrelu_image3_workspace_bytes(int CS);
int spcl_bnrelu_backward_rows_image3(const float* rows11, int nrows, const float* acorr, int nacorr, const float* w_oihw,
                                     int N, int H, int W, int C, int CS, const float* mean, const float* invstd,
                                     const float* scale, int training, float* ws, float* dgamma, float* dbeta, float* dw,
                                     void* stream);

/* The same across a POOLED block boundary: dy = gradient of the next block's first conv output (H x W), g = its input
 * gradient = d loss / d maxpool2x2(relu(bn(y2))), y2 [N][H2][W2][CoutS] the raw second-conv output of the block before
 * (H == H2 / 2, W == W2 / 2; semi_seg/arch/unet.py:118-121,159-166).  rows2 [spcl_conv_stat_rows(...)][2][CoutS]: per conv
 * tile, sum dz and sum dz (y2 - mean) with dz = g routed to the window's first positive maximum -- what
 * spcl_bnrelu_pool_backward's reduction pass computes, without that pass.  bf16, per-wave kernels only: ask
 * spcl_conv_dgrad_poolstats_supported first; finish with spcl_bnrelu_pool_backward_rows. */
int spcl_conv_dgrad_poolstats_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int H2, int W2);
int spcl_conv3x3_dgrad_poolstats(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                                 void* g, const void* y2, int H2, int W2, const float* scale2, const float* shift2,
                                 const float* mean2, float* rows2, void* stream);
int spcl_bnrelu_pool_backward_rows(const void* y, const void* dpool, const float* rows, int nrows, int dtype, int N, int H,
                                   int W, int C, int CS, const float* mean, const float* invstd, const float* scale,
                                   const float* shift, int training, float* ws, float* dgamma, float* dbeta, void* dy,
                                   void* stream);
int spcl_bnrelu_backward_rows(const void* y, const void* dact, const float* image, const float* rows, int nrows, int dtype,
                              int N, int H, int W, int C, int CS, const float* mean, const float* invstd,
                              const float* scale, const float* shift, int training, float* ws, float* dgamma,
                              float* dbeta, void* dy, float* dw, void* stream);


/* ---------------------------------------------------------------------------------------------------------------
 * Segmentation head and fine-tune / evaluation arithmetic.  Activations [npix][CS] of dtype (NHWC,
 * npix = N*H*W); class maps [npix][K] f32 with K <= 16; labels [npix] int64.
 *   spcl_conv1x1_forward/backward   nn.Conv2d(C, K, 1) with bias = `_Deconv_1x1`, semi_seg/arch/unet.py:147,229 (+ autograd):
 *                                   out = x W^T + b;  dx (dtype, padding channels zeroed), dW [K][C], db [K]
 *   spcl_softmax_forward/backward   `logits.softmax(1)` of semi_seg/epochers/new_epocher.py:86,271 (+ autograd)
 *   spcl_kl_div_forward/backward    deepclustering2.loss.KL_div(reduction="mean") as called there (un-vendored; restated):
 *                                   loss = mean_p sum_k -t log((p+eps)/(t+eps));  dprob = grad_loss * (-t/(p+eps))/npix
 *   spcl_one_hot                    class2one_hot (new_epocher.py:84,270) into [npix][K] f32
 *   spcl_argmax_classes             `.max(1)[1]` (new_epocher.py:89,282): first maximum
 *   spcl_dice_counts                UniversalDice._intersaction / ._union on class-coded maps
 *                                   (contrastyou/meters/general_dice_meter.py:131-160): inter/union [B][C] int64, which
 *                                   the caller zeroes (integer atomics: exact and order-independent) */
int spcl_conv1x1_forward(const void* x, int dtype, size_t npix, int C, int CS, int K, const float* w, const float* b,
                         float* out, void* stream);
size_t spcl_conv1x1_bwd_workspace_bytes(int C, int K);
int spcl_conv1x1_backward(const void* x, const float* dout, int dtype, size_t npix, int C, int CS, int K, const float* w,
                          void* dx, float* dw, float* db, float* ws, void* stream);
/* The decoder's LAST BatchNorm + ReLU (unet.py:79-81 of Up_conv2) folded into the 1x1 convolution that is its only
 * consumer (unet.py:229), on both sides: y is the raw output of the last 3x3 convolution ([npix][CS] of dtype), the class map
 * is conv1x1(relu(scale y + shift)) with the activation rounded to dtype exactly as the writer would have stored it -- no
 * activation tensor.  Backward: dact = the gradient w.r.t. that activation, dw / db as spcl_conv1x1_backward, and rows
 * [spcl_conv1x1_bwd_rows(npix)][2][CS] = the partial sums of that BatchNorm's backward (sum dz, sum dz (y - mean)) in the
 * form spcl_bnrelu_backward_rows finishes: its reduction pass over (y, dact) disappears too. */
int spcl_conv1x1_forward_bn(const void* y, int dtype, size_t npix, int C, int CS, int K, const float* scale,
                            const float* shift, const float* w, const float* b, float* out, void* stream);
int spcl_conv1x1_bwd_rows(size_t npix);
int spcl_conv1x1_backward_bn(const void* y, const float* dout, int dtype, size_t npix, int C, int CS, int K,
                             const float* scale, const float* shift, const float* mean, const float* w, void* dact,
                             float* dw, float* db, float* ws, float* rows, void* stream);
int spcl_softmax_forward(const float* logits, size_t npix, int K, float* prob, void* stream);
int spcl_softmax_backward(const float* prob, const float* dprob, size_t npix, int K, float* dlogits, void* stream);
size_t spcl_kl_workspace_bytes(void);
int spcl_kl_div_forward(const float* prob, const float* target, size_t npix, int K, float eps, float* ws, float* loss,
                        void* stream);
int spcl_kl_div_backward(const float* prob, const float* target, size_t npix, int K, float eps, const float* grad_loss,
                         float* dprob, void* stream);
/* The supervised criterion of the fine-tune loop in one pass (semi_seg/epochers/new_epocher.py:268-282:
 * KL_div(logits.softmax(1), class2one_hot(target, C)) and the Dice counts of logits.max(1)[1] against target):
 * logits [B * per_sample][K] f32, labels [B * per_sample] int64 -> loss (mean over positions), dlogits_unit (the gradient
 * w.r.t. the logits for grad_loss == 1: the caller scales it), inter / union [B][K] int64 (written, not added to: they
 * need no zero fill -- the workgroups leave count rows in ws and the finishing launch adds them up in fixed order; K == the
 * number of classes).  Per pixel the arithmetic of spcl_softmax_forward / spcl_kl_div_forward / _backward /
 * spcl_softmax_backward in the same order: the same bits.  ws: spcl_kl_workspace_bytes(). */
int spcl_sup_loss_forward(const float* logits, const int64_t* labels, int B, int per_sample, int K, float eps, float* ws,
                          float* loss, float* dlogits_unit, int64_t* inter_zeroed, int64_t* union_zeroed, void* stream);
int spcl_one_hot(const int64_t* labels, size_t npix, int K, float* out, void* stream);
int spcl_argmax_classes(const float* logits, size_t npix, int K, int64_t* out, void* stream);
int spcl_dice_counts(const int64_t* pred, const int64_t* target, int B, int per_sample, int C, int64_t* inter_zeroed,
                     int64_t* union_zeroed, void* stream);

/* Decoder data movement (NHWC, CS a multiple of 16): nn.Upsample(scale_factor=2) of `_UpConv` (unet.py:89) and its
 * backward (H, W = the low-resolution size in both calls), torch.cat((skip, up), dim=1) (unet.py:194-224) and its
 * backward (split).  Channel runs must be multiples of 16 bytes. */
int spcl_upsample2x_forward(const void* x, void* y, int dtype, int N, int H, int W, int CS, void* stream);
int spcl_upsample2x_backward(const void* dy, void* dx, int dtype, int N, int H, int W, int CS, void* stream);
int spcl_concat2_channels(const void* a, const void* b, void* out, int elem_size, size_t npix, int CA, int CB,
                          void* stream);
int spcl_split2_channels(const void* in, void* a, void* b, int elem_size, size_t npix, int CA, int CB, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Optimizer step of the pre-train iteration (contrastyou/trainer/base.py:62 builds RAdam from the un-vendored
 * deepclustering2; the build follows torch.optim.RAdam(decoupled_weight_decay=False)) on ONE
 * flat fp32 parameter:  g' = g + wd p;  m = lerp(m, g', 1-b1);  v = b2 v + (1-b2) g'^2;  t = ++step;
 *   rho_t = rho_inf - 2 t b2^t/(1-b2^t);  p -= lr m/(1-b1^t) * (rho_t > 5 ? rect(rho_t) sqrt(1-b2^t)/(sqrt(v)+eps) : 1).
 * step (int64) and lr (float) live in device memory so that a captured hipGraph replays with an advancing step
 * count and a host-updated learning rate; coef: 4 floats of device scratch.  All buffers 16-byte aligned. */
int spcl_radam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int64_t* step,
                    const float* lr, double beta1, double beta2, double eps, double weight_decay, float* coef,
                    void* stream);
/* the same, and the step's meter updates (spcl_accumulate_scalars' k <= 8 pairs) ride in its coefficient launch: one
 * launch less per training step.  k == 0: exactly spcl_radam_step. */
int spcl_radam_step_scalars(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int64_t* step,
                            const float* lr, double beta1, double beta2, double eps, double weight_decay, float* coef,
                            int k, const void* const* src, void* const* dst, const float* count, void* stream);
/* the same on grad_scale * grad (0 < grad_scale <= 1): the data-parallel MEAN of the flat gradient bucket is its RCCL sum
 * times 1 / world, and that factor is applied while the bucket streams through the optimizer instead of by a separate
 * pass over it (the reference is single-process; its seam for N ranks is semi_seg/main_infonce.py:35,39).
 * grad_scale == 1: bit-identical to spcl_radam_step_scalars. */
int spcl_radam_step_scaled(float* param, const float* grad, double grad_scale, float* exp_avg, float* exp_avg_sq, size_t n,
                           int64_t* step, const float* lr, double beta1, double beta2, double eps, double weight_decay,
                           float* coef, int k, const void* const* src, void* const* dst, const float* count, void* stream);
/* the same WITHOUT the coefficient launch: coef[0..2] = { lr / (1 - b1^t), rho_t > 5 ? rect(rho_t) sqrt(1 - b2^t) : 0,
 * rho_t > 5 } and coef[3] = t were computed by the caller on the host (the formulas above, in double) and are already in
 * device memory -- a step replayed from a hipGraph uploads a few host-written bytes per replay anyway (labels, flip flags:
 * spcl_stage_bytes / spcl_flip_pair_stage), the four floats travel with them.  The launch records t in *step (so that the
 * counter, which spcl_radam_step reads, stays the state_dict's truth) and performs the k meter updates. */
int spcl_radam_apply_staged(float* param, const float* grad, double grad_scale, float* exp_avg, float* exp_avg_sq, size_t n,
                            int64_t* step, const float* coef, double beta1, double beta2, double eps, double weight_decay,
                            int k, const void* const* src, void* const* dst, const float* count, void* stream);

/* Running means of the host-side meters (contrastyou/meters/averagemeter.py via MeterInterface) kept on the device:
 * for i < k (k <= 8):  dst[i][0] += count[i] * src[i][0];  dst[i][1] += count[i].  src / dst / count are HOST arrays
 * (of device pointers / floats) read at call time: ONE launch for all of a step's meter updates. */
int spcl_accumulate_scalars(int k, const void* const* src, void* const* dst, const float* count, void* stream);

/* Per-step HOST inputs of a hipGraph-captured step (the label vector `torch.Tensor(target)` of
 * contrastyou/losses/contrast_loss3.py:133-139, the per-sample flip decisions of semi_seg/epochers/new_pretrain.py:57-58):
 * nbytes (multiple of 4) bytes of host memory are copied to the persistent device block dst, in stream order.  The bytes
 * travel as kernel arguments (read at call time: host_src may be reused as soon as the call returns), 3 584 per launch. */
int spcl_stage_bytes(void* dst, const void* host_src, size_t nbytes, void* stream);
/* two device-to-device copies in one launch (the labelled batch into the captured fine-tune step's persistent input buffers,
 * semi_seg/epochers/new_epocher.py:260-266 hands the step a fresh image / target pair every iteration): sizes and addresses
 * multiples of 16 bytes */
int spcl_copy_pair(void* dst_a, const void* src_a, size_t bytes_a, void* dst_b, const void* src_b, size_t bytes_b,
                   void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Pre-train augmentation on device (replaces the PIL recipe of semi_seg/augment.py:6-22
 * `ACDCStrongTransforms.pretrain`: RandomRotation -> RandomVerticalFlip -> RandomHorizontalFlip -> RandomCrop ->
 * ColorJitter(brightness, contrast) -> ToTensor): one OH x OW view per parameter row, gathered from the device-resident
 * slice store src [S][HS][WS] f32 in [0,1].  params: device int32 [nviews][8] = {slice, cos * 65536, sin * 65536 (rounded;
 * rotation counter-clockwise about the image centre, nearest sampling, 0 outside), flags (1 horizontal flip, 2 vertical
 * flip, 4 contrast before brightness), crop top, crop left, brightness factor (f32 bits), contrast factor (f32 bits)}.
 * out [nviews][OH][OW] f32.  Geometry is integer arithmetic: bit-exact against oracle.augment_view. */
int spcl_augment_views(const float* src, int S, int HS, int WS, const int* params, int nviews, float* out, int OH,
                       int OW, void* stream);
/* The same recipe in PIL's own arithmetic -- what semi_seg/augment.py:6-22 computes through torchvision on 8-bit slices:
 * Image.rotate(angle, NEAREST) (Geometry.c affine_fixed: 16.16 coefficients a0..a5 rounded on the host as PIL rounds them),
 * transposes, crop, ImageEnhance.Brightness / .Contrast (Image.blend on 8-bit values, truncating; Contrast against
 * int(mean + 0.5)), ToTensor (/ 255).  src holds 8-bit grey levels as k / 255.  params: device int32 [nviews][12] =
 * {slice, a0, a1, a2, a3, a4, a5, flags (1 hflip, 2 vflip, 4 contrast before brightness), crop top, crop left, brightness
 * (f32 bits), contrast (f32 bits)}.  Bit-exact against PIL itself: tests/golden/g9_augment.npz (tools/gen_golden.py augment),
 * oracle.augment_view_pil. */
int spcl_augment_views_pil(const float* src, int S, int HS, int WS, const int* params, int nviews, float* out, int OH,
                           int OW, void* stream);
/* The pixel-wise (1x1-convolution) MLP of the dense projector -- contrastyou/projectors/heads.py:28-39,96-120:
 * Conv2d(C, hid, 1) -> LeakyReLU(0.01) -> Conv2d(hid, out, 1) on every pixel of a decoder feature map -- as tiled
 * matrix products over the M = N*H*W pixel rows on the exact-f32 matrix instruction (round 6; until then the rows went through the
 * global projector's 64-row kernels: 158 ms per training step at Up_conv3, 630 ms at Up_conv2).  Row-major operands, f32
 * accumulation; tensors with a dtype argument are f32 or bf16 rows (a channels-last map is read in place through its pitch):
 *   forward          y[M][N]  = act(x)[M][K] W[N][K]^T + bias[N]      act = LeakyReLU(0.01) when leaky_in (x is then a saved f32
 *                                                                      pre-activation); y f32
 *   forward_act      h[M][N]  = LeakyReLU(x[M][K] W[N][K]^T + bias[N])   h of h_dtype.  h carries the pre-activation's sign: it
 *                                                                      stands in for `pre` below
 *   backward_input   dx[M][K] = (g[M][N] W[N][K]) * (pre ? LeakyReLU'(pre[M][K]) : 1)     (with pre: g is f32)
 *   backward_weight  dW[N][K] = sum_m g[m][N] act(x)[m][K],  db[N] = sum_m g[m][N]  (db may be NULL): slabs of rows folded in
 *                    index order (bit-deterministic); ws of spcl_rows_linear_backward_weight_workspace_bytes(M, N, K).
 *   spcl_adaptive_avgpool2d_backward_act: dx = unpool(dout [N][OH][OW][C] f32) * LeakyReLU'(.) with act = LeakyReLU(pre)
 *                    [N][H][W][C] given instead of pre; act and dx of `dtype`; C % 4 == 0.
 * The dense projector's pooled-hidden form (functional._PixelMlpPooledFn): adaptive average pooling commutes with the linear
 * second layer, which then runs on the pooled rows only; the hidden activation and its gradient are stored in the feature
 * map's dtype.  K, N and the row pitches are multiples of 4. */
int spcl_rows_linear_forward(const void* x, int x_dtype, long ldx, int leaky_in, const float* W, const float* bias, int M,
                             int K, int N, float* y, void* stream);
int spcl_rows_linear_forward_act(const void* x, int x_dtype, long ldx, const float* W, const float* bias, int M, int K, int N,
                                 void* h, int h_dtype, void* stream);
int spcl_adaptive_avgpool2d_backward_act(const float* dout, const void* act, int dtype, int N, int H, int W, int C, int OH,
                                         int OW, void* dx, void* stream);
int spcl_rows_linear_backward_input(const void* g, int g_dtype, const float* W, const float* pre, int M, int N, int K, void* dx,
                                    int dx_dtype, long lddx, void* stream);
size_t spcl_rows_linear_backward_weight_workspace_bytes(int M, int N, int K);
int spcl_rows_linear_backward_weight(const void* g, int g_dtype, const void* x, int x_dtype, long ldx, int leaky_in, int M, int N,
                                     int K, float* ws, size_t ws_bytes, float* dW, float* db, void* stream);
/* The reference's other PIL recipes, and the interpolation its wrapper really selects (semi_seg/augment.py:23-37,54-75;
 * contrastyou/augment/synchronize.py:95-103: BILINEAR on images, NEAREST on targets): params[v][28] =
 *   [0] slice [1] flags (1 hflip, 2 vflip, 4 contrast first, 8 bilinear image rotation, 16 crop first = the rotation turns the
 *   crop about its centre) [2] top [3] left [4] pad [5] brightness [6] contrast (f32 bits) [8..13] PIL's 16.16 nearest
 *   coefficients [14..25] the six doubles of PIL's rotation matrix -- of the image the rotation acts on (slice or crop).
 * labels / label_out (optional, together): the slice's u8 label map through the same geometry with NEAREST -> int64. */
int spcl_augment_views_recipe(const float* src, const unsigned char* labels, int S, int HS, int WS, const int* params,
                              int nviews, float* out, long long* label_out, int OH, int OW, int max_pad, void* stream);
/* The same views (bit for bit) from a grid that fills the chip: spcl_augment_views_recipe runs ONE workgroup per view, which
 * samples its pixels twice (ImageEnhance.Contrast needs the view's mean first) -- 180 us for the 60 views of a pre-train batch
 * on a quarter of the CUs.  Here up to 64 workgroups per view sample a chunk each once, park the 8-bit levels and their integer
 * partial sum in `workspace` (spcl_augment_views_recipe_workspace_bytes, 8-byte aligned, contents irrelevant on entry), and
 * a second launch finishes the pixels.  What semi_seg/data/augment.py RecipeViews calls (ABI 7). */
size_t spcl_augment_views_recipe_workspace_bytes(int nviews, int OH, int OW);
int spcl_augment_views_recipe_ws(const float* src, const unsigned char* labels, int S, int HS, int WS, const int* params,
                                 int nviews, float* out, long long* label_out, int OH, int OW, int max_pad, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* Image.resize((OW, OH), BILINEAR) of every slice of a store of 8-bit levels (k / 255) -- torchvision Resize,
 * semi_seg/augment.py:56,71,79 -- with the coefficient rows of Resample.c precomputed by the caller per axis
 * (bounds[out][2] = first tap, tap count; kk[out][ksize]: 22 fractional bits); tmp: [S][HS][OW] floats. */
int spcl_resize_bilinear_pil(const float* src, int S, int HS, int WS, const int* bounds_x, const int* kk_x, int ksize_x,
                             const int* bounds_y, const int* kk_y, int ksize_y, float* tmp, float* out, int OH, int OW,
                             void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Per-sample random flips of an NCHW batch (TensorRandomFlip(axis=[1,2], threshold=0.8), new_epocher.py:112, applied
 * per sample in new_pretrain.py:57-58): out[n] = x[n] flipped along H when flags[n] & 1 and along W when flags[n] & 2.
 * flags: device uint8[N] (the host draws the decisions from python `random`, as the reference does).  x != out. */
int spcl_flip_batch(const void* x, void* out, int elem_size, int N, int C, int H, int W, const uint8_t* flags,
                    void* stream);
/* The pre-train step's input pair in one launch: out [2N][C][H][W] = [ first | flip(second, flags) ] -- the per-sample
 * flip of view 2 and the torch.cat of semi_seg/epochers/new_pretrain.py:57-58,93. */
int spcl_flip_pair(const void* first, const void* second, void* out, int elem_size, int N, int C, int H, int W,
                   const uint8_t* flags, void* stream);
/* the same launch also performs spcl_stage_bytes(stage_dst, host_src, nbytes) and takes the N flag bytes from
 * host_src + flag_off (kernel arguments: nbytes <= 3 584): the two eager launches in front of a replayed step are one. */
int spcl_flip_pair_stage(const void* first, const void* second, void* out, int elem_size, int N, int C, int H, int W,
                         void* stage_dst, const void* host_src, size_t nbytes, size_t flag_off, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Built-in kernel timer (bench.py's live roofline measurement): spcl_profile_enable(1) clears the log and makes every
 * kernel launch of the library record HIP events on its launch stream; spcl_profile_get returns launch i's kernel
 * symbol (as rocprofv3 prints it), its duration and the algorithmic bytes / FLOPs its entry point declared (0 when
 * none).  Not usable inside a hipGraph capture.  spcl_profile_enable(0) clears the log and switches it off. */
int spcl_profile_enable(int on);
int spcl_profile_count(void);
int spcl_profile_get(int i, char* name, int name_cap, float* usec, double* bytes, double* flops);

/* ---------------------------------------------------------------- BatchNorm sums as fixed-point accumulator blocks --------
 * Replaces the finalize launches between a train-mode nn.BatchNorm2d's producer and consumer (semi_seg/arch/unet.py:73,76 and
 * its autograd backward): the producing kernel's epilogue ADDS its tile's sums to a block of 64-bit fixed-point words
 * (integer atomics: exact, order-free, bit-for-bit deterministic), the next launch derives the coefficients in its prologue.
 * A block is spcl_bn_acc_elems(CS) int64 words, ZEROED by the caller before the producer launch (one memset per step for all
 * blocks of a step).  Forward blocks hold sum x / sum x^2, backward blocks sum dz / sum dz (y - mean).  csrc/bn_acc.hpp has
 * the format.  Offered for bf16 layers the specialised 14-column convolution kernels take, with at most 4096 tiles, CS <= 256. */
typedef struct spcl_bn_acc {
  long long* acc;                 /* the forward block of this BatchNorm */
  const float* gamma;             /* [C] */
  const float* beta;              /* [C] */
  float* running_mean;            /* [C] or NULL (no update) */
  float* running_var;             /* [C] or NULL */
  long long* num_batches_tracked; /* or NULL */
  float* st;                      /* [4][CS] mean, invstd, scale, shift: WRITTEN by the consuming launch's first workgroup */
  float momentum, eps;
  float count;                    /* values per channel: N * H * W */
  int C, CS;
} spcl_bn_acc;
size_t spcl_bn_acc_elems(int CS);
/* can spcl_conv3x3_forward_acc run this layer with its input read as it is (in_kind 0), through a BatchNorm + ReLU whose
 * coefficients are derived from a block (1) or come as arrays (2), and (with_stats_acc) its output statistics added to a block? */
int spcl_conv_bn_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int in_kind, int with_stats_acc);
/* spcl_conv3x3_forward with blocks on either side: in_bn != NULL -> the input is the previous convolution's RAW output and
 * relu(scale x + shift) of its BatchNorm is applied by the loader, scale / shift derived from in_bn->acc (no spcl_bn_finalize);
 * in_bn == NULL: scale / shift from in_scale / in_shift when given, else the input is read as it is;
 * stats_acc != NULL -> the output's statistics are added to that block; else stats_rows (per-tile rows, or NULL). */
int spcl_conv3x3_forward_acc(const void* x, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                             const spcl_bn_acc* in_bn, const float* in_scale, const float* in_shift, void* y,
                             long long* stats_acc, float* stats_rows, void* stream);
/* spcl_bn_finalize + spcl_bnrelu_pool_forward in one launch (coefficients derived in the prologue) */
int spcl_bnrelu_pool_forward_acc(const void* y, int dtype, int N, int H, int W, int CS, const spcl_bn_acc* bn, void* act_out,
                                 void* pool_out, void* stream);
/* BN-apply + ReLU of a block's last convolution (unet.py:76-77) with the activation's global average per (image, channel) as
 * a side output: gap_out[n][c] = mean over the pixels of act_out as stored -- the AdaptiveAvgPool2d((1, 1)) at the head of
 * the projector (contrastyou/projectors/heads.py:78-92, nn.py:56-58) then has nothing to read back (spcl_proj_forward with
 * feat == NULL takes the rows).  bn == NULL: coefficients from scale / shift; else derived from the accumulator block as in
 * spcl_bnrelu_pool_forward_acc (scale / shift unused).  spcl_bnrelu_gap_supported: small maps (H W <= 4096), CS <= 256. */
int spcl_bnrelu_gap_supported(int dtype, int H, int W, int C, int CS);
int spcl_bnrelu_gap_forward(const void* y, int dtype, int N, int H, int W, int C, int CS, const float* scale,
                            const float* shift, const spcl_bn_acc* bn, void* act_out, float* gap_out, void* stream);
/* spcl_conv3x3_dgrad_bnstats / _poolstats with the BatchNorm-backward sums added to a block instead of written as rows */
int spcl_conv_dgrad_bnstats_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS);
int spcl_conv3x3_dgrad_bnstats_acc(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                                   void* g, const void* y2, const float* scale2, const float* shift2, const float* mean2,
                                   long long* acc, void* stream);
int spcl_conv_dgrad_poolstats_acc_supported(int dtype, int N, int H, int W, int CinK, int CoutS, int H2, int W2);
int spcl_conv3x3_dgrad_poolstats_acc(const void* dy, int dtype, int N, int H, int W, int CinK, int CoutS, const void* w_packed,
                                     void* g, const void* y2, int H2, int W2, const float* scale2, const float* shift2,
                                     const float* mean2, long long* acc, void* stream);
/* the BatchNorm + ReLU (+ max-pool) backward's apply pass with A, B derived from a backward block in its prologue (no reduction
 * pass where a dgrad filled the block, no finalize launch): exactly one of dact [N][H][W][CS], dpool [N][H/2][W/2][CS]
 * (block already filled) or dact_nc [N][CS] (one value per image and channel: this call's reduction pass fills the block).
 * st4 = the forward's [4][CS].  dgamma / dbeta [C] are written by the apply launch. */
int spcl_bnrelu_backward_acc(const void* y, const void* dact, const void* dpool, const void* dact_nc, int dtype, int N, int H,
                             int W, int C, int CS, const float* st4, int training, long long* acc, float* dgamma, float* dbeta,
                             void* dy, void* stream);
/* ... and for a gradient that did NOT come with its sums -- an activation with several consumers (the decoder's skip
 * connections, unet.py:193-230), the up-convolutions (:85-97), a gradient that arrives as a channel slice of a wider tensor
 * (dact_stride > 0 elements per pixel) or at twice the resolution (d_up [N][2H][2W][CS] non-NULL: dact is then scratch the call
 * fills with the 2 x 2 sums) --: the reduction pass ADDS its sums to the zeroed block `acc`, the apply pass derives the
 * coefficients from it; two launches, no finalize launch between them.  Exactly spcl_bnrelu_pool_backward(_strided) /
 * spcl_bnrelu_backward_up2's results (the sums in another, fixed-point order).  CS <= 256. */
int spcl_bnrelu_backward_fill_acc(const void* y, void* dact, int dact_stride, const void* dpool, const void* d_up, int dtype,
                                  int N, int H, int W, int C, int CS, const float* st4, int training, long long* acc,
                                  float* dgamma, float* dbeta, void* dy, void* stream);
/* ... and for a gradient whose sums came as per-tile rows [nrows][2][CS] (sum dz, sum dz (y - mean): what the dgrad epilogues
 * leave where a layer has more tiles than the blocks take adds from): one small launch adds the rows to the zeroed block, the
 * apply pass (exactly one of dact / dpool) derives its coefficients: spcl_bnrelu_backward_rows / spcl_bnrelu_pool_backward_rows
 * without their finalize launch. */
int spcl_bnrelu_backward_rows_acc(const void* y, const void* dact, const void* dpool, const float* rows, int nrows, int dtype,
                                  int N, int H, int W, int C, int CS, const float* st4, int training, long long* acc,
                                  float* dgamma, float* dbeta, void* dy, void* stream);

#ifdef __cplusplus
}
#endif
#endif
