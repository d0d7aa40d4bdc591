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
 *  - dtype: SPCL_F32 = 0 (parity mode, exact-f32 MFMA), SPCL_BF16 = 1 (bf16 storage, f32 accumulate)
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
/* lazily materialise the [2n,2n] hook taps (contrast_loss3.py:175-178,188): any pointer may be NULL */
int spcl_supcon_materialize(const float* labels, const float* mask, int n, int d, float temperature, int sp_mode,
                            float gamma, const float* ws_fwd, float* sim_logits, float* sim_exp, float* pos_mask,
                            float* neg_mask, float* sp_mask, void* stream);

/* ---------------------------------------------------------------- projector ------------------------------
 * Replaces contrastyou/projectors/heads.py:9-25,78-92 + nn.py:8-15,29-36,56-58:
 * AdaptiveAvgPool2d((1,1)) -> Flatten -> Linear -> LeakyReLU(0.01) -> Linear -> F.normalize(p=2,dim=1).
 * feat [N,H,W,Cs] NHWC (dtype), first C of Cs channels used; w1 [hid,C] b1 [hid] w2 [out,hid] b2 [out] f32
 * (hid == 0 -> "linear" head: w1 is [out,C], w2/b2 ignored).  Saved for backward: pooled [N,C], pre [N,hid],
 * o [N,out] (un-normalised), all f32.  z [N,out] f32.
 */
int spcl_proj_forward(const void* feat, int dtype, int N, int HW, int C, int Cs, const float* w1, const float* b1,
                      const float* w2, const float* b2, int hid, int out_dim, int normalize, float* pooled,
                      float* pre, float* o, float* z, void* stream);
/* dz [N,out] -> dw1,db1,dw2,db2 (f32, overwritten) and dfeat [N,H,W,Cs] (dtype, overwritten; NULL to skip) */
int spcl_proj_backward(const float* dz, int dtype, int N, int HW, int C, int Cs, const float* w1, const float* w2,
                       int hid, int out_dim, int normalize, const float* pooled, const float* pre, const float* o,
                       float* dw1, float* db1, float* dw2, float* db2, float* scratch /* [N,hid]+[N,out]+[N,C] f32 */,
                       void* dfeat, void* stream);

/* ---------------------------------------------------------------- encoder --------------------------------
 * Replaces semi_seg/arch/unet.py:67-82 (_ConvBlock: Conv2d 3x3 no bias -> BatchNorm2d -> ReLU, x2),
 * :118-121 (MaxPool2d 2x2), :156-190 (forward until Conv5) and their autograd backward (K1-K4, K17).
 */

/* weights: OIHW f32 master (state_dict layout) <-> MFMA fragment-packed kernel layout.
 * kind 0: forward  (K = tap*CinP + ci, N = co)      kind 1: dgrad (K = tap'*CoutP + co, N = ci, taps flipped)
 * packed size in elements: spcl_conv_packed_elems(Cin,Cout,kind) of `dtype` */
size_t spcl_conv_packed_elems(int Cin, int Cout, int kind, int dtype);
int spcl_conv_pack_weights(const float* w_oihw, int Cin, int Cout, int kind, int dtype, void* packed, void* stream);

/* y = conv3x3(act(x)) for Cin>=1 via implicit GEMM on MFMA; NHWC.
 * x [N,H,W,CinS]; in_mode 0: act = identity, 1: act = relu(scale[c]*x+shift[c]) (fused BN-apply+ReLU of the
 * producer layer, unet.py:73-74 applied on load).  y [N,H,W,CoutS] raw conv output.
 * stats != NULL: per-workgroup Chan partials (count, mean, M2) per output channel written to
 * stats[spcl_conv_num_tiles(...)][CoutS][3] f32 for the train-mode BatchNorm that follows (unet.py:73,76). */
int spcl_conv_num_tiles(int N, int H, int W);
int spcl_conv3x3_forward(const void* x, int dtype, int N, int H, int W, int CinS, int CoutS, const void* w_packed,
                         int in_mode, const float* in_scale, const float* in_shift, void* y, float* stats,
                         void* stream);
/* first layer (tiny Cin, e.g. 1): direct conv, x [N,H,W,Cin] f32 (== NCHW for Cin==1), w OIHW f32 */
int spcl_conv3x3_first_forward(const float* x, int N, int H, int W, int Cin, int Cout, int CoutS, const float* w_oihw,
                               int dtype, void* y, float* stats, void* stream);
int spcl_conv3x3_first_wgrad(const float* x, const void* dy, int dtype, int N, int H, int W, int Cin, int Cout,
                             int CoutS, float* partial /* [nblk][Cout*Cin*9] */, int nblk, float* dw_oihw,
                             void* stream);
/* dW (OIHW f32, overwritten) = sum_pixels act(x)[p+tap] (x) dy[p];  x/in_mode as in forward.
 * partial: workspace of spcl_conv_wgrad_workspace_bytes() */
size_t spcl_conv_wgrad_workspace_bytes(int N, int H, int W, int CinS, int CoutS);
int spcl_conv3x3_wgrad(const void* x, const void* dy, int dtype, int N, int H, int W, int Cin, int CinS, int Cout,
                       int CoutS, int in_mode, const float* in_scale, const float* in_shift, float* partial,
                       float* dw_oihw, void* stream);

/* train-mode BatchNorm statistics (unet.py:73,76; torch.nn.BatchNorm2d momentum/eps semantics):
 * combines the conv epilogue partials -> mean, invstd, scale=gamma*invstd, shift=beta-mean*scale and updates
 * running_mean/var (unbiased var, momentum).  eval mode: spcl_bn_eval_affine builds scale/shift from running stats */
int spcl_bn_finalize(const float* stats, int ntiles, int C, int CS, const float* gamma, const float* beta,
                     float momentum, float eps, float* running_mean, float* running_var, float* mean, float* invstd,
                     float* scale, float* shift, void* stream);
int spcl_bn_eval_affine(int C, int CS, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift, void* stream);

/* a = relu(scale*y+shift) [N,H,W,CS] (act_out, may be NULL) and/or 2x2/2 max-pooled p [N,H/2,W/2,CS]
 * (pool_out, may be NULL)  -- unet.py:74,77 + :118-121 */
int spcl_bnrelu_pool_forward(const void* y, int dtype, int N, int H, int W, int CS, const float* scale,
                             const float* shift, void* act_out, void* pool_out, void* stream);
/* backward of the above + BatchNorm backward:
 *   g = d(act) (+ scatter of d(pool) to the window arg-max, first max in scan order as torch.max_pool2d)
 *   dz = g*[scale*y+shift>0];  dbeta=sum dz;  dgamma=sum dz*yhat;  dy = scale*(dz - dbeta/M - yhat*dgamma/M)
 * two launches inside: reduce (deterministic two-stage) then apply.  dact/dpool may be NULL (not both). */
size_t spcl_bnrelu_bwd_workspace_bytes(int N, int H, int W, int CS);
int spcl_bnrelu_pool_backward(const void* y, const void* dact, const void* dpool, int dtype, int N, int H, int W,
                              int C, int CS, const float* gamma, const float* mean, const float* invstd,
                              const float* scale, const float* shift, float* ws, float* dgamma, float* dbeta,
                              void* dy, void* stream);

/* layout helpers: NCHW f32 <-> NHWC(dtype, channel-padded) */
int spcl_nchw_to_nhwc(const float* src, int N, int C, int H, int W, int CS, int dtype, void* dst, void* stream);
int spcl_nhwc_to_nchw(const void* src, int dtype, int N, int C, int H, int W, int CS, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif
