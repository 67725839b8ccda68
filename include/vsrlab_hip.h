/* vsrlab_hip.h -- C ABI of libvsrlab_hip.so: the MI355X-native BasicVSR forward/backward path.
 *
 * The reference (santurini/vsrlab) has no FFI: its boundary for this path is the Python
 * nn.Module protocol `sr = BasicVSR(...)(lrs)` resolved by Hydra `_target_`
 * (src/core/utils.py:138; src/vsr/models/RealBasicVSR/modules/basicvsr.py:39-83).  This header is
 * what a binding for that path attaches to: plain pointers and sizes, no torch types.  Every entry
 * point
 *   - takes DEVICE pointers (HBM) unless stated otherwise, and a hipStream_t passed as void*;
 *   - only enqueues work on that stream (the whole-path engine forks a helper stream from it and joins it back
 *     with events, so for the caller all work is ordered in the stream it passed): no device allocation, no
 *     synchronisation.  It may be called from PyTorch's main thread and from its autograd thread.  Process-wide
 *     state is limited to caches that never change results: per-device "dynamic-LDS attribute set" flags and
 *     compute-unit counts, the A/B environment switches read once (VSRLAB_AMD_*), and one helper stream +
 *     two events per (thread, device), created on first use and released when the process exits;
 *   - returns 0 (VSR_OK) or a negative status, never throws; NULL / non-positive arguments are VSR_STATUS_BADARG.
 *
 * Tensor conventions
 *   boundary tensors are the reference's own: LR clip (n,t,3,h,w) fp32 planar, SR clip
 *   (n,t,3,S h,S w) fp32 planar (S = upscale), parameters/gradients fp32 OIHW, flow (N,2,H,W) fp32 planar with
 *   channel 0 = dx.  "pm" (pixel-major) tensors are the library's internal BLOCKED layout
 *       [N][H][ceil(W/32)][C/8][32 pixels][8 channels]          (csrc/common.h: pm_off())
 *   with element type `dtype` (VSR_DT_F32 / VSR_DT_BF16) and C a multiple of 16: inside a 32-pixel row segment
 *   the 8-channel chunks of the 32 pixels are contiguous (512 B per chunk in bf16) -- the MFMA accumulator
 *   layout of the conv kernels.  It is NOT plain NHWC: fill and read pm tensors ONLY through
 *   vsr_planar_to_pm / vsr_pm_to_planar (rows are padded to whole 32-pixel segments; the padding is never read).
 *   The per-op entry points that take pm tensors exist so that each kernel can be parity-tested in isolation.
 */
#ifndef VSRLAB_HIP_H
#define VSRLAB_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VSR_DT_F32 0   /* exact-fp32 build: v_mfma_f32_32x32x2_f32, parity gate */
#define VSR_DT_BF16 1  /* bf16 storage, fp32 accumulate: v_mfma_f32_16x16x32_bf16 (hot 3x3 64->64 kernels) /
                          v_mfma_f32_32x32x16_bf16 (generic shapes), perf build */

#define VSR_STATUS_OK 0
#define VSR_STATUS_BADARG (-1)
#define VSR_STATUS_UNSUPPORTED (-2)
#define VSR_STATUS_HIP (-3)
#define VSR_STATUS_WORKSPACE (-4)

int vsr_abi_version(void);
const char* vsr_status_string(int status);

/* ---- whole-path engine -------------------------------------------------------------------
 * Replaces BasicVSR.forward and its autograd backward (basicvsr.py:39-83).
 * mid_channels must be 64 (the reference's default, basicvsr.py:12-13); upscale 4 (default: two
 * PixelShufflePacks, x4 bilinear skip) or 2 (ONE PixelShufflePack = upscale // 2, conv_last at 2h x 2w,
 * x2 bilinear skip: basicvsr.py:19-22; conf/train/model/basicvsr.yaml:4 takes the scale from the dataset
 * config); res_blocks >= 1; t <= 32 (test.py's window).  A trunk-chain launch that gave up a dependency
 * wait (csrc/conv3x3_chain.hip) overwrites the head of its last image with NaN: sr / the gradients of
 * that call are then not finite -- results of such a call are never silently wrong.            */
typedef struct VsrBasicVSRDesc {
    int n, t, h, w;      /* LR clip (n,t,3,h,w) */
    int mid_channels;
    int res_blocks;
    int upscale;
    int dtype;           /* VSR_DT_* of the internal activations */
    int arena_mode;      /* training workspace (ABI 3): 0 = every activation and activation gradient of the clip stays in the
                          * workspace and the weight gradients run as all-frames launches (fastest; 113 GiB per clip at BASELINE config 2
                          * since round 4 -- the two propagation directions share one set of activation-gradient buffers, the backward
                          * runs them one after the other -- so two clips fit one 288 GB GPU);
                          * 1 = "diet": the trunks' activation gradients live in a two-block ring and each frame's weight gradients
                          * are launched behind its data gradients; the two HR tensors of a frame (upsample.1 / conv_last.0 outputs)
                          * are recomputed in the backward (65 GiB at config 2; +18 % time per step; same results up to the
                          * summation order of the weight gradients over frames).  Ignored by inference (need_backward = 0).     */
} VsrBasicVSRDesc;

/* Number of parameter tensors.  Order, by the reference module's state_dict KEYS (state_dict()
 * itself lists spynet.mean/std before the SPyNet convs; this ABI puts them last):
 *   for trunk in (backward_resblocks, forward_resblocks):
 *       conv.0.weight, conv.0.bias, then res_block.{i}.conv1.{weight,bias}, conv2.{weight,bias}
 *   point_conv.0.{weight,bias}; upsample.{0,1}.upconv.{weight,bias} (upscale 2: upsample.0 only);
 *   conv_last.0.{weight,bias}; conv_last.2.{weight,bias};
 *   spynet.basic_module.{0..5}.basic_module.{0..4}.conv.0.{weight,bias}; spynet.mean; spynet.std */
int vsr_basicvsr_num_params(const VsrBasicVSRDesc* d);

/* Bytes of workspace the caller must provide (same buffer for forward and its backward).
 * need_backward = 0: inference (activations are not retained); 1: training with the flow net
 * frozen (train_flow=False, basicvsr.py:25-28); 2: training incl. SPyNet (train_flow=True:
 * SPyNet's activations are retained too).                                                    */
size_t vsr_basicvsr_workspace_bytes(const VsrBasicVSRDesc* d, int need_backward);

/* sr = BasicVSR(lrs).  `params`: HOST array of num_params device pointers.                   */
int vsr_basicvsr_forward(const VsrBasicVSRDesc* d, const float* const* params, int nparams,
                         const float* lrs, float* sr, void* workspace, size_t workspace_bytes,
                         int need_backward, void* stream);

/* Back-propagates dsr (n,t,3,S h,S w) through the forward that last ran on `workspace`
 * (need_backward >= 1).  grads[k] (same order/shape as params; NULL = not wanted) are ACCUMULATED
 * into (+=).  SPyNet entries: all NULL = frozen flow net; otherwise the forward must have run with
 * need_backward = 2 and all 60 conv tensors get their gradient (flow gradient of the propagation
 * warps, spynet.py:95-106, then SPyNet's own backward, spynet.py:38-93).
 * dlrs (n,t,3,h,w) fp32 or NULL: the gradient w.r.t. the input clip is WRITTEN here (what RealBasicVSR's
 * pre-clean stack receives): bilinear xS skip + the stems' LR channels + the flows through SPyNet's image
 * pyramid; also needs need_backward = 2.                                                        */
int vsr_basicvsr_backward(const VsrBasicVSRDesc* d, const float* const* params, float* const* grads,
                          int nparams, const float* lrs, const float* dsr, float* dlrs, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Copies the optical flows computed by the last forward: (n,t-1,2,h,w) each
 * (BasicVSR.compute_flow, basicvsr.py:30-37).                                               */
int vsr_basicvsr_get_flows(const VsrBasicVSRDesc* d, const void* workspace, float* flow_forward,
                           float* flow_backward, void* stream);

/* ---- SPyNet alone: flow = Spynet(ref, supp)  (RealBasicVSR/modules/spynet.py:69-93) -------
 * ref/supp (N,3,h,w) fp32 planar; params: the 62 spynet tensors in state_dict order.        */
size_t vsr_spynet_workspace_bytes(int N, int h, int w, int dtype, int need_backward);
int vsr_spynet_forward(int N, int h, int w, int dtype, const float* const* params, int nparams,
                       const float* ref, const float* supp, float* flow, void* workspace,
                       size_t workspace_bytes, int need_backward, void* stream);
/* Parameter gradients of the forward that last ran on `workspace` with need_backward = 1, for the
 * cotangent dflow (N,2,h,w): grads[k] += d<flow, dflow>/d params[k] for the 60 conv tensors (NULL =
 * not wanted; a bias needs its weight's entry).  ref / supp are not differentiated.             */
int vsr_spynet_backward(int N, int h, int w, int dtype, float* const* grads, int nparams,
                        const float* dflow, void* workspace, size_t workspace_bytes, void* stream);

/* Forward of the canonical SPyNet variants: last_relu = 0 drops the ReLU the RealBasicVSR copy has after every level's last
 * conv (vsr/models/VRT/modules/spynet.py:68-157); level_out: HOST array of 6 device pointers (NULL = not wanted), entry l
 * receives level l's flow resized to (N,2, h >> (5-l), w >> (5-l)) -- VRT's return_levels.  Same workspace query;
 * need_backward != 0 (ABI 3) keeps the activations for vsr_spynet_backward_ex.                                          */
int vsr_spynet_forward_ex(int N, int h, int w, int dtype, const float* const* params, int nparams, const float* ref,
                          const float* supp, int last_relu, float* const* level_out, void* workspace,
                          size_t workspace_bytes, int need_backward, void* stream);
/* Backward of vsr_spynet_forward / _forward_ex (need_backward != 0) incl. the gradient w.r.t. the two input frames (through
 * the border warps, the image pyramid, the /32 resize and the normalisation): dref, dsupp (N,3,h,w) fp32 are WRITTEN
 * (either may be NULL); grads may be NULL (frozen flow net); params: the forward's 62 tensors.  Cotangents: dflow (N,2,h,w)
 * of the full-resolution flow and / or (ABI 3) dlevel, a HOST array of 6 device pointers (NULL = none) in the layout of
 * forward_ex's level_out; last_relu as in the forward (vsr_spynet_forward: 1).                                          */
int vsr_spynet_backward_ex(int N, int h, int w, int dtype, const float* const* params, float* const* grads, int nparams,
                           const float* dflow, int last_relu, const float* const* dlevel, float* dref, float* dsupp,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- RealBasicVSR pre-clean stack, forward: lq = IterativeRefinement(lr) ----------------------
 * (vsr/models/RealBasicVSR/realbasicvsr.py:17-30): `steps` times x <- x + conv(ResidualBlock(x)) on
 * the F = n*t frames (F,3,h,w) fp32 planar.  params (4 + 4*blocks tensors): resblock.conv.0.{weight,
 * bias}, resblock.res_block.{i}.conv1.{weight,bias}, conv2.{weight,bias} ..., conv.{weight,bias}.
 * lq is a fresh tensor (the reference updates lr in place: SURVEY.md appendix A3).             */
size_t vsr_cleaner_workspace_bytes(int F, int h, int w, int blocks, int steps, int dtype, int need_backward);
int vsr_cleaner_forward(int F, int h, int w, int mid_channels, int blocks, int steps, int dtype,
                        const float* const* params, int nparams, const float* lr, float* lq,
                        void* workspace, size_t workspace_bytes, int need_backward, void* stream);
/* Backward of the forward that last ran on `workspace` with need_backward = 1, for the cotangent dlq (F,3,h,w):
 * grads[k] (same order/shape as params; NULL = not wanted, a bias needs its weight's entry) are ACCUMULATED
 * into (the parameters are shared by the `steps` iterations); dlr (F,3,h,w) or NULL is written.  `lr` is the
 * forward's input (its first step's stem weight gradient needs it).                                        */
int vsr_cleaner_backward(int F, int h, int w, int mid_channels, int blocks, int steps, int dtype,
                         float* const* grads, int nparams, const float* lr, const float* dlq, float* dlr,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-op entry points (pixel-major tensors) --------------------------------------------- */
/* flow_warp, zeros padding (spynet.py:95-106): out[n,y,x,:] = bilinear(in[n], x+fx, y+fy)     */
int vsr_flow_warp_fwd(int dtype, const void* in_pm, const float* flow, void* out_pm, int N, int H,
                      int W, int C, void* stream);
/* its backward w.r.t. `in`: dacc (fp32 pixel-major, caller-zeroed) += scatter(dout)           */
int vsr_flow_warp_bwd(int dtype, const void* dout_pm, const float* flow, float* dacc_pm_f32, int N,
                      int H, int W, int C, void* stream);

/* its backward w.r.t. the flow (grid_sampler_2d_backward's grid gradient): dflow (N,2,H,W) fp32   */
int vsr_flow_warp_bwd_flow(int dtype, const void* in_pm, const void* dout_pm, const float* flow,
                           float* dflow, int N, int H, int W, int C, void* stream);

/* the same three with grid_sample's padding_mode: 0 = 'zeros', 1 = 'border' (flow_warp(..., padding_mode='border'),
 * spynet.py:60,95: clamped sample coordinates; a clamped coordinate has zero flow gradient)                        */
int vsr_flow_warp_fwd_ex(int dtype, const void* in_pm, const float* flow, void* out_pm, int N, int H, int W, int C,
                         int padding_mode, void* stream);
int vsr_flow_warp_bwd_ex(int dtype, const void* dout_pm, const float* flow, float* dacc_pm_f32, int N, int H, int W, int C,
                         int padding_mode, void* stream);
int vsr_flow_warp_bwd_flow_ex(int dtype, const void* in_pm, const void* dout_pm, const float* flow, float* dflow, int N,
                              int H, int W, int C, int padding_mode, void* stream);

/* layout converters between the reference's planar fp32 and pixel-major `dtype`              */
int vsr_planar_to_pm(int dtype, const float* in, void* out_pm, int N, int Cin, int H, int W, int C,
                     void* stream);
int vsr_pm_to_planar(int dtype, const void* in_pm, float* out, int N, int Cout, int H, int W, int C,
                     void* stream);

/* y = act(conv3x3(x, w) + b) [+ res]   64->64, stride 1, pad 1 (core/modules/conv.py:85-86).
 * w: fp32 OIHW (64,64,3,3) ; wpack: scratch of 9*64*64 elements of `dtype`, filled from w by this call
 * (w == NULL: wpack is taken as already packed by an earlier call -- lets a caller time the conv alone);
 * act: 0 none / 1 ReLU / 2 LeakyReLU(0.1); res_pm may be NULL.                               */
int vsr_conv3x3_c64_fwd(int dtype, const void* x_pm, const float* w, const float* b, void* wpack,
                        void* y_pm, const void* res_pm, int act, int N, int H, int W, void* stream);

/* A chain of `nlayers` (2 x blocks, <= 64) dependent 3x3 64->64 convolutions in ONE launch: the body of ResidualBlock.res_block, a
 * Sequential of ResidualConv (core/modules/conv.py:85-92, 99-103): x + conv2(relu(conv1(x))) per block.  What BasicVSR's two
 * propagation branches run per frame behind their stem conv (basicvsr.py:56-58, 71-73), and with the flipped weight packs their
 * data gradients.  The workgroups of the launch stay resident over all layers and hand finished 8x32-pixel tiles to each other
 * through per-tile flags in `sync` (vsrlab_amd/csrc/conv3x3_chain.hip); results are bit-identical to nlayers calls of
 * vsr_conv3x3_c64_fwd.  bf16 only.
 *   images: nlayers + 1 consecutive blocked images of N x H x W x 64 bf16 (vsr_planar_to_pm layout): image 0 = x (input);
 *           layer l reads image l and writes image l + 1.  Even l: act = ReLU(conv + bias); odd l: conv + bias + image l - 1
 *           (the block's identity).  Every intermediate stays (the training layout: nothing is overwritten).
 *   wpack : nlayers packed weight sets of 9*64*64 bf16 (as vsr_conv3x3_c64_fwd leaves them in its `wpack`), bias: nlayers x 64 fp32
 *   sync  : vsr_conv3x3_c64_chain_sync_bytes(...) bytes of device scratch, owned by the launch until it has finished
 * images, wpack and bias must be 256-byte aligned and within 1 TiB of each other (VSR_ERR_UNSUPPORTED otherwise).             */
/* (ABI 4) */
size_t vsr_conv3x3_c64_chain_sync_bytes(int nlayers, int N, int H, int W);
int vsr_conv3x3_c64_chain_fwd(const void* images, const void* wpack, const float* bias, int nlayers, int N, int H, int W,
                              void* sync, void* stream);
/* dx = conv3x3_transpose(dy, w) (+res) (* mask(aux)): the data gradient of the same conv;
 * mask_mode: 0 none / 1 ReLU' of aux / 2 LeakyReLU' of aux                                   */
int vsr_conv3x3_c64_dgrad(int dtype, const void* dy_pm, const float* w, void* wpack, void* dx_pm,
                          const void* res_pm, const void* aux_pm, int mask_mode, int N, int H,
                          int W, void* stream);
/* One convolution layer (forward; its backward is vsr_conv_layer_bwd below), for the building blocks the reference's modules expose on their own: ConvReLU
 * (core/modules/conv.py:15-22), SpynetModule (spynet.py:13-21), PixelShufflePack (upsampling.py:4-12), the stem of
 * ResidualBlock (conv.py:97).  w fp32 OIHW, b fp32 or NULL.
 *   ks 3 / 1: 64 -> 64 (pixel_shuffle != 0: 64 -> 256 written as (N,2H,2W,64));
 *   ks 7    : (cin_pm, cout_real) in {(16,32), (32,64), (64,32), (32,16), (16,2: y_planar)} with cin_real = 8,32,64,32,16;
 *   lr_planar (N,3,H,W), ks 3: the conv reads cat([lr, x_pm]) (cin_real 67) or lr alone (x_pm NULL, cin_real 3).
 * y_pm: cd channels per pixel; y_planar (N,cout_real,H,W) for cout_real <= 4.  wpack: scratch, 49*64*64*4 elements.   */
int vsr_conv_layer_fwd(int dtype, int ks, const void* x_pm, int cin_pm, const float* lr_planar, const float* w,
                       const float* b, int cin_real, int cout_real, void* wpack, void* y_pm, int cd, float* y_planar,
                       int act, float slope, int pixel_shuffle, int N, int H, int W, void* stream);
/* Backward of ONE layer of vsr_conv_layer_fwd (same shapes and argument meaning; the reference's ConvReLU / ResidualBlock stem /
 * PixelShufflePack / SpynetModule layers are ordinary autograd modules: core/modules/conv.py:15-22,94-103, upsampling.py:4-12,
 * spynet.py:13-21).  y_pm / y_planar: the forward's OUTPUT (source of the activation mask; may be NULL for act == 0);
 * dy_pm / dy_planar: its cotangent, same layout.  Outputs (each may be NULL = not needed): dx_pm (layout of x_pm), dlr_planar
 * (N,3,H,W) fp32, gw (OIHW fp32, overwritten), gb (overwritten).  scratch: vsr_conv_layer_bwd_scratch_bytes() bytes.        */
size_t vsr_conv_layer_bwd_scratch_bytes(int dtype, int N, int H, int W, int pixel_shuffle);
int vsr_conv_layer_bwd(int dtype, int ks, const void* x_pm, int cin_pm, const float* lr_planar, const float* w, int cin_real,
                       int cout_real, const void* y_pm, const float* y_planar, const void* dy_pm, const float* dy_planar, int cd,
                       int act, float slope, int pixel_shuffle, void* dx_pm, float* dlr_planar, float* gw, float* gb,
                       void* scratch, size_t scratch_bytes, int N, int H, int W, void* stream);
/* gw (64,64,3,3) and gb (64) fp32, overwritten.  slab: fp32 scratch of
 * vsr_conv3x3_c64_wgrad_slab_floats() floats.                                                */
size_t vsr_conv3x3_c64_wgrad_slab_floats(void);
int vsr_conv3x3_c64_wgrad(int dtype, const void* x_pm, const void* dy_pm, float* gw, float* gb,
                          float* slab, int N, int H, int W, void* stream);

/* Charbonnier loss value and gradient (core/losses.py:10-18): loss = mean(sqrt((sr-hr)^2+eps)).
 * sr / hr / dsr 16-byte aligned.  scratch: vsr_charbonnier_scratch_floats() floats (per-workgroup partial sums; the value is
 * reduced in a fixed order, so it is bit-identical from run to run -- ABI 2; ABI 1 had no scratch and summed with atomics). */
size_t vsr_charbonnier_scratch_floats(void);
int vsr_charbonnier_fwd_bwd(const float* sr, const float* hr, float* dsr, float* loss, float* scratch, long long numel,
                            float eps, void* stream);

/* ---- training-step glue over FLAT fp32 arenas (csrc/train_step.hip) ---------------------------------------
 * Replaces the reference's `update_weights` tail (core/utils.py:270-280): clip_grad_norm_(model.parameters(),
 * grad_clip) + torch.optim.Adam.step() (conf/train/optimizer/adam.yaml), for a model whose parameters, gradients
 * and Adam moments each live in ONE contiguous, 16-byte-aligned fp32 buffer of `numel` elements (the HIP backward
 * writes all gradients into such an arena).  scratch: vsr_optim_scratch_floats() floats.
 *   total_norm = |grad_scale| * ||grads||_2                        -> norm_out[0] (device, optional)
 *   g' = grad_scale * min(1, max_norm / (total_norm + 1e-6)) * g   (max_norm <= 0: no clipping) (+ weight_decay * p)
 *   exp_avg, exp_avg_sq, params updated with torch.optim.Adam's formulas for step number `step` (1-based).
 * `grads` is not modified (clip_grad_norm_ scales it in place; the reference zeroes it right after the step).
 * A non-finite total_norm leaves params and moments untouched (GradScaler.step's behaviour, train.py:74).   */
size_t vsr_optim_scratch_floats(void);
int vsr_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                       float max_norm, float* scratch, float* norm_out, void* stream);
/* norm_out[0] = |grad_scale| * ||grads||_2 alone (clip_grad_norm_'s return value). */
int vsr_grad_norm(const float* grads, long long numel, float grad_scale, float* scratch, float* norm_out, void* stream);

/* out (planes,h,w) = bilinear resize of in (planes,H,W), align_corners=False, no antialiasing: kornia's
 * resize(hr, (h, w)) in compute_loss (core/utils.py:235-240), the target of the pre-clean stack's loss term.  */
int vsr_resize_bilinear(const float* in, float* out, long long planes, int H, int W, int h, int w, void* stream);

/* ---- GAN side of RealBasicVSR training (BASELINE config 3; csrc/conv_wide.hip, csrc/disc_engine.hip) -------------
 * UNetDiscriminator.forward and its backward (vsr/models/RealBasicVSR/modules/unet-discriminator.py:4-31) on a batch of
 * frames img (n,3,h,w) fp32 planar, h and w multiples of 8 -> logits out (n,1,h,w) fp32 planar.  mid_ch must be 64.
 * params: 12 device pointers, fp32 OIHW, in module order: conv_0.weight, conv_0.bias, the EFFECTIVE weights of conv_1 ..
 * conv_8 (= weight_orig / sigma, see vsr_spectral_norm; SpectralConv has no bias, core/modules/conv.py:6-13),
 * conv_9.weight, conv_9.bias.                                                                                     */
typedef struct VsrDiscDesc { int n, h, w, mid_ch, dtype; } VsrDiscDesc;
size_t vsr_disc_workspace_bytes(const VsrDiscDesc* d, int need_backward);
int vsr_disc_forward(const VsrDiscDesc* d, const float* const* params, int nparams, const float* img, float* out,
                     void* workspace, size_t workspace_bytes, int need_backward, void* stream);
/* Back-propagates dout (n,1,h,w) through the forward that last ran on `workspace` (need_backward = 1).  grads[k]
 * (layout of params; NULL = not wanted, a bias needs its weight's entry) are ACCUMULATED into; dimg (n,3,h,w) or NULL is
 * written (the generator's adversarial gradient, train_gan.py:35-48).                                              */
int vsr_disc_backward(const VsrDiscDesc* d, float* const* grads, int nparams, const float* img, const float* dout,
                      float* dimg, void* workspace, size_t workspace_bytes, void* stream);

/* torch.nn.utils.spectral_norm's compute_weight (n_power_iterations = 1, eps 1e-12) for one SpectralConv
 * (core/modules/conv.py:9): w_orig viewed as (rows = cout, cols = cin*kh*kw).  training != 0: u and v are UPDATED in
 * place by one power iteration first (v = normalize(W^T u), u = normalize(W v)); then sigma[0] = u.(W v) and
 * w_out = w_orig / sigma.  rows <= 512.                                                                            */
int vsr_spectral_norm(const float* w_orig, float* u, float* v, float* w_out, float* sigma, int rows, int cols,
                      int training, void* stream);
/* dw_orig += dw / sigma - (sum(dw * w_orig) / sigma^2) u v^T   (u, v are constants of the graph, as in torch).
 * scratch: VSR_SN_SCRATCH_FLOATS floats of device memory (per-workgroup partial sums; fixed summation order).        */
#define VSR_SN_SCRATCH_FLOATS 1024
int vsr_spectral_norm_backward(const float* dw, const float* w_orig, const float* u, const float* v, const float* sigma,
                               float* dw_orig, int rows, int cols, float* scratch, void* stream);

/* AdversarialLoss's core (core/losses.py:66-74): loss[0] = mean BCE-with-logits(x, target); dx (optional) = its gradient */
int vsr_bce_with_logits(const float* x, float target, float* dx, float* loss, long long numel, void* stream);

/* ---- VRT window attention (BASELINE config 5; csrc/window_attention.hip) ------------------------------------------
 * The core of WindowAttention.attention (vsr/models/VRT/modules/window_attention.py:140-162):
 *     out = softmax((q * scale) k^T [+ bias[head]] [+ mask[window % nW]]) v
 * for every (window b, head h), fused on the matrix cores (no N x N tensor in HBM).  qkv is the output of the qkv Linear,
 * (B, N, 3, heads, head_dim) contiguous in `dtype`; queries are tokens [q0, q0+Nq), keys / values tokens [k0, k0+Nk)
 * (self attention: both all N tokens; mutual attention of a 2-frame window: the two halves, :128-134), and the result
 * goes to rows [o0, o0+Nq), channels [c_off, c_off + heads*head_dim) of out (B, N, Cout) -- where the reference's torch.cat
 * (:131-134) would put it.  head_dim <= 32; Nq, Nk multiples of 32, <= 384.
 * bias: dense fp32 (heads, Nq, Nk) or NULL (vsr_rpb_gather builds it from relative_position_bias_table, :146-148);
 * mask: fp32 (nW, Nm, Nm) from compute_mask (:61-77) or NULL -- its top-left Nq x Nk block is used, as the reference does.  */
typedef struct VsrAttnDesc {
    int B, N, heads, head_dim;
    int q0, k0, o0, Nq, Nk;
    int Cout, c_off;
    int nW, Nm;
    float scale;
    int dtype;
    int mask_packed;     /* 1: `mask` is vsr_mask_pack's bit-packed form (uint32 [nW][Nm][Nm/32], bit = entry != 0) and */
    float mask_value;    /*    every non-zero entry equals mask_value (-100 for compute_mask); Nq = Nk in {64, 128} only */
} VsrAttnDesc;
int vsr_mask_pack(const float* mask, unsigned* bits, int nW, int Nm, void* stream);
int vsr_window_attention_fwd(const VsrAttnDesc* d, const void* qkv, const float* bias, const float* mask, void* out,
                             float* lse /* (B, heads, Nq) fp32, needed by the backward; may be NULL */, void* stream);
/* dqkv (layout of qkv): d/dq for tokens [q0, q0+Nq) and d/dk, d/dv for tokens [k0, k0+Nk) are WRITTEN (other elements
 * untouched); dbias (heads, Nq, Nk) fp32 or NULL is ACCUMULATED into (atomics); delta: scratch (B, heads, Nq) fp32.   */
int vsr_window_attention_bwd(const VsrAttnDesc* d, const void* qkv, const float* bias, const float* mask, const void* dout,
                             const float* lse, float* delta, void* dqkv, float* dbias, void* stream);
/* dense[h][i][j] = table[index[i*idx_stride + j]][h] (index: int64, the module's relative_position_index) / its adjoint */
int vsr_rpb_gather(const float* table, const long long* index, int idx_stride, float* dense, int heads, int N, void* stream);
int vsr_rpb_scatter(const float* ddense, const long long* index, int idx_stride, float* dtable, int heads, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSRLAB_HIP_H */
