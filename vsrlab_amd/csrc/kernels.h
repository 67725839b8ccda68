// Internal C++ launch interface between the engine / C-ABI layer and the kernel translation units.
#pragma once
#include "common.h"

int vsr_launch_conv(int dtype, int ks, int nsrc, int ca, int cb, int last_planar, int cout_t, int epi,
                    const ConvArgs& a, hipStream_t st);
int vsr_launch_sign_bits_c64(const void* x_pm, void* bits, int N, int H, int W, hipStream_t st);     // conv3x3_persist.hip

// ---- a chain of dependent 3x3 64 -> 64 layers in one launch (conv3x3_chain.hip) ----
// Layer l reads `src`, writes `dst` (both whole blocked images of N x H x W x 64 bf16, each buffer written by exactly one layer of
// the chain and read only by later layers).  Offsets are in units of 256 bytes from ChainArgs::base; 0xffffffff = none.
enum { CHAIN_RELU = 0,     // dst = relu(conv(src) + bias), sign bits of dst -> sout (conv1 of a ResidualConv, conv.py:91)
       CHAIN_SKIP = 1,     // dst = conv(src) + bias + res            (conv2 + identity, conv.py:92; dgrad(conv1) + dX)
       CHAIN_MASK = 2 };   // dst = conv(src) * bit(sbits)            (dgrad(conv2) * ReLU')
#define VSR_CHAIN_MAX_LAYERS 64
struct ChainLayer { unsigned src, dst, res, sbits, sout, w, bias; int variant; };
struct ChainArgs {
    char* base;
    unsigned* sync;              // vsr_chain_sync_bytes(): work counter, error word, one flag word per (layer, tile, MFMA wave); zeroed by the launcher
    int N, H, W, nlayers;
    ChainLayer layer[VSR_CHAIN_MAX_LAYERS];
};
size_t vsr_chain_sync_bytes(int nlayers, int N, int H, int W);
int vsr_launch_conv3x3_chain(const ChainArgs& a, int num_cus, hipStream_t st);
void vsr_wgrad_slab_dims(int ks, int cx, int cout, int* coutp, int* cxp, int* stride);
int vsr_launch_wgrad(int dtype, int ks, int cx, int x_planar, int cout, int dy_planar, const WgradArgs& a, int nwg,
                     int* nslabs, hipStream_t st);      // *nslabs: partial slabs written (what the reduction sums)
int vsr_launch_wgrad_reduce(const float* slab, int nwg, int ks, int cx, int cout, int cout_real, int cin_real, float* gw,
                            int I_total, int i_off, int o_mul, int o_add, float* gb, int accumulate, hipStream_t st);
// border != 0: grid_sample padding_mode='border' (clamped coordinates) instead of 'zeros'
int vsr_launch_warp_fwd(int dtype, const void* in, const float* flow, void* out, int N, int H, int W, int C,
                        long long flow_nstride, hipStream_t st, int border = 0);
int vsr_launch_warp_bwd(int dtype, const void* dout, const float* flow, float* dacc, int N, int H, int W, int C,
                        long long flow_nstride, hipStream_t st, int border = 0);
// S: 64-bit fixed-point accumulator [N][H][W][64] (all-zero on entry and on exit)
int vsr_launch_warp_bwd_gather(int dtype, const void* dout, const float* flow, const void* dtop, long long* S, int* far_count, void* out,
                               int N, int H, int W, long long flow_nstride, hipStream_t st);
int vsr_launch_add_cast(int dtype, const void* a, const float* s, void* out, int N, int H, int W, int C, hipStream_t st);
int vsr_launch_planar_to_pm(int dtype, const float* in, void* out, int N, int Cin, int H, int W, int C, hipStream_t st);
int vsr_launch_pm_to_planar(int dtype, const void* in, float* out, int N, int Cout, int H, int W, int C, hipStream_t st);
int vsr_launch_resize_norm(const float* in, float* out, const float* mean, const float* std, int F, int h, int w, int hu,
                           int wu, hipStream_t st);
int vsr_launch_avgpool2(const float* in, float* out, long long planes, int h, int w, hipStream_t st);
int vsr_launch_spynet_prepare(int dtype, const float* frames, const float* flow_prev, float* flow_up, void* x16, int n,
                              int t, int P, int pair_mode, int h, int w, int level0, hipStream_t st);
int vsr_launch_flow_out(const float* in, float* out, int P, int hu, int wu, int h, int w, hipStream_t st);
int vsr_launch_warp_bwd_flow(int dtype, const void* in, const void* dout, const float* flow, float* dflow, int N, int H, int W,
                             int C, long long flow_nstride, hipStream_t st, int border = 0);
int vsr_launch_spynet_dres(int dtype, const float* dflow, const float* res, void* out, int P, int h, int w, hipStream_t st);
int vsr_launch_spynet_prepare_bwd(int dtype, const void* dx16, const float* dflow_l, const float* frames, const float* flow_up,
                                  float* dflow_prev, float* dframes, int n, int t, int P, int pair_mode, int h, int w, hipStream_t st);
int vsr_launch_avgpool2_bwd_add(const float* dcoarse, float* dfine, long long planes, int h, int w, hipStream_t st);
int vsr_launch_resize_norm_bwd(const float* dnorm, float* dframes, const float* std, int F, int h, int w, int hu, int wu, hipStream_t st);
int vsr_launch_bilinear4_bwd(const float* dsr, float* dlr, long long planes, int h, int w, hipStream_t st, int scale = 4);
int vsr_launch_flow_out_bwd(const float* dout, float* din, int P, int hu, int wu, int h, int w, hipStream_t st);
int vsr_launch_add_f32(const float* a, const float* b, float* out, long long n, hipStream_t st);
int vsr_launch_c64_to_planar(const ConvArgs& a, hipStream_t st);
int vsr_launch_last2_wgrad(const void* x, const float* dy, long long dy_nstride, float* slab, int slab_stride, int N, int H, int W,
                           int* nslabs, hipStream_t st, int pc = 3);      // pc: planes of the planar cotangent (1..3)
int vsr_launch_last2_dgrad(const float* dsr, long long dsr_nstride, const float* w, const void* aux, void* dst, int N, int H, int W,
                           int mask_mode, hipStream_t st, const void* sign_bits = nullptr, float slope = 0.f);
// One weight pack of vsr_launch_pack_weights as data: a forward packs ~465 tensors, batched into a handful of launches
// (vsr_launch_pack_multi: descriptors travel in the kernel arguments, VSR_PACK_BATCH per launch).
struct VsrPackDesc {
    const float* w; void* dst;
    int total, blk0;                                       // elements; first 256-thread block of this tensor within its launch
    short KK, RP, CPd, r_real, c_real, I_total, i_off, o_mul, o_add;
    unsigned char mode, dtype;
};
#define VSR_PACK_BATCH 80
int vsr_launch_pack_multi(const VsrPackDesc* descs, int n, hipStream_t st);
int vsr_launch_pack_weights(int dtype, const float* w, void* dst, int KK, int RP, int CPd, int r_real, int c_real,
                            int I_total, int i_off, int o_mul, int o_add, int mode, hipStream_t st);
int vsr_charbonnier_scratch_floats_impl();
int vsr_launch_charbonnier_grad(const float* sr, const float* hr, float* dsr, float* loss_acc, float* scratch, long long n, float eps,
                                hipStream_t st);

// ---- GAN side (conv_wide.hip) ----
struct VsrWideConv {
    const void* x; int xC, Hx, Wx, in_step, nsl;
    int N, H, W;
    const void* wpack; const float* bias;
    void* y; int yC, Hy, Wy, out_step, ncob;
    int act; float slope;
    void* y_act; const void* res; void* y_pre; const void* aux;
    const void* wpack2;            // bf16, yC a multiple of 128: weights in conv_wide2's image (vsr_launch_pack_wide2); wpack unused
};
int vsr_launch_conv_wide(int dtype, const VsrWideConv& c, hipStream_t st);
long long vsr_wide_pack_elems(int cout, int cin, int mode);
long long vsr_wide2_pack_elems(int cout, int cin, int mode);
int vsr_launch_pack_wide2(const float* w, void* dst, int cout, int cin, int mode, hipStream_t st);
int vsr_launch_pack_wide(int dtype, const float* w, void* dst, int cout, int cin, int mode, hipStream_t st);
int vsr_launch_wgrad_reduce_s2(const float* slab, int nwg, int slab_stride, float* gw, int cin_total, int co0, int ci0, int view, hipStream_t st);
int vsr_launch_up2_fwd(int dtype, const void* a, const void* b, void* out, int N, int H, int W, int C, hipStream_t st);
int vsr_launch_up2_bwd(int dtype, const void* dout, void* din, void* dmask, const void* m, float slope, int N, int H, int W, int C, hipStream_t st);
int vsr_launch_mask_pm(int dtype, const void* g, const void* m, void* out, float slope, long long elems, hipStream_t st);
int vsr_launch_add_pm(int dtype, const void* a, const void* b, void* out, long long elems, hipStream_t st);

#define VSR_WGRAD_MAX_PAIR_SLABS 1024   // pair-batched wgrad launches: (view, cout block, cin slice) pairs x pixel parts
int vsr_launch_wgrad_pairs(int dtype, const WgradArgs& a, int nsl, int ncob, int views, int ksplit, float* gw, int cin_total, hipStream_t st);
#define VSR_WGRAD_NWG 512   // persistent wgrad workgroups: 2 per CU x 256 CUs
