// UNetDiscriminator forward / backward scheduler (BASELINE config 3).
// Reference: vsr/models/RealBasicVSR/modules/unet-discriminator.py:4-31.  The caller passes EFFECTIVE conv weights
// (already divided by the spectral norm: vsr_spectral_norm, core/modules/conv.py:6-13) and gets gradients w.r.t. them;
// the spectral-norm chain rule is vsr_spectral_norm_backward.  Like engine.hip: one caller-provided workspace, a plan
// that is a pure function of the descriptor, work only enqueued on the given stream.
//
//   f0 = lrelu(conv_0(img))                 (H,   64)     3x3, bias          conv_mfma (planar source)
//   f1 = lrelu(conv_1(f0))                  (H/2, 128)    4x4 stride 2       conv_wide, parity views
//   f2 = lrelu(conv_2(f1))                  (H/4, 256)
//   f3 = lrelu(conv_3(f2))                  (H/8, 512)
//   u3 = up2(f3)                            (H/4, 512)
//   a4 = lrelu(conv_4(u3)); u4 = up2(a4 + f2)   (H/2, 256)   3x3, 8 slices
//   a5 = lrelu(conv_5(u4)); u5 = up2(a5 + f1)   (H,   128)
//   a6 = lrelu(conv_6(u5)); s6 = a6 + f0        (H,   64)
//   o7 = lrelu(conv_7(s6)); o8 = lrelu(conv_8(o7))           conv3x3_persist (64 -> 64)
//   out = conv_9(o8)                        (H, 1) planar fp32, bias
// The pre-residual activations a4, a5, a6 are kept besides the sums: LeakyReLU' needs the sign of the activation, which
// the sum with the skip no longer has.
#include <vector>
#include "kernels.h"
#include "../../include/vsrlab_hip.h"

namespace {

constexpr float SLOPE = 0.2f;      // unet-discriminator.py:19
constexpr int C = 64;

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};
inline size_t esize(int dtype) { return dtype == VSR_BF16 ? 2 : 4; }

#define CK(expr) do { int _s = (expr); if (_s != VSR_OK) return _s; } while (0)

struct DPlan {
    int n, h, w, dtype; size_t es;
    // packed weights
    size_t w0, b0, w0d, w9, b9, w9d, w7, w7d, w8, w8d;
    size_t wf[7], wd[7];                 // conv_1 .. conv_6 forward / data-gradient packs
    bool fw2[7], dw2[7];                   // ... in conv_wide2's image
    // activations
    size_t f0, f1, f2, f3, u3, a4, u4, a5, u5, a6, s6, o7, o8;
    // backward scratch
    size_t d_o8, d_o7, G6, dc6, du5, ds5, dc5, du4, ds4, dc4, du3, dc3, dc2, dc1, dc0, slab;
    size_t total;

    size_t pm(int H, int W, int Cc) const { return (size_t)n * pm_image_elems(H, W, Cc) * es; }
    size_t pm1(int H, int W, int Cc) const { return (size_t)pm_image_elems(H, W, Cc) * es; }      // one frame
    int build(const VsrDiscDesc& d, int need_backward) {
        n = d.n; h = d.h; w = d.w; dtype = d.dtype;
        if (d.mid_ch != C || n < 1 || h < 8 || w < 8 || (h & 7) || (w & 7)) return VSR_ERR_UNSUPPORTED;
        if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
        es = esize(dtype);
        Bump b;
        const size_t w64 = (size_t)9 * C * C * es;
        w0 = b.take((size_t)9 * C * 16 * es); b0 = b.take(C * 4); w0d = b.take((size_t)9 * 32 * C * es);
        w9 = b.take((size_t)9 * 32 * C * es); b9 = b.take(64 * 4); w9d = b.take((size_t)9 * C * 16 * es);
        w7 = b.take(w64); w7d = b.take(w64); w8 = b.take(w64); w8d = b.take(w64);
        const int co[7] = {0, 128, 256, 512, 256, 128, 64}, ci[7] = {0, 64, 128, 256, 512, 256, 128};
        for (int k = 1; k <= 6; ++k) {
            const bool s2 = k <= 3;
            // bf16 runs on conv_wide2 (its own weight image; 64-channel outputs: the K-split form)
            fw2[k] = dw2[k] = dtype == VSR_BF16;
            wf[k] = b.take((size_t)(fw2[k] ? vsr_wide2_pack_elems(co[k], ci[k], s2 ? 1 : 0) : vsr_wide_pack_elems(co[k], ci[k], s2 ? 1 : 0)) * es);
            wd[k] = b.take((size_t)(dw2[k] ? vsr_wide2_pack_elems(co[k], ci[k], s2 ? 3 : 2) : vsr_wide_pack_elems(co[k], ci[k], s2 ? 3 : 2)) * es);
        }
        f0 = b.take(pm(h, w, 64)); f1 = b.take(pm(h / 2, w / 2, 128)); f2 = b.take(pm(h / 4, w / 4, 256)); f3 = b.take(pm(h / 8, w / 8, 512));
        u3 = b.take(pm(h / 4, w / 4, 512)); a4 = b.take(pm(h / 4, w / 4, 256)); u4 = b.take(pm(h / 2, w / 2, 256));
        a5 = b.take(pm(h / 2, w / 2, 128)); u5 = b.take(pm(h, w, 128)); a6 = b.take(pm(h, w, 64)); s6 = b.take(pm(h, w, 64));
        o7 = b.take(pm(h, w, 64)); o8 = b.take(pm(h, w, 64));
        if (need_backward) {
            // backward scratch for ONE frame: the backward walks the batch frame by frame (weight gradients accumulate), so a
            // 7-frame 2160x3840 batch keeps 76 GB of activations but only 8.5 GB of scratch (not 60)
            d_o8 = b.take(pm1(h, w, 64)); d_o7 = b.take(pm1(h, w, 64)); G6 = b.take(pm1(h, w, 64)); dc6 = b.take(pm1(h, w, 64));
            du5 = b.take(pm1(h, w, 128)); ds5 = b.take(pm1(h / 2, w / 2, 128)); dc5 = b.take(pm1(h / 2, w / 2, 128));
            du4 = b.take(pm1(h / 2, w / 2, 256)); ds4 = b.take(pm1(h / 4, w / 4, 256)); dc4 = b.take(pm1(h / 4, w / 4, 256));
            du3 = b.take(pm1(h / 4, w / 4, 512)); dc3 = b.take(pm1(h / 8, w / 8, 512)); dc2 = b.take(pm1(h / 4, w / 4, 256));
            dc1 = b.take(pm1(h / 2, w / 2, 128)); dc0 = b.take(pm1(h, w, 64));
            int cp, xp, stride;
            vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
            slab = b.take((size_t)(VSR_WGRAD_NWG > VSR_WGRAD_MAX_PAIR_SLABS ? VSR_WGRAD_NWG : VSR_WGRAD_MAX_PAIR_SLABS) * stride * 4);
        }
        total = b.off;
        return VSR_OK;
    }
};

// the plan as seen by the backward of frame f: batch 1, activation offsets moved to that frame
DPlan frame_view(const DPlan& p, int f) {
    DPlan q = p;
    q.n = 1;
    const int h = p.h, w = p.w;
    auto sh = [&](size_t& off, int H, int W, int Cc) { off += (size_t)f * pm_image_elems(H, W, Cc) * p.es; };
    sh(q.f0, h, w, 64); sh(q.f1, h / 2, w / 2, 128); sh(q.f2, h / 4, w / 4, 256); sh(q.f3, h / 8, w / 8, 512);
    sh(q.u3, h / 4, w / 4, 512); sh(q.a4, h / 4, w / 4, 256); sh(q.u4, h / 2, w / 2, 256); sh(q.a5, h / 2, w / 2, 128);
    sh(q.u5, h, w, 128); sh(q.a6, h, w, 64); sh(q.s6, h, w, 64); sh(q.o7, h, w, 64); sh(q.o8, h, w, 64);
    return q;
}

struct DCtx {
    const DPlan& p; char* ws; hipStream_t st; int dtype;
    void* at(size_t o) const { return ws + o; }
    const float* fat(size_t o) const { return reinterpret_cast<const float*>(ws + o); }

    ConvArgs base(int H, int W) const {
        ConvArgs a = {};
        a.in_step = 1; a.Hs = H; a.Ws = W; a.N = p.n; a.H = H; a.W = W; a.nz = 1;
        a.out_step = 1; a.Hd = H; a.Wd = W; a.CD = C; a.cout_real = C; a.dst_nstride = pm_image_elems(H, W, C);
        a.src_nstride[0] = pm_image_elems(H, W, C);
        a.leaky_slope = SLOPE;
        return a;
    }
    // 64 -> 64 at full resolution on the persistent kernel
    int conv64(const void* x, size_t wpack, void* y, int act, const void* aux, int mask) const {
        ConvArgs a = base(p.h, p.w);
        a.src[0] = x; a.wpack = at(wpack); a.dst[0] = y; a.act = act; a.aux[0] = aux; a.mask_mode = mask;
        return vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
    }
    int wide(const void* x, int xC, int Hx, int Wx, int in_step, int H, int W, size_t wpack, bool w2, void* y, int yC, int out_step, int act,
             void* y_act = nullptr, const void* res = nullptr, void* y_pre = nullptr, const void* aux = nullptr) const {
        VsrWideConv c = {};
        c.x = x; c.xC = xC; c.Hx = Hx; c.Wx = Wx; c.in_step = in_step; c.nsl = xC / 64;
        c.N = p.n; c.H = H; c.W = W; c.bias = nullptr;
        if (w2) c.wpack2 = at(wpack); else c.wpack = at(wpack);
        c.y = y; c.yC = yC; c.Hy = H * out_step; c.Wy = W * out_step; c.out_step = out_step; c.ncob = yC / 64;
        c.act = act; c.slope = SLOPE; c.y_act = y_act; c.res = res; c.y_pre = y_pre; c.aux = aux;
        return vsr_launch_conv_wide(dtype, c, st);
    }
    // one 64 x 64 block of a weight gradient: slabs -> reduce.  view < 0: 3x3 layer (OIHW with 9 taps); else parity view of a 4x4 layer
    int wgrad_block(const void* x, int xC, int x_slice, int Hx, int Wx, int x_step, int view, const void* dy, int dyC, int dy_slice,
                    int H, int W, float* gw, int cin_total, int co0, int ci0) const {
        WgradArgs a = {};
        a.N = p.n; a.H = H; a.W = W; a.nseg = 1;
        a.x[0] = x; a.x_step = x_step; a.x_oy = view >= 0 ? (view >> 1) : 0; a.x_ox = view >= 0 ? (view & 1) : 0;
        a.Hx = Hx; a.Wx = Wx; a.x_nstride = pm_image_elems(Hx, Wx, xC); a.x_ctotal = xC; a.x_coff = x_slice * 8;
        a.dy[0] = dy; a.dy_step = 1; a.Hy = H; a.Wy = W; a.dy_nstride = pm_image_elems(H, W, dyC); a.dy_ctotal = dyC; a.dy_coff = dy_slice * 8;
        int cp, xp, stride;
        vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
        a.slab = (float*)at(p.slab); a.slab_stride = stride;
        const int tiles = p.n * cdiv(H, 8) * cdiv(W, 32);
        int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
        if (nwg > 1) nwg &= ~1;
        int nslabs = 0;
        CK(vsr_launch_wgrad(dtype, 3, 64, 0, 64, 0, a, nwg, &nslabs, st));
        if (view < 0)
            return vsr_launch_wgrad_reduce((const float*)at(p.slab), nslabs, 3, 64, 64, 64, 64, gw + (size_t)co0 * cin_total * 9, cin_total, ci0, 1, 0,
                                           nullptr, 1, st);
        return vsr_launch_wgrad_reduce_s2((const float*)at(p.slab), nslabs, stride, gw, cin_total, co0, ci0, view, st);
    }
    // the whole weight gradient of a wide layer: x (xC channels; views == 4: the four parity views of a 4x4 stride-2 conv's input),
    // dy (dyC channels).  bf16: every (view, cout block, cin slice) pair in one launch + one reduction; fp32: pair by pair.
    int wgrad_layer(const void* x, int xC, int Hx, int Wx, int views, const void* dy, int dyC, int H, int W, float* gw) const {
        const int nsl = xC / 64, ncob = dyC / 64, npairs = nsl * ncob * views, x_step = views == 4 ? 2 : 1;
        if (dtype == VSR_BF16 && npairs > 1) {
            WgradArgs a = {};
            a.N = p.n; a.H = H; a.W = W; a.nseg = 1;
            a.x[0] = x; a.x_step = x_step; a.Hx = Hx; a.Wx = Wx; a.x_nstride = pm_image_elems(Hx, Wx, xC); a.x_ctotal = xC;
            a.dy[0] = dy; a.dy_step = 1; a.Hy = H; a.Wy = W; a.dy_nstride = pm_image_elems(H, W, dyC); a.dy_ctotal = dyC;
            int cp, xp, stride;
            vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
            a.slab = (float*)at(p.slab); a.slab_stride = stride;
            const int tiles = p.n * cdiv(H, 8) * cdiv(W, 32);
            int ksplit = ((512 / npairs) + 7) & ~7;                       // ~512 workgroups per launch
            if (ksplit < 8) ksplit = 8;
            const int cap = (tiles + 7) & ~7;
            if (ksplit > cap) ksplit = cap;
            return vsr_launch_wgrad_pairs(dtype, a, nsl, ncob, views, ksplit, gw, xC, st);
        }
        for (int v = 0; v < views; ++v) for (int o = 0; o < ncob; ++o) for (int s = 0; s < nsl; ++s)
            CK(wgrad_block(x, xC, s, Hx, Wx, x_step, views == 4 ? v : -1, dy, dyC, o, H, W, gw, xC, 64 * o, 64 * s));
        return VSR_OK;
    }
};

int disc_pack(const DCtx& c, const float* const* prm, bool bwd) {
    const DPlan& p = c.p;
    const int dt = c.dtype;
    // conv_0: 3 -> 64 on the planar image (16-channel padded source) ; its data gradient 64 -> 3 planar
    CK(vsr_launch_pack_weights(dt, prm[0], c.at(p.w0), 9, C, 16, C, 3, 3, 0, 1, 0, 0, c.st));
    CK(vsr_launch_pack_weights(VSR_F32, prm[1], c.at(p.b0), 1, C, 1, C, 1, 1, 0, 1, 0, 0, c.st));
    if (bwd) CK(vsr_launch_pack_weights(dt, prm[0], c.at(p.w0d), 9, 32, C, 3, C, 3, 0, 1, 0, 1, c.st));
    const int co[7] = {0, 128, 256, 512, 256, 128, 64}, ci[7] = {0, 64, 128, 256, 512, 256, 128};
    for (int k = 1; k <= 6; ++k) {
        const bool s2 = k <= 3;
        if (p.fw2[k]) CK(vsr_launch_pack_wide2(prm[1 + k], c.at(p.wf[k]), co[k], ci[k], s2 ? 1 : 0, c.st));
        else CK(vsr_launch_pack_wide(dt, prm[1 + k], c.at(p.wf[k]), co[k], ci[k], s2 ? 1 : 0, c.st));
        if (bwd) {
            if (p.dw2[k]) CK(vsr_launch_pack_wide2(prm[1 + k], c.at(p.wd[k]), co[k], ci[k], s2 ? 3 : 2, c.st));
            else CK(vsr_launch_pack_wide(dt, prm[1 + k], c.at(p.wd[k]), co[k], ci[k], s2 ? 3 : 2, c.st));
        }
    }
    CK(vsr_launch_pack_weights(dt, prm[8], c.at(p.w7), 9, C, C, C, C, C, 0, 1, 0, 0, c.st));
    CK(vsr_launch_pack_weights(dt, prm[9], c.at(p.w8), 9, C, C, C, C, C, 0, 1, 0, 0, c.st));
    if (bwd) {
        CK(vsr_launch_pack_weights(dt, prm[8], c.at(p.w7d), 9, C, C, C, C, C, 0, 1, 0, 1, c.st));
        CK(vsr_launch_pack_weights(dt, prm[9], c.at(p.w8d), 9, C, C, C, C, C, 0, 1, 0, 1, c.st));
    }
    // conv_9: 64 -> 1, planar destination (32-row template) ; its data gradient: planar 1-channel source -> 64
    CK(vsr_launch_pack_weights(dt, prm[10], c.at(p.w9), 9, 32, C, 1, C, C, 0, 1, 0, 0, c.st));
    CK(vsr_launch_pack_weights(VSR_F32, prm[11], c.at(p.b9), 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, c.st));
    if (bwd) CK(vsr_launch_pack_weights(dt, prm[10], c.at(p.w9d), 9, C, 16, C, 1, C, 0, 1, 0, 1, c.st));
    return VSR_OK;
}

int disc_forward(const DCtx& c, const float* img, float* out) {
    const DPlan& p = c.p;
    const int h = p.h, w = p.w, n = p.n, dt = c.dtype;
    {   // conv_0 + LeakyReLU(0.2)
        ConvArgs a = c.base(h, w);
        a.src[0] = img; a.src_nstride[0] = (long long)3 * h * w; a.wpack = c.at(p.w0); a.bias = c.fat(p.b0); a.dst[0] = c.at(p.f0); a.act = ACT_LEAKY;
        CK(vsr_launch_conv(dt, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, c.st));
    }
    CK(c.wide(c.at(p.f0), 64, h, w, 2, h / 2, w / 2, p.wf[1], p.fw2[1], c.at(p.f1), 128, 1, ACT_LEAKY));
    CK(c.wide(c.at(p.f1), 128, h / 2, w / 2, 2, h / 4, w / 4, p.wf[2], p.fw2[2], c.at(p.f2), 256, 1, ACT_LEAKY));
    CK(c.wide(c.at(p.f2), 256, h / 4, w / 4, 2, h / 8, w / 8, p.wf[3], p.fw2[3], c.at(p.f3), 512, 1, ACT_LEAKY));
    CK(vsr_launch_up2_fwd(dt, c.at(p.f3), nullptr, c.at(p.u3), n, h / 8, w / 8, 512, c.st));
    CK(c.wide(c.at(p.u3), 512, h / 4, w / 4, 1, h / 4, w / 4, p.wf[4], p.fw2[4], c.at(p.a4), 256, 1, ACT_LEAKY));
    CK(vsr_launch_up2_fwd(dt, c.at(p.a4), c.at(p.f2), c.at(p.u4), n, h / 4, w / 4, 256, c.st));
    CK(c.wide(c.at(p.u4), 256, h / 2, w / 2, 1, h / 2, w / 2, p.wf[5], p.fw2[5], c.at(p.a5), 128, 1, ACT_LEAKY));
    CK(vsr_launch_up2_fwd(dt, c.at(p.a5), c.at(p.f1), c.at(p.u5), n, h / 2, w / 2, 128, c.st));
    CK(c.wide(c.at(p.u5), 128, h, w, 1, h, w, p.wf[6], p.fw2[6], c.at(p.s6), 64, 1, ACT_LEAKY, c.at(p.a6), c.at(p.f0)));
    CK(c.conv64(c.at(p.s6), p.w7, c.at(p.o7), ACT_LEAKY, nullptr, 0));
    CK(c.conv64(c.at(p.o7), p.w8, c.at(p.o8), ACT_LEAKY, nullptr, 0));
    {   // conv_9: 64 -> 1 logits, planar fp32
        ConvArgs a = c.base(h, w);
        a.src[0] = c.at(p.o8); a.wpack = c.at(p.w9); a.bias = c.fat(p.b9); a.cout_real = 1;
        a.dst[0] = out; a.dst_nstride = (long long)h * w;
        CK(vsr_launch_conv(dt, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, c.st));
    }
    return VSR_OK;
}

int disc_backward(const DCtx& c, float* const* g, const float* img, const float* dout, float* dimg) {
    const DPlan& p = c.p;
    const int h = p.h, w = p.w, n = p.n, dt = c.dtype;
    int cp, xp, stride;
    // ---- conv_9 ----
    {   // d o8 = dgrad(conv_9)(dout) * LeakyReLU'(o8)
        ConvArgs a = c.base(h, w);
        a.src[0] = dout; a.src_nstride[0] = (long long)h * w; a.planar_c = 1; a.wpack = c.at(p.w9d); a.dst[0] = c.at(p.d_o8);
        a.aux[0] = c.at(p.o8); a.mask_mode = MASK_LEAKY;
        CK(vsr_launch_conv(dt, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, c.st));
    }
    if (g[10]) {
        WgradArgs a = {};
        a.N = n; a.H = h; a.W = w; a.nseg = 1; a.x[0] = c.at(p.o8); a.x_step = 1; a.Hx = h; a.Wx = w; a.x_nstride = pm_image_elems(h, w, C);
        a.dy[0] = dout; a.dy_step = 1; a.Hy = h; a.Wy = w; a.dy_nstride = (long long)h * w; a.dy_planar_c = 1;
        vsr_wgrad_slab_dims(3, 64, 16, &cp, &xp, &stride);
        a.slab = (float*)c.at(p.slab); a.slab_stride = stride;
        const int tiles = n * cdiv(h, 8) * cdiv(w, 32);
        int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
        int nslabs = 0;
        CK(vsr_launch_wgrad(dt, 3, 64, 0, 16, 1, a, nwg, &nslabs, c.st));
        CK(vsr_launch_wgrad_reduce((const float*)c.at(p.slab), nslabs, 3, 64, 16, 1, C, g[10], C, 0, 1, 0, g[11], 1, c.st));
    } else if (g[11]) {
        return VSR_ERR_BADARG;
    }
    // ---- conv_8, conv_7 (64 -> 64) ----
    CK(c.conv64(c.at(p.d_o8), p.w8d, c.at(p.d_o7), ACT_NONE, c.at(p.o7), MASK_LEAKY));
    if (g[9]) CK(c.wgrad_block(c.at(p.o7), 64, 0, h, w, 1, -1, c.at(p.d_o8), 64, 0, h, w, g[9], 64, 0, 0));
    CK(c.conv64(c.at(p.d_o7), p.w7d, c.at(p.G6), ACT_NONE, nullptr, 0));                      // d s6: also the skip gradient of f0
    if (g[8]) CK(c.wgrad_block(c.at(p.s6), 64, 0, h, w, 1, -1, c.at(p.d_o7), 64, 0, h, w, g[8], 64, 0, 0));
    CK(vsr_launch_mask_pm(dt, c.at(p.G6), c.at(p.a6), c.at(p.dc6), SLOPE, (long long)n * pm_image_elems(h, w, 64), c.st));
    // ---- conv_6 (128 -> 64 at H) ----
    CK(c.wide(c.at(p.dc6), 64, h, w, 1, h, w, p.wd[6], p.dw2[6], c.at(p.du5), 128, 1, ACT_NONE));
    if (g[7]) CK(c.wgrad_layer(c.at(p.u5), 128, h, w, 1, c.at(p.dc6), 64, h, w, g[7]));
    CK(vsr_launch_up2_bwd(dt, c.at(p.du5), c.at(p.ds5), c.at(p.dc5), c.at(p.a5), SLOPE, n, h / 2, w / 2, 128, c.st));
    // ---- conv_5 (256 -> 128 at H/2) ----
    CK(c.wide(c.at(p.dc5), 128, h / 2, w / 2, 1, h / 2, w / 2, p.wd[5], p.dw2[5], c.at(p.du4), 256, 1, ACT_NONE));
    if (g[6]) CK(c.wgrad_layer(c.at(p.u4), 256, h / 2, w / 2, 1, c.at(p.dc5), 128, h / 2, w / 2, g[6]));
    CK(vsr_launch_up2_bwd(dt, c.at(p.du4), c.at(p.ds4), c.at(p.dc4), c.at(p.a4), SLOPE, n, h / 4, w / 4, 256, c.st));
    // ---- conv_4 (512 -> 256 at H/4) ----
    CK(c.wide(c.at(p.dc4), 256, h / 4, w / 4, 1, h / 4, w / 4, p.wd[4], p.dw2[4], c.at(p.du3), 512, 1, ACT_NONE));
    if (g[5]) CK(c.wgrad_layer(c.at(p.u3), 512, h / 4, w / 4, 1, c.at(p.dc4), 256, h / 4, w / 4, g[5]));
    CK(vsr_launch_up2_bwd(dt, c.at(p.du3), nullptr, c.at(p.dc3), c.at(p.f3), SLOPE, n, h / 8, w / 8, 512, c.st));
    // ---- conv_3 (256 -> 512, 4x4 stride 2): d c2 = (dgrad + d s4) * LeakyReLU'(f2) ----
    CK(c.wide(c.at(p.dc3), 512, h / 8, w / 8, 1, h / 8, w / 8, p.wd[3], p.dw2[3], c.at(p.dc2), 256, 2, ACT_NONE, nullptr, c.at(p.ds4), nullptr, c.at(p.f2)));
    if (g[4]) CK(c.wgrad_layer(c.at(p.f2), 256, h / 4, w / 4, 4, c.at(p.dc3), 512, h / 8, w / 8, g[4]));
    // ---- conv_2 (128 -> 256) ----
    CK(c.wide(c.at(p.dc2), 256, h / 4, w / 4, 1, h / 4, w / 4, p.wd[2], p.dw2[2], c.at(p.dc1), 128, 2, ACT_NONE, nullptr, c.at(p.ds5), nullptr, c.at(p.f1)));
    if (g[3]) CK(c.wgrad_layer(c.at(p.f1), 128, h / 2, w / 2, 4, c.at(p.dc2), 256, h / 4, w / 4, g[3]));
    // ---- conv_1 (64 -> 128) ----
    CK(c.wide(c.at(p.dc1), 128, h / 2, w / 2, 1, h / 2, w / 2, p.wd[1], p.dw2[1], c.at(p.dc0), 64, 2, ACT_NONE, nullptr, c.at(p.G6), nullptr, c.at(p.f0)));
    if (g[2]) CK(c.wgrad_layer(c.at(p.f0), 64, h, w, 4, c.at(p.dc1), 128, h / 2, w / 2, g[2]));
    // ---- conv_0 (3 -> 64 on the planar image) ----
    if (g[0]) {
        WgradArgs a = {};
        a.N = n; a.H = h; a.W = w; a.nseg = 1; a.x[0] = img; a.x_step = 1; a.Hx = h; a.Wx = w; a.x_nstride = (long long)3 * h * w;
        a.dy[0] = c.at(p.dc0); a.dy_step = 1; a.Hy = h; a.Wy = w; a.dy_nstride = pm_image_elems(h, w, C);
        vsr_wgrad_slab_dims(3, 16, 64, &cp, &xp, &stride);
        a.slab = (float*)c.at(p.slab); a.slab_stride = stride;
        const int tiles = n * cdiv(h, 8) * cdiv(w, 32);
        int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
        int nslabs = 0;
        CK(vsr_launch_wgrad(dt, 3, 16, 1, 64, 0, a, nwg, &nslabs, c.st));
        CK(vsr_launch_wgrad_reduce((const float*)c.at(p.slab), nslabs, 3, 16, 64, C, 3, g[0], 3, 0, 1, 0, g[1], 1, c.st));
    } else if (g[1]) {
        return VSR_ERR_BADARG;
    }
    if (dimg) {
        ConvArgs a = c.base(h, w);
        a.src[0] = c.at(p.dc0); a.wpack = c.at(p.w0d); a.cout_real = 3;
        a.dst[0] = dimg; a.dst_nstride = (long long)3 * h * w;
        CK(vsr_launch_conv(dt, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, c.st));
    }
    return VSR_OK;
}

}  // namespace

extern "C" {

size_t vsr_disc_workspace_bytes(const VsrDiscDesc* d, int need_backward) {
    if (!d) return 0;
    DPlan p;
    if (p.build(*d, need_backward) != VSR_OK) return 0;
    return p.total;
}

int vsr_disc_forward(const VsrDiscDesc* d, const float* const* params, int nparams, const float* img, float* out, void* workspace,
                     size_t workspace_bytes, int need_backward, void* stream) {
    if (!d || !params || nparams != 12 || !img || !out || !workspace) return VSR_ERR_BADARG;
    for (int k = 0; k < 12; ++k) if (!params[k]) return VSR_ERR_BADARG;
    DPlan p;
    CK(p.build(*d, need_backward));
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    const DCtx c{p, (char*)workspace, (hipStream_t)stream, p.dtype};
    CK(disc_pack(c, params, need_backward != 0));
    return disc_forward(c, img, out);
}

int vsr_disc_backward(const VsrDiscDesc* d, float* const* grads, int nparams, const float* img, const float* dout, float* dimg,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !grads || nparams != 12 || !img || !dout || !workspace) return VSR_ERR_BADARG;
    DPlan p;
    CK(p.build(*d, 1));
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    for (int f = 0; f < p.n; ++f) {
        const DPlan pf = frame_view(p, f);
        const DCtx c{pf, (char*)workspace, (hipStream_t)stream, p.dtype};
        CK(disc_backward(c, grads, img + (size_t)f * 3 * p.h * p.w, dout + (size_t)f * p.h * p.w, dimg ? dimg + (size_t)f * 3 * p.h * p.w : nullptr));
    }
    return VSR_OK;
}

}  // extern "C"
