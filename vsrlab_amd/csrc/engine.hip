// Whole-clip scheduler for the BasicVSR path: plans one workspace arena, then enqueues every
// kernel of BasicVSR.forward (basicvsr.py:39-83) and of its backward on one HIP
// stream.  No allocation, no synchronisation, no global state: the plan is a pure function of the
// descriptor, so forward and backward recompute identical offsets.
//
// Memory-rich by design (288 GB HBM3E): every trunk activation AND every trunk activation-gradient
// of the clip is retained, so each layer's weight gradient is ONE split-K launch over all t frames
// (see wgrad_mfma.hip) instead of t launches + t reductions.
#include <vector>
#include "kernels.h"
#include "../../include/vsrlab_hip.h"

namespace {

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~size_t(255);
        return o;
    }
};

inline size_t esize(int dtype) { return dtype == VSR_BF16 ? 2 : 4; }

constexpr int C = 64;           // mid channels of the HIP path
constexpr int NSPY = 5;         // convs per SPyNet level
const int SPY_CI[NSPY] = {8, 32, 64, 32, 16}, SPY_CO[NSPY] = {32, 64, 32, 16, 2};
const int SPY_CIP[NSPY] = {16, 32, 64, 32, 16}, SPY_COP[NSPY] = {32, 64, 32, 32, 32};   // padded (template) sizes
const int SPY_CD[NSPY] = {32, 64, 32, 16, 0};     // channels per pixel of each layer's pixel-major output
// backward (train_flow): pixel-major channels of each layer's dY, and the dgrad launch's template rows (COUT)
const int SPY_DK[NSPY] = {32, 64, 32, 16, 16}, SPY_DROWS[NSPY] = {32, 32, 64, 32, 32};

struct SpyPlan {
    int P, F, h, w, hu, wu;
    size_t pyr[6];          // planar fp32 normalised frames, level 0 = coarsest
    size_t x16, b32a, b64, b32b, b16;   // pixel-major T at the finest level size
    size_t flow_a, flow_b, flow_up;     // planar fp32 [P][2][hu][wu]
    size_t wpack[6][NSPY], bias[6][NSPY];
    // train_flow (need_backward = 2): per-level saved activations, dgrad weights and backward scratch
    bool save = false;
    size_t sx[6][NSPY];                 // inputs of the 5 convs of each level: x16, b32a, b64, b32b, b16
    size_t sfup[6], sres[6];            // planar fp32 flow_up and residue (= ReLU(conv5)) per level
    size_t wpackd[6][NSPY];
    size_t gA, gB, dres, dfa, dfb;
    size_t dpyr[6];                     // gradient of the normalised pyramid (input-frame gradient)
    void plan_save(Bump& b, int dtype) {
        save = true;
        const size_t es = esize(dtype);
        for (int l = 0; l < 6; ++l) {
            const int hl = hu >> (5 - l), wl = wu >> (5 - l);
            for (int j = 0; j < NSPY; ++j) sx[l][j] = b.take((size_t)P * pm_image_elems(hl, wl, SPY_CIP[j]) * es);
            sfup[l] = b.take((size_t)P * 2 * hl * wl * 4);
            sres[l] = b.take((size_t)P * 2 * hl * wl * 4);
            for (int j = 0; j < NSPY; ++j) wpackd[l][j] = b.take((size_t)49 * 64 * 64 * es);
        }
        gA = b.take((size_t)P * pm_image_elems(hu, wu, 64) * es);
        gB = b.take((size_t)P * pm_image_elems(hu, wu, 64) * es);
        dres = b.take((size_t)P * pm_image_elems(hu, wu, 16) * es);
        dfa = b.take((size_t)P * 2 * hu * wu * 4);
        dfb = b.take((size_t)P * 2 * hu * wu * 4);
        for (int l = 0; l < 6; ++l) dpyr[l] = b.take((size_t)F * 3 * (hu >> (5 - l)) * (wu >> (5 - l)) * 4);
    }
    void plan(Bump& b, int P_, int F_, int h_, int w_, int dtype) {
        P = P_; F = F_; h = h_; w = w_;
        wu = (w % 32) == 0 ? w : 32 * (w / 32 + 1);      // spynet.py:72-73
        hu = (h % 32) == 0 ? h : 32 * (h / 32 + 1);
        const size_t es = esize(dtype);
        for (int l = 0; l < 6; ++l) {
            const int s = 5 - l;
            pyr[l] = b.take((size_t)F * 3 * (hu >> s) * (wu >> s) * 4);
        }
        const size_t px = (size_t)P * hu * wu;
        auto pm = [&](int Cc) { return (size_t)P * pm_image_elems(hu, wu, Cc) * es; };     // blocked pixel-major tensors
        x16 = b.take(pm(16)); b32a = b.take(pm(32)); b64 = b.take(pm(64));
        b32b = b.take(pm(32)); b16 = b.take(pm(16));
        flow_a = b.take(px * 2 * 4); flow_b = b.take(px * 2 * 4); flow_up = b.take(px * 2 * 4);
        for (int l = 0; l < 6; ++l)
            for (int j = 0; j < NSPY; ++j) {
                wpack[l][j] = b.take((size_t)49 * SPY_COP[j] * SPY_CIP[j] * es);
                bias[l][j] = b.take(64 * 4);
            }
    }
};

struct Plan {
    VsrBasicVSRDesc d;
    bool bwd, flowgrad;          // flowgrad: train_flow (basicvsr.py:25-28), SPyNet is differentiated too
    bool diet;                   // VsrBasicVSRDesc.arena_mode = 1 (training only): see vsrlab_hip.h
    int rb, n, t, h, w, dtype;
    int scale, ups;              // upscale (2 or 4, basicvsr.py:12-23) and its PixelShufflePack count scale / 2
    bool unsh;                   // the gradients INTO the pixel-shuffle layers (G_U1, G_U0) are kept phase-separated (ConvArgs::unshuffle)
    size_t es;
    size_t px1;                 // elements of one blocked (n,h,w,64) tensor
    size_t s_elems;             // elements of the plain 64-bit fixed-point warp-scatter accumulator (n,h,w,64)
    // packed weights / biases
    size_t stem_w[2], stem_wd[2], stem_b[2];
    std::vector<size_t> blk_w[2], blk_wd[2], blk_b[2];     // [2*rb]: conv1, conv2 alternating
    size_t point_w, point_wd, point_b;
    size_t up_w[2], up_wd[2], up_b[2];
    size_t last0_w, last0_wd, last0_b, last2_w, last2_wd, last2_b;
    SpyPlan spy;
    size_t flows;               // fp32 planar [2*n*(t-1)][2][h][w]: first half backward, second half forward
    // trunk activations: [dir][frame]
    std::vector<size_t> Wp[2], X[2], A[2];      // X: (rb+1) per frame, A: rb per frame (saved mode)
    std::vector<size_t> SB[2];                  // sign bits of A (bf16 build): rb per frame, 8 bytes per lane and 8x32 tile
    std::vector<size_t> feat[2];                // = X[rb]
    size_t scratchA[2], scratchW[2];            // inference mode, per direction (the directions run concurrently)
    // reconstruction
    std::vector<size_t> Pt, U0, U1, C0;
    std::vector<size_t> SBC0;                   // sign bits of C0 = LeakyReLU(conv_last.0) per frame (bf16 build, training)
    std::vector<size_t> SBPt;                   // sign bits of Pt = LeakyReLU(point_conv) per frame (bf16 build, training)
    std::vector<size_t> SBX0[2];                // sign bits of the stem outputs X_0 (bf16 build, training): mask of the block-0 data gradient
    // backward
    std::vector<size_t> G0[2], G1[2], DX[2];    // G1: rb per frame, DX: (rb+1) per frame (DX[0] unused -> G0)
    std::vector<size_t> dFeatB, dFF;            // per frame: d outputs[i], d feat_prop(i) from the reconstruction
    size_t S[2], dWp[2], slab[2];               // per direction / stream
    size_t far_cnt;                             // [2][t] ints: far-source counters of the gather-form warp backward
    size_t chain_sync[2] = {0, 0};              // counters of the trunk chain launches (0: chains not planned)
    int chain_mode = 1;                         // VSRLAB_AMD_CHAIN, read ONCE per engine call (build()): 0 one launch per layer, 1 chains, 2 diagnostic
    size_t G_C0, G_U1, G_U0, G_P;
    // r04 (full arena, bf16): the gradients into the two pixel-shuffle layers and into the 1x1 fuse conv are kept PER FRAME (+9.8 GB per
    // clip at config 2), so that those layers' weight gradients are all-frames launches like the trunks' (28 + 28 + 14 one-frame launches
    // of 50-140 us each per step before: a one-frame launch at LR size is mostly prologue, slab write and first-tile latency)
    bool hrdef = false;
    std::vector<size_t> GU1f, GU0f, GPf;
    size_t dflows;              // fp32 planar, layout of `flows`: gradient w.r.t. the flows (train_flow / input gradient)
    size_t stem_wd_lr[2];       // data-gradient weights of the stems' 3 LR input channels (input gradient)
    size_t total;

    size_t xoff(int dir, int i, int b) const { return X[dir][(size_t)i * (rb + 1) + b]; }
    size_t aoff(int dir, int i, int b) const { return A[dir][(size_t)i * rb + b]; }
    size_t sboff(int dir, int i, int b) const { return SB[dir][(size_t)i * rb + b]; }
    // diet: the trunk's activation gradients live in a ring of two buffers per kind and direction (block parity): a block's weight
    // gradients are launched right behind its data gradients, on the same stream, so its buffers are free two blocks later
    size_t g1off(int dir, int i, int b) const { return diet ? G1[dir][b & 1] : G1[dir][(size_t)i * rb + b]; }
    size_t dxoff(int dir, int i, int b) const { return diet ? DX[dir][b & 1] : DX[dir][(size_t)i * (rb + 1) + b]; }

    int build(const VsrBasicVSRDesc& desc, int mode) {     // 0 inference, 1 training (frozen flow), 2 training incl. SPyNet
        d = desc; bwd = mode >= 1; flowgrad = mode >= 2;
        { const char* e = getenv("VSRLAB_AMD_CHAIN"); chain_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
        if (d.arena_mode != 0 && d.arena_mode != 1) return VSR_ERR_BADARG;
        diet = bwd && d.arena_mode == 1;
        rb = d.res_blocks; n = d.n; t = d.t; h = d.h; w = d.w; dtype = d.dtype;
        scale = d.upscale; ups = scale / 2; unsh = false;
        if (d.mid_channels != C || (scale != 4 && scale != 2) || rb < 1 || n < 1 || t < 1 || t > 32 || h < 1 || w < 1) return VSR_ERR_UNSUPPORTED;
        if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
        es = esize(dtype);
        px1 = (size_t)n * pm_image_elems(h, w, C);
        s_elems = (size_t)n * h * w * C;
        // a2: a blocked 64-channel tensor at 2h x 2w; a4: at the output size scale h x scale w (= a2 for upscale 2, which has no U1)
        const size_t a2 = (size_t)n * pm_image_elems(2 * h, 2 * w, C) * es, a4 = (size_t)n * pm_image_elems(scale * h, scale * w, C) * es;
        Bump b;
        const size_t w64 = (size_t)9 * C * C * es;
        for (int dir = 0; dir < 2; ++dir) {
            stem_w[dir] = b.take(w64 + (size_t)9 * C * 16 * es);
            stem_wd[dir] = b.take(w64);
            stem_b[dir] = b.take(C * 4);
            blk_w[dir].resize(2 * rb); blk_wd[dir].resize(2 * rb); blk_b[dir].resize(2 * rb);
            for (int k = 0; k < 2 * rb; ++k) { blk_w[dir][k] = b.take(w64); blk_wd[dir][k] = b.take(w64); blk_b[dir][k] = b.take(C * 4); }
        }
        point_w = b.take((size_t)2 * C * C * es); point_wd = b.take((size_t)2 * C * C * es); point_b = b.take(C * 4);
        for (int k = 0; k < 2; ++k) { up_w[k] = b.take(4 * w64); up_wd[k] = b.take(4 * w64); up_b[k] = b.take(4 * C * 4); }
        last0_w = b.take(w64); last0_wd = b.take(w64); last0_b = b.take(C * 4);
        last2_w = b.take((size_t)9 * 32 * C * es); last2_wd = b.take((size_t)9 * C * 16 * es); last2_b = b.take(64 * 4);
        if (t > 1) spy.plan(b, 2 * n * (t - 1), n * t, h, w, dtype);
        flows = b.take((size_t)2 * n * (t > 1 ? t - 1 : 1) * 2 * h * w * 4);
        const size_t a1 = px1 * es;
        for (int dir = 0; dir < 2; ++dir) {
            Wp[dir].assign(t, 0); feat[dir].assign(t, 0);
            if (bwd) {
                X[dir].assign((size_t)t * (rb + 1), 0); A[dir].assign((size_t)t * rb, 0); SB[dir].assign((size_t)t * rb, 0);
                SBX0[dir].assign(t, 0);
                const size_t sbytes = dtype == VSR_BF16 ? (size_t)n * cdiv(h, 8) * cdiv(w, 32) * 2048 : 256;
                for (int i = 0; i < t; ++i) {
                    SBX0[dir][i] = b.take(sbytes);
                    Wp[dir][i] = b.take(a1);
                    for (int k = 0; k <= rb; ++k) X[dir][(size_t)i * (rb + 1) + k] = b.take(a1);
                    for (int k = 0; k < rb; ++k) A[dir][(size_t)i * rb + k] = b.take(a1);
                    for (int k = 0; k < rb; ++k) SB[dir][(size_t)i * rb + k] = b.take(sbytes);
                    feat[dir][i] = xoff(dir, i, rb);
                }
            } else {
                for (int i = 0; i < t; ++i) feat[dir][i] = b.take(a1);
            }
        }
        for (int dir = 0; dir < 2; ++dir) { scratchA[dir] = b.take(a1); scratchW[dir] = b.take(a1); }
        const int nrec = bwd ? t : 1;
        Pt.assign(t, 0); U0.assign(t, 0); U1.assign(t, 0); C0.assign(t, 0);
        // diet: the upsampled tensors of a frame (U0, U1 = the upsample layers' outputs, C0 = conv_last.0's) are not kept: one shared
        // buffer each, recomputed from Pt[i] at the top of the frame's reconstruction backward (2.4 GB per frame at 540p x4)
        for (int i = 0; i < nrec; ++i) {
            Pt[i] = b.take(a1);
            if (diet && i > 0) { U0[i] = U0[0]; U1[i] = U1[0]; C0[i] = C0[0]; } else { U0[i] = b.take(a2); U1[i] = ups == 2 ? b.take(a4) : U0[i]; C0[i] = b.take(a4); }
        }
        for (int i = nrec; i < t; ++i) { Pt[i] = Pt[0]; U0[i] = U0[0]; U1[i] = U1[0]; C0[i] = C0[0]; }
        SBC0.assign(t, 0); SBPt.assign(t, 0);
        if (bwd && dtype == VSR_BF16)
            for (int i = 0; i < t; ++i) {
                SBC0[i] = b.take((size_t)n * cdiv(scale * h, 8) * cdiv(scale * w, 32) * 2048);
                SBPt[i] = b.take((size_t)n * cdiv(h, 8) * cdiv(w, 32) * 2048);
            }
        if (bwd) {
            for (int dir = 0; dir < 2; ++dir) {
                G0[dir].assign(t, 0);
                for (int i = 0; i < t; ++i) G0[dir][i] = b.take(a1);
                if (diet) {
                    G1[dir].assign(2, 0); DX[dir].assign(2, 0);
                    for (int k = 0; k < 2; ++k) { G1[dir][k] = b.take(a1); DX[dir][k] = b.take(a1); }
                } else if (dir == 0) {
                    G1[dir].assign((size_t)t * rb, 0); DX[dir].assign((size_t)t * (rb + 1), 0);
                    for (int i = 0; i < t; ++i) {
                        for (int k = 0; k < rb; ++k) G1[dir][(size_t)i * rb + k] = b.take(a1);
                        for (int k = 1; k <= rb; ++k) DX[dir][(size_t)i * (rb + 1) + k] = b.take(a1);
                    }
                } else {
                    // ONE set of trunk activation gradients for both directions (r04: 2 x 7 x 60 x 66.8 MB = 56 GB at config 2 before):
                    // a direction's gradients are dead once its all-frames weight-gradient launches have read them, and
                    // backward_impl runs direction 1 (data gradients, then weight gradients) and then direction 0 on ONE stream
                    G1[dir] = G1[0]; DX[dir] = DX[0];
                }
            }
            dFeatB.assign(t, 0); dFF.assign(t, 0);
            for (int i = 0; i < t; ++i) { dFeatB[i] = b.take(a1); dFF[i] = b.take(a1); }
            for (int dir = 0; dir < 2; ++dir) { S[dir] = b.take(s_elems * 8); dWp[dir] = b.take(a1); }
            far_cnt = b.take((size_t)2 * 32 * 4);
            G_C0 = b.take(a4); G_P = b.take(a1);
            // diet: dU1 is written after C0's last use and dU0 after U1's (recon_backward): they take those buffers
            // (upscale 2: there is no U1 / dU1; dU0 is conv_last.0's data gradient and takes C0's buffer in the diet arena)
            // r04 (full arena, bf16, persistent kernels): both are stored as four phase planes, whose 32-pixel row padding can exceed the
            // whole image's (2 ceil(x) >= ceil(2 x)): four planes of the next lower resolution each
            unsh = !diet && dtype == VSR_BF16 && !vsr_env().generic_conv;
            const size_t gu1 = unsh && 4 * a2 > a4 ? 4 * a2 : a4, gu0 = unsh && 4 * a1 > a2 ? 4 * a1 : a2;
            if (ups == 2) { if (diet) { G_U1 = C0[0]; G_U0 = U1[0]; } else { G_U1 = b.take(gu1); G_U0 = b.take(gu0); } }
            else { G_U0 = diet ? C0[0] : b.take(gu0); G_U1 = G_U0; }
            hrdef = unsh;
            GU1f.assign(t, G_U1); GU0f.assign(t, G_U0); GPf.assign(t, G_P);
            if (hrdef)
                for (int i = 1; i < t; ++i) {             // frame 0 keeps the buffers above
                    GU0f[i] = b.take(gu0);
                    GU1f[i] = ups == 2 ? b.take(gu1) : GU0f[i];
                    GPf[i] = b.take(a1);
                }
            int cp, xp, stride;
            vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
            for (int k = 0; k < 2; ++k) slab[k] = b.take((size_t)VSR_WGRAD_NWG * stride * 4);
        }
        // the trunk chains' work / row counters (conv3x3_chain.hip), one block per direction (= per stream)
        for (int dir = 0; dir < 2; ++dir) chain_sync[dir] = (bwd && dtype == VSR_BF16) ? b.take(vsr_chain_sync_bytes(VSR_CHAIN_MAX_LAYERS, n, h, w)) : 0;
        // appended last, so that every other offset is the same in modes 1 and 2
        dflows = 0;
        if (flowgrad && t > 1) {
            dflows = b.take((size_t)2 * n * (t - 1) * 2 * h * w * 4);
            spy.plan_save(b, dtype);
        }
        if (flowgrad) {
            for (int dir = 0; dir < 2; ++dir) stem_wd_lr[dir] = b.take((size_t)9 * 32 * C * es);
        }
        total = b.off;
        return VSR_OK;
    }
};

#define CK(expr) do { int _s = (expr); if (_s != VSR_OK) return _s; } while (0)

struct Ctx {
    const Plan& p;
    char* ws;
    hipStream_t st;
    int dtype;
    int lane;                   // 0: the caller's stream, 1: the helper stream (selects per-stream scratch)
    mutable std::vector<VsrPackDesc>* batch = nullptr;      // while set, pack() collects descriptors for ONE multi-tensor launch
    void* at(size_t off) const { return ws + off; }
    const float* fat(size_t off) const { return reinterpret_cast<const float*>(ws + off); }

    ConvArgs base(int N, int H, int W) const {
        ConvArgs a = {};
        a.in_step = 1; a.Hs = H; a.Ws = W; a.N = N; a.H = H; a.W = W; a.nz = 1;
        a.out_step = 1; a.Hd = H; a.Wd = W; a.CD = C; a.cout_real = C; a.dst_nstride = pm_image_elems(H, W, C);
        for (int s = 0; s < VSR_MAX_SRC; ++s) a.src_nstride[s] = pm_image_elems(H, W, C);
        return a;
    }
    // the destination of `a` (an N x H x W x 64 image, H and W even) as four phase planes of N x H/2 x W/2 x 64 (ConvArgs::unshuffle)
    static long long plane_elems(int N, int H, int W) { return (long long)N * pm_image_elems(H / 2, W / 2, C); }
    static void set_unshuffle(ConvArgs& a, int N, int H, int W) {
        a.unshuffle = 1; a.unshuffle_plane = plane_elems(N, H, W); a.dst_nstride = pm_image_elems(H / 2, W / 2, C);
    }
    // y = act(conv3x3(x) + bias) (+res) (*mask(aux)) -- 64 -> 64 at one resolution
    int conv64(const void* x, size_t wpack, const float* bias, void* y, int act, const void* res, const void* aux, int mask,
               int N, int H, int W, void* sign_out = nullptr, const void* sign_bits = nullptr, bool unshuffle = false) const {
        ConvArgs a = base(N, H, W);
        a.src[0] = x; a.wpack = at(wpack); a.bias = bias; a.dst[0] = y; a.act = act; a.res[0] = res; a.aux[0] = aux; a.mask_mode = mask;
        a.sign_out[0] = sign_out; a.sign_bits[0] = sign_bits;
        if (unshuffle) set_unshuffle(a, N, H, W);
        const int rc = vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
        // VSRLAB_AMD_GENERIC_CONV=1 (A/B switch): the generic kernel does not write sign bits, but later launches (hr_tail.hip's
        // conv_last.2 data gradient, the masked data gradients) read them -- r04: the switch gave wrong gradients since round 2
        if (rc == VSR_OK && sign_out && dtype == VSR_BF16 && vsr_env().generic_conv) return vsr_launch_sign_bits_c64(y, sign_out, N, H, W, st);
        return rc;
    }
    // conv3x3 64->256 + PixelShuffle(2): x (N,H,W,64) -> y (N,2H,2W,64)   (upsampling.py:10-12)
    int conv_ps(const void* x, size_t wpack, const float* bias4, void* y, int N, int H, int W) const {
        ConvArgs a = base(N, H, W);
        a.src[0] = x; a.wpack = at(wpack); a.w_zstride = 9 * C * C; a.bias = bias4; a.bias_zstride = C; a.nz = 4;
        a.out_step = 2; a.Hd = 2 * H; a.Wd = 2 * W; a.dst_nstride = pm_image_elems(2 * H, 2 * W, C);
        for (int z = 0; z < 4; ++z) { a.dst[z] = y; a.out_oy[z] = z >> 1; a.out_ox[z] = z & 1; }
        return vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
    }
    // data gradient of the above: dy (N,2H,2W,64) -> dx (N,H,W,64) (* mask(aux)) = sum over the 4 pixel-shuffle
    // phases z of a transposed 3x3 64->64 conv of dy's phase z.  bf16: four launches of the persistent kernel (the
    // four weight sets do not fit LDS together), phase z reading phase z-1's partial sum as its residual in place --
    // the partial sums pass through bf16 three times (~1.6x the rounding error of the final store alone), for
    // 560 us instead of 1470 us at 1080x1920.  fp32: one generic 4-source launch, accumulated in registers.
    // dy_planes: dy is stored phase-separated (four N x H x W planes, written by a launch with ConvArgs::unshuffle): phase z is then a
    // contiguous tensor instead of every second pixel of every second row of the 2H x 2W image (r04: the strided form fetched every
    // line of dy twice per data gradient and again twice per weight gradient); dx_planes: write dx phase-separated in turn.
    int conv_ps_dgrad(const void* dy, size_t wpackd, void* dx, const void* aux, int mask, int N, int H, int W, const void* sign_bits = nullptr,
                      bool dy_planes = false, bool dx_planes = false) const {
        if (dtype == VSR_BF16) {
            for (int z = 0; z < 4; ++z) {
                ConvArgs a = base(N, H, W);
                if (dy_planes) {
                    a.src[0] = (const char*)dy + (size_t)z * plane_elems(N, 2 * H, 2 * W) * p.es;      // plane z: N x H x W, stride 1 (base() set it up)
                } else {
                    a.in_step = 2; a.Hs = 2 * H; a.Ws = 2 * W;
                    a.src[0] = dy; a.src_oy[0] = z >> 1; a.src_ox[0] = z & 1; a.src_nstride[0] = pm_image_elems(2 * H, 2 * W, C);
                }
                a.wpack = at(wpackd + (size_t)z * 9 * C * C * p.es); a.dst[0] = dx;
                a.res[0] = z > 0 ? dx : nullptr;
                if (dx_planes) set_unshuffle(a, N, H, W);
                if (z == 3) { a.aux[0] = aux; a.mask_mode = mask; a.sign_bits[0] = aux ? sign_bits : nullptr; }
                int rc = vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
                if (rc != VSR_OK) return rc;
            }
            return VSR_OK;
        }
        ConvArgs a = base(N, H, W);
        a.nz = 1; a.in_step = 2; a.Hs = 2 * H; a.Ws = 2 * W;
        for (int s = 0; s < 4; ++s) { a.src[s] = dy; a.src_oy[s] = s >> 1; a.src_ox[s] = s & 1; a.src_nstride[s] = pm_image_elems(2 * H, 2 * W, C); }
        a.wpack = at(wpackd); a.dst[0] = dx; a.aux[0] = aux; a.mask_mode = mask;
        return vsr_launch_conv(dtype, 3, 4, 64, 64, 0, 64, EPI_NHWC, a, st);
    }
    int pack(const float* w, size_t dst, int KK, int RP, int CPd, int r_real, int c_real, int I_total, int i_off, int o_mul,
             int o_add, int mode) const {
        return pack_any(dtype, w, at(dst), KK, RP, CPd, r_real, c_real, I_total, i_off, o_mul, o_add, mode);
    }
    int pack_bias(const float* b, size_t dst, int nreal, int o_mul = 1, int o_add = 0) const {
        return pack_any(VSR_F32, b, at(dst), 1, nreal, 1, nreal, 1, 1, 0, o_mul, o_add, 0);
    }
    int pack_any(int dt, const float* w, void* dst, int KK, int RP, int CPd, int r_real, int c_real, int I_total, int i_off, int o_mul,
                 int o_add, int mode) const {
        if (!batch) return vsr_launch_pack_weights(dt, w, dst, KK, RP, CPd, r_real, c_real, I_total, i_off, o_mul, o_add, mode, st);
        VsrPackDesc d = {};
        d.w = w; d.dst = dst; d.total = KK * RP * CPd;
        d.KK = (short)KK; d.RP = (short)RP; d.CPd = (short)CPd; d.r_real = (short)r_real; d.c_real = (short)c_real;
        d.I_total = (short)I_total; d.i_off = (short)i_off; d.o_mul = (short)o_mul; d.o_add = (short)o_add;
        d.mode = (unsigned char)mode; d.dtype = (unsigned char)dt;
        batch->push_back(d);
        return VSR_OK;
    }
};

// parameter index helpers (order documented in vsrlab_hip.h)
struct PIdx {
    int rb;
    int ups = 2;                 // PixelShufflePack count = upscale // 2 (basicvsr.py:19)
    int trunk_base(int dir) const { return dir * (2 + 4 * rb); }
    int stem_w(int dir) const { return trunk_base(dir); }
    int stem_b(int dir) const { return trunk_base(dir) + 1; }
    int blk_w(int dir, int k) const { return trunk_base(dir) + 2 + 2 * k; }      // k = 2*block + (conv2 ? 1 : 0)
    int blk_b(int dir, int k) const { return trunk_base(dir) + 3 + 2 * k; }
    int point_w() const { return 2 * (2 + 4 * rb); }
    int point_b() const { return point_w() + 1; }
    int up_w(int k) const { return point_w() + 2 + 2 * k; }
    int up_b(int k) const { return up_w(k) + 1; }
    int last0_w() const { return point_w() + 2 + 2 * ups; }
    int last0_b() const { return last0_w() + 1; }
    int last2_w() const { return last0_w() + 2; }
    int last2_b() const { return last0_w() + 3; }
    int spy_base() const { return last0_w() + 4; }
    int spy_w(int lvl, int j) const { return spy_base() + (lvl * NSPY + j) * 2; }
    int spy_b(int lvl, int j) const { return spy_w(lvl, j) + 1; }
    int spy_mean() const { return spy_base() + 60; }
    int spy_std() const { return spy_base() + 61; }
    int count() const { return spy_base() + 62; }
};

// ---- SPyNet (spynet.py:38-93) for P frame pairs ------------------------------------------------
// frames: planar fp32 (F,3,h,w).  pair_mode 0: BasicVSR pairing over (n,t) (basicvsr.py:32-35);
// pair_mode 1: frames = [ref_0..ref_{P-1}, supp_0..supp_{P-1}].
int spynet_pack(const Ctx& c, const SpyPlan& sp, const float* const* params, int base_idx) {
    for (int l = 0; l < 6; ++l)
        for (int j = 0; j < NSPY; ++j) {
            const float* w = params[base_idx + (l * NSPY + j) * 2];
            const float* b = params[base_idx + (l * NSPY + j) * 2 + 1];
            CK(c.pack(w, sp.wpack[l][j], 49, SPY_COP[j], SPY_CIP[j], SPY_CO[j], SPY_CI[j], SPY_CI[j], 0, 1, 0, 0));
            CK(c.pack_bias(b, sp.bias[l][j], SPY_CO[j]));
            // data-gradient weights: rows = the conv's input channels (template COUT of the dgrad launch), K = its outputs
            if (sp.save) CK(c.pack(w, sp.wpackd[l][j], 49, SPY_DROWS[j], SPY_DK[j], SPY_CI[j], SPY_CO[j], SPY_CI[j], 0, 1, 0, 1));
        }
    return VSR_OK;
}

// last_relu: the reference's RealBasicVSR Spynet ends every level in a ReLU (spynet.py:16-18); the canonical SPyNet of
// vsr/models/VRT/modules/spynet.py:76 does not.  level_out[l] (optional, l = 0..5): the level's flow resized to
// (h >> (5-l), w >> (5-l)) like VRT's return_levels (VRT/modules/spynet.py:134-141).
int spynet_run(const Ctx& c, const SpyPlan& sp, const float* frames, const float* mean, const float* std, int n, int t,
               int pair_mode, float* flows_out, bool last_relu = true, float* const* level_out = nullptr) {
    const int P = sp.P, F = sp.F, hu = sp.hu, wu = sp.wu;
    CK(vsr_launch_resize_norm(frames, (float*)c.at(sp.pyr[5]), mean, std, F, sp.h, sp.w, hu, wu, c.st));
    for (int l = 5; l > 0; --l)
        CK(vsr_launch_avgpool2(c.fat(sp.pyr[l]), (float*)c.at(sp.pyr[l - 1]), (long long)F * 3, hu >> (5 - l), wu >> (5 - l), c.st));
    size_t fprev = sp.flow_a, fcur = sp.flow_b;
    for (int l = 0; l < 6; ++l) {
        const int hl = hu >> (5 - l), wl = wu >> (5 - l);
        const size_t fup = sp.save ? sp.sfup[l] : sp.flow_up;
        const size_t bufs[NSPY + 1] = {sp.save ? sp.sx[l][0] : sp.x16, sp.save ? sp.sx[l][1] : sp.b32a, sp.save ? sp.sx[l][2] : sp.b64,
                                       sp.save ? sp.sx[l][3] : sp.b32b, sp.save ? sp.sx[l][4] : sp.b16, 0};
        CK(vsr_launch_spynet_prepare(c.dtype, c.fat(sp.pyr[l]), l == 0 ? nullptr : c.fat(fprev), (float*)c.at(fup), c.at(bufs[0]),
                                     n, t, P, pair_mode, hl, wl, l == 0, c.st));
        for (int j = 0; j < NSPY; ++j) {
            ConvArgs a = c.base(P, hl, wl);
            a.src[0] = c.at(bufs[j]); a.src_nstride[0] = pm_image_elems(hl, wl, SPY_CIP[j]);
            a.wpack = c.at(sp.wpack[l][j]); a.bias = c.fat(sp.bias[l][j]);
            a.act = (j < NSPY - 1 || last_relu) ? ACT_RELU : ACT_NONE;     // RealBasicVSR's Spynet: ReLU after the LAST conv too (spynet.py:16-18)
            a.cout_real = SPY_CO[j];
            if (j < NSPY - 1) {
                a.dst[0] = c.at(bufs[j + 1]); a.CD = SPY_CD[j]; a.dst_nstride = pm_image_elems(hl, wl, SPY_CD[j]);
                CK(vsr_launch_conv(c.dtype, 7, 1, SPY_CIP[j], SPY_CIP[j], 0, SPY_COP[j], EPI_NHWC, a, c.st));
            } else if (!sp.save) {
                a.dst[0] = c.at(fcur); a.dst_nstride = (long long)2 * hl * wl; a.pres = c.fat(fup);   // flow = flow_up + residue (spynet.py:65)
                CK(vsr_launch_conv(c.dtype, 7, 1, 16, 16, 0, 32, EPI_PLANAR, a, c.st));
            } else {                                           // keep the residue: its sign is the last ReLU's mask
                a.dst[0] = c.at(sp.sres[l]); a.dst_nstride = (long long)2 * hl * wl;
                CK(vsr_launch_conv(c.dtype, 7, 1, 16, 16, 0, 32, EPI_PLANAR, a, c.st));
                CK(vsr_launch_add_f32(c.fat(fup), c.fat(sp.sres[l]), (float*)c.at(fcur), (long long)P * 2 * hl * wl, c.st));
            }
        }
        if (level_out && level_out[l])
            CK(vsr_launch_flow_out(c.fat(fcur), level_out[l], P, hl, wl, sp.h >> (5 - l), sp.w >> (5 - l), c.st));
        size_t tmp = fprev; fprev = fcur; fcur = tmp;
    }
    if (!flows_out) return VSR_OK;
    return vsr_launch_flow_out(c.fat(fprev), flows_out, P, hu, wu, sp.h, sp.w, c.st);
}

int pack_all_collect(const Ctx& c, const Plan& p, const float* const* prm);
// every weight / bias pack of a forward (~465 tensors) in a handful of multi-tensor launches
int pack_all(const Ctx& c, const Plan& p, const float* const* prm) {
    std::vector<VsrPackDesc> descs;
    descs.reserve(512);
    c.batch = &descs;
    const int rc = pack_all_collect(c, p, prm);
    c.batch = nullptr;
    if (rc != VSR_OK) return rc;
    return vsr_launch_pack_multi(descs.data(), (int)descs.size(), c.st);
}

int pack_all_collect(const Ctx& c, const Plan& p, const float* const* prm) {
    const PIdx ix{p.rb, p.ups};
    const int dt = c.dtype; (void)dt;
    for (int dir = 0; dir < 2; ++dir) {
        const float* sw = prm[ix.stem_w(dir)];
        // cat([lr_i(3), feat(64)]) (basicvsr.py:56,71): source 0 = feat = input channels 3..66, source 1 = LR = 0..2
        CK(c.pack(sw, p.stem_w[dir], 9, C, C, C, C, C + 3, 3, 1, 0, 0));
        CK(c.pack(sw, p.stem_w[dir] + (size_t)9 * C * C * p.es, 9, C, 16, C, 3, C + 3, 0, 1, 0, 0));
        CK(c.pack_bias(prm[ix.stem_b(dir)], p.stem_b[dir], C));
        if (p.bwd) CK(c.pack(sw, p.stem_wd[dir], 9, C, C, C, C, C + 3, 3, 1, 0, 1));
        if (p.flowgrad) CK(c.pack(sw, p.stem_wd_lr[dir], 9, 32, C, 3, C, C + 3, 0, 1, 0, 1));
        for (int k = 0; k < 2 * p.rb; ++k) {
            CK(c.pack(prm[ix.blk_w(dir, k)], p.blk_w[dir][k], 9, C, C, C, C, C, 0, 1, 0, 0));
            if (p.bwd) CK(c.pack(prm[ix.blk_w(dir, k)], p.blk_wd[dir][k], 9, C, C, C, C, C, 0, 1, 0, 1));
            CK(c.pack_bias(prm[ix.blk_b(dir, k)], p.blk_b[dir][k], C));
        }
    }
    for (int s = 0; s < 2; ++s) {
        CK(c.pack(prm[ix.point_w()], p.point_w + (size_t)s * C * C * p.es, 1, C, C, C, C, 2 * C, s * C, 1, 0, 0));
        if (p.bwd) CK(c.pack(prm[ix.point_w()], p.point_wd + (size_t)s * C * C * p.es, 1, C, C, C, C, 2 * C, s * C, 1, 0, 1));
    }
    CK(c.pack_bias(prm[ix.point_b()], p.point_b, C));
    for (int k = 0; k < p.ups; ++k)
        for (int z = 0; z < 4; ++z) {
            // PixelShuffle(2): out[c, 2y+i, 2x+j] = conv[4c+2i+j, y, x]  => sub-conv z uses rows 4c+z
            CK(c.pack(prm[ix.up_w(k)], p.up_w[k] + (size_t)z * 9 * C * C * p.es, 9, C, C, C, C, C, 0, 4, z, 0));
            if (p.bwd) CK(c.pack(prm[ix.up_w(k)], p.up_wd[k] + (size_t)z * 9 * C * C * p.es, 9, C, C, C, C, C, 0, 4, z, 1));
            CK(c.pack_bias(prm[ix.up_b(k)], p.up_b[k] + (size_t)z * C * 4, C, 4, z));
        }
    CK(c.pack(prm[ix.last0_w()], p.last0_w, 9, C, C, C, C, C, 0, 1, 0, 0));
    if (p.bwd) CK(c.pack(prm[ix.last0_w()], p.last0_wd, 9, C, C, C, C, C, 0, 1, 0, 1));
    CK(c.pack_bias(prm[ix.last0_b()], p.last0_b, C));
    CK(c.pack(prm[ix.last2_w()], p.last2_w, 9, 32, C, 3, C, C, 0, 1, 0, 0));
    if (p.bwd) CK(c.pack(prm[ix.last2_w()], p.last2_wd, 9, C, 16, C, 3, C, 0, 1, 0, 1));
    CK(c.pack_bias(prm[ix.last2_b()], p.last2_b, 3));
    if (p.t > 1) CK(spynet_pack(c, p.spy, prm, ix.spy_base()));
    return VSR_OK;
}

// The 2 rb convolutions of a frame's residual blocks (and their data gradients) as ONE launch (conv3x3_chain.hip): training
// arena only (every layer has a buffer of its own there), bf16.  VSRLAB_AMD_CHAIN=0 (read once per engine call, Plan::build) = one launch per layer.
bool chain_on(const Plan& p) { return p.chain_mode != 0 && p.chain_sync[0] && p.chain_sync[1] && 2 * p.rb <= VSR_CHAIN_MAX_LAYERS; }
unsigned chain_off(size_t o) { return (unsigned)(o >> 8); }
int chain_launch(const Plan& p, const ChainArgs& a, hipStream_t st) {
    // A batch of clips, image by image (r04).  Layer l + 1 reads what layer l has just written and what layer l - 1 wrote (the
    // identity): with one 540p image in flight those ~200 MB live in the 256 MiB Infinity Cache, with two they do not -- measured on
    // `bench.py --clips 2`: 37.0 us per image and layer in one launch of both against 34.6 us with one clip.  Large images only:
    // small ones need both images' tiles to fill the chip.
    const size_t img_bytes = (size_t)pm_image_elems(a.H, a.W, 64) * 2, sb_bytes = (size_t)cdiv(a.H, 8) * cdiv(a.W, 32) * 2048;
    if (a.N > 1 && img_bytes * 3 * a.N > ((size_t)192 << 20) && cdiv(a.H, 8) * cdiv(a.W, 32) >= 2 * vsr_num_cus()) {
        for (int m = 0; m < a.N; ++m) {
            ChainArgs b = a;
            b.N = 1;
            const unsigned di = (unsigned)((m * img_bytes) >> 8), ds = (unsigned)((m * sb_bytes) >> 8);
            for (int l = 0; l < a.nlayers; ++l) {
                ChainLayer& L = b.layer[l];
                L.src += di; L.dst += di;
                if (L.res != 0xffffffffu) L.res += di;
                if (L.sbits != 0xffffffffu) L.sbits += ds;
                if (L.sout != 0xffffffffu) L.sout += ds;
            }
            CK(chain_launch(p, b, st));
        }
        return VSR_OK;
    }
    if (p.chain_mode == 2) {            // diagnostic: the same kernel, one layer per launch (no hand-off between workgroups)
        for (int l = 0; l < a.nlayers; ++l) {
            ChainArgs b = a;
            b.nlayers = 1; b.layer[0] = a.layer[l];
            CK(vsr_launch_conv3x3_chain(b, vsr_num_cus(), st));
        }
        return VSR_OK;
    }
    return vsr_launch_conv3x3_chain(a, vsr_num_cus(), st);
}
int trunk_chain_forward(const Ctx& c, const Plan& p, int dir, int i) {
    ChainArgs a = {};
    a.base = c.ws; a.sync = (unsigned*)c.at(p.chain_sync[dir]); a.N = p.n; a.H = p.h; a.W = p.w; a.nlayers = 2 * p.rb;
    for (int b = 0; b < p.rb; ++b) {
        ChainLayer& l1 = a.layer[2 * b];
        l1 = {chain_off(p.xoff(dir, i, b)), chain_off(p.aoff(dir, i, b)), 0xffffffffu, 0xffffffffu, chain_off(p.sboff(dir, i, b)),
              chain_off(p.blk_w[dir][2 * b]), chain_off(p.blk_b[dir][2 * b]), CHAIN_RELU};
        ChainLayer& l2 = a.layer[2 * b + 1];
        l2 = {chain_off(p.aoff(dir, i, b)), chain_off(p.xoff(dir, i, b + 1)), chain_off(p.xoff(dir, i, b)), 0xffffffffu, 0xffffffffu,
              chain_off(p.blk_w[dir][2 * b + 1]), chain_off(p.blk_b[dir][2 * b + 1]), CHAIN_SKIP};
    }
    return chain_launch(p, a, c.st);
}
// dA_b = dgrad(conv2)(dX_{b+1}) * ReLU'(A_b), dX_b = dX_{b+1} + dgrad(conv1)(dA_b) for b = rb-1 .. 1, and dA_0: 2 rb - 1 layers
int trunk_chain_backward(const Ctx& c, const Plan& p, int dir, int i) {
    ChainArgs a = {};
    a.base = c.ws; a.sync = (unsigned*)c.at(p.chain_sync[dir]); a.N = p.n; a.H = p.h; a.W = p.w;
    int L = 0;
    for (int b = p.rb - 1; b >= 0; --b) {
        a.layer[L++] = {chain_off(p.dxoff(dir, i, b + 1)), chain_off(p.g1off(dir, i, b)), 0xffffffffu, chain_off(p.sboff(dir, i, b)), 0xffffffffu,
                        chain_off(p.blk_wd[dir][2 * b + 1]), 0xffffffffu, CHAIN_MASK};
        if (b > 0)
            a.layer[L++] = {chain_off(p.g1off(dir, i, b)), chain_off(p.dxoff(dir, i, b)), chain_off(p.dxoff(dir, i, b + 1)), 0xffffffffu, 0xffffffffu,
                            chain_off(p.blk_wd[dir][2 * b]), 0xffffffffu, CHAIN_SKIP};
    }
    a.nlayers = L;
    return chain_launch(p, a, c.st);
}

// one call of ResidualBlock (conv.py:94-103) on cat([lr_i, warped feat])
int trunk_forward(const Ctx& c, const Plan& p, int dir, int i, const void* warped, const float* lrs) {
    const int n = p.n, h = p.h, w = p.w, rb = p.rb;
    void* x = p.bwd ? c.at(p.xoff(dir, i, 0)) : c.at(p.feat[dir][i]);
    {
        ConvArgs a = c.base(n, h, w);
        a.src[0] = warped;                                   // null => zeros (first frame of the direction)
        a.src[1] = lrs + (size_t)i * 3 * h * w; a.src_nstride[1] = (long long)p.t * 3 * h * w;
        a.wpack = c.at(p.stem_w[dir]); a.bias = c.fat(p.stem_b[dir]); a.dst[0] = x; a.act = ACT_LEAKY;
        CK(vsr_launch_conv(c.dtype, 3, 2, 64, 16, 1, 64, EPI_NHWC, a, c.st));
        // the stem runs on the generic two-source kernel: its LeakyReLU sign bits for the block-0 data gradient come from a 66 MB pass
        if (p.bwd && c.dtype == VSR_BF16) CK(vsr_launch_sign_bits_c64(x, c.at(p.SBX0[dir][i]), n, h, w, c.st));
    }
    if (chain_on(p)) return trunk_chain_forward(c, p, dir, i);
    for (int b = 0; b < rb; ++b) {      // x + conv2(relu(conv1(x)))   (conv.py:89-92)
        void* act = p.bwd ? c.at(p.aoff(dir, i, b)) : c.at(p.scratchA[dir]);
        void* xn = p.bwd ? c.at(p.xoff(dir, i, b + 1)) : x;   // inference: in place (residual read = own pixel)
        CK(c.conv64(x, p.blk_w[dir][2 * b], c.fat(p.blk_b[dir][2 * b]), act, ACT_RELU, nullptr, nullptr, 0, n, h, w,
                    (p.bwd && c.dtype == VSR_BF16) ? c.at(p.sboff(dir, i, b)) : nullptr));
        CK(c.conv64(act, p.blk_w[dir][2 * b + 1], c.fat(p.blk_b[dir][2 * b + 1]), xn, ACT_NONE, x, nullptr, 0, n, h, w));
        x = xn;
    }
    return VSR_OK;
}

int recon_forward(const Ctx& c, const Plan& p, int i, const float* lrs, float* sr) {
    const int n = p.n, h = p.h, w = p.w;
    {
        ConvArgs a = c.base(n, h, w);       // point_conv on cat([outputs[i], feat_prop]) (basicvsr.py:75-77)
        a.src[0] = c.at(p.feat[0][i]); a.src[1] = c.at(p.feat[1][i]);
        a.wpack = c.at(p.point_w); a.bias = c.fat(p.point_b); a.dst[0] = c.at(p.Pt[i]); a.act = ACT_LEAKY;
        CK(vsr_launch_conv(c.dtype, 1, 2, 64, 64, 0, 64, EPI_NHWC, a, c.st));
        if (p.bwd && c.dtype == VSR_BF16) CK(vsr_launch_sign_bits_c64(c.at(p.Pt[i]), c.at(p.SBPt[i]), n, h, w, c.st));   // mask of upsample.0's data gradient
    }
    const int S = p.scale;                     // upscale: S / 2 PixelShufflePacks (basicvsr.py:19); U1 = U0 for S = 2 (Plan::build)
    CK(c.conv_ps(c.at(p.Pt[i]), p.up_w[0], c.fat(p.up_b[0]), c.at(p.U0[i]), n, h, w));
    if (p.ups == 2) CK(c.conv_ps(c.at(p.U0[i]), p.up_w[1], c.fat(p.up_b[1]), c.at(p.U1[i]), n, 2 * h, 2 * w));
    CK(c.conv64(c.at(p.U1[i]), p.last0_w, c.fat(p.last0_b), c.at(p.C0[i]), ACT_LEAKY, nullptr, nullptr, 0, n, S * h, S * w,
                (p.bwd && c.dtype == VSR_BF16) ? c.at(p.SBC0[i]) : nullptr));
    {
        ConvArgs a = c.base(n, S * h, S * w);   // conv_last.2 + bilinear xS skip (basicvsr.py:21-22,82)
        a.src[0] = c.at(p.C0[i]); a.wpack = c.at(p.last2_w); a.bias = c.fat(p.last2_b); a.cout_real = 3;
        a.dst[0] = sr + (size_t)i * 3 * S * S * h * w; a.dst_nstride = (long long)p.t * 3 * S * S * h * w;
        a.base_lr = lrs + (size_t)i * 3 * h * w; a.base_nstride = (long long)p.t * 3 * h * w; a.base_h = h; a.base_w = w; a.base_scale = S;
        CK(vsr_launch_conv(c.dtype, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, c.st));
    }
    return VSR_OK;
}

const float* flow_ptr(const Ctx& c, const Plan& p, int forward, int i) {   // flow of pair i, batch stride (t-1)*2*h*w
    return c.fat(p.flows) + ((size_t)forward * p.n * (p.t - 1) + i) * 2 * p.h * p.w;
}

// The two propagation directions are independent until the fusion conv (basicvsr.py:46-77), so their
// chains of 61 dependent launches per frame run on two streams: each stream's launch gap and pipeline
// fill is covered by the other's kernels.  The helper stream is forked from and joined to the caller's
// stream with events, so for the caller all work is still ordered in the stream it passed.
struct Helper { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; int dev = -1; };
int get_helper(Helper** out) {
    thread_local Helper hp[16];
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return VSR_ERR_UNSUPPORTED;
    Helper& x = hp[dev];
    if (!x.s) {
        HIP_CHECK_RET(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
        HIP_CHECK_RET(hipEventCreateWithFlags(&x.fork, hipEventDisableTiming));
        HIP_CHECK_RET(hipEventCreateWithFlags(&x.join, hipEventDisableTiming));
        x.dev = dev;
    }
    *out = &x;
    return VSR_OK;
}
bool single_stream() { return vsr_env().single_stream; }
struct Fork {   // RAII: the join is enqueued on every exit path
    hipStream_t main; Helper* h; bool active = false;
    int begin(bool fork = true) {
        if (single_stream() || !fork) return VSR_OK;
        CK(get_helper(&h));
        HIP_CHECK_RET(hipEventRecord(h->fork, main));
        HIP_CHECK_RET(hipStreamWaitEvent(h->s, h->fork, 0));
        active = true;
        return VSR_OK;
    }
    hipStream_t side() const { return active ? h->s : main; }
    int end() {
        if (!active) return VSR_OK;
        active = false;
        HIP_CHECK_RET(hipEventRecord(h->join, h->s));
        HIP_CHECK_RET(hipStreamWaitEvent(main, h->join, 0));
        return VSR_OK;
    }
    ~Fork() { (void)end(); }
};

int forward_chain(const Ctx& c, const Plan& p, int dir, const float* lrs) {
    const int n = p.n, t = p.t, h = p.h, w = p.w;
    const long long fstride = (long long)(t - 1) * 2 * h * w;
    for (int k = 0; k < t; ++k) {
        // dir 0: backward-time propagation, frames t-1 .. 0 (basicvsr.py:46-60); dir 1: forward-time (basicvsr.py:62-73)
        const int i = dir == 0 ? t - 1 - k : k;
        void* warped = nullptr;
        if (k > 0) {
            const int prev = dir == 0 ? i + 1 : i - 1;
            warped = p.bwd ? c.at(p.Wp[dir][i]) : c.at(p.scratchW[dir]);
            CK(vsr_launch_warp_fwd(c.dtype, c.at(p.feat[dir][prev]), flow_ptr(c, p, dir, dir == 0 ? i : i - 1), warped, n, h, w, C, fstride, c.st));
        }
        CK(trunk_forward(c, p, dir, i, warped, lrs));
    }
    return VSR_OK;
}

int forward_impl(const Plan& p, const float* const* prm, const float* lrs, float* sr, char* ws, hipStream_t st) {
    const Ctx c{p, ws, st, p.dtype, 0};
    const PIdx ix{p.rb, p.ups};
    const int n = p.n, t = p.t;
    CK(pack_all(c, p, prm));
    if (t > 1) CK(spynet_run(c, p.spy, lrs, prm[ix.spy_mean()], prm[ix.spy_std()], n, t, 0, (float*)c.at(p.flows)));
    {
        Fork f{st, nullptr};
        // (chain launches spin on each other's tiles: two of them side by side could each hold the CUs the other's unstarted
        // workgroups need -- conv3x3_chain.hip, "Work distribution" -- so with the chains on, both directions share the caller's stream)
        CK(f.begin(!chain_on(p)));
        const Ctx c0{p, ws, st, p.dtype, 0}, c1{p, ws, f.side(), p.dtype, 1};
        CK(forward_chain(c0, p, 0, lrs));
        CK(forward_chain(c1, p, 1, lrs));
        CK(f.end());
    }
    for (int i = 0; i < t; ++i) CK(recon_forward(c, p, i, lrs, sr));      // fusion + upsampling (basicvsr.py:75-82)
    return VSR_OK;
}

// ---- backward ------------------------------------------------------------------------------------
struct WG {   // one weight-gradient launch + reduction
    const Ctx& c;
    int run(int ks, int cx, bool xp, int cout, bool dyp, WgradArgs& a, int cout_real, int cin_real, float* gw, int I_total,
            int i_off, int o_mul, int o_add, float* gb) const {
        if (!gw && !gb) return VSR_OK;
        int cp, xpd, stride;
        vsr_wgrad_slab_dims(ks, cx, cout, &cp, &xpd, &stride);
        a.slab = (float*)c.at(c.p.slab[c.lane]); a.slab_stride = stride;
        const int tiles = a.N * cdiv(a.H, 8) * cdiv(a.W, 32);
        int cp3, xp3, stride3;
        vsr_wgrad_slab_dims(3, 64, 64, &cp3, &xp3, &stride3);          // the slab buffer holds VSR_WGRAD_NWG of these
        const long long cap = (long long)VSR_WGRAD_NWG * stride3 / stride;
        int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
        if (nwg > cap) nwg = (int)cap;
        if (nwg > 1) nwg &= ~1;
        int nslabs = 0;
        if (c.dtype == VSR_BF16 && ks == 3 && cx == 64 && !xp && cout == 16 && dyp && a.nseg == 1 && a.x_step == 1 && a.dy_step == 1 &&
            a.Hx == a.H && a.Wx == a.W && a.Hy == a.H && a.Wy == a.W) {
            // 64 -> 3 conv with a planar cotangent (conv_last.2, the pre-clean out conv): streaming kernel of hr_tail.hip
            CK(vsr_launch_last2_wgrad(a.x[0], reinterpret_cast<const float*>(a.dy[0]), a.dy_nstride, a.slab, stride, a.N, a.H, a.W, &nslabs, c.st));
        } else
        CK(vsr_launch_wgrad(c.dtype, ks, cx, xp, cout, dyp, a, nwg, &nslabs, c.st));
        if (!gw) return VSR_ERR_BADARG;
        return vsr_launch_wgrad_reduce(a.slab, nslabs, ks, cx, cout, cout_real, cin_real, gw, I_total, i_off, o_mul, o_add, gb, 1, c.st);
    }
};

WgradArgs wg_base(int N, int H, int W) {
    WgradArgs a = {};
    a.N = N; a.H = H; a.W = W; a.nseg = 1;
    a.x_step = 1; a.Hx = H; a.Wx = W; a.x_nstride = pm_image_elems(H, W, C);
    a.dy_step = 1; a.Hy = H; a.Wy = W; a.dy_nstride = pm_image_elems(H, W, C);
    return a;
}

// Backward of spynet_run for train_flow (spynet.py:38-93): dflows_out = d loss / d flows (P,2,h,w) ->
// weight / bias gradients of the 6 x 5 convs (g[base_idx ...], OIHW fp32).  The frames are not differentiated.
// dframes (optional): (F,3,h,w) fp32, ACCUMULATED into: the gradient w.r.t. the input frames (through the pyramid).
// last_relu = false: the canonical SPyNet (no ReLU behind a level's last conv); dlevel (optional, 6 entries, NULL = none):
// cotangents of the per-level outputs of spynet_run's level_out (VRT's return_levels); dflows_out may then be NULL.
int spynet_backward(const Ctx& c, const SpyPlan& sp, const float* dflows_out, int n, int t, int pair_mode, float* const* g,
                    int base_idx, float* dframes = nullptr, const float* std = nullptr, bool last_relu = true,
                    const float* const* dlevel = nullptr) {
    const WG wg{c};
    const int P = sp.P, hu = sp.hu, wu = sp.wu;
    size_t dcur = sp.dfa, dprev = sp.dfb;
    HIP_CHECK_RET(hipMemsetAsync(c.at(dcur), 0, (size_t)P * 2 * hu * wu * 4, c.st));
    if (dflows_out) CK(vsr_launch_flow_out_bwd(dflows_out, (float*)c.at(dcur), P, hu, wu, sp.h, sp.w, c.st));
    if (dframes)
        for (int l = 0; l < 6; ++l)
            HIP_CHECK_RET(hipMemsetAsync(c.at(sp.dpyr[l]), 0, (size_t)sp.F * 3 * (hu >> (5 - l)) * (wu >> (5 - l)) * 4, c.st));
    for (int l = 5; l >= 0; --l) {
        const int hl = hu >> (5 - l), wl = wu >> (5 - l);
        if (dlevel && dlevel[l])   // this level's flow was also an output (resized to the frame's own pyramid size): add its cotangent
            CK(vsr_launch_flow_out_bwd(dlevel[l], (float*)c.at(dcur), P, hl, wl, sp.h >> (5 - l), sp.w >> (5 - l), c.st));
        // flow_l = flow_up + ReLU(conv5): dY of the last conv, as a 16-channel pixel-major tensor
        CK(vsr_launch_spynet_dres(c.dtype, c.fat(dcur), last_relu ? c.fat(sp.sres[l]) : nullptr, c.at(sp.dres), P, hl, wl, c.st));
        size_t dy = sp.dres;
        for (int j = NSPY - 1; j >= 0; --j) {
            const int CI = SPY_CIP[j], CO = SPY_DK[j];
            float* gw = g ? g[base_idx + (l * NSPY + j) * 2] : nullptr;
            float* gb = g ? g[base_idx + (l * NSPY + j) * 2 + 1] : nullptr;
            if (gw || gb) {
                // fp32, 64 input channels: two 32-channel halves (the 14x38-pixel fp32 tile of 64 channels exceeds LDS)
                const int nhalf = (c.dtype == VSR_F32 && CI == 64) ? 2 : 1;
                for (int hf = 0; hf < nhalf; ++hf) {
                    WgradArgs a = wg_base(P, hl, wl);
                    a.x[0] = c.at(sp.sx[l][j]); a.x_nstride = pm_image_elems(hl, wl, CI);
                    a.dy[0] = c.at(dy); a.dy_nstride = pm_image_elems(hl, wl, CO);
                    const int cx = CI / nhalf;
                    if (nhalf == 2) { a.x_ctotal = CI; a.x_coff = hf * (cx / 8); }
                    const int cin_real = nhalf == 2 ? cx : SPY_CI[j];
                    CK(wg.run(7, cx, false, CO, false, a, SPY_CO[j], cin_real, gw, SPY_CI[j], hf * cx, 1, 0, hf == 0 ? gb : nullptr));
                }
            }
            if (j == 0 && l == 0 && !dframes) break;       // level 0's input depends on the frames only
            // dX_j = dgrad(conv_j)(dY_j) (* ReLU'(X_j) for j > 0: X_j is the previous conv's ReLU output)
            const size_t dx = (dy == sp.gA) ? sp.gB : sp.gA;
            ConvArgs a = c.base(P, hl, wl);
            a.src[0] = c.at(dy); a.src_nstride[0] = pm_image_elems(hl, wl, CO);
            a.wpack = c.at(sp.wpackd[l][j]); a.dst[0] = c.at(dx);
            a.CD = CI; a.cout_real = j == 0 ? 8 : CI; a.dst_nstride = pm_image_elems(hl, wl, CI);
            if (j > 0) { a.aux[0] = c.at(sp.sx[l][j]); a.mask_mode = MASK_RELU; }
            CK(vsr_launch_conv(c.dtype, 7, 1, CO, CO, 0, SPY_DROWS[j], EPI_NHWC, a, c.st));
            dy = dx;
        }
        float* dfr = dframes ? (float*)c.at(sp.dpyr[l]) : nullptr;
        if (l == 0) {
            if (dfr) CK(vsr_launch_spynet_prepare_bwd(c.dtype, c.at(dy), nullptr, c.fat(sp.pyr[0]), nullptr, nullptr, dfr, n, t, P, pair_mode, hl, wl, c.st));
            break;
        }
        // x16 = [ref | warp(supp, flow_up) | flow_up], flow_up = 2 * up(flow_{l-1}): everything that reaches flow_{l-1}
        HIP_CHECK_RET(hipMemsetAsync(c.at(dprev), 0, (size_t)P * 2 * (hl / 2) * (wl / 2) * 4, c.st));
        CK(vsr_launch_spynet_prepare_bwd(c.dtype, c.at(dy), c.fat(dcur), c.fat(sp.pyr[l]), c.fat(sp.sfup[l]), (float*)c.at(dprev), dfr,
                                         n, t, P, pair_mode, hl, wl, c.st));
        const size_t tmp = dcur; dcur = dprev; dprev = tmp;
    }
    if (dframes) {   // pyramid adjoint: avg_pool2d from fine to coarse (spynet.py:44-45), then the /32 resize + normalisation
        for (int l = 1; l < 6; ++l)
            CK(vsr_launch_avgpool2_bwd_add(c.fat(sp.dpyr[l - 1]), (float*)c.at(sp.dpyr[l]), (long long)sp.F * 3, hu >> (5 - l), wu >> (5 - l), c.st));
        CK(vsr_launch_resize_norm_bwd(c.fat(sp.dpyr[5]), dframes, std, sp.F, sp.h, sp.w, hu, wu, c.st));
    }
    return VSR_OK;
}

// Weight gradients of PixelShufflePack k (upsample.k, upsampling.py:7) for the frames [f0, f1): one launch per pixel-shuffle phase z,
// X = the layer's input, dY = phase z of the gradient into it (a contiguous plane when p.unsh, else every second pixel of every second row)
int recon_ps_wgrads(const Ctx& c, const Plan& p, int k, int f0, int f1, float* const* g) {
    const PIdx ix{p.rb, p.ups};
    const WG wg{c};
    const int n = p.n, H = (k + 1) * p.h, W = (k + 1) * p.w;              // the layer's input size: h x w (k = 0), 2h x 2w (k = 1)
    for (int i0 = f0; i0 < f1; i0 += VSR_WG_MAXSEG) {
        const int i1 = i0 + VSR_WG_MAXSEG < f1 ? i0 + VSR_WG_MAXSEG : f1;
        for (int z = 0; z < 4; ++z) {
            WgradArgs a = wg_base(n, H, W);
            a.nseg = 0;
            for (int i = i0; i < i1; ++i) {
                a.x[a.nseg] = c.at(k == 1 ? p.U0[i] : p.Pt[i]);
                const char* gu = (const char*)c.at(k == 1 ? p.GU1f[i] : p.GU0f[i]);
                a.dy[a.nseg] = p.unsh ? gu + (size_t)z * Ctx::plane_elems(n, 2 * H, 2 * W) * p.es : gu;      // plane z: n x H x W, contiguous
                ++a.nseg;
            }
            if (!p.unsh) { a.dy_step = 2; a.dy_oy = z >> 1; a.dy_ox = z & 1; a.Hy = 2 * H; a.Wy = 2 * W; a.dy_nstride = pm_image_elems(2 * H, 2 * W, C); }
            CK(wg.run(3, 64, false, 64, false, a, C, C, g[ix.up_w(k)], C, 0, 4, z, g[ix.up_b(k)]));
        }
    }
    return VSR_OK;
}
// ... and of the 1x1 fuse conv on cat([outputs[i], feat_prop]) (basicvsr.py:18,75-77): one launch per 64-channel half of its input
int recon_point_wgrads(const Ctx& c, const Plan& p, int f0, int f1, float* const* g) {
    const PIdx ix{p.rb, p.ups};
    const WG wg{c};
    for (int i0 = f0; i0 < f1; i0 += VSR_WG_MAXSEG) {
        const int i1 = i0 + VSR_WG_MAXSEG < f1 ? i0 + VSR_WG_MAXSEG : f1;
        for (int s = 0; s < 2; ++s) {
            WgradArgs a = wg_base(p.n, p.h, p.w);
            a.nseg = 0;
            for (int i = i0; i < i1; ++i) { a.x[a.nseg] = c.at(p.feat[s][i]); a.dy[a.nseg] = c.at(p.GPf[i]); ++a.nseg; }
            CK(wg.run(1, 64, false, 64, false, a, C, C, g[ix.point_w()], 2 * C, s * C, 1, 0, s == 0 ? g[ix.point_b()] : nullptr));
        }
    }
    return VSR_OK;
}

int recon_backward(const Ctx& c, const Plan& p, int i, const float* lrs, const float* dsr, float* const* g, const float* last2_w) {
    const PIdx ix{p.rb, p.ups};
    const WG wg{c};
    if (p.diet) {   // U0, U1 and C0 of this frame were not kept: the forward's three launches again (the sign bits of C0 were)
        CK(c.conv_ps(c.at(p.Pt[i]), p.up_w[0], c.fat(p.up_b[0]), c.at(p.U0[i]), p.n, p.h, p.w));
        if (p.ups == 2) CK(c.conv_ps(c.at(p.U0[i]), p.up_w[1], c.fat(p.up_b[1]), c.at(p.U1[i]), p.n, 2 * p.h, 2 * p.w));
        CK(c.conv64(c.at(p.U1[i]), p.last0_w, c.fat(p.last0_b), c.at(p.C0[i]), ACT_LEAKY, nullptr, nullptr, 0, p.n, p.scale * p.h, p.scale * p.w, nullptr));
    }
    const int n = p.n, h = p.h, w = p.w, H4 = p.scale * h, W4 = p.scale * w;      // (the output size: 4h x 4w, or 2h x 2w for upscale 2)
    const float* dsr_i = dsr + (size_t)i * 3 * H4 * W4;
    const long long dsr_ns = (long long)p.t * 3 * H4 * W4;
    if (c.dtype == VSR_BF16 && last2_w) {   // d(conv_last.0 pre-activation) = dgrad(conv_last.2)(dsr) * LeakyReLU'(C0): hr_tail.hip
        CK(vsr_launch_last2_dgrad(dsr_i, dsr_ns, last2_w, c.at(p.C0[i]), c.at(p.G_C0), n, H4, W4, MASK_LEAKY, c.st, c.at(p.SBC0[i])));
    } else {
        ConvArgs a = c.base(n, H4, W4);
        a.src[0] = dsr_i; a.src_nstride[0] = dsr_ns; a.wpack = c.at(p.last2_wd); a.dst[0] = c.at(p.G_C0);
        a.aux[0] = c.at(p.C0[i]); a.mask_mode = MASK_LEAKY;
        CK(vsr_launch_conv(c.dtype, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, c.st));
    }
    {   // conv_last.2: X = C0, dY = dsr (planar)
        WgradArgs a = wg_base(n, H4, W4);
        a.x[0] = c.at(p.C0[i]); a.dy[0] = dsr_i; a.dy_nstride = dsr_ns;
        CK(wg.run(3, 64, false, 16, true, a, 3, C, g[ix.last2_w()], C, 0, 1, 0, g[ix.last2_b()]));
    }
    // (r04: G_U1 / G_U0, the gradients into the pixel-shuffle layers, are written phase-separated when p.unsh)
    const size_t gu1 = p.GU1f[i], gu0 = p.GU0f[i], gp = p.GPf[i];     // (per frame when p.hrdef: their weight gradients run later, all frames per launch)
    CK(c.conv64(c.at(p.G_C0), p.last0_wd, nullptr, c.at(gu1), ACT_NONE, nullptr, nullptr, 0, n, H4, W4, nullptr, nullptr, p.unsh));
    {   // conv_last.0: X = U1, dY = G_C0
        WgradArgs a = wg_base(n, H4, W4);
        a.x[0] = c.at(p.U1[i]); a.dy[0] = c.at(p.G_C0);
        CK(wg.run(3, 64, false, 64, false, a, C, C, g[ix.last0_w()], C, 0, 1, 0, g[ix.last0_b()]));
    }
    // upsample.1 (at 2h x 2w; upscale 4 only -- for upscale 2 G_U1 IS G_U0): dgrad, then wgrad per pixel-shuffle phase z
    if (p.ups == 2) {
        CK(c.conv_ps_dgrad(c.at(gu1), p.up_wd[1], c.at(gu0), nullptr, 0, n, 2 * h, 2 * w, nullptr, p.unsh, p.unsh));
        if (!p.hrdef) CK(recon_ps_wgrads(c, p, 1, i, i + 1, g));
    }
    // upsample.0 (at h x w): its input is LeakyReLU(point_conv) => mask with P
    CK(c.conv_ps_dgrad(c.at(gu0), p.up_wd[0], c.at(gp), c.at(p.Pt[i]), MASK_LEAKY, n, h, w, c.dtype == VSR_BF16 ? c.at(p.SBPt[i]) : nullptr, p.unsh, false));
    if (!p.hrdef) CK(recon_ps_wgrads(c, p, 0, i, i + 1, g));
    {   // point_conv dgrad: two 64-channel outputs (d outputs[i], d feat_prop)
        ConvArgs a = c.base(n, h, w);
        a.src[0] = c.at(gp); a.wpack = c.at(p.point_wd); a.w_zstride = C * C; a.nz = 2;
        a.dst[0] = c.at(p.dFeatB[i]); a.dst[1] = c.at(p.dFF[i]);
        CK(vsr_launch_conv(c.dtype, 1, 1, 64, 64, 0, 64, EPI_NHWC, a, c.st));
    }
    if (!p.hrdef) CK(recon_point_wgrads(c, p, i, i + 1, g));
    return VSR_OK;
}

// BPTT through one ResidualBlock call; top gradient = dtop (T) + S (fp32 scatter, optional)
// pending_flow: the flow of the warp whose output gradient sits in dWp[dir] (from the previous frame of the chain), or null:
// the top gradient of this frame is dtop + warp^T(dWp) -- gather form, fused with the cast (elementwise.hip)
int stem_wgrads(const Ctx& c, const Plan& p, int dir, int f0, int f1, const float* lrs, float* const* g);
int block_wgrads(const Ctx& c, const Plan& p, int dir, int b, int f0, int f1, float* const* g);
// diet (Plan::diet): the frame's weight gradients are launched here, block by block behind the data gradients (one frame per
// launch, accumulated into g), because the activation gradients are not kept for backward_chain's all-frames launches
int trunk_backward(const Ctx& c, const Plan& p, int dir, int i, const void* dtop, const float* pending_flow, int pending_k, bool has_warp,
                   const float* lrs, float* const* g) {
    const int n = p.n, h = p.h, w = p.w, rb = p.rb;
    if (pending_flow) {
        const long long fstride = (long long)(p.t - 1) * 2 * h * w;
        CK(vsr_launch_warp_bwd_gather(c.dtype, c.at(p.dWp[dir]), pending_flow, dtop, (long long*)c.at(p.S[dir]),
                                      (int*)c.at(p.far_cnt) + dir * 32 + pending_k, c.at(p.dxoff(dir, i, rb)), n, h, w, fstride, c.st));
    } else {
        CK(vsr_launch_add_cast(c.dtype, dtop, nullptr, c.at(p.dxoff(dir, i, rb)), n, h, w, C, c.st));
    }
    const bool chain = chain_on(p) && !p.diet;            // diet: the activation gradients live in a two-block ring, the frame's weight
    if (chain) CK(trunk_chain_backward(c, p, dir, i));     // gradients are launched block by block behind them (the forward chain stays)
    for (int b = chain ? 0 : rb - 1; b >= 0; --b) {
        const void* dxn = c.at(p.dxoff(dir, i, b + 1));
        // dA = dgrad(conv2)(dX_{b+1}) * ReLU'(A_b)
        if (!chain) CK(c.conv64(dxn, p.blk_wd[dir][2 * b + 1], nullptr, c.at(p.g1off(dir, i, b)), ACT_NONE, nullptr, c.at(p.aoff(dir, i, b)), MASK_RELU, n, h, w,
                    nullptr, c.dtype == VSR_BF16 ? c.at(p.sboff(dir, i, b)) : nullptr));
        // dX_b = dX_{b+1} + dgrad(conv1)(dA); for b == 0 also through the stem's LeakyReLU
        void* out = b > 0 ? c.at(p.dxoff(dir, i, b)) : c.at(p.G0[dir][i]);
        CK(c.conv64(c.at(p.g1off(dir, i, b)), p.blk_wd[dir][2 * b], nullptr, out, ACT_NONE, dxn, b == 0 ? c.at(p.xoff(dir, i, 0)) : nullptr,
                    b == 0 ? MASK_LEAKY : 0, n, h, w, nullptr, (b == 0 && c.dtype == VSR_BF16) ? c.at(p.SBX0[dir][i]) : nullptr));
        if (p.diet) CK(block_wgrads(c, p, dir, b, i, i + 1, g));
    }
    if (has_warp)   // gradient w.r.t. the warped state (feat part of the stem's input)
        CK(c.conv64(c.at(p.G0[dir][i]), p.stem_wd[dir], nullptr, c.at(p.dWp[dir]), ACT_NONE, nullptr, nullptr, 0, n, h, w));
    if (p.diet) CK(stem_wgrads(c, p, dir, i, i + 1, lrs, g));
    return VSR_OK;
}

// Weight gradients of one direction's trunk for the frames [f0, f1) (at most VSR_WG_MAXSEG per launch): the stem ...
int stem_wgrads(const Ctx& c, const Plan& p, int dir, int f0, int f1, const float* lrs, float* const* g) {
    const PIdx ix{p.rb, p.ups};
    const WG wg{c};
    const int n = p.n, t = p.t, h = p.h, w = p.w;
    {   // LR part (+ bias)
        WgradArgs a = wg_base(n, h, w);
        a.nseg = 0;
        for (int i = f0; i < f1; ++i) { a.x[a.nseg] = lrs + (size_t)i * 3 * h * w; a.dy[a.nseg] = c.at(p.G0[dir][i]); ++a.nseg; }
        a.x_nstride = (long long)t * 3 * h * w;
        CK(wg.run(3, 16, true, 64, false, a, C, 3, g[ix.stem_w(dir)], C + 3, 0, 1, 0, g[ix.stem_b(dir)]));
    }
    {   // feat part: only frames that had a warped state
        WgradArgs a = wg_base(n, h, w);
        a.nseg = 0;
        for (int i = f0; i < f1; ++i) {
            const bool has = dir == 0 ? (i < t - 1) : (i > 0);
            if (has) { a.x[a.nseg] = c.at(p.Wp[dir][i]); a.dy[a.nseg] = c.at(p.G0[dir][i]); ++a.nseg; }
        }
        if (a.nseg) CK(wg.run(3, 64, false, 64, false, a, C, C, g[ix.stem_w(dir)], C + 3, 3, 1, 0, nullptr));
    }
    return VSR_OK;
}
// ... and the two convs of ResidualConv block b
int block_wgrads(const Ctx& c, const Plan& p, int dir, int b, int f0, int f1, float* const* g) {
    const PIdx ix{p.rb, p.ups};
    const WG wg{c};
    WgradArgs a1 = wg_base(p.n, p.h, p.w), a2 = wg_base(p.n, p.h, p.w);
    a1.nseg = a2.nseg = 0;
    for (int i = f0; i < f1; ++i) {
        a1.x[a1.nseg] = c.at(p.xoff(dir, i, b)); a1.dy[a1.nseg] = c.at(p.g1off(dir, i, b)); ++a1.nseg;
        a2.x[a2.nseg] = c.at(p.aoff(dir, i, b)); a2.dy[a2.nseg] = c.at(p.dxoff(dir, i, b + 1)); ++a2.nseg;
    }
    CK(wg.run(3, 64, false, 64, false, a1, C, C, g[ix.blk_w(dir, 2 * b)], C, 0, 1, 0, g[ix.blk_b(dir, 2 * b)]));
    CK(wg.run(3, 64, false, 64, false, a2, C, C, g[ix.blk_w(dir, 2 * b + 1)], C, 0, 1, 0, g[ix.blk_b(dir, 2 * b + 1)]));
    return VSR_OK;
}
int trunk_wgrads(const Ctx& c, const Plan& p, int dir, const float* lrs, float* const* g) {
    for (int i0 = 0; i0 < p.t; i0 += VSR_WG_MAXSEG) {
        const int i1 = i0 + VSR_WG_MAXSEG < p.t ? i0 + VSR_WG_MAXSEG : p.t;
        CK(stem_wgrads(c, p, dir, i0, i1, lrs, g));
        for (int b = 0; b < p.rb; ++b) CK(block_wgrads(c, p, dir, b, i0, i1, g));
    }
    return VSR_OK;
}

// BPTT through one direction's chain, then that direction's weight gradients (all frames per launch)
int backward_chain(const Ctx& c, const Plan& p, int dir, const float* lrs, float* const* g) {
    const int n = p.n, t = p.t, h = p.h, w = p.w;
    const long long fstride = (long long)(t - 1) * 2 * h * w;
    // the gather-form warp backward keeps S all-zero between uses: ONE memset per chain (was one per frame)
    HIP_CHECK_RET(hipMemsetAsync(c.at(p.S[dir]), 0, p.s_elems * 8, c.st));
    HIP_CHECK_RET(hipMemsetAsync((int*)c.at(p.far_cnt) + dir * 32, 0, 32 * 4, c.st));
    const float* pending = nullptr;
    for (int k = 0; k < t; ++k) {
        // the state of dir 1 flows 0 -> t-1, so its gradient flows t-1 -> 0; dir 0 the other way round
        const int i = dir == 1 ? t - 1 - k : k;
        const void* dtop = dir == 1 ? c.at(p.dFF[i]) : c.at(p.dFeatB[i]);
        const bool last = k == t - 1;                     // the chain's first frame had no warped state
        CK(trunk_backward(c, p, dir, i, dtop, pending, k, !last, lrs, g));
        pending = nullptr;
        if (!last) {   // feat(i) = trunk(warp(feat(prev), flow)): the gradient of the warped state goes back through the warp
            pending = flow_ptr(c, p, dir, dir == 1 ? i - 1 : i);     // ... at the top of the next frame's trunk_backward
            if (p.flowgrad) {   // the same warp's gradient w.r.t. its flow (each flow is used by exactly one warp)
                const int prev = dir == 1 ? i - 1 : i + 1, fi = dir == 1 ? i - 1 : i;
                float* df = (float*)c.at(p.dflows) + ((size_t)dir * p.n * (p.t - 1) + fi) * 2 * p.h * p.w;
                CK(vsr_launch_warp_bwd_flow(c.dtype, c.at(p.feat[dir][prev]), c.at(p.dWp[dir]), flow_ptr(c, p, dir, fi), df, n, h, w, C, fstride, c.st));
            }
        }
    }
    return p.diet ? VSR_OK : trunk_wgrads(c, p, dir, lrs, g);
}

int backward_impl(const Plan& p, const float* const* prm, float* const* g, const float* lrs, const float* dsr, float* dlrs,
                  char* ws, hipStream_t st) {
    const Ctx c{p, ws, st, p.dtype, 0};
    // input gradient, part 1: the bilinear x4 skip (basicvsr.py:22,82) -- overwrites dlrs, everything else accumulates
    if (dlrs) CK(vsr_launch_bilinear4_bwd(dsr, dlrs, (long long)p.n * p.t * 3, p.h, p.w, st, p.scale));
    for (int i = p.t - 1; i >= 0; --i) CK(recon_backward(c, p, i, lrs, dsr, g, prm ? prm[PIdx{p.rb, p.ups}.last2_w()] : nullptr));   // -> dFeatB[i], dFF[i]
    if (p.hrdef) {            // the weight gradients recon_backward left out: all frames per launch
        if (p.ups == 2) CK(recon_ps_wgrads(c, p, 1, 0, p.t, g));
        CK(recon_ps_wgrads(c, p, 0, 0, p.t, g));
        CK(recon_point_wgrads(c, p, 0, p.t, g));
    }
    Fork f{st, nullptr};
    // One stream: the two directions share their activation-gradient buffers (Plan::build), and chain launches of two streams could
    // each hold the CUs the other's unstarted workgroups need (conv3x3_chain.hip).  diet: rings per direction, no chains: two streams.
    CK(f.begin(p.diet));
    const Ctx c0{p, ws, st, p.dtype, 0}, c1{p, ws, f.side(), p.dtype, 1};
    CK(backward_chain(c0, p, 1, lrs, g));
    CK(backward_chain(c1, p, 0, lrs, g));
    CK(f.end());
    if (dlrs) {   // part 2: the stems' LR channels (conv.py:97 on cat([lr_i, feat])), both directions, every frame
        for (int dir = 0; dir < 2; ++dir)
            for (int i = 0; i < p.t; ++i) {
                ConvArgs a = c.base(p.n, p.h, p.w);
                a.src[0] = c.at(p.G0[dir][i]); a.wpack = c.at(p.stem_wd_lr[dir]); a.cout_real = 3;
                float* d = dlrs + (size_t)i * 3 * p.h * p.w;
                a.dst[0] = d; a.pres = d; a.dst_nstride = (long long)p.t * 3 * p.h * p.w;
                CK(vsr_launch_conv(c.dtype, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, c.st));
            }
    }
    if (p.flowgrad && p.t > 1)   // part 3 (and train_flow): through the flows into SPyNet's parameters / image pyramid
        CK(spynet_backward(c, p.spy, c.fat(p.dflows), p.n, p.t, 0, g, PIdx{p.rb, p.ups}.spy_base(), dlrs, prm[PIdx{p.rb, p.ups}.spy_std()]));
    return VSR_OK;
}

}  // namespace

// ================================ C ABI =========================================================
extern "C" {

int vsr_abi_version(void) { return 4; }

const char* vsr_status_string(int s) {
    switch (s) {
        case VSR_OK: return "ok";
        case VSR_ERR_BADARG: return "bad argument";
        case VSR_ERR_UNSUPPORTED: return "unsupported shape/configuration for the HIP path";
        case VSR_ERR_HIP: return "HIP runtime error";
        case VSR_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

int vsr_basicvsr_num_params(const VsrBasicVSRDesc* d) {
    if (!d || d->res_blocks < 0) return VSR_ERR_BADARG;
    return PIdx{d->res_blocks, d->upscale == 2 ? 1 : 2}.count();
}

size_t vsr_basicvsr_workspace_bytes(const VsrBasicVSRDesc* d, int need_backward) {
    if (!d) return 0;
    Plan p;
    if (p.build(*d, need_backward) != VSR_OK) return 0;
    return p.total;
}

int vsr_basicvsr_forward(const VsrBasicVSRDesc* d, const float* const* params, int nparams, const float* lrs, float* sr,
                         void* workspace, size_t workspace_bytes, int need_backward, void* stream) {
    if (!d || !params || !lrs || !sr || !workspace) return VSR_ERR_BADARG;
    Plan p;
    CK(p.build(*d, need_backward));
    if (nparams != PIdx{p.rb, p.ups}.count()) return VSR_ERR_BADARG;
    for (int k = 0; k < nparams; ++k) if (!params[k]) return VSR_ERR_BADARG;
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    return forward_impl(p, params, lrs, sr, (char*)workspace, (hipStream_t)stream);
}

int vsr_basicvsr_backward(const VsrBasicVSRDesc* d, const float* const* params, float* const* grads, int nparams,
                          const float* lrs, const float* dsr, float* dlrs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !params || !grads || !lrs || !dsr || !workspace) return VSR_ERR_BADARG;
    if (d->res_blocks < 1) return VSR_ERR_UNSUPPORTED;
    const PIdx ix{d->res_blocks, d->upscale == 2 ? 1 : 2};
    if (nparams != ix.count()) return VSR_ERR_BADARG;
    bool flow = dlrs != nullptr;                           // input or SPyNet gradient wanted => the forward ran with need_backward = 2
    for (int k = ix.spy_base(); k < ix.spy_mean(); ++k) flow = flow || grads[k];
    Plan p;
    CK(p.build(*d, flow ? 2 : 1));
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    return backward_impl(p, params, grads, lrs, dsr, dlrs, (char*)workspace, (hipStream_t)stream);
}

int vsr_basicvsr_get_flows(const VsrBasicVSRDesc* d, const void* workspace, float* flow_forward, float* flow_backward, void* stream) {
    if (!d || !workspace) return VSR_ERR_BADARG;
    Plan p;
    CK(p.build(*d, 0));         // weight/flow offsets do not depend on need_backward
    if (p.t < 2) return VSR_OK;
    const size_t half = (size_t)p.n * (p.t - 1) * 2 * p.h * p.w * 4;
    const char* f = (const char*)workspace + p.flows;
    if (flow_backward) HIP_CHECK_RET(hipMemcpyAsync(flow_backward, f, half, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (flow_forward) HIP_CHECK_RET(hipMemcpyAsync(flow_forward, f + half, half, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VSR_OK;
}

// ---- SPyNet alone ---------------------------------------------------------------------------------
struct SpyAlone { SpyPlan sp; size_t frames, slab, dframes; size_t total; };
static SpyAlone spy_alone_plan(int N, int h, int w, int dtype, bool save) {
    SpyAlone s; Bump b;
    s.frames = b.take((size_t)2 * N * 3 * h * w * 4);
    s.sp.plan(b, N, 2 * N, h, w, dtype);
    s.slab = 0;
    if (save) {                                  // appended: the forward-only offsets do not move
        s.sp.plan_save(b, dtype);
        int cp, xp, stride;
        vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
        s.slab = b.take((size_t)VSR_WGRAD_NWG * stride * 4);
        s.dframes = b.take((size_t)2 * N * 3 * h * w * 4);
    }
    s.total = b.off;
    return s;
}

size_t vsr_spynet_workspace_bytes(int N, int h, int w, int dtype, int need_backward) {
    if (N < 1 || h < 1 || w < 1) return 0;
    return spy_alone_plan(N, h, w, dtype, need_backward != 0).total;
}

int vsr_spynet_forward(int N, int h, int w, int dtype, const float* const* params, int nparams, const float* ref,
                       const float* supp, float* flow, void* workspace, size_t workspace_bytes, int need_backward, void* stream) {
    if (N < 1 || h < 1 || w < 1 || !params || nparams != 62 || !ref || !supp || !flow || !workspace) return VSR_ERR_BADARG;
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    const SpyAlone s = spy_alone_plan(N, h, w, dtype, need_backward != 0);
    if (workspace_bytes < s.total) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Plan dummy; dummy.es = esize(dtype);
    const Ctx c{dummy, (char*)workspace, st, dtype};
    const size_t fb = (size_t)N * 3 * h * w * 4;
    HIP_CHECK_RET(hipMemcpyAsync(c.at(s.frames), ref, fb, hipMemcpyDeviceToDevice, st));
    HIP_CHECK_RET(hipMemcpyAsync((char*)c.at(s.frames) + fb, supp, fb, hipMemcpyDeviceToDevice, st));
    CK(spynet_pack(c, s.sp, params, 0));
    return spynet_run(c, s.sp, c.fat(s.frames), params[60], params[61], N, 2, 1, flow);
}

/* SPyNet with the options of the reference's OTHER SpyNet classes: last_relu = 0 is the canonical network
 * (vsr/models/VRT/modules/spynet.py:68-157); level_out: HOST array of 6 device pointers (NULL entries = not wanted), level l
 * receives the flow of pyramid level l resized to (h >> (5-l), w >> (5-l)) (its `return_levels`).  Forward only.       */
int vsr_spynet_forward_ex(int N, int h, int w, int dtype, const float* const* params, int nparams, const float* ref,
                          const float* supp, int last_relu, float* const* level_out, void* workspace, size_t workspace_bytes,
                          int need_backward, void* stream) {
    if (N < 1 || h < 32 || w < 32 || !params || nparams != 62 || !ref || !supp || !level_out || !workspace) return VSR_ERR_BADARG;
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    const SpyAlone s = spy_alone_plan(N, h, w, dtype, need_backward != 0);
    if (workspace_bytes < s.total) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Plan dummy; dummy.es = esize(dtype);
    const Ctx c{dummy, (char*)workspace, st, dtype};
    const size_t fb = (size_t)N * 3 * h * w * 4;
    HIP_CHECK_RET(hipMemcpyAsync(c.at(s.frames), ref, fb, hipMemcpyDeviceToDevice, st));
    HIP_CHECK_RET(hipMemcpyAsync((char*)c.at(s.frames) + fb, supp, fb, hipMemcpyDeviceToDevice, st));
    CK(spynet_pack(c, s.sp, params, 0));
    return spynet_run(c, s.sp, c.fat(s.frames), params[60], params[61], N, 2, 1, nullptr, last_relu != 0, level_out);
}

int vsr_spynet_backward(int N, int h, int w, int dtype, float* const* grads, int nparams, const float* dflow, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (N < 1 || h < 1 || w < 1 || !grads || nparams != 62 || !dflow || !workspace) return VSR_ERR_BADARG;
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    for (int k = 0; k < 60; k += 2) if (!grads[k] && grads[k + 1]) return VSR_ERR_BADARG;   // a bias gradient comes with its weight's
    const SpyAlone s = spy_alone_plan(N, h, w, dtype, true);
    if (workspace_bytes < s.total) return VSR_ERR_WORKSPACE;
    Plan dummy; dummy.es = esize(dtype); dummy.slab[0] = dummy.slab[1] = s.slab;
    const Ctx c{dummy, (char*)workspace, (hipStream_t)stream, dtype};
    return spynet_backward(c, s.sp, dflow, N, 2, 1, grads, 0);
}

/* the same, plus the gradient w.r.t. the two input frames: dref, dsupp (N,3,h,w) fp32 are WRITTEN (either may be NULL).
 * params: the 62 tensors of the forward (std is needed for the normalisation's adjoint); grads may be NULL (frozen net). */
int vsr_spynet_backward_ex(int N, int h, int w, int dtype, const float* const* params, float* const* grads, int nparams, const float* dflow,
                           int last_relu, const float* const* dlevel, float* dref, float* dsupp, void* workspace, size_t workspace_bytes,
                           void* stream) {
    if (N < 1 || h < 1 || w < 1 || !params || nparams != 62 || !workspace || !params[61]) return VSR_ERR_BADARG;
    bool any = dflow != nullptr;
    if (dlevel) for (int l = 0; l < 6; ++l) any = any || dlevel[l];
    if (!any) return VSR_ERR_BADARG;
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    if (grads) for (int k = 0; k < 60; k += 2) if (!grads[k] && grads[k + 1]) return VSR_ERR_BADARG;
    const SpyAlone s = spy_alone_plan(N, h, w, dtype, true);
    if (workspace_bytes < s.total) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Plan dummy; dummy.es = esize(dtype); dummy.slab[0] = dummy.slab[1] = s.slab;
    const Ctx c{dummy, (char*)workspace, st, dtype};
    const bool want = dref || dsupp;
    const size_t fb = (size_t)N * 3 * h * w * 4;
    if (want) HIP_CHECK_RET(hipMemsetAsync(c.at(s.dframes), 0, 2 * fb, st));
    CK(spynet_backward(c, s.sp, dflow, N, 2, 1, grads, 0, want ? (float*)c.at(s.dframes) : nullptr, params[61], last_relu != 0, dlevel));
    if (dref) HIP_CHECK_RET(hipMemcpyAsync(dref, c.at(s.dframes), fb, hipMemcpyDeviceToDevice, st));
    if (dsupp) HIP_CHECK_RET(hipMemcpyAsync(dsupp, (char*)c.at(s.dframes) + fb, fb, hipMemcpyDeviceToDevice, st));
    return VSR_OK;
}

// ---- RealBasicVSR pre-clean stack, forward (realbasicvsr.py:17-30) ------------------------------------
// x <- x + conv(ResidualBlock(x)), `steps` times, on the F = n*t frames of the clip.
// params: resblock.conv.0.{weight,bias}, resblock.res_block.{i}.conv{1,2}.{weight,bias} ..., conv.{weight,bias}
struct CleanPlan {
    size_t stem_w, stem_b, out_w, out_b, feat, act, xa, xb, total;
    std::vector<size_t> blk_w, blk_b;
    // need_backward: data-gradient weights, per-step saved tensors, backward scratch
    bool save = false;
    size_t stem_wd, out_wd, slab, dXa, dXb, dA, G0, dxa, dxb;
    std::vector<size_t> blk_wd;
    std::vector<size_t> xs;             // [steps]: planar fp32 input of each step (xs[0] unused: the caller's lr)
    std::vector<size_t> X, A;           // [steps][blocks+1] / [steps][blocks]
};
static CleanPlan clean_plan(int F, int h, int w, int blocks, int dtype, int steps, bool save) {
    CleanPlan p; Bump b;
    const size_t es = esize(dtype), w64 = (size_t)9 * C * C * es;
    p.stem_w = b.take((size_t)9 * C * 16 * es); p.stem_b = b.take(C * 4);
    p.blk_w.resize(2 * blocks); p.blk_b.resize(2 * blocks);
    for (int k = 0; k < 2 * blocks; ++k) { p.blk_w[k] = b.take(w64); p.blk_b[k] = b.take(C * 4); }
    p.out_w = b.take((size_t)9 * 32 * C * es); p.out_b = b.take(64 * 4);
    const size_t a1 = (size_t)F * pm_image_elems(h, w, C) * es;
    const size_t x1 = (size_t)F * 3 * h * w * 4;
    p.feat = b.take(a1); p.act = b.take(a1);
    p.xa = b.take(x1); p.xb = b.take(x1);
    p.save = save;
    if (save) {                                   // appended: the forward-only offsets do not move
        p.stem_wd = b.take((size_t)9 * 32 * C * es); p.out_wd = b.take((size_t)9 * C * 16 * es);
        p.blk_wd.resize(2 * blocks);
        for (int k = 0; k < 2 * blocks; ++k) p.blk_wd[k] = b.take(w64);
        int cp, xp, stride;
        vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
        p.slab = b.take((size_t)VSR_WGRAD_NWG * stride * 4);
        p.dXa = b.take(a1); p.dXb = b.take(a1); p.dA = b.take(a1); p.G0 = b.take(a1);
        p.dxa = b.take(x1); p.dxb = b.take(x1);
        p.xs.assign(steps, 0); p.X.assign((size_t)steps * (blocks + 1), 0); p.A.assign((size_t)steps * blocks, 0);
        for (int s = 0; s < steps; ++s) {
            if (s > 0) p.xs[s] = b.take(x1);
            for (int k = 0; k <= blocks; ++k) p.X[(size_t)s * (blocks + 1) + k] = b.take(a1);
            for (int k = 0; k < blocks; ++k) p.A[(size_t)s * blocks + k] = b.take(a1);
        }
    }
    p.total = b.off;
    return p;
}

size_t vsr_cleaner_workspace_bytes(int F, int h, int w, int blocks, int steps, int dtype, int need_backward) {
    if (F < 1 || h < 1 || w < 1 || blocks < 0 || steps < 1) return 0;
    return clean_plan(F, h, w, blocks, dtype, steps, need_backward != 0).total;
}

int vsr_cleaner_forward(int F, int h, int w, int mid_channels, int blocks, int steps, int dtype, const float* const* params,
                        int nparams, const float* lr, float* lq, void* workspace, size_t workspace_bytes, int need_backward,
                        void* stream) {
    if (F < 1 || h < 1 || w < 1 || blocks < 0 || steps < 1 || !params || !lr || !lq || !workspace) return VSR_ERR_BADARG;
    if (mid_channels != C) return VSR_ERR_UNSUPPORTED;
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    if (nparams != 4 + 4 * blocks) return VSR_ERR_BADARG;
    const bool save = need_backward != 0;
    const CleanPlan p = clean_plan(F, h, w, blocks, dtype, steps, save);
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Plan dummy; dummy.es = esize(dtype);
    const Ctx c{dummy, (char*)workspace, st, dtype};
    CK(c.pack(params[0], p.stem_w, 9, C, 16, C, 3, 3, 0, 1, 0, 0));
    CK(c.pack_bias(params[1], p.stem_b, C));
    for (int k = 0; k < 2 * blocks; ++k) {
        CK(c.pack(params[2 + 2 * k], p.blk_w[k], 9, C, C, C, C, C, 0, 1, 0, 0));
        CK(c.pack_bias(params[3 + 2 * k], p.blk_b[k], C));
        if (save) CK(c.pack(params[2 + 2 * k], p.blk_wd[k], 9, C, C, C, C, C, 0, 1, 0, 1));
    }
    CK(c.pack(params[2 + 4 * blocks], p.out_w, 9, 32, C, 3, C, C, 0, 1, 0, 0));
    CK(c.pack_bias(params[3 + 4 * blocks], p.out_b, 3));
    if (save) {
        CK(c.pack(params[0], p.stem_wd, 9, 32, C, 3, C, 3, 0, 1, 0, 1));                  // 64 -> 3 (planar epilogue)
        CK(c.pack(params[2 + 4 * blocks], p.out_wd, 9, C, 16, C, 3, C, 0, 1, 0, 1));      // 3 (planar source) -> 64
    }
    const float* xin = lr;
    for (int s = 0; s < steps; ++s) {
        float* xout = (s == steps - 1) ? lq : (float*)c.at(save ? p.xs[s + 1] : ((s & 1) ? p.xb : p.xa));
        auto Xs = [&](int k) { return save ? c.at(p.X[(size_t)s * (blocks + 1) + k]) : c.at(p.feat); };
        {   // ResidualBlock stem: conv3x3 3->64 + LeakyReLU(0.1) on the planar frames (conv.py:97)
            ConvArgs a = c.base(F, h, w);
            a.src[0] = xin; a.src_nstride[0] = (long long)3 * h * w;
            a.wpack = c.at(p.stem_w); a.bias = c.fat(p.stem_b); a.dst[0] = Xs(0); a.act = ACT_LEAKY;
            CK(vsr_launch_conv(dtype, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, st));
        }
        for (int b = 0; b < blocks; ++b) {
            void* act = save ? c.at(p.A[(size_t)s * blocks + b]) : c.at(p.act);
            CK(c.conv64(Xs(b), p.blk_w[2 * b], c.fat(p.blk_b[2 * b]), act, ACT_RELU, nullptr, nullptr, 0, F, h, w));
            CK(c.conv64(act, p.blk_w[2 * b + 1], c.fat(p.blk_b[2 * b + 1]), Xs(b + 1), ACT_NONE, Xs(b), nullptr, 0, F, h, w));
        }
        {   // x + conv3x3 64->3 (realbasicvsr.py:28-29; a fresh tensor instead of the reference's in-place +=)
            ConvArgs a = c.base(F, h, w);
            a.src[0] = Xs(blocks); a.wpack = c.at(p.out_w); a.bias = c.fat(p.out_b); a.cout_real = 3;
            a.dst[0] = xout; a.dst_nstride = (long long)3 * h * w; a.pres = xin;
            CK(vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, st));
        }
        xin = xout;
    }
    return VSR_OK;
}

// Backward of the above for the cotangent dlq (F,3,h,w): grads[k] (the 4 + 4*blocks tensors; NULL = not wanted, a bias
// needs its weight's entry) are ACCUMULATED into; dlr (F,3,h,w, optional) is written.  The parameters are shared by the
// `steps` iterations, so each iteration adds its weight gradients.
int vsr_cleaner_backward(int F, int h, int w, int mid_channels, int blocks, int steps, int dtype, float* const* grads, int nparams,
                         const float* lr, const float* dlq, float* dlr, void* workspace, size_t workspace_bytes, void* stream) {
    if (F < 1 || h < 1 || w < 1 || blocks < 0 || steps < 1 || !grads || !lr || !dlq || !workspace) return VSR_ERR_BADARG;
    if (mid_channels != C || blocks < 1) return VSR_ERR_UNSUPPORTED;      // the stem's LeakyReLU mask is fused into block 0's dgrad
    if (dtype != VSR_F32 && dtype != VSR_BF16) return VSR_ERR_BADARG;
    if (nparams != 4 + 4 * blocks) return VSR_ERR_BADARG;
    const CleanPlan p = clean_plan(F, h, w, blocks, dtype, steps, true);
    if (workspace_bytes < p.total) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Plan dummy; dummy.es = esize(dtype); dummy.slab[0] = dummy.slab[1] = p.slab;
    const Ctx c{dummy, (char*)workspace, st, dtype};
    const WG wg{c};
    const float* dx = dlq;                                   // gradient w.r.t. x_{s+1}
    for (int s = steps - 1; s >= 0; --s) {
        const float* xs = s == 0 ? lr : c.fat(p.xs[s]);
        auto Xs = [&](int k) { return c.at(p.X[(size_t)s * (blocks + 1) + k]); };
        auto As = [&](int k) { return c.at(p.A[(size_t)s * blocks + k]); };
        float* gow = grads[2 + 4 * blocks]; float* gob = grads[3 + 4 * blocks];
        {   // out conv 64->3: X = X_blocks, dY = dx (planar)
            WgradArgs a = wg_base(F, h, w);
            a.x[0] = Xs(blocks); a.dy[0] = dx; a.dy_nstride = (long long)3 * h * w;
            CK(wg.run(3, 64, false, 16, true, a, 3, C, gow, C, 0, 1, 0, gob));
        }
        size_t dcur = p.dXa, dnext = p.dXb;
        {   // d X_blocks = dgrad(out conv)(dx)
            ConvArgs a = c.base(F, h, w);
            a.src[0] = dx; a.src_nstride[0] = (long long)3 * h * w; a.wpack = c.at(p.out_wd); a.dst[0] = c.at(dcur);
            CK(vsr_launch_conv(dtype, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, st));
        }
        for (int b = blocks - 1; b >= 0; --b) {   // x + conv2(relu(conv1(x)))   (conv.py:89-92)
            CK(c.conv64(c.at(dcur), p.blk_wd[2 * b + 1], nullptr, c.at(p.dA), ACT_NONE, nullptr, As(b), MASK_RELU, F, h, w));
            {
                WgradArgs a = wg_base(F, h, w);
                a.x[0] = As(b); a.dy[0] = c.at(dcur);
                CK(wg.run(3, 64, false, 64, false, a, C, C, grads[2 + 2 * (2 * b + 1)], C, 0, 1, 0, grads[3 + 2 * (2 * b + 1)]));
            }
            void* out = b > 0 ? c.at(dnext) : c.at(p.G0);      // b == 0: also through the stem's LeakyReLU
            CK(c.conv64(c.at(p.dA), p.blk_wd[2 * b], nullptr, out, ACT_NONE, c.at(dcur), b == 0 ? Xs(0) : nullptr,
                        b == 0 ? MASK_LEAKY : 0, F, h, w));
            {
                WgradArgs a = wg_base(F, h, w);
                a.x[0] = Xs(b); a.dy[0] = c.at(p.dA);
                CK(wg.run(3, 64, false, 64, false, a, C, C, grads[2 + 2 * (2 * b)], C, 0, 1, 0, grads[3 + 2 * (2 * b)]));
            }
            const size_t tmp = dcur; dcur = dnext; dnext = tmp;
        }
        const void* g0 = c.at(p.G0);
        {   // stem 3->64: X = x_s (planar), dY = G0
            WgradArgs a = wg_base(F, h, w);
            a.x[0] = xs; a.x_nstride = (long long)3 * h * w; a.dy[0] = g0;
            CK(wg.run(3, 16, true, 64, false, a, C, 3, grads[0], 3, 0, 1, 0, grads[1]));
        }
        const bool last = s == 0;
        if (!last || dlr) {   // d x_s = d x_{s+1} + dgrad(stem)(G0)
            float* dxs = last ? dlr : (float*)c.at((s & 1) ? p.dxa : p.dxb);
            ConvArgs a = c.base(F, h, w);
            a.src[0] = g0; a.wpack = c.at(p.stem_wd); a.cout_real = 3;
            a.dst[0] = dxs; a.dst_nstride = (long long)3 * h * w; a.pres = dx;
            CK(vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, st));
            dx = dxs;
        }
    }
    return VSR_OK;
}

// ---- per-op entry points -------------------------------------------------------------------------
static bool bad_dtype(int dtype) { return dtype != VSR_F32 && dtype != VSR_BF16; }
static bool bad_dims(int N, int H, int W) { return N < 1 || H < 1 || W < 1; }

int vsr_flow_warp_fwd(int dtype, const void* in_pm, const float* flow, void* out_pm, int N, int H, int W, int Cc, void* stream) {
    if (bad_dtype(dtype) || !in_pm || !flow || !out_pm || bad_dims(N, H, W) || Cc < 16 || (Cc & 15)) return VSR_ERR_BADARG;
    return vsr_launch_warp_fwd(dtype, in_pm, flow, out_pm, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream);
}
int vsr_flow_warp_bwd(int dtype, const void* dout_pm, const float* flow, float* dacc, int N, int H, int W, int Cc, void* stream) {
    if (bad_dtype(dtype) || !dout_pm || !flow || !dacc || bad_dims(N, H, W) || Cc < 16 || (Cc & 15)) return VSR_ERR_BADARG;
    return vsr_launch_warp_bwd(dtype, dout_pm, flow, dacc, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream);
}
int vsr_flow_warp_bwd_flow(int dtype, const void* in_pm, const void* dout_pm, const float* flow, float* dflow, int N, int H, int W,
                           int Cc, void* stream) {
    if (bad_dtype(dtype) || !in_pm || !dout_pm || !flow || !dflow || bad_dims(N, H, W) || Cc < 16 || (Cc & 15)) return VSR_ERR_BADARG;
    return vsr_launch_warp_bwd_flow(dtype, in_pm, dout_pm, flow, dflow, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream);
}
/* padding_mode: 0 = 'zeros' (the propagation warps), 1 = 'border' (the warps inside SPyNet, spynet.py:60) */
int vsr_flow_warp_fwd_ex(int dtype, const void* in_pm, const float* flow, void* out_pm, int N, int H, int W, int Cc, int padding_mode,
                         void* stream) {
    if (bad_dtype(dtype) || !in_pm || !flow || !out_pm || bad_dims(N, H, W) || Cc < 16 || (Cc & 15) || (padding_mode & ~1)) return VSR_ERR_BADARG;
    return vsr_launch_warp_fwd(dtype, in_pm, flow, out_pm, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream, padding_mode);
}
int vsr_flow_warp_bwd_ex(int dtype, const void* dout_pm, const float* flow, float* dacc, int N, int H, int W, int Cc, int padding_mode,
                         void* stream) {
    if (bad_dtype(dtype) || !dout_pm || !flow || !dacc || bad_dims(N, H, W) || Cc < 16 || (Cc & 15) || (padding_mode & ~1)) return VSR_ERR_BADARG;
    return vsr_launch_warp_bwd(dtype, dout_pm, flow, dacc, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream, padding_mode);
}
// Test hook (not in the header): the engine's GATHER form of the 64-channel zeros-padding warp adjoint on its own -- out = T(dtop + adjoint(dout)),
// near sources gathered, far ones through the 64-bit fixed-point accumulator S (N*H*W*64 long longs, all zero on entry and exit;
// far_count: one zeroed int).  tests: against the scatter form above, and that a non-finite far contribution stays non-finite.
extern "C" int vsr_debug_warp_bwd_gather(int dtype, const void* dout_pm, const float* flow, const void* dtop_pm, long long* S, int* far_count,
                                         void* out_pm, int N, int H, int W, void* stream) {
    if (bad_dtype(dtype) || !dout_pm || !flow || !S || !far_count || !out_pm || bad_dims(N, H, W)) return VSR_ERR_BADARG;
    return vsr_launch_warp_bwd_gather(dtype, dout_pm, flow, dtop_pm, S, far_count, out_pm, N, H, W, (long long)2 * H * W, (hipStream_t)stream);
}
int vsr_flow_warp_bwd_flow_ex(int dtype, const void* in_pm, const void* dout_pm, const float* flow, float* dflow, int N, int H, int W,
                              int Cc, int padding_mode, void* stream) {
    if (bad_dtype(dtype) || !in_pm || !dout_pm || !flow || !dflow || bad_dims(N, H, W) || Cc < 16 || (Cc & 15) || (padding_mode & ~1)) return VSR_ERR_BADARG;
    return vsr_launch_warp_bwd_flow(dtype, in_pm, dout_pm, flow, dflow, N, H, W, Cc, (long long)2 * H * W, (hipStream_t)stream, padding_mode);
}
int vsr_planar_to_pm(int dtype, const float* in, void* out_pm, int N, int Cin, int H, int W, int Cc, void* stream) {
    if (bad_dtype(dtype) || !in || !out_pm || bad_dims(N, H, W) || Cin < 1 || Cc < Cin || (Cc & 15)) return VSR_ERR_BADARG;
    return vsr_launch_planar_to_pm(dtype, in, out_pm, N, Cin, H, W, Cc, (hipStream_t)stream);
}
int vsr_pm_to_planar(int dtype, const void* in_pm, float* out, int N, int Cout, int H, int W, int Cc, void* stream) {
    if (bad_dtype(dtype) || !in_pm || !out || bad_dims(N, H, W) || Cout < 1 || Cc < Cout || (Cc & 15)) return VSR_ERR_BADARG;
    return vsr_launch_pm_to_planar(dtype, in_pm, out, N, Cout, H, W, Cc, (hipStream_t)stream);
}

static ConvArgs plain64(const void* x, const void* wpack, const float* b, void* y, int N, int H, int W) {
    ConvArgs a = {};
    a.in_step = 1; a.Hs = H; a.Ws = W; a.N = N; a.H = H; a.W = W; a.nz = 1; a.out_step = 1; a.Hd = H; a.Wd = W; a.CD = C; a.cout_real = C;
    a.dst_nstride = pm_image_elems(H, W, C); a.src_nstride[0] = pm_image_elems(H, W, C);
    a.src[0] = x; a.wpack = wpack; a.bias = b; a.dst[0] = y;
    return a;
}

int vsr_conv3x3_c64_fwd(int dtype, const void* x_pm, const float* w, const float* b, void* wpack, void* y_pm, const void* res_pm,
                        int act, int N, int H, int W, void* stream) {
    if (bad_dtype(dtype) || !x_pm || !wpack || !y_pm || bad_dims(N, H, W) || act < ACT_NONE || act > ACT_LEAKY) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (w) CK(vsr_launch_pack_weights(dtype, w, wpack, 9, C, C, C, C, C, 0, 1, 0, 0, st));   // w == NULL: wpack already packed
    ConvArgs a = plain64(x_pm, wpack, b, y_pm, N, H, W);
    a.act = act; a.res[0] = res_pm;
    return vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
}

size_t vsr_conv3x3_c64_chain_sync_bytes(int nlayers, int N, int H, int W) {
    if (nlayers < 1 || nlayers > VSR_CHAIN_MAX_LAYERS || bad_dims(N, H, W)) return 0;
    return vsr_chain_sync_bytes(nlayers, N, H, W);
}

int vsr_conv3x3_c64_chain_fwd(const void* images, const void* wpack, const float* bias, int nlayers, int N, int H, int W, void* sync,
                              void* stream) {
    if (!images || !wpack || !bias || !sync || nlayers < 1 || nlayers > VSR_CHAIN_MAX_LAYERS || bad_dims(N, H, W)) return VSR_ERR_BADARG;
    const size_t img = (size_t)N * pm_image_elems(H, W, C) * 2, wset = (size_t)9 * C * C * 2;
    // the kernel addresses every operand as a 32-bit offset (x 256 B) from one base
    uintptr_t lo = (uintptr_t)images;
    if ((uintptr_t)wpack < lo) lo = (uintptr_t)wpack;
    if ((uintptr_t)bias < lo) lo = (uintptr_t)bias;
    lo &= ~(uintptr_t)255;
    auto off = [&](const void* p, size_t add, unsigned* out) {
        const uintptr_t d = (uintptr_t)p + add - lo;
        if ((d & 255) || (d >> 8) >= 0xffffffffull) return false;
        *out = (unsigned)(d >> 8);
        return true;
    };
    ChainArgs a = {};
    a.base = (char*)lo; a.sync = (unsigned*)sync; a.N = N; a.H = H; a.W = W; a.nlayers = nlayers;
    for (int l = 0; l < nlayers; ++l) {
        ChainLayer& L = a.layer[l];
        L = {0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, (l & 1) ? CHAIN_SKIP : CHAIN_RELU};
        bool ok = off(images, (size_t)l * img, &L.src) && off(images, (size_t)(l + 1) * img, &L.dst) && off(wpack, (size_t)l * wset, &L.w) &&
                  off(bias, (size_t)l * C * 4, &L.bias);
        if (l & 1) ok = ok && off(images, (size_t)(l - 1) * img, &L.res);
        if (!ok) return VSR_ERR_UNSUPPORTED;
    }
    return vsr_launch_conv3x3_chain(a, vsr_num_cus(), (hipStream_t)stream);
}

int vsr_conv3x3_c64_dgrad(int dtype, const void* dy_pm, const float* w, void* wpack, void* dx_pm, const void* res_pm,
                          const void* aux_pm, int mask_mode, int N, int H, int W, void* stream) {
    if (bad_dtype(dtype) || !dy_pm || !w || !wpack || !dx_pm || bad_dims(N, H, W) || mask_mode < MASK_NONE || mask_mode > MASK_LEAKY ||
        (mask_mode != MASK_NONE && !aux_pm))
        return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    CK(vsr_launch_pack_weights(dtype, w, wpack, 9, C, C, C, C, C, 0, 1, 0, 1, st));
    ConvArgs a = plain64(dy_pm, wpack, nullptr, dx_pm, N, H, W);
    a.res[0] = res_pm; a.aux[0] = aux_pm; a.mask_mode = mask_mode;
    return vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
}

/* Backward of ONE conv layer of vsr_conv_layer_fwd (same shapes, same argument meaning): the reference's building blocks are ordinary
 * autograd modules (core/modules/conv.py:15-22,94-103, upsampling.py:4-12, spynet.py:13-21), so each layer needs its data gradient,
 * weight gradient and bias gradient on its own.  Every piece is a kernel the engines already run:
 *   dyM = dy * act'(y)           (mask_pm / spynet_dres; act'(y) from the layer's stored OUTPUT y: ReLU / LeakyReLU keep the sign)
 *   dx  = conv(dyM, flipped W)   (the forward kernels on weights packed with mode 1; pixel-shuffle: four phase launches)
 *   dlr = the stems' 3 planar LR channels (64 -> 3 planar kernel)
 *   gw, gb = wgrad(x, dyM)       (producer/consumer kernel for 3x3 64->64, generic kernel otherwise; reduced in a fixed order)
 * x_pm / lr_planar / y_* as given to / returned by the forward; dy_* has y's layout.  dx_pm, dlr_planar, gw, gb may be NULL (not
 * needed); gw / gb are OVERWRITTEN.  scratch: vsr_conv_layer_bwd_scratch_bytes(...) bytes.                                         */
size_t vsr_conv_layer_bwd_scratch_bytes(int dtype, int N, int H, int W, int pixel_shuffle) {
    if (bad_dtype(dtype) || bad_dims(N, H, W)) return 0;
    const size_t es = esize(dtype);
    const int s = pixel_shuffle ? 2 : 1;
    int cp3, xp3, stride3;
    vsr_wgrad_slab_dims(3, 64, 64, &cp3, &xp3, &stride3);
    return (size_t)49 * 64 * 64 * 4 * es + (size_t)N * pm_image_elems(s * H, s * W, C) * es + (size_t)VSR_WGRAD_NWG * stride3 * 4 + 1024;
}

int vsr_conv_layer_bwd(int dtype, int ks, const void* x_pm, int cin_pm, const float* lr_planar, const float* w, int cin_real, int cout_real,
                       const void* y_pm, const float* y_planar, const void* dy_pm, const float* dy_planar, int cd, int act, float slope,
                       int pixel_shuffle, void* dx_pm, float* dlr_planar, float* gw, float* gb, void* scratch, size_t scratch_bytes,
                       int N, int H, int W, void* stream) {
    if (bad_dtype(dtype) || !w || !scratch || bad_dims(N, H, W) || (!x_pm && !lr_planar) || (!dy_pm && !dy_planar) || act < ACT_NONE || act > ACT_LEAKY)
        return VSR_ERR_BADARG;
    if (act != ACT_NONE && !(dy_pm ? y_pm != nullptr : y_planar != nullptr)) return VSR_ERR_BADARG;     // the mask needs the layer's output
    if (scratch_bytes < vsr_conv_layer_bwd_scratch_bytes(dtype, N, H, W, pixel_shuffle)) return VSR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = esize(dtype);
    char* wp = (char*)scratch;
    char* dym = wp + (size_t)49 * 64 * 64 * 4 * es;
    const int sps = pixel_shuffle ? 2 : 1;
    float* slab = reinterpret_cast<float*>(dym + (((size_t)N * pm_image_elems(sps * H, sps * W, C) * es + 255) & ~(size_t)255));
    const float mslope = act == ACT_LEAKY ? vsr_slope(slope) : 0.f;

    // one weight-gradient launch + its reduction (gw / gb overwritten on the first call of a layer, accumulated by later ones)
    auto wgrad = [&](int wks, int cx, bool xp, int cout, WgradArgs& a, int co_real, int ci_real, int I_total, int i_off, int o_mul, int o_add,
                     float* gbias, int accumulate) -> int {
        int cp, xpd, stride;
        vsr_wgrad_slab_dims(wks, cx, cout, &cp, &xpd, &stride);
        a.slab = slab; a.slab_stride = stride;
        int cp3, xp3, stride3;
        vsr_wgrad_slab_dims(3, 64, 64, &cp3, &xp3, &stride3);
        const int tiles = a.N * cdiv(a.H, 8) * cdiv(a.W, 32);
        const long long cap = (long long)VSR_WGRAD_NWG * stride3 / stride;
        int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
        if (nwg > cap) nwg = (int)cap;
        if (nwg > 1) nwg &= ~1;
        int nslabs = 0;
        CK(vsr_launch_wgrad(dtype, wks, cx, xp, cout, false, a, nwg, &nslabs, st));
        return vsr_launch_wgrad_reduce(slab, nslabs, wks, cx, cout, co_real, ci_real, gw, I_total, i_off, o_mul, o_add, gbias, accumulate, st);
    };

    if (!dy_pm) {   // planar output: the 16 -> 2 SPyNet layer.  dyM as a 16-channel pixel-major tensor
        if (ks != 7 || cout_real != 2 || cin_pm != 16 || cin_real != 16 || !x_pm || pixel_shuffle) return VSR_ERR_UNSUPPORTED;
        if (act == ACT_RELU) CK(vsr_launch_spynet_dres(dtype, dy_planar, y_planar, dym, N, H, W, st));
        else if (act == ACT_NONE) CK(vsr_launch_planar_to_pm(dtype, dy_planar, dym, N, 2, H, W, 16, st));
        else return VSR_ERR_UNSUPPORTED;
    } else if (act != ACT_NONE) {
        const int cdy = pixel_shuffle ? C : cd;
        CK(vsr_launch_mask_pm(dtype, dy_pm, y_pm, dym, mslope, (long long)N * pm_image_elems(sps * H, sps * W, cdy), st));
    }
    const void* dyM = (!dy_pm || act != ACT_NONE) ? (const void*)dym : dy_pm;

    if (lr_planar) {                                       // stems: cat([lr(3), feat(64)]) or lr alone  (conv.py:97)
        if (ks != 3 || cout_real != C || cd != C || pixel_shuffle) return VSR_ERR_UNSUPPORTED;
        const bool cat = x_pm != nullptr;
        if ((cat && (cin_pm != C || cin_real != C + 3)) || (!cat && cin_real != 3)) return VSR_ERR_UNSUPPORTED;
        const int I_total = cat ? C + 3 : 3;
        if (cat && dx_pm) {
            CK(vsr_launch_pack_weights(dtype, w, wp, 9, C, C, C, C, I_total, 3, 1, 0, 1, st));
            ConvArgs a = plain64(dyM, wp, nullptr, dx_pm, N, H, W);
            CK(vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st));
        }
        if (dlr_planar) {
            CK(vsr_launch_pack_weights(dtype, w, wp, 9, 32, C, 3, C, I_total, 0, 1, 0, 1, st));
            ConvArgs a = plain64(dyM, wp, nullptr, dlr_planar, N, H, W);
            a.cout_real = 3; a.dst_nstride = (long long)3 * H * W;
            CK(vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 32, EPI_PLANAR, a, st));
        }
        if (gw) {
            WgradArgs a = wg_base(N, H, W);
            a.x[0] = lr_planar; a.x_nstride = (long long)3 * H * W; a.dy[0] = dyM;
            CK(wgrad(3, 16, true, 64, a, C, 3, I_total, 0, 1, 0, gb, 0));
            if (cat) {
                WgradArgs b2 = wg_base(N, H, W);
                b2.x[0] = x_pm; b2.dy[0] = dyM;
                CK(wgrad(3, 64, false, 64, b2, C, C, I_total, 3, 1, 0, nullptr, 0));
            }
        }
        return VSR_OK;
    }
    if (ks == 3 || ks == 1) {
        if (cin_pm != C || cin_real != C || !x_pm) return VSR_ERR_UNSUPPORTED;
        if (pixel_shuffle) {                               // conv3x3 64 -> 256 + PixelShuffle(2)  (upsampling.py:10-12)
            if (ks != 3 || cout_real != 4 * C || cd != C || act != ACT_NONE) return VSR_ERR_UNSUPPORTED;
            if (dx_pm) {
                for (int z = 0; z < 4; ++z) CK(vsr_launch_pack_weights(dtype, w, wp + (size_t)z * 9 * C * C * es, 9, C, C, C, C, C, 0, 4, z, 1, st));
                if (dtype == VSR_BF16) {
                    for (int z = 0; z < 4; ++z) {          // phase z reads phase z-1's partial sum as its residual, in place
                        ConvArgs a = plain64(dyM, wp + (size_t)z * 9 * C * C * es, nullptr, dx_pm, N, H, W);
                        a.in_step = 2; a.Hs = 2 * H; a.Ws = 2 * W; a.src_oy[0] = z >> 1; a.src_ox[0] = z & 1;
                        a.src_nstride[0] = pm_image_elems(2 * H, 2 * W, C);
                        a.res[0] = z > 0 ? dx_pm : nullptr;
                        CK(vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st));
                    }
                } else {
                    ConvArgs a = plain64(dyM, wp, nullptr, dx_pm, N, H, W);
                    a.in_step = 2; a.Hs = 2 * H; a.Ws = 2 * W;
                    for (int q = 0; q < 4; ++q) { a.src[q] = dyM; a.src_oy[q] = q >> 1; a.src_ox[q] = q & 1; a.src_nstride[q] = pm_image_elems(2 * H, 2 * W, C); }
                    CK(vsr_launch_conv(dtype, 3, 4, 64, 64, 0, 64, EPI_NHWC, a, st));
                }
            }
            if (gw)
                for (int z = 0; z < 4; ++z) {
                    WgradArgs a = wg_base(N, H, W);
                    a.x[0] = x_pm; a.dy[0] = dyM;
                    a.dy_step = 2; a.dy_oy = z >> 1; a.dy_ox = z & 1; a.Hy = 2 * H; a.Wy = 2 * W; a.dy_nstride = pm_image_elems(2 * H, 2 * W, C);
                    CK(wgrad(3, 64, false, 64, a, C, C, C, 0, 4, z, gb, 0));
                }
            return VSR_OK;
        }
        if (cout_real != C || cd != C) return VSR_ERR_UNSUPPORTED;
        if (dx_pm) {
            CK(vsr_launch_pack_weights(dtype, w, wp, ks * ks, C, C, C, C, C, 0, 1, 0, 1, st));
            ConvArgs a = plain64(dyM, wp, nullptr, dx_pm, N, H, W);
            CK(vsr_launch_conv(dtype, ks, 1, 64, 64, 0, 64, EPI_NHWC, a, st));
        }
        if (gw) {
            WgradArgs a = wg_base(N, H, W);
            a.x[0] = x_pm; a.dy[0] = dyM;
            CK(wgrad(ks, 64, false, 64, a, C, C, C, 0, 1, 0, gb, 0));
        }
        return VSR_OK;
    }
    if (ks != 7 || pixel_shuffle || !x_pm) return VSR_ERR_UNSUPPORTED;
    for (int j = 0; j < NSPY; ++j) {                       // the SPyNet layer shapes (spynet.py:16-18), as spynet_backward runs them
        if (cin_pm != SPY_CIP[j] || cout_real != SPY_CO[j] || cin_real != SPY_CI[j]) continue;
        const int CI = SPY_CIP[j], CO = SPY_DK[j];
        if (j < NSPY - 1 && cd != SPY_CD[j]) return VSR_ERR_BADARG;
        if (gw) {
            const int nhalf = (dtype == VSR_F32 && CI == 64) ? 2 : 1;          // fp32, 64 input channels: two 32-channel halves (LDS)
            for (int hf = 0; hf < nhalf; ++hf) {
                WgradArgs a = wg_base(N, H, W);
                a.x[0] = x_pm; a.x_nstride = pm_image_elems(H, W, CI);
                a.dy[0] = dyM; a.dy_nstride = pm_image_elems(H, W, CO);
                const int cx = CI / nhalf;
                if (nhalf == 2) { a.x_ctotal = CI; a.x_coff = hf * (cx / 8); }
                CK(wgrad(7, cx, false, CO, a, SPY_CO[j], nhalf == 2 ? cx : SPY_CI[j], SPY_CI[j], hf * cx, 1, 0, hf == 0 ? gb : nullptr, 0));
            }
        }
        if (dx_pm) {
            CK(vsr_launch_pack_weights(dtype, w, wp, 49, SPY_DROWS[j], SPY_DK[j], SPY_CI[j], SPY_CO[j], SPY_CI[j], 0, 1, 0, 1, st));
            ConvArgs a = {};
            a.in_step = 1; a.Hs = H; a.Ws = W; a.N = N; a.H = H; a.W = W; a.nz = 1; a.out_step = 1; a.Hd = H; a.Wd = W;
            a.src[0] = dyM; a.src_nstride[0] = pm_image_elems(H, W, CO);
            a.wpack = wp; a.dst[0] = dx_pm; a.CD = CI; a.cout_real = j == 0 ? 8 : CI; a.dst_nstride = pm_image_elems(H, W, CI);
            CK(vsr_launch_conv(dtype, 7, 1, CO, CO, 0, SPY_DROWS[j], EPI_NHWC, a, st));
        }
        return VSR_OK;
    }
    return VSR_ERR_UNSUPPORTED;
}

/* One convolution layer on blocked pixel-major tensors, forward only: the building blocks the reference's modules expose
 * on their own (ConvReLU core/modules/conv.py:15-22, SpynetModule spynet.py:13-21, PixelShufflePack upsampling.py:4-12,
 * the stem of ResidualBlock conv.py:97).  w: fp32 OIHW (cout_real, cin_real [+3 for lr_planar], ks, ks); b: cout_real or NULL.
 *   ks 3 or 1: x_pm has 64 channels, 64 outputs (pixel_shuffle: 256 outputs written as (N,2H,2W,64), upsampling.py:10-12);
 *   ks 7: (cin_pm, cout) in {(16,32), (32,64), (64,32), (32,16), (16,2 -> y_planar)} (the SPyNet layers);
 *   lr_planar (N,3,H,W) fp32, ks 3: the conv reads cat([lr, x]) (x_pm may be NULL = lr only): the trunk / pre-clean stems.
 * y_pm has cd channels per pixel (16 / 32 / 64); y_planar (N,cout_real,H,W) fp32 instead when cout_real <= 4.
 * wpack: scratch of 49 * 64 * 64 * 4 elements of `dtype` (packed weights of this call).                               */
int vsr_conv_layer_fwd(int dtype, int ks, const void* x_pm, int cin_pm, const float* lr_planar, const float* w, const float* b,
                       int cin_real, int cout_real, void* wpack, void* y_pm, int cd, float* y_planar, int act, float slope,
                       int pixel_shuffle, int N, int H, int W, void* stream) {
    if (bad_dtype(dtype) || !w || !wpack || bad_dims(N, H, W) || (!x_pm && !lr_planar) || (!y_pm && !y_planar) || act < ACT_NONE || act > ACT_LEAKY)
        return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = esize(dtype);
    float* bias = nullptr;
    char* wp = (char*)wpack;
    if (b) {   // the kernels read the bias as fp32 from device memory: the caller's tensor is used as is (cout_real values, padded reads are masked by cout_real)
        bias = const_cast<float*>(b);
    }
    ConvArgs a = {};
    a.in_step = 1; a.Hs = H; a.Ws = W; a.N = N; a.H = H; a.W = W; a.nz = 1; a.out_step = 1; a.Hd = H; a.Wd = W;
    a.act = act; a.leaky_slope = slope; a.bias = bias; a.wpack = wpack;
    if (lr_planar) {                                       // stems: cat([lr(3), feat(64)]) or lr alone
        if (ks != 3 || cout_real != C || !y_pm || cd != C || pixel_shuffle) return VSR_ERR_UNSUPPORTED;
        a.CD = C; a.cout_real = C; a.dst[0] = y_pm; a.dst_nstride = pm_image_elems(H, W, C);
        if (x_pm) {
            if (cin_pm != C || cin_real != C + 3) return VSR_ERR_UNSUPPORTED;
            CK(vsr_launch_pack_weights(dtype, w, wp, 9, C, C, C, C, C + 3, 3, 1, 0, 0, st));
            CK(vsr_launch_pack_weights(dtype, w, wp + (size_t)9 * C * C * es, 9, C, 16, C, 3, C + 3, 0, 1, 0, 0, st));
            a.src[0] = x_pm; a.src_nstride[0] = pm_image_elems(H, W, C);
            a.src[1] = lr_planar; a.src_nstride[1] = (long long)3 * H * W;
            return vsr_launch_conv(dtype, 3, 2, 64, 16, 1, 64, EPI_NHWC, a, st);
        }
        if (cin_real != 3) return VSR_ERR_UNSUPPORTED;
        CK(vsr_launch_pack_weights(dtype, w, wp, 9, C, 16, C, 3, 3, 0, 1, 0, 0, st));
        a.src[0] = lr_planar; a.src_nstride[0] = (long long)3 * H * W;
        return vsr_launch_conv(dtype, 3, 1, 16, 16, 1, 64, EPI_NHWC, a, st);
    }
    a.src[0] = x_pm; a.src_nstride[0] = pm_image_elems(H, W, cin_pm);
    if (ks == 3 || ks == 1) {
        if (cin_pm != C || cin_real != C) return VSR_ERR_UNSUPPORTED;
        if (pixel_shuffle) {
            if (ks != 3 || cout_real != 4 * C || !y_pm || cd != C) return VSR_ERR_UNSUPPORTED;
            for (int z = 0; z < 4; ++z) CK(vsr_launch_pack_weights(dtype, w, wp + (size_t)z * 9 * C * C * es, 9, C, C, C, C, C, 0, 4, z, 0, st));
            // PixelShuffle(2): out[c, 2y+i, 2x+j] = conv[4c + 2i + j]: sub-conv z uses rows 4c+z and bias entries 4c+z
            float* b4 = reinterpret_cast<float*>(wp + (size_t)4 * 9 * C * C * es);
            if (b) for (int z = 0; z < 4; ++z) CK(vsr_launch_pack_weights(VSR_F32, b, b4 + z * C, 1, C, 1, C, 1, 1, 0, 4, z, 0, st));
            a.bias = b ? b4 : nullptr; a.bias_zstride = C; a.nz = 4; a.w_zstride = 9 * C * C;
            a.out_step = 2; a.Hd = 2 * H; a.Wd = 2 * W; a.CD = C; a.cout_real = C; a.dst_nstride = pm_image_elems(2 * H, 2 * W, C);
            for (int z = 0; z < 4; ++z) { a.dst[z] = y_pm; a.out_oy[z] = z >> 1; a.out_ox[z] = z & 1; }
            return vsr_launch_conv(dtype, 3, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
        }
        if (cout_real != C || !y_pm || cd != C) return VSR_ERR_UNSUPPORTED;
        CK(vsr_launch_pack_weights(dtype, w, wp, ks * ks, C, C, C, C, C, 0, 1, 0, 0, st));
        a.CD = C; a.cout_real = C; a.dst[0] = y_pm; a.dst_nstride = pm_image_elems(H, W, C);
        return vsr_launch_conv(dtype, ks, 1, 64, 64, 0, 64, EPI_NHWC, a, st);
    }
    if (ks != 7 || pixel_shuffle) return VSR_ERR_UNSUPPORTED;
    for (int j = 0; j < NSPY; ++j) {
        if (cin_pm != SPY_CIP[j] || cout_real != SPY_CO[j] || cin_real != SPY_CI[j]) continue;
        CK(vsr_launch_pack_weights(dtype, w, wp, 49, SPY_COP[j], SPY_CIP[j], SPY_CO[j], SPY_CI[j], SPY_CI[j], 0, 1, 0, 0, st));
        a.cout_real = SPY_CO[j];
        if (j < NSPY - 1) {
            if (!y_pm || cd != SPY_CD[j]) return VSR_ERR_BADARG;
            a.dst[0] = y_pm; a.CD = SPY_CD[j]; a.dst_nstride = pm_image_elems(H, W, SPY_CD[j]);
            return vsr_launch_conv(dtype, 7, 1, SPY_CIP[j], SPY_CIP[j], 0, SPY_COP[j], EPI_NHWC, a, st);
        }
        if (!y_planar) return VSR_ERR_BADARG;
        a.dst[0] = y_planar; a.dst_nstride = (long long)2 * H * W;
        return vsr_launch_conv(dtype, 7, 1, 16, 16, 0, 32, EPI_PLANAR, a, st);
    }
    return VSR_ERR_UNSUPPORTED;
}

size_t vsr_conv3x3_c64_wgrad_slab_floats(void) {
    int cp, xp, stride;
    vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
    return (size_t)VSR_WGRAD_NWG * stride;
}

int vsr_conv3x3_c64_wgrad(int dtype, const void* x_pm, const void* dy_pm, float* gw, float* gb, float* slab, int N, int H, int W,
                          void* stream) {
    if (bad_dtype(dtype) || !x_pm || !dy_pm || !gw || !slab || bad_dims(N, H, W)) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a = wg_base(N, H, W);
    a.x[0] = x_pm; a.dy[0] = dy_pm;
    int cp, xp, stride;
    vsr_wgrad_slab_dims(3, 64, 64, &cp, &xp, &stride);
    a.slab = slab; a.slab_stride = stride;
    const int tiles = N * cdiv(H, 8) * cdiv(W, 32);
    const int nwg = tiles < VSR_WGRAD_NWG ? tiles : VSR_WGRAD_NWG;
    int nslabs = 0;
    CK(vsr_launch_wgrad(dtype, 3, 64, 0, 64, 0, a, nwg, &nslabs, st));
    return vsr_launch_wgrad_reduce(slab, nslabs, 3, 64, 64, C, C, gw, C, 0, 1, 0, gb, 0, st);
}

size_t vsr_charbonnier_scratch_floats(void) { return (size_t)vsr_charbonnier_scratch_floats_impl(); }
int vsr_charbonnier_fwd_bwd(const float* sr, const float* hr, float* dsr, float* loss, float* scratch, long long numel, float eps, void* stream) {
    if (!sr || !hr || !dsr || !loss || !scratch || numel < 1) return VSR_ERR_BADARG;
    return vsr_launch_charbonnier_grad(sr, hr, dsr, loss, scratch, numel, eps, (hipStream_t)stream);
}

}  // extern "C"
