// conv_last.2 (basicvsr.py:21: conv3x3 64 -> 3 at 4h x 4w) has 3 real output channels: its backward is a
// streaming problem, not a GEMM.  Dedicated bf16 kernels for the HR tail of the backward pass (the generic tiled
// kernel pads 3 channels to 16/32 and synchronises per tap: 0.9 ms per frame at 2160x3840 against an HBM floor of ~0.45).
#include "kernels.h"
#include <type_traits>

namespace {

__device__ float g_lt_zero[4];

struct __attribute__((aligned(8))) bf4_t { bf16_t v[4]; };

constexpr int LT_W = 32, LT_H = 8, LT_NT = 256;
constexpr int LT_RS = LT_W + 2, LT_PL = (LT_H + 2) * LT_RS;     // haloed planar tile: 3 planes of 10 x 34 floats

// Workgroup barrier WITHOUT the fence of __syncthreads(): hipcc drains vmcnt in front of a fenced barrier, i.e. a wave would wait for
// its output stores to complete once per tile.  LDS traffic of the caller must be complete (lgkmcnt(0)).
__device__ __forceinline__ void hr_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Data gradient: dX[p][ci] = mask(aux[p][ci]) * sum_{tap,c} dSR[c][p + tap - 1] * w[c][ci][flip(tap)].
// K = 9 taps x 3 channels = 27 <= 32: ONE v_mfma_f32_16x16x32_bf16 per (16 ci) x (16 pixels) block, with the B
// operand gathered from the fp32 dSR tile in LDS (k = 3 tap + c) and the A operand (flipped weights) built once per
// workgroup from the fp32 OIHW tensor.  Output / mask addressing is the blocked pixel-major layout of common.h.
// grid: persistent over 8x32-pixel tiles; block 256 = 4 waves, wave w covers tile rows 2w, 2w+1.
// The 64 input channels are dealt to the MFMA rows in paired-block order (pm_acc_chan, common.h), so a lane holds whole 16-byte
// pieces of the blocked layout: 8 sixteen-byte stores per wave and tile.
// sign_bits (optional): the activation's signs as conv3x3_c64_persist_kernel left them, [tile][wave][lane] x 8 bytes (piece
// (k, nb), packed word jj: bits 4 nb + jj and 16 + 4 nb + jj of word k): this kernel uses the SAME tile decomposition, channel
// order and lane -> pixel permutation, so a lane reads its own 64 bits (4 MB per 540p-sized tile set) instead of 8 x 16 bytes
// of the bf16 activation (the whole 1.06 GB C0 at HR).
// MODE: where the activation-gradient mask comes from -- 0 none, 1 the bf16 activation `aux`, 2 its sign bits (compile-time: with run-time
// branches on the two pointers hipcc put a vmcnt(0) at the top of every tile, i.e. waited for the previous tile's output stores)
// PACKED: the same kernel as the generic planar (1 or 3 channels, fp32) -> 64-channel 3x3 convolution of the engines (discriminator conv_0
// and conv_9's data gradient, the pre-clean stack's stem and its out conv's data gradient): weights in pack_weights_kernel's layout
// [tap][64 rows][16] (the packer already flipped / transposed them for a data gradient), + bias, + LeakyReLU(slope), `pc` input planes.
template <int MODE, bool PACKED>
__global__ __launch_bounds__(LT_NT) void last2_dgrad_kernel(const float* __restrict__ dsr, long long dsr_nstride,
                                                            const float* __restrict__ w, const bf16_t* __restrict__ aux,
                                                            const uint2* __restrict__ sign_bits, float neg,
                                                            bf16_t* __restrict__ dst, int N, int H, int W, int mask_mode,
                                                            const bf16_t* __restrict__ wpack, const float* __restrict__ bias, float act_slope, int pc) {
    __shared__ float tile[2][3 * LT_PL];
    const int tid = threadIdx.x, lane = tid & 63, w4 = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    // pixel of a 16-pixel block that lane column l15 works on: the persistent conv kernel's permutation (conv3x3_persist.hip)
    const int i15 = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);
    const int ntx = cdiv(W, LT_W), nty = cdiv(H, LT_H);
    const int total = N * ntx * nty;
    const long long plane = (long long)H * W;

    // A[m = ci 16 mb + i][k = 8 q + j] = w[c][ci][2 - ky][2 - kx],  k = 3 (3 ky + kx) + c  (zero for k >= 27)
    bf16x8_t fa[4];
    int boff[8];                                             // B gather: float offset of k = 8 q + j inside the haloed tile
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const int tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
        boff[j] = k < 27 ? c * LT_PL + ky * LT_RS + kx : -1;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ci = pm_acc_chan(mb, l15);             // A rows are indexed by the lane column itself; paired-block channel order
            if (PACKED) fa[mb][j] = k < 27 ? wpack[(tap * 64 + ci) * 16 + c] : (bf16_t)0.f;
            else fa[mb][j] = (bf16_t)(k < 27 ? w[((long long)c * 64 + ci) * 9 + (2 - ky) * 3 + (2 - kx)] : 0.f);
        }
    }
    (void)mask_mode;
    // PACKED epilogue: bias of this lane's 16 output channels (piece k: channels 8 (4 k + q) + j) and the LeakyReLU slope (1 = none)
    float bsv[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsv[k][j] = (PACKED && bias) ? bias[8 * (4 * k + q) + j] : 0.f;

    // haloed dSR tile -> LDS (zeros outside the image) by LDS-DMA, 4 bytes per lane (the 34-float rows of a plane start 3 floats before
    // a 16-byte boundary), one whole tile ahead into the other buffer: 16 instructions of 64 floats per tile, 4 per wave.  r03: the
    // register-staged form waited, at its LDS write behind the tile's 8 output stores, for those stores (vmcnt counts in order), the B
    // gather had a branch and an lgkmcnt(0) per element (32 exposed LDS round trips per tile), and the epilogue branched per pixel block.
    int srel[4];                                             // source byte offset of this lane's float of piece w4 + 4 i from the tile origin
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (w4 + 4 * i) * 64 + lane;
        const int c = e / LT_PL, rem = e - c * LT_PL, yy = rem / LT_RS, xx = rem - yy * LT_RS;
        srel[i] = (int)(((long long)c * plane + (long long)(yy - 1) * W + (xx - 1)) * 4);
    }
    const char* zsrc = reinterpret_cast<const char*>(g_lt_zero);
    auto stage_dma = [&](int t, int b) {
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        const char* org = reinterpret_cast<const char*>(dsr + (long long)n * dsr_nstride + (long long)ty0 * W + tx0);
        // straight-line: exactly 4 vector-memory instructions per call, so that hipcc can count them (with a branch per border tile in
        // here it waited with vmcnt(0) for the sign-bit load in front of them, i.e. for this DMA itself)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (w4 + 4 * i) * 64 + lane;
            const int c = e / LT_PL, rem = e - c * LT_PL, yy = rem / LT_RS, xx = rem - yy * LT_RS;
            const int vy = ty0 + yy - 1, vx = tx0 + xx - 1;
            const char* sp = (vy >= 0 && vy < H && vx >= 0 && vx < W && (!PACKED || c < pc)) ? org + srel[i] : zsrc;
            if (e < 3 * LT_PL)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                                 (__attribute__((address_space(3))) void*)(&tile[b][(w4 + 4 * i) * 64]), 4, 0, 0);
        }
    };
    // B gather addresses: element j of this lane's fragment is float boff[j] of the tile (k = 8 q + j >= 27: any float, its A column is zero)
    unsigned gaddr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gaddr[j] = (unsigned)(((boff[j] >= 0 ? boff[j] : 0) + (2 * w4) * LT_RS + i15) * 4);
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(&tile[0][0]);     // LDS byte offset of the tile buffers

    int t = blockIdx.x, buf = 0;
    if (t < total) stage_dma(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; t < total; t += gridDim.x) {
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        const long long obase = (long long)n * pm_image_elems(H, W, 64) + pm_off(ty0, tx0, 0, W, 64);
        const bool full = ty0 + LT_H <= H && tx0 + LT_W <= W;
        // mask operands first (older than the DMA below: waiting for them does not wait for the DMA)
        uint4 mm[2][4];
        bool ok[4];
        int loff[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int row = 2 * w4 + (nb >> 1), px = (nb & 1) * 16 + i15;
            ok[nb] = ty0 + row < H && tx0 + px < W;
            loff[nb] = (row * pm_ws(W) * 8 + q) * 256 + px * 8;
            if (MODE == 1 && ok[nb]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) mm[k][nb] = *reinterpret_cast<const uint4*>(aux + obase + loff[nb] + k * 1024);
            }
        }
        // the 64 sign bits of this lane's outputs: requested by inline asm and waited for by hand below -- four DMA instructions follow it
        // (behind exec-mask branches hipcc cannot count), and its own wait in front of the first use was a vmcnt(0), i.e. for that DMA
        typedef __attribute__((ext_vector_type(2))) unsigned sb2_t;
        sb2_t sbv = {0u, 0u};
        if (MODE == 2) {
            const auto* sp = (const __attribute__((address_space(1))) sb2_t*)(sign_bits + ((long long)t * 256 + w4 * 64 + lane));
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sbv) : "v"(sp) : "memory");
        }
        const int tn = t + gridDim.x;
        stage_dma(tn < total ? tn : t, buf ^ 1);             // lands during this tile's work (the walk's last tile re-stages itself: no branch)
        // B[k][n = pixel i] for this wave's 4 pixel blocks (32 ds_read_b32 in flight, inline asm: hipcc would drain the DMA in front of a
        // plain LDS read), then 16 MFMAs
        f32x4_t acc[4][4];
        {
            const unsigned tb = tile_lds + (unsigned)(buf * 3 * LT_PL * 4);
            float gv[4][8];
#define LD_G(nb_, j_) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(gv[nb_][j_]) : "v"(tb + gaddr[j_]), "n"((((nb_) >> 1) * LT_RS + ((nb_) & 1) * 16) * 4))
#define LD_G8(nb_) LD_G(nb_, 0); LD_G(nb_, 1); LD_G(nb_, 2); LD_G(nb_, 3); LD_G(nb_, 4); LD_G(nb_, 5); LD_G(nb_, 6); LD_G(nb_, 7);
            LD_G8(0) LD_G8(1) LD_G8(2) LD_G8(3)
#undef LD_G8
#undef LD_G
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                asm volatile("" : "+v"(gv[nb][0]), "+v"(gv[nb][1]), "+v"(gv[nb][2]), "+v"(gv[nb][3]), "+v"(gv[nb][4]), "+v"(gv[nb][5]), "+v"(gv[nb][6]), "+v"(gv[nb][7]));
                bf16x8_t fb;
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = (bf16_t)gv[nb][j];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mb], fb, z, 0, 0, 0);
                }
            }
        }
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); asm volatile("" : "+v"(sbv)); }     // the sign bits are here; the 4 DMA pieces may be in flight
        const uint2 sb = make_uint2(sbv.x, sbv.y);
        auto epilogue = [&](auto FULL) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (!(decltype(FULL)::value || ok[nb])) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = acc[2 * k][nb][j]; v[4 + j] = acc[2 * k + 1][nb][j]; }
                if (PACKED) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = v[j] + bsv[k][j]; v[j] = fmaxf(t, t * act_slope); }       // slope 1: no activation
                }
                if (MODE == 2) {
                    const unsigned wbits = k ? sb.y : sb.x;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= ((wbits >> (4 * nb + (j >> 1) + 16 * (j & 1))) & 1u) ? 1.f : neg;
                } else if (MODE == 1) {
                    const unsigned mw[4] = {mm[k][nb].x, mm[k][nb].y, mm[k][nb].z, mm[k][nb].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float m = __uint_as_float((j & 1) ? (mw[j >> 1] & 0xffff0000u) : (mw[j >> 1] << 16));
                        v[j] *= (m > 0.f ? 1.f : neg);
                    }
                }
                bf16x8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
                *reinterpret_cast<bf16x8_t*>(dst + obase + loff[nb] + k * 1024) = o;
            }
        }
        };
        if (full) {
            epilogue(std::true_type{});
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the 8 stores of this tile may stay in flight: the DMA in front of them has landed
        } else {
            epilogue(std::false_type{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        hr_barrier();                                        // the next tile is staged; this one is consumed
        buf ^= 1;
    }
}

// PyTorch upsample_bilinear2d(align_corners=False) source index for scale `inv` = 1/4 (or 1/2: upscale = 2) (basicvsr.py:22)
__device__ __forceinline__ void hr_bil4(int d, int in_size, int& i0, int& i1, float& l1, float inv = 0.25f) {
    float s = (d + 0.5f) * inv - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// conv3x3 64 -> (<= 4) channels with a planar fp32 destination: conv_last.2 + bilinear x4 skip (basicvsr.py:21-22,82),
// the pre-clean stack's `x + conv` (realbasicvsr.py:28-29), the stems' LR-channel data gradient.  M = 16 rows (<= 4
// real) x N = 16 pixels x K = 32 channels per v_mfma_f32_16x16x32_bf16: all 18 (tap, channel-half) weight fragments live
// in registers; the haloed 10 x 34 pixel tile is staged in LDS in the image the persistent kernel uses
// ([row][8 chunks][34 px][16 B]: a B fragment is one ds_read_b128, 16 lanes = 16 consecutive slots).  The kernel streams:
// 43.5 KB in, <= 4 KB out per tile; three workgroups per CU cover each other's staging latency.
constexpr int FP_CH = LT_RS * 16, FP_ROW = 8 * FP_CH, FP_TILE = (LT_H + 2) * FP_ROW;      // 544 / 4352 / 43,520 bytes
constexpr int FP_CHUNKS = FP_TILE / 16;                                                   // 2,720 16-byte slots
constexpr int FP_NPIECE = (FP_CHUNKS + 63) / 64, FP_NPIECE_W = (FP_NPIECE + 3) / 4;      // 43 DMA pieces, 11 per wave
constexpr int FP_NT = 512;
__device__ uint4 g_fp_zero_chunk[2];

#define FP_GLDS16(src, dst)                                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// Same structure as conv3x3_c64_persist_kernel with ONE 16-row cout block: a persistent 512-thread workgroup per CU,
// 4 producer waves filling the other tile buffer by LDS-DMA a tile ahead, 4 MFMA waves (tile rows 2w, 2w+1).
__global__ __launch_bounds__(FP_NT, 1) void c64_to_planar_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char tiles[];      // 2 x FP_TILE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
    const int i15 = lane & 15, q = lane >> 4;
    // pixel of a 16-pixel block that lane column i15 works on: the persistent conv kernel's permutation (even pixels on lanes 4-11, odd ones
    // on 0-3 / 12-15), which makes every lane group of a ds_read_b128 hit 16 distinct bank groups (r03 PMC: 44 % of this kernel's LDS
    // cycles were bank conflicts with pixel = lane; 437 -> 424 us per 2160x3840 frame.  A ring of three tile buffers on top of it measured
    // 427 us: with the conflicts gone the consumers, not the DMA, set the tile time)
    const int px15 = (i15 >= 4 && i15 < 12) ? 2 * (i15 - 4) : (i15 < 4 ? 2 * i15 + 1 : 2 * (i15 - 8) + 1);
    const int ntx = cdiv(a.W, LT_W), nty = cdiv(a.H, LT_H);
    const int total = a.N * ntx * nty;
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);

    if (role == 1) {
        const char* src = reinterpret_cast<const char*>(a.src[0]);
        const char* zsrc = reinterpret_cast<const char*>(g_fp_zero_chunk);
        const int WSs = pm_ws(a.Ws);
        int rel[FP_NPIECE_W];
#pragma unroll
        for (int i = 0; i < FP_NPIECE_W; ++i) {
            const int idx = (w4 + 4 * i) * 64 + lane;           // LDS slot = [row ty][chunk c][34 pixels tx] x 16 B
            const int ty = idx / (8 * LT_RS), rem = idx - ty * (8 * LT_RS);
            const int c = rem / LT_RS, dx = rem - c * LT_RS - 1;
            rel[i] = ((((ty - 1) * WSs + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
        }
        auto issue = [&](int tile, int buf) {
            const int n = tile / (ntx * nty), r = tile - n * (ntx * nty);
            const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
            const char* org = src + ((long long)n * a.src_nstride[0] + pm_off(ty0, tx0, 0, a.Ws, 64)) * 2;
            char* dstb = tiles + buf * FP_TILE;
            const bool interior = ty0 >= 1 && ty0 + LT_H < a.H && tx0 >= 1 && tx0 + LT_W < a.W;
#pragma unroll
            for (int i = 0; i < FP_NPIECE_W; ++i) {
                const int piece = w4 + 4 * i;
                const int idx = piece * 64 + lane;
                const char* s = org + rel[i];
                if (!interior) {
                    const int ty = idx / (8 * LT_RS), tx = (idx - ty * (8 * LT_RS)) % LT_RS;
                    const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                    if (!(vy >= 0 && vy < a.H && vx >= 0 && vx < a.W)) s = zsrc;
                }
                if (piece < FP_NPIECE && idx < FP_CHUNKS) FP_GLDS16(s, dstb + piece * 1024);
            }
        };
        int cur = 0, tile = walk.first;
        if (tile < walk.end) issue(tile, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (; tile < walk.end; tile += walk.stride) {
            const int next = tile + walk.stride;
            if (next < walk.end) issue(next, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        return;
    }

    // A[m = cout i][k = 8 q + j] of every (tap, channel half) from the packed weights [9][32 rows][64] bf16: in registers
    bf16x8_t fa[9][2];
    {
        const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.wpack);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fa[tap][kk] = *reinterpret_cast<const bf16x8_t*>(wp + ((tap * 32 + i15) * 64 + kk * 32 + 8 * q));
    }
    const long long plane = (long long)a.Hd * a.Wd;
    // The BasicVSR case (conv_last.2 + bilinear x4 skip, 3 channels, an LR frame of exactly H/4 x W/4) without per-tile divisions,
    // float index arithmetic or per-channel branches (r03: the tile time of this kernel is set by its consumers).  For d = 4 m + r,
    // (d + 0.5) / 4 - 0.5 = m + (r - 1.5) / 4: source index m - 1 (r < 2) or m, weight 0.625 / 0.875 / 0.125 / 0.375 -- exactly what
    // hr_bil4 computes in fp32; tile origins are multiples of 8 and 32, so r and the weight are per-lane constants.
    const int oyr = 2 * w4 + (q >> 1), oxr = (q & 1) * 16 + px15;
    const int yk = (oyr >> 2) - ((oyr & 3) < 2 ? 1 : 0), xk = (oxr >> 2) - ((oxr & 3) < 2 ? 1 : 0);
    const float wq[4] = {0.625f, 0.875f, 0.125f, 0.375f};
    const float lyc = wq[oyr & 3], lxc = wq[oxr & 3];
    const bool fast = a.base_lr && !a.pres && a.cout_real == 3 && a.base_scale != 2 && a.base_h * 4 == a.H && a.base_w * 4 == a.W;
    const float binv = a.base_scale == 2 ? 0.5f : 0.25f;
    float bs[3] = {0.f, 0.f, 0.f};
    if (fast && a.bias) { bs[0] = a.bias[0]; bs[1] = a.bias[1]; bs[2] = a.bias[2]; }
    const long long lr_plane = (long long)a.base_h * a.base_w;
    int tn = walk.first / (ntx * nty), tyi, txi;               // image, tile row, tile column of the walk, advanced by adds and carries
    { const int r0 = walk.first - tn * (ntx * nty); tyi = r0 / ntx; txi = r0 - tyi * ntx; }
    const int sn = walk.stride / (ntx * nty), sty = (walk.stride - sn * (ntx * nty)) / ntx, stx = walk.stride - sn * (ntx * nty) - sty * ntx;
    // fast path: the 12 LR values of a tile are requested one tile AHEAD (at the end of the previous iteration, behind its stores) and
    // combined after the tile's K loop: with the K loop down to ~1.2 k cycles their L2 round trip was the longest thing in a tile
    float lvn[3][4], lyn = 0.f, lxn = 0.f;
    bool okn = false;
    auto prefetch = [&](int n, int ty0, int tx0) {
        okn = ty0 + oyr < a.H && tx0 + oxr < a.W;
        int y0 = (ty0 >> 2) + yk, x0 = (tx0 >> 2) + xk;
        lyn = lyc; lxn = lxc;
        if (y0 < 0) { y0 = 0; lyn = 0.f; }
        if (x0 < 0) { x0 = 0; lxn = 0.f; }
        if (!okn) { y0 = 0; x0 = 0; }                        // lanes of an outside pixel load the frame's first pixel and store nothing
        const int y1 = y0 + (y0 < a.base_h - 1 ? 1 : 0), x1 = x0 + (x0 < a.base_w - 1 ? 1 : 0);
        const float* bp = a.base_lr + (long long)n * a.base_nstride;
        const int o00 = y0 * a.base_w + x0, o01 = y0 * a.base_w + x1, o10 = y1 * a.base_w + x0, o11 = y1 * a.base_w + x1;
#pragma unroll
        for (int c = 0; c < 3; ++c) { lvn[c][0] = bp[c * lr_plane + o00]; lvn[c][1] = bp[c * lr_plane + o01]; lvn[c][2] = bp[c * lr_plane + o10]; lvn[c][3] = bp[c * lr_plane + o11]; }
    };
    if (fast && walk.first < walk.end) prefetch(tn, tyi * LT_H, txi * LT_W);
    __syncthreads();                                         // the first tile is in LDS
    int cur = 0;
    for (int t = walk.first; t < walk.end; t += walk.stride) {
        const int n = tn, ty0 = tyi * LT_H, tx0 = txi * LT_W;
        txi += stx; if (txi >= ntx) { txi -= ntx; ++tyi; }
        tyi += sty; if (tyi >= nty) { tyi -= nty; ++tn; }
        tn += sn;
        // epilogue operands first (their latency hides behind the MFMAs).  The accumulator rows that matter (m = 0..3)
        // sit in lanes 0-15 only; after the K loop they are broadcast so that lane group q finishes pixel block nb = q:
        // all 64 lanes load / store, one pixel each.
        const int oy = ty0 + oyr, ox = tx0 + oxr;
        const bool ok = oy < a.H && ox < a.W;
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        float lv[3][4], ly = 0.f, lx = 0.f;
        if (fast) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) lv[c][k] = lvn[c][k];
            ly = lyn; lx = lxn;
        } else if (ok) {
            int y0 = 0, y1 = 0, x0 = 0, x1 = 0; float ly = 0.f, lx = 0.f;
            if (a.base_lr) { hr_bil4(oy, a.base_h, y0, y1, ly, binv); hr_bil4(ox, a.base_w, x0, x1, lx, binv); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= a.cout_real) break;
                float v = a.bias ? a.bias[c] : 0.f;
                if (a.pres) v += a.pres[(long long)n * a.dst_nstride + c * plane + (long long)oy * a.Wd + ox];
                if (a.base_lr) {
                    const float* bp = a.base_lr + (long long)n * a.base_nstride + (long long)c * a.base_h * a.base_w;
                    const float v00 = bp[y0 * a.base_w + x0], v01 = bp[y0 * a.base_w + x1];
                    const float v10 = bp[y1 * a.base_w + x0], v11 = bp[y1 * a.base_w + x1];
                    v += (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                }
                add[c] = v;
            }
        }
        f32x4_t acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // K loop in 8 groups g = (tile row R = g >> 1 of the wave's 4 haloed rows, channel half kk = g & 1): the group's 6 B fragments
        // (pixels kx, 16 + kx) serve the output rows rowsel = 0, 1 with ky = R - rowsel: 6 or 12 MFMAs.  The fragments of group g + 1 are
        // requested (inline asm: hipcc puts an lgkmcnt(0) in front of almost every MFMA of a plain loop -- 22 exposed LDS round trips
        // per tile, r03) behind the first MFMAs of group g into the other register set.
        const unsigned tbu = (unsigned)(cur * FP_TILE + (2 * w4) * FP_ROW + q * FP_CH + px15 * 16);   // the dynamic LDS segment starts at 0
        bf16x8_t fbq[2][6];
#define CP_DSR(dst_, imm_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(tbu), "n"(imm_))
#define CP_LOAD1(g_, i_) if constexpr ((g_) < 8) { CP_DSR(fbq[(g_) & 1][i_], ((g_) >> 1) * FP_ROW + ((g_) & 1) * 4 * FP_CH + (((i_) / 3) * 16 + (i_) % 3) * 16); }
#define CP_SB __builtin_amdgcn_sched_barrier(0);
        // MFMA m (0..11) of group g: rowsel = m / 6, half = (m / 3) & 1, kx = m % 3
#define CP_MFMA(g_, m_)                                                                                                \
        { constexpr int R_ = (g_) >> 1, kk_ = (g_) & 1, rs_ = (m_) / 6, hf_ = ((m_) / 3) & 1, kx_ = (m_) % 3, ky_ = R_ - rs_;  \
          if constexpr (ky_ >= 0 && ky_ <= 2)                                                                           \
              acc[rs_ * 2 + hf_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(ky_ < 0 ? 0 : ky_ > 2 ? 2 : ky_) * 3 + kx_][kk_], fbq[(g_) & 1][hf_ * 3 + kx_], acc[rs_ * 2 + hf_], 0, 0, 0); }
#define CP_GROUP(g_)                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); CP_SB                                                      \
        CP_MFMA(g_, 0) CP_MFMA(g_, 6) CP_SB CP_LOAD1((g_) + 1, 0) CP_SB CP_MFMA(g_, 1) CP_MFMA(g_, 7) CP_SB CP_LOAD1((g_) + 1, 1) CP_SB  \
        CP_MFMA(g_, 2) CP_MFMA(g_, 8) CP_SB CP_LOAD1((g_) + 1, 2) CP_SB CP_MFMA(g_, 3) CP_MFMA(g_, 9) CP_SB CP_LOAD1((g_) + 1, 3) CP_SB  \
        CP_MFMA(g_, 4) CP_MFMA(g_, 10) CP_SB CP_LOAD1((g_) + 1, 4) CP_SB CP_MFMA(g_, 5) CP_MFMA(g_, 11) CP_SB CP_LOAD1((g_) + 1, 5) CP_SB
        CP_SB CP_LOAD1(0, 0) CP_LOAD1(0, 1) CP_LOAD1(0, 2) CP_LOAD1(0, 3) CP_LOAD1(0, 4) CP_LOAD1(0, 5)
        CP_GROUP(0) CP_GROUP(1) CP_GROUP(2) CP_GROUP(3) CP_GROUP(4) CP_GROUP(5) CP_GROUP(6) CP_GROUP(7)
#undef CP_GROUP
#undef CP_MFMA
#undef CP_SB
#undef CP_LOAD1
#undef CP_DSR
        if (fast) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                add[c] = bs[c] + ((1.f - ly) * ((1.f - lx) * lv[c][0] + lx * lv[c][1]) + ly * ((1.f - lx) * lv[c][2] + lx * lv[c][3]));
        }
        float* dst = reinterpret_cast<float*>(a.dst[0]) + (long long)n * a.dst_nstride + (long long)oy * a.Wd + ox;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= a.cout_real) break;
            const float s0 = __shfl(acc[0][c], i15, 64), s1 = __shfl(acc[1][c], i15, 64);
            const float s2 = __shfl(acc[2][c], i15, 64), s3 = __shfl(acc[3][c], i15, 64);
            const float v = (q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : s3) + add[c];
            if (ok) dst[c * plane] = v;
        }
        if (fast && t + walk.stride < walk.end) prefetch(tn, tyi * LT_H, txi * LT_W);    // (tn, tyi, txi) already name the next tile
        __syncthreads();                                     // the producers' next tile has landed; this one is consumed
        cur ^= 1;
    }
}

// Weight gradient of a 64 -> 3 conv3x3 whose cotangent is planar fp32 (conv_last.2: dY = d SR; the pre-clean stack's out conv):
//   dW[co][ci][tap] = sum_p dY[co][p] * X[p + tap - 1][ci],   db[co] = sum_p dY[co][p]
// As v_mfma_f32_16x16x32_bf16 with K = the 32 pixels of a tile row: A[m = ci][k = pixel] by transposing LDS reads
// (ds_read_b64_tr_b16) of the haloed X tile, B[k = pixel][n = co] straight from global (8 consecutive floats of plane co;
// only n < 3 is real).  Accumulators: 9 taps x 4 ci blocks.  The kernel streams X once (45 KB per tile with halo):
// persistent 512-thread workgroup per CU, 4 producer waves (LDS-DMA, a tile ahead) + 4 MFMA waves (tile rows 2w, 2w+1),
// one partial slab per workgroup in wgrad_reduce_kernel's layout [tap][32][64] + [32].
constexpr int LW_XS = 36;                                              // slots per chunk row: 576-byte stride, conflict-free tr reads
constexpr int LW_ROW = 8 * LW_XS * 16, LW_TILE = (LT_H + 2) * LW_ROW;  // 4,608 / 46,080 bytes
constexpr int LW_SLOTS = LW_TILE / 16, LW_NPIECE = LW_SLOTS / 64;      // 2,880 slots = 45 DMA pieces
constexpr int LW_NPIECE_W = (LW_NPIECE + 3) / 4;                       // 12 per producer wave
constexpr int LW_DY = 3 * LT_H * LT_W * 4;                             // 3,072 bytes: the dY tile of the ring kernel
__device__ __forceinline__ s16x4_t hr_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

__global__ __launch_bounds__(FP_NT, 1) void last2_wgrad_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dy,
                                                               long long dy_nstride, float* __restrict__ slab, int slab_stride,
                                                               int N, int H, int W, int pc) {
    extern __shared__ __attribute__((aligned(16))) char tiles[];      // 2 x LW_TILE; reused for the final reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
    const int i15 = lane & 15, q = lane >> 4;
    const int ntx = cdiv(W, LT_W), nty = cdiv(H, LT_H);
    const int total = N * ntx * nty;
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);
    const long long plane = (long long)H * W;

    if (role == 1) {
        const char* src = reinterpret_cast<const char*>(x);
        const char* zsrc = reinterpret_cast<const char*>(g_fp_zero_chunk);
        const int WSs = pm_ws(W);
        auto issue = [&](int tile, int buf) {
            const int n = tile / (ntx * nty), r = tile - n * (ntx * nty);
            const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
            const char* org = src + ((long long)n * pm_image_elems(H, W, 64) + pm_off(ty0, tx0, 0, W, 64)) * 2;
            char* dstb = tiles + buf * LW_TILE;
#pragma unroll
            for (int i = 0; i < LW_NPIECE_W; ++i) {
                const int piece = w4 + 4 * i;
                if (piece >= LW_NPIECE) break;
                const int idx = piece * 64 + lane;           // LDS slot = [row ty][chunk c][36 slots tx] x 16 B
                const int ty = idx / (8 * LW_XS), rem = idx - ty * (8 * LW_XS);
                const int c = rem / LW_XS, tx = rem - c * LW_XS;
                const int dx = tx - 1;
                const int vy = ty0 + ty - 1, vx = tx0 + dx;
                const char* sp = org + ((((ty - 1) * WSs + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
                if (!(tx < LT_W + 2 && vy >= 0 && vy < H && vx >= 0 && vx < W)) sp = zsrc;
                FP_GLDS16(sp, dstb + piece * 1024);
            }
        };
        int cur = 0, tile = walk.first;
        if (tile < walk.end) issue(tile, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (; tile < walk.end; tile += walk.stride) {
            const int next = tile + walk.stride;
            if (next < walk.end) issue(next, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        __syncthreads();                                     // matches the consumers' barrier before the final reduction
        __syncthreads();
        return;
    }

    f32x4_t acc[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[tap][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // transposing-read address of this lane: lane 4 q4 + p of a 16-lane group supplies pixel q4, channels 4p..4p+3;
    // the group index (= q of the MFMA operand) selects the pixel octet 8 q .. 8 q + 7 of the row
    const int q4 = (lane & 15) >> 2, p4 = (lane & 3) * 4;
    const int abase = (p4 >> 3) * (LW_XS * 16) + (8 * q + q4) * 16 + (p4 & 7) * 2;     // + mb * 2 chunks, + row, + kx
    __syncthreads();                                         // the first tile is in LDS
    int cur = 0;
    for (int t = walk.first; t < walk.end; t += walk.stride) {
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        // B[k = pixel 8 q + j][n = co i15] of this wave's two rows, straight from global (zeros outside / for co >= 3)
        bf16x8_t fb[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oy = ty0 + 2 * w4 + rr, ox = tx0 + 8 * q;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (i15 < pc && oy < H) {
                const float* dp = dy + (long long)n * dy_nstride + i15 * plane + (long long)oy * W + ox;
                if (ox + 8 <= W && ((reinterpret_cast<size_t>(dp) & 15) == 0)) {
                    const float4 a0 = *reinterpret_cast<const float4*>(dp), a1 = *reinterpret_cast<const float4*>(dp + 4);
                    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (ox + j < W) v[j] = dp[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { fb[rr][j] = (bf16_t)v[j]; bsum += v[j]; }
        }
        const char* tb = tiles + cur * LW_TILE + abase;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const char* pa = tb + (2 * w4 + rr + tap / 3) * LW_ROW + mb * 2 * (LW_XS * 16) + (tap % 3) * 16;
                    union { s16x4_t s[2]; bf16x8_t b; } u;
                    u.s[0] = hr_tr_read(pa);
                    u.s[1] = hr_tr_read(pa + 64);
                    acc[tap][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.b, fb[rr], acc[tap][mb], 0, 0, 0);
                }
        __syncthreads();                                     // the producers' next tile has landed; this one is consumed
        cur ^= 1;
    }

    // ---- sum the 4 MFMA waves through LDS (the tile buffers are free), then one slab per workgroup ----
    __syncthreads();
    float* xch = reinterpret_cast<float*>(tiles);            // [3 waves][144][64 lanes] fp32 = 110,592 B (of 92,160 x ... see launcher)
    if (w4 > 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) xch[(((w4 - 1) * 36 + tap * 4 + mb) * 4 + j) * 64 + lane] = acc[tap][mb][j];
    }
    float* bx = xch + 3 * 144 * 64;                          // [4 waves][64 lanes] bias partial sums
    bx[w4 * 64 + lane] = bsum;
    __syncthreads();
    if (w4 == 0) {
        float* sl = slab + (long long)blockIdx.x * slab_stride;
        const int co = i15;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[tap][mb][j];
#pragma unroll
                    for (int ww = 0; ww < 3; ++ww) v += xch[((ww * 36 + tap * 4 + mb) * 4 + j) * 64 + lane];
                    if (co < pc) sl[((long long)tap * 32 + co) * 64 + mb * 16 + 4 * q + j] = v;     // C[m = ci 4q+j][n = co]
                }
        if (lane < pc) {                                     // db[co]: lanes (co, q = 0..3) of the 4 waves
            float b = 0.f;
            for (int ww = 0; ww < 4; ++ww)
                for (int qq = 0; qq < 4; ++qq) b += bx[ww * 64 + qq * 16 + lane];
            sl[9 * 32 * 64 + lane] = b;
        }
    }
}

__global__ __launch_bounds__(FP_NT, 1) void last2_wgrad_ring_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dy,
                                                               long long dy_nstride, float* __restrict__ slab, int slab_stride,
                                                               int N, int H, int W, int pc) {
    extern __shared__ __attribute__((aligned(16))) char tiles[];      // ring: 3 x LW_TILE + 3 x LW_DY; reused for the final reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
    const int i15 = lane & 15, q = lane >> 4;
    const int ntx = cdiv(W, LT_W), nty = cdiv(H, LT_H);
    const int total = N * ntx * nty;
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);
    const long long plane = (long long)H * W;

    if (role == 1) {
        const char* src = reinterpret_cast<const char*>(x);
        const char* zsrc = reinterpret_cast<const char*>(g_fp_zero_chunk);
        const int WSs = pm_ws(W);
        // rel[i]: source byte offset of this lane's slot of piece w4 + 4 i from the tile origin (per-lane constants; r03: computed per
        // piece and tile before, hipcc hoisted the invariant parts and spilled them -- two 16-byte scratch round trips per tile in the
        // producers' loop); (image, tile row, tile column) advance by adds and carries
        int rel[LW_NPIECE_W];
#pragma unroll
        for (int i = 0; i < LW_NPIECE_W; ++i) {
            const int idx = (w4 + 4 * i) * 64 + lane;        // LDS slot = [row ty][chunk c][36 slots tx] x 16 B
            const int ty = idx / (8 * LW_XS), rem = idx - ty * (8 * LW_XS);
            const int c = rem / LW_XS, dx = rem - c * LW_XS - 1;
            rel[i] = ((((ty - 1) * WSs + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
        }
        const long long img_bytes = pm_image_elems(H, W, 64) * 2;
        auto issue = [&](int n, int tyi, int txi, int buf) {
            const int ty0 = tyi * LT_H, tx0 = txi * LT_W;
            const char* org = src + (long long)n * img_bytes + pm_off(ty0, tx0, 0, W, 64) * 2;
            char* dstb = tiles + buf * LW_TILE;
            const bool interior = ty0 >= 1 && ty0 + LT_H < H && tx0 >= 1 && tx0 + LT_W < W;      // wave-uniform
#pragma unroll
            for (int i = 0; i < LW_NPIECE_W; ++i) {
                const int piece = w4 + 4 * i;
                if (piece >= LW_NPIECE) break;
                const int idx = piece * 64 + lane;
                const int ty = idx / (8 * LW_XS), tx = (idx - ty * (8 * LW_XS)) % LW_XS;
                const char* sp = org + rel[i];
                if (tx >= LT_W + 2) sp = zsrc;               // pad slots
                else if (!interior) {
                    const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                    if (!(vy >= 0 && vy < H && vx >= 0 && vx < W)) sp = zsrc;
                }
                FP_GLDS16(sp, dstb + piece * 1024);
            }
        };
        // dY tile [co (3)][8 rows][32 px] fp32 = 3 pieces of 1 KiB: piece c on producer wave c + 1 (those waves carry 11 X pieces, so every
        // wave issues 12 DMA instructions per tile).  W is a multiple of 4 here, so a 16-byte chunk is inside the image or outside it.
        auto issue_dy = [&](int n, int tyi, int txi, int buf) {
            if (w4 == 0) return;
            const int ty0 = tyi * LT_H, tx0 = txi * LT_W;
            const int row = lane >> 3, xc = (lane & 7) * 4;
            const float* sp = dy + (long long)n * dy_nstride + (long long)(w4 - 1) * plane + (long long)(ty0 + row) * W + tx0 + xc;
            const char* spc = (w4 - 1 < pc && ty0 + row < H && tx0 + xc < W) ? reinterpret_cast<const char*>(sp) : zsrc;     // planes >= pc do not exist: zeros
            FP_GLDS16(spc, tiles + 3 * LW_TILE + buf * LW_DY + (w4 - 1) * 1024);
        };
        struct { int T, n, ty, tx; } it;
        const int per = ntx * nty;
        { int r0 = walk.first; it.T = r0; it.n = r0 / per; r0 -= it.n * per; it.ty = r0 / ntx; it.tx = r0 - it.ty * ntx; }
        const int sn = walk.stride / per, sty = (walk.stride - sn * per) / ntx, stx = walk.stride - sn * per - sty * ntx;
        auto advance = [&]() {
            it.T += walk.stride;
            it.tx += stx; if (it.tx >= ntx) { it.tx -= ntx; ++it.ty; }
            it.ty += sty; if (it.ty >= nty) { it.ty -= nty; ++it.n; }
            it.n += sn;
        };
        // ring of three tile buffers, DMA two tiles ahead (the tile's work is far shorter than the loaded memory latency: two tiles in
        // flight instead of one); at the end of tile i only tile i+1 must have landed: the 12 pieces of tile i+2 stay in flight
        int cur = 0;                                         // `it` = the tile two ahead of the consumers once the loop runs
        if (it.T < walk.end) { issue(it.n, it.ty, it.tx, 0); issue_dy(it.n, it.ty, it.tx, 0); }
        advance();
        if (it.T < walk.end) { issue(it.n, it.ty, it.tx, 1); issue_dy(it.n, it.ty, it.tx, 1); }
        advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int tile = walk.first; tile < walk.end; tile += walk.stride) {
            if (it.T < walk.end) {
                const int b2 = cur == 0 ? 2 : cur - 1;
                issue(it.n, it.ty, it.tx, b2); issue_dy(it.n, it.ty, it.tx, b2);
                advance();
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
        __syncthreads();                                     // matches the consumers' barrier before the final reduction
        __syncthreads();
        return;
    }

    f32x4_t acc[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[tap][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // transposing-read address of this lane: lane 4 q4 + p of a 16-lane group supplies pixel q4, channels 4p..4p+3;
    // the group index (= q of the MFMA operand) selects the pixel octet 8 q .. 8 q + 7 of the row
    const int q4 = (lane & 15) >> 2, p4 = (lane & 3) * 4;
    const int abase = (p4 >> 3) * (LW_XS * 16) + (8 * q + q4) * 16 + (p4 & 7) * 2;     // + mb * 2 chunks, + row, + kx
    __syncthreads();                                         // the first tile is in LDS
    int cur = 0;
    for (int t = walk.first; t < walk.end; t += walk.stride) {
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        // B[k = pixel 8 q + j][n = co i15] of this wave's two rows, from the dY tile the producers staged (zeros for co >= 3)
        bf16x8_t fb[2];
        {
            const float* dt = reinterpret_cast<const float*>(tiles + 3 * LW_TILE + cur * LW_DY) + (i15 < pc ? i15 : 0) * 256 + 8 * q;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float4 a0 = *reinterpret_cast<const float4*>(dt + (2 * w4 + rr) * 32), a1 = *reinterpret_cast<const float4*>(dt + (2 * w4 + rr) * 32 + 4);
                const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float vv = i15 < pc ? v[j] : 0.f; fb[rr][j] = (bf16_t)vv; bsum += vv; }
            }
        }
        // K loop in 8 groups g = (X row R = g >> 1 of the wave's 4 haloed rows, ci-block pair mp = g & 1): the group's 6 A fragments (ci
        // block 2 mp + i / 3, tap column kx = i % 3; two transposing reads each) serve the output rows rr = 0, 1 with ky = R - rr: 6 or 12
        // MFMAs.  The reads of group g + 1 go out behind the MFMAs of group g into the other register set (inline asm; the plain loop
        // had an lgkmcnt(0) in front of 40 of a tile's 72 MFMAs and spilled two accumulator blocks, r03).
        const unsigned tbu = (unsigned)(cur * LW_TILE + abase + (2 * w4) * LW_ROW);      // the dynamic LDS segment starts at 0
        union { s16x4_t s[2]; bf16x8_t b; } fq[2][6];
#define LW_TRR(dst_, imm_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(tbu), "n"(imm_))
#define LW_LOAD1(g_, i_) if constexpr ((g_) < 8) {                                                                     \
            LW_TRR(fq[(g_) & 1][i_].s[0], ((g_) >> 1) * LW_ROW + (2 * ((g_) & 1) + (i_) / 3) * 2 * (LW_XS * 16) + ((i_) % 3) * 16);       \
            LW_TRR(fq[(g_) & 1][i_].s[1], ((g_) >> 1) * LW_ROW + (2 * ((g_) & 1) + (i_) / 3) * 2 * (LW_XS * 16) + ((i_) % 3) * 16 + 64); }
#define LW_SB __builtin_amdgcn_sched_barrier(0);
        // MFMA m (0..11) of group g: rr = m / 6, fragment i = m % 6
#define LW_MFMA(g_, m_)                                                                                                \
        { constexpr int R_ = (g_) >> 1, rr_ = (m_) / 6, i_ = (m_) % 6, ky_ = R_ - rr_, mb_ = 2 * ((g_) & 1) + i_ / 3, kyc_ = ky_ < 0 ? 0 : ky_ > 2 ? 2 : ky_;  \
          if constexpr (ky_ >= 0 && ky_ <= 2)                                                                           \
              acc[kyc_ * 3 + i_ % 3][mb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[(g_) & 1][i_].b, fb[rr_], acc[kyc_ * 3 + i_ % 3][mb_], 0, 0, 0); }
#define LW_GROUP(g_)                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LW_SB                                                      \
        LW_MFMA(g_, 0) LW_MFMA(g_, 6) LW_SB LW_LOAD1((g_) + 1, 0) LW_SB LW_MFMA(g_, 1) LW_MFMA(g_, 7) LW_SB LW_LOAD1((g_) + 1, 1) LW_SB  \
        LW_MFMA(g_, 2) LW_MFMA(g_, 8) LW_SB LW_LOAD1((g_) + 1, 2) LW_SB LW_MFMA(g_, 3) LW_MFMA(g_, 9) LW_SB LW_LOAD1((g_) + 1, 3) LW_SB  \
        LW_MFMA(g_, 4) LW_MFMA(g_, 10) LW_SB LW_LOAD1((g_) + 1, 4) LW_SB LW_MFMA(g_, 5) LW_MFMA(g_, 11) LW_SB LW_LOAD1((g_) + 1, 5) LW_SB
        LW_SB LW_LOAD1(0, 0) LW_LOAD1(0, 1) LW_LOAD1(0, 2) LW_LOAD1(0, 3) LW_LOAD1(0, 4) LW_LOAD1(0, 5)
        LW_GROUP(0) LW_GROUP(1) LW_GROUP(2) LW_GROUP(3) LW_GROUP(4) LW_GROUP(5) LW_GROUP(6) LW_GROUP(7)
#undef LW_GROUP
#undef LW_MFMA
#undef LW_SB
#undef LW_LOAD1
#undef LW_TRR
        hr_barrier();                                        // the producers' next tile has landed; this one is consumed
        cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- sum the 4 MFMA waves through LDS (the tile buffers are free), then one slab per workgroup ----
    __syncthreads();
    float* xch = reinterpret_cast<float*>(tiles);            // [3 waves][144][64 lanes] fp32 = 110,592 B (of 92,160 x ... see launcher)
    if (w4 > 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) xch[(((w4 - 1) * 36 + tap * 4 + mb) * 4 + j) * 64 + lane] = acc[tap][mb][j];
    }
    float* bx = xch + 3 * 144 * 64;                          // [4 waves][64 lanes] bias partial sums
    bx[w4 * 64 + lane] = bsum;
    __syncthreads();
    if (w4 == 0) {
        float* sl = slab + (long long)blockIdx.x * slab_stride;
        const int co = i15;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[tap][mb][j];
#pragma unroll
                    for (int ww = 0; ww < 3; ++ww) v += xch[((ww * 36 + tap * 4 + mb) * 4 + j) * 64 + lane];
                    if (co < pc) sl[((long long)tap * 32 + co) * 64 + mb * 16 + 4 * q + j] = v;     // C[m = ci 4q+j][n = co]
                }
        if (lane < pc) {                                     // db[co]: lanes (co, q = 0..3) of the 4 waves
            float b = 0.f;
            for (int ww = 0; ww < 4; ++ww)
                for (int qq = 0; qq < 4; ++qq) b += bx[ww * 64 + qq * 16 + lane];
            sl[9 * 32 * 64 + lane] = b;
        }
    }
}

}  // namespace

// dX (pixel-major bf16, 64 channels) = mask(aux) * dgrad of a 64 -> 3 3x3 conv, from the planar fp32 cotangent dsr
// (N images `dsr_nstride` floats apart) and the conv's fp32 OIHW weight (3,64,3,3).
int vsr_launch_last2_dgrad(const float* dsr, long long dsr_nstride, const float* w, const void* aux, void* dst, int N, int H, int W,
                           int mask_mode, hipStream_t st, const void* sign_bits, float slope) {
    if (!dsr || !w || !dst || N < 1 || H < 1 || W < 1) return VSR_ERR_BADARG;
    const int tiles = N * cdiv(W, LT_W) * cdiv(H, LT_H);
    const int grid = tiles < 256 * 8 ? tiles : 256 * 8;
    const float neg = mask_mode == MASK_LEAKY ? vsr_slope(slope) : 0.f;
    if ((long long)3 * H * W * 4 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;                     // in-tile source offsets are 32-bit
    if (sign_bits)
        hipLaunchKernelGGL((last2_dgrad_kernel<2, false>), dim3(grid), dim3(LT_NT), 0, st, dsr, dsr_nstride, w, (const bf16_t*)aux, (const uint2*)sign_bits, neg, (bf16_t*)dst, N, H, W, mask_mode, (const bf16_t*)nullptr, (const float*)nullptr, 1.f, 3);
    else if (aux)
        hipLaunchKernelGGL((last2_dgrad_kernel<1, false>), dim3(grid), dim3(LT_NT), 0, st, dsr, dsr_nstride, w, (const bf16_t*)aux, (const uint2*)sign_bits, neg, (bf16_t*)dst, N, H, W, mask_mode, (const bf16_t*)nullptr, (const float*)nullptr, 1.f, 3);
    else
        hipLaunchKernelGGL((last2_dgrad_kernel<0, false>), dim3(grid), dim3(LT_NT), 0, st, dsr, dsr_nstride, w, (const bf16_t*)aux, (const uint2*)sign_bits, neg, (bf16_t*)dst, N, H, W, mask_mode, (const bf16_t*)nullptr, (const float*)nullptr, 1.f, 3);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// bf16 conv3x3 from a planar fp32 source of 1 or 3 channels to 64 pixel-major channels with PACKED weights ([9][64][16], as every engine
// packs them for the generic kernel's (3, 1, 16, 16, planar, 64) shape): + bias, LeakyReLU, activation-gradient mask from `aux`.
// VSR_ERR_UNSUPPORTED: the caller uses the generic kernel.
int vsr_launch_planar_c64_conv(const ConvArgs& a, hipStream_t st) {
    if (a.nz != 1 || a.in_step != 1 || a.src_oy[0] || a.src_ox[0] || a.Hs != a.H || a.Ws != a.W || a.out_step != 1 || a.out_oy[0] || a.out_ox[0] ||
        a.Hd != a.H || a.Wd != a.W || !a.src[0] || !a.dst[0] || a.res[0] || a.CD != 64 || a.cout_real != 64 || a.sign_out[0] ||
        (a.act != ACT_NONE && a.act != ACT_LEAKY) || (a.aux[0] && a.mask_mode != MASK_RELU && a.mask_mode != MASK_LEAKY) ||
        a.dst_nstride != pm_image_elems(a.H, a.W, 64))
        return VSR_ERR_UNSUPPORTED;
    const int pc = a.planar_c ? a.planar_c : 3;
    if (pc < 1 || pc > 3 || (long long)3 * a.H * a.W * 4 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
    const int tiles = a.N * cdiv(a.W, LT_W) * cdiv(a.H, LT_H);
    const int grid = tiles < 256 * 8 ? tiles : 256 * 8;
    const float slope = a.act == ACT_LEAKY ? vsr_slope(a.leaky_slope) : 1.f;
    const float neg = (a.aux[0] && a.mask_mode == MASK_LEAKY) ? vsr_slope(a.leaky_slope) : 0.f;
    const float* src = reinterpret_cast<const float*>(a.src[0]);
    if (a.aux[0])
        hipLaunchKernelGGL((last2_dgrad_kernel<1, true>), dim3(grid), dim3(LT_NT), 0, st, src, a.src_nstride[0], (const float*)nullptr, (const bf16_t*)a.aux[0], (const uint2*)nullptr, neg,
                           (bf16_t*)a.dst[0], a.N, a.H, a.W, a.mask_mode, (const bf16_t*)a.wpack, a.bias, slope, pc);
    else
        hipLaunchKernelGGL((last2_dgrad_kernel<0, true>), dim3(grid), dim3(LT_NT), 0, st, src, a.src_nstride[0], (const float*)nullptr, (const bf16_t*)nullptr, (const uint2*)nullptr, neg,
                           (bf16_t*)a.dst[0], a.N, a.H, a.W, a.mask_mode, (const bf16_t*)a.wpack, a.bias, slope, pc);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// bf16 conv3x3 64 -> cout_real <= 4 channels, planar fp32 destination (+ bias, + planar residual, + bilinear x4 skip);
// returns VSR_ERR_UNSUPPORTED for anything else (the caller then uses the generic kernel).
int vsr_launch_c64_to_planar(const ConvArgs& a, hipStream_t st) {
    if (a.nz != 1 || a.act != ACT_NONE || a.cout_real < 1 || a.cout_real > 4 || a.in_step != 1 || a.src_oy[0] != 0 || a.src_ox[0] != 0 ||
        a.Hs != a.H || a.Ws != a.W || a.out_step != 1 || a.Hd != a.H || a.Wd != a.W || !a.src[0] || a.aux[0] || a.res[0])
        return VSR_ERR_UNSUPPORTED;
    if (pm_image_elems(LT_H + 4, a.Ws, 64) * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;     // in-tile source offsets are 32-bit
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(c64_to_planar_kernel), 2 * FP_TILE); if (rc != VSR_OK) return rc; }
    const int num_cus = vsr_num_cus();
    const int tiles = a.N * cdiv(a.W, LT_W) * cdiv(a.H, LT_H);
    const int grid = tiles < num_cus ? tiles : num_cus;
    hipLaunchKernelGGL(c64_to_planar_kernel, dim3(grid), dim3(FP_NT), 2 * FP_TILE, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// Partial slabs of the weight gradient of a 64 -> 3 conv3x3 with a planar fp32 cotangent (bf16 X, pixel-major 64 channels);
// *nslabs = slabs written, in the layout of vsr_wgrad_slab_dims(3, 64, 16).
int vsr_launch_last2_wgrad(const void* x, const float* dy, long long dy_nstride, float* slab, int slab_stride, int N, int H, int W,
                           int* nslabs, hipStream_t st, int pc) {
    if (!x || !dy || !slab || !nslabs || N < 1 || H < 1 || W < 1 || pc < 1 || pc > 3) return VSR_ERR_BADARG;
    if (pm_image_elems(LT_H + 4, W, 64) * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
    constexpr int LDS = (3 * 144 * 64 + 4 * 64) * 4 > 2 * LW_TILE ? (3 * 144 * 64 + 4 * 64) * 4 : 2 * LW_TILE;     // 111,616 B
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(last2_wgrad_kernel), LDS); if (rc != VSR_OK) return rc; }
    const int num_cus = vsr_num_cus();
    const int tiles = N * cdiv(W, LT_W) * cdiv(H, LT_H);
    const int grid = tiles < num_cus ? tiles : num_cus;
    *nslabs = grid;
    if ((W & 3) == 0 && (dy_nstride & 3) == 0 && (reinterpret_cast<size_t>(dy) & 15) == 0) {     // 16-byte dY chunks: the ring kernel
        constexpr int LDSR = 3 * LW_TILE + 3 * LW_DY;                                                // 147,456 B
        static VsrDevOnce once_r;
        { const int rc = vsr_set_max_dynamic_lds(once_r, reinterpret_cast<const void*>(last2_wgrad_ring_kernel), LDSR); if (rc != VSR_OK) return rc; }
        hipLaunchKernelGGL(last2_wgrad_ring_kernel, dim3(grid), dim3(FP_NT), LDSR, st, (const bf16_t*)x, dy, dy_nstride, slab, slab_stride, N, H, W, pc);
        HIP_CHECK_RET(hipGetLastError());
        return VSR_OK;
    }
    hipLaunchKernelGGL(last2_wgrad_kernel, dim3(grid), dim3(FP_NT), LDS, st, (const bf16_t*)x, dy, dy_nstride, slab, slab_stride, N, H, W, pc);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}
