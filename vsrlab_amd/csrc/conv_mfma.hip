// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// Replaces every nn.Conv2d on the BasicVSR path of the reference:
//   core/modules/conv.py:85-86,97 (trunk 3x3), basicvsr.py:18 (1x1 fuse), upsampling.py:7
//   (3x3 + PixelShuffle), basicvsr.py:20-21 (reconstruction), RealBasicVSR/modules/spynet.py:16-18
//   (SPyNet 7x7) -- and, fed with flipped/transposed weights, their data gradients.
//
// GEMM view per tap (ky,kx):  D[cout][pixel] += W_tap[cout][cin] * X[cin][pixel + (ky,kx)]
//   A operand = weights (M = cout), B operand = pixels (N = 32 consecutive x), K = cin.
//   v_mfma_f32_32x32x16_bf16 (T = bf16) or 8 x v_mfma_f32_32x32x2_f32 (T = fp32, exact fp32);
//   both take "8 consecutive channels of one row" per lane, so one kernel body serves both.
// Work decomposition: a 256-thread workgroup (4 waves) owns an 8-row x 32-column pixel tile for
// all COUT channels; wave w owns rows 2w,2w+1.  The haloed input tile is staged once per source
// in LDS ([pixel][channel], XOR-swizzled 16-byte chunks => conflict-free ds_read_b128); the
// per-tap weight slab [cout][cin] is double-buffered through LDS, prefetched one tap ahead in
// registers (issue early / write late).  Two workgroups fit a CU (<= 76 KB LDS each in bf16), so
// one workgroup's staging overlaps the other's MFMA phase.
// Epilogue (fused, in accumulator layout): bias, ReLU/LeakyReLU(0.1), residual add,
// activation-gradient mask, pixel-shuffle placement, or planar fp32 store with residual /
// bilinear x4 skip.
#include <cstdlib>
#include "common.h"

namespace {

__device__ uint4 g_cm_zero[4];  // 64 zero bytes: the source of every out-of-image chunk

constexpr int TW = 32;          // tile width  (MFMA N)
constexpr int RW = 2;           // rows per wave
constexpr int TH = 4 * RW;      // tile height
constexpr int NTHREADS = 256;

// 8-channel chunks in registers are NATIVE vectors: arrays of HIP's uint4 class (the weight-slab prefetch registers) were
// not promoted to registers by hipcc and went through scratch memory in every instantiation of this kernel (r03: 469 scratch
// instructions in this file).
typedef __attribute__((ext_vector_type(4))) unsigned cu32x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned cu32x8_t;
template <typename T> struct Elt;
template <> struct Elt<bf16_t> {
    static constexpr int CHB = 16;                 // bytes per 8-channel chunk
    typedef bf16x8_t frag_t;
    typedef cu32x4_t chunk_t;
};
struct f32x8_t { float v[8]; };
typedef cu32x8_t chunk32_t;
template <> struct Elt<float> {
    static constexpr int CHB = 32;
    typedef f32x8_t frag_t;
    typedef chunk32_t chunk_t;
};

__device__ __forceinline__ void mma(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16_t& acc, const f32x8_t& a, const f32x8_t& b) {
    // k-slot j of the 32x32x2 step = channel {j (lanes 0-31), 8+j (lanes 32-63)} of the 16-group:
    // any consistent k order is a valid reduction order.
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename Elt<T>::frag_t lds_frag(const char* p) {
    return *reinterpret_cast<const typename Elt<T>::frag_t*>(p);
}

// chunk swizzle: pixel (or weight row) p, 8-channel chunk c of CP chunks per row
template <int CP> __device__ __forceinline__ int swz(int p, int c) {
    return c ^ ((p / (16 / CP)) & (CP - 1));
}

template <typename T> __device__ __forceinline__ typename Elt<T>::chunk_t zero_chunk();
template <> __device__ __forceinline__ cu32x4_t zero_chunk<bf16_t>() { return cu32x4_t{0u, 0u, 0u, 0u}; }
template <> __device__ __forceinline__ chunk32_t zero_chunk<float>() { return chunk32_t{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}; }

__device__ __forceinline__ cu32x4_t make_chunk3(float a, float b, float c, bf16_t*) {
    union { bf16_t h[8]; cu32x4_t u; } t;
    t.u = cu32x4_t{0u, 0u, 0u, 0u};
    t.h[0] = (bf16_t)a; t.h[1] = (bf16_t)b; t.h[2] = (bf16_t)c;
    return t.u;
}
__device__ __forceinline__ chunk32_t make_chunk3(float a, float b, float c, float*) {
    return chunk32_t{__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), 0u, 0u, 0u, 0u, 0u};
}

__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f(float v) { return v; }

template <typename T> struct Vec4;
template <> struct __attribute__((aligned(8))) Vec4<bf16_t> { bf16_t v[4]; };
template <> struct __attribute__((aligned(16))) Vec4<float> { float v[4]; };

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : slope * v;
    return v;
}

// PyTorch upsample_bilinear2d(align_corners=False) source index for scale `inv` = 1/4 (or 1/2: upscale = 2) (basicvsr.py:22)
__device__ __forceinline__ void bil4(int d, int in_size, int& i0, int& i1, float& l1, float inv) {
    float s = (d + 0.5f) * inv - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

#ifndef VSR_CONV_ROW_STAGES
#define VSR_CONV_ROW_STAGES 0      // r03: the row-stage form measured SLOWER (SPyNet 7x7 layers 13.7 vs 7.5 ms/step): kept for reference, off
#endif
// taps per weight stage: a whole kernel row for the single-source 7x7 layers whose tile + 7 slabs fit (<= 150 KB), else 1
template <int KS, int NSRC> constexpr int conv_taps_per_stage(int tile_bytes, int slab_bytes) {
    return (VSR_CONV_ROW_STAGES && KS == 7 && NSRC == 1 && tile_bytes + 7 * slab_bytes <= 150 * 1024) ? 7 : 1;
}

template <typename T, int KS, int NSRC, int CA, int CB, bool LASTPLANAR, int COUT, int EPI>
__global__ __launch_bounds__(NTHREADS) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TWH = TW + KS - 1;
    constexpr int THH = TH + KS - 1;
    constexpr int NPIX = THH * TWH;
    constexpr int CMAX = CA > CB ? CA : CB;
    constexpr int CHB = Elt<T>::CHB;
    constexpr int NCB = COUT / 32;
    constexpr int KK = KS * KS;
    constexpr int TILE_BYTES = NPIX * (CMAX / 8) * CHB;
    constexpr int SLAB_BYTES = COUT * (CMAX / 8) * CHB;
    constexpr int TPS = conv_taps_per_stage<KS, NSRC>(TILE_BYTES, SLAB_BYTES);
    typedef typename Elt<T>::chunk_t chunk_t;
    typedef typename Elt<T>::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_in = smem;
    char* lds_w = smem + TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int tx0 = blockIdx.x * TW;
    const int ty0 = blockIdx.y * TH;
    const int n = blockIdx.z / a.nz;
    const int z = blockIdx.z - n * a.nz;

    f32x16_t acc[NCB][RW];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][rw][i] = 0.f;

    const T* wz = reinterpret_cast<const T*>(a.wpack) + (long long)z * a.w_zstride;
    long long woff = 0;   // element offset of the current source's weight block inside wz

#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        constexpr int dummy = 0; (void)dummy;
        const bool last = (s == NSRC - 1);
        const int C = last ? CB : CA;            // folded per unrolled iteration
        const int CP = C / 8;
        if (s > 0) __syncthreads();              // previous source fully consumed

        // ---- stage the haloed input tile of source s ----------------------------------
        if (a.src[s] != nullptr) {
            if (last && LASTPLANAR) {
                // planar fp32, 3 real channels -> 16-channel padded pixels
                const float* base = reinterpret_cast<const float*>(a.src[s]) + (long long)n * a.src_nstride[s];
                const long long plane = (long long)a.Hs * a.Ws;
                const float* zf = reinterpret_cast<const float*>(g_cm_zero);
                const int pc = a.planar_c ? a.planar_c : 3;
                for (int p0 = tid; p0 < NPIX; p0 += 2 * NTHREADS) {        // two pixels per thread and round: six unconditional loads in flight
                    float cv[2][3];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int p = p0 + k * NTHREADS < NPIX ? p0 + k * NTHREADS : 0;
                        const int ty = p / TWH, tx = p - ty * TWH;
                        const int vy = ty0 + ty - PAD, vx = tx0 + tx - PAD;
                        const bool in = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                        const float* sp = in ? base + (long long)(vy * a.in_step + a.src_oy[s]) * a.Ws + (vx * a.in_step + a.src_ox[s]) : zf;
                        cv[k][0] = sp[0]; cv[k][1] = *(in && pc > 1 ? sp + plane : zf); cv[k][2] = *(in && pc > 2 ? sp + 2 * plane : zf);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int p = p0 + k * NTHREADS;
                        if (p < NPIX) {
                            *reinterpret_cast<chunk_t*>(lds_in + (p * 2 + swz<2>(p, 0)) * CHB) = make_chunk3(cv[k][0], cv[k][1], cv[k][2], (T*)nullptr);
                            *reinterpret_cast<chunk_t*>(lds_in + (p * 2 + swz<2>(p, 1)) * CHB) = zero_chunk<T>();
                        }
                    }
                }
            } else {
                // batches of 4 chunks per thread: every load of a batch is requested (out-of-image chunks from a zero word: a
                // conditional load compiles to a branch and a vmcnt(0) per element, i.e. one memory round trip after the other --
                // r03 ISA of the trunk-stem instantiation) before the first one is written to LDS
                const T* base = reinterpret_cast<const T*>(a.src[s]) + (long long)n * a.src_nstride[s];
                const T* zsrc = reinterpret_cast<const T*>(g_cm_zero);
                constexpr int SB = 4;
                for (int idx0 = tid; idx0 < NPIX * CP; idx0 += SB * NTHREADS) {
                    chunk_t v[SB];
#pragma unroll
                    for (int k = 0; k < SB; ++k) {
                        const int idx = idx0 + k * NTHREADS;
                        const int ic = idx < NPIX * CP ? idx : 0;
                        const int p = ic / CP, c = ic - p * CP;
                        const int ty = p / TWH, tx = p - ty * TWH;
                        const int vy = ty0 + ty - PAD, vx = tx0 + tx - PAD;
                        const bool in = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                        const T* sp = in ? base + pm_off(vy * a.in_step + a.src_oy[s], vx * a.in_step + a.src_ox[s], c, a.Ws, C) : zsrc;
                        v[k] = *reinterpret_cast<const chunk_t*>(sp);
                    }
#pragma unroll
                    for (int k = 0; k < SB; ++k) {
                        const int idx = idx0 + k * NTHREADS;
                        if (idx < NPIX * CP) {
                            const int p = idx / CP, c = idx - p * CP;
                            int sc;
                            if (CP == 8) sc = swz<8>(p, c); else if (CP == 4) sc = swz<4>(p, c); else sc = swz<2>(p, c);
                            *reinterpret_cast<chunk_t*>(lds_in + (p * CP + sc) * CHB) = v[k];
                        }
                    }
                }
            }
        }
        // a null source contributes zeros: skip its taps entirely (first frame of a propagation
        // direction, basicvsr.py:47,63: feat_prop = zeros)
        if (a.src[s] != nullptr) {
            const T* ws = wz + woff;
            constexpr int dummy2 = 0; (void)dummy2;
            const int SLAB_CHUNKS = COUT * CP;
            constexpr int MAXPT = (COUT * (CMAX / 8) + NTHREADS - 1) / NTHREADS;   // chunks per thread
            if constexpr (TPS > 1) {
                // 7x7 kernels (SPyNet): a STAGE = one kernel row of 7 taps.  The row's seven weight slabs sit in one LDS buffer; the
                // next row's slabs are requested into registers before the row's MFMAs and written behind them: two barriers
                // per 7 taps instead of two per tap (r03: the per-tap form ran the SPyNet layers at 0.54-0.64 PFLOP/s).
                constexpr int NST = KK / TPS;
                static_assert(KK % TPS == 0, "whole stages");
                chunk_t wreg[TPS][MAXPT];
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
                    for (int i = 0; i < MAXPT; ++i) {
                        const int idx = tid + i * NTHREADS;
                        wreg[tt][i] = *reinterpret_cast<const chunk_t*>(ws + ((long long)tt * SLAB_CHUNKS + (idx < SLAB_CHUNKS ? idx : 0)) * 8);
                    }
                for (int st = 0; st < NST; ++st) {
                    if (st > 0) __syncthreads();                      // every wave has finished reading the previous row's slabs
#pragma unroll
                    for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
                        for (int i = 0; i < MAXPT; ++i) {
                            const int idx = tid + i * NTHREADS;
                            if (idx < SLAB_CHUNKS) {
                                const int r = idx / CP, c = idx - r * CP;
                                int sc;
                                if (CP == 8) sc = swz<8>(r, c); else if (CP == 4) sc = swz<4>(r, c); else sc = swz<2>(r, c);
                                *reinterpret_cast<chunk_t*>(lds_w + tt * SLAB_BYTES + (r * CP + sc) * CHB) = wreg[tt][i];
                            }
                        }
                    __syncthreads();
                    if (st + 1 < NST) {
#pragma unroll
                        for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
                            for (int i = 0; i < MAXPT; ++i) {
                                const int idx = tid + i * NTHREADS;
                                wreg[tt][i] = *reinterpret_cast<const chunk_t*>(ws + ((long long)((st + 1) * TPS + tt) * SLAB_CHUNKS + (idx < SLAB_CHUNKS ? idx : 0)) * 8);
                            }
                    }
#pragma unroll
                    for (int tt = 0; tt < TPS; ++tt) {
                        const int ky = st, kx = tt;                  // TPS == KS: stage = kernel row
                        const char* wbuf = lds_w + tt * SLAB_BYTES;
#pragma unroll
                        for (int ks = 0; ks < C / 16; ++ks) {
                            const int ch = 2 * ks + h;
                            frag_t af[NCB], bf[RW];
#pragma unroll
                            for (int cb = 0; cb < NCB; ++cb) {
                                const int r = cb * 32 + l31;
                                int sc;
                                if (CP == 8) sc = swz<8>(r, ch); else if (CP == 4) sc = swz<4>(r, ch); else sc = swz<2>(r, ch);
                                af[cb] = lds_frag<T>(wbuf + (r * CP + sc) * CHB);
                            }
#pragma unroll
                            for (int rw = 0; rw < RW; ++rw) {
                                const int p = (wave * RW + rw + ky) * TWH + l31 + kx;
                                int sc;
                                if (CP == 8) sc = swz<8>(p, ch); else if (CP == 4) sc = swz<4>(p, ch); else sc = swz<2>(p, ch);
                                bf[rw] = lds_frag<T>(lds_in + (p * CP + sc) * CHB);
                            }
#pragma unroll
                            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                                for (int rw = 0; rw < RW; ++rw) mma(acc[cb][rw], af[cb], bf[rw]);
                        }
                    }
                }
                __syncthreads();
            } else {
            chunk_t wreg[MAXPT];
            // tap 0 slab
#pragma unroll
            for (int i = 0; i < MAXPT; ++i) {
                const int idx = tid + i * NTHREADS;
                wreg[i] = *reinterpret_cast<const chunk_t*>(ws + (long long)(idx < SLAB_CHUNKS ? idx : 0) * 8);     // unconditional: a guarded load is a branch + vmcnt(0) per element
            }
#pragma unroll
            for (int i = 0; i < MAXPT; ++i) {
                const int idx = tid + i * NTHREADS;
                if (idx < SLAB_CHUNKS) {
                    const int r = idx / CP, c = idx - r * CP;
                    int sc;
                    if (CP == 8) sc = swz<8>(r, c); else if (CP == 4) sc = swz<4>(r, c); else sc = swz<2>(r, c);
                    *reinterpret_cast<chunk_t*>(lds_w + (r * CP + sc) * CHB) = wreg[i];
                }
            }
            __syncthreads();

            for (int tap = 0; tap < KK; ++tap) {
                const int ky = tap / KS, kx = tap - ky * KS;
                const char* wbuf = lds_w + (tap & 1) * SLAB_BYTES;
                if (tap + 1 < KK) {
#pragma unroll
                    for (int i = 0; i < MAXPT; ++i) {
                        const int idx = tid + i * NTHREADS;
                        wreg[i] = *reinterpret_cast<const chunk_t*>(ws + ((long long)(tap + 1) * SLAB_CHUNKS + (idx < SLAB_CHUNKS ? idx : 0)) * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);         // the requests stay HERE, in front of the tap's MFMAs (hipcc sinks them to their use behind the MFMAs otherwise)
                }
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) {
                    const int ch = 2 * ks + h;
                    frag_t af[NCB], bf[RW];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int r = cb * 32 + l31;
                        int sc;
                        if (CP == 8) sc = swz<8>(r, ch); else if (CP == 4) sc = swz<4>(r, ch); else sc = swz<2>(r, ch);
                        af[cb] = lds_frag<T>(wbuf + (r * CP + sc) * CHB);
                    }
#pragma unroll
                    for (int rw = 0; rw < RW; ++rw) {
                        const int p = (wave * RW + rw + ky) * TWH + l31 + kx;
                        int sc;
                        if (CP == 8) sc = swz<8>(p, ch); else if (CP == 4) sc = swz<4>(p, ch); else sc = swz<2>(p, ch);
                        bf[rw] = lds_frag<T>(lds_in + (p * CP + sc) * CHB);
                    }
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int rw = 0; rw < RW; ++rw) mma(acc[cb][rw], af[cb], bf[rw]);
                }
                if (tap + 1 < KK) {
                    char* nbuf = lds_w + ((tap + 1) & 1) * SLAB_BYTES;
#pragma unroll
                    for (int i = 0; i < MAXPT; ++i) {
                        const int idx = tid + i * NTHREADS;
                        if (idx < SLAB_CHUNKS) {
                            const int r = idx / CP, c = idx - r * CP;
                            int sc;
                            if (CP == 8) sc = swz<8>(r, c); else if (CP == 4) sc = swz<4>(r, c); else sc = swz<2>(r, c);
                            *reinterpret_cast<chunk_t*>(nbuf + (r * CP + sc) * CHB) = wreg[i];
                        }
                    }
                }
                __syncthreads();
            }
        }
            }   // TPS == 1
        woff += (long long)KK * COUT * C;
    }

    // ---- epilogue (accumulator layout: lane = pixel column, registers = 4-channel groups) ----
    const int vx = tx0 + l31;
    if (vx >= a.W) return;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int vy = ty0 + wave * RW + rw;
        if (vy >= a.H) continue;
        const int oy = vy * a.out_step + a.out_oy[z];
        const int ox = vx * a.out_step + a.out_ox[z];
        if (EPI == EPI_NHWC) {
            // blocked pixel-major destination: chunk (co>>3) of pixel (oy,ox), then channel (co&7) inside it;
            // for out_step == 1 a wave instruction below covers 512 contiguous bytes
            const long long pix = (long long)n * a.dst_nstride + pm_off(oy, ox, 0, a.Wd, a.CD);
            T* dst = reinterpret_cast<T*>(a.dst[z]) + pix;
            const T* res = a.res[z] ? reinterpret_cast<const T*>(a.res[z]) + pix : nullptr;
            const T* aux = a.aux[z] ? reinterpret_cast<const T*>(a.aux[z]) + pix : nullptr;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * h;
                    if (co >= a.cout_real) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[cb][rw][4 * g + j];
                    if (a.bias) {
                        const float4 b = *reinterpret_cast<const float4*>(a.bias + (long long)z * a.bias_zstride + co);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], a.act, vsr_slope(a.leaky_slope));
                    if (res) {
                        const Vec4<T> r = *reinterpret_cast<const Vec4<T>*>(res + (co >> 3) * 256 + (co & 7));
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += to_f(r.v[j]);
                    }
                    if (aux) {
                        const Vec4<T> m = *reinterpret_cast<const Vec4<T>*>(aux + (co >> 3) * 256 + (co & 7));
                        const float neg = a.mask_mode == MASK_LEAKY ? vsr_slope(a.leaky_slope) : 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= (to_f(m.v[j]) > 0.f ? 1.f : neg);
                    }
                    Vec4<T> o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o.v[j] = (T)v[j];
                    *reinterpret_cast<Vec4<T>*>(dst + (co >> 3) * 256 + (co & 7)) = o;
                }
            }
        } else {
            // planar fp32 destination, cout_real <= 4 channels: they sit in registers 0..3 of the h=0 lanes
            if (h == 0) {
                const long long plane = (long long)a.Hd * a.Wd;
                float* dst = reinterpret_cast<float*>(a.dst[z]) + (long long)n * a.dst_nstride + (long long)oy * a.Wd + ox;
                int y0 = 0, y1 = 0, x0 = 0, x1 = 0; float ly = 0.f, lx = 0.f;
                if (a.base_lr) { const float binv = a.base_scale == 2 ? 0.5f : 0.25f; bil4(oy, a.base_h, y0, y1, ly, binv); bil4(ox, a.base_w, x0, x1, lx, binv); }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= a.cout_real) break;
                    float v = acc[0][rw][c];
                    if (a.bias) v += a.bias[(long long)z * a.bias_zstride + c];
                    v = act_apply(v, a.act, vsr_slope(a.leaky_slope));
                    if (a.pres) v += a.pres[(long long)n * a.dst_nstride + c * plane + (long long)oy * a.Wd + ox];
                    if (a.base_lr) {
                        const float* bp = a.base_lr + (long long)n * a.base_nstride + (long long)c * a.base_h * a.base_w;
                        const float v00 = bp[y0 * a.base_w + x0], v01 = bp[y0 * a.base_w + x1];
                        const float v10 = bp[y1 * a.base_w + x0], v11 = bp[y1 * a.base_w + x1];
                        v += (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                    }
                    dst[c * plane] = v;
                }
            }
        }
    }
}

template <typename T, int KS, int NSRC, int CA, int CB, bool LP, int COUT, int EPI>
int launch_inst(const ConvArgs& a, hipStream_t st) {
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1;
    constexpr int CMAX = CA > CB ? CA : CB;
    constexpr int CHB = Elt<T>::CHB;
    constexpr int TPS = conv_taps_per_stage<KS, NSRC>(THH * TWH * (CMAX / 8) * CHB, COUT * (CMAX / 8) * CHB);
    constexpr int LDS = THH * TWH * (CMAX / 8) * CHB + (TPS > 1 ? TPS : 2) * COUT * (CMAX / 8) * CHB;
    static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS of a CU");
    auto kern = conv_mfma_kernel<T, KS, NSRC, CA, CB, LP, COUT, EPI>;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), LDS); if (rc != VSR_OK) return rc; }
    dim3 grid(cdiv(a.W, TW), cdiv(a.H, TH), a.N * a.nz);
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

// Shape table: (KS, NSRC, CA, CB, last source planar, COUT template, EPI)
#define VSR_CONV_SHAPES(X)                  \
    X(3, 1, 64, 64, false, 64, EPI_NHWC)    /* trunk / upsample / conv_last.0 and their dgrads */ \
    X(3, 2, 64, 16, true, 64, EPI_NHWC)     /* trunk stem on cat(lr_i, feat)  conv.py:97        */ \
    X(3, 1, 16, 16, true, 64, EPI_NHWC)     /* dgrad of conv_last.2 (planar dSR -> 64ch), cleaner stem */ \
    X(3, 1, 64, 64, false, 32, EPI_PLANAR)  /* conv_last.2 (+ bilinear x4 skip), cleaner out   */ \
    X(3, 4, 64, 64, false, 64, EPI_NHWC)    /* dgrad of conv3x3+PixelShuffle(2)                 */ \
    X(1, 2, 64, 64, false, 64, EPI_NHWC)    /* point_conv 128->64  basicvsr.py:18               */ \
    X(1, 1, 64, 64, false, 64, EPI_NHWC)    /* its dgrad (two 64-channel outputs)               */ \
    X(7, 1, 16, 16, false, 32, EPI_NHWC)    /* SPyNet 8->32                                     */ \
    X(7, 1, 32, 32, false, 64, EPI_NHWC)    /* SPyNet 32->64                                    */ \
    X(7, 1, 64, 64, false, 32, EPI_NHWC)    /* SPyNet 64->32 ; 32->16 uses (7,1,32,32,..,32)    */ \
    X(7, 1, 32, 32, false, 32, EPI_NHWC)    \
    X(7, 1, 16, 16, false, 32, EPI_PLANAR)  /* SPyNet 16->2 (+ReLU) + flow_up residual          */

int vsr_launch_conv3x3_c64_persist(const ConvArgs& a, int num_cus, hipStream_t st);
int vsr_launch_c64_to_planar(const ConvArgs& a, hipStream_t st);      // hr_tail.hip
int vsr_launch_planar_c64_conv(const ConvArgs& a, hipStream_t st);    // hr_tail.hip
int vsr_launch_conv7x7_persist(const ConvArgs& a, int cin, int cop, int epi, int num_cus, hipStream_t st);   // conv7x7_persist.hip

// VSRLAB_AMD_GENERIC_CONV=1 routes the hot shape through the generic tiled kernel (A/B testing only; common.h: vsr_env()).
static bool vsr_force_generic_conv() { return vsr_env().generic_conv; }

// Host dispatcher (C++ linkage, used by the engine and by the C-ABI per-op entry points).
int vsr_launch_conv(int dtype, int ks, int nsrc, int ca, int cb, int last_planar, int cout_t, int epi,
                    const ConvArgs& a, hipStream_t st) {
    if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.nz < 1 || a.nz > VSR_MAX_Z) return VSR_ERR_BADARG;
    // the hot shape has its own persistent, weights-resident kernel (conv3x3_persist.hip)
    if (dtype == VSR_BF16 && ks == 3 && nsrc == 1 && ca == 64 && cb == 64 && !last_planar && cout_t == 64 && epi == EPI_NHWC &&
        ((a.in_step == 1 && a.src_oy[0] == 0 && a.src_ox[0] == 0 && a.Hs == a.H && a.Ws == a.W) ||
         (a.in_step == 2 && a.Hs == 2 * a.H && a.Ws == 2 * a.W && (unsigned)a.src_oy[0] < 2u && (unsigned)a.src_ox[0] < 2u)) &&
        a.CD == 64 && a.cout_real == 64 &&
        a.src[0] != nullptr && !vsr_force_generic_conv()) {
        const int ps = vsr_launch_conv3x3_c64_persist(a, vsr_num_cus(), st);
        if (ps != VSR_ERR_UNSUPPORTED) return ps;
    }
    if (a.unshuffle) return VSR_ERR_UNSUPPORTED;               // the phase-separated destination exists in the persistent kernel only
    // 64 -> (<= 4) channels with a planar destination: streaming kernel of hr_tail.hip
    if (dtype == VSR_BF16 && ks == 3 && nsrc == 1 && ca == 64 && cb == 64 && !last_planar && cout_t == 32 && epi == EPI_PLANAR &&
        !vsr_force_generic_conv()) {
        const int ps = vsr_launch_c64_to_planar(a, st);
        if (ps != VSR_ERR_UNSUPPORTED) return ps;
    }
    // planar fp32 (1 or 3 channels) -> 64 channels: the streaming kernel of hr_tail.hip (one K = 27 MFMA per 16 x 16 block)
    if (dtype == VSR_BF16 && ks == 3 && nsrc == 1 && last_planar && ca == 16 && cb == 16 && cout_t == 64 && epi == EPI_NHWC && !vsr_force_generic_conv()) {
        const int ps = vsr_launch_planar_c64_conv(a, st);
        if (ps != VSR_ERR_UNSUPPORTED) return ps;
    }
    // SPyNet's 7x7 layers: persistent kernels with streamed / resident weights (conv7x7_persist.hip)
    if (dtype == VSR_BF16 && ks == 7 && nsrc == 1 && ca == cb && !last_planar && !vsr_force_generic_conv()) {
        const int ps = vsr_launch_conv7x7_persist(a, ca, cout_t, epi, vsr_num_cus(), st);
        if (ps != VSR_ERR_UNSUPPORTED) return ps;
    }
#define X(KS, NSRC, CA, CB, LP, COUT, EPI)                                                             \
    if (ks == KS && nsrc == NSRC && ca == CA && cb == CB && (last_planar != 0) == LP && cout_t == COUT && epi == EPI) { \
        if (dtype == VSR_BF16) return launch_inst<bf16_t, KS, NSRC, CA, CB, LP, COUT, EPI>(a, st);    \
        if (dtype == VSR_F32) return launch_inst<float, KS, NSRC, CA, CB, LP, COUT, EPI>(a, st);      \
        return VSR_ERR_BADARG;                                                                        \
    }
    VSR_CONV_SHAPES(X)
#undef X
    return VSR_ERR_UNSUPPORTED;
}
