// Weight / bias gradients of the convolutions on the BasicVSR path, on the matrix cores.
//
// What autograd's convolution_backward computes for the reference's nn.Conv2d layers
// (635 calls per step), restated as one split-K GEMM per layer:
//     dW[tap][cout][cin] = sum over pixels p of  dY[p][cout] * X[p + tap][cin]
//   M = cout, N = cin, K = pixels.  Both operands are pixel-major ([pixel][channel]), i.e.
//   K-strided, so the fragments are fetched from LDS with the CDNA4 transposing read
//   ds_read_b64_tr_b16 (bf16) -- or plain ds_read_b32 for the exact-fp32 build.
// A persistent 256-thread workgroup walks 8x32-pixel tiles of all frames ("segments") of the clip,
// keeps its slice of dW in MFMA accumulators the whole time (wave = one (cout-block, cin-block),
// all taps), and writes ONE fp32 partial slab at the end; wgrad_reduce sums the slabs into the
// OIHW fp32 gradient.  Batching the t frames of a clip into one launch makes the slab traffic
// (gridDim x 147 KB) small against the activations streamed (t x 132 MB for a 540p trunk conv).
// The bias gradient rides along: column sums of dY accumulated while staging dY.
#include <cstdlib>
#include "kernels.h"

namespace {

constexpr int TW = 32, TH = 8, NTHREADS = 256;

template <typename T> struct WElt;
template <> struct WElt<bf16_t> { static constexpr int CHB = 16; typedef uint4 chunk_t; };
struct wchunk32_t { uint4 a, b; };
template <> struct WElt<float> { static constexpr int CHB = 32; typedef wchunk32_t chunk_t; };

template <typename T> __device__ __forceinline__ typename WElt<T>::chunk_t wzero();
template <> __device__ __forceinline__ uint4 wzero<bf16_t>() { return make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ wchunk32_t wzero<float>() { wchunk32_t z; z.a = make_uint4(0, 0, 0, 0); z.b = z.a; return z; }

__device__ __forceinline__ uint4 wchunk3(float a, float b, float c, bf16_t*) {
    union { bf16_t h[8]; uint4 u; } t; t.u = make_uint4(0, 0, 0, 0);
    t.h[0] = (bf16_t)a; t.h[1] = (bf16_t)b; t.h[2] = (bf16_t)c; return t.u;
}
__device__ __forceinline__ wchunk32_t wchunk3(float a, float b, float c, float*) {
    wchunk32_t t; t.a = make_uint4(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), 0); t.b = make_uint4(0, 0, 0, 0); return t;
}
__device__ __forceinline__ void chunk_sum(const uint4& v, float* s) {
    union { uint4 u; bf16_t h[8]; } t; t.u = v;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += (float)t.h[j];
}
__device__ __forceinline__ void chunk_sum(const wchunk32_t& v, float* s) {
    s[0] += __uint_as_float(v.a.x); s[1] += __uint_as_float(v.a.y); s[2] += __uint_as_float(v.a.z); s[3] += __uint_as_float(v.a.w);
    s[4] += __uint_as_float(v.b.x); s[5] += __uint_as_float(v.b.y); s[6] += __uint_as_float(v.b.z); s[7] += __uint_as_float(v.b.w);
}

// LDS chunk swizzle for the transposing reads: with 128-byte pixel rows (CP = 8) rows q and q+2 of a
// 4-row tr block would share banks; flipping chunk bit 2 on every other pixel pair separates them.
template <int CP> __device__ __forceinline__ int wswz(int pix, int c) {
    return CP == 8 ? (c ^ (((pix >> 1) & 1) << 2)) : c;
}

// byte address of channel `ch` (multiple of 4 for tr reads) of pixel `pix`
template <typename T, int CP> __device__ __forceinline__ int lds_addr(int pix, int ch) {
    constexpr int CHB = WElt<T>::CHB;
    return (pix * CP + wswz<CP>(pix, ch >> 3)) * CHB + (ch & 7) * (int)sizeof(T);
}

__device__ __forceinline__ s16x4_t tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

__device__ uint4 g_wg_zero[4];      // 64 zero bytes: the source of every out-of-image chunk of the generic kernel's staging loops

// KS: kernel size; CX/COUT: LDS channel counts of X / dY (multiples of 16); *PLANAR: operand is a
// planar fp32 3-channel image (LR frames, SR cotangent) padded to 16 channels on the way into LDS.
template <typename T, int KS, int CX, bool XPLANAR, int COUT, bool DYPLANAR>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int PAD = KS / 2, KK = KS * KS;
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1, NPIXX = THH * TWH, NPIXY = TH * TW;
    constexpr int CPX = CX / 8, CPY = COUT / 8;
    constexpr int CHB = WElt<T>::CHB;
    constexpr int NCB = COUT >= 32 ? COUT / 32 : 1;
    constexpr int NIB = CX >= 32 ? CX / 32 : 1;
    constexpr int NT = NCB * NIB;                 // (cout-block, cin-block) pairs: 1, 2 or 4
    constexpr int TAPSPLIT = 4 / NT;              // waves sharing one pair split the taps
    // 7x7 (SPyNet): one kernel row of 7 taps per launch (a.tap_begin = 7 ky), or 49 taps would need ~400 accumulator
    // registers per wave; the slab collects all 49 taps over the 7 launches and is reduced once.
    constexpr int KT = KS == 7 ? 7 : KK;
    constexpr int NTAP = (KT + TAPSPLIT - 1) / TAPSPLIT;
    constexpr int COUTP = NCB * 32, CXP = NIB * 32;
    constexpr int XBYTES = NPIXX * CPX * CHB;
    typedef typename WElt<T>::chunk_t chunk_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lx = smem;
    char* ly = smem + XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int cb = wave % NCB, ib = (wave / NCB) % NIB, tap0 = wave / NT;

    f32x16_t acc[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    const int tiles_per_img = a.ntiles_x * a.ntiles_y;
    const int ntiles = a.N * tiles_per_img;

    for (int seg = 0; seg < a.nseg; ++seg) {
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int n = tile / tiles_per_img;
            const int tr = tile - n * tiles_per_img;
            const int ty0 = (tr / a.ntiles_x) * TH, tx0 = (tr % a.ntiles_x) * TW;
            __syncthreads();   // previous tile fully consumed
            // ---- stage X (haloed) ----
            if (XPLANAR) {
                const float* base = reinterpret_cast<const float*>(a.x[seg]) + (long long)n * a.x_nstride;
                const long long plane = (long long)a.Hx * a.Wx;
                const float* zf = reinterpret_cast<const float*>(g_wg_zero);
                for (int p0 = tid; p0 < NPIXX; p0 += 2 * NTHREADS) {      // two pixels per thread and round, six unconditional loads in flight
                    float cv[2][3];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int p = p0 + k * NTHREADS < NPIXX ? p0 + k * NTHREADS : 0;
                        const int ty = p / TWH, tx = p - ty * TWH;
                        const int vy = ty0 + ty - PAD, vx = tx0 + tx - PAD;
                        const bool in = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                        const float* sp = in ? base + (long long)(vy * a.x_step + a.x_oy) * a.Wx + (vx * a.x_step + a.x_ox) : zf;
                        const long long pl = in ? plane : 0;
                        cv[k][0] = sp[0]; cv[k][1] = sp[pl]; cv[k][2] = sp[2 * pl];
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int p = p0 + k * NTHREADS;
                        if (p < NPIXX) {
                            *reinterpret_cast<chunk_t*>(lx + (p * 2 + 0) * CHB) = wchunk3(cv[k][0], cv[k][1], cv[k][2], (T*)nullptr);
                            *reinterpret_cast<chunk_t*>(lx + (p * 2 + 1) * CHB) = wzero<T>();
                        }
                    }
                }
            } else {
                // batches of 4 chunks per thread, every load requested before the first LDS write; out-of-image chunks come from a zero
                // word (a guarded load compiles to a branch + vmcnt(0) per element: one memory round trip after the other, r03)
                const T* base = reinterpret_cast<const T*>(a.x[seg]) + (long long)n * a.x_nstride;
                const T* zsrc = reinterpret_cast<const T*>(g_wg_zero);
                for (int idx0 = tid; idx0 < NPIXX * CPX; idx0 += 4 * NTHREADS) {
                    chunk_t v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = idx0 + k * NTHREADS, ic = idx < NPIXX * CPX ? idx : 0;
                        const int p = ic / CPX, c = ic - p * CPX;
                        const int ty = p / TWH, tx = p - ty * TWH;
                        const int vy = ty0 + ty - PAD, vx = tx0 + tx - PAD;
                        const bool in = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                        const T* sp = in ? base + pm_off(vy * a.x_step + a.x_oy, vx * a.x_step + a.x_ox, c + a.x_coff, a.Wx, a.x_ctotal ? a.x_ctotal : CX) : zsrc;
                        v[k] = *reinterpret_cast<const chunk_t*>(sp);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = idx0 + k * NTHREADS;
                        if (idx < NPIXX * CPX) {
                            const int p = idx / CPX, c = idx - p * CPX;
                            *reinterpret_cast<chunk_t*>(lx + (p * CPX + wswz<CPX>(p, c)) * CHB) = v[k];
                        }
                    }
                }
            }
            // ---- stage dY (no halo) + bias partial sums ----
            if (DYPLANAR) {
                const float* base = reinterpret_cast<const float*>(a.dy[seg]) + (long long)n * a.dy_nstride;
                const long long plane = (long long)a.Hy * a.Wy;
                const float* zf = reinterpret_cast<const float*>(g_wg_zero);
                for (int p = tid; p < NPIXY; p += NTHREADS) {
                    const int ty = p / TW, tx = p - ty * TW;
                    const int vy = ty0 + ty, vx = tx0 + tx;
                    const int pc = a.dy_planar_c ? a.dy_planar_c : 3;
                    const bool in = vy < a.H && vx < a.W;
                    const float* sp = in ? base + (long long)(vy * a.dy_step + a.dy_oy) * a.Wy + (vx * a.dy_step + a.dy_ox) : zf;
                    const float c0 = sp[0], c1 = *(in && pc > 1 ? sp + plane : zf), c2 = *(in && pc > 2 ? sp + 2 * plane : zf);   // unconditional loads
                    bsum[0] += c0; bsum[1] += c1; bsum[2] += c2;
                    *reinterpret_cast<chunk_t*>(ly + (p * 2 + 0) * CHB) = wchunk3(c0, c1, c2, (T*)nullptr);
                    *reinterpret_cast<chunk_t*>(ly + (p * 2 + 1) * CHB) = wzero<T>();
                }
            } else {
                const T* base = reinterpret_cast<const T*>(a.dy[seg]) + (long long)n * a.dy_nstride;
                const T* zsrc = reinterpret_cast<const T*>(g_wg_zero);
                for (int idx0 = tid; idx0 < NPIXY * CPY; idx0 += 4 * NTHREADS) {   // NTHREADS % CPY == 0: chunk id is fixed per thread
                    chunk_t v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = idx0 + k * NTHREADS, ic = idx < NPIXY * CPY ? idx : 0;
                        const int p = ic / CPY, c = ic - p * CPY;
                        const int ty = p / TW, tx = p - ty * TW;
                        const int vy = ty0 + ty, vx = tx0 + tx;
                        const bool in = idx < NPIXY * CPY && vy < a.H && vx < a.W;
                        const T* sp = in ? base + pm_off(vy * a.dy_step + a.dy_oy, vx * a.dy_step + a.dy_ox, c + a.dy_coff, a.Wy, a.dy_ctotal ? a.dy_ctotal : COUT) : zsrc;
                        v[k] = *reinterpret_cast<const chunk_t*>(sp);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = idx0 + k * NTHREADS;
                        if (idx < NPIXY * CPY) {
                            const int p = idx / CPY, c = idx - p * CPY;
                            chunk_sum(v[k], bsum);
                            *reinterpret_cast<chunk_t*>(ly + (p * CPY + wswz<CPY>(p, c)) * CHB) = v[k];
                        }
                    }
                }
            }
            __syncthreads();

            // ---- K loop: 8 rows x 2 half-rows of 16 pixels ----
            for (int row = 0; row < TH; ++row) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int xk = half * 16;
                    if constexpr (sizeof(T) == 2) {
                        const int g2 = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = (lane & 3) * 4;
                        // channels 16..31 of a 16-channel operand do not exist: duplicate the low group
                        const int chy = cb * 32 + (COUT >= 32 ? 16 * g2 : 0) + p4;
                        const int chx = ib * 32 + (CX >= 32 ? 16 * g2 : 0) + p4;
                        const int py = row * TW + xk + 8 * h + q;
                        const s16x4_t a0 = tr_read(ly + lds_addr<T, CPY>(py, chy));
                        const s16x4_t a1 = tr_read(ly + lds_addr<T, CPY>(py + 4, chy));
                        bf16x8_t af;
                        { union { s16x4_t s[2]; bf16x8_t b; } u; u.s[0] = a0; u.s[1] = a1; af = u.b; }
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            const int tap = a.tap_begin + tap0 + i * TAPSPLIT;
                            if (tap0 + i * TAPSPLIT < KT) {
                                const int ky = tap / KS, kx = tap - ky * KS;
                                const int px = (row + ky) * TWH + xk + kx + 8 * h + q;
                                const s16x4_t b0 = tr_read(lx + lds_addr<T, CPX>(px, chx));
                                const s16x4_t b1 = tr_read(lx + lds_addr<T, CPX>(px + 4, chx));
                                union { s16x4_t s[2]; bf16x8_t b; } u; u.s[0] = b0; u.s[1] = b1;
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, u.b, acc[i], 0, 0, 0);
                            }
                        }
                    } else {
                        const int chy = cb * 32 + (COUT >= 32 ? l31 : (l31 & 15));
                        const int chx = ib * 32 + (CX >= 32 ? l31 : (l31 & 15));
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int py = row * TW + xk + 2 * j + h;
                            const float av = *reinterpret_cast<const float*>(ly + lds_addr<T, CPY>(py, chy));
#pragma unroll
                            for (int i = 0; i < NTAP; ++i) {
                                const int tap = a.tap_begin + tap0 + i * TAPSPLIT;
                                if (tap0 + i * TAPSPLIT < KT) {
                                    const int ky = tap / KS, kx = tap - ky * KS;
                                    const int px = (row + ky) * TWH + xk + kx + 2 * j + h;
                                    const float bv = *reinterpret_cast<const float*>(lx + lds_addr<T, CPX>(px, chx));
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- write this workgroup's partial slab: [tap][COUTP][CXP] then [COUTP] bias sums ----
    float* slab = a.slab + (long long)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int i = 0; i < NTAP; ++i) {
        const int tap = a.tap_begin + tap0 + i * TAPSPLIT;
        if (tap0 + i * TAPSPLIT < KT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[((long long)tap * COUTP + co) * CXP + ib * 32 + l31] = acc[i][r];
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);       // [NTHREADS][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < COUTP) {
        float s = 0.f;
        if (DYPLANAR) {
            if (tid < 3) for (int t = 0; t < NTHREADS; ++t) s += red[t * 8 + tid];
        } else if (tid < COUT) {
            const int c = tid >> 3, j = tid & 7;       // thread t staged chunk (t % CPY)
            for (int t = c; t < NTHREADS; t += CPY) s += red[t * 8 + j];
        }
        slab[(long long)KK * COUTP * CXP + tid] = s;
    }
}

// Sum the partial slabs and scatter into the reference's OIHW fp32 gradient tensors.
//   gw[(co*o_mul + o_add)][i_off + ci][ky][kx] (+)= sum_wg slab[wg][tap][co][ci]
//   gb[co*o_mul + o_add]                        (+)= sum_wg slab[wg][bias tail][co]      (if gb && add_bias)
__global__ void wgrad_reduce_kernel(const float* slab, int nwg, int slab_stride, int KK, int COUTP, int CXP,
                                    int cout_real, int cin_real, float* gw, int I_total, int i_off, int o_mul, int o_add,
                                    float* gb, int accumulate) {
    // block = 64 consecutive output elements x 4 slab groups: 64-lane rows of every slab are read
    // coalesced, 4 groups keep enough loads in flight; fixed summation order => bitwise reproducible.
    __shared__ float part[4][64];
    const int total = KK * cout_real * cin_real;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    const float* p = nullptr;
    float* d = nullptr;
    if (idx < total) {
        const int ci = idx % cin_real;
        const int co = (idx / cin_real) % cout_real;
        const int tap = idx / (cin_real * cout_real);
        p = slab + ((long long)tap * COUTP + co) * CXP + ci;
        d = gw + ((long long)(co * o_mul + o_add) * I_total + i_off + ci) * KK + tap;
    } else if (gb && idx < total + cout_real) {
        const int co = idx - total;
        p = slab + (long long)KK * COUTP * CXP + co;
        d = gb + co * o_mul + o_add;
    }
    float s = 0.f;
    if (p) {
        int w = grp;
        for (; w + 44 < nwg; w += 48) {                      // 12 slabs per thread in flight (4 were one memory round trip per 4 slabs: 16 us per launch); the same sums in the same order
            float v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = p[(long long)(w + 4 * k) * slab_stride];
            s += (v[0] + v[1]) + (v[2] + v[3]);
            s += (v[4] + v[5]) + (v[6] + v[7]);
            s += (v[8] + v[9]) + (v[10] + v[11]);
        }
        for (; w + 12 < nwg; w += 16) {
            const float v0 = p[(long long)w * slab_stride], v1 = p[(long long)(w + 4) * slab_stride];
            const float v2 = p[(long long)(w + 8) * slab_stride], v3 = p[(long long)(w + 12) * slab_stride];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; w < nwg; w += 4) s += p[(long long)w * slab_stride];
    }
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && d) {
        s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        *d = accumulate ? *d + s : s;
    }
}

// Pair-batched launch (WgradArgs::pair_*): slab (pair * ksplit + part) holds the [9][64][64] partial of pair = (view, cout block,
// cin slice) over the part-th pixel range.  One launch reduces every pair of a layer into its OIHW gradient (accumulating):
//   3x3 stride-1 layer (views == 1):  gw[o*64 + co][s*64 + ci][tap]        += sum_part
//   4x4 stride-2 layer (views == 4):  gw[o*64 + co][s*64 + ci][ky*4 + kx]  += sum_part   for the 4 taps view v owns
//                                     ((view, d) -> k as in conv_wide.hip: view 0: k = 2 d + 1 (d in {0,1}); view 1: k = 2 (d + 1) (d in {-1,0}))
__global__ void wgrad_reduce_pairs_kernel(const float* __restrict__ slab, int slab_stride, int ksplit, int nsl, int ncob, int views,
                                          float* __restrict__ gw, int cin_total) {
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;                          // < 9 * 64 * 64 (grid.x = 576)
    const int pair = blockIdx.y;
    const int sl = pair % nsl, o = (pair / nsl) % ncob, v = pair / (nsl * ncob);
    const int ci = idx & 63, co = (idx >> 6) & 63, tap = idx >> 12;
    float* d = nullptr;
    if (views == 1) {
        d = gw + ((long long)(o * 64 + co) * cin_total + sl * 64 + ci) * 9 + tap;
    } else {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int ky = (v >> 1) == 0 ? (dy >= 0 ? 2 * dy + 1 : -1) : (dy <= 0 ? 2 * (dy + 1) : -1);
        const int kx = (v & 1) == 0 ? (dx >= 0 ? 2 * dx + 1 : -1) : (dx <= 0 ? 2 * (dx + 1) : -1);
        if (ky >= 0 && kx >= 0) d = gw + ((long long)(o * 64 + co) * cin_total + sl * 64 + ci) * 16 + ky * 4 + kx;
    }
    float s = 0.f;
    if (d) {                                                      // tap is block-uniform (4096 = 64 * 64 elements per tap)
        const float* p = slab + (long long)pair * ksplit * slab_stride + idx;
        for (int w = grp; w < ksplit; w += 4) s += p[(long long)w * slab_stride];
    }
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && d) *d += (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

template <typename T, int KS, int CX, bool XP, int COUT, bool DP>
int launch_wgrad_inst(const WgradArgs& a0, int nwg, hipStream_t st) {
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1;
    constexpr int CHB = WElt<T>::CHB;
    constexpr int LDS_T = THH * TWH * (CX / 8) * CHB + TH * TW * (COUT / 8) * CHB;
    constexpr int LDS = LDS_T > NTHREADS * 8 * 4 ? LDS_T : NTHREADS * 8 * 4;
    static_assert(LDS <= 160 * 1024, "wgrad tiles do not fit LDS");
    auto kern = wgrad_kernel<T, KS, CX, XP, COUT, DP>;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), LDS); if (rc != VSR_OK) return rc; }
    WgradArgs a = a0;
    a.ntiles_x = cdiv(a.W, TW);
    a.ntiles_y = cdiv(a.H, TH);
    for (int tb = 0; tb < KS * KS; tb += (KS == 7 ? 7 : KS * KS)) {     // 7x7: one launch per kernel row
        a.tap_begin = tb;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), LDS, st, a);
        HIP_CHECK_RET(hipGetLastError());
    }
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------------
// bf16, 3x3, 64 -> 64: the shape that carries ~all weight-gradient FLOPs.  Same maths as wgrad_kernel,
// restructured around the memory system (this GEMM streams X and dY exactly once: it is HBM-bound and
// everything is about keeping loads in flight):
//   * ONE 512-thread workgroup per CU; both operand tiles are DOUBLE-BUFFERED in LDS (2 x 80 KiB = all of
//     a CU's LDS) and filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), issued
//     one whole tile ahead of the MFMAs.  The swizzle-free padded images below make the transposing reads
//     bank-conflict-free; out-of-image pixels read a 16-byte zero word, so border tiles need no special path.
// (Round 1's kernel of this shape -- all 8 waves doing DMA then MFMA, 41 us per frame-conv -- was replaced by
//  the producer / consumer kernel below, 29 us: profiles/r02_bench_kernel_stats_single_stream.csv.)
// ---------------------------------------------------------------------------------------------------
__device__ uint4 g_zero_chunk[2];

// LDS images mirror the blocked global layout so that a DMA piece reads long contiguous runs:
//   X  : [10 rows][8 chunks][36 slots of 16 B]  (34 haloed pixels + 2 pad; chunk stride 576 B = 144 banks, so the
//        4 chunks x 4 pixels x 8 B that a 32-lane half of ds_read_b64_tr_b16 touches fall on 64 distinct banks)
//   dY : [ 8 rows][8 chunks][35 slots]          (32 pixels + 3 pad: 560 B = 140 banks, 12 mod 64 -- the two chunks c, c + 1 that a half
//        of an A read touches overlapped in one pixel's four banks: 27.6 % of this kernel's LDS cycles were bank conflicts, r03
//        PMC.  r04: the pixels of the ODD chunks sit one slot later (slot = pixel + (c & 1), still inside the chunk's 35 slots), which
//        makes the distance 144 banks = 16 mod 64 like the X image's: conflict-free.  Measured: DESIGN 4.2.)
// 2 x (46,080 + 35,840) B = exactly the 160 KiB of a CU.
constexpr int XS = 36, YS = 35;                      // slots per chunk row
constexpr int XROW = 8 * XS * 16, YROW = 8 * YS * 16;   // 4,608 / 4,480 bytes per tile row
constexpr int DXB = (TH + 2) * XROW;                 // 46,080
constexpr int DYB = TH * YROW;                       // 35,840
constexpr int DSET = DXB + DYB;                      // 81,920 per buffer set
constexpr int DX_SLOTS = (TH + 2) * 8 * XS, DY_SLOTS = TH * 8 * YS;   // 2,880 / 2,240
constexpr int DX_PIECES = DX_SLOTS / 64;             // 45
constexpr int DY_PIECES = DY_SLOTS / 64;             // 35
constexpr int DNT = 512;
static_assert(DX_SLOTS % 64 == 0 && DY_SLOTS % 64 == 0, "whole DMA pieces");

// ---------------------------------------------------------------------------------------------------
// Producer / consumer form of the same kernel (round 2).  The DMA kernel above serialises a tile's memory
// phase (4.1 k cycles) and matrix phase (5.3 k): all 8 waves do both, and hipcc waits for the LDS-DMA in front
// of the first LDS read that follows it.  Here the two phases belong to different waves, as in
// conv3x3_persist.hip:
//   * waves 4-7 (producers): LDS-DMA of the next tile's X (10x34 haloed) and dY (8x32) images into the other
//     buffer set, 20 pieces of 1 KiB each, then the bias-gradient column sums of the CURRENT dY tile (their LDS
//     reads come BEFORE the DMA issue, so hipcc's alias wait finds nothing outstanding), then vmcnt(0) + the
//     tile's one workgroup barrier;
//   * waves 0-3 (consumers): wave = (cout block of 32, cin block of 32), all 9 taps, ALL 8 rows of the tile:
//     144 accumulator registers and no second copy of the accumulators (the DMA kernel's row-halves summed two
//     copies through LDS at the end).  v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it than on
//     32x32x16, MI355X_MICROARCH DVFS give-back item 7): a K step = one tile row of 32 pixels = 2 A fragments
//     (dY^T, cout x pixel) + 18 B fragments (X, pixel x cin, one per tap and cin half) = 40 transposing reads
//     for 36 MFMAs, the same LDS bytes per MFMA cycle as before.  Software pipeline in units of a group = (row,
//     ky) = 6 B fragments / 12 MFMAs: a B fragment register is re-requested for group g+2 right behind the two
//     MFMAs of group g that read it, so a request has ~1.8 groups (~350 cycles) to return and only two fragment
//     sets (48 registers) are live.  hipcc counts the lgkmcnt waits itself; sched_group_barrier pins the
//     interleave "2 MFMAs, 2 reads".
// ---------------------------------------------------------------------------------------------------
// Diagnostic build only (make ABL=<bits> ABLSRC=wgrad_mfma): bit 0: the producers issue only the first tile's DMA (consumer-only
// period); bit 1: the consumers skip the K loop (producer-only period).  Results are wrong by construction; only run time is read.
// bit 2: the round-3 dY image (no slot shift of the odd chunks: bank conflicts on the A reads; results unchanged) for the A/B;
// bit 3: nt instead of the default cache policy on the producers. LDS-DMA pieces (results unchanged).
// (r03: "B fragments for the ky = 0 groups only" measured -4 % with the DMA on, -7 % without: the K loop now shares an X row's
// fragments across ky.)
#ifdef VSR_ABL
#define WABL(bit) ((VSR_ABL >> (bit)) & 1)
__device__ unsigned long long g_wclk[256 * 4];          // [workgroup][cycles, 100 MHz ticks, consumer barrier-wait cycles, tiles] of the last launch
#else
#define WABL(bit) 0
#endif
#define DY_SHIFT (WABL(2) ? 0 : 1)      // slots by which the pixels of an odd chunk are shifted in the dY image
// X and dY are streamed: every byte is read once per launch (halo rows twice, by a neighbour tile of the same XCD in flight at the same
// time).  `nt` (aux = 2) on the LDS-DMA pieces: tools/bw_probe.hip reads 8 GiB at 6.4-7.0 TB/s with nt loads against 5.7-6.2 with plain
// ones -- but this kernel does not care: 233.2 / 231.7 us (nt) against 230.1 / 234.3 (default) per 7-frame launch, the same 349 k cycles:
// its producers are bound by the issue cost of their 80 pieces per tile, not by the policy.  Default policy kept; diagnostic bit 3 = nt.
#define WG_DMA_AUX (WABL(3) ? 2 : 0)
__global__ __launch_bounds__(DNT, 2) void wgrad3x3_c64_pc_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);            // 0: MFMA consumer, 1: LDS-DMA producer
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
#ifdef VSR_ABL
    unsigned long long clk_t0 = 0, clk_r0 = 0, clk_bar = 0, clk_tiles = 0;
    if (tid == 0) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int tiles_per_img = a.ntiles_x * a.ntiles_y;
    const int per_seg = a.N * tiles_per_img;
    const int total = a.nseg * per_seg;
    TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);
    int x_coff = a.x_coff, dy_coff = a.dy_coff, x_oy = a.x_oy, x_ox = a.x_ox;
    int slab_idx = blockIdx.x;
    if (a.pair_ksplit > 0) {
        // one launch for all 64x64 blocks of a wide layer's weight gradient: this workgroup owns pair (view, cout block,
        // cin slice) and the part-th contiguous range of tiles; the 8 XCDs each see one eighth of the pixel ranges for ALL
        // pairs, so a slice of a pixel range is fetched into one L2 and shared by the pairs that use it
        const int kq = a.pair_ksplit >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int pair = j / kq, part = xcd + 8 * (j - pair * kq);
        const int s = pair % a.pair_nsl, o = (pair / a.pair_nsl) % a.pair_ncob, v = pair / (a.pair_nsl * a.pair_ncob);
        x_coff = s * 8; dy_coff = o * 8;
        if (a.pair_views > 1) { x_oy = v >> 1; x_ox = v & 1; }
        walk.first = (int)((long long)total * part / a.pair_ksplit);
        walk.end = (int)((long long)total * (part + 1) / a.pair_ksplit);
        walk.stride = 1;
        slab_idx = pair * a.pair_ksplit + part;
    }
    float* slab = a.slab + (long long)slab_idx * a.slab_stride;

    if (role == 1) {
        // =================== producers ===================
        const char* zsrc = reinterpret_cast<const char*>(g_zero_chunk);
        // X / dY may be 64-channel SLICES of wider tensors (the discriminator's 128..512-channel layers): chunks per pixel
        const int xcp = (a.x_ctotal ? a.x_ctotal : 64) >> 3, ycp = (a.dy_ctotal ? a.dy_ctotal : 64) >> 3;
        constexpr int NPIECE = (DX_PIECES + DY_PIECES) / 4;           // 20 per wave
        static_assert((DX_PIECES + DY_PIECES) % 4 == 0, "even split over 4 producer waves");
        int rel[NPIECE];
        unsigned padmask = 0;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int piece = w4 + 4 * i;
            if (piece < DX_PIECES) {
                const int idx = piece * 64 + lane;
                const int row = idx / (8 * XS), rem = idx - row * (8 * XS);
                const int c = rem / XS, tx = rem - c * XS;
                const int dx = (tx - 1) * a.x_step + x_ox;
                rel[i] = (((((row - 1) * a.x_step) * pm_ws(a.Wx) + (dx >> 5)) * xcp + c) * 256 + (dx & 31) * 8) * 2;
                if (tx >= TW + 2) padmask |= 1u << i;
            } else {
                const int idx = (piece - DX_PIECES) * 64 + lane;
                const int row = idx / (8 * YS), rem = idx - row * (8 * YS);
                const int c = rem / YS, tx = rem - c * YS - (c & 1) * DY_SHIFT;       // tx: the pixel of this slot
                const int dx = (tx < 0 ? 0 : tx) * a.dy_step + a.dy_ox;
                rel[i] = ((((row * a.dy_step) * pm_ws(a.Wy) + (dx >> 5)) * ycp + c) * 256 + (dx & 31) * 8) * 2;
                if (tx < 0 || tx >= TW) padmask |= 1u << i;
            }
        }
        // (segment, image, tile row, tile column) of the walk, advanced by adds and carries (four runtime divisions per tile
        // before); a segment's base pointers are re-read from the kernel arguments only when the segment changes
        struct { int T, seg, n, ty, tx, sseg, sn, sty, stx; } it;
        {
            int r = walk.first;
            it.T = r; it.seg = r / per_seg; r -= it.seg * per_seg; it.n = r / tiles_per_img; r -= it.n * tiles_per_img;
            it.ty = r / a.ntiles_x; it.tx = r - it.ty * a.ntiles_x;
            r = walk.stride;
            it.sseg = r / per_seg; r -= it.sseg * per_seg; it.sn = r / tiles_per_img; r -= it.sn * tiles_per_img;
            it.sty = r / a.ntiles_x; it.stx = r - it.sty * a.ntiles_x;
        }
        auto advance = [&]() {
            it.T += walk.stride;
            it.tx += it.stx; if (it.tx >= a.ntiles_x) { it.tx -= a.ntiles_x; ++it.ty; }
            it.ty += it.sty; if (it.ty >= a.ntiles_y) { it.ty -= a.ntiles_y; ++it.n; }
            it.n += it.sn; if (it.n >= a.N) { it.n -= a.N; ++it.seg; }
            it.seg += it.sseg;
        };
        int cseg = -1;
        const char *xseg = nullptr, *yseg = nullptr;
        auto issue = [&](int s) {
            if (it.seg != cseg) {                                  // wave-uniform, a few times per launch
                cseg = it.seg;
                xseg = reinterpret_cast<const char*>(a.x[cseg]);
                yseg = reinterpret_cast<const char*>(a.dy[cseg]);
            }
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            const char* xb = xseg + (long long)it.n * a.x_nstride * 2;
            const char* yb = yseg + (long long)it.n * a.dy_nstride * 2;
            const char* xo = xb + pm_off(ty0 * a.x_step + x_oy, tx0 * a.x_step, x_coff, a.Wx, xcp * 8) * 2;      // tx0*step: multiple of 32
            const char* yo = yb + pm_off(ty0 * a.dy_step + a.dy_oy, tx0 * a.dy_step, dy_coff, a.Wy, ycp * 8) * 2;
            char* lxs = smem + s * DSET;
            const bool interior = ty0 >= 1 && ty0 + TH < a.H && tx0 >= 1 && tx0 + TW < a.W;              // wave-uniform
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const int piece = w4 + 4 * i;
                const bool isx = piece < DX_PIECES;
                const char* src = (isx ? xo : yo) + rel[i];
                const bool pad = (padmask >> i) & 1u;              // padding slots of the LDS images are never read: their lanes load nothing
                bool valid = true;
                if (!interior && !pad) {
                    if (isx) {
                        const int idx = piece * 64 + lane;
                        const int row = idx / (8 * XS), tx = (idx - row * (8 * XS)) % XS;
                        const int vy = ty0 + row - 1, vx = tx0 + tx - 1;
                        valid = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                    } else {
                        const int idx = (piece - DX_PIECES) * 64 + lane;
                        const int row = idx / (8 * YS), rem = idx - row * (8 * YS), c = rem / YS, tx = rem - c * YS - (c & 1) * DY_SHIFT;
                        valid = ty0 + row < a.H && tx0 + tx < a.W;
                    }
                }
                if (!valid) src = zsrc;
                char* dst = isx ? lxs + piece * 1024 : lxs + DXB + (piece - DX_PIECES) * 1024;
                if (!pad) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                           (__attribute__((address_space(3))) void*)dst, 16, 0, WG_DMA_AUX);
            }
        };
        float bsum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
        const int pt = tid - 256;                              // 0..255: chunk pt & 7, pixels (pt >> 3) + 32 i
        int cur = 0;
        if (it.T < walk.end) issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // the first tile is in LDS
        while (it.T < walk.end) {
            // The next tile's DMA goes out FIRST: the stream of 80 KiB per tile is what bounds the producers (24 GB/s per CU,
            // the chip's LDS-DMA rate), and anything in front of the issue delays the whole transfer.  The bias-gradient partial
            // sums of the CURRENT dY tile follow, read by inline asm (hipcc would put a vmcnt(0) in front of a plain LDS read
            // while LDS-DMA is in flight; the DMA writes the OTHER buffer set).
            const unsigned lyb = (unsigned)(cur * DSET + DXB + (pt & 7) * (YS * 16) + ((pt >> 3) + (pt & 1) * DY_SHIFT) * 16);
            advance();
            if (it.T < walk.end && !WABL(0)) issue(cur ^ 1);
            uint4 bq[8];
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:%9\n\tds_read_b128 %2, %8 offset:%10\n\tds_read_b128 %3, %8 offset:%11\n\t"
                         "ds_read_b128 %4, %8 offset:%12\n\tds_read_b128 %5, %8 offset:%13\n\tds_read_b128 %6, %8 offset:%14\n\tds_read_b128 %7, %8 offset:%15\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[4]), "=&v"(bq[5]), "=&v"(bq[6]), "=&v"(bq[7])
                         : "v"(lyb), "i"(YROW), "i"(2 * YROW), "i"(3 * YROW), "i"(4 * YROW), "i"(5 * YROW), "i"(6 * YROW), "i"(7 * YROW) : "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) chunk_sum(bq[i], bsum);   // pixel (pt >> 3) of tile row i, chunk pt & 7
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                   // next tile landed; the consumers are done with `cur`
            cur ^= 1;
        }
        __syncthreads();                                       // (A) consumers have stored their accumulators; LDS is free
        float* red = reinterpret_cast<float*>(smem);           // [256][8]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pt * 8 + j] = bsum[j];
        __syncthreads();                                       // (B)
    } else {
        // =================== consumers ===================
        const int q = lane >> 4, l15 = lane & 15, qq = l15 >> 2, p4 = l15 & 3;
        const int cb = w4 & 1, ib = w4 >> 1;
        // transposing-read lane addresses (cdna_hip_programming T10): lane 4 qq + p4 of 16-lane group q supplies pixel
        // 8 q + qq, channels 4 p4 .. 4 p4 + 3 of the fragment's 16; lane i of the group receives channel i of pixels
        // 8 q .. 8 q + 3 (second read, + 64 B: 8 q + 4 .. 8 q + 7) = the 16x16x32 operand's k = 8 q .. 8 q + 7.
        const int xoff = (ib * 4 + (p4 >> 1)) * (XS * 16) + (8 * q + qq) * 16 + (p4 & 1) * 8;
        const int yoff = DXB + (cb * 4 + (p4 >> 1)) * (YS * 16) + (8 * q + qq + (p4 >> 1) * DY_SHIFT) * 16 + (p4 & 1) * 8;   // chunk parity = p4 >> 1
        typedef union { s16x4_t s[2]; bf16x8_t b; } frag_u;
        f32x4_t acc[9][2][2];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) { acc[t][m][n][0] = 0.f; acc[t][m][n][1] = 0.f; acc[t][m][n][2] = 0.f; acc[t][m][n][3] = 0.f; }
        int cur = 0;
        __syncthreads();                                       // the first tile is in LDS
        for (int T = walk.first; T < walk.end; T += walk.stride) {
            const unsigned lx = (unsigned)(cur * DSET + xoff);  // LDS byte addresses (the dynamic segment starts at 0)
            const unsigned ly = (unsigned)(cur * DSET + yoff);
            // The K loop walks the tile's 10 haloed X rows R: the six B fragments of row R (kx x cin half) serve the groups
            // (ky, r = R - ky) of up to three output rows, so an X row is read from LDS once instead of three times (r03: 160
            // instead of 320 transposing reads per tile and wave; a read costs the wave ~4 issue cycles that the MFMA cadence does
            // not hide: 5.9 k -> 5.3 k cycles per tile).  A fragments (dY rows r) live in three slots r % 3 for the three steps
            // that use them.  Reads of step R + 1 are issued behind the MFMA pairs of step R (inline asm: hipcc sinks builtin
            // LDS reads back in front of their consumers); LDS operations return in order, and at the top of a step everything
            // issued during the previous one is needed: lgkmcnt(0).
            frag_u A[3][2], B[2][6];
#define PC_TRR(dst_, addr_, imm_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(imm_))
#define PC_LDA(r_, mb_, f_) { PC_TRR((f_).s[0], ly, (r_) * YROW + (mb_) * (2 * YS * 16)); PC_TRR((f_).s[1], ly, (r_) * YROW + (mb_) * (2 * YS * 16) + 64); }
#define PC_LDB(R_, fi_, f_) { PC_TRR((f_).s[0], lx, (R_) * XROW + ((fi_) & 1) * (2 * XS * 16) + ((fi_) >> 1) * 16);      \
                              PC_TRR((f_).s[1], lx, (R_) * XROW + ((fi_) & 1) * (2 * XS * 16) + ((fi_) >> 1) * 16 + 64); }
            __builtin_amdgcn_sched_barrier(0);
            PC_LDA(0, 0, A[0][0]) PC_LDA(0, 1, A[0][1])
            PC_LDB(0, 0, B[0][0]) PC_LDB(0, 1, B[0][1]) PC_LDB(0, 2, B[0][2]) PC_LDB(0, 3, B[0][3]) PC_LDB(0, 4, B[0][4]) PC_LDB(0, 5, B[0][5])
            __builtin_amdgcn_sched_barrier(0);
            // pair P of step R = (ky, fragment fi): two MFMAs (cout halves), then the reads scheduled behind it: B[R+1][P] for
            // P < 6; A[R+1] (slot (R+1) % 3 = the slot of row R-2, free once the ky = 2 group -- the first six pairs -- is
            // done) behind pairs 6 and 7, or behind pairs 4 and 5 of the one-group step R = 0.
#define PC_PAIR(R_, ky_, fi_, P_, NV_)                                                                                     \
            {                                                                                                              \
                constexpr int r_ = (R_) - (ky_), t_ = (ky_) * 3 + ((fi_) >> 1), nb_ = (fi_) & 1;                           \
                acc[t_][0][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r_ % 3][0].b, B[(R_) & 1][fi_].b, acc[t_][0][nb_], 0, 0, 0); \
                acc[t_][1][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r_ % 3][1].b, B[(R_) & 1][fi_].b, acc[t_][1][nb_], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                if ((P_) < 6 && (R_) + 1 <= 9) PC_LDB((R_) + 1, P_, B[((R_) + 1) & 1][P_])                                 \
                if ((R_) + 1 <= 7 && (P_) == ((NV_) >= 2 ? 6 : 4)) PC_LDA((R_) + 1, 0, A[((R_) + 1) % 3][0])                \
                if ((R_) + 1 <= 7 && (P_) == ((NV_) >= 2 ? 7 : 5)) PC_LDA((R_) + 1, 1, A[((R_) + 1) % 3][1])                \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }
#define PC_GRP(R_, ky_, P0_, NV_) PC_PAIR(R_, ky_, 0, (P0_), NV_) PC_PAIR(R_, ky_, 1, (P0_) + 1, NV_) PC_PAIR(R_, ky_, 2, (P0_) + 2, NV_) \
                                  PC_PAIR(R_, ky_, 3, (P0_) + 3, NV_) PC_PAIR(R_, ky_, 4, (P0_) + 4, NV_) PC_PAIR(R_, ky_, 5, (P0_) + 5, NV_)
#define PC_TOP { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            if (!WABL(1)) {
            PC_TOP PC_GRP(0, 0, 0, 1)                                                     // R = 0: output row 0
            PC_TOP PC_GRP(1, 1, 0, 2) PC_GRP(1, 0, 6, 2)                                  // R = 1: rows 0 (ky 1), 1 (ky 0)
            PC_TOP PC_GRP(2, 2, 0, 3) PC_GRP(2, 1, 6, 3) PC_GRP(2, 0, 12, 3)
            PC_TOP PC_GRP(3, 2, 0, 3) PC_GRP(3, 1, 6, 3) PC_GRP(3, 0, 12, 3)
            PC_TOP PC_GRP(4, 2, 0, 3) PC_GRP(4, 1, 6, 3) PC_GRP(4, 0, 12, 3)
            PC_TOP PC_GRP(5, 2, 0, 3) PC_GRP(5, 1, 6, 3) PC_GRP(5, 0, 12, 3)
            PC_TOP PC_GRP(6, 2, 0, 3) PC_GRP(6, 1, 6, 3) PC_GRP(6, 0, 12, 3)
            PC_TOP PC_GRP(7, 2, 0, 3) PC_GRP(7, 1, 6, 3) PC_GRP(7, 0, 12, 3)
            PC_TOP PC_GRP(8, 2, 0, 2) PC_GRP(8, 1, 6, 2)                                  // R = 8: rows 6 (ky 2), 7 (ky 1)
            PC_TOP PC_GRP(9, 2, 0, 1)                                                     // R = 9: row 7
            }
#undef PC_TOP
#undef PC_GRP
#undef PC_PAIR
#undef PC_LDB
#undef PC_LDA
#undef PC_TRR
#ifdef VSR_ABL
            unsigned long long b0; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b0) :: "memory");
#endif
            __syncthreads();                                   // next tile landed; everybody is done with `cur`
#ifdef VSR_ABL
            unsigned long long b1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b1) :: "memory");
            clk_bar += b1 - b0; ++clk_tiles;
#endif
            cur ^= 1;
        }
        // ---- ONE partial slab per workgroup: [tap][64 cout][64 cin].  Accumulator (tap, mb, nb), register j =
        // dW[tap][cout cb*32 + mb*16 + 4 q + j][cin ib*32 + nb*16 + l15] ----
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        slab[((long long)t * 64 + cb * 32 + m * 16 + 4 * q + j) * 64 + ib * 32 + n * 16 + l15] = acc[t][m][n][j];
        __syncthreads();                                       // (A)
        __syncthreads();                                       // (B) the producers' bias partial sums are in LDS
        if (tid < 64) {
            const float* red = reinterpret_cast<const float*>(smem);
            float s = 0.f;
            const int c = tid >> 3, j = tid & 7;               // producer thread pt summed chunk (pt & 7)
            for (int t = c; t < 256; t += 8) s += red[t * 8 + j];
            slab[9 * 64 * 64 + tid] = s;
        }
    }
#ifdef VSR_ABL
    if (tid == 0 && blockIdx.x < 256) {
        g_wclk[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
        g_wclk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        g_wclk[blockIdx.x * 4 + 2] = clk_bar;
        g_wclk[blockIdx.x * 4 + 3] = clk_tiles;
    }
#endif
}

int launch_wgrad_pc(const WgradArgs& a0, int nwg, hipStream_t st) {      // nwg workgroups = nwg slabs
    constexpr int LDS = 2 * DSET;                              // 163,840: all of a CU's LDS
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(wgrad3x3_c64_pc_kernel), LDS); if (rc != VSR_OK) return rc; }
    WgradArgs a = a0;
    a.ntiles_x = cdiv(a.W, TW);
    a.ntiles_y = cdiv(a.H, TH);
    hipLaunchKernelGGL(wgrad3x3_c64_pc_kernel, dim3(nwg), dim3(DNT), LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

#ifdef VSR_ABL
extern "C" int vsr_debug_read_wclk(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wclk), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -3;
}
#endif



#define VSR_WGRAD_SHAPES(X)           \
    X(3, 64, false, 64, false)        /* trunk / upsample / conv_last.0 */ \
    X(3, 16, true, 64, false)         /* stem, LR-frame part (conv.py:97) ; cleaner stem */ \
    X(3, 64, false, 16, true)         /* conv_last.2 (dY = planar SR cotangent) */ \
    X(1, 64, false, 64, false)        /* point_conv halves */ \
    X(7, 16, false, 32, false)        /* SPyNet (train_flow): 8->32 */ \
    X(7, 32, false, 64, false)        /* 32->64 */ \
    X(7, 64, false, 32, false)        /* 64->32 (bf16; fp32 runs it as two 32-channel halves: LDS) */ \
    X(7, 32, false, 32, false)        \
    X(7, 32, false, 16, false)        /* 32->16 */ \
    X(7, 16, false, 16, false)        /* 16->2 (dY = masked flow gradient, 16-channel padded) */

int vsr_launch_wgrad7x7_pc(int cx, int cout, const WgradArgs& a, int max_slabs, int* nslabs, hipStream_t st);   // wgrad7x7_pc.hip

// slab layout helper shared with the engine
void vsr_wgrad_slab_dims(int ks, int cx, int cout, int* coutp, int* cxp, int* stride) {
    const int ncb = cout >= 32 ? cout / 32 : 1, nib = cx >= 32 ? cx / 32 : 1;
    *coutp = ncb * 32;
    *cxp = nib * 32;
    *stride = ks * ks * (*coutp) * (*cxp) + (*coutp);
}

int vsr_launch_wgrad(int dtype, int ks, int cx, int x_planar, int cout, int dy_planar,
                     const WgradArgs& a, int nwg, int* nslabs, hipStream_t st) {
    if (a.nseg < 1 || a.nseg > VSR_WG_MAXSEG || nwg < 1 || !nslabs) return VSR_ERR_BADARG;
    *nslabs = nwg;
    {   // hot shape: LDS-DMA double-buffered kernel (needs an even slab count: 2 row-halves per workgroup)
        const bool force_generic = vsr_env().generic_wgrad;
        if (!force_generic && dtype == VSR_BF16 && ks == 3 && cx == 64 && !x_planar && cout == 64 && !dy_planar && nwg >= 2) {
            *nslabs = nwg / 2;                                 // one 512-thread workgroup per CU, one slab each
            return launch_wgrad_pc(a, nwg / 2, st);
        }
    }
    // 64 -> (1..3) channels with a planar fp32 cotangent (conv_last.2, the pre-clean out conv, the discriminator's conv_9): the streaming
    // kernel of hr_tail.hip, one slab per CU
    if (!vsr_env().generic_wgrad && dtype == VSR_BF16 && ks == 3 && cx == 64 && !x_planar && cout == 16 && dy_planar && a.nseg == 1 && a.x_step == 1 &&
        a.dy_step == 1 && !a.x_oy && !a.x_ox && !a.dy_oy && !a.dy_ox && a.Hx == a.H && a.Wx == a.W && a.Hy == a.H && a.Wy == a.W && !a.x_ctotal && !a.x_coff &&
        a.x_nstride == pm_image_elems(a.H, a.W, 64)) {
        const int rc = vsr_launch_last2_wgrad(a.x[0], reinterpret_cast<const float*>(a.dy[0]), a.dy_nstride, a.slab, a.slab_stride, a.N, a.H, a.W, nslabs, st,
                                              a.dy_planar_c ? a.dy_planar_c : 3);
        if (rc != VSR_ERR_UNSUPPORTED) return rc;
        *nslabs = nwg;
    }
    // SPyNet's 7x7 layers (train_flow): producer / consumer kernel, all seven kernel rows in one launch (wgrad7x7_pc.hip); the
    // caller's nwg is also what its slab buffer holds of this shape's partials
    if (!vsr_env().generic_wgrad && dtype == VSR_BF16 && ks == 7 && !x_planar && !dy_planar) {
        const int rc = vsr_launch_wgrad7x7_pc(cx, cout, a, nwg, nslabs, st);
        if (rc != VSR_ERR_UNSUPPORTED) return rc;
        *nslabs = nwg;
    }
#define X(KS, CX, XP, COUT, DP)                                                                        \
    if (ks == KS && cx == CX && (x_planar != 0) == XP && cout == COUT && (dy_planar != 0) == DP) {    \
        if (dtype == VSR_BF16) return launch_wgrad_inst<bf16_t, KS, CX, XP, COUT, DP>(a, nwg, st);    \
        if (dtype == VSR_F32) {                                                                       \
            if constexpr (KS == 7 && CX == 64) return VSR_ERR_UNSUPPORTED;                             \
            else return launch_wgrad_inst<float, KS, CX, XP, COUT, DP>(a, nwg, st);                   \
        }                                                                                             \
        return VSR_ERR_BADARG;                                                                        \
    }
    VSR_WGRAD_SHAPES(X)
#undef X
    return VSR_ERR_UNSUPPORTED;
}

int vsr_launch_wgrad_reduce(const float* slab, int nwg, int ks, int cx, int cout, int cout_real, int cin_real,
                            float* gw, int I_total, int i_off, int o_mul, int o_add, float* gb, int accumulate,
                            hipStream_t st) {
    int coutp, cxp, stride;
    vsr_wgrad_slab_dims(ks, cx, cout, &coutp, &cxp, &stride);
    const int total = ks * ks * cout_real * cin_real + (gb ? cout_real : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, slab, nwg, stride, ks * ks,
                       coutp, cxp, cout_real, cin_real, gw, I_total, i_off, o_mul, o_add, gb, accumulate);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// Every 64x64 block of a wide layer's weight gradient in ONE launch of the producer / consumer kernel + ONE reduction
// (bf16 only; a.x / a.dy describe the full tensors, the kernel derives each pair's slices and parity view).  The slab
// must hold nsl * ncob * views * ksplit partials of slab_stride floats.
int vsr_launch_wgrad_pairs(int dtype, const WgradArgs& a0, int nsl, int ncob, int views, int ksplit, float* gw, int cin_total, hipStream_t st) {
    if (dtype != VSR_BF16) return VSR_ERR_UNSUPPORTED;
    if (nsl < 1 || ncob < 1 || (views != 1 && views != 4) || ksplit < 8 || (ksplit & 7) || a0.nseg < 1 || a0.nseg > VSR_WG_MAXSEG) return VSR_ERR_BADARG;
    const int npairs = nsl * ncob * views;
    if ((long long)npairs * ksplit > VSR_WGRAD_MAX_PAIR_SLABS) return VSR_ERR_BADARG;
    constexpr int LDS = 2 * DSET;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(wgrad3x3_c64_pc_kernel), LDS); if (rc != VSR_OK) return rc; }
    WgradArgs a = a0;
    a.ntiles_x = cdiv(a.W, TW);
    a.ntiles_y = cdiv(a.H, TH);
    a.pair_ksplit = ksplit; a.pair_nsl = nsl; a.pair_ncob = ncob; a.pair_views = views;
    hipLaunchKernelGGL(wgrad3x3_c64_pc_kernel, dim3(npairs * ksplit), dim3(DNT), LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    hipLaunchKernelGGL(wgrad_reduce_pairs_kernel, dim3(9 * 64, npairs), dim3(256), 0, st, a.slab, a.slab_stride, ksplit, nsl, ncob, views, gw, cin_total);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}
