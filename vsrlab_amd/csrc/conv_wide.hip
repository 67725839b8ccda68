// Kernels of the GAN side of RealBasicVSR training (BASELINE config 3):
//   UNetDiscriminator (vsr/models/RealBasicVSR/modules/unet-discriminator.py:4-31): 3x3 stride-1 convs at 128..512 channels,
//   4x4 stride-2 convs, bilinear x2 upsampling between the U-Net levels, LeakyReLU(0.2);
//   SpectralConv (core/modules/conv.py:6-13): torch.nn.utils.spectral_norm around every inner conv;
//   AdversarialLoss (core/losses.py:66-74): BCE-with-logits against a constant target.
//
// "Wide" convolution = the implicit GEMM of conv_mfma.hip generalised in the channel direction: the input is a blocked
// pixel-major tensor with 64 k channels, consumed as k SLICES of 64 (one haloed 10x34x64 tile in LDS at a time, the 64x64
// weight slab of each (slice, tap) streamed through a double buffer); the output's 64 m channels are m independent
// workgroup columns (blockIdx.z).  A 4x4 stride-2 convolution (pad 1) is the sum of four 2x2-tap convolutions on the
// four PARITY VIEWS of its input (view_p(y, x) = in(2y + p_y, 2x + p_x)):
//     out(y) = sum_ky W[ky] in(2y + ky - 1):  ky odd  -> view 0 at y + dy, dy = (ky - 1) / 2 in {0, 1}
//                                             ky even -> view 1 at y + dy, dy = ky / 2 - 1   in {-1, 0}
// and its data gradient writes the four OUTPUT PHASES of a twice-as-large image, each a 2x2-tap convolution of dY:
//     dX(2q + p) = sum_ky W[ky] dY(q + dy):   p = 0 -> (dy 0, ky 1), (dy -1, ky 3);   p = 1 -> (dy +1, ky 0), (dy 0, ky 2)
// so both run on the SAME 3x3-tap kernel with a per-view / per-phase tap mask (masked taps are skipped: no wasted MFMAs).
// Two kernels: conv_wide_kernel (fp32 parity build: the generic 256-thread tile structure, two workgroups per CU) and, for bf16,
// conv_wide2_kernel (persistent, LDS-DMA weight slabs two steps ahead; see its header below): 0.88 PFLOP/s at config 3.
#include <cmath>
#include <cstdlib>
#include "kernels.h"
#include "../../include/vsrlab_hip.h"

namespace {

constexpr int TW = 32, RW = 2, TH = 8, NTHREADS = 256;
constexpr int TWH = TW + 2, THH = TH + 2, NPIX = THH * TWH;

template <typename T> struct WE;
template <> struct WE<bf16_t> { static constexpr int CHB = 16; typedef bf16x8_t frag_t; typedef uint4 chunk_t; };
struct wf32x8_t { float v[8]; };
struct wchunk_t { uint4 a, b; };
template <> struct WE<float> { static constexpr int CHB = 32; typedef wf32x8_t frag_t; typedef wchunk_t chunk_t; };

__device__ __forceinline__ void wmma(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void wmma(f32x16_t& acc, const wf32x8_t& a, const wf32x8_t& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ int wswz(int p, int c) { return c ^ ((p >> 1) & 7); }     // 8 chunks per 64-channel row
template <typename T> __device__ __forceinline__ typename WE<T>::chunk_t wzero();
template <> __device__ __forceinline__ uint4 wzero<bf16_t>() { return make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ wchunk_t wzero<float>() { wchunk_t z; z.a = make_uint4(0, 0, 0, 0); z.b = z.a; return z; }
__device__ __forceinline__ float wf(bf16_t v) { return (float)v; }
__device__ __forceinline__ float wf(float v) { return v; }
template <typename T> struct WV4;
template <> struct __attribute__((aligned(8))) WV4<bf16_t> { bf16_t v[4]; };
template <> struct __attribute__((aligned(16))) WV4<float> { float v[4]; };

struct WideArgs {
    const void* x; long long x_nstride; int xC, Hx, Wx;   // source: blocked pixel-major, xC channels per pixel
    int in_step;                                           // 2: the conv reads the four parity views of x
    int nsl;                                               // 64-channel slices of x that are reduced over
    int N, H, W;                                           // conv domain = output VIEW size
    const void* wpack;                                     // [ophase][cob][source][9][64][64] of T
    const float* bias;                                     // [ncob * 64] or null
    void* y; long long y_nstride; int yC, Hy, Wy;          // destination image (blocked pixel-major, yC channels)
    int out_step;                                          // 2: four output phases (transposed stride-2 conv)
    int ncob;
    unsigned tapmask[4];
    int act; float slope;
    void* y_act;                                           // optional: act(acc + bias), before the residual
    const void* res;                                       // optional residual (destination layout)
    void* y_pre;                                           // optional: value before the mask
    const void* aux;                                       // optional mask source: aux > 0 ? 1 : slope
};

template <typename T>
__global__ __launch_bounds__(NTHREADS) void conv_wide_kernel(const WideArgs a) {
    constexpr int CHB = WE<T>::CHB;
    constexpr int TILE_BYTES = NPIX * 8 * CHB;
    constexpr int SLAB_BYTES = 64 * 8 * CHB;
    typedef typename WE<T>::chunk_t chunk_t;
    typedef typename WE<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_in = smem;
    char* lds_w = smem + TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int nph = a.out_step == 2 ? 4 : 1;
    int zz = blockIdx.z;
    const int cob = zz % a.ncob; zz /= a.ncob;
    const int oph = zz % nph;
    const int n = zz / nph;
    const int nsrc = a.nsl * (a.in_step == 2 ? 4 : 1);
    const int xCP = a.xC >> 3;

    f32x16_t acc[2][RW];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][rw][i] = 0.f;

    const T* wbase = reinterpret_cast<const T*>(a.wpack) + ((long long)(oph * a.ncob + cob) * nsrc) * 9 * 4096;
    const T* xb = reinterpret_cast<const T*>(a.x) + (long long)n * a.x_nstride;
    const int xws = pm_ws(a.Wx);

    for (int s = 0; s < nsrc; ++s) {
        const int slice = s % a.nsl, par = s / a.nsl;
        const int oy = a.in_step == 2 ? (par >> 1) : 0, ox = a.in_step == 2 ? (par & 1) : 0;
        const unsigned mask = a.tapmask[a.in_step == 2 ? par : oph];
        __syncthreads();                                   // the previous source's tile and slabs are consumed
        // ---- stage the haloed tile of this slice / view: [pixel][8 chunks], XOR-swizzled ----
        for (int idx = tid; idx < NPIX * 8; idx += NTHREADS) {
            const int p = idx >> 3, c = idx & 7;
            const int ty = p / TWH, tx = p - ty * TWH;
            const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
            chunk_t v = wzero<T>();
            if (vy >= 0 && vy < a.H && vx >= 0 && vx < a.W) {
                const int sy = vy * a.in_step + oy, sx = vx * a.in_step + ox;
                const long long o = (((long long)sy * xws + (sx >> 5)) * xCP + slice * 8 + c) * 256 + (sx & 31) * 8;
                v = *reinterpret_cast<const chunk_t*>(xb + o);
            }
            *reinterpret_cast<chunk_t*>(lds_in + (p * 8 + wswz(p, c)) * CHB) = v;
        }
        const T* ws = wbase + (long long)s * 9 * 4096;
        int buf = 0;
        // first valid tap's slab
        int tap = __builtin_ctz(mask | 0x200u);
        chunk_t wreg[2];
        if (tap < 9) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wreg[i] = *reinterpret_cast<const chunk_t*>(ws + ((long long)tap * 512 + tid + i * NTHREADS) * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * NTHREADS, r = idx >> 3, c = idx & 7;
                *reinterpret_cast<chunk_t*>(lds_w + (r * 8 + wswz(r, c)) * CHB) = wreg[i];
            }
        }
        __syncthreads();
        while (tap < 9) {
            const unsigned rest = (mask >> (tap + 1)) << (tap + 1);
            const int next = __builtin_ctz(rest | 0x200u);
            if (next < 9) {
#pragma unroll
                for (int i = 0; i < 2; ++i) wreg[i] = *reinterpret_cast<const chunk_t*>(ws + ((long long)next * 512 + tid + i * NTHREADS) * 8);
            }
            const int ky = tap / 3, kx = tap - ky * 3;
            const char* wbuf = lds_w + buf * SLAB_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = 2 * ks + h;
                frag_t af[2], bf[RW];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int r = cb * 32 + l31;
                    af[cb] = *reinterpret_cast<const frag_t*>(wbuf + (r * 8 + wswz(r, ch)) * CHB);
                }
#pragma unroll
                for (int rw = 0; rw < RW; ++rw) {
                    const int p = (wave * RW + rw + ky) * TWH + l31 + kx;
                    bf[rw] = *reinterpret_cast<const frag_t*>(lds_in + (p * 8 + wswz(p, ch)) * CHB);
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rw = 0; rw < RW; ++rw) wmma(acc[cb][rw], af[cb], bf[rw]);
            }
            if (next < 9) {
                char* nbuf = lds_w + (buf ^ 1) * SLAB_BYTES;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = tid + i * NTHREADS, r = idx >> 3, c = idx & 7;
                    *reinterpret_cast<chunk_t*>(nbuf + (r * 8 + wswz(r, c)) * CHB) = wreg[i];
                }
            }
            __syncthreads();
            buf ^= 1;
            tap = next;
        }
    }

    // ---- epilogue (accumulator layout: lane = pixel column, registers = 4-channel groups) ----
    const int vx = tx0 + l31;
    if (vx >= a.W) return;
    const int yCP = a.yC >> 3;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int vy = ty0 + wave * RW + rw;
        if (vy >= a.H) continue;
        const int oyy = vy * a.out_step + (a.out_step == 2 ? (oph >> 1) : 0);
        const int oxx = vx * a.out_step + (a.out_step == 2 ? (oph & 1) : 0);
        const long long pix = (long long)n * a.y_nstride + (((long long)oyy * pm_ws(a.Wy) + (oxx >> 5)) * yCP + cob * 8) * 256 + (oxx & 31) * 8;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cb * 32 + 8 * g + 4 * h;
                const long long o = pix + (co >> 3) * 256 + (co & 7);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[cb][rw][4 * g + j];
                if (a.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(a.bias + cob * 64 + co);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (a.act == ACT_LEAKY) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : a.slope * v[j];
                } else if (a.act == ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                }
                WV4<T> t;
                if (a.y_act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t.v[j] = (T)v[j];
                    *reinterpret_cast<WV4<T>*>(reinterpret_cast<T*>(a.y_act) + o) = t;
                }
                if (a.res) {
                    const WV4<T> r = *reinterpret_cast<const WV4<T>*>(reinterpret_cast<const T*>(a.res) + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += wf(r.v[j]);
                }
                if (a.y_pre) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t.v[j] = (T)v[j];
                    *reinterpret_cast<WV4<T>*>(reinterpret_cast<T*>(a.y_pre) + o) = t;
                }
                if (a.aux) {
                    const WV4<T> m = *reinterpret_cast<const WV4<T>*>(reinterpret_cast<const T*>(a.aux) + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= (wf(m.v[j]) > 0.f ? 1.f : a.slope);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) t.v[j] = (T)v[j];
                *reinterpret_cast<WV4<T>*>(reinterpret_cast<T*>(a.y) + o) = t;
            }
        }
    }
}

// ---- weight packing for the wide kernel: OIHW fp32 (cout, cin, ks, ks) -> [ophase][cob][source][9][64 rows][64 cols] ----
//   mode 0: forward, 3x3 stride 1      rows = cout block, cols = cin slice, tap = ky*3+kx
//   mode 1: forward, 4x4 stride 2      sources = (parity view, cin slice); (view v, dy) -> ky = v ? 2*(dy+1) : 2*dy+1
//   mode 2: data gradient of mode 0    rows = cin block, cols = cout slice, flipped taps
//   mode 3: data gradient of mode 1    phases p: (dy, ky) in p = 0: {(0,1), (-1,3)}, p = 1: {(1,0), (0,2)}; rows = cin block
__device__ __forceinline__ int s2_fwd_k(int view, int d) {             // d in {-1,0,1}; -1 = tap not used
    if (view == 0) return d >= 0 ? 2 * d + 1 : -1;
    return d <= 0 ? 2 * (d + 1) : -1;
}
__device__ __forceinline__ int s2_bwd_k(int phase, int d) {
    if (phase == 0) return d == 0 ? 1 : (d == -1 ? 3 : -1);
    return d == 1 ? 0 : (d == 0 ? 2 : -1);
}
template <typename T>
__global__ void pack_wide_kernel(const float* __restrict__ w, T* __restrict__ dst, int cout, int cin, int mode, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx & 63), r = (int)((idx >> 6) & 63), tap = (int)((idx >> 12) % 9);
    long long rest = (idx >> 12) / 9;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    float v = 0.f;
    if (mode == 0) {
        const int nsl = cin / 64;
        const int slice = (int)(rest % nsl), cob = (int)(rest / nsl);
        v = w[((long long)(cob * 64 + r) * cin + slice * 64 + c) * 9 + tap];
    } else if (mode == 1) {
        const int nsl = cin / 64;
        const int slice = (int)(rest % nsl); rest /= nsl;
        const int view = (int)(rest % 4), cob = (int)(rest / 4);
        const int ky = s2_fwd_k(view >> 1, dy), kx = s2_fwd_k(view & 1, dx);
        if (ky >= 0 && kx >= 0) v = w[((long long)(cob * 64 + r) * cin + slice * 64 + c) * 16 + ky * 4 + kx];
    } else if (mode == 2) {
        const int nsl = cout / 64;                          // the gradient reduces over the conv's OUTPUT channels
        const int slice = (int)(rest % nsl), cob = (int)(rest / nsl);
        v = w[((long long)(slice * 64 + c) * cin + cob * 64 + r) * 9 + (8 - tap)];
    } else {
        const int nsl = cout / 64, ncob = cin / 64;
        const int slice = (int)(rest % nsl); rest /= nsl;
        const int cob = (int)(rest % ncob), phase = (int)(rest / ncob);
        const int ky = s2_bwd_k(phase >> 1, dy), kx = s2_bwd_k(phase & 1, dx);
        if (ky >= 0 && kx >= 0) v = w[((long long)(slice * 64 + c) * cin + cob * 64 + r) * 16 + ky * 4 + kx];
    }
    dst[idx] = (T)v;
}

// =====================================================================================================================
// conv_wide2: the bf16 kernel of the layers whose output has >= 128 channels (everything in the discriminator except
// conv_6 forward and conv_1's data gradient).  Same implicit GEMM, built like conv3x3_persist.hip:
//   * one 512-thread workgroup per CU, persistent over 8x32-pixel tiles of ONE output column = (output phase, 128-channel
//     block); 8 MFMA waves = 4 pixel groups (2 tile rows) x 2 halves of 64 output channels: 64 accumulator registers,
//     v_mfma_f32_16x16x32_bf16, fragment reads as inline-asm ds_read_b128 at base + immediate;
//   * a STEP = one active tap of one 64-channel source slice = 32 MFMAs per wave against a 16 KiB weight slab
//     [128 rows][64 k], which arrives by LDS-DMA two steps ahead into a ring of three buffers; the haloed 10x34 tile of
//     the NEXT source (slice / parity view, or the next tile's first source) is DMA'd into the other of two 42.5 KiB
//     tile buffers during the first step of the current one.  The weights are packed in execution order with the LDS
//     swizzle already applied (pack_wide2_kernel), so a slab is a linear 16 KiB copy and masked taps do not exist;
//   * one s_barrier per step (2 waves per SIMD x 32 MFMAs = 1024 matrix-core cycles between barriers) with COUNTED vmcnt
//     waits: at the end of step K only W(K+1) has to be in LDS, so the 2 (+5..6 tile) youngest DMA requests stay in flight.
//     hipcc does not see the inline-asm reads as LDS accesses, so it inserts no alias waits of its own.
//   * stores of an epilogue may complete out of order with loads, so the kernel drains vmcnt before an epilogue and once,
//     one step later, after it; in between nothing is needed that is not already in LDS.
constexpr int W2_NT = 512;
constexpr int W2_WSLAB = 128 * 64 * 2;                 // 16,384 B
constexpr int W2_TILE = NPIX * 128;                    // 43,520 B
constexpr int W2_LDS = 3 * W2_WSLAB + 2 * W2_TILE;     // 136,192 B
constexpr int W2_CHUNKS = NPIX * 8;                    // 2,720 16-byte chunks per tile
constexpr int W2_NPIECE = (W2_CHUNKS + 63) / 64;       // 43 DMA pieces of 1 KiB
constexpr int W2_TPW = (W2_NPIECE + 7) / 8;            // 6 per wave (waves 3..7: 5)

struct Wide2Args {
    const void* x; long long x_nstride; int xCP, Hx, Wx, in_step, nsl, nsrc;
    int N, H, W;
    const void* wpack; int nsteps;
    unsigned tapmask[4];
    const float* bias;
    void* y; long long y_nstride; int yCP, Hy, Wy, out_step, ncp;
    int act; float slope; const void* res; const void* aux; void* y_act;
};

__device__ uint4 g_w2_zero_chunk[1];

#define W2_GLDS16(src, dst)                                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

struct __attribute__((aligned(8))) w2bf4 { bf16_t v[4]; };

//   * KS = true (64-channel output blocks: conv_6 forward, conv_1's data gradient): the two wave halves split K instead of the
//     output channels -- half kh takes the channels 32 kh .. 32 kh + 31 of every slice, a slab is [2 taps][64 rows][64 k] (same
//     16 KiB, same 32 MFMAs per wave and step), and the halves' accumulators are summed through the idle tile buffer before the
//     epilogue, which waves kh = 0 run.
template <bool KS>
__global__ __launch_bounds__(W2_NT, 1) void conv_wide2_kernel(const Wide2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, ch = wave >> 2;
    const int l15 = lane & 15, q = lane >> 4;
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);   // as conv3x3_persist
    char* lds_w = smem;
    char* lds_t = smem + 3 * W2_WSLAB;

    const int col = blockIdx.y, cp = col % a.ncp, oph = col / a.ncp;
    const int ntx = cdiv(a.W, TW), nty = cdiv(a.H, TH), per_img = ntx * nty, total = a.N * per_img;
    const int my_tiles = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const long long Ktot = (long long)my_tiles * a.nsteps;
    const char* wcol = reinterpret_cast<const char*>(a.wpack) + (long long)col * a.nsteps * W2_WSLAB;
    const int WSx = pm_ws(a.Wx), WSy = pm_ws(a.Wy);
    const char* zsrc = reinterpret_cast<const char*>(g_w2_zero_chunk);

    // tile DMA: piece = wave + 8 i, LDS slot = [row ty][chunk c][34 pixels tx] x 16 B (the image conv3x3_persist uses)
    int rel[W2_TPW];
#pragma unroll
    for (int i = 0; i < W2_TPW; ++i) {
        const int idx = (wave + 8 * i) * 64 + lane;
        const int ty = idx / (8 * TWH), rem = idx - ty * (8 * TWH);
        const int c = rem / TWH, tx = rem - c * TWH;
        const int dx = (tx - 1) * a.in_step;
        rel[i] = (((((ty - 1) * a.in_step) * WSx + (dx >> 5)) * a.xCP + c) * 256 + (dx & 31) * 8) * 2;
    }
    auto issue_tile = [&](int tile, int s, int buf) {
        const int n = tile / per_img, r = tile - n * per_img;
        const int ty0 = (r / ntx) * TH, tx0 = (r % ntx) * TW;
        const int slice = s % a.nsl, view = s / a.nsl;
        const int oy = a.in_step == 2 ? (view >> 1) : 0, ox = a.in_step == 2 ? (view & 1) : 0;
        const char* org = reinterpret_cast<const char*>(a.x) +
            ((long long)n * a.x_nstride + (((long long)(ty0 * a.in_step + oy) * WSx + ((tx0 * a.in_step) >> 5)) * a.xCP + slice * 8) * 256 + ox * 8) * 2;
        char* dstb = lds_t + buf * W2_TILE;
        const bool interior = ty0 >= 1 && ty0 + TH < a.H && tx0 >= 1 && tx0 + TW < a.W;
#pragma unroll
        for (int i = 0; i < W2_TPW; ++i) {
            const int piece = wave + 8 * i;
            if (piece < W2_NPIECE) {
                const int idx = piece * 64 + lane;
                const char* src = org + rel[i];
                if (!interior) {
                    const int ty = idx / (8 * TWH), tx = (idx - ty * (8 * TWH)) % TWH;
                    const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                    if (!(vy >= 0 && vy < a.H && vx >= 0 && vx < a.W)) src = zsrc;
                }
                if (idx < W2_CHUNKS) W2_GLDS16(src, dstb + piece * 1024);
            }
        }
    };
    auto issue_w = [&](int kmod, int wb) {
        const char* src = wcol + (long long)kmod * W2_WSLAB + lane * 16;
        char* dst = lds_w + wb * W2_WSLAB;
        W2_GLDS16(src + wave * 1024, dst + wave * 1024);
        W2_GLDS16(src + (wave + 8) * 1024, dst + (wave + 8) * 1024);
    };

    // A (weights) lane base: row ch*64 + 16 mb + l15 of the slab, chunk (4 kk + q) ^ ((row >> 1) & 7); B (pixels) lane base as persist
    unsigned a_lane[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_lane[kk] = KS ? (unsigned)((l15 * 8 + ((4 * ch + q) ^ ((l15 >> 1) & 7))) * 16)
                                                   : (unsigned)(((ch * 64 + l15) * 8 + ((4 * kk + q) ^ ((l15 >> 1) & 7))) * 16);
    const unsigned b_lane = (unsigned)(3 * W2_WSLAB + pg * 2 * (TWH * 128) + q * (TWH * 16) + pxl * 16 + (KS ? ch * (4 * TWH * 16) : 0));

    // ---- prologue: first tile's first source, W(0), W(1) ----
    int tile = blockIdx.x;
    issue_tile(tile, 0, 0);
    issue_w(0, 0);
    if (Ktot > 1) issue_w(1 % a.nsteps, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    long long K = 0;
    int wbi = 0;                       // K % 3
    int k2 = 2 % a.nsteps, wb2 = 2;    // (K + 2) % nsteps, (K + 2) % 3
    int tb = 0;
    int post_epi = 2;                  // 2: first step after a drain (nothing to wait for), 1: drain again (stores), 0: counted waits
    bool prev_tile_dma = false;

    for (int t = 0; t < my_tiles; ++t, tile += gridDim.x) {
        f32x4_t acc[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && !(KS && ch)) bv = *reinterpret_cast<const float4*>(a.bias + (KS ? cp * 64 : cp * 128 + ch * 64) + mb * 16 + 4 * q);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { acc[mb][nb][0] = bv.x; acc[mb][nb][1] = bv.y; acc[mb][nb][2] = bv.z; acc[mb][nb][3] = bv.w; }
        }
        // epilogue operands (residual, activation-gradient mask source): requested now, used after the K loop (loads return in
        // order, so they are older than every DMA the counted waits below leave in flight)
        w2bf4 rr[4][4], mm[4][4];
        long long pixo[4];
        bool ok[4];
        {
            const int n = tile / per_img, r = tile - n * per_img;
            const int ty0 = (r / ntx) * TH, tx0 = (r % ntx) * TW;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int vy = ty0 + 2 * pg + (nb >> 1), vx = tx0 + (nb & 1) * 16 + pxl;
                ok[nb] = vy < a.H && vx < a.W && (!KS || ch == 0);
                const int oyy = vy * a.out_step + (a.out_step == 2 ? (oph >> 1) : 0);
                const int oxx = vx * a.out_step + (a.out_step == 2 ? (oph & 1) : 0);
                pixo[nb] = (long long)n * a.y_nstride + (((long long)oyy * WSy + (oxx >> 5)) * a.yCP + (KS ? cp * 8 : cp * 16 + ch * 8)) * 256 + (oxx & 31) * 8
                           + (q >> 1) * 256 + 4 * (q & 1);
                if (ok[nb]) {
                    if (a.res) {
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) rr[mb][nb] = *reinterpret_cast<const w2bf4*>(reinterpret_cast<const bf16_t*>(a.res) + pixo[nb] + mb * 512);
                    }
                    if (a.aux) {
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) mm[mb][nb] = *reinterpret_cast<const w2bf4*>(reinterpret_cast<const bf16_t*>(a.aux) + pixo[nb] + mb * 512);
                    }
                }
            }
        }
        for (int s = 0; s < a.nsrc; ++s) {
            const unsigned mask = a.tapmask[a.in_step == 2 ? s / a.nsl : oph];
            const bool last_src = s + 1 == a.nsrc;
            const bool has_next = !last_src || t + 1 < my_tiles;
            bool first = true, step_w = false, step_t = false;
            int half = 0;
            unsigned ab[2] = {0u, 0u};
            const unsigned bb = b_lane + (unsigned)(tb * W2_TILE);
            bf16x8_t fa[2][4], fb[2][4];
#define W2_DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define W2_LOADA(kk_, slot) { W2_DSR(fa[slot][0], ab[kk_], 0); W2_DSR(fa[slot][1], ab[kk_], 2048); W2_DSR(fa[slot][2], ab[kk_], 4096); W2_DSR(fa[slot][3], ab[kk_], 6144); }
#define W2_LOADB1(ky_, kx_, kk_, slot, nb) W2_DSR(fb[slot][nb], bb, (((nb) >> 1) + ky_) * (TWH * 128) + kk_ * (4 * TWH * 16) + (((nb) & 1) * 16 + kx_) * 16);
#define W2_LOADB(ky_, kx_, kk_, slot) { W2_LOADB1(ky_, kx_, kk_, slot, 0) W2_LOADB1(ky_, kx_, kk_, slot, 1) W2_LOADB1(ky_, kx_, kk_, slot, 2) W2_LOADB1(ky_, kx_, kk_, slot, 3) }
#define W2_MFMA(slot, mb, nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][mb], fb[slot][nb], acc[mb][nb], 0, 0, 0);
#define W2_MFMA_ROW(slot, mb) W2_MFMA(slot, mb, 0) W2_MFMA(slot, mb, 1) W2_MFMA(slot, mb, 2) W2_MFMA(slot, mb, 3)
#define W2_STEP_BEGIN                                                                                                   \
                const bool w_issued = K + 2 < Ktot;                                                                     \
                if (w_issued) issue_w(k2, wb2);                                                                         \
                const bool t_issued = first && has_next;                                                                \
                if (t_issued) { if (last_src) issue_tile(tile + gridDim.x, 0, tb ^ 1); else issue_tile(tile, s + 1, tb ^ 1); } \
                ab[0] = a_lane[0] + (unsigned)(wbi * W2_WSLAB); ab[1] = a_lane[1] + (unsigned)(wbi * W2_WSLAB);         \
                step_w = w_issued; step_t = t_issued;
#define W2_STEP_END                                                                                                     \
                if (post_epi == 2) { post_epi = 1; }                                                                    \
                else if (post_epi == 1 || !step_w) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); post_epi = 0; }   \
                else if (step_t || prev_tile_dma) { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }                  \
                else { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }                                               \
                __builtin_amdgcn_s_barrier();                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                prev_tile_dma = step_t;                                                                                 \
                first = false;                                                                                          \
                ++K; wbi = wbi == 2 ? 0 : wbi + 1; wb2 = wb2 == 2 ? 0 : wb2 + 1; k2 = k2 + 1 == a.nsteps ? 0 : k2 + 1;
#define W2_STEP(tap_)                                                                                                   \
            if (mask & (1u << (tap_))) {                                                                                \
                constexpr int ky_ = (tap_) / 3, kx_ = (tap_) % 3;                                                       \
                if (!KS) {                                                                                              \
                    W2_STEP_BEGIN                                                                                       \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_LOADA(0, 0) W2_LOADB(ky_, kx_, 0, 0)                                                             \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_MFMA_ROW(0, 0)                                                                                   \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_LOADA(1, 1)                                                                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_MFMA_ROW(0, 1)                                                                                   \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_LOADB(ky_, kx_, 1, 1)                                                                            \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_MFMA_ROW(0, 2) W2_MFMA_ROW(0, 3)                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_MFMA_ROW(1, 0) W2_MFMA_ROW(1, 1) W2_MFMA_ROW(1, 2) W2_MFMA_ROW(1, 3)                             \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_STEP_END                                                                                         \
                } else {                                                                                                \
                    if (half == 0) { W2_STEP_BEGIN }                                                                    \
                    else { ab[0] += 8192u; }             /* second tap of the slab: rows 64 .. 127 */                   \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_LOADA(0, 0) W2_LOADB(ky_, kx_, 0, 0)                                                             \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    W2_MFMA_ROW(0, 0) W2_MFMA_ROW(0, 1) W2_MFMA_ROW(0, 2) W2_MFMA_ROW(0, 3)                             \
                    __builtin_amdgcn_sched_barrier(0);                                                                  \
                    if (half == 1 || (mask >> ((tap_) + 1)) == 0u) { W2_STEP_END half = 0; }                            \
                    else half = 1;                                                                                      \
                }                                                                                                       \
            }
            W2_STEP(0) W2_STEP(1) W2_STEP(2) W2_STEP(3) W2_STEP(4) W2_STEP(5) W2_STEP(6) W2_STEP(7) W2_STEP(8)
#undef W2_STEP_BEGIN
#undef W2_STEP_END
#undef W2_STEP
#undef W2_MFMA_ROW
#undef W2_MFMA
#undef W2_LOADB
#undef W2_LOADB1
#undef W2_LOADA
#undef W2_DSR
            tb ^= 1;
        }
        // ---- epilogue: drain the DMA queue first (see the header), then registers -> blocked layout, 8-byte stores ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        post_epi = 2;
        prev_tile_dma = false;
        if (KS) {
            // sum the K halves through the tile buffer the last source used (tb was flipped: the OTHER buffer holds the next tile)
            float* ex = reinterpret_cast<float*>(lds_t + (tb ^ 1) * W2_TILE) + pg * 32 * 64 + lane;
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                if (ch == 1) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) ex[((m * 4 + nb) * 4 + j) * 64] = acc[2 * hp + m][nb][j];
                }
                __syncthreads();
                if (ch == 0) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[2 * hp + m][nb][j] += ex[((m * 4 + nb) * 4 + j) * 64];
                }
                __syncthreads();
            }
        }
        {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (ok[nb]) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const long long o = pixo[nb] + mb * 512;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[mb][nb][j];
                        if (a.act == ACT_LEAKY) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : a.slope * v[j];
                        } else if (a.act == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                        }
                        if (a.y_act) {
                            w2bf4 t;
#pragma unroll
                            for (int j = 0; j < 4; ++j) t.v[j] = (bf16_t)v[j];
                            *reinterpret_cast<w2bf4*>(reinterpret_cast<bf16_t*>(a.y_act) + o) = t;
                        }
                        if (a.res) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += (float)rr[mb][nb].v[j];
                        }
                        if (a.aux) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= ((float)mm[mb][nb].v[j] > 0.f ? 1.f : a.slope);
                        }
                        w2bf4 out;
#pragma unroll
                        for (int j = 0; j < 4; ++j) out.v[j] = (bf16_t)v[j];
                        *reinterpret_cast<w2bf4*>(reinterpret_cast<bf16_t*>(a.y) + o) = out;
                    }
                }
            }
        }
    }
}

// Weight image of conv_wide2: [column = ophase * ncp + cp][step k][128 rows][8 chunks of 8, chunk c stored at c ^ ((row >> 1) & 7)],
// steps in execution order: source (view-major, then slice), then the ACTIVE taps of that source in increasing tap index.
// Modes as pack_wide_kernel.
template <typename T>
__global__ void pack_wide2_kernel(const float* __restrict__ w, T* __restrict__ dst, int cout, int cin, int mode, int ks, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 7), cs = (int)((idx >> 3) & 7), r = (int)((idx >> 6) & 127);
    long long rest = idx >> 13;
    const bool s2 = mode == 1 || mode == 3;
    const int tps = s2 ? 4 : 9, spt = ks ? (tps + 1) / 2 : tps;                 // taps / steps per source
    const int nsl = (mode == 0 || mode == 1 ? cin : cout) / 64, nviews = mode == 1 ? 4 : 1;
    const int rows = (mode == 0 || mode == 1) ? cout : cin, ncp = rows / (ks ? 64 : 128);
    const int nsteps = nsl * nviews * spt;
    const int k = (int)(rest % nsteps), col = (int)(rest / nsteps);
    const int cp = col % ncp, oph = col / ncp;
    const int src = k / spt, st = k - src * spt;
    const int j = ks ? 2 * st + (r >> 6) : st;                                   // KS: slab = [2 taps][64 rows]
    const int slice = src % nsl, view = src / nsl;
    const int ci = ((cs ^ ((r >> 1) & 7)) << 3) + e;
    const int row = ks ? cp * 64 + (r & 63) : cp * 128 + r, colc = slice * 64 + ci;
    float v = 0.f;
    if (j < tps) {
        if (mode == 0) {
            v = w[((long long)row * cin + colc) * 9 + j];
        } else if (mode == 2) {
            v = w[((long long)colc * cin + row) * 9 + (8 - j)];
        } else if (mode == 1) {
            const int dy = ((view >> 1) ? -1 : 0) + (j >> 1), dx = ((view & 1) ? -1 : 0) + (j & 1);
            v = w[((long long)row * cin + colc) * 16 + s2_fwd_k(view >> 1, dy) * 4 + s2_fwd_k(view & 1, dx)];
        } else {
            const int dy = ((oph >> 1) ? 0 : -1) + (j >> 1), dx = ((oph & 1) ? 0 : -1) + (j & 1);
            v = w[((long long)colc * cin + row) * 16 + s2_bwd_k(oph >> 1, dy) * 4 + s2_bwd_k(oph & 1, dx)];
        }
    }
    dst[idx] = (T)v;
}

// Partial slabs of a 3x3 weight gradient taken on parity view `view` of a 4x4 stride-2 conv's input -> the 4x4 OIHW gradient
// (accumulated): gw[co0 + co][ci0 + ci][ky][kx] += sum_wg slab[wg][tap][co][ci] for the 4 taps the view owns.
__global__ void wgrad_reduce_s2_kernel(const float* __restrict__ slab, int nwg, int slab_stride, float* __restrict__ gw, int cin_total,
                                       int co0, int ci0, int view) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 9 * 64 * 64) return;
    const int ci = idx & 63, co = (idx >> 6) & 63, tap = idx >> 12;
    const int ky = s2_fwd_k(view >> 1, tap / 3 - 1), kx = s2_fwd_k(view & 1, tap % 3 - 1);
    if (ky < 0 || kx < 0) return;
    float s = 0.f;
    for (int wg = 0; wg < nwg; ++wg) s += slab[(long long)wg * slab_stride + idx];
    gw[((long long)(co0 + co) * cin_total + ci0 + ci) * 16 + ky * 4 + kx] += s;
}

// ---- bilinear x2 (align_corners=False) on blocked pixel-major tensors, and its adjoint ----
__device__ __forceinline__ void up2_src(int d, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)d + 0.5f) * 0.5f - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* f);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* f) {
    union { uint4 u; bf16_t h[8]; } t; t.u = *reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)t.h[j];
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* f) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float* f);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* f) {
    union { uint4 u; bf16_t h[8]; } t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.h[j] = (bf16_t)f[j];
    *reinterpret_cast<uint4*>(p) = t.u;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float* f) {
    reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// out (N, 2H, 2W, C) = up2(a [+ b]);  a, b: (N, H, W, C).  One thread = one SOURCE pixel x 8 channels -> its 2 x 2 output pixels from the
// clamped 3 x 3 source neighbourhood (x2 bilinear, align_corners=False, is separable with weights (1/4, 3/4) / (3/4, 1/4); at the image
// border the clamped neighbour carries the missing weight, which is exactly torch's source-index clamp): 2.25 (4.5 with b) 16-byte loads
// per output instead of 4 (8).  grid = (chunks of a source row / 256, source row, image); consecutive lanes = consecutive pixels of a
// 32-pixel segment, so loads are 512-byte runs and the two stores of an output row cover two full output segments between them.
template <typename T>
__global__ void up2_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int N, int H, int W, int C) {
    const int CP = C >> 3, W2 = 2 * W, WS = pm_ws(W);
    const long long in_img = pm_image_elems(H, W, C), out_img = pm_image_elems(2 * H, W2, C);
    const int y = blockIdx.y, n = blockIdx.z;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= WS * CP * 32) return;
    const int px = idx & 31, r = idx >> 5;
    const int c = r % CP, x = (r / CP) * 32 + px;
    if (x >= W) return;
    const int ys[3] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1};
    const int xs[3] = {x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
    float h0[3][8], h1[3][8];                                 // horizontally interpolated rows: output columns 2x, 2x + 1
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v[3][8];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long long o = (long long)n * in_img + pm_off(ys[i], xs[j], c, W, C);
            ld8<T>(a + o, v[j]);
            if (b) {
                float g[8];
                ld8<T>(b + o, g);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[j][k] += g[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h0[i][k] = 0.25f * v[0][k] + 0.75f * v[1][k];
            h1[i][k] = 0.75f * v[1][k] + 0.25f * v[2][k];
        }
    }
    float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        o00[k] = 0.25f * h0[0][k] + 0.75f * h0[1][k]; o01[k] = 0.25f * h1[0][k] + 0.75f * h1[1][k];
        o10[k] = 0.75f * h0[1][k] + 0.25f * h0[2][k]; o11[k] = 0.75f * h1[1][k] + 0.25f * h1[2][k];
    }
    T* ob = out + (long long)n * out_img;
    st8<T>(ob + pm_off(2 * y, 2 * x, c, W2, C), o00);
    st8<T>(ob + pm_off(2 * y, 2 * x + 1, c, W2, C), o01);
    st8<T>(ob + pm_off(2 * y + 1, 2 * x, c, W2, C), o10);
    st8<T>(ob + pm_off(2 * y + 1, 2 * x + 1, c, W2, C), o11);
}

// adjoint: din (N, H, W, C) = up2^T(dout (N, 2H, 2W, C)); optionally also din * (m > 0 ? 1 : slope) into dmask
template <typename T>
__global__ void up2_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, T* __restrict__ dmask, const T* __restrict__ m, float slope,
                               int N, int H, int W, int C) {
    const int CP = C >> 3, H2 = 2 * H, W2 = 2 * W, WS = pm_ws(W);
    const long long in_img = pm_image_elems(H, W, C), out_img = pm_image_elems(H2, W2, C);
    const int y = blockIdx.y, n = blockIdx.z;
    {
        const int idx = blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= WS * CP * 32) return;
        const int px = idx & 31, r = idx >> 5;
        const int c = r % CP, x = (r / CP) * 32 + px;
        if (x >= W) return;
        // x2 bilinear (align_corners=False) is separable with weights (1/4, 3/4) / (3/4, 1/4); its adjoint gathers the 4 x 4 outputs
        // 2y-1 .. 2y+2 with weights (1/4, 3/4, 3/4, 1/4) per axis.  At the border the output that clamped its missing neighbour onto this
        // pixel carries that weight too (row 0: Y = 0 counts 1.0; row H-1: Y = 2H-1 counts 1.0): branch-free, 16 loads.
        float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        if (y == 0) { wy[0] = 0.f; wy[1] += 0.25f; }
        if (y == H - 1) { wy[3] = 0.f; wy[2] += 0.25f; }
        if (x == 0) { wx[0] = 0.f; wx[1] += 0.25f; }
        if (x == W - 1) { wx[3] = 0.f; wx[2] += 0.25f; }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const T* ob = dout + (long long)n * out_img;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int Y = 2 * y - 1 + i;
            Y = Y < 0 ? 0 : (Y > H2 - 1 ? H2 - 1 : Y);
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int X = 2 * x - 1 + k;
                X = X < 0 ? 0 : (X > W2 - 1 ? W2 - 1 : X);
                float f[8];
                ld8<T>(ob + pm_off(Y, X, c, W2, C), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += wx[k] * f[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wy[i] * t[j];
        }
        const long long o = (long long)n * in_img + pm_off(y, x, c, W, C);
        if (din) st8<T>(din + o, acc);
        if (dmask) {
            float mv[8];
            ld8<T>(m + o, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= (mv[j] > 0.f ? 1.f : slope);
            st8<T>(dmask + o, acc);
        }
    }
}

// out = a + b  (blocked tensors of identical geometry: plain elementwise over the padded image)
template <typename T>
__global__ void add_pm_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float f[8], g[8];
        ld8<T>(a + i * 8, f); ld8<T>(b + i * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += g[j];
        st8<T>(out + i * 8, f);
    }
}

// out = g * (m > 0 ? 1 : slope)
template <typename T>
__global__ void mask_pm_kernel(const T* __restrict__ g, const T* __restrict__ m, T* __restrict__ out, float slope, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float f[8], mv[8];
        ld8<T>(g + i * 8, f); ld8<T>(m + i * 8, mv);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= (mv[j] > 0.f ? 1.f : slope);
        st8<T>(out + i * 8, f);
    }
}

// ---- spectral norm (torch/nn/utils/spectral_norm.py: compute_weight, n_power_iterations = 1, eps = 1e-12) ----
// W: (R, K) row-major = weight_orig.reshape(cout, -1).  One workgroup; R <= 512, K <= 8192: the matrices of a discriminator
// are a few MB and this runs once per forward, so a single-workgroup two-pass mat-vec is plenty.
__device__ __forceinline__ float blk_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
__global__ __launch_bounds__(1024) void spectral_norm_kernel(const float* __restrict__ W, float* __restrict__ u, float* __restrict__ v,
                                                             float* __restrict__ Wn, float* __restrict__ sigma_out, int R, int K,
                                                             int training, float eps) {
    __shared__ float red[16];
    __shared__ float su[512];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int r = tid; r < R; r += nt) su[r] = u[r];
    __syncthreads();
    if (training) {
        // v = normalize(W^T u)
        float part = 0.f;
        for (int k = tid; k < K; k += nt) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += W[(long long)r * K + k] * su[r];
            v[k] = s;
            part += s * s;
        }
        const float nv = fmaxf(sqrtf(blk_sum(part, red)), eps);
        for (int k = tid; k < K; k += nt) v[k] = v[k] / nv;
        __syncthreads();
        // u = normalize(W v): one wave per row
        const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
        for (int r = wave; r < R; r += nw) {
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += W[(long long)r * K + k] * v[k];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) su[r] = s;
        }
        __syncthreads();
        float p2 = 0.f;
        for (int r = tid; r < R; r += nt) p2 += su[r] * su[r];
        const float nu = fmaxf(sqrtf(blk_sum(p2, red)), eps);
        for (int r = tid; r < R; r += nt) { su[r] = su[r] / nu; u[r] = su[r]; }
        __syncthreads();
    }
    // sigma = u . (W v)
    {
        const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
        float part = 0.f;
        for (int r = wave; r < R; r += nw) {
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += W[(long long)r * K + k] * v[k];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) part += s * su[r];
        }
        const float sigma = blk_sum(part, red);
        if (tid == 0) sigma_out[0] = sigma;
        const long long n = (long long)R * K;
        for (long long i = tid; i < n; i += nt) Wn[i] = W[i] / sigma;
    }
}

// Training-mode power iteration in five small launches (the single-workgroup kernel above streams a <= 9.4 MB matrix three times through
// one CU: ~480 us per layer, 23 ms per GAN iteration).  Same arithmetic, fixed summation orders (no atomics):
//   sn_wtu: v <- W^T u (unnormalised), one thread per column, rows streamed coalesced
//   sn_norm: x <- x / max(||x||, eps) in one workgroup; optionally sigma = sum(x_old^2) / ||x_old|| (= u . (W v) for the fresh u)
//   sn_wv:  u <- W v (unnormalised), one wave per row
//   sn_scale: Wn = W / sigma
__global__ void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ v, int R, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 0;
    for (; r + 3 < R; r += 4) {
        s0 += W[(long long)r * K + k] * u[r];
        s1 += W[(long long)(r + 1) * K + k] * u[r + 1];
        s2 += W[(long long)(r + 2) * K + k] * u[r + 2];
        s3 += W[(long long)(r + 3) * K + k] * u[r + 3];
    }
    for (; r < R; ++r) s0 += W[(long long)r * K + k] * u[r];
    v[k] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(1024) void sn_norm_kernel(float* __restrict__ x, int n, float eps, float* __restrict__ sigma_out) {
    __shared__ float red[16];
    float part = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) part += x[i] * x[i];
    const float ss = blk_sum(part, red);
    const float nrm = fmaxf(sqrtf(ss), eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = x[i] / nrm;
    if (sigma_out && threadIdx.x == 0) sigma_out[0] = ss / nrm;          // sum_r (s_r / nrm) s_r
}
__global__ void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ u, int R, int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= R) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += W[(long long)r * K + k] * v[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) u[r] = s;
}
__global__ void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sigma, float* __restrict__ Wn, long long n) {
    const float sg = sigma[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) Wn[i] = W[i] / sg;
}
// backward in two launches: per-block partial sums of dWn * W (scratch), then the rank-1 corrected update
__global__ void sn_bwd_dot_kernel(const float* __restrict__ dWn, const float* __restrict__ W, float* __restrict__ partial, long long n) {
    __shared__ float red[16];
    float part = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) part += dWn[i] * W[i];
    const float s = blk_sum(part, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sn_bwd_apply_kernel(const float* __restrict__ dWn, const float* __restrict__ u, const float* __restrict__ v,
                                    const float* __restrict__ sigma_p, const float* __restrict__ partial, int nparts, float* __restrict__ dW,
                                    int R, int K) {
    float dot = 0.f;
    for (int i = 0; i < nparts; ++i) dot += partial[i];                 // same order in every thread
    const float sigma = sigma_p[0];
    const float coef = dot / (sigma * sigma);
    const long long n = (long long)R * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        dW[i] += dWn[i] / sigma - coef * u[r] * v[k];
    }
}

// gradient through weight = W / sigma, sigma = u^T W v (u, v constants):  dW = dWn / sigma - (sum(dWn * W) / sigma^2) * u v^T
// (sn_bwd_dot_kernel + sn_bwd_apply_kernel above)

// ---- BCE with logits against a constant target, mean reduction (core/losses.py:66-74), value + gradient ----
__global__ void bce_logits_kernel(const float* __restrict__ x, float* __restrict__ dx, float* __restrict__ loss, long long n, float target, float scale) {
    float local = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        // max(x,0) - x*t + log(1 + exp(-|x|))   (ATen binary_cross_entropy_with_logits)
        local += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
        const float sg = 1.f / (1.f + expf(-v));
        if (dx) dx[i] = scale * (sg - target);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, local * scale);
}

inline int wgrid(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

// ======================================= host side (C++ linkage, used by disc_engine.hip) ==================================
int vsr_launch_conv_wide(int dtype, const VsrWideConv& c, hipStream_t st) {
    if (!c.x || !c.y || (!c.wpack && !c.wpack2) || c.N < 1 || c.H < 1 || c.W < 1 || c.nsl < 1 || c.ncob < 1) return VSR_ERR_BADARG;
    if ((c.xC & 63) || (c.yC & 63) || c.nsl * 64 > c.xC || c.ncob * 64 > c.yC) return VSR_ERR_BADARG;
    if ((c.in_step != 1 && c.in_step != 2) || (c.out_step != 1 && c.out_step != 2) || (c.in_step == 2 && c.out_step == 2)) return VSR_ERR_BADARG;
    WideArgs a = {};
    a.x = c.x; a.x_nstride = pm_image_elems(c.Hx, c.Wx, c.xC); a.xC = c.xC; a.Hx = c.Hx; a.Wx = c.Wx;
    a.in_step = c.in_step; a.nsl = c.nsl; a.N = c.N; a.H = c.H; a.W = c.W; a.wpack = c.wpack; a.bias = c.bias;
    a.y = c.y; a.y_nstride = pm_image_elems(c.Hy, c.Wy, c.yC); a.yC = c.yC; a.Hy = c.Hy; a.Wy = c.Wy; a.out_step = c.out_step; a.ncob = c.ncob;
    // tap masks over (dy, dx) in {-1,0,1}^2, bit = (dy+1)*3 + (dx+1)
    auto m1 = [](int lo, int hi) { unsigned m = 0; for (int d = lo; d <= hi; ++d) m |= 1u << (d + 1); return m; };
    auto m2 = [&](unsigned my, unsigned mx) { unsigned m = 0; for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x) if ((my >> y & 1) && (mx >> x & 1)) m |= 1u << (y * 3 + x); return m; };
    if (c.in_step == 2) {          // view 0: dy in {0,1}; view 1: dy in {-1,0}
        for (int v = 0; v < 4; ++v) a.tapmask[v] = m2((v >> 1) ? m1(-1, 0) : m1(0, 1), (v & 1) ? m1(-1, 0) : m1(0, 1));
    } else if (c.out_step == 2) {  // phase 0: dy in {-1,0}; phase 1: dy in {0,1}
        for (int p = 0; p < 4; ++p) a.tapmask[p] = m2((p >> 1) ? m1(0, 1) : m1(-1, 0), (p & 1) ? m1(0, 1) : m1(-1, 0));
    } else {
        a.tapmask[0] = 0x1FF;
    }
    a.act = c.act; a.slope = c.slope; a.y_act = c.y_act; a.res = c.res; a.y_pre = c.y_pre; a.aux = c.aux;
    const int nph = c.out_step == 2 ? 4 : 1;
    if (c.wpack2) {                                  // bf16: the DMA-fed persistent kernel (64-channel output blocks: its K-split form)
        if (dtype != VSR_BF16 || c.y_pre) return VSR_ERR_BADARG;
        const bool ks = (c.ncob & 1) != 0;
        Wide2Args b = {};
        b.x = c.x; b.x_nstride = a.x_nstride; b.xCP = c.xC >> 3; b.Hx = c.Hx; b.Wx = c.Wx; b.in_step = c.in_step; b.nsl = c.nsl;
        b.nsrc = c.nsl * (c.in_step == 2 ? 4 : 1);
        b.N = c.N; b.H = c.H; b.W = c.W; b.wpack = c.wpack2;
        const int tps = (c.in_step == 2 || c.out_step == 2) ? 4 : 9;
        b.nsteps = b.nsrc * (ks ? (tps + 1) / 2 : tps);
        for (int i = 0; i < 4; ++i) b.tapmask[i] = a.tapmask[i];
        b.bias = c.bias; b.y = c.y; b.y_nstride = a.y_nstride; b.yCP = c.yC >> 3; b.Hy = c.Hy; b.Wy = c.Wy; b.out_step = c.out_step;
        b.ncp = ks ? c.ncob : c.ncob / 2; b.act = c.act; b.slope = c.slope; b.res = c.res; b.aux = c.aux; b.y_act = c.y_act;
        const int cols = b.ncp * nph;
        const int tiles = c.N * cdiv(c.H, TH) * cdiv(c.W, TW);
        int nwx = vsr_num_cus() / cols;
        if (nwx < 1) nwx = 1;
        if (nwx > tiles) nwx = tiles;
        {   // test knob: cap the workgroups per column so that small frames exercise the many-tiles-per-workgroup path
            const int cap = vsr_env().wide2_max_wg;
            if (cap > 0 && nwx > cap) nwx = cap;
        }
        static VsrDevOnce once2, once2k;
        if (ks) {
            { const int rc = vsr_set_max_dynamic_lds(once2k, reinterpret_cast<const void*>(conv_wide2_kernel<true>), W2_LDS); if (rc != VSR_OK) return rc; }
            hipLaunchKernelGGL(conv_wide2_kernel<true>, dim3(nwx, cols), dim3(W2_NT), W2_LDS, st, b);
        } else {
            { const int rc = vsr_set_max_dynamic_lds(once2, reinterpret_cast<const void*>(conv_wide2_kernel<false>), W2_LDS); if (rc != VSR_OK) return rc; }
            hipLaunchKernelGGL(conv_wide2_kernel<false>, dim3(nwx, cols), dim3(W2_NT), W2_LDS, st, b);
        }
        HIP_CHECK_RET(hipGetLastError());
        return VSR_OK;
    }
    if (!c.wpack) return VSR_ERR_BADARG;
    const long long gz = (long long)c.N * nph * c.ncob;
    if (gz > 65535) return VSR_ERR_UNSUPPORTED;
    dim3 grid(cdiv(c.W, TW), cdiv(c.H, TH), (unsigned)gz);
    if (dtype == VSR_BF16) {
        constexpr int LDS = NPIX * 8 * 16 + 2 * 64 * 8 * 16;
        hipLaunchKernelGGL(conv_wide_kernel<bf16_t>, grid, dim3(NTHREADS), LDS, st, a);
    } else if (dtype == VSR_F32) {
        constexpr int LDS = NPIX * 8 * 32 + 2 * 64 * 8 * 32;       // 87,040 + 32,768 = 119,808 B
        static VsrDevOnce once;
        { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(conv_wide_kernel<float>), LDS); if (rc != VSR_OK) return rc; }
        hipLaunchKernelGGL(conv_wide_kernel<float>, grid, dim3(NTHREADS), LDS, st, a);
    } else {
        return VSR_ERR_BADARG;
    }
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

long long vsr_wide_pack_elems(int cout, int cin, int mode) {
    const long long blocks = (mode == 1 || mode == 3) ? 4LL * (cout / 64) * (cin / 64) : (long long)(cout / 64) * (cin / 64);
    return blocks * 9 * 4096;
}

int vsr_launch_pack_wide(int dtype, const float* w, void* dst, int cout, int cin, int mode, hipStream_t st) {
    if (!w || !dst || (cout & 63) || (cin & 63) || mode < 0 || mode > 3) return VSR_ERR_BADARG;
    const long long total = vsr_wide_pack_elems(cout, cin, mode);
    const int grid = (int)((total + 255) / 256);
    if (dtype == VSR_BF16) hipLaunchKernelGGL(pack_wide_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)dst, cout, cin, mode, total);
    else if (dtype == VSR_F32) hipLaunchKernelGGL(pack_wide_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)dst, cout, cin, mode, total);
    else return VSR_ERR_BADARG;
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// rows = output channels of the launch (mode 0/1: cout, mode 2/3: cin); a multiple of 128 -> 128-row slabs, else the K-split image
long long vsr_wide2_pack_elems(int cout, int cin, int mode) {
    const bool s2 = mode == 1 || mode == 3;
    const int nsl = (mode == 0 || mode == 1 ? cin : cout) / 64, rows = (mode == 0 || mode == 1) ? cout : cin;
    const bool ks = (rows & 127) != 0;
    const int tps = s2 ? 4 : 9;
    const long long cols = (long long)(rows / (ks ? 64 : 128)) * (mode == 3 ? 4 : 1);
    return cols * nsl * (mode == 1 ? 4 : 1) * (ks ? (tps + 1) / 2 : tps) * 8192;
}

int vsr_launch_pack_wide2(const float* w, void* dst, int cout, int cin, int mode, hipStream_t st) {
    const int rows = (mode == 0 || mode == 1) ? cout : cin;
    if (!w || !dst || (cout & 63) || (cin & 63) || mode < 0 || mode > 3) return VSR_ERR_BADARG;
    const long long total = vsr_wide2_pack_elems(cout, cin, mode);
    hipLaunchKernelGGL(pack_wide2_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (bf16_t*)dst, cout, cin, mode,
                       (rows & 127) ? 1 : 0, total);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_wgrad_reduce_s2(const float* slab, int nwg, int slab_stride, float* gw, int cin_total, int co0, int ci0, int view, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_reduce_s2_kernel, dim3(cdiv(9 * 64 * 64, 256)), dim3(256), 0, st, slab, nwg, slab_stride, gw, cin_total, co0, ci0, view);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_up2_fwd(int dtype, const void* a, const void* b, void* out, int N, int H, int W, int C, hipStream_t st) {
    if (!a || !out || N < 1 || H < 1 || W < 1 || (C & 7)) return VSR_ERR_BADARG;
    if (N > 65535 || H > 65535) return VSR_ERR_UNSUPPORTED;
    const dim3 grid(cdiv(pm_ws(W) * 32 * (C / 8), 256), H, N);
    if (dtype == VSR_BF16) hipLaunchKernelGGL(up2_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, N, H, W, C);
    else if (dtype == VSR_F32) hipLaunchKernelGGL(up2_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, N, H, W, C);
    else return VSR_ERR_BADARG;
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_up2_bwd(int dtype, const void* dout, void* din, void* dmask, const void* m, float slope, int N, int H, int W, int C, hipStream_t st) {
    if (!dout || (!din && !dmask) || (dmask && !m) || N < 1 || H < 1 || W < 1 || (C & 7)) return VSR_ERR_BADARG;
    if (N > 65535 || H > 65535) return VSR_ERR_UNSUPPORTED;
    const dim3 grid(cdiv(pm_ws(W) * 32 * (C / 8), 256), H, N);
    if (dtype == VSR_BF16) hipLaunchKernelGGL(up2_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)din, (bf16_t*)dmask, (const bf16_t*)m, slope, N, H, W, C);
    else if (dtype == VSR_F32) hipLaunchKernelGGL(up2_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (float*)din, (float*)dmask, (const float*)m, slope, N, H, W, C);
    else return VSR_ERR_BADARG;
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_add_pm(int dtype, const void* a, const void* b, void* out, long long elems, hipStream_t st) {
    const long long n8 = elems / 8;
    if (dtype == VSR_BF16) hipLaunchKernelGGL(add_pm_kernel<bf16_t>, dim3(wgrid(n8)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8);
    else if (dtype == VSR_F32) hipLaunchKernelGGL(add_pm_kernel<float>, dim3(wgrid(n8)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n8);
    else return VSR_ERR_BADARG;
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_mask_pm(int dtype, const void* g, const void* m, void* out, float slope, long long elems, hipStream_t st) {
    const long long n8 = elems / 8;
    if (dtype == VSR_BF16) hipLaunchKernelGGL(mask_pm_kernel<bf16_t>, dim3(wgrid(n8)), dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)m, (bf16_t*)out, slope, n8);
    else if (dtype == VSR_F32) hipLaunchKernelGGL(mask_pm_kernel<float>, dim3(wgrid(n8)), dim3(256), 0, st, (const float*)g, (const float*)m, (float*)out, slope, n8);
    else return VSR_ERR_BADARG;
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

extern "C" {

int vsr_spectral_norm(const float* w_orig, float* u, float* v, float* w_out, float* sigma, int rows, int cols, int training, void* stream) {
    if (!w_orig || !u || !v || !w_out || !sigma || rows < 1 || rows > 512 || cols < 1) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (!training) {             // eval: u, v untouched; one workgroup is plenty off the training path
        hipLaunchKernelGGL(spectral_norm_kernel, dim3(1), dim3(1024), 0, st, w_orig, u, v, w_out, sigma, rows, cols, 0, 1e-12f);
        HIP_CHECK_RET(hipGetLastError());
        return VSR_OK;
    }
    hipLaunchKernelGGL(sn_wtu_kernel, dim3(cdiv(cols, 64)), dim3(64), 0, st, w_orig, (const float*)u, v, rows, cols);
    hipLaunchKernelGGL(sn_norm_kernel, dim3(1), dim3(1024), 0, st, v, cols, 1e-12f, (float*)nullptr);
    hipLaunchKernelGGL(sn_wv_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, w_orig, (const float*)v, u, rows, cols);
    hipLaunchKernelGGL(sn_norm_kernel, dim3(1), dim3(1024), 0, st, u, rows, 1e-12f, sigma);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(wgrid((long long)rows * cols)), dim3(256), 0, st, w_orig, (const float*)sigma, w_out, (long long)rows * cols);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_spectral_norm_backward(const float* dw, const float* w_orig, const float* u, const float* v, const float* sigma, float* dw_orig,
                               int rows, int cols, float* scratch, void* stream) {
    if (!dw || !w_orig || !u || !v || !sigma || !dw_orig || !scratch || rows < 1 || cols < 1) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)rows * cols;
    int parts = (int)((n + 256 * 64 - 1) / (256 * 64));
    if (parts > VSR_SN_SCRATCH_FLOATS) parts = VSR_SN_SCRATCH_FLOATS;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(parts), dim3(256), 0, st, dw, w_orig, scratch, n);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(wgrid(n)), dim3(256), 0, st, dw, u, v, sigma, (const float*)scratch, parts, dw_orig, rows, cols);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_bce_with_logits(const float* x, float target, float* dx, float* loss, long long numel, void* stream) {
    if (!x || !loss || numel < 1) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK_RET(hipMemsetAsync(loss, 0, sizeof(float), st));
    hipLaunchKernelGGL(bce_logits_kernel, dim3(wgrid(numel)), dim3(256), 0, st, x, dx, loss, numel, target, 1.0f / (float)numel);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // extern "C"
