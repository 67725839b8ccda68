// conv_last.2 (basicvsr.py:21: conv3x3 64 -> 3 at 4h x 4w) has 3 real output channels: its backward is a
// streaming problem, not a GEMM.  Dedicated bf16 kernels for the HR tail of the backward pass (the generic tiled
// kernel pads 3 channels to 16/32 and synchronises per tap: 0.9 ms per frame at 2160x3840 against an HBM floor of ~0.45).
#include "kernels.h"

namespace {

struct __attribute__((aligned(8))) bf4_t { bf16_t v[4]; };

constexpr int LT_W = 32, LT_H = 8, LT_NT = 256;
constexpr int LT_RS = LT_W + 2, LT_PL = (LT_H + 2) * LT_RS;     // haloed planar tile: 3 planes of 10 x 34 floats

// Data gradient: dX[p][ci] = mask(aux[p][ci]) * sum_{tap,c} dSR[c][p + tap - 1] * w[c][ci][flip(tap)].
// K = 9 taps x 3 channels = 27 <= 32: ONE v_mfma_f32_16x16x32_bf16 per (16 ci) x (16 pixels) block, with the B
// operand gathered from the fp32 dSR tile in LDS (k = 3 tap + c) and the A operand (flipped weights) built once per
// workgroup from the fp32 OIHW tensor.  Output / mask addressing is the blocked pixel-major layout of common.h.
// grid: persistent over 8x32-pixel tiles; block 256 = 4 waves, wave w covers tile rows 2w, 2w+1.
__global__ __launch_bounds__(LT_NT) void last2_dgrad_kernel(const float* __restrict__ dsr, long long dsr_nstride,
                                                            const float* __restrict__ w, const bf16_t* __restrict__ aux,
                                                            bf16_t* __restrict__ dst, int N, int H, int W, int mask_mode) {
    __shared__ float tile[2][3 * LT_PL];
    const int tid = threadIdx.x, lane = tid & 63, w4 = tid >> 6;
    const int i15 = lane & 15, q = lane >> 4;
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
            const int ci = mb * 16 + i15;
            fa[mb][j] = (bf16_t)(k < 27 ? w[((long long)c * 64 + ci) * 9 + (2 - ky) * 3 + (2 - kx)] : 0.f);
        }
    }
    const float neg = mask_mode == MASK_LEAKY ? 0.1f : 0.f;

    auto stage = [&](int t, int buf) {                       // haloed dSR tile -> LDS (zeros outside the image)
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        const float* base = dsr + (long long)n * dsr_nstride;
        for (int e = tid; e < 3 * LT_PL; e += LT_NT) {
            const int c = e / LT_PL, rem = e - c * LT_PL;
            const int yy = rem / LT_RS, xx = rem - yy * LT_RS;
            const int vy = ty0 + yy - 1, vx = tx0 + xx - 1;
            tile[buf][e] = (vy >= 0 && vy < H && vx >= 0 && vx < W) ? base[c * plane + (long long)vy * W + vx] : 0.f;
        }
    };

    int t = blockIdx.x, buf = 0;
    if (t < total) stage(t, 0);
    __syncthreads();
    for (; t < total; t += gridDim.x) {
        const int n = t / (ntx * nty), r = t - n * (ntx * nty);
        const int ty0 = (r / ntx) * LT_H, tx0 = (r % ntx) * LT_W;
        const long long obase = (long long)n * pm_image_elems(H, W, 64) + pm_off(ty0, tx0, 0, W, 64);
        // mask operands first: their latency hides behind the staging of the next tile and the MFMAs
        bf4_t mm[4][4];
        bool ok[4];
        int loff[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int row = 2 * w4 + (nb >> 1), px = (nb & 1) * 16 + i15;
            ok[nb] = ty0 + row < H && tx0 + px < W;
            loff[nb] = (row * pm_ws(W) * 8 + (q >> 1)) * 256 + px * 8 + 4 * (q & 1);
            if (aux && ok[nb]) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) mm[mb][nb] = *reinterpret_cast<const bf4_t*>(aux + obase + loff[nb] + mb * 512);
            }
        }
        const int tn = t + gridDim.x;
        if (tn < total) stage(tn, buf ^ 1);                  // the other buffer: nobody reads it during this tile
        // B[k][n = pixel i] for this wave's 4 pixel blocks, then 16 MFMAs
        f32x4_t acc[4][4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int row = 2 * w4 + (nb >> 1), px = (nb & 1) * 16 + i15;
            const float* tp = tile[buf] + row * LT_RS + px;
            bf16x8_t fb;
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[j] = (bf16_t)(boff[j] >= 0 ? tp[boff[j]] : 0.f);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mb], fb, z, 0, 0, 0);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (!ok[nb]) continue;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                bf4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[mb][nb][j];
                    if (aux) v *= ((float)mm[mb][nb].v[j] > 0.f ? 1.f : neg);
                    o.v[j] = (bf16_t)v;
                }
                *reinterpret_cast<bf4_t*>(dst + obase + loff[nb] + mb * 512) = o;
            }
        }
        __syncthreads();                                     // the next tile is staged; this one is consumed
        buf ^= 1;
    }
}

}  // namespace

// dX (pixel-major bf16, 64 channels) = mask(aux) * dgrad of a 64 -> 3 3x3 conv, from the planar fp32 cotangent dsr
// (N images `dsr_nstride` floats apart) and the conv's fp32 OIHW weight (3,64,3,3).
int vsr_launch_last2_dgrad(const float* dsr, long long dsr_nstride, const float* w, const void* aux, void* dst, int N, int H, int W,
                           int mask_mode, hipStream_t st) {
    if (!dsr || !w || !dst || N < 1 || H < 1 || W < 1) return VSR_ERR_BADARG;
    const int tiles = N * cdiv(W, LT_W) * cdiv(H, LT_H);
    const int grid = tiles < 256 * 8 ? tiles : 256 * 8;
    hipLaunchKernelGGL(last2_dgrad_kernel, dim3(grid), dim3(LT_NT), 0, st, dsr, dsr_nstride, w, (const bf16_t*)aux, (bf16_t*)dst, N, H, W,
                       mask_mode);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}
