// The hot kernel of the path: 3x3, 64 -> 64 channels, bf16 in / fp32 accumulate, pixel-major.
//   core/modules/conv.py:85-86 (ResidualConv conv1/conv2), basicvsr.py:20 (conv_last.0),
//   upsampling.py:7 (as 4 pixel-shuffle phases), and all their data gradients: > 70 % of the FLOPs.
//
// Persistent design for one MI355X CU (160 KiB LDS, 4 SIMDs):
//   * ONE 512-thread workgroup per CU (8 waves, 2 per SIMD), grid = #CUs, tiles strided over it.
//   * the whole packed weight tensor [9 taps][64 cout][64 cin] bf16 (72 KiB) is loaded into LDS once
//     per workgroup and stays there; the haloed 18x34-pixel input tile (76.5 KiB) sits beside it.
//   * no barrier inside the 144-MFMA K loop: A (weights) and B (pixels) fragments are plain
//     ds_read_b128 at register base + immediate offset.  Both LDS images are XOR-swizzled on the
//     16-byte chunk index (weights by cout row, pixels by x only, so a tap shift in y is a pure
//     immediate offset) => conflict-free reads.
//   * the NEXT tile is fetched from HBM/L2 into registers while the current one is multiplied
//     (issue early / write late), and written to LDS between the tile's two barriers.
//   * fused epilogue in accumulator layout: bias, ReLU / LeakyReLU(0.1), residual add,
//     activation-gradient mask, pixel-shuffle placement.
// Per 16x32-pixel tile a wave issues 144 v_mfma_f32_32x32x16_bf16 against 144 ds_read_b128: the LDS
// runs at half its read rate, the matrix pipe is the limiter.
#include "common.h"

namespace {

constexpr int PTW = 32, PTH = 16, PNT = 512;
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;       // 34 x 18 = 612 haloed pixels
constexpr int W_BYTES = 9 * 64 * 64 * 2;                                  // 73,728
constexpr int IN_BYTES = PNPIX * 128;                                     // 78,336
constexpr int BIAS_BYTES = 256;                                           // 64 fp32
constexpr int P_LDS = W_BYTES + IN_BYTES + BIAS_BYTES;                    // 152,320 <= 163,840
constexpr int IN_CHUNKS = PNPIX * 8;                                      // 4,896 16-byte chunks
constexpr int PRE = (IN_CHUNKS + PNT - 1) / PNT;                          // 10 chunks per thread

template <int ACT> __device__ __forceinline__ float p_act(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}

struct __attribute__((aligned(8))) bf4 { bf16_t v[4]; };

__device__ __forceinline__ void tile_coords(const ConvArgs& a, int tile, int ntx, int nty, int& n, int& ty0, int& tx0) {
    const int per = ntx * nty;
    n = tile / per;
    const int r = tile - n * per;
    const int ty = r / ntx;
    ty0 = ty * PTH;
    tx0 = (r - ty * ntx) * PTW;
}

// Epilogue variants are compile-time: a runtime-selected epilogue serialises 16 load->use->store
// chains per tile (measured: 14 us of a 71 us launch).
template <int ACT, bool HAS_RES, int MASK>
__global__ __launch_bounds__(PNT, 2) void conv3x3_c64_persist_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_in = smem + W_BYTES;
    float* lds_bias = reinterpret_cast<float*>(smem + W_BYTES + IN_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int z = blockIdx.y;
    const int ntx = cdiv(a.W, PTW), nty = cdiv(a.H, PTH);
    const int total = a.N * ntx * nty;
    if (tid < 64) lds_bias[tid] = a.bias ? a.bias[(long long)z * a.bias_zstride + tid] : 0.f;

    // ---- weights of this z: global [tap][cout][cin] -> LDS, chunk c of row r at (r*8 + (c ^ ((r>>1)&7))) ----
    {
        const uint4* wg = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.wpack) + (long long)z * a.w_zstride);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int idx = tid + i * PNT;                 // 4608 chunks
            const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;
            *reinterpret_cast<uint4*>(lds_w + tap * 8192 + (r * 8 + (c ^ ((r >> 1) & 7))) * 16) = wg[idx];
        }
    }

    // per-thread staging slots of the input tile: chunk idx -> pixel p = idx>>3 (ty = p/34, tx = p%34), chunk c = idx&7
    const bf16_t* src = reinterpret_cast<const bf16_t*>(a.src[0]);
    uint4 pre[PRE];
    auto fetch = [&](int tile) {
        int n, ty0, tx0;
        tile_coords(a, tile, ntx, nty, n, ty0, tx0);
        const bf16_t* base = src + (long long)n * a.src_nstride[0];
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = tid + i * PNT;
            const int p = idx >> 3, c = idx & 7;
            const int ty = p / PTWH, tx = p - ty * PTWH;
            const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < IN_CHUNKS && vy >= 0 && vy < a.H && vx >= 0 && vx < a.W)
                v = *reinterpret_cast<const uint4*>(base + ((long long)vy * a.W + vx) * 64 + c * 8);
            pre[i] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = tid + i * PNT;
            const int p = idx >> 3, c = idx & 7;
            const int tx = p % PTWH;
            if (idx < IN_CHUNKS) *reinterpret_cast<uint4*>(lds_in + (p * 8 + (c ^ ((tx >> 1) & 7))) * 16) = pre[i];
        }
    };

    int tile = blockIdx.x;
    if (tile < total) { fetch(tile); stash(); }
    __syncthreads();

    // fragment base addresses (bytes): A per (cb, ks), B per (kx, ks); taps / rows are immediates
    int a_off[2][4], b_off[3][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = cb * 32 + l31;
            a_off[cb][ks] = (r * 8 + ((2 * ks + h) ^ ((r >> 1) & 7))) * 16;
        }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int tx = l31 + kx;
            b_off[kx][ks] = ((wave * 2 * PTWH + tx) * 8 + ((2 * ks + h) ^ ((tx >> 1) & 7))) * 16;
        }

    for (; tile < total; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < total) fetch(next);                      // in flight during the K loop

        f32x16_t acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][rw][i] = 0.f;

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8_t af[2], bf[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) af[cb] = *reinterpret_cast<const bf16x8_t*>(lds_w + tap * 8192 + a_off[cb][ks]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) bf[rw] = *reinterpret_cast<const bf16x8_t*>(lds_in + (rw + ky) * (PTWH * 128) + b_off[kx][ks]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw)
                        acc[cb][rw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb], bf[rw], acc[cb][rw], 0, 0, 0);
            }
        }

        // Swap the input tile FIRST (the prefetch registers die here), then run the epilogue: no barrier
        // follows it, so waves drift apart and one wave's epilogue overlaps its SIMD partner's next K loop.
        __syncthreads();                 // every wave has finished reading this tile from LDS
        if (next < total) stash();
        __syncthreads();

        // ---- epilogue, in 4 batches of (row, cout-block): the batch's residual / mask loads are issued
        // together (one latency per batch), then math + 8-byte stores ----
        int n, ty0, tx0;
        tile_coords(a, tile, ntx, nty, n, ty0, tx0);
        const int vx = tx0 + l31;
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            const int vy = ty0 + wave * 2 + rw;
            if (vx < a.W && vy < a.H) {
                const int oy = vy * a.out_step + a.out_oy[z], ox = vx * a.out_step + a.out_ox[z];
                const long long pixo = (long long)n * a.dst_nstride + ((long long)oy * a.Wd + ox) * 64 + 4 * h;
                bf16_t* dst = reinterpret_cast<bf16_t*>(a.dst[z]) + pixo;
                const bf16_t* resp = HAS_RES ? reinterpret_cast<const bf16_t*>(a.res[z]) + pixo : nullptr;
                const bf16_t* auxp = MASK != MASK_NONE ? reinterpret_cast<const bf16_t*>(a.aux[z]) + pixo : nullptr;
                constexpr float neg = MASK == MASK_LEAKY ? 0.1f : 0.f;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    bf4 rres[4], raux[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (HAS_RES) rres[g] = *reinterpret_cast<const bf4*>(resp + cb * 32 + 8 * g);
                        if (MASK != MASK_NONE) raux[g] = *reinterpret_cast<const bf4*>(auxp + cb * 32 + 8 * g);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b = *reinterpret_cast<const float4*>(lds_bias + cb * 32 + 8 * g + 4 * h);
                        float v[4];
                        v[0] = acc[cb][rw][4 * g + 0] + b.x; v[1] = acc[cb][rw][4 * g + 1] + b.y;
                        v[2] = acc[cb][rw][4 * g + 2] + b.z; v[3] = acc[cb][rw][4 * g + 3] + b.w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = p_act<ACT>(v[j]);
                        if (HAS_RES) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += (float)rres[g].v[j];
                        }
                        if (MASK != MASK_NONE) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= ((float)raux[g].v[j] > 0.f ? 1.f : neg);
                        }
                        bf4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o.v[j] = (bf16_t)v[j];
                        *reinterpret_cast<bf4*>(dst + cb * 32 + 8 * g) = o;
                    }
                }
            }
        }
    }
}

}  // namespace

// Eligibility is decided by the dispatcher in conv_mfma.hip: bf16, 3x3, one pixel-major 64-channel
// source at unit step, 64 output channels, pixel-major destination.
template <int ACT, bool HAS_RES, int MASK>
static int launch_persist(const ConvArgs& a, int num_cus, hipStream_t st) {
    auto kern = conv3x3_c64_persist_kernel<ACT, HAS_RES, MASK>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS));
        attr_set = true;
    }
    const int tiles = a.N * cdiv(a.W, PTW) * cdiv(a.H, PTH);
    int gx = num_cus / a.nz;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    hipLaunchKernelGGL(kern, dim3(gx, a.nz), dim3(PNT), P_LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// Returns VSR_ERR_UNSUPPORTED for an epilogue combination that has no instantiation (the caller then
// uses the generic kernel).
int vsr_launch_conv3x3_c64_persist(const ConvArgs& a, int num_cus, hipStream_t st) {
    bool res = false, aux = false;
    for (int z = 0; z < a.nz; ++z) { res = res || a.res[z]; aux = aux || a.aux[z]; }
    for (int z = 0; z < a.nz; ++z) if ((res && !a.res[z]) || (aux && !a.aux[z])) return VSR_ERR_UNSUPPORTED;
    const int mask = aux ? a.mask_mode : MASK_NONE;
#define PERSIST_CASE(ACT, RES, MASK) if (a.act == ACT && res == RES && mask == MASK) return launch_persist<ACT, RES, MASK>(a, num_cus, st);
    PERSIST_CASE(ACT_RELU, false, MASK_NONE)     // conv1 of a ResidualConv
    PERSIST_CASE(ACT_NONE, true, MASK_NONE)      // conv2 + skip ; dgrad(conv1) + dX
    PERSIST_CASE(ACT_LEAKY, false, MASK_NONE)    // conv_last.0
    PERSIST_CASE(ACT_NONE, false, MASK_NONE)     // upsample phases, plain dgrads
    PERSIST_CASE(ACT_NONE, false, MASK_RELU)     // dgrad(conv2) * ReLU'
    PERSIST_CASE(ACT_NONE, true, MASK_LEAKY)     // (dgrad(conv1 of block 0) + dX) * LeakyReLU' of the stem
#undef PERSIST_CASE
    return VSR_ERR_UNSUPPORTED;
}
