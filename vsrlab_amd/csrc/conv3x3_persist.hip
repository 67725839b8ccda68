// The hot kernel of the path: 3x3, 64 -> 64 channels, bf16 in / fp32 accumulate, pixel-major.
//   core/modules/conv.py:85-86 (ResidualConv conv1/conv2), basicvsr.py:20 (conv_last.0),
//   upsampling.py:7 (as 4 pixel-shuffle phases), and all their data gradients: > 70 % of the FLOPs.
//
// At 540p one launch moves 133-200 MB for 38 GFLOP: 21-32 us at the 6.3 TB/s a CU-side stream reaches,
// 15 us on the matrix cores -- the kernel is HBM-bound, and a CU needs ~75 KB in flight to cover ~3 us of
// loaded HBM latency.  Design for one MI355X CU (160 KiB LDS, 4 SIMDs):
//   * ONE 256-thread workgroup per CU (one wave per SIMD, up to 512 VGPRs each), grid = #CUs, persistent
//     over 8x32-pixel tiles.  The packed weights [9 taps][64 cout][64 cin] bf16 (72 KiB) are loaded into LDS
//     once and stay.
//   * the haloed 10x34-pixel input tile is DOUBLE-BUFFERED in LDS (2 x 42.5 KiB) and filled by LDS-DMA
//     (global_load_lds_dwordx4) one whole tile ahead; the residual / mask operands of the epilogue are
//     requested into registers before the K loop.  Nothing the tile needs is waited for at first use.
//   * K loop: 144 v_mfma_f32_32x32x16_bf16 per wave and tile, no barrier inside; A (weights) and B (pixels)
//     fragments are ds_read_b128 at register base + immediate, requested two k-steps ahead of their MFMAs
//     (explicit 3-deep ring, order pinned with sched_group_barrier: a lone wave per SIMD has nobody to hide
//     an LDS round trip behind).  Both LDS images are XOR-swizzled on the 16-byte chunk (weights by cout
//     row, pixels by x only => a ky shift is an immediate); the pixel swizzle is applied on the DMA's
//     per-lane SOURCE address.  SQ_LDS_BANK_CONFLICT = 0 measured.
//   * epilogue: the accumulator layout (lane = pixel, registers = channels) would store 8-byte pieces into
//     32 different 128-byte lines per instruction (measured: as expensive as the K loop).  Each wave
//     transposes its rows through a private 8 KiB fp32 slot of the tile buffer it has just finished with and
//     writes whole 128-byte pixel lines, 16 bytes per lane.  Fused: bias, ReLU / LeakyReLU(0.1), residual
//     add, activation-gradient mask, pixel-shuffle placement; one rounding to bf16 at the very end.
#include "common.h"

namespace {

constexpr int PTW = 32, PTH = 8, PNT = 256;
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;       // 34 x 10 = 340 haloed pixels
constexpr int W_BYTES = 9 * 64 * 64 * 2;                                  // 73,728
constexpr int IN_BYTES = PNPIX * 128;                                     // 43,520 per buffer (>= 4 waves x 8 KiB slots)
constexpr int BIAS_BYTES = 256;                                           // 64 fp32
constexpr int P_LDS = W_BYTES + 2 * IN_BYTES + BIAS_BYTES;                // 161,024 <= 163,840
constexpr int IN_CHUNKS = PNPIX * 8;                                      // 2,720 16-byte chunks
constexpr int NPIECE_T = (IN_CHUNKS + 63) / 64;                           // 43 DMA pieces of 1 KiB (last half full)
constexpr int NPIECE_W = (NPIECE_T + 3) / 4;                              // 11 per wave

__device__ uint4 g_conv_zero_chunk[2];

template <int ACT> __device__ __forceinline__ float p_act(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}

// Diagnostic build only (make STAMPS=1): per-wave cycle sums of the phases of a tile.  The stamp values
// go to a buffer of their own and never into an output.
#ifdef VSR_STAMPS
__device__ unsigned long long g_stamps[256 * 8 * 8];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp()
#define STAMP_ADD(slot, a, b) st_sum[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

__device__ __forceinline__ void tile_coords(int tile, int ntx, int nty, int& n, int& ty0, int& tx0) {
    const int per = ntx * nty;
    n = tile / per;
    const int r = tile - n * per;
    const int ty = r / ntx;
    ty0 = ty * PTH;
    tx0 = (r - ty * ntx) * PTW;
}

__device__ __forceinline__ void unpack_bf8(const uint4& u, float* f) {
    union { uint4 q; bf16_t h[8]; } t; t.q = u;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)t.h[j];
}

#define GLDS16(src, dst)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// Epilogue variants are compile-time: a runtime-selected epilogue serialises 16 load->use->store
// chains per tile (measured: 14 us of a 71 us launch).
template <int ACT, bool HAS_RES, int MASK>
__global__ __launch_bounds__(PNT, 1) void conv3x3_c64_persist_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    char* lds_w = smem;
    char* lds_t = smem + W_BYTES;                                         // two tile buffers
    float* lds_bias = reinterpret_cast<float*>(smem + W_BYTES + 2 * IN_BYTES);

    const int z = blockIdx.y;
    const int ntx = cdiv(a.W, PTW), nty = cdiv(a.H, PTH);
    const int total = a.N * ntx * nty;
    if (tid < 64) lds_bias[tid] = a.bias ? a.bias[(long long)z * a.bias_zstride + tid] : 0.f;

    // ---- weights of this z: global [tap][cout][cin] -> LDS, chunk c of row r at (r*8 + (c ^ ((r>>1)&7))) ----
    {
        const uint4* wg = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.wpack) + (long long)z * a.w_zstride);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int idx = tid + i * PNT;                 // 4608 chunks
            const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;
            *reinterpret_cast<uint4*>(lds_w + tap * 8192 + (r * 8 + (c ^ ((r >> 1) & 7))) * 16) = wg[idx];
        }
    }

    // ---- DMA pieces of this wave: piece = wave + 4 i.  LDS slot (pixel p, position q) of a piece receives
    // global chunk q ^ ((tx>>1)&7) of pixel p; rel[i] = this lane's source byte offset from the tile origin ----
    const char* src = reinterpret_cast<const char*>(a.src[0]);
    const char* zsrc = reinterpret_cast<const char*>(g_conv_zero_chunk);
    int rel[NPIECE_W];
#pragma unroll
    for (int i = 0; i < NPIECE_W; ++i) {
        const int idx = (wave + 4 * i) * 64 + lane;
        const int p = idx >> 3;
        const int ty = p / PTWH, tx = p - ty * PTWH;
        rel[i] = (((ty - 1) * a.W + (tx - 1)) * 64 + ((idx & 7) ^ ((tx >> 1) & 7)) * 8) * 2;
    }
    auto issue = [&](int tile, int buf) {
        int n, ty0, tx0;
        tile_coords(tile, ntx, nty, n, ty0, tx0);
        const char* org = src + ((long long)n * a.src_nstride[0] + ((long long)ty0 * a.W + tx0) * 64) * 2;
        char* dstb = lds_t + buf * IN_BYTES;
        if (ty0 >= 1 && ty0 + PTH < a.H && tx0 >= 1 && tx0 + PTW < a.W) {          // interior tile (wave-uniform)
#pragma unroll
            for (int i = 0; i < NPIECE_W; ++i) {
                const int piece = wave + 4 * i;
                if (piece < NPIECE_T && piece * 64 + lane < IN_CHUNKS) GLDS16(org + rel[i], dstb + piece * 1024);
            }
        } else {                                                                    // border: bounds per lane, zero source
#pragma unroll
            for (int i = 0; i < NPIECE_W; ++i) {
                const int piece = wave + 4 * i;
                const int idx = piece * 64 + lane;
                const int p = idx >> 3;
                const int ty = p / PTWH, tx = p - ty * PTWH;
                const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                const char* s = (vy >= 0 && vy < a.H && vx >= 0 && vx < a.W) ? org + rel[i] : zsrc;
                if (piece < NPIECE_T && idx < IN_CHUNKS) GLDS16(s, dstb + piece * 1024);
            }
        }
    };

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
    const int rpx = lane >> 3, rch = lane & 7;            // epilogue role: pixel rpx + 8*it, channels 8*rch..8*rch+7
    int loff[2][4];                                       // lane-constant part of the destination element offset
#pragma unroll
    for (int rw = 0; rw < 2; ++rw)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            loff[rw][it] = (((wave * 2 + rw) * a.out_step) * a.Wd + (rpx + 8 * it) * a.out_step) * 64 + rch * 8;

#ifdef VSR_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int cur = 0;
    int tile = blockIdx.x;
    if (tile < total) issue(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // weights, bias and the first tile are in LDS

    float4 breg[2][4];                                    // this lane's 32 bias values (accumulator layout), for the whole launch
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) breg[cb][g] = *reinterpret_cast<const float4*>(lds_bias + cb * 32 + 8 * g + 4 * h);

    for (; tile < total; tile += gridDim.x) {
        STAMP(t0);
        const int next = tile + gridDim.x;
        if (next < total) issue(next, cur ^ 1);            // a whole tile ahead of the MFMAs
        const char* lds_in = lds_t + cur * IN_BYTES;

        // epilogue operands, requested now, used after the K loop
        int n, ty0, tx0;
        tile_coords(tile, ntx, nty, n, ty0, tx0);
        // destination offset = wave-uniform tile part (64-bit, scalar unit) + lane-constant part loff (32-bit)
        const long long tbase = (long long)n * a.dst_nstride + ((long long)(ty0 * a.out_step + a.out_oy[z]) * a.Wd + (tx0 * a.out_step + a.out_ox[z])) * 64;
        bool ok[2][4];
        uint4 rr[2][4], mm[2][4];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                ok[rw][it] = (ty0 + wave * 2 + rw < a.H) && (tx0 + rpx + 8 * it < a.W);
                if (HAS_RES && ok[rw][it]) rr[rw][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.res[z]) + tbase + loff[rw][it]);
                if (MASK != MASK_NONE && ok[rw][it]) mm[rw][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.aux[z]) + tbase + loff[rw][it]);
            }
        }
        STAMP(t1);

        f32x16_t acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][rw][i] = 0.f;

        // ---- K loop: 36 steps s = (tap, ks) of 4 MFMAs.  The 4 fragment reads of step s+2 are issued before
        // the MFMAs of step s, by hand: the reads are inline asm with a counted s_waitcnt (lgkmcnt(8) = "all
        // but the 8 youngest LDS reads have returned" = step s is in registers), because hipcc's scheduler
        // sinks builtin LDS reads back in front of their consumers (measured: ds_read x4, lgkmcnt(0), mfma x4).
        // A lone wave per SIMD has nobody else to hide an LDS round trip behind.
        bf16x8_t fa[3][2], fb[3][2];
        unsigned bb[3][4];                                   // B base addresses of this tile's buffer
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bb[kx][ks] = (unsigned)(W_BYTES + cur * IN_BYTES + b_off[kx][ks]);
#define DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define CV_LOAD(s, slot)                                                                                               \
        {                                                                                                              \
            constexpr int tap_ = (s) / 4, ks_ = (s) % 4, ky_ = tap_ / 3, kx_ = tap_ % 3;                               \
            if (tap_ < 8) {                                                                                            \
                DSR(fa[slot][0], (unsigned)a_off[0][ks_], tap_ < 8 ? tap_ * 8192 : 0);                                 \
                DSR(fa[slot][1], (unsigned)a_off[1][ks_], tap_ < 8 ? tap_ * 8192 : 0);                                 \
            } else {                                                                                                   \
                DSR(fa[slot][0], (unsigned)a_off[0][ks_] + 8192u, 57344);                                              \
                DSR(fa[slot][1], (unsigned)a_off[1][ks_] + 8192u, 57344);                                              \
            }                                                                                                          \
            DSR(fb[slot][0], bb[kx_][ks_], ky_ * (PTWH * 128));                                                        \
            DSR(fb[slot][1], bb[kx_][ks_], (1 + ky_) * (PTWH * 128));                                                  \
        }
#define CV_STEP(s)                                                                                                     \
        {                                                                                                              \
            if ((s) + 2 < 36) CV_LOAD((s) + 2, ((s) + 2) % 3)                                                          \
            if ((s) + 2 < 36) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                       \
            else if ((s) + 1 < 36) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                  \
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(s) % 3][0], fb[(s) % 3][0], acc[0][0], 0, 0, 0);   \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(s) % 3][0], fb[(s) % 3][1], acc[0][1], 0, 0, 0);   \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(s) % 3][1], fb[(s) % 3][0], acc[1][0], 0, 0, 0);   \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(s) % 3][1], fb[(s) % 3][1], acc[1][1], 0, 0, 0);   \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of ours is outstanding: the counts below are exact
        __builtin_amdgcn_sched_barrier(0);
        CV_LOAD(0, 0)
        CV_LOAD(1, 1)
        CV_STEP(0) CV_STEP(1) CV_STEP(2) CV_STEP(3) CV_STEP(4) CV_STEP(5) CV_STEP(6) CV_STEP(7) CV_STEP(8)
        CV_STEP(9) CV_STEP(10) CV_STEP(11) CV_STEP(12) CV_STEP(13) CV_STEP(14) CV_STEP(15) CV_STEP(16) CV_STEP(17)
        CV_STEP(18) CV_STEP(19) CV_STEP(20) CV_STEP(21) CV_STEP(22) CV_STEP(23) CV_STEP(24) CV_STEP(25) CV_STEP(26)
        CV_STEP(27) CV_STEP(28) CV_STEP(29) CV_STEP(30) CV_STEP(31) CV_STEP(32) CV_STEP(33) CV_STEP(34) CV_STEP(35)
#undef CV_STEP
#undef CV_LOAD
#undef DSR
        STAMP(t2);
        __syncthreads();                 // every wave has finished reading this tile: its buffer is free for the slots
        STAMP(t3);

        // ---- epilogue ----
        char* slot = lds_t + cur * IN_BYTES + wave * 8192;   // fp32 [32 pixels][64 ch], chunk q of pixel px at q ^ (px & 15)
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * h;
                    const float4 b = breg[cb][g];
                    float4 v;
                    v.x = p_act<ACT>(acc[cb][rw][4 * g + 0] + b.x); v.y = p_act<ACT>(acc[cb][rw][4 * g + 1] + b.y);
                    v.z = p_act<ACT>(acc[cb][rw][4 * g + 2] + b.z); v.w = p_act<ACT>(acc[cb][rw][4 * g + 3] + b.w);
                    *reinterpret_cast<float4*>(slot + l31 * 256 + (((co >> 2) ^ (l31 & 15)) << 4)) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float4 lo[4], hi[4];                          // all 8 reads in flight together: one LDS round trip per row
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = rpx + 8 * it;
                lo[it] = *reinterpret_cast<const float4*>(slot + px * 256 + (((2 * rch) ^ (px & 15)) << 4));
                hi[it] = *reinterpret_cast<const float4*>(slot + px * 256 + (((2 * rch + 1) ^ (px & 15)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (ok[rw][it]) {
                    float v[8] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w, hi[it].x, hi[it].y, hi[it].z, hi[it].w};
                    if (HAS_RES) {
                        float r[8];
                        unpack_bf8(rr[rw][it], r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += r[j];
                    }
                    if (MASK != MASK_NONE) {
                        float m[8];
                        unpack_bf8(mm[rw][it], m);
                        constexpr float neg = MASK == MASK_LEAKY ? 0.1f : 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= (m[j] > 0.f ? 1.f : neg);
                    }
                    union { uint4 q; bf16_t hh[8]; } pk;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pk.hh[j] = (bf16_t)v[j];
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.dst[z]) + tbase + loff[rw][it]) = pk.q;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        STAMP(t4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile has landed (this wave's pieces)
        __syncthreads();                                      // ... everybody's; the slots are no longer read
        cur ^= 1;
        STAMP(t5);
        STAMP_ADD(1, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4); STAMP_ADD(5, t4, t5);
    }
#ifdef VSR_STAMPS
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 256)
        for (int k = 0; k < 8; ++k) g_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_sum[k];
#endif
}

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

}  // namespace

#ifdef VSR_STAMPS
extern "C" int vsr_debug_read_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -3;
}
#endif

// Eligibility is decided by the dispatcher in conv_mfma.hip (bf16, 3x3, one pixel-major 64-channel
// source at unit step, 64 output channels, pixel-major destination).  Returns VSR_ERR_UNSUPPORTED for
// an epilogue combination that has no instantiation; the caller then uses the generic kernel.
int vsr_launch_conv3x3_c64_persist(const ConvArgs& a, int num_cus, hipStream_t st) {
    bool res = false, aux = false;
    for (int z = 0; z < a.nz; ++z) { res = res || a.res[z]; aux = aux || a.aux[z]; }
    for (int z = 0; z < a.nz; ++z) if ((res && !a.res[z]) || (aux && !a.aux[z])) return VSR_ERR_UNSUPPORTED;
    const int mask = aux ? a.mask_mode : MASK_NONE;
    if ((long long)(PTHH + 1) * a.W * 128 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;   // in-tile byte offsets are 32-bit
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
