// The hot kernel of the path: 3x3, 64 -> 64 channels, bf16 in / fp32 accumulate, pixel-major.
//   core/modules/conv.py:85-86 (ResidualConv conv1/conv2), basicvsr.py:20 (conv_last.0),
//   upsampling.py:7 (as 4 pixel-shuffle phases), and all their data gradients: > 70 % of the FLOPs.
//
// Persistent design for one MI355X CU (160 KiB LDS, 4 SIMDs):
//   * ONE 512-thread workgroup per CU, grid = #CUs.  The packed weight tensor [9 taps][64 cout][64 cin]
//     bf16 (72 KiB) is loaded into LDS once and shared by everybody.
//   * the 8 waves form TWO independent groups of 4 waves (one wave of each group per SIMD).  Each
//     group walks its own 8x32-pixel tiles with its own haloed 10x34-pixel LDS buffer (42.5 KiB) and
//     synchronises only within the group (a 4-wave barrier on an LDS counter), so the two groups drift
//     apart: while one group waits for HBM or runs its epilogue, the other one owns the matrix pipes.
//   * no barrier inside the 144-MFMA K loop: A (weights) and B (pixels) fragments are plain
//     ds_read_b128 at register base + immediate offset.  Both LDS images are XOR-swizzled on the
//     16-byte chunk index (weights by cout row, pixels by x only, so a tap shift in y is a pure
//     immediate offset) => conflict-free reads (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * epilogue: the accumulator layout (lane = pixel, registers = channels) would store 8-byte pieces
//     into 32 different 128-byte lines per instruction (measured: as expensive as the K loop).  Each
//     wave therefore transposes its rows through a private 8 KiB slot of its group's (now idle) tile
//     buffer, fp32, and then reads residual / mask and writes the result as whole 128-byte pixel
//     lines, 16 bytes per lane.  Fused: bias, ReLU / LeakyReLU(0.1), residual add, activation-gradient
//     mask, pixel-shuffle placement; one rounding to bf16 at the very end.
#include "common.h"

namespace {

constexpr int PTW = 32, PTH = 8, PNT = 512, GNT = 256;
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;       // 34 x 10 = 340 haloed pixels
constexpr int W_BYTES = 9 * 64 * 64 * 2;                                  // 73,728
constexpr int IN_BYTES = PNPIX * 128;                                     // 43,520 per group (>= 4 waves x 8 KiB slots)
constexpr int BIAS_BYTES = 256;                                           // 64 fp32
constexpr int CNT_BYTES = 64;                                             // group-barrier counters
constexpr int P_LDS = W_BYTES + 2 * IN_BYTES + BIAS_BYTES + CNT_BYTES;    // 161,088 <= 163,840
constexpr int IN_CHUNKS = PNPIX * 8;                                      // 2,720 16-byte chunks
constexpr int PRE = (IN_CHUNKS + GNT - 1) / GNT;                          // 11 chunks per thread

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

// Barrier over the 4 waves of one group, on a monotonic LDS counter.  Every wave first drains its own
// LDS queue (its reads have returned, its writes have landed: the LDS executes a wave's operations in
// order), then arrives; all lanes poll the same word, so the loop is wave-uniform.  Global-memory
// operations are deliberately NOT waited for: epilogue stores stay in flight across the barrier.
__device__ __forceinline__ void group_barrier(unsigned* cnt, unsigned& target, int lane) {
    target += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// Epilogue variants are compile-time: a runtime-selected epilogue serialises 16 load->use->store
// chains per tile (measured: 14 us of a 71 us launch).
template <int ACT, bool HAS_RES, int MASK>
__global__ __launch_bounds__(PNT, 2) void conv3x3_c64_persist_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & (GNT - 1);
    const int l31 = lane & 31, h = lane >> 5;
    char* lds_w = smem;
    char* lds_in = smem + W_BYTES + grp * IN_BYTES;                      // this group's tile buffer
    float* lds_bias = reinterpret_cast<float*>(smem + W_BYTES + 2 * IN_BYTES);
    unsigned* gcnt = reinterpret_cast<unsigned*>(smem + W_BYTES + 2 * IN_BYTES + BIAS_BYTES) + grp * 8;

    const int z = blockIdx.y;
    const int ntx = cdiv(a.W, PTW), nty = cdiv(a.H, PTH);
    const int total = a.N * ntx * nty;
    if (tid < 64) lds_bias[tid] = a.bias ? a.bias[(long long)z * a.bias_zstride + tid] : 0.f;
    if (tid < 16) reinterpret_cast<unsigned*>(smem + W_BYTES + 2 * IN_BYTES + BIAS_BYTES)[tid] = 0u;

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
    __syncthreads();                                        // the only workgroup-wide barrier
    unsigned btarget = 0;

    const bf16_t* src = reinterpret_cast<const bf16_t*>(a.src[0]);
    // stage one haloed tile: chunk idx = gtid + 256 i -> pixel p = idx>>3 (ty = p/34, tx = p%34), chunk c = idx&7
    auto stage = [&](int tile) {
        int n, ty0, tx0;
        tile_coords(tile, ntx, nty, n, ty0, tx0);
        const bf16_t* org = src + (long long)n * a.src_nstride[0] + ((long long)(ty0 - 1) * a.W + (tx0 - 1)) * 64;
        uint4 pre[PRE];
        const bool interior = ty0 >= 1 && ty0 + PTH < a.H && tx0 >= 1 && tx0 + PTW < a.W;   // wave-uniform
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = gtid + i * GNT;
            const int p = idx >> 3, c = idx & 7;
            const int ty = p / PTWH, tx = p - ty * PTWH;
            const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < IN_CHUNKS && (interior || (vy >= 0 && vy < a.H && vx >= 0 && vx < a.W)))
                v = *reinterpret_cast<const uint4*>(org + (ty * a.W + tx) * 64 + c * 8);
            pre[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = gtid + i * GNT;
            const int p = idx >> 3, c = idx & 7;
            const int tx = p % PTWH;
            if (idx < IN_CHUNKS) *reinterpret_cast<uint4*>(lds_in + (p * 8 + (c ^ ((tx >> 1) & 7))) * 16) = pre[i];
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
            b_off[kx][ks] = ((gw * 2 * PTWH + tx) * 8 + ((2 * ks + h) ^ ((tx >> 1) & 7))) * 16;
        }
    // epilogue transposition slot of this wave: fp32 [32 pixels][64 channels], 16-byte chunk q of pixel
    // px stored at chunk (q ^ (px & 15))
    char* slot = lds_in + gw * 8192;
    const int rpx = lane >> 3, rch = lane & 7;            // read-back role: pixel rpx + 8*it, channels 8*rch..8*rch+7

#ifdef VSR_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int tile = blockIdx.x * 2 + grp; tile < total; tile += gridDim.x * 2) {
        STAMP(t0);
        stage(tile);
        STAMP(t1);
        group_barrier(gcnt, btarget, lane);                 // tile visible to the group
        STAMP(t2);

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
        STAMP(t3);
        group_barrier(gcnt, btarget, lane);                 // the group has finished reading the tile
        STAMP(t4);

        // ---- epilogue ----
        int n, ty0, tx0;
        tile_coords(tile, ntx, nty, n, ty0, tx0);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            // (1) bias + activation in accumulator layout, fp32, into the wave's slot
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * h;
                    const float4 b = *reinterpret_cast<const float4*>(lds_bias + co);
                    float4 v;
                    v.x = p_act<ACT>(acc[cb][rw][4 * g + 0] + b.x); v.y = p_act<ACT>(acc[cb][rw][4 * g + 1] + b.y);
                    v.z = p_act<ACT>(acc[cb][rw][4 * g + 2] + b.z); v.w = p_act<ACT>(acc[cb][rw][4 * g + 3] + b.w);
                    *reinterpret_cast<float4*>(slot + l31 * 256 + (((co >> 2) ^ (l31 & 15)) << 4)) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (2) read back pixel-major, finish in whole 128-byte lines: all loads first, then math + stores
            const int vy = ty0 + gw * 2 + rw;
            const int oy = vy * a.out_step + a.out_oy[z];
            float4 lo[4], hi[4];
            uint4 rr[4], mm[4];
            long long off[4];
            bool ok[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = rpx + 8 * it;
                lo[it] = *reinterpret_cast<const float4*>(slot + px * 256 + (((2 * rch) ^ (px & 15)) << 4));
                hi[it] = *reinterpret_cast<const float4*>(slot + px * 256 + (((2 * rch + 1) ^ (px & 15)) << 4));
                const int vx = tx0 + px;
                ok[it] = vy < a.H && vx < a.W;
                const int ox = vx * a.out_step + a.out_ox[z];
                off[it] = (long long)n * a.dst_nstride + ((long long)oy * a.Wd + ox) * 64 + rch * 8;
                if (HAS_RES && ok[it]) rr[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.res[z]) + off[it]);
                if (MASK != MASK_NONE && ok[it]) mm[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.aux[z]) + off[it]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (ok[it]) {
                    float v[8] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w, hi[it].x, hi[it].y, hi[it].z, hi[it].w};
                    if (HAS_RES) {
                        float r[8];
                        unpack_bf8(rr[it], r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += r[j];
                    }
                    if (MASK != MASK_NONE) {
                        float m[8];
                        unpack_bf8(mm[it], m);
                        constexpr float neg = MASK == MASK_LEAKY ? 0.1f : 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= (m[j] > 0.f ? 1.f : neg);
                    }
                    union { uint4 q; bf16_t hh[8]; } pk;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pk.hh[j] = (bf16_t)v[j];
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.dst[z]) + off[it]) = pk.q;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        STAMP(t5);
        group_barrier(gcnt, btarget, lane);                 // every wave of the group has left the tile buffer
        STAMP(t6);
        STAMP_ADD(1, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4); STAMP_ADD(5, t4, t5); STAMP_ADD(6, t5, t6);
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
    if (gx > (tiles + 1) / 2) gx = (tiles + 1) / 2;
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
    if ((long long)(PTHH + 1) * a.W * 64 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;   // in-tile offsets are 32-bit
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
