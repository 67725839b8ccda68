// EXPERIMENT (round 3, DESIGN 4.1c "where the energy goes"): the 3x3 64 -> 64 convolution with its WEIGHTS IN REGISTERS.
//
// On a sustained stream of these convolutions the chip is energy-bound, and the ablations of conv3x3_chain.hip price the weight
// (A) fragment reads from LDS at ~23 % of a layer's time -- every MFMA wave re-reads all 72 KiB of weights for every tile.  Here a
// workgroup is FOUR waves, one per SIMD with the whole 512-register file: each wave keeps the 72 A fragments of all nine taps in 288
// registers for the whole launch, reads only the pixel (B) fragments from LDS (half the LDS bytes), and issues the LDS-DMA of the
// next tile itself, one piece between two K steps (there are no producer waves: nothing else fits beside a 512-register wave).
// LDS holds just the two haloed tile buffers.  Same tile, same accumulator layout, same epilogue arithmetic as
// conv3x3_persist.hip (results bit-identical); only the two trunk epilogues (bias+ReLU, bias+identity) and unit steps.
// Selected by VSRLAB_AMD_WREG=1 at start-up or vsr_debug_set_wreg() (an A/B switch: tools/ab_wreg.py and the parity test flip it).
//
// STATUS (end of round 3): bit-identical to conv3x3_persist on every size tried, and SLOWER per launch: 54.8 us against 39.4-40.6 (540p,
// out of cache, one box).  Where the difference goes, by ablation (make ABL=<bits> ABLSRC=conv3x3_wreg):
//   * 12.5 us: the LDS-DMA pieces issued by the MFMA waves themselves -- ~260 cycles of the issuing wave per piece, 11 pieces per
//     tile, with one wave per SIMD nothing else runs meanwhile (the producer waves of conv3x3_persist exist for exactly this);
//   * ~12 us: 288 weight registers + 64 accumulators + fragments do not fit 256 + 256: hipcc keeps the ACCUMULATORS in the "a" half
//     and copies the ~31 weight fragments that overflow into VGPRs in front of their MFMAs (125 v_accvgpr_read per tile, each
//     padded against the MFMA hazards).  Inline-asm MFMAs with "a" operands remove the copies and lose hipcc's hazard padding
//     (register re-use right behind an asm MFMA: wrong results, measured); pinning with an empty asm does not survive allocation.
//   * with both ablated 37.7 us (conv3x3_persist 39.4 on the same box): the structure's own cycle count is competitive, and on a
//     sustained stream it would run with ~23 % less energy per layer.  Weights fetched by every wave for itself cost
//     another 4 us before they were staged through LDS.
// What a version that wins needs: the input tile through registers (global_load + ds_write, 30-odd cycles of issue per piece
// instead of 260) or two of the nine taps left in LDS to make room, and hand-written hazard padding around asm MFMAs.
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace {

// diagnostic builds (make ABL=<bits> ABLSRC=conv3x3_wreg; results wrong, time only): bit 0: no DMA after the first two tiles; bit 1: the
// weights' LDS image is not read into the registers' own values (all fragments = the first)
#ifdef VSR_ABL
#define WABL(bit) ((VSR_ABL >> (bit)) & 1)
#else
#define WABL(bit) 0
#endif

constexpr int PTW = 32, PTH = 8, WNT = 256;                              // 4 waves, each 2 rows of the 8x32 tile
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;
constexpr int IN_BYTES = PNPIX * 128;                                     // 43,520 per buffer
constexpr int W_LDS = 3 * IN_BYTES;                                       // three tile buffers: the DMA runs a whole tile ahead
constexpr int IN_CHUNKS = PNPIX * 8;
constexpr int NPIECE_T = (IN_CHUNKS + 63) / 64;                           // 43 DMA pieces of 1 KiB
constexpr int NPIECE_W = (NPIECE_T + 3) / 4;                              // 11 per wave

__device__ uint4 g_wreg_zero_chunk[2];

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2_t p = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

#define GLDS16(src, dst)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define GP(T, x) ((__attribute__((address_space(1))) T*)(x))

template <int ACT, bool HAS_RES>
__global__ __launch_bounds__(WNT, 1) void conv3x3_c64_wreg_kernel(const ConvArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);
    char* lds_t = smem;
    const int H = ka.H, W = ka.W;
    const int ntx = cdiv(W, PTW), nty = cdiv(H, PTH), per = ntx * nty;
    const int total = ka.N * per;
    const int WS = pm_ws(W);
    const long long img = pm_image_elems(H, W, 64);
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);

    // ---- the weights: A fragment (tap, kk, mb) of this lane = channels 32 kk + 8 q .. + 7 of output channel pm_acc_chan(mb, l15).
    // Staged through LDS once per workgroup (coalesced, 72 KiB from L2 per CU; the image and its XOR swizzle are conv3x3_persist's),
    // then read into registers by every wave: fetched by each wave for itself from global memory the 4 x 72 KiB per CU cost 16 us of
    // a 59 us launch.  The staging area is tile buffers 1 and 2; the first tile's DMA (buffer 0) runs meanwhile. ----
    bf16x8_t wreg[9][2][4];
    char* lds_w = smem + IN_BYTES;
    constexpr int WCH = 9 * 64 * 8 / 256;                  // 18 sixteen-byte chunks per thread
    u32x4_t wv[WCH];
    {
        const auto* wg = GP(const u32x4_t, ka.wpack);
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int idx = tid + i * 256;
            const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;
            wv[i] = wg[(tap * 64 + pm_acc_chan(r >> 4, r & 15)) * 8 + c];
        }
    }
    f32x4_t bvec[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) bvec[mb][j] = ka.bias ? ka.bias[pm_acc_chan(mb, 4 * q + j)] : 0.f;
    bf16x8_t idA[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int j = 0; j < 8; ++j) idA[hb][j] = (bf16_t)((q == (l15 >> 2) && j == 4 * hb + (l15 & 3)) ? 1.f : 0.f);

    // ---- DMA pieces of this wave: piece = w4 + 4 i ----
    const auto* src = GP(const char, ka.src[0]);
    const auto* zsrc = GP(const char, g_wreg_zero_chunk);
    int rel[NPIECE_W];
#pragma unroll
    for (int i = 0; i < NPIECE_W; ++i) {
        const int idx = (w4 + 4 * i) * 64 + lane;
        const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
        const int c = rem / PTWH, tx = rem - c * PTWH;
        const int dx = tx - 1;
        rel[i] = ((((ty - 1) * WS + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
    }
    // one piece of tile (n, ty0, tx0) into buffer `buf`; `interior`: no bounds needed (wave-uniform)
    auto piece = [&](int i, const char* org, int ty0, int tx0, bool interior, int buf) {
        const int pc = w4 + 4 * i;
        const int idx = pc * 64 + lane;
        if (pc < NPIECE_T && idx < IN_CHUNKS) {
            const char* s = org + rel[i];
            if (!interior) {
                const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
                const int tx = rem % PTWH;
                const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                if (!(vy >= 0 && vy < H && vx >= 0 && vx < W)) s = (const char*)zsrc;
            }
            GLDS16(s, lds_t + buf * IN_BYTES + pc * 1024);
        }
    };
    auto tile_org = [&](int tile, int& n, int& ty0, int& tx0) {
        n = tile / per;
        const int r = tile - n * per, ty = r / ntx;
        ty0 = ty * PTH; tx0 = (r - ty * ntx) * PTW;
        return src + ((long long)n * ka.src_nstride[0] + pm_off(ty0, tx0, 0, W, 64)) * 2;
    };

    unsigned loff[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int dx = (nb & 1) * 16 + pxl;
        loff[nb] = (((((w4 * 2 + (nb >> 1))) * WS + (dx >> 5)) * 8 + q) * 256 + (dx & 31) * 8) * 2;
    }
    const int b_lane = w4 * 2 * (PTWH * 128) + q * (PTWH * 16) + pxl * 16;

    int tile = walk.first;
    // (a workgroup without tiles still takes part in nothing: it leaves before the first barrier)
    if (tile >= walk.end) return;
    auto first_tiles = [&](int pre) {            // tile `pre` (0 or 1) of this workgroup into buffer `pre`
        const int t0 = tile + pre * walk.stride;
        if (t0 < walk.end) {
            int n, ty0, tx0;
            const char* org = (const char*)tile_org(t0, n, ty0, tx0);
            const bool interior = ty0 >= 1 && ty0 + PTH < H && tx0 >= 1 && tx0 + PTW < W;
#pragma unroll
            for (int i = 0; i < NPIECE_W; ++i) piece(i, org, ty0, tx0, interior, pre);
        }
    };
    first_tiles(0);
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int idx = tid + i * 256;
        const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;
        *reinterpret_cast<u32x4_t*>(lds_w + tap * 8192 + (r * 8 + (c ^ ((r >> 1) & 7))) * 16) = wv[i];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                wreg[tap][kk][mb] = *reinterpret_cast<const bf16x8_t*>(lds_w + tap * 8192 + mb * 2048 + (l15 * 8 + ((4 * kk + q) ^ ((l15 >> 1) & 7))) * 16);
    __syncthreads();                             // everybody has its fragments: buffers 1 and 2 are free
    first_tiles(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int cur = 0;
    for (; tile < walk.end; tile += walk.stride) {
        int n, ty0, tx0;
        (void)tile_org(tile, n, ty0, tx0);
        const long long tbase = (long long)n * img + pm_off(ty0, tx0, 0, W, 64);
        const bool full = ty0 + PTH <= H && tx0 + PTW <= W;
        bool ok[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) ok[nb] = (tx0 + (nb & 1) * 16 + pxl < W) && (ty0 + w4 * 2 + (nb >> 1) < H);
        // the tile after the next, requested piece by piece between the K steps below (the next one was requested a tile ago)
        const int ntile = tile + 2 * walk.stride;
        const bool have_next = ntile < walk.end;
        int nn = 0, nty0 = 0, ntx0 = 0;
        const char* norg = have_next ? (const char*)tile_org(ntile, nn, nty0, ntx0) : (const char*)src;
        const bool ninterior = nty0 >= 1 && nty0 + PTH < H && ntx0 >= 1 && ntx0 + PTW < W;

        // residual operands first: they are the OLDEST entries of the wave's vector-memory queue, the DMA pieces come behind them
        u32x4_t rr[2][4];
        if (HAS_RES) {
            const unsigned long long rbase = (unsigned long long)ka.res[0] + (unsigned long long)tbase * 2ull;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const unsigned lo = loff[nb];
                rr[0][nb] = u32x4_t{0u, 0u, 0u, 0u}; rr[1][nb] = u32x4_t{0u, 0u, 0u, 0u};
                if (full || ok[nb]) {
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(rr[0][nb]) : "v"(lo), "s"(rbase) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(rr[1][nb]) : "v"(lo), "s"(rbase) : "memory");
                }
            }
        }

        f32x4_t acc[4][4];
        bf16x8_t fb[2][4];
        const unsigned bb = (unsigned)(cur * IN_BYTES + b_lane);
        const int nxt2 = cur == 0 ? 2 : cur - 1;                    // (cur + 2) % 3
#define DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define WR_LOADB(ky_, kx_, kk_, slot, nb) DSR(fb[slot][nb], bb, (((nb) >> 1) + ky_) * (PTWH * 128) + kk_ * (4 * PTWH * 16) + (((nb) & 1) * 16 + kx_) * 16);
#define WR_MFMA(s, mb, nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[(s) / 2][(s) % 2][mb], fb[(s) % 2][nb], (s) == 0 ? bvec[mb] : acc[mb][nb], 0, 0, 0);
#define WR_ML(s, mb, nb, lnb)                                                                                          \
        WR_MFMA(s, mb, nb)                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((s) + 1 < 18) { constexpr int t1_ = ((s) + 1) / 2, k1_ = ((s) + 1) % 2; WR_LOADB(t1_ / 3, t1_ % 3, k1_, ((s) + 1) % 2, lnb) } \
        __builtin_amdgcn_sched_barrier(0);
#define WR_STEP(s)                                                                                                     \
        {                                                                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            WR_ML(s, 0, 0, 0) WR_ML(s, 0, 1, 1) WR_ML(s, 0, 2, 2) WR_ML(s, 0, 3, 3)                                    \
            WR_MFMA(s, 1, 0) WR_MFMA(s, 1, 1) WR_MFMA(s, 1, 2) WR_MFMA(s, 1, 3)                                        \
            WR_MFMA(s, 2, 0) WR_MFMA(s, 2, 1) WR_MFMA(s, 2, 2) WR_MFMA(s, 2, 3)                                        \
            WR_MFMA(s, 3, 0) WR_MFMA(s, 3, 1) WR_MFMA(s, 3, 2) WR_MFMA(s, 3, 3)                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((s) < NPIECE_W && have_next && !WABL(0)) piece((s), norg, nty0, ntx0, ninterior, nxt2);                         \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        WR_LOADB(0, 0, 0, 0, 0) WR_LOADB(0, 0, 0, 0, 1) WR_LOADB(0, 0, 0, 0, 2) WR_LOADB(0, 0, 0, 0, 3)
        WR_STEP(0) WR_STEP(1) WR_STEP(2) WR_STEP(3) WR_STEP(4) WR_STEP(5) WR_STEP(6) WR_STEP(7) WR_STEP(8)
        WR_STEP(9) WR_STEP(10) WR_STEP(11) WR_STEP(12) WR_STEP(13) WR_STEP(14) WR_STEP(15) WR_STEP(16) WR_STEP(17)
#undef WR_STEP
#undef WR_ML
#undef WR_MFMA
#undef WR_LOADB
#undef DSR
        // ---- epilogue.  The residual loads are older than the 11 DMA pieces: a counted wait leaves the pieces in flight ----
        if (HAS_RES) {
            // (wave 3 issues 10 pieces, the others 11: "all but the 10 youngest" covers every residual load either way)
            if (have_next) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idA[mb & 1], __builtin_bit_cast(bf16x8_t, rr[mb >> 1][nb]), acc[mb][nb], 0, 0, 0);
        }
        const unsigned long long dbase = (unsigned long long)ka.dst[0] + (unsigned long long)tbase * 2ull;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (full || ok[nb]) {
                const unsigned lo = loff[nb];
#pragma unroll
                for (int kq = 0; kq < 2; ++kq) {
                    unsigned ow[4];
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[2 * kq][nb][j]; v[4 + j] = acc[2 * kq + 1][nb][j]; }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        ow[jj] = pk_bf16(v[2 * jj], v[2 * jj + 1]);
                        if (ACT == ACT_RELU) ow[jj] = pk_max_i16(ow[jj], 0u);
                    }
                    const u32x4_t o = {ow[0], ow[1], ow[2], ow[3]};
                    if (kq == 0) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(lo), "v"(o), "s"(dbase) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048\n\ts_nop 1" :: "v"(lo), "v"(o), "s"(dbase) : "memory");
                }
            }
        }
        // the next tile has to have landed before anybody reads it
        // (a bare s_barrier: __syncthreads() would make hipcc wait for vmcnt(0), i.e. for the stores just issued)
        // (the NEXT tile's pieces were issued a tile ago: they are older than this tile's 10-11 pieces and 8 stores)
        if (full) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur = cur == 2 ? 0 : cur + 1;
    }
}

template <int ACT, bool HAS_RES>
int launch_wreg(const ConvArgs& a, int num_cus, hipStream_t st) {
    auto kern = conv3x3_c64_wreg_kernel<ACT, HAS_RES>;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), W_LDS); if (rc != VSR_OK) return rc; }
    const int tiles = a.N * cdiv(a.W, PTW) * cdiv(a.H, PTH);
    int gx = num_cus & ~7;
    if (gx < 8) gx = num_cus;
    if (gx > tiles) gx = tiles;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WNT), W_LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

// The two trunk epilogues of a plain unit-step launch only; anything else: VSR_ERR_UNSUPPORTED (the caller goes on to conv3x3_persist)
int vsr_launch_conv3x3_c64_wreg(const ConvArgs& a, int num_cus, hipStream_t st) {
    if (a.nz != 1 || a.in_step != 1 || a.out_step != 1 || a.src_ox[0] || a.src_oy[0] || a.out_ox[0] || a.out_oy[0] || a.Ws != a.W || a.Wd != a.W ||
        a.aux[0] || a.sign_out[0] || a.sign_bits[0] || a.dst_nstride != pm_image_elems(a.H, a.W, 64))
        return VSR_ERR_UNSUPPORTED;
    if (pm_image_elems(2 * PTH + 2, a.W, 64) * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
    if (a.act == ACT_RELU && !a.res[0]) return launch_wreg<ACT_RELU, false>(a, num_cus, st);
    if (a.act == ACT_NONE && a.res[0]) return launch_wreg<ACT_NONE, true>(a, num_cus, st);
    return VSR_ERR_UNSUPPORTED;
}
