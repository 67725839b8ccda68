// The hot kernel of the path: 3x3, 64 -> 64 channels, bf16 in / fp32 accumulate.
//   core/modules/conv.py:85-86 (ResidualConv conv1/conv2), basicvsr.py:20 (conv_last.0),
//   upsampling.py:7 (as 4 pixel-shuffle phases), and all their data gradients: > 70 % of the FLOPs.
//
// At 540p one launch moves 133-200 MB for 38 GFLOP: 21-32 us at the 6.3 TB/s a CU-side stream reaches,
// 15 us on the matrix cores -- the kernel is HBM-bound, and a CU needs ~75 KB in flight to cover ~3 us of
// loaded HBM latency.  Design for one MI355X CU (160 KiB LDS, 4 SIMDs):
//   * ONE 512-thread workgroup per CU, grid = #CUs, persistent over 8x32-pixel tiles (one 32-pixel segment
//     of the blocked layout wide, see common.h): 4 MFMA waves + 4 LDS-DMA producer waves, one of each per
//     SIMD.  The packed weights [9 taps][64 cout][64 cin] bf16 (72 KiB) are loaded into LDS once and stay.
//   * the haloed 10x34-pixel input tile is DOUBLE-BUFFERED in LDS (2 x 42.5 KiB) and filled by the producer
//     waves' LDS-DMA (global_load_lds_dwordx4) one whole tile ahead; the residual / mask operands of the
//     epilogue are requested into registers before the K loop.  Nothing the tile needs is waited for at first use.
//   * K loop: 288 v_mfma_f32_16x16x32_bf16 per MFMA wave and tile, no barrier inside; A (weights) and B (pixels)
//     fragments are ds_read_b128 at register base + immediate, issued by hand one k-step ahead of their
//     MFMAs (inline asm + counted s_waitcnt: hipcc sinks builtin LDS reads back in front of their
//     consumers).  The weight image is
//     [cout row][8 chunks of 16 B], XOR-swizzled on the chunk; the pixel image mirrors the blocked global
//     layout, [tile row][chunk][34 pixels][16 B]: 16 consecutive lanes read 16 consecutive 16-byte slots
//     (conflict-free without a swizzle), a ky shift is an immediate, and a DMA piece reads runs of up to 512
//     contiguous bytes (a [pixel][chunk] image made every lane of a piece hit a different line:
//     +40 % DMA issue time measured).
//   * epilogue: the accumulator layout (lane = pixel of the segment, registers = 4 consecutive rows of a 16-row block) IS
//     the blocked global layout once the 64 output channels are dealt to the MFMA rows in "paired-block order"
//     (pm_acc_chan, common.h): lane (pixel i, quarter q) then holds the 8 consecutive channels of chunk 4k + q in the
//     accumulators of blocks 2k and 2k+1, i.e. one whole 16-byte piece of the blocked layout.  Bias / ReLU / LeakyReLU(0.1) /
//     residual add / activation-gradient mask / pixel-shuffle placement are applied in registers and every store (and
//     residual / mask load) is a 16-byte-per-lane wave instruction over four 256-byte runs: 8 stores + 8 loads per wave and
//     tile.  (r03: with 4 channels per lane = 16 + 16 eight-byte instructions, the vector-memory issue path, shared with
//     the producers' 44 DMA pieces per tile, cost 1.2 k cycles per tile for the operand loads alone; with plain
//     [pixel][64 ch] rows the stores hit 32 lines per instruction; an LDS transposition cost 2.7 k cycles per tile.)
//     ReLU and the sign bits are computed on the PACKED bf16 words (v_pk_max_i16, v_pk_min_u16, v_lshl_or_b32: 1.5
//     instructions per element instead of 4).
#include "common.h"
#include <type_traits>

namespace {

constexpr int PTW = 32, PTH = 8, PNT = 512;                              // 4 MFMA waves + 4 DMA waves
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;       // 34 x 10 = 340 haloed pixels
constexpr int W_BYTES = 9 * 64 * 64 * 2;                                  // 73,728
constexpr int IN_BYTES = PNPIX * 128;                                     // 43,520 per buffer
constexpr int BIAS_OFF = W_BYTES + 2 * IN_BYTES;                          // 64 fp32 bias values behind the tiles
constexpr int P_LDS = BIAS_OFF + 256;                                     // 161,024 <= 163,840
constexpr int IN_CHUNKS = PNPIX * 8;                                      // 2,720 16-byte chunks
constexpr int NPIECE_T = (IN_CHUNKS + 63) / 64;                           // 43 DMA pieces of 1 KiB (last half full)
constexpr int NPIECE_W = (NPIECE_T + 3) / 4;                              // 11 per wave

__device__ uint4 g_conv_zero_chunk[2];

template <int ACT> __device__ __forceinline__ float p_act(float v, float slope) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_LEAKY) return v > 0.f ? v : slope * v;
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

// Diagnostic build only (make ABL=<bits>): ablations that bound what a restructuring could buy, and the in-kernel clock
// (s_memtime / s_memrealtime around the whole kernel, lane 0 of wave 0, to a buffer of their own).  Results of an ablated
// build are WRONG by construction; only its run time and clock are read.  bit 0: no B-fragment reads for ky = 2 (what
// ky-shared B fragments would save); bit 1: no sign-bit output; bit 2: no A-fragment reads for ky = 2; bit 3: producers
// issue only the first tile; bit 4: epilogue = convert + store only; bit 5: no epilogue stores; bit 6: no fragment reads
// after step 0 (bare MFMA loop); bit 7: no lgkmcnt wait at the top of a step.
#ifdef VSR_ABL
__device__ unsigned long long g_clk[256 * 4];
#define ABL(bit) ((VSR_ABL >> (bit)) & 1)
#else
#define ABL(bit) 0
#endif

__device__ __forceinline__ void tile_coords(int tile, int ntx, int nty, int& n, int& ty0, int& tx0) {
    const int per = ntx * nty;
    n = tile / per;
    const int r = tile - n * per;
    const int ty = r / ntx;
    ty0 = ty * PTH;
    tx0 = (r - ty * ntx) * PTW;
}

// Walks a workgroup's tiles first, first + stride, ... keeping (image, tile row, tile column): two divisions once, then a handful of
// scalar adds and compares per tile (tile_coords per tile = two runtime divisions through v_rcp_iflag_f32, ~60 instructions).
struct TileIter {
    int tile, n, ty, tx;          // tile index; image, tile row, tile column
    int sn, sty, stx;             // the stride in those units
    __device__ __forceinline__ void init(int first, int stride, int ntx, int nty) {
        const int per = ntx * nty;
        tile = first; n = first / per; int r = first - n * per; ty = r / ntx; tx = r - ty * ntx;
        sn = stride / per; r = stride - sn * per; sty = r / ntx; stx = r - sty * ntx;
    }
    __device__ __forceinline__ void advance(int stride, int ntx, int nty) {
        tile += stride;
        tx += stx; if (tx >= ntx) { tx -= ntx; ++ty; }
        ty += sty; if (ty >= nty) { ty -= nty; ++n; }
        n += sn;
    }
};

// two fp32 -> one dword of two bf16 (v_cvt_pk_bf16_f32), low half = a
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;     // a 16-byte piece (native vector: usable behind address-space pointers)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2_t p = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_mul_lo_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Diagnostic bit 8: nt on the tile DMA; bit 9: nt on the output stores.  tools/bw_probe.hip streams reads at 6.4-7.0 TB/s with nt against
// 5.7-6.2 without, and out of cache (tools/ab_conv.py, 8 rotating buffer sets) nt STORES take 7 % off a bias+ReLU launch (36.7 -> 34.0 us;
// bias+skip 42.3 -> 41.5; nt loads: 36.2 / 44.3) -- but inside a step, applied to the outputs of >= 512 MB (the 16P tensors of the
// reconstruction), the step is 0.2-0.7 ms SLOWER (124.6-124.9 -> 125.1-125.6 ms, three interleaved pairs): their consumers do find the
// tail of such a tensor in the Infinity Cache.  Not used (r04).
#ifdef VSR_ABL
#define P_DMA_AUX (((VSR_ABL >> 8) & 1) ? 2 : 0)
#define P_NT_STORE ((VSR_ABL >> 9) & 1)
#else
#define P_DMA_AUX 0
#define P_NT_STORE 0
#endif
#define GLDS16(src, dst)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, P_DMA_AUX)

// Epilogue variants are compile-time: a runtime-selected epilogue serialises 16 load->use->store
// chains per tile (measured: 14 us of a 71 us launch).
template <int ACT, bool HAS_RES, int MASK>
__global__ __launch_bounds__(PNT, 1) void conv3x3_c64_persist_kernel(const ConvArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Every kernel argument the kernel uses, requested in ONE batch at the top and pinned in scalar registers: hipcc otherwise
    // loads arguments lazily, next to their first use, and the producers' way to the first DMA piece went through three
    // dependent s_load + lgkmcnt(0) rounds on the (cold) kernarg segment (r03 stamps: 3.0 k cycles from the start of the
    // launch to the first piece; the first tile is the critical path of the prologue).
    const int z = blockIdx.y;
    struct {
        int N, H, W, Ws, Wd, in_step, out_step, src_ox0, src_oy0, out_oxz, out_oyz, bias_zstride, unshuffle;
        long long src_nstride0, dst_nstride, w_zstride, plane;
        unsigned long long src0, wpack, resz, auxz, sign_bitsz, dstz, sign_outz, bias; float leaky_slope;   // pointers as integers: GP() below
    } a = {ka.N, ka.H, ka.W, ka.Ws, ka.Wd, ka.in_step, ka.out_step, ka.src_ox[0], ka.src_oy[0], ka.out_ox[z], ka.out_oy[z], ka.bias_zstride, ka.unshuffle,
           ka.src_nstride[0], ka.dst_nstride, ka.w_zstride, ka.unshuffle_plane,
           (unsigned long long)ka.src[0], (unsigned long long)ka.wpack, (unsigned long long)ka.res[z], (unsigned long long)ka.aux[z],
           (unsigned long long)ka.sign_bits[z], (unsigned long long)ka.dst[z], (unsigned long long)ka.sign_out[z], (unsigned long long)ka.bias, ka.leaky_slope};
    asm volatile("" : "+s"(a.N), "+s"(a.H), "+s"(a.W), "+s"(a.Ws), "+s"(a.Wd), "+s"(a.in_step), "+s"(a.out_step), "+s"(a.src_ox0), "+s"(a.src_oy0),
                 "+s"(a.out_oxz), "+s"(a.out_oyz), "+s"(a.bias_zstride), "+s"(a.unshuffle), "+s"(a.src_nstride0), "+s"(a.dst_nstride), "+s"(a.w_zstride), "+s"(a.plane));
    asm volatile("" : "+s"(a.src0), "+s"(a.wpack), "+s"(a.resz), "+s"(a.auxz), "+s"(a.sign_bitsz), "+s"(a.dstz), "+s"(a.sign_outz), "+s"(a.bias), "+s"(a.leaky_slope));
    // a pinned pointer comes back as an integer: name its address space, or hipcc addresses it with flat_ instructions
#define GP(T, x) ((__attribute__((address_space(1))) T*)(x))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);            // 0: MFMA + epilogue, 1: LDS-DMA producer
    const int w4 = wave & 3;                                               // index within the role
    const int l15 = lane & 15, q = lane >> 4;                              // 16x16x32 operand / accumulator coordinates
    // Pixel of a block that lane column l15 works on.  ds_read_b128 is serviced in lane groups {0-3,12-15,20-27}, {4-11,16-19,
    // 28-31}, ... (MI355X_MICROARCH, LDS): a group mixes two channel chunks (q, q+1) whose images are 34 slots = 2 mod 16
    // apart, so with pixel = l15 two of its 16 slots share banks (SQ_LDS_BANK_CONFLICT: 2 extra cycles per B read).  Lanes
    // 4-11 on the even pixels and lanes 0-3, 12-15 on the odd ones make every group hit 16 distinct 16-byte bank groups.
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);
    char* lds_w = smem;
#ifdef VSR_STAMPS
    const unsigned long long st_begin = stamp();
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    char* lds_t = smem + W_BYTES;                                         // two tile buffers
#ifdef VSR_ABL
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (tid == 0) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif

    const int ntx = cdiv(a.W, PTW), nty = cdiv(a.H, PTH);
    const int total = a.N * ntx * nty;
    const int WSs = pm_ws(a.Ws), WSd = pm_ws(a.Wd);           // source / destination images may be larger than the view
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);

    // bias of this z as fp32 in LDS: the accumulators of every tile start from it (32 fewer live registers
    // than carrying it, which is what lets two waves share a SIMD)
    // (slot r = the channel MFMA row r of the 64 computes: paired-block order, see the epilogue)
    if (tid < 64) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = a.bias ? GP(const float, a.bias)[(long long)z * a.bias_zstride + pm_acc_chan(tid >> 4, tid & 15)] : 0.f;

    // ---- weights of this z: global [tap][cout][cin] -> LDS, row r = MFMA row (block r >> 4, row r & 15) holds output channel
    // pm_acc_chan(r >> 4, r & 15); chunk c of row r at (r*8 + (c ^ ((r>>1)&7))): the 16 rows a ds_read_b128 pass touches (same
    // chunk, rows 16 mb .. 16 mb + 15) then fall on 16 distinct 16-byte bank groups ----
    // The four MFMA waves stage all 4608 chunks (18 per thread; the loads are issued first, the waves' own set-up arithmetic runs
    // under their latency, then the LDS writes); the producer waves meanwhile do nothing but get the first tile's DMA out: at
    // the start of a launch the tile is the critical path (r03 stamps: the producers took 2.7 k cycles of set-up + 2.2 k of
    // issue + 3.2 k for their half of the weights before the first barrier; 8.4 k cycles of a 63 k-cycle launch).
    constexpr int WCH = 9 * 64 * 8 / 256;                  // 18
    const auto* wg = GP(const u32x4_t, a.wpack + (unsigned long long)((long long)z * a.w_zstride) * 2);
    // (written out in the MFMA branch: a register array handed to a lambda by reference goes through scratch memory)
#define W_LOAD(wv)                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                                    \
        const int idx = tid + i * 256;                                                                                   \
        const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;                                                      \
        wv[i] = wg[(tap * 64 + pm_acc_chan(r >> 4, r & 15)) * 8 + c];                                                    \
    }
#define W_STORE(wv)                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                                    \
        const int idx = tid + i * 256;                                                                                   \
        const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;                                                      \
        *reinterpret_cast<u32x4_t*>(lds_w + tap * 8192 + (r * 8 + (c ^ ((r >> 1) & 7))) * 16) = wv[i];                     \
    }

    if (role == 1) {
        // =================== producer waves: LDS-DMA of the haloed tiles, one tile ahead ===================
        // Their VMEM issue slots (an LDS-DMA instruction does not finish issuing until the memory pipe takes it) run
        // beside the MFMA waves' matrix-core time on the same SIMDs.  DMA pieces of this wave: piece = w4 + 4 i (64
        // consecutive 16-byte LDS slots).  rel[i] = the source BYTE offset of this lane's slot (row ty, chunk c, pixel
        // tx) from the tile origin in the blocked layout (tx0 is a multiple of 32: the 32 inner pixels of a row are one
        // global segment, i.e. 512 contiguous bytes per chunk; the halo columns are the last / first pixel of the
        // neighbouring segments).
        const auto* src = GP(const char, a.src0);
        const auto* zsrc = GP(const char, g_conv_zero_chunk);
        int rel[NPIECE_W];
#pragma unroll
        for (int i = 0; i < NPIECE_W; ++i) {
            const int idx = (w4 + 4 * i) * 64 + lane;           // LDS slot = [row ty][chunk c][34 pixels tx] x 16 B
            const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
            const int c = rem / PTWH, tx = rem - c * PTWH;
            // view pixel (ty-1, tx-1) is source pixel (v*in_step + src_o): in_step 2 = one pixel-shuffle phase of a
            // twice-as-large tensor (the data gradient of conv3x3 + PixelShuffle, one launch per phase)
            const int dx = (tx - 1) * a.in_step + a.src_ox0;
            rel[i] = (((((ty - 1) * a.in_step + a.src_oy0) * WSs + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
        }
        auto issue = [&](const TileIter& it, int buf) {
            const int n = it.n, ty0 = it.ty * PTH, tx0 = it.tx * PTW;
            const auto* org = src + ((long long)n * a.src_nstride0 + pm_off(ty0 * a.in_step, tx0 * a.in_step, 0, a.Ws, 64)) * 2;
            char* dstb = lds_t + buf * IN_BYTES;
            if (ty0 >= 1 && ty0 + PTH < a.H && tx0 >= 1 && tx0 + PTW < a.W) {          // interior tile (wave-uniform)
#pragma unroll
                for (int i = 0; i < NPIECE_W; ++i) {
                    const int piece = w4 + 4 * i;
                    if (piece < NPIECE_T && piece * 64 + lane < IN_CHUNKS) GLDS16(org + rel[i], dstb + piece * 1024);
                }
            } else {                                                                    // border: bounds per lane, zero source
#pragma unroll
                for (int i = 0; i < NPIECE_W; ++i) {
                    const int piece = w4 + 4 * i;
                    const int idx = piece * 64 + lane;
                    const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
                    const int tx = rem % PTWH;
                    const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                    const auto* s = (vy >= 0 && vy < a.H && vx >= 0 && vx < a.W) ? org + rel[i] : zsrc;
                    if (piece < NPIECE_T && idx < IN_CHUNKS) GLDS16(s, dstb + piece * 1024);
                }
            }
        };
        int cur = 0;
        TileIter it;
        it.init(walk.first, walk.stride, ntx, nty);
        STAMP(q0);
        if (it.tile < walk.end) issue(it, 0);              // first tile in flight while the MFMA waves stage the weights
        STAMP(q1);
        STAMP_ADD(1, st_begin, q0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(q3);
        __syncthreads();                                   // weights and the first tile are in LDS
        STAMP_ADD(2, st_begin, q1); STAMP_ADD(4, q1, q3);
        while (it.tile < walk.end) {
            STAMP(t0);
            it.advance(walk.stride, ntx, nty);
            if (it.tile < walk.end && !ABL(3)) issue(it, cur ^ 1);
            STAMP(p1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(p2);
            __syncthreads();
            STAMP(p3);
            STAMP_ADD(5, t0, p1); STAMP_ADD(6, p1, p2); STAMP_ADD(7, p2, p3);
            cur ^= 1;
        }
    } else {
        // =================== MFMA waves: K loop + epilogue of tile rows 2 w4, 2 w4 + 1 ===================
        // v_mfma_f32_16x16x32_bf16 (under MFMA load on random data the chip holds a higher clock on this shape than
        // on 32x32x16, MI355X_MICROARCH "DVFS give-back" item 7): wave tile = 4 cout blocks x 4 pixel blocks (row,
        // half) of 16; a step = one tap x 32 channels = 8 fragment reads (4 A + 4 B, ds_read_b128) + 16 MFMAs.
        // Operand lane l = (i = l & 15, q = l >> 4): A[cout 16 mb + i][8 channels 8q..8q+7], B[same 8 channels][pixel i].
        // Fragment addresses: A two lane bases per channel half (taps 0-5 / 6-8: the immediate is 16 bits) + immediates
        // (tap, mb); B ONE lane base + immediates (row, ky, kx, half, channel half) of the [row][chunk][34 px][16 B] image.
        STAMP(m0);
        u32x4_t wv[WCH];
        W_LOAD(wv)
        unsigned a_lo[2], a_hi[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_lo[kk] = (unsigned)((l15 * 8 + ((4 * kk + q) ^ ((l15 >> 1) & 7))) * 16);
            a_hi[kk] = a_lo[kk] + 6 * 8192;
        }
        const int b_lane = w4 * 2 * (PTWH * 128) + q * (PTWH * 16) + pxl * 16;
        // epilogue: accumulator blocks (2k, nb) and (2k+1, nb), registers j = channels 8 (4k + q) + j and + 4 + j of pixel
        // (row nb >> 1, 16 (nb & 1) + i): chunk 4k + q of the blocked layout, whole.  loff[nb] = lane-constant part of the
        // destination element offset relative to the tile's origin pm_off(ty0*os + ooy, tx0*os) (tx0*os: multiple of 32)
        int loff[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int dx = ((nb & 1) * 16 + pxl) * a.out_step + a.out_oxz;
            loff[nb] = ((((w4 * 2 + (nb >> 1)) * a.out_step) * WSd + (dx >> 5)) * 8 + q) * 256 + (dx & 31) * 8;
            if (a.unshuffle) {      // phase-separated destination (ConvArgs::unshuffle): plane 2 (row & 1) + (px & 1), position (row >> 1, px >> 1)
                const int row = w4 * 2 + (nb >> 1), px = (nb & 1) * 16 + pxl;
                loff[nb] = (int)(((row & 1) * 2 + (px & 1)) * a.plane) + (((row >> 1) * pm_ws(a.Wd >> 1)) * 8 + q) * 256 + (px >> 1) * 8;
            }
        }
        W_STORE(wv)
        STAMP(m1);
        STAMP_ADD(6, m0, m1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // weights and the first tile are in LDS
#ifdef VSR_STAMPS
        st_sum[5] = stamp() - st_begin;                    // prologue of the MFMA waves (weights staged, first tile landed)
#endif
        auto* const dst_z = GP(bf16_t, a.dstz);
        const auto* const res_z = GP(const bf16_t, a.resz);
        const auto* const aux_z = GP(const bf16_t, a.auxz);
        const auto* const sbits_z = GP(const u32x2_t, a.sign_bitsz);
        auto* const sout_z = GP(u32x2_t, a.sign_outz);
        const float slope = vsr_slope(a.leaky_slope);      // LeakyReLU slope (0.1 on the BasicVSR path, 0.2 in the discriminator)

        // The bias of the wave's 16 rows per block (couts pm_acc_chan(mb, 4q + j)) stays in 16 registers and is the C operand of
        // every accumulator's first MFMA: no per-tile initialisation (64 moves + 4 LDS reads per tile before).
        f32x4_t bvec[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) bvec[mb] = *reinterpret_cast<const f32x4_t*>(smem + BIAS_OFF + (mb * 16 + 4 * q) * 4);
        // The residual is added ON THE MATRIX CORES (r03): the 16-byte residual piece a lane prefetches (chunk 4k + q of its pixel) is
        // exactly a B fragment of a K step over the 32 channels of pieces k, and "+ residual" is one more MFMA per accumulator with a
        // 0 / 1 selection matrix as A: row r of block mb is channel pm_acc_chan(mb, r) = k group r >> 2, element 4 (mb & 1) + (r & 3) of
        // that step.  16 MFMAs (256 cycles) replace 64 unpacks + 64 adds per tile and wave (512 issue cycles); fp32 accumulation of
        // 1.0 x bf16 is the same single rounding as the v_add_f32 it replaces.
        bf16x8_t idA[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 8; ++j) idA[hb][j] = (bf16_t)((q == (l15 >> 2) && j == 4 * hb + (l15 & 3)) ? 1.f : 0.f);
        constexpr bool RES_MFMA = HAS_RES && !ABL(4);

        int cur = 0;
        TileIter it;
        for (it.init(walk.first, walk.stride, ntx, nty); it.tile < walk.end; it.advance(walk.stride, ntx, nty)) {
            STAMP(t0);
            // epilogue operands, requested now, used after the K loop
            const int tile = it.tile, n = it.n, ty0 = it.ty * PTH, tx0 = it.tx * PTW;
            const long long tbase = a.unshuffle ? (long long)n * a.dst_nstride + pm_off(ty0 >> 1, tx0 >> 1, 0, a.Wd >> 1, 64)      // (tx0 >> 1: a multiple of 16)
                                                : (long long)n * a.dst_nstride + pm_off(ty0 * a.out_step + a.out_oyz, tx0 * a.out_step, 0, a.Wd, 64);
            // a tile inside the image (all but the last tile row / column of a ragged size) needs no per-lane bounds: straight-line
            // operand loads and epilogue (wave-uniform choice)
            const bool full = ty0 + PTH <= a.H && tx0 + PTW <= a.W;
            bool ok[4];
            constexpr bool BITS = MASK == MASK_RELU_BITS || MASK == MASK_LEAKY_BITS;
            constexpr bool LATE_MASK = HAS_RES && MASK != MASK_NONE && !BITS;   // both bf16 operands early would not fit 2 waves / SIMD
            u32x4_t rr[2][4], mm[2][4];                                // [k][nb]: 8 channels of one pixel
            u32x2_t sbits = {0u, 0u};                         // 64 sign bits of this lane's 64 outputs of the tile
            if (BITS) sbits = sbits_z[(long long)tile * 256 + w4 * 64 + lane];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ok[nb] = (tx0 + (nb & 1) * 16 + pxl < a.W) && (ty0 + w4 * 2 + (nb >> 1) < a.H);
#define CV_OPERANDS(OKN)                                                                                               \
            _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) {                                                         \
                if (OKN) {                                                                                             \
                    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                    \
                        const long long o = tbase + loff[nb] + k * 1024;                                               \
                        if (HAS_RES) rr[k][nb] = *GP(const u32x4_t, res_z + o);                                        \
                        if (MASK != MASK_NONE && !LATE_MASK && !BITS) mm[k][nb] = *GP(const u32x4_t, aux_z + o);       \
                    }                                                                                                  \
                }                                                                                                      \
            }
            if (HAS_RES || (MASK != MASK_NONE && !LATE_MASK && !BITS)) {
                if (full) { CV_OPERANDS(true) } else { CV_OPERANDS(ok[nb]) }
            }
#undef CV_OPERANDS
            STAMP(t1);

            f32x4_t acc[4][4];                                      // first written by step 0's MFMAs, whose C operand is the bias

            // ---- K loop: 18 steps s = (tap, channel half) of 16 MFMAs.  The 8 fragment reads of step s+1 are issued
            // before the MFMAs of step s, by hand: lgkmcnt(8) = "all but the 8 youngest LDS reads have returned" = step s
            // is in registers (hipcc sinks builtin LDS reads back in front of their consumers). ----
            bf16x8_t fa[2][4], fb[2][4];
            const unsigned bb = (unsigned)(W_BYTES + cur * IN_BYTES + b_lane);   // B base of this tile's buffer
#define DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define CV_LOADA(tap_, kk_, slot, mb)                                                                                  \
            if (!(ABL(2) && tap_ >= 6) && !(ABL(6) && tap_ + kk_ > 0)) DSR(fa[slot][mb], (tap_ < 6 ? a_lo[kk_] : a_hi[kk_]), (tap_ < 6 ? tap_ : tap_ - 6) * 8192 + (mb) * 2048);
#define CV_LOADB(ky_, kx_, kk_, slot, nb)                                                                              \
            if (!(ABL(0) && ky_ == 2) && !(ABL(6) && ky_ + kx_ + kk_ > 0)) DSR(fb[slot][nb], bb, (((nb) >> 1) + ky_) * (PTWH * 128) + kk_ * (4 * PTWH * 16) + (((nb) & 1) * 16 + kx_) * 16);
#define CV_LOAD(s, slot)                                                                                               \
            {                                                                                                          \
                constexpr int tap_ = (s) / 2, kk_ = (s) % 2, ky_ = tap_ / 3, kx_ = tap_ % 3;                           \
                CV_LOADA(tap_, kk_, slot, 0) CV_LOADA(tap_, kk_, slot, 1) CV_LOADA(tap_, kk_, slot, 2) CV_LOADA(tap_, kk_, slot, 3) \
                CV_LOADB(ky_, kx_, kk_, slot, 0) CV_LOADB(ky_, kx_, kk_, slot, 1) CV_LOADB(ky_, kx_, kk_, slot, 2) CV_LOADB(ky_, kx_, kk_, slot, 3) \
            }
#define CV_MFMA(s, mb, nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(s) % 2][mb], fb[(s) % 2][nb], (s) == 0 ? bvec[mb] : acc[mb][nb], 0, 0, 0);
            // one fragment read of step s+1 behind each of the first 8 MFMAs of step s: the wave's LDS issue slots sit in
            // the shadow of its own MFMAs, and the last read has 8 MFMAs (128 cycles) to return before step s+1 starts
#define CV_ML_A(s, mb, nb, lmb)                                                                                        \
            CV_MFMA(s, mb, nb)                                                                                         \
            __builtin_amdgcn_sched_barrier(0);        /* hipcc would otherwise bunch the reads behind the MFMAs */       \
            if ((s) + 1 < 18) { constexpr int t1_ = ((s) + 1) / 2, k1_ = ((s) + 1) % 2; CV_LOADA(t1_, k1_, ((s) + 1) % 2, lmb) } \
            __builtin_amdgcn_sched_barrier(0);
#define CV_ML_B(s, mb, nb, lnb)                                                                                        \
            CV_MFMA(s, mb, nb)                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((s) + 1 < 18) { constexpr int t1_ = ((s) + 1) / 2, k1_ = ((s) + 1) % 2; CV_LOADB(t1_ / 3, t1_ % 3, k1_, ((s) + 1) % 2, lnb) } \
            __builtin_amdgcn_sched_barrier(0);
#define CV_STEP(s)                                                                                                     \
            {                                                                                                          \
                if (!ABL(7)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* step s is in registers */        \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                CV_ML_A(s, 0, 0, 0) CV_ML_A(s, 0, 1, 1) CV_ML_A(s, 0, 2, 2) CV_ML_A(s, 0, 3, 3)                        \
                CV_ML_B(s, 1, 0, 0) CV_ML_B(s, 1, 1, 1) CV_ML_B(s, 1, 2, 2) CV_ML_B(s, 1, 3, 3)                        \
                CV_MFMA(s, 2, 0) CV_MFMA(s, 2, 1) CV_MFMA(s, 2, 2) CV_MFMA(s, 2, 3)                                    \
                CV_MFMA(s, 3, 0) CV_MFMA(s, 3, 1) CV_MFMA(s, 3, 2) CV_MFMA(s, 3, 3)                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of ours is outstanding
            __builtin_amdgcn_sched_barrier(0);
            CV_LOAD(0, 0)
            CV_STEP(0) CV_STEP(1) CV_STEP(2) CV_STEP(3) CV_STEP(4) CV_STEP(5) CV_STEP(6) CV_STEP(7) CV_STEP(8)
            CV_STEP(9) CV_STEP(10) CV_STEP(11) CV_STEP(12) CV_STEP(13) CV_STEP(14) CV_STEP(15) CV_STEP(16) CV_STEP(17)
#undef CV_ML_A
#undef CV_ML_B
#undef CV_STEP
#undef CV_MFMA
#undef CV_LOAD
#undef CV_LOADB
#undef CV_LOADA
#undef DSR
            STAMP(t2);

            // ---- epilogue, entirely in registers; a store covers two 256-byte runs (two chunks x 16 pixels) per wave ----
            // The residual / mask operands were requested before the K loop and have long returned, but vmcnt counts loads and
            // stores together in issue order: hipcc's per-use waits (vmcnt(3) in every ok[nb] branch) made each store wait for
            // the write acknowledgement of the store three before it -- ~5 exposed store latencies per tile (r03: 48 -> ~42 us
            // per bias+skip launch).  One explicit wait here (the builtin: hipcc's scoreboard then knows the queue is empty)
            // covers the operands; the stores after it are never waited for inside the tile.
            if (HAS_RES || MASK != MASK_NONE) __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) alone
            if constexpr (RES_MFMA) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idA[mb & 1], __builtin_bit_cast(bf16x8_t, rr[mb >> 1][nb]), acc[mb][nb], 0, 0, 0);
            }
            // Sign bits of the tile: the lane's 64 outputs are 32 packed words wd = (4k + nb) * 4 + jj (channels 2jj, 2jj+1 of the
            // piece); word wd owns bits (wd & 15) [even channel] and 16 + (wd & 15) [odd channel] of sout[wd >> 4].
            unsigned sout[2] = {0u, 0u};
            const unsigned k11 = 0x00010001u;
            auto epilogue = [&](auto FULL) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (decltype(FULL)::value || ok[nb]) {
                    auto* dst = dst_z + tbase + loff[nb];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] = acc[2 * k][nb][j]; v[4 + j] = acc[2 * k + 1][nb][j]; }
                        const unsigned wbits = k ? sbits.y : sbits.x;            // words wd = 16 k + 4 nb + jj
                        unsigned ow[4];
                        if (ABL(4)) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) ow[jj] = pk_bf16(v[2 * jj], v[2 * jj + 1]);
                        } else if (ACT == ACT_RELU && !HAS_RES && MASK == MASK_NONE) {
                            // conv1 of a ResidualConv: round, then ReLU on the packed words (bf16 bit patterns order like int16
                            // for this purpose: negative and -0 -> +0), sign bit = "the stored half is non-zero"
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                ow[jj] = pk_max_i16(pk_bf16(v[2 * jj], v[2 * jj + 1]), 0u);
                                if (!ABL(1)) sout[k] |= pk_min_u16(ow[jj], k11) << (4 * nb + jj);
                            }
                        } else if (ACT == ACT_LEAKY && !HAS_RES && MASK == MASK_NONE && !ABL(1)) {
                            // conv_last.0: LeakyReLU = max(v, slope v) (0 < slope < 1), sign bits from the packed words as in the ReLU case
                            // (bit = "the stored half is positive", what sign_bits_c64_kernel computes from a stored activation)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                ow[jj] = pk_bf16(fmaxf(v[2 * jj], v[2 * jj] * slope), fmaxf(v[2 * jj + 1], v[2 * jj + 1] * slope));
                                sout[k] |= pk_min_u16(pk_max_i16(ow[jj], 0u), k11) << (4 * nb + jj);
                            }
                        } else if (ACT == ACT_NONE && !HAS_RES && MASK == MASK_RELU_BITS) {
                            // dgrad(conv2) * ReLU': multiply the packed halves by their 0 / 1 bits
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                ow[jj] = pk_mul_lo_u16(pk_bf16(v[2 * jj], v[2 * jj + 1]), (wbits >> (4 * nb + jj)) & k11);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = p_act<ACT>(v[j], slope);
                            if (HAS_RES && !(RES_MFMA && ACT == ACT_NONE)) {
                                const unsigned rw[4] = {rr[k][nb].x, rr[k][nb].y, rr[k][nb].z, rr[k][nb].w};
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) { v[2 * jj] += bf_lo(rw[jj]); v[2 * jj + 1] += bf_hi(rw[jj]); }
                            }
                            if (ACT != ACT_NONE && !HAS_RES && !ABL(1)) {      // sign of the activation = sign of its argument (ReLU and LeakyReLU alike)
#pragma unroll
                                for (int j = 0; j < 8; ++j) sout[k] |= (v[j] > 0.f ? 1u : 0u) << (4 * nb + (j >> 1) + 16 * (j & 1));
                            }
                            if (BITS) {
                                const float neg = MASK == MASK_LEAKY_BITS ? slope : 0.f;
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] *= ((wbits >> (4 * nb + (j >> 1) + 16 * (j & 1))) & 1u) ? 1.f : neg;
                            } else if (MASK != MASK_NONE) {
                                const float neg = MASK == MASK_LEAKY ? slope : 0.f;
                                u32x4_t mq = mm[k][nb];
                                if (LATE_MASK) mq = *GP(const u32x4_t, aux_z + tbase + loff[nb] + k * 1024);
                                const unsigned mw[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    v[2 * jj] *= (bf_lo(mw[jj]) > 0.f ? 1.f : neg);
                                    v[2 * jj + 1] *= (bf_hi(mw[jj]) > 0.f ? 1.f : neg);
                                }
                            }
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) ow[jj] = pk_bf16(v[2 * jj], v[2 * jj + 1]);
                        }
                        const u32x4_t o = {ow[0], ow[1], ow[2], ow[3]};
                        if (P_NT_STORE) __builtin_nontemporal_store(o, GP(u32x4_t, dst + k * 1024));
                        else if (!ABL(5)) *GP(u32x4_t, dst + k * 1024) = o;
                        else asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                    }
                }
            }
            };
            if (full) epilogue(std::true_type{}); else epilogue(std::false_type{});
            if (ACT != ACT_NONE && !HAS_RES && sout_z && !ABL(1)) sout_z[(long long)tile * 256 + w4 * 64 + lane] = u32x2_t{sout[0], sout[1]};
            STAMP(t3);
            __syncthreads();                               // the producers' next tile has landed; everybody has finished reading `cur`
            cur ^= 1;
            STAMP(t4);
            STAMP_ADD(1, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4);
        }
    }
#ifdef VSR_ABL
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) {
        g_clk[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
        g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
#ifdef VSR_STAMPS
    st_sum[0] = stamp() - st_begin;
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 256)
        for (int k = 0; k < 8; ++k) g_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_sum[k];
#endif
}

// Sign bits of a stored bf16 activation in THIS kernel's tile order ([tile][MFMA wave][lane] x 64 bits; the lane's piece (k, nb) =
// chunk 4k + q of pixel (row 2 w4 + (nb >> 1), column 16 (nb & 1) + pxl), its packed word jj = channels 2jj, 2jj+1: bits
// 4 nb + jj and 16 + 4 nb + jj of word k): for activations produced by another kernel (the trunk stem runs on the generic
// two-source kernel) whose mask a persistent data-gradient launch needs.  66 MB read, 4 MB written at 540p.
__global__ void sign_bits_c64_kernel(const bf16_t* __restrict__ x, uint2* __restrict__ bits, int N, int H, int W) {
    const int ntx = cdiv(W, PTW), nty = cdiv(H, PTH);
    const long long total = (long long)N * ntx * nty * 256;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int lane = (int)(gid & 63), w4 = (int)((gid >> 6) & 3);
    const int tile = (int)(gid >> 8);
    const int l15 = lane & 15, q = lane >> 4;
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);
    int n, ty0, tx0;
    tile_coords(tile, ntx, nty, n, ty0, tx0);
    unsigned out[2] = {0u, 0u};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int yy = ty0 + 2 * w4 + (nb >> 1), xx = tx0 + (nb & 1) * 16 + pxl;
        if (yy < H && xx < W) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint4 v = *reinterpret_cast<const uint4*>(x + (long long)n * pm_image_elems(H, W, 64) + pm_off(yy, xx, 4 * k + q, W, 64));
                const unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    out[k] |= ((bf_lo(vw[jj]) > 0.f ? 1u : 0u) | (bf_hi(vw[jj]) > 0.f ? 0x10000u : 0u)) << (4 * nb + jj);
            }
        }
    }
    bits[gid] = make_uint2(out[0], out[1]);
}

template <int ACT, bool HAS_RES, int MASK>
static int launch_persist(const ConvArgs& a, int num_cus, hipStream_t st) {
    auto kern = conv3x3_c64_persist_kernel<ACT, HAS_RES, MASK>;
    static VsrDevOnce once;                                 // one per instantiation, remembered per device
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), P_LDS); if (rc != VSR_OK) return rc; }
    const int tiles = a.N * cdiv(a.W, PTW) * cdiv(a.H, PTH);
    int gx = num_cus / a.nz;
    gx &= ~7;                                               // whole XCD groups (xcd_tile_walk)
    if (gx < 8) gx = num_cus / a.nz;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    hipLaunchKernelGGL(kern, dim3(gx, a.nz), dim3(PNT), P_LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

#ifdef VSR_ABL
extern "C" int vsr_debug_read_clk(unsigned long long* host_out) {      // [256 workgroups][cycles, 100 MHz ticks, -, -] of the last launch
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clk), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -3;
}
#endif
#ifdef VSR_STAMPS
extern "C" int vsr_debug_read_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -3;
}
#endif

// Eligibility is decided by the dispatcher in conv_mfma.hip (bf16, 3x3, one 64-channel source at unit
// step, 64 output channels, blocked pixel-major destination).  Returns VSR_ERR_UNSUPPORTED for an epilogue
// combination that has no instantiation; the caller then uses the generic kernel.
int vsr_launch_conv3x3_c64_persist(const ConvArgs& a, int num_cus, hipStream_t st) {
    bool res = false, aux = false;
    for (int z = 0; z < a.nz; ++z) { res = res || a.res[z]; aux = aux || a.aux[z]; }
    for (int z = 0; z < a.nz; ++z) if ((res && !a.res[z]) || (aux && !a.aux[z])) return VSR_ERR_UNSUPPORTED;
    int mask = aux ? a.mask_mode : MASK_NONE;
    if (mask == MASK_RELU && !res) {                       // sign bits instead of the activation, if every z has them
        bool bits = true;
        for (int z = 0; z < a.nz; ++z) bits = bits && a.sign_bits[z];
        if (bits) mask = MASK_RELU_BITS;
    }
    if (mask == MASK_LEAKY && res) {                       // (dgrad + dX) * LeakyReLU': the mask from sign bits keeps the residual prefetch
        bool bits = true;
        for (int z = 0; z < a.nz; ++z) bits = bits && a.sign_bits[z];
        if (bits) mask = MASK_LEAKY_BITS;
    }
    if (pm_image_elems((PTHH + 2) * a.in_step, a.Ws, 64) * 2 > 0x7fffffffLL || pm_image_elems(2 * PTH + 2, a.Wd, 64) > 0x7fffffffLL)
        return VSR_ERR_UNSUPPORTED;                                                  // in-tile offsets are 32-bit
    if (a.unshuffle) {                                         // phase-separated destination: plain stride-1 launches without an activation-mask operand
        if (a.nz != 1 || a.out_step != 1 || a.out_ox[0] || a.out_oy[0] || (a.Hd & 1) || (a.Wd & 1) || a.Hd != a.H || a.Wd != a.W || aux ||
            a.unshuffle_plane < (long long)a.N * pm_image_elems(a.Hd / 2, a.Wd / 2, 64) || a.dst_nstride != pm_image_elems(a.Hd / 2, a.Wd / 2, 64) ||
            3 * a.unshuffle_plane + pm_image_elems(PTH + 2, a.Wd / 2, 64) > 0x7fffffffLL)
            return VSR_ERR_UNSUPPORTED;
    }
#define PERSIST_CASE(ACT, RES, MASK) if (a.act == ACT && res == RES && mask == MASK) return launch_persist<ACT, RES, MASK>(a, num_cus, st);
    PERSIST_CASE(ACT_RELU, false, MASK_NONE)     // conv1 of a ResidualConv
    PERSIST_CASE(ACT_NONE, true, MASK_NONE)      // conv2 + skip ; dgrad(conv1) + dX
    PERSIST_CASE(ACT_LEAKY, false, MASK_NONE)    // conv_last.0
    PERSIST_CASE(ACT_NONE, false, MASK_NONE)     // upsample phases, plain dgrads
    PERSIST_CASE(ACT_NONE, false, MASK_RELU)     // dgrad(conv2) * ReLU'
    PERSIST_CASE(ACT_NONE, false, MASK_RELU_BITS)   // the same, from the sign bits the bias+ReLU launch left (4 MB instead of 66 MB)
    PERSIST_CASE(ACT_NONE, true, MASK_LEAKY)     // (dgrad(conv1 of block 0) + dX) * LeakyReLU' of the stem
    PERSIST_CASE(ACT_NONE, true, MASK_LEAKY_BITS)   // the same with the stem's sign bits (vsr_launch_sign_bits_c64): no late 66 MB mask read
    PERSIST_CASE(ACT_NONE, false, MASK_LEAKY)    // dgrad * LeakyReLU' (discriminator conv_7 / conv_8)
#undef PERSIST_CASE
    return VSR_ERR_UNSUPPORTED;
}

// bits: N * ceil(H/8) * ceil(W/32) * 256 uint2 (the size the engine reserves for a persistent launch's sign_out)
int vsr_launch_sign_bits_c64(const void* x_pm, void* bits, int N, int H, int W, hipStream_t st) {
    if (!x_pm || !bits || N < 1 || H < 1 || W < 1) return VSR_ERR_BADARG;
    const long long total = (long long)N * cdiv(W, PTW) * cdiv(H, PTH) * 256;
    hipLaunchKernelGGL(sign_bits_c64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x_pm, (uint2*)bits, N, H, W);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}
