// SPyNet's 7x7 layers (RealBasicVSR/modules/spynet.py:16-18: 8->32, 32->64, 64->32, 32->16, 16->2 channels), bf16 in / fp32
// accumulate, as persistent producer / consumer kernels in the style of conv3x3_persist.hip.
//
// The generic tiled kernel (conv_mfma.hip) ran these layers at 0.54-0.64 PFLOP/s: it re-fetches a [cout][cin] weight slab per
// TAP (49 barrier pairs per tile, the slab one tap = ~256 MFMA cycles ahead of its use, less than an L2 round trip) and pads
// 8 / 16 input channels to a K of 16 per tap.  Here:
//   * one 512-thread workgroup per CU, persistent over 8x32-pixel tiles: 4 MFMA waves (2 tile rows x 32 pixels x every output
//     channel each, v_mfma_f32_16x16x32_bf16) + 4 LDS-DMA producer waves.
//   * the haloed 14x38-pixel input tile is double-buffered in LDS in the blocked layout's own order ([row][chunk][38 px][16 B],
//     as in the 3x3 kernel) and filled one tile ahead.  A 64-channel input is walked as two PASSES of 32 channels (34 KB per
//     pass instead of 68 KB per tile), so that two buffers and the weight ring fit the 160 KiB.
//   * the weights are STREAMED from L2 one kernel row (7 taps) at a time into a ring of three LDS slots (32->64: 28.7 KB per
//     row, 200 KB per tile), two rows ahead of their use (a row is 0.9-1.8 k MFMA cycles); one workgroup barrier per kernel
//     row.  Layers whose 49 taps fit (8->32, 32->16, 16->2: 28-57 KB) keep them resident and synchronise once per tile.
//     The ring's image is the A-fragment order itself: one 1 KiB DMA piece = [k group q][16 rows][16 B] of one (K step, row
//     block), gathered by per-lane source addresses from the packed [tap][cout][cin] weights (pack_weights_kernel).
//   * a K step is one MFMA K of 32: (tap, 32 channels) for 32 / 64 input channels; for the 16-channel layers a PAIR of taps
//     (k groups 0-1 = tap kx, 2-3 = tap kx + 1; the 8th tap of a row has zero weights), 28 instead of 49 steps.
//   * epilogue as in the 3x3 kernel: bias is the accumulators' initial value, ReLU on the packed bf16 words, 16-byte stores in
//     paired-block channel order (pm_acc_chan); the 2-channel flow layer stores planar fp32 (+ the upsampled flow, spynet.py:65).
#include "common.h"

namespace {

constexpr int QTW = 32, QTH = 8, QNT = 512, QHALO = 3;
constexpr int QTHH = QTH + 2 * QHALO;                  // 14 haloed rows

typedef __attribute__((ext_vector_type(2))) __bf16 qbf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned qu32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned qu32x2_t;

__device__ uint4 g_c7_zero_chunk[2];

// Diagnostic build only (make ABL=<bits> ABLSRC=conv7x7_persist): results are WRONG by construction, only run time and the
// in-kernel clock are read.  bit 0: the producers stream no weight rows after the prologue; bit 1: no fragment reads after a
// kernel row's first step (bare MFMA loop); bit 2: no epilogue; bit 3: the producers load only the first tile.
#ifdef VSR_ABL
__device__ unsigned long long g_clk7[256 * 4];
#define QABL(bit) ((VSR_ABL >> (bit)) & 1)
#else
#define QABL(bit) 0
#endif

__device__ __forceinline__ unsigned q_pk_bf16(float a, float b) {
    qbf16x2_t p = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned q_pk_max_i16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned q_pk_min_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned q_pk_mul_lo_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// ReLU' of a stored bf16 activation applied to a packed pair: both halves times (activation > 0) (bf16 patterns order like int16
// for this purpose, see conv3x3_persist.hip)
__device__ __forceinline__ unsigned q_relu_mask(unsigned v, unsigned act) { return q_pk_mul_lo_u16(v, q_pk_min_u16(q_pk_max_i16(act, 0u), 0x00010001u)); }

#define QGLDS16(src, dst)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define QGP(T, x) ((__attribute__((address_space(1))) T*)(x))

// vmcnt(n) for a wave-uniform run-time n (the counter's immediate is part of the instruction)
__device__ __forceinline__ void q_wait_vm(int n) {
    switch (n) {
#define QW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        QW(0) QW(1) QW(2) QW(3) QW(4) QW(5) QW(6) QW(7) QW(8) QW(9) QW(10) QW(11)
#undef QW
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

struct QTileIter {            // as TileIter of conv3x3_persist.hip
    int tile, n, ty, tx, sn, sty, stx;
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

// CIN: channels per source pixel (16 | 32 | 64); NB: 16-row blocks of output channels (1 | 2 | 4)
template <int CIN, int NB> struct C7 {
    static constexpr int PASSES = CIN == 64 ? 2 : 1;
    static constexpr int CP = CIN == 16 ? 2 : 4;             // 16-byte chunks per pixel of a pass's LDS tile
    static constexpr int CG = CIN / 8;                        // chunks per pixel in HBM
    static constexpr int TWHP = CIN == 16 ? 40 : 38;          // pixels per LDS tile row (the tap-pair layers read one pixel past the halo: zero-filled)
    static constexpr int KROW = CIN == 16 ? 4 : 7;            // K steps per kernel row
    static constexpr int ROWPITCH = CP * TWHP * 16;
    static constexpr int TILE_CHUNKS = QTHH * CP * TWHP;
    static constexpr int TILE_BYTES = TILE_CHUNKS * 16;
    static constexpr int NPIECE_T = (TILE_CHUNKS + 63) / 64, NPIECE_TW = (NPIECE_T + 3) / 4;
    static constexpr int ROW_PIECES = KROW * NB, NPIECE_RW = (ROW_PIECES + 3) / 4;
    static constexpr int SLOT_BYTES = ROW_PIECES * 1024;
    static constexpr int WROWS = 7 * PASSES;                  // weight rows of the stream's period
    static constexpr bool STREAM = WROWS * SLOT_BYTES > 64 * 1024;
    static constexpr int NSLOT = STREAM ? 3 : WROWS;
    static constexpr int NTB = STREAM ? 2 : 3;                // tile buffers: the resident layers' passes are shorter than an HBM round trip, two tiles in flight
    static constexpr int T_OFF = NSLOT * SLOT_BYTES;
    static constexpr int BIAS_OFF = T_OFF + NTB * TILE_BYTES;
    static constexpr int LDS = BIAS_OFF + 256;
    static_assert(LDS <= 160 * 1024, "does not fit the LDS of a CU");
    static_assert(NPIECE_RW + (NPIECE_TW + 4) / 5 <= 12 && NPIECE_TW <= 12, "q_wait_vm covers 0..12");
};

template <int NB> __device__ __forceinline__ int c7_chan(int mb, int row) { return NB == 1 ? row : pm_acc_chan(mb, row); }

template <int CIN, int NB, int EPI>
__global__ __launch_bounds__(QNT, 1) void conv7x7_persist_kernel(const ConvArgs ka, const int cop) {
    typedef C7<CIN, NB> K;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    struct {
        int N, H, W, CD, act, cout_real, cop;
        long long src_nstride, dst_nstride;
        unsigned long long src, wpack, bias, dst, pres, aux;
    } a = {ka.N, ka.H, ka.W, ka.CD, ka.act, ka.cout_real, cop, ka.src_nstride[0], ka.dst_nstride,
           (unsigned long long)ka.src[0], (unsigned long long)ka.wpack, (unsigned long long)ka.bias, (unsigned long long)ka.dst[0],
           (unsigned long long)ka.pres, (unsigned long long)ka.aux[0]};
    asm volatile("" : "+s"(a.N), "+s"(a.H), "+s"(a.W), "+s"(a.CD), "+s"(a.act), "+s"(a.cout_real), "+s"(a.cop), "+s"(a.src_nstride), "+s"(a.dst_nstride));
    asm volatile("" : "+s"(a.src), "+s"(a.wpack), "+s"(a.bias), "+s"(a.dst), "+s"(a.pres), "+s"(a.aux));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);            // 0: MFMA + epilogue, 1: LDS-DMA producer
    const int w4 = wave & 3;
    const int l15 = lane & 15, q = lane >> 4;
#ifdef VSR_ABL
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (tid == 0) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // pixel of a 16-pixel block that lane column l15 works on (bank-conflict-free B reads for any even row pitch: see the 3x3 kernel)
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);

    const int ntx = cdiv(a.W, QTW), nty = cdiv(a.H, QTH);
    const int total = a.N * ntx * nty;
    const int WS = pm_ws(a.W);
    const TileWalk walk = xcd_tile_walk(total, blockIdx.x, gridDim.x);
    const int my_tiles = walk.first < walk.end ? (walk.end - 1 - walk.first) / walk.stride + 1 : 0;
    const int my_passes = my_tiles * K::PASSES;

    if (tid < NB * 16) reinterpret_cast<float*>(smem + K::BIAS_OFF)[tid] = a.bias ? QGP(const float, a.bias)[c7_chan<NB>(tid >> 4, tid & 15)] : 0.f;

    if (role == 1) {
        // =================== producer waves ===================
        const auto* src = QGP(const char, a.src);
        const auto* wsrc = QGP(const char, a.wpack);
        const auto* zsrc = QGP(const char, g_c7_zero_chunk);
        // tile pieces of this wave: piece = w4 + 4 i; LDS slot idx = [row ty][chunk c][TWHP pixels tx]
        int rel[K::NPIECE_TW];
#pragma unroll
        for (int i = 0; i < K::NPIECE_TW; ++i) {
            const int idx = (w4 + 4 * i) * 64 + lane;
            const int ty = idx / (K::CP * K::TWHP), rem = idx - ty * (K::CP * K::TWHP);
            const int c = rem / K::TWHP, tx = rem - c * K::TWHP;
            const int dx = tx - QHALO;
            rel[i] = (((((ty - QHALO) * WS + (dx >> 5)) * K::CG + c) * 256 + (dx & 31) * 8) * 2);
        }
        // weight pieces of this wave within a kernel row: piece p = w4 + 4 i = (K step s, row block mb) = s * NB + mb;
        // lane (q, l15) = k group q of row l15: byte offset from the row's first tap (wrel < 0: zero weights)
        int wrel[K::NPIECE_RW];
#pragma unroll
        for (int i = 0; i < K::NPIECE_RW; ++i) {
            const int p = w4 + 4 * i, s = p / NB, mb = p - s * NB;
            const int row = c7_chan<NB>(mb, l15);
            if (CIN == 16) {
                const int kx = 2 * s + (q >> 1);
                wrel[i] = kx < 7 ? ((kx * a.cop + row) * 16 + 8 * (q & 1)) * 2 : -1;
            } else {
                wrel[i] = ((s * a.cop + row) * CIN + 8 * q) * 2;
            }
        }
        const int my_row_pieces = (K::ROW_PIECES - w4 + 3) / 4;            // pieces this wave issues per weight row
        // pieces i0 <= i < i1 of this wave's share of a tile; returns how many LDS-DMA instructions that were (wave-uniform)
        auto issue_tile = [&](const QTileIter& it, int hf, int buf, int i0, int i1) -> int {
            const int ty0 = it.ty * QTH, tx0 = it.tx * QTW;
            const auto* org = src + ((long long)it.n * a.src_nstride + pm_off(ty0, tx0, hf * 4, a.W, CIN)) * 2;
            char* dstb = smem + K::T_OFF + buf * K::TILE_BYTES;
            if (ty0 >= QHALO && ty0 + QTH + QHALO <= a.H && tx0 >= QHALO && tx0 + QTW + QHALO <= a.W && K::TWHP == 38) {
#pragma unroll
                for (int i = 0; i < K::NPIECE_TW; ++i) {
                    const int piece = w4 + 4 * i;
                    if (i >= i0 && i < i1 && piece < K::NPIECE_T && piece * 64 + lane < K::TILE_CHUNKS) QGLDS16(org + rel[i], dstb + piece * 1024);
                }
            } else {
#pragma unroll
                for (int i = 0; i < K::NPIECE_TW; ++i) {
                    const int piece = w4 + 4 * i;
                    const int idx = piece * 64 + lane;
                    const int ty = idx / (K::CP * K::TWHP), rem = idx - ty * (K::CP * K::TWHP);
                    const int tx = rem % K::TWHP;
                    const int vy = ty0 + ty - QHALO, vx = tx0 + tx - QHALO;
                    const auto* s = (vy >= 0 && vy < a.H && vx >= 0 && vx < a.W && tx < QTW + 2 * QHALO) ? org + rel[i] : zsrc;
                    if (i >= i0 && i < i1 && piece < K::NPIECE_T && idx < K::TILE_CHUNKS) QGLDS16(s, dstb + piece * 1024);
                }
            }
            const int mine = (K::NPIECE_T - w4 + 3) / 4;       // this wave's pieces of a tile: i < mine
            const int hi = i1 < mine ? i1 : mine;
            return hi > i0 ? hi - i0 : 0;
        };
        auto issue_wrow = [&](int gr, int slot) {              // weight row gr of the period (half hf = gr / 7, kernel row ky = gr % 7) into ring slot `slot`
            const int hf = gr >= 7 ? 1 : 0, ky = gr - 7 * hf;
            const auto* org = wsrc + ((long long)ky * 7 * a.cop * CIN + hf * 32) * 2;
            char* dstb = smem + slot * K::SLOT_BYTES;
#pragma unroll
            for (int i = 0; i < K::NPIECE_RW; ++i) {
                const int p = w4 + 4 * i;
                const auto* s = wrel[i] >= 0 ? org + wrel[i] : zsrc;
                if (p < K::ROW_PIECES) QGLDS16(s, dstb + p * 1024);
            }
        };
        QTileIter it;
        it.init(walk.first, walk.stride, ntx, nty);
        if (my_passes > 0) issue_tile(it, 0, 0, 0, K::NPIECE_TW);
        if (K::STREAM) {
            // interval g = the consumers work on weight row g: issue row g + 2 into the slot row g - 1 left, then a slice of the
            // NEXT pass's tile (spread over the pass's first five intervals); the counted wait leaves exactly what this interval
            // issued in flight, i.e. row g + 1 and every older tile slice have landed -- nothing is waited for in the interval it
            // was requested in (r03: the whole tile ahead of the row at the pass's first interval exposed one HBM round trip per pass)
            constexpr int SL = (K::NPIECE_TW + 4) / 5;
            const int nrows = my_passes * 7;                   // weight rows this workgroup consumes
            if (nrows > 0) issue_wrow(0, 0);
            if (nrows > 1) issue_wrow(1, 1);
            q_wait_vm(nrows > 1 ? my_row_pieces : 0);          // the tile and row 0 have landed
            __builtin_amdgcn_s_barrier();                      // (not __syncthreads: its fence would drain vmcnt, i.e. the row just issued)
            int g = 0, gr2 = 2 % K::WROWS, sl2 = 2;             // gr2 = (g + 2) mod the period, sl2 = (g + 2) mod the ring
            for (int pass = 0; pass < my_passes; ++pass) {
                const int hf = K::PASSES == 2 ? (pass & 1) : 0;
                const bool next = pass + 1 < my_passes && !QABL(3);
                if (pass + 1 < my_passes && (K::PASSES == 1 || hf == 1)) it.advance(walk.stride, ntx, nty);
                for (int ky = 0; ky < 7; ++ky) {
                    const bool more = g + 2 < nrows && !QABL(0);
                    int inflight = 0;
                    if (more) { issue_wrow(gr2, sl2); inflight = my_row_pieces; }
                    if (next && ky < 5) inflight += issue_tile(it, K::PASSES == 2 ? (hf ^ 1) : 0, (pass + 1) & 1, ky * SL, ky * SL + SL);
                    q_wait_vm(inflight);
                    __builtin_amdgcn_s_barrier();
                    ++g;
                    gr2 = gr2 + 1 == K::WROWS ? 0 : gr2 + 1;
                    sl2 = sl2 + 1 == K::NSLOT ? 0 : sl2 + 1;
                }
            }
        } else {
            // resident weights: all of them now, then two tiles ahead of the consumers (three tile buffers)
#pragma unroll 1
            for (int r = 0; r < K::WROWS; ++r) issue_wrow(r, r);
            int inflight = 0;
            if (my_passes > 1) { it.advance(walk.stride, ntx, nty); inflight = issue_tile(it, 0, 1, 0, K::NPIECE_TW); }
            q_wait_vm(inflight);                               // the weights and the first tile have landed
            __builtin_amdgcn_s_barrier();
            for (int pass = 0; pass < my_passes; ++pass) {
                inflight = 0;
                if (pass + 2 < my_passes && !QABL(3)) {
                    it.advance(walk.stride, ntx, nty);
                    inflight = issue_tile(it, 0, (pass + 2) % K::NTB, 0, K::NPIECE_TW);
                }
                q_wait_vm(inflight);                           // tile pass + 1 has landed
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        // =================== MFMA waves: tile rows 2 w4, 2 w4 + 1 ===================
        const unsigned a_lane = (unsigned)(q * 256 + l15 * 16);
        const int bq = CIN == 16 ? (q & 1) : q, bt = CIN == 16 ? (q >> 1) : 0;
        const unsigned b_lane = (unsigned)(w4 * 2 * K::ROWPITCH + bq * K::TWHP * 16 + (pxl + bt) * 16);
        const int CDc = a.CD >> 3;
        int loff[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int dx = (nb & 1) * 16 + pxl;
            if (NB == 1) loff[nb] = (((w4 * 2 + (nb >> 1)) * WS) * CDc + (q >> 1)) * 256 + dx * 8 + (q & 1) * 4;
            else loff[nb] = (((w4 * 2 + (nb >> 1)) * WS) * CDc + q) * 256 + dx * 8;
        }
        __syncthreads();                                       // bias, the first tile and the first weight row(s) are in LDS
        f32x4_t bvec[NB];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) bvec[mb] = *reinterpret_cast<const f32x4_t*>(smem + K::BIAS_OFF + (mb * 16 + 4 * q) * 4);
        auto* const dstp = QGP(bf16_t, a.dst);
        const auto* const auxp = QGP(const bf16_t, a.aux);     // MASK_RELU source (destination layout): the data gradients of train_flow
        auto* const dpl = QGP(float, a.dst);
        const auto* const ppl = QGP(const float, a.pres);
        const long long plane = (long long)a.H * a.W;

        // K loop.  A "sequence" is TS K steps with compile-time fragment addresses: one kernel row (7 or 4 steps) between two
        // barriers for the streamed layers, the whole tile (49 or 28 steps) for the resident ones.  Fragment reads run D steps
        // ahead of their MFMAs (NF = D + 1 register sets, counted lgkmcnt: the LDS returns in order): a step of a 16-row layer is
        // 4 MFMAs = 64 cycles, less than an LDS round trip (r03: one step ahead ran 32->16 at 175 cycles per step).
        constexpr int TS = K::STREAM ? K::KROW : 7 * K::KROW;
        constexpr int D = NB == 1 ? 3 : (NB == 2 ? 2 : 1);
        constexpr int NF = D + 1, NL = NB + 4, NM = NB * 4;
        static_assert((D - 1) * NL <= 15, "lgkmcnt is a 4-bit counter");
        f32x4_t acc[NB][4];
        bf16x8_t fa[NF][NB], fb[NF][4];
#define DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
        // fragment read l of step t_: l < 4: B block l (tile row (l >> 1) + ky, pixels 16 (l & 1) + kx ..), else A block l - 4
#define QX_L(t_, l_)                                                                                                   \
        if constexpr ((t_) < TS && (l_) < NL && !(QABL(1) && (t_) % K::KROW > 0)) {                                     \
            constexpr int ky_ = (t_) / K::KROW, s_ = (t_) % K::KROW;                                                   \
            if constexpr ((l_) < 4) { DSR(fb[(t_) % NF][(l_) & 3], bb, (ky_ + ((l_) >> 1)) * K::ROWPITCH + (((l_) & 1) * 16 + (CIN == 16 ? 2 * s_ : s_)) * 16); } \
            else { DSR(fa[(t_) % NF][((l_) - 4) & (NB - 1)], ab, (t_) * NB * 1024 + ((l_) - 4) * 1024); }               \
        }
#define QX_M(t_, m_) if constexpr ((m_) < NM) {                                                                        \
            acc[((m_) >> 2) & (NB - 1)][(m_) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(t_) % NF][((m_) >> 2) & (NB - 1)], fb[(t_) % NF][(m_) & 3], acc[((m_) >> 2) & (NB - 1)][(m_) & 3], 0, 0, 0); }
#define Q_SB __builtin_amdgcn_sched_barrier(0);
#define QX_ML(t_, i_) QX_M(t_, i_) Q_SB QX_L((t_) + D, i_) Q_SB
#define QX_STEP(t_)                                                                                                    \
        if constexpr ((t_) < TS) {                                                                                     \
            constexpr int last_ = (t_) + D - 1 < TS - 1 ? (t_) + D - 1 : TS - 1;                                        \
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"((last_ - (t_)) * NL) : "memory"); Q_SB                          \
            QX_ML(t_, 0) QX_ML(t_, 1) QX_ML(t_, 2) QX_ML(t_, 3) QX_ML(t_, 4) QX_ML(t_, 5) QX_ML(t_, 6) QX_ML(t_, 7)     \
            QX_M(t_, 8) QX_M(t_, 9) QX_M(t_, 10) QX_M(t_, 11) QX_M(t_, 12) QX_M(t_, 13) QX_M(t_, 14) QX_M(t_, 15) Q_SB \
        }
#define QX_PRO(t_) if constexpr ((t_) < D) { QX_L(t_, 0) QX_L(t_, 1) QX_L(t_, 2) QX_L(t_, 3) QX_L(t_, 4) QX_L(t_, 5) QX_L(t_, 6) QX_L(t_, 7) }
#define QX_SEQ                                                                                                         \
        Q_SB QX_PRO(0) QX_PRO(1) QX_PRO(2)                                                                             \
        QX_STEP(0) QX_STEP(1) QX_STEP(2) QX_STEP(3) QX_STEP(4) QX_STEP(5) QX_STEP(6) QX_STEP(7) QX_STEP(8) QX_STEP(9)   \
        QX_STEP(10) QX_STEP(11) QX_STEP(12) QX_STEP(13) QX_STEP(14) QX_STEP(15) QX_STEP(16) QX_STEP(17) QX_STEP(18)    \
        QX_STEP(19) QX_STEP(20) QX_STEP(21) QX_STEP(22) QX_STEP(23) QX_STEP(24) QX_STEP(25) QX_STEP(26) QX_STEP(27)    \
        QX_STEP(28) QX_STEP(29) QX_STEP(30) QX_STEP(31) QX_STEP(32) QX_STEP(33) QX_STEP(34) QX_STEP(35) QX_STEP(36)    \
        QX_STEP(37) QX_STEP(38) QX_STEP(39) QX_STEP(40) QX_STEP(41) QX_STEP(42) QX_STEP(43) QX_STEP(44) QX_STEP(45)    \
        QX_STEP(46) QX_STEP(47) QX_STEP(48)
        QTileIter it;
        it.init(walk.first, walk.stride, ntx, nty);
        int slot = 0;                                          // STREAM: ring slot of the current weight row
        for (int pass = 0; pass < my_passes; ++pass) {
            const int hf = K::PASSES == 2 ? (pass & 1) : 0;
            const int ty0 = it.ty * QTH, tx0 = it.tx * QTW;
            const bool full = ty0 + QTH <= a.H && tx0 + QTW <= a.W;      // no per-lane bounds in the epilogue (wave-uniform)
            float pr[4][4];                                    // planar epilogue: the values added to the output (flow_up), requested now
            if (EPI == EPI_PLANAR && a.pres && q == 0) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int vx = tx0 + (nb & 1) * 16 + pxl, vy = ty0 + w4 * 2 + (nb >> 1);
                    const long long o = (long long)it.n * a.dst_nstride + (long long)vy * a.W + vx;
#pragma unroll
                    for (int c = 0; c < 4; ++c) pr[nb][c] = (c < a.cout_real && vx < a.W && vy < a.H) ? ppl[o + c * plane] : 0.f;
                }
            }
            qu32x4_t mm[NB > 1 ? NB / 2 : 1][4];                // mask operand pieces of this tile, requested before the K loop
            if (hf == 0) {
#pragma unroll
                for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = bvec[mb];
            }
            if (EPI == EPI_NHWC && a.aux && (K::PASSES == 1 || hf == 1)) {
                const long long tb = (long long)it.n * a.dst_nstride + pm_off(ty0, tx0, 0, a.W, a.CD);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const bool okn = (tx0 + (nb & 1) * 16 + pxl < a.W) && (ty0 + w4 * 2 + (nb >> 1) < a.H);
                    if constexpr (NB == 1) {
                        qu32x2_t t2 = {0u, 0u};
                        if (okn && (q >> 1) < CDc) t2 = *QGP(const qu32x2_t, auxp + tb + loff[nb]);
                        mm[0][nb] = qu32x4_t{t2.x, t2.y, 0u, 0u};
                    } else {
#pragma unroll
                        for (int k = 0; k < NB / 2; ++k) {
                            mm[k][nb] = qu32x4_t{0u, 0u, 0u, 0u};
                            if (okn && 4 * k + q < CDc) mm[k][nb] = *QGP(const qu32x4_t, auxp + tb + loff[nb] + k * 1024);
                        }
                    }
                }
            }
            unsigned bb = (unsigned)(K::T_OFF + (pass % K::NTB) * K::TILE_BYTES) + b_lane;
            if constexpr (K::STREAM) {
#pragma unroll 1
                for (int ky = 0; ky < 7; ++ky) {
                    const unsigned ab = (unsigned)(slot * K::SLOT_BYTES) + a_lane;
                    QX_SEQ
                    bb += K::ROWPITCH;
                    slot = slot + 1 == K::NSLOT ? 0 : slot + 1;
                    __builtin_amdgcn_s_barrier();              // everybody has finished this row's slot; the next row has landed
                }
            } else {
                const unsigned ab = a_lane;
                QX_SEQ
            }
            if (K::PASSES == 1 || hf == 1) {
                // ---- epilogue ----
                if (QABL(2)) {
#pragma unroll
                    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) asm volatile("" :: "v"(acc[mb][nb]));
                } else if (EPI == EPI_NHWC) {
                    const long long tbase = (long long)it.n * a.dst_nstride + pm_off(ty0, tx0, 0, a.W, a.CD);
                    const bool relu = a.act == ACT_RELU;
#define QX_EPI(OKN)                                                                                                    \
                    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) {                                                 \
                        if (OKN) {                                                                                     \
                            auto* d = dstp + tbase + loff[nb];                                                         \
                            if constexpr (NB == 1) {                                                                   \
                                unsigned o0 = q_pk_bf16(acc[0][nb][0], acc[0][nb][1]), o1 = q_pk_bf16(acc[0][nb][2], acc[0][nb][3]); \
                                if (relu) { o0 = q_pk_max_i16(o0, 0u); o1 = q_pk_max_i16(o1, 0u); }                    \
                                if (a.aux) { o0 = q_relu_mask(o0, mm[0][nb].x); o1 = q_relu_mask(o1, mm[0][nb].y); }    \
                                if ((q >> 1) < CDc) *QGP(qu32x2_t, d) = qu32x2_t{o0, o1};                              \
                            } else {                                                                                   \
                                _Pragma("unroll") for (int k = 0; k < NB / 2; ++k) {                                   \
                                    unsigned ow[4];                                                                    \
                                    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                 \
                                        ow[jj] = q_pk_bf16(acc[2 * k][nb][2 * jj], acc[2 * k][nb][2 * jj + 1]);        \
                                        ow[2 + jj] = q_pk_bf16(acc[(2 * k + 1) & (NB - 1)][nb][2 * jj], acc[(2 * k + 1) & (NB - 1)][nb][2 * jj + 1]); \
                                    }                                                                                  \
                                    if (relu) { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) ow[jj] = q_pk_max_i16(ow[jj], 0u); } \
                                    if (a.aux) { ow[0] = q_relu_mask(ow[0], mm[k][nb].x); ow[1] = q_relu_mask(ow[1], mm[k][nb].y);  \
                                                 ow[2] = q_relu_mask(ow[2], mm[k][nb].z); ow[3] = q_relu_mask(ow[3], mm[k][nb].w); } \
                                    if (4 * k + q < CDc) *QGP(qu32x4_t, d + k * 1024) = qu32x4_t{ow[0], ow[1], ow[2], ow[3]}; \
                                }                                                                                      \
                            }                                                                                          \
                        }                                                                                              \
                    }
                    if (full) { QX_EPI(true) } else { QX_EPI((tx0 + (nb & 1) * 16 + pxl < a.W) && (ty0 + w4 * 2 + (nb >> 1) < a.H)) }
#undef QX_EPI
                } else {
                    // planar fp32 destination [N][cout_real][H][W], cout_real <= 4: rows 0..3 of block 0 sit in the q = 0 lanes
                    if (q == 0) {
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            const int vx = tx0 + (nb & 1) * 16 + pxl, vy = ty0 + w4 * 2 + (nb >> 1);
                            if (vx >= a.W || vy >= a.H) continue;
                            const long long o = (long long)it.n * a.dst_nstride + (long long)vy * a.W + vx;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (c >= a.cout_real) break;
                                float v = acc[0][nb][c];
                                if (a.act == ACT_RELU) v = v > 0.f ? v : 0.f;
                                if (a.pres) v += pr[nb][c];
                                dpl[o + c * plane] = v;
                            }
                        }
                    }
                }
                it.advance(walk.stride, ntx, nty);
            }
            if (!K::STREAM) __builtin_amdgcn_s_barrier();      // the producers' next tile has landed; everybody has finished this one (no fence: the epilogue's stores stay in flight)
        }
#undef QX_SEQ
#undef QX_PRO
#undef QX_STEP
#undef QX_ML
#undef Q_SB
#undef QX_M
#undef QX_L
#undef DSR
    }
#ifdef VSR_ABL
    if (tid == 0 && blockIdx.x < 256) {
        g_clk7[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
        g_clk7[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
}

template <int CIN, int NB, int EPI>
int launch_c7(const ConvArgs& a, int cop, int num_cus, hipStream_t st) {
    typedef C7<CIN, NB> K;
    auto kern = conv7x7_persist_kernel<CIN, NB, EPI>;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), K::LDS); if (rc != VSR_OK) return rc; }
    const int tiles = a.N * cdiv(a.W, QTW) * cdiv(a.H, QTH);
    int gx = num_cus & ~7;
    if (gx < 8) gx = num_cus;
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(QNT), K::LDS, st, a, cop);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

#ifdef VSR_ABL
extern "C" int vsr_debug_read_clk7(unsigned long long* host_out) {     // [256 workgroups][cycles, 100 MHz ticks, -, -] of the last launch
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clk7), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -3;
}
#endif

// Eligibility (bf16, 7x7, one pixel-major source at unit step, no residual operand; mask: ReLU' of a stored activation) is checked here; the caller falls
// back to the generic kernel on VSR_ERR_UNSUPPORTED.  cin: channels per source pixel; cop: rows per tap of the packed weights.
int vsr_launch_conv7x7_persist(const ConvArgs& a, int cin, int cop, int epi, int num_cus, hipStream_t st) {
    if (a.nz != 1 || a.in_step != 1 || a.src_oy[0] != 0 || a.src_ox[0] != 0 || a.Hs != a.H || a.Ws != a.W || !a.src[0]) return VSR_ERR_UNSUPPORTED;
    if (a.out_step != 1 || a.out_oy[0] != 0 || a.out_ox[0] != 0 || a.Hd != a.H || a.Wd != a.W) return VSR_ERR_UNSUPPORTED;
    if (a.res[0] || a.base_lr || (a.act != ACT_NONE && a.act != ACT_RELU) || a.cout_real < 1) return VSR_ERR_UNSUPPORTED;
    if (a.aux[0] && (a.mask_mode != MASK_RELU || epi != EPI_NHWC)) return VSR_ERR_UNSUPPORTED;
    if (pm_image_elems(QTHH + 2, a.W, cin) * 2 > 0x7fffffffLL || (long long)49 * cop * cin * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
    if (epi == EPI_PLANAR) {
        if (a.cout_real > 4 || cop < 16 || cin != 16) return VSR_ERR_UNSUPPORTED;
        return launch_c7<16, 1, EPI_PLANAR>(a, cop, num_cus, st);
    }
    if (a.pres) return VSR_ERR_UNSUPPORTED;
    const int nb = a.cout_real <= 16 ? 1 : (a.cout_real <= 32 ? 2 : 4);
    if (a.cout_real > 64 || nb * 16 > cop || a.CD != nb * 16) return VSR_ERR_UNSUPPORTED;
    if (pm_image_elems(QTH + 2, a.W, a.CD) > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
#define C7_CASE(CIN, NBv) if (cin == CIN && nb == NBv) return launch_c7<CIN, NBv, EPI_NHWC>(a, cop, num_cus, st);
    C7_CASE(16, 2)      // 8 -> 32
    C7_CASE(32, 4)      // 32 -> 64
    C7_CASE(64, 2)      // 64 -> 32
    C7_CASE(32, 1)      // 32 -> 16
    C7_CASE(16, 1)      // data gradient of 16 -> 2 (16 padded rows -> 16 channels)
#undef C7_CASE
    return VSR_ERR_UNSUPPORTED;
}
