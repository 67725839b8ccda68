// A whole chain of dependent 3x3 64 -> 64 convolutions in ONE launch: the 60 layers of a frame's 30 ResidualConv blocks
// (core/modules/conv.py:85-92 inside conv.py:94-103; basicvsr.py:56-58,71-73) forward, and their 59 data gradients backward.
//
// Why: a 540p layer is 2040 tiles of 8x32 pixels = 8 per CU.  As its own launch (conv3x3_persist.hip) a layer pays ~6.4 k
// cycles of prologue (kernel arguments, 72 KiB of weights, the first tile's HBM round trip) and ~6 us of launch / drain around 8 x
// 6.1 k cycles of tiles.  Here the workgroups stay resident over all layers of the chain; a layer boundary costs a workgroup one
// weight reload, and nobody waits for the slowest workgroup of a layer.  What it buys, measured: 58.6 k cycles per
// workgroup and layer instead of ~66 k -- but the chip answers the denser MFMA stream with a lower clock (1.5 against 1.75-1.9
// GHz in the one-layer kernel; MI355X_MICROARCH "DVFS give-back" item 3), so a layer takes ~40 us either way out of cache and
// the step gains 2 % (the two-stream engine already hid most of the per-launch gaps).  Energy per tile, not idle time, is what
// bounds this convolution on this chip.
//
// How the layers are ordered without a grid barrier:
//   * a finished tile is PUBLISHED per MFMA wave: outputs are stored write-through (global_store ... sc1: the per-XCD L2s are not
//     coherent), the wave waits for its stores (s_waitcnt vmcnt(0)), then one lane stores 1 into the wave's flag word of the tile
//     (sc1): flags[item][wave], 16 bytes per tile.  MI355X_MICROARCH "inter-workgroup visibility": payload sc1, every storing
//     wave drains, flag written sc1, consumer polls sc1.  (Row COUNTERS instead of flags -- 120 atomic adds per tile row -- cost
//     30 ms per step: same-line atomics serialise at the memory side and sit in the waves' in-order vmcnt queues.)  The flag
//     store is DEFERRED to the top of the wave's next epilogue, where the stores have long been acknowledged -- unless the
//     workgroup's next item is not known to be ready (then the wave drains and publishes at once: on small images the next item
//     can depend on this one).
//   * consumers: the leader producer wave looks at the flags of the 3x3 tiles around an item two tiles ahead (nine lanes, two
//     8-byte sc1 loads each) and leaves the verdict in LDS; a workgroup barrier later the producer waves issue the tile's LDS-DMA
//     (sc1: bypasses the CU's L1).  Not ready at that point -> every producer wave polls for itself (bounded by a 1 s clock; a
//     timeout raises the launch's error word, which ends every wait, so the grid always drains).  A launch that gave up a wait
//     has consumed unfinished tiles: chain_poison_kernel, enqueued behind every chain launch, then overwrites the head of the
//     chain's LAST image with NaN, so that sr, the loss and every gradient norm downstream go non-finite (FusedAdam skips the
//     step) without a host synchronisation; vsr_debug_chain_timeouts counts the give-ups for diagnosis and the tests require 0.
//     The residual operand of a layer is the output of the layer before the previous one at the same tile: complete by
//     transitivity, loaded sc1.
//   * every buffer of a chain is written exactly once and read only after its flags say so (training arena: one buffer per
//     layer), so no CU can hold a stale line of it.
//   * work distribution and the progress argument: "Work distribution" below.
//
// Inline-asm notes (both cost a wrong result each before they were in): a 16-byte asm store ends in `s_nop 1` (hipcc's next
// instruction may otherwise overwrite the data registers before the store has read them) and the first vector-memory asm
// behind the scalar adds of its base starts with `s_nop 4` (cdna_hip_programming.md 5.7).
//
// The tile loop itself (K loop, fragment schedule, epilogues) is conv3x3_persist.hip's; see there for the LDS images.
#include "common.h"
#include "kernels.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int PTW = 32, PTH = 8, PNT = 512;
constexpr int PTWH = PTW + 2, PTHH = PTH + 2, PNPIX = PTHH * PTWH;
constexpr int W_BYTES = 9 * 64 * 64 * 2;
constexpr int IN_BYTES = PNPIX * 128;
constexpr int BIAS_OFF = W_BYTES + 2 * IN_BYTES;
constexpr int CTL_OFF = BIAS_OFF + 256;                                   // 3 item slots of 8 ints
constexpr int P_LDS = CTL_OFF + 128;                                      // 161,152 <= 163,840
constexpr int IN_CHUNKS = PNPIX * 8;
constexpr int NPIECE_T = (IN_CHUNKS + 63) / 64;
constexpr int NPIECE_W = (NPIECE_T + 3) / 4;                              // 11; the leader wave (w4 = 0) always issues exactly 11

// Diagnostic build only (make ABL=<bits> ABLSRC=conv3x3_chain; results may be WRONG, only the run time is read): bit 0: plain
// instead of write-through stores; bit 1: plain instead of sc1 loads; bit 2: no dependency waits (every item counts as ready);
// bit 3: no publishes; bit 4: no wait for the previous tile's stores at the top of an epilogue without operands; bit 5: no fragment
// reads after step 0 (bare MFMA loop); bit 6: the producers issue only their first tile (no DMA); bit 7: no output stores; bit 11: every fragment read
// of a tile re-reads the first K step's addresses (the LDS reads stay, the operands stop changing); bit 12: epilogue = convert + store; bit 13 / 14: no A (weight) / no B (pixel)
// fragment reads after the first K step (what weights held in registers would save); bit 15: the weight image in LDS in MFMA-fragment
// order (every A read = 1 KiB contiguous) instead of [cout row][8 swizzled chunks] -- same results, an energy experiment; bit 16: no A
// (weight) fragment reads for the taps of kernel row ky = 1 (the upper bound of what holding that row's 24 fragments = 96 registers
// per lane in registers would save: round-3 VERDICT next #1a); bit 17: only the first 8 of the 18 K steps = 128 of 288 MFMAs and
// 64 of 144 fragment reads per tile and wave (the matrix work of a Winograd F(2x2, 3x3) kernel WITHOUT its transforms: VERDICT #1c);
// bit 18: a new layer's weights are loaded at the layer change itself (round 3's exposed L2 round trip) instead of a tile ahead.  On the
// back-to-back leg the kernel's time is its energy, so these price the energy of LDS reads / DMA / stores.
#ifdef VSR_ABL
#define CABL(bit) ((VSR_ABL >> (bit)) & 1)
__device__ unsigned long long g_clk_chain[256 * 2];      // [workgroup][cycles, 100 MHz ticks] of the last chain launch
#else
#define CABL(bit) 0
#endif
#if CABL(0)
#define CH_SC1_ST ""
#elif CABL(8)
#define CH_SC1_ST " sc1 nt"
#elif CABL(9)
#define CH_SC1_ST " nt"
#elif CABL(10)
#define CH_SC1_ST " sc0 sc1"
#else
#define CH_SC1_ST " sc1"
#endif
#if CABL(1)
#define CH_SC1_LD ""
#define CH_AUX 0
#else
#define CH_SC1_LD " sc1"
#define CH_AUX 16
#endif
__device__ uint4 g_chain_zero_chunk[2];
__device__ unsigned g_chain_timeouts;           // waits given up since the module was loaded (never reset: vsr_debug_chain_timeouts)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2_t p = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_mul_lo_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

#define GLDS16_SC1(src, dst)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, CH_AUX)
#define GP(T, x) ((__attribute__((address_space(1))) T*)(x))

// The 3 x 3 tiles around (ty, tx) of one layer's image have been published by all four of their MFMA waves: flg points at the
// image's first tile, one 16-byte record of four flag words per tile; lanes 0..8 look at one tile each (two 8-byte sc1 loads).
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ bool halo_ready(const gu32* flg, int ty, int tx, int nty, int ntx, int lane) {
    bool ok = true;
    const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
    const int y = ty + dy, x = tx + dx;
    if (lane < 9 && y >= 0 && y < nty && x >= 0 && x < ntx) {
        const gu64* f = reinterpret_cast<const gu64*>(flg) + 2 * (long long)(y * ntx + x);
        const unsigned long long a = __hip_atomic_load(f, RLX_AGENT), b = __hip_atomic_load(f + 1, RLX_AGENT);
        ok = a == 0x0000000100000001ull && b == 0x0000000100000001ull;
    }
    return __all(ok);
}
// the slow path of a consumer: poll until ready; false = gave up (the error word is set: results are void)
__device__ __forceinline__ bool halo_wait(const gu32* flg, int ty, int tx, int nty, int ntx, int lane, gu32* err) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (halo_ready(flg, ty, tx, nty, ntx, lane)) return true;
        if (__hip_atomic_load(err, RLX_AGENT) != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {                // 1 s of the 100 MHz clock
            if (lane == 0) { __hip_atomic_fetch_add(err, 1u, RLX_AGENT); __hip_atomic_fetch_add(GP(unsigned, &g_chain_timeouts), 1u, RLX_AGENT); }
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

// Work distribution.  A layer's tiles are split into R = 8 contiguous regions, one per XCD (workgroup b runs on XCD b % 8: the
// tiles a region's workgroups load at the same time are neighbours, so their halos are served by that XCD's L2, as with
// xcd_tile_walk in the one-layer kernel).  Each region has ONE hand-out counter for the whole chain: position P of region r is
// tile chain_region_tile(P % size_r) of layer P / size_r, so a fetch is a single atomic and a workgroup's items never go back a
// layer.  Progress: take the unfinished item m of the lowest layer.  Everything in lower layers is finished, so every TAKEN item
// of m's layer can complete; the workgroups of m's region work through the region's positions in order, so they reach m -- as
// long as every region has a workgroup that is or becomes resident, i.e. as long as no other kernel that waits for THIS one
// holds its CUs: the engine therefore runs its chain launches on one stream.  (R = 1, one counter = one global order, when the
// grid is not a multiple of 8 workgroups.)
__host__ __device__ __forceinline__ int chain_rlo(int r, int tiles, int R) { return (int)((long long)r * tiles / R); }
// The p-th tile handed out of a region of `sz` tiles: chunks of `ntx` consecutive tiles (about one tile row), taken MIDDLE-OUT
// (m, m+1, m-1, m+2, ...).  In the next layer a chunk needs its neighbour chunks of this layer (the 3x3 halo): with the same
// order in every layer those were handed out a whole layer earlier, also across region borders, whose chunks come last --
// top-to-bottom, the first chunk of every region would wait for the previous region's LAST chunk of the layer before.
__host__ __device__ __forceinline__ int chain_region_tile(int p, int sz, int ntx) {
    const int nch = (sz + ntx - 1) / ntx, m = nch >> 1, D = m, U = nch - 1 - m;
    const int lim = 2 * (U < D ? U : D);
    for (int sq = 0; sq < nch; ++sq) {
        int c;
        if (sq == 0) c = m;
        else if (sq <= lim) c = (sq & 1) ? m + ((sq + 1) >> 1) : m - (sq >> 1);
        else c = U > D ? m + (sq - D) : m - (sq - U);
        const int c0 = c * ntx, cn = (c0 + ntx <= sz ? ntx : sz - c0);
        if (p < cn) return c0 + p;
        p -= cn;
    }
    return sz - 1;      // (not reached for p < sz)
}
// position P of region `own` -> item (layer * tiles + tile), or -1 behind the last layer
__host__ __device__ __forceinline__ int chain_item(unsigned P, int own, int R, int tiles, int ntx, int nlayers) {
    const int lo = chain_rlo(own, tiles, R), sz = chain_rlo(own + 1, tiles, R) - lo;
    if (sz <= 0) return -1;
    const int l = (int)(P / (unsigned)sz);
    if (l >= nlayers) return -1;
    return l * tiles + lo + chain_region_tile((int)(P - (unsigned)l * (unsigned)sz), sz, ntx);
}

// EVEN: the variant of the chain's layers that are not CHAIN_SKIP (forward chains: CHAIN_RELU, backward chains: CHAIN_MASK)
template <int EVEN>
__global__ __launch_bounds__(PNT, 1) void conv3x3_c64_chain_kernel(const ChainArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);            // 0: MFMA + epilogue, 1: LDS-DMA producer
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
    const int l15 = lane & 15, q = lane >> 4;
    const int pxl = (l15 >= 4 && l15 < 12) ? 2 * (l15 - 4) : (l15 < 4 ? 2 * l15 + 1 : 2 * (l15 - 8) + 1);
    char* lds_w = smem;
    char* lds_t = smem + W_BYTES;
    int* const ctl = reinterpret_cast<int*>(smem + CTL_OFF);              // slot s: ctl + 8 s = {item, layer, n, ty, tx, ready, -, -}

    const int H = ka.H, W = ka.W;
    const int ntx = cdiv(W, PTW), nty = cdiv(H, PTH);
    const int per = ntx * nty, tiles = ka.N * per;
    const int WS = pm_ws(W);
    const long long img = pm_image_elems(H, W, 64);
    char* const base = ka.base;
    gu32* const work = GP(unsigned, ka.sync);
    gu32* const err = work + 1;
    const int R = (gridDim.x >= 8 && (gridDim.x & 7) == 0) ? 8 : 1;
    const int own = R == 8 ? (int)(blockIdx.x & 7) : 0;
    gu32* const qown = work + 64 + 16 * own;                               // this region's hand-out counter (a 64-byte line of its own)
    gu32* const flg0 = work + 256;                                         // [item = layer * tiles + tile][4 MFMA waves]

#ifdef VSR_ABL
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (tid == 0) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // ---- the first two items of this workgroup ----
    if (tid == 256) {
        const unsigned v = __hip_atomic_fetch_add(qown, 2u, RLX_AGENT);
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const int it = chain_item(v + sidx, own, R, tiles, ntx, ka.nlayers);
            int* sl = ctl + 8 * sidx;
            if (it >= 0) {
                const int l = it / tiles, r = it - l * tiles, n = r / per, r2 = r - n * per, ty = r2 / ntx;
                sl[0] = it; sl[1] = l; sl[2] = n; sl[3] = ty; sl[4] = r2 - ty * ntx; sl[5] = (l == 0 || CABL(2)) ? 1 : 0;
            } else {
                sl[0] = -1; sl[1] = 0; sl[2] = 0; sl[3] = 0; sl[4] = 0; sl[5] = 0;
            }
        }
    }
    __syncthreads();
    if (ctl[0] < 0) return;                                                // more workgroups than items (uniform)

#define W_LOAD(wv, wg)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                                    \
        const int idx = (tid & 255) + i * 256;        /* 256 threads stage a weight set: the MFMA waves or the producers */ \
        const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;                                                      \
        wv[i] = (wg)[(tap * 64 + pm_acc_chan(r >> 4, r & 15)) * 8 + c];                                                  \
    }
#define W_STORE(wv)                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                                    \
        const int idx = (tid & 255) + i * 256;                                                                           \
        const int tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;                                                      \
        *reinterpret_cast<u32x4_t*>(lds_w + (CABL(15) ? ((((tap * 2 + (c >> 2)) * 4 + (r >> 4)) * 64 + (c & 3) * 16 + (r & 15)) * 16) : tap * 8192 + (r * 8 + (c ^ ((r >> 1) & 7))) * 16)) = wv[i]; \
    }
    constexpr int WCH = 9 * 64 * 8 / 256;

    if (role == 1) {
        // =================== producer waves ===================
        const bool leader = w4 == 0;
        const auto* zsrc = GP(const char, g_chain_zero_chunk);
        int rel[NPIECE_W];
#pragma unroll
        for (int i = 0; i < NPIECE_W; ++i) {
            const int idx = (w4 + 4 * i) * 64 + lane;
            const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
            const int c = rem / PTWH, tx = rem - c * PTWH;
            const int dx = tx - 1;
            rel[i] = ((((ty - 1) * WS + (dx >> 5)) * 8 + c) * 256 + (dx & 31) * 8) * 2;
        }
        u32x4_t wvp[WCH];                                          // the next layer's weights on their way to LDS
        float bvp = 0.f;
        const int pt = tid - 256;
        int p_layer = -1;
        const char* p_src = nullptr;
        auto issue = [&](int layer, int n, int tyi, int txi, int buf) {
            if (layer != p_layer) { p_layer = layer; p_src = base + (unsigned long long)ka.layer[layer].src * 256ull; }
            const int ty0 = tyi * PTH, tx0 = txi * PTW;
            const auto* org = GP(const char, p_src) + ((long long)n * img + pm_off(ty0, tx0, 0, W, 64)) * 2;
            char* dstb = lds_t + buf * IN_BYTES;
            if (ty0 >= 1 && ty0 + PTH < H && tx0 >= 1 && tx0 + PTW < W) {
#pragma unroll
                for (int i = 0; i < NPIECE_W; ++i) {
                    const int piece = w4 + 4 * i;
                    if (piece < NPIECE_T && piece * 64 + lane < IN_CHUNKS) GLDS16_SC1(org + rel[i], dstb + piece * 1024);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPIECE_W; ++i) {
                    const int piece = w4 + 4 * i;
                    const int idx = piece * 64 + lane;
                    const int ty = idx / (8 * PTWH), rem = idx - ty * (8 * PTWH);
                    const int tx = rem % PTWH;
                    const int vy = ty0 + ty - 1, vx = tx0 + tx - 1;
                    const auto* s = (vy >= 0 && vy < H && vx >= 0 && vx < W) ? org + rel[i] : zsrc;
                    if (piece < NPIECE_T && idx < IN_CHUNKS) GLDS16_SC1(s, dstb + piece * 1024);
                }
            }
        };
        // item 0: nothing was polled for it yet
        {
            const int l0 = ctl[1], n0 = ctl[2], ty0 = ctl[3], tx0 = ctl[4];
            if (l0 > 0 && !CABL(2)) (void)halo_wait(flg0 + 4 * ((long long)(l0 - 1) * tiles + (long long)n0 * per), ty0, tx0, nty, ntx, lane, err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            issue(l0, n0, ty0, tx0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // weights of the first layer and the first tile are in LDS
        int cur_layer = ctl[1];
        for (int k = 0;; ++k) {
            const int* sc = ctl + 8 * (k % 3);
            if (sc[0] < 0) break;
            const int layer_k = __builtin_amdgcn_readfirstlane(sc[1]);
            // The MFMA waves' weight reload of a layer change: joined FIRST.  Behind the dependency wait below it would deadlock:
            // the MFMA waves would sit in this barrier in front of tile k while the wait is for tile k-1's deferred publish,
            // which they make in tile k's epilogue.
            if (layer_k != cur_layer) {
                // r04: the PRODUCERS stage a new layer's weights.  They were requested a whole tile ago (below) and are in registers;
                // the weight image is free (every MFMA wave is past the K loop of the previous tile: the barrier that ended it).
                // Before, the MFMA waves loaded, waited and wrote here, with the L2 round trip exposed once per layer and workgroup.
                cur_layer = layer_k;
                if (CABL(18)) {                                        // (diagnostic: the loads exposed here, as in round 3)
                    const auto* wgx = GP(const u32x4_t, base + (unsigned long long)ka.layer[layer_k].w * 256ull);
                    W_LOAD(wvp, wgx)
                    const unsigned bo = ka.layer[layer_k].bias;
                    bvp = 0.f;
                    if (pt < 64 && bo != 0xffffffffu) bvp = GP(const float, base + (unsigned long long)bo * 256ull)[pm_acc_chan(pt >> 4, pt & 15)];
                }
                W_STORE(wvp)
                if (pt < 64) reinterpret_cast<float*>(smem + BIAS_OFF)[pt] = bvp;
                __syncthreads();
            }
            int* s1 = ctl + 8 * ((k + 1) % 3);
            const int it1 = __builtin_amdgcn_readfirstlane(s1[0]);
            // the next item opens a new layer: its 72 KiB of packed weights (18 x 16 bytes per producer thread) and its bias are
            // requested now, the OLDEST vector-memory operations of this iteration (the leader's counted wait below looks at the
            // youngest ones only); they are written to LDS at the top of the next iteration
            if (it1 >= 0 && !CABL(18)) {
                const int l1w = __builtin_amdgcn_readfirstlane(s1[1]);
                if (l1w != cur_layer) {
                    const auto* wgx = GP(const u32x4_t, base + (unsigned long long)ka.layer[l1w].w * 256ull);
                    W_LOAD(wvp, wgx)
                    const unsigned bo = ka.layer[l1w].bias;
                    bvp = 0.f;
                    if (pt < 64 && bo != 0xffffffffu) bvp = GP(const float, base + (unsigned long long)bo * 256ull)[pm_acc_chan(pt >> 4, pt & 15)];
                }
            }
            // the leader asks for item k+2 first: the atomic's round trip runs under the poll / DMA issue below
            unsigned nx = 0xffffffffu;                                    // (never a counter value)
            const bool fetch = leader && it1 >= 0;
            if (fetch && lane == 0)
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(nx) : "v"(qown), "v"(1u) : "memory");
            if (it1 >= 0) {
                const int l1 = __builtin_amdgcn_readfirstlane(s1[1]), n1 = __builtin_amdgcn_readfirstlane(s1[2]);
                const int ty1 = __builtin_amdgcn_readfirstlane(s1[3]), tx1 = __builtin_amdgcn_readfirstlane(s1[4]);
                if (__builtin_amdgcn_readfirstlane(s1[5]) == 0) {
                    (void)halo_wait(flg0 + 4 * ((long long)(l1 - 1) * tiles + (long long)n1 * per), ty1, tx1, nty, ntx, lane, err);
                    if (leader && lane == 0) s1[5] = 1;   // tells the MFMA waves that their deferred publish cannot be what we wait for
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (!CABL(6)) issue(l1, n1, ty1, tx1, (k + 1) & 1);
            }
            if (leader) {
                int* s2 = ctl + 8 * ((k + 2) % 3);
                int it2 = -1, l2 = 0, n2 = 0, ty2 = 0, tx2 = 0;
                bool ok2 = false;
                if (fetch) {
                    // exactly NPIECE_W vector-memory instructions of this wave are younger than the atomic (vmcnt counts in issue order)
                    asm volatile("s_waitcnt vmcnt(11)" : "+v"(nx) :: "memory");
                    static_assert(NPIECE_W == 11, "the counted wait above");
                    if (__builtin_amdgcn_readfirstlane((int)nx) == -1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx) :: "memory");   // (belt and braces)
                    it2 = chain_item((unsigned)__builtin_amdgcn_readfirstlane((int)nx), own, R, tiles, ntx, ka.nlayers);
                    if (it2 >= 0) {
                        l2 = it2 / tiles; const int r = it2 - l2 * tiles; n2 = r / per; const int r2 = r - n2 * per; ty2 = r2 / ntx; tx2 = r2 - ty2 * ntx;
                        // its dependencies, looked at now, used a barrier later (the result returns behind this wave's DMA pieces,
                        // i.e. under the wait for the tile that is needed anyway)
                        ok2 = l2 == 0 || CABL(2) || halo_ready(flg0 + 4 * ((long long)(l2 - 1) * tiles + (long long)n2 * per), ty2, tx2, nty, ntx, lane);
                    }
                }
                if (lane == 0) { s2[0] = it2; s2[1] = l2; s2[2] = n2; s2[3] = ty2; s2[4] = tx2; s2[5] = ok2 ? 1 : 0; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // =================== MFMA waves ===================
        u32x4_t wv[WCH];                                           // the FIRST layer's weights (prologue only: the producers are busy with the first tile)
        {
            const auto* wg = GP(const u32x4_t, base + (unsigned long long)ka.layer[ctl[1]].w * 256ull);
            W_LOAD(wv, wg)
        }
        unsigned a_lo[2], a_hi[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_lo[kk] = CABL(15) ? (unsigned)((q * 16 + l15) * 16 + kk * 4096) : (unsigned)((l15 * 8 + ((4 * kk + q) ^ ((l15 >> 1) & 7))) * 16);
            a_hi[kk] = a_lo[kk] + 6 * 8192;
        }
        const int b_lane = w4 * 2 * (PTWH * 128) + q * (PTWH * 16) + pxl * 16;
        unsigned loff[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int dx = (nb & 1) * 16 + pxl;
            loff[nb] = (((((w4 * 2 + (nb >> 1))) * WS + (dx >> 5)) * 8 + q) * 256 + (dx & 31) * 8) * 2;     // BYTES from the tile's origin
        }
        W_STORE(wv)
        if (tid < 64) {
            const unsigned bo = ka.layer[ctl[1]].bias;
            reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = bo != 0xffffffffu ? GP(const float, base + (unsigned long long)bo * 256ull)[pm_acc_chan(tid >> 4, tid & 15)] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4_t bvec[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) bvec[mb] = *reinterpret_cast<const f32x4_t*>(smem + BIAS_OFF + (mb * 16 + 4 * q) * 4);
        bf16x8_t idA[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 8; ++j) idA[hb][j] = (bf16_t)((q == (l15 >> 2) && j == 4 * hb + (l15 & 3)) ? 1.f : 0.f);

        // the layer whose operands are loaded
        int cur_layer = __builtin_amdgcn_readfirstlane(ctl[1]);
        unsigned long long p_dst, p_res, p_sbits, p_sout;
        int variant;
        auto layer_ptrs = [&](int l) {
            const ChainLayer L = ka.layer[l];
            p_dst = (unsigned long long)base + (unsigned long long)L.dst * 256ull;
            p_res = (unsigned long long)base + (unsigned long long)L.res * 256ull;
            p_sbits = (unsigned long long)base + (unsigned long long)L.sbits * 256ull;
            p_sout = L.sout != 0xffffffffu ? (unsigned long long)base + (unsigned long long)L.sout * 256ull : 0ull;
            variant = L.variant;
        };
        layer_ptrs(cur_layer);

        bool pending = false;                 // a finished tile of this wave whose flag word has not been set yet
        gu32* pend_cnt = nullptr;             // that flag word: flags[item][w4]
        int cur = 0;
        // item k of this workgroup (slot k % 3); the next slot is read at the top of a tile and used at its end
        int item = __builtin_amdgcn_readfirstlane(ctl[0]), layer = cur_layer, tn = __builtin_amdgcn_readfirstlane(ctl[2]);
        int tyi = __builtin_amdgcn_readfirstlane(ctl[3]), txi = __builtin_amdgcn_readfirstlane(ctl[4]);
        for (int k = 0; item >= 0; ++k) {
            const int* sn = ctl + 8 * ((k + 1) % 3);     // the next item: read behind the K loop (registers), used at the end of the tile
            int nx_item = -1, nx_layer = 0, nx_n = 0, nx_ty = 0, nx_tx = 0, nx_ok = 0;
            if (layer != cur_layer) {
                // (the producers write the new weight image and bias: see there)
                cur_layer = layer;
                layer_ptrs(layer);
                __syncthreads();
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) bvec[mb] = *reinterpret_cast<const f32x4_t*>(smem + BIAS_OFF + (mb * 16 + 4 * q) * 4);
            }
            const int tile = item - layer * tiles;
            const int ty0 = tyi * PTH, tx0 = txi * PTW;
            const long long tbase = (long long)tn * img + pm_off(ty0, tx0, 0, W, 64);
            const bool full = ty0 + PTH <= H && tx0 + PTW <= W;
            bool ok[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ok[nb] = (tx0 + (nb & 1) * 16 + pxl < W) && (ty0 + w4 * 2 + (nb >> 1) < H);
            gu32* const my_cnt = flg0 + 4 * (long long)item + w4;

            auto body = [&](auto VC) {
                constexpr int V = decltype(VC)::value;
                constexpr bool HAS_RES = V == CHAIN_SKIP, BITS = V == CHAIN_MASK;
                u32x4_t rr[2][4];
                u32x2_t sbits = {0u, 0u};
                if (BITS) sbits = GP(const u32x2_t, p_sbits)[(long long)tile * 256 + w4 * 64 + lane];
                if (HAS_RES) {
                    // the residual = the output of layer - 2 of this chain (or the chain's input): write-through stored there, sc1 here
                    const unsigned long long rbase = p_res + (unsigned long long)tbase * 2ull;
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const unsigned lo = loff[nb];
                        // (inline asm is not padded by hipcc: `s_nop 4` = the wait states between the scalar adds that made rbase and a
                        // vector-memory instruction using it as its base; cdna_hip_programming.md 5.7 item 2)
                        if (full) {
                            if (nb == 0) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" CH_SC1_LD : "=v"(rr[0][nb]) : "v"(lo), "s"(rbase) : "memory");
                            else asm volatile("global_load_dwordx4 %0, %1, %2" CH_SC1_LD : "=v"(rr[0][nb]) : "v"(lo), "s"(rbase) : "memory");
                            asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" CH_SC1_LD : "=v"(rr[1][nb]) : "v"(lo), "s"(rbase) : "memory");
                        } else {
                            rr[0][nb] = u32x4_t{0u, 0u, 0u, 0u}; rr[1][nb] = u32x4_t{0u, 0u, 0u, 0u};
                            if (ok[nb]) {
                                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" CH_SC1_LD : "+v"(rr[0][nb]) : "v"(lo), "s"(rbase) : "memory");
                                asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" CH_SC1_LD : "+v"(rr[1][nb]) : "v"(lo), "s"(rbase) : "memory");
                            }
                        }
                    }
                }
                f32x4_t acc[4][4];
                bf16x8_t fa[2][4], fb[2][4];
                const unsigned bb = (unsigned)(W_BYTES + cur * IN_BYTES + b_lane);
#define DSR(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define CV_LOADA(tap_, kk_, slot, mb) if (!((CABL(5) || CABL(13)) && tap_ + kk_ > 0) && !(CABL(16) && tap_ >= 3 && tap_ <= 5)) DSR(fa[slot][mb], (CABL(11) ? a_lo[0] : (tap_ < 6 ? a_lo[kk_] : a_hi[kk_])), (CABL(11) ? 0 : (tap_ < 6 ? tap_ : tap_ - 6) * 8192) + (mb) * (CABL(15) ? 1024 : 2048));
#define CV_LOADB(ky_, kx_, kk_, slot, nb) if (!((CABL(5) || CABL(14)) && ky_ + kx_ + kk_ > 0)) DSR(fb[slot][nb], bb, CABL(11) ? (((nb) >> 1)) * (PTWH * 128) + (((nb) & 1) * 16) * 16 : (((nb) >> 1) + ky_) * (PTWH * 128) + kk_ * (4 * PTWH * 16) + (((nb) & 1) * 16 + kx_) * 16);
#define CV_LOAD(s, slot)                                                                                               \
                {                                                                                                      \
                    constexpr int tap_ = (s) / 2, kk_ = (s) % 2, ky_ = tap_ / 3, kx_ = tap_ % 3;                       \
                    CV_LOADA(tap_, kk_, slot, 0) CV_LOADA(tap_, kk_, slot, 1) CV_LOADA(tap_, kk_, slot, 2) CV_LOADA(tap_, kk_, slot, 3) \
                    CV_LOADB(ky_, kx_, kk_, slot, 0) CV_LOADB(ky_, kx_, kk_, slot, 1) CV_LOADB(ky_, kx_, kk_, slot, 2) CV_LOADB(ky_, kx_, kk_, slot, 3) \
                }
#define CV_MFMA(s, mb, nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(s) % 2][mb], fb[(s) % 2][nb], (s) == 0 ? bvec[mb] : acc[mb][nb], 0, 0, 0);
#define CV_ML_A(s, mb, nb, lmb)                                                                                        \
                CV_MFMA(s, mb, nb)                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if ((s) + 1 < 18) { constexpr int t1_ = ((s) + 1) / 2, k1_ = ((s) + 1) % 2; CV_LOADA(t1_, k1_, ((s) + 1) % 2, lmb) } \
                __builtin_amdgcn_sched_barrier(0);
#define CV_ML_B(s, mb, nb, lnb)                                                                                        \
                CV_MFMA(s, mb, nb)                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if ((s) + 1 < 18) { constexpr int t1_ = ((s) + 1) / 2, k1_ = ((s) + 1) % 2; CV_LOADB(t1_ / 3, t1_ % 3, k1_, ((s) + 1) % 2, lnb) } \
                __builtin_amdgcn_sched_barrier(0);
#define CV_STEP(s)                                                                                                     \
                {                                                                                                      \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                    CV_ML_A(s, 0, 0, 0) CV_ML_A(s, 0, 1, 1) CV_ML_A(s, 0, 2, 2) CV_ML_A(s, 0, 3, 3)                    \
                    CV_ML_B(s, 1, 0, 0) CV_ML_B(s, 1, 1, 1) CV_ML_B(s, 1, 2, 2) CV_ML_B(s, 1, 3, 3)                    \
                    CV_MFMA(s, 2, 0) CV_MFMA(s, 2, 1) CV_MFMA(s, 2, 2) CV_MFMA(s, 2, 3)                                \
                    CV_MFMA(s, 3, 0) CV_MFMA(s, 3, 1) CV_MFMA(s, 3, 2) CV_MFMA(s, 3, 3)                                \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                CV_LOAD(0, 0)
                CV_STEP(0) CV_STEP(1) CV_STEP(2) CV_STEP(3) CV_STEP(4) CV_STEP(5) CV_STEP(6) CV_STEP(7)
                if (!CABL(17)) {
                CV_STEP(8)
                CV_STEP(9) CV_STEP(10) CV_STEP(11) CV_STEP(12) CV_STEP(13) CV_STEP(14) CV_STEP(15) CV_STEP(16) CV_STEP(17)
                }
#undef CV_ML_A
#undef CV_ML_B
#undef CV_STEP
#undef CV_MFMA
#undef CV_LOAD
#undef CV_LOADB
#undef CV_LOADA
#undef DSR
                nx_item = sn[0]; nx_layer = sn[1]; nx_n = sn[2]; nx_ty = sn[3]; nx_tx = sn[4]; nx_ok = sn[5];
                // ---- epilogue ----
                // vmcnt(0): this tile's operands are here, and the PREVIOUS tile's stores (issued a K loop ago) are acknowledged:
                // now its flag word may be set (R1: every storing wave drains, then publishes for itself)
                // (the builtin, so that hipcc's scoreboard knows the queue is empty: behind an asm wait it would wait again, for the
                // publishing atomic below, in front of the first use of the sign bits)
                if (!(CABL(4) && V == CHAIN_RELU)) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) alone
                __builtin_amdgcn_sched_barrier(0);
                if (pending) {
                    if (lane == 0 && !CABL(3)) __hip_atomic_store(pend_cnt, 1u, RLX_AGENT);
                    pending = false;
                }
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idA[mb & 1], __builtin_bit_cast(bf16x8_t, rr[mb >> 1][nb]), acc[mb][nb], 0, 0, 0);
                }
                unsigned sout[2] = {0u, 0u};
                const unsigned k11 = 0x00010001u;
                const unsigned long long dbase = p_dst + (unsigned long long)tbase * 2ull;
                auto epilogue = [&](auto FULL) {
                    const unsigned long long db = dbase;          // (asm operands do not capture by themselves in a generic lambda)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        if (decltype(FULL)::value || ok[nb]) {
                            const unsigned lo = loff[nb];
#pragma unroll
                            for (int kq = 0; kq < 2; ++kq) {
                                float v[8];
#pragma unroll
                                for (int j = 0; j < 4; ++j) { v[j] = acc[2 * kq][nb][j]; v[4 + j] = acc[2 * kq + 1][nb][j]; }
                                const unsigned wbits = kq ? sbits.y : sbits.x;
                                unsigned ow[4];
                                if (CABL(12)) {
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj) ow[jj] = pk_bf16(v[2 * jj], v[2 * jj + 1]);
                                } else if (V == CHAIN_RELU) {
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj) {
                                        ow[jj] = pk_max_i16(pk_bf16(v[2 * jj], v[2 * jj + 1]), 0u);
                                        sout[kq] |= pk_min_u16(ow[jj], k11) << (4 * nb + jj);
                                    }
                                } else if (V == CHAIN_MASK) {
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj)
                                        ow[jj] = pk_mul_lo_u16(pk_bf16(v[2 * jj], v[2 * jj + 1]), (wbits >> (4 * nb + jj)) & k11);
                                } else {
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj) ow[jj] = pk_bf16(v[2 * jj], v[2 * jj + 1]);
                                }
                                const u32x4_t o = {ow[0], ow[1], ow[2], ow[3]};
                                // write-through: the next layer's tiles may be loaded on another XCD
                                // (asm: `s_nop 1` behind a 16-byte store, or hipcc's next instruction may overwrite the data registers before
                                // the store has read them; `s_nop 4` in front of the first user of the freshly added base; 5.7 items 1, 2)
                                if (CABL(7)) asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                                else if (kq == 0) {
                                    if (nb == 0 || !decltype(FULL)::value) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" CH_SC1_ST "\n\ts_nop 1" :: "v"(lo), "v"(o), "s"(db) : "memory");
                                    else asm volatile("global_store_dwordx4 %0, %1, %2" CH_SC1_ST "\n\ts_nop 1" :: "v"(lo), "v"(o), "s"(db) : "memory");
                                } else asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048" CH_SC1_ST "\n\ts_nop 1" :: "v"(lo), "v"(o), "s"(db) : "memory");
                            }
                        }
                    }
                };
                if (full) epilogue(std::true_type{}); else epilogue(std::false_type{});
                if (V == CHAIN_RELU && p_sout) GP(u32x2_t, p_sout)[(long long)tile * 256 + w4 * 64 + lane] = u32x2_t{sout[0], sout[1]};
            };
            if (variant == CHAIN_SKIP) body(std::integral_constant<int, CHAIN_SKIP>{});
            else body(std::integral_constant<int, EVEN>{});

            pending = true; pend_cnt = my_cnt;
            // The deferred publish is safe only if the workgroup's next tile does not wait for it: unless the producers have
            // already seen that tile's dependencies complete, drain and publish now.
            if (nx_item < 0 || nx_ok == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0 && !CABL(3)) __hip_atomic_store(pend_cnt, 1u, RLX_AGENT);
                pending = false;
            }
            __syncthreads();                               // the next tile has landed; everybody has finished reading `cur`
            cur ^= 1;
            item = __builtin_amdgcn_readfirstlane(nx_item); layer = __builtin_amdgcn_readfirstlane(nx_layer); tn = __builtin_amdgcn_readfirstlane(nx_n);
            tyi = __builtin_amdgcn_readfirstlane(nx_ty); txi = __builtin_amdgcn_readfirstlane(nx_tx);
        }
    }
#ifdef VSR_ABL
    if (tid == 0 && blockIdx.x < 256) {
        g_clk_chain[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
        g_clk_chain[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
}

// Behind every chain launch, on its stream: a launch whose error word is set (a dependency wait was given up, or the word was
// forced by vsr_debug_chain_inject_error) has read tiles nobody had finished.  Its results are void, and they are MADE to look
// void: the first 32 pixels x 64 channels of the chain's last image become NaN (pixel (0, 0) of image 0 is real data in every
// shape), which every consumer of that image carries into sr / the loss / the weight gradients.  No host round trip.
__global__ void chain_poison_kernel(const unsigned* err, unsigned short* last_image) {
    if (__hip_atomic_load(err, RLX_AGENT) == 0u) return;
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) last_image[i] = 0x7fc0u;        // bf16 quiet NaN
}

}  // namespace

#ifdef VSR_ABL
extern "C" int vsr_debug_read_clk_chain(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clk_chain), sizeof(unsigned long long) * 256 * 2) == hipSuccess ? 0 : VSR_ERR_HIP;
}
#endif

// The work-distribution map of the chain kernel on the host (tests: every (layer, tile) must be handed out exactly once, and a
// region's positions must never go back a layer): position P of region `own` of R -> item = layer * tiles + tile, or -1.
extern "C" int vsr_debug_chain_item(unsigned P, int own, int R, int tiles, int ntx, int nlayers) {
    return chain_item(P, own, R, tiles, ntx, nlayers);
}

// dependency waits that were given up (1 s each) since the library was loaded: anything but 0 voids the results of that run
// (the launches concerned have poisoned their outputs: chain_poison_kernel)
extern "C" int vsr_debug_chain_timeouts(unsigned* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chain_timeouts), sizeof(unsigned)) == hipSuccess ? 0 : VSR_ERR_HIP;
}

// Test hook: the next `launches` chain launches of this process start with their error word SET, i.e. as if a wait had been given
// up at once (no wait is honoured, the grid drains, the poison kernel fires) -- the failure mode of a timed-out chain without
// the second of spinning.  Returns the number of forced launches still pending before the call.
static std::atomic<int>& chain_inject() { static std::atomic<int> n{0}; return n; }
extern "C" int vsr_debug_chain_inject_error(int launches) { return chain_inject().exchange(launches < 0 ? 0 : launches); }

size_t vsr_chain_sync_bytes(int nlayers, int N, int H, int W) {
    return 1024 + (size_t)nlayers * N * cdiv(H, PTH) * cdiv(W, PTW) * 16;
}

struct ChainTiming {
    static constexpr int MAX = 64;
    bool on = false; int n = 0;
    hipEvent_t ev[2 * MAX] = {};
    int layers[MAX], variant[MAX]; long long pixels[MAX];
};
static ChainTiming& chain_timing() { static ChainTiming t; return t; }      // process-wide: autograd runs the backward on a thread of its own (a diagnostic: calls are not concurrent)

// a.sync: vsr_chain_sync_bytes() of device memory owned by this launch until it has finished (zeroed here, on the stream)
int vsr_launch_conv3x3_chain(const ChainArgs& a, int num_cus, hipStream_t st) {
    if (!a.base || !a.sync || a.nlayers < 1 || a.nlayers > VSR_CHAIN_MAX_LAYERS || a.N < 1 || a.H < 1 || a.W < 1) return VSR_ERR_BADARG;
    if (pm_image_elems(2 * PTH + 2, a.W, 64) * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;
    const long long tiles = (long long)a.N * cdiv(a.W, PTW) * cdiv(a.H, PTH);
    if (tiles * a.nlayers > 0x3fffffffLL) return VSR_ERR_UNSUPPORTED;
    int even = -1;
    for (int l = 0; l < a.nlayers; ++l) {
        const ChainLayer& L = a.layer[l];
        if (L.variant != CHAIN_RELU && L.variant != CHAIN_SKIP && L.variant != CHAIN_MASK) return VSR_ERR_BADARG;
        if (L.variant != CHAIN_SKIP) { if (even >= 0 && even != L.variant) return VSR_ERR_UNSUPPORTED; even = L.variant; }
        if (L.src == 0xffffffffu || L.dst == 0xffffffffu || L.w == 0xffffffffu) return VSR_ERR_BADARG;
        if ((L.variant == CHAIN_SKIP && L.res == 0xffffffffu) || (L.variant == CHAIN_MASK && L.sbits == 0xffffffffu)) return VSR_ERR_BADARG;
    }
    auto kern = even == CHAIN_MASK ? conv3x3_c64_chain_kernel<CHAIN_MASK> : conv3x3_c64_chain_kernel<CHAIN_RELU>;
    static VsrDevOnce once[2];
    { const int rc = vsr_set_max_dynamic_lds(once[even == CHAIN_MASK], reinterpret_cast<const void*>(kern), P_LDS); if (rc != VSR_OK) return rc; }
    HIP_CHECK_RET(hipMemsetAsync(a.sync, 0, vsr_chain_sync_bytes(a.nlayers, a.N, a.H, a.W), st));
    {
        int left = chain_inject().load();
        while (left > 0 && !chain_inject().compare_exchange_weak(left, left - 1)) {}
        if (left > 0) HIP_CHECK_RET(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.sync + 1), 1, 1, st));
    }
    int gx = num_cus < 1 ? 1 : num_cus;
    if (gx > tiles) gx = (int)tiles;
    ChainTiming& T = chain_timing();
    const int slot = T.on && T.n < ChainTiming::MAX ? T.n : -1;
    if (slot >= 0) HIP_CHECK_RET(hipEventRecord(T.ev[2 * slot], st));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(PNT), P_LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    if (slot >= 0) {
        HIP_CHECK_RET(hipEventRecord(T.ev[2 * slot + 1], st));
        T.layers[slot] = a.nlayers; T.variant[slot] = even; T.pixels[slot] = (long long)a.N * a.H * a.W;
        T.n = slot + 1;
    }
    hipLaunchKernelGGL(chain_poison_kernel, dim3(1), dim3(256), 0, st, a.sync + 1,
                       reinterpret_cast<unsigned short*>(a.base + (unsigned long long)a.layer[a.nlayers - 1].dst * 256ull));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// Timing of the chain launches of the engine, live: between begin and read every chain launch of the process is bracketed by a
// pair of HIP events on the stream it is launched on (bench.py: one extra, un-timed step behind the timed region).
extern "C" int vsr_debug_chain_timing_begin(void) {
    ChainTiming& T = chain_timing();
    if (!T.ev[0])
        for (int i = 0; i < 2 * ChainTiming::MAX; ++i) HIP_CHECK_RET(hipEventCreate(&T.ev[i]));
    T.n = 0; T.on = true;
    return VSR_OK;
}
// us[i], layers[i], variant[i] (CHAIN_RELU forward / CHAIN_MASK backward), pixels[i] of launch i; returns the number of launches
// (the caller has synchronised the stream) or a negative status
extern "C" int vsr_debug_chain_timing_read(float* us, int* layers, int* variant, long long* pixels, int max) {
    ChainTiming& T = chain_timing();
    T.on = false;
    const int n = T.n < max ? T.n : max;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_CHECK_RET(hipEventElapsedTime(&ms, T.ev[2 * i], T.ev[2 * i + 1]));
        us[i] = ms * 1e3f; layers[i] = T.layers[i]; variant[i] = T.variant[i]; pixels[i] = T.pixels[i];
    }
    return n;
}
