// Weight / bias gradients of SPyNet's 7x7 layers (train_flow, conf/experiment/basic.yaml:7; spynet.py:16-18 under autograd):
//   dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p + (ky-3, kx-3)][ci],   db[co] = sum_p dY[p][co]        (bf16 operands, fp32 sums)
// The generic kernel (wgrad_mfma.hip: wgrad_kernel<.., 7, ..>) runs one kernel ROW per launch, stages a full 14x38 haloed X
// tile and the dY tile through registers for every tile of every launch, and synchronises twice per tile: 52 ms of a 197 ms
// train_flow step at BASELINE config 2 (r03 trace), ~0.1 PFLOP/s.  This is the producer / consumer form of wgrad3x3_c64_pc_kernel
// for the 7x7 shapes:
//   * grid (G, 7): workgroup (j, ky) owns kernel row ky (7 taps = 7 x 2 x 2 accumulator blocks per wave, 112 registers) and walks
//     tiles j, j + G, ...; with G a multiple of 8 the seven workgroups of a j land on ONE XCD (workgroup id mod 8) and walk the
//     same tiles at the same pace, so a tile's operands come from HBM once and from that XCD's L2 six times.
//   * 512 threads: 4 LDS-DMA producer waves fill the other buffer set a tile ahead -- only the 8 X rows this kernel row needs
//     ([8 rows][chunks][44 slots of 16 B]: chunk stride 704 B = 192 mod 256, the conflict-free residue for the transposing
//     reads, like the 3x3 kernel's 576) and the 8x32 dY tile ([8][chunks][35 slots]) -- and sum the bias gradient from the
//     current dY tile (ky = 0 workgroups only); 4 MFMA consumer waves: v_mfma_f32_16x16x32_bf16 with K = the 32 pixels of a tile
//     row, A = dY^T and B = X by ds_read_b64_tr_b16 (B fragment of tap kx = the same row image 16 kx bytes further).
//   * a wave covers a (32 cout, 32 cin) block; layers with fewer blocks than waves split the tile's rows between waves and
//     write one partial slab per row group (the reduction sums them in its fixed order).
// Slab layout = the generic kernel's ([49 taps][COUTP][CXP] + [COUTP] bias sums), so wgrad_reduce_kernel is shared.
#include "common.h"

namespace {

constexpr int GTW = 32, GTH = 8, GNT = 512;
constexpr int GXS = 44, GYS = 35;                        // 16-byte slots per chunk row of the X / dY images

typedef __attribute__((ext_vector_type(4))) unsigned gu32x4_t;

__device__ uint4 g_w7_zero_chunk[2];

__device__ __forceinline__ s16x4_t g_tr_read(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr);
}

template <int CX, int COUT> struct W7 {
    static constexpr int CPX = CX / 8, CPY = COUT / 8;
    static constexpr int NCB = COUT >= 32 ? COUT / 32 : 1, NIB = CX >= 32 ? CX / 32 : 1;
    static constexpr int MB = COUT >= 32 ? 2 : 1, NBK = CX >= 32 ? 2 : 1;        // 16-row / 16-column blocks per wave
    static constexpr int NBLK = NCB * NIB, KSPLIT = 4 / NBLK, ROWS = GTH / KSPLIT;
    static constexpr int COUTP = NCB * 32, CXP = NIB * 32;
    static constexpr int XROW = CPX * GXS * 16, YROW = CPY * GYS * 16;
    static constexpr int XB = GTH * XROW, YB = GTH * YROW, SET = XB + YB;
    static constexpr int X_SLOTS = GTH * CPX * GXS, Y_SLOTS = GTH * CPY * GYS;
    static constexpr int X_PIECES = (X_SLOTS + 63) / 64, Y_PIECES = (Y_SLOTS + 63) / 64;
    static constexpr int NPIECE = (X_PIECES + Y_PIECES + 3) / 4;                  // per producer wave
    static constexpr int LDS = 2 * SET > 256 * 8 * 4 ? 2 * SET : 256 * 8 * 4;
    static_assert(NBLK == 1 || NBLK == 2 || NBLK == 4, "blocks per workgroup");
    static_assert(LDS <= 160 * 1024, "does not fit the LDS of a CU");
};

template <int CX, int COUT>
__global__ __launch_bounds__(GNT, 1) void wgrad7x7_pc_kernel(const WgradArgs a) {
    typedef W7<CX, COUT> K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);            // 0: MFMA consumer, 1: LDS-DMA producer
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
    const int ky = blockIdx.y;
    const int tiles_per_img = a.ntiles_x * a.ntiles_y;
    const int total = a.N * tiles_per_img;
    const int stride = gridDim.x;
    const int WS = pm_ws(a.W);

    if (role == 1) {
        // =================== producers ===================
        const char* zsrc = reinterpret_cast<const char*>(g_w7_zero_chunk);
        const char* xsrc = reinterpret_cast<const char*>(a.x[0]);
        const char* ysrc = reinterpret_cast<const char*>(a.dy[0]);
        // piece = w4 + 4 i: X pieces first, then dY pieces; rel[i] = source byte offset of this lane's slot from the tile origin
        int rel[K::NPIECE];
#pragma unroll
        for (int i = 0; i < K::NPIECE; ++i) {
            const int piece = w4 + 4 * i;
            if (piece < K::X_PIECES) {
                const int idx = piece * 64 + lane;
                const int row = idx / (K::CPX * GXS), rem = idx - row * (K::CPX * GXS);
                const int c = rem / GXS, dx = rem - c * GXS - 3;
                rel[i] = ((((row + ky - 3) * WS + (dx >> 5)) * K::CPX + c) * 256 + (dx & 31) * 8) * 2;
            } else {
                const int idx = (piece - K::X_PIECES) * 64 + lane;
                const int row = idx / (K::CPY * GYS), rem = idx - row * (K::CPY * GYS);
                const int c = rem / GYS, tx = rem - c * GYS;
                rel[i] = (((row * WS) * K::CPY + c) * 256 + tx * 8) * 2;
            }
        }
        auto issue = [&](int tile, int s) {
            const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
            const int ty0 = (r / a.ntiles_x) * GTH, tx0 = (r % a.ntiles_x) * GTW;
            const char* xo = xsrc + ((long long)n * a.x_nstride + pm_off(ty0, tx0, 0, a.W, CX)) * 2;
            const char* yo = ysrc + ((long long)n * a.dy_nstride + pm_off(ty0, tx0, 0, a.W, COUT)) * 2;
            char* ls = smem + s * K::SET;
#pragma unroll
            for (int i = 0; i < K::NPIECE; ++i) {
                const int piece = w4 + 4 * i;
                if (piece < K::X_PIECES) {
                    const int idx = piece * 64 + lane;
                    const int row = idx / (K::CPX * GXS), tx = (idx - row * (K::CPX * GXS)) % GXS;
                    const int vy = ty0 + row + ky - 3, vx = tx0 + tx - 3;
                    const bool valid = vy >= 0 && vy < a.H && vx >= 0 && vx < a.W;
                    const char* src = valid ? xo + rel[i] : zsrc;
                    if (idx < K::X_SLOTS && tx < GTW + 6)          // the pad slots of a row are never read: their lanes load nothing
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(ls + piece * 1024), 16, 0, 0);
                } else if (piece < K::X_PIECES + K::Y_PIECES) {
                    const int idx = (piece - K::X_PIECES) * 64 + lane;
                    const int row = idx / (K::CPY * GYS), tx = (idx - row * (K::CPY * GYS)) % GYS;
                    const bool valid = ty0 + row < a.H && tx0 + tx < a.W;
                    const char* src = valid ? yo + rel[i] : zsrc;
                    if (idx < K::Y_SLOTS && tx < GTW)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(ls + K::XB + (piece - K::X_PIECES) * 1024), 16, 0, 0);
                }
            }
        };
        float bsum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
        const int pt = tid - 256;                              // 0..255: chunk pt % CPY of the pixels pt / CPY + (256 / CPY) i
        int cur = 0, tile = blockIdx.x;
        if (tile < total) issue(tile, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // the first tile is in LDS
        while (tile < total) {
            const int next = tile + stride;
            if (next < total) issue(next, cur ^ 1);
            if (ky == 0) {                                     // bias gradient: column sums of the CURRENT dY tile (inline asm: see wgrad3x3_c64_pc_kernel)
#pragma unroll
                for (int i = 0; i < K::CPY; ++i) {
                    const int p = pt / K::CPY + (256 / K::CPY) * i;
                    const unsigned addr = (unsigned)(cur * K::SET + K::XB + (p >> 5) * K::YROW + (pt % K::CPY) * (GYS * 16) + (p & 31) * 16);
                    gu32x4_t v;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
                    const unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        bsum[2 * jj] += __uint_as_float(vw[jj] << 16);
                        bsum[2 * jj + 1] += __uint_as_float(vw[jj] & 0xffff0000u);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                   // next tile landed; the consumers are done with `cur`
            cur ^= 1;
            tile = next;
        }
        __syncthreads();                                       // (A) consumers have stored their accumulators; LDS is free
        float* red = reinterpret_cast<float*>(smem);           // [256][8]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pt * 8 + j] = bsum[j];
        __syncthreads();                                       // (B)
    } else {
        // =================== consumers ===================
        const int q = lane >> 4, l15 = lane & 15, qq = l15 >> 2, p4 = l15 & 3;
        const int blk = w4 % K::NBLK, ks = w4 / K::NBLK;
        const int cb = blk % K::NCB, ib = blk / K::NCB;
        // transposing-read lane addresses as in wgrad3x3_c64_pc_kernel: lane 4 qq + p4 of 16-lane group q supplies pixel 8 q + qq,
        // channels 4 p4 .. 4 p4 + 3 of the fragment's 16 (chunk p4 >> 1, half p4 & 1)
        const int xoff = (ib * 4 + (p4 >> 1)) * (GXS * 16) + (8 * q + qq) * 16 + (p4 & 1) * 8;
        const int yoff = K::XB + (cb * 4 + (p4 >> 1)) * (GYS * 16) + (8 * q + qq) * 16 + (p4 & 1) * 8;
        typedef union { s16x4_t s[2]; bf16x8_t b; } frag_u;
        f32x4_t acc[7][K::MB][K::NBK];
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int m = 0; m < K::MB; ++m)
#pragma unroll
                for (int n = 0; n < K::NBK; ++n) acc[t][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int cur = 0;
        __syncthreads();                                       // the first tile is in LDS
        for (int tile = blockIdx.x; tile < total; tile += stride) {
            const unsigned lx = (unsigned)(cur * K::SET + xoff), ly = (unsigned)(cur * K::SET + yoff);
#pragma unroll
            for (int rr = 0; rr < K::ROWS; ++rr) {
                const int r = ks * K::ROWS + rr;               // tile row (= LDS row of both images: the X image starts at row ky - 3)
                frag_u A[K::MB];
#pragma unroll
                for (int m = 0; m < K::MB; ++m) {
                    A[m].s[0] = g_tr_read(ly + r * K::YROW + m * (2 * GYS * 16));
                    A[m].s[1] = g_tr_read(ly + r * K::YROW + m * (2 * GYS * 16) + 64);
                }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
                    for (int n = 0; n < K::NBK; ++n) {
                        frag_u B;
                        B.s[0] = g_tr_read(lx + r * K::XROW + n * (2 * GXS * 16) + kx * 16);
                        B.s[1] = g_tr_read(lx + r * K::XROW + n * (2 * GXS * 16) + kx * 16 + 64);
#pragma unroll
                        for (int m = 0; m < K::MB; ++m)
                            acc[kx][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m].b, B.b, acc[kx][m][n], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                   // next tile landed; everybody is done with `cur`
            cur ^= 1;
        }
        // ---- partial slab (blockIdx.x, ks): taps 7 ky .. 7 ky + 6 of [49][COUTP][CXP].  Accumulator (kx, m, n), register j =
        // dW[7 ky + kx][cout cb*32 + m*16 + 4 q + j][cin ib*32 + n*16 + l15] ----
        float* slab = a.slab + (long long)(blockIdx.x * K::KSPLIT + ks) * a.slab_stride;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int m = 0; m < K::MB; ++m)
#pragma unroll
                for (int n = 0; n < K::NBK; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        slab[((long long)(7 * ky + kx) * K::COUTP + cb * 32 + m * 16 + 4 * q + j) * K::CXP + ib * 32 + n * 16 + l15] = acc[kx][m][n][j];
        __syncthreads();                                       // (A)
        __syncthreads();                                       // (B) the producers' bias partial sums are in LDS
        if (ky == 0 && tid < K::COUTP) {                       // bias tail of every slab of this workgroup: the sums in slab (.., 0), zeros in the others
            const float* red = reinterpret_cast<const float*>(smem);
            float s = 0.f;
            if (tid < COUT) {
                const int c = tid >> 3, j = tid & 7;           // producer thread pt summed chunk pt % CPY
                for (int t = c; t < 256; t += K::CPY) s += red[t * 8 + j];
            }
            for (int k2 = 0; k2 < K::KSPLIT; ++k2)
                a.slab[(long long)(blockIdx.x * K::KSPLIT + k2) * a.slab_stride + 49 * K::COUTP * K::CXP + tid] = k2 == 0 ? s : 0.f;
        }
    }
}

template <int CX, int COUT>
int launch_w7(const WgradArgs& a0, int max_slabs, int* nslabs, hipStream_t st) {
    typedef W7<CX, COUT> K;
    auto kern = wgrad7x7_pc_kernel<CX, COUT>;
    static VsrDevOnce once;
    { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), K::LDS); if (rc != VSR_OK) return rc; }
    WgradArgs a = a0;
    a.ntiles_x = cdiv(a.W, GTW);
    a.ntiles_y = cdiv(a.H, GTH);
    const int tiles = a.N * a.ntiles_x * a.ntiles_y;
    int G = vsr_num_cus() / 8;                                 // 7 kernel rows x G workgroups <= one workgroup per CU
    G &= ~7;                                                   // the seven workgroups of a j on one XCD (see the header)
    if (G < 8) G = 8;
    if (G > tiles) G = tiles;
    if (G * K::KSPLIT > max_slabs) G = max_slabs / K::KSPLIT;
    if (G < 1) return VSR_ERR_UNSUPPORTED;
    *nslabs = G * K::KSPLIT;
    hipLaunchKernelGGL(kern, dim3(G, 7), dim3(GNT), K::LDS, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // namespace

// bf16, 7x7, one (X, dY) pair of pixel-major tensors at unit step with CX / COUT channels per pixel; slab: room for max_slabs
// partials of vsr_wgrad_slab_dims(7, cx, cout) floats; *nslabs = partials written.  VSR_ERR_UNSUPPORTED: use the generic kernel.
int vsr_launch_wgrad7x7_pc(int cx, int cout, const WgradArgs& a, int max_slabs, int* nslabs, hipStream_t st) {
    if (a.nseg != 1 || a.x_step != 1 || a.dy_step != 1 || a.x_oy || a.x_ox || a.dy_oy || a.dy_ox || a.Hx != a.H || a.Wx != a.W ||
        a.Hy != a.H || a.Wy != a.W || a.x_ctotal || a.dy_ctotal || a.x_coff || a.dy_coff || !a.x[0] || !a.dy[0] || !a.slab || !nslabs)
        return VSR_ERR_UNSUPPORTED;
    if (pm_image_elems(GTH + 8, a.W, cx > cout ? cx : cout) * 2 > 0x7fffffffLL) return VSR_ERR_UNSUPPORTED;      // in-tile offsets are 32-bit
#define W7_CASE(CX, COUT) if (cx == CX && cout == COUT) return launch_w7<CX, COUT>(a, max_slabs, nslabs, st);
    W7_CASE(16, 32)     // 8 -> 32
    W7_CASE(32, 64)     // 32 -> 64
    W7_CASE(64, 32)     // 64 -> 32
    W7_CASE(32, 16)     // 32 -> 16
    W7_CASE(16, 16)     // 16 -> 2 (dY = the masked flow gradient, 16-channel padded)
#undef W7_CASE
    return VSR_ERR_UNSUPPORTED;
}
