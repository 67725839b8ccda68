// Shared device/host declarations for the vsrlab_amd HIP library (gfx950 / CDNA4 only).
//
// Data layout in HBM
//   * boundary tensors (what PyTorch sees): LR clip / SR clip / cotangents are planar
//     fp32 NCHW, parameters and their gradients are fp32 OIHW -- exactly the reference's
//     tensors (basicvsr.py:39-83).
//   * everything internal to the path is "pixel-major", BLOCKED by 32 pixels:
//     [N][H][ceil(W/32)][C/8][32 pixels][8 channels] (pm_off() below), C a multiple of 16,
//     element type T = bf16 (perf build) or fp32 (parity build): the MFMA accumulator layout
//     of the conv kernels, so epilogues store / read residuals with 512-byte wave instructions;
//     an MFMA B-fragment (8 consecutive channels of one pixel) is still one 16-byte piece.
//   * optical flow is planar fp32 [N][2][H][W] (channel 0 = dx), coalesced along x.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VSR_OK 0
#define VSR_ERR_BADARG (-1)
#define VSR_ERR_UNSUPPORTED (-2)
#define VSR_ERR_HIP (-3)
#define VSR_ERR_WORKSPACE (-4)

#define VSR_F32 0
#define VSR_BF16 1

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define VSR_MAX_SRC 4
#define VSR_MAX_Z 4

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
enum { MASK_NONE = 0, MASK_RELU = 1, MASK_LEAKY = 2, MASK_RELU_BITS = 3, MASK_LEAKY_BITS = 4 };   // _BITS: internal to the persistent kernel
enum { EPI_NHWC = 0, EPI_PLANAR = 1 };

// One implicit-GEMM convolution launch (stride 1, "same" zero padding, KS x KS).
// The input is the concatenation of up to 4 sources along the channel axis; each source is
// a strided *view* of an underlying pixel-major image:  view(y,x) = img(y*in_step+oy, x*in_step+ox).
// Up to 4 independent sub-convolutions (z) share the input and differ in weights/bias/output
// placement -- that is how conv+PixelShuffle(2) (core/modules/upsampling.py:10-12) is one launch.
struct ConvArgs {
    const void* src[VSR_MAX_SRC];
    long long src_nstride[VSR_MAX_SRC];  // elements between images of the batch
    int src_oy[VSR_MAX_SRC], src_ox[VSR_MAX_SRC];
    int in_step, Hs, Ws;                 // underlying source image (pixels)
    int N, H, W;                         // batch and conv (= view = output-view) size
    const void* wpack;                   // packed weights, see pack_weights_kernel
    long long w_zstride;                 // elements between the weight sets of consecutive z
    const float* bias;                   // [z][bias_zstride] fp32 or null
    int bias_zstride;
    int nz;
    void* dst[VSR_MAX_Z];
    long long dst_nstride;
    int out_step, Hd, Wd;                // underlying destination image (pixels)
    int out_oy[VSR_MAX_Z], out_ox[VSR_MAX_Z];
    int CD;                              // channels per destination pixel (pixel-major dst)
    int cout_real;                       // real output channels (<= COUT template)
    int act;
    const void* res[VSR_MAX_Z];          // optional residual, destination layout, type T
    const void* aux[VSR_MAX_Z];          // optional activation-mask source, destination layout, type T
    int mask_mode;
    // EPI_PLANAR: fp32 planar destination [N][cout_real][Hd][Wd] (+ optional planar residual,
    // + optional bilinear x base_scale (4, or 2 for upscale = 2; 0 = 4) of a planar LR frame: basicvsr.py:22,82)
    const float* pres;
    const float* base_lr;
    long long base_nstride;
    int base_h, base_w, base_scale;
    // Sign bits of a ReLU output, 1 bit per element, in the persistent kernel's own tile order ([tile][wave][lane] x
    // 8 bytes): written by the bias+ReLU launch (sign_out), read by the masked data-gradient launch of the same
    // tensor shape (sign_bits) INSTEAD of re-reading the bf16 activation `aux` (66 MB -> 4 MB at 540p).  Other kernels
    // ignore both and use `aux`.
    void* sign_out[VSR_MAX_Z];
    const void* sign_bits[VSR_MAX_Z];
    // LeakyReLU slope of ACT_LEAKY / MASK_LEAKY: 0 = the path's 0.1 (basicvsr.py:18,21; conv.py:97); the discriminator
    // uses 0.2 (unet-discriminator.py:19)
    float leaky_slope;
    int planar_c;                        // channels of a planar fp32 source (LASTPLANAR): 0 = 3
    // conv3x3_c64_persist only (r04): the destination (and the residual, which has the destination's layout) is stored PHASE-SEPARATED:
    // pixel (y, x) of the Hd x Wd image goes to plane z = 2 (y & 1) + (x & 1), position (y >> 1, x >> 1) of four Hd/2 x Wd/2 images
    // (each a whole N-image tensor, `unshuffle_plane` elements apart; dst_nstride = one Hd/2 x Wd/2 image).  What a pixel-shuffle
    // layer's gradients read is then four contiguous tensors instead of every second pixel of every second row.  Hd, Wd even.
    int unshuffle;
    long long unshuffle_plane;
};
static inline __host__ __device__ float vsr_slope(float s) { return s != 0.f ? s : 0.1f; }

// Weight-gradient launch: dW[z][tap][cout][cin] = sum_p dY[p][cout] * X[p + tap][cin] over up to
// VSR_WG_MAXSEG (X, dY) segment pairs (the frames of a clip share one launch and one reduction).
#define VSR_WG_MAXSEG 8
struct WgradArgs {
    const void* x[VSR_WG_MAXSEG];
    const void* dy[VSR_WG_MAXSEG];
    long long x_nstride, dy_nstride;
    int nseg;
    int N, H, W;                 // view size (x view == dy view)
    int x_step, x_oy, x_ox, Hx, Wx;   // x view -> underlying image
    int dy_step, dy_oy, dy_ox, Hy, Wy; // dy view -> underlying image
    float* slab;                 // [gridDim.x][slab_stride] fp32 partials
    int tap_begin;               // first tap of this launch (7x7 kernels go one kernel row per launch)
    int x_ctotal, x_coff;        // channels per pixel of the X tensor (0 = CX) and first 8-channel chunk used (64-channel slices of a wider tensor)
    int dy_ctotal, dy_coff;      // the same for dY
    int dy_planar_c;             // channels of a planar fp32 dY (DYPLANAR): 0 = 3
    int slab_stride;
    int ntiles_x, ntiles_y;
    // pair batching (wide layers: one launch covers every (parity view, cout block, cin slice) pair of a layer):
    // workgroup b -> pair = (b >> 3) / (ksplit / 8), pixel part = (b & 7) + 8 * ((b >> 3) % (ksplit / 8)); 0 = off
    int pair_ksplit, pair_nsl, pair_ncob, pair_views;
};

static inline __host__ __device__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- the internal "pixel-major" layout, blocked by 32 pixels ---------------------------------------
//   [N][H][WS = ceil(W/32)][C/8 chunks][32 pixels][8 channels]
// i.e. inside a 32-pixel row segment the 16-byte (bf16) channel chunks of the 32 pixels are contiguous
// (512 B per chunk).  This is exactly the MFMA accumulator layout of the conv kernels (lane = pixel of
// a 32-pixel segment, registers = 4 consecutive channels), so epilogues store and read residuals with
// fully coalesced wave instructions and no transposition; operand tiles are gathered by per-lane
// addresses (LDS-DMA / staging loops), which costs nothing.  Rows are padded to whole segments; the
// padding pixels are never read (loaders bound-check x < W) and never written.
static inline __host__ __device__ int pm_ws(int W) { return (W + 31) >> 5; }
static inline __host__ __device__ long long pm_image_elems(int H, int W, int C) { return (long long)H * pm_ws(W) * 32 * C; }
// element offset of channel chunk `chunk` (8 channels) of pixel (y, x); x may be negative (halo)
static inline __host__ __device__ long long pm_off(int y, int x, int chunk, int W, int C) {
    const int seg = x >> 5;                       // arithmetic shift: floor
    return (((long long)y * pm_ws(W) + seg) * (C >> 3) + chunk) * 256 + (x & 31) * 8;
}

// "Paired-block" channel order of a 64-channel MFMA output whose accumulators are 16x16 blocks (lane = pixel i + 16 q, register j
// = row 4q + j of block mb): dealing channel pm_acc_chan(mb, row) to row `row` of block mb makes lane (i, q) hold, in blocks 2k
// and 2k+1, the 8 consecutive channels of chunk 4k + q of its pixel = one whole 16-byte piece of the blocked layout.
static inline __host__ __device__ int pm_acc_chan(int mb, int row) { return 8 * (4 * (mb >> 1) + (row >> 2)) + 4 * (mb & 1) + (row & 3); }

// XCD-aware work split for persistent kernels: workgroups are dealt round-robin over the 8 XCDs (b % 8 shares an
// XCD, each with its own 4 MiB L2), so XCD x walks the x-th CONTIGUOUS eighth of the tile list: neighbouring tiles,
// whose halos overlap, are then served by one L2.  Speed only: any placement computes every tile exactly once.
struct TileWalk { int first, end, stride; };
static inline __device__ TileWalk xcd_tile_walk(int total, int block, int nblocks) {
    TileWalk w;
    if (nblocks >= 8 && (nblocks & 7) == 0) {
        const int xcd = block & 7, j = block >> 3;
        const int lo = (int)((long long)xcd * total / 8), hi = (int)((long long)(xcd + 1) * total / 8);
        w.first = lo + j; w.end = hi; w.stride = nblocks >> 3;
    } else {
        w.first = block; w.end = total; w.stride = nblocks;
    }
    return w;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process that drives several GPUs must
// apply it on each of them, so "done" is remembered per (kernel, device).  Set twice by two racing threads = harmless.
#define VSR_MAX_DEVICES 32
struct VsrDevOnce { bool done[VSR_MAX_DEVICES] = {}; };
static inline int vsr_set_max_dynamic_lds(VsrDevOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VSR_ERR_HIP;
    if (dev < 0 || dev >= VSR_MAX_DEVICES) return VSR_ERR_UNSUPPORTED;
    if (!__atomic_load_n(&once.done[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return VSR_ERR_HIP;
        __atomic_store_n(&once.done[dev], true, __ATOMIC_RELEASE);
    }
    return VSR_OK;
}

// compute units of the current device (cached per device; hipGetDeviceProperties costs ~a millisecond)
static inline int vsr_num_cus() {
    static int cus[VSR_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VSR_MAX_DEVICES) return 256;
    int v = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        __atomic_store_n(&cus[dev], v, __ATOMIC_RELAXED);
    }
    return v;
}

// A/B and test switches of the library, read from the environment ONCE per process: C++11 initialises a function-local static
// exactly once, also under concurrent first calls (the library is callable from several threads), and no launch path calls getenv.
#include <cstdlib>
struct VsrEnvSwitches {
    bool generic_conv, generic_wgrad, attn_generic, single_stream;
    int wide2_max_wg;
};
static inline const VsrEnvSwitches& vsr_env() {
    static const VsrEnvSwitches s = [] {
        auto on = [](const char* n) { const char* e = getenv(n); return e && e[0] == '1'; };
        VsrEnvSwitches v;
        v.generic_conv = on("VSRLAB_AMD_GENERIC_CONV");
        v.generic_wgrad = on("VSRLAB_AMD_GENERIC_WGRAD");
        v.attn_generic = getenv("VSRLAB_AMD_ATTN_GENERIC") != nullptr;
        v.single_stream = on("VSRLAB_AMD_SINGLE_STREAM");
        const char* e = getenv("VSRLAB_AMD_WIDE2_MAX_WG");
        v.wide2_max_wg = e ? atoi(e) : 0;
        return v;
    }();
    return s;
}

#define HIP_CHECK_RET(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return VSR_ERR_HIP;             \
    } while (0)
