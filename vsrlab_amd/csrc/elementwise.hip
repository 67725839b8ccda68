// Bandwidth-bound kernels of the BasicVSR path: flow_warp (forward gather / backward scatter),
// the SPyNet pyramid plumbing (resize, normalise, average-pool, per-level warp+concat, flow
// upsample), weight packing and small glue.  All are pure streaming kernels: one pass, 16-byte
// accesses on pixel-major tensors, planar fp32 accesses coalesced along x.
#include "kernels.h"

namespace {

template <typename T> struct EW;
template <> struct EW<bf16_t> { typedef uint4 chunk_t; };
struct echunk32_t { uint4 a, b; };
template <> struct EW<float> { typedef echunk32_t chunk_t; };

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    union { uint4 u; bf16_t h[8]; } t; t.u = v;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)t.h[j];
}
__device__ __forceinline__ void unpack8(const echunk32_t& v, float* f) {
    f[0] = __uint_as_float(v.a.x); f[1] = __uint_as_float(v.a.y); f[2] = __uint_as_float(v.a.z); f[3] = __uint_as_float(v.a.w);
    f[4] = __uint_as_float(v.b.x); f[5] = __uint_as_float(v.b.y); f[6] = __uint_as_float(v.b.z); f[7] = __uint_as_float(v.b.w);
}
__device__ __forceinline__ void pack8(const float* f, uint4& v) {
    union { uint4 u; bf16_t h[8]; } t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.h[j] = (bf16_t)f[j];
    v = t.u;
}
__device__ __forceinline__ void pack8(const float* f, echunk32_t& v) {
    v.a = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    v.b = make_uint4(__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7]));
}

// Sample position of flow_warp (spynet.py:95-106).  The reference normalises the pixel grid to
// [-1,1] and grid_sample(align_corners=True) maps it back; we reproduce that round trip so the
// fp32 rounding of the coordinate matches the reference's.
__device__ __forceinline__ float warp_coord(float base, float flow, int dim) {
    const float d = (float)(dim - 1 > 1 ? dim - 1 : 1);
    const float g = 2.0f * (base + flow) / d - 1.0f;
    return (g + 1.0f) * 0.5f * (float)(dim - 1);
}

// out[n,y,x,:] = bilinear(in[n], (x+fx, y+fy)), zeros padding (each out-of-image tap contributes 0).
template <typename T>
__global__ void warp_fwd_kernel(const T* __restrict__ in, const float* __restrict__ flow, T* __restrict__ out,
                                int N, int H, int W, int C, long long flow_nstride, int border) {
    typedef typename EW<T>::chunk_t chunk_t;
    const int CP = C / 8;
    const long long total = (long long)N * H * W * CP;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CP);
        const long long pix = idx / CP;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        const float* fp = flow + (long long)n * flow_nstride + (long long)y * W + x;
        float px = warp_coord((float)x, fp[0], W);
        float py = warp_coord((float)y, fp[(long long)H * W], H);
        // padding_mode='border' (grid_sample): the coordinate is clamped to the image; a clamped coordinate has no
        // gradient w.r.t. the flow (ATen clip_coordinates_set_grad)
        const bool in_x = !border || (px >= 0.f && px <= (float)(W - 1)), in_y = !border || (py >= 0.f && py <= (float)(H - 1));
        if (border) { px = fminf(fmaxf(px, 0.f), (float)(W - 1)); py = fminf(fmaxf(py, 0.f), (float)(H - 1)); }
        (void)in_x; (void)in_y;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        float accv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) accv[j] = 0.f;
        const T* img = in + (long long)n * pm_image_elems(H, W, C);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
            const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
            if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
                float f[8];
                unpack8(*reinterpret_cast<const chunk_t*>(img + pm_off(yi, xi, c, W, C)), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) accv[j] += f[j] * wgt;
            }
        }
        chunk_t o;
        pack8(accv, o);
        *reinterpret_cast<chunk_t*>(out + (long long)n * pm_image_elems(H, W, C) + pm_off(y, x, c, W, C)) = o;
    }
}

// Backward of the above w.r.t. the warped tensor: scatter-add dOut into an fp32 accumulator
// (grid_sampler_2d_backward).  One lane per channel => every wave-level atomic is 256 contiguous
// bytes, the shape that runs at the full memory-side atomic rate on gfx950.
template <typename T>
__global__ void warp_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ flow, float* __restrict__ dacc,
                                int N, int H, int W, int C, long long flow_nstride, int border) {
    const long long total = (long long)N * H * W * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long long pix = idx / C;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        const float* fp = flow + (long long)n * flow_nstride + (long long)y * W + x;
        float px = warp_coord((float)x, fp[0], W);
        float py = warp_coord((float)y, fp[(long long)H * W], H);
        // padding_mode='border' (grid_sample): the coordinate is clamped to the image; a clamped coordinate has no
        // gradient w.r.t. the flow (ATen clip_coordinates_set_grad)
        const bool in_x = !border || (px >= 0.f && px <= (float)(W - 1)), in_y = !border || (py >= 0.f && py <= (float)(H - 1));
        if (border) { px = fminf(fmaxf(px, 0.f), (float)(W - 1)); py = fminf(fmaxf(py, 0.f), (float)(H - 1)); }
        (void)in_x; (void)in_y;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float g = (float)dout[(long long)n * pm_image_elems(H, W, C) + pm_off(y, x, c >> 3, W, C) + (c & 7)];
        float* img = dacc + (long long)n * H * W * C;              // the fp32 accumulator stays plain [N][H][W][C]: 256-byte atomics
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
            const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
            if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(img + ((long long)yi * W + xi) * C + c, g * wgt);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gather form of the same backward for the propagation warps inside the engine (round 2).  The scatter above pushes
// 4 taps x 64 channels x 4 B of fp32 atomics per pixel through the memory-side atomic units (1.3 TB/s: 450 us per 540p
// frame) into an accumulator that then needs a memset and a read-back.  Optical flows are smooth and a few pixels long, so
// the adjoint is computed from the DESTINATION side instead:
//     dIn[yi][xi] = sum over sources (y, x) whose sample position (x + fx, y + fy) lies within one pixel of (xi, yi)
//                   of  w(y, x -> yi, xi) * dOut[y][x]
//   * a workgroup owns an 8 x 32 tile of dIn; the sample positions of the sources in the tile + R halo go to LDS once;
//   * phase 1: one thread per destination pixel scans the (2R+1)^2 candidate sources and records its hits (source, weight)
//     in a short LDS list (bilinear footprint: ~4 hits per pixel);
//   * phase 2: 8 lanes per destination pixel (one 16-byte channel chunk each) walk the list, accumulate in fp32 registers
//     in a FIXED order and write T(dtop + sum) -- the add_cast of the scatter form is fused, nothing is atomic, the result is
//     bitwise reproducible;
//   * sources displaced by more than R - 1 pixels ("far") are skipped here and scattered with atomics by
//     warp_bwd_far_kernel into the accumulator S, which this kernel adds (and re-zeroes) only when the launch's
//     far counter is non-zero: correct for any flow, fast for the flows SPyNet produces.  S holds 64-bit FIXED-POINT sums
//     (2^-44 units, |sum| < 5.2e5): integer addition is associative, so the scattered share is bit-identical from run to run
//     whatever the arrival order of the atomics (round 2 added floats: trunk gradients wobbled whenever a flow was far).
// ---------------------------------------------------------------------------------------------------------------
constexpr int GR = 5;                                   // search radius: sources within +-5 pixels; |flow| <= 4 is "near"
constexpr int GTH = 8, GTW = 32, GCAP = 8;
constexpr int GHH = GTH + 2 * GR, GHW = GTW + 2 * GR;   // 18 x 42 sources per tile
constexpr double FAR_FX = 17592186044416.0;             // 2^44: fixed-point unit of the far accumulator
// A far contribution that the fixed-point accumulator cannot hold (a NaN / Inf cotangent, |g w| >= 2.5e5) must not turn into finite
// garbage (the fp32 atomics this replaced propagated it, and FusedAdam's skip-on-non-finite-norm relies on that): the far kernel
// sets this bit in the launch's far counter instead of adding, and the gather kernel then writes NaN to EVERY pixel of its output
// (coarser than grid_sampler_2d_backward, which would poison the four destinations only; what matters downstream is that the
// gradient norm is not finite).
constexpr int FAR_BAD = 0x40000000;

__device__ __forceinline__ bool flow_is_far(float fx, float fy) {
    return !(fabsf(fx) <= (float)(GR - 1) && fabsf(fy) <= (float)(GR - 1));      // NaN counts as far
}

template <typename T>
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const T* __restrict__ dout, const float* __restrict__ flow, const T* __restrict__ dtop,
                                                              long long* __restrict__ S, const int* __restrict__ far_count, T* __restrict__ out,
                                                              int N, int H, int W, long long flow_nstride) {
    constexpr int C = 64;
    typedef typename EW<T>::chunk_t chunk_t;
    __shared__ float2 pos[GHH * GHW];                   // sample position of each source of the haloed tile; x = NaN: far / outside
    __shared__ unsigned short hit_src[256 * GCAP];
    __shared__ float hit_w[256 * GCAP];
    __shared__ int hit_n[256];
    const int tid = threadIdx.x;
    const int ntx = cdiv(W, GTW), nty = cdiv(H, GTH);
    const int tile = blockIdx.x;
    const int n = tile / (ntx * nty), tr = tile - n * ntx * nty;
    const int ty0 = (tr / ntx) * GTH, tx0 = (tr % ntx) * GTW;
    const float* fbase = flow + (long long)n * flow_nstride;
    const long long img = (long long)n * pm_image_elems(H, W, C);
    for (int i = tid; i < GHH * GHW; i += 256) {
        const int sy = ty0 - GR + i / GHW, sx = tx0 - GR + i % GHW;
        float2 p = make_float2(__int_as_float(0x7fc00000), 0.f);
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const float fx = fbase[(long long)sy * W + sx], fy = fbase[(long long)H * W + (long long)sy * W + sx];
            if (!flow_is_far(fx, fy)) p = make_float2(warp_coord((float)sx, fx, W), warp_coord((float)sy, fy, H));
        }
        pos[i] = p;
    }
    __syncthreads();
    {   // phase 1: hits of destination pixel tid
        const int dy = tid >> 5, dx = tid & 31;
        const int yi = ty0 + dy, xi = tx0 + dx;
        int cnt = 0;
        if (yi < H && xi < W) {
            for (int sy = 0; sy <= 2 * GR; ++sy)
                for (int sx = 0; sx <= 2 * GR; ++sx) {
                    const int si = (dy + sy) * GHW + dx + sx;
                    const float2 p = pos[si];
                    const float fx0 = floorf(p.x), fy0 = floorf(p.y);
                    const int ox = xi - (int)fx0, oy = yi - (int)fy0;          // 0 or 1 for a hit (NaN -> no hit)
                    if (p.x == p.x && (unsigned)ox <= 1u && (unsigned)oy <= 1u) {
                        const float wx1 = p.x - fx0, wy1 = p.y - fy0;
                        const float wgt = (ox ? wx1 : 1.f - wx1) * (oy ? wy1 : 1.f - wy1);
                        if (cnt < GCAP) { hit_src[tid * GCAP + cnt] = (unsigned short)si; hit_w[tid * GCAP + cnt] = wgt; }
                        ++cnt;
                    }
                }
        }
        hit_n[tid] = cnt;
    }
    __syncthreads();
    const int far_n = far_count[0];
    const bool add_far = far_n != 0;
    const bool far_bad = (far_n & FAR_BAD) != 0;        // a far contribution was not representable (NaN / Inf / huge cotangent)
    const int c = tid & 7;
    for (int pass = 0; pass < 8; ++pass) {              // phase 2: 32 destination pixels per pass, 8 lanes (chunks) each
        const int d = pass * 32 + (tid >> 3);
        const int dy = d >> 5, dx = d & 31;
        const int yi = ty0 + dy, xi = tx0 + dx;
        if (yi >= H || xi >= W) continue;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int cnt = hit_n[d];
        if (cnt <= GCAP) {
            for (int k = 0; k < cnt; ++k) {
                const int si = hit_src[d * GCAP + k];
                const float wgt = hit_w[d * GCAP + k];
                const int sy = ty0 - GR + si / GHW, sx = tx0 - GR + si % GHW;
                float f[8];
                unpack8(*reinterpret_cast<const chunk_t*>(dout + img + pm_off(sy, sx, c, W, C)), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += wgt * f[j];
            }
        } else {                                        // more sources converge on this pixel than the list holds: rescan (rare)
            for (int sy = 0; sy <= 2 * GR; ++sy)
                for (int sx = 0; sx <= 2 * GR; ++sx) {
                    const int si = (dy + sy) * GHW + dx + sx;
                    const float2 p = pos[si];
                    const float fx0 = floorf(p.x), fy0 = floorf(p.y);
                    const int ox = xi - (int)fx0, oy = yi - (int)fy0;
                    if (p.x == p.x && (unsigned)ox <= 1u && (unsigned)oy <= 1u) {
                        const float wx1 = p.x - fx0, wy1 = p.y - fy0;
                        const float wgt = (ox ? wx1 : 1.f - wx1) * (oy ? wy1 : 1.f - wy1);
                        float f[8];
                        unpack8(*reinterpret_cast<const chunk_t*>(dout + img + pm_off(ty0 - GR + dy + sy, tx0 - GR + dx + sx, c, W, C)), f);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += wgt * f[j];
                    }
                }
        }
        if (add_far) {                                  // the atomically scattered share; S goes back to all-zero
            long long* sp = S + (((long long)n * H + yi) * W + xi) * C + c * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += (float)((double)sp[j] * (1.0 / FAR_FX)); sp[j] = 0; }
            if (far_bad) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_nanf("");
            }
        }
        const long long o = img + pm_off(yi, xi, c, W, C);
        if (dtop) {
            float t[8];
            unpack8(*reinterpret_cast<const chunk_t*>(dtop + o), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t[j];
        }
        chunk_t r;
        pack8(acc, r);
        *reinterpret_cast<chunk_t*>(out + o) = r;
    }
}

// the "far" sources of the gather form: |flow| > GR - 1 (or non-finite): scattered with atomics like warp_bwd_kernel, and counted
template <typename T>
__global__ void warp_bwd_far_kernel(const T* __restrict__ dout, const float* __restrict__ flow, long long* __restrict__ S, int* __restrict__ far_count,
                                    int N, int H, int W, long long flow_nstride) {
    constexpr int C = 64;
    const long long total = (long long)N * H * W;
    const int lane = threadIdx.x & 63;
    // a wave scans 64 consecutive pixels (one flow vector per lane, coalesced) and then serves the far ones among them -- usually none --
    // one at a time with lane = channel (r02: one wave per pixel cost 81 us per 540p launch with nothing to do)
    for (long long base = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; base < total; base += (long long)gridDim.x * (blockDim.x >> 6) * 64) {
        const long long mine = base + lane;
        float mfx = 0.f, mfy = 0.f;
        if (mine < total) {
            const int mn = (int)(mine / ((long long)W * H));
            const long long hw = mine - (long long)mn * W * H;
            const float* fq = flow + (long long)mn * flow_nstride + hw;
            mfx = fq[0]; mfy = fq[(long long)H * W];
        }
        unsigned long long todo = __ballot(mine < total && flow_is_far(mfx, mfy));
        while (todo) {
        const int b = __builtin_ctzll(todo);
        todo &= todo - 1;
        const long long pix = base + b;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        const float fx = __shfl(mfx, b, 64), fy = __shfl(mfy, b, 64);
        const int c = threadIdx.x & 63;
        if (c == 0) atomicAdd(far_count, 1);
        const float px = warp_coord((float)x, fx, W), py = warp_coord((float)y, fy, H);
        if (!(px == px && py == py)) continue;
        const float fx0 = floorf(px), fy0 = floorf(py);
        if (fabsf(fx0) > 1e9f || fabsf(fy0) > 1e9f) continue;
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float g = (float)dout[(long long)n * pm_image_elems(H, W, C) + pm_off(y, x, c >> 3, W, C) + (c & 7)];
        long long* imgp = S + (long long)n * H * W * C;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
            const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
            if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
                const float gw = g * wgt;
                if (!(fabsf(gw) < 2.5e5f)) atomicOr(far_count, FAR_BAD);                          // NaN, Inf or beyond the fixed-point range
                else atomicAdd(reinterpret_cast<unsigned long long*>(imgp + ((long long)yi * W + xi) * C + c),
                               (unsigned long long)__double2ll_rn((double)gw * FAR_FX));       // two's complement: signed sums
            }
        }
        }
    }
}

// out = T(a + s)   a, out: blocked pixel-major T (a may be null); s: plain [N][H][W][C] fp32 (the warp scatter
// accumulator) or null.  One thread per (pixel, 8-channel chunk).
template <typename T>
__global__ void add_cast_kernel(const T* __restrict__ a, const float* __restrict__ s, T* __restrict__ out, int N, int H, int W, int C) {
    typedef typename EW<T>::chunk_t chunk_t;
    const int CP = C / 8;
    const long long total = (long long)N * H * W * CP;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CP);
        const long long pix = idx / CP;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        const long long o = (long long)n * pm_image_elems(H, W, C) + pm_off(y, x, c, W, C);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a) unpack8(*reinterpret_cast<const chunk_t*>(a + o), v);
        if (s) {
            const float4* sp = reinterpret_cast<const float4*>(s + pix * C + c * 8);
            const float4 s0 = sp[0], s1 = sp[1];
            v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
        }
        chunk_t r;
        pack8(v, r);
        *reinterpret_cast<chunk_t*>(out + o) = r;
    }
}

// planar fp32 -> pixel-major T with C channels (c >= Cin are zero); and back.
template <typename T>
__global__ void planar_to_pm_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int Cin, int H, int W, int C) {
    const long long total = (long long)N * H * W * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long long pix = idx / C;
        const long long hw = pix % ((long long)H * W);
        const int n = (int)(pix / ((long long)H * W));
        const int y = (int)(hw / W), x = (int)(hw % W);
        out[(long long)n * pm_image_elems(H, W, C) + pm_off(y, x, c >> 3, W, C) + (c & 7)] =
            (T)(c < Cin ? in[((long long)n * Cin + c) * H * W + hw] : 0.f);
    }
}
template <typename T>
__global__ void pm_to_planar_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int Cout, int H, int W, int C) {
    const long long total = (long long)N * Cout * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long hw = idx % ((long long)H * W);
        const int c = (int)((idx / ((long long)H * W)) % Cout);
        const int n = (int)(idx / ((long long)H * W * Cout));
        out[idx] = (float)in[(long long)n * pm_image_elems(H, W, C) + pm_off((int)(hw / W), (int)(hw % W), c >> 3, W, C) + (c & 7)];
    }
}

// ---- SPyNet plumbing (RealBasicVSR/modules/spynet.py:38-93) -----------------------------------
__device__ __forceinline__ void src_index(int d, float scale, int in_size, bool align, int& i0, int& i1, float& l1) {
    float s = align ? d * scale : (d + 0.5f) * scale - 0.5f;
    if (!align && s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// frames (F,3,h,w) -> ((bilinear resize to (hu,wu), align_corners=False) - mean) / std   spynet.py:72-80,40-41
__global__ void resize_norm_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ mean,
                                   const float* __restrict__ std, int F, int h, int w, int hu, int wu) {
    const long long total = (long long)F * 3 * hu * wu;
    const float sy = (float)h / (float)hu, sx = (float)w / (float)wu;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % wu);
        const int y = (int)((idx / wu) % hu);
        const int c = (int)((idx / ((long long)wu * hu)) % 3);
        const long long f = idx / ((long long)wu * hu * 3);
        int y0, y1, x0, x1; float ly, lx;
        src_index(y, sy, h, false, y0, y1, ly);
        src_index(x, sx, w, false, x0, x1, lx);
        const float* p = in + (f * 3 + c) * (long long)h * w;
        const float v = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) +
                        ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
        out[idx] = (v - mean[c]) / std[c];
    }
}

// 2x2 average pool, planar (spynet.py:44-45)
__global__ void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const long long total = planes * ho * wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % wo);
        const int y = (int)((idx / wo) % ho);
        const long long pl = idx / ((long long)wo * ho);
        const float* p = in + pl * h * w + (long long)(2 * y) * w + 2 * x;
        out[idx] = 0.25f * (p[0] + p[1] + p[w] + p[w + 1]);
    }
}

// One pyramid level's network input (spynet.py:51-63), for all frame pairs at once:
//   flow_up = level 0 ? 0 : 2 * bilinear_x2(flow_prev, align_corners=True)
//   x16[p] = [ref(3) | warp(supp, flow_up, border)(3) | flow_up(2) | 0 x 8]      pixel-major, 16 channels
// Pair p of P = 2*n*(t-1): p < P/2 -> backward flow (ref = frame i, supp = i+1), else forward
// (ref = frame i+1, supp = i)  (basicvsr.py:32-35).
template <typename T>
__global__ void spynet_prepare_kernel(const float* __restrict__ frames, const float* __restrict__ flow_prev,
                                      float* __restrict__ flow_up, T* __restrict__ x16, int n, int t, int P, int pair_mode,
                                      int h, int w, int level0) {
    const long long total = (long long)P * h * w;
    const int hp = h / 2, wp = w / 2;
    const float sy = hp > 1 ? (float)(hp - 1) / (float)(h - 1) : 0.f;   // align_corners=True, out = 2*in
    const float sx = wp > 1 ? (float)(wp - 1) / (float)(w - 1) : 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int p = (int)(idx / ((long long)w * h));
        int fref, fsup;
        if (pair_mode) {            // frames = [ref_0..ref_{P-1}, supp_0..supp_{P-1}]
            fref = p; fsup = P + p;
        } else {
            const int half = P / 2;
            const int q = p < half ? p : p - half;
            const int b = q / (t - 1), i = q % (t - 1);
            fref = p < half ? b * t + i : b * t + i + 1;
            fsup = p < half ? b * t + i + 1 : b * t + i;
        }
        float fu[2] = {0.f, 0.f};
        if (!level0) {
            int y0, y1, x0, x1; float ly, lx;
            src_index(y, sy, hp, true, y0, y1, ly);
            src_index(x, sx, wp, true, x0, x1, lx);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float* fp = flow_prev + ((long long)p * 2 + c) * hp * wp;
                fu[c] = 2.0f * ((1.f - ly) * ((1.f - lx) * fp[y0 * wp + x0] + lx * fp[y0 * wp + x1]) +
                                ly * ((1.f - lx) * fp[y1 * wp + x0] + lx * fp[y1 * wp + x1]));
            }
        }
        const long long hw = (long long)h * w;
        flow_up[((long long)p * 2 + 0) * hw + (long long)y * w + x] = fu[0];
        flow_up[((long long)p * 2 + 1) * hw + (long long)y * w + x] = fu[1];
        // border-mode warp of the supporting frame
        float px = warp_coord((float)x, fu[0], w), py = warp_coord((float)y, fu[1], h);
        px = fminf(fmaxf(px, 0.f), (float)(w - 1));
        py = fminf(fmaxf(py, 0.f), (float)(h - 1));
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const int x1 = x0 + 1 < w ? x0 + 1 : w - 1, y1 = y0 + 1 < h ? y0 + 1 : h - 1;   // weight of a clamped tap is 0
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = frames[((long long)fref * 3 + c) * hw + (long long)y * w + x];
            const float* sp = frames + ((long long)fsup * 3 + c) * hw;
            v[3 + c] = wy0 * (wx0 * sp[y0 * w + x0] + wx1 * sp[y0 * w + x1]) + wy1 * (wx0 * sp[y1 * w + x0] + wx1 * sp[y1 * w + x1]);
        }
        v[6] = fu[0]; v[7] = fu[1];
        typename EW<T>::chunk_t lo, hi;
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        pack8(v, lo);
        pack8(z, hi);
        T* o = x16 + (long long)p * pm_image_elems(h, w, 16);
        *reinterpret_cast<typename EW<T>::chunk_t*>(o + pm_off(y, x, 0, w, 16)) = lo;
        *reinterpret_cast<typename EW<T>::chunk_t*>(o + pm_off(y, x, 1, w, 16)) = hi;
    }
}

// flow at (hu,wu) -> (h,w) bilinear align_corners=False, u *= w/wu, v *= h/hu   (spynet.py:83-91)
__global__ void flow_out_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int hu, int wu, int h, int w) {
    const long long total = (long long)P * 2 * h * w;
    const float sy = (float)hu / (float)h, sx = (float)wu / (float)w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int c = (int)((idx / ((long long)w * h)) % 2);
        const long long p = idx / ((long long)w * h * 2);
        int y0, y1, x0, x1; float ly, lx;
        src_index(y, sy, hu, false, y0, y1, ly);
        src_index(x, sx, wu, false, x0, x1, lx);
        const float* ip = in + (p * 2 + c) * (long long)hu * wu;
        const float v = (1.f - ly) * ((1.f - lx) * ip[y0 * wu + x0] + lx * ip[y0 * wu + x1]) +
                        ly * ((1.f - lx) * ip[y1 * wu + x0] + lx * ip[y1 * wu + x1]);
        out[idx] = v * (c == 0 ? (float)w / (float)wu : (float)h / (float)hu);
    }
}

// ---- train_flow: gradients w.r.t. the optical flow (spynet.py:95-106 backward, basicvsr.py:25-28) --------------
// d flow of flow_warp(in, flow) with zeros padding (the propagation warps): grid_sampler_2d_backward's grid
// gradient; with align_corners=True and the reference's normalisation d(sample x)/d(flow x) = 1 (0 if W == 1).
//   dflow[n][0][y][x] = sum_c dout[c] * ( wy0*(in[y0][x1]-in[y0][x0]) + wy1*(in[y1][x1]-in[y1][x0]) )   (OOB taps = 0)
// One thread per pixel, looping over the 8-channel chunks.
template <typename T>
__global__ void warp_bwd_flow_kernel(const T* __restrict__ in, const T* __restrict__ dout, const float* __restrict__ flow,
                                     float* __restrict__ dflow, int N, int H, int W, int C, long long flow_nstride, int border) {
    typedef typename EW<T>::chunk_t chunk_t;
    const int CP = C / 8;
    const long long total = (long long)N * H * W;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        const float* fp = flow + (long long)n * flow_nstride + (long long)y * W + x;
        float px = warp_coord((float)x, fp[0], W);
        float py = warp_coord((float)y, fp[(long long)H * W], H);
        // padding_mode='border' (grid_sample): the coordinate is clamped to the image; a clamped coordinate has no
        // gradient w.r.t. the flow (ATen clip_coordinates_set_grad)
        const bool in_x = !border || (px >= 0.f && px <= (float)(W - 1)), in_y = !border || (py >= 0.f && py <= (float)(H - 1));
        if (border) { px = fminf(fmaxf(px, 0.f), (float)(W - 1)); py = fminf(fmaxf(py, 0.f), (float)(H - 1)); }
        (void)in_x; (void)in_y;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const T* img = in + (long long)n * pm_image_elems(H, W, C);
        const T* dimg = dout + (long long)n * pm_image_elems(H, W, C);
        float gx = 0.f, gy = 0.f;
        for (int c = 0; c < CP; ++c) {
            float g[8], v[4][8];
            unpack8(*reinterpret_cast<const chunk_t*>(dimg + pm_off(y, x, c, W, C)), g);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
                if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
                    unpack8(*reinterpret_cast<const chunk_t*>(img + pm_off(yi, xi, c, W, C)), v[t]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[t][j] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gx += g[j] * (wy0 * (v[1][j] - v[0][j]) + wy1 * (v[3][j] - v[2][j]));
                gy += g[j] * (wx0 * (v[2][j] - v[0][j]) + wx1 * (v[3][j] - v[1][j]));
            }
        }
        float* dp = dflow + (long long)n * flow_nstride + (long long)y * W + x;
        dp[0] = (W > 1 && in_x) ? gx : 0.f;
        dp[(long long)H * W] = (H > 1 && in_y) ? gy : 0.f;
    }
}

// d(residue) of a SPyNet level as the 16-channel pixel-major dY of its last conv: the layer ends in a ReLU
// (spynet.py:16-18), flow = flow_up + residue (spynet.py:65):  out[p][y][x][c] = dflow[p][c][y][x] * (res > 0), c < 2
template <typename T>
__global__ void spynet_dres_kernel(const float* __restrict__ dflow, const float* __restrict__ res, T* __restrict__ out,
                                   int P, int h, int w) {
    const long long total = (long long)P * h * w;
    const long long hw = (long long)h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long long p = idx / hw;
        const long long o = (p * 2) * hw + (long long)y * w + x;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        v[0] = (!res || res[o] > 0.f) ? dflow[o] : 0.f;             // res == NULL: no ReLU behind the last conv (the canonical SPyNet of the VRT tree)
        v[1] = (!res || res[o + hw] > 0.f) ? dflow[o + hw] : 0.f;
        typename EW<T>::chunk_t lo, hi;
        pack8(v, lo);
        pack8(z, hi);
        T* ob = out + p * pm_image_elems(h, w, 16);
        *reinterpret_cast<typename EW<T>::chunk_t*>(ob + pm_off(y, x, 0, w, 16)) = lo;
        *reinterpret_cast<typename EW<T>::chunk_t*>(ob + pm_off(y, x, 1, w, 16)) = hi;
    }
}

// Backward of spynet_prepare_kernel w.r.t. the previous level's flow.  Per fine pixel:
//   g = dflow_l (flow = flow_up + residue)  +  dx16[6..7] (flow_up is a network input)
//     + d/d(flow_up) of the border-mode warp of the supporting frame, contracted with dx16[3..5]
//   dflow_prev += 2 * bilinear_x2^T(g)          (atomics; dflow_prev is zeroed by the caller)
//   dframes (optional, (F,3,h,w) fp32, zeroed by the caller): d ref += dx16[0..2];  d supp += warp^T(dx16[3..5])
//   (atomics: a frame is the reference of one pair and the supporting frame of another)
template <typename T>
__global__ void spynet_prepare_bwd_kernel(const T* __restrict__ dx16, const float* __restrict__ dflow_l,
                                          const float* __restrict__ frames, const float* __restrict__ flow_up,
                                          float* __restrict__ dflow_prev, float* __restrict__ dframes, int n, int t, int P,
                                          int pair_mode, int h, int w) {
    const long long total = (long long)P * h * w;
    const int hp = h / 2, wp = w / 2;
    const float sy = hp > 1 ? (float)(hp - 1) / (float)(h - 1) : 0.f;
    const float sx = wp > 1 ? (float)(wp - 1) / (float)(w - 1) : 0.f;
    const long long hw = (long long)h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int p = (int)(idx / hw);
        int fsup, fref;
        if (pair_mode) {
            fref = p; fsup = P + p;
        } else {
            const int half = P / 2;
            const int q = p < half ? p : p - half;
            const int b = q / (t - 1), i = q % (t - 1);
            fref = p < half ? b * t + i : b * t + i + 1;
            fsup = p < half ? b * t + i + 1 : b * t + i;
        }
        float d[8];
        unpack8(*reinterpret_cast<const typename EW<T>::chunk_t*>(dx16 + (long long)p * pm_image_elems(h, w, 16) + pm_off(y, x, 0, w, 16)), d);
        const float fu0 = flow_up ? flow_up[((long long)p * 2 + 0) * hw + (long long)y * w + x] : 0.f;   // level 0: flow_up = 0
        const float fu1 = flow_up ? flow_up[((long long)p * 2 + 1) * hw + (long long)y * w + x] : 0.f;
        float px = warp_coord((float)x, fu0, w), py = warp_coord((float)y, fu1, h);
        // border padding: clip_coordinates_set_grad (the gradient is 0 where the coordinate was clipped)
        const float mx = (px <= 0.f || px >= (float)(w - 1)) ? 0.f : 1.f;
        const float my = (py <= 0.f || py >= (float)(h - 1)) ? 0.f : 1.f;
        px = fminf(fmaxf(px, 0.f), (float)(w - 1));
        py = fminf(fmaxf(py, 0.f), (float)(h - 1));
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const bool x1ok = x0 + 1 < w, y1ok = y0 + 1 < h;
        const int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        float gx = 0.f, gy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* sp = frames + ((long long)fsup * 3 + c) * hw;
            const float v00 = sp[y0 * w + x0];
            const float v01 = x1ok ? sp[y0 * w + x1] : 0.f;
            const float v10 = y1ok ? sp[y1 * w + x0] : 0.f;
            const float v11 = (x1ok && y1ok) ? sp[y1 * w + x1] : 0.f;
            gx += d[3 + c] * (wy0 * (v01 - v00) + wy1 * (v11 - v10));
            gy += d[3 + c] * (wx0 * (v10 - v00) + wx1 * (v11 - v01));
        }
        if (dframes) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(dframes + ((long long)fref * 3 + c) * hw + (long long)y * w + x, d[c]);
                float* sp = dframes + ((long long)fsup * 3 + c) * hw;
                const float gv = d[3 + c];
                atomicAdd(sp + y0 * w + x0, gv * wy0 * wx0);
                if (x1ok) atomicAdd(sp + y0 * w + x1, gv * wy0 * wx1);
                if (y1ok) atomicAdd(sp + y1 * w + x0, gv * wy1 * wx0);
                if (x1ok && y1ok) atomicAdd(sp + y1 * w + x1, gv * wy1 * wx1);
            }
        }
        if (!dflow_prev) continue;                          // level 0: nothing below
        float g[2];
        g[0] = d[6] + (w > 1 ? gx * mx : 0.f);
        g[1] = d[7] + (h > 1 ? gy * my : 0.f);
        if (dflow_l) {
            g[0] += dflow_l[((long long)p * 2 + 0) * hw + (long long)y * w + x];
            g[1] += dflow_l[((long long)p * 2 + 1) * hw + (long long)y * w + x];
        }
        int yy0, yy1, xx0, xx1; float ly, lx;
        src_index(y, sy, hp, true, yy0, yy1, ly);
        src_index(x, sx, wp, true, xx0, xx1, lx);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* dp = dflow_prev + ((long long)p * 2 + c) * hp * wp;
            const float v = 2.0f * g[c];
            atomicAdd(dp + yy0 * wp + xx0, v * (1.f - ly) * (1.f - lx));
            atomicAdd(dp + yy0 * wp + xx1, v * (1.f - ly) * lx);
            atomicAdd(dp + yy1 * wp + xx0, v * ly * (1.f - lx));
            atomicAdd(dp + yy1 * wp + xx1, v * ly * lx);
        }
    }
}

// Backward of flow_out_kernel: din (P,2,hu,wu) += resize^T(dout (P,2,h,w)) * scale   (din zeroed by the caller)
__global__ void flow_out_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int P, int hu, int wu, int h, int w) {
    const long long total = (long long)P * 2 * h * w;
    const float sy = (float)hu / (float)h, sx = (float)wu / (float)w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int c = (int)((idx / ((long long)w * h)) % 2);
        const long long p = idx / ((long long)w * h * 2);
        int y0, y1, x0, x1; float ly, lx;
        src_index(y, sy, hu, false, y0, y1, ly);
        src_index(x, sx, wu, false, x0, x1, lx);
        float* ip = din + (p * 2 + c) * (long long)hu * wu;
        const float v = dout[idx] * (c == 0 ? (float)w / (float)wu : (float)h / (float)hu);
        atomicAdd(ip + y0 * wu + x0, v * (1.f - ly) * (1.f - lx));
        atomicAdd(ip + y0 * wu + x1, v * (1.f - ly) * lx);
        atomicAdd(ip + y1 * wu + x0, v * ly * (1.f - lx));
        atomicAdd(ip + y1 * wu + x1, v * ly * lx);
    }
}

// pyramid adjoints (spynet.py:44-45, 72-80): dfine += 0.25 * dcoarse[y/2][x/2]   (avg_pool2d backward)
__global__ void avgpool2_bwd_add_kernel(const float* __restrict__ dcoarse, float* __restrict__ dfine, long long planes, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const long long total = planes * h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long long pl = idx / ((long long)w * h);
        if ((y >> 1) < ho && (x >> 1) < wo) dfine[idx] += 0.25f * dcoarse[pl * ho * wo + (long long)(y >> 1) * wo + (x >> 1)];
    }
}

// dframes (F,3,h,w) += resize^T(dnorm (F,3,hu,wu)) / std   (backward of resize_norm_kernel; atomics)
__global__ void resize_norm_bwd_kernel(const float* __restrict__ dnorm, float* __restrict__ dframes, const float* __restrict__ std,
                                       int F, int h, int w, int hu, int wu) {
    const long long total = (long long)F * 3 * hu * wu;
    const float sy = (float)h / (float)hu, sx = (float)w / (float)wu;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % wu);
        const int y = (int)((idx / wu) % hu);
        const int c = (int)((idx / ((long long)wu * hu)) % 3);
        const long long f = idx / ((long long)wu * hu * 3);
        int y0, y1, x0, x1; float ly, lx;
        src_index(y, sy, h, false, y0, y1, ly);
        src_index(x, sx, w, false, x0, x1, lx);
        float* p = dframes + (f * 3 + c) * (long long)h * w;
        const float v = dnorm[idx] / std[c];
        atomicAdd(p + y0 * w + x0, v * (1.f - ly) * (1.f - lx));
        atomicAdd(p + y0 * w + x1, v * (1.f - ly) * lx);
        atomicAdd(p + y1 * w + x0, v * ly * (1.f - lx));
        atomicAdd(p + y1 * w + x1, v * ly * lx);
    }
}

// dlr (F,3,h,w) = bilinear_xS^T(dsr (F,3,S h,S w)), S = 4 or 2 (the `+ upscale(lr_i)` skip, basicvsr.py:22,82), as a gather: LR
// pixel y receives from the HR rows whose two source rows (align_corners=False, clamped at 0) include y: rows S y - S .. S y + 2 S - 1
template <int S>
__global__ void bilinear_bwd_kernel(const float* __restrict__ dsr, float* __restrict__ dlr, long long planes, int h, int w) {
    const long long total = planes * h * w;
    const int H = S * h, W = S * w;
    constexpr int K = 3 * S;
    constexpr float INV = 1.f / (float)S;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long long pl = idx / ((long long)w * h);
        const float* sp = dsr + pl * H * W;
        float wy[K], wx[K];
        int Y0 = S * y - S, X0 = S * x - S;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int Y = Y0 + k, X = X0 + k;
            wy[k] = 0.f; wx[k] = 0.f;
            if (Y >= 0 && Y < H) {
                float s = (Y + 0.5f) * INV - 0.5f; s = s < 0.f ? 0.f : s;
                const int i0 = (int)s; const int i1 = i0 + (i0 < h - 1 ? 1 : 0); const float l1 = s - (float)i0;
                wy[k] = (i0 == y ? 1.f - l1 : 0.f) + (i1 == y ? l1 : 0.f);
            }
            if (X >= 0 && X < W) {
                float s = (X + 0.5f) * INV - 0.5f; s = s < 0.f ? 0.f : s;
                const int i0 = (int)s; const int i1 = i0 + (i0 < w - 1 ? 1 : 0); const float l1 = s - (float)i0;
                wx[k] = (i0 == x ? 1.f - l1 : 0.f) + (i1 == x ? l1 : 0.f);
            }
        }
        float acc = 0.f;
        for (int ky = 0; ky < K; ++ky) {
            if (wy[ky] == 0.f) continue;
            const float* row = sp + (long long)(Y0 + ky) * W;
            float r = 0.f;
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
                if (wx[kx] != 0.f) r += wx[kx] * row[X0 + kx];
            acc += wy[ky] * r;
        }
        dlr[idx] = acc;
    }
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

// OIHW fp32 -> packed T [tap][R][Cc] consumed by conv_mfma (A operand rows = R, K = Cc).
//   mode 0 (forward):  dst[tap][r][c] = w[r*o_mul + o_add][i_off + c][tap]
//   mode 1 (data grad): dst[tap][r][c] = w[c*o_mul + o_add][i_off + r][KK-1-tap]   (flipped taps, roles swapped)
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ dst, int KK, int RP, int CPd,
                                    int r_real, int c_real, int I_total, int i_off, int o_mul, int o_add, int mode) {
    const int total = KK * RP * CPd;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % CPd, r = (idx / CPd) % RP, tap = idx / (CPd * RP);
    float v = 0.f;
    if (r < r_real && c < c_real) {
        if (mode == 0) v = w[((long long)(r * o_mul + o_add) * I_total + i_off + c) * KK + tap];
        else v = w[((long long)(c * o_mul + o_add) * I_total + i_off + r) * KK + (KK - 1 - tap)];
    }
    dst[idx] = (T)v;
}

struct PackArgs { int n; VsrPackDesc d[VSR_PACK_BATCH]; };
static_assert(sizeof(PackArgs) <= 4096, "kernel arguments");

// Many pack_weights_kernel launches in one: block -> tensor by binary search over the descriptors' first-block table.
__global__ void pack_multi_kernel(const PackArgs a) {
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.d[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const VsrPackDesc& d = a.d[lo];
    const int idx = ((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
    if (idx >= d.total) return;
    const int KK = d.KK, RP = d.RP, CPd = d.CPd;
    const int c = idx % CPd, r = (idx / CPd) % RP, tap = idx / (CPd * RP);
    float v = 0.f;
    if (r < d.r_real && c < d.c_real) {
        if (d.mode == 0) v = d.w[((long long)(r * d.o_mul + d.o_add) * d.I_total + d.i_off + c) * KK + tap];
        else v = d.w[((long long)(c * d.o_mul + d.o_add) * d.I_total + d.i_off + r) * KK + (KK - 1 - tap)];
    }
    if (d.dtype == VSR_BF16) reinterpret_cast<bf16_t*>(d.dst)[idx] = (bf16_t)v;
    else reinterpret_cast<float*>(d.dst)[idx] = v;
}

// The loss VALUE is summed in a fixed order (per-thread strided sums, wave shuffle tree, one partial per workgroup, then ONE
// workgroup adds the partials in index order): bit-identical from run to run (round 2 added the workgroup partials with atomicAdd,
// whose arrival order is not).
constexpr int CHARB_BLOCKS = 1024;
__global__ void charbonnier_grad_kernel(const float* __restrict__ sr, const float* __restrict__ hr, float* __restrict__ dsr,
                                        float* __restrict__ partial, long long n, float eps, float scale, int vec) {
    __shared__ float red[4];
    float local = 0.f;
    const long long n4 = vec ? n >> 2 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(sr);
    const float4* h4 = reinterpret_cast<const float4*>(hr);
    float4* d4 = reinterpret_cast<float4*>(dsr);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = s4[i], b = h4[i];
        float4 o;
        float d, r;
        d = a.x - b.x; r = sqrtf(d * d + eps); local += r; o.x = scale * d / r;
        d = a.y - b.y; r = sqrtf(d * d + eps); local += r; o.y = scale * d / r;
        d = a.z - b.z; r = sqrtf(d * d + eps); local += r; o.z = scale * d / r;
        d = a.w - b.w; r = sqrtf(d * d + eps); local += r; o.w = scale * d / r;
        d4[i] = o;
    }
    if (!vec) {                                                   // a pointer that is not 16-byte aligned (a view at an odd storage offset): scalar
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const float d = sr[i] - hr[i];
            const float r = sqrtf(d * d + eps);
            local += r;
            dsr[i] = scale * d / r;
        }
    } else if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {   // tail of a length that is not a multiple of 4
        const long long i = (n4 << 2) + threadIdx.x;
        const float d = sr[i] - hr[i];
        const float r = sqrtf(d * d + eps);
        local += r;
        dsr[i] = scale * d / r;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}
__global__ void charbonnier_sum_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0];
}

inline int grid_for(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                              \
    if ((dtype) == VSR_BF16) { typedef bf16_t T; CALL; }     \
    else if ((dtype) == VSR_F32) { typedef float T; CALL; }  \
    else return VSR_ERR_BADARG;

int vsr_launch_warp_fwd(int dtype, const void* in, const float* flow, void* out, int N, int H, int W, int C,
                        long long flow_nstride, hipStream_t st, int border) {
    if (C % 8) return VSR_ERR_BADARG;
    const long long total = (long long)N * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(warp_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)in, flow, (T*)out, N, H, W, C, flow_nstride, border));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_warp_bwd(int dtype, const void* dout, const float* flow, float* dacc, int N, int H, int W, int C,
                        long long flow_nstride, hipStream_t st, int border) {
    const long long total = (long long)N * H * W * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(warp_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)dout, flow, dacc, N, H, W, C, flow_nstride, border));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// out = T(dtop + warp^T(dout)) for the 64-channel propagation warps (zeros padding): gather kernel + far-source scatter.
// S: fp32 [N][H][W][64], ALL ZERO on entry and on exit; far_count: one int, zero on entry (the number of far sources on exit).
int vsr_launch_warp_bwd_gather(int dtype, const void* dout, const float* flow, const void* dtop, long long* S, int* far_count, void* out,
                               int N, int H, int W, long long flow_nstride, hipStream_t st) {
    if (!dout || !flow || !S || !far_count || !out) return VSR_ERR_BADARG;
    const long long npix = (long long)N * H * W;
    const int tiles = N * cdiv(H, GTH) * cdiv(W, GTW);
    long long gb = (npix + 255) / 256;                      // 4 waves per block x 64 pixels per wave
    if (gb > 256 * 16) gb = 256 * 16;
    if (gb < 1) gb = 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL(warp_bwd_far_kernel<T>, dim3((int)gb), dim3(256), 0, st, (const T*)dout, flow, S, far_count, N, H, W, flow_nstride));
    DISPATCH_T(dtype, hipLaunchKernelGGL(warp_bwd_gather_kernel<T>, dim3(tiles), dim3(256), 0, st, (const T*)dout, flow, (const T*)dtop, S,
                                         (const int*)far_count, (T*)out, N, H, W, flow_nstride));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_add_cast(int dtype, const void* a, const float* s, void* out, int N, int H, int W, int C, hipStream_t st) {
    const long long total = (long long)N * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(add_cast_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)a, s, (T*)out, N, H, W, C));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_planar_to_pm(int dtype, const float* in, void* out, int N, int Cin, int H, int W, int C, hipStream_t st) {
    const long long total = (long long)N * H * W * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(planar_to_pm_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, in, (T*)out, N, Cin, H, W, C));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_pm_to_planar(int dtype, const void* in, float* out, int N, int Cout, int H, int W, int C, hipStream_t st) {
    const long long total = (long long)N * Cout * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pm_to_planar_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)in, out, N, Cout, H, W, C));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_resize_norm(const float* in, float* out, const float* mean, const float* std, int F, int h, int w, int hu, int wu, hipStream_t st) {
    hipLaunchKernelGGL(resize_norm_kernel, dim3(grid_for((long long)F * 3 * hu * wu)), dim3(256), 0, st, in, out, mean, std, F, h, w, hu, wu);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_avgpool2(const float* in, float* out, long long planes, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(planes * (h / 2) * (w / 2))), dim3(256), 0, st, in, out, planes, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_spynet_prepare(int dtype, const float* frames, const float* flow_prev, float* flow_up, void* x16,
                              int n, int t, int P, int pair_mode, int h, int w, int level0, hipStream_t st) {
    const long long total = (long long)P * h * w;
    DISPATCH_T(dtype, hipLaunchKernelGGL(spynet_prepare_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, frames, flow_prev, flow_up, (T*)x16, n, t, P, pair_mode, h, w, level0));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_flow_out(const float* in, float* out, int P, int hu, int wu, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(flow_out_kernel, dim3(grid_for((long long)P * 2 * h * w)), dim3(256), 0, st, in, out, P, hu, wu, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_warp_bwd_flow(int dtype, const void* in, const void* dout, const float* flow, float* dflow, int N, int H, int W,
                             int C, long long flow_nstride, hipStream_t st, int border) {
    if (C % 8) return VSR_ERR_BADARG;
    DISPATCH_T(dtype, hipLaunchKernelGGL(warp_bwd_flow_kernel<T>, dim3(grid_for((long long)N * H * W)), dim3(256), 0, st, (const T*)in, (const T*)dout, flow, dflow, N, H, W, C, flow_nstride, border));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_spynet_dres(int dtype, const float* dflow, const float* res, void* out, int P, int h, int w, hipStream_t st) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(spynet_dres_kernel<T>, dim3(grid_for((long long)P * h * w)), dim3(256), 0, st, dflow, res, (T*)out, P, h, w));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_spynet_prepare_bwd(int dtype, const void* dx16, const float* dflow_l, const float* frames, const float* flow_up,
                                  float* dflow_prev, float* dframes, int n, int t, int P, int pair_mode, int h, int w, hipStream_t st) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(spynet_prepare_bwd_kernel<T>, dim3(grid_for((long long)P * h * w)), dim3(256), 0, st, (const T*)dx16, dflow_l, frames, flow_up, dflow_prev, dframes, n, t, P, pair_mode, h, w));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_flow_out_bwd(const float* dout, float* din, int P, int hu, int wu, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(flow_out_bwd_kernel, dim3(grid_for((long long)P * 2 * h * w)), dim3(256), 0, st, dout, din, P, hu, wu, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_avgpool2_bwd_add(const float* dcoarse, float* dfine, long long planes, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(avgpool2_bwd_add_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0, st, dcoarse, dfine, planes, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_resize_norm_bwd(const float* dnorm, float* dframes, const float* std, int F, int h, int w, int hu, int wu, hipStream_t st) {
    hipLaunchKernelGGL(resize_norm_bwd_kernel, dim3(grid_for((long long)F * 3 * hu * wu)), dim3(256), 0, st, dnorm, dframes, std, F, h, w, hu, wu);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_bilinear4_bwd(const float* dsr, float* dlr, long long planes, int h, int w, hipStream_t st, int scale) {
    if (scale == 2) hipLaunchKernelGGL(bilinear_bwd_kernel<2>, dim3(grid_for(planes * h * w)), dim3(256), 0, st, dsr, dlr, planes, h, w);
    else hipLaunchKernelGGL(bilinear_bwd_kernel<4>, dim3(grid_for(planes * h * w)), dim3(256), 0, st, dsr, dlr, planes, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_add_f32(const float* a, const float* b, float* out, long long n, hipStream_t st) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, b, out, n);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_pack_weights(int dtype, const float* w, void* dst, int KK, int RP, int CPd, int r_real, int c_real,
                            int I_total, int i_off, int o_mul, int o_add, int mode, hipStream_t st) {
    const int total = KK * RP * CPd;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, st, w, (T*)dst, KK, RP, CPd, r_real, c_real, I_total, i_off, o_mul, o_add, mode));
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_launch_pack_multi(const VsrPackDesc* descs, int n, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += VSR_PACK_BATCH) {
        PackArgs a;
        a.n = n - i0 < VSR_PACK_BATCH ? n - i0 : VSR_PACK_BATCH;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            a.d[i] = descs[i0 + i];
            if (a.d[i].total < 1 || !a.d[i].w || !a.d[i].dst || (a.d[i].dtype != VSR_BF16 && a.d[i].dtype != VSR_F32)) return VSR_ERR_BADARG;
            a.d[i].blk0 = blocks;
            blocks += cdiv(a.d[i].total, 256);
        }
        hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks), dim3(256), 0, st, a);
        HIP_CHECK_RET(hipGetLastError());
    }
    return VSR_OK;
}

int vsr_charbonnier_scratch_floats_impl() { return CHARB_BLOCKS; }
int vsr_launch_charbonnier_grad(const float* sr, const float* hr, float* dsr, float* loss_acc, float* scratch, long long n, float eps,
                                hipStream_t st) {
    if (!sr || !hr || !dsr || !loss_acc || !scratch || n < 1) return VSR_ERR_BADARG;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(sr) | reinterpret_cast<uintptr_t>(hr) | reinterpret_cast<uintptr_t>(dsr);
    if (bits & 3) return VSR_ERR_BADARG;
    const int vec = (bits & 15) == 0;                      // float4 access needs 16-byte alignment; otherwise the scalar loop (same values, another fixed order)
    const long long want = ((n >> 2) + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > CHARB_BLOCKS ? CHARB_BLOCKS : want));
    hipLaunchKernelGGL(charbonnier_grad_kernel, dim3(blocks), dim3(256), 0, st, sr, hr, dsr, scratch, n, eps, 1.0f / (float)n, vec);
    hipLaunchKernelGGL(charbonnier_sum_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, loss_acc);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}
