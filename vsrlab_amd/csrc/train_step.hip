// Training-step glue around the BasicVSR path (reference core/utils.py:270-280 `update_weights`:
// clip_grad_norm_(model.parameters(), grad_clip) + Adam.step(), and core/utils.py:235-240 `compute_loss`'s second
// term loss_fn(lq, resize(hr, (h, w)))), as streaming kernels over FLAT fp32 arenas.
//
// The HIP backward already delivers every gradient of the model into one contiguous arena
// (vsrlab_amd/functional.py), so "254 tensors" is one buffer of 4.85 M floats: the gradient norm is ONE
// reduction launch and clip + Adam is ONE elementwise launch (torch: 254-tensor foreach lists, ~10 launches
// and a host-side norm stack).  Everything here is HBM-bound: 4 reads + 3 writes of 19.4 MB ~ 25 us.
#include "kernels.h"
#include "../../include/vsrlab_hip.h"

namespace {

constexpr int OPT_BLOCKS = 1024, OPT_THREADS = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {       // red: 4 floats of LDS; all threads get the sum
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b] = sum of g[i]^2 over block b's grid-stride slice (fixed order: bitwise reproducible)
__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long long)OPT_BLOCKS * OPT_THREADS) {
        const float4 v = g4[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; long long n;
    float lr_over_bc1, beta1, beta2, eps, sqrt_bc2, weight_decay, grad_scale, max_norm;
    const float* partial; float* norm_out;
};

// torch.optim.Adam (single-tensor formulas, torch/optim/adam.py: lerp, mul+addcmul, sqrt/bias_correction2_sqrt + eps,
// addcdiv) with torch.nn.utils.clip_grad_norm_'s coefficient applied to the gradient on the fly:
//   total = ||grad_scale * g||_2 ; coef = min(1, max_norm / (total + 1e-6)) ; g' = grad_scale * coef * g (+ wd * p)
// A non-finite total norm skips the update (what GradScaler.step does on inf/nan, reference train.py:74).
__global__ __launch_bounds__(OPT_THREADS) void adam_clip_kernel(const AdamArgs a) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < OPT_BLOCKS; i += OPT_THREADS) s += a.partial[i];
    s = block_sum(s, red);
    const float total = sqrtf(s) * fabsf(a.grad_scale);
    float coef = a.grad_scale;
    if (a.max_norm > 0.f) {
        const float c = a.max_norm / (total + 1e-6f);
        coef *= c < 1.f ? c : 1.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) a.norm_out[0] = total;
    if (!(total <= 3.0e38f)) return;                      // inf / nan: skip the step
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2;
    const long long n4 = a.n >> 2;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    auto upd = [&](float& p, float g, float& m, float& v) {
        g *= coef;
        if (a.weight_decay != 0.f) g += a.weight_decay * p;
        m = m + w1 * (g - m);
        v = v * a.beta2 + (w2 * g) * g;                 // addcmul_: value * tensor1 * tensor2, left to right
        const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
        p = p - a.lr_over_bc1 * (m / denom);
    };
    for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * OPT_THREADS) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        upd(a.p[i], a.g[i], a.m[i], a.v[i]);
    }
}

// F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False, antialias=False) on planar fp32 images --
// what kornia.geometry.transform.resize(hr, (h, w)) computes in compute_loss (core/utils.py:239; core/losses.py:4).
__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int H, int W, int h, int w) {
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const long long total = planes * h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long long pl = idx / ((long long)w * h);
        float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* b = in + pl * H * W;
        const float top = b[(long long)y0 * W + x0] * (1.f - lx) + b[(long long)y0 * W + x1] * lx;
        const float bot = b[(long long)y1 * W + x0] * (1.f - lx) + b[(long long)y1 * W + x1] * lx;
        out[idx] = top * (1.f - ly) + bot * ly;
    }
}

}  // namespace

extern "C" {

size_t vsr_optim_scratch_floats(void) { return OPT_BLOCKS; }

int vsr_grad_norm(const float* grads, long long numel, float grad_scale, float* scratch, float* norm_out, void* stream) {
    if (!grads || !scratch || !norm_out || numel < 1) return VSR_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(grads) & 15) != 0) return VSR_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, st, grads, numel, scratch);
    AdamArgs a = {};
    a.n = 0; a.grad_scale = grad_scale; a.partial = scratch; a.norm_out = norm_out;
    hipLaunchKernelGGL(adam_clip_kernel, dim3(1), dim3(OPT_THREADS), 0, st, a);     // n = 0: only the norm is written
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, float grad_scale, float max_norm, float* scratch,
                       float* norm_out, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || numel < 1 || step < 1) return VSR_ERR_BADARG;
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(lr >= 0.f)) return VSR_ERR_BADARG;
    if (((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
          reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) != 0)
        return VSR_ERR_BADARG;                                      // float4 accesses
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, st, grads, numel, scratch);
    // bias corrections in double on the host, as torch/optim/adam.py computes them from Python floats
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a = {};
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = numel;
    a.lr_over_bc1 = (float)((double)lr / bc1); a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.sqrt_bc2 = (float)sqrt(bc2); a.weight_decay = weight_decay; a.grad_scale = grad_scale; a.max_norm = max_norm;
    a.partial = scratch; a.norm_out = norm_out;
    long long blocks = ((numel >> 2) + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((int)blocks), dim3(OPT_THREADS), 0, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_resize_bilinear(const float* in, float* out, long long planes, int H, int W, int h, int w, void* stream) {
    if (!in || !out || planes < 1 || H < 1 || W < 1 || h < 1 || w < 1) return VSR_ERR_BADARG;
    const long long total = planes * h * w;
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, in, out, planes, H, W, h, w);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

}  // extern "C"
