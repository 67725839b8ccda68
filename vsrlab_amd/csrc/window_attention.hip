// VRT window attention on the matrix cores (BASELINE config 5).
// Reference: vsr/models/VRT/modules/window_attention.py:100-188 (WindowAttention.forward / .attention):
//     attn = (q * scale) @ k^T  [+ relative_position_bias[h]]  [+ mask[window]] ; softmax ; x = attn @ v
// for self attention over the N tokens of a (2|6, 8, 8) window (N = 128 | 384) and, for the mutual attention of a
// 2-frame window, between the two halves of the tokens (queries of one frame, keys / values of the other, N/2 = 64 | 192
// tokens).  head_dim = dim / heads = 20 or 30 in the VRT configurations: it is zero-padded to ONE 32-deep MFMA step.
//
// One kernel computes QK^T, the softmax and AV without writing the N x N scores: a wave owns 16 queries and ALL keys
// (N <= 384: the 16 x N scores of a wave are 96 accumulator registers per lane), flash-attention style but without the
// online rescaling.  The product is formed TRANSPOSED (S^T = K Q^T) -- "swapped QK^T", cdna_hip_programming T12 -- so
// that a query's scores sit in ONE lane column of the 16x16 accumulators: the row reductions are register loops plus two
// cross-lane steps, and P^T is directly the B operand of O^T = V^T P^T (the k order of that MFMA is permuted to the
// accumulator's row order {4q+j, 16+4q+j}; the A operand V^T is read from LDS in the same order), no LDS round trip.
// The backward is the usual two passes: query-stationary (delta, dQ, bias gradient) and key-stationary (dK, dV).
// T = bf16 (v_mfma_f32_16x16x32_bf16) or fp32 (8 x v_mfma_f32_16x16x4_f32 per step: the same fragment layout, exact fp32).
#include <cmath>
#include <cstdlib>
#include "kernels.h"
#include "../../include/vsrlab_hip.h"

namespace {

constexpr int HP = 32;                      // padded head dim
constexpr int WQ = 16;                      // queries (pass 1) / keys (pass 2) per wave
constexpr int MAXW = 8;                      // waves per workgroup: one (window, head) per workgroup, its query / key blocks shared by the waves
constexpr int MAXT = 24;                    // score tiles of 16 per wave: N <= 384

struct af8 { float v[8]; };
template <typename T> struct AT;
template <> struct AT<bf16_t> { typedef bf16x8_t frag_t; };
template <> struct AT<float> { typedef af8 frag_t; };

__device__ __forceinline__ void amma(f32x4_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
}
__device__ __forceinline__ void amma(f32x4_t& d, const af8& a, const af8& b) {
    // slot j of lane group q is k = (q, j): any k order is valid as long as A and B agree
#pragma unroll
    for (int j = 0; j < 8; ++j) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], d, 0, 0, 0);
}
__device__ __forceinline__ void fset(bf16x8_t& f, int j, float v) { f[j] = (bf16_t)v; }
__device__ __forceinline__ void fset(af8& f, int j, float v) { f.v[j] = v; }
__device__ __forceinline__ float tof(bf16_t v) { return (float)v; }
__device__ __forceinline__ float tof(float v) { return v; }

struct AttnArgs {
    const void* qkv;            // [B][N][3][nH][hd] of T
    void* out;                  // fwd: [B][N][Cout] of T          bwd: unused
    const void* dout;           // bwd: [B][N][Cout] of T
    void* dqkv;                 // bwd: [B][N][3][nH][hd] of T (each element written once)
    float* lse;                 // [B][nH][Nq] logsumexp of the scores (fwd writes, bwd reads)
    float* delta;               // [B][nH][Nq] rowsum(dO * O) (bwd pass 1 writes, pass 2 reads)
    const float* bias;          // dense [nH][Nq][Nk] or null (relative position bias, self attention only)
    float* dbias;               // dense gradient accumulator (atomics) or null
    const float* mask;          // [nW][Nm][Nm] (0 / -100) or null; the top-left Nq x Nk block is used (window_attention.py:156)
    const unsigned* mask_bits;  // the same mask bit-packed, [nW][Nm][Nm/32] (bit = entry != 0), with mask_value: persistent kernels
    float mask_value;
    int nW, Nm;
    int B, N, nH, hd;
    int q0, k0, o0;             // first query token, first key/value token, first output token row
    int Nq, Nk;
    int Cout, c_off;            // output row length and first channel of this call's heads
    float scale;
};

// 8 consecutive head-dim values [8q, 8q+8) of a token's q / k / v vector, zero beyond hd, times mul
template <typename T> struct V4;
template <> struct __attribute__((aligned(8))) V4<bf16_t> { bf16_t v[4]; };
template <> struct __attribute__((aligned(16))) V4<float> { float v[4]; };

template <typename T>
__device__ __forceinline__ typename AT<T>::frag_t load_hd8(const T* p, int q, int hd, float mul) {
    typename AT<T>::frag_t f;
    if ((hd & 3) == 0) {                      // groups of 4 are all-valid or all-padding: two vector loads
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c0 = 8 * q + 4 * g;
            V4<T> v;
            if (c0 < hd) v = *reinterpret_cast<const V4<T>*>(p + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) fset(f, 4 * g + j, c0 < hd ? tof(v.v[j]) * mul : 0.f);
        }
        return f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * q + j;
        fset(f, j, c < hd ? tof(p[c]) * mul : 0.f);
    }
    return f;
}

// 4 consecutive head-dim values [c0, c0+4) of an output row (accumulator registers of one m-block), x mul
template <typename T>
__device__ __forceinline__ void store_hd4(T* p, int c0, int hd, const f32x4_t& v, float mul) {
    if ((hd & 3) == 0) {
        if (c0 < hd) {
            V4<T> o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = (T)(v[j] * mul);
            *reinterpret_cast<V4<T>*>(p + c0) = o;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < hd) p[c0 + j] = (T)(v[j] * mul);
}

__device__ __forceinline__ float xsum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float xmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// ------------------------------------------------------------------------------------------------------------
// Shared by the forward and backward pass 1: S^T tiles of this wave's 16 queries against all keys, + bias + mask.
// lds_k: [Nk][HP] of T (row-major keys).  st[t][j] = score(key 16 t + 4 q + j, query n).
// ------------------------------------------------------------------------------------------------------------
template <typename T, int NT>
__device__ __forceinline__ void scores_T(const AttnArgs& a, const T* lds_k, const typename AT<T>::frag_t& qf, int b, int h, int qrow, int lane,
                                         f32x4_t (&st)[NT]) {
    constexpr int nt = NT;
    const int n = lane & 15, q = lane >> 4;
    typedef typename AT<T>::frag_t frag_t;
#pragma unroll
    for (int t = 0; t < nt; ++t) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(lds_k + (t * 16 + n) * HP + 8 * q);
        f32x4_t d = {0.f, 0.f, 0.f, 0.f};
        amma(d, kf, qf);
        st[t] = d;
    }
    if (a.bias) {
        const float* bp = a.bias + ((long long)h * a.Nq + qrow) * a.Nk + 4 * q;
#pragma unroll
        for (int t = 0; t < nt; ++t) {
            const float4 bv = *reinterpret_cast<const float4*>(bp + t * 16);
            st[t][0] += bv.x; st[t][1] += bv.y; st[t][2] += bv.z; st[t][3] += bv.w;
        }
    }
    if (a.mask) {
        const float* mp = a.mask + ((long long)(b % a.nW) * a.Nm + qrow) * a.Nm + 4 * q;
#pragma unroll
        for (int t = 0; t < nt; ++t) {
            const float4 mv = *reinterpret_cast<const float4*>(mp + t * 16);
            st[t][0] += mv.x; st[t][1] += mv.y; st[t][2] += mv.z; st[t][3] += mv.w;
        }
    }
}

// row stride (elements) of a transposed [HP][n] image: + 8 keeps the 8-byte fragment reads aligned and spreads the 32
// rows that consecutive staging threads write over 16 bank groups (a stride of n = 128 elements = 256 B put all on one)
__device__ __forceinline__ int tstride(int n) { return n + 8; }

// stage rows [t0, t0 + n) of component `which` (0 q, 1 k, 2 v) of head h, window b: row-major [n][HP] and/or transposed
// [HP][tstride(n)]
template <typename T>
__device__ __forceinline__ void stage_rows(const AttnArgs& a, int b, int h, int which, int t0, int n, T* rowmajor, T* transposed, float mul) {
    const int ts = tstride(n);
    const T* base = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + t0) * 3 + which) * a.nH * a.hd + (long long)h * a.hd;
    const long long rs = (long long)3 * a.nH * a.hd;
    // 4 independent global loads in flight per thread before the first LDS store (a load -> store loop is one HBM/L2
    // round trip per iteration: the staging, not the MFMAs, was the kernel's critical path)
    const int total = n * HP, step = blockDim.x;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * step) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + u * step;
            const int r = idx / HP, c = idx - r * HP;
            v[u] = (idx < total && c < a.hd) ? tof(base[r * rs + c]) * mul : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + u * step;
            if (idx < total) {
                const int r = idx / HP, c = idx - r * HP;
                if (rowmajor) rowmajor[r * HP + c] = (T)v[u];
                if (transposed) transposed[c * ts + r] = (T)v[u];
            }
        }
    }
}

// =========================================== forward ===========================================
template <typename T, int NT>
__global__ __launch_bounds__(MAXW * 64) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    T* lds_k = reinterpret_cast<T*>(smem);                  // [Nk][HP]
    T* lds_vt = lds_k + a.Nk * HP;                          // [HP][tstride(Nk)]
    const int b = blockIdx.z, h = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int nwaves = blockDim.x >> 6, vts = tstride(a.Nk);
    stage_rows<T>(a, b, h, 1, a.k0, a.Nk, lds_k, nullptr, 1.f);
    stage_rows<T>(a, b, h, 2, a.k0, a.Nk, nullptr, lds_vt, 1.f);
    __syncthreads();
    // K and V of this (window, head) are staged ONCE; the workgroup's waves share the blocks of 16 queries
    for (int qb = wave; qb * WQ < a.Nq; qb += nwaves) {
    const int qrow = qb * WQ + n;
    constexpr int nt = NT;                                  // = Nk / 16: the score tiles live in registers
    const T* qp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
    const frag_t qf = load_hd8<T>(qp, q, a.hd, a.scale);
    f32x4_t st[NT];
    scores_T<T, NT>(a, lds_k, qf, b, h, qrow, lane, st);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < nt; ++t) m = fmaxf(fmaxf(m, fmaxf(st[t][0], st[t][1])), fmaxf(st[t][2], st[t][3]));
    m = xmax(m);
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { st[t][j] = __expf(st[t][j] - m); l += st[t][j]; }
    }
    l = xsum(l);
    if (q == 0 && a.lse) a.lse[((long long)b * a.nH + h) * a.Nq + qrow] = m + __logf(l);
    // O^T[hd][query] = sum_keys V^T[hd][key] P^T[key][query]; k order of a step = {32 s + 4 q + j} U {32 s + 16 + 4 q + j}
    f32x4_t o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < nt / 2; ++s) {
        frag_t pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { fset(pf, j, st[2 * s][j]); fset(pf, 4 + j, st[2 * s + 1][j]); }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const T* vp = lds_vt + (mb * 16 + n) * vts + 32 * s + 4 * q;
            frag_t vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fset(vf, j, tof(vp[j])); fset(vf, 4 + j, tof(vp[16 + j])); }
            amma(o[mb], vf, pf);
        }
    }
    const float inv = 1.f / l;
    T* op = reinterpret_cast<T*>(a.out) + ((long long)b * a.N + a.o0 + qrow) * a.Cout + a.c_off + h * a.hd;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = mb * 16 + 4 * q + j;
            if (c < a.hd) op[c] = (T)(o[mb][j] * inv);
        }
    }
}

// ============================ backward pass 1: query-stationary (delta, dQ, dbias) ============================
template <typename T, int NT>
__global__ __launch_bounds__(MAXW * 64) void attn_bwd_q_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    T* lds_k = reinterpret_cast<T*>(smem);                  // [Nk][HP]
    T* lds_v = lds_k + a.Nk * HP;                           // [Nk][HP]
    T* lds_kt = lds_v + a.Nk * HP;                          // [HP][tstride(Nk)]
    const int b = blockIdx.z, h = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int nwaves = blockDim.x >> 6, kts = tstride(a.Nk);
    stage_rows<T>(a, b, h, 1, a.k0, a.Nk, lds_k, lds_kt, 1.f);
    stage_rows<T>(a, b, h, 2, a.k0, a.Nk, lds_v, nullptr, 1.f);
    __syncthreads();
    for (int qb = wave; qb * WQ < a.Nq; qb += nwaves) {
    const int qrow = qb * WQ + n;
    constexpr int nt = NT;
    const T* qp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
    const frag_t qf = load_hd8<T>(qp, q, a.hd, a.scale);
    const T* dop = reinterpret_cast<const T*>(a.dout) + ((long long)b * a.N + a.o0 + qrow) * a.Cout + a.c_off + h * a.hd;
    const frag_t dof = load_hd8<T>(dop, q, a.hd, 1.f);
    f32x4_t st[NT];
    scores_T<T, NT>(a, lds_k, qf, b, h, qrow, lane, st);
    const float lse = a.lse[((long long)b * a.nH + h) * a.Nq + qrow];
    // P^T, dP^T = V dO^T, delta = sum_keys P dP
    float delta = 0.f;
    f32x4_t dp[NT];
#pragma unroll
    for (int t = 0; t < nt; ++t) {
        const frag_t vf = *reinterpret_cast<const frag_t*>(lds_v + (t * 16 + n) * HP + 8 * q);
        f32x4_t d = {0.f, 0.f, 0.f, 0.f};
        amma(d, vf, dof);
        dp[t] = d;
#pragma unroll
        for (int j = 0; j < 4; ++j) { st[t][j] = __expf(st[t][j] - lse); delta += st[t][j] * d[j]; }
    }
    delta = xsum(delta);
    if (q == 0) a.delta[((long long)b * a.nH + h) * a.Nq + qrow] = delta;
    // dS^T = P^T (dP^T - delta)
#pragma unroll
    for (int t = 0; t < nt; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) st[t][j] = st[t][j] * (dp[t][j] - delta);
    if (a.dbias) {
        float* bp = a.dbias + ((long long)h * a.Nq + qrow) * a.Nk + 4 * q;
#pragma unroll
        for (int t = 0; t < nt; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(bp + t * 16 + j, st[t][j]);
    }
    // dQ^T[hd][query] = scale * sum_keys K^T[hd][key] dS^T[key][query]
    f32x4_t dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < nt / 2; ++s) {
        frag_t sf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { fset(sf, j, st[2 * s][j]); fset(sf, 4 + j, st[2 * s + 1][j]); }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const T* kp = lds_kt + (mb * 16 + n) * kts + 32 * s + 4 * q;
            frag_t kf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fset(kf, j, tof(kp[j])); fset(kf, 4 + j, tof(kp[16 + j])); }
            amma(dq[mb], kf, sf);
        }
    }
    T* gp = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = mb * 16 + 4 * q + j;
            if (c < a.hd) gp[c] = (T)(dq[mb][j] * a.scale);
        }
    }
}

// ============================ backward pass 2: key-stationary (dK, dV) ============================
constexpr int QC = 128;                      // queries staged in LDS at a time (4 x QC x HP elements: 64 KiB in fp32)
template <typename T>
__global__ __launch_bounds__(MAXW * 64) void attn_bwd_kv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    const int qc = a.Nq < QC ? a.Nq : QC;
    T* lds_q = reinterpret_cast<T*>(smem);                  // [qc][HP]   (q * scale)
    T* lds_do = lds_q + qc * HP;                            // [qc][HP]
    const int qts = tstride(qc);
    T* lds_qt = lds_do + qc * HP;                           // [HP][tstride(qc)]
    T* lds_dot = lds_qt + HP * qts;                         // [HP][tstride(qc)]
    float* lds_lse = reinterpret_cast<float*>(lds_dot + HP * qts);   // [qc]
    float* lds_delta = lds_lse + qc;                        // [qc]
    const int b = blockIdx.z, h = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int kb = blockIdx.x * (blockDim.x >> 6) + wave;   // this wave's block of 16 keys
    const bool active = kb * WQ < a.Nk;                     // (every wave takes part in the staging and its barriers)
    const int krow = kb * WQ + n;
    frag_t kf, vf;                                          // B operands: K^T / V^T columns of this lane's key
    if (active) {
        const T* kp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 1) * a.nH * a.hd + (long long)h * a.hd;
        const T* vp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 2) * a.nH * a.hd + (long long)h * a.hd;
        kf = load_hd8<T>(kp, q, a.hd, 1.f);
        vf = load_hd8<T>(vp, q, a.hd, 1.f);
    }
    f32x4_t dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int c0 = 0; c0 < a.Nq; c0 += qc) {
        __syncthreads();                                    // the previous chunk is consumed
        stage_rows<T>(a, b, h, 0, a.q0 + c0, qc, lds_q, lds_qt, a.scale);
        {   // dO rows of this head: [qc][hd] at row stride Cout
            const T* base = reinterpret_cast<const T*>(a.dout) + ((long long)b * a.N + a.o0 + c0) * a.Cout + a.c_off + h * a.hd;
            const int total = qc * HP, step = blockDim.x;
            for (int i0 = threadIdx.x; i0 < total; i0 += 4 * step) {
                T v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * step;
                    const int r = idx / HP, c = idx - r * HP;
                    v[u] = (idx < total && c < a.hd) ? base[(long long)r * a.Cout + c] : (T)0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * step;
                    if (idx < total) {
                        const int r = idx / HP, c = idx - r * HP;
                        lds_do[r * HP + c] = v[u];
                        lds_dot[c * qts + r] = v[u];
                    }
                }
            }
            for (int i = threadIdx.x; i < qc; i += blockDim.x) {
                lds_lse[i] = a.lse[((long long)b * a.nH + h) * a.Nq + c0 + i];
                lds_delta[i] = a.delta[((long long)b * a.nH + h) * a.Nq + c0 + i];
            }
        }
        __syncthreads();
        if (!active) continue;
        for (int s = 0; s < qc / 32; ++s) {
            frag_t pf, sf;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * s + half;                 // 16-query tile of the chunk: S[query 16 t + 4 q + j][key n]
                const frag_t qf = *reinterpret_cast<const frag_t*>(lds_q + (t * 16 + n) * HP + 8 * q);
                const frag_t dof = *reinterpret_cast<const frag_t*>(lds_do + (t * 16 + n) * HP + 8 * q);
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dpp = {0.f, 0.f, 0.f, 0.f};
                amma(sc, qf, kf);
                amma(dpp, dof, vf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ql = t * 16 + 4 * q + j, qi = c0 + ql;
                    float sv = sc[j];
                    if (a.bias) sv += a.bias[((long long)h * a.Nq + qi) * a.Nk + krow];
                    if (a.mask) sv += a.mask[((long long)(b % a.nW) * a.Nm + qi) * a.Nm + krow];
                    const float p = __expf(sv - lds_lse[ql]);
                    fset(pf, half * 4 + j, p);
                    fset(sf, half * 4 + j, p * (dpp[j] - lds_delta[ql]));
                }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const T* dop = lds_dot + (mb * 16 + n) * qts + 32 * s + 4 * q;
                const T* qtp = lds_qt + (mb * 16 + n) * qts + 32 * s + 4 * q;
                frag_t df, qtf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fset(df, j, tof(dop[j])); fset(df, 4 + j, tof(dop[16 + j]));
                    fset(qtf, j, tof(qtp[j])); fset(qtf, 4 + j, tof(qtp[16 + j]));
                }
                amma(dv[mb], df, pf);                       // dV^T[hd][key] += dO^T[hd][queries] P[queries][key]
                amma(dk[mb], qtf, sf);                      // dK^T[hd][key] += (scale Q)^T[hd][queries] dS[queries][key]
            }
        }
    }
    if (!active) return;
    T* gk = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 1) * a.nH * a.hd + (long long)h * a.hd;
    T* gv = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 2) * a.nH * a.hd + (long long)h * a.hd;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = mb * 16 + 4 * q + j;
            if (c < a.hd) { gk[c] = (T)dk[mb][j]; gv[c] = (T)dv[mb][j]; }
        }
}

// =====================================================================================================================
// Persistent, register-resident form for the window shapes of VRT's main stages (window (2,8,8): Nq = Nk = 128 for the self
// attention, 64 for the mutual attention).  What the per-(window, head) kernels above pay for at 7360 windows x 6 heads:
// 64 KiB of relative-position bias re-read from L2 by every workgroup, 16 K bias-gradient atomics per workgroup (723 M per
// backward), 2-byte staging loads whose round trip nothing hides, and the six heads of a window -- which share every
// 128-byte line of qkv -- landing on six different XCDs (FETCH_SIZE 2.6 x the tensor).  Here a workgroup is bound to ONE
// head and walks a strided list of windows:
//   * one wave = one block of 16 queries (keys in the key-stationary pass) for the whole launch, so its rows of the bias
//     live in registers, and so does its block of the bias gradient (flushed with atomics once, at the end);
//   * K / V (Q / dO) of the NEXT window are requested into registers before the current window is computed and written to
//     the other LDS buffer after it: the HBM round trip hides behind the MFMAs, one barrier per window;
//   * workgroup id = xcd + 8 (head + heads * part): the heads of a window run on the same XCD at about the same time, so
//     a line fetched for one head is an L2 hit for the other five.
// =====================================================================================================================
template <typename T> struct Pr;
template <> struct Pr<bf16_t> { typedef unsigned raw_t; };
template <> struct Pr<float> { typedef uint2 raw_t; };
__device__ __forceinline__ void unpack2(unsigned r, float& e0, float& e1) { e0 = __uint_as_float(r << 16); e1 = __uint_as_float(r & 0xffff0000u); }
__device__ __forceinline__ void unpack2(uint2 r, float& e0, float& e1) { e0 = __uint_as_float(r.x); e1 = __uint_as_float(r.y); }
__device__ __forceinline__ unsigned zero_raw(unsigned*) { return 0u; }
__device__ __forceinline__ uint2 zero_raw(uint2*) { return make_uint2(0u, 0u); }

// NROWS x HP image handled by NTH threads: element PAIRS in registers between the global load and the LDS store
template <typename T, int NROWS, int NTH>
struct Stager {
    static constexpr int PAIRS = NROWS * (HP / 2) / NTH;
    static_assert(NROWS * (HP / 2) % NTH == 0, "whole pairs per thread");
    typedef typename Pr<T>::raw_t raw_t;
    raw_t r[PAIRS];
    __device__ __forceinline__ void load(const T* base, long long rs, int hd, int tid) {
#pragma unroll
        for (int u = 0; u < PAIRS; ++u) {
            const int idx = tid + u * NTH, row = idx >> 4, c = (idx & 15) * 2;
            r[u] = c < hd ? *reinterpret_cast<const raw_t*>(base + row * rs + c) : zero_raw((raw_t*)nullptr);
        }
    }
    __device__ __forceinline__ void store(T* rowmajor, T* transposed, float mul, int tid) const {
        constexpr int ts = NROWS + 8;
#pragma unroll
        for (int u = 0; u < PAIRS; ++u) {
            const int idx = tid + u * NTH, row = idx >> 4, c = (idx & 15) * 2;
            float e0, e1;
            unpack2(r[u], e0, e1);
            const T t0 = (T)(e0 * mul), t1 = (T)(e1 * mul);
            if (rowmajor) { rowmajor[row * HP + c] = t0; rowmajor[row * HP + c + 1] = t1; }
            if (transposed) { transposed[c * ts + row] = t0; transposed[(c + 1) * ts + row] = t1; }
        }
    }
};

// mask of this lane's query row against keys 16 t + 4 q + j: bit-packed (NK / 32 words per row, one 16-byte load for
// NK = 128) when the caller packed it, else the dense fp32 rows
#define PMASK_APPLY                                                                                                    \
        if (a.mask_bits) {                                                                                             \
            const unsigned* wp = a.mask_bits + ((long long)(b % a.nW) * a.Nm + qrow) * (a.Nm >> 5);                    \
            unsigned wd[NT / 2];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NT / 2; ++i) wd[i] = wp[i];                                          \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
                    if ((wd[t >> 1] >> (((t & 1) * 16) + 4 * q + j)) & 1u) st[t][j] += a.mask_value;                   \
        } else if (a.mask) {                                                                                           \
            const float* mp = a.mask + ((long long)(b % a.nW) * a.Nm + qrow) * a.Nm + 4 * q;                           \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                           \
                const float4 mv = *reinterpret_cast<const float4*>(mp + t * 16);                                       \
                st[t][0] += mv.x; st[t][1] += mv.y; st[t][2] += mv.z; st[t][3] += mv.w;                                \
            }                                                                                                          \
        }

struct PWalk { int h, part, nparts; };
__device__ __forceinline__ PWalk pwalk(const AttnArgs& a) {
    const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
    PWalk w;
    w.h = r % a.nH;
    const int pp = r / a.nH, npp = (gridDim.x >> 3) / a.nH;
    w.part = pp * 8 + xcd;
    w.nparts = npp * 8;
    return w;
}

template <typename T, int NT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void attn_fwd_p_kernel(const AttnArgs a) {      // 2 workgroups per CU: <= 128 VGPRs
    constexpr int NK = NT * 16, NTH = NW * 64, TS = NK + 8;
    constexpr int BUF = NK * HP + HP * TS;                   // elements per buffer: K [NK][HP] + V^T [HP][TS]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    T* lds = reinterpret_cast<T*>(smem);
    const PWalk w = pwalk(a);
    const int h = w.h, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int qrow = wave * WQ + n;
    const long long rs = (long long)3 * a.nH * a.hd;
    auto kv_base = [&](int b, int which) {
        return reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0) * 3 + which) * a.nH * a.hd + (long long)h * a.hd;
    };
    f32x4_t bias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (a.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + ((long long)h * a.Nq + qrow) * a.Nk + t * 16 + 4 * q);
            bias_r[t][0] = bv.x; bias_r[t][1] = bv.y; bias_r[t][2] = bv.z; bias_r[t][3] = bv.w;
        } else {
            bias_r[t][0] = bias_r[t][1] = bias_r[t][2] = bias_r[t][3] = 0.f;
        }
    }
    Stager<T, NK, NTH> sk, sv;
    int b = w.part;
    if (b < a.B) {
        sk.load(kv_base(b, 1), rs, a.hd, tid); sv.load(kv_base(b, 2), rs, a.hd, tid);
        sk.store(lds, nullptr, 1.f, tid); sv.store(nullptr, lds + NK * HP, 1.f, tid);
    }
    __syncthreads();
    for (int cur = 0; b < a.B; b += w.nparts, cur ^= 1) {
        const int bn = b + w.nparts;
        const bool has_next = bn < a.B;
        if (has_next) { sk.load(kv_base(bn, 1), rs, a.hd, tid); sv.load(kv_base(bn, 2), rs, a.hd, tid); }
        const T* lds_k = lds + cur * BUF;
        const T* lds_vt = lds_k + NK * HP;
        const T* qp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
        const frag_t qf = load_hd8<T>(qp, q, a.hd, a.scale);
        f32x4_t st[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const frag_t kf = *reinterpret_cast<const frag_t*>(lds_k + (t * 16 + n) * HP + 8 * q);
            f32x4_t d = bias_r[t];
            amma(d, kf, qf);
            st[t] = d;
        }
        PMASK_APPLY
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) m = fmaxf(fmaxf(m, fmaxf(st[t][0], st[t][1])), fmaxf(st[t][2], st[t][3]));
        m = xmax(m);
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { st[t][j] = __expf(st[t][j] - m); l += st[t][j]; }
        }
        l = xsum(l);
        if (q == 0 && a.lse) a.lse[((long long)b * a.nH + h) * a.Nq + qrow] = m + __logf(l);
        f32x4_t o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
            frag_t pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fset(pf, j, st[2 * s][j]); fset(pf, 4 + j, st[2 * s + 1][j]); }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const T* vp = lds_vt + (mb * 16 + n) * TS + 32 * s + 4 * q;
                frag_t vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { fset(vf, j, tof(vp[j])); fset(vf, 4 + j, tof(vp[16 + j])); }
                amma(o[mb], vf, pf);
            }
        }
        const float inv = 1.f / l;
        T* op = reinterpret_cast<T*>(a.out) + ((long long)b * a.N + a.o0 + qrow) * a.Cout + a.c_off + h * a.hd;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) store_hd4<T>(op, mb * 16 + 4 * q, a.hd, o[mb], inv);
        if (has_next) {
            T* nb = lds + (cur ^ 1) * BUF;
            sk.store(nb, nullptr, 1.f, tid); sv.store(nullptr, nb + NK * HP, 1.f, tid);
        }
        __syncthreads();
    }
}

template <typename T, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_q_p_kernel(const AttnArgs a) {
    constexpr int NK = NT * 16, NTH = NW * 64, TS = NK + 8;
    constexpr int BUF = 2 * NK * HP + HP * TS;               // K [NK][HP], V [NK][HP], K^T [HP][TS]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    T* lds = reinterpret_cast<T*>(smem);
    const PWalk w = pwalk(a);
    const int h = w.h, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int qrow = wave * WQ + n;
    const long long rs = (long long)3 * a.nH * a.hd;
    auto kv_base = [&](int b, int which) {
        return reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0) * 3 + which) * a.nH * a.hd + (long long)h * a.hd;
    };
    f32x4_t bias_r[NT], dbias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        dbias_r[t][0] = dbias_r[t][1] = dbias_r[t][2] = dbias_r[t][3] = 0.f;
        if (a.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + ((long long)h * a.Nq + qrow) * a.Nk + t * 16 + 4 * q);
            bias_r[t][0] = bv.x; bias_r[t][1] = bv.y; bias_r[t][2] = bv.z; bias_r[t][3] = bv.w;
        } else {
            bias_r[t][0] = bias_r[t][1] = bias_r[t][2] = bias_r[t][3] = 0.f;
        }
    }
    Stager<T, NK, NTH> sk, sv;
    int b = w.part;
    if (b < a.B) {
        sk.load(kv_base(b, 1), rs, a.hd, tid); sv.load(kv_base(b, 2), rs, a.hd, tid);
        sk.store(lds, lds + 2 * NK * HP, 1.f, tid); sv.store(lds + NK * HP, nullptr, 1.f, tid);
    }
    __syncthreads();
    for (int cur = 0; b < a.B; b += w.nparts, cur ^= 1) {
        const int bn = b + w.nparts;
        const bool has_next = bn < a.B;
        if (has_next) { sk.load(kv_base(bn, 1), rs, a.hd, tid); sv.load(kv_base(bn, 2), rs, a.hd, tid); }
        const T* lds_k = lds + cur * BUF;
        const T* lds_v = lds_k + NK * HP;
        const T* lds_kt = lds_v + NK * HP;
        const T* qp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
        const frag_t qf = load_hd8<T>(qp, q, a.hd, a.scale);
        const T* dop = reinterpret_cast<const T*>(a.dout) + ((long long)b * a.N + a.o0 + qrow) * a.Cout + a.c_off + h * a.hd;
        const frag_t dof = load_hd8<T>(dop, q, a.hd, 1.f);
        const float lse = a.lse[((long long)b * a.nH + h) * a.Nq + qrow];
        f32x4_t st[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const frag_t kf = *reinterpret_cast<const frag_t*>(lds_k + (t * 16 + n) * HP + 8 * q);
            f32x4_t d = bias_r[t];
            amma(d, kf, qf);
            st[t] = d;
        }
        PMASK_APPLY
        float delta = 0.f;
        f32x4_t dp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const frag_t vf = *reinterpret_cast<const frag_t*>(lds_v + (t * 16 + n) * HP + 8 * q);
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            amma(d, vf, dof);
            dp[t] = d;
#pragma unroll
            for (int j = 0; j < 4; ++j) { st[t][j] = __expf(st[t][j] - lse); delta += st[t][j] * d[j]; }
        }
        delta = xsum(delta);
        if (q == 0) a.delta[((long long)b * a.nH + h) * a.Nq + qrow] = delta;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { st[t][j] = st[t][j] * (dp[t][j] - delta); dbias_r[t][j] += st[t][j]; }
        f32x4_t dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
            frag_t sf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fset(sf, j, st[2 * s][j]); fset(sf, 4 + j, st[2 * s + 1][j]); }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const T* kp = lds_kt + (mb * 16 + n) * TS + 32 * s + 4 * q;
                frag_t kf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { fset(kf, j, tof(kp[j])); fset(kf, 4 + j, tof(kp[16 + j])); }
                amma(dq[mb], kf, sf);
            }
        }
        T* gp = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.q0 + qrow) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) store_hd4<T>(gp, mb * 16 + 4 * q, a.hd, dq[mb], a.scale);
        if (has_next) {
            T* nb = lds + (cur ^ 1) * BUF;
            sk.store(nb, nb + 2 * NK * HP, 1.f, tid); sv.store(nb + NK * HP, nullptr, 1.f, tid);
        }
        __syncthreads();
    }
    if (a.dbias) {       // this wave's block of the bias gradient, summed over its windows: ONE flush
        float* bp = a.dbias + ((long long)h * a.Nq + qrow) * a.Nk + 4 * q;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(bp + t * 16 + j, dbias_r[t][j]);
    }
}

// key-stationary pass: wave = 16 keys; NQT = Nq / 16 query tiles, all staged (Q scaled, dO, both also transposed)
template <typename T, int NQT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kv_p_kernel(const AttnArgs a) {
    constexpr int NQ = NQT * 16, NTH = NW * 64, TS = NQ + 8;
    constexpr int BUF = 2 * NQ * HP + 2 * HP * TS;           // Q [NQ][HP], dO [NQ][HP], Q^T [HP][TS], dO^T [HP][TS]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AT<T>::frag_t frag_t;
    T* lds = reinterpret_cast<T*>(smem);
    float* lds_f = reinterpret_cast<float*>(lds + 2 * BUF);  // [2][2][NQ]: lse, delta per buffer
    const PWalk w = pwalk(a);
    const int h = w.h, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int krow = wave * WQ + n;
    const long long rs = (long long)3 * a.nH * a.hd;
    auto q_base = [&](int b) {
        return reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.q0) * 3 + 0) * a.nH * a.hd + (long long)h * a.hd;
    };
    auto do_base = [&](int b) { return reinterpret_cast<const T*>(a.dout) + ((long long)b * a.N + a.o0) * a.Cout + a.c_off + h * a.hd; };
    f32x4_t bias_r[NQT];                                     // bias[h][query 16 t + 4 q + j][key krow]
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bias_r[t][j] = a.bias ? a.bias[((long long)h * a.Nq + t * 16 + 4 * q + j) * a.Nk + krow] : 0.f;
    Stager<T, NQ, NTH> sq, sd;
    float lse_r = 0.f, delta_r = 0.f;
    int b = w.part;
    if (b < a.B) {
        sq.load(q_base(b), rs, a.hd, tid); sd.load(do_base(b), a.Cout, a.hd, tid);
        sq.store(lds, lds + 2 * NQ * HP, a.scale, tid); sd.store(lds + NQ * HP, lds + 2 * NQ * HP + HP * TS, 1.f, tid);
        if (tid < NQ) { lds_f[tid] = a.lse[((long long)b * a.nH + h) * a.Nq + tid]; lds_f[NQ + tid] = a.delta[((long long)b * a.nH + h) * a.Nq + tid]; }
    }
    __syncthreads();
    for (int cur = 0; b < a.B; b += w.nparts, cur ^= 1) {
        const int bn = b + w.nparts;
        const bool has_next = bn < a.B;
        if (has_next) {
            sq.load(q_base(bn), rs, a.hd, tid); sd.load(do_base(bn), a.Cout, a.hd, tid);
            if (tid < NQ) { lse_r = a.lse[((long long)bn * a.nH + h) * a.Nq + tid]; delta_r = a.delta[((long long)bn * a.nH + h) * a.Nq + tid]; }
        }
        const T* lds_q = lds + cur * BUF;
        const T* lds_do = lds_q + NQ * HP;
        const T* lds_qt = lds_do + NQ * HP;
        const T* lds_dot = lds_qt + HP * TS;
        const float* l_lse = lds_f + cur * 2 * NQ;
        const float* l_delta = l_lse + NQ;
        const T* kp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 1) * a.nH * a.hd + (long long)h * a.hd;
        const T* vp = reinterpret_cast<const T*>(a.qkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 2) * a.nH * a.hd + (long long)h * a.hd;
        const frag_t kf = load_hd8<T>(kp, q, a.hd, 1.f);
        const frag_t vf = load_hd8<T>(vp, q, a.hd, 1.f);
        const float* mrow = (a.mask && !a.mask_bits) ? a.mask + (long long)(b % a.nW) * a.Nm * a.Nm + krow : nullptr;
        const unsigned* mbits = a.mask_bits ? a.mask_bits + (long long)(b % a.nW) * a.Nm * (a.Nm >> 5) + (krow >> 5) : nullptr;
        f32x4_t dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < NQT / 2; ++s) {
            frag_t pf, sf;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * s + half;
                const frag_t qf = *reinterpret_cast<const frag_t*>(lds_q + (t * 16 + n) * HP + 8 * q);
                const frag_t dof = *reinterpret_cast<const frag_t*>(lds_do + (t * 16 + n) * HP + 8 * q);
                f32x4_t sc = bias_r[t], dpp = {0.f, 0.f, 0.f, 0.f};
                amma(sc, qf, kf);
                amma(dpp, dof, vf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qi = t * 16 + 4 * q + j;
                    float sv = sc[j];
                    if (mrow) sv += mrow[(long long)qi * a.Nm];
                    if (mbits && ((mbits[(long long)qi * (a.Nm >> 5)] >> (krow & 31)) & 1u)) sv += a.mask_value;
                    const float p = __expf(sv - l_lse[qi]);
                    fset(pf, half * 4 + j, p);
                    fset(sf, half * 4 + j, p * (dpp[j] - l_delta[qi]));
                }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const T* dop = lds_dot + (mb * 16 + n) * TS + 32 * s + 4 * q;
                const T* qtp = lds_qt + (mb * 16 + n) * TS + 32 * s + 4 * q;
                frag_t df, qtf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fset(df, j, tof(dop[j])); fset(df, 4 + j, tof(dop[16 + j]));
                    fset(qtf, j, tof(qtp[j])); fset(qtf, 4 + j, tof(qtp[16 + j]));
                }
                amma(dv[mb], df, pf);
                amma(dk[mb], qtf, sf);
            }
        }
        T* gk = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 1) * a.nH * a.hd + (long long)h * a.hd;
        T* gv = reinterpret_cast<T*>(a.dqkv) + (((long long)b * a.N + a.k0 + krow) * 3 + 2) * a.nH * a.hd + (long long)h * a.hd;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) { store_hd4<T>(gk, mb * 16 + 4 * q, a.hd, dk[mb], 1.f); store_hd4<T>(gv, mb * 16 + 4 * q, a.hd, dv[mb], 1.f); }
        if (has_next) {
            T* nb = lds + (cur ^ 1) * BUF;
            sq.store(nb, nb + 2 * NQ * HP, a.scale, tid); sd.store(nb + NQ * HP, nb + 2 * NQ * HP + HP * TS, 1.f, tid);
            if (tid < NQ) { float* nf = lds_f + (cur ^ 1) * 2 * NQ; nf[tid] = lse_r; nf[NQ + tid] = delta_r; }
        }
        __syncthreads();
    }
}

// bits[w][i][j / 32] |= (mask[w][i][j] != 0) << (j % 32)
__global__ void mask_pack_kernel(const float* __restrict__ mask, unsigned* __restrict__ bits, long long nwords) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) {
        const float* p = mask + i * 32;
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) w |= (p[k] != 0.f ? 1u : 0u) << k;
        bits[i] = w;
    }
}

template <typename K>
int launch_attn_p(K kern, VsrDevOnce& once, const AttnArgs& a, int waves, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds); if (rc != VSR_OK) return rc; }
    // ~2048 resident threads per CU; one partition = 8 workgroups per head (one per XCD)
    const int per_cu = 2048 / (waves * 64);
    int pp = (vsr_num_cus() * per_cu) / (8 * a.nH);
    const int need = (a.B + 7) / 8;
    if (pp > need) pp = need;
    if (pp < 1) pp = 1;
    hipLaunchKernelGGL(kern, dim3(8 * a.nH * pp), dim3(waves * 64), lds, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

// dense[h][i][j] = table[index[i * idx_stride + j]][h]   (window_attention.py:146-148) and its adjoint
__global__ void rpb_gather_kernel(const float* __restrict__ table, const long long* __restrict__ index, int idx_stride, float* __restrict__ dense,
                                  int nH, int N) {
    const int total = nH * N * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int h = i / (N * N), r = i - h * N * N, y = r / N, x = r - y * N;
        dense[i] = table[index[(long long)y * idx_stride + x] * nH + h];
    }
}
__global__ void rpb_scatter_kernel(const float* __restrict__ ddense, const long long* __restrict__ index, int idx_stride, float* __restrict__ dtable,
                                   int nH, int N) {
    const int total = nH * N * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int h = i / (N * N), r = i - h * N * N, y = r / N, x = r - y * N;
        atomicAdd(dtable + index[(long long)y * idx_stride + x] * nH + h, ddense[i]);
    }
}

template <typename K>
int launch_attn(K kern, VsrDevOnce& once, const AttnArgs& a, int blocks_x, int waves, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) { const int rc = vsr_set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds); if (rc != VSR_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3(blocks_x, a.nH, a.B), dim3(waves * 64), lds, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int check_args(const VsrAttnDesc* d, const void* qkv) {
    if (!d || !qkv) return VSR_ERR_BADARG;
    if (d->dtype != VSR_F32 && d->dtype != VSR_BF16) return VSR_ERR_BADARG;
    if (d->B < 1 || d->N < 1 || d->heads < 1 || d->head_dim < 1 || d->head_dim > HP) return VSR_ERR_UNSUPPORTED;
    if (d->Nq < 32 || (d->Nq & 31) || d->Nq > 16 * MAXT) return VSR_ERR_UNSUPPORTED;
    if (d->Nk != 64 && d->Nk != 128 && d->Nk != 192 && d->Nk != 256 && d->Nk != 384) return VSR_ERR_UNSUPPORTED;   // instantiated key counts
    if (d->q0 < 0 || d->k0 < 0 || d->o0 < 0 || d->q0 + d->Nq > d->N || d->k0 + d->Nk > d->N || d->o0 + d->Nq > d->N) return VSR_ERR_BADARG;
    if (d->c_off < 0 || d->c_off + d->heads * d->head_dim > d->Cout) return VSR_ERR_BADARG;
    if (d->B > 65535 || d->heads > 65535) return VSR_ERR_UNSUPPORTED;
    return VSR_OK;
}

AttnArgs make_args(const VsrAttnDesc* d) {
    AttnArgs a = {};
    a.B = d->B; a.N = d->N; a.nH = d->heads; a.hd = d->head_dim; a.q0 = d->q0; a.k0 = d->k0; a.o0 = d->o0; a.Nq = d->Nq; a.Nk = d->Nk;
    a.Cout = d->Cout; a.c_off = d->c_off; a.scale = d->scale; a.nW = d->nW > 0 ? d->nW : 1; a.Nm = d->Nm;
    a.mask_value = d->mask_value;
    return a;
}

}  // namespace

// d->mask_packed: `mask` points at vsr_mask_pack's bit-packed form (persistent kernels only)
static int set_mask(const VsrAttnDesc* d, const float* mask, AttnArgs& a) {
    a.mask = nullptr; a.mask_bits = nullptr;
    if (!mask) return VSR_OK;
    if (d->mask_packed) {
        if (!(d->Nq == d->Nk && (d->Nk == 128 || d->Nk == 64)) || (d->Nm & 31)) return VSR_ERR_UNSUPPORTED;
        a.mask_bits = reinterpret_cast<const unsigned*>(mask);
    } else {
        a.mask = mask;
    }
    return VSR_OK;
}

extern "C" {

int vsr_mask_pack(const float* mask, unsigned* bits, int nW, int Nm, void* stream) {
    if (!mask || !bits || nW < 1 || Nm < 32 || (Nm & 31)) return VSR_ERR_BADARG;
    const long long nwords = (long long)nW * Nm * (Nm / 32);
    long long g = (nwords + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(mask_pack_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, mask, bits, nwords);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_rpb_gather(const float* table, const long long* index, int idx_stride, float* dense, int heads, int N, void* stream) {
    if (!table || !index || !dense || heads < 1 || N < 1 || idx_stride < N) return VSR_ERR_BADARG;
    hipLaunchKernelGGL(rpb_gather_kernel, dim3(cdiv(heads * N * N, 256)), dim3(256), 0, (hipStream_t)stream, table, index, idx_stride, dense, heads, N);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_rpb_scatter(const float* ddense, const long long* index, int idx_stride, float* dtable, int heads, int N, void* stream) {
    if (!ddense || !index || !dtable || heads < 1 || N < 1 || idx_stride < N) return VSR_ERR_BADARG;
    hipLaunchKernelGGL(rpb_scatter_kernel, dim3(cdiv(heads * N * N, 256)), dim3(256), 0, (hipStream_t)stream, ddense, index, idx_stride, dtable, heads, N);
    HIP_CHECK_RET(hipGetLastError());
    return VSR_OK;
}

int vsr_window_attention_fwd(const VsrAttnDesc* d, const void* qkv, const float* bias, const float* mask, void* out, float* lse, void* stream) {
    int rc = check_args(d, qkv);
    if (rc != VSR_OK) return rc;
    if (!out) return VSR_ERR_BADARG;
    AttnArgs a = make_args(d);
    a.qkv = qkv; a.out = out; a.lse = lse; a.bias = bias;
    rc = set_mask(d, mask, a);
    if (rc != VSR_OK) return rc;
    const int bx = 1, wv = d->Nq / WQ < MAXW ? d->Nq / WQ : MAXW;
    const size_t es = d->dtype == VSR_BF16 ? 2 : 4;
    const size_t lds = ((size_t)d->Nk * HP + (size_t)HP * (d->Nk + 8)) * es;
    if (d->Nq == d->Nk && (d->Nk == 128 || d->Nk == 64) && !vsr_env().attn_generic) {
        static VsrDevOnce p1, p2, p3, p4;
        if (d->Nk == 128) {
            const size_t l = (size_t)2 * (128 * HP + HP * 136) * es;
            if (d->dtype == VSR_BF16) return launch_attn_p(attn_fwd_p_kernel<bf16_t, 8, 8>, p1, a, 8, l, (hipStream_t)stream);
            return launch_attn_p(attn_fwd_p_kernel<float, 8, 8>, p2, a, 8, l, (hipStream_t)stream);
        }
        const size_t l = (size_t)2 * (64 * HP + HP * 72) * es;
        if (d->dtype == VSR_BF16) return launch_attn_p(attn_fwd_p_kernel<bf16_t, 4, 4>, p3, a, 4, l, (hipStream_t)stream);
        return launch_attn_p(attn_fwd_p_kernel<float, 4, 4>, p4, a, 4, l, (hipStream_t)stream);
    }
#define ATTN_NT_CASES(X) X(4) X(8) X(12) X(16) X(24)
#define X(NT_)                                                                                                        \
    if (d->Nk == 16 * NT_) {                                                                                          \
        static VsrDevOnce o1, o2;                                                                                     \
        if (d->dtype == VSR_BF16) return launch_attn(attn_fwd_kernel<bf16_t, NT_>, o1, a, bx, wv, lds, (hipStream_t)stream); \
        return launch_attn(attn_fwd_kernel<float, NT_>, o2, a, bx, wv, lds, (hipStream_t)stream);                        \
    }
    ATTN_NT_CASES(X)
#undef X
    return VSR_ERR_UNSUPPORTED;
}

int vsr_window_attention_bwd(const VsrAttnDesc* d, const void* qkv, const float* bias, const float* mask, const void* dout, const float* lse,
                             float* delta, void* dqkv, float* dbias, void* stream) {
    int rc = check_args(d, qkv);
    if (rc != VSR_OK) return rc;
    if (!dout || !lse || !delta || !dqkv) return VSR_ERR_BADARG;
    AttnArgs a = make_args(d);
    a.qkv = qkv; a.dout = dout; a.dqkv = dqkv; a.lse = const_cast<float*>(lse); a.delta = delta; a.bias = bias; a.dbias = dbias;
    rc = set_mask(d, mask, a);
    if (rc != VSR_OK) return rc;
    const size_t es = d->dtype == VSR_BF16 ? 2 : 4;
    hipStream_t st = (hipStream_t)stream;
    static VsrDevOnce o2, o4;
    const size_t lds1 = ((size_t)2 * d->Nk * HP + (size_t)HP * (d->Nk + 8)) * es;
    const int wq = d->Nq / WQ < MAXW ? d->Nq / WQ : MAXW, wk = d->Nk / WQ < MAXW ? d->Nk / WQ : MAXW;
    const int qc = d->Nq < QC ? d->Nq : QC;
    if (d->Nq % qc != 0) return VSR_ERR_UNSUPPORTED;
    const size_t lds2 = ((size_t)2 * qc * HP + (size_t)2 * HP * (qc + 8)) * es + (size_t)2 * qc * 4;
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return VSR_ERR_UNSUPPORTED;
    if (d->Nq == d->Nk && (d->Nk == 128 || d->Nk == 64) && !vsr_env().attn_generic) {
        static VsrDevOnce p1, p2, p3, p4, p5, p6, p7, p8;
        const int nn = d->Nk;
        const size_t lq = (size_t)2 * (2 * nn * HP + HP * (nn + 8)) * es;
        const size_t lk = (size_t)2 * (2 * nn * HP + 2 * HP * (nn + 8)) * es + (size_t)4 * nn * 4;
        if (nn == 128) {
            rc = d->dtype == VSR_BF16 ? launch_attn_p(attn_bwd_q_p_kernel<bf16_t, 8, 8>, p1, a, 8, lq, st)
                                      : launch_attn_p(attn_bwd_q_p_kernel<float, 8, 8>, p2, a, 8, lq, st);
            if (rc != VSR_OK) return rc;
            return d->dtype == VSR_BF16 ? launch_attn_p(attn_bwd_kv_p_kernel<bf16_t, 8, 8>, p3, a, 8, lk, st)
                                        : launch_attn_p(attn_bwd_kv_p_kernel<float, 8, 8>, p4, a, 8, lk, st);
        }
        rc = d->dtype == VSR_BF16 ? launch_attn_p(attn_bwd_q_p_kernel<bf16_t, 4, 4>, p5, a, 4, lq, st)
                                  : launch_attn_p(attn_bwd_q_p_kernel<float, 4, 4>, p6, a, 4, lq, st);
        if (rc != VSR_OK) return rc;
        return d->dtype == VSR_BF16 ? launch_attn_p(attn_bwd_kv_p_kernel<bf16_t, 4, 4>, p7, a, 4, lk, st)
                                    : launch_attn_p(attn_bwd_kv_p_kernel<float, 4, 4>, p8, a, 4, lk, st);
    }
    rc = VSR_ERR_UNSUPPORTED;
#define X(NT_)                                                                                                        \
    if (d->Nk == 16 * NT_) {                                                                                          \
        static VsrDevOnce o1, o3;                                                                                     \
        rc = d->dtype == VSR_BF16 ? launch_attn(attn_bwd_q_kernel<bf16_t, NT_>, o1, a, 1, wq, lds1, st) \
                                  : launch_attn(attn_bwd_q_kernel<float, NT_>, o3, a, 1, wq, lds1, st); \
    }
    ATTN_NT_CASES(X)
#undef X
    if (rc != VSR_OK) return rc;
    if (d->dtype == VSR_BF16) return launch_attn(attn_bwd_kv_kernel<bf16_t>, o2, a, cdiv(d->Nk, WQ * wk), wk, lds2, st);
    return launch_attn(attn_bwd_kv_kernel<float>, o4, a, cdiv(d->Nk, WQ * wk), wk, lds2, st);
}

}  // extern "C"
