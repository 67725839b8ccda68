// Read-only HBM bandwidth probe (round 4): what can a kernel that does nothing but stream 16-byte loads reach on MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o ab_tmp/bw_probe && ab_tmp/bw_probe
// One 512-thread workgroup per CU times WPC, every thread keeps U independent 16-byte loads in flight (grid-stride over 8 GiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int U>
__global__ __launch_bounds__(512) void rd(const u4* __restrict__ p, size_t n, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *out = acc;
}
template <int U>
__global__ __launch_bounds__(512) void rd_plain(const u4* __restrict__ p, size_t n, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *out = acc;
}
template <typename K>
static void run(const char* name, K kern, int wgs, const u4* p, size_t n, unsigned* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, p, n, out);
    hipEventRecord(e0, 0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, p, n, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %4d workgroups: %.2f TB/s\n", name, wgs, (double)n * 16 * it / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    u4* p; unsigned* out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4);
    hipMemset(p, 1, bytes);
    for (int wgs : {256, 512, 1024, 2048}) {
        run("nt loads, 4 in flight", rd<4>, wgs, p, n, out);
        run("nt loads, 8 in flight", rd<8>, wgs, p, n, out);
        run("nt loads, 16 in flight", rd<16>, wgs, p, n, out);
        run("plain loads, 8 in flight", rd_plain<8>, wgs, p, n, out);
        run("plain loads, 16 in flight", rd_plain<16>, wgs, p, n, out);
    }
    return 0;
}
