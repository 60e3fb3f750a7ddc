// Effective shader clock under load: s_memtime (shader cycles) against wall_clock64 (constant 100 MHz) around
//   (a) a bare-MFMA loop on every SIMD, (b) a float4 streaming copy, (c) both at once (two streams).
// build: hipcc --offload-arch=gfx950 -O3 -o clk clk.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* out, int iters, float seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = c1 - c0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (unsigned long long)s; }
}
__global__ __launch_bounds__(256) void copy_loop(const float4* src, float4* dst, size_t n, unsigned long long* out, int reps) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) dst[i] = src[i];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = c1 - c0; out[blockIdx.x * 4 + 1] = w1 - w0; }
}
static void report(const char* what, unsigned long long* d, int blocks, float ms, double work, const char* unit) {
    unsigned long long h[4096 * 4];
    CHECK(hipMemcpy(h, d, blocks * 32, hipMemcpyDeviceToHost));
    double c = 0, w = 0;
    for (int i = 0; i < blocks; ++i) { c += h[i * 4]; w += h[i * 4 + 1]; }
    printf("%-28s %8.2f ms  effective shader clock %6.0f MHz  (%.1f %s)\n", what, ms, c / w * 100.0, work / (ms * 1e-3) / 1e12, unit);
}
int main() {
    unsigned long long *o1, *o2;
    CHECK(hipMalloc(&o1, 4096 * 32));
    CHECK(hipMalloc(&o2, 4096 * 32));
    const size_t n = (1ull << 30) / 16;  // 1 GiB
    float4 *src, *dst;
    CHECK(hipMalloc(&src, n * 16));
    CHECK(hipMalloc(&dst, n * 16));
    CHECK(hipMemset(src, 1, n * 16));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms;
    const int iters = 400000;  // x8 MFMA per wave
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(mfma_loop, dim3(1024), dim3(256), 0, s1, o1, iters, 1.0f);
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("bare MFMA, 4 waves/SIMD", o1, 1024, ms, 1024.0 * 4 * iters * 8 * 16384.0, "PFLOP/s x1e-3=TF");
        CHECK(hipEventRecord(e0, s2));
        hipLaunchKernelGGL(copy_loop, dim3(2048), dim3(256), 0, s2, src, dst, n, o2, 40);
        CHECK(hipEventRecord(e1, s2));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("float4 copy 1 GiB x40", o2, 2048, ms, 40.0 * 2 * n * 16, "TB/s");
        // both at once
        CHECK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(mfma_loop, dim3(512), dim3(256), 0, s1, o1, iters, 1.0f);
        hipLaunchKernelGGL(copy_loop, dim3(2048), dim3(256), 0, s2, src, dst, n, o2, 40);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("  concurrent: MFMA part", o1, 512, ms, 512.0 * 4 * iters * 8 * 16384.0, "TF (lower bound)");
        report("  concurrent: copy part", o2, 2048, ms, 40.0 * 2 * n * 16, "TB/s (lower bound)");
    }
    return 0;
}
