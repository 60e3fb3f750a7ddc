// Micro-benchmark: issue cost of LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave
// instruction) vs plain global_load_dwordx4, as a function of waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue dma_issue.hip ; run: ./dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int NP, int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* src, unsigned bytes, unsigned long long* out, int waves, int stride_kb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave >= waves) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    // each wave reads its own region (L2-resident after the warm-up pass)
    int voff = ((blockIdx.x * 4 + wave) * NP * stride_kb) * 1024 + lane * 16;
    voff &= (bytes - 1);
    unsigned char* dst = smem + wave * NP * 1024;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int rep = 0; rep < 2; ++rep) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         voff + i * 1024 * stride_kb, 0, 0, 0);
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(src + voff + i * 1024 * stride_kb));
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) acc.x ^= v[i].x;
        }
        if (rep == 1 && lane == 0) {
            out[(blockIdx.x * 4 + wave) * 2 + 0] = t1 - t0;
            out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0;
        }
    }
    if (acc.x == 12345) out[0] = 0;
}

template <int NP, int MODE>
void run(const unsigned char* src, unsigned bytes, unsigned long long* out, int blocks, int waves, int stride_kb, const char* what) {
    CHECK(hipMemset(out, 0, blocks * 4 * 2 * 8));
    hipLaunchKernelGGL((k<NP, MODE>), dim3(blocks), dim3(256), 64 * 1024, 0, src, bytes, out, waves, stride_kb);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4 * 2);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double s0 = 0, s1 = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            s0 += h[(b * 4 + w) * 2];
            s1 += h[(b * 4 + w) * 2 + 1];
            ++n;
        }
    printf("%-28s blocks %4d waves/blk %d pieces %2d: issue %7.1f cyc/piece, issue->landed total %8.1f cyc (%.1f B/cyc/wave)\n", what,
           blocks, waves, NP, s0 / n / NP, s1 / n, NP * 1024.0 / (s1 / n));
}

int main() {
    const unsigned bytes = 1u << 28;
    unsigned char* src;
    unsigned long long* out;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&out, 4096 * 4 * 2 * 8));
    CHECK(hipFuncSetAttribute((const void*)k<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int blocks : {1, 256, 512}) {
        for (int waves : {1, 4}) {
            run<2, 0>(src, bytes, out, blocks, waves, 1, "lds-dma contiguous");
            run<8, 0>(src, bytes, out, blocks, waves, 1, "lds-dma contiguous");
            run<16, 0>(src, bytes, out, blocks, waves, 1, "lds-dma contiguous");
            run<8, 1>(src, bytes, out, blocks, waves, 1, "global_load_dwordx4");
            run<16, 1>(src, bytes, out, blocks, waves, 1, "global_load_dwordx4");
        }
    }
    return 0;
}
