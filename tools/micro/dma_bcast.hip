// Micro-benchmark: the convolution kernels' weight-tile traffic pattern in isolation.
// Every block of a 256- or 512-block grid DMA's a 16 KiB tile (128 rows x 128 B) per iteration into a 2-deep LDS ring
// (4 waves x 4 LDS-DMA pieces, one barrier + vmcnt(0) per iteration, as conv3x3_slab_kernel's K-loop does), nothing else.
//   same = 1: all blocks fetch the SAME tile sequence (the weights of a convolution: every M-tile reads them) -- L2-hot lines
//   same = 0: every block its own region
//   rowstride: bytes between consecutive tile rows in global memory (128 = contiguous tile, 2304 = [K][3][3][C=128] rows)
// Prints cycles per iteration (s_memtime) and wall time.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_bcast dma_bcast.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(const unsigned char* src, unsigned bytes, unsigned long long* out, int iters, int same,
                                         int rowstride, int pieces_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const int tile_bytes = 128 * rowstride;  // footprint of one tile in global memory
    const unsigned base = same ? 0u : (unsigned)blockIdx.x * (unsigned)(iters > 64 ? 64 : iters) * (unsigned)tile_bytes;
    unsigned long long t0 = 0;
    for (int it = 0; it < iters + 1; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it == 1) t0 = __builtin_amdgcn_s_memtime();
        unsigned char* dst = smem + (it & 1) * 16384;
        const unsigned tb = base + (unsigned)(it & 63) * (unsigned)tile_bytes;
        for (int i = 0; i < pieces_per_wave; ++i) {
            const int row = (wave * pieces_per_wave + i) * 8 + (lane >> 3);  // 8 rows of 128 B per piece
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + (wave * pieces_per_wave + i) * 1024), 16,
                                                     (int)((tb + (unsigned)row * rowstride + (lane & 7) * 16) & (bytes - 1)), 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const unsigned bytes = 1u << 30;
    unsigned char* src;
    unsigned long long* out;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&out, 4096 * 8));
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const int iters = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int lds_kb : {80, 48}) {  // 80 KB -> 2 blocks per CU, 48 KB -> 3
        for (int blocks : {256, 512, 768}) {
            for (int same : {1, 0}) {
                for (int rowstride : {128, 2304}) {
                    for (int ppw : {4, 2}) {
                        for (int rep = 0; rep < 2; ++rep) {
                            CHECK(hipEventRecord(e0));
                            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_kb * 1024, 0, src, bytes, out, iters, same, rowstride, ppw);
                            CHECK(hipEventRecord(e1));
                            CHECK(hipDeviceSynchronize());
                        }
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        std::vector<unsigned long long> h(blocks);
                        CHECK(hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost));
                        double s = 0;
                        for (auto v : h) s += v;
                        printf("lds %2d KB blocks %4d same %d rowstride %4d pieces/wave %d: %7.1f cyc/iter per block, wall %.1f us -> %.2f TB/s L2->LDS\n",
                               lds_kb, blocks, same, rowstride, ppw, s / blocks / iters, ms * 1e3,
                               (double)blocks * iters * ppw * 4 * 1024 / (ms * 1e-3) / 1e12);
                    }
                }
            }
        }
    }
    return 0;
}
