// Micro-benchmark (round 6): what ONE wave per SIMD can sustain on the instruction mix of a conv_wgrad9 stage -- 72 x
// v_mfma_f32_16x16x32_bf16 on 36 accumulators (18 steps of 4: nine taps x two 32-pixel K-steps), 16 + 36 ds_read_b64_tr_b16
// fragment reads and 4 ds_read_b32 mask reads, two VALU selects per x-fragment read, one barrier, five 1 KiB LDS-DMA pieces -- in the
// kernel's order (reads + DMA burst behind the barrier, then the 18 steps with counted waits) and in alternatives.
// build: hipcc --offload-arch=gfx950 -O3 -o w9stage w9stage.hip ; run: ./w9stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                 \
    do {                                                         \
        hipError_t e = (x);                                      \
        if (e != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                             \
        }                                                        \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int OFF>
__device__ __forceinline__ uint2 tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned rd32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int Q, int QEND, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (Q < QEND) {
        f(std::integral_constant<int, Q>{});
        static_for<Q + 1, QEND>(f);
    }
}

// MODE bits: 1 = the five DMA pieces as a burst behind the barrier (the kernel's order); 2 = instead one piece behind the MFMAs
// of steps 1, 4, 7, 10, 13 (directly between two MFMA groups); 4 = no counted waits inside the 18 steps (one lgkmcnt(0) per
// K-step: a lower bound for the read latency handling); 8 = x-fragment lookahead 6 instead of 3; 16 = no VALU selects (the
// addresses are used as they are); 32 = the pieces of mode 2 between the 2nd and 3rd MFMA of their step (bare MFMAs around them)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const unsigned char* src, unsigned bytes, unsigned long long* out, float* sink, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int li = lane & 15, g = lane >> 4;
    const int lrow = g * 8 + (li >> 2);
    unsigned aaddr[4][2], baddr[9][2];
    for (int h = 0; h < 2; ++h) {
        const int row = lrow + h * 4;
        for (int i = 0; i < 4; ++i) aaddr[i][h] = base + row * 128 + ((i ^ ((row >> 1) & 1)) << 5) + (li & 3) * 8;
        for (int t = 0; t < 9; ++t) {
            const int sr = row + 29 + (t / 3 - 1) * 28 + (t % 3 - 1);
            baddr[t][h] = base + 8192 + sr * 128 + ((wave ^ ((sr >> 1) & 1)) << 5) + (li & 3) * 8;
        }
    }
    const unsigned zaddr = base + 49152 + (li & 3) * 8, maddr = base + 50176 + lrow * 4;
    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 16384; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u;
    for (int i = tid; i < 512; i += 256) ((unsigned*)(smem + 49152))[i] = (i < 256) ? 0u : 0x1ffu;
    __syncthreads();
    int voff = ((blockIdx.x * 4 + wave) * 8192 + lane * 16) & (bytes - 1);
    auto dma = [&](int q) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 65536 + (wave * 5 + q) * 1024), 16, voff,
                                                 q * 1024, 0, 0);
    };
    constexpr int AHEAD = (MODE & 8) ? 6 : 3, A1 = 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        unsigned pmask[2][2];
        uint2 af[2][4][2];
        pmask[0][0] = rd32<0>(maddr);
        pmask[0][1] = rd32<16>(maddr);
        pmask[1][0] = rd32<128>(maddr);
        pmask[1][1] = rd32<144>(maddr);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) af[0][i][h] = tr<0>(aaddr[i][h]);
        if (MODE & 1) {
#pragma unroll
            for (int q = 0; q < 5; ++q) dma(q);
        }
        lgkm<0>();
        uint2 bf[AHEAD + 1][2];
        auto issue_b = [&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr int ks = u / 9, t = u % 9;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned ad = baddr[t][h];
                if (!(MODE & 16)) {
                    unsigned sel;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(pmask[ks][h]), "n"(t));
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(sel), "v"(baddr[t][h]), "v"(ks ? zaddr - 4096u : zaddr));
                }
                bf[u % (AHEAD + 1)][h] = ks ? tr<4096>(ad) : tr<0>(ad);
            }
        };
        static_for<0, AHEAD>([&](auto uc) { issue_b(uc); });
        static_for<0, 18>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int ks = u / 9, t = u % 9;
            if constexpr (u + AHEAD < 18) issue_b(std::integral_constant<int, u + AHEAD>{});
            if constexpr (u == A1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) af[1][i][h] = tr<4096>(aaddr[i][h]);
            }
            constexpr int nb = (18 - 1 - u < AHEAD ? 18 - 1 - u : AHEAD) * 2;
            constexpr bool a1y = u >= A1 && u <= A1 + AHEAD && u < 9;
            if constexpr (MODE & 4) {
                if constexpr (u == 0 || u == 9) lgkm<0>();
            } else {
                lgkm<(nb + (a1y ? 8 : 0) > 15 ? 15 : nb + (a1y ? 8 : 0))>();
            }
            const uint2* b2 = bf[u % (AHEAD + 1)];
            const uint4 fb = make_uint4(b2[0].x, b2[0].y, b2[1].x, b2[1].y);
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const uint4 fa = make_uint4(af[ks][i][0].x, af[ks][i][0].y, af[ks][i][1].x, af[ks][i][1].y);
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc[t][i],
                                                                   0, 0, 0);
                if constexpr ((MODE & 32) != 0 && i == 1 && u % 3 == 1 && u < 15) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma(u / 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr ((MODE & 2) != 0 && u % 3 == 1 && u < 15) {
                __builtin_amdgcn_sched_barrier(0);
                dma(u / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) aaddr[i][h] ^= 16384u;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) baddr[t][h] = (baddr[t][h] + 8192u) & 0x7fffu | (baddr[t][h] & 0xffff8000u);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
    if (sum == 12345.f) sink[0] = sum;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const unsigned char* src, unsigned bytes, unsigned long long* out, float* sink, int blocks, const char* what) {
    const int steps = 200;
    CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 96 * 1024, 0, src, bytes, out, sink, steps);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-78s %7.1f clk per stage (72 MFMAs: %.1f clk each)\n", what, s / h.size() / steps, s / h.size() / steps / 72);
}

int main() {
    const unsigned bytes = 1u << 26;
    unsigned char* src;
    unsigned long long* out;
    float* sink;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&out, 2048 * 4 * 8));
    CHECK(hipMalloc(&sink, 16));
    const int blocks = 256;
    run<0>(src, bytes, out, sink, blocks, "reads + selects + counted waits + barrier, no DMA");
    run<1>(src, bytes, out, sink, blocks, "+ five DMA pieces as a burst behind the barrier (the kernel's order)");
    run<2>(src, bytes, out, sink, blocks, "+ five DMA pieces, one behind the MFMAs of steps 1, 4, 7, 10, 13");
    run<32>(src, bytes, out, sink, blocks, "+ five DMA pieces, one between the 2nd and 3rd MFMA of those steps");
    run<4>(src, bytes, out, sink, blocks, "no DMA, one lgkmcnt(0) per K-step instead of counted waits");
    run<8>(src, bytes, out, sink, blocks, "no DMA, x-fragment lookahead 6");
    run<8 + 32>(src, bytes, out, sink, blocks, "lookahead 6 + pieces between MFMAs");
    run<16>(src, bytes, out, sink, blocks, "no DMA, no VALU selects");
    run<16 + 8>(src, bytes, out, sink, blocks, "no DMA, no VALU selects, lookahead 6");
    return 0;
}
