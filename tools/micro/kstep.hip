// Micro-benchmark (round 6): what ONE wave per SIMD can sustain on the instruction mix of a slab-convolution K-step --
// 48 x v_mfma_f32_16x16x32_bf16 on 24 accumulators (two halves of 24), 22 x ds_read_b128 fragment reads, one barrier, four
// 1 KiB LDS-DMA pieces -- as the mix is built up piece by piece.  Cycles per step from s_memtime around 200 steps.
// build: hipcc --offload-arch=gfx950 -O3 -o kstep kstep.hip ; run: ./kstep
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
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int OFF>
__device__ __forceinline__ uint4 rd(unsigned addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void mm(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <int Q, int QEND, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (Q < QEND) {
        f(std::integral_constant<int, Q>{});
        static_for<Q + 1, QEND>(f);
    }
}

// MODE bits: 1 = fragment reads interleaved with the MFMAs (else: no reads, operands stay), 2 = a barrier per step,
// 4 = four LDS-DMA pieces per step, 32 = every slipped-in instruction behind a uniform run-time branch (not taken for the reads / DMA, eight
// more that ARE taken: the skipped slab pieces) as conv3x3_pslab_kernel's first version had them, 8 = the reads issued in one burst BEFORE each half's MFMAs (the round-5 order) instead of
// interleaved, 16 = conflicting read addresses (all lanes the same 16-byte slot column: 16-way)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const unsigned char* src, unsigned bytes, unsigned long long* out, float* sink, int steps, int fa, int fb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int frow = lane & 15, fg = lane >> 4;
    const int swz = ((frow >> 1) & 3) << 1;
    unsigned ad = base + frow * 128 + (((fg) ^ swz) << 4);
    if (MODE & 16) ad = base + frow * 256 + fg * 2048;  // same 16-byte slot for all 16 rows of a group
    f32x4_t acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 px0[3], wf0[8], px1[3], wf1[8];
    for (int i = 0; i < 3; ++i) px0[i] = px1[i] = make_uint4(lane, i, 3, 4);
    for (int i = 0; i < 8; ++i) wf0[i] = wf1[i] = make_uint4(lane, i, 5, 6);
    for (int i = tid; i < 16384; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u;
    __syncthreads();
    auto frag = [&](auto rc, uint4 (&px)[3], uint4 (&wf)[8], unsigned a0) __attribute__((always_inline)) {
        constexpr int q = decltype(rc)::value;
        if constexpr (q < 3)
            px[q] = rd<2048 * q>(a0);
        else
            wf[q - 3] = rd<2048 * (q - 3)>(a0 + 8192);
    };
    u32x4_t stg[4];
    for (int i = 0; i < 4; ++i) stg[i] = u32x4_t{0u, 0u, 0u, 0u};
    int sacc[4] = {fa, fb, steps, 1};
    int vacc[4] = {lane, tid, fg, frow};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int voff = ((blockIdx.x * 4 + wave) * 4096 + lane * 16) & (bytes - 1);
    for (int s = 0; s < steps; ++s) {
        // [A]
        if (MODE & 8) {
            static_for<0, 11>([&](auto rc) { frag(rc, px1, wf1, ad ^ 64u); });
        }
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((MODE & 8) ? 11 : 0) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 24>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            mm(wf0[q / 3], px0[q % 3], acc[q]);
            if (MODE & 128) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc[q & 3]) : "s"(steps));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 256) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(vacc[q & 3]) : "v"(lane));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (q < 11) {
                if ((MODE & 1) && !(MODE & 8)) {
                    __builtin_amdgcn_sched_barrier(0);
                    frag(std::integral_constant<int, q>{}, px1, wf1, ad ^ 64u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        });
        // [B]
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (MODE & 2) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // [C]
        if (MODE & 8) {
            static_for<0, 11>([&](auto rc) { frag(rc, px0, wf0, ad); });
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, 24>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            mm(wf1[q / 3], px1[q % 3], acc[q]);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 128) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc[q & 3]) : "s"(steps));
            if (MODE & 256) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vacc[q & 3]) : "v"(lane));
            if constexpr (q < 4) {
                if (MODE & 64) {
                    // register-staged copy of the same KiB: the piece requested a step ago goes to LDS, the next one is requested
                    asm volatile("ds_write_b128 %0, %1" ::"v"(base + 32768u + (wave * 4 + q) * 1024 + lane * 16), "v"(stg[q]) : "memory");
                    stg[q] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, q * 1024, 0);
                }
                if (MODE & 4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 32768 + (wave * 4 + q) * 1024),
                                                             16, voff, q * 1024, 0, 0);
            } else if constexpr (q - 4 < 11) {
                if ((MODE & 1) && !(MODE & 8)) {
                    if (!(MODE & 32) || fa) frag(std::integral_constant<int, q - 4>{}, px0, wf0, ad);
                }
            } else if constexpr (q - 15 < 8) {
                if (MODE & 32)
                    if (fb) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 40960), 16, voff, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.f) sink[0] = sum + (float)(sacc[0] + sacc[1] + sacc[2] + sacc[3]) + (float)(vacc[0] + vacc[1] + vacc[2] + vacc[3]);
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const unsigned char* src, unsigned bytes, unsigned long long* out, float* sink, int blocks, const char* what) {
    const int steps = 200;
    CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 80 * 1024, 0, src, bytes, out, sink, steps, 1, 0);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-64s blocks %4d: %7.1f clk per step (48 MFMAs: %.1f clk each)\n", what, blocks, s / h.size() / steps, s / h.size() / steps / 48);
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
    for (int blocks : {256, 512}) {
        run<0>(src, bytes, out, sink, blocks, "48 MFMAs");
        run<1>(src, bytes, out, sink, blocks, "+ 22 ds_read_b128 interleaved");
        run<1 + 16>(src, bytes, out, sink, blocks, "+ 22 ds_read_b128 interleaved, bank-conflicting addresses");
        run<8 + 1>(src, bytes, out, sink, blocks, "+ 22 ds_read_b128 in two bursts in front of the halves");
        run<1 + 2>(src, bytes, out, sink, blocks, "+ reads + barrier");
        run<1 + 2 + 4>(src, bytes, out, sink, blocks, "+ reads + barrier + 4 LDS-DMA pieces");
        run<1 + 2 + 4 + 32>(src, bytes, out, sink, blocks, "+ reads + barrier + DMA, each behind a uniform branch (+ 8 taken)");
        run<2 + 4>(src, bytes, out, sink, blocks, "MFMAs + barrier + 4 LDS-DMA pieces (no reads)");
        run<4>(src, bytes, out, sink, blocks, "MFMAs + 4 LDS-DMA pieces");
        run<128>(src, bytes, out, sink, blocks, "MFMAs + 48 scalar adds, one behind every MFMA");
        run<256>(src, bytes, out, sink, blocks, "MFMAs + 48 vector adds, one behind every MFMA");
        run<1 + 2 + 4 + 128>(src, bytes, out, sink, blocks, "+ reads + barrier + DMA + 48 scalar adds");
        run<1 + 2 + 4 + 256>(src, bytes, out, sink, blocks, "+ reads + barrier + DMA + 48 vector adds");
        run<64>(src, bytes, out, sink, blocks, "MFMAs + 4 register-staged KiB (buffer_load_dwordx4 + ds_write_b128)");
        run<1 + 2 + 64>(src, bytes, out, sink, blocks, "+ reads + barrier + 4 register-staged KiB");
    }
    return 0;
}
