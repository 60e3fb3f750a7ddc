// Reproducer for the run-to-run different BatchNorm partial sums of the convolution epilogues when the library is
// built WITH the SLP vectoriser (csrc/Makefile, DESIGN.md "Toolchain note").  The per-channel fold of
// conv_epilogue::tile_sums -- __shfl_xor chains over (s1[e], s2[e]) -- becomes, with SLP on,
//     ds_bpermute_b32 vLo, addr, a ; ds_bpermute_b32 vHi, addr, b ; s_waitcnt lgkmcnt(n) ; v_pk_add_f32 v[Lo:Hi], ...
//     ds_bpermute_b32 x, addr2, vLo   <- reads the low half of the packed result in the very next issue slot
// This file runs that exact instruction sequence (inline asm, so that both builds of THIS file execute the same
// code) in three forms and counts lanes whose result differs from the scalar-add reference:
//   form 0: v_add_f32 x2 (what -fno-slp-vectorize emits)              -- reference
//   form 1: v_pk_add_f32, ds_bpermute of its halves right behind it  -- what SLP emits
//   form 2: form 1 with `s_nop 1` between the packed add and the first ds_bpermute that reads it
// build: hipcc --offload-arch=gfx950 -O3 -o slp_fold slp_fold.hip ; run: ./slp_fold [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// dynamic LDS bytes requested per block: 0 = as many waves per SIMD as fit (a wave's instructions are then rarely issued back
// to back), 100 KiB = one block per CU = one wave per SIMD (the convolution kernels run at two)
static size_t g_lds = 0;
static int g_mate = 0;  // 1: MFMA partner waves beside the accumulating ones

template <int FORM>
__global__ __launch_bounds__(256) void fold(const float2* __restrict__ in, float2* __restrict__ out, float* __restrict__ sink, int rounds) {
    const int lane = threadIdx.x & 63;
    const size_t gid = blockIdx.x * 256ull + threadIdx.x;
    float2 v = in[gid];
    const int a8 = ((lane ^ 8)) << 2, a16 = ((lane ^ 16)) << 2, a32 = ((lane ^ 32)) << 2;
    float2 acc = make_float2(0.f, 0.f);
    for (int r = 0; r < rounds; ++r) {
        // a store in flight and a data-dependent amount of VALU work in front: the timing of the sequence varies
        sink[gid] = v.x;
        float2 p = make_float2(v.x + (float)r, v.y - (float)r), t, q;
        if (FORM == 0) {
            asm volatile(
                "ds_bpermute_b32 %0, %4, %2\n ds_bpermute_b32 %1, %4, %3\n s_waitcnt lgkmcnt(0)\n"
                "v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n s_nop 1\n"
                "ds_bpermute_b32 %0, %5, %2\n ds_bpermute_b32 %1, %5, %3\n s_waitcnt lgkmcnt(0)\n"
                "v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n s_nop 1\n"
                "ds_bpermute_b32 %0, %6, %2\n ds_bpermute_b32 %1, %6, %3\n s_waitcnt lgkmcnt(0)\n"
                "v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n"
                : "=&v"(t.x), "=&v"(t.y), "+v"(p.x), "+v"(p.y)
                : "v"(a8), "v"(a16), "v"(a32)
                : "memory");
            q = p;
        } else {
            // the packed form needs the pair in an aligned register pair: fixed registers v[10:11] (pair) and v[12:13] (partner)
            if (FORM == 1)
                asm volatile(
                    "v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n s_nop 3\n"
                    "ds_bpermute_b32 v12, %4, v10\n ds_bpermute_b32 v13, %4, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n"
                    "ds_bpermute_b32 v12, %5, v10\n ds_bpermute_b32 v13, %5, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n"
                    "ds_bpermute_b32 v12, %6, v10\n ds_bpermute_b32 v13, %6, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n s_nop 1\n"
                    "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n"
                    : "=v"(q.x), "=v"(q.y)
                    : "v"(p.x), "v"(p.y), "v"(a8), "v"(a16), "v"(a32)
                    : "v10", "v11", "v12", "v13", "memory");
            else
                asm volatile(
                    "v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n s_nop 3\n"
                    "ds_bpermute_b32 v12, %4, v10\n ds_bpermute_b32 v13, %4, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n s_nop 1\n"
                    "ds_bpermute_b32 v12, %5, v10\n ds_bpermute_b32 v13, %5, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n s_nop 1\n"
                    "ds_bpermute_b32 v12, %6, v10\n ds_bpermute_b32 v13, %6, v11\n s_waitcnt lgkmcnt(0)\n"
                    "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n s_nop 1\n"
                    "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n"
                    : "=v"(q.x), "=v"(q.y)
                    : "v"(p.x), "v"(p.y), "v"(a8), "v"(a16), "v"(a32)
                    : "v10", "v11", "v12", "v13", "memory");
        }
        acc.x += q.x;
        acc.y += q.y;
        v.x = v.x * 1.0001f + 0.25f;
        v.y = v.y * 0.9999f - 0.125f;
    }
    out[gid] = acc;
}

// ---- second candidate: the ACCUMULATION of the store loop.  With SLP on, `ssum[e] += f[e]; ssq[e] += f[e] * f[e]` over the
// eight channels of a 16-byte vector becomes four pairs of
//     v_pk_add_f32 S, S, F op_sel:[0,1] op_sel_hi:[1,0]      (S = (ssum[2i], ssum[2i+1]), F = (f[2i+1], f[2i]): halves crossed)
//     v_pk_fma_f32 Q, F, F, Q
// behind the bf16 unpack (v_lshlrev_b32 / v_and_b32).  PACKED = 1 runs that sequence, PACKED = 0 the scalar one; both accumulate
// `rounds` vectors read from LDS and must agree bit for bit.
typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
// blockDim 512: waves 4-7 (one per SIMD, beside waves 0-3) issue nothing but MFMAs while waves 0-3 accumulate -- what a
// convolution block's epilogue sees when the CU's other block is in its K-loop
template <int PACKED>
__global__ __launch_bounds__(512) void accum(const uint4* __restrict__ in, float* __restrict__ out, int rounds) {
    __shared__ uint4 tile[256 * 9];
    if (threadIdx.x >= 256) {
        mf32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        mbf16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(1.0f + threadIdx.x * 0.001f + i), b[i] = (__bf16)(0.5f + i);
        for (int it = 0; it < rounds * 24; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        if (c[0][0] + c[1][0] + c[2][0] + c[3][0] == 12345.678f) out[0] = 1.f;  // keep
        return;
    }
    const int tid = threadIdx.x;
    for (int r = 0; r < 9; ++r) tile[r * 256 + tid] = in[(blockIdx.x * 9ull + r) * 256 + tid];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (each thread re-reads rows other threads wrote: see below)
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        const uint4 v = tile[(r % 9) * 256 + tid];
        out[(blockIdx.x * 256ull + tid) * 16 + (r & 15)] = __uint_as_float(v.x);  // a store in flight, as in the epilogue
        if (PACKED) {
            asm volatile(
                "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, %3\n"
                "v_mov_b32 v24, %4\n v_mov_b32 v25, %5\n v_mov_b32 v26, %6\n v_mov_b32 v27, %7\n"
                "v_mov_b32 v29, %8\n v_mov_b32 v28, %9\n v_mov_b32 v31, %10\n v_mov_b32 v30, %11\n"
                "v_mov_b32 v33, %12\n v_mov_b32 v32, %13\n v_mov_b32 v35, %14\n v_mov_b32 v34, %15\n"
                "s_nop 3\n"
                // (exactly the compiler's order: the four unpacks, then the packed pairs with nothing in between)
                "v_lshlrev_b32 v41, 16, %16\n v_and_b32 v40, 0xffff0000, %16\n"
                "v_lshlrev_b32 v43, 16, %17\n v_and_b32 v42, 0xffff0000, %17\n"
                "v_lshlrev_b32 v45, 16, %18\n v_and_b32 v44, 0xffff0000, %18\n"
                "v_lshlrev_b32 v47, 16, %19\n v_and_b32 v46, 0xffff0000, %19\n"
                "v_pk_add_f32 v[20:21], v[20:21], v[40:41] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "v_pk_fma_f32 v[28:29], v[40:41], v[40:41], v[28:29]\n"
                "v_pk_add_f32 v[22:23], v[22:23], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "v_pk_fma_f32 v[30:31], v[42:43], v[42:43], v[30:31]\n"
                "v_pk_add_f32 v[24:25], v[24:25], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "v_pk_fma_f32 v[32:33], v[44:45], v[44:45], v[32:33]\n"
                "v_pk_add_f32 v[26:27], v[26:27], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "v_pk_fma_f32 v[34:35], v[46:47], v[46:47], v[34:35]\n"
                "s_nop 3\n"
                "v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n v_mov_b32 %2, v22\n v_mov_b32 %3, v23\n"
                "v_mov_b32 %4, v24\n v_mov_b32 %5, v25\n v_mov_b32 %6, v26\n v_mov_b32 %7, v27\n"
                "v_mov_b32 %8, v29\n v_mov_b32 %9, v28\n v_mov_b32 %10, v31\n v_mov_b32 %11, v30\n"
                "v_mov_b32 %12, v33\n v_mov_b32 %13, v32\n v_mov_b32 %14, v35\n v_mov_b32 %15, v34\n"
                : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(q[0]), "+v"(q[1]),
                  "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)
                : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40",
                  "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        } else {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f0 = __uint_as_float(w[i] << 16), f1 = __uint_as_float(w[i] & 0xffff0000u);
                s[2 * i] += f0;
                s[2 * i + 1] += f1;
                q[2 * i] = __builtin_fmaf(f0, f0, q[2 * i]);
                q[2 * i + 1] = __builtin_fmaf(f1, f1, q[2 * i + 1]);
                asm volatile("" : "+v"(s[2 * i]), "+v"(s[2 * i + 1]), "+v"(q[2 * i]), "+v"(q[2 * i + 1]));
            }
        }
    }
    float* o = out + (size_t)gridDim.x * 256 * 16 + (blockIdx.x * 256ull + tid) * 16;
    for (int e = 0; e < 8; ++e) o[e] = s[e], o[8 + e] = q[e];
}

static void run_accum(int rounds) {
    const int blocks = 2048;
    const size_t n = (size_t)blocks * 9 * 256;
    std::vector<uint4> h(n);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; ++i) {
        unsigned w[4];
        for (int k = 0; k < 4; ++k) {
            x = x * 1664525u + 1013904223u;
            const unsigned lo = 0x3c00u + ((x >> 8) & 0x7ffu) + ((x >> 20) & 1u) * 0x8000u;  // bf16 in (+-)[2^-7, 2^9)
            x = x * 1664525u + 1013904223u;
            const unsigned hi = 0x3c00u + ((x >> 8) & 0x7ffu) + ((x >> 20) & 1u) * 0x8000u;
            w[k] = lo | (hi << 16);
        }
        h[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint4* in;
    float* o[2];
    CHECK(hipMalloc(&in, n * 16));
    CHECK(hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice));
    const size_t on = (size_t)blocks * 256 * 32;
    for (int f = 0; f < 2; ++f) CHECK(hipMalloc(&o[f], on * 4));
    std::vector<float> r0(on), r1(on);
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(accum<0>, dim3(blocks), dim3(g_mate ? 512 : 256), g_lds, 0, in, o[0], rounds);
        hipLaunchKernelGGL(accum<1>, dim3(blocks), dim3(g_mate ? 512 : 256), g_lds, 0, in, o[1], rounds);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(r0.data(), o[0], on * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(r1.data(), o[1], on * 4, hipMemcpyDeviceToHost));
        size_t ds = 0, dq = 0;
        const size_t base = (size_t)blocks * 256 * 16;
        for (size_t t = 0; t < (size_t)blocks * 256; ++t)
            for (int e = 0; e < 8; ++e) {
                ds += r0[base + t * 16 + e] != r1[base + t * 16 + e];
                dq += r0[base + t * 16 + 8 + e] != r1[base + t * 16 + 8 + e];
            }
        printf("accum rep %d: packed (v_pk_add_f32 op_sel + v_pk_fma_f32) vs scalar: %zu sums / %zu sums of squares differ (of %zu each)\n", rep, ds,
               dq, (size_t)blocks * 256 * 8);
    }
}

int main(int argc, char** argv) {
    g_lds = argc > 2 ? (size_t)atoi(argv[2]) * 1024 : 0;
    g_mate = argc > 3 ? atoi(argv[3]) : 0;
    if (g_lds) {
        CHECK(hipFuncSetAttribute((const void*)accum<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
        CHECK(hipFuncSetAttribute((const void*)accum<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
        CHECK(hipFuncSetAttribute((const void*)fold<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
        CHECK(hipFuncSetAttribute((const void*)fold<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
        CHECK(hipFuncSetAttribute((const void*)fold<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
    }
    printf("dynamic LDS per block: %zu bytes, MFMA partner waves: %d\n", g_lds, g_mate);
    run_accum(argc > 1 ? atoi(argv[1]) : 64);
    const int rounds = argc > 1 ? atoi(argv[1]) : 64, blocks = 4096;
    const size_t n = (size_t)blocks * 256;
    std::vector<float2> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 0.37f - 180.f, (float)((i * 40503u) % 777) * 0.11f);
    float2 *in, *o[3];
    float* sink;
    CHECK(hipMalloc(&in, n * 8));
    CHECK(hipMalloc(&sink, n * 4));
    CHECK(hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice));
    for (int f = 0; f < 3; ++f) CHECK(hipMalloc(&o[f], n * 8));
    std::vector<float2> r[3];
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(fold<0>, dim3(blocks), dim3(256), g_lds, 0, in, o[0], sink, rounds);
        hipLaunchKernelGGL(fold<1>, dim3(blocks), dim3(256), g_lds, 0, in, o[1], sink, rounds);
        hipLaunchKernelGGL(fold<2>, dim3(blocks), dim3(256), g_lds, 0, in, o[2], sink, rounds);
        CHECK(hipDeviceSynchronize());
        for (int f = 0; f < 3; ++f) {
            r[f].resize(n);
            CHECK(hipMemcpy(r[f].data(), o[f], n * 8, hipMemcpyDeviceToHost));
        }
        size_t d1x = 0, d1y = 0, d2x = 0, d2y = 0;
        for (size_t i = 0; i < n; ++i) {
            d1x += r[1][i].x != r[0][i].x;
            d1y += r[1][i].y != r[0][i].y;
            d2x += r[2][i].x != r[0][i].x;
            d2y += r[2][i].y != r[0][i].y;
        }
        printf("rep %d: v_pk_add_f32 + ds_bpermute back to back: %zu low / %zu high halves differ from the scalar form;  with s_nop 1: %zu / %zu   (of %zu lanes)\n",
               rep, d1x, d1y, d2x, d2y, n);
    }
    return 0;
}
