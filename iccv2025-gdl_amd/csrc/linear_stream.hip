// linear_stream.hip -- streaming GEMM for the short-K Linears of the Swin branch (round 4).
//
// nn.Linear (/root/reference/models/swin_transformer.py:30-46 Mlp.fc1, :98-101 qkv / proj) at stages 1-2 of Swin-T:
//     out[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]) (+ addend[m][n]),   gelu_out = gelu(out)          K = 128 or 192
// over M = 602 112 / 150 528 token rows.  These GEMMs are HBM-bound by two orders of magnitude (0.5 flop/B of weights aside), and
// the tile-per-block implicit-GEMM kernel spends a block's life on them waiting: request both K-steps, wait one HBM latency, 2-3
// K-steps of MFMAs, stage, store, exit -- 2.3-3.0 TB/s (tools/bench_gemm.py).  Here NOTHING is shared between waves:
//   * a wave keeps the weights of its N-chunk in REGISTERS for the whole launch (the MFMA operand of lane l is 16 contiguous
//     bytes of one weight row: one global_load_dwordx4 each, once);
//   * it walks over 16-row tiles of A with a stride; a lane's operand of a tile is again 16 contiguous bytes of one A row, loaded
//     straight into registers, the NEXT tile's operands requested before the current tile's MFMAs (no LDS-DMA, no barrier);
//   * the 16 x NC accumulator tile goes through a wave-private LDS patch (4 KiB) to become full 16-byte row chunks; bias, residual
//     addend, rounding and GELU follow conv_epilogue's order exactly (round to bf16, add in fp32, round again), so the results are
//     bit-identical to the tile kernel's.
// The N-chunks of one row range run on the same XCD at the same time (block -> (xcd, slot, chunk) below), so A comes from HBM
// once and from that XCD's L2 for the other chunks.
#include "common.h"
#include "ops.h"
#include "prof.h"

#include <hip/hip_ext.h>

namespace gdl {

struct LinArgs {
    const bf16* A;       // [M][K]
    const bf16* W;       // [N][K]
    bf16* out;           // [M][N]
    const bf16* addend;  // optional [M][N]
    const float* bias;   // optional [N]
    bf16* gelu_out;      // optional [M][N]
    int M, K, N;
    int nchunks;         // N / (NF * 16)
    int mslots;          // blocks per (XCD, chunk)
};

// KF: K / 32 (4 or 6); NF: 16-column fragments of a wave's N-chunk (8: 128 columns, 6: 96 columns); ADD: the launch has an addend
//
// Memory pipeline of a wave: ONE wait per tile.  Loads and stores share `vmcnt` and a wait in front of a load's first use waits for
// everything older -- so a load issued inside a tile and used inside the same tile (the next tile's operands requested in front of
// this tile's MFMAs, an addend fetched in the epilogue) made every tile wait for an HBM round trip of its own (9 100 clk per tile,
// 2.8 TB/s in the first version of this kernel).  Here everything a tile needs -- its A operands, its addend chunks -- is requested
// during the PREVIOUS tile, right behind that tile's single `s_waitcnt vmcnt(0)`, and the chunk's bias sits in LDS for the whole
// launch: between the wait at the top of a tile and its stores nothing depends on global memory.
template <int KF, int NF, bool ADD>
__global__ __launch_bounds__(256, 2) void linear_stream_kernel(LinArgs a) {
    constexpr int NC = NF * 16, K = KF * 32;
    constexpr int PITCH = NC * 2 + 16;   // bytes of a staged row
    constexpr int CPR = NC / 8;          // 16-byte chunks per row
    constexpr int NIT = 16 * CPR / 64;   // chunks per lane (4 or 3)
    static_assert(16 * CPR % 64 == 0, "row chunks must divide among the lanes");
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[4][16 * PITCH];
    __shared__ __attribute__((aligned(16))) float bias_s[NC];  // the chunk's bias (zeros without one): LDS reads do not touch vmcnt
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* stg = stage_all[wave];
    // block -> (xcd, chunk, slot): the chunks of a slot are neighbours on one XCD (block b runs on XCD b % 8)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nch = j % a.nchunks, slot = j / a.nchunks;
    const int n0 = nch * NC;
    const int nwav = 8 * a.mslots * 4;                   // row-walking waves per chunk
    const int wid = (xcd * a.mslots + slot) * 4 + wave;  // this wave's index among them
    const int ntile = (a.M + 15) >> 4;
    const int fr = lane & 15, fk = (lane >> 4) * 8;      // operand row / first k of the lane
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- this wave's weights: operand (n, k) = W[n0 + 16 n + fr][32 k + fk .. + 8)
    u32x4 wreg[NF][KF];
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int k = 0; k < KF; ++k) wreg[n][k] = *(const u32x4*)(a.W + (size_t)(n0 + n * 16 + fr) * K + k * 32 + fk);
    // ---- the lane's row chunks: q = it * 64 + lane -> row q / CPR, column chunk q % CPR; their bias values
    int crow[NIT], cch[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * 64 + lane;
        crow[it] = q / CPR;
        cch[it] = q - crow[it] * CPR;
    }
    if (threadIdx.x < NC) bias_s[threadIdx.x] = a.bias ? a.bias[n0 + threadIdx.x] : 0.f;
    __syncthreads();  // (the only block-wide synchronisation of the launch)

    struct Ops {
        u32x4 a[KF];
        uint4 add[ADD ? NIT : 1];
    };
    auto request = [&](int tile, Ops& o) __attribute__((always_inline)) {
        int m = tile * 16 + fr;
        if (m >= a.M) m = a.M - 1;  // (ragged last tile: rows past the end repeat the last one and are not stored)
        const bf16* p = a.A + (size_t)m * K + fk;
#pragma unroll
        for (int k = 0; k < KF; ++k) o.a[k] = *(const u32x4*)(p + k * 32);
        if constexpr (ADD) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int mr = tile * 16 + crow[it];
                if (mr >= a.M) mr = a.M - 1;
                o.add[it] = *(const uint4*)(a.addend + (size_t)mr * a.N + n0 + cch[it] * 8);
            }
        }
    };
    // one tile: wait for what was requested during the previous tile, request the next tile's, multiply, stage, finish, store
    auto process = [&](int tile, Ops& cu, Ops& nx) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < KF; ++k) asm volatile("" : "+v"(cu.a[k]));  // (they have arrived: no compiler wait at their first use)
        if constexpr (ADD) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(cu.add[it].x), "+v"(cu.add[it].y), "+v"(cu.add[it].z), "+v"(cu.add[it].w));
        }
        if (tile + nwav < ntile) request(tile + nwav, nx);
        f32x4_t acc[NF];
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // (K in ascending 32-wide steps, as the tile kernel's K-loop: the same fp32 accumulation order)
#pragma unroll
        for (int k = 0; k < KF; ++k)
#pragma unroll
            for (int n = 0; n < NF; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[n][k]),
                                                                __builtin_bit_cast(bf16x8_t, cu.a[k]), acc[n], 0, 0, 0);
        // D[i][j]: i = column n0 + 16 n + (lane >> 4) * 4 + reg, j = row fr  ->  the wave's LDS patch [16 rows][NC] bf16
#pragma unroll
        for (int n = 0; n < NF; ++n)
            *(uint2*)(stg + fr * PITCH + (n * 16 + (lane >> 4) * 4) * 2) =
                make_uint2(pack2bf(acc[n][0], acc[n][1]), pack2bf(acc[n][2], acc[n][3]));
        // (one wave: its own LDS writes are visible to its own later reads, in order)
        const int m0 = tile * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = m0 + crow[it];
            uint4 v = *(const uint4*)(stg + crow[it] * PITCH + cch[it] * 16);
            if (m >= a.M) continue;
            const size_t goff = (size_t)m * a.N + n0 + cch[it] * 8;
            if (ADD || a.bias) {
                float f[8];
                unpack16<bf16>(v, f);
                if (a.bias) {
                    const float4 b0 = *(const float4*)(bias_s + cch[it] * 8), b1 = *(const float4*)(bias_s + cch[it] * 8 + 4);
                    f[0] += b0.x, f[1] += b0.y, f[2] += b0.z, f[3] += b0.w;
                    f[4] += b1.x, f[5] += b1.y, f[6] += b1.z, f[7] += b1.w;
                }
                if constexpr (ADD) {
                    float g[8];
                    unpack16<bf16>(cu.add[it], g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                }
                v = pack16<bf16>(f);
            }
            *(uint4*)(a.out + goff) = v;
            if (a.gelu_out) {  // exact GELU of the value as stored
                float f[8];
                unpack16<bf16>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = gelu_val<bf16>(f[e]);
                *(uint4*)(a.gelu_out + goff) = pack16<bf16>(f);
            }
        }
    };
    Ops o0, o1;  // (two named sets, not an indexed pair: a run-time index would put them in scratch memory)
    int tile = wid;
    if (tile < ntile) request(tile, o0);
    while (tile < ntile) {
        process(tile, o0, o1);
        tile += nwav;
        if (tile >= ntile) break;
        process(tile, o1, o0);
        tile += nwav;
    }
}

// true if this plain GEMM runs on the streaming kernel
bool linear_stream_ok(int dtype, int M, int K, int N, bool has_addend) {
    static int on = -1;
    if (on < 0) {
        const char* e = tune_env("GDL_LINEAR_STREAM");  // tuning aid: 0 = the tile kernel everywhere
        on = e ? atoi(e) : 1;
    }
    if (!on || dtype != GDL_BF16 || M < 16384) return false;
    if (K == 128) return N % 128 == 0;
    if (K == 192) return N % 96 == 0 && !has_addend;  // (no registers left for the addend's prefetch at K = 192: tile kernel)
    return false;
}

int linear_stream_fwd(const void* A, const void* W, void* out, const void* addend, const float* bias, void* gelu_out, int M, int K,
                      int N, hipStream_t st) {
    GDL_REQUIRE(linear_stream_ok(GDL_BF16, M, K, N, addend != nullptr), "linear_stream: unsupported shape %d x %d -> %d", M, K, N);
    GDL_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)addend | (uintptr_t)gelu_out | (uintptr_t)bias) & 15) == 0,
                "linear_stream: operands must be 16-byte aligned");
    LinArgs a{};
    a.A = (const bf16*)A, a.W = (const bf16*)W, a.out = (bf16*)out, a.addend = (const bf16*)addend, a.bias = bias;
    a.gelu_out = (bf16*)gelu_out;
    a.M = M, a.K = K, a.N = N;
    const int nc = K == 128 ? 128 : 96;
    a.nchunks = N / nc;
    // two blocks per CU: 64 per XCD, shared among the chunks
    int ms = 64 / a.nchunks;
    if (ms < 1) ms = 1;
    const int need = ((M + 15) / 16 + 31) / 32;  // (no more row-walking waves than tiles: 8 XCDs x 4 waves per slot)
    if (ms > need) ms = need > 0 ? need : 1;
    a.mslots = ms;
    const int grid = 8 * a.nchunks * a.mslots;
    static char pname[2][64] = {"", ""};
    char* pn = pname[K == 128 ? 0 : 1];
    if (!pn[0]) snprintf(pn, 64, "gdl::linear_stream_kernel<%d, %d>", K / 32, nc / 16);
    const double bytes = 2.0 * ((double)M * K + (double)M * N * (1 + (addend ? 1 : 0) + (gelu_out ? 1 : 0)) + (double)N * K);
    ProfScope prof(pn, PROF_MFMA, st, 2.0 * (double)M * K * N, true, bytes);
    if (K == 128 && addend)
        hipExtLaunchKernelGGL((linear_stream_kernel<4, 8, true>), dim3(grid), dim3(256), 0, st, prof.e0(), prof.e1(), 0, a);
    else if (K == 128)
        hipExtLaunchKernelGGL((linear_stream_kernel<4, 8, false>), dim3(grid), dim3(256), 0, st, prof.e0(), prof.e1(), 0, a);
    else
        hipExtLaunchKernelGGL((linear_stream_kernel<6, 6, false>), dim3(grid), dim3(256), 0, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("linear_stream_kernel");
    return GDL_OK;
}

}  // namespace gdl
