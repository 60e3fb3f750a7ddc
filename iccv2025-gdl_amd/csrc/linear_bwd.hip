// linear_bwd.hip -- both gradients of an nn.Linear from ONE pass over its output gradient (bf16, gfx950).
//
// Replaces, for the short-K Linears of the Swin branch (/root/reference/models/swin_transformer.py:26-42 Mlp.fc1, :78-157
// WindowAttention.qkv at stage 1: y[M][384] = x[M][128] W^T), the pair
//     dx[M][K] = dy[M][N] . W          (gdl_conv_dgrad, R = S = 1)
//     dW[N][K] = dy^T . x              (gdl_conv_wgrad, R = S = 1)
// Both read dy -- 462 MB at 192 frames -- and both are HBM-bound (4.0-4.5 TB/s, tools/bench_gemm.py): 1 232 MB for the pair,
// of which this kernel moves 770 (dy and x read once, dx written).
//
// Block = 8 waves = the K / 16 column chunks of dx / dW; the block walks over 32-row tiles of (dy, x) with a stride.
//   * the dy tile [32][N] is shared: every thread requests 3 of its 16-byte chunks for the NEXT tile right at the top of an
//     iteration (12 registers), computes the current tile from LDS, and only then parks the chunks in the other buffer -- one
//     barrier per tile, no wait in front of the MFMAs.  Rows are 768 bytes: the 16-byte chunks of a row are XOR-swizzled within
//     groups of eight by the row number, so that the row reads (16 lanes = 16 rows, same column chunk) spread over the banks.
//   * wave c keeps W^T[16 c .. 16 c + 15][N] in registers (a lane's operand = 16 contiguous bytes of one W^T row: 48 registers)
//     and the dW chunk [N][16] in accumulators (24 tiles: 96 registers).
//   * dx^T tile = W^T chunk (A) x dy rows (B: 16-byte LDS row reads): a lane ends with four consecutive columns of one row ->
//     8-byte stores; dW tile += dy^T (A: ds_read_b64_tr_b16 down the tile's columns) x x chunk (B: transpose reads of the
//     wave's own [32][16] x tile).
// The dW chunks of the blocks are partials [block][N][K] float, folded in block order by a second kernel (fixed order:
// run-to-run bit-identical).  Rows beyond M load as zeros and are not stored (raw buffer accesses).
//
// Padded input widths (stage 1 of Swin-T: 96 real columns in rows of 128): the 16-column tiles of dW beyond the real width
// multiply zeros -- they are skipped (KT = 6 of 8 tiles), and with DB the first of them gets an all-ones B operand instead of x
// columns: its accumulators are then the column sums of dy, i.e. the Linear's bias gradient, from MFMAs that were idle anyway
// (the separate column-sum pass over dy was 80 us per qkv at 192 frames).  The fold kernel moves that column to db and zeroes
// the padding columns of dW.
#include "common.h"
#include "ops.h"
#include "prof.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace gdl {

constexpr int LB_ROWS = 32;   // rows of a tile
constexpr int LB_K = 128;     // input width: 8 waves x 16 columns
constexpr int LB_WAVES = LB_K / 16;
constexpr int LB_MAX_BLOCKS = 256;

struct LbArgs {
    const bf16* dy;   // [M][N]
    const bf16* x;    // [M][K]
    const bf16* wT;   // [K][N]
    bf16* dx;         // [M][K]
    float* part;      // [blocks][N][K]
    int M, tiles;
};

typedef __attribute__((ext_vector_type(4))) unsigned int lb_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int lb_u32x2;

__device__ __forceinline__ uint2 lb_tr(const unsigned char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ bf16x8_t lb_frag(uint2 lo, uint2 hi) { return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y)); }
// swizzle key of a row: a bijection of row & 7 under which EIGHT consecutive rows get eight different keys (the 16-byte row reads:
// eight lanes = eight rows, one column chunk) and FOUR consecutive rows get keys with four different upper bit pairs (the transpose
// reads: four rows x two neighbouring chunks -- with key = row & 7 rows 0 / 1 and 2 / 3 met in the same pair of chunks)
__device__ __forceinline__ int lb_key(int row) { return ((row & 3) << 1) | ((row >> 2) & 1); }
// byte offset of 16-byte column chunk `chunk` of row `row` in a swizzled [32][N] tile (row pitch = N * 2 bytes)
template <int NF>
__device__ __forceinline__ int lb_pos(int row, int chunk) {
    return row * (NF * 64) + (((chunk & ~7) | ((chunk & 7) ^ lb_key(row))) << 4);
}

template <int NF, int KT, bool DB>  // N = 32 NF; KT real 16-column tiles of x; DB: tile KT = column sums of dy
__global__ __launch_bounds__(64 * LB_WAVES) void linear_bwd_kernel(LbArgs a) {
    constexpr int N = 32 * NF, CPR = N / 8;          // 16-byte chunks per dy row
    constexpr int TILE = LB_ROWS * N * 2;            // bytes of a dy tile
    constexpr int CHUNKS = LB_ROWS * CPR, PER_THREAD = CHUNKS / (64 * LB_WAVES);
    static_assert(CHUNKS % (64 * LB_WAVES) == 0, "chunks per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    unsigned char* const dyt = lb_smem;                                    // [2][TILE]
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const xt = lb_smem + 2 * TILE;                      // x tiles [2][32 rows][K]: whole 256-byte rows, one 16-byte chunk per thread
    unsigned char* const dxs = xt + 2 * (LB_ROWS * LB_K * 2);           // dx tiles [2][32 rows][K], staged so that they leave as whole rows too
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.M * LB_K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, (int)((size_t)a.M * LB_K * 2), 0x00020000);
    // W^T chunk of this wave: row 16 wave + l16, columns 32 ks + 8 lq .. + 7
    bf16x8_t wreg[NF];
#pragma unroll
    for (int ks = 0; ks < NF; ++ks)
        wreg[ks] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(a.wT + (size_t)(16 * wave + l16) * N + 32 * ks + 8 * lq));
    constexpr int KTA = KT + (DB ? 1 : 0);  // tiles with accumulators
    static_assert(KTA <= 8 && N / 16 == 3 * LB_WAVES, "dW tiles per wave");
    f32x4_t dw[3 * KTA];
#pragma unroll
    for (int nt = 0; nt < 3 * KTA; ++nt) dw[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this thread's chunks of a dy tile: linear chunk p = tid + 512 i -> (row, position in the row); it fetches the source chunk
    // that the swizzle puts there
    // (recomputed at every use instead of kept: nine registers the kernel does not have)
    auto chunk_of = [&](int i, int& row, int& src, int& dst) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int p = t + 64 * LB_WAVES * i;
        row = p / CPR;
        const int pos = p - row * CPR;
        src = (((pos & ~7) | ((pos & 7) ^ lb_key(row))) << 4);
        dst = row * (N * 2) + (pos << 4);
    };
    const int xoff = tid * 16;  // this thread's chunk of an x / dx tile: row tid >> 4, bytes 16 (tid & 15) .. + 15
    // two tiles ahead: tile t + 2 is requested at the top of iteration t, tile t + 1 (requested an iteration ago) is parked at
    // its end -- a request has a whole iteration (~1 us of MFMAs and LDS reads) plus the other waves' barrier skew to land (with
    // one tile of distance every iteration ended in a full memory round trip: 4.3 us per tile, 2.4 TB/s)
    lb_u32x4 nx[2][PER_THREAD], nxx[2];
    auto request = [&](int tile, auto SET) {
        constexpr int set = decltype(SET)::value;  // (compile-time: a run-time register-set index would put the sets in scratch)
        const int m0 = tile * LB_ROWS;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int row, src, dst;
            chunk_of(i, row, src, dst);
            nx[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, (m0 + row) * (N * 2) + src, 0, 0);
        }
        nxx[set] = __builtin_amdgcn_raw_buffer_load_b128(rx, m0 * (LB_K * 2) + xoff, 0, 0);
    };
    auto park = [&](int buf, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int row, src, dst;
            chunk_of(i, row, src, dst);
            *(lb_u32x4*)(dyt + buf * TILE + dst) = nx[set][i];
        }
        *(lb_u32x4*)(xt + buf * (LB_ROWS * LB_K * 2) + (xoff ^ (((tid >> 4) & 3) << 5))) = nxx[set];  // (x tile: chunk ^ 2 (row & 3))
    };
    int tile = blockIdx.x;
    // (requests beyond the last tile are out of range: zeros, no memory access -- no branches around them)
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    request(tile, S0{});
    park(0, S0{});
    request(tile + gridDim.x, S0{});
    __syncthreads();
    int buf = 0;
    for (; tile < a.tiles; tile += gridDim.x) {
#ifndef LB_SKIP_LOAD
        request(tile + 2 * gridDim.x, S1{});
#endif
        const unsigned char* T = dyt + buf * TILE;
        const int m0 = tile * LB_ROWS;
        if (tile != (int)blockIdx.x)  // the previous tile's dx rows: staged by all waves, one 16-byte chunk per thread
            __builtin_amdgcn_raw_buffer_store_b128(*(const lb_u32x4*)(dxs + (buf ^ 1) * (LB_ROWS * LB_K * 2) + xoff), rdx,
                                                   (m0 - (int)gridDim.x * LB_ROWS) * (LB_K * 2) + xoff, 0, 0);
        // LDS addresses: the swizzle XORs the low three bits of a row's 16-byte chunk number with the row number, so a lane's
        // reads of one kind differ only by compile-time offsets once the XOR'd part is in the base (per-read address arithmetic
        // was a third of the kernel's instructions)
        // ---- dx^T[16 columns][16 rows] per row tile: A = W^T chunk, B = dy rows (chunk 4 ks + lq of row 16 rt + l16)
#ifndef LB_SKIP_DX
        const int e0 = ((lq ^ lb_key(l16)) << 4), e1 = (((lq ^ lb_key(l16)) ^ 4) << 4);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int row = 16 * rt + l16;
            const unsigned char* pe = T + row * (N * 2) + e0;
            const unsigned char* po = T + row * (N * 2) + e1;
#pragma unroll
            for (int ks = 0; ks < NF; ++ks) {
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *(const uint4*)((ks & 1 ? po : pe) + 128 * (ks >> 1)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], b, acc, 0, 0, 0);
            }
            // lane (l16 = row, lq): columns 16 wave + 4 lq .. + 3
            *(uint2*)(dxs + buf * (LB_ROWS * LB_K * 2) + row * (LB_K * 2) + wave * 32 + lq * 8) = make_uint2(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]));
        }
#endif
#ifndef LB_SKIP_DW
        // ---- dW += dy^T . x: wave w owns the rows 48 w .. 48 w + 47 of dW (three 16-row tiles) for ALL 128 columns -- its A operands
        // are three transpose reads of the dy tile's columns instead of all 24 (every wave reading the whole tile twice, once by rows
        // for dx and once by columns for dW, made the kernel LDS-bound: 284 us with the global loads removed), its B operands the
        // eight 16-column chunks of the shared x tile
        const int rho = l16 >> 2, sub = (l16 & 3) >> 1, boff = ((l16 & 3) & 1) * 8;
        const unsigned char* r_lo = T + (8 * lq + rho) * (N * 2) + boff;   // rows 8 lq + rho (key 2 rho) and + 4 (key 2 rho + 1)
        const unsigned char* r_hi = r_lo + 4 * (N * 2);
        bf16x8_t af[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int nt = 3 * wave + i;  // (wave-uniform)
            const int c = 2 * nt + sub;
            const uint2 lo = lb_tr(r_lo + (((c & ~7) | ((c & 7) ^ (2 * rho))) << 4)), hi = lb_tr(r_hi + (((c & ~7) | ((c & 7) ^ (2 * rho + 1))) << 4));
            af[i] = lb_frag(lo, hi);
        }
        const unsigned char* xr = xt + buf * (LB_ROWS * LB_K * 2) + (8 * lq + rho) * (LB_K * 2) + boff;  // x rows 8 lq + rho (and + 4: same key)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int xo = ((((2 * kt) & ~7) | (((2 * kt + sub) & 7) ^ (2 * rho))) << 4);
            const bf16x8_t xb = lb_frag(lb_tr(xr + xo), lb_tr(xr + 4 * (LB_K * 2) + xo));
#pragma unroll
            for (int i = 0; i < 3; ++i) dw[3 * kt + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], xb, dw[3 * kt + i], 0, 0, 0);
        }
        if constexpr (DB) {  // sum over the tile's 32 rows of dy[m][n] * 1
            const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
#pragma unroll
            for (int i = 0; i < 3; ++i) dw[3 * KT + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, dw[3 * KT + i], 0, 0, 0);
        }
#endif
        park(buf ^ 1, S0{});
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) nx[0][i] = nx[1][i];
        nxx[0] = nxx[1];
    }
    // the last tile's dx rows (staged in the buffer the loop left)
    if ((int)blockIdx.x < a.tiles)
        __builtin_amdgcn_raw_buffer_store_b128(*(const lb_u32x4*)(dxs + (buf ^ 1) * (LB_ROWS * LB_K * 2) + xoff), rdx,
                                               (tile - (int)gridDim.x) * LB_ROWS * (LB_K * 2) + xoff, 0, 0);
    // dW rows 48 wave .. + 47 -> partial [block][N][K]: dw[3 kt + i][r] of lane (l16, lq) = (row 16 (3 wave + i) + 4 lq + r, column 16 kt + l16)
    float* pp = a.part + (size_t)blockIdx.x * N * LB_K + (size_t)(48 * wave + 4 * lq) * LB_K + l16;
#pragma unroll
    for (int kt = 0; kt < KTA; ++kt)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[(size_t)(16 * i + r) * LB_K + 16 * kt] = dw[3 * kt + i][r];
}

// dW[n][k] = sum over the blocks' partials in a fixed order; columns from kcols on are padding (not written by the blocks): zero,
// except that with db the partials' column kcols holds the column sums of dy -> db[n].
// A block = 16 partial groups x 16 float4 columns: thread (g, v) sums the partials g, g + 16, ... of four outputs (all its loads in
// flight at once), the 16 groups fold through LDS in group order.  (One thread per output walking all 256 partials was a chain of
// 256 dependent-latency loads: 78 us per launch for 50 MB.)
__global__ __launch_bounds__(256) void linear_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                                int nblk, int total, int kcols) {
    __shared__ float4 red[16][16];
    const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i4 = (blockIdx.x * 16 + v) * 4;  // first of this thread's four outputs (total is a multiple of 64)
    const int col = i4 % LB_K;
    const bool live = col < kcols || (db && col == kcols);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        constexpr int UN = 8;
        for (int b0 = g; b0 < nblk; b0 += 16 * UN) {
            float4 q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int b = b0 + 16 * u;
                q[u] = b < nblk ? *(const float4*)(part + (size_t)b * total + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) s.x += q[u].x, s.y += q[u].y, s.z += q[u].z, s.w += q[u].w;
        }
    }
    red[g][v] = s;
    __syncthreads();
    if (g != 0) return;
    float4 t = red[0][v];
#pragma unroll
    for (int k = 1; k < 16; ++k) t.x += red[k][v].x, t.y += red[k][v].y, t.z += red[k][v].z, t.w += red[k][v].w;
    if (col == kcols) {  // (only reached with live sums when db is given)
        if (db) db[i4 / LB_K] = t.x;
        t = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *(float4*)(dw + i4) = t;
}

static int lb_blocks(int M) {
    const int tiles = (M + LB_ROWS - 1) / LB_ROWS;
    return tiles < LB_MAX_BLOCKS ? tiles : LB_MAX_BLOCKS;
}
bool linear_bwd_ok(int dtype, size_t M, int K, int N) {
    static int on = -1;
    if (on < 0) {
        const char* e = tune_env("GDL_LINEAR_BWD");  // tuning aid: 0 = data gradient and weight gradient as two GEMMs
        on = e ? atoi(e) : 1;
    }
    return on && dtype == GDL_BF16 && K == LB_K && N == 384 && M >= 16384 && M * (size_t)N * 2 < (1ull << 31);
}
size_t linear_bwd_ws_bytes(size_t M, int K, int N) { return (size_t)lb_blocks((int)M) * N * K * sizeof(float); }
int linear_bwd(const void* dy, const void* x, const void* wT, void* dx, float* dw, float* db, void* ws, size_t ws_bytes, size_t M, int K,
               int Kreal, int N, hipStream_t st) {
    GDL_REQUIRE(dy && x && wT && dx && dw && ws, "linear_bwd: null pointer");
    GDL_REQUIRE(linear_bwd_ok(GDL_BF16, M, K, N), "linear_bwd: unsupported shape M=%zu K=%d N=%d", M, K, N);
    GDL_REQUIRE(Kreal >= 1 && Kreal <= K, "linear_bwd: %d real columns of %d", Kreal, K);
    GDL_REQUIRE(!db || Kreal <= 96, "linear_bwd: the bias gradient rides in a padding tile (needs at most 96 real columns, got %d)", Kreal);
    GDL_REQUIRE(ws_bytes >= linear_bwd_ws_bytes(M, K, N), "linear_bwd: workspace %zu < %zu bytes", ws_bytes, linear_bwd_ws_bytes(M, K, N));
    constexpr int NF = 12;
    const size_t lds = 2 * (size_t)LB_ROWS * 32 * NF * 2 + 4 * (size_t)LB_ROWS * LB_K * 2;
    const int form = Kreal > 96 ? 0 : (db ? 2 : 1);
    const void* fn = form == 0 ? (const void*)linear_bwd_kernel<NF, 8, false>
                               : (form == 1 ? (const void*)linear_bwd_kernel<NF, 6, false> : (const void*)linear_bwd_kernel<NF, 6, true>);
    static DevOnce attr[3];
    if (!attr[form]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(linear_bwd)");
        attr[form] = true;
    }
    LbArgs a{(const bf16*)dy, (const bf16*)x, (const bf16*)wT, (bf16*)dx, (float*)ws, (int)M, (int)((M + LB_ROWS - 1) / LB_ROWS)};
    const int nblk = lb_blocks((int)M);
    {
        ProfScope prof("gdl::linear_bwd_kernel", PROF_MFMA, st, 4.0 * (double)M * K * N, true, 2.0 * (double)M * (N + 2 * K));
        if (form == 0)
            hipExtLaunchKernelGGL((linear_bwd_kernel<NF, 8, false>), dim3(nblk), dim3(64 * LB_WAVES), lds, st, prof.e0(), prof.e1(), 0, a);
        else if (form == 1)
            hipExtLaunchKernelGGL((linear_bwd_kernel<NF, 6, false>), dim3(nblk), dim3(64 * LB_WAVES), lds, st, prof.e0(), prof.e1(), 0, a);
        else
            hipExtLaunchKernelGGL((linear_bwd_kernel<NF, 6, true>), dim3(nblk), dim3(64 * LB_WAVES), lds, st, prof.e0(), prof.e1(), 0, a);
        GDL_CHECK_LAUNCH("linear_bwd_kernel");
    }
    hipLaunchKernelGGL(linear_bwd_reduce_kernel, dim3(N * K / 64), dim3(256), 0, st, (const float*)ws, dw, db, nblk, N * K,
                       form == 0 ? LB_K : 96);
    GDL_CHECK_LAUNCH("linear_bwd_reduce_kernel");
    return GDL_OK;
}

}  // namespace gdl
