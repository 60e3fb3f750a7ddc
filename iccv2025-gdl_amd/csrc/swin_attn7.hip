// swin_attn7.hip -- (S)W-MSA for 7 x 7 windows, bf16 storage, on the matrix cores: forward and backward.
//
// Replaces WindowAttention.forward with the cyclic shift, the shift mask and the relative-position bias
// (/root/reference/models/swin_transformer.py:124-157, :222-240, :256-285) for window_size 7 -- every stage of Swin-T / -S / -B.
// Other window sizes and the float32 mode stay on the kernels of swin.hip.
//
// Why a second form: the round-3 kernels (one wave per (window, head), window size a run-time value) execute ~4 400 (forward)
// and ~6 000 (backward) instructions per window for 32 / 80 MFMAs -- index arithmetic with run-time divisions, per-element
// bias lookups and padding selects, 16-lane DPP reductions per query row, 2-byte LDS scatters for every transposed operand,
// a 49 x 49 scalar walk for d(table) -- and ran at 1.2-2.4 TB/s of their 4 / 7 tensor passes (instruction-issue bound at two
// waves per SIMD; tools/bench_swin_attn.py).  Here, per window and wave:
//   * scores TRANSPOSED on the accumulators (lane = one query of the strip, 16 of its 64 keys): row maximum / sum / sum(P dP)
//     are 15 in-lane operations + two cross-row steps (v_permlane32_swap, v_permlane16_swap), once per 16-query strip;
//   * the relative-position bias, pre-multiplied by log2(e) and with -3e38 on the 15 padding keys, is a [4 strips][64 lanes]
//     [16] float tile in LDS built once per block (the block's four waves work on windows of ONE head): four 16-byte reads per
//     strip, one FMA per element (scale log2(e) folded in), exp2 directly; padding needs no selects anywhere -- padding
//     queries / keys are zero rows (out-of-range buffer loads return 0), their probabilities never reach a stored value;
//   * the shift mask only in windows of the last window row / column (wave-uniform branch), from per-lane bit masks;
//   * P (then dS) is stored ONCE, row-major, and read as the MFMA operand either directly (k = key) or through the LDS
//     transpose read ds_read_b64_tr_b16 (k = query); V / dO / K / Q rows go to LDS row-major with four 16-byte stores per lane
//     and are read transposed the same way -- no 2-byte scatters;
//   * every product is computed transposed with the channel rows permuted (8 q + r | 8 q + 4 + r), so a lane owns eight
//     consecutive channels of one token and loads / stores 16 bytes at the SAME buffer offset for q, k, v, dout, dq, dk, dv;
//   * d(table): the window sum of dS is accumulated on the matrix cores, D += dS . I (sixteen extra MFMAs per window, no VALU
//     work, 64 accumulator registers), folded into the (2 ws - 1)^2 entries once per block in a fixed order.
// All loads and stores are raw buffer accesses relative to the image (32-bit offsets, one multiply-add per token).
// Run-to-run bit-identical: no atomics, every sum in a fixed order.
#include "common.h"
#include "ops.h"
#include "prof.h"

namespace gdl {
#ifdef GDL_TIMING
extern unsigned long long* g_timing_buf;
#endif

constexpr int A7_T = 49;     // tokens per window
constexpr int A7_TP = 144;   // bytes per row of the 64 x 64 bf16 tile (pitch 36 banks: rows 4 banks apart)
constexpr int A7_XP = 96;    // bytes per row of a [token][32 channels] bf16 operand (4 rows x 32 B of a transpose read: 4 x 8 distinct banks)
constexpr float A7_LOG2E = 1.4426950408889634f;
constexpr float A7_C = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 log2(e)
constexpr float A7_SCALE = 0.17677669529663687f;
constexpr float A7_MASK = -100.f * 1.4426950408889634f;
constexpr unsigned A7_OOB = 0x80000000u;

struct A7Geom {
    int H, W, nh, ld, shift;
    int wpr, wpc;      // windows per row / column of an image
    int total;         // n_img * wpr * wpc windows per head
    int chunk;         // windows per block (the block's four waves take them round-robin)
    int xcd;
#ifdef GDL_TIMING
    unsigned long long* dbg;  // [block][wave][8]: clocks per phase summed over the wave's windows, [6] = windows (tools/probe_attn7.py)
#endif
};
#ifdef GDL_TIMING
#define A7_STAMP(k)                                         \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        tacc[k] += t_ - tprev;                              \
        tprev = t_;                                         \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
#define A7_STAMP0()                                  \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        tprev = __builtin_amdgcn_s_memtime();        \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
#else
#define A7_STAMP(k)
#define A7_STAMP0()
#endif
struct A7Wave {
    unsigned char tile[64 * A7_TP];  // P, then dS: [query][key]
    unsigned char opnd[64 * A7_XP];  // V (forward) / dO, K, Q in turn (backward): [token][channel]
};
struct A7Lds {
    float bias[4 * 64 * 16];  // [strip][lane][key tile * 4 + r]; the backward's d(table) fold reuses it as [strip][key tile][lane][r]
    A7Wave w[4];
};

__device__ __forceinline__ unsigned a7_block(unsigned b, unsigned nb, int on) {  // see sw_xcd_block (swin.hip)
    if (!on) return b;
    const unsigned q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}
__device__ __forceinline__ void a7_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// within each 16-lane group the lanes' 8-byte reads form a 4 x 16 block of 16-bit elements (lane i supplies row i >> 2, columns
// 4 (i & 3) .. + 3); lane i receives column i (conv_wgrad.hip)
__device__ __forceinline__ uint2 a7_tr(const unsigned char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ bf16x8_t a7_frag(uint2 lo, uint2 hi) { return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y)); }
// all-reduce over the four lanes that share a query (lane ^ 16, lane ^ 32): the same operands in the same order in all four
__device__ __forceinline__ float a7_sum4(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float a7_max4(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
typedef __attribute__((ext_vector_type(4))) unsigned int a7_u32x4;
__device__ __forceinline__ bf16x8_t a7_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void a7_store(__amdgpu_buffer_rsrc_t r, uint4 v, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(a7_u32x4, v), r, (int)voff, soff, 0);
}

// bias tile of head h (once per block; 16 entries per thread: its four queries x four keys)
__device__ __forceinline__ void a7_bias_build(float* bias, const float* __restrict__ table, int nh, int h) {
    const int tid = threadIdx.x, l16 = tid >> 4, jt = (tid >> 2) & 3, r = tid & 3;
    int ri[4], ci[4], rj[4], cj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = 16 * a + l16, j = 16 * jt + 4 * a + r;  // a = strip for i, = lq for j
        ri[a] = i / 7, ci[a] = i % 7, rj[a] = j / 7, cj[a] = j % 7;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int lq = 0; lq < 4; ++lq) {
            const int i = 16 * it + l16, j = 16 * jt + 4 * lq + r;
            float v;
            if (j >= A7_T)
                v = -3.0e38f;
            else if (i >= A7_T)
                v = 0.f;
            else
                v = table[((ri[it] - rj[lq] + 6) * 13 + (ci[it] - cj[lq] + 6)) * nh + h] * A7_LOG2E;
            bias[(it * 64 + lq * 16 + l16) * 16 + jt * 4 + r] = v;
        }
}

// per-lane constants of the 4 x 4 slots a lane touches, packed (the backward sits at the register limit)
struct A7Lane {
    unsigned rs, cs;      // byte t: slot 16 t + l16 -> (row, column) in the window
    unsigned rowd[2], cold[2];  // half-word (it & 1) of word it >> 1: bit (4 jt + r) set where query 16 it + l16 and key 16 jt + 4 lq + r
                          // lie on different sides of the shift boundary (rows / columns): the mask of a window of the last
                          // window row / column
};
__device__ __forceinline__ void a7_lane_init(A7Lane& c, int l16, int lq, int shift) {
    unsigned ai = 0, bi = 0, aj = 0, bj = 0;
    c.rs = c.cs = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int s = 16 * t + l16, rs = s / 7, cs = s % 7;
        c.rs |= (unsigned)rs << (8 * t);
        c.cs |= (unsigned)cs << (8 * t);
        ai |= (unsigned)(rs >= 7 - shift) << t;
        bi |= (unsigned)(cs >= 7 - shift) << t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * lq + r;
            aj |= (unsigned)(j / 7 >= 7 - shift) << (4 * t + r);
            bj |= (unsigned)(j % 7 >= 7 - shift) << (4 * t + r);
        }
    }
    c.rowd[0] = c.rowd[1] = c.cold[0] = c.cold[1] = 0;
    if (shift) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            c.rowd[t >> 1] |= (aj ^ (((ai >> t) & 1u) ? 0xffffu : 0u)) << (16 * (t & 1));
            c.cold[t >> 1] |= (bj ^ (((bi >> t) & 1u) ? 0xffffu : 0u)) << (16 * (t & 1));
        }
    }
}
// byte offsets (within the image) of the lane's four token rows, for the [q | k | v] rows and for the ld-wide rows
__device__ __forceinline__ void a7_offsets(const A7Geom& g, const A7Lane& c, int wy, int wx, int l16, unsigned cl, unsigned (&qoff)[4],
                                           unsigned (&ooff)[4]) {
    const int r0 = wy * 7 + g.shift, c0 = wx * 7 + g.shift;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int R = r0 + (int)((c.rs >> (8 * t)) & 0xffu), C = c0 + (int)((c.cs >> (8 * t)) & 0xffu);
        R -= R >= g.H ? g.H : 0;
        C -= C >= g.W ? g.W : 0;
        const unsigned tok = (unsigned)(R * g.W + C);
        qoff[t] = tok * (unsigned)(6 * g.ld) + cl;
        ooff[t] = tok * (unsigned)(2 * g.ld) + cl;
    }
    if (l16 != 0) qoff[3] = ooff[3] = A7_OOB;  // slots 49 .. 63
}

// softmax numerators of one strip on the transposed accumulators: s[jt][r] = score of (query 16 it + l16, key 16 jt + 4 lq + r) in,
// exp2(a - max) out; returns 1 / sum
template <bool MASK>
__device__ __forceinline__ float a7_exp_strip(f32x4_t (&s)[4], const float* brow, unsigned mb) {
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const float4 b = *(const float4*)(brow + 4 * jt);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = fmaf(s[jt][r], A7_C, bb[r]);
            if (MASK) a = fmaf((float)((mb >> (4 * jt + r)) & 1u), A7_MASK, a);
            s[jt][r] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = a7_max4(mx);
    float den = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[jt][r] - mx);
            s[jt][r] = e;
            den += e;
        }
    den = a7_sum4(den);
    return __builtin_amdgcn_rcpf(den);
}
__device__ __forceinline__ uint2 a7_pack4(f32x4_t v) { return make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }

// transposed operand fragments of the [token][channel] rows in `opnd`: xa[half][ks], MFMA row m = l16 <-> channel
// 8 (l16 >> 2) + 4 half + (l16 & 3), k = tokens 32 ks + 8 lq .. + 7
__device__ __forceinline__ void a7_opnd_frags(const unsigned char* opnd, int l16, int lq, bf16x8_t (&xa)[2][2]) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* p = opnd + (32 * ks + 8 * lq + (l16 >> 2)) * A7_XP + (8 * (l16 & 3) + 4 * half) * 2;
            xa[half][ks] = a7_frag(a7_tr(p), a7_tr(p + 4 * A7_XP));
        }
}
// tile fragment for output slot tile t, K-step ks: n = slot 16 t + l16, k = 32 ks + 8 lq .. + 7 -- along the tile's rows (TR = false:
// out[slot] = sum_k tile[slot][k] x[k]) or down its columns (TR = true: out[slot] = sum_k tile[k][slot] x[k])
template <bool TR>
__device__ __forceinline__ bf16x8_t a7_tile_frag(const unsigned char* tile, int t, int ks, int l16, int lq) {
    if (!TR) return __builtin_bit_cast(bf16x8_t, *(const uint4*)(tile + (16 * t + l16) * A7_TP + (32 * ks + 8 * lq) * 2));
    const unsigned char* p = tile + (32 * ks + 8 * lq + (l16 >> 2)) * A7_TP + (16 * t + 4 * (l16 & 3)) * 2;
    return a7_frag(a7_tr(p), a7_tr(p + 4 * A7_TP));
}
__device__ __forceinline__ uint4 a7_pack8(f32x4_t a, f32x4_t b, float mul) {
    return make_uint4(pack2bf(a[0] * mul, a[1] * mul), pack2bf(a[2] * mul, a[3] * mul), pack2bf(b[0] * mul, b[1] * mul),
                      pack2bf(b[2] * mul, b[3] * mul));
}
// out[slot][32 channels] (segment offset soff of the rows behind `rs`) = mul * product; a lane stores channels 8 lq .. + 7 of slot
// 16 t + l16 -- the offset its q / k / v / dout fragment came from
template <bool TR>
__device__ __forceinline__ void a7_product(const unsigned char* tile, const unsigned char* opnd, __amdgpu_buffer_rsrc_t rs,
                                           const unsigned (&off)[4], int soff, float mul, int l16, int lq) {
    bf16x8_t xa[2][2];
    a7_opnd_frags(opnd, l16, lq, xa);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4_t o0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, o1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t b = a7_tile_frag<TR>(tile, t, ks, l16, lq);
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[0][ks], b, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[1][ks], b, o1, 0, 0, 0);
        }
        a7_store(rs, a7_pack8(o0, o1, mul), off[t], soff);
    }
}
// the lane's four fragments (token rows 16 t + l16, channels 8 lq .. + 7) -> row-major operand tile
__device__ __forceinline__ void a7_rows_to_lds(unsigned char* opnd, const bf16x8_t (&f)[4], int l16, int lq) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *(uint4*)(opnd + (16 * t + l16) * A7_XP + lq * 16) = __builtin_bit_cast(uint4, f[t]);
}
// zero the padding channels [32 nh, ld) of the window's token rows (segment offset soff): lane = slot
__device__ __forceinline__ void a7_zero_pad(const A7Geom& g, __amdgpu_buffer_rsrc_t rs, int rowbytes, int wy, int wx, int lane, int soff) {
    const int rl = lane / 7, cl7 = lane % 7;
    int R = wy * 7 + g.shift + rl, C = wx * 7 + g.shift + cl7;
    R -= R >= g.H ? g.H : 0;
    C -= C >= g.W ? g.W : 0;
    const unsigned off = lane < A7_T ? (unsigned)(R * g.W + C) * (unsigned)rowbytes + (unsigned)(g.nh * 64) : A7_OOB;
    for (int c = 0; c < (g.ld - 32 * g.nh) / 8; ++c) a7_store(rs, make_uint4(0u, 0u, 0u, 0u), off + 16 * c, soff);
}
// the wave's next window: + 4 in the order (image, window row, window column)
__device__ __forceinline__ void a7_advance(const A7Geom& g, int& n, int& wy, int& wx) {
    wx += 4;
    while (wx >= g.wpr) wx -= g.wpr, ++wy;
    while (wy >= g.wpc) wy -= g.wpc, ++n;
}

// The forward needs no probability tile: O = P V is computed transposed, O^T = V^T P^T, whose second operand has n = query and
// k = key -- and a lane of the score accumulators already holds one query (n = l16) and eight keys of every 32-key K-step: keys
// 32 ks + 4 lq + (0 .. 3) from tile 2 ks and 32 ks + 16 + 4 lq + (0 .. 3) from tile 2 ks + 1.  That is a permutation of the K-step's
// keys among the (lq, element) slots, and a dot product does not care as long as the other operand uses the same one: the V^T
// fragments take their two transpose reads from token rows 32 ks + 4 lq and 32 ks + 16 + 4 lq.  So P goes from the softmax registers
// straight into the MFMA -- no LDS round trip, one wave barrier less per window, 9 KB of LDS less per wave: four blocks per CU
// instead of two.
struct A7LdsF {
    float bias[4 * 64 * 16];
    unsigned char opnd[4][64 * A7_XP];  // per wave: the V rows of its window
};
__device__ __forceinline__ void a7_v_frags(const unsigned char* opnd, int l16, int lq, bf16x8_t (&xa)[2][2]) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* p = opnd + (32 * ks + 4 * lq + (l16 >> 2)) * A7_XP + (8 * (l16 & 3) + 4 * half) * 2;
            xa[half][ks] = a7_frag(a7_tr(p), a7_tr(p + 16 * A7_XP));
        }
}
// One window of the forward.  (A software-pipelined form -- the next window's fragments requested before this window's strips,
// 166 registers -- measured the same or 4 % slower at every stage shape: the kernel is not waiting for its loads.)
template <bool MASK>
__device__ __forceinline__ void a7_fwd_window(const A7Geom& g, const float* bias, unsigned char* opnd, const A7Lane& c,
                                              __amdgpu_buffer_rsrc_t rq, __amdgpu_buffer_rsrc_t ro, const unsigned (&qoff)[4],
                                              const unsigned (&ooff)[4], unsigned lastrow, unsigned lastcol, int lane, int l16, int lq) {
    bf16x8_t qf[4], kf[4];
    bf16x8_t xa[2][2];
    {
        bf16x8_t vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            qf[t] = a7_load(rq, qoff[t], 0);
            kf[t] = a7_load(rq, qoff[t], 2 * g.ld);
            vf[t] = a7_load(rq, qoff[t], 4 * g.ld);
        }
        a7_rows_to_lds(opnd, vf, l16, lq);
    }
    a7_wave_sync();
    a7_v_frags(opnd, l16, lq, xa);
    a7_wave_sync();  // (the next window's V rows may land once these reads are done)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        f32x4_t s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], qf[it], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const unsigned mb = MASK ? ((c.rowd[it >> 1] & lastrow) | (c.cold[it >> 1] & lastcol)) >> (16 * (it & 1)) : 0u;
        const float inv = a7_exp_strip<MASK>(s, &bias[(it * 64 + lane) * 16], mb);
        f32x4_t o0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, o1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, a7_pack8(s[2 * ks], s[2 * ks + 1], inv));  // P of the strip's queries
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[0][ks], b, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[1][ks], b, o1, 0, 0, 0);
        }
        a7_store(ro, a7_pack8(o0, o1, 1.f), ooff[it], 0);
    }
}

__global__ __launch_bounds__(256, 4) void swin_attn7_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ table,
                                                                bf16* __restrict__ out, A7Geom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char a7_smem[];
    A7LdsF& S = *(A7LdsF*)a7_smem;
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned bid = a7_block(blockIdx.x, gridDim.x, g.xcd);
    const int h = (int)(bid % (unsigned)g.nh), chunk = (int)(bid / (unsigned)g.nh);
    a7_bias_build(S.bias, table, g.nh, h);
    A7Lane c;
    a7_lane_init(c, l16, lq, g.shift);
    __syncthreads();
    unsigned char* opnd = S.opnd[wave];
    const int nwin = g.wpr * g.wpc, L = g.H * g.W;
    const int gw_end = min(g.total, (chunk + 1) * g.chunk);
    int gw = chunk * g.chunk + wave;
    int n = gw / nwin, wy = (gw - n * nwin) / g.wpr, wx = gw - n * nwin - wy * g.wpr;
    const unsigned cl = (unsigned)(h * 64 + lq * 16);
    for (; gw < gw_end; gw += 4) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (size_t)n * L * 3 * g.ld), 0, L * 6 * g.ld, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)n * L * g.ld), 0, L * 2 * g.ld, 0x00020000);
        unsigned qoff[4], ooff[4];
        a7_offsets(g, c, wy, wx, l16, cl, qoff, ooff);
        const bool lr = g.shift && wy == g.wpc - 1, lc = g.shift && wx == g.wpr - 1;
        if (lr || lc)
            a7_fwd_window<true>(g, S.bias, opnd, c, rq, ro, qoff, ooff, lr ? 0xffffffffu : 0u, lc ? 0xffffffffu : 0u, lane, l16, lq);
        else
            a7_fwd_window<false>(g, S.bias, opnd, c, rq, ro, qoff, ooff, 0u, 0u, lane, l16, lq);
        if (h == 0 && g.ld > 32 * g.nh) a7_zero_pad(g, ro, 2 * g.ld, wy, wx, lane, 0);
        a7_advance(g, n, wy, wx);
    }
}

template <bool MASK>
__device__ __forceinline__ void a7_bwd_window(const A7Geom& g, A7Lds& S, A7Wave& Wv, const A7Lane& c, __amdgpu_buffer_rsrc_t rq,
                                              __amdgpu_buffer_rsrc_t ro, __amdgpu_buffer_rsrc_t rd, const unsigned (&qoff)[4],
                                              const unsigned (&ooff)[4], unsigned lastrow, unsigned lastcol, f32x4_t (&D)[4][3],
                                              float (&dcol)[4], int lane, int l16, int lq
#ifdef GDL_TIMING
                                              , unsigned long long (&tacc)[6]
#endif
) {
#ifdef GDL_TIMING
    unsigned long long tprev;
#endif
    A7_STAMP0();
    bf16x8_t qf[4], kf[4];
    uint2 pkd[4][4];  // dS of (strip it, key tile jt): keys 16 jt + 4 lq .. + 3 of query 16 it + l16
    {
        bf16x8_t vf[4];
        {
            bf16x8_t of[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                qf[t] = a7_load(rq, qoff[t], 0);
                kf[t] = a7_load(rq, qoff[t], 2 * g.ld);
                vf[t] = a7_load(rq, qoff[t], 4 * g.ld);
                of[t] = a7_load(ro, ooff[t], 0);
            }
            // the dO rows go to the operand tile at once (dV = P^T dO reads them transposed after the strips); a strip's dP
            // takes its dO fragment back from there -- 16 registers less across the strip loop
            a7_rows_to_lds(Wv.opnd, of, l16, lq);
        }
        a7_wave_sync();
        A7_STAMP(0);  // offsets, loads issued and landed, dO rows in LDS
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4_t s[4], dp[4];
            const bf16x8_t ofr = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Wv.opnd + (16 * it + l16) * A7_XP + lq * 16));
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], qf[it], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt], ofr, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);  // dP = dO V^T
            }
            const unsigned mb = MASK ? ((c.rowd[it >> 1] & lastrow) | (c.cold[it >> 1] & lastcol)) >> (16 * (it & 1)) : 0u;
            const float inv = a7_exp_strip<MASK>(s, &S.bias[(it * 64 + lane) * 16], mb);
            float pd = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[jt][r] *= inv;  // P
                    pd = fmaf(s[jt][r], dp[jt][r], pd);
                }
            pd = a7_sum4(pd);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                *(uint2*)(Wv.tile + (16 * it + l16) * A7_TP + (16 * jt + 4 * lq) * 2) = a7_pack4(s[jt]);
                f32x4_t ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[r] = s[jt][r] * (dp[jt][r] - pd);
                pkd[it][jt] = a7_pack4(ds);
                if (jt == 3) dcol[it] += ds[0];  // key 48 (lanes lq = 0; the other lanes' sums are never read)
            }
        }
    }
    A7_STAMP(1);  // the four strips
    // dV = P^T dO
    a7_wave_sync();
    a7_product<true>(Wv.tile, Wv.opnd, rd, qoff, 4 * g.ld, 1.f, l16, lq);
    a7_wave_sync();
    A7_STAMP(2);
    // dQ = scale dS K, and the window's dS added to the block's D on the matrix cores (D += dS . I)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) *(uint2*)(Wv.tile + (16 * it + l16) * A7_TP + (16 * jt + 4 * lq) * 2) = pkd[it][jt];
    a7_rows_to_lds(Wv.opnd, kf, l16, lq);
    a7_wave_sync();
    {
        // identity fragments, made here for every window (eight registers that would otherwise live across the strip loop):
        // I[k][n] of the key tile of parity p within a 32-key K-step: k = 8 lq + e, n = l16 -> 1 where 8 lq + e = 16 p + l16
        int e0 = l16 - 8 * lq;
        asm volatile("" : "+v"(e0));
        bf16x8_t idf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = 16 * p + e0;
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = (e == 2 * q ? 0x3f80u : 0u) | (e == 2 * q + 1 ? 0x3f800000u : 0u);
            idf[p] = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
        }
        bf16x8_t xa[2][2];
        a7_opnd_frags(Wv.opnd, l16, lq, xa);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4_t o0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, o1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8_t b = a7_tile_frag<false>(Wv.tile, t, ks, l16, lq);  // dS rows 16 t + l16, keys 32 ks + 8 lq .. + 7
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[0][ks], b, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[1][ks], b, o1, 0, 0, 0);
                D[t][2 * ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, idf[0], D[t][2 * ks], 0, 0, 0);
                if (ks == 0) D[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, idf[1], D[t][1], 0, 0, 0);  // (key tile 3 = key 48 alone: dcol)
            }
            a7_store(rd, a7_pack8(o0, o1, A7_SCALE), qoff[t], 0);
        }
    }
    a7_wave_sync();
    A7_STAMP(3);
    // dK = scale dS^T Q
    a7_rows_to_lds(Wv.opnd, qf, l16, lq);
    a7_wave_sync();
    a7_product<true>(Wv.tile, Wv.opnd, rd, qoff, 2 * g.ld, A7_SCALE, l16, lq);
    a7_wave_sync();
    A7_STAMP(4);
}

__global__ __launch_bounds__(256, 2) void swin_attn7_bwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ table,
                                                                const bf16* __restrict__ dout, bf16* __restrict__ dqkv,
                                                                float* __restrict__ tpart, A7Geom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char a7_smem[];
    A7Lds& S = *(A7Lds*)a7_smem;
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned bid = a7_block(blockIdx.x, gridDim.x, g.xcd);
    const int h = (int)(bid % (unsigned)g.nh), chunk = (int)(bid / (unsigned)g.nh);
    a7_bias_build(S.bias, table, g.nh, h);
    A7Lane c;
    a7_lane_init(c, l16, lq, g.shift);
    // window sums of dS: D[it][jt][r] of lane (l16, lq) = (query 16 it + 4 lq + r, key 16 jt + l16) for the key tiles 0 .. 2 (on the
    // matrix cores); dcol[it] of the lanes lq = 0 = (query 16 it + l16, key 48), added on the VALU from the unrounded dS
    f32x4_t D[4][3];
    float dcol[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) D[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#ifdef GDL_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#define A7_TACC , tacc
#else
#define A7_TACC
#endif
    __syncthreads();
    A7Wave& Wv = S.w[wave];
    const int nwin = g.wpr * g.wpc, L = g.H * g.W;
    const int gw_end = min(g.total, (chunk + 1) * g.chunk);
    int gw = chunk * g.chunk + wave;
    int n = gw / nwin, wy = (gw - n * nwin) / g.wpr, wx = gw - n * nwin - wy * g.wpr;
    const unsigned cl = (unsigned)(h * 64 + lq * 16);
    for (; gw < gw_end; gw += 4) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (size_t)n * L * 3 * g.ld), 0, L * 6 * g.ld, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(dout + (size_t)n * L * g.ld), 0, L * 2 * g.ld, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dqkv + (size_t)n * L * 3 * g.ld), 0, L * 6 * g.ld, 0x00020000);
        unsigned qoff[4], ooff[4];
        a7_offsets(g, c, wy, wx, l16, cl, qoff, ooff);
        const bool lr = g.shift && wy == g.wpc - 1, lc = g.shift && wx == g.wpr - 1;
        if (lr || lc)
            a7_bwd_window<true>(g, S, Wv, c, rq, ro, rd, qoff, ooff, lr ? 0xffffffffu : 0u, lc ? 0xffffffffu : 0u, D, dcol, lane, l16, lq A7_TACC);
        else
            a7_bwd_window<false>(g, S, Wv, c, rq, ro, rd, qoff, ooff, 0u, 0u, D, dcol, lane, l16, lq A7_TACC);
        if (h == 0 && g.ld > 32 * g.nh)
            for (int sgm = 0; sgm < 3; ++sgm) a7_zero_pad(g, rd, 6 * g.ld, wy, wx, lane, sgm * 2 * g.ld);
        a7_advance(g, n, wy, wx);
    }
#ifdef GDL_TIMING
    if (g.dbg && lane == 0 && bid < 4096) {
        unsigned long long* d = g.dbg + ((size_t)bid * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) d[k] = tacc[k];
        d[5] = __builtin_amdgcn_s_memtime() - tk0;  // the wave's life so far (its windows; the bias tile came before)
        d[6] = (unsigned long long)((gw_end - (chunk * g.chunk + wave) + 3) / 4);
    }
#endif
    // d(table) of the block: D of the four waves summed in wave order (a lane owns the same elements in every wave), then entry
    // (dh, dw) = sum of D[i][j] over the pairs with (ri - rj, ci - cj) = (dh, dw), rows ascending
    float* area = S.bias;
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
#pragma unroll
                for (int jt = 0; jt < 3; ++jt) {
                    float4* p = (float4*)&area[((it * 4 + jt) * 64 + lane) * 4];
                    float4 v = wv ? *p : make_float4(0.f, 0.f, 0.f, 0.f);
                    v.x += D[it][jt][0], v.y += D[it][jt][1], v.z += D[it][jt][2], v.w += D[it][jt][3];
                    *p = v;
                }
                if (lq == 0) {  // (query 16 it + l16, key 48) -> the slot of key tile 3, lane (0, l16 >> 2), r = l16 & 3
                    float* p = &area[((it * 4 + 3) * 64 + (l16 >> 2) * 16) * 4 + (l16 & 3)];
                    *p = (wv ? *p : 0.f) + dcol[it];
                }
            }
        }
    }
    __syncthreads();
    if (tid < 169) {
        const int dh = tid / 13 - 6, dw = tid % 13 - 6;
        float sum = 0.f;
        for (int rj = max(0, -dh); rj < min(7, 7 - dh); ++rj)
            for (int cj = max(0, -dw); cj < min(7, 7 - dw); ++cj) {
                const int i = (rj + dh) * 7 + cj + dw, j = rj * 7 + cj;
                // D[it][jt][r] of lane (l16, lq): query 16 it + 4 lq + r, key 16 jt + l16
                sum += area[(((i >> 4) * 4 + (j >> 4)) * 64 + ((i >> 2) & 3) * 16 + (j & 15)) * 4 + (i & 3)];
            }
        tpart[(size_t)bid * 169 + tid] = sum;
    }
}

static int a7_knob(const char* name, int dflt) {
    const char* e = tune_env(name);
    return e ? atoi(e) : dflt;
}
// windows per block: the four waves of a block share one head's bias tile (built once per block), so more windows per block
// amortise it.  Measured on MI355X at the four Swin-T stage shapes, 192 frames (forward / backward ms per step, 12 layers): 4 windows
// 0.80 / 1.74, 8: 0.67 / 1.41, 12: 0.62 / 1.34, 16: 0.65 / 1.40, 32: 0.71 / 1.46 -- three windows per wave, also where that
// leaves fewer blocks than CU slots (stage 4: 384 blocks).
static int a7_chunk(long total, int nh, bool bwd) {
    static int forced = -1;
    if (forced < 0) forced = a7_knob("GDL_SWIN_ATTN7_CHUNK", 0);
    if (forced > 0) return forced < 4 ? 4 : forced;  // (the workspace is sized for chunks of >= 4 windows)
    // ... unless a slightly larger block count fits the launch into ONE round of resident blocks (backward: 2 per CU, forward: 4)
    // where three windows per wave need a second, half-empty one: stage 3 of Swin-T at 192 frames, 768 windows x 12 heads, is
    // 768 blocks of 12 windows (1.5 rounds of 512) or 468 blocks of 20 -- backward 64 -> 57 us.
    const long slots = 256L * (bwd ? 2 : 4);
    if ((total + 11) / 12 * nh > slots)
        for (int c = 16; c <= 24; c += 4)
            if ((total + c - 1) / c * nh <= slots) return c;
    return 12;
}
static void a7_geom(A7Geom* g, int n_img, int H, int W, int shift, int nh, int ld, bool bwd) {
    g->H = H, g->W = W, g->nh = nh, g->ld = ld, g->shift = shift;
    g->wpr = W / 7, g->wpc = H / 7;
    g->total = n_img * g->wpr * g->wpc;
    g->chunk = a7_chunk(g->total, nh, bwd);
    static int xcd = -1;
    if (xcd < 0) xcd = a7_knob("GDL_SWIN_XCD", 1);
    g->xcd = xcd;
#ifdef GDL_TIMING
    g->dbg = g_timing_buf;
#endif
}
template <typename K>
static int a7_attr(K kernel, DevOnce* done, size_t lds = sizeof(A7Lds)) {
    if (*done) return GDL_OK;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(swin_attn7)");
    *done = true;
    return GDL_OK;
}

bool swin_attn7_ok(int dt, int H, int W, int ws, int shift, int nh, int ld, int n_img) {
    static int on = -1;
    if (on < 0) on = a7_knob("GDL_SWIN_ATTN7", 1);  // tuning aid: 0 = the general kernels of swin.hip for window 7 too
    return on && dt == GDL_BF16 && ws == 7 && H % 7 == 0 && W % 7 == 0 && shift >= 0 && shift < 7 && nh * 32 <= ld && ld % 8 == 0 &&
           (long)H * W * 6 * ld < (1l << 31) && n_img >= 1;
}
int swin_attn7_fwd(const void* qkv, const float* table, void* out, int n_img, int H, int W, int shift, int nh, int ld, hipStream_t st) {
    A7Geom g;
    a7_geom(&g, n_img, H, W, shift, nh, ld, false);
    static DevOnce attr;
    int rc = a7_attr(swin_attn7_fwd_kernel, &attr, sizeof(A7LdsF));
    if (rc) return rc;
    const int nchunks = (g.total + g.chunk - 1) / g.chunk;
    ProfScope prof("gdl::swin_attn7_fwd_kernel", PROF_HBM, st, (double)n_img * H * W * ld * 2 * 4);
    hipLaunchKernelGGL(swin_attn7_fwd_kernel, dim3(nchunks * nh), dim3(256), sizeof(A7LdsF), st, (const bf16*)qkv, table, (bf16*)out, g);
    GDL_CHECK_LAUNCH("swin_attn7_fwd_kernel");
    return GDL_OK;
}
// partial d(table) blocks [nparts][heads][169] float
size_t swin_attn7_bwd_ws_bytes(int n_img, int H, int W, int nh) {
    const long total = (long)n_img * (H / 7) * (W / 7);
    return (size_t)((total + 3) / 4) * nh * 169 * sizeof(float);  // the smallest chunk: an upper bound for every choice
}
int swin_attn7_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* tpart, int* nparts, int n_img, int H, int W,
                   int shift, int nh, int ld, hipStream_t st) {
    A7Geom g;
    a7_geom(&g, n_img, H, W, shift, nh, ld, true);
    static DevOnce attr;
    int rc = a7_attr(swin_attn7_bwd_kernel, &attr);
    if (rc) return rc;
    const int nchunks = (g.total + g.chunk - 1) / g.chunk;
    *nparts = nchunks;
    ProfScope prof("gdl::swin_attn7_bwd_kernel", PROF_HBM, st, (double)n_img * H * W * ld * 2 * 7);
    hipLaunchKernelGGL(swin_attn7_bwd_kernel, dim3(nchunks * nh), dim3(256), sizeof(A7Lds), st, (const bf16*)qkv, table, (const bf16*)dout,
                       (bf16*)dqkv, tpart, g);
    GDL_CHECK_LAUNCH("swin_attn7_bwd_kernel");
    return GDL_OK;
}

}  // namespace gdl
