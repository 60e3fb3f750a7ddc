// conv_wgrad9.hip -- weight gradient of the 3x3 stride-1 pad-1 convolutions, all 9 taps per block.
//
// Replaces the weight-gradient of nn.Conv2d(3x3, stride 1, padding 1)
// (/root/reference/models/backbone.py:20-23; 32 of the 40 convolutions of the two encoders):
//     dw[k][r][s][c] = sum_m dy[m][k] * x[m + (r-1)*W + (s-1)][c]      (flat NHWC pixel index m,
//                                                                        zero where the tap is padding)
// The per-tap kernel of conv_wgrad.hip re-reads dy and x once per tap: measured on MI355X it spends as
// long issuing its LDS-DMA (the L2 -> LDS path saturates at ~52 B/clk/CU, tools/micro/dma_issue.hip)
// as multiplying.  Here one block owns a 64(k) x 64(c) tile for ALL NINE taps and one slice of the
// pixel range: per 64-pixel stage it fetches the dy tile (8 KB) and ONE contiguous slab of x,
// pixels [m0-(W+1), m0+64+(W+1)), and every tap is a row shift of the LDS read address -- ~5x fewer
// bytes through the DMA path per flop.  The reduction index (pixels) is the slow index of both LDS
// tiles, so the MFMA fragments come from ds_read_b64_tr_b16 transpose reads, as in conv_wgrad.hip.
//
// Wave w of the 4 owns channels c0+16w..+15 for all 64 k: 36 accumulator fragments (9 taps x 4
// k-fragments = 144 VGPRs); per 32-pixel K-step it reads the 4 dy fragments once and one x fragment
// per tap, three taps ahead of the MFMAs (counted lgkmcnt).  All LDS addresses are precomputed per
// lane (the swizzle of a shifted slab row does not change from stage to stage); a padding tap is a
// per-lane select of the zero row, driven by the forward gather table's tap mask of the lane's pixel.
//
// Memory pipeline (see the kernel): every load of the loop is an LDS-DMA (the pixels' tap masks included:
// 4 bytes per lane into a 256-byte mask stage) and every wait is explicit -- with an ordinary global load
// in the loop the compiler, which cannot count the variable number of DMA instructions, put
// `s_waitcnt vmcnt(0)` in front of the first use of the masks, right after the next stage had been issued.
// Partials go to the workspace in fragment order ([slice][tile][wave][tap][kfrag][lane][4], 16-byte
// stores, 1 KiB per wave instruction); wgrad9_reduce_kernel folds the slices in a fixed order and
// scatters to the reference's [K][C][3][3] layout (deterministic, no atomics).
#include <stdlib.h>

#include "common.h"
#include "gather.h"
#include "prof.h"

#include <hip/hip_ext.h>

namespace gdl {

struct Wgrad9Args {
    const void* dy;            // [M][K] bf16
    const void* x;             // [M][C] bf16 (stride 1: same pixel grid as dy)
    float* partial;            // [nsplit][K/64][C/64][4 waves][9][4][64][4]
    const GatherEntry* table;  // forward gather table (tap masks), [M]
    int C, K, M, W;
    int nsplit, chunk;         // pixels per split (multiple of 64)
    int tiles_k, tiles_c;
    int slab_rows;             // 64 + 2W + 2
    unsigned dy_bytes, x_bytes;
#ifdef GDL_TIMING
    unsigned long long* dbg;
#endif
};
#ifdef GDL_TIMING
extern unsigned long long* g_timing_buf;
#endif

constexpr int W9_BP = 64;        // pixels per stage
// (slab ring: 256 rows = 32 KiB for 2W+2 <= 128, 512 rows = 64 KiB for 2W+2 <= 384; template parameter RING = KiB / 32)

__device__ __forceinline__ void w9_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, 0, 0, 0);
#else
    (void)rsrc;
    (void)lds_base;
    (void)voffset;
#endif
}
// 4 bytes per lane into LDS (tap masks)
__device__ __forceinline__ void w9_dma4(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 4, voffset, 0, 0, 0);
#else
    (void)rsrc;
    (void)lds_base;
    (void)voffset;
#endif
}
template <int OFF>
__device__ __forceinline__ unsigned w9_lds32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// transpose read hidden from the compiler (see conv_wgrad.hip), with an instruction byte offset
template <int OFF>
__device__ __forceinline__ uint2 w9_tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void w9_wait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// 32-byte granule swizzle of a 128-byte [pixel][64 channels] row (same as conv_wgrad.hip's wg_swz<128>)
__device__ __forceinline__ int w9_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// RING: consecutive stages' slabs overlap in 2W+2 of their 64+2W+2 rows, so the slab lives in a ring of 256 rows
// (32 KiB at LDS offset 0, pixel p at row (p - p0) mod 256) and a stage only loads its 64 NEW rows: 17-18 DMA pieces
// per stage instead of 32 at W = 56 (25 at W = 28).  A read address advances by 64 rows per stage and wraps with one
// AND; the second K-step's +4096 instruction offset may run past the ring's end, so rows 0..31 are mirrored behind it
// (4 extra pieces every fourth stage).  Used where it saves at least 6 pieces per stage (W >= 24); the narrow layers
// keep the plain double buffer, whose LDS footprint is smaller there.
template <int RING>
__global__ __launch_bounds__(256, 2) void conv_wgrad9_kernel(Wgrad9Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;

    // block -> (slice, ktile, ctile); the tiles of one pixel slice are neighbours on one XCD (or two / four when there are fewer
    // slices than XCDs: xcd_linear, common.h)
    const int per_slice = a.tiles_k * a.tiles_c;
    const int L = xcd_linear(a.nsplit * per_slice);
    if (L < 0) return;
    const int slice = L / per_slice, rem = L - slice * per_slice;
    const int kt = rem % a.tiles_k, ct = rem / a.tiles_k;
    const int k0 = kt * 64, c0 = ct * 64;
    const int m_begin = slice * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int nst = (m_end - m_begin + W9_BP - 1) / W9_BP;

    const int nins = (a.slab_rows + 7) >> 3;        // 1 KiB DMA pieces of the slab
    // plain: [stage 0: dy tile + slab][stage 1][zero row][mask stages]
    // RING:  [ring 32 KiB][mirror of rows 0..31, 4 KiB][dy tile 0][dy tile 1][zero row][mask stages]
    constexpr int W9_RING = RING * 32768;  // bytes of the ring (0: plain double buffer)
    const int STAGE = RING ? W9_BP * 128 : W9_BP * 128 + nins * 1024;
    const int DY0 = RING ? W9_RING + 4096 : 0;      // byte offset of dy tile 0
    const int ZOFF = RING ? DY0 + 2 * STAGE : 2 * STAGE;
    const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    if (RING && (smem_base & (unsigned)(W9_RING > 0 ? W9_RING - 1 : 0)) != 0) __builtin_trap();  // the wrap-by-AND needs the ring at a 32 KiB boundary
    unsigned char* zero_p = smem + ZOFF;            // 1 KiB zero row
    unsigned char* mask_st = zero_p + 1024;         // two 256-byte mask stages
    const unsigned zrow = smem_base + ZOFF;
    const int ringT = ((2 * a.W + 2 + 63) >> 6) << 6;  // ring row of the first new row of stage 0's successor block

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rtab =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.table, 0, (unsigned)a.M * (unsigned)sizeof(GatherEntry), 0x00020000);
    if (wave == 0) w9_dma16(rx, zero_p, (int)0x80000000);  // zero row (out-of-range DMA deposits zeros)

    // ---- DMA bookkeeping.  A piece = 8 rows x 128 B; lane L -> row L>>3, physical 16-byte chunk L&7;
    // the source chunk is the one whose swizzled position that is.  Pieces of a stage: 8 of dy, `nins` of
    // the slab, 1 of tap masks (4 bytes per lane: the second word of the pixels' gather entries; past the
    // slice: zeros); wave w takes pieces w, w+4, ...
    const int prow = lane >> 3, pch = lane & 7;
    int ld_m = m_begin;  // first pixel of the next stage to load
    int ld_s = 0;        // its index
    auto load_stage = [&](int buf) {
        unsigned char* Ks = smem + DY0 + buf * STAGE;
        unsigned char* Xs = Ks + W9_BP * 128;
        // RING: stage 0 fills ring rows [0, ringT + 64) with pixels p0 + row, p0 = m_begin - (W+1) - (ringT - 2W - 2);
        // stage s > 0 brings pixels m_s + (W+1) + [0, 64) into block (s + ringT/64) mod 4 (and its first 32 rows
        // into the mirror when that is block 0)
        const int blk = (ld_s + (ringT >> 6)) & (RING ? RING * 4 - 1 : 3);  // 64-row blocks of the ring
        const int nx = !RING ? nins : (ld_s == 0 ? (ringT >> 3) + 8 : (blk == 0 ? 12 : 8));
        const int np = 8 + nx + 1;
        for (int p = wave; p < np; p += 4) {
            if (p < 8) {
                const int row = p * 8 + prow, m = ld_m + row;
                const int ch = (((pch >> 1) ^ w9_swz(row)) << 1) | (pch & 1);
                w9_dma16(rdy, Ks + p * 1024, m < m_end ? m * (a.K * 2) + k0 * 2 + ch * 16 : (int)0x80000000);
            } else if (RING && p < 8 + nx) {
                int jj = p - 8;  // piece: 8 ring rows
                int pix, dst;
                if (ld_s == 0) {
                    pix = m_begin - (a.W + 1) - (ringT - 2 * a.W - 2) + jj * 8 + prow;
                    dst = jj * 1024;
                } else {
                    const bool mir = jj >= 8;  // second copy of the block's first 4 pieces
                    if (mir) jj -= 8;
                    pix = ld_m + (a.W + 1) + jj * 8 + prow;
                    dst = mir ? W9_RING + jj * 1024 : blk * 8192 + jj * 1024;
                }
                const int sr = jj * 8 + prow;  // bits 1 and 3 of the ring row (blocks are 64 rows: they do not change them)
                const int ch = (((pch >> 1) ^ w9_swz(sr)) << 1) | (pch & 1);
                w9_dma16(rx, smem + dst, (unsigned)pix < (unsigned)a.M ? pix * (a.C * 2) + c0 * 2 + ch * 16 : (int)0x80000000);
            } else if (!RING && p < 8 + nins) {
                const int jj = p - 8, sr = jj * 8 + prow;
                const int pix = ld_m - (a.W + 1) + sr;
                const bool ok = sr < a.slab_rows && (unsigned)pix < (unsigned)a.M;
                const int ch = (((pch >> 1) ^ w9_swz(sr)) << 1) | (pch & 1);
                w9_dma16(rx, Xs + jj * 1024, ok ? pix * (a.C * 2) + c0 * 2 + ch * 16 : (int)0x80000000);
            } else {
                const int m = ld_m + lane;
                w9_dma4(rtab, mask_st + buf * 256, m < m_end ? m * (int)sizeof(GatherEntry) + 4 : (int)0x80000000);
            }
        }
        ld_m += W9_BP;
        ld_s += 1;
    };

    // ---- stages >= 1, four-wave form: the same pieces with every per-lane term hoisted out of the loop.  Wave w issues dy
    // pieces w and w+4, slab pieces w, w+4 (, ...), wave 0 the mask piece; a piece's swizzled source chunk depends on bits 1 and
    // 3 of its row, which w and w+4 share, so ONE per-lane byte offset per operand advances by 64 pixels per stage and the
    // other pieces are a scalar step away.  Bounds need no test: a pixel outside the tensor is outside the buffer descriptor
    // (zeros), and a slice is a whole number of stages except at the tensor's end.  (The generic loop above spent about as
    // long computing piece addresses as the DMA instructions themselves take.)
    const int K2 = a.K * 2, C2 = a.C * 2;
    int dyb, xb, mb;
    {
        const int row = wave * 8 + prow;
        const int ch = (((pch >> 1) ^ w9_swz(row)) << 1) | (pch & 1);
        dyb = (m_begin + W9_BP + row) * K2 + k0 * 2 + ch * 16;
        xb = (m_begin + W9_BP + (RING ? a.W + 1 : -(a.W + 1)) + row) * C2 + c0 * 2 + ch * 16;
        mb = (m_begin + W9_BP + lane) * (int)sizeof(GatherEntry) + 4;
    }
    auto load_fast = [&](int buf) {
        unsigned char* Ks = smem + DY0 + buf * STAGE;
        w9_dma16(rdy, Ks + wave * 1024, dyb);
        w9_dma16(rdy, Ks + (wave + 4) * 1024, dyb + 32 * K2);
        if (RING) {
            const int rb = (ld_s + (ringT >> 6)) & (RING ? RING * 4 - 1 : 3);
            unsigned char* dst = smem + rb * 8192;
            w9_dma16(rx, dst + wave * 1024, xb);
            w9_dma16(rx, dst + (wave + 4) * 1024, xb + 32 * C2);
            if (rb == 0) w9_dma16(rx, smem + W9_RING + wave * 1024, xb);  // mirror of the ring's first 32 rows
        } else {
            unsigned char* Xs = Ks + W9_BP * 128;
            int o = 0;
            for (int jj = wave; jj < nins; jj += 4, o += 32 * C2) w9_dma16(rx, Xs + jj * 1024, xb + o);
        }
        if (wave == 0) w9_dma4(rtab, mask_st + buf * 256, mb);
        dyb += W9_BP * K2;
        xb += W9_BP * C2;
        mb += W9_BP * (int)sizeof(GatherEntry);
        ld_m += W9_BP;
        ld_s += 1;
    };

    // ---- per-lane LDS read addresses (stage buffer 0; the other buffer is +STAGE).
    // transpose read: lane supplies row (li>>2) of its 16-lane group's 4x16 block, 8 bytes at element
    // (li&3)*4 of the fragment's 16 channels; group g covers pixels g*8 + h*4 + 0..3 of the K-step.
    const int lrow = g * 8 + (li >> 2);  // + h*4 + ks*32
    unsigned aaddr[4][2];                // dy fragment i (k = 16i..), half h
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lrow + h * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) aaddr[i][h] = smem_base + DY0 + row * 128 + ((i ^ w9_swz(row)) << 5) + (li & 3) * 8;
    }
    unsigned baddr[9][2];  // x fragment of tap t (this wave's 16 channels), half h
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sh = (t / 3 - 1) * a.W + (t % 3 - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // plain: slab row of stage buffer 0; RING: ring row of stage 0 (the ring starts ringT - (2W+2) pixels early)
            const int sr = lrow + h * 4 + (a.W + 1) + sh + (RING ? ringT - 2 * a.W - 2 : 0);
            baddr[t][h] = smem_base + (RING ? 0 : W9_BP * 128) + sr * 128 + ((wave ^ w9_swz(sr)) << 5) + (li & 3) * 8;
        }
    }
    const unsigned zaddr = zrow + (li & 3) * 8;  // any 8 bytes of the zero KiB (K-step 1 reads at +4096: pre-biased there)
    // tap masks of this lane's 4 pixels of a stage ([ks][h]: pixel ks*32 + lrow + h*4) come from the mask stage
    const unsigned maddr = zrow + 1024 + lrow * 4;

    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#ifdef GDL_TIMING
    unsigned long long t_entry = __builtin_amdgcn_s_memtime(), t_wait = 0, t_issue = 0, t_a, t_b;
#endif
    if (nst > 0) load_stage(0);
#ifdef GDL_TIMING
    unsigned long long t_first = __builtin_amdgcn_s_memtime();
#endif
    for (int st = 0; st < nst; ++st) {
#ifdef GDL_TIMING
        t_a = __builtin_amdgcn_s_memtime();
#endif
        // the stage issued one iteration ago has landed in every wave's part; every wave is done with the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef GDL_TIMING
        t_b = __builtin_amdgcn_s_memtime();
        t_wait += t_b - t_a;
#endif
        // ---- round 4: the stage's first LDS reads (tap masks, the dy fragments of K-step 0) are requested BEFORE the next
        // stage's DMA is issued, so that their latency runs under the ~250 clk the issue holds the wave (they were two exposed
        // round trips per stage: ~200 clk for the masks, ~150 for the fragments; tools/timing_wgrad.py)
        unsigned pmask[2][2];
        uint2 af[2][4][2];  // [K-step][k fragment][half]: K-step 1's set is requested in the middle of K-step 0
        {
            const unsigned ma = maddr + (st & 1) * 256;
            pmask[0][0] = w9_lds32<0>(ma);
            pmask[0][1] = w9_lds32<16>(ma);
            pmask[1][0] = w9_lds32<128>(ma);
            pmask[1][1] = w9_lds32<144>(ma);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) af[0][i][h] = w9_tr<0>(aaddr[i][h]);
        }
        if (st + 1 < nst) load_fast((st + 1) & 1);
        w9_wait<0>();
#ifdef GDL_TIMING
        t_a = __builtin_amdgcn_s_memtime();
        t_issue += t_a - t_b;
#endif
        // ---- 18 steps u = ks * 9 + t (two K-steps of 32 pixels x nine taps; K-step 1 is 32 rows = 4096 bytes further:
        // instruction offset).  One software pipeline over all 18: the x fragment of step u is requested W9_AHEAD steps early
        // (also across the K-step boundary), K-step 1's dy fragments at step W9_A1, every wait is a counted lgkmcnt on the
        // in-order LDS queue: younger requests = 2 per x fragment issued behind step u's + 8 when the dy set sits behind it.
        constexpr int W9_AHEAD = 3, W9_A1 = 4;
        uint2 bf[W9_AHEAD + 1][2];
        auto issue_b = [&](int u) __attribute__((always_inline)) {
            const int ks = u / 9, t = u % 9;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // bit t of the mask, sign-extended, selects between the slab row and the zero row (bfe + bfi)
                // (two VALU instructions; the compiler's own select takes three: and, compare, cndmask)
                unsigned sel, ad;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(pmask[ks][h]), "n"(t));
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(sel), "v"(baddr[t][h]), "v"(ks ? zaddr - 4096u : zaddr));
                bf[u % (W9_AHEAD + 1)][h] = ks ? w9_tr<4096>(ad) : w9_tr<0>(ad);
            }
        };
#pragma unroll
        for (int u = 0; u < W9_AHEAD; ++u) issue_b(u);
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int ks = u / 9, t = u % 9;
            if (u + W9_AHEAD < 18) issue_b(u + W9_AHEAD);
            if (u == W9_A1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) af[1][i][h] = w9_tr<4096>(aaddr[i][h]);
            }
            // requests younger than step u's x fragment
            const int nb = (18 - 1 - u < W9_AHEAD ? 18 - 1 - u : W9_AHEAD) * 2;
            // the dy set of K-step 1 was issued at step W9_A1, behind the x fragments of steps <= W9_A1 + W9_AHEAD
            const bool a1_younger = u >= W9_A1 && u <= W9_A1 + W9_AHEAD && u < 9;
            // (steps 9.. need the dy set itself: it is older than their x fragment, so the same wait covers it)
            switch (nb + (a1_younger ? 8 : 0)) {
                case 0: w9_wait<0>(); break;
                case 2: w9_wait<2>(); break;
                case 4: w9_wait<4>(); break;
                case 6: w9_wait<6>(); break;
                case 8: w9_wait<8>(); break;
                case 10: w9_wait<10>(); break;
                case 12: w9_wait<12>(); break;
                default: w9_wait<14>(); break;
            }
            const uint2* b2 = bf[u % (W9_AHEAD + 1)];
            const uint4 fb = make_uint4(b2[0].x, b2[0].y, b2[1].x, b2[1].y);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 fa = make_uint4(af[ks][i][0].x, af[ks][i][0].y, af[ks][i][1].x, af[ks][i][1].y);
                acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                   __builtin_bit_cast(bf16x8_t, fb), acc[t][i], 0, 0, 0);
            }
        }
        // next stage lives in the other buffer: move every read address over
        const int dlt = (st & 1) ? -STAGE : STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) aaddr[i][h] += dlt;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) baddr[t][h] = RING ? ((baddr[t][h] + 8192u) & (unsigned)(W9_RING > 0 ? W9_RING - 1 : 0)) : baddr[t][h] + dlt;
    }
#ifdef GDL_TIMING
    unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
    if (a.dbg && lane == 0) {  // one record per wave: [block][wave]
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = t_entry;
        d[1] = t_first - t_entry;
        d[2] = t_wait;
        d[3] = t_issue;  // (mask + first dy fragment requests, DMA issue, their wait)
        d[4] = (t_loop_end - t_first) - t_wait - t_issue;
        d[6] = nst;
        d[7] = 0;
    }
#endif
    // ---- partial tile in fragment order: [slice][kt][ct][wave][tap][i][lane][4]
    float4* part = (float4*)a.partial + ((((size_t)slice * per_slice + rem) * 4 + wave) * 36) * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            part[(t * 4 + i) * 64 + lane] = make_float4(acc[t][i][0], acc[t][i][1], acc[t][i][2], acc[t][i][3]);
#ifdef GDL_TIMING
    if (a.dbg && lane == 0) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + 5] = __builtin_amdgcn_s_memtime() - t_loop_end;
#endif
}

// out[k][c][r][s] = sum_slice partial[slice][tile][wave][tap][i][lane][e], fixed order.
// Block = 256 threads = 256/SL consecutive float4 outputs x SL split-lanes (lane l sums slices l, l+SL, ..., four loads
// in flight); thread = l * (256/SL) + o, so the 256/SL threads of one split-lane read one contiguous run of a slice.
// SL = 4 when there are at most 4 slices (the 512-channel layers: no idle split-lanes), else 16.
// D[i][j] fragment layout: k = 16i + (lane>>4)*4 + e, c = 16*wave + (lane&15).
template <int SL>
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(const float4* __restrict__ partial, float* __restrict__ out,
                                                            int nsplit, int K, int C, int tiles_k, int tiles_c) {
    constexpr int OPB = 256 / SL;
    __shared__ float4 red[SL][OPB];
    const size_t total4 = (size_t)K * C * 9 / 4;  // float4 elements per slice
    const int o = threadIdx.x % OPB, l = threadIdx.x / OPB;
    const size_t idx = blockIdx.x * (size_t)OPB + o;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total4) {
        const float4* p = partial + idx;
        int sp = l;
        for (; sp + 3 * SL < nsplit; sp += 4 * SL) {
            const float4 v0 = p[(size_t)sp * total4], v1 = p[(size_t)(sp + SL) * total4];
            const float4 v2 = p[(size_t)(sp + 2 * SL) * total4], v3 = p[(size_t)(sp + 3 * SL) * total4];
            s.x += v0.x, s.y += v0.y, s.z += v0.z, s.w += v0.w;
            s.x += v1.x, s.y += v1.y, s.z += v1.z, s.w += v1.w;
            s.x += v2.x, s.y += v2.y, s.z += v2.z, s.w += v2.w;
            s.x += v3.x, s.y += v3.y, s.z += v3.z, s.w += v3.w;
        }
        for (; sp < nsplit; sp += SL) {
            const float4 v0 = p[(size_t)sp * total4];
            s.x += v0.x, s.y += v0.y, s.z += v0.z, s.w += v0.w;
        }
    }
    red[l][o] = s;
    __syncthreads();
    if (l == 0 && idx < total4) {
#pragma unroll
        for (int jj = 1; jj < SL; ++jj) {
            const float4 v = red[jj][o];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        // decode fragment order: idx = (((tile*4 + wave)*9 + tap)*4 + i)*64 + lane
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int i = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % 9);
        r /= 9;
        const int wave = (int)(r & 3);
        const int tile = (int)(r >> 2);
        const int kt = tile % tiles_k, ct = tile / tiles_k;
        const int k = kt * 64 + i * 16 + (lane >> 4) * 4, c = ct * 64 + wave * 16 + (lane & 15);
        float* dst = out + ((size_t)k * C + c) * 9 + tap;
        const size_t kstride = (size_t)C * 9;
        dst[0] = s.x;
        dst[kstride] = s.y;
        dst[2 * kstride] = s.z;
        dst[3 * kstride] = s.w;
    }
}

// ---------------------------------------------------------------- host side
// The fold for at most 4 slices (the 512-channel layers: 9.4 MB of output), with the scatter staged through LDS:
// a block owns one (tile, wave, k-fragment i) = a 16(k) x 16(c) patch for ALL nine taps, so each of its 16 k rows is
// one contiguous run of 16 c x 9 taps = 576 bytes of out[k][c][3][3] (4-byte stores at a 36-byte stride made the plain
// kernel store-bound: 32 us against 13 us for the 5x smaller outputs of the other layers).
__global__ __launch_bounds__(256) void wgrad9_reduce_rows_kernel(const float4* __restrict__ partial, float* __restrict__ out,
                                                                 int nsplit, int K, int C, int tiles_k, int tiles_c) {
    __shared__ float patch[16][16 * 9];
    const size_t total4 = (size_t)K * C * 9 / 4;
    const int lane = threadIdx.x & 63, w4 = threadIdx.x >> 6;  // wave w4 folds taps w4, w4+4, w4+8 over all slices
    const int grp = blockIdx.x;                                // (tile*4 + wave)*4 + i
    const int i = grp & 3, wave = (grp >> 2) & 3, tile = grp >> 4;
    const int kk = (lane >> 4) * 4, cc = lane & 15;  // patch row = k - 16i, column = (c - 16 wave)*9 + tap
    for (int tap = w4; tap < 9; tap += 4) {
        const size_t idx = (((size_t)(tile * 4 + wave) * 9 + tap) * 4 + i) * 64 + lane;
        float4 s = partial[idx];
        for (int sp = 1; sp < nsplit; ++sp) {  // fixed order
            const float4 v = partial[(size_t)sp * total4 + idx];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        patch[kk + 0][cc * 9 + tap] = s.x;
        patch[kk + 1][cc * 9 + tap] = s.y;
        patch[kk + 2][cc * 9 + tap] = s.z;
        patch[kk + 3][cc * 9 + tap] = s.w;
    }
    __syncthreads();
    const int kt = tile % tiles_k, ct = tile / tiles_k;
    const int k0 = kt * 64 + i * 16, c0 = ct * 64 + wave * 16;
    for (int t = threadIdx.x; t < 16 * 36; t += 256) {  // 16 rows x 36 float4
        const int r = t / 36, q = t - r * 36;
        *(float4*)(out + ((size_t)(k0 + r) * C + c0) * 9 + 4 * q) = *(const float4*)&patch[r][4 * q];
    }
}

struct W9Plan {
    int nsplit, chunk;
};
static W9Plan plan_w9(int M, int C, int K) {
    const int tiles = (K / 64) * (C / 64);
    // 512 blocks win when the kernel runs alone (tools/bench_conv.py); inside the step, where four streams share the CUs and
    // every block leaves 144 KB of partials, 256 do (bench.py: -2 % step time; re-swept in rounds 3 and 4: the default stayed --
    // the GDL_WGRAD9_BLOCKS knob went to tools/experiments/r5_pruned_knobs.diff.txt)
    constexpr int target = 256;
    // every block leaves 144 KB of fp32 partials that the reduce kernel reads back: at least `min_st` 64-pixel stages of work
    // per block.  Round 3 (tools/ab_env.sh, same box, three rounds, B = 64 step): 8 -> 5.72 ms, 36 -> 5.70, 48 -> 5.70,
    // 72 -> 5.79, 110 -> 6.07.  36 leaves the visual layers at their 256 slices of 37 stages and brings the audio layers
    // (whose slices were 12 stages long) from 256 to 85-128 blocks: 680 MB instead of 980 MB of partials written and read back
    // per step at the same step time.  (48 -- 196 visual slices, 535 MB -- is as fast in the step, but the kernel itself
    // then runs on three quarters of the CUs: 94 instead of 84 us per launch in the step, 53 instead of 39 us alone.)
    static int min_st = -1;
    if (min_st < 0) {
        // Round 6, re-swept after the chains' kernels got shorter (conv_pslab.h; same box, ms per step, three rounds each):
        // 36 -> 4.96, 48 -> 4.96, 52 -> 4.88, 56 -> 4.88, 60 -> 4.88, 64 -> 4.92, 72 -> 4.91, 90 -> 4.97.  56: the visual layers run
        // 168-192 blocks of 56 stages (two thirds of the CUs; the chains get the rest), 0.45 GB less of partials written and read
        // back per step -- the kernel's HBM traffic falls from 1.53x to 1.35x of its algorithmic bytes -- and a launch takes 92
        // instead of 81 us in the step (53 instead of 40 alone): the PER-LAUNCH roofline fraction bench.py reports for this kernel
        // goes DOWN (0.146 -> 0.129) while the step gets faster.  The step is what is optimised.
        const char* e = tune_env("GDL_WGRAD9_MINST");  // tuning aid
        min_st = e ? atoi(e) : 56;
    }
    int ns = (target + tiles - 1) / tiles;
    const int max_ns = (M + min_st * W9_BP - 1) / (min_st * W9_BP);
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    int chunk = (M + ns - 1) / ns;
    chunk = (chunk + W9_BP - 1) / W9_BP * W9_BP;
    W9Plan p;
    p.nsplit = (M + chunk - 1) / chunk;
    p.chunk = chunk;
    return p;
}

bool conv_wgrad9_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_WGRAD9");  // tuning aid: 0 = per-tap kernel everywhere
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

size_t conv_wgrad9_ws_bytes(int M, int C, int K) {
    const W9Plan p = plan_w9(M, C, K);
    return (size_t)p.nsplit * K * 9 * C * sizeof(float);
}

// ring size class of a layer width: 0 = plain double buffer, 1 = 256-row ring, 2 = 512-row ring
static int w9_ring(int W) {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_WGRAD9_RING");  // tuning aid: 0 = plain double buffer everywhere
        v = e ? atoi(e) : 1;
    }
    if (v == 0 || W < 24) return 0;
    if (2 * W + 2 <= 128) return 1;
    // wider layers (the 157-pixel audio layer 1 of the Kinetics-Sounds / VGGSound shapes): the plain double buffer would
    // not fit at all; the 512-row ring does (85.5 KB: one block per CU)
    const int rows = W9_BP + 2 * W + 2;
    const size_t plain = 2 * (size_t)(W9_BP * 128 + ((rows + 7) / 8) * 1024) + 1024 + 512;
    return (plain > 80 * 1024 && 2 * W + 2 <= 384) ? 2 : 0;
}
static size_t w9_lds_bytes(int W) {
    const int ring = w9_ring(W);
    if (ring) return (size_t)ring * 32768 + 4096 + 2 * W9_BP * 128 + 1024 + 512;  // ring + mirror + dy tiles + zero row + masks
    const int rows = W9_BP + 2 * W + 2;
    // two stages (dy tile + slab) + zero row + two mask stages
    return 2 * (size_t)(W9_BP * 128 + ((rows + 7) / 8) * 1024) + 1024 + 512;
}

// true if this geometry runs on the 9-tap kernel
bool conv_wgrad9_ok(int dtype, int W, int C, int K, int R, int S, int stride, int pad) {
    return conv_wgrad9_enabled() && dtype == GDL_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && C % 64 == 0 &&
           K % 64 == 0 && w9_lds_bytes(W) <= 96 * 1024;
}

int conv_wgrad9(const void* dy, const void* x, float* dw, const void* table, int N, int H, int W, int C, int K, void* ws,
                size_t ws_bytes, hipStream_t st) {
    Wgrad9Args a{};
#ifdef GDL_TIMING
    a.dbg = g_timing_buf;
#endif
    a.dy = dy;
    a.x = x;
    a.table = (const GatherEntry*)table;
    a.C = C;
    a.K = K;
    a.M = N * H * W;
    a.W = W;
    GDL_REQUIRE(a.M < (1 << 24), "wgrad: M exceeds 2^24");
    GDL_REQUIRE((size_t)a.M * K * 2 < (1UL << 31) && (size_t)a.M * C * 2 < (1UL << 31), "wgrad: tensor exceeds 2 GiB");
    a.dy_bytes = (unsigned)((size_t)a.M * K * 2);
    a.x_bytes = (unsigned)((size_t)a.M * C * 2);
    const W9Plan p = plan_w9(a.M, C, K);
    a.nsplit = p.nsplit;
    a.chunk = p.chunk;
    a.tiles_k = K / 64;
    a.tiles_c = C / 64;
    a.slab_rows = W9_BP + 2 * W + 2;
    const size_t need = (size_t)p.nsplit * K * 9 * C * sizeof(float);
    if (ws_bytes < need || !ws) {
        set_error("wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return GDL_ERR_WORKSPACE;
    }
    a.partial = (float*)ws;
    const size_t lds = w9_lds_bytes(W);
    const int ring = w9_ring(W);
    using KernT = void (*)(Wgrad9Args);
    static const KernT kerns[3] = {conv_wgrad9_kernel<0>, conv_wgrad9_kernel<1>, conv_wgrad9_kernel<2>};
    const KernT kern = kerns[ring];
    static DevOnce attr_set[3];
    if (!attr_set[ring]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv_wgrad9)");
        attr_set[ring] = true;
    }
    const int per_slice = a.tiles_k * a.tiles_c;
    const int grid = xcd_grid(a.nsplit * per_slice);
    {
        // (algorithmic bytes: dy and x once + the fp32 result; the partials and their fold are overhead, not algorithm)
        ProfScope prof("gdl::conv_wgrad9_kernel", PROF_MFMA, st, 2.0 * (double)a.M * K * C * 9, true,
                       2.0 * (double)a.M * (K + C) + 4.0 * (double)K * C * 9);
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, prof.e0(), prof.e1(), 0, a);
        GDL_CHECK_LAUNCH("conv_wgrad9_kernel");
    }
    const size_t total4 = (size_t)K * C * 9 / 4;
#ifdef GDL_EXPERIMENT
    if (experiment_mask2() & 1) return GDL_OK;
#endif
    ProfScope prof("gdl::wgrad9_reduce_kernel", PROF_HBM, st, (double)total4 * 16.0 * (p.nsplit + 1));
    if (p.nsplit <= 4)
        hipLaunchKernelGGL(wgrad9_reduce_rows_kernel, dim3((unsigned)(a.tiles_k * a.tiles_c * 16)), dim3(256), 0, st,
                           (const float4*)a.partial, dw, p.nsplit, K, C, a.tiles_k, a.tiles_c);
    else
        hipLaunchKernelGGL(wgrad9_reduce_kernel<16>, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, st,
                           (const float4*)a.partial, dw, p.nsplit, K, C, a.tiles_k, a.tiles_c);
    GDL_CHECK_LAUNCH("wgrad9_reduce_kernel");
    return GDL_OK;
}

}  // namespace gdl
