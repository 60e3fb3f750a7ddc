// conv_pslab.h -- persistent, software-pipelined 3x3 stride-1 slab kernel for >= 128 channels (round 6; bf16).
// Included by conv_igemm.hip behind conv3x3_slab_kernel (it uses that file's ConvArgs, dma16, the asm LDS reads and swz128).
//
// Same GEMM view, slab and weight-ring layout as conv3x3_slab_kernel (nn.Conv2d 3x3 / stride 1 / pad 1 forward and its input
// gradient, /root/reference/models/backbone.py:20-23,52-68), rebuilt around what tools/timing_probe.py showed on HEAD of round 6
// (profiles/r06_timing_probe_before.txt): a wave of the old kernel spends ~2 000 clk per K-step (tap x 64 channels) of which
// 768 are its 48 MFMAs -- also when it is ALONE on its SIMD (layer 4: one block per CU) -- because a K-step is a serial chain
// barrier -> fragment reads -> DMA issue -> LDS latency -> 24 MFMAs -> reads -> LDS latency -> 24 MFMAs; and a block spends a
// third of its life outside the K-loop (prologue 13 %, epilogue 19 %: 26 % for the data gradient).
//
//  * K-loop as a two-buffer pipeline at HALF-step granularity.  F0 / F1 are the two fragment sets (MI pixel + 8 weight
//    operands each).  Step k:  [A] MFMAs on F0 = (k, channels 0-31), with the reads of (k, channels 32-63) -> F1 slipped
//    between them;  [B] wait + ONE barrier (every wave holds step k's fragments in registers: ring slot k % 2 is free; every
//    wave's pieces of weight tile k + 1 have landed);  [C] MFMAs on F1, with the DMA pieces of weight tile k + 2 -> slot k % 2
//    and the reads of (k + 1, channels 0-31) -> F0 slipped between them.  Every LDS read has >= 200 clk of MFMAs to land behind,
//    a DMA piece a whole K-step; the order is pinned with sched_barriers; there is no run-time branch inside an MFMA stream
//    (tools/micro/kstep.hip: ~20 clk each there).  With two slab buffers (template parameter TWO) the next chunk's slab arrives
//    ONE piece per step; with one, its request follows the MFMAs of the chunk's last step and is waited for at once.
//  * Persistent: <= 512 blocks walk the (M-tile, N-tile) items in XCD-linear order; the weight stream simply continues across
//    a block's tiles (the N-tile of a block is constant), the next tile's first slab chunk and tap masks are requested in
//    the last step of the current one and land behind its epilogue.
//  * Epilogue without block barriers and without touching the slab / ring: a wave stages 16 pixels x 64 channels at a time
//    through its OWN 2.25 KiB of LDS (accumulator layout -> rows), 2 * MI rounds; the data gradient's partner vectors are
//    requested behind the tile's last MFMAs -- all rounds' for the common launch (ReLU bits + one BatchNorm partner), two rounds
//    ahead for the launches with an addend / second partner (template parameter RICH).
//  * Statistics (forward: sum / sum of squares; data gradient: the BatchNorm-backward sums of ops.h BwdStats) are reduced AND
//    scattered across the eight lanes that hold the same channels (v_permlane32/16_swap of value pairs + a DPP rotation), added
//    per tile into the wave's LDS row -- every lane its own 16 bytes -- and leave the block ONCE, as one partial row per block
//    (conv_tiles_m / conv_dgrad_tiles_m report the grid) or as one set of integer atomics (bnacc.h).  Every sum has a fixed
//    order: run-to-run bit-identical.
// LDS: [weight ring 2 x 16 KiB][slab x 1 or 2][1 KiB of zeros unless the slab has spare rows][staging 4 x 2 304 B][sums 6 KiB]
// = 80 128 B for the 28-pixel-wide visual layer 2 with one slab buffer: two blocks per CU.
#pragma once

constexpr int PS_GRID = 512;              // two blocks per CU
constexpr int PS_PITCH = 144;             // staged row: 64 bf16 + 16 bytes
constexpr int PS_STAGE = 16 * PS_PITCH;   // per wave
constexpr int PS_STAT = 4 * 1536;         // [wave]: [chunk 8][k 2][h 2][e 8] floats of the first two sums + [chunk 8][h 2][e 8] of the third
constexpr int PS_SLAB_PIECES = 8;         // DMA pieces per wave and slab load (a slab buffer is at most 32 KiB)

static size_t pslab_lds_bytes(int BM, int W, int nslab) {
    const int rows = BM + 2 * W + 2;
    const size_t slab = (size_t)((rows + 7) / 8) * 1024;
    return 2 * (size_t)128 * 128 + nslab * slab + ((rows & 7) ? 0 : 1024) + 4 * PS_STAGE + PS_STAT;
}
// grid of a launch: at most PS_GRID blocks, eight equal XCD shares; with several items per block the blocks of an XCD stride by a
// multiple of 4 (the N-tile counts are 1, 2 and 4: a block keeps its N-tile).  GDL_PSLAB_GRID (tuning aid, also what the operator
// tests use to run several tiles per block at small sizes) caps it.
static int pslab_grid(int items) {
    static int cap = -1;
    if (cap < 0) {
        const char* e = tune_env("GDL_PSLAB_GRID");
        cap = e ? atoi(e) : PS_GRID;
        if (cap < 32 || cap > PS_GRID) cap = PS_GRID;
        cap &= ~31;
    }
    const int per = (items + 7) / 8;
    return 8 * (per <= cap / 8 ? per : cap / 8);
}

template <int OFF>
__device__ __forceinline__ void lds_write8_asm_off(unsigned addr, const uint2& v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ f32x4_t lds_read_f4_asm(unsigned addr) {
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_write_f4_asm(unsigned addr, const f32x4_t& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// LDS-DMA with a wave-uniform part of the source offset in the instruction's scalar offset (no VALU per piece)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_row_base, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_row_base, 16, voffset, soffset, 0, 0);
#else
    (void)rsrc, (void)lds_row_base, (void)voffset, (void)soffset;
#endif
}

// timing experiments of the -DGDL_TIMING build (WRONG results): GDL_PSLAB_DBG bits -- 1: no vmcnt wait at the step's barrier,
// 2: no weight DMA inside the K-loop, 4: no barrier, 8: no slab prefetch (two-buffer mode)
#ifdef GDL_TIMING
#define PS_DBG(bit) ((a.dbg_mode & (bit)) != 0)
// per-step cycle split of wave 0 without disturbing the pipeline: s_memtime results are only consumed behind the NEXT step's own
// lgkmcnt(0) wait (tools/timing_probe.py prints them: F0 wait | [A] | wait + barrier | [C] | tail)
#ifdef GDL_TIMING_SPLIT  // (five s_memtime per step cost ~15 % themselves: the split is its own build)
#define PS_T(v) asm volatile("s_memtime %0" : "=s"(v))
#else
#define PS_T(v) \
    do {        \
    } while (0)
#endif
#else
#define PS_DBG(bit) false
#define PS_T(v) \
    do {        \
    } while (0)
#endif

// compile-time experiments (WRONG results; `make BUILD=build_expN EXTRA="-DGDL_TIMING -DPS_EXP=N"`): bits -- 1: no tap masks (no zero
// row), 2: no weight DMA instructions, 4: no wait / barrier at [B] beyond lgkmcnt(0), 8: the fragment addresses are not recomputed
#ifndef PS_EXP
#define PS_EXP 0
#endif

template <int MI, int MODE, bool TWO, bool RICH>
__global__ __launch_bounds__(256, 2) void conv3x3_pslab_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NI = 8, BM = 64 * MI, BN = 128, WTM = 16 * MI, WSTAGE = BN * 128, NQ = NI * MI, NR = MI + NI;
    constexpr bool BWD = MODE == MODE_DGRAD;
    static_assert(MI == 2 || MI == 3, "MI");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    GDL_STAMP(0);

    // ---- this block's items: XCD x owns a consecutive run of the (M-tile, N-tile) order, its blocks take every bpx-th item of it
    // (neighbours in time share a slab / the weight tiles in the XCD's L2; bpx is a multiple of the N-tile count whenever a block
    // has more than one item, so a block's N-tile -- its output channels, its weight stream, its statistics -- never changes)
    const int ntn = a.OC / BN, items = a.mtiles * ntn;
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int per_xcd = (items + 7) >> 3;
    const int it_end = min(items, (xcd + 1) * per_xcd);
    int item = xcd * per_xcd + bj;
    const int ntiles = item < it_end ? (it_end - 1 - item) / bpx + 1 : 0;
    const int n0 = (item % ntn) * BN;
    const int kpt = a.IC >> 6, NS = kpt * 9;

    const int nins = (a.slab_rows + 7) >> 3;
    const int slab_bytes = nins * 1024;
    constexpr int nslab = TWO ? 2 : 1;  // slab buffers (the launcher picks the instantiation: ConvArgs.single_slab)
    const unsigned smem_base = lds_addr(smem);
    const unsigned slab_base = smem_base + 2 * WSTAGE;
    const bool spare_row = (a.slab_rows & 7) != 0;
    const unsigned zrow = spare_row ? slab_base + a.slab_rows * 128 : slab_base + nslab * slab_bytes;
    const unsigned tail = 2 * WSTAGE + nslab * slab_bytes + (spare_row ? 0 : 1024);
    const unsigned stg = smem_base + tail + wave * PS_STAGE;
    const unsigned statw = smem_base + tail + 4 * PS_STAGE + wave * 1536;  // this wave's sums row

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

    // the wave's sums row <- 0 (only this wave touches it until the final fold)
    {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        lds_write_f4_asm(statw + lane * 16, z);
        if (lane < 32) lds_write_f4_asm(statw + 1024 + lane * 16, z);
    }
    if (wave == 0 && !spare_row) dma16(rin, smem + (zrow - smem_base), (int)0x80000000);  // an out-of-range LDS-DMA deposits zeros

    // weight tile DMA: lane -> row (tid >> 3) + 32 i, physical chunk tid & 7 (source chunk swizzled); the row group i, the tap
    // and the channel chunk go into the scalar offset
    const int b_off = (n0 + (tid >> 3)) * 9 * a.IC * 2 + (((tid & 7) ^ swz128(tid >> 3)) << 4);
    const int w_istride = 32 * 9 * a.IC * 2;
    int wtap = 0, wkc = 0;     // cursor of the weight stream: the next tile to request
    int w_left = ntiles * NS;  // tiles still to request
    int b_cur = ntiles > 0 ? b_off : (int)0x80000000;  // the lane's source offset; out of range once the block's stream has ended
    auto w_piece = [&](int slot, int i) __attribute__((always_inline)) {
        dma16s(rwt, smem + slot * WSTAGE + (wave * 8 + 32 * i) * 128, b_cur, i * w_istride + wtap * a.IC * 2 + wkc * 128);
    };
    auto w_advance = [&]() __attribute__((always_inline)) {
        --w_left;
        b_cur = (w_left <= 0 || PS_DBG(2)) ? (int)0x80000000 : b_cur;
        const bool wr = wtap == 8;
        wtap = wr ? 0 : wtap + 1;
        wkc = wr ? (wkc + 1 == kpt ? 0 : wkc + 1) : wkc;
    };
    // slab DMA piece p of this wave: instruction jj covers slab rows 8 jj .. 8 jj + 7; every wave issues PS_SLAB_PIECES of them
    // (counted waits): the ones past the end repeat the last piece (the same bytes again)
    auto slab_piece = [&](int sbuf, int m0s, int kc, int p, bool valid) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // (recomputed here, not kept in registers across the K-loop)
        int jj = wave + 4 * p;
        jj = jj < nins ? jj : nins - 1;
        const int sr = jj * 8 + (ln >> 3);
        const int pix = m0s - (a.W + 1) + sr;
        const bool ok = valid & (sr < a.slab_rows) & ((unsigned)pix < (unsigned)a.in_pixels);
        const int v = ok ? pix * a.IC * 2 + kc * 128 + (((ln & 7) ^ swz128(sr)) << 4) : (int)0x80000000;
        dma16(rin, smem + 2 * WSTAGE + sbuf * slab_bytes + jj * 1024, v);
    };
    auto load_masks = [&](int m0s, unsigned (&fm)[MI]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MI; ++m) {
            const int mm = m0s + (wave * WTM) + m * 16 + frow;
            fm[m] = mm < a.M ? a.table[mm].mask : 0u;
        }
    };

    const int fswz = swz128(frow);
    const unsigned woff0 = frow * 128 + (((0 + fg) ^ fswz) << 4), woff1 = woff0 ^ 64u;  // (chunk ^ 4: the second 32-channel half)
    const int prow0 = wave * WTM + frow + (a.W + 1);
    const unsigned zb0 = zrow + (fg << 4);  // zero row; fragment m reads it through the pre-biased address zb0 - 2048 m
    const int sh0 = a.pshift[0], d1 = a.pshift[1] - a.pshift[0], d3 = a.pshift[3] - a.pshift[2];

    // ---- prologue: the first tile's first slab chunk, the first two weight tiles, its tap masks
    unsigned fmask[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) fmask[m] = 0u;
    if (ntiles > 0) {
        const int m0s = (item / ntn) * BM;
#pragma unroll
        for (int p = 0; p < PS_SLAB_PIECES; ++p) slab_piece(0, m0s, 0, p, true);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w_piece(s, i);
            w_advance();
        }
        load_masks(m0s, fmask);
    }
    GDL_STAMP(1);

    f32x4_t acc[NI][MI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int m = 0; m < MI; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 px0[MI], wf0[NI], px1[MI], wf1[NI];
#ifdef GDL_TIMING
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tqn = 0, tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tacc4 = 0, tsteps = 0;
#endif
    int slot = 0;              // ring slot of the current step's weight tile
    int sbuf = 0;              // slab buffer of the current chunk

    // one fragment read of a set: r < MI pixel fragment r, else weight fragment r - MI
    auto frag_read = [&](auto rc, uint4 (&px)[MI], uint4 (&wf)[NI], const unsigned (&pb)[MI], unsigned wb) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < MI)
            px[r] = lds_read16_asm_off<2048 * r>(pb[r]);
        else
            wf[r - MI] = lds_read16_asm_off<2048 * (r - MI)>(wb);
    };
    // fragment addresses of a tap: the slab row of fragment 0 shifted by the tap (16 rows = 2048 bytes per further fragment:
    // an instruction offset), or the zero row where the gather table says "padding"
    auto tap_addr = [&](unsigned (&pb)[MI], int sbuf_, int shift, int tap_) __attribute__((always_inline)) {
        const unsigned slab = slab_base + sbuf_ * slab_bytes;
        const int sr0 = prow0 + shift;
        const unsigned ad0 = slab + sr0 * 128 + ((fg ^ swz128(sr0)) << 4);
#pragma unroll
        for (int m = 0; m < MI; ++m) pb[m] = ((fmask[m] >> tap_) & 1u) ? ad0 : zb0 - 2048u * m;
    };

    bf16* __restrict__ gout = (bf16*)a.out;
    const bf16* __restrict__ gadd = (BWD && RICH) ? (const bf16*)a.addend : nullptr;
    const uint8_t* __restrict__ rbits = BWD ? a.relu_bits : nullptr;
    const bool bw = BWD && a.bw_y != nullptr, bw2 = RICH && bw && a.bw_y2 != nullptr;
    const bool fst = !BWD && (a.stats != nullptr || a.sacc.acc != nullptr);

    for (int t = 0; t < ntiles; ++t, item += bpx) {
        const int m0 = (item / ntn) * BM;
        const bool more = t + 1 < ntiles;
        const int m0n = ((item + bpx) / ntn) * BM, dm0 = m0n - m0;
        const int rbase = m0 + wave * WTM;
        // the prefetched slab chunk, the first two weight tiles and the masks are here; the stores of the tile before are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < MI; ++m) asm volatile("" : "+v"(fmask[m]));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t == 0) GDL_STAMP(2);
        int sh = sh0, scol = 0;
        unsigned pb[MI];
        tap_addr(pb, sbuf, sh, 0);
        {
            const unsigned wb = smem_base + slot * WSTAGE + woff0;
            static_for<0, NR>([&](auto rc) { frag_read(rc, px0, wf0, pb, wb); });
        }
        // one round's partner vectors (data gradient).  They come from HBM / the memory-side cache (forward activations last touched
        // a whole backward ago: ~2 000 clk), a round is ~600 clk of work: requested ONE round ahead -- the first version -- every
        // round stalled on them (a data gradient took 62 us alone where the bare kernel takes 51).  The common launch (ReLU bits +
        // one BatchNorm partner: conv2's data gradient, RICH = false) therefore requests ALL its rounds behind the tile's last
        // MFMAs (10 registers per round); the launches with an addend or a second partner (conv1's: 26 registers per round) keep
        // two rounds in flight.
        struct Pre {
            uint4 y[2];
            unsigned mk[2];
            uint4 g[RICH ? 2 : 1], y2[RICH ? 2 : 1];
        };
        constexpr int NRND = 2 * MI, PFD = RICH ? 2 : NRND, NPRE = RICH ? 3 : NRND;  // rounds, prefetch distance, register sets
        auto request = [&](auto rc, Pre& q) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value, m = r >> 1, h = r & 1;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int erow = ln >> 3, ec = ln & 7;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int mrow = rbase + m * 16 + erow + 8 * p;
                const size_t goff = (size_t)mrow * a.OC + n0 + h * 64 + ec * 8;
                q.y[p] = make_uint4(0u, 0u, 0u, 0u);
                q.mk[p] = 0xffu;
                if constexpr (RICH) q.g[p] = q.y2[p] = make_uint4(0u, 0u, 0u, 0u);
                if (mrow < a.M) {
                    if (rbits) q.mk[p] = rbits[goff >> 3];
                    if (bw) q.y[p] = *(const uint4*)((const bf16*)a.bw_y + goff);
                    if constexpr (RICH) {
                        if (gadd) q.g[p] = *(const uint4*)(gadd + goff);
                        if (bw2) q.y2[p] = *(const uint4*)((const bf16*)a.bw_y2 + goff);
                    }
                }
            }
        };
        Pre pre[NPRE];

        // ---- one K-step (kc = channel chunk, tap); LAST = the tile's last one (peeled: no next step's reads, round 0 of the data
        // gradient's partner vectors requested).  No run-time branch sits between the MFMAs -- they cost ~20 clk each there
        // (tools/micro/kstep.hip: 946 -> 1 315 clk per step with 19 of them): the weight pieces behind the end of the block's
        // stream carry an out-of-range offset instead of being skipped, the rare slab requests follow the MFMAs.
        int kc = 0, tap = 0;  // the current step
        unsigned pb1[MI];     // its second-half fragment addresses (pb ^ 64: chunk ^ 4)
#pragma unroll
        for (int m = 0; m < MI; ++m) pb1[m] = pb[m] ^ 64u;
        auto kstep = [&](auto lastc) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(lastc)::value;
            // slots of [A] behind the reads: Q_SC.. scalar bookkeeping of the next step (no order pinned: the scheduler spreads it
            // over the MFMAs), then the next step's fragment addresses (VALU), one piece per MFMA
            constexpr int Q_AD = NQ - MI - 1;
            int nsh = 0, nscol = 0, ntap = 0, nkc = 0, nsbuf = 0;
            // slab requests.  Two buffers: the chunk after this one -- of this tile or the first of the next -- at tap 0, into the
            // buffer every wave has left.  One buffer: at the chunk's last step, when its slab is dead in every wave; the next
            // chunk's is waited for at once (exposed once per chunk), the next tile's lands behind the epilogue.
            bool do_slab = false;
            int s_m0 = m0, s_kc = 0, s_buf = 0;
            unsigned pbn[MI], pbn1[MI];
            unsigned ad0n = 0;
            // [A]: MFMAs on F0; the reads of the second 32-channel half -> F1 between the first NR of them
            {
                const unsigned wb1 = smem_base + slot * WSTAGE + woff1;
                PS_T(tqn);
                lds_wait();  // F0 has landed
#ifdef GDL_TIMING_SPLIT
                if (tq0) tacc0 += tq1 - tq0, tacc1 += tq2 - tq1, tacc2 += tq3 - tq2, tacc3 += tq4 - tq3, tacc4 += tqn - tq4, ++tsteps;
                tq0 = tqn;
                asm volatile("" ::: "memory");
                PS_T(tq1);
#endif
                static_for<0, NQ>([&](auto qc) {
                    constexpr int q = decltype(qc)::value, n = q / MI, m = q % MI;
                    Mma<bf16>::run(wf0[n], px0[m], acc[n][m]);
                    if constexpr (q < NR) {
                        __builtin_amdgcn_sched_barrier(0);
                        frag_read(std::integral_constant<int, q>{}, px1, wf1, pb1, wb1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (q == NR) {
                        // the next step's tap shift / slab buffer (scalar)
                        // (selects, not branches)
                        const bool roww = scol == 2, chw = tap == 8;
                        nsh = chw ? sh0 : sh + (roww ? d3 : d1);
                        nscol = roww ? 0 : scol + 1;
                        ntap = chw ? 0 : tap + 1;
                        nkc = kc + (chw ? 1 : 0);
                        nsbuf = sbuf ^ ((chw && TWO) ? 1 : 0);
                        // the slab this step asks for (0 / 1 arithmetic: && / || / ?: on bools became branches between the MFMAs).
                        // Two buffers: ONE piece per step of the chunk after this one -- of this tile, or the first of the next --
                        // into the buffer every wave has left (tap 8 repeats piece 7: the same bytes); out of range when nothing follows.
                        // One buffer: at the chunk's last step, when its slab is dead in every wave, behind the MFMAs.
                        const int here = kc + 1 < kpt ? 1 : 0, morei = more ? 1 : 0;
                        const int nxt = TWO ? (here ^ 1) : (LAST ? 1 : 0);  // 1: the slab of the NEXT tile's first chunk
                        do_slab = TWO ? ((here | morei) & (PS_DBG(8) ? 0 : 1)) != 0 : (LAST ? more : chw);
                        s_m0 = m0 + nxt * dm0, s_kc = (nxt ^ 1) * (kc + 1), s_buf = TWO ? (sbuf ^ 1) : 0;
                    } else if constexpr (!LAST && q == Q_AD) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int sr0 = prow0 + nsh;
                        ad0n = slab_base + nsbuf * slab_bytes + sr0 * 128 + ((fg ^ swz128(sr0)) << 4);
                        asm volatile("" : "+v"(ad0n));  // (computed HERE, in the MFMAs' shadow: the optimiser sinks it to its use otherwise)
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (!LAST && q > Q_AD) {
                        constexpr int mm = q - Q_AD - 1;
                        if constexpr ((PS_EXP & 8) != 0)
                            pbn[mm] = pb[mm];
                        else if constexpr ((PS_EXP & 1) != 0)
                            pbn[mm] = ad0n;
                        else
                            pbn[mm] = ((fmask[mm] >> ntap) & 1u) ? ad0n : zb0 - 2048u * mm;
                        pbn1[mm] = pbn[mm] ^ 64u;
                        asm volatile("" : "+v"(pbn[mm]), "+v"(pbn1[mm]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
            // [B]: F1 has landed, this wave's pieces of the next weight tile (and the slab piece of the step before) too; behind
            // the barrier that holds for every wave
            __builtin_amdgcn_sched_barrier(0);
            PS_T(tq2);
            if constexpr ((PS_EXP & 4) != 0)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if constexpr ((PS_EXP & 4) == 0)
                if (!PS_DBG(4)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (t == 0 && kc == 0 && tap == 0) GDL_STAMP(3);
            PS_T(tq3);
            // [C]: MFMAs on F1; weight tile k + 2 -> this step's slot; the next step's first half -> F0; behind them (no order
            // pinned) the weight cursor and the slab decision
            const int nslot = slot ^ 1;
            const unsigned wbn = smem_base + nslot * WSTAGE + woff0;
            if constexpr (LAST && BWD)  // the first rounds' partner vectors: behind the last MFMAs
                static_for<0, PFD>([&](auto rc) { request(rc, pre[decltype(rc)::value % NPRE]); });
            static_for<0, NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value, n = q / MI, m = q % MI;
                Mma<bf16>::run(wf1[n], px1[m], acc[n][m]);
                if constexpr (q == 0 && TWO) {
                    __builtin_amdgcn_sched_barrier(0);
                    slab_piece(s_buf, s_m0, s_kc, tap < PS_SLAB_PIECES ? tap : PS_SLAB_PIECES - 1, do_slab);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (q >= 1 && q < 5) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((PS_EXP & 2) == 0) w_piece(slot, q - 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (q >= 5 && q - 5 < NR) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!LAST) frag_read(std::integral_constant<int, q - 5>{}, px0, wf0, pbn, wbn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            w_advance();
            __builtin_amdgcn_sched_barrier(0);
            PS_T(tq4);
            if constexpr (!TWO) {
                if (do_slab) {
#pragma unroll
                    for (int p = 0; p < PS_SLAB_PIECES; ++p) slab_piece(s_buf, s_m0, s_kc, p, true);
                    if constexpr (!LAST) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        static_for<0, NR>([&](auto rc) { frag_read(rc, px0, wf0, pbn, wbn); });  // (again, from the new slab)
                    }
                }
            }
            if constexpr (!LAST) {
#pragma unroll
                for (int m = 0; m < MI; ++m) pb[m] = pbn[m], pb1[m] = pbn1[m];
            }
            sh = nsh, scol = nscol, slot = nslot, sbuf = nsbuf, tap = ntap, kc = nkc;
        };
#pragma nounroll
        for (int s = 0; s < NS - 1; ++s) kstep(std::false_type{});
        kstep(std::true_type{});
        if (more) load_masks(m0n, fmask);  // (waited for at the top of the next tile)
        if (t == 0) GDL_STAMP(4);

        // ---- epilogue: 2 MI rounds of 16 pixels x 64 channels through the wave's own staging rows
        // lane mapping: rows (lane >> 3) + 8 p of a staged 16-pixel group, 16-byte chunk lane & 7 of its 64 channels
        int eln = lane;
        asm volatile("" : "+v"(eln));
        const int erow = eln >> 3, ec = eln & 7;
        const unsigned st_wr = stg + (eln & 15) * PS_PITCH + (eln >> 4) * 8, st_rd = stg + erow * PS_PITCH + ec * 16;
        constexpr int NS3 = BWD ? 8 : 1;
        float s1[2][8], s2[2][8], s3[2][NS3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[h][e] = s2[h][e] = 0.f;
#pragma unroll
            for (int e = 0; e < NS3; ++e) s3[h][e] = 0.f;
        }
        static_for<0, 2 * MI>([&](auto rc) {
            constexpr int r = decltype(rc)::value, m = r >> 1, h = r & 1;
            // accumulators -> staging rows: D[i][j], i = channel (lane >> 4) * 4 + reg, j = pixel lane & 15
            static_for<0, 4>([&](auto nc) {
                constexpr int nn = decltype(nc)::value, n = 4 * h + nn;
                const uint2 v = make_uint2(pack2bf(acc[n][m][0], acc[n][m][1]), pack2bf(acc[n][m][2], acc[n][m][3]));
                lds_write8_asm_off<nn * 32>(st_wr, v);
            });
            if constexpr (BWD && r + PFD < NRND) request(std::integral_constant<int, r + PFD>{}, pre[(r + PFD) % NPRE]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint4 q[2];
            q[0] = lds_read16_asm_off<0>(st_rd);
            q[1] = lds_read16_asm_off<8 * PS_PITCH>(st_rd);
            float mu[8], mu2[NS3];
            if constexpr (BWD) {
                if (bw) {
                    const int c0 = n0 + h * 64 + ec * 8;
                    *(float4*)&mu[0] = *(const float4*)(a.bw_mean + c0), *(float4*)&mu[4] = *(const float4*)(a.bw_mean + c0 + 4);
                    if (bw2) *(float4*)&mu2[0] = *(const float4*)(a.bw_mean2 + c0), *(float4*)&mu2[4] = *(const float4*)(a.bw_mean2 + c0 + 4);
                }
            }
            lds_wait();
            Pre& pq = pre[r % NPRE];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int mrow = rbase + m * 16 + erow + 8 * p;
                if (mrow >= a.M) continue;
                const size_t goff = (size_t)mrow * a.OC + n0 + h * 64 + ec * 8;
                uint4 v = q[p];
                float f[8];
                if constexpr (BWD) {
                    // what conv_epilogue does: + addend (rounded again), ReLU bits, then the sums of the value as stored
                    if (gadd) {
                        float g[8];
                        unpack16<bf16>(v, f);
                        unpack16<bf16>(pq.g[p], g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g[e];
                        v = pack16<bf16>(f);
                    }
                    if (rbits) {
                        const unsigned mk = pq.mk[p];
                        unpack16<bf16>(v, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = ((mk >> e) & 1u) ? f[e] : 0.f;
                        v = pack16<bf16>(f);
                    }
                    if (bw) {
                        float yv[8];
                        unpack16<bf16>(v, f);
                        unpack16<bf16>(pq.y[p], yv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s1[h][e] += f[e];
                            s2[h][e] += f[e] * (yv[e] - mu[e]);
                        }
                        if (bw2) {
                            unpack16<bf16>(pq.y2[p], yv);
#pragma unroll
                            for (int e = 0; e < NS3; ++e) s3[h][e] += f[e] * (yv[e] - mu2[e]);
                        }
                    }
                } else {
                    if (fst) {
                        unpack16<bf16>(v, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s1[h][e] += f[e];
                            s2[h][e] += f[e] * f[e];
                        }
                    }
                }
                *(uint4*)(gout + goff) = v;
            }
        });
        if (t == 0) GDL_STAMP(6);  // (behind the first tile's rounds)
        // this tile's sums -> the wave's LDS row.  The eight lanes that hold the same channels (lane ^ 8, ^ 16, ^ 32) reduce AND scatter:
        // every level halves the values a lane carries -- across the wave's halves one v_permlane32_swap of a pair leaves sum 1 in
        // the lower half and sum 2 in the upper, across rows v_permlane16_swap does the same for the two channel halves, inside a row
        // a select + DPP rotation for channels e / e + 4 -- 60 VALU instructions instead of the 224 of eight all-reduces per
        // value (and of the 96 ds_bpermute round trips of the first version: 5 100 clk of an 11 000 clk epilogue).  A lane ends with
        // the four sums (k = lane bit 5, half h = bit 4, channels 4 * bit 3 .. + 3 of its chunk) and adds them into its OWN 16 bytes
        // of the wave's row, [chunk 8][k 2][h 2][e 8] floats: one LDS read, one write, every lane.  Fixed order: bit-identical from
        // run to run.
        if (fst || bw) {
            auto scatter = [&](float (&lo)[2][8], float (&hi)[2][8], unsigned ad) __attribute__((always_inline)) {
                float r[2][8], q[8], res[4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {  // lower half of the wave: lo summed over lane ^ 32; upper half: hi
                        auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[h][e]), __float_as_uint(hi[h][e]), false, false);
                        r[h][e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                    }
#pragma unroll
                for (int e = 0; e < 8; ++e) {  // even rows: half 0 summed over lane ^ 16; odd rows: half 1
                    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0][e]), __float_as_uint(r[1][e]), false, false);
                    q[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                }
                const bool up = (eln & 8) != 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // lanes 0-7 of a row: channels e summed over lane ^ 8; lanes 8-15: channels e + 4
                    const float keep = up ? q[e + 4] : q[e], give = up ? q[e] : q[e + 4];
                    res[e] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                }
                f32x4_t v = lds_read_f4_asm(ad);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(v));
                v[0] += res[0], v[1] += res[1], v[2] += res[2], v[3] += res[3];
                lds_write_f4_asm(ad, v);
            };
            // the lane's 16 bytes: chunk ec = lane & 7, then (k, h, e >> 2) = lane bits 5, 4, 3
            scatter(s1, s2, statw + (eln & 7) * 128 + (eln >> 3) * 16);
            if constexpr (BWD && RICH) {
                if (bw2) {
                    // the second partner's sum, 16 values: halves across the wave's halves, channels e / e + 4 across rows, e / e + 2 inside
                    // a row: a lane ends with two, [chunk 8][h 2][e 8] floats behind the first 1 KiB
                    float r[8], q[4], res[2];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s3[0][e]), __float_as_uint(s3[1][e]), false, false);
                        r[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[e]), __float_as_uint(r[e + 4]), false, false);
                        q[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                    }
                    const bool up = (eln & 8) != 0;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float keep = up ? q[e + 2] : q[e], give = up ? q[e] : q[e + 2];
                        res[e] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                    }
                    const unsigned ad = statw + 1024 + (eln & 7) * 64 + (eln >> 3) * 8;
                    uint2 v;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(v));
                    v.x = __float_as_uint(__uint_as_float(v.x) + res[0]), v.y = __float_as_uint(__uint_as_float(v.y) + res[1]);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(v) : "memory");
                }
            }
        }
        if (t == 0) GDL_STAMP(7);  // (behind its fold)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int m = 0; m < MI; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // ---- the block's sums: the four waves' rows in fixed order -> one partial row (zeros outside this block's channels) or
    // the integer accumulators
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const float* sr = (const float*)(smem + tail + 4 * PS_STAGE);
    // a wave's row (1 536 B): [chunk 8][k 2][h 2][e 8] floats of sums 0 / 1, then [chunk 8][h 2][e 8] of sum 2; channel c = h * 64 + chunk * 8 + e
    auto fold4 = [&](int c, int w) {
        const int ch = (c >> 3) & 7, h = c >> 6, e = c & 7;
        const int o = w < 2 ? ch * 32 + w * 16 + h * 8 + e : 256 + ch * 16 + h * 8 + e;
        return ((sr[o] + sr[384 + o]) + sr[768 + o]) + sr[1152 + o];
    };
    if (fst) {
        if (a.sacc.acc) {
            if (ntiles > 0) bn_acc_add(a.sacc, n0 + (tid >> 1), tid & 1, fold4(tid >> 1, tid & 1));
        } else {
            for (int i = tid; i < a.OC * 2; i += 256) {
                const int c = (i >> 1) - n0, w = i & 1;
                const float s = (ntiles > 0 && c >= 0 && c < BN) ? fold4(c, w) : 0.f;
                st_agent(a.stats + (size_t)blockIdx.x * a.OC * 2 + i, s);
            }
        }
    }
    if (bw) {
        for (int i = tid; i < a.OC * 2; i += 256) {
            const int cg = i >> 1, c = cg - n0, w = i & 1;
            const bool mine = ntiles > 0 && c >= 0 && c < BN;
            const float t1 = mine ? fold4(c, 0) : 0.f, t2 = mine ? fold4(c, 1) : 0.f;
            st_agent(a.bw_partial + (size_t)blockIdx.x * a.OC * 2 + i, w ? t2 * a.bw_rstd[cg] : t1);
            if (bw2) {
                const float t3 = mine ? fold4(c, 2) : 0.f;
                st_agent(a.bw_partial2 + (size_t)blockIdx.x * a.OC * 2 + i, w ? t3 * a.bw_rstd2[cg] : t1);
            }
        }
    }
    GDL_STAMP(5);
#ifdef GDL_TIMING
    if (a.dbg && threadIdx.x == 0) {
        unsigned long long* d = a.dbg + ((size_t)(1 << 15) + blockIdx.x) * 8;
        d[0] = tacc0, d[1] = tacc1, d[2] = tacc2, d[3] = tacc3, d[4] = tacc4, d[5] = tsteps, d[6] = 0x5053ull;  // "PS": this kernel's split
    }
#endif
}
