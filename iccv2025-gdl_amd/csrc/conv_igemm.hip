// conv_igemm.hip -- implicit-GEMM convolution forward / data-gradient on MFMA (gfx950).
//
// Replaces nn.Conv2d forward and its input-gradient for the shapes the reference
// encoder uses (/root/reference/models/backbone.py:20-28 conv3x3 / conv1x1, and the
// 7x7/2 stem of :96-101 as a direct implicit GEMM over a zero-padded NHWC4 copy of the input).  GEMM view: M = N*OH*OW output pixels, Ngemm = OC
// output channels, Kgemm = R*S*IC; NHWC activations make every K-step a
// contiguous 128-byte slice of one input pixel (fixed tap, 64 bf16 / 32 f32
// channels) and the [OC][R][S][IC] weight layout makes the matching weight slice
// contiguous too, so both operands load as 16-byte chunks straight into the MFMA
// fragment order.
//
// Block = 256 threads (4 waves), tile BM pixels x BN channels, BK = 128 bytes.
// Two LDS stages filled by LDS-DMA (buffer_load ... lds): the gather's zero fill
// (padding, stride-2 parity, M tail) comes from the buffer descriptor's bounds
// check, the per-pixel offsets from a precomputed gather table (gather.h); one
// barrier per K-step.  LDS rows are 128 B with a 16-byte
// chunk XOR swizzle chunk ^= swz128(row) (below), conflict-free for the ds_read_b128
// fragment reads.  Operands are swapped (weights = MFMA "A", pixels = MFMA "B")
// so a lane ends up with 4 consecutive output channels of one pixel.  The
// epilogue stages the tile through LDS, then streams full rows: optional addend
// (dgrad accumulation), optional per-channel sum / sum-of-squares of the STORED
// values (BatchNorm statistics), 16-byte coalesced stores.
#include <stdlib.h>

#include <type_traits>

#include "bnacc.h"
#include "common.h"
#include "gather.h"
#include "ops.h"
#include "prof.h"

#include <hip/hip_ext.h>

namespace gdl {

struct ConvArgs {
    const void* in;            // gather source (NHWC)
    const void* wt;            // [OC][ntaps][IC]
    void* out;                 // NHWC rows of OC channels, row m = GEMM row m
    const void* addend;        // optional, like out
    const float* bias;         // optional [OC] float32, added with the addend (nn.Linear bias of the Swin GEMMs)
    void* gelu_out;            // optional, like out: out keeps the (biased) value u, gelu_out gets gelu(u) (Swin Mlp.fc1 + act)
    const void* gelu_u;        // optional (data gradient): the stored value is out * gelu'(gelu_u[row][c]) (backward of Mlp.act)
    const uint8_t* relu_bits;  // optional: one byte per 16-byte vector of out; the stored value is zeroed where its bit is 0
    float* stats;              // optional [mtiles][OC][2]
    BnAcc sacc;                // optional (forward): the same two sums added to 64-bit integer accumulators instead (bnacc.h)
    // optional (data gradient): BatchNorm-backward sums of the stored rows against one or two partner tensors (ops.h BwdStats)
    const void* bw_y;
    const float *bw_mean, *bw_rstd;
    float* bw_partial;
    const void* bw_y2;
    const float *bw_mean2, *bw_rstd2;
    float* bw_partial2;
    const GatherEntry* table;  // [M]
    int M, OC, IC, ntaps;
    int mtiles;  // ceil(M/BM)
    unsigned in_bytes, wt_bytes;
    int delta[9];
    // slab kernel only (3x3, stride 1): image width, per-tap pixel shift, source pixel count, slab rows
    int W, in_pixels, slab_rows;
    int single_slab;  // 1: one slab buffer (the next channel chunk's slab is loaded at the chunk boundary, exposed)
    int pshift[9];
    // permuted stride-2 data gradient (gather.h): output pixel of each GEMM row (-1: padding row)
    const int* orow;
    const unsigned* tile_taps;  // per M-tile: OR of its rows' tap masks
    const int* tile_order;      // launch slot -> M-tile (gather.hip: the classes of an image group meet in one XCD's L2)
    // ... with the 1x1 stride-2 data gradient of the block's downsample branch folded in (flat kernel): the (even, even)
    // input pixels -- the class whose tiles carry the centre tap -- get one more "tap" that gathers the SAME dy pixel as
    // the centre tap from a second gradient tensor (in2, [.][IC]) and multiplies it with a second matrix (wt2, [OC][IC])
    const void* in2;
    const void* wt2;
    unsigned in2_bytes, wt2_bytes;
    // direct bf16 stem (layout.hip): a 128-byte K-step is TWO 64-byte filter rows -- chunks 0..3 of a tile row
    // come from off0 + delta[tap], chunks 4..7 from off0 + delta_hi[tap]
    int split;
    int delta_hi[9];
    // plain GEMM (1x1, stride 1, no padding: the Swin Linears): GEMM row m gathers source row m -- the flat kernel computes the
    // offset instead of fetching table[m] (one dependent HBM round trip less at the head of a block whose K-loop is 2-6 steps)
    int plain;
    // split-K of the slab kernel (round 3; the layers whose tile count leaves CUs idle -- layer 4 of both encoders, the audio
    // layer 3): ksplit > 1 blocks share an output tile, block `split` multiplies channel chunks [split, split + 1) * kpt / ksplit
    // and stores its fp32 accumulators to split_ws[split][M][OC]; splitk_finish_kernel folds them in fixed order and does what
    // the epilogue would have done (rounding, addend, ReLU bits, statistics)
    int ksplit;
    float* split_ws;
    // row-slab stem (conv_stem_rows_kernel): GEMM row g = stage*64 + i, stage = (image row R, segment sg): output pixel
    // R*seg_Q + ow0 + i with ow0 = min(64*sg, seg_Q - 64); rows i < 64*sg - ow0 repeat the previous stage: not stored
    int seg_Q, seg_nseg, seg_stages;
    double flops;  // algorithmic work of the launch (measurement tap)
    double hbm_bytes;  // its algorithmic HBM bytes: both activation tensors once + the filter (combined roofline, prof.h)
#ifdef GDL_TIMING
    unsigned long long* dbg;  // [block][8] s_memtime stamps of wave 0 (tools/timing_probe.py)
    int dbg_mode;             // conv3x3_pslab_kernel, timing experiments (WRONG results): GDL_PSLAB_DBG bits, see conv_pslab.h
#endif
};

#ifdef GDL_TIMING
unsigned long long* g_timing_buf = nullptr;
#define GDL_STAMP(i)                                                                            \
    do {                                                                                        \
        if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GDL_STAMP(i) \
    do {             \
    } while (0)
#endif

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// 16 bytes per lane, global -> LDS without passing through VGPRs (buffer_load_dwordx4 ... lds).
// LDS destination = wave-uniform `lds_row_base` + lane*16; `voffset` is the per-lane byte offset
// into the buffer (out of range -> zeros).  The builtin only exists in the device pass.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_row_base, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_row_base, 16, voffset, 0, 0,
                                             0);
#else
    (void)rsrc;
    (void)lds_row_base;
    (void)voffset;
#endif
}

template <typename T>
struct Mma;
template <>
struct Mma<bf16> {
    // one 16-byte chunk per lane = 8 bf16 = the whole K=32 operand of one 16x16x32 MFMA
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                    0, 0);
    }
};
template <>
struct Mma<float> {
    // one 16-byte chunk per lane = 4 floats = the operands of four 16x16x4 MFMAs
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// 16-byte-chunk XOR swizzle of a 128-byte LDS row: logical chunk c of row r sits at chunk c ^ swz128(r).
// ds_read_b128 serves a wave in four 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32) and a group
// is conflict-free when its 16 lanes hit 16 different 16-byte slots of the 256-byte bank row (= two LDS rows).  A fragment
// read (lane -> row r0 + (lane & 15), chunk (lane >> 4) + 4 kk) therefore needs, among the 8 rows of one parity, distinct
// physical chunks for a mix of logical chunks c (4 rows) and c + 1 (4 rows), c even.  The term below only flips chunk bits
// 1-2 (values 0, 2, 4, 6, consecutive over consecutive same-parity rows), so c-rows stay on even and (c+1)-rows on odd chunks
// and each set of four is distinct FOR EVERY r0 -- the tap-shifted slab reads of the 3x3 kernels start at arbitrary rows.
// (Rounds 1-2 used (r >> 1) & 7: conflict-free only for r0 = 0 mod 4, 1.75x the LDS cycles averaged over the shifts; that
// was the 42-44 % SQ_LDS_BANK_CONFLICT of the 64-channel kernel and the 22 % of the slab kernel.)
__device__ __forceinline__ int swz128(int row) { return ((row >> 1) & 3) << 1; }

// LDS fragment read hidden from the compiler: hipcc waits vmcnt(0) before any ds_read it can see
// while an LDS-DMA is in flight (it cannot prove the ring slots disjoint), which would serialise
// the whole pipeline.  The data is valid only after lds_wait() (form (iii) of the guide's inline-asm
// rules: "=v" loads, one wait-only statement, sched_barrier).
__device__ __forceinline__ uint4 lds_read16_asm(unsigned addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// same with a compile-time byte offset folded into the instruction (no VALU for base + constant)
template <int OFF>
__device__ __forceinline__ uint4 lds_read16_asm_off(unsigned addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// The sched_barrier in front pins the MFMAs that precede the wait in program order (register-only
// instructions may otherwise sink below an asm statement), the one behind pins the consumers.
__device__ __forceinline__ void lds_wait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// counted form: returns once at most N of the LDS reads issued so far are still outstanding (LDS
// returns in order; a stray scalar load in the count only makes the wait longer, never shorter)
template <int N>
__device__ __forceinline__ void lds_wait_n() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

// compile-time loop: f(std::integral_constant<int, Q>{}) for Q in [Q0, QEND) -- instruction offsets, register-array indices and the
// slots of a hand-interleaved MFMA / LDS / DMA stream need constants
template <int Q, int QEND, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (Q < QEND) {
        f(std::integral_constant<int, Q>{});
        static_for<Q + 1, QEND>(f);
    }
}

template <int BM, int BN, typename T>
struct ConvSmem {
    static constexpr int STAGE = (BM + BN) * 128;
    static constexpr int NSTAGE = 2;  // LDS ring: one stage being consumed, NSTAGE-1 in flight
    static constexpr int MAIN = NSTAGE * STAGE;
    static constexpr int PITCH = BN * (int)sizeof(T) + 16;
    static constexpr int CS = BM * PITCH;
    static constexpr int RED = 8 * BN * 2 * 4;  // up to 8 waves
    static constexpr int BYTES = (MAIN > CS + RED) ? MAIN : (CS + RED);
};

// ---- epilogue shared by the kernels below: accumulators -> LDS tile [BM][BN] of T -> full rows.
// Must be entered after every wave is done with the main-loop LDS contents (barrier) and with no
// LDS-DMA in flight.
// PRE: the forward's bias and addend vectors are requested before the staging (the flat kernel's forward instantiations only --
// the Swin Linears; in the slab / persistent forward kernels of the ResNets the prefetch registers cost more than they return:
// 5.81 vs 5.69 ms when every instantiation had them)
// F32ST (round 4; flat data-gradient kernels whose LDS holds a whole fp32 tile: the BN = 64 tiles, i.e. every gradient of a
// 64-channel tensor): when the launch carries BatchNorm-backward sums, the accumulators are staged as fp32 and the sums are
// taken from the values BEFORE their rounding to bf16.  A BatchNorm parameter gradient of layer 1 is the sum of 602 112 signed
// gradients per channel that nearly cancel; summed from the stored bf16 values, the rounding errors' random walk alone was
// ~0.1 of it (tests/test_step_gpu.py::test_full_size_bf16_against_fp64).  The stored tensor is rounded once (accumulator +
// addend -> bf16) instead of twice.
template <typename T, int BM, int BN, int WM, int WN, bool BWD = false, bool PRE = false, bool F32ST = false>
__device__ __forceinline__ void conv_epilogue(f32x4_t (&acc)[BN / WN / 16][BM / WM / 16], unsigned char* smem,
                                              const ConvArgs& a, int m0, int n0, int mtile, int ntile = 0) {
    constexpr int NT = WM * WN * 64;  // threads of the block (256 or 512)
    using SM = ConvSmem<BM, BN, T>;
    constexpr int EPC = TT<T>::EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    // output row of every pass of the store loop below, fetched now so that the (permuted data
    // gradient's) row-index loads overlap the staging through LDS; -1 = nothing to store
    constexpr int CH = BN * (int)sizeof(T) / 16;  // 16-byte chunks per tile row
    constexpr int RPI = NT / CH;                   // rows per pass
    constexpr int NPASS = BM / RPI;
    const int ec = tid % CH, er0 = tid / CH;
    int om[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int m = m0 + er0 + p * RPI;
        if (a.seg_Q) {
            const int stg = m >> 6, i = m & 63;
            const int R = stg / a.seg_nseg, sg = stg - R * a.seg_nseg;
            const int ow0 = min(64 * sg, a.seg_Q - 64);
            om[p] = (stg < a.seg_stages && i >= 64 * sg - ow0) ? R * a.seg_Q + ow0 + i : -1;
        } else {
            om[p] = m < a.M ? (a.orow ? a.orow[m] : m) : -1;
        }
    }
    // BatchNorm-backward sums (ops.h BwdStats): the partner tensors' vectors of every pass are requested NOW, so that their
    // latency (HBM: the partner is a forward activation last touched a whole backward ago) runs under the staging of the tile
    // through LDS; loaded inside the store loop they cost the slab data gradients +19 us per launch (59 -> 78 us alone)
    // (BWD: only the data-gradient instantiations carry this code and its registers)
    const bool bw = BWD && a.bw_y != nullptr, bw2 = bw && a.bw_y2 != nullptr;
    const bool gl = BWD && a.gelu_u != nullptr;  // (excludes bw: the host checks) -- the GELU input rides in the partner registers
    // (the SECOND partner -- three launches per encoder -- is read inside the store loop: its NPASS vectors in registers on top
    // of the first partner's made the 192-row data gradient spill 144 bytes per lane in every launch)
    uint4 byq[BWD ? NPASS : 1];
    unsigned mkq[BWD ? NPASS : 1];
    if constexpr (BWD) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            mkq[p] = 0xffu;
            if (a.relu_bits && om[p] >= 0) mkq[p] = a.relu_bits[((size_t)om[p] * a.OC + n0 + ec * EPC) / EPC];
        }
    }
    if constexpr (BWD) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            byq[p] = make_uint4(0u, 0u, 0u, 0u);
            if (gl && om[p] >= 0) byq[p] = *(const uint4*)((const T*)a.gelu_u + (size_t)om[p] * a.OC + n0 + ec * EPC);
            if (bw && om[p] >= 0) {
                const size_t goff = (size_t)om[p] * a.OC + n0 + ec * EPC;
                byq[p] = *(const uint4*)((const T*)a.bw_y + goff);
            }
        }
    }
    // The forward's bias (the same EPC channels in every pass) and addend vectors are requested here as well: a load inside the
    // store loop makes the compiler wait for vmcnt(0) before its first use -- loads and stores share the counter, so every pass
    // then waits for the PREVIOUS pass's stores to be acknowledged (~1 000 clk each; tools/timing_probe.py: 8 200 clk of epilogue
    // for the eight passes of a 128 x 128 tile of the Swin Linears)
    T* __restrict__ gout = (T*)a.out;
    const T* __restrict__ gadd = (const T*)a.addend;
    float biasv[PRE ? EPC : 1];
    uint4 gaq[PRE ? NPASS : 1];
    if constexpr (PRE) {
        if (a.bias) {
#pragma unroll
            for (int e4 = 0; e4 < EPC / 4; ++e4) {
                const float4 b = *(const float4*)(a.bias + n0 + ec * EPC + 4 * e4);
                biasv[4 * e4] = b.x, biasv[4 * e4 + 1] = b.y, biasv[4 * e4 + 2] = b.z, biasv[4 * e4 + 3] = b.w;
            }
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            gaq[p] = make_uint4(0u, 0u, 0u, 0u);
            if (gadd && om[p] >= 0) gaq[p] = *(const uint4*)(gadd + (size_t)om[p] * a.OC + n0 + ec * EPC);
        }
    }
    // accumulators -> LDS tile [BM][BN] of T.
    // D[i][j]: i = channel = (lane>>4)*4 + reg, j = pixel = lane&15.
    unsigned char* Cs = smem;
    constexpr int PITCH32 = BN * 4 + 16;  // F32ST: rows of BN floats
    static_assert(!F32ST || (BWD && sizeof(T) == 2 && BM * PITCH32 + SM::RED <= SM::BYTES), "F32ST: the fp32 tile must fit");
    const bool f32st = F32ST && bw;
    if (f32st) {
        const int px_row = wm * WTM + (lane & 15);
        const int ch = wn * WTN + (lane >> 4) * 4;
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int m = 0; m < MI; ++m)
                *(float4*)(Cs + (px_row + m * 16) * PITCH32 + (ch + n * 16) * 4) =
                    make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
    } else {
        const int px_row = wm * WTM + (lane & 15);
        const int ch = wn * WTN + (lane >> 4) * 4;
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                unsigned char* p = Cs + (px_row + m * 16) * SM::PITCH + (ch + n * 16) * (int)sizeof(T);
                if (sizeof(T) == 2) {
                    *(uint2*)p = make_uint2(pack2bf(acc[n][m][0], acc[n][m][1]), pack2bf(acc[n][m][2], acc[n][m][3]));
                } else {
                    *(float4*)p = make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
                }
            }
    }
    __syncthreads();
    float ssum[EPC], ssq[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) ssum[e] = ssq[e] = 0.f;
    // BatchNorm-backward sums of the stored rows against their partner tensor(s) (ops.h BwdStats): per-channel constants of this
    // thread's EPC channels, three accumulators
    float bmu[EPC], bmu2[EPC], bs1[EPC], bs2[EPC], bs3[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) bmu[e] = bmu2[e] = bs1[e] = bs2[e] = bs3[e] = 0.f;
    if (bw) {
        const int c0 = n0 + ec * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            bmu[e] = a.bw_mean[c0 + e];
            if (bw2) bmu2[e] = a.bw_mean2[c0 + e];
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (om[p] < 0) continue;
        const int row = er0 + p * RPI;
        const size_t goff = (size_t)om[p] * a.OC + n0 + ec * EPC;
        if constexpr (F32ST) {
            if (f32st) {  // fp32 row: addend, ReLU mask and the sums on the unrounded values, one rounding at the store
                float f[EPC], yv[EPC];
                *(float4*)&f[0] = *(const float4*)(Cs + row * PITCH32 + ec * 32);
                *(float4*)&f[4] = *(const float4*)(Cs + row * PITCH32 + ec * 32 + 16);
                if (gadd) {
                    float g[EPC];
                    unpack16<T>(*(const uint4*)(gadd + goff), g);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] += g[e];
                }
                if (a.relu_bits) {
                    const unsigned mk = mkq[p];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] = ((mk >> e) & 1u) ? f[e] : 0.f;
                }
                unpack16<T>(byq[p], yv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    bs1[e] += f[e];
                    bs2[e] += f[e] * (yv[e] - bmu[e]);
                }
                if (bw2) {
                    float y2[EPC];
                    unpack16<T>(*(const uint4*)((const T*)a.bw_y2 + goff), y2);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) bs3[e] += f[e] * (y2[e] - bmu2[e]);
                }
                *(uint4*)(gout + goff) = pack16<T>(f);
                continue;
            }
        }
        uint4 v = *(const uint4*)(Cs + row * SM::PITCH + ec * 16);
        if (gadd || a.bias) {
            float f[EPC];
            unpack16<T>(v, f);
            if (a.bias) {
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] += biasv[e];
                } else {
#pragma unroll
                    for (int e4 = 0; e4 < EPC / 4; ++e4) {
                        const float4 b = *(const float4*)(a.bias + n0 + ec * EPC + 4 * e4);
                        f[4 * e4] += b.x, f[4 * e4 + 1] += b.y, f[4 * e4 + 2] += b.z, f[4 * e4 + 3] += b.w;
                    }
                }
            }
            if (gadd) {
                float g[EPC];
                uint4 w;
                if constexpr (PRE)
                    w = gaq[p];
                else
                    w = *(const uint4*)(gadd + goff);
                unpack16<T>(w, g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] += g[e];
            }
            v = pack16<T>(f);
        }
        if (a.relu_bits) {  // ReLU backward of the tensor this is the gradient of, folded into the store
            unsigned mk;
            if constexpr (BWD)
                mk = mkq[p];
            else
                mk = a.relu_bits[goff / EPC];
            float f[EPC];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = ((mk >> e) & 1u) ? f[e] : 0.f;
            v = pack16<T>(f);
        }
        if constexpr (BWD) {
            if (gl) {  // d gelu(u) / du times the gradient as a separate pass would have read it back (rounded to T once already)
                float f[EPC], uu[EPC];
                unpack16<T>(v, f);
                unpack16<T>(byq[p], uu);
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] = f[e] * gelu_grad<T>(uu[e]);
                v = pack16<T>(f);
            }
        }
        if (a.stats || a.sacc.acc) {
            float f[EPC];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                ssum[e] += f[e];
                ssq[e] += f[e] * f[e];
            }
        }
        if (bw) {
            float f[EPC], yv[EPC];
            unpack16<T>(v, f);  // the value as it will be read back (a ReLU mask came through relu_bits above)
            unpack16<T>(byq[BWD ? p : 0], yv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                bs1[e] += f[e];
                bs2[e] += f[e] * (yv[e] - bmu[e]);  // (x rstd once per channel, below)
            }
            if (bw2) {
                float y2[EPC];
                unpack16<T>(*(const uint4*)((const T*)a.bw_y2 + goff), y2);
#pragma unroll
                for (int e = 0; e < EPC; ++e) bs3[e] += f[e] * (y2[e] - bmu2[e]);
            }
        }
        *(uint4*)(gout + goff) = v;
        if (a.gelu_out) {  // exact GELU of the value as stored
            float f[EPC];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = gelu_val<T>(f[e]);
            *(uint4*)((T*)a.gelu_out + goff) = pack16<T>(f);
        }
    }
    // per-channel sums of the block's rows -> one partial row [OC][2] per M-tile: lanes with equal (lane % CH) hold the same
    // channels: fold them, then fold the waves in fixed order (deterministic)
    auto tile_sums = [&](float (&s1)[EPC], float (&s2)[EPC], float* dst) {
        float* red = (float*)(smem + (f32st ? BM * PITCH32 : SM::CS));  // [waves][BN][2]
        if constexpr (CH == 8 && EPC == 8) {
            // Round 6 (from conv_pslab.h; the 64-channel-wide bf16 tiles: every stride-2 / 1x1 launch): the eight row-lanes that hold
            // the same channels reduce AND scatter -- the wave's halves split the two sums (v_permlane32_swap of the pair), the rows
            // channels e / e + 4 (v_permlane16_swap), a row's halves e / e + 2 (select + DPP rotation): 30 VALU instructions for what
            // took 48 ds_bpermute round trips per call; a lane ends with two values of the wave's row.
            float r[8], q[4], res[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1[e]), __float_as_uint(s2[e]), false, false);
                r[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[e]), __float_as_uint(r[e + 4]), false, false);
                q[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
            }
            const bool up = (lane & 8) != 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float keep = up ? q[e + 2] : q[e], give = up ? q[e] : q[e + 2];
                res[e] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
            }
            // sum k = lane bit 5, channel = 8 * (lane & 7) + 4 * bit 4 + 2 * bit 3 (+ 0 / 1)
            const int c = (lane & 7) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2, k = (lane >> 5) & 1;
            red[(wave * BN + c) * 2 + k] = res[0];
            red[(wave * BN + c + 1) * 2 + k] = res[1];
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                for (int msk = CH; msk < 64; msk <<= 1) {
                    s1[e] += __shfl_xor(s1[e], msk);
                    s2[e] += __shfl_xor(s2[e], msk);
                }
            }
            if (CH >= 64 || lane < CH) {
                // when CH < 64 every wave covers all CH chunks; lane < CH holds chunk `lane`
                const int c = (lane % CH) * EPC;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    red[(wave * BN + c + e) * 2 + 0] = s1[e];
                    red[(wave * BN + c + e) * 2 + 1] = s2[e];
                }
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int c = tid >> 1, w = tid & 1;
            float s = ((red[(0 * BN + c) * 2 + w] + red[(1 * BN + c) * 2 + w]) + red[(2 * BN + c) * 2 + w]) +
                      red[(3 * BN + c) * 2 + w];
            if constexpr (NT == 512)
                s += ((red[(4 * BN + c) * 2 + w] + red[(5 * BN + c) * 2 + w]) + red[(6 * BN + c) * 2 + w]) +
                     red[(7 * BN + c) * 2 + w];
            if (dst)
                st_agent(dst + ((size_t)mtile * a.OC + n0 + c) * 2 + w, s);
            else
                bn_acc_add(a.sacc, n0 + c, w, s);
        }
    };
    if (a.stats || a.sacc.acc) {
        tile_sums(ssum, ssq, a.sacc.acc ? nullptr : a.stats);
    }
    if (bw) {
        float k1[EPC];
        const int c0 = n0 + ec * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            k1[e] = bs1[e];  // (tile_sums folds its arguments in place: the second BatchNorm needs them again)
            bs2[e] *= a.bw_rstd[c0 + e];
            if (bw2) bs3[e] *= a.bw_rstd2[c0 + e];
        }
        tile_sums(bs1, bs2, a.bw_partial);
        if (bw2) {
            __syncthreads();  // the scratch rows are reused
            tile_sums(k1, bs3, a.bw_partial2);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using SM = ConvSmem<BM, BN, T>;
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;  // elements per K-step
    constexpr int AROWS = BM / 32, BROWS = BN / 32;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    static_assert(WM * WN == 4, "4 waves");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;

    // ---- tile mapping: consecutive M-tiles stay on one XCD (block b runs on XCD b%8), all
    // N-tiles of an M-tile are neighbours on that XCD, so the input halo and the A panel hit L2
    const int ntn = a.OC / BN;
    const int L = xcd_linear(a.mtiles * ntn);  // (common.h: every XCD gets an eighth of the tiles)
    if (L < 0) return;
    const int slot = L / ntn, ntile = L - slot * ntn;
    const int mtile = a.tile_order ? a.tile_order[slot] : slot;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // ---- per-thread gather bookkeeping: rows row0 + 32*i of the tile, one 16-byte chunk each.
    // Source addresses are 32-bit byte offsets into buffer descriptors: an invalid tap (padding,
    // stride-2 parity, M tail) gets an offset beyond the descriptor and the hardware returns zeros.
    // Tiles go global -> LDS directly (buffer_load ... lds, no VGPR staging, no ds_write): one wave
    // instruction fills 8 consecutive 128-byte rows, lane L -> row L>>3, physical chunk L&7, so the
    // XOR swizzle is applied to the SOURCE chunk each lane fetches.
    const int pchunk = tid & 7, row0 = tid >> 3;
    const int schunk = pchunk ^ swz128(row0);  // (row0 + 32*i)>>1 & 7 is the same for every i
    int a_off[AROWS];
    unsigned a_mask[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + row0 + 32 * i;
        GatherEntry e;
        e.off0 = 0;
        e.mask = 0;
        if (a.plain) {
            if (m < a.M) e.off0 = m * a.IC * (int)sizeof(T), e.mask = 1u;
        } else if (m < a.M) {
            e = a.table[m];
        }
        a_off[i] = e.off0 + (a.split ? (schunk & 3) : schunk) * 16;
        a_mask[i] = e.mask;
    }
    const bool a_hi = a.split && (schunk >> 2);  // this lane's chunk belongs to the second filter row
    const int esz = (int)sizeof(T);
    int b_off[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_off[i] = (n0 + row0 + 32 * i) * a.ntaps * a.IC * esz + schunk * 16;
    int b_off2[BROWS];  // rows of the folded downsample matrix [OC][IC]
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_off2[i] = (n0 + row0 + 32 * i) * a.IC * esz + schunk * 16;
    const int kpt = a.IC / BKE;  // K-steps per tap
    // taps this tile multiplies: all of them, or (class-pure tiles of a permuted stride-2 data gradient)
    // only those that are valid for at least one of its rows
    unsigned tapset = a.tile_taps ? a.tile_taps[mtile] : (1u << a.ntaps) - 1u;
    // folded downsample branch: pseudo-tap number ntaps (= 9) wherever the centre tap (4) is present
    const bool ds = MODE == MODE_DGRAD && a.in2 != nullptr;
    const int ntaps_all = ds ? a.ntaps + 1 : a.ntaps;
    if (ds && (tapset & 16u)) tapset |= 1u << a.ntaps;
    const int nk = __popc(tapset) * kpt;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rin2 = __builtin_amdgcn_make_buffer_rsrc((void*)(ds ? a.in2 : a.in), 0, ds ? a.in2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt2 = __builtin_amdgcn_make_buffer_rsrc((void*)(ds ? a.wt2 : a.wt), 0, ds ? a.wt2_bytes : 0u, 0x00020000);
    const int wrow = (tid >> 6) * 8;  // first tile row this wave's DMA instruction covers (plus 32*i)

    int ld_tap = tapset ? __builtin_ctz(tapset) : ntaps_all, ld_kc = 0;  // (tap, channel chunk) of the NEXT tile to load
    // Pixel shift of that tap, fetched (scalar loads from the kernel arguments) when ld_tap changes, i.e. one tile
    // ahead of its use.  Indexing a.delta[] / a.delta_hi[] by lane inside load_tile made it a VECTOR load whose
    // latency sat between the barrier and the DMA issue of every K-step.
    // (the folded downsample tap gathers where the centre tap does)
    auto tap_src = [&](int t) { return t < a.ntaps ? t : (t == a.ntaps && ds ? 4 : 0); };
    int d_lo = a.delta[tap_src(ld_tap)], d_hi = a.split ? a.delta_hi[tap_src(ld_tap)] : 0;
    // Every call issues exactly AROWS+BROWS DMA instructions per wave (the counted s_waitcnt below
    // relies on it); past the last K-step they are all out of range: no memory traffic, zeros into
    // a ring slot nobody reads.
    auto load_tile = [&](int buf) {
        const bool live = ld_tap < ntaps_all;
        const bool x2 = ds && ld_tap == a.ntaps;  // the folded downsample tap (uniform)
        const int tap = live ? tap_src(ld_tap) : 0;
        const int ua = (a_hi ? d_hi : d_lo) + ld_kc * 128;  // uniform unless split
        const int ub = x2 ? ld_kc * 128 : (tap * a.IC) * esz + ld_kc * 128;  // uniform
        unsigned char* As = smem + buf * SM::STAGE;
        unsigned char* Bs = As + BM * 128;
        const __amdgpu_buffer_rsrc_t ra = x2 ? rin2 : rin, rb = x2 ? rwt2 : rwt;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int v = (live && ((a_mask[i] >> tap) & 1u)) ? a_off[i] + ua : (int)0x80000000;
            dma16(ra, As + (wrow + 32 * i) * 128, v);
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
            dma16(rb, Bs + (wrow + 32 * i) * 128, live ? (x2 ? b_off2[i] : b_off[i]) + ub : (int)0x80000000);
        if (++ld_kc == kpt) {
            ld_kc = 0;
            const unsigned rest = ld_tap + 1 < 32 ? tapset >> (ld_tap + 1) : 0u;
            ld_tap = rest ? ld_tap + 1 + __builtin_ctz(rest) : ntaps_all;  // next tap of the set
            const int nt = tap_src(ld_tap);
            d_lo = a.delta[nt];
            if (a.split) d_hi = a.delta_hi[nt];
        }
    };

    f32x4_t acc[NI][MI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int m = 0; m < MI; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = tilebase + (lane&15), logical chunk = kk*4 + (lane>>4)
    const int frow = lane & 15, fg = lane >> 4, fswz = swz128(frow);
    const int off_kk0 = frow * 128 + (((0 + fg) ^ fswz) << 4);
    const int off_kk1 = frow * 128 + (((4 + fg) ^ fswz) << 4);

    // ---- main loop: NSTAGE-deep LDS ring, loads NSTAGE-1 K-steps ahead of the MFMAs.
    //   wait (counted vmcnt: only the OLDEST stage must have landed) -> barrier (every wave's part of
    //   that stage landed, every wave is done reading the slot about to be refilled) -> issue the
    //   stage NSTAGE-1 ahead -> MFMAs on the landed stage.  One barrier per K-step, no vmcnt(0).
    const unsigned smem_base = lds_addr(smem);
    constexpr int NST = SM::NSTAGE;
    constexpr int LPS = AROWS + BROWS;  // DMA instructions per stage per wave
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) load_tile(s);
    // a two-step K-loop (K = 128: the stage-1 Swin Linears) has both stages requested up front -- the ring has the room, and the
    // second request would otherwise only go out once the first has landed: two HBM latencies in series per block
    const bool pre2 = NST == 2 && nk == 2;
    if (pre2) load_tile(1);
    int buf = 0, ldbuf = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned As = smem_base + buf * SM::STAGE + (wm * WTM) * 128;
        const unsigned Bs = smem_base + buf * SM::STAGE + BM * 128 + (wn * WTN) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = kk ? off_kk1 : off_kk0;
            uint4 px[MI], wf[NI];
#pragma unroll
            for (int m = 0; m < MI; ++m) px[m] = lds_read16_asm(As + m * 16 * 128 + off);
#pragma unroll
            for (int n = 0; n < NI; ++n) wf[n] = lds_read16_asm(Bs + n * 16 * 128 + off);
            if (kk == 0) {
                // the next stage's DMA is issued while the first half's fragment reads are in flight (issuing it
                // blocks the wave for about as long as the reads take; see the slab kernel)
                asm volatile("" ::: "memory");
                if (!pre2 && (NST > 2 || kt + 1 < nk)) load_tile(ldbuf);  // deeper rings need the constant DMA count per K-step
                asm volatile("" ::: "memory");
            }
            lds_wait();
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int m = 0; m < MI; ++m) Mma<T>::run(wf[n], px[m], acc[n][m]);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        ldbuf = ldbuf + 1 == NST ? 0 : ldbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the (empty) tail stages before LDS is reused
    __syncthreads();

    // (fp32 staging of the data gradient where the whole fp32 tile fits the main loop's LDS: the 64-channel-wide tiles)
    constexpr bool F32ST = MODE == MODE_DGRAD && sizeof(T) == 2 && BM * (BN * 4 + 16) + SM::RED <= SM::BYTES;
    conv_epilogue<T, BM, BN, WM, WN, MODE == MODE_DGRAD, MODE == MODE_FWD, F32ST>(acc, smem, a, m0, n0, mtile, ntile);
}

// =====================================================================================================
// Row-slab stem forward (bf16): the 7x7 stride-2 convolution over the padded NHWC4 input, 256 GEMM rows per block =
// four stages of 64 consecutive pixels of one output row each (see stem_wgrad_rows_kernel in conv_wgrad.hip for the
// geometry).  Wave w owns stage 4*tile + w: its A operand is the stage's 8 padded-input rows (8 contiguous runs of
// 1 072 B at an LDS pitch of 1 088 B, 9 DMA pieces, loaded and waited for by the wave itself), the B operand the
// whole 64 x (4 taps x 64) weight matrix (32 KB, loaded once per block).  The flat kernel moved 160 pieces per 256
// rows through the L2->LDS path, this one 68, with no K-loop barrier; every fragment address is ONE per-lane base
// plus an instruction offset: A fragment (tap t, half kk = filter row 2t+kk, pixel fragment m) sits at
// (2t+kk)*1088 + m*256 from  slab + pixel*16 + kgroup*16.
// =====================================================================================================
constexpr int SRF_PITCH = 1088, SRF_SLAB = 9 * 1024, SRF_W = 4 * 64 * 128;
template <int T4, int KK, int M4>
__device__ __forceinline__ uint4 srf_a(unsigned base) { return lds_read16_asm_off<(2 * T4 + KK) * SRF_PITCH + M4 * 256>(base); }
__global__ __launch_bounds__(256) void conv_stem_rows_kernel(ConvArgs a, int P, int Hp, int Wp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int mt_per_xcd = (a.mtiles + 7) >> 3;
    const int mtile = xcd * mt_per_xcd + j;
    if (mtile >= a.mtiles) return;
    const unsigned smem_base = lds_addr(smem);
    unsigned char* Ws = smem;                            // [4 taps][64 oc][128 B]
    unsigned char* Xs = smem + SRF_W + wave * SRF_SLAB;  // this wave's stage
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    // ---- loads: the wave's slab (9 pieces), then its quarter of the weights (8 pieces: tap = wave)
    {
        const int stg = mtile * 4 + wave;
        const int R = stg / a.seg_nseg, sg = stg - R * a.seg_nseg;
        const int n = R / P, oh = R - n * P;
        const int ow0 = min(64 * sg, a.seg_Q - 64);
        const int x_base = ((n * Hp + 2 * oh) * Wp + 2 * ow0) * 8;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int off = 1024 * p + 16 * lane;
            const int row = off / SRF_PITCH, col = off - row * SRF_PITCH;
            const bool ok = stg < a.seg_stages && row < 8 && col < 1072;
            dma16(rin, Xs + p * 1024, ok ? x_base + row * Wp * 8 + col : (int)0x80000000);
        }
        // weight tile of tap `wave`: 64 rows x 128 B, chunk swizzle as in the flat kernel
        const int prow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + prow;
            const int sch = pch ^ swz128(row);
            dma16(rwt, Ws + wave * 8192 + i * 1024, (row * 4 + wave) * 128 + sch * 16);
        }
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[nn][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fg = lane >> 4, fswz = swz128(frow);
    const unsigned abase = lds_addr(Xs) + frow * 16 + fg * 16;
    const unsigned wb0 = smem_base + frow * 128 + (((0 + fg) ^ fswz) << 4), wb1 = smem_base + frow * 128 + (((4 + fg) ^ fswz) << 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's share of the weights has landed
    asm volatile("" ::: "memory");
    auto half = [&](auto T4c, auto KKc) __attribute__((always_inline)) {
        constexpr int T4 = decltype(T4c)::value, KK = decltype(KKc)::value;
        uint4 px[4], wf[4];
        px[0] = srf_a<T4, KK, 0>(abase);
        px[1] = srf_a<T4, KK, 1>(abase);
        px[2] = srf_a<T4, KK, 2>(abase);
        px[3] = srf_a<T4, KK, 3>(abase);
        const unsigned wb = KK ? wb1 : wb0;
        wf[0] = lds_read16_asm_off<T4 * 8192 + 0>(wb);
        wf[1] = lds_read16_asm_off<T4 * 8192 + 2048>(wb);
        wf[2] = lds_read16_asm_off<T4 * 8192 + 4096>(wb);
        wf[3] = lds_read16_asm_off<T4 * 8192 + 6144>(wb);
        lds_wait();
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int m = 0; m < 4; ++m) Mma<bf16>::run(wf[nn], px[m], acc[nn][m]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    half(I0{}, I0{});
    half(I0{}, I1{});
    half(I1{}, I0{});
    half(I1{}, I1{});
    half(I2{}, I0{});
    half(I2{}, I1{});
    half(I3{}, I0{});
    half(I3{}, I1{});
    __syncthreads();
    conv_epilogue<bf16, 256, 64, 4, 1>(acc, smem, a, mtile * 256, 0, mtile);
}

// =====================================================================================================
// Persistent form of the row-slab stem forward (the one launched): same geometry and fragment addressing, but
// * the whole 64 x (4 taps x 64) weight matrix lives in each wave's registers (32 MFMA operands, 128 VGPRs; staged through
//   LDS once per block),
// * a WAVE walks over stages on its own -- its two slab buffers, the staged output tile (over the slab just consumed) and its
//   stores are private to it, so the loop has no barrier at all -- and the slab of its next stage is in flight (LDS-DMA)
//   while it multiplies the current one,
// * the BatchNorm statistics accumulate in registers across the wave's stages: ONE partial row per block
//   (conv_stem_tiles_m() reports SRP_GRID rows).
// The one-tile-per-block form spent ~85 % of a block's life outside its 128 MFMAs per wave (weights + slab from L2 / HBM,
// eight exposed LDS round trips, a 256 x 64 epilogue): 6-9 % of the MFMA peak against an HBM floor (308 MB of output at
// B = 64) three times lower than its run time.
// LDS per wave: [slab 0][slab 1] of 9 KiB; the block's weight staging (32 KiB) uses the same bytes before the loop.
// =====================================================================================================
constexpr int SRP_GRID = 512;
__global__ __launch_bounds__(256, 2) void conv_stem_pers_kernel(ConvArgs a, int P, int Hp, int Wp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PITCH = 64 * 2 + 16;  // staged tile: [64 pixels][64 channels] bf16, 144-byte pitch = 9 216 B = one slab buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned smem_base = lds_addr(smem);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t wreg[4][2][4];  // [tap pair t4][filter row 2*t4 + kk][16-channel fragment]
    {
        const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
        for (int p = wave; p < 32; p += 4) {  // rows (oc*4 + t4) of 128 B, chunk-swizzled
            const int row = p * 8 + (lane >> 3);
            dma16(rwt, smem + p * 1024, row * 128 + (((lane & 7) ^ swz128(row)) << 4));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const int row = (nn * 16 + frow) * 4 + t4;
                    const unsigned ad = smem_base + row * 128 + (((kk * 4 + fg) ^ swz128(row)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(wreg[t4][kk][nn]) : "v"(ad));
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // stages of this wave: XCD x owns a contiguous range, its 4 * gridDim/8 waves take every (that many)-th stage of it
    const int xcd = blockIdx.x & 7, wj = (blockIdx.x >> 3) * 4 + wave, wpx = (gridDim.x >> 3) * 4;
    const int st_per_xcd = (a.seg_stages + 7) >> 3;
    const int s_end = min(a.seg_stages, (xcd + 1) * st_per_xcd);
    int stg = xcd * st_per_xcd + wj;
    unsigned char* Xw = smem + wave * (2 * SRF_SLAB);
    const unsigned xw_base = smem_base + wave * (2 * SRF_SLAB);
    auto load_slab = [&](int buf, int sg_all) {
        const int R = sg_all / a.seg_nseg, sg = sg_all - R * a.seg_nseg;
        const int n = R / P, oh = R - n * P;
        const int ow0 = min(64 * sg, a.seg_Q - 64);
        const int x_base = ((n * Hp + 2 * oh) * Wp + 2 * ow0) * 8;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // (recomputed per stage, not kept in registers)
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int off = 1024 * p + 16 * ln;
            const int row = off / SRF_PITCH, col = off - row * SRF_PITCH;
            const bool ok = row < 8 && col < 1072;
            dma16(rin, Xw + buf * SRF_SLAB + p * 1024, ok ? x_base + row * Wp * 8 + col : (int)0x80000000);
        }
    };
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
    bf16* __restrict__ gout = (bf16*)a.out;
    const int ec = lane & 7, er0 = lane >> 3;  // row pass p: tile row er0 + 8p, 16-byte chunk ec
    if (stg < s_end) load_slab(0, stg);
    if (stg + wpx < s_end) load_slab(1, stg + wpx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) asm volatile("" : "+v"(wreg[t4][kk][nn]));
    for (int it = 0; stg < s_end; ++it, stg += wpx) {
        const unsigned slab = xw_base + (it & 1) * SRF_SLAB;
        const unsigned abase = slab + frow * 16 + fg * 16;
        f32x4_t acc[4][4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[nn][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // eight half-steps (filter rows 0..7), one set of pixel fragments (the registers hold the filter: a second set would
        // spill; the other wave of the SIMD multiplies while this one waits for its reads)
        uint4 px[4];
        auto half = [&](auto T4c, auto KKc) __attribute__((always_inline)) {
            constexpr int T4 = decltype(T4c)::value, KK = decltype(KKc)::value;
            px[0] = srf_a<T4, KK, 0>(abase);
            px[1] = srf_a<T4, KK, 1>(abase);
            px[2] = srf_a<T4, KK, 2>(abase);
            px[3] = srf_a<T4, KK, 3>(abase);
            lds_wait();
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[nn][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[T4][KK][nn]),
                                                                        __builtin_bit_cast(bf16x8_t, px[m]), acc[nn][m], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        half(I0{}, I0{});
        half(I0{}, I1{});
        half(I1{}, I0{});
        half(I1{}, I1{});
        half(I2{}, I0{});
        half(I2{}, I1{});
        half(I3{}, I0{});
        half(I3{}, I1{});
        // stage the 64 x 64 tile over the slab just consumed (this wave's own reads of it are complete).
        // D[i][j]: i = channel = (lane>>4)*4 + reg, j = pixel = lane&15
        {
            const unsigned cs0 = slab + frow * PITCH + fg * 8;
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint2 v = make_uint2(pack2bf(acc[nn][m][0], acc[nn][m][1]), pack2bf(acc[nn][m][2], acc[nn][m][3]));
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(cs0), "v"(v), "n"(m * 16 * PITCH + nn * 32) : "memory");
                }
        }
        // the tile is in LDS; the next stage's slab has landed (it had this stage's K-loop); the stores of the stage before
        // are done
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int R = stg / a.seg_nseg, sg = stg - R * a.seg_nseg;
        const int ow0 = min(64 * sg, a.seg_Q - 64);
        const int skip = 64 * sg - ow0;  // rows below repeat the previous stage of the image row: neither stored nor counted
        const unsigned char* Cs = smem + wave * (2 * SRF_SLAB) + (it & 1) * SRF_SLAB;
        uint4 vq[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = er0 + 8 * p;
            vq[p] = *(const uint4*)(Cs + row * PITCH + ec * 16);
            if ((a.stats || a.sacc.acc) && row >= skip) {
                float f[8];
                unpack16<bf16>(vq[p], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ssum[e] += f[e];
                    ssq[e] += f[e] * f[e];
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = er0 + 8 * p;
            if (row >= skip) *(uint4*)(gout + ((size_t)R * a.seg_Q + ow0 + row) * 64 + ec * 8) = vq[p];
        }
        // this buffer is free (the row reads above have returned: their values are in the stores): the stage after next
        if (stg + 2 * wpx < s_end) load_slab(it & 1, stg + 2 * wpx);
    }
    if (a.stats || a.sacc.acc) {
        // lanes with equal (lane & 7) hold the same 8 channels: fold them, then the 4 waves in fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e)
            for (int msk = 8; msk < 64; msk <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], msk);
                ssq[e] += __shfl_xor(ssq[e], msk);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave is out of its loop: the LDS is free
        float* red = (float*)smem;  // [4 waves][64][2]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 64 + lane * 8 + e) * 2 + 0] = ssum[e];
                red[(wave * 64 + lane * 8 + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, w = tid & 1;
            const float s2 = ((red[(0 * 64 + c) * 2 + w] + red[(1 * 64 + c) * 2 + w]) + red[(2 * 64 + c) * 2 + w]) +
                             red[(3 * 64 + c) * 2 + w];
            if (a.sacc.acc)
                bn_acc_add(a.sacc, c, w, s2);
            else
                st_agent(a.stats + ((size_t)blockIdx.x * 64 + c) * 2 + w, s2);
        }
    }
}

// =====================================================================================================
// 3x3 stride-1 "slab" kernel (forward and data gradient).
//
// With NHWC storage the pixels [m0-(W+1), m0+BM+(W+1)) around a flat output tile [m0, m0+BM) are ONE
// contiguous slab of the input tensor, and tap (r,s) of output pixel m reads pixel m + pshift[tap],
// pshift in [-(W+1), W+1] -- also across image-row and image boundaries, where the gather table's
// mask bit says "padding" and the lane reads a zero row instead.  So the A operand of all 9 taps is
// loaded ONCE per 64-channel chunk (contiguous LDS-DMA, no gather) and the K-loop only streams the
// 8 KB weight tiles: ~3x less L2->LDS traffic than the flat kernel, which is what bounds it
// (memory latency x limited bytes in flight per CU).
// Measured alternatives that LOST on MI355X (tools/bench_conv.py, B=64): a 3-deep weight ring with
// counted vmcnt and dummy DMA padding (-26 %), weight fragments straight from L2 to registers with a
// barrier-free K-loop (-80 %).
// LDS: [weight ring: 2 x BN x 128][slab 0][slab 1 (only when IC > one chunk)][zero row].
// =====================================================================================================
// fragment reads of one 32-channel half: MI pixel fragments (base pb[m] + 2048*m) and NI weight
// fragments (wb + 2048*n), all through instruction offsets
template <int MI, int NI>
__device__ __forceinline__ void slab_reads(uint4 (&px)[MI], uint4 (&wf)[NI], const unsigned (&pb)[MI], unsigned wb) {
    static_assert(MI >= 2 && MI <= 4, "MI");
    static_assert(NI == 4 || NI == 8, "NI");
    px[0] = lds_read16_asm_off<0>(pb[0]);
    px[1] = lds_read16_asm_off<2048>(pb[1]);
    if constexpr (MI >= 3) px[2] = lds_read16_asm_off<4096>(pb[2]);
    if constexpr (MI == 4) px[3] = lds_read16_asm_off<6144>(pb[3]);
    wf[0] = lds_read16_asm_off<0>(wb);
    wf[1] = lds_read16_asm_off<2048>(wb);
    wf[2] = lds_read16_asm_off<4096>(wb);
    wf[3] = lds_read16_asm_off<6144>(wb);
    if constexpr (NI == 8) {
        wf[4] = lds_read16_asm_off<8192>(wb);
        wf[5] = lds_read16_asm_off<10240>(wb);
        wf[6] = lds_read16_asm_off<12288>(wb);
        wf[7] = lds_read16_asm_off<14336>(wb);
    }
}

// (NWV = 8 -- the same tile on 512 threads, 4 x 2 waves, four waves per SIMD with two blocks per CU -- was built and
// measured: no faster than four waves (128 x 128 at 128 / 256 channels: 53 / 57 us forward either way; the step 6.45
// against 6.48 ms).  Occupancy is not what bounds this kernel; the parameter stays for the record, only 4 is launched.)
// (second bound = waves per SIMD the compiler must leave room for.  The 8-wave tile had 4 -- a 128-register cap that made its data
// gradient spill 64-108 bytes per lane into the epilogue; it only runs where the launch has fewer blocks than CUs, so 2 -- one block
// per CU, 256 registers -- costs no residency)
template <typename T, int BM, int BN, int MODE, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, (NWV == 8) ? 2 : ((BM >= 192 && BN == 128) ? 2 : 1)) void conv3x3_slab_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WM = 4, WN = NWV / 4;
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int BROWS = BN / (8 * NWV);  // weight-tile DMA instructions per wave and K-step
    constexpr int WTM = BM / WM, MI = WTM / 16, WTN = BN / WN, NI = WTN / 16;
    constexpr int WSTAGE = BN * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int ntn = a.OC / BN;
    const int nsplit = a.ksplit > 1 ? a.ksplit : 1;
    const int L = xcd_linear(a.mtiles * ntn * nsplit);
    if (L < 0) return;
    const int mtile = L / (ntn * nsplit), jr = L - mtile * (ntn * nsplit);
    const int ntile = jr % ntn, split = jr / ntn;  // (the splits of a tile are neighbours on one XCD)
    const int m0 = mtile * BM, n0 = ntile * BN;
    GDL_STAMP(0);
#ifdef GDL_TIMING
    if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
#endif
    const int esz = (int)sizeof(T);
    const int kpt = a.IC / BKE;  // channel chunks
    const int kc0 = split * (kpt / nsplit), kc1 = kc0 + kpt / nsplit;  // this block's chunks
    const int nins = (a.slab_rows + 7) >> 3;
    const int slab_bytes = nins * 1024;
    const unsigned smem_base = lds_addr(smem);
    const unsigned slab_base = smem_base + 2 * WSTAGE;
    // zero row for masked (padding) taps: the rows of slab 0 past slab_rows (its last DMA piece is only partly
    // used; the out-of-range lanes deposit zeros) when there are any, else 1 KiB after the slabs
    const bool spare_row = (a.slab_rows & 7) != 0;
    const int nslab = (kc1 - kc0 > 1 && !a.single_slab) ? 2 : 1;
    const unsigned zrow = spare_row ? slab_base + a.slab_rows * 128 : slab_base + nslab * slab_bytes;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

    // zero row: an out-of-range LDS-DMA deposits zeros (1 KiB, one wave instruction) -- no LDS store, no barrier
    if (wave == 0 && !spare_row) dma16(rin, smem + (zrow - smem_base), (int)0x80000000);

    // per-lane validity masks of the MI pixel fragments this wave multiplies
    const int frow = lane & 15, fg = lane >> 4;
    unsigned fmask[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int mm = m0 + wm * WTM + m * 16 + frow;
        fmask[m] = mm < a.M ? a.table[mm].mask : 0u;
    }
    // weight tile DMA: lane -> row (tid>>3) + 8*NWV*i, physical chunk tid&7 (source chunk swizzled)
    const int row0 = tid >> 3, schunk = (tid & 7) ^ swz128(row0);
    int b_off[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_off[i] = (n0 + row0 + 8 * NWV * i) * a.ntaps * a.IC * esz + schunk * 16;
    const int wrow = wave * 8;
    auto load_w = [&](int buf, int kc, int tap) {
        const int ub = (tap * a.IC) * esz + kc * 128;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) dma16(rwt, smem + buf * WSTAGE + (wrow + 8 * NWV * i) * 128, b_off[i] + ub);
    };
    // slab DMA: instruction jj covers slab rows 8*jj .. 8*jj+7; waves take jj = wave, wave+NWV, ...
    auto load_slab = [&](int sbuf, int kc) {
        unsigned char* dst = smem + 2 * WSTAGE + sbuf * slab_bytes;
        for (int jj = wave; jj < nins; jj += NWV) {
            const int sr = jj * 8 + (lane >> 3);
            const int pix = m0 - (a.W + 1) + sr;
            const bool ok = sr < a.slab_rows && (unsigned)pix < (unsigned)a.in_pixels;
            const int v = ok ? pix * a.IC * esz + kc * 128 + (((lane & 7) ^ swz128(sr)) << 4) : (int)0x80000000;
            dma16(rin, dst + jj * 1024, v);
        }
    };

    f32x4_t acc[NI][MI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int m = 0; m < MI; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fswz = swz128(frow);
    // (this wave's weight fragments start WTN rows into the tile: 64 rows keep the swizzle term)
    const int woff0 = (wn * WTN + frow) * 128 + (((0 + fg) ^ fswz) << 4), woff1 = (wn * WTN + frow) * 128 + (((4 + fg) ^ fswz) << 4);
    int prow[MI];
    unsigned zb[MI];  // zero-row address of fragment m, pre-biased by the instruction offset 2048*m
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        prow[m] = wm * WTM + m * 16 + frow + (a.W + 1);
        zb[m] = zrow + (fg << 4) - 2048u * m;
    }

    load_slab(0, kc0);
    load_w(0, kc0, 0);
    GDL_STAMP(1);
    // pixel shift of tap t without a scalar load per tap: +d1 inside a filter row, +d3 at a row change
    const int sh0 = a.pshift[0], d1 = a.pshift[1] - a.pshift[0], d3 = a.pshift[3] - a.pshift[2];
    int wbuf = 0;
#ifdef GDL_TIMING
    unsigned long long ta, tb, acc_wait = 0, acc_issue = 0, acc_rd = 0, acc_m0 = 0, acc_m1 = 0;
#define TSEG(accv)                          \
    tb = __builtin_amdgcn_s_memtime();      \
    accv += tb - ta;                        \
    ta = tb;
#else
#define TSEG(accv)
#endif
    for (int kc = kc0; kc < kc1; ++kc) {
        const unsigned slab = slab_base + (nslab == 2 ? ((kc - kc0) & 1) * slab_bytes : 0);
        int sh = sh0, scol = 0;
        for (int tap = 0; tap < a.ntaps; ++tap) {
#ifdef GDL_TIMING
            ta = __builtin_amdgcn_s_memtime();
#endif
            // everything issued during the previous K-step (next weight tile, next slab) has landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (nslab == 1 && tap == 0 && kc > kc0) {
                // single slab buffer: every wave is past its last read of the previous chunk's slab (barrier above);
                // fetch this chunk's slab now -- the wait is exposed once per chunk (9 K-steps), the other block of
                // the CU works meanwhile, and the LDS it saves is what lets two 256-row blocks share a CU
                load_slab(0, kc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            TSEG(acc_wait)
            if (kc == kc0 && tap == 0) GDL_STAMP(2);
            if (kc == kc0 && tap == 1) GDL_STAMP(3);
            const unsigned Bs = smem_base + wbuf * WSTAGE;
            // Pixel fragment m sits 16 rows (2048 bytes) below fragment 0 and 16 rows do not change the
            // swizzle term, so ONE address is computed per tap; the per-fragment part is an instruction
            // offset.  A masked (padding) tap reads the zero row: its base is pre-biased by -2048*m.
            // The second 32-channel half is chunk^4, i.e. address^64.
            const int sr0 = prow[0] + sh;
            const unsigned ad0 = slab + sr0 * 128 + (((fg ^ swz128(sr0))) << 4);
            unsigned pb[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m) pb[m] = ((fmask[m] >> tap) & 1u) ? ad0 : zb[m];
            const unsigned wb0 = Bs + woff0, wb1 = Bs + woff1;
            // Prefetch AFTER fragment reads have been requested: issuing LDS-DMA blocks the wave for ~150 clk per
            // 1 KiB piece (tools/timing_probe.py: 490-610 clk per K-step), about the time the fragment reads need to
            // come back (~460 clk) -- issued in front of the reads the two latencies added up.  The weight tile of the
            // next K-step goes to the OTHER ring slot, at tap 0 the next chunk's slab to the other slab buffer:
            // nothing the pending reads touch.
            auto prefetch = [&]() __attribute__((always_inline)) {
                asm volatile("" ::: "memory");
                int ntap = tap + 1, nkc = kc;
                if (ntap == a.ntaps) {
                    ntap = 0;
                    ++nkc;
                }
                if (nkc < kc1) load_w(wbuf ^ 1, nkc, ntap);
                if (nslab == 2 && tap == 0 && kc + 1 < kc1) load_slab((kc + 1 - kc0) & 1, kc + 1);
                asm volatile("" ::: "memory");
            };
            if constexpr ((MI + NI) * 8 > 80) {
                // big tile (MI = 4, NI = 8): 128 accumulator registers leave room for ONE half's fragments (48) if two
                // waves are to share a SIMD: the halves are read one after the other
                uint4 px[MI], wf[NI];
                slab_reads<MI, NI>(px, wf, pb, wb0);
                TSEG(acc_rd)
                prefetch();
                TSEG(acc_issue)
                lds_wait();
#pragma unroll
                for (int n = 0; n < NI; ++n)
#pragma unroll
                    for (int m = 0; m < MI; ++m) Mma<T>::run(wf[n], px[m], acc[n][m]);
                TSEG(acc_m0)
#pragma unroll
                for (int m = 0; m < MI; ++m) pb[m] ^= 64u;
                slab_reads<MI, NI>(px, wf, pb, wb1);
                lds_wait();
#pragma unroll
                for (int n = 0; n < NI; ++n)
#pragma unroll
                    for (int m = 0; m < MI; ++m) Mma<T>::run(wf[n], px[m], acc[n][m]);
                TSEG(acc_m1)
            } else {
                // both 32-channel halves' fragments are requested up front; the second half's LDS latency
                // hides behind the first half's MFMAs
                uint4 px[2][MI], wf[2][NI];
                slab_reads<MI, NI>(px[0], wf[0], pb, wb0);
#pragma unroll
                for (int m = 0; m < MI; ++m) pb[m] ^= 64u;
                slab_reads<MI, NI>(px[1], wf[1], pb, wb1);
                TSEG(acc_rd)
                prefetch();
                TSEG(acc_issue)
                // (lgkmcnt counts at most 15: with 8 weight fragments the second half's 10 reads are all that may remain)
                lds_wait_n<MI + NI>();
#pragma unroll
                for (int n = 0; n < NI; ++n)
#pragma unroll
                    for (int m = 0; m < MI; ++m) Mma<T>::run(wf[0][n], px[0][m], acc[n][m]);
                lds_wait();
                TSEG(acc_m0)
#pragma unroll
                for (int n = 0; n < NI; ++n)
#pragma unroll
                    for (int m = 0; m < MI; ++m) Mma<T>::run(wf[1][n], px[1][m], acc[n][m]);
                TSEG(acc_m1)
            }
            wbuf ^= 1;
            if (++scol == 3) {
                scol = 0;
                sh += d3;
            } else {
                sh += d1;
            }
        }
    }
    GDL_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsplit > 1) {
        // split-K: the fp32 accumulators go out as they are (D[i][j]: channel (lane >> 4) * 4 + reg, pixel lane & 15 -- a lane's
        // four values are 16 contiguous bytes of one row); splitk_finish_kernel does the rest
        float* ws = a.split_ws + (size_t)split * a.M * a.OC;
#pragma unroll
        for (int m = 0; m < MI; ++m) {
            const int row = m0 + wm * WTM + m * 16 + (lane & 15);
            if (row < a.M) {
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    *(f32x4_t*)(ws + (size_t)row * a.OC + n0 + wn * WTN + n * 16 + (lane >> 4) * 4) = acc[n][m];
            }
        }
        GDL_STAMP(5);
        return;
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN, MODE == MODE_DGRAD>(acc, smem, a, m0, n0, mtile, ntile);
    GDL_STAMP(5);
#ifdef GDL_TIMING
    if (a.dbg && threadIdx.x == 0) {
        unsigned long long* d = a.dbg + ((size_t)(1 << 15) + blockIdx.x) * 8;
        d[0] = acc_wait;
        d[1] = acc_issue;
        d[2] = acc_rd;
        d[3] = acc_m0;
        d[4] = acc_m1;
        d[5] = (unsigned long long)kpt * a.ntaps;
    }
#endif
}

#include "conv_pslab.h"

// =====================================================================================================
// 3x3 stride-1 convolution between 64 and 64 channels (layer 1 of both encoders: forward and data gradient; bf16),
// persistent, weights in registers.
//
// With 64 input channels the slab kernel's K-loop is 9 K-steps: its blocks spend half their life in the prologue (slab
// DMA from HBM) and the epilogue, and every K-step pays a barrier, a weight-tile DMA and 6 fragment reads per 8 MFMAs
// (tools/timing_probe.py: 21 % MFMA duty per wave, the LDS array the busiest unit).  Here the WHOLE filter of a wave's
// 32 output channels (9 taps x 64 channels = 36 MFMA operands, 144 VGPRs) is loaded once and stays in registers, a block
// walks over M-tiles, and the next tile's slab is in flight (LDS-DMA into the other slab buffer) while the current one is
// multiplied: no weight traffic, no barrier and no DMA inside a tile's K-loop, 4 fragment reads per 8 MFMAs, HBM latency
// hidden.  4 waves = 2 (64-pixel halves) x 2 (32-channel halves) of a 128 x 64 tile, two blocks per CU.
// Per tile: K-loop | barrier | accumulators -> LDS (over the slab just consumed) | barrier | wait for the prefetched
// slab | rows -> HBM (addend / ReLU bits / statistics as in conv_epilogue) | barrier.  The tile's stores are only waited
// for one tile later.  BatchNorm statistics accumulate in registers ACROSS the block's tiles; the block leaves ONE
// partial row (stats[blockIdx.x][64][2]): conv_tiles_m() reports C64_GRID rows for these layers.
// LDS: [slab 0][slab 1][1 KiB of zeros] (at least the 72 KiB the filter staging of the prologue needs).
// =====================================================================================================
constexpr int C64_BM = 128;
constexpr int C64_GRID = 512;  // two blocks per CU of an MI355X; also the number of BatchNorm partial rows
constexpr int C64_F32_TILE = 35 * 1024;  // BW: a slab buffer also holds the staged tile as [128][64] floats at a 272-byte pitch
// BW (data gradient only): the BatchNorm-backward sums of the stored rows against the partner tensor a.bw_y (ops.h BwdStats,
// one partner; optional ReLU mask), accumulated across the block's tiles like the forward's statistics -- and like the forward it
// then keeps ONE set of pixel fragments (the accumulators take the registers of the second).
// ADD (data gradient only): the launch has an addend (the identity shortcut's gradient); the others carry no registers for it
// (20 -> 8 bytes of scratch in the BatchNorm-sums form: step -0.2 %, four A/B rounds).
template <int MODE, bool BW = false, bool ADD = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(ConvArgs a) {
    static_assert(!BW || MODE == MODE_DGRAD, "BW: data gradient only");
    static_assert(!ADD || MODE == MODE_DGRAD, "ADD: data gradient only");
    constexpr bool STATS = MODE == MODE_FWD;  // per-channel sums accumulated in registers (tsum / tsq) across the tiles
    // (BW accumulates in LDS instead -- per tile: fold the eight row-lanes of a wave that hold the same channels, then lanes 0-7
    // add into the wave's row of [4 waves][64][2] floats -- so that no accumulator is live across the K-loop: with sixteen more
    // registers there the compiler spilled per-tap addresses, and a spill reload inside the K-loop is a `s_waitcnt vmcnt(0)`,
    // i.e. a wait for the slab prefetch just issued: 35 -> 90 us per launch)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = C64_BM, PITCH = 64 * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 15, fg = lane >> 4;
    GDL_STAMP(0);
    // tiles of this block: XCD x owns a contiguous range, its blocks take every (gridDim/8)-th tile of it
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int mt_per_xcd = (a.mtiles + 7) >> 3;
    const int t_end = min(a.mtiles, (xcd + 1) * mt_per_xcd);
    int tile = xcd * mt_per_xcd + bj;
    const int nins = (a.slab_rows + 7) >> 3;
    const int slab_bytes = BW ? max(nins * 1024, C64_F32_TILE) : nins * 1024;
    const unsigned smem_base = lds_addr(smem);
    const int nbuf = a.single_slab ? 1 : 2;  // wide images: ONE slab buffer (the next tile's slab is fetched behind the row pass, exposed)
    const unsigned zrow = smem_base + nbuf * slab_bytes;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    // this wave's filter: operand (tap, 32-channel half h, 16-output-channel fragment n) = 8 input channels
    // h*32 + fg*8.. of output channel wn*32 + n*16 + frow
    // The 72 KiB filter comes through LDS ONCE per block (72 DMA pieces, a straight copy of [oc][tap][64 c]) and each wave picks
    // its 36 operands from there: four waves fetching them from L2 themselves asked the same 576 lines 8 times per CU, and with
    // every block of the launch starting at once that request rate, not the bytes, made the prologue 18 000 clk.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t wreg[9][2][2];
    {
        const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
        // (16-byte chunks of a 128-byte row XOR-swizzled by (row >> 1) & 7, as in the slabs: the operand reads below
        // walk rows 9 apart)
        for (int p = wave; p < 72; p += 4) {
            const int row = p * 8 + (lane >> 3);
            dma16(rwt, smem + p * 1024, row * 128 + (((lane & 7) ^ swz128(row)) << 4));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int row = (wn * 32 + n * 16 + frow) * 9 + tap;
                    const unsigned ad = smem_base + row * 128 + (((h * 4 + fg) ^ swz128(row)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(wreg[tap][h][n]) : "v"(ad));
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if constexpr (BW) {  // the partner BatchNorm's mean / rstd: 2 x 64 floats behind the zero KiB (the filter staging is over)
        if (tid < 128) ((float*)(smem + nbuf * slab_bytes + 1024))[tid] = tid < 64 ? a.bw_mean[tid] : a.bw_rstd[tid - 64];
        ((float*)(smem + nbuf * slab_bytes + 1536))[tid] = 0.f;  // the waves' accumulator rows [4][64][2]
        ((float*)(smem + nbuf * slab_bytes + 1536))[tid + 256] = 0.f;
    }
    if (wave == 0) dma16(rin, smem + nbuf * slab_bytes, (int)0x80000000);  // out-of-range LDS-DMA deposits zeros
    auto load_slab = [&](int sbuf, int m0) {
        unsigned char* dst = smem + sbuf * slab_bytes;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // the row / chunk terms are recomputed here, not kept in registers across the K-loop
        for (int jj = wave; jj < nins; jj += 4) {
            const int sr = jj * 8 + (ln >> 3);
            const int pix = m0 - (a.W + 1) + sr;
            const bool ok = sr < a.slab_rows && (unsigned)pix < (unsigned)a.in_pixels;
            const int v = ok ? pix * 128 + (((ln & 7) ^ swz128(sr)) << 4) : (int)0x80000000;
            dma16(rin, dst + jj * 1024, v);
        }
    };
    auto load_masks = [&](int m0, unsigned (&fm)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int mm = m0 + wm * 64 + m * 16 + frow;
            fm[m] = mm < a.M ? a.table[mm].mask : 0u;
        }
    };
    const int prow0 = wm * 64 + frow + (a.W + 1);
    // a masked (padding) tap reads zeros: any 16 bytes of the zero KiB will do, so the address is wave-uniform (a scalar
    // operand of the select); pre-biased by the instruction offset 2048*m of fragment m
    unsigned zb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) zb[m] = zrow - 2048u * m;
    // epilogue mapping: thread -> 16-byte chunk ec of rows er0 + 32*p
    const int ec = tid & 7, er0 = tid >> 3;
    // BatchNorm statistics of the thread's 8 channels, accumulated across the block's tiles (forward only; these 16 registers
    // are why the forward keeps ONE set of pixel fragments: the K-loop sits at the 256-register limit of two waves per SIMD)
    float tsum[8], tsq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) tsum[e] = tsq[e] = 0.f;
    bf16* __restrict__ gout = (bf16*)a.out;
    // (addend and ReLU bits belong to data gradients: the forward instantiation carries neither their registers nor their branches)
    const bf16* __restrict__ gadd = ADD ? (const bf16*)a.addend : nullptr;
    const uint8_t* __restrict__ rbits = MODE == MODE_DGRAD ? a.relu_bits : nullptr;

    unsigned fmask[4] = {0u, 0u, 0u, 0u};
    if (tile < t_end) {
        load_slab(0, tile * BM);
        load_masks(tile * BM, fmask);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (a use the compiler can see: it places its own wait for the filter loads HERE, not in front of the first MFMA of
    // every tile -- where it would also wait for the slab prefetch issued just before)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wreg[tap][h][n]));
    asm volatile("" : "+v"(fmask[0]), "+v"(fmask[1]), "+v"(fmask[2]), "+v"(fmask[3]));
#ifdef GDL_TIMING
    unsigned long long tq_a, tq_b, tq[7] = {0, 0, 0, 0, 0, 0, 0};
    int tq_n = 0;
#define C64_SEG(i)                       \
    tq_b = __builtin_amdgcn_s_memtime(); \
    tq[i] += tq_b - tq_a;                \
    tq_a = tq_b;
    GDL_STAMP(1);
#else
#define C64_SEG(i)
#endif
    for (int it = 0; tile < t_end; ++it, tile += bpx) {
#ifdef GDL_TIMING
        tq_a = __builtin_amdgcn_s_memtime();
        ++tq_n;
#endif
        const int m0 = tile * BM;
        const bool more = tile + bpx < t_end;
        const int sb = nbuf == 2 ? (it & 1) : 0;
        if (more && nbuf == 2) {  // the other slab buffer is free: every wave is past its row reads of the tile before (barrier)
            load_slab(sb ^ 1, (tile + bpx) * BM);
        }
        C64_SEG(0)
        const unsigned slab = smem_base + sb * slab_bytes;
        f32x4_t acc[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // 18 half-steps (tap, 32-channel half); the pixel fragments of half-step s+1 are requested before the MFMAs of s
        constexpr int NPX = STATS ? 1 : 2;  // forward / BW: one set (the statistics live in registers instead)
        uint4 px[NPX][4];
        unsigned pb[4];
        auto tap_addr = [&](int tap) __attribute__((always_inline)) {
            const int sr0 = prow0 + a.pshift[tap];
            const unsigned ad0 = slab + sr0 * 128 + ((fg ^ swz128(sr0)) << 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) pb[m] = ((fmask[m] >> tap) & 1u) ? ad0 : zb[m];
        };
        auto reads = [&](uint4 (&p)[4]) __attribute__((always_inline)) {
            p[0] = lds_read16_asm_off<0>(pb[0]);
            p[1] = lds_read16_asm_off<2048>(pb[1]);
            p[2] = lds_read16_asm_off<4096>(pb[2]);
            p[3] = lds_read16_asm_off<6144>(pb[3]);
        };
        tap_addr(0);
        reads(px[0]);
#pragma unroll
        for (int hs = 0; hs < 18; ++hs) {
            const int tap = hs >> 1, h = hs & 1;
            if (hs + 1 < 18) {
                if (h == 0) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) pb[m] ^= 64u;  // the second 32-channel half: chunk ^ 4
                } else {
                    tap_addr(tap + 1);
                }
                if constexpr (NPX == 2) {
                    reads(px[(hs + 1) & 1]);
                    lds_wait_n<4>();
                } else {
                    lds_wait();
                }
            } else {
                lds_wait();
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[tap][h][n]),
                                                                       __builtin_bit_cast(bf16x8_t, px[hs & (NPX - 1)][m]), acc[n][m], 0, 0, 0);
            if constexpr (NPX == 1)
                if (hs + 1 < 18) reads(px[0]);  // (behind the MFMAs that read the registers: they are issued in order)
        }
        C64_SEG(1)
        if (more) load_masks((tile + bpx) * BM, fmask);  // the next tile's tap masks (waited for with the slab below)
        // the addend / ReLU-bit loads of the tile's four row passes: in flight while the tile is staged
        uint4 gq[4], yq[4];
        unsigned mkq[4];

#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + er0 + p * 32;
            const size_t goff = (size_t)m * 64 + ec * 8;
            gq[p] = make_uint4(0u, 0u, 0u, 0u);
            yq[p] = make_uint4(0u, 0u, 0u, 0u);
            mkq[p] = 0xffu;
            if (m < a.M) {
                if (gadd) gq[p] = *(const uint4*)(gadd + goff);
                if (rbits) mkq[p] = rbits[goff / 8];
                if constexpr (BW) yq[p] = *(const uint4*)((const bf16*)a.bw_y + goff);
            }
        }

        uint4 vq[4];
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
        float cm[8];  // BW: means of this thread's 8 channels, from the constants' LDS rows (live only during the row pass)
        unsigned char* Cs = smem + sb * slab_bytes;
        if constexpr (BW) {
            // Round 4: the BatchNorm-backward sums are taken from the fp32 accumulators (+ addend, masked) BEFORE the rounding to
            // bf16 (see conv_epilogue's F32ST): the tile is staged as [128][64] floats at a 272-byte pitch (34 KiB; the BW
            // instantiations size their slab buffers for it, C64_F32_TILE) and rounded once, at the store.
            constexpr int P32 = 64 * 4 + 16;
            static_assert(C64_BM * P32 <= C64_F32_TILE, "fp32 tile");
            // every wave is done with this slab
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            C64_SEG(2)
            {
                // (stores the compiler cannot see: in front of a visible LDS store it would wait, vmcnt(0), for the slab
                // prefetch AND for the addend loads issued just above)
                const unsigned cs0 = slab + (wm * 64 + (lane & 15)) * P32 + (wn * 32 + (lane >> 4) * 4) * 4;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(cs0), "v"(acc[n][m]), "n"(m * 16 * P32 + n * 64) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            C64_SEG(3)
            // the prefetched slab and masks have landed (they had the whole K-loop); the stores of the tile before are done
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C64_SEG(4)
            asm volatile("" : "+v"(fmask[0]), "+v"(fmask[1]), "+v"(fmask[2]), "+v"(fmask[3]));
            {  // (the means now, the rstd behind the row pass: eight registers less across it)
                const float4* cn = (const float4*)(smem + nbuf * slab_bytes + 1024);
                *(float4*)&cm[0] = cn[ec * 2], *(float4*)&cm[4] = cn[ec * 2 + 1];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = er0 + p * 32;
                float f[8];
                *(float4*)&f[0] = *(const float4*)(Cs + row * P32 + ec * 32);
                *(float4*)&f[4] = *(const float4*)(Cs + row * P32 + ec * 32 + 16);
                if (gadd) {
                    float g[8];
                    unpack16<bf16>(gq[p], g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                }
                if (rbits) {
                    const unsigned mk = mkq[p];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = ((mk >> e) & 1u) ? f[e] : 0.f;
                }
                if (m0 + row < a.M) {
                    float yv[8];
                    unpack16<bf16>(yq[p], yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        ssum[e] += f[e];
                        ssq[e] += f[e] * (yv[e] - cm[e]);  // (x rstd below)
                    }
                }
                vq[p] = pack16<bf16>(f);
            }
        } else {
        // every wave is done with this slab: the tile is staged over it, [128][64] bf16 at a 144-byte pitch
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        C64_SEG(2)
        {
            // D[i][j]: i = channel = (lane>>4)*4 + reg, j = pixel = lane&15
            // (stores the compiler cannot see: in front of a visible LDS store it would wait, vmcnt(0), for the slab prefetch
            // AND for the addend loads issued just above)
            const int px_row = wm * 64 + (lane & 15), ch = wn * 32 + (lane >> 4) * 4;
            const unsigned cs0 = slab + px_row * PITCH + ch * 2;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint2 v = make_uint2(pack2bf(acc[n][m][0], acc[n][m][1]), pack2bf(acc[n][m][2], acc[n][m][3]));
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(cs0), "v"(v), "n"(m * 16 * PITCH + n * 32) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        C64_SEG(3)
        // the prefetched slab and masks have landed (they had the whole K-loop); the stores of the tile before are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C64_SEG(4)
        // (tells the compiler the masks are here: it would otherwise wait for them -- and with them for the slab prefetch just
        // issued -- at their first use in the next tile's K-loop)
        asm volatile("" : "+v"(fmask[0]), "+v"(fmask[1]), "+v"(fmask[2]), "+v"(fmask[3]));
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = er0 + p * 32;
            uint4 v = *(const uint4*)(Cs + row * PITCH + ec * 16);
            if (gadd) {
                float f[8], g[8];
                unpack16<bf16>(v, f);
                unpack16<bf16>(gq[p], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g[e];
                v = pack16<bf16>(f);
            }
            if (rbits) {
                const unsigned mk = mkq[p];
                float f[8];
                unpack16<bf16>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = ((mk >> e) & 1u) ? f[e] : 0.f;
                v = pack16<bf16>(f);
            }
            if (MODE == MODE_FWD && (a.stats || a.sacc.acc) && m0 + row < a.M) {
                float f[8];
                unpack16<bf16>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ssum[e] += f[e];
                    ssq[e] += f[e] * f[e];
                }
            }
            vq[p] = v;
        }
        }
        if constexpr (BW) {
            // this tile's sums -> the wave's LDS row (before the stores: an LDS access behind pending stores waits for them)
            float rs[8];
            {
                const float4* cn = (const float4*)(smem + nbuf * slab_bytes + 1024);
                *(float4*)&rs[0] = cn[16 + ec * 2], *(float4*)&rs[4] = cn[16 + ec * 2 + 1];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ssq[e] *= rs[e];
            // Round 6 (from conv_pslab.h): the eight row-lanes that hold the same channels reduce AND scatter -- the wave's halves
            // split the two sums (v_permlane32_swap of the pair), the rows channels e / e + 4 (v_permlane16_swap), the row's halves
            // e / e + 2 (select + DPP rotation) -- 30 VALU instructions; a lane ends with two values and adds them into its own 8
            // bytes of the wave's row, [k 2][64 channels] floats.  (Before: 48 ds_bpermute round trips per 128 x 64 tile -- as long as
            // the tile's 144 MFMAs -- and a 16-float read-modify-write by eight lanes.)
            {
                float r[8], q[4], res[2];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssum[e]), __float_as_uint(ssq[e]), false, false);
                    r[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[e]), __float_as_uint(r[e + 4]), false, false);
                    q[e] = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                }
                const bool up = (lane & 8) != 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float keep = up ? q[e + 2] : q[e], give = up ? q[e] : q[e + 2];
                    res[e] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                }
                // k = lane bit 5, channel = 8 * (lane & 7) + 4 * bit 4 + 2 * bit 3 (+ 0 / 1)
                const unsigned ad = smem_base + nbuf * slab_bytes + 1536 + wave * 512 +
                                    (((lane >> 5) & 1) * 64 + (lane & 7) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2) * 4;
                uint2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(v));
                v.x = __float_as_uint(__uint_as_float(v.x) + res[0]), v.y = __float_as_uint(__uint_as_float(v.y) + res[1]);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(v) : "memory");
            }
        }
        // the four stores back to back (a load between two stores would wait for the first)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + er0 + p * 32;
            if (m < a.M) *(uint4*)(gout + (size_t)m * 64 + ec * 8) = vq[p];
        }
        if (MODE == MODE_FWD && (a.stats || a.sacc.acc)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                tsum[e] += ssum[e];
                tsq[e] += ssq[e];
            }
        }
        C64_SEG(5)
        __syncthreads();  // the staged tile has been read: its buffer is the next prefetch's target
        if (nbuf == 1 && more) {  // one buffer: the next tile's slab now, waited for at once (the CU's other block works meanwhile)
            load_slab(0, (tile + bpx) * BM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        C64_SEG(6)
    }
#ifdef GDL_TIMING
    GDL_STAMP(2);
    if (a.dbg && threadIdx.x == 0) {
        unsigned long long* d = a.dbg + ((size_t)(1 << 15) + blockIdx.x) * 8;
        for (int i = 0; i < 7; ++i) d[i] = tq[i];
        d[7] = tq_n;
    }
#endif
    if constexpr (BW) {
        __syncthreads();
        if (tid < 128) {  // the four waves' rows ([k][64] floats each) in fixed order -> the partial row's [64][2]
            const float* ar = (const float*)(smem + nbuf * slab_bytes + 1536) + (tid & 1) * 64 + (tid >> 1);
            st_agent(a.bw_partial + (size_t)blockIdx.x * 128 + tid, ((ar[0] + ar[128]) + ar[256]) + ar[384]);
        }
    }
    if (MODE == MODE_FWD && (a.stats || a.sacc.acc)) {
        float* const sdst = a.stats;
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ssum[e] = tsum[e];
            ssq[e] = tsq[e];
        }
        // lanes with equal (lane & 7) hold the same 8 channels: fold them, then the 4 waves in fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e)
            for (int msk = 8; msk < 64; msk <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], msk);
                ssq[e] += __shfl_xor(ssq[e], msk);
            }
        float* red = (float*)smem;  // [4 waves][64][2]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 64 + lane * 8 + e) * 2 + 0] = ssum[e];
                red[(wave * 64 + lane * 8 + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, w = tid & 1;
            const float s2 = ((red[(0 * 64 + c) * 2 + w] + red[(1 * 64 + c) * 2 + w]) + red[(2 * 64 + c) * 2 + w]) +
                             red[(3 * 64 + c) * 2 + w];
            if (a.sacc.acc)
                bn_acc_add(a.sacc, c, w, s2);
            else
                st_agent(sdst + ((size_t)blockIdx.x * 64 + c) * 2 + w, s2);
        }
    }
    GDL_STAMP(3);
}

static size_t slab_lds_bytes(int BM, int BN, int W, int IC, int dtype, bool single = false) {
    const int bke = dtype == GDL_BF16 ? 64 : 32, esz = dtype == GDL_BF16 ? 2 : 4;
    const int rows = BM + 2 * W + 2;
    const size_t slab = (size_t)((rows + 7) / 8) * 1024;
    const size_t main = 2 * (size_t)BN * 128 + ((IC / bke > 1 && !single) ? 2 : 1) * slab + ((rows & 7) ? 0 : 1024);
    const size_t epi = (size_t)BM * (BN * esz + 16) + 8 * (size_t)BN * 2 * 4;
    return main > epi ? main : epi;
}

// ---------------------------------------------------------------- host side
struct TileCfg {
    int bm, bn;
};

static int forced_cfg() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_CONV_CFG");  // tuning aid: 1 = 256x64, 3 = 64x64, 4 = 128x64
        v = e ? atoi(e) : 0;
    }
    return v;
}

static TileCfg pick_cfg(int M, int OC, int dtype) {
    (void)dtype;
    switch (forced_cfg()) {
        case 1: return {256, 64};
        case 3: return {64, 64};
        case 4: return {128, 64};
        default: break;
    }
    // Measured on MI355X (tools/bench_conv.py, CREMA-D B=64 shapes): 256x64 wins whenever it still
    // yields >= ~160 blocks, also for wide layers (the A panel re-read per N-tile is served by L2);
    // below that the smaller M-tiles fill the chip better.
    // (a 128 x 128 flat tile for the wide stride-2 layers -- 10 % faster alone, 0.4 % slower in the step -- was removed in round 4:
    // tools/experiments/r4_pruned_knobs.diff.txt)
    const long b256 = (long)((M + 255) / 256) * (OC / 64);
    if (b256 >= 160) return {256, 64};
    const long b128 = (long)((M + 127) / 128) * (OC / 64);
    if (b128 >= 160) return {128, 64};
    return {64, 64};
}

template <typename T, int BM, int BN, int WM, int WN, int MODE>
static int launch_one(ConvArgs& a, hipStream_t st) {
    using SM = ConvSmem<BM, BN, T>;
    a.mtiles = ceil_div(a.M, BM);
    auto kfn = conv_igemm_kernel<T, BM, BN, WM, WN, MODE>;
    static DevOnce attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv_igemm)");
        attr_set = true;
    }
    const int ntn = a.OC / BN;
    const int grid = xcd_grid(a.mtiles * ntn);
    static char pname[96] = "";
    if (!pname[0])
        snprintf(pname, sizeof(pname), "gdl::conv_igemm_kernel<%s, %d, %d, %d, %d, %d>", prof_tname<T>(), BM, BN, WM, WN, MODE);
    ProfScope prof(pname, PROF_MFMA, st, a.flops, true, a.hbm_bytes);
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(256), SM::BYTES, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("conv_igemm_kernel");
    return GDL_OK;
}

template <typename T, int BM, int BN, int MODE, int NWV = 4>
static int launch_slab(ConvArgs& a, size_t lds, hipStream_t st) {
    a.mtiles = ceil_div(a.M, BM);
    auto kfn = conv3x3_slab_kernel<T, BM, BN, MODE, NWV>;
    static DevOnce attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv3x3_slab)");
        attr_set = true;
    }
    const int grid = xcd_grid(a.mtiles * (a.OC / BN) * (a.ksplit > 1 ? a.ksplit : 1));
    static char pname[96] = "";
    if (!pname[0])
        snprintf(pname, sizeof(pname), "gdl::conv3x3_slab_kernel<%s, %d, %d, %d, %d>", prof_tname<T>(), BM, BN, MODE, NWV);
    ProfScope prof(pname, PROF_MFMA, st, a.flops, true, a.hbm_bytes);
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(NWV * 64), lds, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("conv3x3_slab_kernel");
    return GDL_OK;
}

// which kernel / tile a convolution runs with (shared by the launcher and by the BatchNorm partial count)
struct ConvPlan {
    int slab;  // 1: conv3x3_slab_kernel
    int bm, bn;
    size_t lds;
    int single;  // slab kernel: one slab buffer
    int c64;     // 1: conv3x3_c64_kernel (persistent; BatchNorm partial rows = C64_GRID)
    int nwv8;    // slab kernel: the 128 x 128 tile on 512 threads (small layers: one block per CU at most)
    int pslab;   // 1: conv3x3_pslab_kernel (persistent, pipelined; round 6) -- bm / bn / single / lds describe ITS launch
    int grid;    // pslab: blocks of the launch = BatchNorm partial rows
};
static size_t c64_lds_bytes(int W, bool single = false) {
    // slabs, zero KiB, BW: constants (512 B) + accumulator rows (2 KiB)
    const size_t b = (single ? 1 : 2) * (size_t)((C64_BM + 2 * W + 2 + 7) / 8) * 1024 + 1024 + 512 + 2048;
    return b > (size_t)72 * 1024 ? b : (size_t)72 * 1024;  // the prologue stages the 72 KiB filter through the same LDS
}
static bool c64_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_C64");  // tuning aid: 0 = the slab kernel for the 64 -> 64 channel layers too
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}
static int slab_cfg() {
    static int v = -1;
    if (v < 0) {
        // tuning aid, bits: 1 = allow the 256 x 128 single-slab tile, 2 = the 192 x 128 one.  Re-swept inside the step after the
        // layer-1 / stem kernels became persistent (tools/ab_env.sh, 4 A/B rounds): 192-row tiles 5.85 ms, both 5.86, 256-row 5.91
        const char* e = tune_env("GDL_SLAB_CFG");
        v = e ? atoi(e) : 2;
    }
    return v;
}
// fewest blocks for which the 192 / 256 x 128 tiles are used, and for which the 128-channel tile is used at all: the small layers
// (visual layer 4, audio layers 3 / 4) keep 64-wide tiles or the 8-wave form -- twice the blocks (knob sweeps of rounds 2-4: 100 ->
// 384 is -0.5 % step time, the defaults stayed twice; GDL_SLAB_BIG_MIN / GDL_SLAB_BN128_MIN: tools/experiments/r5_pruned_knobs.diff.txt)
static constexpr long slab_big_min() { return 128; }
static constexpr long slab_bn128_min() { return 384; }
static ConvPlan plan_conv_base(int dtype, int M, int OC, int IC, int W, int R, int S, int stride, int pad, bool allow_nwv8 = true) {
    ConvPlan p{};
    static int noslab = -1;
    if (noslab < 0) {
        const char* e = tune_env("GDL_CONV_NOSLAB");  // tuning aid
        noslab = e ? atoi(e) : 0;
    }
    static int slab_bm = -1;
    if (slab_bm < 0) {
        const char* e = tune_env("GDL_SLAB_BM");  // tuning aid: force the slab kernel's M-tile (128 / 256)
        slab_bm = e ? atoi(e) : 0;
    }
    // LDS budget of the slab kernel: 80 KB (two blocks per CU); a layer too wide for that (the 79-pixel audio layer 2 of the
    // Kinetics-Sounds shapes) still runs better on it with one block per CU than on the flat kernel (+0.7 % of that step)
    // tuning aid: GDL_PLAN="M:OC:BM:BN,..." forces the slab tile of the stride-1 3x3 layers with that many GEMM rows and
    // output channels (forward and data gradient share it); tools/plan_search.py walks it
    // 64 -> 64 channels (layer 1): the persistent weights-in-registers kernel, two blocks per CU (80 KB of LDS each at most;
    // the staged output tile, 128 x 144 bytes, has to fit into one slab buffer)
    if (!noslab && c64_enabled() && dtype == GDL_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && IC == 64 && OC == 64 &&
        c64_lds_bytes(W, true) <= (size_t)80 * 1024 && (size_t)((C64_BM + 2 * W + 2 + 7) / 8) * 1024 >= (size_t)C64_BM * 144 &&
        M >= 64 * C64_BM) {
        // (images too wide for two slab buffers in 80 KB -- the 157-pixel audio layer 1 of the Kinetics-Sounds / VGGSound
        // shapes -- run it with one)
        p.single = c64_lds_bytes(W) > (size_t)80 * 1024;
        static int wide = -1;
        if (wide < 0) {
            const char* e = tune_env("GDL_C64_WIDE");  // tuning aid: 0 = wide images stay on the slab kernel
            wide = e ? atoi(e) : 1;
        }
        if (p.single && !wide) {
            p.single = 0;
            goto no_c64;
        }
        p.slab = 1, p.c64 = 1, p.bm = C64_BM, p.bn = 64, p.lds = c64_lds_bytes(W, p.single != 0);
        return p;
    }
no_c64:
    if (!noslab && R == 3 && S == 3 && stride == 1 && pad == 1) {
        static const char* plan_env = tune_env("GDL_PLAN");
        for (const char* q = plan_env; q && *q;) {
            int m = 0, oc = 0, bm = 0, bn = 0;
            if (sscanf(q, "%d:%d:%d:%d", &m, &oc, &bm, &bn) == 4 && m == M && oc == OC && OC % bn == 0 &&
                (bm == 128 || bm == 256 || (bm == 192 && bn == 128)) && (bn == 64 || bn == 128) && (dtype == GDL_BF16 || bm != 192)) {
                const bool single = bn == 128 && bm != 128;
                const size_t lds = slab_lds_bytes(bm, bn, W, IC, dtype, single);
                if (lds <= (size_t)112 * 1024 && (dtype == GDL_BF16 || !single)) {
                    p.slab = 1, p.bm = bm, p.bn = bn, p.single = single, p.lds = lds;
                    return p;
                }
            }
            q = strchr(q, ',');
            if (q) ++q;
        }
    }
    for (const size_t slab_cap : {(size_t)80 * 1024, (size_t)112 * 1024})
    if (!noslab && R == 3 && S == 3 && stride == 1 && pad == 1) {
        // Measured end to end (bench.py, four streams sharing the CUs): the 128-row tile (48 KB of LDS, three
        // blocks per CU) beats the 256-row one (64 KB, two) by ~1 %, although they tie when run alone.
        static int slab_bn = -1;
        if (slab_bn < 0) {
            const char* e = tune_env("GDL_SLAB_BN");  // tuning aid: 64 = never use the 128-channel tile
            slab_bn = e ? atoi(e) : 0;
        }
        // 128 x 128 tile when the layer is wide enough and still yields a block per CU: the slab is fetched
        // once per 128 output channels and a K-step carries twice the MFMAs for the same barrier / DMA issue
        // 128-channel tiles of 192 / 256 rows with ONE slab buffer (bf16): more MFMAs per barrier / weight-tile DMA /
        // address arithmetic than the 128 x 128 tile and fewer fragment bytes per MFMA (PMC of the 128 x 128 kernel:
        // MFMA 24 % of a wave's cycles, the rest waits, LDS issue stalls and scalar / vector address work); two blocks
        // per CU still fit because the next chunk's slab is not double buffered.  Which M-tile wins is mostly a matter
        // of how the tile count divides by the 512 block slots: measured block times fit T(BM) ~ 130 + BM (256 x 128:
        // 26.5 us against 17.7 us for 128 x 128 at 128 channels), so pick the smallest rounds(BM) * (130 + BM).
        if (slab_cfg() != 0 && !slab_bm && dtype == GDL_BF16 && OC % 128 == 0 && slab_cap == (size_t)80 * 1024) {
            int best_bm = 0;
            double best = 0.0;
            for (int bm : {128, 192, 256}) {
                if (bm == 192 && !(slab_cfg() & 2)) continue;
                if (bm == 256 && !(slab_cfg() & 1)) continue;
                const bool single = bm != 128;
                const size_t lds = slab_lds_bytes(bm, 128, W, IC, dtype, single);
                if (lds > slab_cap) continue;
                const long blocks = (long)((M + bm - 1) / bm) * (OC / 128);
                if (bm != 128 && blocks < slab_big_min()) continue;
                if (bm == 128 && blocks < slab_bn128_min()) continue;
                const double rounds = (double)((blocks + 511) / 512);
                const double cost = rounds * (130.0 + bm);
                if (!best_bm || cost < best) {
                    best = cost;
                    best_bm = bm;
                }
            }
            if (best_bm) {
                p.slab = 1;
                p.bm = best_bm;
                p.bn = 128;
                p.single = best_bm != 128;
                p.lds = slab_lds_bytes(best_bm, 128, W, IC, dtype, p.single != 0);
                return p;
            }
        }
        static int small8 = -1;
        if (small8 < 0) {
            // layers too small for 384 blocks of 128 x 128 (here: audio layer 4, M = 3 456) on the 8-wave form of that tile
            // instead of 128 x 64 tiles: half the blocks, half the filter re-reads.  Alone it is SLOWER (46 -> 54 us), inside the
            // step faster (5.73 -> 5.67 ms, three A/B rounds): tuning aid, 0 = off
            const char* e = tune_env("GDL_SLAB_SMALL8");
            small8 = e ? atoi(e) : 1;
        }
        if (small8 && allow_nwv8 && !slab_bm && dtype == GDL_BF16 && OC % 128 == 0 && slab_cap == (size_t)80 * 1024) {
            const size_t lds = slab_lds_bytes(128, 128, W, IC, dtype);
            const long blocks = (long)((M + 127) / 128) * (OC / 128);
            if (lds <= slab_cap && blocks < slab_bn128_min()) {
                p.slab = 1, p.bm = 128, p.bn = 128, p.lds = lds, p.nwv8 = 1;
                return p;
            }
        }
        if (!slab_bm && slab_bn != 64 && OC % 128 == 0) {
            const size_t lds = slab_lds_bytes(128, 128, W, IC, dtype);
            const long blocks = (long)((M + 127) / 128) * (OC / 128);
            if (lds <= slab_cap && blocks >= slab_bn128_min()) {
                p.slab = 1;
                p.bm = 128;
                p.bn = 128;
                p.lds = lds;
                return p;
            }
        }
        for (int bm : {128, 256}) {
            if (slab_bm && bm != slab_bm) continue;
            const size_t lds = slab_lds_bytes(bm, 64, W, IC, dtype);
            const long blocks = (long)((M + bm - 1) / bm) * (OC / 64);
            if (lds <= slab_cap && (blocks >= 160 || bm == 128)) {
                p.slab = 1;
                p.bm = bm;
                p.bn = 64;
                p.lds = lds;
                return p;
            }
        }
    }
    TileCfg c = pick_cfg(M, OC, dtype);
    // plain GEMMs (1x1, stride 1: the Swin encoder's Linears -- the ResNets have none): the 128x128 tile is the faster one
    // alone (8 DMA pieces per 32 MFMAs instead of 10, each A row gathered for half as many N-tiles), and these run without
    // a second MFMA-bound stream beside them
    if (R == 1 && S == 1 && stride == 1 && forced_cfg() == 0 && OC % 128 == 0 && (long)((M + 127) / 128) * (OC / 128) >= 256)
        c = {128, 128};
    static int n64bm = -1;
    if (n64bm < 0) {
        const char* e = tune_env("GDL_PLAIN_N64_BM");  // tuning aid: M-tile of plain GEMMs whose width is not a multiple of 128
        n64bm = e ? atoi(e) : 0;
    }
    if (R == 1 && S == 1 && stride == 1 && forced_cfg() == 0 && OC % 128 != 0 && n64bm == 128 && c.bm == 256) c = {128, 64};
    p.slab = 0;
    p.bm = c.bm;
    p.bn = c.bn;
    return p;
}

// Round 6: the 128-channel-wide slab tiles of 128 / 192 rows (four waves, bf16) run on the persistent pipelined kernel
// (conv_pslab.h) when its LDS fits: two slab buffers where they fit into 80 KiB (two blocks per CU) -- 96 KiB for a launch that
// cannot fill two blocks per CU anyway (<= 256 items) --, else one.  (A launch of <= 256 blocks could take 160 KiB for itself; in
// the step the LDS it leaves to the CU's other kernels is worth more than the slab reloads it saves: profiles/r06_ab_pslab.txt.)
static int pslab_mode() {
    static int v = -1;
    if (v < 0) {
        // tuning aid: 0 = off (conv3x3_slab_kernel), 1 = the 128 / 192-row tiles, 2 (default) = also in place of the 8-wave 128 x 128
        // tile of the smallest layers (audio layer 4: 108 blocks; step 5.09 -> 5.05 ms, three A/B rounds, profiles/r06_ab_pslab.txt)
        const char* e = tune_env("GDL_PSLAB");
        v = e ? atoi(e) : 2;
        const char* sk = tune_env("GDL_SPLITK");  // (the split-K alternative path keeps the kernel it was written for)
        if (sk && atoi(sk)) v = 0;
    }
    return v;
}
static ConvPlan plan_conv(int dtype, int M, int OC, int IC, int W, int R, int S, int stride, int pad, bool allow_nwv8 = true) {
    ConvPlan p = plan_conv_base(dtype, M, OC, IC, W, R, S, stride, pad, allow_nwv8);
    if (!pslab_mode() || !p.slab || p.c64 || dtype != GDL_BF16 || p.bn != 128 || (p.bm != 128 && p.bm != 192)) return p;
    if (p.nwv8 && pslab_mode() < 2) return p;
    const int items = ceil_div(M, p.bm) * (OC / 128);
    if ((p.bm + 2 * W + 2 + 7) / 8 > 4 * PS_SLAB_PIECES) return p;  // a slab buffer of at most 32 KiB
    static int big_kb = -1;
    if (big_kb < 0) {
        // tuning aid: LDS budget (KiB) of a launch that cannot fill two blocks per CU anyway -- with 160 it takes two slab buffers
        // (~100 KiB) and leaves the CU's other kernels 60 KiB; 80 = one slab buffer there too
        const char* e = tune_env("GDL_PSLAB_LDS");
        big_kb = e ? atoi(e) : 96;  // (step: 160 -> 4.98 ms, 80 -> 4.94, 96 -- two buffers for the 128-row tile of the audio layer 4 only -> 4.92; three A/B rounds)
    }
    const size_t budget = items <= 256 ? (size_t)big_kb * 1024 : (size_t)80 * 1024;
    int nslab = 0;
    if (IC > 64 && pslab_lds_bytes(p.bm, W, 2) <= budget)
        nslab = 2;
    else if (pslab_lds_bytes(p.bm, W, 1) <= budget)
        nslab = 1;
    if (!nslab) return p;
    // (the 128-row tile of the audio layer 2 -- 396 blocks, two per CU -- had two slab buffers in 80 KiB; with this kernel's staging
    // and sums rows there is room for one: 23 vs 24 us alone, tools/bench_conv.py -- it keeps the round-5 kernel)
    if (nslab == 1 && !p.single && IC > 64 && pslab_mode() < 3) return p;  // (GDL_PSLAB=3, tuning aid: the persistent kernel there too -- a tie in the step)
    p.pslab = 1, p.nwv8 = 0, p.single = nslab == 1, p.lds = pslab_lds_bytes(p.bm, W, nslab), p.grid = pslab_grid(items);
    return p;
}

template <int MI, int MODE, bool TWO, bool RICH = false>
static int launch_pslab(ConvArgs& a, const ConvPlan& pl, hipStream_t st) {
    // (data gradients with an addend or a second BatchNorm partner: the instantiation that carries their registers)
    if constexpr (MODE == MODE_DGRAD && !RICH)
        if (a.addend || a.bw_y2) return launch_pslab<MI, MODE, TWO, true>(a, pl, st);
    a.mtiles = ceil_div(a.M, 64 * MI);
    GDL_REQUIRE(!a.bias && !a.gelu_out && !a.gelu_u && !a.orow && a.ksplit <= 1 && !a.seg_Q,
                "conv: unsupported option for the persistent slab kernel");
    GDL_REQUIRE(MODE == MODE_DGRAD || (!a.addend && !a.relu_bits), "conv: the persistent slab forward has no addend / ReLU bits");
    GDL_REQUIRE(a.IC % 64 == 0 && a.OC % 128 == 0 && a.ntaps == 9, "conv: persistent slab kernel shape");
#ifdef GDL_TIMING
    {
        const char* e = getenv("GDL_PSLAB_DBG");
        a.dbg_mode = e ? atoi(e) : 0;
    }
#endif
    auto kfn = conv3x3_pslab_kernel<MI, MODE, TWO, RICH>;
    static DevOnce attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv3x3_pslab)");
        attr_set = true;
    }
    static char pname[64] = "";
    if (!pname[0]) snprintf(pname, sizeof(pname), "gdl::conv3x3_pslab_kernel<%d, %d, %d, %d>", 64 * MI, MODE, TWO ? 2 : 1, RICH ? 1 : 0);
    ProfScope prof(pname, PROF_MFMA, st, a.flops, true, a.hbm_bytes);
    hipExtLaunchKernelGGL(kfn, dim3(pl.grid), dim3(256), pl.lds, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("conv3x3_pslab_kernel");
    return GDL_OK;
}

template <int MODE, bool BW = false, bool ADD = false>
static int launch_c64(ConvArgs& a, size_t lds, hipStream_t st) {
    a.mtiles = ceil_div(a.M, C64_BM);
    GDL_REQUIRE(!a.bias && !a.gelu_out && !a.orow && !a.bw_y2, "conv: unsupported option for the 64-channel persistent kernel");
    GDL_REQUIRE(MODE == MODE_DGRAD || !a.addend, "conv: the 64-channel persistent forward kernel has no addend");
    if constexpr (MODE == MODE_DGRAD && !BW)
        if (a.bw_y) return launch_c64<MODE, true, ADD>(a, lds, st);
    if constexpr (MODE == MODE_DGRAD && !ADD)
        if (a.addend) return launch_c64<MODE, BW, true>(a, lds, st);
    if constexpr (BW) {  // (slab buffers of at least C64_F32_TILE bytes: see the kernel)
        const size_t sl = (size_t)((a.slab_rows + 7) / 8) * 1024;
        const size_t need = (a.single_slab ? 1 : 2) * (sl > (size_t)C64_F32_TILE ? sl : (size_t)C64_F32_TILE) + 1024 + 512 + 2048;
        if (need > lds) lds = need;
        GDL_REQUIRE(lds <= (size_t)80 * 1024, "conv: LDS of the 64-channel data gradient with BatchNorm sums");
    }
    auto kfn = conv3x3_c64_kernel<MODE, BW, ADD>;
    static DevOnce attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv3x3_c64)");
        attr_set = true;
    }
    static char pname[64] = "";
    if (!pname[0]) snprintf(pname, sizeof(pname), BW ? "gdl::conv3x3_c64_kernel<%d, true>" : "gdl::conv3x3_c64_kernel<%d>", MODE);
    ProfScope prof(pname, PROF_MFMA, st, a.flops, true, a.hbm_bytes);
    hipExtLaunchKernelGGL(kfn, dim3(C64_GRID), dim3(256), lds, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("conv3x3_c64_kernel");
    return GDL_OK;
}

template <typename T, int MODE>
static int launch_mode(ConvArgs& a, const ConvPlan& pl, hipStream_t st) {
    if (pl.c64) return launch_c64<MODE>(a, pl.lds, st);
    if constexpr (std::is_same<T, bf16>::value)
        if (pl.pslab) {
            if (pl.single) return pl.bm == 192 ? launch_pslab<3, MODE, false>(a, pl, st) : launch_pslab<2, MODE, false>(a, pl, st);
            return pl.bm == 192 ? launch_pslab<3, MODE, true>(a, pl, st) : launch_pslab<2, MODE, true>(a, pl, st);
        }
    if (pl.slab) {
        if constexpr (std::is_same<T, bf16>::value)
            if (pl.bn == 128 && pl.bm == 256) return launch_slab<T, 256, 128, MODE>(a, pl.lds, st);
        if constexpr (std::is_same<T, bf16>::value)
            if (pl.bn == 128 && pl.bm == 192) return launch_slab<T, 192, 128, MODE>(a, pl.lds, st);
        if constexpr (std::is_same<T, bf16>::value)
            if (pl.bn == 128 && pl.nwv8) return launch_slab<T, 128, 128, MODE, 8>(a, pl.lds, st);
        if (pl.bn == 128) return launch_slab<T, 128, 128, MODE>(a, pl.lds, st);
        if (pl.bm == 256) return launch_slab<T, 256, 64, MODE>(a, pl.lds, st);
        return launch_slab<T, 128, 64, MODE>(a, pl.lds, st);
    }
    if (pl.bn == 128) return launch_one<T, 128, 128, 2, 2, MODE>(a, st);
    if (pl.bm == 256) return launch_one<T, 256, 64, 4, 1, MODE>(a, st);
    if (pl.bm == 128) return launch_one<T, 128, 64, 2, 2, MODE>(a, st);
    return launch_one<T, 64, 64, 2, 2, MODE>(a, st);
}

// number of M-tiles (= BatchNorm partial rows) the forward kernel of this convolution produces
int conv_tiles_m(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    const ConvPlan pl = plan_conv(dtype, N * P * Q, K, C, W, R, S, stride, pad);
    return pl.c64 ? C64_GRID : pl.pslab ? pl.grid : ceil_div(N * P * Q, pl.bm);
}

// true if the forward of this convolution runs on a persistent kernel (one BatchNorm partial row per block): the in-launch
// finalize costs a ticket per block LIFE there and is used by default
bool conv_fwd_persistent(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    const ConvPlan pl = plan_conv(dtype, N * P * Q, K, C, W, R, S, stride, pad);
    return pl.c64 != 0 || pl.pslab != 0;
}
// partial rows a data gradient with BatchNorm-backward statistics (ops.h BwdStats) writes
int conv_dgrad_tiles_m(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    const ConvPlan pl = plan_conv(dtype, N * H * W, C, K, W, R, S, stride, pad);
    if (pl.c64) return C64_GRID;
    if (pl.pslab) return pl.grid;
    const int rows = stride == 2 ? dgrad_perm_rows(N, H, W, pl.bm) : N * H * W;
    return ceil_div(rows, pl.bm);
}
// M-tile of a data gradient (the permuted stride-2 table is laid out for it)
int conv_dgrad_bm(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    return plan_conv(dtype, N * H * W, C, K, W, R, S, stride, pad).bm;
}

// ---------------------------------------------------------------- split-K finish
// One block per M-tile of the producing launch (so the per-tile statistics rows keep their count): out[row][c] = what
// conv_epilogue would have stored from the sum of the `nsplit` fp32 partial tensors, folded in split order (deterministic):
// rounded to T, + addend (rounded again, as the epilogue does), ReLU bits, then either the forward statistics (partial row or
// integer accumulators) or the BatchNorm-backward sums against one or two partner tensors.  A thread owns one 16-byte channel
// vector and every (256 / vectors-per-row)-th row of the tile; the row groups are folded through LDS in index order.
struct SplitFinArgs {
    const float* ws;
    int nsplit, M, OC, BM;
    void* out;
    const void* addend;
    const uint8_t* relu_bits;
    float* stats;
    BnAcc sacc;
    const void* bw_y;
    const float *bw_mean, *bw_rstd;
    float* bw_partial;
    const void* bw_y2;
    const float *bw_mean2, *bw_rstd2;
    float* bw_partial2;
};
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(SplitFinArgs f) {
    constexpr int EPC = TT<T>::EPC;
    // grid (M-tiles, OC / 64): a block owns 64 channels of a tile's rows -- 8 vectors per row, 32 rows per trip; one block per
    // M-tile (27 blocks for the audio layer 4) was latency-bound: 60 us for 7 MB
    constexpr int FC = 64, cpr = FC / EPC, rpi = 256 / cpr;
    __shared__ float red[rpi * FC * 3];  // [row group][channel][sum]
    const int vc = threadIdx.x % cpr, rsub = threadIdx.x / cpr, c0 = blockIdx.y * FC + vc * EPC;
    const int mtile = blockIdx.x, r0 = mtile * f.BM, r1 = min(f.M, r0 + f.BM);
    const bool st = f.stats != nullptr || f.sacc.acc != nullptr, bw = f.bw_y != nullptr, bw2 = bw && f.bw_y2 != nullptr;
    float s1[EPC], s2[EPC], s3[EPC], mu[EPC], mu2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        s1[e] = s2[e] = s3[e] = 0.f;
        mu[e] = bw ? f.bw_mean[c0 + e] : 0.f;
        mu2[e] = bw2 ? f.bw_mean2[c0 + e] : 0.f;
    }
    const size_t plane = (size_t)f.M * f.OC;
    for (int row = r0 + rsub; row < r1; row += rpi) {
        const size_t off = (size_t)row * f.OC + c0;
        float v[EPC];
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
            float4 a = *(const float4*)(f.ws + off + 4 * q);
            for (int sp = 1; sp < f.nsplit; ++sp) {
                const float4 b = *(const float4*)(f.ws + sp * plane + off + 4 * q);
                a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
            }
            v[4 * q] = a.x, v[4 * q + 1] = a.y, v[4 * q + 2] = a.z, v[4 * q + 3] = a.w;
        }
        uint4 pk = pack16<T>(v);  // (the epilogue stages the tile through LDS as T)
        if (f.addend) {
            float g[EPC];
            unpack16<T>(pk, v);
            unpack16<T>(*(const uint4*)((const T*)f.addend + off), g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += g[e];
            pk = pack16<T>(v);
        }
        unpack16<T>(pk, v);
        if (f.relu_bits) {
            const unsigned mk = f.relu_bits[off / EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = ((mk >> e) & 1u) ? v[e] : 0.f;
            pk = pack16<T>(v);
        }
        if (st) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * v[e];
            }
        }
        if (bw) {
            float y[EPC];
            unpack16<T>(*(const uint4*)((const T*)f.bw_y + off), y);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * (y[e] - mu[e]);
            }
            if (bw2) {
                unpack16<T>(*(const uint4*)((const T*)f.bw_y2 + off), y);
#pragma unroll
                for (int e = 0; e < EPC; ++e) s3[e] += v[e] * (y[e] - mu2[e]);
            }
        }
        *(uint4*)((T*)f.out + off) = pk;
    }
    if (!st && !bw) return;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        float* d = red + (rsub * FC + vc * EPC + e) * 3;
        d[0] = s1[e], d[1] = s2[e], d[2] = s3[e];
    }
    __syncthreads();
    if (threadIdx.x < FC) {
        const int cl = threadIdx.x, c = blockIdx.y * FC + cl;
        float t1 = red[cl * 3], t2 = red[cl * 3 + 1], t3 = red[cl * 3 + 2];
        for (int r = 1; r < rpi; ++r) {
            const float* d = red + (r * FC + cl) * 3;
            t1 += d[0], t2 += d[1], t3 += d[2];
        }
        if (st) {
            if (f.sacc.acc) {
                bn_acc_add(f.sacc, c, 0, t1);
                bn_acc_add(f.sacc, c, 1, t2);
            } else {
                f.stats[((size_t)mtile * f.OC + c) * 2 + 0] = t1;
                f.stats[((size_t)mtile * f.OC + c) * 2 + 1] = t2;
            }
        }
        if (bw) {
            f.bw_partial[((size_t)mtile * f.OC + c) * 2 + 0] = t1;
            f.bw_partial[((size_t)mtile * f.OC + c) * 2 + 1] = t2 * f.bw_rstd[c];
            if (bw2) {
                f.bw_partial2[((size_t)mtile * f.OC + c) * 2 + 0] = t1;
                f.bw_partial2[((size_t)mtile * f.OC + c) * 2 + 1] = t3 * f.bw_rstd2[c];
            }
        }
    }
}
// splits of a slab launch: only where the tiles leave CUs idle and the K-loop is long (bf16, 128 / 256 / 512 output channels)
static int splitk_threshold() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_SPLITK_BLOCKS");  // tuning aid: the block count below which a slab launch splits K when it is given a workspace
        v = e ? atoi(e) : 200;
    }
    return v;
}
static int plan_ksplit(const ConvPlan& pl, int dtype, int M, int OC, int IC) {
    if (!pl.slab || pl.c64 || pl.pslab || dtype != GDL_BF16 || splitk_threshold() <= 0) return 1;
    if (OC != 128 && OC != 256 && OC != 512) return 1;
    const long blocks = (long)ceil_div(M, pl.bm) * (OC / pl.bn);
    if (blocks >= splitk_threshold()) return 1;
    const int kpt = IC / 64;
    int s = 1;
    while (s < 4 && kpt % (2 * s) == 0 && kpt / (2 * s) >= 1 && blocks * (2 * s) <= 512) s *= 2;
    return s;
}
size_t conv_split_ws_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dgrad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    const int M = dgrad ? N * H * W : N * P * Q, OC = dgrad ? C : K, IC = dgrad ? K : C;
    // (split-K belongs to the round-5 slab kernel: a launch that is given a workspace runs on it, below)
    const ConvPlan pl = plan_conv_base(dtype, M, OC, IC, W, R, S, stride, pad);
    const int s = plan_ksplit(pl, dtype, M, OC, IC);
    return s > 1 ? (size_t)s * M * OC * sizeof(float) : 0;
}

static int run_conv(int mode, int dtype, const void* in, const void* wt, void* out, const void* addend, float* stats,
                    const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                    hipStream_t st,
                    const uint8_t* relu_bits = nullptr, const void* dy_ds = nullptr, const void* w_ds = nullptr,
                    const float* bias = nullptr, void* gelu_out = nullptr, const BwdStats* bw = nullptr,
                    const BnAcc* sacc = nullptr, const void* gelu_u = nullptr, const SplitWs* split = nullptr) {
    GDL_REQUIRE(dtype == GDL_BF16 || dtype == GDL_F32, "conv: bad dtype %d", dtype);
    GDL_REQUIRE(table, "conv: gather table is null (build it with gdl_conv_build_table)");
    const int bke = (dtype == GDL_BF16) ? 64 : 32;
    const int esz = (dtype == GDL_BF16) ? 2 : 4;
    GatherGeom g;
    int rc = gather_geom(mode, dtype, N, H, W, C, K, R, S, stride, pad, &g);
    if (rc) return rc;
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    ConvArgs a{};
#ifdef GDL_TIMING
    a.dbg = g_timing_buf;
#endif
    a.in = in;
    a.wt = wt;
    a.out = out;
    a.addend = addend;
    a.bias = bias;
    a.gelu_out = gelu_out;
    a.relu_bits = relu_bits;
    a.stats = stats;
    a.gelu_u = gelu_u;
    if (gelu_u) GDL_REQUIRE(mode == GATHER_DGRAD && !(bw && bw->y), "conv: the GELU derivative is a data-gradient option (without BatchNorm sums)");
    if (sacc && sacc->acc) {
        GDL_REQUIRE((mode == GATHER_FWD || gelu_u) && !stats, "conv: integer accumulators belong to the forward statistics and to the GELU data gradient's column sums (without partial rows)");
        // without a flag word the channel's SECOND accumulator word is the overflow mark (bnacc.h bn_acc_add): such an accumulator
        // must not also collect sums of squares
        GDL_REQUIRE(sacc->flag || sacc->s2 == 0.0, "conv: integer accumulators without a flag word take column sums only (s2 = 0)");
        a.sacc = *sacc;
    }
    if (bw && bw->y) {
        GDL_REQUIRE(mode == GATHER_DGRAD && bw->mean && bw->rstd && bw->partial &&
                        (!bw->y2 || (bw->mean2 && bw->rstd2 && bw->partial2)),
                    "conv: bad BatchNorm-backward statistics arguments");
        a.bw_y = bw->y, a.bw_mean = bw->mean, a.bw_rstd = bw->rstd;
        a.bw_partial = bw->partial;
        a.bw_y2 = bw->y2, a.bw_mean2 = bw->mean2, a.bw_rstd2 = bw->rstd2, a.bw_partial2 = bw->partial2;
    }
    a.table = (const GatherEntry*)table;
    a.plain = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
    a.M = g.rows;
    a.ntaps = g.ntaps;
    for (int t = 0; t < 9; ++t) a.delta[t] = g.delta[t];
    if (mode == GATHER_FWD) {
        a.IC = C;
        a.OC = K;
        a.in_bytes = (unsigned)((size_t)N * H * W * C * esz);
    } else {
        a.IC = K;
        a.OC = C;
        a.in_bytes = (unsigned)((size_t)N * P * Q * K * esz);
    }
    a.wt_bytes = (unsigned)((size_t)K * C * R * S * esz);
    GDL_REQUIRE(a.IC % bke == 0, "conv: gather channels %d not a multiple of %d", a.IC, bke);
    GDL_REQUIRE(a.OC % 64 == 0, "conv: output channels %d not a multiple of 64", a.OC);
    GDL_REQUIRE(a.M < (1 << 24), "conv: M = %d exceeds 2^24", a.M);
    // the short-K Linears of the Swin branch (plain GEMM, K = 128 / 192, many rows): the streaming kernel (linear_stream.hip)
    if (a.plain && !stats && !(sacc && sacc->acc) && !relu_bits && !(bw && bw->y) && !gelu_u && !dy_ds &&
        linear_stream_ok(dtype, a.M, a.IC, a.OC, addend != nullptr))
        return linear_stream_fwd(in, wt, out, addend, bias, gelu_out, a.M, a.IC, a.OC, st);
    // the gathered tensor has the output's spatial size for the stride-1 3x3 case the slab kernel serves
    ConvPlan pl = plan_conv(dtype, a.M, a.OC, a.IC, W, R, S, stride, pad);
    a.flops = 2.0 * (double)N * P * Q * K * C * R * S;  // the convolution's multiply-adds, whatever the direction
    {
        const double esz = dtype == GDL_BF16 ? 2.0 : 4.0;
        a.hbm_bytes = esz * ((double)N * H * W * C + (double)N * P * Q * K + (double)K * C * R * S);
    }
    if (dy_ds) {
        GDL_REQUIRE(mode == GATHER_DGRAD && stride == 2 && R == 3 && S == 3 && pad == 1 && w_ds && !pl.slab,
                    "conv: the folded downsample branch needs the 3x3 stride-2 pad-1 data gradient");
        a.in2 = dy_ds;
        a.wt2 = w_ds;
        a.in2_bytes = a.in_bytes;  // same [N][P][Q][K] shape as dy
        a.wt2_bytes = (unsigned)((size_t)K * C * esz);
        a.flops += 2.0 * (double)N * P * Q * K * C;
    }
    if (mode == GATHER_DGRAD && stride == 2) {
        a.orow = (const int*)((const GatherEntry*)table + dgrad_perm_cap(N, H, W));
        a.tile_taps = (const unsigned*)(a.orow + dgrad_perm_cap(N, H, W));
        a.tile_order = (const int*)(a.tile_taps + (dgrad_perm_cap(N, H, W) / 64 + 1));
        a.M = dgrad_perm_rows(N, H, W, pl.bm);
    }
    if (pl.slab) {
        a.W = W;
        a.single_slab = pl.single;
        a.in_pixels = N * H * W;
        a.slab_rows = pl.bm + 2 * W + 2;
        for (int r = 0; r < 3; ++r)
            for (int s2 = 0; s2 < 3; ++s2)
                a.pshift[r * 3 + s2] = mode == GATHER_FWD ? (r - 1) * W + (s2 - 1) : (1 - r) * W + (1 - s2);
    }
    // split-K (an alternative path, off in the engine by default): a launch that is given a workspace and would split runs on the
    // round-5 slab kernel (its plan), whatever the unsplit launch of this geometry runs on; the partial-row count stays the unsplit
    // form's (conv_tiles_m / conv_dgrad_tiles_m): the rows past the split form's M-tiles are zeroed
    const int rows_unsplit = pl.c64 ? C64_GRID : pl.pslab ? pl.grid : 0;
    int ks = 1;
    if (split && split->ptr && !bias && !gelu_out && !gelu_u && !dy_ds && !pl.c64) {
        const ConvPlan plb = pl.pslab ? plan_conv_base(dtype, a.M, a.OC, a.IC, W, R, S, stride, pad) : pl;
        ks = plan_ksplit(plb, dtype, a.M, a.OC, a.IC);
        if (ks > 1 && pl.pslab) {
            pl = plb;
            a.single_slab = pl.single;
            a.slab_rows = pl.bm + 2 * W + 2;
        }
    }
    if (ks > 1) {
        GDL_REQUIRE(split->bytes >= (size_t)ks * a.M * a.OC * sizeof(float), "conv: split-K workspace of %zu bytes, need %zu",
                    split->bytes, (size_t)ks * a.M * a.OC * sizeof(float));
        ConvArgs b = a;  // the producing launch leaves fp32 accumulators only
        b.ksplit = ks;
        b.split_ws = (float*)split->ptr;
        b.addend = nullptr, b.relu_bits = nullptr, b.stats = nullptr, b.sacc = BnAcc{nullptr, 0.0, 0.0, nullptr};
        b.bw_y = b.bw_y2 = nullptr;
        const int rc = mode == GATHER_FWD ? launch_mode<bf16, MODE_FWD>(b, pl, st) : launch_mode<bf16, MODE_DGRAD>(b, pl, st);
        if (rc != GDL_OK) return rc;
        SplitFinArgs f{};
        f.ws = b.split_ws, f.nsplit = ks, f.M = a.M, f.OC = a.OC, f.BM = pl.bm;
        f.out = a.out, f.addend = a.addend, f.relu_bits = a.relu_bits, f.stats = a.stats, f.sacc = a.sacc;
        f.bw_y = a.bw_y, f.bw_mean = a.bw_mean, f.bw_rstd = a.bw_rstd, f.bw_partial = a.bw_partial;
        f.bw_y2 = a.bw_y2, f.bw_mean2 = a.bw_mean2, f.bw_rstd2 = a.bw_rstd2, f.bw_partial2 = a.bw_partial2;
        ProfScope prof("gdl::splitk_finish_kernel", PROF_HBM, st, (double)a.M * a.OC * (4.0 * ks + 2.0 * (1 + (a.addend ? 1 : 0) + (a.bw_y ? 1 : 0) + (a.bw_y2 ? 1 : 0))));
        hipLaunchKernelGGL(splitk_finish_kernel<bf16>, dim3(ceil_div(a.M, pl.bm), a.OC / 64), dim3(256), 0, st, f);
        GDL_CHECK_LAUNCH("splitk_finish_kernel");
        const int mt = ceil_div(a.M, pl.bm);
        if (rows_unsplit > mt) {  // the caller sized its partial rows for the unsplit (persistent) form
            const size_t off = (size_t)mt * a.OC * 2, n = (size_t)(rows_unsplit - mt) * a.OC * 2 * sizeof(float);
            for (float* p : {a.stats, a.bw_partial, a.bw_y2 ? a.bw_partial2 : (float*)nullptr})
                if (p) {
                    hipError_t e = hipMemsetAsync(p + off, 0, n, st);
                    if (e != hipSuccess) return check_hip(e, "hipMemsetAsync(split-K partial rows)");
                }
        }
        return GDL_OK;
    }
    if (dtype == GDL_BF16)
        return mode == GATHER_FWD ? launch_mode<bf16, MODE_FWD>(a, pl, st) : launch_mode<bf16, MODE_DGRAD>(a, pl, st);
    return mode == GATHER_FWD ? launch_mode<float, MODE_FWD>(a, pl, st) : launch_mode<float, MODE_DGRAD>(a, pl, st);
}

int conv_fwd(int dtype, const void* x, const void* w, void* y, float* bn_partial, const void* table, int N, int H, int W,
             int C, int K, int R, int S, int stride, int pad, hipStream_t st, const BnAcc* sacc, const SplitWs* split) {
    return run_conv(GATHER_FWD, dtype, x, w, y, nullptr, bn_partial, table, N, H, W, C, K, R, S, stride, pad, st, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, sacc, nullptr, split);
}

// ---- direct stem forward (layout.hip / gather.h): implicit GEMM over the padded NHWC4 input
int stem_taps(int dtype);
int stem_ic(int dtype);
// the bf16 stem runs on the row-slab kernels when an output row has at least 64 pixels
static bool stem_rows(int dtype, int W) {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_STEM_ROWS_FWD");  // tuning aid: 0 = flat kernel
        v = e ? atoi(e) : 1;
    }
    return v != 0 && dtype == GDL_BF16 && (W - 1) / 2 + 1 >= 64;
}
static bool stem_pers() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_STEM_PERS");  // tuning aid: 0 = the one-tile-per-block row-slab kernel
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}
bool conv_stem_persistent(int dtype, int W) { return stem_rows(dtype, W) && stem_pers(); }
int conv_stem_tiles_m(int dtype, int n_img, int H, int W) {
    const int M = n_img * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    if (stem_rows(dtype, W))
        return stem_pers() ? SRP_GRID : ceil_div(n_img * ((H - 1) / 2 + 1) * ceil_div((W - 1) / 2 + 1, 64), 4);
    return ceil_div(M, pick_cfg(M, 64, dtype).bm);
}
int conv_stem_fwd(int dtype, const void* xp, const void* wp, void* y, float* bn_partial, const void* table, int n_img, int H,
                  int W, int Cin, hipStream_t st, const BnAcc* sacc) {
    GDL_REQUIRE(dtype == GDL_BF16 || dtype == GDL_F32, "stem: bad dtype %d", dtype);
    GDL_REQUIRE(xp && wp && y && table, "stem: null pointer");
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1, Hp = H + 6, Wp = W + 8;
    const int esz = dtype == GDL_BF16 ? 2 : 4, pix = 4 * esz;
    ConvArgs a{};
#ifdef GDL_TIMING
    a.dbg = nullptr;
#endif
    a.in = xp;
    a.wt = wp;
    a.out = y;
    a.stats = bn_partial;
    if (sacc && sacc->acc) {
        GDL_REQUIRE(!bn_partial, "stem: integer statistics accumulators exclude partial rows");
        GDL_REQUIRE(sacc->flag || sacc->s2 == 0.0, "stem: integer accumulators without a flag word take column sums only (s2 = 0)");
        a.sacc = *sacc;
    }
    a.table = (const GatherEntry*)table;
    a.M = n_img * P * Q;
    a.OC = 64;
    a.IC = stem_ic(dtype);
    a.ntaps = stem_taps(dtype);
    a.in_bytes = (unsigned)((size_t)n_img * Hp * Wp * pix);
    a.wt_bytes = (unsigned)((size_t)64 * a.ntaps * a.IC * esz);
    if (dtype == GDL_BF16) {
        a.split = 1;
        for (int t = 0; t < 4; ++t) {
            a.delta[t] = (2 * t) * Wp * pix;
            a.delta_hi[t] = 2 * t + 1 < 7 ? (2 * t + 1) * Wp * pix : 0x40000000;  // no 8th filter row: out of range -> zeros
        }
    } else {
        for (int t = 0; t < 7; ++t) a.delta[t] = t * Wp * pix;
    }
    GDL_REQUIRE(a.M < (1 << 24), "stem: M = %d exceeds 2^24", a.M);
    a.flops = 2.0 * (double)a.M * 64 * 49 * Cin;  // what the layer is worth, not the zero padding of the K-steps
    a.hbm_bytes = (dtype == GDL_BF16 ? 2.0 : 4.0) * ((double)a.M * 64 + (double)n_img * H * W * Cin + 64.0 * 49 * Cin);
    if (stem_rows(dtype, W)) {
        a.seg_Q = Q;
        a.seg_nseg = ceil_div(Q, 64);
        a.seg_stages = n_img * P * a.seg_nseg;
        a.mtiles = ceil_div(a.seg_stages, 4);
        GDL_REQUIRE((size_t)a.seg_stages * 64 < (1UL << 31), "stem: too many rows");
        if (stem_pers()) {
            static DevOnce attr_p;
            if (!attr_p) {
                hipError_t e = hipFuncSetAttribute((const void*)conv_stem_pers_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   8 * SRF_SLAB);
                if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv_stem_pers)");
                attr_p = true;
            }
            ProfScope prof("gdl::conv_stem_pers_kernel", PROF_MFMA, st, a.flops, true, a.hbm_bytes);
            hipExtLaunchKernelGGL(conv_stem_pers_kernel, dim3(SRP_GRID), dim3(256), 8 * SRF_SLAB, st, prof.e0(), prof.e1(), 0, a, P,
                                  Hp, Wp);
            GDL_CHECK_LAUNCH("conv_stem_pers_kernel");
            return GDL_OK;
        }
        static DevOnce attr_set;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)conv_stem_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               SRF_W + 4 * SRF_SLAB);
            if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv_stem_rows)");
            attr_set = true;
        }
        ProfScope prof("gdl::conv_stem_rows_kernel", PROF_MFMA, st, a.flops, true, a.hbm_bytes);
        hipExtLaunchKernelGGL(conv_stem_rows_kernel, dim3((a.mtiles + 7) / 8 * 8), dim3(256), SRF_W + 4 * SRF_SLAB, st, prof.e0(),
                              prof.e1(), 0, a, P, Hp, Wp);
        GDL_CHECK_LAUNCH("conv_stem_rows_kernel");
        return GDL_OK;
    }
    const TileCfg c = pick_cfg(a.M, 64, dtype);
    ConvPlan pl{};
    pl.slab = 0;
    pl.bm = c.bm;
    pl.bn = c.bn;
    if (dtype == GDL_BF16) return launch_mode<bf16, MODE_FWD>(a, pl, st);
    return launch_mode<float, MODE_FWD>(a, pl, st);
}

int conv_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const void* table, int N, int H,
               int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st, const uint8_t* relu_bits,
               const BwdStats* bw, const SplitWs* split) {
    return run_conv(GATHER_DGRAD, dtype, dy, w_crsk, dx, addend, nullptr, table, N, H, W, C, K, R, S, stride, pad, st,
                    relu_bits, nullptr, nullptr, nullptr, nullptr, bw, nullptr, nullptr, split);
}

// dx = dgrad(dy) * gelu'(u), elementwise in the epilogue (u laid out like dx), and the column sums of dx as stored added to
// the fixed-point accumulators acc (bnacc.h) -- Mlp.fc2's data gradient, the activation's backward and fc1's bias gradient
// (/root/reference/models/swin_transformer.py:32-47) in one launch
int conv_dgrad_gelu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* u, const BnAcc* acc, const void* table,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st) {
    GDL_REQUIRE(u, "conv_dgrad_gelu: null pointer");
    const ConvPlan pl = plan_conv(dtype, N * H * W, C, K, W, R, S, stride, pad);
    GDL_REQUIRE(!pl.c64, "conv_dgrad_gelu: not available on the 64-channel persistent kernel");
    return run_conv(GATHER_DGRAD, dtype, dy, w_crsk, dx, nullptr, nullptr, table, N, H, W, C, K, R, S, stride, pad, st,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, acc, u);
}

// forward with the epilogue's bias / residual: y = conv(x, w) + bias (+ addend), each optional (the Swin Linears)
int conv_fwd_bias(int dtype, const void* x, const void* w, void* y, const float* bias, const void* addend, void* gelu_out,
                  const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st) {
    return run_conv(GATHER_FWD, dtype, x, w, y, addend, nullptr, table, N, H, W, C, K, R, S, stride, pad, st, nullptr,
                    nullptr, nullptr, bias, gelu_out);
}

int conv_dgrad_ds(int dtype, const void* dy, const void* w_crsk, const void* dy_ds, const void* w_ds_ck, void* dx,
                  const void* table, int N, int H, int W, int C, int K, hipStream_t st, const uint8_t* relu_bits,
                  const BwdStats* bw) {
    GDL_REQUIRE(dy_ds && w_ds_ck, "conv_dgrad_ds: null pointer");
    return run_conv(GATHER_DGRAD, dtype, dy, w_crsk, dx, nullptr, nullptr, table, N, H, W, C, K, 3, 3, 2, 1, st,
                    relu_bits, dy_ds, w_ds_ck, nullptr, nullptr, bw);
}

}  // namespace gdl
