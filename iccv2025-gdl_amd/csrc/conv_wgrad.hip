// conv_wgrad.hip -- convolution weight-gradient on MFMA (gfx950).
//
// Replaces the weight-gradient of nn.Conv2d (/root/reference/models/backbone.py:20-28,
// 96-101) for NHWC activations:  dw[k][r][s][c] = sum_m dy[m][k] * x[gather(m,r,s)][c].
// GEMM view per tap: rows i = k (output channels), cols j = c (input channels),
// reduction over the M = N*P*Q output pixels.  Both operands are stored pixel-major
// ([pixel][channel]), i.e. the reduction index is the slow one, so the MFMA fragments
// (8 consecutive reduction elements per lane) are produced by the gfx950 LDS transpose
// read ds_read_b64_tr_b16 for bf16; the f32 MFMA takes one element per lane and needs
// no transpose.
//
// Block = 256 threads (2x2 waves), tile TK x TC output channels for ONE tap and ONE
// slice of the pixel range (split-K); LDS stages of 64 pixels, double buffered, filled by
// LDS-DMA (buffer_load ... lds) through the forward gather table.  Partial tiles go to a
// workspace [split][K][R*S][C] f32; a second kernel folds the splits in a fixed order and
// writes the reference's [K][C][R][S] gradient layout.  Tap is the fastest-varying index
// among blocks that share an XCD, so the taps of one pixel slice reuse dy / x from that
// XCD's L2.  This is the general kernel (stride-2 3x3, 1x1, the direct stem in `split` mode);
// the 3x3 stride-1 convolutions run on conv_wgrad9.hip, which handles all nine taps per block.
#include <stdlib.h>

#include "common.h"
#include "gather.h"
#include "prof.h"

#include <hip/hip_ext.h>

namespace gdl {

struct WgradArgs {
    const void* dy;            // [M][K]
    const void* x;             // gather source [N][H][W][C]
    float* partial;            // [nsplit][K][RS][C]
    const GatherEntry* table;  // forward gather table of this convolution, [M]
    int C, K, RS, M;
    int nsplit, chunk;  // pixels per split (multiple of 64)
    int tiles_k, tiles_c;
    unsigned dy_bytes, x_bytes;
    int delta[9];
    // direct stem (layout.hip): the TC channels of a tap are TWO filter rows of the padded input -- the first
    // half of a tile row comes from off0 + delta[tap], the second half from off0 + delta_hi[tap]
    int split;
    int delta_hi[9];
#ifdef GDL_TIMING
    unsigned long long* dbg;
#endif
};
#ifdef GDL_TIMING
extern unsigned long long* g_timing_buf;
#endif

constexpr int WG_BP = 64;  // pixels per LDS stage

// ds_read_b64_tr_b16: within each 16-lane group the 16 lanes' 8-byte reads form a 4x16 block of
// 16-bit elements (lane i supplies row i>>2, columns (i&3)*4..+3); lane i receives column i.
__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}

// 16 bytes per lane, global -> LDS without VGPR staging (see conv_igemm.hip::dma16)
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, 0, 0, 0);
#else
    (void)rsrc;
    (void)lds_base;
    (void)voffset;
#endif
}
// transpose read hidden from the compiler (it would wait vmcnt(0) before any LDS read it can see while
// an LDS-DMA is in flight); valid after wg_lds_wait()
__device__ __forceinline__ uint2 lds_tr16_asm(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ float lds_read_f32_asm(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void wg_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// 32-byte granule swizzle of a [pixel][channel] LDS tile (see DESIGN.md "wgrad LDS image")
template <int PITCH>
__device__ __forceinline__ int wg_swz(int row) {
    if (PITCH == 128) return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    return (row & 3) | (((row >> 3) & 1) << 2);
}

template <typename T, int TK, int TC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PK = TK * (int)sizeof(T), PC = TC * (int)sizeof(T);  // row pitches (bytes)
    constexpr int STAGE = WG_BP * (PK + PC);
    constexpr int CPR_K = PK / 16, CPR_C = PC / 16;                    // 16-byte chunks per row
    constexpr int LK = WG_BP * CPR_K / 256, LC = WG_BP * CPR_C / 256;  // loads per thread
    constexpr int WTK = TK / 2, WTC = TC / 2, FK = WTK / 16, FC = WTC / 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wc = wave >> 1;

    // block -> (slice, tap, ktile, ctile); blocks b, b+8, b+16.. share an XCD: keep the taps and
    // channel tiles of one pixel slice on it
    const int RS = a.RS;
    const int per_slice = RS * a.tiles_k * a.tiles_c;
    const int L = xcd_linear(a.nsplit * per_slice);
    if (L < 0) return;
    const int slice = L / per_slice;
    int rem = L - slice * per_slice;
    const int tap = rem % RS;
    rem /= RS;
    const int kt = rem % a.tiles_k, ct = rem / a.tiles_k;
    const int k0 = kt * TK, c0 = ct * TC;

    const int m_begin = slice * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int nst = (m_end - m_begin + WG_BP - 1) / WG_BP;

    // 32-bit byte offsets into buffer descriptors; invalid rows / taps read zeros (offset out of range)
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const int esz = (int)sizeof(T);
    const int tap_delta = a.delta[tap] + c0 * esz;
    // Tiles go global -> LDS by LDS-DMA: chunk idx = tid + 256*i lands at LDS byte idx*16 (lane-linear),
    // i.e. row idx/CPR, PHYSICAL chunk idx%CPR, so each lane fetches the source chunk whose swizzled
    // position that is (the 32-byte-granule XOR is an involution).
    int dy_off[LK];  // offset of this thread's chunk in the CURRENT stage
#pragma unroll
    for (int i = 0; i < LK; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / CPR_K, pc = idx % CPR_K;
        const int ch = (((pc >> 1) ^ wg_swz<PK>(row)) << 1) | (pc & 1);
        dy_off[i] = (m_begin + row) * (a.K * esz) + k0 * esz + ch * 16;
    }
    int x_ch[LC];
    bool x_hi[LC];
#pragma unroll
    for (int i = 0; i < LC; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / CPR_C, pc = idx % CPR_C;
        const int lc = (((pc >> 1) ^ wg_swz<PC>(row)) << 1) | (pc & 1);  // logical chunk of this lane's LDS slot
        x_hi[i] = a.split && lc >= CPR_C / 2;
        x_ch[i] = (x_hi[i] ? lc - CPR_C / 2 : lc) * 16;
    }
    const int tap_delta_hi = a.delta_hi[tap];
    GatherEntry ent[LC];  // table entries of the NEXT stage to load
#pragma unroll
    for (int i = 0; i < LC; ++i) {
        const int idx = tid + 256 * i;
        const int m = m_begin + idx / CPR_C;
        ent[i].off0 = 0;
        ent[i].mask = 0;
        if (m < m_end) ent[i] = a.table[m];
    }
    int ld_m = m_begin;  // first pixel of the next stage to load
    const int wbase = (tid >> 6) * 64 * 16;  // LDS byte offset of this wave's first lane within a 256-chunk group

    auto load_stage = [&](int buf) {
        unsigned char* Ks = smem + buf * STAGE;
        unsigned char* Cs = Ks + WG_BP * PK;
#pragma unroll
        for (int i = 0; i < LK; ++i) {
            const int idx = tid + 256 * i;
            const int m = ld_m + idx / CPR_K;
            wg_dma16(rdy, Ks + 256 * 16 * i + wbase, m < m_end ? dy_off[i] : (int)0x80000000);
            dy_off[i] += WG_BP * a.K * esz;
        }
#pragma unroll
        for (int i = 0; i < LC; ++i) {
            const int v = ((ent[i].mask >> tap) & 1u) ? ent[i].off0 + (x_hi[i] ? tap_delta_hi : tap_delta) + x_ch[i]
                                                      : (int)0x80000000;
            wg_dma16(rx, Cs + 256 * 16 * i + wbase, v);
        }
        ld_m += WG_BP;
#pragma unroll
        for (int i = 0; i < LC; ++i) {  // prefetch the table entries of the following stage
            const int idx = tid + 256 * i;
            const int m = ld_m + idx / CPR_C;
            GatherEntry e;
            e.off0 = 0;
            e.mask = 0;
            if (m < m_end) e = a.table[m];
            ent[i] = e;
        }
    };

    f32x4_t acc[FK][FC];
#pragma unroll
    for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int jj = 0; jj < FC; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;

    const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
#ifdef GDL_TIMING
    unsigned long long t_entry = __builtin_amdgcn_s_memtime(), t_wait = 0, t_comp = 0, t_issue = 0, t_a, t_b;
#endif
    if (nst > 0) load_stage(0);
#ifdef GDL_TIMING
    unsigned long long t_first = __builtin_amdgcn_s_memtime();
#endif
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
#ifdef GDL_TIMING
        t_a = __builtin_amdgcn_s_memtime();
#endif
        // the stage issued one iteration ago has landed in every wave's part; all waves are done
        // reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef GDL_TIMING
        t_b = __builtin_amdgcn_s_memtime();
        t_wait += t_b - t_a;
#endif
        if (st + 1 < nst) load_stage(buf ^ 1);
#ifdef GDL_TIMING
        t_a = __builtin_amdgcn_s_memtime();
        t_issue += t_a - t_b;
#endif
        const unsigned Ks = smem_base + buf * STAGE;
        const unsigned Cs = Ks + WG_BP * PK;
        if (sizeof(T) == 2) {
            // bf16: K-step = 32 pixels; lane group g owns pixels ks*32 + g*8 + {0..7}, fetched as two
            // transposed 4x16 blocks: lane li supplies row (li>>2), columns (li&3)*4.. and receives
            // column li of the block (verified on gfx950 by csrc/probe).
#pragma unroll
            for (int ks = 0; ks < WG_BP / 32; ++ks) {
                uint2 alo[FK], ahi[FK], blo[FC], bhi[FC];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = ks * 32 + g * 8 + h * 4 + (li >> 2);
#pragma unroll
                    for (int i = 0; i < FK; ++i) {
                        const int chb = (wk * WTK + i * 16 + (li & 3) * 4) * 2;  // byte offset in row
                        const int addr = row * PK + ((((chb >> 5) ^ wg_swz<PK>(row)) << 5) | (chb & 31));
                        if (h == 0)
                            alo[i] = lds_tr16_asm(Ks + addr);
                        else
                            ahi[i] = lds_tr16_asm(Ks + addr);
                    }
#pragma unroll
                    for (int jj = 0; jj < FC; ++jj) {
                        const int chb = (wc * WTC + jj * 16 + (li & 3) * 4) * 2;
                        const int addr = row * PC + ((((chb >> 5) ^ wg_swz<PC>(row)) << 5) | (chb & 31));
                        if (h == 0)
                            blo[jj] = lds_tr16_asm(Cs + addr);
                        else
                            bhi[jj] = lds_tr16_asm(Cs + addr);
                    }
                }
                wg_lds_wait();
#pragma unroll
                for (int i = 0; i < FK; ++i) {
                    const uint4 fa = make_uint4(alo[i].x, alo[i].y, ahi[i].x, ahi[i].y);
#pragma unroll
                    for (int jj = 0; jj < FC; ++jj) {
                        const uint4 fb = make_uint4(blo[jj].x, blo[jj].y, bhi[jj].x, bhi[jj].y);
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                           __builtin_bit_cast(bf16x8_t, fb), acc[i][jj], 0,
                                                                           0, 0);
                    }
                }
            }
        } else {
            // f32: 16x16x4 MFMA, A[i = li][k = g], B[k = g][j = li]: one float per lane, pixel = ks*4 + g
#pragma unroll 4
            for (int ks = 0; ks < WG_BP / 4; ++ks) {
                const int row = ks * 4 + g;
                float fa[FK], fb[FC];
#pragma unroll
                for (int i = 0; i < FK; ++i) {
                    const int chb = (wk * WTK + i * 16 + li) * 4;
                    fa[i] = lds_read_f32_asm(Ks + row * PK + ((((chb >> 5) ^ wg_swz<PK>(row)) << 5) | (chb & 31)));
                }
#pragma unroll
                for (int jj = 0; jj < FC; ++jj) {
                    const int chb = (wc * WTC + jj * 16 + li) * 4;
                    fb[jj] = lds_read_f32_asm(Cs + row * PC + ((((chb >> 5) ^ wg_swz<PC>(row)) << 5) | (chb & 31)));
                }
                wg_lds_wait();
#pragma unroll
                for (int i = 0; i < FK; ++i)
#pragma unroll
                    for (int jj = 0; jj < FC; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
            }
        }
    }
#ifdef GDL_TIMING
    // the loop body's compute time = everything not wait / issue
    unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
    t_comp = (t_loop_end - t_first) - t_wait - t_issue;
#endif
    // D[i][j]: i = k channel = g*4 + reg, j = c channel = li
    float* part = a.partial + (size_t)slice * a.K * RS * a.C;
#pragma unroll
    for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int jj = 0; jj < FC; ++jj) {
            const int c = c0 + wc * WTC + jj * 16 + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + wk * WTK + i * 16 + g * 4 + e;
                part[((size_t)k * RS + tap) * a.C + c] = acc[i][jj][e];
            }
        }
#ifdef GDL_TIMING
    if (a.dbg && tid == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 8;
        d[0] = t_entry;
        d[1] = t_first - t_entry;
        d[2] = t_wait;
        d[3] = t_issue;
        d[4] = t_comp;
        d[5] = __builtin_amdgcn_s_memtime() - t_loop_end;
        d[6] = nst;
    }
#endif
}

// out[k][c][r][s] = sum_split partial[split][k][rs][c], c < Cout (Cout <= C drops padding channels)
// Block = 256 threads = (256/LANES) consecutive outputs x LANES split-lanes; lane l sums splits
// l, l+LANES, ... and the lanes are folded through LDS in a fixed order (deterministic).
template <int LANES>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nsplit, int K, int RS, int C, int Cout) {
    constexpr int OUTS = 256 / LANES;
    __shared__ float red[LANES][OUTS];
    const size_t total = (size_t)K * RS * C;
    const int o = threadIdx.x % OUTS, l = threadIdx.x / OUTS;
    const size_t i = blockIdx.x * (size_t)OUTS + o;
    float s = 0.f;
    if (i < total)
        for (int sp = l; sp < nsplit; sp += LANES) s += partial[(size_t)sp * total + i];
    if (LANES > 1) {
        red[l][o] = s;
        __syncthreads();
        if (l == 0) {
            s = red[0][o];
#pragma unroll
            for (int j = 1; j < LANES; ++j) s += red[j][o];
        }
    }
    if (l == 0 && i < total) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int rs = (int)(t % RS);
        const int k = (int)(t / RS);
        if (c < Cout) out[((size_t)k * Cout + c) * RS + rs] = s;
    }
}

// conv_wgrad9.hip: all 9 taps per block for the 3x3 stride-1 convolutions (bf16)
bool conv_wgrad9_enabled();
bool conv_wgrad9_ok(int dtype, int W, int C, int K, int R, int S, int stride, int pad);
size_t conv_wgrad9_ws_bytes(int M, int C, int K);
int conv_wgrad9(const void* dy, const void* x, float* dw, const void* table, int N, int H, int W, int C, int K, void* ws,
                size_t ws_bytes, hipStream_t st);

// ---------------------------------------------------------------- host side
struct WgradPlan {
    int tk, tc, nsplit, chunk;
};

static WgradPlan plan_wgrad(int M, int C, int K, int RS, bool gemm = false) {
    WgradPlan p;
    p.tk = (K % 128 == 0) ? 128 : 64;
    p.tc = (C % 128 == 0) ? 128 : 64;
    const int tiles = (K / p.tk) * (C / p.tc) * RS;
    // Every block leaves a TKxTC fp32 partial tile, so partial traffic = blocks x 16..64 KB; at least 2 stages of work per
    // block.  Target block count: 128 inside the ResNet step, where these
    // weight gradients (stride-2 3x3, 1x1 downsample) run on the side stream beside the data-gradient chain (knob sweep of
    // round 2: 6.18 -> 6.14 ms against 384; round 5 with exact split counts: 96 / 128 / 192 / 256 within 0.3 %); about 400 -- one
    // resident wave of blocks -- for the plain GEMMs (1x1, stride 1: the Swin encoder's Linears), which have the device to
    // themselves (round 5, exact split counts, Swin step: 256 -> 20.71 ms, 320 -> 20.50, 352 -> 20.23, 384 -> 20.02, 416 -> 19.89,
    // 448 -> 20.00, 512 -> 20.39, 768 -> 20.32: profiles/r05_ab_wgrad_split.txt).  GDL_WGRAD_BLOCKS overrides both,
    // GDL_WGRAD_GEMM_BLOCKS the second (tuning aids).
    static int forced = -1;
    if (forced < 0) {
        const char* e = tune_env("GDL_WGRAD_BLOCKS");
        forced = e ? atoi(e) : 0;
    }
    static int forced_gemm = -1;
    if (forced_gemm < 0) {
        const char* e = tune_env("GDL_WGRAD_GEMM_BLOCKS");  // tuning aid: the plain GEMMs only
        forced_gemm = e ? atoi(e) : 0;
    }
    const int target = (gemm && forced_gemm > 0) ? forced_gemm : forced > 0 ? forced : (gemm ? 416 : 128);
    int ns = (target + tiles - 1) / tiles;
    const int max_ns = (M + 2 * WG_BP - 1) / (2 * WG_BP);
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    // (the split count was rounded up to a multiple of 8 while whole slices were dealt to the XCDs; with the linear mapping --
    // common.h xcd_linear -- the exact count is 0.5 % faster in the ResNet step and 1.3 % in the Swin composition, whose 36-tile
    // weight gradients took 16 slices for 11: profiles/r05_ab_wgrad_split.txt)
    int chunk = (M + ns - 1) / ns;
    chunk = (chunk + WG_BP - 1) / WG_BP * WG_BP;
    p.nsplit = (M + chunk - 1) / chunk;
    p.chunk = chunk;
    return p;
}

size_t conv_wgrad_ws_bytes(int M, int C, int K, int RS) {
    const WgradPlan p = plan_wgrad(M, C, K, RS, RS == 1);  // (the larger of the two split targets: an upper bound)
    size_t b = (size_t)p.nsplit * K * RS * C * sizeof(float);
    if (RS == 9 && conv_wgrad9_enabled()) {  // the 9-tap kernel's slices (geometry-independent upper bound)
        const size_t b9 = conv_wgrad9_ws_bytes(M, C, K);
        if (b9 > b) b = b9;
    }
    return b;
}

template <typename T, int TK, int TC>
static int launch_wg(WgradArgs& a, hipStream_t st) {
    constexpr int BYTES = 2 * WG_BP * (TK + TC) * (int)sizeof(T);
    auto kfn = conv_wgrad_kernel<T, TK, TC>;
    static DevOnce attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(conv_wgrad)");
        attr_set = true;
    }
    const int per_slice = a.RS * a.tiles_k * a.tiles_c;
    const int grid = xcd_grid(a.nsplit * per_slice);
    static char pname[96] = "";
    if (!pname[0]) snprintf(pname, sizeof(pname), "gdl::conv_wgrad_kernel<%s, %d, %d>", prof_tname<T>(), TK, TC);
    ProfScope prof(pname, PROF_MFMA, st, 2.0 * (double)a.M * a.K * a.C * a.RS, true,
                   (double)a.dy_bytes + (double)a.x_bytes + 4.0 * (double)a.K * a.C * a.RS);
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(256), BYTES, st, prof.e0(), prof.e1(), 0, a);
    GDL_CHECK_LAUNCH("conv_wgrad_kernel");
    return GDL_OK;
}

// Cout: number of leading input channels kept in dw (== C unless the channel axis is padded)
int conv_wgrad(int dtype, const void* dy, const void* x, float* dw, const void* table, int N, int H, int W, int C, int K,
               int R, int S, int stride, int pad, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    GDL_REQUIRE(dtype == GDL_BF16 || dtype == GDL_F32, "wgrad: bad dtype %d", dtype);
    GDL_REQUIRE(table, "wgrad: gather table is null (build it with gdl_conv_build_table)");
    GDL_REQUIRE(C % 64 == 0 && K % 64 == 0, "wgrad: C=%d K=%d must be multiples of 64", C, K);
    if (Cout == C && conv_wgrad9_ok(dtype, W, C, K, R, S, stride, pad))
        return conv_wgrad9(dy, x, dw, table, N, H, W, C, K, ws, ws_bytes, st);
    GatherGeom g;
    int rc = gather_geom(GATHER_FWD, dtype, N, H, W, C, K, R, S, stride, pad, &g);
    if (rc) return rc;
    const int esz = dtype == GDL_BF16 ? 2 : 4;
    WgradArgs a{};
#ifdef GDL_TIMING
    a.dbg = g_timing_buf;
#endif
    a.dy = dy;
    a.x = x;
    a.table = (const GatherEntry*)table;
    a.C = C;
    a.K = K;
    a.RS = R * S;
    a.M = g.rows;
    for (int t = 0; t < 9; ++t) a.delta[t] = g.delta[t];
    GDL_REQUIRE(a.M < (1 << 24), "wgrad: M exceeds 2^24");
    GDL_REQUIRE((size_t)a.M * K * esz < (1UL << 31), "wgrad: dy exceeds 2 GiB");
    a.dy_bytes = (unsigned)((size_t)a.M * K * esz);
    a.x_bytes = (unsigned)((size_t)N * H * W * C * esz);
    const WgradPlan p = plan_wgrad(a.M, C, K, R * S, R == 1 && S == 1 && stride == 1);
    a.nsplit = p.nsplit;
    a.chunk = p.chunk;
    a.tiles_k = K / p.tk;
    a.tiles_c = C / p.tc;
    const size_t need = (size_t)p.nsplit * K * R * S * C * sizeof(float);
    if (ws_bytes < need || !ws) {
        set_error("wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return GDL_ERR_WORKSPACE;
    }
    a.partial = (float*)ws;
    if (dtype == GDL_BF16) {
        if (p.tk == 128 && p.tc == 128)
            rc = launch_wg<bf16, 128, 128>(a, st);
        else if (p.tk == 128)
            rc = launch_wg<bf16, 128, 64>(a, st);
        else if (p.tc == 128)
            rc = launch_wg<bf16, 64, 128>(a, st);
        else
            rc = launch_wg<bf16, 64, 64>(a, st);
    } else {
        if (p.tk == 128 && p.tc == 128)
            rc = launch_wg<float, 128, 128>(a, st);
        else if (p.tk == 128)
            rc = launch_wg<float, 128, 64>(a, st);
        else if (p.tc == 128)
            rc = launch_wg<float, 64, 128>(a, st);
        else
            rc = launch_wg<float, 64, 64>(a, st);
    }
    if (rc) return rc;
    const size_t total = (size_t)K * R * S * C;
#ifdef GDL_EXPERIMENT
    if (experiment_mask2() & 1) return GDL_OK;
#endif
    ProfScope prof(p.nsplit >= 64 ? "gdl::wgrad_reduce_kernel<16>" : (p.nsplit > 8 ? "gdl::wgrad_reduce_kernel<4>" : "gdl::wgrad_reduce_kernel<1>"),
                   PROF_HBM, st, (double)total * 4.0 * (p.nsplit + 1));
    if (p.nsplit >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, a.partial, dw,
                           p.nsplit, K, R * S, C, Cout);
    else if (p.nsplit > 8)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, a.partial, dw,
                           p.nsplit, K, R * S, C, Cout);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.partial, dw,
                           p.nsplit, K, R * S, C, Cout);
    GDL_CHECK_LAUNCH("wgrad_reduce_kernel");
    return GDL_OK;
}

// ---- direct stem weight gradient: dw[k][c][r][s] from dy [M][64] and the padded NHWC4 input.
// GEMM per "tap" t = filter rows 2t, 2t+1: 64 k x 64 columns (2 rows x 8 pixels x 4 channels); the
// reduce kernel picks the 7x7xCin real entries out of the [64][4][64] tile.
// Block = 16 consecutive float4 of the [64][4][64] tile x 16 split-lanes (lane l adds slices l, l+16, ..., four loads in
// flight; a thread per OUTPUT looping over all slices serially took 67 us at 512 slices); the lanes are folded through LDS
// in a fixed order and the 7x7xCin real entries are scattered to [64][Cin][7][7].
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                int nsplit, int Cin) {
    __shared__ float4 red[16][16];
    constexpr int TILE4 = 64 * 4 * 64 / 4;
    const int o = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int idx4 = blockIdx.x * 16 + o;  // < TILE4 (grid = TILE4 / 16)
    const float4* p = (const float4*)partial + idx4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = l;
    for (; sp + 48 < nsplit; sp += 64) {
        const float4 v0 = p[(size_t)sp * TILE4], v1 = p[(size_t)(sp + 16) * TILE4];
        const float4 v2 = p[(size_t)(sp + 32) * TILE4], v3 = p[(size_t)(sp + 48) * TILE4];
        s.x += v0.x, s.y += v0.y, s.z += v0.z, s.w += v0.w;
        s.x += v1.x, s.y += v1.y, s.z += v1.z, s.w += v1.w;
        s.x += v2.x, s.y += v2.y, s.z += v2.z, s.w += v2.w;
        s.x += v3.x, s.y += v3.y, s.z += v3.z, s.w += v3.w;
    }
    for (; sp < nsplit; sp += 16) {
        const float4 v0 = p[(size_t)sp * TILE4];
        s.x += v0.x, s.y += v0.y, s.z += v0.z, s.w += v0.w;
    }
    red[l][o] = s;
    __syncthreads();
    if (l != 0) return;
#pragma unroll
    for (int jj = 1; jj < 16; ++jj) {
        const float4 v = red[jj][o];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    // float4 idx4 -> (k, tap, c = 4q .. 4q+3) with q = (r & 1) * 8 + s_px: channel = component
    const int q = idx4 & 15, tap = (idx4 >> 4) & 3, k = idx4 >> 6;
    const int r = 2 * tap + (q >> 3), px = q & 7;
    if (r < 7 && px < 7) {
        const float v[4] = {s.x, s.y, s.z, s.w};
        for (int c = 0; c < Cin; ++c) out[((k * Cin + c) * 7 + r) * 7 + px] = v[c];
    }
}

// ---- bf16 stem weight gradient, row-slab form.  The per-tap kernel above fetches, per 64 output pixels and tap, a
// dy tile and a GATHERED x tile (64 x 128 B) -- 64 LDS-DMA pieces per 32 MFMAs, and it ran at the end of every encoder
// backward with nothing beside it (switching it off: -0.40 ms of a 6.7 ms step).  Here a stage is 64 consecutive output
// pixels of ONE output row: their dy rows are one contiguous 8 KiB run and the 8 padded-input rows they touch are 8
// contiguous runs of 134 pixels (1 072 B), stored in LDS at a pitch of 1 088 B -- 17 pieces per stage for the same 32
// MFMAs, no masks (the padding is materialised in xp), every LDS read address static per lane:
//   B fragment of tap t (filter rows 2t, 2t+1), wave w (columns 16w..16w+15 = row 2t+(w>>1), filter pixels 4(w&1)..+3):
//   output pixel m reads input pixels 2m + 4(w&1) + {0..3} of slab row 2t+(w>>1): 8 bytes per lane, transpose read.
// Rows wider than 64 pixels take ceil(Q/64) stages; the last one starts at Q-64 and the pixels it shares with its
// predecessor are loaded as zeros (dy out of range).  Partials: [slice][k][tap][c], folded by stem_wgrad_reduce_kernel.
struct StemRowsArgs {
    const void* dy;      // [n_img*P*Q][64] bf16
    const void* xp;      // [n_img][Hp][Wp][4] bf16
    float* partial;      // [nsplit][64][4][64]
    int P, Q, Hp, Wp, nseg;
    int stages, chunk, nsplit;  // stages = n_img*P*nseg, chunk = stages per slice
    unsigned dy_bytes, x_bytes;
};
constexpr int SR_PITCH = 1088, SR_XBYTES = 9 * 1024, SR_STAGE = 8192 + SR_XBYTES;
template <int OFF>
__device__ __forceinline__ uint2 sr_tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void sr_wait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__global__ __launch_bounds__(256) void stem_wgrad_rows_kernel(StemRowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int slice = blockIdx.x;
    const int s_begin = slice * a.chunk, s_end = min(a.stages, s_begin + a.chunk);
    const int nst = s_end - s_begin;
    const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.xp, 0, a.x_bytes, 0x00020000);

    // ---- DMA bookkeeping: dy pieces p = wave, wave+4 (8 rows x 128 B each, 32-byte-granule swizzle as in the kernels
    // above); x pieces p = wave, wave+4, wave+8 (< 9): lane L of piece p lands at linear byte 1024p + 16L of the
    // [8][1088] slab, i.e. slab row (1024p+16L)/1088, byte (1024p+16L)%1088 of that row's 1072-byte run.
    const int prow = lane >> 3, pch = lane & 7;
    int dy_rel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 4 * i) * 8 + prow;
        const int ch = (((pch >> 1) ^ wg_swz<128>(row)) << 1) | (pch & 1);
        dy_rel[i] = row * 128 + ch * 16;
    }
    int x_rel[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = wave + 4 * i, off = 1024 * p + 16 * lane;
        const int row = off / SR_PITCH, col = off % SR_PITCH;
        x_rel[i] = (p < 9 && row < 8 && col < 1072) ? row * a.Wp * 8 + col : -1;
    }
    int ld_s = s_begin;
    auto load_stage = [&](int buf) {
        unsigned char* Ks = smem + buf * SR_STAGE;
        unsigned char* Xs = Ks + 8192;
        const int sg = ld_s % a.nseg, R = ld_s / a.nseg;  // R = n*P + oh
        const int n = R / a.P, oh = R - n * a.P;
        const int ow0 = min(64 * sg, a.Q - 64), first = 64 * sg - ow0;  // tile rows < first belong to the previous stage
        const int dy_base = (R * a.Q + ow0) * 128;
        const int x_base = ((n * a.Hp + 2 * oh) * a.Wp + 2 * ow0) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave + 4 * i) * 8 + prow;
            wg_dma16(rdy, Ks + (wave + 4 * i) * 1024, row >= first ? dy_base + dy_rel[i] : (int)0x80000000);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (wave + 4 * i < 9) wg_dma16(rx, Xs + (wave + 4 * i) * 1024, x_rel[i] >= 0 ? x_base + x_rel[i] : (int)0x80000000);
        ld_s += 1;
    };

    // ---- per-lane LDS read addresses (stage buffer 0)
    const int lrow = g * 8 + (li >> 2);  // + h*4 + ks*32
    unsigned aaddr[4][2], baddr[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lrow + h * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) aaddr[i][h] = smem_base + row * 128 + ((i ^ wg_swz<128>(row)) << 5) + (li & 3) * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            baddr[t][h] = smem_base + 8192 + (2 * t + (wave >> 1)) * SR_PITCH + (2 * row + 4 * (wave & 1) + (li & 3)) * 8;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nst > 0) load_stage(0);
    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < nst) load_stage((st + 1) & 1);
        // two K-steps of 32 pixels: dy rows +32 (4096 B), x pixels +64 (512 B): instruction offsets
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint2 af[4][2], bf[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) af[i][h] = ks ? sr_tr<4096>(aaddr[i][h]) : sr_tr<0>(aaddr[i][h]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) bf[t][h] = ks ? sr_tr<512>(baddr[t][h]) : sr_tr<0>(baddr[t][h]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t == 0)
                    sr_wait<6>();
                else if (t == 1)
                    sr_wait<4>();
                else if (t == 2)
                    sr_wait<2>();
                else
                    sr_wait<0>();
                const uint4 fb = make_uint4(bf[t][0].x, bf[t][0].y, bf[t][1].x, bf[t][1].y);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 fa = make_uint4(af[i][0].x, af[i][0].y, af[i][1].x, af[i][1].y);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                       __builtin_bit_cast(bf16x8_t, fb), acc[t][i], 0, 0, 0);
                }
            }
        }
        const int dlt = (st & 1) ? -SR_STAGE : SR_STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) aaddr[i][h] += dlt;
#pragma unroll
            for (int t = 0; t < 4; ++t) baddr[t][h] += dlt;
        }
    }
    // D[i][j]: k = 16i + (lane>>4)*4 + e, c = 16*wave + (lane&15);  partial[slice][k][tap][c]
    float* part = a.partial + (size_t)slice * (64 * 4 * 64);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((16 * i + g * 4 + e) * 4 + t) * 64 + 16 * wave + li] = acc[t][i][e];
}
// ---- fused stem backward (round 5): max-pool gather + ReLU mask + BatchNorm-backward apply + the stem's weight gradient in ONE
// launch.  The gradient of the stem output, dy0 -- as large as the stem output itself, the largest activation of the network:
// 308 MB written by maxpool_bn_bwd_apply_kernel and read back by stem_wgrad_rows_kernel for the visual batch -- is never
// stored: a stage's [64 pixels][64 channels] dy tile is COMPUTED into the LDS image the LDS-DMA used to deposit (same rows, same
// 32-byte-granule swizzle), from the stem output y0, the pooled gradient dz and the arg-max codes of the (at most four) pooling
// windows that contain each pixel.  The arithmetic of a tile element is maxpool_bn_bwd_apply_kernel's, in the same order
// (windows (p, q), (p, q+1), (p+1, q), (p+1, q+1) of the pixel's 2 x 2 patch; A g' + (B y + D)), the stages, the x slab, the
// MFMA loop and the partial layout are stem_wgrad_rows_kernel's: the weight gradient is bit-identical to the two-launch form.
// Thread t owns the logical 16-byte chunk t & 7 (channels 8 (t & 7) ..) of tile rows t >> 3 and 32 + (t >> 3): its forty
// per-channel constants live in registers for the whole launch.  Per stage: the operands of the NEXT stage's two elements are
// requested behind this stage's tile writes and land under the MFMAs.
struct StemFusedArgs {
    const void* y0;      // [n_img*P*Q][64] bf16: the stem convolution's raw output
    const void* dz;      // [n_img*PP*QQ][64] bf16: gradient of the pooled map
    const uint8_t* idx;  // [n_img*PP*QQ][64]: arg-max code of every pooling window
    const float *scale, *shift, *mean, *rstd, *gamma, *coef;
    const void* xp;
    float* partial;
    int P, Q, PP, QQ, Hp, Wp, nseg;
    int stages, chunk, nsplit;
    unsigned x_bytes, y_bytes, dz_bytes;  // (buffer descriptors: the tensors are below 2 GiB)
#ifdef GDL_TIMING
    unsigned long long* dbg;  // [block][wave][8]: entry, prologue, wait (memory + barrier), emit + issue, multiply, epilogue, stages
#endif
};
typedef unsigned int sr_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sr_write16(unsigned addr, const uint4& v) {
    const sr_u32x4_t w = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(w) : "memory");
}
__global__ __launch_bounds__(256, 2) void stem_bwd_fused_kernel(StemFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int slice = blockIdx.x;
    const int s_begin = slice * a.chunk, s_end = min(a.stages, s_begin + a.chunk);
    const int nst = s_end - s_begin;
    const unsigned smem_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.xp, 0, a.x_bytes, 0x00020000);

    // ---- per-channel constants of this thread's chunk (maxpool_bn_bwd_apply_kernel's three-constant form)
    const int vc = tid & 7;
    float sc[8], sf[8], A[8], Bc[8], D[8];
    {
        float mu[8], rs[8], gr[8], k1[8], k2[8];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            auto ld = [&](const float* p, float* v) {
                const float4 t = *(const float4*)(p + vc * 8 + 4 * q4);
                v[4 * q4 + 0] = t.x, v[4 * q4 + 1] = t.y, v[4 * q4 + 2] = t.z, v[4 * q4 + 3] = t.w;
            };
            ld(a.scale, sc), ld(a.shift, sf), ld(a.mean, mu), ld(a.rstd, rs), ld(a.gamma, gr), ld(a.coef, k1), ld(a.coef + 64, k2);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            A[e] = gr[e] * rs[e];
            Bc[e] = -A[e] * rs[e] * k2[e];
            D[e] = -A[e] * k1[e] - Bc[e] * mu[e];
        }
    }
    // This thread's two tile elements are the pixels 2j and 2j + 1 of the stage (j = t >> 3), taken so that element 0 is the one
    // in an EVEN image column and element 1 the one in an ODD column (par = the parity of the stage's first column: odd only in
    // the last segment of an odd-width row): an even column lies in the windows (p, q) [and (p + 1, q) in odd rows], an odd one in
    // (p, q), (p, q + 1) [and (p + 1, q), (p + 1, q + 1)] -- the window sets are static per element, the row parity is uniform.
    // LDS byte offset inside a stage's dy tile: logical chunk vc of row r sits at physical chunk ((vc >> 1) ^ wg_swz<128>(r)) * 2
    // + (vc & 1), the image the LDS-DMA of stem_wgrad_rows_kernel leaves.
    const int j2 = (tid >> 3) * 2;
    unsigned eoffs[2];  // [tile row parity]
#pragma unroll
    for (int i = 0; i < 2; ++i) eoffs[i] = (j2 + i) * 128 + ((((vc >> 1) ^ wg_swz<128>(j2 + i)) << 5) | ((vc & 1) << 4));

    // ---- x slab DMA bookkeeping (as stem_wgrad_rows_kernel)
    int x_rel[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = wave + 4 * i, off = 1024 * p + 16 * lane;
        const int row = off / SR_PITCH, col = off % SR_PITCH;
        x_rel[i] = (p < 9 && row < 8 && col < 1072) ? row * a.Wp * 8 + col : -1;
    }
    // stage coordinates, advanced incrementally (wave-uniform): segment, output row, image
    struct Coord {
        int sg, oh, n;
    };
    auto advance = [&](Coord& c) {
        if (++c.sg == a.nseg) {
            c.sg = 0;
            if (++c.oh == a.P) {
                c.oh = 0;
                ++c.n;
            }
        }
    };
    Coord cq;  // of the stage whose operands were requested last (= the next one to emit)
    {
        const int R = s_begin / a.nseg;
        cq.sg = s_begin - R * a.nseg;
        cq.n = R / a.P;
        cq.oh = R - cq.n * a.P;
    }
    // operands in flight from one stage to the next: the stem output of both pixels, gradient + codes of their windows
    uint4 y_e, y_o, d_e[2], d_o[4];
    uint2 i_e[2], i_o[4];
    bool live_e = false, live_o = false;
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    // Operand loads through buffer descriptors with 32-bit offsets (round 5, second half: as flat loads under exec-mask branches
    // they cost ~60 VALU of 64-bit address arithmetic, ~40 moves that preset the operands of windows that do not exist and eight
    // branches per stage in a VALU-bound kernel): a window that does not exist -- or a pixel outside the stage -- is an offset
    // outside the descriptor, which reads as zeros: a zero gradient adds +0 whatever its (zero) code matches.  Static VALU count
    // -18 %, 243 -> 231 us alone at the visual shape, nothing in the step (tools/bench_stem_bwd.py --cycles: a wave's stage is 680 clk of
    // waiting, 2 850 of emit + issue -- two waves per SIMD taking turns on the VALU -- and 625 of multiplication).
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y0, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc((void*)a.idx, 0, a.dz_bytes >> 1, 0x00020000);
    auto bl16 = [](__amdgpu_buffer_rsrc_t r, int off) __attribute__((always_inline)) {
        const sr_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    typedef unsigned int sr_u32x2_t __attribute__((ext_vector_type(2)));
    auto bl8 = [](__amdgpu_buffer_rsrc_t r, int off) __attribute__((always_inline)) {
        const sr_u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        return make_uint2(v.x, v.y);
    };
    constexpr int OOB = (int)0x80000000;
    auto request = [&](const Coord& c) {  // loads of the two elements of the stage at c
        const int ow0 = min(64 * c.sg, a.Q - 64), first = 64 * c.sg - ow0;
        const int par = ow0 & 1;
        const int r_e = j2 + par, r_o = j2 + (par ^ 1);  // tile rows of the even- / odd-column pixel
        live_e = r_e >= first, live_o = r_o >= first;
        const int p = c.oh >> 1, hb = c.oh & 1;
        const int rowbase = (c.n * a.P + c.oh) * a.Q + ow0;  // (pixels; the tensor is below 2 GiB: host check)
        const int q_e = (ow0 + r_e) >> 1, q_o = (ow0 + r_o) >> 1;
        const int w0 = (c.n * a.PP + p) * a.QQ;                // pooled row p
        const bool row1 = hb && p + 1 < a.PP;                  // (uniform) the pixels also lie in pooled row p + 1
        const int w1 = row1 ? w0 + a.QQ : -1;
        const bool o1 = live_o && q_o + 1 < a.QQ;
        y_e = bl16(ry, live_e ? (rowbase + r_e) * 128 + vc * 16 : OOB);
        y_o = bl16(ry, live_o ? (rowbase + r_o) * 128 + vc * 16 : OOB);
        const int we = live_e ? (w0 + q_e) * 64 + vc * 8 : OOB, wo = live_o ? (w0 + q_o) * 64 + vc * 8 : OOB;
        const int wo1 = o1 ? (w0 + q_o + 1) * 64 + vc * 8 : OOB;
        // (an idx offset is half the dz offset; OOB * 2 wraps to 0: the dz offsets are selected on their own)
        d_e[0] = bl16(rdz, live_e ? (w0 + q_e) * 128 + vc * 16 : OOB), i_e[0] = bl8(ridx, we);
        d_o[0] = bl16(rdz, live_o ? (w0 + q_o) * 128 + vc * 16 : OOB), i_o[0] = bl8(ridx, wo);
        d_o[1] = bl16(rdz, o1 ? (w0 + q_o + 1) * 128 + vc * 16 : OOB), i_o[1] = bl8(ridx, wo1);
        const bool e2 = live_e && row1, o2 = live_o && row1, o3 = o1 && row1;
        d_e[1] = bl16(rdz, e2 ? (w1 + q_e) * 128 + vc * 16 : OOB), i_e[1] = bl8(ridx, e2 ? (w1 + q_e) * 64 + vc * 8 : OOB);
        d_o[2] = bl16(rdz, o2 ? (w1 + q_o) * 128 + vc * 16 : OOB), i_o[2] = bl8(ridx, o2 ? (w1 + q_o) * 64 + vc * 8 : OOB);
        d_o[3] = bl16(rdz, o3 ? (w1 + q_o + 1) * 128 + vc * 16 : OOB), i_o[3] = bl8(ridx, o3 ? (w1 + q_o + 1) * 64 + vc * 8 : OOB);
    };
    // one window's contribution: gsum[c] += dz[c] where the window's arg-max code names this pixel
    auto window = [&](float (&gsum)[8], const uint4& dq, const uint2& u, uint32_t code) __attribute__((always_inline)) {
        float d[8];
        unpack16<bf16>(dq, d);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (((u.x >> (8 * c)) & 0xff) == code) gsum[c] += d[c];
            if (((u.y >> (8 * c)) & 0xff) == code) gsum[4 + c] += d[4 + c];
        }
    };
    auto finish = [&](float (&gsum)[8], const uint4& yq, bool live, unsigned addr) __attribute__((always_inline)) {
        float yv[8];
        unpack16<bf16>(yq, yv);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float gg = (yv[c] * sc[c] + sf[c] > 0.f) ? gsum[c] : 0.f;
            gsum[c] = A[c] * gg + (Bc[c] * yv[c] + D[c]);
        }
        sr_write16(addr, live ? pack16<bf16>(gsum) : z4);
    };
    auto emit = [&](const Coord& c, int buf) {  // the dy tile elements of the stage at c (operands: the last request) -> LDS
        const int ow0 = min(64 * c.sg, a.Q - 64);
        const int par = ow0 & 1;
        const uint32_t hb = (uint32_t)(c.oh & 1);
        const unsigned base = smem_base + buf * SR_STAGE;
        // codes of a pixel inside the windows of its 2 x 2 patch, in maxpool_bn_bwd_apply_kernel's order: (p, q), (p, q + 1),
        // (p + 1, q), (p + 1, q + 1); window p covers the rows 2p - 1 .. 2p + 1
        float ge[8], go[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ge[k] = go[k] = 0.f;
        window(ge, d_e[0], i_e[0], (1 + hb) * 3 + 1);
        window(go, d_o[0], i_o[0], (1 + hb) * 3 + 2);
        window(go, d_o[1], i_o[1], (1 + hb) * 3);
        if (hb) {
            window(ge, d_e[1], i_e[1], 1);
            window(go, d_o[2], i_o[2], 2);
            window(go, d_o[3], i_o[3], 0);
        }
        finish(ge, y_e, live_e, base + eoffs[par]);
        finish(go, y_o, live_o, base + eoffs[par ^ 1]);
    };
    auto load_x = [&](const Coord& c, int buf) {
        unsigned char* Xs = smem + buf * SR_STAGE + 8192;
        const int ow0 = min(64 * c.sg, a.Q - 64);
        const int x_base = ((c.n * a.Hp + 2 * c.oh) * a.Wp + 2 * ow0) * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (wave + 4 * i < 9) wg_dma16(rx, Xs + (wave + 4 * i) * 1024, x_rel[i] >= 0 ? x_base + x_rel[i] : (int)0x80000000);
    };

    // ---- per-lane LDS read addresses (stage buffer 0), as stem_wgrad_rows_kernel
    const int lrow = g * 8 + (li >> 2);
    unsigned aaddr[4][2], baddr[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lrow + h * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) aaddr[i][h] = smem_base + row * 128 + ((i ^ wg_swz<128>(row)) << 5) + (li & 3) * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            baddr[t][h] = smem_base + 8192 + (2 * t + (wave >> 1)) * SR_PITCH + (2 * row + 4 * (wave & 1) + (li & 3)) * 8;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#ifdef GDL_TIMING
    unsigned long long t_entry = __builtin_amdgcn_s_memtime(), t_wait = 0, t_emit = 0, t_mul = 0, t_a, t_b;
#endif
    if (nst > 0) {
        request(cq);
        load_x(cq, 0);
        emit(cq, 0);  // (the compiler waits for the requested operands here)
        if (nst > 1) {
            advance(cq);
            request(cq);
        }
    }
#ifdef GDL_TIMING
    const unsigned long long t_first = __builtin_amdgcn_s_memtime();
#endif
    for (int st = 0; st < nst; ++st) {
#ifdef GDL_TIMING
        t_a = __builtin_amdgcn_s_memtime();
#endif
        // the x slab of stage st (and the operands of stage st + 1) have landed; every wave's tile writes of stage st are visible
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef GDL_TIMING
        t_b = __builtin_amdgcn_s_memtime();
        t_wait += t_b - t_a;
#endif
        if (st + 1 < nst) {
            emit(cq, (st + 1) & 1);
            load_x(cq, (st + 1) & 1);
            if (st + 2 < nst) {
                advance(cq);
                request(cq);
            }
        }
#ifdef GDL_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the tile writes: so that the split below is the arithmetic's)
        t_a = __builtin_amdgcn_s_memtime();
        t_emit += t_a - t_b;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint2 af[4][2], bf[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) af[i][h] = ks ? sr_tr<4096>(aaddr[i][h]) : sr_tr<0>(aaddr[i][h]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) bf[t][h] = ks ? sr_tr<512>(baddr[t][h]) : sr_tr<0>(baddr[t][h]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t == 0)
                    sr_wait<6>();
                else if (t == 1)
                    sr_wait<4>();
                else if (t == 2)
                    sr_wait<2>();
                else
                    sr_wait<0>();
                const uint4 fb = make_uint4(bf[t][0].x, bf[t][0].y, bf[t][1].x, bf[t][1].y);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 fa = make_uint4(af[i][0].x, af[i][0].y, af[i][1].x, af[i][1].y);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                       __builtin_bit_cast(bf16x8_t, fb), acc[t][i], 0, 0, 0);
                }
            }
        }
#ifdef GDL_TIMING
        t_mul += __builtin_amdgcn_s_memtime() - t_a;
#endif
        const int dlt = (st & 1) ? -SR_STAGE : SR_STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) aaddr[i][h] += dlt;
#pragma unroll
            for (int t = 0; t < 4; ++t) baddr[t][h] += dlt;
        }
    }
#ifdef GDL_TIMING
    if (a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        d[0] = t_entry, d[1] = t_first - t_entry, d[2] = t_wait, d[3] = t_emit, d[4] = t_mul;
        d[5] = __builtin_amdgcn_s_memtime() - t_first, d[6] = nst, d[7] = 0;
    }
#endif
    float* part = a.partial + (size_t)slice * (64 * 4 * 64);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((16 * i + g * 4 + e) * 4 + t) * 64 + 16 * wave + li] = acc[t][i][e];
}
struct StemRowsPlan {
    int nseg, stages, chunk, nsplit;
};
static bool stem_rows_ok(int dtype, int W) {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_STEM_ROWS");  // tuning aid: 0 = per-tap kernel
        v = e ? atoi(e) : 1;
    }
    return v != 0 && dtype == GDL_BF16 && (W - 1) / 2 + 1 >= 64;
}
static StemRowsPlan plan_stem_rows(int n_img, int H, int W) {
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1;
    StemRowsPlan p;
    p.nseg = (Q + 63) / 64;
    p.stages = n_img * P * p.nseg;
    static int target = -1;
    if (target < 0) {
        const char* e = tune_env("GDL_STEM_ROWS_BLOCKS");  // tuning aid
        target = e ? atoi(e) : 512;
    }
    int ns = target;
    if (ns > (p.stages + 3) / 4) ns = (p.stages + 3) / 4;  // at least 4 stages per 64 KB partial tile
    if (ns < 1) ns = 1;
    p.chunk = (p.stages + ns - 1) / ns;
    p.nsplit = (p.stages + p.chunk - 1) / p.chunk;
    return p;
}

static WgradPlan plan_stem_wgrad(int M) { return plan_wgrad(M, 64, 64, 4); }
size_t conv_stem_wgrad_ws_bytes(int n_img, int H, int W) {
    const int M = n_img * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    int ns = plan_stem_wgrad(M).nsplit;
    if ((W - 1) / 2 + 1 >= 64) ns = max(ns, plan_stem_rows(n_img, H, W).nsplit);  // (either kernel may run: dtype decides)
    return (size_t)ns * 64 * 4 * 64 * sizeof(float);
}
int conv_stem_wgrad(int dtype, const void* dy, const void* xp, float* dw, const void* table, int n_img, int H, int W, int Cin,
                    void* ws, size_t ws_bytes, hipStream_t st) {
    GDL_REQUIRE(dtype == GDL_BF16 || dtype == GDL_F32, "stem wgrad: bad dtype %d", dtype);
    GDL_REQUIRE(dy && xp && dw && table, "stem wgrad: null pointer");
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1, Hp = H + 6, Wp = W + 8;
    const int esz = dtype == GDL_BF16 ? 2 : 4, pix = 4 * esz;
    if (stem_rows_ok(dtype, W)) {
        const StemRowsPlan p = plan_stem_rows(n_img, H, W);
        const size_t need = (size_t)p.nsplit * 64 * 4 * 64 * sizeof(float);
        if (ws_bytes < need || !ws) {
            set_error("stem wgrad: workspace %zu < %zu bytes", ws_bytes, need);
            return GDL_ERR_WORKSPACE;
        }
        GDL_REQUIRE((size_t)n_img * P * Q * 128 < (1UL << 31) && (size_t)n_img * Hp * Wp * 8 < (1UL << 31), "stem wgrad: tensor exceeds 2 GiB");
        StemRowsArgs r{};
        r.dy = dy;
        r.xp = xp;
        r.partial = (float*)ws;
        r.P = P, r.Q = Q, r.Hp = Hp, r.Wp = Wp, r.nseg = p.nseg;
        r.stages = p.stages, r.chunk = p.chunk, r.nsplit = p.nsplit;
        r.dy_bytes = (unsigned)((size_t)n_img * P * Q * 128);
        r.x_bytes = (unsigned)((size_t)n_img * Hp * Wp * 8);
        {
            ProfScope prof("gdl::stem_wgrad_rows_kernel", PROF_MFMA, st, 2.0 * (double)n_img * P * Q * 64 * 49 * Cin, true);
            hipExtLaunchKernelGGL(stem_wgrad_rows_kernel, dim3(p.nsplit), dim3(256), 2 * SR_STAGE, st, prof.e0(), prof.e1(), 0, r);
            GDL_CHECK_LAUNCH("stem_wgrad_rows_kernel");
        }
        ProfScope prof("gdl::stem_wgrad_reduce_kernel", PROF_HBM, st, (double)64 * Cin * 49 * 4.0 * (p.nsplit + 1));
        hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * 4 * 64 / 4 / 16), dim3(256), 0, st, r.partial, dw, p.nsplit,
                           Cin);
        GDL_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
        return GDL_OK;
    }
    WgradArgs a{};
#ifdef GDL_TIMING
    a.dbg = nullptr;
#endif
    a.dy = dy;
    a.x = xp;
    a.table = (const GatherEntry*)table;
    a.C = 64;
    a.K = 64;
    a.RS = 4;
    a.M = n_img * P * Q;
    a.split = 1;
    for (int t = 0; t < 4; ++t) {
        a.delta[t] = (2 * t) * Wp * pix;
        a.delta_hi[t] = 2 * t + 1 < 7 ? (2 * t + 1) * Wp * pix : 0x40000000;  // no 8th filter row: out of range -> zeros
    }
    GDL_REQUIRE(a.M < (1 << 24), "stem wgrad: M exceeds 2^24");
    GDL_REQUIRE((size_t)a.M * 64 * esz < (1UL << 31), "stem wgrad: dy exceeds 2 GiB");
    a.dy_bytes = (unsigned)((size_t)a.M * 64 * esz);
    a.x_bytes = (unsigned)((size_t)n_img * Hp * Wp * pix);
    const WgradPlan p = plan_stem_wgrad(a.M);
    a.nsplit = p.nsplit;
    a.chunk = p.chunk;
    a.tiles_k = 1;
    a.tiles_c = 1;
    const size_t need = (size_t)p.nsplit * 64 * 4 * 64 * sizeof(float);
    if (ws_bytes < need || !ws) {
        set_error("stem wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return GDL_ERR_WORKSPACE;
    }
    a.partial = (float*)ws;
    int rc = dtype == GDL_BF16 ? launch_wg<bf16, 64, 64>(a, st) : launch_wg<float, 64, 64>(a, st);
    if (rc) return rc;
    ProfScope prof("gdl::stem_wgrad_reduce_kernel", PROF_HBM, st, (double)64 * Cin * 49 * 4.0 * (p.nsplit + 1));
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * 4 * 64 / 4 / 16), dim3(256), 0, st, a.partial, dw, p.nsplit,
                       Cin);
    GDL_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
    return GDL_OK;
}

// the fused stem backward (stem_bwd_fused_kernel): bf16, output rows of at least 64 pixels -- where conv_stem_wgrad runs the
// row-slab kernel
bool stem_bwd_fused_ok(int dtype, int W) {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_STEM_FUSED");  // tuning aid: 0 = maxpool_bn_bwd_apply + conv_stem_wgrad (two launches, dy0 stored)
        v = e ? atoi(e) : 1;
    }
    return v != 0 && stem_rows_ok(dtype, W);
}
int stem_bwd_fused(const void* dz, const uint8_t* idx, const void* y0, const float* scale, const float* shift, const float* mean,
                   const float* rstd, const float* gamma, const float* coef, const void* xp, float* dw, int n_img, int H, int W,
                   int Cin, void* ws, size_t ws_bytes, hipStream_t st) {
    GDL_REQUIRE(dz && idx && y0 && scale && shift && mean && rstd && gamma && coef && xp && dw, "stem_bwd_fused: null pointer");
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1, Hp = H + 6, Wp = W + 8;
    GDL_REQUIRE(Q >= 64, "stem_bwd_fused: output rows of %d pixels (at least 64)", Q);
    const StemRowsPlan p = plan_stem_rows(n_img, H, W);
    const size_t need = (size_t)p.nsplit * 64 * 4 * 64 * sizeof(float);
    if (ws_bytes < need || !ws) {
        set_error("stem_bwd_fused: workspace %zu < %zu bytes", ws_bytes, need);
        return GDL_ERR_WORKSPACE;
    }
    GDL_REQUIRE((size_t)n_img * P * Q * 128 < (1UL << 31) && (size_t)n_img * Hp * Wp * 8 < (1UL << 31), "stem_bwd_fused: tensor exceeds 2 GiB");
    StemFusedArgs r{};
    r.y0 = y0, r.dz = dz, r.idx = idx;
    r.scale = scale, r.shift = shift, r.mean = mean, r.rstd = rstd, r.gamma = gamma, r.coef = coef;
    r.xp = xp;
    r.partial = (float*)ws;
    r.P = P, r.Q = Q, r.PP = (P - 1) / 2 + 1, r.QQ = (Q - 1) / 2 + 1, r.Hp = Hp, r.Wp = Wp, r.nseg = p.nseg;
    r.stages = p.stages, r.chunk = p.chunk, r.nsplit = p.nsplit;
    r.x_bytes = (unsigned)((size_t)n_img * Hp * Wp * 8);
    r.y_bytes = (unsigned)((size_t)n_img * P * Q * 128);
#ifdef GDL_TIMING
    r.dbg = g_timing_buf;
#endif
    r.dz_bytes = (unsigned)((size_t)n_img * r.PP * r.QQ * 128);
    {
        // (priced as the weight gradient; its algorithmic bytes: y0 + the pooled gradient and codes + the padded input, once)
        ProfScope prof("gdl::stem_bwd_fused_kernel", PROF_MFMA, st, 2.0 * (double)n_img * P * Q * 64 * 49 * Cin, true,
                       (double)n_img * P * Q * 128.0 + (double)n_img * r.PP * r.QQ * 192.0 + (double)n_img * Hp * Wp * 8.0);
        hipExtLaunchKernelGGL(stem_bwd_fused_kernel, dim3(p.nsplit), dim3(256), 2 * SR_STAGE, st, prof.e0(), prof.e1(), 0, r);
        GDL_CHECK_LAUNCH("stem_bwd_fused_kernel");
    }
    ProfScope prof("gdl::stem_wgrad_reduce_kernel", PROF_HBM, st, (double)64 * Cin * 49 * 4.0 * (p.nsplit + 1));
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * 4 * 64 / 4 / 16), dim3(256), 0, st, r.partial, dw, p.nsplit, Cin);
    GDL_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
    return GDL_OK;
}

}  // namespace gdl
