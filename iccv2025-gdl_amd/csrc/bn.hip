// bn.hip -- BatchNorm2d (+ReLU, +residual add) forward / backward on NHWC tensors.
//
// Replaces nn.BatchNorm2d (train + eval), nn.ReLU(inplace) and `out += identity` of
// /root/reference/models/backbone.py:45-48,57,62-66,104-105,144.  All of these are HBM
// bound: every kernel here streams 16-byte vectors, keeps per-channel statistics in fp32
// (finalised in double), and reduces deterministically (fixed-order trees, no atomics).
//
// Channel of a vector: with NHWC storage the flat element index modulo C; a thread that
// strides by a multiple of C keeps the same channels for its whole loop, so its scale /
// shift / partial sums live in registers.
#include <stdlib.h>

#include "common.h"
#include "ops.h"
#include "prof.h"

namespace gdl {

constexpr int BN_THREADS = 256;

// ---------------------------------------------------------------- statistics of an existing tensor
// grid.x = tiles; block handles rows [tile*rows_per_tile, ...).  partial[tile][C][2].
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ y, float* __restrict__ partial, int M,
                                                              int C, int rows_per_tile) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ float red[];  // [BN_THREADS/cpr][C][2]
    const int cpr = C / EPC;        // vectors per row
    const int rpp = BN_THREADS / cpr;
    const int vc = threadIdx.x % cpr, vr = threadIdx.x / cpr;
    const int r0 = blockIdx.x * rows_per_tile, r1 = min(M, r0 + rows_per_tile);
    float s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.f;
    if (vr < rpp)
        for (int r = r0 + vr; r < r1; r += rpp) {
            float f[EPC];
            unpack16<T>(*(const uint4*)(y + (size_t)r * C + vc * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s[e] += f[e];
                q[e] += f[e] * f[e];
            }
        }
    if (vr < rpp) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[((size_t)vr * C + vc * EPC + e) * 2 + 0] = s[e];
            red[((size_t)vr * C + vc * EPC + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 2; i += BN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < rpp; ++r) a += red[(size_t)r * C * 2 + i];
        partial[(size_t)blockIdx.x * C * 2 + i] = a;
    }
}

constexpr int BN_STATS_ROWS = 1024;
int bn_stats_tiles(int M) { return ceil_div(M, BN_STATS_ROWS); }

int bn_stats(int dtype, const void* y, float* partial, int M, int C, hipStream_t st) {
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    GDL_REQUIRE(C % epc == 0 && C / epc <= BN_THREADS && BN_THREADS % (C / epc) == 0, "bn_stats: C=%d unsupported", C);
    const int rpp = BN_THREADS / (C / epc);
    const size_t sh = (size_t)rpp * C * 2 * sizeof(float);
    const int tiles = bn_stats_tiles(M);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16>, dim3(tiles), dim3(BN_THREADS), sh, st, (const bf16*)y, partial, M, C,
                           BN_STATS_ROWS);
    else
        hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(tiles), dim3(BN_THREADS), sh, st, (const float*)y, partial, M, C,
                           BN_STATS_ROWS);
    GDL_CHECK_LAUNCH("bn_stats_kernel");
    return GDL_OK;
}

// ---------------------------------------------------------------- finalize (one block per channel)
__device__ __forceinline__ double block_sum_double(double v, double* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// Column sums of a [rows][C][2] float array in double, CH channels per block of 256 threads: thread (rl, cl) adds
// rows rl, rl+RL, .. of channel c0+cl, eight loads in flight, the RL = 256/CH row-lanes are then folded through LDS
// in a fixed order.  Returns the two sums to every thread (of its channel).  These kernels sit between two dependent
// kernels of a chain 80 times per step and cost the step their full duration (skipping them: -0.56 ms of 6.97), so
// CH is picked for ~64 blocks (fin_ch).
constexpr int FIN_THREADS = 256;  // (1024-thread blocks waited for a CU with 16 free wave slots: ~8 us even for 74 rows)
template <int CH>
__device__ __forceinline__ void fin_colsum(const float* __restrict__ partial, int rows, int C, int c, double& s, double& q) {
    constexpr int RL = FIN_THREADS / CH;
    __shared__ double sh[RL][CH][2];
    const int rl = threadIdx.x / CH, cl = threadIdx.x % CH;
    const float2* __restrict__ p2 = (const float2*)partial;
    s = 0.0, q = 0.0;
    // (channels past C, when C is not a multiple of CH, only take part in the barriers)
    for (int t = c < C ? rl : rows; t < rows; t += 8 * RL) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = t + u * RL;
            v[u] = r < rows ? p2[(size_t)r * C + c] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += (double)v[u].x;
            q += (double)v[u].y;
        }
    }
    sh[rl][cl][0] = s;
    sh[rl][cl][1] = q;
    __syncthreads();
    // fixed binary tree over the row-lanes (deterministic; a serial fold by one thread cost ~3 us)
#pragma unroll
    for (int st = RL / 2; st > 0; st >>= 1) {
        if (rl < st) {
            sh[rl][cl][0] += sh[rl + st][cl][0];
            sh[rl][cl][1] += sh[rl + st][cl][1];
        }
        __syncthreads();
    }
    s = sh[0][cl][0];
    q = sh[0][cl][1];
}
static int fin_ch(int C) { return C >= 512 ? 8 : (C >= 256 ? 4 : (C >= 128 ? 2 : 1)); }

// (grid.y = 1 or 2: two independent BatchNorms -- bn2 and the downsample BatchNorm of a block -- share one launch)
template <int CH>
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_train_kernel(BnFinTrain a0, BnFinTrain a1, float eps, float momentum) {
    const BnFinTrain& a = blockIdx.y ? a1 : a0;
    const int C = a.C;
    const int c = blockIdx.x * CH + threadIdx.x % CH;
    double s, q;
    fin_colsum<CH>(a.partial, a.tiles, C, c, s, q);
    if (threadIdx.x < CH && c < C) {
        const double count = a.count;
        const double mean = s / count;
        double var = q / count - mean * mean;  // biased
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)((double)a.gamma[c] * rstd);
        a.save_mean[c] = (float)mean;
        a.save_rstd[c] = (float)rstd;
        a.scale[c] = sc;
        a.shift[c] = (float)((double)a.beta[c] - mean * (double)a.gamma[c] * rstd);
        if (a.running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - (double)momentum) * (double)a.running_mean[c] + (double)momentum * mean);
            a.running_var[c] = (float)((1.0 - (double)momentum) * (double)a.running_var[c] + (double)momentum * unb);
        }
        if (a.nbt && c == 0) *a.nbt += 1;
    }
}
static int launch_fin_train(const BnFinTrain& a0, const BnFinTrain& a1, int n, float eps, float momentum, hipStream_t st) {
    const int ch = fin_ch(a0.C);
    auto kern = ch == 8 ? bn_finalize_train_kernel<8>
                        : (ch == 4 ? bn_finalize_train_kernel<4> : (ch == 2 ? bn_finalize_train_kernel<2> : bn_finalize_train_kernel<1>));
    hipLaunchKernelGGL(kern, dim3(ceil_div(a0.C, ch), n), dim3(FIN_THREADS), 0, st, a0, a1, eps, momentum);
    GDL_CHECK_LAUNCH("bn_finalize_train_kernel");
    return GDL_OK;
}
int bn_finalize_train(const float* partial, int tiles, int C, double count, const float* gamma, const float* beta,
                      float eps, float momentum, float* rm, float* rv, int64_t* nbt, float* save_mean, float* save_rstd,
                      float* scale, float* shift, hipStream_t st) {
    const BnFinTrain a{partial, tiles, C, count, gamma, beta, rm, rv, nbt, save_mean, save_rstd, scale, shift};
    return launch_fin_train(a, a, 1, eps, momentum, st);
}
int bn_finalize_train_pair(const BnFinTrain& a, const BnFinTrain& b, float eps, float momentum, hipStream_t st) {
    GDL_REQUIRE(a.C == b.C, "bn_finalize_train_pair: channel counts differ (%d, %d)", a.C, b.C);
    return launch_fin_train(a, b, 2, eps, momentum, st);
}

__global__ void acc_to_float_kernel(const long long* __restrict__ acc, int n, double inv_scale, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    // (acc[2c + 1] != 0: a block's sum did not fit the fixed-point range or was NaN / inf -- bnacc.h bn_acc_add without a flag word)
    if (c < n) out[c] = acc[2 * c + 1] != 0 ? __builtin_nanf("") : (float)((double)acc[2 * c] * inv_scale);
}
int acc_to_float(const long long* acc, int n, double inv_scale, float* out, hipStream_t st) {
    hipLaunchKernelGGL(acc_to_float_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, acc, n, inv_scale, out);
    GDL_CHECK_LAUNCH("acc_to_float_kernel");
    return GDL_OK;
}

__global__ void bn_finalize_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        const float* __restrict__ rm, const float* __restrict__ rv, float* scale,
                                        float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    scale[c] = (float)((double)gamma[c] * rstd);
    shift[c] = (float)((double)beta[c] - (double)rm[c] * (double)gamma[c] * rstd);
}
int bn_finalize_eval(int C, const float* gamma, const float* beta, float eps, const float* rm, const float* rv,
                     float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, st, C, gamma, beta, eps, rm, rv,
                       scale, shift);
    GDL_CHECK_LAUNCH("bn_finalize_eval_kernel");
    return GDL_OK;
}

// per-channel constants of the EPC consecutive channels a thread owns: 16-byte loads (c0 % EPC == 0),
// not EPC scalar ones -- the scalar form cost more than the data traffic on the small layers
template <int EPC>
__device__ __forceinline__ void ld_chan(const float* __restrict__ p, int c0, float (&v)[EPC]) {
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
        const float4 t = *(const float4*)(p + c0 + 4 * q);
        v[4 * q + 0] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
    }
}

// ---------------------------------------------------------------- forward apply
// out = [relu]( y*scale+shift + residual ),  RES: 0 none, 1 raw tensor, 2 tensor*res_scale+res_shift
// ACC: the BatchNorm constants are not finalized yet -- every block derives them from the integer accumulators of the
// producing convolution(s) (bnacc.h) into LDS, one channel per thread, and block 0 publishes them for the backward
constexpr int BN_ACC_MAXC = 512;
template <typename T, int RES, bool RELU, bool ACC>
__global__ __launch_bounds__(BN_THREADS) void bn_act_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const T* __restrict__ res,
                                                            const float* __restrict__ rscale,
                                                            const float* __restrict__ rshift, T* __restrict__ out,
                                                            uint8_t* __restrict__ bits, size_t nvec, int C,
                                                            size_t stride_vec, BnAccFin fa, BnAccFin fr) {
    constexpr int EPC = TT<T>::EPC;
    __shared__ __attribute__((aligned(16))) float csm[ACC ? 4 : 1][ACC ? BN_ACC_MAXC : 4];
    size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x;
    // ACC: the first trip's vectors are requested BEFORE the constants are derived (accumulator loads, double-precision
    // arithmetic, a barrier): the prologue then hides behind the memory latency it would otherwise add to
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = p0, q0 = p0, q1 = p0;
    const bool live = i < nvec, pair = live && i + stride_vec < nvec;
    if (ACC) {
        if (live) {
            p0 = *(const uint4*)(y + i * EPC);
            if (RES) q0 = *(const uint4*)(res + i * EPC);
        }
        if (pair) {
            p1 = *(const uint4*)(y + (i + stride_vec) * EPC);
            if (RES) q1 = *(const uint4*)(res + (i + stride_vec) * EPC);
        }
        for (int c = threadIdx.x; c < C; c += BN_THREADS) {
            bn_acc_channel(fa, c, blockIdx.x == 0, csm[0][c], csm[1][c]);
            if (RES == 2) {
                if (fr.acc) {
                    bn_acc_channel(fr, c, blockIdx.x == 0, csm[2][c], csm[3][c]);
                } else {
                    csm[2][c] = rscale[c];
                    csm[3][c] = rshift[c];
                }
            }
        }
        __syncthreads();
    }
    if (!live) return;
    const int c0 = (int)((i * EPC) % C);  // stride_vec*EPC is a multiple of C: channels are loop invariant
    float sc[EPC], sf[EPC], rsc[EPC], rsf[EPC];
    if (ACC) {
        ld_chan<EPC>(csm[0], c0, sc);
        ld_chan<EPC>(csm[1], c0, sf);
        if (RES == 2) {
            ld_chan<EPC>(csm[2], c0, rsc);
            ld_chan<EPC>(csm[3], c0, rsf);
        }
    } else {
        ld_chan<EPC>(scale, c0, sc);
        ld_chan<EPC>(shift, c0, sf);
        if (RES == 2) {
            ld_chan<EPC>(rscale, c0, rsc);
            ld_chan<EPC>(rshift, c0, rsf);
        }
    }
    auto one = [&](size_t j, const uint4& yv, const uint4& rv) {
        float f[EPC], g[EPC];
        unpack16<T>(yv, f);
        if (RES) unpack16<T>(rv, g);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float v = f[e] * sc[e] + sf[e];
            if (RES == 1) v += g[e];
            if (RES == 2) v += g[e] * rsc[e] + rsf[e];
            if (RELU) v = v > 0.f ? v : 0.f;
            f[e] = v;
        }
        *(uint4*)(out + j * EPC) = pack16<T>(f);
        if (RELU && bits) {  // sign bits of the output (one byte per 16-byte vector): the backward's ReLU mask
            unsigned mk = 0u;
#pragma unroll
            for (int e = 0; e < EPC; ++e) mk |= (f[e] > 0.f ? 1u : 0u) << e;
            bits[j] = (uint8_t)mk;
        }
    };
    if (ACC) {  // the trip requested above
        one(i, p0, q0);
        if (!pair) return;
        one(i + stride_vec, p1, q1);
        i += 2 * stride_vec;
    }
    // two vectors per trip: both loads are in flight before either is consumed
    for (; i + stride_vec < nvec; i += 2 * stride_vec) {
        const size_t j = i + stride_vec;
        const uint4 y0 = *(const uint4*)(y + i * EPC), y1 = *(const uint4*)(y + j * EPC);
        uint4 r0 = y0, r1 = y1;
        if (RES) {
            r0 = *(const uint4*)(res + i * EPC);
            r1 = *(const uint4*)(res + j * EPC);
        }
        one(i, y0, r0);
        one(j, y1, r1);
    }
    if (i < nvec) {
        const uint4 y0 = *(const uint4*)(y + i * EPC);
        uint4 r0 = y0;
        if (RES) r0 = *(const uint4*)(res + i * EPC);
        one(i, y0, r0);
    }
}

static bool ew_v4() {  // tuning aid: GDL_EW_V4=1 -> 4 instead of 8 vectors per thread on small tensors
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_EW_V4");
        v = e ? atoi(e) : 0;
    }
    return v != 0;
}

// grid sizing for channel-invariant grid-stride loops: total threads is a multiple of C/EPC
static inline void ew_grid(size_t nvec, int cpr, int& blocks, size_t& stride_vec) {
    // threads = blocks*256 must be a multiple of cpr (cpr divides 256 or is a multiple of it handled by lcm)
    // Measured (bench.py kernel table): 8 vectors per thread amortise the per-channel constant loads
    // on the big tensors (34 us vs 50 us per launch with 1 vector per thread); at most 2048 blocks
    // (the kernels keep two vectors in flight per trip); tensors too small to fill 2048 blocks that
    // way get 4 vectors per thread so that the launch still spreads over every CU
    size_t want = (nvec + BN_THREADS * 8 - 1) / (BN_THREADS * 8);
    if (want < 2048 && ew_v4()) want = (nvec + BN_THREADS * 4 - 1) / (BN_THREADS * 4);
    if (want > 2048) want = 2048;
    if (want < 1) want = 1;
    // make blocks*256 % cpr == 0
    size_t unit = 1;
    while ((unit * BN_THREADS) % (size_t)cpr) ++unit;
    want = (want + unit - 1) / unit * unit;
    blocks = (int)want;
    stride_vec = want * BN_THREADS;
}

template <typename T>
static int bn_act_t(const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
                    const float* rshift, int relu, void* out, uint8_t* bits, size_t M, int C, hipStream_t st,
                    const BnAccFin* fa, const BnAccFin* fr) {
    constexpr int EPC = TT<T>::EPC;
    GDL_REQUIRE(C % EPC == 0, "bn_act: C=%d not a multiple of %d", C, EPC);
    const bool acc = fa && fa->acc;
    GDL_REQUIRE(acc || !(fr && fr->acc), "bn_act: accumulators for the residual BatchNorm only");
    GDL_REQUIRE(!acc || C <= BN_ACC_MAXC, "bn_act: C=%d above %d with accumulators", C, BN_ACC_MAXC);
    const BnAccFin none{};
    const BnAccFin& a0 = acc ? *fa : none;
    const BnAccFin& a1 = (acc && fr) ? *fr : none;
    const size_t nvec = M * (size_t)C / EPC;
    int blocks;
    size_t stride;
    ew_grid(nvec, C / EPC, blocks, stride);
    const int resmode = res ? (rscale ? 2 : 1) : 0;
    static char pname[6][96];
    char* pn = pname[resmode * 2 + (relu ? 1 : 0)];
    if (!pn[0]) snprintf(pn, 96, "gdl::bn_act_kernel<%s, %d, %s>", prof_tname<T>(), resmode, relu ? "true" : "false");
    ProfScope prof(pn, PROF_HBM, st, (double)nvec * 16.0 * (resmode ? 3 : 2));
#define BN_ACT_LAUNCH(RM, RL)                                                                                                  \
    do {                                                                                                                      \
        if (acc)                                                                                                              \
            hipLaunchKernelGGL((bn_act_kernel<T, RM, RL, true>), dim3(blocks), dim3(BN_THREADS), 0, st, (const T*)y, scale,   \
                               shift, (const T*)res, rscale, rshift, (T*)out, bits, nvec, C, stride, a0, a1);                \
        else                                                                                                                  \
            hipLaunchKernelGGL((bn_act_kernel<T, RM, RL, false>), dim3(blocks), dim3(BN_THREADS), 0, st, (const T*)y, scale,  \
                               shift, (const T*)res, rscale, rshift, (T*)out, bits, nvec, C, stride, a0, a1);                \
    } while (0)
    if (resmode == 0 && relu) BN_ACT_LAUNCH(0, true);
    if (resmode == 0 && !relu) BN_ACT_LAUNCH(0, false);
    if (resmode == 1 && relu) BN_ACT_LAUNCH(1, true);
    if (resmode == 1 && !relu) BN_ACT_LAUNCH(1, false);
    if (resmode == 2 && relu) BN_ACT_LAUNCH(2, true);
    if (resmode == 2 && !relu) BN_ACT_LAUNCH(2, false);
#undef BN_ACT_LAUNCH
    GDL_CHECK_LAUNCH("bn_act_kernel");
    return GDL_OK;
}
int bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
           const float* rshift, int relu, void* out, size_t M, int C, hipStream_t st, uint8_t* relu_bits, const BnAccFin* fa,
           const BnAccFin* fr) {
    GDL_REQUIRE(!relu_bits || relu, "bn_act: relu_bits without relu");
    if (dtype == GDL_BF16) return bn_act_t<bf16>(y, scale, shift, res, rscale, rshift, relu, out, relu_bits, M, C, st, fa, fr);
    return bn_act_t<float>(y, scale, shift, res, rscale, rshift, relu, out, relu_bits, M, C, st, fa, fr);
}

// ---------------------------------------------------------------- backward
// pass 1: partial[block][C][2] = { sum g', sum g'*xhat },  g' = MASK ? g*(y*scale+shift>0) : g
template <typename T, bool MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   float* __restrict__ partial, size_t nvec, int C,
                                                                   size_t stride_vec) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ float red[];  // [rpp][C][2]
    const int cpr = C / EPC;
    size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x;
    const int c0 = (int)((i * EPC) % C);
    float sc[EPC], sf[EPC], mu[EPC], rs[EPC], s1[EPC], s2[EPC];
    ld_chan<EPC>(mean, c0, mu);
    ld_chan<EPC>(rstd, c0, rs);
    if (MASK) {
        ld_chan<EPC>(scale, c0, sc);
        ld_chan<EPC>(shift, c0, sf);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    auto one = [&](const uint4& gq, const uint4& yq) {
        float gv[EPC], yv[EPC];
        unpack16<T>(gq, gv);
        unpack16<T>(yq, yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float gg = gv[e];
            if (MASK) gg = (yv[e] * sc[e] + sf[e] > 0.f) ? gg : 0.f;
            s1[e] += gg;
            s2[e] += gg * ((yv[e] - mu[e]) * rs[e]);
        }
    };
    for (; i + stride_vec < nvec; i += 2 * stride_vec) {
        const size_t j = i + stride_vec;
        const uint4 g0 = *(const uint4*)(g + i * EPC), g1 = *(const uint4*)(g + j * EPC);
        const uint4 y0 = *(const uint4*)(y + i * EPC), y1 = *(const uint4*)(y + j * EPC);
        one(g0, y0);
        one(g1, y1);
    }
    if (i < nvec) one(*(const uint4*)(g + i * EPC), *(const uint4*)(y + i * EPC));
    // 256 % cpr == 0: thread t holds channel vector t % cpr of row-lane t / cpr
    const int rpp = BN_THREADS / cpr, vr = threadIdx.x / cpr;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[((size_t)vr * C + c0 + e) * 2 + 0] = s1[e];
        red[((size_t)vr * C + c0 + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * 2; k += BN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < rpp; ++r) a += red[(size_t)r * C * 2 + k];
        st_agent(partial + (size_t)blockIdx.x * C * 2 + k, a);
    }
}

// one block per 16384 elements (8 bf16 / 16 f32 vectors per thread), at most 2048 blocks; tensors
// that would not fill 2048 blocks that way get one block per 8192 elements
int bn_bwd_blocks(size_t M, int C) {
    size_t b = (M * (size_t)C + 16383) / 16384;
    if (b < 2048 && ew_v4()) b = (M * (size_t)C + 8191) / 8192;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
static int bn_bwd_reduce_t(const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                           const float* rstd, int relu_mask, float* partial, size_t M, int C, hipStream_t st) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    GDL_REQUIRE(C % EPC == 0 && cpr <= BN_THREADS && BN_THREADS % cpr == 0, "bn_bwd: C=%d unsupported", C);
    const size_t nvec = M * (size_t)C / EPC;
    const int blocks = bn_bwd_blocks(M, C);
    const size_t stride = (size_t)blocks * BN_THREADS;  // multiple of cpr since 256 % cpr == 0
    const size_t sh = (size_t)(BN_THREADS / cpr) * C * 2 * sizeof(float);
    static char pname[2][96];
    char* pn = pname[relu_mask ? 1 : 0];
    if (!pn[0]) snprintf(pn, 96, "gdl::bn_bwd_reduce_kernel<%s, %s>", prof_tname<T>(), relu_mask ? "true" : "false");
    ProfScope prof(pn, PROF_HBM, st, (double)nvec * 16.0 * 2);
    if (relu_mask)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), dim3(blocks), dim3(BN_THREADS), sh, st, (const T*)g,
                           (const T*)y, scale, shift, mean, rstd, partial, nvec, C, stride);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false>), dim3(blocks), dim3(BN_THREADS), sh, st, (const T*)g,
                           (const T*)y, scale, shift, mean, rstd, partial, nvec, C, stride);
    GDL_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    return GDL_OK;
}
int bn_bwd_reduce(int dtype, const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                  const float* rstd, int relu_mask, float* partial, size_t M, int C, hipStream_t st) {
    if (dtype == GDL_BF16) return bn_bwd_reduce_t<bf16>(g, y, scale, shift, mean, rstd, relu_mask, partial, M, C, st);
    return bn_bwd_reduce_t<float>(g, y, scale, shift, mean, rstd, relu_mask, partial, M, C, st);
}

// Fused head of a BasicBlock's backward: do2 = dz * (z > 0) (written out: the identity / downsample
// path and both BN applies read it) together with the reductions of bn2 and, when the block has a
// downsample branch, of its BatchNorm -- one pass over dz, z, y2 (, yd) instead of relu_bwd +
// two reduce kernels.  partial2 / partiald: [blocks][C][2].
// PRE: dz already carries the ReLU mask (the data gradient that produced it applied the block output's sign bits in its
// epilogue, conv_igemm.hip): z is not read and do2 (== dz) not written -- two tensor passes instead of four.
template <typename T, bool DS, bool PRE>
__global__ __launch_bounds__(BN_THREADS) void block_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                                      const T* __restrict__ y2, const T* __restrict__ yd,
                                                                      const float* __restrict__ mean2,
                                                                      const float* __restrict__ rstd2,
                                                                      const float* __restrict__ meand,
                                                                      const float* __restrict__ rstdd, T* __restrict__ do2,
                                                                      float* __restrict__ partial2,
                                                                      float* __restrict__ partiald, size_t nvec, int C,
                                                                      size_t stride_vec) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ float red[];  // [rpp][C][2] (+ the same again for the downsample branch)
    const int cpr = C / EPC;
    size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x;
    const int c0 = (int)((i * EPC) % C);
    float mu2[EPC], rs2[EPC], mud[EPC], rsd[EPC], a1[EPC], a2[EPC], b1[EPC], b2[EPC];
    ld_chan<EPC>(mean2, c0, mu2);
    ld_chan<EPC>(rstd2, c0, rs2);
    if (DS) {
        ld_chan<EPC>(meand, c0, mud);
        ld_chan<EPC>(rstdd, c0, rsd);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) a1[e] = a2[e] = b1[e] = b2[e] = 0.f;
    auto one = [&](size_t j, const uint4& gq, const uint4& zq, const uint4& yq, const uint4& dq) {
        float gv[EPC], zv[EPC], yv[EPC], dv[EPC];
        unpack16<T>(gq, gv);
        if (!PRE) unpack16<T>(zq, zv);
        unpack16<T>(yq, yv);
        if (DS) unpack16<T>(dq, dv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float gg = PRE ? gv[e] : (zv[e] > 0.f ? gv[e] : 0.f);
            gv[e] = gg;
            a1[e] += gg;
            a2[e] += gg * ((yv[e] - mu2[e]) * rs2[e]);
            if (DS) b2[e] += gg * ((dv[e] - mud[e]) * rsd[e]);
        }
        if (!PRE) *(uint4*)(do2 + j * EPC) = pack16<T>(gv);
    };
    for (; i + stride_vec < nvec; i += 2 * stride_vec) {
        const size_t j = i + stride_vec;
        const uint4 g0 = *(const uint4*)(dz + i * EPC), g1 = *(const uint4*)(dz + j * EPC);
        uint4 z0 = g0, z1 = g1;
        if (!PRE) {
            z0 = *(const uint4*)(z + i * EPC);
            z1 = *(const uint4*)(z + j * EPC);
        }
        const uint4 y0 = *(const uint4*)(y2 + i * EPC), y1 = *(const uint4*)(y2 + j * EPC);
        uint4 d0 = y0, d1 = y1;
        if (DS) {
            d0 = *(const uint4*)(yd + i * EPC);
            d1 = *(const uint4*)(yd + j * EPC);
        }
        one(i, g0, z0, y0, d0);
        one(j, g1, z1, y1, d1);
    }
    if (i < nvec) {
        const uint4 y0 = *(const uint4*)(y2 + i * EPC);
        uint4 d0 = y0;
        if (DS) d0 = *(const uint4*)(yd + i * EPC);
        const uint4 g0 = *(const uint4*)(dz + i * EPC);
        one(i, g0, PRE ? g0 : *(const uint4*)(z + i * EPC), y0, d0);
    }
    const int rpp = BN_THREADS / cpr, vr = threadIdx.x / cpr;
    float* redd = red + (size_t)rpp * C * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[((size_t)vr * C + c0 + e) * 2 + 0] = a1[e];
        red[((size_t)vr * C + c0 + e) * 2 + 1] = a2[e];
        if (DS) {
            redd[((size_t)vr * C + c0 + e) * 2 + 0] = a1[e];  // sum of do2 is shared by both BatchNorms
            redd[((size_t)vr * C + c0 + e) * 2 + 1] = b2[e];
        }
    }
    (void)b1;
    __syncthreads();
    for (int k = threadIdx.x; k < C * 2; k += BN_THREADS) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rpp; ++r) {
            a += red[(size_t)r * C * 2 + k];
            if (DS) b += redd[(size_t)r * C * 2 + k];
        }
        st_agent(partial2 + (size_t)blockIdx.x * C * 2 + k, a);
        if (DS) st_agent(partiald + (size_t)blockIdx.x * C * 2 + k, b);
    }
}

template <typename T>
static int block_bwd_reduce_t(const void* dz, const void* z, const void* y2, const void* yd, const float* mean2,
                              const float* rstd2, const float* meand, const float* rstdd, void* do2, float* partial2,
                              float* partiald, size_t M, int C, hipStream_t st, bool pre) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    GDL_REQUIRE(C % EPC == 0 && cpr <= BN_THREADS && BN_THREADS % cpr == 0, "block_bwd_reduce: C=%d unsupported", C);
    const size_t nvec = M * (size_t)C / EPC;
    const int blocks = bn_bwd_blocks(M, C);
    const size_t stride = (size_t)blocks * BN_THREADS;
    const size_t sh = (size_t)(BN_THREADS / cpr) * C * 2 * sizeof(float) * (yd ? 2 : 1);
    static char pname[4][96];
    char* pn = pname[(yd ? 1 : 0) + (pre ? 2 : 0)];
    if (!pn[0])
        snprintf(pn, 96, "gdl::block_bwd_reduce_kernel<%s, %s, %s>", prof_tname<T>(), yd ? "true" : "false", pre ? "true" : "false");
    ProfScope prof(pn, PROF_HBM, st, (double)nvec * 16.0 * ((yd ? 5 : 4) - (pre ? 2 : 0)));
#define GDL_BBR(DSV, PREV)                                                                                                  \
    hipLaunchKernelGGL((block_bwd_reduce_kernel<T, DSV, PREV>), dim3(blocks), dim3(BN_THREADS), sh, st, (const T*)dz,        \
                       (const T*)z, (const T*)y2, (const T*)yd, mean2, rstd2, meand, rstdd, (T*)do2, partial2, partiald, nvec, \
                       C, stride)
    if (yd && pre) GDL_BBR(true, true);
    if (yd && !pre) GDL_BBR(true, false);
    if (!yd && pre) GDL_BBR(false, true);
    if (!yd && !pre) GDL_BBR(false, false);
#undef GDL_BBR
    GDL_CHECK_LAUNCH("block_bwd_reduce_kernel");
    return GDL_OK;
}
int block_bwd_reduce(int dtype, const void* dz, const void* z, const void* y2, const void* yd, const float* mean2,
                     const float* rstd2, const float* meand, const float* rstdd, void* do2, float* partial2,
                     float* partiald, size_t M, int C, hipStream_t st, bool premasked) {
    GDL_REQUIRE(premasked || (z && do2), "block_bwd_reduce: z / do2 missing");
    if (dtype == GDL_BF16)
        return block_bwd_reduce_t<bf16>(dz, z, y2, yd, mean2, rstd2, meand, rstdd, do2, partial2, partiald, M, C, st, premasked);
    return block_bwd_reduce_t<float>(dz, z, y2, yd, mean2, rstd2, meand, rstdd, do2, partial2, partiald, M, C, st, premasked);
}

// dgamma = sum g'*xhat, dbeta = sum g'; coef[0][c] = dbeta/M, coef[1][c] = dgamma/M
template <int CH>
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(BnFinBwd a0, BnFinBwd a1) {
    const BnFinBwd& p = blockIdx.y ? a1 : a0;
    const int C = p.C;
    const int c = blockIdx.x * CH + threadIdx.x % CH;
    double a, b;
    fin_colsum<CH>(p.partial, p.blocks, C, c, a, b);
    if (threadIdx.x < CH && c < C) {
        p.dbeta[c] = (float)a;
        p.dgamma[c] = (float)b;
        p.coef[c] = (float)(a / p.count);
        p.coef[C + c] = (float)(b / p.count);
    }
}
static int launch_fin_bwd(const BnFinBwd& a0, const BnFinBwd& a1, int n, hipStream_t st) {
    const int ch = fin_ch(a0.C);
    auto kern = ch == 8 ? bn_bwd_finalize_kernel<8>
                        : (ch == 4 ? bn_bwd_finalize_kernel<4> : (ch == 2 ? bn_bwd_finalize_kernel<2> : bn_bwd_finalize_kernel<1>));
    ProfScope prof("gdl::bn_bwd_finalize_kernel", PROF_HBM, st, (double)n * a0.blocks * a0.C * 8.0);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a0.C, ch), n), dim3(FIN_THREADS), 0, st, a0, a1);
    GDL_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    return GDL_OK;
}
int bn_bwd_finalize(const float* partial, int blocks, int C, double count, float* dgamma, float* dbeta, float* coef,
                    hipStream_t st) {
    const BnFinBwd a{partial, blocks, C, count, dgamma, dbeta, coef};
    return launch_fin_bwd(a, a, 1, st);
}
int bn_bwd_finalize_pair(const BnFinBwd& a, const BnFinBwd& b, hipStream_t st) {
    GDL_REQUIRE(a.C == b.C, "bn_bwd_finalize_pair: channel counts differ (%d, %d)", a.C, b.C);
    return launch_fin_bwd(a, b, 2, st);
}

// pass 2: dy = gamma*rstd*(g' - coef0 - xhat*coef1)
template <typename T, bool MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ coef, T* __restrict__ dy,
                                                                  size_t nvec, int C, size_t stride_vec) {
    constexpr int EPC = TT<T>::EPC;
    size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x;
    if (i >= nvec) return;
    const int c0 = (int)((i * EPC) % C);
    float sc[EPC], sf[EPC], mu[EPC], rs[EPC], gr[EPC], k1[EPC], k2[EPC];
    ld_chan<EPC>(mean, c0, mu);
    ld_chan<EPC>(rstd, c0, rs);
    ld_chan<EPC>(gamma, c0, gr);
    ld_chan<EPC>(coef, c0, k1);
    ld_chan<EPC>(coef + C, c0, k2);
    if (MASK) {
        ld_chan<EPC>(scale, c0, sc);
        ld_chan<EPC>(shift, c0, sf);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) gr[e] *= rs[e];
    auto one = [&](size_t j, const uint4& gq, const uint4& yq) {
        float gv[EPC], yv[EPC];
        unpack16<T>(gq, gv);
        unpack16<T>(yq, yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float gg = gv[e];
            if (MASK) gg = (yv[e] * sc[e] + sf[e] > 0.f) ? gg : 0.f;
            gv[e] = gr[e] * (gg - k1[e] - (yv[e] - mu[e]) * rs[e] * k2[e]);
        }
        *(uint4*)(dy + j * EPC) = pack16<T>(gv);
    };
    for (; i + stride_vec < nvec; i += 2 * stride_vec) {
        const size_t j = i + stride_vec;
        const uint4 g0 = *(const uint4*)(g + i * EPC), g1 = *(const uint4*)(g + j * EPC);
        const uint4 y0 = *(const uint4*)(y + i * EPC), y1 = *(const uint4*)(y + j * EPC);
        one(i, g0, y0);
        one(j, g1, y1);
    }
    if (i < nvec) one(i, *(const uint4*)(g + i * EPC), *(const uint4*)(y + i * EPC));
}
template <typename T>
static int bn_bwd_apply_t(const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                          const float* rstd, const float* gamma, const float* coef, int relu_mask, void* dy, size_t M,
                          int C, hipStream_t st) {
    constexpr int EPC = TT<T>::EPC;
    const size_t nvec = M * (size_t)C / EPC;
    int blocks;
    size_t stride;
    ew_grid(nvec, C / EPC, blocks, stride);
    static char pname[2][96];
    char* pn = pname[relu_mask ? 1 : 0];
    if (!pn[0]) snprintf(pn, 96, "gdl::bn_bwd_apply_kernel<%s, %s>", prof_tname<T>(), relu_mask ? "true" : "false");
    ProfScope prof(pn, PROF_HBM, st, (double)nvec * 16.0 * 3);
    if (relu_mask)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(blocks), dim3(BN_THREADS), 0, st, (const T*)g, (const T*)y,
                           scale, shift, mean, rstd, gamma, coef, (T*)dy, nvec, C, stride);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(blocks), dim3(BN_THREADS), 0, st, (const T*)g,
                           (const T*)y, scale, shift, mean, rstd, gamma, coef, (T*)dy, nvec, C, stride);
    GDL_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return GDL_OK;
}
// The two BatchNorms that share one upstream gradient (bn2 and the downsample BatchNorm of a block): g is read once,
//   dyA = gammaA*rstdA*(g - kA1 - xhatA*kA2),  dyB likewise from yB.  No ReLU mask (g is already masked by the block).
struct BnApplySide {
    const void* y;
    const float *mean, *rstd, *gamma, *coef;
    void* dy;
};
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply2_kernel(const T* __restrict__ g, BnApplySide A, BnApplySide B,
                                                                   size_t nvec, int C, size_t stride_vec) {
    constexpr int EPC = TT<T>::EPC;
    size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x;
    if (i >= nvec) return;
    const int c0 = (int)((i * EPC) % C);
    float mu[2][EPC], rs[2][EPC], gr[2][EPC], k1[2][EPC], k2[2][EPC];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const BnApplySide& P = s ? B : A;
        ld_chan<EPC>(P.mean, c0, mu[s]);
        ld_chan<EPC>(P.rstd, c0, rs[s]);
        ld_chan<EPC>(P.gamma, c0, gr[s]);
        ld_chan<EPC>(P.coef, c0, k1[s]);
        ld_chan<EPC>(P.coef + C, c0, k2[s]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) gr[s][e] *= rs[s][e];
    }
    const T *ya = (const T*)A.y, *yb = (const T*)B.y;
    T *da = (T*)A.dy, *db = (T*)B.dy;
    for (; i < nvec; i += stride_vec) {
        const uint4 gq = *(const uint4*)(g + i * EPC), aq = *(const uint4*)(ya + i * EPC), bq = *(const uint4*)(yb + i * EPC);
        float gv[EPC], yv[EPC], o[EPC];
        unpack16<T>(gq, gv);
        unpack16<T>(aq, yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = gr[0][e] * (gv[e] - k1[0][e] - (yv[e] - mu[0][e]) * rs[0][e] * k2[0][e]);
        *(uint4*)(da + i * EPC) = pack16<T>(o);
        unpack16<T>(bq, yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = gr[1][e] * (gv[e] - k1[1][e] - (yv[e] - mu[1][e]) * rs[1][e] * k2[1][e]);
        *(uint4*)(db + i * EPC) = pack16<T>(o);
    }
}
int bn_bwd_apply2(int dtype, const void* g, const void* yA, const float* meanA, const float* rstdA, const float* gammaA,
                  const float* coefA, void* dyA, const void* yB, const float* meanB, const float* rstdB, const float* gammaB,
                  const float* coefB, void* dyB, size_t M, int C, hipStream_t st) {
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    const size_t nvec = M * (size_t)C / epc;
    int blocks;
    size_t stride;
    ew_grid(nvec, C / epc, blocks, stride);
    const BnApplySide A{yA, meanA, rstdA, gammaA, coefA, dyA}, B{yB, meanB, rstdB, gammaB, coefB, dyB};
    ProfScope prof(dtype == GDL_BF16 ? "gdl::bn_bwd_apply2_kernel<gdl::bf16>" : "gdl::bn_bwd_apply2_kernel<float>", PROF_HBM, st,
                   (double)nvec * 16.0 * 5);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(bn_bwd_apply2_kernel<bf16>, dim3(blocks), dim3(BN_THREADS), 0, st, (const bf16*)g, A, B, nvec, C, stride);
    else
        hipLaunchKernelGGL(bn_bwd_apply2_kernel<float>, dim3(blocks), dim3(BN_THREADS), 0, st, (const float*)g, A, B, nvec, C,
                           stride);
    GDL_CHECK_LAUNCH("bn_bwd_apply2_kernel");
    return GDL_OK;
}

int bn_bwd_apply(int dtype, const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                 const float* rstd, const float* gamma, const float* coef, int relu_mask, void* dy, size_t M, int C,
                 hipStream_t st) {
    if (dtype == GDL_BF16)
        return bn_bwd_apply_t<bf16>(g, y, scale, shift, mean, rstd, gamma, coef, relu_mask, dy, M, C, st);
    return bn_bwd_apply_t<float>(g, y, scale, shift, mean, rstd, gamma, coef, relu_mask, dy, M, C, st);
}

// ---------------------------------------------------------------- relu backward: dx = dy * (out > 0)
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ out,
                                                              T* __restrict__ dx, size_t nvec) {
    constexpr int EPC = TT<T>::EPC;
    for (size_t i = blockIdx.x * (size_t)BN_THREADS + threadIdx.x; i < nvec; i += (size_t)gridDim.x * BN_THREADS) {
        float d[EPC], o[EPC];
        unpack16<T>(*(const uint4*)(dy + i * EPC), d);
        unpack16<T>(*(const uint4*)(out + i * EPC), o);
#pragma unroll
        for (int e = 0; e < EPC; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        *(uint4*)(dx + i * EPC) = pack16<T>(d);
    }
}
int relu_bwd(int dtype, const void* dy, const void* out, void* dx, size_t n, hipStream_t st) {
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    GDL_REQUIRE(n % epc == 0, "relu_bwd: n not a multiple of %d", epc);
    const size_t nvec = n / epc;
    size_t blocks = (nvec + BN_THREADS - 1) / BN_THREADS;
    if (blocks > 4096) blocks = 4096;
    ProfScope prof(dtype == GDL_BF16 ? "gdl::relu_bwd_kernel<gdl::bf16>" : "gdl::relu_bwd_kernel<float>", PROF_HBM, st,
                   (double)nvec * 16.0 * 3);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(relu_bwd_kernel<bf16>, dim3((int)blocks), dim3(BN_THREADS), 0, st, (const bf16*)dy,
                           (const bf16*)out, (bf16*)dx, nvec);
    else
        hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3((int)blocks), dim3(BN_THREADS), 0, st, (const float*)dy,
                           (const float*)out, (float*)dx, nvec);
    GDL_CHECK_LAUNCH("relu_bwd_kernel");
    return GDL_OK;
}

}  // namespace gdl
