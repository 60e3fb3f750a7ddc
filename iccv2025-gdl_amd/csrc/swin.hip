// swin.hip -- the non-GEMM operators of the Swin visual encoder (the reference's
// /root/reference/models/swin_transformer.py, which its DGL script never reaches: SURVEY G5 -- a new composition).
//
// Tokens live as rows [N*L][ld] of the activation type T (bf16 / f32), ld = the channel count rounded up to a multiple
// of 64 (96 -> 128 in stage 1 of Swin-T; every other width already is one), padding columns always zero.  Every Linear is
// a 1x1 convolution of this library (conv_igemm.hip / conv_wgrad.hip: forward, data gradient, weight gradient on MFMA),
// whose weights are zero-padded copies made by `swin_pack_matrix`; a QKV row is three ld-wide segments [q | k | v], so a
// head's 32 channels sit at h*32 of each segment.  What this file adds:
//
//   patch_gather       4x4 patches of the [B,3,T,H,W] float32 frames -> GEMM rows [N*Hp*Wp][64] (48 real columns,
//                      order (c, kh, kw) = the flattened Conv2d weight, swin_transformer.py:463,480)
//   bias_act           y += b  |  u = y + b, y = gelu(u)  |  y = y + b + residual           (Mlp :26-42, block :287-290)
//   layernorm fwd/bwd  16 / 32 / 64 lanes per token row of 16-byte vectors, up to four row groups of loads in flight,
//                      two-pass statistics over the real channels in registers; the backward adds the residual
//                      gradient and leaves per-block partials of d(gamma), d(beta)
//   colsum / gelu_bwd  bias gradients (column sums, fixed order), fused with the GELU derivative where one precedes it
//   window attention   (S)W-MSA without materialising rolled / partitioned copies (:124-157, :256-285): a window slot is
//                      mapped to its token of the un-rolled grid on the fly, the shift mask comes from the slots' region
//                      ids (:222-240), the relative-position bias from the (2ws-1)^2 table (:103-113).  One wave per
//                      (window, head).  bf16: the 49 x 49 x 32 products run on the matrix cores (tokens padded to a
//                      64 x 64 grid of 16x16x32 MFMA tiles, softmax on the accumulator layout); f32: plain FMAs, the
//                      exact-mode arithmetic.  The backward recomputes the probabilities and sums d(bias table) per
//                      (image, window group, head) in fixed order.
//   merge gather / scatter   PatchMerging's 2x2 concatenation (:336-344) and its adjoint
//   token mean fwd / bwd     the final AdaptiveAvgPool2d over the 7x7 tokens (:629-631)
//   pack / unpack            float32 parameters [n][k] <-> zero-padded (and transposed) kernel layouts, segment-wise;
//                            all of a step's conversions in one launch each way (descriptor table)
//
// All reductions are fixed-order (partials + a second kernel): two runs are bit-identical.
#include "common.h"
#include "ops.h"
#include "prof.h"

namespace gdl {

constexpr int SW_HD = 32;      // head dimension of every Swin-T / -S / -B stage (dim / heads)
constexpr int SW_MAXT = 49;    // tokens per window (7 x 7)
constexpr int SW_PD = SW_HD + 4;  // LDS pitch of a 32-float row: 16-byte aligned (float4 broadcast reads)
constexpr int SW_MAXV = 24;    // row elements per lane in the LayerNorm kernels: ld <= 1536

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------ patch embedding rows
template <typename T>
__global__ __launch_bounds__(256) void swin_patch_gather_kernel(const float* __restrict__ x, T* __restrict__ a, int B, int Tt,
                                                                int H, int W, int p) {
    const int Hp = H / p, Wp = W / p;
    const size_t total = (size_t)B * Tt * Hp * Wp * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int col = (int)(i & 63);
        size_t r = i >> 6;
        const int wp = (int)(r % Wp);
        r /= Wp;
        const int hp = (int)(r % Hp);
        r /= Hp;
        const int t = (int)(r % Tt), b = (int)(r / Tt);
        float v = 0.f;
        if (col < 3 * p * p) {
            const int c = col / (p * p), kh = (col / p) % p, kw = col % p;
            v = x[((((size_t)b * 3 + c) * Tt + t) * H + hp * p + kh) * W + wp * p + kw];
        }
        storeT(a + i, v);
    }
}

// patch 4, bf16: a thread makes eight consecutive columns of a token's row = two filter rows (c, kh), (c, kh + 1) of four pixels
// each -- two 16-byte loads, one 16-byte store; the eight threads of a token store its 128-byte row, neighbouring tokens read
// neighbouring 16-byte runs of the same image rows.  (One element per thread: 142 us for the 192 frames of config 5, 1.35 TB/s.)
__global__ __launch_bounds__(256) void swin_patch_gather4_kernel(const float* __restrict__ x, bf16* __restrict__ a, int B, int Tt, int H,
                                                                 int W) {
    const int Hp = H / 4, Wp = W / 4;
    const size_t total = (size_t)B * Tt * Hp * Wp * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i & 7);
        size_t r = i >> 3;
        const int wp = (int)(r % Wp);
        r /= Wp;
        const int hp = (int)(r % Hp);
        r /= Hp;
        const int t = (int)(r % Tt), b = (int)(r / Tt);
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (v < 6) {
            const int c = v >> 1, kh = (v & 1) * 2;
            const float* src = x + ((((size_t)b * 3 + c) * Tt + t) * H + hp * 4 + kh) * W + wp * 4;
            const float4 r0 = *(const float4*)src, r1 = *(const float4*)(src + W);
            o = make_uint4(pack2bf(r0.x, r0.y), pack2bf(r0.z, r0.w), pack2bf(r1.x, r1.y), pack2bf(r1.z, r1.w));
        }
        ((uint4*)a)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------ bias / GELU / residual
// mode 0: y = y + b;  mode 1: u = y + b (stored), y = gelu(u);  mode 2: y = y + b + res
template <typename T, int MODE>
__global__ __launch_bounds__(256) void swin_bias_act_kernel(T* __restrict__ y, const float* __restrict__ bias, T* __restrict__ u,
                                                            const T* __restrict__ res, size_t M, int ld) {
    constexpr int EPC = TT<T>::EPC;
    const int vpr = ld / EPC;
    const size_t total = M * vpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c0 = (int)(i % vpr) * EPC;
        float f[EPC], r[EPC];
        unpack16<T>(((const uint4*)y)[i], f);
        if (MODE == 2) unpack16<T>(((const uint4*)res)[i], r);
        float bv[EPC];
#pragma unroll
        for (int e4 = 0; e4 < EPC / 4; ++e4) *(float4*)&bv[4 * e4] = *(const float4*)(bias + c0 + 4 * e4);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            f[e] += bv[e];
            if (MODE == 2) f[e] += r[e];
        }
        if (MODE == 1) {
            ((uint4*)u)[i] = pack16<T>(f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = gelu_val<T>(roundT<T>(f[e]));
        }
        ((uint4*)y)[i] = pack16<T>(f);
    }
}

// ------------------------------------------------------------------------------------------------ stochastic depth
// DropPath (timm.models.layers.drop_path as the reference's blocks use it, swin_transformer.py:218, 290, 293): the residual
// branch of frame n is multiplied by scale[n] (0, or 1 / keep_prob) before it joins the stream.
//   out[row][c] = (res ? res[row][c] : 0) + scale[row / L] * y[row][c]        (fp32, rounded once; out may be y)
// The forward uses it with the residual, the backward without (the branch's share of the stream's gradient).  Only models with
// drop_path_rate > 0 in training mode come here: the Linears' epilogues add the residual themselves otherwise.
template <typename T>
__global__ __launch_bounds__(256) void swin_drop_path_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                             const float* __restrict__ scale, T* __restrict__ out, size_t M, int L,
                                                             int ld) {
    constexpr int EPC = TT<T>::EPC;
    const int vpr = ld / EPC;
    const size_t total = M * vpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float s = scale[(i / vpr) / L];
        float f[EPC], r[EPC];
        unpack16<T>(((const uint4*)y)[i], f);
        if (res) unpack16<T>(((const uint4*)res)[i], r);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = res ? fmaf(s, f[e], r[e]) : s * f[e];
        ((uint4*)out)[i] = pack16<T>(f);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// A row is ld / EPC 16-byte vectors; `lpr` lanes (a power of two <= 64, host-chosen: the next one >= the vector count)
// share a row, 64 / lpr rows per wave, each lane up to SW_MAXVPL vectors (vector sub + lpr*i).  A 128-channel bf16 row is
// 16 vectors: four rows per wave and one 16-byte load per lane instead of 2-byte loads.  stats[row] = (mean, rstd),
// two-pass over the C real channels (padding columns hold zeros and are written as zeros).

// all-reduce over the lpr (16, 32 or 64) lanes that share a row.  The 16 lanes of a DPP row through four rotate-and-add VALU
// instructions (row_ror:8 / 4 / 2 / 1: the rotation butterfly gives every lane the same association tree, so the 16 results are
// bit-identical) instead of four ds_bpermute round trips through the LDS crossbar -- a row's two dependent reductions were what
// the LayerNorm kernels spent their time on (221 -> 144 us backward at stage 1 came from loads alone; the forward did not move until
// this) -- then one or two cross-row exchanges.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group_sum(float v, int lpr) {
    v += dpp_rot<0x128>(v);
    v += dpp_rot<0x124>(v);
    v += dpp_rot<0x122>(v);
    v += dpp_rot<0x121>(v);
    if (lpr > 16) v += __shfl_xor(v, 16, 64);
    if (lpr > 32) v += __shfl_xor(v, 32, 64);
    return v;
}
// the same rotation butterfly over one DPP row of 16 lanes (every lane gets the result)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_rot<0x128>(v);
    v += dpp_rot<0x124>(v);
    v += dpp_rot<0x122>(v);
    v += dpp_rot<0x121>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_rot<0x128>(v));
    v = fmaxf(v, dpp_rot<0x124>(v));
    v = fmaxf(v, dpp_rot<0x122>(v));
    v = fmaxf(v, dpp_rot<0x121>(v));
    return v;
}
// VPL: vectors per lane (compile-time bound of ceil(vectors per row / lpr)); U: row groups in flight per wave and
// iteration -- all their loads are issued before the first reduction, so a lane keeps U x VPL 16-byte loads in flight
// instead of one (the one-row form ran at ~1 TB/s).
template <typename T, int VPL, int U>
__global__ __launch_bounds__(256) void swin_ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float2* __restrict__ stats, size_t M, int C, int ld, int lpr) {
    constexpr int EPC = TT<T>::EPC;
    const int lane = threadIdx.x & 63, sub = lane & (lpr - 1), rpw = 64 / lpr;
    const int vpr = ld / EPC;
    const float invC = 1.f / (float)C;
    float gm[VPL][EPC], bt[VPL][EPC];  // this lane's columns are the same for every row
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = (sub + lpr * i) * EPC + e;
            const bool real = sub + lpr * i < vpr && c < C;
            gm[i][e] = real ? gamma[c] : 0.f;
            bt[i][e] = real ? beta[c] : 0.f;
        }
    const size_t stride = (size_t)gridDim.x * 4 * rpw;
    // the NEXT iteration's rows are requested before this iteration's are reduced and stored: a wave's loads stay in flight through
    // its reductions (with the request at the top of the iteration it served, three waves per SIMD left the memory pipe idle for
    // a third of the time: 3.5 TB/s at stage 1)
    auto request = [&](size_t b, uint4 (&q)[U][VPL]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t row = b + u * stride;
            if (row < M) {
                const uint4* xr = (const uint4*)(x + row * ld);
#pragma unroll
                for (int i = 0; i < VPL; ++i)
                    if (sub + lpr * i < vpr) q[u][i] = xr[sub + lpr * i];
            }
        }
    };
    size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / lpr;
    uint4 vq[U][VPL], vn[U][VPL];  // (packed until a row group's turn: U x VPL x 4 registers in flight instead of U x VPL x EPC)
    request(base, vq);
    for (; base < M; base += stride * U) {
        request(base + stride * U, vn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t row = base + u * stride;
            if (row >= M) continue;  // (whole row groups drop out together: the shuffles below stay inside a group)
            float v[1][VPL][EPC];
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) unpack16<T>(vq[u][i], v[0][i]);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) s += v[0][i][e];  // padding columns are zero
            const float mu = group_sum(s, lpr) * invC;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) {
                    const int c0 = (sub + lpr * i) * EPC;
#pragma unroll
                    for (int e = 0; e < EPC; ++e)
                        if (c0 + e < C) q += (v[0][i][e] - mu) * (v[0][i][e] - mu);
                }
            const float rstd = rsqrtf(group_sum(q, lpr) * invC + 1e-5f);
            if (sub == 0) stats[row] = make_float2(mu, rstd);
            uint4* yr = (uint4*)(y + row * ld);
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) {
                    const int c0 = (sub + lpr * i) * EPC;
                    float o[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] = c0 + e < C ? (v[0][i][e] - mu) * rstd * gm[i][e] + bt[i][e] : 0.f;
                    yr[sub + lpr * i] = pack16<T>(o);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < VPL; ++i) vq[u][i] = vn[u][i];
    }
}

// dx = (add ? add : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  partial[blk][0][c] = sum dy*xhat,
// partial[blk][1][c] = sum dy over the rows of the block: every lane sums its rows in ascending order, then the block's
// 4 * (64 / lpr) row groups are folded through LDS in group order.
// CS: a third partial row, partial[blk][2][c] = sum of dx AS STORED over the rows of the block -- the bias gradient of the
// Linear whose output gradient dx is (x_out = x_mid + fc2(...) + b: d b = column sums of d x_out), instead of a pass of its own.
template <typename T, int VPL, int U, bool CS, bool PIPE = true>
__global__ __launch_bounds__(256) void swin_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float2* __restrict__ stats, const float* __restrict__ gamma,
                                                          const T* __restrict__ add, T* __restrict__ dx,
                                                          float* __restrict__ partial, size_t M, int C, int ld, int lpr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_smem[];
    float* red = (float*)ln_smem;  // [groups][NR][ld]
    constexpr int EPC = TT<T>::EPC;
    constexpr int NR = CS ? 3 : 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & (lpr - 1), rpw = 64 / lpr;
    const int vpr = ld / EPC;
    const float invC = 1.f / (float)C;
    float ag[VPL][EPC], ab[VPL][EPC], gm[VPL][EPC], ac[CS ? VPL : 1][EPC];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (sub + lpr * i) * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            if (CS) ac[CS ? i : 0][e] = 0.f;
            ag[i][e] = ab[i][e] = 0.f;
            gm[i][e] = (sub + lpr * i < vpr && c0 + e < C) ? gamma[c0 + e] : 0.f;
        }
    }
    const size_t stride = (size_t)gridDim.x * 4 * rpw;
    // every vector of the U row groups (dy, x and the addend) is requested before the first is used, and stays packed until its
    // group's turn: 3 x U x VPL x 4 registers in flight (the unpacked form held 2 x U x VPL x EPC and fetched the addend behind the
    // reductions) -- and, as in the forward, the NEXT iteration's request goes out before this iteration's arithmetic
    auto request = [&](size_t b, uint4 (&dq_)[U][VPL], uint4 (&xq_)[U][VPL], uint4 (&aq_)[U][VPL], float2 (&st_)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t row = b + u * stride;
            if (row < M) {
                st_[u] = stats[row];
                const uint4 *dr = (const uint4*)(dy + row * ld), *xr = (const uint4*)(x + row * ld);
#pragma unroll
                for (int i = 0; i < VPL; ++i)
                    if (sub + lpr * i < vpr) {
                        dq_[u][i] = dr[sub + lpr * i];
                        xq_[u][i] = xr[sub + lpr * i];
                        if (add) aq_[u][i] = ((const uint4*)(add + row * ld))[sub + lpr * i];
                    }
            }
        }
    };
    uint4 dq[U][VPL], xq[U][VPL], aq[U][VPL], dn[U][VPL], xn[U][VPL], an[U][VPL];
    float2 st[U], sn[U];
    size_t base = ((size_t)blockIdx.x * 4 + wave) * rpw + lane / lpr;
    request(base, dq, xq, aq, st);
    for (; base < M; base += stride * U) {
        if (PIPE) request(base + stride * U, dn, xn, an, sn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            __builtin_amdgcn_sched_barrier(0);  // one row group at a time: interleaved, the groups' unpacked values pile up in registers
            const size_t row = base + u * stride;
            if (row >= M) continue;
            float d[1][VPL][EPC], xv[1][VPL][EPC];
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) {
                    unpack16<T>(dq[u][i], d[0][i]);
                    unpack16<T>(xq[u][i], xv[0][i]);
                }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) {
                    const int c0 = (sub + lpr * i) * EPC;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const bool real = c0 + e < C;
                        const float dd = d[0][i][e];
                        const float xh = real ? (xv[0][i][e] - st[u].x) * st[u].y : 0.f;
                        const float g = dd * gm[i][e];
                        xv[0][i][e] = xh;  // (reuse: xhat)
                        d[0][i][e] = g;    // (reuse: g)
                        s1 += g;
                        s2 += g * xh;
                        ag[i][e] += dd * xh;
                        ab[i][e] += real ? dd : 0.f;
                    }
                }
            const float m1 = group_sum(s1, lpr) * invC, m2 = group_sum(s2, lpr) * invC;
            uint4* outr = (uint4*)(dx + row * ld);
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (sub + lpr * i < vpr) {
                    const int c0 = (sub + lpr * i) * EPC;
                    float o[EPC], av[EPC];
                    if (add) unpack16<T>(aq[u][i], av);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        o[e] = c0 + e < C ? st[u].y * (d[0][i][e] - m1 - xv[0][i][e] * m2) : 0.f;
                        if (add) o[e] += av[e];
                        if (CS) ac[CS ? i : 0][e] += roundT<T>(o[e]);
                    }
                    outr[sub + lpr * i] = pack16<T>(o);
                }
        }
        if (!PIPE) request(base + stride * U, dn, xn, an, sn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            st[u] = sn[u];
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                dq[u][i] = dn[u][i];
                xq[u][i] = xn[u][i];
                aq[u][i] = an[u][i];
            }
        }
    }
    // block partials: groups (wave, row group) in order
    const int grp = wave * rpw + lane / lpr, ngrp = 4 * rpw;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (sub + lpr * i < vpr) {
            const int c0 = (sub + lpr * i) * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                red[((size_t)grp * NR + 0) * ld + c0 + e] = ag[i][e];
                red[((size_t)grp * NR + 1) * ld + c0 + e] = ab[i][e];
                if (CS) red[((size_t)grp * NR + 2) * ld + c0 + e] = ac[CS ? i : 0][e];
            }
        }
    __syncthreads();
    float* outp = partial + (size_t)blockIdx.x * NR * ld;
    for (int c = threadIdx.x; c < NR * ld; c += 256) {
        float sum = red[c];
        for (int k = 1; k < ngrp; ++k) sum += red[(size_t)k * NR * ld + c];
        outp[c] = sum;
    }
}

// out[j] = sum_b partial[b][j]: a 16-lane group per column (lane k sums rows k, k+16, ... ascending, then a fixed
// butterfly), 16 columns per block -- coalesced 64-byte reads; a thread per column walking 512 rows was ~100 us of latency
// per launch, 78 launches per backward
__global__ __launch_bounds__(256) void swin_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                  int nblk, int width) {
    const int j = blockIdx.x * 16 + (threadIdx.x & 15), k = threadIdx.x >> 4;
    __shared__ float red[16][17];
    float s = 0.f;
    if (j < width) {  // eight loads in flight, added in the order b = k, k + 16, ... (a load per iteration made 32 round trips: 11 us)
        constexpr int UN = 8;
        for (int b0 = k; b0 < nblk; b0 += 16 * UN) {
            float q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) q[u] = b0 + 16 * u < nblk ? partial[(size_t)(b0 + 16 * u) * width + j] : 0.f;
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (b0 + 16 * u < nblk) s += q[u];
        }
    }
    red[k][threadIdx.x & 15] = s;
    __syncthreads();
    if (k == 0 && j < width) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q][threadIdx.x];
        out[j] = t;
    }
}

// Round 6: every fold of a backward in ONE launch.  The LayerNorm backwards and column-sum passes of a Swin backward (42 for
// Swin-T) each left `nblk` partial rows and were followed by a fold launch of their own on the branch's only chain -- but their
// results are parameter gradients nobody reads before the optimizer.  With a partial buffer per call site the folds wait until the
// end of the backward: a descriptor per job, block b serves the job whose [blk0, next blk0) holds b (16 columns per block as
// above), the arithmetic -- and therefore every bit of the result -- is swin_partial_reduce_kernel's.
struct SwinRedDesc {
    const float* partial;
    float* out;
    int nblk, width, blk0, stride;  // stride: floats between partial rows (>= width)
};
static_assert(sizeof(SwinRedDesc) == 32, "SwinRedDesc layout (mirrored by gdl/swin.py)");
__global__ __launch_bounds__(256) void swin_partial_reduce_batched_kernel(const SwinRedDesc* __restrict__ descs, int nd) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {  // last descriptor with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= (int)blockIdx.x)
            lo = mid;
        else
            hi = mid - 1;
    }
    const SwinRedDesc d = descs[lo];
    const float* __restrict__ partial = d.partial;
    const int nblk = d.nblk, width = d.width, stride = d.stride;
    const int j = ((int)blockIdx.x - d.blk0) * 16 + (threadIdx.x & 15), k = threadIdx.x >> 4;
    __shared__ float red[16][17];
    float s = 0.f;
    if (j < width) {
        constexpr int UN = 8;
        for (int b0 = k; b0 < nblk; b0 += 16 * UN) {
            float q[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) q[u] = b0 + 16 * u < nblk ? partial[(size_t)(b0 + 16 * u) * stride + j] : 0.f;
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (b0 + 16 * u < nblk) s += q[u];
        }
    }
    red[k][threadIdx.x & 15] = s;
    __syncthreads();
    if (k == 0 && j < width) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q][threadIdx.x];
        d.out[j] = t;
    }
}

// ------------------------------------------------------------------------------------------------ column sums (+ GELU')
// GELU = 1: g <- g * gelu'(u) in place first.  partial[blk][c] = sum over the block's rows, fixed order.  A thread owns one
// 16-byte column vector of one of the block's `rpb` = 256 / (vectors per row) concurrent rows (a 128-channel bf16 row is
// only 16 vectors: one row at a time would leave 240 threads idle); the row groups are folded through LDS in index order.
template <typename T, int GELU>
__global__ __launch_bounds__(256) void swin_colsum_kernel(T* __restrict__ g, const T* __restrict__ u, float* __restrict__ partial,
                                                          size_t M, int ld) {
    constexpr int EPC = TT<T>::EPC;
    __shared__ float red[256][EPC + 1];
    const int vpr = ld / EPC;
    const int rpb = vpr <= 256 ? 256 / vpr : 1;
    for (int v0 = 0; v0 < vpr; v0 += 256) {
        const int rsub = vpr <= 256 ? threadIdx.x / vpr : 0;
        const int vc = vpr <= 256 ? threadIdx.x % vpr : v0 + threadIdx.x;
        const bool live = rsub < rpb && vc < vpr;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        if (live) {
            const size_t stride = (size_t)gridDim.x * rpb;
            for (size_t base = (size_t)blockIdx.x * rpb + rsub; base < M; base += 4 * stride) {  // four rows in flight
                uint4 gv[4], uv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (base + q * stride < M) {
                        gv[q] = ((const uint4*)g)[(base + q * stride) * vpr + vc];
                        if (GELU) uv[q] = ((const uint4*)u)[(base + q * stride) * vpr + vc];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (base + q * stride < M) {  // (rows ascending: the summation order does not depend on the unroll)
                        float f[EPC];
                        unpack16<T>(gv[q], f);
                        if (GELU) {
                            float uu[EPC];
                            unpack16<T>(uv[q], uu);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) f[e] = roundT<T>(f[e] * gelu_grad<T>(uu[e]));
                            ((uint4*)g)[(base + q * stride) * vpr + vc] = pack16<T>(f);
                        }
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e] += f[e];
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[threadIdx.x][e] = acc[e];
        __syncthreads();
        if (live && rsub == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float sum = red[threadIdx.x][e];
                for (int r = 1; r < rpb; ++r) sum += red[threadIdx.x + r * vpr][e];
                partial[(size_t)blockIdx.x * ld + vc * EPC + e] = sum;
            }
        }
        if (vpr <= 256) break;
    }
}

// ------------------------------------------------------------------------------------------------ window attention
struct SwinAttnGeom {
    int H, W, ws, shift, nh, ld;  // token grid, window, roll, heads, row stride of one q / k / v segment and of `out`
    int nwin;                     // windows per image
};
// token row (within the image) of slot i of window w, and its mask region
__device__ __forceinline__ int sw_token(const SwinAttnGeom& g, int w, int i, int* region) {
    const int wpr = g.W / g.ws;
    const int R = (w / wpr) * g.ws + i / g.ws, Cc = (w % wpr) * g.ws + i % g.ws;  // rolled coordinates
    if (region) {
        const int rh = R < g.H - g.ws ? 0 : (R < g.H - g.shift ? 1 : 2), rw = Cc < g.W - g.ws ? 0 : (Cc < g.W - g.shift ? 1 : 2);
        *region = rh * 3 + rw;
    }
    int h = R + g.shift, c = Cc + g.shift;
    if (h >= g.H) h -= g.H;
    if (c >= g.W) c -= g.W;
    return h * g.W + c;
}

// forward: one wave per (image, window, head); lane i < T owns query i.  K, V of the window in LDS.
template <typename T>
__global__ __launch_bounds__(256) void swin_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table,
                                                            T* __restrict__ out, SwinAttnGeom g, int n_img) {
    __shared__ __attribute__((aligned(16))) float Ks[4][SW_MAXT][SW_PD], Vs[4][SW_MAXT][SW_PD];
    __shared__ int regs[4][SW_MAXT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Tn = g.ws * g.ws, L = g.H * g.W;
    const long unit = (long)blockIdx.x * 4 + wave, nunits = (long)n_img * g.nwin * g.nh;
    const bool live = unit < nunits;
    const int h = live ? (int)(unit % g.nh) : 0;
    const int w = live ? (int)((unit / g.nh) % g.nwin) : 0;
    const int n = live ? (int)(unit / ((long)g.nh * g.nwin)) : 0;
    int reg = 0;
    const bool act = live && lane < Tn;
    const int tok = act ? sw_token(g, w, lane, g.shift ? &reg : nullptr) : 0;
    const size_t row = (size_t)n * L + tok;
    float q[SW_HD];
    if (act) {
        const T* base = qkv + row * 3 * g.ld + h * SW_HD;
        const float scale = 0.17677669529663687f;  // 32^-0.5
#pragma unroll
        for (int d = 0; d < SW_HD; ++d) {
            q[d] = loadT(base + d) * scale;
            Ks[wave][lane][d] = loadT(base + g.ld + d);
            Vs[wave][lane][d] = loadT(base + 2 * g.ld + d);
        }
        regs[wave][lane] = reg;
    }
    __syncthreads();
    if (!act) return;
    const int ri = lane / g.ws, ci = lane % g.ws, tw = 2 * g.ws - 1;
    float s[SW_MAXT];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < SW_MAXT; ++j)
        if (j < Tn) {
            float a = 0.f;
            const float4* kr = (const float4*)Ks[wave][j];
#pragma unroll
            for (int d4 = 0; d4 < SW_HD / 4; ++d4) {
                const float4 kv = kr[d4];
                a += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
            }
            const int rj = j / g.ws, cj = j % g.ws;
            a += table[((ri - rj + g.ws - 1) * tw + (ci - cj + g.ws - 1)) * g.nh + h];
            if (g.shift && regs[wave][j] != reg) a -= 100.f;
            s[j] = a;
            mx = fmaxf(mx, a);
        }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < SW_MAXT; ++j)
        if (j < Tn) {
            s[j] = __expf(s[j] - mx);
            den += s[j];
        }
    const float inv = 1.f / den;
    float o[SW_HD];
#pragma unroll
    for (int d = 0; d < SW_HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < SW_MAXT; ++j)
        if (j < Tn) {
            const float p = s[j] * inv;
            const float4* vr = (const float4*)Vs[wave][j];
#pragma unroll
            for (int d4 = 0; d4 < SW_HD / 4; ++d4) {
                const float4 vv = vr[d4];
                o[4 * d4] += p * vv.x, o[4 * d4 + 1] += p * vv.y, o[4 * d4 + 2] += p * vv.z, o[4 * d4 + 3] += p * vv.w;
            }
        }
    T* ob = out + row * g.ld + h * SW_HD;
#pragma unroll
    for (int d = 0; d < SW_HD; ++d) storeT(ob + d, o[d]);
    if (h == 0)  // the row's padding columns (nh*32 .. ld) belong to nobody: keep them zero
        for (int c = g.nh * SW_HD; c < g.ld; ++c) storeT(out + row * g.ld + c, 0.f);
}

// (The round-2/3 matrix-core attention kernels -- one wave per (image, window, head), any window up to 7 x 7 -- were removed in
// round 5: csrc/swin_attn7.hip serves every 7 x 7 bf16 case; tools/experiments/swin_attn_mfma_r2.hip.txt keeps the text.)

// 32 consecutive channels of a token row -> LDS row (float)
template <typename T>
__device__ __forceinline__ void sw_row_to_lds(const T* __restrict__ src, float* dst) {
    constexpr int EPC = TT<T>::EPC;
#pragma unroll
    for (int v = 0; v < SW_HD / EPC; ++v) {
        float f[EPC];
        unpack16<T>(((const uint4*)src)[v], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) dst[v * EPC + e] = f[e];
    }
}
template <typename T>
__device__ __forceinline__ void sw_row_store(T* dst, const float* f, float mul) {
    constexpr int EPC = TT<T>::EPC;
#pragma unroll
    for (int v = 0; v < SW_HD / EPC; ++v) {
        float t[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) t[e] = f[v * EPC + e] * mul;
        ((uint4*)dst)[v] = pack16<T>(t);
    }
}

// backward: one wave (block of 64) per (image, group of G windows, head).  Two passes per window, nothing of size T x T in
// LDS but the accumulated d(bias): pass 1, lane = query i: the probabilities of row i are recomputed in registers, dP_ij =
// dO_i . V_j, pd_i = sum_j P_ij dP_ij, dQ_i; the row's softmax statistics and pd_i go to LDS.  Pass 2, lane = key j:
// column j is recomputed from the statistics -- P_ij, dS_ij -- giving dV_j, dK_j, and dS_ij is added to the block's
// [T][T] accumulator (column j is this lane's).  At the end the accumulator is folded into the (2ws-1)^2 entries of the
// block's partial of d(table): tpart[block][r].
template <typename T>
__global__ __launch_bounds__(64) void swin_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table,
                                                           const T* __restrict__ dout, T* __restrict__ dqkv,
                                                           float* __restrict__ tpart, SwinAttnGeom g, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    constexpr int PD = SW_PD, PT = SW_MAXT + 1;
    float(*Qs)[PD] = (float(*)[PD])sw_smem;  // scaled queries
    float(*Ks)[PD] = Qs + SW_MAXT;
    float(*Vs)[PD] = Ks + SW_MAXT;
    float(*Os)[PD] = Vs + SW_MAXT;  // d(out)
    float(*Da)[PT] = (float(*)[PT])(Os + SW_MAXT);
    float* rmax = (float*)(Da + SW_MAXT);
    float* rinv = rmax + SW_MAXT;
    float* rpd = rinv + SW_MAXT;
    float* tab = rpd + SW_MAXT;  // this head's column of the bias table, (2ws-1)^2 entries
    int* regs = (int*)(tab + (2 * 7 - 1) * (2 * 7 - 1));
    const int lane = threadIdx.x;
    const int Tn = g.ws * g.ws, L = g.H * g.W, tw = 2 * g.ws - 1;
    const int ngrp = (g.nwin + G - 1) / G;
    const int h = blockIdx.x % g.nh, grp = (blockIdx.x / g.nh) % ngrp, n = blockIdx.x / (g.nh * ngrp);
    const float scale = 0.17677669529663687f;
    const bool act = lane < Tn;
    const int ri = lane / g.ws, ci = lane % g.ws;
    for (int r = lane; r < tw * tw; r += 64) tab[r] = table[r * g.nh + h];
    if (act)
        for (int j = 0; j < Tn; ++j) Da[j][lane] = 0.f;
    for (int w = grp * G; w < min(g.nwin, grp * G + G); ++w) {
        int reg = 0;
        const int tok = act ? sw_token(g, w, lane, g.shift ? &reg : nullptr) : 0;
        const size_t row = (size_t)n * L + tok;
        __syncthreads();  // (everybody is done with the previous window's rows)
        if (act) {
            const T* base = qkv + row * 3 * g.ld + h * SW_HD;
            sw_row_to_lds<T>(base, Qs[lane]);
            sw_row_to_lds<T>(base + g.ld, Ks[lane]);
            sw_row_to_lds<T>(base + 2 * g.ld, Vs[lane]);
            sw_row_to_lds<T>(dout + row * g.ld + h * SW_HD, Os[lane]);
#pragma unroll
            for (int d = 0; d < SW_HD; ++d) Qs[lane][d] *= scale;
            regs[lane] = reg;
        }
        __syncthreads();
        if (act) {  // ---- pass 1: lane = query i
            float q[SW_HD], go[SW_HD];
#pragma unroll
            for (int d = 0; d < SW_HD; ++d) {
                q[d] = Qs[lane][d];
                go[d] = Os[lane][d];
            }
            float s[SW_MAXT], dp[SW_MAXT];
            float mx = -3.0e38f;
            int rj = 0, cj = 0;
#pragma unroll
            for (int j = 0; j < SW_MAXT; ++j)
                if (j < Tn) {
                    float a = 0.f, b = 0.f;
                    const float4 *kr = (const float4*)Ks[j], *vr = (const float4*)Vs[j];
#pragma unroll
                    for (int d4 = 0; d4 < SW_HD / 4; ++d4) {
                        const float4 kv = kr[d4], vv = vr[d4];
                        a += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
                        b += go[4 * d4] * vv.x + go[4 * d4 + 1] * vv.y + go[4 * d4 + 2] * vv.z + go[4 * d4 + 3] * vv.w;
                    }
                    a += tab[(ri - rj + g.ws - 1) * tw + (ci - cj + g.ws - 1)];
                    if (g.shift && regs[j] != reg) a -= 100.f;
                    s[j] = a;
                    dp[j] = b;
                    mx = fmaxf(mx, a);
                    if (++cj == g.ws) cj = 0, ++rj;
                }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < SW_MAXT; ++j)
                if (j < Tn) {
                    s[j] = __expf(s[j] - mx);
                    den += s[j];
                }
            const float inv = 1.f / den;
            float pd = 0.f;
#pragma unroll
            for (int j = 0; j < SW_MAXT; ++j)
                if (j < Tn) pd += s[j] * inv * dp[j];
            float dq[SW_HD];
#pragma unroll
            for (int d = 0; d < SW_HD; ++d) dq[d] = 0.f;
#pragma unroll
            for (int j = 0; j < SW_MAXT; ++j)
                if (j < Tn) {
                    const float ds = s[j] * inv * (dp[j] - pd);
                    const float4* kr = (const float4*)Ks[j];
#pragma unroll
                    for (int d4 = 0; d4 < SW_HD / 4; ++d4) {
                        const float4 kv = kr[d4];
                        dq[4 * d4] += ds * kv.x, dq[4 * d4 + 1] += ds * kv.y, dq[4 * d4 + 2] += ds * kv.z, dq[4 * d4 + 3] += ds * kv.w;
                    }
                }
            rmax[lane] = mx;
            rinv[lane] = inv;
            rpd[lane] = pd;
            sw_row_store<T>(dqkv + row * 3 * g.ld + h * SW_HD, dq, scale);
        }
        __syncthreads();
        if (act) {  // ---- pass 2: lane = key / value j; queries i ascending
            float k[SW_HD], v[SW_HD], dk[SW_HD], dv[SW_HD];
#pragma unroll
            for (int d = 0; d < SW_HD; ++d) {
                k[d] = Ks[lane][d];
                v[d] = Vs[lane][d];
                dk[d] = dv[d] = 0.f;
            }
            int rq = 0, cq = 0;
            for (int i = 0; i < Tn; ++i) {
                float a = 0.f, b = 0.f;
                const float4 *qr = (const float4*)Qs[i], *orow = (const float4*)Os[i];
                float4 qv[SW_HD / 4], ov[SW_HD / 4];
#pragma unroll
                for (int d4 = 0; d4 < SW_HD / 4; ++d4) {
                    qv[d4] = qr[d4];
                    ov[d4] = orow[d4];
                    a += qv[d4].x * k[4 * d4] + qv[d4].y * k[4 * d4 + 1] + qv[d4].z * k[4 * d4 + 2] + qv[d4].w * k[4 * d4 + 3];
                    b += ov[d4].x * v[4 * d4] + ov[d4].y * v[4 * d4 + 1] + ov[d4].z * v[4 * d4 + 2] + ov[d4].w * v[4 * d4 + 3];
                }
                a += tab[(rq - ri + g.ws - 1) * tw + (cq - ci + g.ws - 1)];
                if (g.shift && regs[i] != reg) a -= 100.f;
                const float p = __expf(a - rmax[i]) * rinv[i];
                const float ds = p * (b - rpd[i]);
                Da[i][lane] += ds;
#pragma unroll
                for (int d4 = 0; d4 < SW_HD / 4; ++d4) {  // (Qs carries the scale)
                    dv[4 * d4] += p * ov[d4].x, dv[4 * d4 + 1] += p * ov[d4].y, dv[4 * d4 + 2] += p * ov[d4].z, dv[4 * d4 + 3] += p * ov[d4].w;
                    dk[4 * d4] += ds * qv[d4].x, dk[4 * d4 + 1] += ds * qv[d4].y, dk[4 * d4 + 2] += ds * qv[d4].z, dk[4 * d4 + 3] += ds * qv[d4].w;
                }
                if (++cq == g.ws) cq = 0, ++rq;
            }
            T* ob = dqkv + row * 3 * g.ld + h * SW_HD;
            sw_row_store<T>(ob + g.ld, dk, 1.f);
            sw_row_store<T>(ob + 2 * g.ld, dv, 1.f);
            if (h == 0)  // padding columns of the three segments stay zero
                for (int sgm = 0; sgm < 3; ++sgm)
                    for (int c = g.nh * SW_HD; c < g.ld; ++c) storeT(dqkv + row * 3 * g.ld + sgm * g.ld + c, 0.f);
        }
    }
    __syncthreads();
    // d(table)[r] of this block: the pairs (i, j) with i - j = (dh, dw), j ascending
    float* tp = tpart + (size_t)blockIdx.x * tw * tw;
    for (int r = lane; r < tw * tw; r += 64) {
        const int dh = r / tw - (g.ws - 1), dw = r % tw - (g.ws - 1);
        float a = 0.f;
        for (int rj = 0; rj < g.ws; ++rj) {
            const int rr = rj + dh;
            if (rr < 0 || rr >= g.ws) continue;
            for (int cj = 0; cj < g.ws; ++cj) {
                const int cc = cj + dw;
                if (cc < 0 || cc >= g.ws) continue;
                a += Da[rr * g.ws + cc][rj * g.ws + cj];
            }
        }
        tp[r] = a;
    }
}

// dtable[r][h] = sum_u tpart[u][h][r], u = (image, window group) ascending: one wave per entry, 64 partial sums in
// lane order + a fixed butterfly
__global__ __launch_bounds__(256) void swin_table_reduce_kernel(const float* __restrict__ tpart, float* __restrict__ dtable,
                                                                int nu, int nh, int tt) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nh * tt) return;
    const int h = i / tt, r = i % tt;
    float s = 0.f;
    for (int u = lane; u < nu; u += 64) s += tpart[((size_t)u * nh + h) * tt + r];
    s = wave_sum(s);
    if (lane == 0) dtable[r * nh + h] = s;
}

// ------------------------------------------------------------------------------------------------ patch merging
// fwd: cat[n][h2][w2][q*C + c] = x[n][2 h2 + (q & 1)][2 w2 + (q >> 1)][c]   (q = 0..3: (0,0), (1,0), (0,1), (1,1));
// bwd (SCATTER = 1): the adjoint, padding columns of dx zeroed
template <typename T, int SCATTER>
__global__ __launch_bounds__(256) void swin_merge_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W,
                                                         int C, int ldx) {
    const int H2 = H / 2, W2 = W / 2;
    const size_t total = (size_t)N * H * W * ldx;  // iterate over the un-merged tensor's elements (padding included)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % ldx);
        size_t r = i / ldx;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), n = (int)(r / H);
        if (c >= C) {
            if (SCATTER) storeT(dst + i, 0.f);
            continue;
        }
        const int q = (h & 1) + 2 * (w & 1);
        const size_t m = (((size_t)n * H2 + (h >> 1)) * W2 + (w >> 1)) * (4 * C) + q * C + c;
        if (SCATTER)
            dst[i] = src[m];
        else
            dst[m] = src[i];
    }
}

// the same by 16-byte vectors (C a multiple of the vector's element count: every Swin width): a thread moves one vector of the
// un-merged tensor, row / column from 32-bit divisions of the pixel index -- the element-wise form ran at 1.45 TB/s
template <typename T, int SCATTER>
__global__ __launch_bounds__(256) void swin_merge_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W,
                                                             int C, int ldx) {
    constexpr int EPC = TT<T>::EPC;
    const int H2 = H / 2, W2 = W / 2, vpr = ldx / EPC;
    const size_t total = (size_t)N * H * W * vpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned pix = (unsigned)(i / vpr);
        const int c0 = (int)(i - (size_t)pix * vpr) * EPC;
        if (c0 >= C) {
            if (SCATTER) ((uint4*)dst)[i] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        const unsigned row = pix / (unsigned)W, w = pix - row * (unsigned)W;
        const unsigned n = row / (unsigned)H, h = row - n * (unsigned)H;
        const int q = (int)(h & 1u) + 2 * (int)(w & 1u);
        const size_t m = (((size_t)n * H2 + (h >> 1)) * W2 + (w >> 1)) * (4 * C) + q * C + c0;
        if (SCATTER)
            ((uint4*)dst)[i] = *(const uint4*)(src + m);
        else
            *(uint4*)(dst + m) = ((const uint4*)src)[i];
    }
}

// ------------------------------------------------------------------------------------------------ token mean
template <typename T>
__global__ __launch_bounds__(256) void swin_token_mean_kernel(const T* __restrict__ x, float* __restrict__ y, int L, int C, int ld) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += loadT(x + ((size_t)n * L + l) * ld + c);
        y[(size_t)n * C + c] = s / (float)L;
    }
}
// The same mean with 16-byte loads and the tokens spread over the block (round 5: the per-channel walk above is L dependent
// 2-byte loads per thread on 64 blocks -- 137 us at 64 x 147 x 768, on the Swin chain between its forward and its backward;
// this form: 5 us).
// grid = (ceil(C / (8 EPC)), N): thread (tl = tid >> 3, ch = tid & 7) adds tokens tl, tl + 32, ... of its 16-byte chunk in
// ascending order, then the 32 token lanes are folded in ascending order through LDS: a fixed order, not the walk's.
template <typename T>
__global__ __launch_bounds__(256) void swin_token_mean_vec_kernel(const T* __restrict__ x, float* __restrict__ y, int L, int C, int ld) {
    constexpr int EPC = TT<T>::EPC;
    __shared__ float red[32][8 * EPC + 1];
    const int n = blockIdx.y, ch = threadIdx.x & 7, tl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + ch) * EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        const T* p = x + (size_t)n * L * ld + c0;
        for (int l = tl; l < L; l += 32) {
            float f[EPC];
            unpack16<T>(*(const uint4*)(p + (size_t)l * ld), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tl][ch * EPC + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 8 * EPC + threadIdx.x;
    if (threadIdx.x < 8 * EPC && c < C) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 32; ++t) s += red[t][threadIdx.x];
        y[(size_t)n * C + c] = s / (float)L;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void swin_token_mean_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int N, int L,
                                                                  int C, int ld) {
    const size_t total = (size_t)N * L * ld;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % ld);
        const size_t n = i / ((size_t)L * ld);
        storeT(dx + i, c < C ? dy[n * C + c] / (float)L : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ parameter layouts
// A matrix [n][k] whose rows come in segments of nseg real rows stored at a pitch of nseg_pad (QKV: 96 -> 128 per
// segment), columns likewise (kseg / kseg_pad).  PACK: src float32 real [n][k] -> dst T padded [np][kp] (+ dstT T
// [kp][np] when given).  UNPACK (T = float): src padded float32 -> dst real float32.
struct SwinSeg {
    int n, k, nseg, nseg_pad, kseg, kseg_pad, np, kp;
};
template <typename T>
__global__ __launch_bounds__(256) void swin_pack_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT,
                                                        SwinSeg s) {
    const size_t total = (size_t)s.np * s.kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pk = (int)(i % s.kp), pn = (int)(i / s.kp);
        const int sn = pn / s.nseg_pad, on = pn % s.nseg_pad, sk = pk / s.kseg_pad, ok = pk % s.kseg_pad;
        const int rn = sn * s.nseg + on, rk = sk * s.kseg + ok;
        const float v = (on < s.nseg && ok < s.kseg && rn < s.n && rk < s.k) ? src[(size_t)rn * s.k + rk] : 0.f;
        storeT(dst + i, v);
        if (dstT) storeT(dstT + (size_t)pk * s.np + pn, v);
    }
}
__global__ __launch_bounds__(256) void swin_unpack_kernel(const float* __restrict__ src, float* __restrict__ dst, SwinSeg s) {
    const size_t total = (size_t)s.n * s.k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int rk = (int)(i % s.k), rn = (int)(i / s.k);
        const int pn = (rn / s.nseg) * s.nseg_pad + rn % s.nseg, pk = (rk / s.kseg) * s.kseg_pad + rk % s.kseg;
        dst[i] = src[(size_t)pn * s.kp + pk];
    }
}

// All of a step's layout conversions in ONE launch each way (Swin-T has 171 parameter tensors: one launch per tensor was
// 340 tiny kernels per step, ~8 ms of launch gaps).  A descriptor per tensor; block b serves the descriptor whose
// [blk0, next blk0) holds b, 1024 elements per block.  `dir` 0: pack (src float32 real -> dst `dt` padded, dstT optional),
// 1: unpack (src float32 padded -> dst float32 real).
struct SwinPackDesc {
    const float* src;
    void* dst;
    void* dstT;
    SwinSeg s;
    int dt;
    int blk0;
};
static_assert(sizeof(SwinPackDesc) == 64, "SwinPackDesc layout (mirrored by gdl/swin.py)");
__global__ __launch_bounds__(256) void swin_pack_batched_kernel(const SwinPackDesc* __restrict__ descs, int nd, int dir) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {  // last descriptor with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= (int)blockIdx.x)
            lo = mid;
        else
            hi = mid - 1;
    }
    const SwinPackDesc d = descs[lo];
    const SwinSeg s = d.s;
    const int blk = (int)blockIdx.x - d.blk0;
    // (32-bit index arithmetic throughout: a matrix has fewer than 2^31 elements; the 64-bit divisions of the first form were
    // half of its time, the 2-byte scattered stores of the transposed copy the other half: 287 us per step for Swin-T)
    auto real = [&](int pn, int pk) -> float {
        const int sn = pn / s.nseg_pad, on = pn - sn * s.nseg_pad, sk = pk / s.kseg_pad, ok = pk - sk * s.kseg_pad;
        const int rn = sn * s.nseg + on, rk = sk * s.kseg + ok;
        return (on < s.nseg && ok < s.kseg && rn < s.n && rk < s.k) ? d.src[(size_t)rn * s.k + rk] : 0.f;
    };
    if (dir == 0 && (s.np & 31) == 0 && (s.kp & 31) == 0) {
        // a 32 x 32 tile of the padded matrix per block (1024 elements, like the linear form: the caller's block count holds):
        // rows of 32 elements in, rows of 32 elements out for both copies, the transposed one through LDS
        __shared__ float tile[32][33];
        const int tpr = s.kp >> 5, tn = blk / tpr, tk = blk - tn * tpr;
        const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pn = 32 * tn + r + 8 * u, pk = 32 * tk + c;
            const float v = real(pn, pk);
            tile[r + 8 * u][c] = v;
            if (d.dt == GDL_F32)
                ((float*)d.dst)[(size_t)pn * s.kp + pk] = v;
            else
                storeT((bf16*)d.dst + (size_t)pn * s.kp + pk, v);
        }
        if (!d.dstT) return;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pk = 32 * tk + r + 8 * u, pn = 32 * tn + c;
            const float v = tile[c][r + 8 * u];
            if (d.dt == GDL_F32)
                ((float*)d.dstT)[(size_t)pk * s.np + pn] = v;
            else
                storeT((bf16*)d.dstT + (size_t)pk * s.np + pn, v);
        }
        return;
    }
    const int total = dir == 0 ? s.np * s.kp : s.n * s.k;
    for (int u = 0; u < 4; ++u) {
        const int i = (blk * 4 + u) * 256 + (int)threadIdx.x;
        if (i >= total) return;
        if (dir == 0) {
            const int pn = i / s.kp, pk = i - pn * s.kp;
            const float v = real(pn, pk);
            if (d.dt == GDL_F32) {
                ((float*)d.dst)[i] = v;
                if (d.dstT) ((float*)d.dstT)[(size_t)pk * s.np + pn] = v;
            } else {
                storeT((bf16*)d.dst + i, v);
                if (d.dstT) storeT((bf16*)d.dstT + (size_t)pk * s.np + pn, v);
            }
        } else {
            const int rn = i / s.k, rk = i - rn * s.k;
            const int pn = (rn / s.nseg) * s.nseg_pad + rn % s.nseg, pk = (rk / s.kseg) * s.kseg_pad + rk % s.kseg;
            ((float*)d.dst)[i] = d.src[(size_t)pn * s.kp + pk];
        }
    }
}

// ================================================================================================ host side
static int sw_grid(size_t work, int per_block = 256, int cap = 256 * 16) {
    const size_t b = (work + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? (size_t)cap : b));
}
#define SW_DISPATCH(dt, CALL_F32, CALL_BF16) \
    do {                                      \
        if ((dt) == GDL_F32) {                \
            CALL_F32;                         \
        } else {                              \
            CALL_BF16;                        \
        }                                     \
    } while (0)

int swin_patch_gather(int dt, const float* x, void* a, int B, int T, int H, int W, int p, hipStream_t st) {
    GDL_REQUIRE(3 * p * p <= 64 && H % p == 0 && W % p == 0, "swin_patch_gather: patch %d on %dx%d", p, H, W);
    const size_t total = (size_t)B * T * (H / p) * (W / p) * 64;
    ProfScope prof("gdl::swin_patch_gather_kernel", PROF_HBM, st, (double)total * (dt == GDL_F32 ? 4 : 2) + (double)B * T * 3 * H * W * 4);
    if (dt == GDL_BF16 && p == 4 && W % 4 == 0 && ((uintptr_t)x & 15) == 0) {
        hipLaunchKernelGGL(swin_patch_gather4_kernel, dim3(sw_grid(total / 8)), dim3(256), 0, st, x, (bf16*)a, B, T, H, W);
        GDL_CHECK_LAUNCH("swin_patch_gather4_kernel");
        return GDL_OK;
    }
    SW_DISPATCH(dt, hipLaunchKernelGGL(swin_patch_gather_kernel<float>, dim3(sw_grid(total)), dim3(256), 0, st, x, (float*)a, B, T, H, W, p),
                hipLaunchKernelGGL(swin_patch_gather_kernel<bf16>, dim3(sw_grid(total)), dim3(256), 0, st, x, (bf16*)a, B, T, H, W, p));
    GDL_CHECK_LAUNCH("swin_patch_gather_kernel");
    return GDL_OK;
}

template <typename T>
static int bias_act_t(T* y, const float* bias, T* u, const T* res, size_t M, int ld, int mode, hipStream_t st) {
    const int g = sw_grid(M * (ld / TT<T>::EPC));
    if (mode == 0)
        hipLaunchKernelGGL((swin_bias_act_kernel<T, 0>), dim3(g), dim3(256), 0, st, y, bias, u, res, M, ld);
    else if (mode == 1)
        hipLaunchKernelGGL((swin_bias_act_kernel<T, 1>), dim3(g), dim3(256), 0, st, y, bias, u, res, M, ld);
    else
        hipLaunchKernelGGL((swin_bias_act_kernel<T, 2>), dim3(g), dim3(256), 0, st, y, bias, u, res, M, ld);
    GDL_CHECK_LAUNCH("swin_bias_act_kernel");
    return GDL_OK;
}
int swin_bias_act(int dt, void* y, const float* bias, void* u, const void* res, size_t M, int ld, int mode, hipStream_t st) {
    GDL_REQUIRE(ld % 64 == 0 && mode >= 0 && mode <= 2 && (mode != 1 || u) && (mode != 2 || res), "swin_bias_act: bad arguments");
    const double esz = dt == GDL_F32 ? 4 : 2;
    ProfScope prof("gdl::swin_bias_act_kernel", PROF_HBM, st, (double)M * ld * esz * (mode == 0 ? 2 : 3));
    if (dt == GDL_F32) return bias_act_t<float>((float*)y, bias, (float*)u, (const float*)res, M, ld, mode, st);
    return bias_act_t<bf16>((bf16*)y, bias, (bf16*)u, (const bf16*)res, M, ld, mode, st);
}

int swin_drop_path(int dt, const void* y, const void* res, const float* scale, void* out, size_t M, int L, int ld, hipStream_t st) {
    GDL_REQUIRE(y && scale && out && ld % 64 == 0 && L >= 1 && M % (size_t)L == 0, "swin_drop_path: bad arguments (M=%zu, L=%d, ld=%d)", M, L, ld);
    const double esz = dt == GDL_F32 ? 4 : 2;
    ProfScope prof("gdl::swin_drop_path_kernel", PROF_HBM, st, (double)M * ld * esz * (res ? 3 : 2));
    if (dt == GDL_F32) {
        const int g = sw_grid(M * (ld / 4));
        hipLaunchKernelGGL(swin_drop_path_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)y, (const float*)res, scale, (float*)out, M, L, ld);
    } else {
        const int g = sw_grid(M * (ld / 8));
        hipLaunchKernelGGL(swin_drop_path_kernel<bf16>, dim3(g), dim3(256), 0, st, (const bf16*)y, (const bf16*)res, scale, (bf16*)out, M, L, ld);
    }
    GDL_CHECK_LAUNCH("swin_drop_path_kernel");
    return GDL_OK;
}

// lanes per row of the LayerNorm kernels: the power of two >= the row's 16-byte vectors, at most 64
static int ln_lpr(int dt, int ld) {
    const int vpr = ld / (dt == GDL_F32 ? 4 : 8);
    int l = 16;
    while (l < vpr && l < 64) l <<= 1;
    return l;
}
int swin_ln_fwd(int dt, const void* x, const float* gamma, const float* beta, void* y, float* stats, size_t M, int C, int ld,
                hipStream_t st) {
    GDL_REQUIRE(ld % 64 == 0 && ld <= 64 * SW_MAXV && C <= ld, "swin_ln_fwd: width %d / %d", C, ld);
    const int lpr = ln_lpr(dt, ld), vpl = (ld / (dt == GDL_F32 ? 4 : 8) + lpr - 1) / lpr;
    static int u2 = -1;
    if (u2 < 0) {
        const char* e = tune_env("GDL_SW_LN_U2");  // tuning aid: twice the row groups in flight per wave
        u2 = e ? atoi(e) : 0;
    }
    const int uu = (vpl == 1 ? 4 : (vpl == 2 ? 2 : 1)) * (u2 && vpl <= 3 ? 2 : 1);
    static int fcap = -1;
    if (fcap < 0) {
        const char* e = tune_env("GDL_SW_LN_CAP");  // tuning aid: block cap of the LayerNorm forward
        fcap = e ? atoi(e) : 768;  // (one round of resident blocks at 112-135 registers; 8192 -> 1024 -> 768: tools/bench_swin_ln.py)
    }
    const int g = sw_grid(M, 4 * (64 / lpr) * uu, fcap);
    ProfScope prof("gdl::swin_ln_fwd_kernel", PROF_HBM, st, (double)M * ld * (dt == GDL_F32 ? 8 : 4));
#define SW_LN_FWD(T, V, U_) \
    hipLaunchKernelGGL((swin_ln_fwd_kernel<T, V, U_>), dim3(g), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, (float2*)stats, M, C, ld, lpr)
    if (dt == GDL_F32) {
        if (vpl == 1) SW_LN_FWD(float, 1, 4); else if (vpl == 2) SW_LN_FWD(float, 2, 2); else if (vpl == 3) SW_LN_FWD(float, 3, 1); else SW_LN_FWD(float, 6, 1);
    } else {
        if (u2) {
            if (vpl == 1) SW_LN_FWD(bf16, 1, 8); else if (vpl == 2) SW_LN_FWD(bf16, 2, 4); else SW_LN_FWD(bf16, 3, 2);
        } else {
            if (vpl == 1) SW_LN_FWD(bf16, 1, 4); else if (vpl == 2) SW_LN_FWD(bf16, 2, 2); else SW_LN_FWD(bf16, 3, 1);
        }
    }
#undef SW_LN_FWD
    GDL_CHECK_LAUNCH("swin_ln_fwd_kernel");
    return GDL_OK;
}

constexpr int SW_PARTIAL_BLOCKS = 1024;  // most blocks (= partial rows) of the LayerNorm backward and column-sum kernels
size_t swin_partial_bytes(int ld) { return (size_t)SW_PARTIAL_BLOCKS * 2 * ld * sizeof(float); }
static int sw_pblocks() {
    static int v = -1;
    if (v < 0) {
        const char* e = tune_env("GDL_SW_PBLOCKS");  // tuning aid: blocks of the LayerNorm backward / column sums (<= 1024)
        v = e ? atoi(e) : 512;
        if (v < 1 || v > SW_PARTIAL_BLOCKS) v = 512;
    }
    return v;
}

static int partial_reduce(const float* partial, float* out, int nblk, int width, hipStream_t st) {
    hipLaunchKernelGGL(swin_partial_reduce_kernel, dim3((width + 15) / 16), dim3(256), 0, st, partial, out, nblk, width);
    GDL_CHECK_LAUNCH("swin_partial_reduce_kernel");
    return GDL_OK;
}

// partial rows (= blocks) a LayerNorm backward over M rows of width ld leaves
int swin_ln_bwd_rows(int dt, size_t M, int ld) {
    const int lpr = ln_lpr(dt, ld), rpw = 64 / lpr, vpl = (ld / (dt == GDL_F32 ? 4 : 8) + lpr - 1) / lpr;
    const int uu = vpl == 1 ? 4 : (vpl == 2 ? 2 : 1);
    const size_t gg = (M + (size_t)4 * rpw * uu - 1) / ((size_t)4 * rpw * uu);
    const size_t cap = tune_env("GDL_SW_PBLOCKS") ? (size_t)sw_pblocks() : 512;
    return (int)(gg > cap ? cap : gg);
}

// dgamma_dbeta: [2][ld] (padded layout); colsum: [3][ld], row 2 = column sums of dx as stored.  dgamma_dbeta == nullptr: the
// partial rows (swin_ln_bwd_rows of them, 2 or 3 * ld wide) stay in `partial` for swin_partial_reduce_batched
int swin_ln_bwd(int dt, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
                float* dgamma_dbeta, float* partial, size_t M, int C, int ld, hipStream_t st, bool colsum) {
    GDL_REQUIRE(ld % 64 == 0 && ld <= 64 * SW_MAXV && C <= ld && partial, "swin_ln_bwd: width %d / %d", C, ld);
    const int lpr = ln_lpr(dt, ld), rpw = 64 / lpr, vpl = (ld / (dt == GDL_F32 ? 4 : 8) + lpr - 1) / lpr;
    // large launches: the pipelined form (next iteration's rows requested before this one's arithmetic; ~250 registers: two blocks
    // per CU, 512 blocks); the 9 408-row shapes of stage 4 lose with it and keep the plain form (tools/bench_swin_ln.py:
    // stage 3 44.6 -> 40.0 us, first merge 124 -> 111, stage 4 27.6 -> 30.0)
    const bool pipe = M >= 30000;
    const int g = swin_ln_bwd_rows(dt, M, ld);
    const int nr = colsum ? 3 : 2;
    const size_t lds = (size_t)4 * rpw * nr * ld * sizeof(float);
    GDL_REQUIRE(lds <= 64 * 1024, "swin_ln_bwd: width %d needs %zu bytes of LDS", ld, lds);
    {
        ProfScope prof("gdl::swin_ln_bwd_kernel", PROF_HBM, st, (double)M * ld * (dt == GDL_F32 ? 4 : 2) * (add ? 4 : 3));
#define SW_LN_BWD1(T, V, U_, CS_, P_)                                                                                                  \
    do {                                                                                                                            \
        static DevOnce attr;                                                                                                   \
        if (!attr) {                                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)swin_ln_bwd_kernel<T, V, U_, CS_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
            if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(swin_ln_bwd)");                                          \
            attr = true;                                                                                                            \
        }                                                                                                                           \
        hipLaunchKernelGGL((swin_ln_bwd_kernel<T, V, U_, CS_, P_>), dim3(g), dim3(256), lds, st, (const T*)dy, (const T*)x, (const float2*)stats, gamma, \
                           (const T*)add, (T*)dx, partial, M, C, ld, lpr);                                                          \
    } while (0)
#define SW_LN_BWD(T, V, U_)                \
    do {                                   \
        if (colsum && pipe)                       \
            SW_LN_BWD1(T, V, U_, true, true);     \
        else if (colsum)                          \
            SW_LN_BWD1(T, V, U_, true, false);    \
        else if (pipe)                            \
            SW_LN_BWD1(T, V, U_, false, true);    \
        else                                      \
            SW_LN_BWD1(T, V, U_, false, false);   \
    } while (0)
        if (dt == GDL_F32) {
            if (vpl == 1) SW_LN_BWD(float, 1, 4); else if (vpl == 2) SW_LN_BWD(float, 2, 2); else if (vpl == 3) SW_LN_BWD(float, 3, 1); else SW_LN_BWD(float, 6, 1);
        } else {
            if (vpl == 1) SW_LN_BWD(bf16, 1, 4); else if (vpl == 2) SW_LN_BWD(bf16, 2, 2); else SW_LN_BWD(bf16, 3, 1);
        }
#undef SW_LN_BWD
#undef SW_LN_BWD1
        GDL_CHECK_LAUNCH("swin_ln_bwd_kernel");
    }
    if (!dgamma_dbeta) return GDL_OK;
    return partial_reduce(partial, dgamma_dbeta, g, nr * ld, st);
}

int swin_colsum_rows(int dt, size_t M, int ld) {
    const int vpr = ld / (dt == GDL_F32 ? 4 : 8), rpb = vpr <= 256 ? 256 / vpr : 1;
    const size_t passes = (M + rpb - 1) / rpb;
    return (int)(passes < (size_t)sw_pblocks() ? passes : (size_t)sw_pblocks());
}

// db[ld] = column sums of g; gelu != 0: g <- g * gelu'(u) first.  db == nullptr: the partial rows (swin_colsum_rows of them, ld
// wide) stay in `partial` for swin_partial_reduce_batched
int swin_colsum(int dt, void* g, const void* u, float* db, float* partial, size_t M, int ld, hipStream_t st) {
    GDL_REQUIRE(ld % 64 == 0 && partial, "swin_colsum: bad arguments");
    const int nb = swin_colsum_rows(dt, M, ld);
    {
        ProfScope prof("gdl::swin_colsum_kernel", PROF_HBM, st, (double)M * ld * (dt == GDL_F32 ? 4 : 2) * (u ? 3 : 1));
        if (dt == GDL_F32) {
            if (u)
                hipLaunchKernelGGL((swin_colsum_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (float*)g, (const float*)u, partial, M, ld);
            else
                hipLaunchKernelGGL((swin_colsum_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (float*)g, (const float*)u, partial, M, ld);
        } else {
            if (u)
                hipLaunchKernelGGL((swin_colsum_kernel<bf16, 1>), dim3(nb), dim3(256), 0, st, (bf16*)g, (const bf16*)u, partial, M, ld);
            else
                hipLaunchKernelGGL((swin_colsum_kernel<bf16, 0>), dim3(nb), dim3(256), 0, st, (bf16*)g, (const bf16*)u, partial, M, ld);
        }
        GDL_CHECK_LAUNCH("swin_colsum_kernel");
    }
    if (!db) return GDL_OK;
    return partial_reduce(partial, db, nb, ld, st);
}

int swin_partial_reduce_batched(const void* descs, int nd, int total_blocks, hipStream_t st) {
    GDL_REQUIRE(descs && nd > 0 && total_blocks > 0, "swin_partial_reduce_batched: bad arguments");
    hipLaunchKernelGGL(swin_partial_reduce_batched_kernel, dim3(total_blocks), dim3(256), 0, st, (const SwinRedDesc*)descs, nd);
    GDL_CHECK_LAUNCH("swin_partial_reduce_batched_kernel");
    return GDL_OK;
}

static int attn_geom(SwinAttnGeom* g, int H, int W, int ws, int shift, int nh, int ld) {
    GDL_REQUIRE(ws >= 1 && ws * ws <= SW_MAXT && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws && nh * SW_HD <= ld,
                "swin_attn: window %d (shift %d) on %dx%d tokens, %d heads in %d channels", ws, shift, H, W, nh, ld);
    g->H = H, g->W = W, g->ws = ws, g->shift = shift, g->nh = nh, g->ld = ld;
    g->nwin = (H / ws) * (W / ws);
    return GDL_OK;
}
int swin_attn_fwd(int dt, const void* qkv, const float* table, void* out, int n_img, int H, int W, int ws, int shift, int nh, int ld,
                  hipStream_t st) {
    SwinAttnGeom g;
    int rc = attn_geom(&g, H, W, ws, shift, nh, ld);
    if (rc) return rc;
    if (swin_attn7_ok(dt, H, W, ws, shift, nh, ld, n_img)) return swin_attn7_fwd(qkv, table, out, n_img, H, W, shift, nh, ld, st);
    const long units = (long)n_img * g.nwin * nh;
    ProfScope prof("gdl::swin_attn_fwd_kernel", PROF_HBM, st, (double)n_img * H * W * ld * (dt == GDL_F32 ? 4 : 2) * 4);
    SW_DISPATCH(dt, hipLaunchKernelGGL(swin_attn_fwd_kernel<float>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, (const float*)qkv, table, (float*)out, g, n_img),
                hipLaunchKernelGGL(swin_attn_fwd_kernel<bf16>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, (const bf16*)qkv, table, (bf16*)out, g, n_img));
    GDL_CHECK_LAUNCH("swin_attn_fwd_kernel");
    return GDL_OK;
}
// windows per block of the backward: enough blocks to fill the chip, few enough partials
static int attn_bwd_group(int nwin) { return nwin >= 64 ? 4 : 1; }
size_t swin_attn_bwd_ws_bytes(int n_img, int nwin, int ws, int nh) {
    const int G = attn_bwd_group(nwin);
    return (size_t)n_img * ((nwin + G - 1) / G) * nh * (2 * ws - 1) * (2 * ws - 1) * sizeof(float);
}
int swin_attn_bwd(int dt, const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, float* tpart, int n_img,
                  int H, int W, int ws, int shift, int nh, int ld, hipStream_t st) {
    SwinAttnGeom g;
    int rc = attn_geom(&g, H, W, ws, shift, nh, ld);
    if (rc) return rc;
    GDL_REQUIRE(tpart && dtable, "swin_attn_bwd: null workspace");
    const int tt = (2 * ws - 1) * (2 * ws - 1);
    if (swin_attn7_ok(dt, H, W, ws, shift, nh, ld, n_img)) {
        int nparts = 0;
        rc = swin_attn7_bwd(qkv, table, dout, dqkv, tpart, &nparts, n_img, H, W, shift, nh, ld, st);
        if (rc) return rc;
        hipLaunchKernelGGL(swin_table_reduce_kernel, dim3((nh * tt + 3) / 4), dim3(256), 0, st, tpart, dtable, nparts, nh, tt);
        GDL_CHECK_LAUNCH("swin_table_reduce_kernel");
        return GDL_OK;
    }
    const size_t lds = (size_t)4 * SW_MAXT * SW_PD * 4 + (size_t)SW_MAXT * (SW_MAXT + 1) * 4 + 3 * SW_MAXT * 4 + 169 * 4 + SW_MAXT * 4;
    const int G = attn_bwd_group(g.nwin), ngrp = (g.nwin + G - 1) / G;
    static DevOnce attr[2];
    const int di = dt == GDL_F32 ? 0 : 1;
    if (!attr[di]) {
        hipError_t e = dt == GDL_F32 ? hipFuncSetAttribute((const void*)swin_attn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                     : hipFuncSetAttribute((const void*)swin_attn_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(swin_attn_bwd)");
        attr[di] = true;
    }
    {
        ProfScope prof("gdl::swin_attn_bwd_kernel", PROF_HBM, st, (double)n_img * H * W * ld * (dt == GDL_F32 ? 4 : 2) * 7);
        SW_DISPATCH(dt, hipLaunchKernelGGL(swin_attn_bwd_kernel<float>, dim3(n_img * ngrp * nh), dim3(64), lds, st, (const float*)qkv, table, (const float*)dout, (float*)dqkv, tpart, g, G),
                    hipLaunchKernelGGL(swin_attn_bwd_kernel<bf16>, dim3(n_img * ngrp * nh), dim3(64), lds, st, (const bf16*)qkv, table, (const bf16*)dout, (bf16*)dqkv, tpart, g, G));
        GDL_CHECK_LAUNCH("swin_attn_bwd_kernel");
    }
    hipLaunchKernelGGL(swin_table_reduce_kernel, dim3((nh * tt + 3) / 4), dim3(256), 0, st, tpart, dtable, n_img * ngrp, nh, tt);
    GDL_CHECK_LAUNCH("swin_table_reduce_kernel");
    return GDL_OK;
}

int swin_merge(int dt, const void* src, void* dst, int N, int H, int W, int C, int ldx, int scatter, hipStream_t st) {
    GDL_REQUIRE(H % 2 == 0 && W % 2 == 0 && C <= ldx, "swin_merge: %dx%d tokens, %d / %d channels", H, W, C, ldx);
    const size_t total = (size_t)N * H * W * ldx;
    ProfScope prof("gdl::swin_merge_kernel", PROF_HBM, st, (double)total * (dt == GDL_F32 ? 8 : 4));
    const int epc = dt == GDL_F32 ? 4 : 8;
    if (C % epc == 0 && ldx % epc == 0 && (size_t)N * H * W < ((size_t)1 << 31)) {
        const int g = sw_grid(total / epc, 256, 256 * 16);
        if (dt == GDL_F32) {
            if (scatter)
                hipLaunchKernelGGL((swin_merge_vec_kernel<float, 1>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, ldx);
            else
                hipLaunchKernelGGL((swin_merge_vec_kernel<float, 0>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, ldx);
        } else {
            if (scatter)
                hipLaunchKernelGGL((swin_merge_vec_kernel<bf16, 1>), dim3(g), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, N, H, W, C, ldx);
            else
                hipLaunchKernelGGL((swin_merge_vec_kernel<bf16, 0>), dim3(g), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, N, H, W, C, ldx);
        }
        GDL_CHECK_LAUNCH("swin_merge_vec_kernel");
        return GDL_OK;
    }
    if (dt == GDL_F32) {
        if (scatter)
            hipLaunchKernelGGL((swin_merge_kernel<float, 1>), dim3(sw_grid(total)), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, ldx);
        else
            hipLaunchKernelGGL((swin_merge_kernel<float, 0>), dim3(sw_grid(total)), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, ldx);
    } else {
        if (scatter)
            hipLaunchKernelGGL((swin_merge_kernel<bf16, 1>), dim3(sw_grid(total)), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, N, H, W, C, ldx);
        else
            hipLaunchKernelGGL((swin_merge_kernel<bf16, 0>), dim3(sw_grid(total)), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, N, H, W, C, ldx);
    }
    GDL_CHECK_LAUNCH("swin_merge_kernel");
    return GDL_OK;
}

int swin_token_mean(int dt, const void* x, float* y, int N, int L, int C, int ld, hipStream_t st) {
    const int epc = dt == GDL_BF16 ? 8 : 4;
    if (C % epc == 0 && ld % epc == 0 && ((uintptr_t)x & 15) == 0) {  // whole 16-byte chunks: the vector form
        const dim3 grid((unsigned)((C + 8 * epc - 1) / (8 * epc)), (unsigned)N);
        SW_DISPATCH(dt, hipLaunchKernelGGL(swin_token_mean_vec_kernel<float>, grid, dim3(256), 0, st, (const float*)x, y, L, C, ld),
                    hipLaunchKernelGGL(swin_token_mean_vec_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, y, L, C, ld));
        GDL_CHECK_LAUNCH("swin_token_mean_vec_kernel");
        return GDL_OK;
    }
    SW_DISPATCH(dt, hipLaunchKernelGGL(swin_token_mean_kernel<float>, dim3(N), dim3(256), 0, st, (const float*)x, y, L, C, ld),
                hipLaunchKernelGGL(swin_token_mean_kernel<bf16>, dim3(N), dim3(256), 0, st, (const bf16*)x, y, L, C, ld));
    GDL_CHECK_LAUNCH("swin_token_mean_kernel");
    return GDL_OK;
}
int swin_token_mean_bwd(int dt, const float* dy, void* dx, int N, int L, int C, int ld, hipStream_t st) {
    const size_t total = (size_t)N * L * ld;
    SW_DISPATCH(dt, hipLaunchKernelGGL(swin_token_mean_bwd_kernel<float>, dim3(sw_grid(total)), dim3(256), 0, st, dy, (float*)dx, N, L, C, ld),
                hipLaunchKernelGGL(swin_token_mean_bwd_kernel<bf16>, dim3(sw_grid(total)), dim3(256), 0, st, dy, (bf16*)dx, N, L, C, ld));
    GDL_CHECK_LAUNCH("swin_token_mean_bwd_kernel");
    return GDL_OK;
}

static int seg_of(SwinSeg* s, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad) {
    GDL_REQUIRE(n > 0 && k > 0 && nseg > 0 && kseg > 0 && n % nseg == 0 && k % kseg == 0 && nseg_pad >= nseg && kseg_pad >= kseg,
                "swin_pack: bad segmentation");
    s->n = n, s->k = k, s->nseg = nseg, s->nseg_pad = nseg_pad, s->kseg = kseg, s->kseg_pad = kseg_pad;
    s->np = (n / nseg) * nseg_pad, s->kp = (k / kseg) * kseg_pad;
    return GDL_OK;
}
int swin_pack_matrix(int dt, const float* src, void* dst, void* dstT, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad,
                     hipStream_t st) {
    SwinSeg s;
    int rc = seg_of(&s, n, k, nseg, nseg_pad, kseg, kseg_pad);
    if (rc) return rc;
    const size_t total = (size_t)s.np * s.kp;
    if (dt == GDL_F32)
        hipLaunchKernelGGL(swin_pack_kernel<float>, dim3(sw_grid(total)), dim3(256), 0, st, src, (float*)dst, (float*)dstT, s);
    else if (dt == GDL_BF16)
        hipLaunchKernelGGL(swin_pack_kernel<bf16>, dim3(sw_grid(total)), dim3(256), 0, st, src, (bf16*)dst, (bf16*)dstT, s);
    else
        GDL_REQUIRE(false, "swin_pack_matrix: dtype %d", dt);
    GDL_CHECK_LAUNCH("swin_pack_kernel");
    return GDL_OK;
}
int swin_unpack_matrix(const float* src, float* dst, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad, hipStream_t st) {
    SwinSeg s;
    int rc = seg_of(&s, n, k, nseg, nseg_pad, kseg, kseg_pad);
    if (rc) return rc;
    hipLaunchKernelGGL(swin_unpack_kernel, dim3(sw_grid((size_t)n * k)), dim3(256), 0, st, src, dst, s);
    GDL_CHECK_LAUNCH("swin_unpack_kernel");
    return GDL_OK;
}

int swin_pack_batched(const void* descs, int nd, int total_blocks, int dir, hipStream_t st) {
    GDL_REQUIRE(descs && nd > 0 && total_blocks > 0 && (dir == 0 || dir == 1), "swin_pack_batched: bad arguments");
    hipLaunchKernelGGL(swin_pack_batched_kernel, dim3(total_blocks), dim3(256), 0, st, (const SwinPackDesc*)descs, nd, dir);
    GDL_CHECK_LAUNCH("swin_pack_batched_kernel");
    return GDL_OK;
}

}  // namespace gdl
