// fold.h -- "the last block folds": in-launch, fixed-order reduction of per-block partial rows + the BatchNorm
// finalize arithmetic, so that the 80 finalize launches per step (bn_finalize_train / bn_bwd_finalize: one tiny
// kernel between two dependent kernels of a chain costs the chain 5-10 us each) disappear from the chains.
//
// Protocol (placement independent, no spinning; MI355X guide, guideline 16, counter form):
//   * every block writes its partial row [C][2] with agent-scope (write-through, `sc1`) stores, every storing wave
//     drains `vmcnt(0)`, the block synchronises, ONE lane takes a ticket on the counter of the row's group of
//     FOLD_G rows (relaxed agent atomic);
//   * the block that draws the last ticket of a group issues ONE agent acquire, folds the group's rows in row order
//     (double accumulators), publishes the group row the same way and takes a ticket on the top counter;
//   * the block that draws the last top ticket folds the group rows in order and runs the finalize functor.
// Every sum has a fixed association (rows in index order within a row-lane, row-lanes in order, groups in order):
// results are bit-identical from run to run whatever the arrival order.  Counters are zeroed once when the
// workspace is bound and reset by the block that draws the last ticket (kernels that fold are ordered by their
// stream, which owns the workspace).
#pragma once
#include "common.h"

namespace gdl {

constexpr int FOLD_G = 64;      // partial rows per first-level group
constexpr int FOLD_NCG = 8;     // column groups (conv: OC / BN <= 8)
constexpr int FOLD_MAXG = 256;  // first-level groups (rows <= 16384)
constexpr int FOLD_CMAX = 512;  // channels
constexpr int FOLD_LDS = 16 + 256 * 2 * 8;  // LDS scratch bytes the fold needs
constexpr int FOLD_U = 16;       // rows a thread keeps in flight

struct FoldWs {
    unsigned* ctr;  // [FOLD_NCG][1 + FOLD_MAXG]; nullptr = no fold (the caller launches the finalize kernel)
    double* gpart;  // [2][FOLD_MAXG][FOLD_CMAX][2] (the second array: a kernel that folds two BatchNorms)
};
static inline size_t fold_ctr_bytes() { return (size_t)FOLD_NCG * (1 + FOLD_MAXG) * sizeof(unsigned); }
static inline size_t fold_gpart_bytes() { return 2 * (size_t)FOLD_MAXG * FOLD_CMAX * 2 * sizeof(double); }
static inline bool fold_fits(long rows, int C) { return rows <= (long)FOLD_G * FOLD_MAXG && C <= FOLD_CMAX; }

// forward: batch statistics -> save_mean / save_rstd / scale / shift and the running statistics
// (nn.BatchNorm2d in training mode, /root/reference/models/backbone.py:45,48,104,144)
struct FinTrain {
    const float *gamma, *beta;
    float *running_mean, *running_var;
    int64_t* nbt;
    float *save_mean, *save_rstd, *scale, *shift;
    double count;
    float eps, momentum;
    __device__ __forceinline__ void operator()(int c, double s, double q) const {
        const double mean = s / count;
        double var = q / count - mean * mean;  // biased
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        save_mean[c] = (float)mean;
        save_rstd[c] = (float)rstd;
        scale[c] = (float)((double)gamma[c] * rstd);
        shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
        if (nbt && c == 0) *nbt += 1;
    }
};
// backward: dbeta = sum g', dgamma = sum g'*xhat, coef = {dbeta / M, dgamma / M}
struct FinBwd {
    float *dgamma, *dbeta, *coef;
    int C;
    double count;
    __device__ __forceinline__ void operator()(int c, double a, double b) const {
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
        coef[c] = (float)(a / count);
        coef[C + c] = (float)(b / count);
    }
};

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 ld2_agent(const float* p) {  // 8-byte aligned pair
    const unsigned long long v =
        __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane: ticket on `ctr`; true for the block that draws ticket `expected - 1` (which resets the counter and
// acquires).  Call after every storing wave has drained and the block has synchronised.
__device__ __forceinline__ bool fold_ticket(unsigned* ctr, unsigned expected) {
    const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == expected - 1u;
    if (last) {
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return last;
}

// Called by ALL 256 threads of every block that wrote a partial row (with st_agent) of `partial` [rows][Ctot][2]:
// this block wrote the channels [c0, c0 + ncols) of row `row`; `cg` numbers the column group (blocks with equal c0).
// ncols is 64, 128 or a multiple of 256.  `lds`: FOLD_LDS bytes of the block's LDS, free to overwrite.
template <class Fin>
__device__ __forceinline__ void fold_finalize(const float* partial, int rows, int Ctot, int c0, int ncols, int row, int cg,
                                              const FoldWs& ws, unsigned char* lds, const Fin& fin) {
    const int tid = threadIdx.x;
    volatile unsigned* flag = (volatile unsigned*)lds;
    double* comb = (double*)(lds + 16);
    unsigned* ctr = ws.ctr + (size_t)cg * (1 + FOLD_MAXG);
    const int g = row / FOLD_G, ng = (rows + FOLD_G - 1) / FOLD_G;
    const int r0 = g * FOLD_G, gsize = min(FOLD_G, rows - r0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the row has left the CU
    __syncthreads();
    if (tid == 0) flag[0] = fold_ticket(&ctr[1 + g], (unsigned)gsize) ? 1u : 0u;
    __syncthreads();
    if (!flag[0]) return;
    // ---- level 1: rows [r0, r0 + gsize) -> one group row (or, with a single group, straight to the finalize)
    bool top = ng == 1;
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass ? ng : gsize;
        for (int cb = 0; cb < ncols; cb += 256) {
            const int ncc = min(256, ncols - cb), RL = 256 / ncc;
            const int cl = tid % ncc, rl = tid / ncc, c = c0 + cb + cl;
            double s = 0.0, q = 0.0;
            // plain loads (legal behind the acquire), FOLD_U rows in flight per thread: the fold's cost is the latency of
            // these dependent round trips to the memory side, not their bytes
            for (int t = rl; t < n; t += FOLD_U * RL) {
                double vs[FOLD_U], vq[FOLD_U];
#pragma unroll
                for (int u = 0; u < FOLD_U; ++u) {
                    const int r = t + u * RL;
                    vs[u] = vq[u] = 0.0;
                    if (r < n) {
                        if (pass) {
                            const double2 v = *(const double2*)(ws.gpart + ((size_t)r * Ctot + c) * 2);
                            vs[u] = v.x;
                            vq[u] = v.y;
                        } else {
                            const float2 v = *(const float2*)(partial + ((size_t)(r0 + r) * Ctot + c) * 2);
                            vs[u] = (double)v.x;
                            vq[u] = (double)v.y;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < FOLD_U; ++u) {
                    s += vs[u];
                    q += vq[u];
                }
            }
            comb[tid * 2 + 0] = s;
            comb[tid * 2 + 1] = q;
            __syncthreads();
            if (rl == 0) {
                for (int j = 1; j < RL; ++j) {  // row-lanes in order
                    s += comb[(j * ncc + cl) * 2 + 0];
                    q += comb[(j * ncc + cl) * 2 + 1];
                }
                if (top) {
                    fin(c, s, q);
                } else {
                    st_agent(ws.gpart + ((size_t)g * Ctot + c) * 2, s);
                    st_agent(ws.gpart + ((size_t)g * Ctot + c) * 2 + 1, q);
                }
            }
            __syncthreads();
        }
        if (top) return;
        // ---- the group row is published; the last group to arrive folds the group rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) flag[1] = fold_ticket(&ctr[0], (unsigned)ng) ? 1u : 0u;
        __syncthreads();
        if (!flag[1]) return;
        top = true;
    }
}

}  // namespace gdl
