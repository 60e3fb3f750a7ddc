// head.hip -- gradient-truncated fusion head and cross-entropy, float32.
//
// ConcatFusion_DGL.forward (/root/reference/models/fusion_modules.py:51-59):
//   output = fc_out(cat(x, y).detach());  x_out = fc_out(cat(x, 0));  y_out = fc_out(cat(0, y))
// ConcatFusion.forward (:38-42): output = fc_out(cat(x, y)).
// nn.CrossEntropyLoss (main_dgl.py:71,102-104).
// The reference spends ~20 tiny launches here (3 cats, 2 zero fills, 3 GEMMs and their
// backwards); the three logit sets share the two half dot-products W[:, :512].x and
// W[:, 512:].y, so one launch produces all of them.  Sizes are tiny (B x 1024 x n_classes):
// latency bound, no MFMA.
#include "common.h"
#include "prof.h"

namespace gdl {

constexpr int HD = 512;  // features per modality

// The two linear maps of a head and their biases.  ConcatFusion(_DGL): one [n][1024] matrix, Wx = W,
// Wy = W + 512, ldw = 1024, one bias used by all three logit sets.  SumFusion_DGL (fusion_modules.py:16-30):
// two [n][512] matrices fc_x / fc_y with their own biases; `output` carries bx + by.
struct HeadW {
    const float *Wx, *Wy;
    int ldw;
    const float *bx, *by;
    int sum_bias;  // 1: out gets bx + by (sum head), 0: out gets bx (== by, concat head)
};

// grid = B; each wave handles classes j = wave, wave+4, ...; lanes split the 512-long dots.
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, HeadW w,
                                                       float* __restrict__ out, float* __restrict__ x_out,
                                                       float* __restrict__ y_out, int n) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float xv[8], yv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xv[i] = x[(size_t)b * HD + lane + 64 * i];
        yv[i] = y[(size_t)b * HD + lane + 64 * i];
    }
    for (int j = wave; j < n; j += 4) {
        const float* wx = w.Wx + (size_t)j * w.ldw;
        const float* wy = w.Wy + (size_t)j * w.ldw;
        float pa = 0.f, pv = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pa += wx[lane + 64 * i] * xv[i];
            pv += wy[lane + 64 * i] * yv[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pa += __shfl_xor(pa, o);
            pv += __shfl_xor(pv, o);
        }
        if (lane == 0) {
            const float bxj = w.bx[j], byj = w.by[j];
            out[(size_t)b * n + j] = pa + pv + (w.sum_bias ? bxj + byj : bxj);
            if (x_out) x_out[(size_t)b * n + j] = pa + bxj;
            if (y_out) y_out[(size_t)b * n + j] = pv + byj;
        }
    }
}
int head_concat_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out, float* y_out,
                    int B, int n, hipStream_t st) {
    const HeadW w{W, W + HD, 2 * HD, b, b, 0};
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, st, x, y, w, out, x_out, y_out, n);
    GDL_CHECK_LAUNCH("head_fwd_kernel");
    return GDL_OK;
}
int head_sum_fwd(const float* x, const float* y, const float* Wx, const float* bx, const float* Wy, const float* by, float* out,
                 float* x_out, float* y_out, int B, int n, hipStream_t st) {
    const HeadW w{Wx, Wy, HD, bx, by, 1};
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, st, x, y, w, out, x_out, y_out, n);
    GDL_CHECK_LAUNCH("head_fwd_kernel");
    return GDL_OK;
}

// dx[b][i] = sum_j gx[b][j] * Wx[j][i]  (gx = g_x_out (+ g_out)), same for dy with Wy
__global__ __launch_bounds__(256) void head_bwd_feat_kernel(HeadW w, const float* __restrict__ g_x_out,
                                                            const float* __restrict__ g_y_out,
                                                            const float* __restrict__ g_out, int out_reaches_xy,
                                                            float* __restrict__ dx, float* __restrict__ dy, int n) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * HD; i += 256) {
        const bool is_y = i >= HD;
        const float* gu = is_y ? g_y_out : g_x_out;
        const float* W = is_y ? w.Wy : w.Wx;
        const int fi = is_y ? i - HD : i;
        float s = 0.f;
        for (int j = 0; j < n; ++j) {
            float g = gu ? gu[(size_t)b * n + j] : 0.f;
            if (out_reaches_xy && g_out) g += g_out[(size_t)b * n + j];
            s += g * W[(size_t)j * w.ldw + fi];
        }
        if (is_y)
            dy[(size_t)b * HD + fi] = s;
        else
            dx[(size_t)b * HD + fi] = s;
    }
}
// dWx[j][i] = sum_b gw[b][j] * x[b][i], gw = g_out (+ g_x_out if uni_in_dw); dWy likewise with y / g_y_out.
// concat head: one bias, db = sum_b (g_out + uni*(g_x_out + g_y_out)); sum head: dbx = sum_b (g_out + uni*g_x_out),
// dby likewise.   grid = n classes.
// grid = (n classes, 2*HD/256): one thread per weight-gradient element (six blocks looping over 8 x 64 dependent
// iterations took 77 us -- in front of both encoders' backward passes); the batch's logit gradients of class j are staged
// in LDS, the sum over the batch runs in index order (bit-identical to the serial form).
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ g_x_out,
                                                         const float* __restrict__ g_y_out, const float* __restrict__ g_out,
                                                         int uni_in_dw, float* __restrict__ dWx, float* __restrict__ dWy,
                                                         int ldw, float* __restrict__ dbx, float* __restrict__ dby, int B,
                                                         int n) {
    __shared__ float gs[256];
    const int j = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;  // 0 .. 2*HD-1 (HD is a multiple of 256: a block is all-x or all-y)
    const bool is_y = i >= HD;
    const float* feat = is_y ? y : x;
    const float* gu = is_y ? g_y_out : g_x_out;
    const int fi = is_y ? i - HD : i;
    float s = 0.f;
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int nb = min(256, B - b0);
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const int b = b0 + threadIdx.x;
            float g = g_out ? g_out[(size_t)b * n + j] : 0.f;
            if (uni_in_dw && gu) g += gu[(size_t)b * n + j];
            gs[threadIdx.x] = g;
        }
        __syncthreads();
#pragma unroll 8
        for (int b = 0; b < nb; ++b) s += gs[b] * feat[(size_t)(b0 + b) * HD + fi];
    }
    (is_y ? dWy : dWx)[(size_t)j * ldw + fi] = s;
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float so = 0.f, sx = 0.f, sy = 0.f;
        for (int b = 0; b < B; ++b) {
            so += g_out ? g_out[(size_t)b * n + j] : 0.f;
            if (uni_in_dw) {
                if (g_x_out) sx += g_x_out[(size_t)b * n + j];
                if (g_y_out) sy += g_y_out[(size_t)b * n + j];
            }
        }
        if (dby) {  // sum head: two biases
            dbx[j] = so + sx;
            dby[j] = so + sy;
        } else {
            dbx[j] = so + sx + sy;
        }
    }
}
static_assert(HD % 256 == 0, "head_bwd_w_kernel: a block must not straddle the two feature vectors");
int head_concat_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out,
                    const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db,
                    int B, int n, hipStream_t st) {
    if (dx && dy) {
        const HeadW w{W, W + HD, 2 * HD, nullptr, nullptr, 0};
        hipLaunchKernelGGL(head_bwd_feat_kernel, dim3(B), dim3(256), 0, st, w, g_x_out, g_y_out, g_out, out_reaches_xy, dx, dy,
                           n);
        GDL_CHECK_LAUNCH("head_bwd_feat_kernel");
    }
    if (dW && db) {
        hipLaunchKernelGGL(head_bwd_w_kernel, dim3(n, 2 * HD / 256), dim3(256), 0, st, x, y, g_x_out, g_y_out, g_out, uni_in_dw, dW, dW + HD,
                           2 * HD, db, (float*)nullptr, B, n);
        GDL_CHECK_LAUNCH("head_bwd_w_kernel");
    }
    return GDL_OK;
}
// ---------------------------------------------------------------------------------------------------------------------
// Concat head with unequal feature widths (the Swin composition: 512 audio + 768 visual; cf. ConcatFusion_Swin,
// fusion_modules.py:79-88, and ConcatFusion_DGL, :45-59): W [n][dxw + dyw].  Same contract as head_concat_fwd / _bwd.
// Tiny problems (B x 1280 x n): plain FMA, fixed summation order.
// grid = (B, ceil(n / HXY_CH)): a block's four waves own HXY_CH classes of one sample (64 blocks walking all 309 classes of
// ConcatFusion_Swin's head, reloading the sample's features for every class, took 550 us); per class the sums run over
// i = lane, lane + 64, ... and the xor butterfly as before.
constexpr int HXY_CH = 8;
__global__ __launch_bounds__(256) void head_xy_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ out, float* __restrict__ x_out,
                                                          float* __restrict__ y_out, int n, int dxw, int dyw) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXV = 16;  // feature values per lane kept in registers (widths up to 1024; beyond: read per class)
    float xs[MAXV], ys[MAXV];
    const bool reg = dxw <= 64 * MAXV && dyw <= 64 * MAXV;
    if (reg) {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            xs[v] = lane + 64 * v < dxw ? x[(size_t)b * dxw + lane + 64 * v] : 0.f;
            ys[v] = lane + 64 * v < dyw ? y[(size_t)b * dyw + lane + 64 * v] : 0.f;
        }
    }
    // (round 5: two classes per wave at a time -- their loads and butterflies overlap -- and 8 classes per block: 36 us at
    // 64 x 309 x 1280 against 63 with 32 classes per block walked one by one)
    const int j1 = min(n, ((int)blockIdx.y + 1) * HXY_CH);
    for (int j = blockIdx.y * HXY_CH + wave; j < j1; j += 8) {
        const int j2 = j + 4 < j1 ? j + 4 : j;
        const float* w = W + (size_t)j * (dxw + dyw);
        const float* w2 = W + (size_t)j2 * (dxw + dyw);
        float pa = 0.f, pv = 0.f, qa = 0.f, qv = 0.f;
        if (reg) {
#pragma unroll
            for (int v = 0; v < MAXV; ++v)
                if (lane + 64 * v < dxw) pa += w[lane + 64 * v] * xs[v];
#pragma unroll
            for (int v = 0; v < MAXV; ++v)
                if (lane + 64 * v < dyw) pv += w[dxw + lane + 64 * v] * ys[v];
#pragma unroll
            for (int v = 0; v < MAXV; ++v)
                if (lane + 64 * v < dxw) qa += w2[lane + 64 * v] * xs[v];
#pragma unroll
            for (int v = 0; v < MAXV; ++v)
                if (lane + 64 * v < dyw) qv += w2[dxw + lane + 64 * v] * ys[v];
        } else {
            for (int i = lane; i < dxw; i += 64) pa += w[i] * x[(size_t)b * dxw + i];
            for (int i = lane; i < dyw; i += 64) pv += w[dxw + i] * y[(size_t)b * dyw + i];
            for (int i = lane; i < dxw; i += 64) qa += w2[i] * x[(size_t)b * dxw + i];
            for (int i = lane; i < dyw; i += 64) qv += w2[dxw + i] * y[(size_t)b * dyw + i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pa += __shfl_xor(pa, o);
            pv += __shfl_xor(pv, o);
            qa += __shfl_xor(qa, o);
            qv += __shfl_xor(qv, o);
        }
        if (lane == 0) {
            out[(size_t)b * n + j] = pa + pv + bias[j];
            if (x_out) x_out[(size_t)b * n + j] = pa + bias[j];
            if (y_out) y_out[(size_t)b * n + j] = pv + bias[j];
            if (j2 != j) {
                out[(size_t)b * n + j2] = qa + qv + bias[j2];
                if (x_out) x_out[(size_t)b * n + j2] = qa + bias[j2];
                if (y_out) y_out[(size_t)b * n + j2] = qv + bias[j2];
            }
        }
    }
}
// grid = (B, ceil((dxw + dyw) / 256)): a thread per feature; the sample's class gradients staged in LDS once, the walk over the
// classes in ascending order with eight weight loads in flight
__global__ __launch_bounds__(256) void head_xy_bwd_feat_kernel(const float* __restrict__ W, const float* __restrict__ g_x_out,
                                                               const float* __restrict__ g_y_out, const float* __restrict__ g_out,
                                                               int out_reaches_xy, float* __restrict__ dx, float* __restrict__ dy,
                                                               int n, int dxw, int dyw) {
    extern __shared__ float hxy_g[];  // [2][n]: the x side's and the y side's class gradients of this sample
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float go = (out_reaches_xy && g_out) ? g_out[(size_t)b * n + j] : 0.f;
        float gx = g_x_out ? g_x_out[(size_t)b * n + j] : 0.f, gy = g_y_out ? g_y_out[(size_t)b * n + j] : 0.f;
        if (out_reaches_xy && g_out) gx += go, gy += go;
        hxy_g[j] = gx;
        hxy_g[n + j] = gy;
    }
    __syncthreads();
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= dxw + dyw) return;
    const bool is_y = i >= dxw;
    const float* g = hxy_g + (is_y ? n : 0);
    const float* w = W + i;
    const int ldw = dxw + dyw;
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = w[(size_t)(j + u) * ldw];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += g[j + u] * q[u];
    }
    for (; j < n; ++j) s += g[j] * w[(size_t)j * ldw];
    if (is_y)
        dy[(size_t)b * dyw + i - dxw] = s;
    else
        dx[(size_t)b * dxw + i] = s;
}
// grid = (n, ceil((dxw + dyw) / 256)); db by the first block row
__global__ __launch_bounds__(256) void head_xy_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ g_x_out, const float* __restrict__ g_y_out,
                                                            const float* __restrict__ g_out, int uni_in_dw, float* __restrict__ dW,
                                                            float* __restrict__ db, int B, int n, int dxw, int dyw) {
    const int j = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;
    if (i < dxw + dyw) {
        const bool is_y = i >= dxw;
        const float* gu = is_y ? g_y_out : g_x_out;
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            float g = g_out ? g_out[(size_t)b * n + j] : 0.f;
            if (uni_in_dw && gu) g += gu[(size_t)b * n + j];
            s += g * (is_y ? y[(size_t)b * dyw + i - dxw] : x[(size_t)b * dxw + i]);
        }
        dW[(size_t)j * (dxw + dyw) + i] = s;
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            s += g_out ? g_out[(size_t)b * n + j] : 0.f;
            if (uni_in_dw) s += (g_x_out ? g_x_out[(size_t)b * n + j] : 0.f) + (g_y_out ? g_y_out[(size_t)b * n + j] : 0.f);
        }
        db[j] = s;
    }
}
int head_concat_xy_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out, float* y_out, int B,
                       int n, int dxw, int dyw, hipStream_t st) {
    hipLaunchKernelGGL(head_xy_fwd_kernel, dim3(B, (n + HXY_CH - 1) / HXY_CH), dim3(256), 0, st, x, y, W, b, out, x_out, y_out, n, dxw, dyw);
    GDL_CHECK_LAUNCH("head_xy_fwd_kernel");
    return GDL_OK;
}
int head_concat_xy_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out, const float* g_out,
                       int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db, int B, int n, int dxw, int dyw,
                       hipStream_t st) {
    if (dx && dy) {
        GDL_REQUIRE(n <= 4096, "head_concat_xy_bwd: at most 4096 classes");
        hipLaunchKernelGGL(head_xy_bwd_feat_kernel, dim3(B, (dxw + dyw + 255) / 256), dim3(256), 2 * n * sizeof(float), st, W, g_x_out, g_y_out,
                           g_out, out_reaches_xy, dx, dy, n, dxw, dyw);
        GDL_CHECK_LAUNCH("head_xy_bwd_feat_kernel");
    }
    if (dW && db) {
        hipLaunchKernelGGL(head_xy_bwd_w_kernel, dim3(n, (dxw + dyw + 255) / 256), dim3(256), 0, st, x, y, g_x_out, g_y_out, g_out,
                           uni_in_dw, dW, db, B, n, dxw, dyw);
        GDL_CHECK_LAUNCH("head_xy_bwd_w_kernel");
    }
    return GDL_OK;
}

int head_sum_bwd(const float* x, const float* y, const float* Wx, const float* Wy, const float* g_x_out, const float* g_y_out,
                 const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dWx, float* dbx,
                 float* dWy, float* dby, int B, int n, hipStream_t st) {
    if (dx && dy) {
        const HeadW w{Wx, Wy, HD, nullptr, nullptr, 1};
        hipLaunchKernelGGL(head_bwd_feat_kernel, dim3(B), dim3(256), 0, st, w, g_x_out, g_y_out, g_out, out_reaches_xy, dx, dy,
                           n);
        GDL_CHECK_LAUNCH("head_bwd_feat_kernel");
    }
    if (dWx && dWy) {
        hipLaunchKernelGGL(head_bwd_w_kernel, dim3(n, 2 * HD / 256), dim3(256), 0, st, x, y, g_x_out, g_y_out, g_out, uni_in_dw, dWx, dWy, HD,
                           dbx, dby, B, n);
        GDL_CHECK_LAUNCH("head_bwd_w_kernel");
    }
    return GDL_OK;
}

// loss = mean_b ( logsumexp(l_b) - l_b[label_b] ); dlogits = scale*(softmax - onehot)/B.  One block.
// A wave per sample (samples b = wave, wave + 4, ...): the lanes evaluate the exponentials, lane 0 adds them in class order --
// every sum in the order of the first form of this function (one THREAD per sample walking the classes twice with expf in the
// dependent chain: 141 us at 64 samples x 309 classes, three blocks busy), so losses and gradients keep their bits.
// Round 5: the block has 1024 threads and the wave form runs on up to 16 waves (as many as fit the 32 KiB of staged
// exponentials, a power of two so that samples b and b + 256 stay with one wave): 93 -> 31 us at 64 x 309 on the step's tail.
constexpr int CE_MAXN = 1024;  // classes the wave form stages in LDS (beyond: the thread-per-sample walk)
constexpr int CE_EX = 8192;    // floats of staged exponentials
__device__ __forceinline__ void softmax_ce_block(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         float scale, float* __restrict__ loss, float* __restrict__ dlogits,
                                                         int B, int n) {
    __shared__ float part[256];
    __shared__ float ex[CE_EX];
    if (threadIdx.x < 256) part[threadIdx.x] = 0.f;
    __syncthreads();
    if (n >= 64 && n <= CE_MAXN) {  // (few classes: a thread per sample is the faster walk -- 6 us against 24 at 64 x 6; same bits)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int aw = min((int)(blockDim.x >> 6), CE_EX / n);  // active waves: 16, 8 (n > 512)
        aw = aw >= 16 ? 16 : (aw >= 8 ? 8 : 4);
        float* exw = ex + wave * n;
        for (int b = wave; b < B && wave < aw; b += aw) {  // (b, b + 256, ... belong to the same wave, ascending: part[b % 256] sums in the old order)
            const float* l = logits + (size_t)b * n;
            float mx = -INFINITY;
            for (int j = lane; j < n; j += 64) mx = fmaxf(mx, l[j]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            for (int j = lane; j < n; j += 64) exw[j] = expf(l[j] - mx);
            __builtin_amdgcn_wave_barrier();
            float se = 0.f;
            if (lane == 0)
                for (int j = 0; j < n; ++j) se += exw[j];
            se = __shfl(se, 0);
            const float lse = mx + logf(se);
            const long lab64 = (long)labels[b];
            const bool lab_ok = lab64 >= 0 && lab64 < n;
            const int lab = lab_ok ? (int)lab64 : -1;
            if (lane == 0) part[b & 255] += lab_ok ? lse - l[lab] : __builtin_nanf("");
            if (dlogits)
                for (int j = lane; j < n; j += 64)
                    dlogits[(size_t)b * n + j] = scale * (expf(l[j] - lse) - (j == lab ? 1.f : 0.f)) / (float)B;
            __builtin_amdgcn_wave_barrier();
        }
    } else if (threadIdx.x < 256) {
        float acc = 0.f;
        for (int b = threadIdx.x; b < B; b += 256) {
            const float* l = logits + (size_t)b * n;
            float mx = l[0];
            for (int j = 1; j < n; ++j) mx = fmaxf(mx, l[j]);
            float se = 0.f;
            for (int j = 0; j < n; ++j) se += expf(l[j] - mx);
            const float lse = mx + logf(se);
            // a class index outside [0, n) (a device assert in the reference's CrossEntropyLoss) must not become an
            // out-of-bounds read: the sample contributes NaN to the loss (visible in DGLTrainer.read()) and no one-hot term
            const long lab64 = (long)labels[b];
            const bool lab_ok = lab64 >= 0 && lab64 < n;
            const int lab = lab_ok ? (int)lab64 : -1;
            acc += lab_ok ? lse - l[lab] : __builtin_nanf("");
            if (dlogits)
                for (int j = 0; j < n; ++j)
                    dlogits[(size_t)b * n + j] = scale * (expf(l[j] - lse) - (j == lab ? 1.f : 0.f)) / (float)B;
        }
        part[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0] / (float)B;
}
// ---------------------------------------------------------------- GatedFusion_DGL (fusion_modules.py:213-250)
//   hx = fc_x(x), hy = fc_y(y)                              (Linear(512, 512) each)
//   output = fc_out(sigmoid(hx.detach()) * hy.detach())     (x_gate = True, basic_model.py:38)
//   out_x  = fc_out(sigmoid(hx) * hx),  out_y = fc_out(sigmoid(hy) * hy)       (the "gate" of each unimodal
//   logit set is the modality's own hidden vector: h * sigmoid(h))
// B x 512 x 512 products: tiny (33 MFLOP), latency bound, plain FMA kernels.
// ---------------------------------------------------------------- one modality's auxiliary path on its own
// u = f Wp^T + bp, d(u) = scale*(softmax(u) - onehot)/B, df = d(u) Wp  -- the unimodal logits, their cross-entropy gradient and
// the feature gradient the encoder's backward starts from, for ONE modality of a DGL concat / sum head (Wp = that modality's
// 512 columns of fc_out, or fc_x / fc_y).  In the DGL step an encoder learns from its own auxiliary loss only
// (main_dgl.py:102-122), so its backward need not wait for the other encoder's forward: DGLTrainer launches this on the
// encoder's own stream right behind its forward.  grid = B.  Every sum runs in the order head_fwd_kernel, softmax_ce_block and
// head_bwd_feat_kernel use: df is bit-identical to the three-launch path.
// Round 5: 16 waves per sample and no serial walk beyond the sums whose order is the contract (at 309 classes x 768 features
// the four-wave form -- 78 classes per wave one after the other, thread 0 alone through max / exp / sum, 927 dependent
// loads per thread for df -- took 113-131 us ON the chain between an encoder's forward and its backward): the classes go
// round the 16 waves two at a time, max and exp are evaluated by all threads (max is exact in any order; the exponentials
// are the same values) and only their SUM is walked in class order by one thread, df keeps its ascending walk per feature
// with eight weight loads in flight.  113 -> 28 us there (kernel trace, in the step).
template <int ND>  // feature width = 64 ND
__global__ __launch_bounds__(1024) void head_uni_dfeat_kernel(const float* __restrict__ f, const float* __restrict__ Wp, int ldw,
                                                             const float* __restrict__ bp, const int64_t* __restrict__ labels,
                                                             float scale, float* __restrict__ df, int B, int n) {
    constexpr int D = 64 * ND, NW = 16;
    __shared__ float lg[512], dl[512], ex[512];
    __shared__ float wmx[NW];
    __shared__ float lse_s;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float fv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) fv[i] = f[(size_t)b * D + lane + 64 * i];
    for (int j = wave; j < n; j += 2 * NW) {
        const int j2 = j + NW;
        const float* w = Wp + (size_t)j * ldw;
        const float* w2 = Wp + (size_t)(j2 < n ? j2 : j) * ldw;
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int i = 0; i < ND; ++i) pa += w[lane + 64 * i] * fv[i];
#pragma unroll
        for (int i = 0; i < ND; ++i) pb += w2[lane + 64 * i] * fv[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pa += __shfl_xor(pa, o);
            pb += __shfl_xor(pb, o);
        }
        if (lane == 0) {
            lg[j] = pa + bp[j];
            if (j2 < n) lg[j2] = pb + bp[j2];
        }
    }
    __syncthreads();
    {
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < n; j += 1024) mx = fmaxf(mx, lg[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) wmx[wave] = mx;
    }
    __syncthreads();
    float mx = wmx[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, wmx[w]);
    for (int j = threadIdx.x; j < n; j += 1024) ex[j] = expf(lg[j] - mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += ex[j];
        lse_s = mx + logf(se);
    }
    __syncthreads();
    const long lab64 = (long)labels[b];
    const int lab = (lab64 >= 0 && lab64 < n) ? (int)lab64 : -1;
    for (int j = threadIdx.x; j < n; j += 1024) dl[j] = scale * (expf(lg[j] - lse_s) - (j == lab ? 1.f : 0.f)) / (float)B;
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 1024) {
        const float* w = Wp + i;
        float s2 = 0.f;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = w[(size_t)(j + u) * ldw];
#pragma unroll
            for (int u = 0; u < 8; ++u) s2 += dl[j + u] * q[u];
        }
        for (; j < n; ++j) s2 += dl[j] * w[(size_t)j * ldw];
        df[(size_t)b * D + i] = s2;
    }
}
int head_uni_dfeat(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale, float* df, int B,
                   int n, int width, hipStream_t st) {
    GDL_REQUIRE(n <= 512, "head_uni_dfeat: at most 512 classes");
    GDL_REQUIRE(width == 512 || width == 768 || width == 1024, "head_uni_dfeat: feature width %d (512, 768 or 1024)", width);
    ProfScope prof("gdl::head_uni_dfeat_kernel", PROF_HBM, st, (double)B * width * 8.0 + (double)n * width * 4.0);
    if (width == 512)
        hipLaunchKernelGGL(head_uni_dfeat_kernel<8>, dim3(B), dim3(1024), 0, st, f, Wp, ldw, bp, labels, scale, df, B, n);
    else if (width == 768)
        hipLaunchKernelGGL(head_uni_dfeat_kernel<12>, dim3(B), dim3(1024), 0, st, f, Wp, ldw, bp, labels, scale, df, B, n);
    else
        hipLaunchKernelGGL(head_uni_dfeat_kernel<16>, dim3(B), dim3(1024), 0, st, f, Wp, ldw, bp, labels, scale, df, B, n);
    GDL_CHECK_LAUNCH("head_uni_dfeat_kernel");
    return GDL_OK;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// grid = (B, 2): which = 0 -> hx = x W1^T + b1, 1 -> hy = y W2^T + b2.  Waves own outputs j = wave, wave+4, ..
__global__ __launch_bounds__(256) void gated_hidden_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           float* __restrict__ hx, float* __restrict__ hy) {
    const int b = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* in = (which ? y : x) + (size_t)b * HD;
    const float* W = which ? W2 : W1;
    const float* bias = which ? b2 : b1;
    float* h = (which ? hy : hx) + (size_t)b * HD;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in[lane + 64 * i];
    for (int j = wave; j < HD; j += 4) {
        const float* w = W + (size_t)j * HD;
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) p += w[lane + 64 * i] * v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
        if (lane == 0) h[j] = p + bias[j];
    }
}
// grid = B: the three logit sets from the hidden vectors
__global__ __launch_bounds__(256) void gated_out_kernel(const float* __restrict__ hx, const float* __restrict__ hy,
                                                        const float* __restrict__ Wo, const float* __restrict__ bo,
                                                        float* __restrict__ out, float* __restrict__ x_out,
                                                        float* __restrict__ y_out, int n) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sx[8], sy[8], gz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = hx[(size_t)b * HD + lane + 64 * i], c = hy[(size_t)b * HD + lane + 64 * i];
        const float ga = sigmoidf_(a);
        sx[i] = ga * a;
        sy[i] = sigmoidf_(c) * c;
        gz[i] = ga * c;
    }
    for (int j = wave; j < n; j += 4) {
        const float* w = Wo + (size_t)j * HD;
        float px = 0.f, py = 0.f, pz = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float wv = w[lane + 64 * i];
            px += wv * sx[i];
            py += wv * sy[i];
            pz += wv * gz[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            px += __shfl_xor(px, o);
            py += __shfl_xor(py, o);
            pz += __shfl_xor(pz, o);
        }
        if (lane == 0) {
            const float bj = bo[j];
            out[(size_t)b * n + j] = pz + bj;
            if (x_out) x_out[(size_t)b * n + j] = px + bj;
            if (y_out) y_out[(size_t)b * n + j] = py + bj;
        }
    }
}
int head_gated_fwd(const float* x, const float* y, const float* W1, const float* b1, const float* W2, const float* b2,
                   const float* Wo, const float* bo, float* hx, float* hy, float* out, float* x_out, float* y_out, int B, int n,
                   hipStream_t st) {
    hipLaunchKernelGGL(gated_hidden_kernel, dim3(B, 2), dim3(256), 0, st, x, y, W1, b1, W2, b2, hx, hy);
    GDL_CHECK_LAUNCH("gated_hidden_kernel");
    hipLaunchKernelGGL(gated_out_kernel, dim3(B), dim3(256), 0, st, hx, hy, Wo, bo, out, x_out, y_out, n);
    GDL_CHECK_LAUNCH("gated_out_kernel");
    return GDL_OK;
}

// grid = (B, 2): d_h[b][j] = (sum_c g[b][c] Wo[c][j]) * swish'(h[b][j]),  swish'(h) = s (1 + h (1 - s)), s = sigmoid(h)
__global__ __launch_bounds__(256) void gated_dh_kernel(const float* __restrict__ hx, const float* __restrict__ hy,
                                                       const float* __restrict__ Wo, const float* __restrict__ g_x_out,
                                                       const float* __restrict__ g_y_out, float* __restrict__ dhx,
                                                       float* __restrict__ dhy, int n) {
    const int b = blockIdx.x, which = blockIdx.y;
    const float* g = which ? g_y_out : g_x_out;
    const float* h = (which ? hy : hx) + (size_t)b * HD;
    float* dh = (which ? dhy : dhx) + (size_t)b * HD;
    for (int j = threadIdx.x; j < HD; j += 256) {
        float s = 0.f;
        if (g)
            for (int c = 0; c < n; ++c) s += g[(size_t)b * n + c] * Wo[(size_t)c * HD + j];
        const float hv = h[j], sg = sigmoidf_(hv);
        dh[j] = s * sg * (1.f + hv * (1.f - sg));
    }
}
// grid = (B, 2): dx[b][i] = sum_j dhx[b][j] W1[j][i]   (threads along i: coalesced rows of W1)
__global__ __launch_bounds__(256) void gated_dx_kernel(const float* __restrict__ dhx, const float* __restrict__ dhy,
                                                       const float* __restrict__ W1, const float* __restrict__ W2,
                                                       float* __restrict__ dx, float* __restrict__ dy) {
    __shared__ float dh[HD];
    const int b = blockIdx.x, which = blockIdx.y;
    const float* src = (which ? dhy : dhx) + (size_t)b * HD;
    const float* W = which ? W2 : W1;
    float* dst = (which ? dy : dx) + (size_t)b * HD;
    for (int j = threadIdx.x; j < HD; j += 256) dh[j] = src[j];
    __syncthreads();
    for (int i = threadIdx.x; i < HD; i += 256) {
        float s = 0.f;
        for (int j = 0; j < HD; ++j) s += dh[j] * W[(size_t)j * HD + i];
        dst[i] = s;
    }
}
// grid = n: dWo[c][j] = sum_b ( g_out[b][c] * sig(hx)*hy  +  uni * (g_x_out[b][c] * swish(hx) + g_y_out[b][c] * swish(hy)) )
__global__ __launch_bounds__(256) void gated_dwo_kernel(const float* __restrict__ hx, const float* __restrict__ hy,
                                                        const float* __restrict__ g_x_out, const float* __restrict__ g_y_out,
                                                        const float* __restrict__ g_out, int uni_in_dw,
                                                        float* __restrict__ dWo, float* __restrict__ dbo, int B, int n) {
    const int c = blockIdx.x;
    for (int j = threadIdx.x; j < HD; j += 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float a = hx[(size_t)b * HD + j], v = hy[(size_t)b * HD + j];
            const float ga = sigmoidf_(a);
            if (g_out) s += g_out[(size_t)b * n + c] * ga * v;
            if (uni_in_dw) {
                if (g_x_out) s += g_x_out[(size_t)b * n + c] * ga * a;
                if (g_y_out) s += g_y_out[(size_t)b * n + c] * sigmoidf_(v) * v;
            }
        }
        dWo[(size_t)c * HD + j] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            if (g_out) s += g_out[(size_t)b * n + c];
            if (uni_in_dw) {
                if (g_x_out) s += g_x_out[(size_t)b * n + c];
                if (g_y_out) s += g_y_out[(size_t)b * n + c];
            }
        }
        dbo[c] = s;
    }
}
// grid = (512, 2): dW1[j][i] = sum_b dhx[b][j] x[b][i], db1[j] = sum_b dhx[b][j]   (plain-autograd callers only)
__global__ __launch_bounds__(256) void gated_dw1_kernel(const float* __restrict__ dhx, const float* __restrict__ dhy,
                                                        const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ dW1, float* __restrict__ db1,
                                                        float* __restrict__ dW2, float* __restrict__ db2, int B) {
    const int j = blockIdx.x, which = blockIdx.y;
    const float* dh = which ? dhy : dhx;
    const float* in = which ? y : x;
    float* dW = which ? dW2 : dW1;
    float* db = which ? db2 : db1;
    for (int i = threadIdx.x; i < HD; i += 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dh[(size_t)b * HD + j] * in[(size_t)b * HD + i];
        dW[(size_t)j * HD + i] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dh[(size_t)b * HD + j];
        db[j] = s;
    }
}
// ws: 2 * B * 512 floats (d hx, d hy)
int head_gated_bwd(const float* x, const float* y, const float* hx, const float* hy, const float* W1, const float* W2,
                   const float* Wo, const float* g_x_out, const float* g_y_out, const float* g_out, int uni_in_dw, float* dx,
                   float* dy, float* dW1, float* db1, float* dW2, float* db2, float* dWo, float* dbo, float* ws, int B, int n,
                   hipStream_t st) {
    float *dhx = ws, *dhy = ws + (size_t)B * HD;
    if ((dx && dy) || dW1) {
        hipLaunchKernelGGL(gated_dh_kernel, dim3(B, 2), dim3(256), 0, st, hx, hy, Wo, g_x_out, g_y_out, dhx, dhy, n);
        GDL_CHECK_LAUNCH("gated_dh_kernel");
    }
    if (dx && dy) {
        hipLaunchKernelGGL(gated_dx_kernel, dim3(B, 2), dim3(256), 0, st, dhx, dhy, W1, W2, dx, dy);
        GDL_CHECK_LAUNCH("gated_dx_kernel");
    }
    if (dW1) {
        hipLaunchKernelGGL(gated_dw1_kernel, dim3(HD, 2), dim3(256), 0, st, dhx, dhy, x, y, dW1, db1, dW2, db2, B);
        GDL_CHECK_LAUNCH("gated_dw1_kernel");
    }
    if (dWo && dbo) {
        hipLaunchKernelGGL(gated_dwo_kernel, dim3(n), dim3(256), 0, st, hx, hy, g_x_out, g_y_out, g_out, uni_in_dw, dWo, dbo, B, n);
        GDL_CHECK_LAUNCH("gated_dwo_kernel");
    }
    return GDL_OK;
}

// valid() of /root/reference/main_dgl.py:206-219 without its per-sample host loop: for every sample the
// arg-max of the three logit sets (softmax is monotone, np.argmax takes the first maximum) is compared with
// the label and four per-class counters are bumped: num[label]++, acc*[label] += (argmax == label).
// Integer atomics: the result does not depend on the order.
__global__ __launch_bounds__(256) void eval_count_kernel(const float* __restrict__ out, const float* __restrict__ out_a,
                                                         const float* __restrict__ out_v, const int64_t* __restrict__ labels,
                                                         int B, int n, unsigned long long* num, unsigned long long* acc,
                                                         unsigned long long* acc_a, unsigned long long* acc_v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int lab = (int)labels[i];
    if (lab < 0 || lab >= n) return;  // (the reference would raise an IndexError)
    const float* rows[3] = {out + (size_t)i * n, out_a ? out_a + (size_t)i * n : nullptr, out_v ? out_v + (size_t)i * n : nullptr};
    unsigned long long* cnt[3] = {acc, acc_a, acc_v};
    atomicAdd(&num[lab], 1ULL);
    for (int h = 0; h < 3; ++h) {
        if (!rows[h] || !cnt[h]) continue;
        int best = 0;
        float bv = rows[h][0];
        for (int c = 1; c < n; ++c)
            if (rows[h][c] > bv) {  // strict: first maximum wins, like np.argmax
                bv = rows[h][c];
                best = c;
            }
        if (best == lab) atomicAdd(&cnt[h][lab], 1ULL);
    }
}
int eval_count(const float* out, const float* out_a, const float* out_v, const int64_t* labels, int B, int n, int64_t* num,
               int64_t* acc, int64_t* acc_a, int64_t* acc_v, hipStream_t st) {
    hipLaunchKernelGGL(eval_count_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, out, out_a, out_v, labels, B, n,
                       (unsigned long long*)num, (unsigned long long*)acc, (unsigned long long*)acc_a,
                       (unsigned long long*)acc_v);
    GDL_CHECK_LAUNCH("eval_count_kernel");
    return GDL_OK;
}

// one block per loss: the three cross-entropies of the DGL step (main_dgl.py:102-104) are ONE launch
struct CeSets {
    const float* logits[4];
    float* dlogits[4];
    float scale[4];
};
__global__ __launch_bounds__(1024) void softmax_ce_kernel(CeSets s, const int64_t* __restrict__ labels, float* __restrict__ loss,
                                                         int B, int n) {
    const int k = blockIdx.x;
    softmax_ce_block(s.logits[k], labels, s.scale[k], loss + k, s.dlogits[k], B, n);
}
int softmax_ce_multi(int nsets, const float* const* logits, const int64_t* labels, const float* scales, float* losses,
                     float* const* dlogits, int B, int n, hipStream_t st) {
    CeSets s{};
    for (int k = 0; k < nsets; ++k) {
        s.logits[k] = logits[k];
        s.dlogits[k] = dlogits ? dlogits[k] : nullptr;
        s.scale[k] = scales[k];
    }
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(nsets), dim3(n >= 64 && n <= CE_MAXN ? 1024 : 256), 0, st, s, labels, losses, B, n);
    GDL_CHECK_LAUNCH("softmax_ce_kernel");
    return GDL_OK;
}
int softmax_ce(const float* logits, const int64_t* labels, float scale, float* loss, float* dlogits, int B, int n,
               hipStream_t st) {
    return softmax_ce_multi(1, &logits, labels, &scale, loss, &dlogits, B, n, st);
}

}  // namespace gdl
