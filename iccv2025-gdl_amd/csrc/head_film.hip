// head_film.hip -- FiLM_DGL fusion head (/root/reference/models/fusion_modules.py:126-178), float32.
//
//   fc     : Linear(512*512, 512)   (134 M parameters: W[k][i*512 + j], the k-th row is a 512x512 matrix W_k)
//   fc_out : Linear(512, n)
//   output = fc_out(fc(flatten(x.detach() (x) y.detach())))     h_f[b][k] = x_b^T W_k y_b + bias_k
//   z_x    = fc_out(fc(flatten(x (x) x)))                       h_x[b][k] = x_b^T W_k x_b + bias_k
//   z_y    = fc_out(fc(flatten(y (x) y)))                       h_y[b][k] = y_b^T W_k y_b + bias_k
// The outer products are never materialised.  With fc.weight viewed as a [262144][512] row-major matrix
// A (row (k,i), column j) the heavy contractions are plain GEMMs over its 537 MB, and they run on the
// convolution kernels of this library in their exact-f32 mode (f32-input MFMA):
//   forward    T = A V^T,  V = [x; y] (2B rows)                      -> conv_fwd (1x1), T[(k,i)][b] = (W_k v_b)[i]
//              h_x[b][k] = sum_i x_b[i] T[(k,i)][b],  h_f / h_y from the y-columns of T
//   d x, d y   dx_b[i] = sum_k dh_x[b][k] ( (W_k x_b)[i] + (W_k^T x_b)[i] )
//              first term from T; second = U^T A with U[(k,j)][b] = dh_x[b][k] x_b[j]   -> conv_wgrad (1x1)
//   d fc.weight  dA = P Z^T, P[(k,i)][b] = dh[b][k] * (x|y)_b[i], Z = matching (y|x) rows -> conv_fwd (1x1)
// Everything else (V, U, P, the small contractions, fc_out and its gradients) is elementwise / tiny.
#include "common.h"
#include "gather.h"
#include "ops.h"

namespace gdl {

constexpr int FD = 512;            // feature width of both modalities and of fc's output
constexpr int FROWS = FD * FD;     // rows of fc.weight viewed as [FROWS][FD]

// the operand matrices have 2 Bp .. 3 Bp columns of FROWS rows (3 Bp: 1.6 GB at the limit), every index is size_t
constexpr int FILM_MAX_B = 512;
static inline int film_bp(int B) { return (B + 31) / 32 * 32; }  // batch padded so that 2*Bp % 64 == 0, 3*Bp % 32 == 0

// ---- V[2*Bp][512]: rows 0..B-1 = x, rows Bp..Bp+B-1 = y, padding rows zero
__global__ void film_v_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ V, int B, int Bp) {
    const int r = blockIdx.x;  // 0 .. 2*Bp-1
    const int b = r < Bp ? r : r - Bp;
    const float* src = b < B ? (r < Bp ? x : y) + (size_t)b * FD : nullptr;
    for (int i = threadIdx.x; i < FD; i += blockDim.x) V[(size_t)r * FD + i] = src ? src[i] : 0.f;
}

// ---- h[b][k] = bias[k] + sum_i left_b[i] * T[(k,i)][col(b)]     grid = (512 k, 3 forms, ceil(B / 64) sample groups):
// 0 = x form, 1 = fused, 2 = y form
__global__ __launch_bounds__(256) void film_h_kernel(const float* __restrict__ T, const float* __restrict__ x,
                                                     const float* __restrict__ y, const float* __restrict__ bias,
                                                     float* __restrict__ hx, float* __restrict__ hf, float* __restrict__ hy,
                                                     int B, int Bp) {
    __shared__ float red[4][64];
    const int k = blockIdx.x, form = blockIdx.y;
    const float* left = form == 2 ? y : x;                 // x^T W_k x, x^T W_k y, y^T W_k y
    const int coff = form == 0 ? 0 : Bp;                   // T columns: W_k x_b at b, W_k y_b at Bp + b
    float* h = form == 0 ? hx : (form == 1 ? hf : hy);
    const int bl = threadIdx.x & 63, b = blockIdx.z * 64 + bl, part = threadIdx.x >> 6;  // 4 slices of i
    float s = 0.f;
    if (b < B) {
        const float* Tk = T + (size_t)k * FD * (2 * Bp) + coff + b;
        for (int i = part; i < FD; i += 4) s += left[(size_t)b * FD + i] * Tk[(size_t)i * (2 * Bp)];
    }
    red[part][bl] = s;
    __syncthreads();
    if (part == 0 && b < B) h[(size_t)b * FD + k] = ((red[0][bl] + red[1][bl]) + red[2][bl]) + red[3][bl] + bias[k];
}

// ---- logits of the three hidden vectors: grid = B
__global__ __launch_bounds__(256) void film_out_kernel(const float* __restrict__ hx, const float* __restrict__ hf,
                                                       const float* __restrict__ hy, const float* __restrict__ Wo,
                                                       const float* __restrict__ bo, float* __restrict__ out,
                                                       float* __restrict__ x_out, float* __restrict__ y_out, int n) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8], f[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = hx[(size_t)b * FD + lane + 64 * i];
        f[i] = hf[(size_t)b * FD + lane + 64 * i];
        c[i] = hy[(size_t)b * FD + lane + 64 * i];
    }
    for (int j = wave; j < n; j += 4) {
        const float* w = Wo + (size_t)j * FD;
        float px = 0.f, pf = 0.f, py = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float wv = w[lane + 64 * i];
            px += wv * a[i];
            pf += wv * f[i];
            py += wv * c[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            px += __shfl_xor(px, o);
            pf += __shfl_xor(pf, o);
            py += __shfl_xor(py, o);
        }
        if (lane == 0) {
            out[(size_t)b * n + j] = pf + bo[j];
            if (x_out) x_out[(size_t)b * n + j] = px + bo[j];
            if (y_out) y_out[(size_t)b * n + j] = py + bo[j];
        }
    }
}

// ---- dh[form][b][k] = sum_c g_form[b][c] Wo[c][k]   (zero where the upstream gradient is absent)   grid = (B, 3)
__global__ __launch_bounds__(256) void film_dh_kernel(const float* __restrict__ g_x_out, const float* __restrict__ g_out,
                                                      const float* __restrict__ g_y_out, const float* __restrict__ Wo,
                                                      float* __restrict__ dh, int B, int n) {
    const int b = blockIdx.x, form = blockIdx.y;
    const float* g = form == 0 ? g_x_out : (form == 1 ? g_out : g_y_out);
    for (int k = threadIdx.x; k < FD; k += 256) {
        float s = 0.f;
        if (g)
            for (int c = 0; c < n; ++c) s += g[(size_t)b * n + c] * Wo[(size_t)c * FD + k];
        dh[((size_t)form * B + b) * FD + k] = s;
    }
}

// ---- dWo[c][k] = sum_b ( g_out h_f + uni (g_x_out h_x + g_y_out h_y) ), dbo likewise    grid = n
__global__ __launch_bounds__(256) void film_dwo_kernel(const float* __restrict__ hx, const float* __restrict__ hf,
                                                       const float* __restrict__ hy, const float* __restrict__ g_x_out,
                                                       const float* __restrict__ g_out, const float* __restrict__ g_y_out,
                                                       int uni, float* __restrict__ dWo, float* __restrict__ dbo, int B, int n) {
    const int c = blockIdx.x;
    for (int k = threadIdx.x; k < FD; k += 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            if (g_out) s += g_out[(size_t)b * n + c] * hf[(size_t)b * FD + k];
            if (uni) {
                if (g_x_out) s += g_x_out[(size_t)b * n + c] * hx[(size_t)b * FD + k];
                if (g_y_out) s += g_y_out[(size_t)b * n + c] * hy[(size_t)b * FD + k];
            }
        }
        dWo[(size_t)c * FD + k] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            if (g_out) s += g_out[(size_t)b * n + c];
            if (uni) {
                if (g_x_out) s += g_x_out[(size_t)b * n + c];
                if (g_y_out) s += g_y_out[(size_t)b * n + c];
            }
        }
        dbo[c] = s;
    }
}

// ---- U[(k,j)][col]: col b -> dh_x[b][k] * x_b[j], col Bp+b -> dh_y[b][k] * y_b[j], padding columns zero.
// one thread per (row, 4 columns)
__global__ __launch_bounds__(256) void film_u_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                     const float* __restrict__ y, float* __restrict__ U, int B, int Bp) {
    const int cols = 2 * Bp, c4 = cols / 4;
    const size_t total = (size_t)FROWS * c4;
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int q = (int)(t % c4);
        const size_t row = t / c4;
        const int k = (int)(row / FD), j = (int)(row % FD);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = q * 4 + e, isy = col >= Bp, b = isy ? col - Bp : col;
            v[e] = b < B ? dh[((size_t)(isy ? 2 : 0) * B + b) * FD + k] * (isy ? y : x)[(size_t)b * FD + j] : 0.f;
        }
        *(float4*)(U + row * cols + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- dx[b][i] = out2[b][i] + sum_k dh_x[b][k] T[(k,i)][b];  dy from the y columns     grid = (B, 2)
__global__ __launch_bounds__(256) void film_dxy_kernel(const float* __restrict__ T, const float* __restrict__ dh,
                                                       const float* __restrict__ out2, float* __restrict__ dx,
                                                       float* __restrict__ dy, int B, int Bp) {
    const int b = blockIdx.x, isy = blockIdx.y;
    const float* dhb = dh + ((size_t)(isy ? 2 : 0) * B + b) * FD;
    const int col = (isy ? Bp : 0) + b;
    float* dst = (isy ? dy : dx) + (size_t)b * FD;
    for (int i = threadIdx.x; i < FD; i += 256) {
        float s = out2[(size_t)col * FD + i];
        for (int k = 0; k < FD; ++k) s += dhb[k] * T[((size_t)k * FD + i) * (2 * Bp) + col];
        dst[i] = s;
    }
}

// ---- operands of the fc.weight gradient GEMM: P[(k,i)][c], Z[j][c], c in blocks of Bp columns:
//   block 0: dh_f[b][k] * x_b[i]  with  Z = y_b[j];   (uni) block 1: dh_x[b][k] * x_b[i] with x_b[j];  block 2: dh_y[b][k] * y_b[i] with y_b[j]
__global__ __launch_bounds__(256) void film_p_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                     const float* __restrict__ y, float* __restrict__ P, int B, int Bp,
                                                     int nblk) {
    const int cols = nblk * Bp, c4 = cols / 4;
    const size_t total = (size_t)FROWS * c4;
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int q = (int)(t % c4);
        const size_t row = t / c4;
        const int k = (int)(row / FD), i = (int)(row % FD);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = q * 4 + e, blk = col / Bp, b = col - blk * Bp;
            const int form = blk == 0 ? 1 : (blk == 1 ? 0 : 2);  // dh index: 0 = x form, 1 = fused, 2 = y form
            v[e] = b < B ? dh[((size_t)form * B + b) * FD + k] * (blk == 2 ? y : x)[(size_t)b * FD + i] : 0.f;
        }
        *(float4*)(P + row * cols + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
__global__ void film_z_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ Z, int B, int Bp,
                              int nblk) {
    const int j = blockIdx.x, cols = nblk * Bp;  // Z[j][c]
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        const int blk = c / Bp, b = c - blk * Bp;
        Z[(size_t)j * cols + c] = b < B ? (blk == 1 ? x : y)[(size_t)b * FD + j] : 0.f;
    }
}
// dbfc[k] = sum_b ( dh_f + uni (dh_x + dh_y) )
__global__ void film_dbfc_kernel(const float* __restrict__ dh, int uni, float* __restrict__ dbfc, int B) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= FD) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        s += dh[((size_t)1 * B + b) * FD + k];
        if (uni) s += dh[((size_t)0 * B + b) * FD + k] + dh[((size_t)2 * B + b) * FD + k];
    }
    dbfc[k] = s;
}

// ---------------------------------------------------------------- workspace layout (bytes, 256-aligned pieces)
struct FilmWs {
    float *V, *T, *dh, *U, *out2, *P, *Z;
    void *tab_a, *tab_p, *wg;
    size_t wg_bytes, total;
};
static FilmWs film_layout(unsigned char* base, int B) {
    const int Bp = film_bp(B);
    FilmWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return p;
    };
    w.V = (float*)take((size_t)2 * Bp * FD * 4);
    w.T = (float*)take((size_t)FROWS * 2 * Bp * 4);
    w.dh = (float*)take((size_t)3 * B * FD * 4);
    w.U = (float*)take((size_t)FROWS * 2 * Bp * 4);
    w.out2 = (float*)take((size_t)2 * Bp * FD * 4);
    w.P = (float*)take((size_t)FROWS * 3 * Bp * 4);
    w.Z = (float*)take((size_t)FD * 3 * Bp * 4);
    w.tab_a = take(gather_table_bytes(GATHER_FWD, 1, FROWS, 1, 1, 1, 1, 0));
    w.tab_p = take(gather_table_bytes(GATHER_FWD, 1, FROWS, 1, 1, 1, 1, 0));
    w.wg_bytes = conv_wgrad_ws_bytes(FROWS, FD, 2 * Bp, 1);
    w.wg = take(w.wg_bytes);
    w.total = off;
    return w;
}
size_t head_film_ws_bytes(int B) { return (B >= 1 && B <= FILM_MAX_B) ? film_layout(nullptr, B).total : 0; }

// hidden: [3][B][512] = h_x, h_f, h_y (kept by the caller for the backward, together with ws: T is reused)
int head_film_fwd(const float* x, const float* y, const float* Wfc, const float* bfc, const float* Wo, const float* bo,
                  float* hidden, float* out, float* x_out, float* y_out, int B, int n, void* ws, size_t ws_bytes, hipStream_t st) {
    GDL_REQUIRE(B >= 1 && B <= FILM_MAX_B, "head_film: B=%d (1..%d per call)", B, FILM_MAX_B);
    const FilmWs w = film_layout((unsigned char*)ws, B);
    if (!ws || ws_bytes < w.total) {
        set_error("head_film_fwd: workspace %zu < %zu bytes", ws_bytes, w.total);
        return GDL_ERR_WORKSPACE;
    }
    const int Bp = film_bp(B);
    float *hx = hidden, *hf = hidden + (size_t)B * FD, *hy = hidden + (size_t)2 * B * FD;
    hipLaunchKernelGGL(film_v_kernel, dim3(2 * Bp), dim3(256), 0, st, x, y, w.V, B, Bp);
    GDL_CHECK_LAUNCH("film_v_kernel");
    int rc = build_gather_table(GATHER_FWD, GDL_F32, 1, FROWS, 1, FD, 2 * Bp, 1, 1, 1, 0, (GatherEntry*)w.tab_a, st);
    if (rc) return rc;
    // T[(k,i)][col] = sum_j A[(k,i)][j] V[col][j]
    rc = conv_fwd(GDL_F32, Wfc, w.V, w.T, nullptr, w.tab_a, 1, FROWS, 1, FD, 2 * Bp, 1, 1, 1, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(film_h_kernel, dim3(FD, 3, (B + 63) / 64), dim3(256), 0, st, w.T, x, y, bfc, hx, hf, hy, B, Bp);
    GDL_CHECK_LAUNCH("film_h_kernel");
    hipLaunchKernelGGL(film_out_kernel, dim3(B), dim3(256), 0, st, hx, hf, hy, Wo, bo, out, x_out, y_out, n);
    GDL_CHECK_LAUNCH("film_out_kernel");
    return GDL_OK;
}

// Needs the workspace of the matching forward untouched (T).  Any upstream gradient may be NULL.
// dx/dy (pair), dWo/dbo (pair), dWfc/dbfc (pair) may be NULL.  uni: the unimodal logit sets also contribute to the
// parameter gradients (plain autograd); 0 in the DGL step (main_dgl.py:114-122 drops them).
int head_film_bwd(const float* x, const float* y, const float* Wfc, const float* Wo, const float* hidden,
                  const float* g_x_out, const float* g_y_out, const float* g_out, int uni, float* dx, float* dy, float* dWfc,
                  float* dbfc, float* dWo, float* dbo, int B, int n, void* ws, size_t ws_bytes, hipStream_t st) {
    GDL_REQUIRE(B >= 1 && B <= FILM_MAX_B, "head_film: B=%d (1..%d per call)", B, FILM_MAX_B);
    const FilmWs w = film_layout((unsigned char*)ws, B);
    if (!ws || ws_bytes < w.total) {
        set_error("head_film_bwd: workspace %zu < %zu bytes", ws_bytes, w.total);
        return GDL_ERR_WORKSPACE;
    }
    const int Bp = film_bp(B);
    const float *hx = hidden, *hf = hidden + (size_t)B * FD, *hy = hidden + (size_t)2 * B * FD;
    hipLaunchKernelGGL(film_dh_kernel, dim3(B, 3), dim3(256), 0, st, g_x_out, g_out, g_y_out, Wo, w.dh, B, n);
    GDL_CHECK_LAUNCH("film_dh_kernel");
    if (dWo && dbo) {
        hipLaunchKernelGGL(film_dwo_kernel, dim3(n), dim3(256), 0, st, hx, hf, hy, g_x_out, g_out, g_y_out, uni, dWo, dbo, B, n);
        GDL_CHECK_LAUNCH("film_dwo_kernel");
    }
    int rc;
    if (dx && dy) {
        hipLaunchKernelGGL(film_u_kernel, dim3(4096), dim3(256), 0, st, w.dh, x, y, w.U, B, Bp);
        GDL_CHECK_LAUNCH("film_u_kernel");
        // out2[col][i] = sum_(k,j) U[(k,j)][col] A[(k,j)][i]
        rc = conv_wgrad(GDL_F32, w.U, Wfc, w.out2, w.tab_a, 1, FROWS, 1, FD, 2 * Bp, 1, 1, 1, 0, FD, w.wg, w.wg_bytes, st);
        if (rc) return rc;
        hipLaunchKernelGGL(film_dxy_kernel, dim3(B, 2), dim3(256), 0, st, w.T, w.dh, w.out2, dx, dy, B, Bp);
        GDL_CHECK_LAUNCH("film_dxy_kernel");
    }
    if (dWfc && dbfc) {
        const int nblk = uni ? 3 : 1;
        hipLaunchKernelGGL(film_p_kernel, dim3(4096), dim3(256), 0, st, w.dh, x, y, w.P, B, Bp, nblk);
        GDL_CHECK_LAUNCH("film_p_kernel");
        hipLaunchKernelGGL(film_z_kernel, dim3(FD), dim3(128), 0, st, x, y, w.Z, B, Bp, nblk);
        GDL_CHECK_LAUNCH("film_z_kernel");
        rc = build_gather_table(GATHER_FWD, GDL_F32, 1, FROWS, 1, nblk * Bp, FD, 1, 1, 1, 0, (GatherEntry*)w.tab_p, st);
        if (rc) return rc;
        // dA[(k,i)][j] = sum_c P[(k,i)][c] Z[j][c]
        rc = conv_fwd(GDL_F32, w.P, w.Z, dWfc, nullptr, w.tab_p, 1, FROWS, 1, nblk * Bp, FD, 1, 1, 1, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(film_dbfc_kernel, dim3(2), dim3(256), 0, st, w.dh, uni, dbfc, B);
        GDL_CHECK_LAUNCH("film_dbfc_kernel");
    }
    return GDL_OK;
}

}  // namespace gdl
