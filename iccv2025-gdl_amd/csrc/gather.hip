// gather.hip -- builds the per-pixel gather tables described in gather.h (one tiny kernel per
// convolution geometry, run once when the encoder engine binds its workspace).
#include "gather.h"

namespace gdl {

int gather_geom(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, GatherGeom* g) {
    GDL_REQUIRE(R * S <= 9 && R >= 1 && S >= 1, "gather: %dx%d taps unsupported", R, S);
    GDL_REQUIRE(stride == 1 || stride == 2, "gather: stride %d unsupported", stride);
    const int esz = dtype == GDL_BF16 ? 2 : 4;
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    g->ntaps = R * S;
    for (int t = 0; t < 9; ++t) g->delta[t] = 0;
    if (mode == GATHER_FWD) {
        g->rows = N * P * Q;
        g->row_bytes = C * esz;
        GDL_REQUIRE((size_t)N * H * W * g->row_bytes < (1UL << 31), "gather: source tensor exceeds 2 GiB");
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) g->delta[r * S + s] = (r * W + s) * g->row_bytes;
    } else {
        g->rows = N * H * W;
        g->row_bytes = K * esz;
        GDL_REQUIRE((size_t)N * P * Q * g->row_bytes < (1UL << 31), "gather: source tensor exceeds 2 GiB");
        const int sh = stride == 2 ? 1 : 0;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) g->delta[r * S + s] = -(((r >> sh) * Q + (s >> sh)) * g->row_bytes);
    }
    return GDL_OK;
}

__global__ void gather_table_kernel(GatherEntry* __restrict__ tab, int mode, int rows, int IH, int IW, int OH, int OW,
                                    int R, int S, int stride, int pad, int row_bytes) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= rows) return;
    const int n = m / (OH * OW);
    const int rem = m - n * OH * OW;
    const int oh = rem / OW, ow = rem - oh * OW;
    GatherEntry e;
    e.mask = 0;
    if (mode == GATHER_FWD) {
        const int hs = oh * stride - pad, ws = ow * stride - pad;
        e.off0 = ((n * IH + hs) * IW + ws) * row_bytes;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s)
                if ((unsigned)(hs + r) < (unsigned)IH && (unsigned)(ws + s) < (unsigned)IW) e.mask |= 1u << (r * S + s);
    } else {
        // rows are INPUT pixels (oh, ow) = (h, w); source is dy with IH x IW = P x Q
        const int hs = oh + pad, ws = ow + pad, sh = stride == 2 ? 1 : 0;
        e.off0 = ((n * IH + (hs >> sh)) * IW + (ws >> sh)) * row_bytes;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) {
                const int th = hs - r, tw = ws - s;
                if (th < 0 || tw < 0) continue;
                if (sh && ((th | tw) & 1)) continue;
                if ((th >> sh) < IH && (tw >> sh) < IW) e.mask |= 1u << (r * S + s);
            }
    }
    tab[m] = e;
}

int conv_dgrad_bm(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);  // conv_igemm.hip

int build_gather_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                       GatherEntry* table, hipStream_t st) {
    if (mode == GATHER_DGRAD && stride == 2)
        return build_dgrad_perm_table(dtype, N, H, W, C, K, R, S, pad, conv_dgrad_bm(dtype, N, H, W, C, K, R, S, stride, pad),
                                      table, st);
    GatherGeom g;
    int rc = gather_geom(mode, dtype, N, H, W, C, K, R, S, stride, pad, &g);
    if (rc) return rc;
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    const int grid = ceil_div(g.rows, 256);
    if (mode == GATHER_FWD)
        hipLaunchKernelGGL(gather_table_kernel, dim3(grid), dim3(256), 0, st, table, mode, g.rows, H, W, P, Q, R, S, stride,
                           pad, g.row_bytes);
    else
        hipLaunchKernelGGL(gather_table_kernel, dim3(grid), dim3(256), 0, st, table, mode, g.rows, P, Q, H, W, R, S, stride,
                           pad, g.row_bytes);
    GDL_CHECK_LAUNCH("gather_table_kernel");
    return GDL_OK;
}

// ---- direct-stem table: output pixel (n,p,q) -> padded input pixel (n, 2p, 2q); every K-step valid
__global__ void stem_table_kernel(GatherEntry* __restrict__ tab, int rows, int P, int Q, int Hp, int Wp, int pixbytes,
                                  unsigned mask) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= rows) return;
    const int n = m / (P * Q), rem = m - n * P * Q;
    const int p = rem / Q, q = rem - p * Q;
    GatherEntry e;
    e.off0 = ((n * Hp + 2 * p) * Wp + 2 * q) * pixbytes;
    e.mask = mask;
    tab[m] = e;
}
int build_stem_table(int dtype, int n_img, int H, int W, int ntaps, GatherEntry* table, hipStream_t st) {
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1, Hp = H + 6, Wp = W + 8;
    const int pixbytes = 4 * (dtype == GDL_BF16 ? 2 : 4);
    GDL_REQUIRE((size_t)n_img * Hp * Wp * pixbytes < (1UL << 30), "stem table: padded input exceeds 1 GiB");
    const int rows = n_img * P * Q;
    hipLaunchKernelGGL(stem_table_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, st, table, rows, P, Q, Hp, Wp, pixbytes,
                       (1u << ntaps) - 1u);
    GDL_CHECK_LAUNCH("stem_table_kernel");
    return GDL_OK;
}

// ---- permuted stride-2 data-gradient table (see gather.h)
struct PermGeom {
    int seg[5];     // first GEMM row of class c = (h&1)*2 + (w&1); seg[4] = total rows
    int hc[2], wc[2];
};
static PermGeom perm_geom(int N, int H, int W, int bm) {
    PermGeom g;
    g.hc[0] = (H + 1) / 2, g.hc[1] = H / 2;
    g.wc[0] = (W + 1) / 2, g.wc[1] = W / 2;
    int at = 0;
    for (int c = 0; c < 4; ++c) {
        g.seg[c] = at;
        const int cnt = N * g.hc[c >> 1] * g.wc[c & 1];
        at += ceil_div(cnt, bm) * bm;
    }
    g.seg[4] = at;
    return g;
}
int dgrad_perm_rows(int N, int H, int W, int bm) { return perm_geom(N, H, W, bm).seg[4]; }

size_t gather_table_bytes(int mode, int N, int H, int W, int R, int S, int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    if (mode == GATHER_FWD) return (size_t)N * P * Q * sizeof(GatherEntry);
    if (stride == 2) return dgrad_perm_cap(N, H, W) * (sizeof(GatherEntry) + sizeof(int)) + (dgrad_perm_cap(N, H, W) / 64 + 1) * 8;
    return (size_t)N * H * W * sizeof(GatherEntry);
}

__global__ void dgrad_perm_table_kernel(GatherEntry* __restrict__ tab, int* __restrict__ orow, unsigned* __restrict__ ttaps,
                                        int bm, PermGeom pg, int N, int H, int W, int OH, int OW, int R, int S, int pad,
                                        int row_bytes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pg.seg[4]) return;
    int c = 0;
    while (c < 3 && j >= pg.seg[c + 1]) ++c;
    const int ch = c >> 1, cw = c & 1, hc = pg.hc[ch], wc = pg.wc[cw];
    const int q = j - pg.seg[c];
    GatherEntry e;
    e.off0 = 0;
    e.mask = 0;
    int o = -1;
    if (q < N * hc * wc) {
        const int n = q / (hc * wc), rem = q - n * hc * wc;
        const int h = 2 * (rem / wc) + ch, w = 2 * (rem % wc) + cw;
        o = (n * H + h) * W + w;
        // source dy is [N][OH][OW][K]; same arithmetic as gather_table_kernel's stride-2 branch
        const int hs = h + pad, ws = w + pad;
        e.off0 = ((n * OH + (hs >> 1)) * OW + (ws >> 1)) * row_bytes;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) {
                const int th = hs - r, tw = ws - s;
                if (th < 0 || tw < 0) continue;
                if ((th | tw) & 1) continue;
                if ((th >> 1) < OH && (tw >> 1) < OW) e.mask |= 1u << (r * S + s);
            }
    }
    tab[j] = e;
    orow[j] = o;
    if (e.mask) atomicOr(&ttaps[j / bm], e.mask);  // integer OR: order-independent
}

// Launch order of the class-pure tiles (round 3): order[slot] = tile.  The table is class-major -- all (even, even) pixels of
// the batch, then the next class -- and the kernel gives XCD x the slots [x, x + 1) * ceil(tiles / 8): in table order an XCD
// worked through ONE class of the WHOLE batch and pulled all of dy through its L2, four XCD pairs each (PMC: 123 MB read per
// launch against 60 MB of operands).  Here XCD x gets the x-th eighth of EVERY class, its tiles taken round-robin over the
// classes, so that the four classes of a group of images -- which gather the same dy pixels -- meet in one L2 close in time.
// Alone (tools/bench_conv.py, visual 64->128 / 128->256 / 256->512): 82 / 67 / 69 -> 74 / 57 / 51 us; the step does not move
// (5.629 vs 5.620 ms, four A/B rounds).
__global__ void dgrad_perm_order_kernel(int* __restrict__ order, PermGeom pg, int bm) {
    if (blockIdx.x || threadIdx.x) return;
    int t0[4], tc[4], at = 0;
    for (int c = 0; c < 4; ++c) {
        t0[c] = pg.seg[c] / bm;
        tc[c] = (pg.seg[c + 1] - pg.seg[c]) / bm;
    }
    for (int x = 0; x < 8; ++x) {
        int lo[4], n[4], most = 0;
        for (int c = 0; c < 4; ++c) {
            lo[c] = (int)((long long)x * tc[c] / 8);
            n[c] = (int)((long long)(x + 1) * tc[c] / 8) - lo[c];
            most = n[c] > most ? n[c] : most;
        }
        for (int k = 0; k < most; ++k)
            for (int c = 0; c < 4; ++c)
                if (k < n[c]) order[at++] = t0[c] + lo[c] + k;
    }
}

int build_dgrad_perm_table(int dtype, int N, int H, int W, int C, int K, int R, int S, int pad, int bm, GatherEntry* table,
                           hipStream_t st) {
    GatherGeom g;
    int rc = gather_geom(GATHER_DGRAD, dtype, N, H, W, C, K, R, S, 2, pad, &g);
    if (rc) return rc;
    const int P = (H + 2 * pad - R) / 2 + 1, Q = (W + 2 * pad - S) / 2 + 1;
    const PermGeom pg = perm_geom(N, H, W, bm);
    GDL_REQUIRE((size_t)pg.seg[4] <= dgrad_perm_cap(N, H, W), "gather: permuted table overflow");
    const size_t cap = dgrad_perm_cap(N, H, W);
    int* orow = (int*)(table + cap);
    unsigned* ttaps = (unsigned*)(orow + cap);
    hipError_t he = hipMemsetAsync(ttaps, 0, (cap / 64 + 1) * 4, st);
    if (he != hipSuccess) return check_hip(he, "hipMemsetAsync(tile taps)");
    hipLaunchKernelGGL(dgrad_perm_table_kernel, dim3(ceil_div(pg.seg[4], 256)), dim3(256), 0, st, table, orow, ttaps, bm, pg, N,
                       H, W, P, Q, R, S, pad, g.row_bytes);
    GDL_CHECK_LAUNCH("dgrad_perm_table_kernel");
    hipLaunchKernelGGL(dgrad_perm_order_kernel, dim3(1), dim3(1), 0, st, (int*)(ttaps + (cap / 64 + 1)), pg, bm);
    GDL_CHECK_LAUNCH("dgrad_perm_order_kernel");
    return GDL_OK;
}

}  // namespace gdl
