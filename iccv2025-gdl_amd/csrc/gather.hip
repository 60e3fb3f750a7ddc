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

int build_gather_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                       GatherEntry* table, hipStream_t st) {
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

}  // namespace gdl
