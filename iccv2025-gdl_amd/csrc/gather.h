// gather.h -- per-pixel gather tables for the implicit-GEMM convolution kernels.
//
// Every convolution kernel addresses its gathered operand as
//     byte_offset(m, tap) = table[m].off0 + delta[tap] + channel_byte_offset,
//     valid(m, tap)       = (table[m].mask >> tap) & 1,
// where m is the GEMM row (an output pixel of that kernel), tap = r*S + s.  The table is built
// once per convolution geometry (the encoder engine does it when it binds its workspace), so the
// kernels carry no integer division, no multiplication and no bounds arithmetic in their K-loops:
// measured on MI355X the on-the-fly decode cost 2.5x the MFMA time in VALU cycles.
//
//   GATHER_FWD   rows = output pixels (n,p,q); source = conv input x [N][H][W][C]
//                (forward, and the x operand of the weight gradient)
//   GATHER_DGRAD rows = input pixels (n,h,w);  source = dy [N][P][Q][K]
//                (data gradient; for stride 2 the mask also encodes the parity test)
#pragma once
#include "common.h"

namespace gdl {

struct GatherEntry {
    int32_t off0;   // byte offset of tap 0 (wraps when the tap is outside; only used where mask says valid)
    uint32_t mask;  // bit t set <=> tap t reads a real element
};

struct GatherGeom {
    int rows;       // number of GEMM rows (table entries)
    int ntaps;      // R*S
    int delta[9];   // byte delta per tap
    int row_bytes;  // bytes of one source pixel (channels * element size)
};

enum { GATHER_FWD = 0, GATHER_DGRAD = 1 };

// Stride-2 data gradients use a PERMUTED table: the input pixels (GEMM rows) are grouped by the parity
// class (h&1, w&1) -- only taps of matching parity reach a pixel, 1 / 2 / 2 / 4 of the 9 taps of a 3x3
// kernel and none at all for three of the four classes of a 1x1 -- each class padded to a multiple of
// the M-tile `bm`, so that every tile is class-pure and its K-loop runs over the taps that can be valid
// only (4x fewer MFMAs than masking all 9).  Layout: GatherEntry[cap], int32 orow[cap] (output pixel of
// the GEMM row, -1 for padding rows), uint32 tile_taps[cap/64 + 1] (OR of the row masks of every M-tile),
// int32 tile_order[cap/64 + 1] (launch slot -> M-tile: every XCD works through its eighth of all four
// classes in turn, gather.hip), cap = N*H*W + 4*256.
inline size_t dgrad_perm_cap(int N, int H, int W) { return (size_t)N * H * W + 4 * 256; }
// number of GEMM rows of the permuted table for M-tile bm
int dgrad_perm_rows(int N, int H, int W, int bm);
int build_dgrad_perm_table(int dtype, int N, int H, int W, int C, int K, int R, int S, int pad, int bm, GatherEntry* table,
                           hipStream_t st);
// direct stem (layout.hip): rows = output pixels (n,p,q), off0 = byte offset of padded pixel (n, 2p, 2q)
int build_stem_table(int dtype, int n_img, int H, int W, int ntaps, GatherEntry* table, hipStream_t st);
// bytes of a table (any mode / stride)
size_t gather_table_bytes(int mode, int N, int H, int W, int R, int S, int stride, int pad);

// fills `g`; returns GDL_OK or an error
int gather_geom(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, GatherGeom* g);
int build_gather_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                       GatherEntry* table, hipStream_t st);

}  // namespace gdl
