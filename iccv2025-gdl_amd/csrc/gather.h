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

// fills `g`; returns GDL_OK or an error
int gather_geom(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, GatherGeom* g);
int build_gather_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                       GatherEntry* table, hipStream_t st);

}  // namespace gdl
