// prof.h -- optional per-launch timing tap (HIP events on the launching stream).
// Disabled by default (zero cost: one branch per launch).  bench.py enables it around its timed
// region to report, for each kernel class, launches / total device time / algorithmic work.
#pragma once
#include "common.h"

namespace gdl {

enum ProfSlot {
    PROF_CONV_FWD_256x64 = 0,
    PROF_CONV_FWD_128x128,
    PROF_CONV_FWD_64x64,
    PROF_CONV_DGRAD_256x64,
    PROF_CONV_DGRAD_128x128,
    PROF_CONV_DGRAD_64x64,
    PROF_CONV_WGRAD,
    PROF_WGRAD_REDUCE,
    PROF_BN_ACT,
    PROF_BN_BWD_REDUCE,
    PROF_BN_BWD_APPLY,
    PROF_RELU_BWD,
    PROF_MAXPOOL_FWD,
    PROF_MAXPOOL_BWD,
    PROF_STEM_IM2COL,
    PROF_PACK_WEIGHT,
    PROF_SGD,
    PROF_GRAD_STATS,
    PROF_NSLOTS
};

bool prof_enabled();
// returns a token (>= 0) when enabled, -1 otherwise
int prof_begin(int slot, hipStream_t st);
// work: algorithmic flops (MFMA-bound kernels) or algorithmic bytes (HBM-bound kernels)
void prof_end(int token, hipStream_t st, double work);

struct ProfScope {
    int tok;
    hipStream_t st;
    double work;
    ProfScope(int slot, hipStream_t s, double w) : tok(prof_begin(slot, s)), st(s), work(w) {}
    ~ProfScope() {
        if (tok >= 0) prof_end(tok, st, work);
    }
};

}  // namespace gdl
