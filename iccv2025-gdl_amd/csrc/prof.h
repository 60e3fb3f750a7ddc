// prof.h -- optional per-launch timing tap (HIP events on the launching stream).
// Disabled by default (one branch per launch).  bench.py enables it around its timed region to
// report, per kernel (named exactly as rocprofv3 names it, so the two can be compared line by
// line), launches / total device time / algorithmic work.
#pragma once
#include "common.h"

namespace gdl {

enum { PROF_HBM = 0, PROF_MFMA = 1 };

// kernel names as rocprofv3 prints them (without the leading "void " and the argument list)
template <typename T>
inline const char* prof_tname();
template <>
inline const char* prof_tname<float>() {
    return "float";
}
template <>
inline const char* prof_tname<bf16>() {
    return "gdl::bf16";
}

bool prof_enabled();
// `name` must be a string with static lifetime; returns a token (>= 0) when enabled, -1 otherwise
int prof_begin(const char* name, int bound, hipStream_t st);
// work: algorithmic flops (MFMA-bound kernels) or algorithmic bytes (HBM-bound kernels)
void prof_end(int token, hipStream_t st, double work);

struct ProfScope {
    int tok;
    hipStream_t st;
    double work;
    ProfScope(const char* name, int bound, hipStream_t s, double w) : tok(prof_begin(name, bound, s)), st(s), work(w) {}
    ~ProfScope() {
        if (tok >= 0) prof_end(tok, st, work);
    }
};

}  // namespace gdl
