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
// `name` must be a string with static lifetime; returns a token (>= 0) when enabled, -1 otherwise.
// ext: the caller passes the two events to hipExtLaunchKernelGGL, which stamps them with the kernel's own
// begin / end times (what rocprofv3 reports); otherwise the events are recorded on the stream around the
// launch, which on a device shared by several streams also counts the wait for free CUs.
int prof_begin(const char* name, int bound, hipStream_t st, bool ext = false);
// work: algorithmic flops (MFMA-bound kernels) or algorithmic bytes (HBM-bound kernels)
// bytes: for an MFMA-bound kernel additionally its algorithmic HBM bytes (operands once + result), so that the launch can be
// priced against the COMBINED roofline max(flop / MFMA peak, bytes / HBM peak) -- gdl_prof_collect_floor (0: not stated)
void prof_end(int token, hipStream_t st, double work, bool ext = false, double bytes = 0.0);
void prof_events(int token, hipEvent_t* e0, hipEvent_t* e1);

struct ProfScope {
    int tok;
    hipStream_t st;
    double work;
    bool ext;
    double bytes;
    ProfScope(const char* name, int bound, hipStream_t s, double w, bool ext_launch = false, double hbm_bytes = 0.0)
        : tok(prof_begin(name, bound, s, ext_launch)), st(s), work(w), ext(ext_launch), bytes(hbm_bytes) {}
    ~ProfScope() {
        if (tok >= 0) prof_end(tok, st, work, ext, bytes);
    }
    // events for hipExtLaunchKernelGGL (null when the tap is off: a plain launch)
    hipEvent_t e0() const {
        hipEvent_t a = nullptr, b = nullptr;
        if (tok >= 0 && ext) prof_events(tok, &a, &b);
        return a;
    }
    hipEvent_t e1() const {
        hipEvent_t a = nullptr, b = nullptr;
        if (tok >= 0 && ext) prof_events(tok, &a, &b);
        return b;
    }
};

}  // namespace gdl
