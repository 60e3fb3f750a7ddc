// bnacc.h -- forward BatchNorm statistics through 64-bit integer accumulators (round 3).
//
// nn.BatchNorm2d in training mode (/root/reference/models/backbone.py:45-48,104,144) needs, per channel, sum(y) and sum(y^2) over
// the whole convolution output before the first normalised value can be produced.  Rounds 1-2: the convolution's epilogue
// left one fp32 partial row per M-tile and a finalize kernel folded the rows -- a 5 us launch between two dependent kernels,
// 34 times per step, that costs the chain its whole duration whatever else runs on the device (skip bound: 0.19 ms of a 5.67 ms
// step, tools/skip_bounds.sh).  Here the epilogue adds its tile's two sums, converted to fixed point, to acc[C][2] with
// device-scope 64-bit integer atomics (no return value, no fence, no ticket: integer addition is associative, so the total is
// bit-identical whatever the arrival order -- the determinism of the fixed-order fold without a second launch), and the
// CONSUMER (bn_act, the stem's max pool) derives scale / shift of its own channels from the totals in its prologue; its first
// block also publishes mean / rstd / scale / shift for the backward and updates the running statistics.
// Fixed point: sum * 2^sh1, sum of squares * 2^sh2 with sh1 = 62 - 13 - L, sh2 = 62 - 26 - L, L = ceil(log2(rows)): exact
// headroom for |y| < 8192 (bf16 / f32 activations of this network are O(1-100)); resolution at rows = 602 112: 3e-8 / 1.5e-5
// per tile sum.  The accumulators of an encoder are zeroed by ONE hipMemsetAsync at the start of its forward.
// Guard (round 4): the headroom is finite and integer wrap-around is silent, where the fp32 partial rows gave inf / NaN.  A block
// whose sum does not fit its share of the range -- |sum * 2^sh| >= 2^62 / gridDim.x, i.e. mean |y| of the block's rows beyond
// 8192 / (channel tiles), or a NaN / inf -- does not add it: it sets the BatchNorm's flag word (acc[2 C], cleared with the
// accumulators) and the consumer turns the statistics of that BatchNorm into NaN, so a diverged activation is as loud as it
// was with floats (tests/test_step_gpu.py::test_bn_accumulator_guard).
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace gdl {

struct BnAcc {  // producer side (convolution epilogue)
    long long* acc;   // [C][2], zeroed; nullptr: off
    double s1, s2;    // 2^sh1, 2^sh2
    long long* flag;  // optional: set to non-zero when a sum exceeds the headroom (see "Guard")
};
struct BnAccFin {  // consumer side
    const long long* acc;  // nullptr: the constants come from the scale / shift arrays (a finalize kernel wrote them)
    double inv_s1, inv_s2, count;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    int64_t* nbt;
    float *save_mean, *save_rstd, *scale, *shift;
    float eps, momentum;
    const long long* flag;  // optional: non-zero = a producer exceeded the headroom -> statistics = NaN
};

static inline void bn_acc_scales(size_t rows, double* s1, double* s2) {
    int L = 0;
    while (((size_t)1 << L) < rows) ++L;
    *s1 = (double)((uint64_t)1 << (62 - 13 - L));
    *s2 = (62 - 26 - L) >= 0 ? (double)((uint64_t)1 << (62 - 26 - L)) : 1.0 / (double)((uint64_t)1 << (26 + L - 62));
}

#if defined(__HIPCC__)
// scale / shift of channel c from the totals; `publish`: also write the saved statistics and update the running ones (exactly
// one thread per channel and launch does)
__device__ __forceinline__ void bn_acc_channel(const BnAccFin& a, int c, bool publish, float& sc, float& sf) {
    double s = (double)a.acc[2 * c] * a.inv_s1;
    const double q = (double)a.acc[2 * c + 1] * a.inv_s2;
    if (a.flag && *a.flag != 0) s = __builtin_nan("");  // (headroom exceeded somewhere in this BatchNorm: be loud)
    const double mean = s / a.count;
    double var = q / a.count - mean * mean;  // biased
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)a.eps);
    const double g = (double)a.gamma[c];
    sc = (float)(g * rstd);
    sf = (float)((double)a.beta[c] - mean * g * rstd);
    if (publish) {
        a.save_mean[c] = (float)mean;
        a.save_rstd[c] = (float)rstd;
        a.scale[c] = sc;
        a.shift[c] = sf;
        if (a.running_mean) {
            const double unb = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_mean[c] + (double)a.momentum * mean);
            a.running_var[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_var[c] + (double)a.momentum * unb);
        }
        if (a.nbt && c == 0) *a.nbt += 1;
    }
}
// one tile's (or one persistent block's) sum -> the accumulator
__device__ __forceinline__ void bn_acc_add(const BnAcc& a, int c, int w, float s) {
    const double v = (double)s * (w ? a.s2 : a.s1);
    // the block's share of 2^62 (gridDim.x = EVERY block of the launch, i.e. m-tiles x n-tiles for the flat kernel: the threshold
    // per block is 2^62 / that count, conservative by the n-tile factor); also true for NaN / inf
    if (!(fabs(v) < 4.6e18 / (double)gridDim.x)) {
        // no flag word (the GELU data gradient's column sums, gdl_conv_dgrad_gelu: s2 = 0, so acc[2c + 1] only ever receives
        // zeros): the channel's second word is the poison mark, gdl_acc_to_float turns a marked channel into NaN
        unsigned long long* f = a.flag ? (unsigned long long*)a.flag : (unsigned long long*)a.acc + 2 * c + 1;
        (void)__hip_atomic_fetch_or(f, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const long long q = __double2ll_rn(v);
    (void)__hip_atomic_fetch_add((unsigned long long*)a.acc + 2 * c + w, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

}  // namespace gdl
