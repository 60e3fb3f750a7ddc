// common.h -- shared device/host helpers for libgdl_hip (gfx950 / MI355X only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gdl_hip.h"

namespace gdl {

// ---------------------------------------------------------------- storage types
// Activations are NHWC in one of two storage types: float (parity mode, exact f32
// MFMA) or bf16 (the benchmark configuration; fp32 accumulate / statistics).
struct bf16 {
    uint16_t v;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    bf16x2_t h;
    h.x = (__bf16)lo;
    h.y = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.f) & 0xffffu); }

// ---------------------------------------------------------------- GELU (nn.GELU(), exact erf form: swin_transformer.py:26-42)
// f32 storage (the exact-parity mode): erff.  bf16 storage: Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, branch-free: one
// reciprocal, one exponential, a degree-5 Horner chain -- a dozen instructions against libdevice erff's ~60 with its two
// divergent branches).  The forward of Mlp.fc1 evaluates GELU 231 M times per Swin-T stage-1 block at 192 frames: with erff
// the fc1 GEMM's epilogue was VALU-bound (0.474 ms against 0.280 ms for the QKV GEMM of the same shape); the error is five
// orders of magnitude below a bf16 ulp.
__device__ __forceinline__ void erf_as_parts(float x, float& erf_abs, float& ex2) {  // erf(|x|), exp(-x^2)
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    ex2 = __expf(-ax * ax);
    erf_abs = fmaf(-p * t, ex2, 1.f);
}
template <typename T>
__device__ __forceinline__ float gelu_val(float u) {
    if constexpr (sizeof(T) == 2) {
        float ea, ex;
        erf_as_parts(u * 0.70710678118654752f, ea, ex);
        return 0.5f * u * (1.f + copysignf(ea, u));
    } else {
        return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
    }
}
// d gelu / du = Phi(u) + u * phi(u)
template <typename T>
__device__ __forceinline__ float gelu_grad(float u) {
    if constexpr (sizeof(T) == 2) {
        float ea, ex;  // ex = exp(-u^2 / 2)
        erf_as_parts(u * 0.70710678118654752f, ea, ex);
        return 0.5f * (1.f + copysignf(ea, u)) + u * 0.3989422804014327f * ex;
    } else {
        return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
    }
}

template <typename T>
struct TT;
template <>
struct TT<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int DT = GDL_F32;
};
template <>
struct TT<bf16> {
    static constexpr int EPC = 8;
    static constexpr int DT = GDL_BF16;
};

// 16 bytes of T  <->  EPC floats
template <typename T>
__device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x);
    f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z);
    f[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void unpack16<bf16>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16);
    f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16);
    f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16);
    f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16);
    f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ uint4 pack16(const float* f);
template <>
__device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <>
__device__ __forceinline__ uint4 pack16<bf16>(const float* f) {
    return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}
// value as it will be read back from storage
template <typename T>
__device__ __forceinline__ float roundT(float f);
template <>
__device__ __forceinline__ float roundT<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ float roundT<bf16>(float f) {
    return bf2f(f2bf(f));
}
template <typename T>
__device__ __forceinline__ float loadT(const T* p);
template <>
__device__ __forceinline__ float loadT<float>(const float* p) {
    return *p;
}
template <>
__device__ __forceinline__ float loadT<bf16>(const bf16* p) {
    return bf2f(p->v);
}
template <typename T>
__device__ __forceinline__ void storeT(T* p, float f);
template <>
__device__ __forceinline__ void storeT<float>(float* p, float f) {
    *p = f;
}
template <>
__device__ __forceinline__ void storeT<bf16>(bf16* p, float f) {
    p->v = f2bf(f);
}

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// A "done once" mark kept per DEVICE, used like the `static bool` it replaces (`if (!x) { ...; x = true; }`): the >64 KiB
// dynamic-LDS opt-in (hipFuncSetAttribute) is a per-device attribute, so a process-wide flag would launch the large-LDS kernels on a
// second GPU without it; a bit per device ordinal, atomic (the launchers may run on the autograd threads).
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    explicit operator bool() const { return (mask.load(std::memory_order_acquire) >> dev()) & 1ull; }
    bool operator!() const { return !static_cast<bool>(*this); }
    DevOnce& operator=(bool v) {
        if (v) mask.fetch_or(1ull << dev(), std::memory_order_release);
        return *this;
    }
};

// XCD-aware linear work index.  Block b runs on XCD b % 8; the `total` work items of a launch are cut into 8 consecutive runs of
// ceil(total / 8), one per XCD, so neighbours in the linear order (the tiles of one pixel slice or M-tile) share an L2 AND every XCD
// gets its share when the outer count is small or does not divide by 8 (round 5: the 512-channel weight gradients have 4 / 2 pixel
// slices of 64 tiles -- cut by slices they ran two blocks per CU on 4 / 2 XCDs and none on the others).  -1: no work for this
// block.  Grid = xcd_grid(total).
__device__ __forceinline__ int xcd_linear(int total) {
    const int per = (total + 7) >> 3;
    const int L = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    return L < total ? L : -1;
}
static inline int xcd_grid(int total) { return ((total + 7) / 8) * 8; }

#define GDL_CHECK_LAUNCH(name)                                  \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return gdl::check_hip(e__, name); \
    } while (0)

#define GDL_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            gdl::set_error(__VA_ARGS__); \
            return GDL_ERR_ARG;           \
        }                                 \
    } while (0)

// Tuning aids: the GDL_* environment knobs of the kernels' planners are read only when GDL_TUNING=1 is set (tools/ and
// the A/B runs behind DESIGN.md's numbers); a production process never consults the environment.
const char* tune_env(const char* name);
#ifdef GDL_EXPERIMENT
// timing experiments only (tools/skip_bounds.sh; encoder.cpp documents the bits): GDL_SKIP2 = launches left out inside the
// kernels' own launchers -- 1 the weight-gradient fold launches (9-tap, per-tap, stem)
inline unsigned experiment_mask2() {
    static long v = -1;
    if (v < 0) {
        const char* env = getenv("GDL_SKIP2");
        v = env ? atol(env) : 0;
    }
    return (unsigned)v;
}
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// exact n / d for 0 <= n < 2^24 via float reciprocal + one correction step
__device__ __forceinline__ int fdiv_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    int r = n - q * d;
    if (r < 0) {
        --q;
    } else if (r >= d) {
        ++q;
    }
    return q;
}

#if defined(__HIPCC__)
// agent-scope (write-through) stores / loads of results other kernels or the host read
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

}  // namespace gdl
