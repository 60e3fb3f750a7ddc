// pool.hip -- stem max-pool (fused with BN+ReLU) and the global average pools, NHWC.
//
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1): /root/reference/models/backbone.py:106,
// applied to relu(bn1(conv1(x))) (:166-173).  F.adaptive_avg_pool2d(a,1) /
// F.adaptive_avg_pool3d(v,1): /root/reference/models/basic_model.py:73-82.
#include "bnacc.h"
#include "common.h"
#include "prof.h"

namespace gdl {

// out[n,p,q,c] = max_{window} relu(y*scale+shift); idx = first maximum in row-major window
// order (ATen's CPU kernel updates on strict '>'); windows never are empty (pad 1 < kernel 3).
// One thread = one 16-byte channel vector of one output pixel.
// YMAX: also store the RAW (pre-BatchNorm) value at the chosen position, ymax[n,p,q,c] = y[argmax]: with it the stem's
// BatchNorm-backward reduction runs over pooled-size tensors (sum_pos g0*xhat = sum_windows dout*xhat(ymax), see
// maxpool_bn_bwd_apply_kernel) instead of over the stem output, the largest activation of the network.
// ACC: scale / shift come from the stem convolution's integer accumulators (bnacc.h), derived per block into LDS; block 0
// publishes them (and the saved / running statistics) for the backward.
constexpr int POOL_ACC_MAXC = 256;
template <typename T, bool YMAX, bool ACC>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const T* __restrict__ y, const float* __restrict__ scale_g,
                                                              const float* __restrict__ shift_g, T* __restrict__ out,
                                                              uint8_t* __restrict__ idx, T* __restrict__ ymax, int N, int H,
                                                              int W, int C, int P, int Q, BnAccFin fa) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    __shared__ __attribute__((aligned(16))) float csm[ACC ? 2 : 1][ACC ? POOL_ACC_MAXC : 4];
    if (ACC) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) bn_acc_channel(fa, c, blockIdx.x == 0, csm[0][c], csm[1][c]);
        __syncthreads();
    }
    const float* const scale = ACC ? csm[0] : scale_g;  // (address space resolved at compile time: ACC is a template parameter)
    const float* const shift = ACC ? csm[1] : shift_g;
    // a block walks whole output rows (n,p): one 32-bit division per row instead of three 64-bit ones per element
    const int rows = N * P, per_row = Q * cpr;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
      const int n = row / P, p = row - n * P;
      for (int jj = threadIdx.x; jj < per_row; jj += blockDim.x) {
        const int q = jj / cpr, vc = jj - q * cpr;
        const size_t i = (size_t)row * per_row + jj;
        float sc[EPC], sf[EPC], best[EPC];
        int bi[EPC];
#pragma unroll
        for (int q4 = 0; q4 < EPC / 4; ++q4) {  // 16-byte loads of the per-channel constants
            const float4 a = *(const float4*)(scale + vc * EPC + 4 * q4), b = *(const float4*)(shift + vc * EPC + 4 * q4);
            sc[4 * q4 + 0] = a.x, sc[4 * q4 + 1] = a.y, sc[4 * q4 + 2] = a.z, sc[4 * q4 + 3] = a.w;
            sf[4 * q4 + 0] = b.x, sf[4 * q4 + 1] = b.y, sf[4 * q4 + 2] = b.z, sf[4 * q4 + 3] = b.w;
        }
        if constexpr (sizeof(T) == 2) {
            // bf16 storage: the candidates are compared as fp32 (as the reference compares them) through ONE unsigned
            // key per candidate -- a non-negative float orders like its bit pattern, and its 4 lowest mantissa bits
            // carry 15 - code, so that among equal values the first window position wins (values closer than 16 fp32
            // ulps count as equal).  The winner's raw input rides along for YMAX.
            uint32_t key[EPC];
            float ym[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                key[e] = 0u;
                ym[e] = 0.f;
            }
            // all nine window vectors are requested before the first is used (clamped address + validity flag instead of a
            // branch around each load: nine loads in flight per thread)
            uint4 v9[9];
            bool ok9[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = p * 2 - 1 + r;
                const int ihc = min(max(ih, 0), H - 1);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int iw = q * 2 - 1 + s;
                    const int iwc = min(max(iw, 0), W - 1);
                    ok9[r * 3 + s] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    v9[r * 3 + s] = *(const uint4*)(y + (((size_t)n * H + ihc) * W + iwc) * C + vc * EPC);
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float f[EPC];
                unpack16<T>(v9[t], f);
                const uint32_t tag = 15u - (uint32_t)t;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float v = f[e] * sc[e] + sf[e];
                    v = v > 0.f ? v : 0.f;  // (+0 also for -0 and NaN: the key compare is unsigned)
                    const uint32_t k = (__float_as_uint(v) & 0xfffffff0u) | tag;
                    const bool gt = ok9[t] && k > key[e];
                    key[e] = gt ? k : key[e];
                    if constexpr (YMAX) ym[e] = gt ? f[e] : ym[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                best[e] = __uint_as_float(key[e] & 0xfffffff0u);
                bi[e] = 15 - (int)(key[e] & 15u);
            }
            if constexpr (YMAX) *(uint4*)(ymax + i * EPC) = pack16<T>(ym);
        } else {
        float ym[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            best[e] = -INFINITY;
            bi[e] = -1;
            ym[e] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = p * 2 - 1 + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = q * 2 - 1 + s;
                if ((unsigned)iw >= (unsigned)W) continue;
                float f[EPC];
                unpack16<T>(*(const uint4*)(y + (((size_t)n * H + ih) * W + iw) * C + vc * EPC), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float v = f[e] * sc[e] + sf[e];
                    v = roundT<T>(v > 0.f ? v : 0.f);
                    if (v > best[e] || bi[e] < 0) {
                        best[e] = v;
                        bi[e] = r * 3 + s;
                        ym[e] = f[e];
                    }
                }
            }
        }
        if constexpr (YMAX) *(uint4*)(ymax + i * EPC) = pack16<T>(ym);
        }
        *(uint4*)(out + i * EPC) = pack16<T>(best);
        uint8_t* ip = idx + i * EPC;
        if (EPC == 8) {
            *(uint2*)ip = make_uint2((uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24),
                                     (uint32_t)bi[4 % EPC] | ((uint32_t)bi[5 % EPC] << 8) | ((uint32_t)bi[6 % EPC] << 16) |
                                         ((uint32_t)bi[7 % EPC] << 24));
        } else {
            *(uint32_t*)ip = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        }
      }
    }
}
int bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* ymax,
                        int N, int H, int W, int C, hipStream_t st, const BnAccFin* fa) {
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1;
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    GDL_REQUIRE(C % epc == 0, "maxpool: C=%d", C);
    const bool acc = fa && fa->acc;
    GDL_REQUIRE(!acc || C <= POOL_ACC_MAXC, "maxpool: C=%d above %d with accumulators", C, POOL_ACC_MAXC);
    const BnAccFin a0 = acc ? *fa : BnAccFin{};
    const size_t total = (size_t)N * P * Q * (C / epc);
    const int grid = N * P > 8192 ? 8192 : N * P;
    // read the stem output once, write pooled values + 1-byte indices
    ProfScope prof(dtype == GDL_BF16 ? "gdl::bn_relu_maxpool_kernel<gdl::bf16>" : "gdl::bn_relu_maxpool_kernel<float>", PROF_HBM, st,
                   (double)N * H * W * C * (16.0 / epc) + (double)total * ((ymax ? 32.0 : 16.0) + epc));
#define GDL_POOL_LAUNCH(TT_, YM)                                                                                                 \
    do {                                                                                                                         \
        if (acc)                                                                                                                 \
            hipLaunchKernelGGL((bn_relu_maxpool_kernel<TT_, YM, true>), dim3(grid), dim3(256), 0, st, (const TT_*)y, scale,      \
                               shift, (TT_*)out, idx, (TT_*)ymax, N, H, W, C, P, Q, a0);                                         \
        else                                                                                                                     \
            hipLaunchKernelGGL((bn_relu_maxpool_kernel<TT_, YM, false>), dim3(grid), dim3(256), 0, st, (const TT_*)y, scale,     \
                               shift, (TT_*)out, idx, (TT_*)ymax, N, H, W, C, P, Q, a0);                                         \
    } while (0)
    if (dtype == GDL_BF16) {
        if (ymax)
            GDL_POOL_LAUNCH(bf16, true);
        else
            GDL_POOL_LAUNCH(bf16, false);
    } else {
        if (ymax)
            GDL_POOL_LAUNCH(float, true);
        else
            GDL_POOL_LAUNCH(float, false);
    }
#undef GDL_POOL_LAUNCH
    GDL_CHECK_LAUNCH("bn_relu_maxpool_kernel");
    return GDL_OK;
}

// dx[n,h,w,c] = sum over the <= 2x2 windows (p,q) that contain (h,w) and whose idx names it.
// Window (p,q) covers rows 2p-1..2p+1: h belongs to p = floor(h/2) (r = h-2p+1 in {1,2}) and,
// if h is odd, also to p = (h+1)/2 (r = 0).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int P, int Q) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    const int rows = N * H, per_row = W * cpr;  // a block walks whole input rows (n,h)
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
      const int n = row / H, h = row - n * H;
      for (int jj = threadIdx.x; jj < per_row; jj += blockDim.x) {
        const int w = jj / cpr, vc = jj - w * cpr;
        const size_t i = (size_t)row * per_row + jj;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int p = (h >> 1) + a;
            if (a == 1 && !(h & 1)) continue;
            if (p >= P) continue;
            const int r = h - (2 * p - 1);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int q = (w >> 1) + b;
                if (b == 1 && !(w & 1)) continue;
                if (q >= Q) continue;
                const int s = w - (2 * q - 1);
                const int code = r * 3 + s;
                const size_t o = (((size_t)n * P + p) * Q + q) * C + vc * EPC;
                float d[EPC];
                unpack16<T>(*(const uint4*)(dout + o), d);
                uint8_t ix[EPC];
                if (EPC == 8) {
                    const uint2 u = *(const uint2*)(idx + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ix[e] = (u.x >> (8 * e)) & 0xff;
                        ix[(4 + e) % EPC] = (u.y >> (8 * e)) & 0xff;
                    }
                } else {
                    const uint32_t u = *(const uint32_t*)(idx + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ix[e] = (u >> (8 * e)) & 0xff;
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (ix[e] == code) acc[e] += d[e];
            }
        }
        *(uint4*)(dx + i * EPC) = pack16<T>(acc);
      }
    }
}
// The same gather, one thread per 2x2 patch of input pixels (rows 2p, 2p+1, columns 2q, 2q+1): the patch belongs to the
// four windows (p,q), (p,q+1), (p+1,q), (p+1,q+1), each of which is loaded and decoded ONCE for the (up to four)
// positions it may name -- 4 window reads per 4 outputs instead of 9, the compares unchanged:
//   (2p,2q): (p,q)=4 | (2p,2q+1): (p,q)=5, (p,q+1)=3 | (2p+1,2q): (p,q)=7, (p+1,q)=1 |
//   (2p+1,2q+1): (p,q)=8, (p,q+1)=6, (p+1,q)=2, (p+1,q+1)=0        (code = r*3 + s of the position inside the window)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_patch_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                T* __restrict__ dx, int N, int H, int W, int C, int P, int Q) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    const int PH = (H + 1) >> 1, QW = (W + 1) >> 1;
    const int rows = N * PH, per_row = QW * cpr;  // a block walks whole patch rows (n, p)
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / PH, p = row - n * PH;
        for (int jj = threadIdx.x; jj < per_row; jj += blockDim.x) {
            const int q = jj / cpr, vc = jj - q * cpr;
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            float a00[EPC], a01[EPC], a10[EPC], a11[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) a00[e] = a01[e] = a10[e] = a11[e] = 0.f;
            // the gradient / arg-max vectors of the four windows that touch the patch: twelve loads in flight
            // per thread instead of four dependent load -> decode round trips (an absent window reads as arg-max 255: no match)
            // (named scalars, not arrays indexed by the window number: the index array ended up in scratch memory -- 32 bytes per
            // lane read back inside the hottest loop of the stem's backward)
            uint4 dq0 = z, dq1 = z, dq2 = z, dq3 = z;
            uint2 iq0 = make_uint2(0xffffffffu, 0xffffffffu), iq1 = iq0, iq2 = iq0, iq3 = iq0;
            auto fetch = [&](int k, uint4& dqk, uint2& iqk) __attribute__((always_inline)) {
                const int pp = p + (k >> 1), qq = q + (k & 1);
                if (pp < P && qq < Q) {
                    const size_t o = (((size_t)n * P + pp) * Q + qq) * C + vc * EPC;
                    dqk = *(const uint4*)(dout + o);
                    if constexpr (EPC == 8)
                        iqk = *(const uint2*)(idx + o);
                    else
                        iqk.x = *(const uint32_t*)(idx + o);
                }
            };
            fetch(0, dq0, iq0), fetch(1, dq1, iq1), fetch(2, dq2, iq2), fetch(3, dq3, iq3);
            auto window = [&](const uint4& dqk, const uint2& u, int c00, int c01, int c10, int c11) __attribute__((always_inline)) {
                float d[EPC];
                unpack16<T>(dqk, d);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    // (the arg-max code of element e straight from the packed bytes: an ix[] array here lived in scratch memory)
                    const uint32_t code = ((e < 4 ? u.x : u.y) >> (8 * (e & 3))) & 0xffu;
                    if (c00 >= 0 && code == (uint32_t)c00) a00[e] += d[e];
                    if (c01 >= 0 && code == (uint32_t)c01) a01[e] += d[e];
                    if (c10 >= 0 && code == (uint32_t)c10) a10[e] += d[e];
                    if (c11 >= 0 && code == (uint32_t)c11) a11[e] += d[e];
                }
            };
            // (the order of the additions into one position is that of maxpool_bwd_kernel: window rows, then columns)
            window(dq0, iq0, 4, 5, 7, 8);
            window(dq1, iq1, -1, 3, -1, 6);
            window(dq2, iq2, -1, -1, 1, 2);
            window(dq3, iq3, -1, -1, -1, 0);
            const int h = 2 * p, w = 2 * q;
            T* o0 = dx + (((size_t)n * H + h) * W + w) * C + vc * EPC;
            *(uint4*)o0 = pack16<T>(a00);
            if (w + 1 < W) *(uint4*)(o0 + C) = pack16<T>(a01);
            if (h + 1 < H) {
                T* o1 = o0 + (size_t)W * C;
                *(uint4*)o1 = pack16<T>(a10);
                if (w + 1 < W) *(uint4*)(o1 + C) = pack16<T>(a11);
            }
        }
    }
}
int maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, hipStream_t st) {
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1;
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    const size_t total = (size_t)N * H * W * (C / epc);
    const int grid = N * H > 16384 ? 16384 : N * H;
    ProfScope prof(dtype == GDL_BF16 ? "gdl::maxpool_bwd_kernel<gdl::bf16>" : "gdl::maxpool_bwd_kernel<float>", PROF_HBM, st, (double)total * 16.0 + (double)N * P * Q * C * (16.0 / epc + 1.0));
    static int patch = -1;
    if (patch < 0) {
        const char* e = tune_env("GDL_POOL_PATCH");  // tuning aid: 0 = one thread per input pixel
        patch = e ? atoi(e) : 1;
    }
    if (patch) {
        const int prow = N * ((H + 1) / 2), pgrid = prow > 16384 ? 16384 : prow;
        if (dtype == GDL_BF16)
            hipLaunchKernelGGL(maxpool_bwd_patch_kernel<bf16>, dim3(pgrid), dim3(256), 0, st, (const bf16*)dout, idx, (bf16*)dx, N,
                               H, W, C, P, Q);
        else
            hipLaunchKernelGGL(maxpool_bwd_patch_kernel<float>, dim3(pgrid), dim3(256), 0, st, (const float*)dout, idx, (float*)dx,
                               N, H, W, C, P, Q);
    } else if (dtype == GDL_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dout, idx, (bf16*)dx, N, H,
                           W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dout, idx, (float*)dx, N,
                           H, W, C, P, Q);
    GDL_CHECK_LAUNCH("maxpool_bwd_kernel");
    return GDL_OK;
}

// Stem backward, max-pool gather + ReLU mask + BatchNorm-backward apply in one pass (maxpool_bwd + bn_bwd_apply<MASK>):
//   g0[pos]  = sum of dout over the windows whose idx names pos            (never stored: the stem output is the largest
//   dy0[pos] = gamma*rstd*( (bn(y0[pos]) > 0 ? g0[pos] : 0) - coef0 - xhat(y0[pos])*coef1 )           activation)
// One thread per 2x2 patch of stem-output pixels and 16-byte channel vector, as maxpool_bwd_patch_kernel.  The two
// reductions behind coef come from the pooled tensors: sum_pos g0' = sum_windows dout',  sum_pos g0'*xhat(y0[pos]) =
// sum_windows dout'*xhat(ymax)  with  dout' = dout*(bn(ymax) > 0)  -- i.e. bn_bwd_reduce<MASK> over (dout, ymax).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                   const T* __restrict__ y0, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ coef, T* __restrict__ dy, int N,
                                                                   int H, int W, int C, int P, int Q) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    const int PH = (H + 1) >> 1, QW = (W + 1) >> 1;
    const int rows = N * PH, per_row = QW * cpr;  // a block walks whole patch rows (n, p)
    // blockDim % cpr == 0: the thread's channel vector is the same for every patch it visits
    const int vc = threadIdx.x % cpr;
    // dy = gamma*rstd*(g' - k1 - (y - mean)*rstd*k2) = A*g' + Bc*y + D   (three constants per channel instead of five)
    float sc[EPC], sf[EPC], A[EPC], Bc[EPC], D[EPC];
    {
        float mu[EPC], rs[EPC], gr[EPC], k1[EPC], k2[EPC];
#pragma unroll
        for (int q4 = 0; q4 < EPC / 4; ++q4) {
            auto ld = [&](const float* p, float* v) {
                const float4 t = *(const float4*)(p + vc * EPC + 4 * q4);
                v[4 * q4 + 0] = t.x, v[4 * q4 + 1] = t.y, v[4 * q4 + 2] = t.z, v[4 * q4 + 3] = t.w;
            };
            ld(scale, sc), ld(shift, sf), ld(mean, mu), ld(rstd, rs), ld(gamma, gr), ld(coef, k1), ld(coef + C, k2);
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            A[e] = gr[e] * rs[e];
            Bc[e] = -A[e] * rs[e] * k2[e];
            D[e] = -A[e] * k1[e] - Bc[e] * mu[e];
        }
    }
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / PH, p = row - n * PH;
        for (int jj = threadIdx.x; jj < per_row; jj += blockDim.x) {
            const int q = jj / cpr;
            const int h = 2 * p, w = 2 * q;
            const bool w1 = w + 1 < W, h1 = h + 1 < H;
            const size_t o00 = (((size_t)n * H + h) * W + w) * C + vc * EPC;
            // the four y0 vectors of the patch are requested before the windows are decoded
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            const uint4 y00 = *(const uint4*)(y0 + o00);
            // (clamped addresses instead of `cond ? *p : z`: the compiler turned that into a load through `cond ? p : &z` -- a
            // FLAT load from a pointer select with z spilled to scratch memory every iteration; an absent neighbour's vector is
            // read from the patch's own pixel and never stored)
            const size_t ow = w1 ? (size_t)C : 0, oh = h1 ? (size_t)W * C : 0;
            const uint4 y01 = *(const uint4*)(y0 + o00 + ow);
            const uint4 y10 = *(const uint4*)(y0 + o00 + oh);
            const uint4 y11 = *(const uint4*)(y0 + o00 + oh + ow);
            float a00[EPC], a01[EPC], a10[EPC], a11[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) a00[e] = a01[e] = a10[e] = a11[e] = 0.f;
            // ... and so are the gradient / arg-max vectors of the four windows that touch the patch: twelve loads in flight
            // per thread instead of four dependent load -> decode round trips (an absent window reads as arg-max 255: no match)
            uint4 dq[4];
            uint2 iq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pp = p + (k >> 1), qq = q + (k & 1);
                dq[k] = z;
                iq[k] = make_uint2(0xffffffffu, 0xffffffffu);
                if (pp < P && qq < Q) {
                    const size_t o = (((size_t)n * P + pp) * Q + qq) * C + vc * EPC;
                    dq[k] = *(const uint4*)(dout + o);
                    if (EPC == 8)
                        iq[k] = *(const uint2*)(idx + o);
                    else
                        iq[k].x = *(const uint32_t*)(idx + o);
                }
            }
            auto window = [&](int k, int c00, int c01, int c10, int c11) __attribute__((always_inline)) {
                float d[EPC];
                unpack16<T>(dq[k], d);
                uint32_t ix[EPC];
                if (EPC == 8) {
                    const uint2 u = iq[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ix[e] = (u.x >> (8 * e)) & 0xff;
                        ix[(4 + e) % EPC] = (u.y >> (8 * e)) & 0xff;
                    }
                } else {
                    const uint32_t u = iq[k].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ix[e] = (u >> (8 * e)) & 0xff;
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    if (c00 >= 0 && ix[e] == (uint32_t)c00) a00[e] += d[e];
                    if (c01 >= 0 && ix[e] == (uint32_t)c01) a01[e] += d[e];
                    if (c10 >= 0 && ix[e] == (uint32_t)c10) a10[e] += d[e];
                    if (c11 >= 0 && ix[e] == (uint32_t)c11) a11[e] += d[e];
                }
            };
            window(0, 4, 5, 7, 8);
            window(1, -1, 3, -1, 6);
            window(2, -1, -1, 1, 2);
            window(3, -1, -1, -1, 0);
            auto apply = [&](const uint4& yq, float (&g)[EPC]) __attribute__((always_inline)) {
                float yv[EPC];
                unpack16<T>(yq, yv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float gg = (yv[e] * sc[e] + sf[e] > 0.f) ? g[e] : 0.f;
                    g[e] = A[e] * gg + (Bc[e] * yv[e] + D[e]);
                }
                return pack16<T>(g);
            };
            T* d0 = dy + o00;
            *(uint4*)d0 = apply(y00, a00);
            if (w1) *(uint4*)(d0 + C) = apply(y01, a01);
            if (h1) {
                T* d1 = d0 + (size_t)W * C;
                *(uint4*)d1 = apply(y10, a10);
                if (w1) *(uint4*)(d1 + C) = apply(y11, a11);
            }
        }
    }
}
int maxpool_bn_bwd_apply(int dtype, const void* dout, const uint8_t* idx, const void* y0, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* gamma, const float* coef, void* dy, int N, int H,
                         int W, int C, hipStream_t st) {
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1;
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    GDL_REQUIRE(C % epc == 0 && 256 % (C / epc) == 0, "maxpool_bn_bwd_apply: C=%d", C);
    const size_t total = (size_t)N * H * W * (C / epc);
    const int prow = N * ((H + 1) / 2), pgrid = prow > 16384 ? 16384 : prow;
    // read y0 + the pooled gradient and indices, write dy0
    ProfScope prof(dtype == GDL_BF16 ? "gdl::maxpool_bn_bwd_apply_kernel<gdl::bf16>" : "gdl::maxpool_bn_bwd_apply_kernel<float>",
                   PROF_HBM, st, (double)total * 32.0 + (double)N * P * Q * C * (16.0 / epc + 1.0));
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(maxpool_bn_bwd_apply_kernel<bf16>, dim3(pgrid), dim3(256), 0, st, (const bf16*)dout, idx, (const bf16*)y0,
                           scale, shift, mean, rstd, gamma, coef, (bf16*)dy, N, H, W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool_bn_bwd_apply_kernel<float>, dim3(pgrid), dim3(256), 0, st, (const float*)dout, idx,
                           (const float*)y0, scale, shift, mean, rstd, gamma, coef, (float*)dy, N, H, W, C, P, Q);
    GDL_CHECK_LAUNCH("maxpool_bn_bwd_apply_kernel");
    return GDL_OK;
}

// ---------------------------------------------------------------- global average pools
// x [B*T][HW][C] -> feat[b][c] = mean over t, hw.  One block per (b, 16-byte channel vector group).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ feat, int T_, int HW,
                                                          int C) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ float red[];  // [rpp][C]
    const int cpr = C / EPC;
    const int b = blockIdx.x;
    const int rows = T_ * HW;
    const T* xb = x + (size_t)b * rows * C;
    const int rpp = 256 / cpr > 0 ? 256 / cpr : 1;
    for (int v0 = 0; v0 < cpr; v0 += 256) {  // cpr <= 256 in practice (C = 512: 64 / 128 vectors)
        const int vc = v0 + threadIdx.x % (cpr < 256 ? cpr : 256);
        const int vr = threadIdx.x / (cpr < 256 ? cpr : 256);
        float s[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] = 0.f;
        if (vc < cpr) {  // eight rows requested at a time, added in row order (a load per iteration was 37 round trips: 15-20 us
                         // on both chains between the forward's last convolution and the backward's first)
            constexpr int UN = 8;
            for (int r0 = vr; r0 < rows; r0 += UN * rpp) {
                uint4 q[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    if (r0 + u * rpp < rows) q[u] = *(const uint4*)(xb + (size_t)(r0 + u * rpp) * C + vc * EPC);
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    if (r0 + u * rpp < rows) {
                        float f[EPC];
                        unpack16<T>(q[u], f);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) s[e] += f[e];
                    }
            }
        }
        if (vc < cpr) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) red[(size_t)vr * C + vc * EPC + e] = s[e];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f;
            for (int r = 0; r < rpp; ++r) a += red[(size_t)r * C + c];
            feat[(size_t)b * C + c] = a / (float)rows;
        }
        __syncthreads();
    }
}
int avgpool_fwd(int dtype, const void* x, float* feat, int B, int T, int HW, int C, hipStream_t st) {
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    const int cpr = C / epc;
    GDL_REQUIRE(C % epc == 0 && cpr <= 256 && 256 % cpr == 0, "avgpool: C=%d unsupported", C);
    const size_t sh = (size_t)(256 / cpr) * C * sizeof(float);
    ProfScope prof("gdl::avgpool_fwd_kernel", PROF_HBM, st, (double)B * T * HW * C * (16.0 / epc));
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(avgpool_fwd_kernel<bf16>, dim3(B), dim3(256), sh, st, (const bf16*)x, feat, T, HW, C);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(B), dim3(256), sh, st, (const float*)x, feat, T, HW, C);
    GDL_CHECK_LAUNCH("avgpool_fwd_kernel");
    return GDL_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx, int rows,
                                                          int C, size_t nvec) {
    constexpr int EPC = TT<T>::EPC;
    const int cpr = C / EPC;
    const float inv = 1.0f / (float)rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int vc = (int)(i % cpr);
        const size_t b = (i / cpr) / rows;
        float f[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = dfeat[b * C + vc * EPC + e] * inv;
        *(uint4*)(dx + i * EPC) = pack16<T>(f);
    }
}
int avgpool_bwd(int dtype, const float* dfeat, void* dx, int B, int T, int HW, int C, hipStream_t st) {
    const int epc = dtype == GDL_BF16 ? 8 : 4;
    const size_t nvec = (size_t)B * T * HW * (C / epc);
    const int grid = (int)((nvec + 255) / 256 > 4096 ? 4096 : (nvec + 255) / 256);
    ProfScope prof("gdl::avgpool_bwd_kernel", PROF_HBM, st, (double)nvec * 16.0);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(avgpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, dfeat, (bf16*)dx, T * HW, C, nvec);
    else
        hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, dfeat, (float*)dx, T * HW, C, nvec);
    GDL_CHECK_LAUNCH("avgpool_bwd_kernel");
    return GDL_OK;
}

}  // namespace gdl
