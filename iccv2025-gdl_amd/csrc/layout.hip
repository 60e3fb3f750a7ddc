// layout.hip -- weight packing, the stem's padded NHWC4 input copy and NHWC<->NCHW conversion (HBM-bound helpers).
//
// The reference keeps conv weights as float32 [K][C][R][S] (nn.Conv2d, backbone.py:20-28,
// 96-101) and activations NCHW.  Inside the library activations are NHWC `dtype`, weights
// are [K][R][S][C] for forward / wgrad and [C][R][S][K] for the data gradient.
#include "common.h"
#include "prof.h"

namespace gdl {

// one thread per (k, rs, c) element: reads follow the source order along c? No: the source is
// [K][C][RS]; each thread gathers one element.  Total traffic is 2 * 11 M elements per encoder.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ krsc, T* __restrict__ crsk, int K, int C,
                                   int RS) {
    const size_t total = (size_t)K * C * RS;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i enumerates the destination krsc order: ((k*RS)+rs)*C + c
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int rs = (int)(t % RS);
        const int k = (int)(t / RS);
        const float v = w[((size_t)k * C + c) * RS + rs];
        if (krsc) storeT<T>(krsc + i, v);
        if (crsk) storeT<T>(crsk + ((size_t)c * RS + rs) * K + k, v);
    }
}

int pack_weight(int dtype, const float* w, void* krsc, void* crsk, int K, int C, int R, int S, hipStream_t st) {
    const size_t total = (size_t)K * C * R * S;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    const double esz = dtype == GDL_BF16 ? 2.0 : 4.0;
    ProfScope prof(dtype == GDL_BF16 ? "gdl::pack_weight_kernel<gdl::bf16>" : "gdl::pack_weight_kernel<float>", PROF_HBM, st,
                   (double)total * (4.0 + esz * ((krsc ? 1 : 0) + (crsk ? 1 : 0))));
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(grid), dim3(256), 0, st, w, (bf16*)krsc, (bf16*)crsk, K, C, R * S);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)krsc, (float*)crsk, K, C,
                           R * S);
    GDL_CHECK_LAUNCH("pack_weight_kernel");
    return GDL_OK;
}

// ---------------------------------------------------------------- stem
// The 7x7/2 pad-3 stem as an implicit GEMM needs no im2col matrix if the network input is first copied
// ONCE into a zero-padded channels-last image with 4 channels per pixel:
//     xp [n_img][H+6][W+8][4] of T   (3 rows / columns of zeros before, >= 3 rows / 5 columns after)
// A filter row of an output pixel (p,q) is then the 8 consecutive pixels (2p+r, 2q..2q+7) = 32 elements
// = 64 bytes (bf16) / 128 bytes (f32), always in bounds, 8 / 16-byte aligned; the 8th pixel and the 4th
// channel meet zero weights.  The GEMM's K-steps ("taps") are 128 bytes: one filter row in f32, two in
// bf16.  For CREMA-D's visual batch this replaces a 925 MB matrix (written once, read by the forward
// GEMM and by the weight gradient) by an 82 MB image that stays cache resident.
int stem_pad_hp(int H) { return H + 6; }
int stem_pad_wp(int W) { return W + 8; }
size_t stem_pad_bytes(int dtype, int n_img, int H, int W) {
    return (size_t)n_img * stem_pad_hp(H) * stem_pad_wp(W) * 4 * (dtype == GDL_BF16 ? 2 : 4);
}
int stem_taps(int dtype) { return dtype == GDL_BF16 ? 4 : 7; }  // K-steps of the forward GEMM
int stem_ic(int dtype) { return dtype == GDL_BF16 ? 64 : 32; }   // elements per K-step

// x float32 [B][Cin][T][H][W] (image n = b*T + t, backbone.py:162-164) -> xp.  One thread = one pixel.
template <typename T>
__global__ __launch_bounds__(256) void stem_pad_kernel(const float* __restrict__ x, T* __restrict__ xp, int Cin, int Tn, int H,
                                                       int W, int Hp, int Wp, size_t total, uint4* __restrict__ zero,
                                                       size_t zero_vec) {
    // (the encoder's BatchNorm accumulators, bnacc.h, are zeroed here -- the first launch of a forward -- instead of by a memset of their own)
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < zero_vec; i += (size_t)gridDim.x * 256) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    if (sizeof(T) == 2 && (Wp & 1) == 0) {
        // bf16: two pixels per thread -- up to six 4-byte loads in flight, one 16-byte store (one pixel per thread ran at 2.8 TB/s)
        const int Wh = Wp / 2;
        const size_t pairs = total / 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < pairs; i += (size_t)gridDim.x * 256) {
            const int wq = (int)(i % Wh);
            size_t t = i / Wh;
            const int hp = (int)(t % Hp);
            const int n = (int)(t / Hp);
            const int h = hp - 3, w = 2 * wq - 3;
            float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if ((unsigned)h < (unsigned)H) {
                const int b = n / Tn, tt = n - b * Tn;
                const float* row = x + (((size_t)b * Cin * Tn + tt) * H + h) * W;
                const size_t cs = (size_t)Tn * H * W;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if ((unsigned)(w + q) < (unsigned)W)
                        for (int c = 0; c < Cin; ++c) v[q][c] = row[c * cs + w + q];
            }
            *(uint4*)(xp + i * 8) = make_uint4(pack2bf(v[0][0], v[0][1]), pack2bf(v[0][2], v[0][3]), pack2bf(v[1][0], v[1][1]),
                                               pack2bf(v[1][2], v[1][3]));
        }
        return;
    }
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int wp = (int)(i % Wp);
        size_t t = i / Wp;
        const int hp = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const int h = hp - 3, w = wp - 3;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const int b = n / Tn, tt = n - b * Tn;
            for (int c = 0; c < Cin; ++c) v[c] = x[((((size_t)b * Cin + c) * Tn + tt) * H + h) * W + w];
        }
        if (sizeof(T) == 2)
            *(uint2*)(xp + i * 4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        else
            *(float4*)(xp + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
int stem_pad(int dtype, const float* x, void* xp, int B, int Cin, int T, int H, int W, hipStream_t st, void* zero,
             size_t zero_bytes) {
    GDL_REQUIRE(Cin >= 1 && Cin <= 4, "stem_pad: Cin=%d", Cin);
    GDL_REQUIRE(zero_bytes % 16 == 0 && ((uintptr_t)zero & 15) == 0, "stem_pad: zero region not 16-byte aligned");
    const size_t zv = zero ? zero_bytes / 16 : 0;
    const int Hp = stem_pad_hp(H), Wp = stem_pad_wp(W);
    const size_t total = (size_t)B * T * Hp * Wp;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    ProfScope prof(dtype == GDL_BF16 ? "gdl::stem_pad_kernel<gdl::bf16>" : "gdl::stem_pad_kernel<float>", PROF_HBM, st,
                   (double)B * Cin * T * H * W * 4.0 + (double)total * 4 * (dtype == GDL_BF16 ? 2 : 4));
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(stem_pad_kernel<bf16>, dim3(grid), dim3(256), 0, st, x, (bf16*)xp, Cin, T, H, W, Hp, Wp, total, (uint4*)zero, zv);
    else
        hipLaunchKernelGGL(stem_pad_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float*)xp, Cin, T, H, W, Hp, Wp, total, (uint4*)zero, zv);
    GDL_CHECK_LAUNCH("stem_pad_kernel");
    return GDL_OK;
}

// float32 [64][Cin][7][7] -> T [64][taps][IC]: element e of K-step t is filter row r = t*rows + e/32
// (rows = filter rows per K-step), pixel s = (e%32)/4, channel e%4; everything else is zero.
template <typename T>
__global__ void pack_stem_rows_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cin, int taps, int ic) {
    const int total = 64 * taps * ic, rows = ic / 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i % ic, t = (i / ic) % taps, k = i / (ic * taps);
        const int r = t * rows + e / 32, s = (e % 32) / 4, c = e % 4;
        storeT<T>(wp + i, (r < 7 && s < 7 && c < Cin) ? w[((k * Cin + c) * 7 + r) * 7 + s] : 0.f);
    }
}
int pack_stem_rows(int dtype, const float* w, void* wp, int cin, hipStream_t st) {
    const int taps = stem_taps(dtype), ic = stem_ic(dtype);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(pack_stem_rows_kernel<bf16>, dim3(64), dim3(256), 0, st, w, (bf16*)wp, cin, taps, ic);
    else
        hipLaunchKernelGGL(pack_stem_rows_kernel<float>, dim3(64), dim3(256), 0, st, w, (float*)wp, cin, taps, ic);
    GDL_CHECK_LAUNCH("pack_stem_rows_kernel");
    return GDL_OK;
}

// ---------------------------------------------------------------- batched weight packing
// One launch packs every conv weight of an encoder: desc[i] names a float32 [K][C][RS] source and
// its two destinations; block ranges [blk0, blk0+nblk) are assigned per tensor.
struct PackDesc {
    const float* w;
    void* krsc;
    void* crsk;
    int K, C, RS, blk0;
};
// Block = one 32(k) x 32(c) patch of one tensor, all R*S taps: the float32 source rows
// w[k][c0..c0+31][:] are contiguous runs, the patch is transposed through LDS and both
// destinations are written in 64-byte runs (krsc: 32 consecutive c; crsk: 32 consecutive k), 16 bytes per thread.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[32][32 * 9 + 1];
    int d = 0;
    while (d + 1 < ndesc && (int)blockIdx.x >= desc[d + 1].blk0) ++d;
    const PackDesc pd = desc[d];
    const int RS = pd.RS;
    const int ctiles = pd.C / 32;
    const int local = (int)blockIdx.x - pd.blk0;
    const int k0 = (local / ctiles) * 32, c0 = (local % ctiles) * 32;
    const int run = 32 * RS;  // contiguous floats per k row of the patch
    if ((run & 3) == 0) {  // (32*RS is a multiple of 4 and the row starts are 16-byte aligned: C*RS*4 and c0*RS*4 are)
        const int run4 = run >> 2;
        for (int i = threadIdx.x; i < 32 * run4; i += 256) {
            const int kk = i / run4, j4 = i - kk * run4;
            const float4 v = *(const float4*)(pd.w + ((size_t)(k0 + kk) * pd.C + c0) * RS + 4 * j4);
            float* t = &tile[kk][4 * j4];
            t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < 32 * run; i += 256) {
            const int kk = i / run, j = i - kk * run;  // j = cc*RS + rs
            tile[kk][j] = pd.w[((size_t)(k0 + kk) * pd.C + c0) * RS + j];
        }
    }
    __syncthreads();
    T* krsc = (T*)pd.krsc;
    T* crsk = (T*)pd.crsk;
    // 16-byte stores (EPC elements of the fast index per thread; 2-byte stores made this kernel 3x slower than its traffic)
    constexpr int EPC = TT<T>::EPC, G = 32 / EPC;
    for (int i = threadIdx.x; i < 32 * RS * G; i += 256) {  // krsc[k][rs][c]: c fastest
        const int cg = i % G, t = i / G;
        const int rs = t % RS, kk = t / RS;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = tile[kk][(cg * EPC + e) * RS + rs];
        *(uint4*)(krsc + ((size_t)(k0 + kk) * RS + rs) * pd.C + c0 + cg * EPC) = pack16<T>(v);
    }
    if (crsk)
        for (int i = threadIdx.x; i < 32 * RS * G; i += 256) {  // crsk[c][rs][k]: k fastest
            const int kg = i % G, t = i / G;
            const int rs = t % RS, cc = t / RS;
            float v[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = tile[kg * EPC + e][cc * RS + rs];
            *(uint4*)(crsk + ((size_t)(c0 + cc) * RS + rs) * pd.K + k0 + kg * EPC) = pack16<T>(v);
        }
}
int pack_weights_batched(int dtype, const void* desc_dev, int ndesc, int total_blocks, double bytes, hipStream_t st) {
    ProfScope prof(dtype == GDL_BF16 ? "gdl::pack_weights_batched_kernel<gdl::bf16>" : "gdl::pack_weights_batched_kernel<float>",
                   PROF_HBM, st, bytes);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL(pack_weights_batched_kernel<bf16>, dim3(total_blocks), dim3(256), 0, st,
                           (const PackDesc*)desc_dev, ndesc);
    else
        hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, st,
                           (const PackDesc*)desc_dev, ndesc);
    GDL_CHECK_LAUNCH("pack_weights_batched_kernel");
    return GDL_OK;
}

// ---------------------------------------------------------------- NHWC <-> NCHW (module boundary)
// Tiled through LDS: a block transposes a [32 pixels][32 channels] patch.
template <typename T, bool TO_NCHW>
__global__ void transpose_kernel(const void* __restrict__ src, void* __restrict__ dst, int HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    if (TO_NCHW) {
        const T* s = (const T*)src + (size_t)n * HW * C;
        float* d = (float*)dst + (size_t)n * C * HW;
        for (int r = ty; r < 32; r += 8) {
            const int p = p0 + r, c = c0 + tx;
            tile[r][tx] = (p < HW && c < C) ? loadT<T>(s + (size_t)p * C + c) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, p = p0 + tx;
            if (p < HW && c < C) d[(size_t)c * HW + p] = tile[tx][r];
        }
    } else {
        const float* s = (const float*)src + (size_t)n * C * HW;
        T* d = (T*)dst + (size_t)n * HW * C;
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, p = p0 + tx;
            tile[r][tx] = (p < HW && c < C) ? s[(size_t)c * HW + p] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int p = p0 + r, c = c0 + tx;
            if (p < HW && c < C) storeT<T>(d + (size_t)p * C + c, tile[tx][r]);
        }
    }
}
int nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, hipStream_t st) {
    const dim3 grid(ceil_div(H * W, 32), ceil_div(C, 32), N);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16, true>), grid, dim3(256), 0, st, x, (void*)y, H * W, C);
    else
        hipLaunchKernelGGL((transpose_kernel<float, true>), grid, dim3(256), 0, st, x, (void*)y, H * W, C);
    GDL_CHECK_LAUNCH("transpose_kernel(nhwc->nchw)");
    return GDL_OK;
}
int nchw_f32_to_nhwc(int dtype, const float* x, void* y, int N, int H, int W, int C, hipStream_t st) {
    const dim3 grid(ceil_div(H * W, 32), ceil_div(C, 32), N);
    if (dtype == GDL_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16, false>), grid, dim3(256), 0, st, (const void*)x, y, H * W, C);
    else
        hipLaunchKernelGGL((transpose_kernel<float, false>), grid, dim3(256), 0, st, (const void*)x, y, H * W, C);
    GDL_CHECK_LAUNCH("transpose_kernel(nchw->nhwc)");
    return GDL_OK;
}

}  // namespace gdl
