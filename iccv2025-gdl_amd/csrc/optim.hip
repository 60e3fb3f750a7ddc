// optim.hip -- fused multi-tensor gradient statistics, clipping and SGD over flat arenas.
//
// Replaces, for the step of /root/reference/main_dgl.py:
//   nn.utils.clip_grad_norm_(model.parameters(), max_norm=40, norm_type=2)      (:129)
//   sum_p torch.abs(p.grad).mean() over audio_net / visual_net parameters        (:132-143)
//   optim.SGD(lr, momentum=0.9, weight_decay=1e-4).step()                        (:249,154)
// which the reference runs as ~370 small launches and 120 host syncs.  Here: one pass over
// the gradient arena (sum of squares + sum of |g| per parameter, fixed-order reduction in
// double), one tiny finalise kernel that leaves {total_norm, clip_coef, audio_sum,
// visual_sum, per-parameter norms} on the device, and one streaming update kernel.
#include "common.h"
#include "prof.h"

#include <string.h>

#include <vector>

struct gdl_optim {
    int nseg = 0;
    int64_t total = 0;
    int nchunks = 0;
    std::vector<int64_t> offs;
    std::vector<int32_t> group;
    // The object owns NO device memory (the library never allocates or frees device memory).  Its descriptor
    // tables -- chunk descriptors ChunkDesc[nchunks], per-segment chunk ranges int32[nseg][4] = {first_chunk, n_chunks, group, 0},
    // per-segment element counts double[nseg] -- are kept on the host and uploaded into the head of the caller-provided
    // workspace the first time gdl_optim_grad_stats sees that workspace (stream-ordered, once per workspace pointer):
    //   ws = [chunk descriptors | segment ranges | element counts | per-chunk partials | per-segment sums]
    std::vector<unsigned char> h_tables;  // the first three, concatenated as they sit in the workspace
    size_t off_segrange = 0, off_segnumel = 0, off_partial = 0;
    const void* bound_ws = nullptr;
};

namespace gdl {

constexpr int OPT_CHUNK = 8192;

struct ChunkDesc {
    int64_t start;
    int32_t len;
    int32_t seg;
};

__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ g, const ChunkDesc* __restrict__ chunks,
                                                         double* __restrict__ partial) {
    __shared__ double sh[2][256];
    const ChunkDesc c = chunks[blockIdx.x];
    const float* p = g + c.start;
    float s2 = 0.f, s1 = 0.f;
    // 16-byte loads over the aligned middle of the chunk (segments start at arbitrary element offsets), scalar ends
    const int head = min(c.len, (int)((4 - (c.start & 3)) & 3));
    const int nv = (c.len - head) >> 2;
    const float4* p4 = (const float4*)(p + head);
    for (int i = threadIdx.x; i < nv; i += 256) {
        const float4 v = p4[i];
        s2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        s1 += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
    }
    {  // at most 3 elements in front of and 3 behind the aligned middle
        const int i = (int)threadIdx.x < head ? (int)threadIdx.x : head + 4 * nv + ((int)threadIdx.x - head);
        if (i < c.len && ((int)threadIdx.x < head || (int)threadIdx.x - head < c.len - head - 4 * nv)) {
            const float v = p[i];
            s2 += v * v;
            s1 += fabsf(v);
        }
    }
    sh[0][threadIdx.x] = (double)s2;
    sh[1][threadIdx.x] = (double)s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 2 + 0] = sh[0][0];
        partial[(size_t)blockIdx.x * 2 + 1] = sh[1][0];
    }
}

// stats[0] total_norm (pre-clip, after grad_scale), [1] clip_coef, [2] audio_sum, [3] visual_sum,
// [4+s] post-clip L2 norm of segment s, [4+nseg+s] post-clip mean|g| of segment s.
constexpr int FIN_NT = 1024;
__global__ __launch_bounds__(FIN_NT) void grad_stats_final_kernel(const double* __restrict__ partial,
                                                               const int32_t* __restrict__ segrange,
                                                               const double* __restrict__ segnumel, int nseg,
                                                               float max_norm, float grad_scale, float* __restrict__ stats,
                                                               double* __restrict__ segsum /*[nseg][2] scratch*/) {
    __shared__ double sh[3][FIN_NT];
    // A 16-lane group per segment (a layer-4 convolution has 288 chunk partials: one thread adding them serially
    // took 47 us): lane l adds chunks l, l+16, ..., the 16 lanes are folded in a fixed order.  Round 5: 1024 threads = 64 groups
    // (the 122 segments in two rounds instead of eight: this launch sits alone between the last gradient and the update, 18 us).
    const int grp16 = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    for (int s = grp16; s < nseg; s += FIN_NT / 16) {
        const int first = segrange[s * 4 + 0], cnt = segrange[s * 4 + 1];
        double a = 0.0, b = 0.0;
        for (int k = l16; k < cnt; k += 16) {
            a += partial[(size_t)(first + k) * 2 + 0];
            b += partial[(size_t)(first + k) * 2 + 1];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 16);
            b += __shfl_down(b, o, 16);
        }
        if (l16 == 0) {
            segsum[s * 2 + 0] = a;
            segsum[s * 2 + 1] = b;
        }
    }
    __syncthreads();  // (segsum is global memory written and read by this one block)
    double tot = 0.0;
    for (int s = threadIdx.x; s < nseg; s += FIN_NT) tot += segsum[s * 2 + 0];
    sh[0][threadIdx.x] = tot;
    __syncthreads();
    for (int o = FIN_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
        __syncthreads();
    }
    const double norm = sqrt(sh[0][0]) * (double)grad_scale;
    // torch: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1 (float arithmetic there; double here)
    double coef = (double)max_norm / (norm + 1e-6);
    if (coef > 1.0) coef = 1.0;
    __syncthreads();
    double au = 0.0, vi = 0.0;
    for (int s = threadIdx.x; s < nseg; s += FIN_NT) {
        const double l2 = sqrt(segsum[s * 2 + 0]) * (double)grad_scale * coef;
        const double am = segsum[s * 2 + 1] / segnumel[s] * (double)grad_scale * coef;
        stats[4 + s] = (float)l2;
        stats[4 + nseg + s] = (float)am;
        const int grp = segrange[s * 4 + 2];
        if (grp == 1) au += am;
        if (grp == 2) vi += am;
    }
    sh[1][threadIdx.x] = au;
    sh[2][threadIdx.x] = vi;
    __syncthreads();
    for (int o = FIN_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = (float)norm;
        stats[1] = (float)coef;
        stats[2] = (float)sh[1][0];
        stats[3] = (float)sh[2][0];
    }
}

// g *= coef*grad_scale (stored back: p.grad is clipped in place by the reference);
// d = g + wd*p; m = mu*m + d; p -= lr*m
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                  const float* __restrict__ stats, float grad_scale, float lr, float mu,
                                                  float wd, int64_t n) {
    const float k = (stats ? stats[1] : 1.f) * grad_scale;
    const bool wb = k != 1.f;  // (clip inactive on one rank: g * 1 is g -- its 4 n bytes are not written back)
    const int64_t nv = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float4 pv = ((float4*)p)[i], gv = ((float4*)g)[i], mv = ((float4*)m)[i];
        gv.x *= k;
        gv.y *= k;
        gv.z *= k;
        gv.w *= k;
        mv.x = mu * mv.x + (gv.x + wd * pv.x);
        mv.y = mu * mv.y + (gv.y + wd * pv.y);
        mv.z = mu * mv.z + (gv.z + wd * pv.z);
        mv.w = mu * mv.w + (gv.w + wd * pv.w);
        pv.x -= lr * mv.x;
        pv.y -= lr * mv.y;
        pv.z -= lr * mv.z;
        pv.w -= lr * mv.w;
        if (wb) ((float4*)g)[i] = gv;
        ((float4*)m)[i] = mv;
        ((float4*)p)[i] = pv;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 256) {
            const float gg = g[i] * k;
            const float mm = mu * m[i] + (gg + wd * p[i]);
            if (wb) g[i] = gg;
            m[i] = mm;
            p[i] -= lr * mm;
        }
    }
}

}  // namespace gdl

using namespace gdl;

extern "C" {

int gdl_optim_create(gdl_optim_t** out, const int64_t* seg_offsets, const int32_t* seg_group, int nseg) {
    GDL_REQUIRE(out && seg_offsets && seg_group && nseg > 0, "optim_create: bad arguments");
    GDL_REQUIRE(seg_offsets[0] == 0, "optim_create: arena must start at offset 0");
    gdl_optim* o = new gdl_optim();
    o->nseg = nseg;
    o->offs.assign(seg_offsets, seg_offsets + nseg + 1);
    o->group.assign(seg_group, seg_group + nseg);
    o->total = seg_offsets[nseg];
    std::vector<ChunkDesc> chunks;
    std::vector<int32_t> segrange(nseg * 4);
    std::vector<double> numel(nseg);
    for (int s = 0; s < nseg; ++s) {
        const int64_t b = seg_offsets[s], e = seg_offsets[s + 1];
        if (e <= b) {
            delete o;
            set_error("optim_create: empty or unordered segment %d", s);
            return GDL_ERR_ARG;
        }
        segrange[s * 4 + 0] = (int32_t)chunks.size();
        for (int64_t c = b; c < e; c += OPT_CHUNK) {
            ChunkDesc d;
            d.start = c;
            d.len = (int32_t)((e - c) < OPT_CHUNK ? (e - c) : OPT_CHUNK);
            d.seg = s;
            chunks.push_back(d);
        }
        segrange[s * 4 + 1] = (int32_t)chunks.size() - segrange[s * 4 + 0];
        segrange[s * 4 + 2] = seg_group[s];
        segrange[s * 4 + 3] = 0;
        numel[s] = (double)(e - b);
    }
    o->nchunks = (int)chunks.size();
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    o->off_segrange = up(chunks.size() * sizeof(ChunkDesc));
    o->off_segnumel = up(o->off_segrange + segrange.size() * sizeof(int32_t));
    o->off_partial = up(o->off_segnumel + numel.size() * sizeof(double));
    o->h_tables.assign(o->off_partial, 0);
    memcpy(o->h_tables.data(), chunks.data(), chunks.size() * sizeof(ChunkDesc));
    memcpy(o->h_tables.data() + o->off_segrange, segrange.data(), segrange.size() * sizeof(int32_t));
    memcpy(o->h_tables.data() + o->off_segnumel, numel.data(), numel.size() * sizeof(double));
    *out = o;
    return GDL_OK;
}

void gdl_optim_destroy(gdl_optim_t* o) { delete o; }

size_t gdl_optim_workspace_bytes(const gdl_optim_t* o) {
    if (!o) return 0;
    return o->off_partial + ((size_t)o->nchunks * 2 + (size_t)o->nseg * 2) * sizeof(double);
}

int gdl_optim_stats_len(const gdl_optim_t* o) { return o ? 4 + 2 * o->nseg : 0; }

int gdl_optim_bind_workspace(gdl_optim_t* o, void* ws, size_t ws_bytes, void* stream) {
    GDL_REQUIRE(o && ws, "optim_bind_workspace: null argument");
    if (ws_bytes < gdl_optim_workspace_bytes(o)) {
        set_error("optim_bind_workspace: workspace %zu < %zu", ws_bytes, gdl_optim_workspace_bytes(o));
        return GDL_ERR_WORKSPACE;
    }
    GDL_REQUIRE(((uintptr_t)ws & 15) == 0, "optim_bind_workspace: workspace must be 16-byte aligned");
    hipError_t he = hipMemcpyAsync(ws, o->h_tables.data(), o->h_tables.size(), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (he != hipSuccess) return check_hip(he, "optim_bind_workspace: descriptor upload");
    o->bound_ws = ws;
    return GDL_OK;
}

int gdl_optim_grad_stats(gdl_optim_t* o, const float* grads, float max_norm, float grad_scale, float* stats, void* ws,
                         size_t ws_bytes, void* stream) {
    GDL_REQUIRE(o && grads && stats && ws, "optim_grad_stats: null argument");
    if (ws_bytes < gdl_optim_workspace_bytes(o)) {
        set_error("optim_grad_stats: workspace %zu < %zu", ws_bytes, gdl_optim_workspace_bytes(o));
        return GDL_ERR_WORKSPACE;
    }
    GDL_REQUIRE(((uintptr_t)ws & 15) == 0, "optim_grad_stats: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned char* w = (unsigned char*)ws;
    if (o->bound_ws != ws) {  // descriptor tables -> the head of this workspace (once; ordered on `stream` like the kernels)
        hipError_t he = hipMemcpyAsync(w, o->h_tables.data(), o->h_tables.size(), hipMemcpyHostToDevice, st);
        if (he != hipSuccess) return check_hip(he, "optim_grad_stats: descriptor upload");
        o->bound_ws = ws;
    }
    double* partial = (double*)(w + o->off_partial);
    double* segsum = partial + (size_t)o->nchunks * 2;
    {
        ProfScope prof("gdl::grad_stats_kernel", PROF_HBM, st, (double)o->total * 4.0);
        hipLaunchKernelGGL(grad_stats_kernel, dim3(o->nchunks), dim3(256), 0, st, grads, (const ChunkDesc*)w, partial);
    }
    GDL_CHECK_LAUNCH("grad_stats_kernel");
    hipLaunchKernelGGL(grad_stats_final_kernel, dim3(1), dim3(FIN_NT), 0, st, (const double*)partial,
                       (const int32_t*)(w + o->off_segrange), (const double*)(w + o->off_segnumel), o->nseg, max_norm, grad_scale,
                       stats, segsum);
    GDL_CHECK_LAUNCH("grad_stats_final_kernel");
    return GDL_OK;
}

int gdl_optim_sgd_step(gdl_optim_t* o, float* params, float* grads, float* momentum, const float* stats, float grad_scale,
                       float lr, float mu, float wd, void* stream) {
    GDL_REQUIRE(o && params && grads && momentum, "optim_sgd_step: null argument");
    GDL_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum) & 15) == 0,
                "optim_sgd_step: arenas must be 16-byte aligned");
    const int64_t nv = o->total >> 2;
    int64_t blocks = (nv + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    // (p, g, m read + p, m written = 20 n bytes; the clipped gradient's 4 n-byte write-back only happens when the clip is active --
    // decided on the device, stats[1] -- and is not charged: round 3's 24 n overstated the kernel at 0.96 of the HBM peak)
    ProfScope prof("gdl::sgd_kernel", PROF_HBM, (hipStream_t)stream, (double)o->total * 4.0 * 5);
    hipLaunchKernelGGL(sgd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum, stats,
                       grad_scale, lr, mu, wd, o->total);
    GDL_CHECK_LAUNCH("sgd_kernel");
    return GDL_OK;
}

}  // extern "C"
