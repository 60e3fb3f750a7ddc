// encoder.cpp -- ResNet18 encoder engine: a planned sequence of libgdl_hip kernels on ONE stream.
//
// Mirrors `resnet18(modality, args)` / `ResNet.forward` of /root/reference/models/backbone.py
// (:75-156 topology, :158-201 forward, BasicBlock.forward :52-68) plus the pooling glue of
// AVClassifier_DGL.forward (/root/reference/models/basic_model.py:73-82).  No autograd tape:
// the engine keeps exactly the tensors its own backward needs in a caller-provided workspace
// (sized for 288 GB HBM: nothing is recomputed; the stem keeps its zero-padded NHWC4 input copy for the weight gradient).
//
// Per BasicBlock forward:   y1 = conv1(x) [+BN stats in the conv epilogue]; a1 = relu(bn1(y1));
//                           y2 = conv2(a1); [yd = convd(x)]; z = relu(bn2(y2) + (bnd(yd) | x)).
// Backward:  do2 = dz*(z>0); bn2/bnd backward (two passes each); wgrad; dgrad with the identity /
//            downsample gradient accumulated in the dgrad epilogue.
#include <map>
#include <tuple>
#include <vector>

#include "ops.h"

using namespace gdl;

namespace {

struct Conv {
    int cin, cout, r, s, stride, pad;
    int h, w, p, q;  // input / output spatial size
    int pidx;        // index of the weight in the 60-parameter table
    void* w_krsc = nullptr;
    void* w_crsk = nullptr;
    void* tab_fwd = nullptr;    // gather tables (shared between convolutions of equal geometry)
    void* tab_dgrad = nullptr;
};
struct BN {
    int c;
    int pidx;  // gamma index (beta = pidx + 1)
    int bidx;  // 0..19
    float *scale = nullptr, *shift = nullptr, *mean = nullptr, *rstd = nullptr, *coef = nullptr;
    long long* acc = nullptr;  // [c][2] fixed-point forward statistics (bnacc.h), a slice of gdl_encoder::acc_arena
};
struct Block {
    Conv c1, c2, cd;
    BN b1, b2, bd;
    bool has_ds = false;
    int n, h, w, p, q, cin, cout;
    void *y1 = nullptr, *a1 = nullptr, *y2 = nullptr, *yd = nullptr, *z = nullptr;
    uint8_t* zbits = nullptr;  // sign bits of z (one byte per 16-byte vector): the backward's ReLU mask, applied by the
                               // data gradient that produces the gradient of z (conv_dgrad's relu_bits)
    uint8_t* abits = nullptr;  // the same for a1 = relu(bn1(y1)): conv2's data gradient stores the masked gradient
    const void* xin = nullptr;
};

struct Bump {
    size_t off = 0;
    unsigned char* base = nullptr;
    void* take(size_t bytes) {
        off = align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

}  // namespace

struct gdl_encoder {
    int modality, dtype, B, T, H, W, cin, n_img;
    int esz;  // bytes per activation element
    // stem
    int h0, w0, h1, w1;
    long m0;  // n_img*h0*w0
    BN bn0;
    void *col = nullptr, *w0p = nullptr, *y0 = nullptr, *x1 = nullptr, *ymax = nullptr;
    uint8_t* idx = nullptr;
    std::vector<Block> blocks;
    int hf, wf;  // final map
    // scratch
    void *gA = nullptr, *gE = nullptr, *g0 = nullptr;
    void *gB[2] = {nullptr, nullptr}, *gC[2] = {nullptr, nullptr}, *gD[2] = {nullptr, nullptr};  // by block parity
    // Weight gradients run on an engine-owned side stream, forked from / joined into the caller's
    // stream with events: they are off the dy -> dx dependency chain, so they fill the CUs the
    // (often small) kernels of that chain leave idle.  (Stream priorities were measured and do
    // nothing here: the range is {0, -1} and the step time is the same with any assignment.)  The gradient buffers they read alternate
    // with the block parity; ev_side[p] = "the side stream is done with parity p's buffers".
    hipStream_t side = nullptr;
    bool side_owned = true;  // false: a stream of the caller's (gdl_encoder_borrow_side_stream)
    bool has_side = false;   // (the borrowed stream may be the null stream)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_side[2] = {nullptr, nullptr};
    bool side_pending[2] = {false, false};
    // gdl_encoder_backward_phase: state carried from phase 1 (layer4) to phase 2 (the rest)
    void *bw_dz = nullptr, *bw_spare = nullptr;
    bool bw_premasked = false;  // bw_dz already carries the ReLU mask of its block's output
    int bw_b2_rows = 0;         // > 0: ... and the bn2 (downsample BatchNorm) sums of that block, in this many partial rows
    // BatchNorm-backward sums written by the data gradients' epilogues (ops.h BwdStats): A = bn1 of the current block (from
    // conv2's data gradient), B / B2 = bn2 / downsample BatchNorm of the block before (from conv1's data gradient)
    float *bwA = nullptr, *bwB = nullptr, *bwB2 = nullptr;
    long bw_serial = -1;  // serial of the forward whose phase 1 ran
    ~gdl_encoder() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (int p = 0; p < 2; ++p)
            if (ev_side[p]) (void)hipEventDestroy(ev_side[p]);
        if (side && side_owned) (void)hipStreamDestroy(side);
    }
    float *bn_partial = nullptr, *bn_partial2 = nullptr, *bnb_partial = nullptr, *bnb_partial2 = nullptr;
    SplitWs sk{nullptr, 0};          // split-K workspace of the slab convolutions (ops.h): forward / data-gradient chain only
    long long* acc_arena = nullptr;  // the 20 BatchNorms' integer accumulators, contiguous: one memset per forward
    size_t acc_bytes = 0;
    void* wg_ws = nullptr;
    size_t wg_ws_bytes = 0, bn_partial_floats = 0, bnb_partial_floats = 0;
    size_t ws_bytes = 0;
    void* ws = nullptr;
    // bound tables
    const float* params[GDL_ENC_NPARAMS] = {nullptr};
    float* rmean[GDL_ENC_NBN] = {nullptr};
    float* rvar[GDL_ENC_NBN] = {nullptr};
    int64_t* nbt[GDL_ENC_NBN] = {nullptr};
    bool params_set = false;
    int64_t serial = 0;
    bool have_train_fwd = false;
    bool acc_last = false;  // the last forward ran its statistics through the integer accumulators (their flag words are valid)
    int64_t numel[GDL_ENC_NPARAMS];
    // gather tables: geometry -> workspace slot; built lazily on the first forward's stream
    struct TabJob {
        int mode, N, H, W, C, K, R, S, stride, pad;
        void* dst;
    };
    std::vector<TabJob> tab_jobs;
    bool tabs_dirty = true;
    void* tab_stem = nullptr;
    // batched weight packing
    std::vector<PackDescHost> pack_host;
    void* pack_dev = nullptr;
    bool pack_dirty = true;
    int pack_blocks = 0;
    double pack_bytes = 0.0;

    size_t plan(unsigned char* base);
};

// lays out every buffer in the workspace; with base == nullptr only measures
size_t gdl_encoder::plan(unsigned char* base) {
    Bump b;
    b.base = base;
    const size_t e = (size_t)esz;
    col = b.take(stem_pad_bytes(dtype, n_img, H, W));  // zero-padded NHWC4 copy of the input (direct stem)
    w0p = b.take((size_t)64 * stem_taps(dtype) * stem_ic(dtype) * e);
    y0 = b.take((size_t)m0 * 64 * e);
    x1 = b.take((size_t)n_img * h1 * w1 * 64 * e);
    idx = (uint8_t*)b.take((size_t)n_img * h1 * w1 * 64);
    ymax = b.take((size_t)n_img * h1 * w1 * 64 * e);  // raw stem output at each pooling window's argmax
    pack_dev = b.take(32 * sizeof(PackDescHost));
    // (per BatchNorm [c][2] sums + its overflow flag word, padded to 16 bytes: bnacc.h)
    size_t acc_ch = 64 + 1;
    for (const Block& k : blocks) acc_ch += (size_t)(k.cout + 1) * (k.has_ds ? 3 : 2);
    acc_bytes = acc_ch * 2 * sizeof(long long);
    acc_arena = (long long*)b.take(acc_bytes);
    size_t acc_used = 0;
    auto bn_alloc = [&](BN& n) {
        n.acc = acc_arena ? acc_arena + acc_used : nullptr;
        acc_used += 2 * (size_t)n.c + 2;
        n.scale = (float*)b.take(sizeof(float) * n.c);
        n.shift = (float*)b.take(sizeof(float) * n.c);
        n.mean = (float*)b.take(sizeof(float) * n.c);
        n.rstd = (float*)b.take(sizeof(float) * n.c);
        n.coef = (float*)b.take(sizeof(float) * 2 * n.c);
    };
    auto conv_alloc = [&](Conv& c) {
        const size_t n = (size_t)c.cout * c.cin * c.r * c.s * e;
        c.w_krsc = b.take(n);
        c.w_crsk = b.take(n);
    };
    bn_alloc(bn0);
    // gather tables, one per distinct (mode, geometry)
    tab_jobs.clear();
    std::map<std::tuple<int, int, int, int, int, int, int, int, int>, void*> seen;
    auto table_for = [&](int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) -> void* {
        const int src_ch = mode == GATHER_FWD ? C : K;  // the table depends on the gathered tensor's row size
        // (a stride-2 data-gradient table is permuted for the M-tile its convolution runs with, which
        // depends on the other channel count too: fold it into the key)
        auto key = std::make_tuple(mode, N, H, W, (mode == GATHER_DGRAD && stride == 2) ? src_ch * 4096 + C : src_ch, R, S,
                                   stride, pad);
        auto it = seen.find(key);
        if (it != seen.end()) return it->second;
        void* dst = b.take(gather_table_bytes(mode, N, H, W, R, S, stride, pad));
        seen[key] = dst;
        tab_jobs.push_back(TabJob{mode, N, H, W, C, K, R, S, stride, pad, dst});
        return dst;
    };
    tab_stem = b.take((size_t)m0 * sizeof(GatherEntry));
    size_t max_act = (size_t)n_img * h1 * w1 * 64;  // elements
    size_t max_tiles_c = (size_t)conv_stem_tiles_m(dtype, n_img, H, W) * 64;
    size_t max_bnb = (size_t)bn_bwd_blocks((size_t)m0, 64) * 64;
    size_t max_bwt = 64;
    size_t wg = conv_stem_wgrad_ws_bytes(n_img, H, W);
    const void* prev = x1;
    for (Block& k : blocks) {
        const size_t out_el = (size_t)k.n * k.p * k.q * k.cout;
        k.xin = prev;
        k.y1 = b.take(out_el * e);
        k.a1 = b.take(out_el * e);
        k.y2 = b.take(out_el * e);
        k.z = b.take(out_el * e);
        k.zbits = (uint8_t*)b.take(out_el * e / 16);
        k.abits = (uint8_t*)b.take(out_el * e / 16);
        conv_alloc(k.c1);
        conv_alloc(k.c2);
        for (Conv* c : {&k.c1, &k.c2}) {
            c->tab_fwd = table_for(GATHER_FWD, k.n, c->h, c->w, c->cin, c->cout, c->r, c->s, c->stride, c->pad);
            c->tab_dgrad = table_for(GATHER_DGRAD, k.n, c->h, c->w, c->cin, c->cout, c->r, c->s, c->stride, c->pad);
        }
        bn_alloc(k.b1);
        bn_alloc(k.b2);
        if (k.has_ds) {
            k.yd = b.take(out_el * e);
            conv_alloc(k.cd);
            k.cd.tab_fwd = table_for(GATHER_FWD, k.n, k.cd.h, k.cd.w, k.cd.cin, k.cd.cout, 1, 1, k.cd.stride, 0);
            k.cd.tab_dgrad = table_for(GATHER_DGRAD, k.n, k.cd.h, k.cd.w, k.cd.cin, k.cd.cout, 1, 1, k.cd.stride, 0);
            bn_alloc(k.bd);
        }
        const size_t in_el = (size_t)k.n * k.h * k.w * k.cin;
        if (out_el > max_act) max_act = out_el;
        if (in_el > max_act) max_act = in_el;
        const int M = k.n * k.p * k.q;
        for (const Conv* c : {&k.c1, &k.c2, &k.cd}) {
            if (c == &k.cd && !k.has_ds) continue;
            const size_t tc = (size_t)conv_tiles_m(dtype, k.n, c->h, c->w, c->cin, c->cout, c->r, c->s, c->stride, c->pad) *
                              c->cout;
            if (tc > max_tiles_c) max_tiles_c = tc;
        }
        const size_t bb = (size_t)bn_bwd_blocks((size_t)M, k.cout) * k.cout;
        if (bb > max_bnb) max_bnb = bb;
        const size_t ta = (size_t)conv_dgrad_tiles_m(dtype, k.n, k.c2.h, k.c2.w, k.c2.cin, k.c2.cout, 3, 3, 1, 1) * k.cout;
        const size_t tb = (size_t)conv_dgrad_tiles_m(dtype, k.n, k.c1.h, k.c1.w, k.c1.cin, k.c1.cout, 3, 3, k.c1.stride, 1) * k.cin;
        if (ta > max_bwt) max_bwt = ta;
        if (tb > max_bwt) max_bwt = tb;
        size_t w1 = conv_wgrad_ws_bytes(M, k.cin, k.cout, 9);
        size_t w2 = conv_wgrad_ws_bytes(M, k.cout, k.cout, 9);
        size_t w3 = k.has_ds ? conv_wgrad_ws_bytes(M, k.cin, k.cout, 1) : 0;
        if (w1 > wg) wg = w1;
        if (w2 > wg) wg = w2;
        if (w3 > wg) wg = w3;
        prev = k.z;
    }
    gA = b.take(max_act * e);
    for (int p = 0; p < 2; ++p) {
        gB[p] = b.take(max_act * e);
        gC[p] = b.take(max_act * e);
        gD[p] = b.take(max_act * e);
    }
    gE = b.take(max_act * e);
    // the gradient of the stem output exists only on the two-launch stem backward; the fused one (bf16, output rows of at least 64
    // pixels: the benchmark shapes) never stores it -- 308 + 99 MB of workspace less at B = 64
    g0 = stem_bwd_fused_ok(dtype, W) ? nullptr : b.take((size_t)m0 * 64 * e);
    bn_partial_floats = max_tiles_c * 2;
    bnb_partial_floats = max_bnb * 2;
    bn_partial = (float*)b.take(bn_partial_floats * sizeof(float));
    bn_partial2 = (float*)b.take(bn_partial_floats * sizeof(float));
    bnb_partial = (float*)b.take(bnb_partial_floats * sizeof(float));
    bnb_partial2 = (float*)b.take(bnb_partial_floats * sizeof(float));
    bwA = (float*)b.take(max_bwt * 2 * sizeof(float));
    bwB = (float*)b.take(max_bwt * 2 * sizeof(float));
    bwB2 = (float*)b.take(max_bwt * 2 * sizeof(float));
    wg_ws_bytes = wg;
    wg_ws = b.take(wg);
    // Split-K of the under-filled slab convolutions (ops.h SplitWs), tuning aid GDL_SPLITK=1 -- OFF by default.  Measured (B = 64):
    // alone the audio layer-4 convolutions go from 54-64 us to 26 us + a 15 us finish pass (their K-steps are latency-bound: 1 400
    // clk per step of which 190 are MFMAs, one block on 108 of 256 CUs), inside the step nothing: 5.752 vs 5.752 ms with the audio
    // layer 4 split four ways (four A/B rounds), 5.69 vs 5.64 ms with every layer below 200 tiles split (the visual layer 4 and its
    // 38 MB of partials included) -- the CUs such a launch leaves idle are not idle in the step, the other streams' blocks run there.
    static int splitk = -1;
    if (splitk < 0) {
        const char* env = tune_env("GDL_SPLITK");
        splitk = env ? atoi(env) : 0;
    }
    if (splitk) {
        size_t need = 0;
        for (const Block& k : blocks)
            for (const Conv* c : {&k.c1, &k.c2})
                for (int dg = 0; dg < 2; ++dg) {
                    const size_t nb = conv_split_ws_bytes(dtype, k.n, c->h, c->w, c->cin, c->cout, c->r, c->s, c->stride, c->pad, dg);
                    need = nb > need ? nb : need;
                }
        sk.bytes = need;
        sk.ptr = need ? b.take(need) : nullptr;
    }
    return align_up(b.off, 256);
}

extern "C" {

int gdl_encoder_create(gdl_encoder_t** out, int modality, int dtype, int B, int T, int H, int W) {
    GDL_REQUIRE(out, "encoder_create: null out");
    GDL_REQUIRE(modality == GDL_AUDIO || modality == GDL_VISUAL, "encoder_create: modality %d", modality);
    GDL_REQUIRE(dtype == GDL_F32 || dtype == GDL_BF16, "encoder_create: dtype %d", dtype);
    GDL_REQUIRE(B > 0 && T > 0 && H >= 7 && W >= 7, "encoder_create: bad shape B=%d T=%d H=%d W=%d", B, T, H, W);
    GDL_REQUIRE(modality == GDL_VISUAL || T == 1, "encoder_create: audio takes T=1");
    gdl_encoder* e = new gdl_encoder();
    e->modality = modality;
    e->dtype = dtype;
    e->B = B;
    e->T = T;
    e->H = H;
    e->W = W;
    e->cin = modality == GDL_AUDIO ? 1 : 3;  // backbone.py:96-101
    e->n_img = B * T;
    e->esz = dtype == GDL_BF16 ? 2 : 4;
    e->h0 = (H + 6 - 7) / 2 + 1;
    e->w0 = (W + 6 - 7) / 2 + 1;
    e->h1 = (e->h0 - 1) / 2 + 1;  // MaxPool2d(3,2,1)
    e->w1 = (e->w0 - 1) / 2 + 1;
    e->m0 = (long)e->n_img * e->h0 * e->w0;
    if (e->m0 >= (1L << 24)) {
        set_error("encoder_create: stem output has %ld pixels (limit 2^24); lower the batch", e->m0);
        delete e;
        return GDL_ERR_ARG;
    }
    int pi = 0, bi = 0;
    e->numel[pi] = (int64_t)64 * e->cin * 49;
    pi++;  // conv1.weight
    e->bn0.c = 64;
    e->bn0.pidx = pi;
    e->bn0.bidx = bi++;
    e->numel[pi] = 64;
    e->numel[pi + 1] = 64;
    pi += 2;
    int inpl = 64, h = e->h1, w = e->w1;
    const int planes_of[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li) {  // _make_layer, backbone.py:134-156
        for (int bk = 0; bk < 2; ++bk) {
            Block k;
            const int planes = planes_of[li];
            const int stride = (bk == 0 && li > 0) ? 2 : 1;
            k.has_ds = (stride != 1 || inpl != planes);
            k.n = e->n_img;
            k.h = h;
            k.w = w;
            k.cin = inpl;
            k.cout = planes;
            k.p = (h + 2 - 3) / stride + 1;
            k.q = (w + 2 - 3) / stride + 1;
            k.c1 = Conv{inpl, planes, 3, 3, stride, 1, h, w, k.p, k.q, pi};
            e->numel[pi] = (int64_t)planes * inpl * 9;
            k.b1.c = planes;
            k.b1.pidx = pi + 1;
            k.b1.bidx = bi++;
            e->numel[pi + 1] = e->numel[pi + 2] = planes;
            k.c2 = Conv{planes, planes, 3, 3, 1, 1, k.p, k.q, k.p, k.q, pi + 3};
            e->numel[pi + 3] = (int64_t)planes * planes * 9;
            k.b2.c = planes;
            k.b2.pidx = pi + 4;
            k.b2.bidx = bi++;
            e->numel[pi + 4] = e->numel[pi + 5] = planes;
            pi += 6;
            if (k.has_ds) {
                k.cd = Conv{inpl, planes, 1, 1, stride, 0, h, w, k.p, k.q, pi};
                e->numel[pi] = (int64_t)planes * inpl;
                k.bd.c = planes;
                k.bd.pidx = pi + 1;
                k.bd.bidx = bi++;
                e->numel[pi + 1] = e->numel[pi + 2] = planes;
                pi += 3;
            }
            e->blocks.push_back(k);
            inpl = planes;
            h = k.p;
            w = k.q;
        }
    }
    e->hf = h;
    e->wf = w;
    if (pi != GDL_ENC_NPARAMS || bi != GDL_ENC_NBN) {
        delete e;
        set_error("encoder_create: internal topology error (%d params, %d bn)", pi, bi);
        return GDL_ERR_STATE;
    }
    e->ws_bytes = e->plan(nullptr);
    *out = e;
    return GDL_OK;
}

void gdl_encoder_destroy(gdl_encoder_t* e) { delete e; }

static int side_events(gdl_encoder_t* e) {  // fork / join / per-parity events, created once
    hipError_t he = hipSuccess;
    if (!e->ev_fork) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    if (he == hipSuccess && !e->ev_join) he = hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming);
    for (int p = 0; p < 2 && he == hipSuccess; ++p)
        if (!e->ev_side[p]) he = hipEventCreateWithFlags(&e->ev_side[p], hipEventDisableTiming);
    return he == hipSuccess ? GDL_OK : check_hip(he, "encoder side stream: events");
}

int gdl_encoder_side_stream(gdl_encoder_t* e, int enable) {
    GDL_REQUIRE(e, "encoder_side_stream: null");
    if (enable && !e->has_side) {
        int rc = side_events(e);
        if (rc) return rc;
        hipError_t he = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking);
        if (he != hipSuccess) return check_hip(he, "encoder_side_stream");
        e->side_owned = true;
        e->has_side = true;
    } else if (!enable && e->has_side) {
        hipError_t he = hipStreamSynchronize(e->side);
        if (he != hipSuccess) return check_hip(he, "encoder_side_stream: sync");
        if (e->side_owned) (void)hipStreamDestroy(e->side);
        e->side = nullptr;
        e->side_owned = true;
        e->has_side = false;
    }
    return GDL_OK;
}

int gdl_encoder_borrow_side_stream(gdl_encoder_t* e, void* stream) {
    GDL_REQUIRE(e && !(e->has_side && e->side_owned), "encoder_borrow_side_stream: null, or the engine owns a side stream");
    int rc = side_events(e);
    if (rc) return rc;
    e->side = (hipStream_t)stream;  // (NULL = the null stream: has_side says whether there is one)
    e->side_owned = false;
    e->has_side = true;
    return GDL_OK;
}

size_t gdl_encoder_workspace_bytes(const gdl_encoder_t* e) { return e ? e->ws_bytes : 0; }

int gdl_encoder_param_numel(const gdl_encoder_t* e, int64_t* numel) {
    GDL_REQUIRE(e && numel, "encoder_param_numel: null");
    for (int i = 0; i < GDL_ENC_NPARAMS; ++i) numel[i] = e->numel[i];
    return GDL_OK;
}

int gdl_encoder_out_shape(const gdl_encoder_t* e, int* n_img, int* h, int* w) {
    GDL_REQUIRE(e, "encoder_out_shape: null");
    if (n_img) *n_img = e->n_img;
    if (h) *h = e->hf;
    if (w) *w = e->wf;
    return GDL_OK;
}

int gdl_encoder_bind(gdl_encoder_t* e, void* workspace, size_t bytes) {
    GDL_REQUIRE(e && workspace, "encoder_bind: null");
    if (bytes < e->ws_bytes) {
        set_error("encoder_bind: workspace %zu < %zu bytes", bytes, e->ws_bytes);
        return GDL_ERR_WORKSPACE;
    }
    GDL_REQUIRE(((uintptr_t)workspace & 255) == 0, "encoder_bind: workspace must be 256-byte aligned");
    e->ws = workspace;
    e->plan((unsigned char*)workspace);
    e->pack_dirty = true;
    e->tabs_dirty = true;
    e->have_train_fwd = false;
    return GDL_OK;
}

int gdl_encoder_set_params(gdl_encoder_t* e, const float* const* params, float* const* running_mean,
                           float* const* running_var, int64_t* const* num_batches_tracked) {
    GDL_REQUIRE(e && params && running_mean && running_var, "encoder_set_params: null");
    for (int i = 0; i < GDL_ENC_NPARAMS; ++i) {
        GDL_REQUIRE(params[i], "encoder_set_params: params[%d] is null", i);
        e->params[i] = params[i];
    }
    for (int i = 0; i < GDL_ENC_NBN; ++i) {
        GDL_REQUIRE(running_mean[i] && running_var[i], "encoder_set_params: running stats[%d] null", i);
        e->rmean[i] = running_mean[i];
        e->rvar[i] = running_var[i];
        e->nbt[i] = num_batches_tracked ? num_batches_tracked[i] : nullptr;
    }
    e->params_set = true;
    e->pack_dirty = true;
    return GDL_OK;
}

int64_t gdl_encoder_forward_serial(const gdl_encoder_t* e) { return e ? e->serial : -1; }

// Number of BatchNorms of the LAST training forward whose fixed-point statistics exceeded their headroom (bnacc.h "Guard"; their
// statistics were turned into NaN).  ReLU swallows NaN (max(NaN, 0) = 0), so the activations alone would not show it: callers
// that read results back (DGLTrainer.read) ask here.  Synchronises `stream`.  < 0: error code.
int gdl_encoder_bn_overflow(gdl_encoder_t* e, void* stream) {
    GDL_REQUIRE(e, "encoder_bn_overflow: null encoder");
    if (!e->acc_arena || !e->acc_last) return 0;
    std::vector<long long> host(e->acc_bytes / sizeof(long long));
    hipError_t he = hipMemcpyAsync(host.data(), e->acc_arena, e->acc_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    if (he != hipSuccess) return -check_hip(he, "encoder_bn_overflow: read-back");
    int n = 0;
    auto flagged = [&](const BN& b) {
        if (b.acc && host[(size_t)(b.acc - e->acc_arena) + 2 * (size_t)b.c] != 0) ++n;
    };
    flagged(e->bn0);
    for (const Block& k : e->blocks) {
        flagged(k.b1);
        flagged(k.b2);
        if (k.has_ds) flagged(k.bd);
    }
    return n;
}

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__; \
    } while (0)

// Timing experiments only (a -DGDL_EXPERIMENT build, never the product library): GDL_SKIP is a bit mask of launches left out
// -- the results are then WRONG; what is measured is the bound on what removing / fusing that pass could return
// (tools/README.md: "skip bounds").  bits: 1 bn_act of a1 (all layers), 2 the same for 64-channel layers only, 4 bn1's backward
// apply, 8 bn2's / the downsample BatchNorm's backward apply, 16 forward finalize, 32 backward finalize, 64 the stem's
// maxpool_bn_bwd_apply, 128 the stem's bn_relu_maxpool, 256 the weight pack; round 5, convolution classes (what a rebuilt kernel
// of that class could return at most): 1024 the stride-2 3x3 data gradients (+ folded shortcut), 2048 layer 4's stride-1 forward
// convolutions, 4096 layer 4's stride-1 data gradients, 8192 the stride-2 / 1x1 weight gradients, 16384 the 9-tap weight gradients,
// 32768 the stride-2 / 1x1 forward convolutions, 65536 every stride-1 3x3 forward, 131072 every stride-1 3x3 data gradient,
// 262144 the backward's BatchNorm finalize launches only (= bit 32), 524288 the stem's forward convolution, 1048576 the stem's
// weight gradient
#ifdef GDL_EXPERIMENT
static unsigned skip_mask() {
    static long v = -1;
    if (v < 0) {
        const char* env = getenv("GDL_SKIP");
        v = env ? atol(env) : 0;
    }
    return (unsigned)v;
}
static long g_skip_serial = 0;  // (skips start after a dozen complete steps, so that every buffer a skipped pass would have written holds sane -- stale -- values)
#define GDL_SKIPPED(bit) (g_skip_serial > 12 && (skip_mask() & (bit)) != 0)
#else
#define GDL_SKIPPED(bit) false
#endif

static int bn_finalize(gdl_encoder* e, BN& n, int training, int tiles, double count, hipStream_t st) {
    if (GDL_SKIPPED(16)) return GDL_OK;
    const float* gamma = e->params[n.pidx];
    const float* beta = e->params[n.pidx + 1];
    if (training)
        return bn_finalize_train(e->bn_partial, tiles, n.c, count, gamma, beta, 1e-5f, 0.1f, e->rmean[n.bidx],
                                 e->rvar[n.bidx], e->nbt[n.bidx], n.mean, n.rstd, n.scale, n.shift, st);
    return bn_finalize_eval(n.c, gamma, beta, 1e-5f, e->rmean[n.bidx], e->rvar[n.bidx], n.scale, n.shift, st);
}

// (The in-launch BatchNorm finalize -- "the last block folds", GDL_FOLD / GDL_PERS_FOLD -- lost twice (6.69 vs 6.22 ms in round 1;
// 5.855 vs 5.841 ms on the persistent kernels in round 2) and was removed in round 4: tools/experiments/r4_pruned_fold.diff.txt.)
// ReLU mask of a block output applied by the producing data gradient (default on; GDL_PREMASK=0: the block masks itself)
static bool ds_fold_on() {
    static int v = -1;
    if (v < 0) {
        const char* env = tune_env("GDL_DS_FOLD");  // tuning aid: 0 = separate 1x1 data gradient + addend
        v = env ? atoi(env) : 1;
    }
    return v != 0;
}
static bool premask_on() {
    static int v = -1;
    if (v < 0) {
        const char* env = tune_env("GDL_PREMASK");
        v = env ? atoi(env) : 1;
    }
    return v != 0;
}

// BatchNorm-backward reductions in the producing data gradient's epilogue (ops.h BwdStats; default on; GDL_BW_FUSE=0: the
// separate bn_bwd_reduce / block_bwd_reduce passes over g and y)
static bool bw_fuse_on() {
    static int v = -1;
    if (v < 0) {
        const char* env = tune_env("GDL_BW_FUSE");
        v = env ? atoi(env) : 1;
    }
    return v != 0;
}

static bool separate_stats() {
    static int sep = -1;
    if (sep < 0) {
        const char* env = tune_env("GDL_SEPARATE_STATS");  // tuning aid: statistics by a separate pass over y
        sep = env ? atoi(env) : 0;
    }
    return sep != 0;
}

// Forward BatchNorm statistics through integer accumulators (bnacc.h): no finalize launches in the forward; default on
// (GDL_BN_ACC=0: fp32 partial rows + finalize kernels).  Excluded by the in-launch fold variants and the separate pass.
static bool bn_acc_on() {
    static int v = -1;
    if (v < 0) {
        const char* env = tune_env("GDL_BN_ACC");
        v = env ? atoi(env) : 1;
    }
    return v != 0;
}
static BnAcc acc_producer(const BN& n, size_t rows) {
    BnAcc a{n.acc, 0.0, 0.0, n.acc ? n.acc + 2 * (size_t)n.c : nullptr};
    bn_acc_scales(rows, &a.s1, &a.s2);
    return a;
}
static BnAccFin acc_consumer(gdl_encoder* e, BN& n, size_t rows) {
    double s1, s2;
    bn_acc_scales(rows, &s1, &s2);
    return BnAccFin{n.acc, 1.0 / s1, 1.0 / s2, (double)rows, e->params[n.pidx], e->params[n.pidx + 1], e->rmean[n.bidx],
                    e->rvar[n.bidx], e->nbt[n.bidx], n.mean, n.rstd, n.scale, n.shift, 1e-5f, 0.1f,
                    n.acc ? n.acc + 2 * (size_t)n.c : nullptr};
}

static BnFinTrain fin_train_args(gdl_encoder* e, BN& n, const float* partial, int tiles, double count) {
    return BnFinTrain{partial, tiles, n.c, count, e->params[n.pidx], e->params[n.pidx + 1], e->rmean[n.bidx], e->rvar[n.bidx],
                      e->nbt[n.bidx], n.mean, n.rstd, n.scale, n.shift};
}

static int conv_bn(gdl_encoder* e, Conv& c, BN& n, const void* x, void* y, int nimg, int training, hipStream_t st,
                   float* partial = nullptr) {
    const bool sep = separate_stats();
    if (!partial) partial = e->bn_partial;
    const int M = nimg * c.p * c.q;
    if (sep && training) {
        RC(conv_fwd(e->dtype, x, c.w_krsc, y, nullptr, c.tab_fwd, nimg, c.h, c.w, c.cin, c.cout, c.r, c.s, c.stride, c.pad, st));
        RC(bn_stats(e->dtype, y, partial, M, c.cout, st));
        return bn_finalize(e, n, training, bn_stats_tiles(M), (double)M, st);
    }
    const int tiles = conv_tiles_m(e->dtype, nimg, c.h, c.w, c.cin, c.cout, c.r, c.s, c.stride, c.pad);
    RC(conv_fwd(e->dtype, x, c.w_krsc, y, training ? partial : nullptr, c.tab_fwd, nimg, c.h, c.w, c.cin, c.cout, c.r, c.s,
                c.stride, c.pad, st, nullptr, &e->sk));
    return bn_finalize(e, n, training, tiles, (double)M, st);
}

int gdl_encoder_forward(gdl_encoder_t* e, const float* x, int training, float* feat_out, float* fmap_nchw,
                        void* stream) {
    GDL_REQUIRE(e && x, "encoder_forward: null");
    GDL_REQUIRE(e->ws, "encoder_forward: no workspace bound");
    GDL_REQUIRE(e->params_set, "encoder_forward: parameters not set");
    hipStream_t st = (hipStream_t)stream;
    const int dt = e->dtype;
#ifdef GDL_EXPERIMENT
    g_skip_serial = (long)e->serial;
#endif
    if (!training) e->have_train_fwd = false;  // an eval pass overwrites the saved activations
    // weights -> kernel layouts (float32 master copies stay with the caller)
    RC(pack_stem_rows(dt, e->params[0], e->w0p, e->cin, st));
    if (e->tabs_dirty) {  // gather tables: once per bound workspace
        RC(build_stem_table(dt, e->n_img, e->H, e->W, stem_taps(dt), (GatherEntry*)e->tab_stem, st));
        for (const gdl_encoder::TabJob& j : e->tab_jobs)
            RC(build_gather_table(j.mode, dt, j.N, j.H, j.W, j.C, j.K, j.R, j.S, j.stride, j.pad, (GatherEntry*)j.dst, st));
        e->tabs_dirty = false;
    }
    if (e->pack_dirty) {  // (re)build the descriptor table of the batched packing launch
        e->pack_host.clear();
        int blk = 0;
        double bytes = 0.0;
        auto add = [&](const Conv& c) {
            PackDescHost d;
            d.w = e->params[c.pidx];
            d.krsc = c.w_krsc;
            d.crsk = c.w_crsk;
            d.K = c.cout;
            d.C = c.cin;
            d.RS = c.r * c.s;
            d.blk0 = blk;
            const size_t total = (size_t)c.cout * c.cin * c.r * c.s;
            blk += (c.cout / 32) * (c.cin / 32);  // one block per 32x32 (k, c) patch
            bytes += (double)total * (4.0 + 2.0 * e->esz);
            e->pack_host.push_back(d);
        };
        for (Block& k : e->blocks) {
            add(k.c1);
            add(k.c2);
            if (k.has_ds) add(k.cd);
        }
        e->pack_blocks = blk;
        e->pack_bytes = bytes;
        hipError_t he = hipMemcpyAsync(e->pack_dev, e->pack_host.data(), e->pack_host.size() * sizeof(PackDescHost),
                                       hipMemcpyHostToDevice, st);
        if (he != hipSuccess) return check_hip(he, "encoder_forward: descriptor upload");
        e->pack_dirty = false;
    }
    // (Packing on the side stream, beside the stem -- GDL_PACK_SIDE -- returned 0.4 %, inside the noise, against a skip bound of
    // 0.12 ms: tools/experiments/r5_pruned_knobs.diff.txt)
    if (!GDL_SKIPPED(256))
        RC(pack_weights_batched(dt, e->pack_dev, (int)e->pack_host.size(), e->pack_blocks, e->pack_bytes, st));
    const bool acc = training && bn_acc_on() && !separate_stats();
    e->acc_last = acc;
    // stem: conv1 (7x7/2) as a direct implicit GEMM over the padded input, bn1, relu, maxpool   (backbone.py:166-173 / 186-189)
    // (the padding launch also clears the BatchNorm accumulators of this forward)
    if (GDL_SKIPPED(4194304)) {  // (experiment: the input staging pass -- the bound of staging the next batch behind the previous step)
        if (acc) (void)hipMemsetAsync(e->acc_arena, 0, e->acc_bytes, st);
    } else
    RC(stem_pad(dt, x, e->col, e->B, e->cin, e->T, e->H, e->W, st, acc ? e->acc_arena : nullptr, acc ? e->acc_bytes : 0));
    if (acc) {
        const BnAcc pa = acc_producer(e->bn0, e->m0);
        if (!GDL_SKIPPED(524288))
        RC(conv_stem_fwd(dt, e->col, e->w0p, e->y0, nullptr, e->tab_stem, e->n_img, e->H, e->W, e->cin, st, &pa));
    } else {
        const int tiles = conv_stem_tiles_m(dt, e->n_img, e->H, e->W);
        RC(conv_stem_fwd(dt, e->col, e->w0p, e->y0, training ? e->bn_partial : nullptr, e->tab_stem, e->n_img, e->H, e->W, e->cin,
                         st));
        RC(bn_finalize(e, e->bn0, training, tiles, (double)e->m0, st));
    }
    if (!GDL_SKIPPED(128)) {
        const BnAccFin f0 = acc ? acc_consumer(e, e->bn0, e->m0) : BnAccFin{};
        RC(bn_relu_maxpool_fwd(dt, e->y0, e->bn0.scale, e->bn0.shift, e->x1, e->idx, training ? e->ymax : nullptr, e->n_img,
                               e->h0, e->w0, 64, st, &f0));
    }
    // layer1..layer4   (backbone.py:175-178; BasicBlock.forward :52-68)
    for (Block& k : e->blocks) {
        const size_t Mo = (size_t)k.n * k.p * k.q;
        if (acc) {  // convolutions add their tiles' sums to the accumulators; the applies derive the constants themselves
            auto conv = [&](const Conv& c, const BN& n, const void* x, void* y) {
                if (c.stride == 1 && c.r == 3 && (GDL_SKIPPED(65536) || (GDL_SKIPPED(2048) && c.cout == 512))) return (int)GDL_OK;
                if ((c.stride == 2 || c.r == 1) && GDL_SKIPPED(32768)) return (int)GDL_OK;
                if (c.r == 1 && GDL_SKIPPED(2097152)) return (int)GDL_OK;  // the shortcut's 1x1 convolutions only
                const BnAcc pa = acc_producer(n, Mo);
                return conv_fwd(dt, x, c.w_krsc, y, nullptr, c.tab_fwd, k.n, c.h, c.w, c.cin, c.cout, c.r, c.s, c.stride, c.pad, st,
                                &pa, &e->sk);
            };
            RC(conv(k.c1, k.b1, k.xin, k.y1));
            const BnAccFin f1 = acc_consumer(e, k.b1, Mo), f2 = acc_consumer(e, k.b2, Mo);
            RC(bn_act(dt, k.y1, k.b1.scale, k.b1.shift, nullptr, nullptr, nullptr, 1, k.a1, Mo, k.cout, st,
                      bw_fuse_on() ? k.abits : nullptr, &f1));
            RC(conv(k.c2, k.b2, k.a1, k.y2));
            if (k.has_ds) {
                RC(conv(k.cd, k.bd, k.xin, k.yd));
                const BnAccFin fd = acc_consumer(e, k.bd, Mo);
                RC(bn_act(dt, k.y2, k.b2.scale, k.b2.shift, k.yd, k.bd.scale, k.bd.shift, 1, k.z, Mo, k.cout, st, k.zbits, &f2, &fd));
            } else {
                RC(bn_act(dt, k.y2, k.b2.scale, k.b2.shift, k.xin, nullptr, nullptr, 1, k.z, Mo, k.cout, st, k.zbits, &f2));
            }
            continue;
        }
        RC(conv_bn(e, k.c1, k.b1, k.xin, k.y1, k.n, training, st));
        if (!(GDL_SKIPPED(1) || (GDL_SKIPPED(2) && k.cout == 64)))
            RC(bn_act(dt, k.y1, k.b1.scale, k.b1.shift, nullptr, nullptr, nullptr, 1, k.a1, Mo, k.cout, st,
                      (training && bw_fuse_on()) ? k.abits : nullptr));
        if (k.has_ds && training && !separate_stats()) {
            // bn2 and the downsample BatchNorm are independent: both convolutions first, ONE finalize launch for the two
            // (a finalize kernel costs the chain its whole ~6 us; 80 of them were 0.56 ms of the step)
            auto fin = [&](const Conv& c, BN& n, const float* partial) {
                return BnFinTrain{partial, conv_tiles_m(dt, k.n, c.h, c.w, c.cin, c.cout, c.r, c.s, c.stride, c.pad), n.c,
                                  (double)Mo, e->params[n.pidx], e->params[n.pidx + 1], e->rmean[n.bidx], e->rvar[n.bidx],
                                  e->nbt[n.bidx], n.mean, n.rstd, n.scale, n.shift};
            };
            RC(conv_fwd(dt, k.a1, k.c2.w_krsc, k.y2, e->bn_partial, k.c2.tab_fwd, k.n, k.c2.h, k.c2.w, k.c2.cin, k.c2.cout,
                        k.c2.r, k.c2.s, k.c2.stride, k.c2.pad, st));
            RC(conv_fwd(dt, k.xin, k.cd.w_krsc, k.yd, e->bn_partial2, k.cd.tab_fwd, k.n, k.cd.h, k.cd.w, k.cd.cin, k.cd.cout,
                        k.cd.r, k.cd.s, k.cd.stride, k.cd.pad, st));
            if (!GDL_SKIPPED(16))
                RC(bn_finalize_train_pair(fin(k.c2, k.b2, e->bn_partial), fin(k.cd, k.bd, e->bn_partial2), 1e-5f, 0.1f, st));
            RC(bn_act(dt, k.y2, k.b2.scale, k.b2.shift, k.yd, k.bd.scale, k.bd.shift, 1, k.z, Mo, k.cout, st,
                      training ? k.zbits : nullptr));
            continue;
        }
        RC(conv_bn(e, k.c2, k.b2, k.a1, k.y2, k.n, training, st));
        if (k.has_ds) {
            RC(conv_bn(e, k.cd, k.bd, k.xin, k.yd, k.n, training, st, e->bn_partial2));
            RC(bn_act(dt, k.y2, k.b2.scale, k.b2.shift, k.yd, k.bd.scale, k.bd.shift, 1, k.z, Mo, k.cout, st,
                      training ? k.zbits : nullptr));
        } else {
            RC(bn_act(dt, k.y2, k.b2.scale, k.b2.shift, k.xin, nullptr, nullptr, 1, k.z, Mo, k.cout, st,
                      training ? k.zbits : nullptr));
        }
    }
    const Block& last = e->blocks.back();
    if (feat_out) RC(avgpool_fwd(dt, last.z, feat_out, e->B, e->T, e->hf * e->wf, 512, st));
    if (fmap_nchw) RC(nhwc_to_nchw_f32(dt, last.z, fmap_nchw, e->n_img, e->hf, e->wf, 512, st));
    if (training) {
        e->serial++;
        e->have_train_fwd = true;
    }
    return GDL_OK;
}

// reductions + finalize of one BatchNorm's backward: gamma / beta gradients and coef.  `count`: elements per channel the
// BatchNorm normalised over (differs from M only for the stem's pooled form)
static int bn_backward_reduce(gdl_encoder* e, BN& n, const void* g, const void* y, int relu_mask, size_t M, double count,
                              float* const* grads, hipStream_t st) {
    const int blocks = bn_bwd_blocks(M, n.c);
    RC(bn_bwd_reduce(e->dtype, g, y, n.scale, n.shift, n.mean, n.rstd, relu_mask, e->bnb_partial, M, n.c, st));
    return bn_bwd_finalize(e->bnb_partial, blocks, n.c, count, grads[n.pidx], grads[n.pidx + 1], n.coef, st);
}
static int bn_backward(gdl_encoder* e, BN& n, const void* g, const void* y, int relu_mask, void* dy, size_t M,
                       float* const* grads, hipStream_t st) {
    RC(bn_backward_reduce(e, n, g, y, relu_mask, M, (double)M, grads, st));
    return bn_bwd_apply(e->dtype, g, y, n.scale, n.shift, n.mean, n.rstd, e->params[n.pidx], n.coef, relu_mask, dy, M, n.c,
                        st);
}

// phase 0: the whole backward.  phase 1: the upstream gradient and layer4 (blocks 7, 6) -- afterwards the last 15
// gradient tensors (8.4 M of the 11.2 M parameters) are final on `stream`, so a data-parallel caller can start
// their all-reduce while phase 2 (layer3 .. layer1 and the stem) runs.
static int encoder_backward_impl(gdl_encoder* e, const float* dfeat, const float* dfmap_nchw, float* const* grads,
                                 void* stream, int phase) {
    GDL_REQUIRE(e && grads, "encoder_backward: null");
    GDL_REQUIRE(phase >= 0 && phase <= 2, "encoder_backward: phase %d", phase);
    if (phase != 2)
        GDL_REQUIRE((dfeat != nullptr) != (dfmap_nchw != nullptr), "encoder_backward: pass exactly one of dfeat / dfmap_nchw");
    if (!e->have_train_fwd) {
        set_error("encoder_backward: no training-mode forward to differentiate");
        return GDL_ERR_STATE;
    }
    if (phase == 2 && e->bw_serial != (long)e->serial) {
        set_error("encoder_backward: phase 2 without phase 1 of the same forward");
        return GDL_ERR_STATE;
    }
    for (int i = 0; i < GDL_ENC_NPARAMS; ++i) GDL_REQUIRE(grads[i], "encoder_backward: grads[%d] null", i);
    hipStream_t st = (hipStream_t)stream;
    const int dt = e->dtype;
    const int nblk = (int)e->blocks.size();
    const int bi_first = phase == 2 ? nblk - 3 : nblk - 1;  // blocks run from bi_first down to bi_last
    const int bi_last = phase == 1 ? nblk - 2 : 0;
    void* dz;     // gradient w.r.t. the current block's output
    void* spare;  // receives the gradient w.r.t. the block's input
    if (phase != 2) {
        // upstream gradient on the layer4 map -> gA
        if (dfeat)
            RC(avgpool_bwd(dt, dfeat, e->gA, e->B, e->T, e->hf * e->wf, 512, st));
        else
            RC(nchw_f32_to_nhwc(dt, dfmap_nchw, e->gA, e->n_img, e->hf, e->wf, 512, st));
        dz = e->gA;
        spare = e->gE;
        e->side_pending[0] = e->side_pending[1] = false;
    } else {
        dz = e->bw_dz;
        spare = e->bw_spare;
    }
    // The gradient of a block's output reaches the block already multiplied by the output's ReLU mask when the data
    // gradient that produced it applied the saved sign bits in its epilogue (every block but the last one): the block
    // then skips one read of z and one write of the masked gradient (two of its four tensor passes).
    bool premasked = phase == 2 ? e->bw_premasked : false;
    // ... and with the bn2 / downsample-BatchNorm sums of this block in e->bwB / e->bwB2 (b2_rows partial rows) when the data
    // gradient that produced dz computed them in its epilogue
    int b2_rows = phase == 2 ? e->bw_b2_rows : 0;
    const bool fuse = bw_fuse_on();
    // weight gradients: forked onto the side stream (sw) once their dy exists on st
    hipStream_t sw = e->has_side ? e->side : st;
    auto fork = [&]() -> int {  // sw waits for everything enqueued on st so far
        if (!e->has_side) return GDL_OK;
        hipError_t he = hipEventRecord(e->ev_fork, st);
        if (he == hipSuccess) he = hipStreamWaitEvent(e->side, e->ev_fork, 0);
        return he == hipSuccess ? GDL_OK : check_hip(he, "encoder_backward: fork");
    };
    for (int bi = bi_first; bi >= bi_last; --bi) {
        Block& k = e->blocks[bi];
        const size_t Mo = (size_t)k.n * k.p * k.q;
        const int par = bi & 1;
        void *gB = e->gB[par], *gC = e->gC[par], *gD = e->gD[par];
        if (e->has_side && e->side_pending[par]) {  // the side stream still reads this parity's buffers (two blocks ago)
            hipError_t he = hipStreamWaitEvent(st, e->ev_side[par], 0);
            if (he != hipSuccess) return check_hip(he, "encoder_backward: buffer wait");
            e->side_pending[par] = false;
        }
        // fused: do2 = dz * (z > 0) in place (relu of backbone.py:66) + reductions of bn2 (and of the
        // downsample BatchNorm); then finalize + apply per BatchNorm
        void* do2 = dz;
        {
            const int blocks = bn_bwd_blocks(Mo, k.cout);
            const BnFinBwd f2{e->bnb_partial, blocks, k.cout, (double)Mo, grads[k.b2.pidx], grads[k.b2.pidx + 1], k.b2.coef};
            const BnFinBwd fd{e->bnb_partial2, blocks, k.cout, (double)Mo, k.has_ds ? grads[k.bd.pidx] : nullptr,
                              k.has_ds ? grads[k.bd.pidx + 1] : nullptr, k.has_ds ? k.bd.coef : nullptr};
            if (b2_rows > 0) {
                // the sums came with dz (conv1's data gradient of the block behind this one): only the finalize is left
                const BnFinBwd g2{e->bwB, b2_rows, k.cout, (double)Mo, grads[k.b2.pidx], grads[k.b2.pidx + 1], k.b2.coef};
                const BnFinBwd gd{e->bwB2, b2_rows, k.cout, (double)Mo, k.has_ds ? grads[k.bd.pidx] : nullptr,
                                  k.has_ds ? grads[k.bd.pidx + 1] : nullptr, k.has_ds ? k.bd.coef : nullptr};
                if ((GDL_SKIPPED(32) || GDL_SKIPPED(262144))) {
                } else if (k.has_ds)
                    RC(bn_bwd_finalize_pair(g2, gd, st));
                else
                    RC(bn_bwd_finalize(g2.partial, g2.blocks, k.cout, (double)Mo, g2.dgamma, g2.dbeta, g2.coef, st));
            } else
            RC(block_bwd_reduce(dt, dz, k.z, k.y2, k.has_ds ? k.yd : nullptr, k.b2.mean, k.b2.rstd, k.has_ds ? k.bd.mean : nullptr,
                                k.has_ds ? k.bd.rstd : nullptr, do2, e->bnb_partial, e->bnb_partial2, Mo, k.cout, st, premasked));
            if (b2_rows == 0) {
                if (k.has_ds)  // one finalize launch for bn2 and the downsample BatchNorm
                    RC(bn_bwd_finalize_pair(f2, fd, st));
                else
                    RC(bn_bwd_finalize(e->bnb_partial, blocks, k.cout, (double)Mo, grads[k.b2.pidx], grads[k.b2.pidx + 1],
                                       k.b2.coef, st));
            }
            if (GDL_SKIPPED(8)) {
            } else if (k.has_ds)  // gB = dy2 and gD = dyd from one pass over do2
                RC(bn_bwd_apply2(dt, do2, k.y2, k.b2.mean, k.b2.rstd, e->params[k.b2.pidx], k.b2.coef, gB, k.yd, k.bd.mean,
                                 k.bd.rstd, e->params[k.bd.pidx], k.bd.coef, gD, Mo, k.cout, st));
            else
                RC(bn_bwd_apply(dt, do2, k.y2, k.b2.scale, k.b2.shift, k.b2.mean, k.b2.rstd, e->params[k.b2.pidx], k.b2.coef,
                                0, gB, Mo, k.cout, st));  // gB = dy2
        }
        auto wgrad2 = [&]() -> int {  // weight gradients of conv2 (and of the downsample convolution): need dy2 (dyd)
            RC(fork());
            if (!GDL_SKIPPED(16384))
            RC(conv_wgrad(dt, gB, k.a1, grads[k.c2.pidx], k.c2.tab_fwd, k.n, k.p, k.q, k.cout, k.cout, 3, 3, 1, 1, k.cout,
                          e->wg_ws, e->wg_ws_bytes, sw));
            if (k.has_ds && !GDL_SKIPPED(8192))
                RC(conv_wgrad(dt, gD, k.xin, grads[k.cd.pidx], k.cd.tab_fwd, k.n, k.h, k.w, k.cin, k.cout, 1, 1, k.cd.stride, 0,
                              k.cin, e->wg_ws, e->wg_ws_bytes, sw));
            return GDL_OK;
        };
        // (forking the weight gradients AFTER the data gradient that consumes the same dy -- GDL_WGRAD_LATE -- let them run at their
        // stand-alone speed and made the step 0.7 % slower, rounds 2 and 4: tools/experiments/r5_pruned_knobs.diff.txt)
        RC(wgrad2());
        if (fuse) {
            // gC = da1 * (a1 > 0) (sign bits of a1) with bn1's two sums from the epilogue; then finalize + apply
            const BwdStats bwa{k.y1, k.b1.mean, k.b1.rstd, e->bwA, nullptr, nullptr, nullptr, nullptr};
            if (!(GDL_SKIPPED(131072) || (GDL_SKIPPED(4096) && k.cout == 512)))
            RC(conv_dgrad(dt, gB, k.c2.w_crsk, gC, nullptr, k.c2.tab_dgrad, k.n, k.p, k.q, k.cout, k.cout, 3, 3, 1, 1, st,
                          k.abits, &bwa, &e->sk));
            const int rows = conv_dgrad_tiles_m(dt, k.n, k.p, k.q, k.cout, k.cout, 3, 3, 1, 1);
            if (!(GDL_SKIPPED(32) || GDL_SKIPPED(262144)))
                RC(bn_bwd_finalize(e->bwA, rows, k.cout, (double)Mo, grads[k.b1.pidx], grads[k.b1.pidx + 1], k.b1.coef, st));
            if (!GDL_SKIPPED(4))
            RC(bn_bwd_apply(dt, gC, k.y1, k.b1.scale, k.b1.shift, k.b1.mean, k.b1.rstd, e->params[k.b1.pidx], k.b1.coef, 0, gC, Mo,
                            k.cout, st));  // gC = dy1 (in place; the gradient is masked already)
        } else {
        RC(conv_dgrad(dt, gB, k.c2.w_crsk, gC, nullptr, k.c2.tab_dgrad, k.n, k.p, k.q, k.cout, k.cout, 3, 3, 1, 1,
                      st, nullptr, nullptr, &e->sk));  // gC = da1
        // relu + bn1 / conv1
        RC(bn_backward(e, k.b1, gC, k.y1, 1, gC, Mo, grads, st));  // gC = dy1 (in place)
        }
        auto wgrad1 = [&]() -> int {  // weight gradient of conv1: needs dy1
            RC(fork());
            if (!(k.c1.stride == 1 ? GDL_SKIPPED(16384) : GDL_SKIPPED(8192)))
            RC(conv_wgrad(dt, gC, k.xin, grads[k.c1.pidx], k.c1.tab_fwd, k.n, k.h, k.w, k.cin, k.cout, 3, 3, k.c1.stride, 1,
                          k.cin, e->wg_ws, e->wg_ws_bytes, sw));
            if (e->has_side) {
                hipError_t he = hipEventRecord(e->ev_side[par], e->side);
                if (he != hipSuccess) return check_hip(he, "encoder_backward: side event");
                e->side_pending[par] = true;
            }
            return GDL_OK;
        };
        RC(wgrad1());
        void* dxin;
        // the block's input is the previous block's output z (for block 0: the pooled stem output, whose mask the
        // stem's own backward applies): its sign bits turn dx into the masked gradient the previous block wants
        const uint8_t* inbits = (bi > 0 && premask_on()) ? e->blocks[bi - 1].zbits : nullptr;
        // the gradient this launch stores is (masked by inbits) the gradient of the previous block's output: its epilogue also
        // leaves the sums of that block's bn2 (and downsample BatchNorm) backward, so the block needs no pass of its own for them
        BwdStats bwb{};
        int next_rows = 0;
        if (fuse && inbits) {
            Block& pv = e->blocks[bi - 1];
            bwb = BwdStats{pv.y2, pv.b2.mean, pv.b2.rstd, e->bwB, pv.has_ds ? pv.yd : nullptr,
                           pv.has_ds ? pv.bd.mean : nullptr, pv.has_ds ? pv.bd.rstd : nullptr, pv.has_ds ? e->bwB2 : nullptr};
            next_rows = conv_dgrad_tiles_m(dt, k.n, k.h, k.w, k.cin, k.cout, 3, 3, k.c1.stride, 1);
        }
        const BwdStats* bwp = next_rows ? &bwb : nullptr;
        if (k.has_ds && ds_fold_on() && k.c1.stride == 2) {
            // the shortcut's 1x1 stride-2 data gradient rides in the 3x3 one as a tenth tap of the (even, even) pixels:
            // one launch and one tensor write instead of two launches, two writes and a read
            if (!GDL_SKIPPED(1024))
            RC(conv_dgrad_ds(dt, gC, k.c1.w_crsk, gD, k.cd.w_crsk, spare, k.c1.tab_dgrad, k.n, k.h, k.w, k.cin, k.cout, st,
                             inbits, bwp));
            dxin = spare;
            spare = dz;
        } else if (k.has_ds) {
            RC(conv_dgrad(dt, gD, k.cd.w_crsk, spare, nullptr, k.cd.tab_dgrad, k.n, k.h, k.w, k.cin, k.cout, 1, 1,
                          k.cd.stride, 0, st));
            RC(conv_dgrad(dt, gC, k.c1.w_crsk, spare, spare, k.c1.tab_dgrad, k.n, k.h, k.w, k.cin, k.cout, 3, 3,
                          k.c1.stride, 1, st, inbits, bwp));
            dxin = spare;
            spare = dz;  // the old dz buffer is free now
        } else {
            // identity shortcut: dx = dgrad(conv1) + do2, accumulated in place over do2
            if (!(GDL_SKIPPED(131072) || (GDL_SKIPPED(4096) && k.cout == 512)))
            RC(conv_dgrad(dt, gC, k.c1.w_crsk, do2, do2, k.c1.tab_dgrad, k.n, k.h, k.w, k.cin, k.cout, 3, 3, 1, 1, st,
                          inbits, bwp, &e->sk));
            dxin = do2;
        }
        premasked = inbits != nullptr;
        b2_rows = next_rows;
        dz = dxin;
    }
    if (phase == 1) {  // hand over to phase 2; the layer4 gradients must be complete on st
        e->bw_dz = dz;
        e->bw_spare = spare;
        e->bw_premasked = premasked;
        e->bw_b2_rows = b2_rows;
        e->bw_serial = (long)e->serial;
        if (e->has_side) {
            hipError_t he = hipEventRecord(e->ev_join, e->side);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, e->ev_join, 0);
            if (he != hipSuccess) return check_hip(he, "encoder_backward: phase join");
            e->side_pending[0] = e->side_pending[1] = false;
        }
        return GDL_OK;
    }
    e->bw_serial = -1;
    // stem: maxpool -> relu -> bn1 -> conv1 weight gradient (the input needs no gradient).  The gathered gradient
    // g0 = maxpool_bwd(dz) is never stored: bn1's two reductions run over the POOLED pair (dz, ymax) --
    // sum_pos g0'*xhat(y0[pos]) = sum_windows dz'*xhat(ymax) -- and the gather is fused into the apply pass.
    {
        BN& n = e->bn0;
        const size_t Mp = (size_t)e->n_img * e->h1 * e->w1;
        RC(bn_backward_reduce(e, n, dz, e->ymax, 1, Mp, (double)e->m0, grads, st));
        if (stem_bwd_fused_ok(dt, e->W)) {
            // round 5: gather + mask + BatchNorm-backward apply + weight gradient in one launch; the gradient of the stem output
            // (e->g0: as large as the stem output, the largest activation) is never written (conv_wgrad.hip stem_bwd_fused_kernel)
            RC(fork());
            if (!GDL_SKIPPED(64) && !GDL_SKIPPED(1048576))
            RC(stem_bwd_fused(dz, e->idx, e->y0, n.scale, n.shift, n.mean, n.rstd, e->params[n.pidx], n.coef, e->col, grads[0],
                              e->n_img, e->H, e->W, e->cin, e->wg_ws, e->wg_ws_bytes, sw));
        } else {
            // two launches (f32, or output rows under 64 pixels): the gathered gradient goes through e->g0
            if (!GDL_SKIPPED(64))
            RC(maxpool_bn_bwd_apply(dt, dz, e->idx, e->y0, n.scale, n.shift, n.mean, n.rstd, e->params[n.pidx], n.coef, e->g0,
                                    e->n_img, e->h0, e->w0, 64, st));
            RC(fork());
            if (!GDL_SKIPPED(1048576))
            RC(conv_stem_wgrad(dt, e->g0, e->col, grads[0], e->tab_stem, e->n_img, e->H, e->W, e->cin, e->wg_ws, e->wg_ws_bytes, sw));
        }
    }
    if (e->has_side) {  // join: everything the caller enqueues on st after this call sees all 60 gradients
        hipError_t he = hipEventRecord(e->ev_join, e->side);
        if (he == hipSuccess) he = hipStreamWaitEvent(st, e->ev_join, 0);
        if (he != hipSuccess) return check_hip(he, "encoder_backward: join");
        e->side_pending[0] = e->side_pending[1] = false;
    }
    return GDL_OK;
}

int gdl_encoder_backward(gdl_encoder_t* e, const float* dfeat, const float* dfmap_nchw, float* const* grads,
                         void* stream) {
    return encoder_backward_impl(e, dfeat, dfmap_nchw, grads, stream, 0);
}
int gdl_encoder_backward_phase(gdl_encoder_t* e, int phase, const float* dfeat, const float* dfmap_nchw,
                               float* const* grads, void* stream) {
    GDL_REQUIRE(phase == 1 || phase == 2, "encoder_backward_phase: phase must be 1 or 2");
    return encoder_backward_impl(e, dfeat, dfmap_nchw, grads, stream, phase);
}

}  // extern "C"
