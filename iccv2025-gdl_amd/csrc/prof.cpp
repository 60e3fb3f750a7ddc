// prof.cpp -- implementation of the timing tap declared in prof.h and its C ABI.
#include "prof.h"

#include <mutex>
#include <vector>

namespace gdl {

static const char* kSlotNames[PROF_NSLOTS] = {
    "conv_igemm_kernel<bf16|f32,256,64,4,1,FWD>", "conv_igemm_kernel<128,128,2,2,FWD>", "conv_igemm_kernel<64,64,2,2,FWD>",
    "conv_igemm_kernel<256,64,4,1,DGRAD>", "conv_igemm_kernel<128,128,2,2,DGRAD>", "conv_igemm_kernel<64,64,2,2,DGRAD>",
    "conv_wgrad_kernel", "wgrad_reduce_kernel", "bn_act_kernel", "bn_bwd_reduce_kernel", "bn_bwd_apply_kernel",
    "relu_bwd_kernel", "bn_relu_maxpool_kernel", "maxpool_bwd_kernel", "stem_im2col_kernel", "pack_weight_kernel",
    "sgd_kernel", "grad_stats_kernel"};
static const int kSlotBound[PROF_NSLOTS] = {1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 1 = mfma, 0 = hbm

struct Rec {
    int slot;
    hipEvent_t e0, e1;
    double work;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;

bool prof_enabled() { return g_on; }

int prof_begin(int slot, hipStream_t st) {
    if (!g_on) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r;
    r.slot = slot;
    r.work = 0;
    if (!g_pool.empty()) {
        r.e0 = g_pool.back().first;
        r.e1 = g_pool.back().second;
        g_pool.pop_back();
    } else {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    }
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void prof_end(int token, hipStream_t st, double work) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < 0 || token >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[token].e1, st);
    g_recs[token].work = work;
}

}  // namespace gdl

using namespace gdl;

extern "C" {

int gdl_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return GDL_OK;
}

int gdl_prof_nslots(void) { return PROF_NSLOTS; }

const char* gdl_prof_slot_name(int slot) { return (slot >= 0 && slot < PROF_NSLOTS) ? kSlotNames[slot] : ""; }

int gdl_prof_slot_bound(int slot) { return (slot >= 0 && slot < PROF_NSLOTS) ? kSlotBound[slot] : -1; }

// Synchronises the device, folds all recorded launches into per-slot totals and clears the log.
// launches[s], ms[s], work[s] for s < gdl_prof_nslots().
int gdl_prof_collect(int64_t* launches, double* ms, double* work) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return check_hip(e, "prof_collect: hipDeviceSynchronize");
    std::lock_guard<std::mutex> lk(g_mu);
    for (int s = 0; s < PROF_NSLOTS; ++s) {
        launches[s] = 0;
        ms[s] = 0.0;
        work[s] = 0.0;
    }
    for (Rec& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            launches[r.slot] += 1;
            ms[r.slot] += (double)t;
            work[r.slot] += r.work;
        }
        g_pool.push_back({r.e0, r.e1});
    }
    g_recs.clear();
    return GDL_OK;
}

}  // extern "C"
