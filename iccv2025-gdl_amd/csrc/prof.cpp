// prof.cpp -- implementation of the timing tap declared in prof.h and its C ABI.
#include "prof.h"

#include <mutex>
#include <string>
#include <vector>

namespace gdl {

struct Slot {
    std::string name;
    int bound;
};
struct Rec {
    int slot;
    hipEvent_t e0, e1;
    double work;
};
static std::mutex g_mu;
static bool g_on = false;
static std::string g_filter;  // when non-empty only launches of this kernel name are recorded
static std::vector<Slot> g_slots;
static std::vector<Rec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;

bool prof_enabled() { return g_on; }

void prof_events(int token, hipEvent_t* e0, hipEvent_t* e1) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < 0 || token >= (int)g_recs.size()) return;
    *e0 = g_recs[token].e0;
    *e1 = g_recs[token].e1;
}

int prof_begin(const char* name, int bound, hipStream_t st, bool ext) {
    if (!g_on) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_filter.empty() && g_filter != name) return -1;
    int slot = -1;
    for (size_t i = 0; i < g_slots.size(); ++i)
        if (g_slots[i].name == name) {
            slot = (int)i;
            break;
        }
    if (slot < 0) {
        g_slots.push_back(Slot{name, bound});
        slot = (int)g_slots.size() - 1;
    }
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
    if (!ext) (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void prof_end(int token, hipStream_t st, double work, bool ext) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < 0 || token >= (int)g_recs.size()) return;
    if (!ext) (void)hipEventRecord(g_recs[token].e1, st);
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

int gdl_prof_enabled(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_on ? 1 : 0;
}

int gdl_prof_set_filter(const char* name) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_filter = name ? name : "";
    return GDL_OK;
}

int gdl_prof_nslots(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_slots.size();
}

const char* gdl_prof_slot_name(int slot) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (slot >= 0 && slot < (int)g_slots.size()) ? g_slots[slot].name.c_str() : "";
}

int gdl_prof_slot_bound(int slot) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (slot >= 0 && slot < (int)g_slots.size()) ? g_slots[slot].bound : -1;
}

// Synchronises the device, folds all recorded launches into per-slot totals and clears the log.
// launches[s], ms[s], work[s] for s < gdl_prof_nslots() (call gdl_prof_nslots() AFTER the timed region).
int gdl_prof_collect(int64_t* launches, double* ms, double* work) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return check_hip(e, "prof_collect: hipDeviceSynchronize");
    std::lock_guard<std::mutex> lk(g_mu);
    const int ns = (int)g_slots.size();
    for (int s = 0; s < ns; ++s) {
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
