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
    double bytes;  // MFMA-bound launches: algorithmic HBM bytes (0 = not stated)
    hipStream_t st;  // the launching stream (gdl_prof_timeline: which lane the launch ran on)
};
static std::mutex g_mu;
static bool g_on = false;
static std::string g_filter;  // when non-empty only launches of this kernel name are recorded
static std::vector<Slot> g_slots;
static std::vector<Rec> g_recs;
static std::vector<double> g_floor_ms, g_bytes;  // per slot, filled by the last gdl_prof_collect (gdl_prof_collect_floor)
static double g_peak_flops = 2.5e15, g_peak_bytes = 8.0e12;
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
    r.bytes = 0;
    r.st = st;
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

void prof_end(int token, hipStream_t st, double work, bool ext, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < 0 || token >= (int)g_recs.size()) return;
    if (!ext) (void)hipEventRecord(g_recs[token].e1, st);
    g_recs[token].work = work;
    g_recs[token].bytes = bytes;
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
    g_floor_ms.assign(ns, 0.0);
    g_bytes.assign(ns, 0.0);
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
            // per-launch floor of the combined roofline: the slower of its arithmetic and its bytes at the peaks
            const bool mfma = g_slots[r.slot].bound == PROF_MFMA;
            const double tf = mfma ? r.work / g_peak_flops : 0.0;
            const double by = mfma ? r.bytes : r.work;
            const double tb = by / g_peak_bytes;
            g_floor_ms[r.slot] += (tf > tb ? tf : tb) * 1e3;
            g_bytes[r.slot] += by;
        }
        g_pool.push_back({r.e0, r.e1});
    }
    g_recs.clear();
    return GDL_OK;
}

// Per-launch timeline of the records gathered so far (call it BEFORE gdl_prof_collect, which clears them): for record i, in
// enqueue order, slot[i], lane[i] (index of its stream in order of first appearance), start_ms[i] / end_ms[i] relative to the
// earliest start among the records, work[i] (flops or bytes, as the slot's bound says) and bytes[i] (algorithmic HBM bytes of an
// MFMA-bound launch, 0 = not stated).  Synchronises the device.  Returns the number of records (<= cap are written), < 0: error.
int gdl_prof_timeline(int cap, int32_t* slot, int32_t* lane, double* start_ms, double* end_ms, double* work, double* bytes) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return -check_hip(e, "prof_timeline: hipDeviceSynchronize");
    std::lock_guard<std::mutex> lk(g_mu);
    const int n = (int)g_recs.size();
    if (n == 0) return 0;
    std::vector<hipStream_t> lanes;
    std::vector<double> s0((size_t)n, 0.0), s1((size_t)n, 0.0);
    double first = 0.0;
    for (int i = 0; i < n; ++i) {
        float a = 0.f, d = 0.f;
        (void)hipEventElapsedTime(&a, g_recs[0].e0, g_recs[i].e0);  // (negative when record i started before record 0)
        (void)hipEventElapsedTime(&d, g_recs[i].e0, g_recs[i].e1);
        s0[i] = (double)a;
        s1[i] = (double)a + (double)d;
        if (s0[i] < first) first = s0[i];
    }
    for (int i = 0; i < n && i < cap; ++i) {
        size_t l = 0;
        while (l < lanes.size() && lanes[l] != g_recs[i].st) ++l;
        if (l == lanes.size()) lanes.push_back(g_recs[i].st);
        if (slot) slot[i] = g_recs[i].slot;
        if (lane) lane[i] = (int32_t)l;
        if (start_ms) start_ms[i] = s0[i] - first;
        if (end_ms) end_ms[i] = s1[i] - first;
        if (work) work[i] = g_recs[i].work;
        if (bytes) bytes[i] = g_recs[i].bytes;
    }
    return n;
}

// Combined roofline of the launches folded by the LAST gdl_prof_collect: floor_ms[s] = sum over the slot's launches of
// max(flop / peak_flops, bytes / peak_bytes) (the peaks given to gdl_prof_set_peaks; defaults 2.5e15 / 8e12), bytes[s] = their
// algorithmic HBM bytes (0 where a launcher states none: the floor is then the arithmetic's).  No device work.
int gdl_prof_set_peaks(double peak_flops, double peak_bytes) {
    GDL_REQUIRE(peak_flops > 0 && peak_bytes > 0, "prof_set_peaks: peaks must be positive");
    std::lock_guard<std::mutex> lk(g_mu);
    g_peak_flops = peak_flops;
    g_peak_bytes = peak_bytes;
    return GDL_OK;
}

int gdl_prof_collect_floor(double* floor_ms, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t s = 0; s < g_floor_ms.size(); ++s) {
        if (floor_ms) floor_ms[s] = g_floor_ms[s];
        if (bytes) bytes[s] = g_bytes[s];
    }
    return GDL_OK;
}

}  // extern "C"
