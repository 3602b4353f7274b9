// Optional in-library timing of the instrumented kernels (bench.py's `roofline` object):
// hipEvent pairs recorded on the SAME stream as the kernel, one pair per launch, summed per kernel class.
#include "common.h"

#include <map>
#include <string>
#include <vector>

namespace bd {

struct ProfRec { int cls; hipEvent_t a, b; };
struct ProfCls { std::string name; int64_t launches = 0; double flops = 0, bytes = 0, ms = 0; };

static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<ProfCls> g_cls;
static std::map<std::string, int> g_idx;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;

static hipEvent_t get_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}

bool prof_on() { return g_on; }

int prof_begin(const char* name, double flops, double bytes, hipStream_t st) {
    auto it = g_idx.find(name);
    int c;
    if (it == g_idx.end()) {
        c = (int)g_cls.size();
        g_idx[name] = c;
        ProfCls pc; pc.name = name;
        g_cls.push_back(pc);
    } else c = it->second;
    g_cls[c].launches++; g_cls[c].flops += flops; g_cls[c].bytes += bytes;
    ProfRec r; r.cls = c; r.a = get_event(); r.b = get_event();
    if (!r.a || !r.b) return -1;
    hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_end(int rec, hipStream_t st) {
    if (rec >= 0) hipEventRecord(g_recs[rec].b, st);
}
static void drain() {
    for (auto& r : g_recs) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) g_cls[r.cls].ms += ms;
    }
    g_recs.clear();
    g_pool_next = 0;
}

}  // namespace bd

using namespace bd;

extern "C" int bd_prof_enable(int on) { g_on = on != 0; return BD_OK; }
extern "C" int bd_prof_enabled(void) { return g_on ? 1 : 0; }
extern "C" int bd_prof_reset(void) {
    drain();
    g_cls.clear(); g_idx.clear();
    return BD_OK;
}
extern "C" int bd_prof_num_classes(void) { drain(); return (int)g_cls.size(); }
extern "C" int bd_prof_get(int cls, const char** name, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    drain();
    BD_CHECK(cls >= 0 && cls < (int)g_cls.size(), BD_ERR_INVALID, "bd_prof_get: class %d out of range", cls);
    const ProfCls& c = g_cls[cls];
    if (name) *name = c.name.c_str();
    if (launches) *launches = c.launches;
    if (total_ms) *total_ms = c.ms;
    if (flops) *flops = c.flops;
    if (bytes) *bytes = c.bytes;
    return BD_OK;
}
