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
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_end(int rec, hipStream_t st) {
    if (rec >= 0) (void)hipEventRecord(g_recs[rec].b, st);
}
static void drain() {
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) g_cls[r.cls].ms += ms;
    }
    g_recs.clear();
    g_pool_next = 0;
}

// register-operand MFMA loop (no memory traffic): the issue rate the board sustains for the matrix pipe alone
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void mfma_probe_kernel(float* sink, int iters, int random_operands) {
    probe_f32x16 acc[2];
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    probe_bf16x8 x, y;
    unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
        if (!random_operands) { x[j] = (__bf16)(float)(threadIdx.x & 3); y[j] = (__bf16)1.0f; }
        else {
            seed = seed * 1664525u + 1013904223u; x[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
            seed = seed * 1664525u + 1013904223u; y[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) sink[0] = s;     // never true for these operands: keeps the loop alive
}

}  // namespace bd

using namespace bd;

extern "C" int bd_mfma_probe(int random_operands, int iters, int launches, double* tflops, bd_stream_t stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BD_CHECK(tflops && iters > 0 && launches > 0, BD_ERR_INVALID, "bd_mfma_probe: tflops must be non-null, iters and launches positive");
    int dev = 0, cus = 256;
    BD_HIP_TRY(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* sink = nullptr;
    BD_HIP_TRY(hipMalloc(&sink, 64));
    hipEvent_t a, b;
    BD_HIP_TRY(hipEventCreate(&a)); BD_HIP_TRY(hipEventCreate(&b));
    const dim3 grid(2 * cus), block(512);
    hipLaunchKernelGGL(mfma_probe_kernel, grid, block, 0, st, sink, iters, random_operands);     // warm-up: clocks settle
    BD_HIP_TRY(hipEventRecord(a, st));
    for (int k = 0; k < launches; ++k) hipLaunchKernelGGL(mfma_probe_kernel, grid, block, 0, st, sink, iters, random_operands);
    BD_HIP_TRY(hipEventRecord(b, st));
    BD_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    BD_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(sink);
    const double waves = (double)grid.x * block.x / 64;
    *tflops = waves * iters * 12.0 * (32.0 * 32 * 16 * 2) * launches / ((double)ms * 1e-3) / 1e12;
    return BD_OK;
}

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
