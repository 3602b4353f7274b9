// VERDICT round 4, task 5(c): do a STREAMING role (GroupNorm-backward-like: read two tensors, write one) and an MFMA role (register-operand
// v_mfma_f32_32x32x16_bf16, random bf16 operands) finish sooner CO-RESIDENT on every CU -- one launch, even workgroups matrix, odd workgroups
// streaming -- than back to back, on a board that is power-limited under the matrix role alone?  Also: the same two roles as two launches on two
// streams (what the product's two-stream backward does today: the hardware partitions CUs between them).
//   hipcc --offload-arch=gfx950 -O3 two_role_probe.hip -o two_role_probe && ./two_role_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_role(float* out, int iters, unsigned id) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    unsigned seed = id * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
        seed = seed * 1664525u + 1013904223u; x[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
        seed = seed * 1664525u + 1013904223u; y[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
    }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) out[0] = s;
}
// z = a * b + a  over n float4 per role-workgroup set (grid-stride)
__device__ __forceinline__ void stream_role(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ z, long long n, int wg, int nwg) {
    for (long long i = (long long)wg * blockDim.x + threadIdx.x; i < n; i += (long long)nwg * blockDim.x) {
        const float4 u = a[i], v = b[i];
        z[i] = make_float4(u.x * v.x + u.x, u.y * v.y + u.y, u.z * v.z + u.z, u.w * v.w + u.w);
    }
}
// mode 0: matrix only, 1: streaming only, 2: both (even / odd workgroups)
__global__ __launch_bounds__(256) void two_role(int mode, float* out, int iters, const float4* a, const float4* b, float4* z, long long n) {
    if (mode == 0) mfma_role(out, iters, blockIdx.x * 256u + threadIdx.x);
    else if (mode == 1) stream_role(a, b, z, n, blockIdx.x, gridDim.x);
    else if (blockIdx.x & 1) stream_role(a, b, z, n, blockIdx.x >> 1, gridDim.x >> 1);
    else mfma_role(out, iters, (blockIdx.x >> 1) * 256u + threadIdx.x);
}

static float timeit(int reps, void (*fn)(void*), void* ctx) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn(ctx); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < reps; ++k) fn(ctx);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}
struct Ctx { float* out; int iters; float4 *a, *b, *z; long long n; int wgs; hipStream_t s1, s2; hipEvent_t ej; };
static void run_m(void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, 0, 0, c->out, c->iters, c->a, c->b, c->z, c->n); }
static void run_s(void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, 0, 1, c->out, c->iters, c->a, c->b, c->z, c->n); }
static void run_seq(void* p) { run_m(p); run_s(p); }
static void run_co(void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(two_role, dim3(2 * c->wgs), dim3(256), 0, 0, 2, c->out, c->iters, c->a, c->b, c->z, c->n); }
static void run_2s(void* p) {
    Ctx* c = (Ctx*)p;
    hipEventRecord(c->ej, 0); hipStreamWaitEvent(c->s1, c->ej, 0); hipStreamWaitEvent(c->s2, c->ej, 0);
    hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, c->s1, 0, c->out, c->iters, c->a, c->b, c->z, c->n);
    hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, c->s2, 1, c->out, c->iters, c->a, c->b, c->z, c->n);
    hipEventRecord(c->ej, c->s1); hipStreamWaitEvent(0, c->ej, 0);
    hipEventRecord(c->ej, c->s2); hipStreamWaitEvent(0, c->ej, 0);
}

// two streams, no per-pair fork / join: the caller joins once per timing loop (the product's deferred join)
static void run_2s_free(void* p) {
    Ctx* c = (Ctx*)p;
    hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, c->s1, 0, c->out, c->iters, c->a, c->b, c->z, c->n);
    hipLaunchKernelGGL(two_role, dim3(c->wgs), dim3(256), 0, c->s2, 1, c->out, c->iters, c->a, c->b, c->z, c->n);
}
static float timeit_free(int reps, Ctx* c) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run_2s_free(c); hipDeviceSynchronize();
    hipEventRecord(e0, 0); hipStreamWaitEvent(c->s1, e0, 0); hipStreamWaitEvent(c->s2, e0, 0);
    for (int k = 0; k < reps; ++k) run_2s_free(c);
    hipEventRecord(c->ej, c->s1); hipStreamWaitEvent(0, c->ej, 0);
    hipEventRecord(e1, c->s2); hipStreamWaitEvent(0, e1, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    Ctx c;
    hipMalloc(&c.out, 8); hipMemset(c.out, 0, 8);
    c.n = (64ll << 20) / 16 * 4;                       // 256 MB per tensor: 768 MB of traffic per streaming pass (past the Infinity Cache)
    hipMalloc(&c.a, c.n * 16); hipMalloc(&c.b, c.n * 16); hipMalloc(&c.z, c.n * 16);
    hipMemset(c.a, 0, c.n * 16); hipMemset(c.b, 0, c.n * 16);
    // argv[1]: priority of the streaming role's stream relative to the matrix role's (0 = both default; -1 = higher; 1 = lower)
    int plo = 0, phi = 0; hipDeviceGetStreamPriorityRange(&plo, &phi);
    const int rel = argc > 1 ? atoi(argv[1]) : 0;
    hipStreamCreateWithPriority(&c.s1, hipStreamDefault, rel < 0 ? plo : (rel > 0 ? phi : 0));
    hipStreamCreateWithPriority(&c.s2, hipStreamDefault, rel < 0 ? phi : (rel > 0 ? plo : 0));
    printf("stream priorities: matrix %d, streaming %d (range %d .. %d, lower number = higher priority)\n", rel < 0 ? plo : (rel > 0 ? phi : 0), rel < 0 ? phi : (rel > 0 ? plo : 0), plo, phi);
    hipEventCreateWithFlags(&c.ej, hipEventDisableTiming);
    c.wgs = 256 * 4;                                   // four 256-thread workgroups per CU per role (1 wave per SIMD each)
    // size the matrix role so that it takes about as long as one streaming pass
    c.iters = 2000;
    float ts = timeit(10, run_s, &c), tm = timeit(10, run_m, &c);
    c.iters = (int)(c.iters * ts / tm);
    const int it1 = c.iters; const long long n1 = c.n;
    for (int rep = 0; rep < 4; ++rep) {
        // rep 3: a quarter of the work per launch (the size of the product's 32 x 32 layer kernels): is the two-stream loss a fixed cost per fork / join?
        if (rep == 3) { c.iters = it1 / 4; c.n = n1 / 4; }
        tm = timeit(20, run_m, &c); ts = timeit(20, run_s, &c);
        const float tseq = timeit(20, run_seq, &c), tco = timeit(20, run_co, &c), t2s = timeit(20, run_2s, &c), t2f = timeit_free(20, &c);
        const double tf = (double)c.wgs * 4 * c.iters * 12 * 32768.0 / tm / 1e6, gbs = 3.0 * c.n * 16 / ts / 1e3;
        printf("matrix alone %7.1f us (%6.0f TFLOP/s)   streaming alone %7.1f us (%5.0f GB/s)   back to back %7.1f   co-resident (one launch) %7.1f = %.3f x   "
               "two streams, fork + join per pair %7.1f = %.3f x   two streams, one join per 20 pairs %7.1f = %.3f x\n", tm, tf, ts, gbs, tseq, tco, tco / tseq,
               t2s, t2s / tseq, t2f, t2f / tseq);
    }
    system("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | head -4");
    return 0;
}
