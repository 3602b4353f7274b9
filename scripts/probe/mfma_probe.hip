// How fast can v_mfma_f32_32x32x16_bf16 issue on gfx950 as a function of waves per SIMD and of the number of independent
// accumulators per wave?  No memory traffic.  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    floatx16 acc[ACC];
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    unsigned seed = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
        if (out[1] == 0.f) { x[j] = (__bf16)(float)(threadIdx.x & 3); y[j] = (__bf16)1.0f; }   // out[1] selects constant / random operands
        else {
            seed = seed * 1664525u + 1013904223u; x[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
            seed = seed * 1664525u + 1013904223u; y[j] = (__bf16)(((int)(seed >> 9) & 0xffff) / 32768.f - 1.f);
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 12 / ACC; ++rep)
#pragma unroll
            for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) out[0] = s;
}

template <int ACC>
void run(int threads, int blocks_per_cu, float* d) {
    const int iters = 40000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<ACC>, dim3(grid), dim3(threads), 0, 0, d, iters);
    hipEventRecord(a, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<ACC>, dim3(grid), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double waves = (double)grid * threads / 64;
    const double flops = waves * iters * 12 * 32.0 * 32 * 16 * 2;
    printf("acc %2d  threads %4d  blocks/CU %d  waves/SIMD %4.1f : %7.1f us  %7.1f TFLOP/s (%.0f %% of 2500)\n", ACC, threads, blocks_per_cu,
           (double)threads * blocks_per_cu / 256, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 25);
}

int main() {
    float* d; hipMalloc(&d, 8);
    for (int rnd = 0; rnd < 2; ++rnd) {
        float h[2] = {0.f, (float)rnd};
        hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
        printf("operands: %s\n", rnd ? "random bf16 in [-1, 1), different per lane" : "constant small integers");
        for (int rep = 0; rep < 3; ++rep) {   // long enough for the power management to settle
            run<2>(512, 2, d);
            run<4>(512, 1, d);
        }
        system("rocm-smi --showclocks --showpower | grep -E 'sclk|Power'");
    }
    return 0;
}
