// What does a cross-stream dependency (hipEventRecord + hipStreamWaitEvent) cost on this ROCm stack?  The backward plan forks every weight gradient onto
// a side stream and joins per node (~350 edges per CIFAR step).  N iterations of: main-stream kernel K (duration d), side-stream kernel K; variants:
//   none     : no dependencies at all (two independent queues)
//   fork     : side kernel i waits for main kernel i (record on main, wait on side); one join at the end
//   forkjoin : fork as above + main kernel i+1 waits for side kernel i
// usage: edge_probe [kernel_us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long cycles, float* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (cycles < 0) out[0] = 1.f;
}
int main(int argc, char** argv) {
    const double us = argc > 1 ? atof(argv[1]) : 20.0;
    const long long cyc = (long long)(us * 100.0);          // wall_clock64: 100 MHz
    float* d; hipMalloc(&d, 4);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreateWithPriority(&s1, hipStreamDefault, 0);
    const int N = 200;
    hipEvent_t ev[2 * N + 2];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int grid : {1, 256, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                hipEventRecord(t0, s0);
                hipEventRecord(ev[2 * N], s0); hipStreamWaitEvent(s1, ev[2 * N], 0);
                for (int i = 0; i < N; ++i) {
                    if (mode == 3) { hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s0, cyc, d); hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s0, cyc, d); continue; }
                    if (mode == 2 && i > 0) hipStreamWaitEvent(s0, ev[N + i - 1], 0);
                    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s0, cyc, d);
                    if (mode >= 1) { hipEventRecord(ev[i], s0); hipStreamWaitEvent(s1, ev[i], 0); }
                    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s1, cyc, d);
                    if (mode == 2) hipEventRecord(ev[N + i], s1);
                }
                hipEventRecord(ev[2 * N + 1], s1); hipStreamWaitEvent(s0, ev[2 * N + 1], 0);
                hipEventRecord(t1, s0); hipEventSynchronize(t1);
                float ms; hipEventElapsedTime(&ms, t0, t1);
                if (ms < best) best = ms;
            }
            const char* nm[4] = {"none (two independent queues)", "fork per kernel, one join", "fork + join per kernel", "single stream, 2N kernels"};
            printf("grid %4d  kernel %.0f us  N %d  %-32s: %8.1f us total = %6.2f us per iteration\n", grid, us, N, nm[mode], best * 1e3, best * 1e3 / N);
        }
    }
    return 0;
}
