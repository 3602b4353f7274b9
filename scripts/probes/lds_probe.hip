// LDS read-rate probe (gfx950): bytes per clock per CU for ds_read_b128, ds_read_b64 and ds_read_b64_tr_b16 issued from 1, 2 or 4 waves per
// SIMD, 16 reads per s_waitcnt, conflict-free addresses (the K-contiguous / K-major images of conv_ps.hip / gemm_sp.hip / attn_sp.hip).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
template <int OFF> __device__ __forceinline__ short4_t rd_tr(unsigned a) { short4_t v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }
template <int OFF> __device__ __forceinline__ short4_t rd_64(unsigned a) { short4_t v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }
template <int OFF> __device__ __forceinline__ short8_t rd_128(unsigned a) { short8_t v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }

template <int KIND>   // 0: b128 on a K-contiguous image, 1: tr_b16 on a K-major image, 2: plain b64 at the tr addresses
__global__ void probe(int iters, unsigned long long* cycles, int* sink) {
    __shared__ __attribute__((aligned(128))) char smem[65536];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    unsigned a;
    if (KIND == 0) a = base + li * 128 + (((h) ^ ((li >> 1) & 7)) << 4);
    else {
        const int sl = lane & 15, hb = (lane >> 4) & 1, kq = sl >> 2, rq = sl & 3;
        a = base + (8 * h + kq) * 512 + ((0 ^ kq) << 6) + hb * 32 + rq * 8;
    }
    int acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            short8_t v[16];
#define R(i) v[i] = rd_128<(i) * 4096 % 49152>(a);
            R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(v[i])); acc += v[i][0]; }
        } else {
            short4_t v[16];
#define R(i) v[i] = (KIND == 1) ? rd_tr<((i) & 7) * 2048 + ((i) >> 3) * 256>(a) : rd_64<((i) & 7) * 2048 + ((i) >> 3) * 256>(a);
            R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(v[i])); acc += v[i][0]; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    unsigned long long* d_c; int* d_s;
    hipMalloc(&d_c, 4096 * 8); hipMalloc(&d_s, 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, iters = 2000;
    const char* names[3] = {"ds_read_b128 (K-contiguous image)", "ds_read_b64_tr_b16 (K-major image)", "ds_read_b64 (same addresses)"};
    const int bytes[3] = {1024, 512, 512};
    for (int kind = 0; kind < 3; ++kind)
        for (int waves = 4; waves <= 16; waves *= 2) {   // waves per CU = one workgroup of `waves` waves per CU
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(waves * 64), 0, 0, iters, d_c, d_s);
                if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(waves * 64), 0, 0, iters, d_c, d_s);
                if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(waves * 64), 0, 0, iters, d_c, d_s);
                hipDeviceSynchronize();
            }
            static unsigned long long h[4096];
            hipMemcpy(h, d_c, cus * 8, hipMemcpyDeviceToHost);
            double c = 0; for (int i = 0; i < cus; ++i) c += (double)h[i]; c /= cus;
            const double total = (double)iters * 16 * bytes[kind] * waves;
            printf("%-40s %2d waves/CU: %8.0f s_memtime ticks per workgroup, %7.1f bytes per tick per CU\n", names[kind], waves, c, total / c);
        }
    // s_memtime tick rate vs wall: time one probe with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(256), 0, 0, iters * 4, d_c, d_s); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h2[4096]; hipMemcpy(h2, d_c, cus * 8, hipMemcpyDeviceToHost);
    printf("tick rate: %.0f ticks in %.3f ms -> %.1f MHz\n", (double)h2[0], ms, (double)h2[0] / ms / 1e3);
    return 0;
}
