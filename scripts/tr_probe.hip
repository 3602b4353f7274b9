#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, const int* addr_in) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int a = addr_in[threadIdx.x];
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256]; int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 4; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 4;                         // consecutive 8-byte chunks
            if (pat == 1) h_addr[l] = (l & 15) * 100 + (l >> 4) * 2000;  // each lane its own "row" of stride 100
            if (pat == 2) h_addr[l] = 0;
            if (pat == 3) h_addr[l] = (l & 3) * 4 + ((l >> 2) & 3) * 16 + (l >> 4) * 64;  // 4x16 block: lane -> (k-row = (l>>2)&3, col quad = l&3)
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) { printf("l%02d a=%4d: %4d %4d %4d %4d%s", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3], (l % 4 == 3) ? "\n" : " | "); }
    }
    return 0;
}
