// Shared host/device helpers for libbd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/bd_hip.h"

namespace bd {

void set_error(const char* fmt, ...);

#define BD_CHECK(cond, status, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            ::bd::set_error(__VA_ARGS__);           \
            return (status);                        \
        }                                           \
    } while (0)

#define BD_LAUNCH_CHECK(name)                                                         \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ::bd::set_error("%s: launch failed: %s", (name), hipGetErrorString(e__)); \
            return BD_ERR_LAUNCH;                                                     \
        }                                                                             \
    } while (0)

#define BD_HIP_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::bd::set_error("%s failed: %s", #expr, hipGetErrorString(e__));               \
            return BD_ERR_LAUNCH;                                                          \
        }                                                                                  \
    } while (0)

#define BD_TRY(expr)                  \
    do {                              \
        int s__ = (expr);             \
        if (s__ != BD_OK) return s__; \
    } while (0)

static inline hipStream_t S(bd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave64 reductions (DPP/shuffle, no LDS) --------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
// d/dz [z * sigmoid(z)] = s * (1 + z * (1 - s))
__device__ __forceinline__ float silu_grad_f(float z) {
    float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// ---- THE split of this library (round 6): x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi).  x - hi is exact in fp32 (at most 16
// significant bits), so hi + lo reproduces x to 2^-18 |x| and the dropped lo * lo product is <= 2^-18 of the term -- rounds 1 - 5 truncated
// hi (the upper 16 bits of x), which costs one bit on both (per-convolution error vs fp64 8.8e-6 -> 4e-6, scripts/wino/numerics.py).  Every
// producer of split planes and every on-the-fly split goes through these three functions: the MFMA engines stay bit-identical to one
// another because they split identically.  (__bf16)float is v_cvt_pk_bf16_f32 on gfx950: round to nearest even.
typedef __bf16 bd_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bd_pack_hi(float a, float b) {          // bf16 RNE(a) | bf16 RNE(b) << 16
    bd_bf16x2_t t;
    t[0] = (__bf16)a; t[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ unsigned bd_pack_lo(float a, float b) {          // the remainders' planes
    const unsigned h = bd_pack_hi(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16);
    const float rb = b - __builtin_bit_cast(float, h & 0xFFFF0000u);
    bd_bf16x2_t t;
    t[0] = (__bf16)ra; t[1] = (__bf16)rb;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ unsigned bd_split1(float v) {                    // hi | lo << 16 of one value
    const __bf16 h = (__bf16)v;
    const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h);
    const __bf16 l = (__bf16)(v - __builtin_bit_cast(float, hb << 16));
    return hb | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

// ---- run-time tuning knobs (bd_tune_set): a generation counter that makes every plan lay its workspace out again -----------------
extern int g_tune_gen;

// ---- internal (non-ABI) launchers shared between files -------------------------------------------
int igemm_launch(const bd_igemm_desc& d, hipStream_t stream);
size_t igemm_workspace_bytes(const bd_igemm_desc& d);
bool prof_on();
int prof_begin(const char* name, double flops, double bytes, hipStream_t st);
void prof_end(int rec, hipStream_t st);
int add_launch(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C, float scale, int acc,
               hipStream_t st);

}  // namespace bd
