// Weight gradients of the two "thin" 3x3 convolutions of the UNet -- conv_in (3 -> C, unet_2d.py:124) and conv_out
// (C -> 3, unet_2d.py:217) -- as direct kernels.  With 3 channels on one side the implicit GEMM has a 27-wide (or
// 3-tall) output: a 64x64 MFMA tile is > 90 % padding and its operand needs the generic per-element loaders.  These are
// plain fp32 FMA streaming reductions instead (both compute modes): every workgroup reduces a slab of pixels into a
// partial dW (+ db), a second pass sums the partials in a fixed order (deterministic, no atomics).
//   thin-Cin  (conv_in):  thread = output channel; the 9*Cin input taps of a pixel are broadcast from LDS
//   thin-Cout (conv_out): thread = input channel;  the Cout dy values of a pixel are broadcast from LDS
// Stride 1, no upsampling (what the two layers use); anything else stays on the igemm path.
#include "common.h"

namespace bd {

constexpr int THIN_TP = 64;   // pixels staged per LDS tile
__device__ const float kThinZero = 0.f;

struct ThinGeom {
    int B, H, W, pad_t, pad_l;          // stride-1 conv: output grid == input grid (H, W)
    long long pixels; int pix_per_block;
};

template <int CIN>
__global__ __launch_bounds__(256) void wgrad_thin_cin_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                           long long lddy, ThinGeom g, int Cout,
                                                           float* __restrict__ partial /* [P][Cout*(9*CIN+1)] */) {
    constexpr int NT = 9 * CIN;
    constexpr int LDX = (NT + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float xs[THIN_TP][LDX];
    __shared__ float red[128][NT + 1];
    const int tid = threadIdx.x;
    const int col = tid & 127, half = tid >> 7;
    const int co = blockIdx.y * 128 + col;
    const long long p0 = (long long)blockIdx.x * g.pix_per_block;
    long long p1 = p0 + g.pix_per_block;
    if (p1 > g.pixels) p1 = g.pixels;
    float acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = 0.f;
    float accb = 0.f;
    const int HW = g.H * g.W;
    for (long long pb = p0; pb < p1; pb += THIN_TP) {
        for (int i = tid; i < THIN_TP * NT; i += 256) {   // gather the input taps of THIN_TP pixels (zero padding)
            const int tp = i / NT, j = i - tp * NT;
            const long long p = pb + tp;
            float v = 0.f;
            if (p < p1) {
                const int b = (int)(p / HW), r = (int)(p - (long long)b * HW);
                const int y = r / g.W, xx = r - y * g.W;
                const int tap = j / CIN, ci = j - tap * CIN;
                const int ys = y - g.pad_t + tap / 3, xs_ = xx - g.pad_l + tap % 3;
                if ((unsigned)ys < (unsigned)g.H && (unsigned)xs_ < (unsigned)g.W)
                    v = x[((long long)b * HW + (long long)ys * g.W + xs_) * ldx + ci];
            }
            xs[tp][j] = v;
        }
        __syncthreads();
        const int ntp = (int)((p1 - pb) < THIN_TP ? (p1 - pb) : THIN_TP);
        for (int tp = half; tp < ntp; tp += 2) {
            const float gg = dy[(pb + tp) * lddy + co];
            accb += gg;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] += gg * xs[tp][j];
        }
        __syncthreads();
    }
    if (half == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j) red[col][j] = acc[j];
        red[col][NT] = accb;
    }
    __syncthreads();
    if (half == 0) {
        float* o = partial + (long long)blockIdx.x * Cout * (NT + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) o[(long long)co * NT + j] = acc[j] + red[col][j];
        o[(long long)Cout * NT + co] = accb + red[col][NT];
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void wgrad_thin_cout_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                            long long lddy, ThinGeom g, int Cin,
                                                            float* __restrict__ partial /* [P][COUT*9*Cin + COUT] */) {
    __shared__ float ds[THIN_TP][COUT];
    __shared__ float red[128][COUT * 9];
    __shared__ float redb[4][COUT];
    const int tid = threadIdx.x;
    const int col = tid & 127, half = tid >> 7;
    const int ci = blockIdx.y * 128 + col;
    const long long p0 = (long long)blockIdx.x * g.pix_per_block;
    long long p1 = p0 + g.pix_per_block;
    if (p1 > g.pixels) p1 = g.pixels;
    float acc[COUT][9];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    float accb[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) accb[c] = 0.f;
    const int HW = g.H * g.W;
    for (long long pb = p0; pb < p1; pb += THIN_TP) {
        const int ntp = (int)((p1 - pb) < THIN_TP ? (p1 - pb) : THIN_TP);
        for (int i = tid; i < ntp * COUT; i += 256) ds[i / COUT][i % COUT] = dy[(pb + i / COUT) * lddy + i % COUT];
        __syncthreads();
        // two pixels per iteration, all 18 tap loads issued before the FMAs; out-of-image taps read a zero (no branches,
        // so the loads of both pixels are in flight together)
        for (int tp = half; tp < ntp; tp += 4) {
            float a[2][9], gv[2][COUT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tq = tp + 2 * u;
                const bool live = tq < ntp;
                const long long p = pb + (live ? tq : tp);
                const int b = (int)(p / HW), r = (int)(p - (long long)b * HW);
                const int y = r / g.W, xx = r - y * g.W;
#pragma unroll
                for (int c = 0; c < COUT; ++c) gv[u][c] = live ? ds[live ? tq : tp][c] : 0.f;
                const float* xb = x + (long long)b * HW * ldx + ci;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ys = y - g.pad_t + t / 3, xs_ = xx - g.pad_l + t % 3;
                    const bool ok = (unsigned)ys < (unsigned)g.H && (unsigned)xs_ < (unsigned)g.W;
                    a[u][t] = *(ok ? xb + ((long long)ys * g.W + xs_) * ldx : &kThinZero);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (blockIdx.y == 0 && col == 0) {
#pragma unroll
                    for (int c = 0; c < COUT; ++c) accb[c] += gv[u][c];
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < COUT; ++c) acc[c][t] += gv[u][c] * a[u][t];
            }
        }
        __syncthreads();
    }
    if (half == 1) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) red[col][c * 9 + t] = acc[c][t];
    }
    if (blockIdx.y == 0 && col == 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) redb[half][c] = accb[c];
    }
    __syncthreads();
    float* o = partial + (long long)blockIdx.x * ((long long)COUT * 9 * Cin + COUT);
    if (half == 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) o[((long long)c * 9 + t) * Cin + ci] = acc[c][t] + red[col][c * 9 + t];
    }
    if (blockIdx.y == 0 && tid < COUT) o[(long long)COUT * 9 * Cin + tid] = redb[0][tid] + redb[1][tid];
}

// out[i] = sum over the P partial rows (fixed order: 16 interleaved row phases, then the phases); i < n_w -> dw, the
// rest -> db (if requested).  64 columns x 16 phases per workgroup.
__global__ __launch_bounds__(1024) void thin_reduce_kernel(const float* __restrict__ partial, int P, long long n, long long n_w,
                                                         float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < n)
        for (int p = ty; p < P; p += 16) s += partial[(long long)p * n + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        if (i < n_w) dw[i] = t;
        else if (db) db[i - n_w] = t;
    }
}

// =====================================================================================================================
// Round 3: the same two layers' FORWARD and DATA-GRADIENT passes as direct fp32 streaming kernels (they used to run
// on the generic igemm loaders at 10-17 TFLOP/s: 53-89 us per launch for a 13 us stream of bytes), and wave-per-row
// versions of the two weight gradients above.
//   expand   (few -> many channels): conv_in forward (Cin = 3 -> C) and conv_out data gradient (dy[.,3] -> dx[.,C]):
//            out[p][n] = sum_{tap, j<J} in[nbr(p, tap)][j] * coef[tap][j][n].  Thread = 4 adjacent output channels (its 9*J
//            float4 coefficients stay in registers) x a run of 4 pixels of one image row (the 3 x 6 x J inputs of the run
//            are loaded once); the 32 threads of a run write 512 contiguous bytes per pixel.
//   contract (many -> few): conv_out forward (C = 128 -> 3): wave = 32 adjacent pixels of one image row, lane = 2 channels;
//            a 3 x 3 window of float2 slides along the row (3 coalesced 512-byte loads per pixel), the per-lane partial sums
//            of the 32 pixels are folded across the wave by a halving butterfly (one shuffle per value, not six).
// Both compute modes take these kernels: plain fp32 FMAs are exact products, i.e. at least as accurate as either MFMA path.
__device__ __attribute__((aligned(16))) const float kThinZero4[4] = {0.f, 0.f, 0.f, 0.f};

struct ThinCoef { long long base, sn, st, sj; };   // coefficient of (out channel n, tap t, in channel j) = w[base + n*sn + t*st + j*sj]

template <int J>
__global__ __launch_bounds__(256) void thin_expand_kernel(const float* __restrict__ in, long long ldi, const float* __restrict__ w, ThinCoef cm,
                                                        const float* __restrict__ bias, float* __restrict__ out, long long ldo, int H,
                                                        int W, long long pixels, int iters, float out_scale) {
    const int q = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int n0 = blockIdx.y * 128 + q * 4;
    float4 cf[9 * J];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float* c0 = w + cm.base + (long long)n0 * cm.sn + t * cm.st + j * cm.sj;
            cf[t * J + j] = make_float4(c0[0], c0[cm.sn], c0[2 * cm.sn], c0[3 * cm.sn]);
        }
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int HW = H * W;
    for (int it = 0; it < iters; ++it) {
        const long long p0 = ((long long)blockIdx.x * iters + it) * 32 + slot * 4;   // W % 4 == 0: a run never leaves its row
        if (p0 >= pixels) break;
        const int b = (int)(p0 / HW), r = (int)(p0 - (long long)b * HW);
        const int y = r / W, x0 = r - y * W;
        // the 3 x 6 x J inputs of the run: out-of-image taps read a clamped (valid) pixel and are multiplied by 0 -- no pointer
        // select (it turns the loads into FLAT loads with 64-bit address arithmetic each), one 32-bit offset per position
        float v[3][6][J];
        unsigned roff[3], coff[6]; float rm[3], cmk[6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ys = y - 1 + dy;
            rm[dy] = (unsigned)ys < (unsigned)H ? 1.f : 0.f;
            roff[dy] = (unsigned)(b * HW + min(max(ys, 0), H - 1) * W);
        }
#pragma unroll
        for (int dx = 0; dx < 6; ++dx) {
            const int xs = x0 - 1 + dx;
            cmk[dx] = (unsigned)xs < (unsigned)W ? 1.f : 0.f;
            coff[dx] = (unsigned)min(max(xs, 0), W - 1);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) {
                const float* src = in + (size_t)(roff[dy] + coff[dx]) * (size_t)ldi;
                const float m = rm[dy] * cmk[dx];
#pragma unroll
                for (int j = 0; j < J; ++j) v[dy][dx][j] = src[j] * m;
            }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            float4 a = b4;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const float s = v[t / 3][px + t % 3][j];
                    const float4 c = cf[t * J + j];
                    a.x = fmaf(s, c.x, a.x); a.y = fmaf(s, c.y, a.y); a.z = fmaf(s, c.z, a.z); a.w = fmaf(s, c.w, a.w);
                }
            a.x *= out_scale; a.y *= out_scale; a.z *= out_scale; a.w *= out_scale;
            *reinterpret_cast<float4*>(out + (p0 + px) * ldo + n0) = a;
        }
    }
}

// Round 4: the expand direction on the matrix pipe (BD_MODE_BF16X3 only).  thin_expand_kernel is bound by plain v_fma_f32 issue (3 456 FMAs per pixel
// for 128 output channels: 92 us for a 27 us stream of bytes at 256 x 256).  As a GEMM the layer is [pixels x 27] x [27 x C]: K = 27 pads to 32 = two
// v_mfma_f32_32x32x16_bf16 steps, and the im2col operand needs no LDS: a lane of the A fragment IS one pixel and one octet of K, so it loads its
// own 16 tap values (the 3-channel tensor is tiny: every read hits L1 / L2), splits them into bf16 hi | lo in registers (the same RNE / RNE
// split as everywhere else) and multiplies against weight fragments built once per wave.  24 MFMAs (3 passes x 4 channel tiles x 2 K steps) + 16
// loads + 64 coalesced 128-byte stores per 32 pixels and 128 channels: the kernel writes at the rate of its output.
typedef __bf16 thin_bf16x8 __attribute__((ext_vector_type(8)));
typedef float thin_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void thin_split8(const float (&v)[8], thin_bf16x8& hi, thin_bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];                       // common.h: hi = RNE_bf16(x), lo = RNE_bf16(x - hi)
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}

template <int J>
__global__ __launch_bounds__(256) void thin_expand_mfma_kernel(const float* __restrict__ in, long long ldi, const float* __restrict__ w, ThinCoef cm,
                                                             const float* __restrict__ bias, float* __restrict__ out, long long ldo, int H,
                                                             int W, long long runs, int iters, float out_scale) {
    static_assert(9 * J <= 32, "K = 9 J must fit two 16-deep MFMA steps");
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.y * 128;
    // k = s*16 + h*8 + j  ->  tap t = k / J, channel k % J (k >= 9 J: padding)
    int ktap[16], kch[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (i >> 3) * 16 + h * 8 + (i & 7);
        ktap[i] = k < 9 * J ? k / J : -1;
        kch[i] = k < 9 * J ? k % J : 0;
    }
    // weight fragments: lane = output channel li of tile q, its 16 k values, hi | lo
    thin_bf16x8 bh[4][2], bl[4][2];
    float bn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + q * 32 + li;
        bn[q] = bias ? bias[n] : 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float c[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = s * 8 + j;
                const float v = w[cm.base + (long long)n * cm.sn + (ktap[i] < 0 ? 0 : ktap[i]) * cm.st + kch[i] * cm.sj];
                c[j] = ktap[i] < 0 ? 0.f : v;
            }
            thin_split8(c, bh[q][s], bl[q][s]);
        }
    }
    const int HW = H * W;
    for (int it = 0; it < iters; ++it) {
        const long long run = ((long long)blockIdx.x * 4 + wave) * iters + it;      // wave-uniform
        if (run >= runs) break;
        const long long p0 = run * 32;                 // W % 32 == 0: a run is 32 pixels of one image row
        const int b = (int)(p0 / HW), r = (int)(p0 - (long long)b * HW);
        const int y = r / W, x = r - y * W + li;       // this lane's pixel
        // its 16 tap values: out-of-image taps read a clamped pixel of the same neighbourhood and are multiplied by 0 (no select around the load)
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = ktap[i] < 0 ? 4 : ktap[i];
            const int ys = y - 1 + t / 3, xs = x - 1 + t % 3;
            const float m = (ktap[i] >= 0 && (unsigned)ys < (unsigned)H && (unsigned)xs < (unsigned)W) ? 1.f : 0.f;
            const int yc = min(max(ys, 0), H - 1), xc = min(max(xs, 0), W - 1);
            a[i] = in[((long long)b * HW + (long long)yc * W + xc) * ldi + kch[i]] * m;
        }
        thin_bf16x8 ah[2], al[2];
        {
            float lo8[8], hi8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo8[j] = a[j]; hi8[j] = a[8 + j]; }
            thin_split8(lo8, ah[0], al[0]);
            thin_split8(hi8, ah[1], al[1]);
        }
        thin_f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[q][s], acc[q], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[q][s], acc[q], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[q][s], acc[q], 0, 0, 0);
        }
        // lane holds output channel li of pixels (e & 3) + 8 (e >> 2) + 4 h of the run: 128 contiguous bytes per half-wave and store
        float* ob = out + p0 * ldo + n0 + li;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
                ob[(long long)m * ldo + q * 32] = (acc[q][e] + bn[q]) * out_scale;
            }
    }
}

// halving butterfly: on entry every lane holds NV partial values ("slots", NV a power of two <= 32); on exit lane l holds the
// sum over all 64 lanes of slot l / (64 / NV).  NV - 1 + log2(64 / NV) shuffles for NV values instead of 6 per value.
template <int NV>
__device__ __forceinline__ float thin_fold(float (&v)[NV], int lane) {
#pragma unroll
    for (int half = NV / 2, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
        // lanes with (lane & bit) == 0 keep the lower `half` slots and hand over the upper ones, and vice versa
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            const float keep = up ? v[k + half] : v[k];
            const float give = up ? v[k] : v[k + half];
            v[k] = keep + __shfl_xor(give, bit, 64);
        }
    }
    float s = v[0];
#pragma unroll
    for (int m = 64 / NV / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);   // the 64 / NV lanes that share a slot
    return s;
}

template <int O>
__global__ __launch_bounds__(256, 3) void thin_contract_kernel(const float* __restrict__ a, long long lda, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, long long ldo, int H,
                                                          int W, long long nseg, float out_scale) {
    const int lane = threadIdx.x & 63;
    const long long seg = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (seg >= nseg) return;
    float2 cf[O][9];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) cf[o][t] = *reinterpret_cast<const float2*>(w + ((long long)o * 9 + t) * 128 + lane * 2);
    const int HW = H * W;
    const long long p0 = seg * 32;
    const int b = (int)(p0 / HW), r = (int)(p0 - (long long)b * HW);
    const int y = r / W, x0 = r - y * W;
    const float* base = a + (long long)b * HW * lda + lane * 2;
    // out-of-image taps: the load goes to a clamped (valid) pixel -- one that is inside the same 3 x 3 neighbourhood -- and is
    // multiplied by a 0 / 1 mask.  (A select would be turned back into a branch around the load with s_waitcnt vmcnt(0)
    // behind it: 30 serialised round trips per group; DESIGN.md "a compiler trap".)
    auto ld = [&](int ys, int xs) -> float2 {
        const float m = ((unsigned)ys < (unsigned)H && (unsigned)xs < (unsigned)W) ? 1.f : 0.f;
        const int yc = min(max(ys, 0), H - 1), xc = min(max(xs, 0), W - 1);
        const float2 v = *reinterpret_cast<const float2*>(base + (long long)(yc * W + xc) * lda);
        return make_float2(v.x * m, v.y * m);
    };
    // four groups of GP = 8 pixels: their 3 x 10 neighbourhood is loaded up front (30 coalesced 512-byte loads in flight), then
    // per output channel 8 partial sums per lane are folded across the wave; lane l ends with pixel (l >> 3) of the group
    constexpr int GP = 8;
#pragma unroll 1
    for (int g = 0; g < 32 / GP; ++g) {
        float2 win[3][GP + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < GP + 2; ++dx) win[dy][dx] = ld(y - 1 + dy, x0 + g * GP + dx - 1);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float part[GP];
#pragma unroll
            for (int px = 0; px < GP; ++px) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    s = fmaf(win[t / 3][px + t % 3].x, cf[o][t].x, s);
                    s = fmaf(win[t / 3][px + t % 3].y, cf[o][t].y, s);
                }
                part[px] = s;
            }
            const float s = thin_fold<GP>(part, lane);
            if ((lane & (64 / GP - 1)) == 0) out[(p0 + g * GP + lane / (64 / GP)) * ldo + o] = (s + (bias ? bias[o] : 0.f)) * out_scale;
        }
    }
}

// Weight gradients of the two layers, wave-per-row form (W % 32 == 0): a wave walks 32-pixel row segments, lane = 2 channels of
// the wide tensor (one coalesced 512-byte load per pixel), the narrow tensor's 3 x 3 x J neighbourhood is wave-uniform (scalar
// loads, sliding window), 9*J float2 accumulators per lane.  Four waves fold through LDS in a fixed order, every workgroup
// leaves one partial row, thin_reduce_kernel sums the rows (deterministic).
//   WIDE_IS_DY (conv_in):  dW[co][t][j] = sum_p dy[p][co] * x[p + d(t)][j]          wide = dy, narrow = x
//   otherwise  (conv_out): dW[j][t][c]  = sum_q a[q][c]  * dy[q - d(t)][j]          wide = a,  narrow = dy
template <int J, bool WIDE_IS_DY>
__global__ __launch_bounds__(256) void wgrad_thin_row_kernel(const float* __restrict__ wide, long long ldw, const float* __restrict__ nar,
                                                           long long ldn, int Cw, int H, int W, long long nseg, int segs_per_wave,
                                                           float* __restrict__ partial) {
    __shared__ float2 red[3][9 * J + 1][64];
    __shared__ float redn[4][J];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.y * 128;
    float2 acc[9][J];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[t][j] = make_float2(0.f, 0.f);
    float2 accw = make_float2(0.f, 0.f);
    float accn[J];
#pragma unroll
    for (int j = 0; j < J; ++j) accn[j] = 0.f;
    const int HW = H * W;
    for (int s = 0; s < segs_per_wave; ++s) {
        const long long seg = ((long long)blockIdx.x * 4 + wave) * segs_per_wave + s;
        if (seg >= nseg) break;
        const long long p0 = seg * 32;
        const int b = (int)(p0 / HW), r = (int)(p0 - (long long)b * HW);
        const int y = r / W, x0 = r - y * W;
        const float* nb = nar + (long long)b * HW * ldn;
        float win[3][3][J];
        auto ldcol = [&](int xs, int slot) {      // wave-uniform; out-of-image -> clamped pixel x 0 (no pointer select, no FLAT loads)
            const float mx = (unsigned)xs < (unsigned)W ? 1.f : 0.f;
            const int xc = min(max(xs, 0), W - 1);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ys = y - 1 + dy;
                const float m = (unsigned)ys < (unsigned)H ? mx : 0.f;
                const float* src = nb + ((long long)min(max(ys, 0), H - 1) * W + xc) * ldn;
#pragma unroll
                for (int j = 0; j < J; ++j) win[dy][slot][j] = src[j] * m;
            }
        };
        ldcol(x0 - 1, 1); ldcol(x0, 2);
        const float* wb = wide + ((long long)b * HW + (long long)y * W + x0) * ldw + cb + lane * 2;
#pragma unroll 4
        for (int px = 0; px < 32; ++px) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int j = 0; j < J; ++j) { win[dy][0][j] = win[dy][1][j]; win[dy][1][j] = win[dy][2][j]; }
            ldcol(x0 + px + 1, 2);
            const float2 v = *reinterpret_cast<const float2*>(wb + (long long)px * ldw);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const float nv = WIDE_IS_DY ? win[t / 3][t % 3][j] : win[2 - t / 3][2 - t % 3][j];
                    acc[t][j].x = fmaf(v.x, nv, acc[t][j].x); acc[t][j].y = fmaf(v.y, nv, acc[t][j].y);
                }
            if (WIDE_IS_DY) { accw.x += v.x; accw.y += v.y; }
            else {
#pragma unroll
                for (int j = 0; j < J; ++j) accn[j] += win[1][1][j];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j) red[wave - 1][t * J + j][lane] = acc[t][j];
        red[wave - 1][9 * J][lane] = accw;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) redn[wave][j] = accn[j];
    }
    __syncthreads();
    if (wave != 0) return;
    const long long n_w = (long long)Cw * 9 * J, n = n_w + (WIDE_IS_DY ? Cw : J);
    float* o = partial + (long long)blockIdx.x * n;
    const int c0 = cb + lane * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float2 v = acc[t][j];
#pragma unroll
            for (int k = 0; k < 3; ++k) { v.x += red[k][t * J + j][lane].x; v.y += red[k][t * J + j][lane].y; }
            if (WIDE_IS_DY) { o[(long long)c0 * 9 * J + t * J + j] = v.x; o[(long long)(c0 + 1) * 9 * J + t * J + j] = v.y; }
            else *reinterpret_cast<float2*>(o + ((long long)j * 9 + t) * Cw + c0) = v;
        }
    if (WIDE_IS_DY) {
        float2 v = accw;
#pragma unroll
        for (int k = 0; k < 3; ++k) { v.x += red[k][9 * J][lane].x; v.y += red[k][9 * J][lane].y; }
        *reinterpret_cast<float2*>(o + n_w + c0) = v;
    } else if (blockIdx.y == 0 && lane < J) {
        o[n_w + lane] = ((redn[0][lane] + redn[1][lane]) + redn[2][lane]) + redn[3][lane];
    }
}

static bool thin_same_grid(int stride, int ups, int pad_t, int pad_l, int Hs, int Ws, int Ho, int Wo) {
    return stride == 1 && ups == 0 && pad_t == 1 && pad_l == 1 && Ho == Hs && Wo == Ws;
}

static int thin_expand_launch(int J, const float* in, long long ldi, const float* w, ThinCoef cm, const float* bias, float* out, long long ldo,
                              int N, int B, int H, int W, float out_scale, hipStream_t st, int mode) {
    const long long pixels = (long long)B * H * W;
    const long long runs = cdiv(pixels, 32);
    static const bool mfma_off = getenv("BD_THIN_MFMA") && atoi(getenv("BD_THIN_MFMA")) == 0;      // (A/B knob)
    if (!mfma_off && mode == BD_MODE_BF16X3 && J == 3 && W % 32 == 0 && pixels % 32 == 0) {
        // split-bf16 products like every other convolution of this mode; the exact-fp32 mode keeps the FMA kernel below
        int iters = 8;
        while (iters > 1 && runs / (4 * iters) < 512) iters >>= 1;        // >= 512 workgroups where the layer has them
        const dim3 grid((unsigned)cdiv(runs, 4 * iters), (unsigned)(N / 128)), block(256);
        hipLaunchKernelGGL(thin_expand_mfma_kernel<3>, grid, block, 0, st, in, ldi, w, cm, bias, out, ldo, H, W, runs, iters, out_scale);
        BD_LAUNCH_CHECK("conv3x3 thin expand (mfma)");
        return 1;
    }
    const int iters = (int)(runs >= 4096 ? 4 : 1);      // >= 1024 workgroups per 128-channel block at the UNet's sizes
    const dim3 grid((unsigned)cdiv(runs, iters), (unsigned)(N / 128)), block(256);
    if (J == 3) hipLaunchKernelGGL(thin_expand_kernel<3>, grid, block, 0, st, in, ldi, w, cm, bias, out, ldo, H, W, pixels, iters, out_scale);
    else hipLaunchKernelGGL(thin_expand_kernel<1>, grid, block, 0, st, in, ldi, w, cm, bias, out, ldo, H, W, pixels, iters, out_scale);
    BD_LAUNCH_CHECK("conv3x3 thin expand");
    return 1;
}

// returns 1 when it handled the call, 0 when the shape belongs to the igemm path, < 0 on error
int conv3x3_fwd_thin(const bd_conv3x3_fwd_desc& d, hipStream_t st) {
    static const bool off = getenv("BD_THIN_DIRECT") && atoi(getenv("BD_THIN_DIRECT")) == 0;
    if (off || d.rowbias || d.residual || !thin_same_grid(d.stride, d.ups, d.pad_t, d.pad_l, d.Hs, d.Ws, d.Ho, d.Wo)) return 0;
    const float os = d.out_scale == 0.f ? 1.f : d.out_scale;
    const double flops = 2.0 * d.B * d.Hs * d.Ws * 9.0 * d.Cin * d.Cout, bytes = 4.0 * d.B * d.Hs * d.Ws * (d.Cin + d.Cout);
    if ((d.Cin == 3 || d.Cin == 1) && d.Cout % 128 == 0 && d.Ws % 4 == 0 && d.ldy % 4 == 0 && aligned16(d.y) && (!d.bias || aligned16(d.bias))) {
        const int rec = prof_on() ? prof_begin("conv_thin_fwd", flops, bytes, st) : -1;
        const ThinCoef cm = {0, 9ll * d.Cin, d.Cin, 1};
        const int r = thin_expand_launch(d.Cin, d.x, d.ldx, d.w, cm, d.bias, d.y, d.ldy, d.Cout, d.B, d.Hs, d.Ws, os, st, d.mode);
        prof_end(rec, st);
        return r;
    }
    if ((d.Cout == 3 || d.Cout == 1) && d.Cin == 128 && d.Ws % 32 == 0 && d.ldx % 2 == 0 && ((uintptr_t)d.x & 7) == 0) {
        const int rec = prof_on() ? prof_begin("conv_thin_fwd", flops, bytes, st) : -1;
        const long long nseg = (long long)d.B * d.Hs * d.Ws / 32;
        const dim3 grid((unsigned)cdiv(nseg, 4)), block(256);
        if (d.Cout == 3) hipLaunchKernelGGL(thin_contract_kernel<3>, grid, block, 0, st, d.x, (long long)d.ldx, d.w, d.bias, d.y, (long long)d.ldy, d.Hs, d.Ws, nseg, os);
        else hipLaunchKernelGGL(thin_contract_kernel<1>, grid, block, 0, st, d.x, (long long)d.ldx, d.w, d.bias, d.y, (long long)d.ldy, d.Hs, d.Ws, nseg, os);
        BD_LAUNCH_CHECK("conv3x3 thin contract");
        prof_end(rec, st);
        return 1;
    }
    return 0;
}

int conv3x3_dgrad_thin(const bd_conv3x3_dgrad_desc& d, hipStream_t st) {
    static const bool off = getenv("BD_THIN_DIRECT") && atoi(getenv("BD_THIN_DIRECT")) == 0;
    if (off || d.accumulate || !thin_same_grid(d.stride, d.ups, d.pad_t, d.pad_l, d.Hs, d.Ws, d.Ho, d.Wo)) return 0;
    if (!((d.Cout == 3 || d.Cout == 1) && d.Cin % 128 == 0 && d.Ws % 4 == 0 && d.lddx % 4 == 0 && aligned16(d.dx))) return 0;
    const int rec = prof_on() ? prof_begin("conv_thin_dgrad", 2.0 * d.B * d.Hs * d.Ws * 9.0 * d.Cin * d.Cout, 4.0 * d.B * d.Hs * d.Ws * (d.Cin + d.Cout), st) : -1;
    // dx[p][c] = sum_{t', o} dy[p + d(t')][o] * w[o][8 - t'][c]   (w = [Cout][3][3][Cin])
    const ThinCoef cm = {8ll * d.Cin, 1, -(long long)d.Cin, 9ll * d.Cin};
    const int r = thin_expand_launch(d.Cout, d.dy, d.lddy, d.w, cm, nullptr, d.dx, d.lddx, d.Cin, d.B, d.Hs, d.Ws, 1.f, st, d.mode);
    prof_end(rec, st);
    return r;
}

static bool thin_in_shape(const bd_conv3x3_wgrad_desc& d) { return (d.Cin == 3 || d.Cin == 1) && d.Cout % 128 == 0; }
static bool thin_out_shape(const bd_conv3x3_wgrad_desc& d) { return (d.Cout == 3 || d.Cout == 1) && d.Cin % 128 == 0; }
// shapes these kernels take (they also produce the bias gradient d.db for ANY Cout, unlike the igemm fusion)
bool conv3x3_wgrad_is_thin(const bd_conv3x3_wgrad_desc& d) {
    return d.stride == 1 && d.ups == 0 && d.Ho == d.Hs && d.Wo == d.Ws && (thin_in_shape(d) || thin_out_shape(d));
}

// returns 1 when it handled the call, 0 when the shape belongs to the igemm path, < 0 on error
// Round 4: the two weight gradients on the matrix pipe (BD_MODE_BF16X3, W % 32 == 0).  Both are out[c][k] = sum_p wide[p][c] * col[p][k] with a
// 128-channel-wide tensor and a <= 32-column im2col of the 3-channel one (k = tap * 3 + channel; conv_in: col[p][k] = x[p + d(t)][j] and column 27 = 1
// carries the bias gradient; conv_out: col[q][k] = dy[q - d(t)][o]).  The contraction runs over pixels, so the MFMA A operand is wide^T: lane = channel,
// 8 consecutive pixels per K octet -- each of the 8 loads of a step is a coalesced 128-byte row piece per half-wave -- and the B operand is the lane's
// own im2col column for the same 8 pixels (L1 / L2 hits).  A wave owns 32 channels for the whole pixel slab of its workgroup (no cross-wave fold), a
// workgroup leaves one partial row in the layout thin_reduce_kernel sums (fixed order, deterministic).  wgrad_thin_row_kernel needs 27 float2
// accumulators and 27 x 2 FMAs per lane and pixel: 100 - 146 us against ~35 for the bytes at 256 x 256.
template <bool WIDE_IS_DY>
__global__ __launch_bounds__(256) void wgrad_thin_mfma_kernel(const float* __restrict__ wide, long long ldw, const float* __restrict__ nar, long long ldn,
                                                            int Cw, int H, int W, long long nsteps, int steps_per_wg, long long n_row, long long n_w,
                                                            float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // this wave's 32-channel tile
    const int c0 = blockIdx.y * 128 + q * 32;
    const int t = li < 27 ? li / 3 : 4, ch = li < 27 ? li % 3 : 0;
    const int sgn = WIDE_IS_DY ? 1 : -1;
    const int dyl = sgn * (t / 3 - 1), dxl = sgn * (t % 3 - 1);
    const int HW = H * W;
    thin_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float csum = 0.f;
    const long long s_begin = (long long)blockIdx.x * steps_per_wg;
    long long s_end = s_begin + steps_per_wg;
    if (s_end > nsteps) s_end = nsteps;
    for (long long s = s_begin; s < s_end; ++s) {
        const long long p0 = s * 16;                         // 16 pixels of one image row (W % 32 == 0)
        const int b = (int)(p0 / HW), r = (int)(p0 - (long long)b * HW);
        const int y = r / W, x0 = r - y * W + h * 8;
        float av[8], bv[8];
        const float* wp = wide + (p0 + h * 8) * ldw + c0 + li;
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = wp[(long long)j * ldw];
        const int ys = y + dyl;
        const float rm = (li < 27 && (unsigned)ys < (unsigned)H) ? 1.f : 0.f;
        const int yc = min(max(ys, 0), H - 1);
        const float* np_ = nar + ((long long)b * HW + (long long)yc * W) * ldn + ch;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xs = x0 + j + dxl;
            const float m = (unsigned)xs < (unsigned)W ? rm : 0.f;
            const int xc = min(max(xs, 0), W - 1);
            bv[j] = np_[(long long)xc * ldn] * m;
        }
        if (WIDE_IS_DY && li == 27) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = 1.f;       // bias gradient: sum_p dy[p][c]
        }
        if (!WIDE_IS_DY) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csum += bv[j];     // centre-tap columns: sum_p dy[p][o] = the bias gradient of conv_out
        }
        thin_bf16x8 ah, al, bh, bl;
        thin_split8(av, ah, al);
        thin_split8(bv, bh, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    // lane: column k = li, channels c0 + (e & 3) + 8 (e >> 2) + 4 h
    float* row = partial + (long long)blockIdx.x * n_row;
    if (WIDE_IS_DY) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (li < 27) row[(long long)c * 27 + li] = acc[e];          // dW[co][t][j]
            else if (li == 27) row[n_w + c] = acc[e];                  // db[co]
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (li < 27) row[((long long)ch * 9 + t) * Cw + c] = acc[e]; // dW[o][t][c]
        }
        csum += __shfl_xor(csum, 32, 64);
        if (blockIdx.y == 0 && q == 0 && h == 0 && li >= 12 && li < 15) row[n_w + (li - 12)] = csum;   // db[o]
    }
}

int conv3x3_wgrad_thin(const bd_conv3x3_wgrad_desc& d, hipStream_t st) {
    if (!conv3x3_wgrad_is_thin(d)) return 0;
    const bool thin_in = thin_in_shape(d);
    ThinGeom g;
    g.B = d.B; g.H = d.Hs; g.W = d.Ws; g.pad_t = d.pad_t; g.pad_l = d.pad_l;
    g.pixels = (long long)d.B * d.Hs * d.Ws;
    long long ppb = cdiv(g.pixels, 1024);   // <= 1024 partial rows (four workgroups per CU; the second pass stays ~10 us)
    ppb = cdiv(ppb, THIN_TP) * THIN_TP;
    if (ppb < 2 * THIN_TP) ppb = 2 * THIN_TP;
    g.pix_per_block = (int)ppb;
    const int P = (int)cdiv(g.pixels, ppb);
    const long long n_w = (long long)d.Cout * 9 * d.Cin, n = n_w + d.Cout;
    const size_t need = (size_t)P * n * sizeof(float);
    BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "conv3x3_wgrad (thin): workspace %zu < %zu", d.workspace_bytes,
             need);
    float* part = reinterpret_cast<float*>(d.workspace);
    static const bool mfma_off = getenv("BD_THIN_MFMA") && atoi(getenv("BD_THIN_MFMA")) == 0;      // (A/B knob)
    if (!mfma_off && d.mode == BD_MODE_BF16X3 && d.Ws % 32 == 0 && d.pad_t == 1 && d.pad_l == 1 && (thin_in ? d.Cin == 3 : d.Cout == 3)) {
        const long long nsteps = g.pixels / 16;
        const int spw = (int)cdiv(nsteps, 512);                  // <= 512 partial rows (7 MB for the second pass to fold)
        const int rows = (int)cdiv(nsteps, spw);
        const int Cw = thin_in ? d.Cout : d.Cin;
        BD_CHECK(d.workspace_bytes >= (size_t)rows * n * sizeof(float), BD_ERR_WORKSPACE, "conv3x3_wgrad (thin mfma): workspace too small");
        const dim3 grid((unsigned)rows, (unsigned)(Cw / 128));
        if (thin_in) hipLaunchKernelGGL(wgrad_thin_mfma_kernel<true>, grid, dim3(256), 0, st, d.dy, (long long)d.lddy, d.x, (long long)d.ldx, Cw, d.Hs, d.Ws,
                                        nsteps, spw, n, n_w, part);
        else hipLaunchKernelGGL(wgrad_thin_mfma_kernel<false>, grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, Cw, d.Hs, d.Ws,
                                nsteps, spw, n, n_w, part);
        BD_LAUNCH_CHECK("conv3x3_wgrad_thin_mfma");
        hipLaunchKernelGGL(thin_reduce_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, part, rows, n, n_w, d.dw, d.db);
        BD_LAUNCH_CHECK("conv3x3_wgrad_thin_reduce");
        return 1;
    }
    static const bool row_off = getenv("BD_THIN_DIRECT") && atoi(getenv("BD_THIN_DIRECT")) == 0;
    if (!row_off && d.Ws % 32 == 0 && d.pad_t == 1 && d.pad_l == 1 && (d.Cin == 3 || d.Cout == 3) &&
        (thin_in ? d.lddy : d.ldx) % 2 == 0 && ((uintptr_t)(thin_in ? d.dy : d.x) & 7) == 0) {   // float2 loads of the wide tensor
        const long long nseg = g.pixels / 32;
        const int spw = (int)cdiv(nseg, 1024);
        const int Pr = (int)cdiv(nseg, 4ll * spw);
        const int Cw = thin_in ? d.Cout : d.Cin;
        BD_CHECK(d.workspace_bytes >= (size_t)Pr * n * sizeof(float), BD_ERR_WORKSPACE, "conv3x3_wgrad (thin rows): workspace too small");
        const dim3 grid((unsigned)Pr, (unsigned)(Cw / 128));
        if (thin_in) hipLaunchKernelGGL((wgrad_thin_row_kernel<3, true>), grid, dim3(256), 0, st, d.dy, (long long)d.lddy, d.x, (long long)d.ldx, Cw, d.Hs, d.Ws, nseg, spw, part);
        else hipLaunchKernelGGL((wgrad_thin_row_kernel<3, false>), grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, Cw, d.Hs, d.Ws, nseg, spw, part);
        BD_LAUNCH_CHECK("conv3x3_wgrad_thin_row");
        hipLaunchKernelGGL(thin_reduce_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, part, Pr, n, n_w, d.dw, d.db);
        BD_LAUNCH_CHECK("conv3x3_wgrad_thin_reduce");
        return 1;
    }
    if (thin_in) {
        const dim3 grid((unsigned)P, (unsigned)(d.Cout / 128));
        if (d.Cin == 3) hipLaunchKernelGGL(wgrad_thin_cin_kernel<3>, grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, g, d.Cout, part);
        else hipLaunchKernelGGL(wgrad_thin_cin_kernel<1>, grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, g, d.Cout, part);
    } else {
        const dim3 grid((unsigned)P, (unsigned)(d.Cin / 128));
        if (d.Cout == 3) hipLaunchKernelGGL(wgrad_thin_cout_kernel<3>, grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, g, d.Cin, part);
        else hipLaunchKernelGGL(wgrad_thin_cout_kernel<1>, grid, dim3(256), 0, st, d.x, (long long)d.ldx, d.dy, (long long)d.lddy, g, d.Cin, part);
    }
    BD_LAUNCH_CHECK("conv3x3_wgrad_thin");
    hipLaunchKernelGGL(thin_reduce_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, part, P, n, n_w, d.dw, d.db);
    BD_LAUNCH_CHECK("conv3x3_wgrad_thin_reduce");
    return 1;
}



}  // namespace bd
