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

static bool thin_in_shape(const bd_conv3x3_wgrad_desc& d) { return (d.Cin == 3 || d.Cin == 1) && d.Cout % 128 == 0; }
static bool thin_out_shape(const bd_conv3x3_wgrad_desc& d) { return (d.Cout == 3 || d.Cout == 1) && d.Cin % 128 == 0; }
// shapes these kernels take (they also produce the bias gradient d.db for ANY Cout, unlike the igemm fusion)
bool conv3x3_wgrad_is_thin(const bd_conv3x3_wgrad_desc& d) {
    return d.stride == 1 && d.ups == 0 && d.Ho == d.Hs && d.Wo == d.Ws && (thin_in_shape(d) || thin_out_shape(d));
}

// returns 1 when it handled the call, 0 when the shape belongs to the igemm path, < 0 on error
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
