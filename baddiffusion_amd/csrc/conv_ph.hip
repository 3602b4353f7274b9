// Phase-decomposed ("sub-pixel") 3x3 convolutions on split planes, fed by LDS-DMA (gfx950) -- round 3.
//
// Upsample2D (resnet.py:126-161) is  y = conv3x3(nearest_up2(x)).  Computed literally (rounds 1-2) the convolution runs on the
// 2H x 2W grid although every output pixel's 3 x 3 window covers only 2 x 2 DISTINCT source pixels: output pixel (2a+p, 2b+q)
// reads source rows {a-1, a} (p = 0) or {a, a+1} (p = 1), and likewise for columns.  Pre-summing the taps that share a source
// pixel,
//        E[oy][ox] = sum_{ky in S(oy)} sum_{kx in S(ox)} W[ky][kx],      S(-1) = {2}, S(0) = {1,2}, S(1) = {0,1}, S(2) = {0},
// gives three exact restatements with 16 instead of 36 tap products per source pixel (2.25x fewer MFMAs; 12.7 % of the
// CIFAR step's flops are these three layers):
//   forward   y[2a+p, 2b+q] = sum_{dy in D(p), dx in D(q)} x[a+dy, b+dx] . E[oy(p,dy)][ox(q,dx)]     D(0) = {-1,0}, D(1) = {0,1}
//             -> four 2x2-tap convolutions ("classes" (p,q)) on the H x W grid, each writing one pixel class of the fine grid;
//   dgrad     dx[a, b] = sum_{oy, ox in -1..2} dY[2a+oy, 2b+ox] . E[oy][ox]^T
//             -> ONE 16-tap convolution that samples the fine-grid gradient at stride 2 (replaces conv dgrad + the 2x2 sum);
//   wgrad     dE[oy][ox] = sum_{a,b} dY[2a+p, 2b+q]^T x[a+dy, b+dx], then dW[ky][kx] = sum over the E entries that contain it
//             (conv_ps.hip's weight-gradient kernel in its PHASE form + ups_dweff_combine below).
// The data gradient of a stride-2 convolution (Downsample2D, resnet.py:199-208) has the same shape as the forward above: by the
// parity of the input pixel only 4 / 2 / 2 / 1 of the 9 taps can contribute, so four classes on the OUTPUT grid replace a 9-tap
// gather in which 3 of 4 products are structurally zero.
// Same arithmetic as conv_ps.hip (x = hi + lo, lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate); the only
// numerical difference to the literal form is that weights are summed (in fp32) before the split, ~1e-7 relative.
#include "common.h"

#include <cstdlib>

namespace bd {

typedef float ph_floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 ph_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ph_bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* ph_lds_ptr;
typedef const __attribute__((address_space(1))) void* ph_gbl_ptr;

constexpr int PH_BM = 256, PH_BN = 128, PH_NT = 512, PH_STAGES = 3;
constexpr int PH_A_BYTES = PH_BM * 128, PH_B_BYTES = PH_BN * 128, PH_STAGE_BYTES = PH_A_BYTES + PH_B_BYTES;
constexpr int PH_LDS_BYTES = PH_STAGES * PH_STAGE_BYTES;
constexpr int PH_MAXT = 16;

struct PhClass {
    int ntaps, p, q, pad_;
    int oy[PH_MAXT], ox[PH_MAXT];   // tap geometry: a_fine ? offsets on the fine grid from (2a, 2b) : offsets on the coarse grid from (a, b)
    int wt[PH_MAXT];                // which of the WT weight taps of a plane row
};
struct PhParams {
    const char* a;       // A split planes (rows = coarse pixels, or fine pixels when a_fine)
    const char* w;       // weight planes: row n, weight tap t, block cb at ((n*WT + t)*C + cb*32)*4 bytes
    float* y; const float* bias;
    long long lda, ldy;
    int C, H, W, lw, lhw;    // contraction channels per tap; COARSE grid (powers of two)
    int M, N, tiles_m, tiles_n, WT;
    int a_fine, y_fine, accumulate;
    float out_scale;
    int ncls;
    int tsplit;          // > 1: ONE class whose taps are dealt to `tsplit` workgroups per tile (blockIdx.y = tap group); every group
    float* partial;      //      leaves a partial tile in partial[group][M][N], ph_tsplit_reduce folds them in fixed order
    PhClass cls[4];
};

__device__ __forceinline__ int ph_swz(int row) { return (row >> 1) & 7; }
__device__ __attribute__((aligned(16))) const float kPhZero[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ void ph_dma16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((ph_gbl_ptr)src, (ph_lds_ptr)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void ph_sync() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// y = out_scale * (acc + bias) (+ y) for a wave's 2 x 2 grid of 32x32 tiles; rows optionally scattered to one pixel class of the fine
// grid.  FULL removes the per-element predicate (behind an exec-masked branch hipcc serialises the stores with s_waitcnt vmcnt(0),
// DESIGN.md "a compiler trap"); the old values of an accumulating store are loaded first, all in flight.
template <bool FULL>
__device__ __forceinline__ void ph_epilogue(const PhParams& p, const ph_floatx16 (&acc)[2][2], int mw, int nw, int li, int h, int yoff) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long rowoff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!FULL && m >= p.M) m = -1;
            // class pixel (2a+p, 2b+q) of the fine grid: img*4HW + (2a+p)*2W + 2b+q = 4m - 2(m & (W-1)) + p*2W + q
            const long long row = p.y_fine ? 4ll * m - 2 * (m & (p.W - 1)) + yoff : (long long)m;
            rowoff[r] = m < 0 ? -1 : row * p.ldy;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = nw + q * 32 + li;
            const float bv = p.bias ? p.bias[n] : 0.f;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (acc[i][q][r] + bv) * p.out_scale;
            if (p.accumulate) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = *((FULL || rowoff[r] >= 0) ? p.y + rowoff[r] + n : kPhZero);   // branch-free
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += old[r];
            }
            if (FULL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p.y[rowoff[r] + n] = v[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (rowoff[r] >= 0) p.y[rowoff[r] + n] = v[r];
            }
        }
    }
}

// 256 x 128 tile, 8 waves of 64 x 64, three LDS stages, one barrier per K chunk (one tap x 32 channels): conv_ps_kernel's
// main loop with (i) a tap TABLE per class instead of the 3 x 3 cursor, (ii) the A rows optionally taken from the fine grid at
// stride 2, (iii) the output rows optionally scattered to one pixel class of the fine grid, (iv) any chunk count.
__global__ __launch_bounds__(PH_NT, 2) void conv_ph_kernel(PhParams p) {
    __shared__ __attribute__((aligned(128))) char smem[PH_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    const int grp = p.tsplit > 1 ? blockIdx.y : 0;            // tap group of a tap-split launch (single class)
    const int cls = p.tsplit > 1 ? 0 : blockIdx.y;
    const PhClass& pc = p.cls[cls];
    const int ntaps = p.tsplit > 1 ? pc.ntaps / p.tsplit : pc.ntaps;
    const int tbase = grp * ntaps;

    int tm, tn;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        tm = j / p.tiles_n;
        tn = j - tm * p.tiles_n;
    }
    const int m0 = tm * PH_BM, n0 = tn * PH_BN;
    const int Wf = p.a_fine ? 2 * p.W : p.W;            // row pitch (pixels) of the grid A lives on
    const int Hv = p.a_fine ? 2 * p.H : p.H, Wv = Wf;   // validity bounds of a tap position on that grid

    const int dr = lane >> 3, ps = lane & 7;
    const char* ap[4];
    int vm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        const int m = m0 + r;
        const int bx = m & (p.W - 1), by = (m >> p.lw) & (p.H - 1), img = m >> p.lhw;
        const int gy = p.a_fine ? 2 * by : by, gx = p.a_fine ? 2 * bx : bx;      // position of the row's origin on A's grid
        int mask = 0;
        for (int t = 0; t < ntaps; ++t) {
            const int yy = gy + pc.oy[tbase + t], xx = gx + pc.ox[tbase + t];
            if ((unsigned)yy < (unsigned)Hv && (unsigned)xx < (unsigned)Wv) mask |= 1 << t;
        }
        vm[j] = m < p.M ? mask : 0;
        const long long row = ((long long)img * Hv + gy) * Wf + gx;
        ap[j] = p.a + row * p.lda * 4 + ((ps ^ ph_swz(r)) << 4);
    }
    const char* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        int n = n0 + r;
        if (n >= p.N) n = p.N - 1;
        wp[j] = p.w + (long long)n * p.WT * p.C * 4 + ((ps ^ ph_swz(r)) << 4);
    }
    const int pix_bytes = (int)p.lda * 4;
    const int nchunks = ntaps * (p.C >> 5);

    // the tap table lives in the lanes of two VGPRs (lane t = tap t) and is fetched with v_readlane: table reads from the kernel
    // arguments inside the loop are s_load + s_waitcnt lgkmcnt(0), which also drains the LDS fragment reads (184 -> TFLOP/s)
    int v_aoff, v_woff;
    {
        const int t = lane & (PH_MAXT - 1);
        v_aoff = (pc.oy[t] * Wf + pc.ox[t]) * pix_bytes;
        v_woff = pc.wt[t] * p.C * 4;
    }
    // cursor of the next chunk to issue (wave-uniform): tap inner, 32-channel block outer
    int q_t = 0, q_cb = 0;
    auto issue = [&](char* stage) {
        const int aoff = __builtin_amdgcn_readlane(v_aoff, tbase + q_t) + q_cb * 128;
        const int woff = __builtin_amdgcn_readlane(v_woff, tbase + q_t) + q_cb * 128;
        const int bit = 1 << q_t;
#pragma unroll
        for (int j = 0; j < 4; ++j) ph_dma16((vm[j] & bit) ? ap[j] + aoff : reinterpret_cast<const char*>(kPhZero), stage + (wave + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) ph_dma16(wp[j] + woff, stage + PH_A_BYTES + (wave + 8 * j) * 1024);
        if (++q_t == ntaps) { q_t = 0; ++q_cb; }
    };

    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) foff[s][pl] = li * 128 + (((pl * 4 + s * 2 + h) ^ ph_swz(li)) << 4);
    const int abase = wm * 64 * 128, bbase = PH_A_BYTES + wn * 64 * 128;

    ph_floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            ph_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const ph_bf16x8*>(stage + abase + i * 4096 + foff[s][0]);
                al[i] = *reinterpret_cast<const ph_bf16x8*>(stage + abase + i * 4096 + foff[s][1]);
                bh[i] = *reinterpret_cast<const ph_bf16x8*>(stage + bbase + i * 4096 + foff[s][0]);
                bl[i] = *reinterpret_cast<const ph_bf16x8*>(stage + bbase + i * 4096 + foff[s][1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[q], acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[q], acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[q], acc[i][q], 0, 0, 0);
        }
    };

    // ring of three stages, any chunk count >= 1: chunk c+2 is issued behind the barrier that retires chunk c-1's stage
    // (three chunks per trip with the stage addresses as compile-time offsets: a rotating pointer triple makes every fragment
    //  address a run-time VALU add)
    char* const s0 = smem; char* const s1 = smem + PH_STAGE_BYTES; char* const s2 = smem + 2 * PH_STAGE_BYTES;
    auto step = [&](const char* cur, char* nxt, int c) {
        if (c + 1 < nchunks) ph_sync<6>(); else ph_sync<0>();
        if (c + 2 < nchunks) issue(nxt);
        compute(cur);
    };
    issue(s0);
    if (nchunks > 1) issue(s1);
    for (int c = 0; c < nchunks; c += 3) {
        step(s0, s2, c);
        if (c + 1 < nchunks) step(s1, s0, c + 1);
        if (c + 2 < nchunks) step(s2, s1, c + 2);
    }

    // ---- epilogue: lane holds column n = li of rows (r&3) + 8*(r>>2) + 4*h of every 32x32 tile
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
    const int yoff = pc.p * 2 * p.W + pc.q;
    if (p.tsplit > 1) {      // partial tile of this tap group: plain rows, no bias / scale / accumulate (the fold applies them)
        float* out = p.partial + (long long)grp * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M) out[(long long)m * p.N + nw + q * 32 + li] = acc[i][q][r];
                }
        return;
    }
    if (mw + 64 <= p.M) ph_epilogue<true>(p, acc, mw, nw, li, h, yoff);     // wave-uniform: whole sub-tile inside M, no predicates
    else ph_epilogue<false>(p, acc, mw, nw, li, h, yoff);
}

// fold of a tap-split launch: y[m][n] (+)= out_scale * sum_g partial[g][m][n], fixed order, float4 per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void ph_tsplit_reduce(const float* __restrict__ part, int groups, long long mn, int N, float* __restrict__ y,
                                                      long long ldy, float out_scale, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (4 * i >= mn) return;
    float4 a = *reinterpret_cast<const float4*>(part + 4 * i);
    for (int g = 1; g < groups; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(part + (long long)g * mn + 4 * i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const long long m = (4 * i) / N;
    const int n = (int)(4 * i - m * N);
    float4* dst = reinterpret_cast<float4*>(y + m * ldy + n);
    a.x *= out_scale; a.y *= out_scale; a.z *= out_scale; a.w *= out_scale;
    if (accumulate) { const float4 o = *dst; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
    *dst = a;
}

// ---- weights of the upsample convolution ----------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ph_pack_hi(float a, float b) { return bd_pack_hi(a, b); }   // common.h: the library's split
__device__ __forceinline__ unsigned ph_pack_lo(float a, float b) { return bd_pack_lo(a, b); }
// S(o) as a bit set over ky: o = -1 -> {2}, 0 -> {1,2}, 1 -> {0,1}, 2 -> {0}
__device__ __host__ __forceinline__ int ph_set(int o) { return o == -1 ? 4 : (o == 0 ? 6 : (o == 1 ? 3 : 1)); }

// W [Cout][3][3][Cin] fp32 -> E planes [Cout][16][Cin] (rows co, tap e = (oy+1)*4 + (ox+1)) and E^T planes [Cin][16][Cout].
// One workgroup = one (tap e, 32 co x 32 ci) block; the transposed copy goes through LDS.
__global__ __launch_bounds__(256) void ups_weff_kernel(const float* __restrict__ w, int Cin, int Cout, unsigned short* __restrict__ e_out,
                                                     unsigned short* __restrict__ et_out) {
    __shared__ float t[32][33];
    const int cib = blockIdx.x, cob = blockIdx.y, e = blockIdx.z;
    const int sy = ph_set(e / 4 - 1), sx = ph_set(e % 4 - 1);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cob * 32 + ty + 8 * i;
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                if ((sy >> ky) & 1 && (sx >> kx) & 1) s += w[((long long)co * 9 + ky * 3 + kx) * Cin + cib * 32 + tx];   // fixed order
        t[ty + 8 * i][tx] = s;
    }
    __syncthreads();
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    {   // E: row co = cob*32 + r, 4 consecutive ci
        const float v0 = t[r][c4], v1 = t[r][c4 + 1], v2 = t[r][c4 + 2], v3 = t[r][c4 + 3];
        unsigned short* o = e_out + 2 * (((long long)(cob * 32 + r) * 16 + e) * Cin + cib * 32) + c4;
        *reinterpret_cast<uint2*>(o) = make_uint2(ph_pack_hi(v0, v1), ph_pack_hi(v2, v3));
        *reinterpret_cast<uint2*>(o + 32) = make_uint2(ph_pack_lo(v0, v1), ph_pack_lo(v2, v3));
    }
    if (et_out) {   // E^T: row ci = cib*32 + r, 4 consecutive co  (the data gradient's operand: null for inference)
        const float v0 = t[c4][r], v1 = t[c4 + 1][r], v2 = t[c4 + 2][r], v3 = t[c4 + 3][r];
        unsigned short* o = et_out + 2 * (((long long)(cib * 32 + r) * 16 + e) * Cout + cob * 32) + c4;
        *reinterpret_cast<uint2*>(o) = make_uint2(ph_pack_hi(v0, v1), ph_pack_hi(v2, v3));
        *reinterpret_cast<uint2*>(o + 32) = make_uint2(ph_pack_lo(v0, v1), ph_pack_lo(v2, v3));
    }
}

// dE [Cout][16][Cin] fp32 -> dW [Cout][3][3][Cin]:  dW[ky][kx] = sum over the (oy, ox) whose sets contain (ky, kx), fixed order
__global__ __launch_bounds__(256) void ups_dweff_combine_kernel(const float* __restrict__ de, int Cin, int Cout, float* __restrict__ dw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)Cout * 9 * (Cin >> 2);
    if (i >= total) return;
    const int c4 = (int)(i % (Cin >> 2)) * 4;
    const long long rt = i / (Cin >> 2);
    const int tap = (int)(rt % 9);
    const long long co = rt / 9;
    const int ky = tap / 3, kx = tap - 3 * ky;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int oy = -1; oy <= 2; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 2; ++ox)
            if ((ph_set(oy) >> ky) & 1 && (ph_set(ox) >> kx) & 1) {
                const float4 v = *reinterpret_cast<const float4*>(de + ((co * 16 + (oy + 1) * 4 + (ox + 1)) * Cin + c4));
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
    *reinterpret_cast<float4*>(dw + (co * 9 + tap) * Cin + c4) = s;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static int ph_ilog2(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// (p, d) -> o for the forward classes: p = 0: d in {-1, 0} -> o in {2, 0};  p = 1: d in {0, 1} -> o in {1, -1}
static int ph_o_of(int p, int d) { return p == 0 ? (d == -1 ? 2 : 0) : (d == 0 ? 1 : -1); }

bool upsample_conv_ps_supported(int B, int H, int W, int Cin, int Cout) {
    static const bool off = getenv("BD_CONV_PHASE") && atoi(getenv("BD_CONV_PHASE")) == 0;
    return !off && B > 0 && ph_ilog2(H) >= 0 && ph_ilog2(W) >= 0 && Cin % 128 == 0 && Cout % 128 == 0;
}

static int ph_launch(PhParams& p, hipStream_t st, const char* what) {
    p.tiles_m = (int)cdiv((long long)p.M, PH_BM); p.tiles_n = p.N / PH_BN;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(p.tsplit > 1 ? p.tsplit : p.ncls)), block(PH_NT);
    hipLaunchKernelGGL(conv_ph_kernel, grid, block, 0, st, p);
    BD_LAUNCH_CHECK(what);
    if (p.tsplit > 1) {
        const long long mn = (long long)p.M * p.N;
        hipLaunchKernelGGL(ph_tsplit_reduce, dim3((unsigned)cdiv(mn / 4, 256)), dim3(256), 0, st, p.partial, p.tsplit, mn, p.N, p.y, p.ldy,
                           p.out_scale, p.accumulate);
        BD_LAUNCH_CHECK("ph_tsplit_reduce");
    }
    return BD_OK;
}
// tiles of a one-class launch on this device's CUs: below half a wave of workgroups the 16 taps are dealt to 4 groups
static int ph_tap_groups(long long M, int N) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    return cdiv(M, PH_BM) * (N / PH_BN) * 2 <= cus ? 4 : 1;
}
size_t upsample_conv_dgrad_workspace_bytes(const bd_upsample_conv_desc& d) {
    const long long M = (long long)d.B * d.H * d.W;
    return ph_tap_groups(M, d.Cin) > 1 ? (size_t)4 * M * d.Cin * sizeof(float) : 0;
}

static int ph_common(PhParams& p, int B, int H, int W, int C, int N, const void* a, long long lda, const void* w, float* y, long long ldy,
                     const char* who) {
    BD_CHECK(a && w && y, BD_ERR_INVALID, "%s: null pointer", who);
    BD_CHECK(B > 0 && ph_ilog2(H) >= 0 && ph_ilog2(W) >= 0, BD_ERR_UNSUPPORTED, "%s: H, W must be powers of two", who);
    BD_CHECK(C > 0 && C % 32 == 0 && N > 0 && N % PH_BN == 0, BD_ERR_UNSUPPORTED, "%s: K channels %% 32 and N channels %% %d must be 0 (got %d, %d)",
             who, PH_BN, C, N);
    BD_CHECK(lda % 32 == 0 && ((uintptr_t)a & 127) == 0 && ((uintptr_t)w & 127) == 0, BD_ERR_UNSUPPORTED,
             "%s: split planes need ld %% 32 == 0 and 128-byte aligned bases", who);
    BD_CHECK((long long)B * H * W * 4 < (1ll << 31) && (long long)(4 * W + 4) * lda * 4 < (1ll << 31), BD_ERR_UNSUPPORTED, "%s: grid too large", who);
    p.a = reinterpret_cast<const char*>(a); p.w = reinterpret_cast<const char*>(w); p.y = y;
    p.lda = lda; p.ldy = ldy; p.C = C; p.H = H; p.W = W; p.lw = ph_ilog2(W); p.lhw = ph_ilog2(W) + ph_ilog2(H);
    p.M = B * H * W; p.N = N; p.out_scale = 1.f;
    return BD_OK;
}

int upsample_weights(const float* w, int Cin, int Cout, uint16_t* e_split, uint16_t* et_split, hipStream_t st) {
    BD_CHECK(w && e_split, BD_ERR_INVALID, "bd_upsample_weights: null pointer");
    BD_CHECK(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0, BD_ERR_UNSUPPORTED, "bd_upsample_weights: channels %% 32 must be 0");
    hipLaunchKernelGGL(ups_weff_kernel, dim3(Cin / 32, Cout / 32, 16), dim3(256), 0, st, w, Cin, Cout, e_split, et_split);
    BD_LAUNCH_CHECK("ups_weff");
    return BD_OK;
}

// y [B, 2H, 2W, Cout] = conv3x3(nearest_up2(x)) + bias, x given as split planes on the SOURCE grid
int upsample_conv_fwd(const bd_upsample_conv_desc& d, hipStream_t st) {
    PhParams p = {};
    BD_TRY(ph_common(p, d.B, d.H, d.W, d.Cin, d.Cout, d.x_split, d.ldx, d.e_split, d.y, d.ldy, "bd_upsample_conv_fwd"));
    p.bias = d.bias; p.WT = 16; p.a_fine = 0; p.y_fine = 1; p.accumulate = 0; p.ncls = 4;
    for (int c = 0; c < 4; ++c) {
        PhClass& k = p.cls[c];
        k.p = c >> 1; k.q = c & 1; k.ntaps = 4;
        for (int t = 0; t < 4; ++t) {
            const int dy = (k.p == 0 ? -1 : 0) + (t >> 1), dx = (k.q == 0 ? -1 : 0) + (t & 1);
            k.oy[t] = dy; k.ox[t] = dx;
            k.wt[t] = (ph_o_of(k.p, dy) + 1) * 4 + (ph_o_of(k.q, dx) + 1);
        }
    }
    const int rec = prof_on() ? prof_begin("conv_ph_ups_fwd", 2.0 * d.B * 4.0 * d.H * d.W * d.Cout * 9.0 * d.Cin,
                                           4.0 * d.B * d.H * d.W * (d.Cin + 4.0 * d.Cout) + 36.0 * d.Cin * d.Cout, st) : -1;
    const int rc = ph_launch(p, st, "conv_ph (upsample forward)");
    prof_end(rec, st);
    return rc;
}

// dx [B, H, W, Cin] (+)= the data gradient of the same layer from dY [B, 2H, 2W, Cout] given as split planes on the FINE grid
int upsample_conv_dgrad(const bd_upsample_conv_desc& d, hipStream_t st) {
    PhParams p = {};
    BD_TRY(ph_common(p, d.B, d.H, d.W, d.Cout, d.Cin, d.dy_split, d.lddy, d.et_split, d.dx, d.lddx, "bd_upsample_conv_dgrad"));
    p.bias = nullptr; p.WT = 16; p.a_fine = 1; p.y_fine = 0; p.accumulate = d.accumulate; p.ncls = 1;
    PhClass& k = p.cls[0];
    k.p = k.q = 0; k.ntaps = 16;
    for (int t = 0; t < 16; ++t) { k.oy[t] = t / 4 - 1; k.ox[t] = t % 4 - 1; k.wt[t] = t; }
    if (ph_tap_groups(p.M, p.N) > 1) {     // few tiles: four workgroups per tile, four taps each, + a fixed-order fold
        const size_t need = upsample_conv_dgrad_workspace_bytes(d);
        BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "bd_upsample_conv_dgrad: workspace %zu < %zu", d.workspace_bytes, need);
        p.tsplit = 4; p.partial = reinterpret_cast<float*>(d.workspace);
    }
    const int rec = prof_on() ? prof_begin("conv_ph_ups_dgrad", 2.0 * d.B * 4.0 * d.H * d.W * d.Cout * 9.0 * d.Cin,
                                           4.0 * d.B * d.H * d.W * (d.Cin + 4.0 * d.Cout) + 36.0 * d.Cin * d.Cout, st) : -1;
    const int rc = ph_launch(p, st, "conv_ph (upsample dgrad)");
    prof_end(rec, st);
    return rc;
}

// dx [B, 2Ho, 2Wo, Cin] (+)= data gradient of a stride-2 3x3 convolution (pad_t = pad_l = pad in {0, 1}; pad 0 = the reference's
// asymmetric F.pad(0,1,0,1) + padding 0) from dy [B, Ho, Wo, Cout] split planes and the transposed weight planes Wt[ci][9][co]
int conv3x3_s2_dgrad_ps(const bd_conv3x3_s2_dgrad_desc& d, hipStream_t st) {
    PhParams p = {};
    BD_TRY(ph_common(p, d.B, d.Ho, d.Wo, d.Cout, d.Cin, d.dy_split, d.lddy, d.wT_split, d.dx, d.lddx, "bd_conv3x3_s2_dgrad_ps"));
    BD_CHECK(d.pad == 0 || d.pad == 1, BD_ERR_UNSUPPORTED, "bd_conv3x3_s2_dgrad_ps: pad must be 0 or 1");
    p.bias = nullptr; p.WT = 9; p.a_fine = 0; p.y_fine = 1; p.accumulate = d.accumulate; p.ncls = 4;
    for (int c = 0; c < 4; ++c) {
        PhClass& k = p.cls[c];
        k.p = c >> 1; k.q = c & 1; k.ntaps = 0;
        // input pixel i = 2a' + p receives dy[a] w[ky] with 2a + ky - pad = i  <=>  ky = p + pad (mod 2), a = a' + (p + pad - ky) / 2
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                if (((k.p + d.pad - ky) & 1) || ((k.q + d.pad - kx) & 1)) continue;
                k.oy[k.ntaps] = (k.p + d.pad - ky) / 2; k.ox[k.ntaps] = (k.q + d.pad - kx) / 2; k.wt[k.ntaps] = ky * 3 + kx;
                ++k.ntaps;
            }
    }
    const int rec = prof_on() ? prof_begin("conv_ph_s2_dgrad", 2.0 * d.B * d.Ho * d.Wo * d.Cout * 9.0 * d.Cin,
                                           4.0 * d.B * d.Ho * d.Wo * (4.0 * d.Cin + d.Cout) + 36.0 * d.Cin * d.Cout, st) : -1;
    const int rc = ph_launch(p, st, "conv_ph (stride-2 dgrad)");
    prof_end(rec, st);
    return rc;
}

int ups_dweff_combine(const float* de, int Cin, int Cout, float* dw, hipStream_t st) {
    const long long total = (long long)Cout * 9 * (Cin / 4);
    hipLaunchKernelGGL(ups_dweff_combine_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, de, Cin, Cout, dw);
    BD_LAUNCH_CHECK("ups_dweff_combine");
    return BD_OK;
}

}  // namespace bd

extern "C" int bd_upsample_weights(const float* w, int Cin, int Cout, uint16_t* e_split, uint16_t* et_split, bd_stream_t s) {
    return bd::upsample_weights(w, Cin, Cout, e_split, et_split, bd::S(s));
}
extern "C" int bd_upsample_conv_fwd(const bd_upsample_conv_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_upsample_conv_fwd: null descriptor");
    return bd::upsample_conv_fwd(*d, bd::S(s));
}
extern "C" size_t bd_upsample_conv_dgrad_workspace_bytes(const bd_upsample_conv_desc* d) {
    return d ? bd::upsample_conv_dgrad_workspace_bytes(*d) : 0;
}
extern "C" int bd_upsample_conv_dgrad(const bd_upsample_conv_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_upsample_conv_dgrad: null descriptor");
    return bd::upsample_conv_dgrad(*d, bd::S(s));
}
extern "C" int bd_conv3x3_s2_dgrad_ps(const bd_conv3x3_s2_dgrad_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_s2_dgrad_ps: null descriptor");
    return bd::conv3x3_s2_dgrad_ps(*d, bd::S(s));
}
