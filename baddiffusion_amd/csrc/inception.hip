// Measure path (SURVEY f-3): the kernels of the FID feature extractor -- pytorch_fid's InceptionV3 up to pool3
// (/root/reference/fid_score.py:53, 91-148, 255: `model(batch)[0]`, block index 3 = 2048-d pool3 features).
//
// pytorch_fid (requirements.txt: pytorch-fid==0.2.1) is a third-party dependency that is absent from the reference tree; its
// published network is torchvision's inception_v3 graph with the FIDInceptionA / C / E_1 / E_2 blocks: BasicConv2d = conv (no
// bias) + BatchNorm(eps 1e-3, inference statistics) + ReLU, 3x3 max / average pools, a bilinear resize to 299 x 299
// (align_corners = False) and x -> 2x - 1 in front.  Everything here is exact fp32 arithmetic (v_mfma_f32_32x32x2_f32 for the
// convolutions): the features feed a covariance whose matrix square root amplifies noise.
//
//  * bd_conv2d_nhwc  : generic NHWC convolution (any kernel size / stride / padding; 1x1, 3x3, 5x5, 1x7, 7x1, 1x3, 3x1 occur),
//                      BatchNorm folded into weight and bias by the caller, optional ReLU, output written at a channel offset of a
//                      wider buffer so the block's torch.cat never happens.  Implicit GEMM: M = output pixels, N = Cout,
//                      K = (kh, kw, ci); 64 x 64 tile, 4 waves of 32 x 32, K chunks of 16 staged through LDS.
//  * bd_pool2d_nhwc  : 3x3-style max / average pooling (count_include_pad selectable: the FID blocks use False).
//  * bd_resize_bilinear_nhwc : F.interpolate(mode="bilinear", align_corners=False) of uint8 or float images, then v -> a v + b.
//  * bd_global_avgpool_nhwc  : AdaptiveAvgPool2d((1, 1)).
// These are HBM / latency-level kernels of the measure path, not the train step: ~6 GFLOP (2 x MAC) per 299 x 299 image.
#include "common.h"

namespace bd {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int IC_BM = 64, IC_BN = 64, IC_BK = 16, IC_LD = 96;   // LDS row stride 96 floats: rows k and k+1 sit 32 banks apart

struct IcConvParams {
    const float* x; const float* w; const float* bias; float* y;
    long long ldx, ldy;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, pt, pl, relu;
    long long M;
};

__global__ __launch_bounds__(256) void ic_conv_kernel(IcConvParams p) {
    __shared__ float As[IC_BK][IC_LD];
    __shared__ float Bs[IC_BK][IC_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.x * IC_BM;
    const int n0 = blockIdx.y * IC_BN;

    // A loader: thread -> output pixel am = tid % 64, channel quad akq = tid / 64 (wave-uniform: conflict-free LDS stores)
    const int am = tid & 63, akq = tid >> 6;
    const long long m = m0 + am;
    const bool mv = m < p.M;
    int ab = 0, aoy = 0, aox = 0;
    if (mv) {
        ab = (int)(m / ((long long)p.Ho * p.Wo));
        const int r = (int)(m - (long long)ab * p.Ho * p.Wo);
        aoy = r / p.Wo; aox = r - aoy * p.Wo;
    }
    const int iy0 = aoy * p.sh - p.pt, ix0 = aox * p.sw - p.pl;
    // B loader: thread -> k row bk = tid / 16, output-channel quad bn4 = (tid % 16) * 4
    const int bk = tid >> 4, bn4 = (tid & 15) * 4;
    const bool nv = n0 + bn4 < p.Cout;      // Cout % 4 == 0 (host-checked): a quad is entirely inside or outside
    const bool cin4 = (p.Cin & 3) == 0;

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int kh = 0; kh < p.KH; ++kh) {
        const int iy = iy0 + kh;
        for (int kw = 0; kw < p.KW; ++kw) {
            const int ix = ix0 + kw;
            const bool pv = mv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* xp = p.x + (((long long)ab * p.H + (pv ? iy : 0)) * p.W + (pv ? ix : 0)) * p.ldx;
            const float* wp = p.w + (long long)(kh * p.KW + kw) * p.Cin * p.Cout;
            for (int c0 = 0; c0 < p.Cin; c0 += IC_BK) {
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
                const int ci = c0 + akq * 4;
                if (pv) {
                    if (cin4) {
                        if (ci < p.Cin) { const float4 v = *reinterpret_cast<const float4*>(xp + ci); a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (ci + j < p.Cin) a4[j] = xp[ci + j];
                    }
                }
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nv && c0 + bk < p.Cin) b4 = *reinterpret_cast<const float4*>(wp + (long long)(c0 + bk) * p.Cout + n0 + bn4);
                __syncthreads();          // the previous chunk's fragment reads are done
#pragma unroll
                for (int j = 0; j < 4; ++j) As[akq * 4 + j][am] = a4[j];
                *reinterpret_cast<float4*>(&Bs[bk][bn4]) = b4;
                __syncthreads();
#pragma unroll
                for (int k = 0; k < IC_BK; k += 2) {
                    const float a = As[k + (lane >> 5)][wm * 32 + (lane & 31)];
                    const float b = Bs[k + (lane >> 5)][wn * 32 + (lane & 31)];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }
    // lane holds column n = lane % 32 of rows (r & 3) + 8 (r >> 2) + 4 (lane / 32)
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= p.Cout) return;
    const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long mm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (mm < p.M) {
            float v = acc[r] + bn;
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[mm * p.ldy + n] = v;
        }
    }
}

// ---- round 5: the same convolution on a double-buffered 128 x 64 tile with K-contiguous operands and vector LDS reads.
// Weights [KH][KW][Cout][Cin] (w_kc layout: K contiguous like the activations), so both operand tiles are stored [row][k] (row stride 20 floats:
// 16-byte reads of 16 different rows touch 16 different bank quads) and ONE ds_read_b128 feeds four MFMAs: lane half h holds k = 4h .. 4h+3 of an
// 8-wide K group and MFMA e of the group multiplies (k = e | k = 4 + e) -- a permutation of the contraction order, every k exactly once.
// 4 waves x (32 rows x 64 columns); the global loads of chunk c + 1 are issued before the MFMAs of chunk c (registers), stored to the other LDS
// buffer behind them: one __syncthreads per chunk, 30 KB of LDS, ~5 workgroups per CU hide the rest of the latency.
constexpr int IC2_BM = 128, IC2_BN = 64, IC2_BK = 16, IC2_LD = 20;

template <bool CIN4>
__global__ __launch_bounds__(256) void ic_conv2_kernel(IcConvParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][IC2_BM][IC2_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][IC2_BN][IC2_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const long long m0 = (long long)blockIdx.x * IC2_BM;
    const int n0 = blockIdx.y * IC2_BN;

    // A loader: thread -> output pixel am, 8 consecutive k at akh * 8;  B loader: thread -> output channel bn, 4 consecutive k at bkq * 4
    const int am = tid & 127, akh = tid >> 7;
    const long long m = m0 + am;
    const bool mv = m < p.M;
    int ab = 0, aoy = 0, aox = 0;
    if (mv) {
        ab = (int)(m / ((long long)p.Ho * p.Wo));
        const int r = (int)(m - (long long)ab * p.Ho * p.Wo);
        aoy = r / p.Wo; aox = r - aoy * p.Wo;
    }
    const int iy0 = aoy * p.sh - p.pt, ix0 = aox * p.sw - p.pl;
    const int bn = tid & 63, bkq = tid >> 6;
    const bool nv = n0 + bn < p.Cout;
    const int kchunks = (p.Cin + IC2_BK - 1) / IC2_BK;
    const int nchunks = p.KH * p.KW * kchunks;

    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float4 ra0, ra1, rb;
    float fa0 = 0.f, fa1 = 0.f, fb = 0.f;      // 0 / 1 masks of the three loads, applied when the registers are stored to LDS (not before: the
                                              // loads stay in flight across the MFMAs of the current chunk)
    int kh = 0, kw = 0, kc = 0;            // cursor of the NEXT chunk to load
    auto load = [&]() {
        const int iy = iy0 + kh, ix = ix0 + kw;
        const bool pv = mv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* xp = p.x + (((long long)ab * p.H + (pv ? iy : 0)) * p.W + (pv ? ix : 0)) * p.ldx;
        const int ci = kc * IC2_BK + akh * 8;
        // branch-free loads (an exec-masked `if (valid) load` makes hipcc wait vmcnt(0) behind the branch: the prefetch would be synchronous):
        // the address is clamped into the tensor, the value is multiplied by a 0 / 1 mask
        if constexpr (CIN4) {
            const bool v0 = pv && ci < p.Cin, v1 = pv && ci + 4 < p.Cin;
            ra0 = *reinterpret_cast<const float4*>(xp + (v0 ? ci : 0));
            ra1 = *reinterpret_cast<const float4*>(xp + (v1 ? ci + 4 : 0));
            fa0 = v0 ? 1.f : 0.f; fa1 = v1 ? 1.f : 0.f;
        } else {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const bool v = pv && ci + j < p.Cin; t[j] = xp[v ? ci + j : 0] * (v ? 1.f : 0.f); }
            ra0 = make_float4(t[0], t[1], t[2], t[3]); ra1 = make_float4(t[4], t[5], t[6], t[7]);
            fa0 = fa1 = 1.f;
        }
        const int ck = kc * IC2_BK + bkq * 4;
        {
            const bool vb = nv && ck < p.Cin;
            const float* wp = p.w + ((long long)(kh * p.KW + kw) * p.Cout + (nv ? n0 + bn : 0)) * p.Cin;
            if constexpr (CIN4) {
                rb = *reinterpret_cast<const float4*>(wp + (vb ? ck : 0));
                fb = vb ? 1.f : 0.f;
            } else {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const bool v = vb && ck + j < p.Cin; t[j] = wp[v ? ck + j : 0] * (v ? 1.f : 0.f); }
                rb = make_float4(t[0], t[1], t[2], t[3]);
                fb = 1.f;
            }
        }
        if (++kc == kchunks) { kc = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };
    auto store = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][am][akh * 8]) = make_float4(ra0.x * fa0, ra0.y * fa0, ra0.z * fa0, ra0.w * fa0);
        *reinterpret_cast<float4*>(&As[buf][am][akh * 8 + 4]) = make_float4(ra1.x * fa1, ra1.y * fa1, ra1.z * fa1, ra1.w * fa1);
        *reinterpret_cast<float4*>(&Bs[buf][bn][bkq * 4]) = make_float4(rb.x * fb, rb.y * fb, rb.z * fb, rb.w * fb);
    };

    load();
    store(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) load();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 av = *reinterpret_cast<const float4*>(&As[buf][wave * 32 + li][g * 8 + 4 * h]);
            const float4 b0 = *reinterpret_cast<const float4*>(&Bs[buf][li][g * 8 + 4 * h]);
            const float4 b1 = *reinterpret_cast<const float4*>(&Bs[buf][32 + li][g * 8 + 4 * h]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
    }
    // lane holds columns li (acc0) and 32 + li (acc1) of rows (r & 3) + 8 (r >> 2) + 4 h of the wave's 32 rows
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = n0 + q * 32 + li;
        if (n >= p.Cout) continue;
        const float bnv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long mm = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mm < p.M) {
                float v = (q ? acc1[r] : acc0[r]) + bnv;
                if (p.relu) v = fmaxf(v, 0.f);
                p.y[mm * p.ldy + n] = v;
            }
        }
    }
}

// mode 0 = max, 1 = average (count_include_pad per flag).  One thread = one output pixel x 4 channels.
__global__ __launch_bounds__(256) void ic_pool_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y, long long ldy, int B, int H,
                                                      int W, int C, int Ho, int Wo, int K, int stride, int pad, int mode, int count_include_pad) {
    const int c4n = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Ho * Wo * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const long long pix = i / c4n;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long long)Wo * Ho));
    float4 a = mode == 0 ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * stride - pad + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ox * stride - pad + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((long long)b * H + iy) * W + ix) * ldx + c);
            if (mode == 0) { a.x = fmaxf(a.x, v.x); a.y = fmaxf(a.y, v.y); a.z = fmaxf(a.z, v.z); a.w = fmaxf(a.w, v.w); }
            else { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            ++cnt;
        }
    }
    if (mode == 1) {
        const float d = (float)(count_include_pad ? K * K : cnt);
        a.x /= d; a.y /= d; a.z /= d; a.w /= d;
    }
    *reinterpret_cast<float4*>(y + pix * ldy + c) = a;
}

// F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) (ATen upsample_bilinear2d: source index
// scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out in fp32), then y = a * v + b.  NHWC; the source is uint8 (value / 255:
// what ToTensor yields, fid_score.py:113) or float.
template <typename T>
__global__ __launch_bounds__(256) void ic_resize_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo,
                                                        float sa, float sb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long long)Wo * Ho));
    const float hs = (float)H / (float)Ho, ws = (float)W / (float)Wo;
    float fy = hs * ((float)oy + 0.5f) - 0.5f, fx = ws * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    auto at = [&](int yy, int xx) -> float {
        const T v = x[(((long long)b * H + yy) * W + xx) * C + c];
        if constexpr (sizeof(T) == 1) return (float)v / 255.f;
        else return (float)v;
    };
    const float v = hy * (hx * at(y0, x0) + lx * at(y0, x1)) + ly * (hx * at(y1, x0) + lx * at(y1, x1));
    y[i] = sa * v + sb;
}

// [B, HW, C] -> [B, C]: mean over the pixels (fixed order).  One thread per (b, c).
__global__ __launch_bounds__(256) void ic_gap_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y, int B, int HW, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C) return;
    const int c = (int)(i % C), b = (int)(i / C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((long long)b * HW + p) * ldx + c];
    y[i] = s / (float)HW;
}

}  // namespace bd

using namespace bd;

extern "C" int bd_conv2d_nhwc(const bd_conv2d_desc* d, bd_stream_t stream) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv2d_nhwc: null descriptor");
    BD_CHECK(d->x && d->w && d->y, BD_ERR_INVALID, "bd_conv2d_nhwc: null pointer");
    BD_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride_h > 0 && d->stride_w > 0 &&
                 d->pad_h >= 0 && d->pad_w >= 0, BD_ERR_INVALID, "bd_conv2d_nhwc: bad shape");
    BD_CHECK(d->Cout % 4 == 0, BD_ERR_UNSUPPORTED, "bd_conv2d_nhwc: Cout %% 4 must be 0 (got %d)", d->Cout);
    BD_CHECK(d->ldx >= d->Cin && d->ldy >= d->Cout, BD_ERR_INVALID, "bd_conv2d_nhwc: leading dimension below the channel count");
    BD_CHECK(aligned16(d->w) && (d->Cin % 4 != 0 || (aligned16(d->x) && d->ldx % 4 == 0)), BD_ERR_INVALID,
             "bd_conv2d_nhwc: weights (and x when Cin %% 4 == 0) must be 16-byte aligned with ldx %% 4 == 0");
    IcConvParams p = {};
    p.x = d->x; p.w = d->w; p.bias = d->bias; p.y = d->y; p.ldx = d->ldx; p.ldy = d->ldy;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.sh = d->stride_h; p.sw = d->stride_w; p.pt = d->pad_h; p.pl = d->pad_w; p.relu = d->relu;
    p.Ho = (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1;
    p.Wo = (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1;
    BD_CHECK(p.Ho > 0 && p.Wo > 0, BD_ERR_INVALID, "bd_conv2d_nhwc: empty output");
    p.M = (long long)d->B * p.Ho * p.Wo;
    if (d->w_kc) {      // round 5: weights [KH][KW][Cout][Cin] -> the double-buffered 128 x 64 kernel
        const long long gx2 = cdiv(p.M, IC2_BM);
        BD_CHECK(gx2 < (1ll << 31), BD_ERR_UNSUPPORTED, "bd_conv2d_nhwc: grid too large");
        if (d->Cin % 4 == 0) hipLaunchKernelGGL(ic_conv2_kernel<true>, dim3((unsigned)gx2, (unsigned)cdiv(d->Cout, IC2_BN)), dim3(256), 0, S(stream), p);
        else hipLaunchKernelGGL(ic_conv2_kernel<false>, dim3((unsigned)gx2, (unsigned)cdiv(d->Cout, IC2_BN)), dim3(256), 0, S(stream), p);
        BD_LAUNCH_CHECK("bd_conv2d_nhwc");
        return BD_OK;
    }
    const long long gx = cdiv(p.M, IC_BM);
    BD_CHECK(gx < (1ll << 31), BD_ERR_UNSUPPORTED, "bd_conv2d_nhwc: grid too large");
    hipLaunchKernelGGL(ic_conv_kernel, dim3((unsigned)gx, (unsigned)cdiv(d->Cout, IC_BN)), dim3(256), 0, S(stream), p);
    BD_LAUNCH_CHECK("bd_conv2d_nhwc");
    return BD_OK;
}

extern "C" int bd_pool2d_nhwc(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int H, int W, int C, int kernel, int stride, int pad,
                              int mode, int count_include_pad, bd_stream_t stream) {
    BD_CHECK(x && y, BD_ERR_INVALID, "bd_pool2d_nhwc: null pointer");
    BD_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && kernel > 0 && stride > 0 && pad >= 0 && pad < kernel && (mode == 0 || mode == 1),
             BD_ERR_INVALID, "bd_pool2d_nhwc: bad arguments (C %% 4 must be 0)");
    BD_CHECK(aligned16(x) && aligned16(y) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, BD_ERR_INVALID,
             "bd_pool2d_nhwc: pointers must be 16-byte aligned, leading dimensions multiples of 4 and >= C");
    const int Ho = (H + 2 * pad - kernel) / stride + 1, Wo = (W + 2 * pad - kernel) / stride + 1;
    BD_CHECK(Ho > 0 && Wo > 0, BD_ERR_INVALID, "bd_pool2d_nhwc: empty output");
    const long long n = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(ic_pool_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, S(stream), x, (long long)ldx, y, (long long)ldy, B, H, W, C, Ho, Wo,
                       kernel, stride, pad, mode, count_include_pad);
    BD_LAUNCH_CHECK("bd_pool2d_nhwc");
    return BD_OK;
}

extern "C" int bd_resize_bilinear_nhwc(const void* x, int x_is_u8, float* y, int B, int H, int W, int C, int Ho, int Wo, float scale, float shift,
                                       bd_stream_t stream) {
    BD_CHECK(x && y && B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, BD_ERR_INVALID, "bd_resize_bilinear_nhwc: bad arguments");
    const long long n = (long long)B * Ho * Wo * C;
    if (x_is_u8)
        hipLaunchKernelGGL(ic_resize_kernel<unsigned char>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, S(stream),
                           reinterpret_cast<const unsigned char*>(x), y, B, H, W, C, Ho, Wo, scale, shift);
    else
        hipLaunchKernelGGL(ic_resize_kernel<float>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, S(stream), reinterpret_cast<const float*>(x), y, B, H,
                           W, C, Ho, Wo, scale, shift);
    BD_LAUNCH_CHECK("bd_resize_bilinear_nhwc");
    return BD_OK;
}

extern "C" int bd_global_avgpool_nhwc(const float* x, int64_t ldx, float* y, int B, int HW, int C, bd_stream_t stream) {
    BD_CHECK(x && y && B > 0 && HW > 0 && C > 0 && ldx >= C, BD_ERR_INVALID, "bd_global_avgpool_nhwc: bad arguments");
    hipLaunchKernelGGL(ic_gap_kernel, dim3((unsigned)cdiv((long long)B * C, 256)), dim3(256), 0, S(stream), x, (long long)ldx, y, B, HW, C);
    BD_LAUNCH_CHECK("bd_global_avgpool_nhwc");
    return BD_OK;
}
