// HBM-bound elementwise / reduction kernels of the BadDiffusion hot path (gfx950).
// Compiled with -ffp-contract=off so the fp32 operation order of the reference's eager PyTorch
// ops (separate mul / add kernels, no FMA contraction) is reproduced exactly where the formula
// order is copied (q_sample, scheduler steps).
#include "common.h"

#include <mutex>
#include <string>

namespace bd {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

constexpr int TPB = 256;
static inline unsigned nblocks(int64_t n, int per = TPB, int64_t cap = 1 << 20) {
    int64_t b = cdiv(n, per);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// a-1 + a-2: trigger blend + BadDiffusion q_sample.  One thread per (b, pixel); loops channels.
// dataset.py:275-276, :288-315 ; util.py:111 ; loss.py:264-285 ; scheduling_ddpm.py:429-442
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void poison_qsample_kernel(bd_poison_qsample_desc d) {
    const int64_t hw = (int64_t)d.H * d.W;
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)d.B * hw) return;
    const int b = (int)(idx / hw);
    const int64_t pix = idx - (int64_t)b * hw;
    const int64_t t = d.timesteps[b];
    const float ac = d.alphas_cumprod[t], al = d.alphas[t];
    const float a = sqrtf(ac);                       // alphas_cumprod[t] ** 0.5
    const float s = sqrtf(1.0f - ac);                // (1 - alphas_cumprod[t]) ** 0.5
    const float rho = (1.0f - sqrtf(al)) * s / (1.0f - al);   // loss.py:270
    const bool poison = d.is_poison[b] != 0;
    // fused DataLoader gather + RandomHorizontalFlip (dataset.py:127-128): batch row b reads image row_index[b] of the
    // resident array, mirrored along W when flip[b]
    const int64_t sb = d.row_index ? d.row_index[b] : b;
    int64_t spix = pix;
    if (d.flip && d.flip[b]) {
        const int64_t y = pix / d.W;
        spix = y * d.W + (d.W - 1 - (pix - y * d.W));
    }
    for (int c = 0; c < d.C; ++c) {
        const int64_t chw = (int64_t)c * hw + pix;
        float x;
        if (d.images_u8) {
            // ToTensor (/255) then normalize(0,1 -> -1,1, eps=1e-5): ((x-0)/(1-0+eps))*(2)+(-1)
            float u = (float)d.images_u8[(sb * hw + spix) * d.C + c] / 255.0f;
            x = (u / (1.0f + 1e-5f)) * 2.0f + -1.0f;
        } else {
            x = d.images_f32[sb * d.C * hw + (int64_t)c * hw + spix];
        }
        const float g = d.trigger[chw];
        const float m = g > d.vmin ? 0.0f : 1.0f;                     // get_mask
        float R = 0.0f, x0 = x;
        if (poison) {
            R = m * x + (1.0f - m) * g;                               // dataset.py:312
            x0 = d.target_img[chw];
        }
        const float eps = d.noise[(int64_t)b * d.C * hw + chw];
        const float noisy = a * x0 + s * eps;                         // add_noise
        d.x_noisy[idx * d.ld_noisy + c] = noisy + (1.0f - a) * R;     // loss.py:285
        d.target[idx * d.ld_target + c] = rho * R + eps;
        if (d.R_out) d.R_out[(int64_t)b * d.C * hw + chw] = R;
        if (d.x0_out) d.x0_out[(int64_t)b * d.C * hw + chw] = x0;
        if (d.mask_out && b == 0) d.mask_out[chw] = g > d.vmin ? 0 : 1;
        if (d.image_out) d.image_out[(int64_t)b * d.C * hw + chw] = x;
    }
}

__global__ __launch_bounds__(TPB) void qsample_kernel(bd_qsample_desc d) {
    const int64_t hw = (int64_t)d.H * d.W;
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)d.B * hw) return;
    const int b = (int)(idx / hw);
    const int64_t pix = idx - (int64_t)b * hw;
    const int64_t t = d.timesteps[b];
    const float ac = d.alphas_cumprod[t], al = d.alphas[t];
    const float a = sqrtf(ac), s = sqrtf(1.0f - ac);
    const float rho = (1.0f - sqrtf(al)) * s / (1.0f - al);
    for (int c = 0; c < d.C; ++c) {
        const int64_t src = (int64_t)b * d.C * hw + (int64_t)c * hw + pix;
        const float x0 = d.x0[src], R = d.R[src], eps = d.noise[src];
        const float noisy = a * x0 + s * eps;
        d.x_noisy[idx * d.ld_noisy + c] = noisy + (1.0f - a) * R;
        d.target[idx * d.ld_target + c] = rho * R + eps;
    }
}

__global__ __launch_bounds__(TPB) void nchw_to_nhwc_kernel(const float* src, float* dst, int B, int C, int64_t hw, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)B * hw) return;
    const int b = (int)(idx / hw);
    const int64_t pix = idx - (int64_t)b * hw;
    for (int c = 0; c < C; ++c) dst[idx * ld + c] = src[((int64_t)b * C + c) * hw + pix];
}
__global__ __launch_bounds__(TPB) void nhwc_to_nchw_kernel(const float* src, int64_t ld, float* dst, int B, int C, int64_t hw) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)B * hw) return;
    const int b = (int)(idx / hw);
    const int64_t pix = idx - (int64_t)b * hw;
    for (int c = 0; c < C; ++c) dst[((int64_t)b * C + c) * hw + pix] = src[idx * ld + c];
}

// ------------------------------------------------------------------------------------------------
// a-5: DDPM reverse step (scheduling_ddpm.py:350-415).  Coefficients recomputed per thread from the
// device table in the same fp32 order the reference uses on 0-dim CPU tensors.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void ddpm_step_kernel(bd_ddpm_step_desc d) {
    const float a_t = d.alphas_cumprod[d.t];
    const float a_prev = d.prev_t >= 0 ? d.alphas_cumprod[d.prev_t] : 1.0f;
    const float b_t = 1.0f - a_t, b_prev = 1.0f - a_prev;
    const float cur_alpha = a_t / a_prev, cur_beta = 1.0f - cur_alpha;
    const float sb = sqrtf(b_t), sa = sqrtf(a_t);
    const float c0 = (sqrtf(a_prev) * cur_beta) / b_t;
    const float ct = sqrtf(cur_alpha) * b_prev / b_t;
    float sd = 0.f;
    if (d.t > 0) {
        float var = b_prev / b_t * cur_beta;          // (1 - a_prev) / (1 - a_t) * current_beta_t
        var = fmaxf(var, 1e-20f);
        if (d.variance_type == 1) var = cur_beta;     // fixed_large
        sd = sqrtf(var);
    }
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * TPB) {
        const float x = d.sample[i], e = d.model_output[i];
        float x0 = (x - sb * e) / sa;
        if (d.clip_sample) x0 = fminf(fmaxf(x0, -d.clip_sample_range), d.clip_sample_range);
        float prev = c0 * x0 + ct * x;
        if (d.t > 0) prev = prev + sd * d.noise[i];
        if (d.clip_defense) prev = fminf(fmaxf(prev, -d.clip_defense_range), d.clip_defense_range);
        d.prev_sample[i] = prev;
        if (d.pred_original) d.pred_original[i] = x0;
    }
}

// a-6: DDIM step (scheduling_ddim.py:300-381), epsilon prediction.
__global__ __launch_bounds__(TPB) void ddim_step_kernel(bd_ddim_step_desc d) {
    const float a_t = d.alphas_cumprod[d.t];
    const float a_prev = d.prev_t >= 0 ? d.alphas_cumprod[d.prev_t] : d.final_alpha_cumprod;
    const float b_t = 1.0f - a_t, b_prev = 1.0f - a_prev;
    const float sb = sqrtf(b_t), sa = sqrtf(a_t);
    const float var = (b_prev / b_t) * (1.0f - a_t / a_prev);
    const float sd = d.eta * sqrtf(var);
    const float dir = sqrtf(1.0f - a_prev - sd * sd);
    const float sap = sqrtf(a_prev);
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * TPB) {
        const float x = d.sample[i], e = d.model_output[i];
        float x0 = (x - sb * e) / sa;
        if (d.clip_sample) x0 = fminf(fmaxf(x0, -d.clip_sample_range), d.clip_sample_range);
        float prev = sap * x0 + dir * e;
        if (d.eta > 0.f) prev = prev + sd * d.noise[i];
        d.prev_sample[i] = prev;
        if (d.pred_original) d.pred_original[i] = x0;
    }
}

// a-7: (x/2+0.5).clamp(0,1) -> NHWC float / uint8 (pipeline_ddpm.py:115-116, model.py:499)
__global__ __launch_bounds__(TPB) void to_image_kernel(const float* x, int nhwc, int64_t ld, int B, int C, int64_t hw,
                                                     float* of, uint8_t* ou) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)B * hw) return;
    const int b = (int)(idx / hw);
    const int64_t pix = idx - (int64_t)b * hw;
    for (int c = 0; c < C; ++c) {
        float v = nhwc ? x[idx * ld + c] : x[((int64_t)b * C + c) * hw + pix];
        v = fminf(fmaxf(v / 2.0f + 0.5f, 0.0f), 1.0f);
        if (of) of[idx * C + c] = v;
        if (ou) ou[idx * C + c] = (uint8_t)rintf(v * 255.0f);
    }
}

// a-4a: get_timestep_embedding (embeddings.py:40-57)
__global__ __launch_bounds__(TPB) void timestep_embedding_kernel(const int64_t* t, int t_stride, int B, int dim, int flip,
                                                               float shift, float* out) {
    const int idx = blockIdx.x * TPB + threadIdx.x;
    const int half = dim / 2;
    if (idx >= B * half) return;
    const int b = idx / half, j = idx - b * half;
    const float neg_log = -9.210340371976184f;  // -math.log(10000) rounded to fp32
    float ex = neg_log * (float)j;
    ex = ex / ((float)half - shift);
    const float f = expf(ex);
    const float arg = (float)t[(int64_t)b * t_stride] * f;
    const float sn = sinf(arg), cs = cosf(arg);
    float* o = out + (int64_t)b * dim;
    if (flip) { o[j] = cs; o[half + j] = sn; } else { o[j] = sn; o[half + j] = cs; }
    if ((dim & 1) && j == 0) o[dim - 1] = 0.f;
}

__global__ __launch_bounds__(TPB) void silu_fwd_kernel(const float* x, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float z = x[i];
        y[i] = z / (1.0f + expf(-z));
    }
}
__global__ __launch_bounds__(TPB) void silu_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n, int acc) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float z = x[i];
        const float s = 1.0f / (1.0f + expf(-z));
        float g = dy[i] * (s * (1.0f + z * (1.0f - s)));
        if (acc) g += dx[i];
        dx[i] = g;
    }
}

// ------------------------------------------------------------------------------------------------
// Row softmax (one wave per row) and backward (attention.py:161).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void softmax_fwd_kernel(const float* s, float* p, int64_t rows, int n) {
    const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sr = s + row * n;
    float* pr = p + row * n;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sr[i]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < n; i += 64) sum += expf(sr[i] - mx);
    sum = wave_sum(sum);
    for (int i = lane; i < n; i += 64) pr[i] = expf(sr[i] - mx) / sum;
}
__global__ __launch_bounds__(TPB) void softmax_bwd_kernel(const float* p, const float* dp, float* ds, int64_t rows, int n) {
    const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * n;
    const float* dr = dp + row * n;
    float* o = ds + row * n;
    float dot = 0.f;
    for (int i = lane; i < n; i += 64) dot += pr[i] * dr[i];
    dot = wave_sum(dot);
    for (int i = lane; i < n; i += 64) o[i] = pr[i] * (dr[i] - dot);
}

// ------------------------------------------------------------------------------------------------
// out[g, n] = sum_{m in group g} x[m, n].  Block = 64 columns x 4 row phases.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void colsum_kernel(const float* x, int64_t ldx, int64_t rows, int N, int64_t rpg,
                                                   float* out, int64_t ldo, int acc) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    const int64_t g = blockIdx.y;
    const int64_t m0 = g * rpg;
    int64_t m1 = m0 + rpg;
    if (m1 > rows) m1 = rows;
    float s = 0.f;
    if (n < N)
        for (int64_t m = m0 + ty; m < m1; m += 4) s += x[m * ldx + n];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) {
        float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        float* o = out + g * ldo + n;
        if (acc) v += *o;
        *o = v;
    }
}

// nearest-upsample backward: dx[b,y,x,c] (+)= sum of the 2x2 block of du
__global__ __launch_bounds__(TPB) void sum2x2_kernel(const float* du, int64_t ldu, float* dx, int64_t lddx, int B, int H, int W,
                                                   int C, int acc) {
    const int c4n = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)B * H * W * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    const int64_t pix = idx / c4n;
    const int x = (int)(pix % W);
    const int64_t by = pix / W;  // b*H + y
    const int y = (int)(by % H);
    const int b = (int)(by / H);
    const int64_t W2 = 2 * W;
    const int64_t base = ((int64_t)b * 2 * H + 2 * y) * W2 + 2 * x;
    const float4 v00 = *reinterpret_cast<const float4*>(du + base * ldu + c);
    const float4 v01 = *reinterpret_cast<const float4*>(du + (base + 1) * ldu + c);
    const float4 v10 = *reinterpret_cast<const float4*>(du + (base + W2) * ldu + c);
    const float4 v11 = *reinterpret_cast<const float4*>(du + (base + W2 + 1) * ldu + c);
    float4 r = make_float4((v00.x + v01.x) + (v10.x + v11.x), (v00.y + v01.y) + (v10.y + v11.y),
                           (v00.z + v01.z) + (v10.z + v11.z), (v00.w + v01.w) + (v10.w + v11.w));
    float4* o = reinterpret_cast<float4*>(dx + pix * lddx + c);
    if (acc) { float4 e = *o; r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w; }
    *o = r;
}

// dst[m, c] (+)= scale * src[m, c]   (residual-path gradient, channel-sliced views)
__global__ __launch_bounds__(TPB) void add_kernel(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C,
                                                float scale, int acc) {
    const int c4n = C / 4;
    for (int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x; idx < rows * c4n; idx += (int64_t)gridDim.x * TPB) {
        const int c = (int)(idx % c4n) * 4;
        const int64_t m = idx / c4n;
        float4 v = *reinterpret_cast<const float4*>(src + m * lds + c);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        float4* o = reinterpret_cast<float4*>(dst + m * ldd + c);
        if (acc) { float4 e = *o; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        *o = v;
    }
}

// ------------------------------------------------------------------------------------------------
// a-3 loss + dpred; a-8 sum of squares.  Two-stage deterministic reductions (fixed order).
// ------------------------------------------------------------------------------------------------
constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < TPB / 64; ++i) r += sh[i];
    return r;  // valid on thread 0
}

__global__ __launch_bounds__(TPB) void loss_kernel(const float* pred, int64_t ldp, const float* tgt, int64_t ldt, int64_t rows,
                                                 int C, int type, float gscale, float* dpred, int64_t lddp, double* part) {
    __shared__ double sh[TPB / 64];
    const int64_t n = rows * C;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const int64_t m = i / C;
        const int c = (int)(i - m * C);
        const float p = pred[m * ldp + c], t = tgt[m * ldt + c];
        const float df = p - t;   // d/dpred of loss(target, pred)
        float l, g;
        if (type == 0) { l = df * df; g = 2.0f * df; }
        else if (type == 1) { l = fabsf(df); g = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f); }
        else { float ad = fabsf(df); if (ad < 1.0f) { l = 0.5f * df * df; g = df; } else { l = ad - 0.5f; g = df > 0.f ? 1.f : -1.f; } }
        acc += (double)l;
        if (dpred) dpred[m * lddp + c] = g / (float)n * gscale;
    }
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}
// second stage of the two-stage reductions: one wave, lane l sums partials l, l+64, ... and the lanes are folded with a
// fixed shuffle tree (deterministic)
__device__ __forceinline__ double final_sum(const double* part, int nb) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    return wave_sum(s);
}
__global__ __launch_bounds__(64) void loss_final_kernel(const double* part, int nb, int64_t n, float* loss) {
    const double s = final_sum(part, nb);
    if (threadIdx.x == 0) *loss = (float)(s / (double)n);
}

__global__ __launch_bounds__(TPB) void sumsq_kernel(const float* g, int64_t n, double* part) {
    __shared__ double sh[TPB / 64];
    double acc = 0.0;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; acc += (double)v * v; }
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const double* part, int nb, double* out) {
    const double s = final_sum(part, nb);
    if (threadIdx.x == 0) *out = s;
}

// clip_grad_norm_(max_norm) + torch.optim.Adam (single-tensor formulas), flat buffers
__global__ __launch_bounds__(TPB) void adam_clip_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                      const double* sumsq, float max_norm, float step_size, float omb1,
                                                      float b2, float omb2, float eps, float bc2_sqrt, float* gn_out,
                                                      const float* hyper) {
    if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }
    const float norm = (float)sqrt(*sumsq);
    float coef = max_norm / (norm + 1e-6f);
    coef = fminf(coef, 1.0f);
    if (gn_out && blockIdx.x == 0 && threadIdx.x == 0) *gn_out = norm;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float gi = g[i] * coef;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * b2 + omb2 * gi * gi;                     // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);            // addcdiv_(exp_avg, denom, -step_size)
        m[i] = mi;
        v[i] = vi;
    }
}


// f-4 (PNDM family): out = clamp(sum_j c[j] * term[j]) over up to BD_LINCOMB_MAX dense fp32 tensors of one layout.
// Every PRK / PLMS update of scheduling_pndm.py:236-397 is such a combination of the running sample, the stored model
// outputs and the running accumulator, with host-computed scalar coefficients; the pipeline's post-step clip
// (pipeline_pndm.py:108-109) rides along.
struct LincombArgs { const float* t[BD_LINCOMB_MAX]; float c[BD_LINCOMB_MAX]; int k; int clip; float range; };
__global__ __launch_bounds__(TPB) void lincomb_kernel(LincombArgs a, float* __restrict__ out, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 v[BD_LINCOMB_MAX];
#pragma unroll
        for (int j = 0; j < BD_LINCOMB_MAX; ++j)
            v[j] = j < a.k ? reinterpret_cast<const float4*>(a.t[j])[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < BD_LINCOMB_MAX; ++j)
            if (j < a.k) { r.x += a.c[j] * v[j].x; r.y += a.c[j] * v[j].y; r.z += a.c[j] * v[j].z; r.w += a.c[j] * v[j].w; }
        if (a.clip) {
            r.x = fminf(fmaxf(r.x, -a.range), a.range); r.y = fminf(fmaxf(r.y, -a.range), a.range);
            r.z = fminf(fmaxf(r.z, -a.range), a.range); r.w = fminf(fmaxf(r.w, -a.range), a.range);
        }
        reinterpret_cast<float4*>(out)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail (n % 4 elements)
        const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
        float r = 0.f;
        for (int j = 0; j < a.k; ++j) r += a.c[j] * a.t[j][i];
        if (a.clip) r = fminf(fmaxf(r, -a.range), a.range);
        out[i] = r;
    }
}


// ---- f-4: Adversarial Neuron Pruning (anp_model.py:490-514 PerturbConv2d = conv followed by an eval-mode batch norm with mean 0, variance 1,
// eps 0, i.e. y_c <- w_c * conv(x; W, b)_c + b_c with ONE learnable (w_c, b_c) per output channel).  The affine map commutes with the convolution:
//   w_c * (W_c . x + b_c0) + b_c = (w_c W_c) . x + (w_c b_c0 + b_c),
// so the perturbed network IS the ordinary network on effective weights (anp_apply: one scaled copy of every conv weight row and bias), and the
// gradient of the perturbation is a row-wise contraction of the ordinary weight gradient (anp_grad):
//   dL/dw_c = sum_k dL/dW'_ck W_ck + dL/db'_c b_c0 ,   dL/db_c = dL/db'_c.
// items: [n][5] int64 = (weight offset, bias offset or -1, Cout, row length, offset of the layer's channels in the perturbation vectors);
// one workgroup per output channel of every convolution, fixed-order fold (deterministic).
__device__ __forceinline__ void anp_locate(const int64_t* __restrict__ items, int n, int64_t row, int64_t& woff, int64_t& boff, int64_t& len,
                                           int64_t& p) {
    int64_t base = 0;
    woff = boff = -1; len = 0; p = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t cout = items[i * 5 + 2];
        if (row < base + cout) {
            const int64_t c = row - base;
            len = items[i * 5 + 3];
            woff = items[i * 5 + 0] + c * len;
            boff = items[i * 5 + 1] < 0 ? -1 : items[i * 5 + 1] + c;
            p = items[i * 5 + 4] + c;
            return;
        }
        base += cout;
    }
}
__global__ __launch_bounds__(TPB) void anp_apply_kernel(const float* __restrict__ params, const float* __restrict__ pw,
                                                        const float* __restrict__ pb, const int64_t* __restrict__ items, int n,
                                                        float* __restrict__ eff) {
    int64_t woff, boff, len, p;
    anp_locate(items, n, blockIdx.x, woff, boff, len, p);
    if (woff < 0) return;
    const float w = pw[p];
    for (int64_t k = threadIdx.x; k < len; k += TPB) eff[woff + k] = w * params[woff + k];
    if (threadIdx.x == 0 && boff >= 0) eff[boff] = w * params[boff] + pb[p];
}
// row_norm (optional): the norm of the gradient of the layer's OWN weight row and bias in the perturbed network, |w_c| * ||(dL/dW'_c, dL/db'_c)|| --
// the reference's clip_grad_norm_ runs over them too (PerturbConv2d's weights are fresh, trainable Parameters: anp_model.py:492-505)
__global__ __launch_bounds__(TPB) void anp_grad_kernel(const float* __restrict__ params, const float* __restrict__ geff,
                                                       const int64_t* __restrict__ items, int n, const float* __restrict__ pw,
                                                       float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ row_norm) {
    __shared__ float red[TPB / 64][2];
    int64_t woff, boff, len, p;
    anp_locate(items, n, blockIdx.x, woff, boff, len, p);
    if (woff < 0) return;
    float a = 0.f, q = 0.f;
    for (int64_t k = threadIdx.x; k < len; k += TPB) {
        const float g = geff[woff + k];
        a += g * params[woff + k];
        q += g * g;
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < TPB / 64; ++i) { t += red[i][0]; tq += red[i][1]; }
        const float g = boff >= 0 ? geff[boff] : 0.f;
        gw[p] = t + (boff >= 0 ? g * params[boff] : 0.f);
        gb[p] = g;
        if (row_norm) row_norm[p] = fabsf(pw[p]) * sqrtf(tq + g * g);
    }
}

}  // namespace bd

using namespace bd;

extern "C" const char* bd_last_error(void) { return g_err.c_str(); }
extern "C" int bd_version(void) { return 1; }

extern "C" int bd_poison_qsample(const bd_poison_qsample_desc* d, bd_stream_t stream) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_poison_qsample: null descriptor");
    BD_CHECK(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0, BD_ERR_INVALID, "bd_poison_qsample: bad shape");
    BD_CHECK((d->images_f32 != nullptr) != (d->images_u8 != nullptr), BD_ERR_INVALID,
             "bd_poison_qsample: exactly one of images_f32 / images_u8 must be given");
    BD_CHECK(d->is_poison && d->trigger && d->target_img && d->noise && d->timesteps && d->alphas && d->alphas_cumprod &&
                 d->x_noisy && d->target, BD_ERR_INVALID, "bd_poison_qsample: null pointer");
    BD_CHECK(d->ld_noisy >= d->C && d->ld_target >= d->C, BD_ERR_INVALID, "bd_poison_qsample: ld < C");
    hipLaunchKernelGGL(poison_qsample_kernel, dim3(nblocks((int64_t)d->B * d->H * d->W)), dim3(TPB), 0, S(stream), *d);
    BD_LAUNCH_CHECK("poison_qsample");
    return BD_OK;
}
extern "C" int bd_qsample(const bd_qsample_desc* d, bd_stream_t stream) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_qsample: null descriptor");
    BD_CHECK(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0, BD_ERR_INVALID, "bd_qsample: bad shape");
    BD_CHECK(d->x0 && d->R && d->noise && d->timesteps && d->alphas && d->alphas_cumprod && d->x_noisy && d->target,
             BD_ERR_INVALID, "bd_qsample: null pointer");
    BD_CHECK(d->ld_noisy >= d->C && d->ld_target >= d->C, BD_ERR_INVALID, "bd_qsample: ld < C");
    hipLaunchKernelGGL(qsample_kernel, dim3(nblocks((int64_t)d->B * d->H * d->W)), dim3(TPB), 0, S(stream), *d);
    BD_LAUNCH_CHECK("qsample");
    return BD_OK;
}
extern "C" int bd_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int64_t ld, bd_stream_t stream) {
    BD_CHECK(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && ld >= C, BD_ERR_INVALID, "bd_nchw_to_nhwc: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks((int64_t)B * H * W)), dim3(TPB), 0, S(stream), src, dst, B, C,
                       (int64_t)H * W, ld);
    BD_LAUNCH_CHECK("nchw_to_nhwc");
    return BD_OK;
}
extern "C" int bd_nhwc_to_nchw(const float* src, int64_t ld, float* dst, int B, int C, int H, int W, bd_stream_t stream) {
    BD_CHECK(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && ld >= C, BD_ERR_INVALID, "bd_nhwc_to_nchw: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks((int64_t)B * H * W)), dim3(TPB), 0, S(stream), src, ld, dst, B, C,
                       (int64_t)H * W);
    BD_LAUNCH_CHECK("nhwc_to_nchw");
    return BD_OK;
}
extern "C" int bd_ddpm_step(const bd_ddpm_step_desc* d, bd_stream_t stream) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_ddpm_step: null descriptor");
    BD_CHECK(d->n > 0 && d->model_output && d->sample && d->prev_sample && d->alphas_cumprod, BD_ERR_INVALID,
             "bd_ddpm_step: bad args");
    BD_CHECK(d->t >= 0, BD_ERR_INVALID, "bd_ddpm_step: t < 0");
    BD_CHECK(d->t == 0 || d->noise, BD_ERR_INVALID, "bd_ddpm_step: noise is required when t > 0");
    BD_CHECK(d->variance_type == 0 || d->variance_type == 1, BD_ERR_UNSUPPORTED,
             "bd_ddpm_step: variance_type must be fixed_small(0) or fixed_large(1)");
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(nblocks(d->n, TPB, 4096)), dim3(TPB), 0, S(stream), *d);
    BD_LAUNCH_CHECK("ddpm_step");
    return BD_OK;
}
extern "C" int bd_lincomb(int k, const float* const* terms, const float* coeffs, int64_t n, int clip, float clip_range, float* out,
                          bd_stream_t stream) {
    BD_CHECK(k >= 1 && k <= BD_LINCOMB_MAX && terms && coeffs && out && n > 0, BD_ERR_INVALID, "bd_lincomb: bad args (k=%d)", k);
    LincombArgs a = {};
    a.k = k; a.clip = clip; a.range = clip_range;
    for (int j = 0; j < k; ++j) {
        BD_CHECK(terms[j] && aligned16(terms[j]), BD_ERR_INVALID, "bd_lincomb: term %d is null or not 16-byte aligned", j);
        a.t[j] = terms[j]; a.c[j] = coeffs[j];
    }
    BD_CHECK(aligned16(out), BD_ERR_INVALID, "bd_lincomb: out must be 16-byte aligned");
    hipLaunchKernelGGL(lincomb_kernel, dim3(nblocks(cdiv(n, 4), TPB, 4096)), dim3(TPB), 0, S(stream), a, out, n / 4, n);
    BD_LAUNCH_CHECK("lincomb");
    return BD_OK;
}
extern "C" int bd_ddim_step(const bd_ddim_step_desc* d, bd_stream_t stream) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_ddim_step: null descriptor");
    BD_CHECK(d->n > 0 && d->model_output && d->sample && d->prev_sample && d->alphas_cumprod, BD_ERR_INVALID,
             "bd_ddim_step: bad args");
    BD_CHECK(d->t >= 0, BD_ERR_INVALID, "bd_ddim_step: t < 0");
    BD_CHECK(d->eta == 0.f || d->noise, BD_ERR_INVALID, "bd_ddim_step: noise is required when eta > 0");
    hipLaunchKernelGGL(ddim_step_kernel, dim3(nblocks(d->n, TPB, 4096)), dim3(TPB), 0, S(stream), *d);
    BD_LAUNCH_CHECK("ddim_step");
    return BD_OK;
}
extern "C" int bd_to_image(const float* x, int src_is_nhwc, int64_t ld, int B, int C, int H, int W, float* out_f32,
                           uint8_t* out_u8, bd_stream_t stream) {
    BD_CHECK(x && (out_f32 || out_u8) && B > 0 && C > 0 && H > 0 && W > 0, BD_ERR_INVALID, "bd_to_image: bad args");
    hipLaunchKernelGGL(to_image_kernel, dim3(nblocks((int64_t)B * H * W)), dim3(TPB), 0, S(stream), x, src_is_nhwc, ld, B, C,
                       (int64_t)H * W, out_f32, out_u8);
    BD_LAUNCH_CHECK("to_image");
    return BD_OK;
}
extern "C" int bd_timestep_embedding(const int64_t* t, int t_stride, int B, int dim, int flip_sin_to_cos, float freq_shift,
                                     float* out, bd_stream_t stream) {
    BD_CHECK(t && out && B > 0 && dim >= 2, BD_ERR_INVALID, "bd_timestep_embedding: bad args");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblocks((int64_t)B * (dim / 2))), dim3(TPB), 0, S(stream), t, t_stride,
                       B, dim, flip_sin_to_cos, freq_shift, out);
    BD_LAUNCH_CHECK("timestep_embedding");
    return BD_OK;
}
extern "C" int bd_silu_fwd(const float* x, float* y, int64_t n, bd_stream_t stream) {
    BD_CHECK(x && y && n > 0, BD_ERR_INVALID, "bd_silu_fwd: bad args");
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(nblocks(n, TPB, 4096)), dim3(TPB), 0, S(stream), x, y, n);
    BD_LAUNCH_CHECK("silu_fwd");
    return BD_OK;
}
extern "C" int bd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, int accumulate, bd_stream_t stream) {
    BD_CHECK(x && dy && dx && n > 0, BD_ERR_INVALID, "bd_silu_bwd: bad args");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(nblocks(n, TPB, 4096)), dim3(TPB), 0, S(stream), x, dy, dx, n, accumulate);
    BD_LAUNCH_CHECK("silu_bwd");
    return BD_OK;
}
extern "C" int bd_softmax_fwd(const float* s, float* p, int64_t rows, int n, bd_stream_t stream) {
    BD_CHECK(s && p && rows > 0 && n > 0, BD_ERR_INVALID, "bd_softmax_fwd: bad args");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)cdiv(rows, TPB / 64)), dim3(TPB), 0, S(stream), s, p, rows, n);
    BD_LAUNCH_CHECK("softmax_fwd");
    return BD_OK;
}
extern "C" int bd_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int n, bd_stream_t stream) {
    BD_CHECK(p && dp && ds && rows > 0 && n > 0, BD_ERR_INVALID, "bd_softmax_bwd: bad args");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)cdiv(rows, TPB / 64)), dim3(TPB), 0, S(stream), p, dp, ds, rows, n);
    BD_LAUNCH_CHECK("softmax_bwd");
    return BD_OK;
}
extern "C" int bd_colsum(const float* x, int64_t ldx, int64_t rows, int N, int64_t rows_per_group, float* out, int64_t ld_out,
                         int accumulate, bd_stream_t stream) {
    BD_CHECK(x && out && rows > 0 && N > 0 && rows_per_group > 0, BD_ERR_INVALID, "bd_colsum: bad args");
    const int64_t groups = cdiv(rows, rows_per_group);
    BD_CHECK(groups <= 65535, BD_ERR_UNSUPPORTED, "bd_colsum: too many groups (%lld)", (long long)groups);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)groups), dim3(TPB), 0, S(stream), x, ldx, rows, N,
                       rows_per_group, out, ld_out, accumulate);
    BD_LAUNCH_CHECK("colsum");
    return BD_OK;
}
extern "C" int bd_sum2x2(const float* du, int64_t ldu, float* dx, int64_t lddx, int B, int H, int W, int C, int accumulate,
                         bd_stream_t stream) {
    BD_CHECK(du && dx && B > 0 && H > 0 && W > 0 && C > 0, BD_ERR_INVALID, "bd_sum2x2: bad args");
    BD_CHECK((C & 3) == 0 && (ldu & 3) == 0 && (lddx & 3) == 0 && aligned16(du) && aligned16(dx), BD_ERR_UNSUPPORTED,
             "bd_sum2x2: needs C, ld multiples of 4 and 16B-aligned pointers");
    hipLaunchKernelGGL(sum2x2_kernel, dim3(nblocks((int64_t)B * H * W * (C / 4))), dim3(TPB), 0, S(stream), du, ldu, dx, lddx, B,
                       H, W, C, accumulate);
    BD_LAUNCH_CHECK("sum2x2");
    return BD_OK;
}
namespace bd {
int add_launch(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C, float scale, int acc,
               hipStream_t st) {
    BD_CHECK(src && dst && rows > 0 && C > 0, BD_ERR_INVALID, "add: bad args");
    BD_CHECK((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && aligned16(src) && aligned16(dst), BD_ERR_UNSUPPORTED,
             "add: needs C, ld multiples of 4 and 16B-aligned pointers");
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(rows * (C / 4), TPB, 8192)), dim3(TPB), 0, st, src, lds, dst, ldd, rows, C, scale,
                       acc);
    BD_LAUNCH_CHECK("add");
    return BD_OK;
}
}  // namespace bd
extern "C" size_t bd_reduce_workspace_bytes(void) { return RED_BLOCKS * sizeof(double); }
extern "C" int bd_loss_fwd_bwd(const float* pred, int64_t ldp, const float* target, int64_t ldt, int64_t rows, int C,
                               int loss_type, float grad_scale, float* loss, float* dpred, int64_t lddp, void* workspace,
                               bd_stream_t stream) {
    BD_CHECK(pred && target && loss && workspace && rows > 0 && C > 0, BD_ERR_INVALID, "bd_loss_fwd_bwd: bad args");
    BD_CHECK(loss_type >= 0 && loss_type <= 2, BD_ERR_UNSUPPORTED, "bd_loss_fwd_bwd: loss_type %d (l2=0,l1=1,huber=2)", loss_type);
    const int nb = (int)nblocks(rows * C, TPB, RED_BLOCKS);
    hipLaunchKernelGGL(loss_kernel, dim3(nb), dim3(TPB), 0, S(stream), pred, ldp, target, ldt, rows, C, loss_type, grad_scale,
                       dpred, lddp, (double*)workspace);
    BD_LAUNCH_CHECK("loss");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, S(stream), (const double*)workspace, nb, rows * C, loss);
    BD_LAUNCH_CHECK("loss_final");
    return BD_OK;
}
extern "C" int bd_sumsq(const float* g, int64_t n, double* sumsq, void* workspace, bd_stream_t stream) {
    BD_CHECK(g && sumsq && workspace && n > 0 && aligned16(g), BD_ERR_INVALID, "bd_sumsq: bad args");
    const int nb = (int)nblocks(n / 4 + 1, TPB, RED_BLOCKS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(TPB), 0, S(stream), g, n, (double*)workspace);
    BD_LAUNCH_CHECK("sumsq");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, S(stream), (const double*)workspace, nb, sumsq);
    BD_LAUNCH_CHECK("sumsq_final");
    return BD_OK;
}
extern "C" int bd_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, double max_norm,
                            double lr, double b1, double b2, double eps, int step, float* grad_norm_out, bd_stream_t stream) {
    BD_CHECK(p && g && m && v && sumsq && n > 0 && step >= 1, BD_ERR_INVALID, "bd_adam_clip: bad args");
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_clip_kernel, dim3(nblocks(n, TPB, 8192)), dim3(TPB), 0, S(stream), p, g, m, v, n, sumsq, (float)max_norm,
                       step_size, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, bc2_sqrt, grad_norm_out,
                       (const float*)nullptr);
    BD_LAUNCH_CHECK("adam_clip");
    return BD_OK;
}
extern "C" int bd_adam_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, double max_norm,
                                const float* hyper, double b1, double b2, double eps, float* grad_norm_out, bd_stream_t stream) {
    BD_CHECK(p && g && m && v && sumsq && hyper && n > 0, BD_ERR_INVALID, "bd_adam_clip_dev: bad args");
    hipLaunchKernelGGL(adam_clip_kernel, dim3(nblocks(n, TPB, 8192)), dim3(TPB), 0, S(stream), p, g, m, v, n, sumsq, (float)max_norm,
                       0.f, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, 1.f, grad_norm_out, hyper);
    BD_LAUNCH_CHECK("adam_clip_dev");
    return BD_OK;
}
__global__ __launch_bounds__(256) void axpy_kernel(const float* src, float* dst, int64_t n, float scale, int acc) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = src[i] * scale;
        if (acc) v += dst[i];
        dst[i] = v;
    }
}
extern "C" int bd_axpy(const float* src, float* dst, int64_t n, float scale, int accumulate, bd_stream_t stream) {
    BD_CHECK(src && dst && n > 0, BD_ERR_INVALID, "bd_axpy: bad args");
    hipLaunchKernelGGL(axpy_kernel, dim3(nblocks(n, TPB, 8192)), dim3(TPB), 0, S(stream), src, dst, n, scale, accumulate);
    BD_LAUNCH_CHECK("axpy");
    return BD_OK;
}

extern "C" int bd_anp_apply(const float* params, int64_t nparams, const float* pert_w, const float* pert_b, const int64_t* items, int n_items,
                            int64_t total_rows, float* eff, bd_stream_t stream) {
    BD_CHECK(params && pert_w && pert_b && items && eff && nparams > 0 && n_items > 0 && total_rows > 0 && total_rows < (1ll << 31), BD_ERR_INVALID,
             "bd_anp_apply: bad args");
    BD_HIP_TRY(hipMemcpyAsync(eff, params, (size_t)nparams * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    hipLaunchKernelGGL(anp_apply_kernel, dim3((unsigned)total_rows), dim3(TPB), 0, S(stream), params, pert_w, pert_b, items, n_items, eff);
    BD_LAUNCH_CHECK("anp_apply");
    return BD_OK;
}
extern "C" int bd_anp_grad(const float* params, const float* grad_eff, const int64_t* items, int n_items, int64_t total_rows, const float* pert_w,
                           float* grad_w, float* grad_b, float* row_norm, bd_stream_t stream) {
    BD_CHECK(params && grad_eff && items && grad_w && grad_b && n_items > 0 && total_rows > 0 && total_rows < (1ll << 31), BD_ERR_INVALID,
             "bd_anp_grad: bad args");
    BD_CHECK(!row_norm || pert_w, BD_ERR_INVALID, "bd_anp_grad: row_norm needs pert_w");
    hipLaunchKernelGGL(anp_grad_kernel, dim3((unsigned)total_rows), dim3(TPB), 0, S(stream), params, grad_eff, items, n_items, pert_w, grad_w,
                       grad_b, row_norm);
    BD_LAUNCH_CHECK("anp_grad");
    return BD_OK;
}
