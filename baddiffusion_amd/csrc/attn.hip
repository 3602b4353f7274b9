// Fused attention forward for the UNet's AttentionBlock (attention.py:148-162): per (sample, head, block of 64 queries)
//     S^T = K Q^T * scale  ->  row softmax over the keys  ->  O = P V
// in ONE kernel: the [N, N] score / probability matrices never go to HBM (the unfused path writes S, re-reads it for the
// softmax, writes P and re-reads it for P V: 4 x 33.5 MB per block at B = 128, N = 256).  Same arithmetic as the split-bf16
// igemm engine (x = hi + lo bf16, products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation) and
// an fp32 softmax with the row maximum subtracted, like torch.softmax on S.float().
//
// One workgroup = 8 waves, 64 queries, all N <= 256 keys:
//   phase 1  S^T tile (wave w: keys [32w, 32w+32) x 64 queries) = K Q^T, contraction over d in chunks of 32.  The product
//            is taken "swapped" (A = K rows, B = Q rows) so that a lane holds ONE query column (lane & 31) and 16 keys
//            of it per accumulator tile: the row statistics of the softmax are in-lane reductions + one lane^32 exchange
//            + one 8-way exchange through LDS (cdna_hip_programming.md T12).
//   phase 2  p = exp(s - max) / sum in registers; 4 consecutive keys of a query = 8 bytes of bf16 hi and 8 of lo, written
//            to the LDS image of P ([64 queries][N keys], the k-contiguous layout of conv_ps.hip's operand tiles);
//            optionally P (fp32) and the row log-sum-exp go to HBM for the backward pass.
//   phase 3  O tile (wave w: d columns [32w, 32w+32) (+256), 64 queries) = P V over the keys in chunks of 32; V rows are
//            key-major, so its fragments come from the transpose read ds_read_b64_tr_b16 (conv_ps_wgrad's layout).
// Operands are fp32 in HBM (the QKV GEMM's output); they are split on the way into LDS (global -> registers -> split ->
// LDS, next chunk prefetched into registers while the current one is multiplied).
#include "common.h"

namespace bd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short at_short4 __attribute__((ext_vector_type(4)));
typedef short at_short8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) at_short4 at_lds_short4;

struct AttnParams {
    const float* q; const float* k; const float* v;   // [B, N, ld] each (the three column blocks of the QKV buffer)
    long long ld;
    float* o; long long ldo;                          // [B, N, ldo]
    float* p_out;                                     // optional [B*heads, N, N] fp32 probabilities (training)
    float* lse;                                       // optional [B*heads, N]: max + log(sum)
    int N, dh, heads;
    float scale;
};

__device__ __forceinline__ unsigned at_pack_hi(float a, float b) {
    return (__builtin_bit_cast(unsigned, a) >> 16) | (__builtin_bit_cast(unsigned, b) & 0xFFFF0000u);
}
__device__ __forceinline__ unsigned at_pack_lo(float a, float b) {
    const float ra = a - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFF0000u);
    const float rb = b - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFF0000u);
    bf16x2 t;
    t[0] = (__bf16)ra; t[1] = (__bf16)rb;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ int at_swz(int row) { return (row >> 1) & 7; }

// k-contiguous tile image: row r = 128 bytes = 8 slots of 16 B (slots 0-3: 32 bf16 hi, 4-7: lo), slot ^= swz(r).
// Store the 4 values x[0..3] of columns c4..c4+3 (c4 % 4 == 0, < 32) of row r.
__device__ __forceinline__ void at_store_kc(char* tile, int r, int c4, const float4& x) {
    const int slot = c4 >> 3, in = (c4 & 7) * 2;
    const int sw = at_swz(r);
    *reinterpret_cast<uint2*>(tile + r * 128 + ((slot ^ sw) << 4) + in) = make_uint2(at_pack_hi(x.x, x.y), at_pack_hi(x.z, x.w));
    *reinterpret_cast<uint2*>(tile + r * 128 + (((slot + 4) ^ sw) << 4) + in) = make_uint2(at_pack_lo(x.x, x.y), at_pack_lo(x.z, x.w));
}
// fragment of a k-contiguous tile: lane -> row (tile_row0 + lane&31), k octet (lane>>5) of K16 step s, plane pl
__device__ __forceinline__ bf16x8 at_frag_kc(const char* tile, int row, int s, int h, int pl) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((pl * 4 + s * 2 + h) ^ at_swz(row)) << 4));
}

// key-major ("row contiguous") V image: key k = ROWB bytes = dh/32 blocks of 128 B (hi 64 | lo 64); the 16-byte slots of a
// key row are XOR-ed with (k & 3) << 2 so that the 4 keys a transpose read touches hit 4 different 64-byte bank windows.
__device__ __forceinline__ void at_store_rc(char* tile, int rowb, int k, int d4, const float4& x) {
    const int slot = (d4 >> 5) * 8 + ((d4 & 31) >> 3), in = (d4 & 7) * 2;
    const int sx = (k & 3) << 2;
    *reinterpret_cast<uint2*>(tile + k * rowb + ((slot ^ sx) << 4) + in) = make_uint2(at_pack_hi(x.x, x.y), at_pack_hi(x.z, x.w));
    *reinterpret_cast<uint2*>(tile + k * rowb + (((slot + 4) ^ sx) << 4) + in) = make_uint2(at_pack_lo(x.x, x.y), at_pack_lo(x.z, x.w));
}

template <int TD>   // TD = d tiles per wave in phase 3: 1 (dh <= 256) or 2 (dh <= 512)
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnParams p) {
    constexpr int REG_A = 65536;                 // phase-1 stage (Q 8 KB + K up to 32 KB) / phase-3 V stage (32 keys x dh x 4 B <= 64 KB)
    constexpr int REG_P = 65536;                 // P image: 64 queries x N keys x 4 B
    __shared__ __attribute__((aligned(128))) char smem[REG_A + REG_P + 4096];
    char* const sA = smem;
    char* const sP = smem + REG_A;
    float* const red = reinterpret_cast<float*>(smem + REG_A + REG_P);   // [2][8 waves][64 queries]... reused for max then sum

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int N = p.N, dh = p.dh, nkt = N >> 5;          // key tiles (<= 8: one per wave)
    const int nqb = N >> 6;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int b = bh / p.heads, hd = bh - b * p.heads;
    const long long base = (long long)b * N * p.ld + (long long)hd * dh;
    const float* Q = p.q + base + (long long)qb * 64 * p.ld;
    const float* K = p.k + base;
    const float* V = p.v + base;

    // ---------------------------------------------------------------- phase 1: S^T = K Q^T
    char* const sQ = sA;                 // 64 rows x 128 B
    char* const sK = sA + 8192;          // N rows x 128 B
    const int lr = tid >> 3, lc4 = (tid & 7) * 4;        // loader map: 64 rows x 8 float4 per pass
    // The problem is small (N x dh fp32 per operand) and latency bound: ALL loads of a group of 4 d-chunks (128 columns) are
    // issued at once -- 4 float4 of Q and 16 of K per thread in flight -- and consumed chunk by chunk as they return
    // (loads return in order; the compiler counts vmcnt).  One chunk of prefetch made every chunk pay a full L2 round trip.
    constexpr int GRP = 4;
    float4 rq[GRP], rk[GRP][4];
    floatx16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    const bool has_keys = wave < nkt;
    const int nch = dh >> 5;
    int krow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) krow[i] = lr + 64 * i < N ? lr + 64 * i : N - 1;
    for (int g0 = 0; g0 < nch; g0 += GRP) {
#pragma unroll
        for (int c = 0; c < GRP; ++c) {
            if (g0 + c < nch) {
                const int c0 = (g0 + c) * 32;
                rq[c] = *reinterpret_cast<const float4*>(Q + (long long)lr * p.ld + c0 + lc4);
                // branch-free: rows past N re-read row N-1 (a predicated load would make hipcc wait vmcnt(0) behind every
                // single load, guide section 5 trap (c)); their LDS rows are never read (waves >= nkt hold no keys)
#pragma unroll
                for (int i = 0; i < 4; ++i) rk[c][i] = *reinterpret_cast<const float4*>(K + (long long)krow[i] * p.ld + c0 + lc4);
            }
        }
#pragma unroll
        for (int c = 0; c < GRP; ++c) {
            if (g0 + c < nch) {
                at_store_kc(sQ, lr, lc4, rq[c]);
#pragma unroll
                for (int i = 0; i < 4; ++i) at_store_kc(sK, lr + 64 * i, lc4, rk[c][i]);
                __syncthreads();
                if (has_keys) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 kh = at_frag_kc(sK, wave * 32 + li, s, h, 0), kl = at_frag_kc(sK, wave * 32 + li, s, h, 1);
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt) {
                            const bf16x8 qh = at_frag_kc(sQ, qt * 32 + li, s, h, 0), ql = at_frag_kc(sQ, qt * 32 + li, s, h, 1);
                            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, acc[qt], 0, 0, 0);
                            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, acc[qt], 0, 0, 0);
                            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, acc[qt], 0, 0, 0);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }

    // ---------------------------------------------------------------- phase 2: softmax over the keys of every query column
    // lane: query q = qt*32 + li, keys 32*wave + (r&3) + 8*(r>>2) + 4*h
    float mx[2], sm[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = -INFINITY;
        if (has_keys) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[qt][r] *= p.scale; m = fmaxf(m, acc[qt][r]); }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (h == 0) red[wave * 64 + qt * 32 + li] = m;
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = red[qt * 32 + li];
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w * 64 + qt * 32 + li]);
        mx[qt] = m;
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float s = 0.f;
        if (has_keys) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[qt][r] = expf(acc[qt][r] - mx[qt]); s += acc[qt][r]; }
        }
        s += __shfl_xor(s, 32, 64);
        if (h == 0) red[wave * 64 + qt * 32 + li] = s;
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float s = red[qt * 32 + li];
        for (int w = 1; w < 8; ++w) s += red[w * 64 + qt * 32 + li];
        sm[qt] = s;
    }
    if (has_keys) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float inv = 1.0f / sm[qt];
            const int q = qt * 32 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // 4 consecutive keys: local key index 8*g + 4*h .. +3 of the wave's 32
                const float4 pv = make_float4(acc[qt][4 * g] * inv, acc[qt][4 * g + 1] * inv, acc[qt][4 * g + 2] * inv, acc[qt][4 * g + 3] * inv);
                // P image row q, key chunk `wave` (32 keys = one 128-byte line), columns 8g + 4h
                at_store_kc(sP + wave * 8192, q, 8 * g + 4 * h, pv);
                if (p.p_out)
                    *reinterpret_cast<float4*>(p.p_out + ((long long)bh * N + qb * 64 + q) * N + wave * 32 + 8 * g + 4 * h) = pv;
            }
        }
    }
    if (p.lse && wave == 0 && h == 0) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) p.lse[(long long)bh * N + qb * 64 + qt * 32 + li] = mx[qt] + logf(sm[qt]);
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 3: O = P V over key chunks of 32
    const int rowb = dh * 4;                              // bytes of one key row of the V image
    const int nd4 = dh >> 2;                              // float4 per key row
    constexpr int VPT = 4 * TD;                           // float4 of one 32-key chunk per thread (dh <= 256 * TD)
    constexpr int VGRP = 4 / TD;                          // chunks whose loads are issued together (16 float4 per thread)
    float4 rv[VGRP][VPT];
    int vk[VPT], vd[VPT];                                 // this thread's (key, first column) pieces of a chunk; surplus pieces
#pragma unroll                                            // (dh < 256 * TD) repeat the last one: branch-free loads, idempotent stores
    for (int i = 0; i < VPT; ++i) {
        int e = tid + 512 * i;
        if (e >= 32 * nd4) e = 32 * nd4 - 1;
        vk[i] = e / nd4;
        vd[i] = (e - vk[i] * nd4) * 4;
    }
    floatx16 oacc[TD][2];
#pragma unroll
    for (int t = 0; t < TD; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[t][0][r] = 0.f; oacc[t][1][r] = 0.f; }
    const int ndt = dh >> 5;                               // d tiles
    // transpose-read lane map (igemm.hip rc_frag / conv_ps_wgrad): lane reads 4 d-rows x 1 key (8 bytes)
    const int sl = lane & 15, hb = (lane >> 4) & 1, kq = sl >> 2, rq4 = sl & 3;
    for (int g0 = 0; g0 < nkt; g0 += VGRP) {
#pragma unroll
        for (int c = 0; c < VGRP; ++c) {
            if (g0 + c < nkt) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) rv[c][i] = *reinterpret_cast<const float4*>(V + (long long)((g0 + c) * 32 + vk[i]) * p.ld + vd[i]);
            }
        }
#pragma unroll
        for (int c = 0; c < VGRP; ++c) {
            if (g0 + c < nkt) {
                const int kc = g0 + c;
#pragma unroll
                for (int i = 0; i < VPT; ++i) at_store_rc(sA, rowb, vk[i], vd[i], rv[c][i]);
                __syncthreads();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 ph[2], pl[2];
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        ph[qt] = at_frag_kc(sP + kc * 8192, qt * 32 + li, s, h, 0);
                        pl[qt] = at_frag_kc(sP + kc * 8192, qt * 32 + li, s, h, 1);
                    }
#pragma unroll
                    for (int t = 0; t < TD; ++t) {
                        const int dt = wave + 8 * t;
                        if (dt < ndt) {
                            // V^T fragment: rows = d (32 of tile dt), k = keys 16s + 8h + 0..7 of this chunk
                            bf16x8 vf[2];
#pragma unroll
                            for (int pln = 0; pln < 2; ++pln) {
                                const int key0 = 16 * s + 8 * h + kq;          // this lane's key of the first read; +4 for the second
                                const int slot = dt * 8 + pln * 4 + hb * 2 + (rq4 >> 1);
                                const int o0 = key0 * rowb + ((slot ^ ((key0 & 3) << 2)) << 4) + (rq4 & 1) * 8;
                                const int o1 = (key0 + 4) * rowb + ((slot ^ (((key0 + 4) & 3) << 2)) << 4) + (rq4 & 1) * 8;
                                const at_short4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_short4*)(sA + o0));
                                const at_short4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_short4*)(sA + o1));
                                const at_short8 vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                                vf[pln] = __builtin_bit_cast(bf16x8, vv);
                            }
#pragma unroll
                            for (int qt = 0; qt < 2; ++qt) {
                                oacc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[qt], vf[0], oacc[t][qt], 0, 0, 0);
                                oacc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[qt], vf[1], oacc[t][qt], 0, 0, 0);
                                oacc[t][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[qt], vf[0], oacc[t][qt], 0, 0, 0);
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    // epilogue: oacc[t][qt]: rows = queries (r&3) + 8*(r>>2) + 4*h of q tile qt, column d = dt*32 + li
    float* O = p.o + (long long)b * N * p.ldo + (long long)hd * dh + (long long)qb * 64 * p.ldo;
#pragma unroll
    for (int t = 0; t < TD; ++t) {
        const int dt = wave + 8 * t;
        if (dt < ndt) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    O[(long long)q * p.ldo + dt * 32 + li] = oacc[t][qt][r];
                }
        }
    }
}

// N: whole 64-query blocks, at most 8 key tiles (one per wave); head dim: whole pairs of 32-column tiles (the V image's slot
// swizzle permutes inside aligned groups of 16 slots = 64 columns), at most 2 tiles per wave
bool attn_fwd_supported(int N, int dh) { return N >= 64 && N <= 256 && N % 64 == 0 && dh % 64 == 0 && dh >= 64 && dh <= 512; }

int attn_fwd(const bd_attn_fwd_desc& d, hipStream_t st) {
    BD_CHECK(d.q && d.k && d.v && d.o, BD_ERR_INVALID, "attn_fwd: null pointer");
    BD_CHECK(d.B > 0 && d.heads > 0 && attn_fwd_supported(d.N, d.dh), BD_ERR_UNSUPPORTED,
             "attn_fwd: needs 64 <= N <= 256, N %% 64 == 0, head dim %% 64 == 0 and <= 512 (got N %d, dh %d)", d.N, d.dh);
    BD_CHECK((d.ld & 3) == 0 && (d.ldo & 3) == 0 && aligned16(d.q) && aligned16(d.k) && aligned16(d.v) && aligned16(d.o) && aligned16(d.p_out),
             BD_ERR_UNSUPPORTED, "attn_fwd: pointers must be 16-byte aligned, leading dimensions multiples of 4");
    AttnParams p = {};
    p.q = d.q; p.k = d.k; p.v = d.v; p.ld = d.ld; p.o = d.o; p.ldo = d.ldo; p.p_out = d.p_out; p.lse = d.lse;
    p.N = d.N; p.dh = d.dh; p.heads = d.heads; p.scale = d.scale;
    const dim3 grid((unsigned)(d.B * d.heads * (d.N / 64)));
    int rec = -1;
    if (prof_on()) rec = prof_begin("attn_fwd", 4.0 * d.B * d.heads * (double)d.N * d.N * d.dh, (4.0 * d.B * d.heads * (double)d.N * d.dh) * 4.0, st);
    if (d.dh <= 256) hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(512), 0, st, p);
    BD_LAUNCH_CHECK("attn_fwd");
    prof_end(rec, st);
    return BD_OK;
}

}  // namespace bd

extern "C" int bd_attn_fwd(const bd_attn_fwd_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_attn_fwd: null descriptor");
    return bd::attn_fwd(*d, bd::S(s));
}
