// Dense (batched) GEMMs on PRE-SPLIT operands, fed by LDS-DMA (gfx950) -- round 3: the attention block's twelve GEMMs.
//
//   C[m][n] = out_scale * (alpha * sum_k A[m][k] * B[n][k] + bias[n] + residual[m][n])  (+ C[m][n])
//
// Same arithmetic as igemm.hip / conv_ps.hip (x = hi + lo, lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate), but
// both operands arrive as split planes (per row, every 32-column block is one 128-byte line: 64 B of bf16 hi | 64 B of bf16 lo -- the
// bytes of the fp32 tensor they replace) and the result can LEAVE as split planes too, so a chain of products (QKV projection ->
// attention core -> output projection, and the backward chain, attention.py:85-186) passes planes from kernel to kernel and nobody converts
// an operand on the way into LDS.  Measured: per launch this kernel is at PARITY with the register-staged igemm on
// the K = 256 shapes (both ~8 us + flops at ~270 TFLOP/s: the L2 -> LDS fill bound of a 128 x 128 tile) -- it is in the plan because the
// attention core (attn_sp.hip) wants planes in and hands planes out, which is where the step time went down.
//
// Either operand may be K-CONTIGUOUS (rows = M or N, a K chunk of a row = one line: weights [N][K], activations [M][K]) or K-MAJOR
// (rows = K, the M / N index runs along the row: the "pixel-major" operands of a weight gradient, V in P V, K in dS K):
//   K-contiguous: LDS image [128 rows][128 B], 16-byte slots XOR-swizzled on the DMA source side, fragments by ds_read_b128;
//   K-major:      LDS image [32 k][512 B], slots XOR-permuted by (k & 3) << 2, fragments by ds_read_b64_tr_b16 (hardware transpose).
// One workgroup = 4 waves (2 x 2) on a 128 x 128 tile, 64 x 64 per wave (8 fragment reads per 12 MFMAs), two LDS stages of 32 KB:
// 64 KB and <= 256 registers per wave leave room for TWO workgroups per CU, which de-phase -- one runs its epilogue / prologue while
// the other is in its main loop; with 8 K chunks per tile that overlap is worth more than a deeper ring in one workgroup.
// Long-K products (the weight gradients: K = all tokens) split K over workgroups with a fixed-order second pass (deterministic).
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace bd {

typedef float sp_floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef short sp_short4 __attribute__((ext_vector_type(4)));
typedef short sp_short8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* sp_lds_ptr;
typedef const __attribute__((address_space(1))) void* sp_gbl_ptr;

constexpr int SP_T = 128, SP_NT = 256;
constexpr int SP_OP_BYTES = 128 * 128, SP_STAGE_BYTES = 2 * SP_OP_BYTES, SP_LDS_BYTES = 2 * SP_STAGE_BYTES;

struct SpParams {
    const char* a; const char* b;
    long long lda, ldb;            // bytes between operand rows
    long long a_bs, b_bs;          // bytes between batches
    float* c; long long ldc, c_bs;             // fp32 output (elements), may be null
    char* cs; long long ldcs, cs_bs;           // split-plane output (bytes), may be null
    const float* bias; const float* res; long long ldr, r_bs;
    float alpha, out_scale;
    int accumulate;
    int M, N, K, tiles_m, tiles_n, batch;
    float* partial;                // ksplit > 1: [ksplit][M][N] slabs, then [ksplit][M] column-sum rows
    int ksplit, cps;               // chunks (32 of K) per split
    float* colsum; int want_colsum;            // both operands K-major: colsum[m] = sum_k A[k][m] (a linear layer's bias gradient)
};

__device__ __forceinline__ int sp_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void sp_dma16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((sp_gbl_ptr)src, (sp_lds_ptr)lds_dst, 16, 0, 0);
}
// asm reads (address = LDS byte offset): opaque to hipcc's "LDS-DMA in flight -> s_waitcnt vmcnt(0) before any LDS read" pass
// (conv_ps.hip ps_tr_read); the matching s_waitcnt lgkmcnt(0) is tied to the fragment registers in sp_wait().
template <int OFF>
__device__ __forceinline__ sp_short4 sp_read_tr(unsigned addr) {
    sp_short4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ sp_short8 sp_read128(unsigned addr) {
    sp_short8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// fragments of one 16-wide K step of a wave's two 32-row tiles, planes hi / lo
template <bool KM> struct SpFrag;
template <> struct SpFrag<false> {
    sp_short8 v[2][2];
    __device__ __forceinline__ sp_bf16x8 get(int i, int pl) const { return __builtin_bit_cast(sp_bf16x8, v[i][pl]); }
};
template <> struct SpFrag<true> {
    sp_short4 v0[2][2], v1[2][2];
    __device__ __forceinline__ sp_bf16x8 get(int i, int pl) const {
        const sp_short8 v = __builtin_shufflevector(v0[i][pl], v1[i][pl], 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(sp_bf16x8, v);
    }
};
__device__ __forceinline__ void sp_tie(SpFrag<false>& f) {
    asm volatile("" : "+v"(f.v[0][0]), "+v"(f.v[0][1]), "+v"(f.v[1][0]), "+v"(f.v[1][1]));
}
__device__ __forceinline__ void sp_tie(SpFrag<true>& f) {
    asm volatile("" : "+v"(f.v0[0][0]), "+v"(f.v0[0][1]), "+v"(f.v0[1][0]), "+v"(f.v0[1][1]), "+v"(f.v1[0][0]), "+v"(f.v1[0][1]),
                 "+v"(f.v1[1][0]), "+v"(f.v1[1][1]));
}

// y -> (bf16 hi = RNE, bf16 lo = RNE of the remainder): the split of bd_split_rows (common.h), packed hi | lo << 16
__device__ __forceinline__ unsigned sp_split1(float v) { return bd_split1(v); }

template <bool AKM, bool BKM>
__global__ __launch_bounds__(SP_NT, 2) void gemm_sp_kernel(SpParams p) {
    __shared__ __attribute__((aligned(128))) char smem[SP_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    // logical order: n fastest, m, batch, K split; one contiguous run per XCD (tiles sharing an operand panel meet in one L2)
    int tm, tn, bz, zz;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        const unsigned ntiles = p.tiles_m * p.tiles_n;
        const unsigned per_split = ntiles * p.batch;
        zz = j / per_split; j -= zz * per_split;
        bz = j / ntiles; j -= bz * ntiles;
        tm = j / p.tiles_n;
        tn = j - tm * p.tiles_n;
    }
    const int m0 = tm * SP_T, n0 = tn * SP_T;
    const int nchunks = p.K >> 5;
    const int c_begin = zz * p.cps;
    int c_end = c_begin + p.cps;
    if (c_end > nchunks) c_end = nchunks;
    const int n = c_end - c_begin;

    // ---- DMA: 16 one-KiB instructions per operand and chunk, four per wave
    const char* asrc[4]; const char* bsrc[4];
    long long a_adv, b_adv;
    {
        const char* a = p.a + (long long)bz * p.a_bs;
        const char* b = p.b + (long long)bz * p.b_bs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = wave + 4 * j;
            if constexpr (AKM) {       // 2 k-rows per instruction: lane -> k = 2g + lane/32, slot lane%32
                const int k = 2 * g + (lane >> 5);
                asrc[j] = a + ((long long)c_begin * 32 + k) * p.lda + m0 * 4 + (((lane & 31) ^ ((k & 3) << 2)) << 4);
            } else {                   // 8 rows per instruction: lane -> row 8g + lane/8, physical slot lane%8
                const int r = g * 8 + (lane >> 3);
                asrc[j] = a + (long long)(m0 + r) * p.lda + (long long)c_begin * 128 + (((lane & 7) ^ sp_swz(r)) << 4);
            }
            if constexpr (BKM) {
                const int k = 2 * g + (lane >> 5);
                bsrc[j] = b + ((long long)c_begin * 32 + k) * p.ldb + n0 * 4 + (((lane & 31) ^ ((k & 3) << 2)) << 4);
            } else {
                const int r = g * 8 + (lane >> 3);
                bsrc[j] = b + (long long)(n0 + r) * p.ldb + (long long)c_begin * 128 + (((lane & 7) ^ sp_swz(r)) << 4);
            }
        }
        a_adv = AKM ? 32 * p.lda : 128;
        b_adv = BKM ? 32 * p.ldb : 128;
    }
    auto issue = [&](char* stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sp_dma16(asrc[j], stage + (wave + 4 * j) * 1024); asrc[j] += a_adv; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { sp_dma16(bsrc[j], stage + SP_OP_BYTES + (wave + 4 * j) * 1024); bsrc[j] += b_adv; }
    };

    // ---- fragment addresses (stage 0; the stage, the 16-wide K step and the tile within the wave are immediates)
    const unsigned smem_addr = (unsigned)(uintptr_t)(sp_lds_ptr)smem;
    unsigned aoff[2][2], boff[2][2];   // K-contiguous: [step][plane]; K-major: [tile][plane]
    {
        // K-major (conv_ps.hip wgrad / igemm.hip rc_frag): lane reads 4 rows x 1 k (8 bytes)
        const int sl = lane & 15, hb = (lane >> 4) & 1;
        const int kq = sl >> 2, rq = sl & 3;
        const int lane_base = (8 * h + kq) * 512 + hb * 32 + rq * 8;
        auto km = [&](int t, int pl) { return (unsigned)(lane_base + ((((t & 1) * 2 + pl) ^ kq) << 6) + (t >> 1) * 256); };
        auto kc = [&](int s, int pl) { return (unsigned)(li * 128 + (((pl * 4 + s * 2 + h) ^ sp_swz(li)) << 4)); };
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                aoff[x][pl] = smem_addr + (AKM ? km(wm * 2 + x, pl) : kc(x, pl) + wm * 64 * 128);
                boff[x][pl] = smem_addr + SP_OP_BYTES + (BKM ? km(wn * 2 + x, pl) : kc(x, pl) + wn * 64 * 128);
            }
    }

    sp_floatx16 acc[2][2], accs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accs[i][r] = 0.f; }
    const bool do_sum = AKM && BKM && p.want_colsum && wn == 0 && tn == 0;   // wave-uniform
    sp_bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    auto reads = [&](auto ST, auto S, SpFrag<AKM>& fa, SpFrag<BKM>& fb) {
        constexpr int stage = decltype(ST)::value, s = decltype(S)::value;
        constexpr int SB = stage * SP_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if constexpr (AKM) {
                    fa.v0[i][pl] = sp_read_tr<SB + s * 16 * 512>(aoff[i][pl]);
                    fa.v1[i][pl] = sp_read_tr<SB + s * 16 * 512 + 4 * 512>(aoff[i][pl]);
                } else {
                    fa.v[i][pl] = i == 0 ? sp_read128<SB>(aoff[s][pl]) : sp_read128<SB + 4096>(aoff[s][pl]);
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if constexpr (BKM) {
                    fb.v0[i][pl] = sp_read_tr<SB + s * 16 * 512>(boff[i][pl]);
                    fb.v1[i][pl] = sp_read_tr<SB + s * 16 * 512 + 4 * 512>(boff[i][pl]);
                } else {
                    fb.v[i][pl] = i == 0 ? sp_read128<SB>(boff[s][pl]) : sp_read128<SB + 4096>(boff[s][pl]);
                }
            }
    };
    auto wait = [&](SpFrag<AKM>& fa, SpFrag<BKM>& fb) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sp_tie(fa); sp_tie(fb);
    };
    auto mfmas = [&](const SpFrag<AKM>& fa, const SpFrag<BKM>& fb) {
        sp_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = fa.get(i, 0); al[i] = fa.get(i, 1); bh[i] = fb.get(i, 0); bl[i] = fb.get(i, 1); }
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
        if constexpr (AKM && BKM) {
            if (do_sum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], ones, accs[i], 0, 0, 0);
                    accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], ones, accs[i], 0, 0, 0);
                }
            }
        }
    };
    // one chunk: the second step's fragments are read while the first step's MFMAs run
    auto chunk = [&](auto ST) {
        SpFrag<AKM> a0, a1; SpFrag<BKM> b0, b1;
        reads(ST, std::integral_constant<int, 0>{}, a0, b0);
        wait(a0, b0);
        reads(ST, std::integral_constant<int, 1>{}, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        wait(a1, b1);
        mfmas(a1, b1);
    };

    // two stages: chunk c+1 is fetched while chunk c is multiplied; the second resident workgroup covers the rest of the latency
    if (n > 0) issue(smem);
    for (int c = 0; c < n; c += 2) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 1 < n) issue(smem + SP_STAGE_BYTES);
        chunk(std::integral_constant<int, 0>{});
        if (c + 1 < n) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + 2 < n) issue(smem);
            chunk(std::integral_constant<int, 1>{});
        }
    }

    // ---- epilogue: lane holds column n = li of rows (r&3) + 8*(r>>2) + 4*h of every 32 x 32 tile (whole tiles: M, N % 128 == 0)
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
    if (p.ksplit > 1) {
        float* out = p.partial + (long long)zz * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(long long)(mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * p.N + nw + q * 32 + li] = acc[i][q][r];
        if constexpr (AKM && BKM) {
            if (do_sum && li == 0) {
                float* o = p.partial + (long long)p.ksplit * p.M * p.N + (long long)zz * p.M;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = accs[i][r];
            }
        }
        return;
    }
    if constexpr (AKM && BKM) {
        if (do_sum && li == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) p.colsum[mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = accs[i][r];
        }
    }
    float* c = p.c ? p.c + (long long)bz * p.c_bs : nullptr;
    char* cs = p.cs ? p.cs + (long long)bz * p.cs_bs : nullptr;
    const float* res = p.res ? p.res + (long long)bz * p.r_bs : nullptr;
    const bool odd = li & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nn = nw + q * 32 + li;
            const float bn = p.bias ? p.bias[nn] : 0.f;
            const int m_base = mw + i * 32 + 4 * h;
            float rs[16], pc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m_base + (r & 3) + 8 * (r >> 2);
                rs[r] = res ? res[m * p.ldr + nn] : 0.f;
                pc[r] = p.accumulate ? c[m * p.ldc + nn] : 0.f;
            }
            // split-plane output: lane pairs trade halves so that every lane stores one dword (even lanes the hi pair, odd lanes the lo pair)
            const long long cs_col = (long long)(nn >> 5) * 128 + (odd ? 64 + (li - 1) * 2 : li * 2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m_base + (r & 3) + 8 * (r >> 2);
                float v = (p.alpha * acc[i][q][r] + bn + rs[r]) * p.out_scale + pc[r];
                if (c) c[m * p.ldc + nn] = v;
                if (cs) {
                    const unsigned w = sp_split1(v);
                    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
                    const unsigned o = odd ? ((nb >> 16) | (w & 0xFFFF0000u)) : ((w & 0xFFFFu) | (nb << 16));
                    *reinterpret_cast<unsigned*>(cs + m * p.ldcs + cs_col) = o;
                }
            }
        }
}

// second pass of the K split: fixed-order sum of the slabs + the epilogue; 8 columns per thread (N % 8 == 0)
__global__ __launch_bounds__(256) void gemm_sp_reduce(SpParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n8 = p.N >> 3;
    if (i >= (long long)p.M * n8) {
        // tail threads of the last workgroups fold the column sums
        return;
    }
    const int m = (int)(i / n8), nn = (int)(i - (long long)m * n8) * 8;
    const long long mn = (long long)p.M * p.N, e = (long long)m * p.N + nn;
    float v[8];
    {
        const float4 a = *reinterpret_cast<const float4*>(p.partial + e);
        const float4 b = *reinterpret_cast<const float4*>(p.partial + e + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll 4
    for (int s = 1; s < p.ksplit; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(p.partial + (long long)s * mn + e);
        const float4 b = *reinterpret_cast<const float4*>(p.partial + (long long)s * mn + e + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = p.alpha * v[j];
        if (p.bias) t += p.bias[nn + j];
        if (p.res) t += p.res[(long long)m * p.ldr + nn + j];
        t *= p.out_scale;
        if (p.accumulate) t += p.c[(long long)m * p.ldc + nn + j];
        v[j] = t;
    }
    if (p.c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p.c[(long long)m * p.ldc + nn + j] = v[j];
    }
    if (p.cs) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w0 = sp_split1(v[2 * j]), w1 = sp_split1(v[2 * j + 1]);
            hi[j] = (w0 & 0xFFFFu) | (w1 << 16);
            lo[j] = (w0 >> 16) | (w1 & 0xFFFF0000u);
        }
        char* o = p.cs + (long long)m * p.ldcs + (nn >> 5) * 128 + (nn & 31) * 2;
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 64) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}
// column-sum rows of the K split: one thread per column, eight slab loads in flight (a serial chain of `ksplit` dependent-latency loads took 11 us for 64 slabs), summed in slab order
__global__ __launch_bounds__(64) void gemm_sp_colsum_reduce(const float* __restrict__ rows, int ksplit, int M, float* __restrict__ out) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= ksplit; s += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = rows[(long long)(s + j) * M + m];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    for (; s < ksplit; ++s) v += rows[(long long)s * M + m];
    out[m] = v;
}

// ---- host ------------------------------------------------------------------------------------------------------------
static int sp_cus() {
    static int cus = [] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    return cus;
}
static void sp_split(const bd_gemm_sp_desc& d, int& ksplit, int& cps) {
    const int nch = d.K / 32;
    const long long tiles = (long long)(d.M / SP_T) * (d.N / SP_T) * (d.batch > 0 ? d.batch : 1);
    static const int forced = getenv("BD_SP_KSPLIT") ? atoi(getenv("BD_SP_KSPLIT")) : 0;
    ksplit = 1;
    static const int slots = getenv("BD_SP_SLOTS") ? atoi(getenv("BD_SP_SLOTS")) : sp_cus();   // (A/B knob)
    if (d.batch <= 1 && tiles < sp_cus() && nch >= 32) {   // one workgroup per CU: the slabs are written and read once more each
        ksplit = (int)cdiv(slots, tiles);
        if (ksplit > nch / 8) ksplit = nch / 8;
    }
    if (forced > 0 && d.batch <= 1) ksplit = forced < nch ? forced : nch;
    if (ksplit < 1) ksplit = 1;
    cps = (int)cdiv(nch, ksplit);
    ksplit = (int)cdiv(nch, cps);
}

bool gemm_sp_supported(int M, int N, int K) {
    static const bool off = getenv("BD_GEMM_SP") && atoi(getenv("BD_GEMM_SP")) == 0;
    return !off && M > 0 && N > 0 && K > 0 && M % SP_T == 0 && N % SP_T == 0 && K % 32 == 0;
}

size_t gemm_sp_workspace_bytes(const bd_gemm_sp_desc& d) {
    if (!gemm_sp_supported(d.M, d.N, d.K)) return 0;
    int ksplit, cps;
    sp_split(d, ksplit, cps);
    return ksplit > 1 ? (size_t)ksplit * ((size_t)d.M * d.N + d.M) * sizeof(float) : 0;
}

int gemm_sp(const bd_gemm_sp_desc& d, hipStream_t st) {
    BD_CHECK(d.a && d.b && (d.c || d.c_split), BD_ERR_INVALID, "bd_gemm_sp: null pointer");
    BD_CHECK(d.M > 0 && d.N > 0 && d.K > 0 && d.M % SP_T == 0 && d.N % SP_T == 0 && d.K % 32 == 0, BD_ERR_UNSUPPORTED,
             "bd_gemm_sp: needs M, N %% 128 == 0 and K %% 32 == 0 (M=%d N=%d K=%d)", d.M, d.N, d.K);
    const int batch = d.batch > 0 ? d.batch : 1;
    BD_CHECK(((uintptr_t)d.a & 127) == 0 && ((uintptr_t)d.b & 127) == 0 && d.lda % 32 == 0 && d.ldb % 32 == 0 && d.a_bs % 32 == 0 &&
                 d.b_bs % 32 == 0,
             BD_ERR_UNSUPPORTED, "bd_gemm_sp: operand planes must be 128-byte aligned with row / batch strides multiples of 32");
    BD_CHECK(d.lda >= (d.a_kmajor ? d.M : d.K) && d.ldb >= (d.b_kmajor ? d.N : d.K), BD_ERR_INVALID, "bd_gemm_sp: row stride below the row length");
    BD_CHECK(!d.c_split || (((uintptr_t)d.c_split & 127) == 0 && d.ldcs % 32 == 0 && d.cs_bs % 32 == 0 && d.ldcs >= d.N), BD_ERR_UNSUPPORTED,
             "bd_gemm_sp: split output must be 128-byte aligned with strides multiples of 32");
    BD_CHECK(!d.c || d.ldc >= d.N, BD_ERR_INVALID, "bd_gemm_sp: ldc < N");
    BD_CHECK(!d.accumulate || d.c, BD_ERR_INVALID, "bd_gemm_sp: accumulate needs the fp32 output");
    BD_CHECK(!d.residual || d.ldr >= d.N, BD_ERR_INVALID, "bd_gemm_sp: ldr < N");
    BD_CHECK(!d.a_colsum || (d.a_kmajor && d.b_kmajor && batch == 1), BD_ERR_UNSUPPORTED,
             "bd_gemm_sp: a_colsum needs both operands K-major and no batch");
    SpParams p = {};
    p.a = reinterpret_cast<const char*>(d.a); p.b = reinterpret_cast<const char*>(d.b);
    p.lda = d.lda * 4; p.ldb = d.ldb * 4; p.a_bs = d.a_bs * 4; p.b_bs = d.b_bs * 4;
    p.c = d.c; p.ldc = d.ldc; p.c_bs = d.c_bs;
    p.cs = reinterpret_cast<char*>(d.c_split); p.ldcs = d.ldcs * 4; p.cs_bs = d.cs_bs * 4;
    p.bias = d.bias; p.res = d.residual; p.ldr = d.ldr; p.r_bs = d.r_bs;
    p.alpha = d.alpha; p.out_scale = d.out_scale; p.accumulate = d.accumulate;
    p.M = d.M; p.N = d.N; p.K = d.K; p.tiles_m = d.M / SP_T; p.tiles_n = d.N / SP_T; p.batch = batch;
    p.colsum = d.a_colsum; p.want_colsum = d.a_colsum != nullptr;
    sp_split(d, p.ksplit, p.cps);
    if (p.ksplit > 1) {
        const size_t need = (size_t)p.ksplit * ((size_t)d.M * d.N + d.M) * sizeof(float);
        BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "bd_gemm_sp: workspace %zu < %zu", d.workspace_bytes, need);
        p.partial = reinterpret_cast<float*>(d.workspace);
    }
    const long long grid = (long long)p.tiles_m * p.tiles_n * batch * p.ksplit;
    BD_CHECK(grid < (1ll << 31), BD_ERR_UNSUPPORTED, "bd_gemm_sp: grid too large");
    const dim3 g((unsigned)grid), b(SP_NT);
    int rec = -1;
    if (prof_on()) {   // bench.py roofline: algorithmic flops, unique operand + output bytes (planes are 4 B per element like fp32)
        const double nb = batch;
        const double outs = (d.c ? 1.0 : 0.0) + (d.c_split ? 1.0 : 0.0) + (d.residual ? 1.0 : 0.0) + (d.accumulate ? 1.0 : 0.0);
        rec = prof_begin(d.a_kmajor && d.b_kmajor ? "gemm_sp_tn" : (d.b_kmajor ? "gemm_sp_nn" : (d.a_kmajor ? "gemm_sp_tn_a" : "gemm_sp_nt")),
                         2.0 * d.M * d.N * (double)d.K * nb, 4.0 * nb * ((double)d.M * d.K + (double)d.N * d.K + outs * d.M * d.N), st);
    }
    if (d.a_kmajor && d.b_kmajor) hipLaunchKernelGGL((gemm_sp_kernel<true, true>), g, b, 0, st, p);
    else if (d.a_kmajor) hipLaunchKernelGGL((gemm_sp_kernel<true, false>), g, b, 0, st, p);
    else if (d.b_kmajor) hipLaunchKernelGGL((gemm_sp_kernel<false, true>), g, b, 0, st, p);
    else hipLaunchKernelGGL((gemm_sp_kernel<false, false>), g, b, 0, st, p);
    BD_LAUNCH_CHECK("gemm_sp");
    if (p.ksplit > 1) {
        const long long work = (long long)d.M * (d.N / 8);
        hipLaunchKernelGGL(gemm_sp_reduce, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, st, p);
        if (p.want_colsum)
            hipLaunchKernelGGL(gemm_sp_colsum_reduce, dim3((unsigned)cdiv(d.M, 64)), dim3(64), 0, st,
                               p.partial + (long long)p.ksplit * d.M * d.N, p.ksplit, d.M, d.a_colsum);
        BD_LAUNCH_CHECK("gemm_sp_reduce");
    }
    prof_end(rec, st);
    return BD_OK;
}

}  // namespace bd

extern "C" size_t bd_gemm_sp_workspace_bytes(const bd_gemm_sp_desc* d) { return d ? bd::gemm_sp_workspace_bytes(*d) : 0; }
extern "C" int bd_gemm_sp(const bd_gemm_sp_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_gemm_sp: null descriptor");
    return bd::gemm_sp(*d, bd::S(s));
}
