// Attention core on split planes (gfx950) -- round 3: scores, softmax and the value product of AttentionBlock (attention.py:148-162:
// baddbmm(q, k^T) * scale -> softmax -> bmm(probs, v)) and their backward in THREE kernels that never write an [N, N] fp32 matrix:
//
//   forward    per (sample, head, 128 queries):  S^T = K Q^T * scale -> softmax over the keys -> O = P V        [+ P^T planes for training]
//   backward A per (sample, head, 128 queries):  dP^T = V dO^T -> dS = scale * P o (dP - rowsum(P o dP)) -> dQ = dS K   [+ dS^T planes]
//   backward B per (sample, head, 128 keys):     dV = P^T dO,  dK = dS^T Q
//
// replacing, per attention block, Q K^T + softmax + P V (3 launches, 71 us at B = 128) and dP + dV + softmax backward + dQ + dK (5
// launches, 118 us).  Operands are split planes (bd_split_rows layout, DMA'd global -> LDS without touching registers) and so are
// the results; same arithmetic as the GEMM engines (x = hi + lo, lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate,
// fp32 softmax with the row maximum subtracted).
//
// One workgroup = 4 waves; a wave OWNS 32 rows (queries, or keys in backward B) and all 256 columns of the [N, N] matrix, so the row
// statistics of the softmax (and of its backward) are in-lane reductions plus one lane ^ 32 exchange: the waves only share the LDS
// images of the streamed operand, one barrier per chunk, no cross-wave reduction anywhere.
//   phase S  acc[kt] (8 key tiles x 32 own rows, 128 registers) = streamed[256 rows][dh] . own[32 rows][dh]^T, contraction over dh in
//            chunks of 32: the streamed chunk is a K-contiguous image [256][128 B] (32 KB, ring of three), the own rows a private 4 KB.
//            The product is taken "swapped" (A = streamed rows, B = own rows): a lane then holds ONE own row (lane & 31) and 16 streamed
//            rows of it per tile -- exactly the k-slots an MFMA A operand of the next phase wants, up to a permutation of the contraction
//            index that the transpose reads of the other operand simply follow.  The [N, N] matrix never leaves the registers.
//   phase O  out[own 32 rows][dh] = M[own rows][256] . streamed[256][dh]: M as 128 registers of packed bf16 hi / lo A operands, the
//            streamed operand in K-major units [32 rows][128 columns] (16 KB, ring of eight, four units in flight), fragments by
//            ds_read_b64_tr_b16; the output's two 128-column halves are two sweeps over the 256 rows.
// LDS 144 KB, registers beyond 256 (accumulators in AGPRs): one workgroup per CU, 256 workgroups at B = 128.
// N = 256 (16 x 16 feature maps), head dim 256; other shapes keep the unfused path.
// Measured: 43 / 48 / 49 us at B = 128 against 71 + 118 for the unfused launches; each kernel moves exactly its
// algorithmic bytes behind the L2 at 3.8-4.1 TB/s, and that -- not the CU side -- is what it waits for.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace bd {

typedef float as_floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 as_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 as_bf16x2 __attribute__((ext_vector_type(2)));
typedef short as_short4 __attribute__((ext_vector_type(4)));
typedef short as_short8 __attribute__((ext_vector_type(8)));
typedef unsigned as_uint4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* as_lds_ptr;
typedef const __attribute__((address_space(1))) void* as_gbl_ptr;

constexpr int AS_NT = 256, AS_N = 256;
constexpr int AS_OWN = 32768, AS_SLOT = 49152, AS_LDS = 3 * AS_SLOT;   // phase S slot: streamed chunk 32 KB | 4 waves x 4 KB own rows
constexpr int AS_UNIT = 16384;                                          // phase O unit (ring of 8 inside the same 144 KB)

template <int I, int N, class F>
__device__ __forceinline__ void as_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        as_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int as_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void as_dma16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((as_gbl_ptr)src, (as_lds_ptr)lds_dst, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ as_short4 as_read_tr(unsigned addr) {
    as_short4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ as_short8 as_read128(unsigned addr) {
    as_short8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ as_bf16x8 as_bf(as_short8 v) { return __builtin_bit_cast(as_bf16x8, v); }
__device__ __forceinline__ as_bf16x8 as_bf(as_short4 v0, as_short4 v1) {
    const as_short8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(as_bf16x8, v);
}
__device__ __forceinline__ unsigned as_pack_hi(float a, float b) { return bd_pack_hi(a, b); }   // common.h: the library's split
__device__ __forceinline__ unsigned as_pack_lo(float a, float b) { return bd_pack_lo(a, b); }
__device__ __forceinline__ unsigned as_split1(float v) { return bd_split1(v); }   // hi | lo << 16
__device__ __forceinline__ unsigned as_xor1(unsigned w) {   // the value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true);
}

#ifdef BD_AS_ABLATION
__device__ int g_as_ablate_dummy;
#define AS_ABL(L, bit) ((L).ablate & (bit))
#define AS_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define AS_STAMP(i) do {} while (0)
#define AS_ABL(L, bit) false
#endif
struct AsLane {
    int lane, wave, li, h;
    unsigned long long* dbg;
    int ablate;               // -DBD_AS_ABLATION builds only (BD_AS_ABLATE): 1 = no DMA, 2 = no global stores
    unsigned smem;            // LDS byte address of the workgroup's buffer
    unsigned kc[2][2];        // K-contiguous fragment offset [k16 step][plane] of row li of a 32-row tile (tile = immediate)
    unsigned kq[2][2];        // the same inside this wave's own-row region of a phase-S slot
    unsigned km[2][2];        // K-major fragment offset [tile parity][plane] (natural contraction order)
    unsigned kmp[2][2];       // ... in the permuted order of a phase-S accumulator: k-slots j of lane half h = rows 4h + (j&3) + 8(j>>2) (+16 per step)
};
__device__ __forceinline__ AsLane as_lane(const char* smem, int ablate = 0, unsigned long long* dbg = nullptr) {
    AsLane L;
    L.ablate = ablate; L.dbg = dbg;
    const int tid = threadIdx.x;
    L.lane = tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.li = L.lane & 31; L.h = L.lane >> 5;
    L.smem = (unsigned)(uintptr_t)(as_lds_ptr)smem;
    const int sl = L.lane & 15, hb = (L.lane >> 4) & 1, kq = sl >> 2, rq = sl & 3;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            L.kc[x][pl] = (unsigned)(L.li * 128 + (((pl * 4 + x * 2 + L.h) ^ as_swz(L.li)) << 4));
            L.kq[x][pl] = L.kc[x][pl] + AS_OWN + L.wave * 4096;
            const unsigned win = (unsigned)((((x * 2 + pl) ^ kq) << 6) + hb * 32 + rq * 8);
            L.km[x][pl] = (unsigned)((8 * L.h + kq) * 512) + win;
            L.kmp[x][pl] = (unsigned)((4 * L.h + kq) * 512) + win;
        }
    return L;
}

// ---- phase S ---------------------------------------------------------------------------------------------------------------------
// acc[kt][r] += sum_d streamed[kt*32 + (r&3) + 8(r>>2) + 4h][d] * own[li][d]   (d = 0 .. 32 nch - 1)
// streamed: 256 rows from `a_base` (stride lda bytes); own: this wave's 32 rows from `own_base` (stride ldo bytes); both split planes.
struct AsGroup { as_short8 a[4][2]; };
__device__ __forceinline__ void as_tie(AsGroup& g) {
    asm volatile("" : "+v"(g.a[0][0]), "+v"(g.a[0][1]), "+v"(g.a[1][0]), "+v"(g.a[1][1]), "+v"(g.a[2][0]), "+v"(g.a[2][1]), "+v"(g.a[3][0]),
                 "+v"(g.a[3][1]));
}
__device__ __forceinline__ void as_tie2(as_short8& a, as_short8& b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <class W>
__device__ __forceinline__ void as_phase_s(const char* a_base, long long lda, const char* own_base, long long ldo, int nch, char* smem,
                                           const AsLane& L, as_floatx16 (&acc)[8], W&& work) {
    // uniform chunk pointers + 32-bit lane offsets (saddr + voffset addressing: 12 address registers instead of 24)
    unsigned aoff[8], boff[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (L.wave + 4 * j) * 8 + (L.lane >> 3);
        aoff[j] = (unsigned)(r * (int)lda + (((L.lane & 7) ^ as_swz(r)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = j * 8 + (L.lane >> 3);
        boff[j] = (unsigned)(r * (int)ldo + (((L.lane & 7) ^ as_swz(r)) << 4));
    }
    const char* a_cur = a_base; const char* b_cur = own_base;
    auto issue = [&](int slot) {
        if (AS_ABL(L, 1)) return;
        char* st = smem + slot * AS_SLOT;
#pragma unroll
        for (int j = 0; j < 8; ++j) as_dma16(a_cur + aoff[j], st + (L.wave + 4 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) as_dma16(b_cur + boff[j], st + AS_OWN + L.wave * 4096 + j * 1024);
        a_cur += 128; b_cur += 128;
    };
    // eight chunks (dh = 256), statically unrolled, four fragment groups (k16 step, half of the key tiles) per chunk.  work(c) is VALU
    // work the caller wants under chunk c's first MFMA group.  The barrier that opens chunk c+1 sits BEFORE the last MFMA group of
    // chunk c, so that the first fragment reads of the next chunk run under those MFMAs instead of behind the barrier (one wave per
    // SIMD: nothing else would cover them).
    (void)nch;
    issue(0);
    issue(1);
    as_short8 q[2][2];
    AsGroup g0, g1;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto reads_q = [&](unsigned sb, auto S) {
        constexpr int s = decltype(S)::value;
        q[s][0] = as_read128<0>(sb + L.kq[s][0]);
        q[s][1] = as_read128<0>(sb + L.kq[s][1]);
    };
    auto reads = [&](unsigned sb, auto S, auto GI, AsGroup& g) {
        constexpr int s = decltype(S)::value, gi = decltype(GI)::value;
        as_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            g.a[t][0] = as_read128<(gi * 4 + t) * 4096>(sb + L.kc[s][0]);
            g.a[t][1] = as_read128<(gi * 4 + t) * 4096>(sb + L.kc[s][1]);
        });
    };
    auto wait = [&](AsGroup& g, auto S) {
        constexpr int s = decltype(S)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        as_tie(g);
        as_tie2(q[s][0], q[s][1]);
    };
    auto mfmas = [&](auto S, auto GI, const AsGroup& g) {
        constexpr int s = decltype(S)::value, gi = decltype(GI)::value;
        const as_bf16x8 qh = as_bf(q[s][0]), ql = as_bf(q[s][1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[gi * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(g.a[t][1]), qh, acc[gi * 4 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[gi * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(g.a[t][0]), ql, acc[gi * 4 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[gi * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(g.a[t][0]), qh, acc[gi * 4 + t], 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(2);
    reads_q(L.smem, I0{}); reads(L.smem, I0{}, I0{}, g0); wait(g0, I0{});
    as_for<0, 8>([&](auto CC) {
        constexpr int c = decltype(CC)::value, slot = c % 3, nslot = (c + 1) % 3;
        const unsigned sb = L.smem + slot * AS_SLOT, nb = L.smem + nslot * AS_SLOT;
        reads(sb, I0{}, I1{}, g1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I0{}, I0{}, g0);
        work(CC);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait(g1, I0{});
        reads_q(sb, I1{}); reads(sb, I1{}, I0{}, g0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I0{}, I1{}, g1);
        __builtin_amdgcn_sched_barrier(0);
        wait(g0, I1{});
        reads(sb, I1{}, I1{}, g1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I1{}, I0{}, g0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (c + 1 < 8) {
            // chunk c+1 has landed everywhere, every read of chunk c has returned: its slot takes chunk c+3
            if constexpr (c + 2 < 8) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (c + 3 < 8) issue(slot);
            as_tie(g1); as_tie2(q[1][0], q[1][1]);
            reads_q(nb, I0{}); reads(nb, I0{}, I0{}, g0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, I1{}, g1);
            __builtin_amdgcn_sched_barrier(0);
            wait(g0, I0{});
        } else {
            wait(g1, I1{});
            mfmas(I1{}, I1{}, g1);
        }
    });
}

// ---- phase O ---------------------------------------------------------------------------------------------------------------------
// oacc[dt][r] += sum_k M[own row (r&3) + 8(r>>2) + 4h ... as MFMA A operand][k] * streamed[k][dt*32 + li]     (k = 0..255, dt = 0..7)
// M: ah / al [chunk of 32 k][k16 step] packed A operands.  streamed rows from b_base (stride ldb bytes), 256 columns from the base.
// PERM: M's k-slots are in the permuted order of a phase-S accumulator (AsLane::kmp).
struct AsStream { const char* base; unsigned off[4]; long long ldb; };
__device__ __forceinline__ AsStream as_o_open(const char* b_base, long long ldb, const AsLane& L) {
    AsStream s;
    s.base = b_base; s.ldb = ldb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 2 * (L.wave + 4 * j) + (L.lane >> 5);
        s.off[j] = (unsigned)(k * (int)ldb + (((L.lane & 31) ^ ((k & 3) << 2)) << 4));
    }
    return s;
}
template <int U>
__device__ __forceinline__ void as_o_issue(const AsStream& s, char* smem, const AsLane& L) {
    constexpr int kc = U & 7, half = U >> 3;
    if (AS_ABL(L, 1)) return;
    const char* ub = s.base + (long long)kc * 32 * s.ldb + half * 512;   // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) as_dma16(ub + s.off[j], smem + kc * AS_UNIT + (L.wave + 4 * j) * 1024);
}
// units 0..5 in flight; the caller has passed a barrier behind the last read of the ring
__device__ __forceinline__ void as_o_prologue(const AsStream& s, char* smem, const AsLane& L) {
    as_for<0, 6>([&](auto U) { as_o_issue<decltype(U)::value>(s, smem, L); });
}
struct AsOFrag { as_short4 v0[4][2], v1[4][2]; };
__device__ __forceinline__ void as_tie(AsOFrag& f) {
    asm volatile("" : "+v"(f.v0[0][0]), "+v"(f.v0[0][1]), "+v"(f.v0[1][0]), "+v"(f.v0[1][1]), "+v"(f.v0[2][0]), "+v"(f.v0[2][1]), "+v"(f.v0[3][0]),
                 "+v"(f.v0[3][1]));
    asm volatile("" : "+v"(f.v1[0][0]), "+v"(f.v1[0][1]), "+v"(f.v1[1][0]), "+v"(f.v1[1][1]), "+v"(f.v1[2][0]), "+v"(f.v1[2][1]), "+v"(f.v1[3][0]),
                 "+v"(f.v1[3][1]));
}
// 16 units (2 column halves x 8 chunks of 32 k), six in flight.  work(u) is VALU / store work the caller wants executed in the shadow of
// unit u's first twelve MFMAs (the matrix pipe runs an MFMA for 32 cycles; independent VALU instructions issue meanwhile).  The barrier
// that opens unit u+1 sits before the second MFMA group of unit u, whose shadow then covers the first fragment reads of unit u+1.
// vmcnt counts stores as well as loads and the two complete out of order, so a counted wait is exact only while no store is in flight:
// the last DMA is issued while unit 9 runs, the barrier that opens unit 12 waits for everything, and only work(u >= 12) may store.
constexpr int AS_O_STORE_FROM = 12;
template <bool PERM, class W>
__device__ __forceinline__ void as_o_run(const AsStream& s, char* smem, const AsLane& L, const as_uint4 (&ah)[8][2], const as_uint4 (&al)[8][2],
                                         as_floatx16 (&oacc)[8], W&& work) {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    AsOFrag f0, f1;
    auto reads = [&](unsigned ub, auto S, AsOFrag& f) {
        constexpr int sp = decltype(S)::value;
        constexpr int second = PERM ? 8 * 512 : 4 * 512;
#ifdef BD_AS_ABLATION
        if (AS_ABL(L, 4)) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) f.v0[t][pl] = f.v1[t][pl] = as_short4{0x3f80, 0x3f80, (short)sp, 0x3f80};
            return;
        }
#endif
        as_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned a = ub + (PERM ? L.kmp[t & 1][pl] : L.km[t & 1][pl]);
                f.v0[t][pl] = as_read_tr<sp * 16 * 512 + (t >> 1) * 256>(a);
                f.v1[t][pl] = as_read_tr<sp * 16 * 512 + (t >> 1) * 256 + second>(a);
            }
        });
    };
    auto wait = [&](AsOFrag& f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        as_tie(f);
    };
    asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    as_o_issue<6>(s, smem, L);
    reads(L.smem, I0{}, f0);
    wait(f0);
    as_for<0, 16>([&](auto UU) {
        constexpr int u = decltype(UU)::value, kc = u & 7, half = u >> 3, nkc = (u + 1) & 7;
#ifdef BD_AS_ABLATION
        if (L.dbg && threadIdx.x == 0) L.dbg[blockIdx.x * 32 + 8 + u] = __builtin_readcyclecounter();
#endif
        const unsigned ub = L.smem + kc * AS_UNIT, nub = L.smem + nkc * AS_UNIT;
        auto mfmas = [&](auto S, const AsOFrag& f) {
            constexpr int sp = decltype(S)::value;
            const as_bf16x8 mh = __builtin_bit_cast(as_bf16x8, ah[kc][sp]), ml = __builtin_bit_cast(as_bf16x8, al[kc][sp]);
#pragma unroll
            for (int t = 0; t < 4; ++t) oacc[half * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml, as_bf(f.v0[t][0], f.v1[t][0]), oacc[half * 4 + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) oacc[half * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh, as_bf(f.v0[t][1], f.v1[t][1]), oacc[half * 4 + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) oacc[half * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh, as_bf(f.v0[t][0], f.v1[t][0]), oacc[half * 4 + t], 0, 0, 0);
        };
        reads(ub, I1{}, f1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I0{}, f0);
        work(UU);
#pragma unroll
        for (int i = 0; i < 12; ++i) {      // one MFMA, then the VALU / store work that fits under it
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + 1 < 16) {
            // unit u+1 has landed everywhere and every read of unit u has returned
            if constexpr (u + 1 <= 10) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
            else if constexpr (u + 1 == 11) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if constexpr (u + 1 == 12) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (u + 7 < 16) as_o_issue<u + 7>(s, smem, L);
            as_tie(f1);
            reads(nub, I0{}, f0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, f1);
            __builtin_amdgcn_sched_barrier(0);
            wait(f0);
        } else {
            wait(f1);
            mfmas(I1{}, f1);
        }
    });
}

// ---- stores ----------------------------------------------------------------------------------------------------------------------
// a 32 x 32 accumulator tile (rows (r&3) + 8(r>>2) + 4h, column li of a 32-column block) as split planes: lane pairs trade halves so that
// every lane stores ONE dword per row -- even lanes the hi pair (li, li+1), odd lanes the lo pair (li-1, li).  `ubase` (uniform) = first
// row of the tile + 128 * column block; `lane_off` = 4h * ld + the lane's dword inside the 128-byte line.
struct AsPair { unsigned off, sel; bool odd; bool nostore; };
__device__ __forceinline__ AsPair as_pair(const AsLane& L, long long ld) {
    AsPair q;
    q.odd = L.li & 1;
    q.nostore = AS_ABL(L, 2);
    q.off = (unsigned)(4 * L.h * (int)ld + (q.odd ? 64 + (L.li - 1) * 2 : L.li * 2));
    q.sel = q.odd ? 0x03020706u : 0x05040100u;   // v_perm_b32(neighbour, own): odd (nb >> 16) | (own & 0xFFFF0000), even (own & 0xFFFF) | (nb << 16)
    return q;
}
__device__ __forceinline__ void as_store_word(char* urow, const AsPair& q, unsigned w) {
#ifdef BD_AS_ABLATION
    if (q.nostore) { asm volatile("" ::"v"(__builtin_amdgcn_perm(as_xor1(w), w, q.sel))); return; }
#endif
    *reinterpret_cast<unsigned*>(urow + q.off) = __builtin_amdgcn_perm(as_xor1(w), w, q.sel);
}
__device__ __forceinline__ void as_store_tile(char* ubase, long long ld, const AsPair& q, const as_floatx16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) as_store_word(ubase + (long long)((r & 3) + 8 * (r >> 2)) * ld, q, as_split1(v[r]));
}
// tile kt of the packed A operands back out as planes of the TRANSPOSED matrix [k][own row]: value r = element r&7 of step r>>3
__device__ __forceinline__ void as_store_packed_tile(char* ubase, long long ld, const AsPair& q, const as_uint4 (&ah)[2], const as_uint4 (&al)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned hw = ah[r >> 3][(r & 7) >> 1], lw = al[r >> 3][(r & 7) >> 1];
        const unsigned w = __builtin_amdgcn_perm(lw, hw, (r & 1) ? 0x07060302u : 0x05040100u);   // hi16 | lo16 << 16 of value r
        as_store_word(ubase + (long long)((r & 3) + 8 * (r >> 2)) * ld, q, w);
    }
}
// fp32 values of one tile (accumulator layout), times `mul`, -> packed A operands
__device__ __forceinline__ void as_pack_tile(const as_floatx16& v, float mul, as_uint4 (&ah)[2], as_uint4 (&al)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = __fmul_rn(v[8 * s + 2 * j], mul), b = __fmul_rn(v[8 * s + 2 * j + 1], mul);
            ah[s][j] = as_pack_hi(a, b);
            al[s][j] = as_pack_lo(a, b);
        }
}

struct AsParams {
    const char* qkv; long long ld;          // q | k | v planes: [B*N] rows of ld bytes, q at column byte 0, k at C*4, v at 2*C*4
    char* o; long long ldo;                 // forward: O planes [B*N][ldo bytes]
    char* pt;                               // P^T planes [B*heads][N keys][N queries] (forward: out, may be null; backward: in)
    const char* dO; long long lddo;         // backward: dO planes
    char* dst;                              // backward: dS^T planes [B*heads][N][N] (A: out, B: in)
    char* dqkv; long long lddqkv;           // backward: dq | dk | dv planes (layout of qkv)
    int C, dh, heads;
    float scale;
    int ablate;
    unsigned long long* dbg;   // -DBD_AS_ABLATION builds: [workgroup][8] s_memtime stamps of wave 0
};

// workgroup -> (sample * heads + head, 128-row block); the two blocks of a sample are neighbours in one XCD's run (shared K / V in L2)
__device__ __forceinline__ void as_coord(int& bh, int& blk) {
    const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
    const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
    bh = j >> 1; blk = j & 1;
}

template <bool WRITE_P>
__global__ __launch_bounds__(AS_NT, 1) void attn_sp_fwd_kernel(AsParams p) {
    __shared__ __attribute__((aligned(128))) char smem[AS_LDS];
    const AsLane L = as_lane(smem, p.ablate, p.dbg);
    int bh, qb;
    as_coord(bh, qb);
    const int b = bh / p.heads, hd = bh - b * p.heads;
    const long long colq = (long long)hd * p.dh * 4;
    const char* base = p.qkv + (long long)b * AS_N * p.ld + colq;
    const long long own0 = (long long)qb * 128 + L.wave * 32;

    as_floatx16 acc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
    AS_STAMP(0);
    as_phase_s(base + (long long)p.C * 4, p.ld, base + own0 * p.ld, p.ld, p.dh >> 5, smem, L, acc, [](auto) {});
    AS_STAMP(1);
    __builtin_amdgcn_s_barrier();
    const AsStream vs = as_o_open(base + (long long)p.C * 8, p.ld, L);
    as_o_prologue(vs, smem, L);

    // softmax over the keys of own query li (128 values in this lane, the other 128 in lane ^ 32), in base 2: exp(x) = 2^(x log2 e)
    const float c2 = p.scale * 1.4426950408889634f;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[kt][r] = __fmul_rn(acc[kt][r], c2); m = fmaxf(m, acc[kt][r]); }   // _rn: never contracted, so the
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[kt][r] = __builtin_amdgcn_exp2f(__fsub_rn(acc[kt][r], m)); sum += acc[kt][r]; }   // training and inference forms round alike
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // P as packed A operands, one key tile per unit of the first sweep (tile 0 now); P^T and the first half of O leave in the last units
    as_uint4 ah[8][2], al[8][2];
    as_pack_tile(acc[0], inv, ah[0], al[0]);
    as_floatx16 oacc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    AS_STAMP(2);
    const AsPair po = as_pair(L, p.ldo), pp = as_pair(L, AS_N * 4);
    char* orow = p.o + ((long long)b * AS_N + own0) * p.ldo + colq;                                   // uniform
    char* ptb = WRITE_P ? p.pt + (long long)bh * AS_N * AS_N * 4 + (qb * 4 + L.wave) * 128 : nullptr;  // uniform
    as_o_run<true>(vs, smem, L, ah, al, oacc, [&](auto UU) {
        constexpr int u = decltype(UU)::value;
        if constexpr (u < 7) as_pack_tile(acc[u + 1], inv, ah[u + 1], al[u + 1]);
        if constexpr (u >= AS_O_STORE_FROM) {
            constexpr int i = u - AS_O_STORE_FROM;
            if constexpr (WRITE_P) {
                as_store_packed_tile(ptb + (long long)(2 * i) * 32 * AS_N * 4, AS_N * 4, pp, ah[2 * i], al[2 * i]);
                as_store_packed_tile(ptb + (long long)(2 * i + 1) * 32 * AS_N * 4, AS_N * 4, pp, ah[2 * i + 1], al[2 * i + 1]);
            }
            as_store_tile(orow + i * 128, p.ldo, po, oacc[i]);
        }
    });
    AS_STAMP(3);
#pragma unroll
    for (int dt = 4; dt < 8; ++dt) as_store_tile(orow + dt * 128, p.ldo, po, oacc[dt]);
    AS_STAMP(4);
}

__global__ __launch_bounds__(AS_NT, 1) void attn_sp_bwd_a_kernel(AsParams p) {
    __shared__ __attribute__((aligned(128))) char smem[AS_LDS];
    const AsLane L = as_lane(smem, p.ablate, p.dbg);
    int bh, qb;
    as_coord(bh, qb);
    const int b = bh / p.heads, hd = bh - b * p.heads;
    const long long colq = (long long)hd * p.dh * 4;
    const char* base = p.qkv + (long long)b * AS_N * p.ld + colq;
    const long long own0 = (long long)qb * 128 + L.wave * 32;
    const AsPair pp = as_pair(L, AS_N * 4), pq = as_pair(L, p.lddqkv);

    // P^T of own query column, accumulator layout: one dword per value (even lanes the hi pair, odd lanes the lo pair), issued first
    unsigned pw[8][16];
    {
        const char* pt = p.pt + (long long)bh * AS_N * AS_N * 4 + (qb * 4 + L.wave) * 128;   // uniform
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                pw[kt][r] = *reinterpret_cast<const unsigned*>(pt + (long long)(kt * 32 + (r & 3) + 8 * (r >> 2)) * (AS_N * 4) + pp.off);
    }
    as_floatx16 acc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
    // dP^T = V dO^T; under the MFMAs of chunk c, tile c of P goes from its plane words back to fp32 (hi + lo, exact)
    const unsigned sel_hi = pp.odd ? 0x07060c0cu : 0x01000c0cu, sel_lo = pp.odd ? 0x03020c0cu : 0x05040c0cu;   // v_perm(nb, own): 0x0c = zero byte
    as_phase_s(base + (long long)p.C * 8, p.ld, p.dO + ((long long)b * AS_N + own0) * p.lddo + colq, p.lddo, p.dh >> 5, smem, L, acc, [&](auto CC) {
        constexpr int c = decltype(CC)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned x = pw[c][r], nb = as_xor1(x);
            const float pv = __builtin_bit_cast(float, __builtin_amdgcn_perm(nb, x, sel_hi)) + __builtin_bit_cast(float, __builtin_amdgcn_perm(nb, x, sel_lo));
            pw[c][r] = __builtin_bit_cast(unsigned, pv);
        }
    });
    __builtin_amdgcn_s_barrier();
    const AsStream ks = as_o_open(base + (long long)p.C * 4, p.ld, L);
    as_o_prologue(ks, smem, L);

    // dS = scale * P o (dP - delta), delta = sum over the keys of P o dP
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += __builtin_bit_cast(float, pw[kt][r]) * acc[kt][r];
    delta += __shfl_xor(delta, 32, 64);
    auto ds_tile = [&](int kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = p.scale * __builtin_bit_cast(float, pw[kt][r]) * (acc[kt][r] - delta);
    };
    as_uint4 ah[8][2], al[8][2];
    ds_tile(0);
    as_pack_tile(acc[0], 1.f, ah[0], al[0]);
    as_floatx16 oacc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    char* qrow = p.dqkv + ((long long)b * AS_N + own0) * p.lddqkv + colq;                              // uniform
    char* dsb = p.dst + (long long)bh * AS_N * AS_N * 4 + (qb * 4 + L.wave) * 128;                      // uniform
    as_o_run<true>(ks, smem, L, ah, al, oacc, [&](auto UU) {   // dQ = dS K
        constexpr int u = decltype(UU)::value;
        if constexpr (u < 7) { ds_tile(u + 1); as_pack_tile(acc[u + 1], 1.f, ah[u + 1], al[u + 1]); }
        if constexpr (u >= AS_O_STORE_FROM) {
            constexpr int i = u - AS_O_STORE_FROM;
            as_store_packed_tile(dsb + (long long)(2 * i) * 32 * AS_N * 4, AS_N * 4, pp, ah[2 * i], al[2 * i]);
            as_store_packed_tile(dsb + (long long)(2 * i + 1) * 32 * AS_N * 4, AS_N * 4, pp, ah[2 * i + 1], al[2 * i + 1]);
            as_store_tile(qrow + i * 128, p.lddqkv, pq, oacc[i]);
        }
    });
#pragma unroll
    for (int dt = 4; dt < 8; ++dt) as_store_tile(qrow + dt * 128, p.lddqkv, pq, oacc[dt]);
}

__global__ __launch_bounds__(AS_NT, 1) void attn_sp_bwd_b_kernel(AsParams p) {
    __shared__ __attribute__((aligned(128))) char smem[AS_LDS];
    const AsLane L = as_lane(smem, p.ablate, p.dbg);
    int bh, kb;
    as_coord(bh, kb);
    const int b = bh / p.heads, hd = bh - b * p.heads;
    const long long colq = (long long)hd * p.dh * 4;
    const long long own0 = (long long)kb * 128 + L.wave * 32;
    const AsPair pq = as_pair(L, p.lddqkv);

    // own 32 rows (keys) of a [N keys][N queries] plane matrix as A operands: chunk c, step s = 16 bytes of hi and of lo
    auto load_slab = [&](const char* m, as_uint4 (&ah)[8][2], as_uint4 (&al)[8][2]) {
        const char* row = m + (long long)bh * AS_N * AS_N * 4 + own0 * (AS_N * 4);   // uniform
        const unsigned off = (unsigned)(L.li * (AS_N * 4) + L.h * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ah[c][s] = *reinterpret_cast<const as_uint4*>(row + c * 128 + s * 32 + off);
                al[c][s] = *reinterpret_cast<const as_uint4*>(row + c * 128 + 64 + s * 32 + off);
            }
    };
    as_uint4 ah[8][2], al[8][2];
    as_floatx16 oacc[8];
    char* orow = p.dqkv + ((long long)b * AS_N + own0) * p.lddqkv + colq;   // uniform; dk at + C*4, dv at + C*8
    auto run = [&](const AsStream& st, char* out) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
        as_o_run<false>(st, smem, L, ah, al, oacc, [&](auto UU) {
            constexpr int u = decltype(UU)::value;
            if constexpr (u >= AS_O_STORE_FROM) as_store_tile(out + (u - AS_O_STORE_FROM) * 128, p.lddqkv, pq, oacc[u - AS_O_STORE_FROM]);
        });
#pragma unroll
        for (int dt = 4; dt < 8; ++dt) as_store_tile(out + dt * 128, p.lddqkv, pq, oacc[dt]);
    };
    // dV = P^T dO
    AS_STAMP(0);
    load_slab(p.pt, ah, al);
    const AsStream ds = as_o_open(p.dO + (long long)b * AS_N * p.lddo + colq, p.lddo, L);
    as_o_prologue(ds, smem, L);
    AS_STAMP(1);
    run(ds, orow + (long long)p.C * 8);
    AS_STAMP(2);
    // dK = dS^T Q  (dS already carries the softmax scale)
    load_slab(p.dst, ah, al);
    const AsStream qs = as_o_open(p.qkv + (long long)b * AS_N * p.ld + colq, p.ld, L);
    as_o_prologue(qs, smem, L);
    AS_STAMP(3);
    run(qs, orow + (long long)p.C * 4);
    AS_STAMP(4);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
bool attn_sp_supported(int N, int dh) {
    static const bool off = getenv("BD_ATTN_SP") && atoi(getenv("BD_ATTN_SP")) == 0;
    return !off && N == AS_N && dh == 256;
}

#ifdef BD_AS_ABLATION
static unsigned long long* as_dbg_buf() {
    static unsigned long long* d = nullptr;
    if (!d && getenv("BD_AS_STAMPS")) { (void)hipMalloc(&d, 4096 * 32 * 8); (void)hipMemset(d, 0, 4096 * 32 * 8); }
    return d;
}
static void as_dbg_print(const char* who, int wgs, hipStream_t st) {
    unsigned long long* d = as_dbg_buf();
    if (!d) return;
    (void)hipStreamSynchronize(st);
    static unsigned long long h[4096 * 32];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double seg[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < wgs; ++w)
        for (int i = 0; i < 4; ++i) seg[i] += (double)(h[w * 32 + i + 1] - h[w * 32 + i]);
    fprintf(stderr, "[stamps] %s: segments (cycles, mean over %d workgroups):", who, wgs);
    for (int i = 0; i < 4; ++i) fprintf(stderr, " %.0f", seg[i] / wgs);
    fprintf(stderr, "\n[stamps]   last O run, cycles per unit:");
    for (int u = 0; u < 15; ++u) {
        double t = 0;
        for (int w = 0; w < wgs; ++w) t += (double)(h[w * 32 + 8 + u + 1] - h[w * 32 + 8 + u]);
        fprintf(stderr, " %.0f", t / wgs);
    }
    fprintf(stderr, "\n");
}
#endif
static int attn_sp_common(const bd_attn_sp_desc& d, const char* who, AsParams& p) {
    BD_CHECK(d.B > 0 && d.heads > 0 && attn_sp_supported(d.N, d.dh), BD_ERR_UNSUPPORTED, "%s: needs N == 256 and head dim 256 (got N %d, dh %d)", who,
             d.N, d.dh);
    const int C = d.heads * d.dh;
    BD_CHECK(d.qkv_split && d.ld >= 3 * C && d.ld % 32 == 0 && ((uintptr_t)d.qkv_split & 127) == 0, BD_ERR_INVALID,
             "%s: qkv planes must be 128-byte aligned with a row stride >= 3C, multiple of 32", who);
    p = {};
    p.qkv = reinterpret_cast<const char*>(d.qkv_split); p.ld = d.ld * 4;
    p.C = C; p.dh = d.dh; p.heads = d.heads; p.scale = d.scale;
#ifdef BD_AS_ABLATION
    p.ablate = getenv("BD_AS_ABLATE") ? atoi(getenv("BD_AS_ABLATE")) : 0;
#endif
    return BD_OK;
}

int attn_sp_fwd(const bd_attn_sp_desc& d, hipStream_t st) {
    AsParams p;
    BD_TRY(attn_sp_common(d, "bd_attn_sp_fwd", p));
    BD_CHECK(d.o_split && d.ldo >= p.C && d.ldo % 32 == 0 && ((uintptr_t)d.o_split & 127) == 0, BD_ERR_INVALID, "bd_attn_sp_fwd: bad output planes");
    BD_CHECK(((uintptr_t)d.pt_split & 127) == 0, BD_ERR_INVALID, "bd_attn_sp_fwd: P^T planes must be 128-byte aligned");
    p.o = reinterpret_cast<char*>(d.o_split); p.ldo = d.ldo * 4; p.pt = reinterpret_cast<char*>(d.pt_split);
    const dim3 grid((unsigned)(d.B * d.heads * 2));
#ifdef BD_AS_ABLATION
    p.dbg = as_dbg_buf();
#endif
    // bench.py roofline: S = Q K^T and O = P V (4 B N^2 dh flop per head); bytes = q, k, v in, o (+ P^T for training) out
    const double bh = (double)d.B * d.heads, nn = (double)d.N * d.N, nd = (double)d.N * d.dh;
    const int rec = prof_on() ? prof_begin("attn_sp_fwd", 4.0 * bh * nn * d.dh, 4.0 * bh * (4.0 * nd + (d.pt_split ? nn : 0.0)), st) : -1;
    if (d.pt_split) hipLaunchKernelGGL(attn_sp_fwd_kernel<true>, grid, dim3(AS_NT), 0, st, p);
    else hipLaunchKernelGGL(attn_sp_fwd_kernel<false>, grid, dim3(AS_NT), 0, st, p);
    BD_LAUNCH_CHECK("attn_sp_fwd");
    prof_end(rec, st);
#ifdef BD_AS_ABLATION
    as_dbg_print("fwd  S | softmax | O run | tail stores", (int)grid.x, st);
#endif
    return BD_OK;
}

int attn_sp_bwd(const bd_attn_sp_desc& d, hipStream_t st) {
    AsParams p;
    BD_TRY(attn_sp_common(d, "bd_attn_sp_bwd", p));
    BD_CHECK(d.pt_split && d.do_split && d.dst_split && d.dqkv_split, BD_ERR_INVALID, "bd_attn_sp_bwd: null pointer");
    BD_CHECK(d.lddo >= p.C && d.lddo % 32 == 0 && d.lddqkv >= 3 * p.C && d.lddqkv % 32 == 0, BD_ERR_INVALID, "bd_attn_sp_bwd: bad row strides");
    BD_CHECK((((uintptr_t)d.pt_split | (uintptr_t)d.do_split | (uintptr_t)d.dst_split | (uintptr_t)d.dqkv_split) & 127) == 0, BD_ERR_INVALID,
             "bd_attn_sp_bwd: planes must be 128-byte aligned");
    p.pt = reinterpret_cast<char*>(const_cast<uint16_t*>(d.pt_split));
    p.dO = reinterpret_cast<const char*>(d.do_split); p.lddo = d.lddo * 4;
    p.dst = reinterpret_cast<char*>(d.dst_split);
    p.dqkv = reinterpret_cast<char*>(d.dqkv_split); p.lddqkv = d.lddqkv * 4;
    const dim3 grid((unsigned)(d.B * d.heads * 2));
    // bench.py roofline.  A: dP = dO V^T, dQ = dS K (4 B N^2 dh flop per head; reads k, v, dO, P^T; writes dq, dS^T).
    // B: dV = P^T dO, dK = dS^T Q (same flops; reads P^T, dS^T, dO, q; writes dk, dv).
    const double bh = (double)d.B * d.heads, nn = (double)d.N * d.N, nd = (double)d.N * d.dh;
    int rec = prof_on() ? prof_begin("attn_sp_bwd_a", 4.0 * bh * nn * d.dh, 4.0 * bh * (4.0 * nd + 2.0 * nn), st) : -1;
    hipLaunchKernelGGL(attn_sp_bwd_a_kernel, grid, dim3(AS_NT), 0, st, p);
    prof_end(rec, st);
#ifdef BD_AS_ABLATION
    p.dbg = as_dbg_buf();
#endif
    rec = prof_on() ? prof_begin("attn_sp_bwd_b", 4.0 * bh * nn * d.dh, 4.0 * bh * (4.0 * nd + 2.0 * nn), st) : -1;
    hipLaunchKernelGGL(attn_sp_bwd_b_kernel, grid, dim3(AS_NT), 0, st, p);
    prof_end(rec, st);
    BD_LAUNCH_CHECK("attn_sp_bwd");
#ifdef BD_AS_ABLATION
    as_dbg_print("bwdB slab1 | run1 | slab2+prologue | run2", (int)grid.x, st);
#endif
    return BD_OK;
}

}  // namespace bd

extern "C" int bd_attn_sp_supported(int N, int dh) { return bd::attn_sp_supported(N, dh) ? 1 : 0; }
extern "C" int bd_attn_sp_fwd(const bd_attn_sp_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_attn_sp_fwd: null descriptor");
    return bd::attn_sp_fwd(*d, bd::S(s));
}
extern "C" int bd_attn_sp_bwd(const bd_attn_sp_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_attn_sp_bwd: null descriptor");
    return bd::attn_sp_bwd(*d, bd::S(s));
}
