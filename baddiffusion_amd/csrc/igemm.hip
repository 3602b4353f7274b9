// Implicit-GEMM engine for gfx950 on the f32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// One workgroup = 4 waves (2x2) computing a BMxBN tile (128x128 or 64x64), K walked in chunks of 32.
// Operand tiles are staged global -> registers -> LDS in the layout their memory contiguity dictates:
//   KC ("k contiguous"):   LDS [rows][32+4]  -> fragments read with one ds_read_b128 per 4 MFMAs
//   RC ("row contiguous"): LDS [32][rows+4]  -> fragments read with ds_read_b32 (transpose for free)
// The f32 MFMA takes ONE float per lane per operand (lane l: row l&31, k-slot l>>5), so any
// (row,k) -> memory map works without a transpose pass; the K order inside a chunk is permuted
// identically for A and B (k(g,j,h) = 8g + 4h + j) which leaves the dot product unchanged.
// Products and accumulation are exact fp32 (cdna_hip_programming.md section 3): no reduced-precision inputs.
//
// Operand kinds (bd_hip.h): DENSE, CONV (3x3 gather, fwd and wgrad), TCONV (dgrad gather), WGT
// (weights seen from the dgrad side).  Each kind has a FAST loader (float4 only, no integer division
// in the K loop: the (tap, channel) cursor of a conv operand advances incrementally and is wave-uniform)
// used when the shape is aligned (C % 32 == 0, ld % 4 == 0, ...), and every shape falls back to the
// GENERIC loaders (per-element address computation, scalar loads) otherwise -- conv_in (Cin = 3),
// conv_out (Cout = 3), odd GEMM sizes.  Replaces aten::convolution(_backward), addmm/mm/bmm/baddbmm
// of the reference's UNet.
#include "common.h"
#include <cstdint>
#include <cstdlib>

namespace bd {
#ifndef WGRAD_DEPTH
#define WGRAD_DEPTH 2
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// split 8 fp32 values into hi + lo bf16 planes (common.h: hi = RNE_bf16(x), lo = RNE_bf16(x - hi)); x = hi + lo + O(2^-18 |x|).
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)x[j];
        lo[j] = (__bf16)(x[j] - (float)hi[j]);
    }
}

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // KC row stride (floats): 16B-aligned rows, conflict-free b128 reads

struct Opnd {
    const float* p;
    const unsigned short* split;     // pre-split copy of the same buffer (bd_split_bf16 blocked layout) or null
    long long ld;
    int kind, vec;
    int C, Hs, Ws, Ho, Wo, stride, pad_t, pad_l, ups;
    int rows;      // number of valid rows (M for A, N for B)
    int rows_k;    // extent of the K dimension (RC conv gather: number of output pixels)
    int lw, lhw;   // log2(Wo), log2(Ho*Wo) when both are powers of two, else -1
};

struct IGemmParams {
    Opnd A, B;
    int M, N, K;
    int tiles_m, tiles_n;
    int batch_inner, ksplit, chunks_per_split;
    long long a_bso, a_bsi, b_bso, b_bsi, c_bso, c_bsi;
    float* C;
    long long ldc;
    float alpha, out_scale;
    const float* bias;
    const float* rowbias;
    long long ld_rowbias;
    int rows_per_group;
    const float* residual;
    long long ldr;
    int accumulate;
    float* partial;  // [batch*ksplit][M][N] when ksplit > 1, then [ksplit][M] column-sum partials
    float* a_colsum; // optional: out[m] = sum_k A(m,k) (RC A operand, batch 1)
};

// KC thread map: thread t owns float4 column (t&7)*4 of rows krow(t) + 32*i.  krow permutes the rows inside every
// group of 8 so that the two rows a 16-lane ds_write_b64 group touches are 4 apart: with the 80 B row stride of the
// split-bf16 planes their bank ranges then do not overlap (rows 1 apart overlap in 4 banks -> 2-way conflicts).
__device__ __forceinline__ int krow(int tid) {
    const int q = tid >> 3;
    return (q & ~7) | ((q & 1) << 2) | ((q >> 1) & 3);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// predicated 16-byte load without control flow and without a dependent select: masked-off lanes read 16 zero bytes
// that live in the code object.  Branch-free loads keep the s_waitcnt vmcnt() counting exact, and nothing touches the
// loaded registers until the split, which the two-chunk-deep register prefetch of the kernels depends on (a skipped
// load would force vmcnt(0) at the join, a select would wait for the load right behind its issue).
__device__ __attribute__((aligned(16))) const float kZero16[4] = {0.f, 0.f, 0.f, 0.f};
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1))) * gptr4;   // global address space: global_load, not flat_load
__device__ __forceinline__ float4 ld4_if(const float* p, bool ok, const float*) {
    const v4f t = *(gptr4)(ok ? p : kZero16);
    return make_float4(t.x, t.y, t.z, t.w);
}

// pixel index -> (b, y, x) over an (Ho, Wo) grid
__device__ __forceinline__ void decode_pixel(const Opnd& o, int pix, int& b, int& y, int& x) {
    if (o.lw >= 0) {
        x = pix & (o.Wo - 1);
        y = (pix >> o.lw) & (o.Ho - 1);
        b = pix >> o.lhw;
    } else {
        const int hw = o.Ho * o.Wo;
        b = pix / hw;
        const int rem = pix - b * hw;
        y = rem / o.Wo;
        x = rem - y * o.Wo;
    }
}

// =================================================================================================
// FAST loaders.  Interface: init(opnd, row0, tid, kbase0, K) ; load(v) for the current chunk ; advance().
// KC: tile [R rows][32 k]; thread t owns float4 column k4 = (t&7)*4 of rows (t>>3) + 32*i.
// RC: tile [32 k][R rows]; thread t owns float4 at rows r4 = (t % (R/4))*4, k = t/(R/4) + KS*i.
// =================================================================================================
template <int R, int NT = 256>
struct KCStore {
    static constexpr int NI = R * 8 / NT;
    __device__ __forceinline__ static void store(float* s, int tid, const float4 (&v)[NI]) {
        const int k4 = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(s + (krow(tid) + (NT / 8) * i) * LDK + k4) = v[i];
    }
    __device__ __forceinline__ static void store_split(unsigned short* sh, unsigned short* sl, int tid, const float4 (&v)[NI]);
};
// RC thread map: every thread owns the float4 of rows r4..r4+3 at NI values of k, k_i = kfirst + KSTEP*i.
//   r4 = (t % (R/4))*4, kfirst = t / (R/4), KSTEP = NT*4/R   (NT = threads of the workgroup: 256 or 512)
template <int R, int NT>
struct RCMap {
    static constexpr int NI = R * 8 / NT;          // float4 per thread per chunk
    static constexpr int KSTEP = NT * 4 / R;       // k rows covered per pass
    __device__ __forceinline__ static int r4(int t) { return (t % (R / 4)) * 4; }
    __device__ __forceinline__ static int kfirst(int t) { return t / (R / 4); }
};

constexpr int LDH = BK + 8;  // bf16 row stride of the split-bf16 LDS planes: 80 B (conflict-free b128 reads, b64 writes)

__device__ __forceinline__ unsigned pack_hi(float a, float b) { return bd_pack_hi(a, b); }   // common.h: RNE hi planes
__device__ __forceinline__ unsigned pack_lo(float a, float b) { return bd_pack_lo(a, b); }

template <int R, int NT>
struct RCStore {
    static constexpr int NI = R * 8 / NT;
    __device__ __forceinline__ static void store(float* s, int tid, const float4 (&v)[NI]) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(s + (k0 + RCMap<R, NT>::KSTEP * i) * (R + 4) + r4) = v[i];
    }
    // split-bf16 planes, row-contiguous image [k][R + 32] bf16 (the global loads stay fully coalesced); the MFMA
    // fragments are gathered from it with ds_read_b64_tr_b16 (hardware 4x4 transpose), see rc_frag()
    __device__ __forceinline__ static void store_split(unsigned short* sh, unsigned short* sl, int tid, const float4 (&v)[NI]) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int o = (k0 + RCMap<R, NT>::KSTEP * i) * (R + 32) + r4;
            *reinterpret_cast<uint2*>(sh + o) = make_uint2(pack_hi(v[i].x, v[i].y), pack_hi(v[i].z, v[i].w));
            *reinterpret_cast<uint2*>(sl + o) = make_uint2(pack_lo(v[i].x, v[i].y), pack_lo(v[i].z, v[i].w));
        }
    }
};

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_short4;

// MFMA 32x32x16 bf16 fragment (lane: row = lane&31 of the 32-row tile at `rowtile`, k = kk + 8*(lane>>5) + 0..7) from a
// row-contiguous plane [k][LD].  ds_read_b64_tr_b16 semantics on gfx950 (measured, scripts/tr_probe.hip): inside each
// 16-lane group, lane l receives element (l&3) of the 64-bit words read by lanes (l>>2) + 4j, j = 0..3.  So lane sl
// of a group reads the 4 rows [rb + 4*(sl&3), +4) at k = k0 + (sl>>2) and receives row rb + sl at k0..k0+3.
template <int LD>
__device__ __forceinline__ bf16x8 rc_frag(const unsigned short* plane, int rowtile, int kk, int lane) {
    const int sl = lane & 15;
    const int o = (kk + 8 * (lane >> 5) + (sl >> 2)) * LD + rowtile + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
    const short4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(plane + o));
    const short4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(plane + o + 4 * LD));
    const short8v v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int R, int NT>
__device__ __forceinline__ void KCStore<R, NT>::store_split(unsigned short* sh, unsigned short* sl, int tid, const float4 (&v)[NI]) {
    const int k4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int o = (krow(tid) + (NT / 8) * i) * LDH + k4;
        *reinterpret_cast<uint2*>(sh + o) = make_uint2(pack_hi(v[i].x, v[i].y), pack_hi(v[i].z, v[i].w));
        *reinterpret_cast<uint2*>(sl + o) = make_uint2(pack_lo(v[i].x, v[i].y), pack_lo(v[i].z, v[i].w));
    }
}

template <int R, int NT = 256>
struct DenseKC : KCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    const float* ptr[NI];
    bool ok[NI];
    int krem;  // K - (kbase + k4): > 0 while this thread's float4 is inside K
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int K) {
        const int k4 = (tid & 7) * 4;
        krem = K - kbase - k4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + krow(tid) + (NT / 8) * i;
            ok[i] = r < o.rows;
            ptr[i] = o.p + (long long)r * o.ld + kbase + k4;
        }
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ld4_if(ptr[i], ok[i] && krem > 0, o.p);
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        krem -= BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) ptr[i] += BK;
    }
};

// conv gather, forward direction: rows = output pixels, k = tap*C + c, C % 32 == 0 (tap uniform per chunk)
template <int R, int NT = 256>
struct ConvKC : KCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    const float* base[NI];  // image base + k4
    int y0[NI], x0[NI];
    int kh, kw, c0;         // wave-uniform cursor
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int k4 = (tid & 7) * 4;
        const int chunk = kbase / BK;      // K order: channel block outer, tap inner (9 chunks share one input window)
        const int tap = chunk % 9;
        c0 = (chunk / 9) * BK;
        kh = tap / 3;
        kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + krow(tid) + (NT / 8) * i;
            int b, y, x;
            decode_pixel(o, r, b, y, x);
            base[i] = o.p + (long long)b * o.Hs * o.Ws * o.ld + k4;
            y0[i] = r < o.rows ? y * o.stride - o.pad_t : -(1 << 28);
            x0[i] = x * o.stride - o.pad_l;
        }
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int He = o.Hs << o.ups, We = o.Ws << o.ups;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ys = y0[i] + kh, xs = x0[i] + kw;
            const bool okk = (unsigned)ys < (unsigned)He && (unsigned)xs < (unsigned)We && c0 < o.C;
            const int off = ((ys >> o.ups) * o.Ws + (xs >> o.ups)) * (int)o.ld + c0;
            v[i] = ld4_if(base[i] + off, okk, o.p);
        }
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++kw == 3) {
            kw = 0;
            if (++kh == 3) { kh = 0; c0 += BK; }
        }
    }
};

// forward-conv weights W[co][tap][ci] (ld = 9*C) walked in ConvKC's K order (channel block outer, tap inner)
template <int R, int NT = 256>
struct WgtKC : KCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    const float* ptr[NI];
    bool ok[NI];
    int tap, c0;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int k4 = (tid & 7) * 4;
        const int chunk = kbase / BK;
        tap = chunk % 9;
        c0 = (chunk / 9) * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + krow(tid) + (NT / 8) * i;
            ok[i] = r < o.rows;
            ptr[i] = o.p + (long long)r * o.ld + k4;
        }
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int off = tap * o.C + c0;
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ld4_if(ptr[i] + off, ok[i] && c0 < o.C, o.p);
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++tap == 9) { tap = 0; c0 += BK; }
    }
};

// conv gather, data-gradient direction: rows = input pixels, k = tap*C + c over dY (C = Cout, C % 32 == 0)
template <int R, int NT = 256>
struct TConvKC : KCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    const float* base[NI];
    int y0[NI], x0[NI];
    int kh, kw, c0;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int k4 = (tid & 7) * 4;
        const int chunk = kbase / BK;      // K order: channel block outer, tap inner (9 chunks share one input window)
        const int tap = chunk % 9;
        c0 = (chunk / 9) * BK;
        kh = tap / 3;
        kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + krow(tid) + (NT / 8) * i;
            int b, y, x;
            decode_pixel(o, r, b, y, x);
            base[i] = o.p + (long long)b * o.Hs * o.Ws * o.ld + k4;
            y0[i] = r < o.rows ? y + o.pad_t : -(1 << 28);
            x0[i] = x + o.pad_l;
        }
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int sh = o.stride - 1;  // 0 or 1
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int yn = y0[i] - kh, xn = x0[i] - kw;
            const int ys = yn >> sh, xs = xn >> sh;
            const bool okk = yn >= 0 && xn >= 0 && ((yn | xn) & sh) == 0 && ys < o.Hs && xs < o.Ws && c0 < o.C;
            const int off = (ys * o.Ws + xs) * (int)o.ld + c0;
            v[i] = ld4_if(base[i] + off, okk, o.p);
        }
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++kw == 3) {
            kw = 0;
            if (++kh == 3) { kh = 0; c0 += BK; }
        }
    }
};

template <int R, int NT>
struct DenseRC : RCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr int KS = RCMap<R, NT>::KSTEP;
    static constexpr bool kKC = false;
    const float* ptr;  // p + row + (kbase + k0)*ld
    long long step;    // KS * ld
    int krem;          // K - (kbase + k0)
    bool ok;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int K) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
        ok = row0 + r4 < o.rows;
        ptr = o.p + (row0 + r4) + (long long)(kbase + k0) * o.ld;
        step = (long long)KS * o.ld;
        krem = K - kbase - k0;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ld4_if(ptr + step * i, ok && krem > KS * i, o.p);
    }
    __device__ __forceinline__ void advance(const Opnd& o) {
        ptr += (long long)BK * o.ld;
        krem -= BK;
    }
};

// weights seen from dgrad: rows = ci, k = tap*C + co (C = Cout, C % 32 == 0): W[(co*9 + tap)*ld + ci]
template <int R, int NT>
struct WgtRC : RCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr int KS = RCMap<R, NT>::KSTEP;
    static constexpr bool kKC = false;
    const float* ptr;  // p + row + k0*9*ld
    long long step;    // KS*9*ld
    int tap, c0;
    bool ok;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
        ok = row0 + r4 < o.rows;
        const int chunk = kbase / BK;      // same K order as TConvKC: channel block outer, tap inner
        tap = chunk % 9;
        c0 = (chunk / 9) * BK;
        ptr = o.p + (row0 + r4) + (long long)k0 * 9 * o.ld;
        step = (long long)KS * 9 * o.ld;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const float* q = ptr + ((long long)c0 * 9 + tap) * o.ld;
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ld4_if(q + step * i, ok && c0 < o.C, o.p);
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++tap == 9) { tap = 0; c0 += BK; }
    }
};

// ---- pre-split weights (bd_split_bf16, once per forward).  Blocked layout: the 32 k-values (KC) / 32 rows (RC) of a
// chunk are one 128-byte line, 64 B of bf16 hi then 64 B of lo, at split + 2*e for the fp32 element index e of the block
// start.  A thread moves 16 bytes (8 values of one plane) straight into the LDS plane: no VALU work for this operand,
// full-line global reads like the fp32 source, and the lane -> (row, plane) maps below make every 16-lane group of a
// ds_write_b128 sweep all 64 banks once.
__device__ __forceinline__ float4 ldb8_if(const unsigned short* p, bool ok) {   // 8 bf16 as raw bits in a float4
    const v4f t = *(gptr4)(ok ? (const void*)p : (const void*)kZero16);
    return make_float4(t.x, t.y, t.z, t.w);
}

// forward-conv weights, rows = co, K order of ConvKC (channel block outer, tap inner).  Per wave instruction: 8 rows x
// (hi, lo); quad q = lane/4: plane = (q>>2)&1, row = 4*(q&3) + (q>>3) (+2 for odd waves, +16 per wave pair, +32 per pass):
// a 16-lane group writes rows r, r+4, r+8, r+12 of one plane (row stride 80 B -> 4 x 64 B tile 256 B of banks).
template <int R, int NT = 256>
struct WgtKCs {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    const unsigned short* ptr[NI];   // split + 2*(row*ld) + plane*32 + k8
    bool ok[NI];
    int tap, c0;
    __device__ __forceinline__ static int row_of(int tid) {
        const int q = (tid & 63) >> 2, w = tid >> 6;
        return 16 * (w >> 1) + 2 * (w & 1) + 4 * (q & 3) + (q >> 3);
    }
    __device__ __forceinline__ static int plane_of(int tid) { return (tid >> 4) & 1; }
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int k8 = (tid & 3) * 8;
        const int chunk = kbase / BK;
        tap = chunk % 9;
        c0 = (chunk / 9) * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + row_of(tid) + (NT / 8) * i;
            ok[i] = r < o.rows;
            ptr[i] = o.split + 2 * ((long long)r * o.ld) + plane_of(tid) * 32 + k8;
        }
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int off = 2 * (tap * o.C + c0);
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ldb8_if(ptr[i] + off, ok[i] && c0 < o.C);
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++tap == 9) { tap = 0; c0 += BK; }
    }
    __device__ __forceinline__ static void store_split(unsigned short* sh, unsigned short* sl, int tid, const float4 (&v)[NI]) {
        unsigned short* pl = plane_of(tid) ? sl : sh;
        const int k8 = (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(pl + (row_of(tid) + (NT / 8) * i) * LDH + k8) = v[i];
    }
};

// weights seen from dgrad (rows = ci contiguous, k = (tap, co)), LDS image [k][R + 32] like RCStore.  A k row of the
// tile is R/32 blocks = R*4 contiguous bytes; lane l of a 32-lane half: plane = l>>4 (R = 128), block = (l>>2)&3,
// 8-row piece = l&3; the two halves of a wave take two k rows.  16 lanes = one plane, 256 contiguous LDS bytes.
template <int R, int NT = 256>
struct WgtRCs {
    static constexpr int NI = R * 8 / NT;
    static constexpr int NBLK = R / 32;            // 32-row blocks per k row
    static constexpr int LPK = 8 * NBLK;           // lanes per k row (hi + lo)
    static constexpr int KS = NT / LPK;            // k rows per pass
    static constexpr int NP = BK / KS;             // passes (== NI)
    static constexpr bool kKC = false;
    const unsigned short* ptr;
    long long step;
    int tap, c0;
    bool ok;
    __device__ __forceinline__ static int plane_of(int tid) { return ((tid % LPK) / (4 * NBLK)) & 1; }
    __device__ __forceinline__ static int r8_of(int tid) { return (((tid % LPK) >> 2) % NBLK) * 32 + (tid & 3) * 8; }
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        static_assert(NP == NI, "register budget");
        const int k0 = tid / LPK;
        ok = row0 + r8_of(tid) < o.rows;
        const int chunk = kbase / BK;
        tap = chunk % 9;
        c0 = (chunk / 9) * BK;
        const int blk = ((tid % LPK) >> 2) % NBLK;
        ptr = o.split + 2 * ((long long)row0 + (long long)k0 * 9 * o.ld) + blk * 64 + plane_of(tid) * 32 + (tid & 3) * 8;
        step = 2 * (long long)KS * 9 * o.ld;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const long long q = 2 * (((long long)c0 * 9 + tap) * o.ld);
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ldb8_if(ptr + q + step * i, ok && c0 < o.C);
    }
    __device__ __forceinline__ void advance(const Opnd&) {
        if (++tap == 9) { tap = 0; c0 += BK; }
    }
    __device__ __forceinline__ static void store_split(unsigned short* sh, unsigned short* sl, int tid, const float4 (&v)[NI]) {
        unsigned short* pl = plane_of(tid) ? sl : sh;
        const int k0 = tid / LPK;
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(pl + (k0 + KS * i) * (R + 32) + r8_of(tid)) = v[i];
    }
};

// conv gather, weight-gradient direction: rows = tap*C + ci (fixed per thread), k = output pixels
template <int R, int NT>
struct ConvRC : RCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr int KS = RCMap<R, NT>::KSTEP;
    static constexpr bool kKC = false;
    int kh, kw, ci, pix, Kp;
    bool ok, same;
    const float* ptr;      // "same" geometry: address of input pixel (pix + dtap), channel ci
    long long step, adv;   // KS * ld, BK * ld
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int K) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
        const int r = row0 + r4;
        const int tap = r / o.C;
        ci = r - tap * o.C;
        kh = tap / 3;
        kw = tap - kh * 3;
        ok = r < o.rows;
        pix = kbase + k0;
        Kp = K;
        // stride-1 "same" convolutions (all but the 6 resampling layers): the input pixel of output pixel p and this
        // thread's tap is p + const, so the address is one running pointer -- no integer multiplies in the loop
        same = o.stride == 1 && o.ups == 0 && o.Ho == o.Hs && o.Wo == o.Ws;
        ptr = o.p + ((long long)pix + (long long)(kh - o.pad_t) * o.Ws + (kw - o.pad_l)) * o.ld + ci;
        step = (long long)KS * o.ld;
        adv = (long long)BK * o.ld;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        if (same) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int p = pix + KS * i;
                int b, y, x;
                decode_pixel(o, p, b, y, x);
                const bool okk = ok && p < Kp && (unsigned)(y + kh - o.pad_t) < (unsigned)o.Hs && (unsigned)(x + kw - o.pad_l) < (unsigned)o.Ws;
                v[i] = ld4_if(ptr + step * i, okk, o.p);
            }
            return;
        }
        const int He = o.Hs << o.ups, We = o.Ws << o.ups;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = pix + KS * i;
            int b, y, x;
            decode_pixel(o, p, b, y, x);
            const int ys = y * o.stride - o.pad_t + kh, xs = x * o.stride - o.pad_l + kw;
            const bool okk = ok && p < Kp && (unsigned)ys < (unsigned)He && (unsigned)xs < (unsigned)We;
            const long long off = ((long long)b * o.Hs * o.Ws + (ys >> o.ups) * o.Ws + (xs >> o.ups)) * o.ld + ci;
            v[i] = ld4_if(o.p + off, okk, o.p);
        }
    }
    __device__ __forceinline__ void advance(const Opnd&) { pix += BK; ptr += adv; }
};

// ConvRC restricted to stride-1 "same" convolutions (output grid == input grid; all but the 6 resampling layers):
// the input pixel of output pixel p and this thread's tap is p + const, so the state is one running pointer, the pixel
// index and the tap offsets -- few enough registers that the 512-thread wgrad keeps two prefetch sets.
template <int R, int NT>
struct ConvRCs : RCStore<R, NT> {
    static constexpr int NI = RCMap<R, NT>::NI;
    static constexpr int KS = RCMap<R, NT>::KSTEP;
    static constexpr bool kKC = false;
    const float* ptr;   // address of input pixel (pix + dtap), channel ci
    int pix, dyx;       // output pixel of load 0; (kh - pad_t) in the high half, (kw - pad_l) in the low half (biased by 1)
    bool ok;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kbase, int) {
        const int r4 = RCMap<R, NT>::r4(tid), k0 = RCMap<R, NT>::kfirst(tid);
        const int r = row0 + r4;
        const int tap = r / o.C, ci = r - tap * o.C;
        const int kh = tap / 3, kw = tap - kh * 3;
        ok = r < o.rows;
        pix = kbase + k0;
        dyx = ((kh - o.pad_t + 1) << 8) | (kw - o.pad_l + 1);
        ptr = o.p + ((long long)pix + (long long)(kh - o.pad_t) * o.Ws + (kw - o.pad_l)) * o.ld + ci;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int dy = (dyx >> 8) - 1, dx = (dyx & 255) - 1;
        const long long step = (long long)KS * o.ld;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = pix + KS * i;
            int b, y, x;
            decode_pixel(o, p, b, y, x);
            const bool okk = ok && p < o.rows_k && (unsigned)(y + dy) < (unsigned)o.Hs && (unsigned)(x + dx) < (unsigned)o.Ws;
            v[i] = ld4_if(ptr + step * i, okk, o.p);
        }
    }
    __device__ __forceinline__ void advance(const Opnd& o) { pix += BK; ptr += (long long)BK * o.ld; }
};

// =================================================================================================
// GENERIC loaders (any kind / alignment; per-element address math).
// =================================================================================================
template <int R, int NT = 256>
struct GenericKC : KCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr bool kKC = true;
    long long base[NI];
    int yb[NI], xb[NI];
    int k4, kbase, K;
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kb, int K_) {
        k4 = (tid & 7) * 4; kbase = kb; K = K_;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = row0 + krow(tid) + (NT / 8) * i;
            const bool ok = r < o.rows;
            if (o.kind == BD_OPK_DENSE) {
                base[i] = (long long)r * o.ld;
                yb[i] = ok ? 0 : -(1 << 28);
                xb[i] = 0;
            } else {
                int b, y, x;
                decode_pixel(o, r, b, y, x);
                base[i] = (long long)b * o.Hs * o.Ws * o.ld;
                if (o.kind == BD_OPK_CONV) {
                    yb[i] = ok ? y * o.stride - o.pad_t : -(1 << 28);
                    xb[i] = x * o.stride - o.pad_l;
                } else {
                    yb[i] = ok ? y + o.pad_t : -(1 << 28);
                    xb[i] = x + o.pad_l;
                }
            }
        }
    }
    __device__ __forceinline__ long long addr(const Opnd& o, int i, int k) const {
        if (k >= K) return -1;
        if (o.kind == BD_OPK_DENSE) return yb[i] < 0 ? -1 : base[i] + k;
        const int tap = k / o.C;
        const int c = k - tap * o.C;
        const int kh = tap / 3, kw = tap - kh * 3;
        if (o.kind == BD_OPK_CONV) {
            const int ys = yb[i] + kh, xs = xb[i] + kw;
            if (ys < 0 || xs < 0 || ys >= (o.Hs << o.ups) || xs >= (o.Ws << o.ups)) return -1;
            return base[i] + ((long long)(ys >> o.ups) * o.Ws + (xs >> o.ups)) * o.ld + c;
        }
        int yn = yb[i] - kh, xn = xb[i] - kw;
        if (yn < 0 || xn < 0) return -1;
        if (o.stride == 2) {
            if ((yn | xn) & 1) return -1;
            yn >>= 1; xn >>= 1;
        }
        if (yn >= o.Hs || xn >= o.Ws) return -1;
        return base[i] + ((long long)yn * o.Ws + xn) * o.ld + c;
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
        const int k = kbase + k4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (o.vec) {
                const long long a = addr(o, i, k);
                v[i] = a >= 0 ? ld4(o.p + a) : zero4();
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long a = addr(o, i, k + j);
                    e[j] = a >= 0 ? o.p[a] : 0.f;
                }
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
    __device__ __forceinline__ void advance(const Opnd&) { kbase += BK; }
};

template <int R, int NT>
struct GenericRC : RCStore<R, NT> {
    static constexpr int NI = R * 8 / NT;
    static constexpr int KS = RCMap<R, NT>::KSTEP;
    static constexpr bool kKC = false;
    int k0, row, kbase, K;
    int kh[4], kw[4], ci[4];
    bool rok[4];
    __device__ __forceinline__ void init(const Opnd& o, int row0, int tid, int kb, int K_) {
        const int r4 = RCMap<R, NT>::r4(tid);
        k0 = RCMap<R, NT>::kfirst(tid);
        row = row0 + r4; kbase = kb; K = K_;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = row + j;
            rok[j] = r < o.rows;
            if (o.kind == BD_OPK_CONV) {
                const int tap = r / o.C;
                ci[j] = r - tap * o.C;
                kh[j] = tap / 3;
                kw[j] = tap - kh[j] * 3;
                if (tap >= 9) rok[j] = false;
            } else {
                ci[j] = r;
                kh[j] = kw[j] = 0;
            }
        }
    }
    __device__ __forceinline__ long long addr(const Opnd& o, int j, int k) const {
        if (k >= K || !rok[j]) return -1;
        if (o.kind == BD_OPK_DENSE) return (long long)k * o.ld + ci[j];
        if (o.kind == BD_OPK_WGT) {
            const int tap = k / o.C;
            const int co = k - tap * o.C;
            return ((long long)co * 9 + tap) * o.ld + ci[j];
        }
        int b, y, x;
        decode_pixel(o, k, b, y, x);
        const int ys = y * o.stride - o.pad_t + kh[j], xs = x * o.stride - o.pad_l + kw[j];
        if (ys < 0 || xs < 0 || ys >= (o.Hs << o.ups) || xs >= (o.Ws << o.ups)) return -1;
        return ((long long)b * o.Hs * o.Ws + (long long)(ys >> o.ups) * o.Ws + (xs >> o.ups)) * o.ld + ci[j];
    }
    __device__ __forceinline__ void load(const Opnd& o, float4 (&v)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = kbase + k0 + KS * i;
            if (o.vec) {
                const long long a = addr(o, 0, k);
                v[i] = a >= 0 ? ld4(o.p + a) : zero4();
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long a = addr(o, j, k);
                    e[j] = a >= 0 ? o.p[a] : 0.f;
                }
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
    __device__ __forceinline__ void advance(const Opnd&) { kbase += BK; }
};

template <int R, bool KC>
constexpr int lds_floats() {
    return KC ? R * LDK : BK * (R + 4);
}

// ---- workgroup -> (tile, batch/split) -----------------------------------------------------------------------------
// 1-D grid over tiles x batch x ksplit.  Workgroup b runs on XCD b % 8 (cdna guide T1; speed only, nothing depends on
// it), and each XCD has its own L2: hand every XCD one contiguous run of the logical order (n fastest, then m, then
// split / batch), so the tiles that re-read the same A panel -- and, with split-K, the same k-slice of both operands --
// meet in one L2 instead of being fetched by all eight.
struct WgCoord {
    int tm, tn, zz;
};
__device__ __forceinline__ WgCoord wg_coord(const IGemmParams& p) {
    const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
    const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
    const unsigned ntiles = p.tiles_m * p.tiles_n;
    const unsigned zz = j / ntiles, tile = j - zz * ntiles;
    WgCoord w;
    w.tm = tile / p.tiles_n;
    w.tn = tile - w.tm * p.tiles_n;
    w.zz = zz;
    return w;
}

// ---- epilogue: lane holds column (n) li of rows (r&3) + 8*(r>>2) + 4*h of every 32x32 tile ---------------------
// predicated scalar load without control flow (see ld4_if): masked-off lanes read a zero from the code object
typedef const float __attribute__((address_space(1))) * gptr1;
__device__ __forceinline__ float ld1_if(const float* p, bool ok) { return *(gptr1)(ok ? p : kZero16); }

// All optional addends (row bias, residual, previous C) of a 32x32 tile are LOADED FIRST -- 16 branch-free loads each, all in
// flight together -- and only then combined and stored.  The straightforward form (per element: `if (residual) v += *r;
// if (accumulate) v += *dst; *dst = v`) makes hipcc branch around every load and wait vmcnt(0) behind it: 32-64 serialised
// memory round trips per thread, which dominated the short-K GEMMs that carry a residual or accumulate (guide section 5 (c)).
// Absent addends contribute an exact + 0.0f, so the result is bit-identical to the per-element form.
// FULL (wave-uniform: the wave's whole sub-tile lies inside M x N) drops every per-element predicate, so the 16 stores of
// a tile are straight-line code with their own address registers; behind an exec-masked branch per element hipcc reuses
// one address register pair and waits vmcnt(0) for the previous store before every store.
template <int TM, int TN, bool FULL>
__device__ __forceinline__ void epilogue_impl(const IGemmParams& p, floatx16 (&acc)[TM][TN], int mw, int nw, int li, int h, int bo, int bi, int zz) {
    const long long coff = bo * p.c_bso + bi * p.c_bsi;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int n = nw + q * 32 + li;
            const bool nok = FULL || n < p.N;
            const int m_base = mw + i * 32 + 4 * h;
            if (p.ksplit > 1) {   // (kernel-uniform) raw partial tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    if (FULL || (nok && m < p.M)) p.partial[((long long)zz * p.M + m) * p.N + n] = acc[i][q][r];
                }
                continue;
            }
            const float bn = p.bias ? ld1_if(p.bias + n, nok) : 0.f;
            float rb[16], rs[16], pc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { rb[r] = 0.f; rs[r] = 0.f; pc[r] = 0.f; }
            if (p.rowbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    rb[r] = ld1_if(p.rowbias + (long long)(m / p.rows_per_group) * p.ld_rowbias + n, FULL || (nok && m < p.M));
                }
            }
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    rs[r] = ld1_if(p.residual + coff + (long long)m * p.ldr + n, FULL || (nok && m < p.M));
                }
            }
            if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    pc[r] = ld1_if(p.C + coff + (long long)m * p.ldc + n, FULL || (nok && m < p.M));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (r & 3) + 8 * (r >> 2);
                float v = acc[i][q][r] * p.alpha + bn;
                v += rb[r];
                v += rs[r];
                v *= p.out_scale;
                v += pc[r];
                if (FULL || (nok && m < p.M)) p.C[coff + (long long)m * p.ldc + n] = v;
            }
        }
}
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const IGemmParams& p, floatx16 (&acc)[TM][TN], int mw, int nw, int li, int h, int bo, int bi, int zz) {
    if (mw + TM * 32 <= p.M && nw + TN * 32 <= p.N) epilogue_impl<TM, TN, true>(p, acc, mw, nw, li, h, bo, bi, zz);
    else epilogue_impl<TM, TN, false>(p, acc, mw, nw, li, h, bo, bi, zz);
}

// ---- fused row sums of the A operand (bias gradients: A = dY^T of a wgrad) --------------------------------------------
// Every RC A value passes through the registers of exactly one thread of every n-tile's workgroup; the tn == 0
// workgroups add theirs up (thread-fixed row quad r4, k phases kfirst + KSTEP*i), fold the k phases through LDS in a
// fixed order and emit out[m] (or one split-K partial row).  No extra memory traffic, deterministic.
template <int BM, int NT>
__device__ __forceinline__ void colsum_tail(const IGemmParams& p, float* red, float4 cs, int tid, int m0, int zz) {
    constexpr int NPH = NT * 4 / BM;   // k phases = NT threads / (BM/4) row quads
    *reinterpret_cast<float4*>(red + RCMap<BM, NT>::kfirst(tid) * BM + RCMap<BM, NT>::r4(tid)) = cs;
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
        float v = 0.f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) v += red[ph * BM + tid];
        if (p.ksplit > 1) p.partial[(long long)p.ksplit * p.M * p.N + (long long)zz * p.M + m0 + tid] = v;
        else p.a_colsum[m0 + tid] = v;
    }
}
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <int BM, int BN, class LA, class LB>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmParams p) {
    constexpr bool A_KC = LA::kKC, B_KC = LB::kKC;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float sA[lds_floats<BM, A_KC>()];
    __shared__ __attribute__((aligned(16))) float sB[lds_floats<BN, B_KC>()];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    // tile coordinates: n fastest so neighbouring workgroups share the A (activation) panel
    // XCD-aware order (cdna guide T1): workgroup b runs on XCD b % 8, so give every XCD a contiguous run of
    // logical tiles -- neighbours (same A panel, next n; adjacent m) then share that XCD's L2.  Speed only.
    const WgCoord wg = wg_coord(p);
    const int m0 = wg.tm * BM, n0 = wg.tn * BN;
    const int bz = wg.zz / p.ksplit, ks = wg.zz - bz * p.ksplit;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;

    Opnd A = p.A, B = p.B;
    A.p += bo * p.a_bso + bi * p.a_bsi;
    B.p += bo * p.b_bso + bi * p.b_bsi;

    const int nchunks_total = (p.K + BK - 1) / BK;
    const int c_begin = ks * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;

    LA la;
    LB lb;
    la.init(A, m0, tid, c_begin * BK, p.K);
    lb.init(B, n0, tid, c_begin * BK, p.K);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM / 32], rb[BN / 32];
    if (c_begin < c_end) {
        la.load(A, ra);
        lb.load(B, rb);
    }
    const bool do_cs = !A_KC && p.a_colsum != nullptr && wg.tn == 0;
    float4 cs = zero4();
    for (int c = c_begin; c < c_end; ++c) {
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) add4(cs, ra[i]);
        }
        LA::store(sA, tid, ra);
        LB::store(sB, tid, rb);
        __syncthreads();
        if (c + 1 < c_end) {  // prefetch the next chunk into registers while this one is computed
            la.advance(A);
            lb.advance(B);
            la.load(A, ra);
            lb.load(B, rb);
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * WM + i * 32 + li;
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(sA + r * LDK + g * 8 + 4 * h);
                    fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[i][j] = sA[(g * 8 + 4 * h + j) * (BM + 4) + r];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * WN + i * 32 + li;
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(sB + r * LDK + g * 8 + 4 * h);
                    fb[i][0] = v.x; fb[i][1] = v.y; fb[i][2] = v.z; fb[i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[i][j] = sB[(g * 8 + 4 * h + j) * (BN + 4) + r];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < TN; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[q][j], acc[i][q], 0, 0, 0);
        }
        __syncthreads();
    }

    if (do_cs) colsum_tail<BM, 256>(p, sA, cs, tid, m0, wg.zz);
    epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, li, h, bo, bi, wg.zz);
}

// ------------------------------------------------------------------------------------------------
// split-bf16 kernel: same tiling / staging, but the register -> LDS store splits every fp32 value ONCE into
// hi + lo bf16 planes ([rows][32 + 8] bf16 each, RC operands transposed in registers on the way), and the
// K loop issues hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) from ds_read_b128 fragments.
template <int BM, int BN, class LA, class LB, int NT, int DEPTH>
__global__ __launch_bounds__(NT, NT / 128) void igemm_bf16x3_kernel(IGemmParams p) {   // (threads, min waves per SIMD)
    // waves: (NT/128) x 2 over the tile; 256 threads -> 64x64 per wave (2 workgroups = 2 waves per SIMD), 512 threads ->
    // 32x64 per wave: half the accumulators and half the staging registers per thread, 4 waves per SIMD
    constexpr int WAVES_M = NT / 128;
    constexpr int WM = BM / WAVES_M, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr bool A_KC = LA::kKC, B_KC = LB::kKC;
    constexpr int A_SZ = A_KC ? BM * LDH : BK * (BM + 32), B_SZ = B_KC ? BN * LDH : BK * (BN + 32);
    __shared__ __attribute__((aligned(16))) unsigned short sAh[A_SZ];
    __shared__ __attribute__((aligned(16))) unsigned short sAl[A_SZ];
    __shared__ __attribute__((aligned(16))) unsigned short sBh[B_SZ];
    __shared__ __attribute__((aligned(16))) unsigned short sBl[B_SZ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    const WgCoord wg = wg_coord(p);
    const int m0 = wg.tm * BM, n0 = wg.tn * BN;
    const int bz = wg.zz / p.ksplit, ks = wg.zz - bz * p.ksplit;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;

    Opnd A = p.A, B = p.B;
    A.p += bo * p.a_bso + bi * p.a_bsi;
    B.p += bo * p.b_bso + bi * p.b_bsi;

    const int nchunks_total = (p.K + BK - 1) / BK;
    const int c_begin = ks * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;

    LA la;
    LB lb;
    la.init(A, m0, tid, c_begin * BK, p.K);
    lb.init(B, n0, tid, c_begin * BK, p.K);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register prefetch two chunks deep: ra0/rb0 hold chunk c, ra1/rb1 chunk c+1; after a pair has been split into LDS it
    // is refilled with chunk c+2, whose loads then have two MFMA sections to land.  The loads are unconditional
    // (predicated by address, ld4_if) and there is no branch in the loop body, so the compiler waits with exact
    // vmcnt(N) instead of vmcnt(0).  Chunks past c_end are loaded and never used (in-range addresses or masked off).
    // (DEPTH == 1: a single register set, refilled with chunk c+1 -- for the operand pairs whose address state does not
    // leave room for two sets under the 128-VGPR budget of the 512-thread form.)
    float4 ra0[LA::NI], rb0[LB::NI], ra1[DEPTH == 2 ? LA::NI : 1], rb1[DEPTH == 2 ? LB::NI : 1];
    la.load(A, ra0);
    lb.load(B, rb0);
    if constexpr (DEPTH == 2) {
        la.advance(A);
        lb.advance(B);
        la.load(A, ra1);
        lb.load(B, rb1);
    }
    const bool do_cs = !A_KC && p.a_colsum != nullptr && wg.tn == 0;
    float4 cs = zero4();
    auto step = [&](float4 (&ua)[LA::NI], float4 (&ub)[LB::NI]) {
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < LA::NI; ++i) add4(cs, ua[i]);
        }
        LA::store_split(sAh, sAl, tid, ua);
        LB::store_split(sBh, sBl, tid, ub);
        __syncthreads();
        la.advance(A);
        lb.advance(B);
        la.load(A, ua);
        lb.load(B, ub);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (A_KC) {
                    const int o = (wm * WM + i * 32 + li) * LDH + 16 * s + 8 * h;
                    ah[i] = *reinterpret_cast<const bf16x8*>(sAh + o);
                    al[i] = *reinterpret_cast<const bf16x8*>(sAl + o);
                } else {
                    ah[i] = rc_frag<BM + 32>(sAh, wm * WM + i * 32, 16 * s, lane);
                    al[i] = rc_frag<BM + 32>(sAl, wm * WM + i * 32, 16 * s, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if (B_KC) {
                    const int o = (wn * WN + i * 32 + li) * LDH + 16 * s + 8 * h;
                    bh[i] = *reinterpret_cast<const bf16x8*>(sBh + o);
                    bl[i] = *reinterpret_cast<const bf16x8*>(sBl + o);
                } else {
                    bh[i] = rc_frag<BN + 32>(sBh, wn * WN + i * 32, 16 * s, lane);
                    bl[i] = rc_frag<BN + 32>(sBl, wn * WN + i * 32, 16 * s, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < TN; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[q], acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < TN; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[q], acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < TN; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[q], acc[i][q], 0, 0, 0);
        }
        __syncthreads();
    };
    if constexpr (DEPTH == 2) {
        int c = c_begin;
        for (; c + 1 < c_end; c += 2) {
            step(ra0, rb0);
            step(ra1, rb1);
        }
        if (c < c_end) step(ra0, rb0);
    } else {
        for (int c = c_begin; c < c_end; ++c) step(ra0, rb0);
    }
    if (do_cs) colsum_tail<BM, NT>(p, reinterpret_cast<float*>(sAh), cs, tid, m0, wg.zz);
    epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, li, h, bo, bi, wg.zz);
}

// the on-the-fly split, applied once to a whole buffer (weights): 8 elements per thread, blocked output layout
// (element e: hi at (e/32)*64 + e%32, lo 32 further)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, long long n8, unsigned short* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4 a = ld4(src + 8 * i), b = ld4(src + 8 * i + 4);
        unsigned short* o = out + (i >> 2) * 64 + (i & 3) * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(pack_hi(a.x, a.y), pack_hi(a.z, a.w), pack_hi(b.x, b.y), pack_hi(b.z, b.w));
        *reinterpret_cast<uint4*>(o + 32) = make_uint4(pack_lo(a.x, a.y), pack_lo(a.z, a.w), pack_lo(b.x, b.y), pack_lo(b.z, b.w));
    }
}

// split-K second pass: fixed-order (deterministic) sum of the partial slabs + epilogue
__global__ __launch_bounds__(256) void igemm_splitk_reduce(IGemmParams p, int nbatch) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)p.M * p.N;
    if (idx >= per * nbatch) {   // tail threads: the fused A column sums (batch 1 only)
        const long long m = idx - per * nbatch;
        if (p.a_colsum && m < p.M) {
            float v = 0.f;
            for (int s = 0; s < p.ksplit; ++s) v += p.partial[(long long)p.ksplit * per + (long long)s * p.M + m];
            p.a_colsum[m] = v;
        }
        return;
    }
    const int bz = (int)(idx / per);
    const long long mn = idx - (long long)bz * per;
    const int m = (int)(mn / p.N), n = (int)(mn - (long long)m * p.N);
    float v = 0.f;
#pragma unroll 4      // four slab loads in flight; the additions keep their fixed order
    for (int s = 0; s < p.ksplit; ++s) v += p.partial[((long long)(bz * p.ksplit + s)) * per + mn];
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const long long coff = bo * p.c_bso + bi * p.c_bsi;
    v *= p.alpha;
    if (p.bias) v += p.bias[n];
    if (p.rowbias) v += p.rowbias[(long long)(m / p.rows_per_group) * p.ld_rowbias + n];
    if (p.residual) v += p.residual[coff + (long long)m * p.ldr + n];
    v *= p.out_scale;
    float* dst = p.C + coff + (long long)m * p.ldc + n;
    if (p.accumulate) v += *dst;
    *dst = v;
}

// the same pass on float4 slab loads.  LANES = 8: eight adjacent lanes share one output float4, each sums every eighth
// slab and a butterfly folds them -- few outputs times many slabs (the weight-gradient GEMMs: 50-200 K outputs, 32-128
// slabs) otherwise leave most of the chip idle behind one thread's serial loads.  Fixed order either way (deterministic).
// Needs N, ldc, ldr and the batch strides of C to be multiples of 4 and C 16-byte aligned.
template <int LANES>
__global__ __launch_bounds__(256) void igemm_splitk_reduce4(IGemmParams p, int nbatch) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long idx = gid / LANES;
    const int sub = (int)(gid - idx * LANES);
    const long long per = (long long)p.M * p.N, per4 = per >> 2;
    if (idx >= per4 * nbatch) {   // tail: the fused A column sums (batch 1 only), one lane group per row
        const long long m = idx - per4 * nbatch;
        if (p.a_colsum && m < p.M) {
            float v = 0.f;
            for (int s = sub; s < p.ksplit; s += LANES) v += p.partial[(long long)p.ksplit * per + (long long)s * p.M + m];
            for (int off = 1; off < LANES; off <<= 1) v += __shfl_xor(v, off);
            if (sub == 0) p.a_colsum[m] = v;
        }
        return;
    }
    const int bz = (int)(idx / per4);
    const long long mn = (idx - (long long)bz * per4) << 2;
    const int m = (int)(mn / p.N), n = (int)(mn - (long long)m * p.N);
    const float* src = p.partial + (long long)bz * p.ksplit * per + mn;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = sub; s < p.ksplit; s += LANES) {
        const float4 b = *reinterpret_cast<const float4*>(src + (long long)s * per);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    for (int off = 1; off < LANES; off <<= 1) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    if (sub != 0) return;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const long long coff = bo * p.c_bso + bi * p.c_bsi;
    float o[4] = {v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += p.bias[n + j];
    }
    if (p.rowbias) {
        const float* rb = p.rowbias + (long long)(m / p.rows_per_group) * p.ld_rowbias + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += rb[j];
    }
    if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + coff + (long long)m * p.ldr + n);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] *= p.out_scale;
    float4* dst = reinterpret_cast<float4*>(p.C + coff + (long long)m * p.ldc + n);
    if (p.accumulate) {
        const float4 e = *dst;
        o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
    }
    *dst = make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------
static bool operand_vec_ok(const bd_operand& o, int rows, int K) {
    if (!aligned16(o.p) || (o.ld & 3) || (o.bs_outer & 3) || (o.bs_inner & 3)) return false;
    if (o.kc) {
        if (o.kind == BD_OPK_DENSE) return (K & 3) == 0;
        return (o.C & 3) == 0;  // CONV / TCONV: a float4 never straddles a tap
    }
    if (o.kind == BD_OPK_CONV) return (o.C & 3) == 0;
    return (rows & 3) == 0;
}
// shape conditions of the FAST loader of this operand
static bool operand_fast_ok(const bd_operand& o, int rows, int K) {
    if (!operand_vec_ok(o, rows, K)) return false;
    switch (o.kind) {
        case BD_OPK_DENSE: return true;
        case BD_OPK_CONV:
            if (o.kc) return o.C % BK == 0 && (long long)o.Hs * o.Ws * o.ld < (1ll << 31);
            return true;
        case BD_OPK_TCONV: return o.C % BK == 0 && (long long)o.Hs * o.Ws * o.ld < (1ll << 31);
        case BD_OPK_WGT: return o.C % BK == 0;
    }
    return false;
}

static int validate_operand(const bd_operand& o, const char* which) {
    BD_CHECK(o.p != nullptr, BD_ERR_INVALID, "igemm: operand %s is null", which);
    if (o.kc) {
        BD_CHECK(o.kind == BD_OPK_DENSE || o.kind == BD_OPK_CONV || o.kind == BD_OPK_TCONV, BD_ERR_INVALID,
                 "igemm: operand %s: kind %d not valid for KC storage", which, o.kind);
    } else {
        BD_CHECK(o.kind == BD_OPK_DENSE || o.kind == BD_OPK_CONV || o.kind == BD_OPK_WGT, BD_ERR_INVALID,
                 "igemm: operand %s: kind %d not valid for RC storage", which, o.kind);
    }
    if (o.kind != BD_OPK_DENSE) {
        BD_CHECK(o.C > 0, BD_ERR_INVALID, "igemm: operand %s: C must be > 0", which);
        if (o.kind != BD_OPK_WGT) {
            BD_CHECK(o.Hs > 0 && o.Ws > 0 && o.Ho > 0 && o.Wo > 0, BD_ERR_INVALID, "igemm: operand %s: bad conv geometry", which);
            BD_CHECK(o.stride == 1 || o.stride == 2, BD_ERR_UNSUPPORTED, "igemm: stride %d unsupported", o.stride);
            BD_CHECK(o.ups == 0 || o.ups == 1, BD_ERR_UNSUPPORTED, "igemm: ups %d unsupported", o.ups);
            BD_CHECK(!(o.ups && o.kind == BD_OPK_TCONV), BD_ERR_UNSUPPORTED, "igemm: TCONV with ups");
        }
    }
    return BD_OK;
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

static Opnd make_opnd(const bd_operand& o, int rows, int K) {
    Opnd r;
    r.p = o.p; r.split = o.split; r.ld = o.ld; r.kind = o.kind; r.vec = operand_vec_ok(o, rows, K) ? 1 : 0;
    r.C = o.C > 0 ? o.C : 1; r.Hs = o.Hs; r.Ws = o.Ws; r.Ho = o.Ho > 0 ? o.Ho : 1; r.Wo = o.Wo > 0 ? o.Wo : 1;
    r.stride = o.stride > 0 ? o.stride : 1; r.pad_t = o.pad_t; r.pad_l = o.pad_l; r.ups = o.ups; r.rows = rows; r.rows_k = K;
    const int lw = ilog2_exact(r.Wo), lh = ilog2_exact(r.Ho);
    r.lw = (lw >= 0 && lh >= 0) ? lw : -1;
    r.lhw = (lw >= 0 && lh >= 0) ? lw + lh : -1;
    return r;
}

struct Choice {
    int tile, ksplit, cps;
};

static Choice choose(const bd_igemm_desc& d) {
    Choice c;
    const int nb = d.batch_outer * d.batch_inner;
    auto tiles = [&](int t) { return cdiv(d.M, t) * cdiv(d.N, t) * nb; };
    // 128x128 tiles (32 flop per staged byte) whenever both dims allow; split K to put ~1.5 workgroups on each CU
    c.tile = d.tile ? d.tile : ((d.M >= 128 && d.N >= 128) ? 128 : 64);
    const int nchunks = (int)cdiv(d.K, BK);
    int ks = d.ksplit;
    if (ks <= 0) {
        long long t = tiles(c.tile);
        ks = 1;
        if (t < 256) {
            // ~1.5 workgroups per CU, never more than the budget (rounding the split count UP would spill a few
            // workgroups into a second, nearly empty round).  Sweep of the budget 384 / 448 / 512 / 640 on the CIFAR
            // step: 25.07 / 25.15 / 25.16 / 25.55 ms -- beyond ~1.5 per CU the extra partial tiles cost more than the
            // fuller chip gains (the second stream fills idle CUs anyway).
            static const int slots = [] {
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                return 3 * cus / 2;
            }();
            ks = (int)(slots / t);
            if (ks < 1) ks = 1;
            int maxks = nchunks / 8;
            if (maxks < 1) maxks = 1;
            if (ks > maxks) ks = maxks;
            if (ks > 128) ks = 128;
        }
    }
    if (ks > nchunks) ks = nchunks;
    if (ks < 1) ks = 1;
    c.cps = (int)cdiv(nchunks, ks);
    c.ksplit = (int)cdiv(nchunks, c.cps);  // no empty splits
    return c;
}

size_t igemm_workspace_bytes(const bd_igemm_desc& d) {
    Choice c = choose(d);
    if (c.ksplit <= 1) return 0;
    return (size_t)d.batch_outer * d.batch_inner * c.ksplit * ((size_t)d.M * d.N + (d.a_colsum ? (size_t)d.M : 0)) * sizeof(float);
}

enum Cls { CLS_GENERIC = 0, CLS_CONV_FWD, CLS_CONV_DGRAD, CLS_CONV_WGRAD, CLS_GEMM_NT, CLS_GEMM_NN, CLS_GEMM_TN,
           CLS_CONV_FWD_WS, CLS_CONV_DGRAD_WS,     // _WS: weights (B) taken from pre-split bf16 planes
           CLS_CONV_WGRAD_SAME };                  // stride-1 same-size geometry: slim gather loader
static const char* kClsName[] = {"generic", "conv_fwd", "conv_dgrad", "conv_wgrad", "gemm_nt", "gemm_nn", "gemm_tn",
                                 "conv_fwd", "conv_dgrad", "conv_wgrad"};

static bool presplit_ok(const bd_igemm_desc& d) {
    const bd_operand& o = d.B;
    // whole 32-element blocks: block-aligned base (the caller passes split + 2*e for a base element e % 32 == 0),
    // ld % 32 == 0 (rows start on blocks), channel count % 32 == 0, N % 32 == 0 (RC rows come in blocks)
    return d.mode == BD_MODE_BF16X3 && o.split && ((uintptr_t)o.split & 127) == 0 && (o.ld & 31) == 0 && (o.C & 31) == 0 &&
           (d.N & 31) == 0 && d.batch_outer * d.batch_inner == 1;
}

static Cls classify(const bd_igemm_desc& d, bool fast) {
    if (!fast) return CLS_GENERIC;
    const int ak = d.A.kind, bk = d.B.kind;
    const bool akc = d.A.kc != 0, bkc = d.B.kc != 0;
    if (akc && bkc) {
        if (ak == BD_OPK_CONV && bk == BD_OPK_DENSE) return presplit_ok(d) ? CLS_CONV_FWD_WS : CLS_CONV_FWD;
        if (ak == BD_OPK_DENSE && bk == BD_OPK_DENSE) return CLS_GEMM_NT;
    } else if (akc && !bkc) {
        if (ak == BD_OPK_TCONV && bk == BD_OPK_WGT) return presplit_ok(d) ? CLS_CONV_DGRAD_WS : CLS_CONV_DGRAD;
        if (ak == BD_OPK_DENSE && bk == BD_OPK_DENSE) return CLS_GEMM_NN;
    } else if (!akc && !bkc) {
        if (ak == BD_OPK_DENSE && bk == BD_OPK_CONV) {
            const bd_operand& o = d.B;
            const bool same = o.stride == 1 && o.ups == 0 && o.Ho == o.Hs && o.Wo == o.Ws && o.pad_t <= 1 && o.pad_l <= 1;
            return same ? CLS_CONV_WGRAD_SAME : CLS_CONV_WGRAD;
        }
        if (ak == BD_OPK_DENSE && bk == BD_OPK_DENSE) return CLS_GEMM_TN;
    }
    return CLS_GENERIC;
}

// TR = true selects the split-bf16 kernel (RC operands keep the coalesced thread map; transposition happens in the
// ds_read_b64_tr_b16 fragment reads)
template <int T, bool TR, class LA, class LB, int NT = 256, int DEPTH = 2>
static void launch1(const IGemmParams& p, dim3 grid, hipStream_t st) {
    if constexpr (TR) hipLaunchKernelGGL((igemm_bf16x3_kernel<T, T, LA, LB, NT, DEPTH>), grid, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<T, T, LA, LB>), grid, dim3(256), 0, st, p);
}

// fast classes; NT = threads per workgroup (512 only for the split-bf16 128x128 tiles: 8 waves of 32x64)
template <int T, bool TR, int NT>
static bool launch_fast(const IGemmParams& p, Cls cls, dim3 grid, hipStream_t st) {
    switch (cls) {
        case CLS_CONV_FWD: launch1<T, TR, ConvKC<T, NT>, WgtKC<T, NT>, NT>(p, grid, st); return true;
        case CLS_GEMM_NT: launch1<T, TR, DenseKC<T, NT>, DenseKC<T, NT>, NT>(p, grid, st); return true;
        case CLS_CONV_DGRAD: launch1<T, TR, TConvKC<T, NT>, WgtRC<T, NT>, NT>(p, grid, st); return true;
        case CLS_GEMM_NN: launch1<T, TR, DenseKC<T, NT>, DenseRC<T, NT>, NT, (NT == 512 ? 1 : 2)>(p, grid, st); return true;
        case CLS_CONV_WGRAD: launch1<T, TR, DenseRC<T, NT>, ConvRC<T, NT>, NT, (NT == 512 ? 1 : 2)>(p, grid, st); return true;
        case CLS_CONV_WGRAD_SAME: launch1<T, TR, DenseRC<T, NT>, ConvRCs<T, NT>, NT, (NT == 512 ? 1 : 2)>(p, grid, st); return true;
        case CLS_GEMM_TN: launch1<T, TR, DenseRC<T, NT>, DenseRC<T, NT>, NT, (NT == 512 ? 1 : 2)>(p, grid, st); return true;
        case CLS_CONV_FWD_WS:
            if constexpr (TR) { launch1<T, true, ConvKC<T, NT>, WgtKCs<T, NT>, NT>(p, grid, st); return true; }
            return false;
        case CLS_CONV_DGRAD_WS:
            if constexpr (TR) { launch1<T, true, TConvKC<T, NT>, WgtRCs<T, NT>, NT>(p, grid, st); return true; }
            return false;
        default: return false;
    }
}

template <int T, bool TR>
static void launch_tile(const IGemmParams& p, const bd_igemm_desc& d, Cls cls, dim3 grid, hipStream_t st) {
    // split-bf16 128x128 tiles run with 512 threads (8 waves of 32x64, 4 waves per SIMD): half the accumulators and
    // staging registers per thread buys the occupancy that hides the split / staging work (conv fwd +7 %, dgrad +9 %,
    // K=256 GEMMs +20..30 %).  The conv wgrad (both operands row-contiguous, the largest address state) fits 128 VGPRs
    // only with a single prefetch register set (DEPTH 1).  BD_IGEMM_NT=256 forces the narrow form everywhere.
    static const bool narrow = getenv("BD_IGEMM_NT") && atoi(getenv("BD_IGEMM_NT")) == 256;
    if constexpr (TR && T == 128) {
        if (!narrow && launch_fast<T, TR, 512>(p, cls, grid, st)) return;
    }
    if (launch_fast<T, TR, 256>(p, cls, grid, st)) return;
    const bool akc = d.A.kc != 0, bkc = d.B.kc != 0;
    if (akc && bkc) launch1<T, TR, GenericKC<T>, GenericKC<T>>(p, grid, st);
    else if (akc && !bkc) launch1<T, TR, GenericKC<T>, GenericRC<T, 256>>(p, grid, st);
    else if (!akc && !bkc) launch1<T, TR, GenericRC<T, 256>, GenericRC<T, 256>>(p, grid, st);
    else launch1<T, TR, GenericRC<T, 256>, GenericKC<T>>(p, grid, st);
}

int igemm_launch(const bd_igemm_desc& d, hipStream_t stream) {
    BD_CHECK(d.M > 0 && d.N > 0 && d.K > 0, BD_ERR_INVALID, "igemm: M,N,K must be > 0 (got %d,%d,%d)", d.M, d.N, d.K);
    BD_CHECK(d.batch_outer >= 1 && d.batch_inner >= 1, BD_ERR_INVALID, "igemm: batch counts must be >= 1");
    BD_CHECK(d.C != nullptr, BD_ERR_INVALID, "igemm: C is null");
    BD_CHECK(d.tile == 0 || d.tile == 64 || d.tile == 128, BD_ERR_INVALID, "igemm: tile must be 0, 64 or 128");
    BD_CHECK(d.mode == BD_MODE_F32 || d.mode == BD_MODE_BF16X3, BD_ERR_INVALID, "igemm: unknown compute mode %d", d.mode);
    BD_TRY(validate_operand(d.A, "A"));
    BD_TRY(validate_operand(d.B, "B"));
    BD_CHECK(!(d.rowbias && d.rows_per_group <= 0), BD_ERR_INVALID, "igemm: rowbias needs rows_per_group > 0");

    Choice c = choose(d);
    IGemmParams p;
    p.A = make_opnd(d.A, d.M, d.K);
    p.B = make_opnd(d.B, d.N, d.K);
    p.M = d.M; p.N = d.N; p.K = d.K;
    p.tiles_m = (int)cdiv(d.M, c.tile); p.tiles_n = (int)cdiv(d.N, c.tile);
    p.batch_inner = d.batch_inner; p.ksplit = c.ksplit; p.chunks_per_split = c.cps;
    p.a_bso = d.A.bs_outer; p.a_bsi = d.A.bs_inner; p.b_bso = d.B.bs_outer; p.b_bsi = d.B.bs_inner;
    p.c_bso = d.c_bs_outer; p.c_bsi = d.c_bs_inner;
    p.C = d.C; p.ldc = d.ldc; p.alpha = d.alpha; p.out_scale = d.out_scale;
    p.bias = d.bias; p.rowbias = d.rowbias; p.ld_rowbias = d.ld_rowbias;
    p.rows_per_group = d.rows_per_group > 0 ? d.rows_per_group : 1;
    p.residual = d.residual; p.ldr = d.ldr; p.accumulate = d.accumulate;
    p.partial = nullptr;
    p.a_colsum = d.a_colsum;
    const int nb = d.batch_outer * d.batch_inner;
    BD_CHECK(!d.a_colsum || (nb == 1 && d.A.kc == 0 && d.A.kind == BD_OPK_DENSE && (d.M & 3) == 0), BD_ERR_UNSUPPORTED,
             "igemm: a_colsum needs a row-contiguous DENSE A operand, M %% 4 == 0 and batch 1");
    if (c.ksplit > 1) {
        size_t need = (size_t)nb * c.ksplit * ((size_t)d.M * d.N + (d.a_colsum ? (size_t)d.M : 0)) * sizeof(float);
        BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE,
                 "igemm: split-K needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
        p.partial = reinterpret_cast<float*>(d.workspace);
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * nb * c.ksplit), 1, 1);
    BD_CHECK(grid.z <= 65535, BD_ERR_UNSUPPORTED, "igemm: batch*ksplit %u too large", grid.z);
    const bool fast = operand_fast_ok(d.A, d.M, d.K) && operand_fast_ok(d.B, d.N, d.K);
    const Cls cls = classify(d, fast);
    int rec = -1;
    if (prof_on()) {
        auto op_bytes = [&](const bd_operand& o, int rows) -> double {
            if (o.kind == BD_OPK_CONV || o.kind == BD_OPK_TCONV) {
                const double imgs = o.kc ? (double)d.M / ((double)o.Ho * o.Wo) : (double)d.K / ((double)o.Ho * o.Wo);
                return imgs * o.Hs * o.Ws * o.C * 4.0;
            }
            return (double)rows * d.K * 4.0;
        };
        char name[128];
        static const bool shapes = getenv("BD_PROF_SHAPES") != nullptr;   // one class per shape (tuning aid)
        if (shapes)
            snprintf(name, sizeof(name), "igemm_%s_%d%s M%d N%d K%d b%d ks%d e%d%d%d%d", kClsName[cls], c.tile,
                     d.mode == BD_MODE_BF16X3 ? "_bf16x3" : "", d.M, d.N, d.K, nb, c.ksplit, d.bias ? 1 : 0, d.rowbias ? 1 : 0,
                     d.residual ? 1 : 0, d.accumulate ? 1 : 0);
        else
            snprintf(name, sizeof(name), "igemm_%s_%d%s", kClsName[cls], c.tile, d.mode == BD_MODE_BF16X3 ? "_bf16x3" : "");
        rec = prof_begin(name, 2.0 * d.M * d.N * (double)d.K * nb,
                         (op_bytes(d.A, d.M) + op_bytes(d.B, d.N) + (double)d.M * d.N * 4.0) * nb, stream);
    }
    if (d.mode == BD_MODE_BF16X3) {
        if (c.tile == 128) launch_tile<128, true>(p, d, cls, grid, stream);
        else launch_tile<64, true>(p, d, cls, grid, stream);
    } else {
        if (c.tile == 128) launch_tile<128, false>(p, d, cls, grid, stream);
        else launch_tile<64, false>(p, d, cls, grid, stream);
    }
    BD_LAUNCH_CHECK("igemm");
    if (c.ksplit > 1) {
        const bool vec4 = (d.N & 3) == 0 && (d.ldc & 3) == 0 && (d.c_bs_outer & 3) == 0 && (d.c_bs_inner & 3) == 0 && aligned16(d.C) &&
                          aligned16(d.workspace) && (!d.residual || ((d.ldr & 3) == 0 && aligned16(d.residual)));
        if (vec4) {
            const long long total4 = (long long)d.M * d.N / 4 * nb + (d.a_colsum ? d.M : 0);
            // many slabs over few outputs: eight lanes per output (the butterfly needs whole groups: 256 % 8 == 0)
            if (c.ksplit >= 16 && total4 < (1 << 20))
                hipLaunchKernelGGL(igemm_splitk_reduce4<8>, dim3((unsigned)cdiv(total4 * 8, 256)), dim3(256), 0, stream, p, nb);
            else
                hipLaunchKernelGGL(igemm_splitk_reduce4<1>, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, stream, p, nb);
            BD_LAUNCH_CHECK("igemm_splitk_reduce4");
            prof_end(rec, stream);
            return BD_OK;
        }
        long long total = (long long)d.M * d.N * nb + (d.a_colsum ? d.M : 0);
        hipLaunchKernelGGL(igemm_splitk_reduce, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, p, nb);
        BD_LAUNCH_CHECK("igemm_splitk_reduce");
    }
    prof_end(rec, stream);
    return BD_OK;
}

}  // namespace bd

extern "C" int bd_split_bf16(const float* src, int64_t n, uint16_t* out, bd_stream_t stream) {
    BD_CHECK(src && out && n > 0 && (n & 31) == 0, BD_ERR_INVALID, "bd_split_bf16: bad args (n %% 32 must be 0)");
    BD_CHECK(bd::aligned16(src) && bd::aligned16(out), BD_ERR_UNSUPPORTED, "bd_split_bf16: pointers must be 16B aligned");
    const long long n8 = n / 8;
    long long nb = bd::cdiv(n8, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bd::split_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, bd::S(stream), src, n8, out);
    BD_LAUNCH_CHECK("split_bf16");
    return BD_OK;
}
extern "C" size_t bd_igemm_workspace_bytes(const bd_igemm_desc* d) { return d ? bd::igemm_workspace_bytes(*d) : 0; }
extern "C" int bd_igemm(const bd_igemm_desc* d, bd_stream_t stream) {
    BD_CHECK(d != nullptr, BD_ERR_INVALID, "bd_igemm: null descriptor");
    return bd::igemm_launch(*d, bd::S(stream));
}
