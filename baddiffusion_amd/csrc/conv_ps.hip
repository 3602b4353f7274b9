// 3x3 stride-1 "same" convolution and its data gradient on PRE-SPLIT operands, fed by LDS-DMA (gfx950).
//
// Same arithmetic as the split-bf16 igemm (igemm.hip: x = hi + lo, product = lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate, K order = channel block outer / filter tap inner) and bit-identical to it,
// but the operands arrive already split: producers (GroupNorm apply, GroupNorm backward, bd_split_rows) write the
// bf16 hi|lo planes in the blocked layout of bd_split_bf16 -- per row, every 32-channel block is one 128-byte line,
// 64 B of hi then 64 B of lo: the same 4 bytes per element as the fp32 tensor it replaces.  A K chunk (one tap, 32
// channels) of an operand row is then exactly one line, and the main loop has no VALU work on the data at all:
//   global --(global_load_lds_dwordx4, 16 B per lane, 8 lanes per row)--> LDS ring (3 stages of 256x128 + 128x128 B)
//   LDS --(ds_read_b128, XOR-swizzled slots: conflict-free)--> MFMA fragments.
// One workgroup = 8 waves (4 x 2) on a 256 x 128 tile (64 x 64 per wave), ONE barrier per K chunk; the DMA of chunk
// c+2 is issued right after the barrier that retires chunk c-1's stage and is waited for with a counted vmcnt two
// chunks later (cdna_hip_programming.md section 5, "Pipelining across barriers": raw s_barrier, never vmcnt(0) in the loop).
// Zero padding: lanes whose tap falls outside the image read a 16-byte zero page instead (branch-free).
//
// data gradient of the same convolution: dx[p][ci] = sum_{tap,co} dy[p - tap][co] * W[co][tap][ci] is the same kernel
// with the gather direction negated (sign = -1) over the TRANSPOSED weight planes Wt[ci][tap][co] (bd_split_wt).
// Replaces aten::convolution / convolution_backward(input) of resnet.py:493,514 for the stride-1 convolutions.
#include "common.h"
#include <string.h>

#include <type_traits>
#include <cstdlib>

namespace bd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

constexpr int BD_WT_MAX = 64;   // weights per batched transpose launch (table passed by value)
constexpr int PS_BM = 256, PS_BN = 128, PS_NT = 512, PS_STAGES = 3;
constexpr int PS_A_BYTES = PS_BM * 128, PS_B_BYTES = PS_BN * 128, PS_STAGE_BYTES = PS_A_BYTES + PS_B_BYTES;
constexpr int PS_LDS_BYTES = PS_STAGES * PS_STAGE_BYTES;   // 147456 of 163840

struct PsParams {
    const char* a;        // activation planes: row m at a + m*lda*4, channel block cb at + cb*128 (64 B hi | 64 B lo)
    const char* w;        // weight planes [N][9][C]: row n, tap t, block cb at ((n*9 + t)*C + cb*32)*4
    float* y;
    const float* bias; const float* rowbias; const float* residual;
    long long lda, ldy, ldr, ld_rowbias;
    int C, H, W, lw, lhw;  // K channels ; image size (powers of two) and log2(W), log2(H*W)
    int sign;              // +1 forward gather (x[p + tap]), -1 data-gradient gather (dy[p - tap])
    int M, N, tiles_m, tiles_n;
    float out_scale;
    int accumulate;
    unsigned short* y_split; long long ldys;   // optional second output: y in split planes (same row count, ld)
    int ablate;            // -DBD_PS_ABLATION builds only (BD_PS_ABLATE): 1 = no steady-state DMA, 2 = no MFMA, 4 = no fragment reads
    int lvw;               // conv_ps3_kernel: log2 of the VIRTUAL image width min(W, 32), see ps_v2r
    double* gn_part;       // conv_ps3_kernel, optional: [B][H*W/256][gn_G][2] partial (sum, sum of squares) of y per pixel tile and group
    int gn_G, gn_lcpg;     //   groups of the GroupNorm that reads y next; log2(channels per group), 2 .. 5
};

// Round 4: VIRTUAL pixel order.  The vertical-tap-sharing kernels (conv_ps3_kernel, conv_ps_wgrad3_kernel) want the rows one image row
// apart to sit a SMALL constant number of pixels apart in the order they walk the image (a 256 + 2W row LDS window, an X ring of 2W + 32
// pixels): fine for W <= 32, impossible for the 64 .. 256 wide layers of the 256 x 256 network.  Those are walked strip by strip instead:
// an image is cut into W/32 vertical strips of 32 columns and virtual pixel v = strip * (H * 32) + y * 32 + xs, so that one image row down is
// always v + 32 -- to the kernels every image looks 32 pixels wide and H * W / 32 rows tall, and only the address of a pixel (this
// function) and the "is the row above / below inside the image" test (y = (v >> 5) & (H - 1)) know about the real layout.  A 256-pixel tile is
// then an 8 x 32 pixel block, a 32-pixel chunk one strip row.  W <= 32: identity.
__device__ __forceinline__ int ps_v2r(int v, int lw, int lh, int lvw) {
    if (lw == lvw) return v;
    const int lhw = lw + lh;
    const int r = v & ((1 << lhw) - 1);
    return (v - r) + (((r >> lvw) & ((1 << lh) - 1)) << lw) + ((r >> (lh + lvw)) << lvw) + (r & ((1 << lvw) - 1));
}

// bank swizzle of the 16-byte slots of a 128-byte LDS row (row stride 128 B = half a 256-byte bank row): rows r and
// r^1 share a bank row, f spreads 16 consecutive rows over the 8 slots x 2 halves -> every ds_read_b128 lane group
// (16 lanes = 16 different rows, same logical slot) touches each bank once.
__device__ __forceinline__ int ps_swz(int row) { return (row >> 1) & 7; }

// 16 zero bytes in the code object: the DMA source of lanes whose filter tap falls outside the image
__device__ __attribute__((aligned(16))) const float kPsZero[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void ps_dma16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)lds_dst, 16, 0, 0);
}

// wait until at most N of this wave's DMAs are outstanding AND all of its LDS reads have returned, then the workgroup
// barrier: behind it the chunk is visible to every wave and the stage read last may be refilled.  Raw s_barrier: a
// __syncthreads() would drain the DMAs in flight (vmcnt(0)).
template <int N>
__device__ __forceinline__ void ps_sync() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
    __builtin_amdgcn_s_barrier();
}

// Epilogue of a wave's TM x 2 grid of 32x32 accumulator tiles: y = out_scale * (acc + bias + rowbias + residual) (+ y).
// The addends of a tile are loaded FIRST (16 independent loads each, all in flight), then combined and stored.  FULL
// (wave-uniform: every row of the sub-tile is < M) removes the per-element predicate: behind an exec-masked branch per
// element hipcc reuses one address register pair and waits vmcnt(0) for the previous store before every store -- a chain
// of ~64 serialised memory round trips per wave that used to be a large part of the per-tile overhead.
// mb[i] = the output row of the first of the 32 consecutive rows of the wave's i-th 32 x 32 tile (consecutive in memory: a virtual-order
// tile of conv_ps3_kernel is 8 image-row pieces of 32 pixels, see ps_v2r)
// STATS (round 4, conv_ps3_kernel only): st[q][0 / 1] += the lane's column sums of y and y * y over its 16 rows of every tile -- the
// statistics of the GroupNorm that reads y next come out of this epilogue instead of a pass over y (ps3_gn_partials)
template <int EPI, int TM, bool FULL, bool STATS = false>
__device__ __forceinline__ void ps_epilogue(const PsParams& p, floatx16 (&acc)[TM][2], const int (&mb)[TM], int nw, int li, int h,
                                            float (*st)[2] = nullptr) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = nw + q * 32 + li;
            const float bn = p.bias ? p.bias[n] : 0.f;
            const int m_base = mb[i] + 4 * h;
            float ad[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ad[r] = 0.f;
            float pc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = FULL ? m_base + (r & 3) + 8 * (r >> 2) : min(m_base + (r & 3) + 8 * (r >> 2), p.M - 1);   // clamped: branch-free loads
                if constexpr (EPI & 2) ad[r] = p.rowbias[(long long)(m >> p.lhw) * p.ld_rowbias + n];
            }
            float rs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = FULL ? m_base + (r & 3) + 8 * (r >> 2) : min(m_base + (r & 3) + 8 * (r >> 2), p.M - 1);
                rs[r] = 0.f; pc[r] = 0.f;
                if constexpr (EPI & 1) rs[r] = p.residual[(long long)m * p.ldr + n];
                if constexpr (EPI & 4) pc[r] = p.y[(long long)m * p.ldy + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (r & 3) + 8 * (r >> 2);
                float v = acc[i][q][r] + bn;
                if constexpr (EPI & 2) v += ad[r];
                if constexpr (EPI & 1) v += rs[r];
                v *= p.out_scale;
                if constexpr (EPI & 4) v += pc[r];
                if (FULL || m < p.M) p.y[(long long)m * p.ldy + n] = v;
                if constexpr (STATS) { st[q][0] += v; st[q][1] += v * v; }
            }
        }
}

// The GroupNorm statistics of a 256 x 128 output tile: the lanes' column sums (ps_epilogue<STATS>) are folded over the two half waves and the
// cpg = 4 .. 32 adjacent columns of a group by lane exchanges, over the four wave rows through LDS (fixed order, fp64), and thread g writes the
// partial of group n0 / cpg + g.  A tile lies inside one sample (H * W % 256 == 0), so the partials are [B][H*W/256][G][2]: what gn_stats_kernel
// writes for S = H*W/256 pixel splits -- the finalize kernel of groupnorm.hip reads either.
__device__ __forceinline__ void ps3_gn_partials(const PsParams& p, float (&st)[2][2], float* red /* LDS, >= 4 * 32 * 2 floats */, int m0, int n0,
                                                int wm, int wn, int li, int h, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = st[q][k];
            v += __shfl_xor(v, 32);
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (p.gn_lcpg > 2) v += __shfl_xor(v, 4);
            if (p.gn_lcpg > 3) v += __shfl_xor(v, 8);
            if (p.gn_lcpg > 4) v += __shfl_xor(v, 16);
            st[q][k] = v;
        }
    __syncthreads();                                       // every wave is done with the operand windows
    if (h == 0 && (li & ((1 << p.gn_lcpg) - 1)) == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = (wn * 64 + q * 32 + li) >> p.gn_lcpg;
            red[(wm * 32 + g) * 2 + 0] = st[q][0];
            red[(wm * 32 + g) * 2 + 1] = st[q][1];
        }
    }
    __syncthreads();
    const int ng = PS_BN >> p.gn_lcpg;
    if (tid < ng) {
        double a = 0.0, c2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += (double)red[(w * 32 + tid) * 2 + 0]; c2 += (double)red[(w * 32 + tid) * 2 + 1]; }
        const int b = m0 >> p.lhw, s = (m0 & ((1 << p.lhw) - 1)) >> 8, S = 1 << (p.lhw - 8);
        double* o = p.gn_part + (((long long)b * S + s) * p.gn_G + (n0 >> p.gn_lcpg) + tid) * 2;
        o[0] = a;
        o[1] = c2;
    }
}

// EPI bits: 1 = residual, 2 = per-sample row bias, 4 = accumulate into y
// All eight waves run in lock step, one barrier per chunk.  (A ping-pong schedule -- the two waves of a SIMD one phase
// apart, four barriers per chunk -- and a wave-specialised form with four extra DMA-only loader waves were built and
// measured: bit-identical, and within +-2 % of this form on every layer)
template <int EPI>
__global__ __launch_bounds__(PS_NT, 2) void conv_ps_kernel(PsParams p) {
    __shared__ __attribute__((aligned(128))) char smem[PS_LDS_BYTES];   // ONE LDS object (guide section 5, trap (a))

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;

    // XCD-contiguous tile order (n fastest): tiles sharing an A panel meet in one L2 (igemm.hip wg_coord)
    int tm, tn;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        tm = j / p.tiles_n;
        tn = j - tm * p.tiles_n;
    }
    const int m0 = tm * PS_BM, n0 = tn * PS_BN;

    // ---- DMA state: wave w moves row groups (8 rows x 128 B = one 1-KiB instruction) w, w+8, w+16, w+24 of A and
    // w, w+8 of B.  Lane l: row = 8*rg + l/8, physical slot l%8 <- logical slot (l%8) ^ swz(row).
    const int dr = lane >> 3, ps = lane & 7;
    const char* ap[4];
    int vm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        const int m = m0 + r;
        const int x = m & (p.W - 1), y = (m >> p.lw) & (p.H - 1);
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + p.sign * (t / 3 - 1), xx = x + p.sign * (t % 3 - 1);
            if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) mask |= 1 << t;
        }
        vm[j] = m < p.M ? mask : 0;
        ap[j] = p.a + (long long)m * p.lda * 4 + ((ps ^ ps_swz(r)) << 4);
    }
    const char* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        int n = n0 + r;
        if (n >= p.N) n = p.N - 1;
        wp[j] = p.w + (long long)n * 36 * p.C + ((ps ^ ps_swz(r)) << 4);
    }
    const int pix_bytes = (int)p.lda * 4;   // bytes between pixels (host checks (W + 1) * pix_bytes < 2^31)
    const int nchunks = 9 * (p.C >> 5);

    // chunk q -> (channel block cb, tap): K order = channel block outer, tap inner (the 9 taps share one input window)
    // cursor of the NEXT chunk to issue, all wave-uniform: tap = 3*kh + kw, byte offsets of the tap / channel block
    int q_kh = 0, q_kw = 0, q_bit = 1;
    int q_aoff = -p.sign * (p.W + 1) * pix_bytes, q_woff = 0;   // tap (0,0) of block 0
    int issued = 0;
    auto issue = [&](char* stage) {
#ifdef BD_PS_ABLATION
        if ((p.ablate & 1) && issued >= 2) return;
#endif
        ++issued;
#pragma unroll
        for (int j = 0; j < 4; ++j) ps_dma16((vm[j] & q_bit) ? ap[j] + q_aoff : reinterpret_cast<const char*>(kPsZero), stage + (wave + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) ps_dma16(wp[j] + q_woff, stage + PS_A_BYTES + (wave + 8 * j) * 1024);
        // advance: kw fastest; a full window moves on to the next 32-channel block
        q_bit <<= 1;
        q_woff += p.C * 4;
        q_aoff += p.sign * pix_bytes;
        if (++q_kw == 3) {
            q_kw = 0;
            q_aoff += p.sign * (p.W - 3) * pix_bytes;
            if (++q_kh == 3) {
                q_kh = 0; q_bit = 1;
                q_aoff += 128 - p.sign * 3 * p.W * pix_bytes;
                q_woff += 128 - 9 * p.C * 4;
            }
        }
    };

    // ---- fragment addresses: lane -> row li of a 32-row tile, k octet h; logical slot = plane*4 + step*2 + h
    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) foff[s][pl] = li * 128 + (((pl * 4 + s * 2 + h) ^ ps_swz(li)) << 4);
    const int abase = wm * 64 * 128, bbase = PS_A_BYTES + wn * 64 * 128;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#ifdef BD_PS_ABLATION
            if (p.ablate & 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = al[i] = bh[i] = bl[i] = __builtin_bit_cast(bf16x8, make_float4(1.f, 1.f, (float)s, 1.f)); }
            } else
#endif
            {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(stage + abase + i * 4096 + foff[s][0]);
                al[i] = *reinterpret_cast<const bf16x8*>(stage + abase + i * 4096 + foff[s][1]);
                bh[i] = *reinterpret_cast<const bf16x8*>(stage + bbase + i * 4096 + foff[s][0]);
                bl[i] = *reinterpret_cast<const bf16x8*>(stage + bbase + i * 4096 + foff[s][1]);
            }
            }
#ifdef BD_PS_ABLATION
            if (p.ablate & 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
                continue;
            }
#endif
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

    char* const st0 = smem;
    char* const st1 = smem + PS_STAGE_BYTES;
    char* const st2 = smem + 2 * PS_STAGE_BYTES;

    // prologue: chunks 0 and 1 in flight (nchunks >= 9)
    issue(st0);
    issue(st1);
    // steady state, 3 chunks per trip (nchunks % 9 == 0): wait for chunk c (6 DMAs per wave and chunk, so vmcnt(6) leaves
    // chunk c+1 in flight), barrier (chunk c visible to everyone, chunk c-1's stage free), refill that stage with chunk c+2
    const int trips = nchunks / 3;
    for (int t = 0; t + 1 < trips; ++t) {
        ps_sync<6>(); issue(st2); compute(st0);
        ps_sync<6>(); issue(st0); compute(st1);
        ps_sync<6>(); issue(st1); compute(st2);
    }
    ps_sync<6>(); issue(st2); compute(st0);
    ps_sync<6>(); compute(st1);
    ps_sync<0>(); compute(st2);

    // ---- epilogue: lane holds column n = li of rows (r&3) + 8*(r>>2) + 4*h of every 32x32 tile
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
    const int mb[2] = {mw, mw + 32};
    if (mw + 64 <= p.M) ps_epilogue<EPI, 2, true>(p, acc, mb, nw, li, h);    // wave-uniform: whole sub-tile inside M
    else ps_epilogue<EPI, 2, false>(p, acc, mb, nw, li, h);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same 256 x 128 tile with the three VERTICAL taps of a column shift served from ONE LDS window.
// The tile's pixels are consecutive, so the A tiles of taps (ky, kx), ky = 0..2, are the same rows shifted by W pixels: one window
// of 256 + 2W rows (the tile plus one image row above and below) is DMA'd once per (32-channel block, kx) and the three ky taps read
// it at row offsets 0 / W / 2W -- a constant byte offset, and (W/2) % 8 == 0 keeps the bank swizzle of a row unchanged.  Rows above /
// below the image and columns shifted out of it read the zero page as before: no masks, no VALU in the loop.  L2 -> LDS bytes per
// channel block: 3 x 40 KB + 9 x 16 KB = 264 KB instead of 9 x 48 KB = 432 KB.  W in {16, 32} (window <= 320 rows), K order = channel
// block outer, kx, ky inner (another fp32 summation order than conv_ps_kernel: same products).
// LDS: two A windows (2 x 40 KB) + a ring of three weight chunks (3 x 16 KB) = 128 KB.
constexpr int PS3_WIN_ROWS = 320, PS3_A_BYTES = PS3_WIN_ROWS * 128, PS3_B_BYTES = 128 * 128;
constexpr int PS3_LDS_BYTES = 2 * PS3_A_BYTES + 3 * PS3_B_BYTES;

template <int EPI>
__global__ __launch_bounds__(PS_NT, 2) void conv_ps3_kernel(PsParams p) {
    __shared__ __attribute__((aligned(128))) char smem[PS3_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    int tm, tn;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        tm = j / p.tiles_n;
        tn = j - tm * p.tiles_n;
    }
    const int m0 = tm * PS_BM, n0 = tn * PS_BN;

    // ---- A window: wave w moves row groups (8 rows = one 1-KiB DMA) w, w+8, .., w+32 of the 320-row window; window row wr is
    // (virtual) pixel m0 - VW + wr.  Per row: validity of the image row, and of the three column shifts.
    const int dr = lane >> 3, ps = lane & 7;
    const char* ap[5];
    int vm[5];
    // (m0 and the window rows are VIRTUAL pixels, ps_v2r: VW = min(W, 32) wide rows; a tile = 256 / VW rows of one strip of one image)
    const int VW = 1 << p.lvw, lh = p.lhw - p.lw;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int wr = (wave + 8 * j) * 8 + dr;
        const int m = ps_v2r(m0 - VW + wr, p.lw, lh, p.lvw);  // may be < 0 or past the image: masked below
        const int x = m & (p.W - 1);
        const int yrel = (wr >> p.lvw) - 1;                  // image row relative to the tile's first row
        const int y0 = (m0 >> p.lvw) & (p.H - 1);
        const int yy = y0 + yrel;
        int mask = 0;
        if ((unsigned)yy < (unsigned)p.H && wr < PS_BM + 2 * VW) {       // (tiles never straddle strips / images: H * VW % 256 == 0, host-checked)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                if ((unsigned)(x + p.sign * (kx - 1)) < (unsigned)p.W) mask |= 1 << kx;
        }
        vm[j] = mask;
        ap[j] = p.a + (long long)m * p.lda * 4 + ((ps ^ ps_swz(wr)) << 4);
    }
    const char* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        int n = n0 + r;
        if (n >= p.N) n = p.N - 1;
        wp[j] = p.w + (long long)n * 36 * p.C + ((ps ^ ps_swz(r)) << 4);
    }
    const int pix_bytes = (int)p.lda * 4;
    const int ngroups = 3 * (p.C >> 5);          // (channel block, kx)
    const int nchunks = 3 * ngroups;

    // cursors (wave-uniform): next A group (cb, kx), next weight chunk (cb, kx, ky)
    int ga_kx = 0, ga_cb = 0;
    auto issue_a = [&](char* win) {
        const int aoff = p.sign * (ga_kx - 1) * pix_bytes + ga_cb * 128;
        const int bit = 1 << ga_kx;
#pragma unroll
        for (int j = 0; j < 5; ++j) ps_dma16((vm[j] & bit) ? ap[j] + aoff : reinterpret_cast<const char*>(kPsZero), win + (wave + 8 * j) * 1024);
        if (++ga_kx == 3) { ga_kx = 0; ++ga_cb; }
    };
    int gb_ky = 0, gb_kx = 0, gb_cb = 0;
    auto issue_b = [&](char* slot) {
        const int woff = ((gb_ky * 3 + gb_kx) * p.C + gb_cb * 32) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) ps_dma16(wp[j] + woff, slot + (wave + 8 * j) * 1024);
        if (++gb_ky == 3) { gb_ky = 0; if (++gb_kx == 3) { gb_kx = 0; ++gb_cb; } }
    };

    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) foff[s][pl] = li * 128 + (((pl * 4 + s * 2 + h) ^ ps_swz(li)) << 4);
    const int abase = wm * 64 * 128, bbase = wn * 64 * 128;
    // vertical tap ky of the forward gather reads window rows r + ky*VW; the data-gradient gather (sign -1) mirrors it
    const int wrow_bytes = VW * 128;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const char* win, const char* bslot, int ky) {
        const char* a0 = win + abase + (p.sign > 0 ? ky : 2 - ky) * wrow_bytes;
        const char* b0 = bslot + bbase;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(a0 + i * 4096 + foff[s][0]);
                al[i] = *reinterpret_cast<const bf16x8*>(a0 + i * 4096 + foff[s][1]);
                bh[i] = *reinterpret_cast<const bf16x8*>(b0 + i * 4096 + foff[s][0]);
                bl[i] = *reinterpret_cast<const bf16x8*>(b0 + i * 4096 + foff[s][1]);
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

    char* const A0 = smem; char* const A1 = smem + PS3_A_BYTES;
    char* const B0 = smem + 2 * PS3_A_BYTES; char* const B1 = B0 + PS3_B_BYTES; char* const B2 = B1 + PS3_B_BYTES;

    // issue order per wave: A(0) [5 DMAs], B(0) [2], B(1) [2]; then behind the barrier of chunk c = 3g + ky:
    //   ky == 0: A(g+1) [5] and B(c+2) [2];  ky == 1, 2: B(c+2) [2].
    // Chunk c needs B(c) -- and everything issued before it, i.e. A(g) -- landed: DMAs issued after B(c) are
    //   ky == 0: B(c+1) = 2;   ky == 1: A(g+1) + B(c+1) = 7;   ky == 2: B(c+1) = 2.
    issue_a(A0);
    issue_b(B0);
    issue_b(B1);
    auto sync2 = [&]() { asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
    auto sync7 = [&]() { asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
    auto sync0 = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
    // two groups (six chunks) per trip so that the A windows and the three weight slots are compile-time addresses
    for (int g = 0; g < ngroups; g += 2) {
        const bool more1 = g + 1 < ngroups, more2 = g + 2 < ngroups;
        // group g on A0: chunks 3g .. 3g+2 on slots (3g) % 3 = 0, 1, 2 (3g is a multiple of 3)
        sync2(); if (more1) issue_a(A1); issue_b(B2); compute(A0, B0, 0);
        if (more1) sync7(); else sync2();
        if (more1) issue_b(B0); compute(A0, B1, 1);
        if (more1) sync2(); else sync0();
        if (more1) issue_b(B1); compute(A0, B2, 2);
        if (!more1) break;
        // group g + 1 on A1
        sync2(); if (more2) issue_a(A0); issue_b(B2); compute(A1, B0, 0);
        if (more2) sync7(); else sync2();
        if (more2) issue_b(B0); compute(A1, B1, 1);
        if (more2) sync2(); else sync0();
        if (more2) issue_b(B1); compute(A1, B2, 2);
    }
    (void)nchunks;

    // output rows: the wave's two 32-pixel blocks, virtual -> real (32 consecutive virtual pixels are consecutive in memory)
    const int nw = n0 + wn * 64;
    const int mb[2] = {ps_v2r(m0 + wm * 64, p.lw, lh, p.lvw), ps_v2r(m0 + wm * 64 + 32, p.lw, lh, p.lvw)};
    if constexpr (EPI == 1 || EPI == 2) {                    // (forward conv1 / conv2 of a resnet: the producers of a GroupNorm's input)
        if (p.gn_part) {                                     // kernel-uniform
            float st[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            ps_epilogue<EPI, 2, true, true>(p, acc, mb, nw, li, h, st);
            ps3_gn_partials(p, st, reinterpret_cast<float*>(smem), m0, n0, wm, wn, li, h, tid);
            return;
        }
    }
    ps_epilogue<EPI, 2, true>(p, acc, mb, nw, li, h);        // (M % 256 == 0: every tile is whole, host-checked)
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 128 tile variant for the SMALL layers (8x8 / 4x4 images: too few 256 x 128 tiles to fill the chip): same operands,
// same fragment layout, 8 waves of 32 x 64, two LDS stages (64 KB -> two workgroups per CU, which de-phase), and the K
// chunks (tap, 32 channels) split over `ksplit` workgroups per tile: partial slabs + a fixed-order second pass that also
// applies the epilogue (deterministic, no atomics).
struct PsSmallParams {
    PsParams q;
    float* partial;        // [ksplit][M][N] when ksplit > 1
    int ksplit, cps;       // chunks per split
};

// STAGES = 2: 64 KB, two workgroups per CU (the pair de-phases); the chunk c+1 is fetched while chunk c computes.
// STAGES = 4 (round 6): 128 KB, one workgroup per CU, THREE chunks in flight ahead of the one being computed -- for the short K slices of
// the 4 x 4 level (4 - 9 chunks per workgroup), where a workgroup's life is a handful of DMA round trips and the two-stage form exposes one
// L2 -> LDS latency per chunk.
template <int EPI, int STAGES = 2>
__global__ __launch_bounds__(512, STAGES == 2 ? 4 : 2) void conv_ps128_kernel(PsSmallParams pp) {
    const PsParams& p = pp.q;
    constexpr int A_BYTES = 128 * 128, STAGE = 2 * A_BYTES;
    __shared__ __attribute__((aligned(128))) char smem[STAGES * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    int tm, tn, zz;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        const unsigned ntiles = p.tiles_m * p.tiles_n;
        zz = j / ntiles;
        const unsigned tile = j - zz * ntiles;
        tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
    }
    const int m0 = tm * 128, n0 = tn * 128;
    const int nchunks = 9 * (p.C >> 5);
    const int c_begin = zz * pp.cps;
    int c_end = c_begin + pp.cps;
    if (c_end > nchunks) c_end = nchunks;

    const int dr = lane >> 3, ps = lane & 7;
    const char* ap[2]; const char* wp[2];
    int vm[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 8 + dr;
        const int m = m0 + r;
        const int x = m & (p.W - 1), y = (m >> p.lw) & (p.H - 1);
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + p.sign * (t / 3 - 1), xx = x + p.sign * (t % 3 - 1);
            if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) mask |= 1 << t;
        }
        vm[j] = m < p.M ? mask : 0;
        ap[j] = p.a + (long long)m * p.lda * 4 + ((ps ^ ps_swz(r)) << 4);
        int n = n0 + r;
        if (n >= p.N) n = p.N - 1;
        wp[j] = p.w + (long long)n * 36 * p.C + ((ps ^ ps_swz(r)) << 4);
    }
    const int pix_bytes = (int)p.lda * 4;
    int q_cb = c_begin / 9;
    int q_tap = c_begin - 9 * q_cb;
    int q_kh = q_tap / 3, q_kw = q_tap - 3 * q_kh, q_bit = 1 << q_tap;
    int q_aoff = p.sign * ((q_kh - 1) * p.W + (q_kw - 1)) * pix_bytes + q_cb * 128;
    int q_woff = (q_tap * p.C + q_cb * 32) * 4;
    auto issue = [&](char* stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ps_dma16((vm[j] & q_bit) ? ap[j] + q_aoff : reinterpret_cast<const char*>(kPsZero), stage + (wave + 8 * j) * 1024);
            ps_dma16(wp[j] + q_woff, stage + A_BYTES + (wave + 8 * j) * 1024);
        }
        q_bit <<= 1;
        q_woff += p.C * 4;
        q_aoff += p.sign * pix_bytes;
        if (++q_kw == 3) {
            q_kw = 0;
            q_aoff += p.sign * (p.W - 3) * pix_bytes;
            if (++q_kh == 3) {
                q_kh = 0; q_bit = 1;
                q_aoff += 128 - p.sign * 3 * p.W * pix_bytes;
                q_woff += 128 - 9 * p.C * 4;
            }
        }
    };
    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) foff[s][pl] = li * 128 + (((pl * 4 + s * 2 + h) ^ ps_swz(li)) << 4);
    const int abase = wm * 32 * 128, bbase = A_BYTES + wn * 64 * 128;
    floatx16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    auto compute = [&](const char* stage) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(stage + abase + foff[s][0]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(stage + abase + foff[s][1]);
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bh[i] = *reinterpret_cast<const bf16x8*>(stage + bbase + i * 4096 + foff[s][0]);
                bl[i] = *reinterpret_cast<const bf16x8*>(stage + bbase + i * 4096 + foff[s][1]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[q], acc[q], 0, 0, 0);
        }
    };
    const int n = c_end - c_begin;
    if constexpr (STAGES == 2) {
        if (n > 0) issue(smem);
        for (int c = 0; c < n; ++c) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + 1 < n) issue(smem + ((c + 1) & 1) * STAGE);
            compute(smem + (c & 1) * STAGE);
        }
    } else {
        // ring of STAGES: chunks c+1 .. c+STAGES-1 are in flight while chunk c computes (4 DMAs per wave and chunk: counted waits).  The stage
        // refilled behind the barrier of chunk c is the one chunk c-1 was read from: every wave has passed that barrier with its reads returned.
#pragma unroll
        for (int k = 0; k < STAGES - 1; ++k)
            if (k < n) issue(smem + k * STAGE);
        for (int c = 0; c < n; ++c) {
            const int later = (n < c + STAGES - 1 ? n : c + STAGES - 1) - (c + 1);      // chunks issued after chunk c (wave-uniform)
            if (later >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + STAGES - 1 < n) issue(smem + ((c + STAGES - 1) % STAGES) * STAGE);
            compute(smem + (c % STAGES) * STAGE);
        }
    }
    const int mw = m0 + wm * 32, nw = n0 + wn * 64;
    const bool full = mw + 32 <= p.M;     // wave-uniform
    if (pp.ksplit > 1) {
        float* out = pp.partial + (long long)zz * p.M * p.N;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nn = nw + q * 32 + li;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(long long)(mw + (r & 3) + 8 * (r >> 2) + 4 * h) * p.N + nn] = acc[q][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M) out[(long long)m * p.N + nn] = acc[q][r];
                }
            }
        }
        return;
    }
    floatx16 a2[1][2] = {{acc[0], acc[1]}};
    const int mb[1] = {mw};
    if (full) ps_epilogue<EPI, 1, true>(p, a2, mb, nw, li, h);
    else ps_epilogue<EPI, 1, false>(p, a2, mb, nw, li, h);
}

// second pass of the K split: fixed-order sum of the slabs + the epilogue, float4 per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void conv_ps128_reduce(PsSmallParams pp) {
    const PsParams& p = pp.q;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N >> 2;
    if (i >= (long long)p.M * n4) return;
    const int m = (int)(i / n4), nn = (int)(i - (long long)m * n4) * 4;
    const long long mn = (long long)p.M * p.N, e = (long long)m * p.N + nn;
    float4 a = *reinterpret_cast<const float4*>(pp.partial + e);
#pragma unroll 4      // four slab loads in flight; the additions keep their fixed order
    for (int s = 1; s < pp.ksplit; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(pp.partial + (long long)s * mn + e);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + nn);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.rowbias) {
        const float4 b = *reinterpret_cast<const float4*>(p.rowbias + (long long)(m >> p.lhw) * p.ld_rowbias + nn);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.residual) {
        const float4 b = *reinterpret_cast<const float4*>(p.residual + (long long)m * p.ldr + nn);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= p.out_scale;
    float4* dst = reinterpret_cast<float4*>(p.y + (long long)m * p.ldy + nn);
    if (p.accumulate) {
        const float4 b = *dst;
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    *dst = make_float4(v[0], v[1], v[2], v[3]);
}

// =====================================================================================================================
// Weight gradient of the same convolutions on split planes:  dW[co][tap][ci] = sum_p dY[p][co] * X[p + tap][ci]
//   GEMM view: M = Cout, N = 9*Cin (n = tap*Cin + ci), K = pixels.  Both operands are pixel-major ("row contiguous" in
//   igemm.hip's terms), so a K chunk of 32 pixels x 128 channels is 32 rows of 512 contiguous bytes: two pixels per 1-KiB
//   DMA instruction, and the MFMA fragments (k along pixels) are gathered with the hardware transpose read
//   ds_read_b64_tr_b16.  LDS image of a chunk: [32 pixels][512 B]; pixel k's 16-byte slots are XOR-permuted by
//   (k & 3) << 2 on the SOURCE side of the DMA (cdna guide rule 21), which puts the four pixels a transpose read touches
//   on four different 64-byte bank windows: conflict-free.
//   One workgroup = 8 waves (4 x 2) on a 128 (co) x 128 (one tap, 128 ci) tile, 32 x 64 per wave; K is split over
//   ksplit workgroups per tile (fixed-order second pass, deterministic).  The bias gradient sum_p dY[p][co] rides along
//   as two extra MFMAs against a vector of ones in the tn == 0 workgroups.
// Replaces aten::convolution_backward(weight, bias) of resnet.py:493,514 for the stride-1 convolutions.
constexpr int WG_BM = 128, WG_BN = 128, WG_OP_BYTES = 32 * 512, WG_STAGE_BYTES = 2 * WG_OP_BYTES;

struct PsWgParams {
    const char* dy; const char* x;     // split planes: dY [P][lddy], X [P][ldx]
    long long lddy, ldx;
    float* out;                        // ksplit == 1: dW [M][N]; else partial slabs [ksplit][M][N] then [ksplit][M] bias rows
    float* db;                         // ksplit == 1 only (else the reduce pass writes it); may be null
    int Cin, Cout, H, W, lw;           // image grid (powers of two)
    int P;                             // pixels = K
    int tiles_m, tiles_n, ksplit, cps; // chunks (of 32 pixels) per split
    int want_db;
    int ablate;                        // -DBD_PS_ABLATION builds only, as in PsParams
    int ntaps;                         // taps per output row: 9, or 16 in the PHASE form (conv_ph.hip: upsample convolution)
    int lh, lvw;                       // conv_ps_wgrad3_kernel: log2(H), log2 of the virtual image width min(W, 32) (ps_v2r)
};

typedef short ps_short4 __attribute__((ext_vector_type(4)));
typedef short ps_short8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) ps_short4 ps_lds_short4;

__device__ __forceinline__ bf16x8 ps_tr_frag(const char* lds, int off) {
    const ps_short4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ps_lds_short4*)(lds + off));
    const ps_short4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ps_lds_short4*)(lds + off + 4 * 512));
    const ps_short8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// The same read as inline asm, address = LDS byte offset.  hipcc puts `s_waitcnt vmcnt(0)` in front of the intrinsic form
// whenever LDS-DMA loads are in flight (it cannot tell the stage being read from the stage being filled), so every chunk
// waited for the NEXT chunk's DMA before its first fragment read.  The asm form is opaque to that pass; the matching
// s_waitcnt lgkmcnt(0) is tied to the fragment registers so that no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ ps_short4 ps_tr_read(unsigned addr) {
    ps_short4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ bf16x8 ps_tr_join(ps_short4 v0, ps_short4 v1) {
    const ps_short8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// NW = 8: waves 4 x 2, 32 x 64 each.  NW = 4: waves 2 x 2, 64 x 64 each (a third fewer LDS fragment reads per MFMA, twice
// the MFMAs between barriers, half the waves per SIMD).
// PH = true: the PHASE form for the upsample convolution (conv_ph.hip).  H x W is the SOURCE grid, X lives on it, dY on the 2H x 2W
// grid; the 16 "taps" e = (oy+1)*4 + (ox+1) are the entries of E: entry (oy, ox) belongs to pixel class (p, q) of the fine grid and
// to the source-grid shift (dy, dx) with oy -> (p, dy): -1 -> (1, +1), 0 -> (0, 0), 1 -> (1, 0), 2 -> (0, -1):
//     dE[e][co][ci] = sum_{source pixels (a, b)} dY[2a+p, 2b+q][co] * X[a+dy, b+dx][ci].
// The bias gradient (sum of ALL dY) rides in the four (dy, dx) = (0, 0) entries, one bias row per class and K split.
template <int STAGES, int NW, bool PH = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 2) void conv_ps_wgrad_kernel(PsWgParams p) {
    constexpr int TMW = 8 / NW;        // 32-row co tiles per wave
    constexpr int NDMA = 16 / NW;      // pixel pairs per wave, operand and chunk
    __shared__ __attribute__((aligned(128))) char smem[STAGES * WG_STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // logical order: n fastest, then m, then the K split; one contiguous run per XCD (same k-slice -> same L2)
    int tm, tn, zz;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        const unsigned ntiles = p.tiles_m * p.tiles_n;
        zz = j / ntiles;
        const unsigned tile = j - zz * ntiles;
        tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
    }
    const int co0 = tm * WG_BM, n0 = tn * WG_BN;
    const int tap = n0 / p.Cin, ci0 = n0 - tap * p.Cin;
    int dyt, dxt, cp = 0, cq = 0;
    if constexpr (PH) {
        const int oy = tap / 4 - 1, ox = (tap & 3) - 1;
        cp = oy & 1; cq = ox & 1;                                   // oy in {-1, 1} -> class row 1, {0, 2} -> 0
        dyt = oy == 2 ? -1 : (oy == -1 ? 1 : 0); dxt = ox == 2 ? -1 : (ox == -1 ? 1 : 0);
    } else {
        dyt = tap / 3 - 1; dxt = tap - (tap / 3) * 3 - 1;
    }
    const long long fine_off = (long long)cp * 2 * p.W + cq;        // PH: dY row of source pixel i = 4i - 2(i & (W-1)) + fine_off
    const int nchunks = (p.P + 31) >> 5;   // a ragged last chunk reads zeros past the last pixel
    const int c_begin = zz * p.cps;
    int c_end = c_begin + p.cps;
    if (c_end > nchunks) c_end = nchunks;

    // ---- DMA: wave w moves pixel pairs w and w + 8 of both operands; lane: pixel k = 2*pair + lane/32, slot lane%32
    const int ps = lane & 31;
    const char* asrc[NDMA]; const char* bsrc[NDMA];
    int kpix[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int k = 2 * (wave + NW * j) + (lane >> 5);
        kpix[j] = k;
        const int ls = ps ^ ((k & 3) << 2);
        const long long pa = (long long)c_begin * 32 + k;
        asrc[j] = PH ? p.dy + co0 * 4 + ls * 16 : p.dy + pa * p.lddy * 4 + co0 * 4 + ls * 16;
        bsrc[j] = p.x + (pa + dyt * p.W + dxt) * p.ldx * 4 + ci0 * 4 + ls * 16;
    }
    const long long a_adv = 32 * p.lddy * 4, b_adv = 32 * p.ldx * 4;
    int q_pix = c_begin * 32;   // first pixel of the next chunk to issue
#ifdef BD_PS_ABLATION
    int issued = 0;
#endif
    auto issue_j = [&](char* stage, int j) {
        const int pp = q_pix + kpix[j];
        const int x = pp & (p.W - 1), y = (pp >> p.lw) & (p.H - 1);
        const bool in = pp < p.P;
        const bool ok = in && (unsigned)(y + dyt) < (unsigned)p.H && (unsigned)(x + dxt) < (unsigned)p.W;
        if constexpr (PH) {
            const long long fr = 4ll * pp - 2 * x + fine_off;       // pixel (2a+p, 2b+q) of the fine grid, x = pp & (W-1)
            ps_dma16(in ? asrc[j] + fr * p.lddy * 4 : reinterpret_cast<const char*>(kPsZero), stage + (wave + NW * j) * 1024);
        } else {
            ps_dma16(in ? asrc[j] : reinterpret_cast<const char*>(kPsZero), stage + (wave + NW * j) * 1024);
            asrc[j] += a_adv;
        }
#ifdef BD_PS_ABLATION
        if (!(p.ablate & 8) || ((pp >> 5) % 3) == 0)     // 8: the X operand fetched for one chunk in three (timing model of vertical-tap sharing)
#endif
        ps_dma16(ok ? bsrc[j] : reinterpret_cast<const char*>(kPsZero), stage + WG_OP_BYTES + (wave + NW * j) * 1024);
        bsrc[j] += b_adv;
    };
    auto issue = [&](char* stage) {
#ifdef BD_PS_ABLATION
        if ((p.ablate & 1) && issued >= 2) return;
        ++issued;
#endif
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_j(stage, j);
        q_pix += 32;
    };

    // ---- fragment addresses (ds_read_b64_tr_b16, see igemm.hip rc_frag): lane reads 4 rows x 1 pixel (8 bytes)
    const int sl = lane & 15, hb = (lane >> 4) & 1, h = lane >> 5;
    const int kq = sl >> 2, rq = sl & 3;
    const int lane_base = (8 * h + kq) * 512 + hb * 32 + rq * 8;
    int xw[4];   // 64-byte window of (tile parity, plane) after the per-pixel XOR
#pragma unroll
    for (int v = 0; v < 4; ++v) xw[v] = (v ^ kq) << 6;
    auto foff = [&](int t, int plane, int kk) { return lane_base + xw[(t & 1) * 2 + plane] + (t >> 1) * 256 + kk * 512; };

    floatx16 acc[TMW][2], accb[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
    // wave-uniform.  PH: the (dy, dx) = (0, 0) entry of each pixel class (taps 5, 6, 9, 10), first channel block
    const bool do_db = p.want_db && wn == 0 && (PH ? (dyt == 0 && dxt == 0 && ci0 == 0) : tn == 0);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // stage-relative fragment addresses of this wave's tiles (the 16-pixel step and the pixel half are immediates)
    unsigned aoff[TMW][2], boff[2][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) aoff[i][pl] = (unsigned)foff(wm * TMW + i, pl, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) boff[q][pl] = (unsigned)foff(wn * 2 + q, pl, 0) + WG_OP_BYTES;
    const unsigned smem_addr = (unsigned)(uintptr_t)(lds_ptr)smem;

    struct Frag { ps_short4 a0[TMW][2], a1[TMW][2], b0[2][2], b1[2][2]; };
    auto reads = [&](auto S, unsigned sbase, Frag& f) {
        constexpr int KOFF = decltype(S)::value * 16 * 512;
#ifdef BD_PS_ABLATION
        if (p.ablate & 4) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) f.a0[i][pl] = f.a1[i][pl] = ps_short4{0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) f.b0[q][pl] = f.b1[q][pl] = ps_short4{0x3f80, 0x3f80, 0x3f80, 0x3f80};
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                f.a0[i][pl] = ps_tr_read<KOFF>(sbase + aoff[i][pl]);
                f.a1[i][pl] = ps_tr_read<KOFF + 4 * 512>(sbase + aoff[i][pl]);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                f.b0[q][pl] = ps_tr_read<KOFF>(sbase + boff[q][pl]);
                f.b1[q][pl] = ps_tr_read<KOFF + 4 * 512>(sbase + boff[q][pl]);
            }
    };
    // all LDS reads of this wave have returned; the fragment registers pass through the asm so no MFMA can move above it
    auto wait = [&](Frag& f) {
        if constexpr (TMW == 1)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.a0[0][0]), "+v"(f.a1[0][0]), "+v"(f.a0[0][1]), "+v"(f.a1[0][1]), "+v"(f.b0[0][0]), "+v"(f.b1[0][0]),
                           "+v"(f.b0[0][1]), "+v"(f.b1[0][1]), "+v"(f.b0[1][0]), "+v"(f.b1[1][0]), "+v"(f.b0[1][1]), "+v"(f.b1[1][1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.a0[0][0]), "+v"(f.a1[0][0]), "+v"(f.a0[0][1]), "+v"(f.a1[0][1]), "+v"(f.a0[TMW - 1][0]),
                           "+v"(f.a1[TMW - 1][0]), "+v"(f.a0[TMW - 1][1]), "+v"(f.a1[TMW - 1][1]), "+v"(f.b0[0][0]), "+v"(f.b1[0][0]),
                           "+v"(f.b0[0][1]), "+v"(f.b1[0][1]), "+v"(f.b0[1][0]), "+v"(f.b1[1][0]), "+v"(f.b0[1][1]), "+v"(f.b1[1][1]));
    };
    auto mfmas = [&](const Frag& f) {
        bf16x8 ah[TMW], al[TMW], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < TMW; ++i) { ah[i] = ps_tr_join(f.a0[i][0], f.a1[i][0]); al[i] = ps_tr_join(f.a0[i][1], f.a1[i][1]); }
#pragma unroll
        for (int q = 0; q < 2; ++q) { bh[q] = ps_tr_join(f.b0[q][0], f.b1[q][0]); bl[q] = ps_tr_join(f.b0[q][1], f.b1[q][1]); }
#ifdef BD_PS_ABLATION
        if (p.ablate & 2) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
            for (int q = 0; q < 2; ++q) asm volatile("" ::"v"(bh[q]), "v"(bl[q]));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[q], acc[i][q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[q], acc[i][q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[q], acc[i][q], 0, 0, 0);
        if (do_db) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], ones, accb[i], 0, 0, 0);
                accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], ones, accb[i], 0, 0, 0);
            }
        }
    };
    // one chunk: the second 16-pixel step's fragments are read while the first step's MFMAs run, and the next chunk's DMA
    // pieces are issued between the groups (next == nullptr: nothing left to fetch)
    auto chunk = [&](const char* stage, char* next) {
        const unsigned sbase = smem_addr + (unsigned)(stage - smem);
        Frag f0, f1;
        reads(std::integral_constant<int, 0>{}, sbase, f0);
#ifdef BD_PS_ABLATION
        if ((p.ablate & 1) && issued >= 2) next = nullptr;
        if (next) ++issued;
#endif
        if (next) issue_j(next, 0);
        wait(f0);
        reads(std::integral_constant<int, 1>{}, sbase, f1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0);
        if (next) {
#pragma unroll
            for (int j = 1; j < NDMA; ++j) issue_j(next, j);
            q_pix += 32;
        }
        __builtin_amdgcn_sched_barrier(0);
        wait(f1);
        mfmas(f1);
    };

    const int n = c_end - c_begin;
    if constexpr (STAGES == 3) {
        // ring of three: chunk c+2 is issued behind the barrier that frees chunk c-1's stage
        if (n > 0) issue(smem);
        if (n > 1) issue(smem + WG_STAGE_BYTES);
        for (int c = 0; c < n; ++c) {
            if (c + 1 < n) {
                if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            chunk(smem + (c % 3) * WG_STAGE_BYTES, c + 2 < n ? smem + ((c + 2) % 3) * WG_STAGE_BYTES : nullptr);
        }
    } else {
        if (n > 0) issue(smem);
        for (int c = 0; c < n; ++c) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            chunk(smem + (c & 1) * WG_STAGE_BYTES, c + 1 < n ? smem + ((c + 1) & 1) * WG_STAGE_BYTES : nullptr);
        }
    }

    // ---- epilogue: lane holds column n = li of rows (r&3) + 8*(r>>2) + 4*h
    const int li = lane & 31;
    const int M = p.Cout, N = p.ntaps * p.Cin;
    float* out = p.out + ((p.ksplit > 1 || PH) ? (long long)zz * M * N : 0);
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nn = n0 + wn * 64 + q * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = co0 + (wm * TMW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(long long)m * N + nn] = acc[i][q][r];
            }
        }
    if (do_db && li == 0) {
        float* o = PH ? p.out + (long long)p.ksplit * M * N + ((long long)(cp * 2 + cq) * p.ksplit + zz) * M
                      : (p.ksplit > 1 ? p.out + (long long)p.ksplit * M * N + (long long)zz * M : p.db);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[co0 + (wm * TMW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = accb[i][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: weight gradient with the three VERTICAL taps of a column shift in one workgroup (W in {16, 32}).
// dW[co][(ky, kx)][ci] = sum_p dY[p][co] * X[p + (ky-1) W + (kx-1)][ci]: for a fixed kx the three ky taps contract the SAME dY rows
// with X rows W pixels apart, and consecutive 32-pixel chunks slide that window by 32: every X pixel is DMA'd ONCE into a ring of
// eight 16-pixel units and read by the three taps from three chunks; the dY fragments are read once for three taps.  Tile = 128 co x
// (3 taps x 128 ci), 8 waves (2 x 4) of 64 co x (3 taps x 32 ci), 96 accumulator registers; L2 -> LDS bytes per MFMA a third of
// conv_ps_wgrad_kernel's.  (Round 4, late: the wave tile was 32 co x (3 x 64 ci) -- 4 dY + 3 x 8 X fragment reads per 18 MFMAs; the dY fragments are
// the ones shared by the three taps, so the taller tile reads 8 + 3 x 4 = 20 instead of 28: stand-alone within 1 %, CIFAR step -0.11 ... -0.19 ms on
// two boxes -- the kernel shares the chip with the data-gradient chain, and what it does not read from LDS it does not draw from the power budget.)
// A tap whose X row falls outside the image is skipped for that 16-pixel step (a step never straddles image rows: W % 16 == 0), so no
// zero rows are needed for the vertical direction; columns shifted out of the image read the zero page as before.
// LDS: two dY stages (2 x 16 KB) + the X ring (64 KB) = 96 KB, one workgroup per CU.
constexpr int WG3_RING_UNITS = 8, WG3_UNIT_BYTES = 16 * 512;
constexpr int WG3_LDS_BYTES = 2 * WG_OP_BYTES + WG3_RING_UNITS * WG3_UNIT_BYTES;
// Compile-time ablation of conv_ps_wgrad3_kernel (timing experiments, scripts/abl_wg3.sh; results are wrong by design): 1 no steady-state DMA,
// 2 no MFMA, 4 no X fragment reads, 16 no barrier, 32 no epilogue stores, 64 no main loop.  Compile-time because the run-time form of these
// switches (-DBD_PS_ABLATION, p.ablate) made this kernel 5x slower by itself; DESIGN.md section 3 has the decomposition they gave.
#ifndef BD_WG3_ABL
#define BD_WG3_ABL 0
#endif
constexpr int WG3_ABL = BD_WG3_ABL;

__global__ __launch_bounds__(512, 2) void conv_ps_wgrad3_kernel(PsWgParams p) {
    __shared__ __attribute__((aligned(128))) char smem[WG3_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;       // wave tile = 64 co (two 32-row blocks) x 3 taps x 32 ci: 20 fragment reads per 18 MFMAs (32 x 64: 28)
    int tm, tn, zz;
    {
        const unsigned L = blockIdx.x, T = gridDim.x, q = T >> 3;
        const unsigned j = L < (q << 3) ? (L & 7) * q + (L >> 3) : L;
        const unsigned ntiles = p.tiles_m * p.tiles_n;
        zz = j / ntiles;
        const unsigned tile = j - zz * ntiles;
        tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
    }
    const int co0 = tm * WG_BM;
    const int ncb = p.Cin / WG_BN;                 // 128-channel blocks of X
    const int kx = tn / ncb, ci0 = (tn - kx * ncb) * WG_BN;
    const int dxt = kx - 1;
    const int nchunks = p.P >> 5;                  // host: P % 32 == 0
    const int c_begin = zz * p.cps;
    int c_end = c_begin + p.cps;
    if (c_end > nchunks) c_end = nchunks;
    if constexpr ((WG3_ABL & 64) != 0) c_end = c_begin;

    // ---- DMA: wave w moves pixel pairs w and w + 8 of a 32-pixel chunk; lane: pixel k = 2*pair + lane/32, 16-byte slot lane%32
    const int ps = lane & 31;
    int kpix[2];
    const char* abase_g[2]; const char* bbase_g[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 2 * (wave + 8 * j) + (lane >> 5);
        kpix[j] = k;
        const int ls = ps ^ ((k & 3) << 2);
        abase_g[j] = p.dy + co0 * 4 + ls * 16;
        bbase_g[j] = p.x + ci0 * 4 + ls * 16;
    }
    char* const ring = smem + 2 * WG_OP_BYTES;
    auto issue_dy = [&](int c) {
        char* stage = smem + (c & 1) * WG_OP_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long pp = ps_v2r(c * 32 + kpix[j], p.lw, p.lh, p.lvw);       // chunk c = 32 VIRTUAL pixels (one strip row when W > 32)
            ps_dma16(abase_g[j] + pp * p.lddy * 4, stage + (wave + 8 * j) * 1024);
        }
    };
    const int VW = 1 << p.lvw;
    // X pixels [32 v + VW, 32 v + 32 + VW) (virtual order) of chunk v (v < c_begin: the halo): the rows the LAST vertical tap of chunk v needs
    auto issue_x = [&](int v) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s0 = v * 32 + VW + 2 * (wave + 8 * j);             // first (virtual) pixel of the pair (wave-uniform)
            const int s = s0 + (lane >> 5);
            const int sr = ps_v2r(s, p.lw, p.lh, p.lvw);                 // its place in memory
            const int x = sr & (p.W - 1);
            const bool ok = s >= 0 && s < p.P && (unsigned)(x + dxt) < (unsigned)p.W;
            const int unit = ((s0 + 4096) >> 4) & (WG3_RING_UNITS - 1);
            ps_dma16(ok ? bbase_g[j] + (long long)(sr + dxt) * p.ldx * 4 : reinterpret_cast<const char*>(kPsZero),
                     ring + unit * WG3_UNIT_BYTES + (s0 & 15) * 512);
        }
    };

    // ---- fragment addresses (ds_read_b64_tr_b16): as in conv_ps_wgrad_kernel, relative to a 16-pixel unit
    const int sl = lane & 15, hb = (lane >> 4) & 1, h = lane >> 5;
    const int kq = sl >> 2, rq = sl & 3;
    const int lane_base = (8 * h + kq) * 512 + hb * 32 + rq * 8;
    int xw[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) xw[v] = (v ^ kq) << 6;
    auto foff = [&](int t, int plane) { return lane_base + xw[(t & 1) * 2 + plane] + (t >> 1) * 256; };
    unsigned aoff[2][2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) aoff[i][pl] = (unsigned)foff(wm * 2 + i, pl);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) boff[pl] = (unsigned)foff(wn, pl);
    const unsigned smem_addr = (unsigned)(uintptr_t)(lds_ptr)smem;
    const unsigned ring_addr = smem_addr + 2 * WG_OP_BYTES;

    floatx16 acc[3][2], accb[2];          // acc[tap][co block]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accb[0][r] = 0.f; accb[1][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; }
    }
    const bool do_db = p.want_db && tn == 0 && wn == 0;   // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // one 16-pixel step of one tap: X fragments (one 32-channel tile x hi / lo x two pixel halves)
    struct BFrag { ps_short4 b0[2], b1[2]; };
    auto readB = [&](BFrag& f, unsigned xbase) {
        if constexpr ((WG3_ABL & 4) != 0) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f.b0[pl] = f.b1[pl] = ps_short4{0x3f80, 0x3f80, 0x3f80, 0x3f80};
            return;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            f.b0[pl] = ps_tr_read<0>(xbase + boff[pl]);
            f.b1[pl] = ps_tr_read<4 * 512>(xbase + boff[pl]);
        }
    };
    auto waitB = [&](BFrag& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.b0[0]), "+v"(f.b1[0]), "+v"(f.b0[1]), "+v"(f.b1[1])); };
    struct AFrag { ps_short4 a0[2][2], a1[2][2]; };      // [co block][plane]
    auto readA = [&](AFrag& f, unsigned dbase, int S) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                f.a0[i][pl] = S ? ps_tr_read<16 * 512>(dbase + aoff[i][pl]) : ps_tr_read<0>(dbase + aoff[i][pl]);
                f.a1[i][pl] = S ? ps_tr_read<16 * 512 + 4 * 512>(dbase + aoff[i][pl]) : ps_tr_read<4 * 512>(dbase + aoff[i][pl]);
            }
    };
    auto waitA = [&](AFrag& f) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.a0[0][0]), "+v"(f.a1[0][0]), "+v"(f.a0[0][1]), "+v"(f.a1[0][1]), "+v"(f.a0[1][0]), "+v"(f.a1[1][0]), "+v"(f.a0[1][1]),
                       "+v"(f.a1[1][1]));
    };
    auto tap_mfmas = [&](floatx16 (&a)[2], const bf16x8 (&ah)[2], const bf16x8 (&al)[2], const BFrag& f) {
        const bf16x8 bh = ps_tr_join(f.b0[0], f.b1[0]), bl = ps_tr_join(f.b0[1], f.b1[1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, a[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, a[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, a[i], 0, 0, 0);
    };

    const int wu = VW >> 4;        // units per (virtual) image row
    // six (16-pixel step, tap) stages per chunk, statically scheduled: the fragments of stage i+1 are read while the MFMAs of stage i
    // run.  A tap whose X row is outside the image skips its MFMAs (wave-uniform); its reads hit a live ring slot and are dropped.
    auto compute = [&](int c) {
        const unsigned dbase = smem_addr + (unsigned)((c & 1) * WG_OP_BYTES);
        const int p0 = c * 32;
        const int u0 = (p0 + 4096) >> 4;
        unsigned xb[6]; bool ok[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int S = i / 3, ky = i - 3 * S;
            const int y = ((p0 + 16 * S) >> p.lvw) & (p.H - 1);
            ok[i] = (unsigned)(y + ky - 1) < (unsigned)p.H;
            xb[i] = ring_addr + (unsigned)(((u0 + S + (ky - 1) * wu) & (WG3_RING_UNITS - 1)) * WG3_UNIT_BYTES);
        }
        AFrag fa[2]; BFrag fb[2];
        readA(fa[0], dbase, 0);
        readB(fb[0], xb[0]);
        waitA(fa[0]);
        waitB(fb[0]);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int S = i / 3, ky = i - 3 * S;
            if (i + 1 < 6) readB(fb[(i + 1) & 1], xb[i + 1]);
            if (i == 1) readA(fa[1], dbase, 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 ah[2] = {ps_tr_join(fa[S].a0[0][0], fa[S].a1[0][0]), ps_tr_join(fa[S].a0[1][0], fa[S].a1[1][0])};
            const bf16x8 al[2] = {ps_tr_join(fa[S].a0[0][1], fa[S].a1[0][1]), ps_tr_join(fa[S].a0[1][1], fa[S].a1[1][1])};
            if constexpr ((WG3_ABL & 2) != 0) { asm volatile("" ::"v"(ah[0]), "v"(al[0]), "v"(ah[1]), "v"(al[1]), "v"(fb[i & 1].b0[0]), "v"(fb[i & 1].b1[1])); } else
            if (ok[i]) tap_mfmas(acc[ky], ah, al, fb[i & 1]);
            if (do_db && ky == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accb[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[j], ones, accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[j], ones, accb[j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < 6) {
                if (i == 1) waitA(fa[1]);
                waitB(fb[(i + 1) & 1]);
            }
        }
    };

    if (c_begin < c_end) {
        for (int v = c_begin - (VW >> 4); v < c_begin; ++v) issue_x(v);      // halo: X rows [32 c_begin - VW, 32 c_begin + VW)
        issue_dy(c_begin); issue_x(c_begin);
        for (int c = c_begin; c < c_end; ++c) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if constexpr ((WG3_ABL & 16) == 0) __builtin_amdgcn_s_barrier();
            if (c + 1 < c_end && !((WG3_ABL & 1) && c > c_begin)) { issue_dy(c + 1); issue_x(c + 1); }
            compute(c);
        }
    }

    // ---- epilogue: lane holds column n = li of rows (r&3) + 8*(r>>2) + 4*h; three tap tiles per wave
    const int li = lane & 31;
    const int M = p.Cout, N = 9 * p.Cin;
    float* out = p.out + (p.ksplit > 1 ? (long long)zz * M * N : 0);
    if constexpr ((WG3_ABL & 32) != 0) {     // every accumulator stays live, nothing is stored
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += accb[0][r] + accb[1][r] + acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r] + acc[2][0][r] + acc[2][1][r];
        if (p.P < 0) out[lane] = t;
        return;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nn = (ky * 3 + kx) * p.Cin + ci0 + wn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = co0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(long long)m * N + nn] = acc[ky][i][r];
            }
        }
    if (do_db && li == 0) {
        float* o = p.ksplit > 1 ? p.out + (long long)p.ksplit * M * N + (long long)zz * M : p.db;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[co0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = accb[i][r];
    }
}

// fixed-order sum of the K-split slabs (float4 per thread); the tail threads fold the bias rows.  V3 only names the launch (the second pass of
// conv_ps_wgrad3_kernel is its own row in the profiles / PMC tables: bench.py's roofline.traffic of the dominant kernel counts ITS second pass).
template <bool V3>
__global__ __launch_bounds__(256) void conv_ps_wgrad_reduce(const float* __restrict__ part, int ksplit, long long mn, int M,
                                                            float* __restrict__ dw, float* __restrict__ db, int brows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n4 = mn >> 2;
    if (i < n4) {
        float4 a = *reinterpret_cast<const float4*>(part + 4 * i);
#pragma unroll 4      // four slab loads in flight; the additions keep their fixed order
        for (int s = 1; s < ksplit; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(part + (long long)s * mn + 4 * i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(dw + 4 * i) = a;
    } else if (db && i - n4 < M) {
        const long long m = i - n4;
        float v = 0.f;
        for (int s = 0; s < brows; ++s) v += part[(long long)ksplit * mn + (long long)s * M + m];   // brows = ksplit (x 4 classes in the PHASE form)
        db[m] = v;
    }
}

// ---- producers of split planes -------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ps_pack_hi(float a, float b) { return bd_pack_hi(a, b); }   // common.h: the library's split
__device__ __forceinline__ unsigned ps_pack_lo(float a, float b) { return bd_pack_lo(a, b); }

// rows x C fp32 (row stride lds) -> split planes (row stride ldd elements = 4*ldd bytes); C % 32 == 0.  8 values per thread.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, long long lds, long long rows, int C,
                                                          unsigned short* __restrict__ dst, long long ldd) {
    const int c8 = C >> 3;
    const long long total = rows * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c8;
        const int j = (int)(i - r * c8);   // 8-channel group
        const float4 a = *reinterpret_cast<const float4*>(src + r * lds + j * 8);
        const float4 b = *reinterpret_cast<const float4*>(src + r * lds + j * 8 + 4);
        unsigned short* o = dst + 2 * r * ldd + (j >> 2) * 64 + (j & 3) * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(ps_pack_hi(a.x, a.y), ps_pack_hi(a.z, a.w), ps_pack_hi(b.x, b.y), ps_pack_hi(b.z, b.w));
        *reinterpret_cast<uint4*>(o + 32) = make_uint4(ps_pack_lo(a.x, a.y), ps_pack_lo(a.z, a.w), ps_pack_lo(b.x, b.y), ps_pack_lo(b.z, b.w));
    }
}

// nearest-neighbour x2 upsampling folded into the split: src [B, H, W, C] fp32 -> split planes of [B, 2H, 2W, C]
// (Upsample2D, resnet.py:126-161: F.interpolate(scale_factor=2, mode="nearest") feeding a 3x3 conv)
__global__ __launch_bounds__(256) void split_rows_ups2_kernel(const float* __restrict__ src, long long lds, int B, int H, int W, int C,
                                                               unsigned short* __restrict__ dst, long long ldd) {
    const int c8 = C >> 3;
    const long long total = (long long)B * 4 * H * W * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c8;                      // output pixel (b, y, x) over the 2H x 2W grid
        const int j = (int)(i - r * c8);
        const int x = (int)(r % (2 * W));
        const long long t = r / (2 * W);
        const int y = (int)(t % (2 * H));
        const long long b = t / (2 * H);
        const long long rs = (b * H + (y >> 1)) * W + (x >> 1);
        const float4 a = *reinterpret_cast<const float4*>(src + rs * lds + j * 8);
        const float4 bb = *reinterpret_cast<const float4*>(src + rs * lds + j * 8 + 4);
        unsigned short* o = dst + 2 * r * ldd + (j >> 2) * 64 + (j & 3) * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(ps_pack_hi(a.x, a.y), ps_pack_hi(a.z, a.w), ps_pack_hi(bb.x, bb.y), ps_pack_hi(bb.z, bb.w));
        *reinterpret_cast<uint4*>(o + 32) = make_uint4(ps_pack_lo(a.x, a.y), ps_pack_lo(a.z, a.w), ps_pack_lo(bb.x, bb.y), ps_pack_lo(bb.z, bb.w));
    }
}

// conv weights W[co][tap][ci] fp32 -> split planes of the transpose Wt[ci][tap][co] (rows ci, K = (tap, co)); Cin, Cout % 32 == 0.
// One workgroup = one (tap, 32 co x 32 ci) block through LDS.
__global__ __launch_bounds__(256) void split_wT_kernel(const float* __restrict__ w, int Cin, int Cout, unsigned short* __restrict__ out) {
    __shared__ float t[32][33];
    const int cib = blockIdx.x, cob = blockIdx.y, tap = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cob * 32 + ty + 8 * i;
        t[ty + 8 * i][tx] = w[((long long)co * 9 + tap) * Cin + cib * 32 + tx];
    }
    __syncthreads();
    // thread -> (ci row r = tid/8, 4 consecutive co = (tid%8)*4)
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const float v0 = t[c4][r], v1 = t[c4 + 1][r], v2 = t[c4 + 2][r], v3 = t[c4 + 3][r];
    unsigned short* o = out + 2 * (((long long)(cib * 32 + r) * 9 + tap) * Cout + cob * 32) + c4;
    *reinterpret_cast<uint2*>(o) = make_uint2(ps_pack_hi(v0, v1), ps_pack_hi(v2, v3));
    *reinterpret_cast<uint2*>(o + 32) = make_uint2(ps_pack_lo(v0, v1), ps_pack_lo(v2, v3));
}

// all 3x3 conv weights of a network in ONE launch: the table (by value in the kernel arguments) lists, per weight,
// its element offset in the flat fp32 buffer (== its offset in the transposed split copy), the channel counts and the
// first block of the 1-D grid that belongs to it
struct WtTable {
    int n;
    long long off[BD_WT_MAX];
    int cin[BD_WT_MAX], cout[BD_WT_MAX], blk0[BD_WT_MAX + 1];
};
__global__ __launch_bounds__(256) void split_wT_batched_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, WtTable t) {
    __shared__ float tl[32][33];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.blk0[e + 1]) ++e;
    const int Cin = t.cin[e], Cout = t.cout[e];
    const float* w = params + t.off[e];
    unsigned short* o_base = out + 2 * t.off[e];
    int b = blockIdx.x - t.blk0[e];
    const int ncib = Cin / 32, ncob = Cout / 32;
    const int cib = b % ncib; b /= ncib;
    const int cob = b % ncob; const int tap = b / ncob;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cob * 32 + ty + 8 * i;
        tl[ty + 8 * i][tx] = w[((long long)co * 9 + tap) * Cin + cib * 32 + tx];
    }
    __syncthreads();
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const float v0 = tl[c4][r], v1 = tl[c4 + 1][r], v2 = tl[c4 + 2][r], v3 = tl[c4 + 3][r];
    unsigned short* o = o_base + 2 * (((long long)(cib * 32 + r) * 9 + tap) * Cout + cob * 32) + c4;
    *reinterpret_cast<uint2*>(o) = make_uint2(ps_pack_hi(v0, v1), ps_pack_hi(v2, v3));
    *reinterpret_cast<uint2*>(o + 32) = make_uint2(ps_pack_lo(v0, v1), ps_pack_lo(v2, v3));
}

int split_wt_batched(const float* params, uint16_t* out, const long long* off, const int* cin, const int* cout, int n, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += BD_WT_MAX) {
        WtTable t;
        t.n = n - i0 < BD_WT_MAX ? n - i0 : BD_WT_MAX;
        int blk = 0;
        for (int i = 0; i < t.n; ++i) {
            t.off[i] = off[i0 + i]; t.cin[i] = cin[i0 + i]; t.cout[i] = cout[i0 + i]; t.blk0[i] = blk;
            blk += 9 * (t.cin[i] / 32) * (t.cout[i] / 32);
        }
        t.blk0[t.n] = blk;
        hipLaunchKernelGGL(split_wT_batched_kernel, dim3((unsigned)blk), dim3(256), 0, st, params, out, t);
        BD_LAUNCH_CHECK("split_wT_batched");
    }
    return BD_OK;
}

static int ilog2x(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// shapes the LDS-DMA convolution handles: power-of-two images, whole 32-channel blocks, 128-wide output tiles.
// Large layers (>= 96 tiles of 256 x 128) run the three-stage 256 x 128 kernel, one workgroup per CU and no K split;
// small ones (8x8 / 4x4 images at CIFAR batch sizes) the 128 x 128 split-K variant.
bool conv3x3_ps_supported(int B, int H, int W, int K_channels, int N_channels) {
    return B > 0 && ilog2x(H) >= 0 && ilog2x(W) >= 0 && K_channels % 32 == 0 && N_channels % PS_BN == 0 && K_channels >= 32;
}
static bool ps_large(long long M, int N) {
    static const int force = getenv("BD_PS_TILE") ? atoi(getenv("BD_PS_TILE")) : 0;   // 128 / 256 force a variant (A/B)
    if (force == 128) return false;
    if (force == 256) return true;
    return cdiv(M, PS_BM) * (N / PS_BN) >= 96;
}
static void ps_small_split(long long M, int N, int K, int& ksplit, int& cps) {
    const long long tiles = cdiv(M, 128) * (N / 128);
    const int nchunks = 9 * (K / 32);
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* e = getenv("BD_PS_SMALL_SLOTS");
        return e ? atoi(e) : cus;   // (2 x CUs measured 0.1-0.2 ms/step slower: twice the slab traffic for the 4x4 / 8x8 layers)
    }();
    static const int mincps = getenv("BD_PS_SMALL_MINCPS") ? atoi(getenv("BD_PS_SMALL_MINCPS")) : 4;
    int ks = (int)(slots / tiles);
    if (ks < 1) ks = 1;
    if (ks > nchunks / mincps) ks = nchunks / mincps;
    if (ks < 1) ks = 1;
    cps = (int)cdiv(nchunks, ks);
    ksplit = (int)cdiv(nchunks, cps);
}
size_t conv3x3_ps_workspace_bytes(const bd_conv3x3_ps_desc& d) {
    const long long M = (long long)d.B * d.H * d.W;
    if (d.N % PS_BN || ps_large(M, d.N)) return 0;
    int ks, cps;
    ps_small_split(M, d.N, d.K, ks, cps);
    return ks > 1 ? (size_t)ks * (size_t)M * d.N * sizeof(float) : 0;
}

// does this call run on conv_ps3_kernel (256 x 128 tiles, whole tiles only)?
static bool ps_takes_v3(int B, int H, int W, int N) {
    static const bool v3_off = getenv("BD_PS_V3") && atoi(getenv("BD_PS_V3")) == 0;
    // round 4: any W >= 16 -- images wider than 32 pixels are walked strip by strip (virtual pixel order, ps_v2r)
    static const int v3_maxw = getenv("BD_PS_V3_MAXW") ? atoi(getenv("BD_PS_V3_MAXW")) : (1 << 30);     // (A/B knob: 32 = round 3's gate)
    const long long M = (long long)B * H * W;
    const int vw = W < 32 ? W : 32;
    return ps_large(M, N) && !v3_off && W >= 16 && W <= v3_maxw && ((long long)H * vw) % PS_BM == 0 && M % PS_BM == 0;
}
// Round 4: pixel splits of the GroupNorm partials a forward call can write from its epilogue (0: it cannot).  One split per 256-pixel tile.
int conv3x3_ps_gn_splits(int B, int H, int W, int K, int N, int groups) {
    static const bool off = getenv("BD_GN_EPI_STATS") && atoi(getenv("BD_GN_EPI_STATS")) == 0;       // (A/B knob)
    if (off || B <= 0 || ilog2x(H) < 0 || ilog2x(W) < 0 || K <= 0 || K % 32 || N <= 0 || N % PS_BN || groups <= 0 || N % groups) return 0;
    const int lc = ilog2x(N / groups);
    if (lc < 2 || lc > 5 || (H * W) % PS_BM) return 0;
    return ps_takes_v3(B, H, W, N) ? H * W / PS_BM : 0;
}

int conv3x3_ps(const bd_conv3x3_ps_desc& d, hipStream_t st) {
    BD_CHECK(d.x_split && d.w_split && d.y, BD_ERR_INVALID, "conv3x3_ps: null pointer");
    BD_CHECK(d.B > 0 && ilog2x(d.H) >= 0 && ilog2x(d.W) >= 0, BD_ERR_UNSUPPORTED, "conv3x3_ps: H, W must be powers of two");
    BD_CHECK(d.K > 0 && d.K % 32 == 0 && d.N > 0 && d.N % PS_BN == 0, BD_ERR_UNSUPPORTED,
             "conv3x3_ps: K channels %% 32 and N channels %% %d must be 0 (got %d, %d)", PS_BN, d.K, d.N);
    BD_CHECK(d.ldx % 32 == 0 && ((uintptr_t)d.x_split & 127) == 0 && ((uintptr_t)d.w_split & 127) == 0, BD_ERR_UNSUPPORTED,
             "conv3x3_ps: split planes need ld %% 32 == 0 and 128-byte aligned bases");
    BD_CHECK(d.direction == 1 || d.direction == -1, BD_ERR_INVALID, "conv3x3_ps: direction must be +1 or -1");
    // the epilogues and the split-K second pass move float4: every fp32 operand needs 16-byte rows
    BD_CHECK(d.ldy % 4 == 0 && ((uintptr_t)d.y & 15) == 0, BD_ERR_UNSUPPORTED, "conv3x3_ps: y needs ldy %% 4 == 0 and a 16-byte aligned base");
    BD_CHECK(!d.residual || (d.ldr % 4 == 0 && ((uintptr_t)d.residual & 15) == 0), BD_ERR_UNSUPPORTED,
             "conv3x3_ps: residual needs ldr %% 4 == 0 and a 16-byte aligned base");
    BD_CHECK(!d.rowbias || (d.ld_rowbias % 4 == 0 && ((uintptr_t)d.rowbias & 15) == 0), BD_ERR_UNSUPPORTED,
             "conv3x3_ps: rowbias needs ld_rowbias %% 4 == 0 and a 16-byte aligned base");
    BD_CHECK(!d.bias || ((uintptr_t)d.bias & 15) == 0, BD_ERR_UNSUPPORTED, "conv3x3_ps: bias needs a 16-byte aligned base");
    const long long M = (long long)d.B * d.H * d.W;
    BD_CHECK(M < (1ll << 31), BD_ERR_UNSUPPORTED, "conv3x3_ps: pixel count overflows int32");
    BD_CHECK((long long)(d.W + 1) * d.ldx * 4 < (1ll << 31), BD_ERR_UNSUPPORTED, "conv3x3_ps: row pitch too large");
    PsParams p = {};
    p.a = reinterpret_cast<const char*>(d.x_split); p.w = reinterpret_cast<const char*>(d.w_split);
    p.y = d.y; p.bias = d.bias; p.rowbias = d.rowbias; p.residual = d.residual;
    p.lda = d.ldx; p.ldy = d.ldy; p.ldr = d.ldr; p.ld_rowbias = d.ld_rowbias;
    p.C = d.K; p.H = d.H; p.W = d.W; p.lw = ilog2x(d.W); p.lhw = ilog2x(d.W) + ilog2x(d.H);
    p.sign = d.direction; p.M = (int)M; p.N = d.N;
    p.out_scale = d.out_scale == 0.f ? 1.f : d.out_scale; p.accumulate = d.accumulate;
    const bool large = ps_large(M, d.N);
    int rec = -1;
    if (prof_on()) {
        rec = prof_begin(d.direction > 0 ? (large ? "conv_ps_fwd" : "conv_ps128_fwd") : (large ? "conv_ps_dgrad" : "conv_ps128_dgrad"),
                         2.0 * (double)M * d.N * 9.0 * d.K, ((double)M * d.K + 9.0 * d.K * d.N + (double)M * d.N) * 4.0, st);
    }
    const int epi = (d.residual ? 1 : 0) | (d.rowbias ? 2 : 0) | (d.accumulate ? 4 : 0);
    if (large) {
        p.tiles_m = (int)cdiv(M, PS_BM); p.tiles_n = d.N / PS_BN;
        const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(PS_NT);
#ifdef BD_PS_ABLATION
        p.ablate = getenv("BD_PS_ABLATE") ? atoi(getenv("BD_PS_ABLATE")) : 0;
#endif
        // vertical-tap sharing variant (conv_ps3_kernel): image rows of 16 or 32 pixels, whole images per tile row block
        const int vw = d.W < 32 ? d.W : 32;
        const bool v3 = ps_takes_v3(d.B, d.H, d.W, d.N);
        p.lvw = ilog2x(vw);
        if (d.gn_part) {
            BD_CHECK(d.direction == 1 && (epi == 1 || epi == 2) && conv3x3_ps_gn_splits(d.B, d.H, d.W, d.K, d.N, d.gn_groups) > 0, BD_ERR_UNSUPPORTED,
                     "conv3x3_ps: gn_part needs a forward call with exactly one of rowbias / residual and bd_conv3x3_ps_gn_splits() > 0");
            p.gn_part = d.gn_part; p.gn_G = d.gn_groups; p.gn_lcpg = ilog2x(d.N / d.gn_groups);
        }
#define PS_LAUNCH(E) do { if (v3) hipLaunchKernelGGL((conv_ps3_kernel<E>), grid, block, 0, st, p); \
                          else hipLaunchKernelGGL((conv_ps_kernel<E>), grid, block, 0, st, p); } while (0)
        switch (epi) {          // every combination has its own instantiation: the epilogue's addends are compile-time
            case 0: PS_LAUNCH(0); break;
            case 1: PS_LAUNCH(1); break;
            case 2: PS_LAUNCH(2); break;
            case 3: PS_LAUNCH(3); break;
            case 4: PS_LAUNCH(4); break;
            case 5: PS_LAUNCH(5); break;
            case 6: PS_LAUNCH(6); break;
            default: PS_LAUNCH(7); break;
        }
#undef PS_LAUNCH
        BD_LAUNCH_CHECK("conv_ps");
    } else {
        BD_CHECK(!d.gn_part, BD_ERR_UNSUPPORTED, "conv3x3_ps: gn_part on a small layer (bd_conv3x3_ps_gn_splits() == 0)");
        PsSmallParams pp = {};
        p.tiles_m = (int)cdiv(M, 128); p.tiles_n = d.N / 128;
        pp.q = p;
        ps_small_split(M, d.N, d.K, pp.ksplit, pp.cps);
        if (pp.ksplit > 1) {
            const size_t need = (size_t)pp.ksplit * (size_t)M * d.N * sizeof(float);
            BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "conv3x3_ps: split-K needs %zu workspace bytes, got %zu", need,
                     d.workspace_bytes);
            pp.partial = reinterpret_cast<float*>(d.workspace);
        }
        const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * pp.ksplit)), block(512);
        // round 6: short K slices (the 4 x 4 level) take the four-stage form (three chunks in flight, one workgroup per CU)
        static const int deep_maxcps = getenv("BD_PS128_DEEP_MAXCPS") ? atoi(getenv("BD_PS128_DEEP_MAXCPS")) : 9;       // (A/B knob: 0 = off)
        if (pp.ksplit > 1 && pp.cps <= deep_maxcps) {
            hipLaunchKernelGGL((conv_ps128_kernel<0, 4>), grid, block, 0, st, pp);
        } else
        switch (pp.ksplit > 1 ? 0 : epi) {
            case 0: hipLaunchKernelGGL(conv_ps128_kernel<0>, grid, block, 0, st, pp); break;
            case 1: hipLaunchKernelGGL(conv_ps128_kernel<1>, grid, block, 0, st, pp); break;
            case 2: hipLaunchKernelGGL(conv_ps128_kernel<2>, grid, block, 0, st, pp); break;
            case 3: hipLaunchKernelGGL(conv_ps128_kernel<3>, grid, block, 0, st, pp); break;
            case 4: hipLaunchKernelGGL(conv_ps128_kernel<4>, grid, block, 0, st, pp); break;
            case 5: hipLaunchKernelGGL(conv_ps128_kernel<5>, grid, block, 0, st, pp); break;
            case 6: hipLaunchKernelGGL(conv_ps128_kernel<6>, grid, block, 0, st, pp); break;
            default: hipLaunchKernelGGL(conv_ps128_kernel<7>, grid, block, 0, st, pp); break;
        }
        BD_LAUNCH_CHECK("conv_ps128");
        if (pp.ksplit > 1) {
            hipLaunchKernelGGL(conv_ps128_reduce, dim3((unsigned)cdiv(M * (d.N / 4), 256)), dim3(256), 0, st, pp);
            BD_LAUNCH_CHECK("conv_ps128_reduce");
        }
    }
    prof_end(rec, st);
    return BD_OK;
}

bool conv3x3_ps_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && ilog2x(H) >= 0 && ilog2x(W) >= 0 && Cin % WG_BN == 0 && Cout % WG_BM == 0;
}
// vertical-tap sharing form (conv_ps_wgrad3_kernel): image rows of 16 or 32 pixels, whole 32-pixel chunks
static bool ps_wgrad_v3(const bd_conv3x3_ps_wgrad_desc& d) {
    static const bool off = getenv("BD_PS_WG3") && atoi(getenv("BD_PS_WG3")) == 0;
    static const int maxw = getenv("BD_PS_WG3_MAXW") ? atoi(getenv("BD_PS_WG3_MAXW")) : (1 << 30);        // (A/B knob: 32 = round 3's gate)
    // round 4: any W >= 16 -- wider images are walked strip by strip (virtual pixel order, ps_v2r): a chunk = one 32-pixel strip row
    return !off && d.W >= 16 && d.W <= maxw && ((long long)d.B * d.H * d.W) % 32 == 0;
}
static int g_wg3_slots_override = 0;     // 0: environment / default (set by bd_tune_set)
static void ps_wgrad_split(const bd_conv3x3_ps_wgrad_desc& d, int& ksplit, int& cps) {
    const bool v3 = ps_wgrad_v3(d);
    const long long tiles = (long long)(d.Cout / WG_BM) * ((v3 ? 3 : 9) * d.Cin / WG_BN);
    const int nchunks = (int)cdiv((long long)d.B * d.H * d.W, 32);
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* e = getenv("BD_PS_WG_SLOTS");
        return e ? atoi(e) : 2 * cus;   // two 64-KB workgroups per CU (two LDS stages each): the pair de-phases, 141 vs 189 us
    }();
    // 96 KB of LDS: one workgroup per CU -- and three quarters of the CUs (round 4, late: 192 of 256).  The kernel never runs alone: the data-gradient
    // chain shares the chip with it, so the CUs it leaves are not idle, and a quarter fewer slabs are a quarter less slab traffic: CIFAR step
    // 17.71 -> 17.53 ms and 18.01 -> 17.89 on two boxes (160 / 176 / 208 / 224 / 256 slots: 17.70 / 17.59 / 17.67 / 17.72 / 17.71), 256 x 256 28.06 -> 27.93.
    // Images walked strip by strip (W > 32: the 256 x 256 network at B = 4, whose main stream carries the three-pass GroupNorm and 2 048-tile data
    // gradients): HALF the CUs -- 27.14 -> 26.69 ms per step (144 / 160 / 176 slots: 27.52 / 27.51 / 27.74; 112 / 96 / 64: 27.21 / 28.16 / 32.9), while the
    // CIFAR step wants its 192 (128: 18.08 against 17.89).
    static const int slots3_getenv = getenv("BD_PS_WG3_SLOTS") ? atoi(getenv("BD_PS_WG3_SLOTS")) : 0;
    const int slots3_env = g_wg3_slots_override > 0 ? g_wg3_slots_override : slots3_getenv;      // bd_tune_set("ps_wg3_slots", n) wins over the environment
    const int slots3 = slots3_env > 0 ? slots3_env : (d.W > 32 ? slots / 4 : slots * 3 / 8);
    static const int mincps = getenv("BD_PS_WG_MINCPS") ? atoi(getenv("BD_PS_WG_MINCPS")) : 8;   // chunks per split at least (4x4 layers: 8 slabs instead of 14; 4 / 16 measured +0.2 / +0.1 ms)
    int ks = (int)((v3 ? slots3 : slots) / tiles);
    if (ks < 1) ks = 1;
    if (ks > nchunks / mincps) ks = nchunks / mincps > 0 ? nchunks / mincps : 1;
    cps = (int)cdiv(nchunks, ks);
    ksplit = (int)cdiv(nchunks, cps);
}
size_t conv3x3_ps_wgrad_workspace_bytes(const bd_conv3x3_ps_wgrad_desc& d) {
    int ks, cps;
    ps_wgrad_split(d, ks, cps);
    return ks > 1 ? (size_t)ks * ((size_t)d.Cout * 9 * d.Cin + d.Cout) * sizeof(float) : 0;
}
int conv3x3_ps_wgrad(const bd_conv3x3_ps_wgrad_desc& d, hipStream_t st) {
    BD_CHECK(d.x_split && d.dy_split && d.dw, BD_ERR_INVALID, "conv3x3_ps_wgrad: null pointer");
    BD_CHECK(d.B > 0 && conv3x3_ps_wgrad_supported(d.B, d.H, d.W, d.Cin, d.Cout), BD_ERR_UNSUPPORTED,
             "conv3x3_ps_wgrad: needs power-of-two H, W, Cin and Cout multiples of 128");
    BD_CHECK(d.ldx % 32 == 0 && d.lddy % 32 == 0 && ((uintptr_t)d.x_split & 127) == 0 && ((uintptr_t)d.dy_split & 127) == 0,
             BD_ERR_UNSUPPORTED, "conv3x3_ps_wgrad: split planes need ld %% 32 == 0 and 128-byte aligned bases");
    PsWgParams p = {};
    p.dy = reinterpret_cast<const char*>(d.dy_split); p.x = reinterpret_cast<const char*>(d.x_split);
    p.lddy = d.lddy; p.ldx = d.ldx; p.Cin = d.Cin; p.Cout = d.Cout; p.H = d.H; p.W = d.W; p.lw = ilog2x(d.W);
    p.lh = ilog2x(d.H); p.lvw = ilog2x(d.W < 32 ? d.W : 32);
    p.P = d.B * d.H * d.W;
    const bool v3 = ps_wgrad_v3(d);
    p.tiles_m = d.Cout / WG_BM; p.tiles_n = (v3 ? 3 : 9) * d.Cin / WG_BN; p.ntaps = 9;
    ps_wgrad_split(d, p.ksplit, p.cps);
    p.want_db = d.db != nullptr;
    const long long mn = (long long)d.Cout * 9 * d.Cin;
    if (p.ksplit > 1) {
        const size_t need = (size_t)p.ksplit * ((size_t)mn + d.Cout) * sizeof(float);
        BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "conv3x3_ps_wgrad: workspace %zu < %zu", d.workspace_bytes, need);
        p.out = reinterpret_cast<float*>(d.workspace);
    } else {
        p.out = d.dw; p.db = d.db;
    }
    int rec = -1;
    if (prof_on())
        rec = prof_begin(v3 ? "conv_ps_wgrad3" : "conv_ps_wgrad", 2.0 * (double)p.P * d.Cout * 9.0 * d.Cin,   // (round 6: the large layers' kernel is its own class)
                         ((double)p.P * (d.Cin + d.Cout) + 9.0 * d.Cin * d.Cout) * 4.0, st);
#ifdef BD_PS_ABLATION
    p.ablate = getenv("BD_PS_ABLATE") ? atoi(getenv("BD_PS_ABLATE")) : 0;
#endif
    static const int stages = getenv("BD_PS_WG_STAGES") ? atoi(getenv("BD_PS_WG_STAGES")) : 2;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.ksplit));
    // (NW = 4 -- 2 x 2 waves of 64 x 64 -- measured within +-2 % of NW = 8 on every layer; one form is kept)
#ifdef BD_PS_ABLATION
    static const int nw = getenv("BD_PS_WG_NW") ? atoi(getenv("BD_PS_WG_NW")) : 8;
    if (nw == 4 && stages == 2) hipLaunchKernelGGL((conv_ps_wgrad_kernel<2, 4>), grid, dim3(256), 0, st, p);
    else if (nw == 4) hipLaunchKernelGGL((conv_ps_wgrad_kernel<3, 4>), grid, dim3(256), 0, st, p);
    else
#endif
    if (v3) hipLaunchKernelGGL(conv_ps_wgrad3_kernel, grid, dim3(512), 0, st, p);
    else if (stages == 2) hipLaunchKernelGGL((conv_ps_wgrad_kernel<2, 8>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_ps_wgrad_kernel<3, 8>), grid, dim3(512), 0, st, p);
    BD_LAUNCH_CHECK("conv_ps_wgrad");
    if (p.ksplit > 1) {
        const long long total = mn / 4 + (d.db ? d.Cout : 0);
        if (v3) hipLaunchKernelGGL(conv_ps_wgrad_reduce<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, p.out, p.ksplit, mn, d.Cout, d.dw, d.db, p.ksplit);
        else hipLaunchKernelGGL(conv_ps_wgrad_reduce<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, p.out, p.ksplit, mn, d.Cout, d.dw, d.db, p.ksplit);
        BD_LAUNCH_CHECK("conv_ps_wgrad_reduce");
    }
    prof_end(rec, st);
    return BD_OK;
}

// ---- PHASE form of the weight gradient: the upsample convolution (conv_ph.hip) ---------------------------------------------------
int ups_dweff_combine(const float* de, int Cin, int Cout, float* dw, hipStream_t st);   // conv_ph.hip

static void ups_wgrad_split(const bd_upsample_conv_desc& d, int& ksplit, int& cps) {
    bd_conv3x3_ps_wgrad_desc q = {};
    q.B = d.B; q.H = d.H; q.W = d.W; q.Cin = d.Cin; q.Cout = d.Cout;
    const long long tiles = (long long)(d.Cout / WG_BM) * (16 * d.Cin / WG_BN);
    const int nchunks = (int)cdiv((long long)d.B * d.H * d.W, 32);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int ks = (int)(2 * cus / tiles);
    if (ks < 1) ks = 1;
    if (ks > nchunks / 8) ks = nchunks / 8 > 0 ? nchunks / 8 : 1;
    cps = (int)cdiv(nchunks, ks);
    ksplit = (int)cdiv(nchunks, cps);
}
size_t upsample_conv_wgrad_workspace_bytes(const bd_upsample_conv_desc& d) {
    int ks, cps;
    ups_wgrad_split(d, ks, cps);
    const size_t mn = (size_t)d.Cout * 16 * d.Cin;
    return ((size_t)ks * (mn + 4 * (size_t)d.Cout) + mn) * sizeof(float);       // K-split slabs (+ 4 bias rows each) + dE
}
// dw [Cout,3,3,Cin] (+ db) of y = conv3x3(nearest_up2(x)) from x (split planes, SOURCE grid) and dY (split planes, fine grid)
int upsample_conv_wgrad(const bd_upsample_conv_desc& d, hipStream_t st) {
    BD_CHECK(d.x_split && d.dy_split && d.dw, BD_ERR_INVALID, "bd_upsample_conv_wgrad: null pointer");
    BD_CHECK(d.B > 0 && conv3x3_ps_wgrad_supported(d.B, d.H, d.W, d.Cin, d.Cout), BD_ERR_UNSUPPORTED,
             "bd_upsample_conv_wgrad: needs power-of-two H, W, Cin and Cout multiples of 128");
    BD_CHECK(d.ldx % 32 == 0 && d.lddy % 32 == 0 && ((uintptr_t)d.x_split & 127) == 0 && ((uintptr_t)d.dy_split & 127) == 0,
             BD_ERR_UNSUPPORTED, "bd_upsample_conv_wgrad: split planes need ld %% 32 == 0 and 128-byte aligned bases");
    BD_CHECK((long long)d.B * d.H * d.W * 4 < (1ll << 31), BD_ERR_UNSUPPORTED, "bd_upsample_conv_wgrad: pixel count overflows int32");
    const size_t need = upsample_conv_wgrad_workspace_bytes(d);
    BD_CHECK(d.workspace && d.workspace_bytes >= need, BD_ERR_WORKSPACE, "bd_upsample_conv_wgrad: workspace %zu < %zu", d.workspace_bytes, need);
    PsWgParams p = {};
    p.dy = reinterpret_cast<const char*>(d.dy_split); p.x = reinterpret_cast<const char*>(d.x_split);
    p.lddy = d.lddy; p.ldx = d.ldx; p.Cin = d.Cin; p.Cout = d.Cout; p.H = d.H; p.W = d.W; p.lw = ilog2x(d.W);
    p.P = d.B * d.H * d.W;
    p.tiles_m = d.Cout / WG_BM; p.tiles_n = 16 * d.Cin / WG_BN; p.ntaps = 16;
    ups_wgrad_split(d, p.ksplit, p.cps);
    p.want_db = d.db != nullptr;
    const long long mn = (long long)d.Cout * 16 * d.Cin;
    p.out = reinterpret_cast<float*>(d.workspace);
    float* de = p.out + (size_t)p.ksplit * ((size_t)mn + 4 * (size_t)d.Cout);
    int rec = -1;
    if (prof_on())
        rec = prof_begin("conv_ph_ups_wgrad", 2.0 * (double)p.P * 4.0 * d.Cout * 9.0 * d.Cin,
                         ((double)p.P * (d.Cin + 4.0 * d.Cout) + 9.0 * d.Cin * d.Cout) * 4.0, st);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.ksplit));
    hipLaunchKernelGGL((conv_ps_wgrad_kernel<2, 8, true>), grid, dim3(512), 0, st, p);
    BD_LAUNCH_CHECK("conv_ps_wgrad (phase)");
    const long long total = mn / 4 + (d.db ? d.Cout : 0);
    hipLaunchKernelGGL(conv_ps_wgrad_reduce<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, p.out, p.ksplit, mn, d.Cout, de, d.db, 4 * p.ksplit);
    BD_LAUNCH_CHECK("conv_ps_wgrad_reduce (phase)");
    BD_TRY(ups_dweff_combine(de, d.Cin, d.Cout, d.dw, st));
    prof_end(rec, st);
    return BD_OK;
}

int g_tune_gen = 0;
int tune_set(const char* key, int value) {
    BD_CHECK(key, BD_ERR_INVALID, "bd_tune_set: null key");
    if (!strcmp(key, "ps_wg3_slots")) { g_wg3_slots_override = value > 0 ? value : 0; ++g_tune_gen; return BD_OK; }
    set_error("bd_tune_set: unknown key '%s' (known: ps_wg3_slots)", key);
    return BD_ERR_INVALID;
}
}  // namespace bd

extern "C" int bd_tune_set(const char* key, int value) { return bd::tune_set(key, value); }
extern "C" size_t bd_upsample_conv_wgrad_workspace_bytes(const bd_upsample_conv_desc* d) {
    return d ? bd::upsample_conv_wgrad_workspace_bytes(*d) : 0;
}
extern "C" int bd_upsample_conv_wgrad(const bd_upsample_conv_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_upsample_conv_wgrad: null descriptor");
    return bd::upsample_conv_wgrad(*d, bd::S(s));
}
extern "C" size_t bd_conv3x3_ps_workspace_bytes(const bd_conv3x3_ps_desc* d) { return d ? bd::conv3x3_ps_workspace_bytes(*d) : 0; }
extern "C" int bd_conv3x3_ps_gn_splits(int B, int H, int W, int K, int N, int groups) { return bd::conv3x3_ps_gn_splits(B, H, W, K, N, groups); }
extern "C" int bd_conv3x3_ps(const bd_conv3x3_ps_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_ps: null descriptor");
    return bd::conv3x3_ps(*d, bd::S(s));
}
extern "C" int bd_conv3x3_ps_wgrad(const bd_conv3x3_ps_wgrad_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_ps_wgrad: null descriptor");
    return bd::conv3x3_ps_wgrad(*d, bd::S(s));
}
extern "C" size_t bd_conv3x3_ps_wgrad_workspace_bytes(const bd_conv3x3_ps_wgrad_desc* d) {
    return d ? bd::conv3x3_ps_wgrad_workspace_bytes(*d) : 0;
}
extern "C" int bd_split_rows(const float* src, int64_t ld_src, int64_t rows, int C, uint16_t* dst, int64_t ld_dst, bd_stream_t stream) {
    BD_CHECK(src && dst && rows > 0 && C > 0, BD_ERR_INVALID, "bd_split_rows: bad args");
    BD_CHECK(C % 32 == 0 && ld_dst % 32 == 0 && (ld_src & 3) == 0 && bd::aligned16(src) && ((uintptr_t)dst & 127) == 0, BD_ERR_UNSUPPORTED,
             "bd_split_rows: C, ld_dst %% 32, ld_src %% 4, 16-byte aligned src and 128-byte aligned dst required");
    long long nb = bd::cdiv(rows * (C / 8), 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bd::split_rows_kernel, dim3((unsigned)nb), dim3(256), 0, bd::S(stream), src, (long long)ld_src, (long long)rows, C, dst,
                       (long long)ld_dst);
    BD_LAUNCH_CHECK("split_rows");
    return BD_OK;
}
extern "C" int bd_split_rows_ups2(const float* src, int64_t ld_src, int B, int H, int W, int C, uint16_t* dst, int64_t ld_dst, bd_stream_t stream) {
    BD_CHECK(src && dst && B > 0 && H > 0 && W > 0 && C > 0, BD_ERR_INVALID, "bd_split_rows_ups2: bad args");
    BD_CHECK(C % 32 == 0 && ld_dst % 32 == 0 && (ld_src & 3) == 0 && bd::aligned16(src) && ((uintptr_t)dst & 127) == 0, BD_ERR_UNSUPPORTED,
             "bd_split_rows_ups2: C, ld_dst %% 32, ld_src %% 4, 16-byte aligned src and 128-byte aligned dst required");
    long long nb = bd::cdiv((long long)B * 4 * H * W * (C / 8), 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bd::split_rows_ups2_kernel, dim3((unsigned)nb), dim3(256), 0, bd::S(stream), src, (long long)ld_src, B, H, W, C, dst,
                       (long long)ld_dst);
    BD_LAUNCH_CHECK("split_rows_ups2");
    return BD_OK;
}
extern "C" int bd_split_wt(const float* w, int Cin, int Cout, uint16_t* out, bd_stream_t stream) {
    BD_CHECK(w && out && Cin > 0 && Cout > 0, BD_ERR_INVALID, "bd_split_wt: bad args");
    BD_CHECK(Cin % 32 == 0 && Cout % 32 == 0 && ((uintptr_t)out & 127) == 0, BD_ERR_UNSUPPORTED, "bd_split_wt: Cin, Cout %% 32 and 128-byte aligned out required");
    hipLaunchKernelGGL(bd::split_wT_kernel, dim3(Cin / 32, Cout / 32, 9), dim3(256), 0, bd::S(stream), w, Cin, Cout, out);
    BD_LAUNCH_CHECK("split_wT");
    return BD_OK;
}
