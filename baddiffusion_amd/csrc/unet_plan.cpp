// UNet2DModel as a static execution plan (host-only object; all device memory is the caller's).
//
// Mirrors the topology built by /root/reference/diffusers/src/diffusers/models/unet_2d.py:82-217 and the
// forward of :229-326, with ResnetBlock2D (resnet.py:551-601), AttentionBlock (attention.py:121-174),
// Downsample2D (resnet.py:199-208), Upsample2D (resnet.py:126-161) and the skip bookkeeping of
// unet_2d_blocks.py (DownBlock2D, AttnDownBlock2D, UNetMidBlock2D, UpBlock2D, AttnUpBlock2D).
//
// MI355X-first choices:
//  * activations NHWC with a leading dimension: every skip tensor is BORN inside the concat buffer of the
//    up-block resnet that will consume it, so torch.cat (13 copies/forward in the reference) disappears;
//  * parameters / gradients are ONE flat fp32 buffer each (time-embedding projections of all resnets are
//    contiguous => one GEMM; Adam + clip are single launches; DP all-reduce buckets are plain ranges);
//  * every parameter receives exactly one gradient contribution => grads are written, never accumulated;
//  * forward and backward are fixed launch sequences on one stream: hipGraph-capturable, no host sync.
#include "common.h"

#include <functional>
#include <string>
#include <vector>
#include <cmath>
#include <cstdlib>

namespace bd {

int conv3x3_fwd(const bd_conv3x3_fwd_desc& d, hipStream_t st);
int conv3x3_dgrad(const bd_conv3x3_dgrad_desc& d, hipStream_t st);
int conv3x3_wgrad(const bd_conv3x3_wgrad_desc& d, hipStream_t st);
bool conv3x3_wgrad_is_thin(const bd_conv3x3_wgrad_desc& d);
int conv3x3_ps(const bd_conv3x3_ps_desc& d, hipStream_t st);                         // conv_ps.hip
bool conv3x3_ps_supported(int B, int H, int W, int K_channels, int N_channels);
int conv3x3_ps_wgrad(const bd_conv3x3_ps_wgrad_desc& d, hipStream_t st);
size_t conv3x3_ps_wgrad_workspace_bytes(const bd_conv3x3_ps_wgrad_desc& d);
bool conv3x3_ps_wgrad_supported(int B, int H, int W, int Cin, int Cout);
int upsample_weights(const float* w, int Cin, int Cout, uint16_t* e_split, uint16_t* et_split, hipStream_t st);   // conv_ph.hip
int upsample_conv_fwd(const bd_upsample_conv_desc& d, hipStream_t st);
int upsample_conv_dgrad(const bd_upsample_conv_desc& d, hipStream_t st);
int upsample_conv_wgrad(const bd_upsample_conv_desc& d, hipStream_t st);             // conv_ps.hip (PHASE form of the weight gradient)
size_t upsample_conv_wgrad_workspace_bytes(const bd_upsample_conv_desc& d);
size_t upsample_conv_dgrad_workspace_bytes(const bd_upsample_conv_desc& d);
bool upsample_conv_ps_supported(int B, int H, int W, int Cin, int Cout);
int conv3x3_s2_dgrad_ps(const bd_conv3x3_s2_dgrad_desc& d, hipStream_t st);
int gemm_sp(const bd_gemm_sp_desc& d, hipStream_t st);                              // gemm_sp.hip
size_t gemm_sp_workspace_bytes(const bd_gemm_sp_desc& d);
bool gemm_sp_supported(int M, int N, int K);
int attn_sp_fwd(const bd_attn_sp_desc& d, hipStream_t st);                          // attn_sp.hip
int attn_sp_bwd(const bd_attn_sp_desc& d, hipStream_t st);
bool attn_sp_supported(int N, int dh);
int split_wt_batched(const float* params, uint16_t* out, const long long* off, const int* cin, const int* cout, int n, hipStream_t st);

struct View {
    int buf = -1;   // value buffer id
    int coff = 0;   // channel offset inside the buffer
    int C = 0, ld = 0, H = 0, W = 0;
};

enum Region { R_VALUE = 0, R_GRAD = 1, R_SCRATCH = 2 };
struct Buf {
    int64_t per_sample = 0, fixed = 0;
    int region = R_VALUE, group = 0;
    int64_t off = 0;  // floats, filled by layout()
    int gbuf = -1;    // grad buffer of a value buffer
    // optional: decided per (batch, training) at layout time -- a buffer only one of several kernel paths uses takes no room
    // when the plan selects another path for that batch (b_xuS vs b_xS / E / E^T of the upsample convolutions)
    std::function<bool(int, int)> live;
};

struct Param {
    std::string name;
    int64_t off;
    int rank;
    int64_t shape[4];
    int layout;
};

struct Ctx {
    bool dry = false;
    bool training = true;   // forward only: keep what backward needs
    int B = 0;
    int64_t s0 = 0;   // sample offset of this context inside the batch the workspace is laid out for
    int LB = 0;       // that batch (B of the call); B <= LB is the number of samples this context processes
    float* ws = nullptr;
    const float* params = nullptr;
    float* grads = nullptr;
    hipStream_t st = nullptr;
    const float* x = nullptr; int64_t ldx = 0;
    const int64_t* t = nullptr; int t_stride = 0;
    float* out = nullptr; int64_t ldo = 0;
    const float* dout = nullptr; int64_t lddo = 0;
    char* opws = nullptr; size_t opws_bytes = 0; size_t opws_need = 0;
    // weight-gradient side stream (backward only): wgrad GEMMs feed nothing but the optimizer, so they run on a second
    // stream next to the dgrad / GroupNorm chain that the rest of backward waits for.  Forked per launch; the scratch
    // arena alternates between nodes (group parity), so node k's wgrads may still run during node k+1 and are joined
    // before node k+2 reuses their arena (and at the end of every backward call).  Null = everything on `st`.
    hipStream_t st2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    char* opws2 = nullptr; int par = 0; bool launched = false, pend[2] = {false, false};
    const uint16_t* w_split = nullptr;   // pre-split weights (BF16X3, bd_split_bf16 layout): element e of params <-> 2*e here
    const uint16_t* wT_split = nullptr;  // transposed split planes Wt[ci][tap][co] of the 3x3 conv weights, same element offsets
    std::vector<char> ginit;
    // GroupNorm weight / bias gradients: the layers leave their per-sample partials in `gnpart` and ONE launch per
    // backward call folds them (bd_gn_bwd_params) -- 51 small launches off the dgrad chain of the CIFAR step
    float* gnpart = nullptr; size_t gnpart_floats = 0, gnpart_used = 0, gnpart_need = 0;
    std::vector<bd_gn_param_item> gn_items;
    // hands a GroupNorm backward its slot (or nothing: the layer then reduces its own parameters as before)
    float* gn_slot(int B, int HW, int C, int G, float* dgamma, float* dbeta) {
        if (!bd_gn_bwd_defers(B, HW, C, G)) return nullptr;
        const size_t n = ((size_t)B * 2 * C + 63) / 64 * 64;
        if (dry) { gnpart_need += n; return nullptr; }
        if (!gnpart || gnpart_used + n > gnpart_floats) return nullptr;
        float* p = gnpart + gnpart_used;
        gnpart_used += n;
        bd_gn_param_item it; it.partials = p; it.C = C; it.dgamma = dgamma; it.dbeta = dbeta;
        gn_items.push_back(it);
        return p;
    }
};

typedef std::function<int(Ctx&)> Step;

}  // namespace bd

using namespace bd;

struct bd_unet {
    bd_unet_config cfg;
    hipStream_t aux_stream = nullptr;              // created on the first backward (plan creation stays host-only)
    hipEvent_t aux_ev_fork = nullptr, aux_ev_join[2] = {nullptr, nullptr};
    hipEvent_t aux_ev_seg = nullptr;               // side-stream position after the last segment call's weight gradients
    bool aux_pend[2] = {false, false};             // carried from one bd_unet_backward_segment call to the next (same backward pass)
    int aux_enabled = 1, aux_defer = 0;
    std::vector<Buf> bufs;
    std::vector<Param> params;
    int64_t nparams = 0;
    std::vector<Step> fwd;
    struct BStep { Step fn; int seg; int group; };
    std::vector<BStep> bwd;  // in FORWARD emission order; executed reversed
    // gradient ranges that are final when a backward segment has run: the segment's own parameters [lo, hi) plus, for the
    // blocks that hold resnets, their rows of the batched time_emb_proj weight / bias (which live in the time-embedding
    // segment's range of the flat buffer but get their gradient inside the resnet's backward)
    struct Seg { int64_t lo, hi; int64_t tw_lo = -1, tw_hi = -1, tb_lo = -1, tb_hi = -1; };
    std::vector<Seg> segs;   // indexed by segment id (backward order)
    int cur_seg = 0, cur_group = 0;
    // layout cache
    int lay_B = -1, lay_train = -1, lay_gen = -1;
    int fwd_gen = -1, fwd_B = -1; const void* fwd_ws = nullptr;   // the training forward whose saved activations are live (bd_tune_set guard)
    int64_t value_floats = 0, grad_floats = 0, scratch_floats = 0;
    size_t opws_bytes = 0, gnpart_floats = 0;
    int T = 0, sumC = 0;     // time-embed dim, total time_emb_proj rows
    int64_t p_tw = 0, p_tb = 0;  // offsets of the batched time_emb_proj weight / bias
    int b_tproj = -1, b_dtproj = -1, b_embs = -1;
    std::vector<long long> wt_off; std::vector<int> wt_cin, wt_cout;   // 3x3 conv weights with a transposed split copy
    // Round 4: a tensor's gradient handed over READY-SPLIT.  The producer of tensor T (a resnet / attention / down- / upsample node whose
    // backward starts by turning dL/dT into split planes) registers T; the FIRST node that consumes T in forward order is the LAST one to add
    // to dL/dT in backward, and if that node ends with a GroupNorm backward (resnet norm1, attention group_norm, conv_norm_out) the kernel
    // writes the planes of the final value next to the fp32 store (bd_gn_bwd_desc.dx_split + channel range): the producer's bd_split_rows
    // launch and its 4 B / element read disappear.  Consumers that end otherwise (down- / upsample data gradients) leave the fallback.
    struct GSplit { int buf, coff, C, H, W; int b_pl; std::function<bool(const Ctx&)> want; bool claimed = false, emitted = false; };
    std::vector<GSplit> gsplits;
    int reg_gsplit(const View& y, std::function<bool(const Ctx&)> want) {
        static const bool off = getenv("BD_GSPLIT") && atoi(getenv("BD_GSPLIT")) == 0;      // (A/B knob)
        if (off || y.buf < 0 || y.C % 32 != 0) return -1;
        GSplit e; e.buf = y.buf; e.coff = y.coff; e.C = y.C; e.H = y.H; e.W = y.W; e.want = std::move(want);
        const int keep = cur_group;
        e.b_pl = new_buf((int64_t)y.H * y.W * y.C, 0, R_GRAD);
        cur_group = keep;
        gsplits.push_back(std::move(e));
        const int idx = (int)gsplits.size() - 1;
        bufs[gsplits[idx].b_pl].live = [this, idx](int B, int training) {
            Ctx q; q.dry = true; q.B = q.LB = B;
            return training && gsplits[idx].emitted && gsplits[idx].want(q);
        };
        return idx;
    }
    // called by every node for its input view, in forward order: the first consumer of a registered tensor; `emits`: it can write the planes
    int claim_gsplit(const View& x, bool emits) {
        int got = -1;
        for (int i = 0; i < (int)gsplits.size(); ++i) {
            GSplit& e = gsplits[i];
            if (e.claimed || e.buf != x.buf || e.coff < x.coff || e.coff + e.C > x.coff + x.C) continue;
            e.claimed = true;
            if (emits && got < 0) { e.emitted = true; got = i; }      // (one range per consumer launch)
        }
        return got;
    }
    struct UpsW { int64_t pw; int C, H, W; int b_e, b_et; };           // upsample convolutions on the phase path: E (/ E^T when training) planes
    // weight preprocessing (split copy, E planes) is skipped while the caller promises constant weights (bd_unet_set_static_weights):
    // a sampling loop runs 50-1000 forwards over the same parameters
    int static_weights = 0; const void* prep_params = nullptr; const void* prep_ws = nullptr; int prep_B = -1;
    std::vector<UpsW> upsw;

    // ---------------------------------------------------------------- construction helpers
    int new_buf(int64_t per_sample, int64_t fixed, int region) {
        Buf b;
        b.per_sample = per_sample; b.fixed = fixed; b.region = region; b.group = cur_group;
        bufs.push_back(b);
        return (int)bufs.size() - 1;
    }
    View new_tensor(int H, int W, int C) {  // standalone value tensor with a grad buffer
        View v;
        v.buf = new_buf((int64_t)H * W * C, 0, R_VALUE);
        bufs[v.buf].gbuf = new_buf((int64_t)H * W * C, 0, R_GRAD);
        v.coff = 0; v.C = C; v.ld = C; v.H = H; v.W = W;
        return v;
    }
    View slice(const View& v, int coff, int C) const {
        View s = v;
        s.coff = v.coff + coff; s.C = C;
        return s;
    }
    int scratch(int64_t per_sample, int64_t fixed = 0) { return new_buf(per_sample, fixed, R_SCRATCH); }
    int64_t add_param(const std::string& name, std::initializer_list<int64_t> shape, int layout = 0) {
        Param p;
        p.name = name; p.rank = (int)shape.size(); p.layout = layout;
        int64_t n = 1; int i = 0;
        for (auto s : shape) { p.shape[i++] = s; n *= s; }
        for (; i < 4; ++i) p.shape[i] = 1;
        nparams = (nparams + 31) / 32 * 32;  // 128-byte aligned start: float4 loads, and whole 32-element blocks of the split copy
        p.off = nparams;
        nparams += n;
        params.push_back(p);
        if (!segs.empty()) {
            Seg& s = segs[cur_seg];
            if (s.lo < 0 || p.off < s.lo) s.lo = p.off;
            if (p.off + n > s.hi) s.hi = p.off + n;
        }
        return p.off;
    }
    void alias_param(const std::string& name, int64_t off, std::initializer_list<int64_t> shape) {
        Param p;
        p.name = name; p.rank = (int)shape.size(); p.layout = 0; p.off = off;
        int i = 0;
        for (auto s : shape) p.shape[i++] = s;
        for (; i < 4; ++i) p.shape[i] = 1;
        params.push_back(p);
    }
    void F(Step s) { fwd.push_back(std::move(s)); }
    void Bk(Step s) { bwd.push_back({std::move(s), cur_seg, cur_group}); }

    // ---------------------------------------------------------------- run-time helpers
    // c.s0 = first sample of this context (forward runs the two halves of the batch as two concurrent pipelines)
    float* VP(Ctx& c, const View& v) const { return c.ws + bufs[v.buf].off + c.s0 * bufs[v.buf].per_sample + v.coff; }
    float* GP(Ctx& c, const View& v) const {
        const Buf& g = bufs[bufs[v.buf].gbuf];
        return c.ws + g.off + c.s0 * g.per_sample + v.coff;
    }
    float* BP(Ctx& c, int b) const { return c.ws + bufs[b].off + c.s0 * bufs[b].per_sample; }
    // GroupNorm statistics buffer = [LB][G] means followed by [LB][G] rstds (batch-major, so not a per-sample slab)
    float* MEANP(Ctx& c, int b, int G) const { return c.ws + bufs[b].off + c.s0 * G; }
    float* RSTDP(Ctx& c, int b, int G) const { return c.ws + bufs[b].off + ((int64_t)c.LB + c.s0) * G; }
    static int64_t rows(const Ctx& c, const View& v) { return (int64_t)c.B * v.H * v.W; }

    // enqueue a weight-gradient launch on the side stream (after everything enqueued on c.st so far)
    template <class F>
    int on_aux(Ctx& c, F&& launch) const {
        if (!c.st2 || c.dry) return launch(c.st, c.opws);
        BD_HIP_TRY(hipEventRecord(c.ev_fork, c.st));
        BD_HIP_TRY(hipStreamWaitEvent(c.st2, c.ev_fork, 0));
        c.launched = true;
        return launch(c.st2, c.opws2);
    }
    // before a node of scratch parity p starts: wgrads of the previous node with that parity must be done
    static int aux_wait(Ctx& c, int p) {
        if (!c.pend[p]) return BD_OK;
        BD_HIP_TRY(hipStreamWaitEvent(c.st, c.ev_join[p], 0));
        c.pend[p] = false;
        return BD_OK;
    }
    // after a node: mark the completion point of the wgrads it enqueued
    static int aux_mark(Ctx& c, int p) {
        if (!c.launched) return BD_OK;
        BD_HIP_TRY(hipEventRecord(c.ev_join[p], c.st2));
        c.pend[p] = true; c.launched = false;
        return BD_OK;
    }
    int igemm(Ctx& c, bd_igemm_desc& g) const {
        g.workspace = c.opws; g.workspace_bytes = c.opws_bytes; g.mode = cfg.compute_mode;
        if (c.dry) {
            size_t n = igemm_workspace_bytes(g);
            if (n > c.opws_need) c.opws_need = n;
            return BD_OK;
        }
        return igemm_launch(g, c.st);
    }
    // GEMM on split planes (gemm_sp.hip); aux: a weight gradient, on the side stream
    int gemm_s(Ctx& c, bd_gemm_sp_desc& g, bool aux = false) const {
        g.workspace_bytes = c.opws_bytes;
        if (c.dry) {
            const size_t n = gemm_sp_workspace_bytes(g);
            if (n > c.opws_need) c.opws_need = n;
            return BD_OK;
        }
        static const bool aux_on = !(getenv("BD_SP_WG_AUX") && atoi(getenv("BD_SP_WG_AUX")) == 0);   // (A/B knob: weight gradients on the main stream)
        if (aux && aux_on) return on_aux(c, [&](hipStream_t st, char* ws) { g.workspace = ws; return gemm_sp(g, st); });
        g.workspace = c.opws;
        return gemm_sp(g, c.st);
    }
    static bd_operand dense(const float* p, int64_t ld, int kc) {
        bd_operand o = {};
        o.kind = BD_OPK_DENSE; o.kc = kc; o.p = p; o.ld = ld;
        return o;
    }
    // C[M,N] = alpha * A[M,K] * W[N,K]^T (+bias) (+residual)*out_scale        (Linear / 1x1 conv forward)
    int linear_fwd(Ctx& c, const float* A, int64_t lda, const float* W, const float* bias, float* C, int64_t ldc, int M, int N,
                   int K, const float* residual = nullptr, int64_t ldr = 0, float out_scale = 1.f) const {
        bd_igemm_desc g = {};
        g.A = dense(A, lda, 1); g.B = dense(W, K, 1);
        g.M = M; g.N = N; g.K = K; g.batch_outer = g.batch_inner = 1;
        g.C = C; g.ldc = ldc; g.alpha = 1.f; g.out_scale = out_scale; g.bias = bias; g.residual = residual; g.ldr = ldr;
        return igemm(c, g);
    }
    // dX[M,K] (+)= dY[M,N] * W[N,K]
    int linear_dgrad(Ctx& c, const float* dY, int64_t lddy, const float* W, float* dX, int64_t lddx, int M, int N, int K,
                     int acc) const {
        bd_igemm_desc g = {};
        g.A = dense(dY, lddy, 1); g.B = dense(W, K, 0);
        g.M = M; g.N = K; g.K = N; g.batch_outer = g.batch_inner = 1;
        g.C = dX; g.ldc = lddx; g.alpha = 1.f; g.out_scale = 1.f; g.accumulate = acc;
        return igemm(c, g);
    }
    // dW[N,K] = dY[M,N]^T * X[M,K]; db[N] = column sums of dY, fused into the same launch (N % 4 == 0)
    int linear_wgrad(Ctx& c, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int M, int N, int K,
                     float* db = nullptr) const {
        bd_igemm_desc g = {};
        g.A = dense(dY, lddy, 0); g.B = dense(X, ldx, 0);
        g.M = N; g.N = K; g.K = M; g.batch_outer = g.batch_inner = 1;
        g.C = dW; g.ldc = K; g.alpha = 1.f; g.out_scale = 1.f; g.a_colsum = db;
        if (c.dry) return igemm(c, g);
        g.workspace_bytes = c.opws_bytes; g.mode = cfg.compute_mode;
        return on_aux(c, [&](hipStream_t st, char* ws) { g.workspace = ws; return igemm_launch(g, st); });
    }
    int colsum(Ctx& c, const float* x, int64_t ldx, int64_t nrows, int N, int64_t rpg, float* out, int64_t ldo) const {
        if (c.dry) return BD_OK;
        return bd_colsum(x, ldx, nrows, N, rpg, out, ldo, 0, c.st);
    }
    // bias gradient: per-sample column sums (scratch [B,N]) then a sum over the batch (fixed order)
    int bias_grad(Ctx& c, const float* dY, int64_t ld, int HW, int N, int scratch_buf, float* db, float* db2 = nullptr) const {
        float* s = BP(c, scratch_buf);
        BD_TRY(colsum(c, dY, ld, (int64_t)c.B * HW, N, HW, s, N));
        BD_TRY(colsum(c, s, N, c.B, N, c.B, db, N));
        if (db2) BD_TRY(colsum(c, s, N, c.B, N, c.B, db2, N));
        return BD_OK;
    }
    int gn_fwd(Ctx& c, const View& x, int64_t pg, int64_t pb, float* y, int64_t ldy, int stats_buf, int silu) const {
        bd_gn_fwd_desc d = {};
        d.B = c.B; d.HW = x.H * x.W; d.C = x.C; d.G = cfg.norm_num_groups; d.eps = cfg.norm_eps; d.silu = silu;
        d.x = VP(c, x); d.ldx = x.ld; d.gamma = c.params + pg; d.beta = c.params + pb; d.y = y; d.ldy = ldy;
        d.mean = MEANP(c, stats_buf, d.G); d.rstd = RSTDP(c, stats_buf, d.G);
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes;
        if (c.dry) {
            size_t n = bd_gn_workspace_bytes(c.B, x.C);
            if (n > c.opws_need) c.opws_need = n;
            return BD_OK;
        }
        return bd_gn_fwd(&d, (bd_stream_t)c.st);
    }
    int gn_bwd(Ctx& c, const View& x, int64_t pg, int64_t pb, int stats_buf, const float* dy, int64_t lddy, int silu,
               const float* dx_add = nullptr, int64_t ld_add = 0, int gs = -1) {
        bd_gn_bwd_desc d = {};
        d.dx_add = dx_add; d.ld_add = ld_add;
        if (gs >= 0 && gsplits[gs].want(c)) {      // this launch makes dL/dx final: hand its producer the planes of its channel range
            const GSplit& e = gsplits[gs];
            d.dx_split = U16(BP(c, e.b_pl)); d.lddxs = e.C; d.dx_split_c0 = e.coff - x.coff; d.dx_split_c1 = d.dx_split_c0 + e.C;
        }
        d.B = c.B; d.HW = x.H * x.W; d.C = x.C; d.G = cfg.norm_num_groups; d.silu = silu;
        d.x = VP(c, x); d.ldx = x.ld; d.gamma = c.params + pg; d.beta = c.params + pb;
        d.mean = MEANP(c, stats_buf, d.G); d.rstd = RSTDP(c, stats_buf, d.G);
        d.dy = dy; d.lddy = lddy; d.dx = GP(c, x); d.lddx = x.ld;
        d.accumulate_dx = c.ginit[x.buf];
        c.ginit[x.buf] = 1;
        d.dgamma = c.grads + pg; d.dbeta = c.grads + pb;
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes;
        d.param_partials = c.gn_slot(d.B, d.HW, d.C, d.G, d.dgamma, d.dbeta);
        if (c.dry) return BD_OK;
        return bd_gn_bwd(&d, (bd_stream_t)c.st);
    }
    int add(Ctx& c, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t nrows, int C, float scale, int acc) const {
        if (c.dry) return BD_OK;
        return add_launch(src, lds, dst, ldd, nrows, C, scale, acc, c.st);
    }
    int conv_f(Ctx& c, bd_conv3x3_fwd_desc& d) const {
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes; d.mode = cfg.compute_mode;
        if (c.w_split && !c.dry) d.w_split = c.w_split + 2 * (d.w - c.params);
        if (c.dry) { note_conv(c); return BD_OK; }
        return conv3x3_fwd(d, c.st);
    }
    int conv_d(Ctx& c, bd_conv3x3_dgrad_desc& d) const {
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes; d.mode = cfg.compute_mode;
        if (c.w_split && !c.dry) d.w_split = c.w_split + 2 * (d.w - c.params);
        if (c.dry) { note_conv(c); return BD_OK; }
        return conv3x3_dgrad(d, c.st);
    }
    int conv_w(Ctx& c, bd_conv3x3_wgrad_desc& d) const {
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes; d.mode = cfg.compute_mode;
        if (c.dry) { note_conv(c); return BD_OK; }
        return on_aux(c, [&](hipStream_t st, char* ws) { d.workspace = ws; return conv3x3_wgrad(d, st); });
    }
    // pre-split / LDS-DMA convolution (conv_ps.hip) for the stride-1 3x3 convs whose operands the producers can emit as
    // split planes: BF16X3 mode only (same arithmetic, bit-identical), shapes per conv3x3_ps_supported
    // One decision per convolution, for its forward, data gradient and weight gradient alike (they share the split
    // operands), taken on the batch the workspace is laid out for (c.LB: the forward's half-batch pipelines and the
    // backward must agree).
    bool ps_ok(const Ctx& c, int H, int W, int Cin, int Cout) const {
        static const bool off = getenv("BD_CONV_PS") && atoi(getenv("BD_CONV_PS")) == 0;
        const int B = c.LB > 0 ? c.LB : c.B;
        return !off && cfg.compute_mode == BD_MODE_BF16X3 && (c.dry || c.w_split) && conv3x3_ps_supported(B, H, W, Cin, Cout) &&
               conv3x3_ps_supported(B, H, W, Cout, Cin) && conv3x3_ps_wgrad_supported(B, H, W, Cin, Cout);
    }
    // phase-decomposed forms (conv_ph.hip): the upsample convolution on its SOURCE grid, the stride-2 data gradient by parity class
    // `min_wgs`: the launch must offer at least that many 256 x 128 workgroups (classes x tiles on the batch the workspace is laid
    // out for) -- a one-class 16-tap data gradient of an 8 x 8 source grid is 64 long workgroups on 256 CUs and loses to the literal form
    bool phase_ok(const Ctx& c, int H, int W, int Cin, int Cout, int classes = 4, int min_wgs = 128) const {
        const int B = c.LB > 0 ? c.LB : c.B;
        const long long wgs = (long long)classes * (((long long)B * H * W + 255) / 256) * (Cout / 128);
        return cfg.compute_mode == BD_MODE_BF16X3 && (c.dry || c.w_split) && upsample_conv_ps_supported(B, H, W, Cin, Cout) && wgs >= min_wgs;
    }
    int conv_pw(Ctx& c, bd_conv3x3_ps_wgrad_desc& d) const {
        d.workspace_bytes = c.opws_bytes;
        if (c.dry) {
            note_conv(c);
            const size_t n = conv3x3_ps_wgrad_workspace_bytes(d);      // K-split slabs of this layer (grow with the tile of the shared-tap form)
            if (n > c.opws_need) c.opws_need = n;
            return BD_OK;
        }
        // BD_AUX_MAXPIX (A/B knob): weight gradients over more pixels than this stay on the main stream -- they fill the chip
        // on their own, and beside the dgrad chain they mostly trade clock for overlap
        static const long long maxpix = getenv("BD_AUX_MAXPIX") ? atoll(getenv("BD_AUX_MAXPIX")) : (1ll << 62);
        // BD_AUX_MINPIX (A/B knob): weight gradients over fewer pixels than this stay on the main stream too -- at 4 x 4 / 8 x 8 a launch is
        // 10-30 us and the fork / join events cost as much as the overlap returns
        static const long long minpix = getenv("BD_AUX_MINPIX") ? atoll(getenv("BD_AUX_MINPIX")) : 0;
        if ((long long)d.B * d.H * d.W > maxpix || (long long)d.B * d.H * d.W < minpix) { d.workspace = c.opws; return conv3x3_ps_wgrad(d, c.st); }
        return on_aux(c, [&](hipStream_t st, char* ws) { d.workspace = ws; return conv3x3_ps_wgrad(d, st); });
    }
    static uint16_t* U16(float* p) { return reinterpret_cast<uint16_t*>(p); }
    int conv_p(Ctx& c, bd_conv3x3_ps_desc& d) const {
        d.workspace = c.opws; d.workspace_bytes = c.opws_bytes;
        if (c.dry) { note_conv(c); return BD_OK; }
        return conv3x3_ps(d, c.st);
    }
    int split_rows(Ctx& c, const float* src, int64_t ld, int64_t nrows, int C, float* dst) const {
        if (c.dry) return BD_OK;
        return bd_split_rows(src, ld, nrows, C, U16(dst), C, (bd_stream_t)c.st);
    }
    static void note_conv(Ctx& c) {
        size_t n = bd_conv3x3_workspace_bytes(0, 0, 0, 0, 0, 0, 0, 0);
        if (n > c.opws_need) c.opws_need = n;
    }

    // ---------------------------------------------------------------- nodes
    struct ConvSpec { int Cin, Cout, stride, pad_t, pad_l, ups; };

    void build();
    void node_time_embed();
    void node_conv_in(const View& y);
    void node_resnet(const std::string& pre, const View& x, const View& y, int toff, float scale);
    void node_attention(const std::string& pre, const View& x, const View& y, float scale);
    void node_downsample(const std::string& pre, const View& x, const View& y);
    void node_upsample(const std::string& pre, const View& x, const View& y);
    void node_conv_out(const View& x);
    void layout(int B, int training);
};

// ----------------------------------------------------------------------------------------------------
void bd_unet::node_time_embed() {
    const int c0 = cfg.block_out_channels[0];
    const int64_t pw1 = add_param("time_embedding.linear_1.weight", {T, c0});
    const int64_t pb1 = add_param("time_embedding.linear_1.bias", {T});
    const int64_t pw2 = add_param("time_embedding.linear_2.weight", {T, T});
    const int64_t pb2 = add_param("time_embedding.linear_2.bias", {T});
    // batched time_emb_proj of every resnet: one [sumC, T] weight, one [sumC] bias (aliases registered per resnet)
    segs[cur_seg].hi = nparams;          // the time segment proper: linear_1 / linear_2 (the batched projection below is
    nparams = (nparams + 31) / 32 * 32;   // reduced row range by row range with the blocks that own the resnets)
    p_tw = nparams; nparams += (int64_t)sumC * T;
    p_tb = nparams; nparams += sumC;
    const int b_tsin = new_buf(c0, 0, R_VALUE), b_e1 = new_buf(T, 0, R_VALUE), b_e1s = new_buf(T, 0, R_VALUE);
    const int b_emb = new_buf(T, 0, R_VALUE);
    b_embs = new_buf(T, 0, R_VALUE);
    b_tproj = new_buf(sumC, 0, R_VALUE);
    b_dtproj = new_buf(sumC, 0, R_GRAD);
    const int b_dembs = scratch(T), b_demb = scratch(T), b_de1s = scratch(T), b_de1 = scratch(T);
    F([=](Ctx& c) {
        if (!c.dry)
            BD_TRY(bd_timestep_embedding(c.t, c.t_stride, c.B, c0, cfg.flip_sin_to_cos, cfg.freq_shift, BP(c, b_tsin), (bd_stream_t)c.st));
        BD_TRY(linear_fwd(c, BP(c, b_tsin), c0, c.params + pw1, c.params + pb1, BP(c, b_e1), T, c.B, T, c0));
        if (!c.dry) BD_TRY(bd_silu_fwd(BP(c, b_e1), BP(c, b_e1s), (int64_t)c.B * T, (bd_stream_t)c.st));
        BD_TRY(linear_fwd(c, BP(c, b_e1s), T, c.params + pw2, c.params + pb2, BP(c, b_emb), T, c.B, T, T));
        if (!c.dry) BD_TRY(bd_silu_fwd(BP(c, b_emb), BP(c, b_embs), (int64_t)c.B * T, (bd_stream_t)c.st));
        BD_TRY(linear_fwd(c, BP(c, b_embs), T, c.params + p_tw, c.params + p_tb, BP(c, b_tproj), sumC, c.B, sumC, T));
        return (int)BD_OK;
    });
    Bk([=](Ctx& c) {
        float* dtp = BP(c, b_dtproj);
        // (the weight / bias gradient of the batched projection is produced resnet by resnet, see node_resnet)
        BD_TRY(linear_dgrad(c, dtp, sumC, c.params + p_tw, BP(c, b_dembs), T, c.B, sumC, T, 0));
        if (!c.dry) BD_TRY(bd_silu_bwd(BP(c, b_emb), BP(c, b_dembs), BP(c, b_demb), (int64_t)c.B * T, 0, (bd_stream_t)c.st));
        BD_TRY(linear_wgrad(c, BP(c, b_demb), T, BP(c, b_e1s), T, c.grads + pw2, c.B, T, T, c.grads + pb2));
        BD_TRY(linear_dgrad(c, BP(c, b_demb), T, c.params + pw2, BP(c, b_de1s), T, c.B, T, T, 0));
        if (!c.dry) BD_TRY(bd_silu_bwd(BP(c, b_e1), BP(c, b_de1s), BP(c, b_de1), (int64_t)c.B * T, 0, (bd_stream_t)c.st));
        BD_TRY(linear_wgrad(c, BP(c, b_de1), T, BP(c, b_tsin), c0, c.grads + pw1, c.B, T, c0, c.grads + pb1));
        return (int)BD_OK;
    });
}

void bd_unet::node_conv_in(const View& y) {
    const int Cin = cfg.in_channels, Cout = y.C, S_ = cfg.sample_size;
    const int64_t pw = add_param("conv_in.weight", {Cout, Cin, 3, 3}, 1);
    const int64_t pb = add_param("conv_in.bias", {Cout});
    const int b_bs = scratch(Cout);
    F([=](Ctx& c) {
        bd_conv3x3_fwd_desc d = {};
        d.B = c.B; d.Hs = S_; d.Ws = S_; d.Cin = Cin; d.Cout = Cout; d.stride = 1; d.pad_t = 1; d.pad_l = 1; d.Ho = S_; d.Wo = S_;
        d.x = c.x; d.ldx = c.ldx; d.w = c.params + pw; d.bias = c.params + pb; d.out_scale = 1.f;
        d.y = VP(c, y); d.ldy = y.ld;
        return conv_f(c, d);
    });
    Bk([=](Ctx& c) {
        const float* dy = GP(c, y);
        bd_conv3x3_wgrad_desc d = {};
        d.B = c.B; d.Hs = S_; d.Ws = S_; d.Cin = Cin; d.Cout = Cout; d.stride = 1; d.pad_t = 1; d.pad_l = 1; d.Ho = S_; d.Wo = S_;
        d.x = c.x; d.ldx = c.ldx; d.dy = dy; d.lddy = y.ld; d.dw = c.grads + pw; d.db = c.grads + pb;
        return conv_w(c, d);
    });
}

void bd_unet::node_resnet(const std::string& pre, const View& x, const View& y, int toff, float scale) {
    const int Cin = x.C, Cout = y.C, H = x.H, W = x.W, HW = H * W;
    const bool shortcut = Cin != Cout;
    const int64_t pn1w = add_param(pre + "norm1.weight", {Cin}), pn1b = add_param(pre + "norm1.bias", {Cin});
    const int64_t pc1w = add_param(pre + "conv1.weight", {Cout, Cin, 3, 3}, 1), pc1b = add_param(pre + "conv1.bias", {Cout});
    alias_param(pre + "time_emb_proj.weight", p_tw + (int64_t)toff * T, {Cout, T});
    alias_param(pre + "time_emb_proj.bias", p_tb + toff, {Cout});
    {   // this block's rows of the batched projection become final with this segment (resnets of a block are consecutive)
        Seg& sg = segs[cur_seg];
        const int64_t wlo = p_tw + (int64_t)toff * T, whi = wlo + (int64_t)Cout * T, blo = p_tb + toff, bhi = blo + Cout;
        if (sg.tw_lo < 0) { sg.tw_lo = wlo; sg.tw_hi = whi; sg.tb_lo = blo; sg.tb_hi = bhi; }
        else {
            if (wlo < sg.tw_lo) sg.tw_lo = wlo;
            if (whi > sg.tw_hi) sg.tw_hi = whi;
            if (blo < sg.tb_lo) sg.tb_lo = blo;
            if (bhi > sg.tb_hi) sg.tb_hi = bhi;
        }
    }
    const int b_embs_ = b_embs, T_ = T;
    const int64_t p_tw_ = p_tw, p_tb_ = p_tb;
    const int64_t pn2w = add_param(pre + "norm2.weight", {Cout}), pn2b = add_param(pre + "norm2.bias", {Cout});
    const int64_t pc2w = add_param(pre + "conv2.weight", {Cout, Cout, 3, 3}, 1), pc2b = add_param(pre + "conv2.bias", {Cout});
    int64_t psw = -1, psb = -1;
    if (shortcut) {
        psw = add_param(pre + "conv_shortcut.weight", {Cout, Cin, 1, 1}, 1);
        psb = add_param(pre + "conv_shortcut.bias", {Cout});
    }
    if (Cin % 32 == 0 && Cout % 32 == 0) {
        wt_off.push_back(pc1w); wt_cin.push_back(Cin); wt_cout.push_back(Cout);
        wt_off.push_back(pc2w); wt_cin.push_back(Cout); wt_cout.push_back(Cout);
    }
    const int b_a1 = new_buf((int64_t)HW * Cin, 0, R_VALUE), b_h1 = new_buf((int64_t)HW * Cout, 0, R_VALUE);
    const int b_a2 = new_buf((int64_t)HW * Cout, 0, R_VALUE);
    // split-plane twins of a1 / a2 (operands of the LDS-DMA convolutions; 4 B per element like fp32)
    const int b_a1s = new_buf((int64_t)HW * Cin, 0, R_VALUE), b_a2s = new_buf((int64_t)HW * Cout, 0, R_VALUE);
    const int G = cfg.norm_num_groups;
    const int b_st1 = new_buf(2 * G, 0, R_VALUE), b_st2 = new_buf(2 * G, 0, R_VALUE);
    View h1v; h1v.buf = b_h1; h1v.coff = 0; h1v.C = Cout; h1v.ld = Cout; h1v.H = H; h1v.W = W;
    bufs[b_h1].gbuf = -1;
    const int b_dys = scale != 1.f ? scratch((int64_t)HW * Cout) : -1;
    const int b_bs = scratch(Cout), b_da2 = scratch((int64_t)HW * Cout), b_dh1 = scratch((int64_t)HW * Cout);
    const int b_da1 = scratch((int64_t)HW * Cin);
    // backward operands of the LDS-DMA data gradients: dy / dh1 as split planes, transposed weight planes
    const int b_dyS = scratch((int64_t)HW * Cout), b_dh1S = scratch((int64_t)HW * Cout);
    const float inv = 1.f / scale;
    const int sumC_ = sumC;
    // dL/dx becomes final in this node's norm1 backward (first consumer of x in forward order); dL/dy arrives ready-split when y's first
    // consumer can do the same for us (GSplit)
    const int gs_in = claim_gsplit(x, true);
    const int gs_out = b_dys < 0 ? reg_gsplit(y, [this, H, W, Cout](const Ctx& c) { return ps_ok(c, H, W, Cout, Cout); }) : -1;

    F([=](Ctx& c) {
        const bool ps1 = ps_ok(c, H, W, Cin, Cout), ps2 = ps_ok(c, H, W, Cout, Cout);
        int st2 = 0;          // pixel splits of the norm2 partials conv1's epilogue leaves in the op workspace (0: none)
        if (ps1) {
            bd_gn_fwd_desc g = {};
            g.B = c.B; g.HW = HW; g.C = Cin; g.G = G; g.eps = cfg.norm_eps; g.silu = 1;
            g.x = VP(c, x); g.ldx = x.ld; g.gamma = c.params + pn1w; g.beta = c.params + pn1b;
            g.y = nullptr; g.ldy = Cin;     // split planes only: conv1 fwd and (a1s is a value buffer) its weight gradient read them
            g.y_split = U16(BP(c, b_a1s)); g.ldys = Cin;
            g.mean = MEANP(c, b_st1, G); g.rstd = RSTDP(c, b_st1, G);
            g.workspace = c.opws; g.workspace_bytes = c.opws_bytes;
            if (!c.dry) BD_TRY(bd_gn_fwd(&g, (bd_stream_t)c.st));
            bd_conv3x3_ps_desc d = {};
            d.B = c.B; d.H = H; d.W = W; d.K = Cin; d.N = Cout; d.direction = 1;
            d.x_split = U16(BP(c, b_a1s)); d.ldx = Cin; d.w_split = c.w_split + 2 * pc1w;
            d.bias = c.params + pc1b; d.rowbias = BP(c, b_tproj) + toff; d.ld_rowbias = sumC_; d.out_scale = 1.f;
            d.y = BP(c, b_h1); d.ldy = Cout;
            // round 4: norm2's statistics from this epilogue when norm2 would otherwise make a pass over h1 for them (large images)
            st2 = ps2 && bd_gn_fwd_takes_stats(c.B, HW, Cout, G) ? bd_conv3x3_ps_gn_splits(c.B, H, W, Cin, Cout, G) : 0;
            if (st2 > 0) {
                d.gn_part = reinterpret_cast<double*>(c.opws); d.gn_groups = G;
                const size_t n = (size_t)c.B * st2 * G * 2 * sizeof(double);
                if (c.dry && n > c.opws_need) c.opws_need = n;
            }
            BD_TRY(conv_p(c, d));
        } else {
            BD_TRY(gn_fwd(c, x, pn1w, pn1b, BP(c, b_a1), Cin, b_st1, 1));
            bd_conv3x3_fwd_desc d = {};
            d.B = c.B; d.Hs = H; d.Ws = W; d.Cin = Cin; d.Cout = Cout; d.stride = 1; d.pad_t = 1; d.pad_l = 1; d.Ho = H; d.Wo = W;
            d.x = BP(c, b_a1); d.ldx = Cin; d.w = c.params + pc1w; d.bias = c.params + pc1b;
            d.rowbias = BP(c, b_tproj) + toff; d.ld_rowbias = sumC_; d.out_scale = 1.f;
            d.y = BP(c, b_h1); d.ldy = Cout;
            BD_TRY(conv_f(c, d));
        }
        if (ps2) {
            bd_gn_fwd_desc g = {};
            g.B = c.B; g.HW = HW; g.C = Cout; g.G = G; g.eps = cfg.norm_eps; g.silu = 1;
            g.x = BP(c, b_h1); g.ldx = Cout; g.gamma = c.params + pn2w; g.beta = c.params + pn2b;
            g.y = nullptr; g.ldy = Cout;
            g.y_split = U16(BP(c, b_a2s)); g.ldys = Cout;
            g.mean = MEANP(c, b_st2, G); g.rstd = RSTDP(c, b_st2, G);
            g.workspace = c.opws; g.workspace_bytes = c.opws_bytes;
            if (st2 > 0) { g.stats = reinterpret_cast<const double*>(c.opws); g.stats_splits = st2; }
            if (!c.dry) BD_TRY(bd_gn_fwd(&g, (bd_stream_t)c.st));
        } else {
            BD_TRY(gn_fwd(c, h1v, pn2w, pn2b, BP(c, b_a2), Cout, b_st2, 1));
        }
        const float* res = VP(c, x); int64_t ldr = x.ld;
        if (shortcut) {
            BD_TRY(linear_fwd(c, VP(c, x), x.ld, c.params + psw, c.params + psb, VP(c, y), y.ld, (int)rows(c, x), Cout, Cin));
            res = VP(c, y); ldr = y.ld;
        }
        if (ps2) {
            bd_conv3x3_ps_desc e = {};
            e.B = c.B; e.H = H; e.W = W; e.K = Cout; e.N = Cout; e.direction = 1;
            e.x_split = U16(BP(c, b_a2s)); e.ldx = Cout; e.w_split = c.w_split + 2 * pc2w;
            e.bias = c.params + pc2b; e.residual = res; e.ldr = ldr; e.out_scale = inv;
            e.y = VP(c, y); e.ldy = y.ld;
            return conv_p(c, e);
        }
        bd_conv3x3_fwd_desc e = {};
        e.B = c.B; e.Hs = H; e.Ws = W; e.Cin = Cout; e.Cout = Cout; e.stride = 1; e.pad_t = 1; e.pad_l = 1; e.Ho = H; e.Wo = W;
        e.x = BP(c, b_a2); e.ldx = Cout; e.w = c.params + pc2w; e.bias = c.params + pc2b;
        e.residual = res; e.ldr = ldr; e.out_scale = inv;
        e.y = VP(c, y); e.ldy = y.ld;
        return conv_f(c, e);
    });
    Bk([=](Ctx& c) {
        const int M = (int)rows(c, x);
        const float* dy = GP(c, y); int64_t lddy = y.ld;
        if (b_dys >= 0) {
            BD_TRY(add(c, dy, lddy, BP(c, b_dys), Cout, M, Cout, inv, 0));
            dy = BP(c, b_dys); lddy = Cout;
        }
        const bool ps1 = ps_ok(c, H, W, Cin, Cout), ps2 = ps_ok(c, H, W, Cout, Cout);
        if (ps2) {
            const bool ready = gs_out >= 0 && gsplits[gs_out].emitted;      // y's consumer left the planes of dL/dy (run-time: set after this node was built)
            float* dyS = ready ? BP(c, gsplits[gs_out].b_pl) : BP(c, b_dyS);
            if (!ready) BD_TRY(split_rows(c, dy, lddy, M, Cout, dyS));
            bd_conv3x3_ps_wgrad_desc w2 = {};
            w2.B = c.B; w2.H = H; w2.W = W; w2.Cin = Cout; w2.Cout = Cout;
            w2.x_split = U16(BP(c, b_a2s)); w2.ldx = Cout; w2.dy_split = U16(dyS); w2.lddy = Cout;
            w2.dw = c.grads + pc2w; w2.db = c.grads + pc2b;
            BD_TRY(conv_pw(c, w2));
            bd_conv3x3_ps_desc g2 = {};
            g2.B = c.B; g2.H = H; g2.W = W; g2.K = Cout; g2.N = Cout; g2.direction = -1;
            g2.x_split = U16(dyS); g2.ldx = Cout; g2.w_split = c.wT_split + 2 * pc2w; g2.out_scale = 1.f;
            g2.y = BP(c, b_da2); g2.ldy = Cout;
            BD_TRY(conv_p(c, g2));
        } else {
            bd_conv3x3_wgrad_desc w2 = {};
            w2.B = c.B; w2.Hs = H; w2.Ws = W; w2.Cin = Cout; w2.Cout = Cout; w2.stride = 1; w2.pad_t = 1; w2.pad_l = 1; w2.Ho = H; w2.Wo = W;
            w2.x = BP(c, b_a2); w2.ldx = Cout; w2.dy = dy; w2.lddy = lddy; w2.dw = c.grads + pc2w; w2.db = c.grads + pc2b;
            BD_TRY(conv_w(c, w2));
            bd_conv3x3_dgrad_desc g2 = {};
            g2.B = c.B; g2.Hs = H; g2.Ws = W; g2.Cin = Cout; g2.Cout = Cout; g2.stride = 1; g2.pad_t = 1; g2.pad_l = 1; g2.Ho = H; g2.Wo = W;
            g2.dy = dy; g2.lddy = lddy; g2.w = c.params + pc2w; g2.dx = BP(c, b_da2); g2.lddx = Cout;
            BD_TRY(conv_d(c, g2));
        }
        {   // norm2 backward: dh1 = gn_silu_bwd(h1, da2)   (h1 has no persistent grad buffer: write to scratch)
            bd_gn_bwd_desc d = {};
            d.B = c.B; d.HW = HW; d.C = Cout; d.G = G; d.silu = 1;
            d.x = BP(c, b_h1); d.ldx = Cout; d.gamma = c.params + pn2w; d.beta = c.params + pn2b;
            d.mean = MEANP(c, b_st2, G); d.rstd = RSTDP(c, b_st2, G);
            d.dy = BP(c, b_da2); d.lddy = Cout; d.dx = ps1 ? nullptr : BP(c, b_dh1); d.lddx = Cout; d.accumulate_dx = 0;
            d.dgamma = c.grads + pn2w; d.dbeta = c.grads + pn2b;
            d.workspace = c.opws; d.workspace_bytes = c.opws_bytes;
            // time-embedding gradient = per-sample column sums of dh1, out of the same launch
            d.dx_colsum = BP(c, b_dtproj) + toff; d.ld_colsum = sumC_;
            if (ps1) { d.dx_split = U16(BP(c, b_dh1S)); d.lddxs = Cout; }
            d.param_partials = c.gn_slot(d.B, d.HW, d.C, d.G, d.dgamma, d.dbeta);
            if (!c.dry) BD_TRY(bd_gn_bwd(&d, (bd_stream_t)c.st));
        }
        // (this resnet's rows of the batched time_emb_proj weight / bias gradient, dW = dtproj[:, rows]^T embs (resnet.py:571),
        //  are produced by ONE launch per backward segment over all of the segment's resnets: temb_wgrad below)
        if (c.dry) BD_TRY(linear_wgrad(c, BP(c, b_dtproj), sumC_, BP(c, b_embs_), T_, c.grads, c.B, sumC_, T_, c.grads));   // workspace bound
        if (ps1) {
            bd_conv3x3_ps_wgrad_desc w1 = {};
            w1.B = c.B; w1.H = H; w1.W = W; w1.Cin = Cin; w1.Cout = Cout;
            w1.x_split = U16(BP(c, b_a1s)); w1.ldx = Cin; w1.dy_split = U16(BP(c, b_dh1S)); w1.lddy = Cout;
            w1.dw = c.grads + pc1w; w1.db = c.grads + pc1b;
            BD_TRY(conv_pw(c, w1));
            bd_conv3x3_ps_desc g1 = {};
            g1.B = c.B; g1.H = H; g1.W = W; g1.K = Cout; g1.N = Cin; g1.direction = -1;
            g1.x_split = U16(BP(c, b_dh1S)); g1.ldx = Cout; g1.w_split = c.wT_split + 2 * pc1w; g1.out_scale = 1.f;
            g1.y = BP(c, b_da1); g1.ldy = Cin;
            BD_TRY(conv_p(c, g1));
        } else {
            bd_conv3x3_wgrad_desc w1 = {};
            w1.B = c.B; w1.Hs = H; w1.Ws = W; w1.Cin = Cin; w1.Cout = Cout; w1.stride = 1; w1.pad_t = 1; w1.pad_l = 1; w1.Ho = H; w1.Wo = W;
            w1.x = BP(c, b_a1); w1.ldx = Cin; w1.dy = BP(c, b_dh1); w1.lddy = Cout; w1.dw = c.grads + pc1w; w1.db = c.grads + pc1b;
            BD_TRY(conv_w(c, w1));
            bd_conv3x3_dgrad_desc g1 = {};
            g1.B = c.B; g1.Hs = H; g1.Ws = W; g1.Cin = Cin; g1.Cout = Cout; g1.stride = 1; g1.pad_t = 1; g1.pad_l = 1; g1.Ho = H; g1.Wo = W;
            g1.dy = BP(c, b_dh1); g1.lddy = Cout; g1.w = c.params + pc1w; g1.dx = BP(c, b_da1); g1.lddx = Cin;
            BD_TRY(conv_d(c, g1));
        }
        if (shortcut) {   // 1x1 shortcut convolution: its data gradient goes in FIRST, so that the norm1 backward below is the store that makes
            // dL/dx final (and can hand x's producer the split planes of it)
            BD_TRY(linear_wgrad(c, dy, lddy, VP(c, x), x.ld, c.grads + psw, M, Cout, Cin, c.grads + psb));
            const int acc = c.ginit[x.buf];
            c.ginit[x.buf] = 1;
            // dL/dy exists as split planes whenever conv2 took the plane kernels: the shortcut's data gradient then runs DMA-fed on them
            // (bd_gemm_sp, the weight's per-step plane copy taken K-major) instead of splitting both fp32 operands in the igemm loaders
            static const bool sc_sp = !(getenv("BD_SHORTCUT_SP") && atoi(getenv("BD_SHORTCUT_SP")) == 0);      // (A/B knob)
            if (sc_sp && ps2 && gemm_sp_supported(M, Cin, Cout) && x.ld % 4 == 0) {
                const bool ready = gs_out >= 0 && gsplits[gs_out].emitted;
                bd_gemm_sp_desc g = {};
                g.M = M; g.N = Cin; g.K = Cout; g.batch = 1;
                g.a = U16(ready ? BP(c, gsplits[gs_out].b_pl) : BP(c, b_dyS)); g.lda = Cout;
                g.b = c.w_split + 2 * psw; g.ldb = Cin; g.b_kmajor = 1;
                g.c = GP(c, x); g.ldc = x.ld; g.accumulate = acc; g.alpha = 1.f; g.out_scale = 1.f;
                BD_TRY(gemm_s(c, g));
            } else {
                BD_TRY(linear_dgrad(c, dy, lddy, c.params + psw, GP(c, x), x.ld, M, Cout, Cin, acc));
            }
        }
        // norm1 backward; the identity shortcut's gradient (dy itself) is added in the same store
        return gn_bwd(c, x, pn1w, pn1b, b_st1, BP(c, b_da1), Cin, 1, shortcut ? nullptr : dy, lddy, gs_in);
    });
}

void bd_unet::node_attention(const std::string& pre, const View& x, const View& y, float scale) {
    const int C = x.C, H = x.H, W = x.W, N = H * W;
    const int heads = cfg.attention_head_dim > 0 ? C / cfg.attention_head_dim : 1;
    const int dh = C / heads;
    const float sm_scale = 1.0f / std::sqrt((float)C / (float)heads);
    const int64_t pgw = add_param(pre + "group_norm.weight", {C}), pgb = add_param(pre + "group_norm.bias", {C});
    const int64_t pqw = add_param(pre + "query.weight", {C, C});
    add_param(pre + "key.weight", {C, C});
    add_param(pre + "value.weight", {C, C});
    const int64_t pqb = add_param(pre + "query.bias", {C});
    add_param(pre + "key.bias", {C});
    add_param(pre + "value.bias", {C});
    const int64_t ppw = add_param(pre + "proj_attn.weight", {C, C}), ppb = add_param(pre + "proj_attn.bias", {C});
    const int G = cfg.norm_num_groups;
    const int b_n = new_buf((int64_t)N * C, 0, R_VALUE), b_qkv = new_buf((int64_t)N * 3 * C, 0, R_VALUE);
    const int b_p = new_buf((int64_t)heads * N * N, 0, R_VALUE), b_o = new_buf((int64_t)N * C, 0, R_VALUE);
    const int b_st = new_buf(2 * G, 0, R_VALUE);
    const int b_dys = scale != 1.f ? scratch((int64_t)N * C) : -1;
    const int b_bs = scratch(3 * C), b_do = scratch((int64_t)N * C), b_dp = scratch((int64_t)heads * N * N);
    const int b_dqkv = scratch((int64_t)N * 3 * C), b_dn = scratch((int64_t)N * C);
    const int b_dyS = scratch((int64_t)N * C);
    const float inv = 1.f / scale;

    auto batched = [=](bd_igemm_desc& g, int B) {
        g.batch_outer = B; g.batch_inner = heads;
    };
    // Round 3: the whole block on SPLIT PLANES (gemm_sp.hip + attn_sp.hip): GroupNorm writes planes, the QKV projection reads and writes
    // planes, the attention core (scores, softmax, value product: one launch forward, two backward) keeps the [N, N] matrices on chip and
    // hands planes to the output projection; the backward mirrors it.  The same buffers hold planes instead of fp32 (same bytes); the
    // weights' planes are the per-step bd_split_bf16 copy (K-contiguous rows: the layout a [N][K] weight wants forward, and the K-major
    // operand of its data gradient as it stands).  N = 256 tokens, head dim 256 (attn_sp_supported); the mid block (4 x 4) stays below.
    auto use_sp = [=](const Ctx& c) {
        return cfg.compute_mode == BD_MODE_BF16X3 && (c.dry || c.w_split) && attn_sp_supported(N, dh) &&
               gemm_sp_supported((int)((int64_t)(c.LB > 0 ? c.LB : c.B) * N), C, C) && gemm_sp_supported(3 * C, C, 32);
    };
    const int gs_in = claim_gsplit(x, true);
    const int gs_out = b_dys < 0 ? reg_gsplit(y, [use_sp](const Ctx& c) { return use_sp(c); }) : -1;
    F([=](Ctx& c) {
        const int M = (int)rows(c, x);
        if (use_sp(c)) {
            bd_gn_fwd_desc g = {};
            g.B = c.B; g.HW = N; g.C = C; g.G = G; g.eps = cfg.norm_eps; g.silu = 0;
            g.x = VP(c, x); g.ldx = x.ld; g.gamma = c.params + pgw; g.beta = c.params + pgb;
            g.y = nullptr; g.ldy = C; g.y_split = U16(BP(c, b_n)); g.ldys = C;
            g.mean = MEANP(c, b_st, G); g.rstd = RSTDP(c, b_st, G);
            g.workspace = c.opws; g.workspace_bytes = c.opws_bytes;
            if (c.dry) {
                const size_t n = bd_gn_workspace_bytes(c.B, C);
                if (n > c.opws_need) c.opws_need = n;
            } else {
                BD_TRY(bd_gn_fwd(&g, (bd_stream_t)c.st));
            }
            bd_gemm_sp_desc q = {};      // qkv = n Wqkv^T + b  -> planes
            q.M = M; q.N = 3 * C; q.K = C; q.batch = 1;
            q.a = U16(BP(c, b_n)); q.lda = C; q.b = c.w_split + 2 * pqw; q.ldb = C;
            q.c_split = U16(BP(c, b_qkv)); q.ldcs = 3 * C; q.bias = c.params + pqb; q.alpha = 1.f; q.out_scale = 1.f;
            BD_TRY(gemm_s(c, q));
            if (!c.dry) {
                bd_attn_sp_desc a = {};
                a.B = c.B; a.heads = heads; a.N = N; a.dh = dh; a.qkv_split = U16(BP(c, b_qkv)); a.ld = 3 * C; a.scale = sm_scale;
                a.o_split = U16(BP(c, b_o)); a.ldo = C; a.pt_split = c.training ? U16(BP(c, b_p)) : nullptr;
                BD_TRY(attn_sp_fwd(a, c.st));
            }
            bd_gemm_sp_desc o = {};      // y = (o Wp^T + b + x) / scale
            o.M = M; o.N = C; o.K = C; o.batch = 1;
            o.a = U16(BP(c, b_o)); o.lda = C; o.b = c.w_split + 2 * ppw; o.ldb = C;
            o.c = VP(c, y); o.ldc = y.ld; o.bias = c.params + ppb; o.residual = VP(c, x); o.ldr = x.ld; o.alpha = 1.f; o.out_scale = inv;
            return gemm_s(c, o);
        }
        BD_TRY(gn_fwd(c, x, pgw, pgb, BP(c, b_n), C, b_st, 0));
        BD_TRY(linear_fwd(c, BP(c, b_n), C, c.params + pqw, c.params + pqb, BP(c, b_qkv), 3 * C, M, 3 * C, C));
        float* qkv = BP(c, b_qkv);
        {
            {   // S = scale * Q K^T
                bd_igemm_desc g = {};
                g.A = dense(qkv, 3 * C, 1); g.A.bs_outer = (int64_t)N * 3 * C; g.A.bs_inner = dh;
                g.B = dense(qkv + C, 3 * C, 1); g.B.bs_outer = (int64_t)N * 3 * C; g.B.bs_inner = dh;
                g.M = N; g.N = N; g.K = dh; batched(g, c.B);
                g.C = BP(c, b_p); g.ldc = N; g.c_bs_outer = (int64_t)heads * N * N; g.c_bs_inner = (int64_t)N * N;
                g.alpha = sm_scale; g.out_scale = 1.f;
                BD_TRY(igemm(c, g));
            }
            if (!c.dry) BD_TRY(bd_softmax_fwd(BP(c, b_p), BP(c, b_p), (int64_t)c.B * heads * N, N, (bd_stream_t)c.st));
            {   // O = P V
                bd_igemm_desc g = {};
                g.A = dense(BP(c, b_p), N, 1); g.A.bs_outer = (int64_t)heads * N * N; g.A.bs_inner = (int64_t)N * N;
                g.B = dense(qkv + 2 * C, 3 * C, 0); g.B.bs_outer = (int64_t)N * 3 * C; g.B.bs_inner = dh;
                g.M = N; g.N = dh; g.K = N; batched(g, c.B);
                g.C = BP(c, b_o); g.ldc = C; g.c_bs_outer = (int64_t)N * C; g.c_bs_inner = dh;
                g.alpha = 1.f; g.out_scale = 1.f;
                BD_TRY(igemm(c, g));
            }
        }
        return linear_fwd(c, BP(c, b_o), C, c.params + ppw, c.params + ppb, VP(c, y), y.ld, M, C, C, VP(c, x), x.ld, inv);
    });
    Bk([=](Ctx& c) {
        const int M = (int)rows(c, x);
        const float* dy = GP(c, y); int64_t lddy = y.ld;
        if (b_dys >= 0) {
            BD_TRY(add(c, dy, lddy, BP(c, b_dys), C, M, C, inv, 0));
            dy = BP(c, b_dys); lddy = C;
        }
        if (use_sp(c)) {
            const bool ready = gs_out >= 0 && gsplits[gs_out].emitted;
            float* dyS = ready ? BP(c, gsplits[gs_out].b_pl) : BP(c, b_dyS);
            if (!ready) BD_TRY(split_rows(c, dy, lddy, M, C, dyS));
            bd_gemm_sp_desc w = {};      // dWp = dy^T o, dbp = column sums of dy  (side stream)
            w.M = C; w.N = C; w.K = M; w.batch = 1;
            w.a = U16(dyS); w.lda = C; w.a_kmajor = 1; w.b = U16(BP(c, b_o)); w.ldb = C; w.b_kmajor = 1;
            w.c = c.grads + ppw; w.ldc = C; w.a_colsum = c.grads + ppb; w.alpha = 1.f; w.out_scale = 1.f;
            BD_TRY(gemm_s(c, w, true));
            bd_gemm_sp_desc g = {};      // dO = dy Wp -> planes
            g.M = M; g.N = C; g.K = C; g.batch = 1;
            g.a = U16(dyS); g.lda = C; g.b = c.w_split + 2 * ppw; g.ldb = C; g.b_kmajor = 1;
            g.c_split = U16(BP(c, b_do)); g.ldcs = C; g.alpha = 1.f; g.out_scale = 1.f;
            BD_TRY(gemm_s(c, g));
            if (!c.dry) {
                bd_attn_sp_desc a = {};
                a.B = c.B; a.heads = heads; a.N = N; a.dh = dh; a.qkv_split = U16(BP(c, b_qkv)); a.ld = 3 * C; a.scale = sm_scale;
                a.pt_split = U16(BP(c, b_p)); a.do_split = U16(BP(c, b_do)); a.lddo = C; a.dst_split = U16(BP(c, b_dp));
                a.dqkv_split = U16(BP(c, b_dqkv)); a.lddqkv = 3 * C;
                BD_TRY(attn_sp_bwd(a, c.st));
            }
            bd_gemm_sp_desc wq = {};     // dWqkv = dqkv^T n, dbqkv = column sums of dqkv  (side stream)
            wq.M = 3 * C; wq.N = C; wq.K = M; wq.batch = 1;
            wq.a = U16(BP(c, b_dqkv)); wq.lda = 3 * C; wq.a_kmajor = 1; wq.b = U16(BP(c, b_n)); wq.ldb = C; wq.b_kmajor = 1;
            wq.c = c.grads + pqw; wq.ldc = C; wq.a_colsum = c.grads + pqb; wq.alpha = 1.f; wq.out_scale = 1.f;
            BD_TRY(gemm_s(c, wq, true));
            bd_gemm_sp_desc dn = {};     // dn = dqkv Wqkv
            dn.M = M; dn.N = C; dn.K = 3 * C; dn.batch = 1;
            dn.a = U16(BP(c, b_dqkv)); dn.lda = 3 * C; dn.b = c.w_split + 2 * pqw; dn.ldb = C; dn.b_kmajor = 1;
            dn.c = BP(c, b_dn); dn.ldc = C; dn.alpha = 1.f; dn.out_scale = 1.f;
            BD_TRY(gemm_s(c, dn));
            return gn_bwd(c, x, pgw, pgb, b_st, BP(c, b_dn), C, 0, dy, lddy, gs_in);
        }
        float* qkv = BP(c, b_qkv); float* dqkv = BP(c, b_dqkv); float* P = BP(c, b_p); float* dP = BP(c, b_dp);
        float* dO = BP(c, b_do);
        BD_TRY(linear_wgrad(c, dy, lddy, BP(c, b_o), C, c.grads + ppw, M, C, C, c.grads + ppb));
        BD_TRY(linear_dgrad(c, dy, lddy, c.params + ppw, dO, C, M, C, C, 0));
        {   // dP = dO V^T
            bd_igemm_desc g = {};
            g.A = dense(dO, C, 1); g.A.bs_outer = (int64_t)N * C; g.A.bs_inner = dh;
            g.B = dense(qkv + 2 * C, 3 * C, 1); g.B.bs_outer = (int64_t)N * 3 * C; g.B.bs_inner = dh;
            g.M = N; g.N = N; g.K = dh; batched(g, c.B);
            g.C = dP; g.ldc = N; g.c_bs_outer = (int64_t)heads * N * N; g.c_bs_inner = (int64_t)N * N;
            g.alpha = 1.f; g.out_scale = 1.f;
            BD_TRY(igemm(c, g));
        }
        {   // dV = P^T dO
            bd_igemm_desc g = {};
            g.A = dense(P, N, 0); g.A.bs_outer = (int64_t)heads * N * N; g.A.bs_inner = (int64_t)N * N;
            g.B = dense(dO, C, 0); g.B.bs_outer = (int64_t)N * C; g.B.bs_inner = dh;
            g.M = N; g.N = dh; g.K = N; batched(g, c.B);
            g.C = dqkv + 2 * C; g.ldc = 3 * C; g.c_bs_outer = (int64_t)N * 3 * C; g.c_bs_inner = dh;
            g.alpha = 1.f; g.out_scale = 1.f;
            BD_TRY(igemm(c, g));
        }
        if (!c.dry) BD_TRY(bd_softmax_bwd(P, dP, dP, (int64_t)c.B * heads * N, N, (bd_stream_t)c.st));
        {   // dQ = scale * dS K
            bd_igemm_desc g = {};
            g.A = dense(dP, N, 1); g.A.bs_outer = (int64_t)heads * N * N; g.A.bs_inner = (int64_t)N * N;
            g.B = dense(qkv + C, 3 * C, 0); g.B.bs_outer = (int64_t)N * 3 * C; g.B.bs_inner = dh;
            g.M = N; g.N = dh; g.K = N; batched(g, c.B);
            g.C = dqkv; g.ldc = 3 * C; g.c_bs_outer = (int64_t)N * 3 * C; g.c_bs_inner = dh;
            g.alpha = sm_scale; g.out_scale = 1.f;
            BD_TRY(igemm(c, g));
        }
        {   // dK = scale * dS^T Q
            bd_igemm_desc g = {};
            g.A = dense(dP, N, 0); g.A.bs_outer = (int64_t)heads * N * N; g.A.bs_inner = (int64_t)N * N;
            g.B = dense(qkv, 3 * C, 0); g.B.bs_outer = (int64_t)N * 3 * C; g.B.bs_inner = dh;
            g.M = N; g.N = dh; g.K = N; batched(g, c.B);
            g.C = dqkv + C; g.ldc = 3 * C; g.c_bs_outer = (int64_t)N * 3 * C; g.c_bs_inner = dh;
            g.alpha = sm_scale; g.out_scale = 1.f;
            BD_TRY(igemm(c, g));
        }
        BD_TRY(linear_wgrad(c, dqkv, 3 * C, BP(c, b_n), C, c.grads + pqw, M, 3 * C, C, c.grads + pqb));
        BD_TRY(linear_dgrad(c, dqkv, 3 * C, c.params + pqw, BP(c, b_dn), C, M, 3 * C, C, 0));
        // group-norm backward; the residual connection's gradient (dy itself) is added in the same store
        return gn_bwd(c, x, pgw, pgb, b_st, BP(c, b_dn), C, 0, dy, lddy, gs_in);
    });
}

void bd_unet::node_downsample(const std::string& pre, const View& x, const View& y) {
    const int C = x.C, H = x.H, W = x.W, Ho = y.H, Wo = y.W;
    const int pad = cfg.downsample_padding ? 1 : 0;
    const int64_t pw = add_param(pre + "conv.weight", {C, C, 3, 3}, 1), pb = add_param(pre + "conv.bias", {C});
    if (C % 32 == 0) { wt_off.push_back(pw); wt_cin.push_back(C); wt_cout.push_back(C); }   // transposed planes for the phase data gradient
    const int b_bs = scratch(C), b_dyS = scratch((int64_t)Ho * Wo * C);
    claim_gsplit(x, false);        // (this node's data gradient is a convolution: it cannot leave dL/dx as planes)
    const int gs_out = reg_gsplit(y, [this, H, W, Ho, Wo, C](const Ctx& c) { return H == 2 * Ho && W == 2 * Wo && phase_ok(c, Ho, Wo, C, C); });
    F([=](Ctx& c) {
        bd_conv3x3_fwd_desc d = {};
        d.B = c.B; d.Hs = H; d.Ws = W; d.Cin = C; d.Cout = C; d.stride = 2; d.pad_t = pad; d.pad_l = pad; d.Ho = Ho; d.Wo = Wo;
        d.x = VP(c, x); d.ldx = x.ld; d.w = c.params + pw; d.bias = c.params + pb; d.out_scale = 1.f;
        d.y = VP(c, y); d.ldy = y.ld;
        return conv_f(c, d);
    });
    Bk([=](Ctx& c) {
        const float* dy = GP(c, y);
        bd_conv3x3_wgrad_desc w = {};
        w.B = c.B; w.Hs = H; w.Ws = W; w.Cin = C; w.Cout = C; w.stride = 2; w.pad_t = pad; w.pad_l = pad; w.Ho = Ho; w.Wo = Wo;
        w.x = VP(c, x); w.ldx = x.ld; w.dy = dy; w.lddy = y.ld; w.dw = c.grads + pw; w.db = c.grads + pb;
        BD_TRY(conv_w(c, w));
        const int acc = c.ginit[x.buf];
        c.ginit[x.buf] = 1;
        if (H == 2 * Ho && W == 2 * Wo && phase_ok(c, Ho, Wo, C, C)) {
            // by the parity of the input pixel only 4 / 2 / 2 / 1 of the nine taps contribute: four classes on the output grid
            const bool ready = gs_out >= 0 && gsplits[gs_out].emitted;
            float* dyS = ready ? BP(c, gsplits[gs_out].b_pl) : BP(c, b_dyS);
            if (!ready) BD_TRY(split_rows(c, dy, y.ld, (int64_t)c.B * Ho * Wo, C, dyS));
            if (c.dry) return (int)BD_OK;
            bd_conv3x3_s2_dgrad_desc g = {};
            g.B = c.B; g.Ho = Ho; g.Wo = Wo; g.Cin = C; g.Cout = C; g.pad = pad;
            g.dy_split = U16(dyS); g.lddy = C; g.wT_split = c.wT_split + 2 * pw;
            g.dx = GP(c, x); g.lddx = x.ld; g.accumulate = acc;
            return conv3x3_s2_dgrad_ps(g, c.st);
        }
        bd_conv3x3_dgrad_desc g = {};
        g.B = c.B; g.Hs = H; g.Ws = W; g.Cin = C; g.Cout = C; g.stride = 2; g.pad_t = pad; g.pad_l = pad; g.Ho = Ho; g.Wo = Wo;
        g.dy = dy; g.lddy = y.ld; g.w = c.params + pw; g.dx = GP(c, x); g.lddx = x.ld; g.accumulate = acc;
        return conv_d(c, g);
    });
}

void bd_unet::node_upsample(const std::string& pre, const View& x, const View& y) {
    const int C = x.C, H = x.H, W = x.W;
    const int64_t pw = add_param(pre + "conv.weight", {C, C, 3, 3}, 1), pb = add_param(pre + "conv.bias", {C});
    if (C % 32 == 0) { wt_off.push_back(pw); wt_cin.push_back(C); wt_cout.push_back(C); }
    const int b_bs = scratch(C), b_du = scratch((int64_t)4 * H * W * C);
    // LDS-DMA path: the nearest-upsampled input is materialised ONCE as split planes (4 B per element of the 2H x 2W
    // grid) and feeds the stride-1 "same" kernels of conv_ps.hip: forward, weight gradient and (through b_du + 2x2 sums)
    // the data gradient.  Otherwise the upsampling stays folded into the igemm gather.
    const int b_xuS = new_buf((int64_t)4 * H * W * C, 0, R_VALUE), b_dyS = scratch((int64_t)4 * H * W * C);
    // phase-decomposed path (round 3): x as split planes on the SOURCE grid + the pre-summed tap planes E / E^T of this layer
    const int b_xS = new_buf((int64_t)H * W * C, 0, R_VALUE);
    const int b_e = new_buf(0, (int64_t)16 * C * C, R_VALUE), b_et = new_buf(0, (int64_t)16 * C * C, R_VALUE);
    if (C % 128 == 0) upsw.push_back({pw, C, H, W, b_e, b_et});
    claim_gsplit(x, false);
    const int gs_out = reg_gsplit(y, [this, H, W, C](const Ctx& c) { return phase_ok(c, H, W, C, C) || ps_ok(c, 2 * H, 2 * W, C, C); });
    {   // which of the two operand sets exists is the plan's path choice for the batch the workspace is laid out for
        auto phase_at = [=](int B) { Ctx q; q.dry = true; q.B = q.LB = B; return phase_ok(q, H, W, C, C); };
        auto ps_at = [=](int B) { Ctx q; q.dry = true; q.B = q.LB = B; return ps_ok(q, 2 * H, 2 * W, C, C); };
        bufs[b_xuS].live = [=](int B, int) { return !phase_at(B) && ps_at(B); };
        bufs[b_xS].live = [=](int B, int) { return phase_at(B); };
        bufs[b_e].live = [=](int B, int) { return phase_at(B); };
        bufs[b_et].live = [=](int B, int training) { return training && phase_at(B); };
    }
    F([=](Ctx& c) {
        if (phase_ok(c, H, W, C, C)) {
            BD_TRY(split_rows(c, VP(c, x), x.ld, (int64_t)c.B * H * W, C, BP(c, b_xS)));
            if (c.dry) return (int)BD_OK;
            bd_upsample_conv_desc d = {};
            d.B = c.B; d.H = H; d.W = W; d.Cin = C; d.Cout = C;
            d.x_split = U16(BP(c, b_xS)); d.ldx = C; d.e_split = U16(BP(c, b_e)); d.bias = c.params + pb;
            d.y = VP(c, y); d.ldy = y.ld;
            return upsample_conv_fwd(d, c.st);
        }
        if (ps_ok(c, 2 * H, 2 * W, C, C)) {
            if (!c.dry) BD_TRY(bd_split_rows_ups2(VP(c, x), x.ld, c.B, H, W, C, U16(BP(c, b_xuS)), C, (bd_stream_t)c.st));
            bd_conv3x3_ps_desc d = {};
            d.B = c.B; d.H = 2 * H; d.W = 2 * W; d.K = C; d.N = C; d.direction = 1;
            d.x_split = U16(BP(c, b_xuS)); d.ldx = C; d.w_split = c.w_split + 2 * pw; d.bias = c.params + pb; d.out_scale = 1.f;
            d.y = VP(c, y); d.ldy = y.ld;
            return conv_p(c, d);
        }
        bd_conv3x3_fwd_desc d = {};
        d.B = c.B; d.Hs = H; d.Ws = W; d.Cin = C; d.Cout = C; d.stride = 1; d.pad_t = 1; d.pad_l = 1; d.ups = 1; d.Ho = 2 * H; d.Wo = 2 * W;
        d.x = VP(c, x); d.ldx = x.ld; d.w = c.params + pw; d.bias = c.params + pb; d.out_scale = 1.f;
        d.y = VP(c, y); d.ldy = y.ld;
        return conv_f(c, d);
    });
    Bk([=](Ctx& c) {
        const float* dy = GP(c, y);
        const int acc = c.ginit[x.buf];
        c.ginit[x.buf] = 1;
        const bool ready = gs_out >= 0 && gsplits[gs_out].emitted;
        float* dyS = ready ? BP(c, gsplits[gs_out].b_pl) : BP(c, b_dyS);
        if (phase_ok(c, H, W, C, C)) {
            if (!ready) BD_TRY(split_rows(c, dy, y.ld, (int64_t)c.B * 4 * H * W, C, dyS));
            bd_upsample_conv_desc d = {};
            d.B = c.B; d.H = H; d.W = W; d.Cin = C; d.Cout = C;
            d.x_split = U16(BP(c, b_xS)); d.ldx = C; d.dy_split = U16(dyS); d.lddy = C;
            d.et_split = U16(BP(c, b_et)); d.dx = GP(c, x); d.lddx = x.ld; d.accumulate = acc;
            d.dw = c.grads + pw; d.db = c.grads + pb;
            // ONE class of 16 taps: >= 128 tiles of its own, or (taps dealt to four workgroups per tile) >= 32
            const bool ph_d = phase_ok(c, H, W, C, C, upsample_conv_dgrad_workspace_bytes(d) ? 4 : 1);
            if (c.dry) {
                size_t n = upsample_conv_wgrad_workspace_bytes(d);
                if (n > c.opws_need) c.opws_need = n;
                n = upsample_conv_dgrad_workspace_bytes(d);
                if (n > c.opws_need) c.opws_need = n;
                if (!ph_d) note_conv(c);
                return (int)BD_OK;
            }
            d.workspace_bytes = c.opws_bytes;
            BD_TRY(on_aux(c, [&](hipStream_t st, char* ws) { d.workspace = ws; return upsample_conv_wgrad(d, st); }));
            if (ph_d) {      // 16 taps on dY sampled at stride 2: replaces the fine-grid dgrad + 2x2 sum
                d.workspace = c.opws;
                return upsample_conv_dgrad(d, c.st);
            }
            bd_conv3x3_ps_desc g = {};                           // literal data gradient on the fine grid + 2x2 sums
            g.B = c.B; g.H = 2 * H; g.W = 2 * W; g.K = C; g.N = C; g.direction = -1;
            g.x_split = U16(dyS); g.ldx = C; g.w_split = c.wT_split + 2 * pw; g.out_scale = 1.f;
            g.y = BP(c, b_du); g.ldy = C;
            BD_TRY(conv_p(c, g));
            return bd_sum2x2(BP(c, b_du), C, GP(c, x), x.ld, c.B, H, W, C, acc, (bd_stream_t)c.st);
        }
        if (ps_ok(c, 2 * H, 2 * W, C, C)) {
            if (!ready) BD_TRY(split_rows(c, dy, y.ld, (int64_t)c.B * 4 * H * W, C, dyS));
            bd_conv3x3_ps_wgrad_desc w = {};
            w.B = c.B; w.H = 2 * H; w.W = 2 * W; w.Cin = C; w.Cout = C;
            w.x_split = U16(BP(c, b_xuS)); w.ldx = C; w.dy_split = U16(dyS); w.lddy = C;
            w.dw = c.grads + pw; w.db = c.grads + pb;
            BD_TRY(conv_pw(c, w));
            bd_conv3x3_ps_desc g = {};
            g.B = c.B; g.H = 2 * H; g.W = 2 * W; g.K = C; g.N = C; g.direction = -1;
            g.x_split = U16(dyS); g.ldx = C; g.w_split = c.wT_split + 2 * pw; g.out_scale = 1.f;
            g.y = BP(c, b_du); g.ldy = C;
            BD_TRY(conv_p(c, g));
        } else {
            bd_conv3x3_wgrad_desc w = {};
            w.B = c.B; w.Hs = H; w.Ws = W; w.Cin = C; w.Cout = C; w.stride = 1; w.pad_t = 1; w.pad_l = 1; w.ups = 1; w.Ho = 2 * H; w.Wo = 2 * W;
            w.x = VP(c, x); w.ldx = x.ld; w.dy = dy; w.lddy = y.ld; w.dw = c.grads + pw; w.db = c.grads + pb;
            BD_TRY(conv_w(c, w));
            bd_conv3x3_dgrad_desc g = {};
            g.B = c.B; g.Hs = H; g.Ws = W; g.Cin = C; g.Cout = C; g.stride = 1; g.pad_t = 1; g.pad_l = 1; g.ups = 1; g.Ho = 2 * H; g.Wo = 2 * W;
            g.dy = dy; g.lddy = y.ld; g.w = c.params + pw; g.dx = BP(c, b_du); g.lddx = C;
            BD_TRY(conv_d(c, g));
        }
        if (c.dry) return (int)BD_OK;
        return bd_sum2x2(BP(c, b_du), C, GP(c, x), x.ld, c.B, H, W, C, acc, (bd_stream_t)c.st);
    });
}

void bd_unet::node_conv_out(const View& x) {
    const int C = x.C, H = x.H, W = x.W, Co = cfg.out_channels;
    const int64_t pnw = add_param("conv_norm_out.weight", {C}), pnb = add_param("conv_norm_out.bias", {C});
    const int64_t pw = add_param("conv_out.weight", {Co, C, 3, 3}, 1), pb = add_param("conv_out.bias", {Co});
    const int G = cfg.norm_num_groups;
    const int b_a = new_buf((int64_t)H * W * C, 0, R_VALUE), b_st = new_buf(2 * G, 0, R_VALUE);
    const int b_bs = scratch(Co), b_da = scratch((int64_t)H * W * C);
    const int gs_in = claim_gsplit(x, true);
    F([=](Ctx& c) {
        BD_TRY(gn_fwd(c, x, pnw, pnb, BP(c, b_a), C, b_st, 1));
        bd_conv3x3_fwd_desc d = {};
        d.B = c.B; d.Hs = H; d.Ws = W; d.Cin = C; d.Cout = Co; d.stride = 1; d.pad_t = 1; d.pad_l = 1; d.Ho = H; d.Wo = W;
        d.x = BP(c, b_a); d.ldx = C; d.w = c.params + pw; d.bias = c.params + pb; d.out_scale = 1.f;
        d.y = c.out; d.ldy = c.ldo;
        return conv_f(c, d);
    });
    Bk([=](Ctx& c) {
        bd_conv3x3_wgrad_desc w = {};
        w.B = c.B; w.Hs = H; w.Ws = W; w.Cin = C; w.Cout = Co; w.stride = 1; w.pad_t = 1; w.pad_l = 1; w.Ho = H; w.Wo = W;
        w.x = BP(c, b_a); w.ldx = C; w.dy = c.dout; w.lddy = c.lddo; w.dw = c.grads + pw;
        if (conv3x3_wgrad_is_thin(w)) w.db = c.grads + pb;                    // the direct kernel sums dy on the way
        else BD_TRY(bias_grad(c, c.dout, c.lddo, H * W, Co, b_bs, c.grads + pb));   // Cout = 3: no igemm row-sum fusion
        BD_TRY(conv_w(c, w));
        bd_conv3x3_dgrad_desc g = {};
        g.B = c.B; g.Hs = H; g.Ws = W; g.Cin = C; g.Cout = Co; g.stride = 1; g.pad_t = 1; g.pad_l = 1; g.Ho = H; g.Wo = W;
        g.dy = c.dout; g.lddy = c.lddo; g.w = c.params + pw; g.dx = BP(c, b_da); g.lddx = C;
        BD_TRY(conv_d(c, g));
        return gn_bwd(c, x, pnw, pnb, b_st, BP(c, b_da), C, 1, nullptr, 0, gs_in);
    });
}

// ----------------------------------------------------------------------------------------------------
void bd_unet::build() {
    const int n = cfg.num_blocks, L = cfg.layers_per_block, S0 = cfg.sample_size;
    const int* boc = cfg.block_out_channels;
    T = 4 * boc[0];
    // ---- pass 1: architecture enumeration -----------------------------------------------------------
    // skips (down path outputs, in push order) and up resnets (pop order)
    struct Skip { int C, H; };
    std::vector<Skip> skips;
    skips.push_back({boc[0], S0});
    {
        int res = S0;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < L; ++j) skips.push_back({boc[i], res});
            if (i != n - 1) { res /= 2; skips.push_back({boc[i], res}); }
        }
    }
    struct UpRes { int C1, C2, Cout, H; };
    std::vector<std::vector<UpRes>> ups(n);
    {
        int sk = (int)skips.size();
        int res = S0 >> (n - 1);
        for (int i = 0; i < n; ++i) {
            const int out = boc[n - 1 - i];
            const int prev = i == 0 ? boc[n - 1] : boc[n - i];
            for (int j = 0; j <= L; ++j) {
                const Skip& s = skips[--sk];
                ups[i].push_back({j == 0 ? prev : out, s.C, out, res});
            }
            if (i != n - 1) res *= 2;
        }
    }
    // total time_emb_proj rows, in forward order: down resnets, mid x2, up resnets
    sumC = 0;
    for (int i = 0; i < n; ++i) sumC += L * boc[i];
    sumC += 2 * boc[n - 1];
    for (int i = 0; i < n; ++i) for (auto& u : ups[i]) sumC += u.Cout;

    // segments in BACKWARD order: 0 = out, 1..n = up blocks n-1..0, n+1 = mid, n+2..2n+1 = down n-1..0, 2n+2 = conv_in, 2n+3 = time
    const int nseg = 2 * n + 4;
    segs.assign(nseg, Seg{-1, -1});
    auto seg_out = 0;
    auto seg_up = [&](int i) { return 1 + (n - 1 - i); };
    const int seg_mid = n + 1;
    auto seg_down = [&](int i) { return n + 2 + (n - 1 - i); };
    const int seg_in = 2 * n + 2, seg_time = 2 * n + 3;

    // concat buffers, one per up resnet
    std::vector<std::vector<View>> cat(n);
    for (int i = 0; i < n; ++i)
        for (auto& u : ups[i]) {
            View v = new_tensor(u.H, u.H, u.C1 + u.C2);
            cat[i].push_back(v);
        }
    // skip k lives in the concat buffer of the up resnet that pops it
    std::vector<View> skipv(skips.size());
    {
        int sk = (int)skips.size();
        for (int i = 0; i < n; ++i)
            for (size_t j = 0; j < ups[i].size(); ++j) {
                --sk;
                skipv[sk] = slice(cat[i][j], ups[i][j].C1, ups[i][j].C2);
            }
    }

    // ---- pass 2: nodes in forward order ---------------------------------------------------------------
    cur_group = 0;
    cur_seg = seg_time; segs[seg_time].lo = 0;
    node_time_embed();
    int toff = 0, group = 1;
    cur_seg = seg_in; cur_group = group++;
    node_conv_in(skipv[0]);
    int sk = 1;
    View h = skipv[0];
    int res = S0;
    for (int i = 0; i < n; ++i) {
        cur_seg = seg_down(i);
        const std::string bp = "down_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < L; ++j) {
            cur_group = group++;
            View dst = skipv[sk];
            if (cfg.down_attn[i]) {
                View mid = new_tensor(res, res, boc[i]);
                node_resnet(bp + "resnets." + std::to_string(j) + ".", h, mid, toff, 1.f);
                cur_group = group++;
                node_attention(bp + "attentions." + std::to_string(j) + ".", mid, dst, 1.f);
            } else {
                node_resnet(bp + "resnets." + std::to_string(j) + ".", h, dst, toff, 1.f);
            }
            toff += boc[i];
            h = dst; ++sk;
        }
        if (i != n - 1) {
            cur_group = group++;
            View dst = skipv[sk];
            node_downsample(bp + "downsamplers.0.", h, dst);
            h = dst; ++sk; res /= 2;
        }
    }
    {   // mid block
        cur_seg = seg_mid;
        const int C = boc[n - 1];
        const float s = cfg.mid_block_scale_factor;
        View m1 = new_tensor(res, res, C), m2 = new_tensor(res, res, C);
        View dst = slice(cat[0][0], 0, ups[0][0].C1);
        cur_group = group++;
        node_resnet("mid_block.resnets.0.", h, m1, toff, s); toff += C;
        cur_group = group++;
        node_attention("mid_block.attentions.0.", m1, m2, s);
        cur_group = group++;
        node_resnet("mid_block.resnets.1.", m2, dst, toff, s); toff += C;
    }
    View last;
    for (int i = 0; i < n; ++i) {
        cur_seg = seg_up(i);
        const std::string bp = "up_blocks." + std::to_string(i) + ".";
        const int nl = (int)ups[i].size();
        for (int j = 0; j < nl; ++j) {
            const UpRes& u = ups[i][j];
            // destination of this layer's output: next concat's h-slice, or a standalone tensor before upsample / out
            View dst;
            const bool last_layer = j == nl - 1;
            if (!last_layer) dst = slice(cat[i][j + 1], 0, ups[i][j + 1].C1);
            else dst = new_tensor(u.H, u.H, u.Cout);
            cur_group = group++;
            if (cfg.up_attn[i]) {
                View mid = new_tensor(u.H, u.H, u.Cout);
                node_resnet(bp + "resnets." + std::to_string(j) + ".", cat[i][j], mid, toff, 1.f);
                cur_group = group++;
                node_attention(bp + "attentions." + std::to_string(j) + ".", mid, dst, 1.f);
            } else {
                node_resnet(bp + "resnets." + std::to_string(j) + ".", cat[i][j], dst, toff, 1.f);
            }
            toff += u.Cout;
            last = dst;
        }
        if (i != n - 1) {
            cur_group = group++;
            View dst = slice(cat[i + 1][0], 0, ups[i + 1][0].C1);
            node_upsample(bp + "upsamplers.0.", last, dst);
        }
    }
    cur_seg = seg_out; cur_group = group++;
    node_conv_out(last);
    (void)seg_out;
    nparams = (nparams + 3) / 4 * 4;
}

void bd_unet::layout(int B, int training) {
    if (lay_B == B && lay_train == training && lay_gen == bd::g_tune_gen) return;
    auto al = [](int64_t x) { return (x + 63) / 64 * 64; };
    int64_t v = 0, g = 0;
    for (auto& b : bufs) if (b.region == R_VALUE) { b.off = v; if (!b.live || b.live(B, training)) v += al(b.per_sample * B + b.fixed); }
    value_floats = v;
    for (auto& b : bufs) if (b.region == R_GRAD) { b.off = v + g; if (!b.live || b.live(B, training)) g += al(b.per_sample * B + b.fixed); }
    grad_floats = training ? g : 0;
    const int64_t base = value_floats + grad_floats;
    int64_t smax = 0;
    for (int pass = 0; pass < 2; ++pass) {   // two arenas, chosen by node (group) parity: see Ctx::st2
        int cur = -1; int64_t s = 0;
        for (auto& b : bufs) if (b.region == R_SCRATCH) {
            if (b.group != cur) { cur = b.group; s = 0; }
            const int64_t n = al(b.per_sample * B + b.fixed);
            if (pass == 0) { if (s + n > smax) smax = s + n; }
            else b.off = base + (b.group & 1) * smax + s;
            s += n;
        }
    }
    scratch_floats = training ? 2 * smax : 0;
    // op workspace: dry-run every step
    Ctx c;
    c.dry = true; c.B = B; c.ws = nullptr; c.ginit.assign(bufs.size(), 0);
    c.opws_bytes = (size_t)1 << 62;
    for (auto& f : fwd) f(c);
    if (training) for (auto it = bwd.rbegin(); it != bwd.rend(); ++it) it->fn(c);
    opws_bytes = align_up(c.opws_need, 256);
    gnpart_floats = training ? c.gnpart_need : 0;
    lay_B = B; lay_train = training; lay_gen = bd::g_tune_gen;
    prep_params = nullptr; prep_ws = nullptr; prep_B = -1;   // a new layout moves the prepared planes inside the workspace
}

// ----------------------------------------------------------------------------------------------------
extern "C" int bd_unet_create(const bd_unet_config* cfg, bd_unet** out) {
    BD_CHECK(cfg && out, BD_ERR_INVALID, "bd_unet_create: null argument");
    const int n = cfg->num_blocks;
    BD_CHECK(n >= 1 && n <= 8, BD_ERR_INVALID, "bd_unet_create: num_blocks %d out of range", n);
    BD_CHECK(cfg->layers_per_block >= 1, BD_ERR_INVALID, "bd_unet_create: layers_per_block");
    BD_CHECK(cfg->sample_size > 0 && (cfg->sample_size % (1 << (n - 1))) == 0, BD_ERR_INVALID,
             "bd_unet_create: sample_size %d not divisible by 2^%d", cfg->sample_size, n - 1);
    BD_CHECK(cfg->in_channels > 0 && cfg->out_channels > 0, BD_ERR_INVALID, "bd_unet_create: channels");
    BD_CHECK(cfg->norm_num_groups > 0, BD_ERR_INVALID, "bd_unet_create: norm_num_groups");
    for (int i = 0; i < n; ++i) {
        const int C = cfg->block_out_channels[i];
        BD_CHECK(C > 0 && C % cfg->norm_num_groups == 0 && C % 4 == 0, BD_ERR_UNSUPPORTED,
                 "bd_unet_create: block_out_channels[%d]=%d must be a multiple of norm_num_groups and of 4", i, C);
        if (cfg->attention_head_dim > 0 && (cfg->down_attn[i] || cfg->up_attn[n - 1 - i]))
            BD_CHECK(C % cfg->attention_head_dim == 0 && cfg->attention_head_dim % 4 == 0, BD_ERR_UNSUPPORTED,
                     "bd_unet_create: attention_head_dim %d must divide %d and be a multiple of 4", cfg->attention_head_dim, C);
    }
    BD_CHECK(cfg->mid_block_scale_factor != 0.f, BD_ERR_INVALID, "bd_unet_create: mid_block_scale_factor == 0");
    BD_CHECK(cfg->compute_mode == BD_MODE_F32 || cfg->compute_mode == BD_MODE_BF16X3, BD_ERR_INVALID, "bd_unet_create: compute_mode %d", cfg->compute_mode);
    bd_unet* u = new (std::nothrow) bd_unet();
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_create: out of host memory");
    u->cfg = *cfg;
    u->build();
    *out = u;
    return BD_OK;
}
extern "C" void bd_unet_destroy(bd_unet* u) {
    if (!u) return;
    if (u->aux_stream) {
        (void)hipStreamSynchronize(u->aux_stream);
        (void)hipEventDestroy(u->aux_ev_fork);
        (void)hipEventDestroy(u->aux_ev_join[0]);
        (void)hipEventDestroy(u->aux_ev_join[1]);
        (void)hipEventDestroy(u->aux_ev_seg);
        (void)hipStreamDestroy(u->aux_stream);
    }
    delete u;
}
extern "C" int bd_unet_set_aux_stream(bd_unet* u, int enabled) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_set_aux_stream: null plan");
    u->aux_enabled = enabled ? 1 : 0;
    return BD_OK;
}
extern "C" int bd_unet_set_deferred_join(bd_unet* u, int enabled) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_set_deferred_join: null plan");
    u->aux_defer = enabled ? 1 : 0;
    return BD_OK;
}
extern "C" int bd_unet_stream_wait_aux(bd_unet* u, bd_stream_t stream) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_stream_wait_aux: null plan");
    if (!u->aux_stream || !u->aux_ev_seg) return BD_OK;      // no side stream yet: nothing to wait for
    BD_HIP_TRY(hipStreamWaitEvent(S(stream), u->aux_ev_seg, 0));
    return BD_OK;
}
extern "C" int bd_unet_set_static_weights(bd_unet* u, int enabled) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_set_static_weights: null plan");
    u->static_weights = enabled ? 1 : 0;
    u->prep_params = nullptr; u->prep_ws = nullptr; u->prep_B = -1;     // turning it on or off always re-reads the weights once
    return BD_OK;
}
extern "C" int bd_unet_set_compute_mode(bd_unet* u, int mode) {
    BD_CHECK(u && (mode == BD_MODE_F32 || mode == BD_MODE_BF16X3), BD_ERR_INVALID, "bd_unet_set_compute_mode: bad arguments");
    if (u->cfg.compute_mode != mode) {
        u->lay_B = -1;   // the op-workspace bound depends on which kernels the mode selects: lay out again
        u->prep_params = nullptr; u->prep_ws = nullptr; u->prep_B = -1;   // ... and the prepared weight planes belong to the old mode
    }
    u->cfg.compute_mode = mode;
    return BD_OK;
}
extern "C" int bd_unet_reset_static_cache(bd_unet* u) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet_reset_static_cache: null plan");
    u->prep_params = nullptr; u->prep_ws = nullptr; u->prep_B = -1;
    return BD_OK;
}
extern "C" int64_t bd_unet_num_params(const bd_unet* u) { return u ? u->nparams : 0; }
extern "C" int bd_unet_num_tensors(const bd_unet* u) { return u ? (int)u->params.size() : 0; }
extern "C" int bd_unet_param_info(const bd_unet* u, int i, const char** name, int64_t* offset, int* rank, int64_t shape[4],
                                  int* layout) {
    BD_CHECK(u && i >= 0 && i < (int)u->params.size(), BD_ERR_INVALID, "bd_unet_param_info: index out of range");
    const Param& p = u->params[i];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.off;
    if (rank) *rank = p.rank;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    if (layout) *layout = p.layout;
    return BD_OK;
}
// the pre-split copy of the weights (2 uint16 per parameter, whole 32-element blocks) lives behind the op workspace
static size_t wsplit_elems(const bd_unet* u) { return (size_t)(u->nparams / 32 * 32); }
static size_t wsplit_bytes(const bd_unet* u) { return align_up(wsplit_elems(u) * 2 * sizeof(uint16_t), 256); }

extern "C" size_t bd_unet_workspace_bytes(bd_unet* u, int B, int training) {
    if (!u || B <= 0) return 0;
    u->layout(B, training);
    return (size_t)(u->value_floats + u->grad_floats + u->scratch_floats) * sizeof(float) + u->opws_bytes + 256 +
           2 * wsplit_bytes(u) + align_up(u->opws_bytes, 256) +   // + transposed conv-weight planes + the side stream's op workspace
           align_up(u->gnpart_floats * sizeof(float), 256);      // + GroupNorm parameter partials of one backward
}

static int unet_ctx(bd_unet* u, Ctx& c, int B, int training, void* workspace, size_t workspace_bytes) {
    BD_CHECK(u, BD_ERR_INVALID, "bd_unet: null plan");
    BD_CHECK(B > 0, BD_ERR_INVALID, "bd_unet: B must be > 0");
    const size_t need = bd_unet_workspace_bytes(u, B, training);
    BD_CHECK(workspace && workspace_bytes >= need, BD_ERR_WORKSPACE, "bd_unet: workspace %zu < %zu bytes", workspace_bytes, need);
    BD_CHECK(aligned16(workspace), BD_ERR_INVALID, "bd_unet: workspace must be 16-byte aligned");
    c.B = B; c.LB = B;
    c.ws = reinterpret_cast<float*>(workspace);
    const size_t fl = (size_t)(u->value_floats + u->grad_floats + u->scratch_floats) * sizeof(float);
    c.opws = reinterpret_cast<char*>(workspace) + align_up(fl, 256);
    c.opws_bytes = u->opws_bytes;
    if (u->cfg.compute_mode == BD_MODE_BF16X3) {
        c.w_split = reinterpret_cast<const uint16_t*>(c.opws + align_up(u->opws_bytes, 256));
    }
    c.opws2 = c.opws + align_up(u->opws_bytes, 256) + 2 * wsplit_bytes(u);
    c.gnpart = u->gnpart_floats ? reinterpret_cast<float*>(c.opws2 + align_up(u->opws_bytes, 256)) : nullptr;
    c.gnpart_floats = u->gnpart_floats; c.gnpart_used = 0;
    if (c.w_split) c.wT_split = reinterpret_cast<const uint16_t*>(c.opws + align_up(u->opws_bytes, 256) + wsplit_bytes(u));
    c.ginit.assign(u->bufs.size(), 0);
    return BD_OK;
}

// the plan's side stream and its events, created on first use (plan creation stays host-only)
static int unet_aux_init(bd_unet* u) {
    if (u->aux_stream) return BD_OK;
    int lo = 0, hi = 0;
    BD_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lowest priority: in backward the dgrad chain is the critical path
    const int prio = getenv("BD_AUX_PRIO") ? atoi(getenv("BD_AUX_PRIO")) : lo;     // (A/B knob; `lo` = lowest)
    BD_HIP_TRY(hipStreamCreateWithPriority(&u->aux_stream, hipStreamNonBlocking, prio));
    BD_HIP_TRY(hipEventCreateWithFlags(&u->aux_ev_fork, hipEventDisableTiming));
    BD_HIP_TRY(hipEventCreateWithFlags(&u->aux_ev_join[0], hipEventDisableTiming));
    BD_HIP_TRY(hipEventCreateWithFlags(&u->aux_ev_join[1], hipEventDisableTiming));
    BD_HIP_TRY(hipEventCreateWithFlags(&u->aux_ev_seg, hipEventDisableTiming));
    return BD_OK;
}

extern "C" int bd_unet_forward(bd_unet* u, int B, int training, const float* params, const float* x, int64_t ldx,
                               const int64_t* t, int t_stride, float* out, int64_t ldo, void* workspace, size_t workspace_bytes,
                               bd_stream_t stream) {
    Ctx c;
    BD_TRY(unet_ctx(u, c, B, training, workspace, workspace_bytes));
    BD_CHECK(params && x && t && out, BD_ERR_INVALID, "bd_unet_forward: null pointer");
    BD_CHECK(ldx >= u->cfg.in_channels && ldo >= u->cfg.out_channels, BD_ERR_INVALID, "bd_unet_forward: ld < channels");
    BD_CHECK(aligned16(params), BD_ERR_INVALID, "bd_unet_forward: params must be 16-byte aligned");
    c.params = params; c.x = x; c.ldx = ldx; c.t = t; c.t_stride = t_stride; c.out = out; c.ldo = ldo; c.st = S(stream);
    c.training = training != 0;
    if (training) { u->fwd_gen = bd::g_tune_gen; u->fwd_B = B; u->fwd_ws = workspace; }
    const bool prepared = u->static_weights && !training && u->prep_params == (const void*)params && u->prep_ws == workspace && u->prep_B == B;
    if (c.w_split && !prepared) {   // split the weights once; forward convs and (same workspace) the backward dgrads read the copy
        BD_TRY(bd_split_bf16(params, (int64_t)wsplit_elems(u), const_cast<uint16_t*>(c.w_split), stream));
        // pre-summed tap planes of the upsample convolutions that take the phase path (both forward pipelines read them: before
        // the fork); the transposed set E^T is the data gradient's operand: training only
        for (const auto& uw : u->upsw)
            if (u->phase_ok(c, uw.H, uw.W, uw.C, uw.C))
                BD_TRY(upsample_weights(params + uw.pw, uw.C, uw.C, bd_unet::U16(u->BP(c, uw.b_e)),
                                        training ? bd_unet::U16(u->BP(c, uw.b_et)) : nullptr, c.st));
        u->prep_params = u->static_weights && !training ? (const void*)params : nullptr; u->prep_ws = workspace; u->prep_B = B;
    }
    // transposed split planes of the 3x3 conv weights for the backward's data gradients: one launch, needed only in training
    auto transpose_weights = [&](hipStream_t st) -> int {
        if (!training || !c.wT_split || u->wt_off.empty()) return BD_OK;
        return split_wt_batched(params, const_cast<uint16_t*>(c.wT_split), u->wt_off.data(), u->wt_cin.data(), u->wt_cout.data(),
                                (int)u->wt_off.size(), st);
    };
    static const int fwd_pipes = getenv("BD_FWD_PIPES") ? atoi(getenv("BD_FWD_PIPES")) : 2;   // 1 = forward on one stream (A/B)
    // two pipelines need enough work per half to pay for twice the host enqueue: >= 32 K input pixels in the batch (B = 32 at 32 x 32 as
    // before; round 4: B = 4 at 256 x 256 qualifies too -- 29.05 -> 28.13 ms per 256 x 256 train step, small layers fill the chip in pairs)
    static const long long pipes_min_px = getenv("BD_FWD_PIPES_MINPX") ? atoll(getenv("BD_FWD_PIPES_MINPX")) : 32768;      // (A/B knob)
    if (!u->aux_enabled || B < 2 || (long long)B * u->cfg.sample_size * u->cfg.sample_size < pipes_min_px || fwd_pipes < 2) {
        BD_TRY(transpose_weights(c.st));
        for (auto& f : u->fwd) BD_TRY(f(c));
        return BD_OK;
    }
    // Samples are independent all the way through the network (GroupNorm is per sample), so the batch runs as two
    // half-batch pipelines on the caller's stream and the plan's side stream, node by node: the prologues / epilogues
    // / small kernels of one half overlap the main loops of the other (+4 % on the CIFAR forward).  Same buffers, same
    // layout as a single pass -- backward does not know the difference.
    BD_TRY(unet_aux_init(u));
    Ctx c2 = c;
    const int Bh = B / 2;
    const int64_t hw = (int64_t)u->cfg.sample_size * u->cfg.sample_size;
    c.B = Bh;
    c2.B = B - Bh; c2.s0 = Bh; c2.st = u->aux_stream; c2.opws = c.opws2;
    c2.x = x + (int64_t)Bh * hw * ldx; c2.t = t + (int64_t)Bh * t_stride; c2.out = out + (int64_t)Bh * hw * ldo;
    BD_HIP_TRY(hipEventRecord(u->aux_ev_fork, c.st));
    BD_HIP_TRY(hipStreamWaitEvent(c2.st, u->aux_ev_fork, 0));
    BD_TRY(transpose_weights(c2.st));   // memory-bound, off the first pipeline's critical path
    for (auto& f : u->fwd) {
        BD_TRY(f(c));
        BD_TRY(f(c2));
    }
    BD_HIP_TRY(hipEventRecord(u->aux_ev_join[0], c2.st));
    BD_HIP_TRY(hipStreamWaitEvent(c.st, u->aux_ev_join[0], 0));
    return BD_OK;
}

extern "C" int bd_unet_num_segments(const bd_unet* u) { return u ? (int)u->segs.size() : 0; }
extern "C" int bd_unet_segment_num_ranges(const bd_unet* u, int seg) {
    if (!u || seg < 0 || seg >= (int)u->segs.size()) return 0;
    return u->segs[seg].tw_lo >= 0 ? 3 : 1;
}
extern "C" int bd_unet_segment_range_k(const bd_unet* u, int seg, int k, int64_t* lo, int64_t* hi) {
    BD_CHECK(u && seg >= 0 && seg < (int)u->segs.size() && k >= 0 && k < bd_unet_segment_num_ranges(u, seg), BD_ERR_INVALID,
             "bd_unet_segment_range_k: segment / range out of range");
    const bd_unet::Seg& s = u->segs[seg];
    const int64_t l = k == 0 ? s.lo : (k == 1 ? s.tw_lo : s.tb_lo), h = k == 0 ? s.hi : (k == 1 ? s.tw_hi : s.tb_hi);
    if (lo) *lo = l;
    if (hi) *hi = h;
    return BD_OK;
}
extern "C" int bd_unet_segment_range(const bd_unet* u, int seg, int64_t* lo, int64_t* hi) {
    BD_CHECK(u && seg >= 0 && seg < (int)u->segs.size(), BD_ERR_INVALID, "bd_unet_segment_range: segment out of range");
    if (lo) *lo = u->segs[seg].lo;
    if (hi) *hi = u->segs[seg].hi;
    return BD_OK;
}

extern "C" int bd_unet_backward_segment(bd_unet* u, int seg, int B, const float* params, const float* x, int64_t ldx,
                                        const float* dout, int64_t lddo, float* grads, void* workspace, size_t workspace_bytes,
                                        bd_stream_t stream, int64_t* ready_lo, int64_t* ready_hi) {
    Ctx c;
    // a tuning-knob change re-lays out the workspace lazily: never under the saved activations of a forward that already ran
    BD_CHECK(!u || u->fwd_ws != workspace || u->fwd_B != B || u->fwd_gen == bd::g_tune_gen, BD_ERR_INVALID,
             "bd_unet_backward: bd_tune_set was called between the training forward and its backward (the saved activations would move)");
    BD_TRY(unet_ctx(u, c, B, 1, workspace, workspace_bytes));
    BD_CHECK(params && x && dout && grads, BD_ERR_INVALID, "bd_unet_backward: null pointer");
    BD_CHECK(aligned16(params) && aligned16(grads), BD_ERR_INVALID, "bd_unet_backward: params/grads must be 16-byte aligned");
    BD_CHECK(seg >= -1 && seg < (int)u->segs.size(), BD_ERR_INVALID, "bd_unet_backward: segment %d out of range", seg);
    c.params = params; c.grads = grads; c.dout = dout; c.lddo = lddo; c.st = S(stream);
    c.x = x; c.ldx = ldx;   // conv_in wgrad re-reads the forward input
    // Deferred join (bd_unet_set_deferred_join, used by the data-parallel trainer): the weight gradients a segment puts on the side
    // stream may still be running when the call returns and while the NEXT segment's data-gradient chain runs; the caller orders
    // whatever consumes this segment's gradient ranges behind them with bd_unet_stream_wait_aux(plan, consumer_stream).  The
    // scratch-arena hazard is covered across calls by carrying the per-parity "pending" marks in the plan.  The last segment (and
    // a whole backward, seg == -1) always joins, so the caller's stream ends ordered after everything.
    const bool defer = u->aux_enabled && u->aux_defer && seg >= 0 && seg + 1 < (int)u->segs.size();
    if (u->aux_enabled) {
        BD_TRY(unet_aux_init(u));
        c.st2 = u->aux_stream; c.ev_fork = u->aux_ev_fork; c.ev_join[0] = u->aux_ev_join[0]; c.ev_join[1] = u->aux_ev_join[1];
        if (seg > 0 && u->aux_defer) { c.pend[0] = u->aux_pend[0]; c.pend[1] = u->aux_pend[1]; }
        else if (u->aux_pend[0] || u->aux_pend[1]) {
            // a pass that starts (seg 0 / whole backward) while marks of an earlier, unfinished pass are still set: its weight gradients
            // may still be reading the scratch arenas -- order this pass behind everything that pass put on the side stream
            BD_HIP_TRY(hipStreamWaitEvent(c.st, u->aux_ev_seg, 0));
            u->aux_pend[0] = u->aux_pend[1] = false;
        }
    }
    // grad-init flags must reflect everything executed before this segment: replay them (host-only)
    for (auto it = u->bwd.rbegin(); it != u->bwd.rend(); ++it) {
        if (seg >= 0 && it->seg > seg) break;
        const bool run = seg < 0 || it->seg == seg;
        const bool keep = c.dry;
        c.dry = !run;           // earlier segments: flags only
        const int par = it->group & 1;
        if (run) BD_TRY(bd_unet::aux_wait(c, par));   // this node reuses the scratch arena of parity `par`
        int s = it->fn(c);
        c.dry = keep;
        if (s != BD_OK) return s;
        if (run) BD_TRY(bd_unet::aux_mark(c, par));
    }
    if (!c.gn_items.empty()) BD_TRY(bd_gn_bwd_params(c.gn_items.data(), (int)c.gn_items.size(), c.B, (bd_stream_t)c.st));
    {   // time_emb_proj weight / bias gradient rows of the resnets that just ran: the rows of a segment are adjacent in the batched
        // [sumC, T] projection, so one GEMM per segment (one for a whole backward) replaces one tiny launch per resnet
        int64_t lo = -1, hi = -1;
        for (int sg = 0; sg < (int)u->segs.size(); ++sg) {
            if (seg >= 0 && sg != seg) continue;
            const bd_unet::Seg& q = u->segs[sg];
            if (q.tb_lo < 0) continue;
            if (lo < 0 || q.tb_lo - u->p_tb < lo) lo = q.tb_lo - u->p_tb;
            if (q.tb_hi - u->p_tb > hi) hi = q.tb_hi - u->p_tb;
        }
        if (lo >= 0) {
            BD_TRY(u->linear_wgrad(c, u->BP(c, u->b_dtproj) + lo, u->sumC, u->BP(c, u->b_embs), u->T, c.grads + u->p_tw + lo * u->T, c.B,
                                   (int)(hi - lo), u->T, c.grads + u->p_tb + lo));
            BD_TRY(bd_unet::aux_mark(c, 0));
        }
    }
    if (defer) {
        u->aux_pend[0] = c.pend[0]; u->aux_pend[1] = c.pend[1];
        BD_HIP_TRY(hipEventRecord(u->aux_ev_seg, c.st2));   // everything this (and every earlier) call put on the side stream
    } else {
        BD_TRY(bd_unet::aux_wait(c, 0));   // every weight gradient of this call is ordered before what the caller enqueues next
        BD_TRY(bd_unet::aux_wait(c, 1));
        u->aux_pend[0] = u->aux_pend[1] = false;
        if (c.st2) BD_HIP_TRY(hipEventRecord(u->aux_ev_seg, c.st2));
    }
    if (seg >= 0) {
        if (ready_lo) *ready_lo = u->segs[seg].lo;
        if (ready_hi) *ready_hi = u->segs[seg].hi;
    }
    return BD_OK;
}

extern "C" int bd_unet_backward(bd_unet* u, int B, const float* params, const float* x, int64_t ldx, const float* dout,
                                int64_t lddo, float* grads, void* workspace, size_t workspace_bytes, bd_stream_t stream) {
    return bd_unet_backward_segment(u, -1, B, params, x, ldx, dout, lddo, grads, workspace, workspace_bytes, stream, nullptr,
                                    nullptr);
}
