// 3x3 convolution entry points: thin descriptors over the implicit-GEMM engine (igemm.hip).
//   fwd   : A = CONV gather of x (KC)            B = W[Cout][9*Cin] dense (KC)
//   dgrad : A = TCONV gather of dy (KC)          B = W seen as [Cin rows][k=(tap,co)] (WGT, RC)
//   wgrad : A = dy^T [Cout rows][k=pixels] (RC)  B = CONV gather of x, rows=(tap,ci), k=pixels (RC)
// Replaces aten::convolution / convolution_backward for nn.Conv2d(k=3) at resnet.py:493,514 (ResnetBlock2D),
// :118 (Upsample2D, nearest x2 folded into the gather), :185,201-203 (Downsample2D, stride 2 with the
// asymmetric (0,1,0,1) zero pad folded into the bounds check), unet_2d.py:124,217.
#include "common.h"

namespace bd {

static int conv_geom_check(const char* who, int B, int Hs, int Ws, int Cin, int Cout, int stride, int ups, int Ho, int Wo) {
    BD_CHECK(B > 0 && Hs > 0 && Ws > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, BD_ERR_INVALID, "%s: bad shape", who);
    BD_CHECK(stride == 1 || stride == 2, BD_ERR_UNSUPPORTED, "%s: stride %d", who, stride);
    BD_CHECK(ups == 0 || ups == 1, BD_ERR_UNSUPPORTED, "%s: ups %d", who, ups);
    BD_CHECK(!(ups && stride != 1), BD_ERR_UNSUPPORTED, "%s: upsample with stride", who);
    BD_CHECK((long long)B * Ho * Wo < (1ll << 31) && (long long)B * (Hs << ups) * (Ws << ups) < (1ll << 31), BD_ERR_UNSUPPORTED,
             "%s: pixel count overflows int32", who);
    return BD_OK;
}

int conv3x3_fwd_thin(const bd_conv3x3_fwd_desc& d, hipStream_t st);       // conv_thin.hip
int conv3x3_dgrad_thin(const bd_conv3x3_dgrad_desc& d, hipStream_t st);

int conv3x3_fwd(const bd_conv3x3_fwd_desc& d, hipStream_t st) {
    BD_TRY(conv_geom_check("conv3x3_fwd", d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.stride, d.ups, d.Ho, d.Wo));
    BD_CHECK(d.x && d.w && d.y, BD_ERR_INVALID, "conv3x3_fwd: null pointer");
    {   // 3-channel conv_in / conv_out: direct fp32 streaming kernels (unet_2d.py:124,217)
        const int thin = conv3x3_fwd_thin(d, st);
        if (thin != 0) return thin < 0 ? thin : (int)BD_OK;
    }
    bd_igemm_desc g = {};
    g.A.kind = BD_OPK_CONV; g.A.kc = 1; g.A.p = d.x; g.A.ld = d.ldx;
    g.A.C = d.Cin; g.A.Hs = d.Hs; g.A.Ws = d.Ws; g.A.Ho = d.Ho; g.A.Wo = d.Wo;
    g.A.stride = d.stride; g.A.pad_t = d.pad_t; g.A.pad_l = d.pad_l; g.A.ups = d.ups;
    g.B.kind = BD_OPK_DENSE; g.B.kc = 1; g.B.p = d.w; g.B.ld = 9ll * d.Cin; g.B.C = d.Cin; g.B.split = d.w_split;
    g.M = d.B * d.Ho * d.Wo; g.N = d.Cout; g.K = 9 * d.Cin;
    g.batch_outer = g.batch_inner = 1;
    g.C = d.y; g.ldc = d.ldy;
    g.alpha = 1.f; g.out_scale = d.out_scale == 0.f ? 1.f : d.out_scale;
    g.bias = d.bias;
    g.rowbias = d.rowbias; g.ld_rowbias = d.ld_rowbias; g.rows_per_group = d.Ho * d.Wo;
    g.residual = d.residual; g.ldr = d.ldr;
    g.workspace = d.workspace; g.workspace_bytes = d.workspace_bytes; g.mode = d.mode;
    return igemm_launch(g, st);
}

int conv3x3_dgrad(const bd_conv3x3_dgrad_desc& d, hipStream_t st) {
    BD_TRY(conv_geom_check("conv3x3_dgrad", d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.stride, d.ups, d.Ho, d.Wo));
    BD_CHECK(d.dy && d.w && d.dx, BD_ERR_INVALID, "conv3x3_dgrad: null pointer");
    {
        const int thin = conv3x3_dgrad_thin(d, st);
        if (thin != 0) return thin < 0 ? thin : (int)BD_OK;
    }
    const int Hi = d.Hs << d.ups, Wi = d.Ws << d.ups;  // the conv's own (virtual) input grid
    bd_igemm_desc g = {};
    g.A.kind = BD_OPK_TCONV; g.A.kc = 1; g.A.p = d.dy; g.A.ld = d.lddy;
    g.A.C = d.Cout; g.A.Hs = d.Ho; g.A.Ws = d.Wo; g.A.Ho = Hi; g.A.Wo = Wi;
    g.A.stride = d.stride; g.A.pad_t = d.pad_t; g.A.pad_l = d.pad_l; g.A.ups = 0;
    g.B.kind = BD_OPK_WGT; g.B.kc = 0; g.B.p = d.w; g.B.ld = d.Cin; g.B.C = d.Cout; g.B.split = d.w_split;
    g.M = d.B * Hi * Wi; g.N = d.Cin; g.K = 9 * d.Cout;
    g.batch_outer = g.batch_inner = 1;
    g.C = d.dx; g.ldc = d.lddx;
    g.alpha = 1.f; g.out_scale = 1.f; g.accumulate = d.accumulate;
    g.workspace = d.workspace; g.workspace_bytes = d.workspace_bytes; g.mode = d.mode;
    return igemm_launch(g, st);
}

int conv3x3_wgrad_thin(const bd_conv3x3_wgrad_desc& d, hipStream_t st);   // conv_thin.hip

int conv3x3_wgrad(const bd_conv3x3_wgrad_desc& d, hipStream_t st) {
    BD_TRY(conv_geom_check("conv3x3_wgrad", d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.stride, d.ups, d.Ho, d.Wo));
    BD_CHECK(d.x && d.dy && d.dw, BD_ERR_INVALID, "conv3x3_wgrad: null pointer");
    {   // 3-channel conv_in / conv_out: direct streaming kernels instead of a 90 %-padded MFMA tile
        const int thin = conv3x3_wgrad_thin(d, st);
        if (thin != 0) return thin < 0 ? thin : (int)BD_OK;
    }
    bd_igemm_desc g = {};
    g.A.kind = BD_OPK_DENSE; g.A.kc = 0; g.A.p = d.dy; g.A.ld = d.lddy;
    g.B.kind = BD_OPK_CONV; g.B.kc = 0; g.B.p = d.x; g.B.ld = d.ldx;
    g.B.C = d.Cin; g.B.Hs = d.Hs; g.B.Ws = d.Ws; g.B.Ho = d.Ho; g.B.Wo = d.Wo;
    g.B.stride = d.stride; g.B.pad_t = d.pad_t; g.B.pad_l = d.pad_l; g.B.ups = d.ups;
    g.M = d.Cout; g.N = 9 * d.Cin; g.K = d.B * d.Ho * d.Wo;
    g.batch_outer = g.batch_inner = 1;
    g.C = d.dw; g.ldc = 9ll * d.Cin;
    g.alpha = 1.f; g.out_scale = 1.f;
    g.workspace = d.workspace; g.workspace_bytes = d.workspace_bytes; g.mode = d.mode;
    g.a_colsum = d.db;
    return igemm_launch(g, st);
}

size_t conv3x3_workspace_bytes(int B, int Ho, int Wo, int Hs, int Ws, int Cin, int Cout, int ups) {
    // upper bound over fwd / dgrad / wgrad split-K slabs: ksplit <= 128 only when tiles < 256, so
    // ksplit*tiles <= 512 + 256 tiles of at most 128x128 floats.
    size_t a = (size_t)768 * 128 * 128 * sizeof(float) + ((size_t)1 << 20);   // + fused bias-gradient partial rows
    (void)B; (void)Ho; (void)Wo; (void)Hs; (void)Ws; (void)Cin; (void)Cout; (void)ups;
    return a;
}

}  // namespace bd

extern "C" int bd_conv3x3_fwd(const bd_conv3x3_fwd_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_fwd: null descriptor");
    return bd::conv3x3_fwd(*d, bd::S(s));
}
extern "C" int bd_conv3x3_dgrad(const bd_conv3x3_dgrad_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_dgrad: null descriptor");
    return bd::conv3x3_dgrad(*d, bd::S(s));
}
extern "C" int bd_conv3x3_wgrad(const bd_conv3x3_wgrad_desc* d, bd_stream_t s) {
    BD_CHECK(d, BD_ERR_INVALID, "bd_conv3x3_wgrad: null descriptor");
    return bd::conv3x3_wgrad(*d, bd::S(s));
}
extern "C" size_t bd_conv3x3_workspace_bytes(int B, int Ho, int Wo, int Hs, int Ws, int Cin, int Cout, int ups) {
    return bd::conv3x3_workspace_bytes(B, Ho, Wo, Hs, Ws, Cin, Cout, ups);
}
