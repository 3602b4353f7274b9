/* Stand-alone probe, NOT part of libbd_hip.so (a drop-in ABI does not export experiments).
 * Build: python scripts/wino/build_probe.py  ->  scripts/wino/libbd_wino_probe.so (links baddiffusion_amd/libbd_hip.so for bd_split_* etc.). */
#pragma once
#include "../../include/bd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- Winograd F(2x2, 3x3) (round 5, PROTOTYPE): the stride-1 "same" 3x3 convolution (resnet.py:493,514; direction -1: its data gradient) with
 * 16 instead of 36 products per 2x2 output tile and channel pair.  Products stay split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate), the
 * input / output transforms are fp32 adds inside the kernel, the weight transform U = G g G^T is a separate call:
 *   bd_wino_weights(w [Cout][3][3][Cin] fp32, Cin, Cout, direction) -> u_planes: 64 * C * N bytes, [C/16][16][N][16 bf16 hi | 16 bf16 lo]
 *     (direction +1: N = Cout, C = Cin; -1: the rotated, transposed weights of the data gradient, N = Cin, C = Cout);
 *   bd_conv3x3_wino: x fp32 NHWC [B,H,W,C] (pixel stride ldx) -> y = out_scale * (conv + bias + rowbias[b] + residual), fp32 NHWC.
 * Shapes: W in {16, 32}, (H/2)*(W/2) % 64 == 0, C % 16 == 0, N % 64 == 0 (bd_conv3x3_wino_supported).  Error vs fp64: 8e-6 relative
 * (scripts/wino/numerics.py; the direct split-bf16 convolution: 1.2e-5 with its truncated hi planes). */
typedef struct {
    int B, H, W, C, N;
    const float* x; int64_t ldx;
    const uint16_t* u_planes;
    const float* bias;              /* [N] or NULL */
    const float* rowbias; int64_t ld_rowbias;   /* [B][N] or NULL */
    const float* residual; int64_t ldr;         /* [B*H*W][ldr] or NULL */
    float out_scale;                /* 0 is read as 1 */
    float* y; int64_t ldy;
} bd_conv3x3_wino_desc;
int bd_conv3x3_wino_supported(int B, int H, int W, int C, int N);
int bd_conv3x3_wino(const bd_conv3x3_wino_desc* d, bd_stream_t stream);
int bd_wino_weights(const float* w, int Cin, int Cout, int direction, uint16_t* u_planes, bd_stream_t stream);
#ifdef __cplusplus
}
#endif
