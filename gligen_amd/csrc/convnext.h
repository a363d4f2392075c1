// Kernels of the spatial-map grounding tokenizers (reference ldm/modules/diffusionmodules/convnext.py,
// canny/hed/depth/normal/sem_grounding_net.py): ConvNeXt-tiny over the conditioning map, once per prompt.
// Activations are NHWC bf16 with a row stride `ld` >= C (the 96-channel stage is padded to 128 so its rows are legal GEMM K
// operands); the pointwise / patchify convolutions run on the bf16 MFMA GEMM (gemm.hip), these are the pieces around it.
#pragma once
#include "common.h"

namespace gl {

// F.interpolate(x, R) with the default mode 'nearest' of an fp32 NCHW image, then a k x k stride-k patchify into GEMM rows:
// out[(b, oy, ox)][(ky*k + kx)*Cin + c] = x[b][c][oy*k + ky][ox*k + kx] (bf16), columns >= k*k*Cin zero-filled up to Kpad.
int patchify_f32_launch(const float* x, bf16* out, int B, int Cin, int H, int W, int k, int Kpad, hipStream_t stream);
// the same for an NHWC bf16 tensor with row stride ld (only the first C channels are read): out [B*(H/k)*(W/k)][k*k*C]
int patchify_bf16_launch(const bf16* x, bf16* out, int B, int H, int W, int C, int ld, int k, hipStream_t stream);
// OIHW fp32 conv weight with kernel = stride = k -> [O][Kpad] bf16 in the patchify column order (ky, kx, c)
int pack_patch_weight_launch(const float* w, bf16* out, int O, int I, int k, int Kpad, hipStream_t stream);
// depthwise 7x7 convolution, padding 3, + bias (convnext.py:28,38): x, y NHWC bf16 [B][H][W][ld], w fp32 [C][49]
int dwconv7_launch(const bf16* x, const float* w, const float* bias, bf16* y, int B, int H, int W, int C, int ld, hipStream_t stream);
// 3x3 conv, padding 1, fp32 NCHW, small channel counts (sem_grounding_net.py:21: Conv2d(152, 3, 3, 1, 1))
int conv3x3_f32_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, hipStream_t stream);
// objs = feat * mask + null_feature * (1 - mask) + pos_embedding (canny_grounding_net.py:48-56): feat bf16 [B][T][ld], out bf16 [B][T][C]
int token_mix_launch(const bf16* feat, int ld, const float* mask, const float* null_feat, const float* pos, bf16* out, int B, int T, int C,
                     hipStream_t stream);
// w2[o][k] *= gamma[o], b2[o] *= gamma[o]: layer scale folded into pwconv2 (convnext.py:44-45), fp32 in place
int scale_rows_launch(float* w, float* b, const float* gamma, int O, int K, hipStream_t stream);

}  // namespace gl
