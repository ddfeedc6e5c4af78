/* w2e_irse.h -- C ABI of the kernels behind the ArcFace IR-SE50 identity loss (criteria/id_loss.py:7-40,
 * models/facial_recognition/model_irse.py:9-48, helpers.py:56-119) and the e4e encoder that shares its body
 * (models/encoders/psp_encoders.py:124-200) in libw2e.so (gfx950).
 *
 * The network runs in eval mode (id_loss.py:14): BatchNorm is a per-channel affine map (a = gamma/sqrt(var+eps),
 * b = beta - mean*a) that the host folds into the convolutions' scales / bias; PReLU rides in the conv epilogue.
 * Forward and INPUT gradients only (the network is a frozen critic).
 * Same conventions as w2e.h (device fp32 pointers, NCHW, caller-allocated outputs, stream as void*, 0 = OK).
 */
#ifndef W2E_IRSE_H
#define W2E_IRSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Plain 3x3 convolution on the fp32-MFMA engine of w2e_modconv3x3 (same packed weights, w2e_conv_pack):
 *   y[b,o] = prelu( out_scale[b,o] * conv(Wp, in_scale[b,i] * x[b,i]) + bias[o], slope[o] )
 * mode W2E_CONV_SAME: stride 1, zero padding 1 (nn.Conv2d(K, N, 3, 1, 1), helpers.py:112,114 with stride 1);
 * mode W2E_CONV_DOWN with down_pad = 1: stride 2, zero padding 1 on a [B,K,2h,2w] input -> [B,N,h,w]
 *   (nn.Conv2d(N, N, 3, 2, 1), helpers.py:114 at the first unit of a stage; with a centre-tap-only pack also the 1x1
 *   stride-2 shortcut convolution, helpers.py:103-106);
 * mode W2E_CONV_UP (no epilogue): the adjoint of DOWN -- T[B,N,2h+1,2w+1] phase-planar, of which [1:,1:] is the input
 *   gradient of the padded stride-2 convolution.
 * in_scale / out_scale ([B,K] / [B,N]) / bias / slope ([N]) may be NULL (1 / 1 / 0 / identity).  h,w and y_pitch as in
 * w2e_modconv3x3 (y_pitch: the row pitch the caller allocated the phase-planar UP output with; checked; ignored otherwise). */
int w2e_conv3x3(int mode, const float* x, const float* wp, const float* in_scale, const float* out_scale, float* y, int batch,
                int k_ch, int n_ch, int h, int w, int y_pitch, int down_pad, const float* bias, const float* slope, void* stream);

/* y = prelu(a[c]*x + b[c], slope[c]) over [B,C,HW]; a / b / slope may be NULL (1 / 0 / identity).
 * (eval-mode BatchNorm2d in front of a zero-padded convolution, helpers.py:111: the shift cannot be folded into
 * the convolution because the padding is applied after it.) */
int w2e_affine_act_fwd(const float* x, const float* a, const float* b, const float* slope, float* y, int batch, int channels,
                       int64_t hw, void* stream);
/* gx = a[c] * gy * (y > 0 ? 1 : slope[c])   (y = the forward OUTPUT; valid for slope > 0, where sign(y) = sign(pre)).
 * planar != 0: gy is the phase-planar T of W2E_CONV_UP ([B,C,2,2,h/2+1,WP]), `planar` = the row pitch WP the caller allocated it
 * with (must equal W2E_PLANAR_PITCH(width/2): a mismatch is refused instead of read past), and the element read for (yy,xx) is
 * T[yy+1][xx+1] -- the crop that turns the adjoint of DOWN into the gradient of the padded stride-2 convolution;
 * then height x width (even) describe gx and hw = height*width. */
int w2e_affine_act_bwd(const float* gy, const float* y, const float* a, const float* slope, float* gx, int batch, int channels,
                       int height, int width, int planar, void* stream);

/* sums[b,c] = sum_p x[b,c,p] * (y ? y[b,c,p] : 1): the SE block's global average pool (helpers.py:58,66) and, in the
 * backward, the gradient reaching the gate.  One wave per (b,c) plane, fixed reduction order. */
int w2e_channel_sums(const float* x, const float* y, float* sums, int batch, int channels, int64_t hw, void* stream);

/* The SE gate on the pooled sums (helpers.py:56-72): gate[b,c] = sigmoid(fc2 . relu(fc1 . (sums[b,:] * inv_hw))), with
 * fc1 [R,C] and fc2 [C,R] the two bias-free 1x1 convolutions (R = C / reduction); hidden [B,R] receives the ReLU output
 * (kept for the backward).  One workgroup per sample, fixed reduction order. */
int w2e_se_gate_fwd(const float* sums, const float* fc1, const float* fc2, float* gate, float* hidden, int batch, int channels,
                    int reduced, float inv_hw, void* stream);
/* Its adjoint: dgate[b,c] (= w2e_channel_sums(gout, t)) -> gpool[b,c], the gradient at the pooled mean times inv_hw. */
int w2e_se_gate_bwd(const float* dgate, const float* gate, const float* hidden, const float* fc1, const float* fc2, float* gpool,
                    int batch, int channels, int reduced, float inv_hw, void* stream);

/* out[b,c,p] = t[b,c,p]*gate[b,c] + shortcut  (helpers.py:72 + :118-119 `res + shortcut`).
 * sc_stride = 0: shortcut is a [B,C,H,W] tensor; sc_stride = s >= 1: shortcut is x[b,c,s*y,s*x] of a [B,C,s*H,s*W]
 * tensor (MaxPool2d(1, s), helpers.py:100-101). */
int w2e_se_apply_fwd(const float* t, const float* gate, const float* shortcut, int sc_stride, float* out, int batch, int channels,
                     int height, int width, void* stream);
/* g_t = gout*gate[b,c] + gpool[b,c]  (gpool = the gradient that reached the pooled mean, already divided by H*W). */
int w2e_se_apply_bwd(const float* gout, const float* gate, const float* gpool, float* g_t, int batch, int channels, int64_t hw,
                     void* stream);
/* gx[b,c,s*y,s*x] += g[b,c,y,x]  (adjoint of the strided shortcut; s = 1: a plain in-place add).
 * planar != 0: g is a phase-planar UP output T (`planar` = its row pitch, checked as in w2e_affine_act_bwd) and
 * gx[b,c,y,x] += T[y+1][x+1]. */
int w2e_shortcut_add_bwd(float* gx, const float* g, int batch, int channels, int height, int width, int stride, int planar,
                         void* stream);

/* FPN merge of the pSp / e4e encoders (models/encoders/helpers.py:123-140): out [planes,oh,ow] = bilinear up-sampling of
 * x [planes,ih,iw] (align_corners = True) + y [planes,oh,ow].  Forward only (the encoders are inference networks here). */
int w2e_upsample_add(const float* x, const float* y, float* out, int64_t planes, int ih, int iw, int oh, int ow, void* stream);

#ifdef __cplusplus
}
#endif
#endif
