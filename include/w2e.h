/* w2e.h -- C ABI of libw2e.so: the MI355X (gfx950) kernels for the Where2edit
 * latent-editing hot path.
 *
 * The reference (Big-Brother-Pikachu/Where2edit) has no native code and no FFI:
 * its "operator seam" is the Python export list of models/stylegan2/op/__init__.py:1-2
 * plus the torch op sequences inside models/stylegan2/model.py, attention/attention_model.py
 * and criteria/clip_loss.py.  Each entry point below names the reference lines whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain C, no torch types; every pointer is a DEVICE pointer to contiguous fp32
 *     (NCHW for images) unless stated; the CALLER allocates outputs and workspaces;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     never synchronises and never allocates device memory (safe under hipGraph capture
 *     once each kernel has been launched once on the device: the first launch of a kernel
 *     that needs > 64 KB of LDS calls hipFuncSetAttribute for that device);
 *   - all pointers of one call must belong to the CURRENT HIP device (hipSetDevice), which
 *     must also own `stream`;
 *   - returns 0 on success, non-zero on error; w2e_last_error() then describes it
 *     (thread-local); nothing throws across the ABI;
 *   - global mutable state: the error string (thread-local), the process-wide options below
 *     (read from the environment ONCE when the library is loaded, then only changed by
 *     w2e_set_option; no getenv on any launch path), and one "large LDS enabled" bit per
 *     (kernel, device).  Work-skipping / clock-stamping tuning aids (tune_skip, tune_clock)
 *     exist only in a library compiled with -DW2E_TUNING; the shipped build ignores them.
 */
#ifndef W2E_H
#define W2E_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5: additive over 4 -- w2e_pack_kq_h / w2e_gemm_pk_h / w2e_gemm_pk_h_splits (w2e_vit.h), the options tune_xcd / tune_mw; no existing
 * signature changed */
#define W2E_VERSION 5

int w2e_version(void);
const char* w2e_last_error(void);

/* Process-wide options.  Names (environment variable read at load in brackets):
 *   "conv_precision" [W2E_CONV_PRECISION]  "f32" (default: exact fp32 MFMA) | "bf16x3" (opt-in)
 *   "deterministic"  [W2E_DETERMINISTIC]   "1": bit-reproducible results -- no fp32 atomics anywhere (no split-K,
 *                                          ordered reductions), the counterpart of the reference's
 *                                          cudnn.deterministic=True (attention/run_attention.py:903-904)
 *   "tune_cfg" [W2E_TUNE_CFG] "<cfg>[,<splits>[,<mode>]]" force a conv tile (tests, tools/layer_bench.py); "" = off
 *   "tune_upall", "tune_dma", "tune_fuse", "tune_print", "tune_blur", "tune_gemm_s": kernel-selection aids -- they
 *   choose between kernels / tiles that compute the same result ("tune_blur": only bit 8, keep the LDS-tile FIR kernels
 *   for wide images; its bits 1/2/4 and "tune_skip" / "tune_clock" drop loads, arithmetic or stores, or synchronise,
 *   and are compiled in ONLY by -DW2E_TUNING: the shipped library ignores them)
 *   "debug_poison" [W2E_DEBUG_POISON] "1": host-side aid -- gradient rows the merged forward declares unused are
 *   filled with NaN instead of being left unwritten, so that a consumer that reads them fails loudly (tests)
 * w2e_get_option reads "conv_precision", "deterministic", "tune_cfg", "tuning_build" (1 = compiled with -DW2E_TUNING). */
int w2e_set_option(const char* name, const char* value);

/* Row pitch (floats) of the phase planes of the transposed-conv output T for an input W wide: W+1 columns rounded up to 16 floats =
 * one 64-byte memory sector, so that the 128-byte segments the conv's waves store (32 consecutive columns of a plane row) cover
 * whole sectors and the FIR kernels' rows start on one.  (Round 2 padded to 4 floats: every segment then straddled three
 * sectors and the T write of the 64->32 @ 512 layer ran at 2.2 TB/s, against 4.9 for the same bytes written sector-aligned.) */
#define W2E_PLANAR_PITCH(w) ((((w) + 1) + 15) & ~15)
int w2e_get_option(const char* name, int* value);

/* ---- K2  upfirdn2d  (models/stylegan2/op/upfirdn2d.py:11-60) ---------------------------
 * y[p,oy,ox] = sum_{ky,kx} kern'[ky,kx] * xz[p, oy*down + ky - pad_y0, ox*down + kx - pad_x0]
 * where xz is x zero-stuffed by `up` (sample at multiples of up) and zero outside, and
 * kern' = kern flipped in both axes when flip != 0 (flip=1 is the reference forward: a true
 * convolution; flip=0 with swapped up/down is its adjoint, used for backward).
 * planes = N*C.  kern is [kh,kw] on the device, kh,kw <= 16.
 * Optional fused epilogue (StyledConv after the blur, model.py:260,290 + op/fused_act.py:23-39):
 *   y = lrelu(out_scale[p]*y + noise_w[0]*noise[oy,ox] + bias[p % channels], slope) * gain
 * enabled when act != 0; out_scale / noise / bias may be NULL (treated as 1 / 0 / 0).
 * in_layout: 0 = x is [planes, in_h, in_w]; 1 = x is the phase-planar image W2E_CONV_UP writes,
 * [planes, 2, 2, (in_h+1)/2, WP] with WP = W2E_PLANAR_PITCH((in_w-1)/2) (sector-aligned rows) and
 * x[Y][X] = x'[Y&1][X&1][Y>>1][X>>1]  (4x4 kernel, up=down=1 only).  in_pitch: the row pitch (floats) the CALLER allocated that
 * planar image with -- the caller owns the buffer, the kernels index it with W2E_PLANAR_PITCH, so a caller compiled against another
 * pitch (ABI 2 padded to 4 floats) would hand over a smaller buffer than the kernels read: the call is refused with an error instead
 * (round 3: a memory access fault in a tool still on the old pitch).  Ignored for in_layout = 0. */
int w2e_upfirdn2d(const float* x, const float* kern, float* y, int64_t planes, int in_h, int in_w, int out_h,
                  int out_w, int kh, int kw, int up, int down, int pad_x0, int pad_y0, int flip, int in_layout, int in_pitch,
                  int act, const float* out_scale, const float* noise, const float* noise_w, const float* bias,
                  int channels, float slope, float gain, void* stream);
/* The StyledConv backward of an up-sampling layer in one pass (op/fused_act.py:42-60 + the adjoint of Blur(pad=(1,1)),
 * model.py:253-259): gt [planes, h+1, w+1] = adjoint 4x4 FIR (pad 2, un-flipped taps) of gpre = gy * gain * (y_fwd > 0 ? 1 : slope),
 * and sums [planes,3] = (sum gpre * pre-activation, sum gpre * noise, sum gpre) as w2e_bias_act_bwd_reduce returns them; gpre is
 * never written.  gy, y_fwd [planes, h, w]; noise [h*w] or NULL; w >= 256 and a multiple of 4; not in deterministic mode. */
int w2e_blur_adjoint_actbwd(const float* gy, const float* y_fwd, const float* noise, const float* kern, float* gt, float* sums,
                            int64_t planes, int h, int w, float slope, float gain, void* stream);

/* ---- K3  bias (+noise) + leaky-relu * gain  (op/fused_act.py:23-39, model.py:285-290) -----
 * x viewed as [outer, channels, inner]; bias[channels] or NULL; noise[inner] or NULL with device
 * scalar noise_w.  y = lrelu(x + noise_w*noise[i] + bias[c], slope) * gain. */
int w2e_bias_act_fwd(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                     int64_t outer, int64_t channels, int64_t inner, float slope, float gain, void* stream);
/* gx = gy * gain * (y > 0 ? 1 : slope)   (sign of the pre-activation == sign of y). */
int w2e_bias_act_bwd(const float* gy, const float* y, float* gx, int64_t n, float slope, float gain, void* stream);
/* Same, plus the per-(outer,channel) reductions the demodulation / bias / noise-strength
 * gradients need (one pass over gy,y):  sums[(o*channels+c)*3 + {0,1,2}] =
 *   sum_i gx*pre, sum_i gx*noise[i], sum_i gx   with pre = y>0 ? y/gain : y/(gain*slope).
 * sums must hold outer*channels*3 floats (written, not accumulated). */
int w2e_bias_act_bwd_reduce(const float* gy, const float* y, const float* noise, float* gx, float* sums,
                            int64_t outer, int64_t channels, int64_t inner, float slope, float gain, void* stream);

/* ---- K1  modulated 3x3 convolution on fp32 MFMA  (model.py:234-276) -----------------------
 * Shared-weight form of ModulatedConv2d:  the reference materialises w'[b,o,i,k] =
 * scale*W[o,i,k]*s[b,i]*demod[b,o] per sample; here
 *     y[b,o] = out_scale[b,o] * conv(Wp, in_scale[b,i] * x[b,i])
 * with one packed weight tensor shared by the batch.  Forward: in_scale = s, out_scale = demod.
 * Input-gradient: the same kernel on the flipped/transposed pack with in_scale = demod, out_scale = s.
 *
 * w2e_conv_pack: weight [cout,cin,3,3] -> wp [ceil(K/8)][9][2][N][4] floats, multiplied by `scale`:
 *   wp[kc][tap][h][n][c] = scale * W(k = 8*kc + 2*c + h, tap, n), zero for k >= K  -- the float4 of (kc, tap, h, n) is what
 *   lane-half h of the MFMA wave needs for the 4 channel pairs of chunk kc, so the kernel stages and reads it as one 16-B word;
 *   transpose=0: K=cin, N=cout (forward);  transpose=1: K=cout, N=cin (input gradient);
 *   flip=1 reverses the 9 taps (stride-1 input gradient); flip=0 keeps them.
 *   wp must hold ceil(K/8)*288*N floats. */
int w2e_conv_pack(const float* weight, float* wp, int cout, int cin, float scale, int transpose, int flip,
                  void* stream);

#define W2E_CONV_SAME 0 /* stride 1, zero pad 1: [B,K,H,W] -> [B,N,H,W]              (model.py:270-274) */
#define W2E_CONV_UP 1   /* conv_transpose stride 2, pad 0: T[B,N,2H+1,2W+1] (model.py:249-259), stored
                         * phase-planar as [B,N,2,2,H+1,WP], WP = W2E_PLANAR_PITCH(W): T[Y][X] = y[Y&1][X&1][Y>>1][X>>1] so that every
                         * output phase is written with unit-stride, aligned rows (see w2e_upfirdn2d in_layout=1) */
#define W2E_CONV_DOWN 2 /* stride 2, pad 0 on [B,K,2H+1,2W+1] -> [B,N,H,W] (adjoint of UP) */

/* Epilogue (applied after out_scale), all optional:
 *   act != 0:   y = lrelu(y + noise_w[0]*noise[h,w] + bias[o], 0.2) * sqrt(2)   (SAME only)
 *   dot_with/dot_out: dot_out[b,o] += sum_p conv_unscaled[b,o,p] * dot_with[b,o,p]
 *                     (the direct style gradient sum_p x*g; dot_out must be zeroed by the caller).
 * h,w are the INPUT spatial size for SAME/UP and the OUTPUT size for DOWN.  y_pitch: for W2E_CONV_UP the row pitch (floats) the
 * caller allocated the phase-planar y with; anything but W2E_PLANAR_PITCH(w) is refused (see w2e_upfirdn2d's in_pitch); ignored
 * for SAME / DOWN. */
int w2e_modconv3x3(int mode, const float* x, const float* wp, const float* in_scale, const float* out_scale,
                   float* y, int batch, int k_ch, int n_ch, int h, int w, int y_pitch, int act, const float* noise,
                   const float* noise_w, const float* bias, const float* dot_with, float* dot_out, void* stream);

/* ---- K1w  the FUSED Winograd F(4x4,3x3) form of W2E_CONV_SAME  (model.py:270-274; the same contract as w2e_modconv3x3) -------------
 *     y[b,o] = epilogue(out_scale[b,o] * A^T [ sum_i U[.,o,i] (.) V[.,i,t] ] A)
 * 36 products per 4x4 outputs instead of 144 (interpolation points 0, +-1, +-2, inf); 36 transform-domain positions, T = batch *
 * H/4 * W/4 tiles.  For the narrow, high-resolution layers (N % 32 == 0 output channels, K a power of two in 32 ... 256, H % 16 == 0,
 * W % 32 == 0, x smaller than 4 GB): input transform, the 36 contractions (fp32 MFMA) and the output transform in ONE persistent
 * kernel -- V and M, 2.25x the input / output each, never leave the CU (through HBM they cost more than the 4x fewer FLOPs return at
 * these sizes).  uf [36][K/8][2][N][4] = G W G^T in MFMA operand order, from the packed direct-form weights (w2e_conv_pack: any
 * transpose / flip), once per pack.  Epilogues: act = 1 (noise_w*noise + bias, LeakyReLU 0.2, sqrt 2), act = 2 (+ bias[o],
 * PReLU(slope[o]) if slope: w2e_conv3x3's), or dot_with / dot_out -- NOT accumulated: dot_out receives one partial per (channel,
 * spatial block), [batch][N][H/16 * W/32] floats (written, no atomics: deterministic), which the caller sums over the last axis
 * (w2e_channel_sums).  wgs > 0 caps the persistent grid (tests).  Rounding, max-norm relative to a float64 convolution at
 * K = 128 ... 512: direct 3e-7, this form 1e-5 (its transforms multiply by up to 8 and 1/24) -- against the path's 1e-3 tolerance. */
int w2e_wino_weights_fused(const float* wp, float* uf, int k_ch, int n_ch, void* stream);
int w2e_wino_fused(const float* x, const float* in_scale, const float* uf, const float* out_scale, float* y, int batch, int k_ch,
                   int n_ch, int h, int w, int act, const float* noise, const float* noise_w, const float* bias, const float* slope,
                   const float* dot_with, float* dot_out, int wgs, void* stream);

/* ---- K1g  the F(4x4,3x3) form with the contraction on an OWN fp32-MFMA kernel and the output transform in its epilogue
 * (model.py:270-274; models/facial_recognition/helpers.py:97-119) -- the wide same-resolution layers (N % 64 == 0, K % 8 == 0;
 * H, W multiples of 4 -- or not, with the plain and the bias + PReLU epilogues only: tiles then = ceil(H/4) * ceil(W/4) and the
 * outputs of the last tile row / column past the image are not stored: IR-SE50's 14^2 / 7^2 stages).  Replaces the host-side composition w2e_wino_input -> vendor strided-batched GEMM -> w2e_wino_output:
 *   w2e_wino_gemm_plan   for one layer call: tiles_padded (T = batch*ceil(H/4)*ceil(W/4) rounded up to 32), the K split the library would use
 *                        (1 = none; small layers split K so that 256 CUs have work) and the floats of `workspace` w2e_wino_gemm needs
 *                        (fused-dot partials + split-K slabs; may be 0).  vf must hold 36 * K * tiles_padded floats.
 *   w2e_wino_pack_input  x [B,K,H,W], in_scale [B,K] or NULL -> vf [36][K/8][2][tiles_padded][4]: B^T (in_scale * window) B in the
 *                        order the MFMA consumes it: (xi, kc, h, t, c) = V[xi][k = 8*kc + 2*c + h][t], t = (b, tile row, tile column)
 *   w2e_wino_gemm        y = epilogue(out_scale * A^T [sum_k U V] A) from uf (w2e_wino_weights_fused) and vf.  One workgroup = 64
 *                        channels x 32 tiles x all 36 positions; operands global/L2 -> registers (no LDS in the K loop); M never
 *                        reaches HBM.  Epilogues as w2e_wino_fused: act 1 (noise_w*noise + bias, LeakyReLU 0.2, sqrt 2), act 2
 *                        (+ bias, PReLU(slope) if slope), or dot_with / dot_out: dot_out[b,o] += sum_p conv_unscaled * dot_with,
 *                        WITHOUT atomics (per-segment partials in `workspace`, summed in a fixed order by a second launch; needs
 *                        (H/4)*(W/4) tiles per plane a multiple of 32 or a power of two below it).  splits > 1: each K split writes
 *                        its raw A^T M A into a slab of `workspace`, a second launch sums the slabs in order and applies the epilogue.
 *                        No fp32 atomics on any path: results are bit-reproducible ("deterministic" hosts may use it).
 * Rounding as the other F(4x4,3x3) forms: ~1e-5 max-norm relative to a float64 convolution at K = 128 ... 512. */
int w2e_wino_gemm_plan(int batch, int k_ch, int n_ch, int h, int w, int* tiles_padded, int* splits, int64_t* workspace_floats);
int w2e_wino_pack_input(const float* x, const float* in_scale, float* vf, int batch, int k_ch, int h, int w, int tiles_padded, void* stream);
int w2e_wino_gemm(const float* uf, const float* vf, const float* out_scale, float* y, int batch, int k_ch, int n_ch, int h, int w,
                  int tiles_padded, int splits, float* workspace, int act, const float* noise, const float* noise_w, const float* bias,
                  const float* slope, const float* dot_with, float* dot_out, void* stream);

/* Demodulation coefficients and their style gradient (model.py:241-243), [B,C]-sized:
 *   d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + eps),  wsq[o,i] = sum_k (scale*W[o,i,k])^2  [cout,cin]. */
int w2e_demod_fwd(const float* s, const float* wsq, float* d, int batch, int cin, int cout, float eps, void* stream);
/* The same for every demodulated layer of one generator pass in ONE launch (HOST array of descriptors, copied into the
 * launch): layers[j].d[b,o] = rsqrt(sum_i s_j[b,i]^2 * wsq_j[o,i] + eps), s_j [B,cin_j], wsq_j [cout_j,cin_j], d_j [B,cout_j]. */
#define W2E_DEMOD_MAX_LAYERS 32
typedef struct {
    const float* s;
    const float* wsq;
    float* d;
    int cin, cout;
} w2e_demod_layer;
int w2e_demod_all_fwd(const w2e_demod_layer* layers, int n_layers, int batch, float eps, void* stream);
/* gs[b,i] -= s[b,i] * sum_o dz[b,o]*d[b,o]^2*wsq[o,i]  (gs holds the direct part sum_p x*g on entry), where
 * dz[b,o] = sum_p gpre*(d*z) is either given (`dz`) or rebuilt from w2e_bias_act_bwd_reduce's `sums` as
 * s1 - noise_w*s2 - bias[o]*s3.  Exactly one of sums / dz is non-NULL.  gd (optional) receives dz/d. */
int w2e_demod_bwd(const float* sums, const float* dz, const float* noise_w, const float* bias, const float* d,
                  const float* s, const float* wsq, float* gs, float* gd, int batch, int cin, int cout, void* stream);

/* All style modulations of one generator pass in one launch (model.py:211, `style = self.modulation(style)` in each of
 * the 26 ModulatedConv2d): every layer's EqualLinear(style_dim, cin_l) -- weight*scale and bias*lr_mul, model.py:151-158
 * -- stacked row-wise into w [rows, dim] and bias [rows] (NULL = none).  meta [rows][4] (int32) per stacked row =
 * (W+ index of the layer, rows before the layer, cin_l, row inside the layer); every cin_l is a multiple of 32.
 *   out[rows_before_l*batch + b*cin_l + c] = latent[b, widx_l, :] . w[row, :] + bias[row]     (layer blocks [batch, cin_l])
 * bwd: glatent [batch, n_latent, dim] = the adjoint w.r.t. latent (zeroed inside; layers sharing a W+ index add up). */
int w2e_style_affine_fwd(const float* latent, const float* w, const float* bias, const int* meta, float* out, int batch,
                         int n_latent, int dim, int rows, void* stream);
int w2e_style_affine_bwd(const float* gout, const float* w, const int* meta, float* glatent, int batch, int n_latent,
                         int dim, int rows, void* stream);

/* ---- K1r  ToRGB: modulated 1x1 conv (no demod) + bias + upsampled skip  (model.py:343-362) --
 * wmod [B,3,cin] = scale*W[c,i]*s[b,i] (tiny, built by the host);  skip [B,3,h/2,w/2] or NULL is
 * up-sampled x2 with the 4x4 kernel `upk` (Upsample, model.py:31-49: up=2, pad (2,1)) and added. */
int w2e_torgb_fwd(const float* x, const float* wmod, const float* bias, const float* skip, const float* upk,
                  float* y, int batch, int cin, int h, int w, void* stream);
/* gx[b,i,p] = sum_c wmod[b,c,i]*gy[b,c,p];  gwmod[b,c,i] = sum_p x[b,i,p]*gy[b,c,p] (written). */
int w2e_torgb_bwd(const float* x, const float* wmod, const float* gy, float* gx, float* gwmod, int batch, int cin,
                  int h, int w, void* stream);
/* The same with gx = gx_acc + (that sum): x (a layer's activation, model.py:542-546) feeds both its ToRGB and the next
 * conv; passing the gradient that came back through the conv as gx_acc folds autograd's accumulation of the two into
 * this kernel.  gx_acc NULL = w2e_torgb_bwd. */
int w2e_torgb_bwd_acc(const float* x, const float* wmod, const float* gy, const float* gx_acc, float* gx, float* gwmod,
                      int batch, int cin, int h, int w, void* stream);
/* The same pair with the modulation formed in the kernel (model.py:239 with k = 1, demodulate=False): wsc [3,cin] =
 * scale*W shared by the batch, style [B,cin]; the weight of sample b is wsc[c,i]*style[b,i].  The backward writes the
 * STYLE gradient gstyle[b,i] = sum_c wsc[c,i] * sum_p x[b,i,p]*gy[b,c,p] ([B,cin]; W is frozen on this path) and
 * gx as w2e_torgb_bwd_acc (gx_acc may be NULL). */
int w2e_torgb_styled_fwd(const float* x, const float* wsc, const float* style, const float* bias, const float* skip,
                         const float* upk, float* y, int batch, int cin, int h, int w, void* stream);
int w2e_torgb_styled_bwd(const float* x, const float* wsc, const float* style, const float* gy, const float* gx_acc,
                         float* gx, float* gstyle, int batch, int cin, int h, int w, void* stream);
/* ToRGB backward fused with the activation backward of the StyledConv that produced x (x = its output, model.py:334-340 feeding
 * :353): instead of gx it writes gpre = gx * gain * (x > 0 ? 1 : slope), and sums3 [B,cin,3] = the three per-(b, channel) sums of
 * w2e_bias_act_bwd_reduce (gpre * pre-activation, gpre * noise, gpre), so that layer needs no separate activation-backward pass.
 * style NULL: wmod [B,3,cin], gw [B,3,cin]; style [B,cin]: wmod = the shared [3,cin] weight, gw = the style gradient [B,cin].
 * gx_acc (the gradient that reached x through its other consumer) and noise [h*w] may be NULL. */
int w2e_torgb_bwd_actbwd(const float* x, const float* wmod, const float* style, const float* gy, const float* gx_acc,
                         const float* noise, float* gpre, float* gw, float* sums3, int batch, int cin, int h, int w, float slope,
                         float gain, void* stream);

/* ---- K5  CLIP preprocessing  (criteria/clip_loss.py:11-12,15) ------------------------------
 * AvgPool2d(size/32)(Upsample(x7, nearest)(img)) in closed form: [planes,size,size] -> [planes,224,224];
 * never materialises the 7x image.  bwd is the exact adjoint. */
int w2e_clip_preproc_fwd(const float* img, float* out, int64_t planes, int size, void* stream);
int w2e_clip_preproc_bwd(const float* gout, float* gimg, int64_t planes, int size, void* stream);

/* ---- K5b  ArcFace (IDLoss) preprocessing  (criteria/id_loss.py:13-14,19-23) -----------------
 * AdaptiveAvgPool2d(256) -> crop [35:223, 32:220] -> AdaptiveAvgPool2d(112) in one pass:
 * [planes,size,size] -> [planes,112,112], size a multiple of 256.  bwd is the exact adjoint. */
int w2e_id_preproc_fwd(const float* img, float* out, int64_t planes, int size, void* stream);
int w2e_id_preproc_bwd(const float* gout, float* gimg, int64_t planes, int size, void* stream);

/* ---- K6  region-attention blend  (attention/attention_model.py:548-549 and siblings) --------
 * m = nearest-resize(mask[B,1,ms,ms]) to [h,w];  out = m*a + (1-m)*b, a,b [B,C,h,w]. */
int w2e_mask_blend_fwd(const float* a, const float* b, const float* mask, float* out, int batch, int channels,
                       int h, int w, int ms, void* stream);
/* ga = m*gout; gb = (1-m)*gout (gb may be NULL); gmask[b,my,mx] = sum over channels and covered
 * pixels of gout*(a-b) (gmask may be NULL; written, not accumulated). */
int w2e_mask_blend_bwd(const float* gout, const float* a, const float* b, const float* mask, float* ga, float* gb,
                       float* gmask, int batch, int channels, int h, int w, int ms, void* stream);

/* ---- K7  the latent mapper MLPs  (mapper/latent_mappers.py:10-82; EqualLinear: model.py:130-164) -----------------------------
 * LevelsMapper = one Mapper (PixelNorm + 4 x EqualLinear(512, 512, lr_mul, fused_lrelu)) per level of W+; a level ("group") g is
 * the latent range [l0[g], l0[g] + len[g]) (HOST int arrays, groups <= 4).  Activations are [batch*sum(len), 512] in group-major
 * row order: row r0_g + b*len[g] + l = latent l0[g] + l of sample b.  Weight arrays are HOST arrays of `groups` device pointers to
 * [512,512] matrices (row = output feature).  One launch covers one layer of all groups.
 *   pixelnorm: h = x * rsqrt(mean over the level's latents of x^2 + 1e-8) per (sample, feature) (PixelNorm's dim=1 on a [B,L,512]
 *              slice, latent_mappers.py:16), x [batch, n_latent, 512], gathered into group-major rows.
 *   linear, mode 0: out = lrelu(w_scale * a W_g^T + b_scale * bias_g, 0.2) * sqrt2;  scatter != 0: out is [batch, n_latent, 512].
 *   linear, mode 1: out = w_scale * (a .* lrelu'(y_act)) W_g^T with W_g = the TRANSPOSED weight: the input gradient of a layer whose
 *                   output was y_act and output gradient a (both group-major).
 *   wgrad: gw_g = w_scale * gpre^T h_in, gb_g = b_scale * column sums of gpre, gpre = gy .* lrelu'(y); gathered = 0: gy and y are
 *          [batch, n_latent, 512] (the last layer), else group-major.  Fixed summation order (deterministic).
 *   gather: [batch, n_latent, 512] -> group-major rows.   transpose: wt[j] = w[j]^T for `count` <= 16 matrices. */
int w2e_mapper_pixelnorm(const float* x, float* h, int batch, int n_latent, int groups, const int* l0, const int* len, void* stream);
int w2e_mapper_linear(int mode, const float* a, const float* y_act, float* out, const float* const* w, const float* const* bias,
                      int batch, int n_latent, int groups, const int* l0, const int* len, float w_scale, float b_scale, int scatter,
                      void* stream);
int w2e_mapper_wgrad(const float* gy, const float* y, const float* h_in, float* const* gw, float* const* gb, int batch, int n_latent,
                     int groups, const int* l0, const int* len, float w_scale, float b_scale, int gathered, void* stream);
int w2e_mapper_gather(const float* src, float* dst, int batch, int n_latent, int groups, const int* l0, const int* len, void* stream);
int w2e_mapper_transpose(const float* const* w, int count, float* wt, void* stream);

/* The style-space mappers (mapper/latent_mappers.py:84-128: FullStyleSpaceMapper / WithoutToRGBStyleSpaceMapper): `groups` (<= 32)
 * independent Mappers, Mapper c = PixelNorm over the dims[c] FEATURES of a [batch, dims[c]] code + 4 x EqualLinear(dims[c], dims[c],
 * lr_mul 0.01, fused_lrelu); batch <= 16.  A layer of ALL codes is one launch per direction.  Activations are packed code-major:
 * code c is the row-major [batch, dims[c]] block at float offset batch * sum_{i<c} dims[i]; weights / biases / gradients go by
 * pointer array, w_scale[c] = that code's EqualLinear scale (lr_mul / sqrt(dims[c])), b_scale = lr_mul.
 *   ssmapper_pixelnorm  h = packed PixelNorm of the code tensors x[c] ([batch, dims[c]] each)
 *   ssmapper_gather     dst = packed copy of src[c] ([batch, dims[c]]; a null entry is packed as zeros)
 *   ssmapper_linear     mode 0: out = lrelu(w_scale * a W^T + b_scale * bias) * sqrt2;
 *                       mode 1: out = w_scale * (a .* lrelu'(y_act)) W   (the layer's input gradient)
 *   ssmapper_wgrad      gw[c] = w_scale * (gy .* lrelu'(y))^T h_in,  gb[c] = b_scale * column sums of gy .* lrelu'(y) */
int w2e_ssmapper_pixelnorm(const float* const* x, float* h, int batch, int groups, const int* dims, void* stream);
int w2e_ssmapper_gather(const float* const* src, float* dst, int batch, int groups, const int* dims, void* stream);
int w2e_ssmapper_linear(int mode, const float* a, const float* y_act, float* out, const float* const* w, const float* const* bias,
                        int batch, int groups, const int* dims, const float* w_scale, float b_scale, void* stream);
int w2e_ssmapper_wgrad(const float* gy, const float* y, const float* h_in, float* const* gw, float* const* gb, int batch, int groups,
                       const int* dims, const float* w_scale, float b_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W2E_H */
