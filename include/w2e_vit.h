/* w2e_vit.h -- C ABI of the CLIP ViT-B/32 image-encoder kernels in libw2e.so (gfx950).
 *
 * They replace the torch modules behind `self.model(image, text)` in criteria/clip_loss.py:16, i.e. the
 * visual tower of OpenAI CLIP (clip/model.py: VisionTransformer, ResidualAttentionBlock, LayerNorm,
 * QuickGELU -- a third-party package the reference pins in requirements.txt:31 / cog.yaml:22, not
 * vendored).  CLIP is a frozen critic on this path: forward and INPUT gradients only, no weight gradients.
 * Same conventions as w2e.h (device fp32 pointers, caller-allocated outputs, stream as void*, 0 = OK).
 */
#ifndef W2E_VIT_H
#define W2E_VIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fp32-MFMA GEMM:  C[M,N] = epi( pro(A)[M,K] x B )  with row-major operands.
 *   trans_b = 1:  B is [N,K]  (y = x W^T, nn.Linear forward: QKV / out_proj / c_fc / c_proj / patch embed)
 *   trans_b = 0:  B is [K,N]  (dx = dy W, the input gradient of the same layers)
 *   a_gelu  = 1:  pro(a) = a * sigmoid(1.702 a)   (QuickGELU applied while A is staged: c_proj reads the
 *                 pre-activation h once, gelu(h) is never materialised)
 *   bias[N] or NULL is added;  residual[M,N] or NULL is added;
 *   gelu_grad_aux[M,N] or NULL: the result is multiplied by QuickGELU'(aux) (c_fc's input gradient).
 * lda/ldb/ldc are row strides in elements. */
int w2e_gemm(const float* a, const float* b, float* c, int m, int n, int k, int lda, int ldb, int ldc, int trans_b,
             int a_gelu, const float* bias, const float* residual, const float* gelu_grad_aux, void* stream);
/* The same with c_is_zero != 0: the caller guarantees C holds zeros, so a split-K launch (the skinny N = 768 shapes) adds
 * onto it without the memset of its own -- a transformer pass zero-fills one arena for all such outputs at once. */
int w2e_gemm_ex(const float* a, const float* b, float* c, int m, int n, int k, int lda, int ldb, int ldc, int trans_b,
                int a_gelu, const float* bias, const float* residual, const float* gelu_grad_aux, int c_is_zero, void* stream);

/* LayerNorm over the last dim (eps inside rsqrt), rows x dim.  fwd saves mean and rstd (rows each). */
int w2e_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int64_t rows, int dim, float eps, void* stream);
/* Input gradient only (gamma/beta are frozen). */
int w2e_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* gx, int64_t rows, int dim, void* stream);
/* The same plus `add` [rows, dim] (or NULL): gx = LN'(gy) + add -- the residual branch of a pre-LN block joins here. */
int w2e_layernorm_bwd_add(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* add, float* gx, int64_t rows, int dim, void* stream);

/* Multi-head self-attention core on a packed QKV tensor [B, L, 3, heads, 64] (the layout nn.MultiheadAttention's
 * in_proj produces): out[B, L, heads*64] = softmax(Q K^T / 8) V per (batch, head).  L <= 64, head dim = 64.
 * One workgroup per (batch, head); Q/K/V live in LDS. */
int w2e_attn_fwd(const float* qkv, float* out, int batch, int seq, int heads, void* stream);
/* gqkv [B,L,3,heads,64] from gout [B,L,heads*64]; probabilities are recomputed from qkv. */
int w2e_attn_bwd(const float* qkv, const float* gout, float* gqkv, int batch, int seq, int heads, void* stream);

/* ---- second generation, shaped for M = 50*batch rows (csrc/vit2.hip) ----------------------------------------------------
 * w2e_gemm_fm:  C[M,N] = A[M,K] x B[N,K]^T  (both operands row-major with K contiguous: nn.Linear's weight for the forward,
 * its cached transpose for the input gradient).  One workgroup holds ALL rows of a 224-row M tile (7 waves x 32 rows) and 32
 * columns; K is split over `splits` slices.  epi:
 *   0 PLAIN      c = acc + bias                                   (splits = 1)
 *   1 PARTIAL    c[z] = acc of K-slice z, one [M, ldc] slab per slice (slab stride M*ldc), no bias: the CONSUMER sums the
 *                slabs (w2e_reduce_ln_fwd, w2e_layernorm_bwd_part, w2e_attn2_*) -- no atomics, no memset, deterministic
 *   2 GELU_DUAL  c = acc + bias, c2 = QuickGELU(c)                (c_fc: the pre-activation is kept for the backward)
 *   3 GELU_GRAD  c = acc * QuickGELU'(aux)                        (input gradient through c_proj and the activation)
 * K % 32 == 0, lda/ldb % 4 == 0, 16-byte aligned operands.  w2e_gemm_fm_splits suggests `splits` for a shape. */
int w2e_gemm_fm_splits(int m, int n, int k, int allow_split);
int w2e_gemm_fm(const float* a, const float* b, float* c, float* c2, int m, int n, int k, int lda, int ldb, int ldc, int splits,
                int epi, const float* bias, const float* aux, void* stream);
/* Sum of split-K slabs [rows, n] fused with the QuickGELU pair of the MLP:
 *   mode 0:  h = sum + bias[n],  g = QuickGELU(h)      (after c_fc)
 *   mode 1:  h = sum * QuickGELU'(aux)                 (input gradient through the activation; g unused) */
int w2e_reduce_gelu(const float* part, int nsplit, int64_t slab, const float* bias, const float* aux, float* h, float* g,
                    int64_t rows, int n, int mode, void* stream);
/* x = sum_{s<nsplit} part[s] (+ bias[dim]) (+ residual) -> x_out (may be NULL);  y = LayerNorm(x)*gamma + beta with mean / rstd
 * saved (y may be NULL: reduction only).  part: nsplit slabs of [rows, dim], `slab` elements apart.  dim in {512, 768, 1024}. */
int w2e_reduce_ln_fwd(const float* part, int nsplit, int64_t slab, const float* bias, const float* residual, float* x_out,
                      const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int dim, float eps,
                      void* stream);
/* gx = LN'(sum_s gpart[s]) + add   (add may be NULL). */
int w2e_layernorm_bwd_part(const float* gpart, int nsplit, int64_t slab, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const float* add, float* gx, int64_t rows, int dim, void* stream);
/* The attention core on MFMA.  qkv = sum of nsplit slabs [B*L, 3*heads*64] (+ bias[3*heads*64]); out [B*L, heads*64].
 * Backward: gout = sum of gsplit slabs [B*L, heads*64]; gqkv [B*L, 3*heads*64] (probabilities recomputed). */
int w2e_attn2_fwd(const float* qkv, int nsplit, int64_t slab, const float* bias, float* out, int batch, int seq, int heads,
                  void* stream);
int w2e_attn2_bwd(const float* qkv, int nsplit, int64_t slab, const float* bias, const float* gout, int gsplit, int64_t gslab,
                  float* gqkv, int batch, int seq, int heads, void* stream);

/* ---- the scalar tail of a mapper step (csrc/losstail.hip): ~37 [B,T]- / [B,18,512]-sized stock launches as four -------------------
 * criteria/clip_loss.py:16 + the tail of OpenAI clip.model.CLIP.forward: out[b,t] = exp(*logit_scale) * <f_b, t_t> / (|f_b| |t_t|)
 * (feat [B,D], text [T,D], logit_scale a DEVICE scalar holding the log of the scale); similarity != 0: out = 1 - that / 100 (the
 * loss's similarity).  bwd: the gradient to feat (text and the scale are frozen on this path). */
int w2e_clip_logits_fwd(const float* feat, const float* text, const float* logit_scale, float* out, int batch, int n_text, int dim,
                        int similarity, void* stream);
int w2e_clip_logits_bwd(const float* gout, const float* feat, const float* text, const float* logit_scale, float* gfeat, int batch,
                        int n_text, int dim, int similarity, void* stream);
/* mapper/training/coach.py:223-245 for the clip + latent-L2 terms: out3 = { clip_lambda * mean(sim) + l2_lambda * MSE(w_hat, w),
 * mean(sim), MSE(w_hat, w) } (sim: n_sim floats; w_hat, w: n_w floats; either count may be 0).  bwd: g_sim[i] = g * clip_lambda /
 * n_sim, g_what = g * l2_lambda * 2 (w_hat - w) / n_w with g = *g_loss (a device scalar); g_sim / g_what may be NULL. */
int w2e_step_loss_fwd(const float* sim, int n_sim, const float* w_hat, const float* w, int64_t n_w, float clip_lambda, float l2_lambda,
                      float* out3, void* stream);
int w2e_step_loss_bwd(const float* g_loss, int n_sim, const float* w_hat, const float* w, int64_t n_w, float clip_lambda, float l2_lambda,
                      float* g_sim, float* g_what, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W2E_VIT_H */
