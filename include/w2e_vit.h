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

/* ---- the tower at M = 50*batch rows (csrc/vit2.hip: the kernels around the GEMMs; csrc/vit3.hip: the GEMM) ------------------------
 * Split-K partials of a GEMM are WRITTEN as slabs [rows, n], `slab` elements apart, and summed in ascending order by the kernel that
 * consumes them (no atomics, no memset, bit-reproducible); each producer can write the tensor the NEXT GEMM consumes in that GEMM's
 * operand packing (`*_packed_rows`).
 * Sum of split-K slabs [rows, n] fused with the QuickGELU pair of the MLP:
 *   mode 0:  h = sum + bias[n],  g = QuickGELU(h)      (after c_fc)
 *   mode 1:  h = sum * QuickGELU'(aux)                 (input gradient through the activation; g unused)
 * packed_rows > 0: the tensor the next GEMM consumes (g in mode 0, h in mode 1) is written K-quad-major with that many padded rows
 * (w2e_gemm_pk's A operand: P[q][row] = X[row][4q .. 4q+3]); mode 0's h stays row-major (kept for the backward). */
int w2e_reduce_gelu(const float* part, int nsplit, int64_t slab, const float* bias, const float* aux, float* h, float* g,
                    int64_t rows, int n, int mode, int packed_rows, void* stream);
/* x = sum_{s<nsplit} part[s] (+ bias[dim]) (+ residual) -> x_out (may be NULL);  y = LayerNorm(x)*gamma + beta with mean / rstd
 * saved (y may be NULL: reduction only).  part: nsplit slabs of [rows, dim], `slab` elements apart.  dim in {512, 768, 1024}. */
int w2e_reduce_ln_fwd(const float* part, int nsplit, int64_t slab, const float* bias, const float* residual, float* x_out,
                      const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int dim, float eps,
                      int y_packed_rows, void* stream);   /* y_packed_rows > 0: y written K-quad-major (w2e_gemm_pk's A operand) */
/* gx = LN'(sum_s gpart[s]) + add   (add may be NULL); gx_packed != NULL: a second copy of gx, K-quad-major with packed_rows padded rows. */
int w2e_layernorm_bwd_part(const float* gpart, int nsplit, int64_t slab, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const float* add, float* gx, int64_t rows, int dim, float* gx_packed, int packed_rows,
                           void* stream);
/* The attention core on MFMA.  qkv = sum of nsplit slabs [B*L, 3*heads*64] (+ bias[3*heads*64]); out [B*L, heads*64].
 * Backward: gout = sum of gsplit slabs [B*L, heads*64]; gqkv [B*L, 3*heads*64] (probabilities recomputed). */
int w2e_attn2_fwd(const float* qkv, int nsplit, int64_t slab, const float* bias, float* out, int batch, int seq, int heads,
                  int out_packed_rows, void* stream);     /* *_packed_rows > 0: out / gqkv written K-quad-major (w2e_gemm_pk's A operand) */
int w2e_attn2_bwd(const float* qkv, int nsplit, int64_t slab, const float* bias, const float* gout, int gsplit, int64_t gslab,
                  float* gqkv, int batch, int seq, int heads, int g_packed_rows, void* stream);

/* ---- third-generation tower GEMM (csrc/vit3.hip): both operands pre-packed K-quad-major, P[q][r] (one float4) = X[r][4q .. 4q+3] --
 * the word a lane feeds to four consecutive v_mfma_f32_32x32x2_f32 -- so that they go global/L2 -> registers with no LDS and no barrier:
 * one independent wave per (32 rows x 64 columns x K slice).  w2e_pack_kq packs X [rows, K] (row stride ldx; rows up to rows_padded are
 * zero-filled) or, transposed != 0, X^T of X [K, rows].  w2e_gemm_pk: C[z] = A x B^T over K slice z, written as `splits` slabs
 * [m, ldc] (their consumers -- w2e_reduce_ln_fwd, w2e_layernorm_bwd_part, w2e_reduce_gelu, w2e_attn2_* -- sum them in ascending order).  K %% 8 == 0; m_padded %% 32 == 0, n_padded %% 64 == 0. */
int w2e_pack_kq(const float* x, float* packed, int rows, int rows_padded, int k, int ldx, int transposed, void* stream);
int w2e_gemm_pk(const float* a_packed, const float* b_packed, float* c, int m, int n, int k, int m_padded, int n_padded, int ldc,
                int splits, void* stream);
/* The K split w2e_gemm_pk wants for a shape: about one wave per SIMD (tiles x splits ~ 4 x CUs), every slice >= 4 chunks of 8. */
int w2e_gemm_pk_splits(int m, int n, int k);
/* OPT-IN fp16 operands for the same GEMM (CLIP(...).set_precision("f16"); never the default): the arithmetic of the tower the reference
 * actually runs on a GPU -- criteria/clip_loss.py:10 `clip.load("ViT-B/32", device="cuda")` is OpenAI's fp16 model -- for the four Linear
 * layers of a block.  w2e_pack_kq_h packs X (or X^T) as fp16, PH[(s*2+h)*rows_padded + r] = the 8 halves X[r][16s+4h+c], X[r][16s+8+4h+c],
 * c = 0..3 -- the k a lane-half of one v_mfma_f32_32x32x16_f16 step holds when its A operand comes from two consecutive chunks of the
 * fp32 packing above.  w2e_gemm_pk_h: A stays the fp32 K-quad-major operand its producers write and is rounded to fp16 (nearest even) in
 * registers; B is the fp16 pack; fp32 accumulation; the same slabs.  K %% 16 == 0. */
int w2e_pack_kq_h(const float* x, void* packed_half, int rows, int rows_padded, int k, int ldx, int transposed, void* stream);
int w2e_gemm_pk_h(const float* a_packed, const void* b_packed_half, float* c, int m, int n, int k, int m_padded, int n_padded, int ldc,
                  int splits, void* stream);
int w2e_gemm_pk_h_splits(int m, int n, int k);

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
