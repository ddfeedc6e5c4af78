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

#ifdef __cplusplus
}
#endif
#endif /* W2E_VIT_H */
