/* w2e_attention.h -- C ABI of the region-attention mask kernels in libw2e.so (gfx950).
 *
 * They replace the mask branch of `FullSpaceMapperFEATClusterLinStyle_Net.forward`
 * (attention/run_attention.py:754-893) -- the model the reference repository is named after:
 *   w2e_cluster_assign    :775-792  nearest k-means centroid per feature pixel; the reference builds the position
 *                                   channels, a [B*s*s, C+2P] copy and a [N,K,C+2P] broadcast temp (utils.py:244-263)
 *   w2e_attention_logits  :796-841  1 + 17 `StyledConv(C, 32, 1)` on cached generator activations, nearest-resized to
 *                                   `size`, concatenated, `StyledConv(576, 1, 1)`, + initial_bias, sigmoid
 *   w2e_cluster_pool      :843-884  per-(sample, cluster) mean (the reference's Python loop over B*K boolean masks),
 *                                   straight-through threshold 0.8, torchvision gaussian_blur(5)
 * Same conventions as w2e.h (device fp32 pointers, caller-allocated outputs / workspaces, stream as void*, 0 = OK).
 * Forward only: the reference's schedule keeps every `attention*` / `initial*` parameter frozen for the whole run
 * (run_attention.py:1076-1083, `t < 1.15` is always true), so no gradient ever flows through these kernels.
 */
#ifndef W2E_ATTENTION_H
#define W2E_ATTENTION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* assign[b,y,x] = argmin_k sum_c (f[b,c,y,x] - cen[k,c])^2 + sum_p (xpos(x) - cen[k,C+p])^2 + sum_p (ypos(y) - cen[k,C+P+p])^2
 * with xpos(x) = 2x/(S-1) - 1, ypos likewise; feat [B,C,S,S]; centroids [K, C+2P] row-major; assign int32 [B,S,S].
 * Ties go to the lowest k (torch.argmin).  K <= 32, (C+2P)*K*4 <= 160 KB of LDS. */
int w2e_cluster_assign(const float* feat, const float* centroids, int32_t* assign, int batch, int channels, int pos_channels,
                       int size, int clusters, void* stream);

/* Lloyd's centroid-update step for the offline clustering (attention/clustering_feature.py:212-235, 373-397):
 * partial[b,k,d] = sum over the pixels of image b assigned to cluster k of dimension d (d < C: feat[b,d,.]; then the P
 * x-position and P y-position channels, evaluated as in w2e_cluster_assign), counts[b,k] = number of such pixels.
 * partial [B,K,C+2P], counts [B,K]; written, not accumulated; the caller sums over b.  K <= 32. */
int w2e_cluster_accumulate(const float* feat, const int32_t* assign, float* partial, float* counts, int batch, int channels,
                           int pos_channels, int size, int clusters, void* stream);

#define W2E_ATT_MAX_SOURCES 32
/* One source = one cached activation and the 1x1 StyledConv(C, 32, 1, C) applied to it with a style given in S-space:
 *   a[b,o,p] = lrelu( d[b,o] * sum_i wscaled[o,i] * s[b,i] * feat[b,i,src(p)] + nw*noise[b,p] + bias[o] ) * sqrt2
 * evaluated only at the `size` x `size` pixels p that F.interpolate(., size) (nearest) would keep / replicate:
 * src(y,x) = (floor(y*res/size), floor(x*res/size)).  wscaled = scale*W [32,C];  d = the demodulation coefficients
 * [B,32]; noise [B,size*size] or NULL (NoiseInjection draws randn when no noise is passed: i.i.d., so drawing it
 * at the kept pixels is the same distribution), nw = device scalar noise strength. */
typedef struct {
    const float* feat;     /* [B, channels, res, res] */
    const float* wscaled;  /* [channels, 32] = (conv.weight[0,:,:,0,0] / sqrt(channels))^T: the transposed 1x1 weight */
    const float* style;    /* [B, channels] */
    const float* demod;    /* [B, 32] */
    const float* bias;     /* [32] (activate.bias) */
    const float* noise;    /* [B, size*size] or NULL */
    const float* noise_w;  /* device scalar (noise.weight) */
    int channels, res;
} w2e_att_source;

/* The demodulation coefficients of every source in one launch (model.py:244-246 for the [32,C,1,1] weights):
 *   sources[j].demod[b,o] = rsqrt( sum_i (wscaled_j[i,o] * style_j[b,i])^2 + eps )      (written; [B,32] per source)
 * Reads only wscaled / style / channels of each descriptor.  Deterministic. */
int w2e_attention_demod(const w2e_att_source* sources, int n_sources, int batch, float eps, void* stream);

/* each[b,p] = sigmoid( lrelu( d_last[b] * sum_{j,o} wlast[32j+o] * s_last[b,32j+o] * a_j[b,o,p] + nw_last*noise_last[b,p]
 *                              + bias_last ) * sqrt2 + initial_bias )
 * sources: HOST array of n_sources descriptors (copied into the launch).  wlast = scale*W of attention_last [32*n],
 * s_last [B,32*n], d_last [B], bias_last / initial_bias / nw_last device scalars, noise_last [B,size*size] or NULL.
 * partial: workspace of n_sources*B*size*size floats.  each: [B,size*size].  Deterministic (no atomics). */
int w2e_attention_logits(const w2e_att_source* sources, int n_sources, const float* wlast, const float* s_last,
                         const float* d_last, const float* bias_last, const float* noise_last, const float* nw_last,
                         const float* initial_bias, float* partial, float* each, int batch, int size, void* stream);

/* Per sample: mean of each[b,.] over the pixels of every cluster (assign given at cluster resolution csize, read through
 * the nearest resize to `size`), written back to the pixels -> same[b,p] (1.0 where a pixel's cluster id is out of
 * range); means[b,k] (NaN-free: 0 for empty clusters), counts[b,k];  thr = same < threshold ? 0 : same;
 * final = 5x5 gaussian (sigma 1.1, reflect padding) of thr.  size <= 128, K <= 32.  thr may be NULL. */
int w2e_cluster_pool(const float* each, const int32_t* assign, float* same, float* means, float* counts, float* thr,
                     float* final_map, int batch, int size, int csize, int clusters, float threshold, void* stream);

#ifdef __cplusplus
}
#endif
#endif
