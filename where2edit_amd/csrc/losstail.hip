// The scalar tail of a mapper step as four launches instead of ~37 stock ones (gfx950):
//   clip_logits   criteria/clip_loss.py:16 + OpenAI clip.model.CLIP.forward's tail: normalise image and text features, cosine logits times
//                 exp(logit_scale), optionally as the loss's similarity 1 - logits/100; forward and the gradient to the image features
//   step_loss     mapper/training/coach.py:223-245: clip_lambda * mean(similarity) + l2_lambda * MSE(w_hat, w); forward and backward
// [B, T] / [B, 18, 512]-sized work: nothing here is bound by anything but launch count.  Fixed reduction orders (bit-reproducible).
#include "common.h"
#include "../../include/w2e_vit.h"

namespace w2e {

__device__ __forceinline__ float lt_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block reduction of one value over 256 threads (4 waves), result to every thread
__device__ __forceinline__ float lt_block_sum(float v, float* sm) {
    v = lt_wave_sum(v);
    __syncthreads();  // (sm may still be read from a previous reduction)
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// One workgroup (256 threads) per image b.  out[b,t] = s * <f_b, t_t> / (|f_b| |t_t|), s = exp(*logit_scale); SIM: 1 - that / 100.
template <bool SIM>
__global__ __launch_bounds__(256) void clip_logits_fwd_kernel(const float* __restrict__ f, const float* __restrict__ txt,
                                                              const float* __restrict__ logit_scale, float* __restrict__ out, int D, int T) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    const float* fb = f + (int64_t)b * D;
    float ff = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) ff += fb[i] * fb[i];
    ff = lt_block_sum(ff, sm);
    const float s = __expf(logit_scale[0]);
    for (int t = 0; t < T; ++t) {
        const float* tt = txt + (int64_t)t * D;
        float ft = 0.f, t2 = 0.f;
        for (int i = threadIdx.x; i < D; i += 256) ft += fb[i] * tt[i], t2 += tt[i] * tt[i];
        ft = lt_block_sum(ft, sm);
        t2 = lt_block_sum(t2, sm);
        const float logit = s * ft / (sqrtf(ff) * sqrtf(t2));
        if (threadIdx.x == 0) out[(int64_t)b * T + t] = SIM ? 1.f - logit / 100.f : logit;
    }
}

// gf[b,:] = sum_t g'[b,t] * s / |f_b| * (that_t - c_bt fhat_b),  g' = SIM ? -g/100 : g,  c_bt = <fhat_b, that_t>
template <bool SIM>
__global__ __launch_bounds__(256) void clip_logits_bwd_kernel(const float* __restrict__ g, const float* __restrict__ f,
                                                              const float* __restrict__ txt, const float* __restrict__ logit_scale,
                                                              float* __restrict__ gf, int D, int T) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    const float* fb = f + (int64_t)b * D;
    float ff = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) ff += fb[i] * fb[i];
    ff = lt_block_sum(ff, sm);
    const float inv_f = rsqrtf(ff);
    const float s = __expf(logit_scale[0]);
    for (int i = threadIdx.x; i < D; i += 256) gf[(int64_t)b * D + i] = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* tt = txt + (int64_t)t * D;
        float ft = 0.f, t2 = 0.f;
        for (int i = threadIdx.x; i < D; i += 256) ft += fb[i] * tt[i], t2 += tt[i] * tt[i];
        ft = lt_block_sum(ft, sm);
        t2 = lt_block_sum(t2, sm);
        const float inv_t = rsqrtf(t2);
        const float c = ft * inv_f * inv_t;
        float gg = g[(int64_t)b * T + t];
        if (SIM) gg = -gg / 100.f;
        const float k = gg * s * inv_f;
        for (int i = threadIdx.x; i < D; i += 256) gf[(int64_t)b * D + i] += k * (tt[i] * inv_t - c * fb[i] * inv_f);  // (each thread owns its i's)
    }
}

// out[0] = clip_lambda * mean(sim) + l2_lambda * mean((w_hat - w)^2), out[1] = mean(sim), out[2] = the MSE.  ONE workgroup of 1024.
__global__ __launch_bounds__(1024) void step_loss_fwd_kernel(const float* __restrict__ sim, int n_sim, const float* __restrict__ w_hat,
                                                             const float* __restrict__ w, int64_t n_w, float clip_lambda, float l2_lambda,
                                                             float* __restrict__ out) {
    __shared__ float sm[2][16];
    float a = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < n_sim; i += 1024) a += sim[i];
    for (int64_t i = threadIdx.x; i < n_w; i += 1024) {
        const float d = w_hat[i] - w[i];
        q += d * d;
    }
    a = lt_wave_sum(a), q = lt_wave_sum(q);
    if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = a, sm[1][threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sq = 0.f;
        for (int i = 0; i < 16; ++i) sa += sm[0][i], sq += sm[1][i];
        const float lc = n_sim > 0 ? sa / n_sim : 0.f, l2 = n_w > 0 ? sq / (float)n_w : 0.f;
        out[0] = clip_lambda * lc + l2_lambda * l2, out[1] = lc, out[2] = l2;
    }
}

// g_sim[i] = g * clip_lambda / n_sim;  g_what[i] = g * l2_lambda * 2 (w_hat - w) / n_w     (g = *g_loss, a device scalar)
__global__ __launch_bounds__(256) void step_loss_bwd_kernel(const float* __restrict__ g_loss, int n_sim, const float* __restrict__ w_hat,
                                                            const float* __restrict__ w, int64_t n_w, float clip_lambda, float l2_lambda,
                                                            float* __restrict__ g_sim, float* __restrict__ g_what) {
    const float g = g_loss[0];
    const int64_t step = (int64_t)gridDim.x * 256;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g_sim) {
        const float gs = n_sim > 0 ? g * clip_lambda / n_sim : 0.f;
        for (int64_t i = i0; i < n_sim; i += step) g_sim[i] = gs;
    }
    if (g_what) {
        const float k = n_w > 0 ? g * l2_lambda * 2.f / (float)n_w : 0.f;
        for (int64_t i = i0; i < n_w; i += step) g_what[i] = k * (w_hat[i] - w[i]);
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_clip_logits_fwd(const float* feat, const float* text, const float* logit_scale, float* out, int batch, int n_text, int dim,
                        int similarity, void* stream) {
    W2E_REQUIRE(feat && text && logit_scale && out, "clip_logits_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && n_text > 0 && dim > 0, "clip_logits_fwd: bad dims");
    if (batch == 0) return 0;
    if (similarity) clip_logits_fwd_kernel<true><<<batch, 256, 0, (hipStream_t)stream>>>(feat, text, logit_scale, out, dim, n_text);
    else clip_logits_fwd_kernel<false><<<batch, 256, 0, (hipStream_t)stream>>>(feat, text, logit_scale, out, dim, n_text);
    W2E_LAUNCH_CHECK("clip_logits_fwd");
    return 0;
}

int w2e_clip_logits_bwd(const float* gout, const float* feat, const float* text, const float* logit_scale, float* gfeat, int batch,
                        int n_text, int dim, int similarity, void* stream) {
    W2E_REQUIRE(gout && feat && text && logit_scale && gfeat, "clip_logits_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && n_text > 0 && dim > 0, "clip_logits_bwd: bad dims");
    if (batch == 0) return 0;
    if (similarity) clip_logits_bwd_kernel<true><<<batch, 256, 0, (hipStream_t)stream>>>(gout, feat, text, logit_scale, gfeat, dim, n_text);
    else clip_logits_bwd_kernel<false><<<batch, 256, 0, (hipStream_t)stream>>>(gout, feat, text, logit_scale, gfeat, dim, n_text);
    W2E_LAUNCH_CHECK("clip_logits_bwd");
    return 0;
}

int w2e_step_loss_fwd(const float* sim, int n_sim, const float* w_hat, const float* w, int64_t n_w, float clip_lambda, float l2_lambda,
                      float* out3, void* stream) {
    W2E_REQUIRE(out3, "step_loss_fwd: null output");
    W2E_REQUIRE(n_sim >= 0 && n_w >= 0 && (n_sim == 0 || sim) && (n_w == 0 || (w_hat && w)), "step_loss_fwd: null tensor");
    step_loss_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(sim, n_sim, w_hat, w, n_w, clip_lambda, l2_lambda, out3);
    W2E_LAUNCH_CHECK("step_loss_fwd");
    return 0;
}

int w2e_step_loss_bwd(const float* g_loss, int n_sim, const float* w_hat, const float* w, int64_t n_w, float clip_lambda, float l2_lambda,
                      float* g_sim, float* g_what, void* stream) {
    W2E_REQUIRE(g_loss, "step_loss_bwd: null gradient");
    W2E_REQUIRE(n_sim >= 0 && n_w >= 0 && (!g_what || (w_hat && w)), "step_loss_bwd: null tensor");
    const int64_t n = n_w > n_sim ? n_w : n_sim;
    if (n == 0) return 0;
    step_loss_bwd_kernel<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(g_loss, n_sim, w_hat, w, n_w, clip_lambda, l2_lambda, g_sim, g_what);
    W2E_LAUNCH_CHECK("step_loss_bwd");
    return 0;
}

}  // extern "C"
