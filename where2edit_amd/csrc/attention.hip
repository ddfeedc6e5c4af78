// Region-attention mask kernels (include/w2e_attention.h): nearest-centroid assignment, the 18 gathered 1x1 StyledConvs +
// the 576->1 StyledConv + sigmoid, and the per-cluster pooling + threshold + 5x5 gaussian.  All HBM/latency-bound fp32
// VALU work on at most [B,576,64,64]-sized data; no MFMA on purpose (N = 32 and N = 1 outputs, 6.4 GFLOP per batch of 4).
#include "../../include/w2e_attention.h"
#include "common.h"

namespace w2e {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------- cluster assignment
// A workgroup owns 64 pixels; its 4 waves split the channel range (wave w: channels [w*C/4, (w+1)*C/4), the 2P analytic
// position channels ride with wave 3) and the partial distances are joined through LDS in wave order -- 4x the workgroups
// and 1/4 of the per-thread channel walk of a thread-per-pixel kernel (B*s*s = 16-32 K pixels is far too few threads for
// 256 CUs).  Lanes run along x (coalesced channel-plane reads); centroids sit in LDS as [dim][KP] so the K distances of one
// channel come from KP/4 broadcast ds_read_b128.  Distances accumulate in the reference's form sum (a-b)^2 (no
// |a|^2 - 2ab + |b|^2 cancellation).
template <int KP>
__global__ __launch_bounds__(256) void cluster_assign_kernel(const float* __restrict__ feat, const float* __restrict__ cen,
                                                             int32_t* __restrict__ assign, int C, int P, int S, int K) {
    extern __shared__ float lds[];  // [C + 2P][KP] centroids, then [4][64][KP] partial distances
    const int D = C + 2 * P;
    for (int e = threadIdx.x; e < D * KP; e += 256) {
        const int d = e / KP, k = e % KP;
        lds[e] = k < K ? cen[(int64_t)k * D + d] : 0.f;
    }
    __syncthreads();
    float* part = lds + D * KP;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pix = blockIdx.x * 64 + lane;
    const bool live = pix < S * S;
    float dist[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) dist[k] = 0.f;
    const int c_lo = (C * wave) >> 2, c_hi = (C * (wave + 1)) >> 2;
    const float* f = feat + (int64_t)b * C * S * S + (live ? pix : 0);
    for (int c8 = c_lo; c8 < c_hi; c8 += 8) {  // 8 channel planes' loads in flight before the first use (each is a DRAM/L2 miss)
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (c8 + u < c_hi) ? f[(int64_t)(c8 + u) * S * S] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c8 + u >= c_hi) break;
            const float v = v8[u];
            const float4* row = reinterpret_cast<const float4*>(lds + (c8 + u) * KP);
#pragma unroll
            for (int q = 0; q < KP / 4; ++q) {
                const float4 m = row[q];
                float t;
                t = v - m.x, dist[4 * q + 0] += t * t;
                t = v - m.y, dist[4 * q + 1] += t * t;
                t = v - m.z, dist[4 * q + 2] += t * t;
                t = v - m.w, dist[4 * q + 3] += t * t;
            }
        }
    }
    if (wave == 3) {
        const int y = pix / S, x = pix % S;
        const float xp = (float)x * 2.f / (float)(S - 1) - 1.f, yp = (float)y * 2.f / (float)(S - 1) - 1.f;
        for (int pc = 0; pc < 2 * P; ++pc) {
            const float v = pc < P ? xp : yp;
            const float* row = lds + (C + pc) * KP;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float t = v - row[k];
                dist[k] += t * t;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) part[(wave * 64 + lane) * KP + k] = dist[k];
    __syncthreads();
    if (wave != 0 || !live) return;
    int best = 0;
    float bd = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const float d = ((part[lane * KP + k] + part[(64 + lane) * KP + k]) + part[(128 + lane) * KP + k]) + part[(192 + lane) * KP + k];
        if (k == 0) bd = d;
        else if (k < K && d < bd) bd = d, best = k;
    }
    assign[(int64_t)b * S * S + pix] = best;
}

// ---------------------------------------------------------------------------------------- k-means centroid update
// Lloyd's update step for the offline clustering (attention/clustering_feature.py:212-235): per image b and feature
// dimension d (a channel plane, or one of the 2P analytic position channels) the sum of that dimension over the pixels of
// every cluster.  One workgroup per (d, b); each thread keeps K running sums in registers, then a fixed-order reduction
// (wave shuffles, 4 wave partials).  partial[b][k][d]; the caller sums over b (deterministic, no atomics).
template <int KP>
__global__ __launch_bounds__(256) void cluster_accumulate_kernel(const float* __restrict__ feat, const int32_t* __restrict__ assign,
                                                                 float* __restrict__ partial, float* __restrict__ counts, int C,
                                                                 int P, int S, int K) {
    __shared__ float red[4][KP];
    const int d = blockIdx.x, b = blockIdx.y, D = C + 2 * P, npix = S * S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    const int32_t* a = assign + (int64_t)b * npix;
    const float* f = d < C ? feat + ((int64_t)b * C + d) * npix : nullptr;
    for (int p = threadIdx.x; p < npix; p += 256) {
        float v;
        if (d < C) v = f[p];
        else if (d < C + P) v = (float)(p % S) * 2.f / (float)(S - 1) - 1.f;
        else if (d < D) v = (float)(p / S) * 2.f / (float)(S - 1) - 1.f;
        else v = 1.f;  // d == D: the pixel count
        const int k = a[p];
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) acc[kk] += (kk == k) ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const float t = wave_sum64(acc[k]);
        if (lane == 0) red[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        const float t = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (d < D) partial[((int64_t)b * K + k) * D + d] = t;
        else counts[(int64_t)b * K + k] = t;
    }
}

// Demodulation coefficients of every source's modulated 1x1 conv (model.py:244-246 with a [32,C,1,1] weight) in one launch:
// d_j[b,o] = rsqrt(sum_i (wscaled_j[i,o] * style_j[b,i])^2 + eps).  grid (source, batch); thread = (output o, one of 8 channel
// residues); fixed reduction order.
struct AttDemodLaunch {
    const float* wscaled[W2E_ATT_MAX_SOURCES];
    const float* style[W2E_ATT_MAX_SOURCES];
    float* demod[W2E_ATT_MAX_SOURCES];
    int channels[W2E_ATT_MAX_SOURCES];
    float eps;
};

__global__ __launch_bounds__(256) void att_demod_kernel(const AttDemodLaunch L) {
    __shared__ float part[8][32];
    const int j = blockIdx.x, b = blockIdx.y;
    const int o = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int C = L.channels[j];
    const float* w = L.wscaled[j];
    const float* st = L.style[j] + (int64_t)b * C;
    float acc = 0.f;
    for (int i = r; i < C; i += 8) {
        const float v = w[i * 32 + o] * st[i];
        acc += v * v;
    }
    part[r][o] = acc;
    __syncthreads();
    if (r == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][o];
        L.demod[j][(int64_t)b * 32 + o] = rsqrtf(t + L.eps);
    }
}

// ---------------------------------------------------------------------------------------- attention logits
struct AttLaunch {
    w2e_att_source src[W2E_ATT_MAX_SOURCES];
    const float* wlast;
    const float* s_last;
    float* partial;
    int n_sources, batch, size;
};

constexpr int ATT_CH = 128;  // channels staged per LDS pass: [128][32] floats = 16 KB

// grid (pixel tiles of 256, batch, source).  A thread owns one output pixel and the 32 outputs of its source's 1x1 conv;
// the modulated weights scale*W[o,i]*s[b,i] of a 128-channel slice sit in LDS as [i][32] (8 broadcast ds_read_b128 per
// channel for 32 FMAs).  Writes the source's contribution to the final 576->1 conv: sum_o wlast*s_last*a[o].
__global__ __launch_bounds__(256) void att_source_kernel(const AttLaunch L) {
    __shared__ __attribute__((aligned(16))) float wm[ATT_CH * 32];
    __shared__ float dcoef[32], bcoef[32], lcoef[32];
    const int j = blockIdx.z, b = blockIdx.y;
    const w2e_att_source& s = L.src[j];
    const int C = s.channels, R = s.res, size = L.size;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const bool live = pix < size * size;
    const int y = live ? pix / size : 0, x = live ? pix % size : 0;
    // F.interpolate(mode='nearest') source index: floor(dst * in / out)
    const int sy = (int)(((int64_t)y * R) / size), sx = (int)(((int64_t)x * R) / size);
    const float* f = s.feat + (int64_t)b * C * R * R + (int64_t)sy * R + sx;
    if (threadIdx.x < 32) {
        dcoef[threadIdx.x] = s.demod[b * 32 + threadIdx.x];
        bcoef[threadIdx.x] = s.bias[threadIdx.x];
        lcoef[threadIdx.x] = L.wlast[j * 32 + threadIdx.x] * L.s_last[(int64_t)b * 32 * L.n_sources + j * 32 + threadIdx.x];
    }
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < C; c0 += ATT_CH) {
        const int cn = (C - c0 < ATT_CH) ? C - c0 : ATT_CH;
        __syncthreads();
        for (int e = threadIdx.x; e < cn * 32; e += 256) {
            const int i = e >> 5, o = e & 31;
            wm[e] = s.wscaled[(int64_t)(c0 + i) * 32 + o] * s.style[(int64_t)b * C + c0 + i];  // wscaled is [C][32]: coalesced
        }
        __syncthreads();
        if (live) {
            for (int i = 0; i < cn; ++i) {
                const float v = f[(int64_t)(c0 + i) * R * R];
                const float4* row = reinterpret_cast<const float4*>(wm + i * 32);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 m = row[q];
                    acc[4 * q + 0] += m.x * v, acc[4 * q + 1] += m.y * v, acc[4 * q + 2] += m.z * v, acc[4 * q + 3] += m.w * v;
                }
            }
        }
    }
    if (!live) return;
    const float nz = s.noise ? s.noise_w[0] * s.noise[(int64_t)b * size * size + pix] : 0.f;
    float z = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        float v = acc[o] * dcoef[o] + nz + bcoef[o];
        v = (v > 0.f ? v : 0.2f * v) * 1.4142135623730951f;
        z += lcoef[o] * v;
    }
    L.partial[((int64_t)j * L.batch + b) * size * size + pix] = z;
}

__global__ void att_finish_kernel(const float* __restrict__ partial, const float* __restrict__ d_last,
                                  const float* __restrict__ bias_last, const float* __restrict__ noise_last,
                                  const float* __restrict__ nw_last, const float* __restrict__ initial_bias,
                                  float* __restrict__ each, int n_sources, int batch, int npix) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)batch * npix) return;
    const int b = (int)(e / npix);
    float z = 0.f;
    for (int j = 0; j < n_sources; ++j) z += partial[(int64_t)j * batch * npix + e];  // fixed order: deterministic
    float v = z * d_last[b] + (noise_last ? nw_last[0] * noise_last[e] : 0.f) + bias_last[0];
    v = (v > 0.f ? v : 0.2f * v) * 1.4142135623730951f + initial_bias[0];
    each[e] = 1.f / (1.f + expf(-v));
}

// ---------------------------------------------------------------------------------------- cluster pooling
// One workgroup per sample.  Per-cluster sums by a fixed reduction tree (lanes -> wave shuffles -> 4 wave partials in
// LDS, added in wave order): bit-reproducible.  Then threshold and the separable 5-tap gaussian with reflect padding.
__global__ __launch_bounds__(256) void cluster_pool_kernel(const float* __restrict__ each, const int32_t* __restrict__ assign,
                                                           float* __restrict__ same, float* __restrict__ means,
                                                           float* __restrict__ counts, float* __restrict__ thr_out,
                                                           float* __restrict__ final_map, int size, int csize, int K,
                                                           float threshold) {
    extern __shared__ float lds[];  // [size*size] map, then [size*size] row-blurred
    __shared__ float part[4][2];
    __shared__ float kmean[32];
    const int b = blockIdx.x, npix = size * size;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* e = each + (int64_t)b * npix;
    const int32_t* a = assign + (int64_t)b * csize * csize;
    auto cluster_of = [&](int p) {
        const int y = p / size, x = p % size;
        return a[(int)(((int64_t)y * csize) / size) * csize + (int)(((int64_t)x * csize) / size)];
    };
    for (int k = 0; k < K; ++k) {
        float s = 0.f, n = 0.f;
        for (int p = threadIdx.x; p < npix; p += 256)
            if (cluster_of(p) == k) s += e[p], n += 1.f;
        s = wave_sum64(s), n = wave_sum64(n);
        if (lane == 0) part[wave][0] = s, part[wave][1] = n;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float ts = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
            const float tn = (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]);
            const float m = tn > 0.f ? ts / tn : 0.f;
            kmean[k] = m;
            means[b * K + k] = m, counts[b * K + k] = tn;
        }
        __syncthreads();
    }
    float* map = lds;
    float* tmp = lds + npix;
    for (int p = threadIdx.x; p < npix; p += 256) {
        const int k = cluster_of(p);
        const float v = (k >= 0 && k < K) ? kmean[k] : 1.f;  // same_attention_map starts as ones (:844)
        same[(int64_t)b * npix + p] = v;
        const float t = v < threshold ? 0.f : v;  // value of the straight-through form a - a.detach() (:882-883)
        if (thr_out) thr_out[(int64_t)b * npix + p] = t;
        map[p] = t;
    }
    __syncthreads();
    // torchvision gaussian_blur(kernel_size=5): sigma = 0.3*((5-1)*0.5-1)+0.8 = 1.1, taps exp(-0.5 (t/sigma)^2) normalised
    const float g0 = 1.f, g1 = expf(-0.5f * (1.f / 1.1f) * (1.f / 1.1f)), g2 = expf(-0.5f * (2.f / 1.1f) * (2.f / 1.1f));
    const float gs = g0 + 2.f * g1 + 2.f * g2;
    const float k0 = g0 / gs, k1 = g1 / gs, k2 = g2 / gs;
    auto refl = [&](int i) { return i < 0 ? -i : (i >= size ? 2 * size - 2 - i : i); };
    for (int p = threadIdx.x; p < npix; p += 256) {
        const int y = p / size, x = p % size;
        const float* r = map + y * size;
        tmp[p] = k2 * r[refl(x - 2)] + k1 * r[refl(x - 1)] + k0 * r[x] + k1 * r[refl(x + 1)] + k2 * r[refl(x + 2)];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npix; p += 256) {
        const int y = p / size, x = p % size;
        final_map[(int64_t)b * npix + p] = k2 * tmp[refl(y - 2) * size + x] + k1 * tmp[refl(y - 1) * size + x] + k0 * tmp[p] +
                                          k1 * tmp[refl(y + 1) * size + x] + k2 * tmp[refl(y + 2) * size + x];
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_cluster_assign(const float* feat, const float* centroids, int32_t* assign, int batch, int channels,
                                  int pos_channels, int size, int clusters, void* stream) {
    W2E_REQUIRE(feat && centroids && assign, "cluster_assign: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && pos_channels >= 0 && size > 1, "cluster_assign: bad dims");
    W2E_REQUIRE(clusters >= 1 && clusters <= 32, "cluster_assign: 1 <= clusters <= 32 (got %d)", clusters);
    if (batch == 0) return 0;
    const int kp = clusters <= 8 ? 8 : (clusters <= 16 ? 16 : 32);
    const size_t lds = sizeof(float) * ((size_t)(channels + 2 * pos_channels) * kp + (size_t)256 * kp);
    W2E_REQUIRE(lds <= 160 * 1024, "cluster_assign: centroids need %zu B of LDS", lds);
    dim3 grid((unsigned)ceil_div((int64_t)size * size, 64), (unsigned)batch);
    hipStream_t s = (hipStream_t)stream;
    static unsigned done[3] = {0, 0, 0};
    if (kp == 8) {
        if (lds > 64 * 1024) W2E_REQUIRE(big_lds_once((const void*)cluster_assign_kernel<8>, &done[0]), "cluster_assign: LDS opt-in failed");
        cluster_assign_kernel<8><<<grid, 256, lds, s>>>(feat, centroids, assign, channels, pos_channels, size, clusters);
    } else if (kp == 16) {
        if (lds > 64 * 1024) W2E_REQUIRE(big_lds_once((const void*)cluster_assign_kernel<16>, &done[1]), "cluster_assign: LDS opt-in failed");
        cluster_assign_kernel<16><<<grid, 256, lds, s>>>(feat, centroids, assign, channels, pos_channels, size, clusters);
    } else {
        if (lds > 64 * 1024) W2E_REQUIRE(big_lds_once((const void*)cluster_assign_kernel<32>, &done[2]), "cluster_assign: LDS opt-in failed");
        cluster_assign_kernel<32><<<grid, 256, lds, s>>>(feat, centroids, assign, channels, pos_channels, size, clusters);
    }
    W2E_LAUNCH_CHECK("cluster_assign");
    return 0;
}

extern "C" int w2e_cluster_accumulate(const float* feat, const int32_t* assign, float* partial, float* counts, int batch,
                                      int channels, int pos_channels, int size, int clusters, void* stream) {
    W2E_REQUIRE(feat && assign && partial && counts, "cluster_accumulate: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && pos_channels >= 0 && size > 1, "cluster_accumulate: bad dims");
    W2E_REQUIRE(clusters >= 1 && clusters <= 32, "cluster_accumulate: 1 <= clusters <= 32 (got %d)", clusters);
    if (batch == 0) return 0;
    dim3 grid((unsigned)(channels + 2 * pos_channels + 1), (unsigned)batch);
    hipStream_t s = (hipStream_t)stream;
    if (clusters <= 8) cluster_accumulate_kernel<8><<<grid, 256, 0, s>>>(feat, assign, partial, counts, channels, pos_channels, size, clusters);
    else if (clusters <= 16) cluster_accumulate_kernel<16><<<grid, 256, 0, s>>>(feat, assign, partial, counts, channels, pos_channels, size, clusters);
    else cluster_accumulate_kernel<32><<<grid, 256, 0, s>>>(feat, assign, partial, counts, channels, pos_channels, size, clusters);
    W2E_LAUNCH_CHECK("cluster_accumulate");
    return 0;
}

extern "C" int w2e_attention_demod(const w2e_att_source* sources, int n_sources, int batch, float eps, void* stream) {
    W2E_REQUIRE(sources, "attention_demod: null sources");
    W2E_REQUIRE(n_sources >= 1 && n_sources <= W2E_ATT_MAX_SOURCES, "attention_demod: 1 <= n_sources <= %d", W2E_ATT_MAX_SOURCES);
    W2E_REQUIRE(batch >= 0 && batch < 65536 && eps > 0.f, "attention_demod: bad dims");
    if (batch == 0) return 0;
    AttDemodLaunch L{};
    for (int j = 0; j < n_sources; ++j) {
        const w2e_att_source& s = sources[j];
        W2E_REQUIRE(s.wscaled && s.style && s.demod && s.channels > 0, "attention_demod: source %d: null tensor or bad dims", j);
        L.wscaled[j] = s.wscaled, L.style[j] = s.style, L.demod[j] = const_cast<float*>(s.demod), L.channels[j] = s.channels;
    }
    L.eps = eps;
    att_demod_kernel<<<dim3((unsigned)n_sources, (unsigned)batch), 256, 0, (hipStream_t)stream>>>(L);
    W2E_LAUNCH_CHECK("attention_demod");
    return 0;
}

extern "C" int w2e_attention_logits(const w2e_att_source* sources, int n_sources, const float* wlast, const float* s_last,
                                    const float* d_last, const float* bias_last, const float* noise_last,
                                    const float* nw_last, const float* initial_bias, float* partial, float* each, int batch,
                                    int size, void* stream) {
    W2E_REQUIRE(sources && wlast && s_last && d_last && bias_last && initial_bias && partial && each, "attention_logits: null tensor");
    W2E_REQUIRE(n_sources >= 1 && n_sources <= W2E_ATT_MAX_SOURCES, "attention_logits: 1 <= n_sources <= %d", W2E_ATT_MAX_SOURCES);
    W2E_REQUIRE(batch >= 0 && size > 0, "attention_logits: bad dims");
    W2E_REQUIRE(!noise_last || nw_last, "attention_logits: noise_last without nw_last");
    if (batch == 0) return 0;
    AttLaunch L{};
    for (int j = 0; j < n_sources; ++j) {
        const w2e_att_source& s = sources[j];
        W2E_REQUIRE(s.feat && s.wscaled && s.style && s.demod && s.bias, "attention_logits: source %d has a null tensor", j);
        W2E_REQUIRE(s.channels > 0 && s.res > 0, "attention_logits: source %d: bad dims", j);
        W2E_REQUIRE(!s.noise || s.noise_w, "attention_logits: source %d: noise without noise_w", j);
        L.src[j] = s;
    }
    L.wlast = wlast, L.s_last = s_last, L.partial = partial, L.n_sources = n_sources, L.batch = batch, L.size = size;
    hipStream_t s = (hipStream_t)stream;
    const int npix = size * size;
    dim3 grid((unsigned)ceil_div(npix, 256), (unsigned)batch, (unsigned)n_sources);
    att_source_kernel<<<grid, 256, 0, s>>>(L);
    W2E_LAUNCH_CHECK("attention_logits (sources)");
    att_finish_kernel<<<(unsigned)ceil_div((int64_t)batch * npix, 256), 256, 0, s>>>(partial, d_last, bias_last, noise_last, nw_last,
                                                                                    initial_bias, each, n_sources, batch, npix);
    W2E_LAUNCH_CHECK("attention_logits (finish)");
    return 0;
}

extern "C" int w2e_cluster_pool(const float* each, const int32_t* assign, float* same, float* means, float* counts,
                                float* thr, float* final_map, int batch, int size, int csize, int clusters, float threshold,
                                void* stream) {
    W2E_REQUIRE(each && assign && same && means && counts && final_map, "cluster_pool: null tensor");
    W2E_REQUIRE(batch >= 0 && size >= 3 && size <= 128 && csize > 0, "cluster_pool: 3 <= size <= 128 (got %d)", size);
    W2E_REQUIRE(clusters >= 1 && clusters <= 32, "cluster_pool: 1 <= clusters <= 32 (got %d)", clusters);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 2 * (size_t)size * size;
    static unsigned done = 0;
    if (lds > 64 * 1024) W2E_REQUIRE(big_lds_once((const void*)cluster_pool_kernel, &done), "cluster_pool: LDS opt-in failed");
    cluster_pool_kernel<<<batch, 256, lds, (hipStream_t)stream>>>(each, assign, same, means, counts, thr, final_map, size, csize,
                                                                 clusters, threshold);
    W2E_LAUNCH_CHECK("cluster_pool");
    return 0;
}
