// K1r: ToRGB for gfx950 -- modulated 1x1 conv to 3 channels (no demodulation) + bias + FIR-upsampled
// skip, in one pass over the feature map (replaces models/stylegan2/model.py:343-362 = grouped 1x1
// F.conv2d + bias add + upfirdn2d(up=2) + add).  N=3 is a per-pixel dot product, HBM-bound (the
// feature map is read exactly once), so this is a wavefront-FMA kernel, not MFMA.
#include "common.h"

namespace w2e {

// y[b,c,p] = sum_i wmod[b,c,i] * x[b,i,p] + bias[c] + up2(skip)[b,c,p]
// Block = PQ pixel groups x CG channel groups (PQ*CG = 256); V pixels per thread (4 = float4).
// `style` != null: wmod is the shared [3][cin] scale*W and the per-sample weight wmod[c,i]*style[b,i] is formed here
// (model.py:239 with k = 1) instead of by a [B,3,cin] elementwise launch.
template <int CG, int V>
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wmod,
                                                        const float* __restrict__ style,
                                                        const float* __restrict__ bias, const float* __restrict__ skip,
                                                        const float* __restrict__ upk, float* __restrict__ y, int cin,
                                                        int H, int W) {
    constexpr int PQ = 256 / CG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wsm = smem;                    // [3][cin]
    float* red = smem + 3 * cin;          // [CG][PQ][3*V]   (CG > 1)
    __shared__ float kf[16];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    if (style) {
        for (int i = tid; i < cin; i += 256) {
            const float st = style[(int64_t)b * cin + i];
            wsm[i] = wmod[i] * st, wsm[cin + i] = wmod[cin + i] * st, wsm[2 * cin + i] = wmod[2 * cin + i] * st;
        }
    } else {
        for (int i = tid; i < 3 * cin; i += 256) wsm[i] = wmod[(int64_t)b * 3 * cin + i];
    }
    if (tid < 16 && skip) kf[tid] = upk[15 - tid];  // flipped: true convolution (op/upfirdn2d.py:47)
    __syncthreads();
    const int cg = tid / PQ, pq = tid % PQ;
    const int64_t p0 = ((int64_t)blockIdx.x * PQ + pq) * V;
    float acc[3][V];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] = 0.f;
    if (p0 < HW) {
        const float* xp = x + (int64_t)b * cin * HW + p0;
#pragma unroll 4
        for (int i = cg; i < cin; i += CG) {
            float xv[V];
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xp + i * HW);
                xv[0] = t.x, xv[1] = t.y, xv[2] = t.z, xv[3] = t.w;
            } else {
                xv[0] = xp[i * HW];
            }
            const float w0 = wsm[i], w1 = wsm[cin + i], w2 = wsm[2 * cin + i];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[0][v] += w0 * xv[v], acc[1][v] += w1 * xv[v], acc[2][v] += w2 * xv[v];
        }
    }
    if (CG > 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) red[(cg * PQ + pq) * 3 * V + c * V + v] = acc[c][v];
        __syncthreads();
        if (cg != 0) return;
        for (int g = 1; g < CG; ++g)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[c][v] += red[(g * PQ + pq) * 3 * V + c * V + v];
    }
    if (p0 >= HW) return;
    const int Hs = H >> 1, Ws = W >> 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float bs = bias ? bias[c] : 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float r = acc[c][v] + bs;
            if (skip) {
                // Upsample: zero-stuff x2, pad (2,1), 4x4 kernel (model.py:31-49).  Only taps with
                // (y+ky-2) and (x+kx-2) even hit a sample: 2x2 of the 16.
                const int py = (int)((p0 + v) / W), px = (int)((p0 + v) % W);
                const float* sp = skip + ((int64_t)b * 3 + c) * Hs * Ws;
                float u = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ky = (py & 1) + 2 * a, sy = (py + ky - 2) >> 1;
                    if (py + ky - 2 < 0 || sy >= Hs) continue;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int kx = (px & 1) + 2 * d, sx = (px + kx - 2) >> 1;
                        if (px + kx - 2 < 0 || sx >= Ws) continue;
                        u += kf[ky * 4 + kx] * sp[sy * Ws + sx];
                    }
                }
                r += u;
            }
            acc[c][v] = r;
        }
        float* dst = y + ((int64_t)b * 3 + c) * HW + p0;
        if (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
        else dst[0] = acc[c][0];
    }
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave per (b, channel i, pixel split): gx[b,i,p] = sum_c wmod[b,c,i]*gy[b,c,p] and
// gwmod[b,c,i] (+)= sum_p x[b,i,p]*gy[b,c,p].  The 4 waves of a block take 4 consecutive channels of the
// same pixel range so their gy reads share L1.  ACC: gx = gx_acc + (that sum) -- the gradient that reached x through its
// other consumer (the next conv) is folded in here instead of by a separate elementwise add of two activation-sized tensors.
// `style` != null (see the forward): wmod is [3][cin] and gwmod is the STYLE gradient [B][cin] = sum_c wmod[c,i]*(that sum).
// ACTB: x is the OUTPUT of the fused StyledConv that feeds this ToRGB (bias + noise + LeakyReLU * gain, model.py:334-340), and what
// that layer's backward needs is not gx but gpre = gx * gain * (x > 0 ? 1 : slope) together with the three per-(b, channel) sums
// of w2e_bias_act_bwd_reduce -- computed here, on the value that is in a register anyway: the layer's separate activation
// backward pass (read gx, read x, write gpre) disappears.
template <int V, bool ACC, bool ACTB>
__global__ __launch_bounds__(256) void torgb_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wmod,
                                                        const float* __restrict__ style,
                                                        const float* __restrict__ gy, const float* __restrict__ gx_acc,
                                                        float* __restrict__ gx, float* __restrict__ gwmod, int cin,
                                                        int64_t HW, int splits, int64_t per_split,
                                                        const float* __restrict__ noise, float* __restrict__ sums3, float slope, float gain) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x % splits;
    const int i = (blockIdx.x / splits) * 4 + wave;
    const int b = blockIdx.y;
    if (i >= cin) return;
    float w0, w1, w2;
    if (style) {
        const float st = style[(int64_t)b * cin + i];
        w0 = wmod[i] * st, w1 = wmod[cin + i] * st, w2 = wmod[2 * cin + i] * st;
    } else {
        w0 = wmod[((int64_t)b * 3 + 0) * cin + i], w1 = wmod[((int64_t)b * 3 + 1) * cin + i],
        w2 = wmod[((int64_t)b * 3 + 2) * cin + i];
    }
    const float* xp = x + ((int64_t)b * cin + i) * HW;
    float* gp = gx + ((int64_t)b * cin + i) * HW;
    const float* ap = ACC ? gx_acc + ((int64_t)b * cin + i) * HW : nullptr;
    const float* g0 = gy + (int64_t)b * 3 * HW;
    const int64_t lo = split * per_split, hi = (lo + per_split < HW) ? lo + per_split : HW;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float t_pre = 0.f, t_noise = 0.f, t_sum = 0.f;  // ACTB
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * slope);
    auto act_bwd = [&](float r, float xval, float nz) __attribute__((always_inline)) {
        const bool pos = xval > 0.f;
        const float g = r * gain * (pos ? 1.f : slope);
        t_pre += g * (xval * (pos ? inv_pos : inv_neg)), t_noise += g * nz, t_sum += g;
        return g;
    };
    if (V == 4) {
#pragma unroll 2
        for (int64_t p = lo + lane * 4; p < hi; p += 256) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + p);
            const float4 a = *reinterpret_cast<const float4*>(g0 + p);
            const float4 bb = *reinterpret_cast<const float4*>(g0 + HW + p);
            const float4 c = *reinterpret_cast<const float4*>(g0 + 2 * HW + p);
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ACTB && noise) nz = *reinterpret_cast<const float4*>(noise + p);
            float4 r;
            r.x = w0 * a.x + w1 * bb.x + w2 * c.x;
            r.y = w0 * a.y + w1 * bb.y + w2 * c.y;
            r.z = w0 * a.z + w1 * bb.z + w2 * c.z;
            r.w = w0 * a.w + w1 * bb.w + w2 * c.w;
            if (ACC) {
                const float4 av = *reinterpret_cast<const float4*>(ap + p);
                r.x += av.x, r.y += av.y, r.z += av.z, r.w += av.w;
            }
            if (ACTB) r.x = act_bwd(r.x, xv.x, nz.x), r.y = act_bwd(r.y, xv.y, nz.y), r.z = act_bwd(r.z, xv.z, nz.z), r.w = act_bwd(r.w, xv.w, nz.w);
            *reinterpret_cast<float4*>(gp + p) = r;
            s0 += xv.x * a.x + xv.y * a.y + xv.z * a.z + xv.w * a.w;
            s1 += xv.x * bb.x + xv.y * bb.y + xv.z * bb.z + xv.w * bb.w;
            s2 += xv.x * c.x + xv.y * c.y + xv.z * c.z + xv.w * c.w;
        }
    } else {
        for (int64_t p = lo + lane; p < hi; p += 64) {
            const float xv = xp[p], a = g0[p], bb = g0[HW + p], c = g0[2 * HW + p];
            float r = w0 * a + w1 * bb + w2 * c + (ACC ? ap[p] : 0.f);
            if (ACTB) r = act_bwd(r, xv, noise ? noise[p] : 0.f);
            gp[p] = r;
            s0 += xv * a, s1 += xv * bb, s2 += xv * c;
        }
    }
    if (ACTB) {
        t_pre = wave_sum64(t_pre), t_noise = wave_sum64(t_noise), t_sum = wave_sum64(t_sum);
        if (lane == 0) {
            float* d3 = sums3 + ((int64_t)b * cin + i) * 3;
            if (splits == 1) d3[0] = t_pre, d3[1] = t_noise, d3[2] = t_sum;
            else atomicAdd(d3, t_pre), atomicAdd(d3 + 1, t_noise), atomicAdd(d3 + 2, t_sum);
        }
    }
    s0 = wave_sum64(s0), s1 = wave_sum64(s1), s2 = wave_sum64(s2);
    if (lane == 0) {
        if (style) {
            const float g = wmod[i] * s0 + wmod[cin + i] * s1 + wmod[2 * cin + i] * s2;
            float* d = gwmod + (int64_t)b * cin + i;
            if (splits == 1) d[0] = g;
            else atomicAdd(d, g);
        } else {
            float* d = gwmod + (int64_t)b * 3 * cin + i;
            if (splits == 1) d[0] = s0, d[cin] = s1, d[2 * cin] = s2;
            else atomicAdd(d, s0), atomicAdd(d + cin, s1), atomicAdd(d + 2 * cin, s2);
        }
    }
}

}  // namespace w2e

using namespace w2e;

static int torgb_fwd_impl(const float* x, const float* wmod, const float* style, const float* bias, const float* skip,
                          const float* upk, float* y, int batch, int cin, int h, int w, void* stream) {
    W2E_REQUIRE(x && wmod && y, "torgb_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && cin > 0 && h > 0 && w > 0, "torgb_fwd: bad dims");
    W2E_REQUIRE(!skip || (upk && (h % 2 == 0) && (w % 2 == 0)), "torgb_fwd: skip needs the 4x4 kernel and even h,w");
    W2E_REQUIRE(batch < 65536, "torgb_fwd: batch too large");
    if (batch == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)h * w;
    const bool vec = (hw & 3) == 0;
    const int v = vec ? 4 : 1;
    const int64_t groups = ceil_div(hw, v);
    // low resolutions: split the channel loop over 4 or 16 thread groups to shorten the dependent chain
    const int cg = groups >= 16384 ? 1 : (groups >= 1024 ? 4 : 16);
    const int pq = 256 / cg;
    dim3 grid((unsigned)ceil_div(groups, pq), (unsigned)batch);
    const size_t lds = sizeof(float) * ((size_t)3 * cin + (cg > 1 ? (size_t)256 * 3 * v : 0));
#define W2E_TORGB(CG, V) torgb_fwd_kernel<CG, V><<<grid, 256, lds, s>>>(x, wmod, style, bias, skip, upk, y, cin, h, w)
    if (vec) {
        if (cg == 1) W2E_TORGB(1, 4);
        else if (cg == 4) W2E_TORGB(4, 4);
        else W2E_TORGB(16, 4);
    } else {
        if (cg == 1) W2E_TORGB(1, 1);
        else if (cg == 4) W2E_TORGB(4, 1);
        else W2E_TORGB(16, 1);
    }
#undef W2E_TORGB
    W2E_LAUNCH_CHECK("torgb_fwd");
    return 0;
}

extern "C" int w2e_torgb_fwd(const float* x, const float* wmod, const float* bias, const float* skip, const float* upk,
                             float* y, int batch, int cin, int h, int w, void* stream) {
    return torgb_fwd_impl(x, wmod, nullptr, bias, skip, upk, y, batch, cin, h, w, stream);
}

extern "C" int w2e_torgb_styled_fwd(const float* x, const float* wsc, const float* style, const float* bias, const float* skip,
                                    const float* upk, float* y, int batch, int cin, int h, int w, void* stream) {
    W2E_REQUIRE(style, "torgb_styled_fwd: null style");
    return torgb_fwd_impl(x, wsc, style, bias, skip, upk, y, batch, cin, h, w, stream);
}

static int torgb_bwd_impl(const float* x, const float* wmod, const float* style, const float* gy, const float* gx_acc,
                          float* gx, float* gwmod, int batch, int cin, int h, int w, void* stream, bool actb = false,
                          const float* noise = nullptr, float* sums3 = nullptr, float slope = 0.f, float gain = 1.f);

extern "C" int w2e_torgb_bwd(const float* x, const float* wmod, const float* gy, float* gx, float* gwmod, int batch,
                             int cin, int h, int w, void* stream) {
    return torgb_bwd_impl(x, wmod, nullptr, gy, nullptr, gx, gwmod, batch, cin, h, w, stream);
}

extern "C" int w2e_torgb_bwd_acc(const float* x, const float* wmod, const float* gy, const float* gx_acc, float* gx,
                                 float* gwmod, int batch, int cin, int h, int w, void* stream) {
    return torgb_bwd_impl(x, wmod, nullptr, gy, gx_acc, gx, gwmod, batch, cin, h, w, stream);
}

extern "C" int w2e_torgb_styled_bwd(const float* x, const float* wsc, const float* style, const float* gy,
                                    const float* gx_acc, float* gx, float* gstyle, int batch, int cin, int h, int w,
                                    void* stream) {
    W2E_REQUIRE(style, "torgb_styled_bwd: null style");
    return torgb_bwd_impl(x, wsc, style, gy, gx_acc, gx, gstyle, batch, cin, h, w, stream);
}

extern "C" int w2e_torgb_bwd_actbwd(const float* x, const float* wmod, const float* style, const float* gy, const float* gx_acc,
                                    const float* noise, float* gpre, float* gw, float* sums3, int batch, int cin, int h, int w,
                                    float slope, float gain, void* stream) {
    W2E_REQUIRE(sums3, "torgb_bwd_actbwd: null sums");
    W2E_REQUIRE(gain > 0.f && slope > 0.f, "torgb_bwd_actbwd: gain and slope must be positive");
    return torgb_bwd_impl(x, wmod, style, gy, gx_acc, gpre, gw, batch, cin, h, w, stream, true, noise, sums3, slope, gain);
}

static int torgb_bwd_impl(const float* x, const float* wmod, const float* style, const float* gy, const float* gx_acc,
                          float* gx, float* gwmod, int batch, int cin, int h, int w, void* stream, bool actb, const float* noise,
                          float* sums3, float slope, float gain) {
    W2E_REQUIRE(x && wmod && gy && gx && gwmod, "torgb_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && cin > 0 && h > 0 && w > 0, "torgb_bwd: bad dims");
    W2E_REQUIRE(batch < 65536, "torgb_bwd: batch too large");
    if (batch == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)h * w;
    const int64_t waves = (int64_t)batch * cin;
    int splits = 1;
    if (waves < 8192 && !options().deterministic) {  // deterministic: one wave owns a (b, channel)'s weight gradient
        splits = (int)ceil_div(8192, waves);
        const int64_t max_splits = ceil_div(hw, 4096);
        if (splits > max_splits) splits = (int)max_splits;
    }
    int64_t per_split = ceil_div(hw, splits);
    per_split = (per_split + 255) & ~int64_t(255);
    splits = (int)ceil_div(hw, per_split);
    if (splits > 1 && (zero_async(gwmod, sizeof(float) * (style ? 1 : 3) * (size_t)batch * cin, s) != hipSuccess ||
                       (actb && zero_async(sums3, sizeof(float) * 3 * (size_t)batch * cin, s) != hipSuccess))) {
        set_error("torgb_bwd: memset failed");
        return 2;
    }
    dim3 grid((unsigned)(ceil_div(cin, 4) * splits), (unsigned)batch);
#define W2E_TORGB_BWD(V, ACC, ACTB) \
    torgb_bwd_kernel<V, ACC, ACTB><<<grid, 256, 0, s>>>(x, wmod, style, gy, gx_acc, gx, gwmod, cin, hw, splits, per_split, noise, sums3, slope, gain)
    if ((hw & 3) == 0) {
        if (actb) {
            if (gx_acc) W2E_TORGB_BWD(4, true, true);
            else W2E_TORGB_BWD(4, false, true);
        } else {
            if (gx_acc) W2E_TORGB_BWD(4, true, false);
            else W2E_TORGB_BWD(4, false, false);
        }
    } else {
        if (actb) {
            if (gx_acc) W2E_TORGB_BWD(1, true, true);
            else W2E_TORGB_BWD(1, false, true);
        } else {
            if (gx_acc) W2E_TORGB_BWD(1, true, false);
            else W2E_TORGB_BWD(1, false, false);
        }
    }
#undef W2E_TORGB_BWD
    W2E_LAUNCH_CHECK("torgb_bwd");
    return 0;
}
