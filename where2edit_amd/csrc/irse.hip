// Elementwise / reduction kernels of the IR-SE50 body (include/w2e_irse.h): folded BatchNorm + PReLU, the SE block's
// pooling and gating, the residual joins and their adjoints.  All HBM-bound, grid-strided; the 3x3 convolutions run on
// the MFMA engine of modconv.hip (w2e_conv3x3).
#include "../../include/w2e_irse.h"
#include "common.h"

namespace w2e {

__global__ void affine_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ slope, float* __restrict__ y, int C, int64_t hw, int64_t total4) {
    // hw % 4 == 0: float4 path (total4 = total / 4); one (b,c) plane never splits a float4
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += step) {
        const int c = (int)(((q << 2) / hw) % C);
        const float av = a ? a[c] : 1.f, bv = b ? b[c] : 0.f, sv = slope ? slope[c] : 1.f;
        float4 v = reinterpret_cast<const float4*>(x)[q];
        v.x = v.x * av + bv, v.y = v.y * av + bv, v.z = v.z * av + bv, v.w = v.w * av + bv;
        v.x = v.x > 0.f ? v.x : sv * v.x, v.y = v.y > 0.f ? v.y : sv * v.y, v.z = v.z > 0.f ? v.z : sv * v.z, v.w = v.w > 0.f ? v.w : sv * v.w;
        reinterpret_cast<float4*>(y)[q] = v;
    }
}

__global__ void affine_act_fwd_scalar_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ slope, float* __restrict__ y, int C, int64_t hw, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int c = (int)((e / hw) % C);
        float v = x[e] * (a ? a[c] : 1.f) + (b ? b[c] : 0.f);
        y[e] = v > 0.f ? v : (slope ? slope[c] : 1.f) * v;
    }
}

// gx = a*gy*(y>0 ? 1 : slope); gy plain or read through the (+1,+1) crop of a phase-planar UP output
__global__ void affine_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ a,
                                      const float* __restrict__ slope, float* __restrict__ gx, int C, int H, int W, int planar,
                                      int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const int64_t hw = (int64_t)H * W;
    const int hp = H / 2 + 1, wp = W2E_PLANAR_PITCH(W / 2);  // phase planes of T: [(H/2)+1][WP]
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t plane = e / hw;
        const int c = (int)(plane % C);
        float g;
        if (planar) {
            const int p = (int)(e - plane * hw);
            const int yy = p / W + 1, xx = p % W + 1;
            g = gy[((plane * 4 + (yy & 1) * 2 + (xx & 1)) * hp + (yy >> 1)) * wp + (xx >> 1)];
        } else {
            g = gy[e];
        }
        g *= a ? a[c] : 1.f;
        if (y) g *= y[e] > 0.f ? 1.f : (slope ? slope[c] : 1.f);
        gx[e] = g;
    }
}

__device__ __forceinline__ float wave_sum_irse(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wave per (b,c) plane, 4 planes per block; lanes stride the plane (float4 when aligned)
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           float* __restrict__ sums, int64_t planes, int64_t hw) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* xp = x + plane * hw;
    const float* yp = y ? y + plane * hw : nullptr;
    float acc = 0.f;
    if ((hw & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const float4* y4 = reinterpret_cast<const float4*>(yp);
        for (int64_t q = lane; q < (hw >> 2); q += 64) {
            const float4 v = x4[q];
            if (yp) {
                const float4 w = y4[q];
                acc += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
            } else {
                acc += (v.x + v.y) + (v.z + v.w);
            }
        }
    } else {
        for (int64_t p = lane; p < hw; p += 64) acc += yp ? xp[p] * yp[p] : xp[p];
    }
    acc = wave_sum_irse(acc);
    if (lane == 0) sums[plane] = acc;
}

// The SE block's gate (helpers.py:56-72) on the pooled [B,C] sums: mean -> fc1 (1x1 conv, no bias) -> ReLU -> fc2 -> sigmoid.
// One workgroup per sample: C <= a few hundred channels, R = C/16 hidden units -- one launch instead of the mean /
// two rocBLAS GEMMs of B rows / ReLU / sigmoid.  `hidden` keeps the ReLU output for the backward.
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* __restrict__ sums, const float* __restrict__ fc1,
                                                          const float* __restrict__ fc2, float* __restrict__ gate,
                                                          float* __restrict__ hidden, int C, int R, float inv_hw) {
    extern __shared__ float se_sm[];
    float* pooled = se_sm;    // [C]
    float* hid = se_sm + C;   // [R]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) pooled[c] = sums[(int64_t)b * C + c] * inv_hw;
    __syncthreads();
    for (int r = wave; r < R; r += 4) {  // one wave per hidden unit, fixed reduction order
        const float* w = fc1 + (int64_t)r * C;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w[c] * pooled[c];
        acc = wave_sum_irse(acc);
        if (lane == 0) {
            acc = acc > 0.f ? acc : 0.f;
            hid[r] = acc;
            hidden[(int64_t)b * R + r] = acc;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float* w = fc2 + (int64_t)c * R;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += w[r] * hid[r];
        gate[(int64_t)b * C + c] = 1.f / (1.f + expf(-acc));
    }
}

// Its adjoint: dgate[b,c] = sum_p gout*t (w2e_channel_sums) -> through the sigmoid, fc2, the ReLU and fc1 -> the gradient
// at the pooled mean, divided by H*W (what every pixel of the plane receives, w2e_se_apply_bwd's gpool).
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ hidden, const float* __restrict__ fc1,
                                                          const float* __restrict__ fc2, float* __restrict__ gpool, int C, int R,
                                                          float inv_hw) {
    extern __shared__ float se_sm[];
    float* d2 = se_sm;       // [C] gradient at the sigmoid's input
    float* dh = se_sm + C;   // [R] gradient at the ReLU's input
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        const float g = gate[(int64_t)b * C + c];
        d2[c] = dgate[(int64_t)b * C + c] * g * (1.f - g);
    }
    __syncthreads();
    for (int r = wave; r < R; r += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += fc2[(int64_t)c * R + r] * d2[c];
        acc = wave_sum_irse(acc);
        if (lane == 0) dh[r] = hidden[(int64_t)b * R + r] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += fc1[(int64_t)r * C + c] * dh[r];
        gpool[(int64_t)b * C + c] = acc * inv_hw;
    }
}

__global__ void se_apply_fwd_kernel(const float* __restrict__ t, const float* __restrict__ gate, const float* __restrict__ sc,
                                    int sc_stride, float* __restrict__ out, int H, int W, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const int64_t hw = (int64_t)H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t plane = e / hw;
        float s;
        if (sc_stride == 0) {
            s = sc[e];
        } else {
            const int p = (int)(e - plane * hw);
            const int yy = p / W, xx = p % W;
            s = sc[(plane * (int64_t)H * sc_stride + (int64_t)yy * sc_stride) * ((int64_t)W * sc_stride) + (int64_t)xx * sc_stride];
        }
        out[e] = t[e] * gate[plane] + s;
    }
}

__global__ void se_apply_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ gate, const float* __restrict__ gpool,
                                    float* __restrict__ g_t, int64_t hw, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t plane = e / hw;
        g_t[e] = gout[e] * gate[plane] + gpool[plane];
    }
}

// gx (H*s x W*s per plane) += g (H x W per plane) at the strided positions; or the planar crop form (stride 1 over gx)
__global__ void shortcut_add_bwd_kernel(float* __restrict__ gx, const float* __restrict__ g, int H, int W, int stride, int planar,
                                        int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const int64_t hw = (int64_t)H * W;
    const int hp = H / 2 + 1, wp = W2E_PLANAR_PITCH(W / 2);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t plane = e / hw;
        const int p = (int)(e - plane * hw);
        const int yy = p / W, xx = p % W;
        if (planar) {
            const int y1 = yy + 1, x1 = xx + 1;
            gx[e] += g[((plane * 4 + (y1 & 1) * 2 + (x1 & 1)) * hp + (y1 >> 1)) * wp + (x1 >> 1)];
        } else {
            gx[(plane * (int64_t)H * stride + (int64_t)yy * stride) * ((int64_t)W * stride) + (int64_t)xx * stride] += g[e];
        }
    }
}

}  // namespace w2e

using namespace w2e;


// FPN merge of the pSp / e4e encoders (models/encoders/helpers.py:123-140): out = bilinear(x -> (oh, ow), align_corners=True) + y.
// One thread per output element: the four source values come from the small (cache-resident) map, y and out stream once.
namespace w2e {
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                                          int ih, int iw, int oh, int ow, float sy, float sx, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int ox = (int)(e % ow);
        const int64_t t = e / ow;
        const int oy = (int)(t % oh);
        const int64_t plane = t / oh;
        const float fy = oy * sy, fx = ox * sx;  // align_corners: src = dst * (in - 1) / (out - 1)
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + 1 < ih ? y0 + 1 : ih - 1, x1 = x0 + 1 < iw ? x0 + 1 : iw - 1;
        const float wy = fy - y0, wx = fx - x0;
        const float* xp = x + plane * ih * iw;
        const float top = xp[y0 * iw + x0] * (1.f - wx) + xp[y0 * iw + x1] * wx;
        const float bot = xp[y1 * iw + x0] * (1.f - wx) + xp[y1 * iw + x1] * wx;
        out[e] = top * (1.f - wy) + bot * wy + y[e];
    }
}
}  // namespace w2e

extern "C" {

int w2e_affine_act_fwd(const float* x, const float* a, const float* b, const float* slope, float* y, int batch, int channels,
                       int64_t hw, void* stream) {
    W2E_REQUIRE(x && y, "affine_act_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && hw > 0, "affine_act_fwd: bad dims");
    const int64_t total = (int64_t)batch * channels * hw;
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((hw & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
        affine_act_fwd_kernel<<<stream_grid(total >> 2, 256), 256, 0, s>>>(x, a, b, slope, y, channels, hw, total >> 2);
    else
        affine_act_fwd_scalar_kernel<<<stream_grid(total, 256), 256, 0, s>>>(x, a, b, slope, y, channels, hw, total);
    W2E_LAUNCH_CHECK("affine_act_fwd");
    return 0;
}

int w2e_affine_act_bwd(const float* gy, const float* y, const float* a, const float* slope, float* gx, int batch, int channels,
                       int height, int width, int planar, void* stream) {
    W2E_REQUIRE(gy && gx, "affine_act_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && height > 0 && width > 0, "affine_act_bwd: bad dims");
    W2E_REQUIRE(!planar || ((height & 1) == 0 && (width & 1) == 0), "affine_act_bwd: the planar crop needs even sizes");
    // planar: 0 = gy is dense [B,C,H,W]; otherwise the ROW PITCH of the phase-planar gy the caller allocated -- it must be this library's
    W2E_REQUIRE(!planar || planar == W2E_PLANAR_PITCH(width / 2),
                "affine_act_bwd: planar gradient with a row pitch of %d floats, this library's layout has %d (W2E_PLANAR_PITCH, ABI %d): rebuild the caller",
                planar, W2E_PLANAR_PITCH(width / 2), W2E_VERSION);
    const int64_t total = (int64_t)batch * channels * height * width;
    if (total == 0) return 0;
    affine_act_bwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(gy, y, a, slope, gx, channels, height, width,
                                                                                    planar != 0, total);
    W2E_LAUNCH_CHECK("affine_act_bwd");
    return 0;
}

int w2e_channel_sums(const float* x, const float* y, float* sums, int batch, int channels, int64_t hw, void* stream) {
    W2E_REQUIRE(x && sums, "channel_sums: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && hw > 0, "channel_sums: bad dims");
    const int64_t planes = (int64_t)batch * channels;
    if (planes == 0) return 0;
    W2E_REQUIRE(ceil_div(planes, 4) < ((int64_t)1 << 31), "channel_sums: too many planes");
    const bool al = (((uintptr_t)x | (uintptr_t)(y ? y : x)) & 15) == 0;
    W2E_REQUIRE(al || (hw & 3), "channel_sums: float4 path needs 16-byte aligned tensors");
    channel_sums_kernel<<<(unsigned)ceil_div(planes, 4), 256, 0, (hipStream_t)stream>>>(x, y, sums, planes, hw);
    W2E_LAUNCH_CHECK("channel_sums");
    return 0;
}

int w2e_se_gate_fwd(const float* sums, const float* fc1, const float* fc2, float* gate, float* hidden, int batch, int channels,
                    int reduced, float inv_hw, void* stream) {
    W2E_REQUIRE(sums && fc1 && fc2 && gate && hidden, "se_gate_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && reduced > 0 && channels + reduced <= 8192, "se_gate_fwd: bad dims");
    if (batch == 0) return 0;
    se_gate_fwd_kernel<<<(unsigned)batch, 256, sizeof(float) * (size_t)(channels + reduced), (hipStream_t)stream>>>(
        sums, fc1, fc2, gate, hidden, channels, reduced, inv_hw);
    W2E_LAUNCH_CHECK("se_gate_fwd");
    return 0;
}

int w2e_se_gate_bwd(const float* dgate, const float* gate, const float* hidden, const float* fc1, const float* fc2, float* gpool,
                    int batch, int channels, int reduced, float inv_hw, void* stream) {
    W2E_REQUIRE(dgate && gate && hidden && fc1 && fc2 && gpool, "se_gate_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && reduced > 0 && channels + reduced <= 8192, "se_gate_bwd: bad dims");
    if (batch == 0) return 0;
    se_gate_bwd_kernel<<<(unsigned)batch, 256, sizeof(float) * (size_t)(channels + reduced), (hipStream_t)stream>>>(
        dgate, gate, hidden, fc1, fc2, gpool, channels, reduced, inv_hw);
    W2E_LAUNCH_CHECK("se_gate_bwd");
    return 0;
}

int w2e_se_apply_fwd(const float* t, const float* gate, const float* shortcut, int sc_stride, float* out, int batch, int channels,
                     int height, int width, void* stream) {
    W2E_REQUIRE(t && gate && shortcut && out, "se_apply_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && height > 0 && width > 0 && sc_stride >= 0, "se_apply_fwd: bad dims");
    const int64_t total = (int64_t)batch * channels * height * width;
    if (total == 0) return 0;
    se_apply_fwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(t, gate, shortcut, sc_stride, out, height, width, total);
    W2E_LAUNCH_CHECK("se_apply_fwd");
    return 0;
}

int w2e_se_apply_bwd(const float* gout, const float* gate, const float* gpool, float* g_t, int batch, int channels, int64_t hw,
                     void* stream) {
    W2E_REQUIRE(gout && gate && gpool && g_t, "se_apply_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && hw > 0, "se_apply_bwd: bad dims");
    const int64_t total = (int64_t)batch * channels * hw;
    if (total == 0) return 0;
    se_apply_bwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(gout, gate, gpool, g_t, hw, total);
    W2E_LAUNCH_CHECK("se_apply_bwd");
    return 0;
}

int w2e_shortcut_add_bwd(float* gx, const float* g, int batch, int channels, int height, int width, int stride, int planar,
                         void* stream) {
    W2E_REQUIRE(gx && g, "shortcut_add_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && height > 0 && width > 0 && stride >= 1, "shortcut_add_bwd: bad dims");
    W2E_REQUIRE(!planar || (stride == 1 && (height & 1) == 0 && (width & 1) == 0), "shortcut_add_bwd: planar form is stride 1, even sizes");
    W2E_REQUIRE(!planar || planar == W2E_PLANAR_PITCH(width / 2),
                "shortcut_add_bwd: planar gradient with a row pitch of %d floats, this library's layout has %d (W2E_PLANAR_PITCH, ABI %d): rebuild the caller",
                planar, W2E_PLANAR_PITCH(width / 2), W2E_VERSION);
    const int64_t total = (int64_t)batch * channels * height * width;
    if (total == 0) return 0;
    shortcut_add_bwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(gx, g, height, width, stride, planar != 0, total);
    W2E_LAUNCH_CHECK("shortcut_add_bwd");
    return 0;
}

int w2e_upsample_add(const float* x, const float* y, float* out, int64_t planes, int ih, int iw, int oh, int ow, void* stream) {
    W2E_REQUIRE(x && y && out, "upsample_add: null tensor");
    W2E_REQUIRE(planes >= 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "upsample_add: bad dims");
    const int64_t total = planes * oh * ow;
    if (total == 0) return 0;
    const float sy = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    w2e::upsample_add_kernel<<<w2e::stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, y, out, ih, iw, oh, ow, sy, sx, total);
    W2E_LAUNCH_CHECK("upsample_add");
    return 0;
}

}  // extern "C"
