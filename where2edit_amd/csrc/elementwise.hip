// HBM-bound kernels of the Where2edit hot path for gfx950: fused bias/noise/LeakyReLU (K3),
// region-attention blend (K6), CLIP preprocessing (K5).  Vectorised 16 B/lane, grid-strided.
#include "common.h"

namespace w2e {

// ------------------------------------------------------------------------------------- K3
// x as [outer, C, inner].  VEC4 requires (inner % 4 == 0) or (inner == 1 && C % 4 == 0).
template <bool VEC4>
__global__ void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                    const float* __restrict__ noise, const float* __restrict__ noise_w,
                                    float* __restrict__ y, int64_t total, int64_t C, int64_t inner, float slope,
                                    float gain) {
    const float nw = (noise != nullptr) ? noise_w[0] : 0.f;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    if (VEC4) {
        const int64_t n4 = total >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += step) {
            const int64_t e = q << 2;
            float4 v = reinterpret_cast<const float4*>(x)[q];
            float add[4] = {0.f, 0.f, 0.f, 0.f};
            if (inner == 1) {
                const int64_t c = e % C;
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4*>(bias + c);
                    add[0] = b.x, add[1] = b.y, add[2] = b.z, add[3] = b.w;
                }
                if (noise) {
                    const float nz = nw * noise[0];
                    add[0] += nz, add[1] += nz, add[2] += nz, add[3] += nz;
                }
            } else {
                const int64_t i = e % inner;
                const float b = bias ? bias[(e / inner) % C] : 0.f;
                add[0] = add[1] = add[2] = add[3] = b;
                if (noise) {
                    const float4 nz = *reinterpret_cast<const float4*>(noise + i);
                    add[0] += nw * nz.x, add[1] += nw * nz.y, add[2] += nw * nz.z, add[3] += nw * nz.w;
                }
            }
            float r[4] = {v.x + add[0], v.y + add[1], v.z + add[2], v.w + add[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (r[k] > 0.f ? r[k] : r[k] * slope) * gain;
            reinterpret_cast<float4*>(y)[q] = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
            const int64_t i = e % inner;
            float v = x[e] + (bias ? bias[(e / inner) % C] : 0.f) + (noise ? nw * noise[i] : 0.f);
            y[e] = (v > 0.f ? v : v * slope) * gain;
        }
    }
}

template <bool VEC4>
__global__ void bias_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                                    int64_t total, float slope, float gain) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    if (VEC4) {
        const int64_t n4 = total >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += step) {
            const float4 g = reinterpret_cast<const float4*>(gy)[q];
            const float4 o = reinterpret_cast<const float4*>(y)[q];
            float4 r;
            r.x = g.x * gain * (o.x > 0.f ? 1.f : slope);
            r.y = g.y * gain * (o.y > 0.f ? 1.f : slope);
            r.z = g.z * gain * (o.z > 0.f ? 1.f : slope);
            r.w = g.w * gain * (o.w > 0.f ? 1.f : slope);
            reinterpret_cast<float4*>(gx)[q] = r;
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step)
            gx[e] = gy[e] * gain * (y[e] > 0.f ? 1.f : slope);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One (outer,channel) row of `inner` elements is split over `splits` blocks; each block reduces its
// slice and adds 3 partial sums atomically (sums zeroed by a memset node ahead of the launch).
__global__ void bias_act_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                           const float* __restrict__ noise, float* __restrict__ gx,
                                           float* __restrict__ sums, int64_t inner, int splits, float slope,
                                           float gain) {
    const int64_t row = blockIdx.x / splits;
    const int split = blockIdx.x % splits;
    const int64_t per = ((inner + splits - 1) / splits + 3) & ~int64_t(3);
    const int64_t lo = split * per;
    const int64_t hi = (lo + per < inner) ? lo + per : inner;
    const float* g = gy + row * inner;
    const float* o = y + row * inner;
    float* d = gx + row * inner;
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * slope);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const bool vec = (inner & 3) == 0;
    if (vec) {
        for (int64_t e = lo + threadIdx.x * 4; e < hi; e += blockDim.x * 4) {
            const float4 gv = *reinterpret_cast<const float4*>(g + e);
            const float4 ov = *reinterpret_cast<const float4*>(o + e);
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) nz = *reinterpret_cast<const float4*>(noise + e);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, oa[4] = {ov.x, ov.y, ov.z, ov.w};
            const float na[4] = {nz.x, nz.y, nz.z, nz.w};
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool pos = oa[k] > 0.f;
                r[k] = ga[k] * gain * (pos ? 1.f : slope);
                s0 += r[k] * (oa[k] * (pos ? inv_pos : inv_neg));
                s1 += r[k] * na[k];
                s2 += r[k];
            }
            *reinterpret_cast<float4*>(d + e) = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        for (int64_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
            const bool pos = o[e] > 0.f;
            const float r = g[e] * gain * (pos ? 1.f : slope);
            d[e] = r;
            s0 += r * (o[e] * (pos ? inv_pos : inv_neg));
            s1 += r * (noise ? noise[e] : 0.f);
            s2 += r;
        }
    }
    __shared__ float red[4][3];
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave][0] = s0, red[wave][1] = s1, red[wave][2] = s2;
    __syncthreads();
    if (threadIdx.x < 3) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (splits == 1)
            sums[row * 3 + threadIdx.x] = t;
        else
            atomicAdd(&sums[row * 3 + threadIdx.x], t);
    }
}

// ------------------------------------------------------------------------------------- K6
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    // torch nearest: min(floor(dst * (in/out)), in-1)  (attention_model.py:548 uses the default mode)
    const int s = (int)floorf(dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

__global__ void mask_blend_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ mask, float* __restrict__ out, int64_t total, int C,
                                      int H, int W, int ms) {
    const float sy = (float)ms / (float)H, sx = (float)ms / (float)W;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int x = (int)(e % W);
        const int yy = (int)((e / W) % H);
        const int64_t bi = e / ((int64_t)W * H * C);
        const float m = mask[(bi * ms + nearest_src(yy, sy, ms)) * ms + nearest_src(x, sx, ms)];
        out[e] = m * a[e] + (1.f - m) * b[e];
    }
}

// One block per (batch, row): lanes over x, loop over channels (coalesced), then one atomic per pixel
// into its mask cell.
__global__ void mask_blend_bwd_kernel(const float* __restrict__ go, const float* __restrict__ a,
                                      const float* __restrict__ b, const float* __restrict__ mask,
                                      float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ gmask, int C,
                                      int H, int W, int ms) {
    const int bi = blockIdx.x / H, yy = blockIdx.x % H;
    const float sy = (float)ms / (float)H, sx = (float)ms / (float)W;
    const int my = nearest_src(yy, sy, ms);
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int mx = nearest_src(x, sx, ms);
        const float m = mask[((int64_t)bi * ms + my) * ms + mx];
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const int64_t e = (((int64_t)bi * C + c) * H + yy) * W + x;
            const float g = go[e];
            ga[e] = m * g;
            if (gb) gb[e] = (1.f - m) * g;
            if (gmask) acc += g * (a[e] - b[e]);
        }
        if (gmask) atomicAdd(&gmask[((int64_t)bi * ms + my) * ms + mx], acc);
    }
}

// ------------------------------------------------------------------------------------- K5
// Output pixel (Y,X) of AvgPool_k(Upsample_7(img)) covers up-sampled rows [kY, kY+k) = source rows
// floor(kY/7) .. floor((kY+k-1)/7) with integer overlap weights; same for columns.
__device__ __forceinline__ int overlap(int lo, int hi, int s) {  // |[lo,hi) ∩ [7s,7s+7)|
    const int a = lo > 7 * s ? lo : 7 * s;
    const int b = hi < 7 * s + 7 ? hi : 7 * s + 7;
    return b > a ? b - a : 0;
}

__global__ void clip_preproc_fwd_kernel(const float* __restrict__ img, float* __restrict__ out, int64_t total, int size,
                                        int k) {
    const float inv = 1.f / (float)(k * k);
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int X = (int)(e % 224), Y = (int)((e / 224) % 224);
        const int64_t p = e / (224 * 224);
        const float* src = img + p * (int64_t)size * size;
        const int y0 = (k * Y) / 7, y1 = (k * Y + k - 1) / 7, x0 = (k * X) / 7, x1 = (k * X + k - 1) / 7;
        float acc = 0.f;
        for (int yy = y0; yy <= y1; ++yy) {
            const float wy = (float)overlap(k * Y, k * Y + k, yy);
            float row = 0.f;
            for (int xx = x0; xx <= x1; ++xx) row += (float)overlap(k * X, k * X + k, xx) * src[(int64_t)yy * size + xx];
            acc += wy * row;
        }
        out[e] = acc * inv;
    }
}

__global__ void clip_preproc_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gimg, int64_t total,
                                        int size, int k) {
    const float inv = 1.f / (float)(k * k);
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int x = (int)(e % size), y = (int)((e / size) % size);
        const int64_t p = e / ((int64_t)size * size);
        const float* g = gout + p * 224 * 224;
        const int Y0 = (7 * y) / k, Y1 = (7 * y + 6) / k, X0 = (7 * x) / k, X1 = (7 * x + 6) / k;
        float acc = 0.f;
        for (int Y = Y0; Y <= Y1; ++Y) {
            const float wy = (float)overlap(k * Y, k * Y + k, y);
            float row = 0.f;
            for (int X = X0; X <= X1; ++X) row += (float)overlap(k * X, k * X + k, x) * g[Y * 224 + X];
            acc += wy * row;
        }
        gimg[e] = acc * inv;
    }
}

// ------------------------------------------------------------------------------------- K5b
// IDLoss front end (criteria/id_loss.py:19-23): AdaptiveAvgPool(256) [exact k x k mean, k = size/256] -> crop
// [35:223, 32:220] -> AdaptiveAvgPool(112) on 188 x 188 (windows [floor(i*188/112), ceil((i+1)*188/112)), 2-3 wide).
__device__ __forceinline__ int apool_lo(int i) { return (i * 188) / 112; }
__device__ __forceinline__ int apool_hi(int i) { return ((i + 1) * 188 + 111) / 112; }

__global__ void id_preproc_fwd_kernel(const float* __restrict__ img, float* __restrict__ out, int64_t total, int size, int k) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const float inv_k2 = 1.f / (float)(k * k);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int X = (int)(e % 112), Y = (int)((e / 112) % 112);
        const float* src = img + (e / (112 * 112)) * (int64_t)size * size;
        const int r0 = apool_lo(Y), r1 = apool_hi(Y), c0 = apool_lo(X), c1 = apool_hi(X);
        float acc = 0.f;
        for (int r = r0; r < r1; ++r)
            for (int c = c0; c < c1; ++c) {
                const float* blk = src + (int64_t)(35 + r) * k * size + (32 + c) * k;
                float s = 0.f;
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) s += blk[a * size + b];
                acc += s * inv_k2;
            }
        out[e] = acc / (float)((r1 - r0) * (c1 - c0));
    }
}

__global__ void id_preproc_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gimg, int64_t total, int size, int k) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const float inv_k2 = 1.f / (float)(k * k);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int x = (int)(e % size), y = (int)((e / size) % size);
        const float* g = gout + (e / ((int64_t)size * size)) * 112 * 112;
        const int r = y / k - 35, c = x / k - 32;
        float acc = 0.f;
        if (r >= 0 && r < 188 && c >= 0 && c < 188) {
            // output windows that contain pooled row r / column c: at most two, around r*112/188
            const int Yc = (r * 112) / 188, Xc = (c * 112) / 188;
            for (int Y = Yc - 1; Y <= Yc + 1; ++Y) {
                if (Y < 0 || Y >= 112 || r < apool_lo(Y) || r >= apool_hi(Y)) continue;
                const float wy = 1.f / (float)(apool_hi(Y) - apool_lo(Y));
                for (int X = Xc - 1; X <= Xc + 1; ++X) {
                    if (X < 0 || X >= 112 || c < apool_lo(X) || c >= apool_hi(X)) continue;
                    acc += wy / (float)(apool_hi(X) - apool_lo(X)) * g[Y * 112 + X];
                }
            }
        }
        gimg[e] = acc * inv_k2;
    }
}

// ------------------------------------------------------------------------------------- demodulation
// d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + eps)   (model.py:241-243 with wsq = sum_k (scale*W)^2).
// One wave per (b,o).
__global__ __launch_bounds__(256) void demod_fwd_kernel(const float* __restrict__ s, const float* __restrict__ wsq,
                                                        float* __restrict__ d, int B, int Cin, int Cout, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Cout) return;
    const int b = row / Cout, o = row - b * Cout;
    float acc = 0.f;
    for (int i = lane; i < Cin; i += 64) {
        const float v = s[(int64_t)b * Cin + i];
        acc += v * v * wsq[(int64_t)o * Cin + i];
    }
    acc = wave_sum(acc);
    if (lane == 0) d[row] = rsqrtf(acc + eps);
}

// The same for every demodulated layer of a generator pass in one launch (the styles of all layers are known up front when
// the modulation affines are batched): grid (row groups of the widest layer, layer).
struct DemodAllLaunch {
    w2e_demod_layer layer[W2E_DEMOD_MAX_LAYERS];
    int batch;
    float eps;
};

__global__ __launch_bounds__(256) void demod_all_fwd_kernel(const DemodAllLaunch L) {
    const w2e_demod_layer& y = L.layer[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L.batch * y.cout) return;
    const int b = row / y.cout, o = row - b * y.cout;
    const float* s = y.s + (int64_t)b * y.cin;
    const float* w = y.wsq + (int64_t)o * y.cin;
    float acc = 0.f;
    for (int i = lane; i < y.cin; i += 64) {
        const float v = s[i];
        acc += v * v * w[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) y.d[row] = rsqrtf(acc + L.eps);
}

// Style gradient through the demodulation, added onto the direct part already in gs:
//   dz[b,o] = sum_p gpre * (d*z)  (= s1 - nw*s2 - bias*s3 from the fused-activation reductions, or given)
//   gd = dz / d;  dd/ds[b,i] = -d^3 * wsq[o,i] * s[b,i]   =>   gs[b,i] += -s[b,i] * sum_o dz[b,o]*d[b,o]^2*wsq[o,i]
// Block = (b, 256 input channels): coefficients in LDS, threads over i (coalesced wsq rows).
constexpr int DEMOD_OCH = 32;
__global__ __launch_bounds__(256) void demod_bwd_kernel(const float* __restrict__ sums, const float* __restrict__ dz_in,
                                                        const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                        const float* __restrict__ d, const float* __restrict__ s,
                                                        const float* __restrict__ wsq, float* __restrict__ gs,
                                                        float* __restrict__ gd_out, int Cin, int Cout) {
    // blockIdx.z slices the output channels (DEMOD_OCH per block: the sum over o is a chain of dependent-latency loads,
    // so it is cut short and spread over the chip); the slices add their part onto gs atomically.
    __shared__ float coef[DEMOD_OCH];
    const int b = blockIdx.y;
    const float nw = noise_w ? noise_w[0] : 0.f;
    // deterministic mode launches ONE z-slice that walks every output-channel group in order and adds without atomics
    const bool serial = gridDim.z == 1 && Cout > DEMOD_OCH;
    const int i_ser = blockIdx.x * 256 + threadIdx.x;
    float total = 0.f;
    for (int o_lo = blockIdx.z * DEMOD_OCH; o_lo < Cout; o_lo += DEMOD_OCH) {
    const int o_n = (Cout - o_lo < DEMOD_OCH) ? Cout - o_lo : DEMOD_OCH;
    __syncthreads();
    for (int oo = threadIdx.x; oo < o_n; oo += 256) {
        const int o = o_lo + oo;
        float dz;
        if (sums) {
            const float* q = sums + ((int64_t)b * Cout + o) * 3;
            dz = q[0] - nw * q[1] - (bias ? bias[o] : 0.f) * q[2];
        } else {
            dz = dz_in[(int64_t)b * Cout + o];
        }
        const float dv = d[(int64_t)b * Cout + o];
        coef[oo] = dz * dv * dv;
        if (gd_out && blockIdx.x == 0) gd_out[(int64_t)b * Cout + o] = dz / dv;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < Cin) {
        const float* wq = wsq + (int64_t)o_lo * Cin + i;
#pragma unroll 16
        for (int oo = 0; oo < o_n; ++oo) acc += coef[oo] * wq[(int64_t)oo * Cin];
    }
    if (!serial) {
        if (i < Cin) atomicAdd(&gs[(int64_t)b * Cin + i], -s[(int64_t)b * Cin + i] * acc);
        return;
    }
    total += acc;
    }
    if (i_ser < Cin) gs[(int64_t)b * Cin + i_ser] += -s[(int64_t)b * Cin + i_ser] * total;
}


// All modulation affines of one generator pass in one launch (model.py:211 `style = self.modulation(style)`, 26 times
// per forward): out[l][b][c] = latent[b, widx_l, :] . W_l[c, :] + bias_l[c], with every layer's (scale*W, lr_mul*b)
// stacked row-wise into w [R, D], bias [R].  meta[r] = (W+ index, rows before this layer, layer width, row in layer).
// One wave per stacked row: the weight row is read once and dotted with the row's latent of every batch element.
__global__ __launch_bounds__(256) void style_affine_fwd_kernel(const float* __restrict__ latent, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const int4* __restrict__ meta,
                                                               float* __restrict__ out, int B, int L, int D, int R) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int4 m = meta[r];
    const float bs = bias ? bias[r] : 0.f;
    for (int b = 0; b < B; ++b) {
        const float* lat = latent + ((int64_t)b * L + m.x) * D;
        float acc = 0.f;
        for (int k = lane * 4; k < D; k += 256) {
            const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)r * D + k);
            const float4 lv = *reinterpret_cast<const float4*>(lat + k);
            acc += wv.x * lv.x + wv.y * lv.y + wv.z * lv.z + wv.w * lv.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(int64_t)m.y * B + (int64_t)b * m.z + m.w] = acc + bs;
    }
}

// glatent[b, widx_l, k] += sum_c gout[l][b][c] * W_l[c, k].  Block = (32 stacked rows -- layer widths are multiples of
// 32, so one W+ index per block --, b); threads over k (coalesced weight rows); one atomic per (block, k).
__global__ __launch_bounds__(256) void style_affine_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                               const int4* __restrict__ meta, float* __restrict__ glatent,
                                                               int B, int L, int D, int R) {
    __shared__ float g[32];
    const int b = blockIdx.y;
    // deterministic mode: ONE block per batch element walks every 32-row group in order (plain read-modify-write by the
    // owning thread); default: one block per group, joined by atomics
    const bool serial = gridDim.x == 1 && R > 32;
    for (int r0 = blockIdx.x * 32; r0 < R; r0 += 32) {
    __syncthreads();
    if (threadIdx.x < 32) {
        const int r = r0 + threadIdx.x;
        float v = 0.f;
        if (r < R) {
            const int4 m = meta[r];
            v = gout[(int64_t)m.y * B + (int64_t)b * m.z + m.w];
        }
        g[threadIdx.x] = v;
    }
    __syncthreads();
    const int widx = meta[r0].x;
    const int nr = (R - r0 < 32) ? R - r0 : 32;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < nr; ++i) acc += g[i] * w[(int64_t)(r0 + i) * D + k];
        if (serial) glatent[((int64_t)b * L + widx) * D + k] += acc;
        else atomicAdd(&glatent[((int64_t)b * L + widx) * D + k], acc);
    }
    if (!serial) break;
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" {


int w2e_bias_act_fwd(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                     int64_t outer, int64_t channels, int64_t inner, float slope, float gain, void* stream) {
    W2E_REQUIRE(x && y, "bias_act_fwd: null tensor");
    W2E_REQUIRE(outer >= 0 && channels > 0 && inner > 0, "bias_act_fwd: bad dims %lld %lld %lld", (long long)outer,
                (long long)channels, (long long)inner);
    W2E_REQUIRE(!noise || noise_w, "bias_act_fwd: noise without noise_w");
    const int64_t total = outer * channels * inner;
    if (total == 0) return 0;
    const bool vec = ((inner & 3) == 0) || (inner == 1 && (channels & 3) == 0 && !noise);
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        bias_act_fwd_kernel<true><<<stream_grid(total / 4, 256), 256, 0, s>>>(x, bias, noise, noise_w, y, total,
                                                                                channels, inner, slope, gain);
    else
        bias_act_fwd_kernel<false><<<stream_grid(total, 256), 256, 0, s>>>(x, bias, noise, noise_w, y, total, channels,
                                                                             inner, slope, gain);
    W2E_LAUNCH_CHECK("bias_act_fwd");
    return 0;
}

int w2e_bias_act_bwd(const float* gy, const float* y, float* gx, int64_t n, float slope, float gain, void* stream) {
    W2E_REQUIRE(gy && y && gx, "bias_act_bwd: null tensor");
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0)
        bias_act_bwd_kernel<true><<<stream_grid(n / 4, 256), 256, 0, s>>>(gy, y, gx, n, slope, gain);
    else
        bias_act_bwd_kernel<false><<<stream_grid(n, 256), 256, 0, s>>>(gy, y, gx, n, slope, gain);
    W2E_LAUNCH_CHECK("bias_act_bwd");
    return 0;
}

int w2e_bias_act_bwd_reduce(const float* gy, const float* y, const float* noise, float* gx, float* sums,
                            int64_t outer, int64_t channels, int64_t inner, float slope, float gain, void* stream) {
    W2E_REQUIRE(gy && y && gx && sums, "bias_act_bwd_reduce: null tensor");
    W2E_REQUIRE(slope != 0.f && gain != 0.f, "bias_act_bwd_reduce: slope and gain must be non-zero");
    const int64_t rows = outer * channels;
    if (rows == 0 || inner == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int splits = 1;
    if (rows < 2048 && !options().deterministic) {  // deterministic: one block owns a row's sums (no atomics)
        splits = (int)ceil_div(2048, rows);
        const int64_t max_splits = ceil_div(inner, 1024);
        if (splits > max_splits) splits = (int)max_splits;
        if (splits < 1) splits = 1;
    }
    W2E_REQUIRE(rows * splits < (int64_t)1 << 31, "bias_act_bwd_reduce: grid too large");
    if (splits > 1 && zero_async(sums, sizeof(float) * 3 * rows, s) != hipSuccess) {
        set_error("bias_act_bwd_reduce: memset failed");
        return 2;
    }
    bias_act_bwd_reduce_kernel<<<(int)(rows * splits), 256, 0, s>>>(gy, y, noise, gx, sums, inner, splits, slope, gain);
    W2E_LAUNCH_CHECK("bias_act_bwd_reduce");
    return 0;
}

int w2e_mask_blend_fwd(const float* a, const float* b, const float* mask, float* out, int batch, int channels, int h,
                       int w, int ms, void* stream) {
    W2E_REQUIRE(a && b && mask && out, "mask_blend_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && channels > 0 && h > 0 && w > 0 && ms > 0, "mask_blend_fwd: bad dims");
    const int64_t total = (int64_t)batch * channels * h * w;
    if (total == 0) return 0;
    mask_blend_fwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(a, b, mask, out, total, channels, h,
                                                                                     w, ms);
    W2E_LAUNCH_CHECK("mask_blend_fwd");
    return 0;
}

int w2e_mask_blend_bwd(const float* gout, const float* a, const float* b, const float* mask, float* ga, float* gb,
                       float* gmask, int batch, int channels, int h, int w, int ms, void* stream) {
    W2E_REQUIRE(gout && mask && ga, "mask_blend_bwd: null tensor");
    W2E_REQUIRE(!gmask || (a && b), "mask_blend_bwd: gmask needs a and b");
    W2E_REQUIRE(batch >= 0 && channels > 0 && h > 0 && w > 0 && ms > 0, "mask_blend_bwd: bad dims");
    if (batch == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (gmask && zero_async(gmask, sizeof(float) * (size_t)batch * ms * ms, s) != hipSuccess) {
        set_error("mask_blend_bwd: memset failed");
        return 2;
    }
    // gmask: one atomic per pixel into its mask cell.  With the mask at the layer's resolution (the shipped setting:
    // attention_layer 13 = 64x64 features, 64x64 mask) every cell receives exactly ONE add: order-free.  Otherwise the
    // adds of a cell's pixels race; the deterministic mode refuses that geometry instead of giving run-to-run noise.
    W2E_REQUIRE(!(options().deterministic && gmask && (h != ms || w != ms)),
                "mask_blend_bwd: deterministic mode needs the mask at the feature resolution (%dx%d vs %d)", h, w, ms);
    const int threads = w >= 256 ? 256 : (w >= 128 ? 128 : 64);
    mask_blend_bwd_kernel<<<batch * h, threads, 0, s>>>(gout, a, b, mask, ga, gb, gmask, channels, h, w, ms);
    W2E_LAUNCH_CHECK("mask_blend_bwd");
    return 0;
}

int w2e_clip_preproc_fwd(const float* img, float* out, int64_t planes, int size, void* stream) {
    W2E_REQUIRE(img && out, "clip_preproc_fwd: null tensor");
    W2E_REQUIRE(size >= 32 && size % 32 == 0, "clip_preproc_fwd: size %d must be a positive multiple of 32", size);
    const int64_t total = planes * 224 * 224;
    if (total <= 0) return 0;
    clip_preproc_fwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(img, out, total, size, size / 32);
    W2E_LAUNCH_CHECK("clip_preproc_fwd");
    return 0;
}

int w2e_clip_preproc_bwd(const float* gout, float* gimg, int64_t planes, int size, void* stream) {
    W2E_REQUIRE(gout && gimg, "clip_preproc_bwd: null tensor");
    W2E_REQUIRE(size >= 32 && size % 32 == 0, "clip_preproc_bwd: size %d must be a positive multiple of 32", size);
    const int64_t total = planes * (int64_t)size * size;
    if (total <= 0) return 0;
    clip_preproc_bwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(gout, gimg, total, size, size / 32);
    W2E_LAUNCH_CHECK("clip_preproc_bwd");
    return 0;
}

int w2e_demod_fwd(const float* s, const float* wsq, float* d, int batch, int cin, int cout, float eps, void* stream) {
    W2E_REQUIRE(s && wsq && d, "demod_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && cin > 0 && cout > 0, "demod_fwd: bad dims");
    if (batch == 0) return 0;
    demod_fwd_kernel<<<(unsigned)ceil_div((int64_t)batch * cout, 4), 256, 0, (hipStream_t)stream>>>(s, wsq, d, batch, cin, cout, eps);
    W2E_LAUNCH_CHECK("demod_fwd");
    return 0;
}

int w2e_demod_all_fwd(const w2e_demod_layer* layers, int n_layers, int batch, float eps, void* stream) {
    W2E_REQUIRE(layers, "demod_all_fwd: null layers");
    W2E_REQUIRE(n_layers >= 1 && n_layers <= W2E_DEMOD_MAX_LAYERS, "demod_all_fwd: 1 <= n_layers <= %d", W2E_DEMOD_MAX_LAYERS);
    W2E_REQUIRE(batch >= 0, "demod_all_fwd: bad dims");
    if (batch == 0) return 0;
    DemodAllLaunch L{};
    int max_cout = 0;
    for (int j = 0; j < n_layers; ++j) {
        W2E_REQUIRE(layers[j].s && layers[j].wsq && layers[j].d && layers[j].cin > 0 && layers[j].cout > 0,
                    "demod_all_fwd: layer %d: null tensor or bad dims", j);
        L.layer[j] = layers[j];
        if (layers[j].cout > max_cout) max_cout = layers[j].cout;
    }
    L.batch = batch, L.eps = eps;
    dim3 grid((unsigned)ceil_div((int64_t)batch * max_cout, 4), (unsigned)n_layers);
    demod_all_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(L);
    W2E_LAUNCH_CHECK("demod_all_fwd");
    return 0;
}

int w2e_demod_bwd(const float* sums, const float* dz, const float* noise_w, const float* bias, const float* d,
                  const float* s, const float* wsq, float* gs, float* gd, int batch, int cin, int cout, void* stream) {
    W2E_REQUIRE((sums != nullptr) != (dz != nullptr), "demod_bwd: give exactly one of sums / dz");
    W2E_REQUIRE(d && s && wsq && gs, "demod_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && cin > 0 && cout > 0 && batch < 65536, "demod_bwd: bad dims");
    if (batch == 0) return 0;
    dim3 grid((unsigned)ceil_div(cin, 256), (unsigned)batch, options().deterministic ? 1u : (unsigned)ceil_div(cout, DEMOD_OCH));
    demod_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(sums, dz, noise_w, bias, d, s, wsq, gs, gd, cin, cout);
    W2E_LAUNCH_CHECK("demod_bwd");
    return 0;
}

int w2e_id_preproc_fwd(const float* img, float* out, int64_t planes, int size, void* stream) {
    W2E_REQUIRE(img && out, "id_preproc_fwd: null tensor");
    W2E_REQUIRE(size >= 256 && size % 256 == 0, "id_preproc_fwd: size %d must be a positive multiple of 256", size);
    const int64_t total = planes * 112 * 112;
    if (total <= 0) return 0;
    id_preproc_fwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(img, out, total, size, size / 256);
    W2E_LAUNCH_CHECK("id_preproc_fwd");
    return 0;
}

int w2e_id_preproc_bwd(const float* gout, float* gimg, int64_t planes, int size, void* stream) {
    W2E_REQUIRE(gout && gimg, "id_preproc_bwd: null tensor");
    W2E_REQUIRE(size >= 256 && size % 256 == 0, "id_preproc_bwd: size %d must be a positive multiple of 256", size);
    const int64_t total = planes * (int64_t)size * size;
    if (total <= 0) return 0;
    id_preproc_bwd_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(gout, gimg, total, size, size / 256);
    W2E_LAUNCH_CHECK("id_preproc_bwd");
    return 0;
}

int w2e_style_affine_fwd(const float* latent, const float* w, const float* bias, const int* meta, float* out, int batch,
                         int n_latent, int dim, int rows, void* stream) {
    W2E_REQUIRE(latent && w && meta && out, "style_affine_fwd: null tensor");
    W2E_REQUIRE(batch >= 0 && n_latent > 0 && dim > 0 && (dim & 3) == 0 && rows > 0, "style_affine_fwd: bad dims");
    if (batch == 0) return 0;
    style_affine_fwd_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(
        latent, w, bias, reinterpret_cast<const int4*>(meta), out, batch, n_latent, dim, rows);
    W2E_LAUNCH_CHECK("style_affine_fwd");
    return 0;
}

int w2e_style_affine_bwd(const float* gout, const float* w, const int* meta, float* glatent, int batch, int n_latent,
                         int dim, int rows, void* stream) {
    W2E_REQUIRE(gout && w && meta && glatent, "style_affine_bwd: null tensor");
    W2E_REQUIRE(batch >= 0 && batch < 65536 && n_latent > 0 && dim > 0 && rows > 0, "style_affine_bwd: bad dims");
    if (batch == 0) return 0;
    if (zero_async(glatent, sizeof(float) * (size_t)batch * n_latent * dim, (hipStream_t)stream) != hipSuccess) {
        set_error("style_affine_bwd: memset failed");
        return 2;
    }
    dim3 grid(options().deterministic ? 1u : (unsigned)ceil_div(rows, 32), (unsigned)batch);
    style_affine_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gout, w, reinterpret_cast<const int4*>(meta), glatent, batch,
                                                                 n_latent, dim, rows);
    W2E_LAUNCH_CHECK("style_affine_bwd");
    return 0;
}


}  // extern "C"
