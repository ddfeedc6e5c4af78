// Winograd forms -- F(2x2,3x3) and F(4x4,3x3) -- of the same-resolution modulated 3x3 convolution (K1w, include/w2e.h) for gfx950.
//
//   y[b,o] = out_scale[b,o] * conv3x3(W, in_scale[b,i] * x[b,i])          (model.py:270-274 in the shared-weight form of K1)
//          = out_scale * A^T [ sum_i (G W[o,i] G^T) (.) (B^T (in_scale * d[b,i]) B) ] A     per 2x2 output tile, d = its 4x4 input window
//
// 16 multiplications per 2x2 outputs instead of 36: the contraction over input channels becomes 16 independent
// [N x K] x [K x tiles] GEMMs with 2.25x fewer FLOPs than the direct form.  The three passes here are the HBM-bound ends:
//   wino_weights   U[xi][n][k]  = (G W G^T)[xi] from the packed direct-form weights (once per pack, cached by the caller)
//   wino_input     V[xi][k][t]  = (B^T (in_scale * d) B)[xi], t = (b, tile row, tile column)
//   wino_output    y            = epilogue(out_scale * A^T M A), M[xi][n][t] = U[xi] V[xi] -- the same epilogues as
//                                 w2e_modconv3x3: noise + bias + LeakyReLU, and the fused per-channel dot of the input gradient
// and the 16 GEMMs between them are plain strided-batched fp32 GEMMs (the host uses the vendor library: hipBLASLt through
// torch.bmm; 118-135 TFLOP/s on these shapes, profiles/r03_winograd.txt).  V and M are 4x the size of the input / output, so
// the form pays where the contraction dominates the traffic (functional._wino_form); F(4x4,3x3) further down in this file.
// fp32 throughout; rounding differs from the direct form by ~2x its own error (6e-7 vs 3e-7 relative at K = 512).
#include "common.h"
#include "../../include/w2e.h"

namespace w2e {

// wp [ceil(K/8)][9][2][N][4]: element (kc, tap, h, n, c) = W(k = 8*kc + 2*c + h, tap, n)  ->  U [16][N][K]
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ wp, float* __restrict__ u, int K, int N) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)K * N) return;
    const int k = (int)(e % K), n = (int)(e / K);
    const int kc = k >> 3, c = (k & 7) >> 1, h = k & 1;
    float g[3][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g[tap / 3][tap % 3] = wp[((((int64_t)kc * 9 + tap) * 2 + h) * N + n) * 4 + c];
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]:  t = G g (4x3), U = t G^T (4x4)
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) u[((int64_t)(i * 4 + j) * N + n) * K + k] = v[j];
    }
}

// One thread per (plane (b,k), tile): the 4x4 window at (2*ty - 1, 2*tx - 1), zero outside the image.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                         float* __restrict__ v, int B, int K, int H, int W) {
    const int TX = W >> 1, TY = H >> 1, tiles = TX * TY;
    const int64_t T = (int64_t)B * tiles;
    const int plane = blockIdx.y;  // b * K + k
    const int b = plane / K, k = plane - b * K;
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= tiles) return;
    const int ty = tile / TX, tx = tile - ty * TX;
    const float* xp = x + (int64_t)plane * H * W;
    const float sc = in_scale ? in_scale[plane] : 1.f;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
        const bool rin = iy >= 0 && iy < H;
        const float* row = xp + (int64_t)(rin ? iy : 0) * W + 2 * tx;
        const float2 mid = rin ? *reinterpret_cast<const float2*>(row) : make_float2(0.f, 0.f);
        d[r][0] = (rin && tx > 0) ? row[-1] : 0.f;
        d[r][1] = mid.x, d[r][2] = mid.y;
        d[r][3] = (rin && tx + 1 < TX) ? row[2] : 0.f;
    }
    // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]:  t = B^T d, V = t B
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
    }
    float* vp = v + (int64_t)k * T + (int64_t)b * tiles + tile;
    const int64_t xi_stride = (int64_t)K * T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vp[(i * 4 + 0) * xi_stride] = sc * (t[i][0] - t[i][2]);
        vp[(i * 4 + 1) * xi_stride] = sc * (t[i][1] + t[i][2]);
        vp[(i * 4 + 2) * xi_stride] = sc * (t[i][2] - t[i][1]);
        vp[(i * 4 + 3) * xi_stride] = sc * (t[i][1] - t[i][3]);
    }
}

// One thread per (output channel n, tile t).  A wave's 64 tiles belong to one (b, n) plane (the host requires tiles % 64 == 0),
// so the fused dot is a wave reduction and one atomic per wave.
// ACT: 0 none; 1 = + noise_w*noise + bias, LeakyReLU(0.2) * sqrt 2 (StyledConv); 2 = + bias, PReLU(slope[n]) when slope != NULL (IR-SE50).
template <int ACT, bool DOT>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ m, const float* __restrict__ out_scale,
                                                          float* __restrict__ y, int B, int N, int H, int W,
                                                          const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          const float* __restrict__ dot_with, float* __restrict__ dot_out) {
    const int TX = W >> 1, TY = H >> 1, tiles = TX * TY;
    const int64_t T = (int64_t)B * tiles;
    const int n = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;  // (with the fused dot: whole waves, T % 64 == 0)
    const int b = (int)(t / tiles), tile = (int)(t - (int64_t)b * tiles);
    const int ty = tile / TX, tx = tile - ty * TX;
    const float* mp = m + (int64_t)n * T + t;
    const int64_t xi_stride = (int64_t)N * T;
    float q[4][4];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) q[xi >> 2][xi & 3] = mp[xi * xi_stride];
    // A^T = [[1,1,1,0],[0,1,-1,-1]]:  s = A^T q (2x4), Y = s A (2x2)
    float s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = q[0][j] + q[1][j] + q[2][j];
        s[1][j] = q[1][j] - q[2][j] - q[3][j];
    }
    float o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        o[i][0] = s[i][0] + s[i][1] + s[i][2];
        o[i][1] = s[i][1] - s[i][2] - s[i][3];
    }
    const int64_t plane = (int64_t)b * N + n;
    const int64_t pix = (int64_t)(2 * ty) * W + 2 * tx;
    if (DOT) {  // dot_out[b,n] += sum_p conv_unscaled * dot_with  (w2e.h: before out_scale)
        const float* dp = dot_with + plane * H * W + pix;
        const float2 d0 = *reinterpret_cast<const float2*>(dp), d1 = *reinterpret_cast<const float2*>(dp + W);
        float part = (o[0][0] * d0.x + o[0][1] * d0.y) + (o[1][0] * d1.x + o[1][1] * d1.y);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(dot_out + plane, part);
    }
    const float os = out_scale ? out_scale[plane] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v0 = o[i][0] * os, v1 = o[i][1] * os;
        if (ACT == 1) {
            const float nw = noise ? noise_w[0] : 0.f;
            const float bs = bias ? bias[n] : 0.f;
            float2 nz = make_float2(0.f, 0.f);
            if (noise) nz = *reinterpret_cast<const float2*>(noise + pix + (int64_t)i * W);
            v0 += nw * nz.x + bs, v1 += nw * nz.y + bs;
            v0 = fmaxf(v0, 0.2f * v0) * 1.4142135623730951f, v1 = fmaxf(v1, 0.2f * v1) * 1.4142135623730951f;
        }
        if (ACT == 2) {
            const float bs = bias ? bias[n] : 0.f;
            v0 += bs, v1 += bs;
            if (slope) {
                const float sl = slope[n];
                v0 = v0 > 0.f ? v0 : sl * v0, v1 = v1 > 0.f ? v1 : sl * v1;
            }
        }
        *reinterpret_cast<float2*>(y + plane * H * W + pix + (int64_t)i * W) = make_float2(v0, v1);
    }
}


// ---------------------------------------------------------------------------------------------------------------- F(4x4, 3x3)
// 36 products per 4x4 outputs instead of 144 (4x fewer FLOPs), and transform-domain tensors 2.25x the input / output instead of 4x:
// faster than F(2x2,3x3) everywhere and ahead of the direct kernel down to 128 channels at 256^2.  The price is rounding: the
// transforms multiply by up to 8 and 1/24, and the result sits ~1e-5 (max-norm relative, K = 128 ... 512) from the float64
// convolution where the direct form and F(2x2,3x3) sit at 3e-7 / 6e-7 -- two decades inside the path's 1e-3 tolerance (BASELINE
// north_star), one inside the tests' 1e-4; functional.set_winograd("f2") keeps the tighter form.
// Interpolation points 0, +-1, +-2, inf (Lavin & Gray):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// FUSED: U in the A-operand order of wino4_fused_kernel, uf[36][K/8][2][N][4]: (xi, kc, h, n, c) = U[xi][n][8*kc + 2*c + h]
template <bool FUSED>
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ wp, float* __restrict__ u, int K, int N) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)K * N) return;
    const int k = (int)(e % K), n = (int)(e / K);
    const int kc = k >> 3, c = (k & 7) >> 1, h = k & 1;
    double g[3][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g[tap / 3][tap % 3] = (double)wp[((((int64_t)kc * 9 + tap) * 2 + h) * N + n) * 4 + c];
    // (in float64, rounded once: the 1/6 and 1/24 of G are not exact in binary)
    auto row = [](double a, double b, double cc, double (&o)[6]) {
        o[0] = a / 4.0;
        o[1] = -(a + b + cc) / 6.0;
        o[2] = -(a - b + cc) / 6.0;
        o[3] = a / 24.0 + b / 12.0 + cc / 6.0;
        o[4] = a / 24.0 - b / 12.0 + cc / 6.0;
        o[5] = cc;
    };
    double t[6][3];  // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double col[6];
        row(g[0][j], g[1][j], g[2][j], col);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][j] = col[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double o[6];
        row(t[i][0], t[i][1], t[i][2], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (FUSED) u[((((int64_t)(i * 6 + j) * (K >> 3) + kc) * 2 + h) * N + n) * 4 + c] = (float)o[j];
            else u[((int64_t)(i * 6 + j) * N + n) * K + k] = (float)o[j];
        }
    }
}

__device__ __forceinline__ void wino4_bt(const float (&d)[6], float (&t)[6]) {  // t = B^T d
    const float a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// One thread per (plane (b,k), tile): the 6x6 window at (4*ty - 1, 4*tx - 1), zero outside the image.
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                          float* __restrict__ v, int B, int K, int H, int W) {
    const int TX = W >> 2, TY = H >> 2, tiles = TX * TY;
    const int64_t T = (int64_t)B * tiles;
    const int plane = blockIdx.y;  // b * K + k
    const int b = plane / K, k = plane - b * K;
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= tiles) return;
    const int ty = tile / TX, tx = tile - ty * TX;
    const float* xp = x + (int64_t)plane * H * W;
    const float sc = in_scale ? in_scale[plane] : 1.f;
    float t[6][6];  // rows transformed first: t[r][.] = B^T (row r of the window), then the columns
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int iy = 4 * ty - 1 + r;
        const bool rin = iy >= 0 && iy < H;
        const float* row = xp + (int64_t)(rin ? iy : 0) * W + 4 * tx;
        const float4 mid = rin ? *reinterpret_cast<const float4*>(row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float d[6] = {(rin && tx > 0) ? row[-1] : 0.f, mid.x, mid.y, mid.z, mid.w, (rin && tx + 1 < TX) ? row[4] : 0.f};
        wino4_bt(d, t[r]);
    }
    float* vp = v + (int64_t)k * T + (int64_t)b * tiles + tile;
    const int64_t xi_stride = (int64_t)K * T;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]};
        float o[6];
        wino4_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) vp[(i * 6 + j) * xi_stride] = sc * o[i];
    }
}

__device__ __forceinline__ void wino4_at(const float (&m)[6], float (&y)[4]) {  // y = A^T m
    const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = q + 2.f * s;
    y[2] = p + 4.f * r;
    y[3] = q + 8.f * s + m[5];
}

// One thread per (output channel n, tile t): a 4x4 block of outputs, stored as four 16-byte rows.  SEG = lanes of a wave that share
// one (b, n) plane (min(64, tiles per plane); the host requires it to divide 64): the fused dot reduces over them.
template <int ACT, bool DOT>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ m, const float* __restrict__ out_scale,
                                                           float* __restrict__ y, int B, int N, int H, int W, int seg,
                                                           const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                           const float* __restrict__ bias, const float* __restrict__ slope,
                                                           const float* __restrict__ dot_with, float* __restrict__ dot_out) {
    const int TX = W >> 2, TY = H >> 2, tiles = TX * TY;
    const int64_t T = (int64_t)B * tiles;
    const int n = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < T;
    const int64_t tc = live ? t : T - 1;  // (idle lanes of the last wave keep the shuffles below well defined)
    const int b = (int)(tc / tiles), tile = (int)(tc - (int64_t)b * tiles);
    const int ty = tile / TX, tx = tile - ty * TX;
    const float* mp = m + (int64_t)n * T + tc;
    const int64_t xi_stride = (int64_t)N * T;
    float s[4][6];  // columns transformed first: s[.][j] = A^T (column j of the 6x6 products)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = mp[(i * 6 + j) * xi_stride];
        float o[4];
        wino4_at(col, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = o[i];
    }
    const int64_t plane = (int64_t)b * N + n;
    const int64_t pix = (int64_t)(4 * ty) * W + 4 * tx;
    const float os = out_scale ? out_scale[plane] : 1.f;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
    const float bs = (ACT && bias) ? bias[n] : 0.f;
    const float sl = (ACT == 2 && slope) ? slope[n] : 1.f;
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o[4];
        wino4_at(s[i], o);
        if (DOT) {
            const float4 d = *reinterpret_cast<const float4*>(dot_with + plane * H * W + pix + (int64_t)i * W);
            part += (o[0] * d.x + o[1] * d.y) + (o[2] * d.z + o[3] * d.w);
        }
        float4 r = make_float4(o[0] * os, o[1] * os, o[2] * os, o[3] * os);
        if (ACT == 1) {
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) nz = *reinterpret_cast<const float4*>(noise + pix + (int64_t)i * W);
            r.x += nw * nz.x + bs, r.y += nw * nz.y + bs, r.z += nw * nz.z + bs, r.w += nw * nz.w + bs;
            r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
            r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
        }
        if (ACT == 2) {
            r.x += bs, r.y += bs, r.z += bs, r.w += bs;
            r.x = r.x > 0.f ? r.x : sl * r.x, r.y = r.y > 0.f ? r.y : sl * r.y;
            r.z = r.z > 0.f ? r.z : sl * r.z, r.w = r.w > 0.f ? r.w : sl * r.w;
        }
        if (live) *reinterpret_cast<float4*>(y + plane * H * W + pix + (int64_t)i * W) = r;
    }
    if (DOT) {  // dot_out[b,n] += sum_p conv_unscaled * dot_with  (w2e.h: before out_scale)
        if (!live) part = 0.f;
        for (int off = seg >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (live && ((threadIdx.x & 63) & (seg - 1)) == 0) atomicAdd(dot_out + plane, part);
    }
}


// ------------------------------------------------------------------------------------------------- F(4x4, 3x3), fused
// The 32- and 64-channel layers at 512^2 / 1024^2: their transform-domain tensors (2.25x the input and the output, written and
// read) cost more HBM time than the 4x fewer FLOPs return, so here V and M never leave the CU.  One workgroup = one block of
// 32 output tiles (8 x 4 tiles = 32 x 16 pixels) x all N = 32*NB output channels; 4*NB waves; wave (nb, g) owns the 9 transform
// positions xi = 9g .. 9g+8 of output-channel block nb: 9 accumulators of 32 channels x 32 tiles.  Per 8-channel chunk:
//   threads 0..255: one (tile, channel) each -- the 6x6 window straight from global memory (prefetched during the previous chunk's
//     MFMAs), B^T.B, in_scale, 36 scalars into the LDS image V[xi][half][tile] (float4 = the 4 channel pairs: one ds_read_b128 is
//     the B operand of 4 MFMAs)
//   every wave: for its 9 positions, A = one float4 of the transformed weights (global, L2-resident: 36*K*N floats) x B -> 4 MFMAs
// then the products go through LDS in 4 rounds of 8 output channels ([xi][n8][tile], the same bytes as V) to the threads that
// own an (output channel, tile) pair: A^T.A, out_scale and the epilogues of wino4_output_kernel, four 16-byte row stores.
template <int NB, int ACT, bool DOT>
__global__ __launch_bounds__(256 * NB, NB == 1 ? 2 : 1) void wino4_fused_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                               const float* __restrict__ uf, const float* __restrict__ out_scale,
                                                               float* __restrict__ y, int B, int K, int H, int W,
                                                               const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                               const float* __restrict__ bias, const float* __restrict__ slope,
                                                               const float* __restrict__ dot_with, float* __restrict__ dot_out) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int N = 32 * NB;
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // V: 36*2*32 float4; later M: NB*36*8*32 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb = wave >> 2, g = wave & 3;
    const int bx_n = W >> 5, by_n = H >> 4;
    const int blk = blockIdx.x;
    const int bx = blk % bx_n, by = (blk / bx_n) % by_n, b = blk / (bx_n * by_n);
    const int KC = K >> 3;
    // transform role (threads 0..255): tile tj of the block, channel tch of the chunk
    const int tj = tid & 31, tch = (tid >> 5) & 7;
    const bool xform = tid < 256;
    const int py0 = by * 16 + 4 * (tj >> 3), px0 = bx * 32 + 4 * (tj & 7);
    float win[6][6];
    auto load_win = [&](int kc) __attribute__((always_inline)) {
        const float* xp = x + ((int64_t)b * K + kc * 8 + tch) * H * W;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = py0 - 1 + r;
            const bool rin = iy >= 0 && iy < H;
            const float* row = xp + (int64_t)(rin ? iy : 0) * W + px0;
            const float4 mid = rin ? *reinterpret_cast<const float4*>(row) : make_float4(0.f, 0.f, 0.f, 0.f);
            win[r][0] = (rin && px0 > 0) ? row[-1] : 0.f;
            win[r][1] = mid.x, win[r][2] = mid.y, win[r][3] = mid.z, win[r][4] = mid.w;
            win[r][5] = (rin && px0 + 4 < W) ? row[4] : 0.f;
        }
    };
    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if (xform) load_win(0);
    const float4* uf4 = reinterpret_cast<const float4*>(uf);
    const float4* vs4 = reinterpret_cast<const float4*>(wsm);
    for (int kc = 0; kc < KC; ++kc) {
        if (xform) {
            const float sc = in_scale ? in_scale[(int64_t)b * K + kc * 8 + tch] : 1.f;
            float t[6][6];
#pragma unroll
            for (int r = 0; r < 6; ++r) wino4_bt(win[r], t[r]);
            float* vp = wsm + ((tch & 1) * 32 + tj) * 4 + (tch >> 1);  // [xi][half = tch & 1][tile][c = tch >> 1]
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const float col[6] = {t[0][jj], t[1][jj], t[2][jj], t[3][jj], t[4][jj], t[5][jj]};
                float o[6];
                wino4_bt(col, o);
#pragma unroll
                for (int i = 0; i < 6; ++i) vp[(i * 6 + jj) * 256] = sc * o[i];
            }
        }
        __syncthreads();
        if (xform && kc + 1 < KC) load_win(kc + 1);  // in flight during this chunk's MFMAs
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int xi = 9 * g + q;
            const float4 a4 = uf4[(((int64_t)xi * KC + kc) * 2 + half) * N + nb * 32 + j];
            const float4 b4 = vs4[(xi * 2 + half) * 32 + j];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[q], 0, 0, 0);
        }
        __syncthreads();  // (the next chunk's transform overwrites V)
    }
    // ---- output: 4 rounds of 8 output channels per block of 32.  Accumulator register r of lane (half, j) is row
    // (r & 3) + 8 * (r >> 2) + 4 * half, column j: round q4 moves the registers 4*q4 .. 4*q4+3 = rows 8*q4 .. 8*q4+7.
    const int oj = tid & 31, on8 = (tid >> 5) & 7, onb = tid >> 8;
    const int opy = by * 16 + 4 * (oj >> 3), opx = bx * 32 + 4 * (oj & 7);
    const int64_t opix = (int64_t)opy * W + opx;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
#pragma unroll 1
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) wsm[((nb * 36 + 9 * g + q) * 8 + rr + 4 * half) * 32 + j] = acc[q][4 * q4 + rr];
        __syncthreads();
        const int n = onb * 32 + 8 * q4 + on8;
        const float* mp = wsm + (onb * 36 * 8 + on8) * 32 + oj;
        float s[4][6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            float col[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = mp[(i * 6 + jj) * 256];
            float o[4];
            wino4_at(col, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][jj] = o[i];
        }
        const int64_t plane = (int64_t)b * N + n;
        const float os = out_scale ? out_scale[plane] : 1.f;
        const float bs = (ACT && bias) ? bias[n] : 0.f;
        const float sl = (ACT == 2 && slope) ? slope[n] : 1.f;
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o[4];
            wino4_at(s[i], o);
            if (DOT) {
                const float4 d = *reinterpret_cast<const float4*>(dot_with + plane * H * W + opix + (int64_t)i * W);
                part += (o[0] * d.x + o[1] * d.y) + (o[2] * d.z + o[3] * d.w);
            }
            float4 r = make_float4(o[0] * os, o[1] * os, o[2] * os, o[3] * os);
            if (ACT == 1) {
                float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
                if (noise) nz = *reinterpret_cast<const float4*>(noise + opix + (int64_t)i * W);
                r.x += nw * nz.x + bs, r.y += nw * nz.y + bs, r.z += nw * nz.z + bs, r.w += nw * nz.w + bs;
                r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
                r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
            }
            if (ACT == 2) {
                r.x += bs, r.y += bs, r.z += bs, r.w += bs;
                r.x = r.x > 0.f ? r.x : sl * r.x, r.y = r.y > 0.f ? r.y : sl * r.y;
                r.z = r.z > 0.f ? r.z : sl * r.z, r.w = r.w > 0.f ? r.w : sl * r.w;
            }
            *reinterpret_cast<float4*>(y + plane * H * W + opix + (int64_t)i * W) = r;
        }
        if (DOT) {  // the 32 tiles of a half-wave share (b, n)
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (oj == 0) atomicAdd(dot_out + plane, part);
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------- F(4x4, 3x3), fused, persistent + specialised waves
// wino4_fused_kernel above keeps V and M on the CU but its threads wait on their own window loads: the MFMAs of a chunk (36 per
// wave, ~2.3 k cycles) are far too short to cover an HBM round trip, and its registers (250) leave no room to keep more in flight.
// Here a workgroup is 8 waves with two jobs:
//   waves 4..7 ("loaders")  one (tile, channel) each per 8-channel chunk: window loads FOUR chunks ahead (4 x 36 registers: these waves
//                            have no accumulators), B^T.B, 36 scalars into the V stage of the chunk
//   waves 0..3 ("matrix")   9 transform positions each: A = transformed weights (global, L2; prefetched one chunk ahead) x B from the
//                            V stage -> 36 MFMAs per chunk; then the block's output rounds through their own LDS buffer
// and it is persistent over (spatial block, 32-channel output block) pairs, so the loaders run into the next block while the matrix
// waves are in their output rounds.  Everybody advances in lock-step "ticks" (one barrier each): a matrix tick consumes chunk t from
// V[t & 1] while the loaders write chunk t+1 into V[(t+1) & 1]; an output tick is half an output round.  The loaders may be at most
// two chunks ahead (both stages full).  Output channels: any multiple of 32 (blockIdx.y).  K: a power of two >= 32.
typedef float wf_f32x16 __attribute__((ext_vector_type(16)));

// The 6x6 window of one (tile, channel): per row one 16-byte and two 4-byte buffer loads, UNCONDITIONAL -- rows / columns outside the
// image and whole chunks past the last block get an offset past the descriptor and read 0 -- so that the loader's stream has no
// branches around its loads and the compiler's vmcnt bookkeeping stays exact (with `cond ? *p : 0` loads every wait became vmcnt(0)).
// `base`: byte offset of the window's plane in x, or the marker for "nothing to load".
__device__ __forceinline__ void wf2_load(float (&win)[36], const __amdgpu_buffer_rsrc_t rx, unsigned base, int py0, int px0, int H, int W) {
    typedef float wf_f32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned kOut = 0xfffffff0u;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int iy = py0 - 1 + r;
        const bool rin = base != kOut && iy >= 0 && iy < H;
        const unsigned off = base + (unsigned)(iy * W + px0) * 4u;
        const wf_f32x4 mid = __builtin_bit_cast(wf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, rin ? off : kOut, 0, 0));
        win[r * 6 + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && px0 > 0) ? off - 4u : kOut, 0, 0));
        win[r * 6 + 1] = mid[0], win[r * 6 + 2] = mid[1], win[r * 6 + 3] = mid[2], win[r * 6 + 4] = mid[3];
        win[r * 6 + 5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (rin && px0 + 4 < W) ? off + 16u : kOut, 0, 0));
    }
}

__device__ __forceinline__ void wf2_transform(const float (&win)[36], float* __restrict__ vp, float sc) {
    float t[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float d[6] = {win[r * 6 + 0], win[r * 6 + 1], win[r * 6 + 2], win[r * 6 + 3], win[r * 6 + 4], win[r * 6 + 5]};
        wino4_bt(d, t[r]);
    }
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
        const float col[6] = {t[0][jj], t[1][jj], t[2][jj], t[3][jj], t[4][jj], t[5][jj]};
        float o[6];
        wino4_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) vp[(i * 6 + jj) * 256] = sc * o[i];
    }
}

template <int ACT, bool DOT>
__global__ __launch_bounds__(512, 1) void wino4_fused2_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ uf, const float* __restrict__ out_scale,
                                                              float* __restrict__ y, int B, int K, int N, int H, int W, int kc_log2,
                                                              int n_blocks, const float* __restrict__ noise,
                                                              const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                              const float* __restrict__ slope, const float* __restrict__ dot_with,
                                                              float* __restrict__ dot_partial, int skip) {
#ifndef W2E_TUNING
    skip = 0;  // (work-skipping exists in tuning builds only: bit 0 no window loads after the first, 1 no transform, 2 no MFMAs, 3 no output rounds)
#endif
    constexpr int VS = 36 * 2 * 32 * 4;  // floats of one V stage; the output buffer M[36][16][32] is two of them
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // V[2][VS], M[2 * VS]
    float* const mbuf = wsm + 2 * VS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx_n = W >> 5, by_n = H >> 4, per_img = bx_n * by_n;
    const int KC = 1 << kc_log2;
    const int count = (n_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n0 = blockIdx.y * 32;
    // ---- the output of a block, shared by both roles: 2 rounds of 16 output channels; in each the matrix waves park their products in
    // M[xi][n16][tile] (tick A), then ALL 512 threads own one (channel, tile) pair: A^T.A, epilogue, four 16-byte row stores (tick B).
    // The per-pair operands of the epilogue (noise row quads, scales) are fetched in tick A so that they have landed in tick B.
    const int oj = tid & 31, on16 = tid >> 5;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
    struct OutPre {
        float4 nz[4];
        float os, bs, sl;
    };
    auto out_prefetch = [&](OutPre& pre, int b, int64_t opix, int n) __attribute__((always_inline)) {
        const int64_t plane = (int64_t)b * N + n;
        pre.os = out_scale ? out_scale[plane] : 1.f;
        pre.bs = (ACT && bias) ? bias[n] : 0.f;
        pre.sl = (ACT == 2 && slope) ? slope[n] : 1.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            pre.nz[ii] = (ACT == 1 && noise) ? *reinterpret_cast<const float4*>(noise + opix + (int64_t)ii * W) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto out_items = [&](const OutPre& pre, int blk, int b, int64_t opix, int n) __attribute__((always_inline)) {
        if (skip & 8) return;
        const float* mp = mbuf + on16 * 32 + oj;
        float s[4][6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            float col[6];
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) col[ii] = mp[(ii * 6 + jj) * 512];
            float o[4];
            wino4_at(col, o);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) s[ii][jj] = o[ii];
        }
        const int64_t plane = (int64_t)b * N + n;
        float part = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float o[4];
            wino4_at(s[ii], o);
            if (DOT) {
                const float4 d = *reinterpret_cast<const float4*>(dot_with + plane * H * W + opix + (int64_t)ii * W);
                part += (o[0] * d.x + o[1] * d.y) + (o[2] * d.z + o[3] * d.w);
            }
            float4 r = make_float4(o[0] * pre.os, o[1] * pre.os, o[2] * pre.os, o[3] * pre.os);
            if (ACT == 1) {
                const float4 nz = pre.nz[ii];
                r.x += nw * nz.x + pre.bs, r.y += nw * nz.y + pre.bs, r.z += nw * nz.z + pre.bs, r.w += nw * nz.w + pre.bs;
                r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
                r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
            }
            if (ACT == 2) {
                r.x += pre.bs, r.y += pre.bs, r.z += pre.bs, r.w += pre.bs;
                r.x = r.x > 0.f ? r.x : pre.sl * r.x, r.y = r.y > 0.f ? r.y : pre.sl * r.y;
                r.z = r.z > 0.f ? r.z : pre.sl * r.z, r.w = r.w > 0.f ? r.w : pre.sl * r.w;
            }
            *reinterpret_cast<float4*>(y + plane * H * W + opix + (int64_t)ii * W) = r;
        }
        if (DOT) {  // the 32 tiles of a half-wave share (b, n): one partial per (channel, spatial block of the image), summed by the caller
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (oj == 0) dot_partial[((int64_t)b * N + n) * per_img + (blk - b * per_img)] = part;
        }
    };
    auto block_pix = [&](int blk, int& b, int64_t& opix) __attribute__((always_inline)) {
        b = blk / per_img;
        const int rem = blk - b * per_img;
        const int by = rem / bx_n, bx = rem - by * bx_n;
        opix = (int64_t)(by * 16 + 4 * (oj >> 3)) * W + bx * 32 + 4 * (oj & 7);
    };
    // The two roles are two separate loops with the SAME sequence of barriers (per block: KC matrix ticks, then 4 output ticks), so
    // that each gets its own register allocation: accumulators there, window sets here.
    if (wave >= 4) {
        // ---------------------------------------------------------------------------------------------- loaders
        // Static schedule (so that the window sets are compile-time registers and the compiler's vmcnt waits stay exact): chunk kc of a
        // block lives in set kc % 4 and is produced at a fixed tick -- chunk c+1 at matrix tick c (c >= 1), the NEXT block's chunk 0
        // at the last matrix tick and its chunk 1 at the first output tick; matrix tick 0 is idle.  The loads of the chunk four
        // positions later are issued right behind each transform.  KC % 4 == 0.
        const int tj = tid & 31, tch = (tid >> 5) & 7;  // tile of a block, channel of a chunk
        float w0[36], w1[36], w2[36], w3[36];
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)(unsigned)((int64_t)B * K * H * W * 4), 0x00020000);
        auto issue = [&](float (&win)[36], int bi, int kc) __attribute__((always_inline)) {  // chunk kc of this workgroup's block number bi
            const bool live = bi < count && !((skip & 1) && (bi > 0 || kc >= 4));
            const int blk = (int)blockIdx.x + (live ? bi : 0) * (int)gridDim.x;
            const int b = blk / per_img, rem = blk - b * per_img;
            const int by = rem / bx_n, bx = rem - by * bx_n;
            const unsigned base = live ? (unsigned)(((int64_t)b * K + kc * 8 + tch) * H * W * 4) : 0xfffffff0u;
            wf2_load(win, rx, base, by * 16 + 4 * (tj >> 3), bx * 32 + 4 * (tj & 7), H, W);
        };
        auto produce = [&](float (&win)[36], int bi, int kc) __attribute__((always_inline)) {  // -> V[kc & 1] (KC is even); then the loads 4 chunks on
            if (bi < count) {
                const int blk = (int)blockIdx.x + bi * (int)gridDim.x;
                const int b = blk / per_img;
                const float sc = in_scale ? in_scale[(int64_t)b * K + kc * 8 + tch] : 1.f;
                if (!(skip & 2)) wf2_transform(win, wsm + (kc & 1) * VS + ((tch & 1) * 32 + tj) * 4 + (tch >> 1), sc);
            }
            const int nk = kc + 4;
            issue(win, nk >= KC ? bi + 1 : bi, nk & (KC - 1));
        };
        issue(w0, 0, 0), issue(w1, 0, 1), issue(w2, 0, 2), issue(w3, 0, 3);
        produce(w0, 0, 0), produce(w1, 0, 1);
        __syncthreads();
        for (int i = 0; i < count; ++i) {
            __syncthreads();  // matrix tick 0: idle
            for (int c = 1; c < KC - 1; c += 4) {  // matrix ticks c, c+1 (and c+2, c+3 unless they are past the last but one): chunks c+1 ...
                produce(w2, i, c + 1);
                __syncthreads();
                produce(w3, i, c + 2);
                __syncthreads();
                if (c + 3 < KC) {
                    produce(w0, i, c + 3);
                    __syncthreads();
                    produce(w1, i, c + 4);
                    __syncthreads();
                }
            }
            produce(w0, i + 1, 0);  // the last matrix tick (KC - 1)
            __syncthreads();
            const int blk = (int)blockIdx.x + i * (int)gridDim.x;
            int b;
            int64_t opix;
            block_pix(blk, b, opix);
            OutPre pre;
            // output tick A0 (the next block's chunk 1 first), B0, A1, B1
            produce(w1, i + 1, 1);
            out_prefetch(pre, b, opix, n0 + on16);
            __syncthreads();
            out_items(pre, blk, b, opix, n0 + on16);
            out_prefetch(pre, b, opix, n0 + 16 + on16);
            __syncthreads();
            __syncthreads();
            out_items(pre, blk, b, opix, n0 + 16 + on16);
            __syncthreads();
        }
        return;
    }
    // -------------------------------------------------------------------------------------------------- matrix waves
    const int half = lane >> 5, j = lane & 31;
    const int g = wave;
    wf_f32x16 acc[9];
    const float4* uf4 = reinterpret_cast<const float4*>(uf);
    // A operands: one float4 per position, loaded ONE TICK AHEAD and in place -- a[q] is reloaded with the next chunk's weights right
    // behind the four MFMAs that consumed it (the last tick of a block fetches chunk 0 again: the same weights serve the next block)
    float4 a[9];
    auto a_at = [&](int q, int kc) __attribute__((always_inline)) {
        return uf4[((((int64_t)(9 * g + q) << kc_log2) + kc) * 2 + half) * N + n0 + j];
    };
    auto mfma_tick = [&](int stage, int next_kc) __attribute__((always_inline)) {
        const float4* vs4 = reinterpret_cast<const float4*>(wsm + stage * VS);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 b4 = vs4[((9 * g + q) * 2 + half) * 32 + j];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b4.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b4.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b4.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b4.w, acc[q], 0, 0, 0);
            a[q] = a_at(q, next_kc);
        }
    };
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = a_at(q, 0);
    __syncthreads();  // (prologue tick)
    for (int i = 0; i < count; ++i) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int c = 0; c < KC; ++c) {
            if (!(skip & 4)) mfma_tick(c & 1, (c + 1) & (KC - 1));
            __syncthreads();
        }
        const int blk = (int)blockIdx.x + i * (int)gridDim.x;
        int b;
        int64_t opix;
        block_pix(blk, b, opix);
        OutPre pre;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {  // (unrolled: the accumulator registers are indexed by q2)
            // accumulator register r of lane (half, j) is row (r & 3) + 8 * (r >> 2) + 4 * half: round q2 moves the registers
            // 8*q2 .. 8*q2+7 = rows 16*q2 .. 16*q2+15, row within the round n16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half
            out_prefetch(pre, b, opix, n0 + 16 * q2 + on16);
            if (!(skip & 8)) {
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
                        mbuf[((9 * g + q) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * half) * 32 + j] = acc[q][8 * q2 + rr];
            }
            __syncthreads();
            out_items(pre, blk, b, opix, n0 + 16 * q2 + on16);
            __syncthreads();
        }
    }
}


// ------------------------------------------------------------------ F(4x4, 3x3), fused, persistent, patch staged by LDS-DMA
// Version 3: the structure of wino4_fused2_kernel (4 matrix waves + 4 transform waves in lock-step ticks, persistent over blocks), with
// the input no longer fetched window by window into registers.  The transform waves stage the RAW 8-channel patch of a block --
// [8 ch][18 rows][10 quads] floats, image columns bx*32-4 .. bx*32+35 so that every 16-byte quad is either wholly inside the image
// row or wholly outside it (zero padding = an out-of-range offset, exact) -- by `buffer_load_dwordx4 ... lds` into a ring of three
// stages, three chunks ahead of the MFMAs, with explicit vmcnt counts: no registers, no compiler-managed waits, 6 DMA instructions per
// wave and chunk instead of 18 window loads per thread.  A transform thread reads its 6x6 window from the stage (one ds_read_b128 +
// two ds_read_b32 per row).  Per block: a pre-tick (chunk 0 -> V[0]), KC matrix ticks (MFMAs of chunk c | transform of chunk c+1,
// DMA of chunk g+3), then two output rounds of 16 channels through the V space, in which all 512 threads own a (channel, tile) pair.
// TXN: tiles of a block along x (8: blocks of 32 x 16 pixels; 16: 64 x 8 -- longer row segments per DMA, fewer cache lines per byte).
template <int ACT, bool DOT, int TXN>
__global__ __launch_bounds__(512, 1) void wino4_fused3_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ uf, const float* __restrict__ out_scale,
                                                              float* __restrict__ y, int B, int K, int N, int H, int W, int kc_log2,
                                                              int n_blocks, const float* __restrict__ noise,
                                                              const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                              const float* __restrict__ slope, const float* __restrict__ dot_with,
                                                              float* __restrict__ dot_partial, int skip) {
#ifndef W2E_TUNING
    skip = 0;  // (tuning builds only: bit 0 no DMA after the prologue's, 1 no transform, 2 no MFMAs, 3 no output rounds)
#endif
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int VS = 36 * 2 * 32 * 4;   // floats of one V stage; the two stages together are the output buffer M[36][16][32]
    constexpr int TYN = 32 / TXN, BWP = 4 * TXN, BHP = 4 * TYN;  // tiles along y; block width / height in pixels
    constexpr int PR = BHP + 2, PC = BWP + 8, QR = PC / 4;        // patch rows, columns (image columns bx*BWP-4 ...), quads per row
    constexpr int PQ = 8 * PR * QR;       // quads of one patch chunk (1440 for both shapes)
    static_assert(PR * PC == 720 && PQ <= 24 * 64, "patch geometry");
    constexpr int PS = 24 * 64 * 4;       // floats of one ring stage: 24 wave-instructions x 64 lanes x 16 B (>= PQ quads)
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // V[2][VS], ring[3][PS], in_scale table [2][256]
    float* const mbuf = wsm;
    float* const ring = wsm + 2 * VS;
    float* const sctab = ring + 3 * PS;  // in_scale[b, .] of the current / the next block, written by the matrix waves: the transform
                                         // waves issue no compiler-managed global load at all (its wait would be vmcnt(0) and drain the
                                         // DMA ring) -- the epilogue operands come the same way:
    float* const etab = sctab + 512;     // the block's noise patch [16][32], then out_scale / bias / slope of the 32 channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx_n = W / BWP, by_n = H / BHP, per_img = bx_n * by_n;
    const int KC = 1 << kc_log2;
    const int count = (n_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = count << kc_log2;
    const int n0 = blockIdx.y * 32;
    const int oj = tid & 31, on16 = tid >> 5;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
    auto out_items = [&](int blk, int b, int64_t opix, int nloc) __attribute__((always_inline)) {  // nloc: channel within the block of 32
        const int n = n0 + nloc;
        const float os = etab[512 + nloc], bs = etab[544 + nloc], sl = etab[576 + nloc];
        const float* mp = mbuf + (nloc & 15) * 32 + oj;
        float s[4][6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            float col[6];
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) col[ii] = mp[(ii * 6 + jj) * 512];
            float o[4];
            wino4_at(col, o);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) s[ii][jj] = o[ii];
        }
        const int64_t plane = (int64_t)b * N + n;
        float part = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float o[4];
            wino4_at(s[ii], o);
            if (DOT) {
                const float4 d = *reinterpret_cast<const float4*>(dot_with + plane * H * W + opix + (int64_t)ii * W);
                part += (o[0] * d.x + o[1] * d.y) + (o[2] * d.z + o[3] * d.w);
            }
            float4 r = make_float4(o[0] * os, o[1] * os, o[2] * os, o[3] * os);
            if (ACT == 1) {
                const float4 nz = *reinterpret_cast<const float4*>(etab + (4 * (oj / TXN) + ii) * BWP + 4 * (oj % TXN));
                r.x += nw * nz.x + bs, r.y += nw * nz.y + bs, r.z += nw * nz.z + bs, r.w += nw * nz.w + bs;
                r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
                r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
            }
            if (ACT == 2) {
                r.x += bs, r.y += bs, r.z += bs, r.w += bs;
                r.x = r.x > 0.f ? r.x : sl * r.x, r.y = r.y > 0.f ? r.y : sl * r.y;
                r.z = r.z > 0.f ? r.z : sl * r.z, r.w = r.w > 0.f ? r.w : sl * r.w;
            }
            *reinterpret_cast<float4*>(y + plane * H * W + opix + (int64_t)ii * W) = r;
        }
        if (DOT) {  // one partial per (channel, spatial block of the image), summed by the caller
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (oj == 0) dot_partial[((int64_t)b * N + n) * per_img + (blk - b * per_img)] = part;
        }
    };
    auto block_pix = [&](int blk, int& b, int64_t& opix) __attribute__((always_inline)) {
        b = blk / per_img;
        const int rem = blk - b * per_img;
        const int by = rem / bx_n, bx = rem - by * bx_n;
        opix = (int64_t)(by * BHP + 4 * (oj / TXN)) * W + bx * BWP + 4 * (oj % TXN);
    };
    if (wave >= 4) {
        // ---------------------------------------------------------------------------------------------- transform waves
        const int tw = wave - 4;
        const int tj = tid & 31, tch = (tid >> 5) & 7;  // tile of a block, channel of a chunk
        const uint64_t a64 = (uint64_t)(uintptr_t)x;
        i32x4 qx;
        qx[0] = (int)(unsigned)a64, qx[1] = (int)(unsigned)((a64 >> 32) & 0xffffu), qx[2] = (int)(unsigned)((int64_t)B * K * H * W * 4), qx[3] = 0x00020000;
        qx[0] = __builtin_amdgcn_readfirstlane(qx[0]), qx[1] = __builtin_amdgcn_readfirstlane(qx[1]);
        qx[2] = __builtin_amdgcn_readfirstlane(qx[2]), qx[3] = __builtin_amdgcn_readfirstlane(qx[3]);
        const unsigned lds_ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
        const unsigned plane_bytes = (unsigned)(H * W) * 4u;
        // this lane's quad of each of its wave's 6 DMA instructions: qd = (tw*6 + s)*64 + lane -> (channel, patch row, quad)
        int q_ch[6], q_row[6], q_q[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
            const int qd = (tw * 6 + s6) * 64 + lane;
            q_ch[s6] = qd < PQ ? qd / (PR * QR) : -1;
            const int rem = qd - (qd / (PR * QR)) * (PR * QR);
            q_row[s6] = rem / QR, q_q[s6] = rem - (rem / QR) * QR;
        }
        unsigned voff[6];  // byte offsets inside the image (b) of the block being ISSUED, or the out-of-range marker
        int voff_blk = -1;
        auto set_voff = [&](int bi) __attribute__((always_inline)) {
            const int blk = (int)blockIdx.x + bi * (int)gridDim.x;
            const int b = blk / per_img, rem = blk - b * per_img;
            const int by = rem / bx_n, bx = rem - by * bx_n;
            (void)b;
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const int iy = by * BHP - 1 + q_row[s6], ix = bx * BWP - 4 + 4 * q_q[s6];
                const bool ok = q_ch[s6] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
                voff[s6] = ok ? (unsigned)q_ch[s6] * plane_bytes + (unsigned)(iy * W + ix) * 4u : 0xfffffff0u;
            }
        };
        auto issue = [&](int g) __attribute__((always_inline)) {  // DMA of global chunk g into ring stage g % 3
            const int bi = g >> kc_log2, kc = g & (KC - 1);
            if (bi != voff_blk) set_voff(bi), voff_blk = bi;
            const int blk = (int)blockIdx.x + bi * (int)gridDim.x;
            const int b = blk / per_img;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b * K + kc * 8) * plane_bytes));
            const unsigned stage_b = lds_ring + (unsigned)(g % 3) * (unsigned)(PS * 4) + (unsigned)(tw * 6) * 1024u;
            (void)soff, (void)stage_b;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(stage_b + (unsigned)s6 * 1024u));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff[s6]), "s"(qx), "s"(soff) : "memory");
            }
#endif
        };
        auto transform = [&](int bi, int kc, int g) __attribute__((always_inline)) {  // chunk (bi, kc) = global g: ring stage g % 3 -> V[kc & 1]
            const float sc = sctab[(bi & 1) * 256 + kc * 8 + tch];
            const float* pp = ring + (g % 3) * PS + tch * 720 + (4 * (tj / TXN)) * PC + 4 * (tj % TXN);
            float t[6][6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float4 mid = *reinterpret_cast<const float4*>(pp + r * PC + 4);
                const float d[6] = {pp[r * PC + 3], mid.x, mid.y, mid.z, mid.w, pp[r * PC + 8]};
                wino4_bt(d, t[r]);
            }
            float* vp = wsm + (kc & 1) * VS + ((tch & 1) * 32 + tj) * 4 + (tch >> 1);
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const float col[6] = {t[0][jj], t[1][jj], t[2][jj], t[3][jj], t[4][jj], t[5][jj]};
                float o[6];
                wino4_bt(col, o);
#pragma unroll
                for (int i6 = 0; i6 < 6; ++i6) vp[(i6 * 6 + jj) * 256] = sc * o[i6];
            }
        };
        // prologue: chunks 0, 1, 2 on their way; 0 and 1 landed
        issue(0);
        if (1 < total) issue(1);
        if (2 < total) issue(2);
        if (2 < total) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6): chunk 2 may still fly
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        for (int i = 0; i < count; ++i) {
            const int g0 = i << kc_log2;
            if (!(skip & 2)) transform(i, 0, g0);  // pre-tick
            __syncthreads();
            for (int c = 0; c < KC; ++c) {
                const int g = g0 + c;
                const bool more = g + 3 < total && !(skip & 1);
                if (more) issue(g + 3);  // (its stage held chunk g, transformed one tick ago)
                if (c + 1 < KC && !(skip & 2)) transform(i, c + 1, g + 1);
                if (more) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6): everything but the chunk just issued has landed
                else __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
            }
            const int blk = (int)blockIdx.x + i * (int)gridDim.x;
            int b;
            int64_t opix;
            block_pix(blk, b, opix);
            // (with the fused dot an item loads 16 floats of dot_with from global memory: a compiler-managed wait that drains this wave's
            // DMA ring once per block; leaving all items to the matrix waves instead measured slower: 1.04 vs 0.98 ms, 32 @ 1024^2 batch 8)
            __syncthreads();  // A0
            if (!(skip & 8)) out_items(blk, b, opix, on16);
            __syncthreads();  // B0
            __syncthreads();  // A1
            if (!(skip & 8)) out_items(blk, b, opix, 16 + on16);
            __syncthreads();  // B1
        }
        return;
    }
    // -------------------------------------------------------------------------------------------------- matrix waves
    const int half = lane >> 5, j = lane & 31;
    const int g = wave;
    wf_f32x16 acc[9];
    const float4* uf4 = reinterpret_cast<const float4*>(uf);
    float4 a[9];  // A operands, loaded one tick ahead and in place (wino4_fused2_kernel)
    auto a_at = [&](int q, int kc) __attribute__((always_inline)) {
        return uf4[((((int64_t)(9 * g + q) << kc_log2) + kc) * 2 + half) * N + n0 + j];
    };
    auto mfma_tick = [&](int stage, int next_kc) __attribute__((always_inline)) {
        const float4* vs4 = reinterpret_cast<const float4*>(wsm + stage * VS);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 b4 = vs4[((9 * g + q) * 2 + half) * 32 + j];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b4.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b4.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b4.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b4.w, acc[q], 0, 0, 0);
            a[q] = a_at(q, next_kc);
        }
    };
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = a_at(q, 0);
    auto fill_sctab = [&](int bi) __attribute__((always_inline)) {  // in_scale[b, 0..K) of block number bi (K <= 256)
        if (bi >= count) return;
        const int b = ((int)blockIdx.x + bi * (int)gridDim.x) / per_img;
        if (tid < K) sctab[(bi & 1) * 256 + tid] = in_scale ? in_scale[(int64_t)b * K + tid] : 1.f;
    };
    auto fill_etab = [&](int bi) __attribute__((always_inline)) {  // noise patch and per-channel epilogue operands of block number bi
        const int blk = (int)blockIdx.x + bi * (int)gridDim.x;
        const int b = blk / per_img, rem = blk - b * per_img;
        const int by = rem / bx_n, bx = rem - by * bx_n;
        if (tid < 128) {
            const int row = tid / TXN, qd = tid % TXN;
            *reinterpret_cast<float4*>(etab + row * BWP + 4 * qd) =
                (ACT == 1 && noise) ? *reinterpret_cast<const float4*>(noise + (int64_t)(by * BHP + row) * W + bx * BWP + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (tid < 160) {
            etab[512 + tid - 128] = out_scale ? out_scale[(int64_t)b * N + n0 + tid - 128] : 1.f;
        } else if (tid < 192) {
            etab[544 + tid - 160] = (ACT && bias) ? bias[n0 + tid - 160] : 0.f;
        } else if (tid < 224) {
            etab[576 + tid - 192] = (ACT == 2 && slope) ? slope[n0 + tid - 192] : 1.f;
        }
    };
    fill_sctab(0);
    __syncthreads();  // (prologue)
    for (int i = 0; i < count; ++i) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        fill_etab(i);     // (the previous block's output rounds are behind the last barrier; this block's come KC ticks later)
        __syncthreads();  // pre-tick
        for (int c = 0; c < KC; ++c) {
            if (!(skip & 4)) mfma_tick(c & 1, (c + 1) & (KC - 1));
            __syncthreads();
        }
        const int blk = (int)blockIdx.x + i * (int)gridDim.x;
        int b;
        int64_t opix;
        block_pix(blk, b, opix);
        fill_sctab(i + 1);  // (the last transform of block i was one tick ago; the table of block i+1 is read from its pre-tick on)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {  // (unrolled: the accumulator registers are indexed by q2)
            if (!(skip & 8)) {
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
                        mbuf[((9 * g + q) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * half) * 32 + j] = acc[q][8 * q2 + rr];
            }
            __syncthreads();  // A
            if (!(skip & 8)) out_items(blk, b, opix, 16 * q2 + on16);
            __syncthreads();  // B
        }
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_wino_weights(const float* wp, float* u, int k_ch, int n_ch, int m, void* stream) {
    W2E_REQUIRE(wp && u, "wino_weights: null tensor");
    W2E_REQUIRE(k_ch > 0 && n_ch > 0, "wino_weights: bad dims %d %d", k_ch, n_ch);
    W2E_REQUIRE(m == 2 || m == 4, "wino_weights: output tile %d (2 or 4)", m);
    const int64_t total = (int64_t)k_ch * n_ch;
    if (m == 2) wino_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, u, k_ch, n_ch);
    else wino4_weights_kernel<false><<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, u, k_ch, n_ch);
    W2E_LAUNCH_CHECK("wino_weights");
    return 0;
}

int w2e_wino_input(const float* x, const float* in_scale, float* v, int batch, int k_ch, int h, int w, int m, void* stream) {
    W2E_REQUIRE(x && v, "wino_input: null tensor");
    W2E_REQUIRE(m == 2 || m == 4, "wino_input: output tile %d (2 or 4)", m);
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && h >= m && w >= m && h % m == 0 && w % m == 0, "wino_input: bad dims (H, W must be multiples of %d)", m);
    W2E_REQUIRE(((uintptr_t)x & (m == 2 ? 7 : 15)) == 0, "wino_input: x must be %d-byte aligned", 4 * m);
    if (batch == 0) return 0;
    const int tiles = (h / m) * (w / m);
    W2E_REQUIRE((int64_t)batch * k_ch < 65536, "wino_input: more than 65535 planes");
    dim3 grid((unsigned)ceil_div(tiles, 256), (unsigned)(batch * k_ch));
    if (m == 2) wino_input_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, in_scale, v, batch, k_ch, h, w);
    else wino4_input_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, in_scale, v, batch, k_ch, h, w);
    W2E_LAUNCH_CHECK("wino_input");
    return 0;
}

int w2e_wino_output(const float* mm, const float* out_scale, float* y, int batch, int n_ch, int h, int w, int m, int act,
                    const float* noise, const float* noise_w, const float* bias, const float* slope, const float* dot_with,
                    float* dot_out, void* stream) {
    W2E_REQUIRE(mm && y, "wino_output: null tensor");
    W2E_REQUIRE(m == 2 || m == 4, "wino_output: output tile %d (2 or 4)", m);
    W2E_REQUIRE(act >= 0 && act <= 2, "wino_output: epilogue %d (0 none, 1 StyledConv, 2 bias + PReLU)", act);
    W2E_REQUIRE(batch >= 0 && n_ch > 0 && n_ch < 65536 && h >= m && w >= m && h % m == 0 && w % m == 0,
                "wino_output: bad dims (H, W must be multiples of %d)", m);
    const int tiles = (h / m) * (w / m);
    // the fused dot reduces over the lanes of a wave that share a (b, n) plane: whole waves (F(2x2)) or a power-of-two segment (F(4x4))
    W2E_REQUIRE(!dot_with || (m == 4 ? ((tiles & 63) == 0 || (tiles < 64 && (tiles & (tiles - 1)) == 0)) : (tiles & 63) == 0),
                "wino_output: fused dot with %d tiles per plane (a multiple of 64%s)", tiles, m == 4 ? ", or a power of two below it" : "");
    W2E_REQUIRE(!(act && dot_with), "wino_output: the activation epilogues and the fused dot exclude each other");
    W2E_REQUIRE(!dot_with || dot_out, "wino_output: dot_with without dot_out");
    W2E_REQUIRE(!noise || noise_w, "wino_output: noise without noise_w");
    W2E_REQUIRE(act == 1 || !noise, "wino_output: noise belongs to epilogue 1");
    W2E_REQUIRE(act == 2 || !slope, "wino_output: slope belongs to epilogue 2");
    W2E_REQUIRE((((uintptr_t)y | (uintptr_t)(dot_with ? dot_with : y) | (uintptr_t)(noise ? noise : y)) & (m == 2 ? 7 : 15)) == 0,
                "wino_output: y / dot_with / noise must be %d-byte aligned", 4 * m);
    if (batch == 0) return 0;
    const int64_t T = (int64_t)batch * tiles;
    dim3 grid((unsigned)ceil_div(T, 256), (unsigned)n_ch);
    hipStream_t s = (hipStream_t)stream;
    if (m == 2) {
        if (act == 1) wino_output_kernel<1, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, noise, noise_w, bias, nullptr, nullptr, nullptr);
        else if (act == 2) wino_output_kernel<2, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, nullptr, nullptr, bias, slope, nullptr, nullptr);
        else if (dot_with) wino_output_kernel<0, true><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, nullptr, nullptr, nullptr, nullptr, dot_with, dot_out);
        else wino_output_kernel<0, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    } else {
        const int seg = tiles < 64 ? tiles : 64;
        if (act == 1) wino4_output_kernel<1, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, seg, noise, noise_w, bias, nullptr, nullptr, nullptr);
        else if (act == 2) wino4_output_kernel<2, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, seg, nullptr, nullptr, bias, slope, nullptr, nullptr);
        else if (dot_with) wino4_output_kernel<0, true><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, seg, nullptr, nullptr, nullptr, nullptr, dot_with, dot_out);
        else wino4_output_kernel<0, false><<<grid, 256, 0, s>>>(mm, out_scale, y, batch, n_ch, h, w, seg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    W2E_LAUNCH_CHECK("wino_output");
    return 0;
}

int w2e_wino_weights_fused(const float* wp, float* uf, int k_ch, int n_ch, void* stream) {
    W2E_REQUIRE(wp && uf, "wino_weights_fused: null tensor");
    W2E_REQUIRE(k_ch > 0 && (k_ch & 7) == 0 && n_ch > 0 && (n_ch & 31) == 0, "wino_weights_fused: K %% 8 == 0, N %% 32 == 0 (got %d, %d)", k_ch, n_ch);
    const int64_t total = (int64_t)k_ch * n_ch;
    wino4_weights_kernel<true><<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, uf, k_ch, n_ch);
    W2E_LAUNCH_CHECK("wino_weights_fused");
    return 0;
}

int w2e_wino_fused(const float* x, const float* in_scale, const float* uf, const float* out_scale, float* y, int batch, int k_ch,
                   int n_ch, int h, int w, int act, const float* noise, const float* noise_w, const float* bias, const float* slope,
                   const float* dot_with, float* dot_out, int version, int wgs, void* stream) {
    W2E_REQUIRE(x && uf && y, "wino_fused: null tensor");
    W2E_REQUIRE(act >= 0 && act <= 2, "wino_fused: epilogue %d (0 none, 1 StyledConv, 2 bias + PReLU)", act);
    W2E_REQUIRE(version >= 1 && version <= 3, "wino_fused: version %d (1, 2 or 3)", version);
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && (k_ch & 7) == 0 && n_ch > 0 && (n_ch & 31) == 0, "wino_fused: K %% 8 == 0, N %% 32 == 0 (got %d, %d)", k_ch, n_ch);
    W2E_REQUIRE(h >= 16 && w >= 32 && (h & 15) == 0 && (w & 31) == 0, "wino_fused: H %% 16 == 0 and W %% 32 == 0 (got %d x %d)", h, w);
    W2E_REQUIRE(!(act && dot_with), "wino_fused: the activation epilogues and the fused dot exclude each other");
    W2E_REQUIRE(!dot_with || dot_out, "wino_fused: dot_with without dot_out");
    W2E_REQUIRE(!noise || noise_w, "wino_fused: noise without noise_w");
    W2E_REQUIRE(act == 1 || !noise, "wino_fused: noise belongs to epilogue 1");
    W2E_REQUIRE(act == 2 || !slope, "wino_fused: slope belongs to epilogue 2");
    W2E_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)uf | (uintptr_t)(dot_with ? dot_with : y) | (uintptr_t)(noise ? noise : y)) & 15) == 0,
                "wino_fused: x / y / uf / dot_with / noise must be 16-byte aligned");
    if (batch == 0) return 0;
    const int64_t blocks = (int64_t)batch * (h >> 4) * (w >> 5);
    W2E_REQUIRE(blocks < ((int64_t)1 << 31), "wino_fused: too many tile blocks");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)blocks;
    if (version == 2 || version == 3) {  // persistent, specialised waves: K a power of two >= 32; the fused dot leaves per-block partials
        W2E_REQUIRE(k_ch >= 32 && (k_ch & (k_ch - 1)) == 0, "wino_fused v2: K must be a power of two >= 32 (got %d)", k_ch);
        W2E_REQUIRE((int64_t)batch * k_ch * h * w * 4 < ((int64_t)1 << 32) - 64, "wino_fused v2: x exceeds 4 GB");
        int kc_log2 = 0;
        while ((8 << kc_log2) < k_ch) ++kc_log2;
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const int nby = n_ch / 32;
        int gx = cus / nby;
        if (gx < 1) gx = 1;
        if (gx > blocks) gx = (int)blocks;
        if ((wgs & 0xffff) > 0 && (wgs & 0xffff) < gx) gx = wgs & 0xffff;  // (tests: several blocks per workgroup on small inputs; bits 16+: a tuning build's skip mask)
        const dim3 g2((unsigned)gx, (unsigned)nby);
        if (version == 3) {
            W2E_REQUIRE(k_ch <= 256, "wino_fused v3: K <= 256 (got %d)", k_ch);
            const size_t lds3 = (size_t)(2 * 36 * 2 * 32 * 4 + 3 * 24 * 64 * 4 + 2 * 256 + 608) * 4;
            static unsigned done3[4];
            static unsigned done3w[4];
            const bool wide = (w & 63) == 0 && !((wgs >> 16) & 16);  // blocks of 64 x 8 pixels (bit 4 of a tuning build's mask: keep 32 x 16)
#define W2E_WF3(ACTv, DOTv, slot)                                                                                                          \
    do {                                                                                                                                   \
        if (wide) {                                                                                                                        \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused3_kernel<ACTv, DOTv, 16>, &done3w[slot]), "wino_fused: cannot enable %zu B of LDS", lds3); \
            wino4_fused3_kernel<ACTv, DOTv, 16><<<g2, 512, lds3, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, n_ch, h, w, kc_log2, (int)blocks, \
                                                                     noise, noise_w, bias, slope, dot_with, dot_out, wgs >> 16);           \
        } else {                                                                                                                           \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused3_kernel<ACTv, DOTv, 8>, &done3[slot]), "wino_fused: cannot enable %zu B of LDS", lds3); \
            wino4_fused3_kernel<ACTv, DOTv, 8><<<g2, 512, lds3, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, n_ch, h, w, kc_log2, (int)blocks, \
                                                                    noise, noise_w, bias, slope, dot_with, dot_out, wgs >> 16);            \
        }                                                                                                                                  \
    } while (0)
            if (act == 1) W2E_WF3(1, false, 0);
            else if (act == 2) W2E_WF3(2, false, 1);
            else if (dot_with) W2E_WF3(0, true, 2);
            else W2E_WF3(0, false, 3);
#undef W2E_WF3
            W2E_LAUNCH_CHECK("wino_fused (v3)");
            return 0;
        }
        const size_t lds2 = (size_t)4 * 36 * 2 * 32 * 4 * 4;
        static unsigned done2[4];
#define W2E_WF2(ACTv, DOTv, slot)                                                                                                          \
    do {                                                                                                                                   \
        W2E_REQUIRE(big_lds_once((const void*)wino4_fused2_kernel<ACTv, DOTv>, &done2[slot]), "wino_fused: cannot enable %zu B of LDS", lds2); \
        wino4_fused2_kernel<ACTv, DOTv><<<g2, 512, lds2, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, n_ch, h, w, kc_log2, (int)blocks, \
                                                             noise, noise_w, bias, slope, dot_with, dot_out, wgs >> 16);                   \
    } while (0)
        if (act == 1) W2E_WF2(1, false, 0);
        else if (act == 2) W2E_WF2(2, false, 1);
        else if (dot_with) W2E_WF2(0, true, 2);
        else W2E_WF2(0, false, 3);
#undef W2E_WF2
        W2E_LAUNCH_CHECK("wino_fused (v2)");
        return 0;
    }
    W2E_REQUIRE(n_ch == 32 || n_ch == 64, "wino_fused v1: N = 32 or 64 (got %d)", n_ch);
#define W2E_WF(NBv)                                                                                                                       \
    do {                                                                                                                                   \
        const size_t lds = (size_t)NBv * 36 * 8 * 32 * 4;                                                                                  \
        static unsigned done[3];                                                                                                           \
        if (act == 1) {                                                                                                                    \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused_kernel<NBv, 1, false>, &done[0]), "wino_fused: cannot enable %zu B of LDS", lds); \
            wino4_fused_kernel<NBv, 1, false><<<grid, 256 * NBv, lds, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, h, w, noise, noise_w, bias, nullptr, nullptr, nullptr); \
        } else if (act == 2) {                                                                                                             \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused_kernel<NBv, 2, false>, &done[1]), "wino_fused: cannot enable %zu B of LDS", lds); \
            wino4_fused_kernel<NBv, 2, false><<<grid, 256 * NBv, lds, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, h, w, nullptr, nullptr, bias, slope, nullptr, nullptr); \
        } else if (dot_with) {                                                                                                             \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused_kernel<NBv, 0, true>, &done[2]), "wino_fused: cannot enable %zu B of LDS", lds); \
            wino4_fused_kernel<NBv, 0, true><<<grid, 256 * NBv, lds, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, h, w, nullptr, nullptr, nullptr, nullptr, dot_with, dot_out); \
        } else {                                                                                                                           \
            static unsigned done0;                                                                                                         \
            W2E_REQUIRE(big_lds_once((const void*)wino4_fused_kernel<NBv, 0, false>, &done0), "wino_fused: cannot enable %zu B of LDS", lds); \
            wino4_fused_kernel<NBv, 0, false><<<grid, 256 * NBv, lds, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, h, w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); \
        }                                                                                                                                  \
    } while (0)
    if (n_ch == 32) W2E_WF(1);
    else W2E_WF(2);
#undef W2E_WF
    W2E_LAUNCH_CHECK("wino_fused");
    return 0;
}

}  // extern "C"
