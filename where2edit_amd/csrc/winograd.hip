// Winograd F(2x2, 3x3) form of the same-resolution modulated 3x3 convolution (K1w, include/w2e.h) for gfx950.
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
// the form pays where the contraction dominates the traffic: the 512-channel layers at 16^2 ... 64^2 (functional._wino_auto).
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
template <bool ACT, bool DOT>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ m, const float* __restrict__ out_scale,
                                                          float* __restrict__ y, int B, int N, int H, int W,
                                                          const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                          const float* __restrict__ bias, const float* __restrict__ dot_with,
                                                          float* __restrict__ dot_out) {
    const int TX = W >> 1, TY = H >> 1, tiles = TX * TY;
    const int64_t T = (int64_t)B * tiles;
    const int n = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;  // (whole waves: T % 64 == 0)
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
        if (ACT) {
            const float nw = noise ? noise_w[0] : 0.f;
            const float bs = bias ? bias[n] : 0.f;
            float2 nz = make_float2(0.f, 0.f);
            if (noise) nz = *reinterpret_cast<const float2*>(noise + pix + (int64_t)i * W);
            v0 += nw * nz.x + bs, v1 += nw * nz.y + bs;
            v0 = fmaxf(v0, 0.2f * v0) * 1.4142135623730951f, v1 = fmaxf(v1, 0.2f * v1) * 1.4142135623730951f;
        }
        *reinterpret_cast<float2*>(y + plane * H * W + pix + (int64_t)i * W) = make_float2(v0, v1);
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_wino_weights(const float* wp, float* u, int k_ch, int n_ch, void* stream) {
    W2E_REQUIRE(wp && u, "wino_weights: null tensor");
    W2E_REQUIRE(k_ch > 0 && n_ch > 0, "wino_weights: bad dims %d %d", k_ch, n_ch);
    const int64_t total = (int64_t)k_ch * n_ch;
    wino_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, u, k_ch, n_ch);
    W2E_LAUNCH_CHECK("wino_weights");
    return 0;
}

int w2e_wino_input(const float* x, const float* in_scale, float* v, int batch, int k_ch, int h, int w, void* stream) {
    W2E_REQUIRE(x && v, "wino_input: null tensor");
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && h >= 2 && w >= 2 && !(h & 1) && !(w & 1), "wino_input: bad dims (H, W must be even)");
    W2E_REQUIRE(((uintptr_t)x & 7) == 0, "wino_input: x must be 8-byte aligned");
    if (batch == 0) return 0;
    const int tiles = (h >> 1) * (w >> 1);
    W2E_REQUIRE((int64_t)batch * k_ch < 65536, "wino_input: more than 65535 planes");
    dim3 grid((unsigned)ceil_div(tiles, 256), (unsigned)(batch * k_ch));
    wino_input_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, in_scale, v, batch, k_ch, h, w);
    W2E_LAUNCH_CHECK("wino_input");
    return 0;
}

int w2e_wino_output(const float* m, const float* out_scale, float* y, int batch, int n_ch, int h, int w, int act,
                    const float* noise, const float* noise_w, const float* bias, const float* dot_with, float* dot_out,
                    void* stream) {
    W2E_REQUIRE(m && y, "wino_output: null tensor");
    W2E_REQUIRE(batch >= 0 && n_ch > 0 && n_ch < 65536 && h >= 2 && w >= 2 && !(h & 1) && !(w & 1), "wino_output: bad dims (H, W must be even)");
    W2E_REQUIRE((((h >> 1) * (w >> 1)) & 63) == 0, "wino_output: (H/2)*(W/2) must be a multiple of 64");
    W2E_REQUIRE(!(act && dot_with), "wino_output: the activation epilogue and the fused dot exclude each other");
    W2E_REQUIRE(!dot_with || dot_out, "wino_output: dot_with without dot_out");
    W2E_REQUIRE(!noise || noise_w, "wino_output: noise without noise_w");
    W2E_REQUIRE((((uintptr_t)y | (uintptr_t)(dot_with ? dot_with : y) | (uintptr_t)(noise ? noise : y)) & 7) == 0,
                "wino_output: y / dot_with / noise must be 8-byte aligned");
    if (batch == 0) return 0;
    const int64_t T = (int64_t)batch * (h >> 1) * (w >> 1);
    dim3 grid((unsigned)ceil_div(T, 256), (unsigned)n_ch);
    hipStream_t s = (hipStream_t)stream;
    if (act) wino_output_kernel<true, false><<<grid, 256, 0, s>>>(m, out_scale, y, batch, n_ch, h, w, noise, noise_w, bias, nullptr, nullptr);
    else if (dot_with) wino_output_kernel<false, true><<<grid, 256, 0, s>>>(m, out_scale, y, batch, n_ch, h, w, nullptr, nullptr, nullptr, dot_with, dot_out);
    else wino_output_kernel<false, false><<<grid, 256, 0, s>>>(m, out_scale, y, batch, n_ch, h, w, nullptr, nullptr, nullptr, nullptr, nullptr);
    W2E_LAUNCH_CHECK("wino_output");
    return 0;
}

}  // extern "C"
