// K7: the latent mapper MLPs of mapper/latent_mappers.py:10-82 (LevelsMapper: three Mappers = PixelNorm + 4 x EqualLinear(512,512,
// lr_mul 0.01, fused_lrelu) on the coarse / medium / fine slices of W+) -- the only TRAINED network of the path, forward, input
// gradient and weight / bias gradients.  The products are [B*L, 512] x [512, 512] with B*L = 16..80 rows per level: far too small
// for the matrix cores to matter (42 MFLOP per layer) and, as three rocBLAS calls + a bias/activation launch per layer and
// direction, dominated by launch latency (about 120 launches per step).  Here a layer of ALL levels is one launch per direction:
// fp32 FMA kernels, the level ("group") of a workgroup chosen by blockIdx.y.
//
// Activations live as [R, 512] with R = B*18 rows in group-major order: group g holds rows r0_g .. r0_g + B*L_g - 1, row
// r0_g + b*L_g + l = latent l0_g + l of sample b.
#include "common.h"

namespace w2e {

constexpr int MAP_D = 512;       // latent width
constexpr int MAP_MAXG = 4;      // groups (levels) per launch
constexpr float MAP_SLOPE = 0.2f, MAP_GAIN = 1.4142135623730951f;

struct MapGroups {
    int groups;
    int l0[MAP_MAXG], len[MAP_MAXG];   // latent range of a group: [l0, l0 + len)
    const float* w[MAP_MAXG];          // this layer's weight [512,512] (row n = output feature n) per group
    const float* bias[MAP_MAXG];       // [512] or null
    float* gw[MAP_MAXG];               // weight / bias gradient outputs (backward)
    float* gb[MAP_MAXG];
};

__device__ __forceinline__ int map_row0(const MapGroups& g, int grp, int batch) {
    int r = 0;
    for (int i = 0; i < grp; ++i) r += g.len[i] * batch;
    return r;
}

// PixelNorm over the LATENT axis of each level (latent_mappers.py:16 keeps PixelNorm's default dim=1, which on a [B,L,512] group
// is the layer axis -- Q2) + the gather into group-major rows:  h[r0_g + b*L + l, d] = x[b, l0_g + l, d] * rsqrt(mean_l x^2 + 1e-8)
__global__ __launch_bounds__(256) void mapper_pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ h, MapGroups g, int batch,
                                                              int n_latent) {
    const int grp = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;  // (b, d)
    if (e >= batch * MAP_D) return;
    const int b = e / MAP_D, d = e - b * MAP_D;
    const int L = g.len[grp], l0 = g.l0[grp];
    const float* xp = x + ((int64_t)b * n_latent + l0) * MAP_D + d;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += xp[(int64_t)l * MAP_D] * xp[(int64_t)l * MAP_D];
    const float r = rsqrtf(s / L + 1e-8f);
    float* hp = h + ((int64_t)map_row0(g, grp, batch) + (int64_t)b * L) * MAP_D + d;
    for (int l = 0; l < L; ++l) hp[(int64_t)l * MAP_D] = xp[(int64_t)l * MAP_D] * r;
}

// One EqualLinear of every group:  out[m, n] = epi( scale * sum_k a[m, k] * W_g[n, k] )   (rows m of group g, 16 columns n per
// workgroup; W rows are staged in LDS, the rows of a are read as float4 through L1).
//   MODE 0 (forward):        a = h_in,  epi(v) = lrelu(v + bias_g[n]*b_scale) * sqrt2                    (model.py:151-158)
//   MODE 1 (input gradient): a = gy .* lrelu'(y)  (ACT_A: the activation backward of the layer's OUTPUT y applied to the operand
//                            on the way),  W_g = the TRANSPOSED weight,  epi(v) = v
//   `scatter`: write row r0_g + b*L + l to out[b, l0_g + l, :] of a [B, n_latent, 512] tensor instead (the last forward layer).
constexpr int MAP_TN = 16, MAP_PITCH = MAP_D + 4;
template <int MODE>
__global__ __launch_bounds__(256) void mapper_linear_kernel(const float* __restrict__ a, const float* __restrict__ y_act, float* __restrict__ out,
                                                           MapGroups g, int batch, int n_latent, float w_scale, float b_scale, int scatter) {
    __shared__ __attribute__((aligned(16))) float wt[MAP_TN * MAP_PITCH];
    const int grp = blockIdx.y, n0 = blockIdx.x * MAP_TN, tid = threadIdx.x;
    const float* W = g.w[grp];
    for (int i = tid; i < MAP_TN * (MAP_D / 4); i += 256) {  // 16 weight rows, float4 at a time
        const int n = i / (MAP_D / 4), q = i - n * (MAP_D / 4);
        *reinterpret_cast<float4*>(wt + n * MAP_PITCH + 4 * q) = *reinterpret_cast<const float4*>(W + (int64_t)(n0 + n) * MAP_D + 4 * q);
    }
    __syncthreads();
    const int L = g.len[grp], rows = L * batch, r0 = map_row0(g, grp, batch);
    const int n = tid & (MAP_TN - 1);
    const float bs = (MODE == 0 && g.bias[grp]) ? g.bias[grp][n0 + n] * b_scale : 0.f;
    for (int m = tid / MAP_TN; m < rows; m += 256 / MAP_TN) {
        const float4* ap = reinterpret_cast<const float4*>(a + (int64_t)(r0 + m) * MAP_D);
        const float4* yp = MODE == 1 ? reinterpret_cast<const float4*>(y_act + (int64_t)(r0 + m) * MAP_D) : nullptr;
        const float4* wp = reinterpret_cast<const float4*>(wt + n * MAP_PITCH);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll 4
        for (int k = 0; k < MAP_D / 4; ++k) {
            float4 av = ap[k];
            if (MODE == 1) {
                const float4 yv = yp[k];
                av.x *= MAP_GAIN * (yv.x > 0.f ? 1.f : MAP_SLOPE), av.y *= MAP_GAIN * (yv.y > 0.f ? 1.f : MAP_SLOPE);
                av.z *= MAP_GAIN * (yv.z > 0.f ? 1.f : MAP_SLOPE), av.w *= MAP_GAIN * (yv.w > 0.f ? 1.f : MAP_SLOPE);
            }
            const float4 wv = wp[k];
            acc0 += av.x * wv.x, acc1 += av.y * wv.y, acc2 += av.z * wv.z, acc3 += av.w * wv.w;
        }
        float v = ((acc0 + acc1) + (acc2 + acc3)) * w_scale;
        if (MODE == 0) {
            v += bs;
            v = (v > 0.f ? v : v * MAP_SLOPE) * MAP_GAIN;
        }
        if (scatter) {
            const int b = m / L, l = m - b * L;
            out[((int64_t)b * n_latent + g.l0[grp] + l) * MAP_D + n0 + n] = v;
        } else {
            out[(int64_t)(r0 + m) * MAP_D + n0 + n] = v;
        }
    }
}

// Weight and bias gradients of one EqualLinear of every group:
//   gW_g[n, k] = w_scale * sum_m gpre[m, n] * h_in[m, k],   gb_g[n] = b_scale * sum_m gpre[m, n],   gpre = gy .* lrelu'(y)
// (8 rows n per workgroup; gpre[., n0..n0+7] staged in LDS; a thread owns columns k = tid and tid + 256; rows m summed in
// ascending order: deterministic).  gy / y may be the [B, n_latent, 512] tensors of the LAST layer (`gathered` = 0) or group-major.
constexpr int MAP_WN = 8, MAP_MAXROWS = 18 * 64;
__global__ __launch_bounds__(256) void mapper_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ h_in,
                                                          MapGroups g, int batch, int n_latent, float w_scale, float b_scale, int gathered) {
    extern __shared__ float gp[];  // [rows][MAP_WN]
    const int grp = blockIdx.y, n0 = blockIdx.x * MAP_WN, tid = threadIdx.x;
    const int L = g.len[grp], rows = L * batch, r0 = map_row0(g, grp, batch);
    for (int i = tid; i < rows * MAP_WN; i += 256) {
        const int m = i / MAP_WN, j = i - m * MAP_WN;
        int64_t off;
        if (gathered) off = (int64_t)(r0 + m) * MAP_D + n0 + j;
        else {
            const int b = m / L, l = m - b * L;
            off = ((int64_t)b * n_latent + g.l0[grp] + l) * MAP_D + n0 + j;
        }
        gp[i] = gy[off] * MAP_GAIN * (y[off] > 0.f ? 1.f : MAP_SLOPE);
    }
    __syncthreads();
    float acc[MAP_WN][2];
#pragma unroll
    for (int j = 0; j < MAP_WN; ++j) acc[j][0] = acc[j][1] = 0.f;
    const float* hp = h_in + (int64_t)r0 * MAP_D;
    for (int m = 0; m < rows; ++m) {
        const float h0 = hp[(int64_t)m * MAP_D + tid], h1 = hp[(int64_t)m * MAP_D + tid + 256];
#pragma unroll
        for (int j = 0; j < MAP_WN; ++j) {
            const float gv = gp[m * MAP_WN + j];
            acc[j][0] += gv * h0, acc[j][1] += gv * h1;
        }
    }
    float* gw = g.gw[grp];
#pragma unroll
    for (int j = 0; j < MAP_WN; ++j) {
        gw[(int64_t)(n0 + j) * MAP_D + tid] = acc[j][0] * w_scale;
        gw[(int64_t)(n0 + j) * MAP_D + tid + 256] = acc[j][1] * w_scale;
    }
    if (tid < MAP_WN && g.gb[grp]) {
        float s = 0.f;
        for (int m = 0; m < rows; ++m) s += gp[m * MAP_WN + tid];
        g.gb[grp][n0 + tid] = s * b_scale;
    }
}

// gathered pre-activation operand for the LAST layer's input gradient: a[r0_g + b*L + l, :] = gy[b, l0_g + l, :] (the activation
// backward itself is applied by mapper_linear_kernel<1> from y) -- and y likewise
__global__ __launch_bounds__(256) void mapper_gather_kernel(const float* __restrict__ src, float* __restrict__ dst, MapGroups g, int batch, int n_latent) {
    const int grp = blockIdx.y;
    const int L = g.len[grp], rows = L * batch, r0 = map_row0(g, grp, batch);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * (MAP_D / 4); e += gridDim.x * 256) {
        const int m = e / (MAP_D / 4), q = e - m * (MAP_D / 4);
        const int b = m / L, l = m - b * L;
        reinterpret_cast<float4*>(dst + (int64_t)(r0 + m) * MAP_D)[q] =
            reinterpret_cast<const float4*>(src + ((int64_t)b * n_latent + g.l0[grp] + l) * MAP_D)[q];
    }
}

// wt[j][k][n] = w_j[n][k] for the `count` 512x512 matrices given by pointer (32x32 tiles through LDS)
struct MapPtrs {
    const float* w[16];
};
__global__ __launch_bounds__(256) void mapper_transpose_kernel(MapPtrs ptrs, float* __restrict__ wt) {
    __shared__ float tile[32][33];
    const float* w = ptrs.w[blockIdx.z];
    float* o = wt + (int64_t)blockIdx.z * MAP_D * MAP_D;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = w[(int64_t)(by + i) * MAP_D + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) o[(int64_t)(bx + i) * MAP_D + by + tx] = tile[tx][i];
}

static bool fill_groups(MapGroups& g, int groups, const int* l0, const int* len, int batch, int n_latent) {
    if (groups < 1 || groups > MAP_MAXG || batch < 1) return false;
    int total = 0;
    g.groups = groups;
    for (int i = 0; i < groups; ++i) {
        if (len[i] < 1 || l0[i] < 0 || l0[i] + len[i] > n_latent || len[i] * batch > MAP_MAXROWS) return false;
        g.l0[i] = l0[i], g.len[i] = len[i];
        g.w[i] = nullptr, g.bias[i] = nullptr, g.gw[i] = nullptr, g.gb[i] = nullptr;
        total += len[i];
    }
    return total <= n_latent;
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_mapper_pixelnorm(const float* x, float* h, int batch, int n_latent, int groups, const int* l0, const int* len,
                                    void* stream) {
    W2E_REQUIRE(x && h && l0 && len, "mapper_pixelnorm: null argument");
    MapGroups g{};
    W2E_REQUIRE(fill_groups(g, groups, l0, len, batch, n_latent), "mapper_pixelnorm: bad groups");
    dim3 grid((unsigned)ceil_div((int64_t)batch * MAP_D, 256), (unsigned)groups);
    mapper_pixelnorm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, h, g, batch, n_latent);
    W2E_LAUNCH_CHECK("mapper_pixelnorm");
    return 0;
}

extern "C" int w2e_mapper_linear(int mode, const float* a, const float* y_act, float* out, const float* const* w,
                                 const float* const* bias, int batch, int n_latent, int groups, const int* l0, const int* len,
                                 float w_scale, float b_scale, int scatter, void* stream) {
    W2E_REQUIRE(a && out && w && l0 && len, "mapper_linear: null argument");
    W2E_REQUIRE(mode == 0 || (mode == 1 && y_act), "mapper_linear: mode 0 (forward) or 1 (input gradient, needs the layer output)");
    MapGroups g{};
    W2E_REQUIRE(fill_groups(g, groups, l0, len, batch, n_latent), "mapper_linear: bad groups");
    for (int i = 0; i < groups; ++i) {
        W2E_REQUIRE(w[i] && ((uintptr_t)w[i] & 15) == 0, "mapper_linear: weights must be 16-byte aligned");
        g.w[i] = w[i], g.bias[i] = (mode == 0 && bias) ? bias[i] : nullptr;
    }
    W2E_REQUIRE(((uintptr_t)a & 15) == 0 && (!y_act || ((uintptr_t)y_act & 15) == 0), "mapper_linear: operands must be 16-byte aligned");
    dim3 grid(MAP_D / MAP_TN, (unsigned)groups);
    if (mode == 0) mapper_linear_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a, nullptr, out, g, batch, n_latent, w_scale, b_scale, scatter);
    else mapper_linear_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a, y_act, out, g, batch, n_latent, w_scale, b_scale, scatter);
    W2E_LAUNCH_CHECK("mapper_linear");
    return 0;
}

extern "C" int w2e_mapper_wgrad(const float* gy, const float* y, const float* h_in, float* const* gw, float* const* gb, int batch,
                                int n_latent, int groups, const int* l0, const int* len, float w_scale, float b_scale, int gathered,
                                void* stream) {
    W2E_REQUIRE(gy && y && h_in && gw && l0 && len, "mapper_wgrad: null argument");
    MapGroups g{};
    W2E_REQUIRE(fill_groups(g, groups, l0, len, batch, n_latent), "mapper_wgrad: bad groups");
    int max_rows = 0;
    for (int i = 0; i < groups; ++i) {
        W2E_REQUIRE(gw[i], "mapper_wgrad: null weight gradient");
        g.gw[i] = gw[i], g.gb[i] = gb ? gb[i] : nullptr;
        if (len[i] * batch > max_rows) max_rows = len[i] * batch;
    }
    dim3 grid(MAP_D / MAP_WN, (unsigned)groups);
    mapper_wgrad_kernel<<<grid, 256, sizeof(float) * (size_t)max_rows * MAP_WN, (hipStream_t)stream>>>(gy, y, h_in, g, batch, n_latent, w_scale,
                                                                                                       b_scale, gathered);
    W2E_LAUNCH_CHECK("mapper_wgrad");
    return 0;
}

extern "C" int w2e_mapper_gather(const float* src, float* dst, int batch, int n_latent, int groups, const int* l0, const int* len,
                                 void* stream) {
    W2E_REQUIRE(src && dst && l0 && len, "mapper_gather: null argument");
    MapGroups g{};
    W2E_REQUIRE(fill_groups(g, groups, l0, len, batch, n_latent), "mapper_gather: bad groups");
    dim3 grid(64, (unsigned)groups);
    mapper_gather_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, dst, g, batch, n_latent);
    W2E_LAUNCH_CHECK("mapper_gather");
    return 0;
}

extern "C" int w2e_mapper_transpose(const float* const* w, int count, float* wt, void* stream) {
    W2E_REQUIRE(w && wt && count >= 1 && count <= 16, "mapper_transpose: 1..16 matrices");
    MapPtrs p{};
    for (int i = 0; i < count; ++i) {
        W2E_REQUIRE(w[i], "mapper_transpose: null matrix");
        p.w[i] = w[i];
    }
    dim3 grid(MAP_D / 32, MAP_D / 32, (unsigned)count);
    mapper_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, wt);
    W2E_LAUNCH_CHECK("mapper_transpose");
    return 0;
}
