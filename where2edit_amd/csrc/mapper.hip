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

// ---------------------------------------------------------------------------------------------------------------------------------
// The STYLE-SPACE mappers (mapper/latent_mappers.py:84-128: FullStyleSpaceMapper / WithoutToRGBStyleSpaceMapper): one Mapper per
// S-space code c, Mapper(latent_dim = C_c) = PixelNorm over the C_c FEATURES of a [B, C_c] input (dim = 1 of a 2-D tensor) + 4 x
// EqualLinear(C_c, C_c, lr_mul 0.01, fused_lrelu), C_c in {512, 256, 128, 64, 32}.  26 (17) independent MLPs on B rows each: as stock
// ops ~20 launches per code and direction.  Here a layer of ALL codes is one launch per direction (the code of a workgroup is
// blockIdx.y); activations are packed code-major, code c at float offset B * sum_{i<c} C_i, row-major [B, C_c].
namespace w2e {

constexpr int SS_MAXG = 32, SS_MAXB = 16;
struct SsGroups {
    int groups, batch;
    int dim[SS_MAXG], off[SS_MAXG];        // feature count of a code, float offset of its [batch, dim] block in the packed buffers
    const float* w[SS_MAXG];               // this layer's weight [dim, dim] (row n = output feature n)
    const float* bias[SS_MAXG];            // [dim]
    const float* src[SS_MAXG];             // pixelnorm: the code tensors [batch, dim]; gather: the incoming gradients (null = zeros)
    float* gw[SS_MAXG];
    float* gb[SS_MAXG];
    float w_scale[SS_MAXG];                // EqualLinear's scale = lr_mul / sqrt(dim): differs from code to code
};

// h[off_c + b*C + k] = x_c[b, k] * rsqrt(mean_k x_c[b, k]^2 + 1e-8)     (one wave per (code, row))
__global__ __launch_bounds__(256) void ss_pixelnorm_kernel(SsGroups g, float* __restrict__ h) {
    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = g.dim[c];
    for (int b = blockIdx.x * 4 + wave; b < g.batch; b += gridDim.x * 4) {
        const float* xp = g.src[c] + (int64_t)b * d;
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += xp[k] * xp[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float r = rsqrtf(s / d + 1e-8f);
        float* hp = h + g.off[c] + (int64_t)b * d;
        for (int k = lane; k < d; k += 64) hp[k] = xp[k] * r;
    }
}

// packed copy of a list of [batch, dim] tensors (the incoming gradients of the backward; a null entry = zeros)
__global__ __launch_bounds__(256) void ss_gather_kernel(SsGroups g, float* __restrict__ dst) {
    const int c = blockIdx.y, n = g.batch * g.dim[c];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) dst[g.off[c] + e] = g.src[c] ? g.src[c][e] : 0.f;
}

// forward layer: out[m, n] = lrelu(w_scale * sum_k a[m, k] W[n, k] + b_scale * bias[n]) * sqrt2.  One wave per output feature n
// (4 per workgroup): the lanes split k (coalesced weight-row reads), every row m of the code is accumulated in a register.
__global__ __launch_bounds__(256) void ss_linear_fwd_kernel(SsGroups g, const float* __restrict__ a, float* __restrict__ out, float b_scale) {
    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = g.dim[c], n = blockIdx.x * 4 + wave;
    if (n >= d) return;
    const float* wr = g.w[c] + (int64_t)n * d;
    const float* ap = a + g.off[c];
    float acc[SS_MAXB];
#pragma unroll
    for (int m = 0; m < SS_MAXB; ++m) acc[m] = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int m = 0; m < SS_MAXB; ++m)
            if (m < g.batch) acc[m] += ap[(int64_t)m * d + k] * wv;
    }
    const float bs = g.bias[c] ? g.bias[c][n] * b_scale : 0.f;
#pragma unroll
    for (int m = 0; m < SS_MAXB; ++m) {
        if (m >= g.batch) break;
        float s = acc[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        float v = s * g.w_scale[c] + bs;
        v = (v > 0.f ? v : v * MAP_SLOPE) * MAP_GAIN;
        if (lane == 0) out[g.off[c] + (int64_t)m * d + n] = v;
    }
}

// input gradient of a layer: gx[m, i] = w_scale * sum_o gpre[m, o] W[o, i],  gpre = gy .* lrelu'(y).  A thread owns input feature i
// (weight rows are read coalesced across i); gpre of the code is staged in LDS.
__global__ __launch_bounds__(256) void ss_linear_bwd_kernel(SsGroups g, const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx) {
    extern __shared__ float gp[];  // [batch][d]
    const int c = blockIdx.y, d = g.dim[c], i0 = blockIdx.x * 256;
    if (i0 >= d) return;
    for (int e = threadIdx.x; e < g.batch * d; e += 256) {
        const float yv = y[g.off[c] + e];
        gp[e] = gy[g.off[c] + e] * MAP_GAIN * (yv > 0.f ? 1.f : MAP_SLOPE);
    }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= d) return;
    float acc[SS_MAXB];
#pragma unroll
    for (int m = 0; m < SS_MAXB; ++m) acc[m] = 0.f;
    const float* W = g.w[c];
    for (int o = 0; o < d; ++o) {
        const float wv = W[(int64_t)o * d + i];
#pragma unroll
        for (int m = 0; m < SS_MAXB; ++m)
            if (m < g.batch) acc[m] += gp[m * d + o] * wv;
    }
#pragma unroll
    for (int m = 0; m < SS_MAXB; ++m)
        if (m < g.batch) gx[g.off[c] + (int64_t)m * d + i] = acc[m] * g.w_scale[c];
}

// weight / bias gradients of a layer: gW[o, i] = w_scale * sum_m gpre[m, o] h_in[m, i],  gb[o] = b_scale * sum_m gpre[m, o]
// (a thread owns one (o, i); rows summed in ascending order: deterministic)
__global__ __launch_bounds__(256) void ss_wgrad_kernel(SsGroups g, const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ h_in,
                                                      float b_scale) {
    const int c = blockIdx.y, d = g.dim[c];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)d * d) return;
    const int o = (int)(e / d), i = (int)(e - (int64_t)o * d);
    float s = 0.f, sb = 0.f;
    for (int m = 0; m < g.batch; ++m) {
        const int64_t at = g.off[c] + (int64_t)m * d;
        const float gv = gy[at + o] * MAP_GAIN * (y[at + o] > 0.f ? 1.f : MAP_SLOPE);
        s += gv * h_in[at + i];
        sb += gv;
    }
    g.gw[c][e] = s * g.w_scale[c];
    if (i == 0 && g.gb[c]) g.gb[c][o] = sb * b_scale;
}

static bool ss_fill(SsGroups& g, int groups, int batch, const int* dims) {
    if (groups < 1 || groups > SS_MAXG || batch < 1 || batch > SS_MAXB) return false;
    g.groups = groups, g.batch = batch;
    int off = 0;
    for (int c = 0; c < groups; ++c) {
        if (dims[c] < 1 || dims[c] > 4096) return false;
        g.dim[c] = dims[c], g.off[c] = off;
        off += batch * dims[c];
        g.w[c] = g.bias[c] = g.src[c] = nullptr, g.gw[c] = g.gb[c] = nullptr;
        g.w_scale[c] = 1.f;
    }
    return true;
}
static int ss_maxdim(const SsGroups& g) {
    int m = 0;
    for (int c = 0; c < g.groups; ++c) m = g.dim[c] > m ? g.dim[c] : m;
    return m;
}

}  // namespace w2e

extern "C" int w2e_ssmapper_pixelnorm(const float* const* x, float* h, int batch, int groups, const int* dims, void* stream) {
    W2E_REQUIRE(x && h && dims, "ssmapper_pixelnorm: null argument");
    SsGroups g{};
    W2E_REQUIRE(ss_fill(g, groups, batch, dims), "ssmapper_pixelnorm: 1..32 codes, batch 1..16");
    for (int c = 0; c < groups; ++c) {
        W2E_REQUIRE(x[c], "ssmapper_pixelnorm: null code tensor");
        g.src[c] = x[c];
    }
    ss_pixelnorm_kernel<<<dim3((unsigned)ceil_div(batch, 4), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(g, h);
    W2E_LAUNCH_CHECK("ssmapper_pixelnorm");
    return 0;
}

extern "C" int w2e_ssmapper_gather(const float* const* src, float* dst, int batch, int groups, const int* dims, void* stream) {
    W2E_REQUIRE(src && dst && dims, "ssmapper_gather: null argument");
    SsGroups g{};
    W2E_REQUIRE(ss_fill(g, groups, batch, dims), "ssmapper_gather: 1..32 codes, batch 1..16");
    for (int c = 0; c < groups; ++c) g.src[c] = src[c];
    ss_gather_kernel<<<dim3((unsigned)ceil_div((int64_t)batch * ss_maxdim(g), 256), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(g, dst);
    W2E_LAUNCH_CHECK("ssmapper_gather");
    return 0;
}

extern "C" int w2e_ssmapper_linear(int mode, const float* a, const float* y_act, float* out, const float* const* w, const float* const* bias,
                                   int batch, int groups, const int* dims, const float* w_scale, float b_scale, void* stream) {
    W2E_REQUIRE(a && out && w && dims && w_scale, "ssmapper_linear: null argument");
    W2E_REQUIRE(mode == 0 || (mode == 1 && y_act), "ssmapper_linear: mode 0 (forward) or 1 (input gradient, needs the layer output)");
    SsGroups g{};
    W2E_REQUIRE(ss_fill(g, groups, batch, dims), "ssmapper_linear: 1..32 codes, batch 1..16");
    for (int c = 0; c < groups; ++c) {
        W2E_REQUIRE(w[c], "ssmapper_linear: null weight");
        g.w[c] = w[c], g.bias[c] = (mode == 0 && bias) ? bias[c] : nullptr, g.w_scale[c] = w_scale[c];
    }
    const int dmax = ss_maxdim(g);
    if (mode == 0) {
        ss_linear_fwd_kernel<<<dim3((unsigned)ceil_div(dmax, 4), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(g, a, out, b_scale);
    } else {
        const size_t lds = sizeof(float) * (size_t)batch * dmax;
        W2E_REQUIRE(lds <= 64 * 1024, "ssmapper_linear: batch x width too large for the staged gradient");
        ss_linear_bwd_kernel<<<dim3((unsigned)ceil_div(dmax, 256), (unsigned)groups), 256, lds, (hipStream_t)stream>>>(g, a, y_act, out);
    }
    W2E_LAUNCH_CHECK("ssmapper_linear");
    return 0;
}

extern "C" int w2e_ssmapper_wgrad(const float* gy, const float* y, const float* h_in, float* const* gw, float* const* gb, int batch, int groups,
                                  const int* dims, const float* w_scale, float b_scale, void* stream) {
    W2E_REQUIRE(gy && y && h_in && gw && dims && w_scale, "ssmapper_wgrad: null argument");
    SsGroups g{};
    W2E_REQUIRE(ss_fill(g, groups, batch, dims), "ssmapper_wgrad: 1..32 codes, batch 1..16");
    for (int c = 0; c < groups; ++c) {
        W2E_REQUIRE(gw[c], "ssmapper_wgrad: null weight gradient");
        g.gw[c] = gw[c], g.gb[c] = gb ? gb[c] : nullptr, g.w_scale[c] = w_scale[c];
    }
    const int dmax = ss_maxdim(g);
    ss_wgrad_kernel<<<dim3((unsigned)ceil_div((int64_t)dmax * dmax, 256), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(g, gy, y, h_in, b_scale);
    W2E_LAUNCH_CHECK("ssmapper_wgrad");
    return 0;
}
