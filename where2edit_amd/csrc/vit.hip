// CLIP ViT-B/32 image-encoder kernels for gfx950 (include/w2e_vit.h): fp32-MFMA GEMM with fused
// bias / residual / QuickGELU prologue / QuickGELU' epilogue, LayerNorm fwd+bwd, and the 50-token
// attention core fwd+bwd (one workgroup per (batch, head), everything in LDS).
#include <stdlib.h>

#include "common.h"
#include "../../include/w2e_vit.h"

namespace w2e {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = 1.f / (1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}

// ------------------------------------------------------------------------------------------ GEMM
// 64x64 output tile per 256-thread workgroup, 4 waves as 2(m) x 2(n), each one 32x32 accumulator of
// v_mfma_f32_32x32x2_f32.  C rows come from the A operand, C columns (contiguous in memory) sit on the
// lanes.  K is consumed in 32-deep tiles, software-pipelined through registers like the conv kernel.
constexpr int GBM = 64, GBN = 64, GBK = 32, GPA = GBK + 1;  // +1 pad: conflict-free column reads

struct GemmParams {
    const float* a;
    const float* b;
    float* c;
    int m, n, k, lda, ldb, ldc;
    const float* bias;
    const float* residual;
    const float* aux;
    int k_per;  // K range of one blockIdx.z slice (split-K); == k when gridDim.z == 1
};

template <bool TRANS_B, bool A_GELU>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    __shared__ float As[GBM * GPA];
    __shared__ float Bs[TRANS_B ? GBN * GPA : GBK * GBN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 ar[2], br[2];
    auto prefetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = (tid >> 3) + 32 * it, kq = (tid & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + row < p.m && k0 + kq < p.k) v = *reinterpret_cast<const float4*>(p.a + (int64_t)(m0 + row) * p.lda + k0 + kq);
            ar[it] = v;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TRANS_B) {
                if (n0 + row < p.n && k0 + kq < p.k) w = *reinterpret_cast<const float4*>(p.b + (int64_t)(n0 + row) * p.ldb + k0 + kq);
            } else {
                const int kr = (tid >> 4) + 16 * it, nq = (tid & 15) * 4;
                if (k0 + kr < p.k && n0 + nq < p.n) w = *reinterpret_cast<const float4*>(p.b + (int64_t)(k0 + kr) * p.ldb + n0 + nq);
            }
            br[it] = w;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = (tid >> 3) + 32 * it, kq = (tid & 7) * 4;
            float4 v = ar[it];
            if (A_GELU) v.x = quick_gelu(v.x), v.y = quick_gelu(v.y), v.z = quick_gelu(v.z), v.w = quick_gelu(v.w);
            float* d = As + row * GPA + kq;
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
            if (TRANS_B) {
                float* e = Bs + row * GPA + kq;
                e[0] = br[it].x, e[1] = br[it].y, e[2] = br[it].z, e[3] = br[it].w;
            } else {
                const int kr = (tid >> 4) + 16 * it, nq = (tid & 15) * 4;
                *reinterpret_cast<float4*>(Bs + kr * GBN + nq) = br[it];
            }
        }
    };

    const int k_lo = blockIdx.z * p.k_per;
    const int k_hi = (k_lo + p.k_per < p.k) ? k_lo + p.k_per : p.k;
    prefetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += GBK) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + GBK < k_hi) prefetch(k0 + GBK);
        const float* ap = As + (wm * 32 + j) * GPA + half;
        const float* bp = TRANS_B ? Bs + (wn * 32 + j) * GPA + half : Bs + half * GBN + wn * 32 + j;
#pragma unroll
        for (int s = 0; s < GBK / 2; ++s) {
            const float av = ap[2 * s];
            const float bv = TRANS_B ? bp[2 * s] : bp[2 * s * GBN];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + j;
    if (n >= p.n) return;
    const bool first = blockIdx.z == 0;  // split-K: slice 0 carries bias + residual, every slice adds atomically
    const float bs = (p.bias && first) ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.m) continue;
        const int64_t ci = (int64_t)m * p.ldc + n;
        float v = acc[r] + bs;
        if (p.residual && first) v += p.residual[ci];
        if (gridDim.z > 1) {
            atomicAdd(&p.c[ci], v);
        } else {
            if (p.aux) v *= quick_gelu_grad(p.aux[ci]);
            p.c[ci] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ float wave_sum_ln(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int LN_MAX_PER_LANE = 32;  // dim <= 2048

// One wave per row; the row is held in registers (two-pass mean / variance, like F.layer_norm).
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * dim;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < dim ? xr[c] : 0.f;
        s += v[t];
    }
    const float mean = wave_sum_ln(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        const float d = c < dim ? v[t] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum_ln(q) / dim + eps);
#pragma unroll
    for (int t = 0; t < LN_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        if (c < dim) y[row * dim + c] = (v[t] - mean) * rstd * gamma[c] + beta[c];
    }
    if (lane == 0) mean_out[row] = mean, rstd_out[row] = rstd;
}

// gx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ gx,
                                                            int64_t rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    float gg[LN_MAX_PER_LANE], xh[LN_MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        const bool ok = c < dim;
        gg[t] = ok ? gy[row * dim + c] * gamma[c] : 0.f;
        xh[t] = ok ? (x[row * dim + c] - mu) * rs : 0.f;
        s1 += gg[t];
        s2 += gg[t] * xh[t];
    }
    s1 = wave_sum_ln(s1) / dim;
    s2 = wave_sum_ln(s2) / dim;
#pragma unroll
    for (int t = 0; t < LN_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        if (c < dim) gx[row * dim + c] = rs * (gg[t] - s1 - xh[t] * s2);
    }
}

// ------------------------------------------------------------------------------------------ attention
constexpr int HD = 64, HP = 65, LMAX = 64;

// qkv [B, L, 3, H, 64] -> LDS q[L][64], k[L][65], v[L][65]
__device__ __forceinline__ void load_qkv(const float* qkv, int b, int h, int L, int H, float* q, float* k, float* v) {
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int t = e >> 6, d = e & 63;
        const float* src = qkv + (((int64_t)b * L + t) * 3 * H + h) * HD + d;
        q[t * HD + d] = src[0];
        k[t * HP + d] = src[(int64_t)H * HD];
        v[t * HP + d] = src[(int64_t)2 * H * HD];
    }
}

// p[i][j] = softmax_j(q_i . k_j / 8)
__device__ __forceinline__ void softmax_probs(const float* q, const float* k, float* p, int L) {
    const int LP = L + 1;
    for (int e = threadIdx.x; e < L * L; e += 256) {
        const int i = e / L, jj = e - i * L;
        float s = 0.f;
#pragma unroll 16
        for (int d = 0; d < HD; ++d) s += q[i * HD + d] * k[jj * HP + d];
        p[i * LP + jj] = s * 0.125f;
    }
    __syncthreads();
    if (threadIdx.x < L) {
        float* row = p + threadIdx.x * LP;
        float mx = row[0];
        for (int jj = 1; jj < L; ++jj) mx = fmaxf(mx, row[jj]);
        float sum = 0.f;
        for (int jj = 0; jj < L; ++jj) {
            const float e = __expf(row[jj] - mx);
            row[jj] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        for (int jj = 0; jj < L; ++jj) row[jj] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem;
    float* k = q + LMAX * HD;
    float* v = k + LMAX * HP;
    float* p = v + LMAX * HP;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    load_qkv(qkv, b, h, L, H, q, k, v);
    __syncthreads();
    softmax_probs(q, k, p, L);
    const int LP = L + 1;
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int i = e >> 6, d = e & 63;
        float s = 0.f;
        for (int jj = 0; jj < L; ++jj) s += p[i * LP + jj] * v[jj * HP + d];
        out[(((int64_t)b * L + i) * H + h) * HD + d] = s;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ gout,
                                                       float* __restrict__ gqkv, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem;
    float* k = q + LMAX * HD;
    float* v = k + LMAX * HP;
    float* go = v + LMAX * HP;          // [L][64]
    float* p = go + LMAX * HD;          // [L][L+1]
    float* ds = p + LMAX * (LMAX + 1);  // [L][L+1]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int LP = L + 1;
    load_qkv(qkv, b, h, L, H, q, k, v);
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int t = e >> 6, d = e & 63;
        go[t * HD + d] = gout[(((int64_t)b * L + t) * H + h) * HD + d];
    }
    __syncthreads();
    softmax_probs(q, k, p, L);
    // dP[i][j] = go_i . v_j
    for (int e = threadIdx.x; e < L * L; e += 256) {
        const int i = e / L, jj = e - i * L;
        float s = 0.f;
#pragma unroll 16
        for (int d = 0; d < HD; ++d) s += go[i * HD + d] * v[jj * HP + d];
        ds[i * LP + jj] = s;
    }
    __syncthreads();
    if (threadIdx.x < L) {  // dS = P * (dP - sum_j dP*P)
        const int i = threadIdx.x;
        float dot = 0.f;
        for (int jj = 0; jj < L; ++jj) dot += ds[i * LP + jj] * p[i * LP + jj];
        for (int jj = 0; jj < L; ++jj) ds[i * LP + jj] = p[i * LP + jj] * (ds[i * LP + jj] - dot);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int t = e >> 6, d = e & 63;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int u = 0; u < L; ++u) {
            gq += ds[t * LP + u] * k[u * HP + d];   // dQ_t = sum_j dS[t][j] K_j
            gk += ds[u * LP + t] * q[u * HD + d];   // dK_t = sum_i dS[i][t] Q_i
            gv += p[u * LP + t] * go[u * HD + d];   // dV_t = sum_i P[i][t] dO_i
        }
        float* dst = gqkv + (((int64_t)b * L + t) * 3 * H + h) * HD + d;
        dst[0] = gq * 0.125f;
        dst[(int64_t)H * HD] = gk * 0.125f;
        dst[(int64_t)2 * H * HD] = gv;
    }
}

}  // namespace w2e

using namespace w2e;

// Dynamic LDS above 64 KB has to be enabled per kernel (gfx950 has 160 KB per CU).
static int set_big_lds(const void* fn, size_t bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
}

extern "C" int w2e_gemm(const float* a, const float* b, float* c, int m, int n, int k, int lda, int ldb, int ldc,
                        int trans_b, int a_gelu, const float* bias, const float* residual, const float* gelu_grad_aux,
                        void* stream) {
    W2E_REQUIRE(a && b && c, "gemm: null tensor");
    W2E_REQUIRE(m >= 0 && n > 0 && k > 0, "gemm: bad dims %d %d %d", m, n, k);
    W2E_REQUIRE((k & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0, "gemm: K, lda, ldb must be multiples of 4");
    W2E_REQUIRE(trans_b || (n & 3) == 0, "gemm: N must be a multiple of 4 for a [K,N] B operand");
    W2E_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "gemm: operands must be 16-byte aligned");
    W2E_REQUIRE(!(a_gelu && !trans_b), "gemm: the QuickGELU prologue is implemented for the [N,K] form only");
    if (m == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // Small-M GEMMs (M = 50*batch) leave most CUs idle with one workgroup per 64x64 tile and make every wave a
    // K/2-long dependent MFMA chain: split K over blockIdx.z (fp32 atomics onto a zeroed C) until the grid fills the chip.
    const int64_t tiles = ceil_div(n, GBN) * ceil_div(m, GBM);
    int splits = 1;
    if (!gelu_grad_aux && ldc == n) {
        static const int tune_t = getenv("W2E_TUNE_GEMM_T") ? atoi(getenv("W2E_TUNE_GEMM_T")) : 256;
        static const int tune_k = getenv("W2E_TUNE_GEMM_K") ? atoi(getenv("W2E_TUNE_GEMM_K")) : 256;
        while (tiles * splits < tune_t && splits < 16 && k / (splits * 2) >= tune_k) splits *= 2;
    }
    const int k_per = (int)(ceil_div(ceil_div(k, splits), GBK) * GBK);
    splits = (int)ceil_div(k, k_per);
    if (splits > 1 && hipMemsetAsync(c, 0, sizeof(float) * (size_t)m * n, s) != hipSuccess) {
        set_error("gemm: memset failed");
        return 2;
    }
    GemmParams p{a, b, c, m, n, k, lda, ldb, ldc, bias, residual, gelu_grad_aux, k_per};
    dim3 grid((unsigned)ceil_div(n, GBN), (unsigned)ceil_div(m, GBM), (unsigned)splits);
    if (trans_b) {
        if (a_gelu) gemm_kernel<true, true><<<grid, 256, 0, s>>>(p);
        else gemm_kernel<true, false><<<grid, 256, 0, s>>>(p);
    } else {
        gemm_kernel<false, false><<<grid, 256, 0, s>>>(p);
    }
    W2E_LAUNCH_CHECK("gemm");
    return 0;
}

extern "C" int w2e_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int64_t rows, int dim, float eps, void* stream) {
    W2E_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null tensor");
    W2E_REQUIRE(dim > 0 && dim <= 64 * LN_MAX_PER_LANE, "layernorm_fwd: dim %d unsupported (max %d)", dim, 64 * LN_MAX_PER_LANE);
    if (rows <= 0) return 0;
    layernorm_fwd_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, y, mean, rstd, rows, dim, eps);
    W2E_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int w2e_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 float* gx, int64_t rows, int dim, void* stream) {
    W2E_REQUIRE(gy && x && gamma && mean && rstd && gx, "layernorm_bwd: null tensor");
    W2E_REQUIRE(dim > 0 && dim <= 64 * LN_MAX_PER_LANE, "layernorm_bwd: dim %d unsupported", dim);
    if (rows <= 0) return 0;
    layernorm_bwd_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(gy, x, gamma, mean, rstd, gx, rows, dim);
    W2E_LAUNCH_CHECK("layernorm_bwd");
    return 0;
}

extern "C" int w2e_attn_fwd(const float* qkv, float* out, int batch, int seq, int heads, void* stream) {
    W2E_REQUIRE(qkv && out, "attn_fwd: null tensor");
    W2E_REQUIRE(seq > 0 && seq <= LMAX && heads > 0 && batch >= 0, "attn_fwd: seq %d (max %d), heads %d", seq, LMAX, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * (LMAX * HD + 2 * LMAX * HP + LMAX * (LMAX + 1));
    static int configured_fwd = -1;
    if (configured_fwd != 0) configured_fwd = set_big_lds((const void*)attn_fwd_kernel, lds);
    W2E_REQUIRE(configured_fwd == 0, "attn_fwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn_fwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, out, seq, heads);
    W2E_LAUNCH_CHECK("attn_fwd");
    return 0;
}

extern "C" int w2e_attn_bwd(const float* qkv, const float* gout, float* gqkv, int batch, int seq, int heads, void* stream) {
    W2E_REQUIRE(qkv && gout && gqkv, "attn_bwd: null tensor");
    W2E_REQUIRE(seq > 0 && seq <= LMAX && heads > 0 && batch >= 0, "attn_bwd: seq %d (max %d), heads %d", seq, LMAX, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * (2 * LMAX * HD + 2 * LMAX * HP + 2 * LMAX * (LMAX + 1));
    static int configured = -1;
    if (configured != 0) configured = set_big_lds((const void*)attn_bwd_kernel, lds);
    W2E_REQUIRE(configured == 0, "attn_bwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn_bwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, gout, gqkv, seq, heads);
    W2E_LAUNCH_CHECK("attn_bwd");
    return 0;
}
