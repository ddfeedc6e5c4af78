// CLIP ViT-B/32 tower at M = 50*batch rows (include/w2e_vit.h): the kernels AROUND the four GEMMs of a block (the GEMMs themselves:
// vit3.hip, w2e_gemm_pk -- packed operands, register-fed, split-K partials WRITTEN as [M,N] slabs, never added atomically).  The consumer
// kernels here sum the slabs in ascending order (bit-reproducible, no memset) and replace the separate LayerNorm launch; each of them can
// write the tensor the NEXT GEMM consumes directly in that GEMM's K-quad-major operand packing (`packed_rows`).
//   reduce_ln_fwd    x = sum_s partial_s + bias + residual;  y = LayerNorm(x)      (one wave per row)
//   ln_bwd (partial) gx = LN'(sum_s gy_s) + add
//   reduce_gelu      the QuickGELU pair / its derivative on a sum of slabs
//   attn v2          softmax(QK^T/8)V per (batch, head) on v_mfma_f32_32x32x2_f32, QKV read as a sum of split-K slabs + bias
#include "common.h"
#include "../../include/w2e_vit.h"

namespace w2e {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ float quick_gelu2(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad2(float x) {
    const float s = 1.f / (1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}

// ------------------------------------------------------------------------------------------ reduce + LayerNorm
__device__ __forceinline__ float wave_sum2(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max2(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// x = sum_s part[s] (+ bias) (+ residual), written to x_out;  y = LayerNorm(x) * gamma + beta (y may be null: sum only).
// One wave per row, dim = 256 * T4.  The slabs are added in ascending s: deterministic.
template <int T4>
__global__ __launch_bounds__(64) void reduce_ln_fwd_kernel(const float* __restrict__ part, int nsplit, int64_t slab,
                                                           const float* __restrict__ bias, const float* __restrict__ residual,
                                                           float* __restrict__ x_out, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, int dim, float eps,
                                                           int y_mpad) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    float4 v[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < nsplit; base += 4) {  // four slabs' loads in flight per pass (each is an L2 miss), added in order
        float4 w[4][T4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < T4; ++t)
                w[c][t] = base + c < nsplit ? reinterpret_cast<const float4*>(part + (base + c) * slab + row * dim)[lane + 64 * t]
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < T4; ++t) v[t] = add4(v[t], w[c][t]);
    }
    if (bias)
#pragma unroll
        for (int t = 0; t < T4; ++t) v[t] = add4(v[t], reinterpret_cast<const float4*>(bias)[lane + 64 * t]);
    if (residual)
#pragma unroll
        for (int t = 0; t < T4; ++t) v[t] = add4(v[t], reinterpret_cast<const float4*>(residual + row * dim)[lane + 64 * t]);
    if (x_out)
#pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<float4*>(x_out + row * dim)[lane + 64 * t] = v[t];
    if (!y) return;
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) sm += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    const float mean = wave_sum2(sm) / dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        v[t].x -= mean, v[t].y -= mean, v[t].z -= mean, v[t].w -= mean;
        q += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
    }
    const float rstd = rsqrtf(wave_sum2(q) / dim + eps);
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[lane + 64 * t], bt = reinterpret_cast<const float4*>(beta)[lane + 64 * t];
        // y_mpad > 0: y is the next GEMM's A operand in the K-quad-major packing of vit3.hip, P[q][row] = y[row][4q .. 4q+3]
        float4* const dst = y_mpad > 0 ? reinterpret_cast<float4*>(y) + (int64_t)(lane + 64 * t) * y_mpad + row
                                       : reinterpret_cast<float4*>(y + row * dim) + lane + 64 * t;
        *dst = make_float4(v[t].x * rstd * g.x + bt.x, v[t].y * rstd * g.y + bt.y, v[t].z * rstd * g.z + bt.z, v[t].w * rstd * g.w + bt.w);
    }
    if (lane == 0) mean_out[row] = mean, rstd_out[row] = rstd;
}

// gx = LN'(sum_s gy_part[s]) + add   (input gradient of LayerNorm; gamma frozen)
template <int T4>
__global__ __launch_bounds__(64) void ln_bwd_part_kernel(const float* __restrict__ gpart, int nsplit, int64_t slab,
                                                         const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ add, float* __restrict__ gx, int dim,
                                                         float* __restrict__ gx_packed, int mpad) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    float4 gg[T4], xh[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) gg[t] = make_float4(0.f, 0.f, 0.f, 0.f), xh[t] = reinterpret_cast<const float4*>(x + row * dim)[lane + 64 * t];
    for (int base = 0; base < nsplit; base += 4) {
        float4 w[4][T4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < T4; ++t)
                w[c][t] = base + c < nsplit ? reinterpret_cast<const float4*>(gpart + (base + c) * slab + row * dim)[lane + 64 * t]
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < T4; ++t) gg[t] = add4(gg[t], w[c][t]);
    }
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        const float4 gm = reinterpret_cast<const float4*>(gamma)[lane + 64 * t];
        gg[t].x *= gm.x, gg[t].y *= gm.y, gg[t].z *= gm.z, gg[t].w *= gm.w;
        xh[t].x = (xh[t].x - mu) * rs, xh[t].y = (xh[t].y - mu) * rs, xh[t].z = (xh[t].z - mu) * rs, xh[t].w = (xh[t].w - mu) * rs;
        s1 += (gg[t].x + gg[t].y) + (gg[t].z + gg[t].w);
        s2 += (gg[t].x * xh[t].x + gg[t].y * xh[t].y) + (gg[t].z * xh[t].z + gg[t].w * xh[t].w);
    }
    s1 = wave_sum2(s1) / dim;
    s2 = wave_sum2(s2) / dim;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        const float4 ad = add ? reinterpret_cast<const float4*>(add + row * dim)[lane + 64 * t] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 o = make_float4(rs * (gg[t].x - s1 - xh[t].x * s2) + ad.x, rs * (gg[t].y - s1 - xh[t].y * s2) + ad.y,
                                     rs * (gg[t].z - s1 - xh[t].z * s2) + ad.z, rs * (gg[t].w - s1 - xh[t].w * s2) + ad.w);
        reinterpret_cast<float4*>(gx + row * dim)[lane + 64 * t] = o;  // (row-major: the residual branch of the next LayerNorm backward)
        if (gx_packed) reinterpret_cast<float4*>(gx_packed)[(int64_t)(lane + 64 * t) * mpad + row] = o;  // (packed: the next GEMM's A operand)
    }
}

// sum of split-K slabs with the QuickGELU pair / derivative (the N = 3072 GEMMs of the MLP run split too: 96 column tiles
// alone would leave 160 CUs idle).  mode 0: h = sum + bias, g = QuickGELU(h);  mode 1: out = sum * QuickGELU'(aux).
__global__ void reduce_gelu_kernel(const float* __restrict__ part, int nsplit, int64_t slab, const float* __restrict__ bias,
                                   const float* __restrict__ aux, float* __restrict__ h, float* __restrict__ g, int n, int64_t total4,
                                   int mode, int mpad) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += step) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < nsplit; base += 4) {
            float4 w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = base + c < nsplit ? reinterpret_cast<const float4*>(part + (base + c) * slab)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) v = add4(v, w[c]);
        }
        if (mode == 0) {
            const float4 bv = reinterpret_cast<const float4*>(bias)[q % (n >> 2)];
            v = add4(v, bv);
            reinterpret_cast<float4*>(h)[q] = v;
            // mpad > 0: the GEMM operand (g in mode 0, the gradient in mode 1) is written K-quad-major, P[quad][row] (vit3.hip)
            const int64_t pq = mpad > 0 ? (q % (n >> 2)) * mpad + q / (n >> 2) : q;
            reinterpret_cast<float4*>(g)[pq] = make_float4(quick_gelu2(v.x), quick_gelu2(v.y), quick_gelu2(v.z), quick_gelu2(v.w));
        } else {
            const float4 a = reinterpret_cast<const float4*>(aux)[q];
            const int64_t pq = mpad > 0 ? (q % (n >> 2)) * mpad + q / (n >> 2) : q;
            reinterpret_cast<float4*>(h)[pq] = make_float4(v.x * quick_gelu_grad2(a.x), v.y * quick_gelu_grad2(a.y), v.z * quick_gelu_grad2(a.z),
                                                           v.w * quick_gelu_grad2(a.w));
        }
    }
}

// ------------------------------------------------------------------------------------------ attention on MFMA
// One workgroup (4 waves) per (batch, head); L <= 64 tokens padded to 64, head dim 64.  Q, K, V (and dO, P, dS) live in LDS
// as [64][AS] rows, AS = 68 floats (16-B aligned rows, consecutive rows 4 banks apart: the b128 operand fetches of 16
// consecutive rows hit 16 different bank groups).  Every contraction is 64x64x64 on v_mfma_f32_32x32x2_f32: wave w owns the
// 32x32 output block (w>>1, w&1).  k-slot convention (same for both operands): lane-half h, group g, component c <-> k =
// 8g + 4h + c.
constexpr int AL = 64, AS = 68;

// acc += A_rows . B_rows^T : out[i][j] = sum_k A[i][k] B[j][k]   (both operands row-major, k along the row: b128 fetches)
__device__ __forceinline__ void mm_rows_rows(f32x16& acc, const float* A, const float* B, int i0, int j0, int j, int half) {
    const float4* ar = reinterpret_cast<const float4*>(A + (i0 + j) * AS) + half;
    const float4* br = reinterpret_cast<const float4*>(B + (j0 + j) * AS) + half;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 a4 = ar[2 * g], b4 = br[2 * g];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
}
// acc += A . B : out[i][n] = sum_k A[i][k] B[k][n]   (A row-major b128; B read down its rows, lanes along n)
__device__ __forceinline__ void mm_rows_cols(f32x16& acc, const float* A, const float* B, int i0, int n0, int j, int half) {
    const float4* ar = reinterpret_cast<const float4*>(A + (i0 + j) * AS) + half;
    const float* bc = B + n0 + j;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 a4 = ar[2 * g];
        const int k = 8 * g + 4 * half;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bc[(k + 0) * AS], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bc[(k + 1) * AS], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bc[(k + 2) * AS], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bc[(k + 3) * AS], acc, 0, 0, 0);
    }
}
// acc += A^T . B : out[m][n] = sum_k A[k][m] B[k][n]   (both read down their rows)
__device__ __forceinline__ void mm_cols_cols(f32x16& acc, const float* A, const float* B, int m0, int n0, int j, int half) {
    const float* ac = A + m0 + j;
    const float* bc = B + n0 + j;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int k = 8 * g + 4 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[(k + c) * AS], bc[(k + c) * AS], acc, 0, 0, 0);
    }
}
__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// scatter a wave's 32x32 accumulator block into an LDS [64][AS] matrix
__device__ __forceinline__ void acc_to_lds(const f32x16& a, float* M, int i0, int j0, int j, int half, float scale) {
#pragma unroll
    for (int r = 0; r < 16; ++r) M[(i0 + (r & 3) + 8 * (r >> 2) + 4 * half) * AS + j0 + j] = a[r] * scale;
}

// NH heads' worth of [L x 64] blocks (column offsets col[0..NH-1]) of a [B*L, ld] matrix given as nsplit slabs (+ bias) -> LDS
// [64][AS] images dst[0..NH-1], zero rows >= L.  Loop order: slab outermost, the thread's 4*NH float4s inside -- 4*NH
// independent loads in flight per slab (a per-element slab loop would chain nsplit*4*NH load latencies: measured 18 us of the
// 22 us the kernel took).
template <int NH, int CH>
__device__ __forceinline__ void load_heads(const float* src, int nsplit, int64_t slab, const float* bias, int64_t row0, int ld,
                                           const int (&col)[NH], int L, float* const (&dst)[NH]) {
    float4 v[NH][4];
    const int d = (threadIdx.x & 15) * 4, t0 = threadIdx.x >> 4;  // rows t0, t0+16, t0+32, t0+48
#pragma unroll
    for (int a = 0; a < NH; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[a][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // CH slabs per pass: all 4*NH*CH loads of a pass are issued before the first add (the slabs were written by the previous
    // kernel on other XCDs: every load is an L2 miss of ~1-2 us, and a slab-at-a-time loop would pay that nsplit times)
    for (int base = 0; base < nsplit; base += CH) {
        float4 w[CH][NH][4];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int a = 0; a < NH; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = t0 + 16 * q;
                    w[c][a][q] = (base + c < nsplit && t < L)
                                     ? *reinterpret_cast<const float4*>(src + (base + c) * slab + (row0 + t) * ld + col[a] + d)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int a = 0; a < NH; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[a][q] = add4(v[a][q], w[c][a][q]);
    }
#pragma unroll
    for (int a = 0; a < NH; ++a) {
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + col[a] + d) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + 16 * q;
            *reinterpret_cast<float4*>(dst[a] + t * AS + d) = t < L ? add4(v[a][q], bv) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// Row-wise passes over a [64][AS] LDS matrix with FOUR ADJACENT LANES per row (thread = 4*row + quarter; quarter c owns
// columns c, c+4, ..., c+60: conflict-free LDS access) so that the row reductions are two DPP quad permutes instead of six
// ds_bpermute steps per reduction (one wave per row cost 5.6 us of shuffles per pass).
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_xor1(v);
    return v + quad_xor2(v);
}
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, quad_xor1(v));
    return fmaxf(v, quad_xor2(v));
}

// P (logits) -> softmax over columns [0,L); rows >= L and columns >= L become 0.
__device__ __forceinline__ void softmax_rows(float* P, int L) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
    float x[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int jj = 4 * u + c;
        x[u] = jj < L ? P[i * AS + jj] : -3.0e38f;
        mx = fmaxf(mx, x[u]);
    }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        x[u] = (4 * u + c < L) ? __expf(x[u] - mx) : 0.f;
        sum += x[u];
    }
    const float inv = (i < L) ? 1.f / quad_sum(sum) : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) P[i * AS + 4 * u + c] = x[u] * inv;
}

// dS = P * (dP - sum_j dP*P), in place on dS (holding dP)
__device__ __forceinline__ void ds_rows(const float* P, float* dS) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
    float pv[16], dv[16];
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        pv[u] = P[i * AS + 4 * u + c], dv[u] = dS[i * AS + 4 * u + c];
        dot += pv[u] * dv[u];
    }
    dot = quad_sum(dot);
#pragma unroll
    for (int u = 0; u < 16; ++u) dS[i * AS + 4 * u + c] = pv[u] * (dv[u] - dot);
}

__global__ __launch_bounds__(256) void attn2_fwd_kernel(const float* __restrict__ qkv, int nsplit, int64_t slab,
                                                        const float* __restrict__ bias, float* __restrict__ out, int L, int H, int out_mpad) {
    extern __shared__ __attribute__((aligned(16))) float asm_[];
    float* q = asm_;
    float* k = q + AL * AS;
    float* v = k + AL * AS;
    float* p = v + AL * AS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int ld = 3 * H * 64;
    {
        const int cols[3] = {h * 64, (H + h) * 64, (2 * H + h) * 64};
        float* const dsts[3] = {q, k, v};
        load_heads<3, 3>(qkv, nsplit, slab, bias, (int64_t)b * L, ld, cols, L, dsts);
    }
    __syncthreads();
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32;
    f32x16 acc;
    acc_zero(acc);
    mm_rows_rows(acc, q, k, i0, j0, j, half);
    acc_to_lds(acc, p, i0, j0, j, half, 0.125f);
    __syncthreads();
    softmax_rows(p, L);
    __syncthreads();
    acc_zero(acc);
    mm_rows_cols(acc, p, v, i0, j0, j, half);  // O[i][d] = sum_j P[i][j] V[j][d]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (i >= L) continue;
        const int64_t m = (int64_t)b * L + i;
        const int n = h * 64 + j0 + j;
        if (out_mpad > 0) out[((int64_t)(n >> 2) * out_mpad + m) * 4 + (n & 3)] = acc[r];  // K-quad-major (vit3.hip): the out-projection's A operand
        else out[m * (H * 64) + n] = acc[r];
    }
}

__global__ __launch_bounds__(256) void attn2_bwd_kernel(const float* __restrict__ qkv, int nsplit, int64_t slab,
                                                        const float* __restrict__ bias, const float* __restrict__ gout, int gsplit,
                                                        int64_t gslab, float* __restrict__ gqkv, int L, int H, int g_mpad) {
    extern __shared__ __attribute__((aligned(16))) float asm_[];
    float* q = asm_;
    float* k = q + AL * AS;
    float* v = k + AL * AS;
    float* go = v + AL * AS;
    float* p = go + AL * AS;
    float* ds = p + AL * AS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int ld = 3 * H * 64;
    {
        const int cols[3] = {h * 64, (H + h) * 64, (2 * H + h) * 64};
        float* const dsts[3] = {q, k, v};
        load_heads<3, 3>(qkv, nsplit, slab, bias, (int64_t)b * L, ld, cols, L, dsts);
        const int gcol[1] = {h * 64};
        float* const gdst[1] = {go};
        load_heads<1, 6>(gout, gsplit, gslab, nullptr, (int64_t)b * L, H * 64, gcol, L, gdst);
    }
    __syncthreads();
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32;
    f32x16 acc, acc2;
    acc_zero(acc), acc_zero(acc2);
    mm_rows_rows(acc, q, k, i0, j0, j, half);    // S
    mm_rows_rows(acc2, go, v, i0, j0, j, half);  // dP[i][j] = dO_i . V_j
    acc_to_lds(acc, p, i0, j0, j, half, 0.125f);
    acc_to_lds(acc2, ds, i0, j0, j, half, 1.f);
    __syncthreads();
    softmax_rows(p, L);
    __syncthreads();  // (a row's four lanes wrote each other's columns)
    ds_rows(p, ds);
    __syncthreads();
    f32x16 gq, gk, gv;
    acc_zero(gq), acc_zero(gk), acc_zero(gv);
    mm_rows_cols(gq, ds, k, i0, j0, j, half);   // dQ[i][d] = sum_j dS[i][j] K[j][d]
    mm_cols_cols(gk, ds, q, i0, j0, j, half);   // dK[j][d] = sum_i dS[i][j] Q[i][d]
    mm_cols_cols(gv, p, go, i0, j0, j, half);   // dV[j][d] = sum_i P[i][j] dO[i][d]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (t >= L) continue;
        if (g_mpad > 0) {  // K-quad-major (vit3.hip): the A operand of the in-projection's input-gradient GEMM
            const int64_t m = (int64_t)b * L + t;
            const int n = h * 64 + j0 + j;
            float* dst = gqkv + ((int64_t)(n >> 2) * g_mpad + m) * 4 + (n & 3);
            const int64_t third = (int64_t)(H * 64 / 4) * g_mpad * 4;  // H*64 columns further = H*16 quads further
            dst[0] = gq[r] * 0.125f;
            dst[third] = gk[r] * 0.125f;
            dst[2 * third] = gv[r];
            continue;
        }
        float* dst = gqkv + ((int64_t)b * L + t) * ld + h * 64 + j0 + j;
        dst[0] = gq[r] * 0.125f;
        dst[(int64_t)H * 64] = gk[r] * 0.125f;
        dst[(int64_t)2 * H * 64] = gv[r];
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_reduce_gelu(const float* part, int nsplit, int64_t slab, const float* bias, const float* aux, float* h, float* g,
                              int64_t rows, int n, int mode, int packed_rows, void* stream) {
    W2E_REQUIRE(packed_rows == 0 || packed_rows >= rows, "reduce_gelu: packed_rows %d for %lld rows", packed_rows, (long long)rows);
    W2E_REQUIRE(part && h && nsplit >= 1, "reduce_gelu: null tensor / bad split count");
    W2E_REQUIRE((mode == 0 && bias && g) || (mode == 1 && aux), "reduce_gelu: mode 0 needs bias and g, mode 1 needs aux");
    W2E_REQUIRE(rows >= 0 && n > 0 && (n & 3) == 0 && (slab & 3) == 0, "reduce_gelu: n and the slab stride must be multiples of 4");
    const int64_t total4 = rows * n / 4;
    if (total4 == 0) return 0;
    reduce_gelu_kernel<<<stream_grid(total4, 256), 256, 0, (hipStream_t)stream>>>(part, nsplit, slab, bias, aux, h, g, n, total4, mode, packed_rows);
    W2E_LAUNCH_CHECK("reduce_gelu");
    return 0;
}

extern "C" int w2e_reduce_ln_fwd(const float* part, int nsplit, int64_t slab, const float* bias, const float* residual, float* x_out,
                                 const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int dim,
                                 float eps, int y_packed_rows, void* stream) {
    W2E_REQUIRE(y_packed_rows == 0 || y_packed_rows >= rows, "reduce_ln_fwd: y_packed_rows %d for %lld rows", y_packed_rows, (long long)rows);
    W2E_REQUIRE(part && nsplit >= 1, "reduce_ln_fwd: null tensor / bad split count");
    W2E_REQUIRE(!y || (gamma && beta && mean && rstd), "reduce_ln_fwd: LayerNorm output needs gamma, beta, mean, rstd");
    W2E_REQUIRE(dim == 512 || dim == 768 || dim == 1024, "reduce_ln_fwd: dim %d unsupported (512, 768, 1024)", dim);
    W2E_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31), "reduce_ln_fwd: bad rows");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
#define W2E_RLN(T) reduce_ln_fwd_kernel<T><<<(unsigned)rows, 64, 0, s>>>(part, nsplit, slab, bias, residual, x_out, gamma, beta, y, mean, rstd, dim, eps, y_packed_rows)
    if (dim == 768) W2E_RLN(3);
    else if (dim == 512) W2E_RLN(2);
    else W2E_RLN(4);
#undef W2E_RLN
    W2E_LAUNCH_CHECK("reduce_ln_fwd");
    return 0;
}

extern "C" int w2e_layernorm_bwd_part(const float* gpart, int nsplit, int64_t slab, const float* x, const float* gamma,
                                      const float* mean, const float* rstd, const float* add, float* gx, int64_t rows, int dim,
                                      float* gx_packed, int packed_rows, void* stream) {
    W2E_REQUIRE(!gx_packed || packed_rows >= rows, "layernorm_bwd_part: packed_rows %d for %lld rows", packed_rows, (long long)rows);
    W2E_REQUIRE(gpart && x && gamma && mean && rstd && gx && nsplit >= 1, "layernorm_bwd_part: null tensor / bad split count");
    W2E_REQUIRE(dim == 512 || dim == 768 || dim == 1024, "layernorm_bwd_part: dim %d unsupported (512, 768, 1024)", dim);
    W2E_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31), "layernorm_bwd_part: bad rows");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
#define W2E_LNB(T) ln_bwd_part_kernel<T><<<(unsigned)rows, 64, 0, s>>>(gpart, nsplit, slab, x, gamma, mean, rstd, add, gx, dim, gx_packed, packed_rows)
    if (dim == 768) W2E_LNB(3);
    else if (dim == 512) W2E_LNB(2);
    else W2E_LNB(4);
#undef W2E_LNB
    W2E_LAUNCH_CHECK("layernorm_bwd_part");
    return 0;
}

extern "C" int w2e_attn2_fwd(const float* qkv, int nsplit, int64_t slab, const float* bias, float* out, int batch, int seq, int heads,
                             int out_packed_rows, void* stream) {
    W2E_REQUIRE(out_packed_rows == 0 || out_packed_rows >= batch * seq, "attn2_fwd: out_packed_rows %d for %d rows", out_packed_rows, batch * seq);
    W2E_REQUIRE(qkv && out && nsplit >= 1, "attn2_fwd: null tensor / bad split count");
    W2E_REQUIRE(seq > 0 && seq <= AL && heads > 0 && batch >= 0, "attn2_fwd: seq %d (max %d), heads %d", seq, AL, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 4 * AL * AS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn2_fwd_kernel, &done), "attn2_fwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn2_fwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, nsplit, slab, bias, out, seq, heads, out_packed_rows);
    W2E_LAUNCH_CHECK("attn2_fwd");
    return 0;
}

extern "C" int w2e_attn2_bwd(const float* qkv, int nsplit, int64_t slab, const float* bias, const float* gout, int gsplit, int64_t gslab,
                             float* gqkv, int batch, int seq, int heads, int g_packed_rows, void* stream) {
    W2E_REQUIRE(g_packed_rows == 0 || g_packed_rows >= batch * seq, "attn2_bwd: g_packed_rows %d for %d rows", g_packed_rows, batch * seq);
    W2E_REQUIRE(qkv && gout && gqkv && nsplit >= 1 && gsplit >= 1, "attn2_bwd: null tensor / bad split count");
    W2E_REQUIRE(seq > 0 && seq <= AL && heads > 0 && batch >= 0, "attn2_bwd: seq %d (max %d), heads %d", seq, AL, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 6 * AL * AS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn2_bwd_kernel, &done), "attn2_bwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn2_bwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, nsplit, slab, bias, gout, gsplit, gslab, gqkv, seq, heads, g_packed_rows);
    W2E_LAUNCH_CHECK("attn2_bwd");
    return 0;
}
