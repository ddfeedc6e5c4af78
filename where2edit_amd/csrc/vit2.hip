// CLIP ViT-B/32 tower, second generation (include/w2e_vit.h, "v2" entry points): kernels shaped for M = 50*batch rows.
//
//   gemm_fm   C[M,N] = A[M,K] x B[N,K]^T with the WHOLE M extent in one workgroup: 7 waves = 7 row blocks of 32 (224 rows
//             = 4 images of 50 tokens + padding to the MFMA granule instead of to 256), 32 output columns, K split over
//             blockIdx.z so that ~256 workgroups exist for every shape of the tower (N = 768 ... 3072, K = 768 ... 3072).
//             Split-K partials are WRITTEN (one [M,N] slab per split), never added atomically: the consumer kernel
//             (reduce + LayerNorm, LayerNorm backward, attention) sums them -- bit-reproducible, no memset, and it replaces
//             the separate LayerNorm launch.  Operands stream global -> LDS by buffer_load...lds (16 B per lane) in a
//             2-stage ring (two workgroups per CU) of 32-deep K-steps, raw row-major rows with an XOR swizzle of the 16-B quads chosen at the SOURCE
//             address (lane l fetches quad (l%8)^(l/8) of its row), so the MFMA operand fetch -- one ds_read_b128 per
//             operand per 4 MFMAs -- is bank-conflict free.  One barrier per K-step; waits are vmcnt-counted per wave.
//   reduce_ln_fwd    x = sum_s partial_s + bias + residual;  y = LayerNorm(x)      (one wave per row)
//   ln_bwd (partial) gx = LN'(sum_s gy_s) + add
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

// ------------------------------------------------------------------------------------------ gemm_fm
#ifndef W2E_FM_STAGES
#define W2E_FM_STAGES 2  // 65 KB of LDS: two workgroups per CU hide each other's first-load and epilogue
#endif
constexpr int FM_WAVES = 7, FM_ROWS = 32 * FM_WAVES, FM_BN = 32, FM_BK = 32, FM_STAGES = W2E_FM_STAGES;
constexpr int FM_SLOTS = 5;                                   // DMA wave-instructions per wave per K-step (7 x 5 = 35 >= 32)
constexpr int FM_STAGE_BYTES = (FM_ROWS + FM_BN) * FM_BK * 4;  // 256 rows x 128 B = 32 KB
constexpr int FM_DUMMY_BYTES = 1024;                           // landing area of the 3 surplus slots

enum { FM_EPI_PLAIN = 0, FM_EPI_PARTIAL = 1, FM_EPI_GELU_DUAL = 2, FM_EPI_GELU_GRAD = 3 };

struct GemmFmParams {
    const float* a;
    const float* b;
    float* c;
    float* c2;          // GELU_DUAL: gelu(c)
    const float* bias;  // [N] or null (PLAIN / GELU_DUAL)
    const float* aux;   // GELU_GRAD: c = acc * QuickGELU'(aux)
    int m, n, k, lda, ldb, ldc;
    int k_per;          // K range of one blockIdx.z slice (multiple of FM_BK)
    int epi;
    int64_t slab;       // PARTIAL: elements between consecutive split slabs (m * ldc)
};

__global__ __launch_bounds__(64 * FM_WAVES) void gemm_fm_kernel(const GemmFmParams p) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.x * FM_BN, m0 = blockIdx.y * FM_ROWS;
    const int k_lo = blockIdx.z * p.k_per;
    const int k_hi = (k_lo + p.k_per < p.k) ? k_lo + p.k_per : p.k;
    const int steps = (k_hi - k_lo + FM_BK - 1) / FM_BK;

    // ---- DMA roles.  Slot s (0..31) = rows 8s..8s+7 of the stage image (rows 0..223 = A, 224..255 = B), lane l = (row l/8,
    // destination quad l%8) fetching SOURCE quad (l%8)^(l/8).  Rows past M / N fall past the descriptor and land as zeros.
    const int lrow = lane >> 3, squad = (lane & 7) ^ lrow;
    static_assert(FM_WAVES - 1 + FM_WAVES * (FM_SLOTS - 2) < 28 && FM_WAVES * (FM_SLOTS - 1) >= 28, "slot roles");
    unsigned voff[FM_SLOTS];
    unsigned ldst[FM_SLOTS];
#pragma unroll
    for (int i = 0; i < FM_SLOTS; ++i) {
        const int s = wave + FM_WAVES * i;
        if (s < 28) {
            voff[i] = (unsigned)((m0 + 8 * s + lrow) * p.lda) * 4u + (unsigned)squad * 16u;
        } else if (s < 32) {
            voff[i] = (unsigned)((n0 + 8 * (s - 28) + lrow) * p.ldb) * 4u + (unsigned)squad * 16u;
        } else {
            voff[i] = 0xfffffff0u;  // surplus slot: out of range (zeros) into the dummy area
        }
        ldst[i] = s < 32 ? (unsigned)s * 1024u : (unsigned)(FM_STAGES * FM_STAGE_BYTES);
    }
    // The DMA is issued through inline assembly, not __builtin_amdgcn_raw_ptr_buffer_load_lds: for the builtin the compiler
    // knows an asynchronous LDS write is in flight and, lacking alias information, puts `s_waitcnt vmcnt(0)` in front of every
    // ds_read and every barrier -- which drains the whole ring each K-step (measured: 1.9 us per step instead of 0.5).  Here
    // the counting is explicit: each wave issues exactly FM_SLOTS loads per step and waits with vmcnt(5 * steps in flight).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto rsrc = [](const float* ptr, unsigned bytes) __attribute__((always_inline)) {  // raw buffer descriptor: base, stride 0, size, flags
        const uint64_t a64 = (uint64_t)(uintptr_t)ptr;
        i32x4 d;
        d[0] = (int)(unsigned)a64, d[1] = (int)(unsigned)((a64 >> 32) & 0xffffu), d[2] = (int)bytes, d[3] = 0x00020000;
        return d;
    };
    const i32x4 qa = rsrc(p.a, (unsigned)p.m * (unsigned)p.lda * 4u), qb = rsrc(p.b, (unsigned)p.n * (unsigned)p.ldb * 4u);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)fsm;
    // Every workgroup of a launch streams the SAME rows of A; walking K in the same order they would all ask the same few L2
    // channels for the same lines at the same moment (rows of A are 3-12 KB apart: a K-step's 224 row segments fall on 4, or
    // 1, of the 16 channels).  Each workgroup therefore starts its K walk at a different step (rot) and wraps around.
    const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)steps);
    auto issue = [&](int step) __attribute__((always_inline)) {
        int ks = step + rot;
        ks = ks >= steps ? ks - steps : ks;
        const unsigned soff = (unsigned)(k_lo + ks * FM_BK) * 4u;
        const unsigned lb = lds_base + (unsigned)(step % FM_STAGES) * FM_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < FM_SLOTS; ++i) {
            const unsigned dst = (wave + FM_WAVES * i < 32) ? lb + ldst[i] : lds_base + ldst[i];
#if defined(__HIP_DEVICE_COMPILE__)
            // slots 0..3 of a wave are always rows of A (wave + 21 <= 27), slot 4 is a row group of B (waves 0..3) or surplus
            if (i == FM_SLOTS - 1)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[i]), "s"(qb), "s"(soff) : "memory");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[i]), "s"(qa), "s"(soff) : "memory");
#endif
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int pre = steps < FM_STAGES - 1 ? steps : FM_STAGES - 1;
    for (int s = 0; s < pre; ++s) issue(s);
    // lane's operand addresses inside a stage: row (32*wave + j) of A, row (224 + j) of B; quad (2g + half) ^ (row & 7)
    const int sw = j & 7;
    const float4* const a_row = reinterpret_cast<const float4*>(fsm) + (32 * wave + j) * 8;
    const float4* const b_row = reinterpret_cast<const float4*>(fsm) + (FM_ROWS + j) * 8;
    for (int t = 0; t < steps; ++t) {
        // this wave's pieces of step t have landed when at most the pieces of the younger in-flight steps are outstanding
        const int younger = (steps - 1 - t) < (FM_STAGES - 2) ? (steps - 1 - t) : (FM_STAGES - 2);
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F7A);       // vmcnt(10)
        else if (younger == 1) __builtin_amdgcn_s_waitcnt(0x0F75);  // vmcnt(5)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody finished reading the stage that step t+3 overwrites
        if (t + FM_STAGES - 1 < steps) issue(t + FM_STAGES - 1);
        const int st4 = (t % FM_STAGES) * (FM_STAGE_BYTES / 16);
#pragma unroll
        for (int g = 0; g < FM_BK / 8; ++g) {
            const int quad = (2 * g + half) ^ sw;
            const float4 a4 = a_row[st4 + quad], b4 = b_row[st4 + quad];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[3], 0, 0, 0);
        }
    }
    // ---- epilogue: lane holds column n0 + j of rows m0 + 32*wave + (r&3) + 8*(r>>2) + 4*half
    const int n = n0 + j;
    if (n >= p.n) return;
    float* c = p.c + (p.epi == FM_EPI_PARTIAL ? (int64_t)blockIdx.z * p.slab : 0);
    const float bs = (p.bias && p.epi != FM_EPI_PARTIAL && p.epi != FM_EPI_GELU_GRAD) ? p.bias[n] : 0.f;
    float ax[16];
    if (p.epi == FM_EPI_GELU_GRAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            ax[r] = m < p.m ? p.aux[(int64_t)m * p.ldc + n] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.m) continue;
        float v = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]) + bs;
        const int64_t ci = (int64_t)m * p.ldc + n;
        if (p.epi == FM_EPI_GELU_GRAD) v *= quick_gelu_grad2(ax[r]);
        c[ci] = v;
        if (p.epi == FM_EPI_GELU_DUAL) p.c2[ci] = quick_gelu2(v);
    }
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
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, int dim, float eps) {
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
        reinterpret_cast<float4*>(y + row * dim)[lane + 64 * t] =
            make_float4(v[t].x * rstd * g.x + bt.x, v[t].y * rstd * g.y + bt.y, v[t].z * rstd * g.z + bt.z, v[t].w * rstd * g.w + bt.w);
    }
    if (lane == 0) mean_out[row] = mean, rstd_out[row] = rstd;
}

// gx = LN'(sum_s gy_part[s]) + add   (input gradient of LayerNorm; gamma frozen)
template <int T4>
__global__ __launch_bounds__(64) void ln_bwd_part_kernel(const float* __restrict__ gpart, int nsplit, int64_t slab,
                                                         const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ add, float* __restrict__ gx, int dim) {
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
        reinterpret_cast<float4*>(gx + row * dim)[lane + 64 * t] =
            make_float4(rs * (gg[t].x - s1 - xh[t].x * s2) + ad.x, rs * (gg[t].y - s1 - xh[t].y * s2) + ad.y,
                        rs * (gg[t].z - s1 - xh[t].z * s2) + ad.z, rs * (gg[t].w - s1 - xh[t].w * s2) + ad.w);
    }
}

// sum of split-K slabs with the QuickGELU pair / derivative (the N = 3072 GEMMs of the MLP run split too: 96 column tiles
// alone would leave 160 CUs idle).  mode 0: h = sum + bias, g = QuickGELU(h);  mode 1: out = sum * QuickGELU'(aux).
__global__ void reduce_gelu_kernel(const float* __restrict__ part, int nsplit, int64_t slab, const float* __restrict__ bias,
                                   const float* __restrict__ aux, float* __restrict__ h, float* __restrict__ g, int n, int64_t total4,
                                   int mode) {
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
            reinterpret_cast<float4*>(g)[q] = make_float4(quick_gelu2(v.x), quick_gelu2(v.y), quick_gelu2(v.z), quick_gelu2(v.w));
        } else {
            const float4 a = reinterpret_cast<const float4*>(aux)[q];
            reinterpret_cast<float4*>(h)[q] = make_float4(v.x * quick_gelu_grad2(a.x), v.y * quick_gelu_grad2(a.y), v.z * quick_gelu_grad2(a.z),
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
                                                        const float* __restrict__ bias, float* __restrict__ out, int L, int H) {
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
        if (i < L) out[((int64_t)b * L + i) * (H * 64) + h * 64 + j0 + j] = acc[r];
    }
}

__global__ __launch_bounds__(256) void attn2_bwd_kernel(const float* __restrict__ qkv, int nsplit, int64_t slab,
                                                        const float* __restrict__ bias, const float* __restrict__ gout, int gsplit,
                                                        int64_t gslab, float* __restrict__ gqkv, int L, int H) {
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
        float* dst = gqkv + ((int64_t)b * L + t) * ld + h * 64 + j0 + j;
        dst[0] = gq[r] * 0.125f;
        dst[(int64_t)H * 64] = gk[r] * 0.125f;
        dst[(int64_t)2 * H * 64] = gv[r];
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_reduce_gelu(const float* part, int nsplit, int64_t slab, const float* bias, const float* aux, float* h, float* g,
                              int64_t rows, int n, int mode, void* stream) {
    W2E_REQUIRE(part && h && nsplit >= 1, "reduce_gelu: null tensor / bad split count");
    W2E_REQUIRE((mode == 0 && bias && g) || (mode == 1 && aux), "reduce_gelu: mode 0 needs bias and g, mode 1 needs aux");
    W2E_REQUIRE(rows >= 0 && n > 0 && (n & 3) == 0 && (slab & 3) == 0, "reduce_gelu: n and the slab stride must be multiples of 4");
    const int64_t total4 = rows * n / 4;
    if (total4 == 0) return 0;
    reduce_gelu_kernel<<<stream_grid(total4, 256), 256, 0, (hipStream_t)stream>>>(part, nsplit, slab, bias, aux, h, g, n, total4, mode);
    W2E_LAUNCH_CHECK("reduce_gelu");
    return 0;
}

extern "C" int w2e_reduce_ln_fwd(const float* part, int nsplit, int64_t slab, const float* bias, const float* residual, float* x_out,
                                 const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int dim,
                                 float eps, void* stream) {
    W2E_REQUIRE(part && nsplit >= 1, "reduce_ln_fwd: null tensor / bad split count");
    W2E_REQUIRE(!y || (gamma && beta && mean && rstd), "reduce_ln_fwd: LayerNorm output needs gamma, beta, mean, rstd");
    W2E_REQUIRE(dim == 512 || dim == 768 || dim == 1024, "reduce_ln_fwd: dim %d unsupported (512, 768, 1024)", dim);
    W2E_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31), "reduce_ln_fwd: bad rows");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
#define W2E_RLN(T) reduce_ln_fwd_kernel<T><<<(unsigned)rows, 64, 0, s>>>(part, nsplit, slab, bias, residual, x_out, gamma, beta, y, mean, rstd, dim, eps)
    if (dim == 768) W2E_RLN(3);
    else if (dim == 512) W2E_RLN(2);
    else W2E_RLN(4);
#undef W2E_RLN
    W2E_LAUNCH_CHECK("reduce_ln_fwd");
    return 0;
}

extern "C" int w2e_layernorm_bwd_part(const float* gpart, int nsplit, int64_t slab, const float* x, const float* gamma,
                                      const float* mean, const float* rstd, const float* add, float* gx, int64_t rows, int dim,
                                      void* stream) {
    W2E_REQUIRE(gpart && x && gamma && mean && rstd && gx && nsplit >= 1, "layernorm_bwd_part: null tensor / bad split count");
    W2E_REQUIRE(dim == 512 || dim == 768 || dim == 1024, "layernorm_bwd_part: dim %d unsupported (512, 768, 1024)", dim);
    W2E_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31), "layernorm_bwd_part: bad rows");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
#define W2E_LNB(T) ln_bwd_part_kernel<T><<<(unsigned)rows, 64, 0, s>>>(gpart, nsplit, slab, x, gamma, mean, rstd, add, gx, dim)
    if (dim == 768) W2E_LNB(3);
    else if (dim == 512) W2E_LNB(2);
    else W2E_LNB(4);
#undef W2E_LNB
    W2E_LAUNCH_CHECK("layernorm_bwd_part");
    return 0;
}

extern "C" int w2e_attn2_fwd(const float* qkv, int nsplit, int64_t slab, const float* bias, float* out, int batch, int seq, int heads,
                             void* stream) {
    W2E_REQUIRE(qkv && out && nsplit >= 1, "attn2_fwd: null tensor / bad split count");
    W2E_REQUIRE(seq > 0 && seq <= AL && heads > 0 && batch >= 0, "attn2_fwd: seq %d (max %d), heads %d", seq, AL, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 4 * AL * AS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn2_fwd_kernel, &done), "attn2_fwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn2_fwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, nsplit, slab, bias, out, seq, heads);
    W2E_LAUNCH_CHECK("attn2_fwd");
    return 0;
}

extern "C" int w2e_attn2_bwd(const float* qkv, int nsplit, int64_t slab, const float* bias, const float* gout, int gsplit, int64_t gslab,
                             float* gqkv, int batch, int seq, int heads, void* stream) {
    W2E_REQUIRE(qkv && gout && gqkv && nsplit >= 1 && gsplit >= 1, "attn2_bwd: null tensor / bad split count");
    W2E_REQUIRE(seq > 0 && seq <= AL && heads > 0 && batch >= 0, "attn2_bwd: seq %d (max %d), heads %d", seq, AL, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 6 * AL * AS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn2_bwd_kernel, &done), "attn2_bwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn2_bwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, nsplit, slab, bias, gout, gsplit, gslab, gqkv, seq, heads);
    W2E_LAUNCH_CHECK("attn2_bwd");
    return 0;
}

// How many K-splits gemm_fm uses for (m, n, k): enough workgroups to fill the chip, every slice a multiple of 32 deep.
extern "C" int w2e_gemm_fm_splits(int m, int n, int k, int allow_split) {
    if (!allow_split || m <= 0 || n <= 0 || k <= 0) return 1;
    const int64_t tiles = ceil_div(n, FM_BN) * ceil_div(m, FM_ROWS);
    const int steps = (int)ceil_div(k, FM_BK);
    int best = 1;
    double best_cost = 0.0;
    for (int sp = 1; sp <= 16; ++sp) {
        const int per = (int)ceil_div(steps, sp);
        if (sp > 1 && (int)ceil_div(steps, per) != sp) continue;
        if (sp > 1 && per < 2) break;
        // Measured model (microseconds, M <= 224, 2-stage ring = two workgroups per CU sharing its MFMA pipes): a K-step costs
        // ~1.1 us per co-resident round of 256 workgroups, ~5 us of launch + first-load + epilogue, 0.7 us per 256 workgroups
        // of scheduling, and every slab is one more pass for the consumer that sums them.
        const double wgs = (double)(tiles * sp);
        const double rounds = (double)ceil_div(tiles * sp, 256);
        double cost = rounds * per * 1.1 + 5.0 + 0.7 * wgs / 256.0 + 0.15 * sp;
        if (wgs > 512.0) cost *= 1.1;  // a third round starts behind the first two
        if (sp == 1 || cost < best_cost * 0.98) best = sp, best_cost = cost;
    }
    return best;
}

extern "C" int w2e_gemm_fm(const float* a, const float* b, float* c, float* c2, int m, int n, int k, int lda, int ldb, int ldc,
                           int splits, int epi, const float* bias, const float* aux, void* stream) {
    W2E_REQUIRE(a && b && c, "gemm_fm: null tensor");
    W2E_REQUIRE(m >= 0 && n > 0 && k > 0, "gemm_fm: bad dims %d %d %d", m, n, k);
    W2E_REQUIRE((k & 31) == 0 && (lda & 3) == 0 && (ldb & 3) == 0, "gemm_fm: K must be a multiple of 32, lda/ldb of 4");
    W2E_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "gemm_fm: operands must be 16-byte aligned");
    W2E_REQUIRE(epi >= 0 && epi <= 3, "gemm_fm: bad epilogue %d", epi);
    W2E_REQUIRE(splits >= 1 && (splits == 1 || epi == FM_EPI_PARTIAL), "gemm_fm: split-K writes partial slabs (epi = 1) only");
    W2E_REQUIRE(epi != FM_EPI_GELU_DUAL || c2, "gemm_fm: GELU_DUAL needs c2");
    W2E_REQUIRE(epi != FM_EPI_GELU_GRAD || aux, "gemm_fm: GELU_GRAD needs aux");
    if (m == 0) return 0;
    W2E_REQUIRE((int64_t)m * lda * 4 < ((int64_t)1 << 32) && (int64_t)n * ldb * 4 < ((int64_t)1 << 32), "gemm_fm: an operand exceeds 4 GB");
    GemmFmParams p{};
    p.a = a, p.b = b, p.c = c, p.c2 = c2, p.bias = bias, p.aux = aux;
    p.m = m, p.n = n, p.k = k, p.lda = lda, p.ldb = ldb, p.ldc = ldc, p.epi = epi;
    const int steps = (int)ceil_div(k, FM_BK);
    p.k_per = (int)ceil_div(steps, splits) * FM_BK;
    const int zs = (int)ceil_div(k, p.k_per);
    W2E_REQUIRE(zs == splits, "gemm_fm: %d splits do not divide K = %d into 32-deep steps (use w2e_gemm_fm_splits)", splits, k);
    p.slab = (int64_t)m * ldc;
    constexpr size_t lds = (size_t)FM_STAGES * FM_STAGE_BYTES + FM_DUMMY_BYTES;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)gemm_fm_kernel, &done), "gemm_fm: cannot enable %zu B of dynamic LDS", lds);
    dim3 grid((unsigned)ceil_div(n, FM_BN), (unsigned)ceil_div(m, FM_ROWS), (unsigned)splits);
    gemm_fm_kernel<<<grid, 64 * FM_WAVES, lds, (hipStream_t)stream>>>(p);
    W2E_LAUNCH_CHECK("gemm_fm");
    return 0;
}
