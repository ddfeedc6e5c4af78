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
// M = 50*batch rows is a skinny GEMM: a handful of tiles, each a long dependent chain of K-steps whose cost is one
// HBM/L2 round trip.  So the kernel is built around (a) full-line loads with many bytes in flight per step, (b) few
// steps per workgroup, (c) enough workgroups to fill the chip:
//   * tile 64(M) x 64(N) per 256-thread workgroup, 4 waves as 2(m) x 2(n), each one 32x32 accumulator of
//     v_mfma_f32_32x32x2_f32 (C rows come from A, C columns -- contiguous in memory -- sit on the lanes); small tiles
//     keep split-K -- whose fp32 atomics cost ~50 ns per 256-B wave-instruction per CU -- to the N = 768 shapes;
//   * K in 64-deep steps: 16 lanes x 16 B read one operand row's whole 256-B slice, 8 float4 per thread in flight;
//     two LDS stages, ONE barrier per step: while the MFMAs of step t run out of one stage, the registers holding
//     step t+1 are written to the other one in the shadow of the matrix pipe, then step t+2's loads are issued;
//   * operands sit in LDS in the order the MFMA eats them -- float4 (g, h, row) = the 4 k-pairs {8g+2s+h, s=0..3} of
//     one row -- so one ds_read_b128 per operand feeds 4 MFMAs; those reads run one k-group ahead of the MFMAs;
//   * split-K over blockIdx.z (fp32 atomics onto a zeroed C) sized by the host so that the grid is ~one workgroup per
//     CU and every workgroup runs only a few K-steps.
constexpr int GBM = 64, GBN = 64, GBK = 64;
constexpr int GPA = GBM + 1, GPB = GBN + 1;  // float4s per (g, h) plane (+1 pad)

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

// One K-step of both operands in flight: the raw float4s as loaded plus the K-tail masks that are applied when the
// registers are written to LDS (NOT at the load: any use of a loaded value at the issue site makes the wave wait there).
struct GemmRegs {
    float4 a[GBM / 16];
    float4 b[4];
    unsigned ka;     // all-ones where this thread's K positions of the [rows,K] operands are real
    unsigned kb[4];  // likewise for the 4 k-rows of a [K,N] B operand
};

template <bool TRANS_B, bool A_GELU>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];  // 2 stages x { A [g = 0..7][h = 0..1][row], B likewise }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int wm = wave >> 1, wn = wave & 1;
    // Four accumulators for ONE 32x32 output block, one per k-pair slot of a group (summed in the epilogue): keeps
    // consecutive MFMAs independent.  (Measured on this kernel and in tools/: runs of dependent MFMAs cost nothing
    // either -- the K-step's ~1.4 us is not MFMA issue.)
    f32x16 acc4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc4[q][r] = 0.f;

    // staging roles.  [rows,K] operands (A, and B when TRANS_B): thread = (row = tid/16 (+16 per pass), 4 consecutive k
    // at 4*(tid%16)): a wave-instruction reads 4 whole 256-B row slices.  [K,N] B: thread = (gh = tid/16, c4 = tid%16)
    // reads the 4 k-rows {8g+2s+h} of 4 consecutive columns (whole 256-B lines again) and transposes in registers.
    // Branch-free: out-of-range ROWS / COLUMNS are clamped to a valid one (they only feed accumulator rows / columns
    // that are never stored); an out-of-range K position is read from a clamped address and masked at the LDS write.
    const int s_row = tid >> 4, s_l = tid & 15;
    // Buffer loads: descriptor per operand + a per-thread byte offset computed once + the K position as the scalar
    // offset -- no per-step address arithmetic.  Rows past M / N (and, for a [K,N] B, k-rows past K) fall past the
    // descriptor and read 0.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), (short)0,
                                                                        (int)((unsigned)p.m * (unsigned)p.lda * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.b), (short)0, (int)((unsigned)(TRANS_B ? p.n : p.k) * (unsigned)p.ldb * 4u), 0x00020000);
    unsigned aoff[GBM / 16], boff[4];
#pragma unroll
    for (int it = 0; it < GBM / 16; ++it) aoff[it] = (unsigned)((m0 + s_row + 16 * it) * p.lda + 4 * s_l) * 4u;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        boff[it] = TRANS_B ? (unsigned)((n0 + s_row + 16 * it) * p.ldb + 4 * s_l) * 4u
                           : (unsigned)((8 * (s_row >> 1) + (s_row & 1) + 2 * it) * p.ldb + n0 + 4 * s_l) * 4u;
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) __attribute__((always_inline)) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto prefetch = [&](int k0, GemmRegs& R) __attribute__((always_inline)) {
        const int kk = k0 + 4 * s_l;
        R.ka = kk < p.k ? 0xffffffffu : 0u;  // a K position past the row's end would read the next row: masked at the LDS write
        const unsigned soff = (unsigned)k0 * 4u;
#pragma unroll
        for (int it = 0; it < GBM / 16; ++it) R.a[it] = ld4(ra, aoff[it], soff);
        if (TRANS_B) {
#pragma unroll
            for (int it = 0; it < GBN / 16; ++it) R.b[it] = ld4(rb, boff[it], soff);
        } else {
            const unsigned soffb = (unsigned)k0 * (unsigned)p.ldb * 4u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R.kb[q] = 0xffffffffu;  // k-rows past K are past the descriptor: zeros
                R.b[q] = ld4(rb, boff[q], soffb);
            }
        }
    };
    auto msk = [](float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); };
    // a float4 of 4 consecutive k (k%8 = 0 or 4) is two half-float4s of the LDS image: {k, k+2} -> h=0, {k+1, k+3} -> h=1
    const int w_g = s_l >> 1, w_s = (s_l & 1) * 2;
    // One piece of the register -> LDS write pass (8 pieces per K-step: 4 A row-passes, 4 B); the pieces are dealt to
    // the MFMA groups of the step before, so the LDS writes (and the QuickGELU prologue) issue in the shadow of the
    // matrix pipe.
    auto commit_piece = [&](int piece, const GemmRegs& R, float4* As4, float4* Bs4) __attribute__((always_inline)) {
        if (piece < GBM / 16) {
            const int it = piece;
            float4 v = R.a[it];
            v.x = msk(v.x, R.ka), v.y = msk(v.y, R.ka), v.z = msk(v.z, R.ka), v.w = msk(v.w, R.ka);
            if (A_GELU) v.x = quick_gelu(v.x), v.y = quick_gelu(v.y), v.z = quick_gelu(v.z), v.w = quick_gelu(v.w);
            const int row = s_row + 16 * it;
            float* d0 = reinterpret_cast<float*>(As4 + (w_g * 2 + 0) * GPA + row) + w_s;
            float* d1 = reinterpret_cast<float*>(As4 + (w_g * 2 + 1) * GPA + row) + w_s;
            *reinterpret_cast<float2*>(d0) = make_float2(v.x, v.z);
            *reinterpret_cast<float2*>(d1) = make_float2(v.y, v.w);
        } else if (TRANS_B) {
            const int it = piece - GBM / 16;
            const int row = s_row + 16 * it;
            float* d0 = reinterpret_cast<float*>(Bs4 + (w_g * 2 + 0) * GPB + row) + w_s;
            float* d1 = reinterpret_cast<float*>(Bs4 + (w_g * 2 + 1) * GPB + row) + w_s;
            *reinterpret_cast<float2*>(d0) = make_float2(msk(R.b[it].x, R.ka), msk(R.b[it].z, R.ka));
            *reinterpret_cast<float2*>(d1) = make_float2(msk(R.b[it].y, R.ka), msk(R.b[it].w, R.ka));
        } else {
            const int c = piece - GBM / 16;
            float4* d = Bs4 + s_row * GPB + 4 * s_l;
            const unsigned* km = R.kb;
            if (c == 0) d[0] = make_float4(msk(R.b[0].x, km[0]), msk(R.b[1].x, km[1]), msk(R.b[2].x, km[2]), msk(R.b[3].x, km[3]));
            if (c == 1) d[1] = make_float4(msk(R.b[0].y, km[0]), msk(R.b[1].y, km[1]), msk(R.b[2].y, km[2]), msk(R.b[3].y, km[3]));
            if (c == 2) d[2] = make_float4(msk(R.b[0].z, km[0]), msk(R.b[1].z, km[1]), msk(R.b[2].z, km[2]), msk(R.b[3].z, km[3]));
            if (c == 3) d[3] = make_float4(msk(R.b[0].w, km[0]), msk(R.b[1].w, km[1]), msk(R.b[2].w, km[2]), msk(R.b[3].w, km[3]));
        }
    };
    constexpr int STAGE = (GBK / 4) * (GPA + GPB);
    constexpr int NPIECE = GBM / 16 + 4;
    static_assert(NPIECE % 4 == 0, "pieces are dealt to the last 4 MFMA groups");

    // The slice runs an EVEN number of K-steps (k_per is a multiple of 128; past K the masks feed zeros), and every
    // step issues its prefetch unconditionally (a finished slice re-reads its first tile and drops it): a prefetch
    // under a condition, or a conditional second half of the unrolled loop, makes hipcc merge "loaded" and "kept"
    // values right after the loads -- which puts a vmcnt wait at the issue site and serialises the pipeline.
    const int k_lo = blockIdx.z * p.k_per;
    const int k_hi = (k_lo + p.k_per < p.k) ? k_lo + p.k_per : ((p.k - k_lo + 2 * GBK - 1) / (2 * GBK)) * (2 * GBK) + k_lo;

    // K-step t: MFMAs out of LDS stage t&1.  Its loads were issued at the START of step t-2 into register set t&1 and
    // written to the stage during the second half of step t-1 -- a step and a half of MFMAs to cover the round trip.
    // `Rload` = the set to refill with step t+2 (it held step t, already in LDS), `Rnext` = the set holding step t+1.
    auto step = [&](int k0, int stage, GemmRegs& Rload, const GemmRegs& Rnext) __attribute__((always_inline)) {
        const float4* As4 = gsm + stage * STAGE;
        const float4* Bs4 = As4 + (GBK / 4) * GPA;
        float4* An = gsm + (stage ^ 1) * STAGE;  // the other stage: every wave left it at the barrier that ended the last step
        float4* Bn = An + (GBK / 4) * GPA;
        const bool has_next = k0 + GBK < k_hi;
        const float4* ap = As4 + half * GPA + wm * 32 + j;
        const float4* bp = Bs4 + half * GPB + wn * 32 + j;
        float4 av[2], bv[2];
        av[0] = ap[0], bv[0] = bp[0];
#pragma unroll
        for (int g = 0; g < GBK / 8; ++g) {
            if (g + 1 < GBK / 8) av[(g + 1) & 1] = ap[(g + 1) * 2 * GPA], bv[(g + 1) & 1] = bp[(g + 1) * 2 * GPB];
            __builtin_amdgcn_sched_barrier(0);  // keep the next group's reads above this group's MFMAs
            const float4 a4 = av[g & 1], b4 = bv[g & 1];
            acc4[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc4[0], 0, 0, 0);
            acc4[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc4[1], 0, 0, 0);
            acc4[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc4[2], 0, 0, 0);
            acc4[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc4[3], 0, 0, 0);
            if (g == 0) prefetch(k0 + 2 * GBK < k_hi ? k0 + 2 * GBK : k_lo, Rload);  // in the shadow of the first MFMAs
            if (g >= GBK / 16 && has_next) {  // second half: step t+1 goes to the other stage, two pieces per MFMA group
#pragma unroll
                for (int q = 0; q < NPIECE / 4; ++q) commit_piece((g - GBK / 16) * (NPIECE / 4) + q, Rnext, An, Bn);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    GemmRegs R0, R1;
    prefetch(k_lo, R0);
    prefetch(k_lo + GBK, R1);
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) commit_piece(q, R0, gsm, gsm + (GBK / 4) * GPA);
    __syncthreads();
    for (int k0 = k_lo; k0 < k_hi; k0 += 2 * GBK) {
        step(k0, 0, R0, R1);
        step(k0 + GBK, 1, R1, R0);
    }
    // epilogue.  Split-K: slice 0 carries bias + residual, every slice adds atomically; the QuickGELU' factor is
    // linear in the sum, so each slice applies it to its own partial.  Loads before stores (one in-order vmcnt).
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (acc4[0][r] + acc4[1][r]) + (acc4[2][r] + acc4[3][r]);
    const bool first = blockIdx.z == 0;
    const int n = n0 + wn * 32 + j;
    if (m0 + wm * 32 >= p.m || n >= p.n) return;
    const float bs = (p.bias && first) ? p.bias[n] : 0.f;
    float rs[16], ax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int64_t ci = (int64_t)(m < p.m ? m : 0) * p.ldc + n;
        rs[r] = (p.residual && first) ? p.residual[ci] : 0.f;
        ax[r] = p.aux ? p.aux[ci] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.m) continue;
        const int64_t ci = (int64_t)m * p.ldc + n;
        float v = acc[r] + bs + rs[r];
        if (p.aux) v *= quick_gelu_grad(ax[r]);
        if (gridDim.z > 1) atomicAdd(&p.c[ci], v);
        else p.c[ci] = v;
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ float wave_sum_ln(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int LN_MAX_PER_LANE = 32;  // dim <= 2048

// The ViT widths (dim = 256 * T4: 768 -> T4 = 3): ONE wave per row and per workgroup (rows = 50 * batch workgroups spread
// over the chip: these kernels are pure latency), the row in registers as T4 float4s per lane, every load of the row issued
// before the first reduction.
template <int T4>
__global__ __launch_bounds__(64) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out, int dim,
                                                               float eps) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float4* xr = reinterpret_cast<const float4*>(x + row * dim);
    float4 v[T4], g[T4], bt[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) v[t] = xr[lane + 64 * t];
#pragma unroll
    for (int t = 0; t < T4; ++t) g[t] = reinterpret_cast<const float4*>(gamma)[lane + 64 * t], bt[t] = reinterpret_cast<const float4*>(beta)[lane + 64 * t];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    const float mean = wave_sum_ln(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        v[t].x -= mean, v[t].y -= mean, v[t].z -= mean, v[t].w -= mean;
        q += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
    }
    const float rstd = rsqrtf(wave_sum_ln(q) / dim + eps);
    float4* yr = reinterpret_cast<float4*>(y + row * dim);
#pragma unroll
    for (int t = 0; t < T4; ++t)
        yr[lane + 64 * t] = make_float4(v[t].x * rstd * g[t].x + bt[t].x, v[t].y * rstd * g[t].y + bt[t].y, v[t].z * rstd * g[t].z + bt[t].z,
                                        v[t].w * rstd * g[t].w + bt[t].w);
    if (lane == 0) mean_out[row] = mean, rstd_out[row] = rstd;
}

template <int T4>
__global__ __launch_bounds__(64) void layernorm_bwd_vec_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ add,
                                                               float* __restrict__ gx, int dim) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float4* gr = reinterpret_cast<const float4*>(gy + row * dim);
    const float4* xr = reinterpret_cast<const float4*>(x + row * dim);
    const float4* ar = add ? reinterpret_cast<const float4*>(add + row * dim) : nullptr;
    float4 gg[T4], xh[T4], ad[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) gg[t] = gr[lane + 64 * t], xh[t] = xr[lane + 64 * t];
#pragma unroll
    for (int t = 0; t < T4; ++t) ad[t] = ar ? ar[lane + 64 * t] : make_float4(0.f, 0.f, 0.f, 0.f);
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
    s1 = wave_sum_ln(s1) / dim;
    s2 = wave_sum_ln(s2) / dim;
    float4* out = reinterpret_cast<float4*>(gx + row * dim);
#pragma unroll
    for (int t = 0; t < T4; ++t)
        out[lane + 64 * t] = make_float4(rs * (gg[t].x - s1 - xh[t].x * s2) + ad[t].x, rs * (gg[t].y - s1 - xh[t].y * s2) + ad[t].y,
                                         rs * (gg[t].z - s1 - xh[t].z * s2) + ad[t].z, rs * (gg[t].w - s1 - xh[t].w * s2) + ad[t].w);
}

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
                                                            const float* __restrict__ rstd, const float* __restrict__ add,
                                                            float* __restrict__ gx, int64_t rows, int dim) {
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
        if (c < dim) gx[row * dim + c] = rs * (gg[t] - s1 - xh[t] * s2) + (add ? add[row * dim + c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------ attention
// One workgroup per (batch, head); L <= 64 tokens, head dim 64.  Everything lives in LDS with 68-float rows (16-B
// aligned, consecutive rows shifted by 16 B: conflict-free ds_read_b128) and every contraction is register-tiled
// over float4s -- 16 FMAs per 5 LDS reads instead of 1 per 2, which is what these LDS-bound loops are limited by.
constexpr int HD = 64, LMAX = 64, HS = 68;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, const float4 v) {
    acc.x += s * v.x, acc.y += s * v.y, acc.z += s * v.z, acc.w += s * v.w;
}

// zero `n4` float4s of LDS (rows >= L of every operand must read as zero: the tiled loops run to a multiple of 4)
__device__ __forceinline__ void lds_zero(float* base, int n4) {
    for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(base)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// qkv [B, L, 3, H, 64] -> LDS q, k, v as [L][HS]
__device__ __forceinline__ void load_qkv(const float* qkv, int b, int h, int L, int H, float* q, float* k, float* v) {
    for (int e = threadIdx.x; e < L * 16; e += 256) {
        const int t = e >> 4, d = (e & 15) * 4;
        const float* src = qkv + (((int64_t)b * L + t) * 3 * H + h) * HD + d;
        *reinterpret_cast<float4*>(q + t * HS + d) = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(k + t * HS + d) = *reinterpret_cast<const float4*>(src + (int64_t)H * HD);
        *reinterpret_cast<float4*>(v + t * HS + d) = *reinterpret_cast<const float4*>(src + (int64_t)2 * H * HD);
    }
}

// out[i][j] = scale * a_i . b_j for i < L, j < roundup4(L)   (a, b, out: [.][HS])
__device__ __forceinline__ void rows_dot_rows(const float* a, const float* b, float* out, int L, float scale) {
    const int L4 = (L + 3) >> 2;
    for (int e = threadIdx.x; e < L * L4; e += 256) {
        const int i = e / L4, j = (e - i * L4) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int d = 0; d < HD; d += 4) {
            const float4 av = *reinterpret_cast<const float4*>(a + i * HS + d);
            s.x += dot4(av, *reinterpret_cast<const float4*>(b + (j + 0) * HS + d));
            s.y += dot4(av, *reinterpret_cast<const float4*>(b + (j + 1) * HS + d));
            s.z += dot4(av, *reinterpret_cast<const float4*>(b + (j + 2) * HS + d));
            s.w += dot4(av, *reinterpret_cast<const float4*>(b + (j + 3) * HS + d));
        }
        *reinterpret_cast<float4*>(out + i * HS + j) = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
    }
}

// acc[d..d+3] = sum_u w[t][u] * m[u][d..d+3]  (row weights)  or  sum_u w[u][t] * m[u][d..d+3]  (column weights)
template <bool COLUMN>
__device__ __forceinline__ float4 weighted_rows(const float* w, const float* m, int t, int d, int L) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = 0; u < L; u += 4) {  // rows >= L of m are zero
        float w0, w1, w2, w3;
        if (COLUMN) {
            w0 = w[(u + 0) * HS + t], w1 = w[(u + 1) * HS + t], w2 = w[(u + 2) * HS + t], w3 = w[(u + 3) * HS + t];
        } else {
            const float4 wv = *reinterpret_cast<const float4*>(w + t * HS + u);
            w0 = wv.x, w1 = wv.y, w2 = wv.z, w3 = wv.w;
        }
        fma4(acc, w0, *reinterpret_cast<const float4*>(m + (u + 0) * HS + d));
        fma4(acc, w1, *reinterpret_cast<const float4*>(m + (u + 1) * HS + d));
        fma4(acc, w2, *reinterpret_cast<const float4*>(m + (u + 2) * HS + d));
        fma4(acc, w3, *reinterpret_cast<const float4*>(m + (u + 3) * HS + d));
    }
    return acc;
}

// p[i][j] = softmax_j(q_i . k_j / 8); columns L .. roundup4(L)-1 are set to 0
__device__ __forceinline__ void softmax_probs(const float* q, const float* k, float* p, int L) {
    rows_dot_rows(q, k, p, L, 0.125f);
    __syncthreads();
    if (threadIdx.x < L) {
        float* row = p + threadIdx.x * HS;
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
        for (int jj = L; jj < ((L + 3) & ~3); ++jj) row[jj] = 0.f;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem;
    float* k = q + LMAX * HS;
    float* v = k + LMAX * HS;
    float* p = v + LMAX * HS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    lds_zero(smem, 4 * LMAX * HS / 4);
    __syncthreads();
    load_qkv(qkv, b, h, L, H, q, k, v);
    __syncthreads();
    softmax_probs(q, k, p, L);
    for (int e = threadIdx.x; e < L * 16; e += 256) {
        const int i = e >> 4, d = (e & 15) * 4;
        const float4 o = weighted_rows<false>(p, v, i, d, L);
        *reinterpret_cast<float4*>(out + (((int64_t)b * L + i) * H + h) * HD + d) = o;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ gout,
                                                       float* __restrict__ gqkv, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem;
    float* k = q + LMAX * HS;
    float* v = k + LMAX * HS;
    float* go = v + LMAX * HS;
    float* p = go + LMAX * HS;
    float* ds = p + LMAX * HS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    lds_zero(smem, 6 * LMAX * HS / 4);
    __syncthreads();
    load_qkv(qkv, b, h, L, H, q, k, v);
    for (int e = threadIdx.x; e < L * 16; e += 256) {
        const int t = e >> 4, d = (e & 15) * 4;
        *reinterpret_cast<float4*>(go + t * HS + d) = *reinterpret_cast<const float4*>(gout + (((int64_t)b * L + t) * H + h) * HD + d);
    }
    __syncthreads();
    softmax_probs(q, k, p, L);
    rows_dot_rows(go, v, ds, L, 1.f);  // dP[i][j] = go_i . v_j
    __syncthreads();
    if (threadIdx.x < L) {  // dS = P * (dP - sum_j dP*P)
        const int i = threadIdx.x;
        float dot = 0.f;
        for (int jj = 0; jj < L; ++jj) dot += ds[i * HS + jj] * p[i * HS + jj];
        for (int jj = 0; jj < L; ++jj) ds[i * HS + jj] = p[i * HS + jj] * (ds[i * HS + jj] - dot);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * 16; e += 256) {
        const int t = e >> 4, d = (e & 15) * 4;
        const float4 gq = weighted_rows<false>(ds, k, t, d, L);  // dQ_t = sum_j dS[t][j] K_j
        const float4 gk = weighted_rows<true>(ds, q, t, d, L);   // dK_t = sum_i dS[i][t] Q_i
        const float4 gv = weighted_rows<true>(p, go, t, d, L);   // dV_t = sum_i P[i][t] dO_i
        float* dst = gqkv + (((int64_t)b * L + t) * 3 * H + h) * HD + d;
        *reinterpret_cast<float4*>(dst) = make_float4(gq.x * 0.125f, gq.y * 0.125f, gq.z * 0.125f, gq.w * 0.125f);
        *reinterpret_cast<float4*>(dst + (int64_t)H * HD) = make_float4(gk.x * 0.125f, gk.y * 0.125f, gk.z * 0.125f, gk.w * 0.125f);
        *reinterpret_cast<float4*>(dst + (int64_t)2 * H * HD) = gv;
    }
}

}  // namespace w2e

using namespace w2e;

// Dynamic LDS above 64 KB has to be enabled per kernel (gfx950 has 160 KB per CU).

extern "C" int w2e_gemm_ex(const float* a, const float* b, float* c, int m, int n, int k, int lda, int ldb, int ldc,
                           int trans_b, int a_gelu, const float* bias, const float* residual, const float* gelu_grad_aux,
                           int c_is_zero, void* stream) {
    W2E_REQUIRE(a && b && c, "gemm: null tensor");
    W2E_REQUIRE(m >= 0 && n > 0 && k > 0, "gemm: bad dims %d %d %d", m, n, k);
    W2E_REQUIRE((k & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0, "gemm: K, lda, ldb must be multiples of 4");
    W2E_REQUIRE(trans_b || (n & 3) == 0, "gemm: N must be a multiple of 4 for a [K,N] B operand");
    W2E_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "gemm: operands must be 16-byte aligned");
    W2E_REQUIRE(!(a_gelu && !trans_b), "gemm: the QuickGELU prologue is implemented for the [N,K] form only");
    if (m == 0) return 0;
    W2E_REQUIRE((int64_t)m * lda * 4 < ((int64_t)1 << 32) && (int64_t)(trans_b ? n : k) * ldb * 4 < ((int64_t)1 << 32),
                "gemm: an operand exceeds 4 GB");
    hipStream_t s = (hipStream_t)stream;
    // Small-M GEMMs (M = 50*batch) leave most CUs idle with one workgroup per 64x64 tile and make every wave a
    // K/2-long dependent MFMA chain: split K over blockIdx.z (fp32 atomics onto a zeroed C) until the grid fills the chip.
    const int64_t tiles = ceil_div(n, GBN) * ceil_div(m, GBM);
    int splits = 1;
    if (ldc == n) {
        // per workgroup: ~0.9 us per 64-deep K-step (32 MFMAs per wave) + ~2 us of prologue/epilogue; a split adds the
        // memset launch and the atomics (64 wave-instructions per workgroup at ~50 ns each).  Units: microseconds.
        const int tune_s = options().tune_gemm_s;
        double best = 0.0;
        for (int sp = 1; sp <= 16; ++sp) {
            const int kp = (int)(ceil_div(ceil_div(k, sp), 2 * GBK) * 2 * GBK);
            if (sp > 1 && ceil_div(k, kp) != sp) continue;
            const double per_cu = (double)ceil_div(tiles * sp, 256);
            const double cost = per_cu * ((kp / GBK) * 0.9 + (sp > 1 ? 3.2 : 0.0)) + 2.0 + (sp > 1 ? 3.0 : 0.0);
            if (sp == 1 || cost < best * 0.97) best = cost, splits = sp;
        }
        if (tune_s > 0) splits = tune_s;
        if (options().deterministic) splits = 1;  // no fp32 atomics onto C
    }
    const int k_per = (int)(ceil_div(ceil_div(k, splits), 2 * GBK) * 2 * GBK);  // an even number of 64-deep steps
    splits = (int)ceil_div(k, k_per);
    if (splits > 1 && !c_is_zero && zero_async(c, sizeof(float) * (size_t)m * n, s) != hipSuccess) {
        set_error("gemm: memset failed");
        return 2;
    }
    GemmParams p{a, b, c, m, n, k, lda, ldb, ldc, bias, residual, gelu_grad_aux, k_per};
    dim3 grid((unsigned)ceil_div(n, GBN), (unsigned)ceil_div(m, GBM), (unsigned)splits);
    constexpr size_t lds = sizeof(float4) * 2 * (GBK / 4) * (GPA + GPB);  // two stages, 99 KB
    static unsigned done[3] = {0, 0, 0};  // per (kernel, device)
    const bool lds_ok = big_lds_once((const void*)gemm_kernel<true, true>, &done[0]) &&
                        big_lds_once((const void*)gemm_kernel<true, false>, &done[1]) &&
                        big_lds_once((const void*)gemm_kernel<false, false>, &done[2]);
    W2E_REQUIRE(lds_ok, "gemm: cannot enable %zu B of dynamic LDS", lds);
    if (trans_b) {
        if (a_gelu) gemm_kernel<true, true><<<grid, 256, lds, s>>>(p);
        else gemm_kernel<true, false><<<grid, 256, lds, s>>>(p);
    } else {
        gemm_kernel<false, false><<<grid, 256, lds, s>>>(p);
    }
    W2E_LAUNCH_CHECK("gemm");
    return 0;
}

extern "C" int w2e_gemm(const float* a, const float* b, float* c, int m, int n, int k, int lda, int ldb, int ldc,
                        int trans_b, int a_gelu, const float* bias, const float* residual, const float* gelu_grad_aux,
                        void* stream) {
    return w2e_gemm_ex(a, b, c, m, n, k, lda, ldb, ldc, trans_b, a_gelu, bias, residual, gelu_grad_aux, 0, stream);
}

extern "C" int w2e_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int64_t rows, int dim, float eps, void* stream) {
    W2E_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null tensor");
    W2E_REQUIRE(dim > 0 && dim <= 64 * LN_MAX_PER_LANE, "layernorm_fwd: dim %d unsupported (max %d)", dim, 64 * LN_MAX_PER_LANE);
    if (rows <= 0) return 0;
    const bool al = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) && rows < ((int64_t)1 << 31);
    hipStream_t st = (hipStream_t)stream;
    if (al && dim == 768) layernorm_fwd_vec_kernel<3><<<(unsigned)rows, 64, 0, st>>>(x, gamma, beta, y, mean, rstd, dim, eps);
    else if (al && dim == 512) layernorm_fwd_vec_kernel<2><<<(unsigned)rows, 64, 0, st>>>(x, gamma, beta, y, mean, rstd, dim, eps);
    else if (al && dim == 1024) layernorm_fwd_vec_kernel<4><<<(unsigned)rows, 64, 0, st>>>(x, gamma, beta, y, mean, rstd, dim, eps);
    else layernorm_fwd_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, st>>>(x, gamma, beta, y, mean, rstd, rows, dim, eps);
    W2E_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int w2e_layernorm_bwd_add(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     const float* add, float* gx, int64_t rows, int dim, void* stream) {
    W2E_REQUIRE(gy && x && gamma && mean && rstd && gx, "layernorm_bwd: null tensor");
    W2E_REQUIRE(dim > 0 && dim <= 64 * LN_MAX_PER_LANE, "layernorm_bwd: dim %d unsupported", dim);
    if (rows <= 0) return 0;
    const bool al = ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)gx | (uintptr_t)(add ? add : x)) & 15) == 0) &&
                    rows < ((int64_t)1 << 31);
    hipStream_t st = (hipStream_t)stream;
    if (al && dim == 768) layernorm_bwd_vec_kernel<3><<<(unsigned)rows, 64, 0, st>>>(gy, x, gamma, mean, rstd, add, gx, dim);
    else if (al && dim == 512) layernorm_bwd_vec_kernel<2><<<(unsigned)rows, 64, 0, st>>>(gy, x, gamma, mean, rstd, add, gx, dim);
    else if (al && dim == 1024) layernorm_bwd_vec_kernel<4><<<(unsigned)rows, 64, 0, st>>>(gy, x, gamma, mean, rstd, add, gx, dim);
    else layernorm_bwd_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, st>>>(gy, x, gamma, mean, rstd, add, gx, rows, dim);
    W2E_LAUNCH_CHECK("layernorm_bwd");
    return 0;
}

extern "C" int w2e_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 float* gx, int64_t rows, int dim, void* stream) {
    return w2e_layernorm_bwd_add(gy, x, gamma, mean, rstd, nullptr, gx, rows, dim, stream);
}

extern "C" int w2e_attn_fwd(const float* qkv, float* out, int batch, int seq, int heads, void* stream) {
    W2E_REQUIRE(qkv && out, "attn_fwd: null tensor");
    W2E_REQUIRE(seq > 0 && seq <= LMAX && heads > 0 && batch >= 0, "attn_fwd: seq %d (max %d), heads %d", seq, LMAX, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 4 * LMAX * HS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn_fwd_kernel, &done), "attn_fwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn_fwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, out, seq, heads);
    W2E_LAUNCH_CHECK("attn_fwd");
    return 0;
}

extern "C" int w2e_attn_bwd(const float* qkv, const float* gout, float* gqkv, int batch, int seq, int heads, void* stream) {
    W2E_REQUIRE(qkv && gout && gqkv, "attn_bwd: null tensor");
    W2E_REQUIRE(seq > 0 && seq <= LMAX && heads > 0 && batch >= 0, "attn_bwd: seq %d (max %d), heads %d", seq, LMAX, heads);
    if (batch == 0) return 0;
    const size_t lds = sizeof(float) * 6 * LMAX * HS;
    static unsigned done = 0;
    W2E_REQUIRE(big_lds_once((const void*)attn_bwd_kernel, &done), "attn_bwd: cannot raise the dynamic LDS limit to %zu B", lds);
    attn_bwd_kernel<<<batch * heads, 256, lds, (hipStream_t)stream>>>(qkv, gout, gqkv, seq, heads);
    W2E_LAUNCH_CHECK("attn_bwd");
    return 0;
}
