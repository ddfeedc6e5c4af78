// CLIP ViT-B/32 tower, third-generation GEMM (include/w2e_vit.h): C[M,N] = A[M,K] x W[N,K]^T at M = 50*batch with BOTH operands
// pre-packed "K-quad-major" -- P[q][r] (one float4) = X[r][4q .. 4q+3] -- which is exactly the word a lane of v_mfma_f32_32x32x2_f32 feeds
// to four consecutive MFMAs (k-slot convention of vit2.hip: lane-half h, group g, component c <-> k = 8g + 4h + c, i.e. quad q = 2g + h).
// Operands therefore go global/L2 -> REGISTERS, 512 contiguous bytes per half-wave, with no LDS, no barrier and no cross-wave dependency:
// every wave is an independent 32-row x 64-column tile of one K slice (the structure that took the Winograd contraction, winogemm.hip,
// from 0.49 to 0.87 matrix-pipe occupancy).  gemm_fm (vit2.hip) spends ~10 of its ~15 us per launch outside its MFMAs: 7 waves on 4 SIMDs,
// a barrier and a DMA issue burst per 32-deep step, a ring of 2.  Here: ~1000 single-wave tiles per launch (one per SIMD), a 4-chunk register
// ring (2 k cycles of prefetch distance), split-K slabs written exactly as gemm_fm's (the consumers sum them in order: no atomics).
// The weights are packed once (frozen critic); the activations are WRITTEN packed by their producers (w2e_pack_kq is the stand-alone pass).
#include "common.h"
#include "../../include/w2e_vit.h"

namespace w2e {

typedef float pk_f32x16 __attribute__((ext_vector_type(16)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));

// P[q][r] = X[r][4q..4q+3] for r < rows (zero for rows <= r < rpad); one thread per (q, r), r fastest: coalesced 16-byte stores.
__global__ __launch_bounds__(256) void pack_kq_kernel(const float* __restrict__ x, float* __restrict__ p, int rows, int rpad, int K, int ldx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(K >> 2) * rpad;
    if (e >= total) return;
    const int q = (int)(e / rpad), r = (int)(e - (int64_t)q * rpad);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + 4 * q);
    reinterpret_cast<float4*>(p)[e] = v;
}

// P[q][r] = (X[4q][r], X[4q+1][r], X[4q+2][r], X[4q+3][r]): the packed form of X^T (the input-gradient GEMMs contract over the weight's
// OUTPUT axis).  x [K, rows] row-major with row stride ldx.
__global__ __launch_bounds__(256) void pack_kq_t_kernel(const float* __restrict__ x, float* __restrict__ p, int rows, int rpad, int K, int ldx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(K >> 2) * rpad;
    if (e >= total) return;
    const int q = (int)(e / rpad), r = (int)(e - (int64_t)q * rpad);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
        const float* c = x + (int64_t)(4 * q) * ldx + r;
        v = make_float4(c[0], c[ldx], c[2 * (int64_t)ldx], c[3 * (int64_t)ldx]);
    }
    reinterpret_cast<float4*>(p)[e] = v;
}

struct GemmPkParams {
    const float* a;  // packed [K/4][mpad][4]
    const float* b;  // packed [K/4][npad][4]
    float* c;        // [splits][m][ldc]
    int m, n, k, mpad, npad, ldc;
    int chunks_per;  // 8-deep chunks per K slice (a multiple of 4)
    int n_rb, n_cb, splits;
    int64_t slab;
};

// One WAVE = rows 32*rb .. +31, columns 64*cb .. +63 of K slice z.  Wave id w -> rb fastest, then z, then cb: the n_rb waves that share a
// (cb, z) B panel are neighbours (same workgroup or the next), and so is the L2 that serves them.
__global__ __launch_bounds__(256) void gemm_pk_kernel(const GemmPkParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int per_cb = p.n_rb * p.splits;
    const int cb = wave / per_cb, rem = wave - cb * per_cb;
    const int z = rem / p.n_rb, rb = rem - z * p.n_rb;
    if (cb >= p.n_cb) return;
    const int half = lane >> 5, j = lane & 31;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), (short)0, (int)(unsigned)((int64_t)(p.k >> 2) * p.mpad * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), (short)0, (int)(unsigned)((int64_t)(p.k >> 2) * p.npad * 16), 0x00020000);
    // quad q = 2*chunk + half: byte offset (q * pad + row) * 16
    const int voff_a = (half * p.mpad + 32 * rb + j) * 16;
    const int voff_b = (half * p.npad + 64 * cb + j) * 16;
    const unsigned stride_a = (unsigned)(2 * p.mpad) * 16u, stride_b = (unsigned)(2 * p.npad) * 16u;  // bytes per chunk
    const int c0 = z * p.chunks_per;
    // The chunk offset travels in the VECTOR offset: the hardware range check covers vector + instruction offset only (a scalar offset is
    // added unchecked), and chunks past K -- the tail of the last slice -- must read zeros, not whatever lies behind the operand.
    const int chunks = p.k >> 3;
    auto ld_a = [&](int ch) __attribute__((always_inline)) {
        const unsigned off = ch < chunks ? (unsigned)voff_a + (unsigned)ch * stride_a : 0xfffffff0u;
        return __builtin_bit_cast(pk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)off, 0, 0));
    };
    auto ld_b = [&](int ch, int t) __attribute__((always_inline)) {
        const unsigned off = ch < chunks ? (unsigned)voff_b + (unsigned)t * 512u + (unsigned)ch * stride_b : 0xfffffff0u;
        return __builtin_bit_cast(pk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)off, 0, 0));
    };
    pk_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
    pk_f32x4 a[4], b0[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (in the loop's order, pinned: counted waits on both loop entries -- winogemm.hip)
        a[i] = ld_a(c0 + i), b0[i] = ld_b(c0 + i, 0), b1[i] = ld_b(c0 + i, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // chunks past the end of K fall past the descriptors and read zeros: no branch around the re-loads
    for (int ch = c0; ch < c0 + p.chunks_per; ch += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][0], b0[i][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][0], b1[i][0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][1], b0[i][1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][1], b1[i][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][2], b0[i][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][2], b1[i][2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][3], b0[i][3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][3], b1[i][3], acc1, 0, 0, 0);
            const int nx = ch + i + 4 < c0 + p.chunks_per ? ch + i + 4 : chunks;  // (past this slice: nothing to fetch -- an out-of-range offset)
            a[i] = ld_a(nx), b0[i] = ld_b(nx, 0), b1[i] = ld_b(nx, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane holds column 64*cb + 32*t + j of rows 32*rb + (r & 3) + 8 * (r >> 2) + 4 * half
    float* c = p.c + (int64_t)z * p.slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.m) continue;
        const int n0 = 64 * cb + j;
        if (n0 < p.n) c[(int64_t)m * p.ldc + n0] = acc0[r];
        if (n0 + 32 < p.n) c[(int64_t)m * p.ldc + n0 + 32] = acc1[r];
    }
}

// ---- opt-in fp16 operands (w2e_gemm_pk_h): what the reference computes on a GPU -- criteria/clip_loss.py:10 loads CLIP with device="cuda",
// i.e. OpenAI's fp16 tower -- for the four Linear layers of a block: fp16 weights (packed once: half the bytes), the activations converted
// fp32 -> fp16 (round to nearest even) in registers on their way into the MFMA, fp32 accumulation on v_mfma_f32_32x32x16_f16.  Everything
// around the GEMMs (LayerNorm, attention, GELU, the residual stream, the slab sums) stays fp32.  One 16-deep step takes the A words of two
// consecutive 8-deep chunks of the fp32 packing -- lane-half h then holds k = 16s + 4h + c and 16s + 8 + 4h + c, c = 0 .. 3 -- and the
// weight buffer PH[s][h][row] (8 halves = 16 bytes) holds W[row] at exactly those k: any k assignment serves a contraction as long as both
// operands use the same one.
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));

// PH[(s * 2 + h) * rpad + r] = halves of X[r][16s + 4h + c], X[r][16s + 8 + 4h + c], c = 0..3 (transposed: of X[k][r]); zero for r >= rows
__global__ __launch_bounds__(256) void pack_kq_h_kernel(const float* __restrict__ x, pk_f16x8* __restrict__ p, int rows, int rpad, int K, int ldx,
                                                        int transposed) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(K >> 3) * rpad;
    if (e >= total) return;
    const int sh = (int)(e / rpad), r = (int)(e - (int64_t)sh * rpad);
    const int s16 = sh >> 1, h = sh & 1;
    pk_f16x8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = 16 * s16 + 8 * (c >> 2) + 4 * h + (c & 3);
        float f = 0.f;
        if (r < rows) f = transposed ? x[(int64_t)k * ldx + r] : x[(int64_t)r * ldx + k];
        v[c] = (_Float16)f;
    }
    p[e] = v;
}

// the same wave decomposition as gemm_pk_kernel; `b` is the fp16 pack, chunks_per counts 16-deep STEPS (a multiple of 4)
__global__ __launch_bounds__(256) void gemm_pk_h_kernel(const GemmPkParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int per_cb = p.n_rb * p.splits;
    const int cb = wave / per_cb, rem = wave - cb * per_cb;
    const int z = rem / p.n_rb, rb = rem - z * p.n_rb;
    if (cb >= p.n_cb) return;
    const int half = lane >> 5, j = lane & 31;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), (short)0, (int)(unsigned)((int64_t)(p.k >> 2) * p.mpad * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), (short)0, (int)(unsigned)((int64_t)(p.k >> 3) * p.npad * 16), 0x00020000);
    const int voff_a = (half * p.mpad + 32 * rb + j) * 16;   // fp32 pack: quad q = 2 * chunk + half
    const int voff_b = (half * p.npad + 64 * cb + j) * 16;   // fp16 pack: entry (2 * step + half) * npad + column
    const unsigned stride_a = (unsigned)(2 * p.mpad) * 16u;  // bytes per 8-deep chunk of A: a step is two of them
    const unsigned stride_b = (unsigned)(2 * p.npad) * 16u;  // bytes per 16-deep step of B
    const int c0 = z * p.chunks_per;
    const int steps = p.k >> 4;
    auto ld_a = [&](int st, int which) __attribute__((always_inline)) {
        const unsigned off = st < steps ? (unsigned)voff_a + (unsigned)(2 * st + which) * stride_a : 0xfffffff0u;
        return __builtin_bit_cast(pk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)off, 0, 0));
    };
    auto ld_b = [&](int st, int t) __attribute__((always_inline)) {
        const unsigned off = st < steps ? (unsigned)voff_b + (unsigned)t * 512u + (unsigned)st * stride_b : 0xfffffff0u;
        return __builtin_bit_cast(pk_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)off, 0, 0));
    };
    pk_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
    pk_f32x4 a0[4], a1[4];
    pk_f16x8 b0[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a0[i] = ld_a(c0 + i, 0), a1[i] = ld_a(c0 + i, 1), b0[i] = ld_b(c0 + i, 0), b1[i] = ld_b(c0 + i, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int st = c0; st < c0 + p.chunks_per; st += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pk_f16x8 ah;
#pragma unroll
            for (int c = 0; c < 4; ++c) ah[c] = (_Float16)a0[i][c], ah[4 + c] = (_Float16)a1[i][c];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1[i], acc1, 0, 0, 0);
            const int nx = st + i + 4 < c0 + p.chunks_per ? st + i + 4 : steps;
            a0[i] = ld_a(nx, 0), a1[i] = ld_a(nx, 1), b0[i] = ld_b(nx, 0), b1[i] = ld_b(nx, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* c = p.c + (int64_t)z * p.slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.m) continue;
        const int n0 = 64 * cb + j;
        if (n0 < p.n) c[(int64_t)m * p.ldc + n0] = acc0[r];
        if (n0 + 32 < p.n) c[(int64_t)m * p.ldc + n0 + 32] = acc1[r];
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_pack_kq(const float* x, float* packed, int rows, int rows_padded, int k, int ldx, int transposed, void* stream) {
    W2E_REQUIRE(x && packed, "pack_kq: null tensor");
    W2E_REQUIRE(rows >= 0 && rows_padded >= rows && k > 0 && (k & 3) == 0, "pack_kq: bad dims (K %% 4 == 0)");
    W2E_REQUIRE(transposed || ((ldx & 3) == 0 && ((uintptr_t)x & 15) == 0), "pack_kq: rows must be 16-byte aligned");
    W2E_REQUIRE(((uintptr_t)packed & 15) == 0, "pack_kq: output must be 16-byte aligned");
    const int64_t total = (int64_t)(k >> 2) * rows_padded;
    if (total == 0) return 0;
    W2E_REQUIRE(ceil_div(total, 256) < ((int64_t)1 << 31), "pack_kq: too large");
    if (transposed) pack_kq_t_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(x, packed, rows, rows_padded, k, ldx);
    else pack_kq_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(x, packed, rows, rows_padded, k, ldx);
    W2E_LAUNCH_CHECK("pack_kq");
    return 0;
}

int w2e_gemm_pk_splits(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 1;
    const int64_t tiles = ceil_div(m, 32) * ceil_div(n, 64);
    const int chunks = k >> 3;
    const int64_t simds = (int64_t)cu_count() * 4;
    // Measured (tools/gemm_pk_bench.py): a launch takes rounds x chunks-per-slice MFMA blocks, rounds = waves / SIMDs rounded up (waves
    // that share a SIMD share its matrix pipe), plus ~0.6 of a 4-chunk block per extra slab (its stores, and the consumer's pass over it).
    int best = 1;
    double best_cost = 0.0;
    for (int sp = 1; sp <= 16 && sp <= chunks / 4; ++sp) {
        const int per = (int)(ceil_div(ceil_div(chunks, sp), 4) * 4);
        if ((int64_t)(sp - 1) * per >= chunks) continue;
        const double cost = (double)ceil_div(tiles * sp, simds) * per + 2.5 * sp;
        if (sp == 1 || cost < best_cost) best = sp, best_cost = cost;
    }
    return best;
}

int w2e_gemm_pk(const float* a_packed, const float* b_packed, float* c, int m, int n, int k, int m_padded, int n_padded, int ldc,
                int splits, void* stream) {
    W2E_REQUIRE(a_packed && b_packed && c, "gemm_pk: null tensor");
    W2E_REQUIRE(m >= 0 && n > 0 && k > 0 && (k & 7) == 0, "gemm_pk: bad dims (K %% 8 == 0)");
    W2E_REQUIRE(ldc >= n, "gemm_pk: ldc %d < n %d (rows of a slab would overlap)", ldc, n);
    W2E_REQUIRE(m_padded >= m && (m_padded & 31) == 0 && n_padded >= n && (n_padded & 63) == 0, "gemm_pk: m_padded %% 32 == 0, n_padded %% 64 == 0");
    W2E_REQUIRE((((uintptr_t)a_packed | (uintptr_t)b_packed) & 15) == 0, "gemm_pk: operands must be 16-byte aligned");
    W2E_REQUIRE((int64_t)(k >> 2) * m_padded * 16 < ((int64_t)1 << 32) - 64 && (int64_t)(k >> 2) * n_padded * 16 < ((int64_t)1 << 32) - 64, "gemm_pk: an operand exceeds 4 GB");
    if (m == 0) return 0;
    const int chunks = k >> 3;
    W2E_REQUIRE(splits >= 1 && splits <= chunks, "gemm_pk: %d splits of %d chunks", splits, chunks);
    GemmPkParams p{};
    p.a = a_packed, p.b = b_packed, p.c = c, p.m = m, p.n = n, p.k = k, p.mpad = m_padded, p.npad = n_padded, p.ldc = ldc;
    p.chunks_per = (int)(ceil_div(ceil_div(chunks, splits), 4) * 4);
    W2E_REQUIRE((int64_t)(splits - 1) * p.chunks_per < chunks, "gemm_pk: %d splits leave an empty slice of K = %d", splits, k);
    p.n_rb = (int)ceil_div(m, 32), p.n_cb = (int)ceil_div(n, 64), p.splits = splits;
    p.slab = (int64_t)m * ldc;
    const int64_t waves = (int64_t)p.n_rb * p.n_cb * splits;
    gemm_pk_kernel<<<(unsigned)ceil_div(waves, 4), 256, 0, (hipStream_t)stream>>>(p);
    W2E_LAUNCH_CHECK("gemm_pk");
    return 0;
}

int w2e_pack_kq_h(const float* x, void* packed_half, int rows, int rows_padded, int k, int ldx, int transposed, void* stream) {
    W2E_REQUIRE(x && packed_half, "pack_kq_h: null tensor");
    W2E_REQUIRE(rows >= 0 && rows_padded >= rows && k > 0 && (k & 15) == 0, "pack_kq_h: bad dims (K %% 16 == 0)");
    W2E_REQUIRE(((uintptr_t)packed_half & 15) == 0, "pack_kq_h: output must be 16-byte aligned");
    const int64_t total = (int64_t)(k >> 3) * rows_padded;
    if (total == 0) return 0;
    W2E_REQUIRE(ceil_div(total, 256) < ((int64_t)1 << 31), "pack_kq_h: too large");
    pack_kq_h_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(x, reinterpret_cast<pk_f16x8*>(packed_half), rows, rows_padded, k, ldx,
                                                                                      transposed);
    W2E_LAUNCH_CHECK("pack_kq_h");
    return 0;
}

int w2e_gemm_pk_h(const float* a_packed, const void* b_packed_half, float* c, int m, int n, int k, int m_padded, int n_padded, int ldc,
                  int splits, void* stream) {
    W2E_REQUIRE(a_packed && b_packed_half && c, "gemm_pk_h: null tensor");
    W2E_REQUIRE(m >= 0 && n > 0 && k > 0 && (k & 15) == 0, "gemm_pk_h: bad dims (K %% 16 == 0)");
    W2E_REQUIRE(ldc >= n, "gemm_pk_h: ldc %d < n %d (rows of a slab would overlap)", ldc, n);
    W2E_REQUIRE(m_padded >= m && (m_padded & 31) == 0 && n_padded >= n && (n_padded & 63) == 0, "gemm_pk_h: m_padded %% 32 == 0, n_padded %% 64 == 0");
    W2E_REQUIRE((((uintptr_t)a_packed | (uintptr_t)b_packed_half) & 15) == 0, "gemm_pk_h: operands must be 16-byte aligned");
    W2E_REQUIRE((int64_t)(k >> 2) * m_padded * 16 < ((int64_t)1 << 32) - 64 && (int64_t)(k >> 3) * n_padded * 16 < ((int64_t)1 << 32) - 64, "gemm_pk_h: an operand exceeds 4 GB");
    if (m == 0) return 0;
    const int steps = k >> 4;
    W2E_REQUIRE(splits >= 1 && splits <= steps, "gemm_pk_h: %d splits of %d steps", splits, steps);
    GemmPkParams p{};
    p.a = a_packed, p.b = reinterpret_cast<const float*>(b_packed_half), p.c = c, p.m = m, p.n = n, p.k = k, p.mpad = m_padded, p.npad = n_padded, p.ldc = ldc;
    p.chunks_per = (int)(ceil_div(ceil_div(steps, splits), 4) * 4);
    W2E_REQUIRE((int64_t)(splits - 1) * p.chunks_per < steps, "gemm_pk_h: %d splits leave an empty slice of K = %d", splits, k);
    p.n_rb = (int)ceil_div(m, 32), p.n_cb = (int)ceil_div(n, 64), p.splits = splits;
    p.slab = (int64_t)m * ldc;
    const int64_t waves = (int64_t)p.n_rb * p.n_cb * splits;
    gemm_pk_h_kernel<<<(unsigned)ceil_div(waves, 4), 256, 0, (hipStream_t)stream>>>(p);
    W2E_LAUNCH_CHECK("gemm_pk_h");
    return 0;
}

int w2e_gemm_pk_h_splits(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 1;
    // one wave per SIMD as for the fp32 form; a 16-deep step costs this kernel its operand latency, not its MFMAs, so slices may be
    // shorter in MFMA terms -- the same "rounds x steps per slice + a per-slab term" model over steps
    const int64_t tiles = ceil_div(m, 32) * ceil_div(n, 64);
    const int steps = k >> 4;
    const int64_t simds = (int64_t)cu_count() * 4;
    int best = 1;
    double best_cost = 0.0;
    for (int sp = 1; sp <= 16 && sp <= steps / 4; ++sp) {
        const int per = (int)(ceil_div(ceil_div(steps, sp), 4) * 4);
        if ((int64_t)(sp - 1) * per >= steps) continue;
        const double cost = (double)ceil_div(tiles * sp, simds) * per + 2.5 * sp;
        if (sp == 1 || cost < best_cost) best = sp, best_cost = cost;
    }
    return best;
}

}  // extern "C"
