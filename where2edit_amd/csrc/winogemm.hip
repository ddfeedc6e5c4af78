// K1g -- the Winograd-domain contraction of the F(4x4,3x3) form as an own fp32-MFMA kernel with the OUTPUT TRANSFORM IN ITS EPILOGUE
// (gfx950).  Replaces the strided-batched vendor GEMM + w2e_wino_output pass of rounds 3's K1w for the wide same-resolution layers
// (model.py:270-274 in the shared-weight form of K1; models/facial_recognition/helpers.py:97-119 for the IR-SE50 / e4e stride-1 convs):
//
//     y[b,o] = epilogue(out_scale[b,o] * A^T [ sum_k U[xi][o][k] * V[xi][k][t] ] A)        xi = 0..35, t = (b, tile row, tile column)
//
// Two kernels:
//   wino4_pack_input_kernel   x [B,K,H,W] (* in_scale) -> V in the B-OPERAND order of the MFMA, vf[36][K/8][2][TP][4]:
//                             (xi, kc, h, t, c) = V[xi][k = 8*kc + 2*c + h][t]; TP = tiles padded to 32.  HBM-bound: reads x once, writes 2.25x.
//   wino4_gemm_kernel         one workgroup (8 waves, one per CU) = 64 output channels x 32 tiles x ALL 36 positions: wave (g, hh) holds the
//                             9 positions 9g..9g+8 of channel half hh as 9 accumulators of 32 x 32 (144 registers).  Both operands go
//                             global/L2 -> REGISTERS as the packed float4 that feeds four v_mfma_f32_32x32x2_f32 (the weights are packed
//                             the same way by w2e_wino_weights_fused): no LDS, no barrier and no cross-wave dependency in the K loop; every
//                             operand register is re-loaded for the next 8-channel chunk right behind the four MFMAs that consumed it, so a
//                             load has a whole chunk (36 MFMAs of this wave + 36 of its SIMD partner, ~4.6 k cycles) to land.  Then the
//                             products go through LDS in four rounds of 16 channels (double-buffered: one barrier per round) to the threads
//                             that own an (output channel, tile) pair: A^T . A, out_scale, the epilogues of w2e_modconv3x3 / w2e_conv3x3,
//                             four 16-byte row stores -- M (2.25x the output) never reaches HBM.
// Operand traffic: per 8-channel chunk a workgroup reads 36 x (64 + 32) float4-rows = 110 KB for 36 x 64 x 32 x 8 MACs: 24 B/clk/CU at the full
// MFMA rate, served by the XCD's L2 -- the workgroup -> (channel block, tile block) map below keeps the workgroups of one XCD on the SAME
// few tile blocks (all channel blocks of them run together), so V is fetched once per XCD and U streams from the Infinity Cache.
// Small layers (too few workgroups for 256 CUs) split K (the split index is part of the workgroup id): each split writes its raw A^T M A into a slab and
// wino_finish_kernel sums the slabs in a fixed order and applies the epilogue.  No atomics anywhere: bit-reproducible.
#include "common.h"
#include "wino_common.h"

namespace w2e {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------- input transform
// blockDim 256: lane pair (t, cp): tile t, channel pair cp of the float4 -> channels k = 8*kc + h + 2*(2*cp + e), e = 0, 1.
// grid (ceil(TP / 128), 2 * K/8).  A wave stores 32 tiles x 16 B = 512 contiguous bytes per position.
// RAGGED: H or W is not a multiple of 4 (IR-SE50's 14^2 / 7^2 stages): tiles = ceil(H/4) x ceil(W/4), the windows of the last tile row /
// column run past the image and read zeros, rows are not 16-byte aligned: scalar loads (these tensors are small).
template <bool RAGGED>
__global__ __launch_bounds__(256) void wino4_pack_input_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                               float* __restrict__ vf, int B, int K, int H, int W, int TP) {
    const int TX = (W + 3) >> 2, TY = (H + 3) >> 2, tiles = TX * TY;
    const int T = B * tiles;
    const int cp = threadIdx.x & 1;
    const int t = blockIdx.x * 128 + (threadIdx.x >> 1);
    if (t >= T) return;
    const int kc = blockIdx.y >> 1, h = blockIdx.y & 1, KC = K >> 3;
    const int b = t / tiles, tile = t - b * tiles;
    const int ty = tile / TX, tx = tile - ty * TX;
    float o2[36][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k = 8 * kc + h + 2 * (2 * cp + e);
        const int64_t plane = (int64_t)b * K + k;
        const float* xp = x + plane * H * W;
        const float sc = in_scale ? in_scale[plane] : 1.f;
        float tr[6][6];  // rows transformed first: tr[r][.] = B^T (row r of the window), then the columns
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = 4 * ty - 1 + r;
            const bool rin = iy >= 0 && iy < H;
            const float* row = xp + (int64_t)(rin ? iy : 0) * W + 4 * tx;
            float d[6];
            if (RAGGED) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int ix = 4 * tx - 1 + c;
                    d[c] = (rin && ix >= 0 && ix < W) ? row[c - 1] : 0.f;
                }
            } else {
                const float4 mid = rin ? *reinterpret_cast<const float4*>(row) : make_float4(0.f, 0.f, 0.f, 0.f);
                d[0] = (rin && tx > 0) ? row[-1] : 0.f, d[1] = mid.x, d[2] = mid.y, d[3] = mid.z, d[4] = mid.w;
                d[5] = (rin && tx + 1 < TX) ? row[4] : 0.f;
            }
            wino4_bt(d, tr[r]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {tr[0][j], tr[1][j], tr[2][j], tr[3][j], tr[4][j], tr[5][j]};
            float o[6];
            wino4_bt(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) o2[i * 6 + j][e] = sc * o[i];
        }
    }
    float* vp = vf + (((int64_t)kc * 2 + h) * TP + t) * 4 + 2 * cp;
    const int64_t xi_stride = (int64_t)KC * 2 * TP * 4;
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) *reinterpret_cast<float2*>(vp + xi * xi_stride) = make_float2(o2[xi][0], o2[xi][1]);
}

// ---------------------------------------------------------------------------------------------------------------- contraction
struct WinoGemmArgs {
    const float* uf;         // [36][KC][2][N][4]
    const float* vf;         // [36][KC][2][TP][4]
    const float* out_scale;  // [B,N] or NULL
    float* y;                // [B,N,H,W]  (PART: the slabs [S][B,N,H,W])
    const float* noise;      // [H,W] or NULL (ACT 1)
    const float* noise_w;
    const float* bias;       // [N] or NULL
    const float* slope;      // [N] or NULL (ACT 2)
    const float* dot_with;   // [B,N,H,W] (DOT)
    float* dot_part;         // [B,N,nseg] (DOT): one partial per (plane, 32-tile segment of the image), summed by wino_dot_sum_kernel
    int B, K, N, H, W, KC, TP, T;
    int ncb, ntb;            // channel blocks of 64, tile blocks of 32
    int kcs, splits;         // chunks per K split; number of splits
    int skip;                // (-DW2E_TUNING builds only: bit 0 no A re-loads, bit 1 no B re-loads, bit 2 no MFMAs, bit 3 no epilogue)
    int nseg, seg;           // DOT: segments per image (max(1, tiles/32)); lanes of a half-wave that share an image (min(32, tiles))
};

// ACT: 0 none; 1 = + noise_w*noise + bias, LeakyReLU(0.2) * sqrt 2; 2 = + bias, PReLU(slope) -- as wino4_output_kernel of round 3.
// PART: K split -- the raw A^T M A of this split goes to slab z, no scale / epilogue (wino_finish_kernel applies them).
// RAGGED: H or W not a multiple of 4 -- the tiles of the last row / column hang over the image and their outputs past it are not stored
// (scalar stores: the rows are not 16-byte aligned); no fused dot in this form.
template <int ACT, bool DOT, bool PART, bool RAGGED = false>
__global__ __launch_bounds__(512) void wino4_gemm_kernel(const WinoGemmArgs p) {
    static_assert(!(RAGGED && DOT), "the ragged form has no fused dot");
    constexpr int MS = 36 * 16 * 32;  // floats of one exchange buffer M[36][16][32]
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // M[2][MS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 3, hh = wave >> 2;
    const int half = lane >> 5, j = lane & 31;
    // workgroup -> (channel block cb, tile block tb, K split z): workgroups are dealt round-robin over the 8 XCDs (id % 8 share one).  A
    // "group" = the ncb channel blocks of one (tile block, split): its members read the same V rows, so they are given to ONE XCD, next to
    // each other in its local order, and the groups go round the XCDs -- every XCD gets the same number of groups (+-1) whatever ntb is
    // (round 4's first cut dealt TILE BLOCKS to XCDs: with ntb = 4 half of the chip idled).  The ~32 resident workgroups of an XCD then
    // share 32 / ncb groups of V (fetched once into its L2) and every 64-channel slice of U is read by 32 / ncb of them.
    const int L = (int)blockIdx.x, xcd = L & 7, loc = L >> 3;
    const int cb = loc % p.ncb, gi = (loc / p.ncb) * 8 + xcd;
    if (gi >= p.ntb * p.splits) return;  // (uniform: the whole workgroup)
    const int tb = gi / p.splits, z = gi - tb * p.splits;
#ifdef W2E_TUNING
    const int skip = p.skip;
#else
    constexpr int skip = 0;
#endif
    const int n0 = cb * 64, t0 = tb * 32;
    const int N = p.N, TP = p.TP, KC = p.KC;
    const int kc0 = z * p.kcs;
    const int kc1 = kc0 + p.kcs < KC ? kc0 + p.kcs : KC;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.uf), (short)0, (int)(unsigned)((int64_t)36 * KC * 2 * N * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.vf), (short)0, (int)(unsigned)((int64_t)36 * KC * 2 * TP * 16), 0x00020000);
    const int voff_a = (half * N + n0 + 32 * hh + j) * 16;
    const int voff_b = (half * TP + t0 + j) * 16;
    const unsigned stride_a = (unsigned)(2 * N) * 16u, stride_b = (unsigned)(2 * TP) * 16u;  // bytes per 8-channel chunk
    auto ld_a = [&](int q, int kc) __attribute__((always_inline)) {
        return __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, voff_a, (int)((unsigned)((9 * g + q) * KC + kc) * stride_a), 0));
    };
    auto ld_b = [&](int q, int kc) __attribute__((always_inline)) {
        return __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff_b, (int)((unsigned)((9 * g + q) * KC + kc) * stride_b), 0));
    };
    wg_f32x16 acc[9];
    wg_f32x4 a[9], bq[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {  // (in the K loop's order, pinned: the waits at the loop top are then counted on both of its entries)
        a[q] = ld_a(q, kc0), bq[q] = ld_b(q, kc0);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int kc = kc0; kc < kc1; ++kc) {
        const int nk = kc + 1 < kc1 ? kc + 1 : kc;  // (the last chunk re-loads itself: no branch around the loads, the waits stay counted)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (!(skip & 4)) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], bq[q][0], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], bq[q][1], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], bq[q][2], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], bq[q][3], acc[q], 0, 0, 0);
            }
            if (!(skip & 1)) a[q] = ld_a(q, nk);
            if (!(skip & 2)) bq[q] = ld_b(q, nk);
            // (pinned: left to itself the scheduler hoists all 36 MFMAs of the chunk in front of all 18 loads, and the loop top then
            // waits vmcnt(0) on loads issued a moment ago; kept in source order the waits are counted -- the two oldest of 18)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: 4 rounds of 8 channels per channel half = 16 channels x 32 tiles = one (channel, tile) item per thread.
    // Accumulator register r of lane (half, j) is row (r & 3) + 8 * (r >> 2) + 4 * half, column j: round q4 moves registers 4*q4 .. 4*q4+3
    // = rows 8*q4 .. 8*q4+7 of each wave's 32-channel half into M[xi][hh*8 + 4*half + rr][j].
    const int oj = tid & 31, on16 = tid >> 5;  // item: tile oj of the block, row on16 of the round (channel half on16 >> 3, row on16 & 7)
    const int H = p.H, W = p.W;
    const int TX = (W + 3) >> 2, tiles = TX * ((H + 3) >> 2);
    const int t = t0 + oj;
    const bool live = t < p.T;
    const int tc = live ? t : p.T - 1;
    const int b = tc / tiles, tile = tc - b * tiles;
    const int ty = tile / TX, tx = tile - ty * TX;
    const int64_t pix = (int64_t)(4 * ty) * W + 4 * tx;
    const float nw = (ACT == 1 && p.noise) ? p.noise_w[0] : 0.f;
    float4 nz[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
        nz[ii] = (ACT == 1 && p.noise && !RAGGED) ? *reinterpret_cast<const float4*>(p.noise + pix + (int64_t)ii * W) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* const ybase = PART ? p.y + (int64_t)z * p.B * N * H * W : p.y;
    if (skip & 8) return;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {  // (unrolled: the accumulator registers are indexed by q4)
        float* const mb = wsm + (q4 & 1) * MS;
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) mb[((9 * g + q) * 16 + hh * 8 + 4 * half + rr) * 32 + j] = acc[q][4 * q4 + rr];
        const int n = n0 + 32 * (on16 >> 3) + 8 * q4 + (on16 & 7);
        const int64_t plane = (int64_t)b * N + n;
        // the item's global operands are requested before the barrier: they land while the products are exchanged
        const float os = (!PART && p.out_scale) ? p.out_scale[plane] : 1.f;
        const float bs = (ACT && p.bias) ? p.bias[n] : 0.f;
        const float sl = (ACT == 2 && p.slope) ? p.slope[n] : 1.f;
        float4 dw[4];
        if (DOT) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) dw[ii] = *reinterpret_cast<const float4*>(p.dot_with + plane * H * W + pix + (int64_t)ii * W);
        }
        __syncthreads();
        const float* mp = mb + on16 * 32 + oj;
        float s[4][6];  // columns transformed first: s[.][jj] = A^T (column jj of the 6x6 products)
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
        float part = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float o[4];
            wino4_at(s[ii], o);
            if (DOT) part += (o[0] * dw[ii].x + o[1] * dw[ii].y) + (o[2] * dw[ii].z + o[3] * dw[ii].w);
            float4 r = make_float4(o[0] * os, o[1] * os, o[2] * os, o[3] * os);
            if (ACT == 1) {
                r.x += nw * nz[ii].x + bs, r.y += nw * nz[ii].y + bs, r.z += nw * nz[ii].z + bs, r.w += nw * nz[ii].w + bs;
                r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
                r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
            }
            if (ACT == 2) {
                r.x += bs, r.y += bs, r.z += bs, r.w += bs;
                r.x = r.x > 0.f ? r.x : sl * r.x, r.y = r.y > 0.f ? r.y : sl * r.y;
                r.z = r.z > 0.f ? r.z : sl * r.z, r.w = r.w > 0.f ? r.w : sl * r.w;
            }
            if (RAGGED) {
                float* const yr = ybase + plane * H * W + pix + (int64_t)ii * W;
                const bool rok = live && 4 * ty + ii < H;
                if (rok && 4 * tx + 0 < W) yr[0] = r.x;
                if (rok && 4 * tx + 1 < W) yr[1] = r.y;
                if (rok && 4 * tx + 2 < W) yr[2] = r.z;
                if (rok && 4 * tx + 3 < W) yr[3] = r.w;
            } else if (live) {
                *reinterpret_cast<float4*>(ybase + plane * H * W + pix + (int64_t)ii * W) = r;
            }
        }
        if (DOT) {  // dot_part[plane][segment] = sum over the lanes of this half-wave that share (b, n): fixed order, no atomics
            if (!live) part = 0.f;
            for (int off = p.seg >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (live && (oj & (p.seg - 1)) == 0) p.dot_part[plane * p.nseg + (tile >> 5)] = part;
        }
        // (no second barrier: round q4+1 writes the OTHER buffer, which was last read in round q4-1, i.e. before this round's barrier)
    }
}

// Split-K: y = epilogue(out_scale * sum_z slab[z]); one WAVE per (b, n) plane (the layers that split are <= 64^2), so the fused dot is a
// wave reduction with a single writer: dot_out[plane] += sum_p conv_unscaled * dot_with, in a fixed order.
template <int ACT, bool DOT>
__global__ __launch_bounds__(256) void wino_finish_kernel(const float* __restrict__ slabs, int S, const float* __restrict__ out_scale,
                                                          float* __restrict__ y, int planes, int N, int HW,
                                                          const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          const float* __restrict__ dot_with, float* __restrict__ dot_out) {
    const int plane = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const int n = plane % N;
    const float os = out_scale ? out_scale[plane] : 1.f;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
    const float bs = (ACT && bias) ? bias[n] : 0.f;
    const float sl = (ACT == 2 && slope) ? slope[n] : 1.f;
    const int64_t slab = (int64_t)planes * HW, base = (int64_t)plane * HW;
    float part = 0.f;
    if (HW & 3) {  // (planes that are not whole float4s: IR-SE50's 7^2 stage.  No noise / fused dot there: act 0 or 2)
        for (int i = lane; i < HW; i += 64) {
            float c = slabs[base + i];
            for (int z = 1; z < S; ++z) c += slabs[(int64_t)z * slab + base + i];
            float r = c * os;
            if (ACT == 2) {
                r += bs;
                r = r > 0.f ? r : sl * r;
            }
            y[base + i] = r;
        }
        return;
    }
    for (int i = lane * 4; i < HW; i += 256) {
        float4 c = *reinterpret_cast<const float4*>(slabs + base + i);
        for (int z = 1; z < S; ++z) {
            const float4 d = *reinterpret_cast<const float4*>(slabs + (int64_t)z * slab + base + i);
            c.x += d.x, c.y += d.y, c.z += d.z, c.w += d.w;
        }
        if (DOT) {
            const float4 d = *reinterpret_cast<const float4*>(dot_with + base + i);
            part += (c.x * d.x + c.y * d.y) + (c.z * d.z + c.w * d.w);
        }
        float4 r = make_float4(c.x * os, c.y * os, c.z * os, c.w * os);
        if (ACT == 1) {
            const float4 nzv = noise ? *reinterpret_cast<const float4*>(noise + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            r.x += nw * nzv.x + bs, r.y += nw * nzv.y + bs, r.z += nw * nzv.z + bs, r.w += nw * nzv.w + bs;
            r.x = fmaxf(r.x, 0.2f * r.x) * 1.4142135623730951f, r.y = fmaxf(r.y, 0.2f * r.y) * 1.4142135623730951f;
            r.z = fmaxf(r.z, 0.2f * r.z) * 1.4142135623730951f, r.w = fmaxf(r.w, 0.2f * r.w) * 1.4142135623730951f;
        }
        if (ACT == 2) {
            r.x += bs, r.y += bs, r.z += bs, r.w += bs;
            r.x = r.x > 0.f ? r.x : sl * r.x, r.y = r.y > 0.f ? r.y : sl * r.y;
            r.z = r.z > 0.f ? r.z : sl * r.z, r.w = r.w > 0.f ? r.w : sl * r.w;
        }
        *reinterpret_cast<float4*>(y + base + i) = r;
    }
    if (DOT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) dot_out[plane] += part;
    }
}

// dot_out[plane] += sum_s part[plane][s]  (s ascending: fixed order)
__global__ __launch_bounds__(256) void wino_dot_sum_kernel(const float* __restrict__ part, float* __restrict__ dot_out, int planes, int nseg) {
    const int plane = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (plane >= planes) return;
    float s = 0.f;
    for (int i = 0; i < nseg; ++i) s += part[(int64_t)plane * nseg + i];
    dot_out[plane] += s;
}

static int plan(int batch, int k_ch, int n_ch, int h, int w, int* tp, int* splits, int64_t* ws_floats, int* nseg_out, int force_splits) {
    const int tiles = ((h + 3) / 4) * ((w + 3) / 4);
    const int64_t T = (int64_t)batch * tiles;
    const int64_t TP = (T + 31) & ~(int64_t)31;
    const int KC = k_ch / 8;
    const int64_t G = (int64_t)(n_ch / 64) * (TP / 32);
    int S = 1;
    if (force_splits > 0) S = force_splits;
    else if (G < 192 && (int64_t)h * w <= 64 * 64) {  // too few workgroups for 256 CUs: split K (slabs of <= 64^2 planes, summed per wave)
        const int cus = cu_count();
        S = (int)((cus + G - 1) / G);
        if (S > KC / 4) S = KC / 4;  // at least 4 chunks per split
        if (S > 16) S = 16;
        if (S < 1) S = 1;
    }
    int kcs = (KC + S - 1) / S;
    S = (KC + kcs - 1) / kcs;
    const int nseg = tiles >= 32 ? tiles / 32 : 1;
    *tp = (int)TP, *splits = S, *nseg_out = nseg;
    *ws_floats = (int64_t)batch * n_ch * nseg + (S > 1 ? (int64_t)S * batch * n_ch * h * w : 0);
    return kcs;
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_wino_gemm_plan(int batch, int k_ch, int n_ch, int h, int w, int* tiles_padded, int* splits, int64_t* workspace_floats) {
    W2E_REQUIRE(tiles_padded && splits && workspace_floats, "wino_gemm_plan: null output");
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && (k_ch & 7) == 0 && n_ch > 0 && (n_ch & 63) == 0, "wino_gemm_plan: K %% 8 == 0, N %% 64 == 0 (got %d, %d)", k_ch, n_ch);
    W2E_REQUIRE(h >= 4 && w >= 4, "wino_gemm_plan: H, W >= 4 (got %d x %d)", h, w);
    W2E_REQUIRE((int64_t)batch * ((h + 3) / 4) * ((w + 3) / 4) < ((int64_t)1 << 30), "wino_gemm_plan: too many tiles");
    int nseg = 0;
    plan(batch, k_ch, n_ch, h, w, tiles_padded, splits, workspace_floats, &nseg, 0);
    return 0;
}

int w2e_wino_pack_input(const float* x, const float* in_scale, float* vf, int batch, int k_ch, int h, int w, int tiles_padded, void* stream) {
    W2E_REQUIRE(x && vf, "wino_pack_input: null tensor");
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && (k_ch & 7) == 0 && h >= 4 && w >= 4, "wino_pack_input: K %% 8 == 0 and H, W >= 4 (got K %d, %d x %d)", k_ch, h, w);
    const bool ragged = (h & 3) != 0 || (w & 3) != 0;
    W2E_REQUIRE((((uintptr_t)x | (uintptr_t)vf) & 15) == 0, "wino_pack_input: x / vf must be 16-byte aligned");
    const int64_t T = (int64_t)batch * ((h + 3) / 4) * ((w + 3) / 4);
    W2E_REQUIRE(tiles_padded >= T && (tiles_padded & 31) == 0, "wino_pack_input: tiles_padded %d for %lld tiles (w2e_wino_gemm_plan)", tiles_padded, (long long)T);
    W2E_REQUIRE((int64_t)36 * k_ch * tiles_padded * 4 < ((int64_t)1 << 32) - 64, "wino_pack_input: V exceeds 4 GB");
    if (batch == 0) return 0;
    dim3 grid((unsigned)ceil_div(T, 128), (unsigned)(k_ch / 8 * 2));
    if (ragged) wino4_pack_input_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(x, in_scale, vf, batch, k_ch, h, w, tiles_padded);
    else wino4_pack_input_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, in_scale, vf, batch, k_ch, h, w, tiles_padded);
    W2E_LAUNCH_CHECK("wino_pack_input");
    return 0;
}

int w2e_wino_gemm(const float* uf, const float* vf, const float* out_scale, float* y, int batch, int k_ch, int n_ch, int h, int w,
                  int tiles_padded, int splits, float* workspace, int act, const float* noise, const float* noise_w, const float* bias,
                  const float* slope, const float* dot_with, float* dot_out, void* stream) {
    W2E_REQUIRE(uf && vf && y, "wino_gemm: null tensor");
    W2E_REQUIRE(act >= 0 && act <= 2, "wino_gemm: epilogue %d (0 none, 1 StyledConv, 2 bias + PReLU)", act);
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && (k_ch & 7) == 0 && n_ch > 0 && (n_ch & 63) == 0, "wino_gemm: K %% 8 == 0, N %% 64 == 0 (got %d, %d)", k_ch, n_ch);
    W2E_REQUIRE(h >= 4 && w >= 4, "wino_gemm: H, W >= 4 (got %d x %d)", h, w);
    const bool ragged = (h & 3) != 0 || (w & 3) != 0;  // tiles = ceil(H/4) x ceil(W/4); outputs past the image are not stored
    W2E_REQUIRE(!ragged || (!dot_with && act != 1), "wino_gemm: H, W that are not multiples of 4 take the plain and the bias + PReLU epilogues only");
    W2E_REQUIRE(!(act && dot_with), "wino_gemm: the activation epilogues and the fused dot exclude each other");
    W2E_REQUIRE(!dot_with || dot_out, "wino_gemm: dot_with without dot_out");
    W2E_REQUIRE(!noise || noise_w, "wino_gemm: noise without noise_w");
    W2E_REQUIRE(act == 1 || !noise, "wino_gemm: noise belongs to epilogue 1");
    W2E_REQUIRE(act == 2 || !slope, "wino_gemm: slope belongs to epilogue 2");
    W2E_REQUIRE((((uintptr_t)uf | (uintptr_t)vf | (uintptr_t)y | (uintptr_t)(dot_with ? dot_with : y) | (uintptr_t)(noise ? noise : y) |
                  (uintptr_t)(workspace ? workspace : y)) & 15) == 0, "wino_gemm: uf / vf / y / dot_with / noise / workspace must be 16-byte aligned");
    if (batch == 0) return 0;
    const int tiles = ((h + 3) / 4) * ((w + 3) / 4);
    const int64_t T = (int64_t)batch * tiles;
    W2E_REQUIRE(T < ((int64_t)1 << 30), "wino_gemm: too many tiles");
    W2E_REQUIRE(tiles_padded >= T && (tiles_padded & 31) == 0, "wino_gemm: tiles_padded %d for %lld tiles (w2e_wino_gemm_plan)", tiles_padded, (long long)T);
    const int KC = k_ch / 8;
    W2E_REQUIRE((int64_t)36 * k_ch * tiles_padded * 4 < ((int64_t)1 << 32) - 64 && (int64_t)36 * k_ch * n_ch * 4 < ((int64_t)1 << 32) - 64, "wino_gemm: U or V exceeds 4 GB");
    W2E_REQUIRE(splits >= 1 && splits <= KC, "wino_gemm: %d splits of %d chunks", splits, KC);
    // the fused dot reduces over the lanes of a half-wave that share an image: whole 32-tile segments, or a power of two below
    W2E_REQUIRE(!dot_with || (tiles & 31) == 0 || (tiles < 32 && (tiles & (tiles - 1)) == 0), "wino_gemm: fused dot with %d tiles per plane (a multiple of 32, or a power of two below it)", tiles);
    const int nseg = tiles >= 32 ? tiles / 32 : 1;
    W2E_REQUIRE(!(dot_with || splits > 1) || workspace, "wino_gemm: the fused dot and a split K need the workspace of w2e_wino_gemm_plan");
    W2E_REQUIRE(splits == 1 || (int64_t)h * w % 4 == 0 || (!dot_with && act != 1), "wino_gemm: split planes that are not whole float4s take the plain and the bias + PReLU epilogues only");
    hipStream_t s = (hipStream_t)stream;
    WinoGemmArgs a;
    a.uf = uf, a.vf = vf, a.out_scale = out_scale, a.noise = noise, a.noise_w = noise_w, a.bias = bias, a.slope = slope, a.dot_with = dot_with;
    a.B = batch, a.K = k_ch, a.N = n_ch, a.H = h, a.W = w, a.KC = KC, a.TP = tiles_padded, a.T = (int)T;
    a.ncb = n_ch / 64, a.ntb = tiles_padded / 32;
    a.kcs = (KC + splits - 1) / splits, a.splits = splits;
    a.skip = options().tune_skip;
    W2E_REQUIRE((int64_t)(splits - 1) * a.kcs < KC, "wino_gemm: %d splits of %d chunks leave an empty split (w2e_wino_gemm_plan)", splits, KC);
    a.nseg = nseg, a.seg = tiles < 32 ? tiles : 32;
    float* const dot_part = workspace;                                            // [B*N*nseg]
    float* const slabs = workspace ? workspace + (int64_t)batch * n_ch * nseg : nullptr;  // [S][B,N,H,W]
    a.dot_part = dot_part;
    a.y = splits > 1 ? slabs : y;
    const int64_t gx = (int64_t)8 * a.ncb * (((int64_t)a.ntb * splits + 7) / 8);
    W2E_REQUIRE(gx < ((int64_t)1 << 31), "wino_gemm: grid too large");
    const dim3 grid((unsigned)gx);
    const size_t lds = (size_t)2 * 36 * 16 * 32 * 4;
    static unsigned done[8];
#define W2E_WG(ACTv, DOTv, PARTv, RAGv, slot)                                                                                         \
    do {                                                                                                                             \
        W2E_REQUIRE(big_lds_once((const void*)wino4_gemm_kernel<ACTv, DOTv, PARTv, RAGv>, &done[slot]), "wino_gemm: cannot enable %zu B of LDS", lds); \
        wino4_gemm_kernel<ACTv, DOTv, PARTv, RAGv><<<grid, 512, lds, s>>>(a);                                                         \
    } while (0)
    if (ragged) {
        if (splits > 1) W2E_WG(0, false, true, true, 5);
        else if (act == 2) W2E_WG(2, false, false, true, 6);
        else W2E_WG(0, false, false, true, 7);
    } else if (splits > 1) W2E_WG(0, false, true, false, 4);
    else if (act == 1) W2E_WG(1, false, false, false, 0);
    else if (act == 2) W2E_WG(2, false, false, false, 1);
    else if (dot_with) W2E_WG(0, true, false, false, 2);
    else W2E_WG(0, false, false, false, 3);
#undef W2E_WG
    W2E_LAUNCH_CHECK("wino_gemm");
    const int planes = batch * n_ch;
    if (splits > 1) {
        const unsigned fg = (unsigned)ceil_div(planes, 4);
        if (act == 1) wino_finish_kernel<1, false><<<fg, 256, 0, s>>>(slabs, splits, out_scale, y, planes, n_ch, h * w, noise, noise_w, bias, nullptr, nullptr, nullptr);
        else if (act == 2) wino_finish_kernel<2, false><<<fg, 256, 0, s>>>(slabs, splits, out_scale, y, planes, n_ch, h * w, nullptr, nullptr, bias, slope, nullptr, nullptr);
        else if (dot_with) wino_finish_kernel<0, true><<<fg, 256, 0, s>>>(slabs, splits, out_scale, y, planes, n_ch, h * w, nullptr, nullptr, nullptr, nullptr, dot_with, dot_out);
        else wino_finish_kernel<0, false><<<fg, 256, 0, s>>>(slabs, splits, out_scale, y, planes, n_ch, h * w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        W2E_LAUNCH_CHECK("wino_gemm (finish)");
    } else if (dot_with) {
        wino_dot_sum_kernel<<<(unsigned)ceil_div(planes, 256), 256, 0, s>>>(dot_part, dot_out, planes, nseg);
        W2E_LAUNCH_CHECK("wino_gemm (dot sum)");
    }
    return 0;
}

}  // extern "C"
