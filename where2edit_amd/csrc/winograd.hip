// The FUSED Winograd F(4x4,3x3) form of the same-resolution modulated 3x3 convolution (K1w, include/w2e.h) for gfx950.
//
//   y[b,o] = out_scale[b,o] * conv3x3(W, in_scale[b,i] * x[b,i])          (model.py:270-274 in the shared-weight form of K1)
//          = out_scale * A^T [ sum_i (G W[o,i] G^T) (.) (B^T (in_scale * d[b,i]) B) ] A     per 4x4 output tile, d = its 6x6 input window
//
// 36 multiplications per 4x4 outputs instead of 144: the contraction over input channels becomes 36 independent [N x K] x [K x tiles]
// products with 4x fewer FLOPs than the direct form.  This file: the weight transform (shared with winogemm.hip) and the fused kernel
// for the NARROW high-resolution layers (32 @ 1024^2, 64 @ 512^2, 128 @ 256^2 and the encoders' 64- / 128-channel stages), whose
// transform-domain tensors never leave the CU.  The wide layers run the same form as two kernels (winogemm.hip: packed input
// transform + an own MFMA contraction with the output transform in its epilogue).  Round 3's F(2x2,3x3) form, its separate output
// pass and the first two generations of the fused kernel are gone (profiles/r03_winograd.txt keeps their measurements).
// fp32 throughout; the transforms multiply by up to 8 and 1/24: ~1e-5 max-norm relative to a float64 convolution at K = 128 ... 512
// (direct form 3e-7) -- two decades inside the path's 1e-3 tolerance (BASELINE north_star), one inside the tests' 1e-4.
#include <type_traits>

#include "common.h"
#include "wino_common.h"
#include "../../include/w2e.h"

namespace w2e {

// Interpolation points 0, +-1, +-2, inf (Lavin & Gray): G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
// wp [ceil(K/8)][9][2][N][4]: element (kc, tap, h, n, c) = W(k = 8*kc + 2*c + h, tap, n)  ->
// uf[36][K/8][2][N][4]: (xi, kc, h, n, c) = (G W G^T)[xi][n][8*kc + 2*c + h] -- the A-operand order of the MFMA kernels: the float4 of
// (xi, kc, h, n) feeds four v_mfma_f32_32x32x2_f32
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
            u[((((int64_t)(i * 6 + j) * (K >> 3) + kc) * 2 + h) * N + n) * 4 + c] = (float)o[j];
        }
    }
}

typedef float wf_f32x16 __attribute__((ext_vector_type(16)));
typedef float wf_f32x4 __attribute__((ext_vector_type(4)));
typedef float wf_f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ F(4x4, 3x3), fused, persistent, patch staged by LDS-DMA
// One persistent workgroup = 4 matrix waves + 4 transform waves in lock-step ticks (one barrier each; two separate loops with the same
// barrier sequence, so each role has its own register allocation), looping over (spatial block of 32 tiles, 32-channel output block)
// pairs: wave g of the matrix half owns the 9 transform positions 9g .. 9g+8 as 9 accumulators of 32 channels x 32 tiles; its A
// operands are the packed transformed weights (global, L2-resident; fetched one tick ahead, in place), its B operands the V image the
// transform waves write to LDS.  The transform waves stage the RAW 8-channel patch of a block --
// [8 ch][18 rows][10 quads] floats, image columns bx*32-4 .. bx*32+35 so that every 16-byte quad is either wholly inside the image
// row or wholly outside it (zero padding = an out-of-range offset, exact) -- by `buffer_load_dwordx4 ... lds` into a ring of three
// stages, three chunks ahead of the MFMAs, with explicit vmcnt counts: no registers, no compiler-managed waits, 6 DMA instructions per
// wave and chunk instead of 18 window loads per thread.  A transform thread owns one tile, a channel PAIR and half of the 36 positions,
// in packed fp32 (v_pk_*_f32: see `transform_h`).  Per block: a pre-tick (chunk 0 -> V[0]), KC matrix ticks (MFMAs of chunk c | transform of chunk c+1,
// DMA of chunk g+3), then two output rounds of 16 channels through the V space, in which all 512 threads own a (channel, tile) pair.
// TXN: tiles of a block along x (8: blocks of 32 x 16 pixels; 16: 64 x 8 -- longer row segments per DMA, fewer cache lines per byte).
// MW: matrix waves of a workgroup.  4 (512 threads): 32 output channels per workgroup, everything above.  8 (768 threads, three waves per
// SIMD, <= 168 VGPRs): 64 output channels -- wave (g, hh) owns positions 9g .. 9g+8 of channel half hh, so ONE input transform feeds twice
// the MFMAs.  Why that is the lever (profiles/r05_issue_share_probe.txt): while an fp32 MFMA executes, NO VALU instruction of any other
// wave on that SIMD issues (a partner gets one instruction per MFMA, whatever its kind or priority; with s_nop gaps behind the MFMAs it
// gets exactly the gaps) -- the fp32 matrix rate equals the packed-fp32 vector rate because it IS the vector ALUs.  A tick therefore
// costs MFMA cycles PLUS transform cycles (2304 + ~900 of the 4200 measured), never their maximum, and the only way to make the
// transform cheaper per MFMA is to share it between more output channels.  With 8 matrix waves the accumulators take 144 of a wave's
// 168 registers: the A operands come through a ring of three positions instead of a tick ahead, and the output items (which need ~60
// registers beside live accumulators) are run, in four rounds of 16 channels, by the transform waves and the matrix waves of channel half
// 0 -- whose accumulators are half dead in round 0 and dead from round 1 on; half 1's stay live until round 3 (`matrix_role`).
template <int ACT, bool DOT, int TXN, int MW>
__global__ __launch_bounds__(64 * (MW + 4), 1) void wino4_fused3_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ uf, const float* __restrict__ out_scale,
                                                              float* __restrict__ y, int B, int K, int N, int H, int W, int kc_log2,
                                                              int n_blocks, const float* __restrict__ noise,
                                                              const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                              const float* __restrict__ slope, const float* __restrict__ dot_with,
                                                              float* __restrict__ dot_partial, int skip, int xmap) {
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
    // LDS byte map (the launch passes the same sum as its dynamic size): V 0 .. 73 728, ring .. 147 456, tables .. 151 936.  The ring
    // starts ABOVE 64 KB: the M0 base of `buffer_load ... lds` must carry byte offsets up to 147 455, which gfx950 (160 KB of LDS per
    // CU) does and earlier parts with a 16-bit field do not -- an ISA change here has to fail the build, not the numbers.
    static_assert(2 * VS * 4 == 73728 && (2 * VS + 3 * PS) * 4 == 147456, "fused Winograd kernel: LDS map of the V stages and the DMA ring");
    static_assert((2 * VS + 3 * PS + 512 + 704) * 4 <= 160 * 1024, "fused Winograd kernel: more LDS than a gfx950 CU has");
    static_assert(MW == 4 || MW == 8, "4 matrix waves (32 output channels per workgroup) or 8 (64)");
    static_assert((2 * VS + 3 * PS) * 4 < (1 << 18), "fused Winograd kernel: DMA ring offsets beyond M0's 18-bit LDS address");
    static_assert(6 * 4 == 24 && 24 * 64 >= PQ, "fused Winograd kernel: 4 transform waves x 6 DMA instructions must cover a patch chunk");
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // V[2][VS], ring[3][PS], in_scale table [2][256]
    float* const mbuf = wsm;
    float* const ring = wsm + 2 * VS;
    float* const sctab = ring + 3 * PS;  // in_scale[b, .] of the current / the next block, written by the matrix waves: the transform
                                         // waves issue no compiler-managed global load at all (its wait would be vmcnt(0) and drain the
                                         // DMA ring) -- the epilogue operands come the same way:
    float* const etab = sctab + 512;     // the block's noise patch [16][32], then out_scale / bias / slope of the workgroup's 32 / 64 channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx_n = W / BWP, by_n = H / BHP, per_img = bx_n * by_n;
    const int KC = 1 << kc_log2;
    // Which blocks this workgroup loops over.  xmap = 0: block = blockIdx.x + i * gridDim.x -- at any moment the resident workgroups
    // cover gridDim.x CONSECUTIVE blocks, but consecutive workgroup ids are dealt round-robin to the 8 XCDs, so every horizontal
    // neighbour of a block is worked on by ANOTHER XCD: the 16-byte column halo on either side of a patch row lies in the neighbour's
    // 128-byte lines, which each XCD's own L2 then fetches through the fabric a second time (profiles/r05_fetch_calibration.txt:
    // a 288-byte row costs 4 lines of traffic).  xmap = 1: XCD j (= blockIdx.x & 7) owns the contiguous eighth [j n/8, (j+1) n/8)
    // of the launch's blocks and its gridDim.x / 8 workgroups sweep it front to back -- horizontal neighbours run at the same time
    // on the SAME XCD and share the halo lines in its L2; the row halo between consecutive sweeps of an XCD is re-read from L2 too.
    // (The host sets xmap only when n_blocks and gridDim.x are multiples of 8 and every workgroup gets at least one block.)
    const int xper = (int)gridDim.x >> 3;
    const int xlo = xmap ? (n_blocks >> 3) * ((int)blockIdx.x & 7) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int xstep = xmap ? xper : (int)gridDim.x;
    const int xend = xmap ? (n_blocks >> 3) * (((int)blockIdx.x & 7) + 1) : n_blocks;
    const int count = (xend - xlo + xstep - 1) / xstep;
    const int total = count << kc_log2;
    constexpr int NCH = 8 * MW;  // output channels of a workgroup
    const int n0 = blockIdx.y * NCH;
    // item threads of an output round (16 channels x 32 tiles = 512 items): MW 4: all 512 threads; MW 8: the transform waves (items 0 .. 255)
    // and the matrix waves of channel half 0 (256 .. 511), whose accumulators are dead or half dead by then -- half 1's are live until round 3
    const int item_t = MW == 4 ? tid : (tid >= 64 * MW ? tid - 64 * MW : 256 + (tid & 255));
    const int oj = item_t & 31, on16 = item_t >> 5;
    const float nw = (ACT == 1 && noise) ? noise_w[0] : 0.f;
    auto out_items = [&](int blk, int b, int64_t opix, int nloc) __attribute__((always_inline)) {  // nloc: channel within the block of 32
        const int n = n0 + nloc;
        const float os = etab[512 + nloc], bs = etab[512 + NCH + nloc], sl = etab[512 + 2 * NCH + nloc];
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
    if (wave >= MW) {
        // ---------------------------------------------------------------------------------------------- transform waves
        const int tw = wave - MW;
        const int tj = tid & 31;  // tile of a block
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
            const int blk = xlo + bi * xstep;
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
            const int blk = xlo + bi * xstep;
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
        // chunk (bi, kc) = global g: ring stage g % 3 -> V[kc & 1], in PACKED fp32 (v_pk_*_f32).  What a partner wave issues beside a
        // saturated fp32 matrix pipe is one instruction per ~22 cycles of ANY kind (tools/pk_probe.hip, profiles/r04_pk_probe.txt): the
        // transform's instruction COUNT is its cost.  A thread owns one tile, the channel PAIR (chA, chA + 2) of the chunk -- the
        // float2 (c0, c0 + 1) of V's operand quad -- and HALF of the 36 positions (columns jj = 3 hf .. 3 hf + 2 of the transform domain:
        // the row pass yields 3 of its 6 outputs from 5 of the 6 window columns; hf is wave-uniform and compile-time inside).  The
        // patch columns 4 .. 7 of each channel come as one conflict-free 16-byte read, v_pk_mov_b32 pairs the channels.
        const int hfw = tw & 1, cpair = (lane >> 5) + 2 * (tw >> 1);
        const int chA = 4 * (cpair >> 1) + (cpair & 1);  // = 2 c0 + h with h = cpair & 1, c0 = 2 (cpair >> 1); chB = chA + 2
        auto transform_h = [&](auto hf_c, int bi, int kc, int g) __attribute__((always_inline)) {
            constexpr int hf = decltype(hf_c)::value;
            const wf_f32x2 sc = wf_f32x2{sctab[(bi & 1) * 256 + kc * 8 + chA], sctab[(bi & 1) * 256 + kc * 8 + chA + 2]};
            const float* pp = ring + (g % 3) * PS + chA * 720 + (4 * (tj / TXN)) * PC + 4 * (tj % TXN);
            wf_f32x2 t[6][3];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const wf_f32x4 ma = *reinterpret_cast<const wf_f32x4*>(pp + r * PC + 4), mb = *reinterpret_cast<const wf_f32x4*>(pp + r * PC + 4 + 1440);
                const wf_f32x2 e = wf_f32x2{pp[r * PC + (hf ? 8 : 3)], pp[r * PC + (hf ? 8 : 3) + 1440]};
                const wf_f32x2 a01 = wf_f32x2{ma.x, ma.y}, a23 = wf_f32x2{ma.z, ma.w}, b01 = wf_f32x2{mb.x, mb.y}, b23 = wf_f32x2{mb.z, mb.w};
                wf_f32x2 m0, m1, m2, m3;  // (D.lo = S0[op_sel 0], D.hi = S1[op_sel 1])
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(m0) : "v"(a01), "v"(b01));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(m1) : "v"(a01), "v"(b01));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(m2) : "v"(a23), "v"(b23));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(m3) : "v"(a23), "v"(b23));
                if (hf == 0) {  // window columns d0 .. d4 = e, m0 .. m3 -> t0, t1, t2 of wino4_bt
                    const wf_f32x2 a = m3 - 4.f * m1, b = m2 - 4.f * m0;
                    t[r][0] = 4.f * e - 5.f * m1 + m3;
                    t[r][1] = a + b;
                    t[r][2] = a - b;
                } else {  // d1 .. d5 = m0 .. m3, e -> t3, t4, t5
                    const wf_f32x2 c = m3 - m1, f = m2 - m0;
                    t[r][0] = c + 2.f * f;
                    t[r][1] = c - 2.f * f;
                    t[r][2] = 4.f * m0 - 5.f * m2 + e;
                }
            }
            wf_f32x2* vp = reinterpret_cast<wf_f32x2*>(wsm + (kc & 1) * VS + ((cpair & 1) * 32 + tj) * 4 + 2 * (cpair >> 1)) + (3 * hf) * 128;
#pragma unroll
            for (int jl = 0; jl < 3; ++jl) {
                const wf_f32x2 col[6] = {t[0][jl], t[1][jl], t[2][jl], t[3][jl], t[4][jl], t[5][jl]};
                wf_f32x2 o[6];
                wino4_bt(col, o);
#pragma unroll
                for (int i6 = 0; i6 < 6; ++i6) vp[(i6 * 6 + jl) * 128] = sc * o[i6];
            }
        };
        auto transform = [&](int bi, int kc, int g) __attribute__((always_inline)) {
            if (hfw) transform_h(std::integral_constant<int, 1>{}, bi, kc, g);  // (wave-uniform)
            else transform_h(std::integral_constant<int, 0>{}, bi, kc, g);
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
            const int blk = xlo + i * xstep;
            int b;
            int64_t opix;
            block_pix(blk, b, opix);
            // (with the fused dot an item loads 16 floats of dot_with from global memory: a compiler-managed wait that drains this wave's
            // DMA ring once per block; leaving all items to the matrix waves instead measured slower: 1.04 vs 0.98 ms, 32 @ 1024^2 batch 8)
            if constexpr (MW == 4) {
                __syncthreads();  // A0
                if (!(skip & 8)) out_items(blk, b, opix, on16);
                __syncthreads();  // B0
                __syncthreads();  // A1
                if (!(skip & 8)) out_items(blk, b, opix, 16 + on16);
                __syncthreads();  // B1
            } else {
                for (int r = 0; r < 4; ++r) {  // 16 channels per round; this thread's item: channel 16 r + on16 (on16 = 0 .. 7 here)
                    __syncthreads();  // A
                    if (!(skip & 8)) out_items(blk, b, opix, 16 * r + on16);
                    __syncthreads();  // B
                }
            }
        }
        return;
    }
    // -------------------------------------------------------------------------------------------------- matrix waves
    const int half = lane >> 5, j = lane & 31;
    const int g = wave & 3;  // position group
    // The role of a matrix wave, instantiated per channel half HH (MW 8; wave-uniform branch below): the two halves differ in WHEN their
    // accumulators die -- half 0 hands its 16-channel rounds over first and then runs output items beside the transform waves, half 1
    // holds all 144 accumulator registers until round 2 -- and the compiler's liveness is per code path only if the paths are separate.
    auto matrix_role = [&](auto hh_c) __attribute__((always_inline)) {
    constexpr int hh = decltype(hh_c)::value;
    wf_f32x16 acc[9];
    // A operands, loaded one tick ahead and in place: a[q] is reloaded right behind the four MFMAs that consumed it.  Buffer loads: the
    // lane's part of the address is ONE VGPR, the (position, chunk) part a scalar offset -- no VALU address arithmetic in the wave that
    // issues the MFMAs (a wave's own instructions are never hidden by its MFMAs: tools/issue_probe.hip)
    constexpr int AR = MW == 4 ? 9 : 3;  // A operands held: a tick ahead (MW 4), or a ring of three positions (MW 8: 144 accumulator registers)
    float4 a[AR];
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uf), (short)0, (int)(unsigned)((int64_t)36 * K * N * 4), 0x00020000);
    const int voff_a = (half * N + n0 + 32 * hh + j) * 16;
    const unsigned stride_a = (unsigned)(2 * N) * 16u;  // bytes per (position, 8-channel chunk)
    auto a_at = [&](int q, int kc) __attribute__((always_inline)) {
        typedef float a_f32x4 __attribute__((ext_vector_type(4)));
        const a_f32x4 v = __builtin_bit_cast(a_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, voff_a, (int)((unsigned)(((9 * g + q) << kc_log2) + kc) * stride_a), 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    auto mfma_tick = [&](int stage, int cur_kc, int next_kc) __attribute__((always_inline)) {
        const float4* vs4 = reinterpret_cast<const float4*>(wsm + stage * VS);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 b4 = vs4[((9 * g + q) * 2 + half) * 32 + j];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % AR].x, b4.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % AR].y, b4.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % AR].z, b4.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % AR].w, b4.w, acc[q], 0, 0, 0);
            if constexpr (MW == 4) a[q] = a_at(q, next_kc);
            else a[q % AR] = (q + AR < 9) ? a_at(q + AR, cur_kc) : a_at(q + AR - 9, next_kc);  // three positions (~1500 cycles) ahead
        }
    };
#pragma unroll
    for (int q = 0; q < AR; ++q) a[q] = a_at(q, 0);
    auto fill_sctab = [&](int bi) __attribute__((always_inline)) {  // in_scale[b, 0..K) of block number bi (K <= 256)
        if (bi >= count) return;
        const int b = (xlo + bi * xstep) / per_img;
        if (tid < K) sctab[(bi & 1) * 256 + tid] = in_scale ? in_scale[(int64_t)b * K + tid] : 1.f;
    };
    auto fill_etab = [&](int bi) __attribute__((always_inline)) {  // noise patch and per-channel epilogue operands of block number bi
        const int blk = xlo + bi * xstep;
        const int b = blk / per_img, rem = blk - b * per_img;
        const int by = rem / bx_n, bx = rem - by * bx_n;
        if (tid < 128) {
            const int row = tid / TXN, qd = tid % TXN;
            *reinterpret_cast<float4*>(etab + row * BWP + 4 * qd) =
                (ACT == 1 && noise) ? *reinterpret_cast<const float4*>(noise + (int64_t)(by * BHP + row) * W + bx * BWP + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (tid < 128 + NCH) {
            etab[512 + tid - 128] = out_scale ? out_scale[(int64_t)b * N + n0 + tid - 128] : 1.f;
        } else if (tid < 128 + 2 * NCH) {
            etab[512 + NCH + tid - 128 - NCH] = (ACT && bias) ? bias[n0 + tid - 128 - NCH] : 0.f;
        } else if (tid < 128 + 3 * NCH) {
            etab[512 + 2 * NCH + tid - 128 - 2 * NCH] = (ACT == 2 && slope) ? slope[n0 + tid - 128 - 2 * NCH] : 1.f;
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
            if (!(skip & 4)) mfma_tick(c & 1, c, (c + 1) & (KC - 1));
            __syncthreads();
        }
        const int blk = xlo + i * xstep;
        int b;
        int64_t opix;
        block_pix(blk, b, opix);
        fill_sctab(i + 1);  // (the last transform of block i was one tick ago; the table of block i+1 is read from its pre-tick on)
#pragma unroll
        for (int r = 0; r < MW / 2; ++r) {  // (unrolled: the accumulator registers are indexed by q2)
            const int q2 = r & 1;
            if (!(skip & 8) && (MW == 4 || hh == (r >> 1))) {  // (MW 8: rounds 0, 1 = channel half 0, rounds 2, 3 = half 1)
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
                        mbuf[((9 * g + q) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * half) * 32 + j] = acc[q][8 * q2 + rr];
            }
            __syncthreads();  // A
            if constexpr (MW == 4 || hh == 0) {
                if (!(skip & 8)) out_items(blk, b, opix, 16 * r + on16);
            }
            __syncthreads();  // B
        }
    }
    };  // matrix_role
    if (MW == 8 && (wave >> 2)) matrix_role(std::integral_constant<int, 1>{});  // (wave-uniform)
    else matrix_role(std::integral_constant<int, 0>{});
}

}  // namespace w2e

using namespace w2e;

extern "C" {

int w2e_wino_weights_fused(const float* wp, float* uf, int k_ch, int n_ch, void* stream) {
    W2E_REQUIRE(wp && uf, "wino_weights_fused: null tensor");
    W2E_REQUIRE(k_ch > 0 && (k_ch & 7) == 0 && n_ch > 0 && (n_ch & 31) == 0, "wino_weights_fused: K %% 8 == 0, N %% 32 == 0 (got %d, %d)", k_ch, n_ch);
    const int64_t total = (int64_t)k_ch * n_ch;
    wino4_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(wp, uf, k_ch, n_ch);
    W2E_LAUNCH_CHECK("wino_weights_fused");
    return 0;
}

int w2e_wino_fused(const float* x, const float* in_scale, const float* uf, const float* out_scale, float* y, int batch, int k_ch,
                   int n_ch, int h, int w, int act, const float* noise, const float* noise_w, const float* bias, const float* slope,
                   const float* dot_with, float* dot_out, int wgs, void* stream) {
    W2E_REQUIRE(x && uf && y, "wino_fused: null tensor");
    W2E_REQUIRE(act >= 0 && act <= 2, "wino_fused: epilogue %d (0 none, 1 StyledConv, 2 bias + PReLU)", act);
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
    // persistent, specialised waves: K a power of two in 32 ... 256; the whole input behind ONE buffer descriptor
    W2E_REQUIRE(k_ch >= 32 && k_ch <= 256 && (k_ch & (k_ch - 1)) == 0, "wino_fused: K must be a power of two in 32 ... 256 (got %d)", k_ch);
    W2E_REQUIRE((int64_t)batch * k_ch * h * w * 4 < ((int64_t)1 << 32) - 64, "wino_fused: x exceeds 4 GB");
    int kc_log2 = 0;
    while ((8 << kc_log2) < k_ch) ++kc_log2;
    const int cus = cu_count();
    // 8 matrix waves / 64 output channels per workgroup wherever the layer has them (one input transform per 64 instead of per 32
    // channels; the kernel's comment says why that is what pays); W2E_TUNE_MW=4 keeps the 32-channel form (A/B)
    // -- unless that leaves CUs without a workgroup (128 -> 128 @ 64^2, batch 8: 64 blocks x 2 channel blocks = 128 workgroups, 57.8 us
    // against 41 with 256 workgroups of 32 channels; profiles/r05_irse_shapes.txt)
    const bool mw8 = (n_ch & 63) == 0 && options().tune_mw != 4 && (options().tune_mw == 8 || blocks * (n_ch / 64) >= cus);
    const int nby = n_ch / (mw8 ? 64 : 32);
    int gx = cus / nby;
    if (gx < 1) gx = 1;
    if (gx > blocks) gx = (int)blocks;
    if ((wgs & 0xffff) > 0 && (wgs & 0xffff) < gx) gx = wgs & 0xffff;  // (tests: several blocks per workgroup on small inputs; bits 16+: a tuning build's skip mask)
    const dim3 g2((unsigned)gx, (unsigned)nby);
    // XCD-contiguous block ownership (the kernel's `xmap`): needs whole eighths; W2E_TUNE_XCD bit 0 = 0 switches it off (A/B)
    const int xmap = ((options().tune_xcd < 0 || (options().tune_xcd & 1)) && (gx & 7) == 0 && (blocks & 7) == 0 && (blocks >> 3) >= (gx >> 3)) ? 1 : 0;
    const size_t lds3 = (size_t)(2 * 36 * 2 * 32 * 4 + 3 * 24 * 64 * 4 + 2 * 256 + 704) * 4;
    static unsigned done3[4][4];  // [epilogue][block shape x matrix waves]
    const bool wide = (w & 63) == 0 && !((wgs >> 16) & 16);  // blocks of 64 x 8 pixels (bit 4 of a tuning build's mask: keep 32 x 16)
#define W2E_WF3_(ACTv, DOTv, TXNv, MWv, slot, sub)                                                                                          \
    do {                                                                                                                                   \
        W2E_REQUIRE(big_lds_once((const void*)wino4_fused3_kernel<ACTv, DOTv, TXNv, MWv>, &done3[slot][sub]), "wino_fused: cannot enable %zu B of LDS", lds3); \
        wino4_fused3_kernel<ACTv, DOTv, TXNv, MWv><<<g2, 64 * (MWv + 4), lds3, s>>>(x, in_scale, uf, out_scale, y, batch, k_ch, n_ch, h, w, kc_log2,      \
                                                                                   (int)blocks, noise, noise_w, bias, slope, dot_with, dot_out,      \
                                                                                   wgs >> 16, xmap);                                                 \
    } while (0)
#define W2E_WF3(ACTv, DOTv, slot)                                     \
    do {                                                              \
        if (wide && mw8) W2E_WF3_(ACTv, DOTv, 16, 8, slot, 0);        \
        else if (wide) W2E_WF3_(ACTv, DOTv, 16, 4, slot, 1);          \
        else if (mw8) W2E_WF3_(ACTv, DOTv, 8, 8, slot, 2);            \
        else W2E_WF3_(ACTv, DOTv, 8, 4, slot, 3);                     \
    } while (0)
    if (act == 1) W2E_WF3(1, false, 0);
    else if (act == 2) W2E_WF3(2, false, 1);
    else if (dot_with) W2E_WF3(0, true, 2);
    else W2E_WF3(0, false, 3);
#undef W2E_WF3_
#undef W2E_WF3
    W2E_LAUNCH_CHECK("wino_fused");
    return 0;
}

}  // extern "C"
