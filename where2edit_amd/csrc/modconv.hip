// K1: modulated 3x3 convolution as an fp32-MFMA implicit GEMM for gfx950
// (replaces the grouped-conv arithmetic of models/stylegan2/model.py:234-276).
//
// Shared-weight form: y[b,o] = out_scale[b,o] * conv(Wp, in_scale[b,i] * x[b,i]) -- the per-sample
// modulation is folded into the activation tile while it is staged into LDS, the demodulation into the
// epilogue, so ONE packed weight tensor serves the whole batch (the reference materialises a
// [B,Cout,Cin,3,3] weight per call).
//
// GEMM view per workgroup: D[o, px] += A[o, k] * B[k, px], k = (ci, tap).
//   A = packed weights, global [K/8][9][2][N][4] (w2e_conv_pack) -> LDS ws[tap][h][TN] of float4: the float4 of
//       (tap, h, o) holds the chunk's channels 2c+h, c = 0..3, i.e. the A operand of 4 consecutive MFMAs of lane-half h
//   B = activation patch with halo, pre-multiplied by in_scale -> LDS xs[h][PH*PW] of float4, same channel split;
//       one ds_read_b128 per operand feeds 4 MFMAs, one ds_write_b128 pair stages 8 channels of a pixel
//   v_mfma_f32_32x32x2_f32: lane l supplies A[o = l&31][k = l>>5] and B[k = l>>5][px = l&31];
//   the two k of one MFMA are the channel pair (2c, 2c+1) at the SAME tap, so every LDS address is
//   lane-base + compile-time-regular offset.  D: col = l&31 = pixel, row = (r&3)+8*(r>>2)+4*(l>>5) = o,
//   so an accumulator register stores 32 consecutive pixels of one output channel (128 B, NCHW).
// Waves are arranged WO x WP (8 waves = 512 threads for the big tiles, 4 for the low-resolution ones); each
// wave owns NOB x NPB 32x32 accumulator tiles.  The K loop is software-pipelined: the next chunk's global
// loads are in flight (in registers) while the current chunk's MFMAs issue.
//
// Modes: SAME (stride 1, pad 1), UP (conv_transpose stride 2, computed as its 4 output phases: a
// workgroup owns ONE phase (blockIdx & 3) and runs only that phase's 4/2/2/1 taps, so the zeros of the
// stuffed image are never multiplied and the weight bytes staged per MFMA equal SAME's), DOWN (stride-2
// conv = adjoint of UP).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "../../include/w2e_irse.h"  // w2e_affine_act_fwd: the bias + PReLU pass after a split launch

namespace w2e {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* x;
    const float* wp;
    const float* in_scale;
    const float* out_scale;
    float* y;
    const float* noise;
    const float* noise_w;
    const float* bias;
    const float* dot_with;
    float* dot_out;
    int batch, K, N;
    int H, W;        // tile domain: output pixels for SAME/DOWN, input pixels for UP
    int in_h, in_w;  // input tensor
    int out_h, out_w;
    int th, tw, tw_log2;
    int tiles_x, tiles_y, tiles_n;
    int ph, pw, plane;
    unsigned pw_magic;  // ceil(2^32 / pw): idx / pw == umulhi(idx, magic) for idx < 2^16
    int border_wgs, groups_row, groups_col;  // UP: leading workgroups that compute the last row / column (32-position blocks of each)
    int tune_skip;      // tuning aid (W2E_TUNE_SKIP): bit0 = no output stores, bit1 = no K loop, bit2 = stage only the first chunk, bit5 = no UP border
    const float* slope;  // EPI_PRELU: per-output-channel negative slope (v = v > 0 ? v : slope[o]*v after out_scale and bias)
    int in_off;          // DOWN: the input origin is shifted by in_off (-1 = a stride-2 convolution with padding 1 of an in_h x in_w image)
    int splits, k_per;  // split-K: workgroup ks reduces channels [ks*k_per, (ks+1)*k_per) and adds atomically
    unsigned long long* stamps;  // tuning aid (W2E_TUNE_CLOCK): per workgroup {s_memtime, s_memrealtime} at start and end
};

enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DOT = 2, EPI_PRELU = 3 };

// Internal variant of W2E_CONV_UP chosen by the host per layer: a workgroup computes ALL FOUR output phases of a
// (4x smaller) input-pixel tile from one staged patch and all 9 taps -- the NPB accumulator columns of a wave are
// 4 phases x NPB/4 pixel blocks.  The per-phase form stages the activation patch once per phase (4 patches per 9 taps);
// this one stages it once, which wins where the patch, not the weights, dominates staging (few channels, many pixels).
constexpr int CONV_UPALL = 3;
// The work-skipping / clock-stamping aids exist only in a -DW2E_TUNING build (python -m where2edit_amd.build with
// W2E_HIPCC_FLAGS=-DW2E_TUNING); the shipped library cannot be told to skip its own work.
#ifdef W2E_TUNING
#define W2E_SKIP(p, bit) ((p).tune_skip & (bit))
#else
#define W2E_SKIP(p, bit) false
#endif
#ifdef W2E_STAMPS  // diagnostic build (W2E_HIPCC_FLAGS=-DW2E_STAMPS): 8 words per workgroup, see the W2E_TUNE_CLOCK report
constexpr int STAMP_STRIDE = 8;
#else
constexpr int STAMP_STRIDE = 4;
#endif
__host__ __device__ constexpr bool is_up(int mode) { return mode == W2E_CONV_UP || mode == CONV_UPALL; }

// Upper bound of ceil(patch / threads) for a tile of `tm` pixels (the host refuses geometries beyond it).
__host__ __device__ constexpr int max_patch_slots(int mode, int tm, int nt) {
    return mode == W2E_CONV_DOWN ? (nt == 512 ? (tm >= 1024 ? 9 : 5) : (tm >= 512 ? 9 : (tm >= 256 ? 5 : (tm >= 128 ? 3 : (tm >= 64 ? 2 : 1)))))
                                 : (nt == 512 ? (tm >= 2048 ? 5 : (tm >= 1024 ? 3 : 2)) : (tm >= 512 ? 3 : (tm >= 256 ? 2 : 1)));
}

// Tap bookkeeping.  SAME/DOWN use all 9 taps.  An UP workgroup owns ONE output phase (PY,PX) and only its taps
// exist: ta in {0,2} if PY==0 else {1}; tb likewise -> 4/2/2/1 taps, stored compactly (slot = ia*NB + ib).
template <int MODE, int PY, int PX>
struct Taps {
    static constexpr int NA = (MODE == W2E_CONV_UP) ? (PY ? 1 : 2) : 3;
    static constexpr int NB = (MODE == W2E_CONV_UP) ? (PX ? 1 : 2) : 3;
    static constexpr int N = NA * NB;
    __host__ __device__ static constexpr int ta(int slot) { return (MODE == W2E_CONV_UP) ? (PY ? 1 : 2 * (slot / NB)) : slot / 3; }
    __host__ __device__ static constexpr int tb(int slot) { return (MODE == W2E_CONV_UP) ? (PX ? 1 : 2 * (slot % NB)) : slot % 3; }
};

// Channels per K-chunk: the light UP phases take deeper chunks so that every chunk carries a comparable number
// of MFMAs between its two barriers (phase 0: 8 ch x 4 taps; phases 1,2: 16 x 2; phase 3: 16 x 1).
template <int MODE, int KC, int PY, int PX, int NACC, int MAXX>
struct Chunk {
    static constexpr int NT_ = Taps<MODE, PY, PX>::N;
    // (8-accumulator tiles and tiles with > 2 patch slots per thread keep KC: their register budget has no room
    // for a deeper prefetch)
    static constexpr int KCP = (MODE == W2E_CONV_UP && NACC < 8 && MAXX <= 2) ? ((KC * 4 / NT_) > 16 ? 16 : (KC * 4 / NT_)) : KC;
};

// One K-chunk of MFMAs for a wave.  LDS holds, per 8-channel sub-chunk, ws[slot][h][TN] and xs[h][plane] as float4
// (.x.y.z.w = channel pairs c = 0..3 of lane-half h), so each b128 read feeds 4 MFMAs.
// `hook(pos)` is called after the MFMAs of tap slot 0 (pos 0) and of the middle tap slot (pos 1) of the first sub-chunk:
// the caller issues the NEXT chunk's buffer loads there, in the shadow of the matrix pipe.
__device__ __forceinline__ void mul4(float4& a, const float4& s) { a.x *= s.x, a.y *= s.y, a.z *= s.z, a.w *= s.w; }

// SCALED (the LDS-DMA pipeline, where the staged activations are raw): the per-sample modulation s4[sub] = in_scale of the
// lane-half's 4 channels is applied to the fragments after the LDS read -- to the operand with fewer fragments per chunk.
// `piece(i)` is called after the i-th accumulator group (4 MFMAs) of the chunk, i = 0 .. groups-1: the LDS-DMA pipeline issues
// ONE piece of the next chunk's DMA there (see dma_loop).
struct NoPiece {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int MODE, int NOB, int NPB, int KCP, int TN, int PY, int PX, bool SCALED, bool AHEAD, typename Hook, typename Piece = NoPiece>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[NOB][NPB], const float4* ws, const float4* xs, int a_base,
                                           const int (&base)[NPB], int pw, int plane, const float4 (&s4)[KCP / 8], Hook hook,
                                           Piece piece = Piece()) {
    using T = Taps<MODE, PY, PX>;
    // The operand fragments of tap slot i+1 are fetched BEFORE the MFMAs of slot i issue (two statically indexed register sets):
    // the two waves of a SIMD run this code in lock-step, so an LDS round trip placed between two taps' MFMA blocks is paid by
    // both at once, with nobody feeding the matrix pipe meanwhile -- 9 exposed latencies per chunk were the K loop's 4 %.
    if constexpr (MODE == CONV_UPALL) {
        constexpr int NPP = NPB / 4;  // pixel blocks per phase
        constexpr int STEPS = (KCP / 8) * 9;
        float4 bv[AHEAD ? 2 : 1][4][NPP];  // per 8-channel sub-chunk: the 4 patch offsets (-(a>>1), -(b>>1)) the 9 taps read
        float4 av[AHEAD ? 2 : 1][NOB];
        auto load_b = [&](int sub, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int off = 0; off < 4; ++off)
#pragma unroll
                for (int pb = 0; pb < NPP; ++pb) bv[set][off][pb] = xs[base[pb] + sub * 2 * plane - (off >> 1) * pw - (off & 1)];
        };
        auto load_a = [&](int step, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) av[set][ob] = ws[a_base + step * 2 * TN + ob * 32];
        };
        if (AHEAD) load_b(0, 0), load_a(0, 0);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int sub = step / 9, slot = step % 9;
            const int cb = AHEAD ? (sub & 1) : 0, ca = AHEAD ? (step & 1) : 0;
            if (!AHEAD) {  // (tiles whose register budget has no room for a second fragment set)
                if (slot == 0) load_b(sub, 0);
                load_a(step, 0);
            }
            if (AHEAD && step + 1 < STEPS) {
                load_a(step + 1, ca ^ 1);
                if (slot == 8) load_b(sub + 1, cb ^ 1);
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks each fetch to just in front of its first use)
            }
            if (SCALED && slot == 0) {
#pragma unroll
                for (int off = 0; off < 4; ++off)
#pragma unroll
                    for (int pb = 0; pb < NPP; ++pb) mul4(bv[cb][off][pb], s4[sub]);
            }
            const int ta = slot / 3, tb = slot % 3;
            const int phase = (ta & 1) * 2 + (tb & 1), off = (ta >> 1) * 2 + (tb >> 1);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int pb = 0; pb < NPP; ++pb) {
                    f32x16& a = acc[ob][phase * NPP + pb];
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ca][ob].x, bv[cb][off][pb].x, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ca][ob].y, bv[cb][off][pb].y, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ca][ob].z, bv[cb][off][pb].z, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ca][ob].w, bv[cb][off][pb].w, a, 0, 0, 0);
                    piece(step * (NOB * NPP) + ob * NPP + pb);
                }
            if (step == 0) hook(0);
            if (step == 4) hook(1);
        }
        return;
    }
    constexpr int STEPS = (KCP / 8) * T::N;
    float4 av[AHEAD ? 2 : 1][NOB], bv[AHEAD ? 2 : 1][NPB];
    auto load_one = [&](int step, int set, int piece) __attribute__((always_inline)) {  // piece < NOB: an A fragment; else a B fragment
        const int sub = step / T::N, slot = step % T::N;
        const int ta = T::ta(slot), tb = T::tb(slot);
        if (piece < NOB) {
            av[set][piece] = ws[a_base + (sub * T::N + slot) * 2 * TN + piece * 32];
        } else {
            int toff;
            if (MODE == W2E_CONV_UP) toff = -(ta >> 1) * pw - (tb >> 1);
            else toff = ta * pw + tb;
            bv[set][piece - NOB] = xs[base[piece - NOB] + sub * 2 * plane + toff];
        }
    };
    constexpr int NG = NOB * NPB, NP = NOB + NPB;  // MFMA groups (one per accumulator, 4 MFMAs each) and fragment fetches of a slot
    if (AHEAD) {
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) load_one(0, 0, piece);
        if (SCALED) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) mul4(av[0][ob], s4[0]);
        }
    }
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int sub = step / T::N, slot = step % T::N;
        const int cur = AHEAD ? (step & 1) : 0;
        if (!AHEAD) {  // (tiles whose register budget has no room for a second fragment set)
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) load_one(step, 0, piece);
            if (SCALED) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) mul4(av[0][ob], s4[sub]);
            }
        }
        // AHEAD: the next slot's fragment fetches (NP ds_reads) and the modulation of its A fragments (NOB x 4 v_mul) are dealt out
        // over the accumulator groups of THIS slot's MFMAs instead of standing together between two slots' MFMA blocks.  The waves of
        // a SIMD run this stream in lock-step, so a block of non-MFMA instructions is issued by all of them at once with the matrix
        // pipe idle meanwhile (66.8 cycles per MFMA measured in the K loop, whatever the number of waves per SIMD, with or without
        // the chunk barriers); a single instruction between two MFMAs issues in the shadow of the partner wave's MFMA.
        const bool more = AHEAD && step + 1 < STEPS;
        const int nsub = (step + 1) / T::N;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const int g = ob * NPB + pb;
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ob].x, bv[cur][pb].x, acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ob].y, bv[cur][pb].y, acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ob].z, bv[cur][pb].z, acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ob].w, bv[cur][pb].w, acc[ob][pb], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int piece = 0; piece < NP; ++piece)  // fetch `piece` rides on group piece (NG >= NP) or floor(piece * NG / NP)
                        if ((NG >= NP ? piece : (piece * NG) / NP) == g) load_one(step + 1, cur ^ 1, piece);
                    if (SCALED) {
#pragma unroll
                        for (int o2 = 0; o2 < NOB; ++o2)  // the modulation of A fragment o2 on one of the last NOB groups (after its fetch)
                            if (g == NG - NOB + o2) mul4(av[cur ^ 1][o2], s4[nsub]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                piece(step * NG + g);
            }
        if (sub == 0 && slot == 0) hook(0);
        if (sub == 0 && slot == T::N / 2 && T::N > 1) hook(1);
        if (sub == 0 && slot == 0 && T::N == 1) hook(1);
    }
}

// Last row (Y = 2H) and last column (X = 2W) of the (2H+1)x(2W+1) transposed-conv output: only the a=2 (row) / b=2 (column)
// taps reach them -- a 1-D transposed convolution of the last input row / column.  They are produced by the FIRST `border_wgs`
// workgroups of the UP launch itself, on the matrix pipe: one WAVE per unit = (image, 32 output channels, 32 border positions
// v = v0..v0+31), computing both output parities of its positions,
//   even (X or Y = 2v):   sum_k  W[o,k,tapE0] xs[k,v] + W[o,k,tapE1] xs[k,v-1]
//   odd  (      2v+1):    sum_k  W[o,k,tapO ] xs[k,v]
// with xs = in_scale * x along the last row / column (0 outside it).  Both MFMA operands come straight from global memory in the
// order the MFMA consumes them -- the packed weights hold the A operands of four consecutive MFMAs as one float4 per lane, the
// activations are four dword loads -- with the next 8-channel group's loads in flight during the current group's 12 MFMAs; no LDS.
// (Round 2's version did this on the VALU with one block per 8 positions x 64 channels: 1056 workgroups on the 512->256 @ 64 layer
// at batch 8, one per CU at the kernel's LDS allocation, a latency-bound prefix of 40-190 us per up-sampling launch.)
template <int NT>
__device__ __forceinline__ void upconv_border(const ConvParams& p, float* smem, int wg) {
    constexpr int NW = NT / 64;  // the waves of a workgroup split the unit's channel range (a unit is a serial chain of dependent
                                 // loads: its latency, not its work, is what the launch sees) and join through LDS
    const int lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_ob = (p.N + 31) >> 5;
    const int blocks = p.groups_row + p.groups_col;  // 32-position blocks of the row border, then of the column border
    const int u = wg;                                  // one unit per workgroup
    const int blk = u % blocks, ob = (u / blocks) % n_ob, b = u / (blocks * n_ob);
    const bool is_row = blk < p.groups_row;
    const int v0 = (is_row ? blk : blk - p.groups_row) * 32;
    const int L = is_row ? p.W : p.H;  // input extent along the border
    const int v = v0 + j;              // this lane's position (as the B operand's column and as the output column)
    const int64_t in_plane = (int64_t)p.H * p.W;
    // per-lane byte offsets of xs[., v] and xs[., v-1] inside one channel plane; positions outside the row / column read 0
    const unsigned stride = is_row ? 4u : (unsigned)p.W * 4u;
    const unsigned fixed = is_row ? (unsigned)((p.H - 1) * p.W) * 4u : (unsigned)(p.W - 1) * 4u;
    const bool in0 = v < L, in1 = v >= 1 && v - 1 < L;
    const unsigned off0 = fixed + (unsigned)v * stride, off1 = fixed + (unsigned)(v - 1) * stride;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)b * p.K * in_plane), (short)0, (int)(unsigned)((int64_t)p.K * in_plane * 4), 0x00020000);
    const int groups8 = (p.K + 7) >> 3;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), (short)0, (int)(groups8 * 9 * 2 * p.N * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in_scale ? p.in_scale + (int64_t)b * p.K : p.x), (short)0, p.in_scale ? p.K * 4 : 0, 0x00020000);
    const bool has_is = p.in_scale != nullptr;
    const int tapE0 = is_row ? 6 : 2, tapE1 = 8, tapO = is_row ? 7 : 5;  // (2,0),(2,2),(2,1) for the row; (0,2),(2,2),(1,2) for the column
    const int o = ob * 32 + j;
    const int oc = o < p.N ? o : p.N - 1;
    const unsigned wlane = (unsigned)(half * p.N + oc) * 16u;        // float4 (tap, h, o) of a group: ((tap*2 + h)*N + o)*16 bytes
    const unsigned wtap = 2u * (unsigned)p.N * 16u, wgroup = 9u * wtap;
    const unsigned plane_b = (unsigned)(in_plane * 4);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 acc_e, acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_e[r] = 0.f, acc_o[r] = 0.f;
    constexpr int S = 2;  // register stages: the next group's 15 loads are in flight during a group's 12 MFMAs
    f32x4 wa[S], wb[S], wc[S];
    float x0[S][4], x1[S][4], sc[S][4];
    auto load = [&](int kc, int st) __attribute__((always_inline)) {
        const unsigned wbase = (unsigned)kc * wgroup + wlane;
        wa[st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wbase + (unsigned)tapE0 * wtap, 0, 0));
        wb[st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wbase + (unsigned)tapE1 * wtap, 0, 0));
        wc[st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wbase + (unsigned)tapO * wtap, 0, 0));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned ch = (unsigned)(8 * kc + 2 * c + half);  // channels >= K fall past the descriptors: 0
            // (the out-of-range marker is selected AFTER the channel offset is added: added to it, it would wrap back in range)
            x0[st][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, in0 ? off0 + ch * plane_b : 0xfffffff0u, 0, 0));
            x1[st][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, in1 ? off1 + ch * plane_b : 0xfffffff0u, 0, 0));
            sc[st][c] = has_is ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ch * 4u, 0, 0)) : 1.f;
        }
    };
    auto fma = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float a = x0[st][c] * sc[st][c], bm = x1[st][c] * sc[st][c];
            acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[st][c], a, acc_e, 0, 0, 0);
            acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[st][c], a, acc_o, 0, 0, 0);
            acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[st][c], bm, acc_e, 0, 0, 0);
        }
    };
    const int gper = (groups8 + NW - 1) / NW;
    const int g_lo = wave * gper, g_hi = (g_lo + gper < groups8) ? g_lo + gper : groups8;
#pragma unroll
    for (int st = 0; st < S; ++st)
        if (g_lo + st < g_hi) load(g_lo + st, st);
    for (int kc = g_lo; kc < g_hi; kc += S) {  // S groups per trip: statically indexed register stages
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (kc + st < g_hi) {
                fma(st);
                if (kc + st + S < g_hi) load(kc + st + S, st);
            }
        }
    }
    // join the NW channel slices: red[wave][reg 0..31][lane]; wave w then owns the registers [w*32/NW, (w+1)*32/NW)
    float (*red)[32][64] = reinterpret_cast<float (*)[32][64]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc_e[r], red[wave][16 + r][lane] = acc_o[r];
    __syncthreads();
    // phase-planar output [2][2][H+1][WP]: element (Y,X) lives at [Y&1][X&1][Y>>1][X>>1]; the row border is Y = 2H (plane row H),
    // the column border X = 2W (plane column W); the corner (2H, 2W) belongs to the row border (v = W, even)
    const int hp = p.H + 1, wpp = W2E_PLANAR_PITCH(p.W);
    const bool even_ok = is_row ? v <= p.W : v < p.H, odd_ok = is_row ? v < p.W : v < p.H;
    const int e_idx = is_row ? (0 * hp + p.H) * wpp + v : (0 * hp + v) * wpp + p.W;                   // (Y&1, X&1) = (0,0)
    const int o_idx = is_row ? (1 * hp + p.H) * wpp + v : (2 * hp + v) * wpp + p.W;                   // row: (0,1); column: (1,0)
    constexpr int RPW = 32 / NW;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int rr = wave * RPW + q;  // 0..15: even parity, 16..31: odd parity (uniform per wave)
        const int r = rr & 15;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w][rr][lane];
        const int och = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (och >= p.N) continue;
        const float os = p.out_scale ? p.out_scale[(int64_t)b * p.N + och] : 1.f;
        float* yp = p.y + ((int64_t)b * p.N + och) * 4 * hp * wpp;
        if (rr < 16) {
            if (even_ok) yp[e_idx] = sum * os;
        } else if (odd_ok) {
            yp[o_idx] = sum * os;
        }
    }
}

template <int MODE, int EPI, int NOB, int NPB, int WO, int WP, int KC, int DMA>
__global__ __launch_bounds__(64 * WO * WP, 2) void modconv_kernel(ConvParams p) {
    constexpr int NT = 64 * WO * WP;  // 256 threads (small tiles, 2 workgroups/CU) or 512 (big tiles, 1/CU)
    constexpr int TN = 32 * NOB * WO;
    constexpr int NPX = (MODE == CONV_UPALL) ? NPB / 4 : NPB;  // pixel blocks per wave (UPALL: NPB = 4 phases x NPX)
    constexpr int MAXX = max_patch_slots(MODE == CONV_UPALL ? W2E_CONV_UP : MODE, 32 * NPX * WP, NT);  // activation-patch elements per thread (x KC channels)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WS_FLOATS = (MODE == W2E_CONV_UP ? 32 : KC * 9) * TN;  // every phase stages KCP*NTAPS <= 32 (k,tap) rows
    float4* ws = reinterpret_cast<float4*>(smem);              // [KCP/8][NTAPS][2][TN] float4
    float4* xs = reinterpret_cast<float4*>(smem + WS_FLOATS);  // [KCP/8][2][plane]     float4

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: everything derived from it (the wave's output-channel block, its pixel blocks, the per-register
    // channel offsets of the epilogue's buffer stores) then lives in SGPRs -- left in a VGPR, every store's scalar-offset operand
    // becomes a readfirstlane "waterfall" loop (64-128 of them per wave, two thirds of the kernel's instructions)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int wo = wave / WP, wpx = wave % WP;

    int bid = blockIdx.x;
    if (is_up(MODE)) {
        if (bid < p.border_wgs) {  // uniform per workgroup
            if (W2E_SKIP(p, 32)) return;  // tuning aid: price the border workgroups
            upconv_border<NT>(p, smem, bid);
            return;
        }
        bid -= p.border_wgs;
    }
    // UP: phase-major grid, phase 0 (4 taps) first so the light phases fill the tail; (py,px) = (phase>>1, phase&1)
    const int per_phase = (gridDim.x - p.border_wgs) >> 2;
    const int phase = (MODE == W2E_CONV_UP) ? bid / per_phase : 0;
    if (MODE == W2E_CONV_UP) bid -= phase * per_phase;
    const int ks = bid % p.splits;
    bid /= p.splits;
    const int k_lo = ks * p.k_per;
    const int k_hi = W2E_SKIP(p, 2) ? k_lo : ((k_lo + p.k_per < p.K) ? k_lo + p.k_per : p.K);
#ifdef W2E_TUNING
    if (p.stamps && tid == 0) {
        p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
        p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    bid /= p.tiles_y;
    const int b = bid % p.batch;
    const int nt = bid / p.batch;
    const int n0 = nt * TN;
    const int r0 = ty * p.th, c0 = tx * p.tw;

    // per-lane LDS base of each of the wave's NPB pixel blocks (pixel coordinates are recomputed in the
    // epilogue instead of being kept live across the MFMA loop)
    int base[NPB];
#pragma unroll
    for (int pb = 0; pb < NPX; ++pb) {
        const int m = (wpx * NPX + pb) * 32 + j;
        const int ly = m >> p.tw_log2, lx = m & (p.tw - 1);
        const bool ok = ly < p.th && r0 + ly < p.H && c0 + lx < p.W;
        int off;
        if (MODE == W2E_CONV_SAME) off = ly * p.pw + lx;            // patch origin (r0-1, c0-1); tap (a,b): +a*pw+b
        else if (is_up(MODE)) off = (ly + 1) * p.pw + lx + 1;  // same origin; tap: -(a>>1)*pw-(b>>1)
        else off = 2 * ly * p.pw + 2 * lx;                          // origin (2r0, 2c0); tap: +a*pw+b
        if (!ok) off = is_up(MODE) ? p.pw + 1 : 0;
        base[pb] = half * p.plane + off;
    }

    f32x16 acc[NOB][NPB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][pb][r] = 0.f;

    const int64_t in_plane = (int64_t)p.in_h * p.in_w;
    const int patch = p.ph * p.pw;
    const int oy0 = (MODE == W2E_CONV_DOWN) ? 2 * r0 + p.in_off : r0 - 1;
    const int ox0 = (MODE == W2E_CONV_DOWN) ? 2 * c0 + p.in_off : c0 - 1;

    // ---- software pipeline (T14 "issue early / write late"): the global loads of chunk k+1 are issued into
    // registers right after chunk k is published to LDS and land while chunk k's MFMAs run; they are
    // written to LDS (activations multiplied by in_scale on the way) after the barrier that retires chunk k.
    // Instantiated once per output phase for UP (the phase is uniform per workgroup; the switch sits OUTSIDE
    // the loop so that each phase gets its own loop nest and register allocation).
    auto k_loop = [&](auto py_c, auto px_c) __attribute__((always_inline)) {
        constexpr int PY = decltype(py_c)::value, PX = decltype(px_c)::value;
        using T = Taps<MODE, PY, PX>;
        constexpr int KCP = Chunk<MODE, KC, PY, PX, NOB * NPB, MAXX>::KCP;
        static_assert(KCP % 8 == 0, "chunks are whole 8-channel groups");
        constexpr int WQ4 = (KCP / 8) * T::N * 2 * TN;        // float4s of the staged weight chunk
        constexpr int WQ = (WQ4 + NT - 1) / NT;                // ... per thread
        const int a_base = half * TN + wo * NOB * 32 + j;      // float4 index of (slot 0, h, o)
        float4 wr[WQ];
        float xr[MAXX][KCP];
        float sc[KCP];

        // The staging instructions of a chunk issue while the matrix pipe is idle (every wave of the workgroup is in the
        // same phase), so they are kept to the loads themselves -- buffer loads: `descriptor (SGPRs) + per-thread byte
        // offset (one VGPR, computed once per tile) + per-channel scalar offset`, no branches, no address arithmetic.
        // The hardware range check returns 0 for offsets past the descriptor's size, which IS the zero padding: halo
        // pixels outside the image and unused prefetch slots carry an out-of-range offset.  (The descriptor spans the K
        // channel planes of one image -- on gfx9 the scalar offset takes part in the range check; channels past K are
        // clamped to the last one, their sc is 0.)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + (int64_t)b * p.K * in_plane), (short)0, (int)(unsigned)((int64_t)p.K * in_plane * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.wp), (short)0, (int)(((p.K + 7) >> 3) * 9 * 2 * p.N * 16), 0x00020000);
        unsigned xoff[MAXX], woff[WQ];
#pragma unroll
        for (int t = 0; t < MAXX; ++t) {
            const int idx = tid + t * NT;
            const int py = (int)__umulhi((unsigned)idx, p.pw_magic);
            const int px = idx - py * p.pw;
            const int iy = oy0 + py, ix = ox0 + px;
            const bool inb = idx < patch && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            xoff[t] = inb ? (unsigned)(iy * p.in_w + ix) * 4u : 0xfffffff0u;
        }
        const unsigned wgroup_bytes = 9u * 2u * (unsigned)p.N * 16u;  // one 8-channel group of the packed weights
#pragma unroll
        for (int t = 0; t < WQ; ++t) {
            const int q = tid + t * NT;
            const int o = q % TN, rh = q / TN;             // rh = (sub*NTAPS + slot)*2 + h
            const int hh = rh & 1, ss = rh >> 1;
            const int slot = ss % T::N, sub = ss / T::N;
            const int tap = T::ta(slot) * 3 + T::tb(slot);
            // output channels past N read channel N-1 (their accumulator rows are never stored)
            const int oc = n0 + o < p.N ? n0 + o : p.N - 1;
            woff[t] = q < WQ4 ? (unsigned)sub * wgroup_bytes + (unsigned)(((tap * 2 + hh) * p.N + oc) * 16) : 0xfffffff0u;
        }

        auto prefetch = [&](int k0) __attribute__((always_inline)) {
            const unsigned wbase = (unsigned)(k0 >> 3) * wgroup_bytes;  // groups past K fall out of the descriptor: zeros
#pragma unroll
            for (int t = 0; t < WQ; ++t) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[t] + wbase, 0, 0));
                wr[t] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int ci = 0; ci < KCP; ++ci)
                sc[ci] = (k0 + ci < k_hi) ? (p.in_scale ? p.in_scale[(int64_t)b * p.K + k0 + ci] : 1.f) : 0.f;
#pragma unroll
            for (int ci = 0; ci < KCP; ++ci) {
                const int cc = k0 + ci < p.K ? k0 + ci : p.K - 1;        // uniform
                const unsigned soff = (unsigned)cc * (unsigned)(in_plane * 4);
#pragma unroll
                for (int t = 0; t < MAXX; ++t)
                    xr[t][ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[t], soff, 0));
            }
        };
        auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < WQ; ++t) {
                const int q = tid + t * NT;
                if (q < WQ4) ws[q] = wr[t];
            }
#pragma unroll
            for (int t = 0; t < MAXX; ++t) {
                const int idx = tid + t * NT;
                if (idx < patch) {
#pragma unroll
                    for (int sub = 0; sub < KCP / 8; ++sub) {
                        const int c0 = sub * 8;
                        xs[(sub * 2 + 0) * p.plane + idx] = make_float4(xr[t][c0] * sc[c0], xr[t][c0 + 2] * sc[c0 + 2],
                                                                         xr[t][c0 + 4] * sc[c0 + 4], xr[t][c0 + 6] * sc[c0 + 6]);
                        xs[(sub * 2 + 1) * p.plane + idx] = make_float4(xr[t][c0 + 1] * sc[c0 + 1], xr[t][c0 + 3] * sc[c0 + 3],
                                                                         xr[t][c0 + 5] * sc[c0 + 5], xr[t][c0 + 7] * sc[c0 + 7]);
                    }
                }
            }
        };

        prefetch(k_lo);
        for (int k0 = k_lo; k0 < k_hi; k0 += KCP) {
            if (!W2E_SKIP(p, 4) || k0 == k_lo) {  // tuning aid: bit 2 = stage only the first chunk
                __syncthreads();  // everyone finished reading the previous chunk
                commit();
                __syncthreads();
            }
            // the next chunk's loads are issued from inside the MFMA stream; the two waves that share a SIMD (w, w+4 in a
            // 512-thread workgroup) do it at different taps so that one of them always feeds the matrix pipe
            const bool do_pf = k0 + KCP < k_hi && !W2E_SKIP(p, 4);
            const int my_pos = (NT == 512) ? (wave >> 2) : 0;
            const float4 no_scale[KCP / 8] = {};
            mfma_chunk<MODE, NOB, NPB, KCP, TN, PY, PX, false, false>(acc, ws, xs, a_base, base, p.pw, p.plane, no_scale, [&](int pos) __attribute__((always_inline)) {
                if (do_pf && pos == my_pos) prefetch(k0 + KCP);
            });
        }
    };

    // ---- LDS-DMA pipeline (DMA = true; SAME / all-phase UP / DOWN): `buffer_load ... lds` writes the next chunk straight
    // into the OTHER of two LDS stages while this chunk's MFMAs run -- no staging registers, no LDS write pass, one barrier
    // per chunk.  A DMA wave-instruction fills 64 consecutive LDS dwords (b32) or float4s (b128) from per-lane addresses, so
    // the LDS image keeps its MFMA order and the lanes pick the matching global elements: for the patch, lane 4i+c of an
    // instruction fetches channel 2c+h of pixel i (the float4 of a pixel = 4 lanes); for the weights one lane = one packed
    // float4.  Halo pixels outside the image, channels >= K and weight groups past K carry out-of-range offsets: the
    // hardware writes zeros.  The activations land unscaled; in_scale is applied to the fragments (mfma_chunk SCALED) from
    // a per-image table in LDS.  The host pads p.plane to a multiple of 16 pixels so that no instruction straddles planes.
    auto dma_loop = [&](auto) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) char lds_char;
        typedef __attribute__((address_space(3))) void lds_void;
        constexpr int WQ4 = (KC / 8) * 9 * 2 * TN;   // float4s of a staged weight chunk (a multiple of 64)
        constexpr int WQ = (WQ4 + NT - 1) / NT;
        constexpr int XT = 4 * MAXX + 1;             // >= ceil(4 * plane / NT), dword slots per thread per (sub, h) plane
        // the plane holds at least the tile's own pixels (4x that for DOWN): slots below XT_MIN are whole for every wave
        constexpr int XT_MIN = (MODE == W2E_CONV_DOWN ? 16 : 4) * (32 * NPX * WP) / NT;
        static_assert(WQ4 % 64 == 0, "whole wave-instructions");
        const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int plane4 = p.plane * 4;
        const int stage_floats = WS_FLOATS + (KC / 8) * 2 * plane4;
        lds_char* const l0 = (lds_char*)smem;
        // in_scale table [ceil(K/8)][2] float4: (g, h).c = channel 8g + 2c + h; after the two stages (DMA == 2: after the fp32
        // staging area and the bf16 operand images)
        float* const stw = smem + (DMA == 2 ? stage_floats + 4 * (20 * TN + 2 * p.plane) : 2 * stage_floats);
        const float4* const st = reinterpret_cast<const float4*>(stw);
        const int a_base = half * TN + wo * NOB * 32 + j;
        unsigned xoff[XT], woff[WQ];
        const unsigned cplane = (unsigned)(2 * (lane & 3)) * (unsigned)(in_plane * 4);
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            const int idx = (tid + t * NT) >> 2;  // pixel of the padded plane
            const int py = (int)__umulhi((unsigned)idx, p.pw_magic);
            const int px = idx - py * p.pw;
            const int iy = oy0 + py, ix = ox0 + px;
            const bool inb = idx < patch && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            xoff[t] = inb ? (unsigned)(iy * p.in_w + ix) * 4u + cplane : 0xfffffff0u;
        }
        const unsigned wgroup_bytes = 9u * 2u * (unsigned)p.N * 16u;
#pragma unroll
        for (int t = 0; t < WQ; ++t) {
            const int q = tid + t * NT;
            const int o = q % TN, rh = q / TN;  // rh = (sub*9 + tap)*2 + h
            const int hh = rh & 1, ss = rh >> 1;
            const int tap = ss % 9, sub = ss / 9;
            const int oc = n0 + o < p.N ? n0 + o : p.N - 1;
            woff[t] = q < WQ4 ? (unsigned)sub * wgroup_bytes + (unsigned)(((tap * 2 + hh) * p.N + oc) * 16) : 0xfffffff0u;
        }
        // The DMA goes out through inline assembly, not __builtin_amdgcn_raw_ptr_buffer_load_lds: for the builtin the compiler knows
        // an asynchronous LDS write is in flight and, lacking alias information, puts `s_waitcnt vmcnt(0)` in front of the next
        // ds_read -- the issuing wave then sat out the whole flight time of its pieces (2-5 k cycles per chunk, read off the ISA and
        // off the stamps: "DMA issue" 14 % of a SAME tile's K loop, 26 % of an all-phase UP tile's) right after issuing them.  The
        // only wait the pipeline needs is the vmcnt(0) in front of the barrier that opens the NEXT chunk.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto raw_rsrc = [](const float* ptr, unsigned bytes) __attribute__((always_inline)) {  // base, stride 0, size, raw-buffer flags
            const uint64_t a64 = (uint64_t)(uintptr_t)ptr;
            i32x4 d;
            d[0] = (int)(unsigned)a64, d[1] = (int)(unsigned)((a64 >> 32) & 0xffffu), d[2] = (int)bytes, d[3] = 0x00020000;
            return d;
        };
        const i32x4 qx = raw_rsrc(p.x + (int64_t)b * p.K * in_plane, (unsigned)((int64_t)p.K * in_plane * 4));
        const i32x4 qw = raw_rsrc(p.wp, (unsigned)(((p.K + 7) >> 3) * 9 * 2 * p.N * 16));
        const unsigned lds0 = (unsigned)(uintptr_t)l0;
        // One piece (one wave-instruction) of a chunk's DMA: pieces 0 .. WQ-1 are the weights, then the (sub, h) planes of the patch.
        // Issued back to back, a wave's 15-17 pieces took it 5-6 k cycles (the CU's texture path accepts one such instruction per
        // ~90 cycles, and all waves queue on it at once) during which it fed the matrix pipe nothing; dealt out one per few MFMA
        // groups over the whole chunk (mfma_chunk's `piece`), each finds the path free.
        constexpr int NPIECE = WQ + 2 * (KC / 8) * XT;
        auto issue_piece = [&](int k0, int stage, int i) __attribute__((always_inline)) {
            const unsigned lb = lds0 + (unsigned)(stage * stage_floats * 4);
#if defined(__HIP_DEVICE_COMPILE__)
            if (i < WQ) {
                const int t = i;
                const int start = t * NT + swave * 64;
                const unsigned wbase = (unsigned)(k0 >> 3) * wgroup_bytes;
                if ((t + 1) * NT <= WQ4 || start < WQ4)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lb + (unsigned)(start * 16)), "v"(woff[t]), "s"(qw), "s"(wbase) : "memory");
            } else if (i < NPIECE) {
                const int sh = (i - WQ) / XT, t = (i - WQ) % XT;
                const unsigned soff = (unsigned)(k0 + (sh >> 1) * 8 + (sh & 1)) * (unsigned)(in_plane * 4);
                const unsigned xb = lb + (unsigned)((WS_FLOATS + sh * plane4) * 4);
                const int start = t * NT + swave * 64;
                if (t < XT_MIN || start < plane4)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(xb + (unsigned)(start * 4)), "v"(xoff[t]), "s"(qx), "s"(soff) : "memory");
            }
#endif
        };
        auto issue = [&](int k0, int stage) __attribute__((always_inline)) {
            const unsigned lb = lds0 + (unsigned)(stage * stage_floats * 4);
            const unsigned wbase = (unsigned)(k0 >> 3) * wgroup_bytes;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int t = 0; t < WQ; ++t) {
                const int start = t * NT + swave * 64;
                if ((t + 1) * NT <= WQ4 || start < WQ4)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lb + (unsigned)(start * 16)), "v"(woff[t]), "s"(qw), "s"(wbase) : "memory");
            }
#pragma unroll
            for (int sh = 0; sh < 2 * (KC / 8); ++sh) {  // the (sub, h) planes of the patch
                const unsigned soff = (unsigned)(k0 + (sh >> 1) * 8 + (sh & 1)) * (unsigned)(in_plane * 4);
                const unsigned xb = lb + (unsigned)((WS_FLOATS + sh * plane4) * 4);
#pragma unroll
                for (int t = 0; t < XT; ++t) {
                    const int start = t * NT + swave * 64;
                    if (t < XT_MIN || start < plane4)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(xb + (unsigned)(start * 4)), "v"(xoff[t]), "s"(qx), "s"(soff) : "memory");
                }
            }
#endif
        };
        if (k_lo < k_hi) issue(k_lo, 0);
        for (int e = tid; e < ((p.K + 7) >> 3) * 8; e += NT) {
            const int ch = 8 * (e >> 3) + 2 * (e & 3) + ((e >> 2) & 1);
            stw[e] = ch < p.K ? (p.in_scale ? p.in_scale[(int64_t)b * p.K + ch] : 1.f) : 0.f;
        }
        const int my_pos = (NT == 512) ? (swave >> 2) : 0;
        if constexpr (DMA == 2) {
            // ---- opt-in (W2E_CONV_PRECISION=bf16x3): fp32 as three bf16 products on v_mfma_f32_32x32x16_bf16.  The chunk lands by
            // LDS-DMA in the fp32 layout (staging area S), is split into hi = bf16(a), lo = bf16(a - hi) while it is copied
            // to the operand images -- weights BW[q][tap][TN], patch BX[q][pixel], 16 B = the chunk's 8 channels in the order
            // 0,2,4,6,1,3,5,7, in_scale applied to the patch on the way -- and an MFMA's K = 16 is those 8 channels at TWO taps:
            // lane-half h supplies tap 2*pair + h (tap 9 is a zero weight).  acc += hi*hi + hi*lo + lo*hi: 15 MFMAs of 32
            // cycles per accumulator per chunk instead of 36 of 64.
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8* const bw = reinterpret_cast<bf16x8*>(smem + stage_floats);
            bf16x8* const bx = bw + 20 * TN;
            const float4* const wsf = reinterpret_cast<const float4*>(smem);
            const float4* const xsf = wsf + WS_FLOATS / 4;
            for (int i = tid; i < 2 * TN; i += NT) {  // tap 9 of both halves: zero weights, written once
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
                bw[(i / TN) * 10 * TN + 9 * TN + (i % TN)] = z;
            }
            auto split8 = [](const float (&v)[8], bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const __bf16 h = (__bf16)v[e];
                    hi[e] = h;
                    lo[e] = (__bf16)(v[e] - (float)h);
                }
            };
            // tap pairs.  SAME / DOWN: (2p, 2p+1), all into the same accumulators.  All-phase UP: the two taps of a pair must
            // belong to the same output phase: {0,2} {6,8} -> phase 0, {1,7} -> 1, {3,5} -> 2, {4,-} -> 3
            constexpr bool UPA = MODE == CONV_UPALL;
            constexpr int kTap0[5] = {0, UPA ? 6 : 2, UPA ? 1 : 4, UPA ? 3 : 6, UPA ? 4 : 8};
            constexpr int kTap1[5] = {UPA ? 2 : 1, UPA ? 8 : 3, UPA ? 7 : 5, UPA ? 5 : 7, UPA ? 4 : 8};  // (pair 4: the zero weight)
            constexpr int kPhase[5] = {0, 0, UPA ? 1 : 0, UPA ? 2 : 0, UPA ? 3 : 0};
            constexpr int kSlot[9] = {0, UPA ? 4 : 1, UPA ? 1 : 2, UPA ? 6 : 3, UPA ? 8 : 4, UPA ? 7 : 5, UPA ? 2 : 6, UPA ? 5 : 7, UPA ? 3 : 8};
            int pix[NPX];
#pragma unroll
            for (int pb = 0; pb < NPX; ++pb) pix[pb] = base[pb] - half * p.plane;
            const int a0 = half * TN + wo * NOB * 32 + j;
            for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();  // the chunk has landed in S; everybody is done with the previous chunk's operand images
                for (int idx = tid; idx < 9 * TN; idx += NT) {
                    const int tap = idx / TN, o = idx - tap * TN;
                    const float4 f0 = wsf[(tap * 2 + 0) * TN + o], f1 = wsf[(tap * 2 + 1) * TN + o];
                    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                    bf16x8 hi, lo;
                    split8(v, hi, lo);
                    bw[kSlot[tap] * TN + o] = hi, bw[10 * TN + kSlot[tap] * TN + o] = lo;
                }
                const float4 s0 = st[(k0 >> 3) * 2], s1 = st[(k0 >> 3) * 2 + 1];
                for (int idx = tid; idx < p.plane; idx += NT) {
                    const float4 f0 = xsf[idx], f1 = xsf[p.plane + idx];
                    const float v[8] = {f0.x * s0.x, f0.y * s0.y, f0.z * s0.z, f0.w * s0.w, f1.x * s1.x, f1.y * s1.y, f1.z * s1.z, f1.w * s1.w};
                    bf16x8 hi, lo;
                    split8(v, hi, lo);
                    bx[idx] = hi, bx[p.plane + idx] = lo;
                }
                __syncthreads();  // operand images complete; S is free for the next chunk's DMA
                const bool do_pf = k0 + KC < k_hi;
#pragma unroll
                for (int pair = 0; pair < 5; ++pair) {
                    const int t0 = kTap0[pair], t1 = kTap1[pair];
                    // patch offset of a tap relative to the lane's pixel (as in mfma_chunk)
                    const int off0 = UPA ? -((t0 / 3) >> 1) * p.pw - ((t0 % 3) >> 1) : (t0 / 3) * p.pw + t0 % 3;
                    const int off1 = UPA ? -((t1 / 3) >> 1) * p.pw - ((t1 % 3) >> 1) : (t1 / 3) * p.pw + t1 % 3;
                    const int hoff = half ? off1 : off0;
                    bf16x8 ah[NOB], al[NOB], bh[NPX], bl[NPX];
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) ah[ob] = bw[pair * 2 * TN + a0 + ob * 32], al[ob] = bw[10 * TN + pair * 2 * TN + a0 + ob * 32];
#pragma unroll
                    for (int pb = 0; pb < NPX; ++pb) bh[pb] = bx[pix[pb] + hoff], bl[pb] = bx[p.plane + pix[pb] + hoff];
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                        for (int pb = 0; pb < NPX; ++pb) {
                            f32x16& a = acc[ob][kPhase[pair] * NPX + pb];
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ob], bh[pb], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ob], bl[pb], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ob], bh[pb], a, 0, 0, 0);
                        }
                    if (do_pf && ((pair == 0 && my_pos == 0) || (pair == 2 && my_pos == 1))) issue(k0 + KC, 0);
                }
            }
            return;
        }
        int stage = 0;
#ifdef W2E_STAMPS
        unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_t0 = __builtin_amdgcn_s_memtime();
#endif
        for (int k0 = k_lo; k0 < k_hi; k0 += KC, stage ^= 1) {
#ifdef W2E_STAMPS
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const unsigned long long tb = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long tc = __builtin_amdgcn_s_memtime();
            st_wait += tb - ta, st_bar += tc - tb;
#else
            if (!W2E_SKIP(p, 8) || k0 == k_lo) {  // (tuning aid, bit 3: no wait / barrier after the first chunk)
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's DMA pieces of chunk k0 have landed ...
                __syncthreads();                      // ... everybody's have, and everybody is done reading the other stage
            }
#endif
            const float4* wsc = reinterpret_cast<const float4*>(smem + stage * stage_floats);
            const float4* xsc = wsc + WS_FLOATS / 4;
            float4 s4[KC / 8];
#pragma unroll
            for (int sub = 0; sub < KC / 8; ++sub) s4[sub] = st[((k0 >> 3) + sub) * 2 + half];
            const bool do_pf = k0 + KC < k_hi && !W2E_SKIP(p, 4);  // (tuning aid, bit 2: stage only the first chunk)
            // groups of this chunk: (KC/8) x 9 taps x accumulators per tap (all-phase UP: a tap feeds one phase's accumulators)
            constexpr int GROUPS = (KC / 8) * 9 * (MODE == CONV_UPALL ? NOB * (NPB / 4) : NOB * NPB);
            constexpr int PSTRIDE = GROUPS / NPIECE > 0 ? GROUPS / NPIECE : 1;             // a piece every PSTRIDE groups, or
            constexpr int PPG = NPIECE <= GROUPS ? 1 : (NPIECE + GROUPS - 1) / GROUPS;   // PPG pieces per group (most of them empty slots)
            mfma_chunk<MODE, NOB, NPB, KC, TN, 0, 0, true, (NT == 512)>(
                acc, wsc, xsc, a_base, base, p.pw, p.plane, s4, [&](int) __attribute__((always_inline)) {},
                [&](int g) __attribute__((always_inline)) {
                    if (g % PSTRIDE == 0 && (g / PSTRIDE) * PPG < NPIECE) {
#ifdef W2E_STAMPS
                        const unsigned long long ti = __builtin_amdgcn_s_memtime();
#endif
                        if (do_pf) {
#pragma unroll
                            for (int q = 0; q < PPG; ++q) issue_piece(k0 + KC, stage ^ 1, (g / PSTRIDE) * PPG + q);
                        }
#ifdef W2E_STAMPS
                        st_issue += __builtin_amdgcn_s_memtime() - ti;
#endif
                    }
                });
        }
#ifdef W2E_STAMPS
        if (p.stamps && tid == 0) {
            p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 4] = st_wait;
            p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 5] = st_bar;
            p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 6] = __builtin_amdgcn_s_memtime() - st_t0;
            p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 7] = st_issue;
        }
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if constexpr (DMA) {
        static_assert(MODE != W2E_CONV_UP, "the per-phase UP form keeps the register pipeline");
        dma_loop(0);
    } else if (MODE == W2E_CONV_UP) {
        switch (phase) {
            case 0: k_loop(I0{}, I0{}); break;
            case 1: k_loop(I0{}, I1{}); break;
            case 2: k_loop(I1{}, I0{}); break;
            default: k_loop(I1{}, I1{}); break;
        }
    } else {
        k_loop(I0{}, I0{});
    }

    // ---- epilogue
    // (EPI_ACT: the activation's gain sqrt(2) is folded into out_scale, bias and the noise strength -- lrelu(v)*g = lrelu(v*g), g > 0)
    const float nw = (EPI == EPI_ACT && p.noise) ? p.noise_w[0] * 1.4142135623730951f : 0.f;
    float* red = smem;  // EPI_DOT: TN partial sums (reuses the weight tile after a barrier)
    if (EPI == EPI_DOT) {
        __syncthreads();
        if (tid < TN) red[tid] = 0.f;
        __syncthreads();
    }
    int gy[NPB], gx[NPB];
    bool valid[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int m = (wpx * NPX + pb % NPX) * 32 + j;
        const int ly = m >> p.tw_log2, lx = m & (p.tw - 1);
        gy[pb] = r0 + ly;
        gx[pb] = c0 + lx;
        valid[pb] = ly < p.th && gy[pb] < p.H && gx[pb] < p.W;
    }
    // CDNA4's vmcnt counts loads AND stores in order, so a load issued after a store cannot be consumed before
    // that store has completed: every value the epilogue reads from memory is loaded BEFORE its first store.
    // Offsets are 32-bit (host checks the tensors are < 2^31 elements).
    int pix[NPB];  // element offset of the lane's pixel inside one (b, o) output plane
    float nz[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        // UP writes its phase plane of the phase-planar image [4][H+1][W+1]: unit-stride rows
        const int ph_pb = (MODE == CONV_UPALL) ? pb / NPX : phase;
        if (is_up(MODE)) pix[pb] = (ph_pb * (p.H + 1) + gy[pb]) * W2E_PLANAR_PITCH(p.W) + gx[pb];
        else pix[pb] = gy[pb] * p.out_w + gx[pb];
        nz[pb] = (EPI == EPI_ACT && p.noise && valid[pb]) ? nw * p.noise[gy[pb] * p.out_w + gx[pb]] : 0.f;
    }
    const int out_plane = is_up(MODE) ? 4 * (p.H + 1) * W2E_PLANAR_PITCH(p.W) : p.out_h * p.out_w;
    // Output (and dot_with) accesses as buffer operations on a descriptor spanning this image's N planes: per-lane byte
    // offset (pixel, + 4 planes for the upper lane-half) computed once, per-register scalar offset = the output channel's
    // plane.  Pixels outside the tile/image carry an out-of-range offset; channels >= N fall past the descriptor: the
    // hardware drops those stores / returns 0 -- no branches, no 64-bit address arithmetic per access.
    const unsigned plane_bytes = (unsigned)out_plane * 4u;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        p.y + (int64_t)b * p.N * out_plane, (short)0, (int)((unsigned)p.N * plane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_DOT ? p.dot_with + (int64_t)b * p.N * out_plane : p.x), (short)0,
        EPI == EPI_DOT ? (int)((unsigned)p.N * plane_bytes) : 0, 0x00020000);
    unsigned yoff[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) yoff[pb] = valid[pb] ? (unsigned)pix[pb] * 4u + (unsigned)(4 * half) * plane_bytes : 0xfffffff0u;
    // The per-channel epilogue vectors (out_scale[b,:], bias, PReLU slope) as buffer loads too: one descriptor each (N floats; a
    // null vector = a zero-length descriptor and a uniform default), per-lane offset = the lane-half's channel, the register's
    // channel as an immediate -- no per-lane branches.  Channels >= N read 0: their rows are never stored.  (The epilogue's
    // instruction COUNT is what it costs: while the other waves of the SIMD stream MFMAs it issues roughly one instruction per
    // MFMA slot, so 64-128 stores' worth of branches and address arithmetic were 40 % of a K = 32 tile's lifetime.)
    const bool has_os = p.out_scale != nullptr, has_bs = (EPI == EPI_ACT || EPI == EPI_PRELU) && p.bias != nullptr;
    const bool has_sl = EPI == EPI_PRELU && p.slope != nullptr;
    const __amdgpu_buffer_rsrc_t ros = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_os ? p.out_scale + (int64_t)b * p.N : p.x), (short)0, has_os ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_bs ? p.bias : p.x), (short)0, has_bs ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_sl ? p.slope : p.x), (short)0, has_sl ? p.N * 4 : 0, 0x00020000);
    constexpr float kGain = EPI == EPI_ACT ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        float os[16], bs[16];
        const unsigned coff = (unsigned)(n0 + (wo * NOB + ob) * 32 + 4 * half) * 4u;  // the lane-half's first channel, in bytes
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ro = coff + (unsigned)((r & 3) + 8 * (r >> 2)) * 4u;
            os[r] = has_os ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ros, ro, 0, 0)) * kGain : kGain;
            bs[r] = has_bs ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbs, ro, 0, 0)) * kGain : 0.f;
        }
        float sl[EPI == EPI_PRELU ? 16 : 1];
        if (EPI == EPI_PRELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned ro = coff + (unsigned)((r & 3) + 8 * (r >> 2)) * 4u;
                sl[r] = has_sl ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsl, ro, 0, 0)) : 1.f;
            }
        }
        if (EPI == EPI_DOT) {  // pass 1: loads + reductions only, 8 rows (8*NPB loads in flight per lane) at a time
#pragma unroll
            for (int r8 = 0; r8 < 16; r8 += 8) {
                float dwv[8][NPB];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const unsigned soff = (unsigned)(n0 + (wo * NOB + ob) * 32 + ((r8 + rr) & 3) + 8 * ((r8 + rr) >> 2)) * plane_bytes;
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb)
                        dwv[rr][pb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, yoff[pb], soff, 0));
                }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = r8 + rr;
                    const int ol = (wo * NOB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const bool ov = n0 + ol < p.N;
                    float dsum = 0.f;
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb)
                        if (ov && valid[pb]) dsum += acc[ob][pb][r] * dwv[rr][pb];
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);  // within each 32-lane half
                    if (j == 0 && ov) atomicAdd(&red[ol], dsum);
                }
            }
        }
        if (p.splits > 1) {  // split-K: fp32 atomics through plain pointers
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = n0 + (wo * NOB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.N) continue;
                float* yp = p.y + (int64_t)(b * p.N + o) * out_plane;
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
                    if (valid[pb]) atomicAdd(&yp[pix[pb]], acc[ob][pb][r] * os[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // pass 2: stores only
                const unsigned soff = (unsigned)(n0 + (wo * NOB + ob) * 32 + (r & 3) + 8 * (r >> 2)) * plane_bytes;
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    float v = acc[ob][pb][r] * os[r];
                    if (EPI == EPI_ACT) {  // (os, bs, nz carry the gain)
                        v += bs[r] + nz[pb];
                        v = fmaxf(v, 0.2f * v);
                    }
                    if (EPI == EPI_PRELU) {
                        v += bs[r];
                        v = v > 0.f ? v : sl[EPI == EPI_PRELU ? r : 0] * v;
                    }
                    if (W2E_SKIP(p, 1) && v != 123456.789f) continue;  // tuning aid: no stores, arithmetic kept alive
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, yoff[pb], soff, 0);
                }
            }
        }
    }
    if (EPI == EPI_DOT) {
        __syncthreads();
        if (tid < TN && n0 + tid < p.N) atomicAdd(&p.dot_out[(int64_t)b * p.N + n0 + tid], red[tid]);
    }
#ifdef W2E_TUNING
    if (p.stamps && tid == 0) {
        p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        p.stamps[STAMP_STRIDE * (int64_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


// weight [cout,cin,3,3] -> wp [ceil(K/8)][9][2][N][4]: element (kc, tap, h, n, c) = scale * W(k = 8*kc + 2*c + h, tap', n)
// (zero where k >= K), K/N = cin/cout (transpose=0) or cout/cin (transpose=1), tap' = 8-tap when flip.
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, float scale,
                                 int transpose, int flip) {
    const int Kd = transpose ? cout : cin, Nd = transpose ? cin : cout;
    const int64_t total = (int64_t)((Kd + 7) / 8) * 9 * 2 * Nd * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e & 3);
        const int n = (int)((e >> 2) % Nd);
        const int hh = (int)((e / (4 * (int64_t)Nd)) & 1);
        const int tap = (int)((e / (8 * (int64_t)Nd)) % 9);
        const int kc = (int)(e / (72 * (int64_t)Nd));
        const int k = kc * 8 + 2 * c + hh;
        float v = 0.f;
        if (k < Kd) {
            const int o = transpose ? k : n, i = transpose ? n : k;
            v = scale * w[((int64_t)o * cin + i) * 9 + (flip ? 8 - tap : tap)];
        }
        wp[e] = v;
    }
}

struct TileCfg {
    int nob, npb, wo, wp;
};

template <int MODE, int EPI, int NOB, int NPB, int WO, int WP, int KC, int DMA = 0>
static void launch_cfg(const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {  // dynamic LDS above 64 KB is opt-in per kernel and per device (gfx950: 160 KB per CU)
        static unsigned done = 0;
        big_lds_once((const void*)modconv_kernel<MODE, EPI, NOB, NPB, WO, WP, KC, DMA>, &done);
    }
    modconv_kernel<MODE, EPI, NOB, NPB, WO, WP, KC, DMA><<<grid, 64 * WO * WP, lds, s>>>(p);
}

// The LDS-DMA pipeline is instantiated for the tiles the high-resolution layers use; other tiles keep the register pipeline.
static bool dma_has_cfg(bool all_phase, int cfg) {
    return all_phase ? (cfg == 0 || cfg == 1 || cfg == 2 || cfg == 8 || cfg == 11)
                     : (cfg == 0 || cfg == 1 || cfg == 2 || cfg == 8 || cfg == 9 || cfg == 10);
}

// opt-in bf16x3 form: the DOWN tile
template <int EPI>
static bool launch_x3_down(int cfg, const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    if (cfg != 9) return false;
    launch_cfg<W2E_CONV_DOWN, EPI, 2, 2, 2, 4, 8, 2>(p, grid, lds, s);
    return true;
}

// opt-in bf16x3 form: the all-phase UP tiles
static bool launch_x3_up(int cfg, const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    switch (cfg) {
        case 0: launch_cfg<CONV_UPALL, EPI_PLAIN, 2, 4, 2, 4, 8, 2>(p, grid, lds, s); return true;
        case 1: launch_cfg<CONV_UPALL, EPI_PLAIN, 2, 4, 1, 8, 8, 2>(p, grid, lds, s); return true;
        case 2: launch_cfg<CONV_UPALL, EPI_PLAIN, 1, 4, 1, 8, 8, 2>(p, grid, lds, s); return true;
        case 11: launch_cfg<CONV_UPALL, EPI_PLAIN, 1, 8, 1, 8, 8, 2>(p, grid, lds, s); return true;
    }
    return false;
}

// opt-in bf16x3 form (W2E_CONV_PRECISION=bf16x3): the SAME-mode tiles of the DMA pipeline
template <int EPI>
static bool launch_x3(int cfg, const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    switch (cfg) {
        case 0: launch_cfg<W2E_CONV_SAME, EPI, 2, 4, 2, 4, 8, 2>(p, grid, lds, s); return true;
        case 1: launch_cfg<W2E_CONV_SAME, EPI, 2, 4, 1, 8, 8, 2>(p, grid, lds, s); return true;
        case 2: launch_cfg<W2E_CONV_SAME, EPI, 1, 4, 1, 8, 8, 2>(p, grid, lds, s); return true;
        case 8: launch_cfg<W2E_CONV_SAME, EPI, 1, 4, 1, 4, 8, 2>(p, grid, lds, s); return true;
    }
    return false;
}

template <int MODE, int EPI, int KC>
static bool launch_mode_dma(int cfg, const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    if constexpr (MODE == CONV_UPALL) {
        switch (cfg) {
            case 0: launch_cfg<MODE, EPI, 2, 4, 2, 4, KC, 1>(p, grid, lds, s); return true;
            case 1: launch_cfg<MODE, EPI, 2, 4, 1, 8, KC, 1>(p, grid, lds, s); return true;
            case 2: launch_cfg<MODE, EPI, 1, 4, 1, 8, KC, 1>(p, grid, lds, s); return true;
            case 8: launch_cfg<MODE, EPI, 1, 4, 1, 4, KC, 1>(p, grid, lds, s); return true;
            case 11: launch_cfg<MODE, EPI, 1, 8, 1, 8, KC, 1>(p, grid, lds, s); return true;
        }
        return false;
    } else {
        switch (cfg) {
            case 0: launch_cfg<MODE, EPI, 2, 4, 2, 4, KC, 1>(p, grid, lds, s); return true;
            case 1: launch_cfg<MODE, EPI, 2, 4, 1, 8, KC, 1>(p, grid, lds, s); return true;
            case 2: launch_cfg<MODE, EPI, 1, 4, 1, 8, KC, 1>(p, grid, lds, s); return true;
            case 8: launch_cfg<MODE, EPI, 1, 4, 1, 4, KC, 1>(p, grid, lds, s); return true;
            case 9: launch_cfg<MODE, EPI, 2, 2, 2, 4, KC, 1>(p, grid, lds, s); return true;
            case 10: launch_cfg<MODE, EPI, 2, 2, 1, 8, KC, 1>(p, grid, lds, s); return true;
        }
        return false;
    }
}

template <int MODE, int EPI, int KC>
static bool launch_mode(int cfg, const ConvParams& p, int grid, size_t lds, hipStream_t s) {
    if constexpr (MODE == CONV_UPALL) {  // accumulator columns = 4 phases x NPB/4 pixel blocks
        switch (cfg) {
            case 0: launch_cfg<MODE, EPI, 2, 4, 2, 4, KC>(p, grid, lds, s); return true;
            case 1: launch_cfg<MODE, EPI, 2, 4, 1, 8, KC>(p, grid, lds, s); return true;
            case 2: launch_cfg<MODE, EPI, 1, 4, 1, 8, KC>(p, grid, lds, s); return true;
            case 4: launch_cfg<MODE, EPI, 1, 4, 2, 2, KC>(p, grid, lds, s); return true;
            case 8: launch_cfg<MODE, EPI, 1, 4, 1, 4, KC>(p, grid, lds, s); return true;
            case 11: launch_cfg<MODE, EPI, 1, 8, 1, 8, KC>(p, grid, lds, s); return true;
        }
        return false;
    } else {
        switch (cfg) {
            case 0: launch_cfg<MODE, EPI, 2, 4, 2, 4, KC>(p, grid, lds, s); return true;
            case 1: launch_cfg<MODE, EPI, 2, 4, 1, 8, KC>(p, grid, lds, s); return true;
            case 2: launch_cfg<MODE, EPI, 1, 4, 1, 8, KC>(p, grid, lds, s); return true;
            case 3: launch_cfg<MODE, EPI, 2, 2, 2, 2, KC>(p, grid, lds, s); return true;
            case 4: launch_cfg<MODE, EPI, 1, 4, 2, 2, KC>(p, grid, lds, s); return true;
            case 5: launch_cfg<MODE, EPI, 1, 2, 2, 2, KC>(p, grid, lds, s); return true;
            case 6: launch_cfg<MODE, EPI, 1, 1, 2, 2, KC>(p, grid, lds, s); return true;
            case 7: launch_cfg<MODE, EPI, 1, 1, 4, 1, KC>(p, grid, lds, s); return true;
            case 8: launch_cfg<MODE, EPI, 1, 4, 1, 4, KC>(p, grid, lds, s); return true;
            case 9: launch_cfg<MODE, EPI, 2, 2, 2, 4, KC>(p, grid, lds, s); return true;
            case 10: launch_cfg<MODE, EPI, 2, 2, 1, 8, KC>(p, grid, lds, s); return true;
        }
        return false;
    }
}

static const TileCfg kCfgStd[] = {{2, 4, 2, 4}, {2, 4, 1, 8}, {1, 4, 1, 8},               // 512 threads, 1 workgroup / CU
                                   {2, 2, 2, 2}, {1, 4, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2}, {1, 1, 4, 1}, {1, 4, 1, 4},  // 256 threads
                                   {2, 2, 2, 4}, {2, 2, 1, 8},  // 512 threads, 4 accumulators per wave (register headroom)
                                   {1, 8, 1, 8}};               // all-phase UP only: 32 channels x (4 phases x 2 pixel blocks)
static const int kNumCfg = 11;  // configurations of the per-phase / SAME / DOWN kernels
static const int kNumCfgAll = 12;

static int next_pow2(int v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_conv_pack(const float* weight, float* wp, int cout, int cin, float scale, int transpose, int flip,
                             void* stream) {
    W2E_REQUIRE(weight && wp, "conv_pack: null tensor");
    W2E_REQUIRE(cout > 0 && cin > 0, "conv_pack: bad dims");
    const int64_t total = (int64_t)(((transpose ? cout : cin) + 7) / 8) * 72 * (transpose ? cin : cout) * 4;
    conv_pack_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(weight, wp, cout, cin, scale, transpose,
                                                                             flip);
    W2E_LAUNCH_CHECK("conv_pack");
    return 0;
}

// The engine behind w2e_modconv3x3 (StyleGAN2 layers) and w2e_conv3x3 (plain convolutions with folded BatchNorm / PReLU:
// IR-SE50, the e4e encoder).  prelu: epilogue v = prelu(out_scale*acc + bias[o], slope[o]) (SAME / DOWN, no split-K);
// down_pad: DOWN reads an (2h) x (2w) image with a one-pixel zero border on the top/left (stride 2, padding 1).
static int conv_impl(int mode, const float* x, const float* wp, const float* in_scale, const float* out_scale, float* y,
                     int batch, int k_ch, int n_ch, int h, int w, int act, const float* noise, const float* noise_w,
                     const float* bias, const float* dot_with, float* dot_out, const float* slope, int prelu, int down_pad,
                     void* stream) {
    W2E_REQUIRE(mode >= 0 && mode <= 2, "modconv3x3: bad mode %d", mode);
    W2E_REQUIRE(x && wp && y, "modconv3x3: null tensor");
    W2E_REQUIRE(batch >= 0 && k_ch > 0 && n_ch > 0 && h > 0 && w > 0, "modconv3x3: bad dims");
    W2E_REQUIRE(!(act && mode != W2E_CONV_SAME), "modconv3x3: fused activation only in SAME mode");
    W2E_REQUIRE(!(prelu && (act || dot_with || mode == W2E_CONV_UP)), "conv3x3: the bias/PReLU epilogue is for SAME and DOWN, alone");
    W2E_REQUIRE(!(down_pad && mode != W2E_CONV_DOWN), "conv3x3: down_pad is a DOWN-mode option");
    W2E_REQUIRE(!(dot_with && (act || mode == W2E_CONV_UP)), "modconv3x3: dot epilogue only without act, not in UP mode");
    W2E_REQUIRE((dot_with == nullptr) == (dot_out == nullptr), "modconv3x3: dot_with and dot_out go together");
    W2E_REQUIRE(!noise || noise_w, "modconv3x3: noise without noise_w");
    if (batch == 0) return 0;
    {   // the kernel addresses one image's input with 32-bit byte offsets (buffer loads)
        const int64_t ih = mode == W2E_CONV_DOWN ? 2 * (int64_t)h + 1 - down_pad : h, iw = mode == W2E_CONV_DOWN ? 2 * (int64_t)w + 1 - down_pad : w;
        W2E_REQUIRE(((int64_t)k_ch + (mode == W2E_CONV_UP ? 8 : 0)) * ih * iw * 4 < ((int64_t)1 << 32), "modconv3x3: one image of the input exceeds 4 GB");
        const int64_t oplane = mode == W2E_CONV_UP ? 4 * ((int64_t)h + 1) * W2E_PLANAR_PITCH(w) : (int64_t)h * w;
        W2E_REQUIRE((int64_t)n_ch * oplane * 4 < ((int64_t)1 << 32) - 16 * oplane, "modconv3x3: one image of the output exceeds 4 GB");
    }
    hipStream_t s = (hipStream_t)stream;

    ConvParams p{};
    p.x = x, p.wp = wp, p.in_scale = in_scale, p.out_scale = out_scale, p.y = y;
    p.noise = noise, p.noise_w = noise_w, p.bias = bias, p.dot_with = dot_with, p.dot_out = dot_out;
    p.batch = batch, p.K = k_ch, p.N = n_ch, p.H = h, p.W = w;
    if (mode == W2E_CONV_SAME) p.in_h = h, p.in_w = w, p.out_h = h, p.out_w = w;
    else if (mode == W2E_CONV_UP) p.in_h = h, p.in_w = w, p.out_h = 2 * h + 1, p.out_w = 2 * w + 1;
    else p.in_h = 2 * h + 1 - down_pad, p.in_w = 2 * w + 1 - down_pad, p.out_h = h, p.out_w = w, p.in_off = down_pad ? -1 : 0;
    p.slope = slope;

    // ---- pick the tile configuration with a small cost model.  Unit = MFMA cycles on one SIMD.  A CU works
    // through ceil(wgs/256) workgroups (two resident 256-thread workgroups share its matrix pipes, so they
    // do not go faster than one after the other); a workgroup's time is its per-SIMD MFMA chain plus the
    // part of its staging (barriers, LDS writes) that the register prefetch cannot hide.
    const bool up = mode == W2E_CONV_UP;
    const Options& opt = options();
    const TileCfg* cfgs = kCfgStd;
    const int ncfg = kNumCfg;
    const int kc = 8;  // channels per K-chunk = one float4 group per lane-half
    const int kc_max = up ? 16 : kc;  // deepest chunk any workgroup of this launch uses
    const int wp2 = next_pow2(w);
    int best = -1, best_splits = 1;
    double best_cost = 0.0;
    // no split-K in deterministic mode (no fp32 atomics): the tile is then chosen among the unsplit candidates, not chosen for a
    // split and stripped of it afterwards
    // (a split bias/PReLU launch is followed by one elementwise pass, like the split activation: worth it only where even the
    // smallest tile, 64 channels x 64 pixels, leaves CUs without a workgroup -- measured on the IR-SE50 / style-head shapes)
    int sp_max = opt.deterministic ? 1 : 32;
    if (prelu) {
        const int tw6 = wp2 < 32 ? wp2 : 32, th6 = 64 / tw6 > 0 ? 64 / tw6 : 1;
        if ((int64_t)batch * ceil_div(n_ch, 64) * ceil_div(h, th6) * ceil_div(w, tw6) >= 256) sp_max = 1;
    }
    for (int c = 0; c < ncfg; ++c) {
        const int tn = 32 * cfgs[c].nob * cfgs[c].wo, tm = 32 * cfgs[c].npb * cfgs[c].wp;
        const int tw = wp2 < 32 ? wp2 : ((wp2 >= 64 && tm >= 256) ? 64 : 32);
        const int th = tm / tw;
        if (th < 1) continue;
        const int ph = mode == W2E_CONV_SAME ? th + 2 : (up ? th + 1 : 2 * th + 1);
        const int pw = mode == W2E_CONV_SAME ? tw + 2 : (up ? tw + 1 : 2 * tw + 1);
        const int nt = 64 * cfgs[c].wo * cfgs[c].wp;
        // deepest K-chunk of this tile (UP: light phases go 16 deep unless the register budget forbids it)
        const int kdeep = (up && cfgs[c].nob * cfgs[c].npb < 8 && max_patch_slots(mode, tm, nt) <= 2) ? 16 : kc;
        const size_t lds_c = up ? sizeof(float) * ((size_t)32 * tn + (size_t)kdeep * ph * pw)
                                : sizeof(float) * ((size_t)kc * 9 * tn + (size_t)kc * ph * pw);
        if (lds_c > 150 * 1024) continue;
        if (mode == W2E_CONV_DOWN && max_patch_slots(mode, tm, nt) > 5) continue;  // prefetch registers: 8 channels x slots
        if (ph * pw > nt * max_patch_slots(mode, tm, nt)) continue;  // register-prefetch slots per thread
        const double tiles = (double)batch * ceil_div(n_ch, tn) * ceil_div(h, th) * ceil_div(w, tw);
        const double waves_per_simd = nt / 256.0;
        const double unit = (double)cfgs[c].nob * cfgs[c].npb * waves_per_simd * (k_ch / 2.0) * 64.0;  // one tap
        const double t_stage0 = (double)ceil_div(k_ch, kc) * 1200.0;
        // split-K (low resolutions: a handful of tiles, each a K*9/2-long dependent MFMA chain): S slices of the
        // channel range per tile, summed with fp32 atomics onto a zeroed output
        for (int sp = 1; sp <= sp_max; sp *= 2) {
            if (sp > 1 && (k_ch / sp < 2 * kc_max || (up ? 4.0 : 1.0) * tiles * (sp / 2) >= 256.0)) break;
            double cost;
            // a lone 256-thread workgroup on a CU has nobody to hide its staging behind (1 wave per SIMD)
            const double t_stage = t_stage0 * ((nt == 256 && (up ? 4.0 : 1.0) * tiles * sp <= 256.0) ? 2.5 : 1.0);
            if (up) {  // 4 phases with 4/2/2/1 taps, heaviest dispatched first
                const double per_cu = ceil(4.0 * tiles * sp / 256.0) / 4.0;  // tile-slices (sets of 4 phases) per CU
                cost = per_cu * ((9.0 * unit + 4.0 * t_stage) / sp + 16000.0) + (4.0 * tiles * sp < 1024.0 ? 1.75 * unit / sp : 0.0);
            } else {
                // + prologue/epilogue per workgroup.  With <= 4 chunks per tile (the K = 32 layers) those fixed costs are a
                // third of a tile, and two co-resident 256-thread workgroups hide part of each other's (measured: 32x512
                // tiles beat 32x1024 by 6 % on the 1024^2 layer, not on the deeper ones)
                const double fixed = (nt == 256 && k_ch <= 4 * kc) ? 0.4 : 1.0;
                cost = ceil(tiles * sp / 256.0) * ((9.0 * unit + fixed * t_stage) / sp + fixed * 4000.0);
            }
            if (sp > 1) {  // memset + (with act) the separate bias/act pass + the fp32 atomics: ~120 cycles per 256-B
                           // wave-instruction per CU (MI355X_MICROARCH.md, global float atomics)
                const double wgs_per_cu = ceil((up ? 4.0 : 1.0) * tiles * sp / 256.0);
                cost = cost * 1.06 + 8000.0 + wgs_per_cu * ((double)tn * tm / 64.0) * 120.0;
            }
            if (best < 0 || cost < best_cost * 0.995) best = c, best_cost = cost, best_splits = sp;
        }
    }
    // UP: the all-phase form (one staged patch, 9 taps, 4 phases x NPB/4 pixel blocks per wave) as the alternative
    bool use_all = false;
    if (up) {
        static const int kAll[] = {0, 1, 2, 4, 8, 11};
        const int tune_all = opt.tune_upall;  // -1 the library's choice, 0 never, 1 always
        int best_a = -1, best_a_splits = 1;
        double best_a_cost = 0.0;
        for (int ci = 0; ci < 6 && tune_all != 0; ++ci) {
            const int c = kAll[ci];
            const int tn = 32 * cfgs[c].nob * cfgs[c].wo, tm = 32 * (cfgs[c].npb / 4) * cfgs[c].wp;
            const int tw = wp2 < 32 ? wp2 : ((wp2 >= 64 && tm >= 256) ? 64 : 32);
            const int th = tm / tw;
            if (th < 1) continue;
            const int ph = th + 1, pw = tw + 1;
            const int nt = 64 * cfgs[c].wo * cfgs[c].wp;
            if (sizeof(float) * ((size_t)kc * 9 * tn + (size_t)kc * ph * pw) > 150 * 1024) continue;
            if (ph * pw > nt * max_patch_slots(W2E_CONV_UP, tm, nt)) continue;
            const double tiles = (double)batch * ceil_div(n_ch, tn) * ceil_div(h, th) * ceil_div(w, tw);
            const double unit = (double)cfgs[c].nob * (cfgs[c].npb / 4) * (nt / 256.0) * (k_ch / 2.0) * 64.0;  // one tap
            const double t_stage = (double)ceil_div(k_ch, kc) * 1200.0;
            for (int sp = 1; sp <= sp_max; sp *= 2) {
                if (sp > 1 && (k_ch / sp < 2 * kc || tiles * (sp / 2) >= 256.0)) break;
                // a split adds the memset and the fp32 atomics: ~120 cycles per 256-B wave-instruction per CU
                const double atomics = sp > 1 ? (double)tn * tm * 4.0 / 64.0 * 120.0 : 0.0;
                double cost = ceil(tiles * sp / 256.0) * ((9.0 * unit + t_stage) / sp + 6000.0 + atomics);  // 4 output planes per tile
                if (sp > 1) cost = cost * 1.06 + 8000.0;
                if (c == 0) cost *= 1.03;  // measured: the 64x256 tile (1) beats 128x128 (0) by 2-3 % on the 64^2 / 128^2 layers
                if (best_a < 0 || cost < best_a_cost * 0.995) best_a = c, best_a_cost = cost, best_a_splits = sp;
            }
        }
        // measured (tools/layer_bench.py, batch 4): the all-phase form wins from 16x16 inputs up (10-28 %), the per-phase
        // form below (a handful of tiles: it has 4x the workgroups to spread over the chip)
        if (best_a >= 0 && (tune_all == 1 || best < 0 || (tune_all < 0 && (int64_t)h * w >= 256))) {
            use_all = true, best = best_a, best_splits = best_a_splits, best_cost = best_a_cost;
        }
    }
    // conv_precision = bf16x3 (option; opt-in, default = exact fp32 MFMA): the SAME, all-phase UP and DOWN tiles
    // of the DMA pipeline compute each fp32 product as three bf16 products (see DMA == 2 in the kernel)
    const int tune_x3 = prelu ? 0 : opt.conv_precision;  // (the bias/PReLU epilogue is instantiated for the fp32 register pipeline)
    // DOWN in that mode: the 128x256 tile (9) is the one whose stride-2 patch fits beside the bf16 operand images in LDS (156 KB)
    if (tune_x3 == 1 && mode == W2E_CONV_DOWN && (int64_t)h * w >= 4096 && n_ch >= 128 && best >= 0) best = 9, best_splits = 1;  // (N = 64 fills half the tile: slower than fp32)
#ifdef W2E_TUNING
    p.tune_skip = opt.tune_skip;
#endif
    if (opt.tune_cfg >= 0) {  // tests / tools/layer_bench.py: "<cfg>[,<splits>[,<mode>]]", third field: only launches of that mode
        const int fc = opt.tune_cfg, fs = opt.tune_cfg_splits, fm = opt.tune_cfg_mode;
        if (fc < (use_all ? kNumCfgAll : ncfg) && (fm < 0 || fm == mode)) best = fc, best_splits = fs > 0 ? fs : 1;
    }
    if (opt.deterministic) best_splits = 1;  // no fp32 atomics onto y: one workgroup owns every output element
    if (opt.tune_print) fprintf(stderr, "modconv mode %d%s K %d N %d %dx%d B %d -> cfg %d splits %d\n", mode, use_all ? " (all-phase)" : "", k_ch, n_ch, h, w, batch, best, best_splits);
    W2E_REQUIRE(best >= 0, "modconv3x3: no tile configuration for N=%d H=%d W=%d", n_ch, h, w);
    const TileCfg cfg = cfgs[best];
    W2E_REQUIRE(!use_all || cfg.npb >= 4, "modconv3x3: tile %d has no all-phase form (forced by tune_cfg)", best);  // (a forced tile: 0 pixel blocks per phase)
    const int tn = 32 * cfg.nob * cfg.wo, tm = 32 * (use_all ? cfg.npb / 4 : cfg.npb) * cfg.wp;
    p.tw = wp2 < 32 ? wp2 : ((wp2 >= 64 && tm >= 256) ? 64 : 32);
    p.th = tm / p.tw;
    p.tw_log2 = 0;
    while ((1 << p.tw_log2) < p.tw) ++p.tw_log2;
    p.tiles_x = (int)ceil_div(w, p.tw), p.tiles_y = (int)ceil_div(h, p.th), p.tiles_n = (int)ceil_div(n_ch, tn);
    if (mode == W2E_CONV_SAME) p.ph = p.th + 2, p.pw = p.tw + 2;
    else if (up) p.ph = p.th + 1, p.pw = p.tw + 1;
    else p.ph = 2 * p.th + 1, p.pw = 2 * p.tw + 1;
    p.plane = p.ph * p.pw;
    p.pw_magic = (unsigned)(((uint64_t)1 << 32) / (unsigned)p.pw + 1);
    W2E_REQUIRE(p.plane < 65536, "modconv3x3: patch too large");
    const int nt_best = 64 * cfg.wo * cfg.wp;
    // LDS-DMA pipeline (two LDS stages + the in_scale table): where it is instantiated and fits
    // W2E_TUNE_DMA: 0 never, 1 wherever instantiated; default = where it measured faster (tools/layer_bench.py, batch 4):
    // the 512-thread 8-accumulator SAME tiles (+1.5-3 %) and the all-phase UP tiles 0 / 1 / 11 (+2-5 %; since the DMA is issued piecewise
    // through inline asm also at K < 256); not DOWN (-1-2 %), not the
    // 256-thread 32x512 tile of the 1024^2 layer (two LDS stages leave room for 2 instead of 3 workgroups per CU: -11 %)
    const int tune_dma = opt.tune_dma;
    bool use_dma = false;
    size_t lds_dma = 0;
    const bool dma_auto = prelu ? false : use_all ? (best == 0 || best == 1 || best == 11) : (mode == W2E_CONV_SAME && best <= 2);
    if (!prelu && (tune_dma == 1 || (tune_dma < 0 && dma_auto)) && !(up && !use_all) && dma_has_cfg(use_all, best)) {
        const int plane16 = (p.plane + 15) & ~15;  // whole DMA wave-instructions (16 pixels x 4 channels) per plane
        lds_dma = sizeof(float) * (2 * ((size_t)kc * 9 * tn + (size_t)kc * plane16) + (size_t)((k_ch + 7) / 8) * 8);
        const int slots = (int)ceil_div(4 * plane16, nt_best);
        // (the pipeline addresses channels up to K+7 of an image with 32-bit byte offsets: they must not wrap)
        const bool off_ok = ((int64_t)k_ch + 8) * p.in_h * p.in_w * 4 < ((int64_t)1 << 32);
        if (off_ok && lds_dma <= 150 * 1024 && slots <= 4 * max_patch_slots(up ? W2E_CONV_UP : mode, tm, nt_best) + 1) use_dma = true, p.plane = plane16;
        if (opt.tune_print) fprintf(stderr, "  lds-dma pipeline: %s (%zu B LDS, %d slots)\n", use_dma ? "yes" : "no", lds_dma, slots);
    }
    const int kdeep_best = (up && !use_all && cfg.nob * cfg.npb < 8 && max_patch_slots(mode, tm, nt_best) <= 2) ? 16 : kc;
    size_t lds = (up && !use_all) ? sizeof(float) * ((size_t)32 * tn + (size_t)kdeep_best * p.plane)
                                  : sizeof(float) * ((size_t)kc * 9 * tn + (size_t)kc * p.plane);
    bool use_x3 = false;
    if (tune_x3 == 1 && !use_dma && mode == W2E_CONV_DOWN && best == 9) {  // DOWN: not a DMA tile by default
        const int plane16 = (p.plane + 15) & ~15;
        const int slots = (int)ceil_div(4 * plane16, nt_best);
        const bool off_ok = ((int64_t)k_ch + 8) * p.in_h * p.in_w * 4 < ((int64_t)1 << 32);
        if (off_ok && slots <= 4 * max_patch_slots(mode, tm, nt_best) + 1) use_dma = true, p.plane = plane16;
    }
    if (tune_x3 == 1 && !use_dma && mode == W2E_CONV_SAME && best == 8 && dma_has_cfg(false, best)) {  // the 32x512 tile: not a DMA tile by default
        const int plane16 = (p.plane + 15) & ~15;
        const int slots = (int)ceil_div(4 * plane16, nt_best);
        const bool off_ok = ((int64_t)k_ch + 8) * p.in_h * p.in_w * 4 < ((int64_t)1 << 32);
        if (off_ok && slots <= 4 * max_patch_slots(mode, tm, nt_best) + 1) use_dma = true, p.plane = plane16;
    }
    if (tune_x3 == 1 && !use_dma && use_all && (best <= 2 || best == 11)) {  // all-phase UP tiles that are not DMA tiles by default
        const int plane16 = (p.plane + 15) & ~15;
        const int slots = (int)ceil_div(4 * plane16, nt_best);
        const bool off_ok = ((int64_t)k_ch + 8) * p.in_h * p.in_w * 4 < ((int64_t)1 << 32);
        if (off_ok && slots <= 4 * max_patch_slots(W2E_CONV_UP, tm, nt_best) + 1) use_dma = true, p.plane = plane16;
    }
    if (tune_x3 == 1 && use_dma && ((mode == W2E_CONV_SAME && (best <= 2 || best == 8)) || (use_all && (best <= 2 || best == 11)) ||
                                    (mode == W2E_CONV_DOWN && best == 9))) {
        // fp32 staging area + bf16 operand images (20*tn + 2*plane 16-byte entries) + the in_scale table
        lds_dma = sizeof(float) * (((size_t)kc * 9 * tn + (size_t)kc * p.plane) + 4 * ((size_t)20 * tn + 2 * (size_t)p.plane) + (size_t)((k_ch + 7) / 8) * 8);
        use_x3 = lds_dma <= 160 * 1024;
        if (!use_x3 && mode == W2E_CONV_DOWN) use_dma = false;  // (DOWN takes the DMA pipeline only for this mode)
        if (opt.tune_print) fprintf(stderr, "  bf16x3: %s (%zu B LDS)\n", use_x3 ? "yes" : "no", lds_dma);
    }
    if (use_dma) lds = lds_dma;
    W2E_REQUIRE(lds <= 160 * 1024, "modconv3x3: tile needs %zu B of LDS", lds);
    const int k_gran = use_all ? kc : kc_max;
    p.k_per = (int)(ceil_div(ceil_div(k_ch, best_splits), k_gran) * k_gran);
    p.splits = (int)ceil_div(k_ch, p.k_per);
    if (up) {
        // border units (one wave each): image x 32-channel block x 32-position block of the last row (W+1 positions) / column (H)
        p.groups_row = (int)ceil_div(w + 1, 32), p.groups_col = (int)ceil_div(h, 32);
        const int64_t border = (int64_t)batch * (p.groups_row + p.groups_col) * ceil_div(n_ch, 32);  // one workgroup per unit
        W2E_REQUIRE(border < ((int64_t)1 << 31), "modconv3x3: %lld border units are too many", (long long)border);
        p.border_wgs = (int)border;
        if (lds < (size_t)(nt_best / 64) * 32 * 64 * sizeof(float)) lds = (size_t)(nt_best / 64) * 32 * 64 * sizeof(float);  // their join buffer
    }
    const int64_t grid = (int64_t)p.tiles_x * p.tiles_y * p.tiles_n * batch * ((up && !use_all) ? 4 : 1) * p.splits + p.border_wgs;
    W2E_REQUIRE(grid < ((int64_t)1 << 31), "modconv3x3: grid of %lld workgroups is too large", (long long)grid);  // (cast to int at every launch below)
#ifdef W2E_TUNING
    // tuning aid: W2E_TUNE_CLOCK=1 stamps every workgroup and reports the in-kernel shader clock (s_memtime ticks per
    // 100 MHz s_memrealtime tick) of every 64th launch -- the DVFS-limited clock is what an MFMA-bound kernel is priced by
    const bool tune_clock = opt.tune_clock != 0;
    static unsigned long long* stamp_buf = nullptr;
    static int64_t stamp_cap = 0, stamp_calls = 0;
    bool stamped = false;
    if (tune_clock && (stamp_calls++ & 63) == 63) {
        if (grid > stamp_cap) {
            if (stamp_buf) (void)hipFree(stamp_buf);
            stamp_cap = grid, stamp_buf = nullptr;
            if (hipMalloc((void**)&stamp_buf, sizeof(unsigned long long) * STAMP_STRIDE * (size_t)stamp_cap) != hipSuccess) stamp_buf = nullptr, stamp_cap = 0;
        }
        if (stamp_buf && zero_async(stamp_buf, sizeof(unsigned long long) * STAMP_STRIDE * (size_t)grid, s) == hipSuccess) p.stamps = stamp_buf, stamped = true;
    }
#endif
    if (p.splits > 1 &&
        zero_async(y, sizeof(float) * (size_t)batch * n_ch * (up ? 4 * (h + 1) * W2E_PLANAR_PITCH(w) : p.out_h * p.out_w), s) != hipSuccess) {
        set_error("modconv3x3: memset failed");
        return 2;
    }
    bool ok = false;
    if (use_x3 && use_all) {
        ok = launch_x3_up(best, p, (int)grid, lds, s);
    } else if (use_x3 && mode == W2E_CONV_DOWN) {
        if (dot_with) ok = launch_x3_down<EPI_DOT>(best, p, (int)grid, lds, s);
        else ok = launch_x3_down<EPI_PLAIN>(best, p, (int)grid, lds, s);
    } else if (use_x3) {
        if (act && p.splits == 1) ok = launch_x3<EPI_ACT>(best, p, (int)grid, lds, s);
        else if (dot_with) ok = launch_x3<EPI_DOT>(best, p, (int)grid, lds, s);
        else ok = launch_x3<EPI_PLAIN>(best, p, (int)grid, lds, s);
    } else if (use_dma) {
        if (mode == W2E_CONV_SAME) {
            if (act && p.splits == 1) ok = launch_mode_dma<W2E_CONV_SAME, EPI_ACT, 8>(best, p, (int)grid, lds, s);
            else if (dot_with) ok = launch_mode_dma<W2E_CONV_SAME, EPI_DOT, 8>(best, p, (int)grid, lds, s);
            else ok = launch_mode_dma<W2E_CONV_SAME, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
        } else if (up) {
            ok = launch_mode_dma<CONV_UPALL, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
        } else {
            if (dot_with) ok = launch_mode_dma<W2E_CONV_DOWN, EPI_DOT, 8>(best, p, (int)grid, lds, s);
            else ok = launch_mode_dma<W2E_CONV_DOWN, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
        }
    } else if (prelu && p.splits == 1) {
        if (mode == W2E_CONV_SAME) ok = launch_mode<W2E_CONV_SAME, EPI_PRELU, 8>(best, p, (int)grid, lds, s);
        else ok = launch_mode<W2E_CONV_DOWN, EPI_PRELU, 8>(best, p, (int)grid, lds, s);
    } else if (mode == W2E_CONV_SAME) {
        if (act && p.splits == 1) ok = launch_mode<W2E_CONV_SAME, EPI_ACT, 8>(best, p, (int)grid, lds, s);
        else if (dot_with) ok = launch_mode<W2E_CONV_SAME, EPI_DOT, 8>(best, p, (int)grid, lds, s);
        else ok = launch_mode<W2E_CONV_SAME, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
    } else if (up) {
        if (use_all) ok = launch_mode<CONV_UPALL, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
        else ok = launch_mode<W2E_CONV_UP, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
    } else {
        if (dot_with) ok = launch_mode<W2E_CONV_DOWN, EPI_DOT, 8>(best, p, (int)grid, lds, s);
        else ok = launch_mode<W2E_CONV_DOWN, EPI_PLAIN, 8>(best, p, (int)grid, lds, s);
    }
    W2E_REQUIRE(ok, "modconv3x3: internal: configuration %d not instantiated", best);
    W2E_LAUNCH_CHECK("modconv3x3");
#ifdef W2E_TUNING
    if (stamped) {
        unsigned long long* hb = (unsigned long long*)malloc(sizeof(unsigned long long) * STAMP_STRIDE * (size_t)grid);
        if (hb && hipStreamSynchronize(s) == hipSuccess &&
            hipMemcpy(hb, stamp_buf, sizeof(unsigned long long) * STAMP_STRIDE * (size_t)grid, hipMemcpyDeviceToHost) == hipSuccess) {
            double ticks = 0.0, real = 0.0;
            unsigned long long first = ~0ull, last = 0;
            int64_t n = 0;
            double ph[4] = {0.0, 0.0, 0.0, 0.0};  // diagnostic build: vmcnt wait, barrier wait, K loop, DMA issue (wave 0)
            for (int64_t g0 = p.border_wgs; g0 < grid; ++g0) {
                const unsigned long long* e = hb + STAMP_STRIDE * g0;
                if (!e[1] || e[3] <= e[1]) continue;
                ticks += (double)(e[2] - e[0]), real += (double)(e[3] - e[1]), ++n;
                if (e[1] < first) first = e[1];
                if (e[3] > last) last = e[3];
                for (int q = 4; q < STAMP_STRIDE; ++q) ph[q - 4] += (double)e[q];
            }
            if (n && STAMP_STRIDE == 8 && use_dma)
                fprintf(stderr, "modconv phases (wave 0, mean cycles per workgroup): K loop %.0f, of it vmcnt(0) wait %.0f, barrier %.0f, DMA issue %.0f\n",
                        ph[2] / n, ph[0] / n, ph[1] / n, ph[3] / n);
            if (n) fprintf(stderr, "modconv clock: mode %d K %d N %d %dx%d cfg %d%s: %.3f GHz in-kernel (%lld workgroups, mean %.0f cycles each, launch span %.1f us)\n",
                           mode, k_ch, n_ch, h, w, best, use_dma ? " dma" : "", ticks / real * 0.1, (long long)n, ticks / n, (double)(last - first) * 0.01);
        }
        free(hb);
    }
#endif
    if (act && p.splits > 1) {  // the activation needs the complete sum: one in-place elementwise pass
        const int rc = w2e_bias_act_fwd(y, bias, noise, noise_w, y, batch, n_ch, (int64_t)h * w, 0.2f, 1.4142135623730951f, stream);
        if (rc != 0) return rc;
    }
    if (prelu && p.splits > 1) {  // likewise bias + PReLU (the low-resolution layers of IR-SE50 and of the encoders' style heads: a
                                  // handful of tiles with a K*9/2-long MFMA chain each unless K is split)
        const int rc = w2e_affine_act_fwd(y, nullptr, bias, slope, y, batch, n_ch, (int64_t)p.out_h * p.out_w, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

// The caller allocates the phase-planar output of UP: its row pitch must be THIS library's (round 3's e7 memory fault: a caller built
// against ABI 2's pitch handed a buffer 15 % too small to kernels that index with W2E_PLANAR_PITCH).
#define W2E_CHECK_UP_PITCH(name)                                                                                                       \
    W2E_REQUIRE(mode != W2E_CONV_UP || y_pitch == W2E_PLANAR_PITCH(w),                                                                 \
                name ": UP output with a row pitch of %d floats, this library's layout has %d (W2E_PLANAR_PITCH, ABI %d): rebuild the caller", \
                y_pitch, W2E_PLANAR_PITCH(w), W2E_VERSION)

extern "C" int w2e_modconv3x3(int mode, const float* x, const float* wp, const float* in_scale, const float* out_scale,
                              float* y, int batch, int k_ch, int n_ch, int h, int w, int y_pitch, int act, const float* noise,
                              const float* noise_w, const float* bias, const float* dot_with, float* dot_out,
                              void* stream) {
    W2E_CHECK_UP_PITCH("modconv3x3");
    return conv_impl(mode, x, wp, in_scale, out_scale, y, batch, k_ch, n_ch, h, w, act, noise, noise_w, bias, dot_with, dot_out,
                     nullptr, 0, 0, stream);
}

extern "C" int w2e_conv3x3(int mode, const float* x, const float* wp, const float* in_scale, const float* out_scale, float* y,
                           int batch, int k_ch, int n_ch, int h, int w, int y_pitch, int down_pad, const float* bias, const float* slope,
                           void* stream) {
    W2E_CHECK_UP_PITCH("conv3x3");
    const int prelu = (bias || slope) ? 1 : 0;
    W2E_REQUIRE(!(prelu && mode == W2E_CONV_UP), "conv3x3: no bias / PReLU epilogue in UP mode");
    return conv_impl(mode, x, wp, in_scale, out_scale, y, batch, k_ch, n_ch, h, w, 0, nullptr, nullptr, bias, nullptr, nullptr,
                     slope, prelu, down_pad, stream);
}
