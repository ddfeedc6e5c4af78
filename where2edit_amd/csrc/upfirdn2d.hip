// K2: upfirdn2d for gfx950 (replaces models/stylegan2/op/upfirdn2d.py:11-60).
//
// Two kernels:
//   * upfirdn_tile_kernel  -- up=1, down=1 (the 8 Blur calls per forward, the HBM-heavy case:
//     134 MB in + 134 MB out per image at 1024^2).  64x16 output tile per 256-thread block, the
//     (16+kh-1) x (64+kw-1) input tile staged once in LDS, 4 outputs per thread, float4 stores,
//     optional fused demod/noise/bias/LeakyReLU epilogue so the activation is written once.
//   * upfirdn_generic_kernel -- any up/down/pad/flip (RGB skip up-sampling, Downsample, adjoints).
#include <stdlib.h>

#include "common.h"

namespace w2e {

constexpr int MAXK = 16;

// The work-skipping bits of tune_blur (1 = no global loads, 2 = no FIR arithmetic, 4 = no stores: how the kernels were
// taken apart) exist only in a -DW2E_TUNING build; the shipped library cannot be told to skip its own work.
// Bit 8 (keep the tile kernels for images the streaming kernel would take) selects a kernel and is always honoured.
#ifdef W2E_TUNING
#define W2E_BLUR_SKIP(p, bit) (((p).tune & (bit)) != 0)
#else
#define W2E_BLUR_SKIP(p, bit) false
#endif

struct UpfirdnParams {
    const float* x;
    const float* kern;
    float* y;
    int64_t planes;
    int in_h, in_w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0, flip;
    int tune;    // tuning aid (W2E_TUNE_BLUR), -DW2E_TUNING builds only: 1 = no global loads, 2 = no FIR arithmetic, 4 = no stores
    int planar;  // 1: x is phase-planar [planes][2][2][(in_h+1)/2][(in_w+1)/2]
    int act;
    const float* out_scale;
    const float* noise;
    const float* noise_w;
    const float* bias;
    int channels;
    float slope, gain;
    // fused activation backward (upfirdn_stream4_kernel<..., ACTBWD>): x is the gradient w.r.t. the activated output, y_fwd that
    // output; the filtered quantity is x * gain * (y_fwd > 0 ? 1 : slope), and its three per-plane sums are added to sums[plane][3]
    const float* y_fwd;
    float* sums;
};

__device__ __forceinline__ float epilogue(const UpfirdnParams& p, float v, int64_t plane, int oy, int ox, float nw) {
    if (!p.act) return v;
    if (p.out_scale) v *= p.out_scale[plane];
    if (p.noise) v += nw * p.noise[(int64_t)oy * p.out_w + ox];
    if (p.bias) v += p.bias[plane % p.channels];
    return (v > 0.f ? v : v * p.slope) * p.gain;
}

constexpr int TW = 64, TH = 32;  // output tile of a 256-thread block; each thread: 4 wide x 2 tall
constexpr int MAX_TILE_K = 8;    // taps per axis the tile kernel handles (pitch / LDS sized for it)

// up = down = 1.  The (TH+kh-1) x (TW+kw-1) input tile is staged once into LDS (row pitch a multiple of 4
// floats so each thread's window rows are 16-B aligned: ds_read_b128), every thread then slides the taps
// over a register window: 2*(kw+3) LDS floats read per 8 outputs.
// 4x4 FIR (the generator's blur), up = down = 1, unaligned source rows (odd widths): scalar staging; taps in
// registers, everything unrolled, ACT compile-time.  Aligned sources take upfirdn_blur4_kernel below.
template <bool ACT>
__global__ __launch_bounds__(256) void upfirdn_tile4_kernel(UpfirdnParams p, int tiles_x, int tiles_y, unsigned pw_magic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;
    constexpr int PW = TW + 3, PH = TH + 3, PITCH = (PW + 3) & ~3;
    const int tid = threadIdx.x;
    float kr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kr[i] = p.flip ? p.kern[15 - i] : p.kern[i];
    const float nw = (ACT && p.noise) ? p.noise_w[0] : 0.f;
    const int tiles_per_plane = tiles_x * tiles_y;
    const int n_tiles = tiles_per_plane * (int)p.planes;
    const int ly = (tid >> 4) * 2, lx = (tid & 15) * 4;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int plane = t / tiles_per_plane;
        const int rem = t - plane * tiles_per_plane;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const float* src = p.x + (int64_t)plane * p.in_h * p.in_w;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 - p.pad_y0, ix0 = ox0 - p.pad_x0;
        __syncthreads();  // previous iteration's readers are done
#pragma unroll
        for (int it = 0; it < (PH * PW + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int r = (int)__umulhi((unsigned)i, pw_magic), c = i - r * PW;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            // branch-free: load from a clamped (always valid) address, then select
            const int cy = iy < 0 ? 0 : (iy >= p.in_h ? p.in_h - 1 : iy), cx = ix < 0 ? 0 : (ix >= p.in_w ? p.in_w - 1 : ix);
            const float v = W2E_BLUR_SKIP(p, 1) ? 1.f : src[cy * p.in_w + cx];
            if (i < PH * PW) tile[r * PITCH + c] = ok ? v : 0.f;
        }
        __syncthreads();
        float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < (W2E_BLUR_SKIP(p, 2) ? 1 : 5); ++r) {  // input row ly+r feeds output row 0 with ky=r and output row 1 with ky=r-1
            const float4 a = *reinterpret_cast<const float4*>(tile + (ly + r) * PITCH + lx);
            const float4 b = *reinterpret_cast<const float4*>(tile + (ly + r) * PITCH + lx + 4);
            const float win[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (r < 4) acc0[j] += kr[r * 4 + kx] * win[kx + j];
                    if (r >= 1) acc1[j] += kr[(r - 1) * 4 + kx] * win[kx + j];
                }
            }
        }
        const int ox = ox0 + lx;
        float e_scale = 1.f, e_bias = 0.f;
        if (ACT) {
            if (p.out_scale) e_scale = p.out_scale[plane];
            if (p.bias) e_bias = p.bias[plane % p.channels];
        }
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const int oy = oy0 + ly + yy;
            if (oy >= p.out_h || ox >= p.out_w) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float e = yy ? acc1[j] : acc0[j];
                if (ACT) {
                    e = e * e_scale + e_bias;
                    if (p.noise) e += nw * p.noise[oy * p.out_w + (ox + j < p.out_w ? ox + j : ox)];
                    e = (e > 0.f ? e : e * p.slope) * p.gain;
                }
                v[j] = e;
            }
            float* dst = p.y + ((int64_t)plane * p.out_h + oy) * p.out_w + ox;
            if (W2E_BLUR_SKIP(p, 4) && v[0] != 123456.75f) continue;
            if (ox + 3 < p.out_w && (p.out_w & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ox + j < p.out_w) dst[j] = v[j];
            }
        }
    }
}

// 4x4 FIR, up = down = 1, input rows 16-B aligned (plain layout with in_w % 4 == 0, or the phase-planar T'): the two
// HBM-heavy blur shapes of a step.
//  * A workgroup owns one 64-column strip of tiles (fixed tile column) and walks down (plane, tile row) items, so
//    everything that depends on the column -- chunk offsets, edge masks, LDS addresses -- is computed once.
//  * Staging is a pure copy: an aligned global float4 becomes one conflict-free ds_write_b128 into an LDS row that
//    keeps the SOURCE's alignment (and, for the planar layout, its even/odd column split).  The realignment /
//    interleave happens when the threads read their windows: lanes run along x (stride-1 LDS reads, 256-B
//    coalesced stores even for odd output widths), 8 output rows per thread.
//  * The next item's float4s are fetched into registers before the current one is filtered and stored
//    (issue early / write late).
template <bool ACT, bool PLANAR>
__global__ __launch_bounds__(256) void upfirdn_blur4_kernel(UpfirdnParams p, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PW = TW + 3, PH = TH + 3;
    constexpr int NQ = PLANAR ? ((PW + 1) / 2 + 1 + 3) / 4 + 1 : (PW + 3) / 4 + 1;  // float4s per staged row (per parity)
    constexpr int PE = 4 * NQ;
    constexpr int SLOTS = (PLANAR ? 2 : 1) * PH * NQ;
    constexpr int NIT = (SLOTS + 255) / 256;
    static_assert(TW == 64 && TH == 32, "thread mapping: 64 lanes along x, 4 waves x 8 rows");
    const int tid = threadIdx.x;
    float kr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kr[i] = p.flip ? p.kern[15 - i] : p.kern[i];
    // rank-1 test (uniform): k[ky][kx] == kv[ky] * kh[kx] with kh = row 0, kv = column 0 / k[0][0]
    float kh[4], kv[4];
    bool separable = kr[0] != 0.f;
    {
        float kmax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) kmax = fmaxf(kmax, fabsf(kr[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) kh[i] = kr[i], kv[i] = separable ? kr[4 * i] / kr[0] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) separable = separable && fabsf(kr[i] - kv[i >> 2] * kh[i & 3]) <= 1e-7f * kmax;
    }
    const float nw = (ACT && p.noise) ? p.noise_w[0] : 0.f;
    const int hp = (p.in_h + 1) >> 1, wpp = W2E_PLANAR_PITCH((p.in_w - 1) >> 1);  // (in_w = 2W+1)
    const int64_t plane_stride = PLANAR ? (int64_t)4 * hp * wpp : (int64_t)p.in_h * p.in_w;
    const int row_len = PLANAR ? wpp : p.in_w;                       // floats per source row (a multiple of 4)

    const int tx = blockIdx.x % tiles_x, g0 = blockIdx.x / tiles_x, gstep = gridDim.x / tiles_x;
    const int ox0 = tx * TW, ix0 = ox0 - p.pad_x0;
    const int n_items = (int)p.planes * tiles_y;
    // first source column index of chunk 0 of parity par
    auto chunk0 = [&](int par) { return PLANAR ? (((ix0 - par + 1) >> 1) & ~3) : (ix0 & ~3); };

    // per-slot constants: slot -> (staged row r, parity, chunk q)
    int s_r[NIT], s_par[NIT], s_v4[NIT], s_lds[NIT];
    unsigned s_mask[NIT];  // bit k: element k of the chunk is image data (column inside [0, in_w))
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        int r, par, q;
        if (PLANAR) {
            r = i / (2 * NQ);
            const int rem = i - r * (2 * NQ);
            par = rem / NQ, q = rem - par * NQ;
        } else {
            r = i / NQ, par = 0, q = i - r * NQ;
        }
        const int v4 = chunk0(par) + 4 * q;
        const bool col_ok = r < PH && v4 >= 0 && v4 + 3 < row_len;
        unsigned m = 0;
        if (col_ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ix = PLANAR ? 2 * (v4 + k) + par : v4 + k;
                if (ix >= 0 && ix < p.in_w) m |= 1u << k;
            }
        }
        s_r[it] = r;
        s_mask[it] = m;
        s_par[it] = par;
        s_v4[it] = v4;
        s_lds[it] = (par * PH + (r < PH ? r : 0)) * PE + 4 * q;
    }

    // Loads and stores are buffer operations: a descriptor over the item's plane + a per-thread byte offset + the tile
    // row's scalar offset; an invalid slot (row outside the image, chunk outside the row) carries an out-of-range offset
    // and the hardware returns 0 -- no select on the loaded value (which would make the wave wait for the load at the
    // issue site and undo the prefetch), no 64-bit address arithmetic per access.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0x80000000u;
    auto fetch = [&](int item, float4 (&reg)[NIT]) __attribute__((always_inline)) {
        const int plane = item / tiles_y, ty = item - plane * tiles_y;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + (int64_t)plane * plane_stride), (short)0, (int)(plane_stride * 4), 0x00020000);
        const int iy0 = ty * TH - p.pad_y0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int iy = iy0 + s_r[it];
            const bool ok = s_mask[it] != 0 && iy >= 0 && iy < p.in_h && !W2E_BLUR_SKIP(p, 1);
            // element offset of the chunk inside the plane (all of it in the per-thread offset: never negative when ok)
            const int off = PLANAR ? ((((iy & 1) * 2 + s_par[it]) * hp + (iy >> 1)) * wpp + s_v4[it]) : (iy * p.in_w + s_v4[it]);
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)off * 4u : OOB, 0, 0));
            reg[it] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto commit = [&](const float4 (&reg)[NIT]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NIT * 256 > SLOTS && s_r[it] >= PH) continue;
            float4 v = reg[it];
            if (PLANAR) {  // the padded tail of a source row is not image data
                if (!(s_mask[it] & 1u)) v.x = 0.f;
                if (!(s_mask[it] & 2u)) v.y = 0.f;
                if (!(s_mask[it] & 4u)) v.z = 0.f;
                if (!(s_mask[it] & 8u)) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(smem + s_lds[it]) = v;
        }
    };

    // window of the thread: image columns a .. a+3 (a = ix0 + lane), staged rows yb .. yb+10
    const int lane = tid & 63, yb = (tid >> 6) * 8;
    int woff[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
        const int c = ix0 + lane + kx;
        woff[kx] = PLANAR ? ((c & 1) * PH * PE + (c >> 1) - chunk0(c & 1)) : (c - chunk0(0));
    }
    const int ox = ox0 + lane;
    const bool x_ok = ox < p.out_w;

    float4 reg[NIT];
    int item = g0;
    if (item < n_items) fetch(item, reg);
    for (; item < n_items; item += gstep) {
        __syncthreads();  // the previous tile's readers are done
        commit(reg);
        __syncthreads();
        if (item + gstep < n_items) fetch(item + gstep, reg);  // lands while this tile is filtered and stored

        const int plane = item / tiles_y, ty = item - plane * tiles_y;
        const int oy0 = ty * TH + yb;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (separable) {  // k = kv (x) kh (every StyleGAN2 blur is an outer product): 44 + 32 FMAs instead of 128
#pragma unroll
            for (int r = 0; r < 11; ++r) {  // staged row yb+r: horizontal pass once, then it feeds output row j with ky = r - j
                if (W2E_BLUR_SKIP(p, 2) && r > 0) break;
                const float* rowp = smem + (yb + r) * PE;
                const float hr = kh[0] * rowp[woff[0]] + kh[1] * rowp[woff[1]] + kh[2] * rowp[woff[2]] + kh[3] * rowp[woff[3]];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ky = r - j;
                    if (ky >= 0 && ky < 4) acc[j] += kv[ky] * hr;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 11; ++r) {  // staged row yb+r feeds output row j with ky = r - j
                if (W2E_BLUR_SKIP(p, 2) && r > 0) break;
                const float* rowp = smem + (yb + r) * PE;
                const float w0 = rowp[woff[0]], w1 = rowp[woff[1]], w2 = rowp[woff[2]], w3 = rowp[woff[3]];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ky = r - j;
                    if (ky >= 0 && ky < 4)
                        acc[j] += kr[ky * 4 + 0] * w0 + kr[ky * 4 + 1] * w1 + kr[ky * 4 + 2] * w2 + kr[ky * 4 + 3] * w3;
                }
            }
        }
        float e_scale = 1.f, e_bias = 0.f;
        if (ACT) {
            if (p.out_scale) e_scale = p.out_scale[plane];
            if (p.bias) e_bias = p.bias[plane % p.channels];
        }
        // rows past out_h fall past the descriptor (dropped / 0); columns past out_w carry OOB
        const unsigned out_bytes = (unsigned)(p.out_h * p.out_w) * 4u;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)plane * p.out_h * p.out_w, (short)0,
                                                                            (int)out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ACT && p.noise ? p.noise : p.x), (short)0,
                                                                            (ACT && p.noise) ? (int)out_bytes : 0, 0x00020000);
        const unsigned voff0 = x_ok ? (unsigned)(oy0 * p.out_w + ox) * 4u : OOB;
        float nz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            nz[j] = ACT ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rn, voff0 + (unsigned)(j * p.out_w) * 4u, 0, 0)) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e = acc[j];
            if (ACT) {
                e = e * e_scale + e_bias + nw * nz[j];
                e = (e > 0.f ? e : e * p.slope) * p.gain;
            }
            if (W2E_BLUR_SKIP(p, 4) && e != 123456.75f) continue;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, e), ry, voff0 + (unsigned)(j * p.out_w) * 4u, 0, 0);
        }
    }
}

// 4x4 FIR, up = down = 1, wide images (out_w >= 256): a streaming form without LDS or barriers.
//  * A wave owns a 256-column x SR-row strip of one plane; a lane owns 4 adjacent output columns and walks DOWN the strip.
//    Each source row is read once per lane as a 7-wide window, three rows ahead of its use, and the vertical reuse lives in
//    registers: a source row is scattered into the 4 output rows it feeds (separable kernels: one 4-tap horizontal pass, then
//    4 scaled adds); the oldest of the 4 open output rows is then finished, run through the fused epilogue and stored as one
//    float4 per lane (1 KB per wave; row offsets on the scalar unit).  The noise row of the epilogue is fetched with the
//    window that closes its output row, so no wait ever drains the loads in flight.
//  * The window is held as the PAIRS the packed-fp32 FMAs (v_pk_fma_f32) consume, each pair loaded as one 8-byte buffer load:
//    PLANAR (the UP conv's phase-planar T', pad_x0 = 1): outputs are paired (0,2),(1,3); the pairs (w0,w2),(w2,w4),(w4,w6) are
//    consecutive entries of the odd-column plane and (w1,w3),(w3,w5) of the even-column plane.  Dense source (in_w a multiple
//    of 4, pad_x0 = 2: the adjoint blur): outputs are paired (0,1),(2,3) and the pairs are (w_i, w_i+1), i = 0..5.
//    A pair that straddles the image's left / right edge is fixed up by a select in the waves that hold such a lane; whole
//    pairs outside the image, and rows outside it, fall past a descriptor and read as 0.
//  * Dense source with an odd out_w (in_w + 1): the lane owning the last full group also computes the single last column.
constexpr int SR = 64;  // strip rows per wave (3 halo rows re-read per strip: 4.7 %)

//  * ACTBWD (dense source, the StyledConv backward of an up-sampling layer): the source is read together with the layer's
//    forward output and the activation backward gpre = g * gain * (out > 0 ? 1 : slope) is applied to the window on the way
//    (w2e_bias_act_bwd_reduce's arithmetic); the three per-plane sums of that kernel (gpre * pre-activation, gpre * noise, gpre)
//    are accumulated over the rows and columns the lane OWNS (not its halo) and added to sums[plane][3] -- gpre itself is never
//    written: 3 tensor passes instead of 5.
template <bool ACT, bool PLANAR, bool ACTBWD = false>
__global__ __launch_bounds__(256) void upfirdn_stream4_kernel(UpfirdnParams p, int col_groups, int strips) {
    static_assert(!ACTBWD || (!ACT && !PLANAR), "the fused activation backward is the dense adjoint form");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int NP = PLANAR ? 5 : 1;  // 8-byte pairs loaded per window
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int per_plane = col_groups * strips;
    const int plane = wid / per_plane;
    if (plane >= (int)p.planes) return;
    const int rem = wid - plane * per_plane;
    const int strip = rem / col_groups, cg = rem - strip * col_groups;
    const int y0 = strip * SR;
    const int m = cg * 64 + lane;  // this lane's 4-column group
    const int ox = 4 * m;

    float kr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kr[i] = p.flip ? p.kern[15 - i] : p.kern[i];
    float kh[4], kv[4];
    bool separable = kr[0] != 0.f;
    {
        float kmax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) kmax = fmaxf(kmax, fabsf(kr[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) kh[i] = kr[i], kv[i] = separable ? kr[4 * i] / kr[0] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) separable = separable && fabsf(kr[i] - kv[i >> 2] * kh[i & 3]) <= 1e-7f * kmax;
    }

    // ---- source addressing.  Window columns: w_i = source column ox - pad_x0 + i, i = 0..6 (7 = 0).
    const int hp = (p.in_h + 1) >> 1, wpp = W2E_PLANAR_PITCH((p.in_w - 1) >> 1);  // (in_w = 2W+1)
    const int64_t plane_stride = PLANAR ? (int64_t)4 * hp * wpp : (int64_t)p.in_h * p.in_w;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)plane * plane_stride), (short)0,
                                                                        (int)(plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rzero = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), (short)0, 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int c0 = ox - p.pad_x0;  // source column of w_0
    auto col_ok = [&](int c) { return c >= 0 && c < p.in_w; };
    // pair q: (first window index, second window index); byte offset of the pair inside its source row; which row plane
    unsigned voff[NP];
    unsigned fix_lo = 0, fix_hi = 0;  // bit q: the pair's first / second element lies outside the image while the other is inside
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        int ia, ib, idx;
        if (PLANAR) {  // q = 0,1,2: odd-column plane (w0,w2),(w2,w4),(w4,w6); q = 3,4: even-column plane (w1,w3),(w3,w5)
            ia = q < 3 ? 2 * q : 2 * (q - 3) + 1, ib = ia + 2;
            idx = (c0 + ia) >> 1;  // entry of its parity plane (floor: c0 + ia may be -1)
        } else {
            ia = q, ib = q + 1;
            idx = c0 + ia;
        }
        const bool oka = col_ok(c0 + ia), okb = col_ok(c0 + ib);
        voff[q] = (oka || okb) ? (unsigned)idx * 4u : OOB;
        if (!oka && okb) fix_lo |= 1u << q, voff[q] = (unsigned)(idx + 1) * 4u;  // fetch one entry later, shift up by the select
        if (oka && !okb) fix_hi |= 1u << q;
    }
    const bool wave_fix = PLANAR && __builtin_amdgcn_ballot_w64((fix_lo | fix_hi) != 0) != 0ull;  // (uniform)
    unsigned vd[3];  // dense: w0,w1 (8 B) | w2..w5 (16 B) | w6 (4 B)
    vd[0] = (col_ok(c0) && col_ok(c0 + 1)) ? (unsigned)c0 * 4u : OOB;
    vd[1] = (col_ok(c0 + 2) && col_ok(c0 + 5)) ? (unsigned)(c0 + 2) * 4u : OOB;
    vd[2] = col_ok(c0 + 6) ? (unsigned)(c0 + 6) * 4u : OOB;
    struct Win {
        f32x2 q[NP];  // PLANAR: the pairs as loaded.  Dense: q[0] = (w0,w1), then the 16-byte group d4 = w2..w5 and d1 = w6;
        f32x4 d4;     //         the three pairs that straddle those loads are put together where they are consumed
        float d1;
        f32x2 yq;     // ACTBWD: the forward output at the same 7 columns
        f32x4 y4;
        float y1;
    };
    const __amdgpu_buffer_rsrc_t ryf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ACTBWD ? p.y_fwd + (int64_t)plane * plane_stride : p.x),
                                                                       (short)0, ACTBWD ? (int)(plane_stride * 4) : 0, 0x00020000);
    // (`need` false: a row nobody will consume -- the load is still ISSUED, against the empty descriptor, so that every path
    // through the loop has the same number of loads in flight and the compiler's counter waits stay partial)
    auto load_row = [&](int r, Win& w, bool need) __attribute__((always_inline)) {
        const bool ok = need && r >= 0 && r < p.in_h && !W2E_BLUR_SKIP(p, 1);
        const __amdgpu_buffer_rsrc_t rs = ok ? rx : rzero;
        unsigned so = 0u, se = 0u;
        if (ok) {
            if (PLANAR) {  // column parity of w_0 = parity of pad_x0 (odd): the pairs q0..q2 live in plane px = that parity
                const int par0 = p.pad_x0 & 1;  // (ox is a multiple of 4; kept off the lane id so that the offsets stay scalar)
                so = (unsigned)((((r & 1) * 2 + par0) * hp + (r >> 1)) * wpp) * 4u;
                se = (unsigned)((((r & 1) * 2 + (par0 ^ 1)) * hp + (r >> 1)) * wpp) * 4u;
            } else {
                so = se = (unsigned)(r * p.in_w) * 4u;
            }
        }
        if (PLANAR) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
                w.q[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff[q], q >= 3 ? se : so, 0));
        } else {  // in_w % 4 == 0: each of the three loads is wholly inside or wholly outside the row
            w.q[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vd[0], so, 0));
            w.d4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vd[1], so, 0));
            w.d1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vd[2], so, 0));
            if (ACTBWD) {
                const __amdgpu_buffer_rsrc_t ry2 = ok ? ryf : rzero;
                w.yq = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ry2, vd[0], so, 0));
                w.y4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry2, vd[1], so, 0));
                w.y1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry2, vd[2], so, 0));
            }
        }
    };
    // ACTBWD: window of gradients -> window of pre-activation gradients, and this lane's share of the three sums (its own 4
    // columns w2..w5 of the rows its strip owns; `own` is uniform)
    float s_pre = 0.f, s_noise = 0.f, s_sum = 0.f;
    const float inv_pos = 1.f / p.gain, inv_neg = 1.f / (p.gain * p.slope);
    auto act_bwd = [&](Win& w, const f32x4 nz, bool own) __attribute__((always_inline)) {
        if (!ACTBWD) return;
        const float ya[7] = {w.yq[0], w.yq[1], w.y4[0], w.y4[1], w.y4[2], w.y4[3], w.y1};
        float ga[7] = {w.q[0][0], w.q[0][1], w.d4[0], w.d4[1], w.d4[2], w.d4[3], w.d1};
#pragma unroll
        for (int i = 0; i < 7; ++i) ga[i] = ga[i] * p.gain * (ya[i] > 0.f ? 1.f : p.slope);
        if (own) {
#pragma unroll
            for (int i = 2; i < 6; ++i) {
                s_pre += ga[i] * (ya[i] * (ya[i] > 0.f ? inv_pos : inv_neg));
                s_noise += ga[i] * nz[i - 2];
                s_sum += ga[i];
            }
        }
        w.q[0] = f32x2{ga[0], ga[1]}, w.d4 = f32x4{ga[2], ga[3], ga[4], ga[5]}, w.d1 = ga[6];
    };
    // noise of SOURCE row r at this lane's own columns (ACTBWD: the s_noise term); travels with the row's window
    const __amdgpu_buffer_rsrc_t rnb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((ACTBWD && p.noise) ? p.noise : p.x), (short)0,
                                                                       (ACTBWD && p.noise) ? (int)((unsigned)(p.in_h * p.in_w) * 4u) : 0, 0x00020000);
    auto load_noise_src = [&](int r, bool need) __attribute__((always_inline)) {
        if (!ACTBWD) return f32x4{0.f, 0.f, 0.f, 0.f};
        const bool ok = need && r >= 0 && r < p.in_h;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rnb : rzero, vd[1], ok ? (unsigned)(r * p.in_w) * 4u : 0u, 0));
    };
    auto fix_row = [&](Win& w) __attribute__((always_inline)) {  // only in waves that hold an edge lane
        if (!wave_fix) return;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (fix_lo & (1u << q)) w.q[q][1] = w.q[q][0], w.q[q][0] = 0.f;  // (the load was shifted one entry up)
            if (fix_hi & (1u << q)) w.q[q][1] = 0.f;
        }
    };

    // ---- destination and epilogue constants
    const unsigned out_bytes = (unsigned)(p.out_h * p.out_w) * 4u;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)plane * p.out_h * p.out_w, (short)0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ACT && p.noise ? p.noise : p.x), (short)0,
                                                                      (ACT && p.noise) ? (int)out_bytes : 0, 0x00020000);
    const bool full4 = ox + 3 < p.out_w;
    const bool edge5 = !PLANAR && (p.out_w & 3) == 1 && ox + 4 == p.out_w - 1;  // this lane also owns the single last column
    const bool wave_edge5 = !PLANAR && __builtin_amdgcn_ballot_w64(edge5) != 0ull;  // (uniform)
    const unsigned vout = ox < p.out_w ? (unsigned)ox * 4u : OOB;
    // (v*scale + bias + nw*noise) -> lrelu -> *gain  ==  lrelu(v*scale*gain + bias*gain + nw*gain*noise) for gain > 0
    const bool fold = ACT && p.gain > 0.f && p.slope >= 0.f && p.slope <= 1.f;  // (uniform) then lrelu(e) = max(e, slope*e)
    const float g1 = fold ? p.gain : 1.f;
    const float nw = (ACT && p.noise) ? p.noise_w[0] * g1 : 0.f;
    float e_scale = g1, e_bias = 0.f;
    if (ACT) {
        if (p.out_scale) e_scale *= p.out_scale[plane];
        if (p.bias) e_bias = p.bias[plane % p.channels] * g1;
    }

    // open output rows: slot j holds the row with (Y - y0) & 3 == j as the output pairs A, B (+ the single last column)
    f32x2 accA[4], accB[4];
    float acc5[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accA[j] = accB[j] = f32x2{0.f, 0.f}, acc5[j] = 0.f;

    // source row r0 + k feeds output row y0 + k - a with kernel row a (a <= a_hi: rows of THIS strip); y0 % 4 == 0, so the
    // accumulator slot (k - a) & 3 is a compile-time constant at every call site (k_phase = k & 3 is passed as a literal)
    auto scatter = [&](const Win& w, const int k_phase, const int a_hi) __attribute__((always_inline)) {
        if (W2E_BLUR_SKIP(p, 2)) return;
        // taps of output pair A / B: PLANAR A = outputs (0,2): (w0,w2),(w1,w3),(w2,w4),(w3,w5) = q0,q3,q1,q4;  B = (1,3): q3,q1,q4,q2
        //                            dense  A = outputs (0,1): q0,q1,q2,q3;                                     B = (2,3): q2,q3,q4,q5
        f32x2 dq[6];
        if (!PLANAR) {
            dq[0] = w.q[0], dq[1] = f32x2{w.q[0][1], w.d4[0]}, dq[2] = f32x2{w.d4[0], w.d4[1]}, dq[3] = f32x2{w.d4[1], w.d4[2]};
            dq[4] = f32x2{w.d4[2], w.d4[3]}, dq[5] = f32x2{w.d4[3], w.d1};
        }
        const f32x2 tA0 = PLANAR ? w.q[0] : dq[0], tA1 = PLANAR ? w.q[3] : dq[1], tA2 = PLANAR ? w.q[1] : dq[2], tA3 = PLANAR ? w.q[4] : dq[3];
        const f32x2 tB0 = PLANAR ? w.q[3] : dq[2], tB1 = PLANAR ? w.q[1] : dq[3], tB2 = PLANAR ? w.q[4] : dq[4], tB3 = PLANAR ? w.q[2] : dq[5];
        // single last column (dense): w4, w5, w6
        const float t50 = PLANAR ? 0.f : w.d4[2], t51 = PLANAR ? 0.f : w.d4[3], t52 = PLANAR ? 0.f : w.d1;
        if (separable) {
            const f32x2 hA = kh[0] * tA0 + kh[1] * tA1 + kh[2] * tA2 + kh[3] * tA3;
            const f32x2 hB = kh[0] * tB0 + kh[1] * tB1 + kh[2] * tB2 + kh[3] * tB3;
            const float h5 = wave_edge5 ? kh[0] * t50 + kh[1] * t51 + kh[2] * t52 : 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a > a_hi) continue;
                const int slot = (k_phase - a) & 3;
                accA[slot] += kv[a] * hA, accB[slot] += kv[a] * hB;
                if (wave_edge5) acc5[slot] += kv[a] * h5;
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a > a_hi) continue;
                const int slot = (k_phase - a) & 3;
                accA[slot] += kr[a * 4] * tA0 + kr[a * 4 + 1] * tA1 + kr[a * 4 + 2] * tA2 + kr[a * 4 + 3] * tA3;
                accB[slot] += kr[a * 4] * tB0 + kr[a * 4 + 1] * tB1 + kr[a * 4 + 2] * tB2 + kr[a * 4 + 3] * tB3;
                if (wave_edge5) acc5[slot] += kr[a * 4] * t50 + kr[a * 4 + 1] * t51 + kr[a * 4 + 2] * t52;
            }
        }
    };
    auto load_noise = [&](int Y, bool need) __attribute__((always_inline)) {
        if (!ACT) return f32x4{0.f, 0.f, 0.f, 0.f};
        const bool ok = need && Y < p.out_h;  // (uniform)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rn : rzero, vout, ok ? (unsigned)(Y * p.out_w) * 4u : 0u, 0));
    };
    auto emit = [&](int Y, const int slot, const f32x4 nz) __attribute__((always_inline)) {
        float v[4];
        if (PLANAR) v[0] = accA[slot][0], v[2] = accA[slot][1], v[1] = accB[slot][0], v[3] = accB[slot][1];
        else v[0] = accA[slot][0], v[1] = accA[slot][1], v[2] = accB[slot][0], v[3] = accB[slot][1];
        const float v5 = acc5[slot];
        accA[slot] = accB[slot] = f32x2{0.f, 0.f}, acc5[slot] = 0.f;
        if (Y >= p.out_h) return;  // (uniform)
        const unsigned srow = (unsigned)(Y * p.out_w) * 4u;
        if (ACT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = v[c] * e_scale + e_bias + nw * nz[c];
                v[c] = fold ? fmaxf(e, e * p.slope) : (e > 0.f ? e : e * p.slope) * p.gain;
            }
        }
        if (W2E_BLUR_SKIP(p, 4) && v[0] != 123456.75f) return;
        f32x4 o;
        o[0] = v[0], o[1] = v[1], o[2] = v[2], o[3] = v[3];
        if (PLANAR || (p.out_w & 3) == 0) {  // every lane: a whole group or nothing (past the descriptor: dropped)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), ry, vout, srow, 0);
        } else {
            if (full4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), ry, vout, srow, 0);
            } else if (vout != OOB) {  // ragged right edge
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (ox + c < p.out_w) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[c]), ry, vout + 4u * c, srow, 0);
            }
            if (wave_edge5 && edge5) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v5), ry, vout + 16u, srow, 0);
        }
    };

    // rows: the strip's outputs y0 .. y0+SR-1 need source rows r0 .. r0 + SR + 2, r0 = y0 - pad_y0; source row r0 + 3 + i closes
    // output row y0 + i, whose noise row travels with it (issued BEFORE that window's loads: the counter waits stay partial)
    const int r0 = y0 - p.pad_y0;
    Win w0, w1, w2, w3;
    f32x4 n0, n1, n2, n3;      // ACT: noise of the output row a window closes
    f32x4 m0, m1, m2, m3;      // ACTBWD: noise of the SOURCE row of window w0..w3 (own columns)
    m0 = load_noise_src(r0, true), load_row(r0, w0, true);
    m1 = load_noise_src(r0 + 1, true), load_row(r0 + 1, w1, true);
    m2 = load_noise_src(r0 + 2, true), load_row(r0 + 2, w2, true);
    n0 = load_noise(y0, true), m3 = load_noise_src(r0 + 3, true), load_row(r0 + 3, w3, true);
    // a strip OWNS the source rows y0 .. y0+SR-1 = windows k = pad_y0 .. pad_y0+SR-1 (ACTBWD runs with pad_y0 = 2): the sums count
    // every source element exactly once although the halo rows are read by two strips
    // prologue: the first three source rows only open output rows >= y0  (kernel rows a <= k)
    fix_row(w0), act_bwd(w0, m0, p.pad_y0 <= 0), scatter(w0, 0, 0);
    n1 = load_noise(y0 + 1, true), m0 = load_noise_src(r0 + 4, true), load_row(r0 + 4, w0, true);
    fix_row(w1), act_bwd(w1, m1, p.pad_y0 <= 1), scatter(w1, 1, 1);
    n2 = load_noise(y0 + 2, true), m1 = load_noise_src(r0 + 5, true), load_row(r0 + 5, w1, true);
    fix_row(w2), act_bwd(w2, m2, p.pad_y0 <= 2), scatter(w2, 2, 2);
    n3 = load_noise(y0 + 3, true), m2 = load_noise_src(r0 + 6, true), load_row(r0 + 6, w2, true);
    int rows = p.out_h - y0;  // output rows of this strip
    rows = rows < SR ? rows : SR;
    const int k_own_end = p.pad_y0 + SR;  // windows k >= this belong to the next strip
    for (int i = 0; i < rows; i += 4) {
        const bool more = i + 4 < rows;  // (the last group needs source rows up to r0 + i + 6 only: already in flight)
        fix_row(w3), act_bwd(w3, m3, i + 3 < k_own_end), scatter(w3, 3, 3);
        emit(y0 + i, 0, n0);
        n0 = load_noise(y0 + i + 4, more), m3 = load_noise_src(r0 + 7 + i, more), load_row(r0 + 7 + i, w3, more);
        fix_row(w0), act_bwd(w0, m0, i + 4 < k_own_end), scatter(w0, 0, 3);
        emit(y0 + i + 1, 1, n1);
        n1 = load_noise(y0 + i + 5, more), m0 = load_noise_src(r0 + 8 + i, more), load_row(r0 + 8 + i, w0, more);
        fix_row(w1), act_bwd(w1, m1, i + 5 < k_own_end), scatter(w1, 1, 3);
        emit(y0 + i + 2, 2, n2);
        n2 = load_noise(y0 + i + 6, more), m1 = load_noise_src(r0 + 9 + i, more), load_row(r0 + 9 + i, w1, more);
        fix_row(w2), act_bwd(w2, m2, i + 6 < k_own_end), scatter(w2, 2, 3);
        emit(y0 + i + 3, 3, n3);
        n3 = load_noise(y0 + i + 7, more), m2 = load_noise_src(r0 + 10 + i, more), load_row(r0 + 10 + i, w2, more);
    }
    if (ACTBWD) {  // one atomic per wave and sum (the three sums of a plane are zeroed by the host before the launch)
        float t0 = s_pre, t1 = s_noise, t2 = s_sum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t0 += __shfl_xor(t0, off, 64), t1 += __shfl_xor(t1, off, 64), t2 += __shfl_xor(t2, off, 64);
        if (lane == 0) atomicAdd(p.sums + (int64_t)plane * 3, t0), atomicAdd(p.sums + (int64_t)plane * 3 + 1, t1), atomicAdd(p.sums + (int64_t)plane * 3 + 2, t2);
    }
}

// General tap counts (<= MAX_TILE_K per axis), up = down = 1.
__global__ __launch_bounds__(256) void upfirdn_tile_kernel(UpfirdnParams p, int tiles_x, int tiles_y, unsigned pw_magic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kbuf = smem;  // kh*kw taps, already oriented for correlation
    float* tile = smem + MAXK * MAXK;
    const int kh = p.kh, kw = p.kw;
    const int pw = TW + kw - 1, ph = TH + kh - 1;
    const int pitch = (pw + 3) & ~3;
    const int tid = threadIdx.x;
    for (int i = tid; i < kh * kw; i += 256) {
        const int ky = i / kw, kx = i % kw;
        kbuf[i] = p.flip ? p.kern[(kh - 1 - ky) * kw + (kw - 1 - kx)] : p.kern[i];
    }
    const float nw = (p.act && p.noise) ? p.noise_w[0] : 0.f;
    const int tiles_per_plane = tiles_x * tiles_y;
    const int n_tiles = tiles_per_plane * (int)p.planes;
    const int ly = (tid >> 4) * 2, lx = (tid & 15) * 4;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int plane = t / tiles_per_plane;
        const int rem = t - plane * tiles_per_plane;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const float* src = p.x + (int64_t)plane * p.in_h * p.in_w;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 - p.pad_y0, ix0 = ox0 - p.pad_x0;
        __syncthreads();
        for (int i = tid; i < ph * pw; i += 256) {
            const int r = (int)__umulhi((unsigned)i, pw_magic), c = i - r * pw;
            const int iy = iy0 + r, ix = ix0 + c;
            tile[r * pitch + c] = (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) ? src[iy * p.in_w + ix] : 0.f;
        }
        __syncthreads();
        float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int r = 0; r < kh + 1; ++r) {
            const float* rp = tile + (ly + r) * pitch + lx;
            for (int kx = 0; kx < kw; ++kx) {
                const float k0 = r < kh ? kbuf[r * kw + kx] : 0.f;
                const float k1 = r >= 1 ? kbuf[(r - 1) * kw + kx] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = rp[kx + j];
                    acc[0][j] += k0 * v;
                    acc[1][j] += k1 * v;
                }
            }
        }
        const int ox = ox0 + lx;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const int oy = oy0 + ly + yy;
            if (oy >= p.out_h || ox >= p.out_w) continue;
            float* dst = p.y + ((int64_t)plane * p.out_h + oy) * p.out_w + ox;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ox + j < p.out_w) dst[j] = epilogue(p, acc[yy][j], plane, oy, ox + j, nw);
        }
    }
}

// 4x4 taps, up = 1, any down (the adjoint of the RGB-skip Upsample: [B,3,h,w] -> [B,3,h/2,w/2]): one output per thread, the 16
// source values fetched by 16 independent buffer loads (out-of-image taps carry an out-of-range offset and read 0: no branches),
// all in flight before the first FMA.  The tensors are small (3 channels): the launch is latency-bound, not bandwidth-bound.
__global__ __launch_bounds__(256) void upfirdn_down4_kernel(UpfirdnParams p, int64_t total) {
    float kr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kr[i] = p.flip ? p.kern[15 - i] : p.kern[i];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), (short)0, (int)(p.planes * p.in_h * p.in_w * 4), 0x00020000);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ox = (int)(e % p.out_w);
    const int64_t t = e / p.out_w;
    const int oy = (int)(t % p.out_h);
    const int plane = (int)(t / p.out_h);
    const int iy0 = oy * p.down - p.pad_y0, ix0 = ox * p.down - p.pad_x0;
    float v[16];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const int iy = iy0 + ky;
        const bool row_ok = iy >= 0 && iy < p.in_h;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int ix = ix0 + kx;
            const bool ok = row_ok && ix >= 0 && ix < p.in_w;
            const unsigned off = ok ? (unsigned)((plane * p.in_h + iy) * p.in_w + ix) * 4u : 0x80000000u;
            v[ky * 4 + kx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += kr[i] * v[i];
    p.y[e] = acc;
}

__global__ void upfirdn_generic_kernel(UpfirdnParams p, int64_t total) {
    __shared__ float kbuf[MAXK * MAXK];
    for (int i = threadIdx.x; i < p.kh * p.kw; i += blockDim.x) {
        const int ky = i / p.kw, kx = i % p.kw;
        kbuf[i] = p.flip ? p.kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : p.kern[i];
    }
    __syncthreads();
    const float nw = (p.act && p.noise) ? p.noise_w[0] : 0.f;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int ox = (int)(e % p.out_w), oy = (int)((e / p.out_w) % p.out_h);
        const int64_t plane = e / ((int64_t)p.out_w * p.out_h);
        const float* src = p.x + plane * (int64_t)p.in_h * p.in_w;
        float acc = 0.f;
        for (int ky = 0; ky < p.kh; ++ky) {
            const int zy = oy * p.down + ky - p.pad_y0;  // row in the zero-stuffed image
            if (zy < 0 || zy % p.up != 0) continue;
            const int iy = zy / p.up;
            if (iy >= p.in_h) continue;
            for (int kx = 0; kx < p.kw; ++kx) {
                const int zx = ox * p.down + kx - p.pad_x0;
                if (zx < 0 || zx % p.up != 0) continue;
                const int ix = zx / p.up;
                if (ix >= p.in_w) continue;
                acc += kbuf[ky * p.kw + kx] * src[(int64_t)iy * p.in_w + ix];
            }
        }
        p.y[e] = epilogue(p, acc, plane, oy, ox, nw);
    }
}

}  // namespace w2e

using namespace w2e;

extern "C" int w2e_upfirdn2d(const float* x, const float* kern, float* y, int64_t planes, int in_h, int in_w, int out_h,
                             int out_w, int kh, int kw, int up, int down, int pad_x0, int pad_y0, int flip, int in_layout, int in_pitch,
                             int act, const float* out_scale, const float* noise, const float* noise_w, const float* bias,
                             int channels, float slope, float gain, void* stream) {
    W2E_REQUIRE(x && kern && y, "upfirdn2d: null tensor");
    W2E_REQUIRE(planes >= 0 && in_h > 0 && in_w > 0 && out_h >= 0 && out_w >= 0, "upfirdn2d: bad sizes");
    W2E_REQUIRE(kh >= 1 && kw >= 1 && kh <= MAXK && kw <= MAXK, "upfirdn2d: kernel %dx%d unsupported (max %d)", kh, kw,
                MAXK);
    W2E_REQUIRE(up >= 1 && down >= 1, "upfirdn2d: up/down must be >= 1");
    W2E_REQUIRE(!noise || noise_w, "upfirdn2d: noise without noise_w");
    W2E_REQUIRE(!act || !bias || channels > 0, "upfirdn2d: bias needs channels");
    const int64_t total = planes * out_h * out_w;
    if (total == 0) return 0;
    W2E_REQUIRE(in_layout == 0 || (in_layout == 1 && up == 1 && down == 1 && kh == 4 && kw == 4 && out_w >= 32),
                "upfirdn2d: the phase-planar input layout is implemented for the 4x4, up=down=1 tile kernel only");
    // The caller allocated the planar image: its row pitch must be THIS library's (the kernels index with W2E_PLANAR_PITCH; a caller built
    // against another pitch hands over a buffer of another size and the kernel would read past it -- round 3's e7 memory fault).
    W2E_REQUIRE(in_layout == 0 || in_pitch == W2E_PLANAR_PITCH((in_w - 1) >> 1),
                "upfirdn2d: planar input with a row pitch of %d floats, this library's layout has %d (W2E_PLANAR_PITCH, ABI %d): rebuild the caller",
                in_pitch, W2E_PLANAR_PITCH((in_w - 1) >> 1), W2E_VERSION);
    const int tune = options().tune_blur;
    UpfirdnParams p{x, kern, y, planes, in_h, in_w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0, flip, tune, in_layout,
                    act, out_scale, noise, noise_w, bias, channels > 0 ? channels : 1, slope, gain, nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    if (up == 1 && down == 1 && out_w >= 32 && kh <= MAX_TILE_K && kw <= MAX_TILE_K) {
        const int tiles_x = (int)ceil_div(out_w, TW), tiles_y = (int)ceil_div(out_h, TH);
        const int64_t n_tiles = planes * tiles_x * tiles_y;
        W2E_REQUIRE(n_tiles < ((int64_t)1 << 31) && (int64_t)in_h * in_w < ((int64_t)1 << 31), "upfirdn2d: tensor too large");
        const int pw = TW + kw - 1, pitch = (pw + 3) & ~3;
        const size_t lds = sizeof(float) * (MAXK * MAXK + (size_t)(TH + kh) * pitch + 16);
        const int grid = (int)(n_tiles < 16384 ? n_tiles : 16384);
        const unsigned magic = (unsigned)(((uint64_t)1 << 32) / (unsigned)pw + 1);
        if (kh == 4 && kw == 4) {
            const bool vec = !in_layout && (in_w & 3) == 0 && ((uintptr_t)x & 15) == 0;
            // wide images: the streaming kernel (no LDS, no barriers); tune bit 8 keeps the tile kernels for comparison
            const bool stream_ok = !(tune & 8) && out_w >= 256 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                                   (in_layout ? (pad_x0 == 1 && (out_w & 3) == 0 && in_w == out_w + 1)
                                              : (pad_x0 == 2 && (in_w & 3) == 0 && out_w <= in_w + 1));
            if (stream_ok) {
                const int lanes = (!in_layout && (out_w & 3) == 1) ? out_w / 4 : (int)ceil_div(out_w, 4);  // (a single last column rides on the last full group)
                const int col_groups = (int)ceil_div(lanes, 64), strips = (int)ceil_div(out_h, SR);
                const int64_t waves = planes * col_groups * strips;
                W2E_REQUIRE(waves < ((int64_t)1 << 31), "upfirdn2d: tensor too large");
                const unsigned blocks = (unsigned)ceil_div(waves, 4);
                if (in_layout) {
                    if (act) upfirdn_stream4_kernel<true, true><<<blocks, 256, 0, s>>>(p, col_groups, strips);
                    else upfirdn_stream4_kernel<false, true><<<blocks, 256, 0, s>>>(p, col_groups, strips);
                } else {
                    if (act) upfirdn_stream4_kernel<true, false><<<blocks, 256, 0, s>>>(p, col_groups, strips);
                    else upfirdn_stream4_kernel<false, false><<<blocks, 256, 0, s>>>(p, col_groups, strips);
                }
            } else if (in_layout || vec) {
                // aligned-source kernel: each workgroup walks (plane, tile row) items of one tile column; ~8 per CU
                const int64_t items = planes * tiles_y;
                int gsteps = 256 * 8 / tiles_x;
                if (gsteps < 1) gsteps = 1;
                if (gsteps > items) gsteps = (int)items;
                const int g2 = gsteps * tiles_x;
                constexpr int PWc = TW + 3, PHc = TH + 3;
                const size_t lds2 = in_layout ? sizeof(float) * 2 * PHc * 4 * (((PWc + 1) / 2 + 1 + 3) / 4 + 1)
                                              : sizeof(float) * PHc * 4 * ((PWc + 3) / 4 + 1);
                if (in_layout) {
                    if (act) upfirdn_blur4_kernel<true, true><<<g2, 256, lds2, s>>>(p, tiles_x, tiles_y);
                    else upfirdn_blur4_kernel<false, true><<<g2, 256, lds2, s>>>(p, tiles_x, tiles_y);
                } else {
                    if (act) upfirdn_blur4_kernel<true, false><<<g2, 256, lds2, s>>>(p, tiles_x, tiles_y);
                    else upfirdn_blur4_kernel<false, false><<<g2, 256, lds2, s>>>(p, tiles_x, tiles_y);
                }
            } else {
                if (act) upfirdn_tile4_kernel<true><<<grid, 256, lds, s>>>(p, tiles_x, tiles_y, magic);
                else upfirdn_tile4_kernel<false><<<grid, 256, lds, s>>>(p, tiles_x, tiles_y, magic);
            }
        } else {
            upfirdn_tile_kernel<<<grid, 256, lds, s>>>(p, tiles_x, tiles_y, magic);
        }
    } else if (up == 1 && kh == 4 && kw == 4 && !act && !in_layout && planes * in_h * in_w < ((int64_t)1 << 29) && total < ((int64_t)1 << 31)) {
        upfirdn_down4_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, s>>>(p, total);
    } else {
        upfirdn_generic_kernel<<<stream_grid(total, 256), 256, 0, s>>>(p, total);
    }
    W2E_LAUNCH_CHECK("upfirdn2d");
    return 0;
}

extern "C" int w2e_blur_adjoint_actbwd(const float* gy, const float* y_fwd, const float* noise, const float* kern, float* gt,
                                       float* sums, int64_t planes, int h, int w, float slope, float gain, void* stream) {
    W2E_REQUIRE(gy && y_fwd && kern && gt && sums, "blur_adjoint_actbwd: null tensor");
    W2E_REQUIRE(planes >= 0 && h > 0 && w >= 256 && (w & 3) == 0, "blur_adjoint_actbwd: needs a width >= 256 that is a multiple of 4 (got %dx%d)", h, w);
    W2E_REQUIRE(((uintptr_t)gy & 15) == 0 && ((uintptr_t)y_fwd & 15) == 0 && ((uintptr_t)gt & 15) == 0 && (!noise || ((uintptr_t)noise & 15) == 0),
                "blur_adjoint_actbwd: tensors must be 16-byte aligned");
    W2E_REQUIRE(gain > 0.f && slope > 0.f, "blur_adjoint_actbwd: gain and slope must be positive");
    W2E_REQUIRE(!options().deterministic, "blur_adjoint_actbwd: the sums are joined with fp32 atomics (use w2e_bias_act_bwd_reduce + w2e_upfirdn2d in deterministic mode)");
    if (planes == 0) return 0;
    W2E_REQUIRE((int64_t)(h + 1) * (w + 1) < ((int64_t)1 << 29), "blur_adjoint_actbwd: plane too large");
    hipStream_t s = (hipStream_t)stream;
    if (zero_async(sums, sizeof(float) * 3 * (size_t)planes, s) != hipSuccess) {
        set_error("blur_adjoint_actbwd: zero fill failed");
        return 2;
    }
    UpfirdnParams p{gy, kern, gt, planes, h, w, h + 1, w + 1, 4, 4, 1, 1, 2, 2, 0, options().tune_blur & 7, 0,
                    0, nullptr, noise, nullptr, nullptr, 1, slope, gain, y_fwd, sums};
    const int lanes = w / 4;  // out_w = w + 1: the single last column rides on the last full group
    const int col_groups = (int)ceil_div(lanes, 64), strips = (int)ceil_div(h + 1, SR);
    const int64_t waves = planes * col_groups * strips;
    W2E_REQUIRE(waves < ((int64_t)1 << 31), "blur_adjoint_actbwd: tensor too large");
    upfirdn_stream4_kernel<false, false, true><<<(unsigned)ceil_div(waves, 4), 256, 0, s>>>(p, col_groups, strips);
    W2E_LAUNCH_CHECK("blur_adjoint_actbwd");
    return 0;
}

