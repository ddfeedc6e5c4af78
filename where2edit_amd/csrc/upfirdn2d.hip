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

struct UpfirdnParams {
    const float* x;
    const float* kern;
    float* y;
    int64_t planes;
    int in_h, in_w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0, flip;
    int tune;    // tuning aid (W2E_TUNE_BLUR): 1 = no global loads, 2 = no FIR arithmetic, 4 = no stores
    int planar;  // 1: x is phase-planar [planes][2][2][(in_h+1)/2][(in_w+1)/2]
    int act;
    const float* out_scale;
    const float* noise;
    const float* noise_w;
    const float* bias;
    int channels;
    float slope, gain;
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
            const float v = (p.tune & 1) ? 1.f : src[cy * p.in_w + cx];
            if (i < PH * PW) tile[r * PITCH + c] = ok ? v : 0.f;
        }
        __syncthreads();
        float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < ((p.tune & 2) ? 1 : 5); ++r) {  // input row ly+r feeds output row 0 with ky=r and output row 1 with ky=r-1
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
            if ((p.tune & 4) && v[0] != 123456.75f) continue;
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
    const int hp = (p.in_h + 1) >> 1, wpp = (((p.in_w + 1) >> 1) + 3) & ~3;
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
            const bool ok = s_mask[it] != 0 && iy >= 0 && iy < p.in_h && !(p.tune & 1);
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
                if ((p.tune & 2) && r > 0) break;
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
                if ((p.tune & 2) && r > 0) break;
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
            if ((p.tune & 4) && e != 123456.75f) continue;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, e), ry, voff0 + (unsigned)(j * p.out_w) * 4u, 0, 0);
        }
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
                             int out_w, int kh, int kw, int up, int down, int pad_x0, int pad_y0, int flip, int in_layout, int act,
                             const float* out_scale, const float* noise, const float* noise_w, const float* bias,
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
    const int tune = options().tune_blur;
    UpfirdnParams p{x, kern, y, planes, in_h, in_w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0, flip, tune, in_layout,
                    act, out_scale, noise, noise_w, bias, channels > 0 ? channels : 1, slope, gain};
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
            if (in_layout || vec) {
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
    } else {
        upfirdn_generic_kernel<<<stream_grid(total, 256), 256, 0, s>>>(p, total);
    }
    W2E_LAUNCH_CHECK("upfirdn2d");
    return 0;
}
