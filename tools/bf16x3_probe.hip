// Ceiling probe for the "fp32 as three bf16 products" form of the conv inner loop: what the matrix pipe
// and the LDS operand reads sustain when one 16-channel tap product of a 32x32 tile is
//     acc += A_hi*B_hi + A_hi*B_lo + A_lo*B_hi        (3 x v_mfma_f32_32x32x16_bf16, operands = one ds_read_b128 each)
// instead of 8 x v_mfma_f32_32x32x2_f32.  Tile shape of the big SAME configuration: 8 waves as 2 x 4, each wave 2 x 4
// accumulators, 128 output channels x 512 pixels per workgroup, 9 taps per 16-channel chunk, one barrier per chunk.
// No global traffic, no staging: MFMA + LDS-read ceiling only.  "TFLOP/s-equivalent" counts the fp32 convolution's FLOPs.
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o tools/bin/bf16x3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool LDS>
__global__ __launch_bounds__(512, 2) void probe(float* out, const float* in, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TN = 128, PLANE = 18 * 34;                       // 16x32-pixel tile with halo
    bf16x8* ws = reinterpret_cast<bf16x8*>(smem);                   // [hi/lo][tap][h][TN]
    bf16x8* xs = ws + 2 * 9 * 2 * TN;                               // [hi/lo][h][PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wo = wave >> 2, wp = wave & 3;
    for (int i = tid; i < (2 * 9 * 2 * TN + 2 * 2 * PLANE) * 8; i += 512)  // valid random bf16 values
        reinterpret_cast<__bf16*>(smem)[i] = (__bf16)in[(i * 7 + (i >> 10)) & 1023];
    __syncthreads();
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int a_base = half * TN + wo * 64 + j;
    int base[4];
    for (int pb = 0; pb < 4; ++pb) {
        const int m = (wp * 4 + pb) * 32 + j;
        base[pb] = half * PLANE + (m >> 5) * 34 + (m & 31);
    }
    bf16x8 ra[2][2], rb[2][4];
    if (!LDS) {
        for (int q = 0; q < 2; ++q) {
            for (int ob = 0; ob < 2; ++ob) ra[q][ob] = ws[(q * 9 * 2) * TN + a_base + ob * 32];
            for (int pb = 0; pb < 4; ++pb) rb[q][pb] = xs[q * 2 * PLANE + base[pb]];
        }
    }
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (LDS) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob) ra[q][ob] = ws[((q * 9 + tap) * 2) * TN + a_base + ob * 32];
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) rb[q][pb] = xs[q * 2 * PLANE + base[pb] + (tap / 3) * 34 + tap % 3];
                }
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[0][ob], rb[0][pb], acc[ob][pb], 0, 0, 0);
                    acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[0][ob], rb[1][pb], acc[ob][pb], 0, 0, 0);
                    acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[1][ob], rb[0][pb], acc[ob][pb], 0, 0, 0);
                }
        }
        if (LDS) __syncthreads();  // the chunk boundary of the real loop
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool LDS>
void run(const char* name, float* out, float* in) {
    const int chunks = 400, grid = 256;
    const size_t lds = sizeof(float) * 4 * (2 * 9 * 2 * 128 + 2 * 2 * 18 * 34);
    hipFuncSetAttribute((const void*)probe<LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 40; ++rep) probe<LDS><<<grid, 512, lds, 0>>>(out, in, chunks);  // ~0.5 s of load first
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<LDS><<<grid, 512, lds, 0>>>(out, in, chunks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 8 * chunks * 9.0 * 8 * 16 * 2048.0;  // fp32-conv FLOPs: 8 tiles x 16 channels x 32*32*2
        const double mfma = (double)grid * 8 * chunks * 9.0 * 24 * 32768.0;     // bf16 MFMA FLOPs actually issued
        printf("%s: %.3f ms  %.1f TFLOP/s-equivalent (fp32 conv FLOPs), %.0f TFLOP/s of bf16 MFMA\n", name, ms, flop / ms / 1e9, mfma / ms / 1e9);
    }
}
int main() {
    float *out, *in; hipMalloc(&out, 1 << 22); hipMalloc(&in, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<false>("operands in registers (matrix pipe only)", out, in);
    run<true>("operands re-read from LDS per tap + 1 barrier per chunk", out, in);
    return 0;
}
