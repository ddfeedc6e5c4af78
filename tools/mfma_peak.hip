// Sustained v_mfma_f32_32x32x2_f32 rate of the device this runs on (random operands, 4 independent accumulators per
// wave): the practical ceiling the conv kernel's TFLOP/s should be read against.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, const float* in, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Same loop, but every MFMA reads a different pair of operand registers holding random data (what a real GEMM does).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kv(float* out, const float* in, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) a[u] = in[(threadIdx.x * 8 + u) & 1023], b[u] = in[(threadIdx.x * 8 + u + 517) & 1023];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
        a[0] = -a[0];  // (static index: a dynamically indexed register array costs the loop 7 %)
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES>
void runv(const char* name, float* out, float* in, int wg_per_cu) {
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kv<WAVES><<<grid, 64 * WAVES>>>(out, in, iters / 10);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kv<WAVES><<<grid, 64 * WAVES>>>(out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)grid * WAVES * iters * 32 * 4096.0;
        printf("%s [varying random operands]: %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    }
}
template <int WAVES>
void run(const char* name, float* out, float* in, int wg_per_cu) {
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<WAVES><<<grid, 64 * WAVES>>>(out, in, iters / 10);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<WAVES><<<grid, 64 * WAVES>>>(out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)grid * WAVES * iters * 32 * 4096.0;  // 32 MFMAs per iteration, 2*32*32*2 FLOP each
        printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    }
}
int main() {
    float *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<4>("1 wave/SIMD  (4 waves/WG x 1 WG/CU)", out, in, 1);
    run<8>("2 waves/SIMD (8 waves/WG x 1 WG/CU)", out, in, 1);
    run<4>("2 waves/SIMD (4 waves/WG x 2 WG/CU)", out, in, 2);
    run<4>("4 waves/SIMD (4 waves/WG x 4 WG/CU)", out, in, 4);
    run<8>("4 waves/SIMD (8 waves/WG x 2 WG/CU)", out, in, 2);
    run<8>("8 waves/SIMD (8 waves/WG x 4 WG/CU)", out, in, 4);
    runv<4>("1 wave/SIMD  (4 waves/WG x 1 WG/CU)", out, in, 1);
    runv<8>("2 waves/SIMD (8 waves/WG x 1 WG/CU)", out, in, 1);
    runv<4>("2 waves/SIMD (4 waves/WG x 2 WG/CU)", out, in, 2);
    runv<8>("4 waves/SIMD (8 waves/WG x 2 WG/CU)", out, in, 2);
    return 0;
}
