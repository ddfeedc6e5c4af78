// What a co-resident wave can issue beside a saturated fp32 matrix pipe on gfx950, by instruction kind: scalar fp32 VALU, PACKED fp32
// VALU (v_pk_fma_f32: two fp32 lanes-worth per instruction), LDS stores.   (profiles/r04_pk_probe.txt)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_probe.hip -o pk_probe && ./pk_probe
// Waves 0-3 of a 512-thread workgroup (one per SIMD) play role A, waves 4-7 role B; s_memtime around each wave's loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// roles: 0 idle, 1 MFMA 32x32x2 only, 2 MFMA 16x16x4 only, 3 16 x v_fma_f32 (CH independent chains), 4 16 x v_pk_fma_f32 (CH chains),
//        5 16 x ds_write_b32, 6 8 x v_fma + 8 x ds_write_b32 interleaved
template <int ROLE_A, int ROLE_B, int CH, int PRIO_B = 0>
__global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, int iters, float seed) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    lds[threadIdx.x] = seed;
    __syncthreads();
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    float s[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) s[i] = seed * (i + 1) + threadIdx.x, p[i] = f32x2{seed * i, seed + i};
    float x = seed + (threadIdx.x & 31), y = seed - (threadIdx.x & 7);
    const unsigned laddr = (threadIdx.x & 255) * 4;
    if (PRIO_B && wave >= 4) __builtin_amdgcn_s_setprio(3);  // (wave-uniform)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc4[i], 0, 0, 0);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(s[i % CH]));
        }
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i % CH]));
        }
    } else if (role == 5) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(laddr), "v"(s[i & 7]), "n"(1024 * (i & 7)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (role == 6) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(s[i % CH]));
                asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(laddr), "v"(s[(i + 4) & 7]), "n"(1024 * (i & 7)) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int i = 0; i < 4; ++i) sink += acc[i][0];
    for (int i = 0; i < 8; ++i) sink += acc4[i][0] + s[i] + p[i][0] + p[i][1];
    if (sink == 123.456f) out[64] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB, int CH, int PB = 0>
static void run(const char* name, int per_iter_a, int per_iter_b) {
    unsigned long long* d;
    hipMalloc(&d, 4096);
    const int iters = 2000;
    probe<RA, RB, CH, PB><<<1, 512>>>(d, 10, 1.0f);
    probe<RA, RB, CH, PB><<<1, 512>>>(d, iters, 1.0f);
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 4; ++i) a += h[i] / 4.0, b += h[4 + i] / 4.0;
    printf("%-64s A %8.1f cycles/instr   B %8.1f cycles/instr\n", name, per_iter_a ? a / iters / per_iter_a : 0.0,
           per_iter_b ? b / iters / per_iter_b : 0.0);
    hipFree(d);
}

int main() {
    run<1, 0, 4>("A: MFMA 32x32x2 only", 4, 0);
    run<2, 0, 4>("A: MFMA 16x16x4 only", 8, 0);
    run<3, 0, 4>("A: v_fma_f32, 4 chains", 16, 0);
    run<3, 0, 8>("A: v_fma_f32, 8 chains", 16, 0);
    run<4, 0, 4>("A: v_pk_fma_f32, 4 chains", 16, 0);
    run<4, 0, 8>("A: v_pk_fma_f32, 8 chains", 16, 0);
    run<3, 3, 8>("A, B: v_fma_f32, 8 chains (2 waves / SIMD)", 16, 16);
    run<4, 4, 8>("A, B: v_pk_fma_f32, 8 chains (2 waves / SIMD)", 16, 16);
    run<5, 0, 4>("A: ds_write_b32", 16, 0);
    run<1, 3, 8>("A: MFMA 32x32x2, B: v_fma_f32 8 chains", 4, 16);
    run<1, 4, 8>("A: MFMA 32x32x2, B: v_pk_fma_f32 8 chains", 4, 16);
    run<2, 3, 8>("A: MFMA 16x16x4, B: v_fma_f32 8 chains", 8, 16);
    run<2, 4, 8>("A: MFMA 16x16x4, B: v_pk_fma_f32 8 chains", 8, 16);
    run<1, 3, 8, 1>("A: MFMA 32x32x2, B: v_fma_f32 8 chains, s_setprio 3 in B", 4, 16);
    run<1, 4, 8, 1>("A: MFMA 32x32x2, B: v_pk_fma_f32 8 chains, s_setprio 3 in B", 4, 16);
    run<1, 6, 8, 1>("A: MFMA 32x32x2, B: v_fma + ds_write_b32, s_setprio 3 in B", 4, 16);
    run<1, 5, 4>("A: MFMA 32x32x2, B: ds_write_b32", 4, 16);
    run<1, 6, 8>("A: MFMA 32x32x2, B: v_fma + ds_write_b32 interleaved", 4, 16);
    run<2, 6, 8>("A: MFMA 16x16x4, B: v_fma + ds_write_b32 interleaved", 8, 16);
    return 0;
}
