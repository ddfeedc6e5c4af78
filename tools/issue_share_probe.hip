// Follow-up to tools/pk_probe.hip (profiles/r04_pk_probe.txt: beside a wave that issues fp32 MFMAs back to back, a partner wave on the
// same SIMD issues ONE instruction per ~22 cycles).  Two questions that decide the fused Winograd kernel's structure:
//   1. Is the partner throttled because the MFMA wave's NEXT MFMA sits at the issue stage while the pipe is busy (a stalled VALU-class
//      instruction holding the port)?  Then s_nop padding behind each MFMA -- so that the wave does not present an MFMA until the pipe
//      can take it -- would give the port to the partner.  NOPS = number of `s_nop 15` (16 cycles each) behind every MFMA.
//   2. Do TWO partner waves on a SIMD (12-wave workgroup: one matrix + two VALU waves per SIMD) get twice the slots of one?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_share_probe.hip -o /tmp/isp && /tmp/isp       (profiles/r05_issue_share_probe.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: MFMAs on 4 independent accumulators; 1: on ONE accumulator (each MFMA depends on the previous one); 2: 4 accumulators, 4 dependent
// MFMAs on each in turn (the fused kernel's order: a position's 4 k-steps, then the next position)
template <int NOPS, int BWAVES, int MODE, int BKIND>
__global__ __launch_bounds__(256 + 256 * BWAVES, 1) void probe(unsigned long long* out, int iters, float seed) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = seed;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 p[8];
    float s[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{seed * i, seed + i}, s[i] = seed * (i + 1) + threadIdx.x;
    float x = seed + (threadIdx.x & 31), y = seed - (threadIdx.x & 7);
    const unsigned laddr = (threadIdx.x & 255) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int a = MODE == 0 ? (i & 3) : MODE == 1 ? 0 : (i >> 2);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 15");
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (BKIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i & 7]));
                else if (BKIND == 1) {  // the transform's mix: 2 packed VALU : 1 LDS access
                    if (i % 3 == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(laddr), "v"(s[i & 7]), "n"(1024 * (i & 7)) : "memory");
                    else asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i & 7]));
                }
            }
            if (BKIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int i = 0; i < 4; ++i) sink += acc[i][0];
    for (int i = 0; i < 8; ++i) sink += p[i][0] + p[i][1] + s[i];
    if (sink == 123.456f) out[64] = 1;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int NOPS, int BWAVES, int MODE, int BKIND>
static void run(const char* name) {
    unsigned long long* d;
    hipMalloc(&d, 4096);
    const int iters = 2000;
    probe<NOPS, BWAVES, MODE, BKIND><<<1, 256 + 256 * BWAVES>>>(d, 10, 1.0f);
    probe<NOPS, BWAVES, MODE, BKIND><<<1, 256 + 256 * BWAVES>>>(d, iters, 1.0f);
    unsigned long long h[12];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 4; ++i) a += h[i] / 4.0;
    for (int i = 0; i < 4 * BWAVES; ++i) b += h[4 + i] / (4.0 * BWAVES);
    printf("%-86s MFMA wave %6.1f cycles / MFMA | each partner wave %6.1f cycles / instr, all partners of a SIMD together %6.1f\n", name,
           a / iters / 16, BWAVES ? b / iters / 16 : 0.0, BWAVES ? b / iters / 16 / BWAVES : 0.0);
    hipFree(d);
}

int main() {
    run<0, 1, 0, 0>("4 independent accumulators, no padding, 1 partner (v_pk_fma)");
    run<0, 1, 1, 0>("ONE accumulator (dependent MFMAs), no padding, 1 partner");
    run<0, 1, 2, 0>("4 dependent MFMAs per accumulator in turn, no padding, 1 partner");
    run<1, 1, 0, 0>("4 independent accumulators, 1 x s_nop 15 behind each MFMA, 1 partner");
    run<2, 1, 0, 0>("4 independent accumulators, 2 x s_nop 15, 1 partner");
    run<3, 1, 0, 0>("4 independent accumulators, 3 x s_nop 15, 1 partner");
    run<4, 1, 0, 0>("4 independent accumulators, 4 x s_nop 15, 1 partner");
    run<3, 1, 2, 0>("4 dependent MFMAs per accumulator in turn, 3 x s_nop 15, 1 partner");
    run<0, 2, 0, 0>("4 independent accumulators, no padding, 2 partners per SIMD");
    run<3, 2, 0, 0>("4 independent accumulators, 3 x s_nop 15, 2 partners per SIMD");
    run<0, 1, 0, 1>("4 independent accumulators, no padding, 1 partner (2 v_pk_fma : 1 ds_write)");
    run<3, 1, 0, 1>("4 independent accumulators, 3 x s_nop 15, 1 partner (2 v_pk_fma : 1 ds_write)");
    run<0, 2, 0, 1>("4 independent accumulators, no padding, 2 partners (2 v_pk_fma : 1 ds_write)");
    run<3, 2, 0, 1>("4 independent accumulators, 3 x s_nop 15, 2 partners (2 v_pk_fma : 1 ds_write)");
    return 0;
}
