// Issue-arbitration probe for v_mfma_f32_32x32x2_f32 on gfx950: what can issue in the shadow of a saturated fp32 matrix pipe?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_probe.hip -o issue_probe && ./issue_probe      (profiles/r03_issue_probe.txt)
// Waves 0-3 of a 512-thread workgroup (one per SIMD) play role A, waves 4-7 role B; s_memtime around each wave's loop.
// What it showed: an fp32 MFMA holds its OWN wave's instruction stream for its full 64 cycles (VALU or
// ds_read placed between a wave's MFMAs add their whole issue time); a co-resident wave issues ~3 VALU per MFMA slot (22 cycles
// each); two waves that both mix MFMAs with k VALU per MFMA lose pipe time from k >= 4; dependent accumulator chains cost nothing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bits: roleA (waves 0-3), roleB (waves 4-7).  role: 0 idle, 1 MFMA-only, 2 VALU-only, 3 mixed (1 MFMA + KV VALU), 4 LDS-read only, 5 mixed MFMA + KL ds_read
template <int KV>
__device__ __forceinline__ void valu_block(float& a, float& b, float& c, float& d) {
#pragma unroll
    for (int i = 0; i < KV; ++i) {
        if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
        if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b));
        if ((i & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c));
        if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d));
    }
}

template <int ROLE_A, int ROLE_B, int KV>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, int iters, float seed) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    lds[threadIdx.x] = seed, lds[threadIdx.x + 512] = seed;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 0.5f;
    float x = seed + (threadIdx.x & 31), y = seed - (threadIdx.x & 7);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
        }
    } else if (role == 6) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
            }
        }
    } else if (role == 7) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
        }
    } else if (role == 8) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[1], 0, 0, 0);
            }
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) valu_block<16>(a, b, c, d);
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
                valu_block<KV>(a, b, c, d);
            }
        }
    } else if (role == 4) {
        const float4* l4 = reinterpret_cast<const float4*>(lds);
        float4 s = make_float4(0, 0, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((threadIdx.x & 63) * 16)), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                s.x += v.x;
            }
        }
        a += s.x;
    } else if (role == 5) {
        float4 s = make_float4(0, 0, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (KV + 3) / 4; ++q) {
                    float4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)((threadIdx.x & 63) * 16)), "n"(0));
                    s.x += v.x;
                }
            }
        }
        a += s.x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = a + b + c + d;
    for (int i = 0; i < 4; ++i) sink += acc[i][0];
    if (sink == 123.456f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + (size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB, int KV>
static void run(const char* name, unsigned long long* dout, int iters) {
    const int blocks = 256;
    std::vector<unsigned long long> h(1 + blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        probe<RA, RB, KV><<<blocks, 512>>>(dout, iters, 1.0f);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[1 + b * 8 + w];
    sa /= blocks * 4, sb /= blocks * 4;
    printf("%-44s iters %d: waves0-3 %.0f cycles (%.1f / iter), waves4-7 %.0f cycles (%.1f / iter)\n", name, iters, sa, sa / iters, sb, sb / iters);
}

int main() {
    unsigned long long* dout;
    hipMalloc(&dout, 8 * (1 + 256 * 8));
    const int N = 2000;
    // an "iter" = 4 MFMAs (256 pipe cycles), or 16 VALU, or 4 x (MFMA + KV VALU)
    run<1, 0, 0>("A: MFMA only (1 wave/SIMD)", dout, N);
    run<6, 0, 0>("A: 16 MFMA/iter, same acc 4x in a row", dout, N);
    run<6, 6, 0>("A,B: 16 MFMA/iter, same acc 4x in a row", dout, N);
    run<7, 0, 0>("A: 16 MFMA/iter, ONE accumulator chain", dout, N);
    run<7, 7, 0>("A,B: 16 MFMA/iter, ONE accumulator chain", dout, N);
    run<8, 0, 0>("A: 16 MFMA/iter, two accumulators alternating", dout, N);
    run<1, 1, 0>("A: MFMA, B: MFMA (2 waves/SIMD)", dout, N);
    run<2, 0, 0>("A: VALU only (16 fma / iter)", dout, N);
    run<1, 2, 0>("A: MFMA, B: VALU (16 fma / iter)", dout, N);
    run<3, 0, 2>("A: mixed 1 MFMA + 2 VALU", dout, N);
    run<3, 0, 4>("A: mixed 1 MFMA + 4 VALU", dout, N);
    run<3, 0, 8>("A: mixed 1 MFMA + 8 VALU", dout, N);
    run<3, 0, 12>("A: mixed 1 MFMA + 12 VALU", dout, N);
    run<3, 0, 16>("A: mixed 1 MFMA + 16 VALU", dout, N);
    run<3, 3, 4>("A,B: mixed 1 MFMA + 4 VALU (2 waves/SIMD)", dout, N);
    run<3, 3, 8>("A,B: mixed 1 MFMA + 8 VALU (2 waves/SIMD)", dout, N);
    run<1, 4, 0>("A: MFMA, B: ds_read_b128 + wait", dout, N);
    run<5, 0, 4>("A: mixed 1 MFMA + 1 ds_read_b128", dout, N);
    run<5, 0, 8>("A: mixed 1 MFMA + 2 ds_read_b128", dout, N);
    run<5, 5, 8>("A,B: mixed 1 MFMA + 2 ds_read_b128", dout, N);
    return 0;
}
