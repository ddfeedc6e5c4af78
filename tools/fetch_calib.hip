// Calibration of rocprofv3's FETCH_SIZE on gfx950 against KNOWN byte counts, in the access patterns libw2e.so's kernels use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- /tmp/fetch_calib            (and a second pass with the raw TCC counters)
//   tools/fetch_calib.py <the table this program prints> <counter_collection.csv ...>  ->  profiles/rNN_fetch_calibration.txt
//
// MI355X_MICROARCH.md (HBM): "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane,
// global_load and buffer_load ... lds alike) ... other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern".  The patterns below are the ones whose factor decided nothing last round:
//   P0  global_load_dwordx4, 1 KB contiguous per wave-instruction                      (the guide's calibrated case: control)
//   P1  buffer_load_dwordx4 ... lds, 1 KB contiguous per wave-instruction              (guide: same factor; control)
//   P2  buffer_load_dwordx4 ... lds, rows of 18 quads = 288 B starting 16 B BEFORE a 256-B boundary: the fused Winograd kernel's
//       patch rows for 64 x 8-pixel blocks (image columns bx*64-4 .. bx*64+67).  Row segments are disjoint here (every second block
//       column), so the bytes requested are known exactly: 288 per row, in 4 lines of 128 B (2 of them touched for 16 B only)
//   P3  the same with rows of 10 quads = 160 B (32 x 16-pixel blocks): 160 B touching 3 lines, the outer two shared with the next segment
//   P4  buffer_load_dword ... lds, 256 B contiguous per wave-instruction
//   P5  buffer_load_dword ... lds, lane = (pixel l >> 2, plane l & 3): 4 planes x 64 B per instruction -- the direct conv kernel's
//       channel-interleaving patch load (consecutive instructions continue each 64-B run)
//   P6  raw_buffer_load_b32 to registers, 256 B contiguous per wave-instruction         (the direct kernel's small tiles)
//   P7  global_load_dwordx2, 512 B contiguous per wave-instruction                      (the packed-fp32 FIR kernels' row loads)
// Every probe reads its OWN fresh region (nothing is in L2 / MALL from a previous probe; each region is touched exactly once).
// The program prints, per probe: bytes the lanes asked for (U), and the bytes of the distinct 128-B lines / 64-B half-lines / 32-B
// sectors those requests touch -- FETCH_SIZE x 1024 has to be read against these.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWgs = 2048, kWaves = kWgs * 4;  // 256-thread workgroups
constexpr unsigned kRow = 4096;                // bytes of an "image row" in the patch probes (1024 fp32 pixels)

// byte offset (inside a 2 GiB-limited region) that lane `lane` of wave-instruction `ins` reads, or 0xffffffff (inactive)
template <int P>
__host__ __device__ inline unsigned lane_offset(unsigned ins, int lane) {
    if (P == 0 || P == 1) return ins * 1024u + (unsigned)lane * 16u;
    if (P == 2 || P == 3) {
        constexpr unsigned Q = P == 2 ? 18 : 10;          // quads per row segment
        constexpr unsigned PER_ROW = P == 2 ? 7 : 15;      // disjoint segments per 4 KB row: block columns 2, 4, ... (every second one)
        constexpr unsigned BW = P == 2 ? 256 : 128;        // block width in bytes
        const unsigned g = ins * 64u + (unsigned)lane, seg = g / Q, quad = g % Q;
        const unsigned row = seg / PER_ROW, bx = 2u * (seg % PER_ROW) + 2u;
        return row * kRow + bx * BW - 16u + quad * 16u;
    }
    if (P == 4 || P == 6) return ins * 256u + (unsigned)lane * 4u;
    if (P == 5) {  // 4 planes of 64 MiB; instruction `ins` continues each plane's run by 64 B
        return (unsigned)(lane & 3) * (64u << 20) + ins * 64u + (unsigned)(lane >> 2) * 4u;
    }
    return ins * 512u + (unsigned)lane * 8u;  // P7
}

template <int P>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, unsigned bytes, unsigned n_ins, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned gw = blockIdx.x * 4u + (unsigned)wave;
    const uint64_t a64 = (uint64_t)(uintptr_t)src;
    i32x4 q;
    q[0] = (int)(unsigned)a64, q[1] = (int)(unsigned)((a64 >> 32) & 0xffffu), q[2] = (int)bytes, q[3] = 0x00020000;
    q[0] = __builtin_amdgcn_readfirstlane(q[0]), q[1] = __builtin_amdgcn_readfirstlane(q[1]);
    q[2] = __builtin_amdgcn_readfirstlane(q[2]), q[3] = __builtin_amdgcn_readfirstlane(q[3]);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)wave * 1024u));
    float acc = 0.f;
    (void)m0v;
    for (unsigned ins = gw; ins < n_ins; ins += (unsigned)kWaves) {
        const unsigned off = lane_offset<P>(ins, lane);
        if (P == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src) + off);
            acc += v.x + v.w;
        } else if (P == 7) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(src) + off);
            acc += v.x + v.y;
        } else if (P == 6) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (int)bytes, 0x00020000);
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            if (P == 4 || P == 5)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(q) : "memory");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(q) : "memory");
#endif
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (P != 0 && P != 6 && P != 7) acc = smem[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}

template <int P>
static void run(const char* what, size_t region_bytes, unsigned n_ins, double u, double l128, double l64, double l32) {
    float *src, *sink;
    hipMalloc(&src, region_bytes);
    hipMalloc(&sink, 64);
    // (fresh allocation, never read before: not in L2; written once by the fill below with a kernel-free memset -> it may sit in the MALL,
    // whose hits FETCH_SIZE counts all the same)
    hipMemset(src, 0, region_bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<P><<<kWgs, 256, 4096>>>(src, (unsigned)(region_bytes > 0xfffffff0ull ? 0xfffffff0ull : region_bytes), n_ins, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("P%d, %s, %.0f, %.0f, %.0f, %.0f, %.3f, %.1f\n", P, what, u, l128, l64, l32, ms, u / (ms * 1e-3) / 1e9);
    hipFree(src), hipFree(sink);
}

int main() {
    printf("# probe, pattern, bytes requested by the lanes (U), bytes of the distinct 128-B lines touched, of the 64-B half-lines, of the 32-B sectors, ms, GB/s of U\n");
    const size_t GiB = (size_t)1 << 30;
    {   // P0 / P1: 1 GiB, 1 KB per instruction
        const unsigned n = (unsigned)(GiB / 1024);
        run<0>("global_load_dwordx4 1 KB contiguous", GiB, n, (double)GiB, (double)GiB, (double)GiB, (double)GiB);
        run<1>("buffer_load_dwordx4 lds 1 KB contiguous", GiB, n, (double)GiB, (double)GiB, (double)GiB, (double)GiB);
    }
    {   // P2: rows of 4 KB, 7 disjoint 288-B segments per row; 1 GiB of rows = 262144 rows
        const unsigned rows = (unsigned)(GiB / kRow), segs = rows * 7u;
        const unsigned n = (unsigned)(((uint64_t)segs * 18u) / 64u);  // whole instructions only (the last partial one is dropped)
        const double quads = (double)n * 64.0, nseg = quads / 18.0;
        run<2>("buffer_load_dwordx4 lds rows of 288 B at 256k-16", GiB, n, quads * 16.0, nseg * 4 * 128.0, nseg * 6 * 64.0, nseg * 10 * 32.0);
    }
    {   // P3: 15 disjoint 160-B segments per row
        const unsigned rows = (unsigned)(GiB / kRow), segs = rows * 15u;
        const unsigned n = (unsigned)(((uint64_t)segs * 10u) / 64u);
        const double quads = (double)n * 64.0, nseg = quads / 10.0;
        // (distinct 128-B lines: the segment's own line + the two neighbours it touches for 16 B each, each of which it SHARES with the next
        // segment of the row -- segments sit in every second 128-B block -- so 2 lines per segment, not 3)
        run<3>("buffer_load_dwordx4 lds rows of 160 B at 128k-16", GiB, n, quads * 16.0, nseg * 2 * 128.0, nseg * 4 * 64.0, nseg * 6 * 32.0);
    }
    {   // P4 / P6: 256 MiB, 256 B per instruction
        const size_t sz = GiB / 4;
        const unsigned n = (unsigned)(sz / 256);
        run<4>("buffer_load_dword lds 256 B contiguous", sz, n, (double)sz, (double)sz, (double)sz, (double)sz);
        run<6>("raw_buffer_load_b32 256 B contiguous", sz, n, (double)sz, (double)sz, (double)sz, (double)sz);
    }
    {   // P5: 4 planes x 64 MiB, 64 B per plane per instruction
        const size_t sz = (size_t)256 << 20;
        const unsigned n = (unsigned)((64u << 20) / 64u);
        run<5>("buffer_load_dword lds 4 planes x 64 B", sz, n, (double)sz, (double)sz, (double)sz, (double)sz);
    }
    {   // P7: 512 MiB, 512 B per instruction
        const size_t sz = GiB / 2;
        const unsigned n = (unsigned)(sz / 512);
        run<7>("global_load_dwordx2 512 B contiguous", sz, n, (double)sz, (double)sz, (double)sz, (double)sz);
    }
    return 0;
}
