// Shared host-side helpers for libw2e.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/w2e.h"

namespace w2e {

void set_error(const char* fmt, ...);

// Process-wide options (runtime.hip): read from the environment once at library load, changed by w2e_set_option().
struct Options {
    int conv_precision;  // 0 = exact fp32 MFMA (default); 1 = bf16x3 (W2E_CONV_PRECISION=bf16x3, opt-in)
    int deterministic;   // 1 = no fp32 atomics anywhere: ordered reductions, no split-K (W2E_DETERMINISTIC=1)
    int tune_cfg, tune_cfg_splits, tune_cfg_mode;  // force a conv tile (tests / tools/layer_bench.py); -1 = off
    int tune_upall, tune_dma, tune_fuse;  // -1 = the library's own choice, 0 never, 1 always
    int tune_mw;   // matrix waves of the fused Winograd kernel: -1 = 8 wherever N % 64 == 0; 4 = always the 32-channel form
    int tune_xcd;  // XCD-contiguous block ownership in the fused Winograd kernel: -1 / 1 = on (default), 0 = the old round-robin order (A/B).
                   // (Measured and not kept for the direct kernel: same FETCH_SIZE, same time -- profiles/r05_xcd_map_ab.txt)
    int tune_print, tune_blur, tune_gemm_s;
    int tune_skip, tune_clock;  // only honoured by a -DW2E_TUNING build (they skip work / synchronise)
};
const Options& options();

// Kernels that need more than 64 KB of dynamic LDS opt in once per (kernel, device).  Returns true when the current
// device has the attribute set.  (hipFuncSetAttribute is per device: a process driving several GPUs needs it on each.)
static inline bool big_lds_once(const void* fn, unsigned* done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return false;
    if (*done_mask & (1u << dev)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    *done_mask |= 1u << dev;
    return true;
}

// Compute units of the CURRENT device (cached per device index; 256 when the query fails): the persistent kernels size their
// grids from it.  (Round 3 kept one `static int` per call site: the count of whichever device came first, for every device.)
int cu_count();

// Argument check: records the message and makes the entry point return 1.
#define W2E_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            w2e::set_error(__VA_ARGS__); \
            return 1;                 \
        }                             \
    } while (0)

// After a kernel launch: surface launch-configuration errors without synchronising.
#define W2E_LAUNCH_CHECK(name)                                                          \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            w2e::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero `bytes` (a multiple of 4) at a 4-byte aligned device address with a KERNEL on stream s.  Used instead of
// hipMemsetAsync everywhere in the library: under stream capture (Coach.capture_step) a memset issued from here was not
// replayed with the graph on this ROCm (split-K outputs then accumulated onto the previous replay's values), while kernel
// nodes always are.
hipError_t zero_async(void* p, size_t bytes, hipStream_t s);

// Memory-bound kernels: cap the grid and grid-stride (256 CUs x 8 blocks).
static inline int stream_grid(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace w2e
