// Shared host-side helpers for libw2e.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/w2e.h"

namespace w2e {

void set_error(const char* fmt, ...);

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

// Memory-bound kernels: cap the grid and grid-stride (256 CUs x 8 blocks).
static inline int stream_grid(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace w2e
