// Library-level state of libw2e.so: the thread-local error string and the process-wide options.
//
// Options are read from the environment ONCE, when the library is loaded, and afterwards change only through
// w2e_set_option() -- no entry point calls getenv() on its launch path.  They are the only global mutable state of
// the library besides the per-device "large dynamic LDS enabled" flags of the kernels (see big_lds_once()).
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace w2e {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static Options g_opt;

static int parse_precision(const char* v) { return (v && strcmp(v, "bf16x3") == 0) ? 1 : 0; }

static bool apply(Options& o, const char* name, const char* value) {
    const int iv = value ? atoi(value) : 0;
    if (!strcmp(name, "conv_precision")) o.conv_precision = parse_precision(value);
    else if (!strcmp(name, "deterministic")) o.deterministic = iv != 0;
    else if (!strcmp(name, "tune_cfg")) {  // "<cfg>[,<splits>[,<mode>]]"; "" or "-1" = off
        o.tune_cfg = -1, o.tune_cfg_splits = 1, o.tune_cfg_mode = -1;
        if (value && *value) sscanf(value, "%d,%d,%d", &o.tune_cfg, &o.tune_cfg_splits, &o.tune_cfg_mode);
    } else if (!strcmp(name, "tune_upall")) o.tune_upall = (value && *value) ? iv : -1;
    else if (!strcmp(name, "tune_dma")) o.tune_dma = (value && *value) ? iv : -1;
    else if (!strcmp(name, "tune_print")) o.tune_print = iv;
    else if (!strcmp(name, "tune_blur")) o.tune_blur = iv;
    else if (!strcmp(name, "tune_gemm_s")) o.tune_gemm_s = iv;
    else if (!strcmp(name, "tune_fuse")) o.tune_fuse = (value && *value) ? iv : -1;
    else if (!strcmp(name, "tune_xcd")) o.tune_xcd = (value && *value) ? iv : -1;
    else if (!strcmp(name, "tune_mw")) o.tune_mw = (value && *value) ? iv : -1;
#ifdef W2E_TUNING
    else if (!strcmp(name, "tune_skip")) o.tune_skip = iv;
    else if (!strcmp(name, "tune_clock")) o.tune_clock = iv;
#endif
    else return false;
    return true;
}

static Options from_env() {
    Options o{};
    o.tune_cfg = -1, o.tune_cfg_splits = 1, o.tune_cfg_mode = -1, o.tune_upall = -1, o.tune_dma = -1, o.tune_fuse = -1, o.tune_xcd = -1, o.tune_mw = -1;
    static const char* const kEnv[][2] = {
        {"W2E_CONV_PRECISION", "conv_precision"}, {"W2E_DETERMINISTIC", "deterministic"}, {"W2E_TUNE_CFG", "tune_cfg"},
        {"W2E_TUNE_UPALL", "tune_upall"},         {"W2E_TUNE_DMA", "tune_dma"},           {"W2E_TUNE_PRINT", "tune_print"},
        {"W2E_TUNE_BLUR", "tune_blur"},           {"W2E_TUNE_GEMM_S", "tune_gemm_s"},     {"W2E_TUNE_FUSE", "tune_fuse"},
        {"W2E_TUNE_SKIP", "tune_skip"},           {"W2E_TUNE_CLOCK", "tune_clock"},       {"W2E_TUNE_XCD", "tune_xcd"},
        {"W2E_TUNE_MW", "tune_mw"}};
    for (const auto& e : kEnv)
        if (const char* v = getenv(e[0])) apply(o, e[1], v);
    return o;
}

struct OptionsInit {
    OptionsInit() { g_opt = from_env(); }
};
static OptionsInit g_opt_init;  // runs when the shared library is loaded

const Options& options() { return g_opt; }

int cu_count() {
    static int cached[32];  // 0 = not asked yet; a racing first call stores the same value twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
    if (!cached[dev]) {
        hipDeviceProp_t prop;
        cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cached[dev];
}

__global__ void zero_kernel(unsigned* __restrict__ p, size_t words) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += step) p[i] = 0u;
}

__global__ void zero4_kernel(uint4* __restrict__ p, size_t quads) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += step) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

hipError_t zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) != 0 || ((uintptr_t)p & 3) != 0) return hipErrorInvalidValue;
    if (((uintptr_t)p & 15) == 0 && (bytes & 15) == 0) {
        const size_t quads = bytes >> 4;
        zero4_kernel<<<stream_grid((int64_t)quads, 256), 256, 0, s>>>(reinterpret_cast<uint4*>(p), quads);
    } else {
        const size_t words = bytes >> 2;
        zero_kernel<<<stream_grid((int64_t)words, 256), 256, 0, s>>>(reinterpret_cast<unsigned*>(p), words);
    }
    return hipGetLastError();
}

}  // namespace w2e

extern "C" {

int w2e_version(void) { return W2E_VERSION; }
const char* w2e_last_error(void) { return w2e::g_err; }

int w2e_set_option(const char* name, const char* value) {
    W2E_REQUIRE(name != nullptr, "set_option: null name");
    W2E_REQUIRE(w2e::apply(w2e::g_opt, name, value), "set_option: unknown option '%s'", name);
    return 0;
}

int w2e_get_option(const char* name, int* value) {
    W2E_REQUIRE(name && value, "get_option: null argument");
    const w2e::Options& o = w2e::g_opt;
    if (!strcmp(name, "conv_precision")) *value = o.conv_precision;
    else if (!strcmp(name, "deterministic")) *value = o.deterministic;
    else if (!strcmp(name, "tune_cfg")) *value = o.tune_cfg;
    else if (!strcmp(name, "tune_fuse")) *value = o.tune_fuse;
    else if (!strcmp(name, "tune_blur")) *value = o.tune_blur;
    else if (!strcmp(name, "tuning_build")) {
#ifdef W2E_TUNING
        *value = 1;
#else
        *value = 0;
#endif
    } else W2E_REQUIRE(false, "get_option: unknown option '%s'", name);
    return 0;
}

}  // extern "C"
