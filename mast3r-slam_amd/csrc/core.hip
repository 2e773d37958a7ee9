// ABI version, status strings and the thread-local HIP error text of libm3slam_hip.so.
#include "common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

void m3_set_hip_error(hipError_t e, const char *where) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}

extern "C" {

// 2000 (round 3): m3_chol_solve's workspace is m3_chol_ws_doubles(dim) (was 1 + dim); m3_track_* info carries the
// solver-failure flag; the RoPE GEMM entry points take (rope_tok, tokens_per_image, rope_cols) since 1001 -> callers built
// against a 1xxx header must be rebuilt (tests assert the exact value).
int m3_abi_version(void) { return 2007; }

const char *m3_status_string(int status) {
    switch (status) {
        case M3_OK: return "ok";
        case M3_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or unsupported value)";
        case M3_ERR_LAUNCH: return "HIP launch/runtime error (see m3_last_hip_error)";
        case M3_ERR_UNSUPPORTED: return "problem size not supported by this kernel";
        default: return "unknown status";
    }
}

const char *m3_last_hip_error(void) { return g_err; }

// Compute units of the current device (hipDeviceProp_t.multiProcessorCount; 256 on an MI355X in SPX mode, fewer in a
// CPX / NPS partition), read once per device.  Grid-size thresholds ("does this launch fill the chip") derive from it.
int m3_device_cu_count(void) {
    static std::atomic<int> cached[64];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return 256;
    int v = cached[d].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
    cached[d].store(prop.multiProcessorCount, std::memory_order_relaxed);
    return prop.multiProcessorCount;
}

}  // extern "C"
