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
int m3_abi_version(void) { return 2002; }

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

}  // extern "C"
