// Error plumbing of libaldi_hip.so (no exceptions cross the C ABI).
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int aldi_set_error(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return ALDI_ERR_HIP;
}
int aldi_set_error_msg(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
extern "C" const char* aldi_last_error(void) { return g_err; }
extern "C" int aldi_version(void) { return 1; }
