// Library-level entry points: version, error reporting.
#include "common.h"
#include "vqa_hip.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[256] = "";

int vqa_set_error(int code, const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

// kernel launches are asynchronous: this reports launch-time failures only
// (bad configuration, missing code object), never waits for the stream
int vqa_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -(int)e - 1000;
}

extern "C" {

int vqa_version(void) { return VQA_HIP_VERSION; }

const char* vqa_last_error_string(void) { return g_err; }

}  // extern "C"
