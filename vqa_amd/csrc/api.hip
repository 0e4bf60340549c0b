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

#ifdef VQA_DEV
// Development builds: a kernel that holds `nwg` compute units' worth of 256-thread workgroups for `usec` microseconds,
// the footprint of a collective's kernel (one workgroup per channel), to measure what runs beside it.
// (96 vector registers and 16 KB of LDS per workgroup, about what a collective's kernel takes: a workgroup that needs a
// CU's whole LDS or register file cannot share the CU with it)
__global__ void __launch_bounds__(256) k_dev_occupy(unsigned long long ticks, int* sink) {
    __shared__ int hold[4096];
    hold[threadIdx.x] = (int)threadIdx.x;
    asm volatile("" ::: "v95");
    const unsigned long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == -1) sink[0] = n + hold[(threadIdx.x * 7) & 4095];
}
extern "C" int vqa_dev_occupy(int nwg, int usec, int* sink, hipStream_t stream) {
    k_dev_occupy<<<nwg, 256, 0, stream>>>((unsigned long long)usec * 100ull, sink);
    return (int)hipGetLastError();
}
#endif
