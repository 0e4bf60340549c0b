// Cost of a software grid barrier (device-scope atomic counter + spin) across one workgroup per CU, with and
// without a payload exchanged through global memory.  Spins are bounded: a barrier that does not complete sets a
// flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool gbar(unsigned* ctr, unsigned target, int* abort_flag) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        long spins = 0;
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ void __launch_bounds__(256) k_bars(unsigned* ctr, int* abort_flag, int nbar, float* buf, int payload, float* out) {
    const int nwg = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        if (payload) {       // every workgroup writes a slice, then reads its neighbour's slice after the barrier
            for (int j = threadIdx.x; j < payload; j += blockDim.x) buf[(size_t)blockIdx.x * payload + j] = (float)(i + j);
        }
        if (!gbar(ctr, (unsigned)(nwg * (i + 1)), abort_flag)) return;
        if (payload) {
            const int nb = (blockIdx.x + 37) % nwg;
            for (int j = threadIdx.x; j < payload; j += blockDim.x)
                acc += __builtin_nontemporal_load(buf + (size_t)nb * payload + j) - (float)(i + j);
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    unsigned* ctr; int* ab; float* buf; float* out;
    const int nwg = 225, payload_max = 65536;
    hipMalloc(&ctr, 4); hipMalloc(&ab, 4); hipMalloc(&buf, (size_t)nwg * payload_max * 4); hipMalloc(&out, nwg * 4);
    for (int payload : {0, 1024, 65536}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int nbar = 2000;
            hipMemset(ctr, 0, 4); hipMemset(ab, 0, 4);
            hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
            hipEventRecord(s);
            k_bars<<<nwg, 256>>>(ctr, ab, nbar, buf, payload, out);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            int h_ab; hipMemcpy(&h_ab, ab, 4, hipMemcpyDeviceToHost);
            std::vector<float> ho(nwg); hipMemcpy(ho.data(), out, nwg * 4, hipMemcpyDeviceToHost);
            float bad = 0; for (float v : ho) bad += v * v;
            printf("payload %6d floats/WG: %.2f us per barrier (abort=%d, mismatch=%g)\n", payload, ms * 1e3 / nbar, h_ab, bad);
        }
    }
    return 0;
}
