// Cost of a software grid barrier built from one flag word per workgroup (no contended atomic): every workgroup
// publishes its step number in its own word and one wave polls all words with wide loads.  Measured with the
// exchange pattern of a weight-stationary recurrent step: each of NWG workgroups writes a 24-byte piece of every
// live row of h (ping-pong buffers), then reads ALL of h after the barrier.  Spins are bounded: a barrier that
// does not complete sets an abort word and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool gbar_flags(unsigned* flags, int nwg, unsigned epoch, int* abort_flag) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int good = 1;
        long spins = 0;
        for (;;) {
            bool all = true;
            for (int i = lane; i < nwg; i += 64)
                all &= __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
            if (__all(all)) break;
            if (++spins > 2000000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// hierarchical counter barrier: workgroups count into one of 8 group counters, the last arriver of a group counts
// into the top counter, everybody polls the top counter
__device__ __forceinline__ bool gbar_tree(unsigned* ctr, int nwg, unsigned epoch, int* abort_flag) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int g = blockIdx.x & 7;
        const unsigned members = (nwg - g + 7) / 8;
        const unsigned prev = __hip_atomic_fetch_add(ctr + 16 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == members * epoch) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * epoch) {
            if (++spins > 4000000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// h buffers: [2][rows][PITCH] 16-bit words; workgroup w owns columns 12 w .. 12 w + 11 of every row
constexpr int PITCH = 2432;

template <int KIND>
__global__ void __launch_bounds__(256) k_steps(unsigned* sync, int* abort_flag, int nstep, unsigned short* h, int rows,
                                               int read_all, unsigned* out) {
    const int nwg = gridDim.x;
    unsigned acc = 0;
    for (int s = 0; s < nstep; ++s) {
        unsigned short* hw = h + (size_t)(s & 1) * rows * PITCH;
        // write this workgroup's piece of every row: 12 halfwords = three 8-byte stores per row
        for (int i = threadIdx.x; i < rows * 3; i += 256) {
            const int r = i / 3, q = i % 3;
            const unsigned v = (unsigned)(s + 1) * 0x00010001u;
            uint2 val = make_uint2(v, v);
            *reinterpret_cast<uint2*>(hw + (size_t)r * PITCH + 12 * blockIdx.x + 4 * q) = val;
        }
        const bool fine = KIND == 0 ? gbar_flags(sync, nwg, (unsigned)(s + 1), abort_flag)
                                    : gbar_tree(sync, nwg, (unsigned)(s + 1), abort_flag);
        if (!fine) return;
        if (read_all) {
            // read everything the grid wrote (columns 0 .. 12 nwg - 1 rounded down to 16-byte pieces)
            const int pieces = (12 * nwg) / 8;
            const u32x4* src = reinterpret_cast<const u32x4*>(hw);
            for (int i = threadIdx.x; i < rows * pieces; i += 256) {
                const int r = i / pieces, p = i % pieces;
                u32x4 v = src[(size_t)r * (PITCH / 8) + p];
                const unsigned want = (unsigned)(s + 1) * 0x00010001u;
                acc += (v[0] != want) + (v[1] != want) + (v[2] != want) + (v[3] != want);
            }
        }
    }
    if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 1, 1u);
}

template <int KIND>
int run(const char* name, int nwg, int rows, int read_all) {
    unsigned* sync; int* ab; unsigned short* h; unsigned* out;
    CK(hipMalloc(&sync, 4096)); CK(hipMalloc(&ab, 4)); CK(hipMalloc(&h, (size_t)2 * 256 * PITCH * 2)); CK(hipMalloc(&out, 8));
    for (int rep = 0; rep < 2; ++rep) {
        const int nstep = 1000;
        CK(hipMemset(sync, 0, 4096)); CK(hipMemset(ab, 0, 4)); CK(hipMemset(out, 0, 8));
        CK(hipMemset(h, 0, (size_t)2 * 256 * PITCH * 2));
        hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
        CK(hipEventRecord(s));
        k_steps<KIND><<<nwg, 256>>>(sync, ab, nstep, h, rows, read_all, out);
        CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
        float ms; CK(hipEventElapsedTime(&ms, s, e));
        int h_ab; unsigned h_out[2];
        CK(hipMemcpy(&h_ab, ab, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_out, out, 8, hipMemcpyDeviceToHost));
        printf("%-6s nwg %3d rows %3d read_all %d: %6.2f us per step (abort=%d, stale words seen=%u, finished=%u)\n", name, nwg,
               rows, read_all, ms * 1e3 / nstep, h_ab, h_out[0], h_out[1]);
    }
    CK(hipFree(sync)); CK(hipFree(ab)); CK(hipFree(h)); CK(hipFree(out));
    return 0;
}

int main() {
    for (int nwg : {200, 150}) {
        for (int rows : {0, 32, 128, 256}) {
            if (run<0>("flags", nwg, rows, 0)) return 1;
            if (rows && run<0>("flags", nwg, rows, 1)) return 1;
        }
        if (run<1>("tree", nwg, 0, 0)) return 1;
        if (run<1>("tree", nwg, 256, 1)) return 1;
    }
    return 0;
}
