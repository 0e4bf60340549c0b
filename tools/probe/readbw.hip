// Read-bandwidth probe: how fast can a kernel stream a large buffer from HBM on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ x, size_t n16, unsigned* out) {
    // each block streams a contiguous chunk; each thread UNROLL loads in flight
    const size_t per_block = n16 / gridDim.x;
    const u32x4* p = x + (size_t)blockIdx.x * per_block;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per_block; i += UNROLL * 256) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UNROLL, bool NT>
float run(const u32x4* x, size_t n16, unsigned* out, int grid) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k_read<UNROLL, NT><<<grid, 256>>>(x, n16, out);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) k_read<UNROLL, NT><<<grid, 256>>>(x, n16, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / 5;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    u32x4* x; unsigned* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 4);
    hipMemset(x, 1, bytes);
    const size_t n16 = bytes / 16;
    int grids[] = {256, 512, 1024, 2048, 4096, 16384};
    for (int g : grids) {
        printf("grid %5d: plain u1 %.2f u4 %.2f u8 %.2f | nt u4 %.2f u8 %.2f TB/s\n", g,
               bytes / run<1, false>(x, n16, out, g) / 1e9, bytes / run<4, false>(x, n16, out, g) / 1e9,
               bytes / run<8, false>(x, n16, out, g) / 1e9, bytes / run<4, true>(x, n16, out, g) / 1e9,
               bytes / run<8, true>(x, n16, out, g) / 1e9);
    }
    return 0;
}
