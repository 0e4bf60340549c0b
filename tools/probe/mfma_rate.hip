// Probe: sustained bf16 MFMA rate with every SIMD issuing back to back, 16x16x32 against 32x32x16 (no memory traffic):
// what the chip's power management leaves of the 2.4 GHz peak for each instruction shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int SHAPE>
__global__ void __launch_bounds__(512) k_mfma(float* out, int iters, float seed) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
    } else {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    }
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE>
int run(float* out, int waves_per_cu) {
    const int iters = 20000, grid = 256, threads = 64 * waves_per_cu;
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    k_mfma<SHAPE><<<grid, threads>>>(out, iters / 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(s));
    for (int r = 0; r < 3; ++r) k_mfma<SHAPE><<<grid, threads>>>(out, iters, 1.0f);
    CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e));
    // per wave and iteration: 16x16x32: 8 MFMAs x 16384 flop; 32x32x16: 4 MFMAs x 32768 flop  (131072 either way)
    const double flop = 3.0 * iters * 131072.0 * grid * waves_per_cu;
    printf("shape %2d  waves/CU %d : %7.2f ms  %7.1f TFLOP/s\n", SHAPE, waves_per_cu, ms, flop / ms / 1e9);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 4));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<16>(out, 4)) return 1;
        if (run<32>(out, 4)) return 1;
        if (run<16>(out, 8)) return 1;
        if (run<32>(out, 8)) return 1;
    }
    return 0;
}
