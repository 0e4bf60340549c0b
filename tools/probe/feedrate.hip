// Operand feed rate of one 512-thread workgroup per CU in the access pattern of the 256 x 256 GEMM's K loop
// (k-major operands: one wave instruction = 8 rows x 128 bytes, rows `pitch` bytes apart; a K tile = 64 such
// instructions = 64 KB, 32 K tiles per output tile), three ways:
//   dma   global_load_lds_dwordx4 into an LDS ring (what the GEMM does)
//   reg   global_load_dwordx4 into registers, discarded
//   regl  global_load_dwordx4 into registers, then ds_write_b128 into the LDS ring
// `inflight` K tiles are requested ahead.  Every workgroup walks its own row tile of A (M x K bf16) and a column
// tile of B shared with the workgroups of the same column, like the projection (196 x 5 tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)

constexpr int KT = 32;          // K tiles per output tile (K = 2048)

template <int MODE, int INFLIGHT>
__global__ void __launch_bounds__(512) k_feed(const char* __restrict__ A, const char* __restrict__ B, int pitch,
                                              int tiles_m, int tiles_n, unsigned* out, int xcd_order) {
    __shared__ __attribute__((aligned(1024))) char lds[160 * 1024 - 1024];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (int tile0 = blockIdx.x; tile0 < tiles_m * tiles_n; tile0 += gridDim.x) {
        // xcd_order: the tiles_n column tiles of a row tile go to workgroups of one XCD (blockIdx % 8), one after another
        int tile = tile0;
        if (xcd_order) {
            const int x = tile0 & 7, l = tile0 >> 3;
            tile = ((l / tiles_n) * 8 + x) * tiles_n + l % tiles_n;
            if (tile >= tiles_m * tiles_n) tile = tile0;
        }
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        // per K tile: A 256 rows x 128 B and B 256 rows x 128 B = 64 instructions of 8 rows; wave w issues 8 of them
        const char* src[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ins = wave * 8 + i;                 // 0..63: 0..31 A, 32..63 B
            const int row = (ins & 31) * 8 + (lane >> 3);
            const char* base = ins < 32 ? A + (size_t)(tm * 256 + row) * pitch : B + (size_t)(tn * 256 + row) * pitch;
            src[i] = base + (lane & 7) * 16;
        }
        if (MODE == 0) {
            // ring of 2 slots of 64 KB
            for (int t = 0; t < KT + INFLIGHT; ++t) {
                if (t < KT) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + t * 128), (lptr_t)(lds + ((t & 1) * 64 + wave * 8 + i) * 1024), 16, 0, 0);
                }
                if (t >= INFLIGHT) {
                    if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
        } else {
            u32x4 r[INFLIGHT + 1][8];
#pragma unroll
            for (int t = 0; t < INFLIGHT; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) r[t][i] = *reinterpret_cast<const u32x4*>(src[i] + t * 128);
            for (int t0 = 0; t0 < KT; t0 += INFLIGHT + 1) {
#pragma unroll
                for (int u = 0; u <= INFLIGHT; ++u) {
                    const int t = t0 + u;
                    if (t + INFLIGHT < KT) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            r[(u + INFLIGHT) % (INFLIGHT + 1)][i] = *reinterpret_cast<const u32x4*>(src[i] + (t + INFLIGHT) * 128);
                    }
                    if (t < KT) {
                        if (MODE == 1) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc += r[u][i][0] ^ r[u][i][3];
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                *reinterpret_cast<u32x4*>(lds + ((t & 1) * 64 + wave * 8 + i) * 1024 + lane * 16) = r[u][i];
                            __builtin_amdgcn_s_barrier();
                        }
                    }
                }
            }
            if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4); }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int INFLIGHT>
float run(const char* A, const char* B, int pitch, int tiles_m, int tiles_n, unsigned* out, int grid, int xcd_order) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k_feed<MODE, INFLIGHT><<<grid, 512>>>(A, B, pitch, tiles_m, tiles_n, out, xcd_order);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    for (int i = 0; i < 5; ++i) k_feed<MODE, INFLIGHT><<<grid, 512>>>(A, B, pitch, tiles_m, tiles_n, out, xcd_order);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / 5;
}

int main() {
    const int tiles_m = 196, tiles_n = 5, pitch = 4096;
    char* A; char* B; unsigned* out;
    CK(hipMalloc(&A, (size_t)tiles_m * 256 * pitch)); CK(hipMalloc(&B, (size_t)tiles_n * 256 * pitch)); CK(hipMalloc(&out, 4));
    CK(hipMemset(A, 1, (size_t)tiles_m * 256 * pitch)); CK(hipMemset(B, 1, (size_t)tiles_n * 256 * pitch));
    const double gb = (double)tiles_m * tiles_n * KT * 65536 / 1e9;
#define R(M, F, name, grid, xo) { float ms = run<M, F>(A, B, pitch, tiles_m, tiles_n, out, grid, xo); \
    printf("%-5s inflight %d grid %4d xcd_order %d: %7.1f us  %6.2f TB/s  %5.1f GB/s per active CU\n", name, F, grid, xo, ms * 1e3, \
           gb / ms, gb / ms * 1e3 / (grid < 256 ? grid : 256)); }
    for (int xo = 0; xo < 2; ++xo) {
        R(0, 2, "dma", 980, xo) R(0, 2, "dma", 256, xo) R(0, 2, "dma", 128, xo) R(0, 2, "dma", 64, xo) R(0, 2, "dma", 8, xo)
        R(1, 2, "reg", 980, xo) R(1, 2, "reg", 256, xo) R(1, 2, "reg", 128, xo) R(1, 2, "reg", 64, xo) R(1, 2, "reg", 8, xo)
    }
    return 0;
}
