// LDS tile images, LDS-DMA staging and MFMA fragment reads shared by the bf16 GEMM kernels (gfx950).
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BK 64

// ---- LDS tile images ---------------------------------------------------------------
// Both images are un-padded so that one wave-wide 16-byte LDS-DMA instruction (1 KiB,
// lane-linear) fills whole rows; bank conflicts are removed by XOR-ing the 16-byte chunk
// index with a function of the row, applied to the SOURCE address of the DMA and again
// on the fragment read.
//   k-major [ROWS][64]: 128-byte rows, chunk ^= row & 7            (ds_read_b128 fragments)
//   m-major [64][ROWS]: ROWS*2-byte rows, chunk ^= 2 * f(krow)     (ds_read_b64_tr_b16 fragments;
//            a half-wave reads k-rows {r..r+3, r+8..r+11}: f spreads them over the 8 32-byte
//            bank groups)
template <int ROWS, bool KMAJOR, int BK = 64> struct Tile;
template <int ROWS, int BK> struct Tile<ROWS, true, BK> {
    static constexpr int RBYTES = BK * 2;                   // bytes per tile row (128 or 64)
    static constexpr int CPR = BK / 8;                      // 16-byte chunks per row
    static constexpr int BYTES = ROWS * RBYTES;
    template <int NW> static constexpr int ni() { return BYTES / (NW * 1024); }   // DMA instructions per wave
    // 128-byte rows: chunk ^= row & 7.  64-byte rows: the four 16-lane groups of a ds_read_b128 fragment
    // read rows {0-3, 12-15 | 4-11} with two chunk values; chunk ^= {0, 2, 3, 1}[(row >> 2) & 3] puts
    // every group on 16 distinct 16-byte bank groups.
    static __device__ __forceinline__ int swz(int row) { return BK == 64 ? (row & 7) : ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }
    static __device__ __forceinline__ int off(int row, int c) { return row * RBYTES + ((c ^ swz(row)) << 4); }
    // slot (instruction gi, lane) -> tile row and source chunk
    static __device__ __forceinline__ void slot(int gi, int lane, int& row, int& c) {
        row = gi * (64 / CPR) + lane / CPR;
        c = (lane % CPR) ^ swz(row);
    }
};
template <int ROWS, int BK> struct Tile<ROWS, false, BK> {
    static constexpr int RB = ROWS * 2;                     // bytes per k-row
    static constexpr int CPR = ROWS / 8;                    // chunks per k-row
    static constexpr int BYTES = BK * RB;
    template <int NW> static constexpr int ni() { return BYTES / (NW * 1024); }
    static __device__ __forceinline__ int swz(int krow) {
        return ROWS >= 128 ? (((krow & 3) | (((krow >> 3) & 1) << 2)) << 1)
                           : ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);
    }
    static __device__ __forceinline__ int off(int krow, int c) { return krow * RB + ((c ^ swz(krow)) << 4); }
    static __device__ __forceinline__ void slot(int gi, int lane, int& krow, int& c) {
        constexpr int RPI = 64 / CPR;                       // k-rows per DMA instruction
        krow = gi * RPI + lane / CPR;
        c = (lane % CPR) ^ swz(krow);
    }
};

// Source pointers of this lane's DMA slots for the tile at k = 0 (row / column indices past
// the operand's extent are clamped: they only feed output rows / columns that are never stored).
template <int ROWS, bool KMAJOR, int NW = 4, int BK = 64>
__device__ __forceinline__ void dma_setup(const bf16_t* __restrict__ X, int ld, int row0, int nrows, int wave, int lane,
                                          const bf16_t* (&src)[Tile<ROWS, KMAJOR, BK>::template ni<NW>()]) {
    constexpr int NI = Tile<ROWS, KMAJOR, BK>::template ni<NW>();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int r, c;
        Tile<ROWS, KMAJOR, BK>::slot(wave * NI + i, lane, r, c);
        if (KMAJOR) {
            const int row = min(row0 + r, nrows - 1);
            src[i] = X + (size_t)row * ld + c * 8;
        } else {
            const int col = min(row0 + c * 8, ((nrows - 1) >> 3) << 3);
            src[i] = X + (size_t)r * ld + col;
        }
    }
}

template <int ROWS, bool KMAJOR, int NW = 4, int BK = 64>
__device__ __forceinline__ void dma_issue(const bf16_t* const (&src)[Tile<ROWS, KMAJOR, BK>::template ni<NW>()], size_t koff, char* lds, int wave) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int NI = Tile<ROWS, KMAJOR, BK>::template ni<NW>();
#pragma unroll
    for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + koff), (lptr_t)(lds + (wave * NI + i) * 1024), 16, 0, 0);
}

// register path (partial K tiles): global -> registers with zero fill, then -> the same LDS image
template <int ROWS, bool KMAJOR, int NW = 4, int BK = 64>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ X, int ld, int row0, int nrows, int k0, int kend,
                                          u32x4 (&r)[ROWS * BK / (512 * NW)]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS * BK / (512 * NW); ++i) {
        const int q = tid + NW * 64 * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (KMAJOR) {
            constexpr int CPR = BK / 8;
            const int row = row0 + q / CPR, k = k0 + ((q % CPR) << 3);
            if (row < nrows && k < kend) v = *reinterpret_cast<const u32x4*>(X + (size_t)row * ld + k);
        } else {
            constexpr int CPR = ROWS / 8;
            const int k = k0 + q / CPR, col = row0 + (q % CPR) * 8;
            if (k < kend && col < nrows) v = *reinterpret_cast<const u32x4*>(X + (size_t)k * ld + col);
        }
        r[i] = v;
    }
}

template <int ROWS, bool KMAJOR, int NW = 4, int BK = 64>
__device__ __forceinline__ void tile_store(char* lds, const u32x4 (&r)[ROWS * BK / (512 * NW)]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS * BK / (512 * NW); ++i) {
        const int q = tid + NW * 64 * i;
        int off;
        if (KMAJOR) off = Tile<ROWS, true, BK>::off(q / (BK / 8), q % (BK / 8));
        else off = Tile<ROWS, false, BK>::off(q / (ROWS / 8), q % (ROWS / 8));
        *reinterpret_cast<u32x4*>(lds + off) = r[i];
    }
}

// fragment of the 16 tile rows starting at r0 for K sub-step kk (32 k's)
template <int ROWS, bool KMAJOR, int BK = 64>
__device__ __forceinline__ bf16x8_t frag_read(const char* lds, int r0, int kk, int lane) {
    if (KMAJOR) {
        return *reinterpret_cast<const bf16x8_t*>(lds + Tile<ROWS, true, BK>::off(r0 + (lane & 15), kk * 4 + (lane >> 4)));
    } else {
        const int g = lane >> 4, li = lane & 15;
        const int krow = kk * 32 + 8 * g + (li >> 2);
        const int col = r0 + 4 * (li & 3);
        typedef s16x4_t __attribute__((address_space(3))) * lds_p;
        // rows krow and krow + 4 have the same swizzle (it ignores bit 2 of the row)
        const int off = Tile<ROWS, false, BK>::off(krow, col >> 3) + (col & 7) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off + 4 * Tile<ROWS, false, BK>::RB));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
}

// wait until only the N youngest vector-memory operations of this wave are in flight, then a raw barrier
// (an LDS-DMA in flight survives it; __syncthreads() would drain the queue)
#define VMCNT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); \
                              __builtin_amdgcn_s_barrier(); } while (0)
