// Batched column sums: problem i sums the M_i rows of X_i (fp32 or bf16) into out_i (and out2_i), 32 columns per
// block.  Shared by the stand-alone launch (elementwise.hip) and the grouped GEMM launch, which appends these blocks
// to its own (gemm.hip): one dependent launch less.
#pragma once
#include "common.h"
#include "vqa_hip.h"

struct ColsumBatch { VqaColsumDesc d[VQA_COLSUM_MAX]; int first_block[VQA_COLSUM_MAX + 1]; int n; };

__device__ __forceinline__ void colsum_block(const ColsumBatch& batch, int bid, float (*part)[33]) {
    int pi = 0;
    while (pi + 1 < batch.n && bid >= batch.first_block[pi + 1]) ++pi;
    const VqaColsumDesc d = batch.d[pi];
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = (bid - batch.first_block[pi]) * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < d.N) {
        int m = rg;
        if (d.dtype == VQA_BF16) {
            const bf16_t* X = (const bf16_t*)d.X;
            for (; m + 24 < d.M; m += 32) {
                s0 += bf16_to_f32(X[(size_t)m * d.ld + n]); s1 += bf16_to_f32(X[(size_t)(m + 8) * d.ld + n]);
                s2 += bf16_to_f32(X[(size_t)(m + 16) * d.ld + n]); s3 += bf16_to_f32(X[(size_t)(m + 24) * d.ld + n]);
            }
            for (; m < d.M; m += 8) s0 += bf16_to_f32(X[(size_t)m * d.ld + n]);
        } else {
            const float* X = (const float*)d.X;
            for (; m + 24 < d.M; m += 32) {
                s0 += X[(size_t)m * d.ld + n]; s1 += X[(size_t)(m + 8) * d.ld + n];
                s2 += X[(size_t)(m + 16) * d.ld + n]; s3 += X[(size_t)(m + 24) * d.ld + n];
            }
            for (; m < d.M; m += 8) s0 += X[(size_t)m * d.ld + n];
        }
    }
    part[rg][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && n < d.N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += part[g][cx];
        s *= d.scale;
        d.out[n] = s;
        if (d.out2) d.out2[n] = s;
    }
}

// host side: validate the problems and lay their blocks out one after the other
static inline int colsum_fill(const VqaColsumDesc* descs, int n, ColsumBatch& batch, int& blocks) {
    VQA_REQUIRE(descs && n >= 1 && n <= VQA_COLSUM_MAX, "vqa_colsum_multi: 1..8 problems");
    batch.n = n;
    blocks = 0;
    for (int i = 0; i < n; ++i) {
        const VqaColsumDesc& d = descs[i];
        VQA_REQUIRE(d.X && d.out && d.M >= 0 && d.N > 0 && d.ld >= d.N && (d.dtype == VQA_F32 || d.dtype == VQA_BF16),
                    "vqa_colsum_multi: bad problem");
        batch.d[i] = d;
        batch.first_block[i] = blocks;
        blocks += (d.N + 31) / 32;
    }
    batch.first_block[n] = blocks;
    return 0;
}
