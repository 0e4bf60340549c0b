// Shared device/host helpers for the gfx950 (CDNA4) kernels of libvqa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VQA_F32 0
#define VQA_BF16 1

// error plumbing (api.hip owns the storage)
int vqa_set_error(int code, const char* msg);
int vqa_check_launch(const char* what);

#define VQA_REQUIRE(cond, msg) \
    do { if (!(cond)) return vqa_set_error(-22, msg); } while (0)

typedef unsigned short bf16_t;   // raw bf16 bits

using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using u16x4 = __attribute__((ext_vector_type(4))) unsigned short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
    return __uint_as_float(((unsigned int)h) << 16);
}

// round-to-nearest-even; a NaN stays a NaN (the hardware cvt does this, the
// integer trick does not)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// ---- 8-element vector access, fp32 compute --------------------------------
template <typename T> struct Vec8;

template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        u16x8 r = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(r[i]);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        u16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = f32_to_bf16(v[i]);
        *reinterpret_cast<u16x8*>(p) = r;
    }
};

template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return bf16_to_f32(x); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return f32_to_bf16(x); }

// ---- math ------------------------------------------------------------------
// tanh(x) = 1 - 2 / (exp(2x) + 1); exact limits at +-inf, abs error ~1e-7
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - __fdividef(2.0f, e + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __fdividef(1.0f, 1.0f + __expf(-x));
}

// ---- wave / block reductions (wave = 64 lanes) -------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum; scratch must hold blockDim.x / 64 floats; result to all threads
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// ---- dropout: counter-based keep decision -------------------------------------
// murmur3 finaliser of (index * golden + seed); keep iff hash >= p * 2^32.
__device__ __forceinline__ unsigned int drop_hash(unsigned int seed, unsigned int idx) {
    unsigned int x = idx * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned int drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    if (t < 0.0) t = 0.0;
    return (unsigned int)t;
}
// multiplier applied to element idx: 0 or 1/(1-p); thr == 0 means "no dropout"
__device__ __forceinline__ float drop_mult(unsigned int seed, unsigned int idx, unsigned int thr, float scale) {
    if (thr == 0u) return 1.0f;
    return drop_hash(seed, idx) >= thr ? scale : 0.0f;
}

// Sum over the split-K slabs of one element, in slab order (s = 0, 1, ...) as every consumer does.  The loads of
// the first eight slabs are issued together (a loop with a run-time trip count would wait for each load before
// the next add: one memory round trip per slab).
__device__ __forceinline__ float slab_sum(const float* __restrict__ first, int nslab, size_t stride) {
    float v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = first[(size_t)min(s, nslab - 1) * stride];
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (s < nslab) a += v[s];
    for (int s = 8; s < nslab; ++s) a += first[(size_t)s * stride];
    return a;
}
