// Visual front-ends of the reference's two ResNet-feature models, on precomputed feature maps (gfx950):
//   * multi-scale mean pooling over the H x W grid with question gates
//     (VQAGAPModel, vqa_model/models/model.py:161-172)
//   * 1x1 convolution to Dv / (H W) channels followed by a channel-major flatten
//     (VQAModel, vqa_model/models/model.py:53-54, 61-62)
// All kernels are HBM-streaming: 16-byte accesses, fp32 accumulation, fixed summation order.
#include "common.h"
#include "vqa_hip.h"
#include <mutex>

// ---------------------------------------------------------------------------
// grid mean, rows layout: x is (rows = B * C) x HW contiguous (an NCHW map); out[b, col0 + c].
// A group of LPR lanes (power of two, <= 64) owns one row; a wave owns 64 / LPR rows.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_grid_mean_rows(const T* __restrict__ x, float* __restrict__ out, long rows,
                                                        int C, int HW, int ld_out, int col0, int lpr) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;                                    // rows per wave
    const int sub = lane / lpr, li = lane % lpr;
    const long nwaves = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = HW >> 3;                                      // full 8-element vectors per row
    const float inv = 1.0f / (float)HW;
    for (long base = w0 * rpw; base < rows; base += nwaves * rpw) {
        const long row = base + sub;
        const bool live = row < rows;
        const T* p = x + (size_t)(live ? row : rows - 1) * HW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int v = li;
        if ((((size_t)p) & 15) == 0) {
            for (; v + 3 * lpr < nv; v += 4 * lpr) {             // 4 independent 16-byte loads in flight per lane
                float t0[8], t1[8], t2[8], t3[8];
                Vec8<T>::load(p + (size_t)v * 8, t0);
                Vec8<T>::load(p + (size_t)(v + lpr) * 8, t1);
                Vec8<T>::load(p + (size_t)(v + 2 * lpr) * 8, t2);
                Vec8<T>::load(p + (size_t)(v + 3 * lpr) * 8, t3);
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0 += t0[e]; a1 += t1[e]; a2 += t2[e]; a3 += t3[e]; }
            }
            for (; v < nv; v += lpr) {
                float t0[8];
                Vec8<T>::load(p + (size_t)v * 8, t0);
#pragma unroll
                for (int e = 0; e < 8; ++e) a0 += t0[e];
            }
            for (int i = nv * 8 + li; i < HW; i += lpr) a1 += to_f32<T>(p[i]);
        } else {
            for (int i = li; i < HW; i += lpr) a0 += to_f32<T>(p[i]);
        }
        float s = (a0 + a1) + (a2 + a3);
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (live && li == 0) {
            const long b = row / C;
            const int c = (int)(row % C);
            out[(size_t)b * ld_out + col0 + c] = s * inv;
        }
    }
}

// ---------------------------------------------------------------------------
// grid mean, channels-last layout: x is (B, HW, C), C % 8 == 0.  Block = 32 channel vectors x 32 row
// groups; grid = (C / 256 rounded up, nsplit, B).  Each block sums rows [split * chunk, ...) of one
// sample; nsplit == 1 writes the mean, otherwise partial sums land in part[(split * B + b) * C + c]
// and k_grid_mean_finish adds them in split order.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) k_grid_mean_cl(const T* __restrict__ x, float* __restrict__ out,
                                                       float* __restrict__ part, int B, int C, int HW, int ld_out,
                                                       int col0, int nsplit) {
    __shared__ float red[32][33 * 8];
    const int cv = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cv) * 8;
    const int b = blockIdx.z, split = blockIdx.y;
    const int chunk = (HW + nsplit - 1) / nsplit;
    const int r_lo = split * chunk, r_hi = min(HW, r_lo + chunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const T* p = x + (size_t)b * HW * C + c0;
        int r = r_lo + rg;
        for (; r + 96 < r_hi; r += 128) {
            float t0[8], t1[8], t2[8], t3[8];
            Vec8<T>::load(p + (size_t)r * C, t0);
            Vec8<T>::load(p + (size_t)(r + 32) * C, t1);
            Vec8<T>::load(p + (size_t)(r + 64) * C, t2);
            Vec8<T>::load(p + (size_t)(r + 96) * C, t3);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (t0[e] + t1[e]) + (t2[e] + t3[e]);
        }
        for (; r < r_hi; r += 32) {
            float t0[8];
            Vec8<T>::load(p + (size_t)r * C, t0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += t0[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cv * 8 + e + cv / 4] = acc[e];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int ch = threadIdx.x;                               // channel within the block's 256
        const int c = blockIdx.x * 256 + ch;
        if (c < C) {
            float s = 0.f;
#pragma unroll 8
            for (int g = 0; g < 32; ++g) s += red[g][ch + (ch >> 3) / 4];
            if (nsplit == 1) out[(size_t)b * ld_out + col0 + c] = s * (1.0f / (float)HW);
            else part[((size_t)split * B + b) * C + c] = s;
        }
    }
}

__global__ void __launch_bounds__(256) k_grid_mean_finish(const float* __restrict__ part, float* __restrict__ out,
                                                          int B, int C, int HW, int ld_out, int col0, int nsplit) {
    const long total = (long)B * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * total + i];
        out[(size_t)(i / C) * ld_out + col0 + (int)(i % C)] = s * (1.0f / (float)HW);
    }
}

// ---------------------------------------------------------------------------
// gates:  g = sigmoid(sum_s slab + bias);  gate[b, n] = g;  out[b, n] = mean[b, n] * g * dropout
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_gate_fwd(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                           const float* __restrict__ bias, const float* __restrict__ mean, int ld_mean,
                           float* __restrict__ gate, int ld_gate, T* __restrict__ out, int ld_out, int B, int N,
                           int accurate, unsigned seed, unsigned thr, float dscale) {
    const long total = (long)B * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), n = (int)(i % N);
        float z = bias ? bias[n] : 0.f;
        for (int s = 0; s < nslab; ++s) z += slabs[s * slab_stride + i];
        const float g = accurate ? 1.0f / (1.0f + expf(-z)) : fast_sigmoid(z);
        gate[(size_t)b * ld_gate + n] = g;
        out[(size_t)b * ld_out + n] = from_f32<T>(mean[(size_t)b * ld_mean + n] * g * drop_mult(seed, (unsigned)i, thr, dscale));
    }
}

// dz = d_out * mean * g * (1 - g)      (d_out already carries the dropout multiplier)
template <typename T>
__global__ void k_gate_bwd(const float* __restrict__ d_out, int ld_d, const float* __restrict__ mean, int ld_mean,
                           const float* __restrict__ gate, int ld_gate, T* __restrict__ dz, int ld_dz, int B, int N) {
    const long total = (long)B * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), n = (int)(i % N);
        const float g = gate[(size_t)b * ld_gate + n];
        dz[(size_t)b * ld_dz + n] = from_f32<T>(d_out[(size_t)b * ld_d + n] * mean[(size_t)b * ld_mean + n] * g * (1.0f - g));
    }
}

// ---------------------------------------------------------------------------
// weighted stack of K gated vectors, then max and mean over the stack (model.py:1238-1241):
//   st_k = w[k] * sig[b, k, d];  out[b, d] = max_k st_k;  out[b, D + d] = mean_k st_k   (both times dropout)
//   arg[b, d] = the k of the maximum (first one on ties, as torch.max)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_stack_maxavg_fwd(const float* __restrict__ sig, const float* __restrict__ w, int ld_w, T* __restrict__ out,
                                   int ld_out, unsigned char* __restrict__ arg, int B, int D, int K,
                                   unsigned seed, unsigned thr, float dscale) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / D;
        const int d = (int)(i % D);
        float mx = -INFINITY, sum = 0.f;
        int am = 0;
        for (int k = 0; k < K; ++k) {
            const float v = w[(size_t)k * ld_w] * sig[((size_t)b * K + k) * D + d];
            sum += v;
            if (v > mx) { mx = v; am = k; }
        }
        arg[i] = (unsigned char)am;
        const unsigned i_max = (unsigned)(b * 2 * D + d), i_avg = i_max + (unsigned)D;   // index in the (B, 2D) output
        out[(size_t)b * ld_out + d] = from_f32<T>(mx * drop_mult(seed, i_max, thr, dscale));
        out[(size_t)b * ld_out + D + d] = from_f32<T>(sum * (1.0f / (float)K) * drop_mult(seed, i_avg, thr, dscale));
    }
}

// backward: d_st_k = d_out[b, d] [k == arg] + d_out[b, D + d] / K;
//   dz[b, k, d] = d_st_k * w[k] * s (1 - s)   (pre-activation of the gate's sigmoid)
//   dw_part[b, k] = sum_d d_st_k * s          (one block per sample, fixed-order reduction)
template <typename T>
__global__ void __launch_bounds__(256) k_stack_maxavg_bwd(const float* __restrict__ d_out, int ld_d,
                                  const float* __restrict__ sig, const float* __restrict__ w, int ld_w,
                                  const unsigned char* __restrict__ arg, T* __restrict__ dz,
                                  float* __restrict__ dw_part, int D, int K) {
    __shared__ float red[8];
    const int b = blockIdx.x;
    for (int k = 0; k < K; ++k) {
        const float wk = w[(size_t)k * ld_w];
        float part = 0.f;
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            const float s = sig[((size_t)b * K + k) * D + d];
            const float dst = (arg[(size_t)b * D + d] == k ? d_out[(size_t)b * ld_d + d] : 0.f) +
                              d_out[(size_t)b * ld_d + D + d] * (1.0f / (float)K);
            dz[((size_t)b * K + k) * D + d] = from_f32<T>(dst * wk * s * (1.0f - s));
            part += dst * s;
        }
        part = block_sum(part, red);
        if (threadIdx.x == 0) dw_part[(size_t)b * K + k] = part;
    }
}

// ---------------------------------------------------------------------------
// channel-major flatten of a (B*R, c) product:  out[b, ch * R + r] = y[b * R + r, ch] * dropout(b * c * R + ch * R + r)
// (the .view(-1, VISUAL_OUT) of an NCHW tensor, model.py:61) and its transpose for the backward pass.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_flatten_cm(const T* __restrict__ y, int ld_y, T* __restrict__ out, int ld_out, int B, int R, int c,
                             unsigned seed, unsigned thr, float dscale) {
    const long per = (long)c * R, total = (long)B * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per;
        const int j = (int)(i % per), r = j / c, ch = j % c;     // read-coalesced walk over y
        const long o = (long)ch * R + r;
        const float v = to_f32<T>(y[((size_t)b * R + r) * ld_y + ch]);
        out[(size_t)b * ld_out + o] = from_f32<T>(v * drop_mult(seed, (unsigned)(b * per + o), thr, dscale));
    }
}

template <typename T>
__global__ void k_unflatten_cm(const float* __restrict__ d_out, int ld_d, T* __restrict__ dy, int ld_dy, int B, int R, int c) {
    const long total = (long)B * R * ld_dy;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long br = i / ld_dy;
        const int ch = (int)(i % ld_dy);
        const long b = br / R;
        const int r = (int)(br % R);
        dy[i] = from_f32<T>(ch < c ? d_out[(size_t)b * ld_d + (size_t)ch * R + r] : 0.f);
    }
}

// ===========================================================================
// C ABI
// ===========================================================================
static inline int grid_for(long total, int block, int cap = 2048) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Scratch of vqa_grid_mean_static: split partial sums, [split][b][c] floats.  In host code this name is only a
// handle for hipGetSymbolAddress; kernels receive the per-device address that call returned.
static __device__ __align__(256) float grid_mean_static_scratch[VQA_GRID_MEAN_STATIC_FLOATS];
static constexpr int kMaxDevices = 64;
static float* g_static_scratch_ptr[kMaxDevices] = {};
static std::mutex g_static_scratch_mutex;

#define DISPATCH_T(dtype, CALL)                                      \
    if ((dtype) == VQA_F32) { using T = float; CALL; }              \
    else if ((dtype) == VQA_BF16) { using T = bf16_t; CALL; }       \
    else return vqa_set_error(-22, "dtype must be VQA_F32 or VQA_BF16");

extern "C" {

int vqa_grid_mean_splits(int layout, int B, int C, int HW) {
    if (layout != 1 || B <= 0 || C <= 0 || HW <= 0) return 1;
    const long blocks = (long)B * ((C + 255) / 256);
    int s = 1;
    while (blocks * s < 1024 && HW / (s * 2) >= 128 && s < 64) s *= 2;
    return s;
}

int vqa_grid_mean(const void* x, int layout, int B, int C, int HW, float* out, int ld_out, int col0,
                  float* partial, int nsplit, int dtype, hipStream_t stream) {
    VQA_REQUIRE(x && out && B >= 0 && C > 0 && HW > 0, "vqa_grid_mean: bad arguments");
    VQA_REQUIRE(col0 >= 0 && col0 + C <= ld_out, "vqa_grid_mean: columns [col0, col0 + C) must fit in ld_out");
    VQA_REQUIRE(layout == 0 || layout == 1, "vqa_grid_mean: layout must be 0 (B,C,HW) or 1 (B,HW,C)");
    if (B == 0) return 0;
    if (layout == 0) {
        const long rows = (long)B * C;
        int lpr = 1;
        while (lpr < 64 && lpr * 8 < HW) lpr *= 2;
        const long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
        const unsigned grid = (unsigned)((waves + 3) / 4 < 4096 ? (waves + 3) / 4 : 4096);
        DISPATCH_T(dtype, (k_grid_mean_rows<T><<<grid, 256, 0, stream>>>((const T*)x, out, rows, C, HW, ld_out, col0, lpr)));
        return vqa_check_launch("vqa_grid_mean");
    }
    VQA_REQUIRE(C % 8 == 0, "vqa_grid_mean: channels-last layout needs C % 8 == 0");
    VQA_REQUIRE(nsplit >= 1 && nsplit <= 64 && (nsplit == 1 || partial), "vqa_grid_mean: nsplit in [1, 64]; nsplit > 1 needs the partial buffer");
    VQA_REQUIRE(B <= 65535, "vqa_grid_mean: B must be <= 65535 in the channels-last layout");
    const dim3 grid((C + 255) / 256, nsplit, B);
    DISPATCH_T(dtype, (k_grid_mean_cl<T><<<grid, 1024, 0, stream>>>((const T*)x, out, partial, B, C, HW, ld_out, col0, nsplit)));
    if (nsplit > 1)
        k_grid_mean_finish<<<grid_for((long)B * C, 256), 256, 0, stream>>>(partial, out, B, C, HW, ld_out, col0, nsplit);
    return vqa_check_launch("vqa_grid_mean");
}

// vqa_grid_mean for callers that cannot allocate: the split partial sums of the channels-last layout live in a
// file-scope device array instead of a caller-provided workspace.  Every GPU holds its own copy of the array at
// its own address; the address is looked up once per device (first call with that device current) and kept.
int vqa_grid_mean_static(const void* x, int layout, int B, int C, int HW, float* out, int ld_out, int col0,
                         int dtype, hipStream_t stream) {
    VQA_REQUIRE(x && out && B >= 0 && C > 0 && HW > 0, "vqa_grid_mean_static: bad arguments");
    if (B == 0) return 0;
    const int nsplit = vqa_grid_mean_splits(layout, B, C, HW);
    float* part = nullptr;
    if (nsplit > 1) {
        if ((long)nsplit * B * C > (long)VQA_GRID_MEAN_STATIC_FLOATS)
            return vqa_set_error(-22, "vqa_grid_mean_static: nsplit * B * C partial sums exceed the static scratch "
                                      "(VQA_GRID_MEAN_STATIC_FLOATS floats); use vqa_grid_mean with a workspace");
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
            return vqa_set_error(-1000 - (int)e, "vqa_grid_mean_static: hipGetDevice failed or device index >= 64");
        // the scratch used is the CURRENT device's copy, and the kernels run on the device that owns `stream`: the
        // two must be the same GPU (the null stream belongs to the current device)
        if (stream) {
            hipDevice_t sdev = -1;
            e = hipStreamGetDevice(stream, &sdev);
            if (e != hipSuccess || (int)sdev != dev)
                return vqa_set_error(-22, "vqa_grid_mean_static: `stream` does not belong to the current device "
                                          "(make the stream's GPU current before the call)");
        }
        {
            std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
            if (!g_static_scratch_ptr[dev]) {
                void* p = nullptr;
                e = hipGetSymbolAddress(&p, HIP_SYMBOL(grid_mean_static_scratch));
                if (e != hipSuccess || !p)
                    return vqa_set_error(-1000 - (int)e, "vqa_grid_mean_static: hipGetSymbolAddress(scratch) failed");
                g_static_scratch_ptr[dev] = (float*)p;
            }
            part = g_static_scratch_ptr[dev];
        }
    }
    // every partial sum the finish kernel reads is written by this call's first kernel: nothing to clear
    return vqa_grid_mean(x, layout, B, C, HW, out, ld_out, col0, part, nsplit, dtype, stream);
}

int vqa_gate_fwd(const float* slabs, int nslab, size_t slab_stride, const float* bias, const float* mean, int ld_mean,
                 float* gate, int ld_gate, void* out, int ld_out, int B, int N, unsigned seed, float p, int dtype,
                 hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && mean && gate && out && B >= 0 && N > 0, "vqa_gate_fwd: bad arguments");
    VQA_REQUIRE(ld_mean >= N && ld_gate >= N && ld_out >= N && p >= 0.f && p < 1.f, "vqa_gate_fwd: bad sizes");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_gate_fwd<T><<<grid_for((long)B * N, 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, bias, mean, ld_mean, gate, ld_gate, (T*)out, ld_out, B, N, dtype == VQA_F32, seed, thr, ds)));
    return vqa_check_launch("vqa_gate_fwd");
}

int vqa_gate_bwd(const float* d_out, int ld_d, const float* mean, int ld_mean, const float* gate, int ld_gate,
                 void* dz, int ld_dz, int B, int N, int dtype, hipStream_t stream) {
    VQA_REQUIRE(d_out && mean && gate && dz && B >= 0 && N > 0, "vqa_gate_bwd: bad arguments");
    VQA_REQUIRE(ld_d >= N && ld_mean >= N && ld_gate >= N && ld_dz >= N, "vqa_gate_bwd: bad sizes");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_gate_bwd<T><<<grid_for((long)B * N, 256), 256, 0, stream>>>(
        d_out, ld_d, mean, ld_mean, gate, ld_gate, (T*)dz, ld_dz, B, N)));
    return vqa_check_launch("vqa_gate_bwd");
}

int vqa_stack_maxavg_fwd(const float* sig, const float* w, int ld_w, void* out, int ld_out, unsigned char* argmax, int B, int D,
                         int K, unsigned seed, float p, int dtype, hipStream_t stream) {
    VQA_REQUIRE(sig && w && ld_w >= 1 && out && argmax && B >= 0 && D > 0 && K >= 1 && K <= 255 && ld_out >= 2 * D && p >= 0.f && p < 1.f,
                "vqa_stack_maxavg_fwd: bad arguments");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_stack_maxavg_fwd<T><<<grid_for((long)B * D, 256), 256, 0, stream>>>(
        sig, w, ld_w, (T*)out, ld_out, argmax, B, D, K, seed, thr, ds)));
    return vqa_check_launch("vqa_stack_maxavg_fwd");
}

int vqa_stack_maxavg_bwd(const float* d_out, int ld_d, const float* sig, const float* w, int ld_w, const unsigned char* argmax,
                         void* dz, float* dw_part, int B, int D, int K, int dtype, hipStream_t stream) {
    VQA_REQUIRE(d_out && sig && w && ld_w >= 1 && argmax && dz && dw_part && B >= 0 && D > 0 && K >= 1 && K <= 255 && ld_d >= 2 * D,
                "vqa_stack_maxavg_bwd: bad arguments");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_stack_maxavg_bwd<T><<<B, 256, 0, stream>>>(d_out, ld_d, sig, w, ld_w, argmax, (T*)dz, dw_part, D, K)));
    return vqa_check_launch("vqa_stack_maxavg_bwd");
}

int vqa_flatten_cm(const void* y, int ld_y, void* out, int ld_out, int B, int R, int c, unsigned seed, float p,
                   int dtype, hipStream_t stream) {
    VQA_REQUIRE(y && out && B >= 0 && R > 0 && c > 0, "vqa_flatten_cm: bad arguments");
    VQA_REQUIRE(ld_y >= c && (long)ld_out >= (long)c * R && p >= 0.f && p < 1.f, "vqa_flatten_cm: bad sizes");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_flatten_cm<T><<<grid_for((long)B * R * c, 256), 256, 0, stream>>>(
        (const T*)y, ld_y, (T*)out, ld_out, B, R, c, seed, thr, ds)));
    return vqa_check_launch("vqa_flatten_cm");
}

int vqa_unflatten_cm(const float* d_out, int ld_d, void* dy, int ld_dy, int B, int R, int c, int dtype,
                     hipStream_t stream) {
    VQA_REQUIRE(d_out && dy && B >= 0 && R > 0 && c > 0, "vqa_unflatten_cm: bad arguments");
    VQA_REQUIRE(ld_dy >= c && (long)ld_d >= (long)c * R, "vqa_unflatten_cm: bad sizes");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_unflatten_cm<T><<<grid_for((long)B * R * ld_dy, 256), 256, 0, stream>>>(
        d_out, ld_d, (T*)dy, ld_dy, B, R, c)));
    return vqa_check_launch("vqa_unflatten_cm");
}

}  // extern "C"
