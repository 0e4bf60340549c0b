// Fused question-conditioned attention over the region grid (gfx950).
//
// One workgroup per sample.  Forward, for sample b with R regions:
//     s_r    = w_a . tanh(P[b,r,:] + qa[b,:]) + c
//     p      = softmax_r(s)
//     pooled = sum_r p_r * inv_norm[b,r] * V[b,r,:]          (= sum_r p_r v_hat_r)
// P (R x A) and V (R x Dv) are each streamed from HBM exactly once with 16-byte
// per-lane loads; the scores, the probabilities, qa, w_a and the pooled
// accumulators live in LDS / registers for the whole kernel.
// Backward streams V once (dp_r = inv_norm_r * d_pooled . V_r), P once
// (recomputing tanh) and writes dP once.
#include "common.h"
#include "vqa_hip.h"
#include <atomic>

#define ATT_THREADS 1024
#define ATT_WAVES (ATT_THREADS / 64)
#define ATT_MAX_R 4096
#ifndef ATT_ONLINE
#define ATT_ONLINE 1
#endif

template <typename T> struct TanhSel { static __device__ __forceinline__ float f(float x) { return fast_tanh(x); } };
template <> struct TanhSel<float> { static __device__ __forceinline__ float f(float x) { return tanhf(x); } };

// dynamic LDS layout (floats): qa[A] | wa[A] | s[R] | w[R] | red[ATT_THREADS * 8]
template <typename T>
__global__ void __launch_bounds__(ATT_THREADS)
k_attn_fwd(const T* __restrict__ P, int ldP, const T* __restrict__ V, int ldV, const float* __restrict__ qa,
           int ld_qa, const float* __restrict__ w_a, const float* __restrict__ c_a,
           const float* __restrict__ inv_norm, float* __restrict__ p_out, T* __restrict__ pooled_t,
           float* __restrict__ pooled_f32, int ld_pool, int R, int A, int Dv,
           T* __restrict__ pooled_drop, unsigned seed, unsigned thr, float dscale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qa_s = smem;
    float* wa_s = qa_s + A;
    float* s_s = wa_s + A;
    float* w_s = s_s + R;
    float* red = w_s + ((R + 3) & ~3);
    __shared__ float red2[ATT_WAVES];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* Pb = P + (size_t)b * R * ldP;
    const T* Vb = V + (size_t)b * R * ldV;

    // pooling geometry: thread (grp, cv) owns vector column cv of rows grp, grp + G, ...
    const int nv = Dv >> 3;
    const int G = ATT_THREADS / nv > 0 ? ATT_THREADS / nv : 1;
    const int cpass = (nv + ATT_THREADS - 1) / ATT_THREADS;      // > 1 only when Dv > 8 * ATT_THREADS
    const int grp = tid / nv, cv = tid % nv;
    const bool pool_active = (cpass > 1) || (grp < G);

    for (int a = tid; a < A; a += ATT_THREADS) { qa_s[a] = qa[(size_t)b * ld_qa + a]; wa_s[a] = w_a[a]; }
    __syncthreads();

    // ---- phase 1: scores, one wave per region row --------------------------------
    const int nva = A >> 3;
    for (int r = wv; r < R; r += ATT_WAVES) {
        const T* row = Pb + (size_t)r * ldP;
        float acc = 0.f;
        for (int v = lane; v < nva; v += 64) {
            float x[8];
            Vec8<T>::load(row + v * 8, x);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qa_s + v * 8);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qa_s + v * 8 + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wa_s + v * 8);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wa_s + v * 8 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc += w0[i] * TanhSel<T>::f(x[i] + q0[i]);
                acc += w1[i] * TanhSel<T>::f(x[4 + i] + q1[i]);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) s_s[r] = acc + c_a[0];
    }

    // first pooling loads do not depend on the probabilities: issue them now so
    // that HBM stays busy across the softmax
    float pre[2][8];
    int pre_r[2] = {grp, grp + G};
    if (cpass == 1 && pool_active) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (pre_r[u] < R) Vec8<T>::load(Vb + (size_t)pre_r[u] * ldV + cv * 8, pre[u]);
    }
    __syncthreads();

    // ---- phase 2: softmax over regions ---------------------------------------------
    float mx = -INFINITY;
    for (int r = tid; r < R; r += ATT_THREADS) mx = fmaxf(mx, s_s[r]);
    mx = block_max(mx, red2);
    float se = 0.f;
    for (int r = tid; r < R; r += ATT_THREADS) se += expf(s_s[r] - mx);
    se = block_sum(se, red2);
    const float inv_se = 1.0f / se;
    for (int r = tid; r < R; r += ATT_THREADS) {
        const float p = expf(s_s[r] - mx) * inv_se;
        p_out[(size_t)b * R + r] = p;
        w_s[r] = p * inv_norm[(size_t)b * R + r];
    }
    __syncthreads();

    // ---- phase 3: attention-weighted pooling -----------------------------------------
    if (cpass == 1) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (pool_active) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (pre_r[u] < R) {
                    const float w = w_s[pre_r[u]];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += w * pre[u][i];
                }
            int r = grp + 2 * G;
            for (; r + 3 * G < R; r += 4 * G) {           // 4 independent loads in flight
                float x0[8], x1[8], x2[8], x3[8];
                Vec8<T>::load(Vb + (size_t)r * ldV + cv * 8, x0);
                Vec8<T>::load(Vb + (size_t)(r + G) * ldV + cv * 8, x1);
                Vec8<T>::load(Vb + (size_t)(r + 2 * G) * ldV + cv * 8, x2);
                Vec8<T>::load(Vb + (size_t)(r + 3 * G) * ldV + cv * 8, x3);
                const float w0 = w_s[r], w1 = w_s[r + G], w2 = w_s[r + 2 * G], w3 = w_s[r + 3 * G];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += w0 * x0[i] + w1 * x1[i] + w2 * x2[i] + w3 * x3[i];
            }
            for (; r < R; r += G) {
                float x0[8];
                Vec8<T>::load(Vb + (size_t)r * ldV + cv * 8, x0);
                const float w0 = w_s[r];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += w0 * x0[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) red[(size_t)(grp * nv + cv) * 8 + i] = acc[i];
        }
        __syncthreads();
        // fixed-order sum over the G row groups
        for (int d = tid; d < Dv; d += ATT_THREADS) {
            float s = 0.f;
            for (int g = 0; g < G; ++g) s += red[(size_t)g * Dv + d];
            if (pooled_f32) pooled_f32[(size_t)b * ld_pool + d] = s;
            const T st = from_f32<T>(s);
            if (pooled_t) pooled_t[(size_t)b * ld_pool + d] = st;
            if (pooled_drop)
                pooled_drop[(size_t)b * ld_pool + d] = from_f32<T>(to_f32<T>(st) * drop_mult(seed, (unsigned)(b * Dv + d), thr, dscale));
        }
    } else {
        // very wide features: each thread walks its vector columns over all rows
        for (int c = tid; c < nv; c += ATT_THREADS) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < R; ++r) {
                float x0[8];
                Vec8<T>::load(Vb + (size_t)r * ldV + c * 8, x0);
                const float w0 = w_s[r];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += w0 * x0[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (pooled_f32) pooled_f32[(size_t)b * ld_pool + c * 8 + i] = acc[i];
                const T st = from_f32<T>(acc[i]);
                if (pooled_t) pooled_t[(size_t)b * ld_pool + c * 8 + i] = st;
                if (pooled_drop)
                    pooled_drop[(size_t)b * ld_pool + c * 8 + i] =
                        from_f32<T>(to_f32<T>(st) * drop_mult(seed, (unsigned)(b * Dv + c * 8 + i), thr, dscale));
            }
        }
    }
}

// Backward.  dynamic LDS (floats): dpool[Dv] | ds[R] | wn[R] (inv_norm) | red[...]
//   dp_r  = inv_norm_r * (d_pooled . V_r)
//   ds_r  = p_r * (dp_r - sum_j p_j dp_j)
//   t     = tanh(P_r + qa);  g = ds_r * w_a * (1 - t^2)
//   dP_scaled[r, :] = g * inv_norm_r     (so that dW_va = dP_scaled^T V, V un-normalised)
//   dqa[b, :] = sum_r g ;  dwa_part[b, :] = sum_r ds_r * t ;  dc_part[b] = sum_r ds_r
template <typename T>
__global__ void __launch_bounds__(ATT_THREADS)
k_attn_bwd(const T* __restrict__ P, int ldP, const T* __restrict__ V, int ldV, const float* __restrict__ qa,
           int ld_qa, const float* __restrict__ w_a, const float* __restrict__ inv_norm,
           const float* __restrict__ p_in, const float* __restrict__ d_pooled, int ld_dpool,
           T* __restrict__ dP, int ld_dP, float* __restrict__ dqa_f32, T* __restrict__ dqa_t, int ld_dqa,
           float* __restrict__ dwa_part, float* __restrict__ dc_part, int R, int A, int Dv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dpool_s = smem;
    float* ds_s = dpool_s + Dv;
    float* wn_s = ds_s + ((R + 3) & ~3);
    float* red = wn_s + ((R + 3) & ~3);
    __shared__ float red2[ATT_WAVES];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* Pb = P + (size_t)b * R * ldP;
    const T* Vb = V + (size_t)b * R * ldV;
    T* dPb = dP + (size_t)b * R * ld_dP;

    for (int d = tid; d < Dv; d += ATT_THREADS) dpool_s[d] = d_pooled[(size_t)b * ld_dpool + d];
    for (int r = tid; r < R; r += ATT_THREADS) wn_s[r] = inv_norm[(size_t)b * R + r];
    __syncthreads();

    // ---- phase A: dp_r, one wave per region row ---------------------------------------
    const int nv = Dv >> 3;
    for (int r = wv; r < R; r += ATT_WAVES) {
        const T* row = Vb + (size_t)r * ldV;
        float acc = 0.f;
        for (int v = lane; v < nv; v += 64) {
            float x[8];
            Vec8<T>::load(row + v * 8, x);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dpool_s + v * 8);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dpool_s + v * 8 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += d0[i] * x[i] + d1[i] * x[4 + i];
        }
        acc = wave_sum(acc);
        if (lane == 0) ds_s[r] = acc * wn_s[r];          // dp_r for now
    }
    __syncthreads();

    // ---- phase B: softmax backward -----------------------------------------------------
    float dot = 0.f;
    for (int r = tid; r < R; r += ATT_THREADS) dot += p_in[(size_t)b * R + r] * ds_s[r];
    dot = block_sum(dot, red2);
    float dsum = 0.f;
    for (int r = tid; r < R; r += ATT_THREADS) {
        const float ds = p_in[(size_t)b * R + r] * (ds_s[r] - dot);
        ds_s[r] = ds;
        dsum += ds;
    }
    dsum = block_sum(dsum, red2);      // also the barrier that publishes ds_s
    if (tid == 0 && dc_part) dc_part[b] = dsum;

    // ---- phase C: stream P, write dP, accumulate dqa / dwa --------------------------------
    const int nva = A >> 3;
    const int G = ATT_THREADS / nva > 0 ? ATT_THREADS / nva : 1;
    const int grp = tid / nva, cv = tid % nva;
    for (int c0 = 0; c0 < nva; c0 += ATT_THREADS) {          // one pass unless A > 8 * ATT_THREADS
        const int c = c0 + cv;
        const bool active = (nva > ATT_THREADS) ? (c < nva) : (grp < G);
        const int rstep = (nva > ATT_THREADS) ? 1 : G;
        const int rbeg = (nva > ATT_THREADS) ? 0 : grp;
        float aq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, aw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active) {
            float qv[8], wv8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { qv[i] = qa[(size_t)b * ld_qa + c * 8 + i]; wv8[i] = w_a[c * 8 + i]; }
            for (int r = rbeg; r < R; r += rstep) {
                float x[8], o[8];
                Vec8<T>::load(Pb + (size_t)r * ldP + c * 8, x);
                const float ds = ds_s[r], wn = wn_s[r];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = TanhSel<T>::f(x[i] + qv[i]);
                    const float g = ds * wv8[i] * (1.0f - t * t);
                    aq[i] += g;
                    aw[i] += ds * t;
                    o[i] = g * wn;
                }
                Vec8<T>::store(dPb + (size_t)r * ld_dP + c * 8, o);
            }
        }
        if (nva > ATT_THREADS) {
            if (active) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    dqa_f32[(size_t)b * ld_dqa + c * 8 + i] = aq[i];
                    if (dqa_t) dqa_t[(size_t)b * ld_dqa + c * 8 + i] = from_f32<T>(aq[i]);
                    dwa_part[(size_t)b * ld_dqa + c * 8 + i] = aw[i];
                }
            }
        } else {
            // fixed-order reduction over the G row groups, dqa then dwa through one buffer
            __syncthreads();
            if (active) {
#pragma unroll
                for (int i = 0; i < 8; ++i) red[(size_t)(grp * nva + cv) * 8 + i] = aq[i];
            }
            __syncthreads();
            for (int a = tid; a < A; a += ATT_THREADS) {
                float s = 0.f;
                for (int g = 0; g < G; ++g) s += red[(size_t)g * A + a];
                dqa_f32[(size_t)b * ld_dqa + a] = s;
                if (dqa_t) dqa_t[(size_t)b * ld_dqa + a] = from_f32<T>(s);
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int i = 0; i < 8; ++i) red[(size_t)(grp * nva + cv) * 8 + i] = aw[i];
            }
            __syncthreads();
            for (int a = tid; a < A; a += ATT_THREADS) {
                float s = 0.f;
                for (int g = 0; g < G; ++g) s += red[(size_t)g * A + a];
                dwa_part[(size_t)b * ld_dqa + a] = s;
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Pipelined variants (the ones the bench shapes run): same arithmetic and summation
// order as the kernels above, with the loads of the next region row issued before the
// current one is consumed so that every wave always has HBM requests in flight.
// NVA = ceil(A / 512), NVD = ceil(Dv / 512): 16-byte vectors per lane per row.
// Requirements (checked by the launcher): Dv / 8 <= ATT_THREADS, NVA, NVD <= 4.
// ---------------------------------------------------------------------------
template <typename T> struct RawVec;
template <> struct RawVec<bf16_t> {
    typedef u16x8 type;
    static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const u16x8*>(p); }
    static __device__ __forceinline__ type zero() { return u16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    static __device__ __forceinline__ float get(const type& v, int i) { return bf16_to_f32(v[i]); }
};
template <> struct RawVec<float> {
    struct type { f32x4 a, b; };
    static __device__ __forceinline__ type load(const float* p) {
        type t; t.a = *reinterpret_cast<const f32x4*>(p); t.b = *reinterpret_cast<const f32x4*>(p + 4); return t;
    }
    static __device__ __forceinline__ type zero() { type t; t.a = f32x4{0, 0, 0, 0}; t.b = f32x4{0, 0, 0, 0}; return t; }
    static __device__ __forceinline__ float get(const type& v, int i) { return i < 4 ? v.a[i] : v.b[i - 4]; }
};

// 16-byte streaming load that does not allocate in the caches (each byte is used once)
__device__ __forceinline__ u16x8 ld_stream(const bf16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
}
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// bf16 only.  Score phase arithmetic per element: y = P*k + qa*k (k = 2 log2 e), e = exp2(y),
// r = 1/(e+1); tanh = 1 - 2r, so  w.tanh = sum(w) - sum(2w r): one conversion, two FMAs, one add and
// two transcendentals per element.
template <int NVA>
__global__ void __launch_bounds__(ATT_THREADS)
k_attn_fwd_pipe(const bf16_t* __restrict__ P, int ldP, const bf16_t* __restrict__ V, int ldV,
                const float* __restrict__ qa, int ld_qa, const float* __restrict__ w_a,
                const float* __restrict__ c_a, const float* __restrict__ inv_norm, float* __restrict__ p_out,
                bf16_t* __restrict__ pooled_t, float* __restrict__ pooled_f32, int ld_pool, int R, int A, int Dv,
                bf16_t* __restrict__ pooled_drop, unsigned seed, unsigned thr, float dscale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_s = smem;
    float* w_s = s_s + ((R + 3) & ~3);
    float* red = w_s + ((R + 3) & ~3);
    __shared__ float red2[ATT_WAVES];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16_t* Pb = P + (size_t)b * R * ldP;
    const bf16_t* Vb = V + (size_t)b * R * ldV;
    const int nva = A >> 3, nv = Dv >> 3;
    const int G = ATT_THREADS / nv, grp = tid / nv, cv = tid % nv;

    // ---- phase 1: scores ----------------------------------------------------------------
    // qa and w_a go through LDS (coalesced), then this lane's slices stay in registers.  Every
    // load below is unconditional (indices clamped, contributions of the clamped lanes are
    // multiplied by a zero weight) so that the loads of the next row stay in flight under the
    // arithmetic of the current one.
    const float kk = 2.8853900817779268f;                 // 2 * log2(e)
    for (int a = tid; a < A; a += ATT_THREADS) { red[a] = qa[(size_t)b * ld_qa + a] * kk; red[A + a] = 2.f * w_a[a]; }
    __syncthreads();
    float qk[NVA][8], w2[NVA][8];
    float wsum = 0.f;
    int voff[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int v = lane + 64 * i;
        const int vc = min(v, nva - 1);
        const float live = v < nva ? 1.f : 0.f;
        voff[i] = vc * 8;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(red + vc * 8), q1 = *reinterpret_cast<const f32x4*>(red + vc * 8 + 4);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(red + A + vc * 8), g1 = *reinterpret_cast<const f32x4*>(red + A + vc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qk[i][e] = q0[e]; qk[i][4 + e] = q1[e];
            w2[i][e] = g0[e] * live; w2[i][4 + e] = g1[e] * live;
            wsum += 0.5f * (w2[i][e] + w2[i][4 + e]);
        }
    }
    wsum = wave_sum(wsum) + c_a[0];
    __syncthreads();                                       // red is reused by the pooling phase
    u32x4_t cur[NVA], nxt[NVA];
    {
        const bf16_t* row = Pb + (size_t)min(wv, R - 1) * ldP;
#pragma unroll
        for (int i = 0; i < NVA; ++i) cur[i] = __builtin_bit_cast(u32x4_t, ld_stream(row + voff[i]));
    }
    for (int r = wv; r < R; r += ATT_WAVES) {
        const bf16_t* rown = Pb + (size_t)min(r + ATT_WAVES, R - 1) * ldP;
#pragma unroll
        for (int i = 0; i < NVA; ++i) nxt[i] = __builtin_bit_cast(u32x4_t, ld_stream(rown + voff[i]));
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NVA; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float y0 = fmaf(bf16_lo(cur[i][h]), kk, qk[i][2 * h]);
                const float y1 = fmaf(bf16_hi(cur[i][h]), kk, qk[i][2 * h + 1]);
                acc = fmaf(w2[i][2 * h], __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y0) + 1.0f), acc);
                acc = fmaf(w2[i][2 * h + 1], __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y1) + 1.0f), acc);
            }
        acc = wave_sum(acc);
        if (lane == 0) s_s[r] = wsum - acc;
#pragma unroll
        for (int i = 0; i < NVA; ++i) cur[i] = nxt[i];
    }
    __syncthreads();

    // the first 8 rows of this thread's pooling column are requested now, so that HBM stays busy
    // under the softmax (row indices clamped: every load is unconditional)
    const bool pool = grp < G;
    const bf16_t* col = Vb + (pool ? cv : 0) * 8;
    u32x4_t xc[8], xn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xc[u] = __builtin_bit_cast(u32x4_t, ld_stream(col + (size_t)min(grp + u * G, R - 1) * ldV));

    // ---- phase 2: softmax over regions ---------------------------------------------
    float mx = -INFINITY;
    for (int i = tid; i < R; i += ATT_THREADS) mx = fmaxf(mx, s_s[i]);
    mx = block_max(mx, red2);
    float se = 0.f;
    for (int i = tid; i < R; i += ATT_THREADS) se += expf(s_s[i] - mx);
    se = block_sum(se, red2);
    const float inv_se = 1.0f / se;
    for (int i = tid; i < R; i += ATT_THREADS) {
        const float p = expf(s_s[i] - mx) * inv_se;
        p_out[(size_t)b * R + i] = p;
        w_s[i] = p * inv_norm[(size_t)b * R + i];
    }
    __syncthreads();

    // ---- phase 3: pooling, two batches of 8 rows in flight per thread ---------------------------
    // (the next batch is requested before the current one is consumed: the load queue never drains)
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int rr = grp; rr < R; rr += 8 * G) {
        if (rr + 8 * G < R) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xn[u] = __builtin_bit_cast(u32x4_t, ld_stream(col + (size_t)min(rr + (8 + u) * G, R - 1) * ldV));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rr + u * G;
            const float w = row < R ? w_s[row] : 0.f;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                acc[2 * h] = fmaf(w, bf16_lo(xc[u][h]), acc[2 * h]);
                acc[2 * h + 1] = fmaf(w, bf16_hi(xc[u][h]), acc[2 * h + 1]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xc[u] = xn[u];
    }
    if (pool) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(size_t)(grp * nv + cv) * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int d = tid; d < Dv; d += ATT_THREADS) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += red[(size_t)g * Dv + d];
        if (pooled_f32) pooled_f32[(size_t)b * ld_pool + d] = s;
        const bf16_t st = f32_to_bf16(s);
        if (pooled_t) pooled_t[(size_t)b * ld_pool + d] = st;
        if (pooled_drop)
            pooled_drop[(size_t)b * ld_pool + d] = f32_to_bf16(bf16_to_f32(st) * drop_mult(seed, (unsigned)(b * Dv + d), thr, dscale));
    }
}

// Single-pass forward: scores and pooling advance together over chunks of 16 region rows (one row per wave),
// with a running maximum (online softmax), so the P stream and the V stream are in flight all the time and
// there is no idle phase between them.  Chunk k's rows are pooled while the scores of chunk k + 1 are computed;
// one barrier per chunk publishes the scores.  Every thread keeps the same running maximum m (it is a function
// of the published scores only), rescales its accumulators by exp(m_old - m_new), and its row group keeps a
// partial sum of the exponentials; the groups are combined in a fixed order at the end.
// Requirements (checked by the launcher): Dv / 8 <= 256 (at least 4 row groups), NVA <= 4.
// Small batches (fewer samples than CUs): nsplit > 1 workgroups share a sample, each taking a contiguous range of the
// row chunks with a running maximum of its own; it leaves the unnormalised pooled sums, its maximum and its sum of
// exponentials in `part` ((Dv + 2) floats per workgroup) and exp(s_r - m_split) in p_out, and k_attn_fwd_combine
// merges the ranges of a sample in a fixed order.
template <int NVA, bool SPLIT = false>
__global__ void __launch_bounds__(ATT_THREADS)
k_attn_fwd_online(const bf16_t* __restrict__ P, int ldP, const bf16_t* __restrict__ V, int ldV,
                  const float* __restrict__ qa, int ld_qa, const float* __restrict__ w_a,
                  const float* __restrict__ c_a, const float* __restrict__ inv_norm, float* __restrict__ p_out,
                  bf16_t* __restrict__ pooled_t, float* __restrict__ pooled_f32, int ld_pool, int R, int A, int Dv,
                  bf16_t* __restrict__ pooled_drop, unsigned seed, unsigned thr, float dscale,
                  const float* __restrict__ q_slabs = nullptr, int q_nslab = 0, size_t q_stride = 0,
                  const float* __restrict__ q_bias = nullptr, float* __restrict__ qa_out = nullptr,
                  int nsplit_arg = 1, float* __restrict__ part = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R4 = (R + 3) & ~3;
    const int nsplit = SPLIT ? nsplit_arg : 1;            // a compile-time 1 in the one-workgroup-per-sample kernel
    float* s_s = smem;                       // scores
    float* wn_s = s_s + R4 + 16;             // inv_norm of the sample (+16: a chunk may read past R)
    float* red = wn_s + R4 + 16;             // qa / w_a staging, then the cross-group combine
    __shared__ float l_s[ATT_THREADS];         // one partial sum of exponentials per row group (up to one group per thread)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr int CH = ATT_WAVES;            // rows per chunk

    const int b = SPLIT ? blockIdx.x / nsplit : blockIdx.x, sp = SPLIT ? blockIdx.x % nsplit : 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16_t* Pb = P + (size_t)b * R * ldP;
    const bf16_t* Vb = V + (size_t)b * R * ldV;
    const int nva = A >> 3, nv = Dv >> 3;
    const int G = ATT_THREADS / nv, grp = tid / nv, cv = tid % nv;
    const bool pool = grp < G;
    const int nch_all = (R + CH - 1) / CH;
    const int k0 = SPLIT ? nch_all * sp / nsplit : 0;                             // this workgroup's chunks [k0, nch)
    const int nch = SPLIT ? nch_all * (sp + 1) / nsplit : nch_all;

    int voff[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) voff[i] = min(lane + 64 * i, nva - 1) * 8;
    // row of this wave in chunk k: CH * k + wv; rows of this thread in chunk k: CH * k + grp + G * u, u < 4
    auto load_p = [&](int k, u32x4_t (&dst)[NVA]) {
        const bf16_t* row = Pb + (size_t)min(CH * k + wv, R - 1) * ldP;
#pragma unroll
        for (int i = 0; i < NVA; ++i) dst[i] = __builtin_bit_cast(u32x4_t, ld_stream(row + voff[i]));
    };
    const bf16_t* col = Vb + (pool ? cv : 0) * 8;
    auto load_v = [&](int k, u32x4_t (&dst)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            dst[u] = __builtin_bit_cast(u32x4_t, ld_stream(col + (size_t)min(CH * k + grp + G * u, R - 1) * ldV));
    };
    // the first chunk of both streams is requested before anything else: the staging below (and, when the query
    // projection arrives as partial sums, its reduction) runs under the latency of these loads
    u32x4_t pa[NVA];                                       // P row of the next chunk to score
    u32x4_t va[4];                                         // V rows of the next chunk to pool
    load_p(k0, pa);
    load_v(k0, va);

    const float kk = 2.8853900817779268f;                 // 2 * log2(e)
    for (int a = tid; a < A; a += ATT_THREADS) {
        float v;
        if (q_slabs) {          // the query projection is still in split-K partial sums: reduce it here and store it
            v = slab_sum(q_slabs + (size_t)b * A + a, q_nslab, q_stride);
            if (q_bias) v += q_bias[a];
            qa_out[(size_t)b * ld_qa + a] = v;
        } else {
            v = qa[(size_t)b * ld_qa + a];
        }
        red[a] = v * kk;
        red[A + a] = 2.f * w_a[a];
    }
    for (int r = tid; r < R4 + 16; r += ATT_THREADS) {
        wn_s[r] = r < R ? inv_norm[(size_t)b * R + r] : 0.f;
        s_s[r] = -INFINITY;
    }
    __syncthreads();
    float qk[NVA][8], w2[NVA][8];
    float wsum = 0.f;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int v = lane + 64 * i;
        const int vc = min(v, nva - 1);
        const float live = v < nva ? 1.f : 0.f;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(red + vc * 8), q1 = *reinterpret_cast<const f32x4*>(red + vc * 8 + 4);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(red + A + vc * 8), g1 = *reinterpret_cast<const f32x4*>(red + A + vc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qk[i][e] = q0[e]; qk[i][4 + e] = q1[e];
            w2[i][e] = g0[e] * live; w2[i][4 + e] = g1[e] * live;
            wsum += 0.5f * (w2[i][e] + w2[i][4 + e]);
        }
    }
    wsum = wave_sum(wsum) + c_a[0];
    __syncthreads();                                       // red is reused by the final combine

    auto score = [&](int k, const u32x4_t (&x)[NVA]) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NVA; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float y0 = fmaf(bf16_lo(x[i][h]), kk, qk[i][2 * h]);
                const float y1 = fmaf(bf16_hi(x[i][h]), kk, qk[i][2 * h + 1]);
                acc = fmaf(w2[i][2 * h], __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y0) + 1.0f), acc);
                acc = fmaf(w2[i][2 * h + 1], __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y1) + 1.0f), acc);
            }
        acc = wave_sum(acc);
        const int r = CH * k + wv;
        if (lane == 0 && r < R) s_s[r] = wsum - acc;
    };

    // one chunk of P rows and one chunk of V rows are always in flight (16 x 2.4 KB + 16 x 4 KB per workgroup);
    // a chunk takes several microseconds, so one chunk of look-ahead covers the HBM latency
    if (!SPLIT || k0 < nch) score(k0, pa);
    load_p(k0 + 1, pa);
    __syncthreads();

    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float m = -INFINITY, l = 0.f;
    for (int k = k0; k < nch; ++k) {
        if (k + 1 < nch) {                                 // scores of chunk k + 1, then request chunk k + 2
            score(k + 1, pa);
            load_p(k + 2, pa);
        }
        // pooling of chunk k under the running maximum
        float mk = m;
#pragma unroll
        for (int j = 0; j < CH; ++j) mk = fmaxf(mk, s_s[CH * k + j]);        // entries past R hold -inf
        const float scale = __expf(m - mk);                                   // exp(-inf) = 0 at k = 0
        m = mk;
        l *= scale;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= scale;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = grp + G * u;
            const int row = CH * k + j;
            const float e_r = (j < CH && row < R) ? __expf(s_s[row] - m) : 0.f;
            l += e_r;
            const float w = e_r * wn_s[min(row, R4 + 15)];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                acc[2 * h] = fmaf(w, bf16_lo(va[u][h]), acc[2 * h]);
                acc[2 * h + 1] = fmaf(w, bf16_hi(va[u][h]), acc[2 * h + 1]);
            }
        }
        load_v(k + 1, va);
        __syncthreads();                                   // publishes the scores of chunk k + 1
    }

    // combine the row groups in a fixed order; every group's partial sums share the final maximum m
    if (pool) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(size_t)(grp * nv + cv) * 8 + e] = acc[e];
        if (cv == 0) l_s[grp] = l;
    }
    __syncthreads();
    float lt = 0.f;
    for (int g = 0; g < G; ++g) lt += l_s[g];
    if constexpr (SPLIT) {      // one of several row ranges of the sample: leave the pieces to k_attn_fwd_combine
        float* pw = part + (size_t)blockIdx.x * (Dv + 2);
        for (int i = CH * k0 + tid; i < min(CH * nch, R); i += ATT_THREADS) p_out[(size_t)b * R + i] = __expf(s_s[i] - m);
        for (int d = tid; d < Dv; d += ATT_THREADS) {
            float sacc = 0.f;
            for (int g = 0; g < G; ++g) sacc += red[(size_t)g * Dv + d];
            pw[d] = sacc;
        }
        if (tid == 0) { pw[Dv] = m; pw[Dv + 1] = lt; }
        return;
    }
    const float inv_l = 1.0f / lt;
    for (int i = tid; i < R; i += ATT_THREADS) p_out[(size_t)b * R + i] = __expf(s_s[i] - m) * inv_l;
    for (int d = tid; d < Dv; d += ATT_THREADS) {
        float sacc = 0.f;
        for (int g = 0; g < G; ++g) sacc += red[(size_t)g * Dv + d];
        sacc *= inv_l;
        if (pooled_f32) pooled_f32[(size_t)b * ld_pool + d] = sacc;
        const bf16_t st = f32_to_bf16(sacc);
        if (pooled_t) pooled_t[(size_t)b * ld_pool + d] = st;
        if (pooled_drop)
            pooled_drop[(size_t)b * ld_pool + d] = f32_to_bf16(bf16_to_f32(st) * drop_mult(seed, (unsigned)(b * Dv + d), thr, dscale));
    }
}

// Merge of the nsplit row ranges of a sample (one workgroup per sample, ranges in index order):
//   m = max m_s;  L = sum_s l_s e^(m_s - m);  pooled = sum_s acc_s e^(m_s - m) / L;  p_r = p_r e^(m_s(r) - m) / L
__global__ void __launch_bounds__(256)
k_attn_fwd_combine(const float* __restrict__ part, int nsplit, float* __restrict__ p_out, bf16_t* __restrict__ pooled_t,
                   float* __restrict__ pooled_f32, int ld_pool, int R, int Dv, bf16_t* __restrict__ pooled_drop,
                   unsigned seed, unsigned thr, float dscale) {
    constexpr int CH = ATT_WAVES, MAXS = 8;
    __shared__ float f_s[MAXS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = part + (size_t)b * nsplit * (Dv + 2);
    const size_t pitch = (size_t)Dv + 2;
    // every range's (maximum, sum of exponentials) in one round trip: lane s of the first wave holds range s
    if (tid < 64) {
        const bool live = tid < nsplit;
        const float ms = live ? pb[(size_t)tid * pitch + Dv] : -INFINITY;
        const float ls = live ? pb[(size_t)tid * pitch + Dv + 1] : 0.f;
        const float m = wave_max(ms);
        const float e = live ? __expf(ms - m) : 0.f;
        float L = 0.f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) L += __shfl(ls * e, s, 64);           // ranges in index order
        if (tid < MAXS) f_s[tid] = e / L;
    }
    __syncthreads();
    const int nch_all = (R + CH - 1) / CH;
    for (int r = tid; r < R; r += 256) {
        const int k = r / CH;
        int s = 0;
        while (s + 1 < nsplit && nch_all * (s + 1) / nsplit <= k) ++s;         // the range whose chunks hold row r
        p_out[(size_t)b * R + r] *= f_s[s];
    }
    for (int d = tid; d < Dv; d += 256) {
        float v[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) v[s] = pb[(size_t)min(s, nsplit - 1) * pitch + d];     // all loads in flight together
        float sacc = 0.f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < nsplit) sacc += v[s] * f_s[s];
        if (pooled_f32) pooled_f32[(size_t)b * ld_pool + d] = sacc;
        const bf16_t st = f32_to_bf16(sacc);
        if (pooled_t) pooled_t[(size_t)b * ld_pool + d] = st;
        if (pooled_drop)
            pooled_drop[(size_t)b * ld_pool + d] = f32_to_bf16(bf16_to_f32(st) * drop_mult(seed, (unsigned)(b * Dv + d), thr, dscale));
    }
}

// Backward, bf16, pipelined.  Same results layout as k_attn_bwd.  NVD = ceil(Dv / 512).
// Phase C uses r = 1/(exp2(k (P+qa)) + 1):  t = 1 - 2r,  1 - t^2 = 4 r (1 - r).
// MODE 0: the whole backward of a sample in one workgroup.  Small batches split a sample's rows over nsplit workgroups
// (blockIdx = b * nsplit + range) in two launches, because the softmax backward needs sum_r p_r dp_r over ALL rows:
// MODE 1: phase A on the workgroup's rows only, dp_r stored to dp_rows (B x R);
// MODE 2: phases B and C from dp_rows: the sum over all rows is recomputed by every workgroup (R terms), dP rows of the
//         range are written, and the range's sums go to row blockIdx of dqa_f32 (stored [range][b][A]: split-K-style
//         partial sums the caller reduces), dwa_part and dc_part (the caller's column sums run over all B nsplit rows).
template <int NVD, bool NTS, int MODE = 0>
__global__ void __launch_bounds__(ATT_THREADS)
k_attn_bwd_pipe(const bf16_t* __restrict__ P, int ldP, const bf16_t* __restrict__ V, int ldV,
                const float* __restrict__ qa, int ld_qa, const float* __restrict__ w_a,
                const float* __restrict__ inv_norm, const float* __restrict__ p_in,
                const float* __restrict__ d_pooled, int ld_dpool, bf16_t* __restrict__ dP, int ld_dP,
                float* __restrict__ dqa_f32, bf16_t* __restrict__ dqa_t, int ld_dqa, float* __restrict__ dwa_part,
                float* __restrict__ dc_part, int R, int A, int Dv,
                const float* __restrict__ d_slabs = nullptr, int d_nslab = 0, size_t d_stride = 0,
                unsigned d_seed = 0, unsigned d_thr = 0, float d_scale = 1.f,
                int nsplit = 1, float* __restrict__ dp_rows = nullptr, int nbatch = 0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dpool_s = smem;
    float* ds_s = dpool_s + Dv;
    float* wn_s = ds_s + ((R + 3) & ~3);
    float* red = wn_s + ((R + 3) & ~3);
    __shared__ float red2[ATT_WAVES];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

    const int b = MODE == 0 ? blockIdx.x : blockIdx.x / nsplit, sp = MODE == 0 ? 0 : blockIdx.x % nsplit;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16_t* Pb = P + (size_t)b * R * ldP;
    const bf16_t* Vb = V + (size_t)b * R * ldV;
    bf16_t* dPb = dP + (size_t)b * R * ld_dP;
    const int nv = Dv >> 3, nva = A >> 3;
    // rows of this workgroup (MODE 0: all of them)
    const int rbeg = MODE == 0 ? 0 : (int)((long)R * sp / nsplit), rend = MODE == 0 ? R : (int)((long)R * (sp + 1) / nsplit);
    const int orow = MODE == 2 ? sp * nbatch + b : b;        // row of the range's sums in dqa (slab layout [range][b])
    const int prow = MODE == 2 ? blockIdx.x : b;             // ... and in dwa_part / dc_part (any order: they are summed)

    if (MODE != 2) {
        for (int d = tid; d < Dv; d += ATT_THREADS) {
            float v;
            if (d_slabs) {          // d pooled is still in split-K partial sums (and before its dropout): reduce it here
                v = slab_sum(d_slabs + (size_t)b * Dv + d, d_nslab, d_stride);
                v *= drop_mult(d_seed, (unsigned)(b * Dv + d), d_thr, d_scale);
            } else {
                v = d_pooled[(size_t)b * ld_dpool + d];
            }
            dpool_s[d] = v;
        }
    }
    for (int r = tid; r < R; r += ATT_THREADS) {
        wn_s[r] = inv_norm[(size_t)b * R + r];
        if (MODE == 2) ds_s[r] = dp_rows[(size_t)b * R + r];
    }
    __syncthreads();

    // ---- phase A: dp_r = inv_norm_r * (d_pooled . V_r), one wave per row, next row in flight ----
    if (MODE != 2) {
        float dv[NVD][8];
        int voff[NVD];
#pragma unroll
        for (int i = 0; i < NVD; ++i) {
            const int v = lane + 64 * i, vc = min(v, nv - 1);
            const float live = v < nv ? 1.f : 0.f;
            voff[i] = vc * 8;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dpool_s + vc * 8), d1 = *reinterpret_cast<const f32x4*>(dpool_s + vc * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dv[i][e] = d0[e] * live; dv[i][4 + e] = d1[e] * live; }
        }
        u32x4_t cur[NVD], nxt[NVD];
        {
            const bf16_t* row = Vb + (size_t)min(rbeg + wv, R - 1) * ldV;
#pragma unroll
            for (int i = 0; i < NVD; ++i) cur[i] = __builtin_bit_cast(u32x4_t, ld_stream(row + voff[i]));
        }
        for (int r = rbeg + wv; r < rend; r += ATT_WAVES) {
            const bf16_t* rown = Vb + (size_t)min(r + ATT_WAVES, R - 1) * ldV;
#pragma unroll
            for (int i = 0; i < NVD; ++i) nxt[i] = __builtin_bit_cast(u32x4_t, ld_stream(rown + voff[i]));
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NVD; ++i)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    acc = fmaf(dv[i][2 * h], bf16_lo(cur[i][h]), acc);
                    acc = fmaf(dv[i][2 * h + 1], bf16_hi(cur[i][h]), acc);
                }
            acc = wave_sum(acc);
            if (lane == 0) {
                if (MODE == 1) dp_rows[(size_t)b * R + r] = acc * wn_s[r];
                else ds_s[r] = acc * wn_s[r];
            }
#pragma unroll
            for (int i = 0; i < NVD; ++i) cur[i] = nxt[i];
        }
    }
    if (MODE == 1) return;
    __syncthreads();

    // phase C's geometry, and its first batch of P rows requested now: they do not depend on phase B
    const int G = ATT_THREADS / nva, grp = tid / nva, cv = tid % nva;
    const bool active = grp < G;
    const bf16_t* colP = Pb + (active ? cv : 0) * 8;
    u32x4_t xc[4], xn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xc[u] = __builtin_bit_cast(u32x4_t, ld_stream(colP + (size_t)min(rbeg + min(grp, G - 1) + u * G, R - 1) * ldP));

    // ---- phase B: softmax backward -----------------------------------------------------
    float dot = 0.f;
    for (int r = tid; r < R; r += ATT_THREADS) dot += p_in[(size_t)b * R + r] * ds_s[r];
    dot = block_sum(dot, red2);
    float dsum = 0.f;
    for (int r = tid; r < R; r += ATT_THREADS) {
        const float ds = p_in[(size_t)b * R + r] * (ds_s[r] - dot);
        ds_s[r] = ds;
        if (r >= rbeg && r < rend) dsum += ds;
    }
    dsum = block_sum(dsum, red2);
    if (tid == 0 && dc_part) dc_part[prow] = dsum;
    __syncthreads();

    // ---- phase C: stream P, write dP, accumulate dqa / dwa; 4 rows in flight per thread ------
    const float kk = 2.8853900817779268f;
    float aq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ar[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float dssum = 0.f;
    if (active) {
        float qk[8], w4[8];
        {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qa + (size_t)b * ld_qa + cv * 8);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qa + (size_t)b * ld_qa + cv * 8 + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(w_a + cv * 8), g1 = *reinterpret_cast<const f32x4*>(w_a + cv * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qk[e] = q0[e] * kk; qk[4 + e] = q1[e] * kk; w4[e] = 4.f * g0[e]; w4[4 + e] = 4.f * g1[e]; }
        }
        bf16_t* colD = dPb + cv * 8;
        auto body = [&](const u32x4_t& x, int r) {
            const float ds = ds_s[r], wn = wn_s[r];
            dssum += ds;
            float o[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const int e = 2 * h + z;
                    const float y = fmaf(z ? bf16_hi(x[h]) : bf16_lo(x[h]), kk, qk[e]);
                    const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f);
                    const float g = ds * w4[e] * (rr - rr * rr);
                    aq[e] += g;
                    ar[e] = fmaf(ds, rr, ar[e]);
                    o[e] = g * wn;
                }
            }
            u16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16(o[e]);
            // (a cached store while dP fits the Infinity Cache: the weight-gradient product that follows reads dP, and
            //  what of it is still there arrives sooner than from HBM: -4 us per step at B = 256, R = 196 against a
            //  non-temporal store; a larger dP is streamed past the caches)
            if (NTS) __builtin_nontemporal_store(ov, reinterpret_cast<u16x8*>(colD + (size_t)r * ld_dP));
            else *reinterpret_cast<u16x8*>(colD + (size_t)r * ld_dP) = ov;
        };
        // two batches of 4 rows in flight: the next batch is requested before the current one is consumed
        // (row indices clamped, so every load is unconditional; rows past R are not consumed)
        for (int r = rbeg + grp; r < rend; r += 4 * G) {
            if (r + 4 * G < rend) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    xn[u] = __builtin_bit_cast(u32x4_t, ld_stream(colP + (size_t)min(r + (4 + u) * G, R - 1) * ldP));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + u * G < rend) body(xc[u], r + u * G);
#pragma unroll
            for (int u = 0; u < 4; ++u) xc[u] = xn[u];
        }
    }
    // fixed-order reduction over the G row groups: dqa, then dwa = sum ds - 2 sum ds r
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(size_t)(grp * nva + cv) * 8 + e] = aq[e];
    }
    __syncthreads();
    for (int a = tid; a < A; a += ATT_THREADS) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += red[(size_t)g * A + a];
        dqa_f32[(size_t)orow * ld_dqa + a] = s;
        if (MODE == 0 && dqa_t) dqa_t[(size_t)b * ld_dqa + a] = f32_to_bf16(s);
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(size_t)(grp * nva + cv) * 8 + e] = dssum - 2.f * ar[e];
    }
    __syncthreads();
    for (int a = tid; a < A; a += ATT_THREADS) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += red[(size_t)g * A + a];
        dwa_part[(size_t)prow * ld_dqa + a] = s;
    }
}

// Dynamic LDS beyond 48 KiB has to be granted per kernel: done once per kernel instance and device (to the whole
// 160 KiB of a CU), not on every launch.
template <auto Kernel>
static int attn_reserve_lds(const char* what) {
    static std::atomic<unsigned long long> done{0};          // one bit per device, one word per kernel instance
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return vqa_set_error(-22, what);
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return 0;
    hipFuncAttributes fa;                                     // (the kernel's static LDS counts against the same 160 KiB)
    if (hipFuncGetAttributes(&fa, (const void*)Kernel) != hipSuccess) return vqa_set_error(-22, what);
    const int room = 160 * 1024 - (int)fa.sharedSizeBytes;
    if (hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess)
        return vqa_set_error(-22, what);
    if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}

extern "C" {

size_t vqa_attn_lds_bytes(int R, int A, int Dv) {
    const size_t r4 = (size_t)((R + 3) & ~3);
    const size_t fwd = (size_t)2 * A + R + r4 + (size_t)ATT_THREADS * 8;
    const size_t bwd = (size_t)Dv + 2 * r4 + (size_t)ATT_THREADS * 8;
    return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

static int attn_check(int B, int R, int A, int Dv, int ldP, int ldV) {
    VQA_REQUIRE(B >= 0 && R > 0 && A > 0 && Dv > 0, "vqa_attn: sizes must be positive");
    VQA_REQUIRE(A % 8 == 0 && Dv % 8 == 0 && ldP % 8 == 0 && ldV % 8 == 0 && ldP >= A && ldV >= Dv,
                "vqa_attn: A, Dv and leading dimensions must be multiples of 8");
    VQA_REQUIRE(R <= ATT_MAX_R, "vqa_attn: more than 4096 regions per sample");
    VQA_REQUIRE(vqa_attn_lds_bytes(R, A, Dv) <= 160 * 1024, "vqa_attn: A / Dv / R too large for LDS");
    return 0;
}

// q_slabs != nullptr: the query projection arrives as split-K partial sums (+ bias) and is written to qa
static int attn_fwd_impl(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                         const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                         int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                         float drop_p, const float* q_slabs, int q_nslab, size_t q_stride, const float* q_bias,
                         hipStream_t stream) {
    VQA_REQUIRE(P && V && qa && w_a && c_a && inv_norm && p_out && (pooled_t || pooled_f32), "vqa_attn_fwd: null pointer");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_attn_fwd: drop_p must be in [0, 1)");
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    if (int rc = attn_check(B, R, A, Dv, ldP, ldV)) return rc;
    VQA_REQUIRE(ld_qa >= A && ld_pool >= Dv, "vqa_attn_fwd: ld_qa / ld_pool too small");
    if (B == 0) return 0;
    const size_t lds = ((size_t)2 * A + R + ((R + 3) & ~3) + (size_t)ATT_THREADS * 8) * sizeof(float);
    const int nva_l = (A / 8 + 63) / 64;
    const bool online = dtype == VQA_BF16 && Dv / 8 <= 256 && nva_l <= 4 && ATT_ONLINE;
    if (q_slabs && !online) {            // the other kernels take the reduced projection: reduce it first
        if (int rc = vqa_epilogue(q_slabs, q_nslab, q_stride, B, A, q_bias, nullptr, nullptr, 0, 0, const_cast<float*>(qa),
                                  ld_qa, nullptr, 0, VQA_F32, 0, 0.f, stream)) return rc;
        q_slabs = nullptr;
    }
    if (online) {
        const size_t r4 = (size_t)((R + 3) & ~3);
        const size_t fa = (size_t)2 * A, fc = (size_t)ATT_THREADS * 8;
        const size_t lds_o = (2 * (r4 + 16) + (fa > fc ? fa : fc)) * sizeof(float);
#define ATT_FWD_ONLINE(NVA)                                                                                        \
  {                                                                                                                \
    if (lds_o > 48 * 1024)                                                                                         \
        if (int rc = attn_reserve_lds<k_attn_fwd_online<NVA>>("vqa_attn_fwd: cannot reserve LDS")) return rc;      \
    k_attn_fwd_online<NVA><<<B, ATT_THREADS, lds_o, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, \
        ld_qa, w_a, c_a, inv_norm, p_out, (bf16_t*)pooled_t, pooled_f32, ld_pool, R, A, Dv, (bf16_t*)pooled_drop, drop_seed, thr, dsc, \
        q_slabs, q_nslab, q_stride, q_bias, const_cast<float*>(qa)); \
  }
        if (nva_l == 1) ATT_FWD_ONLINE(1) else if (nva_l == 2) ATT_FWD_ONLINE(2) else if (nva_l == 3) ATT_FWD_ONLINE(3) else ATT_FWD_ONLINE(4)
#undef ATT_FWD_ONLINE
        return vqa_check_launch("vqa_attn_fwd");
    }
    if (dtype == VQA_BF16 && Dv / 8 <= ATT_THREADS && nva_l <= 4) {
        const size_t lds_p = ((size_t)2 * ((R + 3) & ~3) + (size_t)ATT_THREADS * 8) * sizeof(float);
#define ATT_FWD_PIPE(NVA)                                                                                          \
  {                                                                                                                \
    if (lds_p > 48 * 1024)                                                                                         \
        if (int rc = attn_reserve_lds<k_attn_fwd_pipe<NVA>>("vqa_attn_fwd: cannot reserve LDS")) return rc;        \
    k_attn_fwd_pipe<NVA><<<B, ATT_THREADS, lds_p, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, \
        ld_qa, w_a, c_a, inv_norm, p_out, (bf16_t*)pooled_t, pooled_f32, ld_pool, R, A, Dv, (bf16_t*)pooled_drop, drop_seed, thr, dsc); \
  }
        if (nva_l == 1) ATT_FWD_PIPE(1) else if (nva_l == 2) ATT_FWD_PIPE(2) else if (nva_l == 3) ATT_FWD_PIPE(3) else ATT_FWD_PIPE(4)
#undef ATT_FWD_PIPE
        return vqa_check_launch("vqa_attn_fwd");
    }
    if (dtype == VQA_F32) {
        if (int rc = attn_reserve_lds<k_attn_fwd<float>>("vqa_attn_fwd: cannot reserve LDS")) return rc;
        k_attn_fwd<float><<<B, ATT_THREADS, lds, stream>>>((const float*)P, ldP, (const float*)V, ldV, qa, ld_qa, w_a, c_a,
                                                          inv_norm, p_out, (float*)pooled_t, pooled_f32, ld_pool, R, A, Dv, (float*)pooled_drop, drop_seed, thr, dsc);
    } else if (dtype == VQA_BF16) {
        if (int rc = attn_reserve_lds<k_attn_fwd<bf16_t>>("vqa_attn_fwd: cannot reserve LDS")) return rc;
        k_attn_fwd<bf16_t><<<B, ATT_THREADS, lds, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, ld_qa, w_a, c_a,
                                                           inv_norm, p_out, (bf16_t*)pooled_t, pooled_f32, ld_pool, R, A, Dv, (bf16_t*)pooled_drop, drop_seed, thr, dsc);
    } else return vqa_set_error(-22, "vqa_attn_fwd: dtype");
    return vqa_check_launch("vqa_attn_fwd");
}

int vqa_attn_fwd(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                 const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                 int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                 float drop_p, hipStream_t stream) {
    return attn_fwd_impl(P, ldP, V, ldV, qa, ld_qa, w_a, c_a, inv_norm, p_out, pooled_t, pooled_f32, ld_pool, B, R, A, Dv,
                         dtype, pooled_drop, drop_seed, drop_p, nullptr, 0, 0, nullptr, stream);
}

int vqa_attn_fwd_qslabs(const void* P, int ldP, const void* V, int ldV, const float* qa_slabs, int qa_nslab,
                        size_t qa_stride, const float* qa_bias, float* qa, int ld_qa, const float* w_a,
                        const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                        int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                        float drop_p, hipStream_t stream) {
    VQA_REQUIRE(qa_slabs && qa_nslab >= 1 && (qa_nslab == 1 || qa_stride >= (size_t)B * A), "vqa_attn_fwd_qslabs: bad partial sums");
    return attn_fwd_impl(P, ldP, V, ldV, qa, ld_qa, w_a, c_a, inv_norm, p_out, pooled_t, pooled_f32, ld_pool, B, R, A, Dv,
                         dtype, pooled_drop, drop_seed, drop_p, qa_slabs, qa_nslab, qa_stride, qa_bias, stream);
}

// d_slabs != nullptr: d pooled arrives as split-K partial sums still to be multiplied by the dropout mask of
// (d_seed, d_p); d_pooled is then scratch for the kernels that need it reduced
static int attn_bwd_impl(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                         const float* inv_norm, const float* p_in, const float* d_pooled, int ld_dpool, void* dP, int ld_dP,
                         float* dqa_f32, void* dqa_t, int ld_dqa, float* dwa_part, float* dc_part, int B, int R, int A, int Dv,
                         int dtype, const float* d_slabs, int d_nslab, size_t d_stride, unsigned d_seed, float d_p,
                         hipStream_t stream) {
    VQA_REQUIRE(P && V && qa && w_a && inv_norm && p_in && d_pooled && dP && dqa_f32 && dwa_part, "vqa_attn_bwd: null pointer");
    if (int rc = attn_check(B, R, A, Dv, ldP, ldV)) return rc;
    VQA_REQUIRE(ld_qa >= A && ld_dpool >= Dv && ld_dP >= A && ld_dP % 8 == 0 && ld_dqa >= A, "vqa_attn_bwd: leading dimension too small");
    if (B == 0) return 0;
    const size_t r4 = (size_t)((R + 3) & ~3);
    const size_t lds = ((size_t)Dv + 2 * r4 + (size_t)ATT_THREADS * 8) * sizeof(float);
    const int nvd_l = (Dv / 8 + 63) / 64;
    const bool pipe = dtype == VQA_BF16 && A / 8 <= ATT_THREADS && nvd_l <= 4 && ((ld_qa & 3) == 0) && ((((uintptr_t)qa | (uintptr_t)w_a) & 15) == 0);
    VQA_REQUIRE(d_p >= 0.f && d_p < 1.f, "vqa_attn_bwd: dropout probability must be in [0, 1)");
    const unsigned d_thr = drop_threshold(d_p);
    const float d_scale = 1.0f / (1.0f - d_p);
    if (d_slabs && !pipe) {
        if (int rc = vqa_epilogue(d_slabs, d_nslab, d_stride, B, Dv, nullptr, nullptr, nullptr, 0, 0, nullptr, 0,
                                  const_cast<float*>(d_pooled), ld_dpool, VQA_F32, d_seed, d_p, stream)) return rc;
        d_slabs = nullptr;
    }
    if (pipe) {
#define ATT_BWD_PIPE_(NVD, NTS)                                                                                    \
  {                                                                                                                \
    if (lds > 48 * 1024)                                                                                           \
        if (int rc = attn_reserve_lds<k_attn_bwd_pipe<NVD, NTS>>("vqa_attn_bwd: cannot reserve LDS")) return rc;   \
    k_attn_bwd_pipe<NVD, NTS><<<B, ATT_THREADS, lds, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, ld_qa, w_a,  \
        inv_norm, p_in, d_pooled, ld_dpool, (bf16_t*)dP, ld_dP, dqa_f32, (bf16_t*)dqa_t, ld_dqa, dwa_part, dc_part, R, A, Dv, \
        d_slabs, d_nslab, d_stride, d_seed, d_thr, d_scale); \
  }
#define ATT_BWD_PIPE(NVD) { if (nts) ATT_BWD_PIPE_(NVD, true) else ATT_BWD_PIPE_(NVD, false) }
        // dP larger than the Infinity Cache (256 MB) cannot wait there for the weight-gradient product
        const bool nts = (size_t)B * R * ld_dP * 2 > ((size_t)192 << 20);
        if (nvd_l == 1) ATT_BWD_PIPE(1) else if (nvd_l == 2) ATT_BWD_PIPE(2) else if (nvd_l == 3) ATT_BWD_PIPE(3) else ATT_BWD_PIPE(4)
#undef ATT_BWD_PIPE_
#undef ATT_BWD_PIPE
        return vqa_check_launch("vqa_attn_bwd");
    }
    if (dtype == VQA_F32) {
        if (int rc = attn_reserve_lds<k_attn_bwd<float>>("vqa_attn_bwd: cannot reserve LDS")) return rc;
        k_attn_bwd<float><<<B, ATT_THREADS, lds, stream>>>((const float*)P, ldP, (const float*)V, ldV, qa, ld_qa, w_a, inv_norm,
                                                          p_in, d_pooled, ld_dpool, (float*)dP, ld_dP, dqa_f32, (float*)dqa_t,
                                                          ld_dqa, dwa_part, dc_part, R, A, Dv);
    } else if (dtype == VQA_BF16) {
        if (int rc = attn_reserve_lds<k_attn_bwd<bf16_t>>("vqa_attn_bwd: cannot reserve LDS")) return rc;
        k_attn_bwd<bf16_t><<<B, ATT_THREADS, lds, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, ld_qa, w_a, inv_norm,
                                                           p_in, d_pooled, ld_dpool, (bf16_t*)dP, ld_dP, dqa_f32, (bf16_t*)dqa_t,
                                                           ld_dqa, dwa_part, dc_part, R, A, Dv);
    } else return vqa_set_error(-22, "vqa_attn_bwd: dtype");
    return vqa_check_launch("vqa_attn_bwd");
}

// ---- small batches: a sample's rows shared by nsplit workgroups ---------------------------------------------
static bool attn_split_eligible(int R, int A, int Dv, int dtype) {
    const int nva_l = (A / 8 + 63) / 64, nvd_l = (Dv / 8 + 63) / 64;
    return dtype == VQA_BF16 && A % 8 == 0 && Dv % 8 == 0 && Dv / 8 <= 256 && nva_l <= 4 && A / 8 <= ATT_THREADS &&
           nvd_l <= 4 && R <= ATT_MAX_R && ATT_ONLINE;
}

int vqa_attn_splits(int B, int R, int A, int Dv, int dtype) {
    if (B <= 0 || R <= 0 || !attn_split_eligible(R, A, Dv, dtype)) return 1;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    int s = n_cu / B;                                  // one workgroup per CU: as many row ranges as fill the chip once
    const int nch = (R + ATT_WAVES - 1) / ATT_WAVES;   // ... of at least two 16-row chunks each
    if (s > nch / 2) s = nch / 2;
    if (s > 8) s = 8;
    return s < 1 ? 1 : s;
}

size_t vqa_attn_split_workspace_floats(int B, int R, int Dv, int nsplit) {
    return (size_t)B * nsplit * (Dv + 2) + (size_t)B * R;
}

int vqa_attn_fwd_split(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                       const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                       int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                       float drop_p, int nsplit, float* workspace, hipStream_t stream) {
    VQA_REQUIRE(P && V && qa && w_a && c_a && inv_norm && p_out && (pooled_t || pooled_f32) && workspace, "vqa_attn_fwd_split: null pointer");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_attn_fwd_split: drop_p must be in [0, 1)");
    if (int rc = attn_check(B, R, A, Dv, ldP, ldV)) return rc;
    VQA_REQUIRE(ld_qa >= A && ld_pool >= Dv, "vqa_attn_fwd_split: ld_qa / ld_pool too small");
    VQA_REQUIRE(attn_split_eligible(R, A, Dv, dtype), "vqa_attn_fwd_split: shape or dtype outside the streaming bf16 kernels (vqa_attn_splits returns 1 for it)");
    VQA_REQUIRE(nsplit >= 2 && nsplit <= 8 && nsplit <= (R + ATT_WAVES - 1) / ATT_WAVES, "vqa_attn_fwd_split: nsplit in [2, 8], at most one range per 16-row chunk");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    const size_t r4 = (size_t)((R + 3) & ~3);
    const size_t fa = (size_t)2 * A, fc = (size_t)ATT_THREADS * 8;
    const size_t lds_o = (2 * (r4 + 16) + (fa > fc ? fa : fc)) * sizeof(float);
    const int nva_l = (A / 8 + 63) / 64;
#define ATT_FWD_SPLIT(NVA)                                                                                         \
  {                                                                                                                \
    if (lds_o > 48 * 1024)                                                                                         \
        if (int rc = attn_reserve_lds<k_attn_fwd_online<NVA, true>>("vqa_attn_fwd_split: cannot reserve LDS")) return rc; \
    k_attn_fwd_online<NVA, true><<<B * nsplit, ATT_THREADS, lds_o, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, \
        ld_qa, w_a, c_a, inv_norm, p_out, nullptr, nullptr, ld_pool, R, A, Dv, nullptr, 0u, 0u, 1.f,              \
        nullptr, 0, 0, nullptr, nullptr, nsplit, workspace);                                                      \
  }
    if (nva_l == 1) ATT_FWD_SPLIT(1) else if (nva_l == 2) ATT_FWD_SPLIT(2) else if (nva_l == 3) ATT_FWD_SPLIT(3) else ATT_FWD_SPLIT(4)
#undef ATT_FWD_SPLIT
    k_attn_fwd_combine<<<B, 256, 0, stream>>>(workspace, nsplit, p_out, (bf16_t*)pooled_t, pooled_f32, ld_pool, R, Dv,
                                              (bf16_t*)pooled_drop, drop_seed, thr, dsc);
    return vqa_check_launch("vqa_attn_fwd_split");
}

int vqa_attn_bwd_split(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                       const float* inv_norm, const float* p_in, const float* d_pooled, int ld_dpool, void* dP, int ld_dP,
                       float* dqa_slabs, int ld_dqa, float* dwa_part, float* dc_part, int B, int R, int A, int Dv,
                       int dtype, int nsplit, float* workspace, hipStream_t stream) {
    VQA_REQUIRE(P && V && qa && w_a && inv_norm && p_in && d_pooled && dP && dqa_slabs && dwa_part && workspace, "vqa_attn_bwd_split: null pointer");
    if (int rc = attn_check(B, R, A, Dv, ldP, ldV)) return rc;
    VQA_REQUIRE(ld_qa >= A && ld_dpool >= Dv && ld_dP >= A && ld_dP % 8 == 0 && ld_dqa >= A, "vqa_attn_bwd_split: leading dimension too small");
    VQA_REQUIRE(attn_split_eligible(R, A, Dv, dtype) && ((ld_qa & 3) == 0) && ((((uintptr_t)qa | (uintptr_t)w_a) & 15) == 0),
                "vqa_attn_bwd_split: shape, dtype or alignment outside the streaming bf16 kernels");
    VQA_REQUIRE(nsplit >= 2 && nsplit <= 8 && nsplit <= R, "vqa_attn_bwd_split: nsplit in [2, 8]");
    if (B == 0) return 0;
    const size_t r4 = (size_t)((R + 3) & ~3);
    const size_t lds = ((size_t)Dv + 2 * r4 + (size_t)ATT_THREADS * 8) * sizeof(float);
    const int nvd_l = (Dv / 8 + 63) / 64;
    float* dp_rows = workspace + (size_t)B * nsplit * (Dv + 2);
    const bool nts = (size_t)B * R * ld_dP * 2 > ((size_t)192 << 20);
#define ATT_BWD_SPLIT_(NVD, NTS, MODE)                                                                             \
  {                                                                                                                \
    if (lds > 48 * 1024)                                                                                           \
        if (int rc = attn_reserve_lds<k_attn_bwd_pipe<NVD, NTS, MODE>>("vqa_attn_bwd_split: cannot reserve LDS")) return rc; \
    k_attn_bwd_pipe<NVD, NTS, MODE><<<B * nsplit, ATT_THREADS, lds, stream>>>((const bf16_t*)P, ldP, (const bf16_t*)V, ldV, qa, ld_qa, w_a, \
        inv_norm, p_in, d_pooled, ld_dpool, (bf16_t*)dP, ld_dP, dqa_slabs, nullptr, ld_dqa, dwa_part, dc_part, R, A, Dv, \
        nullptr, 0, 0, 0u, 0u, 1.f, nsplit, dp_rows, B);                                                           \
  }
#define ATT_BWD_SPLIT(NVD) { ATT_BWD_SPLIT_(NVD, false, 1) if (nts) ATT_BWD_SPLIT_(NVD, true, 2) else ATT_BWD_SPLIT_(NVD, false, 2) }
    if (nvd_l == 1) ATT_BWD_SPLIT(1) else if (nvd_l == 2) ATT_BWD_SPLIT(2) else if (nvd_l == 3) ATT_BWD_SPLIT(3) else ATT_BWD_SPLIT(4)
#undef ATT_BWD_SPLIT_
#undef ATT_BWD_SPLIT
    return vqa_check_launch("vqa_attn_bwd_split");
}

int vqa_attn_bwd(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                 const float* inv_norm, const float* p_in, const float* d_pooled, int ld_dpool, void* dP, int ld_dP,
                 float* dqa_f32, void* dqa_t, int ld_dqa, float* dwa_part, float* dc_part, int B, int R, int A, int Dv,
                 int dtype, hipStream_t stream) {
    return attn_bwd_impl(P, ldP, V, ldV, qa, ld_qa, w_a, inv_norm, p_in, d_pooled, ld_dpool, dP, ld_dP, dqa_f32, dqa_t,
                         ld_dqa, dwa_part, dc_part, B, R, A, Dv, dtype, nullptr, 0, 0, 0, 0.f, stream);
}

int vqa_attn_bwd_dslabs(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                        const float* inv_norm, const float* p_in, const float* dpool_slabs, int dpool_nslab,
                        size_t dpool_stride, unsigned drop_seed, float drop_p, float* d_pooled, int ld_dpool, void* dP,
                        int ld_dP, float* dqa_f32, void* dqa_t, int ld_dqa, float* dwa_part, float* dc_part, int B, int R,
                        int A, int Dv, int dtype, hipStream_t stream) {
    VQA_REQUIRE(dpool_slabs && dpool_nslab >= 1 && (dpool_nslab == 1 || dpool_stride >= (size_t)B * Dv),
                "vqa_attn_bwd_dslabs: bad partial sums");
    return attn_bwd_impl(P, ldP, V, ldV, qa, ld_qa, w_a, inv_norm, p_in, d_pooled, ld_dpool, dP, ld_dP, dqa_f32, dqa_t,
                         ld_dqa, dwa_part, dc_part, B, R, A, Dv, dtype, dpool_slabs, dpool_nslab, dpool_stride, drop_seed,
                         drop_p, stream);
}

}  // extern "C"
