// HBM-bound row / elementwise kernels of the VQA step (gfx950).
//
// Every kernel streams its operands once with 16-byte (bf16) / 32-byte (fp32)
// per-lane accesses and computes in fp32.  Storage type T is float or raw bf16.
#include "common.h"
#include "vqa_hip.h"
#include "colsum.h"

template <typename T> struct Act {
    static __device__ __forceinline__ float tanh_(float x) { return fast_tanh(x); }
    static __device__ __forceinline__ float sigm_(float x) { return fast_sigmoid(x); }
};
template <> struct Act<float> {   // fp32 storage = the 1e-5 parity mode: libm-accurate
    static __device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
    static __device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + expf(-x)); }
};

// ---------------------------------------------------------------------------
// question side
// ---------------------------------------------------------------------------
__global__ void k_process_lengths(const int64_t* __restrict__ tok, int* __restrict__ len, int B, int L) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int z = 0;
    for (int t = 0; t < L; ++t) z += (tok[(size_t)b * L + t] == 0);
    len[b] = L - z;
}

// Packed-sequence bookkeeping, one workgroup:
//   len[b]   = L - #(tok[b,:] == 0)
//   steps[b] = len[b] (an all-pad row takes all L steps: its last-step index -1 wraps to L - 1)
//   perm     = rows ordered by descending steps, ties in row order (perm[j] = original row of sorted row j)
//   m_t[t]   = rows still running at step t (a prefix of the sorted order); offs = its exclusive prefix sum,
//              offs[L] = total number of (row, step) pairs
__global__ void __launch_bounds__(1024) k_sequence_pack(const int64_t* __restrict__ tok, int B, int L,
                                                        int* __restrict__ len, int* __restrict__ perm,
                                                        int* __restrict__ m_t, int* __restrict__ offs,
                                                        long n_rows, int* __restrict__ status) {
    extern __shared__ int sm[];
    int* steps = sm;                 // [B]   zero count, then step count
    int* above = steps + B;          // [L+1] above[n] = rows with more than n steps (n = 0..L)
    int* run = above + L + 1;        // [L+1] rows with exactly n steps placed by earlier chunks
    int* wcnt = run + L + 1;         // [16][L+1] per-wave counts of the current chunk
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    for (int b = tid; b < B; b += blockDim.x) steps[b] = 0;
    for (int n = tid; n <= L; n += blockDim.x) { above[n] = 0; run[n] = 0; }
    __syncthreads();
    bool bad = false;
    for (long i = tid; i < (long)B * L; i += blockDim.x) {            // coalesced pass over the tokens
        const int64_t v = tok[i];
        if (v == 0) atomicAdd(&steps[i / L], 1);
        bad |= v < 0 || v >= n_rows;
    }
    if (status && n_rows > 0 && bad) status[VQA_STATUS_BAD_TOKEN] = 1;   // (same value from every lane that sees one)
    __syncthreads();
    for (int b = tid; b < B; b += blockDim.x) {
        const int z = steps[b];
        len[b] = L - z;
        const int n = z == L ? L : L - z;
        steps[b] = n;
        atomicAdd(&run[n], 1);                                        // histogram (moved to `above` below)
    }
    __syncthreads();
    for (int n = tid; n <= L; n += blockDim.x) {
        int c = 0;
        for (int k = n + 1; k <= L; ++k) c += run[k];
        above[n] = c;
    }
    __syncthreads();
    for (int n = tid; n <= L; n += blockDim.x) run[n] = 0;
    for (int t = tid; t < L; t += blockDim.x) m_t[t] = above[t];
    if (tid == 0) {
        int o = 0;
        for (int t = 0; t < L; ++t) { offs[t] = o; o += above[t]; }
        offs[L] = o;
    }
    __syncthreads();
    // stable placement, 1024 rows at a time: position = rows with more steps + equal-step rows before it
    for (int b0 = 0; b0 < B; b0 += blockDim.x) {
        const int b = b0 + tid;
        const int n = b < B ? steps[b] : 0;                            // 0 never matches (steps >= 1)
        int before = 0;
        for (int k = 1; k <= L; ++k) {
            const unsigned long long mask = __ballot(n == k);
            if (n == k) before = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[w * (L + 1) + k] = __popcll(mask);
        }
        __syncthreads();
        if (b < B) {
            int pos = above[n] + run[n] + before;
            for (int ww = 0; ww < w; ++ww) pos += wcnt[ww * (L + 1) + n];
            perm[pos] = b;
        }
        __syncthreads();
        for (int k = 1 + tid; k <= L; k += blockDim.x) {
            int c = 0;
            for (int ww = 0; ww < nw; ++ww) c += wcnt[ww * (L + 1) + k];
            run[k] += c;
        }
        __syncthreads();
    }
}

// packed gather: row offs[t] + j of out = table[tok[perm[j], t]] for j < m_t[t]
template <typename T>
__global__ void k_embedding_gather_packed(const int64_t* __restrict__ tok, const T* __restrict__ table,
                                          T* __restrict__ out, int B, int L, int width, int ld_table,
                                          int ld_out, int n_rows, const int* __restrict__ perm,
                                          const int* __restrict__ offs, int act, int* __restrict__ ptok) {
    const int vec_per_row = width >> 3;
    const long total = (long)offs[L] * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % vec_per_row);
        const int prow = (int)(i / vec_per_row);
        int t = 0;
        while (t + 1 < L && offs[t + 1] <= prow) ++t;
        const int b = perm[prow - offs[t]];
        int64_t id = tok[(size_t)b * L + t];
        if (id < 0 || id >= n_rows) id = 0;
        if (ptok && v == 0) ptok[prow] = (int)id;
        float x[8];
        Vec8<T>::load(table + (size_t)id * ld_table + v * 8, x);
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = Act<T>::tanh_(x[e]);
        }
        Vec8<T>::store(out + (size_t)prow * ld_out + v * 8, x);
    }
}

// out[row(b,t), 0:width] = table[tok[b,t], 0:width]; width % 8 == 0.
// time_major: row = t * B + b, else row = b * L + t.  Ids outside the table
// read the pad row 0.
template <typename T>
__global__ void k_embedding_gather(const int64_t* __restrict__ tok, const T* __restrict__ table,
                                   T* __restrict__ out, int B, int L, int width, int ld_table,
                                   int ld_out, int n_rows, int time_major) {
    const int vec_per_row = width >> 3;
    const long total = (long)B * L * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % vec_per_row);
        const long bt = i / vec_per_row;
        const int b = (int)(bt / L), t = (int)(bt % L);
        int64_t id = tok[bt];
        if (id < 0 || id >= n_rows) id = 0;
        const long orow = time_major ? ((long)t * B + b) : bt;
        float x[8];
        Vec8<T>::load(table + (size_t)id * ld_table + v * 8, x);
        Vec8<T>::store(out + (size_t)orow * ld_out + v * 8, x);
    }
}

// One LSTM time step after the recurrent GEMM:  gates = sum_s slab[s] + xg_t
// (xg_t already holds x_t W_ih^T + b_ih + b_hh), order i, f, g, o.
// Writes c (fp32, in place), h (T, operand of the next step's GEMM) and, for
// rows whose last valid token is this step, q (fp32 and T).
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        const u16x4 r = *reinterpret_cast<const u16x4*>(p);
        return f32x4{bf16_to_f32(r[0]), bf16_to_f32(r[1]), bf16_to_f32(r[2]), bf16_to_f32(r[3])};
    }
    static __device__ __forceinline__ void store(bf16_t* p, const f32x4& v) {
        *reinterpret_cast<u16x4*>(p) = u16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
    }
};

// 4 consecutive hidden units per thread (H % 4 == 0, leading dimensions % 4 == 0): 16-byte slab loads
template <typename T>
__global__ void __launch_bounds__(256) k_lstm_cell(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                            const T* __restrict__ xg, int ld_xg, float* __restrict__ c,
                            T* __restrict__ h, int ld_h, const int* __restrict__ len,
                            float* __restrict__ q32, T* __restrict__ qT, int ld_q,
                            int B, int H, int t, int L, const int* __restrict__ perm,
                            const int* __restrict__ m_t, const int* __restrict__ offs,
                            T* __restrict__ hs, int ld_hs, T* __restrict__ gates_out, float* __restrict__ cs_out,
                            T* __restrict__ q_drop, unsigned seed, unsigned thr, float dscale,
                            const int64_t* __restrict__ tok = nullptr, int n_table = 0,
                            const float* __restrict__ hh_table = nullptr) {
    // packed mode (perm != null): state rows are in sorted order, only the first m_t[t] are live, their
    // input-projection rows start at row offs[t] of xg and q goes to the original row perm[b].
    // table mode (tok != null, packed): xg is a table of input projections per token id (a frozen embedding
    // folded into the frozen W_ih) and the row of sorted row b is xg[tok[perm[b], t]].  hh_table (table mode, t >= 1,
    // fp32, same shape): the recurrent product of the previous step's output tabulated per previous token id --
    // exact only where that output is a function of one token, i.e. at t = 1 (h_0 depends on the first token alone).
    const int H4 = H >> 2;
    const int rows = perm ? m_t[t] : B;
    const long xrow0 = perm ? offs[t] : 0;
    const long total = (long)rows * H4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H4), j = (int)(i % H4) * 4;
        const int ob = perm ? perm[b] : b;
        long xrow = xrow0 + b;
        if (tok) {
            int64_t id = tok[(size_t)ob * L + t];
            if (id < 0 || id >= n_table) id = 0;            // as the gather: ids outside the table read the pad row
            xrow = (long)id;
        }
        f32x4 g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = Vec4<T>::load(xg + (size_t)xrow * ld_xg + k * H + j);
        if (hh_table) {
            int64_t pid = tok[(size_t)ob * L + t - 1];
            if (pid < 0 || pid >= n_table) pid = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] += Vec4<float>::load(hh_table + (size_t)pid * 4 * H + k * H + j);
        }
        // the loads of the first four slabs are issued together (a loop with a run-time trip count waits for each)
        if (nslab > 0) {
            f32x4 sv[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* sl = slabs + (size_t)min(s, nslab - 1) * slab_stride + (size_t)b * 4 * H + j;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[s][k] = Vec4<float>::load(sl + k * H);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < nslab) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] += sv[s][k];
                }
            for (int s = 4; s < nslab; ++s) {
                const float* sl = slabs + s * slab_stride + (size_t)b * 4 * H + j;
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] += Vec4<float>::load(sl + k * H);
            }
        }
        f32x4 cprev = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t != 0) cprev = Vec4<float>::load(c + (size_t)b * H + j);
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = Act<T>::sigm_(g[0][e]), fg = Act<T>::sigm_(g[1][e]);
            const float gg = Act<T>::tanh_(g[2][e]), og = Act<T>::sigm_(g[3][e]);
            cn[e] = fg * cprev[e] + ig * gg;
            hn[e] = og * Act<T>::tanh_(cn[e]);
            g[0][e] = ig; g[1][e] = fg; g[2][e] = gg; g[3][e] = og;
        }
        if (gates_out) {                       // saved for backward through time, packed rows
#pragma unroll
            for (int k = 0; k < 4; ++k) Vec4<T>::store(gates_out + (size_t)(xrow0 + b) * 4 * H + k * H + j, g[k]);
            Vec4<float>::store(cs_out + (size_t)(xrow0 + b) * H + j, cn);
        }
        Vec4<float>::store(c + (size_t)b * H + j, cn);
        Vec4<T>::store(h + (size_t)b * ld_h + j, hn);
        if (hs) Vec4<T>::store(hs + (size_t)(xrow0 + b) * ld_hs + j, hn);     // every step's output, packed rows
        int last = len[ob] - 1;
        if (last < 0) last += L;               // all-pad row: negative index -> last step
        if (last == t) {
            Vec4<float>::store(q32 + (size_t)ob * ld_q + j, hn);
            Vec4<T>::store(qT + (size_t)ob * ld_q + j, hn);
            if (q_drop) {                      // dropout of the stored (rounded) question vector, mask of a (B, H) matrix
                f32x4 qd;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    qd[e] = to_f32<T>(from_f32<T>(hn[e])) * drop_mult(seed, (unsigned)(ob * H + j + e), thr, dscale);
                Vec4<T>::store(q_drop + (size_t)ob * ld_q + j, qd);
            }
        }
    }
}

// Steps 0 and 1 of a frozen first layer in one launch: neither needs a recurrent GEMM when the input projections
// (xg_table) and the first recurrent product (hh_table) are tabulated per token id.  Same arithmetic and the same
// outputs as k_lstm_cell in table mode at t = 0 followed by t = 1 (packed rows; rows live at step 1 are a prefix
// of the rows live at step 0).
template <typename T>
__global__ void __launch_bounds__(256) k_lstm_cell01_table(const T* __restrict__ xg, int ld_xg, const float* __restrict__ hh_table,
                            int n_table, const int64_t* __restrict__ tok, float* __restrict__ c, T* __restrict__ h, int ld_h,
                            const int* __restrict__ len, float* __restrict__ q32, T* __restrict__ qT, int ld_q,
                            int H, int L, const int* __restrict__ perm, const int* __restrict__ m_t,
                            const int* __restrict__ offs, T* __restrict__ hs, int ld_hs,
                            T* __restrict__ q_drop, unsigned seed, unsigned thr, float dscale, int h_chunk_rows) {
    const int H4 = H >> 2;
    const int rows0 = m_t[0], rows1 = m_t[1];
    const long total = (long)rows0 * H4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H4), j = (int)(i % H4) * 4;
        const int ob = perm[b];
        int64_t id0 = tok[(size_t)ob * L];
        if (id0 < 0 || id0 >= n_table) id0 = 0;
        int64_t id1 = tok[(size_t)ob * L + 1];
        if (id1 < 0 || id1 >= n_table) id1 = 0;
        const bool second = b < rows1;
        f32x4 g[4], g1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = Vec4<T>::load(xg + (size_t)id0 * ld_xg + k * H + j);
        if (second) {                       // requested before step 0's arithmetic: both gathers are in flight together
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g1[k] = Vec4<T>::load(xg + (size_t)id1 * ld_xg + k * H + j);
                g1[k] += Vec4<float>::load(hh_table + (size_t)id0 * 4 * H + k * H + j);
            }
        }
        int last = len[ob] - 1;
        if (last < 0) last += L;
        f32x4 cn, hn;
        auto emit = [&](int t) {            // what k_lstm_cell stores for step t
            if (hs) Vec4<T>::store(hs + (size_t)(offs[t] + b) * ld_hs + j, hn);
            if (last == t) {
                Vec4<float>::store(q32 + (size_t)ob * ld_q + j, hn);
                Vec4<T>::store(qT + (size_t)ob * ld_q + j, hn);
                if (q_drop) {
                    f32x4 qd;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        qd[e] = to_f32<T>(from_f32<T>(hn[e])) * drop_mult(seed, (unsigned)(ob * H + j + e), thr, dscale);
                    Vec4<T>::store(q_drop + (size_t)ob * ld_q + j, qd);
                }
            }
        };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = Act<T>::sigm_(g[0][e]), fg = Act<T>::sigm_(g[1][e]);
            const float gg = Act<T>::tanh_(g[2][e]), og = Act<T>::sigm_(g[3][e]);
            cn[e] = fg * 0.f + ig * gg;      // (as the one-step kernel: f * c_prev with c_prev = 0)
            hn[e] = og * Act<T>::tanh_(cn[e]);
        }
        emit(0);
        if (second) {
            // the one-step kernel reads h_0 back only through the recurrent product, and c_0 as stored fp32
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = Act<T>::sigm_(g1[0][e]), fg = Act<T>::sigm_(g1[1][e]);
                const float gg = Act<T>::tanh_(g1[2][e]), og = Act<T>::sigm_(g1[3][e]);
                cn[e] = fg * cn[e] + ig * gg;
                hn[e] = og * Act<T>::tanh_(cn[e]);
            }
            emit(1);
        }
        Vec4<float>::store(c + (size_t)b * H + j, cn);
        if (h_chunk_rows > 0)      // h as [H / 8][h_chunk_rows][8]: the first ring slot of vqa_lstm_resident
            Vec4<T>::store(h + ((size_t)(j >> 3) * h_chunk_rows + b) * 8 + (j & 7), hn);
        else
            Vec4<T>::store(h + (size_t)b * ld_h + j, hn);
    }
}

// caption vector: masked mean of the gathered rows, then L2 normalise; one
// block per sample; written into columns [col0, col0 + Ec) of out.
template <typename T>
__global__ void k_caption_pool(const int64_t* __restrict__ cap, const T* __restrict__ table,
                               T* __restrict__ out, int Tc, int Ec, int ld_table, int n_rows,
                               int ld_out, int col0) {
    __shared__ float red[16];
    extern __shared__ float acc[];          // Ec floats
    const int b = blockIdx.x;
    float cnt = 0.f;
    for (int t = 0; t < Tc; ++t) cnt += (cap[(size_t)b * Tc + t] != 0) ? 1.f : 0.f;
    const float inv = 1.0f / fmaxf(cnt, 1e-9f);
    float ss = 0.f;
    for (int e = threadIdx.x; e < Ec; e += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < Tc; ++t) {
            int64_t id = cap[(size_t)b * Tc + t];
            if (id != 0) {
                if (id < 0 || id >= n_rows) id = 0;
                s += to_f32<T>(table[(size_t)id * ld_table + e]);
            }
        }
        s *= inv;
        acc[e] = s;
        ss += s * s;
    }
    ss = block_sum(ss, red);
    const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    for (int e = threadIdx.x; e < Ec; e += blockDim.x)
        out[(size_t)b * ld_out + col0 + e] = from_f32<T>(acc[e] * rn);
}

// ---------------------------------------------------------------------------
// image side: inv_norm[row] = 1 / max(||V[row,:]||_2, 1e-12); one wave per row
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_inv_norm(const T* __restrict__ V, float* __restrict__ inv_norm,
                                                  long rows, int Dv, int ldV) {
    // each wave walks rows wave_id, wave_id + n_waves, ...; the loads of the next row are issued
    // before the current row is reduced (all loads unconditional: indices are clamped)
    const int lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = Dv >> 3;
    if (nv <= 256) {
        int voff[4];
        float live[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int v = lane + 64 * i; voff[i] = min(v, nv - 1) * 8; live[i] = v < nv ? 1.f : 0.f; }
        float cur[4][8], nxt[4][8];
        if (w0 < rows) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Vec8<T>::load(V + (size_t)w0 * ldV + voff[i], cur[i]);
        }
        for (long row = w0; row < rows; row += nw) {
            const long rn = row + nw < rows ? row + nw : rows - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) Vec8<T>::load(V + (size_t)rn * ldV + voff[i], nxt[i]);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += cur[i][e] * cur[i][e];
                ss += a * live[i];
            }
            ss = wave_sum(ss);
            if (lane == 0) inv_norm[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) cur[i][e] = nxt[i][e];
        }
    } else {
        for (long row = w0; row < rows; row += nw) {
            const T* p = V + (size_t)row * ldV;
            float ss = 0.f;
            for (int v = lane; v < nv; v += 64) {
                float x[8];
                Vec8<T>::load(p + v * 8, x);
#pragma unroll
                for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
            }
            ss = wave_sum(ss);
            if (lane == 0) inv_norm[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        }
    }
}

// bf16 rows of up to 2048 elements: raw 16-byte nontemporal loads (the norm pass is the first touch of the
// features in a step and nothing of a row is reused by it), four rows in flight per wave
__device__ __forceinline__ float sumsq_bf16x8(const u16x8& v) {
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float x = bf16_to_f32(v[e]); a = fmaf(x, x, a); }
    return a;
}

template <int NV>      // 16-byte vectors per lane and row = ceil(Dv / 512)
__global__ void __launch_bounds__(256) k_inv_norm_stream(const bf16_t* __restrict__ V, float* __restrict__ inv_norm,
                                                         long rows, int Dv, int ldV) {
    const int lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = Dv >> 3;
    int voff[NV];
    float live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int v = lane + 64 * i; voff[i] = min(v, nv - 1) * 8; live[i] = v < nv ? 1.f : 0.f; }
    constexpr int D = 4;                                   // rows in flight
    u16x8 buf[D][NV];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long r = min(w0 + d * nw, rows - 1);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            buf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(V + (size_t)r * ldV + voff[i]));
    }
    for (long row = w0; row < rows; row += D * nw) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const long r = row + d * nw;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) ss += sumsq_bf16x8(buf[d][i]) * live[i];
            const long rn = min(r + D * nw, rows - 1);     // refill this slot before the reduction waits
#pragma unroll
            for (int i = 0; i < NV; ++i)
                buf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(V + (size_t)rn * ldV + voff[i]));
            ss = wave_sum(ss);
            if (lane == 0 && r < rows) inv_norm[r] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        }
    }
}

// ---------------------------------------------------------------------------
// generic split-K reduce + epilogue
//   v = act(row_scale[m] * sum_s slab[s][m][n] + bias[n] + addend[m][n])
//   out_pre (fp32, optional) = v ; out_post (T or fp32, optional) = v * dropout
// ---------------------------------------------------------------------------
template <typename TP>
__global__ void k_epilogue(const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N,
                           const float* __restrict__ bias, const float* __restrict__ row_scale,
                           const float* __restrict__ addend, int ld_add, int act, int accurate,
                           float* __restrict__ out_pre, int ld_pre, TP* __restrict__ out_post, int ld_post,
                           unsigned seed, unsigned thr, float dscale, const float* __restrict__ mul = nullptr) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = slab_sum(slabs + i, nslab, slab_stride);
        if (row_scale) v *= row_scale[m];
        if (bias) v += bias[n];
        if (addend) v += addend[(size_t)m * ld_add + n];
        if (act == 1) v = accurate ? tanhf(v) : fast_tanh(v);
        else if (act == 2) v = accurate ? 1.0f / (1.0f + expf(-v)) : fast_sigmoid(v);
        if (out_pre) out_pre[(size_t)m * ld_pre + n] = v;
        if (mul) {                       // fusion head: out_post = dropout(tanh(v * mul[m][n])), mul stored (M, N)
            const float w = v * mul[i];
            v = accurate ? tanhf(w) : fast_tanh(w);
        }
        if (out_post) out_post[(size_t)m * ld_post + n] = from_f32<TP>(v * drop_mult(seed, (unsigned)i, thr, dscale));
    }
}

// y = x * dropout(idx); x, y of type T, logical shape (M, N)
template <typename T>
__global__ void k_dropout(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int M, int N,
                          unsigned seed, unsigned thr, float dscale) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        y[(size_t)m * ldy + n] = from_f32<T>(to_f32<T>(x[(size_t)m * ldx + n]) * drop_mult(seed, (unsigned)i, thr, dscale));
    }
}

// fusion forward (reference head): x = dropout(tanh(xv * xq))
template <typename T>
__global__ void k_fusion_fwd(const float* __restrict__ xv, const float* __restrict__ xq, T* __restrict__ x,
                             int ldx, int B, int F, unsigned seed, unsigned thr, float dscale) {
    const long total = (long)B * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i % F);
        const float t = Act<T>::tanh_(xv[i] * xq[i]);
        x[(size_t)b * ldx + f] = from_f32<T>(t * drop_mult(seed, (unsigned)i, thr, dscale));
    }
}

// fusion forward straight from the split-K partial sums of both branches:
//   xv = tanh(sum_s sv + bv), xq = tanh(sum_s sq + bq) (both stored, fp32), x = dropout(tanh(xv * xq))
template <typename T>
__global__ void k_fusion_fwd_slabs(const float* __restrict__ sv, int nv, size_t stride_v, const float* __restrict__ bv,
                                   const float* __restrict__ sq, int nq, size_t stride_q, const float* __restrict__ bq,
                                   float* __restrict__ xv, float* __restrict__ xq, T* __restrict__ x, int ldx, int B, int F,
                                   unsigned seed, unsigned thr, float dscale) {
    const long total = (long)B * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i % F);
        float a = slab_sum(sv + i, nv, stride_v), c = slab_sum(sq + i, nq, stride_q);
        a = Act<T>::tanh_(a + bv[f]);
        c = Act<T>::tanh_(c + bq[f]);
        xv[i] = a;
        xq[i] = c;
        x[(size_t)b * ldx + f] = from_f32<T>(Act<T>::tanh_(a * c) * drop_mult(seed, (unsigned)i, thr, dscale));
    }
}

// fusion backward: dx = sum_s slab (first F of ldslab columns) is d loss / d x
//   dxf  = dx * dropout * (1 - tanh(xv xq)^2)
//   dzv  = dxf * xq * (1 - xv^2)      (pre-activation grad of linear_v)
//   dzq  = dxf * xv * (1 - xq^2)      (pre-activation grad of linear_q)
template <typename T>
__global__ void k_fusion_bwd(const float* __restrict__ slabs, int nslab, size_t slab_stride, int ldslab,
                             const float* __restrict__ xv, const float* __restrict__ xq,
                             T* __restrict__ dzv, T* __restrict__ dzq, int ldz, int B, int F,
                             unsigned seed, unsigned thr, float dscale) {
    const long total = (long)B * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / F), f = (int)(i % F);
        float dx = 0.f;
        for (int s = 0; s < nslab; ++s) dx += slabs[s * slab_stride + (size_t)b * ldslab + f];
        const float a = xv[i], q = xq[i];
        const float t = Act<T>::tanh_(a * q);
        const float dxf = dx * drop_mult(seed, (unsigned)i, thr, dscale) * (1.0f - t * t);
        dzv[(size_t)b * ldz + f] = from_f32<T>(dxf * q * (1.0f - a * a));
        dzq[(size_t)b * ldz + f] = from_f32<T>(dxf * a * (1.0f - q * q));
    }
}

// backward through y = dropout(tanh(z)):  dz = (sum_s slab) * dropout * (1 - t^2), t = tanh(z) stored fp32
template <typename T>
__global__ void k_tanh_drop_bwd(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                const float* __restrict__ t_pre, int ld_t, T* __restrict__ dz, int ldz,
                                int M, int N, unsigned seed, unsigned thr, float dscale) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float d = 0.f;
        for (int s = 0; s < nslab; ++s) d += slabs[s * slab_stride + i];
        const float t = t_pre[(size_t)m * ld_t + n];
        dz[(size_t)m * ldz + n] = from_f32<T>(d * drop_mult(seed, (unsigned)i, thr, dscale) * (1.0f - t * t));
    }
}

// out[n] = sum_m X[m, n]   (bias gradients); fixed summation order.
// 256 threads = 8 row groups x 32 columns; each group strides the rows, LDS combines the groups.
template <typename T>
__global__ void __launch_bounds__(256) k_colsum(const T* __restrict__ X, int ld, float* __restrict__ out, int M, int N) {
    __shared__ float part[8][33];
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = rg;
        for (; m + 24 < M; m += 32) {
            s0 += to_f32<T>(X[(size_t)m * ld + n]);
            s1 += to_f32<T>(X[(size_t)(m + 8) * ld + n]);
            s2 += to_f32<T>(X[(size_t)(m + 16) * ld + n]);
            s3 += to_f32<T>(X[(size_t)(m + 24) * ld + n]);
        }
        for (; m < M; m += 8) s0 += to_f32<T>(X[(size_t)m * ld + n]);
    }
    part[rg][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += part[g][cx];
        out[n] = s;
    }
}

// several split-K reductions in one launch: out_i[e] = sum_s slabs_i[s * stride_i + e], e < count_i (count_i % 4 == 0
// is not required; 16-byte accesses are used when the slab and the output allow it)
struct SlabSumBatch { VqaSlabSumDesc d[VQA_SLABSUM_MAX]; int first_block[VQA_SLABSUM_MAX + 1]; int n; };

__global__ void __launch_bounds__(256) k_slab_sum_multi(SlabSumBatch batch) {
    int pi = 0;
    while (pi + 1 < batch.n && (int)blockIdx.x >= batch.first_block[pi + 1]) ++pi;
    const VqaSlabSumDesc d = batch.d[pi];
    const long i4 = ((long)((int)blockIdx.x - batch.first_block[pi]) * 256 + threadIdx.x) * 4;
    if (i4 >= d.count) return;
    const bool vec = i4 + 4 <= d.count && (d.slab_stride & 3) == 0 && ((((uintptr_t)d.slabs) | ((uintptr_t)d.out)) & 15) == 0;
    if (vec) {
        f32x4 a = *reinterpret_cast<const f32x4*>(d.slabs + i4);
        for (int s = 1; s < d.nslab; ++s) a += *reinterpret_cast<const f32x4*>(d.slabs + s * d.slab_stride + i4);
        *reinterpret_cast<f32x4*>(d.out + i4) = a;
    } else {
        for (long i = i4; i < d.count && i < i4 + 4; ++i) {
            float a = 0.f;
            for (int s = 0; s < d.nslab; ++s) a += d.slabs[s * d.slab_stride + i];
            d.out[i] = a;
        }
    }
}

// out[n][m] = sum_s slabs[s][m][n]: split-K reduce of a product that was computed as its transpose.
// 32 x 32 tiles through LDS: rows of the slabs are read and rows of `out` are written as 128-byte segments.
__global__ void __launch_bounds__(256) k_slab_sum_transpose(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                                            int M, int N, float* __restrict__ out, int ld_out) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 8 * r, n = n0 + tx;
        float a = 0.f;
        if (m < M && n < N)
            for (int s = 0; s < nslab; ++s) a += slabs[s * slab_stride + (size_t)m * N + n];
        t[ty + 8 * r][tx] = a;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + 8 * r, m = m0 + tx;
        if (n < N && m < M) out[(size_t)n * ld_out + m] = t[tx][ty + 8 * r];
    }
}

// The same on 64 x 64 tiles with 16-byte accesses on both sides (M, N, ld_out multiples of 4, 16-byte aligned bases).
__global__ void __launch_bounds__(256) k_slab_sum_transpose4(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                                             int M, int N, float* __restrict__ out, int ld_out) {
    __shared__ float t[64][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ml = ty + 16 * r, m = m0 + ml, n = n0 + tx * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (m < M && n < N) {
            a = *reinterpret_cast<const f32x4*>(slabs + (size_t)m * N + n);
            for (int s = 1; s < nslab; ++s) a += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + (size_t)m * N + n);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) t[ml][tx * 4 + e] = a[e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nl = ty + 16 * r, n = n0 + nl, m = m0 + tx * 4;
        if (n < N && m < M) {
            const f32x4 v = {t[tx * 4][nl], t[tx * 4 + 1][nl], t[tx * 4 + 2][nl], t[tx * 4 + 3][nl]};
            *reinterpret_cast<f32x4*>(out + (size_t)n * ld_out + m) = v;
        }
    }
}

// several column sums in one launch (colsum.h: the work of one block is shared with the grouped GEMM launch)
__global__ void __launch_bounds__(256) k_colsum_multi(ColsumBatch batch) {
    __shared__ float part[8][33];
    colsum_block(batch, (int)blockIdx.x, part);
}

// cast fp32 -> T with row padding: dst[m, 0:N] = src[m, 0:N], dst[m, N:ld_dst] = 0
template <typename T>
__global__ void k_cast_pad(const float* __restrict__ src, int ld_src, T* __restrict__ dst, int ld_dst, int M, int N) {
    const long total = (long)M * ld_dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ld_dst), n = (int)(i % ld_dst);
        dst[i] = from_f32<T>(n < N ? src[(size_t)m * ld_src + n] : 0.f);
    }
}

// ---------------------------------------------------------------------------
// softmax cross-entropy forward + backward, one block per row
//   loss_row[b] = logsumexp(z_b) - z_b[y_b];  dz = (softmax(z_b) - onehot(y_b)) * scale
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_softmax_ce(const float* __restrict__ logits, int ld,
                                                    const int64_t* __restrict__ target,
                                                    float* __restrict__ loss_row, T* __restrict__ dlogits,
                                                    int ld_d, int C, float scale, int* __restrict__ status) {
    __shared__ float red[8];
    const int b = blockIdx.x;
    const float* z = logits + (size_t)b * ld;
    // rows of up to 16 * 256 classes are read once and kept in registers; longer rows are re-read
    constexpr int KEEP = 16;
    const bool keep = C <= KEEP * 256;
    float zr[KEEP];
    float mx = -INFINITY;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int c = threadIdx.x + 256 * k;
            zr[k] = c < C ? z[c] : -INFINITY;
            mx = fmaxf(mx, zr[k]);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, z[c]);
    }
    mx = block_max(mx, red);
    float se = 0.f;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) { zr[k] = expf(zr[k] - mx); se += zr[k]; }     // exp(-inf) = 0 past the row
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) se += expf(z[c] - mx);
    }
    se = block_sum(se, red);
    const int64_t y = target[b];
    // a target outside [0, C) (torch.nn.CrossEntropyLoss raises for it before anything is updated): the row's loss is
    // NaN, its gradient zero (a NaN would travel to every rank of a data-parallel group through the all-reduce), and
    // the status word makes the Adam kernel leave the weights alone and tells the host what happened
    const bool bad = y < 0 || y >= C;
    if (threadIdx.x == 0) {
        loss_row[b] = bad ? __builtin_nanf("") : (logf(se) - (z[y] - mx));
        if (bad && status) status[VQA_STATUS_BAD_TARGET] = 1;
    }
    if (dlogits) {
        const float inv = 1.0f / se;
        if (keep) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int c = threadIdx.x + 256 * k;
                if (c < ld_d) dlogits[(size_t)b * ld_d + c] = from_f32<T>((c < C && !bad) ? (zr[k] * inv - (c == y ? 1.f : 0.f)) * scale : 0.f);
            }
        } else {
            for (int c = threadIdx.x; c < ld_d; c += blockDim.x) {
                float g = 0.f;
                if (c < C && !bad) g = (expf(z[c] - mx) * inv - (c == y ? 1.f : 0.f)) * scale;
                dlogits[(size_t)b * ld_d + c] = from_f32<T>(g);
            }
        }
    }
}

// out[0] = sum(x[0:n]) * scale, single block, fixed order
__global__ void __launch_bounds__(256) k_sum_scale(const float* __restrict__ x, float* __restrict__ out, int n, float scale) {
    __shared__ float red[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// ---------------------------------------------------------------------------
// argmax (first maximum, NaN counts as maximum, as numpy) + per-type counters
// one wave per row; counters[0:nt] = count, counters[nt:2nt] = right
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool argmax_better(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ia < ib);
    return a > b || (a == b && ia < ib);
}

__global__ void __launch_bounds__(256) k_argmax_accuracy(const float* __restrict__ logits, int ld, int B, int C,
                                                         const int64_t* __restrict__ answer,
                                                         const int* __restrict__ type_id, int n_types,
                                                         int64_t* __restrict__ pred,
                                                         unsigned long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* z = logits + (size_t)b * ld;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = z[c];
        if (bi == 0x7fffffff || argmax_better(v, c, bv, bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || argmax_better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        pred[b] = bi;
        if (counters && answer && type_id) {
            const int t = type_id[b];
            if (t >= 0 && t < n_types) {
                atomicAdd(&counters[t], 1ull);
                if (answer[b] == (int64_t)bi) atomicAdd(&counters[n_types + t], 1ull);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// fused multi-tensor Adam on flat buffers (torch.optim.Adam defaults, no decay)
//   g <- grad * grad_scale; m, v updated; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
//   optional shadow[i] = bf16(p[i])
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              bf16_t* __restrict__ shadow, long n, float lr, float b1, float b2,
                                              float eps, float inv_bc1, float inv_sqrt_bc2, float grad_scale,
                                              int* __restrict__ status, int step) {
    // the loss kernel met a target the reference's CrossEntropyLoss raises for (train.py:201, before optimizer.step()):
    // nothing is updated, and the first step skipped this way is recorded for the host
    if (status && status[VQA_STATUS_BAD_TARGET] != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && status[VQA_STATUS_SKIPPED_STEP] == 0) status[VQA_STATUS_SKIPPED_STEP] = step;
        return;
    }
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        // gradients and moments are touched once per step: streamed past the caches; the parameters and
        // their bf16 copies are what the next forward reads, so they go through them
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
        u16x4 sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * grad_scale;
            mm[k] = b1 * mm[k] + (1.0f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.0f - b2) * gk * gk;
            const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
            pp[k] -= lr * inv_bc1 * (mm[k] / denom);
            sh[k] = f32_to_bf16(pp[k]);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
        if (shadow) reinterpret_cast<u16x4*>(shadow)[i] = sh;
    }
    // tail (n % 4 elements)
    const long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && i < n) {
        const float gk = g[i] * grad_scale;
        const float mk = b1 * m[i] + (1.0f - b1) * gk;
        const float vk = b2 * v[i] + (1.0f - b2) * gk * gk;
        const float pk = p[i] - lr * inv_bc1 * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
        m[i] = mk; v[i] = vk; p[i] = pk;
        if (shadow) shadow[i] = f32_to_bf16(pk);
    }
}

// ---------------------------------------------------------------------------
// backward through one LSTM time step on packed sequences (rows j < m_t[t], sorted order)
//   dh   = [row j also runs at t+1] sum_s slab[s][j]  (= dG_{t+1} W_hh)  + [t is row j's last step] dq[perm[j]]
//          + dh_ext[offs[t] + j]                       (gradient arriving from a stacked layer, optional)
//   dc   = [runs at t+1] dc_state[j] + dh o (1 - tanh(c_t)^2)
//   dG   = (dc g i(1-i), dc c_{t-1} f(1-f), dc i (1-g^2), dh tanh(c_t) o(1-o))   -> packed row offs[t] + j
//   dc_state[j] = dc f;  hprev[offs[t] + j] = h_{t-1}[j]  (0 at t = 0): the operand of dW_hh = dG^T hprev
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_lstm_cell_bwd(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                const float* __restrict__ dq, int ld_dq, const float* __restrict__ dh_ext,
                                const T* __restrict__ gates, const float* __restrict__ cs, const T* __restrict__ hs,
                                float* __restrict__ dc_state, T* __restrict__ dG, T* __restrict__ dgs,
                                T* __restrict__ hprev, const int* __restrict__ len, const int* __restrict__ perm,
                                const int* __restrict__ m_t, const int* __restrict__ offs, int H, int t, int L) {
    const int H4 = H >> 2;
    const int rows = m_t[t];
    const int rows_next = t + 1 < L ? m_t[t + 1] : 0;
    const long r0 = offs[t], rp = t > 0 ? offs[t - 1] : 0;
    const long total = (long)rows * H4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H4), j = (int)(i % H4) * 4;
        const bool runs_next = b < rows_next;
        f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f}, dc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (runs_next) {
            for (int s = 0; s < nslab; ++s) dh += Vec4<float>::load(slabs + s * slab_stride + (size_t)b * H + j);
            dc = Vec4<float>::load(dc_state + (size_t)b * H + j);
        }
        const int ob = perm[b];
        int last = len[ob] - 1;
        if (last < 0) last += L;
        if (last == t) dh += Vec4<float>::load(dq + (size_t)ob * ld_dq + j);
        if (dh_ext) dh += Vec4<float>::load(dh_ext + (size_t)(r0 + b) * H + j);
        const T* gr = gates + (size_t)(r0 + b) * 4 * H + j;
        const f32x4 ig = Vec4<T>::load(gr), fg = Vec4<T>::load(gr + H), gg = Vec4<T>::load(gr + 2 * H), og = Vec4<T>::load(gr + 3 * H);
        const f32x4 c = Vec4<float>::load(cs + (size_t)(r0 + b) * H + j);
        f32x4 cprev = f32x4{0.f, 0.f, 0.f, 0.f}, hp = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            cprev = Vec4<float>::load(cs + (size_t)(rp + b) * H + j);
            hp = Vec4<T>::load(hs + (size_t)(rp + b) * H + j);
        }
        f32x4 di, df, dg, dout, dcp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float tc = Act<T>::tanh_(c[e]);
            const float dct = dc[e] + dh[e] * og[e] * (1.0f - tc * tc);
            dout[e] = dh[e] * tc * og[e] * (1.0f - og[e]);
            di[e] = dct * gg[e] * ig[e] * (1.0f - ig[e]);
            df[e] = dct * cprev[e] * fg[e] * (1.0f - fg[e]);
            dg[e] = dct * ig[e] * (1.0f - gg[e] * gg[e]);
            dcp[e] = dct * fg[e];
        }
        Vec4<float>::store(dc_state + (size_t)b * H + j, dcp);
        T* go = dG + (size_t)(r0 + b) * 4 * H + j;
        Vec4<T>::store(go, di); Vec4<T>::store(go + H, df); Vec4<T>::store(go + 2 * H, dg); Vec4<T>::store(go + 3 * H, dout);
        T* gs = dgs + (size_t)b * 4 * H + j;      // the same row in sorted (unpacked) order: operand of the next recurrent product
        Vec4<T>::store(gs, di); Vec4<T>::store(gs + H, df); Vec4<T>::store(gs + 2 * H, dg); Vec4<T>::store(gs + 3 * H, dout);
        Vec4<T>::store(hprev + (size_t)(r0 + b) * H + j, hp);
    }
}

// ---------------------------------------------------------------------------
// embedding gradient without float atomics: one wave per table row v scans the packed token list
// and adds, in ascending packed-row order, the rows of dX = sum_s slab[s] whose token is v.
// x_act (optional): the tanh-activated embeddings the layer consumed; dX is then multiplied by
// 1 - x^2.  The pad row 0 gets a zero gradient (padding_idx).  E <= 1024.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_embedding_grad(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                 int ld_dx, const int* __restrict__ ptok, const int* __restrict__ total_rows,
                                 const T* __restrict__ x_act, int ld_x, float* __restrict__ dE, int V, int E) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int total = *total_rows;
    if (v != 0) {
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            unsigned long long mask = __ballot(p < total && ptok[p] == v);
            while (mask) {
                const int row = base + __ffsll((long long)mask) - 1;
                mask &= mask - 1;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = lane + 64 * i;
                    if (e < E) {
                        float d = 0.f;
                        for (int s = 0; s < nslab; ++s) d += slabs[s * slab_stride + (size_t)row * ld_dx + e];
                        if (x_act) { const float x = to_f32<T>(x_act[(size_t)row * ld_x + e]); d *= 1.0f - x * x; }
                        acc[i] += d;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i;
        if (e < E) dE[(size_t)v * E + e] = acc[i];
    }
}

// column sums over the first *rows rows (a device-side count); otherwise as k_colsum
template <typename T>
__global__ void __launch_bounds__(256) k_colsum_rows(const T* __restrict__ X, int ld, float* __restrict__ out,
                                                     const int* __restrict__ rows, int N) {
    __shared__ float part[8][33];
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cx;
    const int M = *rows;
    float s0 = 0.f;
    if (n < N)
        for (int m = rg; m < M; m += 8) s0 += to_f32<T>(X[(size_t)m * ld + n]);
    part[rg][cx] = s0;
    __syncthreads();
    if (rg == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += part[g][cx];
        out[n] = s;
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

#define DISPATCH_T(dtype, CALL)                                      \
    if ((dtype) == VQA_F32) { using T = float; CALL; }              \
    else if ((dtype) == VQA_BF16) { using T = bf16_t; CALL; }       \
    else return vqa_set_error(-22, "dtype must be VQA_F32 or VQA_BF16");

extern "C" {

int vqa_process_lengths(const int64_t* tokens, int32_t* lengths, int B, int L, hipStream_t stream) {
    VQA_REQUIRE(tokens && lengths && B >= 0 && L > 0, "vqa_process_lengths: bad arguments");
    if (B == 0) return 0;
    k_process_lengths<<<(B + 255) / 256, 256, 0, stream>>>(tokens, lengths, B, L);
    return vqa_check_launch("vqa_process_lengths");
}

int vqa_embedding_gather(const int64_t* tokens, const void* table, void* out, int B, int L, int width,
                         int ld_table, int ld_out, int n_rows, int time_major, int dtype, hipStream_t stream) {
    VQA_REQUIRE(tokens && table && out && B >= 0 && L > 0 && n_rows > 0, "vqa_embedding_gather: bad arguments");
    VQA_REQUIRE(width > 0 && width % 8 == 0 && ld_table % 8 == 0 && ld_out % 8 == 0 && width <= ld_table && width <= ld_out,
                "vqa_embedding_gather: width and leading dimensions must be multiples of 8");
    if (B == 0) return 0;
    const long total = (long)B * L * (width / 8);
    DISPATCH_T(dtype, (k_embedding_gather<T><<<grid_for(total, 256), 256, 0, stream>>>(
        tokens, (const T*)table, (T*)out, B, L, width, ld_table, ld_out, n_rows, time_major)));
    return vqa_check_launch("vqa_embedding_gather");
}

int vqa_sequence_pack(const int64_t* tokens, int B, int L, int32_t* lengths, int32_t* perm, int32_t* active,
                      int32_t* offsets, long n_table_rows, int32_t* status, hipStream_t stream) {
    VQA_REQUIRE(tokens && lengths && perm && active && offsets && B >= 0 && L > 0, "vqa_sequence_pack: bad arguments");
    const size_t lds = (size_t)(B + 18 * (L + 1)) * sizeof(int);
    VQA_REQUIRE(lds <= 64 * 1024, "vqa_sequence_pack: B + 18 (L + 1) must be <= 16384");
    if (B == 0) return 0;
    k_sequence_pack<<<1, 1024, lds, stream>>>(tokens, B, L, lengths, perm, active, offsets, n_table_rows, status);
    return vqa_check_launch("vqa_sequence_pack");
}

int vqa_embedding_gather_packed(const int64_t* tokens, const void* table, void* out, int B, int L, int width,
                                int ld_table, int ld_out, int n_rows, const int32_t* perm, const int32_t* offsets,
                                int act, int32_t* packed_tokens, int dtype, hipStream_t stream) {
    VQA_REQUIRE(tokens && table && out && perm && offsets && B >= 0 && L > 0 && n_rows > 0,
                "vqa_embedding_gather_packed: bad arguments");
    VQA_REQUIRE(width > 0 && width % 8 == 0 && ld_table % 8 == 0 && ld_out % 8 == 0 && width <= ld_table && width <= ld_out,
                "vqa_embedding_gather_packed: width and leading dimensions must be multiples of 8");
    VQA_REQUIRE(act == 0 || act == 1, "vqa_embedding_gather_packed: act must be 0 (none) or 1 (tanh)");
    if (B == 0) return 0;
    const long total = (long)B * L * (width / 8);
    DISPATCH_T(dtype, (k_embedding_gather_packed<T><<<grid_for(total, 256), 256, 0, stream>>>(
        tokens, (const T*)table, (T*)out, B, L, width, ld_table, ld_out, n_rows, perm, offsets, act, packed_tokens)));
    return vqa_check_launch("vqa_embedding_gather_packed");
}

int vqa_lstm_cell_packed(const float* slabs, int nslab, size_t slab_stride, const void* xg, int ld_xg, float* c,
                         void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                         int B, int H, int t, int L, const int32_t* perm, const int32_t* active,
                         const int32_t* offsets, void* hs, int ld_hs, void* gates_out, float* c_out, void* q_drop,
                         unsigned drop_seed, float drop_p, int dtype, hipStream_t stream) {
    VQA_REQUIRE(xg && c && h && lengths && q_f32 && q_t && perm && active && offsets && (nslab == 0 || slabs),
                "vqa_lstm_cell_packed: null pointer");
    VQA_REQUIRE(B >= 0 && H > 0 && t >= 0 && t < L && ld_xg >= 4 * H && ld_h >= H && ld_q >= H, "vqa_lstm_cell_packed: bad sizes");
    VQA_REQUIRE(H % 4 == 0 && ld_xg % 4 == 0 && ld_h % 4 == 0 && ld_q % 4 == 0 && (slab_stride % 4 == 0 || nslab <= 1),
                "vqa_lstm_cell_packed: H and leading dimensions must be multiples of 4");
    VQA_REQUIRE(!hs || (ld_hs >= H && ld_hs % 4 == 0), "vqa_lstm_cell_packed: bad ld_hs");
    VQA_REQUIRE((gates_out == nullptr) == (c_out == nullptr), "vqa_lstm_cell_packed: gates_out and c_out go together");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_lstm_cell_packed: drop_p must be in [0, 1)");
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_lstm_cell<T><<<grid_for((long)B * (H / 4), 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, (const T*)xg, ld_xg, c, (T*)h, ld_h, lengths, q_f32, (T*)q_t, ld_q, B, H, t, L,
        perm, active, offsets, (T*)hs, ld_hs, (T*)gates_out, c_out, (T*)q_drop, drop_seed, thr, dsc)));
    return vqa_check_launch("vqa_lstm_cell_packed");
}

int vqa_lstm_cell_packed_table(const float* slabs, int nslab, size_t slab_stride, const void* xg_table, int ld_xg,
                               int n_table_rows, const int64_t* tokens, float* c, void* h, int ld_h,
                               const int32_t* lengths, float* q_f32, void* q_t, int ld_q, int B, int H, int t, int L,
                               const int32_t* perm, const int32_t* active, const int32_t* offsets, void* hs, int ld_hs,
                               void* q_drop, unsigned drop_seed, float drop_p, const float* hh_table, int dtype,
                               hipStream_t stream) {
    VQA_REQUIRE(!hh_table || (t >= 1 && nslab == 0), "vqa_lstm_cell_packed_table: hh_table replaces the recurrent product of a step t >= 1");
    VQA_REQUIRE(xg_table && tokens && n_table_rows > 0 && c && h && lengths && q_f32 && q_t && perm && active && offsets &&
                (nslab == 0 || slabs), "vqa_lstm_cell_packed_table: null pointer");
    VQA_REQUIRE(B >= 0 && H > 0 && t >= 0 && t < L && ld_xg >= 4 * H && ld_h >= H && ld_q >= H, "vqa_lstm_cell_packed_table: bad sizes");
    VQA_REQUIRE(H % 4 == 0 && ld_xg % 4 == 0 && ld_h % 4 == 0 && ld_q % 4 == 0 && (slab_stride % 4 == 0 || nslab <= 1),
                "vqa_lstm_cell_packed_table: H and leading dimensions must be multiples of 4");
    VQA_REQUIRE(!hs || (ld_hs >= H && ld_hs % 4 == 0), "vqa_lstm_cell_packed_table: bad ld_hs");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_lstm_cell_packed_table: drop_p must be in [0, 1)");
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_lstm_cell<T><<<grid_for((long)B * (H / 4), 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, (const T*)xg_table, ld_xg, c, (T*)h, ld_h, lengths, q_f32, (T*)q_t, ld_q, B, H, t, L,
        perm, active, offsets, (T*)hs, ld_hs, (T*)nullptr, nullptr, (T*)q_drop, drop_seed, thr, dsc, tokens, n_table_rows,
        hh_table)));
    return vqa_check_launch("vqa_lstm_cell_packed_table");
}

int vqa_lstm_cell01_table(const void* xg_table, int ld_xg, const float* hh_table, int n_table_rows, const int64_t* tokens,
                          float* c, void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                          int B, int H, int L, const int32_t* perm, const int32_t* active, const int32_t* offsets,
                          void* hs, int ld_hs, void* q_drop, unsigned drop_seed, float drop_p, int h_chunk_rows, int dtype,
                          hipStream_t stream) {
    VQA_REQUIRE(xg_table && hh_table && tokens && n_table_rows > 0 && c && h && lengths && q_f32 && q_t && perm && active &&
                offsets, "vqa_lstm_cell01_table: null pointer");
    VQA_REQUIRE(h_chunk_rows == 0 || (h_chunk_rows >= B && H % 8 == 0), "vqa_lstm_cell01_table: h_chunk_rows is 0 or at least B, with H a multiple of 8");
    VQA_REQUIRE(B >= 0 && H > 0 && L >= 2 && ld_xg >= 4 * H && ld_h >= H && ld_q >= H, "vqa_lstm_cell01_table: bad sizes");
    VQA_REQUIRE(H % 4 == 0 && ld_xg % 4 == 0 && ld_h % 4 == 0 && ld_q % 4 == 0,
                "vqa_lstm_cell01_table: H and leading dimensions must be multiples of 4");
    VQA_REQUIRE(!hs || (ld_hs >= H && ld_hs % 4 == 0), "vqa_lstm_cell01_table: bad ld_hs");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_lstm_cell01_table: drop_p must be in [0, 1)");
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_lstm_cell01_table<T><<<grid_for((long)B * (H / 4), 256), 256, 0, stream>>>(
        (const T*)xg_table, ld_xg, hh_table, n_table_rows, tokens, c, (T*)h, ld_h, lengths, q_f32, (T*)q_t, ld_q, H, L,
        perm, active, offsets, (T*)hs, ld_hs, (T*)q_drop, drop_seed, thr, dsc, h_chunk_rows)));
    return vqa_check_launch("vqa_lstm_cell01_table");
}

int vqa_lstm_cell(const float* slabs, int nslab, size_t slab_stride, const void* xg_t, int ld_xg, float* c,
                  void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                  int B, int H, int t, int L, int dtype, hipStream_t stream) {
    VQA_REQUIRE(xg_t && c && h && lengths && q_f32 && q_t && (nslab == 0 || slabs), "vqa_lstm_cell: null pointer");
    VQA_REQUIRE(B >= 0 && H > 0 && t >= 0 && t < L && ld_xg >= 4 * H && ld_h >= H && ld_q >= H, "vqa_lstm_cell: bad sizes");
    VQA_REQUIRE(H % 4 == 0 && ld_xg % 4 == 0 && ld_h % 4 == 0 && ld_q % 4 == 0 && (slab_stride % 4 == 0 || nslab <= 1),
                "vqa_lstm_cell: H and leading dimensions must be multiples of 4");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_lstm_cell<T><<<grid_for((long)B * (H / 4), 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, (const T*)xg_t, ld_xg, c, (T*)h, ld_h, lengths, q_f32, (T*)q_t, ld_q, B, H, t, L,
        nullptr, nullptr, nullptr, (T*)nullptr, 0, (T*)nullptr, nullptr, (T*)nullptr, 0u, 0u, 1.0f)));
    return vqa_check_launch("vqa_lstm_cell");
}

int vqa_caption_pool(const int64_t* cap_tokens, const void* table, void* out, int B, int Tc, int Ec,
                     int ld_table, int n_rows, int ld_out, int col0, int dtype, hipStream_t stream) {
    VQA_REQUIRE(cap_tokens && table && out && B >= 0 && Tc > 0 && Ec > 0 && n_rows > 0, "vqa_caption_pool: bad arguments");
    VQA_REQUIRE(col0 >= 0 && col0 + Ec <= ld_out && Ec <= ld_table && Ec <= 12288, "vqa_caption_pool: bad sizes");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_caption_pool<T><<<B, 128, Ec * sizeof(float), stream>>>(
        cap_tokens, (const T*)table, (T*)out, Tc, Ec, ld_table, n_rows, ld_out, col0)));
    return vqa_check_launch("vqa_caption_pool");
}

int vqa_inv_norm(const void* V, float* inv_norm, long rows, int Dv, int ldV, int dtype, hipStream_t stream) {
    VQA_REQUIRE(V && inv_norm && rows >= 0 && Dv > 0, "vqa_inv_norm: bad arguments");
    VQA_REQUIRE(Dv % 8 == 0 && ldV % 8 == 0 && ldV >= Dv, "vqa_inv_norm: Dv and ldV must be multiples of 8");
    if (rows == 0) return 0;
    const unsigned grid = (unsigned)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    if (dtype == VQA_BF16 && Dv <= 2048) {
        const int nvl = (Dv / 8 + 63) / 64;
        if (nvl == 1) k_inv_norm_stream<1><<<grid, 256, 0, stream>>>((const bf16_t*)V, inv_norm, rows, Dv, ldV);
        else if (nvl == 2) k_inv_norm_stream<2><<<grid, 256, 0, stream>>>((const bf16_t*)V, inv_norm, rows, Dv, ldV);
        else if (nvl == 3) k_inv_norm_stream<3><<<grid, 256, 0, stream>>>((const bf16_t*)V, inv_norm, rows, Dv, ldV);
        else k_inv_norm_stream<4><<<grid, 256, 0, stream>>>((const bf16_t*)V, inv_norm, rows, Dv, ldV);
        return vqa_check_launch("vqa_inv_norm");
    }
    DISPATCH_T(dtype, (k_inv_norm<T><<<grid, 256, 0, stream>>>((const T*)V, inv_norm, rows, Dv, ldV)));
    return vqa_check_launch("vqa_inv_norm");
}

int vqa_epilogue(const float* slabs, int nslab, size_t slab_stride, int M, int N, const float* bias,
                 const float* row_scale, const float* addend, int ld_add, int act, float* out_pre, int ld_pre,
                 void* out_post, int ld_post, int post_dtype, unsigned drop_seed, float drop_p, hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && M >= 0 && N > 0, "vqa_epilogue: bad arguments");
    VQA_REQUIRE(out_pre || out_post, "vqa_epilogue: no output");
    VQA_REQUIRE(act >= 0 && act <= 2, "vqa_epilogue: act must be 0 (none), 1 (tanh) or 2 (sigmoid)");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_epilogue: drop_p must be in [0, 1)");
    if (M == 0) return 0;
    const unsigned thr = drop_threshold(drop_p);
    const float ds = 1.0f / (1.0f - drop_p);
    const int grid = grid_for((long)M * N, 256);
    if (post_dtype == VQA_F32)
        k_epilogue<float><<<grid, 256, 0, stream>>>(slabs, nslab, slab_stride, M, N, bias, row_scale, addend, ld_add,
                                                    act, 1, out_pre, ld_pre, (float*)out_post, ld_post, drop_seed, thr, ds);
    else if (post_dtype == VQA_BF16)
        k_epilogue<bf16_t><<<grid, 256, 0, stream>>>(slabs, nslab, slab_stride, M, N, bias, row_scale, addend, ld_add,
                                                     act, 0, out_pre, ld_pre, (bf16_t*)out_post, ld_post, drop_seed, thr, ds);
    else return vqa_set_error(-22, "vqa_epilogue: post_dtype");
    return vqa_check_launch("vqa_epilogue");
}

int vqa_dropout(const void* x, int ldx, void* y, int ldy, int M, int N, unsigned seed, float p, int dtype,
                hipStream_t stream) {
    VQA_REQUIRE(x && y && M >= 0 && N > 0 && p >= 0.f && p < 1.f, "vqa_dropout: bad arguments");
    if (M == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_dropout<T><<<grid_for((long)M * N, 256), 256, 0, stream>>>((const T*)x, ldx, (T*)y, ldy, M, N, seed, thr, ds)));
    return vqa_check_launch("vqa_dropout");
}

int vqa_fusion_fwd(const float* xv, const float* xq, void* x, int ldx, int B, int F, unsigned seed, float p,
                   int dtype, hipStream_t stream) {
    VQA_REQUIRE(xv && xq && x && B >= 0 && F > 0 && ldx >= F && p >= 0.f && p < 1.f, "vqa_fusion_fwd: bad arguments");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_fusion_fwd<T><<<grid_for((long)B * F, 256), 256, 0, stream>>>(xv, xq, (T*)x, ldx, B, F, seed, thr, ds)));
    return vqa_check_launch("vqa_fusion_fwd");
}

int vqa_epilogue_fusion(const float* slabs, int nslab, size_t slab_stride, int B, int F, const float* bias,
                        const float* xv, float* xq, void* x, int ldx, unsigned seed, float p, int dtype,
                        hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && xv && xq && x && B >= 0 && F > 0 && ldx >= F, "vqa_epilogue_fusion: bad arguments");
    VQA_REQUIRE(p >= 0.f && p < 1.f, "vqa_epilogue_fusion: p must be in [0, 1)");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    const int grid = grid_for((long)B * F, 256);
    if (dtype == VQA_F32)
        k_epilogue<float><<<grid, 256, 0, stream>>>(slabs, nslab, slab_stride, B, F, bias, nullptr, nullptr, 0, 1, 1, xq, F,
                                                    (float*)x, ldx, seed, thr, ds, xv);
    else if (dtype == VQA_BF16)
        k_epilogue<bf16_t><<<grid, 256, 0, stream>>>(slabs, nslab, slab_stride, B, F, bias, nullptr, nullptr, 0, 1, 0, xq, F,
                                                     (bf16_t*)x, ldx, seed, thr, ds, xv);
    else return vqa_set_error(-22, "vqa_epilogue_fusion: dtype");
    return vqa_check_launch("vqa_epilogue_fusion");
}

int vqa_fusion_fwd_slabs(const float* slabs_v, int nslab_v, size_t stride_v, const float* bias_v,
                         const float* slabs_q, int nslab_q, size_t stride_q, const float* bias_q,
                         float* xv, float* xq, void* x, int ldx, int B, int F, unsigned seed, float p, int dtype,
                         hipStream_t stream) {
    VQA_REQUIRE(slabs_v && slabs_q && nslab_v >= 1 && nslab_q >= 1 && bias_v && bias_q && xv && xq && x && B >= 0 && F > 0 &&
                ldx >= F, "vqa_fusion_fwd_slabs: bad arguments");
    VQA_REQUIRE(p >= 0.f && p < 1.f, "vqa_fusion_fwd_slabs: p must be in [0, 1)");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_fusion_fwd_slabs<T><<<grid_for((long)B * F, 256), 256, 0, stream>>>(
        slabs_v, nslab_v, stride_v, bias_v, slabs_q, nslab_q, stride_q, bias_q, xv, xq, (T*)x, ldx, B, F, seed, thr, ds)));
    return vqa_check_launch("vqa_fusion_fwd_slabs");
}

int vqa_fusion_bwd(const float* slabs, int nslab, size_t slab_stride, int ldslab, const float* xv, const float* xq,
                   void* dzv, void* dzq, int ldz, int B, int F, unsigned seed, float p, int dtype, hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && xv && xq && dzv && dzq && B >= 0 && F > 0 && ldz >= F && ldslab >= F,
                "vqa_fusion_bwd: bad arguments");
    VQA_REQUIRE(p >= 0.f && p < 1.f, "vqa_fusion_bwd: p must be in [0, 1)");
    if (B == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_fusion_bwd<T><<<grid_for((long)B * F, 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, ldslab, xv, xq, (T*)dzv, (T*)dzq, ldz, B, F, seed, thr, ds)));
    return vqa_check_launch("vqa_fusion_bwd");
}

int vqa_tanh_dropout_bwd(const float* slabs, int nslab, size_t slab_stride, const float* t_pre, int ld_t, void* dz,
                         int ldz, int M, int N, unsigned seed, float p, int dtype, hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && t_pre && dz && M >= 0 && N > 0 && ldz >= N && ld_t >= N, "vqa_tanh_dropout_bwd: bad arguments");
    VQA_REQUIRE(p >= 0.f && p < 1.f, "vqa_tanh_dropout_bwd: p must be in [0, 1)");
    if (M == 0) return 0;
    const unsigned thr = drop_threshold(p);
    const float ds = 1.0f / (1.0f - p);
    DISPATCH_T(dtype, (k_tanh_drop_bwd<T><<<grid_for((long)M * N, 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, t_pre, ld_t, (T*)dz, ldz, M, N, seed, thr, ds)));
    return vqa_check_launch("vqa_tanh_dropout_bwd");
}

int vqa_colsum(const void* X, int ld, float* out, int M, int N, int dtype, hipStream_t stream) {
    VQA_REQUIRE(X && out && M >= 0 && N > 0 && ld >= N, "vqa_colsum: bad arguments");
    DISPATCH_T(dtype, (k_colsum<T><<<(N + 31) / 32, 256, 0, stream>>>((const T*)X, ld, out, M, N)));
    return vqa_check_launch("vqa_colsum");
}

int vqa_slab_sum_multi(const VqaSlabSumDesc* descs, int n, hipStream_t stream) {
    VQA_REQUIRE(descs && n >= 1 && n <= VQA_SLABSUM_MAX, "vqa_slab_sum_multi: 1..8 problems");
    SlabSumBatch batch;
    batch.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const VqaSlabSumDesc& d = descs[i];
        VQA_REQUIRE(d.slabs && d.out && d.nslab >= 1 && d.count > 0 && (d.nslab == 1 || d.slab_stride >= (size_t)d.count),
                    "vqa_slab_sum_multi: bad problem");
        batch.d[i] = d;
        batch.first_block[i] = blocks;
        blocks += (int)((d.count + 1023) / 1024);
    }
    batch.first_block[n] = blocks;
    k_slab_sum_multi<<<blocks, 256, 0, stream>>>(batch);
    return vqa_check_launch("vqa_slab_sum_multi");
}

int vqa_slab_sum_transpose(const float* slabs, int nslab, size_t slab_stride, int M, int N, float* out, int ld_out,
                           hipStream_t stream) {
    VQA_REQUIRE(slabs && out && nslab >= 1 && M >= 0 && N >= 0 && ld_out >= M, "vqa_slab_sum_transpose: bad arguments");
    VQA_REQUIRE(nslab == 1 || slab_stride >= (size_t)M * N, "vqa_slab_sum_transpose: slab_stride too small");
    if (M == 0 || N == 0) return 0;
    const bool vec = M % 4 == 0 && N % 4 == 0 && ld_out % 4 == 0 && (slab_stride & 3) == 0 &&
                     ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0;
    if (vec) k_slab_sum_transpose4<<<dim3((N + 63) / 64, (M + 63) / 64), 256, 0, stream>>>(slabs, nslab, slab_stride, M, N, out, ld_out);
    else k_slab_sum_transpose<<<dim3((N + 31) / 32, (M + 31) / 32), 256, 0, stream>>>(slabs, nslab, slab_stride, M, N, out, ld_out);
    return vqa_check_launch("vqa_slab_sum_transpose");
}

int vqa_colsum_multi(const VqaColsumDesc* descs, int n, hipStream_t stream) {
    ColsumBatch batch;
    int blocks = 0;
    if (int rc = colsum_fill(descs, n, batch, blocks)) return rc;
    k_colsum_multi<<<blocks, 256, 0, stream>>>(batch);
    return vqa_check_launch("vqa_colsum_multi");
}

int vqa_colsum_rows(const void* X, int ld, float* out, const int32_t* rows, int N, int dtype, hipStream_t stream) {
    VQA_REQUIRE(X && out && rows && N > 0 && ld >= N, "vqa_colsum_rows: bad arguments");
    DISPATCH_T(dtype, (k_colsum_rows<T><<<(N + 31) / 32, 256, 0, stream>>>((const T*)X, ld, out, rows, N)));
    return vqa_check_launch("vqa_colsum_rows");
}

int vqa_lstm_cell_bwd_packed(const float* slabs, int nslab, size_t slab_stride, const float* dq, int ld_dq,
                             const float* dh_ext, const void* gates, const float* cs, const void* hs, float* dc_state,
                             void* dG, void* dG_state, void* hprev, const int32_t* lengths, const int32_t* perm, const int32_t* active,
                             const int32_t* offsets, int B, int H, int t, int L, int dtype, hipStream_t stream) {
    VQA_REQUIRE(dq && gates && cs && hs && dc_state && dG && dG_state && hprev && lengths && perm && active && offsets &&
                (nslab == 0 || slabs), "vqa_lstm_cell_bwd_packed: null pointer");
    VQA_REQUIRE(B >= 0 && H > 0 && H % 4 == 0 && t >= 0 && t < L && ld_dq >= H && ld_dq % 4 == 0 &&
                (slab_stride % 4 == 0 || nslab <= 1), "vqa_lstm_cell_bwd_packed: bad sizes");
    VQA_REQUIRE(t + 1 == L || nslab >= 1, "vqa_lstm_cell_bwd_packed: steps before the last need the recurrent slabs");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_lstm_cell_bwd<T><<<grid_for((long)B * (H / 4), 256), 256, 0, stream>>>(
        slabs, nslab, slab_stride, dq, ld_dq, dh_ext, (const T*)gates, cs, (const T*)hs, dc_state, (T*)dG, (T*)dG_state, (T*)hprev,
        lengths, perm, active, offsets, H, t, L)));
    return vqa_check_launch("vqa_lstm_cell_bwd_packed");
}

int vqa_embedding_grad(const float* slabs, int nslab, size_t slab_stride, int ld_dx, const int32_t* packed_tokens,
                       const int32_t* total_rows, const void* x_act, int ld_x, float* dE, int V, int E, int dtype,
                       hipStream_t stream) {
    VQA_REQUIRE(slabs && nslab >= 1 && packed_tokens && total_rows && dE && V > 0 && E > 0 && E <= 1024 && ld_dx >= E,
                "vqa_embedding_grad: bad arguments (E <= 1024)");
    VQA_REQUIRE(!x_act || ld_x >= E, "vqa_embedding_grad: bad ld_x");
    DISPATCH_T(dtype, (k_embedding_grad<T><<<(V + 3) / 4, 256, 0, stream>>>(
        slabs, nslab, slab_stride, ld_dx, packed_tokens, total_rows, (const T*)x_act, ld_x, dE, V, E)));
    return vqa_check_launch("vqa_embedding_grad");
}

int vqa_cast_pad(const float* src, int ld_src, void* dst, int ld_dst, int M, int N, int dtype, hipStream_t stream) {
    VQA_REQUIRE(src && dst && M >= 0 && N > 0 && ld_src >= N && ld_dst >= N, "vqa_cast_pad: bad arguments");
    if (M == 0) return 0;
    DISPATCH_T(dtype, (k_cast_pad<T><<<grid_for((long)M * ld_dst, 256), 256, 0, stream>>>(src, ld_src, (T*)dst, ld_dst, M, N)));
    return vqa_check_launch("vqa_cast_pad");
}

int vqa_softmax_ce(const float* logits, int ld, const int64_t* target, float* loss_rows, void* dlogits, int ld_d,
                   int B, int C, float grad_scale, int dtype, int32_t* status, hipStream_t stream) {
    VQA_REQUIRE(logits && target && loss_rows && B >= 0 && C > 0 && ld >= C, "vqa_softmax_ce: bad arguments");
    VQA_REQUIRE(!dlogits || ld_d >= C, "vqa_softmax_ce: ld_d < C");
    if (B == 0) return 0;
    DISPATCH_T(dtype, (k_softmax_ce<T><<<B, 256, 0, stream>>>(logits, ld, target, loss_rows, (T*)dlogits, ld_d, C, grad_scale, status)));
    return vqa_check_launch("vqa_softmax_ce");
}

int vqa_sum_scale(const float* x, float* out, int n, float scale, hipStream_t stream) {
    VQA_REQUIRE(x && out && n >= 0, "vqa_sum_scale: bad arguments");
    k_sum_scale<<<1, 256, 0, stream>>>(x, out, n, scale);
    return vqa_check_launch("vqa_sum_scale");
}

int vqa_argmax_accuracy(const float* logits, int ld, int B, int C, const int64_t* answers, const int32_t* type_ids,
                        int n_types, int64_t* pred, uint64_t* counters, hipStream_t stream) {
    VQA_REQUIRE(logits && pred && B >= 0 && C > 0 && ld >= C, "vqa_argmax_accuracy: bad arguments");
    if (B == 0) return 0;
    k_argmax_accuracy<<<(B + 3) / 4, 256, 0, stream>>>(logits, ld, B, C, answers, type_ids, n_types, pred,
                                                       (unsigned long long*)counters);
    return vqa_check_launch("vqa_argmax_accuracy");
}

int vqa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, int32_t* status,
                  hipStream_t stream) {
    VQA_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "vqa_adam_step: bad arguments");
    VQA_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                (((uintptr_t)shadow_bf16) & 7) == 0, "vqa_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    k_adam<<<grid_for((n + 3) / 4, 256), 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n, lr,
                                                           beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)),
                                                           grad_scale, status, step);
    return vqa_check_launch("vqa_adam_step");
}

}  // extern "C"
