// Weight-stationary question encoder: the recurrent steps of a frozen LSTM layer -- or the whole encoder, sequence
// packing and the two tabulated steps included -- in ONE launch (gfx950).
//
// The per-step form (vqa_gemm_slabs + vqa_lstm_cell_packed_table) streams the 46 MB of W_hh through every CU
// once per step and pays two dependent launches per step.  Here W_hh is loaded ONCE, into registers: the chip
// has 128 MB of vector registers, W_hh (4H x H bf16, H = 2400) is 46 MB.  Workgroup g owns RES_NH = 12 hidden
// units, i.e. 48 gate rows of W_hh, as 3 MFMA fragments x 19 K sub-steps per wave (228 registers per lane); its
// four waves split K into quarters.  One step of the recurrence is then, per workgroup:
//
//   gates[48][rows] = W_slice[48][K] . h_{t-1}[rows][K]^T      v_mfma_f32_16x16x32_bf16, first operand = W, so a
//                                                             lane ends up with the four gates (i, f, g, o) of ONE
//                                                             (batch row, hidden unit) in one accumulator
//   sum of the four K quarters through LDS (fixed order), + the tabulated input projection, cell update with the
//   cell state resident in LDS, h_t's 12 columns written to the other slot of a two-slot ring in global memory
//   (8-byte write-through stores after the step's last pass), q for the rows whose last token this was;
//   grid barrier (every workgroup needs all of h_t for the next step).
//
// Only the live rows of a step (sequences are sorted by length: the first active[t]) are read, multiplied or
// updated, in passes of 64 rows.  The grid barrier is eight arrival counters (64-bit, monotonic across launches, one
// per value of blockIdx.x % 8) polled by one wave, then one agent-scope acquire; the grid is at most one workgroup
// per CU and every wave needs a whole SIMD's registers, so all workgroups are resident together.  Spins are bounded
// by the wall clock: a barrier that does not complete sets an abort word, every workgroup leaves and the status word
// tells the host.
#include "common.h"
#include "vqa_hip.h"
#include <mutex>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

#ifndef RES_NH
#define RES_NH VQA_LSTM_RESIDENT_UNITS        // hidden units per workgroup
#endif
#define RES_KSUB 19                           // 32-deep K sub-steps per wave
#define RES_PITCH VQA_LSTM_RESIDENT_PITCH     // 4 waves x 19 x 32: the padded contraction length
#ifndef RES_RF
#define RES_RF 4                              // 16-row fragments of h per pass (64 batch rows; 128-row passes with 3 buffers: +2 %)
#endif
#define RES_NJ (RES_NH / 4)                   // 16-row fragments of the W slice (4 gate rows per hidden unit)
#ifndef RES_NBUF
#define RES_NBUF 6                            // fragment buffers per wave: sub-steps s .. s + 5 requested or in use
#endif
#define RES_MAXL VQA_LSTM_RESIDENT_MAX_STEPS   // longest question
#define RES_CELLS ((RES_RF * RES_NJ + 3) / 4) // (fragment, W fragment) pairs a wave updates per pass (the last wave may have fewer)

static_assert(RES_NH * 4 == RES_NJ * 16, "a workgroup's gate rows are whole MFMA fragments");
static_assert(RES_RF >= 3 && RES_RF <= 8, "fragment counts the pass switch below knows");
static_assert(RES_PITCH == 4 * RES_KSUB * 32, "the ring pitch is the padded contraction length");

// synchronisation words (u64 each, one 128-byte line per word): eight arrival counters (workgroup g counts into
// shard g & 7), an abort word, the number of barriers completed by earlier launches
#define SYNC_SHARD(g) (16 * (g))
#define SYNC_ABORT (16 * 8)
#define SYNC_EPOCH (16 * 9)

#ifdef VQA_DEV
// development builds: wall-clock stamps of two workgroups (first and middle), [wg 2][step 16][stamp 12]
__device__ unsigned long long res_trace[2 * 16 * 12];
#define RES_STAMP(step, k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && (step) < 16 && (k) < 12) \
    res_trace[((blockIdx.x != 0) * 16 + (step)) * 12 + (k)] = wall_clock64(); } while (0)
extern "C" int vqa_dev_resident_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(res_trace), sizeof(res_trace));
}
#else
#define RES_STAMP(step, k) do { } while (0)
#endif

// LDS traffic of this wave has landed, then a raw barrier: pending global stores stay in flight across it
// (__syncthreads() would wait for them too)
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

// Grid barrier after a step whose h columns were stored write-through (sc1): every wave waits for its own stores,
// one lane counts the workgroup into its shard, one wave polls all eight shards, then ONE agent-scope acquire
// (invalidates this CU's L1) that is waited for before any wave reads h.
__device__ __forceinline__ bool grid_barrier(u64* sync, u64 epoch, int nwg, int* ok_s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_fetch_add(sync + SYNC_SHARD(blockIdx.x & 7), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 want = lane < 8 ? (u64)((nwg - lane + 7) >> 3) * epoch : 0;   // members of shard `lane` x barriers so far
        const u64* word = sync + (lane < 8 ? SYNC_SHARD(lane) : SYNC_ABORT);
        int good = 1;
        const u64 t0 = wall_clock64();
        for (;;) {
            const u64 v = lane <= 8 ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (__all(lane >= 8 || v >= want)) break;
            if (__any(lane == 8 && v != 0) || wall_clock64() - t0 > 1000000000ull) {    // 10 s of the 100 MHz clock
                if (lane == 0) __hip_atomic_store(sync + SYNC_ABORT, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *ok_s = good;
    }
    __syncthreads();
    return *ok_s != 0;
}

// One pass of the recurrent product for NF 16-row fragments of h (compile-time, so that the loop is straight-line
// code and the compiler counts the loads in flight exactly): acc[i][j] += W fragment j x h fragment i over this
// wave's quarter of K.  Fragment i, sub-step s is chunk kc0 + 4 s of 16 consecutive rows: 256 contiguous bytes per
// chunk.  RES_NBUF - 1 sub-steps are requested ahead.  (Requesting the next pass's first sub-steps before this
// pass's reduction was measured: the fragment buffers then live across the cell update, the kernel spills, and the
// launch takes 2-25 % longer.)
template <int NF>
__device__ __forceinline__ void product_pass(const bf16x8_t (&wf)[RES_KSUB][RES_NJ], const bf16_t* hp0, size_t sub_stride,
                                             f32x4 (&acc)[RES_RF][RES_NJ]) {
    bf16x8_t af[RES_NBUF][NF];
#pragma unroll
    for (int s = 0; s < RES_NBUF - 1; ++s)
#pragma unroll
        for (int i = 0; i < NF; ++i) af[s][i] = *reinterpret_cast<const bf16x8_t*>(hp0 + s * sub_stride + i * 128);
    __builtin_amdgcn_sched_barrier(0);       // (the scheduler would otherwise sink every load to just before its use)
#pragma unroll
    for (int s = 0; s < RES_KSUB; ++s) {
        if (s + RES_NBUF - 1 < RES_KSUB) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
                af[(s + RES_NBUF - 1) % RES_NBUF][i] =
                    *reinterpret_cast<const bf16x8_t*>(hp0 + (s + RES_NBUF - 1) * sub_stride + i * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < RES_NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], af[s % RES_NBUF][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_lstm_resident(const bf16_t* __restrict__ whh, int ldw, const bf16_t* __restrict__ xgu, int n_table,
                const int64_t* __restrict__ tok, const float* __restrict__ c0, bf16_t* hring,
                int* len, float* __restrict__ q32, bf16_t* __restrict__ qT, int ld_q,
                int B, int Bp, int H, int t_first, int L, int* perm, int* m_t,
                bf16_t* __restrict__ q_drop, unsigned seed, unsigned thr, float dscale, u64* sync, int* status,
                const float* __restrict__ hh_table) {
    extern __shared__ __align__(16) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                   // [4 waves][RF * NJ][64 lanes]
    float* hst = reinterpret_cast<float*>(smem + 4 * RES_RF * RES_NJ * 64 * 16);   // [B][RES_NH] h of the running step
    float* cst = hst + (size_t)B * RES_NH;                                         // [B][RES_NH] cell state
    int* ob_l = reinterpret_cast<int*>(cst + (size_t)B * RES_NH);                  // [B] original row of sorted row
    int* last_l = ob_l + B;                                                        // [B] step that ends the row
    int* id_l = last_l + B;                                                        // [B] token id at the current step
    __shared__ int ok_s;
    __shared__ u64 epoch_s;
    __shared__ int mt_l[RES_MAXL + 1];      // rows still running at step t
    __shared__ int hist_l[RES_MAXL + 2];    // t_first == 0: rows with exactly n steps, then rows with more than n steps

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nwg = gridDim.x;
    const int unit0 = blockIdx.x * RES_NH;
    RES_STAMP(15, 0);
    const int kc0 = wave * (RES_KSUB * 4) + (lane >> 4);             // this lane's first 8-element chunk of K

    // ring slots hold h as [K / 8 chunks][Bp rows][8] (Bp = VQA_LSTM_RESIDENT_RING_ROWS(B)): the 16 rows of a fragment are 256
    // contiguous bytes per chunk, so a fragment load (lane = row, chunk) reads whole 128-byte lines (row-major rows
    // cost one L1 tag lookup per lane: 4x the address-processing time)
    const size_t slot_elems = (size_t)Bp * RES_PITCH;
    const size_t sub_stride = (size_t)4 * Bp * 8;       // elements between sub-steps: four chunks
    bool publish_pending = false;                       // a step's h has been stored and not yet met by a barrier
    RES_STAMP(15, 1);
    if (t_first > 0) {
        // the caller packed the sequences and ran the steps before t_first
        for (int b = tid; b < B; b += 256) {
            const int ob = perm[b];
            int last = len[ob] - 1;
            if (last < 0) last += L;                    // all-pad row: index -1 selects the last step
            ob_l[b] = ob;
            last_l[b] = last;
        }
        for (int t = tid; t < L; t += 256) mt_l[t] = m_t[t];
        for (int i = tid; i < B * RES_NH; i += 256) {
            const int b = i / RES_NH, u = i % RES_NH;
            cst[i] = unit0 + u < H ? c0[(size_t)b * H + unit0 + u] : 0.f;
        }
    } else {
        // ---- the whole encoder in this launch.  Every workgroup packs the sequences for itself (the arithmetic of
        // vqa_sequence_pack: lengths, rows sorted by descending step count, ties in row order), then runs steps 0 and 1
        // of its hidden units from the two per-token tables (the arithmetic of vqa_lstm_cell01_table).
        for (int n = tid; n <= L + 1; n += 256) hist_l[n] = 0;
        __syncthreads();
        bool bad = false;
        for (int b = tid; b < B; b += 256) {
            int z = 0;
            for (int t0 = 0; t0 < L; t0 += 16) {        // 16 loads in flight (a plain loop waits for each one)
                int64_t v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = tok[(size_t)b * L + min(t0 + u, L - 1)];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (t0 + u < L) {
                        z += v[u] == 0;
                        bad |= v[u] < 0 || v[u] >= n_table;
                    }
            }
            const int n = z == L ? L : L - z;           // an all-pad row takes all L steps
            id_l[b] = n;
            atomicAdd(&hist_l[n], 1);
            if (blockIdx.x == 0 && len) len[b] = L - z;
        }
        if (bad && blockIdx.x == 0 && status) status[VQA_STATUS_BAD_TOKEN] = 1;
        __syncthreads();
        if (tid == 0) {                                 // hist_l[n] <- rows with more than n steps
            int above = 0;
            for (int n = L; n >= 0; --n) { const int here = hist_l[n]; hist_l[n] = above; above += here; }
        }
        __syncthreads();
        for (int t = tid; t < L; t += 256) {
            mt_l[t] = hist_l[t];
            if (blockIdx.x == 0 && m_t) m_t[t] = hist_l[t];
        }
        for (int b = tid; b < B; b += 256) {
            const int n = id_l[b];
            int before = 0;
            for (int o = 0; o < b; ++o) before += id_l[o] == n;
            const int pos = hist_l[n] + before;
            ob_l[pos] = b;
            last_l[pos] = n - 1;
            if (blockIdx.x == 0 && perm) perm[pos] = b;
        }
        __syncthreads();
        RES_STAMP(15, 2);
        const int rows1 = mt_l[1];
        bf16_t* h1 = hring + slot_elems;                // published into slot 1: the first barrier flips to it
        // three (row, four units) pieces per thread at a time: their token loads go out together, then their table
        // gathers (a piece at a time would pay the two dependent memory round trips once per piece)
        constexpr int PU = 3;
        const int npiece = B * (RES_NH / 4);
        for (int i0 = tid; i0 < npiece; i0 += 256 * PU) {
            int row[PU], unit[PU], ob[PU];
            bool live[PU], second[PU];
            int64_t id0[PU], id1[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int i = min(i0 + 256 * u, npiece - 1);
                row[u] = i / (RES_NH / 4);
                unit[u] = unit0 + 4 * (i % (RES_NH / 4));
                live[u] = i0 + 256 * u < npiece && unit[u] < H;
                unit[u] = min(unit[u], H - 4);
                ob[u] = ob_l[row[u]];
                second[u] = row[u] < rows1;
                id0[u] = tok[(size_t)ob[u] * L];
                id1[u] = tok[(size_t)ob[u] * L + 1];
            }
            u16x8 a0[PU], b0[PU], a1[PU], b1[PU];
            f32x4 hh[PU][4];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (id0[u] < 0 || id0[u] >= n_table) id0[u] = 0;
                if (id1[u] < 0 || id1[u] >= n_table) id1[u] = 0;
                // a unit's four gates are adjacent in xgu: four units = 32 contiguous bytes
                const u16x8* x0 = reinterpret_cast<const u16x8*>(xgu + ((size_t)id0[u] * H + unit[u]) * 4);
                const u16x8* x1 = reinterpret_cast<const u16x8*>(xgu + ((size_t)id1[u] * H + unit[u]) * 4);
                a0[u] = x0[0]; b0[u] = x0[1];
                a1[u] = x1[0]; b1[u] = x1[1];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    hh[u][g] = *reinterpret_cast<const f32x4*>(hh_table + (size_t)id0[u] * 4 * H + (size_t)g * H + unit[u]);
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (!live[u]) continue;
                const int last = last_l[row[u]];
                f32x4 cn, hn;
                auto emit = [&]() {                     // the row's question vector, when this step was its last
                    const size_t o = (size_t)ob[u] * ld_q + unit[u];
                    const u16x4 hb = u16x4{f32_to_bf16(hn[0]), f32_to_bf16(hn[1]), f32_to_bf16(hn[2]), f32_to_bf16(hn[3])};
                    *reinterpret_cast<f32x4*>(q32 + o) = hn;
                    *reinterpret_cast<u16x4*>(qT + o) = hb;
                    if (q_drop) {
                        u16x4 qd;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            qd[e] = f32_to_bf16(bf16_to_f32(hb[e]) * drop_mult(seed, (unsigned)(ob[u] * H + unit[u] + e), thr, dscale));
                        *reinterpret_cast<u16x4*>(q_drop + o) = qd;
                    }
                };
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u16x8& x = e < 2 ? a0[u] : b0[u];
                    const int o4 = (e & 1) * 4;
                    const float ig = fast_sigmoid(bf16_to_f32(x[o4])), fg = fast_sigmoid(bf16_to_f32(x[o4 + 1]));
                    const float gg = fast_tanh(bf16_to_f32(x[o4 + 2])), og = fast_sigmoid(bf16_to_f32(x[o4 + 3]));
                    cn[e] = fg * 0.f + ig * gg;         // (f * c_prev with c_prev = 0, as the one-step kernels compute it)
                    hn[e] = og * fast_tanh(cn[e]);
                }
                if (last == 0) emit();
                if (second[u]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u16x8& x = e < 2 ? a1[u] : b1[u];
                        const int o4 = (e & 1) * 4;
                        const float ig = fast_sigmoid(bf16_to_f32(x[o4]) + hh[u][0][e]), fg = fast_sigmoid(bf16_to_f32(x[o4 + 1]) + hh[u][1][e]);
                        const float gg = fast_tanh(bf16_to_f32(x[o4 + 2]) + hh[u][2][e]), og = fast_sigmoid(bf16_to_f32(x[o4 + 3]) + hh[u][3][e]);
                        cn[e] = fg * cn[e] + ig * gg;
                        hn[e] = og * fast_tanh(cn[e]);
                    }
                    if (last == 1) emit();
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) cst[row[u] * RES_NH + (unit[u] - unit0) + e] = cn[e];
                const u16x4 hb = u16x4{f32_to_bf16(hn[0]), f32_to_bf16(hn[1]), f32_to_bf16(hn[2]), f32_to_bf16(hn[3])};
                u64* dst = reinterpret_cast<u64*>(h1 + ((size_t)(unit[u] >> 3) * Bp + row[u]) * 8 + (unit[u] & 4));
                __hip_atomic_store(dst, __builtin_bit_cast(u64, hb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        publish_pending = true;
        RES_STAMP(15, 3);
    }
    // ---- the workgroup's slice of W_hh, once (requested after the tabulated steps, whose loads it would otherwise queue
    // behind, and landing while the first barrier is waited for): fragment j, lane (m, kc): W row gate(m & 3) * H + unit0 + 4 j + (m >> 2)
    bf16x8_t wf[RES_KSUB][RES_NJ];
    {
        const int m = lane & 15;
#pragma unroll
        for (int j = 0; j < RES_NJ; ++j) {
            const int unit = unit0 + 4 * j + (m >> 2);
            const bf16_t* wrow = whh + (size_t)((m & 3) * H + min(unit, H - 1)) * ldw;
#pragma unroll
            for (int s = 0; s < RES_KSUB; ++s) {
                const int k = (kc0 + 4 * s) * 8;
                u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (unit < H && k < H) v = *reinterpret_cast<const u16x8*>(wrow + k);
                wf[s][j] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    }
    if (tid == 0) epoch_s = __hip_atomic_load(sync + SYNC_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const u64 epoch0 = epoch_s;
    u64 nbar = 0;

    int slot = 0;
    bool ids_ready = false;
    for (int t = t_first > 0 ? t_first : 2; t < L; ++t) {
        const int rows = __builtin_amdgcn_readfirstlane(min(mt_l[t], B));
        if (rows <= 0) break;                            // (active[] never grows again: every workgroup leaves here)
        RES_STAMP(t - t_first, 0);
        if (publish_pending) {
            ++nbar;
            if (!grid_barrier(sync, epoch0 + nbar, nwg, &ok_s)) {
                if (blockIdx.x == 0 && tid == 0 && status) status[VQA_STATUS_RESIDENT_TIMEOUT] = 1;
                return;
            }
            slot ^= 1;
        }
        RES_STAMP(t - t_first, 1);
        const bf16_t* hin = hring + (size_t)slot * slot_elems;
        bf16_t* hout = hring + (size_t)(slot ^ 1) * slot_elems;
        if (!ids_ready) {                               // (later steps: fetched at the end of the step before)
            for (int b = tid; b < rows; b += 256) {
                int64_t id = tok[(size_t)ob_l[b] * L + t];
                if (id < 0 || id >= n_table) id = 0;    // as the gather kernels: ids outside the table read the pad row
                id_l[b] = (int)id;
            }
            __syncthreads();
        }

        const int nfrag = (rows + 15) >> 4;
        // The workgroups of an XCD (blockIdx.x % 8 under round-robin placement) start at different passes and wrap:
        // at any moment they read different rows of h, so most of a workgroup's lines were already brought into the
        // XCD's L2 by a neighbour instead of every workgroup waiting on the same miss.
        const int npass = (nfrag + RES_RF - 1) / RES_RF;
#ifdef RES_NO_ROTATE
        const int rot = 0;
#else
        const int rot = (blockIdx.x >> 3) % npass;
#endif
        for (int ps = 0; ps < npass; ++ps) {
            const int f0 = ((ps + rot) % npass) * RES_RF;
            const int nf = min(RES_RF, nfrag - f0);
            // this wave's cells of the pass: (fragment i, W fragment j) pairs p = RES_CELLS wave ...  The unit's four
            // gates are adjacent in the table: one 8-byte load per cell, in flight under the MFMAs (always issued, from a
            // clamped address where the cell is not live: loads under a branch cannot be counted by the compiler, which
            // then waits for nearly everything in flight).
            u32x2 xgv[RES_CELLS];
            int crow[RES_CELLS], cu[RES_CELLS];
            bool clive[RES_CELLS];
#pragma unroll
            for (int q = 0; q < RES_CELLS; ++q) {
                const int p = wave * RES_CELLS + q, i = p / RES_NJ, j = p - RES_NJ * i;
                crow[q] = (f0 + i) * 16 + (lane & 15);
                cu[q] = 4 * j + (lane >> 4);
                clive[q] = p < RES_RF * RES_NJ && i < nf && crow[q] < rows && unit0 + cu[q] < H;
                xgv[q] = *reinterpret_cast<const u32x2*>(xgu + ((size_t)id_l[min(crow[q], rows - 1)] * H + min(unit0 + cu[q], H - 1)) * 4);
            }

            f32x4 acc[RES_RF][RES_NJ];
#pragma unroll
            for (int i = 0; i < RES_RF; ++i)
#pragma unroll
                for (int j = 0; j < RES_NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16_t* hp0 = hin + ((size_t)kc0 * Bp + f0 * 16 + (lane & 15)) * 8;
            switch (nf) {                      // rows past the live ones of the last fragment are stale, never used
#if RES_RF >= 8
                case 8: product_pass<8>(wf, hp0, sub_stride, acc); break;
#endif
#if RES_RF >= 7
                case 7: product_pass<7>(wf, hp0, sub_stride, acc); break;
#endif
#if RES_RF >= 6
                case 6: product_pass<6>(wf, hp0, sub_stride, acc); break;
#endif
#if RES_RF >= 5
                case 5: product_pass<5>(wf, hp0, sub_stride, acc); break;
#endif
#if RES_RF >= 4
                case 4: product_pass<4>(wf, hp0, sub_stride, acc); break;
#endif
                case 3: product_pass<3>(wf, hp0, sub_stride, acc); break;
                case 2: product_pass<2>(wf, hp0, sub_stride, acc); break;
                default: product_pass<1>(wf, hp0, sub_stride, acc); break;
            }

            RES_STAMP(t - t_first, 2 + 2 * ps);
            // the four K quarters meet in LDS
#pragma unroll
            for (int i = 0; i < RES_RF; ++i)
                if (i < nf) {
#pragma unroll
                    for (int j = 0; j < RES_NJ; ++j) red[(wave * (RES_RF * RES_NJ) + i * RES_NJ + j) * 64 + lane] = acc[i][j];
                }
            LDS_BARRIER();
#pragma unroll
            for (int q = 0; q < RES_CELLS; ++q) {
                if (!clive[q]) continue;
                const int p = wave * RES_CELLS + q;
                f32x4 g = red[(0 * (RES_RF * RES_NJ) + p) * 64 + lane];
                g += red[(1 * (RES_RF * RES_NJ) + p) * 64 + lane];
                g += red[(2 * (RES_RF * RES_NJ) + p) * 64 + lane];
                g += red[(3 * (RES_RF * RES_NJ) + p) * 64 + lane];
                g[0] += __uint_as_float(xgv[q][0] << 16);
                g[1] += __uint_as_float(xgv[q][0] & 0xFFFF0000u);
                g[2] += __uint_as_float(xgv[q][1] << 16);
                g[3] += __uint_as_float(xgv[q][1] & 0xFFFF0000u);
                const int o = crow[q] * RES_NH + cu[q];
                const float ig = fast_sigmoid(g[0]), fg = fast_sigmoid(g[1]);
                const float gg = fast_tanh(g[2]), og = fast_sigmoid(g[3]);
                const float cn = fg * cst[o] + ig * gg;
                cst[o] = cn;
                hst[o] = og * fast_tanh(cn);
            }
            LDS_BARRIER();                    // red may be rewritten by the next pass
            RES_STAMP(t - t_first, 3 + 2 * ps);
        }
        // The step's h leaves after the last pass (a store between the passes would sit in the same in-order counter
        // as the next pass's loads): four units of one row per 8-byte write-through store, which needs no L2
        // write-back before the grid barrier, only this wave's vmcnt(0); rows that end here also write q.
        for (int i = tid; i < rows * (RES_NH / 4); i += 256) {
            const int row = i / (RES_NH / 4), pc = i % (RES_NH / 4), unit = unit0 + 4 * pc;
            if (unit >= H) continue;
            const f32x4 hn = *reinterpret_cast<const f32x4*>(hst + row * RES_NH + 4 * pc);
            const u16x4 hb = u16x4{f32_to_bf16(hn[0]), f32_to_bf16(hn[1]), f32_to_bf16(hn[2]), f32_to_bf16(hn[3])};
            u64* dst = reinterpret_cast<u64*>(hout + ((size_t)(unit >> 3) * Bp + row) * 8 + (unit & 4));
            __hip_atomic_store(dst, __builtin_bit_cast(u64, hb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last_l[row] == t) {
                const int ob = ob_l[row];
                const size_t o = (size_t)ob * ld_q + unit;
                *reinterpret_cast<f32x4*>(q32 + o) = hn;
                *reinterpret_cast<u16x4*>(qT + o) = hb;
                if (q_drop) {
                    u16x4 qd;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        qd[e] = f32_to_bf16(bf16_to_f32(hb[e]) * drop_mult(seed, (unsigned)(ob * H + unit + e), thr, dscale));
                    *reinterpret_cast<u16x4*>(q_drop + o) = qd;
                }
            }
        }
        // the next step's token ids, fetched under the stores and the barrier (every pass of this step is done with id_l)
        if (t + 1 < L) {
            const int rows_next = min(mt_l[t + 1], B);
            for (int b = tid; b < rows_next; b += 256) {
                int64_t id = tok[(size_t)ob_l[b] * L + t + 1];
                if (id < 0 || id >= n_table) id = 0;
                id_l[b] = (int)id;
            }
            ids_ready = true;                           // (published to the other waves by the barrier's __syncthreads)
        }
        publish_pending = true;
    }
    if (nbar != 0 && blockIdx.x == 0 && tid == 0)
        __hip_atomic_store(sync + SYNC_EPOCH, epoch0 + nbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t resident_lds_bytes(int B) {
    return (size_t)4 * RES_RF * RES_NJ * 64 * 16 + (size_t)2 * B * RES_NH * 4 + (size_t)3 * B * 4;
}

extern "C" {

int vqa_lstm_resident_workgroups(int H) {
    if (H <= 0 || H % 8 != 0 || H > RES_PITCH) return 0;
    return (H + RES_NH - 1) / RES_NH;
}

int vqa_lstm_resident(const void* w_hh, int ld_w, const void* xg_table_u, const float* hh_table, int n_table_rows,
                      const int64_t* tokens, const float* c, void* h_ring, int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                      int B, int H, int t_first, int L, int32_t* perm, int32_t* active, void* q_drop,
                      unsigned drop_seed, float drop_p, uint64_t* sync_words, int32_t* status, hipStream_t stream) {
    VQA_REQUIRE(w_hh && xg_table_u && tokens && h_ring && q_f32 && q_t && sync_words, "vqa_lstm_resident: null pointer");
    VQA_REQUIRE(t_first == 0 ? (hh_table != nullptr && L >= 2) : (c && lengths && perm && active),
                "vqa_lstm_resident: t_first = 0 needs hh_table and L >= 2, t_first >= 1 needs c, lengths, perm and active");
    VQA_REQUIRE(H > 0 && H % 8 == 0 && H <= RES_PITCH, "vqa_lstm_resident: H must be a multiple of 8 and at most VQA_LSTM_RESIDENT_PITCH");
    VQA_REQUIRE(B >= 0 && B <= VQA_LSTM_RESIDENT_MAX_ROWS, "vqa_lstm_resident: B must be at most VQA_LSTM_RESIDENT_MAX_ROWS");
    VQA_REQUIRE(t_first >= 0 && t_first <= L && L <= RES_MAXL && n_table_rows > 0 && ld_w >= H && ld_w % 8 == 0 && ld_q >= H &&
                ld_q % 4 == 0, "vqa_lstm_resident: bad sizes");
    VQA_REQUIRE((((uintptr_t)q_f32) & 15) == 0 && (((uintptr_t)q_t) & 7) == 0 && (((uintptr_t)q_drop) & 7) == 0,
                "vqa_lstm_resident: q_f32 must be 16-byte aligned, q_t and q_drop 8-byte aligned");
    VQA_REQUIRE((((uintptr_t)w_hh) & 15) == 0 && (((uintptr_t)h_ring) & 15) == 0 && (((uintptr_t)xg_table_u) & 15) == 0 &&
                (((uintptr_t)hh_table) & 15) == 0 && (((uintptr_t)sync_words) & 127) == 0, "vqa_lstm_resident: misaligned pointer");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_lstm_resident: drop_p must be in [0, 1)");
    if (B == 0 || t_first == L) return 0;
    const int nwg = (H + RES_NH - 1) / RES_NH;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return vqa_set_error(-5, "vqa_lstm_resident: cannot query the device");
    if (stream) {      // the compute-unit count and the kernel attribute are the current device's: the stream must be on it
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess || (int)sdev != dev)
            return vqa_set_error(-22, "vqa_lstm_resident: `stream` does not belong to the current device");
    }
    // every workgroup takes a whole CU's registers, and all of them have to be resident for the grid barrier
    VQA_REQUIRE(nwg <= cus, "vqa_lstm_resident: more workgroups than compute units (use the per-step entry points)");
    const size_t lds = resident_lds_bytes(B);
    static std::mutex attr_mutex;
    static bool attr_set[64] = {};
    std::lock_guard<std::mutex> lock(attr_mutex);           // (a launch is a few microseconds; host threads take turns)
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)k_lstm_resident, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)resident_lds_bytes(VQA_LSTM_RESIDENT_MAX_ROWS)) != hipSuccess)
            return vqa_set_error(-5, "vqa_lstm_resident: cannot raise the dynamic LDS limit");
        if (tracked) attr_set[dev] = true;
    }
    const unsigned thr = drop_threshold(drop_p);
    const float dsc = 1.0f / (1.0f - drop_p);
    k_lstm_resident<<<nwg, 256, lds, stream>>>((const bf16_t*)w_hh, ld_w, (const bf16_t*)xg_table_u, n_table_rows, tokens, c,
                                               (bf16_t*)h_ring, lengths, q_f32, (bf16_t*)q_t, ld_q, B, VQA_LSTM_RESIDENT_RING_ROWS(B), H, t_first, L, perm,
                                               active, (bf16_t*)q_drop, drop_seed, thr, dsc,
                                               reinterpret_cast<u64*>(sync_words), status, hh_table);
    return vqa_check_launch("vqa_lstm_resident");
}

}  // extern "C"
