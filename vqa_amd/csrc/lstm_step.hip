// One LSTM step of the question encoder in one launch (gfx950, bf16 storage): the recurrent product
// h_{t-1} W_hh^T on the matrix cores and the cell update in its epilogue.
//
// The two-launch form (a split-K product into fp32 partial sums, then a cell kernel that sums them) pays two
// dependent launches and a round trip of the partial sums per step; the steps of a recurrence cannot overlap, so
// that fixed cost is paid L times.  Here a workgroup owns 128 batch rows x 16 hidden units and computes ALL FOUR
// gates of those units over the whole contraction: its B operand is the 64 rows {gate * H + u0 + i} of W_hh, and in
// the MFMA accumulator layout a lane then holds the i, f, g, o pre-activations of one (row, unit) pair in the same
// register slot of four fragments -- the cell update needs no exchange between lanes.
//
// Reference arithmetic: torch.nn.LSTM as called at vqa_model/models/seq2vec.py:38, 43 (gate order i, f, g, o), last
// valid step selection of seq2vec.py:16-25, on packed (length-sorted) rows.
#include "common.h"
#include "vqa_hip.h"
#include "gemm_tiles.h"

namespace {

constexpr int LS_BM = 128, LS_BN = 64, LS_NW = 4, LS_NS = 3;

__global__ void __launch_bounds__(256)
k_lstm_step(const bf16_t* __restrict__ h_prev, int ld_h, const bf16_t* __restrict__ w_hh, int ld_w,
            const bf16_t* __restrict__ xg, int ld_xg, int n_table, const int64_t* __restrict__ tok,
            float* __restrict__ c, bf16_t* __restrict__ h_out, int ld_ho, const int* __restrict__ len,
            float* __restrict__ q32, bf16_t* __restrict__ qT, int ld_q, int H, int t, int L,
            const int* __restrict__ perm, const int* __restrict__ m_t, const int* __restrict__ offs,
            bf16_t* __restrict__ hs, int ld_hs, bf16_t* __restrict__ q_drop, unsigned seed, unsigned thr, float dscale) {
    typedef Tile<LS_BM, true> TA;
    typedef Tile<LS_BN, true> TB;
    constexpr int STAGE = TA::BYTES + TB::BYTES;                     // 16 KiB + 8 KiB
    constexpr int NIA = TA::template ni<LS_NW>(), NIB = TB::template ni<LS_NW>();
    static_assert(NIA + NIB == 6, "the counted wait assumes 6 DMA instructions per wave and K tile");
    __shared__ __attribute__((aligned(1024))) char lds[LS_NS * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int live = __builtin_amdgcn_readfirstlane(m_t[t]);         // rows still running at this step (a prefix)
    const int n_slices = H >> 4;
    const int slice = (int)blockIdx.x % n_slices, m0 = ((int)blockIdx.x / n_slices) * LS_BM;
    if (m0 >= live) return;                                          // uniform over the workgroup, before any barrier
    const int u0 = slice * 16;
    const int K = H, nfull = K / GEMM_BK;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA sources: A = rows m0.. of h_{t-1} (rows past the live count re-read the last live row: they feed output
    // rows that are never stored), B = W_hh rows {gate * H + u0 + unit}: tile row r is gate r >> 4, unit r & 15
    const bf16_t* sa[NIA];
    const bf16_t* sb[NIB];
    dma_setup<LS_BM, true, LS_NW>(h_prev, ld_h, m0, live, wave, lane, sa);
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        int r, ch;
        TB::slot(wave * NIB + i, lane, r, ch);
        sb[i] = w_hh + (size_t)((r >> 4) * H + u0 + (r & 15)) * ld_w + ch * 8;
    }
    auto issue = [&](int kt) {
        char* st = lds + (kt % LS_NS) * STAGE;
        dma_issue<LS_BM, true, LS_NW>(sa, (size_t)kt * GEMM_BK, st, wave);
        dma_issue<LS_BN, true, LS_NW>(sb, (size_t)kt * GEMM_BK, st + TA::BYTES, wave);
    };
    auto compute = [&](const char* la) {
        const char* lb = la + TA::BYTES;
        bf16x8_t fa[2][2], fb[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[kk][i] = frag_read<LS_BM, true>(la, wave * 32 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[kk][j] = frag_read<LS_BN, true>(lb, j * 16, kk, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    };

    // three ring slots, two K tiles in flight: tile kt + 2 is issued right after the barrier that opens tile kt (every
    // wave is then done with the slot it goes to), the counted wait leaves the younger tile's 6 instructions in flight
    if (nfull > 0) issue(0);
    if (nfull > 1) issue(1);
    for (int kt = 0; kt < nfull; ++kt) {
        if (kt + 1 < nfull) VMCNT_BARRIER(6); else VMCNT_BARRIER(0);
        if (kt + 2 < nfull) issue(kt + 2);
        compute(lds + (kt % LS_NS) * STAGE);
    }
    if (K > nfull * GEMM_BK) {                              // partial last K tile: through registers with zero fill
        VMCNT_BARRIER(0);
        const int k0 = nfull * GEMM_BK;
        u32x4 ra[LS_BM * GEMM_BK / (512 * LS_NW)], rb[LS_BN * GEMM_BK / (512 * LS_NW)];
        tile_load<LS_BM, true, LS_NW>(h_prev, ld_h, m0, live, k0, K, ra);
#pragma unroll
        for (int i = 0; i < LS_BN * GEMM_BK / (512 * LS_NW); ++i) {
            const int q = threadIdx.x + LS_NW * 64 * i, r = q >> 3, k = k0 + ((q & 7) << 3);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (k < K) v = *reinterpret_cast<const u32x4*>(w_hh + (size_t)((r >> 4) * H + u0 + (r & 15)) * ld_w + k);
            rb[i] = v;
        }
        tile_store<LS_BM, true, LS_NW>(lds, ra);
        tile_store<LS_BN, true, LS_NW>(lds + TA::BYTES, rb);
        VMCNT_BARRIER(0);
        compute(lds);
    }

    // cell update: lane (g = lane >> 4, col = lane & 15) holds the four gates of unit u0 + col for the rows
    // wave * 32 + 16 i + 4 g + r of the tile in acc[i][gate][r]
    const int g = lane >> 4, unit = u0 + (lane & 15);
    const long xrow0 = offs[t];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = m0 + wave * 32 + 16 * i + 4 * g + r;       // sorted row
            if (b >= live) continue;
            const int ob = perm[b];                                  // original row
            long xrow = xrow0 + b;
            if (tok) {                                               // input projections gathered by token id
                int64_t id = tok[(size_t)ob * L + t];
                if (id < 0 || id >= n_table) id = 0;                 // as the gather: ids outside the table read the pad row
                xrow = (long)id;
            }
            const bf16_t* xr = xg + (size_t)xrow * ld_xg + unit;
            const float pi = acc[i][0][r] + bf16_to_f32(xr[0]);
            const float pf = acc[i][1][r] + bf16_to_f32(xr[H]);
            const float pg = acc[i][2][r] + bf16_to_f32(xr[2 * H]);
            const float po = acc[i][3][r] + bf16_to_f32(xr[3 * H]);
            const float cprev = c[(size_t)b * H + unit];
            const float ig = fast_sigmoid(pi), fg = fast_sigmoid(pf), gg = fast_tanh(pg), og = fast_sigmoid(po);
            const float cn = fg * cprev + ig * gg;
            const float hn = og * fast_tanh(cn);
            const bf16_t hb = f32_to_bf16(hn);
            c[(size_t)b * H + unit] = cn;
            h_out[(size_t)b * ld_ho + unit] = hb;
            if (hs) hs[(size_t)(xrow0 + b) * ld_hs + unit] = hb;     // every step's output, packed rows
            int last = len[ob] - 1;
            if (last < 0) last += L;                                 // all-pad row: negative index -> last step
            if (last == t) {
                q32[(size_t)ob * ld_q + unit] = hn;
                qT[(size_t)ob * ld_q + unit] = hb;
                if (q_drop)                                          // dropout of the stored (rounded) question vector
                    q_drop[(size_t)ob * ld_q + unit] =
                        f32_to_bf16(bf16_to_f32(hb) * drop_mult(seed, (unsigned)(ob * H + unit), thr, dscale));
            }
        }
}

}  // namespace

extern "C" int vqa_lstm_step(const void* h_prev, int ld_h, const void* w_hh, int ld_w, const void* xg, int ld_xg,
                             int n_table_rows, const int64_t* tokens, float* c, void* h_out, int ld_ho,
                             const int32_t* lengths, float* q_f32, void* q_t, int ld_q, int B, int H, int t, int L,
                             const int32_t* perm, const int32_t* active, const int32_t* offsets, void* hs, int ld_hs,
                             void* q_drop, unsigned drop_seed, float drop_p, int dtype, hipStream_t stream) {
    VQA_REQUIRE(dtype == VQA_BF16, "vqa_lstm_step: bf16 storage only (fp32 runs vqa_gemm_slabs + vqa_lstm_cell_packed)");
    VQA_REQUIRE(h_prev && w_hh && xg && c && h_out && lengths && q_f32 && q_t && perm && active && offsets,
                "vqa_lstm_step: null pointer");
    VQA_REQUIRE(h_prev != h_out, "vqa_lstm_step: h_out must not alias h_prev (other workgroups still read it)");
    VQA_REQUIRE(!tokens || n_table_rows > 0, "vqa_lstm_step: tokens given without a table");
    VQA_REQUIRE(B >= 0 && H > 0 && t >= 1 && t < L, "vqa_lstm_step: bad sizes (the step needs a previous output: t >= 1)");
    VQA_REQUIRE(H % 16 == 0, "vqa_lstm_step: the hidden width must be a multiple of 16 (16 units x 4 gates per workgroup)");
    VQA_REQUIRE(ld_h >= H && ld_w >= H && ld_ho >= H && ld_q >= H && ld_xg >= 4 * H && ld_h % 8 == 0 && ld_w % 8 == 0,
                "vqa_lstm_step: leading dimensions (h and W_hh rows must be multiples of 8 elements)");
    VQA_REQUIRE(!hs || ld_hs >= H, "vqa_lstm_step: bad ld_hs");
    VQA_REQUIRE(((((uintptr_t)h_prev) | ((uintptr_t)w_hh)) & 15) == 0, "vqa_lstm_step: operands must be 16-byte aligned");
    VQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "vqa_lstm_step: drop_p must be in [0, 1)");
    if (B == 0) return 0;
    const unsigned grid = (unsigned)((H / 16) * ((B + LS_BM - 1) / LS_BM));
    k_lstm_step<<<grid, 256, 0, stream>>>((const bf16_t*)h_prev, ld_h, (const bf16_t*)w_hh, ld_w, (const bf16_t*)xg, ld_xg,
                                          n_table_rows, tokens, c, (bf16_t*)h_out, ld_ho, lengths, q_f32, (bf16_t*)q_t, ld_q,
                                          H, t, L, perm, active, offsets, (bf16_t*)hs, ld_hs, (bf16_t*)q_drop, drop_seed,
                                          drop_threshold(drop_p), 1.0f / (1.0f - drop_p));
    return vqa_check_launch("vqa_lstm_step");
}
