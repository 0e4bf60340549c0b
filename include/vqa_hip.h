/* libvqa_hip.so -- C ABI of the MI355X (gfx950) kernels behind the VQA
 * attention / fusion / classifier train step.
 *
 * The reference (SanyamSaxena/vqa) is pure PyTorch and has no FFI; every entry
 * point below replaces the stock-PyTorch op sequence cited beside it (paths
 * relative to the reference checkout).  The host binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here
 *     allocates, frees or synchronises; work is enqueued on `stream`;
 *   - return 0 on success, a negative code otherwise (-22: bad argument, the
 *     text is available from vqa_last_error_string(); <= -1000: HIP launch
 *     error -(hipError_t) - 1000);
 *   - dtype: VQA_F32 (0) or VQA_BF16 (1) is the STORAGE type of activations and
 *     weights; arithmetic is fp32 (bf16 MFMA products accumulate in fp32);
 *   - matrices are row-major with an explicit leading dimension (elements).
 *     Feature dimensions and leading dimensions of `dtype` matrices are
 *     multiples of 8 elements (16-byte chunks); callers zero-pad.  A GEMM
 *     operand stored with k contiguous whose K is not a multiple of 8 must be
 *     zero-filled up to the next multiple of 8 within its leading dimension.
 *   - dropout is a counter-based hash of (seed, logical element index):
 *     keep iff murmur3_fmix32(index * 0x9E3779B1 + seed) >= p * 2^32, kept
 *     values scaled by 1/(1-p); p == 0 disables it.
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* 200: vqa_sequence_pack (+n_table_rows, status), vqa_softmax_ce (+status) and vqa_adam_step (+status) take more
 * arguments than in version 100, and vqa_gemm_slabs_group takes up to 8 products: code built against the version-100
 * header must be rebuilt (vqa_version() tells which library is loaded).  210 adds entry points only.  220 adds
 * vqa_lstm_resident, a fourth status word (VQA_STATUS_WORDS 3 -> 4: callers allocate the longer array) and the
 * h_chunk_rows argument of vqa_lstm_cell01_table. */
#define VQA_HIP_VERSION 220
#define VQA_F32 0
#define VQA_BF16 1

/* Device-side status words (int32_t status[VQA_STATUS_WORDS], zeroed by the caller): kernels that meet an input the
 * reference would raise for set a word to 1 and carry on (nothing here synchronises); the host reads the words when
 * it next synchronises anyway. */
#define VQA_STATUS_BAD_TOKEN 0     /* a question token id outside the embedding table (nn.Embedding raises IndexError) */
#define VQA_STATUS_BAD_TARGET 1    /* an answer id outside [0, C) (CrossEntropyLoss raises "Target out of bounds") */
#define VQA_STATUS_SKIPPED_STEP 2  /* vqa_adam_step: the `step` of the first update it skipped because BAD_TARGET was set */
#define VQA_STATUS_RESIDENT_TIMEOUT 3 /* vqa_lstm_resident: its grid barrier did not complete; the outputs are not valid */
#define VQA_STATUS_WORDS 4

int vqa_version(void);
const char* vqa_last_error_string(void);

/* ---- question encoder ---------------------------------------------------- */

/* len[b] = L - #(tokens[b,:] == 0).                seq2vec.py:11-14 (process_lengths) */
int vqa_process_lengths(const int64_t* tokens, int32_t* lengths, int B, int L, hipStream_t stream);

/* out[row(b,t), 0:width] = table[tokens[b,t], 0:width]; row = t*B+b when time_major
 * else b*L+t.  Ids outside [0, n_rows) read row 0 (the pad row).
 *                                                   seq2vec.py:35-37, 42 (nn.Embedding) */
int vqa_embedding_gather(const int64_t* tokens, const void* table, void* out, int B, int L, int width,
                         int ld_table, int ld_out, int n_rows, int time_major, int dtype, hipStream_t stream);

/* One LSTM step after the recurrent product: gates = sum_s slabs[s] + xg_t (xg_t holds
 * x_t W_ih^T + b_ih + b_hh), gate order i,f,g,o; updates c (fp32) and h; rows whose last
 * valid token is step t (len-1, or L-1 for an all-pad row) also write q.
 *                                  seq2vec.py:38, 43 (nn.LSTM) and 16-25 (select_last) */
int vqa_lstm_cell(const float* slabs, int nslab, size_t slab_stride, const void* xg_t, int ld_xg, float* c,
                  void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                  int B, int H, int t, int L, int dtype, hipStream_t stream);

/* Packed sequences (rows sorted by length so that the rows still running at step t are a prefix):
 *   lengths as vqa_process_lengths; steps[b] = lengths[b], or L for an all-pad row (seq2vec.py:21: index
 *   -1 selects the last step); perm[j] = original row of sorted row j (descending steps, ties in row
 *   order); active[t] = #rows with steps > t; offsets[t] = sum of active[0:t], offsets[L] = their total.
 *   One workgroup; B + 18 (L + 1) <= 16384.                        seq2vec.py:11-14, 16-25 */
int vqa_sequence_pack(const int64_t* tokens, int B, int L, int32_t* lengths, int32_t* perm, int32_t* active,
                      int32_t* offsets, long n_table_rows, int32_t* status, hipStream_t stream);
/* (status, optional, with n_table_rows > 0: status[VQA_STATUS_BAD_TOKEN] = 1 if any token id lies outside
 *  [0, n_table_rows); the gather kernels read the pad row for such an id) */
/* out[offsets[t] + j, 0:width] = act(table[tokens[perm[j], t], 0:width]) for j < active[t]; act 0 none,
 * 1 tanh (TwoLSTM applies tanh to the embeddings)                  seq2vec.py:42, 63-64 */
int vqa_embedding_gather_packed(const int64_t* tokens, const void* table, void* out, int B, int L, int width,
                                int ld_table, int ld_out, int n_rows, const int32_t* perm, const int32_t* offsets,
                                int act, int32_t* packed_tokens, int dtype, hipStream_t stream);
/* (packed_tokens, optional: packed_tokens[offsets[t] + j] = the token id gathered into that row) */
/* vqa_lstm_cell on packed sequences: state rows c, h and the slabs are in sorted order, only rows
 * j < active[t] are updated, the input projection of row j is row offsets[t] + j of xg, and the
 * selected output goes to q[perm[j]].  hs (optional) receives every step's output at the packed
 * row offsets[t] + j: the input sequence of a stacked second layer (seq2vec.py:65, 70). */
int vqa_lstm_cell_packed(const float* slabs, int nslab, size_t slab_stride, const void* xg, int ld_xg, float* c,
                         void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                         int B, int H, int t, int L, const int32_t* perm, const int32_t* active,
                         const int32_t* offsets, void* hs, int ld_hs, void* gates_out, float* c_out, void* q_drop,
                         unsigned drop_seed, float drop_p, int dtype, hipStream_t stream);
/* (q_drop, optional, leading dimension ld_q: dropout(q_t) with the mask of vqa_dropout on a (B, H) matrix --
 *  dropoutQ / dropoutV of model.py:67, 151 fused into the producer of the question vector) */
/* (gates_out (packed rows x 4H, activated i, f, g, o) and c_out (packed rows x H, fp32), optional and
 *  together: the state saved for backward through time) */

/* The same step with the input projections gathered by token id from a table,
 *   xg_table[v] = act(embedding[v]) W_ih^T + b_ih + b_hh   (n_table_rows x 4H),
 * which the caller builds once per weight version when the embedding and W_ih are frozen (the reference freezes
 * the whole encoder, model.py:42-43): the embedding gather and the input projection of every step collapse into
 * this gather; row j of step t reads xg_table[tokens[perm[j], t]] (ids outside the table read row 0, the pad
 * row).  No state is saved for backward through time.              seq2vec.py:42-43, 63-66 */
int vqa_lstm_cell_packed_table(const float* slabs, int nslab, size_t slab_stride, const void* xg_table, int ld_xg,
                               int n_table_rows, const int64_t* tokens, float* c, void* h, int ld_h,
                               const int32_t* lengths, float* q_f32, void* q_t, int ld_q, int B, int H, int t, int L,
                               const int32_t* perm, const int32_t* active, const int32_t* offsets, void* hs, int ld_hs,
                               void* q_drop, unsigned drop_seed, float drop_p, const float* hh_table, int dtype,
                               hipStream_t stream);
/* (hh_table, optional, fp32 n_table_rows x 4H, only with nslab = 0 and t >= 1: the recurrent product tabulated per
 *  previous token id, hh_table[v] = h_0(v) W_hh^T with h_0(v) the first step's output for token v; the caller passes it
 *  at t = 1, the one step whose recurrent input depends on a single token) */

/* Steps 0 and 1 of the table form in one launch (neither needs a recurrent product once hh_table exists): the
 * outputs of vqa_lstm_cell_packed_table at t = 0 followed by t = 1 with hh_table.  L >= 2.
 * h_chunk_rows: 0, or (with H % 8 == 0) the row count of a chunk-major h, element (row, k) at
 * ((k / 8) * h_chunk_rows + row) * 8 + k % 8 -- the layout of vqa_lstm_resident's ring (ld_h is then unused). */
int vqa_lstm_cell01_table(const void* xg_table, int ld_xg, const float* hh_table, int n_table_rows, const int64_t* tokens,
                          float* c, void* h, int ld_h, const int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                          int B, int H, int L, const int32_t* perm, const int32_t* active, const int32_t* offsets,
                          void* hs, int ld_hs, void* q_drop, unsigned drop_seed, float drop_p, int h_chunk_rows, int dtype,
                          hipStream_t stream);

/* Steps t_first .. L-1 of the table form in ONE launch, with W_hh resident in registers (no counterpart in the
 * reference, which calls nn.LSTM, seq2vec.py:38, 43; the arithmetic is that of vqa_lstm_cell_packed_table after a
 * recurrent product, with the four K quarters summed in a fixed order).  Workgroup g keeps the 4 x
 * VQA_LSTM_RESIDENT_UNITS rows of W_hh (4H x H, row-major, gate order i,f,g,o, bf16) that belong to hidden units
 * g * UNITS .., multiplies them with the live rows of h, updates its columns of the state and meets the other
 * workgroups at a grid barrier before the next step.  bf16 storage only.
 *   t_first = 0 the WHOLE encoder in this launch: every workgroup packs the sequences for itself (the arithmetic of
 *               vqa_sequence_pack: seq2vec.py:11-14, 16-25; a token id outside the table sets
 *               status[VQA_STATUS_BAD_TOKEN]) and runs steps 0 and 1 of its units from the two tables (the arithmetic
 *               of vqa_lstm_cell01_table); needs hh_table and L >= 2; c is not read; lengths / perm / active are
 *               optional OUTPUTS (what vqa_sequence_pack would have written; offsets are not produced)
 *   t_first >= 1 the caller ran vqa_sequence_pack and the steps before t_first: lengths, perm, active and c are inputs
 *   xg_table_u  (n_table_rows, H, 4) bf16: the input-projection table of vqa_lstm_cell_packed_table with the four
 *               gates of a unit adjacent, xg_table_u[v][u][g] = xg_table[v][g * H + u]
 *   hh_table    (n_table_rows, 4H) fp32, as vqa_lstm_cell_packed_table's (t_first = 0 only)
 *   c           (B, H) fp32, sorted row order: cell state after step t_first - 1 (read only; the state of later steps
 *               stays on the chip)
 *   h_ring      2 slots of (VQA_LSTM_RESIDENT_PITCH / 8, Bp, 8) bf16 with Bp = VQA_LSTM_RESIDENT_RING_ROWS(B) (B rounded
 *               up to whole 16-row fragments plus one spare fragment), chunk-major: element (row, k) of a slot at
 *               ((k / 8) * Bp + row) * 8 + k % 8, so that the 16 rows of an MFMA fragment are contiguous per 8-element
 *               chunk of k.  With t_first >= 1 slot 0 holds h after step t_first - 1 in sorted row order
 *               (vqa_lstm_cell01_table with h_chunk_rows = Bp writes it); the elements with k >= H of BOTH slots
 *               must be zero (they are never written); the slots alternate
 *   sync_words  VQA_LSTM_RESIDENT_SYNC_WORDS 64-bit words, 128-byte aligned, zeroed by the caller ONCE (and again
 *               after a timeout); launches that share them must not overlap
 *   status      optional: status[VQA_STATUS_RESIDENT_TIMEOUT] = 1 if a barrier timed out (10 s), e.g. because the
 *               workgroups could not all be resident
 * Needs H % 8 == 0, H <= VQA_LSTM_RESIDENT_PITCH, B <= VQA_LSTM_RESIDENT_MAX_ROWS, L <= VQA_LSTM_RESIDENT_MAX_STEPS,
 * ld_q % 4 == 0, q_f32 / tables 16-byte and q_t / q_drop 8-byte aligned, and at least
 * vqa_lstm_resident_workgroups(H) compute units, all free to take one workgroup each. */
#define VQA_LSTM_RESIDENT_UNITS 12
#define VQA_LSTM_RESIDENT_PITCH 2432
#define VQA_LSTM_RESIDENT_MAX_ROWS 512
#define VQA_LSTM_RESIDENT_MAX_STEPS 64
#define VQA_LSTM_RESIDENT_SYNC_WORDS 176
#define VQA_LSTM_RESIDENT_RING_ROWS(B) ((((B) + 15) & ~15) + 16)
int vqa_lstm_resident_workgroups(int H);      /* 0: H not supported */
int vqa_lstm_resident(const void* w_hh, int ld_w, const void* xg_table_u, const float* hh_table, int n_table_rows,
                      const int64_t* tokens, const float* c, void* h_ring, int32_t* lengths, float* q_f32, void* q_t, int ld_q,
                      int B, int H, int t_first, int L, int32_t* perm, int32_t* active, void* q_drop,
                      unsigned drop_seed, float drop_p, uint64_t* sync_words, int32_t* status, hipStream_t stream);

/* Backward through LSTM step t on packed sequences (a trainable question encoder; the reference
 * freezes it, model.py:42-43, so this has no counterpart there beyond autograd of seq2vec.py:43):
 * slabs = dG_{t+1} W_hh partial sums (rows running at t+1), dq (B, ld_dq) in original row order is the
 * gradient of the selected outputs, dh_ext (optional, packed rows x H fp32) a gradient arriving at
 * every step's output; writes dG and hprev (h_{t-1}) at packed row offsets[t] + j, dG also at row j of
 * dG_state (B x 4H, the operand of the next recurrent product), updates dc_state. */
int vqa_lstm_cell_bwd_packed(const float* slabs, int nslab, size_t slab_stride, const float* dq, int ld_dq,
                             const float* dh_ext, const void* gates, const float* cs, const void* hs, float* dc_state,
                             void* dG, void* dG_state, void* hprev, const int32_t* lengths, const int32_t* perm, const int32_t* active,
                             const int32_t* offsets, int B, int H, int t, int L, int dtype, hipStream_t stream);
/* dE[v, :] = sum over packed rows p with packed_tokens[p] == v, in ascending p, of (sum_s slabs[s][p, :]) *
 * (1 - x_act[p, :]^2 if x_act); dE[0, :] = 0 (padding_idx, seq2vec.py:35-37).  No float atomics. */
int vqa_embedding_grad(const float* slabs, int nslab, size_t slab_stride, int ld_dx, const int32_t* packed_tokens,
                       const int32_t* total_rows, const void* x_act, int ld_x, float* dE, int V, int E, int dtype,
                       hipStream_t stream);

/* out[b, col0:col0+Ec] = l2norm(sum_t table[cap[b,t]] * (cap != 0) / max(#nonzero, 1e-9))
 *                      model.py:1571-1574 (mean_pooling), 1678 (F.normalize), 1734 (cat) */
int vqa_caption_pool(const int64_t* cap_tokens, const void* table, void* out, int B, int Tc, int Ec,
                     int ld_table, int n_rows, int ld_out, int col0, int dtype, hipStream_t stream);

/* ---- image side ------------------------------------------------------------ */

/* inv_norm[row] = 1 / max(||V[row,:]||_2, 1e-12)     model.py:1678 (F.normalize semantics) */
int vqa_inv_norm(const void* V, float* inv_norm, long rows, int Dv, int ldV, int dtype, hipStream_t stream);

/* Fused attention over the region grid, one workgroup per sample:
 *   s_r = w_a . tanh(P[b,r,:] + qa[b,:]) + c;  p = softmax_r(s);
 *   pooled[b,:] = sum_r p_r inv_norm[b,r] V[b,r,:].
 * No reference counterpart (the reference pools with torch.mean, model.py:161-164,
 * which this reduces to for w_a = 0). */
size_t vqa_attn_lds_bytes(int R, int A, int Dv);
int vqa_attn_fwd(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                 const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                 int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                 float drop_p, hipStream_t stream);
/* (pooled_drop, optional: dropout(pooled_t) with the mask of vqa_dropout on a (B, Dv) matrix -- dropoutV of
 *  model.py:62 fused into the producer) */
/* Backward of the above: dP is written pre-multiplied by inv_norm (so dW_va = dP^T V),
 * dqa[b,:] = sum_r dP_unscaled, dwa_part[b,:] and dc_part[b] are per-sample partials. */
int vqa_attn_bwd(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                 const float* inv_norm, const float* p_in, const float* d_pooled, int ld_dpool, void* dP, int ld_dP,
                 float* dqa_f32, void* dqa_t, int ld_dqa, float* dwa_part, float* dc_part, int B, int R, int A, int Dv,
                 int dtype, hipStream_t stream);
/* The same two kernels fed with split-K partial sums instead of a reduced operand (a launch less each):
 * forward: qa = sum_s qa_slabs[s] + qa_bias is formed by the kernel's prologue and stored to qa (B, A);
 * backward: d_pooled = (sum_s dpool_slabs[s]) * dropout mask of (drop_seed, drop_p) on a (B, Dv) matrix is formed
 * by the prologue (d_pooled is scratch).  Shapes outside the bf16 streaming kernels reduce with vqa_epilogue first.
 * No reference counterpart, as above. */
int vqa_attn_fwd_qslabs(const void* P, int ldP, const void* V, int ldV, const float* qa_slabs, int qa_nslab,
                        size_t qa_stride, const float* qa_bias, float* qa, int ld_qa, const float* w_a,
                        const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                        int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                        float drop_p, hipStream_t stream);
int vqa_attn_bwd_dslabs(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                        const float* inv_norm, const float* p_in, const float* dpool_slabs, int dpool_nslab,
                        size_t dpool_stride, unsigned drop_seed, float drop_p, float* d_pooled, int ld_dpool, void* dP,
                        int ld_dP, float* dqa_f32, void* dqa_t, int ld_dqa, float* dwa_part, float* dc_part, int B, int R,
                        int A, int Dv, int dtype, hipStream_t stream);

/* Small batches.  The two kernels above run one workgroup per sample; with fewer samples than CUs (the reference's own
 * batch sizes: train.py:242 bs 40, test.py:120 bs 70) most of the chip idles.  These forms share a sample's region
 * rows among nsplit workgroups (bf16 storage, the shapes of the streaming kernels):
 *   forward: every row range keeps a running maximum of its own; a second small launch merges the ranges of a sample in
 *            index order (same p and pooled up to fp32 rounding of the exponent shifts);
 *   backward: two launches (the softmax backward needs sum_r p_r dp_r over all rows of the sample in between); the
 *            per-range sums of dqa come back as nsplit slabs dqa_slabs[range][b][0:A] (reduce them with vqa_epilogue),
 *            dwa_part has B * nsplit rows and dc_part B * nsplit entries (their consumers sum over rows anyway).
 * vqa_attn_splits: the nsplit that fills the chip once (1: use the plain entry points); workspace: fp32 scratch of
 * vqa_attn_split_workspace_floats(B, R, Dv, nsplit) elements shared by the forward and the backward of a step.
 * No reference counterpart, as above. */
int vqa_attn_splits(int B, int R, int A, int Dv, int dtype);
size_t vqa_attn_split_workspace_floats(int B, int R, int Dv, int nsplit);
int vqa_attn_fwd_split(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                       const float* c_a, const float* inv_norm, float* p_out, void* pooled_t, float* pooled_f32,
                       int ld_pool, int B, int R, int A, int Dv, int dtype, void* pooled_drop, unsigned drop_seed,
                       float drop_p, int nsplit, float* workspace, hipStream_t stream);
int vqa_attn_bwd_split(const void* P, int ldP, const void* V, int ldV, const float* qa, int ld_qa, const float* w_a,
                       const float* inv_norm, const float* p_in, const float* d_pooled, int ld_dpool, void* dP, int ld_dP,
                       float* dqa_slabs, int ld_dqa, float* dwa_part, float* dc_part, int B, int R, int A, int Dv,
                       int dtype, int nsplit, float* workspace, hipStream_t stream);

/* ---- feature-map front-ends of the reference's ResNet models (precomputed maps) ------------- */

/* out[b, col0 + c] = mean over the H*W grid of channel c      model.py:161-164 (torch.mean(x, axis=(2,3)))
 * layout 0: x is (B, C, HW) (an NCHW map); layout 1: x is (B, HW, C), C % 8 == 0.
 * In layout 1 the grid may be split into nsplit row ranges (vqa_grid_mean_splits tells how many keep
 * the chip busy); partial then holds nsplit * B * C floats.  The summation order is fixed. */
int vqa_grid_mean_splits(int layout, int B, int C, int HW);
int vqa_grid_mean(const void* x, int layout, int B, int C, int HW, float* out, int ld_out, int col0,
                  float* partial, int nsplit, int dtype, hipStream_t stream);

/* vqa_grid_mean for callers that cannot allocate device memory or pass a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The split count is
 * vqa_grid_mean_splits(layout, B, C, HW) and the partial sums live in a static device array of
 * VQA_GRID_MEAN_STATIC_FLOATS floats (1 MiB, held by every GPU the library runs on until the process ends);
 * nsplit * B * C beyond that returns -22.  Same results as vqa_grid_mean, bit for bit.  The call allocates and
 * frees nothing; the first call on a device looks the array's address up (hipGetSymbolAddress).  All calls on
 * one GPU share the array: calls that overlap on one GPU (two streams, two threads) race. */
#define VQA_GRID_MEAN_STATIC_FLOATS (1 << 18)
int vqa_grid_mean_static(const void* x, int layout, int B, int C, int HW, float* out, int ld_out, int col0,
                         int dtype, hipStream_t stream);

/* gate[b,n] = sigmoid(sum_s slabs[s][b][n] + bias[n]);  out[b,n] = dropout(mean[b,n] * gate[b,n])
 *                     model.py:125-140 (gapK = Linear + Sigmoid), 166-172 (product, cat, dropoutV) */
int vqa_gate_fwd(const float* slabs, int nslab, size_t slab_stride, const float* bias, const float* mean, int ld_mean,
                 float* gate, int ld_gate, void* out, int ld_out, int B, int N, unsigned seed, float p, int dtype,
                 hipStream_t stream);
/* dz[b,n] = d_out[b,n] * mean[b,n] * gate (1 - gate): gradient at the gate's pre-activation */
int vqa_gate_bwd(const float* d_out, int ld_d, const float* mean, int ld_mean, const float* gate, int ld_gate,
                 void* dz, int ld_dz, int B, int N, int dtype, hipStream_t stream);

/* Weighted stack of K gated vectors sig (B, K, D) fp32, then max and mean over the stack, concatenated and
 * dropped: out[b, 0:D] = max_k w[k] sig[b,k,:], out[b, D:2D] = mean_k w[k] sig[b,k,:]; argmax[b, d] = the k of
 * the maximum (first on ties); weight k is w[k * ld_w].        model.py:1238-1241 (stack, mean, max, cat) */
int vqa_stack_maxavg_fwd(const float* sig, const float* w, int ld_w, void* out, int ld_out, unsigned char* argmax, int B, int D,
                         int K, unsigned seed, float p, int dtype, hipStream_t stream);
/* backward of the above: dz (B, K, D) = gradient at the pre-activation of each gate's sigmoid,
 * dw_part (B, K) = per-sample partial gradients of the stack weights (summed over b by vqa_colsum) */
int vqa_stack_maxavg_bwd(const float* d_out, int ld_d, const float* sig, const float* w, int ld_w, const unsigned char* argmax,
                         void* dz, float* dw_part, int B, int D, int K, int dtype, hipStream_t stream);

/* out[b, ch * R + r] = dropout(y[b * R + r, ch]): the NCHW .view(-1, VISUAL_OUT) of the 1x1-convolved
 * map followed by dropoutV                                              model.py:54, 61-62 */
int vqa_flatten_cm(const void* y, int ld_y, void* out, int ld_out, int B, int R, int c, unsigned seed, float p,
                   int dtype, hipStream_t stream);
/* dy[b * R + r, ch] = d_out[b, ch * R + r] for ch < c, 0 for c <= ch < ld_dy (backward of the above) */
int vqa_unflatten_cm(const float* d_out, int ld_d, void* dy, int ld_dy, int B, int R, int c, int dtype,
                     hipStream_t stream);

/* ---- GEMM:  C[M,N] = sum_k A(m,k) B(n,k) ------------------------------------
 * a_kmajor / b_kmajor: 1 when the operand is stored with k contiguous ([M][K], [N][K] as an
 * nn.Linear weight), 0 when stored [K][M] / [K][N].  engine: 0 = MFMA for bf16 (fp32 always
 * runs the VALU kernel), 1 = force the VALU kernel (cross-check).
 *                   nn.Linear forward / backward: model.py:44, 55-58, 63, 68, 74, 77 */
int vqa_gemm_splitk_hint(int M, int N, int K, int a_kmajor, int b_kmajor);
/* fp32 partial sums: slabs[s][m][n], s < splitk (consumers sum the slabs) */
int vqa_gemm_slabs(const void* A, int lda, int a_kmajor, const void* B, int ldb, int b_kmajor, float* slabs,
                   size_t slab_stride, int ld_slab, int M, int N, int K, int splitk, int dtype, int engine,
                   hipStream_t stream);
/* Up to 8 independent products of the same operand layout in one launch (each product of the head is a few
 * microseconds of work and a dependent launch costs about as much): slabs as above, 64 x 64 tiles for bf16;
 * fp32 runs them one after the other. */
typedef struct {
    const void* A; int lda; const void* B; int ldb;
    float* slabs; size_t slab_stride; int ld_slab;
    int M, N, K, splitk;
} VqaGemmDesc;
int vqa_gemm_slabs_group(const VqaGemmDesc* descs, int n, int a_kmajor, int b_kmajor, int dtype, hipStream_t stream);
/* The same two products restricted to the first *active_rows rows of the output (a device-side
 * count read by the kernel: workgroups whose first row is past it exit; rows past it inside a
 * started tile may or may not be written) and, for the slab form, to the first *active_k terms of
 * the contraction (exact: later k contribute nothing).  Either pointer may be NULL. */
int vqa_gemm_slabs_rows(const void* A, int lda, int a_kmajor, const void* B, int ldb, int b_kmajor, float* slabs,
                        size_t slab_stride, int ld_slab, int M, int N, int K, int splitk, int dtype, int engine,
                        const int32_t* active_rows, const int32_t* active_k, hipStream_t stream);
int vqa_gemm_fused_rows(const void* A, int lda, int a_kmajor, const void* B, int ldb, int b_kmajor, void* C, int ldc,
                        int M, int N, int K, const float* bias, const float* row_scale, int act, int out_dtype,
                        int dtype, int engine, const int32_t* active_rows, hipStream_t stream);
/* C = act(row_scale[m] * (A B^T) + bias[n]); act 0 none, 1 tanh; bias / row_scale may be NULL */
int vqa_gemm_fused(const void* A, int lda, int a_kmajor, const void* B, int ldb, int b_kmajor, void* C, int ldc,
                   int M, int N, int K, const float* bias, const float* row_scale, int act, int out_dtype,
                   int dtype, int engine, hipStream_t stream);

/* C = act(inv_norm[m] * (A B^T) + bias[n]) with inv_norm[m] = 1 / max(||A[m,:]||_2, 1e-12) computed by the
 * same launch and also stored to inv_norm (M floats): F.normalize(v, dim=-1) folded into the projection that
 * consumes it.  A [M][K] and B [N][K] are k-major.  With bf16 operands and a product large enough for the
 * 256 x 256 ring kernel the row norms come from the A tiles already in LDS (A is read once); otherwise the
 * call runs vqa_inv_norm followed by vqa_gemm_fused.  No reference counterpart (the reference has no
 * normalised projection: model.py:161-164 pools with a mean). */
int vqa_gemm_fused_rownorm(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                           const float* bias, float* inv_norm, int act, int out_dtype, int dtype, int engine,
                           hipStream_t stream);

/* ---- split-K reduce + epilogues of the head ------------------------------------ */

/* v = act(row_scale[m] * sum_s slabs[s][m][n] + bias[n] + addend[m][n]), act 0 none, 1 tanh, 2 sigmoid;
 * out_pre (fp32) = v; out_post (post_dtype) = dropout(v).  Either output may be NULL. */
int vqa_epilogue(const float* slabs, int nslab, size_t slab_stride, int M, int N, const float* bias,
                 const float* row_scale, const float* addend, int ld_add, int act, float* out_pre, int ld_pre,
                 void* out_post, int ld_post, int post_dtype, unsigned drop_seed, float drop_p, hipStream_t stream);

/* y = dropout(x)                                        model.py:38-40, 62, 67 (nn.Dropout) */
int vqa_dropout(const void* x, int ldx, void* y, int ldy, int M, int N, unsigned seed, float p, int dtype,
                hipStream_t stream);

/* x = dropout(tanh(xv * xq))                                              model.py:71-73 */
int vqa_fusion_fwd(const float* xv, const float* xq, void* x, int ldx, int B, int F, unsigned seed, float p,
                   int dtype, hipStream_t stream);
/* The same fused into the reduction of linear_q's split-K partial sums: xq = tanh(sum_s slabs + bias) is stored
 * (B, F) fp32 and x = dropout(tanh(xv * xq)); one launch instead of vqa_epilogue + vqa_fusion_fwd, same values
 * and the same dropout mask.                                                  model.py:68-73 */
int vqa_epilogue_fusion(const float* slabs, int nslab, size_t slab_stride, int B, int F, const float* bias,
                        const float* xv, float* xq, void* x, int ldx, unsigned seed, float p, int dtype,
                        hipStream_t stream);
/* Both branches still in split-K partial sums: xv = tanh(sum sv + bias_v), xq = tanh(sum sq + bias_q) are stored
 * (B, F) fp32 and x = dropout(tanh(xv * xq)): the two reductions and the fusion in one launch.   model.py:63-73 */
int vqa_fusion_fwd_slabs(const float* slabs_v, int nslab_v, size_t stride_v, const float* bias_v,
                         const float* slabs_q, int nslab_q, size_t stride_q, const float* bias_q,
                         float* xv, float* xq, void* x, int ldx, int B, int F, unsigned seed, float p, int dtype,
                         hipStream_t stream);
/* backward of the above down to the pre-activations of linear_v / linear_q (xv, xq are
 * the stored tanh outputs): dzv = dx drop (1-t^2) xq (1-xv^2), dzq likewise */
int vqa_fusion_bwd(const float* slabs, int nslab, size_t slab_stride, int ldslab, const float* xv, const float* xq,
                   void* dzv, void* dzq, int ldz, int B, int F, unsigned seed, float p, int dtype, hipStream_t stream);
/* backward through y = dropout(tanh(z)):  dz = (sum_s slabs) * drop * (1 - t^2), t = tanh(z)
 *                                                                         model.py:75-76 */
int vqa_tanh_dropout_bwd(const float* slabs, int nslab, size_t slab_stride, const float* t_pre, int ld_t, void* dz,
                         int ldz, int M, int N, unsigned seed, float p, int dtype, hipStream_t stream);

/* out[n] = sum_m X[m,n] (bias gradients), fixed summation order */
int vqa_colsum(const void* X, int ld, float* out, int M, int N, int dtype, hipStream_t stream);
/* up to VQA_SLABSUM_MAX split-K reductions in one launch: out[e] = sum_s slabs[s * slab_stride + e], e < count
 * (the weight gradients of a bucket, reduced together once their products are done) */
#define VQA_SLABSUM_MAX 8
typedef struct VqaSlabSumDesc { const float* slabs; int nslab; size_t slab_stride; long count; float* out; } VqaSlabSumDesc;
int vqa_slab_sum_multi(const VqaSlabSumDesc* descs, int n, hipStream_t stream);
/* out[n][m] = sum_s slabs[s][m][n] (slabs of M x N floats, out rows of ld_out >= M floats): the reduction of a
 * weight gradient whose product ran with the operand roles swapped -- dW^T = x^T dy is the faster orientation
 * on MI355X when dy has the narrower, unaligned rows -- back into the nn.Linear layout.   autograd of model.py:55-58 */
int vqa_slab_sum_transpose(const float* slabs, int nslab, size_t slab_stride, int M, int N, float* out, int ld_out,
                           hipStream_t stream);
/* up to VQA_COLSUM_MAX such sums in one launch (the bias gradients of a backward pass); out2 (optional)
 * receives the same sums (two biases added to the same pre-activation share their gradient); every sum is
 * multiplied by scale (1 for a gradient, 1/B for the mean of the per-row losses) */
#define VQA_COLSUM_MAX 8
typedef struct VqaColsumDesc { const void* X; int ld; int M; int N; int dtype; float scale; float* out; float* out2; } VqaColsumDesc;
int vqa_colsum_multi(const VqaColsumDesc* descs, int n, hipStream_t stream);
/* vqa_gemm_slabs_group with a batch of column sums (vqa_colsum_multi's problems: the bias gradients that go with the
 * weight gradients) appended as extra workgroups. */
int vqa_gemm_slabs_group_colsum(const VqaGemmDesc* descs, int n, int a_kmajor, int b_kmajor, int dtype,
                                const VqaColsumDesc* colsums, int n_colsums, hipStream_t stream);
/* the same over the first *rows rows (a device-side count) */
int vqa_colsum_rows(const void* X, int ld, float* out, const int32_t* rows, int N, int dtype, hipStream_t stream);
/* dst[m,0:N] = (dtype) src[m,0:N]; dst[m,N:ld_dst] = 0 */
int vqa_cast_pad(const float* src, int ld_src, void* dst, int ld_dst, int M, int N, int dtype, hipStream_t stream);

/* ---- loss, prediction, optimiser ------------------------------------------------- */

/* loss_rows[b] = logsumexp(z_b) - z_b[y_b]; dlogits = (softmax(z_b) - onehot) * grad_scale
 * (columns C..ld_d-1 of dlogits are zeroed); dlogits may be NULL.
 *                             train.py:35, 201 (CrossEntropyLoss) and 203 (backward) */
int vqa_softmax_ce(const float* logits, int ld, const int64_t* target, float* loss_rows, void* dlogits, int ld_d,
                   int B, int C, float grad_scale, int dtype, int32_t* status, hipStream_t stream);
/* (a target outside [0, C): that row's loss is NaN, its gradient zero and, with status given,
 *  status[VQA_STATUS_BAD_TARGET] = 1, which makes vqa_adam_step skip its update: CrossEntropyLoss raises for such a
 *  target before optimizer.step(), train.py:201-204.  There is no ignore_index: -100 is out of range like any other
 *  negative id; the reference's answer ids come from VocabEncoder and are never negative) */
/* out[0] = scale * sum(x[0:n]) in a fixed order (mean reduction of the loss) */
int vqa_sum_scale(const float* x, float* out, int n, float scale, hipStream_t stream);

/* pred[b] = first index of the row maximum (a NaN counts as the maximum, as np.argmax);
 * when answers, type_ids and counters are given: counters[t]++ and
 * counters[n_types + t] += (answers[b] == pred[b]).          train.py:105-110, test.py:85-90 */
int vqa_argmax_accuracy(const float* logits, int ld, int B, int C, const int64_t* answers, const int32_t* type_ids,
                        int n_types, int64_t* pred, uint64_t* counters, hipStream_t stream);

/* Adam on flat fp32 buffers (torch.optim.Adam defaults: no weight decay, no amsgrad);
 * grad is multiplied by grad_scale first; shadow_bf16 (optional) receives bf16(param).  status (optional, the
 * words vqa_softmax_ce writes): while status[VQA_STATUS_BAD_TARGET] is set the call changes nothing and
 * status[VQA_STATUS_SKIPPED_STEP] keeps the `step` of the first call skipped.
 *                                                                   train.py:34, 204 */
int vqa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, int32_t* status,
                  hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQA_HIP_H */
