/* recalgo.h — C-ABI of librecalgo_hip.so: the MI355X-native (gfx950) CTR hot path of
 * tangxyw/RecAlgorithm.
 *
 * The reference has NO FFI / plugin / C-ABI boundary: its hot path is
 * TensorFlow-1.14 op sub-graphs built by plain Python functions.  Each entry point below
 * therefore replaces one such sub-graph; the reference lines it replaces are cited as
 * `file:line` relative to /root/reference.  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add to call these from `algorithm/<MODEL>/...`.
 *
 * Conventions (all entry points):
 *   - return value: hipError_t as int, 0 == success.  Never throws, never aborts.
 *   - all pointers are DEVICE pointers (HBM); the one exception, documented in place, are the short
 *     HOST arrays that list the parts of recalgo_dense1_* (read at launch time).
 *   - fp32 values, int64 ids (the reference's dtypes); id < 0 == OOV / missing value.
 *   - stateless, re-entrant, asynchronous on `stream` (a hipStream_t passed as void*).
 *   - no hidden allocation: outputs and workspaces are caller-owned; required workspace
 *     sizes are given by the matching *_workspace_bytes() query.
 *   - nothing is read or written outside the extents documented per argument.
 *   - one environment knob, read once per process: RECALGO_SCATTER_TILE=32|64|128|256 overrides the
 *     examples-per-workgroup tile of the row-gradient scatter kernels (a tuning aid; results are the same).
 *
 * Embedding storage ("arena"): every table of one model lives in one float arena.  Field f
 * owns rows [row_base[f], row_base[f] + vocab[f]) of width K (uniform-K entry points) —
 * row r of the arena starts at arena + r*K.  A row-sharded deployment keeps
 * rows with r % world == rank on each GPU and passes local row numbers.
 */
#ifndef RECALGO_H_
#define RECALGO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* recalgo_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change).  include/recalgo.abi records the hash of the declarations
 * each version stands for; tests/test_abi.py fails when the declarations change and this number does not. */
#define RECALGO_ABI_VERSION 3
int recalgo_abi_version(void);
/* "gfx950" */
const char* recalgo_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * K1  embedding gather, single-valued fields, uniform width K (K % 4 == 0).
 * Replaces fc.embedding_column + fc.input_layer (TF safe_embedding_lookup_sparse) for
 * single-valued categorical columns: algorithm/DeepFM/deepfm.py:83-93,187-190;
 * algorithm/DCN/dcn.py:97-107,152-153; algorithm/xDeepFM/xdeepfm.py:102-112,157-158;
 * algorithm/PNN/pnn.py:75-85,126-130; algorithm/FiBiNET/fibinet.py:106-116,161-163.
 *   out[b, f, :] = ids[b,f] >= 0 ? arena[row_base[f] + ids[b,f], :] : 0      (bit-exact copy)
 *   ids       [B, F] int64       row_base [F] int64       arena [rows, K]
 *   out       [B, out_stride] fp32; field f is written at columns [out_col + f*K, +K)
 * ------------------------------------------------------------------------------------------ */
int recalgo_embedding_gather_fwd(const int64_t* ids, const float* arena, const int64_t* row_base,
                                 int B, int F, int K, float* out, int out_stride, int out_col,
                                 recalgo_stream_t stream);

/* Live-row bookkeeping of the arena a backward kernel scatters into (optional last argument of the four
 * scatter kernels; NULL or row_live == NULL: none).  Every distinct row a workgroup flushes is test-and-set in
 * row_live (one byte per ARENA row, 4-byte aligned, padded to a multiple of 4) and, on first touch, appended to
 * live_list[live_count[0]++] — exactly what a separate recalgo_mark_live_rows pass over the lookup's ids would
 * do, without that pass.  row_offset = arena row of row 0 of the table the kernel's ids index. */
typedef struct {
    unsigned char* row_live;
    int* live_list;
    int* live_count;
    int64_t row_offset;
} recalgo_live_t;

/* Backward of K1 (TF autodiff of the lookup; SURVEY.md Appendix D "Gather"):
 *   grad_arena[row_base[f] + ids[b,f], :] += g[b, out_col + f*K : +K]   for ids[b,f] >= 0
 * Duplicate ids accumulate (fp32 hardware atomics; order-nondeterministic in the last ulp). */
int recalgo_embedding_gather_bwd(const int64_t* ids, const float* g, const int64_t* row_base,
                                 int B, int F, int K, int g_stride, int g_col, float* grad_arena,
                                 const recalgo_live_t* live, recalgo_stream_t stream);

/* Deterministic alternative to the float-atomic scatter of the four backward kernels (RECALGO_SCATTER=sorted on the
 * host side; parity / checkpoint-resume runs that must be bit-reproducible):
 *   grad[sorted_rows[i], :] += vals[perm[i], :]   for sorted_rows[i] >= 0
 * sorted_rows [M] ascending (a STABLE sort of the lookup's arena rows, -1 = OOV first), perm [M] the sort's
 * permutation, vals [M, K] the per-item row gradients.  Each row is summed in item order by one thread group and
 * written by a plain read-modify-write: no atomics, bit-identical from run to run. */
int recalgo_scatter_rows_sorted(const int64_t* sorted_rows, const int64_t* perm, const float* vals, int64_t M, int K,
                                float* grad, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1m  multi-valued field with combiner='mean' (CSR bags).
 * Replaces fc.embedding_column(col, K, combiner='mean') for list-valued columns:
 * algorithm/DCN/dcn.py:95,98,103 (`manual_tag_list`, shared `his_read_comment_7d_seq`).
 *   out[b, out_col:+K] = mean over valid (>=0) values of bag b of table[value, :], 0 if none.
 *   values [nnz] int64, offsets [B+1] int64, table = arena + row_base*K.
 * The sum runs sequentially in bag order in fp32 (TF SparseSegmentMean order).
 * ------------------------------------------------------------------------------------------ */
int recalgo_embedding_bag_mean_fwd(const int64_t* values, const int64_t* offsets, const float* table,
                                   int B, int K, float* out, int out_stride, int out_col,
                                   recalgo_stream_t stream);
int recalgo_embedding_bag_mean_bwd(const int64_t* values, const int64_t* offsets, const float* g,
                                   int B, int K, int g_stride, int g_col, float* grad_table,
                                   const recalgo_live_t* live, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1s  sequence gather (zero padded) for DIN.
 * Replaces tf.contrib.feature_column.sequence_input_layer: algorithm/DIN/din.py:207-214.
 *   out[b, t, :] = t < len(b) && values[offsets[b]+t] >= 0 ? table[value,:] : 0, t < T
 *   seq_len[b]   = min(offsets[b+1]-offsets[b], T)           (int32; counts OOV entries)
 * ------------------------------------------------------------------------------------------ */
int recalgo_sequence_gather_fwd(const int64_t* values, const int64_t* offsets, const float* table,
                                int B, int T, int K, float* out, int32_t* seq_len,
                                recalgo_stream_t stream);
int recalgo_sequence_gather_bwd(const int64_t* values, const int64_t* offsets, const float* g,
                                int B, int T, int K, float* grad_table, const recalgo_live_t* live,
                                recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1+K2+K3  DeepFM sparse path, fused: gather + FM first order + FM second order + deep_input.
 * Replaces algorithm/DeepFM/deepfm.py:179-181 (indicator -> dense(1)), :184-200 (sum-square
 * FM) and :204 (concat) in one pass over the gathered rows.
 *   emb[b, f*K:+K] = row(b,f)                                  (== deep_input, bit-exact)
 *   fm1[b] = bias[0] + sum_f (ids[b,f]>=0 ? w1[row_base[f]+ids[b,f]] : 0)
 *   fm2[b] = 0.5 * sum_k ( (sum_f e_fk)^2 - sum_f e_fk^2 )
 *   field_sum[b, :] = sum_f e_f   ([B, K]; saved for the backward)
 *   w1 [rows] fp32 (the (sum V,1) dense kernel of `fm_first_order_dense`), bias [1].
 * K % 4 == 0, K <= 64.
 * ------------------------------------------------------------------------------------------ */
int recalgo_deepfm_sparse_fwd(const int64_t* ids, const float* arena, const float* w1,
                              const float* bias, const int64_t* row_base, int B, int F, int K,
                              float* emb, float* fm1, float* fm2, float* field_sum,
                              recalgo_stream_t stream);
/* Backward:
 *   row grad (b,f) = g_emb[b,f,:] + g_fm2[b] * (S_b - e_bf)   -> += grad_arena[row]
 *   grad_w1[row]  += g_fm1[b]
 * `emb`, `field_sum` are the tensors saved by the forward.  d(bias) = sum_b g_fm1[b] is left to
 * the caller. */
int recalgo_deepfm_sparse_bwd(const int64_t* ids, const float* emb, const float* field_sum,
                              const float* g_emb, const float* g_fm1, const float* g_fm2,
                              const int64_t* row_base, int B, int F, int K, float* grad_arena,
                              float* grad_w1, const recalgo_live_t* live, const recalgo_live_t* live_w1,
                              recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K4  DCN CrossNet, L layers fused.
 * Replaces cross_layer(x0, xl, index) algorithm/DCN/cross_layer.py:4-26 stacked by
 * algorithm/DCN/dcn.py:157-160:   x_{l+1} = x0 * (x_l . w_l) + b_l + x_l
 *   x0 [B, d] (row stride x_stride), w [L, d], b [L, d], out [B, d] (row stride out_stride).
 * Evaluated through the stack's closed form  x_l = c_l*x0 + B_l,  c_{l+1} = c_l*(1 + x0.w_l) +
 * B_l.w_l,  B_l = sum_{j<l} b_j  (identical math, one batched reduction per example).
 * d % 4 == 0, d <= 1024, 1 <= L <= 6  (beyond: chain recalgo_cross_layer_fwd).
 * ------------------------------------------------------------------------------------------ */
int recalgo_cross_fwd(const float* x0, int x_stride, const float* w, const float* b, int B, int d,
                      int L, float* out, int out_stride, recalgo_stream_t stream);
/* recalgo_embedding_gather_fwd + recalgo_cross_fwd in ONE launch (fc.input_layer over F single-valued embedding columns of one
 * width K feeding the cross network, dcn.py:152-160): the wave that computes an example's stack gathers the example's row from
 * the arena itself — x0 [B, F*K] is an OUTPUT here (bit-exact copy of the table rows, id < 0: zeros; the MLP branch and the
 * backward read it), out as recalgo_cross_fwd.  K % 4 == 0, F * K <= 1024; arena rows already current (a TRAIN lookup's
 * recalgo_scatter_prepare ran before). */
int recalgo_gather_cross_fwd(const int64_t* ids, const float* arena, const int64_t* row_base, int B, int F, int K,
                             const float* w, const float* b, int L, float* x0, int x_stride, float* out, int out_stride,
                             recalgo_stream_t stream);
/* Backward.  Recomputes the per-example scalars from x0 (nothing but x0 is saved by the forward).
 *   g    [B, d] (row stride g_stride) upstream gradient of `out`
 *   g_x0_extra  optional [B, d] (row stride x_stride) added into dx0 (the DNN branch's dx0)
 *   dx0  [B, d] (row stride x_stride);  dw, db [L, d] (overwritten, deterministic two-pass)
 *   workspace: recalgo_cross_bwd_workspace_bytes(B, d, L) bytes.
 * Every workgroup leaves its share of the final dw / db as one partial row [dw_0..dw_{L-1} | db_0..db_{L-1}]
 * (2*L*d floats, recalgo_cross_bwd_partial_rows(B) rows) in the workspace; dw / db are the column sums of those rows.
 * defer_reduce = 0: the sum is a second launch of this call.  defer_reduce != 0: it is left to the step's
 * deferred-sum launch — pass {workspace, dw, rows, 2*L*d, L*d} and {workspace + L*d floats, db, rows, 2*L*d, L*d} as
 * recalgo_colsum_t jobs to recalgo_dense_bwd_weights_reduce; the workspace must stay untouched until then and dw / db
 * may be NULL here. */
int64_t recalgo_cross_bwd_workspace_bytes(int B, int d, int L);
int recalgo_cross_bwd_partial_rows(int B);
/* Single layer with the reference's exact signature cross_layer(x0, xl, index)
 * (algorithm/DCN/cross_layer.py:4): out = x0 * (xl . w) + b + xl, xl distinct from x0.
 * w, b [d].  Backward also returns dxl; workspace as recalgo_cross_bwd_workspace_bytes(B,d,1). */
int recalgo_cross_layer_fwd(const float* x0, const float* xl, int x_stride, const float* w,
                            const float* b, int B, int d, float* out, int out_stride,
                            recalgo_stream_t stream);
int recalgo_cross_layer_bwd(const float* x0, const float* xl, int x_stride, const float* w,
                            const float* b, const float* g, int g_stride, int B, int d, float* dx0,
                            float* dxl, float* dw, float* db, void* workspace,
                            recalgo_stream_t stream);
int recalgo_cross_bwd(const float* x0, int x_stride, const float* w, const float* b, const float* g,
                      int g_stride, const float* g_x0_extra, int B, int d, int L, float* dx0,
                      float* dw, float* db, void* workspace, int defer_reduce, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5  xDeepFM CIN layer on the fp32 matrix cores (implicit GEMM, outer product never stored).
 * Replaces cin_layer(x0, xk, hk_1, index) algorithm/xDeepFM/cin_layer.py:4-30 (einsum + reshape
 * + width-1 conv1d + transpose) and the sum-pooling of algorithm/xDeepFM/xdeepfm.py:173.
 *   out[b, n, d]  = sum_{i<Hk} sum_{j<m} filters[i*m + j, n] * xk[b, i, d] * x0[b, j, d]
 *   pool[b, pool_col + n] = sum_d out[b, n, d]          (row stride pool_stride; pool may be NULL)
 *   x0 [B, m, D], xk [B, Hk, D], filters [Hk*m, N] (the reference's (1, Hk*m, N) variable),
 *   out [B, N, D].   D in {4, 8, 16, 32}; N <= 128.  No bias, no activation (quirk B-8).
 * ------------------------------------------------------------------------------------------ */
int recalgo_cin_layer_fwd(const float* x0, const float* xk, const float* filters, int B, int m,
                          int Hk, int N, int D, float* out, float* pool, int pool_stride,
                          int pool_col, recalgo_stream_t stream);
/* Backward (CIN).  Upstream gradient G = g_out (or 0 if NULL) +
 * broadcast_d(g_pool[b, pool_col + n]) (or 0 if NULL).
 *   dxk[b,i,d] (=|+=) sum_{j,n} W[(i,j),n] x0[b,j,d] G[b,n,d]      (dxk may be NULL)
 *   dx0[b,j,d] (=|+=) sum_{i,n} W[(i,j),n] xk[b,i,d] G[b,n,d]
 *   dfilters[(i,j),n] = sum_{b,d} xk[b,i,d] x0[b,j,d] G[b,n,d]     (overwritten, deterministic)
 * m, Hk, N <= 128, m >= 4.  workspace: recalgo_cin_layer_bwd_workspace_bytes(). */
int64_t recalgo_cin_layer_bwd_workspace_bytes(int B, int m, int Hk, int N, int D);
int recalgo_cin_layer_bwd(const float* x0, const float* xk, const float* filters, const float* g_out,
                          const float* g_pool, int pool_stride, int pool_col, int B, int m, int Hk,
                          int N, int D, float* dx0, int dx0_accumulate, float* dxk,
                          int dxk_accumulate, float* dfilters, void* workspace,
                          recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K9  DIN attention pooling of the behaviour history.
 * Replaces din_attention(query, keys, keys_length, is_softmax) algorithm/DIN/din_attention.py:4-43
 * (tile/concat, three tf.layers.dense f1_att/f2_att/f3_att, mask, softmax | mask-multiply, matmul).
 *   x_t = [q, k_t, q-k_t, q*k_t];  s_t = f3(relu(f2(relu(f1 x_t))))          (4H -> 64 -> 32 -> 1)
 *   is_softmax: w = softmax_t((t < len ? s_t : -2^32+1) / sqrt(H));  else: w_t = s_t * [t < len]
 *   out[b, :] = sum_t w_t * keys[b, t, :]
 *   query [B,H], keys [B,T,H] (zero padded), keys_length [B] int32, f1_w [4H,64], f1_b [64],
 *   f2_w [64,32], f2_b [32], f3_w [32,1], f3_b [1], out [B,H].   T <= 64, H in {4, 8, 16}.
 * ------------------------------------------------------------------------------------------ */
int recalgo_din_attention_fwd(const float* query, const float* keys, const int32_t* keys_length,
                              const float* f1_w, const float* f1_b, const float* f2_w,
                              const float* f2_b, const float* f3_w, const float* f3_b, int B, int T,
                              int H, int is_softmax, float* out, recalgo_stream_t stream);
/* Backward (DIN attention): recomputes the forward, returns dquery [B,H],
 * dkeys [B,T,H] and the six parameter gradients (overwritten; deterministic two-pass sum).
 * workspace: recalgo_din_attention_bwd_workspace_bytes(B, T, H).  d_f1_w == NULL: the second pass is left to the caller —
 * the workspace then holds recalgo_din_attention_bwd_partial_rows(B) rows of recalgo_din_attention_bwd_partial_floats(H)
 * floats laid out [d_f1_w | d_f1_b | d_f2_w | d_f2_b | d_f3_w | d_f3_b], to be summed column-wise in row order (e.g. as
 * jobs of recalgo_dense_bwd_weights_reduce, the step's deferred-sum launch). */
int64_t recalgo_din_attention_bwd_workspace_bytes(int B, int T, int H);
int recalgo_din_attention_bwd_partial_rows(int B);
int recalgo_din_attention_bwd_partial_floats(int H);
int recalgo_din_attention_bwd(const float* query, const float* keys, const int32_t* keys_length,
                              const float* f1_w, const float* f1_b, const float* f2_w,
                              const float* f2_b, const float* f3_w, const float* f3_b,
                              const float* g_out, int B, int T, int H, int is_softmax, float* dquery,
                              float* dkeys, float* d_f1_w, float* d_f1_b, float* d_f2_w,
                              float* d_f2_b, float* d_f3_w, float* d_f3_b, void* workspace,
                              recalgo_stream_t stream);
/* The same backward reading g_out with a row stride (ldg floats, a multiple of 4, rows 16-byte aligned: the attention output's
 * column block of a wider gradient matrix, in place) and adding dq_extra [B][ld_extra] (or NULL) to dquery — the gradient the
 * query's OTHER consumer produced (din.py:240-249: the target embedding feeds the attention and the fcn input), instead of a
 * slice copy and an add launch. */
int recalgo_din_attention_bwd_joined(const float* query, const float* keys, const int32_t* keys_length, const float* f1_w,
                                     const float* f1_b, const float* f2_w, const float* f2_b, const float* f3_w,
                                     const float* f3_b, const float* g_out, int ldg, const float* dq_extra, int ld_extra, int B,
                                     int T, int H, int is_softmax, float* dquery, float* dkeys, float* d_f1_w, float* d_f1_b,
                                     float* d_f2_w, float* d_f2_b, float* d_f3_w, float* d_f3_b, void* workspace,
                                     recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  FiBiNET SENET re-weighting.
 * Replaces senet(input, embedding_dim, reduction_ratio) algorithm/FiBiNET/senet.py:4-36:
 *   z = mean_k(emb);  a = relu(relu(z @ w1) @ w2);  v_out = emb * a[..., None]     (no biases)
 *   emb, v_out [B, F, K];  w1 [F, reduction_dim];  w2 [reduction_dim, F];  a_out [B, F] or NULL.
 * reduction_dim = embedding_dim // reduction_ratio, derived from K (not F) and required to be
 * < K exactly like the reference's assert (senet.py:18-19).
 * ------------------------------------------------------------------------------------------ */
int recalgo_senet_fwd(const float* emb, const float* w1, const float* w2, int B, int F, int K,
                      int reduction_dim, float* v_out, float* a_out, recalgo_stream_t stream);
/* Backward (SENET): recomputes z, h, a.
 *   d_emb (=|+=) g_v * a + broadcast_k(dz) / K;   dw1 [F, Rd], dw2 [Rd, F] overwritten
 *   (deterministic two-pass sum).  workspace: recalgo_senet_bwd_workspace_bytes(). */
int64_t recalgo_senet_bwd_workspace_bytes(int B, int F, int K, int reduction_dim);
int recalgo_senet_bwd(const float* emb, const float* w1, const float* w2, const float* g_v, int B,
                      int F, int K, int reduction_dim, float* d_emb, int accumulate, float* dw1,
                      float* dw2, void* workspace, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K8  FiBiNET bilinear interaction of up to two (input, weight) sets, concatenated on the last
 * axis.  Replaces bilinear_interaction_layer(input, embedding_dim, type, name)
 * algorithm/FiBiNET/bilinear_interaction_layer.py:5-42 and, with two sets, the
 * tf.concat([original, senet], axis=-1) of algorithm/FiBiNET/fibinet.py:177-186.
 *   pairs (i, j) = itertools.combinations(range(F-1), 2), P = (F-1)(F-2)/2 of them: the LAST
 *   field never participates.  Pair index = i(2n-i-1)/2 + (j-i-1), n = F-1.
 *   out[b, pair, out_col + s*K : +K] = (x_s[b,i,:] @ W_s[sel]) * x_s[b,j,:]
 *   type RECALGO_BILINEAR_ALL: W_s [K,K], sel = 0;   _EACH: W_s [F-1,K,K], sel = i;
 *   _INTERACTION: W_s [>= P, K, K], sel = pair (the reference allocates F(F-1)/2 slices and
 *   zip-truncates to the first P: only those are read, and only those receive gradient).
 *   x_s [B, F, K]; x1 == w1 == NULL for a single set.  out row stride `out_stride` floats per
 *   pair (>= out_col + n_sets*K), out_stride % 4 == 0, out_col % 4 == 0.
 * K in {4, 8, 16, 32, 64};  3 <= F <= 128.
 * ------------------------------------------------------------------------------------------ */
#define RECALGO_BILINEAR_ALL 0
#define RECALGO_BILINEAR_EACH 1
#define RECALGO_BILINEAR_INTERACTION 2
int recalgo_bilinear_fwd(const float* x0, const float* w0, const float* x1, const float* w1, int B,
                         int F, int K, int type, float* out, int out_stride, int out_col,
                         recalgo_stream_t stream);
/* Backward (Bilinear).  g has the layout of `out`.
 *   dx_s [B, F, K] overwritten (row F-1 is zero);  dw_s overwritten: [K,K] | [F-1,K,K] | the
 *   first P slices of the interaction weight (deterministic two-pass sum over the batch).
 * workspace: recalgo_bilinear_bwd_workspace_bytes(B, F, K, n_sets, type). */
int64_t recalgo_bilinear_bwd_workspace_bytes(int B, int F, int K, int n_sets, int type);
int recalgo_bilinear_bwd(const float* x0, const float* w0, const float* x1, const float* w1,
                         const float* g, int g_stride, int g_col, int B, int F, int K, int type,
                         float* dx0, float* dw0, float* dx1, float* dw1, void* workspace,
                         recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K6  PNN product layer, algorithm/PNN/pnn.py:133-181.  The reference's D-iteration loops
 *   IPNN (:146-158)  lp_i = || sum_f theta[i,f] e_f ||^2
 *   OPNN (:160-173)  lp_i = sum_{a,c} (s s^T)[a,c] sym(W_i)[a,c],  s = sum_f e_f,
 *                    sym(W) = triu(W) + triu(W)^T - diag(W)        (quirk B-10)
 * are quadratic forms in per-example second-order statistics: lp = phi @ omega with
 *   phi[b, t]   t = (r <= r') over the upper triangle of a Gram matrix, T = R(R+1)/2 columns,
 *               t(r, r') = r*R - r(r-1)/2 + (r' - r)
 *               IPNN: R = F, phi = <e_r, e_r'>;     OPNN: R = K, phi = s_r * s_r'
 *   omega[t, i] = c_t * theta[i,r] * theta[i,r']  |  c_t * W_i[r, r']   (c_t = 1 if r == r' else 2)
 * These entry points build phi / omega and their gradients; the caller runs the plain GEMM
 * relu(emb_flat @ linear_w + phi @ omega + bias) as ONE recalgo_dense_fwd launch with two operand pairs
 * (pnn.py:139,175-181).
 *   emb [B, F, K];  product_w: IPNN [D, F], OPNN [D, K, K];  phi [B, T] with row stride ld_phi >= T floats;
 *   omega [T, D].  The features forward zero-fills the columns [T, ld_phi) of every row, so a caller that pads the
 *   row stride to a multiple of 4 (T = 351 for 26 fields) hands the GEMM a float4-addressable operand whose padding
 *   is inert; the backward reads dphi with its own row stride and ignores the padding columns.
 * ------------------------------------------------------------------------------------------ */
#define RECALGO_PNN_IPNN 0
#define RECALGO_PNN_OPNN 1
int recalgo_pnn_feature_count(int F, int K, int method); /* T */
int recalgo_pnn_features_fwd(const float* emb, int B, int F, int K, int method, float* phi, int ld_phi,
                             recalgo_stream_t stream);
/* d_emb (=|+=) d(phi)/d(emb)^T dphi  (IPNN / OPNN de_f). */
int recalgo_pnn_features_bwd(const float* emb, const float* dphi, int ld_dphi, int B, int F, int K, int method,
                             float* d_emb, int accumulate, recalgo_stream_t stream);
int recalgo_pnn_weights_fwd(const float* product_w, int D, int F, int K, int method, float* omega,
                            recalgo_stream_t stream);
/* d_product_w overwritten; OPNN: entries below the diagonal get exactly 0. */
int recalgo_pnn_weights_bwd(const float* product_w, const float* domega, int D, int F, int K,
                            int method, float* d_product_w, recalgo_stream_t stream);

/* One-unit dense head over a virtual concatenation of n_parts (1..4) row-major inputs x_p [B, widths[p]]:
 *   out[b] = bias[0] + sum_p <x_p[b,:], w[off_p : off_p + widths[p]]>,   off_p = sum of earlier widths
 * = `tf.layers.dense(tf.concat([...], -1), 1)`, the logit tail of every model (dcn.py:160-162,
 * deepfm.py:300, xdeepfm.py, din.py, fibinet.py, pnn.py: last dense of the dnn part), without
 * materialising the concat.  x_parts / dx_parts / widths are HOST arrays (read at launch).
 * Backward: dx_p[b,:] = g[b] * w[off_p:...] (entries of dx_parts, or dx_parts itself, may be NULL),
 * dw[C] = sum_b g[b] x[b,:], dbias[0] = sum_b g[b] (NULL to skip); gradients are assigned, fixed
 * summation order.  C = sum of widths.  workspace of recalgo_dense1_bwd_workspace_bytes(B, C). */
int recalgo_dense1_fwd(const float* const* x_parts, const int* widths, int n_parts, int B, const float* w,
                       const float* bias, float* out, recalgo_stream_t stream);
int64_t recalgo_dense1_bwd_workspace_bytes(int B, int C);
int recalgo_dense1_bwd(const float* const* x_parts, const int* widths, int n_parts, int B, const float* w,
                       const float* g, float* const* dx_parts, float* dw, float* dbias, void* workspace,
                       recalgo_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Context-MLP glue around the library GEMMs (not interaction layers; fused because the step is
 * otherwise dominated by their launch count).  Widths must satisfy recalgo_mlp_width_supported(C)
 * (C % 4 == 0); callers keep their own path for other widths.
 *
 * Backward epilogue of tf.layers.dense(..., activation=tf.nn.relu) (algorithm/DeepFM/deepfm.py:207
 * and the same line in every model_fn):   g_out = g * [y > 0],  dbias = colsum(g_out).
 * y == g_out == NULL: no activation, dbias = colsum(g).  g, y, g_out [rows, C].
 * ------------------------------------------------------------------------------------------ */
int recalgo_mlp_width_supported(int C);
int64_t recalgo_relu_bwd_bias_workspace_bytes(int rows, int C);
int recalgo_relu_bwd_bias(const float* g, const float* y, int rows, int C, float* g_out, float* dbias,
                          void* workspace, recalgo_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * tf.layers.dense on the fp32 matrix cores (csrc/dense.hip; v_mfma_f32_32x32x2_f32: exact fp32, a
 * k-ordered fmaf chain).  Replaces `tf.layers.dense(x, units, activation=relu)` of every model_fn
 * (algorithm/DCN/dcn.py:163-166, DeepFM/deepfm.py:206-208, xDeepFM/xdeepfm.py:178-181, DIN/din.py:226-227,
 * FiBiNET/fibinet.py:191-193, PNN/pnn.py:186-188) and, with the second operand pair, the D-way contraction
 * of the PNN product layer `relu(lz + lp + bias)` (algorithm/PNN/pnn.py:139,146-181).  All matrices fp32
 * row-major, leading dimensions in floats; any M, K, N (operands whose base is 16-byte aligned and whose
 * leading dimension is a multiple of 4 are read as float4, others element-wise; w / w2 have ld = N).
 *   fwd          y[M,N] = act( x[M,K] w[K,N] (+ x2[M,K2] w2[K2,N]) + bias[N] )     act = ReLU when relu != 0;
 *                x2/w2 and bias may be NULL
 *   bwd_input    dx[M,K] (+)= (g (.) [y_mask > 0])[M,N] w[K,N]^T + beta * c_in[M,K]
 *                y_mask (the forward output, same layout as g) and c_in may be NULL; accumulate != 0 adds to dx
 *   bwd_weights  dw[K,N] = x[M,K]^T (g (.) [y_mask > 0]),  dbias[N] = column sums of the masked g (dbias may be NULL)
 *                split over M, the split partials summed in split order by a second pass (deterministic);
 *                workspace: recalgo_dense_bwd_weights_workspace_bytes(M, K, N) (0 => may be NULL), 16-byte aligned.
 *                defer_reduce != 0 leaves the partials in the workspace: dw / dbias are valid only after
 *                recalgo_dense_bwd_weights_reduce() has been called with this call's (M, K, N, workspace, dw, dbias)
 *                — ONE launch then finishes all the layers of a backward pass (each deferred call needs its own
 *                workspace until then).
 * The masked gradient g (.) [y > 0] is applied while tiles are staged and never materialised.
 * ------------------------------------------------------------------------------------------ */
int recalgo_dense_fwd(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2, int K2,
                      const float* bias, int M, int N, int relu, float* y, int ldy, recalgo_stream_t stream);
/* recalgo_dense_fwd that ALSO leaves the batch moments of y for the BatchNorm layer that follows it (tf.layers.dense ->
 * [dropout 0] -> tf.layers.batch_normalization(training=True): deepfm.py:207-211, pnn.py:187-191, fibinet.py:192-196):
 * bn_partials [recalgo_batchnorm_partial_rows(M)][2 N] in the layout of recalgo_batchnorm_moments (per 64-row tile: column
 * means, then sums of squared deviations), written by the epilogue of the tile's workgroups — recalgo_batchnorm_apply then
 * runs without a moments pass over y.  bn_partials == NULL: recalgo_dense_fwd. */
int recalgo_dense_fwd_bn(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2, int K2,
                         const float* bias, int M, int N, int relu, float* y, int ldy, float* bn_partials,
                         recalgo_stream_t stream);
/* ... with the per-channel activation of DIN's fcn layers in between (tf.layers.dense -> dice | prelu ->
 * tf.layers.batch_normalization, /root/reference algorithm/DIN/din.py:262-266): z = x W + b is written to z [M][ldy] (the
 * activation's backward needs it), y = act(z, act_alpha) and bn_partials are the moments of y.  act_kind: RECALGO_ACT_PRELU /
 * RECALGO_ACT_DICE (needs bn_partials, relu == 0), or RECALGO_ACT_NONE: recalgo_dense_fwd_bn. */
#define RECALGO_ACT_NONE (-1)
int recalgo_dense_fwd_act_bn(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2,
                             int K2, const float* bias, int M, int N, int relu, int act_kind, const float* act_alpha,
                             float* z, float* y, int ldy, float* bn_partials, recalgo_stream_t stream);
int recalgo_dense_bwd_input(const float* g, int ldg, const float* y_mask, const float* w, int M, int N, int K,
                            const float* c_in, int ldc, float beta, float* dx, int lddx, int accumulate,
                            recalgo_stream_t stream);
int64_t recalgo_dense_bwd_weights_workspace_bytes(int M, int K, int N);
int recalgo_dense_bwd_weights(const float* x, int ldx, const float* g, int ldg, const float* y_mask, int M, int K, int N,
                              float* dw, float* dbias, void* workspace, int defer_reduce, recalgo_stream_t stream);
/* bwd: both of the above in ONE launch (dx = ... + beta * c_in, no accumulate mode); same arguments, same results. */
int recalgo_dense_bwd(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w, int M, int K,
                      int N, const float* c_in, int ldc, float beta, float* dx, int lddx, float* dw, float* dbias,
                      void* workspace, int defer_reduce, recalgo_stream_t stream);
/* recalgo_dense_bwd for a layer whose input x IS the output of a training-mode BatchNorm (tf.layers.batch_normalization ->
 * tf.layers.dense: deepfm.py:207-211, din.py:262-266): the input-gradient tiles' epilogue also leaves the two column sums
 * that BatchNorm's backward starts with — bn_partials [recalgo_batchnorm_partial_rows(M)][2 K]: per 64-row tile colsum(dx) and
 * colsum(dx * xhat), xhat = (bn_x - bn_mean) * bn_rstd, bn_x [M][K] contiguous = that BatchNorm's INPUT — i.e. the partial
 * rows of recalgo_batchnorm_bwd_sums, so recalgo_batchnorm_bwd_apply (world 1) follows without a pass over dx and bn_x.
 * bn_partials == NULL: recalgo_dense_bwd.
 * dx_relu_mask (may be NULL) [M][ld_mask]: dx := dx_relu_mask > 0 ? dx : 0 before the beta * c_in term — the layer's input x
 * when x IS the ReLU output of the layer below (for units in hidden_units: net = tf.layers.dense(net, units, relu),
 * dcn.py:163-166): that layer's backward then receives g * [y > 0] ready-made and is called with y_mask == NULL (no mask
 * loads in its two GEMMs). */
int recalgo_dense_bwd_bn(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w, int M, int K,
                         int N, const float* c_in, int ldc, float beta, float* dx, int lddx, float* dw, float* dbias,
                         void* workspace, int defer_reduce, const float* bn_x, const float* bn_mean, const float* bn_rstd,
                         float* bn_partials, const float* dx_relu_mask, int ld_mask, recalgo_stream_t stream);
/* recalgo_dense_bwd_bn carrying RIDERS in the same launch: work of OTHER layers over the same M examples whose operands are ready
 * and which nothing in this launch depends on.  Either may be absent (r_x == NULL / c_x0 == NULL), not both.
 *   weight-gradient rider:  r_dw [r_K][r_N] = r_x^T r_g,  r_dbias [r_N] = colsum(r_g)   (r_g arrives masked: no mask is staged)
 *       as recalgo_dense_bwd_weights(r_x, r_ldx, r_g, r_ldg, NULL, M, r_K, r_N, r_dw, r_dbias, r_workspace, defer_reduce = 1, ..):
 *       the layer above this one when its input gradient came out of recalgo_tail_dense_head_fwd_bwd (dcn.py:166-172:
 *       dnn_dense_2's weight gradient rides with dnn_dense_1's backward);
 *   CrossNet rider:  recalgo_cross_bwd(c_x0, c_x_stride, c_w, c_b, c_g, c_g_stride, NULL, M, c_d, c_L, c_dx0, NULL, NULL, c_workspace,
 *       defer_reduce = 1, ..) — the cross branch's backward (dcn.py:157-160), whose upstream gradient is ready as soon as the head's
 *       is; c_workspace: recalgo_cross_bwd_workspace_bytes(M, c_d, c_L), partial rows as recalgo_cross_bwd leaves them
 *       (recalgo_cross_bwd_partial_rows(M) rows).  Its dx0 is NOT joined here: the layer below adds it (c_in, beta = 1).
 *       Only with y_mask == NULL and recalgo_dense_bwd_cross_rider_supported(c_d, c_L) (c_d % 4 == 0, c_d <= 512, 1 <= c_L <= 4).
 * The riders' split partials / partial rows are summed by recalgo_dense_bwd_weights_reduce.  Only where
 * recalgo_dense_bwd_rider_supported(..) == 1 (the GEMMs on the vectorised tile paths); else the caller launches them separately. */
int recalgo_dense_bwd_rider_supported(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w, int M,
                                      int K, int N, float* dx, int lddx, const float* r_x, int r_ldx, const float* r_g, int r_ldg,
                                      int r_K, int r_N);
int recalgo_dense_bwd_cross_rider_supported(int d, int L);
int recalgo_dense_bwd_rider(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w, int M, int K,
                            int N, const float* c_in, int ldc, float beta, float* dx, int lddx, float* dw, float* dbias,
                            void* workspace, int defer_reduce, const float* bn_x, const float* bn_mean, const float* bn_rstd,
                            float* bn_partials, const float* dx_relu_mask, int ld_mask, const float* r_x, int r_ldx,
                            const float* r_g, int r_ldg, int r_K, int r_N, float* r_dw, float* r_dbias, void* r_workspace,
                            const float* c_x0, int c_x_stride, const float* c_w, const float* c_b, const float* c_g, int c_g_stride,
                            int c_d, int c_L, float* c_dx0, void* c_workspace, recalgo_stream_t stream);
typedef struct {
    int M, K, N;
    const void* workspace;
    float* dw;
    float* dbias;          /* may be NULL */
} recalgo_dense_split_t;
/* Plain fixed-order column sums of partial rows that a kernel of the step left behind (the loss tail's
 * recalgo_logit_loss_fwd_bwd): out[i] = sum_{r < rows} partials[r * row_stride + i], i < n. */
typedef struct {
    const float* partials;
    float* out;
    int rows;
    int64_t row_stride;
    int64_t n;
} recalgo_colsum_t;
/* jobs, sums: HOST arrays, read at launch time.  step_dev (may be NULL): the optimizer's int64 step counter; the launch
 * advances it by one (this launch runs once per training step, right before recalgo_adam_tf1_step with
 * advance = 0, which then only reads it — no separate counter launch, no in-kernel race).  With n_jobs == 0 and
 * step_dev != NULL a one-workgroup launch still advances the counter. */
int recalgo_dense_bwd_weights_reduce(const recalgo_dense_split_t* jobs, int n_jobs, const recalgo_colsum_t* sums,
                                     int n_sums, int64_t* step_dev, recalgo_stream_t stream);
/* tf.layers.batch_normalization(net, training=True) (algorithm/DeepFM/deepfm.py:210-211; PNN
 * pnn.py:190-191; FiBiNET fibinet.py:195-196; DIN din.py:233-234), [TF-ext A-8]:
 *   mean/var = batch moments (biased variance);  y = (x - mean) * rsqrt(var + eps) * gamma + beta
 *   moving_mean/var <- moving * momentum + batch * (1 - momentum)   (updated in place; may be NULL)
 *   save_mean, save_rstd [C] are kept for the backward, which returns
 *   dbeta = colsum(g), dgamma = colsum(g * xhat), dx = gamma * rstd / rows * (rows*g - dbeta - xhat*dgamma).
 * workspace: recalgo_batchnorm_workspace_bytes(rows, C) for both directions.  C % 4 == 0; x, y, g, dx and every [C]
 * vector 16-byte aligned.  Two launches each way (per-tile moments / partial sums, then merge + apply). */
int64_t recalgo_batchnorm_workspace_bytes(int rows, int C);
int recalgo_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, int rows, int C,
                                float eps, float momentum, float* moving_mean, float* moving_var,
                                float* y, float* save_mean, float* save_rstd, void* workspace,
                                recalgo_stream_t stream);
/* dx_relu != 0 (the three backward entry points): x IS the output of a ReLU (tf.layers.dense(..., relu) ->
 * tf.layers.batch_normalization, deepfm.py:206-211) — dx is zeroed where x <= 0, i.e. the dense layer's backward receives
 * g * [y > 0] ready-made and runs without mask loads (see recalgo_dense_bwd_bn). */
int recalgo_batchnorm_train_bwd(const float* x, const float* gamma, const float* save_mean,
                                const float* save_rstd, const float* g, int rows, int C, float* dx,
                                float* dgamma, float* dbeta, void* workspace, int dx_relu, recalgo_stream_t stream);
/* The same backward, continued through the per-channel activation that produced x = act(act_z, act_alpha) (DIN's dense ->
 * dice | prelu -> batch_norm, din.py:262-266; see recalgo_activation_bwd): dx is then dL/d(act_z), dalpha [C] = dL/d(alpha).
 * dalpha == NULL: the recalgo_batchnorm_partial_rows(rows) partial rows [C] of dalpha are left at float offset
 * partial_rows * 2 * C of the workspace for the caller to sum (a job of recalgo_dense_bwd_weights_reduce).
 * workspace: recalgo_batchnorm_bwd_act_workspace_bytes(rows, C).  act_kind == RECALGO_ACT_NONE: recalgo_batchnorm_train_bwd.
 * sums (or NULL): the [partial_rows][2 C] rows (colsum g | colsum g * xhat per 64-row tile) when the kernel that produced g has
 * already left them (recalgo_dense_bwd_bn) — the first of the two launches is then skipped. */
int64_t recalgo_batchnorm_bwd_act_workspace_bytes(int rows, int C);
int recalgo_batchnorm_train_bwd_act(const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                    const float* g, const float* sums, int rows, int C, int act_kind, const float* act_z, const float* act_alpha,
                                    float* dx, float* dgamma, float* dbeta, float* dalpha, void* workspace, int dx_relu,
                                    recalgo_stream_t stream);
/* Sync-BatchNorm building blocks (data parallel, N > 1, `sync_batch_norm`): the two launches of each direction as separate
 * entry points, so that the per-tile partials of all ranks — [recalgo_batchnorm_partial_rows(rows)][2][C] floats per rank:
 * (tile mean | tile M2) forward, (colsum g | colsum g * xhat) backward — can be all-gathered rank-major in between.
 * `partials` of _apply / _bwd_apply: [world][partial_rows][2][C]; every rank holds `rows` examples; the statistics are those
 * of the world * rows examples (== tf.layers.batch_normalization on the concatenated batch, bit-identical to the one-rank
 * kernels on it when rows % 64 == 0).  _bwd_apply: dx uses the sums over all ranks, dgamma / dbeta receive THIS rank's
 * share (the data-parallel all-reduce of the dense gradients adds the ranks up).  world = 1 == the fused entry points. */
int recalgo_batchnorm_partial_rows(int rows);
int recalgo_batchnorm_moments(const float* x, int rows, int C, float* partials, recalgo_stream_t stream);
int recalgo_batchnorm_apply(const float* x, const float* gamma, const float* beta, const float* partials, int world, int rows,
                            int C, float eps, float momentum, float* moving_mean, float* moving_var, float* y,
                            float* save_mean, float* save_rstd, recalgo_stream_t stream);
int recalgo_batchnorm_bwd_sums(const float* x, const float* save_mean, const float* save_rstd, const float* g, int rows, int C,
                               float* partials, recalgo_stream_t stream);
int recalgo_batchnorm_bwd_apply(const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                const float* g, const float* partials, int world, int rank, int rows, int C, float* dx,
                                float* dgamma, float* dbeta, int dx_relu, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sibling models on the same kernels: their remaining interaction steps.
 *   NFM bi-interaction pooling (algorithm/NFM/nfm.py:155-167): emb [B, F, K] ->
 *       out[b, k] = 0.5 * ((sum_f e_fk)^2 - sum_f e_fk^2)          (the FM second order without the sum over k)
 *       bwd: d_emb[b, f, k] = g[b, k] * (S[b, k] - e[b, f, k])
 *   AFM attention pooling (algorithm/AFM/afm.py:184-188): pairs [B, P, K] (the pair Hadamard products: the bilinear
 *       kernel with W = I), att [B, P] (the attention MLP's scores: the dense kernels) ->
 *       score = softmax over P, out[b, :] = sum_p score[b, p] * pairs[b, p, :];  score [B, P] is saved for
 *       bwd: d_pairs[b, p, :] = score * g[b, :], d_att = score * (<g, pairs_p> - sum_q score_q <g, pairs_q>)
 *       (K <= 64; the backward parks <g, pairs_p> in LDS: P <= 4096)
 *   FFM field-aware pair dots (algorithm/FFM/ffm.py:146-160): x [B, F, F-1, K], row (i, s) = field i looked up in
 *       its s-th sub-table -> out[b] = sum_{i<j} <x[b, i, j-1, :], x[b, j, i, :]>
 *       bwd: dx[b, a, s, :] = g[b] * x[b, partner(a, s), :]
 * ------------------------------------------------------------------------------------------ */
int recalgo_bi_interaction_fwd(const float* emb, int B, int F, int K, float* out, recalgo_stream_t stream);
int recalgo_bi_interaction_bwd(const float* emb, const float* g, int B, int F, int K, float* d_emb, recalgo_stream_t stream);
int recalgo_attention_pool_fwd(const float* pairs, const float* att, int B, int P, int K, float* out, float* score,
                               recalgo_stream_t stream);
int recalgo_attention_pool_bwd(const float* pairs, const float* score, const float* g, int B, int P, int K, float* d_pairs,
                               float* d_att, recalgo_stream_t stream);
int recalgo_ffm_pairs_fwd(const float* x, int B, int F, int K, float* out, recalgo_stream_t stream);
int recalgo_ffm_pairs_bwd(const float* x, const float* g, int B, int F, int K, float* dx, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a14  loss tail: sigmoid + mean sigmoid cross entropy, forward and d(loss)/d(logit) fused.
 * Replaces tf.sigmoid + tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits)
 * algorithm/DeepFM/deepfm.py:217,235 (same in all six model_fns).
 *   prob[b] = sigmoid(x_b);  loss[0] = mean_b( max(x,0) - x*z + log1p(exp(-|x|)) )
 *   dlogit[b] = (prob[b] - z_b) * grad_scale / B          (dlogit may be NULL)
 * ------------------------------------------------------------------------------------------ */
int recalgo_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, int B, float grad_scale,
                               float* prob, float* loss, float* dlogit, recalgo_stream_t stream);

/* The whole logit / loss tail of a TRAIN step in ONE launch (replaces recalgo_dense1_fwd + recalgo_sigmoid_ce_fwd_bwd +
 * recalgo_dense1_bwd when the loss-gradient seed `grad_scale` is known before the forward):
 *   logit[b] = sum_p <x_p[b, :], w_p> + bias + addend0[b] + addend1[b]      (1..4 parts, each with ITS OWN weight
 *              vector: tf.layers.dense(concat, 1) or a sum of one-unit heads, xdeepfm.py:163,175,182,184;
 *              addends: DeepFM's FM first / second order logits, deepfm.py:214)
 *   prob, mean sigmoid-CE as recalgo_sigmoid_ce_fwd_bwd;  dlogit[b] = d loss / d logit * grad_scale
 *   dx_p[b, :] = dlogit[b] * w_p          (dx_parts[p] may be NULL);  relu_parts (host array of n_parts flags, may be NULL):
 *              dx_p[b, j] = 0 where x_p[b, j] <= 0 — part p IS a ReLU output (the last hidden layer, dcn.py:166-170) and the
 *              layer that produced it gets its gradient already masked (see recalgo_dense_bwd_bn)
 *   partials [recalgo_logit_loss_partial_rows(B)][C + 2], C = sum of widths: per-workgroup partial sums of
 *   [dw over the C concatenated columns | d bias | loss]; their fixed-order column sums (recalgo_colsum_t jobs of
 *   recalgo_dense_bwd_weights_reduce) are dw_p, d bias and the loss value.  bias, addend0/1 may be NULL.
 *   loss_addend (device scalar, may be NULL) is added to the loss VALUE (a regulariser term whose gradient is
 *   handled elsewhere: DIN's mini-batch-aware regularisation, din.py:254-257). */
int64_t recalgo_logit_loss_partial_rows(int B);
int recalgo_logit_loss_fwd_bwd(const float* const* x_parts, const float* const* w_parts, const int* widths, int n_parts,
                               const float* bias, const float* addend0, const float* addend1, const float* labels,
                               const float* loss_addend, int B, float grad_scale, float* logit, float* prob, float* dlogit,
                               float* const* dx_parts, const int* relu_parts, float* partials, recalgo_stream_t stream);

/* The last hidden layer TOGETHER WITH that tail, in ONE launch (algorithm/DCN/dcn.py:166-172: the last tf.layers.dense(.., relu)
 * of dnn_part, tf.concat([cross_vec, dnn_vec], -1), tf.layers.dense(output, 1), and the loss tail above):
 *   h3 = relu(h2 w3 + b3)                               h2 [B, K2]: the ReLU output of the layer below; w3 [K2, N3], b3 [N3]
 *   logit[b] = <side[b, :], w_side> + <h3[b, :], w_h3> + head_bias       side [B, Cs] (Cs = 0: no side part, side may be NULL)
 *   prob, mean sigmoid-CE, dlogit as recalgo_logit_loss_fwd_bwd;  d_side = dlogit w_side (d_side may be NULL)
 *   dz3 [B, N3] = dlogit w_h3 where h3 > 0, else 0        the gradient at the layer's pre-activation: its weight gradient is
 *                                                         recalgo_dense_bwd_weights(h2, dz3, mask NULL, ..), a launch of its own
 *   dh2 [B, K2] = (dz3 w3^T) where h2 > 0, else 0         (already masked for the layer that produced h2, see recalgo_dense_bwd_bn)
 *   partials [recalgo_tail_partial_rows(B)][Cs + N3 + 2]: [dw over the head's concatenated columns | d bias | loss] per
 *   workgroup, as recalgo_logit_loss_fwd_bwd's; side_first != 0: the head's columns are [side | h3] (dcn.py:171), else [h3 | side].
 * h3 never reaches HBM.  Served shapes: recalgo_tail_dense_head_supported(K2, N3, Cs) (N3 == 128, K2 in {128, 256, 384, 512},
 * Cs % 4 == 0, Cs <= 1024); h2, w3, side, d_side 16-byte aligned. */
int recalgo_tail_partial_rows(int B);
int recalgo_tail_dense_head_supported(int K2, int N3, int Cs);
int recalgo_tail_dense_head_fwd_bwd(const float* h2, int K2, const float* w3, const float* b3, int N3, const float* side, int Cs,
                                    int side_first, const float* w_side, const float* w_h3, const float* head_bias,
                                    const float* labels, const float* loss_addend, int B, float grad_scale, float* logit,
                                    float* prob, float* dlogit, float* d_side, float* dz3, float* dh2, float* partials,
                                    recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a15  TF1 AdamOptimizer, dense semantics (also what TF1 applies to embedding IndexedSlices:
 * duplicates summed, then m and v of ALL rows decay).  algorithm/DeepFM/deepfm.py:246-250.
 *   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr_t * m / (sqrt(v) + eps)
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller (host double precision), or read
 *   from lr_t_dev[0] when lr_t_dev != NULL (see recalgo_adam_tf1_advance).
 *   zero_grad != 0: g is reset to 0 after use (only non-zero words are rewritten).
 * n % 4 == 0 not required.
 * ------------------------------------------------------------------------------------------ */
int recalgo_adam_tf1_dense(float* p, float* g, float* m, float* v, int64_t n, float lr_t,
                           const float* lr_t_dev, float beta1, float beta2, float eps, int zero_grad,
                           recalgo_stream_t stream);
/* The same dense update over an embedding arena [rows, K] (K in {4,8,16,32,64}) with one liveness
 * byte per row.  A row no batch has touched yet has g = m = v = 0, for which the dense update is
 * the identity; such rows cost one read of g (to detect the first touch) and of the byte instead
 * of 28 bytes per parameter.  Bit-identical to recalgo_adam_tf1_dense on the same buffers.
 * row_live [rows] uint8, in/out; invariant row_live[r] == 0 => m[r,:] == v[r,:] == 0 (start from
 * zeros with zero moments; after restoring moments set row_live[r] = any(m[r,:] | v[r,:] != 0)). */
int recalgo_adam_tf1_rows(float* p, float* g, float* m, float* v, unsigned char* row_live, int64_t rows,
                          int K, float lr_t, const float* lr_t_dev, float beta1, float beta2, float eps,
                          int zero_grad, recalgo_stream_t stream);
/* Live-row list (optimizer cost proportional to the rows a model has touched, dense
 * TF1 semantics kept exactly).  recalgo_mark_live_rows visits the ids of one lookup
 * (row = ids[i] + row_base[i % F], id < 0 skipped; row_base may be NULL with F = 1) and appends every
 * row whose liveness byte was 0 to live_list, bumping live_count[0] (device int).  Call it for each
 * lookup whose gradient is scattered into the arena.  row_live must be 4-byte aligned and padded to
 * a multiple of 4 bytes.  recalgo_adam_tf1_list applies the dense update to the listed rows only: for
 * all other rows g = m = v = 0 and the update is the identity.  The launch is sized by max_rows, the
 * count is read on the device (hipGraph replayable).  Any K >= 1 (float4 path for K in {4..64}). */
int recalgo_mark_live_rows(const int64_t* ids, const int64_t* row_base, int64_t n, int F,
                           unsigned char* row_live, int* live_list, int* live_count, recalgo_stream_t stream);
int recalgo_adam_tf1_list(float* p, float* g, float* m, float* v, const int* live_list,
                          const int* live_count, int64_t max_rows, int K, float lr_t, const float* lr_t_dev,
                          float beta1, float beta2, float eps, int zero_grad, recalgo_stream_t stream);
/* One launch per optimizer step (replaces recalgo_adam_tf1_advance + recalgo_adam_tf1_dense + one
 * recalgo_adam_tf1_list per arena): lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (double precision, on the device, per
 * workgroup), the dense TF1 update of the flat buffer p/g/m/v [n] (n may be 0) and of the live rows of up to 4
 * arenas.  Same arithmetic as the separate entry points (bit-identical results).
 *   advance == 0: t = step_dev[0] (already advanced for this step, e.g. by recalgo_dense_bwd_weights_reduce);
 *                 ticket_dev unused (may be NULL)
 *   advance != 0: t = step_dev[0] + 1 and the last workgroup to finish stores step_dev[0] = t (arrival ticket
 *                 ticket_dev: one int, zero before the first call, left zero by every call; one atomic per
 *                 workgroup on one address — ~12 ns each, use advance == 0 inside a training step)
 * arenas: HOST array, read at launch time.  hipGraph replayable. */
typedef struct {
    float* p; float* g; float* m; float* v;      /* [rows, K] */
    const int* live_list;
    const int* live_count;
    int64_t max_rows;
    int K;
    int lazy;      /* != 0: tf.contrib.opt.LazyAdamOptimizer semantics for this arena (the reference's DIEN, dien.py:328): a row
                      whose gradient is all zero in this step keeps p, m and v — a DEVIATION from tf.train.AdamOptimizer,
                      whose m and v decay (and p moves) for every row every step */
} recalgo_adam_arena_t;
int recalgo_adam_tf1_step(float* p, float* g, float* m, float* v, int64_t n, const recalgo_adam_arena_t* arenas,
                          int n_arenas, int64_t* step_dev, int* ticket_dev, int advance, float lr, float beta1,
                          float beta2, float eps, int zero_grad, recalgo_stream_t stream);
/* a scatter plan's bucket counters and where their prefix goes (see recalgo_scatter_plan_scan, RECALGO_SCATTER_PRESCANNED) */
typedef struct {
    const uint32_t* total;   /* [nb << counter_shift] */
    uint32_t* offs;          /* [nb] */
    void* sched;             /* uint4 [nb] */
    uint32_t counter_shift, nb_log2;
} recalgo_plan_scan_t;
/* recalgo_adam_tf1_step plus, as n_scans (<= 4) extra workgroups of the same launch, the prefix scan of the scatter plans whose
 * recalgo_scatter_apply follows with RECALGO_SCATTER_PRESCANNED (every count of those plans must have been enqueued before). */
int recalgo_adam_tf1_step_plans(float* p, float* g, float* m, float* v, int64_t n, const recalgo_adam_arena_t* arenas,
                                int n_arenas, int64_t* step_dev, int* ticket_dev, int advance, float lr, float beta1,
                                float beta2, float eps, int zero_grad, const recalgo_plan_scan_t* scans, int n_scans,
                                recalgo_stream_t stream);
/* Housekeeping for the live-row list: rebuild live_list in ascending row order from the liveness
 * bytes (same set, live_count rewritten with the same total).  recalgo_mark_live_rows appends rows in
 * first-touch order; an address-ordered list lets recalgo_adam_tf1_list walk HBM monotonically.  Run
 * it between steps on the stream the steps run on (every few dozen steps).  row_live as for
 * recalgo_mark_live_rows (4-byte aligned, padded to a multiple of 4 with zeros); workspace of
 * recalgo_order_live_list_workspace_bytes(rows).  Three launches, no host synchronisation. */
int64_t recalgo_order_live_list_workspace_bytes(int64_t rows);
int recalgo_order_live_list(const unsigned char* row_live, int64_t rows, int* live_list, int* live_count,
                            void* workspace, recalgo_stream_t stream);
/* Owner bucketing of one batch's row requests for row-sharded arenas (the reference is
 * single-process and has no counterpart): global row r is owned by rank r % world at local row
 * r / world.  rows [M] int64 (< 0 = no request).  Outputs, all int64: send_local [world*cap] — bucket
 * w occupies [w*cap, (w+1)*cap), holds the owner-local rows requested from rank w, -1 padded;
 * send_pos [world*cap] (may be NULL) — the request index i each bucket entry came from, -1 for padding;
 * req_slot [M] — the bucket entry of request i, -1 for no request or a dropped one.  A bucket that
 * would exceed cap sets overflow[0] = 1 (sticky, never cleared here) and drops the surplus requests.
 * counters [world] int32 is scratch.  Order inside a bucket is unspecified.  Two launches, no host
 * synchronisation (hipGraph replayable). */
int recalgo_exchange_plan(const int64_t* rows, int64_t M, int world, int64_t cap, int64_t* send_local,
                          int64_t* send_pos, int64_t* req_slot, int* counters, unsigned char* overflow,
                          recalgo_stream_t stream);
/* De-duplication of one batch's row requests before the owner bucketing (a Zipf batch asks for its hot rows many
 * times; only one request per distinct row needs to cross xGMI).  rows [M] int64 (< 0 = no request), M < 2^30.
 * rep [M] int64: the SMALLEST request index that asks for the same row as request i (rep[i] = i for the first
 * request of a row and for rows < 0); unique_rows [M] = rows[i] where rep[i] == i, -1 elsewhere.  Plan the exchange
 * on unique_rows; request i then reads / accumulates into the staged row of request rep[i].  Deterministic (does
 * not depend on the arrival order of the atomics), three launches, no host synchronisation.  workspace of
 * recalgo_dedup_rows_workspace_bytes(M) bytes (-1 for an M out of range), contents irrelevant on entry. */
int64_t recalgo_dedup_rows_workspace_bytes(int64_t M);
int recalgo_dedup_rows(const int64_t* rows, int64_t M, int64_t* unique_rows, int64_t* rep, void* workspace,
                       recalgo_stream_t stream);
/* hipGraph-replayable step counter: step_dev[0] += 1; lr_t_dev[0] = lr*sqrt(1-b2^t)/(1-b1^t)
 * (double precision on device).  Pass lr_t_dev to recalgo_adam_tf1_dense to override lr_t. */
int recalgo_adam_tf1_advance(int64_t* step_dev, float lr, float beta1, float beta2, float* lr_t_dev,
                             recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a12 / K10  DIN activations.  Replaces prelu(x, name) / dice(x, name)
 * algorithm/DIN/activations.py:4-17 / :20-37 (used at algorithm/DIN/din.py:228-232).
 *   PReLU: y = max(0,x) + alpha*min(0,x)
 *   Dice : p = sigmoid(x / sqrt(1 + 1e-3));  y = x*p + alpha*x*(1-p)
 *          (the reference's BN has no training= argument: always inference with the
 *           never-updated moving stats (0,1), center=False, scale=False — SURVEY.md B-5)
 *   x, y, gy, dx [rows, C];  alpha, dalpha [C]
 * ------------------------------------------------------------------------------------------ */
#define RECALGO_ACT_PRELU 0
#define RECALGO_ACT_DICE 1
/* tf.concat([...], axis=-1) of 1..4 contiguous row-major parts [B, widths[p]] into out [B, sum widths] AND
 * sum_out[0] = scale * sum(out^2) — the VALUE of DIN's mini-batch-aware regulariser (din.py:249-257; its gradient is the
 * beta * C epilogue of the first fcn layer's recalgo_dense_bwd) — in one launch; the per-workgroup partial sums are added in
 * a fixed order by the workgroup that finishes last (bit-reproducible).  workspace: recalgo_concat_sumsq_workspace_bytes(B)
 * bytes, the first 64 zero-filled once before the first use. */
int64_t recalgo_concat_sumsq_workspace_bytes(int B);
/* dst[0, nbytes) = src[0, nbytes), device to device, one launch sized to the chip (the batch of a captured training step —
 * id matrix + labels, one allocation, ~0.9 MB — into the graph's static input buffers: the runtime's own copy kernel takes
 * 7 us for it, this one ~2).  Any alignment; the ranges must not overlap.  Stands for the feed of
 * dataset.make_one_shot_iterator().get_next() into the graph (algorithm/DCN/dcn.py:216-225). */
int recalgo_copy_bytes(void* dst, const void* src, int64_t nbytes, recalgo_stream_t stream);
int recalgo_concat_sumsq(const float* const* parts, const int* widths, int n_parts, int B, float* out, float scale,
                         float* sum_out, void* workspace, recalgo_stream_t stream);
int recalgo_activation_fwd(const float* x, const float* alpha, int rows, int C, int kind, float* y,
                           recalgo_stream_t stream);
/* dalpha == NULL: the column sum over the recalgo_activation_bwd_partial_rows(rows, C) partial rows [C] the kernel leaves in
 * the workspace is the caller's (a job of the step's deferred-sum launch). */
int recalgo_activation_bwd_partial_rows(int rows, int C);
int64_t recalgo_activation_bwd_workspace_bytes(int rows, int C);
int recalgo_activation_bwd(const float* x, const float* alpha, const float* gy, int rows, int C,
                           int kind, float* dx, float* dalpha, void* workspace,
                           recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * tf.layers.dropout(x, rate, training=True): y = x * keep / (1 - rate), keep ~ Bernoulli(1 - rate) per element
 * (algorithm/DeepFM/deepfm.py:208-209 — dropout_rate defaults to 0.1, :39; DIN/din.py:235-236; FiBiNET/fibinet.py:193-194;
 * PNN/pnn.py:188-189; NFM/nfm.py:170).  The keep decision of flat element i is a counter-based hash of (seed, call, *step, i)
 * (csrc/dropout.h) — `seed` the variable store's (+ rank), `call` the index of the dropout call inside the model_fn, `step` the
 * optimizer's device-side int64 step counter (NULL: 0), read at run time so that a captured step draws a new mask per replay —
 * or, keep_mask != NULL, the explicit mask (1 = keep, 0 = drop; parity tests replay the masks of the reference run).  No mask
 * is stored: the backward is the same map applied to the gradient with the same key.  recalgo_dropout_keep_mask writes the
 * mask the hash stands for (tests, debugging).  n < 2^32 elements; pointers 16-byte aligned; 0 < rate < 1 (a double: TF forms
 * 1 / (1 - rate) from the Python float and casts it to x.dtype once). */
typedef struct recalgo_dropout {   /* one training-mode dropout call for the kernels that apply it themselves (the *_drop entry points) */
    double rate;
    const float* keep_mask;        /* NULL: the hash */
    unsigned seed, call;
    const int64_t* step;           /* device step counter, NULL: 0 */
} recalgo_dropout_t;
int recalgo_dropout_fwd(const float* x, int64_t n, double rate, const float* keep_mask, unsigned seed, unsigned call,
                        const int64_t* step, float* y, recalgo_stream_t stream);
int recalgo_dropout_bwd(const float* g, int64_t n, double rate, const float* keep_mask, unsigned seed, unsigned call,
                        const int64_t* step, float* dx, recalgo_stream_t stream);
int recalgo_dropout_keep_mask(int64_t n, double rate, unsigned seed, unsigned call, const int64_t* step, float* out,
                              recalgo_stream_t stream);
/* The dropout applied by its NEIGHBOURS instead of by launches of its own (element index = row * width + column of the contiguous
 * [rows, width] tensor, the index space of recalgo_dropout_fwd; drop == NULL: the plain entry point):
 *   recalgo_dense_fwd_drop             recalgo_dense_fwd_bn whose epilogue multiplies y = act(x w + b) by keep / (1 - rate) before
 *                                      the store and the BatchNorm tile moments — tf.layers.dense(relu) -> tf.layers.dropout
 *                                      [-> tf.layers.batch_normalization], deepfm.py:207-211 / pnn.py:187-191 / fibinet.py:192-196.
 *                                      ldy == N.  Its backward needs no mask of its own: y > 0 <=> relu > 0 and kept, so the
 *                                      consumer that masks its input gradient with y (recalgo_batchnorm_train_bwd_drop dx_relu)
 *                                      only has to scale it by dx_scale = 1 / (1 - rate);
 *   recalgo_batchnorm_apply_drop       recalgo_batchnorm_apply whose store multiplies by keep / (1 - rate) — tf.layers.
 *                                      batch_normalization -> tf.layers.dropout, din.py:233-236;
 *   recalgo_batchnorm_train_bwd_drop   recalgo_batchnorm_train_bwd_act with g read as g * keep / (1 - rate) (g_drop: the dropout
 *                                      BEHIND the BatchNorm; sums must be NULL — sums left by a dgrad epilogue are those of the
 *                                      un-dropped gradient) and / or dx scaled where dx_relu lets it through (dx_scale: the
 *                                      dropout IN FRONT of the BatchNorm). */
int recalgo_dense_fwd_drop(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2, int K2,
                           const float* bias, int M, int N, int relu, float* y, int ldy, float* bn_partials,
                           const recalgo_dropout_t* drop, recalgo_stream_t stream);
int recalgo_batchnorm_apply_drop(const float* x, const float* gamma, const float* beta, const float* partials, int world, int rows,
                                 int C, float eps, float momentum, float* moving_mean, float* moving_var, float* y, float* save_mean,
                                 float* save_rstd, const recalgo_dropout_t* out_drop, recalgo_stream_t stream);
int recalgo_batchnorm_train_bwd_drop(const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                     const float* g, const float* sums, int rows, int C, int act_kind, const float* act_z,
                                     const float* act_alpha, float* dx, float* dgamma, float* dbeta, float* dalpha, void* workspace,
                                     int dx_relu, float dx_scale, const recalgo_dropout_t* g_drop, recalgo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a15 / a16 / f1  Row-gradient scatter without float atomics, fused with the sparse optimizer.
 * Replaces, for every embedding variable, TF autodiff's IndexedSlices gradient of the lookup plus
 * tf.train.AdamOptimizer(...).minimize (algorithm/DeepFM/deepfm.py:246-250, same in all six hot-path models;
 * dense semantics on embedding variables: SURVEY.md A-10) or tf.contrib.opt.LazyAdamOptimizer
 * (algorithm/DIEN/dien.py:328).
 *
 * A `source` is one lookup's request list and (for `apply`) its per-request gradient rows:
 *   dense id matrix   ids [n_ex, F] int64 (< 0: no row), offsets = NULL:
 *                     request (e, f) -> arena row  ids[e, f] + base + (row_base ? row_base[f] : 0)
 *   ragged sequences  ids = values [nnz], offsets [n_ex + 1]: request (e, f), f < min(len_e, F) ->
 *                     values[offsets[e] + f] + base                                  (din.py:207-214, F = T)
 *   gradient row of request (e, f): K floats at  g + e * g_stride + g_col + f * g_fmul
 *                     (g_fmul = K for a [n_ex, F * K] matrix, 0 when all fields of an example share one row).
 * All requests of a step that address one arena form a PLAN (<= RECALGO_SCATTER_MAX_SOURCES sources, < 2^31
 * slots, arena rows < 2^31).  The plan lives in a SLOT space cut into tiles of 256: a source takes
 * recalgo_scatter_source_slots(n_ex, F, ragged) slots (id matrix: field-major, F x (n_ex rounded up to 256), so that a
 * tile is 256 consecutive examples of one field; ragged: n_ex * F rounded up), source k starts at the sum of the slots
 * of the sources before it (`first_request`), and plan_requests is the slot capacity the workspace was sized for.
 *   recalgo_scatter_prepare  ONE launch per source, any time before `apply` (normally right before the lookup's forward
 *                            kernel); `flags` selects its parts, which run as workgroup ranges of one grid:
 *                              RECALGO_PREPARE_COUNT    the source's tiles add ONE entry per DISTINCT row of a tile to the
 *                                                       row's bucket total (integer atomics on counters spread one per
 *                                                       64 bytes; bucket = hash(row)); needs `source`;
 *                              RECALGO_PREPARE_CATCHUP  with `deferred`: every requested row whose (w, m, v) lags is claimed
 *                                                       (one winner per row over all workgroups) and brought up to step
 *                                                       step_dev[0] + step_offset; `companion_deferred`: the same rows of
 *                                                       the second arena (see `companion` below);
 *                              RECALGO_PREPARE_SWEEP    with `deferred`: this step's share of the arena — every
 *                                                       sweep_period-th block of rows, c = step % sweep_period — is
 *                                                       brought up to that step (once per arena and step: the first
 *                                                       lookup's launch carries it; `source` may be NULL for a
 *                                                       sweep-only launch), so that no row lags more than P + 1 steps.
 *                            plan_workspace is required.
 *   recalgo_scatter_apply    once per plan, with the SAME sources in the SAME order (their g now set): two launches —
 *                            `place` (every entry goes to its bucket: position = the bucket's prefix + a range drawn from
 *                            the bucket's cursor + its rank in the tile; the duplicates of a tile are summed in request
 *                            order into one partial row) and `apply` (one workgroup per bucket orders its entries by
 *                            the unique key (row, slot) and groups them by row).  The prefix of the bucket totals is
 *                            computed by `place` itself, or — mode | RECALGO_SCATTER_PRESCANNED — was left by
 *                            recalgo_adam_tf1_step_plans.  A gradient row may be formed on load from a source's fm_*
 *                            fields: g + fm_scale[e] * (fm_sum[e, :] - fm_emb[e, f, :]), DeepFM's second-order term
 *                            (deepfm.py:196-200).  The owner of a row adds its gradient rows in a fixed order — no float
 *                            atomics, bit-reproducible — and finishes, by `mode`:
 *     RECALGO_SCATTER_GRAD       grad[row, :] += sum            (+ `live`: the row joins the live-row list)
 *     RECALGO_SCATTER_ADAM       TF1 Adam with dense semantics, evaluated lazily but EXACTLY: (w, m, v)[row] first
 *                                replay the g = 0 updates of the steps since last_step[row], then take this step's
 *                                update with lr_t(t), t = step_dev[0] + step_offset; last_step[row] = t; lr_t(t) is
 *                                recorded in deferred->lr_ring.  Bit-identical to recalgo_adam_tf1_dense over the whole
 *                                arena once recalgo_adam_deferred_sweep has flushed it (every Adam update of the library
 *                                uses the same v_sqrt / v_rcp form, csrc/deferred.h).
 *     RECALGO_SCATTER_LAZY_ADAM  LazyAdamOptimizer: exactly the rows of this step's requests take the Adam update
 *                                (whole rows, also rows whose summed gradient is 0); all other rows keep w, m, v.
 *                            In the ADAM modes a non-NULL `grad` gets the touched rows zeroed (a gradient arena
 *                            that an earlier GRAD call of the same plan filled).
 *                            `companion` (NULL, or a recalgo_scatter_companion_t; K <= 32): a second arena of ONE
 *                            float per row that was looked up with exactly these requests (DeepFM's first-order
 *                            weights, deepfm.py:125-141).  It gets its row sums and optimizer step from the SAME placed
 *                            entries: `place` also sums the scalar gradients of a tile's duplicates, `apply` updates its
 *                            row beside the main arena's, and its catch-up and share of the sweep ride in
 *                            recalgo_scatter_prepare — no launch of its own.  companion->sources[k]: only
 *                            g / g_stride / g_col / g_fmul are read (g = NULL: that lookup had no companion and adds
 *                            nothing).  recalgo_scatter_prepare's companion_deferred brings the claimed rows of the
 *                            second arena up to date along with the first's.
 *   recalgo_adam_deferred_sweep  rows [row_begin, row_end) brought to step_dev[0] + step_offset: the flush before
 *                            EVAL / PREDICT / checkpoint / export, and after the last training step.
 * The workspace (recalgo_scatter_plan_workspace_bytes(plan_requests, nb_log2, K), nb_log2 =
 * recalgo_scatter_plan_buckets_log2(plan_requests)) needs its first recalgo_scatter_plan_header_bytes(nb_log2) bytes (the
 * bucket totals and cursors) zero-filled before its first use and whenever counted sources are dropped without having been
 * applied; `apply` leaves them clean.
 * ------------------------------------------------------------------------------------------ */
#define RECALGO_SCATTER_MAX_SOURCES 16
#define RECALGO_LR_RING 1024
#define RECALGO_SCATTER_GRAD 0
#define RECALGO_SCATTER_ADAM 1
#define RECALGO_SCATTER_LAZY_ADAM 2
/* OR-ed into recalgo_scatter_apply's `mode`: the prefix of the plan's bucket totals (offs, sched) has already been written by a
 * launch that ran after the plan's last count — recalgo_adam_tf1_step_plans with this plan's recalgo_scatter_plan_scan record —
 * and `place` reads it instead of every tile scanning the (cache-line-spread) counters itself. */
#define RECALGO_SCATTER_PRESCANNED 0x100
/* the record of a plan workspace (as passed to recalgo_scatter_prepare / _apply) for recalgo_adam_tf1_step_plans */
int recalgo_scatter_plan_scan(void* plan_workspace, int64_t plan_requests, int nb_log2, recalgo_plan_scan_t* out);
#define RECALGO_PREPARE_COUNT 1      /* recalgo_scatter_prepare flags: add the source's entries to the plan's bucket totals */
#define RECALGO_PREPARE_SWEEP 2      /* ... run this step's share of the deferred-Adam sweep of the arena(s) in the launch */
#define RECALGO_PREPARE_CATCHUP 4    /* ... bring the source's lagging rows (and the companion's) up to date: needs `deferred` */
typedef struct {
    const int64_t* ids;
    const int64_t* offsets;
    const int64_t* row_base;
    int64_t base;
    int n_ex, F;
    const float* g;
    int64_t g_stride;
    int g_col, g_fmul;
    /* FM second-order epilogue (DeepFM, deepfm.py:184-200; Appendix D "FM2"); fm_scale == NULL: none.  The gradient row of
     * request (e, f) is  g[...] + fm_scale[e] * (fm_sum[e, 0:K] - fm_emb[e, f*K : (f+1)*K])  formed on load: fm_scale = the
     * gradient of the second-order logit [n_ex], fm_sum = sum over the fields of the looked-up rows [n_ex, K], fm_emb = the
     * looked-up rows themselves [n_ex, F*K] (both contiguous; 16-byte aligned for K % 4 == 0) */
    const float* fm_scale;
    const float* fm_sum;
    const float* fm_emb;
} recalgo_scatter_source_t;
typedef struct {
    float* w; float* m; float* v;   /* [rows, K] */
    int* last_step;                 /* [rows] int32: 0 = never touched, s > 0 = state valid for optimizer step s */
    float* lr_ring;                 /* [RECALGO_LR_RING] fp32: lr_t(j) at j % RECALGO_LR_RING */
    float beta1, beta2, eps;
} recalgo_deferred_adam_t;
typedef struct {
    const recalgo_scatter_source_t* sources;   /* [n_sources], parallel to `sources` of the call */
    float* w; float* m; float* v;              /* [rows, 1] (ADAM modes) */
    float* grad;                               /* GRAD: the target; ADAM modes: touched rows zeroed when non-NULL */
    const recalgo_deferred_adam_t* deferred;   /* RECALGO_SCATTER_ADAM: the second arena's own last_step / lr ring */
    int64_t rows;
} recalgo_scatter_companion_t;
/* The forward lookups on an arena with deferred-Adam state: identical to recalgo_embedding_gather_fwd /
 * _embedding_bag_mean_fwd / _sequence_gather_fwd / _deepfm_sparse_fwd, except that a row whose state lags
 * (last_step[row] < step_dev[0] + step_offset) is read AS OF that step: the missed g = 0 updates are replayed in registers
 * (bit-identical to the dense optimizer pass) and nothing is written back — the optimizer's `apply` / the sweep do the real
 * catch-up.  deferred = NULL: the plain lookup.  table_row_base = arena row of row 0 of `table` (m, v, last_step are arena
 * based). */
int recalgo_embedding_gather_fwd_deferred(const int64_t* ids, const float* arena, const int64_t* row_base, int B, int F, int K,
                                          float* out, int out_stride, int out_col, const recalgo_deferred_adam_t* deferred,
                                          const int64_t* step_dev, int step_offset, recalgo_stream_t stream);
int recalgo_embedding_bag_mean_fwd_deferred(const int64_t* values, const int64_t* offsets, const float* table, int B, int K,
                                            float* out, int out_stride, int out_col, const recalgo_deferred_adam_t* deferred,
                                            int64_t table_row_base, const int64_t* step_dev, int step_offset,
                                            recalgo_stream_t stream);
int recalgo_sequence_gather_fwd_deferred(const int64_t* values, const int64_t* offsets, const float* table, int B, int T, int K,
                                         float* out, int32_t* seq_len, const recalgo_deferred_adam_t* deferred,
                                         int64_t table_row_base, const int64_t* step_dev, int step_offset,
                                         recalgo_stream_t stream);
int recalgo_deepfm_sparse_fwd_deferred(const int64_t* ids, const float* arena, const float* w1, const float* bias,
                                       const int64_t* row_base, int B, int F, int K, float* emb, float* fm1, float* fm2,
                                       float* field_sum, const recalgo_deferred_adam_t* deferred,
                                       const recalgo_deferred_adam_t* deferred_w1, const int64_t* step_dev, int step_offset,
                                       recalgo_stream_t stream);
int recalgo_scatter_plan_buckets_log2(int64_t n_requests);
int64_t recalgo_scatter_source_slots(int n_ex, int F, int ragged);
int64_t recalgo_scatter_plan_workspace_bytes(int64_t n_slots, int nb_log2, int K);
int64_t recalgo_scatter_plan_header_bytes(int nb_log2);
int recalgo_scatter_prepare(const recalgo_scatter_source_t* source, int K, void* plan_workspace, int64_t plan_requests,
                            int nb_log2, int64_t first_request, int flags, const recalgo_deferred_adam_t* deferred,
                            const recalgo_deferred_adam_t* companion_deferred, int64_t rows, int64_t companion_rows,
                            int sweep_period, const int64_t* step_dev, int step_offset, recalgo_stream_t stream);
/* The FORWARD kernels of several plain lookups in one launch (n_jobs <= 4): kind 0 = recalgo_embedding_gather_fwd(ids = ids [B, F],
 * arena = table, row_base = aux, .., out, out_stride, out_col), kind 1 = recalgo_sequence_gather_fwd(values = ids, offsets = aux,
 * table, B, T = F_or_T, K, out, seq_len).  For the lookups a model issues together behind one recalgo_scatter_prepare_multi
 * (DIN: profile fields, target item, history — three launches of ~5 us each): no deferred view, the rows are current. */
typedef struct {
    int kind;                  /* 0: id-matrix gather, 1: zero-padded sequence gather */
    const int64_t* ids;
    const int64_t* aux;        /* kind 0: row_base [F]; kind 1: offsets [B + 1] */
    const float* table;        /* kind 0: the arena; kind 1: row 0 of the table */
    int B, F_or_T, K;
    float* out;
    int out_stride, out_col;   /* kind 0 only */
    int32_t* seq_len;          /* kind 1 only */
} recalgo_lookup_job_t;
int recalgo_lookup_multi_fwd(const recalgo_lookup_job_t* jobs, int n_jobs, recalgo_stream_t stream);
/* recalgo_scatter_prepare for SEVERAL lookups into one arena in ONE launch (n_sources <= 4; first_requests[i]: the first plan slot
 * of source i, as recalgo_scatter_prepare's first_request): their bucket counts, the catch-up of their requests' lagging rows and
 * — RECALGO_PREPARE_SWEEP — the step's share of the sweep, once.  For lookups the model issues together (DIN's profile fields,
 * target item and history, din.py:198-213: three launches of 6 .. 17 us as separate calls); every forward kernel of those lookups
 * must be enqueued AFTER this call.  No companion arena. */
int recalgo_scatter_prepare_multi(const recalgo_scatter_source_t* sources, int n_sources, const int64_t* first_requests, int K,
                                  void* plan_workspace, int64_t plan_requests, int nb_log2, int flags,
                                  const recalgo_deferred_adam_t* deferred, int64_t rows, int sweep_period, const int64_t* step_dev,
                                  int step_offset, recalgo_stream_t stream);
int recalgo_scatter_apply(const recalgo_scatter_source_t* sources, int n_sources, const recalgo_scatter_companion_t* companion,
                          int K, void* plan_workspace, int64_t plan_requests, int nb_log2, int mode, float* w, float* m,
                          float* v, float* grad, const recalgo_deferred_adam_t* deferred, int64_t rows,
                          const recalgo_live_t* live, const int64_t* step_dev, int step_offset, float lr, float beta1,
                          float beta2, float eps, recalgo_stream_t stream);
int recalgo_adam_deferred_sweep(const recalgo_deferred_adam_t* deferred, int K, int64_t row_begin, int64_t row_end,
                                const int64_t* step_dev, int step_offset, recalgo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECALGO_H_ */
