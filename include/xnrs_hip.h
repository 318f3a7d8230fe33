/*
 * xnrs_hip.h -- C ABI of libxnrs_hip.so: the MI355X (gfx950) implementation of the xnrs
 * user-news scoring hot path (news encode -> user encode -> dot-product score).
 *
 * The reference (tan9zj/xnrs) is 100% Python/PyTorch and defines NO FFI of its own; its interface
 * for this path is the `xnrs.models.components` module API.  Every entry point below therefore
 * cites the reference *method* it replaces (file:line relative to the reference repo root); the
 * Python host side (xnrs_amd/models/...) keeps the reference's class names, constructor
 * signatures and state_dict keys and binds these symbols with ctypes.
 *
 * Conventions
 *   - all tensors are contiguous row-major fp32 DEVICE pointers unless stated; masks are fp32 0/1
 *     exactly like the reference (news_encoding.py:34-50); ids are int32 device pointers.
 *   - nn.Linear weights keep the reference layout W[out][in] (K-contiguous), bias[out] or NULL.
 *   - no device-memory allocation and no host sync: the caller supplies the workspace (size from the
 *     *_workspace_bytes query) and a hipStream_t (as void*; NULL = default stream).  Safe to capture
 *     into a hipGraph.  Every result of a call is ordered on THAT stream when the call returns; the
 *     backward entry points may fork weight-gradient launches onto a library-owned stream and join
 *     them back by events inside the call (one stream + 32 events per device, created at the first
 *     eager backward; XNRS_BWD_SIDE_STREAM=0: never).
 *   - process-global state (documented at its entry points, nothing else exists): the forward-GEMM
 *     arithmetic mode (xnrs_set_gemm_mode), the development knobs (xnrs_reload_knobs), the optional
 *     launch timer (xnrs_profile_*), the status word (xnrs_set_status_word) and that side stream.  Encode / score calls on different streams may run from different
 *     threads; changing one of the three while another thread launches is the caller's race.
 *   - return value: 0 = ok; >0 = hipError_t of the failed launch; <0 = XNRS_E* argument error.
 *     xnrs_error_string() explains either.
 */
#ifndef XNRS_HIP_H
#define XNRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XNRS_ABI_VERSION 6

#define XNRS_OK 0
#define XNRS_EINVAL (-1)     /* bad shape / NULL pointer */
#define XNRS_EHEADS (-2)     /* d_model % n_heads != 0 (reference raises RuntimeError, layers.py:111,133) */
#define XNRS_EWORKSPACE (-3) /* workspace too small */
#define XNRS_EUNSUPPORTED (-4)

/* activation fused into a Linear's epilogue */
#define XNRS_ACT_NONE 0
#define XNRS_ACT_RELU 1
#define XNRS_ACT_TANH 2

/* pooler kinds */
#define XNRS_POOL_ADDITIVE 0 /* layers.AdditiveAttention (layers.py:40-69) */
#define XNRS_POOL_MEAN 1     /* layers.MaskedMean        (layers.py:19-37) */

/* layers.MultiHeadAttention parameters (layers.py:106-118): four Linear(D,D) */
typedef struct {
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
  int32_t n_heads;
  int32_t scaled;   /* 1: scores / sqrt(d_k) (layers.py:139-140) */
  float dropout_p;  /* attention-probability dropout (layers.py:117,148); 0 in eval mode */
  uint64_t seed;    /* counter-based RNG seed for dropout; ignored when dropout_p == 0 */
  /* ABI 6, optional: a DEVICE word added to `seed` by the kernels (forward and backward read it; the caller must not change
   * it between a training forward and its backward).  A grad step captured in a hipGraph bakes `seed` into the graph; with
   * this word -- incremented by the caller inside the captured step -- every replay draws a fresh attention-dropout mask
   * (layers.py:148 draws one per call).  NULL: seed alone. */
  const uint64_t *seed_dev;
} xnrs_mha_params;

/* layers.AdditiveAttention parameters (layers.py:42-45): fc1 Linear(D,A), fc2 Linear(A,1).
 * w1_folded / b1_folded (optional; ABI version 3; the training entry points take them too since round 4 -- the backward
 * call must then be given the same pair as its forward): fc1 folded behind the out-projection of
 * the attention stage passed IN THE SAME CALL -- W1.Wo [A,D] and W1.bo + b1 [A], as xnrs_fold_weights computes them
 * (DESIGN.md section 4.6).  The library keeps no state between calls, so without them every call rebuilds the pair
 * (three short launches, ~20 us); a caller that encodes with the same weights again and again computes them once
 * (xnrs_amd/hip.py caches them per module and weight version).  NULL = rebuilt per call; the bits are the same. */
typedef struct {
  const float *w1, *b1, *w2, *b2;
  int32_t hidden; /* A */
  const float *w1_folded, *b1_folded;
} xnrs_additive_params;

/* nn.Sequential(Linear(in,out), activation, Linear(out,out)) head
 * (news_encoding.py:25-31, user_encoding.py:30-34); biases may be NULL (bias=False).
 * activation: XNRS_ACT_RELU (the reference's default nn.ReLU()), XNRS_ACT_TANH or XNRS_ACT_NONE -- the three the GEMM
 * epilogue and its backward implement; the Python mirror refuses any other `activation` module (ABI version 2 added
 * this field). */
typedef struct {
  const float *w0, *b0, *w2, *b2;
  int32_t out_features;
  int32_t activation;
  /* ABI 6, optional, inference entry points with an attention stage + additive pooler in the same call: the head's first
   * layer folded behind the out-projection (layers.py:154 under news_encoding.py:27-31),
   *   w0_folded [E, D] = W0 . Wo      b0_rowvec [E] = W0 . bo      (xnrs_fold_head_weights; exact algebra)
   * so that the per-news out-projection product disappears: head_1 = act((W0 Wo) po + (W0 bo) s + b0) with po the pooled
   * attention rows and s the sum of the pooling weights.  NULL = out-projection, then the head as written. */
  const float *w0_folded, *b0_rowvec;
} xnrs_head_params;

int32_t xnrs_abi_version(void);
/* Hash of the sources this binary was built from (every .hip file of xnrs_amd/csrc, kernels.h, this header; the Makefile computes
 * it): lets a reader tell the measured binary from the tree without a rebuild (tests/test_abi.py, bench.py). */
const char *xnrs_build_id(void);
const char *xnrs_error_string(int32_t code);

/* ---- nn.Linear (layers.py:60,128-130,154; news_encoding.py:27-31) --------------------------
 * y[M,N] = act(x[M,K] . w[N,K]^T + bias[N]).  If gather_ids != NULL, logical row m of x is row
 * gather_ids[m / gather_S] * gather_S + m % gather_S of the table `x` (device-resident news-token
 * table [n_news,S,D] + id gather fused into the load: SURVEY.md section 8 a0). */
int32_t xnrs_linear_fwd(const float *x, const int32_t *gather_ids, int32_t gather_S, const float *w,
                        const float *bias, float *y, int64_t M, int32_t N, int32_t K, int32_t act,
                        void *stream);

/* ---- layers.MultiHeadAttention.forward (layers.py:120-156) ---------------------------------
 * x:(B,S,D), m:(B,S) fp32 0/1 or NULL -> y:(B,S,D).  Row-mask semantics of layers.py:142-144. */
size_t xnrs_mha_workspace_bytes(int64_t B, int32_t S, int32_t D);
int32_t xnrs_mha_fwd(const float *x, const float *m, const xnrs_mha_params *p, float *y, int64_t B,
                     int32_t S, int32_t D, void *ws, size_t ws_bytes, void *stream);

/* ---- layers.AdditiveAttention.forward (layers.py:47-69) ------------------------------------
 * x:(B,N,D), m:(B,N) or NULL -> y:(B,D); a_out:(B,N) optional attention weights (return_weights). */
size_t xnrs_additive_workspace_bytes(int64_t B, int32_t N, int32_t D, int32_t A);
int32_t xnrs_additive_attention_fwd(const float *x, const float *m, const xnrs_additive_params *p,
                                    float *y, float *a_out, int64_t B, int32_t N, int32_t D, void *ws,
                                    size_t ws_bytes, void *stream);

/* ---- layers.MaskedMean.forward (layers.py:26-37) : y = sum(x*m)/(sum(m)+1e-8) -------------- */
int32_t xnrs_masked_mean_fwd(const float *x, const float *m, float *y, int64_t B, int32_t N, int32_t D,
                             void *stream);

/* ---- xnrs.utils.collaps_mask (utils.py:74-75): hm = clamp(sum_S m, 0, 1) ------------------- */
int32_t xnrs_collapse_mask(const float *m, float *hm, int64_t n_rows, int32_t S, void *stream);

/* ---- TextEncoder.forward (news_encoding.py:34-60) -------------------------------------------
 * x:(n_news,S,D) [or table + ids, see xnrs_linear_fwd], m:(n_news,S) -> y:(n_news,E'), hm:(n_news).
 * att == NULL: no self-attention stage; head == NULL: y is the pooled D-vector (E' = D).
 * If ids != NULL, x and m are the TABLE ([n_table,S,D], [n_table,S]) and n_news = len(ids).
 * `chunk` = news per internal pass (bounds the workspace; 0 = library default).
 * With an attention stage and the additive pooler the out-projection (layers.py:154) is applied once per news behind
 * the pooling (exact algebra); <= 32 tokens and D <= 320 run as one fused kernel (section 4.4). */
size_t xnrs_text_encoder_workspace_bytes(int64_t n_news, int32_t S, int32_t D, int32_t A, int32_t E,
                                         int32_t has_att, int32_t pool_kind, int32_t has_head,
                                         int64_t chunk);
int32_t xnrs_text_encoder_fwd(const float *x, const float *m, const int32_t *ids, int64_t n_news,
                              int32_t S, int32_t D, const xnrs_mha_params *att, int32_t pool_kind,
                              const xnrs_additive_params *pool, const xnrs_head_params *head, float *y,
                              float *hm, int64_t chunk, void *ws, size_t ws_bytes, void *stream);

/* The folded fc1 of an (attention stage, additive pooler) pair, for xnrs_additive_params.w1_folded / b1_folded:
 *   w1f [A, D] = W1 . Wo      b1f [A] = W1 . bo + b1      (layers.py:154 behind layers.py:60, exact algebra)
 * ws: xnrs_fold_weights_workspace_bytes(D, A) of scratch. */
size_t xnrs_fold_weights_workspace_bytes(int32_t D, int32_t A);
int32_t xnrs_fold_weights(const xnrs_mha_params *att, const xnrs_additive_params *pool, int32_t D, float *w1f, float *b1f,
                          void *ws, size_t ws_bytes, void *stream);

/* The folded first head layer for xnrs_head_params.w0_folded / b0_rowvec (b0v may be NULL when att->bo is NULL).
 * ws: xnrs_fold_head_weights_workspace_bytes(D, E) of scratch. */
size_t xnrs_fold_head_weights_workspace_bytes(int32_t D, int32_t E);
int32_t xnrs_fold_head_weights(const xnrs_mha_params *att, const xnrs_head_params *head, int32_t D, float *w0f, float *b0v,
                               void *ws, size_t ws_bytes, void *stream);

/* ---- TextEncoder.forward without the padding work (inference) --------------------------------
 * Same result as xnrs_text_encoder_fwd for 0/1 masks, computing only what can reach the output
 * (news_encoding.py:34-60 + layers.py:47-69,120-156): a masked token row has pooling weight exp(e)*0, so its
 * query projection, attention row, output projection and fc1 row are dead; as a KEY it is alive (the reference
 * masks query rows only, layers.py:142-144), so K and V are still projected for every row.
 *   rows    : int32 [n_valid]   source token row (row of x viewed as [*, D]) of every unmasked token, in order
 *   row_off : int64 [n_news+1]  news n owns the compact rows row_off[n] .. row_off[n+1]; row_off[n_news] = n_valid
 * x is (n_news,S,D), or the table when ids != NULL (ids: table row of each news, used for K/V; `rows` then holds
 * table token rows).  att may be NULL (additive-only towers); the pooler is the additive one.  One pass (the
 * caller bounds n_news per call); hm[n] = (row_off[n+1] > row_off[n]).  Requires S <= 64 and d_k <= 64, d_k % 4 == 0
 * when att != NULL (XNRS_EUNSUPPORTED otherwise -- use the padded entry point). */
size_t xnrs_text_encoder_unpadded_workspace_bytes(int64_t n_news, int64_t n_valid, int32_t S, int32_t D, int32_t A,
                                                  int32_t E, int32_t has_att, int32_t has_head);
int32_t xnrs_text_encoder_fwd_unpadded(const float *x, const int32_t *ids, int64_t n_news, int32_t S, int32_t D,
                                       const int32_t *rows, const int64_t *row_off, int64_t n_valid,
                                       const xnrs_mha_params *att, const xnrs_additive_params *pool,
                                       const xnrs_head_params *head, float *y, float *hm, void *ws, size_t ws_bytes,
                                       void *stream);

/* ---- the same, with the row lists built ON THE DEVICE (ABI version 3) --------------------------------
 * xnrs_text_encoder_fwd_unpadded needs the compact row lists and their count from the host (one device->host read of the
 * counts per call).  This entry point takes the padded inputs of xnrs_text_encoder_fwd (x / m, or a table + ids) and
 * compacts on the device: per pass of `chunk` news a single-workgroup kernel builds the CSR offsets, the live-row list
 * and the list of token rows of the non-empty news (an all-masked news has no live query, so its K / V are never read
 * and are not projected either); the GEMMs are launched over the worst-case row count and read the real one from
 * device memory (tiles past it return at once).  No host sync, no data-dependent launch: the whole call can be captured
 * in a hipGraph.  0/1 masks are a PRECONDITION here (not checked: that would be the host read this entry point exists to
 * avoid); inference, additive pooler, fp32 GEMM mode, S <= 64, head width <= 64 and a multiple of 4 with an attention
 * stage (XNRS_EUNSUPPORTED otherwise: use xnrs_text_encoder_fwd).  Results equal xnrs_text_encoder_fwd bit for bit for
 * prefix masks. */
size_t xnrs_text_encoder_compact_workspace_bytes(int64_t n_news, int32_t S, int32_t D, int32_t A, int32_t E,
                                                 int32_t has_att, int32_t has_head, int64_t chunk);
int32_t xnrs_text_encoder_fwd_compact(const float *x, const float *m, const int32_t *ids, int64_t n_news, int32_t S,
                                      int32_t D, const xnrs_mha_params *att, const xnrs_additive_params *pool,
                                      const xnrs_head_params *head, float *y, float *hm, int64_t chunk, void *ws,
                                      size_t ws_bytes, void *stream);

/* ---- UserEncoder.forward (user_encoding.py:50-81) -------------------------------------------
 * x:(B,H,E), m:(B,H) -> y:(B,E) [, a_out:(B,H) when the pooler is additive and a_out != NULL]. */
size_t xnrs_user_encoder_workspace_bytes(int64_t B, int32_t H, int32_t E, int32_t A, int32_t has_att,
                                         int32_t pool_kind, int32_t has_head);
int32_t xnrs_user_encoder_fwd(const float *x, const float *m, int64_t B, int32_t H, int32_t E,
                              const xnrs_mha_params *att, int32_t pool_kind,
                              const xnrs_additive_params *pool, const xnrs_head_params *head, float *y,
                              float *a_out, void *ws, size_t ws_bytes, void *stream);

/* ---- DotScoring.forward (scoring.py:12-23) --------------------------------------------------
 * u:(B,E), c:(B,C,E) -> r:(B,C); normalize != 0 applies the L2 normalisation of scoring.py:20-22. */
int32_t xnrs_dot_scoring_fwd(const float *u, const float *c, float *r, int64_t B, int32_t C, int32_t E,
                             int32_t normalize, void *stream);

/* ---- training: forward that keeps its activations + backward --------------------------------
 * Autograd of the sequence-encoder pipeline (TextEncoder news_encoding.py:34-60, UserEncoder
 * user_encoding.py:50-81, and their parts when pool_kind == XNRS_POOL_NONE / att == NULL), as needed by
 * the grad step (training.py:402-431: loss.backward()) and by integrated gradients (explain.py:160-166,
 * which needs d score / d x).
 *   fwd_train : same arithmetic as xnrs_text_encoder_fwd / xnrs_user_encoder_fwd, one pass (no chunking),
 *               intermediates are kept in the caller's `saved` buffer (xnrs_seq_encoder_saved_bytes).
 *   bwd       : dy:(n_seq,E') [pool_kind == XNRS_POOL_NONE: (n_seq,L,D)] -> parameter gradients (each
 *               pointer of the *_grads structs may be NULL = not wanted; values are WRITTEN, not
 *               accumulated) and, if dx != NULL, the input gradient dx:(n_seq,L,D) (not available with
 *               ids: a gathered table gets no gradient).  Deterministic: no float atomics. */
#define XNRS_POOL_NONE (-1) /* no pooler: y = att(x), the MultiHeadAttention module alone */
typedef struct {
  float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
} xnrs_mha_grads;
typedef struct {
  float *w1, *b1, *w2, *b2;
} xnrs_additive_grads;
typedef struct {
  float *w0, *b0, *w2, *b2;
} xnrs_head_grads;

size_t xnrs_seq_encoder_saved_bytes(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E, int32_t n_heads,
                                    int32_t pool_kind, int32_t has_head);
int32_t xnrs_seq_encoder_fwd_train(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L,
                                   int32_t D, const xnrs_mha_params *att, int32_t pool_kind,
                                   const xnrs_additive_params *pool, const xnrs_head_params *head, float *y,
                                   float *a_out, float *hm, void *saved, size_t saved_bytes, void *stream);
/* The same forward computing only what can reach the output or a gradient (optional; identical results): with
 * live_rows / live_src_rows / n_live as in xnrs_seq_encoder_bwd_live below, the query projection, the output projection
 * and fc1 run over the unmasked token rows in place and the masked rows of Q, Y and T are stored as zeros (a masked
 * row's pooling weight is exp(e) * 0: nothing downstream, forward or backward, depends on its values; K and V stay
 * dense because the reference masks query rows only, layers.py:142-144).  Used with additive pooling + a mask: with an
 * attention tower as described; without one (StandardRec, NAML's views) fc1 runs over the live rows of x and T of the
 * masked rows is zero.  live_rows == NULL = xnrs_seq_encoder_fwd_train. */
int32_t xnrs_seq_encoder_fwd_train_live(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L,
                                        int32_t D, const xnrs_mha_params *att, int32_t pool_kind,
                                        const xnrs_additive_params *pool, const xnrs_head_params *head, float *y,
                                        float *a_out, float *hm, void *saved, size_t saved_bytes,
                                        const int32_t *live_rows, const int32_t *live_src_rows, int64_t n_live,
                                        void *stream);
size_t xnrs_seq_encoder_bwd_workspace_bytes(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E,
                                            int32_t n_heads, int32_t pool_kind, int32_t has_head);
int32_t xnrs_seq_encoder_bwd(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L, int32_t D,
                             const xnrs_mha_params *att, int32_t pool_kind, const xnrs_additive_params *pool,
                             const xnrs_head_params *head, const void *saved, size_t saved_bytes, const float *dy,
                             float *dx, const xnrs_mha_grads *g_att, const xnrs_additive_grads *g_pool,
                             const xnrs_head_grads *g_head, void *ws, size_t ws_bytes, void *stream);
/* The same backward over the unmasked token rows only (optional speed-up of the grad step; identical gradients up to
 * summation order).  A masked token row has pooling weight exp(e)*0, so every gradient that flows through it is
 * exactly zero; with live_rows (int32 [n_live]: indices of the unmasked rows in the padded [n_seq*L] row space, in
 * order) the row-parallel products of the attention tower -- fc1, output projection, Q projection -- run over those
 * rows in place; K / V gradients stay dense (padded tokens are keys, layers.py:142-144).  live_src_rows: the rows of
 * the live tokens in x when ids != NULL (table rows), NULL otherwise.  Used with additive pooling + a mask (without an
 * attention tower: dW1 and the input gradient's fc1 term over the live rows); ignored (= xnrs_seq_encoder_bwd) otherwise. */
int32_t xnrs_seq_encoder_bwd_live(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L, int32_t D,
                                  const xnrs_mha_params *att, int32_t pool_kind, const xnrs_additive_params *pool,
                                  const xnrs_head_params *head, const void *saved, size_t saved_bytes, const float *dy,
                                  float *dx, const xnrs_mha_grads *g_att, const xnrs_additive_grads *g_pool,
                                  const xnrs_head_grads *g_head, const int32_t *live_rows,
                                  const int32_t *live_src_rows, int64_t n_live, void *ws, size_t ws_bytes, void *stream);

/* Both row lists of the grad step in one argument (ABI 4).  live_* as above.  kv_rows (int32 [n_kv], optional, rides on
 * the live-row path): the token rows -- masked ones included -- of the news that have at least one unmasked token, in
 * order.  The keys and values of a news are read by that news' own queries only, and an all-masked news (an empty history
 * slot, dataset.py:82-85) has no live query: its K and V rows reach neither the output nor a gradient, and its dK / dV
 * rows are exactly zero.  With the list the K|V projection of the forward and the dWk / dWv products of the backward run
 * over those rows only (K and V of the other rows are stored as zeros).  kv_src_rows: the same tokens' rows in x when
 * ids != NULL.  Identical results up to summation order; a NULL list pointer = the plain entry points. */
typedef struct {
  const int32_t *live_rows, *live_src_rows;
  int64_t n_live;
  const int32_t *kv_rows, *kv_src_rows;
  int64_t n_kv;
  /* ABI 5, optional: the Q|K|V image ([n_seq*L, 3D] fp32) of ANOTHER training forward over the same input, ids, mask,
   * row lists and projection weights -- the address of its saved blob + xnrs_seq_encoder_saved_qkv_offset().  The
   * reference's train step encodes the history twice (training.py:406,409) and, with input dropout 0 (every shipped
   * config), the two encodes differ only in their attention-dropout draws: the second forward then skips its Q/K/V
   * projection and reads the first one's image, and so does its backward (which still computes its own dQ|dK|dV and weight
   * gradients).  Identical results, bit for bit.  The caller keeps the other blob alive until this forward's backward
   * has run.  NULL: project as usual. */
  const float *qkv_shared;
  /* ABI 6, optional: the two counts as DEVICE scalars, int64 counts_dev[2] = {n_live, n_kv} (xnrs_build_row_lists writes
   * them).  n_live / n_kv above are then only the capacities of the lists (n_seq * L); every product over a list reads
   * its row count on the device, so the caller never reads the counts back: no host synchronisation in the grad step, and
   * its launch sequence does not depend on the data (hipGraph-capturable).  fp32 GEMM mode, 16-byte aligned operands and
   * D, A multiples of 4 (XNRS_EUNSUPPORTED otherwise: pass host counts). */
  const int64_t *counts_dev;
  /* ABI 6, optional, backward only: ONE weight-gradient product per projection for two backward calls over the same
   * Q|K|V image (qkv_shared above: the reference's two history encodes, training.py:406,409) --
   *   dW = (dQKV_1 + dQKV_2)^T . X   instead of   dQKV_1^T . X + dQKV_2^T . X   (layers.py:128-130 under autograd).
   * dqkv_image: caller-owned [n_seq*L, 3D] fp32.  dqkv_mode XNRS_DQKV_DEFER: this call leaves its dQ|dK|dV there and
   * computes NO gradient of wq/bq/wk/bk/wv/bv (and no input gradient); XNRS_DQKV_MERGE: this call ADDS its dQ|dK|dV to
   * the image and computes those gradients from the sum.  Both calls must pass the same row lists.  Rows of all-masked
   * sequences / masked queries that the lists never reach may stay unwritten. */
  float *dqkv_image;
  int32_t dqkv_mode;
} xnrs_row_lists;
#define XNRS_DQKV_OWN 0
#define XNRS_DQKV_DEFER 1
#define XNRS_DQKV_MERGE 2

/* The row lists of xnrs_row_lists built on the device from the mask (m: (n_seq, L) fp32, or the table's mask with ids):
 * live_rows / kv_rows: int32 [n_seq * L] each (capacity; the first counts[0] / counts[1] entries are written, in row
 * order -- what torch.nonzero gives); live_src_rows / kv_src_rows: the same tokens' rows in the table (required with ids,
 * NULL otherwise); counts: int64 [2] = {n_live, n_kv}.  ws: xnrs_row_lists_workspace_bytes(n_seq).  Two short launches,
 * no host synchronisation.  A token is live when its mask value is != 0 (as the host bookkeeping had it). */
size_t xnrs_row_lists_workspace_bytes(int64_t n_seq);
int32_t xnrs_build_row_lists(const float *m, const int32_t *ids, int64_t n_seq, int32_t L, int32_t *live_rows,
                             int32_t *live_src_rows, int32_t *kv_rows, int32_t *kv_src_rows, int64_t *counts, void *ws,
                             size_t ws_bytes, void *stream);
/* byte offset of the Q|K|V image inside the saved blob of a training forward with these shapes (0 without attention) */
size_t xnrs_seq_encoder_saved_qkv_offset(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E, int32_t n_heads,
                                         int32_t pool_kind, int32_t has_head);
int32_t xnrs_seq_encoder_fwd_train_rows(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L,
                                        int32_t D, const xnrs_mha_params *att, int32_t pool_kind,
                                        const xnrs_additive_params *pool, const xnrs_head_params *head, float *y,
                                        float *a_out, float *hm, void *saved, size_t saved_bytes,
                                        const xnrs_row_lists *rows, void *stream);
int32_t xnrs_seq_encoder_bwd_rows(const float *x, const float *m, const int32_t *ids, int64_t n_seq, int32_t L, int32_t D,
                                  const xnrs_mha_params *att, int32_t pool_kind, const xnrs_additive_params *pool,
                                  const xnrs_head_params *head, const void *saved, size_t saved_bytes, const float *dy,
                                  float *dx, const xnrs_mha_grads *g_att, const xnrs_additive_grads *g_pool,
                                  const xnrs_head_grads *g_head, const xnrs_row_lists *rows, void *ws, size_t ws_bytes,
                                  void *stream);

/* autograd of nn.Linear (xnrs_linear_fwd): dx = dy.W (nullable), dw = dy^T.x, db = colsum(dy) (nullable).
 * gather_ids as in the forward (then dx must be NULL). */
size_t xnrs_linear_bwd_workspace_bytes(int64_t M, int32_t N, int32_t K);
int32_t xnrs_linear_bwd(const float *x, const int32_t *gather_ids, int32_t gather_S, const float *w, const float *dy,
                        float *dx, float *dw, float *db, int64_t M, int32_t N, int32_t K, void *ws, size_t ws_bytes,
                        void *stream);

/* autograd of fc(embedder(idx)) (naml.py:82-86; forward = xnrs_linear_fwd with gather_S = 1):
 * dw:(N,K) = dy^T . table[ids], db:(N) = colsum(dy), d_table:(n_rows,K) = scatter-add of dy.W by ids
 * (deterministic).  ws: xnrs_embedding_linear_bwd_workspace_bytes. */
size_t xnrs_embedding_linear_bwd_workspace_bytes(int64_t M, int32_t N, int32_t K);
int32_t xnrs_embedding_linear_bwd(const float *table, const int32_t *ids, const float *w, const float *dy, float *d_table,
                                  float *dw, float *db, int64_t M, int32_t N, int32_t K, int32_t n_rows, void *ws,
                                  size_t ws_bytes, void *stream);

/* autograd of DotScoring.forward with normalize == 0 (scoring.py:23): du:(B,E), dc:(B,C,E), either nullable */
int32_t xnrs_dot_scoring_bwd(const float *u, const float *c, const float *dr, float *du, float *dc, int64_t B,
                             int32_t C, int32_t E, void *stream);
/* the same with normalize == 1 (scoring.py:20-22: u and c L2-normalised before the product); E <= 1024 */
int32_t xnrs_dot_scoring_norm_bwd(const float *u, const float *c, const float *dr, float *du, float *dc, int64_t B,
                                  int32_t C, int32_t E, void *stream);

/* ---- NewsRecDataset.__getitem__ look-ups + torch.cat (xnrs/data/dataset.py:63-85,97-109) as a device copy ----
 * out[i, :] = table[ids[i], :], i < n, rows of row_floats fp32 (a whole news block: S*D token floats, or S mask
 * floats).  For consumers that need the dense batch (input gradients of the explainer, explain.py:160-166); the
 * encoders gather inside their first load and never call this.  HBM-bound; 16-byte streaming accesses when
 * row_floats % 4 == 0 and both pointers are 16-byte aligned. */
int32_t xnrs_gather_rows(const float *table, const int32_t *ids, float *out, int64_t n, int64_t row_floats,
                         void *stream);

/* ---- device-side batch assembly and evaluation -----------------------
 * Click histories / positives / negatives live on the device as CSR arrays of ROWS into the resident news
 * table ([n_rows,S,D] + mask; `pad_row` = the empty slot: all-zero tokens and mask, dataset.py:82-85).
 *
 * xnrs_assemble_train_batch: NewsRecDataset.__getitem__ 'train' mode + custom_collate_fn
 *   (xnrs/data/dataset.py:54-57,77-85,97-109,147; xnrs/utils.py:190-204) -> hist_rows:(B,l_hist) (the LAST
 *   l_hist clicks, padding behind), cand_rows:(B,1+n_neg) (one random positive, n_neg negatives drawn WITH
 *   replacement); targets are the constant [1,0,..].  Draws: splitmix64(seed, session, slot) % n -- a
 *   counter-based stream (Python's `random` cannot be matched), restated bit for bit by the oracle.
 * xnrs_assemble_eval_batch: 'eval' mode (dataset.py:58-61,149): all positives then all negatives;
 *   cand_off:(B+1) is supplied by the caller (prefix sum of the per-session counts).
 * xnrs_score_csr: r[e] = <vecs[cand_rows[e]], u[cand_sess[e]]> (+ReLU, training.py:392) against news
 *   vectors pre-encoded once per epoch (scoring.py:23 per impression, training.py:194-203).
 * xnrs_rank_metrics: xnrs/evaluation/metrics.py:9-64 per impression after np.nan_to_num
 *   (training.py:210-211); out:(B,9) = ndcg@5, ndcg@10, rr, ctr@1, ctr@10, auc, acc, rec, prec.
 *   Ties in the ranking are broken "higher original index first". */
int32_t xnrs_assemble_train_batch(const int64_t *sess, int64_t B, const int64_t *hist_off, const int32_t *hist_val,
                                  const int64_t *pos_off, const int32_t *pos_val, const int64_t *neg_off,
                                  const int32_t *neg_val, int32_t l_hist, int32_t n_neg, int32_t pad_row, uint64_t seed,
                                  int32_t *hist_rows, int32_t *cand_rows, void *stream);
int32_t xnrs_assemble_eval_batch(const int64_t *sess, int64_t B, const int64_t *hist_off, const int32_t *hist_val,
                                 const int64_t *pos_off, const int32_t *pos_val, const int64_t *neg_off,
                                 const int32_t *neg_val, int32_t l_hist, int32_t pad_row, const int64_t *cand_off,
                                 int32_t *hist_rows, int32_t *cand_rows, int32_t *cand_sess, float *targets, void *stream);
int32_t xnrs_score_csr(const float *vecs, const int32_t *cand_rows, const int32_t *cand_sess, const float *u, float *r,
                       int64_t n_cand, int32_t E, int32_t relu, void *stream);
int32_t xnrs_rank_metrics(const float *scores, const float *targets, const int64_t *cand_off, float *out, int64_t B,
                          void *stream);

/* ---- in-batch InfoNCE (ContrastiveRankingTrainer._compute_contrastive_loss, training.py:433-472) ----
 * emb:(B,E) user embeddings, labels:(B) int64 -> loss:(1).  Same epsilons as the reference: F.normalize
 * eps 1e-12, denominator + 1e-12, mean over rows with >= 1 positive / (count + 1e-8).  `saved`
 * (xnrs_infonce_saved_bytes) carries the normalised embeddings and row sums to the backward, which returns
 * d loss / d emb scaled by the upstream scalar gradient *gout (device pointer).  E <= 1024. */
size_t xnrs_infonce_saved_bytes(int64_t B, int32_t E);
int32_t xnrs_infonce_fwd(const float *emb, const int64_t *labels, int64_t B, int32_t E, float temperature, float *loss,
                         void *saved, size_t saved_bytes, void *stream);
int32_t xnrs_infonce_bwd(const int64_t *labels, int64_t B, int32_t E, float temperature, const void *saved,
                         size_t saved_bytes, const float *gout, float *demb, void *stream);

/* ---- measurement aid (no reference counterpart) ----------------------------------------------
 * When enabled, the sequence-encoder pipeline brackets each kernel launch of the selected stages
 * with hipEvents on the caller's stream (process-global, mutex-guarded; off by default; not
 * hipGraph-capturable while on).  stage_mask bit i selects stage i:
 *   0 qkv GEMM | 1 attention core | 2 out-proj GEMM | 3 fc1+tanh GEMM | 4 pooling | 5 head GEMMs
 *   6 fused short-sequence encoder (stages 0-4 in one launch: S <= 32, D <= 320, news_fused.hip)
 *   7 weight-gradient GEMMs of the backward (dW = dY^T . X) | 8 input-gradient GEMMs (dX = dY . W)
 *   9 attention-core backward
 * (stage 3 also counts the one-launch additive encoder, additive_fused.hip: fc1 + pooling, stage 4 then stays empty)
 * xnrs_profile_read synchronises the recorded events and returns, per stage, the summed launch
 * duration (ms), the number of launches and the summed ALGORITHMIC flops of those launches
 * (arrays of XNRS_PROFILE_STAGES entries), then clears the record. */
#define XNRS_PROFILE_STAGES 10
int32_t xnrs_profile_enable(uint32_t stage_mask);
int32_t xnrs_profile_read(double *ms, int64_t *launches, double *flops);

/* Does the TRAINING forward fold the out-projection behind the pooling right now (knob XNRS_FOLD_TRAIN)?
 * The saved-activation blob of xnrs_seq_encoder_fwd_train* is laid out by that decision and xnrs_seq_encoder_bwd* reads it
 * under the decision of ITS call time: a caller that may reload the knobs between the two (tests, A/B tools) records this
 * value at the forward and refuses the backward when it has changed (xnrs_amd/autograd.py does).  1 / 0. */
int32_t xnrs_train_fold_enabled(void);

/* ---- sticky device status word (ABI 6; no reference counterpart) ------------------------------------------
 * Entry points that never synchronise cannot return an error for a precondition only the device can see.  They OR a bit
 * into ONE caller-owned int32 device word instead (and keep their outputs recognisably wrong: NaN), which the caller reads
 * at its next natural synchronisation point.  xnrs_set_status_word registers the word (process-global like the knobs, and
 * bound to the device that is current at the call: launches on another device ignore it; NULL: none -- the NaN outputs
 * are then the only signal); the caller zeroes it.  xnrs_status_string explains a value.
 *   XNRS_STATUS_NONBINARY_MASK  xnrs_text_encoder_fwd_compact met a mask value other than 0 / 1
 *   XNRS_STATUS_ROW_RANGE       (set by the Python host layer, NewsStore.gather) a table row id outside the table; the
 *                               id was clamped so that no kernel read out of bounds */
#define XNRS_STATUS_NONBINARY_MASK 1
#define XNRS_STATUS_ROW_RANGE 2
int32_t xnrs_set_status_word(int32_t *device_word);
const char *xnrs_status_string(int32_t word);

/* ---- arithmetic mode of the forward GEMMs (nn.Linear call sites listed at xnrs_linear_fwd) ----------
 *   XNRS_GEMM_F32     (0, default) v_mfma_f32_32x32x2_f32: an exact fp32 fmaf chain.
 *   XNRS_GEMM_BF16X3  (1) each fp32 operand is split exactly into three bf16 pieces and the product is
 *                     rebuilt from six v_mfma_f32_32x32x16_bf16 with fp32 accumulation: dropped terms are
 *                     <= 2^-26 |a||b| per product.  Measured against an fp64 product: as close as mode 0 on
 *                     N(0,1) operands (7.6e-7 vs 9.5e-7 normwise), about 2x mode 0's error on operands
 *                     with a wide dynamic range (5.5e-7 vs 2.4e-7; tests/test_hip_split_gemm.py bounds it
 *                     at 1e-6); same distance to the CPU oracle as mode 0 on the full workload.
 *   XNRS_GEMM_BF16X2  (2) two pieces, three products: ~1e-5 relative per product; an opt-in speed knob.
 * Process-wide; the initial value comes from the environment variable XNRS_GEMM_MODE.  It covers every GEMM
 * that runs on the forward-layout kernel: the nn.Linear forwards and the input-gradient products dX = dY . W of
 * the backward (computed against a transposed weight copy); the weight-gradient products dW = dY^T . X always
 * run in mode 0, and so does any launch of fewer than 512 128x128 tiles (the fp32 kernel's smaller tiles win
 * there): in modes 1/2 a row's result can therefore differ by fp32 rounding noise between two batch sizes,
 * whereas mode 0 is bitwise independent of the batch.  Returns the previous mode; values outside 0..2 select 0. */
#define XNRS_GEMM_F32 0
#define XNRS_GEMM_BF16X3 1
#define XNRS_GEMM_BF16X2 2
int32_t xnrs_set_gemm_mode(int32_t mode);
int32_t xnrs_get_gemm_mode(void);

/* ---- development knobs (no reference counterpart) ---------------------------------------------
 * Kernel-selection switches for A/B measurements and tests (XNRS_GEMM_PIPE, _BK, _BUF, _GROUP, _TILE,
 * XNRS_GEMM_SPLIT_MIN_TILES, XNRS_GEMM_DW, XNRS_MHA_LDS, XNRS_MHA_HEADWAVE, XNRS_MHA_PAIR, XNRS_MHA_BWD_FUSED,
 * XNRS_NEWS_FUSED, XNRS_NEWS_FUSED_NPW, XNRS_FOLD_OUT, XNRS_FOLD_TRAIN, XNRS_FC1_ROWDOT, XNRS_MHA_SKIP_MASKED, XNRS_BWD_SIDE_STREAM, XNRS_BWD_SIDE_MIN_ROWS).  The library reads
 * them from the environment ONCE when it is loaded -- no launch calls getenv -- and again only when this function is
 * called.  None changes a result beyond summation / association order (XNRS_FOLD_*: whether the attention
 * out-projection is applied per token row, as the reference writes it, or once per sequence behind the additive
 * pooling; XNRS_FOLD_TRAIN must not change between a training forward and its backward). */
int32_t xnrs_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif /* XNRS_HIP_H */
