// Internal launcher declarations shared by the .hip translation units of libxnrs_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xnrs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- development knobs
// Kernel-selection switches for in-process A/B runs (tools/) and tests.  They are read from the XNRS_* environment
// ONCE, when the library is loaded, and again only on xnrs_reload_knobs() (C ABI): the launch path never calls
// getenv.  Process-global and deliberately not synchronised -- do not reload while another thread is launching.
// None of them changes a result beyond summation order.
struct Knobs {
  int gemm_pipe = 6;            // XNRS_GEMM_PIPE=1|5 force a software pipeline; 6 = the shipped configuration
  int gemm_bk = 32;             // XNRS_GEMM_BK=16|32 (with XNRS_GEMM_PIPE=5)
  int gemm_buf = 1;             // XNRS_GEMM_BUF=0: no raw buffer loads
  long long gemm_group = -1;    // XNRS_GEMM_GROUP=n column tiles per walk group (0 = plain walk); -1 = automatic
  int gemm_tile = -1;           // XNRS_GEMM_TILE=0..3 force a block tile; -1 = cost model
  long long split_min_tiles = 512;  // XNRS_GEMM_SPLIT_MIN_TILES: smallest launch the bf16-split kernel takes
  bool fc1_rowdot = true;       // XNRS_FC1_ROWDOT=0: inference materialises tanh(fc1 x) and the pooling kernel takes the fc2 dot
  int fold_train = 1;           // XNRS_FOLD_TRAIN=0: the training forward / backward keep the per-token out-projection
  int fold_out = 1;             // XNRS_FOLD_OUT=0: inference keeps the per-token out-projection (api.hip "fold")
  int gemm_dw = 2;              // XNRS_GEMM_DW: weight gradients on gemm_dw.hip -- 0 none, 1 the live-row launches, 2 (default) those +
                                // the dense ones that take its 256 x 256 tile, 3 every eligible launch (tests)
  int gemm_dw_tile = 256;       // XNRS_GEMM_DW_TILE=128: never the 256 x 256 tile of gemm_dw.hip
  int mha_lds = -1;             // XNRS_MHA_LDS=0|1 force / forbid the LDS-staged attention kernel; -1 = by shape
  bool mha_pair = true;         // XNRS_MHA_PAIR=0: the first-generation LDS-staged attention kernel instead of mha_core_pair_kernel
  int mha_headwave = 1;         // XNRS_MHA_HEADWAVE=0: generic attention kernel only
  int mha_bwd_fused = 1;        // XNRS_MHA_BWD_FUSED=0: two-kernel attention backward
  int news_fused = 1;           // XNRS_NEWS_FUSED=0|2: never / whenever eligible use the fused short-title news encoder
                                // (news_fused.hip); 1 = where it is faster (S >= 26, >= 192 news)
  int news_fused_npw = 0;       // XNRS_NEWS_FUSED_NPW=1|2 force the news per workgroup of the fused encoder; 0 = by batch size
  int gemm_mode_init = 0;       // XNRS_GEMM_MODE=0|1|2: initial forward-GEMM arithmetic (see gemm_mode())
  bool fast_tanh = true;        // XNRS_FAST_TANH=0: ocml tanhf in the tanh epilogues instead of xnrs::fast_tanh
  int additive_fused = 1;       // XNRS_ADDITIVE_FUSED=0|2: never / whenever eligible use the one-launch additive encoder
                                // (additive_fused.hip); 1 = from a batch that fills the chip (results are bitwise equal)
  int af_fbuf = 1;              // XNRS_AF_FBUF=1|2: MFMA fragment register sets of that kernel
  bool bwd_side_stream = true;  // XNRS_BWD_SIDE_STREAM=0: the backward's weight-gradient products stay on the caller's stream (api.hip: SideLane)
  long long bwd_side_min_rows = 0;  // XNRS_BWD_SIDE_MIN_ROWS: ... for attention towers of at least this many token rows
  bool mha_skip_masked = true;  // XNRS_MHA_SKIP_MASKED=0: pooled encoder calls compute the attention rows of all-masked sequences
                                // and query tiles too (the pooler multiplies them by 0: bitwise the same pooled vectors)
};
const Knobs& knobs();
void reload_knobs();

// A device scalar written by an EARLIER kernel on the same stream (the row counts of device-built lists, GemmArgs::m_dev /
// k_dev): wave-uniform, one load per workgroup.
// (hipGraph note, round 4: replays of the grad step read stale values here -- and in other wave-uniform reads -- under ROCm
// 7.2's graph "packet capture" path on gfx950; invalidating the scalar cache at kernel entry and an agent-scope load here
// reduced but did not remove it, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 removed it with or without them (128 of 128 replays equal
// to the eager step: tools/debug_graph_step.py), so the kernels carry no workaround.)
__device__ __forceinline__ int64_t load_dev_scalar(const int64_t* p) { return *p; }

// ---------------------------------------------------------------- Linear (fp32 MFMA GEMM)
// C[M, nseg*Nseg] = act(A[M,K] . W_s[Nseg,K]^T + bias_s), s = column segment (up to 3 weights
// side by side: the Q/K/V projections share one launch while the parameters stay separate
// tensors, as the reference's state_dict requires).
struct GemmArgs {
  const float* A;
  int32_t a_col;              // 0: A[m][k] row-major (lda = row stride); 1: A given k-major, A^T[k][m] (lda = k stride)
  const int32_t* gather_ids;  // nullable: logical row m of A (ROW layout), or logical k row of a k-major A (a_col),
                              // is physical row ids[m/gather_S]*gather_S + m%gather_S
  int32_t gather_S;
  int32_t c_scatter;          // 1 (ROW A with gather_ids): logical row m of C is that same physical row -- the GEMM
                              // works on a row subset of A and C in place (backward over the unmasked token rows)
  const int32_t* c_scatter_ids;  // nullable, with c_scatter: the C rows follow THIS list instead of gather_ids (A rows
                                 // gathered from a news table, C rows in the batch's own row space)
  int64_t lda;
  const float* W[3];
  const float* bias[3];  // nullable each
  int32_t b_kn;          // 0: W[n][k] (nn.Linear weight, ldw = row stride); 1: B[k][n] k-major (ldw = k stride)
  const int32_t* b_gather_ids;  // nullable (KN B only): k-row gather, same rule as gather_ids
  int32_t b_gather_S;
  int32_t nseg;
  int32_t Nseg;  // columns per segment
  int64_t ldw;
  float* C;
  int64_t ldc;
  int64_t M;
  int64_t K;
  int32_t act;         // 0 none, 1 relu, 2 tanh (applied after bias)
  const float* aux;    // nullable: elementwise factor source with the shape of C (ldaux)
  int64_t ldaux;
  int32_t aux_mode;    // 0 none; 1: C *= (1 - aux^2) (tanh'); 2: C *= (aux > 0) (relu')
  int32_t accumulate;  // 1: C += result
  int32_t nt_store;    // filled in by the launcher: non-temporal C stores (outputs of >= 64 MB stream past the caches)
  // bf16-split modes only (nullable): the weights of segment s pre-split into bf16 planes [3][ldp/16][Nseg][16]
  // (launch_split_weights; ldp = K rounded up to 16, zero padded) -- the kernel then loads B planes as they are
  // instead of splitting the fp32 weights again in every row tile
  const unsigned short* Wp[3];
  int64_t ldp;
  // k-major A only (dW = dY^T . X): also emit the column sums of A over the contraction, colsum[split][M] partials (the
  // bias gradient db = sum_rows dY is a by-product of the tiles this kernel stages anyway); nullable.  The caller
  // reduces the nsplit partial rows (launch_colsum_final).
  float* colsum;
  // forward layout, one segment, no split-K (nullable): instead of storing C, emit per row the dot products of the
  // activated row with rowdot_w over each block of 32 columns: rowdot_out[row * ldrd + col / 32] (the additive pooler's
  // fc2 score straight from the fc1 epilogue: tanh(fc1 x) never exists in memory).  The 32-column blocks and the
  // butterfly order inside them do not depend on the tile shape, so every launch shape gives the same bits.
  const float* rowdot_w;
  float* rowdot_out;
  int64_t ldrd;
  // forward layout, one segment, no split-K (nullable pair): rank-1 term of the epilogue, C[m][n] = fmaf(rowscale[m],
  // rowscale_vec[n], acc) before bias / activation (the out-projection bias behind a pooled out-projection: bo * sum_i a_i)
  const float* rowscale;
  const float* rowscale_vec;
  // forward layout (nullable): the row count lives on the DEVICE -- M is the worst case the grid is sized for, *m_dev
  // (<= M) the rows that exist; tiles past it return at once.  (Row lists compacted on the device: no host round trip,
  // the launch sequence does not depend on the data and can be captured in a hipGraph.)
  const int64_t* m_dev;
  // dW layout (k-major A and B; nullable): the CONTRACTION length lives on the device -- K is the worst case the slabs and
  // the row lists are sized for, *k_dev (<= K) the rows that exist; the kernels cut their K slices from it (an empty slice
  // writes a zero slab).  The live-row / kv-row lists of the grad step built on the device (launch_build_row_lists): no
  // host read of the counts, the launch sequence of the step does not depend on the data (hipGraph-capturable).
  const int64_t* k_dev;
  // with colsum (nullable): the final bias gradient [M]; the split-K reduction launch then also adds up the colsum
  // partials (one launch instead of splitk_reduce + colsum_final); without split-K the kernel writes it directly
  float* colsum_out;
  // split-K dW launches (nullable): output rows >= c2_row0 go to C2 (row 0 of C2 = row c2_row0; same ldc) instead of C, and
  // their bias gradients to colsum_out2 -- ONE product for two parameters whose dY columns lie side by side and whose
  // contraction runs over the same rows (dWk | dWv: the K and V columns of the dQ|dK|dV image over the kv-row list)
  float* C2;
  int64_t c2_row0;
  float* colsum_out2;
  // forward layout with m_dev (optional): the expected fraction of the capacity M that exists (0 = unknown = all), for the
  // tile choice only -- a launch sized for 80 000 rows of which ~25 % are live fills the chip better with smaller tiles
  float m_fill_hint;
  // split-K (deterministic slabs + ordered reduce); set by the caller via slabs/nsplit
  float* slabs;        // nullable workspace of nsplit * M * ldc floats
  int32_t nsplit;
  int64_t k_per_split;  // filled in by the launcher
  int64_t slab_stride;  // filled in by the launcher
};
// zero a [rows, width] column block of a row-major image of pitch ld (floats)
hipError_t launch_zero_cols(float* p, int64_t ld, int width, int64_t rows, hipStream_t stream);
// [n_seq*L, 3D] Q|K|V image: zero the Q columns of the masked token rows and all columns of the all-masked sequences
hipError_t launch_zero_dead_qkv(float* qkv, const float* mask, const int32_t* ids, int64_t n_seq, int L, int D, hipStream_t stream);
// Wt[c][r] = W[r][c] for a small row-major matrix W[rows][cols] (weights: a few MB)
hipError_t launch_transpose(const float* W, float* Wt, int rows, int cols, hipStream_t stream);
int gemm_pick_splits(int64_t M, int64_t N, int64_t K, bool dw_kernel = false);
int gemm_group_tiles(int n_tiles, int bn, int64_t K, bool plain);
size_t gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
hipError_t launch_gemm_f32(const GemmArgs& a, hipStream_t stream, int* nsplit_used = nullptr);
// dW = dY^T . X with the transpose done in registers on the way to LDS (gemm_dw.hip); nsplit / k_per_split / slab_stride
// as filled in by launch_gemm_f32
bool gemm_dw_eligible(const GemmArgs& a);
bool gemm_dw_big_tile(const GemmArgs& a);  // would launch_gemm_dw take the 256 x 256 tile?
hipError_t launch_gemm_dw(const GemmArgs& a, int nsplit, hipStream_t stream);
// forward-layout GEMM on the bf16 matrix cores by operand splitting (gemm_split.hip); npl = 3 or 2 planes
hipError_t launch_gemm_split(const GemmArgs& a, int npl, hipStream_t stream);
// planes[p][k/16][n][16] (p = 0..2: hi, mid, lo; k zero padded to ldp = split_plane_ld(K)) of W[N][K]: the
// 128 x 16 tile of one k step is contiguous
inline int64_t split_plane_ld(int64_t K) { return (K + 15) / 16 * 16; }
inline size_t split_planes_bytes(int64_t N, int64_t K) { return (size_t)3 * (size_t)N * (size_t)split_plane_ld(K) * 2; }
hipError_t launch_split_weights(const float* W, int64_t N, int64_t K, unsigned short* planes, hipStream_t stream);
// 0: exact fp32 MFMA (default); 1: bf16x3 split, six products; 2: bf16x2 split, three products.  Applies to the
// forward (ROW x WT) layout only.  PROCESS-GLOBAL (one relaxed atomic int, initialised from XNRS_GEMM_MODE when the
// library is loaded): every later launch from any thread uses the mode set last.
int gemm_mode();
void set_gemm_mode(int mode);

// ---------------------------------------------------------------- attention core (QK^T, row mask, softmax, PV)
struct MhaCoreArgs {
  const float* q;  // element (seq, head, s, j) at q[seq*seq_stride + head*head_stride + s*ld + j]
  const float* k;
  const float* v;
  int64_t ld;         // row stride (floats) of q/k/v
  int64_t seq_stride;   // row-major [rows, 3D] buffer: S*ld ; head-major buffer: n_heads*S*d_k
  int64_t head_stride;  // row-major: d_k ; head-major: S*d_k
  const float* mask;  // [n_seq*S] fp32 0/1 or null (QUERY-row mask, layers.py:142-144)
  const int32_t* mask_gather_ids;  // nullable: mask row of sequence n is ids[n]
  float* out;         // [n_seq*S, ldo]; head h writes columns [h*d_k, (h+1)*d_k)
  int64_t ldo;
  int64_t n_seq;
  int32_t S, n_heads, d_k;
  int32_t scaled;
  float dropout_p;
  uint64_t seed;
  const uint64_t* seed_dev;  // nullable: a device word ADDED to seed (xnrs_mha_params::seed_dev: a fresh dropout draw per hipGraph replay)
  float* stats;  // nullable: per (seq, head, query) softmax row statistics {max, sum} kept for the backward
  // unpadded queries (nullable): the queries of sequence n are the COMPACT rows q_off[n] .. q_off[n+1] of q (row
  // stride ldq, head h at column h*d_k) and `out` is compact the same way; every compact query is an unmasked row
  // (mask is ignored), K and V keep the padded [n_seq*S] addressing.  LDS-staged kernel only (S, d_k <= 64).
  const int64_t* q_off;
  int64_t ldq;
  // with q_off (nullable): K and V of sequence n are block kv_block[n] of the [.., S] row image (seq_stride apart) instead
  // of block n -- the device-compacted encoder projects K|V of the non-empty news into consecutive blocks (no row scatter)
  const int32_t* kv_block;
  // 1 (optional, with mask; pair kernel only, ignored elsewhere): a sequence whose query rows are ALL masked is not
  // computed -- its output rows are written as zeros and its statistics as those of masked rows {-1e9, S}.  For callers
  // whose consumers give masked rows a zero weight (the training forward over live rows, api.hip): such rows reach
  // neither the output nor a gradient, and with K = V = 0 (xnrs_row_lists) zeros are what the kernel would compute.
  int32_t skip_dead;
};
hipError_t launch_mha_core(const MhaCoreArgs& a, hipStream_t stream);

// backward of the attention core: recomputes P from Q, K and the saved row statistics
struct MhaBwdArgs {
  const float* q;
  const float* k;
  const float* v;
  int64_t ld;
  const float* mask;
  const int32_t* mask_gather_ids;
  const float* o;     // forward output (concat heads), [rows, ldo]
  int64_t ldo;
  const float* d_o;   // gradient w.r.t. the forward output, [rows, lddo]
  int64_t lddo;
  const float* stats; // from the forward
  float* delta;       // scratch [n_seq*n_heads*S]: rowsum(dO * O) per head
  float* dq;          // gradients, same addressing rule as q/k/v with row stride ldd
  float* dk;
  float* dv;
  int64_t ldd;
  int64_t n_seq;
  int32_t S, n_heads, d_k;
  int32_t scaled;
  float dropout_p;
  uint64_t seed;
  const uint64_t* seed_dev;   // as in MhaCoreArgs (the backward recomputes the forward's mask: same seed, same word)
  int32_t masked_do_is_zero;  // 1: the caller guarantees d_o == 0 on rows with mask == 0 (a masked pooler sits on top):
                              // query tiles whose 16 rows are all masked are skipped (their dQ rows are written as 0)
  // with masked_do_is_zero, fused kernel: a sequence whose query rows are ALL masked has dQ = dK = dV = 0 exactly.
  // 1: its workgroups write those zeros without reading anything; 2: they write nothing (the caller reads the gradient
  // rows of such sequences nowhere: live-row dWq, kv-row dWk / dWv, no input gradient).  0: no early exit.
  int32_t dead_seq_mode;
  // 1: dq / dk / dv hold the gradients of ANOTHER backward over the same Q|K|V image (the reference's second history
  // encode, training.py:406,409: same input and projection weights, another attention-dropout draw) and this launch ADDS
  // its own -- (dQKV_1 + dQKV_2)^T . X is then ONE weight-gradient product per projection.  Rows this launch would write
  // as zeros (masked query tiles, all-masked sequences) are left alone.
  int32_t accumulate;
};
hipError_t launch_mha_bwd(const MhaBwdArgs& a, hipStream_t stream);

// Softmax arithmetic shared by the attention forward and backward (they must agree bit for bit because
// the backward RECOMPUTES P from the saved row max / row sum).  The scale is a multiply by 1/sqrt(d_k) and
// the normalisation a multiply by 1/sum (one division per row instead of one per element), exp is the
// hardware exp2 path: <= ~2e-7 relative per element against torch's division/expf -- far inside the 1e-4
// parity bar -- and it takes the softmax from ~45 to ~10 VALU instructions per element (the attention
// kernel was VALU-bound, not MFMA-bound).
// (sequence, head) pair of workgroup L out of W = n_seq * n_heads, XCD-aware: the dispatcher deals workgroup L to XCD L % 8.
// The heads of ONE sequence go to one XCD, next to each other in time (a head row is 4 d_k bytes, so neighbouring heads
// share 128-B lines that two L2s would otherwise both fetch), while the SEQUENCES are dealt round-robin over the XCDs:
// work per sequence varies with the mask (the backward skips masked query tiles, unpadded queries are fewer), and a
// contiguous range of sequences per XCD -- all of one user's empty history slots -- left some XCDs 30 % more work.
__device__ __forceinline__ int64_t xcd_pair(int64_t L, int64_t W, int n_heads) {
  const int64_t n_seq = W / n_heads;
  const int64_t full = (n_seq >> 3) << 3;           // sequences dealt in whole rounds of 8
  const int64_t per = (full >> 3) * n_heads;        // workgroups per XCD that serve them
  const int64_t x = L & 7, q = L >> 3;
  if (q < per) return ((q / n_heads) * 8 + x) * n_heads + (q % n_heads);
  return full * n_heads + (q - per) * 8 + x;       // the last n_seq % 8 sequences: plain round-robin
}

__device__ __forceinline__ float attn_exp(float x) { return __expf(x); }

// tanh for the activation epilogues (the additive pooler's tanh(fc1 x), layers.py:60; a Tanh head): the hardware exp2 and
// reciprocal, 1 - 2 / (exp(2x) + 1), instead of ocml's tanhf (6 instead of ~60 VALU instructions per element with both
// branches of the ocml version executed).  ABSOLUTE error <= ~1.5e-7 everywhere (relative <= ~1.2e-6 for |x| >= 1/8;
// for smaller arguments the relative error grows as the result shrinks -- what the value feeds, the fc2 dot and
// 1 - t^2 in the backward, only sees the absolute error; a version that switched to the odd Taylor polynomial below 1/8
// cost 6 more instructions per element, a third of the fused encoder's epilogue, and moved no test).  Saturates to
// +-1, propagates NaN.  Parity bar 1e-4; XNRS_FAST_TANH=0 = tanhf.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x)
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}
// The additive pooler's score, 32 hidden units at a time, from a TRANSPOSED accumulator block (the fc1 product taken as
// W1 . X^T: v_mfma_f32_32x32x2_f32 with the operands swapped -- every element is the same fmaf chain as in X . W1^T):
// lane l holds token row (l & 31) and the 16 hidden units 8 g + 4 (l >> 5) + r of the block (e = 4 g + r), so
//   sum_h w2[h] tanh(acc[h] + b1[h])
// is an IN-LANE chain of 16 fmaf plus ONE exchange with lane l ^ 32 (a + b on one side, b + a on the other: the same
// bits).  The first version kept tokens on the accumulator rows and reduced over the 32 lanes that held a row's columns:
// five ds_bpermute_b32 per ELEMENT (80 per block instead of 1) on the LDS pipe -- the whole difference between the
// Q/K/V projection's 141 TF and the fc1 stage's 122 TF, and 80 of 250 us per tile in the fused additive encoder.
// gemm_f32.hip (RDOT) and additive_fused.hip both call this, which is what keeps them bit-identical.
// bw: {b1[h], w2[h]} of the lane's 16 hidden units (zeros past A), see rowdot_load_bw.
template <bool FAST>
__device__ __forceinline__ float rowdot_block_t(const f32x16& acc, const float2 (&bw)[16]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float pre = acc[e] + bw[e].x;
    s = fmaf(FAST ? fast_tanh(pre) : tanhf(pre), bw[e].y, s);
  }
  return s + __shfl_xor(s, 32);
}
// the same chain with {b1, w2} read from LDS as it goes (8 registers instead of 32: the 128-register GEMM epilogue)
template <bool FAST>
__device__ __forceinline__ float rowdot_block_t_lds(const f32x16& acc, const float2* __restrict__ src, int half) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float2 c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = src[8 * g + 4 * half + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre = acc[4 * g + r] + c[r].x;
      s = fmaf(FAST ? fast_tanh(pre) : tanhf(pre), c[r].y, s);
    }
  }
  return s + __shfl_xor(s, 32);
}
// {b1, w2} of the 16 hidden units a lane holds in a block (element e = 4 g + r <-> unit 8 g + 4 half + r), from the block's
// 32 interleaved pairs in LDS; loaded ONCE per hidden block and reused for every token block of the wave
__device__ __forceinline__ void rowdot_load_bw(float2 (&bw)[16], const float2* __restrict__ src, int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) bw[4 * g + r] = src[8 * g + 4 * half + r];
}

// activation codes inside the kernels: XNRS_ACT_* (0 none, 1 relu, 2 tanh) plus 3 = tanh through fast_tanh; the launchers
// turn 2 into 3 when the knob is on
constexpr int ACT_TANH_FAST = 3;
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 3) return fast_tanh(v);
  if (act == 2) return tanhf(v);
  return v;
}

// Counter-based RNG for attention dropout (training mode only) -> uniform [0,1) per probability.  Parity with torch's CPU
// Philox stream is impossible (SURVEY.md section 7 "hard parts"); only the distribution matters (tests: expectation,
// determinism per seed, forward / backward mask consistency).
// the dropout seed of a launch: the host value plus, when given, a device word (wave-uniform scalar load)
template <class Args>
__device__ __forceinline__ uint64_t drop_seed(const Args& a) {
  return a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
}
// The dropout draw of probability (sequence, head, query, key): ONE splitmix64 of (seed, sequence * heads + head) -- invariant
// in every loop that calls this, so the compiler computes it once per wave -- and per probability a 32-bit finaliser (murmur3
// fmix32, a bijection) of its low word and the element's index inside the pair (query * S + key < 2^32), xor its high word.
// Round 4: the first version ran splitmix64 on a 64-bit element index per probability (two 64-bit multiplies + the 64-bit
// index arithmetic: ~0.2 ms of the 8.4 ms NRMS grad step, tools/bench_dropout_cost.py).  Forward and backward call this
// with the same arguments, so the mask is recomputed, never stored.
template <class Args>
__device__ __forceinline__ float drop_uniform(const Args& a, int64_t seq, int hd, int S, int query, int key) {
  uint64_t z = drop_seed(a) + 0x9E3779B97F4A7C15ull * ((uint64_t)(seq * a.n_heads + hd) + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  uint32_t x = (uint32_t)z ^ ((uint32_t)(query * S + key) * 0x9E3779B9u);
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  x ^= (uint32_t)(z >> 32);
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- fused news encoder, S <= 32 tokens (news_fused.hip)
// att -> additive pooler of TextEncoder.forward in ONE launch (news_encoding.py:48-54, layers.py:60-65,128-154)
struct NewsFusedArgs {
  const float* x;      // [n_seq, S, D] token rows, or the table when ids != null
  const int32_t* ids;  // nullable: news n is table row ids[n] (x and mask are then the table's)
  const float* mask;   // [n_seq, S] fp32 0/1 (or table mask), nullable
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;  // nn.Linear layout [out][in]; biases nullable
  const float *w1, *b1, *w2, *b2;                        // fc1 [A][D], fc2 [A]
  float* img;          // workspace of NewsFusedPlan::img_bytes: the weights in MFMA fragment order (built per call)
  float* p;            // [n_seq, ldp] pooled vectors
  int64_t ldp;
  float* hm;           // nullable [n_seq]: clamp(sum mask, 0, 1)
  int64_t n_seq;
  int32_t S, D, n_heads, d_k, A, scaled;
  int32_t npw;         // news per workgroup: 2 (default, 0 means 2) or 1
  int32_t tanh_act;    // filled in by the launcher: 2 = ocml tanhf, 3 = fast_tanh (knob)
  // fold (api.hip "fold": the out-projection behind the pooling): w1 / b1 are then the FOLDED fc1 (W1.Wo, W1.bo + b1), the
  // kernel skips the out-projection, pools the attention rows O and emits p = sum_i a_i O_i and asum = sum_i a_i; the
  // caller applies Wo once per news.  o_scratch: news_fused_scratch_bytes() of L2-resident scratch (the O tiles of a
  // workgroup's news are parked there while the token rows still occupy LDS).
  int32_t fold;
  float* o_scratch;
  float* asum;         // [n_seq] (fold only)
};
struct NewsFusedPlan {
  int npw, hg, lq, ly;  // news per workgroup, heads per group, LDS row strides of the Q|K|V and Y images
  int n_groups, nk, nkc;  // head groups, 16-wide k steps of D, of one group's hg * d_k attention-output columns
  size_t lds_bytes;
  size_t img_bytes;       // caller-provided workspace for the fragment-ordered weight images
};
// does the fused kernel cover this shape (and with which plan)?
bool news_fused_plan(int S, int D, int n_heads, int A, NewsFusedPlan* plan, int npw = 2);
size_t news_fused_img_bound_bytes(int S, int D, int A);  // >= img_bytes for every n_heads; 0: no head count is eligible
size_t news_fused_scratch_bytes(int S, int D);            // NewsFusedArgs::o_scratch (fold), whatever the batch size
// the launch's remaining preconditions (16-byte aligned operands, 160 KB of dynamic LDS granted on the CURRENT device);
// false = use the GEMM pipeline for this call
bool news_fused_ready(const NewsFusedArgs& a);
hipError_t launch_news_fused(const NewsFusedArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- additive encoder in one launch (additive_fused.hip)
// fc1 + tanh + fc2 + exp * mask + normalise + weighted sum of a TextEncoder WITHOUT self-attention (layers.py:60-65):
// persistent workgroups over tiles of whole news; bit-identical to the GEMM + additive_pool pipeline
struct AdditiveFusedArgs {
  const float* x;      // [n_seq, S, D] token rows, or the table when ids != null
  const int32_t* ids;  // nullable: news n is table row ids[n] (x and mask are then the table's)
  const float* mask;   // [n_seq, S] fp32 0/1 (or table mask), nullable
  const float *w1, *b1, *w2, *b2;  // fc1 [A][D], b1 [A] nullable, fc2 [A], b2 [1] nullable
  float* y;            // [n_seq, ldy] pooled vectors
  int64_t ldy;
  float* hm;           // nullable [n_seq]: clamp(sum mask, 0, 1)
  int64_t n_seq;
  int32_t S, D, A;
  int32_t tanh_act;    // filled in by the launcher
};
bool additive_fused_plan(int S, int D, int A, int* nn_out, int* pp_out);  // does the kernel cover the shape?
bool additive_fused_ready(const AdditiveFusedArgs& a);                     // ... and these operands (alignment)?
int64_t additive_fused_tiles(int64_t n_seq, int S);                        // 256-row tiles of whole news
hipError_t launch_additive_fused(const AdditiveFusedArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- pooling / scoring
struct AdditivePoolArgs {
  const float* t;    // [n_seq*N, A] = tanh(fc1(x)) (the tanh is fused into the fc1 GEMM epilogue); null with epart
  const float* epart;  // nullable, instead of t: [n_seq*N, n_epart] partial fc2 dots per 32-column block (GemmArgs::rowdot_out)
  int32_t n_epart;
  const float* w2;   // [A]
  const float* b2;   // [1]
  const float* mask; // [n_seq*N] or null
  const int32_t* mask_gather_ids;  // nullable: mask rows of sequence n come from table row ids[n]
  const int32_t* x_gather_ids;     // nullable: value rows of sequence n come from table row ids[n]
  const float* x;    // [n_seq*N, ldx] values to pool (or table)
  int64_t ldx;
  float* y;          // [n_seq, D]
  float* a_out;      // [n_seq*N] or null
  float* hm_out;     // [n_seq] or null : clamp(sum mask,0,1)
  float* asum_out;   // [n_seq] or null : sum_i a_i (1 - 1e-8/denominator, or 0 for an all-masked sequence)
  int64_t n_seq;
  int32_t N, D, A;
  // unpadded rows (nullable): sequence n owns the compact rows row_off[n] .. row_off[n+1] (<= N of them) of t and x
  // (x_gather_ids unused); every compact row is unmasked, `mask` is ignored and hm_out = (count > 0).
  const int64_t* row_off;
  const int32_t* row_ids;  // nullable, with row_off: value row of compact row j is x[row_ids[j]] (t stays compact)
  const int64_t* poison;   // nullable device flag: non-zero -> every output of the launch is NaN (a precondition the host
                           // could not check without a sync was violated: launch_compact_rows saw a non-0/1 mask)
};
hipError_t launch_additive_pool(const AdditivePoolArgs& a, hipStream_t stream);
// p[n][d] += s[n] * b[d]   (the out-projection bias behind a pooled out-projection, api.hip "fold")
// bf[a] = W1[a,:] . bo + b1[a]  (b1 nullable): the fc1 bias behind a folded out-projection
hipError_t launch_fold_bias(const float* w1, const float* bo, const float* b1, float* bf, int A, int D, hipStream_t stream);
hipError_t launch_add_rowscaled_bias(float* p, int64_t ld, const float* s, const float* b, int64_t n, int D, hipStream_t stream);

struct MeanPoolArgs {
  const float* x;
  int64_t ldx;
  const float* mask;  // [n_seq*N] (required)
  const int32_t* mask_gather_ids;
  const int32_t* x_gather_ids;
  float* y;           // [n_seq, D]
  float* hm_out;      // nullable
  int64_t n_seq;
  int32_t N, D;
};
hipError_t launch_mean_pool(const MeanPoolArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- backward of pooling / scoring
struct AdditivePoolBwdArgs {
  const float* dp;   // [n_seq, D] gradient of the pooled vector
  const float* x;    // value rows (or table)
  int64_t ldx;
  const int32_t* x_gather_ids;
  const float* a;    // [n_seq*N] attention weights from the forward
  const float* t;    // [n_seq*N, A] tanh(fc1 x) from the forward
  const float* w2;   // [A]
  float* dx;         // nullable [n_seq*N, lddx] = a_i * dp
  int64_t lddx;
  float* dpre;       // [n_seq*N, A] gradient at the fc1 pre-activation
  float* de;         // [n_seq*N] gradient at the fc2 output (score)
  const float* da_shift;  // nullable [n_seq]: added to every da_i of the sequence (folded out-projection: dp . bo)
  const float* shift_u;   // nullable pair: ... or that shift taken here as the dot product of row `seq` of shift_u [n_seq, D]
  const float* shift_v;   // with shift_v [D] (one launch less than a GEMV in front of this kernel)
  int64_t n_seq;
  int32_t N, D, A;
};
hipError_t launch_additive_pool_bwd(const AdditivePoolBwdArgs& a, hipStream_t stream);
hipError_t launch_mean_pool_bwd(const float* dy, const float* mask, const int32_t* mask_ids, float* dx, int64_t lddx,
                                int64_t n_seq, int32_t N, int32_t D, hipStream_t stream);
size_t colsum_workspace_bytes(int N);
hipError_t launch_colsum(const float* X, int64_t ldx, const float* w, int64_t M, int N, float* out, float* partial,
                         hipStream_t stream);
// the same over two row blocks stacked: out[n] = sum_m w[m] X[m][n] + sum_m w2[m] X2[m][n]  (M2 rows of pitch ldx2)
hipError_t launch_colsum2(const float* X, int64_t ldx, const float* w, int64_t M, const float* X2, int64_t ldx2, const float* w2,
                          int64_t M2, int N, float* out, float* partial, hipStream_t stream);
// ... and the sum of the row weights themselves beside it: out[n] = sum_r w[r] X[r][n], wsum_out[0] = sum_r w[r]
// (partial sized for N + 1 columns)
hipError_t launch_colsum_wsum(const float* X, int64_t ldx, const float* w, int64_t M, int N, float* out, float* wsum_out,
                              float* partial, hipStream_t stream);
// out[n] = sum_s partial[s][n] in a fixed order (second stage of launch_colsum; also reduces GemmArgs::colsum)
hipError_t launch_colsum_final(const float* partial, int nsplit, int N, float* out, hipStream_t stream);
hipError_t launch_dot_scoring_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B, int32_t C,
                                  int32_t E, hipStream_t stream);
hipError_t launch_dot_scoring_norm_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B,
                                       int32_t C, int32_t E, hipStream_t stream);

hipError_t launch_embedding_grad(const float* d_rows, const int32_t* ids, int64_t M, int K, float* d_table, int n_rows,
                                 hipStream_t stream);

hipError_t launch_collapse_mask(const float* m, const int32_t* gather_ids, float* hm, int64_t n_rows, int32_t S,
                                hipStream_t stream);
hipError_t launch_dot_scoring(const float* u, const float* c, float* r, int64_t B, int32_t C, int32_t E,
                              int32_t normalize, hipStream_t stream);

// ---------------------------------------------------------------- device-side batch assembly / evaluation
struct BatchArgs {
  const int64_t* sess;       // [B] session (impression) indices
  const int64_t* hist_off;   // CSR offsets [n_sess+1] and values (table rows) of the click histories
  const int32_t* hist_val;
  const int64_t* pos_off;
  const int32_t* pos_val;
  const int64_t* neg_off;
  const int32_t* neg_val;
  int64_t B;
  int32_t l_hist, n_neg;
  int32_t pad_row;           // table row of the empty slot (all-zero tokens and mask)
  uint64_t seed;
  int32_t* hist_out;         // [B, l_hist]
  int32_t* cand_out;         // train: [B, 1+n_neg]; eval: CSR values
  const int64_t* cand_off_out;  // eval: [B+1] offsets (given)
  int32_t* cand_sess_out;    // eval: [n_cand] impression index of every candidate
  float* targets_out;        // eval: [n_cand]
};
// out[i, :] = table[ids[i], :] for rows of row_floats floats (whole news blocks)
hipError_t launch_gather_rows(const float* table, const int32_t* ids, float* out, int64_t n, int64_t row_floats,
                              hipStream_t stream);
hipError_t launch_poison(float* y, int64_t n, const int64_t* flags, int n_flags, int flag_stride, hipStream_t stream);
// the caller's sticky device status word (xnrs_set_status_word; nullptr = none): kernels OR XNRS_STATUS_* bits into it when
// they meet a violated precondition that no host code could check without a synchronisation
int32_t* status_word();
void set_status_word(int32_t* w);
// live-row / kept-K|V-row lists + CSR offsets of every pass of `chunk` news, built on the device in one launch (batch.hip):
// pass p writes row_off[p*(chunk+1) ..], the lists at [p*chunk*S ..], kv_block[p*chunk ..], counts[3*p ..]
hipError_t launch_compact_rows(const float* mask, const int32_t* ids, int64_t n_news, int64_t chunk, int S, int64_t* row_off,
                               int32_t* live_src, int32_t* kv_src, int32_t* kv_block, int64_t* counts, hipStream_t stream);
// the grad step's row lists built on the device (batch.hip): live[<= n_seq*L] unmasked token rows, kv[<= n_seq*L] all token
// rows of the sequences with at least one unmasked token, both in row order; *_src (nullable, with ids): the same tokens'
// rows in the gathered table; counts[0] = live rows, counts[1] = kv rows; cnt_scratch: int32 [n_seq]
hipError_t launch_build_row_lists(const float* mask, const int32_t* ids, int64_t n_seq, int L, int32_t* live, int32_t* live_src,
                                  int32_t* kv, int32_t* kv_src, int64_t* counts, int32_t* cnt_scratch, hipStream_t stream);
hipError_t launch_assemble_train(const BatchArgs& a, hipStream_t stream);
hipError_t launch_assemble_eval(const BatchArgs& a, hipStream_t stream);
hipError_t launch_score_csr(const float* vecs, const int32_t* rows, const int32_t* sess, const float* u, float* r, int64_t n,
                            int E, int relu, hipStream_t stream);
hipError_t launch_rank_metrics(const float* score, const float* target, const int64_t* off, float* out, int64_t B,
                               hipStream_t stream);

// ---------------------------------------------------------------- in-batch InfoNCE (training.py:433-472)
// ws: (B*E + 4*B + 1) floats = normalised embeddings | 1/norm | num | den | L_i | 1/(count+1e-8); kept for the backward
hipError_t launch_infonce_fwd(const float* x, const int64_t* lab, int64_t B, int E, float temperature, float* loss, float* ws,
                              hipStream_t stream);
hipError_t launch_infonce_bwd(const int64_t* lab, int64_t B, int E, float temperature, const float* ws, const float* gout,
                              float* dx, hipStream_t stream);

}  // namespace xnrs
