// extern "C" entry points of libxnrs_hip.so (declared in include/xnrs_hip.h) and the host-side
// orchestration of the kernel pipeline.  No device allocation and no sync: everything is enqueued on the caller's
// stream into the caller's workspace (hipGraph-capturable) -- or, for the backward's weight gradients, on a side stream
// forked from and joined back into the caller's stream inside the call (SideLane below).  Process-global state, all of it here or in
// gemm_f32.hip and none of it touched by a plain encode/score call: the forward-GEMM arithmetic mode
// (xnrs_set_gemm_mode, an atomic int), the development knobs (read once at load, xnrs_reload_knobs) and the
// optional launch timer (xnrs_profile_*, mutex-guarded, off by default).
//
// Sequence-encoder pipeline (TextEncoder news_encoding.py:34-60 / UserEncoder user_encoding.py:50-81),
// per chunk of sequences:
//   [att]   QKV = x.[Wq|Wk|Wv]^T + b   (one 3-segment MFMA GEMM, optional id-gather on the rows)
//           O   = softmax(rowmask(QK^T/sqrt(dk))) V          (mha_core, per sequence/head/q-tile)
//           Y   = O.Wo^T + bo                                  (MFMA GEMM)
//   [pool]  T   = tanh(Y.W1^T + b1)                            (MFMA GEMM, tanh epilogue)
//           p   = sum_i a_i Y_i,  a = exp(T.w2+b2)*m / (sum+1e-8)   (additive_pool)  | masked mean
//   [head]  y   = W4 relu(W3 p + b3) + b4                      (two MFMA GEMMs over all sequences)
// Short sequences (L <= 32, D <= 320: BASELINE configs[1]) take [att] + [pool] as ONE launch (news_fused.hip).
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/xnrs_hip.h"
#include "kernels.h"

using namespace xnrs;

namespace {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

constexpr int FOLD_SPLITS = 8;  // K slices of the folded-weight product W1 . Wo (slabs: FOLD_SPLITS x A x D floats in the plan)

struct Plan {
  int64_t chunk;  // sequences per pass
  size_t off_qkv, off_o, off_y, off_t, off_p, off_h;
  size_t off_stats, off_a;  // training only: softmax row statistics, pooling weights
  size_t off_planes;        // bf16-split GEMM modes: pre-split weight planes (wq, wk, wv, wo, w1)
  size_t off_nf, off_nfo;   // fused short-sequence encoder: fragment-ordered weight images, O-row scratch (fold)
  size_t off_fw, off_fb, off_po, off_as, off_fsl;  // folded out-projection: W1.Wo, W1.bo + b1, pooled O rows, sum of weights, split-K slabs
  size_t total;
};

// workspace carve for one chunk; every region 256-B aligned
Plan make_plan(int64_t n_seq, int L, int D, int A, int E, bool att, bool additive, bool head, bool pooled, int64_t chunk,
               bool train = false, int n_heads = 0) {
  Plan p{};
  if (train) chunk = n_seq > 0 ? n_seq : 1;  // the saved activations of the whole batch live in one carve
  if (chunk <= 0) chunk = 65536 / L;  // <= 64k token rows per pass: 512 full 128-row GEMM tiles (whole rounds of workgroups)
  if (chunk > n_seq) chunk = n_seq;
  if (chunk < 1) chunk = 1;
  p.chunk = chunk;
  const size_t rows = (size_t)chunk * L;
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    size_t o = off;
    off += align_up(nfloat * sizeof(float));
    return o;
  };
  p.off_qkv = att ? take(rows * 3 * (size_t)D) : 0;
  p.off_o = att ? take(rows * (size_t)D) : 0;
  p.off_y = (att && pooled) ? take(rows * (size_t)D) : 0;  // att output when a pooler follows
  p.off_t = (pooled && additive) ? take(rows * (size_t)A) : 0;
  // pooled vectors / head hidden of ALL sequences: the head runs once after the chunk loop (two GEMMs over
  // n_seq rows instead of 2 x n_chunks launches of ~20 workgroups each)
  p.off_p = (pooled && head) ? take((size_t)n_seq * D) : 0;
  p.off_h = (pooled && head) ? take((size_t)n_seq * E) : 0;
  p.off_stats = (train && att) ? take((size_t)chunk * n_heads * L * 2) : 0;
  p.off_a = (train && additive) ? take(rows) : 0;
  // always reserved (15 MB at D = 768), so the plan does not depend on the GEMM mode of the moment
  size_t pl = 0;
  if (att) pl += 4 * align_up(split_planes_bytes(D, D));
  if (pooled && additive) pl += align_up(split_planes_bytes(A, D));
  p.off_planes = take((pl + 3) / 4);
  // fused short-sequence path (news_fused.hip): reserved whenever the shape is eligible, whatever the knobs say
  // (a size query does not know the head count: it reserves the bound over all of them)
  NewsFusedPlan nf{};
  size_t nfb = 0;
  if (att && additive && !train) {
    if (n_heads <= 0) nfb = news_fused_img_bound_bytes(L, D, A);
    else if (news_fused_plan(L, D, n_heads, A, &nf)) nfb = nf.img_bytes;
  }
  p.off_nf = nfb ? take((nfb + 3) / 4) : 0;
  p.off_nfo = nfb ? take(news_fused_scratch_bytes(L, D) / 4) : 0;  // its O rows while the out-projection is folded away
  // folded out-projection (seq_encode "fold"): reserved whenever the shape is eligible, whatever the knob says
  const bool foldable = att && additive;  // (training keeps W', b', the pooled O rows and the weight sums for the backward)
  p.off_fw = foldable ? take((size_t)A * D) : 0;
  p.off_fb = foldable ? take((size_t)A) : 0;
  p.off_po = foldable ? take((size_t)n_seq * D) : 0;
  p.off_as = foldable ? take((size_t)n_seq) : 0;
  p.off_fsl = foldable ? take((size_t)FOLD_SPLITS * A * D) : 0;
  p.total = off;
  return p;
}

int32_t hip_rc(hipError_t e) { return e == hipSuccess ? XNRS_OK : (int32_t)e; }

// ---- optional per-launch event timing (measurement aid; see xnrs_profile_enable in the header)
struct ProfRec {
  hipEvent_t beg, end;
  int stage;
  double flops;
};
uint32_t g_prof_mask = 0;  // 0 (default): ProfScope is a single load and compare
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;      // guards g_prof / g_prof_mask changes; taken only while the timer is on
constexpr size_t PROF_MAX = 1 << 16;

struct ProfScope {
  bool on;
  hipStream_t st;
  ProfRec r{};
  ProfScope(int stage, double flops, hipStream_t s) : on((g_prof_mask >> stage) & 1u), st(s) {
    if (!on) return;
    r.stage = stage;
    r.flops = flops;
    if (hipEventCreate(&r.beg) != hipSuccess || hipEventCreate(&r.end) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(r.beg, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.end, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() < PROF_MAX) g_prof.push_back(r);
    else {
      (void)hipEventDestroy(r.beg);
      (void)hipEventDestroy(r.end);
    }
  }
};

#define XNRS_TRY(expr)                    \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return hip_rc(_e); \
  } while (0)

// ---- side lane of the backward (round 4).  A backward call is two dependency chains: the input-gradient chain (dX products,
// pooling and attention backward -- the critical path) and the weight-gradient products hanging off it (dW GEMM + split-K
// reduction + bias sums: 3-4 launches of 7-25 us per parameter pair, most of them far too small to fill 256 CUs).  Issued on
// one stream they serialise: 230 launches under 30 us made up 2.1 of the 8.1 ms of the NRMS grad step.  The weight-gradient
// launches go to ONE library-owned stream per device instead, ordered behind their producers by events (fork) and joined
// back into the caller's stream before the entry point returns -- so for the caller the call is still "everything enqueued
// on my stream": what follows on that stream sees every result, workspaces may be reused right after the call, and a
// hipGraph capture of the caller's stream captures the fork / join as graph edges.  No host synchronisation.  Results are
// bitwise the same (the same launches, no atomics).  Off: XNRS_BWD_SIDE_STREAM=0, and while the launch timer is on (its stage
// times would overlap).
struct SideLane {
  hipStream_t side = nullptr;
  hipEvent_t ev[32] = {};
  unsigned next = 0;
  bool ok = false, tried = false;
};
constexpr int MAX_LANES = 64;
SideLane g_lanes[MAX_LANES];
std::mutex g_lane_mu;

SideLane* side_lane(hipStream_t main) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_LANES) return nullptr;
  SideLane& l = g_lanes[dev];
  std::lock_guard<std::mutex> lk(g_lane_mu);
  if (!l.tried) {
    // (never created under a stream capture -- resource creation is not a capturable call: a capture whose warm-up did not
    // run a backward simply keeps one stream)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    l.tried = true;
    bool good = hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; good && i < 32; ++i) good = hipEventCreateWithFlags(&l.ev[i], hipEventDisableTiming) == hipSuccess;
    l.ok = good;
  }
  return l.ok ? &l : nullptr;
}

class Fork {
 public:
  Fork(hipStream_t main, bool want) : main_(main) {
    if (want && knobs().bwd_side_stream && g_prof_mask == 0) lane_ = side_lane(main);
  }
  // the stream for work that depends on everything issued on the caller's stream SO FAR (the caller's stream itself when the
  // lane is off or an event call fails)
  hipStream_t after_main() {
    if (!lane_) return main_;
    hipEvent_t e = next_event();
    if (hipEventRecord(e, main_) != hipSuccess || hipStreamWaitEvent(lane_->side, e, 0) != hipSuccess) {
      join();
      lane_ = nullptr;
      return main_;
    }
    used_ = true;
    return lane_->side;
  }
  // the caller's stream waits for the lane (idempotent; the destructor calls it on every return path)
  void join() {
    if (!lane_ || !used_) return;
    hipEvent_t e = next_event();
    if (hipEventRecord(e, lane_->side) == hipSuccess) (void)hipStreamWaitEvent(main_, e, 0);
    used_ = false;
  }
  ~Fork() { join(); }
  Fork(const Fork&) = delete;
  Fork& operator=(const Fork&) = delete;

 private:
  hipEvent_t next_event() {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    return lane_->ev[lane_->next++ & 31u];
  }
  hipStream_t main_;
  SideLane* lane_ = nullptr;
  bool used_ = false;
};


// a row count for the launch timer's FLOP figure: the host value, or -- counts on the device, timer on for this stage --
// read back (the timer is a measurement aid that synchronises anyway; no read happens while it is off)
int64_t prof_count(int stage, const int64_t* cnt, int which, int64_t host_value, hipStream_t stream) {
  if (!cnt || !((g_prof_mask >> stage) & 1u)) return host_value;
  int64_t v = host_value;
  if (hipStreamSynchronize(stream) != hipSuccess) return host_value;
  if (hipMemcpy(&v, cnt + which, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return host_value;
  return v;
}

// can every product over a device-counted row list run on the kernels that read their row count on the device?
// (GemmArgs::m_dev: the fp32 buffer-load forward kernels; GemmArgs::k_dev: any dW kernel)
bool device_counts_ok(const float* x, int D, int A, const xnrs_mha_params* att, const xnrs_additive_params* pool) {
  auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!knobs().gemm_buf || gemm_mode() != 0 || D % 4 != 0 || D < 4 || (A > 0 && A % 4 != 0)) return false;
  if (!al16(x) || (pool && !al16(pool->w1))) return false;
  if (att && !(al16(att->wq) && al16(att->wk) && al16(att->wv) && al16(att->wo))) return false;
  return (int64_t)D * D * 4 <= (1ll << 30) && (int64_t)A * D * 4 <= (1ll << 30);
}

GemmArgs gemm1(const float* A, const int32_t* ids, int gS, int64_t lda, const float* W, const float* b, float* C,
               int64_t ldc, int64_t M, int N, int K, int act, const unsigned short* planes = nullptr) {
  GemmArgs g{};
  g.Wp[0] = planes;
  g.ldp = split_plane_ld(K);
  g.A = A;
  g.gather_ids = ids;
  g.gather_S = gS;
  g.lda = lda;
  g.W[0] = W;
  g.bias[0] = b;
  g.nseg = 1;
  g.Nseg = N;
  g.ldw = K;
  g.C = C;
  g.ldc = ldc;
  g.M = M;
  g.K = K;
  g.act = act;
  return g;
}

// ---- "fold": the out-projection behind the pooling (inference).
// The pooler never needs the attention OUTPUT rows Y_i = Wo O_i + bo one by one (layers.py:154 -> layers.py:60-65):
//   fc1(Y_i)           = W1 (Wo O_i + bo) + b1 = (W1 Wo) O_i + (W1 bo + b1)            -> scores straight from the O rows
//   sum_i a_i Y_i      = Wo (sum_i a_i O_i) + bo (sum_i a_i)                             -> ONE out-projection per sequence
// so the rows x D x D out-projection GEMM (12.4 of 57 ms of the benchmark step) becomes an n_seq x D x D one, plus an
// A x D x D product for the folded weight per call (0.3 GFLOP; the ABI keeps no state between calls).  Exact algebra for
// every input -- only the rounding order differs from the reference's (observed <= 2e-6 on the scores, bar 1e-4); the
// training forward keeps Y (the backward needs it).  XNRS_FOLD_OUT=0 keeps the per-token out-projection.
// The folded weight is rebuilt per call (the ABI keeps no state): a split-K product over 8 slices and a wave-per-row bias
// kernel, ~20 us per call at D = 768 -- three short launches.  One impression (1 250 + 250 token rows, three encoder calls)
// pays ~0.05 ms for that (a single unsliced product cost twice as much); from a few thousand token rows on the fold wins,
// +25 % at the benchmark batch.  The choice deliberately never depends on the batch size (only the short-title dispatch
// below does: fused kernel or pipeline by news count) -- a news item's vector must not change in the last bit with the batch
// it is encoded in (chunking, id gather, skip_empty and the padding-free path are all tested bitwise against the plain
// path).  Knob: 0 never, anything else always.
// Returns the fc1 bias to use (nullptr if there is none).
bool fold_wanted(int knob) { return knob != 0; }

// the fused row dots ride on the raw-buffer-load forward kernel: 16-byte aligned operands
bool fc1_rowdot_ok(const float* x, bool att, const xnrs_additive_params* pool, int D) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // (rows of any count / any gather: beyond the 1-GB descriptor window or with row ids the launcher takes the
  // pointer-gather variant of the same kernel)
  return knobs().fc1_rowdot && knobs().gemm_buf && gemm_mode() == 0 && pool && pool->w2 && D % 4 == 0 && D >= 4 &&
         al16(pool->w1) && (att || al16(x)) && (int64_t)pool->hidden * D * 4 <= (1ll << 30);
}


const float* fold_out_projection(const xnrs_mha_params* att, const xnrs_additive_params* pool, int D, int A, float* wf,
                                 float* bf, float* slabs, hipStream_t stream, hipError_t* err) {
  GemmArgs g{};  // wf[A][D] = W1[A][D] . Wo[D][D]   (B k-major: its row index is the contraction index)
  g.A = pool->w1;
  g.lda = D;
  g.W[0] = att->wo;
  g.b_kn = 1;
  g.ldw = D;
  g.nseg = 1;
  g.Nseg = D;
  g.C = wf;
  g.ldc = D;
  g.M = A;
  g.K = D;
  if (D >= 64 * FOLD_SPLITS) {  // enough contraction to slice: 8 x the workgroups, fixed-order reduce (bitwise reproducible)
    g.slabs = slabs;
    g.nsplit = FOLD_SPLITS;
  }
  *err = launch_gemm_f32(g, stream);
  if (*err != hipSuccess || !att->bo) return pool->b1;
  *err = launch_fold_bias(pool->w1, att->bo, pool->b1, bf, A, D, stream);  // bf = W1 . bo + b1
  return bf;
}

// dst[n] = Wo po[n] + bo s[n]: the out-projection behind the pooling.  fp32 GEMM mode: the bias term rides in the GEMM's
// epilogue (fmaf(s, bo, acc): the same bits as the separate pass, one launch less); split modes keep the separate pass.
hipError_t pooled_out_projection(const float* po, const float* s, const xnrs_mha_params* att, float* dst, int64_t n, int D,
                                 const unsigned short* planes, hipStream_t stream) {
  GemmArgs g = gemm1(po, nullptr, 0, D, att->wo, nullptr, dst, D, n, D, D, XNRS_ACT_NONE, planes);
  const bool in_epilogue = att->bo && gemm_mode() == 0;
  if (in_epilogue) {
    g.rowscale = s;
    g.rowscale_vec = att->bo;
  }
  hipError_t e = launch_gemm_f32(g, stream);
  if (e != hipSuccess || !att->bo || in_epilogue) return e;
  return launch_add_rowscaled_bias(dst, D, s, att->bo, n, D, stream);
}

// The tail of a pooled encoder call: [out-projection behind the pooling] -> [head].  One definition for the padded, the
// host-compacted and the device-compacted pipelines, so that they stay bit for bit equal.
//   fold && head && head->w0_folded (inference, fp32 GEMM mode): the out-projection is folded INTO the head's first layer,
//     W0 (Wo po + bo s) + b0 = (W0 Wo) po + (W0 bo) s + b0 -- the caller's cached pair (xnrs_fold_head_weights) -- and the
//     n x D x D product disappears (0.31 of the 44.4 ms benchmark step, 13 of the 299 us of configs[1]); the rank-1 term
//     rides in the GEMM's epilogue like bo s did.  Exact algebra, another rounding order (~1e-6).
//   otherwise: pooled = Wo po + bo s, then the head's two layers as written (news_encoding.py:27-31).
int32_t pooled_tail(bool fold, const float* pob, const float* asum, const xnrs_mha_params* att, const xnrs_head_params* head,
                    float* pb, float* hb, float* y, int64_t n, int D, int E, const unsigned short* wo_planes, bool train,
                    hipStream_t stream) {
  const bool fold_head = fold && head && !train && head->w0_folded && gemm_mode() == 0 && (!att->bo || head->b0_rowvec);
  if (fold && !fold_head) {
    ProfScope ps(2, 2.0 * n * (double)D * D, stream);
    XNRS_TRY(pooled_out_projection(pob, asum, att, head ? pb : y, n, D, wo_planes, stream));
  }
  if (head) {
    ProfScope ps(5, 2.0 * n * ((double)D * E + (double)E * E), stream);
    GemmArgs g1 = gemm1(fold_head ? pob : pb, nullptr, 0, D, fold_head ? head->w0_folded : head->w0, head->b0, hb, E, n, E, D,
                        head->activation);
    if (fold_head && att->bo) {
      g1.rowscale = asum;
      g1.rowscale_vec = head->b0_rowvec;
    }
    XNRS_TRY(launch_gemm_f32(g1, stream));
    XNRS_TRY(launch_gemm_f32(gemm1(hb, nullptr, 0, E, head->w2, head->b2, y, E, n, E, E, XNRS_ACT_NONE), stream));
  }
  return XNRS_OK;
}

// x:(n_seq,L,D) [or table + ids], m:(n_seq,L) [or table mask] -> y
//   pooled == false: y:(n_seq,L,D) = att(x)            (MultiHeadAttention alone)
//   pooled == true : y:(n_seq,E')  = head(pool(att(x)))
int32_t seq_encode(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int L, int D,
                   const xnrs_mha_params* att, bool pooled, int pool_kind, const xnrs_additive_params* pool,
                   const xnrs_head_params* head, float* y, float* a_out, float* hm, int64_t chunk, void* ws,
                   size_t ws_bytes, hipStream_t stream, bool train = false, const xnrs_row_lists* rl = nullptr) {
  const xnrs_row_lists no_lists{};
  if (!rl) rl = &no_lists;
  const int32_t *live_rows = rl->live_rows, *live_src_rows = rl->live_src_rows, *kv_rows = rl->kv_rows,
                *kv_src_rows = rl->kv_src_rows;
  const float* qkv_shared = rl->qkv_shared;
  // counts on the device (xnrs_row_lists::counts_dev): the list lengths below are then CAPACITIES (every row), the
  // products over a list read their row count on the device (GemmArgs::m_dev)
  const int64_t* cnt = rl->counts_dev;
  const int64_t n_live = cnt ? n_seq * L : rl->n_live, n_kv = cnt ? n_seq * L : rl->n_kv;
  if (n_seq == 0) return XNRS_OK;
  if (n_seq < 0 || L <= 0 || D <= 0 || !x || !y) return XNRS_EINVAL;
  if (att) {
    if (att->n_heads <= 0 || !att->wq || !att->wk || !att->wv || !att->wo) return XNRS_EINVAL;
    if (D % att->n_heads != 0) return XNRS_EHEADS;
    if (L > 128) return XNRS_EUNSUPPORTED;
  }
  const bool additive = pooled && pool_kind == XNRS_POOL_ADDITIVE;
  if (pooled) {
    if (pool_kind != XNRS_POOL_ADDITIVE && pool_kind != XNRS_POOL_MEAN) return XNRS_EINVAL;
    if (additive && (!pool || !pool->w1 || !pool->w2 || pool->hidden <= 0)) return XNRS_EINVAL;
    if (pool_kind == XNRS_POOL_MEAN && !m) return XNRS_EINVAL;
    if (L > 512) return XNRS_EUNSUPPORTED;
    if (head && (!head->w0 || !head->w2 || head->out_features <= 0 || head->activation < 0 || head->activation > 2))
      return XNRS_EINVAL;
  }
  if (ids && !m && pooled && pool_kind == XNRS_POOL_MEAN) return XNRS_EINVAL;
  const int A = additive ? pool->hidden : 0;
  const int E = (pooled && head) ? head->out_features : D;
  const Plan p = make_plan(n_seq, L, D, A, E, att != nullptr, additive, pooled && head, pooled, chunk, train,
                           att ? att->n_heads : 0);
  if (p.total > ws_bytes || (p.total > 0 && !ws)) return XNRS_EWORKSPACE;
  char* w = static_cast<char*>(ws);
  float* stats = (train && att) ? reinterpret_cast<float*>(w + p.off_stats) : nullptr;
  float* a_save = (train && additive) ? reinterpret_cast<float*>(w + p.off_a) : nullptr;
  // (training) the Q|K|V image of another forward over the same input and weights: read it, project nothing (xnrs_row_lists)
  const bool qkv_given = train && att && qkv_shared;
  float* qkv = qkv_given ? const_cast<float*>(qkv_shared) : reinterpret_cast<float*>(w + p.off_qkv);
  float* o = reinterpret_cast<float*>(w + p.off_o);
  float* yb = reinterpret_cast<float*>(w + p.off_y);
  float* t = reinterpret_cast<float*>(w + p.off_t);
  float* pb = reinterpret_cast<float*>(w + p.off_p);
  float* hb = reinterpret_cast<float*>(w + p.off_h);

  // Training forward over the UNMASKED token rows (optional, exact): a masked token row has pooling weight exp(e) * 0, so
  // its query projection, its out-projection row and its fc1 row never reach the output or any gradient (K and V stay
  // dense: padded tokens are keys, layers.py:142-144).  Those three products run over the live rows in place (A rows
  // gathered, C rows scattered through the same list); the dead rows of Q, Y and T are ZEROED first, which keeps every
  // later consumer -- attention core, pooling, and the backward kernels that read the saved activations -- finite and
  // exactly as if the rows had been computed and then multiplied by the zero weight.
  // An attention-FREE additive tower (StandardRec, NAML's views) takes the same list for its one row-parallel product:
  // fc1 runs over the live token rows of x, T of the masked rows is zero.
  const bool live = train && additive && m && live_rows && (cnt || (n_live >= 0 && n_live < n_seq * L)) &&
                    !(ids && !live_src_rows);
  if (cnt && live && !device_counts_ok(x, D, A, att, pool)) return XNRS_EUNSUPPORTED;
  const int32_t* lvx = live ? (ids ? live_src_rows : live_rows) : nullptr;  // rows of x (table rows with ids)
  // ... and K|V over the token rows of the NON-EMPTY news only (kv_rows, optional, exact): the keys and values of a news
  // are read by that news' own queries alone, and an all-masked news has no live query, so its K and V rows (zeroed
  // here: its attention rows then come out as finite zeros) reach neither the output nor a gradient.
  const bool kvl = live && kv_rows && (cnt || (n_kv >= 0 && n_kv < n_seq * L)) && !(ids && !kv_src_rows);
  const int32_t* kvx = kvl ? (ids ? kv_src_rows : kv_rows) : nullptr;

  // Short sequences go through the fused kernel (below); everything else folds the out-projection behind the pooling.
  // The predicate covers EVERY precondition of the launch (shape, 16-byte aligned operands, 160 KB of dynamic LDS on the
  // current device: news_fused_ready), so a batch the kernel cannot take runs on the pipeline instead of failing.
  NewsFusedArgs f{};
  bool fused = att && additive && !train && !a_out && att->dropout_p == 0.f && gemm_mode() == 0 &&
               knobs().news_fused && news_fused_plan(L, D, att->n_heads, A, nullptr) &&
               (D / att->n_heads) * att->n_heads == D &&
               (knobs().news_fused == 2 || (L >= 26 && n_seq >= 192));
  if (fused) {
    f.x = x; f.ids = ids; f.mask = m;
    f.wq = att->wq; f.bq = att->bq; f.wk = att->wk; f.bk = att->bk; f.wv = att->wv; f.bv = att->bv;
    f.wo = att->wo; f.bo = att->bo;
    f.w1 = pool->w1; f.b1 = pool->b1; f.w2 = pool->w2; f.b2 = pool->b2;
    f.img = reinterpret_cast<float*>(w + p.off_nf);
    f.p = head ? pb : y;
    f.ldp = D;
    f.hm = m ? hm : nullptr;
    f.n_seq = n_seq;
    f.S = L; f.D = D; f.n_heads = att->n_heads; f.d_k = D / att->n_heads; f.A = A; f.scaled = att->scaled;
    f.npw = knobs().news_fused_npw ? knobs().news_fused_npw : (n_seq < 512 ? 1 : 2);
    if (fold_wanted(knobs().fold_out)) {  // the kernel pools the attention rows; Wo is applied once per news below
      f.fold = 1;
      f.o_scratch = reinterpret_cast<float*>(w + p.off_nfo);
      f.asum = reinterpret_cast<float*>(w + p.off_as);
      f.p = reinterpret_cast<float*>(w + p.off_po);
    }
    fused = p.off_nf != 0 && news_fused_ready(f);
  }
  const bool fold = att && additive && fold_wanted(train ? knobs().fold_train : knobs().fold_out);
  float* wf = reinterpret_cast<float*>(w + p.off_fw);
  float* bf = reinterpret_cast<float*>(w + p.off_fb);
  float* pob = reinterpret_cast<float*>(w + p.off_po);
  float* asum = reinterpret_cast<float*>(w + p.off_as);
  // inference, fp32 GEMM mode, 16-byte-aligned shapes the buffer-load kernel serves, no gathered rows: the pooler's fc2
  // dot is taken in the fc1 epilogue (GemmArgs::rowdot_out; the T region then holds A/32 partial dots per row)
  const int n_ep = (A + 31) / 32;
  // (every input path -- dense rows, id gather, padding-free -- takes it, so they stay bitwise equal)
  const bool rowdot = additive && !train && !fused && fc1_rowdot_ok(x, att != nullptr, pool, D);
  const float* fc1_w = additive ? pool->w1 : nullptr;
  const float* fc1_b = additive ? pool->b1 : nullptr;
  if (fold) {
    if (pool->w1_folded) {  // the caller's copy of the folded pair (xnrs_fold_weights): nothing to rebuild (training: the backward
                            // call is then given the same pair -- the saved blob's copy stays unwritten)
      fc1_w = pool->w1_folded;
      fc1_b = att->bo ? pool->b1_folded : pool->b1;
      if (att->bo && !pool->b1_folded) return XNRS_EINVAL;
    } else {
      hipError_t fe = hipSuccess;
      fc1_b = fold_out_projection(att, pool, D, A, wf, bf, reinterpret_cast<float*>(w + p.off_fsl), stream, &fe);
      XNRS_TRY(fe);
      fc1_w = wf;
    }
    if (fused) {  // the fused kernel's fc1 image is built from the folded pair
      f.w1 = fc1_w;
      f.b1 = fc1_b;
    }
  }

  // Additive-only towers (no self-attention: StandardRec / BaseRec / NAML / LSTUR news encoders) from a batch that fills
  // the chip: fc1 + tanh + fc2 + exp + mask + normalise + weighted sum as ONE persistent launch (additive_fused.hip).  Its
  // result equals the GEMM + pooling pipeline's bit for bit (same MFMA fragments and k order, same reduction orders), so
  // the choice may depend on the batch size without a news vector ever changing: from two 256-row tiles per CU on the
  // fused launch wins (tools/bench_af.py, settled clocks: 2 560 news x 50 x 768 -- two tiles per CU -- 0.447 vs 0.469 ms for
  // the pipeline; 12 800 news -- ten per CU -- 2.13 vs 2.33 ms, i.e. 0.98 of the plain fc1 GEMM of the same shape with the
  // pooling included); below that the pipeline's smaller tiles fill the chip better.
  bool afused = !att && additive && !train && !a_out && gemm_mode() == 0 && knobs().additive_fused &&
                additive_fused_plan(L, D, A, nullptr, nullptr) && rowdot &&
                (knobs().additive_fused == 2 || additive_fused_tiles(n_seq, L) >= 512);
  if (afused) {
    AdditiveFusedArgs af{};
    af.x = x; af.ids = ids; af.mask = m;
    af.w1 = pool->w1; af.b1 = pool->b1; af.w2 = pool->w2; af.b2 = pool->b2;
    af.y = head ? pb : y;
    af.ldy = D;
    af.hm = m ? hm : nullptr;
    af.n_seq = n_seq;
    af.S = L; af.D = D; af.A = A;
    afused = additive_fused_ready(af);
    if (afused) {
      const double fl = (double)n_seq * (2.0 * L * D * A + 2.0 * L * (A + D));
      ProfScope ps(3, fl, stream);
      XNRS_TRY(launch_additive_fused(af, stream));
    }
  }

  // bf16-split GEMM modes: split the weights ONCE per call (the chunk loop below reuses them ~20 times per step)
  const unsigned short *pq = nullptr, *pk = nullptr, *pv = nullptr, *po = nullptr, *p1 = nullptr;
  if (gemm_mode() != 0) {
    char* pw = w + p.off_planes;
    auto prep = [&](const float* W, int N, int K) -> const unsigned short* {
      unsigned short* dst = reinterpret_cast<unsigned short*>(pw);
      pw += align_up(split_planes_bytes(N, K));
      return launch_split_weights(W, N, K, dst, stream) == hipSuccess ? dst : nullptr;
    };
    if (att) {
      pq = prep(att->wq, D, D);
      pk = prep(att->wk, D, D);
      pv = prep(att->wv, D, D);
      po = prep(att->wo, D, D);
    }
    if (pooled && additive) p1 = prep(fc1_w, A, D);
  }

  // Short sequences: attention + additive pooling of ALL sequences in one launch (news_fused.hip); only the pooled
  // vectors leave the CU.  Inference only (nothing is saved for a backward), fp32 arithmetic only.
  // Dispatch (measured, tools/bench_news_fused.py at D = 320; profiles/r02_news_fused_dispatch_sweep.txt): a workgroup owns
  // news padded to 32 token rows each, so the kernel wins from ~26 tokens (<= 19 % padding) upwards and once there are
  // enough news to fill the CUs -- with 1 news per workgroup (two workgroups per CU) from ~200, with 2 news per workgroup
  // (every weight fragment feeds 4 row tiles) from 512: 256 x 30 tokens 116 vs 149 us for the pipeline, 512 x 30: 177 vs
  // 217 us, 1024 x 30: 325 vs 329 us.  From ~1500 news on the pipeline is ahead again since it folds the out-projection
  // behind the pooling (fold_out_projection above; the fused kernel computes it per token): 2048 x 30: 623 vs 592 us,
  // 28 160 x 30: 8.2 vs 7.1 ms; 64 x 30: 111 vs 104 us, 1024 x 20: 318 vs 255 us.  XNRS_NEWS_FUSED=2 forces the kernel
  // for every eligible shape (tests), 0 turns it off.
  if (fused) {
    const double fl = (double)n_seq * ((f.fold ? 6.0 : 8.0) * L * D * D + 4.0 * L * L * D + 2.0 * L * D * A + 2.0 * L * (A + D));
    ProfScope ps(6, fl, stream);
    XNRS_TRY(launch_news_fused(f, stream));
  }
  for (int64_t c0 = 0; !fused && !afused && c0 < n_seq; c0 += p.chunk) {
    const int64_t nc = (n_seq - c0 < p.chunk) ? (n_seq - c0) : p.chunk;
    const int64_t rows = nc * L;
    // this chunk's view of the inputs
    const int32_t* cids = ids ? ids + c0 : nullptr;
    const float* cx = ids ? x : x + c0 * (int64_t)L * D;      // table stays whole when gathering
    const float* cm = m ? (ids ? m : m + c0 * (int64_t)L) : nullptr;

    const float* seq = cx;            // what the pooler sees
    const int32_t* seq_ids = cids;    // gather for the pooler's value rows
    if (att) {
      GemmArgs g{};
      g.A = cx;
      g.gather_ids = cids;
      g.gather_S = L;
      g.lda = D;
      g.W[0] = att->wq; g.W[1] = att->wk; g.W[2] = att->wv;
      g.Wp[0] = pq; g.Wp[1] = pk; g.Wp[2] = pv;
      g.ldp = split_plane_ld(D);
      g.bias[0] = att->bq; g.bias[1] = att->bk; g.bias[2] = att->bv;
      g.nseg = 3;
      g.Nseg = D;
      g.ldw = D;
      g.C = qkv;
      g.ldc = 3 * (int64_t)D;
      g.M = rows;
      g.K = D;
      g.act = XNRS_ACT_NONE;
      const int dk = D / att->n_heads;
      if (qkv_given) {
        // nothing to project
      } else if (live) {  // K|V of every row (kvl: of the rows of the non-empty news), Q of the live rows only (dead rows = 0)
        const double nl = (double)prof_count(0, cnt, 0, n_live, stream), nkv = (double)prof_count(0, cnt, 1, n_kv, stream);
        ProfScope ps(0, 2.0 * (kvl ? nkv : rows) * 2.0 * D * D + 2.0 * nl * (double)D * D, stream);
        g.W[0] = att->wk; g.W[1] = att->wv; g.W[2] = nullptr;
        g.Wp[0] = pk; g.Wp[1] = pv; g.Wp[2] = nullptr;
        g.bias[0] = att->bk; g.bias[1] = att->bv; g.bias[2] = nullptr;
        g.nseg = 2;
        g.C = qkv + D;
        if (kvl) {
          XNRS_TRY(launch_zero_dead_qkv(qkv, cm, cids, nc, L, D, stream));
          g.gather_ids = kvx;
          g.gather_S = 1;
          g.c_scatter = 1;
          g.c_scatter_ids = kv_rows;
          g.M = n_kv;
          g.m_dev = cnt ? cnt + 1 : nullptr;
          g.m_fill_hint = 0.6f;
          if (n_kv > 0) XNRS_TRY(launch_gemm_f32(g, stream));
        } else {
          XNRS_TRY(launch_gemm_f32(g, stream));
          XNRS_TRY(launch_zero_cols(qkv, 3 * (int64_t)D, D, rows, stream));
        }
        if (n_live > 0) {
          GemmArgs q = gemm1(cx, lvx, 1, D, att->wq, att->bq, qkv, 3 * (int64_t)D, n_live, D, D, XNRS_ACT_NONE, pq);
          q.c_scatter = 1;
          q.c_scatter_ids = live_rows;
          q.m_dev = cnt;
          q.m_fill_hint = 0.4f;
          XNRS_TRY(launch_gemm_f32(q, stream));
        }
      } else {
        ProfScope ps(0, 2.0 * rows * 3.0 * D * D, stream);
        XNRS_TRY(launch_gemm_f32(g, stream));
      }

      MhaCoreArgs ma{};
      // Q/K/V stay a row-major (rows, 3D) image.  A head-major image (every (sequence, head) block one
      // contiguous S x d_k run) was measured: attention -4 %, but the projection's scattered 64-B stores
      // cost it +2 % -- a net loss at the shipped shape, so it was dropped.
      ma.q = qkv;
      ma.k = qkv + D;
      ma.v = qkv + 2 * (int64_t)D;
      ma.ld = 3 * (int64_t)D;
      ma.seq_stride = (int64_t)L * 3 * D;
      ma.head_stride = dk;
      ma.mask = cm;
      ma.mask_gather_ids = cids;
      ma.out = o;
      ma.ldo = D;
      ma.n_seq = nc;
      ma.S = L;
      ma.n_heads = att->n_heads;
      ma.d_k = dk;
      ma.scaled = att->scaled;
      ma.dropout_p = att->dropout_p;
      ma.seed = att->seed + (uint64_t)c0 * 0x9E3779B97F4A7C15ull;
      ma.seed_dev = att->seed_dev;
      ma.stats = stats;
      // an all-masked sequence: zeros instead of attention over keys nobody weights (kernels.h).  Training over row lists,
      // and (round 4) every POOLED call with a mask: both poolers multiply a masked row by exactly 0 (layers.py:33,62-65), so
      // the pooled vector is bit for bit the same whether such a row holds the uniform average of V or zeros -- the empty
      // history slots of the benchmark batch (49.5 % of its news) cost the attention core nothing.  MultiHeadAttention
      // alone (pooled == false) returns its masked rows to the caller and computes them.
      ma.skip_dead = (live || (pooled && cm && knobs().mha_skip_masked)) ? 1 : 0;
      {
        ProfScope ps(1, 4.0 * rows * (double)L * D, stream);
        XNRS_TRY(launch_mha_core(ma, stream));
      }

      float* dst = pooled ? yb : y + c0 * (int64_t)L * D;
      if (fold) {
        dst = o;  // the pooler works on the O rows (fold_out_projection); masked rows of O are finite and carry weight 0
      } else if (live) {
        ProfScope ps(2, 2.0 * (double)prof_count(2, cnt, 0, n_live, stream) * (double)D * D, stream);
        XNRS_TRY(hipMemsetAsync(dst, 0, (size_t)rows * D * sizeof(float), stream));
        if (n_live > 0) {
          GemmArgs og = gemm1(o, live_rows, 1, D, att->wo, att->bo, dst, D, n_live, D, D, XNRS_ACT_NONE, po);
          og.c_scatter = 1;
          og.m_dev = cnt;
          og.m_fill_hint = 0.4f;
          XNRS_TRY(launch_gemm_f32(og, stream));
        }
      } else {
        ProfScope ps(2, 2.0 * rows * (double)D * D, stream);
        XNRS_TRY(launch_gemm_f32(gemm1(o, nullptr, 0, D, att->wo, att->bo, dst, D, rows, D, D, XNRS_ACT_NONE, po), stream));
      }
      seq = dst;
      seq_ids = nullptr;
    }
    if (!pooled) continue;

    float* pooled_dst = (head ? pb : y) + c0 * (int64_t)D;
    float* hm_dst = hm ? hm + c0 : nullptr;
    if (additive) {
      if (live) {  // with attention seq is the dense attention output; without, the rows of x (table rows with ids: lvx)
        ProfScope ps(3, 2.0 * (double)prof_count(3, cnt, 0, n_live, stream) * (double)D * A, stream);
        XNRS_TRY(hipMemsetAsync(t, 0, (size_t)rows * A * sizeof(float), stream));
        if (n_live > 0) {
          GemmArgs fg = gemm1(seq, att ? live_rows : lvx, 1, D, fc1_w, fc1_b, t, A, n_live, A, D, XNRS_ACT_TANH, p1);
          fg.c_scatter = 1;
          fg.c_scatter_ids = live_rows;
          fg.m_dev = cnt;
          fg.m_fill_hint = 0.4f;
          XNRS_TRY(launch_gemm_f32(fg, stream));
        }
      } else {
        ProfScope ps(3, 2.0 * rows * (double)D * A, stream);
        GemmArgs fg = gemm1(seq, seq_ids, L, D, fc1_w, fc1_b, t, A, rows, A, D, XNRS_ACT_TANH, p1);
        if (rowdot) {  // the fc2 dot per 32 hidden columns straight from the epilogue: tanh(fc1 x) is never stored
          fg.rowdot_w = pool->w2;
          fg.rowdot_out = t;
          fg.ldrd = n_ep;
        }
        XNRS_TRY(launch_gemm_f32(fg, stream));
      }
      AdditivePoolArgs pa{};
      pa.t = rowdot ? nullptr : t;
      pa.epart = rowdot ? t : nullptr;
      pa.n_epart = n_ep;
      pa.w2 = pool->w2;
      pa.b2 = pool->b2;
      pa.mask = cm;
      pa.mask_gather_ids = cids;
      pa.x_gather_ids = seq_ids;
      pa.x = seq;
      pa.ldx = D;
      pa.y = fold ? pob + c0 * (int64_t)D : pooled_dst;
      pa.asum_out = fold ? asum + c0 : nullptr;
      pa.a_out = a_save ? a_save : (a_out ? a_out + c0 * (int64_t)L : nullptr);
      pa.hm_out = cm ? hm_dst : nullptr;
      pa.n_seq = nc;
      pa.N = L;
      pa.D = D;
      pa.A = A;
      {
        ProfScope ps(4, 2.0 * rows * (double)(A + D), stream);
        XNRS_TRY(launch_additive_pool(pa, stream));
      }
      if (a_save && a_out)
        XNRS_TRY(hipMemcpyAsync(a_out, a_save, (size_t)rows * sizeof(float), hipMemcpyDeviceToDevice, stream));
    } else {
      MeanPoolArgs mp{};
      mp.x = seq;
      mp.ldx = D;
      mp.mask = cm;
      mp.mask_gather_ids = cids;
      mp.x_gather_ids = seq_ids;
      mp.y = pooled_dst;
      mp.hm_out = hm_dst;
      mp.n_seq = nc;
      mp.N = L;
      mp.D = D;
      {
        ProfScope ps(4, 2.0 * rows * (double)D, stream);
        XNRS_TRY(launch_mean_pool(mp, stream));
      }
    }
  }
  // pooled = Wo (sum_i a_i O_i) + bo (sum_i a_i): one out-projection per sequence (or folded into the head), then the head
  return pooled_tail(fold, pob, asum, att, pooled ? head : nullptr, pb, hb, y, n_seq, D, E, po, train, stream);
}

}  // namespace

extern "C" {

int32_t xnrs_abi_version(void) { return XNRS_ABI_VERSION; }

#ifndef XNRS_BUILD_ID
#define XNRS_BUILD_ID "unknown"
#endif
const char* xnrs_build_id(void) { return XNRS_BUILD_ID; }

int32_t xnrs_set_status_word(int32_t* device_word) {
  set_status_word(device_word);
  return XNRS_OK;
}

const char* xnrs_status_string(int32_t word) {
  switch (word & 3) {
    case 0: return "ok";
    case XNRS_STATUS_NONBINARY_MASK: return "a mask value other than 0 / 1 reached the device-compacted encoder (its outputs are NaN)";
    case XNRS_STATUS_ROW_RANGE: return "a news-table row id outside the table (clamped; the gathered rows are wrong)";
    default: return "a mask value other than 0 / 1 reached the device-compacted encoder AND a table row id was out of range";
  }
}

size_t xnrs_row_lists_workspace_bytes(int64_t n_seq) { return n_seq > 0 ? align_up((size_t)n_seq * sizeof(int32_t)) : 0; }

int32_t xnrs_build_row_lists(const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t* live_rows,
                             int32_t* live_src_rows, int32_t* kv_rows, int32_t* kv_src_rows, int64_t* counts, void* ws,
                             size_t ws_bytes, void* stream) {
  if (n_seq < 0 || L <= 0 || !m || !live_rows || !kv_rows || !counts) return XNRS_EINVAL;
  if (ids && (!live_src_rows || !kv_src_rows)) return XNRS_EINVAL;  // a gathered table needs the tokens' table rows
  if (n_seq * (int64_t)L > 0x7fffffffLL) return XNRS_EUNSUPPORTED;   // int32 row indices
  if (n_seq == 0) return hip_rc(hipMemsetAsync(counts, 0, 2 * sizeof(int64_t), (hipStream_t)stream));
  if (!ws || ws_bytes < xnrs_row_lists_workspace_bytes(n_seq)) return XNRS_EWORKSPACE;
  return hip_rc(launch_build_row_lists(m, ids, n_seq, L, live_rows, ids ? live_src_rows : nullptr, kv_rows,
                                       ids ? kv_src_rows : nullptr, counts, static_cast<int32_t*>(ws), (hipStream_t)stream));
}

const char* xnrs_error_string(int32_t code) {
  switch (code) {
    case XNRS_OK: return "ok";
    case XNRS_EINVAL: return "invalid argument (shape or NULL pointer)";
    case XNRS_EHEADS: return "d_model is not divisible by n_heads";
    case XNRS_EWORKSPACE: return "workspace too small";
    case XNRS_EUNSUPPORTED: return "shape outside the supported range (attention S <= 128, pooling N <= 512)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int32_t xnrs_linear_fwd(const float* x, const int32_t* gather_ids, int32_t gather_S, const float* w, const float* bias,
                        float* y, int64_t M, int32_t N, int32_t K, int32_t act, void* stream) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return XNRS_EINVAL;
  if (gather_ids && gather_S <= 0) return XNRS_EINVAL;
  return hip_rc(launch_gemm_f32(gemm1(x, gather_ids, gather_S, K, w, bias, y, N, M, N, K, act), (hipStream_t)stream));
}

size_t xnrs_mha_workspace_bytes(int64_t B, int32_t S, int32_t D) {
  return make_plan(B, S, D, 0, D, true, false, false, false, 0).total;
}

int32_t xnrs_mha_fwd(const float* x, const float* m, const xnrs_mha_params* p, float* y, int64_t B, int32_t S, int32_t D,
                     void* ws, size_t ws_bytes, void* stream) {
  if (!p) return XNRS_EINVAL;
  return seq_encode(x, m, nullptr, B, S, D, p, false, 0, nullptr, nullptr, y, nullptr, nullptr, 0, ws, ws_bytes,
                    (hipStream_t)stream);
}

size_t xnrs_additive_workspace_bytes(int64_t B, int32_t N, int32_t D, int32_t A) {
  return make_plan(B, N, D, A, D, false, true, false, true, 0).total;
}

int32_t xnrs_additive_attention_fwd(const float* x, const float* m, const xnrs_additive_params* p, float* y, float* a_out,
                                    int64_t B, int32_t N, int32_t D, void* ws, size_t ws_bytes, void* stream) {
  if (!p) return XNRS_EINVAL;
  return seq_encode(x, m, nullptr, B, N, D, nullptr, true, XNRS_POOL_ADDITIVE, p, nullptr, y, a_out, nullptr, 0, ws,
                    ws_bytes, (hipStream_t)stream);
}

int32_t xnrs_masked_mean_fwd(const float* x, const float* m, float* y, int64_t B, int32_t N, int32_t D, void* stream) {
  return seq_encode(x, m, nullptr, B, N, D, nullptr, true, XNRS_POOL_MEAN, nullptr, nullptr, y, nullptr, nullptr, 0,
                    nullptr, 0, (hipStream_t)stream);
}

int32_t xnrs_collapse_mask(const float* m, float* hm, int64_t n_rows, int32_t S, void* stream) {
  if (!m || !hm || n_rows < 0 || S <= 0) return XNRS_EINVAL;
  return hip_rc(launch_collapse_mask(m, nullptr, hm, n_rows, S, (hipStream_t)stream));
}

size_t xnrs_text_encoder_workspace_bytes(int64_t n_news, int32_t S, int32_t D, int32_t A, int32_t E, int32_t has_att,
                                         int32_t pool_kind, int32_t has_head, int64_t chunk) {
  return make_plan(n_news, S, D, A, E, has_att != 0, pool_kind == XNRS_POOL_ADDITIVE, has_head != 0, true, chunk).total;
}

int32_t xnrs_text_encoder_fwd(const float* x, const float* m, const int32_t* ids, int64_t n_news, int32_t S, int32_t D,
                              const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                              const xnrs_head_params* head, float* y, float* hm, int64_t chunk, void* ws, size_t ws_bytes,
                              void* stream) {
  if (n_news == 0) return XNRS_OK;
  if (!m) return XNRS_EINVAL;  // TextEncoder always receives a token mask (news_encoding.py:41-50)
  int32_t rc = seq_encode(x, m, ids, n_news, S, D, att, true, pool_kind, pool, head, y, nullptr, hm, chunk, ws, ws_bytes,
                          (hipStream_t)stream);
  return rc;
}

// ---- unpadded news encoder (inference): workspace carve, every region 256-B aligned
namespace {
struct UnpadPlan {
  size_t off_kv, off_q, off_o, off_y, off_t, off_p, off_h, off_fw, off_fb, off_po, off_as, off_fsl, total;
};
UnpadPlan make_unpad_plan(int64_t n_news, int64_t n_valid, int S, int D, int A, int E, bool att, bool head) {
  UnpadPlan p{};
  size_t cur = 0;
  auto take = [&](size_t floats) {
    const size_t o = cur;
    cur += (floats * sizeof(float) + 255) / 256 * 256;
    return o;
  };
  const size_t nv = (size_t)(n_valid > 0 ? n_valid : 1);
  p.off_kv = att ? take((size_t)n_news * S * 2 * D) : 0;
  p.off_q = att ? take(nv * D) : 0;
  p.off_o = att ? take(nv * D) : 0;
  p.off_y = att ? take(nv * D) : 0;
  p.off_t = take(nv * A);
  p.off_p = head ? take((size_t)n_news * D) : 0;
  p.off_h = head ? take((size_t)n_news * E) : 0;
  p.off_fw = att ? take((size_t)A * D) : 0;  // folded out-projection (seq_encode "fold")
  p.off_fb = att ? take((size_t)A) : 0;
  p.off_po = att ? take((size_t)n_news * D) : 0;
  p.off_as = att ? take((size_t)n_news) : 0;
  p.off_fsl = att ? take((size_t)FOLD_SPLITS * A * D) : 0;
  p.total = cur;
  return p;
}
}  // namespace

size_t xnrs_text_encoder_unpadded_workspace_bytes(int64_t n_news, int64_t n_valid, int32_t S, int32_t D, int32_t A,
                                                  int32_t E, int32_t has_att, int32_t has_head) {
  return make_unpad_plan(n_news, n_valid, S, D, A, E, has_att != 0, has_head != 0).total;
}

int32_t xnrs_text_encoder_fwd_unpadded(const float* x, const int32_t* ids, int64_t n_news, int32_t S, int32_t D,
                                       const int32_t* rows, const int64_t* row_off, int64_t n_valid,
                                       const xnrs_mha_params* att, const xnrs_additive_params* pool,
                                       const xnrs_head_params* head, float* y, float* hm, void* ws, size_t ws_bytes,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_news == 0) return XNRS_OK;
  if (!x || !row_off || !pool || !y || n_news < 0 || n_valid < 0 || S <= 0 || D <= 0 || (n_valid > 0 && !rows))
    return XNRS_EINVAL;
  if (S > 512) return XNRS_EUNSUPPORTED;
  const int A = pool->hidden, E = head ? head->out_features : D;
  if (att) {
    if (att->n_heads <= 0 || D % att->n_heads != 0) return XNRS_EHEADS;
    if (S > 64 || D / att->n_heads > 64 || (D / att->n_heads) % 4 != 0 || D % 4 != 0) return XNRS_EUNSUPPORTED;
  }
  const UnpadPlan p = make_unpad_plan(n_news, n_valid, S, D, A, E, att != nullptr, head != nullptr);
  if (p.total > 0 && (!ws || ws_bytes < p.total)) return XNRS_EWORKSPACE;
  char* w = static_cast<char*>(ws);
  float* kv = reinterpret_cast<float*>(w + p.off_kv);
  float* qc = reinterpret_cast<float*>(w + p.off_q);
  float* oc = reinterpret_cast<float*>(w + p.off_o);
  float* yc = reinterpret_cast<float*>(w + p.off_y);
  float* tc = reinterpret_cast<float*>(w + p.off_t);
  float* pb = reinterpret_cast<float*>(w + p.off_p);
  float* hb = reinterpret_cast<float*>(w + p.off_h);
  const int64_t rows_all = n_news * (int64_t)S;

  const float* vals = x;          // what the pooler weights: compact y rows, or x rows through `rows`
  const int32_t* val_ids = rows;
  // the same folded out-projection as the padded path (seq_encode "fold"), so the two stay bitwise equal
  const bool fold = att && fold_wanted(knobs().fold_out);
  float* wf = reinterpret_cast<float*>(w + p.off_fw);
  float* bf = reinterpret_cast<float*>(w + p.off_fb);
  float* pob = reinterpret_cast<float*>(w + p.off_po);
  float* asum = reinterpret_cast<float*>(w + p.off_as);
  const float* fc1_w = pool->w1;
  const float* fc1_b = pool->b1;
  if (fold) {
    if (pool->w1_folded) {
      fc1_w = pool->w1_folded;
      fc1_b = att->bo ? pool->b1_folded : pool->b1;
      if (att->bo && !pool->b1_folded) return XNRS_EINVAL;
    } else {
      hipError_t fe = hipSuccess;
      fc1_b = fold_out_projection(att, pool, D, A, wf, bf, reinterpret_cast<float*>(w + p.off_fsl), stream, &fe);
      XNRS_TRY(fe);
      fc1_w = wf;
    }
  }
  if (att) {
    const int dk = D / att->n_heads;
    {  // K and V of EVERY token row: padded tokens stay keys (QUERY-row mask, layers.py:142-144)
      GemmArgs g{};
      g.A = x;
      g.gather_ids = ids;
      g.gather_S = S;
      g.lda = D;
      g.W[0] = att->wk; g.W[1] = att->wv;
      g.bias[0] = att->bk; g.bias[1] = att->bv;
      g.nseg = 2;
      g.Nseg = D;
      g.ldw = D;
      g.C = kv;
      g.ldc = 2 * (int64_t)D;
      g.M = rows_all;
      g.K = D;
      ProfScope ps(0, 2.0 * rows_all * 2.0 * D * D + 2.0 * n_valid * (double)D * D, stream);
      XNRS_TRY(launch_gemm_f32(g, stream));
      // Q of the live rows only (row gather through `rows`)
      if (n_valid > 0)
        XNRS_TRY(launch_gemm_f32(gemm1(x, rows, 1, D, att->wq, att->bq, qc, D, n_valid, D, D, XNRS_ACT_NONE), stream));
    }
    if (n_valid > 0) {
      MhaCoreArgs ma{};
      ma.q = qc;
      ma.q_off = row_off;
      ma.ldq = D;
      ma.k = kv;
      ma.v = kv + D;
      ma.ld = 2 * (int64_t)D;
      ma.seq_stride = (int64_t)S * 2 * D;
      ma.head_stride = dk;
      ma.out = oc;
      ma.ldo = D;
      ma.n_seq = n_news;
      ma.S = S;
      ma.n_heads = att->n_heads;
      ma.d_k = dk;
      ma.scaled = att->scaled;
      {
        ProfScope ps(1, 4.0 * n_valid * (double)S * D, stream);
        XNRS_TRY(launch_mha_core(ma, stream));
      }
      if (!fold) {
        ProfScope ps(2, 2.0 * n_valid * (double)D * D, stream);
        XNRS_TRY(launch_gemm_f32(gemm1(oc, nullptr, 0, D, att->wo, att->bo, yc, D, n_valid, D, D, XNRS_ACT_NONE), stream));
      }
    }
    vals = fold ? oc : yc;
    val_ids = nullptr;
  }
  const int n_ep = (A + 31) / 32;
  const bool rowdot = fc1_rowdot_ok(x, att != nullptr, pool, D);  // as in seq_encode (bitwise equal paths)
  if (n_valid > 0) {
    ProfScope ps(3, 2.0 * n_valid * (double)D * A, stream);
    GemmArgs fg = gemm1(vals, val_ids, 1, D, fc1_w, fc1_b, tc, A, n_valid, A, D, XNRS_ACT_TANH);
    if (rowdot) {
      fg.rowdot_w = pool->w2;
      fg.rowdot_out = tc;
      fg.ldrd = n_ep;
    }
    XNRS_TRY(launch_gemm_f32(fg, stream));
  }
  AdditivePoolArgs pa{};
  pa.t = rowdot ? nullptr : tc;
  pa.epart = rowdot ? tc : nullptr;
  pa.n_epart = n_ep;
  pa.w2 = pool->w2;
  pa.b2 = pool->b2;
  pa.x = vals;
  pa.ldx = D;
  pa.row_off = row_off;
  pa.row_ids = val_ids;
  pa.y = fold ? pob : (head ? pb : y);
  pa.asum_out = fold ? asum : nullptr;
  pa.hm_out = hm;
  pa.n_seq = n_news;
  pa.N = S;
  pa.D = D;
  pa.A = A;
  {
    ProfScope ps(4, 2.0 * n_valid * (double)(A + D), stream);
    XNRS_TRY(launch_additive_pool(pa, stream));
  }
  return pooled_tail(fold, pob, asum, att, head, pb, hb, y, n_news, D, E, nullptr, false, stream);
}

// ---- the padding-free encoder with the row lists built ON THE DEVICE (no host sync: hipGraph-capturable)
namespace {
struct CompactPlan {
  int64_t chunk;
  size_t off_kv, off_q, off_o, off_t, off_roff, off_live, off_kvs, off_kvb, off_cnt, off_p, off_h, off_fw, off_fb, off_po,
      off_as, off_fsl, total;
};
CompactPlan make_compact_plan(int64_t n_news, int S, int D, int A, int E, bool att, bool head, int64_t chunk) {
  CompactPlan p{};
  // default pass: ~262 k token rows (3.2 GB of worst-case scratch at D = 768 -- sized for 288 GB of HBM).  Four times the
  // padded path's pass: the row counts are only known on the device, so every pass pays the latency of its five launches
  // even when most of its rows are dead (tools/bench_compact_chunk.py: 95 % empty news 6.1 -> 3.6 ms per 25 600 news,
  // 50 %: 20.4 -> 18.5 ms; beyond ~10 k news per pass the one-workgroup-per-pass compaction kernel becomes the cost)
  if (chunk <= 0) chunk = 262144 / S;
  if (chunk > n_news) chunk = n_news;
  if (chunk < 1) chunk = 1;
  p.chunk = chunk;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    const size_t o = cur;
    cur += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t rows = (size_t)chunk * S;  // worst case of a pass: every token live
  const int n_ep = (A + 31) / 32;
  p.off_kv = att ? take(rows * 2 * D * 4) : 0;
  p.off_q = att ? take(rows * D * 4) : 0;
  p.off_o = att ? take(rows * D * 4) : 0;
  p.off_t = take(rows * (size_t)n_ep * 4);
  const size_t passes = (size_t)((n_news + chunk - 1) / chunk);
  p.off_roff = take(passes * ((size_t)chunk + 1) * 8);  // the row lists of EVERY pass (one compaction launch per call)
  p.off_live = take(passes * rows * 4);
  p.off_kvs = take(passes * rows * 4);
  p.off_kvb = take(passes * (size_t)chunk * 4);
  p.off_cnt = take(passes * 3 * 8);  // {live rows, K|V rows, bad-mask flag} per pass
  p.off_p = head ? take((size_t)n_news * D * 4) : 0;
  p.off_h = head ? take((size_t)n_news * E * 4) : 0;
  p.off_fw = att ? take((size_t)A * D * 4) : 0;
  p.off_fb = att ? take((size_t)A * 4) : 0;
  p.off_po = att ? take((size_t)n_news * D * 4) : 0;
  p.off_as = att ? take((size_t)n_news * 4) : 0;
  p.off_fsl = att ? take((size_t)FOLD_SPLITS * A * D * 4) : 0;
  p.total = cur;
  return p;
}
}  // namespace

size_t xnrs_text_encoder_compact_workspace_bytes(int64_t n_news, int32_t S, int32_t D, int32_t A, int32_t E, int32_t has_att,
                                                 int32_t has_head, int64_t chunk) {
  return make_compact_plan(n_news, S, D, A, E, has_att != 0, has_head != 0, chunk).total;
}

int32_t xnrs_text_encoder_fwd_compact(const float* x, const float* m, const int32_t* ids, int64_t n_news, int32_t S, int32_t D,
                                      const xnrs_mha_params* att, const xnrs_additive_params* pool, const xnrs_head_params* head,
                                      float* y, float* hm, int64_t chunk, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_news == 0) return XNRS_OK;
  if (!x || !m || !pool || !pool->w1 || !pool->w2 || !y || n_news < 0 || S <= 0 || D <= 0 || pool->hidden <= 0) return XNRS_EINVAL;
  if (S > 512) return XNRS_EUNSUPPORTED;
  const int A = pool->hidden, E = head ? head->out_features : D;
  if (att) {
    if (att->n_heads <= 0 || D % att->n_heads != 0) return XNRS_EHEADS;
    if (S > 64 || D / att->n_heads > 64 || (D / att->n_heads) % 4 != 0 || D % 4 != 0) return XNRS_EUNSUPPORTED;
    if (att->dropout_p != 0.f) return XNRS_EUNSUPPORTED;  // inference only
  }
  // the device row counts ride on the fp32 forward kernel with the fc2 dot in its epilogue (what the padded and the
  // host-compacted paths run too: the three stay bitwise equal)
  if (gemm_mode() != 0 || !fc1_rowdot_ok(x, att != nullptr, pool, D)) return XNRS_EUNSUPPORTED;
  const CompactPlan p = make_compact_plan(n_news, S, D, A, E, att != nullptr, head != nullptr, chunk);
  if (!ws || ws_bytes < p.total) return XNRS_EWORKSPACE;
  char* w = static_cast<char*>(ws);
  float* kv = reinterpret_cast<float*>(w + p.off_kv);
  float* qc = reinterpret_cast<float*>(w + p.off_q);
  float* oc = reinterpret_cast<float*>(w + p.off_o);
  float* tc = reinterpret_cast<float*>(w + p.off_t);
  int64_t* roff0 = reinterpret_cast<int64_t*>(w + p.off_roff);
  int32_t* live0 = reinterpret_cast<int32_t*>(w + p.off_live);
  int32_t* kvs0 = reinterpret_cast<int32_t*>(w + p.off_kvs);
  int32_t* kvb0 = reinterpret_cast<int32_t*>(w + p.off_kvb);
  int64_t* cnt0 = reinterpret_cast<int64_t*>(w + p.off_cnt);
  float* pb = reinterpret_cast<float*>(w + p.off_p);
  float* hb = reinterpret_cast<float*>(w + p.off_h);
  float* wf = reinterpret_cast<float*>(w + p.off_fw);
  float* bf = reinterpret_cast<float*>(w + p.off_fb);
  float* pob = reinterpret_cast<float*>(w + p.off_po);
  float* asum = reinterpret_cast<float*>(w + p.off_as);
  const bool fold = att && fold_wanted(knobs().fold_out);
  const float* fc1_w = pool->w1;
  const float* fc1_b = pool->b1;
  if (fold) {
    if (pool->w1_folded) {
      fc1_w = pool->w1_folded;
      fc1_b = att->bo ? pool->b1_folded : pool->b1;
      if (att->bo && !pool->b1_folded) return XNRS_EINVAL;
    } else {
      hipError_t fe = hipSuccess;
      fc1_b = fold_out_projection(att, pool, D, A, wf, bf, reinterpret_cast<float*>(w + p.off_fsl), stream, &fe);
      XNRS_TRY(fe);
      fc1_w = wf;
    }
  }
  if (att && !fold) return XNRS_EUNSUPPORTED;  // (the per-token out-projection order: use the host-compacted entry point)
  const int n_ep = (A + 31) / 32;
  XNRS_TRY(launch_compact_rows(m, ids, n_news, p.chunk, S, roff0, live0, kvs0, kvb0, cnt0, stream));
  for (int64_t c0 = 0; c0 < n_news; c0 += p.chunk) {
    const int64_t nc = (n_news - c0 < p.chunk) ? (n_news - c0) : p.chunk;
    const int64_t rows = nc * S;  // worst case
    const int64_t pass = c0 / p.chunk;
    const int64_t* cnt = cnt0 + 3 * pass;
    const int64_t* roff = roff0 + pass * (p.chunk + 1);
    const int32_t* live = live0 + pass * p.chunk * S;
    const int32_t* kvs = kvs0 + pass * p.chunk * S;
    const int32_t* kvb = kvb0 + pass * p.chunk;
    const float* vals = x;           // what the pooler weights: compact O rows, or x rows through `live`
    const int32_t* val_ids = live;
    if (att) {
      const int dk = D / att->n_heads;
      {  // K and V of every token of the news that have a live token: rows gathered through the device list, written as
         // consecutive S-row blocks (the attention kernel finds a news' block through kv_block: no row scatter)
        GemmArgs g{};
        g.A = x;
        g.gather_ids = kvs;
        g.gather_S = 1;
        g.lda = D;
        g.W[0] = att->wk; g.W[1] = att->wv;
        g.bias[0] = att->bk; g.bias[1] = att->bv;
        g.nseg = 2;
        g.Nseg = D;
        g.ldw = D;
        g.C = kv;
        g.ldc = 2 * (int64_t)D;
        g.M = rows;
        g.m_dev = cnt + 1;
        g.K = D;
        ProfScope ps(0, 2.0 * rows * 3.0 * D * D, stream);  // (worst case: the row counts live on the device)
        XNRS_TRY(launch_gemm_f32(g, stream));
        GemmArgs q = gemm1(x, live, 1, D, att->wq, att->bq, qc, D, rows, D, D, XNRS_ACT_NONE);
        q.m_dev = cnt;
        XNRS_TRY(launch_gemm_f32(q, stream));
      }
      MhaCoreArgs ma{};
      ma.q = qc;
      ma.q_off = roff;
      ma.ldq = D;
      ma.kv_block = kvb;
      ma.k = kv;
      ma.v = kv + D;
      ma.ld = 2 * (int64_t)D;
      ma.seq_stride = (int64_t)S * 2 * D;
      ma.head_stride = dk;
      ma.out = oc;
      ma.ldo = D;
      ma.n_seq = nc;
      ma.S = S;
      ma.n_heads = att->n_heads;
      ma.d_k = dk;
      ma.scaled = att->scaled;
      {
        ProfScope ps(1, 4.0 * rows * (double)S * D, stream);
        XNRS_TRY(launch_mha_core(ma, stream));
      }
      vals = oc;
      val_ids = nullptr;
    }
    {
      ProfScope ps(3, 2.0 * rows * (double)D * A, stream);
      GemmArgs fg = gemm1(vals, val_ids, 1, D, fc1_w, fc1_b, tc, A, rows, A, D, XNRS_ACT_TANH);
      fg.rowdot_w = pool->w2;
      fg.rowdot_out = tc;
      fg.ldrd = n_ep;
      fg.m_dev = cnt;
      XNRS_TRY(launch_gemm_f32(fg, stream));
    }
    AdditivePoolArgs pa{};
    pa.epart = tc;
    pa.n_epart = n_ep;
    pa.w2 = pool->w2;
    pa.b2 = pool->b2;
    pa.x = vals;
    pa.ldx = D;
    pa.row_off = roff;
    pa.row_ids = val_ids;
    pa.poison = cnt + 2;  // a mask value other than 0 / 1: NaN out, not a silently different result
    pa.y = (fold ? pob : (head ? pb : y)) + c0 * (int64_t)D;
    pa.asum_out = fold ? asum + c0 : nullptr;
    pa.hm_out = hm ? hm + c0 : nullptr;
    pa.n_seq = nc;
    pa.N = S;
    pa.D = D;
    pa.A = A;
    {
      ProfScope ps(4, 2.0 * rows * (double)(A + D), stream);
      XNRS_TRY(launch_additive_pool(pa, stream));
    }
  }
  {
    const int32_t rc = pooled_tail(fold, pob, asum, att, head, pb, hb, y, n_news, D, E, nullptr, false, stream);
    if (rc != XNRS_OK) return rc;
  }
  // a mask value other than 0 / 1 in any pass: NaN over the whole result (a ReLU head would swallow a NaN fed in earlier)
  const int n_pass = (int)((n_news + p.chunk - 1) / p.chunk);
  XNRS_TRY(launch_poison(y, n_news * (int64_t)E, cnt0 + 2, n_pass, 3, stream));
  if (hm) XNRS_TRY(launch_poison(hm, n_news, cnt0 + 2, n_pass, 3, stream));
  return XNRS_OK;
}

size_t xnrs_user_encoder_workspace_bytes(int64_t B, int32_t H, int32_t E, int32_t A, int32_t has_att, int32_t pool_kind,
                                         int32_t has_head) {
  return make_plan(B, H, E, A, E, has_att != 0, pool_kind == XNRS_POOL_ADDITIVE, has_head != 0, true, 0).total;
}

int32_t xnrs_user_encoder_fwd(const float* x, const float* m, int64_t B, int32_t H, int32_t E, const xnrs_mha_params* att,
                              int32_t pool_kind, const xnrs_additive_params* pool, const xnrs_head_params* head, float* y,
                              float* a_out, void* ws, size_t ws_bytes, void* stream) {
  return seq_encode(x, m, nullptr, B, H, E, att, true, pool_kind, pool, head, y, a_out, nullptr, 0, ws, ws_bytes,
                    (hipStream_t)stream);
}

int32_t xnrs_set_gemm_mode(int32_t mode) {
  const int prev = xnrs::gemm_mode();
  xnrs::set_gemm_mode(mode);
  return prev;
}

int32_t xnrs_get_gemm_mode(void) { return xnrs::gemm_mode(); }

size_t xnrs_fold_weights_workspace_bytes(int32_t D, int32_t A) {
  return D > 0 && A > 0 ? align_up((size_t)FOLD_SPLITS * A * D * sizeof(float)) : 0;
}

int32_t xnrs_fold_weights(const xnrs_mha_params* att, const xnrs_additive_params* pool, int32_t D, float* w1f, float* b1f,
                          void* ws, size_t ws_bytes, void* stream) {
  if (!att || !pool || !att->wo || !pool->w1 || pool->hidden <= 0 || D <= 0 || !w1f || !b1f) return XNRS_EINVAL;
  if (ws_bytes < xnrs_fold_weights_workspace_bytes(D, pool->hidden) || !ws) return XNRS_EWORKSPACE;
  hipError_t fe = hipSuccess;
  const float* b = fold_out_projection(att, pool, D, pool->hidden, w1f, b1f, static_cast<float*>(ws), (hipStream_t)stream, &fe);
  XNRS_TRY(fe);
  if (b != b1f) {  // no out-projection bias: b1 as it is (or zeros)
    if (pool->b1) XNRS_TRY(hipMemcpyAsync(b1f, pool->b1, (size_t)pool->hidden * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else XNRS_TRY(hipMemsetAsync(b1f, 0, (size_t)pool->hidden * sizeof(float), (hipStream_t)stream));
  }
  return XNRS_OK;
}

size_t xnrs_fold_head_weights_workspace_bytes(int32_t D, int32_t E) {
  return (D > 0 && E > 0) ? (size_t)FOLD_SPLITS * (size_t)E * (size_t)D * sizeof(float) : 0;
}

int32_t xnrs_fold_head_weights(const xnrs_mha_params* att, const xnrs_head_params* head, int32_t D, float* w0f, float* b0v,
                               void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!att || !head || !att->wo || !head->w0 || !w0f || D <= 0 || head->out_features <= 0) return XNRS_EINVAL;
  if (att->bo && !b0v) return XNRS_EINVAL;
  const int E = head->out_features;
  if (ws_bytes < xnrs_fold_head_weights_workspace_bytes(D, E) || !ws) return XNRS_EWORKSPACE;
  GemmArgs g{};  // w0f[E][D] = W0[E][D] . Wo[D][D]   (B k-major: its row index is the contraction index)
  g.A = head->w0;
  g.lda = D;
  g.W[0] = att->wo;
  g.b_kn = 1;
  g.ldw = D;
  g.nseg = 1;
  g.Nseg = D;
  g.C = w0f;
  g.ldc = D;
  g.M = E;
  g.K = D;
  if (D >= 64 * FOLD_SPLITS) {
    g.slabs = static_cast<float*>(ws);
    g.nsplit = FOLD_SPLITS;
  }
  XNRS_TRY(launch_gemm_f32(g, stream));
  if (att->bo) XNRS_TRY(launch_fold_bias(head->w0, att->bo, nullptr, b0v, E, D, stream));  // b0v = W0 . bo
  return XNRS_OK;
}

int32_t xnrs_train_fold_enabled(void) { return fold_wanted(knobs().fold_train) ? 1 : 0; }

int32_t xnrs_reload_knobs(void) {
  xnrs::reload_knobs();
  return XNRS_OK;
}

int32_t xnrs_profile_enable(uint32_t stage_mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
  g_prof_mask = stage_mask;
  return XNRS_OK;
}

int32_t xnrs_profile_read(double* ms, int64_t* launches, double* flops) {
  if (!ms || !launches || !flops) return XNRS_EINVAL;
  for (int i = 0; i < XNRS_PROFILE_STAGES; ++i) {
    ms[i] = 0.0;
    launches[i] = 0;
    flops[i] = 0.0;
  }
  int32_t rc = XNRS_OK;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(r.end);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.beg, r.end);
    if (e != hipSuccess) rc = (int32_t)e;
    else {
      ms[r.stage] += t;
      launches[r.stage] += 1;
      flops[r.stage] += r.flops;
    }
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
  return rc;
}

int32_t xnrs_dot_scoring_fwd(const float* u, const float* c, float* r, int64_t B, int32_t C, int32_t E, int32_t normalize,
                             void* stream) {
  if (!u || !c || !r || B < 0 || C <= 0 || E <= 0) return XNRS_EINVAL;
  return hip_rc(launch_dot_scoring(u, c, r, B, C, E, normalize, (hipStream_t)stream));
}

}  // extern "C"

// =================================================================================================
// training: forward with saved activations, backward
// =================================================================================================
namespace {

struct BwdPlan {
  size_t off_dh, off_dp, off_dseq, off_dpre, off_de, off_docat, off_dqkv, off_delta, off_slabs, off_colsum, off_wt;
  // folded out-projection (training): g = dp.Wo, c = dp.bo, dW', db'
  size_t off_g, off_c, off_dwf, off_dbf;
  size_t total;
};

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

BwdPlan make_bwd_plan(int64_t n_seq, int L, int D, int A, int E, int n_heads, bool additive, bool head, bool pooled) {
  BwdPlan p{};
  const bool att = n_heads > 0;
  const size_t rows = (size_t)n_seq * L;
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    size_t o = off;
    off += align_up(nbytes);
    return o;
  };
  p.off_dh = (pooled && head) ? take((size_t)n_seq * E * 4) : 0;
  p.off_dp = (pooled && head) ? take((size_t)n_seq * D * 4) : 0;
  p.off_dseq = (pooled) ? take(rows * D * 4) : 0;
  p.off_dpre = additive ? take(rows * A * 4) : 0;
  p.off_de = additive ? take(rows * 4) : 0;
  p.off_docat = att ? take(rows * D * 4) : 0;
  p.off_dqkv = att ? take(rows * 3 * D * 4) : 0;
  p.off_delta = att ? take((size_t)n_seq * n_heads * L * 4) : 0;
  const bool foldable = att && additive;
  p.off_g = foldable ? take((size_t)n_seq * D * 4) : 0;
  p.off_c = foldable ? take((size_t)n_seq * 4) : 0;
  p.off_dwf = foldable ? take((size_t)A * D * 4) : 0;
  p.off_dbf = foldable ? take((size_t)A * 4) : 0;
  // split-K slabs: the largest dW this pipeline produces
  size_t slabs = 0;
  if (att) slabs = max_sz(slabs, gemm_splitk_workspace_bytes(2 * (int64_t)D, D, rows));  // (dWk | dWv as one product)
  if (additive) slabs = max_sz(slabs, gemm_splitk_workspace_bytes(A, D, rows));
  if (foldable) slabs = max_sz(slabs, gemm_splitk_workspace_bytes(D, D, (int64_t)n_seq));  // dWo = dp^T po (+ W1^T dW')
  if (pooled && head) {
    slabs = max_sz(slabs, gemm_splitk_workspace_bytes(E, D, n_seq));
    slabs = max_sz(slabs, gemm_splitk_workspace_bytes(E, E, n_seq));
  }
  p.off_slabs = take(slabs);
  int maxn = 3 * D;
  if (A > maxn) maxn = A;
  if (E > maxn) maxn = E;
  p.off_colsum = take(colsum_workspace_bytes(maxn + 1));  // (+ 1: launch_colsum_wsum's column of ones)
  // one transposed weight at a time (gemm_dx): the largest of D x D, A x D, E x D, E x E
  size_t wdim = (size_t)D;
  if ((size_t)A > wdim) wdim = (size_t)A;
  if ((size_t)E > wdim) wdim = (size_t)E;
  p.off_wt = take(wdim * wdim * 4);
  p.total = off;
  return p;
}

// dW[N,K] = dY^T[N,M] . X[M,K]   (A k-major = dY, B k-major = X), split-K over the M rows.
// live (optional): contract over the n_live rows live_dy[j] of dY and live_x[j] of X only (the other rows of dY are
// known to be zero: masked token rows in the backward).
// db (optional, with csum = colsum workspace): the bias gradient db[N] = sum_rows dY, produced by the same launch (the
// kernel adds up the dY chunks it stages; a separate column-sum pass re-read every dY from HBM: 8.6 % of the train step)
hipError_t gemm_dw(const float* dY, int64_t lddy, const float* X, const int32_t* x_ids, int x_S, int64_t ldx, float* dW,
                   int64_t M, int N, int K, float* slabs, hipStream_t stream, const int32_t* live_dy = nullptr,
                   const int32_t* live_x = nullptr, int64_t n_live = 0, float* db = nullptr, float* csum = nullptr,
                   const int64_t* k_dev = nullptr, int accumulate = 0, float* dW2 = nullptr, float* db2 = nullptr, int n1 = 0) {
  // dW2 / db2 / n1 (optional): the product covers TWO parameters -- output rows [0, n1) are dW / db, rows [n1, N) are dW2 /
  // db2 (dY columns side by side, the same contraction rows): one launch instead of two (needs split-K; else two calls)
  const int64_t M_all = M;
  GemmArgs g{};
  g.A = dY;
  g.a_col = 1;
  g.lda = lddy;
  g.W[0] = X;
  g.b_kn = 1;
  g.b_gather_ids = x_ids;
  g.b_gather_S = x_S;
  if (live_dy) {
    g.gather_ids = live_dy;
    g.gather_S = 1;
    g.b_gather_ids = live_x;
    g.b_gather_S = 1;
    M = n_live;
    if (M <= 0) {
      if (db) {
        hipError_t e0 = hipMemsetAsync(db, 0, (size_t)N * sizeof(float), stream);
        if (e0 != hipSuccess) return e0;
      }
      return hipMemsetAsync(dW, 0, (size_t)N * K * sizeof(float), stream);
    }
  }
  g.ldw = ldx;
  g.nseg = 1;
  g.Nseg = K;
  g.C = dW;
  g.ldc = K;
  g.M = N;
  g.K = M;
  g.k_dev = live_dy ? k_dev : nullptr;  // the list's length on the device (M is then its capacity)
  g.accumulate = accumulate;
  const int ns = gemm_pick_splits(N, K, M, knobs().gemm_dw && gemm_dw_eligible(g));
  if (ns > 1) {
    g.slabs = slabs;
    g.nsplit = ns;
  }
  const bool fuse_db = db && csum && (lddy % 4 == 0) && (N % 4 == 0);  // the k-major vector path stages dY as 16-byte chunks
  if (dW2) {
    const int64_t per = ((M + ns - 1) / ns + 31) / 32 * 32;  // (the launcher's slice rule: is there really more than one?)
    if (ns <= 1 || (M + per - 1) / per <= 1 || (db && !fuse_db) || (!db != !db2)) {  // no split-K reduction to route the rows: two products
      hipError_t e1 = gemm_dw(dY, lddy, X, x_ids, x_S, ldx, dW, M_all, n1, K, slabs, stream, live_dy, live_x, n_live, db, csum, k_dev,
                              accumulate);
      if (e1 != hipSuccess) return e1;
      return gemm_dw(dY + n1, lddy, X, x_ids, x_S, ldx, dW2, M_all, N - n1, K, slabs, stream, live_dy, live_x, n_live, db2, csum,
                     k_dev, accumulate);
    }
    g.C2 = dW2;
    g.c2_row0 = n1;
    g.colsum_out2 = db2;
  }
  if (fuse_db) {  // partials per K slice; the split-K reduction launch adds them up into db (GemmArgs::colsum_out)
    g.colsum = csum;
    g.colsum_out = db;
  }
  // M = the rows actually contracted (the live ones)
  ProfScope ps(7, 2.0 * (double)prof_count(7, g.k_dev, 0, M, stream) * N * K, stream);
  hipError_t e = launch_gemm_f32(g, stream);
  if (e != hipSuccess) return e;
  if (db && !fuse_db) return launch_colsum(dY, lddy, nullptr, M_all, N, db, csum, stream);  // all rows (the non-live ones are zero)
  return hipSuccess;
}

// dX[M,K] (+)= (dY[M,N] . W[N,K]) (*) f'(aux)
// With a scratch buffer (>= N*K floats) and enough rows, W is transposed first (a few MB, microseconds) so that
// the product runs on the forward-layout kernel -- both operands k-contiguous, raw buffer loads, 4 workgroups per
// CU: ~133 TF -- instead of the k-major variant (87 TF on the 80 000-row dX GEMMs of the NRMS train step).
hipError_t gemm_dx(const float* dY, int64_t lddy, const float* W, float* dX, int64_t lddx, int64_t M, int N, int K,
                   const float* aux, int64_t ldaux, int aux_mode, int accumulate, hipStream_t stream,
                   float* wt_scratch = nullptr, const int32_t* live = nullptr, int64_t n_live = 0,
                   const int64_t* m_dev = nullptr) {
  GemmArgs g{};
  g.A = dY;
  g.lda = lddy;
  if (live) {  // rows live[j] of dY and dX only, in place (the other rows of dY are zero; dX's are left as they are)
    if (n_live <= 0) return hipSuccess;
    g.gather_ids = live;
    g.gather_S = 1;
    g.c_scatter = 1;
    M = n_live;
    g.m_dev = m_dev;  // the list's length on the device (n_live is then its capacity)
    g.m_fill_hint = 0.4f;
  }
  g.nseg = 1;
  g.Nseg = K;
  g.C = dX;
  g.ldc = lddx;
  g.M = M;
  g.K = N;
  g.aux = aux;
  g.ldaux = ldaux;
  g.aux_mode = aux_mode;
  g.accumulate = accumulate;
  ProfScope ps(8, 2.0 * (double)prof_count(8, g.m_dev, 0, M, stream) * N * K, stream);
  if (g.m_dev && !wt_scratch) return hipErrorInvalidValue;  // device row counts: forward-layout kernel only
  if (wt_scratch && (M >= 4096 || g.m_dev)) {
    hipError_t e = launch_transpose(W, wt_scratch, N, K, stream);  // Wt[K][N]
    if (e != hipSuccess) return e;
    g.W[0] = wt_scratch;
    g.ldw = N;
  } else {
    g.W[0] = W;
    g.b_kn = 1;
    g.ldw = K;
  }
  return launch_gemm_f32(g, stream);
}

}  // namespace

extern "C" {

size_t xnrs_seq_encoder_saved_bytes(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E, int32_t n_heads,
                                    int32_t pool_kind, int32_t has_head) {
  const bool pooled = pool_kind != XNRS_POOL_NONE;
  return make_plan(n_seq, L, D, A, E, n_heads > 0, pool_kind == XNRS_POOL_ADDITIVE, pooled && has_head, pooled, 0, true,
                   n_heads)
      .total;
}

size_t xnrs_seq_encoder_saved_qkv_offset(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E, int32_t n_heads,
                                         int32_t pool_kind, int32_t has_head) {
  const bool pooled = pool_kind != XNRS_POOL_NONE;
  return make_plan(n_seq, L, D, A, E, n_heads > 0, pool_kind == XNRS_POOL_ADDITIVE, pooled && has_head, pooled, 0, true, n_heads)
      .off_qkv;
}

int32_t xnrs_seq_encoder_fwd_train(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                                   const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                                   const xnrs_head_params* head, float* y, float* a_out, float* hm, void* saved,
                                   size_t saved_bytes, void* stream) {
  return xnrs_seq_encoder_fwd_train_live(x, m, ids, n_seq, L, D, att, pool_kind, pool, head, y, a_out, hm, saved, saved_bytes,
                                         nullptr, nullptr, 0, stream);
}

int32_t xnrs_seq_encoder_fwd_train_live(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                                        const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                                        const xnrs_head_params* head, float* y, float* a_out, float* hm, void* saved,
                                        size_t saved_bytes, const int32_t* live_rows, const int32_t* live_src_rows,
                                        int64_t n_live, void* stream) {
  xnrs_row_lists r{};
  r.live_rows = live_rows;
  r.live_src_rows = live_src_rows;
  r.n_live = n_live;
  return xnrs_seq_encoder_fwd_train_rows(x, m, ids, n_seq, L, D, att, pool_kind, pool, head, y, a_out, hm, saved, saved_bytes,
                                         live_rows ? &r : nullptr, stream);
}

int32_t xnrs_seq_encoder_fwd_train_rows(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                                        const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                                        const xnrs_head_params* head, float* y, float* a_out, float* hm, void* saved,
                                        size_t saved_bytes, const xnrs_row_lists* r, void* stream) {
  const bool pooled = pool_kind != XNRS_POOL_NONE;
  xnrs_row_lists none{};
  if (!r) r = &none;
  if (r->kv_rows && !r->live_rows) return XNRS_EINVAL;  // the K|V list rides on the live-row path
  // a gathered table needs the table rows of the listed tokens
  if (ids && ((r->live_rows && !r->live_src_rows) || (r->kv_rows && !r->kv_src_rows))) return XNRS_EINVAL;
  return seq_encode(x, m, ids, n_seq, L, D, att, pooled, pool_kind, pool, pooled ? head : nullptr, y, a_out, hm, 0, saved,
                    saved_bytes, (hipStream_t)stream, true, r);
}

size_t xnrs_seq_encoder_bwd_workspace_bytes(int64_t n_seq, int32_t L, int32_t D, int32_t A, int32_t E, int32_t n_heads,
                                            int32_t pool_kind, int32_t has_head) {
  const bool pooled = pool_kind != XNRS_POOL_NONE;
  return make_bwd_plan(n_seq, L, D, A, E, n_heads, pool_kind == XNRS_POOL_ADDITIVE, pooled && has_head, pooled).total;
}

int32_t xnrs_seq_encoder_bwd(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                             const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                             const xnrs_head_params* head, const void* saved, size_t saved_bytes, const float* dy, float* dx,
                             const xnrs_mha_grads* g_att, const xnrs_additive_grads* g_pool, const xnrs_head_grads* g_head,
                             void* ws, size_t ws_bytes, void* stream_) {
  return xnrs_seq_encoder_bwd_live(x, m, ids, n_seq, L, D, att, pool_kind, pool, head, saved, saved_bytes, dy, dx, g_att,
                                   g_pool, g_head, nullptr, nullptr, 0, ws, ws_bytes, stream_);
}

int32_t xnrs_seq_encoder_bwd_live(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                                  const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                                  const xnrs_head_params* head, const void* saved, size_t saved_bytes, const float* dy,
                                  float* dx, const xnrs_mha_grads* g_att, const xnrs_additive_grads* g_pool,
                                  const xnrs_head_grads* g_head, const int32_t* live_rows, const int32_t* live_src_rows,
                                  int64_t n_live, void* ws, size_t ws_bytes, void* stream_) {
  xnrs_row_lists r{};
  r.live_rows = live_rows;
  r.live_src_rows = live_src_rows;
  r.n_live = n_live;
  return xnrs_seq_encoder_bwd_rows(x, m, ids, n_seq, L, D, att, pool_kind, pool, head, saved, saved_bytes, dy, dx, g_att, g_pool,
                                   g_head, live_rows ? &r : nullptr, ws, ws_bytes, stream_);
}

int32_t xnrs_seq_encoder_bwd_rows(const float* x, const float* m, const int32_t* ids, int64_t n_seq, int32_t L, int32_t D,
                                  const xnrs_mha_params* att, int32_t pool_kind, const xnrs_additive_params* pool,
                                  const xnrs_head_params* head, const void* saved, size_t saved_bytes, const float* dy,
                                  float* dx, const xnrs_mha_grads* g_att, const xnrs_additive_grads* g_pool,
                                  const xnrs_head_grads* g_head, const xnrs_row_lists* rl, void* ws, size_t ws_bytes,
                                  void* stream_) {
  xnrs_row_lists none{};
  if (!rl) rl = &none;
  const int32_t* live_rows = rl->live_rows;
  const int32_t* live_src_rows = rl->live_src_rows;
  // counts on the device (xnrs_row_lists::counts_dev): n_live / n_kv are then the lists' capacities (every row)
  const int64_t* cnt = rl->counts_dev;
  const int64_t n_live = cnt ? n_seq * L : rl->n_live, n_kv = cnt ? n_seq * L : rl->n_kv;
  if (rl->kv_rows && !live_rows) return XNRS_EINVAL;
  if (rl->dqkv_mode != XNRS_DQKV_OWN && (rl->dqkv_mode < 0 || rl->dqkv_mode > XNRS_DQKV_MERGE || !att || !rl->dqkv_image || dx))
    return XNRS_EINVAL;
  if (ids && rl->kv_rows && !rl->kv_src_rows) return XNRS_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_seq == 0) return XNRS_OK;
  if (n_seq < 0 || L <= 0 || D <= 0 || !x || !dy) return XNRS_EINVAL;
  if (ids && dx) return XNRS_EINVAL;
  const bool pooled = pool_kind != XNRS_POOL_NONE;
  const bool additive = pool_kind == XNRS_POOL_ADDITIVE;
  if (!pooled) head = nullptr;
  if (att && D % att->n_heads != 0) return XNRS_EHEADS;
  if (additive && !pool) return XNRS_EINVAL;
  if (pool_kind == XNRS_POOL_MEAN && !m) return XNRS_EINVAL;
  const int A = additive ? pool->hidden : 0;
  const int E = head ? head->out_features : D;
  const int nh = att ? att->n_heads : 0;
  const Plan sp = make_plan(n_seq, L, D, A, E, att != nullptr, additive, head != nullptr, pooled, 0, true, nh);
  if (sp.total > saved_bytes || (sp.total > 0 && !saved)) return XNRS_EWORKSPACE;
  const BwdPlan bp = make_bwd_plan(n_seq, L, D, A, E, nh, additive, head != nullptr, pooled);
  if (bp.total > ws_bytes || (bp.total > 0 && !ws)) return XNRS_EWORKSPACE;
  const char* sv = static_cast<const char*>(saved);
  const float* qkv = (att && rl->qkv_shared) ? rl->qkv_shared : reinterpret_cast<const float*>(sv + sp.off_qkv);
  const float* o = reinterpret_cast<const float*>(sv + sp.off_o);
  const float* yatt = reinterpret_cast<const float*>(sv + sp.off_y);
  const float* t = reinterpret_cast<const float*>(sv + sp.off_t);
  const float* pb = reinterpret_cast<const float*>(sv + sp.off_p);
  const float* hb = reinterpret_cast<const float*>(sv + sp.off_h);
  const float* stats = reinterpret_cast<const float*>(sv + sp.off_stats);
  const float* a_sv = reinterpret_cast<const float*>(sv + sp.off_a);
  char* w = static_cast<char*>(ws);
  float* dh = reinterpret_cast<float*>(w + bp.off_dh);
  float* dp = reinterpret_cast<float*>(w + bp.off_dp);
  float* dseq = reinterpret_cast<float*>(w + bp.off_dseq);
  float* dpre = reinterpret_cast<float*>(w + bp.off_dpre);
  float* de = reinterpret_cast<float*>(w + bp.off_de);
  float* docat = reinterpret_cast<float*>(w + bp.off_docat);
  // (dqkv_mode: the caller's image shared by the two backward calls over one Q|K|V image, xnrs_row_lists)
  float* dqkv = rl->dqkv_mode != XNRS_DQKV_OWN ? rl->dqkv_image : reinterpret_cast<float*>(w + bp.off_dqkv);
  float* delta = reinterpret_cast<float*>(w + bp.off_delta);
  float* slabs = reinterpret_cast<float*>(w + bp.off_slabs);
  float* csum = reinterpret_cast<float*>(w + bp.off_colsum);
  float* wt = reinterpret_cast<float*>(w + bp.off_wt);
  const int64_t rows = n_seq * L;
  // Live rows (optional): the unmasked token rows.  A masked row has pooling weight 0, so every gradient that passes
  // through it is exactly zero (dy_i = a_i dp = 0, dpre_i = 0, dO_i = 0, dS_i = 0): the row-parallel GEMMs of the
  // attention tower run over the live rows only, in place.  K and V gradients stay dense (padded rows are keys).
  const bool live = live_rows && pooled && additive && m && (cnt || (n_live >= 0 && n_live < rows));
  if (cnt && live && !device_counts_ok(x, D, A, att, pool)) return XNRS_EUNSUPPORTED;
  const int64_t* cnt_live = (cnt && live) ? cnt : nullptr;
  const int32_t* lv = live ? live_rows : nullptr;
  const int32_t* lvx = live ? (live_src_rows ? live_src_rows : live_rows) : nullptr;
  if (live_rows && ids && !live_src_rows) return XNRS_EINVAL;  // a gathered table needs the table rows of the live tokens
  // K / V gradients over the token rows of the non-empty news (the forward's kv list): an all-masked news has no live query,
  // so its dK and dV rows are exactly zero
  const bool kvl = live && rl->kv_rows && (cnt || (n_kv >= 0 && n_kv < rows));
  const int64_t* cnt_kv = (cnt && kvl) ? cnt + 1 : nullptr;
  const int32_t* kvr = kvl ? rl->kv_rows : nullptr;
  const int32_t* kvx = kvl ? (rl->kv_src_rows ? rl->kv_src_rows : rl->kv_rows) : nullptr;
  const bool fold = att && pooled && additive && fold_wanted(knobs().fold_train);  // = the forward's decision

  // weight-gradient launches go to the side lane (SideLane above): `sw` = that stream, re-ordered behind the caller's stream
  // (after_main) wherever the input-gradient chain has produced what the next weight gradients read.  slabs / csum are
  // touched by lane launches only, wt by the chain only.
  // Where it pays (tools/bench_side_lane.py, B = 64 grad steps): the NRMS step, GPU-bound at ~30 us per launch, 7.93 -> 7.64
  // ms; the attention-free towers of StandardRec / NAML (1.4 / 3.3 ms steps of ~100 launches: the HOST is the limit there and
  // the fork / join calls only add to it) +6 % / +2 % -- so the lane serves towers with an attention stage and at least
  // XNRS_BWD_SIDE_MIN_ROWS token rows.
  Fork fk(stream, att != nullptr && rows >= knobs().bwd_side_min_rows);
  hipStream_t sw = stream;

  // gradient w.r.t. the sequence rows that fed the pooler (att output, or x itself)
  const float* dseq_src = nullptr;  // [rows, D]
  if (pooled) {
    // ---- head: y = W2 relu(W0 p + b0) + b2
    const float* dpool = dy;  // [n_seq, D]
    if (head) {
      sw = fk.after_main();  // (dy: produced on the caller's stream before this call)
      if (g_head && g_head->w2)
        XNRS_TRY(gemm_dw(dy, E, hb, nullptr, 0, E, g_head->w2, n_seq, E, E, slabs, sw, nullptr, nullptr, 0, g_head->b2, csum));
      else if (g_head && g_head->b2) XNRS_TRY(launch_colsum(dy, E, nullptr, n_seq, E, g_head->b2, csum, sw));
      // f'(saved activation): relu' (aux mode 2), tanh' = 1 - t^2 (1), identity (0)
      const int hmode = head->activation == XNRS_ACT_RELU ? 2 : (head->activation == XNRS_ACT_TANH ? 1 : 0);
      XNRS_TRY(gemm_dx(dy, E, head->w2, dh, E, n_seq, E, E, hmode ? hb : nullptr, E, hmode, 0, stream, wt));
      sw = fk.after_main();  // dh
      if (g_head && g_head->w0)
        XNRS_TRY(gemm_dw(dh, E, pb, nullptr, 0, D, g_head->w0, n_seq, E, D, slabs, sw, nullptr, nullptr, 0, g_head->b0, csum));
      else if (g_head && g_head->b0) XNRS_TRY(launch_colsum(dh, E, nullptr, n_seq, E, g_head->b0, csum, sw));
      XNRS_TRY(gemm_dx(dh, E, head->w0, dp, D, n_seq, E, D, nullptr, 0, 0, 0, stream, wt));
      dpool = dp;
    }
    // ---- pooler
    if (fold) {
      // Folded out-projection (seq_encode "fold"; the forward saved O, tanh(W' O + b'), a, the pooled O rows and sum a):
      //   p = Wo po + bo s,  po = sum_i a_i O_i,  s = sum_i a_i;   pre_i = W' O_i + b',  W' = W1 Wo,  b' = W1 bo + b1
      //   g = Wo^T dp, c = dp . bo:  da_i = g . O_i + c,  dO_i = a_i g + dpre_i W'
      //   dW' = dpre^T O, db' = sum dpre:  dW1 = dW' Wo^T + db' (x) bo,  db1 = db'
      //   dWo = dp^T po + W1^T dW',  dbo = sum_n s_n dp_n + W1^T db'      (one stacked product / column sum each)
      // The three rows x D x D products of the per-token order (forward out-projection, dO = dY Wo, dWo = dY^T O) are gone.
      const float* wf = pool->w1_folded ? pool->w1_folded : reinterpret_cast<const float*>(sv + sp.off_fw);  // (as the forward was given)
      const float* pob = reinterpret_cast<const float*>(sv + sp.off_po);
      const float* asum = reinterpret_cast<const float*>(sv + sp.off_as);
      float* gvec = reinterpret_cast<float*>(w + bp.off_g);
      float* dwf = reinterpret_cast<float*>(w + bp.off_dwf);
      // db' IS db1 (see the algebra above): produced in place when the caller wants it (a device copy per call before)
      float* dbf = (g_pool && g_pool->b1) ? g_pool->b1 : reinterpret_cast<float*>(w + bp.off_dbf);
      XNRS_TRY(gemm_dx(dpool, D, att->wo, gvec, D, n_seq, D, D, nullptr, 0, 0, 0, stream, wt));

      AdditivePoolBwdArgs pa{};
      pa.dp = gvec;
      pa.x = o;
      pa.ldx = D;
      pa.a = a_sv;
      pa.t = t;
      pa.w2 = pool->w2;
      pa.dx = docat;  // dO_i = a_i g (every row written; masked rows get 0)
      pa.lddx = D;
      pa.dpre = dpre;
      pa.de = de;
      pa.shift_u = att->bo ? dpool : nullptr;  // c_n = dp_n . bo, taken inside the kernel
      pa.shift_v = att->bo;
      pa.n_seq = n_seq;
      pa.N = L;
      pa.D = D;
      pa.A = A;
      XNRS_TRY(launch_additive_pool_bwd(pa, stream));
      sw = fk.after_main();  // dpool (dp), dpre, de
      XNRS_TRY(gemm_dx(dpre, A, wf, docat, D, rows, A, D, nullptr, 0, 0, /*accumulate*/ 1, stream, wt, lv, n_live, cnt_live));
      if (g_pool && g_pool->w2 && g_pool->b2) XNRS_TRY(launch_colsum_wsum(t, A, de, rows, A, g_pool->w2, g_pool->b2, csum, sw));
      else if (g_pool && g_pool->w2) XNRS_TRY(launch_colsum(t, A, de, rows, A, g_pool->w2, csum, sw));
      else if (g_pool && g_pool->b2) XNRS_TRY(launch_colsum(de, 1, nullptr, rows, 1, g_pool->b2, csum, sw));
      XNRS_TRY(gemm_dw(dpre, A, o, nullptr, 0, D, dwf, rows, A, D, slabs, sw, lv, lv, n_live, dbf, csum, cnt_live));
      if (g_pool && g_pool->w1) {
        GemmArgs g1 = gemm1(dwf, nullptr, 0, D, att->wo, nullptr, g_pool->w1, D, A, D, D, XNRS_ACT_NONE);
        if (att->bo) {  // + db' (x) bo in the epilogue: fmaf(db'[a], bo[d], acc), the bits of the separate pass it replaces
          g1.rowscale = dbf;
          g1.rowscale_vec = att->bo;
        }
        XNRS_TRY(launch_gemm_f32(g1, sw));
      }
      // dWo = dp^T po + W1^T dW' as two products (the second accumulates), dbo = sum_n s_n dp_n + sum_a db'_a W1[a,:] as ONE
      // column sum over the two row blocks (round 3 staged [dp; W1] and [po; dW'] with six device copies per call)
      if (g_att && g_att->wo) {
        XNRS_TRY(gemm_dw(dpool, D, pob, nullptr, 0, D, g_att->wo, n_seq, D, D, slabs, sw));
        XNRS_TRY(gemm_dw(pool->w1, D, dwf, nullptr, 0, D, g_att->wo, A, D, D, slabs, sw, nullptr, nullptr, 0, nullptr, nullptr,
                         nullptr, /*accumulate*/ 1));
      }
      if (g_att && g_att->bo) XNRS_TRY(launch_colsum2(dpool, D, asum, n_seq, pool->w1, D, dbf, A, D, g_att->bo, csum, sw));

    } else {
    const float* seq = att ? yatt : x;
    const int32_t* seq_ids = att ? nullptr : ids;
    const bool need_dseq = att || dx;
    float* dseq_dst = att ? dseq : dx;  // no attention stage: the sequence rows ARE x
    if (additive) {
      AdditivePoolBwdArgs pa{};
      pa.dp = dpool;
      pa.x = seq;
      pa.ldx = D;
      pa.x_gather_ids = seq_ids;
      pa.a = a_sv;
      pa.t = t;
      pa.w2 = pool->w2;
      pa.dx = need_dseq ? dseq_dst : nullptr;
      pa.lddx = D;
      pa.dpre = dpre;
      pa.de = de;
      pa.n_seq = n_seq;
      pa.N = L;
      pa.D = D;
      pa.A = A;
      XNRS_TRY(launch_additive_pool_bwd(pa, stream));
      sw = fk.after_main();  // dpre, de
      if (g_pool && g_pool->w2 && g_pool->b2) XNRS_TRY(launch_colsum_wsum(t, A, de, rows, A, g_pool->w2, g_pool->b2, csum, sw));
      else if (g_pool && g_pool->w2) XNRS_TRY(launch_colsum(t, A, de, rows, A, g_pool->w2, csum, sw));
      else if (g_pool && g_pool->b2) XNRS_TRY(launch_colsum(de, 1, nullptr, rows, 1, g_pool->b2, csum, sw));
      if (g_pool && g_pool->w1 && live)  // rows of dpre through lv; rows of seq through lv (yatt) or lvx (x / table rows)
        XNRS_TRY(gemm_dw(dpre, A, seq, nullptr, 0, D, g_pool->w1, rows, A, D, slabs, sw, lv, att ? lv : lvx, n_live, g_pool->b1, csum, cnt_live));
      else if (g_pool && g_pool->w1)
        XNRS_TRY(gemm_dw(dpre, A, seq, seq_ids, L, D, g_pool->w1, rows, A, D, slabs, sw, nullptr, nullptr, 0, g_pool->b1, csum));
      else if (g_pool && g_pool->b1) XNRS_TRY(launch_colsum(dpre, A, nullptr, rows, A, g_pool->b1, csum, sw));
      if (need_dseq)
        XNRS_TRY(gemm_dx(dpre, A, pool->w1, dseq_dst, D, rows, A, D, nullptr, 0, 0, /*accumulate*/ 1, stream, wt, lv, n_live, cnt_live));
    } else if (need_dseq) {
      XNRS_TRY(launch_mean_pool_bwd(dpool, m, ids, dseq_dst, D, n_seq, L, D, stream));
    }
    dseq_src = dseq_dst;
    }
  } else {
    dseq_src = dy;  // MultiHeadAttention alone: dy is the gradient of the attention output
  }
  if (!att) return XNRS_OK;

  // ---- out projection: yatt = O Wo^T + bo   (folded: docat and the Wo / bo gradients are complete already)
  if (!fold) {
    sw = fk.after_main();  // the sequence-row gradient is complete (pooler: its fc1 dX product accumulated into it)
    if (g_att && g_att->wo)
      XNRS_TRY(gemm_dw(dseq_src, D, o, nullptr, 0, D, g_att->wo, rows, D, D, slabs, sw, lv, lv, n_live, g_att->bo, csum, cnt_live));
    else if (g_att && g_att->bo) XNRS_TRY(launch_colsum(dseq_src, D, nullptr, rows, D, g_att->bo, csum, sw));
    if (live) XNRS_TRY(hipMemsetAsync(docat, 0, (size_t)rows * D * sizeof(float), stream));  // dO of a masked row is zero
    XNRS_TRY(gemm_dx(dseq_src, D, att->wo, docat, D, rows, D, D, nullptr, 0, 0, 0, stream, wt, lv, n_live, cnt_live));
  }
  // ---- attention core
  MhaBwdArgs mb{};
  mb.q = qkv;
  mb.k = qkv + D;
  mb.v = qkv + 2 * (int64_t)D;
  mb.ld = 3 * (int64_t)D;
  mb.mask = m;
  mb.mask_gather_ids = ids;
  mb.o = o;
  mb.ldo = D;
  mb.d_o = docat;
  mb.lddo = D;
  mb.stats = stats;
  mb.delta = delta;
  mb.dq = dqkv;
  mb.dk = dqkv + D;
  mb.dv = dqkv + 2 * (int64_t)D;
  mb.ldd = 3 * (int64_t)D;
  mb.n_seq = n_seq;
  mb.S = L;
  mb.n_heads = nh;
  mb.d_k = D / nh;
  mb.scaled = att->scaled;
  mb.dropout_p = att->dropout_p;
  mb.seed = att->seed;
  mb.seed_dev = att->seed_dev;
  mb.masked_do_is_zero = (pooled && m) ? 1 : 0;  // both poolers give masked rows a zero gradient
  // an all-masked news has dQ = dK = dV = 0: written without reading (1), or -- when every consumer goes through the row
  // lists and no input gradient is asked for -- not even written (2)
  // (a deferring call: the merging call that consumes its image reads it through the same lists -- the caller's contract)
  const bool lists_only = kvl && !dx && (rl->dqkv_mode == XNRS_DQKV_DEFER || (g_att && g_att->wq && g_att->wk && g_att->wv));  // (a bias-only gradient sums dense rows)
  mb.dead_seq_mode = live ? (lists_only ? 2 : 1) : 0;
  mb.accumulate = rl->dqkv_mode == XNRS_DQKV_MERGE ? 1 : 0;
  {
    ProfScope ps(9, 10.0 * rows * (double)L * D, stream);  // S, dP, dV, dK, dQ: five S x S x d_k products per head
    XNRS_TRY(launch_mha_bwd(mb, stream));
  }
  if (rl->dqkv_mode == XNRS_DQKV_DEFER) return XNRS_OK;  // the merging call computes the projection gradients from the sum
  sw = fk.after_main();  // dQ | dK | dV
  // ---- Q/K/V projections
  float* gw[3] = {g_att ? g_att->wq : nullptr, g_att ? g_att->wk : nullptr, g_att ? g_att->wv : nullptr};
  float* gb[3] = {g_att ? g_att->bq : nullptr, g_att ? g_att->bk : nullptr, g_att ? g_att->bv : nullptr};
  const float* wqkv[3] = {att->wq, att->wk, att->wv};
  // dWk | dWv as ONE product when both are wanted and contract over the same rows: the K and V columns of the image lie side
  // by side (A = the 2D columns from D on), X is staged once per tile for both, the split-K reduction routes the two halves
  // (and their bias sums) to the two parameters -- half the launches and K loops twice as long per workgroup
  bool kv_merged = false;
  if (gw[1] && gw[2] && !dx && (!gb[1] == !gb[2])) {
    const float* dkv = dqkv + D;
    if (kvl)
      XNRS_TRY(gemm_dw(dkv, 3 * (int64_t)D, x, nullptr, 0, D, gw[1], rows, 2 * D, D, slabs, sw, kvr, kvx, n_kv, gb[1], csum, cnt_kv,
                       0, gw[2], gb[2], D));
    else
      XNRS_TRY(gemm_dw(dkv, 3 * (int64_t)D, x, ids, L, D, gw[1], rows, 2 * D, D, slabs, sw, nullptr, nullptr, 0, gb[1], csum,
                       nullptr, 0, gw[2], gb[2], D));
    kv_merged = true;
  }
  for (int s3 = 0; s3 < 3; ++s3) {
    const float* dpart = dqkv + (int64_t)s3 * D;
    if (kv_merged && s3 > 0) continue;
    if (gw[s3]) {
      if (s3 == 0 && live)  // dQ is zero on masked rows; dK / dV are not (padded tokens are keys) ...
        XNRS_TRY(gemm_dw(dpart, 3 * (int64_t)D, x, nullptr, 0, D, gw[s3], rows, D, D, slabs, sw, lv, lvx, n_live, gb[s3], csum, cnt_live));
      else if (s3 > 0 && kvl)  // ... except on the rows of an all-masked news
        XNRS_TRY(gemm_dw(dpart, 3 * (int64_t)D, x, nullptr, 0, D, gw[s3], rows, D, D, slabs, sw, kvr, kvx, n_kv, gb[s3], csum, cnt_kv));
      else
        XNRS_TRY(gemm_dw(dpart, 3 * (int64_t)D, x, ids, L, D, gw[s3], rows, D, D, slabs, sw, nullptr, nullptr, 0, gb[s3], csum));
    } else if (gb[s3]) {
      XNRS_TRY(launch_colsum(dpart, 3 * (int64_t)D, nullptr, rows, D, gb[s3], csum, sw));
    }
    if (dx) XNRS_TRY(gemm_dx(dpart, 3 * (int64_t)D, wqkv[s3], dx, D, rows, D, D, nullptr, 0, 0, s3 > 0 ? 1 : 0, stream, wt));
  }
  return XNRS_OK;
}

size_t xnrs_linear_bwd_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  return align_up(gemm_splitk_workspace_bytes(N, K, M)) + align_up(colsum_workspace_bytes(N));
}

int32_t xnrs_linear_bwd(const float* x, const int32_t* gather_ids, int32_t gather_S, const float* w, const float* dy,
                        float* dx, float* dw, float* db, int64_t M, int32_t N, int32_t K, void* ws, size_t ws_bytes,
                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return XNRS_OK;
  if (!x || !w || !dy || M < 0 || N <= 0 || K <= 0) return XNRS_EINVAL;
  if (gather_ids && (dx || gather_S <= 0)) return XNRS_EINVAL;
  const size_t s1 = align_up(gemm_splitk_workspace_bytes(N, K, M));
  if (s1 + align_up(colsum_workspace_bytes(N)) > ws_bytes || !ws) return XNRS_EWORKSPACE;
  float* slabs = static_cast<float*>(ws);
  float* csum = reinterpret_cast<float*>(static_cast<char*>(ws) + s1);
  if (dw) XNRS_TRY(gemm_dw(dy, N, x, gather_ids, gather_S, K, dw, M, N, K, slabs, stream, nullptr, nullptr, 0, db, csum));
  else if (db) XNRS_TRY(launch_colsum(dy, N, nullptr, M, N, db, csum, stream));
  if (dx) XNRS_TRY(gemm_dx(dy, N, w, dx, K, M, N, K, nullptr, 0, 0, 0, stream));
  return XNRS_OK;
}

size_t xnrs_embedding_linear_bwd_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  return xnrs_linear_bwd_workspace_bytes(M, N, K) + align_up((size_t)M * K * sizeof(float));
}

int32_t xnrs_embedding_linear_bwd(const float* table, const int32_t* ids, const float* w, const float* dy, float* d_table,
                                  float* dw, float* db, int64_t M, int32_t N, int32_t K, int32_t n_rows, void* ws,
                                  size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return XNRS_OK;
  if (!table || !ids || !w || !dy || M < 0 || N <= 0 || K <= 0 || n_rows <= 0) return XNRS_EINVAL;
  const size_t s1 = xnrs_linear_bwd_workspace_bytes(M, N, K);
  if (s1 + align_up((size_t)M * K * sizeof(float)) > ws_bytes || !ws) return XNRS_EWORKSPACE;
  int32_t rc = xnrs_linear_bwd(table, ids, 1, w, dy, nullptr, dw, db, M, N, K, ws, s1, stream_);
  if (rc != XNRS_OK) return rc;
  if (d_table) {
    float* d_rows = reinterpret_cast<float*>(static_cast<char*>(ws) + s1);
    XNRS_TRY(gemm_dx(dy, N, w, d_rows, K, M, N, K, nullptr, 0, 0, 0, stream));
    XNRS_TRY(launch_embedding_grad(d_rows, ids, M, K, d_table, n_rows, stream));
  }
  return XNRS_OK;
}

int32_t xnrs_dot_scoring_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B, int32_t C,
                             int32_t E, void* stream) {
  if (!u || !c || !dr || B < 0 || C <= 0 || E <= 0) return XNRS_EINVAL;
  return hip_rc(launch_dot_scoring_bwd(u, c, dr, du, dc, B, C, E, (hipStream_t)stream));
}

int32_t xnrs_dot_scoring_norm_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B, int32_t C,
                                  int32_t E, void* stream) {
  if (!u || !c || !dr || B < 0 || C <= 0 || E <= 0) return XNRS_EINVAL;
  if (E > 1024) return XNRS_EUNSUPPORTED;
  return hip_rc(launch_dot_scoring_norm_bwd(u, c, dr, du, dc, B, C, E, (hipStream_t)stream));
}

}  // extern "C"

extern "C" {

int32_t xnrs_assemble_train_batch(const int64_t* sess, int64_t B, const int64_t* hist_off, const int32_t* hist_val,
                                  const int64_t* pos_off, const int32_t* pos_val, const int64_t* neg_off,
                                  const int32_t* neg_val, int32_t l_hist, int32_t n_neg, int32_t pad_row, uint64_t seed,
                                  int32_t* hist_rows, int32_t* cand_rows, void* stream) {
  if (B == 0) return XNRS_OK;
  if (!sess || !hist_off || !pos_off || !neg_off || !hist_rows || !cand_rows || B < 0 || l_hist <= 0 || n_neg < 0)
    return XNRS_EINVAL;
  BatchArgs a{};
  a.sess = sess; a.hist_off = hist_off; a.hist_val = hist_val; a.pos_off = pos_off; a.pos_val = pos_val;
  a.neg_off = neg_off; a.neg_val = neg_val; a.B = B; a.l_hist = l_hist; a.n_neg = n_neg; a.pad_row = pad_row;
  a.seed = seed; a.hist_out = hist_rows; a.cand_out = cand_rows;
  return hip_rc(launch_assemble_train(a, (hipStream_t)stream));
}

int32_t xnrs_assemble_eval_batch(const int64_t* sess, int64_t B, const int64_t* hist_off, const int32_t* hist_val,
                                 const int64_t* pos_off, const int32_t* pos_val, const int64_t* neg_off,
                                 const int32_t* neg_val, int32_t l_hist, int32_t pad_row, const int64_t* cand_off,
                                 int32_t* hist_rows, int32_t* cand_rows, int32_t* cand_sess, float* targets, void* stream) {
  if (B == 0) return XNRS_OK;
  if (!sess || !hist_off || !pos_off || !neg_off || !cand_off || !hist_rows || !cand_rows || !cand_sess || !targets ||
      B < 0 || l_hist <= 0)
    return XNRS_EINVAL;
  BatchArgs a{};
  a.sess = sess; a.hist_off = hist_off; a.hist_val = hist_val; a.pos_off = pos_off; a.pos_val = pos_val;
  a.neg_off = neg_off; a.neg_val = neg_val; a.B = B; a.l_hist = l_hist; a.pad_row = pad_row;
  a.hist_out = hist_rows; a.cand_out = cand_rows; a.cand_off_out = cand_off; a.cand_sess_out = cand_sess;
  a.targets_out = targets;
  return hip_rc(launch_assemble_eval(a, (hipStream_t)stream));
}

int32_t xnrs_gather_rows(const float* table, const int32_t* ids, float* out, int64_t n, int64_t row_floats, void* stream) {
  if (n == 0) return XNRS_OK;
  if (!table || !ids || !out || n < 0 || row_floats <= 0) return XNRS_EINVAL;
  return hip_rc(launch_gather_rows(table, ids, out, n, row_floats, (hipStream_t)stream));
}

int32_t xnrs_score_csr(const float* vecs, const int32_t* cand_rows, const int32_t* cand_sess, const float* u, float* r,
                       int64_t n_cand, int32_t E, int32_t relu, void* stream) {
  if (n_cand == 0) return XNRS_OK;
  if (!vecs || !cand_rows || !cand_sess || !u || !r || n_cand < 0 || E <= 0) return XNRS_EINVAL;
  return hip_rc(launch_score_csr(vecs, cand_rows, cand_sess, u, r, n_cand, E, relu, (hipStream_t)stream));
}

int32_t xnrs_rank_metrics(const float* scores, const float* targets, const int64_t* cand_off, float* out, int64_t B,
                          void* stream) {
  if (B == 0) return XNRS_OK;
  if (!scores || !targets || !cand_off || !out || B < 0) return XNRS_EINVAL;
  return hip_rc(launch_rank_metrics(scores, targets, cand_off, out, B, (hipStream_t)stream));
}

}  // extern "C"

extern "C" {

size_t xnrs_infonce_saved_bytes(int64_t B, int32_t E) { return ((size_t)B * E + 4 * (size_t)B + 1) * sizeof(float); }

int32_t xnrs_infonce_fwd(const float* emb, const int64_t* labels, int64_t B, int32_t E, float temperature, float* loss,
                         void* saved, size_t saved_bytes, void* stream) {
  if (!emb || !labels || !loss || !saved || B <= 0 || E <= 0 || E > 1024 || !(temperature > 0.f)) return XNRS_EINVAL;
  if (saved_bytes < xnrs_infonce_saved_bytes(B, E)) return XNRS_EWORKSPACE;
  return hip_rc(launch_infonce_fwd(emb, labels, B, E, temperature, loss, static_cast<float*>(saved), (hipStream_t)stream));
}

int32_t xnrs_infonce_bwd(const int64_t* labels, int64_t B, int32_t E, float temperature, const void* saved, size_t saved_bytes,
                         const float* gout, float* demb, void* stream) {
  if (!labels || !saved || !gout || !demb || B <= 0 || E <= 0 || E > 1024 || !(temperature > 0.f)) return XNRS_EINVAL;
  if (saved_bytes < xnrs_infonce_saved_bytes(B, E)) return XNRS_EWORKSPACE;
  return hip_rc(launch_infonce_bwd(labels, B, E, temperature, static_cast<const float*>(saved), gout, demb, (hipStream_t)stream));
}

}  // extern "C"
