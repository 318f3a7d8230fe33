// Device-side batch assembly and evaluation kernels -- the callers / data formats on either side of the
// encoders.  Integer / index work, HBM-bound, no MFMA.
//
//   assemble_train : NewsRecDataset.__getitem__ in 'train' mode (xnrs/data/dataset.py:54-57,77-85,97-109,147)
//                    + custom_collate_fn (xnrs/utils.py:190-204), producing ROW IDS into the device-resident
//                    news table instead of 4.6 MB of gathered token tensors per impression.
//   assemble_eval  : the same in 'eval' mode (dataset.py:58-61,149): all positives then all negatives,
//                    variable C -> CSR.
//   score_csr      : DotScoring over a CSR candidate list against pre-encoded news vectors
//                    (scoring.py:23 applied per impression, training.py:194-203 with batch_size 1).
//   rank_metrics   : xnrs/evaluation/metrics.py:9-64 per impression (nDCG@k, RR, CTR@k, AUC, acc/rec/prec).
//   gather_rows    : the dict look-ups + torch.cat of NewsRecDataset.__getitem__ (dataset.py:63-85,97-109) as a
//                    device copy: out[i] = table[ids[i]] for whole news blocks (S*D floats, 150 KB at the shipped
//                    shape) -- the materialised batch for consumers that need dense token tensors (input gradients
//                    of the explainer, explain.py:160-166); the encoders themselves gather inside their first load.
#include <atomic>

#include "kernels.h"

namespace xnrs {

__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t a, uint64_t b) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a * 0x100000001B3ull + b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// one thread per output id
__global__ __launch_bounds__(256) void assemble_train_kernel(BatchArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W = a.l_hist + 1 + a.n_neg;
  if (i >= a.B * W) return;
  const int64_t b = i / W;
  const int j = (int)(i - b * W);
  const int64_t s = a.sess[b];
  if (j < a.l_hist) {
    // the LAST l_hist clicked news first, zero padding behind them (dataset.py:77-85)
    const int64_t lo = a.hist_off[s], hi = a.hist_off[s + 1];
    const int64_t n = (hi - lo < a.l_hist) ? (hi - lo) : a.l_hist;
    a.hist_out[b * a.l_hist + j] = (j < n) ? a.hist_val[hi - n + j] : a.pad_row;
  } else {
    const int c = j - a.l_hist;  // 0 = the positive, 1.. = negatives
    int32_t row = a.pad_row;
    if (c == 0) {
      const int64_t lo = a.pos_off[s], n = a.pos_off[s + 1] - lo;
      if (n > 0) row = a.pos_val[lo + (int64_t)(mix64(a.seed, (uint64_t)s, 0) % (uint64_t)n)];  // random.choice
    } else {
      const int64_t lo = a.neg_off[s], n = a.neg_off[s + 1] - lo;
      if (n > 0) row = a.neg_val[lo + (int64_t)(mix64(a.seed, (uint64_t)s, (uint64_t)c) % (uint64_t)n)];  // random.choices (with replacement)
    }
    a.cand_out[b * (1 + a.n_neg) + c] = row;
  }
}

hipError_t launch_assemble_train(const BatchArgs& a, hipStream_t stream) {
  const int64_t n = a.B * (a.l_hist + 1 + a.n_neg);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(assemble_train_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// one wave per impression: history ids + candidate CSR fill (offsets computed by the caller)
__global__ __launch_bounds__(256) void assemble_eval_kernel(BatchArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int lane = threadIdx.x & 63;
  const int64_t s = a.sess[b];
  {
    const int64_t lo = a.hist_off[s], hi = a.hist_off[s + 1];
    const int64_t n = (hi - lo < a.l_hist) ? (hi - lo) : a.l_hist;
    for (int j = lane; j < a.l_hist; j += 64) a.hist_out[b * a.l_hist + j] = (j < n) ? a.hist_val[hi - n + j] : a.pad_row;
  }
  const int64_t plo = a.pos_off[s], np = a.pos_off[s + 1] - plo;
  const int64_t nlo = a.neg_off[s], nn = a.neg_off[s + 1] - nlo;
  const int64_t o = a.cand_off_out[b];
  for (int64_t k = lane; k < np + nn; k += 64) {
    a.cand_out[o + k] = (k < np) ? a.pos_val[plo + k] : a.neg_val[nlo + (k - np)];
    a.cand_sess_out[o + k] = (int32_t)b;
    a.targets_out[o + k] = (k < np) ? 1.f : 0.f;
  }
}

hipError_t launch_assemble_eval(const BatchArgs& a, hipStream_t stream) {
  if (a.B <= 0) return hipSuccess;
  hipLaunchKernelGGL(assemble_eval_kernel, dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// r[e] = <vecs[cand_rows[e], :], u[cand_sess[e], :]> : one wave per candidate entry
__global__ __launch_bounds__(256) void score_csr_kernel(const float* vecs, const int32_t* rows, const int32_t* sess, const float* u,
                                                         float* r, int64_t n, int E, int relu) {
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n) return;
  const int lane = threadIdx.x & 63;
  const float* v = vecs + (int64_t)rows[e] * E;
  const float* uu = u + (int64_t)sess[e] * E;
  float acc = 0.f;
  for (int k = lane; k < E; k += 64) acc = fmaf(v[k], uu[k], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) r[e] = relu ? fmaxf(acc, 0.f) : acc;
}

hipError_t launch_score_csr(const float* vecs, const int32_t* rows, const int32_t* sess, const float* u, float* r, int64_t n,
                            int E, int relu, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(score_csr_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, vecs, rows, sess, u, r, n, E, relu);
  return hipGetLastError();
}

// Per-impression ranking metrics, one wave per impression, O(C^2) rank counting (C is tens to hundreds).
// Order = np.argsort(score)[::-1]; ties are broken "higher original index first" (what a stable ascending
// sort reversed gives; numpy's default sort is stable below 17 elements -- beyond that the reference's own
// tie order is unspecified).  out[b, :] = {ndcg@5, ndcg@10, rr, ctr@1, ctr@10, auc, acc, rec, prec}.
__global__ __launch_bounds__(256) void rank_metrics_kernel(const float* score, const float* target, const int64_t* off, float* out,
                                                            int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int64_t lo = off[b];
  const int C = (int)(off[b + 1] - lo);
  const float* s = score + lo;
  const float* t = target + lo;
  float dcg5 = 0.f, dcg10 = 0.f, rr = 0.f, top1 = 0.f, top10 = 0.f, npos = 0.f, conc = 0.f;
  float tp = 0.f, fp = 0.f, fn = 0.f, tn = 0.f;
  for (int e = lane; e < C; e += 64) {
    float se = s[e];
    if (se != se) se = 0.f;  // np.nan_to_num(nan=0, posinf=1, neginf=0) (training.py:210-211)
    else if (se == INFINITY) se = 1.f;
    else if (se == -INFINITY) se = 0.f;
    const float te = t[e];
    int rank = 1;
    float gt = 0.f, eq = 0.f;  // negatives scored below / equal (for AUC), only used when te == 1
    for (int f = 0; f < C; ++f) {
      float sf = s[f];
      if (sf != sf) sf = 0.f;
      else if (sf == INFINITY) sf = 1.f;
      else if (sf == -INFINITY) sf = 0.f;
      if (sf > se || (sf == se && f > e)) ++rank;
      if (t[f] == 0.f) {
        if (sf < se) gt += 1.f;
        else if (sf == se) eq += 1.f;
      }
    }
    const float gain = exp2f(te) - 1.f;  // 2**y - 1 (metrics.py:12)
    const float disc = log2f((float)rank + 1.f);
    if (rank <= 5) dcg5 += gain / disc;
    if (rank <= 10) dcg10 += gain / disc;
    if (te > 0.f) {
      rr = fmaxf(rr, te / (float)rank);
      conc += gt + 0.5f * eq;
      npos += 1.f;
    }
    if (rank <= 1) top1 += te;
    if (rank <= 10) top10 += te;
    const float pred = rintf(fminf(fmaxf(se, 0.f), 1.f));  // np.round(np.clip(s, 0, 1)): round-half-even
    if (pred > 0.5f) { if (te > 0.5f) tp += 1.f; else fp += 1.f; }
    else { if (te > 0.5f) fn += 1.f; else tn += 1.f; }
  }
  float v[11] = {dcg5, dcg10, top1, top10, npos, conc, tp, fp, fn, tn, 0.f};
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) v[i] += __shfl_xor(v[i], o2);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) rr = fmaxf(rr, __shfl_xor(rr, o2));
  if (lane == 0) {
    const float np_ = v[4], nneg = (float)C - v[4];
    // ideal DCG: the positives first (metrics.py:18-19: dcg_score(y_true, y_true, k))
    float best5 = 0.f, best10 = 0.f;
    for (int i = 0; i < 10 && i < (int)np_; ++i) {
      const float d = 1.f / log2f((float)i + 2.f);
      if (i < 5) best5 += d;
      best10 += d;
    }
    float* o = out + b * 9;
    o[0] = v[0] / best5;
    o[1] = v[1] / best10;
    o[2] = rr;
    o[3] = v[2] / (float)(C < 1 ? C : 1);
    o[4] = v[3] / (float)(C < 10 ? C : 10);
    o[5] = v[5] / (np_ * nneg);
    o[6] = (v[6] + v[9]) / (float)C;
    o[7] = v[6] / (v[6] + v[8]);
    o[8] = (v[6] + v[7]) > 0.f ? v[6] / (v[6] + v[7]) : 0.f;  // zero_division=0 (metrics.py:58)
  }
}

hipError_t launch_rank_metrics(const float* score, const float* target, const int64_t* off, float* out, int64_t B,
                               hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(rank_metrics_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, score, target, off, out, B);
  return hipGetLastError();
}

// ---------------------------------------------------------------- gather_rows (HBM-bound block copy)
// one workgroup per (output row, 16-KB piece): 4 independent 16-byte loads per thread in flight, streaming
// (nontemporal) on both sides -- every byte is touched once
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                           float* __restrict__ out, int64_t row_floats, int pieces) {
  const int64_t row = blockIdx.x / pieces;
  const int piece = (int)(blockIdx.x - row * pieces);
  const int64_t src = (int64_t)ids[row] * row_floats, dst = row * row_floats;
  if (VEC) {
    const int64_t n4 = row_floats >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(table + src);
    f32x4* d4 = reinterpret_cast<f32x4*>(out + dst);
    const int64_t base = (int64_t)piece * 1024 + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + 256 * u < n4) v[u] = __builtin_nontemporal_load(s4 + base + 256 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + 256 * u < n4) __builtin_nontemporal_store(v[u], d4 + base + 256 * u);
  } else {
    const int64_t base = (int64_t)piece * 4096 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (base + 256 * u < row_floats) out[dst + base + 256 * u] = table[src + base + 256 * u];
  }
}

hipError_t launch_gather_rows(const float* table, const int32_t* ids, float* out, int64_t n, int64_t row_floats,
                              hipStream_t stream) {
  if (n <= 0 || row_floats <= 0) return hipSuccess;
  const bool vec = row_floats % 4 == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const int64_t pieces = (row_floats + 4095) / 4096;  // 16 KB per workgroup
  if (n * pieces > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)(n * pieces));
  if (vec) hipLaunchKernelGGL((gather_rows_kernel<true>), grid, dim3(256), 0, stream, table, ids, out, row_floats, (int)pieces);
  else hipLaunchKernelGGL((gather_rows_kernel<false>), grid, dim3(256), 0, stream, table, ids, out, row_floats, (int)pieces);
  return hipGetLastError();
}

// ---- row compaction on the device (the padding-free encoder without a host round trip).
// For every pass of `chunk` news (mask [.., S] fp32 0/1, optionally gathered by news id): CSR offsets of the live token rows
// per news, the list of live token rows, and the list of ALL token rows of the non-empty news (K / V are projected for
// every token of a news that has a live query: the reference masks QUERY rows only, layers.py:142-144 -- and for none of
// an all-masked news, whose keys nobody reads).  ONE launch for the whole call, one 1024-thread workgroup per pass
// (blockIdx.x = pass p, news p*chunk .. ; its lists start at p*chunk*S, its offsets at p*(chunk+1), its counts at 3*p):
// a wave reads one news' mask row coalesced and counts / places its live tokens by ballot, the two scans over the news of
// the pass are wave scans joined through LDS.  (A first version ran per pass with one thread per news walking its mask
// row three times: 109 us per pass, 2.4 ms of a 50 ms step.)
//   row_off [chunk+1]   compact range of news j: row_off[j] .. row_off[j+1]
//   live_src [<= chunk*S]  source token row (x row space: ids[news]*S + s with a table, else news*S + s) of compact row i
//   kv_src [<= chunk*S]  source row of the i-th kept token row: the K|V image of a pass holds the non-empty news as
//                        CONSECUTIVE blocks of S rows; kv_block[j] = block of news j (the attention kernel's MhaCoreArgs::kv_block)
//   counts[0] = live rows, counts[1] = kept K|V rows      (device scalars the GEMMs read: GemmArgs::m_dev)
//   counts[2] = 1 if a mask value other than 0 / 1 was seen: the pooling kernel then writes NaN instead of a result that
//               would silently differ from the reference's exp(e) * m (there is no host read here to raise from)
__global__ __launch_bounds__(1024) void compact_rows_kernel(const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                             int64_t n_news, int64_t chunk, int S, int64_t* __restrict__ row_off_all,
                                                             int32_t* __restrict__ live_all, int32_t* __restrict__ kvs_all,
                                                             int32_t* __restrict__ kvb_all, int64_t* __restrict__ counts_all) {
  __shared__ int s_cnt[1024];
  __shared__ int s_ex[2][1024];
  __shared__ int s_wsum[2][16];
  __shared__ int s_carry[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t news0 = (int64_t)blockIdx.x * chunk;
  const int cn = (int)(n_news - news0 < chunk ? n_news - news0 : chunk);
  int64_t* row_off = row_off_all + (int64_t)blockIdx.x * (chunk + 1);
  int32_t* live_src = live_all + (int64_t)blockIdx.x * chunk * S;
  int32_t* kv_src = kvs_all + (int64_t)blockIdx.x * chunk * S;
  int32_t* kv_block = kvb_all + (int64_t)blockIdx.x * chunk;
  int64_t* counts = counts_all + 3 * (int64_t)blockIdx.x;
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;  // the lanes before this one
  int bad = 0;
  if (tid < 2) s_carry[tid] = 0;
  if (tid == 0) row_off[0] = 0;
  __syncthreads();
  for (int base = 0; base < cn; base += 1024) {
    // live tokens per news: wave w takes the news base + w, base + w + 16, ...
    for (int jj = wave; jj < 1024; jj += 16) {
      const int j = base + jj;
      int cnt = 0;
      if (j < cn) {
        const int64_t mrow = ids ? (int64_t)ids[news0 + j] : news0 + j;
        const float* mp = mask + mrow * S;
        for (int s0 = 0; s0 < S; s0 += 64) {
          const float mv = s0 + lane < S ? mp[s0 + lane] : 0.f;
          bad |= (mv != 0.f && mv != 1.f) ? 1 : 0;
          cnt += __popcll(__ballot(mv != 0.f));
        }
      }
      if (lane == 0) s_cnt[jj] = cnt;
    }
    __syncthreads();
    // exclusive scans of (live count) and (kept K|V rows = S for a non-empty news) over the 1024 news of this round
    const int cnt = s_cnt[tid];
    const int v[2] = {cnt, cnt > 0 ? S : 0};
    int incl[2];
    for (int which = 0; which < 2; ++which) {
      int x = v[which];
      for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(x, off);
        if (lane >= off) x += up;
      }
      incl[which] = x;
      if (lane == 63) s_wsum[which][wave] = x;
    }
    __syncthreads();
    for (int which = 0; which < 2; ++which) {
      int before = s_carry[which];
      for (int w = 0; w < wave; ++w) before += s_wsum[which][w];
      s_ex[which][tid] = before + incl[which] - v[which];
    }
    if (base + tid < cn) row_off[base + tid + 1] = s_ex[0][tid] + cnt;
    __syncthreads();
    if (tid == 1023) {
      s_carry[0] = s_ex[0][1023] + v[0];
      s_carry[1] = s_ex[1][1023] + v[1];
    }
    // the lists, a wave per news again
    for (int jj = wave; jj < 1024; jj += 16) {
      const int j = base + jj;
      if (j >= cn) break;
      const int64_t mrow = ids ? (int64_t)ids[news0 + j] : news0 + j;
      const float* mp = mask + mrow * S;
      const int64_t src0 = mrow * S;  // source token row of (news, 0)
      const bool kept = s_cnt[jj] > 0;
      int w = s_ex[0][jj];
      const int e1 = s_ex[1][jj];
      for (int s0 = 0; s0 < S; s0 += 64) {
        const int sl = s0 + lane;
        const bool on = sl < S && mp[sl] != 0.f;
        const uint64_t b = __ballot(on);
        if (on) live_src[w + __popcll(b & below)] = (int32_t)(src0 + sl);
        w += __popcll(b);
        if (kept && sl < S) kv_src[e1 + sl] = (int32_t)(src0 + sl);
      }
      if (lane == 0) kv_block[j] = e1 / S;  // (an empty news: the block the next non-empty one gets -- never read)
    }
    __syncthreads();
  }
  bad = __syncthreads_or(bad);
  if (tid == 0) {
    counts[0] = s_carry[0];
    counts[1] = s_carry[1];
    counts[2] = bad ? 1 : 0;
  }
}

// The same lists for S <= 64 (every shape with an attention tower), throughput-shaped: the mask of a round of 1 024 news is
// read ONCE, flattened over the workgroup (element i -> news i / S, token i % S: coalesced, ~50 independent loads per
// thread instead of a wave walking 64 news one dependent load at a time), each live token sets its bit in the news' 64-bit
// word in LDS; counts are popcounts, and the lists are written by the same flattened sweep (a live token's place = the
// news' offset + the popcount of the bits below it).  378 -> 249 us per call of 25 600 news x 50 in five passes (it sits on
// the critical path of the device-compacted encoder; one workgroup per pass is what bounds it now: the passes' rounds of
// 1 024 news run one after the other).
__global__ __launch_bounds__(1024) void compact_rows64_kernel(const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                               int64_t n_news, int64_t chunk, int S, int64_t* __restrict__ row_off_all,
                                                               int32_t* __restrict__ live_all, int32_t* __restrict__ kvs_all,
                                                               int32_t* __restrict__ kvb_all, int64_t* __restrict__ counts_all) {
  __shared__ unsigned long long s_bits[1024];
  __shared__ int64_t s_row[1024];
  __shared__ int s_ex[2][1024];
  __shared__ int s_wsum[2][16];
  __shared__ int s_carry[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t news0 = (int64_t)blockIdx.x * chunk;
  const int cn = (int)(n_news - news0 < chunk ? n_news - news0 : chunk);
  int64_t* row_off = row_off_all + (int64_t)blockIdx.x * (chunk + 1);
  int32_t* live_src = live_all + (int64_t)blockIdx.x * chunk * S;
  int32_t* kv_src = kvs_all + (int64_t)blockIdx.x * chunk * S;
  int32_t* kv_block = kvb_all + (int64_t)blockIdx.x * chunk;
  int64_t* counts = counts_all + 3 * (int64_t)blockIdx.x;
  int bad = 0;
  if (tid < 2) s_carry[tid] = 0;
  if (tid == 0) row_off[0] = 0;
  for (int base = 0; base < cn; base += 1024) {
    const int nn = cn - base < 1024 ? cn - base : 1024;
    s_bits[tid] = 0ull;
    s_row[tid] = tid < nn ? (ids ? (int64_t)ids[news0 + base + tid] : news0 + base + tid) : 0;
    __syncthreads();
    const int n_el = nn * S;
    // four elements per trip, their loads issued together (the sweep is latency-bound: one load, one LDS atomic per element)
    for (int i0 = tid; i0 < n_el; i0 += 4096) {
      float mv[4];
      int jj[4], sl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        jj[u] = (i < n_el ? i : 0) / S;
        sl[u] = (i < n_el ? i : 0) - jj[u] * S;
        mv[u] = i < n_el ? mask[s_row[jj[u]] * S + sl[u]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bad |= (mv[u] != 0.f && mv[u] != 1.f) ? 1 : 0;
        if (mv[u] != 0.f) atomicOr(&s_bits[jj[u]], 1ull << sl[u]);
      }
    }
    __syncthreads();
    const unsigned long long bits = s_bits[tid];
    const int cnt = __popcll(bits);
    const int v[2] = {cnt, cnt > 0 ? S : 0};
    int incl[2];
    for (int which = 0; which < 2; ++which) {
      int x = v[which];
      for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(x, off);
        if (lane >= off) x += up;
      }
      incl[which] = x;
      if (lane == 63) s_wsum[which][wave] = x;
    }
    __syncthreads();
    for (int which = 0; which < 2; ++which) {
      int before = s_carry[which];
      for (int w = 0; w < wave; ++w) before += s_wsum[which][w];
      s_ex[which][tid] = before + incl[which] - v[which];
    }
    if (tid < nn) {
      row_off[base + tid + 1] = s_ex[0][tid] + cnt;
      kv_block[base + tid] = s_ex[1][tid] / S;  // (an empty news: the block the next non-empty one gets -- never read)
    }
    __syncthreads();
    if (tid == 1023) {
      s_carry[0] = s_ex[0][1023] + v[0];
      s_carry[1] = s_ex[1][1023] + v[1];
    }
#pragma unroll 4
    for (int i = tid; i < n_el; i += 1024) {
      const int jj = i / S, sl = i - jj * S;
      const unsigned long long b = s_bits[jj];
      if (b == 0ull) continue;
      const int32_t src = (int32_t)(s_row[jj] * S + sl);
      kv_src[s_ex[1][jj] + sl] = src;
      if ((b >> sl) & 1ull) live_src[s_ex[0][jj] + __popcll(b & ((1ull << sl) - 1ull))] = src;
    }
    __syncthreads();
  }
  bad = __syncthreads_or(bad);
  if (tid == 0) {
    counts[0] = s_carry[0];
    counts[1] = s_carry[1];
    counts[2] = bad ? 1 : 0;
  }
}

// ---- the grad step's row lists on the device (xnrs_build_row_lists): the unmasked token rows ("live") and all token rows of
// the non-empty sequences ("kv"), each in the batch's own row space [n_seq*L] and -- with a gathered table -- in the table's,
// plus both counts as device scalars the GEMMs read (GemmArgs::m_dev / k_dev).  Replaces the torch bookkeeping of
// xnrs_amd/autograd.py (a nonzero + ONE host read of the counts per encoder call).  Two short launches, both chip-wide:
//   counts: a wave per sequence -> cnt[seq] = live tokens (ballot popcounts)
//   lists : a workgroup per 64 sequences: its base offsets = sums over cnt[0 .. first) (coalesced, a few KB), a wave scan
//           over its own 64 counts, then a wave per sequence places its tokens by ballot.  Order = row order, exactly the
//           lists torch.nonzero gave.
__global__ __launch_bounds__(256) void row_counts_kernel(const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                          int64_t n_seq, int L, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t seq = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seq >= n_seq) return;
  const float* mp = mask + (ids ? (int64_t)ids[seq] : seq) * L;
  int c = 0;
  for (int s0 = 0; s0 < L; s0 += 64) c += __popcll(__ballot(s0 + lane < L && mp[s0 + lane] != 0.f));
  if (lane == 0) cnt[seq] = c;
}

constexpr int RL_SEQ = 64;  // sequences per workgroup of row_lists_kernel
__global__ __launch_bounds__(1024) void row_lists_kernel(const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                          int64_t n_seq, int L, const int32_t* __restrict__ cnt,
                                                          int32_t* __restrict__ live, int32_t* __restrict__ live_src,
                                                          int32_t* __restrict__ kv, int32_t* __restrict__ kv_src,
                                                          int64_t* __restrict__ counts) {
  __shared__ int s_red[2][16];
  __shared__ int s_base[2];
  __shared__ int s_ex[2][RL_SEQ];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t first = (int64_t)blockIdx.x * RL_SEQ;
  // base offsets of this workgroup: live tokens / non-empty sequences before `first`
  int a0 = 0, a1 = 0;
  for (int64_t i = tid; i < first; i += 1024) {
    const int c = cnt[i];
    a0 += c;
    a1 += c > 0 ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off);
    a1 += __shfl_xor(a1, off);
  }
  if (lane == 0) {
    s_red[0][wave] = a0;
    s_red[1][wave] = a1;
  }
  __syncthreads();
  if (tid == 0) {
    int b0 = 0, b1 = 0;
    for (int w = 0; w < 16; ++w) {
      b0 += s_red[0][w];
      b1 += s_red[1][w];
    }
    s_base[0] = b0;
    s_base[1] = b1 * L;
  }
  if (wave == 0) {  // exclusive scan over this workgroup's 64 sequences
    const int64_t seq = first + lane;
    const int c = seq < n_seq ? cnt[seq] : 0;
    int x0 = c, x1 = c > 0 ? L : 0;
    const int v0 = x0, v1 = x1;
    for (int off = 1; off < 64; off <<= 1) {
      const int u0 = __shfl_up(x0, off), u1 = __shfl_up(x1, off);
      if (lane >= off) {
        x0 += u0;
        x1 += u1;
      }
    }
    s_ex[0][lane] = x0 - v0;
    s_ex[1][lane] = x1 - v1;
  }
  __syncthreads();
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
  for (int j = wave; j < RL_SEQ; j += 16) {
    const int64_t seq = first + j;
    if (seq >= n_seq) break;
    const int64_t mrow = ids ? (int64_t)ids[seq] : seq;
    const float* mp = mask + mrow * L;
    const bool kept = cnt[seq] > 0;
    int w = s_base[0] + s_ex[0][j];
    const int e1 = s_base[1] + s_ex[1][j];
    for (int s0 = 0; s0 < L; s0 += 64) {
      const int sl = s0 + lane;
      const bool on = sl < L && mp[sl] != 0.f;
      const uint64_t b = __ballot(on);
      if (on) {
        const int at = w + __popcll(b & below);
        live[at] = (int32_t)(seq * L + sl);
        if (live_src) live_src[at] = (int32_t)(mrow * L + sl);
      }
      w += __popcll(b);
      if (kept && sl < L) {
        kv[e1 + sl] = (int32_t)(seq * L + sl);
        if (kv_src) kv_src[e1 + sl] = (int32_t)(mrow * L + sl);
      }
    }
  }
  if (first + RL_SEQ >= n_seq && tid == 0) {  // the last workgroup knows the totals
    const int nn = (int)(n_seq - first);
    const int lastc = cnt[n_seq - 1];
    counts[0] = s_base[0] + s_ex[0][nn - 1] + lastc;
    counts[1] = s_base[1] + s_ex[1][nn - 1] + (lastc > 0 ? L : 0);
  }
}

hipError_t launch_build_row_lists(const float* mask, const int32_t* ids, int64_t n_seq, int L, int32_t* live, int32_t* live_src,
                                  int32_t* kv, int32_t* kv_src, int64_t* counts, int32_t* cnt_scratch, hipStream_t stream) {
  if (n_seq <= 0 || L <= 0) return hipSuccess;
  if (n_seq * (int64_t)L > 0x7fffffffLL) return hipErrorInvalidValue;  // int32 row indices
  hipLaunchKernelGGL(row_counts_kernel, dim3((unsigned)((n_seq + 3) / 4)), dim3(256), 0, stream, mask, ids, n_seq, L, cnt_scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(row_lists_kernel, dim3((unsigned)((n_seq + RL_SEQ - 1) / RL_SEQ)), dim3(1024), 0, stream, mask, ids, n_seq, L,
                     cnt_scratch, live, live_src, kv, kv_src, counts);
  return hipGetLastError();
}

// NaN over a result whose precondition turned out violated on the device (the flags of every pass, OR-ed)
__global__ __launch_bounds__(256) void poison_kernel(float* y, int64_t n, const int64_t* flags, int n_flags, int flag_stride,
                                                     int32_t* status) {
  bool bad = false;
  for (int i = 0; i < n_flags; ++i) bad = bad || flags[(int64_t)i * flag_stride] != 0;
  if (!bad) return;
  if (status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1 /* XNRS_STATUS_NONBINARY_MASK */);
  const float nanv = __builtin_nanf("");
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = nanv;
}

hipError_t launch_poison(float* y, int64_t n, const int64_t* flags, int n_flags, int flag_stride, hipStream_t stream) {
  if (n <= 0 || n_flags <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(poison_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, n, flags, n_flags, flag_stride, status_word());
  return hipGetLastError();
}

// ONE word, on the device that was current when it was registered: a launch on another device gets nullptr (its kernels must
// not be handed a pointer into this device's memory; their NaN outputs remain the signal there)
static std::atomic<int32_t*> g_status_word{nullptr};
static std::atomic<int> g_status_device{-1};
int32_t* status_word() {
  int32_t* w = g_status_word.load(std::memory_order_relaxed);
  if (!w) return nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != g_status_device.load(std::memory_order_relaxed)) return nullptr;
  return w;
}
void set_status_word(int32_t* w) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  g_status_device.store(w ? dev : -1, std::memory_order_relaxed);
  g_status_word.store(w, std::memory_order_relaxed);
}

hipError_t launch_compact_rows(const float* mask, const int32_t* ids, int64_t n_news, int64_t chunk, int S, int64_t* row_off,
                               int32_t* live_src, int32_t* kv_src, int32_t* kv_block, int64_t* counts, hipStream_t stream) {
  if (n_news <= 0 || chunk <= 0) return hipSuccess;
  const int64_t passes = (n_news + chunk - 1) / chunk;
  if (passes > 0x7fffffffLL) return hipErrorInvalidValue;
  if (S <= 64)
    hipLaunchKernelGGL(compact_rows64_kernel, dim3((unsigned)passes), dim3(1024), 0, stream, mask, ids, n_news, chunk, S, row_off,
                       live_src, kv_src, kv_block, counts);
  else
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)passes), dim3(1024), 0, stream, mask, ids, n_news, chunk, S, row_off,
                       live_src, kv_src, kv_block, counts);
  return hipGetLastError();
}

}  // namespace xnrs
