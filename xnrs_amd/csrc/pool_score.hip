// Sequence pooling and scoring kernels (fp32, HBM-bound streaming reductions; no MFMA):
//   additive_pool : layers.AdditiveAttention.forward after fc1+tanh (xnrs/models/components/layers.py:60-65)
//   mean_pool     : layers.MaskedMean.forward                         (layers.py:26-37)
//   collapse_mask : xnrs.utils.collaps_mask                           (xnrs/utils.py:74-75)
//   dot_scoring   : scoring.DotScoring.forward                        (xnrs/models/components/scoring.py:12-23)
// One workgroup (256 threads) per sequence for the poolers: N <= a few hundred rows of D floats.
#include "kernels.h"

namespace xnrs {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int POOL_MAX_N = 512;

// a_i = exp(w2 . t_i + b2) * m_i / (sum_j exp(..)*m_j + 1e-8);  y = sum_i a_i x_i.
// exp is NOT max-stabilised and the epsilon is 1e-8, exactly as layers.py:61-64.
__global__ __launch_bounds__(256) void additive_pool_kernel(AdditivePoolArgs a) {
  __shared__ float s_w[POOL_MAX_N];
  __shared__ float s_red[4];
  __shared__ short s_idx[POOL_MAX_N];  // the rows with a non-zero weight, in row order
  __shared__ int s_nlive;
  const int64_t seq = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = a.A, D = a.D;
  const int64_t src = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] : seq;  // row block of the mask
  const int64_t srcx = a.x_gather_ids ? (int64_t)a.x_gather_ids[seq] : seq;       // row block of the values
  // unpadded: the rows of this sequence are a compact range and all of them are live
  const bool csr = a.row_off != nullptr;
  const int64_t r0 = csr ? a.row_off[seq] : seq * a.N;
  const int N = csr ? (int)(a.row_off[seq + 1] - r0) : a.N;
  const float* mask = csr ? nullptr : a.mask;

  // 1. scores: one wave per row, lanes stride over the hidden dimension -- or, with the fc2 dot already taken per block
  //    of 32 hidden columns in the fc1 GEMM's epilogue (epart), one thread per row adding the blocks in column order
  const float b2 = a.b2 ? a.b2[0] : 0.f;
  if (a.epart) {
    for (int i = tid; i < N; i += 256) {
      const float* ep = a.epart + (r0 + i) * (int64_t)a.n_epart;
      float acc = 0.f;
      for (int s = 0; s < a.n_epart; ++s) acc += ep[s];
      float e = expf(acc + b2);
      if (mask) e *= mask[src * N + i];
      s_w[i] = e;
    }
  } else
  for (int i = wave; i < N; i += 4) {
    const float* t = a.t + (r0 + i) * (int64_t)A;
    float acc = 0.f;
    for (int k = lane; k < A; k += 64) acc = fmaf(t[k], a.w2[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      float e = expf(acc + b2);
      if (mask) e *= mask[src * N + i];
      s_w[i] = e;
    }
  }
  __syncthreads();
  // (round 4) the rows whose weight is not exactly 0, in row order: step 3 reads only those -- a masked row contributes
  // fmaf(0, x, acc) = acc, so the sum has the same bits, and the 72 % of the benchmark batch's token rows that are masked
  // (empty history slots + the padding of the titles) are no longer streamed from HBM to be multiplied by zero
  if (wave == 0) {
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int base = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
      const int i = i0 + lane;
      const bool on = i < N && s_w[i] != 0.f;
      const uint64_t b = __ballot(on);
      if (on) s_idx[base + __popcll(b & below)] = (short)i;
      base += __popcll(b);
    }
    if (lane == 0) s_nlive = base;
  }
  // 2. normaliser (+ news mask side output)
  float part = 0.f, mpart = 0.f;
  for (int i = tid; i < N; i += 256) {
    part += s_w[i];
    mpart += mask ? mask[src * N + i] : (csr ? 1.f : 0.f);
  }
  part = wave_sum(part);
  mpart = wave_sum(mpart);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float sumw = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const float denom = sumw + 1e-8f;
  __syncthreads();
  const bool poisoned = a.poison && *a.poison != 0;
  const float nanv = __builtin_nanf("");
  if (a.asum_out && tid == 0) a.asum_out[seq] = poisoned ? nanv : sumw / denom;
  if (a.hm_out) {
    if (lane == 0) s_red[wave] = mpart;
    __syncthreads();
    if (tid == 0) a.hm_out[seq] = poisoned ? nanv : fminf(fmaxf(s_red[0] + s_red[1] + s_red[2] + s_red[3], 0.f), 1.f);
  }
  if (a.a_out)
    for (int i = tid; i < N; i += 256) a.a_out[r0 + i] = s_w[i] / denom;
  // 3. weighted sum of the value rows: thread per column, rows streamed (coalesced across threads)
  const float* x = a.x + (csr ? r0 : srcx * N) * a.ldx;
  const int32_t* rid = (csr && a.row_ids) ? a.row_ids + r0 : nullptr;
  const int nlive = s_nlive;  // (written before the barriers above)
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    if (rid) {
      for (int j = 0; j < nlive; ++j) {
        const int i = s_idx[j];
        acc = fmaf(s_w[i] / denom, a.x[(int64_t)rid[i] * a.ldx + d], acc);
      }
    } else {
      for (int j = 0; j < nlive; ++j) {
        const int i = s_idx[j];
        acc = fmaf(s_w[i] / denom, x[(int64_t)i * a.ldx + d], acc);
      }
    }
    a.y[seq * D + d] = poisoned ? nanv : acc;
  }
}

__global__ __launch_bounds__(256) void add_rowscaled_bias_kernel(float* p, int64_t ld, const float* s, const float* b, int64_t n, int D) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r = i / D;
  const int d = (int)(i - r * D);
  p[r * ld + d] = fmaf(s[r], b[d], p[r * ld + d]);
}

__global__ __launch_bounds__(256) void fold_bias_kernel(const float* w1, const float* bo, const float* b1, float* bf, int A, int D) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= A) return;  // wave-uniform
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc = fmaf(w1[(int64_t)a * D + d], bo[d], acc);
  acc = wave_sum(acc);
  if (lane == 0) bf[a] = acc + (b1 ? b1[a] : 0.f);
}

hipError_t launch_fold_bias(const float* w1, const float* bo, const float* b1, float* bf, int A, int D, hipStream_t stream) {
  if (A <= 0) return hipSuccess;
  hipLaunchKernelGGL(fold_bias_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, stream, w1, bo, b1, bf, A, D);
  return hipGetLastError();
}

hipError_t launch_add_rowscaled_bias(float* p, int64_t ld, const float* s, const float* b, int64_t n, int D, hipStream_t stream) {
  if (n <= 0 || D <= 0) return hipSuccess;
  const int64_t blocks = (n * D + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_rowscaled_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, ld, s, b, n, D);
  return hipGetLastError();
}

hipError_t launch_additive_pool(const AdditivePoolArgs& a, hipStream_t stream) {
  if (a.n_seq <= 0) return hipSuccess;
  if (a.N > POOL_MAX_N || a.N <= 0) return hipErrorInvalidValue;  // with row_off: N = the padded length bounds every count
  hipLaunchKernelGGL(additive_pool_kernel, dim3((unsigned)a.n_seq), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// y = sum_i x_i m_i / (sum_i m_i + 1e-8)
__global__ __launch_bounds__(256) void mean_pool_kernel(MeanPoolArgs a) {
  __shared__ float s_m[POOL_MAX_N];
  __shared__ float s_red[4];
  const int64_t seq = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, D = a.D;
  const int64_t src = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] : seq;
  const int64_t srcx = a.x_gather_ids ? (int64_t)a.x_gather_ids[seq] : seq;
  float part = 0.f;
  for (int i = tid; i < N; i += 256) {
    const float m = a.mask[src * N + i];
    s_m[i] = m;
    part += m;
  }
  part = wave_sum(part);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float msum = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  if (a.hm_out && tid == 0) a.hm_out[seq] = fminf(fmaxf(msum, 0.f), 1.f);
  const float denom = msum + 1e-8f;
  // the rows with a non-zero mask, in row order (see additive_pool_kernel): fmaf(x, 0, acc) = acc
  __shared__ short s_idx[POOL_MAX_N];
  __shared__ int s_nlive;
  if (wave == 0) {
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int base = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
      const int i = i0 + lane;
      const bool on = i < N && s_m[i] != 0.f;
      const uint64_t b = __ballot(on);
      if (on) s_idx[base + __popcll(b & below)] = (short)i;
      base += __popcll(b);
    }
    if (lane == 0) s_nlive = base;
  }
  __syncthreads();
  const int nlive = s_nlive;
  const float* x = a.x + srcx * N * a.ldx;
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int j = 0; j < nlive; ++j) {
      const int i = s_idx[j];
      acc = fmaf(x[(int64_t)i * a.ldx + d], s_m[i], acc);
    }
    a.y[seq * D + d] = acc / denom;
  }
}

hipError_t launch_mean_pool(const MeanPoolArgs& a, hipStream_t stream) {
  if (a.n_seq <= 0) return hipSuccess;
  if (a.N > POOL_MAX_N || a.N <= 0 || !a.mask) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mean_pool_kernel, dim3((unsigned)a.n_seq), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// hm[n] = clamp(sum_s m[n,s], 0, 1): one wave per row
__global__ __launch_bounds__(256) void collapse_mask_kernel(const float* m, const int32_t* ids, float* hm, int64_t n_rows,
                                                             int S) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t src = ids ? (int64_t)ids[row] : row;
  float acc = 0.f;
  for (int s = lane; s < S; s += 64) acc += m[src * S + s];
  acc = wave_sum(acc);
  if (lane == 0) hm[row] = fminf(fmaxf(acc, 0.f), 1.f);
}

hipError_t launch_collapse_mask(const float* m, const int32_t* ids, float* hm, int64_t n_rows, int32_t S,
                                hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(collapse_mask_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, stream, m, ids, hm, n_rows,
                     S);
  return hipGetLastError();
}

// r[b,c] = <c[b,c,:], u[b,:]>  (optionally both L2-normalised): one wave per (b,c)
__global__ __launch_bounds__(256) void dot_scoring_kernel(const float* u, const float* c, float* r, int64_t n_pairs, int C,
                                                           int E, int normalize) {
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = pair / C;
  const float* up = u + b * E;
  const float* cp = c + pair * E;
  float dot = 0.f, uu = 0.f, cc = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float x = up[e], y = cp[e];
    dot = fmaf(x, y, dot);
    uu = fmaf(x, x, uu);
    cc = fmaf(y, y, cc);
  }
  dot = wave_sum(dot);
  if (normalize) {
    uu = wave_sum(uu);
    cc = wave_sum(cc);
    dot = dot / (sqrtf(uu) * sqrtf(cc));
  }
  if (lane == 0) r[pair] = dot;
}

hipError_t launch_dot_scoring(const float* u, const float* c, float* r, int64_t B, int32_t C, int32_t E, int32_t normalize,
                              hipStream_t stream) {
  const int64_t n_pairs = B * C;
  if (n_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(dot_scoring_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, stream, u, c, r, n_pairs, C, E,
                     normalize);
  return hipGetLastError();
}

}  // namespace xnrs
