// Backward kernels of the pooling / scoring stages and the column reductions that produce bias and
// fc2 gradients.  fp32, deterministic (no float atomics: two-stage ordered reductions).
//   additive_pool_bwd : autograd of layers.AdditiveAttention.forward after fc1+tanh (layers.py:60-65)
//   mean_pool_bwd     : autograd of layers.MaskedMean.forward (layers.py:35-36)
//   colsum            : out[n] = sum_m w[m] * X[m][n]  (bias grads: w == NULL; fc2.weight grad: X = tanh(fc1 x), w = de)
//   dot_scoring_bwd   : autograd of DotScoring.forward (scoring.py:20-23), plain and L2-normalised
#include "kernels.h"

namespace xnrs {

__device__ __forceinline__ float wave_sum_b(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int POOLB_MAX_N = 512;

// forward: s_i = exp(e_i) m_i, Z = sum s + 1e-8, a_i = s_i / Z, p = sum a_i x_i  with  e_i = w2.t_i + b2
// backward (dp given):  da_i = dp.x_i ;  de_i = a_i (da_i - sum_j a_j da_j)
//   dx_i  = a_i dp                        (the fc1 path adds dpre.W1 on top, as a GEMM)
//   dpre_i = de_i * w2 * (1 - t_i^2)      (gradient at the fc1 pre-activation)
// VEC: D, A, the row pitches and every pointer are multiples of 16 bytes (launcher): rows move as 16-byte chunks and the
// (row, chunk) loops are flattened over the workgroup -- 38 + 13 store instructions per thread at 50 x 768 / A = 256 instead
// of 150 + 50 four-byte ones; the scalar form ran at 2.4 TB/s of its own traffic, latency-bound on the per-row loops.
template <bool VEC>
__global__ __launch_bounds__(256) void additive_pool_bwd_kernel(AdditivePoolBwdArgs a) {
  __shared__ float s_a[POOLB_MAX_N];
  __shared__ float s_da[POOLB_MAX_N];
  __shared__ float s_red[4];
  const int64_t seq = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, A = a.A, D = a.D;
  const int64_t srcx = a.x_gather_ids ? (int64_t)a.x_gather_ids[seq] : seq;
  const float* x = a.x + srcx * N * a.ldx;
  const float* dp = a.dp + seq * D;
  float shift = a.da_shift ? a.da_shift[seq] : 0.f;
  for (int i = tid; i < N; i += 256) s_a[i] = a.a[seq * N + i];
  if (a.shift_u) {  // (kernel argument: uniform) shift = <shift_u[seq], shift_v>, summed in a fixed order
    const float* u = a.shift_u + seq * D;
    float p = 0.f;
    for (int d = tid; d < D; d += 256) p = fmaf(u[d], a.shift_v[d], p);
    p = wave_sum_b(p);
    if (lane == 0) s_red[wave] = p;
    __syncthreads();
    shift = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  __syncthreads();
  // rows with a_i == 0 (masked tokens: exp(e) * 0) contribute nothing -- de_i = a_i (...) = 0 whatever da_i is -- so their
  // value rows and tanh rows are not read at all (55 % of the rows of the benchmark batch), only their zeros written
  for (int i = wave; i < N; i += 4) {
    if (s_a[i] == 0.f) {  // (wave-uniform)
      if (lane == 0) s_da[i] = 0.f;
      continue;
    }
    const float* xi = x + (int64_t)i * a.ldx;
    float acc = 0.f;
    if constexpr (VEC) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(xi);
      const f32x4* d4 = reinterpret_cast<const f32x4*>(dp);
      for (int c = lane; c < (D >> 2); c += 64) {
        const f32x4 xv = x4[c], dv = d4[c];
        acc = fmaf(dv[3], xv[3], fmaf(dv[2], xv[2], fmaf(dv[1], xv[1], fmaf(dv[0], xv[0], acc))));
      }
    } else {
      for (int d = lane; d < D; d += 64) acc = fmaf(dp[d], xi[d], acc);
    }
    acc = wave_sum_b(acc);
    if (lane == 0) s_da[i] = acc + shift;
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < N; i += 256) part = fmaf(s_a[i], s_da[i], part);
  part = wave_sum_b(part);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float cdot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    const float de = s_a[i] * (s_da[i] - cdot);
    s_da[i] = de;
    a.de[seq * N + i] = de;
  }
  __syncthreads();
  if constexpr (VEC) {
    if (a.dx) {
      const int D4 = D >> 2;
      const f32x4* d4 = reinterpret_cast<const f32x4*>(dp);
      int i = 0, c = tid;
      while (c >= D4) { c -= D4; ++i; }
      while (i < N) {
        const float ai = s_a[i];
        const f32x4 dv = d4[c];
        *reinterpret_cast<f32x4*>(a.dx + (seq * N + i) * a.lddx + 4 * c) = f32x4{ai * dv[0], ai * dv[1], ai * dv[2], ai * dv[3]};
        c += 256;
        while (c >= D4) { c -= D4; ++i; }
      }
    }
    const int A4 = A >> 2;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w2);
    int i = 0, c = tid;
    while (c >= A4) { c -= A4; ++i; }
    while (i < N) {
      const float de = s_da[i];
      f32x4 out = {0.f, 0.f, 0.f, 0.f};
      if (s_a[i] != 0.f) {  // (de is exactly 0 otherwise: dpre = 0 without reading tanh(fc1 x_i))
        const f32x4 tv = *reinterpret_cast<const f32x4*>(a.t + (seq * N + i) * (int64_t)A + 4 * c);
        const f32x4 wv = w4[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = de * wv[e] * (1.f - tv[e] * tv[e]);
      }
      *reinterpret_cast<f32x4*>(a.dpre + (seq * N + i) * (int64_t)A + 4 * c) = out;
      c += 256;
      while (c >= A4) { c -= A4; ++i; }
    }
  } else {
    if (a.dx) {
      for (int i = 0; i < N; ++i) {
        const float ai = s_a[i];
        float* dxi = a.dx + (seq * N + i) * a.lddx;
        for (int d = tid; d < D; d += 256) dxi[d] = ai * dp[d];
      }
    }
    for (int i = 0; i < N; ++i) {
      const float de = s_da[i];
      float* dpre = a.dpre + (seq * N + i) * (int64_t)A;
      if (s_a[i] == 0.f) {  // (uniform) de is exactly 0: dpre = 0 without reading tanh(fc1 x_i)
        for (int k = tid; k < A; k += 256) dpre[k] = 0.f;
        continue;
      }
      const float* ti = a.t + (seq * N + i) * (int64_t)A;
      for (int k = tid; k < A; k += 256) {
        const float tv = ti[k];
        dpre[k] = de * a.w2[k] * (1.f - tv * tv);
      }
    }
  }
}

hipError_t launch_additive_pool_bwd(const AdditivePoolBwdArgs& a, hipStream_t stream) {
  if (a.n_seq <= 0) return hipSuccess;
  if (a.N > POOLB_MAX_N || a.N <= 0) return hipErrorInvalidValue;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = a.D % 4 == 0 && a.A % 4 == 0 && a.ldx % 4 == 0 && (!a.dx || (a.lddx % 4 == 0 && al16(a.dx))) && al16(a.x) &&
                   al16(a.dp) && al16(a.t) && al16(a.w2) && al16(a.dpre);
  if (vec) hipLaunchKernelGGL(additive_pool_bwd_kernel<true>, dim3((unsigned)a.n_seq), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(additive_pool_bwd_kernel<false>, dim3((unsigned)a.n_seq), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// y = sum_i x_i m_i / (sum m + 1e-8)  ->  dx_i = dy * m_i / (sum m + 1e-8)
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* dy, const float* mask, const int32_t* mask_ids,
                                                             float* dx, int64_t lddx, int N, int D) {
  __shared__ float s_red[4];
  const int64_t seq = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t src = mask_ids ? (int64_t)mask_ids[seq] : seq;
  float part = 0.f;
  for (int i = tid; i < N; i += 256) part += mask[src * N + i];
  part = wave_sum_b(part);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float denom = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) + 1e-8f;
  for (int i = 0; i < N; ++i) {
    const float w = mask[src * N + i] / denom;
    for (int d = tid; d < D; d += 256) dx[(seq * N + i) * lddx + d] = dy[seq * D + d] * w;
  }
}

hipError_t launch_mean_pool_bwd(const float* dy, const float* mask, const int32_t* mask_ids, float* dx, int64_t lddx,
                                int64_t n_seq, int32_t N, int32_t D, hipStream_t stream) {
  if (n_seq <= 0) return hipSuccess;
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, stream, dy, mask, mask_ids, dx, lddx, N, D);
  return hipGetLastError();
}

// ---- column sums: stage 1 writes partial[split][n] for blocks of COLSUM_ROWS rows (enough blocks to fill the
// chip: the first version used 64 row slices and ran 462 us for 80 000 x 768), stage 2 sums the splits in
// order -> bitwise reproducible.
constexpr int COLSUM_ROWS = 128;
constexpr int COLSUM_MAX_SPLITS = 4096;

// NX = N, or N + 1: one more column of ones -- its sum is the sum of the row weights themselves (the additive pooler's fc2
// BIAS gradient sum_r de_r beside its weight gradient sum_r de_r t_r: one pass over de instead of two launch pairs)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* X, int64_t ldx, const float* w, int64_t M, int N,
                                                              int64_t rows_per, float* partial, const float* X2, int64_t ldx2,
                                                              const float* w2, int64_t M1, int NX) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int sp = blockIdx.y;
  const int64_t r0 = sp * rows_per;
  const int64_t r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  if (n >= NX) return;
  const bool ones = n >= N;
  // rows >= M1 come from the second block (launch_colsum2; M1 = M without one)
  // a row whose weight is exactly 0 is not read (block-uniform test): fmaf(0, x, acc) = acc for every finite x -- the same
  // bits -- and the weighted sums of the grad step (w = de: 0 on every masked token row, 45-55 % of a MIND batch) stream
  // half the bytes
  auto term = [&](int64_t r, float acc) {
    const bool second = r >= M1;
    const float* ww = second ? w2 : w;
    const float wv = ww ? ww[second ? r - M1 : r] : 1.f;
    if (ww && wv == 0.f) return acc;
    const float x = ones ? 1.f : (second ? X2[(r - M1) * ldx2 + n] : X[r * ldx + n]);
    return ww ? fmaf(wv, x, acc) : acc + x;
  };
  // four independent chains (rows r, r+1, r+2, r+3 of every group of four), combined in a fixed order: the loads of a
  // slice are in flight together instead of one row per memory latency
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = term(r + e, a4[e]);
  }
  for (int e = 0; r < r1; ++r, ++e) a4[e] = term(r, a4[e]);
  const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  partial[(int64_t)sp * NX + n] = acc;
}

// one wave per column: lane l adds partials l, l+64, ... in order, then a fixed shuffle tree
// (out_last: column N - 1 goes there instead -- the weight-sum column of launch_colsum_wsum)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int nsplit, int N, float* out, float* out_last) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int s = lane; s < nsplit; s += 64) acc += partial[(int64_t)s * N + n];
  acc = wave_sum_b(acc);
  if (lane == 0) {
    if (out_last && n == N - 1) out_last[0] = acc;
    else out[n] = acc;
  }
}

hipError_t launch_colsum_final(const float* partial, int nsplit, int N, float* out, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, partial, nsplit, N, out,
                     (float*)nullptr);
  return hipGetLastError();
}

size_t colsum_workspace_bytes(int N) { return (size_t)COLSUM_MAX_SPLITS * (size_t)N * sizeof(float); }

static hipError_t colsum_impl(const float* X, int64_t ldx, const float* w, int64_t M1, const float* X2, int64_t ldx2, const float* w2,
                              int64_t M2, int N, float* out, float* wsum_out, float* partial, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  if (wsum_out && (!w || X2)) return hipErrorInvalidValue;
  const int NX = N + (wsum_out ? 1 : 0);
  const int64_t M = M1 + (X2 ? M2 : 0);
  // rows per slice: enough slices to fill the chip (~1024 workgroups with the column blocks), 8 .. COLSUM_ROWS rows each --
  // a 320-row reduction used to run as three workgroups walking 128 rows serially (31 us for 0.3 MB)
  const int64_t col_blocks = (NX + 255) / 256;
  const int64_t want = col_blocks >= 1024 ? 1 : 1024 / col_blocks;
  int64_t rows_per = (M + want - 1) / want;
  if (rows_per < 8) rows_per = 8;
  if (rows_per > COLSUM_ROWS) rows_per = COLSUM_ROWS;
  // (80 000 rows x 256 columns -- the fc2 weight gradient of a history tower -- ran as 1 013 slices of 79 rows, 20 dependent
  // groups of four loads per thread: 78 us for 82 MB; 32-row slices keep four times as many workgroups in flight)
  if (rows_per > 32 && M >= 16384) rows_per = 32;
  if ((M + rows_per - 1) / rows_per > COLSUM_MAX_SPLITS) rows_per = (M + COLSUM_MAX_SPLITS - 1) / COLSUM_MAX_SPLITS;
  int nsplit = (int)((M + rows_per - 1) / rows_per);
  if (nsplit < 1) nsplit = 1;
  const dim3 g1((unsigned)((NX + 255) / 256), (unsigned)nsplit);
  hipLaunchKernelGGL(colsum_partial_kernel, g1, dim3(256), 0, stream, X, ldx, w, M, N, rows_per, partial, X2, ldx2, w2, M1, NX);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((NX + 3) / 4)), dim3(256), 0, stream, partial, nsplit, NX, out, wsum_out);
  return hipGetLastError();
}

hipError_t launch_colsum2(const float* X, int64_t ldx, const float* w, int64_t M1, const float* X2, int64_t ldx2, const float* w2,
                          int64_t M2, int N, float* out, float* partial, hipStream_t stream) {
  return colsum_impl(X, ldx, w, M1, X2, ldx2, w2, M2, N, out, nullptr, partial, stream);
}

// out[n] = sum_r w[r] X[r][n] AND wsum_out[0] = sum_r w[r] in one pass (partial: colsum_workspace_bytes(N + 1))
hipError_t launch_colsum_wsum(const float* X, int64_t ldx, const float* w, int64_t M, int N, float* out, float* wsum_out,
                              float* partial, hipStream_t stream) {
  return colsum_impl(X, ldx, w, M, nullptr, 0, nullptr, 0, N, out, wsum_out, partial, stream);
}

hipError_t launch_colsum(const float* X, int64_t ldx, const float* w, int64_t M, int N, float* out, float* partial,
                         hipStream_t stream) {
  return launch_colsum2(X, ldx, w, M, nullptr, 0, nullptr, 0, N, out, partial, stream);
}

// r[b,c] = <c[b,c,:], u[b,:]>  ->  du[b,e] = sum_c dr[b,c] c[b,c,e] ;  dc[b,c,e] = dr[b,c] u[b,e]
__global__ __launch_bounds__(256) void dot_scoring_bwd_kernel(const float* u, const float* c, const float* dr, float* du,
                                                               float* dc, int64_t B, int C, int E) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * E) return;
  const int64_t b = i / E;
  const int e = (int)(i - b * E);
  const float uv = u[i];
  float acc = 0.f;
  for (int k = 0; k < C; ++k) {
    const float g = dr[b * C + k];
    acc = fmaf(g, c[(b * C + k) * E + e], acc);
    if (dc) dc[(b * C + k) * E + e] = g * uv;
  }
  if (du) du[i] = acc;
}

// normalize=True (scoring.py:20-22): r_k = <c_k, u> / (|c_k| |u|).  With uh = u/|u|, ch = c_k/|c_k|:
//   dr_k/du = (ch - r_k uh) / |u| ,  dr_k/dc_k = (uh - r_k ch) / |c_k|
// one workgroup per impression; wave w takes candidates w, w+4, ...; per-wave partial du rows summed in a fixed order
constexpr int DOTN_MAX_E = 1024;
__global__ __launch_bounds__(256) void dot_scoring_norm_bwd_kernel(const float* u, const float* c, const float* dr, float* du,
                                                                    float* dc, int C, int E) {
  __shared__ float s_du[4][DOTN_MAX_E];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ub = u + b * E;
  float nu2 = 0.f;
  for (int e = lane; e < E; e += 64) {
    nu2 = fmaf(ub[e], ub[e], nu2);
    s_du[wave][e] = 0.f;
  }
  nu2 = wave_sum_b(nu2);
  const float inv_u = 1.f / sqrtf(nu2);
  for (int k = wave; k < C; k += 4) {
    const float* ck = c + (b * C + k) * E;
    float nc2 = 0.f, dot = 0.f;
    for (int e = lane; e < E; e += 64) {
      nc2 = fmaf(ck[e], ck[e], nc2);
      dot = fmaf(ck[e], ub[e], dot);
    }
    nc2 = wave_sum_b(nc2);
    dot = wave_sum_b(dot);
    const float inv_c = 1.f / sqrtf(nc2);
    const float r = dot * inv_u * inv_c;
    const float g = dr[b * C + k];
    for (int e = lane; e < E; e += 64) {
      const float uh = ub[e] * inv_u, ch = ck[e] * inv_c;
      if (dc) dc[(b * C + k) * E + e] = g * (uh - r * ch) * inv_c;
      s_du[wave][e] += g * (ch - r * uh) * inv_u;
    }
  }
  __syncthreads();
  if (du)
    for (int e = threadIdx.x; e < E; e += 256) du[b * E + e] = (s_du[0][e] + s_du[1][e]) + (s_du[2][e] + s_du[3][e]);
}

hipError_t launch_dot_scoring_norm_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B,
                                       int32_t C, int32_t E, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  if (E > DOTN_MAX_E || B > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dot_scoring_norm_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, u, c, dr, du, dc, C, E);
  return hipGetLastError();
}

hipError_t launch_dot_scoring_bwd(const float* u, const float* c, const float* dr, float* du, float* dc, int64_t B, int32_t C,
                                  int32_t E, hipStream_t stream) {
  const int64_t n = B * E;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(dot_scoring_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, c, dr, du, dc, B, C,
                     E);
  return hipGetLastError();
}

}  // namespace xnrs

namespace xnrs {

// d_table[r, :] = sum over m with ids[m] == r of d_rows[m, :]   (nn.Embedding backward for the small
// category tables of NAML, naml.py:82-86).  One workgroup per table row: its four waves take the four quarters of the id
// list, each reads 64 ids at a time, finds its row's occurrences with a ballot and adds their gradient rows (lanes =
// features) in list order; the four partial rows are added in wave order.  Deterministic, no atomics.  (Round 4: the first
// version -- one wave walking the ids one by one, once per 64 features -- took 80-90 us per table at M = 1 600: 0.5 of the
// 3.2 ms of the NAML grad step.)
constexpr int EMB_MAX_PASSES = 8;  // features per lane: K <= 512 on the fast kernel

__global__ __launch_bounds__(256) void embedding_grad_kernel(const float* __restrict__ d_rows, const int32_t* __restrict__ ids,
                                                              int64_t M, int K, float* __restrict__ d_table) {
  __shared__ float s_part[4][64 * EMB_MAX_PASSES];
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per = (M + 3) / 4;
  const int64_t m0 = wave * per, m1 = (m0 + per < M) ? m0 + per : M;
  float acc[EMB_MAX_PASSES];
#pragma unroll
  for (int p = 0; p < EMB_MAX_PASSES; ++p) acc[p] = 0.f;
  for (int64_t base = m0; base < m1; base += 64) {
    const int64_t m = base + lane;
    const bool hit = m < m1 && ids[m] == r;
    unsigned long long mask = __ballot(hit);
    while (mask) {  // (wave-uniform)
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float* row = d_rows + (base + b) * K;
#pragma unroll
      for (int p = 0; p < EMB_MAX_PASSES; ++p) {
        const int k = p * 64 + lane;
        if (k < K) acc[p] += row[k];
      }
    }
  }
#pragma unroll
  for (int p = 0; p < EMB_MAX_PASSES; ++p) s_part[wave][p * 64 + lane] = acc[p];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    d_table[(int64_t)r * K + k] = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
}

// any K: one wave per table row, the ids one by one
__global__ __launch_bounds__(64) void embedding_grad_wide_kernel(const float* d_rows, const int32_t* ids, int64_t M, int K,
                                                                  float* d_table) {
  const int r = blockIdx.x;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + threadIdx.x;
    float acc = 0.f;
    for (int64_t m = 0; m < M; ++m)
      if (ids[m] == r && k < K) acc += d_rows[m * K + k];
    if (k < K) d_table[(int64_t)r * K + k] = acc;
  }
}

hipError_t launch_embedding_grad(const float* d_rows, const int32_t* ids, int64_t M, int K, float* d_table, int n_rows,
                                 hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  if (K <= 64 * EMB_MAX_PASSES)
    hipLaunchKernelGGL(embedding_grad_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, d_rows, ids, M, K, d_table);
  else
    hipLaunchKernelGGL(embedding_grad_wide_kernel, dim3((unsigned)n_rows), dim3(64), 0, stream, d_rows, ids, M, K, d_table);
  return hipGetLastError();
}

}  // namespace xnrs
