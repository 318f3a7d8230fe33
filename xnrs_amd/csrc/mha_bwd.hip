// Backward of the attention core (autograd of xnrs/models/components/layers.py:138-151).
//
//   P  = softmax(rowmask(Q K^T / sqrt(dk)))          (recomputed from Q, K and the saved row {max,sum})
//   Pd = dropout(P)                                   (mask recomputed from the counter RNG)
//   O  = Pd V
//   dV = Pd^T dO ;  dPd = dO V^T ;  dP = dropout'(dPd)
//   dS = P * (dP - delta),  delta = rowsum(dO * O)    (softmax backward, flash-attention form)
//   dS = 0 on rows with mask == 0 (masked_fill blocks the gradient), dS /= sqrt(dk)
//   dQ = dS K ;  dK = dS^T Q
//
// Three kernels, all fp32 MFMA 16x16x4, no LDS, no atomics (bitwise reproducible):
//   delta : thread per (row, head)
//   dq    : wave per (seq, head, 16-query tile); queries on the lanes' column index (the forward's
//           layout), so dS is directly the B operand of dQ^T = K^T dS^T.
//   dkdv  : wave per (seq, head, 16-key tile); keys on the lanes' column index, loops over the query
//           tiles and keeps dK^T / dV^T of its 16 keys in accumulators, so nothing is summed across
//           waves.  Pd and dS are directly the B operands of dV^T = dO^T Pd and dK^T = Q^T dS.
#include <cstdlib>

#include "kernels.h"

namespace xnrs {

constexpr int MAX_FT = 8;  // d_k <= 128

// delta[(seq*h + hd)*S + query] = sum_j O[row, hd*dk + j] * dO[row, hd*dk + j].  One workgroup per token row:
// the threads sweep the D columns (coalesced 1-KB reads), products go to LDS, one thread per head sums its
// d_k products in index order.  (The first version gave every (row, head) to one thread reading 192-B pieces
// 3 KB apart: 1 981 us for 80 000 rows; this form is bandwidth-bound.)
__global__ __launch_bounds__(256) void mha_delta_kernel(MhaBwdArgs a, int64_t n_rows) {
  __shared__ float s_p[2048];
  const int D = a.n_heads * a.d_k;
  for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const float* o = a.o + row * a.ldo;
    const float* d = a.d_o + row * a.lddo;
    for (int c = threadIdx.x; c < D; c += 256) s_p[c] = o[c] * d[c];
    __syncthreads();
    const int64_t seq = row / a.S;
    const int query = (int)(row - seq * a.S);
    for (int hd = threadIdx.x; hd < a.n_heads; hd += 256) {
      float acc = 0.f;
      for (int e = 0; e < a.d_k; ++e) acc += s_p[hd * a.d_k + e];
      a.delta[(seq * a.n_heads + hd) * (int64_t)a.S + query] = acc;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ f32x4 load4(const float* p, int f0, int lim, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    if (f0 < lim) v = *reinterpret_cast<const f32x4*>(p + f0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (f0 + e < lim) v[e] = p[f0 + e];
  }
  return v;
}

__device__ __forceinline__ void store4(float* p, int f0, int lim, bool vec, f32x4 v, bool acc = false) {
  if (vec) {
    if (f0 < lim) {
      if (acc) v += *reinterpret_cast<const f32x4*>(p + f0);
      *reinterpret_cast<f32x4*>(p + f0) = v;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (f0 + e < lim) p[f0 + e] = acc ? p[f0 + e] + v[e] : v[e];
  }
}

template <int KT, bool VEC>
__global__ __launch_bounds__(256) void mha_dq_kernel(MhaBwdArgs a, int QT, int64_t n_units) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int qt = (int)(unit % QT);
  const int64_t uh = unit / QT;
  const int hd = (int)(uh % a.n_heads);
  const int64_t seq = uh / a.n_heads;
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, dk = a.d_k;
  const int64_t row0 = seq * S;
  const int hoff = hd * dk;
  const int query = qt * 16 + c;
  const bool qvalid = query < S;
  const int64_t qrow_i = row0 + (qvalid ? query : 0);

  f32x4 s[KT], dp[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* qrow = a.q + qrow_i * a.ld + hoff;
  const float* dorow = a.d_o + qrow_i * a.lddo + hoff;
  const int nfb = (dk + 15) >> 4;
  for (int fb = 0; fb < nfb; ++fb) {
    const int f0 = fb * 16 + 4 * g;
    f32x4 qf = {0.f, 0.f, 0.f, 0.f}, df = {0.f, 0.f, 0.f, 0.f};
    if (qvalid) {
      qf = load4(qrow, f0, dk, VEC);
      df = load4(dorow, f0, dk, VEC);
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int key = kt * 16 + c;
      f32x4 kf = {0.f, 0.f, 0.f, 0.f}, vf = {0.f, 0.f, 0.f, 0.f};
      if (key < S) {
        kf = load4(a.k + (row0 + key) * a.ld + hoff, f0, dk, VEC);
        vf = load4(a.v + (row0 + key) * a.ld + hoff, f0, dk, VEC);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[e], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], df[e], dp[kt], 0, 0, 0);
      }
    }
  }
  float mq = 1.f, mx = 0.f, sum = 1.f, delta = 0.f;
  if (qvalid) {
    if (a.mask) {
      const int64_t mrow = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0;
      mq = a.mask[mrow + query];
    }
    const int64_t si = (seq * a.n_heads + hd) * (int64_t)S + query;
    mx = a.stats[2 * si];
    sum = a.stats[2 * si + 1];
    delta = a.delta[si];
  }
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  const float inv_sum = 1.f / sum;
  const float keep = 1.f - a.dropout_p;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const float sv = s[kt][r] * inv_sq;
      float ds = 0.f;
      if (mq != 0.f && key < S && qvalid) {
        const float p = attn_exp(sv - mx) * inv_sum;
        float dpv = dp[kt][r];
        if (a.dropout_p > 0.f) {
          dpv = (drop_uniform(a, seq, hd, S, query, key) < keep) ? dpv / keep : 0.f;
        }
        ds = p * (dpv - delta) * inv_sq;
      }
      s[kt][r] = ds;
    }
  }
  // dQ^T[f][query] = sum_key K^T[f][key] dS^T[key][query]
  const int nft = (dk + 15) >> 4;
  for (int ft = 0; ft < nft; ++ft) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int fa = ft * 16 + c;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float kk[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        kk[r] = (key < S && fa < dk) ? a.k[(row0 + key) * a.ld + hoff + fa] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[r], s[kt][r], o, 0, 0, 0);
    }
    if (qvalid) store4(a.dq + (row0 + query) * a.ldd + hoff, ft * 16 + 4 * g, dk, VEC, o, a.accumulate != 0);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void mha_dkdv_kernel(MhaBwdArgs a, int KTn, int64_t n_units) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int kt0 = (int)(unit % KTn);
  const int64_t uh = unit / KTn;
  const int hd = (int)(uh % a.n_heads);
  const int64_t seq = uh / a.n_heads;
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, dk = a.d_k;
  const int64_t row0 = seq * S;
  const int hoff = hd * dk;
  const int key = kt0 * 16 + c;
  const bool kvalid = key < S;
  const int64_t krow_i = row0 + (kvalid ? key : 0);
  const float* krow = a.k + krow_i * a.ld + hoff;
  const float* vrow = a.v + krow_i * a.ld + hoff;
  const int nfb = (dk + 15) >> 4;
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  const float keep = 1.f - a.dropout_p;
  const int64_t mrow = a.mask ? (a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0) : 0;
  const int64_t sbase = (seq * a.n_heads + hd) * (int64_t)S;

  f32x4 dkT[MAX_FT], dvT[MAX_FT];
#pragma unroll
  for (int t = 0; t < MAX_FT; ++t) {
    dkT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int QT = (S + 15) >> 4;
  for (int qt = 0; qt < QT; ++qt) {
    const int qa = qt * 16 + c;  // the query this lane feeds as an A-operand row
    const bool qa_valid = qa < S;
    const int64_t qa_row = row0 + (qa_valid ? qa : 0);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
    for (int fb = 0; fb < nfb; ++fb) {
      const int f0 = fb * 16 + 4 * g;
      f32x4 qf = {0.f, 0.f, 0.f, 0.f}, df = {0.f, 0.f, 0.f, 0.f}, kf = {0.f, 0.f, 0.f, 0.f}, vf = {0.f, 0.f, 0.f, 0.f};
      if (qa_valid) {
        qf = load4(a.q + qa_row * a.ld + hoff, f0, dk, VEC);
        df = load4(a.d_o + qa_row * a.lddo + hoff, f0, dk, VEC);
      }
      if (kvalid) {
        kf = load4(krow, f0, dk, VEC);
        vf = load4(vrow, f0, dk, VEC);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[e], kf[e], s, 0, 0, 0);    // S[query 4g+r][key c]
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(df[e], vf[e], dp, 0, 0, 0);  // dPd[query 4g+r][key c]
      }
    }
    f32x4 pd, ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int query = qt * 16 + 4 * g + r;
      float pdv = 0.f, dsv = 0.f;
      if (query < S && kvalid) {
        const float mq = a.mask ? a.mask[mrow + query] : 1.f;
        const float mx = a.stats[2 * (sbase + query)];
        const float sum = a.stats[2 * (sbase + query) + 1];
        const float delta = a.delta[sbase + query];
        float sv = s[r] * inv_sq;
        if (mq == 0.f) sv = -1e9f;
        const float p = attn_exp(sv - mx) * (1.f / sum);
        float dpv = dp[r];
        pdv = p;
        if (a.dropout_p > 0.f) {
          const bool kp = drop_uniform(a, seq, hd, S, query, key) < keep;
          pdv = kp ? p / keep : 0.f;
          dpv = kp ? dpv / keep : 0.f;
        }
        if (mq != 0.f) dsv = p * (dpv - delta) * inv_sq;
      }
      pd[r] = pdv;
      ds[r] = dsv;
    }
    // dV^T[dv][key] += dO^T[dv][query] Pd[query][key];  dK^T[f][key] += Q^T[f][query] dS[query][key]
#pragma unroll
    for (int t = 0; t < MAX_FT; ++t) {
      if (t < nfb) {
        const int fa = t * 16 + c;
        float dd[4], qq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int query = qt * 16 + 4 * g + r;
          const bool ok = query < S && fa < dk;
          dd[r] = ok ? a.d_o[(row0 + query) * a.lddo + hoff + fa] : 0.f;
          qq[r] = ok ? a.q[(row0 + query) * a.ld + hoff + fa] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dvT[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dd[r], pd[r], dvT[t], 0, 0, 0);
          dkT[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[r], ds[r], dkT[t], 0, 0, 0);
        }
      }
    }
  }
  if (kvalid) {
    float* dkrow = a.dk + (row0 + key) * a.ldd + hoff;
    float* dvrow = a.dv + (row0 + key) * a.ldd + hoff;
#pragma unroll
    for (int t = 0; t < MAX_FT; ++t) {
      if (t < nfb) {
        store4(dkrow, t * 16 + 4 * g, dk, VEC, dkT[t], a.accumulate != 0);
        store4(dvrow, t * 16 + 4 * g, dk, VEC, dvT[t], a.accumulate != 0);
      }
    }
  }
}

// Fused dQ / dK / dV for S <= 64, d_k <= 64 (16-byte aligned): one workgroup per (sequence, head), wave w owns key
// tile w.  S and dPd are computed ONCE per (query tile, key tile) -- the two-kernel form computes them in the dq
// AND in the dkdv kernel -- and every operand is staged once:
//   * the head's Q and dO rows go to LDS as row-major images (row stride 52 floats: the row-fragment read --
//     lane (c, g) takes 16 B of row c -- and the transposed fragment read -- 4 x ds_read_b32 of rows 4g+r,
//     column c -- are both conflict-free; 16*NFB + 4 in general), read from global in row order (192-byte runs
//     at d_k = 48);
//   * the wave's 16 keys of K (row and transposed fragments) and V (row fragments) stay in registers;
//   * per query tile: S, dPd (keys on the lanes' column), Pd and dS in registers, dV^T += dO^T Pd and
//     dK^T += Q^T dS from the transposed LDS reads; dS is transposed through a 1-KB wave-private LDS tile so that
//     it is the B operand of dQ^T(partial) = K^T dS^T; the four waves' partial dQ tiles meet in LDS and ONE wave
//     adds them in key-tile order (deterministic, no atomics) and stores the rows.
// 960 instead of 1 344 MFMAs per (sequence, head) at S = 50, d_k = 48, no scalar gathers from global.
constexpr int BWD_TLD = 20;  // row stride of the dS transpose tile
#ifndef BWD_PBUF
#define BWD_PBUF 1  // partial-dQ buffers: 1 (+ one more barrier per query tile, 45 KB -> 3 workgroups/CU) or 2 (58 KB -> 2/CU)
#endif

template <int NFB>
__global__ __launch_bounds__(256) void mha_bwd_fused_kernel(MhaBwdArgs a) {
  // LDS row stride (floats) of the Q / dO / partial-dQ images: 16*NFB + 4 keeps 16 consecutive rows on 16 distinct
  // 16-byte bank groups (20, 36, 52, 68 floats) and 4 rows apart on banks +16
  constexpr int BWD_LD = 16 * NFB + 4;
  __shared__ __attribute__((aligned(16))) float Qs[64 * BWD_LD];
  __shared__ __attribute__((aligned(16))) float Ds[64 * BWD_LD];
  __shared__ __attribute__((aligned(16))) float Ts[4][16 * BWD_TLD];
  __shared__ __attribute__((aligned(16))) float Ps[BWD_PBUF][4][16 * BWD_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  // XCD-aware pair order (kernels.h xcd_pair): heads of a sequence on one XCD, sequences round-robin over the XCDs
  const int64_t pair = xcd_pair(blockIdx.x, gridDim.x, a.n_heads);
  const int hd = (int)(pair % a.n_heads);
  const int64_t seq = pair / a.n_heads;
  const int S = a.S, dk = a.d_k;
  const int64_t row0 = seq * S;
  const int hoff = hd * dk;
  const int QT = (S + 15) >> 4;  // query tiles = key tiles
  const bool active = wave < QT;
  if (a.dead_seq_mode && a.masked_do_is_zero && a.mask) {  // (kernel arguments: uniform)
    const int64_t mr = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0;
    if (!__syncthreads_or(tid < S && a.mask[mr + tid] != 0.f)) {
      // every query row masked: dO = 0 and dS = 0 on every row, so dQ = dK = dV = 0 -- written without reading anything
      // (mode 1), or left alone when the caller reads these rows nowhere (mode 2)
      if (a.dead_seq_mode == 1 && !a.accumulate) {
        for (int idx = tid; idx < S * (NFB * 4); idx += 256) {
          const int row = idx / (NFB * 4), f0 = (idx - row * (NFB * 4)) * 4;
          if (f0 < dk) {
            const int64_t off = (row0 + row) * a.ldd + hoff + f0;
            *reinterpret_cast<f32x4*>(a.dq + off) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(a.dk + off) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(a.dv + off) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      return;
    }
  }

  // ---- stage Q and dO of this head (rows >= S and features >= d_k as zeros); delta = rowsum(dO * O) of the head is
  // taken on the way (a separate pass over dO and O -- mha_delta_kernel -- read 590 MB per 96 000-row encode for it):
  // one partial per 16-byte chunk into LDS (the dS transpose tiles are idle until the first query tile), summed per row
  // in chunk order below
  constexpr int NCH = NFB * 4;
  __shared__ float s_delta[64];
  // round 4: the query mask and the saved softmax statistics {max, 1 / sum} of this head's queries, read from global ONCE
  // (the q-tile loop read them per lane and per tile: 12 dependent global loads in front of every exp)
  __shared__ float s_mq[64], s_mx[64], s_isum[64];
  if (tid < 64) {
    const int64_t mrow0 = a.mask ? (a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0) : 0;
    const int64_t sb0 = (seq * a.n_heads + hd) * (int64_t)S;
    const bool in = tid < S;
    s_mq[tid] = (in && a.mask) ? a.mask[mrow0 + tid] : (in ? 1.f : 0.f);
    s_mx[tid] = in ? a.stats[2 * (sb0 + tid)] : 0.f;
    s_isum[tid] = in ? 1.f / a.stats[2 * (sb0 + tid) + 1] : 0.f;
  }
  float* s_dpart = &Ts[0][0];  // 64 x NCH <= 1024 floats of the 1280
  static_assert(64 * NFB * 4 <= 4 * 16 * BWD_TLD, "delta partials must fit the transpose tiles");
  for (int idx = tid; idx < 64 * NCH; idx += 256) {
    const int row = idx / NCH, f0 = (idx - row * NCH) * 4;
    f32x4 qv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
    if (row < S && f0 < dk) {
      qv = *reinterpret_cast<const f32x4*>(a.q + (row0 + row) * a.ld + hoff + f0);
      dv = *reinterpret_cast<const f32x4*>(a.d_o + (row0 + row) * a.lddo + hoff + f0);
      ov = *reinterpret_cast<const f32x4*>(a.o + (row0 + row) * a.ldo + hoff + f0);
    }
    *reinterpret_cast<f32x4*>(&Qs[row * BWD_LD + f0]) = qv;
    *reinterpret_cast<f32x4*>(&Ds[row * BWD_LD + f0]) = dv;
    s_dpart[idx] = fmaf(dv[3], ov[3], fmaf(dv[2], ov[2], fmaf(dv[1], ov[1], dv[0] * ov[0])));
  }
  // ---- this wave's key tile in registers
  const int key = wave * 16 + c;
  const bool kvalid = active && key < S;
  f32x4 kf[NFB], vf[NFB], kT[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    const int f0 = fb * 16 + 4 * g;
    kf[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
    vf[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kvalid && f0 < dk) {
      kf[fb] = *reinterpret_cast<const f32x4*>(a.k + (row0 + key) * a.ld + hoff + f0);
      vf[fb] = *reinterpret_cast<const f32x4*>(a.v + (row0 + key) * a.ld + hoff + f0);
    }
    // transposed fragment: element e = K[16w + 4g + e][16fb + c]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = wave * 16 + 4 * g + e, fa = fb * 16 + c;
      kT[fb][e] = (active && kk < S && fa < dk) ? a.k[(row0 + kk) * a.ld + hoff + fa] : 0.f;
    }
  }
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  const float keep = 1.f - a.dropout_p;
  f32x4 dkT[NFB], dvT[NFB];
#pragma unroll
  for (int t = 0; t < NFB; ++t) {
    dkT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  if (tid < 64) {
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) acc += s_dpart[tid * NCH + ch];
    s_delta[tid] = acc;
  }
  __syncthreads();

  for (int qt = 0; qt < QT; ++qt) {
    if (a.masked_do_is_zero && a.mask) {
      // all 16 queries of this tile masked (or beyond S): S / dPd / dS are dead and dO is zero, so nothing reaches
      // dK, dV or dQ -- uniform over the workgroup (depends on qt only), so the barriers below are skipped together
      const int qq = qt * 16 + c;
      const bool lv = qq < S && s_mq[qq] != 0.f;
      if (!__any(lv)) {
        if (wave == (qt & 3) && !a.accumulate) {
          for (int idx = lane; idx < 16 * NCH; idx += 64) {
            const int qr = idx / NCH, f0 = (idx - qr * NCH) * 4;
            if (qt * 16 + qr < S && f0 < dk)
              *reinterpret_cast<f32x4*>(a.dq + (row0 + qt * 16 + qr) * a.ldd + hoff + f0) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        continue;
      }
    }
    if (active) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(&Qs[(qt * 16 + c) * BWD_LD + fb * 16 + 4 * g]);
        const f32x4 df = *reinterpret_cast<const f32x4*>(&Ds[(qt * 16 + c) * BWD_LD + fb * 16 + 4 * g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[e], kf[fb][e], s, 0, 0, 0);    // S[query 4g+r][key c]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(df[e], vf[fb][e], dp, 0, 0, 0);  // dPd[query 4g+r][key c]
        }
      }
      f32x4 pd, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int query = qt * 16 + 4 * g + r;
        float pdv = 0.f, dsv = 0.f;
        if (query < S && kvalid) {
          const float mq = s_mq[query];
          const float mx = s_mx[query];
          const float delta = s_delta[query];
          float sv = s[r] * inv_sq;
          if (mq == 0.f) sv = -1e9f;
          const float p = attn_exp(sv - mx) * s_isum[query];
          float dpv = dp[r];
          pdv = p;
          if (a.dropout_p > 0.f) {
            const bool kp = drop_uniform(a, seq, hd, S, query, key) < keep;
            pdv = kp ? p / keep : 0.f;
            dpv = kp ? dpv / keep : 0.f;
          }
          if (mq != 0.f) dsv = p * (dpv - delta) * inv_sq;
        }
        pd[r] = pdv;
        ds[r] = dsv;
      }
      // dV^T[dv][key] += dO^T[dv][query] Pd[query][key];  dK^T[f][key] += Q^T[f][query] dS[query][key]
#pragma unroll
      for (int t = 0; t < NFB; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dd = Ds[(qt * 16 + 4 * g + r) * BWD_LD + t * 16 + c];
          const float qq = Qs[(qt * 16 + 4 * g + r) * BWD_LD + t * 16 + c];
          dvT[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dd, pd[r], dvT[t], 0, 0, 0);
          dkT[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq, ds[r], dkT[t], 0, 0, 0);
        }
      }
      // dS^T through the wave-private tile: written [query 4g+r][key c], read back [query c][keys 4g .. 4g+3]
      float* T = Ts[wave];
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * g + r) * BWD_TLD + c] = ds[r];
      const f32x4 dsT = *reinterpret_cast<const f32x4*>(&T[c * BWD_TLD + 4 * g]);
      // partial dQ^T[f][query] = K^T[f][key] dS^T[key][query] over this wave's 16 keys
      float* P = Ps[qt & (BWD_PBUF - 1)][wave];
#pragma unroll
      for (int t = 0; t < NFB; ++t) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(kT[t][e], dsT[e], o, 0, 0, 0);
        *reinterpret_cast<f32x4*>(&P[c * BWD_LD + t * 16 + 4 * g]) = o;  // query c, features 16t+4g .. +3
      }
    }
    __syncthreads();
    {
      // sum the key tiles' partials in order and store the 16 query rows (row-order 16-byte chunks).  Round 4: by ALL four
      // waves (192 chunks over 256 threads) -- one wave did it while three waited at the barrier below
      for (int idx = tid; idx < 16 * NCH; idx += 256) {
        const int qr = idx / NCH, f0 = (idx - qr * NCH) * 4;
        const int query = qt * 16 + qr;
        if (query < S && f0 < dk) {
          f32x4 acc = *reinterpret_cast<const f32x4*>(&Ps[qt & (BWD_PBUF - 1)][0][qr * BWD_LD + f0]);
          for (int w2 = 1; w2 < QT; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Ps[qt & (BWD_PBUF - 1)][w2][qr * BWD_LD + f0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
          }
          float* dst = a.dq + (row0 + query) * a.ldd + hoff + f0;
          if (a.accumulate) acc += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = acc;
        }
      }
    }
    if (BWD_PBUF == 1) __syncthreads();  // the single partial buffer is rewritten by the next query tile
  }
  if (kvalid) {
    float* dkrow = a.dk + (row0 + key) * a.ldd + hoff;
    float* dvrow = a.dv + (row0 + key) * a.ldd + hoff;
#pragma unroll
    for (int t = 0; t < NFB; ++t) {
      const int f0 = t * 16 + 4 * g;
      if (f0 < dk) {
        f32x4 kk = dkT[t], vv = dvT[t];
        if (a.accumulate) {
          kk += *reinterpret_cast<const f32x4*>(dkrow + f0);
          vv += *reinterpret_cast<const f32x4*>(dvrow + f0);
        }
        *reinterpret_cast<f32x4*>(dkrow + f0) = kk;
        *reinterpret_cast<f32x4*>(dvrow + f0) = vv;
      }
    }
  }
}

template <int KT>
static hipError_t launch_dq(const MhaBwdArgs& a, bool vec, hipStream_t stream) {
  const int QT = (a.S + 15) / 16;
  const int64_t n_units = a.n_seq * a.n_heads * QT;
  const dim3 grid((unsigned)((n_units + 3) / 4));
  if (vec) hipLaunchKernelGGL((mha_dq_kernel<KT, true>), grid, dim3(256), 0, stream, a, QT, n_units);
  else hipLaunchKernelGGL((mha_dq_kernel<KT, false>), grid, dim3(256), 0, stream, a, QT, n_units);
  return hipGetLastError();
}

hipError_t launch_mha_bwd(const MhaBwdArgs& a, hipStream_t stream) {
  if (a.n_seq <= 0 || a.S <= 0) return hipSuccess;
  if (a.S > 128 || a.d_k > 16 * MAX_FT) return hipErrorInvalidValue;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = (a.d_k % 4 == 0) && (a.ld % 4 == 0) && (a.lddo % 4 == 0) && (a.ldd % 4 == 0) && al16(a.q) && al16(a.k) &&
                   al16(a.v) && al16(a.d_o) && al16(a.dq) && al16(a.dk) && al16(a.dv);
  if (a.n_heads * a.d_k > 2048) return hipErrorInvalidValue;
  const int64_t n_rows = a.n_seq * a.S;
  hipError_t e = hipSuccess;
  const int KT = (a.S + 15) / 16;
  const bool fused = vec && a.S <= 64 && a.d_k <= 64 && a.ldo % 4 == 0 && al16(a.o) && knobs().mha_bwd_fused;
  if (!fused) {  // the fused kernel takes delta = rowsum(dO * O) while it stages dO
    const unsigned dgrid = (unsigned)(n_rows < 65536 ? n_rows : 65536);
    hipLaunchKernelGGL(mha_delta_kernel, dim3(dgrid), dim3(256), 0, stream, a, n_rows);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (fused) {  // (knobs().mha_bwd_fused: development knob for A/B runs; default on)
    const int64_t n_pairs = a.n_seq * a.n_heads;
    if (n_pairs > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 fgrid((unsigned)n_pairs);
    switch ((a.d_k + 15) / 16) {
      case 1: hipLaunchKernelGGL((mha_bwd_fused_kernel<1>), fgrid, dim3(256), 0, stream, a); break;
      case 2: hipLaunchKernelGGL((mha_bwd_fused_kernel<2>), fgrid, dim3(256), 0, stream, a); break;
      case 3: hipLaunchKernelGGL((mha_bwd_fused_kernel<3>), fgrid, dim3(256), 0, stream, a); break;
      default: hipLaunchKernelGGL((mha_bwd_fused_kernel<4>), fgrid, dim3(256), 0, stream, a); break;
    }
    return hipGetLastError();
  }
  switch (KT) {
    case 1: e = launch_dq<1>(a, vec, stream); break;
    case 2: e = launch_dq<2>(a, vec, stream); break;
    case 3: e = launch_dq<3>(a, vec, stream); break;
    case 4: e = launch_dq<4>(a, vec, stream); break;
    case 5: e = launch_dq<5>(a, vec, stream); break;
    case 6: e = launch_dq<6>(a, vec, stream); break;
    case 7: e = launch_dq<7>(a, vec, stream); break;
    default: e = launch_dq<8>(a, vec, stream); break;
  }
  if (e != hipSuccess) return e;
  const int64_t n_units = a.n_seq * a.n_heads * KT;
  const dim3 grid((unsigned)((n_units + 3) / 4));
  if (vec) hipLaunchKernelGGL((mha_dkdv_kernel<true>), grid, dim3(256), 0, stream, a, KT, n_units);
  else hipLaunchKernelGGL((mha_dkdv_kernel<false>), grid, dim3(256), 0, stream, a, KT, n_units);
  return hipGetLastError();
}

}  // namespace xnrs
