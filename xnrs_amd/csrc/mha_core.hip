// Attention core of layers.MultiHeadAttention.forward (xnrs/models/components/layers.py:133-153):
// per (sequence, head):  P = softmax(rowmask(Q K^T / sqrt(d_k)));  O = P V   -- fp32, MFMA 16x16x4.
//
// Parity-critical semantics reproduced exactly (SURVEY.md finding 4):
//   * the mask is a QUERY-ROW mask (layers.py:142-144): a row with m==0 is filled with -1e9 for ALL
//     keys, so its softmax is exactly uniform 1/S; valid rows attend over all S keys incl. padded.
//   * softmax is max-stabilised like torch.softmax; the scale is a true division by sqrt(d_k).
//
// Work decomposition: one wave per (sequence, head, 16-query tile); 4 waves per workgroup.  S <= 128
// (8 key tiles).  No LDS: all operands go global -> VGPR in MFMA layout.
//   * S^T tile = K_h . Q_h^T is computed (keys on the MFMA rows, queries on the lanes' column),
//     so after the MFMA lane (c = lane&15, g = lane>>4) holds S[query c][key 16*kt + 4g + r],
//     r = 0..3: the softmax reduction over keys is in-lane + two wavefront shuffles (xor 16, 32),
//     and the probabilities are ALREADY in the B-operand layout of the second product
//     O^T = V_h^T . P^T  (k index = key), so P never leaves registers.
//   * Q/K fragments are 16-byte loads (4 consecutive features per lane; MFMA step r uses feature
//     16*fb + 4g + r for both operands), V fragments are dword loads with the 16 lanes of a group on
//     16 consecutive floats.
//   * the output comes out as O^T[dv = 16*dt + 4g + r][query c]: 4 consecutive dv per lane -> one
//     16-byte store per lane straight into the (M, D) "concat heads" layout (layers.py:153).
#include <cstdlib>

#include "kernels.h"

namespace xnrs {

template <int KT, bool VEC>
__global__ __launch_bounds__(256) void mha_core_kernel(MhaCoreArgs a, int QT, int64_t n_units) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;  // wave-uniform
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

  // ---------------- S^T = K Q^T
  f32x4 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t hbase = seq * a.seq_stride + hd * a.head_stride;  // (seq, head) block of q / k / v
  const float* qrow = a.q + hbase + (int64_t)(qvalid ? query : 0) * a.ld;
  const int nfb = (dk + 15) >> 4;
  for (int fb = 0; fb < nfb; ++fb) {
    const int f0 = fb * 16 + 4 * g;
    f32x4 qf = {0.f, 0.f, 0.f, 0.f};
    if (qvalid) {
      if (VEC) {
        if (f0 < dk) qf = *reinterpret_cast<const f32x4*>(qrow + f0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (f0 + e < dk) qf[e] = qrow[f0 + e];
      }
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int key = kt * 16 + c;
      f32x4 kf = {0.f, 0.f, 0.f, 0.f};
      if (key < S) {
        const float* krow = a.k + hbase + (int64_t)key * a.ld;
        if (VEC) {
          if (f0 < dk) kf = *reinterpret_cast<const f32x4*>(krow + f0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (f0 + e < dk) kf[e] = krow[f0 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[e], acc[kt], 0, 0, 0);
    }
  }

  // ---------------- scale, row mask, softmax over keys (acc[kt][r] = S[query][16kt+4g+r])
  float mq = 1.f;
  if (a.mask && qvalid) {
    const int64_t mrow = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0;
    mq = a.mask[mrow + query];
  }
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float s = acc[kt][r] * inv_sq;
      if (mq == 0.f) s = -1e9f;
      if (key >= S) s = -INFINITY;
      acc[kt][r] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = attn_exp(acc[kt][r] - mx);  // exp(-inf) = 0 for padding keys
      acc[kt][r] = e;
      sum += e;
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (a.stats && qvalid && g == 0) {
    float* st = a.stats + ((seq * a.n_heads + hd) * (int64_t)S + query) * 2;
    st[0] = mx;
    st[1] = sum;
  }
  const float keep = 1.f - a.dropout_p;
  const float inv_sum = 1.f / sum;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = acc[kt][r] * inv_sum;
      if (a.dropout_p > 0.f) {  // nn.Dropout on the probabilities (layers.py:148), train mode only
        const int key = kt * 16 + 4 * g + r;
        p = (drop_uniform(a, seq, hd, S, query, key) < keep) ? p / keep : 0.f;
      }
      acc[kt][r] = p;
    }
  }

  // ---------------- O^T = V^T P^T
  const int ndt = (dk + 15) >> 4;
  for (int dt = 0; dt < ndt; ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int dv_a = dt * 16 + c;  // the V column this lane feeds as MFMA row
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float vv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        vv[r] = (key < S && dv_a < dk) ? a.v[hbase + (int64_t)key * a.ld + dv_a] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], acc[kt][r], o, 0, 0, 0);
    }
    // o[r] = O[query c][dv = 16dt + 4g + r]
    if (qvalid) {
      const int dv0 = dt * 16 + 4 * g;
      float* orow = a.out + (row0 + query) * a.ldo + hoff + dv0;
      if (VEC) {
        if (dv0 < dk) *reinterpret_cast<f32x4*>(orow) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (dv0 + r < dk) orow[r] = o[r];
      }
    }
  }
}

// Fast path (S <= 64, d_k <= 64, 16-byte aligned): one wave per (sequence, head).  K and V of the head
// are fetched ONCE into registers in MFMA operand layout (K: 16-byte row fragments; V: the transposed
// gather the second product needs) and all query tiles are swept against them, so the per-q-tile
// traffic is only Q in and O out.  Same arithmetic, same order of operations per element as
// mha_core_kernel (bitwise identical results).
template <int KT, int NFB>
__global__ __launch_bounds__(256) void mha_core_head_kernel(MhaCoreArgs a, int64_t n_units) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;  // wave-uniform
  const int hd = (int)(unit % a.n_heads);
  const int64_t seq = unit / a.n_heads;
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, dk = a.d_k;
  const int64_t row0 = seq * S;
  const int hoff = hd * dk;
  const int ld = (int)a.ld;

  // ---- K fragments: kf[kt][fb] = K[16kt + c][16fb + 4g .. +3]
  const int64_t hbase = seq * a.seq_stride + hd * a.head_stride;
  const float* kbase = a.k + hbase;
  const float* vbase = a.v + hbase;
  f32x4 kf[KT][NFB];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kt * 16 + c;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      const int f0 = fb * 16 + 4 * g;
      kf[kt][fb] = (key < S && f0 < dk) ? *reinterpret_cast<const f32x4*>(kbase + key * ld + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- V gather: vv[kt][dt][r] = V[16kt + 4g + r][16dt + c]
  float vv[KT][NFB][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < NFB; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const int dv = dt * 16 + c;
        vv[kt][dt][r] = (key < S && dv < dk) ? vbase[key * ld + dv] : 0.f;
      }

  const int64_t mrow = a.mask ? (a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0) : 0;
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  const float keep = 1.f - a.dropout_p;
  const int QT = (S + 15) >> 4;
  for (int qt = 0; qt < QT; ++qt) {
    const int query = qt * 16 + c;
    const bool qvalid = query < S;
    const float* qrow = a.q + hbase + (int64_t)(qvalid ? query : 0) * a.ld;
    f32x4 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      const int f0 = fb * 16 + 4 * g;
      const f32x4 qf = (qvalid && f0 < dk) ? *reinterpret_cast<const f32x4*>(qrow + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][fb][e], qf[e], acc[kt], 0, 0, 0);
    }
    const float mq = (a.mask && qvalid) ? a.mask[mrow + query] : 1.f;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float sv = acc[kt][r] * inv_sq;
        if (mq == 0.f) sv = -1e9f;
        if (key >= S) sv = -INFINITY;
        acc[kt][r] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ev = attn_exp(acc[kt][r] - mx);
        acc[kt][r] = ev;
        sum += ev;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (a.stats && qvalid && g == 0) {
      float* st = a.stats + ((seq * a.n_heads + hd) * (int64_t)S + query) * 2;
      st[0] = mx;
      st[1] = sum;
    }
    const float inv_sum = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = acc[kt][r] * inv_sum;
        if (a.dropout_p > 0.f) {
          const int key = kt * 16 + 4 * g + r;
          p = (drop_uniform(a, seq, hd, S, query, key) < keep) ? p / keep : 0.f;
        }
        acc[kt][r] = p;
      }
#pragma unroll
    for (int dt = 0; dt < NFB; ++dt) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[kt][dt][r], acc[kt][r], o, 0, 0, 0);
      const int dv0 = dt * 16 + 4 * g;
      if (qvalid && dv0 < dk) *reinterpret_cast<f32x4*>(a.out + (row0 + query) * a.ldo + hoff + dv0) = o;
    }
  }
}

template <int KT, int NFB>
static hipError_t launch_head(const MhaCoreArgs& a, hipStream_t stream) {
  const int64_t n_units = a.n_seq * a.n_heads;
  const int64_t grid = (n_units + 3) / 4;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((mha_core_head_kernel<KT, NFB>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_units);
  return hipGetLastError();
}

template <int KT>
static hipError_t launch_head_kt(const MhaCoreArgs& a, hipStream_t stream) {
  switch ((a.d_k + 15) / 16) {
    case 1: return launch_head<KT, 1>(a, stream);
    case 2: return launch_head<KT, 2>(a, stream);
    case 3: return launch_head<KT, 3>(a, stream);
    default: return launch_head<KT, 4>(a, stream);
  }
}

// LDS-staged path (S <= 64, d_k <= 64): a workgroup serves G = 4/QTP (sequence, head) pairs; the QTP waves of
// a pair first stage K and V of the head into LDS cooperatively, ALREADY IN MFMA FRAGMENT ORDER (slot
// ((kt*NFB + fb)*64 + lane) holds exactly the float4 that lane feeds to the MFMAs of key tile kt and
// feature block fb -- for V the four keys 16kt+4g+r of column 16dt+c, i.e. the transposed gather of the
// second product), so every fragment read is a linear, conflict-free ds_read_b128 and the padding (keys
// >= S, features >= d_k) is written as zeros once.  Then each wave owns one 16-query tile.  Compared
// with the head-per-wave kernel: K/V are loaded once per head by 4x as many lanes, the query tiles run in
// parallel, and a wave needs ~64 instead of 176 VGPRs (6 instead of 2 waves per SIMD to hide latency).
template <int KT, int NFB, int QTP>
__global__ __launch_bounds__(256) void mha_core_lds_kernel(MhaCoreArgs a, int64_t n_pairs) {
  constexpr int G = 4 / QTP;          // pairs per workgroup
  constexpr int NTHR = QTP * 64;      // threads per pair
  constexpr int NSLOT = KT * NFB * 64;
  __shared__ __attribute__((aligned(16))) f32x4 Ks[G][NSLOT];
  __shared__ __attribute__((aligned(16))) f32x4 Vs[G][NSLOT];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int gp = tid / NTHR;           // pair slot inside the workgroup
  const int tp = tid - gp * NTHR;      // thread inside the pair
  const int qt = tp >> 6;              // this wave's query tile
  const int64_t pair = (int64_t)blockIdx.x * G + gp;
  const bool pvalid = pair < n_pairs;
  const int hd = pvalid ? (int)(pair % a.n_heads) : 0;
  const int64_t seq = pvalid ? pair / a.n_heads : 0;
  const int S = a.S, dk = a.d_k;
  const int64_t row0 = seq * S;
  const int hoff = hd * dk;
  const int ld = (int)a.ld;
  const int64_t hbase = seq * a.seq_stride + hd * a.head_stride;
  const int64_t kvbase = (a.q_off && a.kv_block) ? (int64_t)a.kv_block[seq] * a.seq_stride + hd * a.head_stride : hbase;
  const float* kbase = a.k + kvbase;
  const float* vbase = a.v + kvbase;

  // this wave's query fragments and mask value are fetched FIRST so their latency overlaps the K/V staging
  // (they do not depend on it)
  // unpadded queries: this sequence's live rows are the compact range [q0, q0 + nq) of q / out
  const int64_t q0 = a.q_off ? a.q_off[seq] : row0;
  const int nq = a.q_off ? (pvalid ? (int)(a.q_off[seq + 1] - q0) : 0) : S;
  const int QT = (nq + 15) >> 4;
  const int c = lane & 15, g = lane >> 4;
  const int query = qt * 16 + c;
  const bool qvalid = pvalid && qt < QT && query < nq;
  const float* qrow = a.q_off ? a.q + (q0 + (qvalid ? query : 0)) * a.ldq + hd * dk
                              : a.q + hbase + (int64_t)(qvalid ? query : 0) * a.ld;
  f32x4 qf[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    const int f0 = fb * 16 + 4 * g;
    qf[fb] = (qvalid && f0 < dk) ? *reinterpret_cast<const f32x4*>(qrow + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool use_mask = a.mask && !a.q_off;
  const int64_t mrow = use_mask ? (a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : row0) : 0;
  const float mq = (use_mask && qvalid) ? a.mask[mrow + query] : 1.f;

  if (pvalid) {
    // Both operands are READ in row order as 16-byte chunks (192-B runs per head row at d_k = 48) and scattered
    // into fragment order.  Slot of fragment (tile t, block b, lane (c, g)) = (t*NFB + b)*64 + g*16 + (c ^ x):
    // the XOR (K: x = 4b + g, V: x = b) spreads the chunk-order writes over the banks and keeps every
    // fragment read (16 lanes = 16 consecutive c) conflict-free.
    constexpr int NCH = NFB * 4;  // 16-byte chunks per padded row
    float* vs = reinterpret_cast<float*>(&Vs[gp][0]);
    for (int idx = tp; idx < KT * 16 * NCH; idx += NTHR) {
      const int key = idx / NCH, ch = idx - key * NCH, f0 = ch * 4;
      const bool ok = key < S && f0 < dk;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        kv = *reinterpret_cast<const f32x4*>(kbase + key * ld + f0);
        vv = *reinterpret_cast<const f32x4*>(vbase + key * ld + f0);
      }
      const int kt = key >> 4, kc = key & 15, b = ch >> 2, g4 = ch & 3;
      // K fragment (kt, fb = b, lane (c = kc, g = g4)) = K[16kt + c][16fb + 4g .. +3]
      Ks[gp][(kt * NFB + b) * 64 + g4 * 16 + (kc ^ ((4 * b + g4) & 15))] = kv;
      // V fragment (kt, dt = b, lane (c = 4*g4 + j, g = kc >> 2)) element r = kc & 3  = V[16kt + 4g + r][16dt + c]
      float* dst = vs + ((kt * NFB + b) * 64 + (kc >> 2) * 16) * 4 + (kc & 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[((4 * g4 + j) ^ b) * 4] = vv[j];
    }
  }
  __syncthreads();
  if (!pvalid || qt >= QT) return;

  f32x4 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const f32x4 kf = Ks[gp][(kt * NFB + fb) * 64 + g * 16 + (c ^ ((4 * fb + g) & 15))];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[fb][e], acc[kt], 0, 0, 0);
    }
  }
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float sv = acc[kt][r] * inv_sq;
      if (mq == 0.f) sv = -1e9f;
      if (key >= S) sv = -INFINITY;
      acc[kt][r] = sv;
      mx = fmaxf(mx, sv);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ev = attn_exp(acc[kt][r] - mx);
      acc[kt][r] = ev;
      sum += ev;
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (a.stats && qvalid && g == 0) {
    float* st = a.stats + ((seq * a.n_heads + hd) * (int64_t)S + query) * 2;
    st[0] = mx;
    st[1] = sum;
  }
  const float inv_sum = 1.f / sum;
  const float keep = 1.f - a.dropout_p;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = acc[kt][r] * inv_sum;
      if (a.dropout_p > 0.f) {
        const int key = kt * 16 + 4 * g + r;
        p = (drop_uniform(a, seq, hd, S, query, key) < keep) ? p / keep : 0.f;
      }
      acc[kt][r] = p;
    }
#pragma unroll
  for (int dt = 0; dt < NFB; ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const f32x4 vf = Vs[gp][(kt * NFB + dt) * 64 + g * 16 + (c ^ dt)];
#pragma unroll
      for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], acc[kt][r], o, 0, 0, 0);
    }
    const int dv0 = dt * 16 + 4 * g;
    if (qvalid && dv0 < dk) *reinterpret_cast<f32x4*>(a.out + (q0 + query) * a.ldo + hoff + dv0) = o;
  }
}

template <int KT, int NFB, int QTP>
static hipError_t launch_lds(const MhaCoreArgs& a, hipStream_t stream) {
  constexpr int G = 4 / QTP;
  const int64_t n_pairs = a.n_seq * a.n_heads;
  const int64_t grid = (n_pairs + G - 1) / G;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((mha_core_lds_kernel<KT, NFB, QTP>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_pairs);
  return hipGetLastError();
}

// Pair-per-workgroup kernel, second generation (S <= 64, d_k <= 64): the LDS-staged kernel above
// spent, per wave at S = 50 / d_k = 48, 96 MFMAs (64 x 64 padded scores for 50 x 50) and 633 VALU instructions, the two
// adding up rather than overlapping (profiles/r01_rocprof_news_encoder_pass.txt: 0.83 ms per 5 240 news against a 0.42 ms
// matrix floor and a 0.51 ms HBM floor).  Changes:
//   * TAIL: when S = 16 KTM + rem with rem <= 4 (S = 50: rem 2) the last `rem` keys do not get an MFMA tile of their
//     own: their rows of K and V are kept row-major in LDS, their scores are 12-FMA dot products against the query
//     fragments the lane already holds (reduced over the 4 feature groups with two shuffles) and their P.V terms are 4
//     FMAs per output block.  72 instead of 96 MFMAs per wave at S = 50.
//   * MFMAs interleave over independent accumulators (key tiles in the first product, output blocks in the second)
//     instead of running 4-16 deep dependent chains;
//   * dropout and the key-padding selects are compile-time; every thread stages one FIXED 16-byte column of K and V
//     (chunk = tid % NCH), so the LDS slot arithmetic is loop-invariant.
// Arithmetic per element is the one of the kernels above except for the summation order of the tail keys.
// Measured (tools/bench_stages.py, 5 240 news of 50 x 768, 16 heads): 0.83 ms first generation -> 0.70 ms -> 0.64 ms with
// the XCD-contiguous pair order = 5.0 TB/s of Q/K/V in + O out.  Counters of this kernel (profiles/r02_attention_core_pmc.txt):
// 531 VALU + 72 MFMA instructions per wave (~2 200 cycles of each pipe: ~42 % busy each), 62 % of HBM peak, LDS bank
// conflicts 2 % of the LDS cycles (10.7 % before V's transpose moved into registers), a wave alive
// ~11 us for ~2 us of issue: no single bound left, the rest is latency at the 8-waves-per-SIMD cap.  Tried and dropped: a
// PERSISTENT variant (grid = 4 workgroups per CU walking XCD-contiguous pair ranges, the next pair's K / V / Q chunks
// prefetched into registers during the products: 124 VGPRs -> 4 waves per SIMD) measured 0.80 ms -- with the global latency
// hidden inside the wave, the MFMA / VALU / LDS latencies of its serial instruction stream had half as many waves to
// hide behind.
template <int KTM, int NFB, bool TAIL, bool DROP>
__global__ __launch_bounds__(256, (KTM * NFB <= 9 ? 8 : 5)) void mha_core_pair_kernel(MhaCoreArgs a) {
  constexpr int NSLOT = KTM * NFB * 64;
  constexpr int NCH = NFB * 4;        // 16-byte chunks per padded row
  constexpr int KP = 256 / NCH;       // keys staged per pass
  constexpr int NKEYS = KTM * 16 + (TAIL ? 4 : 0);
  constexpr int NPASS = (NKEYS + KP - 1) / KP;
  __shared__ __attribute__((aligned(16))) f32x4 Ks[NSLOT];
  __shared__ __attribute__((aligned(16))) f32x4 Vs[NSLOT];
  __shared__ __attribute__((aligned(16))) f32x4 Kt[TAIL ? 4 * NCH : 1];  // tail keys, row-major [t][chunk]
  __shared__ __attribute__((aligned(16))) f32x4 Vt[TAIL ? 4 * NCH : 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int qt = tid >> 6;
  // XCD-aware pair order: workgroup L runs on XCD L % 8.  The heads of one sequence run on one XCD next to each other in
  // time -- a head row is 4 d_k bytes (192 B at d_k = 48), so neighbouring heads share 128-B lines that would otherwise be
  // fetched by two L2s (0.70 -> 0.65 ms).  Padded queries: every pair costs the same, so XCD x simply takes the contiguous
  // range of pairs its workgroups number (neighbouring sequences are neighbours in memory too: 1.5 % faster than dealing
  // them round-robin); unpadded queries: the work follows the masks, sequences are dealt round-robin (kernels.h xcd_pair).
  int pair;
  if (a.q_off || a.skip_dead) {  // (skip_dead: the empty history slots of a user are neighbours -- deal them out too)
    pair = (int)xcd_pair(blockIdx.x, gridDim.x, a.n_heads);
  } else {
    const int L = blockIdx.x, W = gridDim.x, x = L & 7, per = W >> 3, rm = W & 7;
    pair = per * x + (x < rm ? x : rm) + (L >> 3);
  }
  const int hd = pair % a.n_heads;
  const int seq = pair / a.n_heads;
  const int S = a.S, dk = a.d_k;
  const int ld = (int)a.ld;
  const int64_t hbase = (int64_t)seq * a.seq_stride + (int64_t)hd * a.head_stride;
  const int64_t kvbase = (a.q_off && a.kv_block) ? (int64_t)a.kv_block[seq] * a.seq_stride + (int64_t)hd * a.head_stride : hbase;
  const float* kbase = a.k + kvbase;
  const float* vbase = a.v + kvbase;
  const int rem = TAIL ? S - KTM * 16 : 0;  // 1..4 (launcher)

  // ---- this wave's mask value, the dead-sequence test, then its query fragments (their latency overlaps the staging)
  // unpadded queries (a.q_off): this sequence's live rows are the compact range [q0, q0 + nq) of q / out
  const int64_t q0 = a.q_off ? a.q_off[seq] : (int64_t)seq * S;
  const int nq = a.q_off ? (int)(a.q_off[seq + 1] - q0) : S;
  const int QT = (nq + 15) >> 4;
  if (nq == 0) return;  // (unpadded queries: a news without a live token -- nothing to stage, nothing to write; uniform)
  const int c = lane & 15, g = lane >> 4;
  const int query = qt * 16 + c;
  const bool qvalid = query < nq;
  float mq = 1.f;
  if (a.mask && !a.q_off && qvalid) {
    const int64_t mrow = a.mask_gather_ids ? (int64_t)a.mask_gather_ids[seq] * S : (int64_t)seq * S;
    mq = a.mask[mrow + query];
  }
  if (a.skip_dead && a.mask && !a.q_off) {  // (kernel argument: uniform)
    if (!__syncthreads_or(qvalid && mq != 0.f)) {  // every query row of this sequence is masked: zeros, nothing read
      if (qvalid) {
        float* orow = a.out + (q0 + query) * a.ldo + hd * dk;
#pragma unroll
        for (int dt = 0; dt < NFB; ++dt) {
          const int dv0 = dt * 16 + 4 * g;
          if (dv0 < dk) *reinterpret_cast<f32x4*>(orow + dv0) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a.stats && g == 0) {
          float* sp = a.stats + (((int64_t)seq * a.n_heads + hd) * (int64_t)S + query) * 2;
          sp[0] = -1e9f;
          sp[1] = (float)S;
        }
      }
      return;
    }
  }
  // (the query fragments are loaded BEHIND the dead-sequence test: since round 4 every pooled encoder call takes it, and a
  // dead sequence -- 49.5 % of the benchmark's history slots -- must not pull its Q rows through HBM to throw them away)
  // The same for a query TILE whose 16 rows are all masked (the 4th tile of a 30-token title in 50 slots, ...): its wave
  // stages K / V with the others, then writes zeros instead of computing rows that the pooler multiplies by 0.
  const bool tile_dead = a.skip_dead && a.mask && !a.q_off && !__any(qvalid && mq != 0.f);  // wave-uniform
  const float* qrow = a.q_off ? a.q + (q0 + (qvalid ? query : 0)) * a.ldq + hd * dk
                              : a.q + hbase + (int64_t)(qvalid ? query : 0) * ld;
  f32x4 qf[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    const int f0 = fb * 16 + 4 * g;
    qf[fb] = (qvalid && !tile_dead && f0 < dk) ? *reinterpret_cast<const f32x4*>(qrow + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging.  K: thread = (chunk ch of the row, key slot ks), keys ks, ks + KP, ...; one ds_write_b128 per chunk.
  // V: thread = (chunk vch, group vkg of FOUR consecutive keys); the 4 x 4 block is transposed in registers and its four
  // columns go out as four ds_write_b128 (the first version scattered 16 ds_write_b32 per block with a 2-way bank
  // conflict on every one: 10.7 % of the LDS cycles).  Slot of fragment (tile t, block b, lane (c, g)) =
  // (t*NFB + b)*64 + g*16 + swizzle(c): K: c ^ (4b + g); V: c ^ ((c >> 3) | ((b & 1) << 1)) -- bijections of the 16
  // slots of a (tile, block, g) group, so every fragment read stays a conflict-free ds_read_b128, chosen so that the 8
  // contiguous lanes of a store group (chunks of one or two feature blocks) hit 8 distinct 16-byte bank groups.
  {
    const int ch = tid % NCH, ks = tid / NCH;
    const int f0 = ch * 4, b = ch >> 2, g4 = ch & 3;
    const bool act = ks < KP;
    const bool fok = f0 < dk;
    const int vch = tid & 15, vkg = tid >> 4;  // V: 16 lanes per key group, the first NCH of them active
    const int vf0 = vch * 4, vb = vch >> 2, vg4 = vch & 3;
    const bool vact = vch < NCH && vkg < KTM * 4 + (TAIL ? 1 : 0);
    f32x4 kv[NPASS], vv[4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int key = ks + ps * KP;
      kv[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (act && fok && key < S) kv[ps] = *reinterpret_cast<const f32x4*>(kbase + key * ld + f0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * vkg + r;
      vv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (vact && vf0 < dk && key < S) vv[r] = *reinterpret_cast<const f32x4*>(vbase + key * ld + vf0);
    }
    const int kx = (4 * b + g4) & 15;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int key = ks + ps * KP;
      if (!act) continue;
      if (key < KTM * 16) {
        const int kt = key >> 4, kc = key & 15;
        // K fragment (kt, fb = b, lane (c = kc, g = g4)) = K[16kt + c][16fb + 4g .. +3]
        Ks[(kt * NFB + b) * 64 + g4 * 16 + (kc ^ kx)] = kv[ps];
      } else if (TAIL && key < KTM * 16 + 4) {
        Kt[(key - KTM * 16) * NCH + ch] = kv[ps];
      }
    }
    if (vact) {
      if (vkg < KTM * 4) {
        // V fragment (kt, dt = vb, lane (c = 4*vg4 + j, g = vkg & 3)) elements r = 0..3 = V[16kt + 4g + r][16dt + c]
        const int base = ((vkg >> 2) * NFB + vb) * 64 + (vkg & 3) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c4 = 4 * vg4 + j;
          Vs[base + (c4 ^ ((c4 >> 3) | ((vb & 1) << 1)))] = f32x4{vv[0][j], vv[1][j], vv[2][j], vv[3][j]};
        }
      } else if (TAIL) {  // the key group behind the full tiles: the tail keys, row-major
#pragma unroll
        for (int r = 0; r < 4; ++r) Vt[r * NCH + vch] = vv[r];
      }
    }
  }
  __syncthreads();
  if (qt >= QT) return;
  if (tile_dead) {  // every query of this tile is masked: zeros (and the statistics of masked rows) instead of the products
    if (qvalid) {
      float* orow0 = a.out + (q0 + query) * a.ldo + hd * dk;
#pragma unroll
      for (int dt = 0; dt < NFB; ++dt) {
        const int dv0 = dt * 16 + 4 * g;
        if (dv0 < dk) *reinterpret_cast<f32x4*>(orow0 + dv0) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (a.stats && g == 0) {
        float* sp = a.stats + (((int64_t)seq * a.n_heads + hd) * (int64_t)S + query) * 2;
        sp[0] = -1e9f;
        sp[1] = (float)S;
      }
    }
    return;
  }

  // ---- S^T = K Q^T over the full key tiles
  f32x4 acc[KTM];
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    f32x4 kf[KTM];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) kf[kt] = Ks[(kt * NFB + fb) * 64 + g * 16 + (c ^ ((4 * fb + g) & 15))];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][e], qf[fb][e], acc[kt], 0, 0, 0);
  }
  // tail keys: dot products against the query fragment (features 16fb + 4g .. +3 of query c in this lane)
  float st[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if constexpr (TAIL) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < rem) {
        float part = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          const f32x4 kr = Kt[t * NCH + fb * 4 + g];
#pragma unroll
          for (int e = 0; e < 4; ++e) part = fmaf(qf[fb][e], kr[e], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        st[t] = part;
      }
    }
  }

  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = acc[kt][r] * inv_sq;
      if (mq == 0.f) sv = -1e9f;
      if (!TAIL && kt == KTM - 1 && kt * 16 + 4 * g + r >= S) sv = -INFINITY;
      acc[kt][r] = sv;
      mx = fmaxf(mx, sv);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if constexpr (TAIL) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < rem) {
        st[t] = (mq == 0.f) ? -1e9f : st[t] * inv_sq;
        mx = fmaxf(mx, st[t]);
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ev = attn_exp(acc[kt][r] - mx);
      acc[kt][r] = ev;
      sum += ev;
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if constexpr (TAIL) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = (t < rem) ? attn_exp(st[t] - mx) : 0.f;
      sum += st[t];
    }
  }
  if (a.stats && qvalid && g == 0) {
    float* sp = a.stats + (((int64_t)seq * a.n_heads + hd) * (int64_t)S + query) * 2;
    sp[0] = mx;
    sp[1] = sum;
  }
  const float inv_sum = 1.f / sum;
  const float keep = 1.f - a.dropout_p;
  auto drop = [&](float p, int key) {
    if constexpr (DROP) {
      return (drop_uniform(a, seq, hd, S, query, key) < keep) ? p / keep : 0.f;
    } else {
      return p;
    }
  };
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[kt][r] = drop(acc[kt][r] * inv_sum, kt * 16 + 4 * g + r);
  if constexpr (TAIL) {
#pragma unroll
    for (int t = 0; t < 4; ++t) st[t] = drop(st[t] * inv_sum, KTM * 16 + t);
  }

  // ---- O^T = V^T P^T
  f32x4 o[NFB];
#pragma unroll
  for (int dt = 0; dt < NFB; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    f32x4 vf[NFB];
#pragma unroll
    for (int dt = 0; dt < NFB; ++dt) vf[dt] = Vs[(kt * NFB + dt) * 64 + g * 16 + (c ^ ((c >> 3) | ((dt & 1) << 1)))];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < NFB; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt][r], acc[kt][r], o[dt], 0, 0, 0);
  }
  if constexpr (TAIL) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < rem) {
#pragma unroll
        for (int dt = 0; dt < NFB; ++dt) {
          const f32x4 vr = Vt[t * NCH + dt * 4 + g];  // V[16 KTM + t][16dt + 4g .. +3] = output rows dv of this lane
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] = fmaf(st[t], vr[r], o[dt][r]);
        }
      }
    }
  }
  const int hoff = hd * dk;
  float* orow = a.out + (q0 + query) * a.ldo + hoff;
#pragma unroll
  for (int dt = 0; dt < NFB; ++dt) {
    const int dv0 = dt * 16 + 4 * g;
    if (qvalid && dv0 < dk) *reinterpret_cast<f32x4*>(orow + dv0) = o[dt];
  }
}

template <int KTM, int NFB, bool TAIL>
static hipError_t launch_pair_nfb(const MhaCoreArgs& a, hipStream_t stream) {
  const int64_t n_pairs = a.n_seq * a.n_heads;
  if (n_pairs > 0x7fffffffLL) return hipErrorInvalidValue;
  if (a.dropout_p > 0.f)
    hipLaunchKernelGGL((mha_core_pair_kernel<KTM, NFB, TAIL, true>), dim3((unsigned)n_pairs), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((mha_core_pair_kernel<KTM, NFB, TAIL, false>), dim3((unsigned)n_pairs), dim3(256), 0, stream, a);
  return hipGetLastError();
}

template <int KTM, bool TAIL>
static hipError_t launch_pair(const MhaCoreArgs& a, hipStream_t stream) {
  switch ((a.d_k + 15) / 16) {
    case 1: return launch_pair_nfb<KTM, 1, TAIL>(a, stream);
    case 2: return launch_pair_nfb<KTM, 2, TAIL>(a, stream);
    case 3: return launch_pair_nfb<KTM, 3, TAIL>(a, stream);
    default: return launch_pair_nfb<KTM, 4, TAIL>(a, stream);
  }
}

template <int KT, int QTP>
static hipError_t launch_lds_kt(const MhaCoreArgs& a, hipStream_t stream) {
  switch ((a.d_k + 15) / 16) {
    case 1: return launch_lds<KT, 1, QTP>(a, stream);
    case 2: return launch_lds<KT, 2, QTP>(a, stream);
    case 3: return launch_lds<KT, 3, QTP>(a, stream);
    default: return launch_lds<KT, 4, QTP>(a, stream);
  }
}

template <int KT>
static hipError_t launch_kt(const MhaCoreArgs& a, bool vec, hipStream_t stream) {
  const int QT = (a.S + 15) / 16;
  const int64_t n_units = a.n_seq * a.n_heads * QT;
  const int64_t grid = (n_units + 3) / 4;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (vec)
    hipLaunchKernelGGL((mha_core_kernel<KT, true>), dim3((unsigned)grid), dim3(256), 0, stream, a, QT, n_units);
  else
    hipLaunchKernelGGL((mha_core_kernel<KT, false>), dim3((unsigned)grid), dim3(256), 0, stream, a, QT, n_units);
  return hipGetLastError();
}

hipError_t launch_mha_core(const MhaCoreArgs& a, hipStream_t stream) {
  if (a.n_seq <= 0 || a.S <= 0) return hipSuccess;
  if (a.S > 128) return hipErrorInvalidValue;
  const bool vec = (a.d_k % 4 == 0) && (a.ld % 4 == 0) && (a.ldo % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.k) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  const int KT = (a.S + 15) / 16;
  // knobs().mha_headwave / mha_lds: development knobs for A/B runs
  const bool fast = vec && a.S <= 64 && a.d_k <= 64 && ((int64_t)a.S * a.ld < (1ll << 31)) && knobs().mha_headwave;
  // three or four query tiles: LDS-staged kernel (one wave per tile, K/V shared through LDS; 0.99 vs 1.06 ms
  // at S=50); one or two tiles: head-per-wave kernel (0.17 vs 0.21 ms at S=30).  XNRS_MHA_LDS=0|1 forces one.
  const bool use_lds = a.q_off ? true : (knobs().mha_lds >= 0 ? knobs().mha_lds != 0 : (KT >= 3));
  if (a.q_off && (!fast || a.stats || a.dropout_p > 0.f || a.ldq % 4 != 0)) return hipErrorInvalidValue;
  if (fast && use_lds && knobs().mha_pair && (int64_t)a.n_seq * a.S < (1ll << 31)) {
    // full key tiles through the MFMAs, a remainder of 1..4 keys through the VALU
    const int rem = a.S & 15;
    const bool tail = a.S > 16 && rem >= 1 && rem <= 4;
    const int KTM = tail ? a.S >> 4 : KT;
    if (tail && KTM == 2) return launch_pair<2, true>(a, stream);
    if (tail && KTM == 3) return launch_pair<3, true>(a, stream);
    if (!tail && KTM == 3) return launch_pair<3, false>(a, stream);
    if (!tail && KTM == 4) return launch_pair<4, false>(a, stream);
  }
  if (fast && use_lds) {
    switch (KT) {
      case 1: return launch_lds_kt<1, 1>(a, stream);
      case 2: return launch_lds_kt<2, 2>(a, stream);
      case 3: return launch_lds_kt<3, 4>(a, stream);
      default: return launch_lds_kt<4, 4>(a, stream);
    }
  }
  if (fast) {
    switch (KT) {
      case 1: return launch_head_kt<1>(a, stream);
      case 2: return launch_head_kt<2>(a, stream);
      case 3: return launch_head_kt<3>(a, stream);
      default: return launch_head_kt<4>(a, stream);
    }
  }
  switch (KT) {
    case 1: return launch_kt<1>(a, vec, stream);
    case 2: return launch_kt<2>(a, vec, stream);
    case 3: return launch_kt<3>(a, vec, stream);
    case 4: return launch_kt<4>(a, vec, stream);
    case 5: return launch_kt<5>(a, vec, stream);
    case 6: return launch_kt<6>(a, vec, stream);
    case 7: return launch_kt<7>(a, vec, stream);
    default: return launch_kt<8>(a, vec, stream);
  }
}

}  // namespace xnrs
