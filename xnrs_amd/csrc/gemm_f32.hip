// fp32 Linear on the CDNA4 matrix cores:  C = act(A . W^T + bias)
//
// Replaces nn.Linear as used by the reference hot path (xnrs/models/components/layers.py:60,
// 128-130,154; news_encoding.py:27-31).  Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf
// chain (no TF32/xf32 on gfx950), so results differ from torch-CPU only by summation order.
//
// Design (gfx950, 64-wide waves):
//   * workgroup = 256 threads = 4 waves in a 2x2 grid; wave tile (32*TM)x(32*TN), i.e. the block
//     tile is (64*TM)x(64*TN); BK = 32.
//   * both operands are K-contiguous (x rows, nn.Linear weight rows), so a lane fetches 4
//     consecutive k with ONE ds_read_b128 and feeds 4 MFMA steps from it: MFMA step j of an
//     8-wide k group uses k = 4*(lane>>5) + j for A and B alike (any bijection of k is legal as
//     long as A and B agree).  Per 8 k: TM+TN LDS reads vs 4*TM*TN MFMAs (64 cycles each).
//   * LDS rows padded to 36 floats: the 16-lane groups of ds_read_b128 then hit 64 distinct banks.
//   * register-staged double buffering (issue global loads for tile t+1, compute tile t, then write
//     LDS) with one barrier per K tile; 2 workgroups/CU co-reside (73.7 KB LDS each).
//   * optional row gather on A (device-resident news-token table + ids: SURVEY.md section 8 a0) folded into
//     the per-thread row pointers, so gathered rows are still read as full 128-B lines.
//   * XCD-aware block order: the 8 XCDs each walk a contiguous range of tiles with the N tiles of
//     one M tile adjacent, so an A tile is fetched into one L2 only.
#include "kernels.h"

namespace xnrs {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // padded row length in floats (144 B, 16-B aligned)

template <int TM, int TN, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs a, int m_tiles, int n_tiles_seg) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread per operand tile
  __shared__ __attribute__((aligned(16))) float As[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDS_LD];

  // ---- XCD-aware tile order (bijective for any grid size)
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_tiles = n_tiles_seg * a.nseg;
  const int mt = wgid / n_tiles;
  const int nt = wgid - mt * n_tiles;
  const int seg = nt / n_tiles_seg;
  const int nts = nt - seg * n_tiles_seg;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nts * BN;  // column inside the segment

  const float* __restrict__ W = a.W[seg];
  const float* __restrict__ bias = a.bias[seg];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = tid & 7;   // 16-byte chunk inside the 32-float k tile
  const int lr = tid >> 3;  // 0..31

  // ---- per-thread source rows
  const float* pa[AR];
  const float* pb[BR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int64_t gr = m0 + lr + 32 * i;
    if (gr < a.M) {
      int64_t src = gr;
      if (a.gather_ids) {
        const int64_t n = gr / a.gather_S;
        src = (int64_t)a.gather_ids[n] * a.gather_S + (gr - n * a.gather_S);
      }
      pa[i] = a.A + src * a.lda + 4 * lc;
    } else {
      pa[i] = nullptr;
    }
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int col = n0 + lr + 32 * i;
    pb[i] = (col < a.Nseg) ? (W + (int64_t)col * a.ldw + 4 * lc) : nullptr;
  }

  f32x4 ra[AR], rb[BR];
  auto gload = [&](int k0) {
    const int k = k0 + 4 * lc;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pa[i]) {
        if (VEC) {
          if (k < a.K) v = *reinterpret_cast<const f32x4*>(pa[i] + k0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < a.K) v[e] = pa[i][k0 + e];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pb[i]) {
        if (VEC) {
          if (k < a.K) v = *reinterpret_cast<const f32x4*>(pb[i] + k0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < a.K) v[e] = pb[i][k0 + e];
        }
      }
      rb[i] = v;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(&As[buf][lr + 32 * i][4 * lc]) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][lr + 32 * i][4 * lc]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;        // row of the 32x32 operand tile this lane feeds
  const int fk = (lane >> 5) * 4;    // k offset inside an 8-wide k group
  const int a_row0 = wm * 32 * TM + frow;
  const int b_row0 = wn * 32 * TN + frow;

  const int nk = (a.K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) gload((t + 1) * BK);
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&As[buf][a_row0 + 32 * i][kq * 8 + fk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][b_row0 + 32 * j][kq * 8 + fk]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int ccol = lane & 31;
  const int crow = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 32 * TN + 32 * j + ccol;
    if (col >= a.Nseg) continue;
    const float bv = bias ? bias[col] : 0.f;
    float* cbase = a.C + (int64_t)seg * a.Nseg + col;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
        if (row < a.M) {
          float v = acc[i][j][e] + bv;
          if (a.act == 1) v = fmaxf(v, 0.f);
          else if (a.act == 2) v = tanhf(v);
          cbase[row * a.ldc] = v;
        }
      }
    }
  }
}

template <int TM, int TN>
static hipError_t launch_cfg(const GemmArgs& a, bool vec, hipStream_t stream) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int64_t m_tiles = (a.M + BM - 1) / BM;
  const int n_tiles_seg = (a.Nseg + BN - 1) / BN;
  const int64_t grid = m_tiles * n_tiles_seg * a.nseg;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, true>), dim3((unsigned)grid), dim3(256), 0, stream, a, (int)m_tiles,
                       n_tiles_seg);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false>), dim3((unsigned)grid), dim3(256), 0, stream, a, (int)m_tiles,
                       n_tiles_seg);
  return hipGetLastError();
}

hipError_t launch_gemm_f32(const GemmArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.Nseg <= 0 || a.K <= 0) return hipSuccess;
  // 16-byte vector loads need K-contiguous rows on 16-B boundaries
  bool vec = (a.K % 4 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
  for (int s = 0; s < a.nseg; ++s) vec = vec && ((reinterpret_cast<uintptr_t>(a.W[s]) & 15) == 0);

  // pick the tile with the least estimated time: rounds of co-resident workgroups x padded tile work
  // (2 workgroups/CU x 256 CUs per round); bigger tiles have slightly better MFMA duty.
  const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const double eff[4] = {1.0, 0.94, 0.94, 0.86};
  int best = 0;
  double best_t = 1e300;
  for (int c = 0; c < 4; ++c) {
    const int64_t bm = 64 * cand[c][0], bn = 64 * cand[c][1];
    const int64_t wgs = ((a.M + bm - 1) / bm) * ((a.Nseg + bn - 1) / bn) * a.nseg;
    const double rounds = (double)((wgs + 511) / 512);
    const double t = rounds * (double)(bm * bn) / eff[c];
    if (t < best_t * 0.999) {
      best_t = t;
      best = c;
    }
  }
  switch (best) {
    case 0: return launch_cfg<2, 2>(a, vec, stream);
    case 1: return launch_cfg<2, 1>(a, vec, stream);
    case 2: return launch_cfg<1, 2>(a, vec, stream);
    default: return launch_cfg<1, 1>(a, vec, stream);
  }
}

}  // namespace xnrs
