// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3"): forward layout only,
//   C = act(A . W^T + bias),  A row-major [M,K] (optionally row-gathered), W = nn.Linear weight [N,K].
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (157 TF vs ~2.5 PF dense) and the
// reference hot path is bound by it.  Every fp32 value is an exact sum of three
// bf16 pieces, x = hi + mid + lo (8 + 8 + 8 significant bits; bf16 has the fp32 exponent range, so no
// scaling and no overflow that fp32 itself would not have), and
//   a.b = ah.bh + (ah.bm + am.bh) + (am.bm + ah.bl + al.bh) + O(2^-26 |a||b|)
// i.e. SIX v_mfma_f32_32x32x16_bf16 with fp32 accumulation reproduce the fp32 product to better than one
// fp32 rounding, at 16/6 = 2.7x the fp32 MFMA rate (NPL = 3).  NPL = 2 keeps two pieces and three products
// (error ~3 * 2^-18 |a||b| per product, 5.3x the fp32 rate); it is a knob, not the default.
// Replaces the same nn.Linear call sites as gemm_f32.hip (layers.py:60,128-130,154; news_encoding.py:27-31).
//
// Kernel: 256 threads = 2x2 waves, block tile 128x128, wave tile 64x64 = 2x2 MFMA tiles, BK = 16 = ONE MFMA
// k step per stage.  The split happens on the way from global memory to LDS: a thread loads 8 consecutive k
// of one row (2 x dwordx4), splits them with v_cvt_pk_bf16_f32 (5.5 VALU per element) and stores one
// ds_write_b128 per plane; LDS holds NPL planes per operand, 32-byte rows, 16-byte chunk XOR-swizzled by
// (row >> 3) & 1 so that fragment reads (lane = row, 16 B) and stores are conflict-free.  Two LDS stages
// (48 KB at NPL = 3), one barrier per stage, one raw register set; the split / store / load work of the next
// stages is spread over the 24 MFMA slots of the current one (about 4 VALU per 32-cycle MFMA).
//
// Measured (MI355X, Q/K/V projection 65 500 x 2304 x 768, random data): 190 TF algorithmic at NPL = 3
// (= 1.14 PF of issued bf16 MFMA work), 260 TF at NPL = 2, against 132 TF for the fp32 kernel in the same
// process.  The kernel is power/clock-bound, not issue-bound: on all-zero operands the same binary runs
// 258 TF (the chip holds its clock), with the split arithmetic removed altogether 212 TF on random data, at
// 2 instead of 3 workgroups per CU the same 190 TF (development runs; tools/bench_split.py).
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace xnrs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// out-of-range lanes of the pointer path read this line (select on the ADDRESS, see gemm_f32.hip)
__device__ __attribute__((aligned(16))) float g_zero_line_split[4] = {0.f, 0.f, 0.f, 0.f};
constexpr unsigned SPLIT_OOB = 0x40000000u;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

template <int NPL, bool BUF, int MINW, bool BPRE = false>
__global__ __launch_bounds__(256, MINW) void gemm_split_kernel(GemmArgs a, int m_tiles, int n_tiles_seg, int gn) {
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int NPROD = NPL == 3 ? 6 : 3;
  // product p multiplies plane PA[p] of A with plane PB[p] of B; the biggest term first so the first MFMA
  // only waits for the first two fragment reads
  constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
  constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
  constexpr int PA2[3] = {0, 0, 1};
  constexpr int PB2[3] = {0, 1, 0};
  __shared__ u32x4s As[2][NPL][BM * 2];  // [stage][plane][row*2 + swizzled chunk], chunk = 8 bf16 of k
  __shared__ u32x4s Bs[2][NPL][BN * 2];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_tiles = n_tiles_seg * a.nseg;  // grouped tile walk: see gemm_f32_kernel
  const int grp = wgid / (gn * m_tiles);
  const int rem = wgid - grp * gn * m_tiles;
  const int gw = (n_tiles - grp * gn < gn) ? n_tiles - grp * gn : gn;
  const int mt = rem / gw;
  const int nt = grp * gn + (rem - mt * gw);
  const int seg = nt / n_tiles_seg;
  const int nts = nt - seg * n_tiles_seg;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nts * BN;
  const int64_t kend = a.K;

  const float* __restrict__ W = a.W[seg];
  const float* __restrict__ bias = a.bias[seg];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging map: thread -> row tid>>1, k half tid&1 (8 consecutive k = two 16-byte loads)
  const int srow = tid >> 1, shalf = tid & 1;
  const int st_chunk = srow * 2 + (shalf ^ ((srow >> 3) & 1));
  unsigned offA = 0, offB = 0;
  const float* pa = nullptr;
  const float* pb = nullptr;
  bool a_ok = true, b_ok = true;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if constexpr (BUF) {
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)(a.M * a.lda * 4), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((int64_t)a.Nseg * a.ldw * 4), 0x00020000);
    const int64_t gr = m0 + srow;
    offA = gr < a.M ? (unsigned)((gr * a.lda + 8 * shalf) * 4) : SPLIT_OOB;
    const int col = n0 + srow;
    offB = col < a.Nseg ? (unsigned)(((int64_t)col * a.ldw + 8 * shalf) * 4) : SPLIT_OOB;
  } else {
    int64_t gr = m0 + srow;
    if (gr >= a.M) { a_ok = false; gr = a.M - 1; }
    int64_t src = gr;
    if (a.gather_ids) {
      const int64_t n = gr / a.gather_S;
      src = (int64_t)a.gather_ids[n] * a.gather_S + (gr - n * a.gather_S);
    }
    pa = a.A + src * a.lda + 8 * shalf;
    int col = n0 + srow;
    if (col >= a.Nseg) { b_ok = false; col = a.Nseg - 1; }
    pb = W + (int64_t)col * a.ldw + 8 * shalf;
  }

  // BPRE: the weights arrive pre-split (GemmArgs::Wp): plane p of this thread's (column, k half)
  // layout (launch_split_weights): planes[p][k/16][n][16] -- the 128 x 16 tile of a k step is 4 KB contiguous, so the
  // plane loads are fully coalesced raw buffer loads: per-plane byte offset in a VGPR, the k step in the scalar offset
  __amdgpu_buffer_rsrc_t rsP;
  unsigned offP[NPL];
  int kstepP = 0;  // bytes per k step = Nseg * 32
  if constexpr (BPRE) {
    int colc = n0 + srow;
    if (colc >= a.Nseg) colc = a.Nseg - 1;  // clamped: the columns beyond Nseg are discarded by the epilogue
    const int64_t kb = a.ldp / 16;
    rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Wp[seg]), 0, (int)(3 * kb * a.Nseg * 32), 0x00020000);
#pragma unroll
    for (int p = 0; p < NPL; ++p) offP[p] = (unsigned)(((p * kb * a.Nseg + colc) * 16 + 8 * shalf) * 2);
    kstepP = a.Nseg * 32;
  }
  u32x4s rbp[NPL];     // BPRE: the bf16 planes of the NEXT stage as loaded
  f32x4 ra[2], rb[2];  // raw fp32 of the NEXT stage: k (8*shalf + 0..3) and (+4..7)
  auto gloadA = [&](int64_t k0) {
    const int64_t k = k0 + 8 * shalf;
    if constexpr (BUF) {
      const int soff = (int)(k0 * 4);
      ra[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(offA | (k < kend ? 0u : SPLIT_OOB)), soff, 0));
      ra[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)((offA + 16u) | (k + 4 < kend ? 0u : SPLIT_OOB)), soff, 0));
    } else {
      ra[0] = *reinterpret_cast<const f32x4*>((a_ok && k < kend) ? pa + k0 : g_zero_line_split);
      ra[1] = *reinterpret_cast<const f32x4*>((a_ok && k + 4 < kend) ? pa + k0 + 4 : g_zero_line_split);
    }
  };
  auto gloadB = [&](int64_t k0) {
    const int64_t k = k0 + 8 * shalf;
    if constexpr (BPRE) {
      (void)k;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        rbp[p] = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)offP[p], (int)(k0 >> 4) * kstepP, 0));
    } else if constexpr (BUF) {
      const int soff = (int)(k0 * 4);
      rb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(offB | (k < kend ? 0u : SPLIT_OOB)), soff, 0));
      rb[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)((offB + 16u) | (k + 4 < kend ? 0u : SPLIT_OOB)), soff, 0));
    } else {
      rb[0] = *reinterpret_cast<const f32x4*>((b_ok && k < kend) ? pb + k0 : g_zero_line_split);
      rb[1] = *reinterpret_cast<const f32x4*>((b_ok && k + 4 < kend) ? pb + k0 + 4 : g_zero_line_split);
    }
  };

  // split state of one operand chunk: packed planes + residuals of the pair being worked on
  u32x4s plA[NPL], plB[NPL];
  float res[2];
  // phase 0 of pair j: hi piece + residual; phase 1: mid (and lo) piece
  auto split_phase = [&](const f32x4 (&x)[2], u32x4s (&pl)[NPL], int j, int phase) {
    if (phase == 0) {
      const float x0 = x[j >> 1][(j & 1) * 2], x1 = x[j >> 1][(j & 1) * 2 + 1];
      const unsigned h = pk_bf16(x0, x1);
      pl[0][j] = h;
      res[0] = x0 - lo_f32(h);
      res[1] = x1 - hi_f32(h);
    } else {
      const unsigned m = pk_bf16(res[0], res[1]);
      pl[1][j] = m;
      if constexpr (NPL == 3) pl[2][j] = pk_bf16(res[0] - lo_f32(m), res[1] - hi_f32(m));
    }
  };
  auto storeA = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) As[buf][p][st_chunk] = plA[p];
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) Bs[buf][p][st_chunk] = BPRE ? rbp[p] : plB[p];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  int fchA[2], fchB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ar = wm * 64 + 32 * i + frow, br = wn * 64 + 32 * i + frow;
    fchA[i] = ar * 2 + (fh ^ ((ar >> 3) & 1));
    fchB[i] = br * 2 + (fh ^ ((br >> 3) & 1));
  }

  const int nk = (int)((kend + BK - 1) / BK);
  const int last = nk - 1;
  auto ktile = [&](int t) { return (int64_t)(t < last ? t : last) * BK; };
  if (nk > 0) {
    gloadA(ktile(0));
    gloadB(ktile(0));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      split_phase(ra, plA, j, 0);
      split_phase(ra, plA, j, 1);
    }
    if constexpr (!BPRE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        split_phase(rb, plB, j, 0);
        split_phase(rb, plB, j, 1);
      }
    }
    storeA(0);
    storeB(0);
    gloadA(ktile(1));
    gloadB(ktile(1));
  }
  __syncthreads();

  // task n of an iteration (run behind MFMA slot n+1): 0..7 split A, 8 store A, 9 load A(t+2), 10..17 split B,
  // 18 store B, 19 load B(t+2)
  constexpr int NSLOT = NPROD * 4;
  constexpr int TPS = (20 + NSLOT - 2) / (NSLOT - 1);  // tasks per slot
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const int64_t kn = ktile(t + 2);
    bf16x8 fa[2][NPL], fb[2][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      fa[0][p] = __builtin_bit_cast(bf16x8, As[buf][p][fchA[0]]);
      fb[0][p] = __builtin_bit_cast(bf16x8, Bs[buf][p][fchB[0]]);
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      fb[1][p] = __builtin_bit_cast(bf16x8, Bs[buf][p][fchB[1]]);
      fa[1][p] = __builtin_bit_cast(bf16x8, As[buf][p][fchA[1]]);
    }
#pragma unroll
    for (int slot = 0; slot < NSLOT; ++slot) {
      const int tile = slot / NPROD, p = slot % NPROD;
      const int ti = (tile == 0 || tile == 1) ? 0 : 1;
      const int tj = (tile == 0 || tile == 3) ? 0 : 1;  // (0,0) (0,1) (1,1) (1,0)
      const int ppa = NPL == 3 ? PA[p] : PA2[p], ppb = NPL == 3 ? PB[p] : PB2[p];
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ti][ppa], fb[tj][ppb], acc[ti][tj], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TPS; ++u) {
        const int n = (slot - 1) * TPS + u;
        if (slot < 1 || n >= 20) continue;
        if (n < 8) split_phase(ra, plA, n >> 1, n & 1);
        else if (n == 8) storeA(buf ^ 1);
        else if (n == 9) gloadA(kn);
        else if (n < 18) { if constexpr (!BPRE) split_phase(rb, plB, (n - 10) >> 1, (n - 10) & 1); }
        else if (n == 18) storeB(buf ^ 1);
        else gloadB(kn);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int ccol = lane & 31;
  const int crow = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + 32 * j + ccol;
    if (col >= a.Nseg) continue;
    const float bv = bias ? bias[col] : 0.f;
    const int64_t coff = (int64_t)seg * a.Nseg + col;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int64_t row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
        if (row < a.M) {
          if (a.c_scatter) {  // row subset in place (see gemm_f32_kernel)
            const int64_t n = row / a.gather_S;
            row = (int64_t)(a.c_scatter_ids ? a.c_scatter_ids : a.gather_ids)[n] * a.gather_S + (row - n * a.gather_S);
          }
          float v = apply_act(acc[i][j][e] + bv, a.act);
          if (a.aux_mode) {
            const float x = a.aux[row * a.ldaux + coff];
            v *= (a.aux_mode == 1) ? (1.f - x * x) : (x > 0.f ? 1.f : 0.f);
          }
          if (a.accumulate) v += a.C[row * a.ldc + coff];
          a.C[row * a.ldc + coff] = v;
        }
      }
    }
  }
}

// planes[p][k/16][n][16] of W[n][k]: the exact 3-way bf16 split (same arithmetic as split_phase), zero padded to ldp columns
__global__ __launch_bounds__(256) void split_weights_kernel(const float* W, int64_t N, int64_t K, int64_t ldp, unsigned short* planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per PAIR of k
  const int64_t pairs = ldp / 2;
  if (i >= N * pairs) return;
  const int64_t n = i / pairs, k = (i - n * pairs) * 2;
  const float x0 = k < K ? W[n * K + k] : 0.f, x1 = k + 1 < K ? W[n * K + k + 1] : 0.f;
  const unsigned h = pk_bf16(x0, x1);
  const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);
  const unsigned m = pk_bf16(r0, r1);
  const unsigned l = pk_bf16(r0 - lo_f32(m), r1 - hi_f32(m));
  unsigned* out = reinterpret_cast<unsigned*>(planes);
  const int64_t plane = N * pairs;
  const int64_t o = ((k >> 4) * N + n) * 8 + ((k & 15) >> 1);  // [k/16][n][16] in pairs
  out[o] = h;
  out[plane + o] = m;
  out[2 * plane + o] = l;
}

hipError_t launch_split_weights(const float* W, int64_t N, int64_t K, unsigned short* planes, hipStream_t stream) {
  const int64_t ldp = split_plane_ld(K);
  const int64_t n = N * (ldp / 2);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, W, N, K, ldp, planes);
  return hipGetLastError();
}

// forward-layout launcher; the caller (launch_gemm_f32) has checked: !a_col, !b_kn, no split-K, K % 4 == 0,
// 16-byte aligned operands.  npl = 3 (six products, fp32-grade) or 2 (three products).
hipError_t launch_gemm_split(const GemmArgs& a, int npl, hipStream_t stream) {
  const int64_t m_tiles = (a.M + 127) / 128;
  const int n_tiles_seg = (a.Nseg + 127) / 128;
  const int64_t grid = m_tiles * n_tiles_seg * a.nseg;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 g((unsigned)grid, 1);
  const int gn = gemm_group_tiles(n_tiles_seg * a.nseg, 128, a.K, false);
  const bool buf = knobs().gemm_buf && !a.gather_ids && a.M * a.lda * 4 <= (int64_t)SPLIT_OOB &&
                   (int64_t)a.Nseg * a.ldw * 4 <= (int64_t)SPLIT_OOB;
#define XNRS_LAUNCH_SPLIT(NPLV, BUFV, MINWV) \
  hipLaunchKernelGGL((gemm_split_kernel<NPLV, BUFV, MINWV>), g, dim3(256), 0, stream, a, (int)m_tiles, n_tiles_seg, gn)
  bool pre = a.Wp[0] != nullptr;
  for (int s2 = 1; s2 < a.nseg; ++s2) pre = pre && a.Wp[s2] != nullptr;
  if (pre && buf) {
    if (npl == 3) hipLaunchKernelGGL((gemm_split_kernel<3, true, 3, true>), g, dim3(256), 0, stream, a, (int)m_tiles, n_tiles_seg, gn);
    else hipLaunchKernelGGL((gemm_split_kernel<2, true, 3, true>), g, dim3(256), 0, stream, a, (int)m_tiles, n_tiles_seg, gn);
  } else if (pre) {
    if (npl == 3) hipLaunchKernelGGL((gemm_split_kernel<3, false, 3, true>), g, dim3(256), 0, stream, a, (int)m_tiles, n_tiles_seg, gn);
    else hipLaunchKernelGGL((gemm_split_kernel<2, false, 3, true>), g, dim3(256), 0, stream, a, (int)m_tiles, n_tiles_seg, gn);
  } else if (npl == 3) {
    if (buf) XNRS_LAUNCH_SPLIT(3, true, 3);
    else XNRS_LAUNCH_SPLIT(3, false, 3);
  } else {
    if (buf) XNRS_LAUNCH_SPLIT(2, true, 3);
    else XNRS_LAUNCH_SPLIT(2, false, 3);
  }
#undef XNRS_LAUNCH_SPLIT
  return hipGetLastError();
}

}  // namespace xnrs
