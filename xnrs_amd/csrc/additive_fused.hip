// The additive-attention encoder as ONE launch (the K1' of SURVEY.md section 2.3: a TextEncoder without self-attention,
// i.e. the news towers of StandardRec / BaseRec / NAML / LSTUR -- BASELINE configs[3] and [4]):
//
//   e_i = w2 . tanh(W1 x_i + b1) + b2 ;  a_i = exp(e_i) m_i / (sum_j exp(e_j) m_j + 1e-8) ;  p = sum_i a_i x_i
//   (layers.AdditiveAttention.forward, xnrs/models/components/layers.py:60-65; un-stabilised exp and 1e-8 as there)
//
// instead of the pipeline's fc1 GEMM (with the fc2 dot in its epilogue) + additive_pool_kernel, which (i) ran the GEMM as
// exactly ONE round of co-resident workgroups per 65 500-row pass, so every workgroup reached its tanh epilogue together
// and the matrix pipe idled through it, and (ii) re-read every token row from HBM in a second launch (6 TB/s for 33 us per
// pass, 12 % of the StandardRec step).  Here:
//
//   * PERSISTENT workgroups, one per CU (512 threads = 8 waves = 2 per SIMD, <= 256 registers each), each walking tiles of
//     WHOLE NEWS: nn = floor(256 / S) news = nn*S <= 256 token rows (S = 50: 5 news, 250 rows, 2.3 % padding) against all
//     A <= 256 hidden units -- a 256 x 256 block tile, wave tile 128 x 64 (4 x 2 MFMA blocks of v_mfma_f32_32x32x2_f32:
//     6 fragment reads per 32 MFMAs instead of the GEMM's 4 per 16).  A news never straddles two workgroups, so its
//     vector does not depend on where in the batch it sits.
//   * The K loop is the GEMM's (gemm_f32.hip PIPE 5): K tiles of 16, two LDS buffers, XOR-swizzled 64-byte rows, one
//     register set, tile loads / LDS stores / fragment reads dealt out one per slot of 8 MFMAs behind sched_barriers.
//   * Scores in the epilogue: tanh, x w2, butterfly over the 32 columns of an MFMA block, the A/32 block sums added in
//     column order, exp, mask; one wave per news takes the normaliser.  The normalised weights stay in LDS.
//   * THE WEIGHTED SUM OF TILE t RIDES INSIDE THE K LOOP OF TILE t+1: a thread owns (news, 16-byte column chunk) pairs and
//     per K iteration loads the next two value rows of each pair (coalesced 3-KB rows, L2 / Infinity-Cache hits: the
//     workgroup streamed the same rows a tile ago) and folds the previous two in -- a handful of loads and FMAs next to
//     64 MFMAs.  The pooling costs no time of its own and the token rows come from HBM once.  (The last tile of a
//     workgroup is pooled in a short loop of its own.)
//
// Bit-exactness: every per-row dot product has the GEMM's k order (same fragments, same MFMA), the score reduction has
// the RDOT epilogue's block / butterfly order, the normaliser the pooling kernel's wave_sum order and the weighted sum
// its row order -- the result equals the pipeline's BIT FOR BIT (tests/test_hip_additive_fused.py), so the dispatcher
// may pick either by batch size without a news vector ever changing.
#include <mutex>
#include <type_traits>

#include "kernels.h"

namespace xnrs {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned AF_OOB = 0x40000000u;
constexpr int AF_BM = 256, AF_BN = 256, AF_BK = 16;
// 8 waves (2 per SIMD, <= 256 registers: the K loop then holds no spill) of 128 x 64 each.  16 waves of 64 x 64 at <= 128
// registers (-DAF_WAVES=16) spill 9-12 registers per K iteration next to the pooling state and are not built.
#ifndef AF_WAVES
#define AF_WAVES 8
#endif
constexpr int AF_T = 64 * AF_WAVES;
// diagnostic builds only (make afexp; tools/bench_af.py): bit 0 no pooling ticks, bit 1 no score epilogue, bit 2 no
// barrier in the K loop, bit 3 no tile staging in the K loop, bit 4 no fragment reads -- wrong results, timing only; never shipped
#ifndef AF_EXP
#define AF_EXP 0
#endif
constexpr int AF_TM = AF_WAVES == 16 ? 2 : 4, AF_TN = 2;
constexpr int AF_RPP = AF_T / 4;             // tile rows staged per pass (4 chunks of 16 bytes per row)
constexpr int AF_NR = AF_BM / AF_RPP;        // rows per thread per operand tile
constexpr int AF_NCH = 2 * AF_NR;            // 16-byte chunks per thread per K tile (A rows, then B rows)
constexpr int AF_SPI = 2;             // pooling steps (value rows per pair) per K iteration
constexpr int AF_EPS = 9;             // s_ep row stride (8 block sums + 1 pad)

__device__ __forceinline__ float af_wave_sum(float v) {  // = pool_score.hip wave_sum (same order)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// PP = (news, column chunk) pairs per thread in the pooling phase: ceil(nn * D/4 / 512); AF_FBUF = fragment register sets
template <int PP, int AF_FBUF>
__global__ __launch_bounds__(AF_T, AF_WAVES / 4) void additive_fused_kernel(AdditiveFusedArgs a, int nn, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float As[2][AF_BM * AF_BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][AF_BN * AF_BK];
  __shared__ float s_ep[AF_BM * AF_EPS];
  __shared__ float s_w[AF_BM];
  __shared__ float s_wn[2][AF_BM];
  __shared__ float2 s_bw[AF_BN];  // {b1[h], w2[h]} of every hidden unit, zeros past A

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // AF_WAVES / 4 x 4 waves
  const int S = a.S, D = a.D, A = a.A;
  const int64_t SD = (int64_t)S * D;
  const int lc = tid & 3, lr = tid >> 2;  // staging: 16-byte chunk lc of tile rows lr (+ AF_RPP)
  const int nk = (D + AF_BK - 1) / AF_BK;
  const int last = nk - 1;
  const int D4 = D >> 2;
  const int npairs = nn * D4;
  const int n_ep = (A + 31) >> 5;
  const float b2 = a.b2 ? a.b2[0] : 0.f;

  if (tid < AF_BN) s_bw[tid] = make_float2((a.b1 && tid < A) ? a.b1[tid] : 0.f, tid < A ? a.w2[tid] : 0.f);
  // (published by the first K loop's barriers long before the first epilogue reads it)

  // ---- B operand (W1 [A][D], L2-resident): raw buffer loads, rows >= A and the k tail read zeros via the bounds check
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, (int)((int64_t)A * D * 4), 0x00020000);
  const unsigned offB0 = lr < A ? (unsigned)((lr * D + 4 * lc) * 4) : AF_OOB;
  const int dB = AF_RPP * D * 4;

  auto kswz = [](int row, int c) { return c ^ ((row >> 2) & 3); };
  const int frow = lane & 31;
  const int fkc = lane >> 5;  // which 16-byte chunk of an 8-wide k group this lane feeds
  const int a_row0 = wm * 32 * AF_TM + frow;
  const int b_row0 = wn * 32 * AF_TN + frow;

  // pooling state of the PREVIOUS tile (folded into this tile's K loop)
  bool have_prev = false;
  int64_t prev_news0 = 0;
  int prev_buf = 0;
  const float* pb[PP];  // value rows of pair p: pb[p] + step * D
  int pwo[PP];          // index of its news' first weight in s_wn
  f32x4 pacc[PP];
  f32x4 px[PP][AF_SPI];
  int p_issued = 0, p_cons = 0;
#pragma unroll
  for (int p = 0; p < PP; ++p) {  // before the first tile: any valid row (the sums are discarded)
    pb[p] = a.x;
    pwo[p] = 0;
    pacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < AF_SPI; ++s) px[p][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto pool_setup = [&](int64_t news0) {
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int q = tid + AF_T * p;
      const int jn = q < npairs ? q / D4 : 0;
      const int c = q < npairs ? q - jn * D4 : 0;
      int64_t news = news0 + jn;
      if (news >= a.n_seq) news = a.n_seq - 1;
      const int64_t src = a.ids ? (int64_t)a.ids[news] : news;
      pb[p] = a.x + src * SD + 4 * c;
      pwo[p] = jn * S;
      pacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    p_issued = 0;
    p_cons = 0;
  };
  // One tick per K iteration, BRANCH-FREE: fold in the AF_SPI rows loaded by the previous tick (in row order), then load
  // the next AF_SPI.  Steps past S keep loading row S-1 (an L1 hit) and fold it in with weight 0 (x + 0 * finite = x; a
  // non-finite token poisons the pipeline's sum just the same), and the first tile of a workgroup -- nothing to pool yet --
  // runs the same instructions on its own first row and throws the sums away.  With conditional loads the compiler had
  // to assume at every LDS store of the K loop that the pooling loads might NOT have been issued (s_waitcnt vmcnt(3)
  // instead of 7), i.e. it drained them a few hundred cycles after their issue: 0.65 of the matrix peak.
  auto pool_consume = [&]() {
#pragma unroll
    for (int s = 0; s < AF_SPI; ++s) {
      const int st = p_cons + s;
      const int stc = st < S ? st : S - 1;
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        float w = s_wn[prev_buf][pwo[p] + stc];
        w = st < S ? w : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) pacc[p][r] = fmaf(w, px[p][s][r], pacc[p][r]);
      }
    }
    p_cons += AF_SPI;
  };
  auto pool_issue = [&]() {
#pragma unroll
    for (int s = 0; s < AF_SPI; ++s) {
      const int st = (p_issued + s < S) ? p_issued + s : S - 1;
#pragma unroll
      for (int p = 0; p < PP; ++p) px[p][s] = *reinterpret_cast<const f32x4*>(pb[p] + (int64_t)st * D);
    }
    p_issued += AF_SPI;
  };
  auto pool_store = [&](int64_t news0) {
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const int q = tid + AF_T * p;
      if (q < npairs) {
        const int jn = q / D4;
        const int c = q - jn * D4;
        const int64_t news = news0 + jn;
        if (news < a.n_seq) *reinterpret_cast<f32x4*>(a.y + news * a.ldy + 4 * c) = pacc[p];
      }
    }
  };

  // ---- A operand rows of a tile: tile row r = news r / S, token r % S (rows past nn*S and news past the batch are clamped
  // to a valid row: their results are never used)
  const float* pa[AF_NR];
  f32x4 ra[AF_NR], rb[AF_NR];
  auto tile_rows = [&](int64_t news0) {
#pragma unroll
    for (int i = 0; i < AF_NR; ++i) {
      const int r = lr + AF_RPP * i;
      int jn = r / S;
      int tok = r - jn * S;
      if (jn >= nn) {
        jn = nn - 1;
        tok = 0;
      }
      int64_t news = news0 + jn;
      if (news >= a.n_seq) news = a.n_seq - 1;
      const int64_t src = a.ids ? (int64_t)a.ids[news] : news;
      pa[i] = a.x + src * SD + (int64_t)tok * D + 4 * lc;
    }
  };
  auto gload0 = [&]() {  // K tile 0 of the tile pa[] points at
    const int klim = D - 4 - 4 * lc;
    const int ka = 0 < klim ? 0 : klim;
    const unsigned sel = (4 * lc < D) ? 0u : AF_OOB;
#pragma unroll
    for (int i = 0; i < AF_NR; ++i) {
      ra[i] = *reinterpret_cast<const f32x4*>(pa[i] + ka);
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(offB0 | sel), i * dB, 0));
    }
  };
  if ((int)blockIdx.x < n_tiles) {
    tile_rows((int64_t)blockIdx.x * nn);
    gload0();
  }
  int it = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
    const int64_t news0 = (int64_t)tile * nn;
    const int cur = it & 1;  // s_wn buffer of this tile (the other one still feeds the previous tile's pooling)
    // (pa[]: the A operand rows of this tile, set up -- with its first K tile already in flight -- in front of the
    // previous tile's epilogue, or before the loop for the first tile)
    auto gload = [&](int k0, int which) {  // which < AF_NR: A row lr + AF_RPP * which; then the B rows
      if (which < AF_NR) {
        const int klim = D - 4 - 4 * lc;  // k tail: clamped (it meets the zeros B returns there)
        const int ka = k0 < klim ? k0 : klim;
        ra[which] = *reinterpret_cast<const f32x4*>(pa[which] + ka);
      } else {
        const unsigned sel = (k0 + 4 * lc < D) ? 0u : AF_OOB;
        rb[which - AF_NR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(offB0 | sel), k0 * 4 + (which - AF_NR) * dB, 0));
      }
    };
    auto sstore = [&](int buf, int which) {
      const int row = lr + AF_RPP * (which < AF_NR ? which : which - AF_NR);
      float* dst = (which < AF_NR ? As[buf] : Bs[buf]) + row * AF_BK + 4 * kswz(row, lc);
      *reinterpret_cast<f32x4*>(dst) = which < AF_NR ? ra[which] : rb[which - AF_NR];
    };

    f32x16 acc[AF_TM][AF_TN];
#pragma unroll
    for (int i = 0; i < AF_TM; ++i)
#pragma unroll
      for (int j = 0; j < AF_TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 fa[AF_FBUF][AF_TM], fb[AF_FBUF][AF_TN];
    auto ldfrag = [&](int st, int buf, int kq) {
#pragma unroll
      for (int i = 0; i < AF_TM; ++i) {
        const int row = a_row0 + 32 * i;
        fa[st][i] = *reinterpret_cast<const f32x4*>(&As[buf][row * AF_BK + 4 * kswz(row, kq * 2 + fkc)]);
      }
#pragma unroll
      for (int j = 0; j < AF_TN; ++j) {
        const int row = b_row0 + 32 * j;
        fb[st][j] = *reinterpret_cast<const f32x4*>(&Bs[buf][row * AF_BK + 4 * kswz(row, kq * 2 + fkc)]);
      }
    };

    // ---- K loop (gemm_f32.hip PIPE 5) with the previous tile's pooling folded in
    // (K tile 0 is in ra / rb already: loaded in front of the previous tile's epilogue)
#pragma unroll
    for (int w = 0; w < AF_NCH; ++w) sstore(0, w);
#pragma unroll
    for (int w = 0; w < AF_NCH; ++w) gload(AF_BK * (1 < last ? 1 : last), w);
    // the previous tile's rows 0 .. AF_SPI-1; every tick of the K loop folds in what the tick before loaded, then loads.
    // Issued HERE, behind the tile loads, so that the loop is entered with the loads outstanding in the same order as
    // around its back edge (tile loads, then pooling loads): the compiler's s_waitcnt vmcnt(N) in front of the LDS stores
    // is the minimum over both ways in, and a pooling load issued before the prologue made it drain them every iteration
    pool_issue();
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const int buf = t & 1;
      const int kn = AF_BK * (t + 2 < last ? t + 2 : last);
      if (!(AF_EXP & 16) || t == 0) ldfrag(0, buf, 0);
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int slot = kq * 4 + e;
#pragma unroll
          for (int i = 0; i < AF_TM; ++i)
#pragma unroll
            for (int j = 0; j < AF_TN; ++j)
              // W1 . X^T: transposed blocks (tokens on the lanes, hidden units down the accumulator), see rowdot_block_t
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kq % AF_FBUF][j][e], fa[kq % AF_FBUF][i][e], acc[i][j], 0, 0, 0);
          if ((AF_FBUF == 2 ? slot == 1 : slot == 3) && !(AF_EXP & 16)) ldfrag(1 % AF_FBUF, buf, 1);  // one register set: behind the group's last MFMAs
          if (!(AF_EXP & 8)) {
            if (slot >= 1 && slot <= AF_NCH) sstore(buf ^ 1, slot - 1);      // tile t+1 -> LDS (a redundant re-store at the tail)
            if (slot >= 2 && slot <= AF_NCH + 1) gload(kn, slot - 2);        // tile t+2 -> the register just stored
          }
          if (slot == 6 && !(AF_EXP & 1)) {
            pool_consume();
            pool_issue();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(AF_EXP & 4)) __syncthreads();
    }
    if (have_prev) {  // rows the K loop had no iterations for (S > 2 (nk - 1): short contractions), then the result
      while (p_cons < S) {
        pool_consume();
        pool_issue();
      }
      pool_store(prev_news0);
    }

    // K tile 0 of this workgroup's NEXT tile flies during the epilogue (2-3 us of HBM latency per tile otherwise)
    if (tile + (int)gridDim.x < n_tiles) {
      tile_rows((int64_t)(tile + gridDim.x) * nn);
      gload0();
    }

    // ---- scores: per transposed block an in-lane fmaf chain over 16 hidden units + one exchange with lane l ^ 32
    // (rowdot_block_t, shared with the GEMM's RDOT epilogue); lanes 0..31 park the block sum of their token row
    const int half = lane >> 5;
    auto scores = [&](auto FAST) {
#pragma unroll
      for (int j = 0; j < AF_TN; ++j) {
        const int slot = wn * AF_TN + j;
        float2 bw[16];
        rowdot_load_bw(bw, s_bw + 32 * slot, half);
        float sc[AF_TM];
#pragma unroll
        for (int i = 0; i < AF_TM; ++i) sc[i] = rowdot_block_t<decltype(FAST)::value>(acc[i][j], bw);
#pragma unroll
        for (int i = 0; i < AF_TM; ++i)
          if (lane < 32) s_ep[(wm * 32 * AF_TM + 32 * i + lane) * AF_EPS + slot] = sc[i];
      }
    };
    if (AF_EXP & 2) {  // every accumulator element used (no dead MFMAs), none of the score arithmetic
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < AF_TM; ++i)
#pragma unroll
        for (int j = 0; j < AF_TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) t += acc[i][j][e];
      if (lane < 32) s_ep[(wm * 32 * AF_TM + lane) * AF_EPS + wn] = t;
    } else if (a.tanh_act == ACT_TANH_FAST) scores(std::true_type{});
    else scores(std::false_type{});
    __syncthreads();
    // exp(score) * mask per tile row (the pooling kernel's thread-per-row arithmetic)
    if (tid < AF_BM) {
      const int jn = tid / S;
      const int tok = tid - jn * S;
      const int64_t news = news0 + jn;
      float ev = 0.f;
      if (jn < nn && news < a.n_seq) {
        float sc = 0.f;
        for (int s = 0; s < n_ep; ++s) sc += s_ep[tid * AF_EPS + s];
        ev = expf(sc + b2);
        if (a.mask) {
          const int64_t msrc = a.ids ? (int64_t)a.ids[news] : news;
          ev *= a.mask[msrc * S + tok];
        }
      }
      s_w[tid] = ev;
    }
    __syncthreads();
    // normaliser and news mask: one wave per news (S <= 64; the pooling kernel's wave_sum order, its three idle waves add 0)
    for (int jn = wave; jn < nn; jn += 8) {
      const int64_t news = news0 + jn;
      if (news >= a.n_seq) break;  // wave-uniform
      const float ev = lane < S ? s_w[jn * S + lane] : 0.f;
      const float sumw = af_wave_sum(ev);
      const float denom = sumw + 1e-8f;
      if (lane < S) s_wn[cur][jn * S + lane] = ev / denom;
      if (a.hm) {
        const int64_t msrc = a.ids ? (int64_t)a.ids[news] : news;
        const float mv = (a.mask && lane < S) ? a.mask[msrc * S + lane] : 0.f;
        const float msum = af_wave_sum(mv);
        if (lane == 0) a.hm[news] = fminf(fmaxf(msum, 0.f), 1.f);
      }
    }
    // (s_wn[cur] is read by the pooling ticks of the NEXT tile's K loop, behind that loop's first barrier)
    have_prev = true;
    prev_news0 = news0;
    prev_buf = cur;
    pool_setup(news0);
  }
  // ---- the last tile of this workgroup: pooled on its own (8 rows in flight per pair)
  if (have_prev) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PP; ++p) pacc[p] = f32x4{0.f, 0.f, 0.f, 0.f};  // from row 0 again (the pre-issued rows are dropped)
    constexpr int TB = PP <= 2 ? 8 : 4;  // rows in flight per pair
    for (int st0 = 0; st0 < S; st0 += TB) {
      f32x4 v[PP][TB];
#pragma unroll
      for (int s = 0; s < TB; ++s) {
        const int st = st0 + s < S ? st0 + s : S - 1;
#pragma unroll
        for (int p = 0; p < PP; ++p) v[p][s] = *reinterpret_cast<const f32x4*>(pb[p] + (int64_t)st * D);
      }
#pragma unroll
      for (int s = 0; s < TB; ++s) {
        if (st0 + s < S) {
#pragma unroll
          for (int p = 0; p < PP; ++p) {
            const float w = s_wn[prev_buf][pwo[p] + st0 + s];
#pragma unroll
            for (int r = 0; r < 4; ++r) pacc[p][r] = fmaf(w, v[p][s][r], pacc[p][r]);
          }
        }
      }
    }
    pool_store(prev_news0);
  }
}

}  // namespace

// shapes the kernel serves: whole news of <= 64 tokens (one wave takes a normaliser), at least 4 of them per 256-row tile
// worth of work per pair list, 128 < A <= 256 hidden units (fewer would idle half the tile), 16-byte chunks
bool additive_fused_plan(int S, int D, int A, int* nn_out, int* pp_out) {
  if (S < 4 || S > 64 || D < 16 || D % 4 != 0 || A <= 128 || A > 256) return false;
  if ((int64_t)A * D * 4 > (int64_t)AF_OOB) return false;
  const int nn = AF_BM / S;
  const int pairs = nn * (D / 4);
  const int pp = (pairs + AF_T - 1) / AF_T;
  if (pp < 1 || pp > 4) return false;
  if (nn_out) *nn_out = nn;
  if (pp_out) *pp_out = pp;
  return true;
}

bool additive_fused_ready(const AdditiveFusedArgs& a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return additive_fused_plan(a.S, a.D, a.A, nullptr, nullptr) && a.x && a.w1 && a.w2 && a.y && al16(a.x) && al16(a.w1) &&
         al16(a.y) && a.ldy % 4 == 0;
}

namespace {
int cu_count() {  // per device, cached
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}
}  // namespace

int64_t additive_fused_tiles(int64_t n_seq, int S) { return S > 0 ? (n_seq + AF_BM / S - 1) / (AF_BM / S) : 0; }

hipError_t launch_additive_fused(const AdditiveFusedArgs& a_in, hipStream_t stream) {
  AdditiveFusedArgs a = a_in;
  if (a.n_seq <= 0) return hipSuccess;
  int nn = 0, pp = 0;
  if (!additive_fused_ready(a) || !additive_fused_plan(a.S, a.D, a.A, &nn, &pp)) return hipErrorInvalidValue;
  a.tanh_act = knobs().fast_tanh ? ACT_TANH_FAST : 2;
  const int64_t n_tiles = (a.n_seq + nn - 1) / nn;
  if (n_tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  const int cus = cu_count();
  const int grid = (int)(n_tiles < cus ? n_tiles : cus);
  const bool fb2 = knobs().af_fbuf == 2;
#define AF_LAUNCH(P)                                                                                                       \
  if (fb2) hipLaunchKernelGGL((additive_fused_kernel<P, 2>), dim3((unsigned)grid), dim3(AF_T), 0, stream, a, nn, (int)n_tiles); \
  else hipLaunchKernelGGL((additive_fused_kernel<P, 1>), dim3((unsigned)grid), dim3(AF_T), 0, stream, a, nn, (int)n_tiles)
  switch (pp) {
    case 1: AF_LAUNCH(1); break;
    case 2: AF_LAUNCH(2); break;
    case 3: AF_LAUNCH(3); break;
    default: AF_LAUNCH(4); break;
  }
#undef AF_LAUNCH
  return hipGetLastError();
}

}  // namespace xnrs
