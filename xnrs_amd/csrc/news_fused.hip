// Fused news encoder for short titles (S <= 32 tokens): ONE launch (plus a weight-reordering prologue) for
//   TextEncoder.forward   xnrs/models/components/news_encoding.py:48-54   (att -> pooler; the MLP head stays a GEMM pair)
//   MultiHeadAttention    xnrs/models/components/layers.py:128-154        (Q/K/V projection, row-masked softmax, PV, out)
//   AdditiveAttention     xnrs/models/components/layers.py:60-65          (fc1, tanh, fc2, exp, mask, normalise, sum)
// BASELINE configs[1] (1024 news x 30 tokens, D = 300 / 320): the six-launch pipeline of api.hip runs at 0.50-0.54
// of the fp32 matrix peak there because every stage is a few dozen microseconds of partially filled tiles.
//
// Decomposition.  A workgroup of 8 waves owns one CU and NPW = 2 news = 4 tiles of 16 token rows, and keeps them on
// the CU from the token rows to the pooled vector.  Everything is computed TRANSPOSED -- out^T[feature][row] = W .
// act^T -- with v_mfma_f32_16x16x4_f32: the weight rows are the MFMA's A operand, the token rows its B operand, so
//   * both operands are 16-byte fragments of 4 consecutive k (lane (c, g): row c, k = 4g .. 4g+3; MFMA step j uses
//     k = 4g + j on both sides);
//   * an accumulator holds 4 consecutive FEATURES of one token row per lane -> one ds_write_b128 into a row-major
//     [row][feature] LDS image, which is exactly what the next product reads back as its B fragments;
//   * the waves split the OUTPUT FEATURES of each product in 16-feature tiles (Q/K/V projection and fc1: 8 ways x all
//     4 row tiles; out-projection: 4 ways x the 2 row tiles of one news, so that its 20 tiles at D = 320 divide evenly).
//
// Weights (4 D^2 + A D floats, 2 MB at D = 320) are the same for every workgroup and stay L2-resident.  A prologue
// kernel (news_fused_prep_kernel, ~2 MB written, a few microseconds per call -- the ABI keeps no state between calls)
// rewrites them into MFMA FRAGMENT ORDER, zero padded: image[k step][wave][tile][lane][4] holds exactly the 16 bytes
// lane `lane` feeds to the four MFMAs of that (k step, tile), so a weight fragment load is one fully coalesced 1-KB
// read at a wave-uniform address + 16 * lane: no per-lane row pointers, no k-tail / feature-tail selects, no address
// arithmetic in the k loops.  (The first version -- 4 waves, 16 rows x 64 B per load through 64-bit row pointers,
// token rows re-read from global by every wave and head group -- issued 2.5 VALU instructions per MFMA, pulled
// 2.5 GB through L2 per 1024 news and left the matrix pipe 50 % idle: profiles/r02_news_fused_v1_pmc.txt.)
// The fragments of k step ks+1 are loaded before the MFMAs of step ks (two register sets): MFMA arbitration between
// the waves of a SIMD is fair, so they finish a burst together and, without the prefetch, wait for memory together.
//
// LDS (<= 160 KB, one workgroup per CU):
//   R1 [rows][LY]  the token rows X (read from HBM once, coalesced; optionally gathered by news id), later Y
//   R2 [rows][LQ]  the Q|K|V columns of ONE GROUP of heads (4 heads of 20 at D = 320)
// Per group: project (B fragments from R1) -> per (news, head) softmax(rowmask(K Q^T)) and V^T P^T with P in registers
// (same arithmetic as mha_core.hip), O overwrites Q in place -> the out-projection accumulates the group's columns of
// Wo into Y^T accumulators that live in registers across the groups (40 VGPRs).  Then Y + bo -> R1, fc1 + tanh with
// the fc2 dot product taken straight from the accumulators (T never exists in memory), exp * mask / (sum + 1e-8) and
// the weighted sum of the Y rows.
#include <mutex>
#include <type_traits>

#include "kernels.h"

namespace xnrs {

namespace {

// news per workgroup: a template parameter.  2: one workgroup per CU, every weight fragment feeds 4 row tiles;
// 1: two workgroups per CU (<= 128 VGPRs, ~70 KB LDS) whose barrier / softmax / epilogue phases overlap each other's
// MFMA loops, at twice the weight traffic per news.
constexpr int NF_TF = 2;      // Q/K/V projection: 16-feature tiles per wave, 8 waves  (<= 256 columns per head group)
constexpr int NF_TY = 5;      // out-projection: tiles per wave, 4 feature waves x 2 row waves  (D <= 320)
constexpr int NF_TA = 2;      // fc1: tiles per wave, 8 waves  (A <= 256)
constexpr int NF_FRAG = 256;  // floats of one (k step, tile) weight fragment: 64 lanes x 4
constexpr int NF_THREADS = 512;
// Column swizzle of both LDS row images: element (row, col) lives at row * L + (col ^ NF_SWZ(row)), bit 2 of the column
// flipped on rows whose bit 2 is set.  Row strides are 8 mod 16 floats, which makes the row-fragment ds_read_b128s
// conflict-free under the hardware's lane groups but leaves the accumulator ds_write_b128s (8 contiguous lanes = 8 token
// rows at one feature chunk, banks mod 32) on 4 of their 8 slots -- no plain stride serves both (enumerated); with the
// flip rows r and r + 4 of a store group land 16 bytes apart and the reads stay clean (brute-forced against the lane
// groups of MI355X_MICROARCH.md, LDS).  The d_k % 16 tail columns go from 4-way to 2-way; the scalar gathers of V^T keep
// their 2-way conflict (rows 4 apart are 32 banks apart whatever bit 2 does).  -DNF_SWZ_OFF: the plain layout.
#ifdef NF_SWZ_OFF
#define NF_SWZ(row) 0
#else
#define NF_SWZ(row) ((((row) >> 2) & 1) << 2)
#endif

template <int N>
using IC = std::integral_constant<int, N>;

// offsets (floats) of the fragment-ordered images inside the prepared-weight workspace
struct NfImg {
  int n_groups, nk, nkc;
  __host__ __device__ size_t qkv_stride() const { return (size_t)nk * 8 * NF_TF * NF_FRAG; }
  __host__ __device__ size_t wo_stride() const { return (size_t)nkc * 4 * NF_TY * NF_FRAG; }
  __host__ __device__ size_t off_qkv(int g) const { return (size_t)g * qkv_stride(); }
  __host__ __device__ size_t off_wo(int g) const { return (size_t)n_groups * qkv_stride() + (size_t)g * wo_stride(); }
  __host__ __device__ size_t off_w1() const { return off_wo(n_groups); }
  __host__ __device__ size_t off_bqkv(int g) const { return off_w1() + (size_t)nk * 8 * NF_TA * NF_FRAG + (size_t)g * 8 * NF_TF * 16; }
  __host__ __device__ size_t off_bo() const { return off_bqkv(n_groups); }
  __host__ __device__ size_t off_b1() const { return off_bo() + 4 * NF_TY * 16; }
  __host__ __device__ size_t off_w2() const { return off_b1() + 8 * NF_TA * 16; }
  __host__ __device__ size_t total() const { return off_w2() + 8 * NF_TA * 16; }
};

// image[ks][wave][t][lane][j] = W[16 (wave + NWAVE t) + (lane & 15)][16 ks + 4 (lane >> 4) + j], zero outside W
// (NWAVE = 8 for the Q|K|V and fc1 images, 4 for the out-projection image)
__global__ __launch_bounds__(256) void news_fused_prep_kernel(NewsFusedArgs a, NfImg im, int HG, float* img) {
  const int D = a.D, dk = a.d_k, H = a.n_heads, A = a.A;
  const size_t n4 = im.off_bqkv(0) / 4;  // float4 slots of the three weight images
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 < n4) {
    const size_t e = i4 * 4;
    const int lane = (int)(i4 & 63), c = lane & 15, g4 = lane >> 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (e < im.off_wo(0)) {  // Q|K|V columns of a head group
      const int grp = (int)(e / im.qkv_stride());
      size_t r = (e - (size_t)grp * im.qkv_stride()) / NF_FRAG;  // (ks * 8 + wave) * TF + t
      const int t = (int)(r % NF_TF);
      r /= NF_TF;
      const int w = (int)(r & 7), ks = (int)(r >> 3);
      const int h0 = grp * HG, nh = (H - h0 < HG) ? H - h0 : HG, NW = nh * dk;
      const int f = 16 * (w + 8 * t) + c, k = 16 * ks + 4 * g4;
      if (f < 3 * NW && k < D) {
        const int seg = f / NW, wi = f - seg * NW;
        const float* Ws = seg == 0 ? a.wq : (seg == 1 ? a.wk : a.wv);
        v = *reinterpret_cast<const f32x4*>(Ws + (size_t)(h0 * dk + wi) * D + k);
      }
    } else if (e < im.off_w1()) {  // this group's d_k * nh columns of Wo
      const size_t e2 = e - im.off_wo(0);
      const int grp = (int)(e2 / im.wo_stride());
      size_t r = (e2 - (size_t)grp * im.wo_stride()) / NF_FRAG;  // (ks * 4 + wf) * TY + t
      const int t = (int)(r % NF_TY);
      r /= NF_TY;
      const int w = (int)(r & 3), ks = (int)(r >> 2);
      const int h0 = grp * HG, nh = (H - h0 < HG) ? H - h0 : HG, NW = nh * dk;
      const int d = 16 * (w + 4 * t) + c, k = 16 * ks + 4 * g4;
      if (d < D && k < NW) v = *reinterpret_cast<const f32x4*>(a.wo + (size_t)d * D + h0 * dk + k);
    } else {  // fc1
      size_t r = (e - im.off_w1()) / NF_FRAG;  // (ks * 8 + wave) * TA + t
      const int t = (int)(r % NF_TA);
      r /= NF_TA;
      const int w = (int)(r & 7), ks = (int)(r >> 3);
      const int ar = 16 * (w + 8 * t) + c, k = 16 * ks + 4 * g4;
      if (ar < A && k < D) v = *reinterpret_cast<const f32x4*>(a.w1 + (size_t)ar * D + k);
    }
    *reinterpret_cast<f32x4*>(img + e) = v;
    return;
  }
  // bias / fc2 images: [wave][t][16], element i = vector[16 (wave + NWAVE t) + i] or 0
  const size_t e = (i4 - n4) * 4 + im.off_bqkv(0);
  if (e >= im.total()) return;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  auto pick = [&](size_t rel, int T, int nwave, const float* src, int n) {
    const int i0 = (int)(rel & 15);
    const int t = (int)((rel >> 4) % T), w = (int)((rel >> 4) / T);
    const int f = 16 * (w + nwave * t) + i0;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (src && f + r < n) ? src[f + r] : 0.f;
  };
  if (e < im.off_bo()) {
    const size_t rel0 = e - im.off_bqkv(0);
    const int grp = (int)(rel0 / (8 * NF_TF * 16));
    const int h0 = grp * HG, nh = (H - h0 < HG) ? H - h0 : HG, NW = nh * dk;
    const size_t rel = rel0 - (size_t)grp * 8 * NF_TF * 16;
    const int f = 16 * ((int)((rel >> 4) / NF_TF) + 8 * (int)((rel >> 4) % NF_TF)) + (int)(rel & 15);
    if (f < 3 * NW) {  // four consecutive features never straddle a segment (NW % 4 == 0)
      const int seg = f / NW, wi = f - seg * NW;
      const float* bs = seg == 0 ? a.bq : (seg == 1 ? a.bk : a.bv);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bs ? bs[h0 * dk + wi + r] : 0.f;
    }
  } else if (e < im.off_b1()) {
    pick(e - im.off_bo(), NF_TY, 4, a.bo, D);
  } else if (e < im.off_w2()) {
    pick(e - im.off_b1(), NF_TA, 8, a.b1, A);
  } else {
    pick(e - im.off_w2(), NF_TA, 8, a.w2, A);
  }
  *reinterpret_cast<f32x4*>(img + e) = v;
}

// Diagnostic build only (make stamps; tools/nf_stamps.py): per-wave s_memtime stamps at the phase boundaries, written
// to a buffer nothing else reads.  The shipped library contains none of this.
#ifdef XNRS_NF_STAMPS
__device__ unsigned long long* g_nf_stamps_dev = nullptr;
#define g_nf_stamps g_nf_stamps_dev
constexpr int NF_NSTAMP = 32;
#define NF_STAMP(I)                                                                                             \
  do {                                                                                                          \
    if (g_nf_stamps && lane == 0 && blockIdx.x < 1024)                                                          \
      g_nf_stamps[((size_t)blockIdx.x * 8 + wave) * NF_NSTAMP + (I)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
#else
#define NF_STAMP(I)
#endif

// k loop of the three big products.  Two fragment register sets: the fragments of step ks+1 are loaded WHILE the
// MFMAs of step ks issue.  Three things hipcc does not do by itself, each measured (tools/nf_stamps.py):
//   * the loop body is straight-line code -- every load unconditional, the last one re-reads step NK-1 -- so the
//     vmcnt / lgkmcnt waits are exact counts (with the prefetch inside `if (ks + 1 < NK)` it waited vmcnt(0) at the
//     join, i.e. for the prefetch it had just issued);
//   * sched_barriers pin the order (left alone the scheduler sinks the loads ~20 MFMAs into the burst);
//   * the loads are SPREAD over the burst, one group behind each quarter of the MFMAs (step j of the four k of a
//     fragment): the two waves of a SIMD run in lockstep (fair MFMA arbitration), so a cluster of ~25 non-MFMA
//     instructions at the top of the body left the matrix pipe idle ~10 % of every step in BOTH of them.
#define NF_SB __builtin_amdgcn_sched_barrier(0)
#define NF_HALF(CUR, NXT, KSN) \
  mma(CUR, IC<0>{});           \
  NF_SB;                       \
  load_w(NXT, KSN);            \
  NF_SB;                       \
  mma(CUR, IC<1>{});           \
  NF_SB;                       \
  load_b(NXT, KSN, IC<0>{});   \
  NF_SB;                       \
  mma(CUR, IC<2>{});           \
  NF_SB;                       \
  load_b(NXT, KSN, IC<1>{});   \
  NF_SB;                       \
  mma(CUR, IC<3>{});           \
  NF_SB;
// MFMA arbitration between the two waves of a SIMD goes to the OLDER wave: left alone, waves 0-3 finished a 41 k-cycle
// projection loop 6 k cycles before waves 4-7, which then ran the tail alone (exposed waits) while waves 0-3 sat in the
// barrier.  Alternating a raised priority between the two halves every k step keeps them level.
#define NF_PRIO(FIRST)                                             \
  if ((wave >= 4) == (FIRST)) __builtin_amdgcn_s_setprio(1);       \
  else __builtin_amdgcn_s_setprio(0);
#define NF_KLOOP(NK)                                                \
  load_w(IC<0>{}, 0);                                               \
  load_b(IC<0>{}, 0, IC<0>{});                                      \
  load_b(IC<0>{}, 0, IC<1>{});                                      \
  NF_SB;                                                            \
  for (int ks = 0; ks + 1 < (NK); ks += 2) {                        \
    NF_PRIO(true)                                                   \
    NF_HALF(IC<0>{}, IC<1>{}, ks + 1)                               \
    const int ksn = ks + 2 < (NK) ? ks + 2 : (NK)-1;                \
    NF_PRIO(false)                                                  \
    NF_HALF(IC<1>{}, IC<0>{}, ksn)                                  \
  }                                                                 \
  __builtin_amdgcn_s_setprio(0);                                    \
  if ((NK)&1) {                                                     \
    mma(IC<0>{}, IC<0>{});                                          \
    mma(IC<0>{}, IC<1>{});                                          \
    mma(IC<0>{}, IC<2>{});                                          \
    mma(IC<0>{}, IC<3>{});                                          \
  }

// FOLD: the out-projection is applied by the caller once per news (NewsFusedArgs::fold): no Y accumulators, no
// out-projection loops (16 % of the workgroup's time at D = 320), the attention rows O of every head group go to the
// workgroup's slot of an L2-resident scratch and come back into R1 once the token rows are dead.
// PERSISTENT: a workgroup walks news groups blockIdx.x, + gridDim.x, ... (its scratch slot is its own for the launch).
template <int DK4, int NPW, bool FOLD>
__global__ __launch_bounds__(NF_THREADS, NPW == 1 ? 4 : 2) void news_fused_kernel(NewsFusedArgs a, NfImg im, const float* __restrict__ img, int HG,
                                                                   int LQ, int LY, int64_t n_wg) {
  constexpr int TR = 2 * NPW, TRC = NPW;  // 16-row tiles per workgroup (32 virtual rows per news); per wave in (c)
  constexpr int TF = NF_TF, TY = NF_TY, TA = NF_TA;
  constexpr int dk = 4 * DK4;  // head width: compile-time, so the attention core is one straight run of MFMAs
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.S, D = a.D, H = a.n_heads;
  const int nrow = NPW * S;
  float* r1 = smem;                      // X, later Y: [nrow][LY]
  float* r2 = r1 + nrow * LY;            // Q|K|V of a head group: [nrow][LQ]
  float* epart = r2 + nrow * LQ;         // [8 waves][NPW * 32] partial fc2 scores
  float* aw = epart + 8 * NPW * 32;      // [2][NPW * 32] pooling weights: exp(e) m, then normalised

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wave & 3, wr = wave >> 2;  // out-projection: feature quarter, row half (NPW = 2: news wr)
  const int c = lane & 15, g = lane >> 4;
  float* const osc = FOLD ? a.o_scratch + (size_t)blockIdx.x * nrow * D : nullptr;  // this workgroup's O rows [nrow][D]
  for (int64_t wgi = blockIdx.x; wgi < n_wg; wgi += gridDim.x) {
  const int64_t news0 = wgi * NPW;

  // ---- token rows -> R1, coalesced 16-byte chunks.  A news past the end of the batch is a duplicate of the last one
  // (computed, never stored), so no uninitialised LDS is ever read.
  {
    const int cpr = D >> 2;  // 16-byte chunks per row
    for (int i = tid; i < nrow * cpr; i += NF_THREADS) {
      const int row = i / cpr, ch = i - row * cpr;
      const int nw = row / S, s = row - nw * S;
      int64_t news = news0 + nw;
      if (news >= a.n_seq) news = a.n_seq - 1;
      const int64_t src = a.ids ? (int64_t)a.ids[news] : news;
      *reinterpret_cast<f32x4*>(&r1[row * LY + ((4 * ch) ^ NF_SWZ(row))]) =
          __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + (src * S + s) * (int64_t)D + 4 * ch));
    }
  }

  // ---- this lane's token row in each row tile: news nw = rt >> 1 owns the virtual rows 32 nw .. 32 nw + 31; rows
  // >= S (tile padding) are clamped duplicates -- every product treats token rows independently -- and only their
  // stores are predicated
  int prow[TR], psw[TR], pdl[TR];  // physical row of the lane in each row tile, its column swizzle, the same as an offset
  bool lds_ok[TR];
#pragma unroll
  for (int rt = 0; rt < TR; ++rt) {
    const int sp = (rt & 1) * 16 + c;
    lds_ok[rt] = sp < S;
    prow[rt] = (rt >> 1) * S + (sp < S ? sp : S - 1);
    psw[rt] = NF_SWZ(prow[rt]);
    // columns of the form 16 j + 4 g (k chunks, accumulator chunks): bit 2 is g & 1, so the flip is a per-lane constant
    // +-4 that rides in the row offset -- no XOR in the k loops.  (A clamped k-tail read lands on D - 4 +- 4 instead of
    // (D - 4) ^ 4: both are written columns of the row, and the weight image is zero there.)
    pdl[rt] = psw[rt] ? ((g & 1) ? -4 : 4) : 0;
  }
  // the row tiles wr * TRC .. of the out-projection
  int prow_c[TRC], psw_c[TRC], pdl_c[TRC];
  bool ok_c[TRC];
#pragma unroll
  for (int i = 0; i < TRC; ++i) {
    const int rt = wr * TRC + i;
    const int sp = (rt & 1) * 16 + c;
    ok_c[i] = sp < S;
    prow_c[i] = (rt >> 1) * S + (sp < S ? sp : S - 1);
    psw_c[i] = NF_SWZ(prow_c[i]);
    pdl_c[i] = psw_c[i] ? ((g & 1) ? -4 : 4) : 0;
  }

  f32x4 yacc[TY][TRC];
#pragma unroll
  for (int t = 0; t < TY; ++t)
#pragma unroll
    for (int i = 0; i < TRC; ++i) yacc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = im.nk;
  const float* imgl = img + lane * 4;  // this lane's 16 bytes of every weight fragment
  const float inv_sq = a.scaled ? 1.f / sqrtf((float)dk) : 1.f;
  constexpr int nfull = dk >> 4, nrem = (dk & 15) >> 2, ndt = (dk + 15) >> 4;
  NF_STAMP(0);
  __syncthreads();  // X is in R1
  NF_STAMP(1);

  const int n_groups = im.n_groups;
  for (int grp = 0; grp < n_groups; ++grp) {
    const int h0 = grp * HG;
    const int nh = (H - h0 < HG) ? H - h0 : HG;
    const int NW = nh * dk;  // width of each of the Q, K, V column blocks of this group
    const int NF = 3 * NW;

    // ================= (a) Q|K|V columns of the group: out^T[f][row] = W_f . x_row (+ bias) -> R2
    // Every wave computes the same COMPILE-TIME number of tiles (tile t of wave w is tile w + 8 t); a tile past the
    // end of the product is all zeros in the image and its result is dropped (1 tile of 16 at D = 320): no branches
    // in the k loop.
    {
      const float* wa = imgl + im.off_qkv(grp) + (size_t)wave * TF * NF_FRAG;
      f32x4 acc[TF][TR];
#pragma unroll
      for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int rt = 0; rt < TR; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 fa[2][TF], fb[2][TR];
      auto load_w = [&](auto ST, int ks) {
        constexpr int st = decltype(ST)::value;
        const float* wk = wa + (size_t)ks * (8 * TF * NF_FRAG);
#pragma unroll
        for (int t = 0; t < TF; ++t) fa[st][t] = *reinterpret_cast<const f32x4*>(wk + t * NF_FRAG);
      };
      auto load_b = [&](auto ST, int ks, auto HALF) {  // the row tiles of news HALF
        constexpr int st = decltype(ST)::value, h = decltype(HALF)::value;
        int kc = ks * 16 + 4 * g;
        if (kc > D - 4) kc = D - 4;  // k tail: the weight image is zero there, the read only has to stay inside the row
#pragma unroll
        for (int rt = NPW * h; rt < NPW * h + NPW; ++rt) fb[st][rt] = *reinterpret_cast<const f32x4*>(&r1[prow[rt] * LY + pdl[rt] + kc]);
      };
      auto mma = [&](auto ST, auto J) {
        constexpr int st = decltype(ST)::value, j = decltype(J)::value;
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
          for (int rt = 0; rt < TR; ++rt)
            acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][t][j], fb[st][rt][j], acc[t][rt], 0, 0, 0);
      };
      NF_KLOOP(nk)
      NF_STAMP(2 + 6 * (grp & 3));
      // the previous group's out-projection has to be done with R2 before it is overwritten
      __syncthreads();
      NF_STAMP(3 + 6 * (grp & 3));
#pragma unroll
      for (int t = 0; t < TF; ++t) {
        const int f0 = 16 * (wave + 8 * t) + 4 * g;
        if (f0 < NF) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(img + im.off_bqkv(grp) + (wave * TF + t) * 16 + 4 * g);
#pragma unroll
          for (int rt = 0; rt < TR; ++rt)
            if (lds_ok[rt]) *reinterpret_cast<f32x4*>(&r2[prow[rt] * LQ + pdl[rt] + f0]) = acc[t][rt] + bv;
        }
      }
    }
    __syncthreads();
    NF_STAMP(4 + 6 * (grp & 3));

    // ================= (b) attention core per (news, head): a wave owns whole units; O overwrites Q in place
    {
      const int n_units = NPW * nh;
      for (int u = wave; u < n_units; u += 8) {
        const int nw = u / nh, hh = u - nw * nh;
        const int R0 = nw * S;                                        // first row of this news in the image
        const int cQ = hh * dk, cK = NW + hh * dk, cV = 2 * NW + hh * dk;  // column bases of the head's Q, K, V
        int64_t news = news0 + nw;
        if (news >= a.n_seq) news = a.n_seq - 1;
        const int64_t msrc = a.ids ? (int64_t)a.ids[news] : news;
        // K fragments: full 16-feature blocks as 16-byte reads (step j: feature 16 fb + 4 g + j), the d_k % 16
        // tail one MFMA per 4 features (feature 16 nfull + 4 e + g)
        f32x4 kf[2][nfull > 0 ? nfull : 1];
        float kr[2][nrem > 0 ? nrem : 1];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          int key = 16 * kt + c;
          if (key > S - 1) key = S - 1;
          const float* kp = r2 + (R0 + key) * LQ;
          const int swk = NF_SWZ(R0 + key);
#pragma unroll
          for (int fb = 0; fb < nfull; ++fb) kf[kt][fb] = *reinterpret_cast<const f32x4*>(kp + ((cK + 16 * fb + 4 * g) ^ swk));
#pragma unroll
          for (int e = 0; e < nrem; ++e) kr[kt][e] = kp[(cK + 16 * nfull + 4 * e + g) ^ swk];
        }
        // V^T fragments: vv[kt][dt][r] = V[key 16 kt + 4 g + r][16 dt + c]
        float vv[2][ndt][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int dt = 0; dt < ndt; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int key = 16 * kt + 4 * g + r;
              if (key > S - 1) key = S - 1;  // its probability is exactly 0
              const int dv = 16 * dt + c;
              vv[kt][dt][r] = (dv < dk) ? r2[(R0 + key) * LQ + ((cV + dv) ^ NF_SWZ(R0 + key))] : 0.f;
            }
        // both 16-query tiles in one straight run (the two softmax chains interleave); for S <= 16 the second tile
        // is a clamped duplicate whose stores are predicated off
        f32x4 sc[2][2];
        int qrow[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int query = 16 * qt + c;
          qrow[qt] = query < S ? query : S - 1;
          const float* qp = r2 + (R0 + qrow[qt]) * LQ;
          const int swq = NF_SWZ(R0 + qrow[qt]);
          sc[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
          sc[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int fb = 0; fb < nfull; ++fb) {
            const f32x4 qf = *reinterpret_cast<const f32x4*>(qp + ((cQ + 16 * fb + 4 * g) ^ swq));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                sc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][fb][j], qf[j], sc[qt][kt], 0, 0, 0);
          }
#pragma unroll
          for (int e = 0; e < nrem; ++e) {
            const float qr = qp[(cQ + 16 * nfull + 4 * e + g) ^ swq];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][e], qr, sc[qt][kt], 0, 0, 0);
          }
        }
        // sc[qt][kt][r] = S[query 16 qt + c][key 16 kt + 4 g + r]: scale, QUERY-row mask (layers.py:142-144), softmax
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const float mq = a.mask ? a.mask[msrc * S + qrow[qt]] : 1.f;
          float mx = -INFINITY;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = 16 * kt + 4 * g + r;
              float sv = sc[qt][kt][r] * inv_sq;
              if (mq == 0.f) sv = -1e9f;
              if (key >= S) sv = -INFINITY;
              sc[qt][kt][r] = sv;
              mx = fmaxf(mx, sv);
            }
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          float sum = 0.f;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float ev = attn_exp(sc[qt][kt][r] - mx);
              sc[qt][kt][r] = ev;
              sum += ev;
            }
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          const float inv_sum = 1.f / sum;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[qt][kt][r] *= inv_sum;
        }
        // O^T[dv][query] = V^T P^T; 4 consecutive dv per lane -> one 16-byte store over this query's Q columns
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int dt = 0; dt < ndt; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)  // S <= 16: the second key tile is all P = 0
#pragma unroll
              for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[kt][dt][r], sc[qt][kt][r], o, 0, 0, 0);
            const int query = 16 * qt + c, dv0 = 16 * dt + 4 * g;
            if (query < S && dv0 < dk) {
              if (FOLD) *reinterpret_cast<f32x4*>(osc + (size_t)(nw * S + query) * D + (h0 + hh) * dk + dv0) = o;
              else *reinterpret_cast<f32x4*>(r2 + (R0 + query) * LQ + ((cQ + dv0) ^ NF_SWZ(R0 + query))) = o;
            }
          }
      }
    }
    NF_STAMP(5 + 6 * (grp & 3));
    // (FOLD: no out-projection reads R2 here, and the next group's projection epilogue waits at its own barrier before
    // it overwrites R2 -- a wave that is done with its heads walks straight into the next k loop)
    if (!FOLD) __syncthreads();
    NF_STAMP(6 + 6 * (grp & 3));

    // ================= (c) out-projection, this group's columns of Wo: Y^T[d][row] += Wo[d][h0 d_k + k] O[row][k]
    // wave (wf, wr): feature tiles wf + 4 t, the two row tiles of news wr
    if constexpr (!FOLD) {
      const float* wc = imgl + im.off_wo(grp) + (size_t)wf * TY * NF_FRAG;
      const int nkc = (NW + 15) >> 4;
      f32x4 fa[2][TY], fb[2][TRC];
      auto load_w = [&](auto ST, int ks) {
        constexpr int st = decltype(ST)::value;
        const float* wk = wc + (size_t)ks * (4 * TY * NF_FRAG);
#pragma unroll
        for (int t = 0; t < TY; ++t) fa[st][t] = *reinterpret_cast<const f32x4*>(wk + t * NF_FRAG);
      };
      auto load_b = [&](auto ST, int ks, auto HALF) {
        constexpr int st = decltype(ST)::value, h = decltype(HALF)::value;
        const int k = ks * 16 + 4 * g;  // k < NW + 16 <= LQ; columns past NW meet zeros of the image
        if (h < TRC) fb[st][h] = *reinterpret_cast<const f32x4*>(&r2[prow_c[h < TRC ? h : 0] * LQ + pdl_c[h < TRC ? h : 0] + k]);
      };
      auto mma = [&](auto ST, auto J) {
        constexpr int st = decltype(ST)::value, j = decltype(J)::value;
#pragma unroll
        for (int t = 0; t < TY; ++t)
#pragma unroll
          for (int i = 0; i < TRC; ++i)
            yacc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][t][j], fb[st][i][j], yacc[t][i], 0, 0, 0);
      };
      NF_KLOOP(nkc)
      NF_STAMP(7 + 6 * (grp & 3));
    }
  }

  // ================= Y = att output (+ bo) -> R1 [row][LY]  (X is dead: every group's projection has read it)
  __syncthreads();  // (FOLD: also publishes the O rows every wave wrote to the scratch)
  NF_STAMP(26);
  if constexpr (FOLD) {  // the attention rows O come back from the scratch (L2 hits: written a few microseconds ago)
    const int cpr = D >> 2;
    for (int i = tid; i < nrow * cpr; i += NF_THREADS) {
      const int row = i / cpr, ch = i - row * cpr;
      *reinterpret_cast<f32x4*>(&r1[row * LY + ((4 * ch) ^ NF_SWZ(row))]) = *reinterpret_cast<const f32x4*>(osc + (size_t)row * D + 4 * ch);
    }
  } else {
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      const int d0 = 16 * (wf + 4 * t) + 4 * g;
      if (d0 < D) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(img + im.off_bo() + (wf * TY + t) * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < TRC; ++i)
          if (ok_c[i]) *reinterpret_cast<f32x4*>(&r1[prow_c[i] * LY + pdl_c[i] + d0]) = yacc[t][i] + bv;
      }
    }
  }
  __syncthreads();
  NF_STAMP(27);

  // ================= fc1 + tanh + fc2: e[row] = sum_a w2[a] tanh(W1[a] . Y[row] + b1[a])   (layers.py:60)
  {
    const float* w1i = imgl + im.off_w1() + (size_t)wave * TA * NF_FRAG;
    f32x4 acc[TA][TR];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
      for (int rt = 0; rt < TR; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][TA], fb[2][TR];
    auto load_w = [&](auto ST, int ks) {
      constexpr int st = decltype(ST)::value;
      const float* wk = w1i + (size_t)ks * (8 * TA * NF_FRAG);
#pragma unroll
      for (int t = 0; t < TA; ++t) fa[st][t] = *reinterpret_cast<const f32x4*>(wk + t * NF_FRAG);
    };
    auto load_b = [&](auto ST, int ks, auto HALF) {
      constexpr int st = decltype(ST)::value, h = decltype(HALF)::value;
      int kc = ks * 16 + 4 * g;
      if (kc > D - 4) kc = D - 4;
#pragma unroll
      for (int rt = NPW * h; rt < NPW * h + NPW; ++rt) fb[st][rt] = *reinterpret_cast<const f32x4*>(&r1[prow[rt] * LY + pdl[rt] + kc]);
    };
    auto mma = [&](auto ST, auto J) {
      constexpr int st = decltype(ST)::value, j = decltype(J)::value;
#pragma unroll
      for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int rt = 0; rt < TR; ++rt)
          acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][t][j], fb[st][rt][j], acc[t][rt], 0, 0, 0);
    };
    NF_KLOOP(nk)
    NF_STAMP(28);
    float e[TR];
#pragma unroll
    for (int rt = 0; rt < TR; ++rt) e[rt] = 0.f;
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      // b1 / w2 images are zero past A: a padded hidden unit contributes w2 * tanh(0 + 0) = 0
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(img + im.off_b1() + (wave * TA + t) * 16 + 4 * g);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(img + im.off_w2() + (wave * TA + t) * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < TR; ++rt) e[rt] = fmaf(w2[r], apply_act(acc[t][rt][r] + b1[r], a.tanh_act), e[rt]);
    }
#pragma unroll
    for (int rt = 0; rt < TR; ++rt) {
      float v = e[rt];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (g == 0) epart[wave * (NPW * 32) + 16 * rt + c] = v;
    }
  }
  NF_STAMP(29);
  __syncthreads();
  NF_STAMP(30);

  // ================= a = exp(e + b2) * m  (un-stabilised, layers.py:61-62), normalise (+1e-8), weighted sum of Y
  if (tid < NPW * 32) {
    const int nw = tid >> 5, sp = tid & 31;
    int64_t news = news0 + nw;
    if (news >= a.n_seq) news = a.n_seq - 1;
    const int64_t msrc = a.ids ? (int64_t)a.ids[news] : news;
    float v = 0.f;
    if (sp < S) {
      float ev = a.b2 ? a.b2[0] : 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) ev += epart[w * (NPW * 32) + tid];
      v = expf(ev);
      if (a.mask) v *= a.mask[msrc * S + sp];
    }
    aw[tid] = v;
  }
  __syncthreads();
  if (tid < NPW * 32) {  // a / (sum_s a + 1e-8), layers.py:63
    const int nw = tid >> 5;
    float den = 0.f;
    for (int s = 0; s < S; ++s) den += aw[nw * 32 + s];
    aw[NPW * 32 + tid] = aw[tid] / (den + 1e-8f);
    if (FOLD && (tid & 31) == 0 && news0 + nw < a.n_seq) a.asum[news0 + nw] = den / (den + 1e-8f);  // sum_i a_i
  }
  __syncthreads();
  const int d4n = D >> 2;
  for (int idx = tid; idx < NPW * d4n; idx += NF_THREADS) {  // one 16-byte column chunk of one news per thread
    const int nw = idx / d4n, d4 = idx - nw * d4n;
    const int64_t news = news0 + nw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
      const float w = aw[NPW * 32 + nw * 32 + s];
      const f32x4 yv = *reinterpret_cast<const f32x4*>(&r1[(nw * S + s) * LY + ((4 * d4) ^ NF_SWZ(nw * S + s))]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(w, yv[r], acc[r]);
    }
    if (news < a.n_seq) *reinterpret_cast<f32x4*>(a.p + news * a.ldp + 4 * d4) = acc;
  }
  // (The MLP head stays a GEMM pair over ALL news after this kernel.  Running it here -- one thread per output
  // feature, both news per weight chunk -- was measured: every workgroup then streams the head's 590 KB of weights
  // for 2 news, 10 microseconds per round at L2 -> CU bandwidth, no better than the two launches at 1024 news and far
  // worse at 28 000.)
  if (a.hm && tid < NPW) {
    const int64_t news = news0 + tid;
    if (news < a.n_seq) {
      const int64_t msrc = a.ids ? (int64_t)a.ids[news] : news;
      float ms = 0.f;
      for (int s = 0; s < S; ++s) ms += a.mask ? a.mask[msrc * S + s] : 1.f;
      a.hm[news] = fminf(fmaxf(ms, 0.f), 1.f);
    }
  }
  NF_STAMP(31);
  __syncthreads();  // R1 / aw are rewritten by the next news group
  }
}

int stride8(int n) {  // smallest stride >= n with stride % 16 == 8: conflict-free 16-byte fragment reads (16 rows x 4 chunks)
  int s = (n + 15) / 16 * 16 + 8;
  if (s - 16 >= n) s -= 16;
  return s;
}

}  // namespace

bool news_fused_plan(int S, int D, int n_heads, int A, NewsFusedPlan* plan, int npw) {
  if (npw != 1 && npw != 2) return false;
  if (S <= 0 || S > 32 || D < 4 || D % 4 != 0 || D > 16 * 4 * NF_TY || n_heads <= 0 || D % n_heads != 0) return false;
  const int dk = D / n_heads;
  if (dk % 4 != 0 || dk > 32 || A <= 0 || A > 16 * 8 * NF_TA) return false;
  int hg = (16 * 8 * NF_TF) / (3 * dk);
  if (hg > n_heads) hg = n_heads;
  if (hg < 1) return false;
  const int lq = stride8(3 * hg * dk), ly = stride8(D);
  const size_t floats = (size_t)npw * S * (lq + ly) + 8 * npw * 32 + 2 * npw * 32;
  if (floats * 4 > 160 * 1024) return false;
  if (plan) {
    plan->npw = npw;
    plan->hg = hg;
    plan->lq = lq;
    plan->ly = ly;
    plan->lds_bytes = floats * 4;
    plan->n_groups = (n_heads + hg - 1) / hg;
    plan->nk = (D + 15) / 16;
    plan->nkc = (hg * dk + 15) / 16;
    plan->img_bytes = NfImg{plan->n_groups, plan->nk, plan->nkc}.total() * sizeof(float);
  }
  return true;
}

// upper bound of NewsFusedPlan::img_bytes over every head count (workspace queries do not know n_heads): a full head
// group covers at least 192 of the 256 Q|K|V columns a pass can hold, so there are at most ceil(3D / 192) + 1 groups
size_t news_fused_img_bound_bytes(int S, int D, int A) {
  if (S <= 0 || S > 32 || D < 4 || D % 4 != 0 || D > 16 * 4 * NF_TY || A <= 0 || A > 16 * 8 * NF_TA) return 0;
  const int ng = (3 * D + 191) / 192 + 1;
  return NfImg{ng, (D + 15) / 16, 16}.total() * sizeof(float);
}

namespace {
// One instantiation per head width (d_k / 4 = 1 .. 8) and news count; each needs its dynamic-LDS limit raised once PER
// DEVICE (hipFuncSetAttribute acts on the current device's code object): the result is cached per (device, instantiation).
constexpr int NF_MAX_DEV = 64;
std::mutex g_attr_mu;
signed char g_attr_state[NF_MAX_DEV][8][2][2];  // 0 unknown, 1 ok, -1 refused (e.g. a part with less than 160 KB of LDS)

template <int Q, int N, bool F>
bool nf_attr_ok(int dev) {
  if (dev < 0 || dev >= NF_MAX_DEV) return false;
  std::lock_guard<std::mutex> lk(g_attr_mu);
  signed char& st = g_attr_state[dev][Q - 1][N - 1][F ? 1 : 0];
  if (st == 0) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&news_fused_kernel<Q, N, F>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();  // a refusal is an answer, not a sticky error of the caller's stream
    st = e == hipSuccess ? 1 : -1;
  }
  return st == 1;
}

bool nf_attr_ok(int q, int npw, bool fold, int dev) {
#define NF_A(Q) \
  case Q:       \
    return npw == 1 ? (fold ? nf_attr_ok<Q, 1, true>(dev) : nf_attr_ok<Q, 1, false>(dev)) \
                    : (fold ? nf_attr_ok<Q, 2, true>(dev) : nf_attr_ok<Q, 2, false>(dev));
  switch (q) {
    NF_A(1) NF_A(2) NF_A(3) NF_A(4) NF_A(5) NF_A(6) NF_A(7) NF_A(8)
    default: return false;
  }
#undef NF_A
}
}  // namespace

// Everything launch_news_fused needs beyond the shape (news_fused_plan): 16-byte aligned operands and a device that
// grants the kernel its 160 KB of dynamic LDS.  The dispatcher asks BEFORE it commits to the fused kernel, so a batch
// this kernel cannot take runs on the GEMM pipeline instead of failing.
bool news_fused_ready(const NewsFusedArgs& a) {
  NewsFusedPlan p;
  const int npw = a.npw == 1 ? 1 : 2;
  if (!news_fused_plan(a.S, a.D, a.n_heads, a.A, &p, npw) || a.d_k * a.n_heads != a.D) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(a.x) || !al16(a.wq) || !al16(a.wk) || !al16(a.wv) || !al16(a.wo) || !al16(a.w1) || !a.img || !al16(a.img))
    return false;
  if (a.fold && (!a.o_scratch || !a.asum || !al16(a.o_scratch))) return false;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return nf_attr_ok(a.d_k / 4, npw, a.fold != 0, dev);
}

// persistent grid: one workgroup per CU with 2 news each, two per CU with 1 news each (NewsFusedPlan); the scratch holds
// every workgroup's O rows: <= 512 x 32 x 320 floats = 21 MB, written and read back within a news group's ~80 us
namespace {
int nf_cu_count() {
  static std::mutex mu;
  static int cus[NF_MAX_DEV] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NF_MAX_DEV) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}
constexpr int NF_MAX_GRID = 1024;  // scratch slots reserved (2 workgroups x 512 CUs at most)
}  // namespace

size_t news_fused_scratch_bytes(int S, int D) { return (size_t)NF_MAX_GRID * (size_t)S * (size_t)D * sizeof(float); }

hipError_t launch_news_fused(const NewsFusedArgs& a_in, hipStream_t stream) {
  NewsFusedArgs a = a_in;
  a.tanh_act = knobs().fast_tanh ? ACT_TANH_FAST : 2;
  if (a.n_seq <= 0) return hipSuccess;
  NewsFusedPlan p;
  const int npw = a.npw == 1 ? 1 : 2;
  if (!news_fused_ready(a) || !news_fused_plan(a.S, a.D, a.n_heads, a.A, &p, npw)) return hipErrorInvalidValue;
  // prologue: the weights in MFMA fragment order (see the file header); ~2 MB, rebuilt per call -- the ABI keeps no state
  const NfImg im{p.n_groups, p.nk, p.nkc};
  const size_t n4 = im.total() / 4;
  hipLaunchKernelGGL(news_fused_prep_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a, im, p.hg, a.img);
  hipError_t pe = hipGetLastError();
  if (pe != hipSuccess) return pe;
  const int64_t n_wg = (a.n_seq + p.npw - 1) / p.npw;
  int64_t grid = (int64_t)nf_cu_count() * (npw == 1 ? 2 : 1);  // persistent: what is co-resident
  if (grid > NF_MAX_GRID / npw) grid = NF_MAX_GRID / npw;       // (a workgroup's scratch slot is npw * S * D floats)
  if (grid > n_wg) grid = n_wg;
  hipError_t rc = hipErrorInvalidValue;
#define NF_LAUNCH(Q, N, F)                                                                                                    \
  hipLaunchKernelGGL((news_fused_kernel<Q, N, F>), dim3((unsigned)grid), dim3(NF_THREADS), p.lds_bytes, stream, a, im, a.img, \
                     p.hg, p.lq, p.ly, n_wg);                                                                                 \
  rc = hipGetLastError();
#define NF_CASE(Q)                  \
  case Q:                           \
    if (npw == 1) {                 \
      if (a.fold) {                 \
        NF_LAUNCH(Q, 1, true)       \
      } else {                      \
        NF_LAUNCH(Q, 1, false)      \
      }                             \
    } else {                        \
      if (a.fold) {                 \
        NF_LAUNCH(Q, 2, true)       \
      } else {                      \
        NF_LAUNCH(Q, 2, false)      \
      }                             \
    }                               \
    break;
  switch (a.d_k / 4) {
    NF_CASE(1) NF_CASE(2) NF_CASE(3) NF_CASE(4) NF_CASE(5) NF_CASE(6) NF_CASE(7) NF_CASE(8)
    default: break;
  }
#undef NF_CASE
#undef NF_LAUNCH
  return rc;
}

}  // namespace xnrs

#ifdef XNRS_NF_STAMPS
extern "C" int xnrs_debug_nf_set_stamps(unsigned long long* dev_buf) {  // diagnostic build only
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(xnrs::g_nf_stamps_dev), &dev_buf, sizeof(dev_buf));
}
#endif
