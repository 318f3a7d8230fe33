// Weight-gradient GEMM of the backward:  dW[i][j] = sum_m dY[m][i] * X[m][j]   (autograd of nn.Linear, training.py:423)
// Both operands are "k-major": the contraction index m (token row) is the ROW index of dY and of X, so a 16-byte global
// read holds 4 consecutive i (or j) of ONE m -- while v_mfma_f32_32x32x2_f32 wants, per lane, consecutive k of one i.
//
// The generic kernel (gemm_f32.hip, A_COL / B_KN) keeps the tiles k-major in LDS and feeds the MFMAs from four
// ds_read_b32 per fragment: 1 LDS instruction per MFMA, 45 % LDS-array occupancy, 2 workgroups per CU, ~100 TFLOP/s
// (profiles/r02_train_step_pmc.txt: matrix pipe 65 % busy) -- a third of the NRMS train step.
//
// Here the TRANSPOSE happens in registers on the way to LDS, for free: a thread loads a 4 (m) x 4 (i) block -- four
// 16-byte reads of four consecutive rows m, each warp instruction still two full 512-byte row segments -- and stores
// its four COLUMNS as four ds_write_b128 into the i-major, k-contiguous image [i][16 k] of the forward kernel (64-byte
// rows, 16-byte chunk XOR-swizzled by (i >> 2) & 3).  From there on the kernel IS the forward kernel: one ds_read_b128
// per fragment (4 MFMA steps), BK = 16, 32 KB of LDS, 140-154 VGPRs, 3 workgroups per CU (the 128-VGPR build spills).  Threads 0-127 stage dY,
// threads 128-255 stage X (a 16 x 128 tile is 128 blocks of 4 x 4).  One register set: while tile t is multiplied the
// four columns of tile t+1 are stored behind MFMA slots 0-3 and the same registers reloaded with tile t+2 behind slots
// 4-7, one access per slot, pinned by sched_barriers (see gemm_f32.hip PIPE 5 for why).
//   * split-K over the token rows with per-slice slabs + ordered reduce (bitwise reproducible, no float atomics);
//   * k rows optionally gathered: KG 1 = both operands through one-row-per-id lists (the backward over the unmasked
//     token rows), KG 2 = X through a news table (ids, S rows per id); the ids are fetched four slots ahead of the rows;
//   * GemmArgs::colsum: the bias gradient sum_m dY[m][i] falls out of the dY blocks this kernel stages anyway.
#include <type_traits>

#include "kernels.h"

namespace xnrs {

namespace {

__device__ __attribute__((aligned(16))) float g_dw_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int DW_BM = 128, DW_BN = 128, DW_BK = 16;

// LDS image of an operand tile: [i][16 k] as 64-byte rows of four 16-byte chunks -- with the rows PERMUTED inside groups
// of four and the chunk index swizzled so that BOTH access patterns are conflict-free under the hardware's lane groups
// (MI355X_MICROARCH.md, LDS).  Row r = 4 q + d (d = r & 3, q = r >> 2):
//   physical row   = (r & ~3) | ((d + q) & 3)
//   physical chunk = c ^ ((r >> 3) & 3)
// Fragment read (ds_read_b128: banks mod 64 floats = 16 slots of 16 B, slot = 4 (physical row & 3) + physical chunk; the
// four lane groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32; lane l reads row base + (l & 31), one
// logical chunk): a group holds four aligned row quads whose (r >> 3) & 3 are 0, 1, 2, 3 -- each quad fills one chunk
// column with its four physical rows: 16 distinct slots.  Transposed store (ds_write_b128: banks mod 32 floats = 8 slots,
// slot = 4 (physical row & 1) + physical chunk; groups of 8 contiguous lanes = column d of 8 consecutive 4 x 4 blocks,
// rows 4 q + d, q = 8 j .. 8 j + 7): (r >> 3) & 3 pairs the quads (2 m, 2 m + 1) on one chunk, and the row permutation
// gives the two quads of a pair opposite physical-row parities: 8 distinct slots.
// History: v1 swizzled the chunk by (r >> 2) & 3 only (every lane of a store shared d and landed in 4 of the 16 slots:
// 29 % of the LDS cycles were conflict cycles); v2 used c ^ ((b + a) & 3), derived for 16 CONTIGUOUS lanes per read group,
// which the hardware's groups are not: stores clean, every fragment read 2-way -- 33-37 % in the round-3 PMC summary
// (profiles/r03_train_step_pmc.txt).  This is v3.
__device__ __forceinline__ int dw_lds_off(int row, int chunk) {  // float offset of (row, logical chunk) inside a tile image
  const int d = row & 3, q = row >> 2;
  return ((row & ~3) | ((d + q) & 3)) * DW_BK + 4 * (chunk ^ ((row >> 3) & 3));
}

template <int KG>
__global__ __launch_bounds__(256, 3) void gemm_dw_kernel(GemmArgs a, int n_tiles, int tiles, int nsplit) {
  __shared__ __attribute__((aligned(16))) float As[2][DW_BM * DW_BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][DW_BN * DW_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware work order.  The work list is split-major (w = split * tiles + tile); the dispatcher deals workgroup L
  // to XCD L % 8, so XCD x is handed the CONTIGUOUS range of the list that its workgroups (L = x, x + 8, ...) number:
  // all tiles of one K slice then run on one XCD, next to each other in time -- the slice of dY and X comes over the
  // fabric once and its 2 x (tiles per row) re-reads are served by that XCD's L2.  (Dealt round-robin, every XCD pulls
  // nearly every slice: 6 x the HBM/fabric traffic on the 768 x 768 gradients, which is what bounded the k-major kernel.)
  int w;
  {
    const int L = blockIdx.x, W = tiles * nsplit, x = L & 7, q = L >> 3, per = W >> 3, rem = W & 7;
    w = per * x + (x < rem ? x : rem) + q;
  }
  const int split = w / tiles, tile = w - split * tiles;
  const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
  const int m0 = mt * DW_BM, n0 = nt * DW_BN;
  int Ktot = (int)a.K;
  int kps = (int)a.k_per_split;
  if (a.k_dev) {  // contraction length on the device (GemmArgs::k_dev): slices cut here, wave-uniform
    const int kd = __builtin_amdgcn_readfirstlane((int)load_dev_scalar(a.k_dev));
    Ktot = kd < Ktot ? (kd > 0 ? kd : 0) : Ktot;
    kps = ((Ktot + nsplit - 1) / nsplit + DW_BK - 1) / DW_BK * DW_BK;
    if (kps < DW_BK) kps = DW_BK;
  }
  const int kbeg = split * kps;
  const int kend = (kbeg + kps < Ktot) ? kbeg + kps : Ktot;

  // ---- staging role: which operand, which 4 x 4 block of the 16 x 128 tile
  const bool isB = tid >= 128;
  const int t7 = tid & 127;
  const int c4 = t7 & 31;   // columns 4 c4 .. 4 c4 + 3 of the tile
  const int kb = t7 >> 5;   // rows 4 kb .. 4 kb + 3 of the tile
  const float* base = isB ? a.W[0] : a.A;
  const int64_t ld = isB ? a.ldw : a.lda;
  const int col = (isB ? n0 : m0) + 4 * c4;
  const bool col_ok = col < (isB ? a.Nseg : (int)a.M);  // dims are multiples of 4 (launcher): a chunk is all in or all out
  const int32_t* ids = isB ? a.b_gather_ids : a.gather_ids;
  const int gS = isB ? a.b_gather_S : a.gather_S;
  float* sdst = (isB ? &Bs[0][0] : &As[0][0]);
  // transposed store: column e' of the block = row 4 c4 + e' of the LDS image, chunk kb (swizzled)
  int soff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) soff[e] = dw_lds_off(4 * c4 + e, kb);

  const bool do_cs = a.colsum != nullptr && nt == 0 && !isB;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};

  f32x4 R[4];
  int idr[4];  // KG != 0: source rows of the block that is loaded next
  // source row of contraction index k (clamped into the matrix: out-of-slice rows are replaced by the zero line anyway)
  auto fetch_ids = [&](int k0) {
    if constexpr (KG != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int k = k0 + 4 * kb + e;
        if (k > Ktot - 1) k = Ktot - 1;
        if (KG == 1) idr[e] = ids[k];
        else if (ids) {  // KG 2: table gather, S rows per id (X only; dY is dense)
          const int n = k / gS;
          idr[e] = ids[n] * gS + (k - n * gS);
        } else idr[e] = k;
      }
    }
  };
  auto gload = [&](int k0, int e) {
    const int k = k0 + 4 * kb + e;
    const int64_t row = KG != 0 ? idr[e] : k;
    const float* p = (col_ok && k < kend) ? base + row * ld + col : g_dw_zero;
    R[e] = *reinterpret_cast<const f32x4*>(p);
  };
  auto sstore = [&](int buf, int e, bool fresh) {
    const f32x4 v = {R[0][e], R[1][e], R[2][e], R[3][e]};
    *reinterpret_cast<f32x4*>(sdst + buf * (DW_BM * DW_BK) + soff[e]) = v;
    if (do_cs && fresh) cs += R[e];  // (row e of the block: any order, each block element exactly once)
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fc = lane >> 5;
  const int a_row0 = wm * 64 + frow, b_row0 = wn * 64 + frow;
  auto fraga = [&](int buf, int kq, int i) { return *reinterpret_cast<const f32x4*>(&As[buf][dw_lds_off(a_row0 + 32 * i, kq * 2 + fc)]); };
  auto fragb = [&](int buf, int kq, int j) { return *reinterpret_cast<const f32x4*>(&Bs[buf][dw_lds_off(b_row0 + 32 * j, kq * 2 + fc)]); };

  const int nk = (kend - kbeg + DW_BK - 1) / DW_BK;
  const int last = nk - 1;
  auto ktile = [&](int t) { return kbeg + (t < last ? t : last) * DW_BK; };

  // one K tile: multiply tile t (LDS buffer t & 1); the registers hold tile t+1: its four columns go to the other
  // buffer behind slots 0-3, then the same registers are reloaded with tile t+2 behind slots 4-7 (consumed a full half
  // iteration of every co-resident wave later); the row ids of tile t+2 are fetched behind slot 0
  if (nk > 0) {
    fetch_ids(ktile(0));
#pragma unroll
    for (int e = 0; e < 4; ++e) gload(ktile(0), e);
#pragma unroll
    for (int e = 0; e < 4; ++e) sstore(0, e, true);
    fetch_ids(ktile(1));
#pragma unroll
    for (int e = 0; e < 4; ++e) gload(ktile(1), e);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    f32x4 fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[0][i] = fraga(buf, 0, i);
      fb[0][i] = fragb(buf, 0, i);
    }
    const int k2 = ktile(t + 2);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int slot = kq * 4 + e;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kq][i][e], fb[kq][j][e], acc[i][j], 0, 0, 0);
        if (slot == 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[1][i] = fraga(buf, 1, i);
            fb[1][i] = fragb(buf, 1, i);
          }
        }
        if (slot < 4) sstore(buf ^ 1, slot, t < last);
        else gload(k2, slot - 4);
        if (slot == 0) fetch_ids(k2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- bias gradient partials: the four threads (kb = 0..3) that staged the same 4 columns add up through LDS
  if (a.colsum != nullptr && nt == 0) {  // workgroup-uniform
    f32x4* red = reinterpret_cast<f32x4*>(&As[0][0]);
    if (!isB) red[kb * 32 + c4] = cs;
    __syncthreads();
    if (!isB && kb == 0) {
      const f32x4 v = (red[c4] + red[32 + c4]) + (red[64 + c4] + red[96 + c4]);
      float* dst = a.colsum + (int64_t)split * a.M + m0 + 4 * c4;
      if (col_ok) *reinterpret_cast<f32x4*>(dst) = v;
    }
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int ccol = lane & 31, crow = 4 * (lane >> 5);
  float* Cout = nsplit > 1 ? a.slabs + (int64_t)split * a.slab_stride : a.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int colj = n0 + wn * 64 + 32 * j + ccol;
    if (colj >= a.Nseg) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
        if (row < a.M) Cout[row * a.ldc + colj] = acc[i][j][e];
      }
  }
}

// ---- the same product on 256 x 256 tiles: 8 waves (2 per SIMD, <= 256 registers) of 128 x 64 each, ONE workgroup per CU.
// A 128 x 128 tile moves 16 KB of operands per 16-deep K step for 0.5 MFLOP (32 FLOP per byte staged: ~19 GB/s per CU at
// the matrix peak, twice what the per-CU load path sustained in the forward GEMMs); the 256 x 256 tile stages 32 KB per
// 2.1 MFLOP.  Same transposing register stage (threads 0-255 stage dY, 256-511 stage X: a 16 x 256 tile is 256 blocks of
// 4 x 4), same swizzled [i][16 k] LDS image, one register set, stores behind MFMA slots 0-3 and reloads behind 4-7; the
// MFMA loop is additive_fused.hip's (6 fragment reads per 32 MFMAs).
constexpr int DW2_BM = 256, DW2_BN = 256;

template <int KG>
__global__ __launch_bounds__(512, 2) void gemm_dw256_kernel(GemmArgs a, int n_tiles, int tiles, int nsplit) {
  __shared__ __attribute__((aligned(16))) float As[2][DW2_BM * DW_BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][DW2_BN * DW_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves
  int w;
  {  // XCD-contiguous, split-major work list (see gemm_dw_kernel)
    const int L = blockIdx.x, W = tiles * nsplit, x = L & 7, q = L >> 3, per = W >> 3, rem = W & 7;
    w = per * x + (x < rem ? x : rem) + q;
  }
  const int split = w / tiles, tile = w - split * tiles;
  const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
  const int m0 = mt * DW2_BM, n0 = nt * DW2_BN;
  int Ktot = (int)a.K;
  int kps = (int)a.k_per_split;
  if (a.k_dev) {  // contraction length on the device (GemmArgs::k_dev): slices cut here, wave-uniform
    const int kd = __builtin_amdgcn_readfirstlane((int)load_dev_scalar(a.k_dev));
    Ktot = kd < Ktot ? (kd > 0 ? kd : 0) : Ktot;
    kps = ((Ktot + nsplit - 1) / nsplit + DW_BK - 1) / DW_BK * DW_BK;
    if (kps < DW_BK) kps = DW_BK;
  }
  const int kbeg = split * kps;
  const int kend = (kbeg + kps < Ktot) ? kbeg + kps : Ktot;

  const bool isB = tid >= 256;
  const int t8 = tid & 255;
  const int c4 = t8 & 63;   // columns 4 c4 .. 4 c4 + 3 of the tile
  const int kb = t8 >> 6;   // rows 4 kb .. 4 kb + 3 of the K tile
  const float* base = isB ? a.W[0] : a.A;
  const int64_t ld = isB ? a.ldw : a.lda;
  const int col = (isB ? n0 : m0) + 4 * c4;
  const bool col_ok = col < (isB ? a.Nseg : (int)a.M);
  const int32_t* ids = isB ? a.b_gather_ids : a.gather_ids;
  const int gS = isB ? a.b_gather_S : a.gather_S;
  float* sdst = (isB ? &Bs[0][0] : &As[0][0]);
  int soff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) soff[e] = dw_lds_off(4 * c4 + e, kb);
  const bool do_cs = a.colsum != nullptr && nt == 0 && !isB;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  f32x4 R[4];
  int idr[4];
  auto fetch_ids = [&](int k0) {
    if constexpr (KG != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int k = k0 + 4 * kb + e;
        if (k > Ktot - 1) k = Ktot - 1;
        if (KG == 1) idr[e] = ids[k];
        else if (ids) {
          const int n = k / gS;
          idr[e] = ids[n] * gS + (k - n * gS);
        } else idr[e] = k;
      }
    }
  };
  auto gload = [&](int k0, int e) {
    const int k = k0 + 4 * kb + e;
    const int64_t row = KG != 0 ? idr[e] : k;
    const float* p = (col_ok && k < kend) ? base + row * ld + col : g_dw_zero;
    R[e] = *reinterpret_cast<const f32x4*>(p);
  };
  auto sstore = [&](int buf, int e, bool fresh) {
    const f32x4 v = {R[0][e], R[1][e], R[2][e], R[3][e]};
    *reinterpret_cast<f32x4*>(sdst + buf * (DW2_BM * DW_BK) + soff[e]) = v;
    if (do_cs && fresh) cs += R[e];
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fc = lane >> 5;
  const int a_row0 = wm * 128 + frow, b_row0 = wn * 64 + frow;
  f32x4 fa[4], fb[2];
  int fao[4], fbo[2];  // k group 0 of each block row; k group 1: ^ 8 (the chunk index is XORed, and 2 kq + fc = 2 kq ^ fc)
#pragma unroll
  for (int i = 0; i < 4; ++i) fao[i] = dw_lds_off(a_row0 + 32 * i, fc);
#pragma unroll
  for (int j = 0; j < 2; ++j) fbo[j] = dw_lds_off(b_row0 + 32 * j, fc);
  auto ldfrag = [&](int buf, int kq) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&As[buf][fao[i] ^ (8 * kq)]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][fbo[j] ^ (8 * kq)]);
  };

  const int nk = (kend - kbeg + DW_BK - 1) / DW_BK;
  const int last = nk - 1;
  auto ktile = [&](int t) { return kbeg + (t < last ? t : last) * DW_BK; };
  if (nk > 0) {
    fetch_ids(ktile(0));
#pragma unroll
    for (int e = 0; e < 4; ++e) gload(ktile(0), e);
#pragma unroll
    for (int e = 0; e < 4; ++e) sstore(0, e, true);
    fetch_ids(ktile(1));
#pragma unroll
    for (int e = 0; e < 4; ++e) gload(ktile(1), e);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const int k2 = ktile(t + 2);
    ldfrag(buf, 0);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int slot = kq * 4 + e;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        if (slot == 3) ldfrag(buf, 1);  // one fragment set: behind the k group's last MFMAs
        if (slot < 4) sstore(buf ^ 1, slot, t < last);
        else gload(k2, slot - 4);
        if (slot == 0) fetch_ids(k2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  if (a.colsum != nullptr && nt == 0) {  // workgroup-uniform: the four threads (kb) that staged the same 4 columns add up
    f32x4* red = reinterpret_cast<f32x4*>(&As[0][0]);
    if (!isB) red[kb * 64 + c4] = cs;
    __syncthreads();
    if (!isB && kb == 0) {
      const f32x4 v = (red[c4] + red[64 + c4]) + (red[128 + c4] + red[192 + c4]);
      float* dst = a.colsum + (int64_t)split * a.M + m0 + 4 * c4;
      if (col_ok) *reinterpret_cast<f32x4*>(dst) = v;
    }
  }

  const int ccol = lane & 31, crow = 4 * (lane >> 5);
  float* Cout = nsplit > 1 ? a.slabs + (int64_t)split * a.slab_stride : a.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int colj = n0 + wn * 64 + 32 * j + ccol;
    if (colj >= a.Nseg) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 128 + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
        if (row < a.M) Cout[row * a.ldc + colj] = acc[i][j][e];
      }
  }
}

}  // namespace

// Does this kernel serve the launch?  (dW layout, 16-byte aligned operands whose widths are multiples of 4, enough
// output to fill 128 x 128 tiles reasonably; everything else stays on gemm_f32_kernel's k-major variant.)
// Measured on the NRMS train step (profiles/r02_gemm_dw_ab.txt): the live-row gradients (both operands gathered) run
// 329 -> 270 us (~120 -> ~147 TFLOP/s), the dense ones were already at 0.9 of peak on the k-major kernel (765 us for
// 768 x 768 x 96 000) and are 5 % slower here -- so by default (XNRS_GEMM_DW=1) only the live-row launches come here;
// XNRS_GEMM_DW=2 sends every eligible launch (tests), 0 none.
bool gemm_dw_eligible(const GemmArgs& a) {
  if (!a.gather_ids && !(knobs().gemm_dw >= 3 || (knobs().gemm_dw == 2 && gemm_dw_big_tile(a)))) return false;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a.a_col || !a.b_kn || a.nseg != 1 || a.accumulate || a.aux_mode || a.act || a.bias[0]) return false;
  if (a.M % 4 != 0 || a.Nseg % 4 != 0 || a.lda % 4 != 0 || a.ldw % 4 != 0 || !al16(a.A) || !al16(a.W[0])) return false;
  if (a.M < 64 || a.Nseg < 64 || a.K >= (1ll << 31)) return false;
  if (a.K < 8192) return false;  // short contractions: few slices, the 64 x 64 tiles of the generic kernel fill the chip better
  if (a.gather_ids && (a.gather_S != 1 || !a.b_gather_ids || a.b_gather_S != 1)) return false;  // dY: dense, or both
                                                                                                 // operands through live-row lists
  if (a.colsum && (reinterpret_cast<uintptr_t>(a.colsum) & 15) != 0) return false;
  return true;
}

// 256 x 256 tiles (gemm_dw256_kernel) for outputs of at least 1.5 x 1.5 tiles (XNRS_GEMM_DW_TILE=128 forces the small tile)
bool gemm_dw_big_tile(const GemmArgs& a) { return knobs().gemm_dw_tile != 128 && a.M >= 384 && a.Nseg >= 384; }

hipError_t launch_gemm_dw(const GemmArgs& a, int nsplit, hipStream_t stream) {
  if (gemm_dw_big_tile(a)) {
    const int64_t m_tiles = (a.M + DW2_BM - 1) / DW2_BM, n_tiles = (a.Nseg + DW2_BN - 1) / DW2_BN;
    if (m_tiles * n_tiles * nsplit > 0x3fffffffLL) return hipErrorInvalidValue;
    const int tiles = (int)(m_tiles * n_tiles);
    const dim3 g((unsigned)(tiles * nsplit));
    const bool none = !a.gather_ids && !a.b_gather_ids;
    const bool live = a.gather_ids && a.b_gather_ids && a.gather_S == 1 && a.b_gather_S == 1;
    if (none) hipLaunchKernelGGL((gemm_dw256_kernel<0>), g, dim3(512), 0, stream, a, (int)n_tiles, tiles, nsplit);
    else if (live) hipLaunchKernelGGL((gemm_dw256_kernel<1>), g, dim3(512), 0, stream, a, (int)n_tiles, tiles, nsplit);
    else if (!a.gather_ids) hipLaunchKernelGGL((gemm_dw256_kernel<2>), g, dim3(512), 0, stream, a, (int)n_tiles, tiles, nsplit);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  const int64_t m_tiles = (a.M + DW_BM - 1) / DW_BM, n_tiles = (a.Nseg + DW_BN - 1) / DW_BN;
  if (m_tiles * n_tiles * nsplit > 0x3fffffffLL) return hipErrorInvalidValue;
  const int tiles = (int)(m_tiles * n_tiles);
  const dim3 g((unsigned)(tiles * nsplit));
  const bool none = !a.gather_ids && !a.b_gather_ids;
  const bool live = a.gather_ids && a.b_gather_ids && a.gather_S == 1 && a.b_gather_S == 1;
  if (none) hipLaunchKernelGGL((gemm_dw_kernel<0>), g, dim3(256), 0, stream, a, (int)n_tiles, tiles, nsplit);
  else if (live) hipLaunchKernelGGL((gemm_dw_kernel<1>), g, dim3(256), 0, stream, a, (int)n_tiles, tiles, nsplit);
  else if (!a.gather_ids) hipLaunchKernelGGL((gemm_dw_kernel<2>), g, dim3(256), 0, stream, a, (int)n_tiles, tiles, nsplit);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace xnrs
