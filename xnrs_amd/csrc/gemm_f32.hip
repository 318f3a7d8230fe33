// fp32 GEMM on the CDNA4 matrix cores.
//
//   forward   C = act(A . W^T + bias)            A row-major [M,K], W = nn.Linear weight [N,K]     (ROW, WT)
//   backward  dX = (dY . W) (*) f'(aux) [+ C]    A = dY [M,N'] row-major, B = W as [K'=N'][N=K_in] (ROW, KN)
//             dW = dY^T . X                      A = dY read k-major (A^T[k=m][i]), B = X [k=m][n]  (COL, KN)
//
// Replaces nn.Linear (and its autograd) as used by the reference hot path
// (xnrs/models/components/layers.py:60,128-130,154; news_encoding.py:27-31).  Numerics:
// v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain (no TF32/xf32 on gfx950), so results differ from
// torch-CPU only by summation order.
//
// Design (gfx950, 64-wide waves); measurements and the dropped alternatives: DESIGN.md section 4.1
//   * workgroup = 256 threads = 4 waves in a 2x2 grid; wave tile (32*TM)x(32*TN), i.e. the block tile is
//     (64*TM)x(64*TN): 128x128 by default, 128x64 / 64x128 / 64x64 picked by a padded-work estimate.
//   * k-contiguous operands (ROW / WT): a lane fetches 4 consecutive k with ONE ds_read_b128 and feeds 4 MFMA
//     steps from it: step j of an 8-wide k group uses k = 4*(lane>>5) + j for A and B alike (any bijection of k is
//     legal as long as A and B agree).
//   * shipped forward configuration (template PIPE 5, BK 16, BUF, MINW 4): K tiles of 16, 64-byte LDS rows with
//     the 16-byte chunk XOR-swizzled by (row>>2)&3 (conflict-free ds_read_b128 and ds_write_b128), two LDS buffers
//     (32 KB), ONE register set, an explicitly interleaved instruction stream (one LDS store, one tile load and the
//     next fragment reads behind each slot of TM*TN MFMAs, pinned by sched_barrier), raw buffer loads whose
//     bounds check zero-fills ragged rows / k tails, registers capped at 128 -> 4 workgroups per CU.
//     PIPE 1 (plain double buffering, BK 32, rows padded to 36 floats) serves the scalar-load fallback.
//   * k-major operands (COL / KN, the backward's dW and small-M dX): LDS image [k][i] (rows padded by 4),
//     fragments are 4 ds_read_b32 with the 32 lanes of a half-wave on 32 consecutive floats; PIPE 5, BK 32,
//     2 workgroups per CU.
//   * optional row gather on the rows of a ROW-layout A / KN-layout B (device-resident news-token table + ids:
//     SURVEY.md section 8 a0) folded into the per-thread row offsets, so gathered rows are still read as full lines.
//   * XCD-aware block order: each XCD walks a contiguous range of the tile sequence; column GROUPS of tiles
//     outermost so the weight panels an XCD touches stay in its 4 MB L2.
//   * split-K (dW: the contraction is the token-row count, the output is a small weight matrix): each k-slice
//     writes its own fp32 slab; a second kernel sums the slabs in a fixed order (bitwise reproducible, no float
//     atomics).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace xnrs {

constexpr int KALIGN = 32;  // split-K slices are aligned to the largest K tile

// Out-of-range lanes of a tile load read this zero line instead of being zeroed AFTER the load: the
// select is on the ADDRESS, so no instruction depends on the loaded data until the LDS store and the
// loads of two tiles can stay in flight (a select on the data makes hipcc wait right behind the load).
__device__ __attribute__((aligned(16))) float g_zero_line[4] = {0.f, 0.f, 0.f, 0.f};

// PIPE selects the software pipeline:
//   1: plain double buffering -- loads of tile t+1 fly during the MFMAs of tile t, LDS store + barrier at the
//      end of the tile (used for the scalar-load fallback; 117 TF on the Q/K/V projection)
//   5: the same two LDS buffers and ONE register set, but an explicitly interleaved instruction stream
//      (default; 130-136 TF).  Measured alternatives that were dropped: two register sets / LDS store
//      mid-stream without interleave 121 TF; three LDS buffers with cross-barrier fragment prefetch at one
//      workgroup per CU 112 TF, at two per CU 130 TF; 128x256 / 256x128 block tiles (2 waves/SIMD) 131 TF
//      against 133 TF for the default on the same device.
// BUF: tile loads are raw buffer loads (ROW/WT layouts, no gather, operands <= 1 GB): the row offset is a
// loop-invariant 32-bit VGPR, the k offset rides in the scalar offset, and out-of-range rows / k chunks
// are handled by the hardware bounds check (offset bit 30 set -> beyond num_records -> returns 0).  The
// loop then carries ~10 VALU instructions per K tile instead of ~50 (64-bit adds + address selects).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0x40000000u;

// GATH (with BUF): the rows of A are gathered (or A is too large for a buffer descriptor): A is read through per-thread
// 64-bit row pointers, UNCONDITIONALLY -- rows past M are clamped to row M-1 (their results are never stored) and the k
// tail is clamped to K-4 (it meets the zeros the bounds check returns for B) -- while B keeps the raw buffer loads (one
// VGPR offset per weight row here: the scalar-offset form of the dense kernel measured 1.5 % slower in this variant).
// Measured inside the B = 512 step (tools/bench_idpath.py, same ids, interleaved): Q/K/V projection 37.6 ms gathered vs
// 35.5 ms dense = 6 % on that GEMM, 3.7 % on the step (the first gather variant -- a select between row pointer and zero
// line per load, 139 VGPRs, 45 spills, 3 workgroups per CU -- cost 7.1 % / 4.4 %).  What the rest is and is not:
//   * not the random rows: the same kernel over a table that IS the materialised batch, ids = 0..n-1, takes the same
//     time (bench_idpath.py "seq_ids") -- the cost is the per-lane 64-bit address form of the load;
//   * a descriptor cannot replace it: the rows of one tile belong to up to four news anywhere in a 10-GB table; a raw
//     buffer / saddr offset reaches 4 GB from one wave-uniform base, and a STRUCTURED buffer load (idxen: base + row
//     index * 3072 + offset) wraps at the same 4 GB -- rows >= 2^32 / stride read as zeros (tools/probes/struct_buffer.hip);
//   * advancing the row pointers in place (no clamp, no 64-bit temporaries, 4 instead of 12 extra VALU per K tile)
//     measured SLOWER (6.3 % on the step): the add writes the address registers of the load issued just before it.
// KG (k-major operands): how the k rows are gathered -- 0: not at all; 1: both operands through one-row-per-id lists
// (the live-row backward); 2: anything else, decided at run time.  A template parameter because the run-time tests
// (`if (gather_ids) { if (S == 1) ...`) sat in front of each of the 8 tile loads of an iteration: 29 basic blocks, the
// interleaved MFMA / load schedule gone, 3.4 VALU instructions per MFMA and 65 % matrix-pipe occupancy in the dW GEMMs
// (profiles/r02_train_step_pmc.txt).
// MDEV (forward layout): the row count is a device scalar (GemmArgs::m_dev), read once into SGPRs -- its own
// instantiations (a first version wrote it into the by-value argument block: that moved the whole 400-byte struct to
// scratch in EVERY variant -- 59 spilled VGPRs in the main forward kernel, 140 -> 84 TFLOP/s).
template <int TM, int TN, bool A_COL, bool B_KN, bool VEC, int PIPE, int BK, bool BUF = false, int MINW = 2, bool GATH = false,
          int KG = 2, bool RDOT = false, bool MDEV = false>
__global__ __launch_bounds__(256, MINW) void gemm_f32_kernel(const GemmArgs a, int m_tiles, int n_tiles_seg, int gn) {
  static_assert(!MDEV || (!A_COL && !B_KN), "device row counts: forward layout only");
  static_assert(!BUF || (!A_COL && !B_KN && VEC), "buffer loads are implemented for the forward layout");
  static_assert(!RDOT || PIPE == 5, "the fused row dots ride on the interleaved pipeline (its MFMA call is the swapped one)");
  static_assert(!GATH || BUF, "the gathered-A variant keeps buffer loads for B");
  constexpr int BM = 64 * TM, BN = 64 * TN;
  // k-contiguous LDS tile rows: BK = 32 -> padded to 36 floats (conflict-free ds_read_b128, measured
  // SQ_LDS_BANK_CONFLICT = 0); BK = 16 -> 64-B rows, UNPADDED, with the 16-byte chunk index XOR-swizzled by
  // (row >> 2) & 3: reads of a 16-lane group then cover 16 distinct 16-B slots and the 8-lane groups of
  // ds_write_b128 cover two whole rows = 32 distinct banks (the padded 20-float rows measured a 2-way
  // write conflict on every store, 33 % of the LDS cycles), and the tile shrinks from 40 to 32 KB.
  constexpr bool SWZ = BK == 16;
  constexpr int LDK = SWZ ? BK : BK + 4;
  constexpr int CHK = BK / 4;  // 16-byte chunks per k-contiguous row
  constexpr int RPP = 256 / CHK;  // rows per staging pass
  constexpr int AR = BM / RPP, BR = BN / RPP;  // 16-byte chunks per thread per operand tile
  constexpr int LDA = A_COL ? (BM + 4) : LDK;
  constexpr int LDB = B_KN ? (BN + 4) : LDK;
  constexpr int A_SZ = A_COL ? BK * LDA : BM * LDA;
  constexpr int B_SZ = B_KN ? BK * LDB : BN * LDB;
  constexpr int NBUF = 2;
  __shared__ __attribute__((aligned(16))) float As[NBUF][A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][B_SZ];

  // rows that exist: a.M, or -- MDEV -- the device scalar *a.m_dev (a.M was only the grid's worst case), kept in SGPRs.
  // The tile walk below then runs over the row tiles that exist, and the workgroups past them leave at once: mapped over
  // the launched grid instead, the live row tiles would all fall on the first XCDs (each XCD owns a contiguous range of
  // the walk) -- measured: a pass with 25 % of its rows live took as long as a full one.
  int64_t Mrun = a.M;
  int nwg = gridDim.x;
  if constexpr (MDEV) {
    const int64_t md = load_dev_scalar(a.m_dev);
    Mrun = ((int64_t)__builtin_amdgcn_readfirstlane((int)(md >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)md);
    if (Mrun > a.M) Mrun = a.M;
    m_tiles = (int)((Mrun + BM - 1) / BM);
    nwg = m_tiles * n_tiles_seg * a.nseg;
    if ((int)blockIdx.x >= nwg) return;  // workgroup-uniform, before any barrier
  }
  // ---- XCD-aware tile order (bijective for any grid size); blockIdx.y = k slice
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // tile walk: column GROUPS of gn tiles outermost, then the M tiles, then the gn tiles of the group, so the
  // weight panels an XCD touches over a long stretch (gn * BN * K * 4 <= ~2.5 MB) stay in its 4 MB L2 instead
  // of streaming through it once per handful of M tiles (Q/K/V projection: 18 column tiles = 7 MB of W).
  const int n_tiles = n_tiles_seg * a.nseg;
  const int grp = wgid / (gn * m_tiles);
  const int rem = wgid - grp * gn * m_tiles;
  const int gw = (n_tiles - grp * gn < gn) ? n_tiles - grp * gn : gn;  // the last group may be narrower
  const int mt = rem / gw;
  const int nt = grp * gn + (rem - mt * gw);
  const int seg = nt / n_tiles_seg;
  const int nts = nt - seg * n_tiles_seg;
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nts * BN;  // column inside the segment

  // contraction indices are 32-bit in the kernel (the launcher refuses K >= 2^31): the k-tail tests and tile offsets of
  // the inner loop are then single VALU / SALU instructions instead of 64-bit compare-and-select pairs
  int64_t kps = a.k_per_split;
  int Ktot = (int)a.K;
  if constexpr (A_COL && B_KN) {
    if (a.k_dev) {  // contraction length on the device (GemmArgs::k_dev): slices cut here, wave-uniform
      const int kd = __builtin_amdgcn_readfirstlane((int)load_dev_scalar(a.k_dev));
      Ktot = kd < Ktot ? (kd > 0 ? kd : 0) : Ktot;
      kps = (((int64_t)Ktot + gridDim.y - 1) / gridDim.y + KALIGN - 1) / KALIGN * KALIGN;
      if (kps < KALIGN) kps = KALIGN;
    }
  }
  const int kbeg = (int)((int64_t)blockIdx.y * kps);
  const int kend = (int)((kbeg + kps < Ktot) ? kbeg + kps : Ktot);

  const float* __restrict__ W = a.W[seg];
  const float* __restrict__ bias = a.bias[seg];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging maps
  // k-contiguous tile: chunk lc (4 floats of k) of row lr + RPP*i
  const int lc = tid % CHK, lr = tid / CHK;
  // k-major tile: chunk (4 floats along i) cA of k-row kA + KROWS*i
  constexpr int CHA = BM / 4, KRA = 256 / CHA;  // chunks per k-row, k-rows per pass
  constexpr int CHB = BN / 4, KRB = 256 / CHB;
  const int cA = tid % CHA, kA = tid / CHA;
  const int cB = tid % CHB, kB = tid / CHB;

  // Every global load is unconditional (out-of-range lanes are pointed at g_zero_line), so the compiler's
  // vmcnt bookkeeping stays exact and the two-tiles-ahead pipeline below really leaves a whole tile in
  // flight across the LDS store.
  const float* pa[AR];
  const float* pb[BR];
  unsigned a_ok = 0, b_ok = 0;  // bit i: row / column i of this thread is inside the matrix
  // BUF: ONE byte offset per operand in a VGPR -- (row lr, chunk lc) of the tile; row lr + RPP * i rides in the SCALAR
  // offset (k0 * 4 + i * RPP * ld * 4), and rows / columns past the matrix need no flag: their offset is >= num_records
  // (= rows * ld * 4) and the bounds check, which includes the scalar offset (tools/probes/raw_soffset.hip), returns
  // zeros.  (With one VGPR offset per row the kernel sat at the 128-VGPR cap with 4 spilled loop invariants that were
  // re-read from scratch -- through the same vmcnt queue as the tile loads -- in every K tile.)
  unsigned offA0 = 0, offB0 = 0;
  unsigned offBv[BR];  // GATH: one VGPR offset per weight row after all (measured faster there: see the GATH note above)
  int dA = 0, dB = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if constexpr (BUF) {
    if constexpr (!GATH) rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)(Mrun * a.lda * 4), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((int64_t)a.Nseg * a.ldw * 4), 0x00020000);
    if constexpr (GATH) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int64_t gr = m0 + lr + RPP * i;
        int64_t src = gr < Mrun ? gr : Mrun - 1;
        if (a.gather_ids) {
          const int64_t n = src / a.gather_S;
          src = (int64_t)a.gather_ids[n] * a.gather_S + (src - n * a.gather_S);
        }
        pa[i] = a.A + src * a.lda + 4 * lc;
      }
    } else {
      const int64_t gr = m0 + lr;
      offA0 = gr < Mrun ? (unsigned)((gr * a.lda + 4 * lc) * 4) : BUF_OOB;
      dA = (int)(RPP * a.lda * 4);
    }
    const int col = n0 + lr;
    offB0 = col < a.Nseg ? (unsigned)(((int64_t)col * a.ldw + 4 * lc) * 4) : BUF_OOB;
    dB = (int)(RPP * a.ldw * 4);
    if constexpr (GATH) {
#pragma unroll
      for (int i = 0; i < BR; ++i) offBv[i] = offB0 + (unsigned)(i * dB);
    }
  } else if (!A_COL) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int64_t gr = m0 + lr + RPP * i;
      if (gr < Mrun) a_ok |= 1u << i;
      else gr = Mrun - 1;
      int64_t src = gr;
      if (a.gather_ids) {
        const int64_t n = gr / a.gather_S;
        src = (int64_t)a.gather_ids[n] * a.gather_S + (gr - n * a.gather_S);
      }
      pa[i] = a.A + src * a.lda;
    }
  } else {
    if (m0 + 4 * cA < Mrun) a_ok = 1;
  }
  if (!B_KN) {
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      int col = n0 + lr + RPP * i;
      if (col < a.Nseg) b_ok |= 1u << i;
      else col = a.Nseg - 1;
      pb[i] = W + (int64_t)col * a.ldw;
    }
  } else {
    if (n0 + 4 * cB < a.Nseg) b_ok = 1;
  }

  f32x4 ra[1][AR], rb[1][BR];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // P = register set (compile-time), k0 = first k of the tile
  // `only` >= 0 restricts the call to ONE 16-byte chunk (A chunks 0..AR-1, then B chunks): the interleaved
  // pipeline issues the tile loads / LDS stores one at a time between MFMAs.
  auto gload = [&](auto P, int k0, int only = -1) {
    constexpr int p = decltype(P)::value;
    if constexpr (BUF) {
      const unsigned sel = (k0 + 4 * lc < kend) ? 0u : BUF_OOB;  // k tail of the last tile
      const int soff = (int)(k0 * 4);
      int ka = k0;  // GATH: clamp the k tail (B returns zeros there)
      if constexpr (GATH) {
        const int klim = kend - 4 - 4 * lc;
        ka = ka < klim ? ka : klim;
      }
#pragma unroll
      for (int i = 0; i < AR; ++i)
        if (only < 0 || only == i) {
          if constexpr (GATH) ra[p][i] = *reinterpret_cast<const f32x4*>(pa[i] + ka);
          else ra[p][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(offA0 | sel), soff + i * dA, 0));
        }
#pragma unroll
      for (int i = 0; i < BR; ++i)
        if (only < 0 || only == AR + i) {
          if constexpr (GATH) rb[p][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(offBv[i] | sel), soff, 0));
          else rb[p][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(offB0 | sel), soff + i * dB, 0));
        }
      return;
    }
    if (!A_COL) {
      const int64_t k = k0 + 4 * lc;
      const bool kok = k < kend;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        if (only >= 0 && only != i) continue;
        if (VEC) {
          ra[p][i] = *reinterpret_cast<const f32x4*>((kok && ((a_ok >> i) & 1u)) ? pa[i] + k : g_zero_line);
        } else {
          f32x4 v = zero4;
          if ((a_ok >> i) & 1u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < kend) v[e] = pa[i][k + e];
          }
          ra[p][i] = v;
        }
      }
    } else {
      const int64_t mi = m0 + 4 * cA;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        if (only >= 0 && only != i) continue;
        const int k = k0 + kA + KRA * i;
        const bool kok = k < kend;
        // k-row gather in 32-bit arithmetic (the contraction is < 2^31), and with NO division for the one-row-per-id
        // lists of the live-row backward: the 64-bit `srck / gather_S` of the first version was a software division
        // per tile load -- 3.4 VALU instructions per MFMA in the dW GEMMs (profiles/r02_train_step_pmc.txt)
        int srck = kok ? k : kbeg;
        if constexpr (KG == 1) {
          srck = a.gather_ids[srck];
        } else if constexpr (KG == 2) {
          if (a.gather_ids) {
            if (a.gather_S == 1) srck = a.gather_ids[srck];
            else {
              const int n = srck / a.gather_S;
              srck = a.gather_ids[n] * a.gather_S + (srck - n * a.gather_S);
            }
          }
        }
        const float* ptr = a.A + (int64_t)srck * a.lda;
        if (VEC) {
          ra[p][i] = *reinterpret_cast<const f32x4*>((kok && a_ok) ? ptr + mi : g_zero_line);
        } else {
          f32x4 v = zero4;
          if (kok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (mi + e < Mrun) v[e] = ptr[mi + e];
          }
          ra[p][i] = v;
        }
      }
    }
    if (!B_KN) {
      const int64_t k = k0 + 4 * lc;
      const bool kok = k < kend;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        if (only >= 0 && only != AR + i) continue;
        if (VEC) {
          rb[p][i] = *reinterpret_cast<const f32x4*>((kok && ((b_ok >> i) & 1u)) ? pb[i] + k : g_zero_line);
        } else {
          f32x4 v = zero4;
          if ((b_ok >> i) & 1u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < kend) v[e] = pb[i][k + e];
          }
          rb[p][i] = v;
        }
      }
    } else {
      const int ni = n0 + 4 * cB;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        if (only >= 0 && only != AR + i) continue;
        const int k = k0 + kB + KRB * i;
        const bool kok = k < kend;
        int src = kok ? k : kbeg;
        if constexpr (KG == 1) {
          src = a.b_gather_ids[src];
        } else if constexpr (KG == 2) {
          if (a.b_gather_ids) {
            if (a.b_gather_S == 1) src = a.b_gather_ids[src];
            else {
              const int n = src / a.b_gather_S;
              src = a.b_gather_ids[n] * a.b_gather_S + (src - n * a.b_gather_S);
            }
          }
        }
        const float* ptr = W + (int64_t)src * a.ldw;
        if (VEC) {
          rb[p][i] = *reinterpret_cast<const f32x4*>((kok && b_ok) ? ptr + ni : g_zero_line);
        } else {
          f32x4 v = zero4;
          if (kok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ni + e < a.Nseg) v[e] = ptr[ni + e];
          }
          rb[p][i] = v;
        }
      }
    }
  };
  // physical 16-byte chunk of logical chunk c in row `row` of a k-contiguous LDS tile
  auto kswz = [](int row, int c) { return SWZ ? (c ^ ((row >> 2) & 3)) : c; };
  // fused column sums of a k-major A (bias gradients, GemmArgs::colsum): the first column tile of every row tile adds
  // up the A chunks it stages -- each staged exactly once per K tile (`fresh`: the pipeline's redundant re-store of the
  // last tile at the tail must not count twice)
  const bool do_cs = A_COL && a.colsum != nullptr && nt == 0;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  auto sstore = [&](auto P, int buf, int only = -1, bool fresh = true) {
    constexpr int p = decltype(P)::value;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      if (only >= 0 && only != i) continue;
      if (!A_COL) *reinterpret_cast<f32x4*>(&As[buf][(lr + RPP * i) * LDA + 4 * kswz(lr + RPP * i, lc)]) = ra[p][i];
      else {
        *reinterpret_cast<f32x4*>(&As[buf][(kA + KRA * i) * LDA + 4 * cA]) = ra[p][i];
        if (do_cs && fresh) cs += ra[p][i];
      }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      if (only >= 0 && only != AR + i) continue;
      if (!B_KN) *reinterpret_cast<f32x4*>(&Bs[buf][(lr + RPP * i) * LDB + 4 * kswz(lr + RPP * i, lc)]) = rb[p][i];
      else *reinterpret_cast<f32x4*>(&Bs[buf][(kB + KRB * i) * LDB + 4 * cB]) = rb[p][i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;      // row of the 32x32 operand tile this lane feeds
  const int fk = (lane >> 5) * 4;  // k offset inside an 8-wide k group
  const int a_row0 = wm * 32 * TM + frow;
  const int b_row0 = wn * 32 * TN + frow;

  // MFMAs of the 8-wide k groups [KQ0, KQ1) of the tile in LDS buffer `buf`
  auto compute = [&](int buf, auto KQ0, auto KQ1) {
#pragma unroll
    for (int kq = decltype(KQ0)::value; kq < decltype(KQ1)::value; ++kq) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (!A_COL) {
          fa[i] = *reinterpret_cast<const f32x4*>(&As[buf][(a_row0 + 32 * i) * LDA + 4 * kswz(a_row0 + 32 * i, kq * 2 + (fk >> 2))]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fa[i][e] = As[buf][(kq * 8 + fk + e) * LDA + a_row0 + 32 * i];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!B_KN) {
          fb[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(b_row0 + 32 * j) * LDB + 4 * kswz(b_row0 + 32 * j, kq * 2 + (fk >> 2))]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fb[j][e] = Bs[buf][(kq * 8 + fk + e) * LDB + b_row0 + 32 * j];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I4 = std::integral_constant<int, BK / 8>;

  const int nk = (kend - kbeg + BK - 1) / BK;
  auto ktile = [&](int t) { return kbeg + t * BK; };
  if constexpr (PIPE == 1) {
    // one tile ahead: loads of tile t+1 fly during the MFMAs of tile t, LDS store at the end
    if (nk > 0) {
      gload(I0{}, ktile(0));
      sstore(I0{}, 0);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const int buf = t & 1;
      if (t + 1 < nk) gload(I0{}, ktile(t + 1));
      compute(buf, I0{}, I4{});
      if (t + 1 < nk) sstore(I0{}, buf ^ 1);
      __syncthreads();
    }
  } else if constexpr (PIPE == 5) {
    // Two LDS buffers, ONE register set, ONE loop body, explicitly interleaved instruction stream.
    // Why: in-kernel cycle stamps (profiles/r01_gemm_stamps.txt) showed that issuing the 8 loads / 8
    // ds_write_b128 of a tile back to back -- all 4 waves at once, right after the barrier -- blocks an
    // in-order wave for ~1100 of every ~5400 cycles while the shared TA / LDS-write paths drain, and no
    // MFMA of that wave can issue meanwhile.  Here a K tile is cut into "slots" of TM*TN MFMAs (one k
    // step, 256 matrix cycles); behind each slot at most ONE LDS store, ONE tile load and the fragment
    // reads of the next k group are issued, and a sched_barrier pins that order.
    // During iteration t the registers hold tile t+1 (loaded during iteration t-1); chunk j is stored to
    // the other LDS buffer behind slot j+1 and reloaded with tile t+2 behind slot j+2 (~a full iteration
    // of latency hiding).  Fragments are double-buffered so their LDS latency hides behind a k group.  The
    // barrier at the end of the iteration publishes tile t+1 and frees the buffer of tile t.
    constexpr int NKQ = BK / 8;
    constexpr int NSLOT = NKQ * 4;
    constexpr int NCH = AR + BR;
    static_assert(NCH + 2 <= NSLOT, "not enough slots for the tile stores/loads");
    f32x4 fa[2][TM], fb[2][TN];
    auto ldfrag = [&](int st, int buf, int kq) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (!A_COL) {
          fa[st][i] = *reinterpret_cast<const f32x4*>(&As[buf][(a_row0 + 32 * i) * LDA + 4 * kswz(a_row0 + 32 * i, kq * 2 + (fk >> 2))]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fa[st][i][e] = As[buf][(kq * 8 + fk + e) * LDA + a_row0 + 32 * i];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!B_KN) {
          fb[st][j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(b_row0 + 32 * j) * LDB + 4 * kswz(b_row0 + 32 * j, kq * 2 + (fk >> 2))]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fb[st][j][e] = Bs[buf][(kq * 8 + fk + e) * LDB + b_row0 + 32 * j];
        }
      }
    };
    const int last = nk - 1;
    if (nk > 0) {
      gload(I0{}, ktile(0));
      sstore(I0{}, 0);
      gload(I0{}, ktile(1 < last ? 1 : last));
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const int buf = t & 1;
      const int kn = ktile(t + 2 < last ? t + 2 : last);
      ldfrag(0, buf, 0);
#pragma unroll
      for (int kq = 0; kq < NKQ; ++kq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int slot = kq * 4 + e;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              if constexpr (RDOT)  // W . X^T: the block comes out transposed (tokens on the lanes), same fmaf chains
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kq & 1][j][e], fa[kq & 1][i][e], acc[i][j], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kq & 1][i][e], fb[kq & 1][j][e], acc[i][j], 0, 0, 0);
          if (e == 1 && kq + 1 < NKQ) ldfrag((kq + 1) & 1, buf, kq + 1);
          if (slot >= 1 && slot < 1 + NCH) sstore(I0{}, buf ^ 1, slot - 1, t < last);  // tile t+1 -> LDS (redundant at the tail)
          if (slot >= 2 && slot < 2 + NCH) gload(I0{}, kn, slot - 2);        // tile t+2 -> the register just stored
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
  }

  if constexpr (A_COL) {
    if (a.colsum != nullptr && nt == 0) {  // workgroup-uniform
      // the KRA threads that staged the same 4 columns (same cA, k rows kA + KRA i) add up through LDS (the K loop's
      // last barrier has passed: the tile buffers are free)
      f32x4* red = reinterpret_cast<f32x4*>(&As[0][0]);
      red[kA * CHA + cA] = cs;
      __syncthreads();
      if (kA == 0) {
        f32x4 v = red[cA];
#pragma unroll
        for (int r = 1; r < KRA; ++r) v += red[r * CHA + cA];
        float* dst = a.colsum + (int64_t)blockIdx.y * Mrun + m0 + 4 * cA;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (m0 + 4 * cA + e < Mrun) dst[e] = v[e];
      }
    }
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int ccol = lane & 31;
  const int crow = 4 * (lane >> 5);
  const bool split = gridDim.y > 1;
  float* Cout = split ? a.slabs + (int64_t)blockIdx.y * a.slab_stride : a.C;
  if constexpr (RDOT) {
    // fused row dots (GemmArgs::rowdot_w): the accumulators are TRANSPOSED blocks (see the MFMA call) -- lane l holds
    // token row l & 31 and 16 of the block's 32 hidden units -- so the score of a block is an in-lane fmaf chain and one
    // exchange with lane l ^ 32 (rowdot_block_t, kernels.h).  {bias, w} of this workgroup's columns are staged in LDS
    // first (the K loop's last barrier has passed: the tile buffers are free).
    float2* s_bw = reinterpret_cast<float2*>(&As[0][0]);
    static_assert(sizeof(float2) * BN <= sizeof(float) * A_SZ * NBUF, "bias / weight staging fits the A tile buffers");
    for (int c = tid; c < BN; c += 256) {
      const int col = n0 + c;
      const bool cok = col < a.Nseg;
      s_bw[c] = make_float2((bias && cok) ? bias[col] : 0.f, cok ? a.rowdot_w[col] : 0.f);
    }
    __syncthreads();
    const int half = lane >> 5;
    auto dots = [&](auto FAST) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cb = wn * 32 * TN + 32 * j;  // first column of the block inside the tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float sc = rowdot_block_t_lds<decltype(FAST)::value>(acc[i][j], s_bw + cb, half);
          const int64_t row = m0 + wm * 32 * TM + 32 * i + (lane & 31);
          if (lane < 32 && row < Mrun && n0 + cb < a.Nseg) a.rowdot_out[row * a.ldrd + ((n0 + cb) >> 5)] = sc;
        }
      }
    };
    if (a.act == ACT_TANH_FAST) dots(std::true_type{});
    else dots(std::false_type{});
    return;
  }
  if constexpr (!A_COL) {
    if (a.c_scatter) {
      // Row subset in place: the C (and aux) rows follow A's gather list (the live-row / kv-row products of the grad step).
      // The destination row is looked up ONCE per accumulator row -- (i, e) outermost, the TN column blocks inside -- and
      // without the 64-bit division of the general rule when the list holds one row per id (every list of the grad
      // step): the first version divided per ELEMENT, 64 software divisions per thread, a quarter of these launches.
      const int32_t* __restrict__ sids = a.c_scatter_ids ? a.c_scatter_ids : a.gather_ids;
      const bool one = a.gather_S == 1;
      float bvj[TN];
      int colj[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        colj[j] = n0 + wn * 32 * TN + 32 * j + ccol;
        bvj[j] = (bias && !split && colj[j] < a.Nseg) ? bias[colj[j]] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int64_t row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
          if (row >= Mrun) continue;
          if (one) row = sids[row];
          else {
            const int64_t n = row / a.gather_S;
            row = (int64_t)sids[n] * a.gather_S + (row - n * a.gather_S);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (colj[j] >= a.Nseg) continue;
            const int64_t coff = (int64_t)seg * a.Nseg + colj[j];
            float v = acc[i][j][e];
            if (!split) {
              if (a.rowscale) v = fmaf(a.rowscale[row], a.rowscale_vec[colj[j]], v);
              v = apply_act(v + bvj[j], a.act);
              if (a.aux_mode) {
                const float x = a.aux[row * a.ldaux + coff];
                v *= (a.aux_mode == 1) ? (1.f - x * x) : (x > 0.f ? 1.f : 0.f);
              }
              if (a.accumulate) v += Cout[row * a.ldc + coff];
            }
            Cout[row * a.ldc + coff] = v;
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 32 * TN + 32 * j + ccol;
    if (col >= a.Nseg) continue;
    const float bv = (bias && !split) ? bias[col] : 0.f;
    const int64_t coff = (int64_t)seg * a.Nseg + col;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int64_t row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + crow;
        if (row < Mrun) {
          float v = acc[i][j][e];
          if (!split) {
            if (a.rowscale) v = fmaf(a.rowscale[row], a.rowscale_vec[col], v);
            v = apply_act(v + bv, a.act);
            if (a.aux_mode) {
              const float x = a.aux[row * a.ldaux + coff];
              v *= (a.aux_mode == 1) ? (1.f - x * x) : (x > 0.f ? 1.f : 0.f);
            }
            if (a.accumulate) v += Cout[row * a.ldc + coff];
          }
          if (BUF && a.nt_store) __builtin_nontemporal_store(v, &Cout[row * a.ldc + coff]);
          else Cout[row * a.ldc + coff] = v;
        }
      }
    }
  }
}

// Wt[c][r] = W[r][c]: 32x32 tiles through LDS (padded rows), both sides coalesced
__global__ __launch_bounds__(256) void transpose_kernel(const float* W, float* Wt, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * i][tx] = W[(int64_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) Wt[(int64_t)c * rows + r] = tile[tx][ty + 8 * i];
  }
}

// zero `width` floats (multiple of 4, 16-byte aligned) of each of `rows` rows of pitch `ld` floats: the column block of
// one segment inside a [rows, 3D] image.  (hipMemset2DAsync took 311 us for 80 000 x 768 floats, this runs at HBM speed.)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* p, int64_t ld, int w4, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int64_t r = i / w4;
  const int c = (int)(i - r * w4);
  __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(p + r * ld) + c);
}

hipError_t launch_zero_cols(float* p, int64_t ld, int width, int64_t rows, hipStream_t stream) {
  if (rows <= 0 || width <= 0) return hipSuccess;
  if (width % 4 != 0 || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p) & 15) != 0)
    return hipMemset2DAsync(p, (size_t)ld * sizeof(float), 0, (size_t)width * sizeof(float), (size_t)rows, stream);
  const int64_t n4 = rows * (width / 4);
  hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p, ld, width / 4, n4);
  return hipGetLastError();
}

// The rows of a [n_seq*L, 3D] Q|K|V image that the live-row / kv-row projections (api.hip, xnrs_row_lists) leave unwritten,
// zeroed by the mask itself (fp32 [.., L], optionally through news ids): the Q columns of every masked token row, and the
// K|V columns of every row of an all-masked sequence.  One workgroup per sequence; writes 0.37 instead of 0.74 GB at the
// grad step's 80 000 x 2304 image (zero_cols over the whole image: 0.14 ms per encode).
__global__ __launch_bounds__(256) void zero_dead_qkv_kernel(float* qkv, const float* __restrict__ mask,
                                                            const int32_t* __restrict__ ids, int L, int D4) {
  const int64_t seq = blockIdx.x;
  const float* mp = mask + (ids ? (int64_t)ids[seq] : seq) * L;
  int any = 0;
  for (int s = threadIdx.x; s < L; s += 256) any |= mp[s] != 0.f ? 1 : 0;
  const bool empty = !__syncthreads_or(any);
  const int w4 = empty ? 3 * D4 : D4;  // all-masked sequence: Q, K and V; otherwise the Q columns of its masked rows
  f32x4* base = reinterpret_cast<f32x4*>(qkv) + seq * L * (int64_t)(3 * D4);
  for (int s = 0; s < L; ++s) {
    if (!empty && mp[s] != 0.f) continue;  // (uniform)
    f32x4* row = base + (int64_t)s * (3 * D4);
    for (int c = threadIdx.x; c < w4; c += 256) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, row + c);
  }
}

hipError_t launch_zero_dead_qkv(float* qkv, const float* mask, const int32_t* ids, int64_t n_seq, int L, int D, hipStream_t stream) {
  if (n_seq <= 0) return hipSuccess;
  if (D % 4 != 0 || (reinterpret_cast<uintptr_t>(qkv) & 15) != 0 || n_seq > 0x7fffffffLL || !mask)
    return launch_zero_cols(qkv, 3 * (int64_t)D, 3 * D, n_seq * L, stream);
  hipLaunchKernelGGL(zero_dead_qkv_kernel, dim3((unsigned)n_seq), dim3(256), 0, stream, qkv, mask, ids, L, D / 4);
  return hipGetLastError();
}

hipError_t launch_transpose(const float* W, float* Wt, int rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, stream, W,
                     Wt, rows, cols);
  return hipGetLastError();
}

// C (+)= sum_s slabs[s]  (fixed order -> bitwise reproducible); elements past n: the fused bias-gradient partials
// (GemmArgs::colsum, [nsplit][cs_n]) summed the same way into cs_out -- one launch instead of a reduce and a colsum_final
// (C2 / n1 / cs_out2 / cs_n1: elements from n1 on -- whole rows -- go to C2, bias gradients from cs_n1 on to cs_out2:
// GemmArgs::C2, one product for two parameters)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int64_t slab_stride, int nsplit, float* C,
                                                             int64_t n, int accumulate, const float* cs_partial, float* cs_out,
                                                             int cs_n, float* C2, int64_t n1, float* cs_out2, int cs_n1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) {
    const int64_t c = i - n;
    if (c < cs_n) {
      float v = 0.f;
      for (int s = 0; s < nsplit; ++s) v += cs_partial[(int64_t)s * cs_n + c];
      if (cs_out2 && c >= cs_n1) cs_out2[c - cs_n1] = v;
      else cs_out[c] = v;
    }
    return;
  }
  float* dst = (C2 && i >= n1) ? C2 + (i - n1) : C + i;
  float v = accumulate ? *dst : 0.f;
  for (int s = 0; s < nsplit; ++s) v += slabs[(int64_t)s * slab_stride + i];
  *dst = v;
}

// column tiles per group of the tile walk (see the kernel): as many as keep the group's weight panels within
// ~2.5 MB, evened out over the groups.  The k-major (backward) layouts keep the plain M-major walk.
int gemm_group_tiles(int n_tiles, int bn, int64_t K, bool plain) {
  int64_t gn = (int64_t)(2.5 * 1024 * 1024) / ((int64_t)bn * (K > 0 ? K : 1) * 4);
  if (knobs().gemm_group >= 0) gn = knobs().gemm_group;  // development knob: tiles per group, 0 = plain walk
  if (plain || gn <= 0 || gn >= n_tiles) return n_tiles;
  const int groups = (int)((n_tiles + gn - 1) / gn);
  return (n_tiles + groups - 1) / groups;
}

template <int TM, int TN, bool A_COL, bool B_KN>
static hipError_t launch_cfg(const GemmArgs& a, bool vec, int nsplit, hipStream_t stream) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int64_t m_tiles = (a.M + BM - 1) / BM;
  const int n_tiles_seg = (a.Nseg + BN - 1) / BN;
  const int64_t grid = m_tiles * n_tiles_seg * a.nseg;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 g((unsigned)grid, (unsigned)nsplit);
  const int gn = gemm_group_tiles(n_tiles_seg * a.nseg, BN, a.K, A_COL || B_KN);
  // development knobs for in-process A/B runs (tools/bench_gemm.py; kernels.h: Knobs).  Default (pipe 6): PIPE 5,
  // BK 16, registers capped for 4 workgroups per CU on the main tile.
  const int pipe = knobs().gemm_pipe;
  const int bk = knobs().gemm_bk;
  const bool bufB = knobs().gemm_buf && vec && (int64_t)a.Nseg * a.ldw * 4 <= (int64_t)BUF_OOB;
  const bool buf = bufB && !a.gather_ids && a.M * a.lda * 4 <= (int64_t)BUF_OOB;
  // gathered rows / an A operand beyond the 1-GB descriptor window: pointers for A, buffer loads for B (K >= 4: the
  // k-tail clamp reads K-4 .. K-1)
  const bool gath = bufB && !buf && a.K >= 4;  // (row scatter of C is an epilogue matter: any A-load variant serves it)
#define XNRS_LAUNCH(VECV, PIPEV, BKV, BUFV, MINWV)                                                                  \
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, A_COL, B_KN, VECV, PIPEV, BKV, BUFV, MINWV>), g, dim3(256), 0, stream, a, \
                     (int)m_tiles, n_tiles_seg, gn)
#define XNRS_LAUNCH_GATH(MINWV)                                                                                       \
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, A_COL, B_KN, true, 5, 16, true, MINWV, true>), g, dim3(256), 0, stream, a, \
                     (int)m_tiles, n_tiles_seg, gn)
  if constexpr (!A_COL && !B_KN) {
    if (a.m_dev) {  // device row count (the device-compacted padding-free encoder): its own instantiations
      if (!vec || !(buf || gath)) return hipErrorInvalidValue;
      if (a.rowdot_out) {
        if (buf)
          hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, false, 2, true, true>), g, dim3(256), 0,
                             stream, a, (int)m_tiles, n_tiles_seg, gn);
        else
          hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, true, 2, true, true>), g, dim3(256), 0,
                             stream, a, (int)m_tiles, n_tiles_seg, gn);
      } else if (buf) {
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, false, 2, false, true>), g, dim3(256), 0,
                           stream, a, (int)m_tiles, n_tiles_seg, gn);
      } else {
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, true, 2, false, true>), g, dim3(256), 0,
                           stream, a, (int)m_tiles, n_tiles_seg, gn);
      }
      return hipGetLastError();
    }
    if (a.rowdot_out) {  // fused row dots: its own instantiation, so that the plain kernel's registers stay as they are
      if (!vec || !(buf || gath)) return hipErrorInvalidValue;
      if (buf)
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, false, 2, true>), g, dim3(256), 0, stream,
                           a, (int)m_tiles, n_tiles_seg, gn);
      else  // gathered A rows (a news table, the compact rows of the padding-free path) or an A beyond 1 GB
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, false, false, true, 5, 16, true, 4, true, 2, true>), g, dim3(256), 0, stream,
                           a, (int)m_tiles, n_tiles_seg, gn);
      return hipGetLastError();
    }
  }
  if (!vec) XNRS_LAUNCH(false, 1, 32, false, 2);
  else if constexpr (TM == 2 && TN == 2 && !A_COL && !B_KN) {  // forward main tile: all variants are built
    if (pipe == 1 && buf) XNRS_LAUNCH(true, 1, 32, true, 2);
    else if (pipe == 1) XNRS_LAUNCH(true, 1, 32, false, 2);
    else if (pipe == 5 && bk == 16 && buf) XNRS_LAUNCH(true, 5, 16, true, 2);
    else if (pipe == 5 && buf) XNRS_LAUNCH(true, 5, 32, true, 2);
    else if (pipe == 5) XNRS_LAUNCH(true, 5, 32, false, 2);
    else if (buf) XNRS_LAUNCH(true, 5, 16, true, 4);
    else if (gath) XNRS_LAUNCH_GATH(4);
    else XNRS_LAUNCH(true, 5, 16, false, 3);  // pointers + zero-line selects on both operands (row scatter, > 1 GB
                                               // weights): 139 VGPRs -> 3 workgroups per CU
  } else if constexpr (!A_COL && !B_KN) {
    // the smaller forward tiles use the main tile's configuration too (BK 16, registers capped for 4 WG/CU):
    // +19..25 % on the Q/K/V projection at D = 300 / 320 against BK 32 at 2 WG/CU
    if (buf) XNRS_LAUNCH(true, 5, 16, true, 4);
    else if (gath) XNRS_LAUNCH_GATH(4);
    else XNRS_LAUNCH(true, 5, 16, false, 4);
  } else if constexpr (A_COL && B_KN) {
    // backward dW = dY^T . X (both operands k-major): BK 32, 2 workgroups per CU (BK 16 / 4 per CU measured no better);
    // one instantiation per k-row gather mode
    const bool g_none = !a.gather_ids && !a.b_gather_ids;
    const bool g_live = a.gather_ids && a.b_gather_ids && a.gather_S == 1 && a.b_gather_S == 1;
#define XNRS_LAUNCH_KG(KGV)                                                                                              \
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, A_COL, B_KN, true, 5, 32, false, 2, false, KGV>), g, dim3(256), 0, stream, a, \
                     (int)m_tiles, n_tiles_seg, gn)
    if (g_none) XNRS_LAUNCH_KG(0);
    else if (g_live) XNRS_LAUNCH_KG(1);
    else XNRS_LAUNCH_KG(2);
#undef XNRS_LAUNCH_KG
  } else {
    XNRS_LAUNCH(true, 5, 32, false, 2);  // ROW x KN (small-M dX products)
  }
#undef XNRS_LAUNCH
#undef XNRS_LAUNCH_GATH
  return hipGetLastError();
}

template <bool A_COL, bool B_KN>
static hipError_t launch_layout(const GemmArgs& a, bool vec, int nsplit, hipStream_t stream) {
  // pick the tile with the least estimated time: rounds of co-resident workgroups x padded tile work
  // (forward variants: 4 workgroups/CU x 256 CUs per round, k-major variants 2/CU); bigger tiles have slightly
  // better MFMA duty.  Measured with the forced-tile knob: 30 720 x 256 x 300 -> 87 TF at 128x128 (480 of
  // 1 024 slots filled), 98 TF at 64x64.
  const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const double eff[4] = {1.0, 0.94, 0.94, 0.86};
  int best = 0;
  double best_t = 1e300;
  for (int c = 0; c < 4; ++c) {
    const int64_t bm = 64 * cand[c][0], bn = 64 * cand[c][1];
    // (m_dev launches: the rows expected to exist, GemmArgs::m_fill_hint, not the capacity the grid is sized for)
    const int64_t m_exp = (a.m_dev && a.m_fill_hint > 0.f && a.m_fill_hint < 1.f) ? (int64_t)(a.M * (double)a.m_fill_hint) + 1 : a.M;
    const int64_t wgs = ((m_exp + bm - 1) / bm) * ((a.Nseg + bn - 1) / bn) * a.nseg * nsplit;
    const double slots = (A_COL || B_KN) ? 512.0 : 1024.0;
    // fractional rounds above one: workgroups are re-dispatched one by one, so 7.3 rounds of 128x128 tiles do not cost
    // 8 (whole rounds made the k-major dX GEMMs pick 128x64 tiles: 87 TF)
    const double rounds = (double)wgs / slots > 1.0 ? (double)wgs / slots : 1.0;
    const double t = rounds * (double)(bm * bn) / eff[c];
    if (t < best_t * 0.999) {
      best_t = t;
      best = c;
    }
  }
  if (knobs().gemm_tile >= 0 && knobs().gemm_tile <= 3) best = knobs().gemm_tile;  // development knob
  switch (best) {
    case 0: return launch_cfg<2, 2, A_COL, B_KN>(a, vec, nsplit, stream);
    case 1: return launch_cfg<2, 1, A_COL, B_KN>(a, vec, nsplit, stream);
    case 2: return launch_cfg<1, 2, A_COL, B_KN>(a, vec, nsplit, stream);
    default: return launch_cfg<1, 1, A_COL, B_KN>(a, vec, nsplit, stream);
  }
}

size_t gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int ns0 = gemm_pick_splits(M, N, K, false), ns1 = gemm_pick_splits(M, N, K, true);
  const int ns = ns0 > ns1 ? ns0 : ns1;
  return ns > 1 ? (size_t)ns * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int gemm_pick_splits(int64_t M, int64_t N, int64_t K, bool dw_kernel) {
  // aim for one round of 128x128 workgroups (2 per CU x 256 CUs for the k-major variants, 3 per CU for gemm_dw.hip; one
  // 256x256 workgroup per CU for its big tile) and at least 256 contraction steps per slice
  int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int64_t slots = dw_kernel ? 768 : 512;
  if (dw_kernel && knobs().gemm_dw_tile != 128 && M >= 384 && N >= 384) {
    tiles = ((M + 255) / 256) * ((N + 255) / 256);
    slots = 256;
  }
  int64_t ns = slots / tiles;
  const int64_t max_by_k = K / 256;
  if (ns > max_by_k) ns = max_by_k;
  if (ns > 64) ns = 64;
  if (ns < 1) ns = 1;
  return (int)ns;
}

// ---- process-global state of the library: the knobs (read once) and the forward-GEMM arithmetic mode
static Knobs read_knobs() {
  Knobs k;
  auto num = [](const char* name, long long dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoll(e) : dflt;
  };
  k.gemm_pipe = (int)num("XNRS_GEMM_PIPE", 6);
  k.gemm_bk = num("XNRS_GEMM_BK", 32) == 16 ? 16 : 32;
  k.gemm_buf = num("XNRS_GEMM_BUF", 1) != 0;
  k.gemm_group = num("XNRS_GEMM_GROUP", -1);
  k.gemm_tile = (int)num("XNRS_GEMM_TILE", -1);
  k.split_min_tiles = num("XNRS_GEMM_SPLIT_MIN_TILES", 512);
  k.mha_lds = (int)num("XNRS_MHA_LDS", -1);
  k.mha_headwave = num("XNRS_MHA_HEADWAVE", 1) != 0;
  k.mha_pair = num("XNRS_MHA_PAIR", 1) != 0;
  k.mha_bwd_fused = num("XNRS_MHA_BWD_FUSED", 1) != 0;
  k.gemm_dw = (int)num("XNRS_GEMM_DW", 2);
  k.gemm_dw_tile = (int)num("XNRS_GEMM_DW_TILE", 256);
  k.fold_out = (int)num("XNRS_FOLD_OUT", 1);
  k.fc1_rowdot = num("XNRS_FC1_ROWDOT", 1) != 0;
  k.fold_train = (int)num("XNRS_FOLD_TRAIN", 1);
  k.news_fused = (int)num("XNRS_NEWS_FUSED", 1);
  k.news_fused_npw = (int)num("XNRS_NEWS_FUSED_NPW", 0);
  k.fast_tanh = num("XNRS_FAST_TANH", 1) != 0;
  k.additive_fused = (int)num("XNRS_ADDITIVE_FUSED", 1);
  k.af_fbuf = num("XNRS_AF_FBUF", 1) == 2 ? 2 : 1;
  k.mha_skip_masked = num("XNRS_MHA_SKIP_MASKED", 1) != 0;
  k.bwd_side_stream = num("XNRS_BWD_SIDE_STREAM", 1) != 0;
  k.bwd_side_min_rows = num("XNRS_BWD_SIDE_MIN_ROWS", 0);
  const long long m = num("XNRS_GEMM_MODE", 0);
  k.gemm_mode_init = (m >= 0 && m <= 2) ? (int)m : 0;
  return k;
}
static Knobs g_knobs = read_knobs();  // at library load
static std::atomic<int> g_gemm_mode{g_knobs.gemm_mode_init};
const Knobs& knobs() { return g_knobs; }
void reload_knobs() { g_knobs = read_knobs(); }
int gemm_mode() { return g_gemm_mode.load(std::memory_order_relaxed); }
void set_gemm_mode(int mode) { g_gemm_mode.store((mode >= 0 && mode <= 2) ? mode : 0, std::memory_order_relaxed); }

hipError_t launch_gemm_f32(const GemmArgs& a_in, hipStream_t stream, int* nsplit_used) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.Nseg <= 0) return hipSuccess;
  if (a.act == 2 && knobs().fast_tanh) a.act = ACT_TANH_FAST;
  if (a.K >= (1ll << 31)) return hipErrorInvalidValue;  // 32-bit contraction indices in the kernel
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // 16-byte vector loads need the contiguous dimension of each operand on 16-B boundaries
  bool vec = (a.lda % 4 == 0) && (a.ldw % 4 == 0) && al16(a.A);
  vec = vec && (a.a_col ? (a.M % 4 == 0) : (a.K % 4 == 0));
  vec = vec && (a.b_kn ? (a.Nseg % 4 == 0) : (a.K % 4 == 0));
  for (int s = 0; s < a.nseg; ++s) vec = vec && al16(a.W[s]);

  int nsplit = 1;
  if (a.slabs && a.nsplit > 1) {
    nsplit = a.nsplit;
    const int64_t per = ((a.K + nsplit - 1) / nsplit + KALIGN - 1) / KALIGN * KALIGN;  // tile-aligned slices
    a.k_per_split = per;
    nsplit = (int)((a.K + per - 1) / per);
    a.slab_stride = a.M * a.ldc;
    if (a.nseg != 1 || a.ldc != a.Nseg) return hipErrorInvalidValue;
  }
  if (nsplit <= 1) {
    nsplit = 1;
    a.k_per_split = a.K > 0 ? a.K : 1;
  }
  if (nsplit_used) *nsplit_used = nsplit;
  // an output of hundreds of MB (the Q/K/V image of a 65 500-row pass: 604 MB) is written once and read by the next
  // kernel from HBM anyway: non-temporal stores keep it from evicting the operand panels (-0.5 % on the Q/K/V GEMM)
  a.nt_store = (!a.accumulate && !a.c_scatter && a.M * a.ldc * 4 >= (64ll << 20)) ? 1 : 0;
  if (a.colsum && !(a.a_col && a.b_kn)) return hipErrorInvalidValue;  // fused column sums: dW layout only
  if (a.k_dev && !(a.a_col && a.b_kn)) return hipErrorInvalidValue;   // device contraction length: dW layout only
  if (a.colsum_out && !a.colsum) return hipErrorInvalidValue;
  if (a.C2 && (nsplit <= 1 || !(a.a_col && a.b_kn) || a.c2_row0 <= 0 || a.c2_row0 >= a.M)) return hipErrorInvalidValue;  // two
                                                                                                   // destinations: split-K dW only
  bool cs_copy = false;  // one slice: its partial IS the bias gradient (written in place when 16-byte aligned)
  if (a.colsum && a.colsum_out && nsplit == 1) {
    if (al16(a.colsum_out)) a.colsum = a.colsum_out;
    else cs_copy = true;
  }
  if (a.rowdot_out && (a.a_col || a.b_kn || a.nseg != 1 || nsplit != 1 || !a.rowdot_w || a.c_scatter || a.accumulate || a.aux_mode ||
                       (a.act != 2 && a.act != ACT_TANH_FAST)))
    return hipErrorInvalidValue;  // fused row dots: plain forward launches with a tanh epilogue only (the pooler's fc1)
  hipError_t e;
  const int mode = gemm_mode();
  // the split kernel has one tile shape (128x128): below one full round of workgroups the fp32 kernel with its
  // smaller tiles is faster (measured: 4099 x 260 x 300 -> 38 TF fp32 vs 22 TF split)
  const int64_t min_tiles = knobs().split_min_tiles;  // tests force the split kernel onto tiny shapes with 0
  if (a.m_dev && (a.a_col || a.b_kn || nsplit != 1 || mode != 0)) return hipErrorInvalidValue;  // device row count: fp32 forward only
  if (a.rowscale && (a.a_col || a.b_kn || a.nseg != 1 || nsplit != 1 || !a.rowscale_vec || a.c_scatter || a.rowdot_out))
    return hipErrorInvalidValue;  // rank-1 epilogue term: plain forward launches only
  if (mode && !a.rowdot_out && !a.rowscale && !a.a_col && !a.b_kn && vec && nsplit == 1 &&
      ((a.M + 127) / 128) * ((a.Nseg + 127) / 128) * a.nseg >= min_tiles)
    return launch_gemm_split(a, mode == 1 ? 3 : 2, stream);
  if (!a.a_col && !a.b_kn) e = launch_layout<false, false>(a, vec, nsplit, stream);
  else if (!a.a_col && a.b_kn) e = launch_layout<false, true>(a, vec, nsplit, stream);
  else if (a.a_col && a.b_kn) {
    // weight gradients: the register-transposing kernel (gemm_dw.hip) when the shape allows, else the k-major variant
    if (knobs().gemm_dw && gemm_dw_eligible(a)) e = launch_gemm_dw(a, nsplit, stream);
    else e = launch_layout<true, true>(a, vec, nsplit, stream);
  }
  else return hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  if (nsplit > 1) {
    const int64_t n = a.M * a.ldc;
    const int cs_n = (a.colsum && a.colsum_out) ? (int)a.M : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + cs_n + 255) / 256)), dim3(256), 0, stream, a.slabs,
                       a.slab_stride, nsplit, a.C, n, a.accumulate, a.colsum, a.colsum_out, cs_n, a.C2, a.c2_row0 * a.ldc,
                       a.C2 ? a.colsum_out2 : nullptr, (int)a.c2_row0);
    e = hipGetLastError();
  } else if (cs_copy) {
    e = launch_colsum_final(a.colsum, 1, (int)a.M, a.colsum_out, stream);
  }
  return e;
}

}  // namespace xnrs
