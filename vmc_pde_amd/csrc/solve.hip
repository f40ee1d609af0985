// On-device solve of the TDVP equation S theta_dot = F.
//
// Reference: tdvp.py:57-94.  The reference diagonalises S on the host (np.linalg.eigh, tdvp.py:61-64) and applies
// a soft spectral cut-off.  The fast path here is a regularised Cholesky of A = S + diagonalShift*diag(S)
// (tdvp.py:50-51 defines that shifted matrix; north_star asks for the Cholesky), blocked right-looking with
// 128-wide panels: diagonal block factor + triangular inverse in LDS, panel solve and trailing update on the fp64
// MFMA tile engine (mfma_tile.hpp).  The spectral (parity) path keeps the reference's formulae: eigh comes from the
// caller, the cut-off coefficients (tdvp.py:82-89) and the mat-vecs (tdvp.py:66,91,93,94) are the kernels below.
#include <cstdlib>

#include "mfma_tile.hpp"

using namespace vmcpde;

namespace {

constexpr int NB = TILE;        // panel width 128
constexpr int DP = NB + 1;      // LDS pitch of the diagonal block (odd: column walks are conflict-free)

// acc -> C[row0.., col0..] (ldc), mode 0: C = acc, mode 1: C -= acc
__device__ __forceinline__ void tile_store_global(const double4_t (&acc)[4][4], double* __restrict__ C, int64_t ldc, int rows, int cols,
                                                  int mode, int tid = -1) {
  const int t = (tid < 0) ? (int)threadIdx.x : tid, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 64 * wr + 16 * i + (lane >> 4) + 4 * r;
        const int col = 64 * wc + 16 * j + (lane & 15);
        if (row < rows && col < cols) {
          double* p = C + (int64_t)row * ldc + col;
          *p = mode ? (*p - acc[i][j][r]) : acc[i][j][r];
        }
      }
}

constexpr int SB = 32;          // sub-block width inside a 128-wide panel
constexpr int SUPER_MIN_BLOCKS = 56;    // from this many 128-wide panels on (P > 7040) the factorisation groups them into super-panels
constexpr int SUPER_WIDTH = 8;          //   of this many: measured best of 1, 2, 4, 8, 16 at P = 8192, 12 288, 16 181 (tools/diag/chol_schedule_ab.sh)
constexpr int FUSED_MAX_TILES = 1024;   // trailing-update size up to which the next diagonal block rides in the same launch

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// One 16x16 block product on the matrix pipe with LDS-resident operands fetched through per-lane functors:
// acc(i,j) += sum_{k<K} fa(i,k) * fb(j,k);  lane l supplies fa(l&15, k4 + (l>>4)) and fb(l&15, k4 + (l>>4)).
// C/D element q of a lane is (row = (l>>4) + 4q, col = l&15)  (f64 MFMA layout).
template <class FA, class FB>
__device__ __forceinline__ void mfma16(double4_t& acc, int K, FA fa, FB fb) {
  const int lr = threadIdx.x & 15, lk = (threadIdx.x & 63) >> 4;
#pragma unroll 4
  for (int k4 = 0; k4 < K; k4 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(lr, k4 + lk), fb(lr, k4 + lk), acc, 0, 0, 0);
}

// d = sqrt(p) and 1/d for a positive normal p: v_rsq_f64 seed + two coupled Goldschmidt steps + one residual correction
// (ocml's sqrt()/rsqrt() carry range-scaling code that bloats the unrolled factor loop into spills).
__device__ __forceinline__ void sqrt_rsqrt(double p, double& d, double& inv_d) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double e = fma(-g, g, p);
  g = fma(e, h, g);
  d = g;
  inv_d = h + h;
}

// Wave 0: Cholesky of the 32x32 diagonal sub-block at (j0,j0) of the LDS block `a` and, in the same sweep, its inverse.
// Lanes 0..31 keep row i of the block in registers and eliminate column after column (right-looking).  Lanes 32..63
// run the column-oriented forward substitution L x = e_q for column q = lane-32 of X = L^-1:  x_j = b_j / L[j][j],
// b_k -= L[k][j] x_j (k > j).  Both halves need exactly the same broadcast per step -- the multipliers L[k][j] of column j
// -- so one v_readlane pair per (j,k) feeds ONE 64-lane FMA as a scalar operand, with a per-lane multiplier (L[i][j] on
// the lower half, x_j on the upper half).  The loop touches no LDS and has no dependent memory round trip.
// It is fully unrolled (static register indices) and free of per-lane compares: lane-dependent 0/1 factors are built with
// v_add_f64 ... clamp, because hipcc keeps compare results alive as SGPR lane masks and spills; there is no branch
// inside (MachineSink would move the FMAs into later blocks and spill their scalar operands), and the broadcasts are
// fenced in groups of 8 so that at most 16 SGPRs are in flight.
// L goes to the lower triangle; X = L^-1 is kept transposed in the dead strict upper part: X[r][c] <-> a[c][r+1].
// scr: 64-double dump line for the masked-out stores.
// (S = 32: the block of the 32-wide schedule, all 64 lanes; S = 16: the 16-wide schedule -- lanes 0..15 rows, 16..31 columns
// of the inverse, lanes 32..63 shadow them with identical values and identical store addresses.)
template <int S, int PITCH>
__device__ __forceinline__ int factor_inverse(double* a, int base /* offset of the block's (0,0) element */, double* scr, bool do_inverse) {
  int lane = threadIdx.x & 63;
  // opaque per call: otherwise LICM hoists the lane-only values below (delta, store addresses) out of the caller's
  // sub-block loop and keeps them alive across it -- hundreds of spilled VGPRs
  asm volatile("" : "+v"(lane));
  const int i = lane & (S - 1);                     // row (lower half) / column of X (upper half)
  const int lane2 = lane & (2 * S - 1);
  const bool lower = lane2 < S;
  const int rowbase = base + i * PITCH;
  const double di = (double)i;
  double v[S];
#pragma unroll
  for (int c = 0; c < S; ++c) {
    const double aic = a[rowbase + c];
    const double delta = 1.0 - __builtin_fmin(__builtin_fabs(di - (double)c), 1.0);
    v[c] = lower ? aic : delta;
  }
  int pinned = 0;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const double p = readlane_f64(v[j], j);
    const bool bad = !(p > 0.0) || !isfinite(p);     // wave-uniform: dead / padded parameter (H7) -> unit row
    pinned += bad ? 1 : 0;
    double d, inv_d;
    sqrt_rsqrt(p, d, inv_d);                         // one coupled iteration instead of sqrt + a division per column
    d = bad ? 1.0 : d;
    inv_d = bad ? 0.0 : inv_d;                       // pinned: no multipliers below, unit diagonal for the inverse
    const double scale = lower ? inv_d : (bad ? 1.0 : inv_d);
    double m = v[j] * scale;                         // L[i][j] (junk for rows above j: never stored) | x_j of column q
    m = (lane2 == j) ? d : m;                        // lane j keeps the diagonal entry itself (also when the pivot is NaN)
    v[j] = m;
    // groups of 8: all broadcasts of a group first (16 distinct SGPRs), then its FMAs -- left to itself hipcc chains
    // readlane, readlane, s_nop, fma through one SGPR pair, which serialises the whole column
#pragma unroll
    for (int g = 0; g < S / 8; ++g) {
      if (8 * g + 7 > j) {                                     // (compile-time after unrolling)
        double lk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) lk[q] = (8 * g + q > j) ? readlane_f64(m, 8 * g + q) : 0.0;   // L[k][j] from lower lane k
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (8 * g + q > j) v[8 * g + q] = fma(-m, lk[q], v[8 * g + q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // lower lanes: L[i][c] for c <= i -> a[j0+i][j0+c];  upper lanes: X[k][q] for k >= q -> a[j0+q][j0+k+1].
  // Everything else (junk above the diagonal, exact zeros that would land on L) is steered to a dump slot by address
  // arithmetic instead of an exec mask (same SGPR-pressure reason as above).
  const int dump = (int)(scr - a) + lane;
  const int up = lower ? 0 : 1;
#pragma unroll
  for (int c = 0; c < S; ++c) {
    const bool keep = lower ? (c <= i) : (do_inverse && c >= i);
    a[keep ? (rowbase + c + up) : dump] = v[c];
  }
  __builtin_amdgcn_wave_barrier();
  return pinned;
}

// (Tried and dropped for the 16-wide sweep, all within 5 % of this one at ~2.5 us per 16 x 16 block: a root-free variant with
// only a reciprocal on the chain and the sixteen inverse square roots as one vector chain at the end; the same with the
// broadcasts through an LDS line instead of v_readlane; and the matrix-pipe variant below.  In-kernel cycle stamps: a column
// costs ~160 cycles + ~13 per remaining row whatever feeds the FMAs, and load + final stores are a quarter of the sweep.)

__device__ __forceinline__ int factor_inverse_32(double* a, int j0, double* scr, bool do_inverse) {
  return factor_inverse<SB, DP>(a, j0 * DP + j0, scr, do_inverse);
}

// The same job on the matrix pipe, without lane broadcasts.  The 32 x 32 block lives in registers in the f64 MFMA C/D layout
// (element (r, c) of a 16 x 16 block: lane c + 16 (r & 3), register r >> 2) as the three upper blocks C00, C01, C11 of the
// symmetric matrix, next to W = [W00 0; W10 W11] which starts as the identity.  One elimination step j divides row j by
// sqrt(pivot) -- that row IS U[j][:] = L[:][j]^T, and in this layout it sits in ONE register of the 16 lanes of group j & 3,
// which is exactly where an A operand (rows, k = j & 3) and a B operand (k = j & 3, columns) of v_mfma_f64_16x16x4 are read
// from.  So the rank-1 trailing update C -= m m^T is one MFMA per block with both operands taken from that register (other
// lane groups masked to zero), and the same row operations applied to W turn the identity into L^-1 (W -= m w_j^T).
// Per step: one v_readlane pair for the pivot, the rsq / Newton chain, a few selects, then 5 MFMAs (pivot in C00) or 3
// (pivot in C11) -- 128 MFMAs per block against ~1100 v_readlane + 1000 FMA-lane-instructions in the broadcast version.
// NEGATIVE RESULT, kept for the record (-DVMC_CHOL_SWEEP_MFMA): 7.3 us per block against 6.7 us for factor_inverse_32.  fp64
// MFMA and fp64 VALU share one datapath on this chip (tools/microbench/dual_issue.hip), so the 128 rank-1 MFMAs (64 cycles
// each, three quarters of their K = 4 unused) cost 8200 cycles by themselves and the pivot chains do not hide behind them.
// Batching four pivots per MFMA would need the four rows of a micro-panel (same register, four lane groups) eliminated
// against each other first, i.e. cross-lane-group moves per pivot -- not pursued.
// Interface as factor_inverse_32: L to the lower triangle of `a`, X = L^-1 transposed into the strict upper part
// (X[r][c] <-> a[c][r+1]); returns the number of pinned pivots.
__device__ __forceinline__ void pivot_step_scalars(double p, double& d, double& inv_d, double& wscale, int& pinned) {
  const bool bad = !(p > 0.0) || !isfinite(p);     // wave-uniform: dead / padded parameter (H7) -> unit row
  pinned += bad ? 1 : 0;
  sqrt_rsqrt(p, d, inv_d);
  d = bad ? 1.0 : d;
  inv_d = bad ? 0.0 : inv_d;                       // pinned: no multipliers below ...
  wscale = bad ? 1.0 : inv_d;                      // ... and a unit diagonal for the inverse
}

__device__ __forceinline__ int factor_inverse_32_mfma(double* a, int j0) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));                   // (see factor_inverse_32: keeps lane-only values out of the caller's loop)
  const int g = lane >> 4, cc = lane & 15;
  double4_t C00, C01, C11, W00, W10, W11;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = g + 4 * q;
    // everything is read from the LOWER triangle (the caller's matrix need not be stored symmetrically -- the border row of
    // the bordered factorisation exists only there --, and only the lower blocks are kept up to date by the trailing updates)
    const int lo = (cc <= r) ? r : cc, hi = (cc <= r) ? cc : r;
    C00[q] = a[(j0 + lo) * DP + j0 + hi];
    C11[q] = a[(j0 + 16 + lo) * DP + j0 + 16 + hi];
    C01[q] = a[(j0 + 16 + cc) * DP + j0 + r];       // A[r][16 + cc] = A[16 + cc][r]
    W00[q] = (r == cc) ? 1.0 : 0.0;
    W11[q] = W00[q];
    W10[q] = 0.0;
  }
  int pinned = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {                   // pivots of the first 16 x 16 block
    const int gj = j & 3, qj = j >> 2;
    const double p = readlane_f64(C00[qj], 16 * gj + j);
    double d, inv_d, wscale;
    pivot_step_scalars(p, d, inv_d, wscale, pinned);
    const bool in = (g == gj);
    const double row0 = (cc == j) ? d : C00[qj] * inv_d;     // U[j][cc], cc >= j (entries left of the diagonal are never stored)
    const double row1 = C01[qj] * inv_d;                      // U[j][16 + cc]
    const double w0 = W00[qj] * wscale;                       // scaled row j of W
    C00[qj] = in ? row0 : C00[qj];
    C01[qj] = in ? row1 : C01[qj];
    W00[qj] = in ? w0 : W00[qj];
    const double m0 = (in && cc > j) ? row0 : 0.0;            // multipliers of the rows below the pivot, first / second block
    const double m1 = in ? row1 : 0.0;
    const double bw = in ? w0 : 0.0;
    const double nm0 = -m0, nm1 = -m1;
    C00 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm0, m0, C00, 0, 0, 0);      // (the next pivot comes out of this one: issue it first)
    C01 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm0, m1, C01, 0, 0, 0);
    C11 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm1, m1, C11, 0, 0, 0);
    W00 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm0, bw, W00, 0, 0, 0);
    W10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm1, bw, W10, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {                   // pivots of the second block
    const int gj = j & 3, qj = j >> 2;
    const double p = readlane_f64(C11[qj], 16 * gj + j);
    double d, inv_d, wscale;
    pivot_step_scalars(p, d, inv_d, wscale, pinned);
    const bool in = (g == gj);
    const double row1 = (cc == j) ? d : C11[qj] * inv_d;
    const double w0 = W10[qj] * wscale, w1 = W11[qj] * wscale;
    C11[qj] = in ? row1 : C11[qj];
    W10[qj] = in ? w0 : W10[qj];
    W11[qj] = in ? w1 : W11[qj];
    const double m1 = (in && cc > j) ? row1 : 0.0;
    const double bw0 = in ? w0 : 0.0, bw1 = in ? w1 : 0.0;
    const double nm1 = -m1;
    C11 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm1, m1, C11, 0, 0, 0);
    W10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm1, bw0, W10, 0, 0, 0);
    W11 = __builtin_amdgcn_mfma_f64_16x16x4f64(nm1, bw1, W11, 0, 0, 0);
  }
  // U (upper triangular, rows r) -> L[c][r] in the lower triangle; W = L^-1 (lower triangular) -> transposed, one column to the right
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = g + 4 * q;
    if (cc >= r) {
      a[(j0 + cc) * DP + j0 + r] = C00[q];
      a[(j0 + 16 + cc) * DP + j0 + 16 + r] = C11[q];
    }
    a[(j0 + 16 + cc) * DP + j0 + r] = C01[q];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = g + 4 * q;
    if (cc <= r) {
      a[(j0 + cc) * DP + j0 + r + 1] = W00[q];
      a[(j0 + 16 + cc) * DP + j0 + 16 + r + 1] = W11[q];
    }
    a[(j0 + cc) * DP + j0 + 16 + r + 1] = W10[q];
  }
  __builtin_amdgcn_wave_barrier();
  return pinned;
}

// Factor the diagonal block A[k0:k0+n, k0:k0+n] = L L^T in LDS (four 32-wide sub-steps), write L back (lower triangle)
// and X = L^-1 (dense lower triangular, zero elsewhere) to Linv.  One workgroup of 256 threads; rows/cols >= n are
// padded with the identity.  LDS: the 128 x 129 block + three 32 x 33 scratch tiles.
__device__ __forceinline__ void chol_diag_body(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                               int* __restrict__ info, int phase /* diagnostic: 5 = everything */) {
  extern __shared__ double a[];          // [NB][DP] followed by four wave-private 32 x 17 scratch tiles and the dump line
  double* T = a + NB * DP;
  double* scr = T + 4 * SB * 17;          // dump line of factor_inverse_32
  const int t = threadIdx.x;
  // 128 x 128 block -> LDS.  Only the ten 32 x 32 blocks on and below the block diagonal are needed (the strict upper part
  // of the LDS image is storage for X, written before it is read); 20 double2 loads per thread, all in flight at once.
  {
    double2_t v[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const int blk = q >> 1;                                     // lower block (bi, bj), 512 double2 each, 2 per thread
      int bi, bj;
      tri_decode(blk, bi, bj);
      const int e = t + 256 * (q & 1), i = SB * bi + (e >> 4), j = SB * bj + (e & 15) * 2;
      v[q] = (double2_t){0.0, 0.0};
      if (i < n && j < n) v[q] = *reinterpret_cast<const double2_t*>(A + (int64_t)(k0 + i) * lda + k0 + j);   // pitch even: 16-B aligned
    }
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const int blk = q >> 1;
      int bi, bj;
      tri_decode(blk, bi, bj);
      const int e = t + 256 * (q & 1), i = SB * bi + (e >> 4), j = SB * bj + (e & 15) * 2;
      a[i * DP + j] = (i < n && j < n) ? v[q].x : ((i == j) ? 1.0 : 0.0);
      a[i * DP + j + 1] = (i < n && j + 1 < n) ? v[q].y : ((i == j + 1) ? 1.0 : 0.0);
    }
  }
  if (t < NB) a[t * DP + NB] = 0.0;
  __syncthreads();

  // Software-pipelined over the four 32-wide sub-steps: once sub-step jb has produced its panel (P3) and updated the next
  // diagonal sub-block (P4a), wave 0 sweeps that sub-block (factor + inverse, a single-wave job) while waves 1..3 finish the
  // trailing update of sub-step jb (P4b) and write the now final block column jb of L back to memory.
  const int lane = t & 63, wave = t >> 6;
  const int crow = lane >> 4, ccol = lane & 15;    // C/D element q of this lane: (crow + 4q, ccol)
  auto trailing_pair = [&](int j0, int pr) {        // A[i][c] -= sum_m L[i][j0+m] L[c][j0+m] on one lower 16x16 block pair
    const int base = j0 + SB;
    int bi, bc;
    tri_decode(pr, bi, bc);
    double4_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = a[(base + 16 * bi + crow + 4 * q) * DP + base + 16 * bc + ccol];
    mfma16(acc, SB, [&](int r, int k) { return -a[(base + 16 * bi + r) * DP + j0 + k]; },
           [&](int c, int k) { return a[(base + 16 * bc + c) * DP + j0 + k]; });
#pragma unroll
    for (int q = 0; q < 4; ++q) a[(base + 16 * bi + crow + 4 * q) * DP + base + 16 * bc + ccol] = acc[q];
  };
  // Off-diagonal 32x32 blocks of X = L^-1, row by row: X_{ib,jb} = -X_{ib,ib} * sum_{mb=jb}^{ib-1} L_{ib,mb} X_{mb,jb}.
  // One item = 16 columns (cb) of one block (ib, jb), done by ONE wave from start to end -- T = sum L X goes through a
  // wave-private 32 x 17 scratch tile, so no workgroup barrier is involved and items can run while wave 0 is inside a sweep.
  // Storage reminder: X[row][col] (row >= col) <-> a[col][row+1]; entries with row < col of a diagonal X block are L's, so
  // they are masked to zero when read.
  double* Tw = T + wave * (SB * 17);
  auto xrow_item = [&](int ib, int jb, int cb) {
    double4_t acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      acc[rb] = (double4_t){0.0, 0.0, 0.0, 0.0};
      for (int mb = jb; mb < ib; ++mb) {
        const bool diag = (mb == jb);
        mfma16(acc[rb], SB, [&](int r, int k) { return a[(SB * ib + 16 * rb + r) * DP + SB * mb + k]; },
               [&](int c, int k) { return (diag && k < 16 * cb + c) ? 0.0 : a[(SB * jb + 16 * cb + c) * DP + SB * mb + k + 1]; });
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) Tw[(16 * rb + crow + 4 * q) * 17 + ccol] = acc[rb][q];
    __builtin_amdgcn_wave_barrier();      // DS operations of one wave execute in order
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      // X_{ib,jb} = -X_{ib,ib} T   (X_ii lower triangular: only k <= row contributes)
      double4_t x = {0.0, 0.0, 0.0, 0.0};
      mfma16(x, 16 * (rb + 1), [&](int r, int k) { return (k <= 16 * rb + r) ? a[(SB * ib + k) * DP + SB * ib + 16 * rb + r + 1] : 0.0; },
             [&](int c, int k) { return Tw[k * 17 + c]; });
#pragma unroll
      for (int q = 0; q < 4; ++q) a[(SB * jb + 16 * cb + ccol) * DP + SB * ib + 16 * rb + crow + 4 * q + 1] = -x[q];
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto sweep = [&](int j0) {
    if (phase >= 1 && t < 64) {
#ifdef VMC_CHOL_SWEEP_MFMA     // measured slower (7.3 us against 6.7 us per block), see factor_inverse_32_mfma
      (void)scr;
      const int pinned = factor_inverse_32_mfma(a, j0);
#else
      const int pinned = factor_inverse_32(a, j0, scr, phase >= 2);
#endif
      if (t == 0 && pinned) atomicAdd(info, pinned);
    }
  };
  sweep(0);
  __syncthreads();
#pragma nounroll
  for (int jb = 0; jb < NB / SB; ++jb) {
    const int j0 = jb * SB;
    const int R = NB - j0 - SB;          // rows (and trailing columns) below this sub-block
    if (R > 0 && phase >= 3) {
      // P3, panel below on the matrix pipe: L[i][j0+c] = sum_{m<=c} A[i][j0+m] X[c][m]; 16-row blocks round-robin over the
      // waves, both 16-column blocks per row block held in registers until every wave has read its operands
      const int nib = R / 16;
      double4_t accp[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ib16 = wave + 4 * s2;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          accp[s2][cb] = (double4_t){0.0, 0.0, 0.0, 0.0};
          if (ib16 < nib)
            mfma16(accp[s2][cb], SB, [&](int r, int k) { return a[(j0 + SB + 16 * ib16 + r) * DP + j0 + k]; },
                   [&](int c, int k) { return (k <= 16 * cb + c) ? a[(j0 + k) * DP + j0 + 16 * cb + c + 1] : 0.0; });   // X[c][k]
        }
      }
      __syncthreads();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ib16 = wave + 4 * s2;
        if (ib16 < nib)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[(j0 + SB + 16 * ib16 + crow + 4 * q) * DP + j0 + 16 * cb + ccol] = accp[s2][cb][q];
      }
      __syncthreads();
      const int npair = nib * (nib + 1) / 2;
      // P4a: the three 16x16 pairs of the next diagonal sub-block, one per wave
      if (phase >= 4 && wave < 3) trailing_pair(j0, wave);
      __syncthreads();
      if (wave == 0) {
        sweep(j0 + SB);
      } else {
        // P4b: the rest of the trailing update, then the write-back of block column jb of L (final since P3)
        if (phase >= 4)
          for (int pr = 3 + (wave - 1); pr < npair; pr += 3) trailing_pair(j0, pr);
        for (int idx = t - 64; idx < (NB - j0) * SB; idx += 192) {
          const int i = j0 + idx / SB, j = j0 + (idx & (SB - 1));
          if (i < n && j <= i) A[(int64_t)(k0 + i) * lda + k0 + j] = a[i * DP + j];
        }
        // row jb of the off-diagonal blocks of X: everything it needs (X_jj, rows < jb of X, block row jb of L) is final
        if (phase >= 5)
          for (int it = wave - 1; it < 2 * jb; it += 3) xrow_item(jb, it >> 1, it & 1);
      }
      __syncthreads();
    } else if (R > 0) {     // (diagnostic phases < 3: keep the sweeps)
      sweep(j0 + SB);
      __syncthreads();
    }
  }

  // write back the last block column of L (the earlier ones went out during the sweeps)
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = NB - SB + idx / SB, j = NB - SB + (idx & (SB - 1));
    if (i < n && j <= i) A[(int64_t)(k0 + i) * lda + k0 + j] = a[i * DP + j];
  }

  // last row of off-diagonal blocks of X (rows 1 and 2 were assembled during the sweeps): six items over the four waves
  if (phase >= 5) {
    for (int it = wave; it < 2 * (NB / SB - 1); it += 4) xrow_item(NB / SB - 1, it >> 1, it & 1);
  }
  __syncthreads();
#pragma unroll 16
  for (int q = 0; q < NB * NB / 256; ++q) {
    const int idx = t + 256 * q, i = idx / NB, c = idx - i * NB;
    const double v = a[c * DP + i + 1];
    Linv[idx] = (i < n && c <= i) ? v : 0.0;
  }
}

// ---- 16-wide schedule of the diagonal block -----------------------------------------------------------------------------------
// Same job as chol_diag_body (factor the 128 x 128 block, write L back, X = L^-1 to Linv), organised in eight 16-wide steps so
// that the sequential part -- the single-wave pivot sweep -- shrinks: a 16 x 16 sweep costs ~130 ns per pivot against ~210 ns
// in a 32 x 32 one (the broadcasts per pivot grow with the block width), and everything else is 16 x 16 x 16 MFMA products:
//   S1 (wave 0)  factor_inverse<16> on the diagonal block of step jb                       -> L_jj, X_jj
//   S2 (all)     panel below:  L(i,jb) = A(i,jb) X_jj^T, i > jb;   row jb of the inverse:  X(jb,c) = X_jj R(jb,c), c < jb
//   S3 (wave 0)  next diagonal block A(jb+1,jb+1) -= L(jb+1,jb) L(jb+1,jb)^T, then straight on to S1 of step jb+1, while
//      (waves 1..3) the rest of the trailing update A(i,c) -= L(i,jb) L(c,jb)^T and the running sums of the inverse
//                   R(i,c) -= L(i,jb) X(jb,c)  (i > jb >= c),  and the write-back of block column jb of L.
// R(i,c) = -sum_{m<i} L(i,m) X(m,c) is the block forward substitution of L X = I in right-looking form; it lives where X(i,c)
// will end up: transposed in the strict upper part, one column to the right (X[r][c] <-> a[c][r+1], as in chol_diag_body).
// LDS: the 128 x 129 block + the dump line of the sweep.
#ifdef VMC_DIAG16_STAMPS      // diagnostic build: wall-clock stamps (100 MHz) of wave 0 at the stage boundaries of the last call
__device__ long long g_diag16_stamps[64];
#define D16_STAMP(i) { if (t == 0) g_diag16_stamps[i] = wall_clock64(); }
#define D16_STAMP1(i) { if (t == 64 && jb == 0) g_diag16_stamps[i] = wall_clock64(); }
#else
#define D16_STAMP(i)
#define D16_STAMP1(i)
#endif

// 16 x 16 x 16 block products of the 16-wide schedule, by KIND (compile-time strides and masks), one or two at a time.
// The LDS image is BLOCKED: an 8 x 8 grid of 16 x 17 slots (pitch 17: the 16 rows x 2 k-columns of a half-wave's ds_read_b64 fall
// on distinct banks, as in mfma_tile.hpp; with the 128 x 129 row-major image of the 32-wide schedule every operand fetch and
// every accumulator access is a 2-way bank conflict, and the shared LDS pipe, not the matrix pipe, limited the three update waves).
//   slot (bi, bj), bi >  bj : block (bi, bj) of A, later of L                        element (r, c) at [r 17 + c]
//   slot (bi, bi)           : lower triangle L_ii; X_ii = L_ii^-1 in the strict upper part, X_ii[r][c] (r >= c) at [c 17 + r + 1]
//   slot (bi, bj), bi <  bj : R(bj, bi), later X(bj, bi) -- block (bj, bi) of the inverse, element (r, c) at [r 17 + c]
// p = {first-operand slot, second-operand slot, result slot}:
//   KIND 0  panel:            C[r][c]  = sum_{k<=c} A[r][k] X_jj[c][k]           (p0 = slot (i,jb), p1 = slot (jb,jb), pc = p0)
//   KIND 1  row of inverse:   C[r][c]  = sum_{k<=r} X_jj[r][k] R[k][c]           (p0 = slot (jb,jb), p1 = pc = slot (cb,jb))
//   KIND 2  trailing update:  C[r][c] -= sum_k L(i,jb)[r][k] L(c',jb)[c][k]
//   KIND 3  inverse sums:     C[r][c] -= sum_k L(i,jb)[r][k] X(jb,cb)[k][c]      (dg: X(jb,jb) = X_jj, zero for k < c)
constexpr int BP = 17, BLK = 16 * BP;
__device__ __forceinline__ constexpr int slot16(int bi, int bj) { return (bi * 8 + bj) * BLK; }
struct P3 { int p0, p1, pc; bool dg; };

template <int KIND>
__device__ __forceinline__ void blk16_operands(const double* __restrict__ a, const P3& d, int lr, int k, double& fa, double& fb) {
  if (KIND == 0) {
    fa = a[d.p0 + lr * BP + k];
    fb = (k <= lr) ? a[d.p1 + k * BP + lr + 1] : 0.0;
  } else if (KIND == 1) {
    fa = (k <= lr) ? a[d.p0 + k * BP + lr + 1] : 0.0;
    fb = a[d.p1 + k * BP + lr];
  } else if (KIND == 2) {
    fa = -a[d.p0 + lr * BP + k];
    fb = a[d.p1 + lr * BP + k];
  } else {
    fa = -a[d.p0 + lr * BP + k];
    fb = d.dg ? ((k < lr) ? 0.0 : a[d.p1 + lr * BP + k + 1]) : a[d.p1 + k * BP + lr];
  }
}

template <int KIND>
__device__ __forceinline__ void blk16_one(double* __restrict__ a, const P3& x) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double4_t ax = {0.0, 0.0, 0.0, 0.0};
  if (KIND >= 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ax[q] = a[x.pc + (lk + 4 * q) * BP + lr];
  }
  double fa[4], fb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) blk16_operands<KIND>(a, x, lr, 4 * s + lk, fa[s], fb[s]);
#pragma unroll
  for (int s = 0; s < 4; ++s) ax = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], ax, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) a[x.pc + (lk + 4 * q) * BP + lr] = ax[q];
}

template <int KIND>
__device__ __forceinline__ void blk16_two(double* __restrict__ a, const P3& x, const P3& y) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double4_t ax = {0.0, 0.0, 0.0, 0.0}, ay = {0.0, 0.0, 0.0, 0.0};
  if (KIND >= 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ax[q] = a[x.pc + (lk + 4 * q) * BP + lr];
      ay[q] = a[y.pc + (lk + 4 * q) * BP + lr];
    }
  }
  double fax[4], fbx[4], fay[4], fby[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    blk16_operands<KIND>(a, x, lr, 4 * s + lk, fax[s], fbx[s]);
    blk16_operands<KIND>(a, y, lr, 4 * s + lk, fay[s], fby[s]);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ax = __builtin_amdgcn_mfma_f64_16x16x4f64(fax[s], fbx[s], ax, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f64_16x16x4f64(fay[s], fby[s], ay, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[x.pc + (lk + 4 * q) * BP + lr] = ax[q];
    a[y.pc + (lk + 4 * q) * BP + lr] = ay[q];
  }
}

// items [first, count) step `stride` of one kind, two at a time
template <int KIND, class MK>
__device__ __forceinline__ void blk16_range(double* __restrict__ a, int first, int count, int stride, MK make) {
  int it = first;
  for (; it + stride < count; it += 2 * stride) blk16_two<KIND>(a, make(it), make(it + stride));
  if (it < count) blk16_one<KIND>(a, make(it));
}

template <int NT>      // threads of the workgroup: 256 or 512 (wave 0 runs the chain, the other NT/64 - 1 waves the updates)
__device__ __forceinline__ void chol_diag16_body(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                                 int* __restrict__ info) {
  extern __shared__ double a[];                    // 64 slots of 16 x 17, then the dump line of the sweep
  constexpr int S16 = 16, NS = NB / S16;          // 8 steps
  double* scr = a + 64 * BLK;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform: the item bookkeeping below runs on the scalar unit
  // waves; update waves -- with eight waves, wave 4 shares SIMD 0 with the chain wave and is kept off the matrix pipe (it only
  // writes finished columns of L back); double2 loads per thread
  constexpr int NW = NT / 64, NU = (NW == 8) ? 6 : NW - 1, NLD = NB * (NB / 2) / NT;
  const int worker = (NW == 8) ? ((wave < 4) ? wave - 1 : wave - 2) : wave - 1;     // -1: the chain wave; (NW == 8) wave 4 -> 2, unused
  const bool updates = (wave != 0) && !(NW == 8 && wave == 4);
  // load the lower triangle (identity beyond n); everything else starts at zero (the R sums, the strict upper part of diagonal slots)
  {
    double2_t v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int e = t + NT * q, i = e >> 6, jp = (e & 63) * 2;
      v[q] = (double2_t){0.0, 0.0};
      if (jp <= i && i < n) v[q] = *reinterpret_cast<const double2_t*>(A + (int64_t)(k0 + i) * lda + k0 + jp);   // jp + 1 <= i + 1 <= n: in range of the padded row
    }
    for (int idx = t; idx < 64 * BLK; idx += NT) a[idx] = 0.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int e = t + NT * q, i = e >> 6, jp = (e & 63) * 2;
      const bool rowok = i < n;
      if (jp <= i) {
        const int at = slot16(i >> 4, jp >> 4) + (i & 15) * BP + (jp & 15);
        a[at] = rowok ? v[q].x : ((jp == i) ? 1.0 : 0.0);
        if (jp + 1 <= i) a[at + 1] = (rowok && jp + 1 < n) ? v[q].y : ((jp + 1 == i) ? 1.0 : 0.0);
      }
    }
  }
  __syncthreads();

  auto sweep = [&](int jb) {
    const int pinned = factor_inverse<S16, BP>(a, slot16(jb, jb), scr, true);
    if (lane == 0 && pinned) atomicAdd(info, pinned);
  };
  D16_STAMP(0)
  if (wave == 0) sweep(0);
  __syncthreads();
  D16_STAMP(1)

#pragma nounroll
  for (int jb = 0; jb < NS; ++jb) {
    const int j0 = jb * S16;
    const int nbel = NS - 1 - jb;                  // 16-row blocks below this step's diagonal block
    const int djj = slot16(jb, jb);
    // ---- S2: nbel panel blocks  L(i,jb) = A(i,jb) X_jj^T  and jb blocks of row jb of the inverse  X(jb,c) = X_jj R(jb,c);
    // every block has one owner, which reads its operands completely before it writes the result in place.  The panel blocks
    // go to the waves first (the chain needs block row jb+1).
    blk16_range<0>(a, wave, nbel, NW, [&](int it) { const int sl = slot16(jb + 1 + it, jb); return P3{sl, djj, sl, false}; });
    blk16_range<1>(a, (wave + NW - nbel % NW) % NW, jb, NW, [&](int cb) { const int sl = slot16(cb, jb); return P3{djj, sl, sl, false}; });
    __syncthreads();
    D16_STAMP(2 + 4 * jb)
    if (wave == 0) {
      if (nbel > 0) {
        // ---- S3a: the next diagonal block  A(jb+1,jb+1) -= L(jb+1,jb) L(jb+1,jb)^T,  then its sweep (the chain)
        blk16_one<2>(a, P3{slot16(jb + 1, jb), slot16(jb + 1, jb), slot16(jb + 1, jb + 1), false});
        __builtin_amdgcn_wave_barrier();
        D16_STAMP(3 + 4 * jb)
        sweep(jb + 1);
        D16_STAMP(4 + 4 * jb)
      }
    } else if (updates) {
      // ---- S3b: rest of the trailing update, lower block pairs (i,c) of the remaining grid except the first, and
      // ---- S3c: R(i,cb) -= L(i,jb) X(jb,cb), i > jb, cb <= jb  (X(jb,jb) = X_jj is lower triangular: k >= c only)
      const int n3b = nbel * (nbel + 1) / 2 - 1, n3c = nbel * (jb + 1);
      D16_STAMP1(50)
      blk16_range<2>(a, worker, n3b, NU, [&](int it) {
        // (bi, bc) of the it+1-th lower pair, it + 1 <= 27: row starts 0, 1, 3, 6, 10, 15, 21 (a compare chain, no sqrt)
        const int pr = it + 1;
        const int bi = (pr >= 1) + (pr >= 3) + (pr >= 6) + (pr >= 10) + (pr >= 15) + (pr >= 21);
        const int bc = pr - bi * (bi + 1) / 2;
        return P3{slot16(jb + 1 + bi, jb), slot16(jb + 1 + bc, jb), slot16(jb + 1 + bi, jb + 1 + bc), false};
      });
      D16_STAMP1(51)
      blk16_range<3>(a, (worker + NU - n3b % NU) % NU, n3c, NU, [&](int u) {
        const int ii = u / (jb + 1), cb = u - ii * (jb + 1), i = jb + 1 + ii;
        return P3{slot16(i, jb), (cb == jb) ? djj : slot16(cb, jb), slot16(cb, i), cb == jb};
      });
      D16_STAMP1(52)
    }
    // block column jb of L is final: write it back (rows j0.., 16 columns) -- every wave but the chain wave; with eight waves the
    // upper four, so that the wave that stays off the matrix pipe carries its share from the start
    if (NW == 8 ? (wave >= 4) : (wave != 0)) {
      const int first = (NW == 8) ? t - 256 : t - 64, step = (NW == 8) ? 256 : NT - 64;
      for (int idx = first; idx < (NB - j0) * S16; idx += step) {
        const int i = j0 + idx / S16, jj = idx & (S16 - 1);
        if (i < n && j0 + jj <= i) A[(int64_t)(k0 + i) * lda + k0 + j0 + jj] = a[slot16(i >> 4, jb) + (i & 15) * BP + jj];
      }
    }
    D16_STAMP1(53)
    __syncthreads();
    D16_STAMP(5 + 4 * jb)
  }
  D16_STAMP(40)
  // X = L^-1: dense lower triangular, zero elsewhere
#pragma unroll 16
  for (int q = 0; q < NB * NB / NT; ++q) {
    const int idx = t + NT * q, i = idx / NB, c = idx - i * NB;
    const int ib = i >> 4, cb = c >> 4, ri = i & 15, ci = c & 15;
    double v = 0.0;
    if (i < n && c <= i) v = (ib == cb) ? a[slot16(ib, ib) + ci * BP + ri + 1] : a[slot16(cb, ib) + ri * BP + ci];
    Linv[idx] = v;
  }
  D16_STAMP(41)
}

// Latency-optimised strip kernel for the two short dependent GEMMs of every panel (K = 128 exactly):
//   mode 0 (panel solve):        A[row0.., k0:k0+128] <- A[row0.., k0:k0+128] * Linv^T          (B = Linv, 128 x 128)
//   mode 1 (next-panel update):  A[row0.., c0:c0+128] -= A[row0.., k0:k0+128] * A[c0.., k0:k0+128]^T   (c0 = k0+128)
// One workgroup = a 32-row strip x all 128 columns (4 waves, 32 x 32 each), so a block row is spread over four CUs and
// the whole K range is in flight at once: every global load of both K-halves is issued before the first wait (one
// workgroup per CU, up to 512 VGPRs), then two LDS stage / MFMA rounds.  The 128-wide tile engine needs ~40 us for
// the same work on these small grids (one K-step of prefetch, 13.7 us of MFMA per 128^3 tile on a single CU).
constexpr int SK = 64;            // K half
constexpr int SP = SK + 2;        // LDS pitch: 4r + 2k dwords mod 64 distinct over a half-wave of ds_read_b64
// SR = rows per strip workgroup: 16 while 8 strips per block row still fit one round of the 256 CUs (half the MFMA chain per
// workgroup: 2.88 -> 2.67 ms for the whole factorisation at P = 4213), 32 beyond that (each workgroup re-reads the 128 KB B
// operand, so twice the workgroups is a loss once they no longer run concurrently).
// General form (two-level schedule, see vmcpde_cholesky): the strips cover the rows from c0 downwards; the contraction runs over
// the columns kc0 .. kc0 + 128 * nchunk (nchunk = 1 in mode 0, up to the super-panel width in mode 1), 128 at a time; mode 0
// overwrites the columns kc0.., mode 1 updates the block column c0.. (whose rows c0..c0+127 are also the B operand).
template <int SR>
__device__ __forceinline__ void strip_body(double* __restrict__ A, int64_t lda, int P, int kc0, int nchunk, int c0, int mode,
                                           const double* __restrict__ Linv, int strip, double* __restrict__ lds /* Bs[128][SP] then As[SR][SP] */) {
  double* Bs = lds;
  double* As = lds + NB * SP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = c0 + strip * SR;                      // first row of this strip (a row below the panel)
  constexpr int NA = SR / 8, NI = SR / 16;
  double4_t acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int lr = lane & 15, lk = lane >> 4;
  for (int ch = 0; ch < nchunk; ++ch) {
  const int k0 = kc0 + ch * NB;
  if (ch) __syncthreads();                               // the previous chunk's fragment reads are done
  const double* __restrict__ Bsrc = mode ? (A + (int64_t)c0 * lda + k0) : Linv;
  const int64_t ldb = mode ? lda : NB;
  const double* __restrict__ Asrc = A + (int64_t)row0 * lda + k0;
  // ---- issue every load: per K-half 16 double2 of B (128 rows x 32 double2) and SR/8 of A (SR rows x 32 double2) per thread
  double2_t vb[2][16], va[2][NA];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = q * 256 + t, r = idx >> 5, kp = (idx & 31) * 2;
      vb[h][q] = *reinterpret_cast<const double2_t*>(Bsrc + (int64_t)r * ldb + h * SK + kp);
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = q * 256 + t, r = idx >> 5, kp = (idx & 31) * 2;
      va[h][q] = *reinterpret_cast<const double2_t*>(Asrc + (int64_t)r * lda + h * SK + kp);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = q * 256 + t, r = idx >> 5, kp = (idx & 31) * 2;
      *reinterpret_cast<double2_t*>(Bs + r * SP + kp) = vb[h][q];
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = q * 256 + t, r = idx >> 5, kp = (idx & 31) * 2;
      *reinterpret_cast<double2_t*>(As + r * SP + kp) = va[h][q];
    }
    __syncthreads();
    const double* __restrict__ ap = As + lr * SP + lk;
    const double* __restrict__ bp = Bs + (32 * wave + lr) * SP + lk;
#pragma unroll 4
    for (int k4 = 0; k4 < SK / 4; ++k4) {
      const double b0 = bp[4 * k4], b1 = bp[16 * SP + 4 * k4];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const double ai = ap[16 * i * SP + 4 * k4];
        acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b1, acc[i][1], 0, 0, 0);
      }
    }
  }
  }   // chunks
  // ---- epilogue: wave w owns columns 32w..32w+31 of the strip
  double* __restrict__ C = A + (int64_t)row0 * lda + (mode ? c0 : kc0);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + (lane >> 4) + 4 * r;
        const int col = 32 * wave + 16 * j + (lane & 15);
        if (row0 + row < P && (mode == 0 || c0 + col < P)) {
          double* p = C + (int64_t)row * lda + col;
          *p = mode ? (*p - acc[i][j][r]) : acc[i][j][r];
        }
      }
}

template <int SR>
__global__ __launch_bounds__(256, 1) void chol_strip_kernel(double* __restrict__ A, int64_t lda, int P, int kc0, int nchunk, int c0, int mode,
                                                            const double* __restrict__ Linv) {
  extern __shared__ double lds[];
  strip_body<SR>(A, lda, P, kc0, nchunk, c0, mode, Linv, (int)blockIdx.x, lds);
}

// Trailing update A22 -= L21 L21^T on the lower-triangle tiles (syrk, K = n): see chol_syrk_diag_kernel.
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                                       int* __restrict__ info, int phase) {
  chol_diag_body(A, lda, k0, n, Linv, info, phase);
}

// the 16-wide schedule: 512 threads -- wave 0 runs the chain, seven waves share the 16 x 16 x 16 updates of a step
constexpr int D16_THREADS = 512;
__global__ __launch_bounds__(D16_THREADS) void chol_diag16_kernel(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                                                  int* __restrict__ info) {
  chol_diag16_body<D16_THREADS>(A, lda, k0, n, Linv, info);
}

// One 128 x 128 tile (ti >= tj, block indices inside the trailing matrix) of A22 -= L21 L21^T.
__device__ __forceinline__ void syrk_tile_at(double* __restrict__ A, int64_t lda, int P, int k0, int n, int ti, int tj) {
  extern __shared__ double lds[];
  const int base = k0 + n;
  const int rowA0 = base + ti * TILE, rowB0 = base + tj * TILE;
  const int rowsA = (P - rowA0 < TILE) ? (P - rowA0) : TILE;
  const int rowsB = (P - rowB0 < TILE) ? (P - rowB0) : TILE;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop_dma<16>(acc, A + k0, lda, rowA0, A + k0, lda, rowB0, ti == tj, 0, n, lds);
  tile_store_global(acc, A + (int64_t)rowA0 * lda + rowB0, lda, rowsA, rowsB, 1);
}

// One 128 x 128 tile (block row bi >= block column bj of the whole matrix) of  A[bi][bj] -= A[bi][kc0 : kc0+K] A[bj][kc0 : kc0+K]^T,
// by the 256 threads `tid` of a workgroup with their own 64 KB of LDS (two such tiles per 512-thread workgroup).
__device__ __forceinline__ void syrk_tile_abs(double* __restrict__ A, int64_t lda, int P, int kc0, int K, int bi, int bj,
                                              double* __restrict__ lds, int tid) {
  const int rowA0 = bi * TILE, rowB0 = bj * TILE;
  const int rowsA = (P - rowA0 < TILE) ? (P - rowA0) : TILE;
  const int rowsB = (P - rowB0 < TILE) ? (P - rowB0) : TILE;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop_dma<16>(acc, A + kc0, lda, rowA0, A + kc0, lda, rowB0, bi == bj, 0, K, lds, tid);
  tile_store_global(acc, A + (int64_t)rowA0 * lda + rowB0, lda, rowsA, rowsB, 1, tid);
}

// (ti, tj >= 1: in the multi-launch schedule the first block column belongs to chol_strip_kernel mode 1)
__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int64_t lda, int P, int k0, int n, int tile) {
  int ti, tj;
  tri_decode(tile, ti, tj);
  syrk_tile_at(A, lda, P, k0, n, ti + 1, tj + 1);
}

// ---- one launch per panel ---------------------------------------------------------------------------------------------------
// The chain of a panel is  panel solve (strips, mode 0)  ->  update of the next diagonal tile (mode 1 of the strips of the first
// block row below)  ->  factorisation of that tile  ||  rest of the trailing update.  As three launches it costs 9.5 + 9.5 + 57 us
// per panel, a good part of it launch and drain time; here the pieces are workgroups of ONE launch chained by counters in
// global memory (opt-in, VMCPDE_CHOL_PANEL=1: measured slower than the three launches, see vmcpde_cholesky):
//   blocks [0, S)        strips: mode 0, then count[block row] += 1; the strips of block row 0 wait for count[0] == spb (all of
//                        L21's first block row is there), run mode 1 on their rows of tile (0,0), then count[nrem] += 1
//   block  S             waits for count[nrem] == spb, then factors the next diagonal block
//   blocks (S, S+1+T)    trailing tiles (ti, tj) != (0,0): wait for count[ti] == count[tj] == spb, then the syrk tile
// Workgroups are dispatched in block order and only ever wait for strips, which come first and fit the chip together
// (S + 1 <= 256 workgroups at one per CU, checked by the host), so every wait ends; waits are bounded in time all the same
// (a timeout flags `info` and lets the grid drain).  Release / acquire at agent scope: the L2 of each XCD is written back /
// invalidated by the fences, as for any cross-workgroup hand-over on this chip.
constexpr long long SPIN_TIMEOUT_TICKS = 20LL * 1000 * 1000;     // wall_clock64 runs at 100 MHz: 0.2 s

__device__ __forceinline__ void signal_count(unsigned int* p) {
  __syncthreads();                               // every wave's stores are issued and complete (vmcnt(0) inside)
  if (threadIdx.x == 0) {
    __threadfence();                             // release
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void wait_count(const unsigned int* p, unsigned int target, int* __restrict__ info) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > SPIN_TIMEOUT_TICKS) {
        atomicOr(info, 0x40000000);              // the host's pinned-pivot / residual guard turns this into an error
        break;
      }
    }
    __threadfence();                             // acquire: invalidates this CU's L1 and the stale lines of its XCD's L2 -- cache
  }                                              // operations, so one wave does it for the workgroup before the others load
  __syncthreads();
}

template <int SR>
__global__ __launch_bounds__(256, 1) void chol_panel_kernel(double* __restrict__ A, int64_t lda, int P, int k0,
                                                            const double* __restrict__ Linv_k, double* __restrict__ Linv_next,
                                                            int* __restrict__ info, unsigned int* __restrict__ count, int nrem,
                                                            int phase) {
  extern __shared__ double lds[];
  constexpr int SPB = NB / SR;                   // strips per block row
  const int S = nrem * SPB;
  const int b = (int)blockIdx.x;
  if (b < S) {
    const int br = b / SPB;
    strip_body<SR>(A, lda, P, k0, 1, k0 + NB, 0, Linv_k, b, lds);
    signal_count(count + br);
    if (br == 0) {
      wait_count(count, SPB, info);
      strip_body<SR>(A, lda, P, k0, 1, k0 + NB, 1, Linv_k, b, lds);        // its rows of the next diagonal tile (B = L21's first block row)
      signal_count(count + nrem);
    }
    return;
  }
  if (b == S) {
    wait_count(count + nrem, SPB, info);
    const int k1 = k0 + NB;
    chol_diag_body(A, lda, k1, (P - k1 < NB) ? (P - k1) : NB, Linv_next, info, phase);
    return;
  }
  int ti, tj;
  tri_decode(b - S, ti, tj);                     // b - S >= 1: every lower tile except (0,0)
  wait_count(count + ti, SPB, info);
  if (tj != ti) wait_count(count + tj, SPB, info);
  syrk_tile_at(A, lda, P, k0, NB, ti, tj);
}

// Lookahead, small trailing matrices: one launch holds the trailing update of panel k AND, as workgroup 0, the
// factorisation of the next diagonal block, which depends only on the first block column (already updated by
// chol_strip_kernel mode 1).  The ~75 us single-workgroup critical path then runs next to the update instead of after it,
// with no cross-stream event (about 10 us each way).  Every workgroup carries the diagonal block's 140 KB of LDS, i.e.
// one workgroup per CU -- fine while the update is a few rounds of tiles, too slow for large ones (see vmcpde_cholesky).
__global__ __launch_bounds__(256) void chol_syrk_diag_kernel(double* __restrict__ A, int64_t lda, int P, int k0, int n,
                                                            double* __restrict__ Linv_next, int* __restrict__ info, int phase) {
  if (blockIdx.x == 0) {
    const int k1 = k0 + n;
    chol_diag_body(A, lda, k1, (P - k1 < NB) ? (P - k1) : NB, Linv_next, info, phase);
    return;
  }
  syrk_tile(A, lda, P, k0, n, blockIdx.x - 1);
}

// The update launch of the two-level schedule (vmcpde_cholesky): workgroup 0 factors the next diagonal block (first row k1), every
// other workgroup of 512 threads carries TWO tiles (waves 0-3 and 4-7, 64 KB of LDS each, all tiles of a launch run the same K so
// the shared barriers match) -- the occupancy of two 256-thread workgroups per CU under the diagonal block's 140 KB LDS size, so
// the lookahead rides in the launch for every size.  tpw = tiles per workgroup: 1 while the tiles fit one round of the CUs that
// way (a K = 128 tile keeps the matrix pipes of a CU busy for 15 us; two on one CU take 30 us next to the 35 us diagonal block).  Tiles: block columns [jlo, jhi) of the lower triangle, rows j .. nbk-1 of
// column j, updated with the operand columns kc0 .. kc0+K.
__global__ __launch_bounds__(D16_THREADS) void chol_update_diag16_kernel(double* __restrict__ A, int64_t lda, int P, int kc0, int K, int jlo,
                                                                         int jhi, int nbk, int ntiles, int tpw, int k1,
                                                                         double* __restrict__ Linv_next, int* __restrict__ info) {
  if (blockIdx.x == 0) {
    if (k1 >= 0) chol_diag16_body<D16_THREADS>(A, lda, k1, (P - k1 < NB) ? (P - k1) : NB, Linv_next, info);     // (k1 < 0: a bulk launch without the block)
    return;
  }
  extern __shared__ double lds[];
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  int idx = tpw * ((int)blockIdx.x - 1) + half;
  if (half >= tpw || idx >= ntiles) return;      // (waves that have ended do not take part in the barriers)
  int bi, bj;
  if (jhi == nbk) {
    int ti, tj;
    tri_decode(idx, ti, tj);
    bi = jlo + ti;
    bj = jlo + tj;
  } else {
    bj = jlo;
    while (idx >= nbk - bj) {
      idx -= nbk - bj;
      ++bj;
    }
    bi = bj + idx;
  }
  syrk_tile_abs(A, lda, P, kc0, K, bi, bj, lds + half * (int)(TILE_DMA_LDS_BYTES / sizeof(double)), (int)threadIdx.x & 255);
}

// Large trailing matrices: the update alone, two workgroups per CU; the next diagonal block is factored on a side stream.
__global__ __launch_bounds__(256, 2) void chol_syrk_kernel(double* __restrict__ A, int64_t lda, int P, int k0, int n) {
  syrk_tile(A, lda, P, k0, n, blockIdx.x);
}

// y_blk = Linv_blk * r_blk (trans=0) or Linv_blk^T * r_blk (trans=1), in place in r (n <= 128 valid entries).
// One workgroup of 256 threads; every load of the 128 x 128 block is coalesced and several rows are in flight per wave.
__device__ __forceinline__ void diag_apply(const double* __restrict__ Linv, double* __restrict__ r, int n, int trans,
                                           double* __restrict__ sh /*[NB]*/, double* __restrict__ part /*[2*NB]*/) {
  const int t = threadIdx.x;
  if (t < NB) sh[t] = (t < n) ? r[t] : 0.0;
  __syncthreads();
  if (!trans) {
    const int lane = t & 63, wave = t >> 6;
    const double x0 = sh[2 * lane], x1 = sh[2 * lane + 1];
    // all 32 rows of this wave in flight at once (one memory latency instead of four)
    double2_t v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = *reinterpret_cast<const double2_t*>(Linv + (wave * 32 + u) * NB + 2 * lane);
    double mine = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const double su = wave_sum(v[u].x * x0 + v[u].y * x1);     // Linv is exactly zero above the diagonal and in padded rows
      if (lane == u) mine = su;
    }
    if (lane < 32 && wave * 32 + lane < n) r[wave * 32 + lane] = mine;
  } else {
    const int j = t & (NB - 1), half = t >> 7;
    double s = 0.0;
#pragma unroll 8
    for (int m = half; m < NB; m += 2) s = fma(Linv[m * NB + j], sh[m], s);
    part[half * NB + j] = s;
    __syncthreads();
    if (t < n) r[t] = part[t] + part[NB + t];
  }
  __syncthreads();
}

// Forward substitution step k: every row block b > k does r_b -= L[b][k] y_k; block k+1 then finishes y_{k+1}.
__global__ __launch_bounds__(256) void solve_fwd_step(const double* __restrict__ L, int64_t lda, int P, int k, double* __restrict__ r,
                                                     const double* __restrict__ Linv_all) {
  __shared__ double yk[NB];
  __shared__ double sh[NB];
  __shared__ double part[2 * NB];
  const int t = threadIdx.x;
  const int k0 = k * NB;                      // block k is always full here (a later block exists)
  const int b = k + 1 + blockIdx.x;
  const int b0 = b * NB;
  const int nbk = (P - b0 < NB) ? (P - b0) : NB;
  if (t < NB) yk[t] = r[k0 + t];
  __syncthreads();
  // 4 waves x 32 rows; a row is 64 lanes x 2 columns; rows beyond P are zero-padded storage
  const int lane = t & 63, wave = t >> 6;
  const double y0 = yk[2 * lane], y1 = yk[2 * lane + 1];
  {
    double2_t v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = *reinterpret_cast<const double2_t*>(L + (int64_t)(b0 + wave * 32 + u) * lda + k0 + 2 * lane);
    const double rold = (lane < 32 && wave * 32 + lane < nbk) ? r[b0 + wave * 32 + lane] : 0.0;
    double mine = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const double su = wave_sum(v[u].x * y0 + v[u].y * y1);
      if (lane == u) mine = su;
    }
    if (lane < 32 && wave * 32 + lane < nbk) r[b0 + wave * 32 + lane] = rold - mine;
  }
  __syncthreads();
  if (blockIdx.x == 0) diag_apply(Linv_all + (int64_t)b * NB * NB, r + b0, nbk, 0, sh, part);
}

// Backward substitution step k: every row block b < k does r_b -= L[k][b]^T u_k; block k-1 then finishes u_{k-1}.
__global__ __launch_bounds__(256) void solve_bwd_step(const double* __restrict__ L, int64_t lda, int P, int k, double* __restrict__ r,
                                                     const double* __restrict__ Linv_all) {
  __shared__ double uk[NB];
  __shared__ double part[2 * NB];
  __shared__ double sh[NB];
  const int t = threadIdx.x;
  const int k0 = k * NB;
  const int nk = (P - k0 < NB) ? (P - k0) : NB;
  const int b = blockIdx.x;          // 0 .. k-1
  const int b0 = b * NB;
  if (t < NB) uk[t] = (t < nk) ? r[k0 + t] : 0.0;
  __syncthreads();
  const int j = t & (NB - 1), half = t >> 7;   // column j of the block, two halves of the row range
  double s = 0.0;
#pragma unroll 8
  for (int i = half; i < NB; i += 2) s = fma(L[(int64_t)(k0 + i) * lda + b0 + j], uk[i], s);   // rows >= P are zero padding
  part[half * NB + j] = s;
  __syncthreads();
  if (t < NB) r[b0 + t] -= part[t] + part[NB + t];
  __syncthreads();
  if (b == k - 1) diag_apply(Linv_all + (int64_t)b * NB * NB, r + b0, NB, 1, sh, part);
}

// The whole back substitution L^T u = y as ONE launch: workgroup b owns block b of the solution, applies the updates
// r_b -= L[k][b]^T u_k for k = nb-1 .. b+1 as the u_k are published (a counter per block, release / acquire at agent scope),
// then u_b = Linv_b^T r_b and publishes it.  The 128 x 128 tile of every update and the inverse diagonal block are fetched
// into registers BEFORE the wait, so one link of the chain is: see the flag, read 1 KB of u_k, 64 FMAs per thread, a reduction
// through LDS, publish -- against a launch boundary (~9 us) per block in the one-launch-per-block version.
// All nb workgroups are resident together (nb <= 256 checked by the host), block nb-1 waits for nothing, so every wait ends;
// waits are bounded in time anyway and poison the result with NaN when they give up.
__global__ __launch_bounds__(256, 1) void solve_bwd_persistent(const double* __restrict__ L, int64_t lda, int P, int nb,
                                                               const double* __restrict__ y, double* __restrict__ u,
                                                               const double* __restrict__ Linv_all, unsigned int* __restrict__ flags) {
  __shared__ double uk[NB], rb[NB], part[2 * NB];
  __shared__ int poisoned;
  const int t = threadIdx.x;
  const int b = nb - 1 - (int)blockIdx.x;       // the head of the chain is dispatched first
  const int b0 = b * NB;
  const int nbk = (P - b0 < NB) ? (P - b0) : NB;
  const int j = t & (NB - 1), half = t >> 7;    // column j of a tile, two halves of its row range
  if (t == 0) poisoned = 0;
  if (t < NB) rb[t] = (t < nbk) ? y[b0 + t] : 0.0;
  double w[NB / 2];                              // this thread's half column of Linv_b (rows m = half, half + 2, ...)
  const double* __restrict__ Lb = Linv_all + (int64_t)b * NB * NB;
#pragma unroll
  for (int q = 0; q < NB / 2; ++q) w[q] = Lb[(half + 2 * q) * NB + j];
  for (int k = nb - 1; k > b; --k) {
    const int k0 = k * NB;
    const int nk = (P - k0 < NB) ? (P - k0) : NB;
    double v[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) v[q] = L[(int64_t)(k0 + half + 2 * q) * lda + b0 + j];     // rows >= P: zero padding / masked by uk
    if (t == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > SPIN_TIMEOUT_TICKS) { poisoned = 1; break; }
      }
      __threadfence();                           // acquire (cache operations: one wave does it for the workgroup)
    }
    __syncthreads();
    if (t < NB) uk[t] = (t < nk) ? u[k0 + t] : 0.0;
    __syncthreads();
    double s0 = 0.0;
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) s0 = fma(v[q], uk[half + 2 * q], s0);
    part[half * NB + j] = s0;
    __syncthreads();
    if (t < NB) rb[t] -= part[t] + part[NB + t];
    __syncthreads();
  }
  // u_b = Linv_b^T r_b
  double s1 = 0.0;
#pragma unroll
  for (int q = 0; q < NB / 2; ++q) s1 = fma(w[q], rb[half + 2 * q], s1);
  __syncthreads();
  part[half * NB + j] = s1;
  __syncthreads();
  if (t < nbk) u[b0 + t] = poisoned ? __builtin_nan("") : (part[t] + part[NB + t]);
  __syncthreads();
  if (t == 0) {
    __threadfence();                             // release
    __hip_atomic_fetch_add(flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Both substitutions L y = b, L^T u = y of a SMALL system (nb <= 4 blocks, P <= 512: the r x r matrix G of the Woodbury application) as ONE workgroup:
// the right-hand side lives in LDS, the block updates and the diagonal-block inverses are the arithmetic of solve_fwd_step / solve_bwd_step /
// diag_apply in the same order -- the same bits -- without the six launch boundaries of the block-row chain (round 6).
constexpr int SOLVE_SMALL_MAX_BLOCKS = 4;
__global__ __launch_bounds__(256) void solve_small_kernel(const double* __restrict__ L, int64_t lda, int P, int nb, const double* __restrict__ rhs,
                                                         double* __restrict__ u, const double* __restrict__ Linv_all) {
  __shared__ double rs[SOLVE_SMALL_MAX_BLOCKS * NB];
  __shared__ double sh[NB];
  __shared__ double part[2 * NB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < nb * NB; i += 256) rs[i] = (i < P) ? rhs[i] : 0.0;
  __syncthreads();
  auto nrows = [&](int b) { return (P - b * NB < NB) ? (P - b * NB) : NB; };
  // ---- L y = b
  diag_apply(Linv_all, rs, nrows(0), 0, sh, part);
  for (int k = 0; k + 1 < nb; ++k) {
    const int k0 = k * NB;
    const double y0 = rs[k0 + 2 * lane], y1 = rs[k0 + 2 * lane + 1];
    for (int b = k + 1; b < nb; ++b) {
      const int b0 = b * NB, nbk = nrows(b);
      double2_t v[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) v[q] = *reinterpret_cast<const double2_t*>(L + (int64_t)(b0 + wave * 32 + q) * lda + k0 + 2 * lane);
      const double rold = (lane < 32 && wave * 32 + lane < nbk) ? rs[b0 + wave * 32 + lane] : 0.0;
      double mine = 0.0;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const double su = wave_sum(v[q].x * y0 + v[q].y * y1);
        if (lane == q) mine = su;
      }
      if (lane < 32 && wave * 32 + lane < nbk) rs[b0 + wave * 32 + lane] = rold - mine;
    }
    __syncthreads();
    diag_apply(Linv_all + (int64_t)(k + 1) * NB * NB, rs + (k + 1) * NB, nrows(k + 1), 0, sh, part);
  }
  // ---- L^T u = y
  diag_apply(Linv_all + (int64_t)(nb - 1) * NB * NB, rs + (nb - 1) * NB, nrows(nb - 1), 1, sh, part);
  const int j = t & (NB - 1), half = t >> 7;
  for (int k = nb - 1; k >= 1; --k) {
    const int k0 = k * NB;
    for (int b = 0; b < k; ++b) {
      const int b0 = b * NB;
      double sacc = 0.0;
#pragma unroll 8
      for (int i = half; i < NB; i += 2) sacc = fma(L[(int64_t)(k0 + i) * lda + b0 + j], rs[k0 + i], sacc);   // rows >= P: zero padding, rs zero there
      part[half * NB + j] = sacc;
      __syncthreads();
      if (t < NB) rs[b0 + t] -= part[t] + part[NB + t];
      __syncthreads();
    }
    diag_apply(Linv_all + (int64_t)(k - 1) * NB * NB, rs + (k - 1) * NB, NB, 1, sh, part);
  }
  for (int i = t; i < P; i += 256) u[i] = rs[i];
}

__global__ __launch_bounds__(256) void solve_diag_only(double* __restrict__ r, int P, int k, int trans, const double* __restrict__ Linv_all) {
  __shared__ double sh[NB];
  __shared__ double part[2 * NB];
  const int k0 = k * NB;
  const int nk = (P - k0 < NB) ? (P - k0) : NB;
  diag_apply(Linv_all + (int64_t)k * NB * NB, r + k0, nk, trans, sh, part);
}

// y = A x (trans=0, one wave per row) ; y = A^T x (trans=1, one thread per column, rows split over blockIdx.y + atomics-free
// two-pass is not needed: the grid walks all rows).
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, int64_t lda, int rows, int cols,
                                                    const double* __restrict__ x, double* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const double* __restrict__ a = A + (int64_t)row * lda;
  double s = 0.0;
  for (int c = lane; c < cols; c += 64) s += a[c] * x[c];
  s = wave_sum(s);
  if (lane == 0) y[row] = s;
}

// y = A^T x: 32 columns per workgroup (256-byte row segments), 32 row slices of 32 lanes each -- with 64 columns and 4 slices per
// workgroup a 4213 x 4213 matrix gave 66 workgroups of 1053 dependent iterations: 375 us for 142 MB (V^T F of the spectral solve)
__global__ __launch_bounds__(1024) void gemv_t_kernel(const double* __restrict__ A, int64_t lda, int rows, int cols,
                                                     const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double part[32][33];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  double s = 0.0;
  if (col < cols) {
#pragma unroll 4
    for (int r = q; r < rows; r += 32) s = fma(A[(int64_t)r * lda + col], x[r], s);
  }
  part[q][c] = s;
  __syncthreads();
  if (q == 0 && col < cols) {
    double total = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) total += part[k][c];
    y[col] = total;
  }
}

// tdvp.py:82-89
__global__ void pinv_coeffs_kernel(const double* __restrict__ ev, const double* __restrict__ VtF, const double* __restrict__ snr, int P,
                                   double svdTol, double snrTol, double* __restrict__ coef) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const double rel = fabs(ev[k] / ev[P - 1]);
  const double invEv = (rel > 1e-14) ? 1.0 / ev[k] : 0.0;            // :82
  const double q = svdTol / rel;
  double reg = 1.0 / (1.0 + q * q * q * q * q * q);                  // :85
  if (snr) {
    const double qs = snrTol / snr[k];
    reg *= 1.0 / (1.0 + qs * qs * qs * qs * qs * qs);                // :89
  }
  // :91.  A mode dropped by the hard cut-off contributes an exact zero whatever its other factors are: the rank-revealing engine
  // (eig.hip) reports null-space eigenvalues as exact zeros, for which svdTol = 0 or an SNR of 0 / 0 would make `reg` a NaN.
  coef[k] = (invEv == 0.0) ? 0.0 : invEv * reg * VtF[k];
}

}  // namespace

// workspace: nb inverse diagonal blocks (128 x 128 doubles each), then the per-panel counters of the one-launch-per-panel schedule
static int64_t chol_counter_bytes(int nb) { return (((int64_t)nb * (nb + 2) * (int64_t)sizeof(unsigned int)) + 255) / 256 * 256; }

#ifdef VMC_DIAG16_STAMPS
extern "C" int vmcpde_debug_read_diag16_stamps(long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_diag16_stamps), sizeof(long long) * 64);
}

#endif

extern "C" int64_t vmcpde_chol_workspace_bytes(int32_t P) {
  if (P <= 0) return 0;
  const int nb = (P + NB - 1) / NB;
  return (int64_t)nb * NB * NB * (int64_t)sizeof(double) + chol_counter_bytes(nb);
}

extern "C" int vmcpde_cholesky(double* A, int64_t lda, int32_t P, int32_t* info, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  VMC_REQUIRE(A && info && workspace, "vmcpde_cholesky: null argument");
  VMC_REQUIRE(P >= 1 && lda >= P && (lda % 2) == 0 && lda < (1 << 22), "lda must be even, >= P and < 2^22");
  VMC_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0, "A must be 16-byte aligned");
  VMC_REQUIRE(workspace_bytes >= vmcpde_chol_workspace_bytes(P), "cholesky workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* Linv = static_cast<double*>(workspace);
  const int nb = (P + NB - 1) / NB;
  const size_t diag_lds = ((size_t)NB * DP + 4 * SB * 17 + 2 * SB) * sizeof(double);   // block + 4 wave-private tiles + 64-slot dump line
  static_assert(((size_t)NB * DP + 4 * SB * 17 + 2 * SB) * sizeof(double) >= TILE_DMA_LDS_BYTES, "the fused launch sizes its LDS for the diagonal block");
  const size_t strip_lds = (size_t)(NB + 32) * SP * sizeof(double);
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  {
    DeviceCtxLock lock;
    if (!ctx->attr_chol) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_strip_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_strip_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_syrk_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_syrk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_DMA_LDS_BYTES));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_panel_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_panel_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_diag16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_update_diag16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      ctx->attr_chol = true;
    }
    if (ctx->side == nullptr) {
      VMC_CHECK_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
      VMC_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      VMC_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
  }
  VMC_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), st));
  // Lookahead: the diagonal block of panel k+1 only depends on the first block column of panel k's trailing update, so it is
  // factored next to the rest of that update.  Default: the two-level schedule below (16-wide diagonal block, every update launch
  // carries the block as workgroup 0).  The round-1 schedule around the 32-wide diagonal block is kept for A/B (VMCPDE_CHOL_DIAG16=0):
  //  * up to FUSED_MAX_TILES tiles in the rest of that update: workgroup 0 of the same launch factors it
  //    (chol_syrk_diag_kernel) -- the update is short, the single-workgroup factorisation is the critical path;
  //  * more: the update is the critical path; it runs at two workgroups per CU on the caller's stream while the block is
  //    factored on this device's side stream (fork/join with events anchored on the caller's stream, so that from outside
  //    all work is ordered in `stream` as usual).
#ifdef VMC_DIAG_PHASE_KNOB
  const int diag_phase = env_int("VMCPDE_DIAG_PHASE", 5);   // timing diagnostics only (debug builds): < 5 gives wrong factors
#else
  const int diag_phase = 5;
#endif
  const int fused_max = env_int("VMCPDE_CHOL_FUSED_MAX", FUSED_MAX_TILES);
  // 1: one launch per panel (chol_panel_kernel, with the 32-wide diagonal block).  Off by default -- NEGATIVE RESULT: 2.88 ms against 2.46 ms at P = 4213, 0.324
  // against 0.314 ms at P = 640.  Each hand-over between workgroups costs a release (L2 write-back of the XCD) + a polled counter
  // + an acquire (L1 / L2 invalidate) = ~3 us, as much as the launch boundary it replaces, and the invalidates evict the operand
  // panels the other workgroups of that XCD were re-using.
  const int panel_mode = env_int("VMCPDE_CHOL_PANEL", 0);
  // the 16-wide schedule of the diagonal block (chol_diag16_body; 0: the 32-wide chol_diag_body, which the debug phases belong to)
  const bool diag16 = env_int("VMCPDE_CHOL_DIAG16", 1) != 0 && diag_phase == 5;
  auto launch_diag = [&](hipStream_t s_, int kk0, int nn, double* Li) {
    if (diag16) hipLaunchKernelGGL(chol_diag16_kernel, dim3(1), dim3(D16_THREADS), diag_lds, s_, A, lda, kk0, nn, Li, info);
    else hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), diag_lds, s_, A, lda, kk0, nn, Li, info, diag_phase);
  };
  unsigned int* counters = reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + (int64_t)nb * NB * NB * (int64_t)sizeof(double));
  if (panel_mode) VMC_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)chol_counter_bytes(nb), st));
  hipStream_t side = ctx->side;
  hipEvent_t ev_fork = ctx->ev_fork, ev_join = ctx->ev_join;
  {
    const int n0 = (P < NB) ? P : NB;
    launch_diag(st, 0, n0, Linv);
  }
  // Two-level schedule (default).  Panels are grouped into super-panels of W block columns.  Inside a super-panel panel k updates
  // only the block columns up to the super-panel's end (column k+1 by the strips, the others by tiles with K = 128); when the
  // super-panel is complete, everything to its right is updated ONCE with K = 128 W (the first of those block columns by the strips,
  // which the next diagonal block waits for).  A K = 128 tile re-reads and re-writes its 128 KB of C for 4.2 MFLOP and runs at
  // ~31 TFLOP/s (measured at P = 16 181, tools/diag/chol_large_prof.sh); K = 512 amortises that four times.  W = 1 is the
  // one-level right-looking schedule.  Every update launch carries the next diagonal block as workgroup 0 (chol_update_diag16_kernel).
  if (diag16 && !panel_mode) {
    static_assert(2 * TILE_DMA_LDS_BYTES <= ((size_t)NB * DP + 4 * SB * 17 + 2 * SB) * sizeof(double), "two tiles per workgroup of the update launch");
    int W = env_int("VMCPDE_CHOL_SUPER", 0);
    if (W <= 0) W = (nb >= SUPER_MIN_BLOCKS) ? SUPER_WIDTH : 1;
    // VMCPDE_CHOL_SUPER_HEAD = h (> 0): super-panels only over the first h block columns -- where the trailing matrix is large and a
    // K = 128 update is bound by re-reading and re-writing it --, single panels behind (A/B knob)
    int head = env_int("VMCPDE_CHOL_SUPER_HEAD", 0);
    if (head <= 0 || head > nb) head = nb;
    // Lookahead at super-panel level (two-level schedule only): the K = 128 W update that closes super-panel s is split.  Part (a), the
    // block columns of super-panel s + 1, stays on the caller's stream together with the next diagonal block; part (b), everything to
    // the right of s + 1, goes to a side stream whose queue is masked to 7 of every 8 CUs, so the latency chain of super-panel s + 1
    // (diagonal blocks, strips, its K = 128 tiles) always finds free CUs and runs UNDER (b) instead of behind it.  (b) of super-panel s
    // must be complete before the last panel of s + 1 touches the first block column of s + 2 (mode-1 strips) and before (b) of s + 1.
    bool overlap = (W > 1) && head == nb && env_int("VMCPDE_CHOL_OVERLAP", 0) != 0;
    hipStream_t bulk = nullptr;
    if (overlap) {
      DeviceCtxLock lock;
      if (ctx->side_masked_state == 0) {
        uint32_t mask[8];                                   // 256 CUs; bit i = CU i: leave every eighth CU to the chain
        for (int i = 0; i < 8; ++i) mask[i] = 0xFEFEFEFEu;
        ctx->side_masked_state = (hipExtStreamCreateWithCUMask(&ctx->side_masked, 8, mask) == hipSuccess && ctx->side_masked) ? 1 : -1;
        if (ctx->side_masked_state < 0) (void)hipGetLastError();
      }
      overlap = ctx->side_masked_state == 1;
      bulk = ctx->side_masked;
    }
    bool bulk_pending = false;
    for (int k = 0; k + 1 < nb; ++k) {
      const int k0 = k * NB;
      const int nrem = nb - (k + 1);                                  // block rows below the panel
      // the super-panel of k: block columns [s0, s1)
      const int s0 = (k < head) ? (k / W) * W : k;
      const int s1 = (k < head) ? ((s0 + W < head) ? (s0 + W) : head) : k + 1;
      const bool last_of_super = (k + 1 == s1);
      const int kc0 = last_of_super ? s0 * NB : k0;                   // operand columns of this panel's update
      const int nchunk = last_of_super ? (k + 1 - s0) : 1;
      const int c0 = k0 + NB;
      const double* Lk = Linv + (int64_t)k * NB * NB;
      if (last_of_super && bulk_pending) {                            // (b) of the previous super-panel has to be in place now
        VMC_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
        bulk_pending = false;
      }
      if (8 * nrem <= 256) {
        hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, c0, 0, Lk);
        hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, P, kc0, nchunk, c0, 1, Lk);
      } else {
        hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, c0, 0, Lk);
        hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, P, kc0, nchunk, c0, 1, Lk);
      }
      const int jlo = k + 2;
      int jhi = last_of_super ? nb : s1;
      const int jsplit = (s1 + W < nb) ? (s1 + W) : nb;              // end of the next super-panel
      const bool split = overlap && last_of_super && jsplit < nb && jlo < jsplit;
      if (split) jhi = jsplit;
      auto count_tiles = [&](int a, int b) { int64_t n = 0; for (int j = a; j < b; ++j) n += nb - j; return n; };
      const int64_t ntiles = count_tiles(jlo, jhi);
      const int tpw = (ntiles <= 255) ? 1 : 2;
      if (split) {      // the strips of this panel completed the L columns of the super-panel: fork behind them, in front of part (a)
        VMC_CHECK_HIP(hipEventRecord(ev_fork, st));
        VMC_CHECK_HIP(hipStreamWaitEvent(bulk, ev_fork, 0));
      }
      hipLaunchKernelGGL(chol_update_diag16_kernel, dim3(1u + (unsigned)((ntiles + tpw - 1) / tpw)), dim3(D16_THREADS), diag_lds, st, A, lda, P, kc0,
                         nchunk * NB, jlo, jhi, nb, (int)ntiles, tpw, c0, Linv + (int64_t)(k + 1) * NB * NB, info);
      if (split) {
        const int64_t nt_b = count_tiles(jsplit, nb);
        hipLaunchKernelGGL(chol_update_diag16_kernel, dim3(1u + (unsigned)((nt_b + 1) / 2)), dim3(D16_THREADS), diag_lds, bulk, A, lda, P, kc0,
                           nchunk * NB, jsplit, nb, nb, (int)nt_b, 2, -1, (double*)nullptr, info);
        VMC_CHECK_HIP(hipEventRecord(ev_join, bulk));
        bulk_pending = true;
      }
    }
    if (bulk_pending) VMC_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
    VMC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * NB;
    const int n = (P - k0 < NB) ? (P - k0) : NB;
    const int rem = P - (k0 + n);
    if (rem <= 0) break;
    const int nrem = (rem + TILE - 1) / TILE;
    double* Linv_nextp = Linv + (int64_t)(k + 1) * NB * NB;
    const int64_t tiles_all = tri_count(nrem) - 1;                 // every lower tile of the trailing matrix but the next diagonal one
    if (panel_mode && tiles_all <= fused_max) {
      // one launch per panel (chol_panel_kernel): the strips and the next diagonal block must be resident together
      const bool sr16 = (8 * nrem + 1 <= 232);
      const int S = sr16 ? 8 * nrem : 4 * nrem;
      if (S + 1 <= 248) {
        unsigned int* cnt = counters + (int64_t)k * (nb + 2);
        const unsigned grid = (unsigned)(S + 1 + tiles_all);
        if (sr16)
          hipLaunchKernelGGL(chol_panel_kernel<16>, dim3(grid), dim3(256), diag_lds, st, A, lda, P, k0, Linv + (int64_t)k * NB * NB, Linv_nextp,
                             info, cnt, nrem, diag_phase);
        else
          hipLaunchKernelGGL(chol_panel_kernel<32>, dim3(grid), dim3(256), diag_lds, st, A, lda, P, k0, Linv + (int64_t)k * NB * NB, Linv_nextp,
                             info, cnt, nrem, diag_phase);
        continue;
      }
    }
    // n == NB here (a panel with rows below it is always full); row strips of 32 over the zero-padded row range
    if (8 * nrem <= 256) {
      hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, k0 + NB, 0, Linv + (int64_t)k * NB * NB);
      hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, k0 + NB, 1, Linv);
    } else {
      hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, k0 + NB, 0, Linv + (int64_t)k * NB * NB);
      hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, P, k0, 1, k0 + NB, 1, Linv);
    }
    double* Linv_next = Linv + (int64_t)(k + 1) * NB * NB;
    const int k1 = k0 + n;
    const int n1 = (P - k1 < NB) ? (P - k1) : NB;
    const int64_t tiles = (nrem > 1) ? tri_count(nrem - 1) : 0;
    if (tiles == 0) {
      launch_diag(st, k1, n1, Linv_next);
    } else if (tiles <= fused_max) {
      hipLaunchKernelGGL(chol_syrk_diag_kernel, dim3(1u + (unsigned)tiles), dim3(256), diag_lds, st, A, lda, P, k0, n, Linv_next, info, diag_phase);
    } else {
      VMC_CHECK_HIP(hipEventRecord(ev_fork, st));
      VMC_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
      launch_diag(side, k1, n1, Linv_next);
      VMC_CHECK_HIP(hipEventRecord(ev_join, side));
      hipLaunchKernelGGL(chol_syrk_kernel, dim3((unsigned)tiles), dim3(256), TILE_DMA_LDS_BYTES, st, A, lda, P, k0, n);
      VMC_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
    }
  }
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// ABI 7: Cholesky of the leading ncols x ncols block of a TALL matrix and the panel solve of everything below it,
//   A = [ A11 ; A21 ]  (nrows x ncols, row-major, lda >= ncols)   ->   [ L11 ; A21 L11^-T ],
// with the launches of vmcpde_cholesky (16-wide diagonal block riding in the update launch, 16- / 32-row strips): the block columns behind
// ncols do not exist, so a panel's update stops at the last block column.  What it is for: the factor of S on a FIXED pivot set Pi --
// A11 = S[Pi, Pi], A21 = S[:, Pi] -- i.e. the columns L = S[:, Pi] L11^-T of the low-rank solve without the pivot search (lowrank.py).
extern "C" int vmcpde_cholesky_tall(double* A, int64_t lda, int32_t nrows, int32_t ncols, int32_t* info, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
  VMC_REQUIRE(A && info && workspace, "vmcpde_cholesky_tall: null argument");
  VMC_REQUIRE(ncols >= NB && (ncols % NB) == 0 && nrows >= ncols, "vmcpde_cholesky_tall: ncols must be a positive multiple of %d and <= nrows", NB);
  VMC_REQUIRE(lda >= ncols && (lda % 2) == 0 && lda < (1 << 22), "lda must be even, >= ncols and < 2^22");
  VMC_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0, "A must be 16-byte aligned");
  VMC_REQUIRE(workspace_bytes >= vmcpde_chol_workspace_bytes(ncols), "cholesky workspace too small (vmcpde_chol_workspace_bytes(ncols))");
  hipStream_t st = (hipStream_t)stream;
  double* Linv = static_cast<double*>(workspace);
  const int nbc = ncols / NB, nb = (nrows + NB - 1) / NB;
  const size_t diag_lds = ((size_t)NB * DP + 4 * SB * 17 + 2 * SB) * sizeof(double);
  const size_t strip_lds = (size_t)(NB + 32) * SP * sizeof(double);
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  {
    DeviceCtxLock lock;
    if (!ctx->attr_chol_tall) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_strip_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_strip_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_diag16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_update_diag16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
      ctx->attr_chol_tall = true;
    }
  }
  VMC_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(chol_diag16_kernel, dim3(1), dim3(D16_THREADS), diag_lds, st, A, lda, 0, NB, Linv, info);
  for (int k = 0; k < nbc; ++k) {
    const int k0 = k * NB, c0 = k0 + NB;
    const int nrem = nb - (k + 1);                                     // block rows below the panel (all of A21 included)
    if (nrem <= 0) break;
    const double* Lk = Linv + (int64_t)k * NB * NB;
    const bool last = (k + 1 == nbc);
    if (8 * nrem <= 256) {
      hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, nrows, k0, 1, c0, 0, Lk);
      if (!last) hipLaunchKernelGGL(chol_strip_kernel<16>, dim3(8 * nrem), dim3(256), strip_lds, st, A, lda, nrows, k0, 1, c0, 1, Lk);
    } else {
      hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, nrows, k0, 1, c0, 0, Lk);
      if (!last) hipLaunchKernelGGL(chol_strip_kernel<32>, dim3(4 * nrem), dim3(256), strip_lds, st, A, lda, nrows, k0, 1, c0, 1, Lk);
    }
    if (last) break;
    // the other block columns up to ncols (rows j .. nb-1 of column j), and the next diagonal block as workgroup 0
    const int jlo = k + 2, jhi = nbc;
    int64_t ntiles = 0;
    for (int j = jlo; j < jhi; ++j) ntiles += nb - j;
    const int tpw = (ntiles <= 255) ? 1 : 2;
    hipLaunchKernelGGL(chol_update_diag16_kernel, dim3(1u + (unsigned)((ntiles + tpw - 1) / tpw)), dim3(D16_THREADS), diag_lds, st, A, lda, nrows, k0, NB,
                       jlo, jhi > jlo ? jhi : jlo, nb, (int)ntiles, tpw, c0, Linv + (int64_t)(k + 1) * NB * NB, info);
  }
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_chol_solve(const double* L, int64_t lda, int32_t P, const double* F, double* u, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(L && F && u && workspace, "vmcpde_chol_solve: null argument");
  VMC_REQUIRE(P >= 1 && lda >= P, "bad shape");
  VMC_REQUIRE(workspace_bytes >= vmcpde_chol_workspace_bytes(P), "workspace must be the one vmcpde_cholesky filled");
  hipStream_t st = (hipStream_t)stream;
  const double* Linv = static_cast<const double*>(workspace);
  const int nb = (P + NB - 1) / NB;
  if (nb <= SOLVE_SMALL_MAX_BLOCKS && env_int("VMCPDE_SOLVE_SMALL", 1) != 0) {      // one workgroup, one launch (the r x r system of the Woodbury application)
    hipLaunchKernelGGL(solve_small_kernel, dim3(1), dim3(256), 0, st, L, lda, P, nb, F, u, Linv);
    VMC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (u != F) VMC_CHECK_HIP(hipMemcpyAsync(u, F, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, st));
  // L y = F
  hipLaunchKernelGGL(solve_diag_only, dim3(1), dim3(256), 0, st, u, P, 0, 0, Linv);
  for (int k = 0; k + 1 < nb; ++k)
    hipLaunchKernelGGL(solve_fwd_step, dim3(nb - 1 - k), dim3(256), 0, st, L, lda, P, k, u, Linv);
  // L^T u = y
  hipLaunchKernelGGL(solve_diag_only, dim3(1), dim3(256), 0, st, u, P, nb - 1, 1, Linv);
  for (int k = nb - 1; k >= 1; --k) hipLaunchKernelGGL(solve_bwd_step, dim3(k), dim3(256), 0, st, L, lda, P, k, u, Linv);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_chol_backsolve(const double* L, int64_t lda, int32_t P, const double* y, double* u, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(L && y && u && workspace, "vmcpde_chol_backsolve: null argument");
  VMC_REQUIRE(P >= 1 && lda >= P, "bad shape");
  VMC_REQUIRE(workspace_bytes >= vmcpde_chol_workspace_bytes(P), "workspace must be the one vmcpde_cholesky filled");
  hipStream_t st = (hipStream_t)stream;
  const double* Linv = static_cast<const double*>(workspace);
  const int nb = (P + NB - 1) / NB;
  // L^T u = y
  if (env_int("VMCPDE_BACKSOLVE_PERSISTENT", 1) && nb <= 256 && u != y) {
    // one launch, workgroups chained by per-block flags (the counter area of the workspace is free once the factor exists)
    unsigned int* flags = reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + (int64_t)nb * NB * NB * (int64_t)sizeof(double));
    VMC_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)nb * sizeof(unsigned int), st));
    hipLaunchKernelGGL(solve_bwd_persistent, dim3(nb), dim3(256), 0, st, L, lda, P, nb, y, u, Linv, flags);
    VMC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (u != y) VMC_CHECK_HIP(hipMemcpyAsync(u, y, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(solve_diag_only, dim3(1), dim3(256), 0, st, u, P, nb - 1, 1, Linv);
  for (int k = nb - 1; k >= 1; --k) hipLaunchKernelGGL(solve_bwd_step, dim3(k), dim3(256), 0, st, L, lda, P, k, u, Linv);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_gemv(const double* A, int64_t lda, int32_t rows, int32_t cols, int trans, const double* x, double* y,
                           void* stream) {
  VMC_REQUIRE(A && x && y, "vmcpde_gemv: null argument");
  VMC_REQUIRE(rows >= 1 && cols >= 1 && lda >= cols, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (!trans)
    hipLaunchKernelGGL(gemv_n_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, A, lda, rows, cols, x, y);
  else
    hipLaunchKernelGGL(gemv_t_kernel, dim3((cols + 31) / 32), dim3(1024), 0, st, A, lda, rows, cols, x, y);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_pinv_coeffs(const double* ev, const double* VtF, const double* snr, int32_t P, double svdTol, double snrTol,
                                  double* coef, void* stream) {
  VMC_REQUIRE(ev && VtF && coef && P >= 1, "vmcpde_pinv_coeffs: bad argument");
  hipLaunchKernelGGL(pinv_coeffs_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, ev, VtF, snr, P, svdTol, snrTol,
                     coef);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}
