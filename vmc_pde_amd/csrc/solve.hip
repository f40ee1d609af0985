// On-device solve of the TDVP equation S theta_dot = F.
//
// Reference: tdvp.py:57-94.  The reference diagonalises S on the host (np.linalg.eigh, tdvp.py:61-64) and applies
// a soft spectral cut-off.  The fast path here is a regularised Cholesky of A = S + diagonalShift*diag(S)
// (tdvp.py:50-51 defines that shifted matrix; north_star asks for the Cholesky), blocked right-looking with
// 128-wide panels: diagonal block factor + triangular inverse in LDS, panel solve and trailing update on the fp64
// MFMA tile engine (mfma_tile.hpp).  The spectral (parity) path keeps the reference's formulae: eigh comes from the
// caller, the cut-off coefficients (tdvp.py:82-89) and the mat-vecs (tdvp.py:66,91,93,94) are the kernels below.
#include "mfma_tile.hpp"

using namespace vmcpde;

namespace {

constexpr int NB = TILE;        // panel width 128
constexpr int DP = NB + 1;      // LDS pitch of the diagonal block (odd: column walks are conflict-free)

// acc -> C[row0.., col0..] (ldc), mode 0: C = acc, mode 1: C -= acc
__device__ __forceinline__ void tile_store_global(const double4_t (&acc)[4][4], double* __restrict__ C, int64_t ldc, int rows, int cols,
                                                  int mode) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
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

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Wave 0: Cholesky of the 32x32 diagonal sub-block at (j0,j0) of the LDS block `a` and its triangular inverse.
// Lane i keeps row i in registers (lanes 32..63 mirror 0..31); each column's multipliers are exchanged through a
// 32-double LDS line read back as broadcasts (one wave: DS operations execute in issue order, no workgroup barrier).
// The factor loop is fully unrolled (static register indices) and free of per-lane compares (the diagonal entry goes
// through an LDS slot) because hipcc otherwise hoists the lane masks into SGPRs and spills; a sched_barrier closes every
// column to keep the live set small.
// L goes to the lower triangle; X = L^-1 is kept transposed in the dead strict upper part: X[r][c] <-> a[c][r+1].
// scr: 32-double exchange line.
__device__ __forceinline__ int factor_inverse_32(double* a, int j0, double* scr) {
  const int i = threadIdx.x & 31;
  const int rowbase = (j0 + i) * DP + j0;
  double* col = scr;
  double r[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) r[c] = a[rowbase + c];
  int pinned = 0;
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const double p = readlane_f64(r[j], j);
    const bool bad = !(p > 0.0) || !isfinite(p);     // wave-uniform: dead / padded parameter (H7) -> unit row
    pinned += bad ? 1 : 0;
    const double d = bad ? 1.0 : sqrt(p);
    col[i] = bad ? 0.0 : r[j] / d;                   // multipliers of the rows below (junk for rows above: harmless)
    __builtin_amdgcn_wave_barrier();
    col[j] = d;                                      // the diagonal entry, same value from every lane
    __builtin_amdgcn_wave_barrier();
    const double l = col[i];
    r[j] = l;
#pragma unroll
    for (int c = j + 1; c < SB; ++c) r[c] = fma(-l, col[c], r[c]);   // meaningful for lanes i >= c; others carry junk
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // the junk above the diagonal is overwritten by X below, so the whole row can be stored unmasked
#pragma unroll
  for (int c = 0; c < SB; ++c) a[rowbase + c] = r[c];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // column i of X by forward substitution: x[k] = (delta_ki - sum_{i<=m<k} L[k][m] x[m]) / L[k][k], k = i..31.
  // x lives in its final place (row j0+i of `a`, shifted by one column); the inner loop is unrolled by 8 so eight LDS
  // read pairs are in flight (a fully unrolled register-resident version makes hipcc spill ~600 VGPRs).
  if ((threadIdx.x & 63) < SB) {
    double* xrow = a + rowbase + 1;
    for (int k = i; k < SB; ++k) {
      const double* lrow = a + (j0 + k) * DP + j0;
      double acc = (k == i) ? 1.0 : 0.0;
#pragma unroll 8
      for (int m = i; m < k; ++m) acc = fma(-lrow[m], xrow[m], acc);
      xrow[k] = acc / lrow[k];
    }
  }
  return pinned;
}

// Factor the diagonal block A[k0:k0+n, k0:k0+n] = L L^T in LDS (four 32-wide sub-steps), write L back (lower triangle)
// and X = L^-1 (dense lower triangular, zero elsewhere) to Linv.  One workgroup of 256 threads; rows/cols >= n are
// padded with the identity.  LDS: the 128 x 129 block + three 32 x 33 scratch tiles.
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                                       int* __restrict__ info) {
  extern __shared__ double a[];          // [NB][DP] followed by T[3][SB][SB+1]
  double* T = a + NB * DP;
  double* scr = T + 3 * SB * (SB + 1);   // exchange line of factor_inverse_32
  const int t = threadIdx.x;
  // 128 x 128 block -> LDS: 32 double2 loads per thread, issued 16 at a time so the HBM/L2 latency is paid twice, not 64x
#pragma unroll 1
  for (int q0 = 0; q0 < 32; q0 += 16) {
    double2_t v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx2 = t + 256 * (q0 + q), i = idx2 >> 6, j = (idx2 & 63) * 2;
      v[q] = (double2_t){0.0, 0.0};
      if (i < n && j < n) v[q] = *reinterpret_cast<const double2_t*>(A + (int64_t)(k0 + i) * lda + k0 + j);   // pitch even: 16-B aligned
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx2 = t + 256 * (q0 + q), i = idx2 >> 6, j = (idx2 & 63) * 2;
      a[i * DP + j] = (i < n && j < n) ? v[q].x : ((i == j) ? 1.0 : 0.0);
      a[i * DP + j + 1] = (i < n && j + 1 < n) ? v[q].y : ((i == j + 1) ? 1.0 : 0.0);
    }
  }
  if (t < NB) a[t * DP + NB] = 0.0;
  __syncthreads();

#pragma nounroll
  for (int jb = 0; jb < NB / SB; ++jb) {
    const int j0 = jb * SB;
    if (t < 64) {
      const int pinned = factor_inverse_32(a, j0, scr);
      if (t == 0 && pinned) atomicAdd(info, pinned);
    }
    __syncthreads();
    const int R = NB - j0 - SB;          // rows (and trailing columns) below this sub-block
    if (R > 0) {
      // panel below: L[i][j0+c] = sum_{m<=c} A[i][j0+m] X[c][m]   (thread: row t/8 of each 32-row pass, 4 columns)
      double accp[3][4];
      const int cq = (t & 7) * 4;
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
#pragma unroll
        for (int q = 0; q < 4; ++q) accp[ps][q] = 0.0;
        if (ps * SB < R) {
          const int i = j0 + SB + ps * SB + (t >> 3);
#pragma unroll 2
          for (int m = 0; m < SB; ++m) {
            const double av = a[i * DP + j0 + m];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c = cq + q;
              const double xv = (m <= c) ? a[(j0 + m) * DP + j0 + c + 1] : 0.0;
              accp[ps][q] = fma(av, xv, accp[ps][q]);
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < 3; ++ps)
        if (ps * SB < R) {
          const int i = j0 + SB + ps * SB + (t >> 3);
#pragma unroll
          for (int q = 0; q < 4; ++q) a[i * DP + j0 + cq + q] = accp[ps][q];
        }
      __syncthreads();
      // trailing update inside the block: A[i][c] -= sum_m L[i][j0+m] L[c][j0+m], 4x4 register tiles, lower part only
      const int nt = R / 4;
      for (int tile = t; tile < nt * nt; tile += 256) {
        const int ti = tile / nt, tc = tile - ti * nt;
        if (tc > ti) continue;
        const int i0 = j0 + SB + 4 * ti, c0 = j0 + SB + 4 * tc;
        double acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 2
        for (int m = 0; m < SB; ++m) {
          double li[4], lc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { li[u] = a[(i0 + u) * DP + j0 + m]; lc[u] = a[(c0 + u) * DP + j0 + m]; }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(li[u], lc[v], acc[u][v]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (c0 + v <= i0 + u) a[(i0 + u) * DP + c0 + v] -= acc[u][v];
      }
      __syncthreads();
    }
  }

  // write L back (lower triangle incl. diagonal)
  for (int idx = t; idx < n * NB; idx += 256) {
    const int i = idx / NB, j = idx - i * NB;
    if (j <= i) A[(int64_t)(k0 + i) * lda + k0 + j] = a[i * DP + j];
  }

  // off-diagonal blocks of X = L^-1:  X_{ib,jb} = -X_{ib,ib} * sum_{mb=jb}^{ib-1} L_{ib,mb} X_{mb,jb},  by block distance.
  // One 4x4 register tile per thread (64 tiles per 32x32 block, at most 3 blocks per round).
  // Storage reminder: X[row][col] (row >= col) <-> a[col][row+1]; entries with row < col of a diagonal X block are L's, so
  // they are masked to zero when read.
#pragma nounroll
  for (int dlt = 1; dlt < NB / SB; ++dlt) {
    const int npairs = NB / SB - dlt;
    const bool work = t < npairs * 64;
    const int p = t >> 6, r0 = ((t & 63) >> 3) * 4, c0 = (t & 7) * 4;
    const int jb = p, ib = p + dlt;
    if (work) {
      double acc[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
      for (int mb = jb; mb < ib; ++mb) {
        const double* lp = a + (SB * ib + r0) * DP + SB * mb;
        const double* xp = a + (SB * jb + c0) * DP + SB * mb + 1;
        const bool diag = (mb == jb);
#pragma unroll 2
        for (int m = 0; m < SB; ++m) {
          double lv[4], xv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) lv[u] = lp[u * DP + m];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            xv[v] = xp[v * DP + m];
            if (diag && m < c0 + v) xv[v] = 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(lv[u], xv[v], acc[u][v]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) T[(p * SB + r0 + u) * (SB + 1) + c0 + v] = acc[u][v];
    }
    __syncthreads();
    if (work) {
      double acc[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
      const double* xii = a + (SB * ib) * DP + SB * ib + r0 + 1;      // X_ii[r0+u][m] <-> xii[m*DP + u]
      const double* tp = T + (p * SB) * (SB + 1) + c0;
#pragma unroll 2
      for (int m = 0; m < r0 + 4; ++m) {
        double xv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = xii[m * DP + u];
          if (m > r0 + u) xv[u] = 0.0;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) tv[v] = tp[m * (SB + 1) + v];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = fma(xv[u], tv[v], acc[u][v]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) a[(SB * jb + c0 + v) * DP + SB * ib + r0 + u + 1] = -acc[u][v];
    }
    __syncthreads();
  }
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, c = idx - i * NB;
    Linv[idx] = (i < n && c <= i) ? a[c * DP + i + 1] : 0.0;
  }
}

// Panel solve: A[r0.., k0:k0+n] <- A[r0.., k0:k0+n] * L_kk^-T = B * Linv^T  (TN tile GEMM, K = n)
__global__ __launch_bounds__(256, 2) void chol_trsm_kernel(double* __restrict__ A, int64_t lda, int P, int k0, int n,
                                                           const double* __restrict__ Linv) {
  extern __shared__ double lds[];
  const int row0 = k0 + n + blockIdx.x * TILE;
  const int rows = (P - row0 < TILE) ? (P - row0) : TILE;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop<false>(acc, A + k0, lda, row0, Linv, NB, 0, false, 0, n, nullptr, 0, nullptr, 0, nullptr, lds);
  tile_store_global(acc, A + (int64_t)row0 * lda + k0, lda, rows, n, 0);
}

// Trailing update: A22 -= L21 L21^T on the lower-triangle tiles (syrk, K = n)
__global__ __launch_bounds__(256, 2) void chol_syrk_kernel(double* __restrict__ A, int64_t lda, int P, int k0, int n) {
  extern __shared__ double lds[];
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  const int base = k0 + n;
  const int rowA0 = base + ti * TILE, rowB0 = base + tj * TILE;
  const int rowsA = (P - rowA0 < TILE) ? (P - rowA0) : TILE;
  const int rowsB = (P - rowB0 < TILE) ? (P - rowB0) : TILE;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop<false>(acc, A + k0, lda, rowA0, A + k0, lda, rowB0, ti == tj, 0, n, nullptr, 0, nullptr, 0, nullptr, lds);
  tile_store_global(acc, A + (int64_t)rowA0 * lda + rowB0, lda, rowsA, rowsB, 1);
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
#pragma unroll 1
    for (int rr = 0; rr < 32; rr += 8) {
      double s[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double2_t v = *reinterpret_cast<const double2_t*>(Linv + (wave * 32 + rr + u) * NB + 2 * lane);
        s[u] = v.x * x0 + v.y * x1;     // Linv is exactly zero above the diagonal and in padded rows
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] = wave_sum(s[u]);
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (wave * 32 + rr + u < n) r[wave * 32 + rr + u] = s[u];
      }
    }
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
#pragma unroll 1
  for (int rr = 0; rr < 32; rr += 8) {
    double s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double2_t v = *reinterpret_cast<const double2_t*>(L + (int64_t)(b0 + wave * 32 + rr + u) * lda + k0 + 2 * lane);
      s[u] = v.x * y0 + v.y * y1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = wave_sum(s[u]);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (wave * 32 + rr + u < nbk) r[b0 + wave * 32 + rr + u] -= s[u];
    }
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

__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, int64_t lda, int rows, int cols,
                                                    const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  double s = 0.0;
  if (col < cols)
    for (int r = q; r < rows; r += 4) s += A[(int64_t)r * lda + col] * x[r];
  part[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && col < cols) y[col] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
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
  coef[k] = invEv * reg * VtF[k];                                     // :91
}

}  // namespace

extern "C" int64_t vmcpde_chol_workspace_bytes(int32_t P) {
  if (P <= 0) return 0;
  const int nb = (P + NB - 1) / NB;
  return (int64_t)nb * NB * NB * (int64_t)sizeof(double);
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
  const size_t diag_lds = ((size_t)NB * DP + 3 * SB * (SB + 1) + SB) * sizeof(double);
  const size_t tile_lds = TILE_LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
    VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_trsm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
    VMC_CHECK_HIP(hipFuncSetAttribute((const void*)chol_syrk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
    attr_set = true;
  }
  VMC_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), st));
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * NB;
    const int n = (P - k0 < NB) ? (P - k0) : NB;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), diag_lds, st, A, lda, k0, n, Linv + (int64_t)k * NB * NB, info);
    const int rem = P - (k0 + n);
    if (rem > 0) {
      const int nrem = (rem + TILE - 1) / TILE;
      hipLaunchKernelGGL(chol_trsm_kernel, dim3(nrem), dim3(256), tile_lds, st, A, lda, P, k0, n, Linv + (int64_t)k * NB * NB);
      hipLaunchKernelGGL(chol_syrk_kernel, dim3((unsigned)tri_count(nrem)), dim3(256), tile_lds, st, A, lda, P, k0, n);
    }
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

extern "C" int vmcpde_gemv(const double* A, int64_t lda, int32_t rows, int32_t cols, int trans, const double* x, double* y,
                           void* stream) {
  VMC_REQUIRE(A && x && y, "vmcpde_gemv: null argument");
  VMC_REQUIRE(rows >= 1 && cols >= 1 && lda >= cols, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (!trans)
    hipLaunchKernelGGL(gemv_n_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, A, lda, rows, cols, x, y);
  else
    hipLaunchKernelGGL(gemv_t_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, A, lda, rows, cols, x, y);
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
