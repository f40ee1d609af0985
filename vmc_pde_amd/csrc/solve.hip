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

// Factor the diagonal block A[k0:k0+n, k0:k0+n] = L L^T in LDS, write L back (lower triangle) and its inverse to Linv.
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int64_t lda, int k0, int n, double* __restrict__ Linv,
                                                       int* __restrict__ info) {
  extern __shared__ double a[];   // [NB][DP]
  __shared__ double sh_d;
  __shared__ int sh_pin;
  const int t = threadIdx.x;
  for (int idx = t; idx < n * NB; idx += 256) {
    const int i = idx / NB, j = idx - i * NB;
    if (j < n) a[i * DP + j] = A[(int64_t)(k0 + i) * lda + k0 + j];
  }
  __syncthreads();
  int pinned_count = 0;
  for (int j = 0; j < n; ++j) {
    if (t == 0) {
      const double piv = a[j * DP + j];
      const bool bad = !(piv > 0.0) || !isfinite(piv);
      sh_pin = bad ? 1 : 0;
      sh_d = bad ? 1.0 : sqrt(piv);
      a[j * DP + j] = sh_d;
      if (bad) ++pinned_count;
    }
    __syncthreads();
    const double d = sh_d;
    const bool pin = sh_pin != 0;
    for (int i = j + 1 + t; i < n; i += 256) a[i * DP + j] = pin ? 0.0 : a[i * DP + j] / d;
    __syncthreads();
    const int m = n - j - 1;
    for (int idx = t; idx < m * m; idx += 256) {
      const int ii = idx / m, cc = idx - ii * m;
      if (ii >= cc) {
        const int i = j + 1 + ii, c = j + 1 + cc;
        a[i * DP + c] -= a[i * DP + j] * a[c * DP + j];
      }
    }
    __syncthreads();
  }
  if (t == 0 && pinned_count) atomicAdd(info, pinned_count);
  // write L back (lower triangle incl. diagonal)
  for (int idx = t; idx < n * NB; idx += 256) {
    const int i = idx / NB, j = idx - i * NB;
    if (j <= i) A[(int64_t)(k0 + i) * lda + k0 + j] = a[i * DP + j];
  }
  __syncthreads();
  // X = L^-1 by forward substitution, one column per thread.  Column c of X (rows c..n-1) is kept in the now-dead
  // strict upper triangle of the LDS block: X[i][c] <-> a[c][i+1]  (pitch DP = NB+1 leaves room for i+1 = NB).
  if (t < n) {
    const int c = t;
    a[c * DP + c + 1] = 1.0 / a[c * DP + c];
    for (int i = c + 1; i < n; ++i) {
      double s = 0.0;
      for (int m2 = c; m2 < i; ++m2) s += a[i * DP + m2] * a[c * DP + m2 + 1];
      a[c * DP + i + 1] = -s / a[i * DP + i];
    }
  }
  __syncthreads();
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

// y_blk = Linv_blk * r_blk (trans=0) or Linv_blk^T * r_blk (trans=1), in place in r. One workgroup, 256 threads.
__device__ __forceinline__ void diag_apply(const double* __restrict__ Linv, double* __restrict__ r, int n, int trans, double* sh /*[NB]*/) {
  const int t = threadIdx.x;
  if (t < NB) sh[t] = (t < n) ? r[t] : 0.0;
  __syncthreads();
  if (t < n) {
    double s = 0.0;
    if (!trans) {
      for (int m = 0; m <= t; ++m) s += Linv[t * NB + m] * sh[m];
    } else {
      for (int m = t; m < n; ++m) s += Linv[m * NB + t] * sh[m];
    }
    r[t] = s;
  }
  __syncthreads();
}

// Forward substitution step k: every row block b > k does r_b -= L[b][k] y_k; block k+1 then finishes y_{k+1}.
__global__ __launch_bounds__(256) void solve_fwd_step(const double* __restrict__ L, int64_t lda, int P, int k, double* __restrict__ r,
                                                     const double* __restrict__ Linv_all) {
  __shared__ double yk[NB];
  __shared__ double sh[NB];
  const int t = threadIdx.x;
  const int k0 = k * NB;
  const int nk = (P - k0 < NB) ? (P - k0) : NB;
  const int b = k + 1 + blockIdx.x;
  const int b0 = b * NB;
  const int nbk = (P - b0 < NB) ? (P - b0) : NB;
  if (t < NB) yk[t] = (t < nk) ? r[k0 + t] : 0.0;
  __syncthreads();
  // 4 waves x 32 rows; each row: 64 lanes x 2 columns
  const int lane = t & 63, wave = t >> 6;
  for (int rr = 0; rr < 32; ++rr) {
    const int row = wave * 32 + rr;
    double s = 0.0;
    if (row < nbk) {
      const double* __restrict__ lp = L + (int64_t)(b0 + row) * lda + k0;
      if (2 * lane < nk) s += lp[2 * lane] * yk[2 * lane];
      if (2 * lane + 1 < nk) s += lp[2 * lane + 1] * yk[2 * lane + 1];
    }
    s = wave_sum(s);
    if (lane == 0 && row < nbk) r[b0 + row] -= s;
  }
  __syncthreads();
  if (blockIdx.x == 0) diag_apply(Linv_all + (int64_t)b * NB * NB, r + b0, nbk, 0, sh);
}

// Backward substitution step k: every row block b < k does r_b -= L[k][b]^T u_k; block k-1 then finishes u_{k-1}.
__global__ __launch_bounds__(256) void solve_bwd_step(const double* __restrict__ L, int64_t lda, int P, int k, double* __restrict__ r,
                                                     const double* __restrict__ Linv_all) {
  __shared__ double uk[NB];
  __shared__ double part[2][NB];
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
  for (int i = half; i < nk; i += 2) s += L[(int64_t)(k0 + i) * lda + b0 + j] * uk[i];
  part[half][j] = s;
  __syncthreads();
  if (t < NB) r[b0 + t] -= part[0][t] + part[1][t];
  __syncthreads();
  if (b == k - 1) diag_apply(Linv_all + (int64_t)b * NB * NB, r + b0, NB, 1, sh);
}

__global__ __launch_bounds__(256) void solve_diag_only(double* __restrict__ r, int P, int k, int trans, const double* __restrict__ Linv_all) {
  __shared__ double sh[NB];
  const int k0 = k * NB;
  const int nk = (P - k0 < NB) ? (P - k0) : NB;
  diag_apply(Linv_all + (int64_t)k * NB * NB, r + k0, nk, trans, sh);
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
  const size_t diag_lds = (size_t)NB * DP * sizeof(double);
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
