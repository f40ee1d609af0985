// The r x r symmetric eigenproblem at the core of the rank-revealing spectral solve (reference: tdvp.py:57-64, `eigh` of the full matrix;
// here step 3 of csrc/eig.hip's scheme: Rayleigh-Ritz on range(L)).
//
// With S = L L^T + R and L = Q1 R1 (Householder), Q1^T S Q1 = R1 R1^T + Q1^T R Q1, and the eigenvectors of R1 R1^T are the LEFT singular
// vectors of the r x r triangular factor R1: a one-sided (Hestenes) Jacobi method on its columns -- rotate pairs of columns until all are
// orthogonal; the columns are then u_j sigma_j, the eigenvalues sigma_j^2.  One-sided Jacobi works on the factor (condition sqrt of the
// matrix's), never forms a product, is accurate to the norm of each COLUMN (R1's columns span seven decades), and needs no tridiagonal form.
// A library `syevd` on this 206 x 206 core is a chain of some hundred small launches (4.35 ms of a 6.8 ms decomposition at the headline
// size); here a sweep is a few launches.
//
// Blocked for one CU's LDS: the columns are dealt into blocks of K; a STEP gives every workgroup a PAIR of blocks (2 K columns x r rows in
// LDS, 128 KB at r = 256, K = 32), which it sweeps completely (round-robin over its 2 K columns: 2 K - 1 rounds of K disjoint pairs, one
// pair per wave at a time); a SWEEP is the round-robin over the block pairs: nblk - 1 steps (launches) of nblk / 2 workgroups.  Converged
// when a whole sweep rotates nothing: every pair has |x.y| <= tol |x| |y|.  Measured on the model's factor (r = 194 .. 262): 6 working
// sweeps and one that confirms.  Launches behind the confirming sweep return at once (a device-side counter per sweep), so the host
// launches a fixed number of sweeps and reads nothing.
//
// Storage: CT[c][i] = C[i][c] -- row c of CT is column c of C (contiguous), ld = `rows` (a multiple of 64), `cols` rows (a multiple of
// 2 K), both zero padded (a zero column is never rotated).
#include "common.hpp"
#include "wave_ops.hpp"

namespace vmcpde {
namespace {

constexpr int JC_THREADS = 1024;
constexpr int JC_WAVES = JC_THREADS / 64;
constexpr int JC_MAX_LDS = 128 * 1024;         // 2 K columns x rows x 8 B
constexpr int JC_MAX_SWEEPS = 30;

// round-robin tournament over n players (n even): in round s player n - 1 meets s, and (s + i) mod (n - 1) meets (s - i) mod (n - 1)
__device__ __forceinline__ void tournament_pair(int n, int s, int i, int& a, int& b) {
  const int m = n - 1;
  if (i == 0) {
    a = m;
    b = s % m;
  } else {
    a = (s + i) % m;
    b = ((s - i) % m + m) % m;
  }
}

// One step: workgroup g of nblk / 2 sweeps the columns of block pair (a, b) of round `step`.  counters[sweep] counts the rotations of the
// sweep; a sweep after one that rotated nothing does nothing.  RPL = rows per lane (rows / 64).
template <int RPL>
__global__ __launch_bounds__(JC_THREADS) void jacobi_step_kernel(double* __restrict__ CT, int rows, int K, int nblk, int step, int sweep,
                                                                  int* __restrict__ counters, double tol) {
  if (sweep > 0 && counters[sweep - 1] == 0) return;                 // converged: the previous sweep found every pair orthogonal
  extern __shared__ double col[];                                      // [2 K][rows]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int ba, bb;
  tournament_pair(nblk, step, (int)blockIdx.x, ba, bb);
  const int ncol = 2 * K;
  // load: local column c < K is global column ba K + c, c >= K is bb K + (c - K)
  for (int c = wave; c < ncol; c += JC_WAVES) {
    const double* __restrict__ src = CT + (int64_t)((c < K ? ba * K + c : bb * K + (c - K))) * rows;
#pragma unroll
    for (int q = 0; q < RPL; ++q) col[c * rows + lane + 64 * q] = src[lane + 64 * q];
  }
  __syncthreads();
  int rotated = 0;
  for (int round = 0; round < ncol - 1; ++round) {
    for (int i = wave; i < K; i += JC_WAVES) {
      int u, v;
      tournament_pair(ncol, round, i, u, v);
      double* __restrict__ x = col + u * rows;
      double* __restrict__ y = col + v * rows;
      double xr[RPL], yr[RPL];
      double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        xr[q] = x[lane + 64 * q];
        yr[q] = y[lane + 64 * q];
        al = fma(xr[q], xr[q], al);
        be = fma(yr[q], yr[q], be);
        ga = fma(xr[q], yr[q], ga);
      }
      al = wave_sum(al);
      be = wave_sum(be);
      ga = wave_sum(ga);
      // rotate when the columns are not orthogonal to working accuracy (|cos| > tol); zero columns (the padding) never are
      if (fabs(ga) > tol * sqrt(al * be)) {
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
        const double cs = 1.0 / sqrt(fma(tt, tt, 1.0)), sn = cs * tt;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          x[lane + 64 * q] = fma(cs, xr[q], -sn * yr[q]);
          y[lane + 64 * q] = fma(sn, xr[q], cs * yr[q]);
        }
        ++rotated;
      }
    }
    __syncthreads();
  }
  for (int c = wave; c < ncol; c += JC_WAVES) {
    double* __restrict__ dst = CT + (int64_t)((c < K ? ba * K + c : bb * K + (c - K))) * rows;
#pragma unroll
    for (int q = 0; q < RPL; ++q) dst[lane + 64 * q] = col[c * rows + lane + 64 * q];
  }
  if (lane == 0 && rotated) atomicAdd(counters + sweep, rotated);
}

// columns per workgroup (2 K) for `rows` padded rows: the largest power of two whose LDS image fits, at most 64
int jacobi_cols_per_group(int rows) {
  int ncol = 64;
  while (ncol > 2 && (int64_t)ncol * rows * 8 > JC_MAX_LDS) ncol >>= 1;
  return ncol;
}

}  // namespace
}  // namespace vmcpde

using namespace vmcpde;

extern "C" int vmcpde_jacobi_shape(int32_t r, int32_t* rows_pad, int32_t* cols_pad) {
  VMC_REQUIRE(rows_pad && cols_pad, "vmcpde_jacobi_shape: null argument");
  VMC_REQUIRE(r >= 1 && r <= 1024, "vmcpde_jacobi_shape: 1 <= r <= 1024 (a pair of column blocks must fit one CU's LDS)");
  const int rows = (r + 63) / 64 * 64;
  const int ncol = jacobi_cols_per_group(rows);
  *rows_pad = rows;
  *cols_pad = (r + ncol - 1) / ncol * ncol;
  return 0;
}

extern "C" int vmcpde_jacobi_columns(double* CT, int32_t r, int32_t n_sweeps, int32_t* counters, void* stream) {
  VMC_REQUIRE(CT && counters, "vmcpde_jacobi_columns: null argument");
  VMC_REQUIRE(r >= 1 && r <= 1024, "vmcpde_jacobi_columns: 1 <= r <= 1024");
  VMC_REQUIRE(n_sweeps >= 1 && n_sweeps <= JC_MAX_SWEEPS, "vmcpde_jacobi_columns: 1 <= n_sweeps <= %d", JC_MAX_SWEEPS);
  const int rows = (r + 63) / 64 * 64;
  const int ncol = jacobi_cols_per_group(rows), K = ncol / 2;
  const int cols = (r + ncol - 1) / ncol * ncol, nblk = cols / K;
  const size_t lds = (size_t)ncol * rows * sizeof(double);
  const double tol = sqrt((double)rows) * 1.1102230246251565e-16;     // dgesvj's criterion: sqrt(m) eps
  hipStream_t st = (hipStream_t)stream;
  VMC_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)(n_sweeps + 1) * sizeof(int32_t), st));
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
#define VMC_JACOBI(RPL)                                                                                                            \
  do {                                                                                                                             \
    {                                                                                                                              \
      DeviceCtxLock lock;                                                                                                          \
      if (ctx->jacobi_lds[RPL - 1] < (int)lds) {                                                                                   \
        VMC_CHECK_HIP(hipFuncSetAttribute((const void*)jacobi_step_kernel<RPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ctx->jacobi_lds[RPL - 1] = (int)lds;                                                                                       \
      }                                                                                                                            \
    }                                                                                                                              \
    for (int sw = 0; sw < n_sweeps; ++sw)                                                                                          \
      for (int s = 0; s < nblk - 1; ++s)                                                                                           \
        hipLaunchKernelGGL((jacobi_step_kernel<RPL>), dim3(nblk / 2), dim3(JC_THREADS), lds, st, CT, rows, K, nblk, s, sw, counters, tol); \
  } while (0)
  switch (rows / 64) {
    case 1: VMC_JACOBI(1); break;
    case 2: VMC_JACOBI(2); break;
    case 3: VMC_JACOBI(3); break;
    case 4: VMC_JACOBI(4); break;
    case 5: VMC_JACOBI(5); break;
    case 6: VMC_JACOBI(6); break;
    case 7: VMC_JACOBI(7); break;
    case 8: VMC_JACOBI(8); break;
    case 9: VMC_JACOBI(9); break;
    case 10: VMC_JACOBI(10); break;
    case 11: VMC_JACOBI(11); break;
    case 12: VMC_JACOBI(12); break;
    case 13: VMC_JACOBI(13); break;
    case 14: VMC_JACOBI(14); break;
    case 15: VMC_JACOBI(15); break;
    default: VMC_JACOBI(16); break;
  }
#undef VMC_JACOBI
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}
