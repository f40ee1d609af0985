// The O(N) and O(P) glue of one TDVP right-hand side, fused into three launches.
//
// Reference: tdvp.py:36-52 (ElocMean / ElocMeanAbs / ElocVar, F0, the optional diagonal shift) and tdvp.py:93-94 (TDVP error and
// solver residual).  The reference spends ten all-reduces and a dozen eager array ops on these; a first version here used
// ~70 tiny torch kernels per step (3-5 us each on the GPU, 0.3 ms of a 22 ms step, and most of the host time).  Now:
//   vmcpde_tdvp_prepare      Eloc statistics + the two appended Jacobian rows of the fused-moments Gram        (before the Gram)
//   vmcpde_tdvp_moments      row sums from the Gram's last tile row, F, means, diag(S), scalars, Cholesky border (after the all-reduce)
//   vmcpde_tdvp_diagnostics  residual and TDVP error from u and S0 u                                           (after the solve)
#include "mfma_tile.hpp"

using namespace vmcpde;

namespace {

constexpr int PREP_BLOCKS_MAX = 128;
constexpr int PREP_THREADS = 256;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// Block partials go to ws[block][4]; the last block to arrive (device-scope counter) adds them in block order, so the sums
// do not depend on the arrival order, and leaves the counter at zero for the next launch.
__global__ __launch_bounds__(PREP_THREADS) void tdvp_prepare_kernel(const double* __restrict__ e, int64_t N, const double* __restrict__ cE_dev,
                                                                    double* __restrict__ row_e, double* __restrict__ row_one,
                                                                    double* __restrict__ sums3, double* __restrict__ count,
                                                                    double* __restrict__ max_out, double* __restrict__ ws) {
  __shared__ double sh[4][PREP_THREADS / 64];
  __shared__ bool last;
  const int nb = gridDim.x, t = threadIdx.x;
  const double cE = cE_dev ? cE_dev[0] : 0.0;
  const int64_t per = ((N + nb - 1) / nb + 1) & ~(int64_t)1;      // even chunk: 16-byte aligned pairs
  const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < N) ? (lo + per) : N;
  double s = 0, s2 = 0, sa = 0, mx = -INFINITY;
  for (int64_t i = lo + t; i < hi; i += PREP_THREADS) {
    const double v = e[i];
    s += v; s2 += v * v; sa += fabs(v); mx = fmax(mx, v);
    if (row_e) { row_e[i] = v - cE; row_one[i] = 1.0; }
  }
  s = wave_sum(s); s2 = wave_sum(s2); sa = wave_sum(sa); mx = wave_max(mx);
  if ((t & 63) == 0) { sh[0][t >> 6] = s; sh[1][t >> 6] = s2; sh[2][t >> 6] = sa; sh[3][t >> 6] = mx; }
  __syncthreads();
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws + 4 * PREP_BLOCKS_MAX);
  if (t == 0) {
    double a = 0, b = 0, c = 0, m = -INFINITY;
    for (int k = 0; k < PREP_THREADS / 64; ++k) { a += sh[0][k]; b += sh[1][k]; c += sh[2][k]; m = fmax(m, sh[3][k]); }
    double* mine = ws + 4 * blockIdx.x;
    mine[0] = a; mine[1] = b; mine[2] = c; mine[3] = m;
    __threadfence();
    last = (atomicAdd(counter, 1u) == (unsigned int)(nb - 1));
  }
  __syncthreads();
  if (!last || t != 0) return;
  __threadfence();
  double a = 0, b = 0, c = 0, m = -INFINITY;
  for (int k = 0; k < nb; ++k) {
    const volatile double* p = ws + 4 * k;
    a += p[0]; b += p[1]; c += p[2]; m = fmax(m, p[3]);
  }
  sums3[0] = a; sums3[1] = b; sums3[2] = c;
  if (count) count[0] = (double)N;
  if (max_out) max_out[0] = m;
  *counter = 0u;
}

// One workgroup of 1024 threads over the P parameters (P <= a few 10^4: a handful of elements per thread).
// scal: [0] ElocMean  [1] ElocMeanAbs  [2] ElocVar  [3] <E^2>  [4] max_p mean_p^2 / var_p  [5] max_p diag(S)_p  [6] floor = svdTol*[5]
__global__ __launch_bounds__(1024) void tdvp_moments_kernel(const double* __restrict__ packed, int P, int bordered, double* __restrict__ rowsum,
                                                          double* __restrict__ rowdot, const double* __restrict__ sums3, double n_total,
                                                          const double* __restrict__ c, const double* __restrict__ cg,
                                                          const double* __restrict__ cE_dev, const double* __restrict__ mask,
                                                          double* __restrict__ F, double* __restrict__ mean, double* __restrict__ diagS,
                                                          double* __restrict__ scal, double svdTol, double* __restrict__ border_row,
                                                          double border_pivot) {
  __shared__ double shr[16], shd[16];
  const int t = threadIdx.x;
  const double inv_n = 1.0 / n_total;
  const double cE = cE_dev ? cE_dev[0] : 0.0;
  const double sumE = sums3[0];
  const double deltaE = sumE * inv_n - cE;
  const int nb = (P + (bordered ? 2 : 0) + TILE - 1) / TILE;                   // block rows of the Gram that was run
  const double* __restrict__ lastrow = packed + (int64_t)tri_count(nb - 1) * (TILE * TILE);   // tiles (nb-1, 0..nb-1)
  const int r = P - (nb - 1) * TILE;                                            // row of E - cE inside the last tile row; r + 1: the ones
  double ratio = 0.0, dmax = 0.0;
  for (int p = t; p < P; p += 1024) {
    const double cp = c ? c[p] : 0.0;
    double rs, rd;
    if (bordered) {
      // the Gram's last tile row carries sum_n O_p (E_n - cE) and sum_n O_p; shifted sums as vmcpde_gram_finalize expects them
      const double* __restrict__ tile = lastrow + (int64_t)(p / TILE) * (TILE * TILE) + (p % TILE);
      rs = tile[(r + 1) * TILE] - cp * n_total;
      rd = tile[r * TILE] - cp * (sumE - n_total * cE);
      rowsum[p] = rs;
      rowdot[p] = rd;
    } else {
      rs = rowsum[p];
      rd = rowdot[p];
    }
    const double delta = rs * inv_n;
    const double f = rd * inv_n - delta * deltaE;                               // tdvp.py:45
    const double m = cp + delta;                                                // tdvp.py:41
    F[p] = f;
    mean[p] = m;
    const int ti = p / TILE, q = p % TILE;
    const double raw = packed[((int64_t)tri_count(ti) + ti) * (TILE * TILE) + q * TILE + q];
    const double d = delta + cp - (cg ? cg[p] : 0.0);
    const double var = raw * inv_n - d * d;                                     // diag of tdvp.py:46
    diagS[p] = var;
    const double w = mask ? mask[p] : 1.0;
    if (w != 0.0) {
      dmax = fmax(dmax, var);
      const double m2 = m * m;
      ratio = fmax(ratio, var > 0.0 ? m2 / var : (m2 > 0.0 ? INFINITY : 0.0));
    }
    if (border_row) border_row[p] = f;
  }
  ratio = wave_max(ratio);
  dmax = wave_max(dmax);
  if ((t & 63) == 0) { shr[t >> 6] = ratio; shd[t >> 6] = dmax; }
  __syncthreads();
  if (t == 0) {
    for (int k = 1; k < 16; ++k) { ratio = fmax(ratio, shr[k]); dmax = fmax(dmax, shd[k]); }
    const double em = sumE * inv_n;
    scal[0] = em;                                   // tdvp.py:37
    scal[1] = sums3[2] * inv_n;                     // tdvp.py:38
    scal[2] = sums3[1] * inv_n - em * em;           // tdvp.py:39
    scal[3] = sums3[1] * inv_n;
    scal[4] = ratio;
    scal[5] = dmax;
    scal[6] = svdTol * dmax;
    scal[7] = 0.0;
    if (border_row) border_row[P] = border_pivot;
  }
}

// out: [0] residual |S u - F| / |F| (tdvp.py:94, S = S0 + shift diag(S0) (+ floor I on the Cholesky path))
//      [1] tdvp_error 1 + (u^T S0 u - 2 F.u) / <E^2> (tdvp.py:93)   [2] pinned-pivot count of the factorisation (0 if none)
__global__ __launch_bounds__(1024) void tdvp_diagnostics_kernel(const double* __restrict__ u, const double* __restrict__ S0u,
                                                              const double* __restrict__ F, const double* __restrict__ diagS, int P,
                                                              double shift, const double* __restrict__ floor_dev,
                                                              const double* __restrict__ scal, const int* __restrict__ info,
                                                              double* __restrict__ out) {
  __shared__ double sh[4][16];
  const int t = threadIdx.x;
  const double fl = floor_dev ? floor_dev[0] : 0.0;
  double r2 = 0, f2 = 0, usu = 0, fu = 0;
  for (int p = t; p < P; p += 1024) {
    const double up = u[p], s0u = S0u[p], f = F[p];
    const double su = s0u + (shift * diagS[p] + fl) * up;
    r2 += (su - f) * (su - f);
    f2 += f * f;
    usu += up * s0u;
    fu += f * up;
  }
  r2 = wave_sum(r2); f2 = wave_sum(f2); usu = wave_sum(usu); fu = wave_sum(fu);
  if ((t & 63) == 0) { sh[0][t >> 6] = r2; sh[1][t >> 6] = f2; sh[2][t >> 6] = usu; sh[3][t >> 6] = fu; }
  __syncthreads();
  if (t == 0) {
    for (int k = 1; k < 16; ++k) { r2 += sh[0][k]; f2 += sh[1][k]; usu += sh[2][k]; fu += sh[3][k]; }
    out[0] = sqrt(r2) / sqrt(f2);
    out[1] = 1.0 + (usu - 2.0 * fu) / scal[3];
    out[2] = info ? (double)info[0] : 0.0;
    out[3] = 0.0;
  }
}


// ---- observables of a sample set (tdvp.py:143-150) in two launches -----------------------------------------------------------
// phase 0: sum x_j, sum log p, max Eloc  ->  out: mean[d], entropy, max;   phase 1: central sums about that mean ->  covariance
// (1/n), central moments 3..6.  A workgroup stages OBS_CHUNK samples in LDS and thread q accumulates quantity q over them (at
// most OBS_QMAX quantities, two per thread); block partials go to ws[block][OBS_QMAX] and the last block to arrive adds them in
// block order (deterministic), as in tdvp_prepare_kernel.  The reference runs ~40 eager array ops here; in torch they were 0.4 ms
// of launches per call, a quarter of a step of the reference's own problem sizes.
constexpr int OBS_DMAX = 16, OBS_CHUNK = 256, OBS_THREADS = 256, OBS_BLOCKS_MAX = 64;
constexpr int OBS_QMAX = OBS_DMAX * OBS_DMAX + 4 * OBS_DMAX;            // 320

template <int PHASE>
__global__ __launch_bounds__(OBS_THREADS) void observables_kernel(const double* __restrict__ x, int64_t n, int d, const double* __restrict__ logp,
                                                                  const double* __restrict__ eloc, int64_t n_eloc, double* __restrict__ out,
                                                                  double* __restrict__ ws) {
  __shared__ double xs[OBS_CHUNK * (OBS_DMAX + 1)];     // per sample: d coordinates (centred in phase 1), then log p (phase 0)
  __shared__ double mean[OBS_DMAX];
  __shared__ double emax[OBS_THREADS / 64];
  __shared__ bool last;
  const int t = threadIdx.x, nb = gridDim.x;
  const int w = d + (PHASE == 0 ? 1 : 0);                // doubles per staged sample
  const int nq = (PHASE == 0) ? d + 2 : d * d + 4 * d;
  if (PHASE == 1) {
    if (t < d) mean[t] = out[t];
    __syncthreads();
  }
  const int64_t per = (n + nb - 1) / nb;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < n) ? (lo + per) : n;
  double acc[2] = {0.0, 0.0};
  if (PHASE == 0) {
    // max Eloc (tdvp.py:150) over ITS OWN n_eloc entries: with nSamplesObs > nSamplesTDVP (tdvp.py:130-134) the samples and log p are
    // re-drawn at nSamplesObs while Eloc keeps the nSamplesTDVP entries of the solve
    double mx = -INFINITY;
    if (eloc) {
      const int64_t per_e = (n_eloc + nb - 1) / nb;
      const int64_t lo_e = (int64_t)blockIdx.x * per_e, hi_e = (lo_e + per_e < n_eloc) ? (lo_e + per_e) : n_eloc;
      for (int64_t i = lo_e + t; i < hi_e; i += OBS_THREADS) mx = fmax(mx, eloc[i]);
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) emax[t >> 6] = mx;
    __syncthreads();
    mx = emax[0];
    for (int k = 1; k < OBS_THREADS / 64; ++k) mx = fmax(mx, emax[k]);
    if (t == d + 1) acc[0] = mx;           // (d <= 16: quantity d + 1 always belongs to the first pass of thread d + 1)
  }
  for (int64_t c0 = lo; c0 < hi; c0 += OBS_CHUNK) {
    const int m = (int)((hi - c0 < OBS_CHUNK) ? (hi - c0) : OBS_CHUNK);
    for (int e = t; e < m * d; e += OBS_THREADS) {
      const int sidx = e / d, j = e - sidx * d;
      const double v = x[c0 * d + e];
      xs[sidx * w + j] = (PHASE == 1) ? v - mean[j] : v;
    }
    if (PHASE == 0)
      for (int sidx = t; sidx < m; sidx += OBS_THREADS) xs[sidx * w + d] = logp ? logp[c0 + sidx] : 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = t + k * OBS_THREADS;
      if (q >= nq) break;
      double a = acc[k];
      if (PHASE == 0) {
        if (q <= d) for (int sidx = 0; sidx < m; ++sidx) a += xs[sidx * w + q];
      } else if (q < d * d) {
        const int i = q / d, j = q - i * d;
        for (int sidx = 0; sidx < m; ++sidx) a = fma(xs[sidx * w + i], xs[sidx * w + j], a);
      } else {
        const int pw = 3 + (q - d * d) / d, j = (q - d * d) % d;
        for (int sidx = 0; sidx < m; ++sidx) {
          const double v = xs[sidx * w + j], v2 = v * v;
          a += (pw == 3) ? v2 * v : (pw == 4) ? v2 * v2 : (pw == 5) ? v2 * v2 * v : v2 * v2 * v2;
        }
      }
      acc[k] = a;
    }
    __syncthreads();
  }
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws + (int64_t)OBS_BLOCKS_MAX * OBS_QMAX);
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (t + k * OBS_THREADS < nq) ws[(int64_t)blockIdx.x * OBS_QMAX + t + k * OBS_THREADS] = acc[k];
  __syncthreads();
  if (t == 0) {
    __threadfence();
    last = (atomicAdd(counter, 1u) == (unsigned int)(nb - 1));
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const double inv_n = 1.0 / (double)n;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = t + k * OBS_THREADS;
    if (q >= nq) break;
    const bool is_max = (PHASE == 0 && q == d + 1);
    double a = is_max ? -INFINITY : 0.0;
    for (int b = 0; b < nb; ++b) {
      const double v = const_cast<const volatile double*>(ws)[(int64_t)b * OBS_QMAX + q];
      a = is_max ? fmax(a, v) : a + v;
    }
    // layout of out: mean[d] | covar[d*d] | x3[d] x4[d] x5[d] x6[d] | entropy | max Eloc
    if (PHASE == 0) {
      if (q < d) out[q] = a * inv_n;
      else if (q == d) out[d + d * d + 4 * d] = -a * inv_n;            // tdvp.py:148
      else out[d + d * d + 4 * d + 1] = a;                             // tdvp.py:150
    } else {
      out[d + q] = a * inv_n;
    }
  }
  if (t == 0) *counter = 0u;
}

}  // namespace

extern "C" int64_t vmcpde_tdvp_prepare_workspace_bytes(void) { return (int64_t)(4 * PREP_BLOCKS_MAX + 2) * (int64_t)sizeof(double); }

extern "C" int vmcpde_tdvp_prepare(const double* eloc, int64_t n_samples, const double* cE_dev, double* row_e, double* row_one,
                                   double* sums3, double* count, double* max_out, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(eloc && sums3 && workspace, "vmcpde_tdvp_prepare: null argument");
  VMC_REQUIRE(n_samples >= 1, "empty sample set");
  VMC_REQUIRE((row_e == nullptr) == (row_one == nullptr), "row_e and row_one go together");
  VMC_REQUIRE(workspace_bytes >= vmcpde_tdvp_prepare_workspace_bytes(), "prepare workspace too small");
  int blocks = (int)((n_samples + 2047) / 2048);      // (more, smaller blocks are slower: every block pays a release fence)
  if (blocks > PREP_BLOCKS_MAX) blocks = PREP_BLOCKS_MAX;
  hipLaunchKernelGGL(tdvp_prepare_kernel, dim3(blocks), dim3(PREP_THREADS), 0, (hipStream_t)stream, eloc, n_samples, cE_dev, row_e, row_one,
                     sums3, count, max_out, static_cast<double*>(workspace));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_tdvp_moments(const double* packed, int32_t P, int32_t bordered, double* rowsum, double* rowdot, const double* sums3,
                                   double n_total, const double* c, const double* c_gram, const double* cE_dev, const double* mask,
                                   double* F, double* mean, double* diagS, double* scal, double svdTol, double* border_row,
                                   double border_pivot, void* stream) {
  VMC_REQUIRE(packed && rowsum && rowdot && sums3 && F && mean && diagS && scal, "vmcpde_tdvp_moments: null argument");
  VMC_REQUIRE(P >= 1 && n_total > 0, "bad shape");
  VMC_REQUIRE(!bordered || c_gram == nullptr, "the appended-row moments belong to the uncentred Gram");
  hipLaunchKernelGGL(tdvp_moments_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, packed, P, bordered, rowsum, rowdot, sums3, n_total, c,
                     c_gram, cE_dev, mask, F, mean, diagS, scal, svdTol, border_row, border_pivot);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_tdvp_diagnostics(const double* u, const double* S0u, const double* F, const double* diagS, int32_t P,
                                       double diagonal_shift, const double* floor_dev, const double* scal, const int32_t* info,
                                       double* out, void* stream) {
  VMC_REQUIRE(u && S0u && F && diagS && scal && out && P >= 1, "vmcpde_tdvp_diagnostics: bad argument");
  hipLaunchKernelGGL(tdvp_diagnostics_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, u, S0u, F, diagS, P, diagonal_shift, floor_dev,
                     scal, info, out);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t vmcpde_observables_workspace_bytes(void) { return ((int64_t)OBS_BLOCKS_MAX * OBS_QMAX + 2) * (int64_t)sizeof(double); }

extern "C" int vmcpde_observables(const double* x, int64_t n_samples, int32_t dim, const double* logp, const double* eloc, int64_t n_eloc,
                                  double* out, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(x && out && workspace, "vmcpde_observables: null argument");
  VMC_REQUIRE(n_samples >= 1, "empty sample set");
  VMC_REQUIRE(eloc == nullptr || n_eloc >= 1, "vmcpde_observables: eloc given with n_eloc < 1");
  VMC_REQUIRE(dim >= 1 && dim <= OBS_DMAX, "vmcpde_observables: 1 <= dim <= 16");
  VMC_REQUIRE(workspace_bytes >= vmcpde_observables_workspace_bytes(), "observables workspace too small");
  int blocks = (int)((n_samples + 4 * OBS_CHUNK - 1) / (4 * OBS_CHUNK));
  if (blocks > OBS_BLOCKS_MAX) blocks = OBS_BLOCKS_MAX;
  hipLaunchKernelGGL(observables_kernel<0>, dim3(blocks), dim3(OBS_THREADS), 0, (hipStream_t)stream, x, n_samples, dim, logp, eloc, n_eloc, out,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(observables_kernel<1>, dim3(blocks), dim3(OBS_THREADS), 0, (hipStream_t)stream, x, n_samples, dim, logp, eloc, n_eloc, out,
                     static_cast<double*>(workspace));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}
