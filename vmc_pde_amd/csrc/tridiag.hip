// Householder tridiagonalisation of the r x r core of the rank-revealing spectral solve in ONE workgroup, and the back-transformation of the
// tridiagonal problem's eigenvectors (reference: tdvp.py:57-64, `eigh`; here the r x r matrix Q1^T S Q1 of csrc/eig.hip's scheme).
//
// Why: a library `syevd` on the 206 x 206 core of the headline workload takes 4.3 ms, 2.9 ms of it the tridiagonalisation -- a chain of
// ~10 small launches per column (hemv, dot, axpy, syr2, larfg ...: tools/diag/stedc_probe.py, eigh_core_trace.py).  The reduction is r - 2
// dependent steps of O(r^2) work each; with the lower triangle of the matrix held in the REGISTERS of one 512-thread workgroup a step is a
// few LDS round trips and four workgroup barriers instead of ten kernel boundaries.
//
// Layout (n <= 256): wave w owns the rows i = w + 8 s (s = 0 .. 31), lane l the columns j = l + 64 c (c = 0 .. 3).  The six 64 x 64 blocks
// strictly below the block diagonal live in REGISTERS (a lane keeps a(i, j) for the blocks c < s / 8: 48 doubles, statically indexed -- the
// step loop is instantiated once per 64-column block of the pivot column, so the pivot column's block index is a compile-time constant too),
// the four diagonal blocks in LDS (128 KB) with BOTH their triangles (the symmetric update keeps them consistent), so that no entry ever
// needs a mask and the pivot column inside its diagonal block is simply a row of it.  (All ten blocks in registers, 80 doubles a lane at
// two waves per SIMD, left no room for a step's temporaries: hipcc spilled ~320 registers into the dependent chain, 8.4 us per step.)
// One step k (LAPACK's dsytd2, lower):
//   x = a(k+1:, k)                          pivot column -> LDS (its owner lane of every wave), every wave forms the reflector redundantly:
//   beta = -sign(x0) |x|, tau = (beta - x0) / beta, v = x / (x0 - beta) with v0 = 1          (identical bits in all waves)
//   p = tau A v:  row parts (the stored entries of row i) summed over the lanes of the owning wave by a transposing permlane / DPP
//                 network (32 rows in 175 instructions), column parts (entries left of the diagonal block) lane-local, summed over the
//                 eight waves through LDS in a fixed order
//   w = p - (tau/2)(p.v) v,   A -= v w^T + w v^T   (registers; v_i, w_i of a row are LDS broadcasts)
// No atomics, no data-dependent summation order: the result is a pure function of the input bits (the replicated solve must give every
// rank the same update).  Output: d, e (the tridiagonal matrix), tau and the reflectors v_k (row k of V, zeros before k+1, v_k[k+1] = 1);
// Q = H_0 H_1 ... H_{n-2}, H_k = I - tau_k v_k v_k^T, Q^T A Q = tridiag(d, e).
#include "common.hpp"
#include "wave_ops.hpp"

namespace vmcpde {
namespace {

constexpr int TD_WAVES = 8, TD_THREADS = 64 * TD_WAVES, TD_MAXN = 256, TD_ROWS = TD_MAXN / TD_WAVES /* row slots per wave */;

// registers: the blocks strictly left of the diagonal block of row slot s (row i = wave + 8 s): s / 8 of them
__host__ __device__ constexpr int td_nreg(int s) { return s / 8; }
__host__ __device__ constexpr int td_off(int s) { return 8 * ((s / 8) * (s / 8 - 1) / 2) + (s % 8) * (s / 8); }   // sum_{s' < s} td_nreg(s')
constexpr int TD_REGS = td_off(TD_ROWS - 1) + td_nreg(TD_ROWS - 1);
static_assert(TD_REGS == 48, "register layout");

// ---- row sums across the lanes of a wave, 32 rows at a time ------------------------------------------------------------------------
// A lane holds one partial sum for each of 32 rows; the 32 totals are wanted.  Reducing the rows one by one costs six DPP steps each (576
// instructions); a transposing network halves the number of registers at every level instead -- two registers are combined so that half
// of the lanes go on with one row and half with the other: 175 instructions.  Levels: lanes 32 apart (v_permlane32_swap), 16 apart
// (v_permlane16_swap), then within rows of 16 lanes by mirrored DPP reads written bank-wise, then within quads.  At the end lane l holds
// the total of input row bitreverse5(l >> 1) (both lanes of a pair the same value).
__device__ __forceinline__ void td_swap32(double& a, double& b) {
  const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)(x & 0xffffffffll), (unsigned)(y & 0xffffffffll), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
  a = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
  b = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void td_swap16(double& a, double& b) {
  const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)(x & 0xffffffffll), (unsigned)(y & 0xffffffffll), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
  a = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
  b = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}
// lanes of the banks (groups of 4 lanes within a row of 16) in BANKS read `src` through the DPP pattern CTRL, the others keep `old`
template <int CTRL, int BANKS>
__device__ __forceinline__ double td_dpp_banks(double old, double src) {
  const long long o = __double_as_longlong(old), b = __double_as_longlong(src);
  const int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffll), (int)(b & 0xffffffffll), CTRL, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, 0xf, BANKS, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_IDENT = 0xe4, DPP_ROW_MIRROR = 0x140, DPP_HALF_MIRROR = 0x141;

// eight rows -> one register: 8 partial sums of each row, row m of the group in the lanes with (l >> 3 & 1, l >> 4 & 1, l >> 5) = bits of m
__device__ __forceinline__ double td_reduce8(double (&x)[8]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    td_swap32(x[2 * m], x[2 * m + 1]);
    x[m] = x[2 * m] + x[2 * m + 1];                       // (x[m] with m < 2 m: already consumed)
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    td_swap16(x[2 * m], x[2 * m + 1]);
    x[m] = x[2 * m] + x[2 * m + 1];
  }
  const double base = td_dpp_banks<DPP_IDENT, 0xc>(x[0], x[1]);                                   // lanes 8-15 of every row of 16 go on with x[1]
  const double other = td_dpp_banks<DPP_ROW_MIRROR, 0xc>(dpp_move<DPP_ROW_MIRROR>(x[0]), x[1]);  // partner lane 15 - l of the same source
  return base + other;
}

// four such registers (32 rows) -> lane l holds the total of row bitreverse5(l >> 1)
__device__ __forceinline__ double td_reduce32_finish(double u0, double u1, double u2, double u3, int lane) {
  const double v0 = td_dpp_banks<DPP_IDENT, 0xa>(u0, u1) + td_dpp_banks<DPP_HALF_MIRROR, 0xa>(dpp_move<DPP_HALF_MIRROR>(u0), u1);
  const double v1 = td_dpp_banks<DPP_IDENT, 0xa>(u2, u3) + td_dpp_banks<DPP_HALF_MIRROR, 0xa>(dpp_move<DPP_HALF_MIRROR>(u2), u3);
  const double t0 = v0 + dpp_move<0x4e>(v0), t1 = v1 + dpp_move<0x4e>(v1);      // quad_perm [2,3,0,1]
  const double w = (lane & 2) ? t1 : t0;
  return w + dpp_move<0xb1>(w);                                                  // quad_perm [1,0,3,2]
}

struct TdShared {
  double dg[4][64][64];                                  // the four 64 x 64 diagonal blocks, both triangles (row r of block b: row 64 b + r of the matrix)
  double xcol[TD_MAXN], vL[TD_MAXN], wL[TD_MAXN], prow[TD_MAXN];
  double pcol[TD_WAVES][TD_MAXN];
  double dump[64];
};
static_assert(sizeof(TdShared) <= 160 * 1024, "LDS");

// the steps whose pivot column lies in column block CB of the frame (k = max(64 CB, off) .. 64 CB + 63)
template <int CB>
__device__ __forceinline__ void td_steps(double (&a)[TD_REGS], TdShared& sh, int off, int wave_in, int lane_in, double* __restrict__ D,
                                         double* __restrict__ E, double* __restrict__ tau_out, double* __restrict__ V, int64_t ldv) {
  // (indices are those of the 256 x 256 frame; the n x n matrix sits in its bottom right corner, rows and columns off = 256 - n ...)
  for (int k = (64 * CB > off) ? 64 * CB : off; k < 64 * CB + 64; ++k) {
    // opaque per step: otherwise LICM hoists the row indices wave + 8 s and the lane-only values out of the step loop and keeps them alive
    // across it (SGPRs spilled to VGPR lanes, VGPRs to scratch)
    int wave = wave_in, lane = lane_in;
    asm volatile("" : "+s"(wave));
    asm volatile("" : "+v"(lane));
    const int lk = k - 64 * CB;
    // (A) the part of column k below its diagonal block sits in registers (lane lk of every wave, block CB of the rows of the later bands)
    // -> xcol; the other lanes write to a dump line (an address select instead of exec-mask regions).  The part inside the diagonal
    // block is ROW lk of that block (both triangles are kept) and is read in place.
    {
      const bool mine = lane == lk;
#pragma unroll
      for (int s = 8 * (CB + 1); s < TD_ROWS; ++s) {
        double* dst = mine ? &sh.xcol[wave + 8 * s] : &sh.dump[lane];
        *dst = a[td_off(s) + CB];
      }
    }
    __syncthreads();                                     // (also: the owner's update of row lk of the diagonal block, step k - 1, is visible)
    if (k == TD_MAXN - 1) {                              // the last diagonal entry; no reflector
      if (wave == 0 && lane == 0) {
        D[k - off] = sh.dg[CB][lk][lk];
        E[k - off] = 0.0;
        tau_out[k - off] = 0.0;
      }
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (lane + 64 * c >= off) V[(int64_t)(k - off) * ldv + lane + 64 * c - off] = 0.0;
      }
      break;
    }
    // (B) the reflector, redundantly in every wave (same instructions on the same bits)
    double vc[4], xs[4];
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      double x = 0.0;
      if (c == CB) x = (j > k) ? sh.dg[CB][lk][lane] : 0.0;
      if (c > CB) x = sh.xcol[j];
      xs[c] = x;
      part += (j > k + 1) ? x * x : 0.0;
    }
    const double xnorm2 = wave_sum(part);
    const double* alpha_at = (lk < 63) ? &sh.dg[CB][lk][lk + 1] : &sh.xcol[k + 1];
    const double alpha = *alpha_at, dk = sh.dg[CB][lk][lk];
    double tau = 0.0, beta = alpha, scale = 0.0;
    if (xnorm2 > 0.0) {
      beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      vc[c] = (j == k + 1) ? 1.0 : ((j > k + 1) ? xs[c] * scale : 0.0);
    }
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = lane + 64 * c;
        sh.vL[j] = vc[c];
        if (j >= off) V[(int64_t)(k - off) * ldv + j - off] = vc[c];
      }
      if (lane == 0) {
        D[k - off] = dk;
        E[k - off] = beta;
        tau_out[k - off] = tau;
      }
    }
    __syncthreads();
    // (tau = 0, i.e. H_k = I: the arithmetic below then changes nothing -- p = 0, w = 0 -- and runs anyway.)
    // (C) p = A v.  Row i's stored entries -- its register blocks (columns left of its diagonal block; the blocks of finished column
    // bands, c < CB, are skipped statically: v is zero there) and its WHOLE row of the diagonal block (LDS) -- give its row part, summed
    // over the lanes by the transposing network; the register blocks also give the column parts of their columns (lane-local, then
    // over the waves through LDS in a fixed order).  Rows at or above the pivot have v_i = 0: no column contribution, and their row sums
    // are never read; the row groups of finished bands are skipped statically.
    double colacc[4] = {0.0, 0.0, 0.0, 0.0};
    double u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = CB; g < 4; ++g) {
      double r8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int s = 8 * g + q, i = wave + 8 * s;
        const double vi = sh.vL[i];
        double rowacc = sh.dg[g][wave + 8 * q][lane] * vc[g];
#pragma unroll
        for (int c = CB; c < td_nreg(s); ++c) {
          const double av = a[td_off(s) + c];
          rowacc = fma(av, vc[c], rowacc);
          colacc[c] = fma(av, vi, colacc[c]);
        }
        r8[q] = rowacc;
      }
      u[g] = td_reduce8(r8);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const double tot = td_reduce32_finish(u[0], u[1], u[2], u[3], lane);
      const int srow = (int)(__builtin_bitreverse32((unsigned)(lane >> 1)) >> 27);
      double* dst = (lane & 1) ? &sh.dump[lane] : &sh.prow[wave + 8 * srow];
      *dst = tot;
    }
#pragma unroll
    for (int c = CB; c < 3; ++c) sh.pcol[wave][lane + 64 * c] = colacc[c];      // (block 3 has no rows below it)
    __syncthreads();
    // (D) w = tau p - (tau^2 / 2)(p . v) v, redundantly per wave for its lanes' columns
    double wc[4] = {0.0, 0.0, 0.0, 0.0};
    double dpart = 0.0;
#pragma unroll
    for (int c = CB; c < 4; ++c) {
      const int j = lane + 64 * c;
      double p = 0.0;
      if (j > k) {
        p = sh.prow[j];
        if (c < 3) {
#pragma unroll
          for (int w8 = 0; w8 < TD_WAVES; ++w8) p += sh.pcol[w8][j];
        }
        p *= tau;
      }
      wc[c] = p;
      dpart = fma(p, vc[c], dpart);
    }
    const double half = -0.5 * tau * wave_sum(dpart);
#pragma unroll
    for (int c = CB; c < 4; ++c) wc[c] = fma(half, vc[c], wc[c]);
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) sh.wL[lane + 64 * c] = wc[c];
    }
    __syncthreads();
    // (E) A -= v w^T + w v^T on every stored entry of the live bands (rows at or above the pivot: v_i = w_i = 0, unchanged; columns likewise)
#pragma unroll
    for (int s = 8 * CB; s < TD_ROWS; ++s) {
      const int i = wave + 8 * s;
      const double vi = sh.vL[i], wi = sh.wL[i];
      double* drow = &sh.dg[s / 8][wave + 8 * (s % 8)][lane];
      *drow = fma(-vi, wc[s / 8], fma(-wi, vc[s / 8], *drow));
#pragma unroll
      for (int c = CB; c < td_nreg(s); ++c) a[td_off(s) + c] = fma(-vi, wc[c], fma(-wi, vc[c], a[td_off(s) + c]));
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // (four rows at a time: all the broadcast reads hoisted at once cost more registers than there are)
    }
  }
}

__global__ __launch_bounds__(TD_THREADS) void sytrd_small_kernel(const double* __restrict__ T, int64_t ldt, int n, double* __restrict__ D,
                                                                 double* __restrict__ E, double* __restrict__ tau, double* __restrict__ V,
                                                                 int64_t ldv) {
  extern __shared__ double td_lds[];
  TdShared& sh = *reinterpret_cast<TdShared*>(td_lds);
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  double a[TD_REGS];
  // the matrix goes to the BOTTOM RIGHT corner of the 256 x 256 frame (rows and columns off .. 255): the leading rows and columns are
  // zero and count as finished, so the bands they fill are skipped by the static band logic of the step loop instead of being swept as
  // padding in every step (n = 206: a quarter less work, n = 128: less than half)
  const int off = TD_MAXN - n;
#pragma unroll
  for (int s = 0; s < TD_ROWS; ++s) {
    const int i = wave + 8 * s;
#pragma unroll
    for (int c = 0; c < td_nreg(s); ++c) {
      const int j = lane + 64 * c;
      a[td_off(s) + c] = (i >= off && j >= off) ? T[(int64_t)(i - off) * ldt + (j - off)] : 0.0;
    }
    // the row of the diagonal block: only the lower triangle of T is read, the upper part comes from its mirror image
    const int j = 64 * (s / 8) + lane;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    sh.dg[s / 8][wave + 8 * (s % 8)][lane] = (lo >= off) ? T[(int64_t)(hi - off) * ldt + (lo - off)] : 0.0;
  }
  __syncthreads();
  if (off < 64) td_steps<0>(a, sh, off, wave, lane, D, E, tau, V, ldv);
  if (off < 128) td_steps<1>(a, sh, off, wave, lane, D, E, tau, V, ldv);
  if (off < 192) td_steps<2>(a, sh, off, wave, lane, D, E, tau, V, ldv);
  td_steps<3>(a, sh, off, wave, lane, D, E, tau, V, ldv);
}

// Zt[m][:] <- Q Zt[m][:] = H_0 H_1 ... H_{n-2} z for every row m of Zt (the eigenvectors of the tridiagonal matrix, one per row -- the
// column-major matrix a LAPACK-style solver returns): one wave per vector, the reflectors streamed from V (L2 resident).
template <int NCH>      // 64-column chunks a lane holds: 4 (n <= 256) or 6 (n <= 384)
__global__ __launch_bounds__(256) void apply_reflectors_kernel(const double* __restrict__ V, int64_t ldv, const double* __restrict__ tau, int n,
                                                               double* __restrict__ Zt, int64_t ldz, int nvec) {
  const int lane = (int)threadIdx.x & 63;
  const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (m >= nvec) return;
  double z[NCH], vk[NCH], vn[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) z[c] = (lane + 64 * c < n) ? Zt[(int64_t)m * ldz + lane + 64 * c] : 0.0;
  if (n >= 2) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) vn[c] = (lane + 64 * c < n) ? V[(int64_t)(n - 2) * ldv + lane + 64 * c] : 0.0;
  }
  for (int k = n - 2; k >= 0; --k) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) vk[c] = vn[c];
    if (k > 0) {                                          // the next reflector's loads fly under this one's reduction
#pragma unroll
      for (int c = 0; c < NCH; ++c) vn[c] = (lane + 64 * c < n) ? V[(int64_t)(k - 1) * ldv + lane + 64 * c] : 0.0;
    }
    const double t = tau[k];
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) part = fma(z[c], vk[c], part);
    const double f = -t * wave_sum(part);
#pragma unroll
    for (int c = 0; c < NCH; ++c) z[c] = fma(f, vk[c], z[c]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < n) Zt[(int64_t)m * ldz + lane + 64 * c] = z[c];
}

// ---- ranks between 257 and 384: the leading n - 256 Householder steps on a full symmetric copy, the rest by sytrd_small_kernel --------------
// The single-workgroup kernel holds a 256-frame (its four diagonal blocks fill the LDS); a core of rank 263 (BASELINE C3) needs seven more
// steps than that.  Those steps (LAPACK's dsytd2 on the leading columns) run as two launches each on W = both triangles of T:
//   A  every workgroup forms the reflector of column k from ROW k of W (contiguous; the same bits everywhere), workgroup 0 publishes it
//      (D, E, tau, row k of V); p_i = sum_j W[i][j] v_j for the workgroup's rows -- whole rows, so no cross-workgroup sum;
//   B  every workgroup forms p.v (all of p, a fixed order: the same bits everywhere), w = tau p - (tau^2 / 2)(p.v) v, and updates its rows.
// The trailing 256 x 256 block of W is then the one-workgroup kernel's input.  No atomics; a pure function of the input bits.
constexpr int TM_MAXN = 384, TM_ROWS = 16 /* rows of W per workgroup */, TM_THREADS = 256;

__global__ __launch_bounds__(256) void sytd2_symmetrise_kernel(const double* __restrict__ T, int64_t ldt, int n, double* __restrict__ W) {
  const int i = (int)blockIdx.x, t = (int)threadIdx.x;
  for (int j = t; j < n; j += 256) W[(int64_t)i * n + j] = (j <= i) ? T[(int64_t)i * ldt + j] : T[(int64_t)j * ldt + i];
}

// the reflector of step k from row k of W (entries j > k), by all 256 threads of a workgroup: v -> sh_v[0 .. n), returns (tau, beta)
__device__ __forceinline__ void tm_reflector(const double* __restrict__ W, int n, int k, double* __restrict__ sh_v, double* __restrict__ sh_red, double& tau,
                                             double& beta) {
  const int t = (int)threadIdx.x;
  const double* __restrict__ row = W + (int64_t)k * n;
  double part = 0.0;
  for (int j = t; j < n; j += TM_THREADS) {
    const double x = (j > k) ? row[j] : 0.0;
    sh_v[j] = x;
    part += (j > k + 1) ? x * x : 0.0;
  }
  part = wave_sum(part);
  if ((t & 63) == 0) sh_red[t >> 6] = part;
  __syncthreads();
  const double xnorm2 = (sh_red[0] + sh_red[1]) + (sh_red[2] + sh_red[3]);
  const double alpha = sh_v[k + 1];
  __syncthreads();                                        // (alpha is read before sh_v[k + 1] becomes 1)
  tau = 0.0;
  beta = alpha;
  double scale = 0.0;
  if (xnorm2 > 0.0) {
    beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int j = t; j < n; j += TM_THREADS) sh_v[j] = (j == k + 1) ? 1.0 : ((j > k + 1) ? sh_v[j] * scale : 0.0);
  __syncthreads();
}

__global__ __launch_bounds__(TM_THREADS) void sytd2_step_a_kernel(const double* __restrict__ W, int n, int k, double* __restrict__ D, double* __restrict__ E,
                                                                  double* __restrict__ tau_out, double* __restrict__ V, int64_t ldv, double* __restrict__ p) {
  __shared__ double sh_v[TM_MAXN];
  __shared__ double sh_red[4];
  double tau, beta;
  tm_reflector(W, n, k, sh_v, sh_red, tau, beta);
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  if (blockIdx.x == 0) {
    for (int j = t; j < n; j += TM_THREADS) V[(int64_t)k * ldv + j] = sh_v[j];
    if (t == 0) {
      D[k] = W[(int64_t)k * n + k];
      E[k] = beta;
      tau_out[k] = tau;
    }
  }
  // four rows per wave: p_i = tau * sum_j W[i][j] v_j (rows at or above the pivot: v_i = 0 and p_i is never used -- written as 0)
  for (int q = 0; q < TM_ROWS / 4; ++q) {
    const int i = (int)blockIdx.x * TM_ROWS + wave * (TM_ROWS / 4) + q;
    if (i >= n) continue;
    double s = 0.0;
    if (i > k)
      for (int j = lane; j < n; j += 64) s = fma(W[(int64_t)i * n + j], sh_v[j], s);
    s = wave_sum(s);
    if (lane == 0) p[i] = tau * s;
  }
}

__global__ __launch_bounds__(TM_THREADS) void sytd2_step_b_kernel(double* __restrict__ W, int n, int k, const double* __restrict__ V, int64_t ldv,
                                                                  const double* __restrict__ tau_in, const double* __restrict__ p) {
  __shared__ double sh_v[TM_MAXN], sh_w[TM_MAXN];
  __shared__ double sh_red[4];
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const double tau = tau_in[k];
  double part = 0.0;
  for (int j = t; j < n; j += TM_THREADS) {
    const double v = V[(int64_t)k * ldv + j], pj = p[j];
    sh_v[j] = v;
    sh_w[j] = pj;
    part = fma(pj, v, part);
  }
  part = wave_sum(part);
  if (lane == 0) sh_red[wave] = part;
  __syncthreads();
  const double half = -0.5 * tau * ((sh_red[0] + sh_red[1]) + (sh_red[2] + sh_red[3]));
  for (int j = t; j < n; j += TM_THREADS) sh_w[j] = fma(half, sh_v[j], sh_w[j]);
  __syncthreads();
  for (int q = 0; q < TM_ROWS / 4; ++q) {
    const int i = (int)blockIdx.x * TM_ROWS + wave * (TM_ROWS / 4) + q;
    if (i >= n || i <= k) continue;
    const double vi = sh_v[i], wi = sh_w[i];
    for (int j = lane; j < n; j += 64)
      if (j > k) W[(int64_t)i * n + j] = fma(-vi, sh_w[j], fma(-wi, sh_v[j], W[(int64_t)i * n + j]));
  }
}

}  // namespace
}  // namespace vmcpde

using namespace vmcpde;

extern "C" int32_t vmcpde_sytrd_small_max_n(void) { return TD_MAXN; }

extern "C" int vmcpde_sytrd_small(const double* T, int64_t ldt, int32_t n, double* D, double* E, double* tau, double* V, int64_t ldv, void* stream) {
  VMC_REQUIRE(T && D && E && tau && V, "vmcpde_sytrd_small: null argument");
  VMC_REQUIRE(n >= 1 && n <= TD_MAXN, "vmcpde_sytrd_small: n = %d outside [1, %d]", n, TD_MAXN);
  VMC_REQUIRE(ldt >= n && ldv >= n, "vmcpde_sytrd_small: leading dimensions below n");
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  {
    DeviceCtxLock lock;
    if (!ctx->attr_tridiag) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)sytrd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TdShared)));
      ctx->attr_tridiag = true;
    }
  }
  hipLaunchKernelGGL(sytrd_small_kernel, dim3(1), dim3(TD_THREADS), sizeof(TdShared), (hipStream_t)stream, T, ldt, n, D, E, tau, V, ldv);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_apply_reflectors(const double* V, int64_t ldv, const double* tau, int32_t n, double* Zt, int64_t ldz, int32_t nvec,
                                       void* stream) {
  VMC_REQUIRE(V && tau && Zt, "vmcpde_apply_reflectors: null argument");
  VMC_REQUIRE(n >= 1 && n <= TM_MAXN && nvec >= 1, "vmcpde_apply_reflectors: bad shape (n = %d, nvec = %d)", n, nvec);
  VMC_REQUIRE(ldv >= n && ldz >= n, "vmcpde_apply_reflectors: leading dimensions below n");
  if (n <= TD_MAXN)
    hipLaunchKernelGGL(apply_reflectors_kernel<4>, dim3((unsigned)((nvec + 3) / 4)), dim3(256), 0, (hipStream_t)stream, V, ldv, tau, n, Zt, ldz, nvec);
  else
    hipLaunchKernelGGL(apply_reflectors_kernel<6>, dim3((unsigned)((nvec + 3) / 4)), dim3(256), 0, (hipStream_t)stream, V, ldv, tau, n, Zt, ldz, nvec);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int32_t vmcpde_sytrd_medium_max_n(void) { return TM_MAXN; }

extern "C" int64_t vmcpde_sytrd_medium_workspace_bytes(int32_t n) {
  return (n <= TD_MAXN) ? 0 : ((int64_t)n * n + n) * (int64_t)sizeof(double);
}

extern "C" int vmcpde_sytrd_medium(const double* T, int64_t ldt, int32_t n, double* D, double* E, double* tau, double* V, int64_t ldv, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(n >= 1 && n <= TM_MAXN, "vmcpde_sytrd_medium: n = %d outside [1, %d]", n, TM_MAXN);
  if (n <= TD_MAXN) return vmcpde_sytrd_small(T, ldt, n, D, E, tau, V, ldv, stream);
  VMC_REQUIRE(T && D && E && tau && V && workspace, "vmcpde_sytrd_medium: null argument");
  VMC_REQUIRE(ldt >= n && ldv >= n, "vmcpde_sytrd_medium: leading dimensions below n");
  VMC_REQUIRE(workspace_bytes >= vmcpde_sytrd_medium_workspace_bytes(n), "vmcpde_sytrd_medium: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* W = static_cast<double*>(workspace);
  double* p = W + (int64_t)n * n;
  const int m = n - TD_MAXN;                                // leading steps taken here; the trailing 256 x 256 block goes to the one-workgroup kernel
  const unsigned grid = (unsigned)((n + TM_ROWS - 1) / TM_ROWS);
  hipLaunchKernelGGL(sytd2_symmetrise_kernel, dim3((unsigned)n), dim3(256), 0, st, T, ldt, n, W);
  for (int k = 0; k < m; ++k) {
    hipLaunchKernelGGL(sytd2_step_a_kernel, dim3(grid), dim3(TM_THREADS), 0, st, W, n, k, D, E, tau, V, ldv, p);
    hipLaunchKernelGGL(sytd2_step_b_kernel, dim3(grid), dim3(TM_THREADS), 0, st, W, n, k, V, ldv, tau, p);
  }
  VMC_CHECK_HIP(hipGetLastError());
  // rows m .. n-1 of V: the one-workgroup kernel writes columns m .. n-1 of them; the columns before are zeros of those reflectors
  VMC_CHECK_HIP(hipMemset2DAsync(V + (int64_t)m * ldv, (size_t)ldv * sizeof(double), 0, (size_t)m * sizeof(double), (size_t)TD_MAXN, st));
  return vmcpde_sytrd_small(W + (int64_t)m * n + m, n, TD_MAXN, D + m, E + m, tau + m, V + (int64_t)m * ldv + m, ldv, stream);
}
