// Rank-revealing pieces of the spectral solve (reference: tdvp.py:57-64, `np.linalg.eigh(S)` / `jnp.linalg.eigh(S)` on the full P x P
// matrix).  S = <J^T J> of a smooth low-dimensional model is numerically rank deficient: at the headline size (P = 4213) 238
// eigenvalues lie above 1e-16 of the largest, the other ~4000 are rounding noise around zero.  The full decomposition then needs no
// O(P^3) work:
//
//   1. diagonally pivoted Cholesky  S = L L^T + R,  stopped when the largest remaining diagonal entry is <= tol * max diag(S)
//      (R is positive semi-definite with trace <= (P - r) tol max diag: the same size as the backward error of a dense eigensolver);
//      r dependent steps, each one launch: column k = (S[:, p] - L[:, :k] L[p, :k]^T) / sqrt(d_p), d -= column^2, arg max of d.
//   2. Householder QR of the P x r factor (r dependent steps, each one launch: one workgroup per trailing column) -- column-wise
//      backward stable whatever the grading of L (its columns span 8 decades), and its compact WY form gives BOTH an orthonormal
//      basis Q1 of range(L) and one of the complement, Q2, as plain GEMMs (host side: vmc_pde_amd/lowrank.py).
//   3. Rayleigh-Ritz on range(L): eigh of the r x r matrix Q1^T S Q1, Ritz vectors Q1 W; V = [Q2 | Q1 W] with eigenvalue 0 for Q2.
//
// This file holds 1 and 2 (the dependent chains); the GEMMs, the triangular solve and the r x r decomposition of 3 are library
// calls made from the host.  L is kept TRANSPOSED, LT[k][i] = L[i][k] (row k = column k of L, contiguous): every access of the
// chains is then a contiguous run of a row.
#include <cstddef>
#include "common.hpp"
#include "wave_ops.hpp"

namespace vmcpde {
namespace {

constexpr int PC_ROWS = 64;                          // rows of S per workgroup (one per lane)
constexpr int PC_SLICES = 16;                        // waves per workgroup: wave s sums the earlier columns j = s, s + 16, ...
constexpr int PC_THREADS = PC_ROWS * PC_SLICES;      // 1024
constexpr double NEG_INF = -__builtin_huge_val();

struct PivWork {            // device-side bookkeeping of one factorisation (head of the workspace)
  double d0;                // largest diagonal entry of S: the reference of the stopping criterion
  double trace_left;        // sum of the positive remaining diagonal entries after the last completed step
  // blocked form only: the pivots of an earlier factorisation as preferred candidates (vmcpde_lowrank_hint)
  int hint_magic;           // HINT_MAGIC while a hint is armed (anything else: none -- an uninitialised workspace has no hint)
  int hint_n;               // entries of the hint list
  int hint_cur;             // entries before this one are pivots already or below the tolerance
  int steps;                // block steps of this factorisation that added columns
  // matrix-free form only: the columns of S at the first big_n hint rows, formed ahead of the factorisation in ONE pass over the
  // Jacobian (vmcpde_lowrank_columns); a step whose candidates all come from the list then needs no pass of its own
  int pool_steps;           // block steps of this factorisation that were NOT served from those columns
  int big_n;
  const double* big;        // [big_n][ldl], row j = column hint[j] of S; nullptr: none registered
  int skip[2];              // skip[1] != 0: the step just selected needs no columns formed (done, or served from `big`)
};
static_assert(sizeof(PivWork) <= 64, "PivWork must fit the workspace head");
constexpr int HINT_MAGIC = 0x48494e54;

// (value, index) arg max over a wave, ties to the lower index; every lane returns the result
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(idx, off);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

// d = diag(S); per-workgroup candidates (largest entry, its row, sum of the positive entries) into candidate buffer 0
__global__ __launch_bounds__(PC_ROWS) void pivchol_init_kernel(const double* __restrict__ S, int64_t lds, int P, double* __restrict__ d,
                                                               double* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                               double* __restrict__ cand_sum, int32_t* __restrict__ state, int two,
                                                               PivWork* __restrict__ work) {
  const int i = (int)blockIdx.x * PC_ROWS + (int)threadIdx.x;
  double mine = NEG_INF;
  if (i < P) {
    mine = S[(int64_t)i * lds + i];
    d[i] = mine;
  }
  double v = mine;
  int idx = i;
  double pos = (v > 0.0) ? v : 0.0;
  wave_argmax(v, idx);
  pos = wave_sum(pos);
  if (!two) {
    if (threadIdx.x == 0) {
      cand_val[blockIdx.x] = v;
      cand_idx[blockIdx.x] = idx;
      cand_sum[blockIdx.x] = pos;
    }
  } else {                       // the blocked factorisation keeps the two largest entries of every 64-row block
    double v2 = (i == idx) ? NEG_INF : mine;
    int idx2 = i;
    wave_argmax(v2, idx2);
    if (threadIdx.x == 0) {
      cand_val[2 * blockIdx.x] = v;
      cand_idx[2 * blockIdx.x] = idx;
      cand_sum[2 * blockIdx.x] = pos;
      cand_val[2 * blockIdx.x + 1] = v2;
      cand_idx[2 * blockIdx.x + 1] = idx2;
      cand_sum[2 * blockIdx.x + 1] = 0.0;
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    state[0] = 0;
    state[1] = 0;
    if (work) { work->hint_cur = 0; work->steps = 0; work->pool_steps = 0; work->big = nullptr; work->big_n = 0; work->skip[0] = 0; work->skip[1] = 0; }
  }
}

// One step of the pivoted factorisation.  state[0] = columns completed, state[1] = 1 once the stopping criterion has been met (every
// later launch returns at once).  The candidates of step s are read from buffer s & 1 and those for step s + 1 written to the other.
__global__ __launch_bounds__(PC_THREADS) void pivchol_step_kernel(const double* __restrict__ S, int64_t lds, int P, int step, double tol_rel,
                                                                  double* __restrict__ LT, int64_t ldl, double* __restrict__ d,
                                                                  double* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                                  double* __restrict__ cand_sum, int nwg, int32_t* __restrict__ piv,
                                                                  int32_t* __restrict__ state, PivWork* __restrict__ work) {
  __shared__ double part[PC_SLICES][PC_ROWS];
  __shared__ double s_dp, s_sum;
  __shared__ int s_p, s_done;
  const int t = threadIdx.x, lane = t & 63;
  const int slice = __builtin_amdgcn_readfirstlane(t >> 6);
  const double* cv = cand_val + (step & 1) * nwg;
  const int* ci = cand_idx + (step & 1) * nwg;
  const double* cs = cand_sum + (step & 1) * nwg;
  if (slice == 0) {
    double v = NEG_INF, sum = 0.0;
    int idx = 0x7fffffff;
    for (int q = lane; q < nwg; q += 64) {
      const double ov = cv[q];
      const int oi = ci[q];
      sum += cs[q];
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    wave_argmax(v, idx);
    sum = wave_sum(sum);
    // state[1]: set by an earlier launch, or by workgroup 0 of this one -- a workgroup that still reads 0 reaches the same verdict
    // below.  One lane reads it for the workgroup (the barriers that follow need a uniform decision).
    if (lane == 0) { s_dp = v; s_p = idx; s_sum = sum; s_done = state[1]; }
  }
  __syncthreads();
  if (s_done) return;
  const double dp = s_dp;
  const int p = s_p;
  const double d0 = (step == 0) ? dp : work->d0;
  if (!(dp > tol_rel * d0)) {    // also: NaN anywhere, nothing left (-inf), a non-positive matrix
    if (blockIdx.x == 0 && t == 0) {
      state[1] = 1;
      state[0] = step;
      work->trace_left = s_sum;
      if (step == 0) work->d0 = dp;
    }
    return;
  }
  if (blockIdx.x == 0 && t == 0) {
    if (step == 0) work->d0 = dp;
    state[0] = step + 1;
    piv[step] = p;
  }
  const int i = (int)blockIdx.x * PC_ROWS + lane;      // (< ldl: the pad columns of LT are kept at zero)
  double sum = 0.0;
#pragma unroll 4
  for (int j = slice; j < step; j += PC_SLICES) sum = fma(LT[(int64_t)j * ldl + i], LT[(int64_t)j * ldl + p], sum);
  part[slice][lane] = sum;
  __syncthreads();
  if (slice != 0) return;
  double total = 0.0;
#pragma unroll
  for (int s = 0; s < PC_SLICES; ++s) total += part[s][lane];
  const bool inside = i < P;
  const double di = inside ? d[i] : NEG_INF;
  const bool open = inside && di != NEG_INF;          // rows chosen as pivots earlier are finished: structural zeros of L from here on
  double col = 0.0;
  if (open) col = (S[(int64_t)p * lds + i] - total) / sqrt(dp);
  LT[(int64_t)step * ldl + i] = col;
  double nd = NEG_INF;
  if (open && i != p) nd = di - col * col;
  if (inside) d[i] = nd;
  int idx = i;
  double pos = (nd > 0.0) ? nd : 0.0;
  wave_argmax(nd, idx);
  pos = wave_sum(pos);
  if (lane == 0) {
    const int o = ((step + 1) & 1) * nwg + (int)blockIdx.x;
    cand_val[o] = nd;
    cand_idx[o] = idx;
    cand_sum[o] = pos;
  }
}

// sum of the positive remaining diagonal entries once the loop ended WITHOUT meeting the criterion (rank cap reached): the host reports it
__global__ void pivchol_trace_kernel(const double* __restrict__ cand_sum, int nwg, int steps_done, const int32_t* __restrict__ state,
                                     PivWork* __restrict__ work) {
  if (state[1]) return;
  double sum = 0.0;
  for (int q = threadIdx.x; q < nwg; q += 64) sum += cand_sum[(steps_done & 1) * nwg + q];
  sum = wave_sum(sum);
  if (threadIdx.x == 0) work->trace_left = sum;
}

// ---- the same factorisation, PB pivots per pair of launches ------------------------------------------------------------------
// A pivot order need not be the greedy one: ANY order gives S = L L^T + R with R >= 0, and stopping when no remaining diagonal entry
// exceeds tol * max diag bounds trace(R) as before.  A block step takes up to PB candidates C from the two largest remaining entries
// of every 64-row block (one wave picks the largest of those the previous step left), forms the panel
// A = S[:, C] - L L[C, :]^T (launch 1), factors the PB x PB block A[C, C] with pivoting INSIDE the block -- candidates whose remaining
// entry falls to <= tol * max diag once their predecessors are eliminated are dropped --, and appends the accepted columns
// L[:, new] = A[:, C_acc] L_bb^-T row by row, downdates d and renews the candidates (launch 2).
// d = sqrt(p) and 1 / d for a positive normal p: v_rsq_f64 seed, two coupled Goldschmidt steps, one residual correction (the library's
// sqrt() and a division carry range handling that this dependent chain does not need)
__device__ __forceinline__ void sqrt_and_inverse(double p, double& d, double& inv_d) {
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

constexpr int PB = 32;                               // candidates per block step
constexpr int PBK = 64;                              // earlier columns staged per tile of the panel product
constexpr int PF_ROWS = 256;                         // rows per workgroup of the second launch (one per thread)
constexpr int PB_MAX_BLOCKS = 1024;                  // candidates a step chooses from: two per 64-row block (P <= 32 768)

// Pivots of one block step must not fall far below the largest remaining diagonal entry: a pivot p turns the round-off n of a panel
// entry into an error n / sqrt(p) of the new column, i.e. 2 sqrt(d_i / p) n in the remaining diagonal of row i.  Greedy pivoting has
// d_i <= p always; a block that accepted whatever stayed above the stopping tolerance (pivots of 1e-13 max diag next to rows with
// 1e-6 left -- strongly correlated candidates collapse once their neighbours are eliminated) lost 1e-7 max diag on the model's S.
// Candidates below PB_THETA x (largest candidate of the step) are therefore neither selected nor accepted inside the block; they
// stay open with their recomputed diagonal entry and come back when the rest has caught up.
constexpr double PB_THETA = 0.01;

struct PivDesc {             // written by workgroup 0 of the panel launch, read by every workgroup of the factor launch
  double top;                // largest remaining diagonal entry at the start of the step
  int k, bsel;
  int C[PB];
  int newcur;                // the hint cursor after this step's selection (committed by the factor launch)
  int from_big;              // the step's columns are rows pos[m] of PivWork::big (matrix-free form)
  int pos[PB];
};
static_assert(sizeof(PivDesc) <= 64 * sizeof(double), "PivDesc must fit its slot of the workspace");

__global__ __launch_bounds__(PC_THREADS) void pivblk_panel_kernel(const double* __restrict__ S, int64_t lds, int P, double tol_rel, int max_rank,
                                                                  const double* __restrict__ LT, int64_t ldl,
                                                                  const double* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                                  const double* __restrict__ cand_sum, int nblk, double* __restrict__ AP,
                                                                  PivDesc* __restrict__ desc, int32_t* __restrict__ state,
                                                                  PivWork* __restrict__ work, int phase, const double* __restrict__ Scol,
                                                                  const double* __restrict__ d, const int* __restrict__ hint) {
  // phase 2: selection and panel from the matrix S (the usual step).  The matrix-free form (S never assembled: its columns are formed
  // from the Jacobian on demand) splits a step: phase 0 = selection only (one workgroup; the caller then forms the chosen columns
  // Scol[m][i] = S[C_m][i]), phase 1 = the panel from Scol with the selection read back from `desc`.
  __shared__ double cv[PB_MAX_BLOCKS];
  __shared__ int ci[PB_MAX_BLOCKS];
  __shared__ double Wt[PBK][PB];
  __shared__ double red[3][PC_ROWS][17];
  __shared__ int s_C[PB];
  __shared__ int s_bsel, s_k, s_done, s_newcur, s_from_big;
  __shared__ int s_pos[PB];
  __shared__ double s_sum, s_top;
  const int t = threadIdx.x, lane = t & 63;
  const int slice = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ncand = 2 * nblk;
  if (phase == 1) {
    if (t == 0) { s_k = desc->k; s_bsel = desc->bsel; s_done = state[1]; s_sum = 0.0; s_top = desc->top; s_from_big = desc->from_big; }
    if (t < PB) { s_C[t] = desc->C[t]; s_pos[t] = desc->pos[t]; }
  } else {
    for (int q = t; q < ncand; q += PC_THREADS) { cv[q] = cand_val[q]; ci[q] = cand_idx[q]; }
  }
  __syncthreads();
  if (phase != 1 && slice == 0) {
    const int k = state[0];
    double sum = 0.0;
    for (int q = lane; q < ncand; q += 64) sum += cand_sum[q];
    sum = wave_sum(sum);
    // the candidates of the step: every entry >= max(tol d0, PB_THETA top), at most min(PB, room) of them -- if more qualify the bound
    // is raised in steps of 10^(1/8) until they fit.  Their order does not matter (the block pivots inside), so they are placed by
    // ballot and prefix count: one wave reduction for `top` instead of PB dependent arg-max rounds (19 us of ds_bpermute at PB = 32).
    double top = NEG_INF;
    int topi = 0x7fffffff;
    for (int q = lane; q < ncand; q += 64) {
      const double ov = cv[q];
      const int oi = ci[q];
      if (ov > top || (ov == top && oi < topi)) { top = ov; topi = oi; }
    }
    wave_argmax(top, topi);
    const double d0 = (k == 0) ? top : work->d0;
    const int room = (max_rank - k < PB) ? (max_rank - k) : PB;
    int bsel = 0, newcur = work->hint_cur;
    bool from_big = false;
    if (top > tol_rel * d0 && room > 0) {
      const double floor_v = tol_rel * d0;
      double thr = fmax(floor_v, PB_THETA * top);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      // (a) a hint list (the pivots of an earlier factorisation of a nearby matrix, in their order): its first open entries that
      // qualify NOW -- current remaining diagonal >= thr -- are the candidates.  Picked by the remaining diagonal alone, two per
      // 64-row block, about 13 of 32 candidates survive the block (neighbours collapse once one of them is eliminated); earlier pivots
      // were each new information given their predecessors and mostly still are.  Only the CHOICE of candidates changes: acceptance
      // inside the block, the PB_THETA window and the stopping criterion are those of the unhinted step, and the largest remaining
      // entry is always among the candidates.  The cursor moves past the leading entries that can never qualify again.
      int nh = 0;
      bool has_top = false;
      int hn = (hint && work->hint_magic == HINT_MAGIC) ? work->hint_n : 0;
      if (hn > P) hn = P;
      // matrix-free form with the list's columns formed ahead (PivWork::big): a step takes list entries ONLY while any qualifies --
      // no pass over the Jacobian for it -- and is an ordinary step (largest entry and all) when none does
      const bool big = work->big != nullptr;
      if (big && hn > work->big_n) hn = work->big_n;
      if (hn > 0) {
        int cur = newcur < 0 ? 0 : newcur;
        bool lead = true;
        for (int w0 = cur; w0 < hn && nh < room && w0 < cur + 256; w0 += 64) {
          const int q = w0 + lane;
          const int row = (q < hn) ? hint[q] : -1;
          const bool valid = row >= 0 && row < P;
          const double dv = valid ? d[row] : NEG_INF;
          const bool alive = valid && dv > floor_v;
          const bool take = alive && dv >= thr;
          const unsigned long long mask = __ballot(take);
          const int pos = nh + __popcll(mask & below);
          if (take && pos < room) { s_C[pos] = row; s_pos[pos] = q; }
          has_top = has_top || (__ballot(take && pos < room && row == topi) != 0ull);
          nh += __popcll(mask);
          if (lead) {
            const unsigned long long am = __ballot(alive);
            const int run = am ? (__ffsll(am) - 1) : 64;
            newcur = (w0 + run < hn) ? w0 + run : hn;
            lead = (run == 64);
          }
        }
        if (nh > room) nh = room;
        if (nh > 0 && !has_top && !big) {           // the largest remaining entry rides along (it replaces the last hint if the step is full)
          if (nh < room) ++nh;
          if (lane == 0) s_C[nh - 1] = topi;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // (b) the rest of the step (all of it without a hint) from the two largest entries of every 64-row block
      from_big = big && nh > 0;
      const int room2 = from_big ? 0 : room - nh;
      unsigned excl = 0;                             // bit qi: candidate qi * 64 + lane is among the hints chosen above
      if (nh > 0 && room2 > 0) {
        for (int q0 = 0, qi = 0; q0 < ncand; q0 += 64, ++qi) {
          const int q = q0 + lane;
          if (q < ncand) {
            const int row = ci[q];
            bool dup = false;
            for (int j = 0; j < nh; ++j) dup = dup || (s_C[j] == row);
            if (dup) excl |= 1u << (qi & 31);
          }
        }
      }
      int base = 0;
      if (room2 > 0) {
        for (int tries = 0; tries < 20; ++tries) {
          int count = 0;
          for (int q0 = 0, qi = 0; q0 < ncand; q0 += 64, ++qi) {
            const int q = q0 + lane;
            count += __popcll(__ballot(q < ncand && cv[q] >= thr && cv[q] > floor_v && !((excl >> (qi & 31)) & 1u)));
          }
          if (count <= room2) break;
          thr *= 1.333521432163324;                    // 10^(1/8)
        }
        for (int q0 = 0, qi = 0; q0 < ncand; q0 += 64, ++qi) {
          const int q = q0 + lane;
          const bool take = q < ncand && cv[q] >= thr && cv[q] > floor_v && !((excl >> (qi & 31)) & 1u);
          const unsigned long long mask = __ballot(take);
          const int pos = base + __popcll(mask & below);
          if (take && pos < room2) s_C[nh + pos] = ci[q];
          base += __popcll(mask);
        }
        if (base > room2) base = room2;
      }
      bsel = nh + base;
      if (bsel == 0) {                               // (ties that never fit under the rising bound: the largest entry alone)
        if (lane == 0) s_C[0] = topi;
        bsel = 1;
      }
    }
    if (!(top > NEG_INF)) top = 0.0;
    if (lane == 0) { s_bsel = bsel; s_k = k; s_done = state[1]; s_sum = sum; s_top = top; s_newcur = newcur; s_from_big = from_big ? 1 : 0; }
  }
  __syncthreads();
  if (s_done) {
    if (phase != 1 && blockIdx.x == 0 && t == 0) work->skip[1] = 1;
    return;
  }
  const int k = s_k, bsel = s_bsel;
  if (phase != 1 && blockIdx.x == 0 && t == 0) {
    desc->top = s_top;
    desc->k = k;
    desc->bsel = bsel;
    desc->newcur = s_newcur;
    desc->from_big = s_from_big;
    for (int m = 0; m < PB; ++m) desc->pos[m] = (s_from_big && m < bsel) ? s_pos[m] : 0;
    work->skip[1] = (s_from_big || bsel == 0) ? 1 : 0;
    if (phase == 0 && bsel > 0 && !s_from_big) work->pool_steps += 1;
    for (int m = 0; m < PB; ++m) desc->C[m] = (m < bsel) ? s_C[m] : (bsel ? s_C[0] : 0);   // (unused entries: a valid row for the caller's gather)
    if (k == 0) work->d0 = s_top;
    work->trace_left = s_sum;
    if (bsel == 0 && k < max_rank) state[1] = 1;                       // nothing left above the tolerance (a full `max_rank` is not "done")
  }
  if (bsel == 0 || phase == 0) return;
  // ---- panel: A[i][m] = S[C_m][i] - sum_{j < k} L[i][j] L[C_m][j]; wave s forms the columns 2 s, 2 s + 1 for the 64 rows of the workgroup
  // wave s = (quarter jq = s >> 2 of the earlier columns, group cg = s & 3 of eight panel columns): every row of L is loaded by four
  // waves instead of sixteen and a thread's loop is k / 4 long with eight independent sums (2 columns x all of k per wave took 8 us
  // of dependent loads at k = 150); the four quarters meet in the LDS, sixteen panel columns at a time
  const int i = (int)blockIdx.x * PC_ROWS + lane;                      // (< ldl)
  const int jq = slice >> 2, cg = slice & 3;
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0;
  for (int j0 = 0; j0 < k; j0 += PBK) {
    const int nj = (k - j0 < PBK) ? (k - j0) : PBK;
    __syncthreads();
    for (int e = t; e < nj * PB; e += PC_THREADS) {
      const int jj = e / PB, m = e - jj * PB;
      Wt[jj][m] = (m < bsel) ? LT[(int64_t)(j0 + jj) * ldl + s_C[m]] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int jj = jq; jj < nj; jj += 4) {
      const double v = LT[(int64_t)(j0 + jj) * ldl + i];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = fma(v, Wt[jj][8 * cg + c], acc[c]);
    }
  }
  const bool inside = i < P;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                                   // (red is free: first use, or the previous half has been read)
    if ((cg >> 1) == h && jq > 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[jq - 1][lane][8 * (cg & 1) + c] = acc[c];
    }
    __syncthreads();
    if ((cg >> 1) == h && jq == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int m = 8 * cg + c;
        const double sum = acc[c] + red[0][lane][8 * (cg & 1) + c] + red[1][lane][8 * (cg & 1) + c] + red[2][lane][8 * (cg & 1) + c];
        const double sv = (inside && m < bsel) ? (s_from_big ? work->big[(int64_t)s_pos[m] * ldl + i]
                                                             : (Scol ? Scol[(int64_t)m * ldl + i] : S[(int64_t)s_C[m] * lds + i])) : 0.0;
        AP[(int64_t)m * ldl + i] = (inside && m < bsel) ? sv - sum : 0.0;
      }
    }
  }
}

__global__ __launch_bounds__(PF_ROWS) void pivblk_factor_kernel(int P, double tol_rel, double* __restrict__ LT, int64_t ldl, double* __restrict__ d,
                                                                double* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                                double* __restrict__ cand_sum, int nblk, const double* __restrict__ AP,
                                                                const PivDesc* __restrict__ desc, int32_t* __restrict__ piv,
                                                                int32_t* __restrict__ state, PivWork* __restrict__ work, int diag_flags) {
  // (timing diagnostics, VMCPDE_PIVBLK_DIAG, wrong results: 1 the block's eliminations stop after the first, 2 no row substitution)
  __shared__ double LP[PB][PB + 1];        // row n = the n-th accepted pivot's row of the block factor (columns 0 .. n)
  __shared__ double Lb[PB][PB + 1];        // block factor by candidate: Lb[m][n], n = elimination step
  __shared__ double Abb[PB][PB + 1];       // the block itself: Abb[m][c] = A[C_m][c] (column c of the panel at the row of candidate m)
  __shared__ double s_left[PB];            // what is left of every candidate's diagonal entry after the block
  __shared__ double sdiag[PB];             // the open candidates' remaining entries during the block's elimination
  __shared__ double s_rinv[PB];            // 1 / l_nn of the accepted pivots
  __shared__ int s_perm[PB], s_C[PB], s_acc[PB];
  __shared__ int s_bacc, s_k, s_bsel, s_done;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) { s_k = desc->k; s_bsel = desc->bsel; s_done = state[1]; }
  if (t < PB) { s_C[t] = desc->C[t]; s_acc[t] = 0; }
  __syncthreads();
  const int k = s_k, bsel = s_bsel;
  if (s_done || bsel == 0) return;
  if (wave == 0) {
    // the PB x PB block with pivoting: lane m holds candidate m (its row of the block, then its row of the factor); left-looking
    const int m = lane;
    const bool mine = m < bsel;
    const double thresh = fmax(tol_rel * work->d0, PB_THETA * desc->top);
    double Lm[PB];
#pragma unroll
    for (int n = 0; n < PB; ++n) {           // (all loads of the block in flight at once: the elimination below then runs out of the LDS)
      Lm[n] = 0.0;
      if (mine) Abb[m][n] = (n < bsel) ? AP[(int64_t)n * ldl + s_C[m]] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    double diag = mine ? Abb[m][m] : NEG_INF;
    bool used = !mine, going = true;
    int bacc = 0;
#pragma unroll
    for (int n = 0; n < PB; ++n) {
      if (!going || n >= bsel || ((diag_flags & 1) && n >= 1)) continue;     // (wave-uniform: nothing left to eliminate)
      // arg max of the open candidates' remaining entries: every lane reads all PB of them (a ds_bpermute reduction costs ~1100
      // cycles per step; ties go to the lower candidate)
      if (m < PB) sdiag[m] = used ? NEG_INF : diag;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double a[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) a[c] = sdiag[c];
#pragma unroll
      for (int wd = PB / 2; wd >= 1; wd >>= 1)
#pragma unroll
        for (int c = 0; c < wd; ++c) a[c] = fmax(a[c], a[c + wd]);
      const double v = a[0];
      going = going && (v > thresh);         // (wave-uniform)
      if (!going) continue;
      const int p = __ffsll((unsigned long long)__ballot(m < PB && !used && diag == v)) - 1;     // the lowest candidate that holds the maximum
      double lpp, rpp;
      sqrt_and_inverse(v, lpp, rpp);
      // l[m][n] = (A[m][p] - sum_{q < n} L[m][q] L[p][q]) / l_pp ; row p of the factor comes from the lane that holds it, through LDS
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < PB; ++q)
        if (q < n) sum = fma(Lm[q], Lb[p][q], sum);
      double l = 0.0;
      if (mine && !used) l = (m == p) ? lpp : (Abb[m][p] - sum) * rpp;
      Lm[n] = l;
      if (mine) Lb[m][n] = l;
      if (mine && !used && m != p) diag -= l * l;
      if (m == p) { used = true; s_perm[n] = m; s_acc[m] = 1; s_rinv[n] = rpp; }
      bacc = n + 1;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the row just written is read by every lane in the next step
    }
    if (mine) s_left[m] = diag;
    if (lane == 0) s_bacc = bacc;
  }
  __syncthreads();
  const int bacc = s_bacc;
  for (int e = t; e < bacc * PB; e += PF_ROWS) {           // factor rows in elimination order
    const int n = e / PB, q = e - n * PB;
    LP[n][q] = (q <= n) ? Lb[s_perm[n]][q] : 0.0;
  }
  __syncthreads();
  // ---- one row per thread: x = A[i][C_acc] L_bb^-T (forward substitution in elimination order), L[i][k + n] = x_n, d_i -= |x|^2
  if (diag_flags & 2) return;
  const int i = (int)blockIdx.x * PF_ROWS + t;
  const bool inside = i < P;
  const double di = inside ? d[i] : NEG_INF;
  const bool open = inside && di != NEG_INF;
  int own = PB;                            // elimination step at which this row became a pivot (PB: it is none of this block's pivots)
  for (int n = 0; n < bacc; ++n)
    if (s_C[s_perm[n]] == i) own = n;
  double x[PB];
  double ss = 0.0;
#pragma unroll
  for (int n = 0; n < PB; ++n)               // (all of the row's panel entries in flight at once, not one round trip per column)
    x[n] = (n < bacc && open) ? AP[(int64_t)s_perm[n] * ldl + i] : 0.0;
#pragma unroll
  for (int n = 0; n < PB; ++n) {
    if (n < bacc) {
      double v = x[n];
#pragma unroll
      for (int q = 0; q < PB; ++q)
        if (q < n) v = fma(-x[q], LP[n][q], v);
      v = (open && n <= own) ? v * s_rinv[n] : 0.0;       // (a pivot row is finished: structural zeros in the later columns)
      x[n] = v;
      ss = fma(v, v, ss);
      if (i < (int)ldl) LT[(int64_t)(k + n) * ldl + i] = v;
    }
  }
  double nd = open ? di - ss : NEG_INF;
  if (open) {
    bool pivot = false;                                          // (a row may be listed twice by a careless hint list: once a pivot, always closed)
    for (int m = 0; m < bsel; ++m)
      if (s_C[m] == i) {
        if (s_acc[m]) { nd = NEG_INF; pivot = true; }            // a pivot is finished;
        else if (!pivot) nd = s_left[m];                         // a dropped candidate keeps what the block left of it
      }
  }
  if (inside) d[i] = nd;
  double v1 = nd;
  int idx = i;
  double pos = (nd > 0.0) ? nd : 0.0;
  wave_argmax(v1, idx);
  pos = wave_sum(pos);
  double v2 = (i == idx) ? NEG_INF : nd;      // the block's second largest entry: two candidates per 64 rows
  int idx2 = i;
  wave_argmax(v2, idx2);
  const int blk = (int)blockIdx.x * (PF_ROWS / 64) + wave;
  if (lane == 0 && blk < nblk) {
    cand_val[2 * blk] = v1;
    cand_idx[2 * blk] = idx;
    cand_sum[2 * blk] = pos;
    cand_val[2 * blk + 1] = v2;
    cand_idx[2 * blk + 1] = idx2;
    cand_sum[2 * blk + 1] = 0.0;
  }
  if (blockIdx.x == 0 && t == 0) {
    state[0] = k + bacc;
    for (int n = 0; n < bacc; ++n) piv[k + n] = s_C[s_perm[n]];
    work->hint_cur = desc->newcur;
    work->steps += 1;
  }
}

// ---- matrix-free form: S = J^T J / N - m m^T is never assembled ----------------------------------------------------------------
// d = the given diagonal; the two largest entries of every 64-row block as candidates (as pivchol_init_kernel with two = 1)
__global__ __launch_bounds__(PC_ROWS) void pivchol_init_diag_kernel(const double* __restrict__ diag, int P, double* __restrict__ d,
                                                                    double* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                                    double* __restrict__ cand_sum, int32_t* __restrict__ state,
                                                                    PivWork* __restrict__ work) {
  const int i = (int)blockIdx.x * PC_ROWS + (int)threadIdx.x;
  double mine = NEG_INF;
  if (i < P) {
    mine = diag[i];
    d[i] = mine;
  }
  double v = mine;
  int idx = i;
  double pos = (v > 0.0) ? v : 0.0;
  wave_argmax(v, idx);
  pos = wave_sum(pos);
  double v2 = (i == idx) ? NEG_INF : mine;
  int idx2 = i;
  wave_argmax(v2, idx2);
  if (threadIdx.x == 0) {
    cand_val[2 * blockIdx.x] = v;
    cand_idx[2 * blockIdx.x] = idx;
    cand_sum[2 * blockIdx.x] = pos;
    cand_val[2 * blockIdx.x + 1] = v2;
    cand_idx[2 * blockIdx.x + 1] = idx2;
    cand_sum[2 * blockIdx.x + 1] = 0.0;
    if (blockIdx.x == 0) {
      state[0] = 0; state[1] = 0;
      work->hint_cur = 0; work->steps = 0; work->pool_steps = 0; work->big = nullptr; work->big_n = 0; work->skip[0] = 0; work->skip[1] = 0;
    }
  }
}

// Columns of the Gram matrix on demand: part[chunk][m][i] = sum_{n in chunk} JT[C_m][n] JT[i][n] for the PB rows C and all rows i --
// a skinny GEMM that reads the Jacobian once (HBM-bound: 2.2 GB at the headline size against 1.8e10 flop).  Workgroup = 64 rows i x PB
// rows C_m over one sample chunk; both operands are staged through the LDS by coalesced row-segment loads, wave w multiplies the
// rows 16 w .. 16 w + 15 against the 32 candidates (candidates as the MFMA's row operand, so that a lane group writes 16 consecutive
// i of one candidate's column).
typedef double double4_t __attribute__((ext_vector_type(4)));
// LDS image of an operand: sample c of row r sits at column (c + 4 r) & 31 of a row of 33 doubles.  A fragment read takes row l & 15,
// sample 4 k4 + (l >> 4): bank pair (33 r + c + 4 r) mod 32 = 5 r + c -- two lanes per pair over the 64 lanes, the minimum (unrotated,
// r + c put them on 19 pairs); the stores (row t >> 2, eight samples from 8 (t & 3)) keep 5 r + 8 s + q, also two per pair.
constexpr int CP_ROWS = 64, CP_BK = 32, CP_THREADS = 256, CP_PITCH = CP_BK + 1;
__device__ __forceinline__ int cp_col(int row, int c) { return (c + 4 * row) & (CP_BK - 1); }
constexpr int CP_MAX_CHUNKS = 64;                    // sample chunks (split K) a panel may be formed from
constexpr int64_t CP_RESIDENT = 256 * 6;             // workgroups of colpanel_kernel the chip holds at once
__global__ __launch_bounds__(CP_THREADS) void colpanel_kernel(const double* __restrict__ JT, int64_t ldj, int64_t n, const int32_t* __restrict__ C,
                                                              int64_t chunk, double* __restrict__ part, int64_t ldl,
                                                              const int32_t* __restrict__ state) {
  if (state && state[1]) return;          // (the factorisation this panel is for has met its criterion: a static plan's spare steps)
  __shared__ double As[CP_ROWS][CP_PITCH];
  __shared__ double Bs[PB][CP_PITCH];
  __shared__ int s_C[PB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < PB) s_C[t] = C[t];
  __syncthreads();
  const int i0 = (int)blockIdx.x * CP_ROWS;
  const int64_t n0 = (int64_t)blockIdx.y * chunk;
  const int64_t n1 = (n0 + chunk < n) ? (n0 + chunk) : n;
  const int ar = t >> 2, as = (t & 3) * 8;                 // A: row ar, 8 samples from as
  const int br = t >> 3, bs = (t & 7) * 4;                 // B: candidate br, 4 samples from bs
  const double* __restrict__ arow = JT + (int64_t)(i0 + ar) * ldj;
  const double* __restrict__ brow = JT + (int64_t)s_C[br] * ldj;
  double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  const int lr = lane & 15, lk = lane >> 4;
  typedef double double2v __attribute__((ext_vector_type(2)));
  double2v va[4], vb[2];
  // one step's operands: whole 16-byte loads where the step lies inside the chunk (rows are 128-byte aligned, steps 256 bytes), element
  // by element with zeros past the chunk's end otherwise
  auto fetch = [&](int64_t nk) {
    if (nk + CP_BK <= n1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) va[q] = *reinterpret_cast<const double2v*>(arow + nk + as + 2 * q);
#pragma unroll
      for (int q = 0; q < 2; ++q) vb[q] = *reinterpret_cast<const double2v*>(brow + nk + bs + 2 * q);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        va[q].x = (nk + as + 2 * q < n1) ? arow[nk + as + 2 * q] : 0.0;
        va[q].y = (nk + as + 2 * q + 1 < n1) ? arow[nk + as + 2 * q + 1] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        vb[q].x = (nk + bs + 2 * q < n1) ? brow[nk + bs + 2 * q] : 0.0;
        vb[q].y = (nk + bs + 2 * q + 1 < n1) ? brow[nk + bs + 2 * q + 1] : 0.0;
      }
    }
  };
  if (n0 < n1) fetch(n0);
  for (int64_t nk = n0; nk < n1; nk += CP_BK) {
    __syncthreads();                                       // (the previous step's fragments have been read)
#pragma unroll
    for (int q = 0; q < 4; ++q) { As[ar][cp_col(ar, as + 2 * q)] = va[q].x; As[ar][cp_col(ar, as + 2 * q + 1)] = va[q].y; }
#pragma unroll
    for (int q = 0; q < 2; ++q) { Bs[br][cp_col(br, bs + 2 * q)] = vb[q].x; Bs[br][cp_col(br, bs + 2 * q + 1)] = vb[q].y; }
    __syncthreads();
    if (nk + CP_BK < n1) fetch(nk + CP_BK);                // the next step's loads fly under this step's products
#pragma unroll
    for (int k4 = 0; k4 < CP_BK / 4; ++k4) {
      const double bi = As[16 * wave + lr][cp_col(16 * wave + lr, 4 * k4 + lk)];   // the 16 rows i of this wave: the MFMA's column operand
      const double c0 = Bs[lr][cp_col(lr, 4 * k4 + lk)], c1 = Bs[16 + lr][cp_col(16 + lr, 4 * k4 + lk)];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, bi, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, bi, acc1, 0, 0, 0);
    }
  }
  // C/D element q of lane l: row (candidate) = (l >> 4) + 4 q, column (row i) = l & 15
  double* __restrict__ out = part + (int64_t)blockIdx.y * PB * ldl;
  const int i = i0 + 16 * wave + lr;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    out[(int64_t)(lk + 4 * q) * ldl + i] = acc0[q];
    out[(int64_t)(16 + lk + 4 * q) * ldl + i] = acc1[q];
  }
}

// Scol[m][i] = (sum over the chunks) / N - mean[i] mean[C_m]   (mean NULL: no centring)
__global__ void colpanel_reduce_kernel(const double* __restrict__ part, int nchunk, int64_t ldl, int P, const int32_t* __restrict__ C,
                                       const double* __restrict__ mean, double inv_n, double* __restrict__ Scol,
                                       const int32_t* __restrict__ state) {
  if (state && state[1]) return;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)PB * ldl) return;
  const int m = (int)(e / ldl), i = (int)(e - (int64_t)m * ldl);
  double sum = 0.0;
  for (int c = 0; c < nchunk; ++c) sum += part[(int64_t)c * PB * ldl + e];
  double v = 0.0;
  if (i < P) v = sum * inv_n - (mean ? mean[i] * mean[C[m]] : 0.0);
  Scol[e] = v;
}

// y = JT (JT^T v): the product S0 v of the matrix-free step without its scaling and mean term -- two passes over the Jacobian, both bound
// by HBM.  (The library's gemv for z = JT^T v, 65 536 outputs of length-P sums, reaches 2.3 TB/s; its row-dot form 4-5.)
// Pass 1: z[c] = sum_p v[p] JT[p][c].  A workgroup owns 512 consecutive samples (two per thread: a wave reads 1 KB of a row at a time)
// and one slice of the rows; the slices' partial sums are added by a second small launch, in a fixed order.
constexpr int MV_THREADS = 256, MV_COLS = 2 * MV_THREADS, MV_MAX_SLICES = 64;
__global__ __launch_bounds__(MV_THREADS) void matvec_cols_kernel(const double* __restrict__ JT, int64_t ldj, int P, int64_t n,
                                                                 const double* __restrict__ v, int rows_per_slice,
                                                                 double* __restrict__ zpart, int64_t ldz, int vec2) {
  const int64_t c = ((int64_t)blockIdx.x * MV_THREADS + threadIdx.x) * 2;
  const int p0 = (int)blockIdx.y * rows_per_slice;
  const int p1 = (p0 + rows_per_slice < P) ? p0 + rows_per_slice : P;
  if (c >= n) return;
  const bool pair = c + 1 < n;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  const double* __restrict__ base = JT + c;
  int p = p0;
  if (vec2 && pair) {
#pragma unroll 4
    for (; p + 1 < p1; p += 2) {
      const double2 r0 = *reinterpret_cast<const double2*>(base + (int64_t)p * ldj);
      const double2 r1 = *reinterpret_cast<const double2*>(base + (int64_t)(p + 1) * ldj);
      const double x0 = v[p], x1 = v[p + 1];
      a0 = fma(x0, r0.x, a0);
      a1 = fma(x0, r0.y, a1);
      b0 = fma(x1, r1.x, b0);
      b1 = fma(x1, r1.y, b1);
    }
  }
  for (; p < p1; ++p) {
    const double x = v[p];
    a0 = fma(x, base[(int64_t)p * ldj], a0);
    if (pair) a1 = fma(x, base[(int64_t)p * ldj + 1], a1);
  }
  zpart[(int64_t)blockIdx.y * ldz + c] = a0 + b0;
  if (pair) zpart[(int64_t)blockIdx.y * ldz + c + 1] = a1 + b1;
}

__global__ void matvec_cols_reduce_kernel(const double* __restrict__ zpart, int nslice, int64_t ldz, int64_t n, double* __restrict__ z) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double sum = zpart[c];
  for (int k = 1; k < nslice; ++k) sum += zpart[(int64_t)k * ldz + c];
  z[c] = sum;
}

// Pass 2: y[p] = sum_c JT[p][c] z[c], one workgroup of four waves per row (a wave per row is 4213 waves of 512 KB on 4096 wave slots at
// the headline size: a second, nearly empty round -- 0.50 ms against 0.37 ms for pass 1), four sums in flight per lane
__global__ __launch_bounds__(256) void matvec_rows_kernel(const double* __restrict__ JT, int64_t ldj, int P, int64_t n,
                                                          const double* __restrict__ z, double* __restrict__ y, int vec2) {
  __shared__ double part[4];
  const int row = (int)blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* __restrict__ a = JT + (int64_t)row * ldj;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int64_t c = 0;
  if (vec2) {
    const int64_t n4 = n / 1024 * 1024;               // the workgroup takes 512 samples per load instruction, two instructions per iteration
    for (c = 2 * t; c < n4; c += 1024) {
      const double2 r0 = *reinterpret_cast<const double2*>(a + c), r1 = *reinterpret_cast<const double2*>(a + c + 512);
      const double2 w0 = *reinterpret_cast<const double2*>(z + c), w1 = *reinterpret_cast<const double2*>(z + c + 512);
      s0 = fma(r0.x, w0.x, s0);
      s1 = fma(r0.y, w0.y, s1);
      s2 = fma(r1.x, w1.x, s2);
      s3 = fma(r1.y, w1.y, s3);
    }
    c = n4;
  }
  for (int64_t q = c + t; q < n; q += 256) s0 = fma(a[q], z[q], s0);
  const double s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (t == 0) y[row] = (part[0] + part[1]) + (part[2] + part[3]);
}

constexpr int HQ_THREADS = 1024;
constexpr int HQ_WAVES = HQ_THREADS / 64;

__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int t = threadIdx.x;
  __syncthreads();                                   // (red may still be read from the previous call)
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double total = 0.0;
#pragma unroll
  for (int q = 0; q < HQ_WAVES; ++q) total += red[q];
  return total;
}

// Step k of the Householder QR of L (columns = rows of LT): workgroup 0 writes the reflector (v with v_k = 1, zeros above) to row k of YT
// and tau_k; workgroup m >= 1 applies H_k = I - tau v v^T to column k + m.  Every workgroup derives v from column k itself, which no
// workgroup of this launch writes; the sum of squares of column k below its diagonal comes from the workgroup that brought that
// column up to date in step k - 1 (norm2[k]; step 0 sums it itself), so a step is one pass for the dot product and one for the update.
// (dlarfg's formulas; a column that is already zero below the diagonal gets the reflection I - 2 e_k e_k^T instead of the identity, so
// that tau is never 0 and the compact WY factor stays regular.)
__global__ __launch_bounds__(HQ_THREADS) void hqr_step_kernel(double* __restrict__ LT, int64_t ldl, int P, int k, double* __restrict__ YT,
                                                              double* __restrict__ tau, double* __restrict__ norm2) {
  __shared__ double red[HQ_WAVES];
  const int t = threadIdx.x, m = blockIdx.x;
  const double* __restrict__ x = LT + (int64_t)k * ldl;
  double s2;
  if (k == 0) {
    s2 = 0.0;
    for (int i = 1 + t; i < P; i += HQ_THREADS) s2 = fma(x[i], x[i], s2);
    s2 = block_sum(s2, red);
  } else {
    s2 = norm2[k];
  }
  const double alpha = x[k];
  const double nrm2 = fma(alpha, alpha, s2);
  double tk = 2.0, scale = 0.0;
  if (nrm2 > 0.0) {
    const double beta = -copysign(sqrt(nrm2), alpha);
    tk = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  if (m == 0) {
    double* __restrict__ y = YT + (int64_t)k * ldl;
    for (int i = t; i < (int)ldl; i += HQ_THREADS) y[i] = (i < k || i >= P) ? 0.0 : ((i == k) ? 1.0 : x[i] * scale);
    if (t == 0) tau[k] = tk;
    return;
  }
  double* __restrict__ y = LT + (int64_t)(k + m) * ldl;
  double dot = (t == 0) ? y[k] : 0.0;
#pragma unroll 4
  for (int i = k + 1 + t; i < P; i += HQ_THREADS) dot = fma(x[i] * scale, y[i], dot);
  dot = block_sum(dot, red);
  const double w = tk * dot;
  if (t == 0) y[k] -= w;
  double next = 0.0;                                  // (m == 1) sum of squares of the new column k + 1 below ITS diagonal
#pragma unroll 4
  for (int i = k + 1 + t; i < P; i += HQ_THREADS) {
    const double v = fma(-w, x[i] * scale, y[i]);
    y[i] = v;
    if (i > k + 1) next = fma(v, v, next);
  }
  if (m == 1) {
    next = block_sum(next, red);
    if (t == 0) norm2[k + 1] = next;
  }
}

// NR steps per launch (NR = 2 or 4): every workgroup derives ALL NR reflectors itself -- v_k from column k as above, then for
// j = 1 .. NR - 1 column k + j with H_k .. H_{k+j-1} applied, held in registers and never written (all workgroups read the same
// untouched columns), and v_{k+j} from it -- and applies them to its own column, also through registers: one read and one write of
// the column.  A dependent launch costs ~4.7 us; the redundant part is NR (NR - 1) / 2 dot products and NR - 1 norms of block sums at
// ~0.35 us each.  Workgroup 0 writes the reflectors; the entries of R above the diagonal INSIDE the block (columns k + 1 .. k + NR - 1,
// rows k ..) go to rblk and are copied into LT by a final launch -- those columns must stay untouched while other workgroups of this
// launch read them.  A thread holds elements i = k + 1 + t + 1024 e, e < MAXE (4, 8 or 16 by the number of rows): P <= 16 384.
constexpr int HQ_MAXE_LIMIT = 16;
constexpr int HQB_THREADS = 512;                       // workgroup of the batched form (hqr_batched_kernel)
constexpr int HQ_RB = 4;                                 // doubles of rblk per column
template <int NR, int MAXE>
__global__ __launch_bounds__(HQ_THREADS) void hqr_multi_kernel(double* __restrict__ LT, int64_t ldl, int P, int k, double* __restrict__ YT,
                                                               double* __restrict__ tau, double* __restrict__ norm2, double* __restrict__ rblk) {
  __shared__ double red[HQ_WAVES];
  __shared__ double s_diag;
  const int t = threadIdx.x, m = blockIdx.x;
  const int i0 = k + 1 + t;
  const int ne = (P - k - 1 + HQ_THREADS - 1) / HQ_THREADS;        // (<= MAXE, checked by the host)
  double v[NR][MAXE];             // v[j]: column k + j below row k (first with the earlier reflectors applied, then scaled to v_{k+j})
  double top[NR];                 // entry of column k + j in row k (only thread 0's copy is used)
  double tk[NR];
  // ---- reflector k from column k
  {
    const double* __restrict__ x = LT + (int64_t)k * ldl;
    double s2;
    if (k == 0) {
      s2 = 0.0;
      for (int i = 1 + t; i < P; i += HQ_THREADS) s2 = fma(x[i], x[i], s2);
      s2 = block_sum(s2, red);
    } else {
      s2 = norm2[k];
    }
    const double alpha = x[k];
    const double nrm2 = fma(alpha, alpha, s2);
    double scale = 0.0;
    tk[0] = 2.0;
    if (nrm2 > 0.0) {
      const double beta = -copysign(sqrt(nrm2), alpha);
      tk[0] = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * HQ_THREADS;
      v[0][e] = (e < ne && i < P) ? x[i] * scale : 0.0;
    }
    top[0] = 1.0;                 // (v_k[k] = 1: the row-k term of every dot product with v_k is the column's own row-k entry)
  }
  // ---- columns k + 1 .. k + NR - 1 with the earlier reflectors applied, and their reflectors
#pragma unroll
  for (int j = 1; j < NR; ++j) {
    const double* __restrict__ y = LT + (int64_t)(k + j) * ldl;
    top[j] = y[k];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * HQ_THREADS;
      v[j][e] = (e < ne && i < P) ? y[i] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < j; ++a) {                                   // H_{k+a}: v[a] is zero above row k + a, one at it
      double dot = (a == 0 && t == 0) ? top[j] : 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], v[j][e], dot);
      dot = block_sum(dot, red);
      const double w = tk[a] * dot;
      if (a == 0) top[j] -= w;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) v[j][e] = fma(-w, v[a][e], v[j][e]);
    }
    // what is above the diagonal of column k + j now is final: R[k + a][k + j], a < j (row k: top[j]; row k + a: thread a - 1, e = 0)
    if (m == 0) {
      if (t == 0) rblk[(int64_t)(k + j) * HQ_RB + 0] = top[j];
#pragma unroll
      for (int a = 1; a < j; ++a)
        if (t == a - 1) rblk[(int64_t)(k + j) * HQ_RB + a] = v[j][0];
    }
    // reflector k + j: diagonal entry = element i = k + j (thread j - 1, e = 0), norm of what lies below it
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * HQ_THREADS;
      if (i > k + j) s2 = fma(v[j][e], v[j][e], s2);
    }
    __syncthreads();                                                // (s_diag may still be read from the previous column)
    if (t == j - 1) s_diag = v[j][0];
    s2 = block_sum(s2, red);                                        // (its barriers publish s_diag)
    const double alpha = s_diag;
    const double nrm2 = fma(alpha, alpha, s2);
    double scale = 0.0;
    tk[j] = 2.0;
    if (nrm2 > 0.0) {
      const double beta = -copysign(sqrt(nrm2), alpha);
      tk[j] = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * HQ_THREADS;
      v[j][e] = (i < k + j) ? 0.0 : ((i == k + j) ? 1.0 : v[j][e] * scale);      // (zero beyond P: loaded as zero)
    }
  }
  if (m == 0) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      double* __restrict__ yo = YT + (int64_t)(k + j) * ldl;
      for (int i = t; i <= k; i += HQ_THREADS) yo[i] = (j == 0 && i == k) ? 1.0 : 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const int i = i0 + e * HQ_THREADS;
        if (e < ne && i < (int)ldl) yo[i] = (i < P) ? v[j][e] : 0.0;
      }
      for (int i = i0 + ne * HQ_THREADS; i < (int)ldl; i += HQ_THREADS) yo[i] = 0.0;       // (pad columns past the last full stride)
      if (t == 0) tau[k + j] = tk[j];
    }
    return;
  }
  // ---- the workgroup's own column c = k + NR - 1 + m: H_k .. H_{k+NR-1}
  double* __restrict__ z = LT + (int64_t)(k + NR - 1 + m) * ldl;
  double cr[MAXE];
  double ztop = z[k];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = i0 + e * HQ_THREADS;
    cr[e] = (e < ne && i < P) ? z[i] : 0.0;
  }
#pragma unroll
  for (int a = 0; a < NR; ++a) {
    double dot = (a == 0 && t == 0) ? ztop : 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], cr[e], dot);
    dot = block_sum(dot, red);
    const double w = tk[a] * dot;
    if (a == 0) ztop -= w;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) cr[e] = fma(-w, v[a][e], cr[e]);
  }
  if (t == 0) z[k] = ztop;
  double next = 0.0;                                                // (m == 1) sum of squares of column k + NR below ITS diagonal
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = i0 + e * HQ_THREADS;
    if (e < ne && i < P) {
      z[i] = cr[e];
      if (i > k + NR) next = fma(cr[e], cr[e], next);
    }
  }
  if (m == 1) {
    next = block_sum(next, red);
    if (t == 0) norm2[k + NR] = next;
  }
}

// The same NR steps per launch with the dot products BATCHED (round 5).  hqr_multi_kernel applies the earlier reflectors of the block to a
// column one after the other -- a dependent block sum each, NR (NR - 1) / 2 + 2 NR - 1 of them per launch (13 for NR = 4), which is why four
// steps per launch were no faster than two.  But H_{j-1} ... H_0 y = y - sum_a w_a v_a with  w_0 = tau_0 g_0,  w_b = tau_b (g_b - sum_{a<b} w_a c_ab),
// g_a = v_a . y and c_ab = v_a . v_b: all g_a of a column against the ORIGINAL y, and the c_ab of the newest reflector, are ONE block sum of
// several values; then the column's norm below its diagonal is a second one (recomputed, not downdated: the columns are graded).  2 NR
// block-sum phases per launch instead of 13: four steps per launch now pay (52 instead of 103 dependent launches for 206 columns).
template <int WAVES>
__device__ __forceinline__ double block_sum_w(double v, double* red) {
  v = wave_sum(v);
  const int t = threadIdx.x;
  __syncthreads();                                   // (red may still be read from the previous call)
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double total = 0.0;
#pragma unroll
  for (int q = 0; q < WAVES; ++q) total += red[q];
  return total;
}

template <int K, int WAVES>
__device__ __forceinline__ void block_sum_n(double (&v)[K], double* red /* [K][WAVES] */) {
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = wave_sum(v[q]);
  const int t = threadIdx.x;
  __syncthreads();                                   // (red may still be read from the previous call)
  if ((t & 63) == 0) {
#pragma unroll
    for (int q = 0; q < K; ++q) red[q * WAVES + (t >> 6)] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) {
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) total += red[q * WAVES + w];
    v[q] = total;
  }
}

template <int NR, int MAXE, int THREADS, int RB = HQ_RB>
__global__ __launch_bounds__(THREADS) void hqr_batched_kernel(double* __restrict__ LT, int64_t ldl, int P, int k, double* __restrict__ YT,
                                                                 double* __restrict__ tau, double* __restrict__ norm2, double* __restrict__ rblk) {
  constexpr int WAVES = THREADS / 64;
  __shared__ double red[2 * NR * WAVES];
  __shared__ double s_diag;
  const int t = threadIdx.x, m = blockIdx.x;
  const int i0 = k + 1 + t;
  const int ne = (P - k - 1 + THREADS - 1) / THREADS;        // (<= MAXE, checked by the host)
  double v[NR][MAXE];             // v[j]: column k + j below row k (first as loaded, then with the earlier reflectors applied, then scaled to v_{k+j})
  double tk[NR];
  double cab[NR][NR];             // cab[a][b] = v_a . v_b for a < b (wave-uniform)
  // ---- reflector k from column k
  {
    const double* __restrict__ x = LT + (int64_t)k * ldl;
    double s2;
    if (k == 0) {
      s2 = 0.0;
      for (int i = 1 + t; i < P; i += THREADS) s2 = fma(x[i], x[i], s2);
      s2 = block_sum_w<WAVES>(s2, red);
    } else {
      s2 = norm2[k];
    }
    const double alpha = x[k];
    const double nrm2 = fma(alpha, alpha, s2);
    double scale = 0.0;
    tk[0] = 2.0;
    if (nrm2 > 0.0) {
      const double beta = -copysign(sqrt(nrm2), alpha);
      tk[0] = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * THREADS;
      v[0][e] = (e < ne && i < P) ? x[i] * scale : 0.0;
    }
  }
  // ---- columns k + 1 .. k + NR - 1 with the earlier reflectors applied (batched), and their reflectors
#pragma unroll
  for (int j = 1; j < NR; ++j) {
    const double* __restrict__ y = LT + (int64_t)(k + j) * ldl;
    double ytop = y[k];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * THREADS;
      v[j][e] = (e < ne && i < P) ? y[i] : 0.0;
    }
    // one phase: g_a = v_a . y (a < j; v_0 has a one in row k, the later reflectors a zero) and c_{a,j-1} = v_a . v_{j-1} (a < j - 1)
    double vals[2 * NR];
#pragma unroll
    for (int q = 0; q < 2 * NR; ++q) vals[q] = 0.0;
#pragma unroll
    for (int a = 0; a < j; ++a) {
      double dot = (a == 0 && t == 0) ? ytop : 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], v[j][e], dot);
      vals[a] = dot;
    }
#pragma unroll
    for (int a = 0; a + 1 < j; ++a) {
      double dot = 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], v[j - 1][e], dot);
      vals[NR + a] = dot;
    }
    block_sum_n<2 * NR, WAVES>(vals, red);
#pragma unroll
    for (int a = 0; a + 1 < j; ++a) cab[a][j - 1] = vals[NR + a];
    double w[NR];
#pragma unroll
    for (int b = 0; b < j; ++b) {
      double g = vals[b];
#pragma unroll
      for (int a = 0; a < b; ++a) g = fma(-w[a], cab[a][b], g);
      w[b] = tk[b] * g;
    }
    ytop -= w[0];
#pragma unroll
    for (int a = 0; a < j; ++a)
#pragma unroll
      for (int e = 0; e < MAXE; ++e) v[j][e] = fma(-w[a], v[a][e], v[j][e]);
    // what is above the diagonal of column k + j now is final: R[k + a][k + j], a < j (row k: ytop; row k + a: thread a - 1, e = 0)
    if (m == 0) {
      if (t == 0) rblk[(int64_t)(k + j) * RB + 0] = ytop;
#pragma unroll
      for (int a = 1; a < j; ++a)
        if (t == a - 1) rblk[(int64_t)(k + j) * RB + a] = v[j][0];
    }
    // reflector k + j: diagonal entry = element i = k + j (thread j - 1, e = 0), norm of what lies below it
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * THREADS;
      if (i > k + j) s2 = fma(v[j][e], v[j][e], s2);
    }
    __syncthreads();                                                // (s_diag may still be read from the previous column)
    if (t == j - 1) s_diag = v[j][0];
    s2 = block_sum_w<WAVES>(s2, red);                                        // (its barriers publish s_diag)
    const double alpha = s_diag;
    const double nrm2 = fma(alpha, alpha, s2);
    double scale = 0.0;
    tk[j] = 2.0;
    if (nrm2 > 0.0) {
      const double beta = -copysign(sqrt(nrm2), alpha);
      tk[j] = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = i0 + e * THREADS;
      v[j][e] = (i < k + j) ? 0.0 : ((i == k + j) ? 1.0 : v[j][e] * scale);      // (zero beyond P: loaded as zero)
    }
  }
  if (m == 0) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      double* __restrict__ yo = YT + (int64_t)(k + j) * ldl;
      for (int i = t; i <= k; i += THREADS) yo[i] = (j == 0 && i == k) ? 1.0 : 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const int i = i0 + e * THREADS;
        if (e < ne && i < (int)ldl) yo[i] = (i < P) ? v[j][e] : 0.0;
      }
      for (int i = i0 + ne * THREADS; i < (int)ldl; i += THREADS) yo[i] = 0.0;       // (pad columns past the last full stride)
      if (t == 0) tau[k + j] = tk[j];
    }
    return;
  }
  // ---- the workgroup's own column c = k + NR - 1 + m: H_k .. H_{k+NR-1}, batched
  double* __restrict__ z = LT + (int64_t)(k + NR - 1 + m) * ldl;
  double cr[MAXE];
  double ztop = z[k];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = i0 + e * THREADS;
    cr[e] = (e < ne && i < P) ? z[i] : 0.0;
  }
  {
    double vals[2 * NR];
#pragma unroll
    for (int q = 0; q < 2 * NR; ++q) vals[q] = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
      double dot = (a == 0 && t == 0) ? ztop : 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], cr[e], dot);
      vals[a] = dot;
    }
#pragma unroll
    for (int a = 0; a + 1 < NR; ++a) {
      double dot = 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) dot = fma(v[a][e], v[NR - 1][e], dot);
      vals[NR + a] = dot;
    }
    block_sum_n<2 * NR, WAVES>(vals, red);
#pragma unroll
    for (int a = 0; a + 1 < NR; ++a) cab[a][NR - 1] = vals[NR + a];
    double w[NR];
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      double g = vals[b];
#pragma unroll
      for (int a = 0; a < b; ++a) g = fma(-w[a], cab[a][b], g);
      w[b] = tk[b] * g;
    }
    ztop -= w[0];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
      for (int e = 0; e < MAXE; ++e) cr[e] = fma(-w[a], v[a][e], cr[e]);
  }
  if (t == 0) z[k] = ztop;
  double next = 0.0;                                                // (m == 1) sum of squares of column k + NR below ITS diagonal
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = i0 + e * THREADS;
    if (e < ne && i < P) {
      z[i] = cr[e];
      if (i > k + NR) next = fma(cr[e], cr[e], next);
    }
  }
  if (m == 1) {
    next = block_sum_w<WAVES>(next, red);
    if (t == 0) norm2[k + NR] = next;
  }
}

// the entries of R above the diagonal inside the blocks of NR columns starting at k_begin, k_begin + NR, ... (nblocks of them)
__global__ void hqr_commit_kernel(double* __restrict__ LT, int64_t ldl, int k_begin, int nr, int nblocks, const double* __restrict__ rblk, int rb) {
  const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int per = nr * rb;
  if (q >= nblocks * per) return;
  const int blk = q / per, j = (q - blk * per) / rb, a = q % rb;
  if (j == 0 || a >= j) return;
  const int k = k_begin + blk * nr;
  LT[(int64_t)(k + j) * ldl + (k + a)] = rblk[(int64_t)(k + j) * rb + a];
}

int64_t pivchol_ws_doubles(int P) {
  const int64_t nwg = (P + PC_ROWS - 1) / PC_ROWS;
  return 8 /* PivWork, padded */ + (int64_t)nwg * PC_ROWS /* d */ + 2 * nwg /* cand_val */ + 2 * nwg /* cand_sum */ + nwg /* cand_idx (2 x int32) */ + 8;
}

int64_t pivblk_ws_doubles(int P, int64_t ldl) {
  return pivchol_ws_doubles(P) + 64 /* PivDesc */ + (int64_t)PB * ldl /* panel */ + ldl / 2 + 8 /* hint list: int32[ldl] */;
}
inline int* pivblk_hint(double* w, int P, int64_t ldl) { return reinterpret_cast<int*>(w + pivchol_ws_doubles(P) + 64 + (int64_t)PB * ldl); }

// the pivots of a finished factorisation (piv[0 .. state[0])) become the hint list of the workspace; piv == nullptr disarms it
__global__ void pivblk_hint_kernel(const int32_t* __restrict__ piv, const int32_t* __restrict__ state, int P, int* __restrict__ hint,
                                   PivWork* __restrict__ work) {
  int n = piv ? state[0] : 0;
  if (n < 0) n = 0;
  if (n > P) n = P;
  for (int q = (int)(blockIdx.x * blockDim.x + threadIdx.x); q < n; q += (int)(gridDim.x * blockDim.x)) hint[q] = piv[q];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    work->hint_n = n;
    work->hint_cur = 0;
    work->hint_magic = n > 0 ? HINT_MAGIC : 0;
  }
}

}  // namespace
}  // namespace vmcpde

using namespace vmcpde;

extern "C" int64_t vmcpde_pivoted_cholesky_workspace_bytes(int32_t P) { return (P >= 1) ? pivchol_ws_doubles(P) * (int64_t)sizeof(double) : 0; }

extern "C" int vmcpde_pivoted_cholesky(const double* S, int64_t lds, int32_t P, double tol_rel, int32_t first_step, int32_t n_steps,
                                       int32_t max_rank, double* LT, int64_t ldl, int32_t* piv, int32_t* state, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(S && LT && piv && state && workspace, "vmcpde_pivoted_cholesky: null argument");
  VMC_REQUIRE(P >= 1 && lds >= P, "vmcpde_pivoted_cholesky: bad shape");
  VMC_REQUIRE(ldl >= (int64_t)((P + PC_ROWS - 1) / PC_ROWS) * PC_ROWS, "vmcpde_pivoted_cholesky: ldl must be P rounded up to a multiple of 64");
  VMC_REQUIRE(first_step >= 0 && n_steps >= 0 && max_rank >= 1 && max_rank <= P && first_step + n_steps <= max_rank,
              "vmcpde_pivoted_cholesky: steps outside [0, max_rank]");
  VMC_REQUIRE(tol_rel >= 0.0, "vmcpde_pivoted_cholesky: negative tolerance");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_workspace_bytes(P), "vmcpde_pivoted_cholesky: workspace too small");
  const int nwg = (P + PC_ROWS - 1) / PC_ROWS;
  double* w = static_cast<double*>(workspace);
  PivWork* work = reinterpret_cast<PivWork*>(w);
  double* d = w + 8;
  double* cand_val = d + (int64_t)nwg * PC_ROWS;
  double* cand_sum = cand_val + 2 * nwg;
  int* cand_idx = reinterpret_cast<int*>(cand_sum + 2 * nwg);
  hipStream_t st = (hipStream_t)stream;
  if (first_step == 0)
    hipLaunchKernelGGL(pivchol_init_kernel, dim3(nwg), dim3(PC_ROWS), 0, st, S, lds, P, d, cand_val, cand_idx, cand_sum, state, 0, (PivWork*)nullptr);
  for (int s = first_step; s < first_step + n_steps; ++s)
    hipLaunchKernelGGL(pivchol_step_kernel, dim3(nwg), dim3(PC_THREADS), 0, st, S, lds, P, s, tol_rel, LT, ldl, d, cand_val, cand_idx, cand_sum, nwg,
                       piv, state, work);
  hipLaunchKernelGGL(pivchol_trace_kernel, dim3(1), dim3(64), 0, st, cand_sum, nwg, first_step + n_steps, state, work);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_householder_rows(double* LT, int64_t ldl, int32_t P, int32_t r, double* YT, double* tau, double* scratch, void* stream) {
  VMC_REQUIRE(LT && YT && tau && scratch, "vmcpde_householder_rows: null argument");
  VMC_REQUIRE(P >= 1 && r >= 1 && r <= P && ldl >= P, "vmcpde_householder_rows: bad shape");
  hipStream_t st = (hipStream_t)stream;
  double* norm2 = scratch;
  double* rblk = scratch + r + 1;
  // several steps per launch while a workgroup's registers hold the columns: two up to 16 384 rows, four up to 8192.  (Round 3, sequential
  // dot products: four measured no better than two -- 1.05 against 1.01 ms for 203 columns at P = 4213, 1.49 ms with one; the thirteen block
  // sums of a launch of four, two barriers of sixteen waves each, cost what the two saved launch boundaries do.)
  // Round 5: with the dot products of a launch batched (hqr_batched_kernel: 2 NR block-sum phases instead of 13 at NR = 4) four steps per
  // launch do pay; VMCPDE_HQR_BATCHED=0 restores the sequential form, VMCPDE_HQR_BLOCK caps the steps per launch (default 4).
  // (beyond 12 slices of 512 rows a lane -- P > 6145 -- the 1024-thread sequential pair kernel is faster: 1.33 against 1.51 ms at P = 8000)
  const bool batched = env_int("VMCPDE_HQR_BATCHED", 1) != 0 && P <= 12 * HQB_THREADS + 1;
  // (six per launch, tried: 120 + 20 doubles of slices a lane and the phase values no longer fit 256 registers -- 118 spilled, 1.70 ms)
  const int cap = env_int("VMCPDE_HQR_BLOCK", 4);      // (sequential form: fours pay from ~6000 rows on -- 1.25 against 1.36 ms at P = 8000 -- and cost 4 % below)
  const dim3 block(HQ_THREADS);
  int k = 0;
  if (batched) {
    // 512-thread workgroups (eight waves: cheaper block sums; 256 registers a lane hold four reflectors' and the own column's slices)
    const dim3 bblock(HQB_THREADS);
    const int need = (P - 1 + HQB_THREADS - 1) / HQB_THREADS;       // slices of 512 rows a thread holds of a column
#define VMC_HQRB(NRV)                                                                                                                        \
  do {                                                                                                                                       \
    if (need <= 4) hipLaunchKernelGGL((hqr_batched_kernel<NRV, 4, HQB_THREADS>), grid, bblock, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);   \
    else if (need <= 8) hipLaunchKernelGGL((hqr_batched_kernel<NRV, 8, HQB_THREADS>), grid, bblock, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk); \
    else if (need <= 10) hipLaunchKernelGGL((hqr_batched_kernel<NRV, 10, HQB_THREADS>), grid, bblock, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk); \
    else hipLaunchKernelGGL((hqr_batched_kernel<NRV, 12, HQB_THREADS>), grid, bblock, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);           \
  } while (0)
    if (cap >= 4) {
      const int k_begin = k;
      int nblocks = 0;
      for (; k + 3 < r; k += 4, ++nblocks) {
        const dim3 grid(r - k - 3);
        VMC_HQRB(4);
      }
      if (nblocks) hipLaunchKernelGGL(hqr_commit_kernel, dim3((nblocks * 4 * HQ_RB + 255) / 256), dim3(256), 0, st, LT, ldl, k_begin, 4, nblocks, rblk, HQ_RB);
    }
    if (cap >= 2) {
      const int k_begin = k;
      int nblocks = 0;
      for (; k + 1 < r; k += 2, ++nblocks) {
        const dim3 grid(r - k - 1);
        VMC_HQRB(2);
      }
      if (nblocks) hipLaunchKernelGGL(hqr_commit_kernel, dim3((nblocks * 2 * HQ_RB + 255) / 256), dim3(256), 0, st, LT, ldl, k_begin, 2, nblocks, rblk, HQ_RB);
    }
#undef VMC_HQRB
  } else {
    if (P <= 8 * HQ_THREADS && cap >= 4) {
      const int k_begin = k;
      int nblocks = 0;
      for (; k + 3 < r; k += 4, ++nblocks) {
        const dim3 grid(r - k - 3);
        if (P <= 4 * HQ_THREADS) hipLaunchKernelGGL((hqr_multi_kernel<4, 4>), grid, block, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);
        else hipLaunchKernelGGL((hqr_multi_kernel<4, 8>), grid, block, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);
      }
      if (nblocks) hipLaunchKernelGGL(hqr_commit_kernel, dim3((nblocks * 4 * HQ_RB + 255) / 256), dim3(256), 0, st, LT, ldl, k_begin, 4, nblocks, rblk, HQ_RB);
    }
    if (P <= HQ_MAXE_LIMIT * HQ_THREADS && cap >= 2) {
      const int k_begin = k;
      int nblocks = 0;
      for (; k + 1 < r; k += 2, ++nblocks) {
        const dim3 grid(r - k - 1);
        if (P <= 4 * HQ_THREADS) hipLaunchKernelGGL((hqr_multi_kernel<2, 4>), grid, block, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);
        else if (P <= 8 * HQ_THREADS) hipLaunchKernelGGL((hqr_multi_kernel<2, 8>), grid, block, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);
        else hipLaunchKernelGGL((hqr_multi_kernel<2, 16>), grid, block, 0, st, LT, ldl, P, k, YT, tau, norm2, rblk);
      }
      if (nblocks) hipLaunchKernelGGL(hqr_commit_kernel, dim3((nblocks * 2 * HQ_RB + 255) / 256), dim3(256), 0, st, LT, ldl, k_begin, 2, nblocks, rblk, HQ_RB);
    }
  }
  for (; k < r; ++k) hipLaunchKernelGGL(hqr_step_kernel, dim3(r - k), block, 0, st, LT, ldl, P, k, YT, tau, norm2);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t vmcpde_pivoted_cholesky_blocked_workspace_bytes(int32_t P) {
  if (P < 1) return 0;
  const int64_t ldl = (int64_t)((P + PC_ROWS - 1) / PC_ROWS) * PC_ROWS;
  return pivblk_ws_doubles(P, ldl) * (int64_t)sizeof(double);
}

extern "C" int vmcpde_pivoted_cholesky_blocked(const double* S, int64_t lds, int32_t P, double tol_rel, int32_t restart, int32_t n_blocks,
                                               int32_t max_rank, double* LT, int64_t ldl, int32_t* piv, int32_t* state, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(S && LT && piv && state && workspace, "vmcpde_pivoted_cholesky_blocked: null argument");
  VMC_REQUIRE(P >= 1 && lds >= P, "vmcpde_pivoted_cholesky_blocked: bad shape");
  const int nblk = (P + PC_ROWS - 1) / PC_ROWS;
  VMC_REQUIRE(2 * nblk <= PB_MAX_BLOCKS, "vmcpde_pivoted_cholesky_blocked: P <= 32768");
  VMC_REQUIRE(ldl == (int64_t)nblk * PC_ROWS, "vmcpde_pivoted_cholesky_blocked: ldl must be P rounded up to a multiple of 64");
  VMC_REQUIRE(n_blocks >= 0 && max_rank >= 1 && max_rank <= P, "vmcpde_pivoted_cholesky_blocked: bad step count / rank cap");
  VMC_REQUIRE(tol_rel >= 0.0, "vmcpde_pivoted_cholesky_blocked: negative tolerance");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_blocked_workspace_bytes(P), "vmcpde_pivoted_cholesky_blocked: workspace too small");
  double* w = static_cast<double*>(workspace);
  PivWork* work = reinterpret_cast<PivWork*>(w);
  double* d = w + 8;
  double* cand_val = d + (int64_t)nblk * PC_ROWS;
  double* cand_sum = cand_val + 2 * nblk;
  int* cand_idx = reinterpret_cast<int*>(cand_sum + 2 * nblk);
  PivDesc* desc = reinterpret_cast<PivDesc*>(w + pivchol_ws_doubles(P));
  double* AP = w + pivchol_ws_doubles(P) + 64;
  const int* hint = pivblk_hint(w, P, ldl);
  hipStream_t st = (hipStream_t)stream;
  if (restart) hipLaunchKernelGGL(pivchol_init_kernel, dim3(nblk), dim3(PC_ROWS), 0, st, S, lds, P, d, cand_val, cand_idx, cand_sum, state, 1, work);
  const int nfac = (P + PF_ROWS - 1) / PF_ROWS;
  const int diag_flags = env_int("VMCPDE_PIVBLK_DIAG", 0);
  for (int b = 0; b < n_blocks; ++b) {
    hipLaunchKernelGGL(pivblk_panel_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, S, lds, P, tol_rel, max_rank, LT, ldl, cand_val, cand_idx, cand_sum,
                       nblk, AP, desc, state, work, 2, (const double*)nullptr, d, hint);
    hipLaunchKernelGGL(pivblk_factor_kernel, dim3(nfac), dim3(PF_ROWS), 0, st, P, tol_rel, LT, ldl, d, cand_val, cand_idx, cand_sum, nblk, AP, desc,
                       piv, state, work, diag_flags);
  }
  // one more panel launch settles `state` after the last factor launch: done if nothing is left above the tolerance, and the trace of
  // the remainder (it selects, writes the descriptor and its panel, which the next call's first launch simply recomputes)
  hipLaunchKernelGGL(pivblk_panel_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, S, lds, P, tol_rel, max_rank, LT, ldl, cand_val, cand_idx, cand_sum, nblk,
                     AP, desc, state, work, 2, (const double*)nullptr, d, hint);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- matrix-free form (experiment): the same factorisation with the columns of S formed from the Jacobian on demand ----------------
extern "C" int64_t vmcpde_lowrank_candidates_byte_offset(int32_t P) {     // of the int32[32] candidate rows inside the blocked workspace
  return (P >= 1) ? (pivchol_ws_doubles(P) * (int64_t)sizeof(double) + 16) : 0;
}

extern "C" int vmcpde_lowrank_hint(const int32_t* piv, const int32_t* state, int32_t P, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(workspace && P >= 1 && (piv == nullptr || state != nullptr), "vmcpde_lowrank_hint: bad argument");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_blocked_workspace_bytes(P), "vmcpde_lowrank_hint: workspace too small");
  double* w = static_cast<double*>(workspace);
  const int64_t ldl = (int64_t)((P + PC_ROWS - 1) / PC_ROWS) * PC_ROWS;
  hipLaunchKernelGGL(pivblk_hint_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, piv, state, P, pivblk_hint(w, P, ldl), reinterpret_cast<PivWork*>(w));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void pivblk_columns_kernel(const double* big, int n_rows, PivWork* __restrict__ work) {
  work->big = n_rows > 0 ? big : nullptr;
  work->big_n = n_rows > 0 ? n_rows : 0;
}

extern "C" int64_t vmcpde_lowrank_skip_flag_byte_offset(void) { return (int64_t)offsetof(PivWork, skip); }

extern "C" int vmcpde_lowrank_columns(const double* Sbig, int32_t n_rows, int32_t P, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(workspace && P >= 1 && n_rows >= 0 && n_rows <= P && (Sbig != nullptr || n_rows == 0), "vmcpde_lowrank_columns: bad argument");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_blocked_workspace_bytes(P), "vmcpde_lowrank_columns: workspace too small");
  hipLaunchKernelGGL(pivblk_columns_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, Sbig, n_rows, reinterpret_cast<PivWork*>(workspace));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_lowrank_begin(const double* diag, int32_t P, int32_t* state, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(diag && state && workspace && P >= 1, "vmcpde_lowrank_begin: bad argument");
  const int nblk = (P + PC_ROWS - 1) / PC_ROWS;
  VMC_REQUIRE(2 * nblk <= PB_MAX_BLOCKS, "vmcpde_lowrank_begin: P <= 32768");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_blocked_workspace_bytes(P), "vmcpde_lowrank_begin: workspace too small");
  double* w = static_cast<double*>(workspace);
  double* d = w + 8;
  double* cand_val = d + (int64_t)nblk * PC_ROWS;
  double* cand_sum = cand_val + 2 * nblk;
  int* cand_idx = reinterpret_cast<int*>(cand_sum + 2 * nblk);
  hipLaunchKernelGGL(pivchol_init_diag_kernel, dim3(nblk), dim3(PC_ROWS), 0, (hipStream_t)stream, diag, P, d, cand_val, cand_idx, cand_sum, state, reinterpret_cast<PivWork*>(w));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// phase 0 (Scol NULL): choose the step's candidates (their rows: int32[32] at vmcpde_lowrank_candidates_byte_offset in the workspace);
// phase 1 (Scol = [32][ldl], Scol[m][i] = S[C_m][i]): panel, block factorisation, new columns
extern "C" int vmcpde_lowrank_step(const double* Scol, int32_t P, double tol_rel, int32_t max_rank, double* LT, int64_t ldl, int32_t* piv,
                                   int32_t* state, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(LT && piv && state && workspace && P >= 1, "vmcpde_lowrank_step: bad argument");
  const int nblk = (P + PC_ROWS - 1) / PC_ROWS;
  VMC_REQUIRE(ldl == (int64_t)nblk * PC_ROWS && max_rank >= 1 && max_rank <= P, "vmcpde_lowrank_step: bad shape");
  VMC_REQUIRE(workspace_bytes >= vmcpde_pivoted_cholesky_blocked_workspace_bytes(P), "vmcpde_lowrank_step: workspace too small");
  double* w = static_cast<double*>(workspace);
  PivWork* work = reinterpret_cast<PivWork*>(w);
  double* d = w + 8;
  double* cand_val = d + (int64_t)nblk * PC_ROWS;
  double* cand_sum = cand_val + 2 * nblk;
  int* cand_idx = reinterpret_cast<int*>(cand_sum + 2 * nblk);
  PivDesc* desc = reinterpret_cast<PivDesc*>(w + pivchol_ws_doubles(P));
  double* AP = w + pivchol_ws_doubles(P) + 64;
  const int* hint = pivblk_hint(w, P, ldl);
  hipStream_t st = (hipStream_t)stream;
  if (!Scol) {
    hipLaunchKernelGGL(pivblk_panel_kernel, dim3(1), dim3(PC_THREADS), 0, st, (const double*)nullptr, (int64_t)0, P, tol_rel, max_rank, LT, ldl, cand_val,
                       cand_idx, cand_sum, nblk, AP, desc, state, work, 0, (const double*)nullptr, d, hint);
  } else {
    hipLaunchKernelGGL(pivblk_panel_kernel, dim3(nblk), dim3(PC_THREADS), 0, st, (const double*)nullptr, (int64_t)0, P, tol_rel, max_rank, LT, ldl,
                       cand_val, cand_idx, cand_sum, nblk, AP, desc, state, work, 1, Scol, d, hint);
    hipLaunchKernelGGL(pivblk_factor_kernel, dim3((P + PF_ROWS - 1) / PF_ROWS), dim3(PF_ROWS), 0, st, P, tol_rel, LT, ldl, d, cand_val, cand_idx,
                       cand_sum, nblk, AP, desc, piv, state, work, 0);
  }
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t vmcpde_gram_columns_workspace_bytes(int32_t P, int64_t n_samples) {
  if (P < 1 || n_samples < 1) return 0;
  const int64_t ldl = (int64_t)((P + PC_ROWS - 1) / PC_ROWS) * PC_ROWS;
  return CP_MAX_CHUNKS * (int64_t)PB * ldl * (int64_t)sizeof(double);
}

// Scol[m][i] = sum_n JT[rows[m]][n] JT[i][n] / n_total - mean[i] mean[rows[m]] for 32 rows (mean NULL: no centring); one pass over JT
extern "C" int vmcpde_gram_columns(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const int32_t* rows32, const double* mean,
                                   double n_total, double* Scol, int64_t ldl, const int32_t* state, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
  VMC_REQUIRE(JT && rows32 && Scol && workspace, "vmcpde_gram_columns: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1 && ldj >= n_samples && n_total > 0, "vmcpde_gram_columns: bad shape");
  const int nblk = (P + PC_ROWS - 1) / PC_ROWS;
  VMC_REQUIRE(ldl == (int64_t)nblk * PC_ROWS, "vmcpde_gram_columns: ldl must be P rounded up to a multiple of 64");
  VMC_REQUIRE(workspace_bytes >= vmcpde_gram_columns_workspace_bytes(P, n_samples), "vmcpde_gram_columns: workspace too small");
  // sample chunks so that the workgroups fill whole rounds of the chip (six 25 KB workgroups per CU: 1536 at a time -- 66 row blocks x
  // 32 chunks = 2112 left the second round 37 % full), at least ~1024 samples each
  int nchunk = (int)(CP_RESIDENT / nblk);
  if (nchunk < 1) nchunk = 1;
  while (nchunk > 1 && n_samples / nchunk < 1024) --nchunk;
  if (n_samples / nchunk > 4096 && 2 * nchunk <= CP_MAX_CHUNKS) nchunk = (int)(2 * CP_RESIDENT / nblk);
  if (nchunk > CP_MAX_CHUNKS) nchunk = CP_MAX_CHUNKS;
  int64_t chunk = ((n_samples + nchunk - 1) / nchunk + CP_BK - 1) / CP_BK * CP_BK;
  nchunk = (int)((n_samples + chunk - 1) / chunk);
  double* part = static_cast<double*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colpanel_kernel, dim3(nblk, nchunk), dim3(CP_THREADS), 0, st, JT, ldj, n_samples, rows32, chunk, part, ldl, state);
  const int64_t total = (int64_t)PB * ldl;
  hipLaunchKernelGGL(colpanel_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, nchunk, ldl, P, rows32, mean, 1.0 / n_total,
                     Scol, state);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// y = JT (JT^T v) for the P x n_samples Jacobian block JT (row stride ldj): S0 v = y / N - m (m.v) is the caller's.  z (n_samples doubles) holds
// JT^T v afterwards.  workspace: vmcpde_jt_matvec_workspace_bytes(P, n_samples).
extern "C" int64_t vmcpde_jt_matvec_workspace_bytes(int32_t P, int64_t n_samples) {
  if (P < 1 || n_samples < 1) return 0;
  const int64_t ldz = (n_samples + 63) / 64 * 64;
  return (int64_t)MV_MAX_SLICES * ldz * (int64_t)sizeof(double);
}

extern "C" int vmcpde_jt_matvec(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const double* v, double* z, double* y, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(JT && v && z && workspace, "vmcpde_jt_matvec: null argument");     // (y == NULL: z = JT^T v only, one pass)
  VMC_REQUIRE(P >= 1 && n_samples >= 1 && ldj >= n_samples, "vmcpde_jt_matvec: bad shape");
  VMC_REQUIRE(workspace_bytes >= vmcpde_jt_matvec_workspace_bytes(P, n_samples), "vmcpde_jt_matvec: workspace too small");
  const int64_t ldz = (n_samples + 63) / 64 * 64;
  const int64_t colblocks = (n_samples + MV_COLS - 1) / MV_COLS;
  // row slices so that ~4096 workgroups are in flight (sixteen 4-wave workgroups per CU), each with at least 32 rows
  int nslice = (int)((4096 + colblocks - 1) / colblocks);
  if (nslice > MV_MAX_SLICES) nslice = MV_MAX_SLICES;
  if (nslice > (P + 31) / 32) nslice = (P + 31) / 32;
  if (nslice < 1) nslice = 1;
  const int rows_per_slice = (P + nslice - 1) / nslice;
  nslice = (P + rows_per_slice - 1) / rows_per_slice;
  const int vec2 = ((ldj & 1) == 0 && (reinterpret_cast<uintptr_t>(JT) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0) ? 1 : 0;
  double* zpart = static_cast<double*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(matvec_cols_kernel, dim3((unsigned)colblocks, (unsigned)nslice), dim3(MV_THREADS), 0, st, JT, ldj, P, n_samples, v, rows_per_slice,
                     zpart, ldz, vec2);
  hipLaunchKernelGGL(matvec_cols_reduce_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, st, zpart, nslice, ldz, n_samples, z);
  if (y) hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)P), dim3(256), 0, st, JT, ldj, P, n_samples, z, y, vec2);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- the Woodbury application of the low-rank regularised solve (tdvp.py:73-94 with the factor of csrc/eig.hip) ---------------------------
// M = D + L L^T,  M^-1 v = y - D^-1 L G^-1 L^T y  with  y = D^-1 v  and  G = I + L^T D^-1 L (factored by the caller).
namespace {

// t1[k] = sum_p LT[k][p] y[p] with y = v / D and v = F (u == nullptr) or the residual F - Su - D u of the full system at u; workgroup 0
// also writes y.  One workgroup per column of L (row of LT: contiguous).
__global__ __launch_bounds__(256) void woodbury_project_kernel(const double* __restrict__ LT, int64_t ldl, int P, const double* __restrict__ D,
                                                               const double* __restrict__ F, const double* __restrict__ Su,
                                                               const double* __restrict__ u, double* __restrict__ y, double* __restrict__ t1) {
  __shared__ double part[4];
  const int k = (int)blockIdx.x, t = (int)threadIdx.x;
  const double* __restrict__ row = LT + (int64_t)k * ldl;
  double s = 0.0;
  for (int p = t; p < P; p += 256) {
    const double d = D[p];
    const double v = u ? (F[p] - Su[p] - d * u[p]) : F[p];
    const double yy = v / d;
    if (k == 0) y[p] = yy;
    s = fma(row[p], yy, s);
  }
  s = vmcpde::wave_sum(s);
  if ((t & 63) == 0) part[t >> 6] = s;
  __syncthreads();
  if (t == 0) t1[k] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[p] = (u ? u[p] : 0) + y[p] - (sum_k LT[k][p] t2[k]) / D[p]: 64 parameters per workgroup, the r rows of LT (read coalesced) dealt to the four
// waves, wave partials added in a fixed order (round 5 ran ONE wave per 64 parameters over all r rows: 21 us for 8.6 MB; this form has 4x the loads in flight)
__global__ __launch_bounds__(256) void woodbury_combine_kernel(const double* __restrict__ LT, int64_t ldl, int P, int r, const double* __restrict__ D,
                                                               const double* __restrict__ y, const double* __restrict__ t2,
                                                               const double* u, double* out /* may alias u: no __restrict__ on the pair */) {
  __shared__ double part[4][64];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int p = (int)blockIdx.x * 64 + lane;
  const bool in = p < P;
  const int pp = in ? p : P - 1;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = wave;
  for (; k + 12 < r; k += 16) {
    a0 = fma(LT[(int64_t)k * ldl + pp], t2[k], a0);
    a1 = fma(LT[(int64_t)(k + 4) * ldl + pp], t2[k + 4], a1);
    a2 = fma(LT[(int64_t)(k + 8) * ldl + pp], t2[k + 8], a2);
    a3 = fma(LT[(int64_t)(k + 12) * ldl + pp], t2[k + 12], a3);
  }
  for (; k < r; k += 4) a0 = fma(LT[(int64_t)k * ldl + pp], t2[k], a0);
  part[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && in) {
    const double corr = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / D[p];
    out[p] = (u ? u[p] : 0.0) + y[p] - corr;
  }
}

}  // namespace

// ---- ABI 7: what surrounds the Woodbury application in a static call, without a library GEMM or an elementwise launch ---------------------
namespace {

typedef double wg_double4 __attribute__((ext_vector_type(4)));
// block index t of the lower triangle (row-major: (0,0), (1,0), (1,1), (2,0), ...) -> (bi, bj), bi >= bj
__device__ __forceinline__ void wg_tri_decode(int t, int& bi, int& bj) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  bi = i;
  bj = t - i * (i + 1) / 2;
}
constexpr int WG_BLK = 64;      // G is formed in 64 x 64 blocks of its lower triangle ...
constexpr int WG_KC = 256;      // ... from contraction chunks of this many parameters: (r / 64)(r / 64 + 1) / 2 x ceil(ldl / 256) workgroups

// D[p] = shift * diagS[p] + floor and 1 / D[p] (zero on the pad [P, ldl): pad columns of LT never count), and -- behind the first
// ceil(ldl / 256) workgroups -- one workgroup per row k of LT[:rmax] that zeroes the row when k >= rank (state[0], read on the device): the
// factorisation rewrites rows [0, rank) in full, rows behind it still hold an earlier call's columns and must be an identity block of G.
__global__ __launch_bounds__(256) void woodbury_prepare_kernel(const double* __restrict__ diagS, double shift, const double* __restrict__ floor_dev, int P,
                                                               int ldl, double* __restrict__ D, double* __restrict__ invD, double* __restrict__ LT,
                                                               const int32_t* __restrict__ state, int rmax, int nD) {
  const int b = (int)blockIdx.x, t = (int)threadIdx.x;
  if (b < nD) {
    const int p = b * 256 + t;
    if (p >= ldl) return;
    if (p < P) {
      const double d = fma(shift, diagS[p], floor_dev ? floor_dev[0] : 0.0);
      D[p] = d;
      invD[p] = 1.0 / d;
    } else {
      invD[p] = 0.0;
    }
    return;
  }
  const int k = b - nD;
  if (k < state[0] || k >= rmax) return;
  double* __restrict__ row = LT + (int64_t)k * ldl;
  for (int i = t; i < ldl; i += 256) row[i] = 0.0;
}

// partial[chunk][blk] = LT[rows of bi][chunk] diag(invD[chunk]) LT[rows of bj][chunk]^T for the 64 x 64 block (bi >= bj) of G: operands straight from
// global memory -- the factor is 8.6 MB and was written by the launches just before, 170 workgroups hide the latency; no LDS image, no barrier.
__global__ __launch_bounds__(256) void woodbury_gram_partial_kernel(const double* __restrict__ LT, int64_t ldl, const double* __restrict__ invD, int nblk,
                                                                    double* __restrict__ slabs) {
  int bi, bj;
  wg_tri_decode((int)blockIdx.y, bi, bj);
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int k0 = (int)blockIdx.x * WG_KC;
  const int kn = ((int)ldl - k0 < WG_KC) ? ((int)ldl - k0) : WG_KC;
  // a lane loads FOUR consecutive parameters of its row (32 B; the 16 rows x 4 lanes of an operand fragment then read 128-B runs) and feeds
  // element j of them to MFMA step j: the step contracts over the parameters 16 s + 4 (l >> 4) + j -- A, B and 1 / D agree on that assignment
  const double* __restrict__ ap = LT + (int64_t)(bi * WG_BLK + wr * 32 + lr) * ldl + k0 + 4 * lk;
  const double* __restrict__ bp = LT + (int64_t)(bj * WG_BLK + wc * 32 + lr) * ldl + k0 + 4 * lk;
  const double* __restrict__ dp = invD + k0 + 4 * lk;
  wg_double4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (wg_double4){0.0, 0.0, 0.0, 0.0};
  for (int k32 = 0; k32 < kn; k32 += 32) {            // (ldl is a multiple of 32: two 16-parameter steps' operands in flight together)
    wg_double4 w[2], a0[2], a1[2], b0[2], b1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k32 + 16 * u;
      w[u] = *reinterpret_cast<const wg_double4*>(dp + k);
      a0[u] = *reinterpret_cast<const wg_double4*>(ap + k);
      a1[u] = *reinterpret_cast<const wg_double4*>(ap + 16 * ldl + k);
      b0[u] = *reinterpret_cast<const wg_double4*>(bp + k);
      b1[u] = *reinterpret_cast<const wg_double4*>(bp + 16 * ldl + k);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double x0 = a0[u][j] * w[u][j], x1 = a1[u][j] * w[u][j];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b0[u][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b1[u][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b0[u][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b1[u][j], acc[1][1], 0, 0, 0);
      }
  }
  double* __restrict__ out = slabs + ((int64_t)blockIdx.x * nblk + blockIdx.y) * (WG_BLK * WG_BLK);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)          // C / D layout of the f64 MFMA: row = (lane >> 4) + 4 q, column = lane & 15
        out[(wr * 32 + 16 * i + lk + 4 * q) * WG_BLK + wc * 32 + 16 * j + lr] = acc[i][j][q];
}

// G = I + sum over the chunks in a fixed order (deterministic: every rank forms the same bits), both triangles written
__global__ __launch_bounds__(256) void woodbury_gram_reduce_kernel(const double* __restrict__ slabs, int nchunk, int nblk, double* __restrict__ G, int64_t ldg) {
  int bi, bj;
  wg_tri_decode((int)blockIdx.y, bi, bj);
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;      // element of the 64 x 64 block
  const int r = e >> 6, c = e & 63;
  // (a diagonal block is computed in full, but (a w) b and (b w) a round differently: its upper triangle takes the lower one's sums, so G is symmetric to the bit)
  const int es = (bi == bj && c > r) ? (c * WG_BLK + r) : e;
  const double* __restrict__ src = slabs + (int64_t)blockIdx.y * (WG_BLK * WG_BLK) + es;
  double s = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) s += src[(int64_t)ch * nblk * (WG_BLK * WG_BLK)];
  const int gi = bi * WG_BLK + r, gj = bj * WG_BLK + c;
  if (gi == gj) s += 1.0;
  G[(int64_t)gi * ldg + gj] = s;
  if (bi != bj) G[(int64_t)gj * ldg + gi] = s;
}

struct GatherArgs {
  const unsigned long long* src[8];
  int n[8];
};
// dst = the concatenation of up to eight small device arrays of 8-byte words: the ONE record a static call hands to the host
__global__ __launch_bounds__(64) void gather_words_kernel(GatherArgs a, unsigned long long* __restrict__ dst) {
  int off = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int n = a.n[s];
    if (n > 0 && a.src[s] != nullptr)
      for (int i = (int)threadIdx.x; i < n; i += 64) dst[off + i] = a.src[s][i];
    off += n > 0 ? n : 0;
  }
}

}  // namespace

// ---- ABI 7: the factor of S on a FIXED pivot set (the previous factorisation's), without the pivot search ----------------------------------------
// Consecutive S of a trajectory differ by sampling noise and a small step of theta: the parameters that spanned the numerical range of the last S
// span this one's.  With Pi = the previous pivots in their order, S = L L^T + R with L = S[:, Pi] C^-T, C C^T = S[Pi, Pi] (the partial Cholesky on
// that pivot set; R = the Schur complement, positive semi-definite) needs no search: gather, one tall Cholesky (vmcpde_cholesky_tall), transpose.
// R is measured (trace, largest entry) and reported with the call's health record: the host decides one call later whether the set still holds.
namespace {

// T = [ C ; B^T ]: rows [0, rmax): C[a][b] = S[piv_a][piv_b] (identity beyond r); rows rmax + i: S[i][piv_b] = S[piv_b][i] (zero for b >= r).
// blockIdx.y = 64-column block of T (pivot positions b); blockIdx.x < nI: the rows rmax + 64 it .. of the lower part, read as 64-long runs of the
// rows piv_b of S (coalesced) and transposed through the LDS; blockIdx.x >= nI: a 64 x 64 block of C, gathered entry by entry.
__global__ __launch_bounds__(256) void frozen_gather_kernel(const double* __restrict__ S, int64_t lds, int P, const int32_t* __restrict__ piv, int r, int rmax,
                                                            double* __restrict__ T, int64_t ldt, int nI) {
  __shared__ double tile[64][65];
  __shared__ int rows[64];
  const int t = (int)threadIdx.x, bt = (int)blockIdx.y;
  if (t < 64) rows[t] = (bt * 64 + t < r) ? piv[bt * 64 + t] : -1;
  __syncthreads();
  if ((int)blockIdx.x < nI) {
    const int i0 = (int)blockIdx.x * 64;
    const int ii = t & 63;
    for (int bb = t >> 6; bb < 64; bb += 4) {
      const int row = rows[bb];
      tile[bb][ii] = (row >= 0 && i0 + ii < P) ? S[(int64_t)row * lds + i0 + ii] : 0.0;
    }
    __syncthreads();
    const int b = t & 63;
    for (int q = t >> 6; q < 64; q += 4)
      if (i0 + q < P) T[(int64_t)(rmax + i0 + q) * ldt + bt * 64 + b] = tile[b][q];
    return;
  }
  const int a0 = ((int)blockIdx.x - nI) * 64;
  const int b = t & 63;
  for (int q = t >> 6; q < 64; q += 4) {
    const int a = a0 + q, gb = bt * 64 + b;
    const int ra = (a < r) ? piv[a] : -1, rb = rows[b];
    T[(int64_t)a * ldt + gb] = (ra >= 0 && rb >= 0) ? S[(int64_t)ra * lds + rb] : ((a == gb) ? 1.0 : 0.0);
  }
}

// LT[b][i] = T[rmax + i][b] (pad columns [P, ldl) zero) and left[i] = S[i][i] - sum_b T[rmax + i][b]^2, the diagonal of the remainder R.
// One workgroup per 64 parameters i, all rmax columns: the row sums of squares are complete inside the workgroup (fixed order).
__global__ __launch_bounds__(256) void frozen_finish_kernel(const double* __restrict__ T, int64_t ldt, const double* __restrict__ S, int64_t lds, int P,
                                                            int rmax, double* __restrict__ LT, int64_t ldl, double* __restrict__ left) {
  __shared__ double tile[64][65];
  __shared__ double part[64][5];
  const int t = (int)threadIdx.x, i0 = (int)blockIdx.x * 64;
  const int row = t >> 2, seg = (t & 3) * 16;
  double sq = 0.0;
  for (int b0 = 0; b0 < rmax; b0 += 64) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double v = (i0 + row < P) ? T[(int64_t)(rmax + i0 + row) * ldt + b0 + seg + c] : 0.0;
      tile[row][seg + c] = v;
      sq = fma(v, v, sq);
    }
    __syncthreads();
    // thread (b = t >> 2, 16 parameters from seg): row b0 + b of LT
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (i0 + seg + c < ldl) LT[(int64_t)(b0 + row) * ldl + i0 + seg + c] = tile[seg + c][row];
  }
  part[row][t & 3] = sq;
  __syncthreads();
  if (t < 64) {
    const int i = i0 + t;
    if (i < P) left[i] = S[(int64_t)i * lds + i] - ((part[t][0] + part[t][1]) + (part[t][2] + part[t][3]));
  }
}

// trace and maximum of the remainder's diagonal, min D; the factorisation's bookkeeping as a pivoted run would leave it: state = (r, done),
// PivWork::trace_left.  out = { trace(R) / min D, pinned pivots of the tall Cholesky, max_i R_ii }.  One workgroup, fixed order.
__global__ __launch_bounds__(1024) void frozen_check_kernel(const double* __restrict__ left, int P, const double* __restrict__ D, int r,
                                                            const int32_t* __restrict__ chol_info, int32_t* __restrict__ state, PivWork* __restrict__ work,
                                                            double* __restrict__ out) {
  __shared__ double s_sum[16], s_max[16], s_min[16];
  const int t = (int)threadIdx.x;
  double sum = 0.0, mx = 0.0, mn = __builtin_huge_val();
  for (int i = t; i < P; i += 1024) {
    const double v = left[i];
    sum += (v > 0.0) ? v : 0.0;
    mx = fmax(mx, v);
    mn = fmin(mn, D[i]);
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o)); mn = fmin(mn, __shfl_xor(mn, o)); }
  if ((t & 63) == 0) { s_sum[t >> 6] = sum; s_max[t >> 6] = mx; s_min[t >> 6] = mn; }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0, c = __builtin_huge_val();
    for (int k = 0; k < 16; ++k) { a += s_sum[k]; b = fmax(b, s_max[k]); c = fmin(c, s_min[k]); }
    state[0] = r;
    state[1] = 1;
    work->trace_left = a;
    out[0] = (c > 0.0) ? a / c : __builtin_huge_val();
    out[1] = (double)chol_info[0];
    out[2] = b;
  }
}

}  // namespace

extern "C" int vmcpde_frozen_gather(const double* S, int64_t lds, int32_t P, const int32_t* piv, int32_t r, int32_t rmax, double* T, int64_t ldt,
                                    void* stream) {
  VMC_REQUIRE(S && piv && T, "vmcpde_frozen_gather: null argument");
  VMC_REQUIRE(P >= 1 && lds >= P && r >= 1 && r <= rmax && r <= P && (rmax % 64) == 0 && ldt >= rmax, "vmcpde_frozen_gather: bad shape");
  const int nI = (P + 63) / 64;
  hipLaunchKernelGGL(frozen_gather_kernel, dim3((unsigned)(nI + rmax / 64), (unsigned)(rmax / 64)), dim3(256), 0, (hipStream_t)stream, S, lds, P, piv, r, rmax,
                     T, ldt, nI);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_frozen_finish(const double* T, int64_t ldt, const double* S, int64_t lds, int32_t P, int32_t rmax, double* LT, int64_t ldl,
                                    double* left, void* stream) {
  VMC_REQUIRE(T && S && LT && left, "vmcpde_frozen_finish: null argument");
  VMC_REQUIRE(P >= 1 && lds >= P && (rmax % 64) == 0 && rmax >= 64 && ldt >= rmax && ldl >= P && (ldl % 64) == 0, "vmcpde_frozen_finish: bad shape");
  hipLaunchKernelGGL(frozen_finish_kernel, dim3((unsigned)(ldl / 64)), dim3(256), 0, (hipStream_t)stream, T, ldt, S, lds, P, rmax, LT, ldl, left);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_frozen_check(const double* left, int32_t P, const double* D, int32_t r, const int32_t* chol_info, int32_t* state, void* workspace,
                                   double* out, void* stream) {
  VMC_REQUIRE(left && D && chol_info && state && workspace && out && P >= 1, "vmcpde_frozen_check: bad argument");
  hipLaunchKernelGGL(frozen_check_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, left, P, D, r, chol_info, state, static_cast<PivWork*>(workspace), out);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_woodbury_prepare(const double* diagS, double shift, const double* floor_dev, int32_t P, int64_t ldl, double* D, double* invD,
                                       double* LT, const int32_t* state, int32_t rmax, void* stream) {
  VMC_REQUIRE(diagS && D && invD && LT && state, "vmcpde_woodbury_prepare: null argument");
  VMC_REQUIRE(P >= 1 && ldl >= P && ldl < (1 << 30) && rmax >= 1, "vmcpde_woodbury_prepare: bad shape");
  const int nD = (int)((ldl + 255) / 256);
  hipLaunchKernelGGL(woodbury_prepare_kernel, dim3((unsigned)(nD + rmax)), dim3(256), 0, (hipStream_t)stream, diagS, shift, floor_dev, P, (int)ldl, D, invD, LT,
                     state, rmax, nD);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t vmcpde_woodbury_gram_workspace_bytes(int64_t ldl, int32_t r) {
  if (ldl <= 0 || r <= 0) return 0;
  const int nside = (r + WG_BLK - 1) / WG_BLK;
  return (int64_t)((ldl + WG_KC - 1) / WG_KC) * (nside * (nside + 1) / 2) * WG_BLK * WG_BLK * (int64_t)sizeof(double);
}

extern "C" int vmcpde_woodbury_gram(const double* LT, int64_t ldl, int32_t r, const double* invD, double* G, int64_t ldg, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(LT && invD && G && workspace, "vmcpde_woodbury_gram: null argument");
  VMC_REQUIRE(r >= WG_BLK && (r % WG_BLK) == 0 && ldg >= r, "vmcpde_woodbury_gram: r must be a positive multiple of %d and ldg >= r", WG_BLK);
  VMC_REQUIRE(ldl >= 32 && (ldl % 32) == 0 && ldl < (1 << 30), "vmcpde_woodbury_gram: ldl must be a multiple of 32");
  VMC_REQUIRE(workspace_bytes >= vmcpde_woodbury_gram_workspace_bytes(ldl, r), "vmcpde_woodbury_gram: workspace too small");
  const int nside = r / WG_BLK, nblk = nside * (nside + 1) / 2, nchunk = (int)((ldl + WG_KC - 1) / WG_KC);
  hipStream_t st = (hipStream_t)stream;
  double* slabs = static_cast<double*>(workspace);
  hipLaunchKernelGGL(woodbury_gram_partial_kernel, dim3((unsigned)nchunk, (unsigned)nblk), dim3(256), 0, st, LT, ldl, invD, nblk, slabs);
  hipLaunchKernelGGL(woodbury_gram_reduce_kernel, dim3(WG_BLK * WG_BLK / 256, (unsigned)nblk), dim3(256), 0, st, slabs, nchunk, nblk, G, ldg);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_gather_words(const void* const* srcs, const int32_t* counts, int32_t n_src, void* dst, void* stream) {
  VMC_REQUIRE(srcs && counts && dst && n_src >= 1 && n_src <= 8, "vmcpde_gather_words: 1..8 sources");
  GatherArgs a;
  for (int s = 0; s < 8; ++s) {
    a.src[s] = s < n_src ? static_cast<const unsigned long long*>(srcs[s]) : nullptr;
    a.n[s] = s < n_src ? counts[s] : 0;
    VMC_REQUIRE(a.n[s] >= 0 && a.n[s] <= 4096, "vmcpde_gather_words: a source has %d words", a.n[s]);
  }
  hipLaunchKernelGGL(gather_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, static_cast<unsigned long long*>(dst));
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_woodbury_project(const double* LT, int64_t ldl, int32_t P, int32_t r, const double* D, const double* F, const double* Su,
                                       const double* u, double* y, double* t1, void* stream) {
  VMC_REQUIRE(LT && D && F && y && t1, "vmcpde_woodbury_project: null argument");
  VMC_REQUIRE((u == nullptr) == (Su == nullptr), "vmcpde_woodbury_project: u and Su come together");
  VMC_REQUIRE(P >= 1 && r >= 1 && ldl >= P, "vmcpde_woodbury_project: bad shape");
  hipLaunchKernelGGL(woodbury_project_kernel, dim3((unsigned)r), dim3(256), 0, (hipStream_t)stream, LT, ldl, P, D, F, Su, u, y, t1);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_woodbury_combine(const double* LT, int64_t ldl, int32_t P, int32_t r, const double* D, const double* y, const double* t2,
                                       const double* u, double* out, void* stream) {
  VMC_REQUIRE(LT && D && y && t2 && out, "vmcpde_woodbury_combine: null argument");
  VMC_REQUIRE(P >= 1 && r >= 1 && ldl >= P, "vmcpde_woodbury_combine: bad shape");
  hipLaunchKernelGGL(woodbury_combine_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, (hipStream_t)stream, LT, ldl, P, r, D, y, t2, u, out);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}
