// Gram matrix S = <O~ O~^T> and force vector F = <E~ O~> of the TDVP equation on fp64 MFMA.
//
// Replaces tdvp.py:36-52 (get_tdvp_equation), the pmapped helpers tdvp.py:30-33 (J7-J9) and the reducers
// mpi_wrapper.py:21-25, 61-63, 129-274 (J11-J13).  The reference materialises four more N x P arrays (centred
// O, E*O, logp*O) and runs two un-symmetrised N x P x P GEMMs; here the centring (and the optional per-sample
// weight) is fused into the operand staging of ONE syrk-shaped kernel over the lower-triangle tiles.
//
// Data layout: JT[p][n] (parameter-major, K = sample index contiguous) -- both MFMA operands of
// S[i][j] = sum_n JT[i][n] JT[j][n] are then "K-contiguous rows", i.e. a TN GEMM with A = B = JT.
//
// gfx950 mapping: v_mfma_f64_16x16x4_f64 (one f64 of A and of B per lane: A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15]; C/D: col = lane&15, row = (lane>>4) + 4*reg).  Workgroup = 4 waves (2x2), tile
// 128x128, wave tile 64x64 = 4x4 MFMA blocks = 128 accumulator VGPRs, so two workgroups are resident per CU
// (two waves per SIMD keep the matrix pipe busy while the partner stages).  K-step 16: operands are staged
// global -> registers (prefetched one K-step ahead, centred/weighted in flight) -> LDS rows of 18 doubles
// (144 B pitch: the 16 rows x 2 k-columns a ds_read_b64 wave-instruction touches fall on 64 distinct banks).
// Split-K over the sample axis gives >= 8 work units per CU-slot; partial tiles go to a workspace and are summed by
// a deterministic second kernel (no atomics: bitwise reproducible, which the replicated multi-GPU solve relies on).
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "mfma_tile.hpp"
#include "gram_order.hpp"

using namespace vmcpde;

namespace {

#ifndef VMC_GRAM_BKD
#define VMC_GRAM_BKD 16
#endif
// K step of the DMA-staged Gram kernel: 16 -> 64 KiB of LDS, two workgroups per CU; 8 -> 32 KiB and <= 168 VGPRs, three.
// Measured equal (18.7 ms both on the same box): the kernel sits on the chip's power-managed fp64 MFMA rate, not on issue
// slots, so the smaller split count and workspace of 16 win.
constexpr int GRAM_BKD = VMC_GRAM_BKD;
constexpr int GRAM_FUSED_REDUCE_DEFAULT = 0;      // (round 6 A/B: see the fused split reduction in gram_dma_kernel)

struct GramPlan {
  int nb;          // row blocks of 128
  int ntiles;      // nb(nb+1)/2
  int nsplit;      // split-K factor over samples
  int64_t chunk;   // samples per split (multiple of BK)
};

constexpr int DEFAULT_ORDER = 2;
int gram_order() { return env_int("VMCPDE_GRAM_ORDER", DEFAULT_ORDER); }

GramPlan make_plan(int P, int64_t N, int slots /* resident workgroups: 256 CUs x 2 (register-staged) or x 3 (DMA) */) {
  GramPlan g;
  g.nb = (P + TILE - 1) / TILE;
  g.ntiles = (int)tri_count(g.nb);
  // aim at >= 8 rounds of the resident workgroup slots but keep each unit >= 64 K-steps so prologue/epilogue stay negligible;
  // among the neighbouring split factors take the one whose LAST round is fullest.  Rounds are counted the way the work is dealt
  // out (gram_unit): with order 1 every XCD owns ceil(ntiles/8) tiles of each split and has slots/8 resident workgroups, so the
  // count that matters is per XCD -- at C2 (561 tiles) 10 splits are 710 units on 64 slots = 11.09 rounds, i.e. a twelfth, nearly
  // empty round, while 9 splits are 9.98 rounds (measured 17.25 ms against 17.88 ms).
  const int64_t ksteps = (N + BK - 1) / BK;
  int64_t want = (8 * (int64_t)slots + g.ntiles - 1) / g.ntiles;
  int64_t maxsplit = ksteps / 64;
  if (maxsplit < 1) maxsplit = 1;
  if (maxsplit > 64) maxsplit = 64;
  int64_t ns = want < maxsplit ? want : maxsplit;
  if (ns < 1) ns = 1;
  if (ns > 1) {
    const int order = gram_order();
    const int64_t dom_slots = order >= 1 ? slots / 8 : slots;
    double best = 1e30;
    int64_t pick = ns;
    for (int64_t c = (ns > 3 ? ns - 2 : 1); c <= ns + 10 && c <= maxsplit; ++c) {
      // units of the busiest scheduling domain: order 1 -- ceil(ntiles/8) tiles of every split; order 2 -- balanced to one unit
      const int64_t units = order == 1 ? (int64_t)((g.ntiles + 7) / 8) * c : (order == 2 ? (g.ntiles * c + 7) / 8 : g.ntiles * c);
      const int64_t rounds = (units + dom_slots - 1) / dom_slots;
      const double cost = (double)rounds / (double)c;                      // time ~ rounds x (samples per unit ~ 1/c)
      if (cost < best - 1e-9) { best = cost; pick = c; }
    }
    ns = pick;
  }
  ns = env_int("VMCPDE_GRAM_NSPLIT", (int)ns);   // tuning knob
  int64_t steps_per = (ksteps + ns - 1) / ns;
  g.chunk = steps_per * BK;
  g.nsplit = (int)((N + g.chunk - 1) / g.chunk);
  if (g.nsplit < 1) g.nsplit = 1;
  return g;
}

// centred / weighted / fp32-operand variants: register-staged main loop, two workgroups per CU
template <bool CENTER, bool SCALE, typename TIn>
__global__ __launch_bounds__(256, 2) void gram_kernel(const TIn* __restrict__ JT, int64_t ldj, int P, int64_t N,
                                                      const double* __restrict__ c, const double* __restrict__ scale,
                                                      int ntiles, int nsplit, int64_t chunk, int total_units,
                                                      int units_per_xcd, int order, const int* __restrict__ perm, double* __restrict__ out) {
  extern __shared__ double lds[];
  int split, tile;
  if (!gram_unit(ntiles, nsplit, total_units, units_per_xcd, order, perm, split, tile)) return;
  int ti, tj;
  tri_decode(tile, ti, tj);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = (k0 + chunk < N) ? (k0 + chunk) : N;
  const int rowA0 = ti * TILE, rowB0 = tj * TILE;
  const int rowsA = (P - rowA0 < TILE) ? (P - rowA0) : TILE;
  const int rowsB = (P - rowB0 < TILE) ? (P - rowB0) : TILE;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop<CENTER, SCALE, TIn>(acc, JT, ldj, rowA0, JT, ldj, rowB0, ti == tj, k0, k1, c, rowsA, c, rowsB, scale, lds);
  tile_store(acc, out + ((int64_t)split * ntiles + tile) * (TILE * TILE));
}

// the steady-state variant (fp64, uncentred, unweighted): DMA-staged main loop
// (ntiles: the tiles this launch deals out; ntiles_out: the tiles of the whole lower triangle = the slab pitch -- they differ only for
// vmcpde_gram_rows, whose perm holds the global indices of the tiles of a few tile rows)
__global__ __launch_bounds__(256, GRAM_BKD == 8 ? 3 : 2) void gram_dma_kernel(const double* __restrict__ JT, int64_t ldj, int64_t N, int ntiles, int nsplit,
                                                          int64_t chunk, int total_units, int units_per_xcd, int order,
                                                          const int* __restrict__ perm, double* __restrict__ out, int ntiles_out, int P,
                                                          unsigned int* __restrict__ counters, double* __restrict__ packed) {
  extern __shared__ double lds[];
  int split, tile;
  if (!gram_unit(ntiles, nsplit, total_units, units_per_xcd, order, perm, split, tile)) return;
  int ti, tj;
  tri_decode(tile, ti, tj);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = (k0 + chunk < N) ? (k0 + chunk) : N;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // (rows of the tile's A block below P are zero padding: a wave that holds nothing else idles -- the last tile row when P ends in the first half of a tile)
  tile_mainloop_dma<GRAM_BKD>(acc, JT, ldj, ti * TILE, JT, ldj, tj * TILE, ti == tj, k0, k1, lds, -1, P - ti * TILE);
  tile_store(acc, out + ((int64_t)split * ntiles_out + tile) * (TILE * TILE));
  if (counters == nullptr) return;
  // Fused split reduction (round 6, VMCPDE_GRAM_FUSED_REDUCE): the unit that finishes LAST among the nsplit units of a tile adds the tile's slabs in
  // the fixed order 0 .. nsplit-1 -- the sums reduce_splits_kernel forms, bit for bit, whoever is last -- while the other workgroups are still on the
  // matrix pipe.  Release (this unit's slab is visible device-wide) -> count -> acquire (the others' slabs are).
  __shared__ int s_last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(&counters[tile], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1)) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) counters[tile] = 0u;                       // (ready for the next launch)
  const double* __restrict__ src = out + (int64_t)tile * (TILE * TILE);
  double* __restrict__ dst = packed + (int64_t)tile * (TILE * TILE);
  const int64_t pitch = (int64_t)ntiles_out * (TILE * TILE);
#pragma unroll 4
  for (int e = (int)threadIdx.x * 2; e < TILE * TILE; e += 512) {
    double2_t v = __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(src + e));
    for (int k = 1; k < nsplit; ++k) {
      const double2_t w = __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(src + (int64_t)k * pitch + e));
      v.x += w.x;
      v.y += w.y;
    }
    *reinterpret_cast<double2_t*>(dst + e) = v;
  }
}

// BASELINE config 5, fast variant ("f32 blocks"): fp32 Jacobian, products summed in fp32 by v_mfma_f32_16x16x4_f32 over blocks of
// <= 128 samples, block sums accumulated in fp64.  The fp32 buffer is viewed as an array of sample PAIRS (8 bytes), which makes
// the DMA staging, the swizzled LDS image and the fragment reads those of the fp64 kernel; N2, chunk2 and ldj2 are in pairs.
__global__ __launch_bounds__(256, 2) void gram_f32_kernel(const double* __restrict__ JT2, int64_t ldj2, int64_t N2, int ntiles, int nsplit,
                                                          int64_t chunk2, int total_units, int units_per_xcd, int order,
                                                          const int* __restrict__ perm, double* __restrict__ out) {
  extern __shared__ double lds[];
  int split, tile;
  if (!gram_unit(ntiles, nsplit, total_units, units_per_xcd, order, perm, split, tile)) return;
  int ti, tj;
  tri_decode(tile, ti, tj);
  const int64_t k0 = (int64_t)split * chunk2;
  const int64_t k1 = (k0 + chunk2 < N2) ? (k0 + chunk2) : N2;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop_dma<16, 1>(acc, JT2, ldj2, ti * TILE, JT2, ldj2, tj * TILE, ti == tj, k0, k1, lds);
  tile_store_f32map(acc, out + ((int64_t)split * ntiles + tile) * (TILE * TILE));
}

// BASELINE config 5 proper on the DMA-staged loop (round 5): fp32 Jacobian storage, every product and sum in fp64.  The fp32 buffer is
// viewed as sample pairs like gram_f32_kernel's, both floats of a slot are widened at the fragment read and feed two fp64 MFMAs.  The exact
// Gram of the stored (rounded) Jacobian like the register-staged gram_kernel<.., float>, with another order of the fp64 sums.
__global__ __launch_bounds__(256, 2) void gram_w32_kernel(const double* __restrict__ JT2, int64_t ldj2, int64_t N2, int ntiles, int nsplit,
                                                          int64_t chunk2, int total_units, int units_per_xcd, int order,
                                                          const int* __restrict__ perm, double* __restrict__ out) {
  extern __shared__ double lds[];
  int split, tile;
  if (!gram_unit(ntiles, nsplit, total_units, units_per_xcd, order, perm, split, tile)) return;
  int ti, tj;
  tri_decode(tile, ti, tj);
  const int64_t k0 = (int64_t)split * chunk2;
  const int64_t k1 = (k0 + chunk2 < N2) ? (k0 + chunk2) : N2;
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mainloop_dma<16, 2>(acc, JT2, ldj2, ti * TILE, JT2, ldj2, tj * TILE, ti == tj, k0, k1, lds);
  tile_store(acc, out + ((int64_t)split * ntiles + tile) * (TILE * TILE));
}

// packed[tile][e] = sum_split slabs[split][tile][e]   (fixed order -> deterministic)
// (count doubles of every slab, slabs `pitch` doubles apart: the whole triangle, or the tiles of a few tile rows)
__global__ void reduce_splits_kernel(const double* __restrict__ slabs, int nsplit, int64_t count, int64_t pitch, double* __restrict__ packed) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= count) return;
  double2_t s = *reinterpret_cast<const double2_t*>(slabs + i);
  for (int k = 1; k < nsplit; ++k) {
    const double2_t v = *reinterpret_cast<const double2_t*>(slabs + (int64_t)k * pitch + i);
    s.x += v.x;
    s.y += v.y;
  }
  *reinterpret_cast<double2_t*>(packed + i) = s;
}

// tdvp.py:37-39,150: sums = { sum E, sum E^2, sum |E|, max E }
__global__ __launch_bounds__(1024) void eloc_stats_kernel(const double* __restrict__ e, int64_t N, double* __restrict__ sums) {
  __shared__ double sh[4][16];
  double s = 0, s2 = 0, sa = 0, mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    const double v = e[i];
    s += v; s2 += v * v; sa += fabs(v); mx = fmax(mx, v);
  }
  s = wave_sum(s); s2 = wave_sum(s2); sa = wave_sum(sa);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = s2; sh[2][w] = sa; sh[3][w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, cc = 0, m = -INFINITY;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sh[0][k]; b += sh[1][k]; cc += sh[2][k]; m = fmax(m, sh[3][k]); }
    sums[0] = a; sums[1] = b; sums[2] = cc; sums[3] = m;
  }
}

// one workgroup per parameter row: rowsum[p] = sum_n (JT[p][n]-c[p]), rowdot[p] = sum_n (JT[p][n]-c[p]) (E_n - cE)
template <typename TIn>
__global__ __launch_bounds__(256) void row_moments_kernel(const TIn* __restrict__ JT, int64_t ldj, int64_t N,
                                                         const double* __restrict__ e, const double* __restrict__ c,
                                                         const double* __restrict__ cE_dev, double* __restrict__ rowsum,
                                                         double* __restrict__ rowdot, double* __restrict__ rowsq) {
  __shared__ double sh[3][4];
  const int p = blockIdx.x;
  const double cp = c ? c[p] : 0.0;
  const double cE = cE_dev ? cE_dev[0] : 0.0;
  const TIn* __restrict__ row = JT + (int64_t)p * ldj;
  double s = 0, sd = 0, sq = 0;                       // (sq: sum of squares, for the matrix-free step's diag S -- rowsq may be NULL)
  const int64_t N2 = N & ~(int64_t)1;
  for (int64_t i = (int64_t)threadIdx.x * 2; i < N2; i += 512) {
    const typename PairOf<TIn>::type v = *reinterpret_cast<const typename PairOf<TIn>::type*>(row + i);
    const double2_t ev = *reinterpret_cast<const double2_t*>(e + i);
    const double a = (double)v.x - cp, b = (double)v.y - cp;
    s += a + b;
    sd += a * (ev.x - cE) + b * (ev.y - cE);
    sq = fma(a, a, fma(b, b, sq));
  }
  if (threadIdx.x == 0 && N2 < N) {
    const double a = (double)row[N2] - cp;
    s += a;
    sd += a * (e[N2] - cE);
    sq = fma(a, a, sq);
  }
  s = wave_sum(s); sd = wave_sum(sd); sq = wave_sum(sq);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = sd; sh[2][w] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    rowsum[p] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    rowdot[p] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    if (rowsq) rowsq[p] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
  }
}

// S[i][j] = packed/Ntot - delta_i delta_j ; optional second copy A = S + diag(shift * diag S) for the in-place Cholesky.
// grid: (4, 4, ntiles) 32x32 sub-tiles; the mirrored block goes through an LDS transpose so both writes are coalesced.
__global__ __launch_bounds__(256) void finalize_S_kernel(const double* __restrict__ packed, const double* __restrict__ rowsum,
                                                        const double* __restrict__ c, const double* __restrict__ cg, int P,
                                                        double inv_n, double shift, const double* __restrict__ floor_dev,
                                                        double* __restrict__ S, int64_t lds_, double* __restrict__ A2, int64_t lda2) {
  __shared__ double tr[32][33];
  const double floor_abs = floor_dev ? floor_dev[0] : 0.0;
  int ti, tj;
  tri_decode(blockIdx.z, ti, tj);
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const double* __restrict__ tile = packed + (int64_t)blockIdx.z * (TILE * TILE);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, cidx = c0 + tx;
    const int gi = ti * TILE + r, gj = tj * TILE + cidx;
    double v = 0.0;
    if (gi < P && gj < P) {
      // delta = (global mean) - (shift the Gram kernel used) = rowsum/N + c - c_gram
      const double di = rowsum[gi] * inv_n + (c ? c[gi] : 0.0) - (cg ? cg[gi] : 0.0);
      const double dj = rowsum[gj] * inv_n + (c ? c[gj] : 0.0) - (cg ? cg[gj] : 0.0);
      // diagonal tiles define only their lower triangle (the Gram kernel skips the blocks above it): mirror
      const double raw = (ti == tj && cidx > r) ? tile[cidx * TILE + r] : tile[r * TILE + cidx];
      v = raw * inv_n - di * dj;
      S[(int64_t)gi * lds_ + gj] = v;
      if (A2) A2[(int64_t)gi * lda2 + gj] = (gi == gj) ? v + shift * v + floor_abs : v;
    }
    tr[ty + 8 * q][tx] = v;
  }
  if (ti == tj) return;   // the diagonal tile was written with both triangles above
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cidx = c0 + ty + 8 * q, r = r0 + tx;   // transposed: write S[gj][gi] with gi fastest
    const int gi = ti * TILE + r, gj = tj * TILE + cidx;
    if (gi < P && gj < P) {
      const double v = tr[tx][ty + 8 * q];
      S[(int64_t)gj * lds_ + gi] = v;
      if (A2) A2[(int64_t)gj * lda2 + gi] = v;
    }
  }
}

// F = rowdot/N - delta*deltaE, mean = c + delta (tdvp.py:41,45)
__global__ void finalize_F_kernel(const double* __restrict__ rowsum, const double* __restrict__ rowdot,
                                  const double* __restrict__ sums, const double* __restrict__ c, const double* __restrict__ cE_dev,
                                  int P, double inv_n, double* __restrict__ F, double* __restrict__ mean) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const double cE = cE_dev ? cE_dev[0] : 0.0;
  const double deltaE = sums[0] * inv_n - cE;
  const double delta = rowsum[p] * inv_n;
  if (F) F[p] = rowdot[p] * inv_n - delta * deltaE;
  if (mean) mean[p] = (c ? c[p] : 0.0) + delta;
}

// Diagnostic: back-to-back v_mfma_f64_16x16x4_f64 on 8 independent accumulators per wave (operands in registers, no memory
// traffic; iters is rounded up to a multiple of 16) -- the measured matrix-pipe peak the Gram kernel's utilisation is quoted against (bench.py).
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(int iters, double* __restrict__ out) {
  double4_t acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const double a = 1e-3 * (1 + (threadIdx.x & 63)), b = 1e-3 * (16 - (threadIdx.x & 15));
  for (int it = 0; it < iters; it += 16) {   // 128 MFMAs per trip so loop overhead and register shuffles vanish
#pragma unroll
    for (int rep = 0; rep < 16; ++rep)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

extern "C" int vmcpde_bench_mfma_f64(int32_t iters, int32_t blocks, double* out, void* stream) {
  VMC_REQUIRE(out && iters >= 1 && blocks >= 1, "vmcpde_bench_mfma_f64: bad argument");
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef VMC_GRAM_STAMPS
extern "C" int vmcpde_debug_read_stamps(unsigned long long* out_host, int n) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vmcpde::g_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int64_t vmcpde_gram_packed_doubles(int32_t P) {
  if (P <= 0) return 0;
  const int nb = (P + TILE - 1) / TILE;
  return tri_count(nb) * TILE * TILE;
}

constexpr int SLOTS_STAGED = 512, SLOTS_DMA = (GRAM_BKD == 8) ? 768 : 512;

extern "C" int64_t vmcpde_gram_workspace_bytes(int32_t P, int64_t n_samples) {
  if (P <= 0 || n_samples <= 0) return 0;
  // any variant may run (register-staged, DMA, fp32 blocks over sample pairs)
  const GramPlan a = make_plan(P, n_samples, SLOTS_STAGED), b = make_plan(P, n_samples, SLOTS_DMA), c = make_plan(P, (n_samples + 1) / 2, SLOTS_STAGED);
  int ns = a.nsplit > b.nsplit ? a.nsplit : b.nsplit;
  if (c.nsplit > ns) ns = c.nsplit;
  // (+ one counter per tile behind the slabs: the fused split reduction of the DMA kernel)
  return (int64_t)ns * a.ntiles * TILE * TILE * (int64_t)sizeof(double) + (((int64_t)a.ntiles * 4 + 255) / 256) * 256;
}

extern "C" int vmcpde_eloc_stats(const double* eloc, int64_t n_samples, double* sums, void* stream) {
  VMC_REQUIRE(eloc && sums, "vmcpde_eloc_stats: null argument");
  hipLaunchKernelGGL(eloc_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, eloc, n_samples, sums);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_row_moments(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, const double* eloc,
                                  const double* c, const double* cE_dev, double* rowsum, double* rowdot, void* stream) {
  VMC_REQUIRE(JT && eloc && rowsum && rowdot, "vmcpde_row_moments: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1 && ldj >= n_samples && (ldj % 2) == 0, "bad shape (ldj must be even and >= N)");
  if (jt_f32)
    hipLaunchKernelGGL(row_moments_kernel<float>, dim3(P), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(JT), ldj, n_samples,
                       eloc, c, cE_dev, rowsum, rowdot, (double*)nullptr);
  else
    hipLaunchKernelGGL(row_moments_kernel<double>, dim3(P), dim3(256), 0, (hipStream_t)stream, static_cast<const double*>(JT), ldj, n_samples,
                       eloc, c, cE_dev, rowsum, rowdot, (double*)nullptr);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

// the same pass with the row sums of squares as a third output (diag S of the matrix-free step): rowsq[p] = sum_n (JT[p][n] - c[p])^2
extern "C" int vmcpde_row_moments_sq(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const double* eloc, const double* c,
                                     const double* cE_dev, double* rowsum, double* rowdot, double* rowsq, void* stream) {
  VMC_REQUIRE(JT && eloc && rowsum && rowdot && rowsq, "vmcpde_row_moments_sq: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1 && ldj >= n_samples && (ldj % 2) == 0, "bad shape (ldj must be even and >= N)");
  hipLaunchKernelGGL(row_moments_kernel<double>, dim3(P), dim3(256), 0, (hipStream_t)stream, JT, ldj, n_samples, eloc, c, cE_dev, rowsum, rowdot,
                     rowsq);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_gram(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, const double* c, const double* w,
                           double* packed, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(JT && packed, "vmcpde_gram: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1, "bad shape");
  VMC_REQUIRE(ldj >= n_samples && (ldj % 2) == 0 && ldj < (1 << 22), "ldj must be even, >= n_samples and < 2^22");
  VMC_REQUIRE((reinterpret_cast<uintptr_t>(JT) & 15) == 0, "JT must be 16-byte aligned");
  const bool dma = !jt_f32 && !c && !w;       // the steady-state variant
  const bool f32blocks = jt_f32 == 2 && !c && !w;   // fp32 MFMA block sums (centred / weighted requests fall back to widening)
  VMC_REQUIRE(!f32blocks || (ldj % 4) == 0, "jt_f32 = 2 needs ldj to be a multiple of 4 (16-byte rows)");
  // fp32 storage, fp64 products and sums, uncentred and unweighted: the DMA-staged loop over sample pairs (VMCPDE_GRAM_F32_DMA=0: the
  // register-staged loop, which orders its sums like the fp64 kernel and is bit-identical to it on fp32-valued data)
  // (an odd sample count would let the pad float behind the last sample into the sums: vmcpde.h promises zero padding only for jt_f32 = 2,
  // so those calls keep the register-staged loop, which reads exactly n_samples entries per row)
  const bool w32 = jt_f32 == 1 && !c && !w && (ldj % 4) == 0 && (n_samples % 2) == 0 && (reinterpret_cast<uintptr_t>(JT) & 15) == 0 &&
                   env_int("VMCPDE_GRAM_F32_DMA", 1) != 0;
  // in pairs of samples for the fp32-block variant (an odd N relies on the zero padding of the rows)
  GramPlan g = (f32blocks || w32) ? make_plan(P, (n_samples + 1) / 2, SLOTS_STAGED) : make_plan(P, n_samples, dma ? SLOTS_DMA : SLOTS_STAGED);
  const int64_t need = (int64_t)g.nsplit * g.ntiles * TILE * TILE * (int64_t)sizeof(double);
  double* out = packed;
  if (g.nsplit > 1) {
    VMC_REQUIRE(workspace && workspace_bytes >= need, "gram workspace too small: need %lld bytes", (long long)need);
    out = static_cast<double*>(workspace);
  }
  hipStream_t st = (hipStream_t)stream;
  const int order = gram_order();   // tuning knob
  const int total = g.ntiles * g.nsplit;
  int per_xcd = (order == 1) ? (g.ntiles + 7) / 8 : (total + 7) / 8;
  int grid = (order == 1) ? per_xcd * 8 * g.nsplit : per_xcd * 8;
  if (order == 2) {        // slots of the busiest XCD (gram_unit's balanced deal)
    per_xcd = order2_slots_per_xcd(g.ntiles, g.nsplit);
    grid = per_xcd * 8;
  }
  // tuning knob: extra dynamic LDS forces one workgroup per CU (to measure what the co-resident partner is worth)
  const size_t lds_pad = (size_t)env_int("VMCPDE_GRAM_LDSPAD", 0);
  const size_t lds_bytes = TILE_LDS_BYTES + lds_pad;
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  DeviceCtxLock lock;      // first-use attribute changes; held over the (asynchronous) launches below, which is harmless
  const int* perm = nullptr;
  if (order == 2) {
    if (int rc = ensure_gram_perm(ctx, g.nb, &perm)) return rc;
  }
  if (!ctx->attr_gram || lds_pad) {
#define VMC_ATTR(CEN, SCA, T) VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_kernel<CEN, SCA, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes))
    VMC_ATTR(false, false, double); VMC_ATTR(true, false, double); VMC_ATTR(false, true, double); VMC_ATTR(true, true, double);
    VMC_ATTR(false, false, float); VMC_ATTR(true, false, float); VMC_ATTR(false, true, float); VMC_ATTR(true, true, float);
#undef VMC_ATTR
    ctx->attr_gram = true;
  }
#define VMC_LAUNCH_GRAM(CEN, SCA, T)                                                                                              \
  hipLaunchKernelGGL((gram_kernel<CEN, SCA, T>), dim3(grid), dim3(256), lds_bytes, st, static_cast<const T*>(JT), ldj, P, n_samples, c, w, \
                     g.ntiles, g.nsplit, g.chunk, total, per_xcd, order, perm, out)
#define VMC_LAUNCH_GRAM_T(T)                          \
  do {                                                \
    if (c && w) VMC_LAUNCH_GRAM(true, true, T);       \
    else if (c) VMC_LAUNCH_GRAM(true, false, T);      \
    else if (w) VMC_LAUNCH_GRAM(false, true, T);      \
    else VMC_LAUNCH_GRAM(false, false, T);            \
  } while (0)
  if (f32blocks) {
    if (!ctx->attr_gram_f32) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_DMA_LDS_BYTES));
      ctx->attr_gram_f32 = true;
    }
    hipLaunchKernelGGL(gram_f32_kernel, dim3(grid), dim3(256), TILE_DMA_LDS_BYTES, st, static_cast<const double*>(JT), ldj / 2,
                       (n_samples + 1) / 2, g.ntiles, g.nsplit, g.chunk, total, per_xcd, order, perm, out);
  } else if (w32) {
    if (!ctx->attr_gram_w32) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_w32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_DMA_LDS_BYTES));
      ctx->attr_gram_w32 = true;
    }
    hipLaunchKernelGGL(gram_w32_kernel, dim3(grid), dim3(256), TILE_DMA_LDS_BYTES, st, static_cast<const double*>(JT), ldj / 2,
                       (n_samples + 1) / 2, g.ntiles, g.nsplit, g.chunk, total, per_xcd, order, perm, out);
  } else if (dma) {
    const size_t dma_lds = 4 * DmaGeom<GRAM_BKD>::DT * sizeof(double) + lds_pad;
    if ((!ctx->attr_gram_dma || lds_pad) && dma_lds > 64 * 1024) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dma_lds));
      ctx->attr_gram_dma = true;
    }
    // fused split reduction: the counters live behind the slabs (zero before the first launch on this workspace: cleared here, left at zero by the kernel)
    const bool fused_reduce = g.nsplit > 1 && env_int("VMCPDE_GRAM_FUSED_REDUCE", GRAM_FUSED_REDUCE_DEFAULT) != 0 &&
                              workspace_bytes >= need + (int64_t)g.ntiles * 4;
    unsigned int* counters = fused_reduce ? reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + need) : nullptr;
    if (fused_reduce) VMC_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)g.ntiles * 4, st));
    hipLaunchKernelGGL(gram_dma_kernel, dim3(grid), dim3(256), dma_lds, st, static_cast<const double*>(JT), ldj, n_samples, g.ntiles,
                       g.nsplit, g.chunk, total, per_xcd, order, perm, out, g.ntiles, (int)P, counters, packed);
    if (fused_reduce) {
      VMC_CHECK_HIP(hipGetLastError());
      return 0;
    }
  } else if (jt_f32) VMC_LAUNCH_GRAM_T(float);
  else VMC_LAUNCH_GRAM_T(double);
#undef VMC_LAUNCH_GRAM_T
#undef VMC_LAUNCH_GRAM
  VMC_CHECK_HIP(hipGetLastError());
  if (g.nsplit > 1) {
    const int64_t per_split = (int64_t)g.ntiles * TILE * TILE;
    const int64_t threads = per_split / 2;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, out, g.nsplit, per_split, per_split,
                       packed);
    VMC_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

// The steady-state Gram (fp64, uncentred, unweighted) for the tiles of tile rows [row_begin, row_end) only -- the packed buffer is tile-row
// major, so these are ONE contiguous piece of it, complete (split sums included) when this call's launches have drained.  Same sample
// splits as vmcpde_gram on the whole triangle, same per-unit arithmetic: calling it for a partition of the tile rows gives the packed
// buffer of one vmcpde_gram call bit for bit.  What it is for: a multi-GPU step can hand finished row groups to the all-reduce while
// the next group is still being formed (mpi_wrapper.py:129-163 reduces after the whole sum).
extern "C" int vmcpde_gram_rows(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, int32_t row_begin, int32_t row_end, double* packed,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(JT && packed, "vmcpde_gram_rows: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1, "bad shape");
  VMC_REQUIRE(ldj >= n_samples && (ldj % 2) == 0 && ldj < (1 << 22), "ldj must be even, >= n_samples and < 2^22");
  VMC_REQUIRE((reinterpret_cast<uintptr_t>(JT) & 15) == 0, "JT must be 16-byte aligned");
  const GramPlan g = make_plan(P, n_samples, SLOTS_DMA);             // the plan of the whole triangle: same splits, same chunks
  VMC_REQUIRE(0 <= row_begin && row_begin < row_end && row_end <= g.nb, "tile rows [%d, %d) outside [0, %d)", row_begin, row_end, g.nb);
  const int64_t need = (int64_t)g.nsplit * g.ntiles * TILE * TILE * (int64_t)sizeof(double);
  double* out = packed;
  if (g.nsplit > 1) {
    VMC_REQUIRE(workspace && workspace_bytes >= need, "gram workspace too small: need %lld bytes", (long long)need);
    out = static_cast<double*>(workspace);
  }
  hipStream_t st = (hipStream_t)stream;
  const int t0 = (int)tri_count(row_begin), t1 = (int)tri_count(row_end), nsub = t1 - t0;
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  DeviceCtxLock lock;
  // the group's tiles along a Morton curve (global tile indices), balanced over the XCDs like the whole triangle's (gram_order.hpp)
  VMC_REQUIRE(g.nb < (1 << 20), "vmcpde_gram_rows: %d tile rows", g.nb);
  const int64_t group_bit = (int64_t)1 << 62;                         // (whole-triangle orders use the plain row count as their key)
  const int64_t key = group_bit | ((int64_t)g.nb << 40) | ((int64_t)row_begin << 20) | (int64_t)row_end;
  auto it = ctx->gram_perms.find(key);
  if (it == ctx->gram_perms.end()) {
    // first use of this row group on this GPU: a blocking upload, once.  A caller that keeps changing its cuts must not grow the
    // process for ever: beyond 64 cached groups all of them are dropped (hipFree waits for the kernels that may still read them)
    int n_groups = 0;
    for (auto& kv : ctx->gram_perms) n_groups += (kv.first & group_bit) != 0;
    if (n_groups >= 64) {
      for (auto jt = ctx->gram_perms.begin(); jt != ctx->gram_perms.end();) {
        if (jt->first & group_bit) { (void)hipFree(jt->second); jt = ctx->gram_perms.erase(jt); } else ++jt;
      }
    }
    std::vector<int> host;
    for (int t : morton_tile_order(g.nb))
      if (t >= t0 && t < t1) host.push_back(t);
    int* dev = nullptr;
    VMC_CHECK_HIP(hipMalloc(&dev, host.size() * sizeof(int)));
    VMC_CHECK_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
    it = ctx->gram_perms.emplace(key, dev).first;
  }
  const int per_xcd = order2_slots_per_xcd(nsub, g.nsplit);
  const size_t dma_lds = 4 * DmaGeom<GRAM_BKD>::DT * sizeof(double);
  if (!ctx->attr_gram_dma && dma_lds > 64 * 1024) {
    VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dma_lds));
    ctx->attr_gram_dma = true;
  }
  hipLaunchKernelGGL(gram_dma_kernel, dim3(per_xcd * 8), dim3(256), dma_lds, st, JT, ldj, n_samples, nsub, g.nsplit, g.chunk, nsub * g.nsplit,
                     per_xcd, 2, it->second, out, g.ntiles, (int)P, (unsigned int*)nullptr, (double*)nullptr);
  VMC_CHECK_HIP(hipGetLastError());
  if (g.nsplit > 1) {
    const int64_t count = (int64_t)nsub * TILE * TILE, pitch = (int64_t)g.ntiles * TILE * TILE, off = (int64_t)t0 * TILE * TILE;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((count / 2 + 255) / 256)), dim3(256), 0, st, out + off, g.nsplit, count, pitch,
                       packed + off);
    VMC_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int vmcpde_gram_finalize(const double* packed, const double* rowsum, const double* rowdot, const double* sums,
                                    const double* c, const double* c_gram, const double* cE_dev, int32_t P, double n_total, double diagonal_shift,
                                    const double* floor_dev, double* S, int64_t lds, double* A_shifted, int64_t lda, double* F, double* mean,
                                    void* stream) {
  VMC_REQUIRE(packed && rowsum && S, "vmcpde_gram_finalize: null argument");
  VMC_REQUIRE(P >= 1 && lds >= P && n_total > 0, "bad shape");
  VMC_REQUIRE(!A_shifted || lda >= P, "bad lda");
  VMC_REQUIRE((!F && !mean) || (rowdot && sums), "F/mean need rowdot and sums");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (P + TILE - 1) / TILE;
  const double inv_n = 1.0 / n_total;
  hipLaunchKernelGGL(finalize_S_kernel, dim3(4, 4, (unsigned)tri_count(nb)), dim3(256), 0, st, packed, rowsum, c, c_gram, P, inv_n,
                     diagonal_shift, floor_dev, S, lds, A_shifted, lda);
  VMC_CHECK_HIP(hipGetLastError());
  if (F || mean) {
    hipLaunchKernelGGL(finalize_F_kernel, dim3((P + 255) / 256), dim3(256), 0, st, rowsum, rowdot, sums, c, cE_dev, P, inv_n, F, mean);
    VMC_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
