// EXPERIMENT (opt-in, never the default): the Gram matrix on the INT8 matrix pipe by error-free slicing of the fp64 Jacobian.
//
// Reference: tdvp.py:46-47 / mpi_wrapper.py:21-25 -- the same packed sums as vmcpde_gram (optionally centred / weighted).
// MI355X runs v_mfma_i32_32x32x32_i8 at 64x the rate of v_mfma_f64_16x16x4_f64; an fp64 product can be recovered from EXACT integer
// products of digits (Ozaki scheme, integer form).  Default form (NS = 6 digits of BITS = 8):
//     J[p][n] = 2^e_p * sum_{s<6} d_s[p][n] 2^(-7-8s),   d_s balanced base-256 digits in [-128, 127]  (per-row power-of-two scale with one
//                                                        bit of head room; 47 bits, split from the least significant end)
//     S[p][q] = 2^(e_p+e_q-14) sum_{k<6} 2^(-8k) T_k[p][q],   T_k = sum_{s+t=k} D_s D_t^T   (int32 accumulators: exact)
// Every digit product is <= 2^14 and diagonal k holds k+1 <= 6 pairs, so one int32 accumulator per diagonal is exact over 16 384
// samples -- the size limit of a sample split.  What is dropped: the pair products with s + t >= 6 (<= 2^-62 of max_p max_q per
// sample, random signs: they average out over the samples) EXCEPT the squares (s, s) on the diagonal p = q, which the slicing pass sums
// exactly and diag_fix_kernel adds back.  21 MFMAs replace 64 fp64 MFMA-equivalents per 32 x 32 x 32 block: 0.33 of the fp64
// matrix-pipe time at equal utilisation.  Also built, for the record: six 7-bit digits rounded to
// nearest (BITS = 7: |d| <= 64, 41 bits -- 1.9e-12 of |S|_F on Gaussian data) and seven of them (28 pairs, 1.6e-14, but seven
// diagonals of a 64 x 32 wave tile do not fit 256 registers: the half-tile form at one wave per SIMD is 2.5x slower).
//
// Pipeline (vmcpde_gram_i8):
//   1. row_expo_kernel      e_p from the row maxima of (J - c) w (one pass over J)
//   2. slice_kernel         J -> int8 planes in the BLOCKED layout planes[kb][s][row block of 128][k half][row][16 B]: the 32 samples of
//                           K-block kb of slice s of 128 consecutive rows are 4 KB contiguous -- exactly the LDS image of an operand, so
//                           staging is plain DMA; the two 16-byte halves of a row sit 2 KB apart, so that the 16-lane groups of a
//                           ds_read_b128 fragment read (lanes {0-3,12-15,20-27}, ...) touch 16 distinct 16-byte bank groups -- with
//                           rows of 32 contiguous bytes every fragment read was a 2-way bank conflict
//   3. gram_i8_kernel       lower-triangle 128 x 128 tiles x sample splits (the fp64 kernel's XCD-aware Z-order deal); 8 waves, wave tile
//                           64 x 32, NS int32 diagonals (2 x 16 x NS accumulator VGPRs), operands global -> LDS by global_load_lds
//                           (three 48 KB K-block buffers), fragments by ds_read_b128; epilogue: Horner sum of the diagonals in fp64,
//                           scale by 2^(e_p+e_q-14), fp64 slab in the layout of gram.hip
//   4. reduce_splits_i8_kernel, diag_fix_kernel (the dropped diagonal squares; the heavy-tail figure max_p row max / row rms)
#include <algorithm>
#include <vector>

#include "mfma_tile.hpp"
#include "gram_order.hpp"

using namespace vmcpde;

namespace {

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16v_t __attribute__((ext_vector_type(16)));

constexpr int KB = 32;                 // samples per K-block = bytes per fragment row

// e_p: |J[p][n]| / 2^e_p < 1 for every n (0 for an all-zero row)
// (optional centring shift c[p] and per-sample weight w[n]: the sliced operand is (J[p][n] - c[p]) w[n], as in gram.hip's staged variants)
template <typename TIn>
__global__ __launch_bounds__(256) void row_expo_kernel(const TIn* __restrict__ JT, int64_t ldj, int64_t N, int P, int headroom,
                                                       const double* __restrict__ c, const double* __restrict__ w, int* __restrict__ expo) {
  __shared__ double red[4];
  const int t = threadIdx.x, p = blockIdx.x;
  double mx = 0.0;
  if (p < P) {
    const TIn* __restrict__ row = JT + (int64_t)p * ldj;
    const double cp = c ? c[p] : 0.0;
    const int64_t N2 = N & ~(int64_t)1;
    for (int64_t n = 2 * t; n < N2; n += 512) {
      const typename PairOf<TIn>::type v = *reinterpret_cast<const typename PairOf<TIn>::type*>(row + n);
      const double a = ((double)v.x - cp) * (w ? w[n] : 1.0), b = ((double)v.y - cp) * (w ? w[n + 1] : 1.0);
      mx = fmax(mx, fmax(fabs(a), fabs(b)));
    }
    if (t == 0 && N2 < N) mx = fmax(mx, fabs(((double)row[N2] - cp) * (w ? w[N2] : 1.0)));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  if (t == 0) {
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    int e = 0;
    if (mx > 0.0 && isfinite(mx)) frexp(mx, &e);       // mx = f 2^e, f in [0.5, 1)
    expo[p] = e + headroom;
  }
}

// planes[((kb * NS + s) * Ppad + p) * 32 + j] = digit s of sample 32 kb + j of row p.  One workgroup: 32 rows x KB_PER_WG K-blocks;
// a thread converts 8 consecutive samples, so the four threads of a row cover one K-block and the 128 threads of a K-block write
// 1 KB contiguous per slice.
//   BITS = 7: x = J / 2^e in (-1, 1); d_0 = rint(64 x), then residuals x 128 rounded to nearest: |d_s| <= 64, x = sum d_s 2^(-6-7s).
//   BITS = 8: x = J / 2^e with one more bit of head room (|x| < 1/2); M = rint(x 2^(8 NS - 1)) is split from the least significant end
//             into BALANCED base-256 digits d = ((M + 128) mod 256) - 128 in [-128, 127] (the carry goes up; the top digit is
//             <= 65 in magnitude): x = sum d_s 2^(-7-8s) exactly to 2^(-8 NS + 1) -- 47 bits from six slices, no offsets, digits of
//             either sign, so dropped pair products still average out.
// It also returns, per row, the two dropped diagonals that do NOT average out for p = q: dsum[3p] = sum_n sum_{s+t=NS} d_s d_t and
// dsum[3p+1] = sum_n sum_{s+t=NS+2} d_s d_t (exact integers, integer atomics: deterministic), and dsum[3p+2] = sum_n (d_0 2^BITS + d_1)^2 for the heavy-tail figure of diag_fix_kernel.  For p != q the dropped pair products
// have random signs and sum to ~sqrt(N); on the diagonal the (s,s) terms are squares -- a bias of 5e-12 |S|_F at NS = 6, BITS = 7 if left out.
constexpr int SL_ROWS = 32, SL_KB_PER_WG = 64;
template <int NS, int BITS, typename TIn>
__global__ __launch_bounds__(256) void slice_kernel(const TIn* __restrict__ JT, int64_t ldj, int64_t N, int P, int Ppad, const int* __restrict__ expo,
                                                    const double* __restrict__ c, const double* __restrict__ w, int8_t* __restrict__ planes,
                                                    int64_t nkb, unsigned long long* __restrict__ dsum) {
  const int t = threadIdx.x;
  const int kb_in = t >> 7;                       // two K-blocks per trip
  const int row = (t & 127) >> 2, quarter = t & 3;
  const int p = blockIdx.x * SL_ROWS + row;
  const bool rowok = p < P;
  // expo[p] already holds the head-room bit for BITS = 8
  const int shift = rowok ? (BITS == 7 ? 6 : 8 * NS - 1) - expo[p] : 0;      // (ldexp per value: 2^shift itself may not be representable)
  const TIn* __restrict__ src = JT + (int64_t)(rowok ? p : 0) * ldj;
  const double cp = (c && rowok) ? c[p] : 0.0;
  const int64_t kb_lo = (int64_t)blockIdx.y * SL_KB_PER_WG;
  long long t_lo = 0, t_hi = 0, t_top = 0;
  for (int it = 0; it < SL_KB_PER_WG / 2; ++it) {
    const int64_t kb = kb_lo + 2 * it + kb_in;
    if (kb >= nkb) break;
    const int64_t n0 = kb * KB + quarter * 8;
    double x[8];
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      double2_t v = {0.0, 0.0};
      if (rowok && n0 + q + 1 < N) {
        const typename PairOf<TIn>::type u2 = *reinterpret_cast<const typename PairOf<TIn>::type*>(src + n0 + q);
        v.x = ((double)u2.x - cp) * (w ? w[n0 + q] : 1.0);
        v.y = ((double)u2.y - cp) * (w ? w[n0 + q + 1] : 1.0);
      } else if (rowok && n0 + q < N) {
        v.x = ((double)src[n0 + q] - cp) * (w ? w[n0 + q] : 1.0);
      }
      x[q] = ldexp(v.x, shift);
      x[q + 1] = ldexp(v.y, shift);
    }
    // (kb, s = 0, row block p / 128) + k half (quarter / 2) * 2 KB + row-in-block * 16 + (quarter % 2) * 8
    int8_t* __restrict__ dst = planes + (((int64_t)kb * NS) * Ppad + (p & ~127)) * KB + (quarter >> 1) * 2048 + (p & 127) * 16 + (quarter & 1) * 8;
    int dig[NS][8];
    if (BITS == 7) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double d = rint(x[q]);
          dig[s][q] = (int)d;
          x[q] = (x[q] - d) * 128.0;
        }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        long long M = (long long)rint(x[q]);
#pragma unroll
        for (int s = NS - 1; s >= 1; --s) {
          const int d = (int)((M + 128) & 255) - 128;
          dig[s][q] = d;
          M = (M - d) >> 8;
        }
        dig[0][q] = (int)M;
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      union { int8_t b[8]; int2 w; } u;
#pragma unroll
      for (int q = 0; q < 8; ++q) u.b[q] = (int8_t)dig[s][q];
      *reinterpret_cast<int2*>(dst + (int64_t)s * Ppad * KB) = u.w;
    }
    int lo = 0, hi = 0;
#pragma unroll
    for (int s = 1; s < NS; ++s) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        lo += dig[s][q] * dig[NS - s][q];
        if (s >= 3 && NS + 2 - s < NS) hi += dig[s][q] * dig[NS + 2 - s][q];
      }
    }
    long long top = 0;      // (x to two digits)^2: resolves a row rms down to 2^-15 of the row's scale
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long long v2 = (long long)dig[0][q] * (1 << BITS) + dig[1][q];
      top += v2 * v2;
    }
    t_lo += lo;
    t_hi += hi;
    t_top += top;
  }
  // the four quarters of a row sit in adjacent lanes
  t_lo += __shfl_xor(t_lo, 1); t_lo += __shfl_xor(t_lo, 2);
  t_hi += __shfl_xor(t_hi, 1); t_hi += __shfl_xor(t_hi, 2);
  t_top += __shfl_xor(t_top, 1); t_top += __shfl_xor(t_top, 2);
  if (quarter == 0 && rowok) {
    atomicAdd(dsum + 3 * p, (unsigned long long)t_lo);
    atomicAdd(dsum + 3 * p + 1, (unsigned long long)t_hi);
    atomicAdd(dsum + 3 * p + 2, (unsigned long long)t_top);      // sum of squares to two digits: the row's rms in units of its scale
  }
}

// S_pp += 2^(2 e_p - 12) (2^(-7 NS) dsum[2p] + 2^(-7 (NS+2)) dsum[2p+1]): the diagonal entries of the diagonal tiles of the packed buffer
// It also reports how heavy-tailed the rows are: stats[0] = max over rows of (row maximum / row rms), estimated from the top digits
// (the per-row scale is what the fixed number of digits is relative to: the error of the sliced sums grows like that ratio squared).
__global__ void diag_fix_kernel(const unsigned long long* __restrict__ dsum, const int* __restrict__ expo, int P, int ns, int bits, int64_t nsamp,
                                double* __restrict__ packed, double* __restrict__ stats) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int ti = p >> 7, r = p & 127;
  if (stats) {
    // two leading digits: x 2^(2 bits - 1) = d_0 2^bits + d_1, so rms of x = sqrt(sum / n) / 2^(2 bits - 1); the row maximum is >= 1/4
    // (8-bit digits, head room) or 1/2 (7-bit)
    const double top = (double)(long long)dsum[3 * p + 2];
    if (top > 0.0) {
      const double rms = ldexp(sqrt(top / (double)nsamp), -(2 * bits - 1));
      const double ratio = (bits == 7 ? 1.0 : 0.5) / rms;                  // an upper bound of max / rms within a factor two
      atomicMax(reinterpret_cast<unsigned long long*>(stats), (unsigned long long)__double_as_longlong(ratio));   // (positive doubles order like integers)
    }
  }
  const double add = ldexp((double)(long long)dsum[3 * p], -bits * ns) + ldexp((double)(long long)dsum[3 * p + 1], -bits * (ns + 2));
  packed[((int64_t)ti * (ti + 1) / 2 + ti) * (TILE * TILE) + r * TILE + r] += ldexp(add, 2 * expo[p] - (bits == 7 ? 12 : 14));
}

// One unit = (sample split, lower-triangle tile[, column half]).  WAVES = 8: 512 threads, the whole 128 x 128 tile, wave w -> rows
// 64 (w >> 2).., columns 32 (w & 3)..; two waves per SIMD, which 6 diagonals allow (234 VGPRs) and 7 do not (they spill).  WAVES = 4:
// 256 threads, a 128 x 64 half tile (columns 64 half + 32 (w & 1)..), one wave per SIMD with up to 512 registers -- the form for 7 slices.
template <int NS, int WAVES, int BITS>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void gram_i8_kernel(const int8_t* __restrict__ planes, const int* __restrict__ expo, int Ppad,
                                                                               int64_t nkb, int ntiles, int nsplit, int64_t kb_per_split,
                                                                               int units_per_xcd, int diag_flags, const int* __restrict__ perm, double* __restrict__ out) {
  extern __shared__ int4_t lds_i8[];                  // 2 buffers x (A: NS x 128 rows x 32 B | B: NS x BROWS rows x 32 B)
  constexpr int BROWS = (WAVES == 8) ? 128 : 64;
  constexpr int OPA = NS * 128 * KB / 16, OPB = NS * BROWS * KB / 16;      // int4 per operand image
  constexpr int BUF = OPA + OPB;
  constexpr int PA = NS * 4, PB = NS * BROWS / 32;    // 1-KiB DMA pieces per operand image
  constexpr int HALVES = (WAVES == 8) ? 1 : 2;
  int split, tile, half = 0;
  if (WAVES == 8) {       // the fp64 kernel's balanced Z-order deal: every XCD owns a contiguous piece of the Morton curve in every split
    if (!gram_unit(ntiles, nsplit, ntiles * nsplit, units_per_xcd, 2, perm, split, tile)) return;
  } else {
    const int unit = (int)blockIdx.x, per_split = ntiles * HALVES;
    split = unit / per_split;
    if (split >= nsplit) return;
    const int within = unit - split * per_split;
    tile = within / HALVES;
    half = within - tile * HALVES;
  }
  int ti, tj;
  tri_decode(tile, ti, tj);
  const bool diag = (ti == tj);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = (WAVES == 8) ? (wave >> 2) : (wave >> 1), wc = (WAVES == 8) ? (wave & 3) : (wave & 1);
  const int col0 = 64 * half + 32 * wc;               // first column of this wave inside the 128 x 128 tile
  const int r32 = lane & 31, h = lane >> 5;
  int64_t kb0 = (int64_t)split * kb_per_split;
  int64_t kb1 = (kb0 + kb_per_split < nkb) ? (kb0 + kb_per_split) : nkb;
  // timing diagnostics (VMCPDE_GRAM_I8_DIAG, wrong results): 1 every split reads the first chunk, 2 no barrier in the loop, 4 no DMA in
  // the loop, 8 no DMA wait in the loop, 16 no raised priority for the DMA pieces, 32 every wave issues its pieces first, 64 all loads from four L2-resident K-blocks, 128 no fragment reads and no MFMAs (DMA only), 256 the DMA pieces of a step in one burst
  if (diag_flags & 1) { kb1 -= kb0; kb0 = 0; }
  const bool active = !diag || (64 * wr + 63 >= col0);   // a wave tile entirely above the diagonal of a diagonal tile forms no products

  int16v_t acc[NS][2];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][m][r] = 0;

  // DMA: piece q of a K-block = (operand, slice, 1-KiB part); wave w issues q = w, w + WAVES, ...  An operand image is
  // [slice][k half][rows][16 B]; the A image (128 rows) is a plain copy of 4 KB per slice, the B image of the half-tile variant takes the
  // two 1-KiB runs of its 64 rows
  const int npieces = diag ? PA : PA + PB;
  auto issue_some = [&](int64_t kb, int buf, int first, int count) {     // this wave's pieces first .. first + count - 1 of K-block kb
    if (diag_flags & 64) kb = (kb & 3);      // (timing diagnostic: every load hits the same four K-blocks -- L2-resident)
    const char* __restrict__ g0 = reinterpret_cast<const char*>(planes) + ((int64_t)kb * NS) * Ppad * KB;
    for (int q = wave + first * WAVES, c = 0; q < npieces && c < count; q += WAVES, ++c) {
      const bool isb = q >= PA;
      const int rem = isb ? q - PA : q;
      const int parts = isb ? BROWS / 32 : 4;
      const int s = rem / parts, part = rem - s * parts;
      const char* src;
      int4_t* dst;
      if (!isb) {
        src = g0 + ((int64_t)s * Ppad + ti * 128) * KB + part * 1024 + lane * 16;
        dst = lds_i8 + buf * BUF + s * 256 + part * 64;
      } else if (BROWS == 128) {
        src = g0 + ((int64_t)s * Ppad + tj * 128) * KB + part * 1024 + lane * 16;
        dst = lds_i8 + buf * BUF + OPA + s * 256 + part * 64;
      } else {       // 64 rows of the tile's block: k half `part`, rows 64 half ..
        src = g0 + ((int64_t)s * Ppad + tj * 128) * KB + part * 2048 + half * 1024 + lane * 16;
        dst = lds_i8 + buf * BUF + OPA + s * 128 + part * 64;
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
  };
  auto issue = [&](int64_t kb, int buf) { issue_some(kb, buf, 0, 1 << 20); };
  // Triple-buffered K-blocks (NBUF = 3 where the LDS allows: 3 x 48 KB at 6 slices): the DMA of K-block kb + 2 is issued before the
  // products of kb, and the wait at the end of a step covers only the pieces of kb + 1 (issued a whole step earlier) -- with two
  // buffers every step ended on a full round trip to L2 / HBM.  vmcnt counts this wave's own pieces; the barrier then tells every
  // wave that all pieces of kb + 1 have landed and that nobody reads buffer kb % NBUF any more.
  constexpr int NBUF = (3 * BUF * 16 <= 160 * 1024) ? 3 : 2;
  int mine = 0;                                   // pieces this wave issues per K-block
  for (int q = wave; q < npieces; q += WAVES) ++mine;
  auto wait_all_but_newest = [&](bool newest_in_flight) {
    if (!newest_in_flight) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    switch (mine) {                               // (s_waitcnt takes an immediate)
      case 0: break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  int cur = 0;
  if (kb0 < kb1) {
    issue(kb0, 0);
    if (NBUF == 3 && kb0 + 1 < kb1) issue(kb0 + 1, 1);
    wait_all_but_newest(NBUF == 3 && kb0 + 1 < kb1);
    __syncthreads();
  }
  // Where a wave issues its DMA pieces.  A `global_load_lds` holds the issuing wave for ~100 cycles; in one burst of six per step
  // (before or behind the products, or the SIMD's two waves at opposite ends: all within 1 %) the wave issues no MFMA for ~700 cycles
  // and the load path and the matrix pipe took turns instead of overlapping (timing-only builds: DMA alone 5.1 ms, fragment reads +
  // MFMAs alone 7.6 ms, together 10.1 ms).  One piece behind each slice's group of products -- the other wave of the SIMD multiplies
  // meanwhile -- takes 1.0 ms off (9 %).  Waves without products (above the diagonal of a diagonal tile) and two-buffer builds keep
  // the burst: first half of the waves before their products, second half behind.
  const bool dma_first = (NBUF < 3) || (wave < WAVES / 2) || (diag_flags & 32);
  const bool mid_group = NS <= 4 && wave >= WAVES / 2;    // (four slices: the SIMD's second wave issues in the middle of a group, -0.15 ms; six: no gain)
  for (int64_t kb = kb0; kb < kb1; ++kb) {
    const int64_t ahead = kb + NBUF - 1;
    const bool spread = !(diag_flags & (256 | 128 | 4)) && NBUF == 3 && active && ahead < kb1;   // one piece behind each slice's products
    if (ahead < kb1 && dma_first && !spread && !(diag_flags & 4)) {
      if (!(diag_flags & 16)) __builtin_amdgcn_s_setprio(2);
      issue(ahead, (cur + NBUF - 1) % NBUF);
      __builtin_amdgcn_s_setprio(0);
    }
    if (active && !(diag_flags & 128)) {
      // fragment of lane (r32, h): 16 bytes of row r at k half h = image[slice][h][row] (int4 units: slice 256 / 2 BROWS, half 128 / BROWS)
      const int4_t* __restrict__ ap = lds_i8 + cur * BUF + h * 128 + 64 * wr + r32;
      // B fragment rows: on a diagonal tile they are rows of the A image; else the B image holds the BROWS rows this unit needs
      const int4_t* __restrict__ bp = diag ? lds_i8 + cur * BUF + h * 128 + col0 + r32
                                           : lds_i8 + cur * BUF + OPA + h * BROWS + 32 * wc + r32;
      const int bstride = diag ? 256 : BROWS * 2;
      // the A fragments of slice s + 1 are requested BEFORE the products of slice s are issued: left to itself hipcc reads them
      // behind each group of MFMAs and then waits for the LDS with an empty matrix pipe, six times per step
      // The fragments of a step's first products come in two batches.  With LDS DMA in flight hipcc only ever writes `s_waitcnt
      // lgkmcnt(0)`, so a step used to open with ten reads per wave -- 80 KB through the LDS of a CU -- before its first MFMA.  Now
      // four reads (a0, a1, b0, b1), a wait (the empty asm is a use the compiler has to wait for), then the other six are requested
      // and fly under the first four products.
      int4_t b[NS];
      int4_t a0 = ap[0], a1 = ap[32];
      b[0] = bp[0];
      b[1] = bp[bstride];
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b[0]), "+v"(b[1]));
#pragma unroll
      for (int s = 2; s < NS; ++s) b[s] = bp[s * bstride];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        int4_t n0 = a0, n1 = a1;
        if (s + 1 < NS) {
          n0 = ap[(s + 1) * 256];
          n1 = ap[(s + 1) * 256 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < NS; ++tt)
          if (s + tt < NS) {
            if (spread && mid_group && tt == (NS - s) / 2) {
              __builtin_amdgcn_sched_barrier(0);
              issue_some(ahead, (cur + NBUF - 1) % NBUF, s, s + 1 < NS ? 1 : 1 << 20);
              __builtin_amdgcn_sched_barrier(0);
            }
            acc[s + tt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[tt], acc[s + tt][0], 0, 0, 0);
            acc[s + tt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[tt], acc[s + tt][1], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (spread && !mid_group) {
          issue_some(ahead, (cur + NBUF - 1) % NBUF, s, s + 1 < NS ? 1 : 1 << 20);
          __builtin_amdgcn_sched_barrier(0);
        }
        a0 = n0;
        a1 = n1;
      }
    }
    if (ahead < kb1 && !dma_first && !spread && !(diag_flags & 4)) issue(ahead, (cur + NBUF - 1) % NBUF);
    // K-block kb + 1 must have landed; with three buffers the pieces of kb + 2 issued in this step may stay in flight
    if (!(diag_flags & 8)) wait_all_but_newest(NBUF == 3 && ahead < kb1);
    if (!(diag_flags & 2)) __syncthreads();
    cur = (cur + 1) % NBUF;
  }
  // epilogue: C/D element reg of lane (r32, h) of block m: row = 64 wr + 32 m + (reg & 3) + 8 (reg >> 2) + 4 h, column = col0 + r32
  double* __restrict__ slab = out + ((int64_t)split * ntiles + tile) * (TILE * TILE);
  const int ecol = expo[tj * 128 + col0 + r32];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = 64 * wr + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      double v = (double)acc[NS - 1][m][reg];
#pragma unroll
      for (int k = NS - 2; k >= 0; --k) v = fma(v, BITS == 7 ? 0.0078125 : 0.00390625, (double)acc[k][m][reg]);   // Horner in 2^-BITS, smallest diagonal first
      slab[row * TILE + col0 + r32] = ldexp(v, expo[ti * 128 + row] + ecol - (BITS == 7 ? 12 : 14));
    }
}

struct I8Plan {
  int nb, ntiles, nsplit, Ppad;
  int64_t nkb, kb_per_split;
};

I8Plan make_i8_plan(int P, int64_t N, int bits, int n_slices = 6) {
  I8Plan g;
  g.nb = (P + TILE - 1) / TILE;
  g.ntiles = (int)tri_count(g.nb);
  g.Ppad = g.nb * TILE;
  g.nkb = (N + KB - 1) / KB;
  // one 8-wave workgroup per CU: 256 slots; aim at >= 8 rounds, the last one full.  int32 exactness bounds the samples of a split:
  // 7-bit digits (|d| <= 64, <= 7 pairs per diagonal) 65 536; 8-bit digits (|d| <= 128, <= 6 pairs) 16 384
  const int64_t cap = (bits == 7) ? 65536 : (n_slices <= 4 ? 32768 : 16384);      // (8-bit digits: n_slices pairs of <= 2^14 per diagonal)
  const int64_t min_split = (g.nkb * KB + cap - 1) / cap;
  int64_t want = (8 * 256 + g.ntiles - 1) / g.ntiles;
  if (want < min_split) want = min_split;
  int64_t maxsplit = g.nkb / 64 > 0 ? g.nkb / 64 : 1;
  if (maxsplit < min_split) maxsplit = min_split;
  int64_t pick = want < maxsplit ? want : maxsplit;
  double best = 1e30;
  for (int64_t c = std::max<int64_t>(min_split, pick > 3 ? pick - 2 : 1); c <= pick + 8 && c <= maxsplit; ++c) {
    const int64_t rounds = ((g.ntiles * c + 7) / 8 + 31) / 32;          // per XCD: 32 resident workgroups, tiles dealt evenly
    const double cost = (double)rounds / (double)c;
    if (cost < best - 1e-9) { best = cost; pick = c; }
  }
  pick = env_int("VMCPDE_GRAM_I8_NSPLIT", (int)pick);
  if (pick < min_split) pick = min_split;
  g.kb_per_split = (g.nkb + pick - 1) / pick;
  g.nsplit = (int)((g.nkb + g.kb_per_split - 1) / g.kb_per_split);
  return g;
}

__global__ void reduce_splits_i8_kernel(const double* __restrict__ slabs, int nsplit, int64_t per_split, double* __restrict__ packed) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= per_split) return;
  double2_t s = *reinterpret_cast<const double2_t*>(slabs + i);
  for (int k = 1; k < nsplit; ++k) {
    const double2_t v = *reinterpret_cast<const double2_t*>(slabs + (int64_t)k * per_split + i);
    s.x += v.x;
    s.y += v.y;
  }
  *reinterpret_cast<double2_t*>(packed + i) = s;
}

int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t vmcpde_gram_i8_workspace_bytes(int32_t P, int64_t n_samples, int32_t n_slices, int32_t digit_bits) {
  if (P <= 0 || n_samples <= 0 || n_slices < 2 || n_slices > 8 || (digit_bits != 7 && digit_bits != 8)) return 0;
  const I8Plan g = make_i8_plan(P, n_samples, digit_bits, n_slices);
  return align256((int64_t)g.Ppad * (int64_t)sizeof(int)) + align256((int64_t)g.Ppad * 24) + align256(g.nkb * n_slices * (int64_t)g.Ppad * KB) +
         (int64_t)g.nsplit * g.ntiles * TILE * TILE * (int64_t)sizeof(double);
}

template <int NS, int BITS, typename TIn>
static int gram_i8_run(const TIn* JT, int64_t ldj, int P, int64_t N, const double* c, const double* w, double* packed, double* stats, char* ws, hipStream_t st) {
  const I8Plan g = make_i8_plan(P, N, BITS, NS);
  int* expo = reinterpret_cast<int*>(ws);
  ws += align256((int64_t)g.Ppad * (int64_t)sizeof(int));
  unsigned long long* dsum = reinterpret_cast<unsigned long long*>(ws);
  ws += align256((int64_t)g.Ppad * 24);
  int8_t* planes = reinterpret_cast<int8_t*>(ws);
  double* slabs = reinterpret_cast<double*>(ws + align256(g.nkb * NS * (int64_t)g.Ppad * KB));
  VMC_CHECK_HIP(hipMemsetAsync(dsum, 0, (size_t)g.Ppad * 24, st));
  if (stats) VMC_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double), st));
  hipLaunchKernelGGL(row_expo_kernel<TIn>, dim3(g.Ppad), dim3(256), 0, st, JT, ldj, N, P, BITS == 8 ? 1 : 0, c, w, expo);
  hipLaunchKernelGGL((slice_kernel<NS, BITS, TIn>), dim3(g.Ppad / SL_ROWS, (unsigned)((g.nkb + SL_KB_PER_WG - 1) / SL_KB_PER_WG)), dim3(256), 0, st, JT, ldj, N, P,
                     g.Ppad, expo, c, w, planes, g.nkb, dsum);
  constexpr int WAVES = (NS <= 6) ? 8 : 4;          // 7 diagonals of a 64 x 32 wave tile need more than 256 registers
  const size_t one = (size_t)NS * (128 + (WAVES == 8 ? 128 : 64)) * KB;          // one K-block: A + B
  const size_t lds = (3 * one <= 160 * 1024 ? 3 : 2) * one;
  DeviceCtx* ctx = device_ctx();
  if (!ctx) return -2;
  DeviceCtxLock lock;              // first-use attribute change and the tile-order upload; held over the asynchronous launches (harmless)
  bool& attr_done = ctx->attr_gram_i8[NS == 7 ? 2 : (NS == 4 ? 3 : (BITS == 7 ? 1 : 0))];
  if (!attr_done) {
    VMC_CHECK_HIP(hipFuncSetAttribute((const void*)gram_i8_kernel<NS, WAVES, BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  double* out = (g.nsplit > 1) ? slabs : packed;
  const int* perm = nullptr;
  int per_xcd = 0;
  unsigned grid = (unsigned)(g.ntiles * g.nsplit * 2);
  if (WAVES == 8) {
    if (int rc = ensure_gram_perm(ctx, g.nb, &perm)) return rc;
    per_xcd = order2_slots_per_xcd(g.ntiles, g.nsplit);
    grid = (unsigned)per_xcd * 8u;
  }
  hipLaunchKernelGGL((gram_i8_kernel<NS, WAVES, BITS>), dim3(grid), dim3(WAVES * 64), lds, st, planes, expo, g.Ppad, g.nkb, g.ntiles, g.nsplit,
                     g.kb_per_split, per_xcd, env_int("VMCPDE_GRAM_I8_DIAG", 0), perm, out);
  VMC_CHECK_HIP(hipGetLastError());
  if (g.nsplit > 1) {
    const int64_t per_split = (int64_t)g.ntiles * TILE * TILE;
    hipLaunchKernelGGL(reduce_splits_i8_kernel, dim3((unsigned)((per_split / 2 + 255) / 256)), dim3(256), 0, st, slabs, g.nsplit, per_split, packed);
    VMC_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(diag_fix_kernel, dim3((P + 255) / 256), dim3(256), 0, st, dsum, expo, P, NS, BITS, N, packed, stats);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_gram_i8(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, int32_t n_slices, int32_t digit_bits,
                              const double* c, const double* w, double* packed, double* stats, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  VMC_REQUIRE(JT && packed && workspace, "vmcpde_gram_i8: null argument");
  VMC_REQUIRE(P >= 1 && n_samples >= 1, "bad shape");
  VMC_REQUIRE(ldj >= n_samples && (ldj % 2) == 0, "ldj must be even and >= n_samples");
  VMC_REQUIRE((reinterpret_cast<uintptr_t>(JT) & 15) == 0, "JT must be 16-byte aligned");
  VMC_REQUIRE((n_slices == 6 && (digit_bits == 7 || digit_bits == 8)) || (n_slices == 7 && digit_bits == 7) || (n_slices == 4 && digit_bits == 8),
              "supported: 6 slices of 7 or 8 bits, 7 slices of 7 bits, 4 slices of 8 bits");
  VMC_REQUIRE(!jt_f32 || (n_slices == 4 || (n_slices == 6 && digit_bits == 8)), "an fp32 Jacobian is sliced into 4 (or 6) digits of 8 bits");
  VMC_REQUIRE(workspace_bytes >= vmcpde_gram_i8_workspace_bytes(P, n_samples, n_slices, digit_bits), "gram_i8 workspace too small: need %lld bytes",
              (long long)vmcpde_gram_i8_workspace_bytes(P, n_samples, n_slices, digit_bits));
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  const double* J64 = static_cast<const double*>(JT);
  const float* J32 = static_cast<const float*>(JT);
  if (n_slices == 4) return jt_f32 ? gram_i8_run<4, 8, float>(J32, ldj, P, n_samples, c, w, packed, stats, ws, st)
                                   : gram_i8_run<4, 8, double>(J64, ldj, P, n_samples, c, w, packed, stats, ws, st);
  if (jt_f32) return gram_i8_run<6, 8, float>(J32, ldj, P, n_samples, c, w, packed, stats, ws, st);
  if (n_slices == 7) return gram_i8_run<7, 7, double>(J64, ldj, P, n_samples, c, w, packed, stats, ws, st);
  return digit_bits == 8 ? gram_i8_run<6, 8, double>(J64, ldj, P, n_samples, c, w, packed, stats, ws, st)
                         : gram_i8_run<6, 7, double>(J64, ldj, P, n_samples, c, w, packed, stats, ws, st);
}
