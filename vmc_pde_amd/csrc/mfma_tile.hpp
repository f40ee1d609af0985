// fp64 MFMA tile engine shared by the Gram (gram.hip) and the blocked Cholesky (solve.hip) kernels.
//
// Computes one 128x128 tile of C = A * B^T for row-major operands whose K index is contiguous ("TN" GEMM):
//     C[i][j] = sum_k (A[rowA0 + i][k] - shiftA[i]) * sc[k] * (B[rowB0 + j][k] - shiftB[j]) * sc[k]
// on v_mfma_f64_16x16x4_f64 (lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the C/D block holds
// col = l&15, row = (l>>4) + 4*reg -- the f64 layout differs from every other MFMA dtype, cdna_hip_programming.md:159-162).
// Workgroup = 256 threads = 4 waves as 2x2, 64x64 per wave = 4x4 MFMA blocks = 128 accumulator VGPRs, so two
// workgroups fit per CU (two waves per SIMD: one feeds the matrix pipe while its partner stages).
// K-step BK = 16: operands go global -> registers (issued one K-step ahead, centred / scaled in flight) -> LDS rows of
// 17 doubles (pitch 136 B, conflict-free fragment reads; staged with 8-byte LDS writes since rows are only 8-B aligned).
//
// Contract (checked by the host wrappers): operand matrices have all 128 rows of every tile allocated (row padding
// is zero-filled by the owner), row pitch even (16-B aligned double2 loads), 128 * pitch * 8 < 2^32.
#pragma once
#include "common.hpp"

namespace vmcpde {

#ifdef VMC_GRAM_STAMPS
// diagnostic build only (tools/probe_gram_stamps.py): per-workgroup cycle sums of the main-loop segments
__device__ unsigned long long g_stamps[16384 * 8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define VMC_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = stamp(); seg[i] += _t - tprev; tprev = _t; __builtin_amdgcn_sched_barrier(0); }
#else
#define VMC_STAMP(i)
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int TILE = VMCPDE_GRAM_TILE;  // 128
constexpr int BK = 16;
constexpr int LDP = BK + 1;             // LDS row pitch in doubles: 34 dwords = 2 (mod 32) and (mod 64), so the 16 rows one
                                        // fragment read touches sit on distinct banks for ds_read_b64 AND ds_read2_b64 (hipcc
                                        // merges two K-columns into read2, which banks mod 32: pitch 18 measured 40 % conflict cycles)
constexpr int TILE_LDS = TILE * LDP;    // doubles per staged operand tile
constexpr int TILE_LDS_DOUBLES = 4 * TILE_LDS + 2 * TILE;   // 2 operands x 2 buffers + the two shift vectors
constexpr size_t TILE_LDS_BYTES = TILE_LDS_DOUBLES * sizeof(double);   // 75,776 B -> two workgroups per CU

__host__ __device__ inline int64_t tri_count(int nb) { return (int64_t)nb * (nb + 1) / 2; }

__device__ __forceinline__ void tri_decode(int t, int& ti, int& tj) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((int64_t)(i + 1) * (i + 2) / 2 <= t) ++i;
  while ((int64_t)i * (i + 1) / 2 > t) --i;
  ti = i;
  tj = t - i * (i + 1) / 2;
}

// Global -> register staging of one 128 x BK operand tile (rows = parameters, columns = samples):
// thread t, slot q: row = q*32 + t/8, k-pair = (t%8)*2 -- eight lanes cover one 128-B row segment.
// `tile` is wave-uniform (row block base + k), `off` the per-thread element offset of slot 0; slots are 32 rows apart.
// stage_load only ISSUES the loads (raw values stay in flight behind the MFMAs of the current K-step);
// stage_store centres / scales them and writes the LDS image, so the vmcnt wait lands after the matrix work.
typedef float float2_t __attribute__((ext_vector_type(2)));
template <typename T> struct PairOf;
template <> struct PairOf<double> { typedef double2_t type; };
template <> struct PairOf<float> { typedef float2_t type; };

// TIn = double: the fp64 Jacobian; TIn = float: BASELINE config 5 (fp32 Jacobian, fp64 S/F accumulation) -- the pairs are
// widened to fp64 when they are committed to LDS, the MFMAs are the same v_mfma_f64.
template <typename TIn>
struct Staged {
  typename PairOf<TIn>::type v[4];
  double2_t sc;
};

template <bool SCALE, bool TAIL, typename TIn>
__device__ __forceinline__ void stage_load(Staged<TIn>& r, const TIn* __restrict__ tile, uint32_t off, uint32_t slot_stride,
                                           const double* __restrict__ scale_k /* scale + k, or null */, int64_t krem) {
  const int t = threadIdx.x;
  const int kp = (t & 7) * 2;
  if (SCALE) {
    if (!TAIL) r.sc = *reinterpret_cast<const double2_t*>(scale_k + kp);
    else { r.sc.x = (kp < krem) ? scale_k[kp] : 0.0; r.sc.y = (kp + 1 < krem) ? scale_k[kp + 1] : 0.0; }
  }
  // wave-uniform base (SGPR pair) + 32-bit per-lane BYTE offset: global_load ... v_off, s[base:base+1] -- no 64-bit
  // VALU address arithmetic in the loop (fp64 MFMA and VALU share the SIMD's issue, so every VALU op here stretches
  // the staging burst; the host wrappers guarantee 128 * pitch * 8 < 2^32)
  const char* __restrict__ base = reinterpret_cast<const char*>(tile);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t boff = (off + q * slot_stride) * (uint32_t)sizeof(TIn);
    if (!TAIL) {
      r.v[q] = *reinterpret_cast<const typename PairOf<TIn>::type*>(base + boff);
    } else {
      const TIn* p = reinterpret_cast<const TIn*>(base + boff);
      r.v[q].x = (kp < krem) ? p[0] : (TIn)0;
      r.v[q].y = (kp + 1 < krem) ? p[1] : (TIn)0;
    }
  }
}

template <bool CENTER, bool SCALE, bool TAIL, typename TIn>
__device__ __forceinline__ void stage_store(const Staged<TIn>& r, double* __restrict__ lds, int tile_off, const double (&sh)[4], int64_t krem) {
  const int t = threadIdx.x;
  const int kp = (t & 7) * 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = q * 32 + (t >> 3);
    double2_t v = {(double)r.v[q].x, (double)r.v[q].y};
    if (CENTER) {
      v.x -= sh[q];
      v.y -= sh[q];
    }
    if (TAIL) {   // columns beyond the sample count must contribute 0, not -shift
      if (!(kp < krem)) v.x = 0.0;
      if (!(kp + 1 < krem)) v.y = 0.0;
    }
    if (SCALE) { v.x *= r.sc.x; v.y *= r.sc.y; }
    lds[tile_off + row * LDP + kp] = v.x;       // odd pitch: 8-byte stores (ds_write2_b64)
    lds[tile_off + row * LDP + kp + 1] = v.y;
  }
}

// acc[i][j] += A_tile(64 rows of this wave) * B_tile(64 rows of this wave)^T over K-columns [4*K4A, 4*K4B) of one staged K-step
// LOWER: this wave's quadrant sits on the diagonal of a diagonal tile -- only the 16 x 16 blocks with j <= i are formed.
template <int K4A, int K4B, bool LOWER>
__device__ __forceinline__ void mma_part_blocks(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs,
                                                int wr, int wc) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* __restrict__ ap = lds + offAs + (64 * wr + lr) * LDP + lk;
  const double* __restrict__ bp = lds + offBs + (64 * wc + lr) * LDP + lk;
#pragma unroll
  for (int k4 = K4A; k4 < K4B; ++k4) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = ap[16 * i * LDP + 4 * k4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = bp[16 * j * LDP + 4 * k4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!LOWER || j <= i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// On a diagonal tile (alias) the quadrant above the diagonal is skipped; consumers mirror the lower triangle.
template <int K4A, int K4B>
__device__ __forceinline__ void mma_part(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs,
                                         int wr, int wc, bool alias) {
  if (!alias || wr >= wc) mma_part_blocks<K4A, K4B, false>(acc, lds, offAs, offBs, wr, wc);
}

__device__ __forceinline__ void mma_kstep(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs,
                                          int wr, int wc, bool alias) {
  mma_part<0, BK / 4>(acc, lds, offAs, offBs, wr, wc, alias);
}

// Main loop of one tile.  alias: the B tile is the A tile (diagonal tile of a syrk) -> only A is staged.
// shiftA/shiftB may be null (no centring); they are read for `nshift*` valid rows and zero-extended.
// lds: TILE_LDS_DOUBLES doubles.
template <bool CENTER, bool SCALE, typename TIn = double>
__device__ __forceinline__ void tile_mainloop(double4_t (&acc)[4][4], const TIn* __restrict__ A, int64_t lda, int rowA0,
                                              const TIn* __restrict__ B, int64_t ldb, int rowB0, bool alias, int64_t k0,
                                              int64_t k1, const double* __restrict__ shiftA, int nshiftA,
                                              const double* __restrict__ shiftB, int nshiftB, const double* __restrict__ scale,
                                              double* __restrict__ lds) {
  const int t = threadIdx.x;
  const int wave = t >> 6, wr = wave >> 1, wc = wave & 1;
  // LDS map (doubles): A buffers at 0 and 2*TILE_LDS, B buffers at TILE_LDS and 3*TILE_LDS, shifts behind them.
  // Offsets stay integers so every access keeps the LDS address space (pointer selects degrade to flat_*).
  double* shA = lds + 4 * TILE_LDS;
  double* shB = shA + TILE;
  if (t < TILE) {
    shA[t] = (shiftA != nullptr && t < nshiftA) ? shiftA[rowA0 + t] : 0.0;
    shB[t] = (shiftB != nullptr && t < nshiftB) ? shiftB[rowB0 + t] : 0.0;
  }
  __syncthreads();
  const TIn* __restrict__ tA = A + (int64_t)rowA0 * lda;
  const TIn* __restrict__ tB = B + (int64_t)rowB0 * ldb;
  const uint32_t offA = (uint32_t)(t >> 3) * (uint32_t)lda + (uint32_t)(t & 7) * 2u, strideA = 32u * (uint32_t)lda;
  const uint32_t offB = (uint32_t)(t >> 3) * (uint32_t)ldb + (uint32_t)(t & 7) * 2u, strideB = 32u * (uint32_t)ldb;
  const int64_t nfull = (k1 - k0) / BK;          // full K-steps
  const int64_t tail = (k1 - k0) - nfull * BK;   // ragged remainder (N not a multiple of 16)
  double SHA[4], SHB[4];   // this thread's four row shifts per operand, loop-invariant
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    SHA[q] = shA[q * 32 + (t >> 3)];
    SHB[q] = shB[q * 32 + (t >> 3)];
  }
  const int boff = alias ? 0 : TILE_LDS;
  Staged<TIn> ra, rb;

  // prologue: stage the first K-step
  if (nfull > 0) {
    stage_load<SCALE, false>(ra, tA + k0, offA, strideA, SCALE ? scale + k0 : nullptr, BK);
    if (!alias) stage_load<SCALE, false>(rb, tB + k0, offB, strideB, SCALE ? scale + k0 : nullptr, BK);
    stage_store<CENTER, SCALE, false>(ra, lds, 0, SHA, BK);
    if (!alias) stage_store<CENTER, SCALE, false>(rb, lds, TILE_LDS, SHB, BK);
  } else {
    stage_load<SCALE, true>(ra, tA + k0, offA, strideA, SCALE ? scale + k0 : nullptr, tail);
    if (!alias) stage_load<SCALE, true>(rb, tB + k0, offB, strideB, SCALE ? scale + k0 : nullptr, tail);
    stage_store<CENTER, SCALE, true>(ra, lds, 0, SHA, tail);
    if (!alias) stage_store<CENTER, SCALE, true>(rb, lds, TILE_LDS, SHB, tail);
  }
  __syncthreads();
  int cur = 0;
  // steady state.  Registers ra/rb hold the raw loads of step s+1 on entry.  Each iteration: first half of the MFMAs of
  // step s; then (loads have been in flight for a full iteration) centre + write step s+1 into the other LDS buffer and
  // immediately re-issue the loads of step s+2 into the same registers; second half of the MFMAs covers the LDS writes,
  // so the closing barrier finds them done.  sched_barriers pin this order (hipcc would hoist the vmcnt wait).
  int64_t k = k0;
  if (nfull > 1) {
    stage_load<SCALE, false>(ra, tA + k + BK, offA, strideA, SCALE ? scale + k + BK : nullptr, BK);
    if (!alias) stage_load<SCALE, false>(rb, tB + k + BK, offB, strideB, SCALE ? scale + k + BK : nullptr, BK);
  }
#ifdef VMC_GRAM_STAMPS
  unsigned long long seg[4] = {0, 0, 0, 0}, segw = 0;
  unsigned long long tprev = stamp();
  const unsigned long long tstart = tprev;
#endif
  for (int64_t s = 0; s + 1 < nfull; ++s, k += BK) {
    mma_part<0, BK / 8>(acc, lds, cur * 2 * TILE_LDS, cur * 2 * TILE_LDS + boff, wr, wc, alias);
    VMC_STAMP(0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);   // let the short VALU/LDS staging burst win issue slots over the partner's MFMAs (+0.5 % measured)
#ifdef VMC_GRAM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = stamp(); segw += _t - tprev; tprev = _t; __builtin_amdgcn_sched_barrier(0); }
#endif
    stage_store<CENTER, SCALE, false>(ra, lds, (cur ^ 1) * 2 * TILE_LDS, SHA, BK);
    if (!alias) stage_store<CENTER, SCALE, false>(rb, lds, (cur ^ 1) * 2 * TILE_LDS + TILE_LDS, SHB, BK);
    if (s + 2 < nfull) {
      stage_load<SCALE, false>(ra, tA + k + 2 * BK, offA, strideA, SCALE ? scale + k + 2 * BK : nullptr, BK);
      if (!alias) stage_load<SCALE, false>(rb, tB + k + 2 * BK, offB, strideB, SCALE ? scale + k + 2 * BK : nullptr, BK);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    VMC_STAMP(1)
    mma_part<BK / 8, BK / 4>(acc, lds, cur * 2 * TILE_LDS, cur * 2 * TILE_LDS + boff, wr, wc, alias);
    VMC_STAMP(2)
    __syncthreads();
    VMC_STAMP(3)
    cur ^= 1;
  }
#ifdef VMC_GRAM_STAMPS
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    unsigned long long* o = g_stamps + ((size_t)(blockIdx.x % 4096) * 4 + w) * 8;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3]; o[4] = (unsigned long long)(nfull > 0 ? nfull - 1 : 0); o[5] = tprev - tstart;
    o[6] = segw; o[7] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11)) /* HW_ID */;
  }
#endif
  if (nfull > 0) {   // last full step; the ragged tail (if any) is staged behind it
    if (tail > 0) {
      stage_load<SCALE, true>(ra, tA + k + BK, offA, strideA, SCALE ? scale + k + BK : nullptr, tail);
      if (!alias) stage_load<SCALE, true>(rb, tB + k + BK, offB, strideB, SCALE ? scale + k + BK : nullptr, tail);
    }
    mma_kstep(acc, lds, cur * 2 * TILE_LDS, cur * 2 * TILE_LDS + boff, wr, wc, alias);
    if (tail > 0) {
      stage_store<CENTER, SCALE, true>(ra, lds, (cur ^ 1) * 2 * TILE_LDS, SHA, tail);
      if (!alias) stage_store<CENTER, SCALE, true>(rb, lds, (cur ^ 1) * 2 * TILE_LDS + TILE_LDS, SHB, tail);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (tail > 0) {
    mma_kstep(acc, lds, cur * 2 * TILE_LDS, cur * 2 * TILE_LDS + boff, wr, wc, alias);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// DMA variant of the main loop (no centring, no weights): operands go global -> LDS directly (global_load_lds_dwordx4,
// one 1-KiB piece = 8 rows x 128 B per wave-instruction), so a K-step's staging burst is 8 instructions per wave and no
// VGPR, VALU or ds_write is spent on it -- on this chip every non-MFMA instruction of a wave waits ~one MFMA slot (64
// cycles) behind the co-resident wave's fp64 MFMAs, so the burst length is what the matrix pipe idles on.
// LDS image: unpadded rows of 16 doubles; the eight 16-B pairs of a row are XOR-swizzled with f(row) = (row>>1)&7, applied
// on the per-lane SOURCE address (the DMA destination is lane-linear) and again on the fragment reads, which makes the 16
// rows x 2 columns of a ds_read_b64 half-wave hit 32 distinct bank pairs.
// The DMA path is templated on its own K-step BKD (doubles per row of the LDS image):
//   BKD = 16: rows of 128 B, 64 KiB per workgroup -> two workgroups per CU   (the Cholesky's K = 128 updates)
//   BKD =  8: rows of  64 B, 32 KiB per workgroup -> three workgroups per CU (VGPR-bound; the Gram kernel): a third wave per
//             SIMD fills the matrix-pipe slots the other two leave while they issue DMA pieces or wait at a barrier.
template <int BKD> struct DmaGeom {
  static constexpr int DT = TILE * BKD;                       // doubles per operand tile
  static constexpr int PAIRS = BKD / 2;                       // 16-B pairs per row (8 or 4)
  static constexpr int ROWS_PER_PIECE = 64 / PAIRS;           // rows one 1-KiB wave-instruction covers (8 or 16)
  static constexpr int PIECES = TILE / ROWS_PER_PIECE / 4;    // pieces per wave per operand tile (4 or 2)
  // XOR swizzle of the pair index: rows whose images share banks (128-B rows: r, r+2, ..; 64-B rows: r, r+4, ..) get
  // different pair slots, so the 16 rows x 2 columns of a ds_read_b64 half-wave hit 32 distinct bank pairs
  __device__ static constexpr int f(int row) { return BKD == 16 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};
constexpr int DT = DmaGeom<16>::DT;
constexpr size_t TILE_DMA_LDS_BYTES = 4 * DT * sizeof(double);           // BKD = 16
constexpr size_t TILE_DMA8_LDS_BYTES = 4 * DmaGeom<8>::DT * sizeof(double);

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// issue the pieces this wave owns of one 128 x BKD operand tile
template <int BKD>
__device__ __forceinline__ void dma_tile(const double* __restrict__ tile_k /* row-block base + k */, const uint32_t (&boff)[DmaGeom<BKD>::PIECES],
                                         double* __restrict__ lds_tile, int wave) {
  using G = DmaGeom<BKD>;
  const char* __restrict__ base = reinterpret_cast<const char*>(tile_k);
#pragma unroll
  for (int jj = 0; jj < G::PIECES; ++jj)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + boff[jj]), (lds_ptr_t)(lds_tile + (wave * G::PIECES + jj) * G::ROWS_PER_PIECE * BKD), 16, 0, 0);
}

template <int BKD>
__device__ __forceinline__ void mma_kstep_swz(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs, int wr, int wc,
                                              const int (&koff)[BKD / 4]) {
  const int lr = threadIdx.x & 15;
  const double* __restrict__ ap = lds + offAs + (64 * wr + lr) * BKD;
  const double* __restrict__ bp = lds + offBs + (64 * wc + lr) * BKD;
#pragma unroll
  for (int k4 = 0; k4 < BKD / 4; ++k4) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = ap[16 * i * BKD + koff[k4]];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = bp[16 * j * BKD + koff[k4]];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// fp32 variant (BASELINE config 5, "f32 blocks"): the LDS image holds fp32 values, two per 8-byte slot, so the same swizzled
// ds_read_b64 fragment reads deliver a float pair per lane and every slot feeds TWO v_mfma_f32_16x16x4_f32 (32 cycles each:
// the same 64 matrix-pipe cycles as one fp64 MFMA, for twice the samples).  Which sample lands in which MFMA k index is
// irrelevant for a contraction as long as A and B agree, and they do: both use slot d = 4*k4 + lane/16 -> samples 2d, 2d+1.
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int BKD>
__device__ __forceinline__ void mma_kstep_swz_f32(float4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs, int wr, int wc,
                                                  const int (&koff)[BKD / 4]) {
  const int lr = threadIdx.x & 15;
  const double* __restrict__ ap = lds + offAs + (64 * wr + lr) * BKD;
  const double* __restrict__ bp = lds + offBs + (64 * wc + lr) * BKD;
#pragma unroll
  for (int k4 = 0; k4 < BKD / 4; ++k4) {
    float2_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float2_t*>(ap + 16 * i * BKD + koff[k4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float2_t*>(bp + 16 * j * BKD + koff[k4]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
      }
  }
}

// The same fragments WIDENED: both floats of a slot become doubles and feed two fp64 MFMAs into the fp64 accumulators (fp64 C/D map) --
// BASELINE config 5 proper (fp32 Jacobian storage, fp64 products and sums) on the DMA-staged loop: half the staged bytes, half the
// fragment reads per MFMA, and the conversions (one per operand value, 0.5 per MFMA) on the VALU instead of the register-staged loop's
// conversion while committing to LDS.
template <int BKD>
__device__ __forceinline__ void mma_kstep_swz_w32(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs, int wr, int wc,
                                                  const int (&koff)[BKD / 4]) {
  const int lr = threadIdx.x & 15;
  const double* __restrict__ ap = lds + offAs + (64 * wr + lr) * BKD;
  const double* __restrict__ bp = lds + offBs + (64 * wc + lr) * BKD;
#pragma unroll
  for (int k4 = 0; k4 < BKD / 4; ++k4) {
    float2_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float2_t*>(ap + 16 * i * BKD + koff[k4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float2_t*>(bp + 16 * j * BKD + koff[k4]);
    double ax[4], ay[4], bx[4], by[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ax[i] = (double)a[i].x; ay[i] = (double)a[i].y; bx[i] = (double)b[i].x; by[i] = (double)b[i].y; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[i], bx[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay[i], by[j], acc[i][j], 0, 0, 0);
      }
  }
}

// fp64 running sums += fp32 block sums; the fp64 registers simply shadow the fp32 C/D map (row = 4*(lane/16) + reg)
__device__ __forceinline__ void flush_f32_blocks(double4_t (&acc)[4][4], float4_t (&acc32)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += (double)acc32[i][j][r];
      acc32[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
}

// One K step of this wave's quadrant.  On a diagonal tile (alias) the quadrant above the diagonal is skipped (a quarter of
// the tile's MFMAs; the freed matrix-pipe slots go to the co-resident workgroups' waves).  Also forming only the lower
// 16 x 16 blocks of the two quadrants ON the diagonal needs a second copy of the K step, and with two copies hipcc spills
// ~190 VGPRs around the 128 accumulator registers -- measured 2x slower, so it is not done.  Round 2 tried the balanced split
// twice more (10 / 8 / 8 / 10 lower blocks per wave instead of 16 / 0 / 16 / 16): (a) one loop with a wave-uniform block mask
// and a scalar branch around each MFMA -- the whole Gram 12 % SLOWER (17.25 -> 19.4 ms; the branches break the interleave of
// fragment reads and MFMAs); (b) a separate branch-free loop per wave role for diagonal tiles (no spills, 175 VGPRs) -- no
// change at P = 4213 / 4155 (17.17 vs 17.18 ms, 8.68 vs 8.68 ms, alternating runs on one box) and 4 % slower at P = 16 181
// (259.9 vs 249.8 ms).  Freed matrix-pipe slots do not turn into time on this chip: the fp64 MFMA rate is power-managed.
template <int BKD>
__device__ __forceinline__ void mma_kstep_sym(double4_t (&acc)[4][4], const double* __restrict__ lds, int offAs, int offBs, int wr, int wc,
                                              const int (&koff)[BKD / 4], bool alias) {
  if (!alias || wr >= wc) mma_kstep_swz<BKD>(acc, lds, offAs, offBs, wr, wc, koff);
}

// lds: 4 * DmaGeom<BKD>::DT doubles.  Same contract as tile_mainloop<false,false>; with alias (diagonal tile) only the lower
// triangle of the tile (row >= col) is accumulated.
// F32: the operands are fp32 arrays viewed as doubles (lda, k0, k1 in 8-byte units, i.e. sample pairs; padding samples must be
// zero); products are summed in fp32 by the matrix pipe over blocks of F32_BLOCK_STEPS K steps (<= 128 samples) and the block
// sums are accumulated in fp64 -- acc then follows the fp32 C/D register map (tile_store_f32map).
constexpr int F32_BLOCK_STEPS = 4;
// MODE 0: fp64 operands.  MODE 1 (F32 blocks) and MODE 2 (fp32 operands widened to fp64 MFMAs, fp64 C/D map): sample pairs as above.
template <int BKD, int MODE = 0>
__device__ __forceinline__ void tile_mainloop_dma(double4_t (&acc)[4][4], const double* __restrict__ A, int64_t lda, int rowA0,
                                                  const double* __restrict__ B, int64_t ldb, int rowB0, bool alias, int64_t k0,
                                                  int64_t k1, double* __restrict__ lds, int tid = -1, int rows_valid_a = TILE) {
  // rows_valid_a: rows of the A block that hold data (the rest is zero padding).  A wave whose 64 rows are ALL padding forms no products
  // (round 6): the last tile row of a P that ends in the first half of a tile -- P + 2 = 4157 = 32 x 128 + 61 at BASELINE C4 -- is 33 tiles of
  // which the lower half is zeros; its accumulators stay zero, which is what the products would have been.
  using G = DmaGeom<BKD>;
  constexpr int DTB = G::DT;
  // tid: index of this thread among the 256 that form the tile (default threadIdx.x; workgroups that carry two tiles pass
  // threadIdx.x & 255 and a private `lds` per tile -- both tiles must then run the same number of K steps, the barriers are shared)
  const int t = (tid < 0) ? (int)threadIdx.x : tid, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 1, wc = wave & 1;
  const double* __restrict__ tA = A + (int64_t)rowA0 * lda;
  const double* __restrict__ tB = B + (int64_t)rowB0 * ldb;
  // per-lane source byte offsets of this wave's pieces: lane -> (row in piece = lane / PAIRS, slot = lane % PAIRS)
  uint32_t offA[G::PIECES], offB[G::PIECES];
#pragma unroll
  for (int jj = 0; jj < G::PIECES; ++jj) {
    const int row = (wave * G::PIECES + jj) * G::ROWS_PER_PIECE + lane / G::PAIRS;
    const int pair = (lane % G::PAIRS) ^ G::f(row);
    offA[jj] = ((uint32_t)row * (uint32_t)lda + (uint32_t)pair * 2u) * 8u;
    offB[jj] = ((uint32_t)row * (uint32_t)ldb + (uint32_t)pair * 2u) * 8u;
  }
  // fragment read offsets inside a row: element k = 4*k4 + lane/16 sits at pair (k/2) ^ f(row); f depends on lane&15 only
  int koff[BKD / 4];
  {
    const int lk = lane >> 4, f = G::f(lane & 15);
#pragma unroll
    for (int k4 = 0; k4 < BKD / 4; ++k4) koff[k4] = (((2 * k4 + (lk >> 1)) ^ f) << 1) + (lk & 1);
  }
  constexpr bool F32 = MODE == 1;
  float4_t acc32[F32 ? 4 : 1][F32 ? 4 : 1];
  if constexpr (F32) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc32[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bool active = (!alias || wr >= wc) && (64 * wr < rows_valid_a);      // the quadrant above the diagonal of a diagonal tile forms no products; nor do rows of padding
  const int boff = alias ? 0 : DTB;
  const int64_t nfull = (k1 - k0) / BKD;
  const int64_t tail = (k1 - k0) - nfull * BKD;
  int cur = 0;
  int64_t k = k0;
  if (nfull > 0) {
    dma_tile<BKD>(tA + k, offA, lds, wave);
    if (!alias) dma_tile<BKD>(tB + k, offB, lds + DTB, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#ifdef VMC_GRAM_STAMPS
  unsigned long long seg[4] = {0, 0, 0, 0};
  unsigned long long tprev = stamp();
  const unsigned long long tstart = tprev;
#endif
  for (int64_t s = 0; s < nfull; ++s, k += BKD) {
    if (s + 1 < nfull) {   // the other buffer was last read before the barrier that closed the previous iteration
      // raised priority: with the co-resident waves streaming fp64 MFMAs each of these instructions otherwise waits
      // ~250 cycles for an issue slot (measured with in-kernel stamps; +1 % end to end).  Slotting the pieces between this
      // wave's own MFMA groups instead was 6 % SLOWER.
      __builtin_amdgcn_s_setprio(2);
      dma_tile<BKD>(tA + k + BKD, offA, lds + (cur ^ 1) * 2 * DTB, wave);
      if (!alias) dma_tile<BKD>(tB + k + BKD, offB, lds + (cur ^ 1) * 2 * DTB + DTB, wave);
      __builtin_amdgcn_s_setprio(0);
    }
    VMC_STAMP(0)
    if constexpr (F32) {
      if (active) {
        mma_kstep_swz_f32<BKD>(acc32, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
        if ((s & (F32_BLOCK_STEPS - 1)) == F32_BLOCK_STEPS - 1) flush_f32_blocks(acc, acc32);
      }
    } else if constexpr (MODE == 2) {
      if (active) mma_kstep_swz_w32<BKD>(acc, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
    } else {
      if (active) mma_kstep_swz<BKD>(acc, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
    }
    VMC_STAMP(1)
#ifndef VMC_GRAM_DIAG_NO_WAIT        // timing diagnostics only (results are wrong): what the DMA wait / the K-step barrier cost
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    VMC_STAMP(2)
#ifndef VMC_GRAM_DIAG_NO_BARRIER
    __syncthreads();
#endif
    VMC_STAMP(3)
    cur ^= 1;
  }
#ifdef VMC_GRAM_STAMPS
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    unsigned long long* o = g_stamps + ((size_t)(blockIdx.x % 4096) * 4 + w) * 8;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3]; o[4] = (unsigned long long)nfull; o[5] = tprev - tstart; o[6] = 0; o[7] = 0;
  }
#endif
  if (tail > 0) {   // ragged last step (N not a multiple of BKD): masked loads through registers into the same swizzled image
    constexpr int PER_THREAD = TILE * G::PAIRS / 256;      // pairs per thread (4 or 2)
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
      const int idx = t + 256 * q, row = idx / G::PAIRS, pr = idx % G::PAIRS;
      const int kk = 2 * pr;
      const double* __restrict__ pa = tA + (int64_t)row * lda + k + kk;
      double2_t va = {(kk < tail) ? pa[0] : 0.0, (kk + 1 < tail) ? pa[1] : 0.0};
      *reinterpret_cast<double2_t*>(lds + cur * 2 * DTB + row * BKD + ((pr ^ G::f(row)) << 1)) = va;
      if (!alias) {
        const double* __restrict__ pb = tB + (int64_t)row * ldb + k + kk;
        double2_t vb = {(kk < tail) ? pb[0] : 0.0, (kk + 1 < tail) ? pb[1] : 0.0};
        *reinterpret_cast<double2_t*>(lds + cur * 2 * DTB + DTB + row * BKD + ((pr ^ G::f(row)) << 1)) = vb;
      }
    }
    __syncthreads();
    if constexpr (F32) {
      if (active) mma_kstep_swz_f32<BKD>(acc32, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
    } else if constexpr (MODE == 2) {
      if (active) mma_kstep_swz_w32<BKD>(acc, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
    } else {
      if (active) mma_kstep_swz<BKD>(acc, lds, cur * 2 * DTB, cur * 2 * DTB + boff, wr, wc, koff);
    }
    __syncthreads();
  }
  if constexpr (F32) flush_f32_blocks(acc, acc32);
}

// fp32 C/D map: element (i, j, reg r) of this lane sits at row 64*wr + 16*i + 4*(lane>>4) + r, col 64*wc + 16*j + (lane&15).
__device__ __forceinline__ void tile_store_f32map(const double4_t (&acc)[4][4], double* __restrict__ out /* [128][128] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 64 * wr + 16 * i + 4 * (lane >> 4) + r;
        const int col = 64 * wc + 16 * j + (lane & 15);
        out[row * TILE + col] = acc[i][j][r];
      }
}

// C/D element (i, j, reg r) of this lane sits at row 64*wr + 16*i + (lane>>4) + 4*r, col 64*wc + 16*j + (lane&15).
__device__ __forceinline__ void tile_store(const double4_t (&acc)[4][4], double* __restrict__ out /* [128][128] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 64 * wr + 16 * i + (lane >> 4) + 4 * r;
        const int col = 64 * wc + 16 * j + (lane & 15);
        out[row * TILE + col] = acc[i][j][r];
      }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace vmcpde
