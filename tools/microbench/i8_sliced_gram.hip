// Microbenchmark (not part of the library; round-3 verdict item 6): can an ERROR-FREE sliced Gram on the INT8 matrix pipe beat the fp64
// MFMA syrk?  J[p][n] = 2^e_p * sum_s d_s[p][n] 2^(-6-7s) with int8 slices |d_s| <= 64 (per-row power-of-two scale, round to nearest
// residuals), S = J J^T = 2^(e_p+e_q) sum_k 2^(-12-7k) T_k,  T_k = sum_{s+t=k} D_s D_t^T  accumulated EXACTLY in int32 (K <= 2^16 per
// accumulator with |d| <= 64 and at most 8 pairs per diagonal).  With NS = 7 slices and the 28 pairs s+t <= 6 the truncation is
// ~2^-46 of (row max)^2 N -- below 1e-12 of |S|_F for rows whose max/rms is a few units.
//
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/i8_sliced_gram.hip -o tools/microbench/i8_sliced_gram
//   tools/microbench/i8_sliced_gram rate      what the planned inner loop sustains with LDS-resident operands (no global traffic):
//                                             wave tile 64 x 32, 7 slices, 7 int32 diagonals (224 accumulator VGPRs), 56 MFMAs of
//                                             v_mfma_i32_32x32x32_i8 per 32-sample step, 21 KB of fragment reads per wave and step
//   tools/microbench/i8_sliced_gram slice     the slicing pass alone (row max + 7 int8 planes) at 4096 x 65536
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16_t_ __attribute__((ext_vector_type(16)));
constexpr int NS = 7;          // slices
constexpr int KS = 32;         // samples per MFMA step (bytes per fragment row)

// ---- (1) rate probe: the inner loop on LDS-resident operands ---------------------------------------------------------------------
// workgroup = 4 waves (one per SIMD: 224 accumulator + 84 operand VGPRs leave no room for a second), tile 128 x 64, wave tile 64 x 32
template <int NPAIRMAX>
__global__ __launch_bounds__(256, 1) void i8_rate_kernel(int steps, int* __restrict__ out) {
  __shared__ int4_t ldsA[NS * 128 * 2];      // [slice][row][2 x 16 B]
  __shared__ int4_t ldsB[NS * 64 * 2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
  for (int i = t; i < NS * 128 * 2; i += 256) ldsA[i] = (int4_t){(int)(0x01020304u * (i + 1)), (int)(0x11213141u * (i + 3)), i * 77, i ^ 0x5a5a5a5a};
  for (int i = t; i < NS * 64 * 2; i += 256) ldsB[i] = (int4_t){(int)(0x0a0b0c0du * (i + 7)), (int)(0x01010101u * (i + 5)), i * 31, i ^ 0x3c3c3c3c};
  __syncthreads();
  int16_t_ acc[NS][2];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][m][r] = 0;
  const int r32 = lane & 31, h = lane >> 5;
  for (int s_ = 0; s_ < steps; ++s_) {
    int4_t a[NS][2], b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int m = 0; m < 2; ++m) a[s][m] = ldsA[(s * 128 + 64 * wr + 32 * m + r32) * 2 + h];
      b[s] = ldsB[(s * 64 + 32 * wc + r32) * 2 + h];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int tt = 0; tt < NS; ++tt)
        if (s + tt < NPAIRMAX) {
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[s + tt][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s][m], b[tt], acc[s + tt][m], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
  }
  int sum = 0;
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[k][m][r];
  out[blockIdx.x * 256 + t] = sum;
}

// the same number of MFMAs per step from registers only (no LDS reads): the matrix pipe's own ceiling under this accumulator pattern
__global__ __launch_bounds__(256, 1) void i8_regs_kernel(int steps, int* __restrict__ out) {
  const int t = threadIdx.x;
  int16_t_ acc[NS][2];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][m][r] = 0;
  int4_t a[NS][2], b[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    b[s] = (int4_t){t * 3 + s, t ^ 0x55, t * 7, s * 1234567};
#pragma unroll
    for (int m = 0; m < 2; ++m) a[s][m] = (int4_t){t + s, t * 5 + m, t ^ 0x0f0f0f0f, s * 7654321 + m};
  }
  for (int s_ = 0; s_ < steps; ++s_) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int tt = 0; tt < NS; ++tt)
        if (s + tt < NS) {
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[s + tt][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s][m], b[tt], acc[s + tt][m], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
  }
  int sum = 0;
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[k][m][r];
  out[blockIdx.x * 256 + t] = sum;
}

// ---- (2) the slicing pass ---------------------------------------------------------------------------------------------------------
// one workgroup per parameter row: pass 1 row maximum -> exponent; pass 2 seven int8 planes [slice][row][n]
__global__ __launch_bounds__(256) void slice_rows_kernel(const double* __restrict__ JT, int64_t ldj, int64_t N, int8_t* __restrict__ planes,
                                                         int64_t plane_stride, int* __restrict__ expo) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  const double* __restrict__ row = JT + (int64_t)blockIdx.x * ldj;
  double mx = 0.0;
  for (int64_t n = 2 * t; n < N; n += 512) {
    const double2 v = *reinterpret_cast<const double2*>(row + n);
    mx = fmax(mx, fmax(fabs(v.x), fabs(v.y)));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  int e = 0;
  frexp(mx, &e);                         // mx = f 2^e, f in [0.5, 1): |x| / 2^e < 1
  if (t == 0) expo[blockIdx.x] = e;
  const double inv = ldexp(1.0, 6 - e);  // first slice carries 6 bits + sign (|d_0| <= 64), the others 7 (|d_s| <= 64 after rounding)
  int8_t* __restrict__ prow = planes + (int64_t)blockIdx.x * ldj;
  for (int64_t n = 8 * t; n < N; n += 2048) {
    double x[8];
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      const double2 v = *reinterpret_cast<const double2*>(row + n + q);
      x[q] = v.x * inv;
      x[q + 1] = v.y * inv;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      union { int8_t b[8]; int2 w; } u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double d = rint(x[q]);
        u.b[q] = (int8_t)(int)d;
        x[q] = (x[q] - d) * 128.0;
      }
      *reinterpret_cast<int2*>(prow + s * plane_stride + n) = u.w;
    }
  }
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "rate";
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  if (!strcmp(mode, "rate")) {
    int* out;
    CHECK(hipMalloc(&out, 256 * 256 * sizeof(int)));
    const int steps = argc > 2 ? atoi(argv[2]) : 20000;
    for (int variant = 0; variant < 3; ++variant) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(i8_regs_kernel, dim3(256), dim3(256), 0, 0, steps, out);
        else if (variant == 1) hipLaunchKernelGGL(i8_rate_kernel<NS>, dim3(256), dim3(256), 0, 0, steps, out);
        else hipLaunchKernelGGL(i8_rate_kernel<1>, dim3(256), dim3(256), 0, 0, steps, out);        // fragment reads only + 2 MFMAs
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double mfmas = (variant == 2 ? 2.0 : 56.0) * 1024.0 * steps;
        const double ops = mfmas * 2.0 * 32 * 32 * 32;
        printf("%s: %.3f ms for %d steps -> %.3f POP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
               variant == 0 ? "registers only, 56 MFMA/step" : variant == 1 ? "LDS fragments + 56 MFMA/step " : "LDS fragments + 2 MFMA/step  ",
               ms, steps, ops / (ms * 1e-3) / 1e15, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
      }
    }
    // what that means for the Gram of the headline problem
    printf("headline Gram: 28 pairs x N P (P+1) = %.3e int8 op; at the rates above this is the matrix-pipe phase alone\n", 28.0 * 65536.0 * 4213.0 * 4214.0);
    return 0;
  }
  if (!strcmp(mode, "slice")) {
    const int64_t P = 4096, N = 65536;
    double* J;
    int8_t* planes;
    int* expo;
    CHECK(hipMalloc(&J, P * N * sizeof(double)));
    CHECK(hipMalloc(&planes, (size_t)NS * P * N));
    CHECK(hipMalloc(&expo, P * sizeof(int)));
    std::vector<double> h(N);
    srand(7);
    for (int64_t p = 0; p < P; p += 512) {      // a few distinct rows, repeated (the timing does not depend on the values)
      for (int64_t n = 0; n < N; ++n) h[n] = (rand() / (double)RAND_MAX - 0.5) * (1.0 + (p % 7));
      for (int64_t q = p; q < p + 512; ++q) CHECK(hipMemcpy(J + q * N, h.data(), N * sizeof(double), hipMemcpyHostToDevice));
    }
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)P), dim3(256), 0, 0, J, N, N, planes, P * N, expo);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("slicing %lld x %lld into %d int8 planes: %.3f ms (reads %.2f GB twice, writes %.2f GB)\n", (long long)P, (long long)N, NS, ms,
             P * N * 8e-9, NS * P * N * 1e-9);
    }
    // exactness of the slices on the host for one row
    std::vector<int8_t> hp((size_t)NS * N);
    std::vector<double> hr(N);
    int e = 0;
    CHECK(hipMemcpy(hr.data(), J + 5 * N, N * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&e, expo + 5, sizeof(int), hipMemcpyDeviceToHost));
    for (int s = 0; s < NS; ++s) CHECK(hipMemcpy(hp.data() + (size_t)s * N, planes + (size_t)s * P * N + 5 * N, N, hipMemcpyDeviceToHost));
    double worst = 0, mx = 0;
    int dmax = 0;
    for (int64_t n = 0; n < N; ++n) {
      double rec = 0;
      for (int s = 0; s < NS; ++s) { rec += ldexp((double)hp[(size_t)s * N + n], e - 6 - 7 * s); dmax = abs(hp[(size_t)s * N + n]) > dmax ? abs(hp[(size_t)s * N + n]) : dmax; }
      worst = fmax(worst, fabs(rec - hr[n]));
      mx = fmax(mx, fabs(hr[n]));
    }
    printf("row 5: exponent %d, max |slice| %d, reconstruction error %.3e of row max %.3e (2^-%d)\n", e, dmax, worst, mx, (int)(-log2(worst / mx)));
    return 0;
  }
  fprintf(stderr, "usage: %s rate [steps] | slice\n", argv[0]);
  return 1;
}
