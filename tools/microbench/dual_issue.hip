// Microbenchmark (not part of the library): can fp64 VALU FMAs run next to fp64 MFMAs on gfx950, and what does the chip's
// power management make of it?  Peak rates are equal on paper (MI355X: 78.6 TFLOP/s vector and matrix fp64).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/dual_issue.hip -o tools/microbench/dual_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

// mode bit 0: waves 0..3 of the workgroup run MFMA chains; bit 1: waves 4..7 run VALU FMA chains
__global__ __launch_bounds__(512) void dual_kernel(int iters, int mode, double* __restrict__ out) {
  const int wave = threadIdx.x >> 6;
  double s = 0.0;
  if (wave < 4) {
    if (mode & 1) {
      double4_t acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
      const double a = 1e-3 * (1 + (threadIdx.x & 63)), b = 1e-3 * (16 - (threadIdx.x & 15));
      for (int it = 0; it < iters; it += 16) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
  } else if (mode & 2) {
    // same flop count per wave as an MFMA wave: one MFMA = 2048 flop = 16 wave-wide FMAs
    double c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 1e-3 * j;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12;
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
      for (int rep = 0; rep < 16 * 8; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = fma(c[j], a, b);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += c[j];
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000, blocks = 512;
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(dual_kernel, dim3(blocks), dim3(512), 0, 0, iters, mode, out);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop_m = (mode & 1) ? (double)blocks * 4 * iters * 8 * 2048.0 : 0.0;
      const double flop_v = (mode & 2) ? (double)blocks * 4 * iters * 8 * 16 * 128.0 : 0.0;
      printf("mode %d (%s%s): %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", mode, (mode & 1) ? "mfma " : "", (mode & 2) ? "valu" : "", ms,
             flop_m / ms * 1e-9, flop_v / ms * 1e-9, (flop_m + flop_v) / ms * 1e-9);
    }
  }
  return 0;
}
