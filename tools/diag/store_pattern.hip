// What the Jacobian's store pattern alone costs: N / 64 workgroups, each writing P row segments of 512 B (64 samples of one parameter row)
// at a row stride of N doubles -- flow_eval_kernel's HBM traffic without its arithmetic -- against a streaming fill of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/diag/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// waves of a workgroup take the rows round robin, `burst` rows per turn (the kernel stores blocks of up to six consecutive rows)
__global__ __launch_bounds__(320) void pattern_kernel(double* __restrict__ JT, long ldj, int P, int waves, int burst, int order) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 64 + lane;
  const double v = (double)n;
  for (int r0 = wave * burst; r0 < P; r0 += waves * burst)
    for (int b = 0; b < burst && r0 + b < P; ++b) {
      const int row = order ? (P - 1 - (r0 + b)) : (r0 + b);
      JT[(long)row * ldj + n] = v + row;
    }
}

__global__ void fill_kernel(double* __restrict__ JT, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) JT[i] = (double)i;
}

int main() {
  const int P = 4213;
  const long N = 65536, total = (long)P * N;
  double* JT;
  CK(hipMalloc(&JT, total * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-58s %.3f ms  %.2f TB/s\n", name, best, total * 8.0 / best / 1e9);
  };
  run("streaming fill (2048 x 256 threads, grid stride)", [&] { hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, JT, total); });
  for (int waves : {1, 4, 5}) for (int burst : {1, 6, 21}) {
    char name[128];
    snprintf(name, sizeof name, "512-B row segments, %d wave(s) per workgroup, bursts of %d rows", waves, burst);
    run(name, [&] { hipLaunchKernelGGL(pattern_kernel, dim3(N / 64), dim3(64 * waves), 0, 0, JT, N, P, waves, burst, 0); });
  }
  run("512-B row segments, 5 waves, bursts of 6, rows descending", [&] { hipLaunchKernelGGL(pattern_kernel, dim3(N / 64), dim3(320), 0, 0, JT, N, P, 5, 6, 1); });
  return 0;
}
