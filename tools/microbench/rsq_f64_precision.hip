// Microbenchmark (not part of the library): how accurate is v_rsq_f64 on gfx950?  (The Cholesky sweep refines it with two coupled
// Goldschmidt steps and a residual correction: eight dependent fp64 operations on the critical chain of every column.)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/rsq_f64_precision.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_rsq(x[i]);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> hx(n), hy(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;       // [0, 1)
    hx[i] = std::ldexp(1.0 + u, (int)(s % 41) - 20);                // mantissas uniform, exponents -20 .. 20
  }
  double *dx, *dy;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
  hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(hy.data(), dy, n * 8, hipMemcpyDeviceToHost);
  long double worst = 0;
  for (int i = 0; i < n; ++i) {
    const long double ref = 1.0L / sqrtl((long double)hx[i]);
    const long double rel = fabsl(((long double)hy[i] - ref) / ref);
    if (rel > worst) worst = rel;
  }
  printf("v_rsq_f64: largest relative error over %d inputs: %.3Le = 2^%.2Lf\n", n, worst, log2l(worst));
  return 0;
}
