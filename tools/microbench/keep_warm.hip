// Diagnostic (NOT part of the library): a flag-stopped, compute-only "keep-warm" kernel for the experiment in
// tools/diag/keep_warm_ab.sh -- does keeping the chip's power state up during the solve's dependent chains make the next S-build faster?
// (tools/diag/gram_between_phases_probe.py: 8 ms of idling between two Grams cost the next one 1.8 ms.)
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/keep_warm.hip -o tools/microbench/libkeepwarm.so
#include <hip/hip_runtime.h>
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// every workgroup: bursts of work separated by s_sleep, until *flag != 0 or max_iters bursts (an exit every wave reaches)
__global__ __launch_bounds__(256) void keep_warm_kernel(int mode, int burst, int sleep16, int max_iters, const int* __restrict__ flag,
                                                        float* __restrict__ sink) {
  __builtin_amdgcn_s_setprio(0);
  float16_t acc = {0};
  float c0 = 1e-3f * threadIdx.x, c1 = 0.5f, c2 = 0.25f, c3 = 0.125f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1e-3f * (threadIdx.x + i)); b[i] = (__bf16)(1e-3f * (i + 1)); }
  for (int it = 0; it < max_iters; ++it) {
    if (mode == 0) {
      for (int k = 0; k < burst; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
      for (int k = 0; k < burst * 8; ++k) { c0 = fmaf(c0, 1.0001f, 1e-7f); c1 = fmaf(c1, 1.0001f, 1e-7f); c2 = fmaf(c2, 1.0001f, 1e-7f); c3 = fmaf(c3, 1.0001f, 1e-7f); }
    }
    for (int s = 0; s < sleep16; ++s) __builtin_amdgcn_s_sleep(16);
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
  }
  float s = c0 + c1 + c2 + c3;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456f) sink[0] = s;
}

__global__ void set_flag_kernel(int* flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

extern "C" int kw_launch(int mode, int burst, int sleep16, int max_iters, int nwg, int threads, const int* flag, float* sink, void* stream) {
  hipLaunchKernelGGL(keep_warm_kernel, dim3(nwg), dim3(threads), 0, (hipStream_t)stream, mode, burst, sleep16, max_iters, flag, sink);
  return (int)hipGetLastError();
}
extern "C" int kw_set(int* flag, int v, void* stream) {
  hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, v);
  return (int)hipGetLastError();
}
