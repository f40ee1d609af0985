// Latent-space samplers: exact Gauss / Student-t sampling and the independence-Metropolis MCMC loop.
// Reference: sampler.py:20-45 (cos_dist target, gauss_sample, student_t_sample, radial_update_prop),
// sampler.py:72-111 (Sampler.__call__, _get_samples).  JAX's threefry streams and the reference's unseeded
// numpy chisquare (sampler.py:32) cannot be reproduced bit-for-bit; the generator here is the
// counter-based Philox4x32-10: draw j of sample n in call `stream_id` is a pure function of
// (seed, n, stream_id, j), so any sharding of the samples over GPUs yields the same global sample set.
#include "common.hpp"

using namespace vmcpde;

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    U4 n;
    n.x = hi1 ^ ctr.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ ctr.w ^ k1;
    n.w = lo0;
    ctr = n;
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

__device__ __forceinline__ double u01(uint32_t a, uint32_t b) {  // 53-bit uniform in (0,1)
  const uint64_t v = ((uint64_t)a << 21) ^ ((uint64_t)b >> 11);
  return ((double)(v & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
}

struct Rng {
  uint64_t seed, n;
  uint32_t stream, sub;
  __device__ void two_uniforms(double& a, double& b) {
    U4 c{(uint32_t)n, (uint32_t)(n >> 32), stream, sub++};
    U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    a = u01(r.x, r.y);
    b = u01(r.z, r.w);
  }
  __device__ void two_normals(double& a, double& b) {  // Box-Muller
    double u1, u2;
    two_uniforms(u1, u2);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    a = rad * c;
    b = rad * s;
  }
};

// Gamma(shape a, 1) by Marsaglia-Tsang (chi^2_nu = 2 Gamma(nu/2); reference: np.random.chisquare, sampler.py:32)
__device__ double gamma_sample(Rng& rng, double a) {
  double boost = 1.0;
  if (a < 1.0) {
    double u, dummy;
    rng.two_uniforms(u, dummy);
    boost = pow(u, 1.0 / a);
    a += 1.0;
  }
  const double dd = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
  for (int it = 0; it < 64; ++it) {   // acceptance > 95 % per round; 64 rounds never exhaust in practice
    double x, y, u, u2;
    rng.two_normals(x, y);
    rng.two_uniforms(u, u2);
    double v = 1.0 + cc * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    if (log(u) < 0.5 * x * x + dd - dd * v + dd * log(v)) return boost * dd * v;
  }
  return boost * dd;
}

__global__ void sample_latent_kernel(const FlowProgram* __restrict__ fp, const LatentCache* __restrict__ lc, uint64_t seed,
                                     uint32_t stream_id, int64_t first, int64_t N, double* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int d = fp->dim;
  Rng rng{seed, (uint64_t)(first + i), stream_id, 0};
  double nrm[MAX_DIM];
#pragma unroll
  for (int k = 0; k < MAX_DIM; k += 2) {
    if (k < d) rng.two_normals(nrm[k], nrm[k + 1]);
  }
  double scale = 1.0;
  if (fp->latent == VMCPDE_LATENT_STUDENT_T) {   // sampler.py:31-34
    rng.sub = 64;
    const double chi2 = 2.0 * gamma_sample(rng, 0.5 * lc->nu);
    scale = sqrt(lc->nu / chi2);
  }
#pragma unroll
  for (int r = 0; r < MAX_DIM; ++r) {
    if (r < d) {
      double y = 0.0;
#pragma unroll
      for (int j = 0; j < MAX_DIM; ++j)
        if (j <= r) y += lc->C[r * MAX_DIM + j] * nrm[j];
      z[i * d + r] = scale * y + lc->mu[r] + fp->offset[r];   // sampler.py:26,34,86
    }
  }
}

__device__ __forceinline__ double cos_dist(double x, double y, double ox, double oy) {  // sampler.py:20-22
  const double r = fmin(1.0, 4.0 * sqrt((x - ox) * (x - ox) + (y - oy) * (y - oy)));
  return log(0.5 * (1.0 + cospi(r)));
}

// sampler.py:88-111: one chain per thread, `steps` independence-Metropolis updates, every state emitted.
template <bool INJECT>
__global__ void mcmc_kernel(uint64_t seed, uint32_t stream_id, const double* __restrict__ u_r, const double* __restrict__ u_phi,
                            const double* __restrict__ u_acc, int chains, int steps, double bound, double ox, double oy,
                            double* __restrict__ states, double* __restrict__ out, unsigned long long* __restrict__ n_acc) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= chains) return;
  double sx = states[2 * ch], sy = states[2 * ch + 1];
  double lp_old = cos_dist(sx, sy, ox, oy);
  unsigned long long acc = 0;
  for (int s = 0; s < steps; ++s) {
    double ur, uphi, uacc, dummy;
    if (INJECT) {
      ur = u_r[(int64_t)s * chains + ch];
      uphi = u_phi[(int64_t)s * chains + ch];
      uacc = u_acc[(int64_t)s * chains + ch];
    } else {
      Rng rng{seed, (uint64_t)ch, stream_id, (uint32_t)(2 * s)};
      rng.two_uniforms(ur, uphi);
      rng.two_uniforms(uacc, dummy);
    }
    const double rad = sqrt(ur) * bound;            // sampler.py:39-40
    double sn, cs;
    sincospi(2.0 * uphi, &sn, &cs);                 // sampler.py:41-43
    const double nx = rad * cs + ox, ny = rad * sn + oy;
    const double lp_new = cos_dist(nx, ny, ox, oy);
    const double P = exp(lp_new - lp_old);          // sampler.py:95
    if (uacc < P) {                                 // bernoulli(P), sampler.py:96,102-106
      sx = nx; sy = ny; lp_old = lp_new; ++acc;
    }
    out[((int64_t)s * chains + ch) * 2] = sx;
    out[((int64_t)s * chains + ch) * 2 + 1] = sy;
  }
  states[2 * ch] = sx;
  states[2 * ch + 1] = sy;
  if (n_acc) atomicAdd(n_acc, acc);
}

}  // namespace

// defined in flow.hip
int vmcpde_prepare_latent_internal(vmcpde_model* m, const double* theta, hipStream_t st);

extern "C" int vmcpde_sample_latent(vmcpde_model* m, const double* theta, uint64_t seed, uint64_t stream_id,
                                    int64_t first_sample, int64_t n_samples, double* z, void* stream) {
  VMC_REQUIRE(m && theta && z, "vmcpde_sample_latent: null argument");
  VMC_REQUIRE(stream_id < (1ull << 32), "stream_id must fit 32 bits");
  if (n_samples <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  const int64_t grid = (n_samples + 255) / 256;
  hipLaunchKernelGGL(sample_latent_kernel, dim3((unsigned)grid), dim3(256), 0, st, m->dev, m->cache, seed, (uint32_t)stream_id,
                     first_sample, n_samples, z);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

static int mcmc_common(bool inject, uint64_t seed, uint64_t stream_id, const double* u_r, const double* u_phi, const double* u_acc,
                       int32_t chains, int32_t steps, double bound, const double* offset2, double* states, double* out,
                       unsigned long long* n_acc, void* stream) {
  VMC_REQUIRE(offset2 && states && out, "vmcpde_mcmc: null argument");
  VMC_REQUIRE(chains >= 1 && steps >= 0, "bad chains/steps");
  VMC_REQUIRE(stream_id < (1ull << 32), "stream_id must fit 32 bits");
  if (steps == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (n_acc) VMC_CHECK_HIP(hipMemsetAsync(n_acc, 0, sizeof(unsigned long long), st));
  const int grid = (chains + 63) / 64;
  if (inject)
    hipLaunchKernelGGL(mcmc_kernel<true>, dim3(grid), dim3(64), 0, st, seed, (uint32_t)stream_id, u_r, u_phi, u_acc, chains, steps,
                       bound, offset2[0], offset2[1], states, out, n_acc);
  else
    hipLaunchKernelGGL(mcmc_kernel<false>, dim3(grid), dim3(64), 0, st, seed, (uint32_t)stream_id, u_r, u_phi, u_acc, chains, steps,
                       bound, offset2[0], offset2[1], states, out, n_acc);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_mcmc(uint64_t seed, uint64_t stream_id, int32_t chains, int32_t steps, double bound, const double* offset2,
                           double* states, double* out, unsigned long long* n_acc, void* stream) {
  return mcmc_common(false, seed, stream_id, nullptr, nullptr, nullptr, chains, steps, bound, offset2, states, out, n_acc, stream);
}

extern "C" int vmcpde_mcmc_from_uniforms(const double* u_r, const double* u_phi, const double* u_acc, int32_t chains, int32_t steps,
                                         double bound, const double* offset2, double* states, double* out,
                                         unsigned long long* n_acc, void* stream) {
  VMC_REQUIRE(u_r && u_phi && u_acc, "vmcpde_mcmc_from_uniforms: null uniforms");
  return mcmc_common(true, 0, 0, u_r, u_phi, u_acc, chains, steps, bound, offset2, states, out, n_acc, stream);
}
