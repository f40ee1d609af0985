// Coupling-flow kernels: batched INNwProb forward / inverse, per-sample input gradient, directional second
// derivatives and the per-sample parameter Jacobian (stored parameter-major), plus the PDE right-hand sides.
//
// Replaces the jitted closures J1, J3, J4, J5, J6, J14 of the reference:
//   var_state.py:29-34, net.py:44-62 (SingleTrafo), net.py:84-153 (SingleBlock fwd/inv, all four coupling
//   variants), net.py:168-182 (INN), net.py:209-217 (INNwProb), net.py:10-36 (Gauss / Student_t),
//   util.py:21-26 (covariance factor), evolutionEq.py:23-45, 84-119.
//
// Execution model (gfx950): ONE SAMPLE PER LANE, and up to eight waves per workgroup that COOPERATE on the same 64 samples.
// Every per-sample vector (state, MLP activations, tangents, adjoints) lives in LDS as [unit][lane] (lane-contiguous
// doubles, so each ds_read_b64/ds_write_b64 is conflict-free) and is shared by the waves of the workgroup; MLP weights
// are wave-uniform and are read through the scalar cache (s_load) straight from the flat parameter vector; the
// output-blocked (JB) accumulators live in VGPRs.  The dense layers -- all the O(width^2) work -- are split between
// the waves (forward: by output block; reverse: by input unit, which also splits the Jacobian rows), the O(d) coupling
// algebra is done by wave 0, and workgroup barriers separate the stages.  The per-lane LDS footprint (75 KB per
// 64 samples at hidden (21,21), 140 KB at (43,43)) allows only one or two such sample groups per CU, so without the
// split three of the four SIMDs would idle and every LDS / scalar-load latency would be exposed.
// The Jacobian is written as JT[p][n]: for a fixed parameter p the 64 lanes store 64 consecutive doubles
// (one 512-B coalesced store per parameter per wave) -- no transposition pass is needed, and the Gram kernel
// consumes exactly this K-contiguous layout.
//
// Derivatives are hand-derived (the oracle uses autodiff, so the two are independent):
//   * second derivatives: forward Taylor mode -- (value, d/de, d2/de2) along x + e v is pushed through every
//     layer; v^T H v = d2/de2 logp.  One pass per direction.
//   * grad_x and grad_theta: reverse sweep from the latent end, walking the flow backwards with the exact
//     inverse of each coupling half-step so that only one MLP's activations are alive at a time.
#include "common.hpp"

using namespace vmcpde;

namespace {

#ifndef VMC_FLOW_MAX_WAVES
#define VMC_FLOW_MAX_WAVES 8
#endif
#ifndef VMC_FLOW_UNROLL
#define VMC_FLOW_UNROLL 1            // input units per trip of the dense-layer loops.  Measured (tools/diag/flow_unroll_ab.sh): unrolling by 2 / 3,
                                     // i.e. the LDS reads of several units in flight together, is 3-5 % SLOWER on C2 / C3 / C4
#endif
#ifndef VMC_FLOW_EPS_REGS
#define VMC_FLOW_EPS_REGS 1          // reverse sweep: eps of a wave's input units in registers (0: an LDS line, the round-2 form)
#endif
// JB, the output block a wave holds in registers in the dense-layer loops, is a constant of the namespaces the device code is compiled in
// (below): 6 x up-to-8 waves is the best compromise over C2 / C3 (measured), 5 where it gives the same number of waves (pick_jb)
constexpr int MAX_WAVES = VMC_FLOW_MAX_WAVES;

struct Regions {
  // persistent
  int X;    // state, 3 components of stride d (value, first, second Taylor coefficient)
  int G;    // adjoint of the state (d)
  // Taylor phase
  int A, B; // ping-pong MLP buffers, 3 components of stride cs
  int SOUT; // stash of s (3 x d) while t is evaluated (different_add)
  int LAT;  // latent residual / whitened residual (3 x d)
  // reverse phase (aliases the Taylor phase)
  int ACT;  // inputs of every layer of the current MLP (n_in + sum hidden)
  int OUTS, OUTT;  // s and t outputs (d each)
  int DW;   // cotangent of the MLP outputs (d)
  int DEL0, DEL1, EPS;  // delta ping-pong and W*delta accumulator (cs each)
  int GAM;  // input gradient of the MLPs (d)
  int Y, Q; // whitened residual y = U^-1 r and q = U^-T y (d each)
  int cs;   // component stride of A/B = widest layer
  int inplace;  // Taylor / value MLP passes write a layer's outputs over its inputs (one output block per wave: registers hold
                // the block across the barrier that separates the last read from the first write) -- no second buffer B
  int total;
  int wgt;  // doubles of the workgroup-shared weight stage behind the per-lane vectors (kernels + biases of ALL layers of one MLP)
};

// values-only passes (flow_sample): state, two single-component MLP buffers, the s stash and the latent line
Regions make_regions_values(const FlowProgram& fp) {
  Regions r = {};
  const int d = fp.dim, W = fp.max_width;
  r.cs = W;
  int u = 0;
  r.X = u; u += d;
  r.A = u; u += W;
  r.B = u; u += W;
  r.SOUT = u; u += d;
  r.LAT = u; u += d;
  r.inplace = 0;
  r.total = u;
  r.wgt = fp.stage_all ? fp.all_mlp_weights : fp.max_mlp_weights;
  return r;
}

// kdirs: Hessian directions carried by one forward Taylor pass (1 + 2 kdirs components: value, then first and second derivative
// along each direction)
Regions make_regions(const FlowProgram& fp, int kdirs, int JB) {
  // Round 3 plan: what bounds flow_eval is how many workgroups are resident per CU, i.e. this footprint (C2: 146 units = 75 KB per 64
  // samples -> two workgroups per CU; now 82 units = 42 KB -> three).  (i) Taylor phase: in-place layer buffers (no B) whenever
  // every wave owns at most one output block; the latent line aliases the then idle MLP buffer; the s stash exists only for
  // different_add.  (ii) reverse phase: the cotangent of a layer's inputs overwrites those inputs in ACT (its owner is the only
  // reader of both), so the delta ping-pong is gone; Y / Q of the latent adjoint alias DEL0 / GAM, which are written later.
  Regions r;
  const int d = fp.dim, W = fp.max_width, nc = 1 + 2 * kdirs;
  const bool diff_add = (fp.coupling == VMCPDE_COUPLING_DIFFERENT_ADD);
  r.cs = W;
  r.inplace = ((W + JB - 1) / JB <= MAX_WAVES && env_int("VMCPDE_FLOW_INPLACE", 1)) ? 1 : 0;
  int u = 0;
  r.X = u; u += nc * d;
  r.G = u; u += d;
  const int base = u;
  // Taylor phase
  r.A = u; u += nc * W;
  if (r.inplace) r.B = r.A;
  else { r.B = u; u += nc * W; }
  r.SOUT = u; if (diff_add) u += nc * d;
  if (W >= d) r.LAT = r.A;                 // (latent_forward runs after the last MLP pass)
  else { r.LAT = u; u += nc * d; }
  const int end_taylor = u;
  // reverse phase
  u = base;
  r.ACT = u; u += fp.max_sum_act;
  r.OUTS = u; u += d;
  r.OUTT = u; if (diff_add) u += d;
  r.DW = u; u += d;
  r.DEL0 = u; u += d;                      // cotangent of the MLP's output pre-activations (n_out <= d)
  r.DEL1 = r.DEL0;                         // (unused since round 3: deltas of hidden layers live in ACT)
  r.EPS = u; if (!(VMC_FLOW_EPS_REGS && r.inplace)) u += W;       // (inplace plans keep eps in registers, see mlp_backward)
  r.GAM = u; u += d;
  r.Y = r.DEL0;
  r.Q = r.GAM;
  r.total = (u > end_taylor) ? u : end_taylor;
  r.wgt = fp.stage_all ? fp.all_mlp_weights : fp.max_mlp_weights;
  return r;
}

#define LD(u) (c.lds[(u) * c.lanes + c.lane])
// The waves of a workgroup talk to each other through LDS only (per-sample vectors, the weight stage); what they write to global memory -- the
// Jacobian rows, logp, g, hv -- is never read back inside the kernel.  __syncthreads() is a workgroup-scope release + acquire around s_barrier,
// i.e. it also waits for every outstanding GLOBAL store of the wave.  -DVMC_FLOW_LDS_BARRIER=1 waits for the wave's LDS traffic only, then
// s_barrier -- MEASURED NEUTRAL in round 6 (Eloc + J at C2 0.617 against 0.612 ms, C4 0.511 / 0.505, C3 2.22 / 2.19: profiles/r06_flow_barrier_ab.txt):
// the kernel is not waiting for its stores at the barriers.  Kept as an A/B knob; the default is the plain barrier.
#ifndef VMC_FLOW_LDS_BARRIER
#define VMC_FLOW_LDS_BARRIER 0
#endif
#ifdef VMC_FLOW_DIAG_NO_SYNC       // timing diagnostics only (results are wrong): what the workgroup barriers cost
#define SYNC() ((void)0)
#elif VMC_FLOW_LDS_BARRIER
#define SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define SYNC() __syncthreads()
#endif
#define W0 (c.wave == 0)

// -DVMC_FLOW_STAMPS (diagnostic builds, tools/diag/flow_stamps.py): wave 0 of every workgroup accumulates the shader-clock cycles it
// spends in each phase of flow_eval_kernel; slots: 0 forward gather + barrier, 1 forward MLPs, 2 forward coupling algebra,
// 3 latent forward, 4 latent backward, 5 reverse: activations again (forward-store), 6 reverse coupling algebra,
// 7 reverse MLP backward (+ Jacobian stores), 8 rest, 9 workgroups counted
#ifdef VMC_FLOW_STAMPS
struct FlowStamps { unsigned long long last, acc[10]; };
__device__ unsigned long long g_flow_stamps[10];
#define FSTAMP(i) do { if (c.st) { const unsigned long long now_ = __builtin_readcyclecounter(); c.st->acc[i] += now_ - c.st->last; c.st->last = now_; } } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

// The JB-dependent device code once per output block width (flow_kernels.inc): 6 (C2, C3: widths 21, 43) and 5 (width 19: blocks
// 5, 5, 5, 4 instead of 6, 6, 6, 1 -- Eloc + J of the C4 shard 0.57 -> 0.50 ms); -DVMC_FLOW_JB=n (A/B builds) compiles that width alone.
#ifdef VMC_FLOW_JB
namespace jbA { constexpr int JB = VMC_FLOW_JB;
#include "flow_kernels.inc"
}
namespace jbB = jbA;
#define VMC_FLOW_FAST_PATH 1
namespace jbAf { constexpr int JB = VMC_FLOW_JB;
#include "flow_kernels.inc"
}
#undef VMC_FLOW_FAST_PATH
namespace jbBf = jbAf;
constexpr int JB_A = VMC_FLOW_JB, JB_B = VMC_FLOW_JB;
#else
namespace jbA { constexpr int JB = 6;
#include "flow_kernels.inc"
}
namespace jbB { constexpr int JB = 5;
#include "flow_kernels.inc"
}
// the same device code with the model's switches fixed to the common model (no_add, Gauss, fp64 rows, no global_change): see flow_kernels.inc
#define VMC_FLOW_FAST_PATH 1
namespace jbAf { constexpr int JB = 6;
#include "flow_kernels.inc"
}
namespace jbBf { constexpr int JB = 5;
#include "flow_kernels.inc"
}
#undef VMC_FLOW_FAST_PATH
constexpr int JB_A = 6, JB_B = 5;
#endif

// the common model: every run-time switch of the device code at the value the fast namespaces fix (VMCPDE_FLOW_FAST=0: the general code, for an A/B)
bool flow_fast_path(const FlowProgram& fp, int jt_f32) {
  return fp.coupling == VMCPDE_COUPLING_NO_ADD && fp.latent == VMCPDE_LATENT_GAUSS && !fp.global_change && !jt_f32 && env_int("VMCPDE_FLOW_FAST", 1) != 0;
}

#ifdef VMC_FLOW_STAMPS
extern "C" int vmcpde_debug_read_flow_stamps(unsigned long long* out_host, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_flow_stamps), sizeof(unsigned long long) * 10);
  if (reset) {
    unsigned long long z[10] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_flow_stamps), z, sizeof(z));
  }
  return rc;
}
#endif

__device__ double digamma(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  // asymptotic series: ln x - 1/2x - sum B_2k / (2k x^2k)
  const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0
                   + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

// util.py:21-26 + net.py:18-20, 32-36: everything that depends on theta but not on the sample.
__global__ void latent_prepare_kernel(const FlowProgram* __restrict__ fp, const double* __restrict__ th, LatentCache* __restrict__ lc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int d = fp->dim;
  int k = 0;
  double logdet = 0.0;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) lc->U[i * MAX_DIM + j] = 0.0;
  for (int i = 0; i < d; ++i) {
    for (int j = i + 1; j < d; ++j, ++k) lc->U[i * MAX_DIM + j] = th[fp->off_L + k];
    lc->U[i * MAX_DIM + i] = exp(th[fp->off_Ldiag + i]);
    logdet += 2.0 * th[fp->off_Ldiag + i];
    lc->mu[i] = th[fp->off_mu + i];
  }
  // S = U U^T, C = lower Cholesky factor of S (what jax.random.multivariate_normal(method='cholesky') applies, sampler.py:26)
  for (int i = 0; i < d; ++i) {
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int m = 0; m < d; ++m) s += lc->U[i * MAX_DIM + m] * lc->U[j * MAX_DIM + m];
      for (int m = 0; m < j; ++m) s -= lc->C[i * MAX_DIM + m] * lc->C[j * MAX_DIM + m];
      lc->C[i * MAX_DIM + j] = (i == j) ? sqrt(s) : s / lc->C[j * MAX_DIM + j];
    }
    for (int j = i + 1; j < d; ++j) lc->C[i * MAX_DIM + j] = 0.0;
  }
  lc->logdetS = logdet;
  const double p = (double)d;
  if (fp->latent == VMCPDE_LATENT_GAUSS) {
    lc->nu = 0.0;
    lc->lognorm = -0.5 * (p * log(2.0 * M_PI) + logdet);
    lc->dlognorm_dtau = 0.0;
  } else {
    const double nu = exp(th[fp->off_dist]) + 1.0;
    lc->nu = nu;
    lc->lognorm = lgamma(0.5 * (nu + p)) - lgamma(0.5 * nu) - 0.5 * p * log(nu * M_PI);
    lc->dlognorm_dtau = (nu - 1.0) * (0.5 * digamma(0.5 * (nu + p)) - 0.5 * digamma(0.5 * nu) - 0.5 * p / nu);
  }
}

// sampler.py:25-34,86 with injected draws: z = mu + C n (* sqrt(nu/chi2)) + offset
__global__ void latent_from_normals_kernel(const FlowProgram* __restrict__ fp, const LatentCache* __restrict__ lc,
                                           const double* __restrict__ normals, const double* __restrict__ chi2,
                                           int64_t N, double* __restrict__ z) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int d = fp->dim;
  double scale = 1.0;
  if (fp->latent == VMCPDE_LATENT_STUDENT_T) scale = sqrt(lc->nu / chi2[n]);
  for (int i = 0; i < d; ++i) {
    double y = 0.0;
    for (int j = 0; j <= i; ++j) y += lc->C[i * MAX_DIM + j] * normals[n * d + j];
    z[n * d + i] = scale * y + lc->mu[i] + fp->offset[i];
  }
}

// evolutionEq.py:84-119.  hv column conventions are documented in vmcpde.h.
__global__ void rhs_kernel(vmcpde_eq_consts c, int d, const double* __restrict__ x, const double* __restrict__ g,
                           const double* __restrict__ hv, int n_dirs, const double* __restrict__ Dmat, double t,
                           int64_t N, double* __restrict__ eloc) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double* xn = x + n * d;
  const double* gn = g + n * d;
  const double* hn = hv ? hv + n * n_dirs : nullptr;
  double e = 0.0;
  switch (c.kind) {
    case VMCPDE_EQ_DIFFUSION:
    case VMCPDE_EQ_DIFFUSION_DRIFT: {
      double g2 = 0.0, tr = 0.0, gs = 0.0;
      for (int i = 0; i < d; ++i) { g2 += gn[i] * gn[i]; tr += hn[i]; gs += gn[i]; }
      e = c.D * (g2 + tr);                                        // :86
      if (c.kind == VMCPDE_EQ_DIFFUSION_DRIFT) e += c.mu * gs;    // :91-92
    } break;
    case VMCPDE_EQ_DIFFUSION_ANISOTROPIC: {
      double q = 0.0, tr = 0.0;
      for (int i = 0; i < d; ++i) {
        double r = 0.0;
        for (int j = 0; j < d; ++j) r += Dmat[i * d + j] * gn[j];
        q += gn[i] * r;
      }
      for (int k = 0; k < n_dirs; ++k) tr += hn[k];   // tr(H D) = sum_k l_k^T H l_k with D = L L^T
      e = q + tr;                                      // :98-99
    } break;
    case VMCPDE_EQ_ADVECTION_PAPER: {                 // :23-27, :104
      const double cs = cos(M_PI * t / c.T);
      const double X = xn[0], Y = xn[1];
      const double sx = sin(M_PI * X), sy = sin(M_PI * Y);
      const double v0 = -sx * sx * sin(2.0 * M_PI * Y) * cs;
      const double v1 = sy * sy * sin(2.0 * M_PI * X) * cs;
      e = -(gn[0] * v0 + gn[1] * v1);
    } break;
    case VMCPDE_EQ_ADVECTION_HAMILTONIAN:
    case VMCPDE_EQ_ADVECTION_HAMILTONIAN_WDISS: {
      // v = kron(I,[[0,1],[-1,0]]) grad H, H = m w^2/2 sum x^2 + sum p^2/(2m) + lam sum x^4 (:39-42); with c.coupled the
      // quadratic term is m w^2/2 sum (x_i - x_{i-1})^2 over the ring of positions (:33-38)
      double adv = 0.0;
      const int np = d / 2;
      for (int i = 0; i + 1 < d; i += 2) {
        const double q = xn[i], p = xn[i + 1];
        double lin = q;
        if (c.coupled) {
          const int k = i / 2;
          const double ql = xn[2 * (k == 0 ? np - 1 : k - 1)], qr = xn[2 * (k == np - 1 ? 0 : k + 1)];
          lin = 2.0 * q - ql - qr;
        }
        const double dHdq = c.m * c.omega * c.omega * lin + 4.0 * c.lam * q * q * q;
        const double dHdp = p / c.m;
        adv += gn[i] * dHdp + gn[i + 1] * (-dHdq);
      }
      e = -adv;                                        // :79
      if (c.kind == VMCPDE_EQ_ADVECTION_HAMILTONIAN_WDISS) {
        double g2 = 0.0, tr = 0.0, xg = 0.0;
        int k = 0;
        for (int i = 1; i < d; i += 2, ++k) { g2 += gn[i] * gn[i]; tr += hn[k]; xg += xn[i] * gn[i]; }
        e += c.m * c.gamma * c.T * (g2 + tr) + c.gamma * xg;   // :113-117
      }
    } break;
    default: break;
  }
  eloc[n] = e;
}

// waves that share one workgroup's samples: one per output block of the widest layer, at most MAX_WAVES (two per SIMD)
// the block width for a model: 5 when it needs no more waves than 6 does (width 19: 5, 5, 5, 4 against 6, 6, 6, 1), else 6
int pick_jb(const FlowProgram& fp) {
  const int W = fp.max_width;
  const int jb = ((W + JB_B - 1) / JB_B == (W + JB_A - 1) / JB_A) ? JB_B : JB_A;
  return env_int("VMCPDE_FLOW_JB", jb) == JB_B ? JB_B : JB_A;
}

int pick_waves(const FlowProgram& fp, int JB) {
  const int blocks = (fp.max_width + JB - 1) / JB;
  return blocks < 1 ? 1 : (blocks > MAX_WAVES ? MAX_WAVES : blocks);
}

int pick_lanes(int total_units, int wgt_doubles, int waves, int* lds_bytes) {
  // samples per workgroup (a power of two <= 64): the choice that keeps most samples in flight per CU, counting how many
  // workgroups fit in the 160 KB of LDS (per-lane vectors + the shared weight stage) and in the 16 waves per CU the
  // kernels' 128-VGPR budget allows; ties go to the wider workgroup (fewer half-empty waves).  Measured at hidden (43,43):
  // 64 lanes x 1 workgroup 2.9 ms, 32 x 1 5.0 ms, 16 x 3 6.2 ms.
  const int64_t LDS_CU = 160 * 1024, fixed = (int64_t)wgt_doubles * 8;
  const int max_wg = (16 / waves) > 0 ? (16 / waves) : 1;
  int lanes = 1;
  int64_t best = 0;
  for (int l = 64; l >= 1; l >>= 1) {
    const int64_t bytes = (int64_t)total_units * l * 8 + fixed;
    if (bytes > LDS_CU) continue;
    int64_t wg = LDS_CU / bytes;
    if (wg > max_wg) wg = max_wg;
    if (l * wg > best) { best = l * wg; lanes = l; }
  }
  lanes = env_int("VMCPDE_FLOW_LANES", lanes);   // tuning knob
  *lds_bytes = (int)((int64_t)total_units * lanes * 8 + fixed);
  return lanes;
}

}  // namespace

extern "C" int vmcpde_flow_eval_config(const vmcpde_model* m, int n_dirs, int* lanes, int* lds_bytes) {
  VMC_REQUIRE(m, "null model");
  (void)n_dirs;
  const int jb = pick_jb(m->host);
  Regions r = make_regions(m->host, 1, jb);
  int bytes = 0;
  int l = pick_lanes(r.total, r.wgt, pick_waves(m->host, jb), &bytes);
  if (lanes) *lanes = l;
  if (lds_bytes) *lds_bytes = bytes;
  return 0;
}

int vmcpde_prepare_latent_internal(vmcpde_model* m, const double* theta, hipStream_t st) {
  int dev = -1;
  VMC_CHECK_HIP(hipGetDevice(&dev));     // the flow program and the latent cache live on the GPU the handle was created on
  VMC_REQUIRE(dev == m->device, "model handle belongs to GPU %d but the current device is %d (one handle per GPU)", m->device, dev);
  hipLaunchKernelGGL(latent_prepare_kernel, dim3(1), dim3(64), 0, st, m->dev, theta, m->cache);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t vmcpde_flow_eval_workspace_bytes(const vmcpde_model* m, int64_t n_samples) {
  if (!m || n_samples <= 0) return 0;
  const int64_t ldw = (n_samples + 63) / 64 * 64;
  return (int64_t)m->host.act_units * ldw * (int64_t)sizeof(double);
}

// x given (zin == NULL): the batch by its points.  zin given: the batch by its LATENT points; x (= xout) is then an output (flow_eval_kernel<.., true>).
static int flow_eval_impl(vmcpde_model* m, const double* theta, const double* x, const double* zin, double* xout, int64_t n_samples,
                          const double* dirs, int32_t n_dirs, double* logp, double* g, double* hv,
                          void* JT, int64_t ldj, int32_t jt_f32, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(n_samples >= 0, "negative sample count");
  if (n_samples == 0) return 0;   // empty batch (torch gives a NULL data pointer for it)
  VMC_REQUIRE(m && theta && (x || (zin && xout)), "vmcpde_flow_eval: null argument");
  VMC_REQUIRE(n_dirs >= 0 && n_dirs <= VMCPDE_MAX_DIRS, "n_dirs %d outside [0,%d]", n_dirs, VMCPDE_MAX_DIRS);
  VMC_REQUIRE(n_dirs == 0 || (dirs && hv), "directions given without dirs/hv buffers");
  VMC_REQUIRE(JT == nullptr || ldj >= n_samples, "ldj %lld < n_samples %lld", (long long)ldj, (long long)n_samples);
  const bool from_latent = zin != nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  // Hessian directions per forward pass: 4 or 2 when every workgroup of the launch is still resident at once with the larger
  // per-sample LDS footprint (narrow networks, moderate batches -- the reference's own problem sizes), else 1
  int kd = 1;
  const int jb = pick_jb(m->host);
  const bool useB = (jb == JB_B) && (JB_A != JB_B);
  Regions r = make_regions(m->host, 1, jb);
  int bytes = 0;
  int lanes = pick_lanes(r.total, r.wgt, pick_waves(m->host, jb), &bytes);
  VMC_REQUIRE(bytes <= 160 * 1024, "network too wide for the LDS-resident flow kernel (%d B per sample lane)", r.total * 8);
  if (hv != nullptr && n_dirs >= 2 && env_int("VMCPDE_FLOW_MULTIDIR", 1)) {
    for (int cand : {4, 2}) {
      if (cand == 4 && n_dirs < 3) continue;
      const Regions rc = make_regions(m->host, cand, jb);
      int bc = 0;
      const int lc_ = pick_lanes(rc.total, rc.wgt, pick_waves(m->host, jb), &bc);
      if (bc > 160 * 1024 || lc_ != lanes) continue;              // (same samples per workgroup as the one-direction plan)
      const int64_t resident = 256 * (int64_t)((160 * 1024) / bc);
      if ((n_samples + lc_ - 1) / lc_ <= resident) {
        kd = cand; r = rc; bytes = bc; lanes = lc_;
        break;
      }
    }
  }
  const int waves = pick_waves(m->host, jb);
  // more than twelve waves per CU possible by LDS and by the 16-wave budget of a 128-VGPR kernel -> the capped instantiation
  const int wg_lds = (160 * 1024) / bytes, wg_cap = 16 / waves;
  const bool cap = (kd == 1) && ((wg_lds < wg_cap ? wg_lds : wg_cap) * waves > 12) && env_int("VMCPDE_FLOW_CAP", 1);
  const int variant = (kd == 4) ? 3 : (kd == 2) ? 2 : cap ? 1 : 0;
  const bool fast = flow_fast_path(m->host, (int)jt_f32);
#define VMC_FLOW_EVAL_KERNEL_FL(NS_, FL_)                                                                                                  \
  ((kd == 4) ? (const void*)NS_::flow_eval_kernel<4, false, FL_> : (kd == 2) ? (const void*)NS_::flow_eval_kernel<2, false, FL_>           \
   : cap ? (const void*)NS_::flow_eval_kernel<1, true, FL_> : (const void*)NS_::flow_eval_kernel<1, false, FL_>)
#define VMC_FLOW_EVAL_KERNEL(NS_) (from_latent ? VMC_FLOW_EVAL_KERNEL_FL(NS_, true) : VMC_FLOW_EVAL_KERNEL_FL(NS_, false))
  const void* kern = fast ? (useB ? VMC_FLOW_EVAL_KERNEL(jbBf) : VMC_FLOW_EVAL_KERNEL(jbAf)) : (useB ? VMC_FLOW_EVAL_KERNEL(jbB) : VMC_FLOW_EVAL_KERNEL(jbA));
#undef VMC_FLOW_EVAL_KERNEL
#undef VMC_FLOW_EVAL_KERNEL_FL
  if (bytes > 64 * 1024) {
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return -2;
    DeviceCtxLock lock;
    int& have = ctx->flow_eval_lds_by[(useB ? 1 : 0) + (fast ? 2 : 0) + (from_latent ? 4 : 0)][variant];
    if (bytes > have) {
      VMC_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      have = bytes;
    }
  }
  const int64_t grid = (n_samples + lanes - 1) / lanes;
  VMC_REQUIRE(workspace == nullptr || workspace_bytes >= vmcpde_flow_eval_workspace_bytes(m, n_samples),
              "flow_eval workspace too small: need %lld bytes", (long long)vmcpde_flow_eval_workspace_bytes(m, n_samples));
#define VMC_LAUNCH_FLOW_EVAL(NS_, ...)                                                                                                   \
  hipLaunchKernelGGL((NS_::flow_eval_kernel<__VA_ARGS__>), dim3((unsigned)grid), dim3(64 * waves), bytes, st, m->dev, m->cache, theta, x, \
                     n_samples, dirs, (hv ? n_dirs : 0), logp, g, hv, JT, ldj, (int)jt_f32, static_cast<double*>(workspace),             \
                     (int64_t)((n_samples + 63) / 64 * 64), lanes, r, zin, xout)
#define VMC_LAUNCH_FLOW_EVAL_FL(NS_, FL_)                   \
  do {                                                      \
    if (kd == 4) VMC_LAUNCH_FLOW_EVAL(NS_, 4, false, FL_);  \
    else if (kd == 2) VMC_LAUNCH_FLOW_EVAL(NS_, 2, false, FL_); \
    else if (cap) VMC_LAUNCH_FLOW_EVAL(NS_, 1, true, FL_);  \
    else VMC_LAUNCH_FLOW_EVAL(NS_, 1, false, FL_);          \
  } while (0)
#define VMC_LAUNCH_FLOW_EVAL_NS(NS_)                        \
  do {                                                      \
    if (from_latent) VMC_LAUNCH_FLOW_EVAL_FL(NS_, true);    \
    else VMC_LAUNCH_FLOW_EVAL_FL(NS_, false);               \
  } while (0)
  if (fast) {
    if (useB) VMC_LAUNCH_FLOW_EVAL_NS(jbBf);
    else VMC_LAUNCH_FLOW_EVAL_NS(jbAf);
  } else {
    if (useB) VMC_LAUNCH_FLOW_EVAL_NS(jbB);
    else VMC_LAUNCH_FLOW_EVAL_NS(jbA);
  }
#undef VMC_LAUNCH_FLOW_EVAL_NS
#undef VMC_LAUNCH_FLOW_EVAL_FL
#undef VMC_LAUNCH_FLOW_EVAL
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_flow_eval(vmcpde_model* m, const double* theta, const double* x, int64_t n_samples,
                                const double* dirs, int32_t n_dirs, double* logp, double* g, double* hv,
                                void* JT, int64_t ldj, int32_t jt_f32, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n_samples > 0) VMC_REQUIRE(x, "vmcpde_flow_eval: null argument");
  return flow_eval_impl(m, theta, x, nullptr, nullptr, n_samples, dirs, n_dirs, logp, g, hv, JT, ldj, jt_f32, workspace, workspace_bytes, stream);
}

// ABI 7: the same evaluation for a batch given by its LATENT points z (what vmcpde_sample_latent / the MCMC sampler draw): x = inverse flow of z is
// an OUTPUT, produced by the reverse sweep on its way from the latent end (var_state.py:76-79 + :55-67 in one launch: the separate inverse-flow
// launch of a TDVP call is gone).  Same outputs as vmcpde_flow_sample(inv = 1) followed by vmcpde_flow_eval at its x -- to round-off, not bit for bit:
// the sweep divides by exp(s) where the inverse flow multiplies by exp(-s).
extern "C" int vmcpde_flow_eval_from_latent(vmcpde_model* m, const double* theta, const double* z, int64_t n_samples, const double* dirs, int32_t n_dirs,
                                            double* x_out, double* logp, double* g, double* hv, void* JT, int64_t ldj, int32_t jt_f32, void* stream) {
  if (n_samples > 0) VMC_REQUIRE(z && x_out, "vmcpde_flow_eval_from_latent: null argument");
  // net.py:150-151: with global_change the reference's inverse direction applies the block's affine step AFTER its inverse couplings -- as written
  // it is not the inverse of the forward map (the oracle restates it as written), so the samples of var_state.py:79 are not the points the
  // reverse sweep arrives at.  Such a model keeps the two launches (vmcpde_flow_sample, then vmcpde_flow_eval at its x).
  VMC_REQUIRE(!m || !m->host.global_change, "vmcpde_flow_eval_from_latent: a model with global_change samples through vmcpde_flow_sample (net.py:150-151 is not the inverse map)");
  return flow_eval_impl(m, theta, nullptr, z, x_out, n_samples, dirs, n_dirs, logp, g, hv, JT, ldj, jt_f32, nullptr, 0, stream);
}

extern "C" int vmcpde_flow_sample(vmcpde_model* m, const double* theta, const double* z, int64_t n_samples, int inv,
                                  double* x, double* logp, void* stream) {
  if (n_samples <= 0) return 0;
  VMC_REQUIRE(m && theta && z && x, "vmcpde_flow_sample: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  const int jb = pick_jb(m->host);
  const bool useB = (jb == JB_B) && (JB_A != JB_B);
  const int waves = pick_waves(m->host, jb);
  Regions r = make_regions_values(m->host);
  int bytes = 0;
  const int lanes = pick_lanes(r.total, r.wgt, waves, &bytes);
  VMC_REQUIRE(bytes <= 160 * 1024, "network too wide for the LDS-resident flow kernel");
  const bool fast = flow_fast_path(m->host, 0);
  const void* kern = fast ? (useB ? (const void*)jbBf::flow_sample_kernel : (const void*)jbAf::flow_sample_kernel)
                          : (useB ? (const void*)jbB::flow_sample_kernel : (const void*)jbA::flow_sample_kernel);
  if (bytes > 64 * 1024) {
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return -2;
    DeviceCtxLock lock;
    int& have = ctx->flow_sample_lds_by[(useB ? 1 : 0) + (fast ? 2 : 0)];
    if (bytes > have) {
      VMC_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      have = bytes;
    }
  }
  const int64_t grid = (n_samples + lanes - 1) / lanes;
#define VMC_LAUNCH_FLOW_SAMPLE(NS_) \
  hipLaunchKernelGGL(NS_::flow_sample_kernel, dim3((unsigned)grid), dim3(64 * waves), bytes, st, m->dev, m->cache, theta, z, n_samples, inv, x, logp, lanes, r)
  if (fast) {
    if (useB) VMC_LAUNCH_FLOW_SAMPLE(jbBf);
    else VMC_LAUNCH_FLOW_SAMPLE(jbAf);
  } else {
    if (useB) VMC_LAUNCH_FLOW_SAMPLE(jbB);
    else VMC_LAUNCH_FLOW_SAMPLE(jbA);
  }
#undef VMC_LAUNCH_FLOW_SAMPLE
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_latent_from_normals(vmcpde_model* m, const double* theta, const double* normals, const double* chi2,
                                          int64_t n_samples, double* z, void* stream) {
  VMC_REQUIRE(m && theta && normals && z, "vmcpde_latent_from_normals: null argument");
  VMC_REQUIRE(m->host.latent != VMCPDE_LATENT_STUDENT_T || chi2, "Student-t latent needs chi2 draws");
  if (n_samples <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  const int64_t grid = (n_samples + 255) / 256;
  hipLaunchKernelGGL(latent_from_normals_kernel, dim3((unsigned)grid), dim3(256), 0, st, m->dev, m->cache, normals, chi2,
                     n_samples, z);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_rhs(const vmcpde_eq_consts* c, int32_t dim, const double* x, const double* g, const double* hv,
                          int32_t n_dirs, const double* Dmat, double t, int64_t n_samples, double* eloc, void* stream) {
  VMC_REQUIRE(c && x && g && eloc, "vmcpde_rhs: null argument");
  VMC_REQUIRE(dim >= 2 && dim <= VMCPDE_MAX_DIM, "bad dim");
  switch (c->kind) {
    case VMCPDE_EQ_DIFFUSION:
    case VMCPDE_EQ_DIFFUSION_DRIFT:
      VMC_REQUIRE(hv && n_dirs == dim, "diffusion needs hv along the %d coordinate axes", dim);
      break;
    case VMCPDE_EQ_DIFFUSION_ANISOTROPIC:
      VMC_REQUIRE(hv && n_dirs >= 1 && Dmat, "anisotropic diffusion needs hv along the columns of chol(D) and D");
      break;
    case VMCPDE_EQ_ADVECTION_PAPER:
      VMC_REQUIRE(dim == 2, "advection_paper is two-dimensional (evolutionEq.py:24)");
      break;
    case VMCPDE_EQ_ADVECTION_HAMILTONIAN:
      VMC_REQUIRE(dim % 2 == 0, "phase space needs even dim");
      break;
    case VMCPDE_EQ_ADVECTION_HAMILTONIAN_WDISS:
      VMC_REQUIRE(dim % 2 == 0 && hv && n_dirs == dim / 2, "Kramers needs hv along the %d momentum axes", dim / 2);
      break;
    default:
      VMC_REQUIRE(false, "unknown equation kind %d", c->kind);
  }
  if (n_samples <= 0) return 0;
  const int64_t grid = (n_samples + 255) / 256;
  hipLaunchKernelGGL(rhs_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *c, dim, x, g, hv, n_dirs, Dmat, t,
                     n_samples, eloc);
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}
