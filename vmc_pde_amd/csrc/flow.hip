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

#ifndef VMC_FLOW_JB
#define VMC_FLOW_JB 6
#endif
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
constexpr int JB = VMC_FLOW_JB;  // output block held in registers by the dense-layer loops (measured: 6 x up-to-8 waves is the best compromise over C2/C3/C4)
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
Regions make_regions(const FlowProgram& fp, int kdirs = 1) {
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

#ifdef VMC_FLOW_STAMPS
struct FlowStamps;
#endif
struct Ctx {
  double* lds;
  double* wl;     // weight stage: kernel [n_in][n_out] then bias [n_out] of the layer being evaluated
  int lanes;
  int lane;
  int wave, nw;   // this wave's index among the nw waves that share the workgroup's samples
  int jt_f32;   // Jacobian rows are stored as float (BASELINE config 5) instead of double
  uint32_t joff;      // byte offset of this lane's sample inside a Jacobian row (n * element size)
  int64_t jpitch;     // bytes between consecutive Jacobian rows (ldj * element size) -- wave-uniform
  const double* __restrict__ th;
  const FlowProgram* __restrict__ fp;
  // activation workspace (HBM, [unit][sample]): the forward pass leaves every MLP's hidden activations and outputs there so
  // that the reverse sweep reads them back instead of re-evaluating the networks (half of all tanh evaluations otherwise)
  double* __restrict__ actw;   // nullptr: recompute
  int64_t ldw;                 // samples per workspace row
  int64_t samp_st, samp_ld;    // this lane's column for stores (its sample) / loads (clamped into range for idle lanes)
  bool act_store;              // this lane writes (a live sample) in a pass that is asked to store
#ifdef VMC_FLOW_STAMPS
  FlowStamps* st;              // wave 0 only (nullptr elsewhere)
#endif
};

#define LD(u) (c.lds[(u) * c.lanes + c.lane])
#ifdef VMC_FLOW_DIAG_NO_SYNC       // timing diagnostics only (results are wrong): what the workgroup barriers cost
#define SYNC() ((void)0)
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

// Jacobian stores.  A row pointer is wave-uniform (base + row * pitch: SGPR arithmetic) and the lane only adds its 32-bit
// byte offset, which lets the store use the scalar-base + vector-offset addressing form -- no 64-bit VALU address per element.
__device__ __forceinline__ void jt_store(const Ctx& c, void* __restrict__ JT, int64_t /*ldj*/, int64_t row, int64_t /*n*/, double v) {
#ifdef VMC_FLOW_DIAG_NO_STORE      // timing diagnostics only: what the Jacobian stores cost
  if (v != 1.2345e300) return;
#endif
  char* rowp = static_cast<char*>(JT) + row * c.jpitch;
  if (c.jt_f32) *reinterpret_cast<float*>(rowp + c.joff) = (float)v;
  else *reinterpret_cast<double*>(rowp + c.joff) = v;
}
// JB consecutive rows starting at `row0` (nj <= JB of them valid), values vals[jj] * scale: one precision branch and one
// bounds decision per block instead of per element
__device__ __forceinline__ void jt_store_block(const Ctx& c, void* __restrict__ JT, int64_t row0, int nj, const double (&vals)[JB], double scale) {
#ifdef VMC_FLOW_DIAG_NO_STORE
  if (scale * vals[0] != 1.2345e300) return;
#endif
  char* p = static_cast<char*>(JT) + row0 * c.jpitch;
  if (c.jt_f32) {
    if (nj == JB) {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) *reinterpret_cast<float*>(p + jj * c.jpitch + c.joff) = (float)(scale * vals[jj]);
    } else {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
        if (jj < nj) *reinterpret_cast<float*>(p + jj * c.jpitch + c.joff) = (float)(scale * vals[jj]);
    }
  } else {
    if (nj == JB) {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) *reinterpret_cast<double*>(p + jj * c.jpitch + c.joff) = scale * vals[jj];
    } else {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
        if (jj < nj) *reinterpret_cast<double*>(p + jj * c.jpitch + c.joff) = scale * vals[jj];
    }
  }
}

// tanh with absolute error ~2e-16 and no lane-divergent branches (ocml's tanh costs about as much as a 21 x 6 block of the
// layer product and takes both of its branches in most waves): t = exp(-2|x|) by range reduction + degree-13 Taylor
// (|r| <= ln2/2: remainder 4e-18), tanh|x| = (1 - t)/(1 + t) with v_rcp_f64 refined by two Newton steps and a residual
// correction.  Activations are consumed linearly downstream, so absolute (not relative) accuracy near 0 is what matters.
__device__ __forceinline__ double fast_tanh(double x) {
#ifdef VMC_FLOW_DIAG_NO_TANH       // timing diagnostics only (wrong results): what the activation function costs
  return x * 0.25;
#endif
  const double ax = fmin(fabs(x), 20.0);                   // tanh(20) = 1 - 3e-18
  const double y = -2.0 * ax;
  const double n = rint(y * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                       // 1/13!
  p = fma(p, r, 2.08767569878681e-09);                     // 1/12!
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.755731922398589e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 1.984126984126984e-04);
  p = fma(p, r, 1.388888888888889e-03);
  p = fma(p, r, 8.333333333333333e-03);
  p = fma(p, r, 4.166666666666666e-02);
  p = fma(p, r, 1.666666666666667e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double t = ldexp(p, (int)n);
  const double num = 1.0 - t, den = 1.0 + t;
  double q = __builtin_amdgcn_rcp(den);
  q = fma(fma(-den, q, 1.0), q, q);
  q = fma(fma(-den, q, 1.0), q, q);
  double h = num * q;
  h = fma(fma(-den, h, num), q, h);
  return copysign(h, x);
}

// Copy one layer's kernel and bias from the flat parameter vector into the workgroup's weight stage (all threads; the caller
// synchronises).  The layer loops then read weights as LDS broadcasts: in-order returns that hipcc can pipeline against the
// per-lane LDS reads, where scalar loads (out-of-order, 33 KB of weights against a 16 KB scalar cache) forced a full
// lgkmcnt(0) stall of several hundred cycles per input unit.
// The whole MLP at once (layer k at layer_woff(T, k): kernel [n_in][n_out], then bias [n_out]): one round trip to the
// parameter vector and one barrier per network instead of one per layer -- the kernel is bound by exactly these chains.
__device__ __forceinline__ int layer_woff(const Trafo& T, int k) {
  int off = 0;
  for (int q = 0; q < k; ++q) off += T.layer[q].n_in * T.layer[q].n_out + T.layer[q].n_out;
  return off;
}
__device__ __forceinline__ void stage_weights_at(const Ctx& c, const Trafo& T, int off) {
  for (int k = 0; k < T.n_layers; ++k) {
    const DenseLayer& L = T.layer[k];
    const int nk = L.n_in * L.n_out;
    for (int idx = threadIdx.x; idx < nk + L.n_out; idx += blockDim.x)
      c.wl[off + idx] = c.th[(idx < nk) ? (L.off_kernel + idx) : (L.off_bias + idx - nk)];
    off += nk + L.n_out;
  }
  // JB doubles of zeros behind the last layer: the dense loops read whole output blocks of JB weights unconditionally (a block
  // that sticks out of its row reads the next row, the bias row or this slack -- finite values whose products are dropped)
  if (threadIdx.x < JB) c.wl[off + threadIdx.x] = 0.0;
}
// the stage of one network pass: nothing to do when the kernel staged every network at its start (FlowProgram::stage_all)
__device__ __forceinline__ void stage_weights(const Ctx& c, const Trafo& T) {
  if (!c.fp->stage_all) stage_weights_at(c, T, 0);
}
__device__ __forceinline__ int stage_base(const Ctx& c, const Trafo& T) { return c.fp->stage_all ? T.wbase : 0; }
__device__ __forceinline__ void stage_all_networks(const Ctx& c) {
  const FlowProgram& fp = *c.fp;
  for (int h = 0; h < fp.n_half; ++h) {
    stage_weights_at(c, fp.hs[h].s, fp.hs[h].s.wbase);
    if (fp.hs[h].has_t) stage_weights_at(c, fp.hs[h].t, fp.hs[h].t.wbase);
  }
}

// y = tanh-layer(x): z_j = b_j + sum_i W[i][j] a_i (net.py:55-61), with NC Taylor components.
// Cooperative: the input must be visible to all waves on entry; the caller synchronises before the output is read.
template <int NC>
__device__ __forceinline__ void dense_layer(const Ctx& c, const DenseLayer& L, int woff, int uin, int uout, int cs, double scale, int wofs = -1,
                                            bool inplace = false) {
  // inplace (uout == uin; every wave owns at most ONE output block): the block is finished in registers, a barrier waits for the
  // last reader of the inputs, then the outputs overwrite them.  Every wave makes exactly one trip so that all reach the barrier.
  const int n_in = L.n_in, n_out = L.n_out;
  const double* __restrict__ W = c.wl + woff;           // staged by the caller (stage_weights + barrier)
  const double* __restrict__ Bv = W + n_in * n_out;
  for (int j0 = c.wave * JB; inplace ? (j0 == c.wave * JB) : (j0 < n_out); j0 += c.nw * JB) {     // output blocks round-robin over the waves
    double acc[NC][JB];
    const bool has = j0 < n_out;
    if (has) {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        acc[0][jj] = (j0 + jj < n_out) ? Bv[j0 + jj] : 0.0;
#pragma unroll
        for (int cc = 1; cc < NC; ++cc) acc[cc][jj] = 0.0;
      }
#pragma unroll VMC_FLOW_UNROLL
      for (int i = 0; i < n_in; ++i) {
        double a[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) a[cc] = LD(uin + cc * cs + i);
        const double* __restrict__ wr = W + i * n_out + j0;
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const double w = wr[jj];                              // (unconditional: see stage_weights)
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) acc[cc][jj] = fma(w, a[cc], acc[cc][jj]);
        }
      }
      // activation and its Taylor coefficients, in registers
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
#ifdef VMC_FLOW_OCML_TANH
        const double h = tanh(acc[0][jj]);
#else
        const double h = fast_tanh(acc[0][jj]);
#endif
        acc[0][jj] = scale * h;
        if (NC > 1) {
          const double hp = 1.0 - h * h;
#pragma unroll
          for (int kd = 0; kd < (NC - 1) / 2; ++kd) {           // first / second derivative along direction kd
            const double z1 = acc[2 * kd + 1][jj], z2 = acc[2 * kd + 2][jj];
            acc[2 * kd + 1][jj] = scale * hp * z1;
            acc[2 * kd + 2][jj] = scale * hp * (z2 - 2.0 * h * z1 * z1);
          }
        }
      }
    }
    if (inplace) SYNC();
    if (has) {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        if (j0 + jj < n_out) {
          LD(uout + j0 + jj) = acc[0][jj];
          if (wofs >= 0 && c.act_store) c.actw[(int64_t)(wofs + j0 + jj) * c.ldw + c.samp_st] = acc[0][jj];
#pragma unroll
          for (int cc = 1; cc < NC; ++cc) LD(uout + cc * cs + j0 + jj) = acc[cc][jj];
        }
      }
    }
  }
}

// SingleTrafo forward (net.py:53-61), ping-pong between buffers uA (holding the input) and uB. Returns output base.
// wofs >= 0: the value component of every layer output also goes to the activation workspace, hidden layers first (in layer
// order), then the outputs -- the order mlp_load_activations reads them back in.
template <int NC>
__device__ __forceinline__ int mlp_forward(const Ctx& c, const Trafo& T, int uA, int uB, int cs, int wofs = -1) {
  const bool inplace = (uA == uB);          // Regions::inplace: no second buffer
  int cur = uA, nxt = uB;
  int wl = wofs, woff = stage_base(c, T);
  stage_weights(c, T);
  SYNC();                                   // (also orders the previous user's reads of the stage before this write: every pass ends with a barrier)
  FSTAMP(3);
  for (int k = 0; k < T.n_layers; ++k) {
    dense_layer<NC>(c, T.layer[k], woff, cur, nxt, cs, (k == T.n_layers - 1) ? c.fp->alpha : 1.0, wl, inplace);
    FSTAMP(1);
    SYNC();
    FSTAMP(4);
    if (wl >= 0) wl += T.layer[k].n_out;
    woff += T.layer[k].n_in * T.layer[k].n_out + T.layer[k].n_out;
    const int t = cur; cur = nxt; nxt = t;
  }
  return cur;
}

// One coupling half-step in the forward direction (real -> latent), net.py:86-113, with NC Taylor components.
template <int NC>
__device__ __forceinline__ void halfstep_forward(const Ctx& c, const HalfStep& h, const Regions& r, double* lj) {
  // wave 0 owns the state X and the log-jacobian accumulator lj; the other waves only take part in the MLPs
  const int d = c.fp->dim, cs = r.cs;
  if (W0)
    for (int i = 0; i < h.n_a; ++i)
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) LD(r.A + cc * cs + i) = LD(r.X + cc * d + h.idx_a[i]);
  SYNC();
  FSTAMP(0);
  const bool keep = (c.actw != nullptr);
  int us = mlp_forward<NC>(c, h.s, r.A, r.B, cs, keep ? h.act_off_s : -1);
  FSTAMP(1);
  int ut = 0;
  const int coupling = c.fp->coupling;
  if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) {
    if (W0) {
      for (int k = 0; k < h.n_c; ++k)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) LD(r.SOUT + cc * d + k) = LD(us + cc * cs + k);
      for (int i = 0; i < h.n_a; ++i)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) LD(r.A + cc * cs + i) = LD(r.X + cc * d + h.idx_a[i]);
    }
    SYNC();
    FSTAMP(0);
    ut = mlp_forward<NC>(c, h.t, r.A, r.B, cs, keep ? h.act_off_t : -1);
    FSTAMP(1);
  }
  if (!W0) return;     // (the next stage starts with a barrier that wave 0 reaches after the update below)
  for (int k = 0; k < h.n_c; ++k) {
    double s[NC], t[NC], cv[NC], y[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) s[cc] = t[cc] = cv[cc] = y[cc] = 0.0;
    const int ix = h.idx_c[k];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) {
        s[cc] = LD(r.SOUT + cc * d + k);
        t[cc] = LD(ut + cc * cs + k);
      } else {
        s[cc] = LD(us + cc * cs + k);
      }
      cv[cc] = LD(r.X + cc * d + ix);
    }
    if (coupling == VMCPDE_COUPLING_JAC_EQ_1) {  // net.py:90-92: v = u + s, log-jacobian 0
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) y[cc] = cv[cc] + s[cc];
    } else {
      const double E = exp(s[0]);
      y[0] = cv[0] * E;
#pragma unroll
      for (int kd = 0; kd < (NC - 1) / 2; ++kd) {
        const int c1 = 2 * kd + 1, c2 = 2 * kd + 2;
        const double E1 = E * s[c1];
        const double E2 = E * (s[c2] + s[c1] * s[c1]);
        y[c1] = cv[c1] * E + cv[0] * E1;
        y[c2] = cv[c2] * E + 2.0 * cv[c1] * E1 + cv[0] * E2;
      }
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) y[cc] += t[cc];   // net.py:94
        if (coupling == VMCPDE_COUPLING_PLAIN) y[cc] += s[cc];           // net.py:98
        lj[cc] += s[cc];                                                 // net.py:118
      }
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) LD(r.X + cc * d + ix) = y[cc];
  }
  FSTAMP(2);
}

// The global_change step that closes block b in the forward direction (net.py:115-116): y = scale * r + offset on all d
// coordinates, log-jacobian += d * log(scale).  Wave 0 (the owner of X and lj) only.
template <int NC>
__device__ __forceinline__ void global_forward(const Ctx& c, int b, const Regions& r, double* lj) {
  if (!W0) return;
  const int d = c.fp->dim;
  const double gs = c.th[c.fp->off_gscale[b]];
  for (int i = 0; i < d; ++i) {
    LD(r.X + i) = gs * LD(r.X + i) + c.th[c.fp->off_goffset[b] + i];
#pragma unroll
    for (int cc = 1; cc < NC; ++cc) LD(r.X + cc * d + i) = gs * LD(r.X + cc * d + i);
  }
  lj[0] += (double)d * log(gs);
}

// ... and what the reference's inverse direction does with it (net.py:150-151): AFTER the inverse coupling steps,
// x = (r - offset) / scale and log-jacobian -= d * log(scale).  As written, not the inverse of global_forward (see the oracle).
__device__ __forceinline__ void global_inverse_as_written(const Ctx& c, int b, const Regions& r, double& lj) {
  if (!W0) return;
  const int d = c.fp->dim;
  const double gs = c.th[c.fp->off_gscale[b]];
  for (int i = 0; i < d; ++i) LD(r.X + i) = (LD(r.X + i) - c.th[c.fp->off_goffset[b] + i]) / gs;
  lj -= (double)d * log(gs);
}

// Reverse sweep through global_forward: on entry X = y, G = dlogp/dy; on exit X = r, G = dlogp/dr, and the d + 1 Jacobian
// rows of the block are written: dlogp/d offset_i = G_i, dlogp/d scale = sum_i G_i r_i + d / scale.
__device__ __forceinline__ void global_backward(const Ctx& c, int b, const Regions& r, void* __restrict__ JT, int64_t ldj, int64_t n,
                                                bool active) {
  if (!W0) return;
  const int d = c.fp->dim;
  const double gs = c.th[c.fp->off_gscale[b]];
  double dscale = (double)d / gs;
  for (int i = 0; i < d; ++i) {
    const double gi = LD(r.G + i);
    const double ri = (LD(r.X + i) - c.th[c.fp->off_goffset[b] + i]) / gs;
    if (JT != nullptr && active) jt_store(c, JT, ldj, c.fp->off_goffset[b] + i, n, gi);
    dscale = fma(gi, ri, dscale);
    LD(r.X + i) = ri;
    LD(r.G + i) = gs * gi;
  }
  if (JT != nullptr && active) jt_store(c, JT, ldj, c.fp->off_gscale[b], n, dscale);
}

// One coupling half-step in the inverse direction (latent -> real), values only, net.py:121-148.
__device__ __forceinline__ void halfstep_inverse(const Ctx& c, const HalfStep& h, const Regions& r, double& lj) {
  const int cs = r.cs;
  if (W0)
    for (int i = 0; i < h.n_a; ++i) LD(r.A + i) = LD(r.X + h.idx_a[i]);
  SYNC();
  int us = mlp_forward<1>(c, h.s, r.A, r.B, cs);
  int ut = 0;
  const int coupling = c.fp->coupling;
  if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) {
    if (W0) {
      for (int k = 0; k < h.n_c; ++k) LD(r.SOUT + k) = LD(us + k);
      for (int i = 0; i < h.n_a; ++i) LD(r.A + i) = LD(r.X + h.idx_a[i]);
    }
    SYNC();
    ut = mlp_forward<1>(c, h.t, r.A, r.B, cs);
  }
  if (!W0) return;
  for (int k = 0; k < h.n_c; ++k) {
    const int ix = h.idx_c[k];
    const double v = LD(r.X + ix);
    double u;
    if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) {
      const double s = LD(r.SOUT + k);
      u = (v - LD(ut + k)) * exp(-s);   // net.py:129,140
      lj -= s;
    } else {
      const double s = LD(us + k);
      if (coupling == VMCPDE_COUPLING_JAC_EQ_1) {
        u = v - s;                      // net.py:126,137
      } else if (coupling == VMCPDE_COUPLING_NO_ADD) {
        u = v * exp(-s);                // net.py:131,142
        lj -= s;
      } else {
        u = (v - s) * exp(-s);          // net.py:133,144
        lj -= s;
      }
    }
    LD(r.X + ix) = u;
  }
}

// Latent log-density (net.py:10-36) of the state in X with NC Taylor components; returns logp components in lp[].
// r = z - offset - mu; y = U^-1 r (U upper triangular, U U^T = S, util.py:21-26); Q = |y|^2 = r^T S^-1 r.
template <int NC>
__device__ __forceinline__ void latent_forward(const Ctx& c, const LatentCache* __restrict__ lc, const Regions& r, int ux, double* lp) {
  const int d = c.fp->dim;
  double Q[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) Q[cc] = 0.0;
  for (int i = d - 1; i >= 0; --i) {
    double y[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      y[cc] = LD(ux + cc * d + i);
      if (cc == 0) y[0] -= c.fp->offset[i] + lc->mu[i];
    }
    for (int j = i + 1; j < d; ++j) {
      const double uij = lc->U[i * MAX_DIM + j];
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) y[cc] -= uij * LD(r.LAT + cc * d + j);
    }
    const double inv = 1.0 / lc->U[i * MAX_DIM + i];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      y[cc] *= inv;
      LD(r.LAT + cc * d + i) = y[cc];
    }
    Q[0] += y[0] * y[0];
#pragma unroll
    for (int kd = 0; kd < (NC - 1) / 2; ++kd) {
      Q[2 * kd + 1] += 2.0 * y[0] * y[2 * kd + 1];
      Q[2 * kd + 2] += 2.0 * (y[2 * kd + 1] * y[2 * kd + 1] + y[0] * y[2 * kd + 2]);
    }
  }
  if (c.fp->latent == VMCPDE_LATENT_GAUSS) {
    lp[0] = lc->lognorm - 0.5 * Q[0];   // net.py:18-20
#pragma unroll
    for (int cc = 1; cc < NC; ++cc) lp[cc] = -0.5 * Q[cc];
  } else {
    const double nu = lc->nu, p = (double)d;   // net.py:32-36
    const double hnp = 0.5 * (nu + p);
    lp[0] = lc->lognorm - hnp * log1p(Q[0] / nu);
    if (NC > 1) {
      const double f1 = -hnp / (nu + Q[0]);
      const double f2 = hnp / ((nu + Q[0]) * (nu + Q[0]));
#pragma unroll
      for (int kd = 0; kd < (NC - 1) / 2; ++kd) {
        lp[2 * kd + 1] = f1 * Q[2 * kd + 1];
        lp[2 * kd + 2] = f1 * Q[2 * kd + 2] + f2 * Q[2 * kd + 1] * Q[2 * kd + 1];
      }
    }
  }
}

// SingleTrafo forward keeping every layer input in ACT (input already in ACT[0..n_in)); outputs go to uOUT.
__device__ __forceinline__ void mlp_forward_store(const Ctx& c, const Trafo& T, int uACT, int uOUT) {
  int in = uACT, woff = stage_base(c, T);
  stage_weights(c, T);
  SYNC();
  for (int k = 0; k < T.n_layers; ++k) {
    const DenseLayer& L = T.layer[k];
    const bool last = (k == T.n_layers - 1);
    const int out = last ? uOUT : in + L.n_in;
    dense_layer<1>(c, L, woff, in, out, 0, last ? c.fp->alpha : 1.0);
    SYNC();
    woff += L.n_in * L.n_out + L.n_out;
    in = out;
  }
}

// The same result read back from the activation workspace the forward pass filled (mlp_forward with wofs >= 0): hidden
// activations behind the n_in inputs already in ACT, outputs to uOUT.  Units go round-robin over the waves.
__device__ __forceinline__ void mlp_load_activations(const Ctx& c, const Trafo& T, int uACT, int uOUT, int wofs) {
  const int nh = T.sum_hidden, total = nh + T.n_out;
  for (int u = c.wave; u < total; u += c.nw) {
    const double v = c.actw[(int64_t)(wofs + u) * c.ldw + c.samp_ld];
    LD((u < nh) ? (uACT + T.n_in + u) : (uOUT + u - nh)) = v;
  }
  SYNC();
}

// Reverse pass through one SingleTrafo with output cotangent in uW: writes the parameter-gradient rows of this
// sample into JT (bias: delta_j, kernel: a_i delta_j) and accumulates the input gradient into GAM.
// Cooperative: every layer is split by INPUT unit i (wave w owns i = w, w+nw, ...): the owner forms
// eps_i = sum_j W[i][j] delta_j, stores the kernel rows (i, :) of JT, and turns eps_i into the next delta_i; bias rows go
// round-robin by output block.  On entry OUT/ACT are visible to all waves and uW was written by wave 0.
// staged: the weights of T are already in the LDS stage (the forward-store pass of the same network ran just before).
__device__ __forceinline__ void mlp_backward(const Ctx& c, const Trafo& T, const Regions& r, int uOUT, int uW,
                                             void* __restrict__ JT, int64_t ldj, int64_t n, bool active, bool staged) {
  const double alpha = c.fp->alpha;
  const int nl = T.n_layers;
  if (!staged) stage_weights(c, T);
  if (W0) {
    const DenseLayer& L = T.layer[nl - 1];
    for (int j = 0; j < L.n_out; ++j) {
      const double th = LD(uOUT + j) / alpha;
      LD(r.DEL0 + j) = LD(uW + j) * alpha * (1.0 - th * th);
    }
  }
  int in_off = T.n_in + T.sum_hidden;  // one past the last layer input
  int cur = r.DEL0;                    // cotangent of this layer's output pre-activations: DEL0 for the output layer, then the ACT
                                       // segment of the layer above (overwritten in place by its owner waves, see below)
  for (int k = nl - 1; k >= 0; --k) {
    const DenseLayer& L = T.layer[k];
    const int n_in = L.n_in, n_out = L.n_out;
    in_off -= n_in;
    const int uin = r.ACT + in_off;
    if (k == nl - 1) SYNC();                  // publishes wave 0's DEL0 (and the weights if they were staged here); later layers
                                              // start behind the barrier that closed the previous one
    const double* __restrict__ W = c.wl + stage_base(c, T) + layer_woff(T, k);
    if (VMC_FLOW_EPS_REGS && r.inplace) {
      // eps_i = sum_j W[i][j] delta_j of this wave's input units i = wave + m * nw in REGISTERS: n_in <= JB * nw whenever the plan
      // is "inplace" (one output block per wave), so m < JB -- no EPS line in LDS and no read-modify-write through it
      double e[JB];
#pragma unroll
      for (int m = 0; m < JB; ++m) e[m] = 0.0;
      for (int j0 = 0, blk = 0; j0 < n_out; j0 += JB, ++blk) {
        double dl[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) dl[jj] = (j0 + jj < n_out) ? LD(cur + j0 + jj) : 0.0;
        const int nj = (n_out - j0 < JB) ? (n_out - j0) : JB;
        if (JT != nullptr && active && (blk % c.nw) == c.wave) jt_store_block(c, JT, L.off_bias + j0, nj, dl, 1.0);
#pragma unroll
        for (int m = 0; m < JB; ++m) {
          const int i = c.wave + m * c.nw;
          if (i < n_in) {
            const double a = LD(uin + i);
            const double* __restrict__ wr = W + i * n_out + j0;
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) e[m] = fma(wr[jj], dl[jj], e[m]);      // (dl is 0 beyond n_out; the stage has JB doubles of slack)
            if (JT != nullptr && active) jt_store_block(c, JT, L.off_kernel + i * n_out + j0, nj, dl, a);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < JB; ++m) {
        const int i = c.wave + m * c.nw;
        if (i < n_in) {
          if (k > 0) {
            // delta of the layer below = tanh'(its output) * eps, written OVER that output a_i: wave w is the only reader of a_i
            // for its i = w, w + nw, ..., and the layer below reads this segment only as its output cotangent
            const double a = LD(uin + i);
            LD(uin + i) = (1.0 - a * a) * e[m];
          } else {
            LD(r.GAM + i) += e[m];
          }
        }
      }
    } else {
      for (int i = c.wave; i < n_in; i += c.nw) LD(r.EPS + i) = 0.0;
      for (int j0 = 0, blk = 0; j0 < n_out; j0 += JB, ++blk) {
        double dl[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) dl[jj] = (j0 + jj < n_out) ? LD(cur + j0 + jj) : 0.0;
        const int nj = (n_out - j0 < JB) ? (n_out - j0) : JB;
        if (JT != nullptr && active && (blk % c.nw) == c.wave) jt_store_block(c, JT, L.off_bias + j0, nj, dl, 1.0);
        for (int i = c.wave; i < n_in; i += c.nw) {
          const double a = LD(uin + i);
          const double* __restrict__ wr = W + i * n_out + j0;
          double e = 0.0;
#pragma unroll
          for (int jj = 0; jj < JB; ++jj) {
            const double w = (j0 + jj < n_out) ? wr[jj] : 0.0;
            e = fma(w, dl[jj], e);
          }
          LD(r.EPS + i) += e;
          if (JT != nullptr && active) jt_store_block(c, JT, L.off_kernel + i * n_out + j0, nj, dl, a);
        }
      }
      if (k > 0) {
        for (int i = c.wave; i < n_in; i += c.nw) {
          const double a = LD(uin + i);
          LD(uin + i) = (1.0 - a * a) * LD(r.EPS + i);
        }
      } else {
        for (int i = c.wave; i < n_in; i += c.nw) LD(r.GAM + i) += LD(r.EPS + i);
      }
    }
    if (k > 0) cur = uin;
    SYNC();
  }
}

// Reverse sweep through one coupling half-step: on entry X holds the half-step's OUTPUT y and G = dlogp/dy;
// on exit X holds its INPUT and G = dlogp/d(input).
__device__ __forceinline__ void halfstep_backward(const Ctx& c, const HalfStep& h, const Regions& r,
                                                  void* __restrict__ JT, int64_t ldj, int64_t n, bool active) {
  // wave 0 owns X, G, GAM, DW between the cooperative MLP passes (each of which ends with a barrier)
  const int coupling = c.fp->coupling;
  if (W0) {
    for (int i = 0; i < h.n_a; ++i) LD(r.GAM + i) = 0.0;
    for (int i = 0; i < h.n_a; ++i) LD(r.ACT + i) = LD(r.X + h.idx_a[i]);
  }
  SYNC();
  FSTAMP(6);
  if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) {
    // t enters additively (net.py:94): its output cotangent is G_c itself
    if (c.actw) mlp_load_activations(c, h.t, r.ACT, r.OUTT, h.act_off_t);
    else mlp_forward_store(c, h.t, r.ACT, r.OUTT);
    FSTAMP(5);
    if (W0)
      for (int k = 0; k < h.n_c; ++k) LD(r.DW + k) = LD(r.G + h.idx_c[k]);
    mlp_backward(c, h.t, r, r.OUTT, r.DW, JT, ldj, n, active, c.actw == nullptr);     // (starts with wave 0's part, then a barrier)
    FSTAMP(7);
    // ACT[0..n_a) still holds the half-step input for the s network
  }
  if (c.actw) mlp_load_activations(c, h.s, r.ACT, r.OUTS, h.act_off_s);
  else mlp_forward_store(c, h.s, r.ACT, r.OUTS);
  FSTAMP(5);
  if (W0) {
    for (int k = 0; k < h.n_c; ++k) {
      const int ix = h.idx_c[k];
      const double s = LD(r.OUTS + k);
      const double y = LD(r.X + ix);
      const double gk = LD(r.G + ix);
      double cin, w, gnew;
      if (coupling == VMCPDE_COUPLING_JAC_EQ_1) {
        cin = y - s; w = gk; gnew = gk;
      } else {
        const double E = exp(s);
        double ces;  // c * exp(s)
        if (coupling == VMCPDE_COUPLING_NO_ADD) ces = y;
        else if (coupling == VMCPDE_COUPLING_DIFFERENT_ADD) ces = y - LD(r.OUTT + k);
        else ces = y - s;
        cin = ces / E;
        w = gk * ces + 1.0;                                   // d/ds [ G_c c e^s + s ]
        if (coupling == VMCPDE_COUPLING_PLAIN) w += gk;       // + G_c d(s)/ds of the additive s (net.py:98)
        gnew = gk * E;
      }
      LD(r.DW + k) = w;
      LD(r.X + ix) = cin;
      LD(r.G + ix) = gnew;
    }
  }
  FSTAMP(6);
  mlp_backward(c, h.s, r, r.OUTS, r.DW, JT, ldj, n, active, c.actw == nullptr);
  FSTAMP(7);
  if (W0)
    for (int i = 0; i < h.n_a; ++i) LD(r.G + h.idx_a[i]) += LD(r.GAM + i);
}

// Adjoint of the latent log-density: G = dlogp/dz and the latent-parameter rows of JT (L, L_diag, dist_params, mu).
__device__ __forceinline__ void latent_backward(const Ctx& c, const LatentCache* __restrict__ lc, const Regions& r,
                                                void* __restrict__ JT, int64_t ldj, int64_t n, bool active) {
  const FlowProgram& fp = *c.fp;
  const int d = fp.dim;
  double Q = 0.0;
  for (int i = d - 1; i >= 0; --i) {   // y = U^-1 r
    double y = LD(r.X + i) - fp.offset[i] - lc->mu[i];
    for (int j = i + 1; j < d; ++j) y -= lc->U[i * MAX_DIM + j] * LD(r.Y + j);
    y /= lc->U[i * MAX_DIM + i];
    LD(r.Y + i) = y;
    Q += y * y;
  }
  for (int i = 0; i < d; ++i) {        // q = U^-T y = S^-1 r
    double q = LD(r.Y + i);
    for (int j = 0; j < i; ++j) q -= lc->U[j * MAX_DIM + i] * LD(r.Q + j);
    q /= lc->U[i * MAX_DIM + i];
    LD(r.Q + i) = q;
  }
  // logp_latent = lognorm + f(Q):  Gauss f = -Q/2,  Student-t f = -(nu+p)/2 log(1+Q/nu)
  double m2f1;  // -2 f'(Q)
  const bool gauss = (fp.latent == VMCPDE_LATENT_GAUSS);
  const double nu = lc->nu, p = (double)d;
  if (gauss) m2f1 = 1.0;
  else m2f1 = (nu + p) / (nu + Q);
  for (int i = 0; i < d; ++i) LD(r.G + i) = -m2f1 * LD(r.Q + i);
  if (JT != nullptr && active) {
    int k = 0;
    for (int i = 0; i < d; ++i) {
      const double qi = LD(r.Q + i);
      for (int j = i + 1; j < d; ++j, ++k)   // dQ/dU_ij = -2 q_i y_j ; row-major strict upper triangle (util.py:23-24)
        jt_store(c, JT, ldj, fp.off_L + k, n, m2f1 * qi * LD(r.Y + j));
      // U_ii = exp(L_diag_i); Gauss also carries -1/2 log det S = -sum L_diag (net.py:20)
      jt_store(c, JT, ldj, fp.off_Ldiag + i, n, m2f1 * qi * LD(r.Y + i) * lc->U[i * MAX_DIM + i] - (gauss ? 1.0 : 0.0));
      jt_store(c, JT, ldj, fp.off_mu + i, n, m2f1 * qi);
    }
    if (!gauss) {
      const double dnu = -0.5 * log1p(Q / nu) + 0.5 * (nu + p) * Q / (nu * (nu + Q));
      jt_store(c, JT, ldj, fp.off_dist, n, lc->dlognorm_dtau + (nu - 1.0) * dnu);   // nu = exp(tau)+1 (net.py:32)
    }
  }
}

// KD: Hessian directions per forward Taylor pass (1 + 2 KD components).  One direction per pass needs the least LDS (every
// configuration of BASELINE); narrow networks take KD = 2 or 4, which shares the value component and the weight reads of a
// pass among the directions (d = 8 diffusion: 2 passes instead of 8) -- chosen by vmcpde_flow_eval.
// CAP: the register allocation is held to 128 VGPRs (four waves per SIMD; 52 B of scratch per lane) -- chosen by vmcpde_flow_eval
// when the LDS plan lets more than twelve waves be resident per CU, which the uncapped kernel's 148 VGPRs would not (C2: four
// 4-wave workgroups per CU, Eloc + J 0.645 -> 0.595 ms; where LDS allows no more than twelve waves anyway the uncapped kernel is
// faster -- C3 2.14 against 2.39 ms; tools/diag/flow_plan_ab.sh).  (HIP's second launch-bound is waves per SIMD, not blocks per CU.)
template <int KD, bool CAP = false>
__global__ __launch_bounds__(64 * MAX_WAVES, CAP ? 4 : 1) void flow_eval_kernel(const FlowProgram* __restrict__ fp, const LatentCache* __restrict__ lc,
                                                      const double* __restrict__ th, const double* __restrict__ x,
                                                      int64_t N, const double* __restrict__ dirs, int n_dirs,
                                                      double* __restrict__ logp, double* __restrict__ g,
                                                      double* __restrict__ hv, void* __restrict__ JT, int64_t ldj, int jt_f32,
                                                      double* __restrict__ actw, int64_t ldw, int lanes, Regions r) {
  extern __shared__ double lds[];
  Ctx c;
  c.lds = lds;
  c.wl = lds + (size_t)r.total * lanes;
  c.lanes = lanes;
  c.wave = (int)threadIdx.x >> 6;
  c.nw = (int)blockDim.x >> 6;
  c.th = th;
  c.fp = fp;
  c.jt_f32 = jt_f32;
  const bool reverse = (g != nullptr || JT != nullptr);
  c.actw = reverse ? actw : nullptr;      // nothing to keep if no reverse sweep follows
  c.ldw = ldw;
  // lanes beyond `lanes` (narrow workgroups of very wide networks) shadow lane 0: same sample, same LDS slots, same
  // values, in lockstep with it inside the wave -- they must stay alive for the workgroup barriers
  const int lane = (int)threadIdx.x & 63;
  const bool lane_ok = lane < lanes;
  c.lane = lane_ok ? lane : 0;
  const int d = fp->dim;
  const int64_t n = (int64_t)blockIdx.x * lanes + c.lane;
  const bool active = lane_ok && n < N;
  const int64_t nn = (n < N) ? n : (N - 1);
  c.joff = (uint32_t)n * (jt_f32 ? 4u : 8u);
  c.jpitch = ldj * (jt_f32 ? 4 : 8);
  c.samp_st = n;
  c.samp_ld = nn;
  c.act_store = false;
#ifdef VMC_FLOW_STAMPS
  FlowStamps stamps;
  for (int i = 0; i < 10; ++i) stamps.acc[i] = 0;
  stamps.last = __builtin_readcyclecounter();
  c.st = (c.wave == 0) ? &stamps : nullptr;
#endif

  if (fp->stage_all) stage_all_networks(c);   // (the first network pass starts with a barrier)

  // wave 0 owns the per-sample state and the scalar results; all waves walk the same sequence of stages
  double lp_val = 0.0;
  if (n_dirs > 0 && hv != nullptr) {
    constexpr int NCT = 1 + 2 * KD;
    for (int k = 0; k < n_dirs; k += KD) {
      c.act_store = active && c.actw != nullptr && k == 0;       // the value component is the same for every direction
      if (W0)
        for (int i = 0; i < d; ++i) {
          LD(r.X + i) = x[nn * d + i];
#pragma unroll
          for (int kd = 0; kd < KD; ++kd) {                       // (a group that sticks out of n_dirs carries zero directions)
            LD(r.X + (2 * kd + 1) * d + i) = (k + kd < n_dirs) ? dirs[(k + kd) * d + i] : 0.0;
            LD(r.X + (2 * kd + 2) * d + i) = 0.0;
          }
        }
      double lj[NCT], lp[NCT];
#pragma unroll
      for (int cc = 0; cc < NCT; ++cc) lj[cc] = 0.0;
      for (int s = 0; s < fp->n_half; ++s) {
        halfstep_forward<NCT>(c, fp->hs[s], r, lj);
        if ((s & 1) && fp->global_change) global_forward<NCT>(c, s >> 1, r, lj);
      }
      if (W0) {
        FSTAMP(8);
        latent_forward<NCT>(c, lc, r, r.X, lp);
        lp_val = lp[0] + lj[0];
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
          if (active && k + kd < n_dirs) hv[n * n_dirs + k + kd] = lp[2 * kd + 2] + lj[2 * kd + 2];
        FSTAMP(3);
      }
    }
  } else {
    c.act_store = active && c.actw != nullptr;
    if (W0)
      for (int i = 0; i < d; ++i) LD(r.X + i) = x[nn * d + i];
    double lj[1] = {0}, lp[1];
    for (int s = 0; s < fp->n_half; ++s) {
      halfstep_forward<1>(c, fp->hs[s], r, lj);
      if ((s & 1) && fp->global_change) global_forward<1>(c, s >> 1, r, lj);
    }
    if (W0) {
      latent_forward<1>(c, lc, r, r.X, lp);
      lp_val = lp[0] + lj[0];
    }
  }
  if (logp != nullptr && active && W0) logp[n] = lp_val;
  if (!reverse) return;
  c.act_store = false;
  if (c.actw) {          // the stores above must be visible to the loads of the reverse sweep (other waves of this workgroup)
    __threadfence_block();
    SYNC();
  }

  // reverse sweep: X currently holds z
  FSTAMP(8);
  if (W0) latent_backward(c, lc, r, JT, ldj, n, active);
  FSTAMP(4);
  for (int s = fp->n_half - 1; s >= 0; --s) {
    if ((s & 1) && fp->global_change) global_backward(c, s >> 1, r, JT, ldj, n, active);
    halfstep_backward(c, fp->hs[s], r, JT, ldj, n, active);
  }
  if (g != nullptr && active && W0)
    for (int i = 0; i < d; ++i) g[n * d + i] = LD(r.G + i);
#ifdef VMC_FLOW_STAMPS
  FSTAMP(8);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 9; ++i) atomicAdd(&g_flow_stamps[i], stamps.acc[i]);
    atomicAdd(&g_flow_stamps[9], 1ull);
  }
#endif
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

// var_state.py:33 / net.py:214-217: (y, latent(x_in - offset) - log_jac) for either direction.
__global__ __launch_bounds__(64 * MAX_WAVES, 4) void flow_sample_kernel(const FlowProgram* __restrict__ fp, const LatentCache* __restrict__ lc,
                                                        const double* __restrict__ th, const double* __restrict__ zin,
                                                        int64_t N, int inv, double* __restrict__ xout,
                                                        double* __restrict__ logp, int lanes, Regions r) {
  extern __shared__ double lds[];
  Ctx c;
  c.lds = lds;
  c.wl = lds + (size_t)r.total * lanes;
  c.lanes = lanes;
  c.wave = (int)threadIdx.x >> 6;
  c.nw = (int)blockDim.x >> 6;
  c.th = th;
  c.fp = fp;
  c.jt_f32 = 0;
  c.actw = nullptr;
  c.ldw = 0;
  c.samp_st = c.samp_ld = 0;
  c.act_store = false;
#ifdef VMC_FLOW_STAMPS
  c.st = nullptr;
#endif
  const int lane = (int)threadIdx.x & 63;
  const bool lane_ok = lane < lanes;
  c.lane = lane_ok ? lane : 0;            // see flow_eval_kernel
  const int d = fp->dim;
  const int64_t n = (int64_t)blockIdx.x * lanes + c.lane;
  const bool active = lane_ok && n < N && W0;
  const int64_t nn = (n < N) ? n : (N - 1);
  if (fp->stage_all) stage_all_networks(c);   // (the first network pass starts with a barrier)
  double lp[1] = {0.0};
  if (W0) {
    for (int i = 0; i < d; ++i) LD(r.X + i) = zin[nn * d + i];
    latent_forward<1>(c, lc, r, r.X, lp);   // density of the INPUT point (net.py:215)
  }
  double lj = 0.0;
  if (inv) {
    for (int s = fp->n_half - 1; s >= 0; --s) {
      halfstep_inverse(c, fp->hs[s], r, lj);
      if (!(s & 1) && fp->global_change) global_inverse_as_written(c, s >> 1, r, lj);
    }
  } else {
    double ljv[1] = {0};
    for (int s = 0; s < fp->n_half; ++s) {
      halfstep_forward<1>(c, fp->hs[s], r, ljv);
      if ((s & 1) && fp->global_change) global_forward<1>(c, s >> 1, r, ljv);
    }
    lj = ljv[0];
  }
  if (active) {
    for (int i = 0; i < d; ++i) xout[n * d + i] = LD(r.X + i);
    if (logp != nullptr) logp[n] = lp[0] - lj;   // net.py:217
  }
}

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
      // v = kron(I,[[0,1],[-1,0]]) grad H, H = m w^2/2 sum x^2 + sum p^2/(2m) + lam sum x^4 (:30-45, uncoupled)
      double adv = 0.0;
      for (int i = 0; i + 1 < d; i += 2) {
        const double q = xn[i], p = xn[i + 1];
        const double dHdq = c.m * c.omega * c.omega * q + 4.0 * c.lam * q * q * q;
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
int pick_waves(const FlowProgram& fp) {
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
  Regions r = make_regions(m->host);
  int bytes = 0;
  int l = pick_lanes(r.total, r.wgt, pick_waves(m->host), &bytes);
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

extern "C" int vmcpde_flow_eval(vmcpde_model* m, const double* theta, const double* x, int64_t n_samples,
                                const double* dirs, int32_t n_dirs, double* logp, double* g, double* hv,
                                void* JT, int64_t ldj, int32_t jt_f32, void* workspace, int64_t workspace_bytes, void* stream) {
  VMC_REQUIRE(n_samples >= 0, "negative sample count");
  if (n_samples == 0) return 0;   // empty batch (torch gives a NULL data pointer for it)
  VMC_REQUIRE(m && theta && x, "vmcpde_flow_eval: null argument");
  VMC_REQUIRE(n_dirs >= 0 && n_dirs <= VMCPDE_MAX_DIRS, "n_dirs %d outside [0,%d]", n_dirs, VMCPDE_MAX_DIRS);
  VMC_REQUIRE(n_dirs == 0 || (dirs && hv), "directions given without dirs/hv buffers");
  VMC_REQUIRE(JT == nullptr || ldj >= n_samples, "ldj %lld < n_samples %lld", (long long)ldj, (long long)n_samples);
  if (n_samples == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  // Hessian directions per forward pass: 4 or 2 when every workgroup of the launch is still resident at once with the larger
  // per-sample LDS footprint (narrow networks, moderate batches -- the reference's own problem sizes), else 1
  int kd = 1;
  Regions r = make_regions(m->host, 1);
  int bytes = 0;
  int lanes = pick_lanes(r.total, r.wgt, pick_waves(m->host), &bytes);
  VMC_REQUIRE(bytes <= 160 * 1024, "network too wide for the LDS-resident flow kernel (%d B per sample lane)", r.total * 8);
  if (hv != nullptr && n_dirs >= 2 && env_int("VMCPDE_FLOW_MULTIDIR", 1)) {
    for (int cand : {4, 2}) {
      if (cand == 4 && n_dirs < 3) continue;
      const Regions rc = make_regions(m->host, cand);
      int bc = 0;
      const int lc_ = pick_lanes(rc.total, rc.wgt, pick_waves(m->host), &bc);
      if (bc > 160 * 1024 || lc_ != lanes) continue;              // (same samples per workgroup as the one-direction plan)
      const int64_t resident = 256 * (int64_t)((160 * 1024) / bc);
      if ((n_samples + lc_ - 1) / lc_ <= resident) {
        kd = cand; r = rc; bytes = bc; lanes = lc_;
        break;
      }
    }
  }
  const int waves = pick_waves(m->host);
  // more than twelve waves per CU possible by LDS and by the 16-wave budget of a 128-VGPR kernel -> the capped instantiation
  const int wg_lds = (160 * 1024) / bytes, wg_cap = 16 / waves;
  const bool cap = (kd == 1) && ((wg_lds < wg_cap ? wg_lds : wg_cap) * waves > 12) && env_int("VMCPDE_FLOW_CAP", 1);
  const void* kern = (kd == 4) ? (const void*)flow_eval_kernel<4> : (kd == 2) ? (const void*)flow_eval_kernel<2>
                     : cap ? (const void*)flow_eval_kernel<1, true> : (const void*)flow_eval_kernel<1>;
  if (bytes > 64 * 1024) {
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return -2;
    DeviceCtxLock lock;
    int& have = (kd == 4) ? ctx->flow_eval_lds4 : (kd == 2) ? ctx->flow_eval_lds2 : cap ? ctx->flow_eval_lds_cap : ctx->flow_eval_lds;
    if (bytes > have) {
      VMC_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      have = bytes;
    }
  }
  const int64_t grid = (n_samples + lanes - 1) / lanes;
  VMC_REQUIRE(workspace == nullptr || workspace_bytes >= vmcpde_flow_eval_workspace_bytes(m, n_samples),
              "flow_eval workspace too small: need %lld bytes", (long long)vmcpde_flow_eval_workspace_bytes(m, n_samples));
#define VMC_LAUNCH_FLOW_EVAL(...)                                                                                                        \
  hipLaunchKernelGGL((flow_eval_kernel<__VA_ARGS__>), dim3((unsigned)grid), dim3(64 * waves), bytes, st, m->dev, m->cache, theta, x, \
                     n_samples, dirs, (hv ? n_dirs : 0), logp, g, hv, JT, ldj, (int)jt_f32, static_cast<double*>(workspace),             \
                     (int64_t)((n_samples + 63) / 64 * 64), lanes, r)
  if (kd == 4) VMC_LAUNCH_FLOW_EVAL(4);
  else if (kd == 2) VMC_LAUNCH_FLOW_EVAL(2);
  else if (cap) VMC_LAUNCH_FLOW_EVAL(1, true);
  else VMC_LAUNCH_FLOW_EVAL(1);
#undef VMC_LAUNCH_FLOW_EVAL
  VMC_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vmcpde_flow_sample(vmcpde_model* m, const double* theta, const double* z, int64_t n_samples, int inv,
                                  double* x, double* logp, void* stream) {
  if (n_samples <= 0) return 0;
  VMC_REQUIRE(m && theta && z && x, "vmcpde_flow_sample: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = vmcpde_prepare_latent_internal(m, theta, st)) return rc;
  Regions r = make_regions_values(m->host);
  int bytes = 0;
  const int lanes = pick_lanes(r.total, r.wgt, pick_waves(m->host), &bytes);
  VMC_REQUIRE(bytes <= 160 * 1024, "network too wide for the LDS-resident flow kernel");
  if (bytes > 64 * 1024) {
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return -2;
    DeviceCtxLock lock;
    if (bytes > ctx->flow_sample_lds) {
      VMC_CHECK_HIP(hipFuncSetAttribute((const void*)flow_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      ctx->flow_sample_lds = bytes;
    }
  }
  const int64_t grid = (n_samples + lanes - 1) / lanes;
  hipLaunchKernelGGL(flow_sample_kernel, dim3((unsigned)grid), dim3(64 * pick_waves(m->host)), bytes, st, m->dev, m->cache, theta, z, n_samples,
                     inv, x, logp, lanes, r);
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
