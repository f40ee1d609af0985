/*
 * vmcpde.h -- C ABI of libvmcpde.so: the MI355X (gfx950) TDVP / stochastic-reconfiguration hot path.
 *
 * The reference (RehMoritz/vmc_pde, /root/reference/vmc_fluids) has no FFI layer: its boundary is the
 * Python object protocol stepper <-> TDVP <-> (VarState, EvolutionEquation, Sampler).  The Python classes in
 * vmc_pde_amd/ keep that protocol; every compiled ("kernel-like") callable of the reference (SURVEY.md 2.2,
 * J1..J15) is replaced by one entry point below.  Each entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - every pointer marked [dev] is a device pointer (HBM); [host] is host memory.  The caller (PyTorch) owns
 *     all buffers including the explicit workspaces; nothing here allocates per call, nothing synchronises the
 *     host, every launch goes to the hipStream_t passed as `stream` (void* so that no HIP header is needed).
 *   - return value 0 = success, <0 = error; vmcpde_last_error() gives a thread-local message.
 *   - all arithmetic is IEEE fp64 (reference: global_defs.py:8 tReal = float64, main.py:2 jax_enable_x64).
 *   - sample arrays are row-major [N][d]; the per-sample parameter Jacobian is stored PARAMETER-MAJOR:
 *     JT[p * ldj + n] = d logp(x_n) / d theta_p  (the reference's (1,N,P) array is a strided view of it).
 */
#ifndef VMCPDE_H
#define VMCPDE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMCPDE_MAX_DIM 16
#define VMCPDE_MAX_DEPTH 16
#define VMCPDE_MAX_HIDDEN_LAYERS 3
#define VMCPDE_MAX_DIRS 136 /* d(d+1)/2 at d = 16 */

#define VMCPDE_GRAM_TILE 128 /* S is produced in 128x128 tiles of the lower triangle */

/* coupling variants of SingleBlock, net.py:69-72, 90-109 */
enum { VMCPDE_COUPLING_NO_ADD = 0, VMCPDE_COUPLING_DIFFERENT_ADD = 1, VMCPDE_COUPLING_JAC_EQ_1 = 2, VMCPDE_COUPLING_PLAIN = 3 };
/* latent densities, net.py:10-36 */
enum { VMCPDE_LATENT_GAUSS = 0, VMCPDE_LATENT_STUDENT_T = 1 };
/* PDE right-hand sides, evolutionEq.py:54-60 */
enum {
  VMCPDE_EQ_DIFFUSION = 0,            /* evolutionEq.py:84-87  */
  VMCPDE_EQ_DIFFUSION_DRIFT = 1,      /* evolutionEq.py:89-94  */
  VMCPDE_EQ_DIFFUSION_ANISOTROPIC = 2,/* evolutionEq.py:96-100 */
  VMCPDE_EQ_ADVECTION_PAPER = 3,      /* evolutionEq.py:102-105 with :23-27 */
  VMCPDE_EQ_ADVECTION_HAMILTONIAN = 4,/* evolutionEq.py:102-105 with :30-45 */
  VMCPDE_EQ_ADVECTION_HAMILTONIAN_WDISS = 5 /* evolutionEq.py:107-119 */
};

typedef struct vmcpde_model vmcpde_model; /* opaque handle: one per (network, GPU) */

/* Static description of net.INNwProb as built by VarState.init_net (var_state.py:110-124, net.py:185-207). */
typedef struct {
  int32_t dim;                                /* d */
  int32_t depth;                              /* number of SingleBlock's (net.py:166) */
  int32_t n_hidden;                           /* len(intmediate) (net.py:45,54) */
  int32_t hidden[VMCPDE_MAX_HIDDEN_LAYERS];   /* intmediate */
  int32_t coupling;                           /* VMCPDE_COUPLING_* */
  int32_t latent;                             /* VMCPDE_LATENT_*   */
  int32_t n_up;                               /* len(ind_up) = dim/2 (var_state.py:116) */
  const int32_t* inds_up;   /* [host] depth * n_up          (var_state.py:116, order preserved) */
  const int32_t* inds_down; /* [host] depth * (dim - n_up)  (var_state.py:117) */
  const double* offset;     /* [host] dim, network_args["offset"] (net.py:192,212) */
  int32_t global_change;    /* net.py:72,80-82: per block a scalar global_scale and a d-vector global_offset (ABI 2) */
} vmcpde_model_desc;

/* constants of one EvolutionEquation (evolutionEq.py:61-77); unused fields are ignored by `kind` */
typedef struct {
  int32_t kind;     /* VMCPDE_EQ_* */
  int32_t coupled;  /* hamiltonian fields: 1 = the ring-coupled H(x, coupled=True) of evolutionEq.py:33-38 (the reference only ever
                       evaluates the default, 0); sits in what was the alignment gap after `kind`, so the layout is that of ABI 3 */
  double D;         /* diffusion, diffusion_drift */
  double mu;        /* diffusion_drift */
  double m, omega, lam, T, gamma; /* hamiltonian fields; T is also the period of advection_paper */
} vmcpde_eq_consts;

const char* vmcpde_last_error(void);
int vmcpde_abi_version(void);

/* ---- model -------------------------------------------------------------------------------------------- */
/* var_state.py:110-124 + :25-27: builds the flat-parameter layout (JAX pytree order = sorted keys) on the GPU. */
int vmcpde_create(const vmcpde_model_desc* desc, vmcpde_model** out);
int vmcpde_destroy(vmcpde_model* m);
int vmcpde_num_parameters(const vmcpde_model* m);                 /* var_state.py:27 */
/* offset of a leaf in the flat vector (var_state.py:94-108). block<0 selects the latent leaves:
 * trafo 0='L',1='L_diag',2='dist_params',3='mu'. Otherwise trafo 0='s1',1='s2',2='t1',3='t2';
 * leaf 0='bias',1='kernel' of Dense_<layer>; trafo 4='global_offset', 5='global_scale' of the block (layer, leaf ignored).
 * Returns -1 if absent. */
int vmcpde_param_offset(const vmcpde_model* m, int block, int trafo, int layer, int leaf);
/* bytes of LDS one wave needs / samples per wave chosen for flow_eval with n_dirs directions (diagnostic) */
int vmcpde_flow_eval_config(const vmcpde_model* m, int n_dirs, int* lanes, int* lds_bytes);

/* ---- sampler.py ---------------------------------------------------------------------------------------- */
/* sampler.py:25-34,72-74,85-86 + util.py:21-26: z_n = mu + chol(L L^T) n_n (* sqrt(nu/chi2_nu)) + offset,
 * Philox4x32-10 counter RNG: sample n of call `stream_id` (< 2^32) uses counter (n, stream_id), key = seed. z: [dev] N*d. */
int vmcpde_sample_latent(vmcpde_model* m, const double* theta, uint64_t seed, uint64_t stream_id,
                         int64_t first_sample, int64_t n_samples, double* z, void* stream);
/* the same map with the random draws injected (parity tests, H4): normals [dev] N*d, chi2 [dev] N or NULL */
int vmcpde_latent_from_normals(vmcpde_model* m, const double* theta, const double* normals, const double* chi2,
                               int64_t n_samples, double* z, void* stream);
/* sampler.py:37-45,88-111 (+ target :20-22): independence-Metropolis on a 2-D disc, one chain per lane.
 * states [dev] chains*2 in/out, out [dev] steps*chains*2 (step-major as the scan emits), n_acc [dev] 1,
 * offset2 [host] 2 (mcmc_info["offset"]). */
int vmcpde_mcmc(uint64_t seed, uint64_t stream_id, int32_t chains, int32_t steps, double bound, const double* offset2,
                double* states, double* out, unsigned long long* n_acc, void* stream);
/* the same scan with injected uniforms u_r,u_phi,u_acc: [dev] steps*chains each */
int vmcpde_mcmc_from_uniforms(const double* u_r, const double* u_phi, const double* u_acc, int32_t chains, int32_t steps,
                              double bound, const double* offset2, double* states, double* out,
                              unsigned long long* n_acc, void* stream);

/* ---- net.py / var_state.py ----------------------------------------------------------------------------- */
/* var_state.py:33,79 (J5): latent z -> (x, logp) = INNwProb(evaluate=False, inv=True) net.py:214-217,176-180,120-153.
 * inv=0 runs the forward direction instead (main.py:81,87,93: apply(..., evaluate=False, inv=False)). */
int vmcpde_flow_sample(vmcpde_model* m, const double* theta, const double* z, int64_t n_samples, int inv,
                       double* x, double* logp, void* stream);
/* var_state.py:29,31,32,34 (J1,J3,J4,J6): per sample logp, grad_x logp, directional second derivatives
 * hv[n][k] = v_k^T (Hessian_x logp) v_k for n_dirs caller-given directions dirs[k][d], and the parameter Jacobian.
 * Any of logp,g,hv,JT may be NULL (skipped).  JT: [dev] P * ldj, parameter-major, fp64 or (jt_f32 != 0) fp32 -- BASELINE
 * config 5 "fp32 Jacobian with fp64 S/F accumulate": only the stored Jacobian is rounded, S and F are accumulated in fp64.
 * dirs: [dev] n_dirs*d.
 * workspace: [dev] >= vmcpde_flow_eval_workspace_bytes(m, n_samples) or NULL.  With a workspace the forward pass keeps every
 * network's hidden activations and outputs there (C2: 344 doubles per sample = 180 MB) and the reverse sweep reads them back;
 * without, it re-evaluates each network (identical results, half of all tanh evaluations more). */
int64_t vmcpde_flow_eval_workspace_bytes(const vmcpde_model* m, int64_t n_samples);
int vmcpde_flow_eval(vmcpde_model* m, const double* theta, const double* x, int64_t n_samples,
                     const double* dirs, int32_t n_dirs, double* logp, double* g, double* hv,
                     void* JT, int64_t ldj, int32_t jt_f32, void* workspace, int64_t workspace_bytes, void* stream);
/* ABI 7: the same evaluation for a batch given by its LATENT points z (what the samplers draw, sampler.py:72-86): x_out ([dev] n_samples * dim) is an
 * OUTPUT -- the inverse flow of z (var_state.py:76-79), produced by the reverse sweep as it walks back from the latent end, where the adjoint of the
 * latent density is known -- and the Taylor pass then runs forward from that x.  One launch instead of vmcpde_flow_sample(inv = 1) + vmcpde_flow_eval;
 * the same numbers to round-off (the sweep divides by exp(s) where the inverse flow multiplies by exp(-s)). */
int vmcpde_flow_eval_from_latent(vmcpde_model* m, const double* theta, const double* z, int64_t n_samples, const double* dirs, int32_t n_dirs,
                                 double* x_out, double* logp, double* g, double* hv, void* JT, int64_t ldj, int32_t jt_f32, void* stream);

/* ---- evolutionEq.py ------------------------------------------------------------------------------------ */
/* evolutionEq.py:84-119 epilogue: Eloc_n from x, g and the directional second derivatives (J14 + eager einsums).
 * hv columns expected: diffusion(_drift): e_0..e_{d-1}; anisotropic: the columns f_k of any factor D = F F^T, and Dmat [dev] d*d;
 * hamiltonian_wDiss: e_1,e_3,...  (momentum axes); advection: none. */
int vmcpde_rhs(const vmcpde_eq_consts* c, int32_t dim, const double* x, const double* g, const double* hv, int32_t n_dirs,
               const double* Dmat, double t, int64_t n_samples, double* eloc, void* stream);

/* ---- tdvp.py:36-52 + mpi_wrapper.py:61-63,21-25 -------------------------------------------------------- */
/* sums[0..3] = sum E, sum E^2, sum |E|, max E over the local samples (tdvp.py:37-39,150). sums: [dev] 4. */
int vmcpde_eloc_stats(const double* eloc, int64_t n_samples, double* sums, void* stream);
/* per parameter row p: rowsum[p] = sum_n (JT[p][n]-c[p]),  rowdot[p] = sum_n (JT[p][n]-c[p])*(E_n-cE)
 * (tdvp.py:41,44-45 in shifted one-pass form; c may be NULL = 0).  cE is read from cE_dev[0] if non-NULL. */
int vmcpde_row_moments(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, const double* eloc,
                       const double* c, const double* cE_dev, double* rowsum, double* rowdot, void* stream);
/* the same pass with rowsq[p] = sum_n (JT[p][n]-c[p])^2 as a third output (fp64 JT): diag S of the matrix-free step */
int vmcpde_row_moments_sq(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const double* eloc, const double* c,
                          const double* cE_dev, double* rowsum, double* rowdot, double* rowsq, void* stream);
/* number of 128x128 lower-triangle tiles / doubles of the packed Gram buffer for P parameters */
int64_t vmcpde_gram_packed_doubles(int32_t P);
/* bytes of split-K workspace vmcpde_gram needs for (P, n_samples) */
int64_t vmcpde_gram_workspace_bytes(int32_t P, int64_t n_samples);
/* the Gram ("syrk") contraction, tdvp.py:46-47 / mpi_wrapper.py:21-25 (J13) on fp64 MFMA:
 * packed[tile(i,j)][r][s] = sum_n w_n (JT[i*128+r][n]-c[..]) (JT[j*128+s][n]-c[..]),  j<=i,  w=NULL -> 1.
 * Diagonal tiles (j == i) define their lower triangle r >= s only; entries above it are unspecified (vmcpde_gram_finalize
 * mirrors them).
 * w = logp gives SExp (tdvp.py:33,47: rows scaled by w, i.e. weight w^2); w = Etilde gives the SNR Gram (tdvp.py:68-70).
 * jt_f32 = 1: JT holds fp32 values (ldj in elements), widened to fp64 while staging; products and sums are fp64.
 * jt_f32 = 2 ("f32 blocks", only with c = w = NULL, otherwise treated as 1): fp32 products are summed in fp32 by the matrix pipe
 *   over blocks of <= 128 samples and the block sums are accumulated in fp64 -- twice the MFMA rate; ldj must be a multiple of 4 and
 *   the row padding [n_samples, ldj) must be zero.  S agrees with the fp64 result on the same fp32 data to ~1e-8 relative, but is
 *   no longer exactly positive semi-definite: factor it only with a relative diagonal shift >= 1e-4. */
int vmcpde_gram(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, const double* c, const double* w,
                double* packed, void* workspace, int64_t workspace_bytes, void* stream);
/* ABI 5: the same sums as vmcpde_gram(JT, ldj, 0, P, n, NULL, NULL, ...) for the tiles of tile rows [row_begin, row_end) only (rows of 128
 * parameters; the packed buffer is tile-row major, so they are one contiguous piece of it): same sample splits, same per-unit arithmetic,
 * so a partition of the tile rows reproduces vmcpde_gram's buffer bit for bit.  Lets a sharded step pass finished row groups to the
 * all-reduce (mpi_wrapper.py:129-163) while the next group is being formed.  Same workspace as vmcpde_gram. */
int vmcpde_gram_rows(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, int32_t row_begin, int32_t row_end, double* packed,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* tdvp.py:45-46,50-51: S[i][j] = packed/Ntot - d_i d_j (full symmetric, both triangles written) with
 * d = rowsum/Ntot + c - c_gram = (global mean) - (the shift vmcpde_gram was given; c_gram NULL = 0, i.e. uncentred Gram),
 * A_shifted = S + diag(diagonal_shift * diag S) + floor_dev[0] * I (optional second copy for the in-place Cholesky, may be NULL;
 * floor_dev: [dev] 1 or NULL = 0 -- the absolute floor svdTol * max diag(S) that vmcpde_tdvp_moments leaves in scal[6]),
 * F = rowdot/Ntot - delta*deltaE, mean = c + delta, with delta = rowsum/Ntot, deltaE = sums[0]/Ntot - cE  (c = the shift
 * vmcpde_row_moments was given, NULL = 0; F and mean may be NULL = skipped).  S: [dev] P*lds, A_shifted: [dev] P*lda.  All
 * inputs may already be all-reduced over ranks (Ntot = global count). */
int vmcpde_gram_finalize(const double* packed, const double* rowsum, const double* rowdot, const double* sums,
                         const double* c, const double* c_gram, const double* cE_dev, int32_t P, double n_total,
                         double diagonal_shift, const double* floor_dev, double* S, int64_t lds, double* A_shifted, int64_t lda,
                         double* F, double* mean, void* stream);

/* ---- the O(N) and O(P) glue of tdvp.py:36-52, 93-94, three launches per right-hand side ---------------------------------- */
/* tdvp.py:37-39,150: sums3 = { sum E, sum E^2, sum |E| } over the local samples, count[0] = n_samples, max_out[0] = max E
 * (count, max_out may be NULL).  If row_e / row_one are given (both or neither), also row_e[n] = E_n - cE_dev[0] and
 * row_one[n] = 1 for n < n_samples: the two rows the fused-moments path appends to the Jacobian, so that the syrk's last tile
 * row carries sum_n O (E - cE) and sum_n O.  workspace: [dev] >= vmcpde_tdvp_prepare_workspace_bytes(), zeroed ONCE by the
 * caller (the kernel returns it zeroed); sums are formed in a fixed order (run-to-run reproducible). */
int64_t vmcpde_tdvp_prepare_workspace_bytes(void);
int vmcpde_tdvp_prepare(const double* eloc, int64_t n_samples, const double* cE_dev, double* row_e, double* row_one, double* sums3,
                        double* count, double* max_out, void* workspace, int64_t workspace_bytes, void* stream);
/* After the all-reduce, everything that is O(P) (tdvp.py:37-41,45 and the diagonal of :46):
 *   bordered != 0: rowsum / rowdot are WRITTEN from the last tile row of a Gram over P + 2 rows (rows P, P+1 = E - cE, 1);
 *   bordered == 0: they are read (vmcpde_row_moments, possibly all-reduced).
 *   F, mean as vmcpde_gram_finalize; diagS[p] = diag(S)_p;
 *   scal[0..6] = ElocMean, ElocMeanAbs, ElocVar, <E^2>, max_p mean_p^2 / diagS_p, max_p diagS_p, svdTol * max_p diagS_p
 *   (maxima over the parameters with mask[p] != 0; mask NULL = all);  scal: [dev] 8.
 *   border_row non-NULL: border_row[0..P) = F, border_row[P] = border_pivot -- row P of the bordered Cholesky matrix. */
int vmcpde_tdvp_moments(const double* packed, int32_t P, int32_t bordered, double* rowsum, double* rowdot, const double* sums3,
                        double n_total, const double* c, const double* c_gram, const double* cE_dev, const double* mask, double* F,
                        double* mean, double* diagS, double* scal, double svdTol, double* border_row, double border_pivot,
                        void* stream);
/* EXPERIMENT (opt-in): the same packed sums as vmcpde_gram(JT, ldj, 0, P, n, c, w, ...) -- tdvp.py:46-47, mpi_wrapper.py:21-25; c
 * (centring shift per parameter) and w (weight per sample) may be NULL -- formed on the INT8 matrix pipe from an error-free slicing of the fp64 Jacobian (csrc/gram_i8.hip): per-row
 * power-of-two scale, n_slices int8 digits of digit_bits bits -- (6, 8) the default; (6, 7), (7, 7) for the record; (4, 8) = 31 bits for
 * an fp32 Jacobian (jt_f32 = 1: BASELINE config 5, every stored value is represented to 2^-31 of its row maximum), 10 digit pairs --, the n_slices (n_slices + 1) / 2 slice-pair products with
 * s + t < n_slices summed exactly in int32 and combined in fp64.  Exact on integer data; on real data the dropped terms are
 * ~1e-14 (row max / row rms)^2 of |S|_F for 6 slices; stats ([dev] 1 double, may be NULL) receives the largest row max / row rms
 * (within a factor two; at most sqrt(n)), which a caller can watch (vmc_pde_amd.experimental.TDVP falls back to the fp64 kernel above a limit computed from what regularises the solve, (0.1 reg / 1.5e-14)^(1/1.3) with reg = max(svdTol, diagonalShift): ~25 at svdTol = 1e-11, 5e4 at the bench's 1e-4; sharded samples decide on the largest figure of any rank).  workspace: [dev] vmcpde_gram_i8_workspace_bytes (row exponents, the
 * n_slices int8 planes = n_slices * P_pad * N bytes, split slabs). */
int64_t vmcpde_gram_i8_workspace_bytes(int32_t P, int64_t n_samples, int32_t n_slices, int32_t digit_bits);
int vmcpde_gram_i8(const void* JT, int64_t ldj, int32_t jt_f32, int32_t P, int64_t n_samples, int32_t n_slices, int32_t digit_bits, const double* c,
                   const double* w, double* packed, double* stats, void* workspace, int64_t workspace_bytes, void* stream);

/* tdvp.py:93-94 from u and S0 u: out[0] = |S u - F| / |F| with S u = S0 u + (diagonal_shift * diagS + floor_dev[0]) u
 * (floor_dev NULL = 0), out[1] = 1 + (u.S0u - 2 F.u) / scal[3], out[2] = info[0] (pinned pivots, NULL = 0).  out: [dev] 4. */
int vmcpde_tdvp_diagnostics(const double* u, const double* S0u, const double* F, const double* diagS, int32_t P, double diagonal_shift,
                            const double* floor_dev, const double* scal, const int32_t* info, double* out, void* stream);

/* tdvp.py:143-150, the observables of one sample set x[n][dim] (row-major, dim <= 16) in two launches:
 *   out = mean[dim] (x1) | covar[dim*dim] (1/n, about the mean) | x3[dim] x4[dim] x5[dim] x6[dim] (central moments) |
 *         entropy = -mean(logp) | max(eloc)          ([dev] dim*dim + 5*dim + 2; logp / eloc NULL: 0 / -inf)
 * eloc has its OWN length n_eloc (ABI 3): with nSamplesObs > nSamplesTDVP the reference re-draws x and logp at nSamplesObs
 * (tdvp.py:130-134) while jnp.max(Eloc) (tdvp.py:150) still runs over the nSamplesTDVP entries of the solve.
 * workspace: vmcpde_observables_workspace_bytes(), zero-initialised once by the caller (the kernels leave it zeroed). */
int64_t vmcpde_observables_workspace_bytes(void);
int vmcpde_observables(const double* x, int64_t n_samples, int32_t dim, const double* logp, const double* eloc, int64_t n_eloc,
                       double* out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- mpi_wrapper.py:129-163 (global_sum) ---------------------------------------------------------------------- */
/* The single collective of a TDVP step: packed[0..count) <- sum over the ranks of `comm`, in place, fp64, on `stream`
 * (ncclAllReduce of RCCL over xGMI; replaces the reference's pmap psum + host-buffer MPI.Allreduce, ten calls per step).
 * `packed` is the buffer vmcpde_gram filled followed by the row sums, E~O sums, Eloc sums and the sample count
 * (everything vmcpde_gram_finalize consumes), so one call per step reduces all of them.
 * comm: an ncclComm_t of the RCCL loaded in this process -- either the caller's own, or one made with the three helpers
 * below (rank 0 obtains a 128-byte id, distributes it by any means -- mpi4py bcast in the reference's setting --, and every
 * rank calls vmcpde_comm_init_rank with its GPU current).  RCCL is resolved at run time; single-GPU use never touches it. */
int vmcpde_allreduce(double* packed, int64_t count, void* comm, void* stream);
int vmcpde_comm_unique_id(void* id128 /* [host] 128 bytes out */);
int vmcpde_comm_init_rank(void** comm, int32_t n_ranks, const void* id128, int32_t rank);
int vmcpde_comm_destroy(void* comm);
/* What a multi-GPU record must be able to show (ABI 5): that the communicator has the size the job claims and that every rank
 * sits on a device of its own (global_defs.py:16-24: the reference's rank -> device map).
 * vmcpde_comm_info: out4[0] = ncclCommCount, out4[1] = ncclCommUserRank, out4[2] = the HIP device the communicator is bound to
 * (ncclCommCuDevice), out4[3] = the RCCL version code (ncclGetVersion); out4: [host] 4 x int32.
 * vmcpde_device_bus_id: the PCI bus id ("0000:c5:00.0") of the calling thread's current HIP device, NUL-terminated, into
 * out[0..len) ([host], len >= 16). */
int vmcpde_comm_info(void* comm, int32_t* out4);
int vmcpde_device_bus_id(char* out, int32_t len);

/* diagnostic (no reference counterpart): `blocks` workgroups x 4 waves x `iters` x 8 back-to-back v_mfma_f64_16x16x4_f64
 * (2048 flop each) with operands in registers; out: [dev] blocks*256.  Times the fp64 matrix-pipe peak for the roofline. */
int vmcpde_bench_mfma_f64(int32_t iters, int32_t blocks, double* out, void* stream);

/* ---- tdvp.py:57-94 solve ------------------------------------------------------------------------------- */
/* in-place lower Cholesky of A (P x P, leading dim lda); rows with a non-positive pivot are pinned to the unit
 * row (H7, dead/padded parameters).  info: [dev] 1, number of pinned pivots.  workspace >= vmcpde_chol_workspace_bytes */
int64_t vmcpde_chol_workspace_bytes(int32_t P);
int vmcpde_cholesky(double* A, int64_t lda, int32_t P, int32_t* info, void* workspace, int64_t workspace_bytes, void* stream);
/* u = (L L^T)^{-1} F with the factor from vmcpde_cholesky. u may alias F. */
int vmcpde_chol_solve(const double* L, int64_t lda, int32_t P, const double* F, double* u,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* second half only: u = L^-T y.  Used with the bordered factorisation: the Cholesky of [[A, F], [F^T, beta]] (size P+1,
 * beta large) carries y = L^-1 F in its last row, so the forward substitution costs no launches of its own; the
 * workspace may be the one filled by that (P+1)-sized factorisation. */
int vmcpde_chol_backsolve(const double* L, int64_t lda, int32_t P, const double* y, double* u,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* y = A x for symmetric full-storage A (tdvp.py:66,91,93,94 mat-vecs); trans=1 uses A^T (V^T F) */
int vmcpde_gemv(const double* A, int64_t lda, int32_t rows, int32_t cols, int trans, const double* x, double* y, void* stream);
/* tdvp.py:82-91: coef_k = invEv_k * regularizer_k * VtF_k with the hard 1e-14 and soft svdTol cut-offs,
 * optional SNR factor (snr NULL = off).  ev ascending (eigh). */
int vmcpde_pinv_coeffs(const double* ev, const double* VtF, const double* snr, int32_t P, double svdTol, double snrTol,
                       double* coef, void* stream);

/* ---- tdvp.py:57-64 (eigh of S), the dependent chains of the rank-revealing decomposition (csrc/eig.hip) --------------------------
 * The reference diagonalises the full P x P matrix.  S = <J^T J> of these models is numerically rank deficient (238 of 4213
 * eigenvalues above 1e-16 of the largest at the headline size), and a full decomposition [null-space basis | Ritz vectors] follows
 * from a pivoted Cholesky factor of S and a Householder QR of that factor in O(P r^2) (vmc_pde_amd/lowrank.py makes the GEMMs).
 *
 * vmcpde_pivoted_cholesky: steps [first_step, first_step + n_steps) of S = L L^T + R with diagonal pivoting (dpstrf's algorithm,
 * unblocked, no row interchanges: rows keep their place, L[piv[j]][k] = 0 for j < k).  S: [dev] symmetric, full storage, pitch lds
 * (row p is read as column p).  LT: [dev] max_rank x ldl, LT[k][i] = L[i][k], ldl = P rounded up to a multiple of 64 (pad columns are
 * written as zeros).  piv: [dev] max_rank pivot rows.  state: [dev] 2 x int32 -- state[0] = columns completed, state[1] = 1 once the
 * largest remaining diagonal entry was <= tol_rel * max diag(S) (or not a positive number): further steps return at once.
 * first_step = 0 starts a new factorisation; later calls continue it (same arguments and workspace), so the host can run a chunk,
 * read `state`, and go on.  Workspace head, two doubles: max diag(S), and the sum of the positive remaining diagonal entries (the
 * trace of R) after the last completed step. */
int64_t vmcpde_pivoted_cholesky_workspace_bytes(int32_t P);
int vmcpde_pivoted_cholesky(const double* S, int64_t lds, int32_t P, double tol_rel, int32_t first_step, int32_t n_steps,
                            int32_t max_rank, double* LT, int64_t ldl, int32_t* piv, int32_t* state, void* workspace,
                            int64_t workspace_bytes, void* stream);
/* The same factorisation with up to 32 pivots per pair of launches (any pivot order gives S = L L^T + R with R >= 0; candidates are
 * the largest of the two largest remaining diagonal entries of every 64-row block, none below 1/100 of the largest -- a pivot p turns
 * round-off n into n / sqrt(p) --, with pivoting inside the 32 x 32 block; candidates that fall below that bound there stay open).
 * P <= 32768.  Same buffers and `state` as above (its own workspace size); `restart` != 0 starts a new
 * factorisation, then n_blocks block steps run, then one settling launch updates state[1] and the trace of the remainder.
 * state[0] == max_rank with state[1] == 0: the rank cap was reached before the criterion. */
int64_t vmcpde_pivoted_cholesky_blocked_workspace_bytes(int32_t P);
int vmcpde_pivoted_cholesky_blocked(const double* S, int64_t lds, int32_t P, double tol_rel, int32_t restart, int32_t n_blocks,
                                    int32_t max_rank, double* LT, int64_t ldl, int32_t* piv, int32_t* state, void* workspace,
                                    int64_t workspace_bytes, void* stream);
/* Pivot hints for the blocked factorisation (and its matrix-free form below): the pivots piv[0 .. state[0]) of a FINISHED
 * factorisation of a nearby matrix -- the previous TDVP step's S; the reference's loop solves a slowly moving sequence of them,
 * tdvp.py:57-94 once per right-hand side -- become the preferred candidates of the factorisations started afterwards in this
 * workspace: a block step takes the first 32 open entries of the list whose remaining diagonal entry qualifies NOW, always together
 * with the largest remaining entry, and falls back to the usual choice when the list runs out.  Nothing else changes (pivoting inside
 * the block, the 1/100 window, the stopping criterion), so S = L L^T + R holds as before whatever the list contains; what changes is how
 * many candidates survive a block (~13 of 32 chosen by the diagonal alone, most of a hinted block) and with it the number of dependent
 * block steps.  The list is copied into the workspace (piv may be reused at once) and stays armed until replaced; piv == NULL disarms. */
int vmcpde_lowrank_hint(const int32_t* piv, const int32_t* state, int32_t P, void* workspace, int64_t workspace_bytes, void* stream);
/* ---- EXPERIMENT: the same factorisation without an assembled S (matrix-free TDVP step, vmc_pde_amd/experimental/matrix_free.py) -------
 * S = J^T J / N - m m^T enters the pivoted factorisation only through its diagonal and the 32 columns a block step pivots on, so the
 * Jacobian is read once per block step instead of N P^2 flop:
 *   vmcpde_lowrank_begin(diag)        start from the diagonal of S ([dev] P); same workspace / state as the blocked factorisation
 *   vmcpde_lowrank_step(NULL, ..)     choose the step's candidates: their rows are the int32[32] at
 *                                     vmcpde_lowrank_candidates_byte_offset(P) in the workspace (unused entries repeat a valid row)
 *   vmcpde_gram_columns(..)           Scol[m][i] = sum_n JT[rows[m]][n] JT[i][n] / n_total - mean[i] mean[rows[m]]   ([32][ldl];
 *                                     an fp64-MFMA skinny GEMM, one pass over JT; mean NULL = uncentred)
 *   vmcpde_lowrank_step(Scol, ..)     panel, block factorisation with pivoting, new columns of L (as launches 1-2 of a blocked step) */
/* With a hint list armed (vmcpde_lowrank_hint) the columns of S at its first n_rows rows can be formed AHEAD of the factorisation in one
 * pass over the Jacobian (a gather of those rows and one GEMM, the caller's: Sbig[j][i] = S[hint[j]][i], [n_rows][ldl]) and registered
 * after vmcpde_lowrank_begin: a block step then takes list entries only while any qualifies and reads their columns from Sbig -- no
 * pass for that step -- and is an ordinary step when none does.  The int32[2] at vmcpde_lowrank_skip_flag_byte_offset() in the
 * workspace tells vmcpde_gram_columns (as its `state` argument) whether the step just selected needs its columns formed. */
int vmcpde_lowrank_columns(const double* Sbig, int32_t n_rows, int32_t P, void* workspace, int64_t workspace_bytes, void* stream);
int64_t vmcpde_lowrank_skip_flag_byte_offset(void);
int64_t vmcpde_lowrank_candidates_byte_offset(int32_t P);
int vmcpde_lowrank_begin(const double* diag, int32_t P, int32_t* state, void* workspace, int64_t workspace_bytes, void* stream);
int vmcpde_lowrank_step(const double* Scol, int32_t P, double tol_rel, int32_t max_rank, double* LT, int64_t ldl, int32_t* piv,
                        int32_t* state, void* workspace, int64_t workspace_bytes, void* stream);
/* y = JT (JT^T v) over the P x n_samples Jacobian block (row stride ldj): the product S0 v = y / N - m (m.v) of the matrix-free step's
 * refinement and diagnostics, two HBM-bound passes over the Jacobian; z ([dev] n_samples) receives JT^T v.  y == NULL: the first
 * pass only -- what a quadratic form v^T <w^2 O~ O~^T> v needs (the SExp norm of the adaptive stepper, stepper.py:71 with main.py:24-26). */
int64_t vmcpde_jt_matvec_workspace_bytes(int32_t P, int64_t n_samples);
int vmcpde_jt_matvec(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const double* v, double* z, double* y, void* workspace,
                     int64_t workspace_bytes, void* stream);
int64_t vmcpde_gram_columns_workspace_bytes(int32_t P, int64_t n_samples);
int vmcpde_gram_columns(const double* JT, int64_t ldj, int32_t P, int64_t n_samples, const int32_t* rows32, const double* mean,
                        double n_total, double* Scol, int64_t ldl, const int32_t* state /* NULL, or: skip when state[1] != 0 */,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* Householder QR of the P x r matrix whose columns are the rows of LT (destroyed: the strict upper part of R, the rest is scratch).  YT: [dev] r x ldl,
 * row k = reflector v_k (zeros before k, v_k[k] = 1, zeros from P on); tau: [dev] r, H_k = I - tau_k v_k v_k^T, Q = H_0 ... H_{r-1}
 * = I - Y T Y^T with T^-1 = strict upper part of Y^T Y + diag(1 / tau)  (tau_k is never 0: a column that is already zero below its
 * diagonal gets v = e_k, tau = 2).  Two reflectors per launch while their columns fit the registers of a workgroup (P <= 16 384), one
 * otherwise.  scratch: [dev] 5 r + 5 doubles. */
int vmcpde_householder_rows(double* LT, int64_t ldl, int32_t P, int32_t r, double* YT, double* tau, double* scratch, void* stream);
/* ABI 5: the Woodbury application of the low-rank regularised solve, M^-1 v for M = D + L L^T (D > 0 diagonal, L = LT^T of rank r):
 *   M^-1 v = y - D^-1 L G^-1 L^T y,  y = D^-1 v,  G = I + L^T D^-1 L  (r x r, factored by the caller: vmcpde_cholesky / vmcpde_chol_solve).
 * project: y = v / D and t1 = L^T y, with v = F (u, Su NULL) or the residual F - Su - D.u of the full system (S0 + D) u = F at u
 *          (Su = S0 u from vmcpde_gemv) -- the right-hand side of a refinement step;   y: [dev] P,  t1: [dev] r.
 * combine: out = (u ? u : 0) + y - D^-1 L t2  with t2 = G^-1 t1;  out may alias u.  Replaces two library gemv and six elementwise
 *          launches per application (tdvp.py:91 is one matrix-vector product; this is its counterpart on the factor). */
int vmcpde_woodbury_project(const double* LT, int64_t ldl, int32_t P, int32_t r, const double* D, const double* F, const double* Su,
                            const double* u, double* y, double* t1, void* stream);
int vmcpde_woodbury_combine(const double* LT, int64_t ldl, int32_t P, int32_t r, const double* D, const double* y, const double* t2,
                            const double* u, double* out, void* stream);
/* ABI 7: what surrounds that application in a call on a static plan, so that the replicated part of a step holds no library GEMM, no framework
 * elementwise launch and one device-to-host copy (tdvp.py:91-94 are three array expressions in the reference).
 * prepare: D[p] = shift * diagS[p] + floor_dev[0] (floor_dev NULL: 0) for p < P, invD[p] = 1 / D[p] and 0 on the pad [P, ldl); rows k in
 *          [state[0], rmax) of LT are zeroed (state[0] = the rank the factorisation just reached, read on the device: those rows still hold
 *          an earlier call's columns and must be an identity block of G).  D: [dev] >= P, invD: [dev] ldl, LT: [dev] >= rmax x ldl.
 * gram:    G = I + LT[:r] diag(invD) LT[:r]^T (r x r, both triangles; r a multiple of 64, ldl a multiple of 32) on the fp64 matrix pipe: 64 x 64
 *          blocks of the lower triangle x chunks of 256 parameters as independent workgroups, chunk sums added in a fixed order (the same
 *          bits on every rank).  workspace: vmcpde_woodbury_gram_workspace_bytes(ldl, r).
 * gather_words: dst = the concatenation of n_src (<= 8) device arrays of counts[s] 8-byte words -- the health records of a call (residual
 *          and pinned pivots, the factorisation's state and step count, the mean^2 / var figure) as ONE buffer for ONE asynchronous copy.
 *          srcs / counts are HOST arrays (read during the call). */
int vmcpde_woodbury_prepare(const double* diagS, double shift, const double* floor_dev, int32_t P, int64_t ldl, double* D, double* invD,
                            double* LT, const int32_t* state, int32_t rmax, void* stream);
int64_t vmcpde_woodbury_gram_workspace_bytes(int64_t ldl, int32_t r);
int vmcpde_woodbury_gram(const double* LT, int64_t ldl, int32_t r, const double* invD, double* G, int64_t ldg, void* workspace,
                         int64_t workspace_bytes, void* stream);
int vmcpde_gather_words(const void* const* srcs, const int32_t* counts, int32_t n_src, void* dst, void* stream);
/* ABI 7: the factor of S on a FIXED pivot set, without the pivot search (what a call on a static plan runs while the previous factorisation's pivots
 * still span S: consecutive S of a trajectory differ by sampling noise and a small step of theta).  With Pi = piv[0 .. r) in that order,
 *   S = L L^T + R,  L = S[:, Pi] C^-T,  C C^T = S[Pi, Pi]   (the partial Cholesky on that pivot set; R = the Schur complement, positive semi-definite).
 * gather:  T = [ S[Pi, Pi] ; S[:, Pi] ] as a tall (rmax + P) x rmax matrix (identity / zero columns beyond r); T: [dev] rows allocated to a multiple
 *          of 128, ldt >= rmax.
 * vmcpde_cholesky_tall(T, ldt, rmax + P, rmax, ...): Cholesky of the leading ncols x ncols block and the panel solve of all rows below it, with
 *          the launches of vmcpde_cholesky (workspace: vmcpde_chol_workspace_bytes(ncols); info counts pinned pivots).
 * finish:  LT[b][i] = T[rmax + i][b] (the layout every consumer of the factor reads; pad columns zero) and left[i] = S[i][i] - |L[i][:]|^2 = R_ii.
 * check:   out = { trace(R) / min D, pinned pivots, max_i R_ii }, and the bookkeeping a pivoted run leaves: state = (r, 1), the trace in the head of the
 *          factorisation's workspace.  The host reads it with the call's health record and decides ONE CALL LATER whether the set still holds. */
int vmcpde_frozen_gather(const double* S, int64_t lds, int32_t P, const int32_t* piv, int32_t r, int32_t rmax, double* T, int64_t ldt, void* stream);
int vmcpde_cholesky_tall(double* A, int64_t lda, int32_t nrows, int32_t ncols, int32_t* info, void* workspace, int64_t workspace_bytes, void* stream);
int vmcpde_frozen_finish(const double* T, int64_t ldt, const double* S, int64_t lds, int32_t P, int32_t rmax, double* LT, int64_t ldl, double* left,
                         void* stream);
int vmcpde_frozen_check(const double* left, int32_t P, const double* D, int32_t r, const int32_t* chol_info, int32_t* state, void* workspace, double* out,
                        void* stream);
/* ABI 5: the r x r core of that decomposition (tdvp.py:57-64 on range(L)) without a library eigensolver.  One-sided Jacobi on the columns
 * of an r x r matrix C (r <= 1024): on return the columns are mutually orthogonal, C_out = U diag(sigma) with U the left singular vectors and
 * sigma the singular values of the input -- for C = R1 (the triangular factor of L = Q1 R1) the eigenvectors and the square roots of the
 * eigenvalues of R1 R1^T = Q1^T (L L^T) Q1, in no particular order.  CT: [dev] cols_pad x rows_pad doubles, row c = column c of C, zero
 * padded (vmcpde_jacobi_shape gives the padded sizes: rows to a multiple of 64, columns to a multiple of the columns a workgroup sweeps).
 * n_sweeps (<= 30) sweeps are launched; launches behind a sweep that rotated nothing return at once.  counters: [dev] n_sweeps + 1 int32,
 * counters[s] = rotations of sweep s -- converged iff some counters[s] == 0 (the caller may read it later; nothing synchronises). */
int vmcpde_jacobi_shape(int32_t r, int32_t* rows_pad, int32_t* cols_pad);
int vmcpde_jacobi_columns(double* CT, int32_t r, int32_t n_sweeps, int32_t* counters, void* stream);

/* ABI 6: the same r x r core (tdvp.py:57-64 on range(L)) by the classical route, with the dependent part in own kernels.  A library syevd
 * on the 206 x 206 core of the headline workload takes 4.3 ms, 2.9 ms of it the Householder tridiagonalisation -- a chain of ~10 launches
 * per column.  vmcpde_sytrd_small: T = Q tridiag(D, E) Q^T for a symmetric n x n matrix (n <= vmcpde_sytrd_small_max_n() = 256; only the
 * lower triangle of the row-major T is read, T is not modified) in ONE 512-thread workgroup that keeps the lower triangle in registers
 * (LAPACK's dsytd2, lower: Q = H_0 ... H_{n-2}, H_k = I - tau_k v_k v_k^T).  D, E, tau: [dev] n (E[n-1] = tau[n-1] = 0); V: [dev] n x ldv,
 * row k = v_k (zeros up to k, v_k[k+1] = 1).  Deterministic (no atomics, fixed summation orders): a pure function of the input bits.
 * vmcpde_apply_reflectors: Zt[m][:] <- Q Zt[m][:] for the nvec rows of Zt -- the eigenvectors of tridiag(D, E), one per row (the
 * column-major eigenvector matrix of a LAPACK-style tridiagonal solver such as rocSOLVER's stedc, which the host calls between the two),
 * become those of T. */
int32_t vmcpde_sytrd_small_max_n(void);
int vmcpde_sytrd_small(const double* T, int64_t ldt, int32_t n, double* D, double* E, double* tau, double* V, int64_t ldv, void* stream);
int vmcpde_apply_reflectors(const double* V, int64_t ldv, const double* tau, int32_t n, double* Zt, int64_t ldz, int32_t nvec, void* stream);
/* ABI 7: the same decomposition for 256 < n <= vmcpde_sytrd_medium_max_n() = 384 (BASELINE C3's core has rank 263): the leading n - 256 Householder
 * steps run as two launches each on a full symmetric copy of T in `workspace` (vmcpde_sytrd_medium_workspace_bytes(n); 0 for n <= 256, where the
 * call IS vmcpde_sytrd_small), the trailing 256 x 256 block goes through the one-workgroup kernel.  Same outputs, same conventions; deterministic.
 * vmcpde_apply_reflectors takes n <= 384 since ABI 7. */
int32_t vmcpde_sytrd_medium_max_n(void);
int64_t vmcpde_sytrd_medium_workspace_bytes(int32_t n);
int vmcpde_sytrd_medium(const double* T, int64_t ldt, int32_t n, double* D, double* E, double* tau, double* V, int64_t ldv, void* workspace,
                        int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMCPDE_H */
