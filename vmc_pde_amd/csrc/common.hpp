// Shared definitions of libvmcpde (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <string>

#include "../../include/vmcpde.h"

namespace vmcpde {

void set_error(const char* fmt, ...);

#define VMC_CHECK_HIP(expr)                                                                    \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::vmcpde::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                                               \
    }                                                                                          \
  } while (0)

#define VMC_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::vmcpde::set_error(__VA_ARGS__);   \
      return -1;                          \
    }                                     \
  } while (0)

constexpr int MAX_DIM = VMCPDE_MAX_DIM;
constexpr int MAX_HALF = VMCPDE_MAX_DIM;          // a half-step conditions on / transforms at most d coordinates
constexpr int MAX_LAYERS = VMCPDE_MAX_HIDDEN_LAYERS + 1;  // hidden Dense layers + output Dense
constexpr int MAX_HALFSTEPS = 2 * VMCPDE_MAX_DEPTH;

// One nn.Dense of a SingleTrafo (net.py:55-56, 60-61); offsets index the flat parameter vector.
struct DenseLayer {
  int32_t n_in, n_out;
  int32_t off_bias, off_kernel;  // kernel is (in,out) row-major
};

constexpr int STAGE_ALL_MAX_DOUBLES = 1536;   // networks whose weights (all of them, with slack) fit 12 KB of LDS are staged once per kernel

// One SingleTrafo (net.py:44-62): n_layers-1 tanh hidden layers + alpha*tanh output layer.
struct Trafo {
  int32_t n_layers;
  int32_t n_in, n_out;
  int32_t sum_hidden;   // sum of hidden widths
  int32_t max_width;    // max over all layer widths incl. input/output
  int32_t wbase;        // first double of this network in the all-networks weight stage (FlowProgram::stage_all)
  DenseLayer layer[MAX_LAYERS];
};

// One coupling half-step (half of SingleBlock.__call__, net.py:84-118): y_c = c*exp(s(a)) [+ t(a)], a unchanged.
struct HalfStep {
  int32_t n_a, n_c;
  int32_t idx_a[MAX_HALF];  // coordinates fed to the conditioner
  int32_t idx_c[MAX_HALF];  // coordinates that get transformed
  int32_t has_t;
  int32_t act_off_s, act_off_t;   // first unit of this half-step's s / t network in the activation workspace of flow_eval
  Trafo s, t;
};

// Whole INNwProb (net.py:185-217) in forward (real space -> latent space) order.
struct FlowProgram {
  int32_t dim, n_half, coupling, latent;
  int32_t P;
  int32_t off_L, off_Ldiag, off_dist, off_mu;
  int32_t max_width;     // widest MLP layer (incl. inputs/outputs)
  int32_t max_sum_act;   // max over trafos of (n_in + sum_hidden)
  int32_t max_mlp_weights; // max over the MLPs of the doubles of all its layers' kernels and biases (the LDS weight stage)
  int32_t all_mlp_weights; // the same summed over every MLP (each with its zero slack), and
  int32_t stage_all;       //   whether the flow kernels stage ALL networks once at their start (small networks: one round trip to
                           //   theta per kernel instead of one per network pass -- a 64-sample group of the reference's d = 6, hidden (3,)
                           //   model makes 48 passes whose latency, not work, is the kernel's time)
  int32_t act_units;     // activation workspace: sum over all MLPs of (hidden widths + outputs) doubles per sample
  double alpha;          // SingleTrafo.alpha = 10 (net.py:50)
  int32_t global_change; // net.py:72: block b ends with y -> th[off_gscale[b]] * y + th[off_goffset[b] + i]
  int32_t off_goffset[VMCPDE_MAX_DEPTH], off_gscale[VMCPDE_MAX_DEPTH];
  double offset[MAX_DIM];
  HalfStep hs[MAX_HALFSTEPS];
};

// Per-parameter-vector latent cache written by latent_prepare_kernel (flow.hip):
// U = strict-upper(L_para) + diag(exp(L_diag)) (util.py:21-26), C = chol_lower(U U^T), and the Student-t constants.
struct LatentCache {
  double U[MAX_DIM * MAX_DIM];
  double C[MAX_DIM * MAX_DIM];
  double mu[MAX_DIM];
  double logdetS;     // log det(U U^T) = 2 sum L_diag
  double nu;          // exp(dist_params[0]) + 1 (net.py:32)
  double lognorm;     // all x-independent terms of the log pdf
  double dlognorm_dtau;  // d lognorm / d dist_params[0] (Student-t only)
};

// Per-GPU state of the library (side stream + fork/join events of the Cholesky lookahead, "function attributes already
// raised" flags).  HIP function attributes, streams and events belong to one device, so they are keyed by hipGetDevice()
// and created under a lock: a process may drive several GPUs and call from several threads (one stream per thread).
struct DeviceCtx {
  hipStream_t side = nullptr;
  hipStream_t side_masked = nullptr;            // Cholesky lookahead at super-panel level: bulk updates on all but every eighth CU
  int side_masked_state = 0;                    // 0 not tried, 1 available, -1 the CU-masked stream could not be created
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool attr_chol = false, attr_gram = false, attr_gram_f32 = false, attr_gram_dma = false, attr_solve = false, attr_tridiag = false, attr_gram_w32 = false, attr_chol_tall = false;
  bool attr_gram_i8[4] = {false, false, false, false};   // (6 slices x 8 bits, 6 x 7, 7 x 7, 4 x 8) of the int8 experiment
  int flow_eval_lds_by[8][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // largest dynamic-LDS size flow_eval_kernel was raised to, by [block width variant + 2 x fast path + 4 x from-latent][KD = 1, KD = 1 capped, 2, 4]
  int flow_sample_lds_by[4] = {0, 0, 0, 0};
  int jacobi_lds[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // dynamic-LDS size jacobi_step_kernel<rows / 64> was raised to
  std::map<int64_t, int*> gram_perms;           // [dev] Gram tile orders (position -> lower-triangle tile index): key = number of row blocks (whole triangle, built once)
                                                //       or bit 62 | nb << 40 | row_begin << 20 | row_end (a group of tile rows, vmcpde_gram_rows; at most 64 kept)
                                                // per matrix size and GPU and never freed, so alternating sizes / threads never re-synchronise
};
// returns nullptr (and sets the error string) if the current device cannot be queried
DeviceCtx* device_ctx();
struct DeviceCtxLock {   // serialises first-use initialisation on one device
  DeviceCtxLock();
  ~DeviceCtxLock();
};

// integer tuning knob from the environment, read on every call (a getenv is ~100 ns; a cached static would freeze the
// value of whichever call came first in the process)
int env_int(const char* name, int fallback);

}  // namespace vmcpde

struct vmcpde_model {
  vmcpde::FlowProgram host;
  vmcpde::FlowProgram* dev = nullptr;
  vmcpde::LatentCache* cache = nullptr;   // device
  void* layout = nullptr;                 // host-side LayoutTable (model.hip)
  int device = 0;
};
