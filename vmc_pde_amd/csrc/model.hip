// Model handle: flat-parameter layout of net.INNwProb and the device-side flow program.
// Reference: var_state.py:25-27, 94-124 (layout = JAX pytree order, i.e. sorted dict keys at every level),
// net.py:74-82 (SingleBlock.setup), net.py:196-207 (INNwProb.setup).
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace vmcpde {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
constexpr int MAX_DEVICES = 64;
DeviceCtx g_ctx[MAX_DEVICES];
std::mutex g_ctx_mutex;
}  // namespace

DeviceCtx* device_ctx() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    set_error("device_ctx: cannot query the current HIP device");
    return nullptr;
  }
  return &g_ctx[dev];
}

DeviceCtxLock::DeviceCtxLock() { g_ctx_mutex.lock(); }
DeviceCtxLock::~DeviceCtxLock() { g_ctx_mutex.unlock(); }

int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : fallback;
}

}  // namespace vmcpde

using namespace vmcpde;

extern "C" const char* vmcpde_last_error(void) { return vmcpde::g_err; }
extern "C" int vmcpde_abi_version(void) { return 7; }

namespace {

// offsets[block][trafo][layer][leaf]; trafo index: 0 s1, 1 s2, 2 t1, 3 t2
struct LayoutTable {
  int off[VMCPDE_MAX_DEPTH][4][MAX_LAYERS][2];
};

void fill_trafo(Trafo& t, int n_in, int n_out, const vmcpde_model_desc& d, const int (*off)[2]) {
  t.n_layers = d.n_hidden + 1;
  t.n_in = n_in;
  t.n_out = n_out;
  t.sum_hidden = 0;
  t.max_width = std::max(n_in, n_out);
  int prev = n_in;
  for (int k = 0; k < t.n_layers; ++k) {
    int w = (k < d.n_hidden) ? d.hidden[k] : n_out;
    t.layer[k].n_in = prev;
    t.layer[k].n_out = w;
    t.layer[k].off_bias = off[k][0];
    t.layer[k].off_kernel = off[k][1];
    if (k < d.n_hidden) t.sum_hidden += w;
    t.max_width = std::max(t.max_width, w);
    prev = w;
  }
}

}  // namespace

extern "C" int vmcpde_create(const vmcpde_model_desc* desc, vmcpde_model** out) {
  VMC_REQUIRE(desc && out, "vmcpde_create: null argument");
  const vmcpde_model_desc& d = *desc;
  VMC_REQUIRE(d.dim >= 2 && d.dim <= VMCPDE_MAX_DIM, "dim %d outside [2,%d]", d.dim, VMCPDE_MAX_DIM);
  VMC_REQUIRE(d.depth >= 1 && d.depth <= VMCPDE_MAX_DEPTH, "depth %d outside [1,%d]", d.depth, VMCPDE_MAX_DEPTH);
  VMC_REQUIRE(d.n_hidden >= 0 && d.n_hidden <= VMCPDE_MAX_HIDDEN_LAYERS, "n_hidden %d outside [0,%d]", d.n_hidden,
              VMCPDE_MAX_HIDDEN_LAYERS);
  VMC_REQUIRE(d.n_up >= 1 && d.n_up < d.dim, "n_up %d must be in [1,dim)", d.n_up);
  VMC_REQUIRE(d.coupling >= 0 && d.coupling <= 3, "unknown coupling %d", d.coupling);
  VMC_REQUIRE(d.latent == VMCPDE_LATENT_GAUSS || d.latent == VMCPDE_LATENT_STUDENT_T, "unknown latent %d", d.latent);
  VMC_REQUIRE(d.inds_up && d.inds_down && d.offset, "null partition/offset arrays");
  for (int k = 0; k < d.n_hidden; ++k) VMC_REQUIRE(d.hidden[k] >= 1, "hidden width must be >= 1");
  const int n_dn = d.dim - d.n_up;
  for (int b = 0; b < d.depth; ++b) {
    bool seen[VMCPDE_MAX_DIM] = {false};
    for (int i = 0; i < d.n_up; ++i) {
      int v = d.inds_up[b * d.n_up + i];
      VMC_REQUIRE(v >= 0 && v < d.dim && !seen[v], "block %d: bad ind_up", b);
      seen[v] = true;
    }
    for (int i = 0; i < n_dn; ++i) {
      int v = d.inds_down[b * n_dn + i];
      VMC_REQUIRE(v >= 0 && v < d.dim && !seen[v], "block %d: bad ind_down", b);
      seen[v] = true;
    }
  }

  vmcpde_model* m = new vmcpde_model();
  FlowProgram& fp = m->host;
  memset(&fp, 0, sizeof(fp));
  fp.dim = d.dim;
  fp.n_half = 2 * d.depth;
  fp.coupling = d.coupling;
  fp.latent = d.latent;
  fp.alpha = 10.0;  // net.py:50
  for (int i = 0; i < d.dim; ++i) fp.offset[i] = d.offset[i];

  // ---- flat layout: 'L' < 'L_diag' < 'dist_params' < 'mu' < 'myINN' (ASCII order of the flax param dict keys)
  int pos = 0;
  fp.off_L = pos;
  pos += (d.dim * d.dim - d.dim) / 2;  // net.py:201
  fp.off_Ldiag = pos;
  pos += d.dim;                        // net.py:202
  fp.off_dist = pos;
  pos += (d.latent == VMCPDE_LATENT_STUDENT_T) ? 1 : 0;  // net.py:197-198,204
  fp.off_mu = pos;
  pos += d.dim;                        // net.py:203

  const bool has_t = (d.coupling == VMCPDE_COUPLING_DIFFERENT_ADD);  // net.py:77-79
  LayoutTable* tab = new LayoutTable();
  memset(tab, -1, sizeof(LayoutTable));
  std::vector<std::pair<std::string, int>> blocks;
  for (int b = 0; b < d.depth; ++b) blocks.push_back({"blocks_" + std::to_string(b), b});
  std::sort(blocks.begin(), blocks.end());  // string order: blocks_10 < blocks_2
  fp.global_change = d.global_change ? 1 : 0;
  for (auto& bn : blocks) {
    const int b = bn.second;
    if (fp.global_change) {          // 'global_offset' < 'global_scale' < 's1' (net.py:80-82)
      fp.off_goffset[b] = pos;
      pos += d.dim;
      fp.off_gscale[b] = pos;
      pos += 1;
    }
    const int n_tr = has_t ? 4 : 2;  // sorted names: s1 < s2 < t1 < t2
    for (int tr = 0; tr < n_tr; ++tr) {
      // s1/t1: |up| -> |down| ; s2/t2: |down| -> |up|   (net.py:75-79)
      const bool is1 = (tr == 0 || tr == 2);
      const int n_in = is1 ? d.n_up : n_dn, n_out = is1 ? n_dn : d.n_up;
      int prev = n_in;
      for (int k = 0; k <= d.n_hidden; ++k) {  // Dense_0 < Dense_1 < ... (k <= 3, so string order = numeric)
        const int w = (k < d.n_hidden) ? d.hidden[k] : n_out;
        tab->off[b][tr][k][0] = pos;  // 'bias' < 'kernel'
        pos += w;
        tab->off[b][tr][k][1] = pos;
        pos += prev * w;
        prev = w;
      }
    }
  }
  fp.P = pos;

  // ---- half-steps in forward order (net.py:86-113): first s2 on u2 -> transforms up, then s1 on v1 -> transforms down
  fp.max_width = 0;
  fp.max_sum_act = 0;
  for (int b = 0; b < d.depth; ++b) {
    HalfStep& h0 = fp.hs[2 * b];
    h0.n_a = n_dn;
    h0.n_c = d.n_up;
    for (int i = 0; i < n_dn; ++i) h0.idx_a[i] = d.inds_down[b * n_dn + i];
    for (int i = 0; i < d.n_up; ++i) h0.idx_c[i] = d.inds_up[b * d.n_up + i];
    h0.has_t = has_t;
    fill_trafo(h0.s, n_dn, d.n_up, d, tab->off[b][1]);
    if (has_t) fill_trafo(h0.t, n_dn, d.n_up, d, tab->off[b][3]);
    HalfStep& h1 = fp.hs[2 * b + 1];
    h1.n_a = d.n_up;
    h1.n_c = n_dn;
    for (int i = 0; i < d.n_up; ++i) h1.idx_a[i] = d.inds_up[b * d.n_up + i];
    for (int i = 0; i < n_dn; ++i) h1.idx_c[i] = d.inds_down[b * n_dn + i];
    h1.has_t = has_t;
    fill_trafo(h1.s, d.n_up, n_dn, d, tab->off[b][0]);
    if (has_t) fill_trafo(h1.t, d.n_up, n_dn, d, tab->off[b][2]);
    for (HalfStep* h : {&h0, &h1}) {
      fp.max_width = std::max(fp.max_width, h->s.max_width);
      fp.max_sum_act = std::max(fp.max_sum_act, h->s.n_in + h->s.sum_hidden);
      int wsum = 0;
      for (int k = 0; k < h->s.n_layers; ++k) wsum += h->s.layer[k].n_in * h->s.layer[k].n_out + h->s.layer[k].n_out;
      fp.max_mlp_weights = std::max(fp.max_mlp_weights, wsum + 8);   // + the zero slack behind the stage (flow.hip: stage_weights)
      h->s.wbase = fp.all_mlp_weights;
      fp.all_mlp_weights += wsum + 8;
      if (has_t) {
        h->t.wbase = fp.all_mlp_weights;
        fp.all_mlp_weights += wsum + 8;
      }
      h->act_off_s = fp.act_units;
      fp.act_units += h->s.sum_hidden + h->s.n_out;
      h->act_off_t = fp.act_units;
      if (has_t) fp.act_units += h->t.sum_hidden + h->t.n_out;
    }
  }

  fp.stage_all = (fp.all_mlp_weights <= STAGE_ALL_MAX_DOUBLES) ? 1 : 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipMalloc(&m->dev, sizeof(FlowProgram));
  if (e == hipSuccess) e = hipMalloc(&m->cache, sizeof(LatentCache));
  if (e == hipSuccess) e = hipMemcpy(m->dev, &fp, sizeof(FlowProgram), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("vmcpde_create: %s", hipGetErrorString(e));
    if (m->dev) (void)hipFree(m->dev);
    if (m->cache) (void)hipFree(m->cache);
    delete tab;
    delete m;
    return -2;
  }
  m->device = dev;
  m->layout = tab;
  *out = m;
  return 0;
}

extern "C" int vmcpde_destroy(vmcpde_model* m) {
  if (!m) return 0;
  if (m->dev) (void)hipFree(m->dev);
  if (m->cache) (void)hipFree(m->cache);
  delete static_cast<LayoutTable*>(m->layout);
  delete m;
  return 0;
}

extern "C" int vmcpde_num_parameters(const vmcpde_model* m) { return m ? m->host.P : -1; }

extern "C" int vmcpde_param_offset(const vmcpde_model* m, int block, int trafo, int layer, int leaf) {
  if (!m) return -1;
  const FlowProgram& fp = m->host;
  if (block < 0) {
    switch (trafo) {
      case 0: return fp.off_L;
      case 1: return fp.off_Ldiag;
      case 2: return fp.latent == VMCPDE_LATENT_STUDENT_T ? fp.off_dist : -1;
      case 3: return fp.off_mu;
      default: return -1;
    }
  }
  if (block < fp.n_half / 2 && (trafo == 4 || trafo == 5))
    return fp.global_change ? (trafo == 4 ? fp.off_goffset[block] : fp.off_gscale[block]) : -1;
  if (block >= fp.n_half / 2 || trafo < 0 || trafo > 3 || layer < 0 || layer >= MAX_LAYERS || leaf < 0 || leaf > 1) return -1;
  return static_cast<const LayoutTable*>(m->layout)->off[block][trafo][layer][leaf];
}
