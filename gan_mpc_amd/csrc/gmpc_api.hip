// C-ABI entry points of libgan_mpc_amd.so (declared in include/gan_mpc_amd.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "gmpc_device.h"

// launchers defined in the kernel translation units ---------------------------------------------
void gmpc_launch_rollout(const TrajArgs&, hipStream_t);
typedef void (*gmpc_ls_eval_fn)(void* user, const TrajArgs&, int max_items, hipStream_t);
int gmpc_launch_linesearch(const TrajArgs&, const LsWork&, hipStream_t, gmpc_ls_eval_fn eval = nullptr,
                           void* user = nullptr, const LsSplit* split = nullptr);
void gmpc_launch_masks(int, int, int, int, const MlpDesc&, const float*, const float*, uint32_t*,
                       hipStream_t);
int gmpc_launch_linearize(int, int, int, int, const MlpDesc&, const uint32_t*, const int*, float*,
                          hipStream_t);
bool gmpc_linearize_regs_covers(int, int, const MlpDesc&, const LinPad&);
int gmpc_launch_linearize_regs_list(int cap, int T, int n, int m, const MlpDesc&, const LinPad&, const uint32_t*,
                                    const int* tlist, const int* tcount, float* AB, int grid, hipStream_t);
bool gmpc_ls_rounds_on_ls16(const TrajArgs&);
int gmpc_launch_linearize_regs(int, int, int, int, const MlpDesc&, const LinPad&, const uint32_t*,
                               const int*, float*, int, int, hipStream_t, hipEvent_t mid_event = nullptr);
int gmpc_launch_linearize_mfma(int, int, int, int, const MlpDesc&, const LinPad&, const uint32_t*,
                               const int*, float*, int, int, hipStream_t);
const char* gmpc_linearize_regs_last_name();
void gmpc_launch_bgemm_tn(const BgemmArgs&, hipStream_t);
size_t gmpc_dynfit_stride(const gmpc_shape*);
int gmpc_launch_dynfit(int, int, int, int, const MlpDesc&, const float*, const float*, const float*, float,
                       int, float*, float*, float*, int, float*, hipStream_t);
int gmpc_big_backward(const BigWork&, int, const MlpDesc&, const LinPad&, const uint32_t*, const float*,
                      const float*, const float*, const float*, const float*, const float*, const int*,
                      float*, float*, float*, float*, const float*, float*, hipStream_t,
                      const DynlDesc* dl = nullptr, const float* lam_sol = nullptr);
int gmpc_big_forward_tangent(const BigWork&, int, const MlpDesc&, const LinPad&, const uint32_t*,
                             const float*, const float*, float*, float*, hipStream_t,
                             const DynlDesc* dl = nullptr, const float* X = nullptr, const float* U = nullptr);
void gmpc_launch_big_cont(int, int, int, const float*, const float*, const int*, const float*,
                          const float*, const float*, const float*, const gmpc_ilqr_opts&, const int*,
                          int*, hipStream_t);
size_t gmpc_linpad_floats(const gmpc_shape*);
void gmpc_linpad_prepare(const MlpDesc&, int, int, float*, size_t, LinPad*, hipStream_t);
int gmpc_launch_terminal(int, int, int, const MlpDesc&, const float*, const float*, const int*,
                         float*, float*, hipStream_t);
void gmpc_launch_riccati(const RiccatiArgs&, hipStream_t);
bool gmpc_riccati_w2h_shape(const RiccatiArgs&);
void gmpc_launch_riccati_w2h(const RiccatiArgs&, const float* lx, float* bvec_out, hipStream_t);
size_t gmpc_riccati_lds_bytes(int n, int m);
void gmpc_launch_transpose(int, int, const float*, float*, hipStream_t);
void gmpc_launch_lstm_fwd(int, const CriticDesc&, const float*, float*, float*, float*, float*,
                          const float*, hipStream_t);
void gmpc_launch_lstm_bwd(int, const CriticDesc&, const float*, const float*, const float*, float*,
                          float*, hipStream_t);
void gmpc_launch_wgrad(int, int, int, const float*, int, const float*, int, float*, float*, int,
                       float*, int, hipStream_t, long, bool);
bool gmpc_launch_wgrad_batch(WgProb*, int, float*, long, hipStream_t);
void gmpc_launch_colsum(int, int, const float*, int, float*, float*, hipStream_t);
// second-generation LSTM kernels (gmpc_critic_lstm.hip): n <= 32
bool gmpc_lstm2_supported(const CriticDesc&);
long gmpc_lstm2_wpart_floats(const CriticDesc&, int);
bool gmpc_launch_lstm_fwd2(int, const CriticDesc&, const float*, float*, float*, float*, float*, hipStream_t);
bool gmpc_launch_lstm_bwd2(int, const CriticDesc&, const float*, const float*, const float*, const float*,
                           const float*, float*, float*, float*, float*, float*, hipStream_t);
void gmpc_launch_head2(int, const CriticDesc&, int, const float*, const float*, float*, float*, float*, float*,
                       float*, float*, int, hipStream_t);
void gmpc_launch_mlp_transpose_all(const MlpDesc&, hipStream_t);
#define GMPC_HEAD2_LD 264
void gmpc_launch_sum(int, const float*, float*, int, hipStream_t);
void gmpc_launch_adam(long, float*, const float*, float*, float*, float, int, double, double, double,
                      double, double, float*, hipStream_t);
void gmpc_launch_polyak(long, const float*, const float*, double, float*, hipStream_t);
void gmpc_launch_l2loss(int, int, int, int, const float*, const float*, float*, float*, hipStream_t);
void gmpc_launch_bvec(int, int, int, int, const float*, const float*, float*, hipStream_t);
void gmpc_launch_costvjp(int, int, int, int, const MlpDesc&, const float*, float, const float*,
                         const float*, const float*, int, const float*, const float*, float*, float*,
                         float*, int, hipStream_t);

// LSTM dynamics variant (gmpc_dynl.hip)
void gmpc_launch_dynl_rollout(DynlTrajArgs, hipStream_t);
void gmpc_launch_dynl_candidates(DynlTrajArgs, int, hipStream_t);
void gmpc_launch_dynl_jac(int, int, int, int, const DynlDesc&, const float*, const float*, const int*, float*,
                          hipStream_t);
void gmpc_launch_dynl_curv(int, int, int, int, const DynlDesc&, const float*, const float*, const float*, const int*,
                           float*, hipStream_t);
size_t gmpc_dynl_fit_stride(const DynlDesc&);
void gmpc_launch_dynl_fit(int, int, const DynlDesc&, const float*, const float*, const float*, float, int, float*,
                          float*, float*, int, float*, float*, hipStream_t);
void gmpc_launch_cols_gather(long, int, int, const float*, float*, hipStream_t);
void gmpc_launch_cols_scatter(long, int, int, const float*, float*, hipStream_t);

enum { PROF_ROLLOUT = 0, PROF_LINEARIZE, PROF_TERMINAL, PROF_RICCATI, PROF_LINESEARCH, PROF_LSTM_FWD,
       PROF_HEAD, PROF_LSTM_BWD, PROF_WGRAD, PROF_ADAM };

// error handling -------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// the same record for the other translation units of the library (gmpc_comm.hip)
int gmpc_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(GMPC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                \
  } while (0)

extern "C" const char* gmpc_last_error(void) { return g_err; }
extern "C" const char* gmpc_version(void) { return "gan_mpc_amd 0.1 (gfx950)"; }

static long mlp_count(int L, const int* dims) {
  long c = 0;
  for (int l = 0; l < L; ++l) c += (long)dims[l] * dims[l + 1] + dims[l + 1];
  return c;
}

static int check_shape(const gmpc_shape* s) {
  if (!s) return fail(GMPC_EINVAL, "shape is null");
  if (s->n < 1 || s->m < 1 || s->T < 1) return fail(GMPC_EINVAL, "n, m, T must be positive");
  if (s->n > 1024 || s->m > 64)
    return fail(GMPC_EINVAL, "unsupported shape n=%d m=%d: n <= 1024 and m <= 64 are built", s->n, s->m);
  // (n <= 64 with more than 32 controls: the fused small-state kernels hold an m x m factor per lane and the gain
  // block of a step in LDS; such shapes take the step-major pipeline of the large-state path, which is general)
  if (s->dyn_layers < 1 || s->dyn_layers > GMPC_MAX_LAYERS)
    return fail(GMPC_EINVAL, "dyn_layers must be in [2, %d]", GMPC_MAX_LAYERS);
  if (s->cost_layers < 1 || s->cost_layers > GMPC_MAX_LAYERS)
    return fail(GMPC_EINVAL, "cost_layers must be in [1, %d]", GMPC_MAX_LAYERS);
  if (s->dyn_lstm_features > 0) {
    // LSTM dynamics variant: xc = [x, c, h], dyn_dims describes the relu tail h' -> x
    const int Fd = s->dyn_lstm_features, nx = s->x_size;
    if (Fd > 128) return fail(GMPC_EINVAL, "unsupported shape: dynamics lstm_features = %d > 128", Fd);
    if (nx < 1 || s->n != nx + 2 * Fd)
      return fail(GMPC_EINVAL, "LSTM dynamics: n must be x_size + 2 * dyn_lstm_features (n=%d, x_size=%d, F=%d)",
                  s->n, nx, Fd);
    if (s->dyn_dims[0] != Fd || s->dyn_dims[s->dyn_layers] != nx)
      return fail(GMPC_EINVAL, "LSTM dynamics: dyn_dims (the tail) must start with dyn_lstm_features and end with x_size");
    if (s->dyn_layers < 1) return fail(GMPC_EINVAL, "dyn_layers must be >= 1");
  } else {
    if (s->x_size != 0 && s->x_size != s->n)
      return fail(GMPC_EINVAL, "x_size must equal n (or 0) for the MLP dynamics: its carry is empty");
    if (s->dyn_layers < 2)
      return fail(GMPC_EINVAL, "dyn_layers must be in [2, %d]", GMPC_MAX_LAYERS);
    if (s->dyn_dims[0] != s->n + s->m || s->dyn_dims[s->dyn_layers] != s->n)
      return fail(GMPC_EINVAL, "dyn_dims must start with n+m and end with n");
  }
  if (s->cost_dims[0] != s->n) return fail(GMPC_EINVAL, "cost_dims must start with n");
  if (s->cost_dims[s->cost_layers] > 32)
    return fail(GMPC_EINVAL, "cost fout must be <= 32");
  for (int l = 1; l < s->dyn_layers; ++l)
    if (s->dyn_dims[l] < 1 || s->dyn_dims[l] > GMPC_THREADS)
      return fail(GMPC_EINVAL, "dynamics hidden width must be in [1, %d]", GMPC_THREADS);
  for (int l = 1; l < s->cost_layers; ++l)
    if (s->cost_dims[l] < 1 || s->cost_dims[l] > GMPC_THREADS)
      return fail(GMPC_EINVAL, "cost hidden width must be in [1, %d]", GMPC_THREADS);
  if (s->lstm_features != 0) {
    // (64: the register-weight kernels of gmpc_critic_lstm.hip / gmpc_critic.hip; other counts: k_lstm_fwd_g / _bwd_g)
    if (s->lstm_features < 1 || s->lstm_features > 128)
      return fail(GMPC_EINVAL, "unsupported shape: critic lstm_features = %d outside [1, 128]", s->lstm_features);
    if (s->head_layers < 1 || s->head_layers > GMPC_MAX_LAYERS)
      return fail(GMPC_EINVAL, "head_layers must be in [1, %d]", GMPC_MAX_LAYERS);
    if (s->head_dims[0] != s->lstm_features || s->head_dims[s->head_layers] != 1)
      return fail(GMPC_EINVAL, "head_dims must start with F and end with 1");
    for (int l = 1; l < s->head_layers; ++l)
      if (s->head_dims[l] < 1 || s->head_dims[l] > GMPC_THREADS)
        return fail(GMPC_EINVAL, "head hidden width must be in [1, %d]", GMPC_THREADS);
  }
  return 0;
}

extern "C" long gmpc_param_count(const gmpc_shape* s, int which) {
  if (!s) return -1;
  if (which == 0) {
    const long Fd = s->dyn_lstm_features;
    const long cell = Fd > 0 ? ((long)s->x_size + s->m + Fd) * 4 * Fd + 4 * Fd : 0;
    return cell + mlp_count(s->dyn_layers, s->dyn_dims);
  }
  if (which == 1) return mlp_count(s->cost_layers, s->cost_dims);
  if (which == 2) {
    const long F = s->lstm_features;
    const long nx = s->x_size > 0 ? s->x_size : s->n;     // the critic scores x sequences
    return nx * 4 * F + F * 4 * F + 4 * F + mlp_count(s->head_layers, s->head_dims);
  }
  return -1;
}

// multi-GPU exchange (gmpc_comm.hip)
struct GmpcComm { void* comm = nullptr; int world = 1, rank = 0; };
int gmpc_comm_unique_id_impl(char*);
int gmpc_comm_init_impl(GmpcComm*, int, int, const char*);
int gmpc_comm_allreduce_impl(GmpcComm*, float*, long, hipStream_t);
void gmpc_comm_destroy_impl(GmpcComm*);

#define GMPC_POLL_DEPTH 4   // iterations the host may enqueue ahead of the convergence flags it has seen

struct gmpc_ctx {
  gmpc_shape sh;
  GmpcComm comm;
  int nx = 0;            // x part of xc (= n unless the dynamics carry rides in xc)
  bool dynl = false;     // LSTM dynamics variant
  DynlDesc dl{};
  float *xg = nullptr, *lxg = nullptr;   // x columns of Xs / d loss / dx (critic-facing, dynl only)
  float* phi = nullptr;                  // dynl, small-state path: [B][T][n+m][n+m] curvature for the bilevel solve
  int maxB, device;
  std::vector<void*> allocs;
  // bound parameters
  const float* mpc_w = nullptr;
  MlpDesc dyn{}, cost{};
  float *dynT = nullptr, *costT = nullptr, *linpad = nullptr;
  size_t linpad_floats = 0;
  LinPad lp{};
  bool params_set = false;
  // trajectory workspace
  uint32_t *masks, *maskc;
  float *Xc, *Uc, *AB, *QT, *qT;
  float *Xs, *Us, *goals, *Ks, *ks, *grads, *adjs;
  float *obj, *alpha, *obj_step, *U_step;
  int *iters, *cont;
  int* hcont = nullptr;                       // pinned ring of continuation flags (gmpc_ilqr_solve)
  hipEvent_t poll_ev[GMPC_POLL_DEPTH] = {};
  int solB = 0;
  // bilevel workspace
  float *lx, *Bvec, *Hout, *dX, *gmpc, *cact, *cdel, *bl_loss;
  int cstride;
  // critic workspace
  float *critT, *gates, *cs, *hp, *hT, *dz, *hacts, *hdels, *dhT, *cscore, *closs;
  float* lwp = nullptr;        // weight-gradient partials of k_lstm_bwd2, one [85][256] block per 4 sequences
  float* plast = nullptr;      // k_head2: last layer's act * dscore products and dscore, [Bc + 8][GMPC_HEAD2_LD]
  int hstride;
  LsWork lsw{};
  // large-state (n > 64) backward pass
  bool big = false;
  BigWork bw{};
  float *WhT = nullptr, *xT = nullptr, *xproj = nullptr;   // wide-input critic (n + F > 256)
  // dynamics regression (allocated on first use)
  float *dfpred = nullptr, *dfacts = nullptr, *dfdels = nullptr, *dfloss = nullptr, *dfsave = nullptr;
  int dfstride = 0;
  // shared scratch
  float *wpart, *scratch;
  long wpart_floats;
  // optional per-kernel timing with HIP events on the launch stream (gmpc_profile_*)
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[GMPC_PROF_SLOTS];
  const char* lin_kernel = "";      // kernel the last Jacobian chain ran on (gmpc_profile_kernel_name)
  char lin_kernel_buf[96] = "";     // name of the chain instantiation this ctx launched last (copied at launch time)
  hipEvent_t lin_event = nullptr;   // caller's event, recorded after the Jacobian chain (gmpc_set_linearize_event)
  // the critic's head weight gradients run beside the BPTT sweep (critic_forward_backward): a context-owned side
  // stream forked after k_head2 and joined behind the sweep
  hipStream_t crit_side = nullptr;
  hipEvent_t crit_fork = nullptr, crit_join = nullptr, crit_tr = nullptr;
  // gmpc_ilqr_solve: the Jacobian chain of the trajectories whose line search ends with its first round runs on a
  // context-owned side stream beside the later rounds
  hipStream_t ls_side = nullptr;
  hipEvent_t ls_joined = nullptr;
  LsSplit ls_split{};
};

// RAII bracket: records a start/stop event pair around one kernel launch when profiling is on
struct ProfScope {
  gmpc_ctx* c; int slot; hipStream_t s; hipEvent_t e1 = nullptr;
  ProfScope(gmpc_ctx* c_, int slot_, hipStream_t s_) : c(c_), slot(slot_), s(s_) {
    if (!c->prof) return;
    hipEvent_t e0;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e1 = nullptr; return; }
    (void)hipEventRecord(e0, s);
    c->prof_ev[slot].push_back({e0, e1});
  }
  ~ProfScope() { if (e1) (void)hipEventRecord(e1, s); }
};

template <typename Tp>
static int dalloc(gmpc_ctx* c, Tp** p, size_t count) {
  void* q = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(&q, count * sizeof(Tp));
  if (e != hipSuccess)
    return fail(GMPC_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(Tp),
                hipGetErrorString(e));
  c->allocs.push_back(q);
  *p = static_cast<Tp*>(q);
  return 0;
}
#define TRY(expr) do { int r_ = (expr); if (r_ != 0) return r_; } while (0)

static void bind_mlp(MlpDesc& d, int L, const int* dims, const float* flat, float* flatT) {
  d.L = L;
  for (int l = 0; l <= L; ++l) d.dims[l] = dims[l];
  long off = 0;
  for (int l = 0; l < L; ++l) {
    d.W[l] = flat + off;
    d.WT[l] = flatT ? flatT + off : nullptr;
    off += (long)dims[l] * dims[l + 1];
    d.b[l] = flat + off;
    off += dims[l + 1];
  }
}

static void transpose_mlp(const MlpDesc& d, hipStream_t s) {
  for (int l = 0; l < d.L; ++l)
    gmpc_launch_transpose(d.dims[l], d.dims[l + 1], d.W[l], const_cast<float*>(d.WT[l]), s);
}

extern "C" int gmpc_create(const gmpc_shape* shape, int max_batch, int device, gmpc_ctx** out) {
  if (!out) return fail(GMPC_EINVAL, "out is null");
  *out = nullptr;
  TRY(check_shape(shape));
  if (max_batch < 1) return fail(GMPC_EINVAL, "max_batch must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(GMPC_ENODEV, "no HIP device is visible; libgan_mpc_amd has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(GMPC_EINVAL, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  gmpc_ctx* c = new (std::nothrow) gmpc_ctx();
  if (!c) return fail(GMPC_ENOMEM, "out of host memory");
  c->sh = *shape;
  c->maxB = max_batch;
  c->device = device;
  const gmpc_shape& s = c->sh;
  const size_t B = max_batch, n = s.n, m = s.m, T = s.T, nm = n + m, Lh = s.dyn_layers - 1;
  c->dynl = s.dyn_lstm_features > 0;
  c->nx = c->dynl ? s.x_size : s.n;
  int rc = 0;
#define A_(p, cnt) if (!rc) rc = dalloc(c, &c->p, (cnt))
  A_(dynT, mlp_count(s.dyn_layers, s.dyn_dims));
  A_(costT, mlp_count(s.cost_layers, s.cost_dims));
  c->linpad_floats = c->dynl ? 0 : gmpc_linpad_floats(&c->sh);
  A_(linpad, c->linpad_floats);
  A_(masks, B * T * Lh * GMPC_MW);
  // line-search candidates: GMPC_LS_ITEMS per trajectory
  A_(maskc, GMPC_LS_ITEMS * B * T * Lh * GMPC_MW);
  A_(Xc, GMPC_LS_ITEMS * B * (T + 1) * n);
  A_(Uc, GMPC_LS_ITEMS * B * T * m);
  for (int i = 0; i < 2; ++i) {
    if (!rc) rc = dalloc(c, &c->lsw.item_b[i], GMPC_LS_ITEMS * B);
    if (!rc) rc = dalloc(c, &c->lsw.item_k[i], GMPC_LS_ITEMS * B);
  }
  if (!rc) rc = dalloc(c, &c->lsw.first, B);
  if (!rc) rc = dalloc(c, &c->lsw.slot, GMPC_LS_ITEMS * B);
  if (!rc) rc = dalloc(c, &c->lsw.cnt, B);
  if (!rc) rc = dalloc(c, &c->lsw.kfirst, B);
  if (!rc) rc = dalloc(c, &c->lsw.prevk, B);
  if (!rc) rc = dalloc(c, &c->lsw.counts, GMPC_LS_ROUNDS_MAX + 1 + GMPC_LS_STATS);
  if (!rc) rc = dalloc(c, &c->lsw.run, B);
  if (!rc) rc = dalloc(c, &c->lsw.objc, GMPC_LS_ITEMS * B);
  c->big = s.n > 64 || s.m > 32;
  if (!c->big) {
    A_(AB, B * T * n * nm);
  } else {
    // step-major backward pass: one step's Jacobians and the n x n work matrices, each followed by
    // zeroed rows the GEMMs may read past the end
    c->bw.n = s.n; c->bw.m = s.m; c->bw.T = s.T; c->bw.ng = c->nx;
    const size_t pad = 16 * nm;
#define B_(p, cnt) if (!rc) { rc = dalloc(c, &c->bw.p, (cnt)); if (!rc) (void)hipMemset(c->bw.p, 0, (cnt) * sizeof(float)); }
    B_(ABt, B * n * nm + pad);
    B_(P, B * n * n + pad);
    B_(PAB, B * n * nm + pad);
    B_(T1, B * n * n + pad);
    B_(HG, B * m * nm + pad);
    B_(KV, 2 * B * m * n + pad);
    B_(VK, 2 * B * m * n + pad);
    B_(pvec, B * n); B_(lam, B * n); B_(sbuf, B); B_(gn2, B);
    // low-rank form of the Jacobians (gmpc_large.hip): MLP dynamics whose last hidden width is below n / 2
    // (8 n^2 h instead of 4 n^3 flops per Riccati step); GMPC_BIG_DENSE=1 keeps the dense form (A/B timing)
    {
      const size_t hl = s.dyn_dims[s.dyn_layers - 1];
      size_t hmax = 0;
      for (int l = 1; l < s.dyn_layers; ++l) hmax = (size_t)s.dyn_dims[l] > hmax ? s.dyn_dims[l] : hmax;
      const bool force = getenv("GMPC_BIG_LOWRANK") != nullptr && hl < n;     // (A/B timing)
      if (!c->dynl && (2 * hl < n || force) && getenv("GMPC_BIG_DENSE") == nullptr) {
        c->bw.h = (int)hl;
        B_(Vt, B * hl * nm + pad);
        B_(W1b, B * hl * n + pad);
        B_(W2b, B * hl * nm + pad);
        B_(Sa, B * hmax * hl + pad);
        B_(Sb, B * hmax * hl + pad);
      }
    }
    // one-step-ahead Jacobians on a side stream (MLP dynamics; GMPC_BIG_PIPELINE=0: one stream, one buffer)
    {
      const char* e = getenv("GMPC_BIG_PIPELINE");
      // (the low-rank form supports it too -- Vt2 / Sm -- and gains nothing: its factor GEMMs and k_big_step slow
      // each other down by what the overlap saves, C5 1.980 vs 1.984 s; GMPC_BIG_PIPELINE=2 turns it on there)
      const bool lr_too = e != nullptr && e[0] == '2';
      if (!c->dynl && !(e != nullptr && e[0] == '0') && (c->bw.h == 0 || lr_too)) {
        // (the second copy and the side stream are an optimisation: when any of them cannot be had -- the copy is
        // B n (n + m) floats, 4.5 GB at n = 1024 -- the pass runs on one stream with one buffer)
        if (!rc) {
          const size_t cnt2 = c->bw.h > 0 ? B * (size_t)c->bw.h * nm + pad : B * n * nm + pad;
          const size_t cnt3 = c->bw.h > 0 ? B * (size_t)c->bw.h * c->bw.h + pad : 0;
          float *p2 = nullptr, *p3 = nullptr;
          bool ok = hipMalloc(reinterpret_cast<void**>(&p2), cnt2 * sizeof(float)) == hipSuccess;
          if (ok && cnt3) ok = hipMalloc(reinterpret_cast<void**>(&p3), cnt3 * sizeof(float)) == hipSuccess;
          ok = ok && hipMemset(p2, 0, cnt2 * sizeof(float)) == hipSuccess;
          if (ok && p3) ok = hipMemset(p3, 0, cnt3 * sizeof(float)) == hipSuccess;
          ok = ok && hipStreamCreateWithFlags(&c->bw.side, hipStreamNonBlocking) == hipSuccess;
          ok = ok && hipEventCreateWithFlags(&c->bw.ev_start, hipEventDisableTiming) == hipSuccess;
          for (int i = 0; i < 2 && ok; ++i)
            ok = hipEventCreateWithFlags(&c->bw.ev_ready[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->bw.ev_free[i], hipEventDisableTiming) == hipSuccess;
          if (ok) {
            c->allocs.push_back(p2);
            if (p3) c->allocs.push_back(p3);
            if (c->bw.h > 0) { c->bw.Vt2 = p2; c->bw.Sm = p3; } else { c->bw.ABt2 = p2; }
          } else {
            (void)hipGetLastError();
            if (p2) (void)hipFree(p2);
            if (p3) (void)hipFree(p3);
            if (c->bw.side) { (void)hipStreamDestroy(c->bw.side); c->bw.side = nullptr; }
            if (c->bw.ev_start) { (void)hipEventDestroy(c->bw.ev_start); c->bw.ev_start = nullptr; }
            for (int i = 0; i < 2; ++i) {
              if (c->bw.ev_ready[i]) { (void)hipEventDestroy(c->bw.ev_ready[i]); c->bw.ev_ready[i] = nullptr; }
              if (c->bw.ev_free[i]) { (void)hipEventDestroy(c->bw.ev_free[i]); c->bw.ev_free[i] = nullptr; }
            }
          }
        }
      }
    }
#undef B_
    c->AB = c->bw.ABt;
  }
  A_(QT, B * n * n);
  A_(qT, B * n);
  A_(Xs, B * (T + 1) * n);
  A_(Us, B * T * m);
  A_(goals, B * (T + 1) * n);
  A_(Ks, B * T * m * n + 16 * nm);
  A_(ks, B * T * m);
  A_(grads, B * T * m);
  A_(adjs, B * (T + 1) * n);
  A_(obj, B); A_(alpha, B); A_(obj_step, B); A_(U_step, B);
  A_(iters, B); A_(cont, B);
  // bilevel
  int cin = 0, cout_ = 0;
  for (int l = 0; l < s.cost_layers; ++l) { cin += s.cost_dims[l]; cout_ += s.cost_dims[l + 1]; }
  c->cstride = cin > cout_ ? cin : cout_;
  A_(lx, B * (T + 1) * n);
  A_(Bvec, B * T * m);
  A_(Hout, B * T * m);
  A_(dX, B * (T + 1) * n);
  A_(gmpc, B * 3);
  A_(cact, (2 * B + 8) * c->cstride);
  A_(cdel, (2 * B + 8) * c->cstride);
  A_(bl_loss, B);
  // critic
  long wmax = 0;
  for (int l = 0; l < s.cost_layers; ++l) {
    long w = (long)s.cost_dims[l] * s.cost_dims[l + 1] + s.cost_dims[l + 1];
    if (w > wmax) wmax = w;
  }
  c->hstride = 1;
  if (c->dynl) {
    A_(xg, B * (T + 1) * c->nx);
    A_(lxg, B * (T + 1) * c->nx);
    if (c->big) {
      if (!rc) rc = dalloc(c, &c->bw.Phi, B * nm * nm);
    } else {
      A_(phi, B * T * nm * nm);
    }
  }
  if (s.lstm_features > 0) {
    // (the saves of the LSTM kernels are laid out per workgroup of 4 sequences: round the batch up)
    const size_t Bc = (2 * B + 3) / 4 * 4, F = s.lstm_features, T1 = T + 1, n = c->nx;   // the critic scores x sequences
    int hin = 0, hout = 0;
    for (int l = 0; l < s.head_layers; ++l) { hin += s.head_dims[l]; hout += s.head_dims[l + 1]; }
    c->hstride = hin > hout ? hin : hout;
    A_(critT, (n + F) * 4 * F + mlp_count(s.head_layers, s.head_dims));
    A_(gates, Bc * T1 * 4 * F);
    A_(cs, Bc * T1 * F);
    A_(hp, Bc * T1 * F);
    A_(hT, Bc * F);
    A_(dz, (Bc * T1 + 8) * 4 * F);
    if (n + F > GMPC_THREADS) {
      A_(WhT, 4 * F * F);
      A_(xT, Bc * T1 * n + 16 * Bc * T1);
      A_(xproj, (Bc * T1 + 16) * 4 * F);
    }
    {
      CriticDesc probe{};
      probe.n = (int)n; probe.F = (int)F; probe.T1 = (int)T1;
      if (gmpc_lstm2_supported(probe)) A_(lwp, gmpc_lstm2_wpart_floats(probe, (int)Bc));
    }
    A_(plast, (Bc + 8) * GMPC_HEAD2_LD);
    A_(hacts, (Bc + 8) * c->hstride);
    A_(hdels, (Bc + 8) * c->hstride);
    A_(dhT, Bc * F);
    A_(cscore, Bc);
    A_(closs, Bc);
    long w = (long)F * 4 * F + 4 * F;
    if (w > wmax) wmax = w;
    for (int l = 0; l < s.head_layers; ++l) {
      w = (long)s.head_dims[l] * s.head_dims[l + 1] + s.head_dims[l + 1];
      if (w > wmax) wmax = w;
    }
  }
  c->wpart_floats = 256 * wmax;
  if (c->wpart_floats < (12L << 20)) c->wpart_floats = 12L << 20;
  A_(wpart, c->wpart_floats);
  A_(scratch, 1024);
#undef A_
  // B operands of the MFMA weight-gradient GEMM are read a few rows past the end: keep them finite
  if (!rc && c->cdel) (void)hipMemset(c->cdel, 0, (2 * B + 8) * c->cstride * sizeof(float));
  if (!rc && c->cact) (void)hipMemset(c->cact, 0, (2 * B + 8) * c->cstride * sizeof(float));
  if (!rc && s.lstm_features > 0) {
    (void)hipMemset(c->dz, 0, ((size_t)2 * B * (T + 1) + 8) * 4 * s.lstm_features * sizeof(float));
    (void)hipMemset(c->hdels, 0, ((size_t)2 * B + 8) * c->hstride * sizeof(float));
    (void)hipMemset(c->hacts, 0, ((size_t)2 * B + 8) * c->hstride * sizeof(float));
    (void)hipMemset(c->plast, 0, ((size_t)2 * B + 8) * GMPC_HEAD2_LD * sizeof(float));
  }
  if (!rc) (void)hipMemset(c->Ks, 0, (B * T * m * n + 16 * nm) * sizeof(float));
  if (rc) {
    gmpc_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" int gmpc_destroy(gmpc_ctx* c) {
  if (!c) return 0;
  gmpc_comm_destroy_impl(&c->comm);
  if (c->hcont) {
    (void)hipHostFree(c->hcont);
    for (int i = 0; i < GMPC_POLL_DEPTH; ++i) (void)hipEventDestroy(c->poll_ev[i]);
  }
  if (c->crit_side) (void)hipStreamDestroy(c->crit_side);
  if (c->ls_side) (void)hipStreamDestroy(c->ls_side);
  if (c->ls_joined) (void)hipEventDestroy(c->ls_joined);
  if (c->ls_split.ev) (void)hipEventDestroy(c->ls_split.ev);
  if (c->crit_fork) (void)hipEventDestroy(c->crit_fork);
  if (c->crit_join) (void)hipEventDestroy(c->crit_join);
  if (c->crit_tr) (void)hipEventDestroy(c->crit_tr);
  if (c->bw.side) (void)hipStreamDestroy(c->bw.side);
  if (c->bw.ev_start) (void)hipEventDestroy(c->bw.ev_start);
  for (int i = 0; i < 2; ++i) {
    if (c->bw.ev_ready[i]) (void)hipEventDestroy(c->bw.ev_ready[i]);
    if (c->bw.ev_free[i]) (void)hipEventDestroy(c->bw.ev_free[i]);
  }
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
  return 0;
}

extern "C" int gmpc_set_params(gmpc_ctx* c, const float* mpc_w, const float* dyn, const float* cost,
                               void* stream) {
  if (!c || !mpc_w || !dyn || !cost) return fail(GMPC_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  c->mpc_w = mpc_w;
  c->solB = 0;   // a held solution belongs to the previous parameters
  bind_mlp(c->cost, c->sh.cost_layers, c->sh.cost_dims, cost, c->costT);
  transpose_mlp(c->cost, s);
  if (c->dynl) {
    // LSTM variant: Wx [(nx+m)][4F] | Wh [F][4F] | b [4F] | the tail's Dense layers
    const long Fd = c->sh.dyn_lstm_features, nx = c->nx, m = c->sh.m;
    c->dl.nx = (int)nx; c->dl.F = (int)Fd; c->dl.m = (int)m;
    c->dl.Wx = dyn;
    c->dl.Wh = dyn + (nx + m) * 4 * Fd;
    c->dl.b = c->dl.Wh + Fd * 4 * Fd;
    bind_mlp(c->dl.tail, c->sh.dyn_layers, c->sh.dyn_dims, c->dl.b + 4 * Fd, nullptr);
    c->dyn = c->dl.tail;     // generic fields (layer count for the mask bookkeeping of the line search)
  } else {
    bind_mlp(c->dyn, c->sh.dyn_layers, c->sh.dyn_dims, dyn, c->dynT);
    transpose_mlp(c->dyn, s);
    gmpc_linpad_prepare(c->dyn, c->sh.n, c->sh.m, c->linpad, c->linpad_floats, &c->lp, s);
  }
  if (getenv("GMPC_LIN_STAMPS")) {   // diagnostic build of the timing only; never set in production
    (void)hipMemsetAsync(c->scratch + 768, 0, 64, s);
    c->lp.dbg = reinterpret_cast<unsigned long long*>(c->scratch + 768);
  }
  HIP_TRY(hipGetLastError());
  c->params_set = true;
  return 0;
}

static int check_call(gmpc_ctx* c, int B, bool need_params = true) {
  if (!c) return fail(GMPC_EINVAL, "ctx is null");
  if (need_params && !c->params_set) return fail(GMPC_EINVAL, "gmpc_set_params has not been called");
  if (B < 1 || B > c->maxB) return fail(GMPC_EINVAL, "B=%d outside [1, max_batch=%d]", B, c->maxB);
  if (hipSetDevice(c->device) != hipSuccess) return fail(GMPC_EHIP, "hipSetDevice failed");
  // hipGetLastError() is per-thread and shared with every other HIP user of the process (torch
  // leaves benign errors behind): start every entry point from a clean slate so that the checks
  // after our launches report only our own failures
  (void)hipGetLastError();
  return 0;
}

static TrajArgs base_traj(gmpc_ctx* c, int B, const float* goal) {
  TrajArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.n = c->sh.n; a.m = c->sh.m; a.T = c->sh.T;
  a.dyn = c->dyn; a.cost = c->cost; a.mpc_w = c->mpc_w; a.goal = goal;
  return a;
}

static DynlTrajArgs base_dynl(gmpc_ctx* c, int B, const float* goal) {
  DynlTrajArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = c->sh.T; a.d = c->dl; a.cost = c->cost; a.mpc_w = c->mpc_w; a.goal = goal;
  return a;
}

extern "C" int gmpc_rollout_cost(gmpc_ctx* c, int B, const float* x0, const float* U,
                                 const float* goal, float* X, float* costs, void* stream) {
  TRY(check_call(c, B));
  if (!x0 || !U || !goal || !X) return fail(GMPC_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  c->solB = 0;   // overwrites the ctx's relu masks and objectives: any held solution is gone
  if (c->dynl) {
    DynlTrajArgs d = base_dynl(c, B, goal);
    d.x0 = x0; d.U = U; d.X = X; d.costs = costs; d.obj = c->obj;
    {
      ProfScope ps(c, PROF_ROLLOUT, s);
      gmpc_launch_dynl_rollout(d, s);
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  TrajArgs a = base_traj(c, B, goal);
  a.x0 = x0; a.U = U; a.X = X; a.costs = costs; a.obj = c->obj; a.masks = c->masks;
  {
    ProfScope ps(c, PROF_ROLLOUT, s);
    gmpc_launch_rollout(a, s);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// linearise + terminal quadratisation + Riccati/adjoint sweep on (X, U)
// lin_list / lin_join (optional, gmpc_ilqr_solve): the Jacobian chain of some of the active trajectories has been
// started elsewhere -- the chain here covers the trajectories lin_list->llist[0 .. *lcount), and the sweep waits for
// lin_join
static int backward_pass(gmpc_ctx* c, int B, const float* X, const float* U, const float* goal,
                         const int* active, float* K, float* k, float* grad, float* adj, float* AB,
                         int* cont, const gmpc_ilqr_opts* opts, hipStream_t s, const LsSplit* lin_list = nullptr,
                         hipEvent_t lin_join = nullptr) {
  const gmpc_shape& sh = c->sh;
  if (c->big) {
    if (c->lin_event) HIP_TRY(hipEventRecord(c->lin_event, s));
    // large state: terminal quadratisation, then the step-major MFMA pipeline (gmpc_large.hip)
    if (gmpc_launch_terminal(B, sh.T, sh.n, c->cost, c->mpc_w, X, active, c->QT, c->qT, s) != 0)
      return fail(GMPC_EINVAL, "terminal: unsupported fout");
    HIP_TRY(hipGetLastError());
    {
      ProfScope ps(c, PROF_RICCATI, s);
      // the gains feed the GEMMs as a padded operand: always build them in the ctx buffer
      if (gmpc_big_backward(c->bw, B, c->dyn, c->lp, c->masks, X, U, goal, c->mpc_w, c->QT, c->qT, active,
                            c->Ks, k, grad ? grad : c->grads, adj ? adj : c->adjs, nullptr, nullptr, s,
                            c->dynl ? &c->dl : nullptr) != 0)
        return fail(GMPC_EINVAL, "large-state backward: Jacobian kernel does not cover this shape");
      if (K != c->Ks)
        HIP_TRY(hipMemcpyAsync(K, c->Ks, (size_t)B * sh.T * sh.m * sh.n * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
    }
    if (cont)
      gmpc_launch_big_cont(B, sh.T, sh.m, U, c->bw.gn2, c->iters, c->obj, c->alpha, c->obj_step,
                           c->U_step, *opts, active, cont, s);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // The terminal quadratisation needs X only: it runs BEFORE the Jacobian chain, so that the Riccati sweep is the
  // launch right behind the chain.  With the critic step on a second stream gated by lin_event, the sweep's
  // one-wave workgroups are then dispatched first and take the low end of every SIMD's register file; launched
  // 0.03 ms later (behind k_terminal) they landed BETWEEN the critic's waves, and the 272-register waves of
  // k_lstm_bwd2 found no contiguous block until the sweep had finished (0.105 -> 0.24 ms for that kernel).
  static const bool terminal_first = !(getenv("GMPC_TERMINAL_FIRST") && getenv("GMPC_TERMINAL_FIRST")[0] == '0');
  auto run_terminal = [&]() -> int {
    ProfScope ps(c, PROF_TERMINAL, s);
    if (gmpc_launch_terminal(B, sh.T, sh.n, c->cost, c->mpc_w, X, active, c->QT, c->qT, s) != 0)
      return fail(GMPC_EINVAL, "terminal: unsupported fout");
    return 0;
  };
  if (terminal_first) TRY(run_terminal());
  bool lin_event_done = false;
  {
    ProfScope ps(c, PROF_LINEARIZE, s);
    // matrix-core chain; the VALU chain only serves shapes the MFMA tiling does not cover (or
    // GMPC_LINEARIZE=valu, kept for A/B timing) -- both are HIP kernels of this library
    static const int force = []() {
      const char* e = getenv("GMPC_LINEARIZE");
      return !e ? 0 : strcmp(e, "valu") == 0 ? 2 : strcmp(e, "lds") == 0 ? 1 : 0;
    }();
    // 1st choice: register-resident chain (compiled for the common equal-width shapes), 2nd: the
    // LDS-operand chain (any shape), 3rd: VALU
    int rc_regs = -1;
    if (c->dynl) {
      gmpc_launch_dynl_jac(B, sh.T, sh.T, 0, c->dl, X, U, active, AB, s);
      c->lin_kernel = "k_dynl_jac";
    } else if (force == 0 && lin_list != nullptr) {
      if (gmpc_launch_linearize_regs_list(B, sh.T, sh.n, sh.m, c->dyn, c->lp, c->masks, lin_list->llist, lin_list->lcount,
                                          AB, 0, s) < 0)
        return fail(GMPC_EINVAL, "linearize: the register-resident chain refused a shape it covers");
    } else if (force == 0 && (rc_regs = gmpc_launch_linearize_regs(B * sh.T, sh.T, sh.n, sh.m, c->dyn, c->lp, c->masks,
                                                                   active, AB, 1, 0, s, c->lin_event)) >= 0) {
      // (rc 1: the caller's event sits between the chain's full rounds and its ragged last round)
      snprintf(c->lin_kernel_buf, sizeof(c->lin_kernel_buf), "%s", gmpc_linearize_regs_last_name());
      c->lin_kernel = c->lin_kernel_buf;
      lin_event_done = rc_regs == 1;
    } else if (force == 2 ||
        gmpc_launch_linearize_mfma(B * sh.T, sh.T, sh.n, sh.m, c->dyn, c->lp, c->masks, active, AB, 1, 0,
                                   s) != 0) {
      if (gmpc_launch_linearize(B, sh.T, sh.n, sh.m, c->dyn, c->masks, active, AB, s) != 0)
        return fail(GMPC_EINVAL, "linearize: unsupported row count for n=%d", sh.n);
      c->lin_kernel = "k_linearize (vector ALU)";
    } else {
      c->lin_kernel = "k_linearize_mfma";
    }
  }
  HIP_TRY(hipGetLastError());
  if (lin_join) HIP_TRY(hipStreamWaitEvent(s, lin_join, 0));
  if (c->lin_event && !lin_event_done) HIP_TRY(hipEventRecord(c->lin_event, s));     // gmpc_set_linearize_event
  if (!terminal_first) TRY(run_terminal());
  HIP_TRY(hipGetLastError());
  RiccatiArgs r;
  memset(&r, 0, sizeof(r));
  r.B = B; r.n = sh.n; r.ng = c->nx; r.m = sh.m; r.T = sh.T; r.mode = 0;
  r.X = X; r.U = U; r.goal = goal; r.mpc_w = c->mpc_w; r.AB = AB; r.QT = c->QT; r.qT = c->qT;
  r.active = active; r.K = K; r.k = k; r.grad = grad; r.adj = adj;
  if (cont) {
    r.cont = cont; r.iters = c->iters; r.obj = c->obj; r.alpha = c->alpha;
    r.obj_step = c->obj_step; r.U_step = c->U_step; r.opts = *opts;
  }
  {
    ProfScope ps(c, PROF_RICCATI, s);
    gmpc_launch_riccati(r, s);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gmpc_lqr_backward(gmpc_ctx* c, int B, const float* X, const float* U,
                                 const float* goal, float* K, float* k, float* grad,
                                 float* adjoints, float* AB, void* stream) {
  TRY(check_call(c, B));
  if (!X || !U || !goal) return fail(GMPC_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->big && AB) return fail(GMPC_EINVAL, "AB output is not materialised for n > 64 (pass NULL)");
  c->solB = 0;   // overwrites masks, QT/qT and (with NULL outputs) the ctx's K / AB
  // relu masks at (X, U): recomputed so that any trajectory may be passed (the LSTM variant's Jacobian
  // kernel recomputes its forward pass itself)
  if (!c->dynl) gmpc_launch_masks(B, c->sh.n, c->sh.m, c->sh.T, c->dyn, X, U, c->masks, s);
  return backward_pass(c, B, X, U, goal, nullptr, K ? K : c->Ks, k ? k : c->ks, grad, adjoints,
                       AB ? AB : c->AB, nullptr, nullptr, s);
}

// Same as gmpc_lqr_backward but reuses the relu masks the preceding gmpc_rollout_cost of this ctx
// produced for exactly this (X, U) -- the rollout+backward "step" timed by bench.py.
extern "C" int gmpc_lqr_backward_after_rollout(gmpc_ctx* c, int B, const float* X, const float* U,
                                               const float* goal, float* K, float* k, float* grad,
                                               float* adjoints, float* AB, void* stream) {
  TRY(check_call(c, B));
  if (!X || !U || !goal) return fail(GMPC_EINVAL, "null argument");
  if (c->big && AB) return fail(GMPC_EINVAL, "AB output is not materialised for n > 64 (pass NULL)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  c->solB = 0;   // overwrites QT/qT and (with NULL outputs) the ctx's K / AB
  return backward_pass(c, B, X, U, goal, nullptr, K ? K : c->Ks, k ? k : c->ks, grad, adjoints,
                       AB ? AB : c->AB, nullptr, nullptr, s);
}

// line-search candidate evaluation of the LSTM dynamics variant: same (trajectory, halving) work list and
// the same decide / commit kernels as the MLP path, the rollouts by k_dynl_traj<true>
static void dynl_ls_eval(void* user, const TrajArgs& t, int max_items, hipStream_t s) {
  gmpc_ctx* c = static_cast<gmpc_ctx*>(user);
  DynlTrajArgs d = base_dynl(c, t.B, t.goal);
  d.item_b = t.item_b; d.item_k = t.item_k; d.nitems = t.nitems;
  d.Xn = t.X; d.Un = t.Uio; d.Kg = t.Kg; d.kg = t.kg;
  d.Xc = t.Xc; d.Uc = t.Uc; d.objc = t.objc; d.alpha_0 = t.alpha_0;
  gmpc_launch_dynl_candidates(d, max_items, s);
}

extern "C" int gmpc_ilqr_solve(gmpc_ctx* c, int B, const float* x0, const float* U_init,
                               const float* goal, const gmpc_ilqr_opts* opts, float* X, float* U,
                               float* obj, float* grad, float* adjoints, int* iterations,
                               void* stream) {
  TRY(check_call(c, B));
  if (!x0 || !U_init || !goal || !opts) return fail(GMPC_EINVAL, "null argument");
  if (opts->make_psd) return fail(GMPC_EINVAL, "make_psd=1 is not on the reference path");
  hipStream_t s = static_cast<hipStream_t>(stream);
  c->solB = 0;   // restored only when the solve has completed (an early error return leaves none)
  const gmpc_shape& sh = c->sh;
  const size_t n = sh.n, m = sh.m, T = sh.T;
  HIP_TRY(hipMemcpyAsync(c->Us, U_init, B * T * m * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(c->goals, goal, B * (T + 1) * (size_t)c->nx * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemsetAsync(c->iters, 0, B * sizeof(int), s));
  // alpha = alpha_0, steps = +inf
  std::vector<float> init(3 * (size_t)B);
  for (int b = 0; b < B; ++b) {
    init[b] = opts->alpha_0;
    init[B + b] = INFINITY;
    init[2 * (size_t)B + b] = INFINITY;
  }
  HIP_TRY(hipMemcpyAsync(c->alpha, init.data(), B * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(c->obj_step, init.data() + B, B * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(c->U_step, init.data() + 2 * (size_t)B, B * sizeof(float),
                         hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // `init` goes out of scope below
  if (c->dynl) {
    DynlTrajArgs d = base_dynl(c, B, c->goals);
    d.x0 = x0; d.U = c->Us; d.X = c->Xs; d.costs = nullptr; d.obj = c->obj;
    gmpc_launch_dynl_rollout(d, s);
  } else {
    TrajArgs a = base_traj(c, B, c->goals);
    a.x0 = x0; a.U = c->Us; a.X = c->Xs; a.costs = nullptr; a.obj = c->obj; a.masks = c->masks;
    gmpc_launch_rollout(a, s);
  }
  TRY(backward_pass(c, B, c->Xs, c->Us, c->goals, nullptr, c->Ks, c->ks, c->grads, c->adjs, c->AB,
                    c->cont, opts, s));
  TrajArgs ls = base_traj(c, B, c->goals);
  ls.X = c->Xs; ls.Uio = c->Us; ls.obj = c->obj; ls.masks = c->masks; ls.Kg = c->Ks; ls.kg = c->ks;
  ls.Xc = c->Xc; ls.Uc = c->Uc; ls.maskc = c->maskc; ls.active = c->cont; ls.alpha = c->alpha;
  ls.obj_step = c->obj_step; ls.U_step = c->U_step; ls.iters = c->iters;
  ls.alpha_0 = opts->alpha_0; ls.alpha_min = opts->alpha_min;
  // a fresh solve starts its first line search with a single full step per trajectory
  HIP_TRY(hipMemsetAsync(c->lsw.prevk, 0, B * sizeof(int), s));
  HIP_TRY(hipMemsetAsync(c->lsw.counts + GMPC_LS_ROUNDS_MAX, 0, (1 + GMPC_LS_STATS) * sizeof(int), s));
  // "Has every trajectory stopped?" is answered without stalling the queue: the continuation flags of
  // iteration `it` are copied to a pinned ring slot when the iteration is enqueued and looked at
  // GMPC_POLL_DEPTH iterations later, so the host runs at most that many iterations ahead of what it
  // knows.  Iterations enqueued after the last trajectory stopped are exact no-ops (every kernel of the
  // loop is masked by the same flags), at most GMPC_POLL_DEPTH of them.
  if (!c->hcont) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->hcont), (size_t)GMPC_POLL_DEPTH * c->maxB * sizeof(int),
                          hipHostMallocDefault));
    for (int i = 0; i < GMPC_POLL_DEPTH; ++i) HIP_TRY(hipEventCreateWithFlags(&c->poll_ev[i], hipEventDisableTiming));
  }
  // OPT-IN (GMPC_LS_OVERLAP=1), measured in round 4 and not a gain yet (profiles/EXPERIMENTS.md): the second round of
  // a line search is a short work list (C3: 2.8 k candidates = 176 of k_ls16's workgroups, one per CU, on 256 CUs),
  // while a third of the trajectories are done after the first round.  Up to `cap` of those go, behind the first
  // round's decision, to a Jacobian chain of their own on a side stream, in GMPC_EARLY_CUS workgroups of eight waves,
  // each the whole register file of a CU -- k_ls16 needs a whole CU per workgroup too, so the two kernels share no CU --
  // sized to finish with the second round; the main stream runs the chain of the other trajectories behind the search
  // (C3: 12 rounds of tiles instead of 14, 1.01 ms instead of 1.15) and waits for the side stream in front of the
  // sweep.  Same kernels on the same data, only partitioned: results are bitwise those of the single launch
  // (tests/test_gpu_control_flow.py).  When the second round is not one pass of k_ls16 that leaves those CUs idle the
  // early list stays empty (k_ls_split decides on the device).  The kernels of an iteration add up to 0.09 ms less,
  // the iteration takes 0.02 ms MORE: the fork, the join and the split cost what the shorter chain saves.
  const bool overlap_off = !(getenv("GMPC_LS_OVERLAP") && getenv("GMPC_LS_OVERLAP")[0] == '1') ||    // (per call: the
                           getenv("GMPC_LINEARIZE") != nullptr;                                    // tests switch it)
  constexpr int GMPC_EARLY_CUS = 80;
  int early_cap = 0;
  if (!overlap_off && !c->big && !c->dynl && gmpc_linearize_regs_covers(sh.n, sh.m, c->dyn, c->lp) &&
      gmpc_ls_rounds_on_ls16(ls)) {
    // k_ls16 takes ~9.2 us per step of the horizon (C3: 0.46 ms), a 32-row tile of the 200-wide chain 81 us on a
    // SIMD it shares with a second wave: tiles a wave slot of the side stream finishes inside the second round
    const int tiles_per_slot = (int)(0.1136 * (double)T);
    early_cap = (int)((long)GMPC_EARLY_CUS * 8 * tiles_per_slot * 32 / ((long)T * n));
    if (early_cap > B) early_cap = B;
  }
  const bool overlap = early_cap > 0;
  if (overlap && !c->ls_side) {
    TRY(dalloc(c, &c->ls_split.tlist, c->maxB));
    TRY(dalloc(c, &c->ls_split.tcount, 1));
    TRY(dalloc(c, &c->ls_split.llist, c->maxB));
    TRY(dalloc(c, &c->ls_split.lcount, 1));
    // (an ordinary stream: on a stream created with a CU mask the chain did not start before the main stream's queue
    // had run dry, and every launch of the main stream grew an 8 us gap -- profiles/EXPERIMENTS.md, round 4)
    HIP_TRY(hipStreamCreateWithFlags(&c->ls_side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ls_split.ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ls_joined, hipEventDisableTiming));
  }
  if (overlap) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    c->ls_split.cap = early_cap;
    c->ls_split.wg_max = ncu - GMPC_EARLY_CUS;
    if (getenv("GMPC_LS_EARLY_WGMAX")) c->ls_split.wg_max = atoi(getenv("GMPC_LS_EARLY_WGMAX"));                // (diagnostic)
  }
  for (int it = 0; it < opts->maxiter; ++it) {
    const int slot = it % GMPC_POLL_DEPTH;
    int* hc = c->hcont + (size_t)slot * c->maxB;
    if (it >= GMPC_POLL_DEPTH) {
      HIP_TRY(hipEventSynchronize(c->poll_ev[slot]));     // flags as of iteration it - GMPC_POLL_DEPTH
      bool any = false;
      for (int b = 0; b < B; ++b) any |= hc[b] != 0;
      if (!any) break;
    }
    HIP_TRY(hipMemcpyAsync(hc, c->cont, B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(c->poll_ev[slot], s));
    {
      ProfScope ps(c, PROF_LINESEARCH, s);
      if (gmpc_launch_linesearch(ls, c->lsw, s, c->dynl ? &dynl_ls_eval : nullptr, c,
                                 overlap ? &c->ls_split : nullptr) != 0)
        return fail(GMPC_EINVAL, "line search: alpha_0 / alpha_min need more than %d rounds",
                    GMPC_LS_ROUNDS_MAX);
    }
    if (overlap) {
      HIP_TRY(hipStreamWaitEvent(c->ls_side, c->ls_split.ev, 0));
      {
        ProfScope ps(c, PROF_LINEARIZE, c->ls_side);
        if (gmpc_launch_linearize_regs_list(early_cap, sh.T, sh.n, sh.m, c->dyn, c->lp, c->masks, c->ls_split.tlist,
                                            c->ls_split.tcount, c->AB, GMPC_EARLY_CUS, c->ls_side) < 0)
          return fail(GMPC_EINVAL, "linearize: the register-resident chain refused a shape it covers");
      }
      HIP_TRY(hipEventRecord(c->ls_joined, c->ls_side));
    }
    TRY(backward_pass(c, B, c->Xs, c->Us, c->goals, c->cont, c->Ks, c->ks, c->grads, c->adjs, c->AB,
                      c->cont, opts, s, overlap ? &c->ls_split : nullptr, overlap ? c->ls_joined : nullptr));
  }
  if (X) HIP_TRY(hipMemcpyAsync(X, c->Xs, B * (T + 1) * n * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (U) HIP_TRY(hipMemcpyAsync(U, c->Us, B * T * m * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (obj) HIP_TRY(hipMemcpyAsync(obj, c->obj, B * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (grad) HIP_TRY(hipMemcpyAsync(grad, c->grads, B * T * m * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (adjoints)
    HIP_TRY(hipMemcpyAsync(adjoints, c->adjs, B * (T + 1) * n * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (iterations)
    HIP_TRY(hipMemcpyAsync(iterations, c->iters, B * sizeof(int), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  c->solB = B;
  return 0;
}

// critic ---------------------------------------------------------------------------------------
static int bind_critic(gmpc_ctx* c, const float* critic, CriticDesc& cd, hipStream_t s, bool head_transpose = true) {
  const gmpc_shape& sh = c->sh;
  if (sh.lstm_features <= 0) return fail(GMPC_EINVAL, "this ctx was created without a critic");
  const long n = c->nx, F = sh.lstm_features;      // the critic scores x sequences
  cd.n = c->nx; cd.F = sh.lstm_features; cd.T1 = sh.T + 1;
  cd.Wcat = critic;
  cd.WcatT = c->critT;
  cd.b = critic + (n + F) * 4 * F;
  bind_mlp(cd.head, sh.head_layers, sh.head_dims, critic + (n + F) * 4 * F + 4 * F,
           c->critT + (n + F) * 4 * F);
  // the first-generation LSTM kernels read [Wx; Wh]^T; the second generation (n <= 32) and the head's forward
  // layers read the parameters as they lie, the head's backward layers its transposed kernels (one launch)
  if (!(c->lwp != nullptr && gmpc_lstm2_supported(cd)) || c->xT != nullptr)
    gmpc_launch_transpose((int)(n + F), (int)(4 * F), cd.Wcat, c->critT, s);
  if (head_transpose) gmpc_launch_mlp_transpose_all(cd.head, s);     // (else: the caller, beside the LSTM forward sweep)
  return 0;
}

static int critic_forward_backward(gmpc_ctx* c, int Bc, const float* xseq, const float* label,
                                   const float* critic, int loss_kind, float* dxseq, bool want_wgrad,
                                   float* grad_sum, hipStream_t s, float* loss_sum = nullptr) {
  // The side stream of the critic step (GMPC_CRITIC_SIDE=0: everything on the caller's stream): the transposed head
  // kernels are built beside the LSTM forward sweep (only k_head2 reads them), the head's weight gradients and the
  // loss sum run beside the BPTT sweep (they need k_head2's outputs only; the sweep is a latency chain at one wave
  // per SIMD).
  const char* side_env = getenv("GMPC_CRITIC_SIDE");
  const bool side_on = !(side_env != nullptr && side_env[0] == '0') && c->xT == nullptr && c->lwp != nullptr;
  if (side_on && !c->crit_side) {
    HIP_TRY(hipStreamCreateWithFlags(&c->crit_side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->crit_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->crit_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->crit_tr, hipEventDisableTiming));
  }
  // (an error return between a fork onto the side stream and its join must not leave work in flight there)
  struct SideGuard {
    hipStream_t st = nullptr;
    ~SideGuard() { if (st) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); } }
  } side_guard;
  CriticDesc cd;
  TRY(bind_critic(c, critic, cd, s, !side_on));
  const bool tr_side = side_on && gmpc_lstm2_supported(cd);
  if (side_on && !tr_side) gmpc_launch_mlp_transpose_all(cd.head, s);
  if (tr_side) {
    HIP_TRY(hipEventRecord(c->crit_fork, s));
    HIP_TRY(hipStreamWaitEvent(c->crit_side, c->crit_fork, 0));
    side_guard.st = c->crit_side;
    gmpc_launch_mlp_transpose_all(cd.head, c->crit_side);
    HIP_TRY(hipEventRecord(c->crit_tr, c->crit_side));
  }
  const gmpc_shape& sh = c->sh;
  const int n = c->nx, F = sh.lstm_features, T1 = sh.T + 1;
  // wide inputs (n + F > 256): x_t Wx for all steps is one MFMA GEMM up front and the LSTM kernels
  // run on the recurrent half only (cr: n = 0, Wcat = Wh); dx comes back through a second GEMM
  const bool widein = c->xT != nullptr;
  const int R = Bc * T1, G4w = 4 * F;
  CriticDesc cr = cd;
  auto gemm1 = [&](int M, int N, int K, const float* X, int ldx, const float* Y, int ldy, float* Cp,
                   int ldc) {
    BgemmArgs g;
    g.batch = 1; g.M = M; g.N = N; g.K = K;
    g.X = X; g.sx = 0; g.ldx = ldx; g.Y = Y; g.sy = 0; g.ldy = ldy; g.C = Cp; g.sc = 0; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.active = nullptr;
    gmpc_launch_bgemm_tn(g, s);
  };
  if (widein) {
    cr.n = 0;
    cr.Wcat = critic + (long)n * G4w;
    cr.WcatT = c->WhT;
    gmpc_launch_transpose(F, G4w, cr.Wcat, c->WhT, s);
  }
  const bool gen2 = !widein && c->lwp != nullptr && gmpc_lstm2_supported(cd);
  if (gen2) {
    ProfScope ps(c, PROF_LSTM_FWD, s);
    gmpc_launch_lstm_fwd2(Bc, cd, xseq, c->gates, c->cs, c->hp, c->hT, s);
  } else {
    ProfScope ps(c, PROF_LSTM_FWD, s);
    if (widein) {
      gmpc_launch_transpose(R, n, xseq, c->xT, s);                       // [R][n] -> [n][R]
      gemm1(R, G4w, n, c->xT, R, critic, G4w, c->xproj, G4w);             // xproj = x Wx
    }
    gmpc_launch_lstm_fwd(Bc, cr, xseq, c->gates, c->cs, c->hp, c->hT, widein ? c->xproj : nullptr, s);
  }
  if (tr_side) {
    HIP_TRY(hipStreamWaitEvent(s, c->crit_tr, 0));
    side_guard.st = nullptr;        // joined
  }
  {
    ProfScope ps(c, PROF_HEAD, s);
    gmpc_launch_head2(Bc, cd, loss_kind, c->hT, label, c->cscore, c->closs, c->hacts, c->hdels, c->plast, c->dhT,
                      c->hstride, s);
  }
  hipStream_t sw = s;
  bool forked = false;
  if (gen2 && want_wgrad && side_on) {
    HIP_TRY(hipEventRecord(c->crit_fork, s));
    HIP_TRY(hipStreamWaitEvent(c->crit_side, c->crit_fork, 0));
    sw = c->crit_side;
    forked = true;
    side_guard.st = c->crit_side;
  }
  float* gWx0 = grad_sum;
  if (gen2 && (dxseq || want_wgrad)) {
    // backward sweep with the LSTM weight gradients accumulated in registers (no dz in memory), then the
    // reduction of the per-workgroup partials straight into grad_sum
    ProfScope ps(c, PROF_LSTM_BWD, s);
    float* gWh0 = want_wgrad ? gWx0 + (long)n * 4 * F : nullptr;
    gmpc_launch_lstm_bwd2(Bc, cd, xseq, c->gates, c->cs, c->hp, c->dhT, want_wgrad ? c->lwp : nullptr, gWx0, gWh0,
                          want_wgrad ? gWh0 + (long)F * 4 * F : nullptr, dxseq, s);
  } else if (dxseq || want_wgrad) {
    ProfScope ps(c, PROF_LSTM_BWD, s);
    if (!widein) {
      gmpc_launch_lstm_bwd(Bc, cd, c->gates, c->cs, c->dhT, want_wgrad ? c->dz : nullptr, dxseq, s);
    } else {
      gmpc_launch_lstm_bwd(Bc, cr, c->gates, c->cs, c->dhT, c->dz, nullptr, s);
      if (dxseq) {
        gmpc_launch_transpose(R, G4w, c->dz, c->xproj, s);               // dz^T: [4F][R]
        gemm1(R, n, G4w, c->xproj, R, c->critT, n + F, dxseq, n);         // dx = dz Wx^T
      }
    }
  }
  if (want_wgrad) {
    ProfScope ps(c, PROF_WGRAD, sw);
    const int rows = Bc * T1, G4 = 4 * F;
    float* gWx = grad_sum;
    float* gWh = gWx + (long)n * G4;
    float* gb = gWh + (long)F * G4;
    // every problem with N % 256 == 0 goes into one batched launch (+ one reduction launch)
    WgProb pr[GMPC_WG_MAX];
    int np = 0;
    auto add = [&](int r, int M, int N, const float* A, int lda, const float* Bm, int ldb, float* Cw,
                   float* cs, int cs_rows) {
      WgProb q{};
      q.rows = r; q.M = M; q.N = N; q.lda = lda; q.ldb = ldb; q.cs_rows = cs_rows;
      q.A = A; q.B = Bm; q.C = Cw; q.colsum = cs;
      pr[np++] = q;
    };
    struct Single { int r, M, N; const float* A; int lda; const float* Bm; int ldb; float* Cw; float* cs; int csr; };
    Single single[GMPC_MAX_LAYERS + 2];
    int ns = 0;
    auto route = [&](int r, int M, int N, const float* A, int lda, const float* Bm, int ldb, float* Cw,
                     float* cs, int cs_rows) {
      if (N % 256 == 0 && r >= 64 && np < GMPC_WG_MAX) add(r, M, N, A, lda, Bm, ldb, Cw, cs, cs_rows);
      else single[ns++] = Single{r, M, N, A, lda, Bm, ldb, Cw, cs, cs_rows};
    };
    if (!gen2) {
      route(rows, n, G4, xseq, n, c->dz, G4, gWx, nullptr, 0);
      route(rows, F, G4, c->hp, F, c->dz, G4, gWh, gb, rows);
    }
    float* gh = gb + G4;
    int aoff = 0, doff = 0;
    for (int l = 0; l < sh.head_layers; ++l) {
      const int M = sh.head_dims[l], N = sh.head_dims[l + 1];
      if (l == sh.head_layers - 1 && np < GMPC_WG_MAX) {
        // the last layer has one output: its weight gradient and its bias gradient are the column sums of
        // k_head2's products [act * dscore | dscore] -- a problem without a GEMM part (M = 0)
        add(Bc, 0, M + 1, c->plast, GMPC_HEAD2_LD, c->plast, GMPC_HEAD2_LD, gh, gh, Bc);
      } else {
        route(Bc, M, N, c->hacts + aoff, c->hstride, c->hdels + doff, c->hstride, gh, gh + (long)M * N, Bc);
      }
      gh += (long)M * N + N;
      aoff += M;
      doff += N;
    }
    if (np > 0 && !gmpc_launch_wgrad_batch(pr, np, c->wpart, c->wpart_floats, sw)) {
      for (int i = 0; i < np; ++i)
        single[ns++] = Single{pr[i].rows, pr[i].M, pr[i].N, pr[i].A, pr[i].lda, pr[i].B, pr[i].ldb, pr[i].C,
                              pr[i].colsum, pr[i].cs_rows};
    }
    // the rest one by one, after the batch (they reuse the partial-sum buffer: stream order)
    for (int i = 0; i < ns; ++i) {
      if (single[i].M == 0)
        gmpc_launch_colsum(single[i].csr, single[i].N, single[i].Bm, single[i].ldb, single[i].cs, c->wpart, sw);
      else
        gmpc_launch_wgrad(single[i].r, single[i].M, single[i].N, single[i].A, single[i].lda, single[i].Bm,
                          single[i].ldb, single[i].Cw, single[i].cs, single[i].csr, c->wpart, 256, sw,
                          c->wpart_floats, true);
    }
  }
  if (loss_sum) gmpc_launch_sum(Bc, c->closs, loss_sum, 0, sw);
  if (forked) {
    HIP_TRY(hipEventRecord(c->crit_join, sw));
    HIP_TRY(hipStreamWaitEvent(s, c->crit_join, 0));
    side_guard.st = nullptr;        // joined
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gmpc_critic_loss_grad(gmpc_ctx* c, int Bc, const float* xseq, const float* label,
                                     const float* critic, float* loss_sum, float* grad_sum,
                                     void* stream) {
  if (!c) return fail(GMPC_EINVAL, "ctx is null");
  if (Bc < 1 || Bc > 2 * c->maxB) return fail(GMPC_EINVAL, "Bc=%d outside [1, 2*max_batch]", Bc);
  if (!xseq || !label || !critic || !loss_sum || !grad_sum) return fail(GMPC_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  hipStream_t s = static_cast<hipStream_t>(stream);
  TRY(critic_forward_backward(c, Bc, xseq, label, critic, 0, nullptr, true, grad_sum, s, loss_sum));
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gmpc_critic_score_vjp(gmpc_ctx* c, int Bc, const float* xseq, const float* critic,
                                     float* score, float* dxseq, void* stream) {
  if (!c) return fail(GMPC_EINVAL, "ctx is null");
  if (Bc < 1 || Bc > 2 * c->maxB) return fail(GMPC_EINVAL, "Bc=%d outside [1, 2*max_batch]", Bc);
  if (!xseq || !critic || !score) return fail(GMPC_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  hipStream_t s = static_cast<hipStream_t>(stream);
  TRY(critic_forward_backward(c, Bc, xseq, nullptr, critic, 2, dxseq, false, nullptr, s));
  HIP_TRY(hipMemcpyAsync(score, c->cscore, Bc * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

// upper-level loss only, at the solution held by the ctx (norm/cost_trainer.py:13-21 test loss)
static int upper_loss(gmpc_ctx* c, int B, int loss_kind, const float* desired, const float* critic,
                      float* loss, bool want_lx, hipStream_t s) {
  const gmpc_shape& sh = c->sh;
  if (loss_kind == 0) {
    if (!desired) return fail(GMPC_EINVAL, "desired is null");
    gmpc_launch_l2loss(B, sh.T, sh.n, c->nx, c->Xs, desired, loss, c->lx, s);
  } else if (loss_kind == 1) {
    if (!critic) return fail(GMPC_EINVAL, "critic is null");
    if (c->dynl) {
      // the critic sees the x columns of xc (gan/js_policy.py:64-65); its input gradient goes back into
      // those columns, zero on the carry
      const long rows = (long)B * (sh.T + 1);
      gmpc_launch_cols_gather(rows, sh.n, c->nx, c->Xs, c->xg, s);
      TRY(critic_forward_backward(c, B, c->xg, nullptr, critic, 1, want_lx ? c->lxg : nullptr, false,
                                  nullptr, s));
      if (want_lx) gmpc_launch_cols_scatter(rows, sh.n, c->nx, c->lxg, c->lx, s);
    } else
    TRY(critic_forward_backward(c, B, c->Xs, nullptr, critic, 1, want_lx ? c->lx : nullptr, false,
                                nullptr, s));
    HIP_TRY(hipMemcpyAsync(loss, c->closs, B * sizeof(float), hipMemcpyDeviceToDevice, s));
  } else {
    return fail(GMPC_EINVAL, "loss_kind must be 0 (L2) or 1 (JS)");
  }
  return 0;
}

extern "C" int gmpc_upper_loss(gmpc_ctx* c, int B, int loss_kind, const float* desired,
                               const float* critic, float* loss, void* stream) {
  TRY(check_call(c, B));
  if (c->solB != B) return fail(GMPC_EINVAL, "gmpc_ilqr_solve with B=%d must precede this call", B);
  if (!loss) return fail(GMPC_EINVAL, "null argument");
  TRY(upper_loss(c, B, loss_kind, desired, critic, loss, false, static_cast<hipStream_t>(stream)));
  HIP_TRY(hipGetLastError());
  return 0;
}

// expert sequence model (N2) --------------------------------------------------------------------
int gmpc_launch_expert(const ExpertArgs&, hipStream_t);

static int check_expert_shape(const gmpc_expert_shape* es, int n, int m) {
  if (!es) return fail(GMPC_EINVAL, "expert shape is null");
  if (es->head_layers < 1 || es->head_layers > GMPC_MAX_LAYERS)
    return fail(GMPC_EINVAL, "expert head_layers=%d outside [1, %d]", es->head_layers, GMPC_MAX_LAYERS);
  if (es->lstm_features < 0 || es->lstm_features > 128)
    return fail(GMPC_EINVAL, "expert lstm_features=%d outside [0, 128]", es->lstm_features);
  const int L = es->head_layers;
  if (es->head_dims_x[L] != n || es->head_dims_u[L] != m)
    return fail(GMPC_EINVAL, "expert heads must end in n=%d and m=%d", n, m);
  if (es->head_dims_x[0] != es->head_dims_u[0] ||
      (es->lstm_features > 0 && es->head_dims_x[0] != es->lstm_features))
    return fail(GMPC_EINVAL, "expert heads must start at the width of y");
  for (int l = 0; l <= L; ++l)
    if (es->head_dims_x[l] < 1 || es->head_dims_x[l] > 1024 || es->head_dims_u[l] < 1 ||
        es->head_dims_u[l] > 1024)
      return fail(GMPC_EINVAL, "expert head widths must be in [1, 1024]");
  return 0;
}

extern "C" long gmpc_expert_param_count(int n, const gmpc_expert_shape* es) {
  if (!es || es->head_layers < 1 || es->head_layers > GMPC_MAX_LAYERS) return -1;
  const long F = es->lstm_features, h = es->head_dims_x[0];
  long cnt = F > 0 ? (n + F) * 4 * F + 4 * F : (long)n * h + h;
  return cnt + mlp_count(es->head_layers, es->head_dims_x) + mlp_count(es->head_layers, es->head_dims_u);
}

extern "C" int gmpc_expert_rollout(gmpc_ctx* c, int B, int hist, const gmpc_expert_shape* es,
                                   const float* expert, const float* history, float* goal, float* init_U,
                                   void* stream) {
  TRY(check_call(c, B, false));     // the expert model has its own parameters
  const gmpc_shape& sh = c->sh;
  const int nx = c->nx;     // the expert model predicts x sequences (goals have x_size columns)
  TRY(check_expert_shape(es, nx, sh.m));
  if (hist < 1) return fail(GMPC_EINVAL, "hist=%d: at least one history row is needed (yaml: history >= 1)", hist);
  if (!expert || !history || !goal || !init_U) return fail(GMPC_EINVAL, "null argument");
  ExpertArgs a;
  a.B = B; a.n = nx; a.m = sh.m; a.T = sh.T; a.hist = hist; a.F = es->lstm_features;
  const long F = a.F, h = es->head_dims_x[0];
  a.Wcat = expert;
  a.bcat = expert + (F > 0 ? (nx + F) * 4 * F : (long)nx * h);
  const float* heads = a.bcat + (F > 0 ? 4 * F : h);
  bind_mlp(a.hx, es->head_layers, es->head_dims_x, heads, nullptr);
  bind_mlp(a.hu, es->head_layers, es->head_dims_u, heads + mlp_count(es->head_layers, es->head_dims_x),
           nullptr);
  a.history = history; a.goal = goal; a.U = init_U;
  if (gmpc_launch_expert(a, static_cast<hipStream_t>(stream)) != 0)
    return fail(GMPC_EINVAL, "expert kernel: unsupported shape");
  HIP_TRY(hipGetLastError());
  return 0;
}

// dynamics regression (N3) ---------------------------------------------------------------------
extern "C" int gmpc_dynamics_loss_grad(gmpc_ctx* c, int B, int S, const float* xseq, const float* useq,
                                       const float* next_xseq, double discount, int teacher_forcing,
                                       float* loss_sum, float* grad_sum, void* stream) {
  TRY(check_call(c, B));
  const gmpc_shape& sh = c->sh;
  if (S < 1 || S > sh.T) return fail(GMPC_EINVAL, "S=%d outside [1, T=%d]", S, sh.T);
  if (!xseq || !useq || !next_xseq || !loss_sum || !grad_sum) return fail(GMPC_EINVAL, "null argument");
  if (c->dynl) {
    // LSTM variant: BPTT through the cell and the tail (gmpc_dynl.hip); gradient layout Wx | Wh | b | tail
    const DynlDesc& d = c->dl;
    const long Fd = d.F, kin = d.nx + d.m, G4 = 4 * Fd;
    hipStream_t s2 = static_cast<hipStream_t>(stream);
    if (!c->dfacts) {
      const size_t rows = (size_t)c->maxB * sh.T;
      c->dfstride = (int)gmpc_dynl_fit_stride(d);
      int rc = dalloc(c, &c->dfpred, rows * d.nx);
      if (!rc) rc = dalloc(c, &c->dfacts, (rows + 8) * c->dfstride);
      if (!rc) rc = dalloc(c, &c->dfdels, (rows + 8) * c->dfstride);
      if (!rc) rc = dalloc(c, &c->dfsave, rows * 6 * Fd);
      if (!rc) rc = dalloc(c, &c->dfloss, c->maxB);
      if (rc) return rc;
      HIP_TRY(hipMemsetAsync(c->dfacts, 0, (rows + 8) * c->dfstride * sizeof(float), s2));
      HIP_TRY(hipMemsetAsync(c->dfdels, 0, (rows + 8) * c->dfstride * sizeof(float), s2));
    }
    gmpc_launch_dynl_fit(B, S, d, xseq, useq, next_xseq, (float)discount, teacher_forcing != 0, c->dfpred,
                         c->dfacts, c->dfdels, c->dfstride, c->dfsave, c->dfloss, s2);
    const int rows = B * S;
    float* gWx = grad_sum;
    float* gWh = gWx + kin * G4;
    float* gb = gWh + Fd * G4;
    gmpc_launch_wgrad(rows, (int)kin, (int)G4, c->dfacts, c->dfstride, c->dfdels, c->dfstride, gWx, nullptr, 0,
                      c->wpart, 256, s2, c->wpart_floats, true);
    gmpc_launch_wgrad(rows, (int)Fd, (int)G4, c->dfacts + kin, c->dfstride, c->dfdels, c->dfstride, gWh, gb, rows,
                      c->wpart, 256, s2, c->wpart_floats, true);
    float* g = gb + G4;
    int aoff = (int)(kin + Fd), doff = (int)G4;
    for (int l = 0; l < d.tail.L; ++l) {
      const int M = d.tail.dims[l], N = d.tail.dims[l + 1];
      gmpc_launch_wgrad(rows, M, N, c->dfacts + aoff, c->dfstride, c->dfdels + doff, c->dfstride, g,
                        g + (long)M * N, rows, c->wpart, 256, s2, c->wpart_floats, true);
      g += (long)M * N + N;
      aoff += M;
      doff += N;
    }
    gmpc_launch_sum(B, c->dfloss, loss_sum, 0, s2);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!c->dfacts) {
    const size_t rows = (size_t)c->maxB * sh.T;
    c->dfstride = (int)gmpc_dynfit_stride(&sh);
    int rc = dalloc(c, &c->dfpred, rows * sh.n);
    if (!rc) rc = dalloc(c, &c->dfacts, (rows + 8) * c->dfstride);
    if (!rc) rc = dalloc(c, &c->dfdels, (rows + 8) * c->dfstride);
    if (!rc) rc = dalloc(c, &c->dfloss, c->maxB);
    if (rc) return rc;
    // operands of the MFMA weight-gradient GEMM are read a few rows past the end: keep them finite
    HIP_TRY(hipMemsetAsync(c->dfacts, 0, (rows + 8) * c->dfstride * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(c->dfdels, 0, (rows + 8) * c->dfstride * sizeof(float), s));
  }
  if (gmpc_launch_dynfit(B, S, sh.n, sh.m, c->dyn, xseq, useq, next_xseq, (float)discount,
                         teacher_forcing != 0, c->dfpred, c->dfacts, c->dfdels, c->dfstride, c->dfloss,
                         s) != 0)
    return fail(GMPC_EINVAL, "dynamics regression: unsupported layer width");
  const int rows = B * S;
  float* g = grad_sum;
  int aoff = 0, doff = 0;
  for (int l = 0; l < sh.dyn_layers; ++l) {
    const int M = sh.dyn_dims[l], N = sh.dyn_dims[l + 1];
    gmpc_launch_wgrad(rows, M, N, c->dfacts + aoff, c->dfstride, c->dfdels + doff, c->dfstride, g,
                      g + (long)M * N, rows, c->wpart, 256, s, c->wpart_floats, true);
    g += (long)M * N + N;
    aoff += M;
    doff += N;
  }
  gmpc_launch_sum(B, c->dfloss, loss_sum, 0, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

// multi-GPU exchange -----------------------------------------------------------------------------
extern "C" int gmpc_comm_unique_id(char* id128) {
  if (!id128) return fail(GMPC_EINVAL, "null argument");
  return gmpc_comm_unique_id_impl(id128);
}

extern "C" int gmpc_comm_init(gmpc_ctx* c, int world_size, int rank, const char* id128) {
  if (!c || !id128) return fail(GMPC_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  return gmpc_comm_init_impl(&c->comm, world_size, rank, id128);
}

extern "C" int gmpc_allreduce_grads(gmpc_ctx* c, float* packed, long count, void* stream) {
  if (!c || !packed || count < 1) return fail(GMPC_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  return gmpc_comm_allreduce_impl(&c->comm, packed, count, static_cast<hipStream_t>(stream));
}

extern "C" int gmpc_comm_world(gmpc_ctx* c, int* world_size, int* rank) {
  if (!c || !world_size || !rank) return fail(GMPC_EINVAL, "null argument");
  *world_size = c->comm.world;
  *rank = c->comm.rank;
  return 0;
}

// single model evaluations (the reference's model protocol, base.py:4-49) -----------------------
void gmpc_launch_get_cost(int, int, int, int, const MlpDesc&, const float*, const float*, const float*,
                          const float*, int, float*, hipStream_t);

extern "C" int gmpc_get_cost(gmpc_ctx* c, int B, const float* x, const float* u, const float* goal_row,
                             int terminal, float* cost, void* stream) {
  TRY(check_call(c, B));
  if (!x || !cost || (!terminal && (!u || !goal_row))) return fail(GMPC_EINVAL, "null argument");
  gmpc_launch_get_cost(B, c->sh.n, c->nx, c->sh.m, c->cost, c->mpc_w, x, u, goal_row, terminal != 0, cost,
                       static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gmpc_predict(gmpc_ctx* c, int B, const float* x, const float* u, float* next_x,
                            void* stream) {
  TRY(check_call(c, B));
  if (!x || !u || !next_x) return fail(GMPC_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = c->sh.n;
  c->solB = 0;   // the one-step rollout below overwrites the ctx's relu masks and objectives
  // a horizon-1 rollout through the trajectory kernel: X = [x, f(x, u)] in the line-search scratch
  HIP_TRY(hipMemsetAsync(c->goals, 0, (size_t)B * 2 * n * sizeof(float), s));
  if (c->dynl) {
    DynlTrajArgs d = base_dynl(c, B, c->goals);
    d.T = 1;
    d.x0 = x; d.U = u; d.X = c->Xc; d.costs = nullptr; d.obj = c->obj;
    gmpc_launch_dynl_rollout(d, s);
  } else {
    TrajArgs a = base_traj(c, B, c->goals);
    a.T = 1;
    a.x0 = x; a.U = u; a.X = c->Xc; a.costs = nullptr; a.obj = c->obj; a.masks = c->masks;
    gmpc_launch_rollout(a, s);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy2DAsync(next_x, n * sizeof(float), c->Xc + n, 2 * n * sizeof(float), n * sizeof(float),
                           B, hipMemcpyDeviceToDevice, s));
  return 0;
}

// flat parameter layouts, leaf by leaf (flax naming; see params.py of the host package)
static int add_leaf(gmpc_leaf* out, int max_leaves, int& k, const char* name, long off, int rows, int cols,
                    int ld) {
  if (out && k < max_leaves) {
    snprintf(out[k].name, sizeof(out[k].name), "%s", name);
    out[k].offset = off; out[k].rows = rows; out[k].cols = cols; out[k].ld = ld;
  }
  ++k;
  return 0;
}

static long mlp_leaves(gmpc_leaf* out, int max_leaves, int& k, const char* prefix, int L, const int* dims,
                       long off) {
  char nm[64];
  for (int l = 0; l < L; ++l) {
    snprintf(nm, sizeof(nm), "%sparams/Dense_%d/kernel", prefix, l);
    add_leaf(out, max_leaves, k, nm, off, dims[l], dims[l + 1], dims[l + 1]);
    off += (long)dims[l] * dims[l + 1];
    snprintf(nm, sizeof(nm), "%sparams/Dense_%d/bias", prefix, l);
    add_leaf(out, max_leaves, k, nm, off, 1, dims[l + 1], dims[l + 1]);
    off += dims[l + 1];
  }
  return off;
}

// an OptimizedLSTMCell's leaves: Wx [kin][4F], Wh [F][4F], b [4F]; flax' per-gate kernels are the column
// blocks g*F .. (g+1)*F
static long cell_leaves(gmpc_leaf* out, int max_leaves, int& k, const char* prefix, const char* scope, int kin,
                        int F, long off) {
  static const char gate[4] = {'i', 'f', 'g', 'o'};
  char nm[64];
  for (int g = 0; g < 4; ++g) {
    snprintf(nm, sizeof(nm), "%sparams/%s/i%c/kernel", prefix, scope, gate[g]);
    add_leaf(out, max_leaves, k, nm, off + (long)g * F, kin, F, 4 * F);
  }
  off += (long)kin * 4 * F;
  for (int g = 0; g < 4; ++g) {
    snprintf(nm, sizeof(nm), "%sparams/%s/h%c/kernel", prefix, scope, gate[g]);
    add_leaf(out, max_leaves, k, nm, off + (long)g * F, F, F, 4 * F);
  }
  off += (long)F * 4 * F;
  for (int g = 0; g < 4; ++g) {
    snprintf(nm, sizeof(nm), "%sparams/%s/h%c/bias", prefix, scope, gate[g]);
    add_leaf(out, max_leaves, k, nm, off + (long)g * F, 1, F, F);
  }
  return off + 4 * F;
}

static long critic_leaves(gmpc_leaf* out, int max_leaves, int& k, const char* prefix, const gmpc_shape* s,
                          long off) {
  const int nx = s->x_size > 0 ? s->x_size : s->n;
  off = cell_leaves(out, max_leaves, k, prefix, "ScanOptimizedLSTMCell_0", nx, s->lstm_features, off);
  return mlp_leaves(out, max_leaves, k, prefix, s->head_layers, s->head_dims, off);
}

static long dyn_leaves(gmpc_leaf* out, int max_leaves, int& k, const char* prefix, const gmpc_shape* s,
                       long off) {
  if (s->dyn_lstm_features > 0)
    off = cell_leaves(out, max_leaves, k, prefix, "OptimizedLSTMCell_0", s->x_size + s->m,
                      s->dyn_lstm_features, off);
  return mlp_leaves(out, max_leaves, k, prefix, s->dyn_layers, s->dyn_dims, off);
}

extern "C" int gmpc_pack_layout(const gmpc_shape* s, int which, gmpc_leaf* leaves, int max_leaves) {
  TRY(check_shape(s));
  if (max_leaves < 0 || (max_leaves > 0 && !leaves)) return fail(GMPC_EINVAL, "bad leaf buffer");
  int k = 0;
  switch (which) {
    case 0: dyn_leaves(leaves, max_leaves, k, "", s, 0); break;
    case 1: mlp_leaves(leaves, max_leaves, k, "", s->cost_layers, s->cost_dims, 0); break;
    case 2:
      if (s->lstm_features <= 0) return fail(GMPC_EINVAL, "this shape has no critic");
      critic_leaves(leaves, max_leaves, k, "", s, 0);
      break;
    case 3: {
      // the training vector of the host package: [mpc_weights | cost | dynamics | critic], so that the
      // trainable ranges of the reference's optimisers (gan/runner.py:51-63) are contiguous
      add_leaf(leaves, max_leaves, k, "mpc_weights", 0, 1, 3, 3);
      long off = mlp_leaves(leaves, max_leaves, k, "cost_params/", s->cost_layers, s->cost_dims, 3);
      off = dyn_leaves(leaves, max_leaves, k, "dynamics_params/", s, off);
      if (s->lstm_features > 0) critic_leaves(leaves, max_leaves, k, "critic_params/", s, off);
      break;
    }
    default: return fail(GMPC_EINVAL, "which must be 0 (dyn), 1 (cost), 2 (critic) or 3 (training vector)");
  }
  return k;
}

extern "C" int gmpc_polyak(gmpc_ctx* c, long count, const float* prev, const float* cur, double factor,
                           float* out, void* stream) {
  if (!c || !prev || !cur || !out || count < 1) return fail(GMPC_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  gmpc_launch_polyak(count, prev, cur, factor, out, static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

// bilevel ----------------------------------------------------------------------------------------
extern "C" int gmpc_bilevel_grad(gmpc_ctx* c, int B, int loss_kind, const float* desired,
                                 const float* critic, float sign, float* loss, float* grad_sum,
                                 void* stream) {
  TRY(check_call(c, B));
  if (c->solB != B) return fail(GMPC_EINVAL, "gmpc_ilqr_solve with B=%d must precede this call", B);
  if (!loss || !grad_sum) return fail(GMPC_EINVAL, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const gmpc_shape& sh = c->sh;
  const int n = sh.n, m = sh.m, T = sh.T;
  TRY(upper_loss(c, B, loss_kind, desired, critic, loss, true, s));
  // a8: Bvec; a9+solve: structured Hessian solve; a11: cost_vjp
  if (c->big) {
    // step-major: the loss adjoint (Bvec) and the Riccati sweep of the Hessian solve share one
    // backward pass over re-linearised steps, the tangent roll is a second, forward pass
    if (gmpc_big_backward(c->bw, B, c->dyn, c->lp, c->masks, c->Xs, c->Us, c->goals, c->mpc_w, c->QT,
                          c->qT, nullptr, c->Ks, c->ks, nullptr, nullptr, c->lx, c->Bvec, s,
                          c->dynl ? &c->dl : nullptr, c->dynl ? c->adjs : nullptr) != 0 ||
        gmpc_big_forward_tangent(c->bw, B, c->dyn, c->lp, c->masks, c->Ks, c->ks, c->Hout, c->dX, s,
                                 c->dynl ? &c->dl : nullptr, c->Xs, c->Us) != 0)
      return fail(GMPC_EINVAL, "large-state bilevel: Jacobian kernel does not cover this shape");
  } else {
    RiccatiArgs r;
    memset(&r, 0, sizeof(r));
    r.B = B; r.n = n; r.ng = c->nx; r.m = m; r.T = T; r.mode = 1;
    r.X = c->Xs; r.U = c->Us; r.goal = c->goals; r.mpc_w = c->mpc_w; r.AB = c->AB; r.QT = c->QT;
    r.qT = c->qT; r.K = c->Ks; r.k = c->ks; r.Bvec = c->Bvec; r.Hout = c->Hout; r.dX = c->dX;
    if (!c->dynl && gmpc_riccati_w2h_shape(r)) {
      // two waves per trajectory, products on the matrix pipe, the loss adjoint (a8) in the same sweep
      ProfScope ps(c, PROF_RICCATI, s);      // (bench.py: secondary.bilevel.kernel_ms)
      gmpc_launch_riccati_w2h(r, c->lx, c->Bvec, s);
    } else {
      gmpc_launch_bvec(B, T, n, m, c->AB, c->lx, c->Bvec, s);
      if (c->dynl) {
        // smooth dynamics: the dense Hessian the reference solves with carries lam_{t+1} . d^2 f (oracle
        // second_order_lqr); lam = the adjoints of the solve's last backward pass
        gmpc_launch_dynl_curv(B, T, T, 0, c->dl, c->Xs, c->Us, c->adjs, nullptr, c->phi, s);
        r.Phi = c->phi;
      }
      ProfScope ps(c, PROF_RICCATI, s);      // (bench.py: secondary.bilevel.kernel_ms)
      gmpc_launch_riccati(r, s);
    }
  }
  gmpc_launch_costvjp(B, T, n, m, c->cost, c->mpc_w, sign, c->Xs, c->Us, c->goals, c->nx, c->Hout, c->dX,
                      c->gmpc, c->cact, c->cdel, c->cstride, s);
  // sums over the batch: mpc_w (3 columns of gmpc) and the cost layers
  gmpc_launch_wgrad(B, 1, 3, c->gmpc, 0, c->gmpc, 3, c->scratch + 512, grad_sum, B, c->wpart, 256, s, c->wpart_floats, false);
  float* g = grad_sum + 3;
  int aoff = 0, doff = 0;
  for (int l = 0; l < sh.cost_layers; ++l) {
    const int M = sh.cost_dims[l], N = sh.cost_dims[l + 1];
    gmpc_launch_wgrad(2 * B, M, N, c->cact + aoff, c->cstride, c->cdel + doff, c->cstride, g,
                      g + (long)M * N, B, c->wpart, 256, s, c->wpart_floats, true);
    g += (long)M * N + N;
    aoff += M;
    doff += N;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gmpc_adam_clip_step(gmpc_ctx* c, long count, float* params, const float* grad,
                                   float* m, float* v, float grad_scale, int step, double lr,
                                   double max_norm, double b1, double b2, double eps, void* stream) {
  if (!c || !params || !grad || !m || !v) return fail(GMPC_EINVAL, "null argument");
  if (count < 1 || step < 1) return fail(GMPC_EINVAL, "count and step must be positive");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    ProfScope ps(c, PROF_ADAM, s);
    gmpc_launch_adam(count, params, grad, m, v, grad_scale, step, lr, max_norm, b1, b2, eps,
                     c->scratch, s);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Batched TN GEMM used by the large-state Riccati path, exported for its unit test:
// C[b] = alpha * X[b]^T Y[b] + beta * C[b] with X[b] K x M, Y[b] K x N, C[b] M x N, all row-major
// and densely packed per batch element.  Y must be followed by >= 8 readable rows.
extern "C" int gmpc_bgemm_tn(gmpc_ctx* c, int batch, int M, int N, int K, const float* X, const float* Y,
                             float* C, float alpha, float beta, void* stream) {
  if (!c || !X || !Y || !C || batch < 1 || M < 1 || N < 1 || K < 1) return fail(GMPC_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  BgemmArgs a;
  a.batch = batch; a.M = M; a.N = N; a.K = K;
  a.X = X; a.sx = (long)K * M; a.ldx = M;
  a.Y = Y; a.sy = (long)K * N; a.ldy = N;
  a.C = C; a.sc = (long)M * N; a.ldc = N;
  a.alpha = alpha; a.beta = beta; a.active = nullptr;
  gmpc_launch_bgemm_tn(a, static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" long gmpc_linesearch_candidates(gmpc_ctx* c) {
  if (!c) return -1;
  int v = 0;
  if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(&v, c->lsw.counts + GMPC_LS_ROUNDS_MAX, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return v;
}

extern "C" int gmpc_linesearch_stats(gmpc_ctx* c, long* out, int n) {
  if (!c || !out) return fail(GMPC_EINVAL, "ctx / out is null");
  int v[GMPC_LS_STATS];
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());      // (streams created non-blocking are not ordered against a null-stream copy)
  HIP_TRY(hipMemcpy(v, c->lsw.counts + GMPC_LS_ROUNDS_MAX + 1, sizeof(v), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) out[i] = i < GMPC_LS_STATS ? v[i] : 0;
  return 0;
}

extern "C" int gmpc_set_linearize_event(gmpc_ctx* c, void* ev) {
  if (!c) return fail(GMPC_EINVAL, "ctx is null");
  c->lin_event = static_cast<hipEvent_t>(ev);
  return 0;
}

extern "C" int gmpc_profile_enable(gmpc_ctx* c, int on) {
  if (!c) return fail(GMPC_EINVAL, "ctx is null");
  c->prof = on != 0;
  return 0;
}

extern "C" const char* gmpc_profile_kernel_name(gmpc_ctx* c, int slot) {
  if (!c || slot != PROF_LINEARIZE) return "";
  return c->lin_kernel;
}

extern "C" int gmpc_profile_read(gmpc_ctx* c, int slot, double* total_ms, int* count) {
  if (!c || slot < 0 || slot >= GMPC_PROF_SLOTS || !total_ms || !count)
    return fail(GMPC_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  (void)hipGetLastError();   // clean slate (see check_call)
  double tot = 0.0;
  int n = 0;
  for (auto& pr : c->prof_ev[slot]) {
    HIP_TRY(hipEventSynchronize(pr.second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
    ++n;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->prof_ev[slot].clear();
  *total_ms = tot;
  *count = n;
  return 0;
}

// accessors used by the bilevel parity tests and the Python mirror (device pointers, valid until the
// next solve): 0 X, 1 U, 2 H (A^-1 B), 3 dX, 4 Bvec, 5 AB, 6 K, 7 k
extern "C" long gmpc_debug_buffer_count(gmpc_ctx* c, int which) {
  if (!c) return 0;
  const gmpc_shape& s = c->sh;
  const long B = c->maxB, n = s.n, m = s.m, T = s.T;
  switch (which) {
    case 0: case 3: return B * (T + 1) * n;
    case 1: case 2: case 4: case 7: return B * T * m;
    case 5: return c->big ? B * n * (n + m) : B * T * n * (n + m);
    case 6: return B * T * m * n;
    case 8: case 9: case 10: return B;
    case 11: return B * (T + 1) * n;
    case 12: return (long)GMPC_LS_ITEMS * B * (T + 1) * n;
    case 13: return (long)GMPC_LS_ITEMS * B * T * m;
    case 14: return 256;
    default: return 0;
  }
}

extern "C" const float* gmpc_debug_buffer(gmpc_ctx* c, int which) {
  if (!c) return nullptr;
  switch (which) {
    case 0: return c->Xs; case 1: return c->Us; case 2: return c->Hout; case 3: return c->dX;
    case 4: return c->Bvec; case 5: return c->AB; case 6: return c->Ks; case 7: return c->ks;
    case 14: return c->scratch + 768; case 12: return c->Xc; case 13: return c->Uc;
    case 8: return c->alpha; case 9: return c->obj_step; case 10: return c->U_step;
    case 11: return c->lx;
    default: return nullptr;
  }
}
