// Sequential-in-time trajectory kernels: rollout + cost, and the DDP line search.  This file holds the
// general (any network shape) VALU form, the line-search bookkeeping kernels and the launchers; the
// reference's default dynamics network (3 x 200) takes the register-weight MFMA form of
// gmpc_traj_rw.hip (GMPC_TRAJ=valu forces the general form).
//
// One 256-thread workgroup owns GMPC_TB = 4 trajectories for the whole horizon.  The state and
// control of the current step live in LDS as float4 (one component per trajectory), every layer is
// "one output neuron per thread": the weight row is read coalesced from L2 once and reused for
// the four trajectories from registers.  relu sign bits leave the kernel as ballot bitmasks so the
// backward Jacobian chain never recomputes the forward pass.
//
// Reference arithmetic: dynamics/nn.py:27-34 (residual relu MLP), cost/cost_model.py:20-42,
// cost/nn.py:23-29, trajax rollout / evaluate / ddp_rollout / line_search_ddp as called from
// policy/optimizers.py:19,26-29,55.
#include "gmpc_traj_layers.h"

#ifndef GMPC_TRAJ_MINW
#define GMPC_TRAJ_MINW 4
#endif
#define GMPC_TRAJ_THREADS 512   // k_traj: 8 waves, the upper 4 take the second half of every K range

template <bool LS>
__global__ __launch_bounds__(GMPC_TRAJ_THREADS, GMPC_TRAJ_MINW) void k_traj(TrajArgs a) {
  // dynamic LDS: actA | actB (aw float4 each: max(n+m, widest layer)) | part (pw) | ksp (256) | xcur (n)
  extern __shared__ __attribute__((aligned(16))) char smem_traj[];
  float4* const actA = reinterpret_cast<float4*>(smem_traj);
  float4* const actB = actA + a.aw;
  float4* const part = actB + a.aw;
  float4* const ksp = part + a.pw;
  float4* const xcur = ksp + GMPC_THREADS;
  // the two small weight matrices live in LDS for the whole horizon when they fit (launcher decides)
  float* const w0_s = reinterpret_cast<float*>(xcur + a.n);
  float* const wl_s = w0_s + a.sw0;
  __shared__ float s_alpha[GMPC_TB], s_oo[GMPC_TB];
  __shared__ int s_bi[GMPC_TB], s_in[GMPC_TB], s_live[GMPC_TB];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = a.n, m = a.m, T = a.T;
  // slot c of this block: plain rollout -> trajectory b0 + c; line search -> item b0 + c of the
  // round's work list, i.e. one (trajectory, step size) candidate
  const int b0 = blockIdx.x * GMPC_TB;
  if (LS) {
    const int cnt = *a.nitems;
    if (b0 >= cnt) return;
    if (tid < GMPC_TB) {
      const int it = min(b0 + tid, cnt - 1);
      s_bi[tid] = a.item_b[it];
      s_in[tid] = (b0 + tid) < cnt;
      float al = a.alpha_0;
      for (int k = a.item_k[it]; k > 0; --k) al *= 0.5f;
      s_alpha[tid] = al;
      // the objective to beat: stage and terminal costs are non-negative, so a candidate whose
      // running sum has reached it can no longer be accepted (NaN compares false: dead as well)
      float oo = a.obj[s_bi[tid]];
      if (isnan(oo)) oo = INFINITY;
      s_oo[tid] = oo;
      s_live[tid] = s_in[tid];
    }
  } else if (tid < GMPC_TB) {
    s_bi[tid] = min(b0 + tid, a.B - 1);
    s_in[tid] = (b0 + tid) < a.B;
  }
  __syncthreads();
  // Component c of an LDS float4 is always addressed as a float ([k*4 + c]) when c is a run-time
  // value: one address computation instead of the branch tree hipcc builds for `c == 0 ? v.x : ...` on an
  // LDS reference.  (Round 1 saw that tree hand lanes with c == 3 the .z address inside this kernel; the
  // pattern in isolation compiles and runs correctly, tests/repro/README.md.)
  float* const xf = reinterpret_cast<float*>(xcur);
  float* const aAf = reinterpret_cast<float*>(actA);
  const float* const pf = reinterpret_cast<const float*>(part);
  // trajectory c of this block (reads of the tail block are clamped); no private arrays: a
  // dynamically indexed register array would live in scratch
  auto BI = [&](int c) -> int { return s_bi[c]; };
  auto INB = [&](int c) -> bool { return s_in[c] != 0; };
  // candidate (item) index of slot c: where the line search writes X / U / masks / objective
  auto CI = [&](int c) -> size_t { return (size_t)(b0 + c); };
  const int Lh = a.dyn.L - 1;
  const size_t mstride = (size_t)T * Lh * GMPC_MW;   // mask words per trajectory
  for (int e = tid; e < a.sw0; e += blockDim.x) w0_s[e] = a.dyn.W[0][e];
  for (int e = tid; e < a.swl; e += blockDim.x) wl_s[e] = a.dyn.W[Lh][e];
  const float* const W0 = a.sw0 ? w0_s : a.dyn.W[0];
  const float* const WL = a.swl ? wl_s : a.dyn.W[Lh];
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  {
    // bit c set: slot c writes its outputs
    unsigned wbits = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) wbits |= (INB(c) ? 1u : 0u) << c;
    // ---- initial state
    for (int i = tid; i < n; i += blockDim.x) {
      const float* xs = LS ? a.X : a.x0;
      const size_t st = LS ? (size_t)(T + 1) * n : (size_t)n;
      float4 v = make_float4(xs[BI(0) * st + i], xs[BI(1) * st + i], xs[BI(2) * st + i],
                             xs[BI(3) * st + i]);
      xcur[i] = v;
      if (!LS) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (INB(c)) a.X[(size_t)BI(c) * (T + 1) * n + i] = f4get(v, c);
      }
    }
    float objacc = 0.f;  // lane 0 of wave c accumulates trajectory c
    // goal of the NEXT step, one element per lane of wave c (n <= 64): its load overlaps a whole step
    float gnext = 0.f;
    if (n <= 64 && wave < GMPC_TB && lane < n) gnext = a.goal[(size_t)BI(wave) * (T + 1) * n + lane];
    __syncthreads();
#ifdef GMPC_TRAJ_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
#define TS_(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define TS_(i)
#endif

    bool aborted = false;
    for (int t = 0; t < T; ++t) {
      // line search: stop as soon as none of the four candidates can still be accepted (flags of the
      // previous step's cost evaluation; the barriers of that step ordered them)
      if (LS && t > 0 && (s_live[0] | s_live[1] | s_live[2] | s_live[3]) == 0) { aborted = true; break; }
      // ---- controls and layer-0 input
      for (int i = tid; i < n; i += blockDim.x) actA[i] = xcur[i];
      if (LS) {
        // u = U + alpha k + K (x - X_nominal): 16 lanes share one (slot, control) inner product, so
        // the n gain / state loads of a control are issued together instead of one after another
        const int l16 = tid & 15;
        for (int p = tid >> 4; p < GMPC_TB * m; p += GMPC_TRAJ_THREADS >> 4) {
          const int c = p / m, j = p - c * m;
          const int bc = BI(c);
          const size_t ub = ((size_t)bc * T + t) * m + j;
          const float* Kr = a.Kg + ub * n;
          const float* Xo = a.X + ((size_t)bc * (T + 1) + t) * n;
          float du = 0.f;
          for (int i = l16; i < n; i += 16) du = fmaf(Kr[i], xf[i * 4 + c] - Xo[i], du);
          du += __shfl_xor(du, 8);
          du += __shfl_xor(du, 4);
          du += __shfl_xor(du, 2);
          du += __shfl_xor(du, 1);
          if (l16 == 0) {
            const float u = a.Uio[ub] + fmaf(s_alpha[c], a.kg[ub], du);
            if ((wbits >> c) & 1u) a.Uc[(CI(c) * T + t) * m + j] = u;
            aAf[(n + j) * 4 + c] = u;
          }
        }
      } else if (tid < GMPC_TB * m) {
        const int c = tid / m, j = tid % m;
        aAf[(n + j) * 4 + c] = a.U[((size_t)BI(c) * T + t) * m + j];
      }
      __syncthreads();
      TS_(0)
      // ---- stage cost of (x_t, u_t): wave c (< 4) handles trajectory c
      if (wave < GMPC_TB) {
        const int c = wave;
        float dd = 0.f, uu = 0.f;
        const int bc = BI(c);
        const float* g = a.goal + ((size_t)bc * (T + 1) + t) * n;
        if (n <= 64) {
          const float gi = gnext;
          if (lane < n) {
            gnext = g[n + lane];           // row t + 1 (exists: the goal has T + 1 rows)
            const float d = aAf[lane * 4 + c] - gi;
            dd = d * d;
          }
        } else {
          for (int i = lane; i < n; i += 64) {
            const float d = aAf[i * 4 + c] - g[i];
            dd = fmaf(d, d, dd);
          }
        }
        for (int j = lane; j < m; j += 64) {
          const float u = aAf[(n + j) * 4 + c];
          uu = fmaf(u, u, uu);
        }
        dd = wave_sum(dd);
        uu = wave_sum(uu);
        const float al = GMPC_ALPHA;
        const float cst = w0 * (sqrtf(uu + al * al) - al) + w1 * (sqrtf(dd + al * al) - al);
        objacc += cst;
        if (LS && lane == 0) s_live[c] = (INB(c) && objacc < s_oo[c]) ? 1 : 0;
        if (!LS && lane == 0 && INB(c) && a.costs) a.costs[(size_t)bc * (T + 1) + t] = cst;
      }
      TS_(1)
      // ---- hidden layers
      float4* in = actA;
      float4* out = actB;
      for (int l = 0; l < Lh; ++l) {
        // the tail block's clamped trajectories never write (wbits), so b0-relative addressing is safe
        uint32_t* mbase = (LS ? a.maskc : a.masks) + (size_t)b0 * mstride + ((size_t)t * Lh + l) * GMPC_MW;
        hidden_layer(l == 0 ? W0 : a.dyn.W[l], a.dyn.b[l], a.dyn.dims[l], a.dyn.dims[l + 1], in, out, mbase,
                     mstride, wbits, ksp);
        __syncthreads();
        TS_(2 + l)
        float4* tmp = in; in = out; out = tmp;
      }
      // ---- output layer + residual
      if (n <= 32) {
        out_layer32(WL, a.dyn.dims[Lh], n, in, part);
      } else if (n <= (int)blockDim.x) {
        dense_small<1>(WL, a.dyn.dims[Lh], n, in, part);
      } else {
        // wide state (n > 512): one output per thread, chunk after chunk
        for (int jb = 0; jb < n; jb += blockDim.x) {
          float4 acc[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
          dense_rows<1>(WL, a.dyn.dims[Lh], n, jb + tid, in, acc);
          if (jb + tid < n) part[jb + tid] = acc[0];
        }
        __syncthreads();
      }
      for (int i = tid; i < n; i += blockDim.x) {
        const float bj = a.dyn.b[Lh][i];
        float4 v = part[i];
        const float4 xo = xcur[i];
        v.x = (v.x + bj) + xo.x; v.y = (v.y + bj) + xo.y;
        v.z = (v.z + bj) + xo.z; v.w = (v.w + bj) + xo.w;
        xcur[i] = v;
        float* Xo = LS ? a.Xc : a.X;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if ((wbits >> c) & 1u)
            Xo[((LS ? CI(c) : (size_t)BI(c)) * (T + 1) + t + 1) * n + i] = f4get(v, c);
      }
      __syncthreads();
      TS_(6)
    }
#ifdef GMPC_TRAJ_STAMPS
    if (blockIdx.x == 0 && (tid == 0 || tid == 256))
      printf("tid %d: staging %llu cost %llu L0 %llu L1 %llu L2 %llu out %llu (cycles per step)\n", tid,
             st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T, st_[6] / T);
#endif
    if (LS && aborted) {
      if (tid < GMPC_TB && INB(tid)) a.objc[CI(tid)] = INFINITY;     // rejected without a full rollout
      return;
    }
    // ---- terminal cost w2 * |cost_mlp(x_T)|^2
    {
      float4* in = xcur;
      float4* out = actA;
      const int Lc = a.cost.L - 1;
      for (int l = 0; l < Lc; ++l) {
        hidden_layer(a.cost.W[l], a.cost.b[l], a.cost.dims[l], a.cost.dims[l + 1], in, out, nullptr, 0,
                     0u, ksp);
        __syncthreads();
        in = out;
        out = (out == actA) ? actB : actA;
      }
      const int fo = a.cost.dims[Lc + 1];
      dense_small<1>(a.cost.W[Lc], a.cost.dims[Lc], fo, in, part);
      const int c = wave & (GMPC_TB - 1);
      float yy = 0.f;
      for (int r = lane; r < fo; r += 64) {
        const float y = pf[r * 4 + c] + a.cost.b[Lc][r];
        yy = fmaf(y, y, yy);
      }
      yy = wave_sum(yy);
      const float cst = w2 * yy;
      objacc += cst;
      if (lane == 0 && wave < GMPC_TB) {
        if (!LS) {
          if (INB(c)) {
            if (a.costs) a.costs[(size_t)BI(c) * (T + 1) + T] = cst;
            a.obj[BI(c)] = objacc;
          }
        } else if (INB(c)) {
          a.objc[CI(c)] = objacc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round-based backtracking line search (trajax line_search_ddp: alpha = alpha_0, alpha_0/2, ... while
// alpha > alpha_min, the first candidate whose objective decreases is taken).  The halvings of one
// trajectory are independent rollouts, so a round evaluates several of them speculatively
// (k_traj<true> over a work list of (trajectory, halving count) candidates) and k_ls_decide picks,
// per trajectory, the LARGEST accepted step of the round -- the candidate the sequential loop would
// have stopped at -- commits it, or queues the next 4 halvings (8 from the third round on).  The first round of a trajectory
// covers the halvings up to the one its previous line search accepted (1 candidate for a
// well-conditioned problem that takes full steps, up to 8 for one that backtracks deeply), so the
// rollouts stay close to the sequential loop's count while the launches drop from up to 15
// dependent rollouts to 1-3 rounds.  Nothing is read back by the host.
// ------------------------------------------------------------------------------------------------
#define GMPC_LS_NEXT 4   // candidates queued per trajectory after a round without an accepted step

__global__ void k_ls_init(int B, const int* active, float alpha_0, float alpha_min, int k_max, int first_min, int* iters,
                          int* run, int* cnt, int* kfirst, const int* prevk, float* alpha, float* U_step,
                          float* obj_step) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  cnt[b] = 0;
  if (active != nullptr && active[b] == 0) { run[b] = 0; return; }
  iters[b] += 1;
  if (alpha_0 > alpha_min) {
    int R = prevk[b] + 1;
    R = R < first_min ? first_min : R;      // (first_min >= 1: see gmpc_launch_linesearch)
    R = R > GMPC_LS_ITEMS ? GMPC_LS_ITEMS : R;
    R = R > k_max ? k_max : R;
    run[b] = 1;
    cnt[b] = R;
    kfirst[b] = 0;
  } else {
    run[b] = 0;
    alpha[b] = alpha_0;
    U_step[b] = 0.f;
    obj_step[b] = 0.f;
  }
}

// Work list of a round, ordered by candidate number first and trajectory second: the four slots of a
// k_traj<true> workgroup then hold the SAME halving count of four trajectories.  Large steps are
// rejected early in the horizon (their running cost passes the objective to beat within a few
// steps), and a workgroup whose four candidates are all dead stops -- which only happens when
// candidates of similar fate sit together.  One workgroup; cnt[b] candidates for trajectory b.
__global__ __launch_bounds__(1024) void k_ls_place(int B, const int* cnt, const int* kfirst, int* item_b,
                                                   int* item_k, int* slot, int* count, int* total,
                                                   int* round_total) {
  __shared__ int s_n[GMPC_LS_ITEMS], s_base[GMPC_LS_ITEMS], s_fill[GMPC_LS_ITEMS];
  const int tid = threadIdx.x;
  if (tid < GMPC_LS_ITEMS) { s_n[tid] = 0; s_fill[tid] = 0; }
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) {
    const int c = cnt[b];
    for (int j = 0; j < c; ++j) atomicAdd(&s_n[j], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int j = 0; j < GMPC_LS_ITEMS; ++j) { s_base[j] = acc; acc += s_n[j]; }
    *count = acc;
    *total += acc;          // candidate rollouts since the solve began (one workgroup: no race)
    *round_total += acc;
  }
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) {
    const int c = cnt[b], k0 = kfirst[b];
    for (int j = 0; j < c; ++j) {
      const int pos = s_base[j] + atomicAdd(&s_fill[j], 1);
      item_b[pos] = b;
      item_k[pos] = k0 + j;
      slot[b * GMPC_LS_ITEMS + j] = pos;
    }
  }
}

struct LsDecideArgs {
  int n, m, T, Lh, k_max;
  int next;              // candidates queued for the next round when this one accepts nothing
  float alpha_0;
  const int* slot; int* cnt; int* kfirst; int* prevk; int* run;
  const float* objc; const float* Xc; const float* Uc; const uint32_t* maskc;
  float* X; float* U; uint32_t* masks;
  float* obj; float* obj_step; float* U_step; float* alpha;
  int* stats;            // LsWork::counts + GMPC_LS_ROUNDS_MAX + 1
};

__global__ __launch_bounds__(GMPC_THREADS) void k_ls_decide(LsDecideArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (a.run[b] == 0) return;
  __shared__ int s_acc;
  __shared__ float s_us[GMPC_THREADS / 64];
  const int* sl = a.slot + (size_t)b * GMPC_LS_ITEMS;
  __shared__ int s_item;
  if (tid == 0) {
    float oo = a.obj[b];
    if (isnan(oo)) oo = INFINITY;
    const int R = a.cnt[b], k0 = a.kfirst[b];
    int acc = -1;
    float on_acc = 0.f;
    for (int j = 0; j < R; ++j) {
      float on = a.objc[sl[j]];
      if (isnan(on)) on = oo;
      if (on < oo) { acc = j; on_acc = on; break; }
    }
    s_item = acc >= 0 ? sl[acc] : 0;
    auto halved = [&](int k) { float al = a.alpha_0; for (; k > 0; --k) al *= 0.5f; return al; };
    if (acc >= 0) {
      a.obj[b] = on_acc;
      a.obj_step[b] = fabsf(on_acc - oo);
      a.alpha[b] = halved(k0 + acc + 1);
      a.prevk[b] = k0 + acc;
      a.run[b] = 0;
      a.cnt[b] = 0;
      atomicAdd(a.stats + min(k0 + acc, 15), 1);
      atomicMax(a.stats + 17, k0 + acc);       // deepest halving accepted since the solve began
    } else if (k0 + R >= a.k_max) {      // every step size down to alpha_min failed
      a.alpha[b] = halved(a.k_max);
      a.U_step[b] = 0.f;
      a.obj_step[b] = 0.f;
      a.prevk[b] = a.k_max - 1;
      a.run[b] = 0;
      a.cnt[b] = 0;
      atomicAdd(a.stats + 16, 1);
    } else {                               // queue the next GMPC_LS_NEXT halvings (k_ls_place)
      // (round 4, measured and not kept: a second round that reaches the deepest halving any trajectory of the batch
      // has accepted so far -- the third round it was meant to remove only exists in a solve's first iteration, and
      // the longer second round cost 0.2 ms per iteration)
      const int left = a.k_max - (k0 + R);
      a.cnt[b] = left < a.next ? left : a.next;
      a.kfirst[b] = k0 + R;
    }
    s_acc = acc;
  }
  __syncthreads();
  const int acc = s_acc;
  if (acc < 0) return;
  // commit the accepted candidate as the new iterate
  const size_t it = (size_t)s_item;
  const int n = a.n, m = a.m, T = a.T;
  float* Xd = a.X + (size_t)b * (T + 1) * n;
  const float* Xs = a.Xc + it * (T + 1) * n;
  for (int e = n + tid; e < (T + 1) * n; e += blockDim.x) Xd[e] = Xs[e];
  float us = 0.f;
  float* Ud = a.U + (size_t)b * T * m;
  const float* Us = a.Uc + it * T * m;
  for (int e = tid; e < T * m; e += blockDim.x) {
    const float un = Us[e], d = un - Ud[e];
    us = fmaf(d, d, us);
    Ud[e] = un;
  }
  const size_t mw = (size_t)T * a.Lh * GMPC_MW;
  uint32_t* Md = a.masks + (size_t)b * mw;
  const uint32_t* Ms = a.maskc + it * mw;
  for (size_t e = tid; e < mw; e += blockDim.x) Md[e] = Ms[e];
  us = wave_sum(us);
  if ((tid & 63) == 0) s_us[tid >> 6] = us;
  __syncthreads();
  if (tid == 0) a.U_step[b] = sqrtf((s_us[0] + s_us[1]) + (s_us[2] + s_us[3]));
}

// Forward pass at given (x, u) pairs, masks only: used when gmpc_lqr_backward is handed a
// trajectory that did not come from this context's rollout.  4 samples per workgroup.
__global__ __launch_bounds__(GMPC_THREADS) void k_masks(int NS, int n, int m, int T, MlpDesc dyn,
                                                        const float* X, const float* U,
                                                        uint32_t* masks, int aw) {
  extern __shared__ __attribute__((aligned(16))) char smem_masks[];
  float4* const actA = reinterpret_cast<float4*>(smem_masks);
  float4* const actB = actA + aw;
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * 4;
  unsigned wbits = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) wbits |= ((s0 + c < NS) ? 1u : 0u) << c;
  for (int i = tid; i < n + m; i += blockDim.x) {
    float4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int si = (s0 + c < NS) ? s0 + c : NS - 1;
      const int b = si / T, t = si % T;
      const float x = i < n ? X[((size_t)b * (T + 1) + t) * n + i]
                            : U[((size_t)b * T + t) * m + (i - n)];
      f4set(v, c, x);
    }
    actA[i] = v;
  }
  __syncthreads();
  const int Lh = dyn.L - 1;
  float4* in = actA;
  float4* out = actB;
  for (int l = 0; l < Lh; ++l) {
    hidden_layer(dyn.W[l], dyn.b[l], dyn.dims[l], dyn.dims[l + 1], in, out,
                 masks + ((size_t)s0 * Lh + l) * GMPC_MW, (size_t)Lh * GMPC_MW, wbits);
    __syncthreads();
    float4* tmp = in; in = out; out = tmp;
  }
}

// Host-side launchers ---------------------------------------------------------------------------
static int traj_aw(int n, int m, const MlpDesc& d1, const MlpDesc* d2) {
  int w = n + m > GMPC_THREADS ? n + m : GMPC_THREADS;
  for (int l = 0; l <= d1.L; ++l) w = d1.dims[l] > w ? d1.dims[l] : w;
  if (d2) for (int l = 0; l <= d2->L; ++l) w = d2->dims[l] > w ? d2->dims[l] : w;
  return (w + 3) & ~3;
}
static size_t traj_lds(TrajArgs& a) {
  a.aw = traj_aw(a.n, a.m, a.dyn, &a.cost);
  a.pw = a.n > GMPC_TRAJ_THREADS ? a.n : GMPC_TRAJ_THREADS;
  size_t bytes = ((size_t)2 * a.aw + a.pw + GMPC_THREADS + a.n) * sizeof(float4);
  // W_0 and W_L in LDS while the workgroup stays under 64 KB (two workgroups per CU in the line search)
  const int Lh = a.dyn.L - 1;
  const size_t w0 = (size_t)a.dyn.dims[0] * a.dyn.dims[1], wl = (size_t)a.dyn.dims[Lh] * a.n;
  a.sw0 = a.swl = 0;
  if (bytes + w0 * sizeof(float) <= 64 * 1024) { a.sw0 = (int)w0; bytes += w0 * sizeof(float); }
  if (bytes + wl * sizeof(float) <= 64 * 1024) { a.swl = (int)wl; bytes += wl * sizeof(float); }
  return bytes;
}
// dynamic LDS above the default 64 KB needs the attribute; the kernels also hold a few hundred bytes
// of static LDS, so the full 160 KB cannot be requested
template <typename KernelT>
static void traj_attr(KernelT k) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 1024);
  (void)hipGetLastError();
}

bool gmpc_traj_rw_shape(const TrajArgs& a);
size_t gmpc_traj_rw_lds(TrajArgs& a);
void gmpc_launch_traj_rw(const TrajArgs& a, bool ls, int grid, size_t lds, hipStream_t s);
bool gmpc_ls16_shape(const TrajArgs& a);
int gmpc_ls16_split();
void gmpc_launch_ls16(const TrajArgs& a, long max_items, hipStream_t s);
bool gmpc_ls32_shape(const TrajArgs& a);
int gmpc_ls32_split();
void gmpc_launch_ls32(const TrajArgs& a, long max_items, int min_items, hipStream_t s);

void gmpc_launch_rollout(const TrajArgs& a0, hipStream_t s) {
  TrajArgs a = a0;
  if (gmpc_traj_rw_shape(a)) {
    a.aw = traj_aw(a.n, a.m, a.dyn, &a.cost);
    const size_t rlds = gmpc_traj_rw_lds(a);
    gmpc_launch_traj_rw(a, false, (a.B + GMPC_TB - 1) / GMPC_TB, rlds, s);
    return;
  }
  const size_t lds = traj_lds(a);
  static bool attr = false;
  if (!attr) { traj_attr(&k_traj<false>); attr = true; }
  const int grid = (a.B + GMPC_TB - 1) / GMPC_TB;
  hipLaunchKernelGGL(k_traj<false>, dim3(grid), dim3(GMPC_TRAJ_THREADS), lds, s, a);
}
// `eval` (optional): another evaluator of the candidates of a round -- the LSTM dynamics variant
// (gmpc_dynl.hip) -- behind the same work list and the same decide / commit kernels.
typedef void (*gmpc_ls_eval_fn)(void* user, const TrajArgs&, int max_items, hipStream_t);
// after the first round's decision: up to split.cap of the trajectories whose line search is over go, in index order,
// to the list of the early Jacobian chain; every other active trajectory goes to the list of the late chain.  The early
// chain is worth its CUs only while the second round is one pass of k_ls16 that leaves them idle: with fewer than
// cand_min candidates to come (a round of k_traj_rw, half as long) or more than `wg_max` workgroups of 16, the early
// list stays empty.
__global__ __launch_bounds__(1024) void k_ls_split(int B, const int* active, const int* run, int cap, int next,
                                                   int cand_min, int wg_max, int* tlist, int* tcount, int* llist,
                                                   int* lcount) {
  __shared__ int s_wf[16], s_wo[16], s_bf, s_bo, s_run;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { s_bf = 0; s_bo = 0; s_run = 0; }
  __syncthreads();
  int nrun = 0;
  for (int b = tid; b < B; b += 1024) nrun += ((active == nullptr || active[b] != 0) && run[b] != 0) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) nrun += __shfl_xor(nrun, o);
  if (lane == 0) atomicAdd(&s_run, nrun);
  __syncthreads();
  // (a second round below cand_min candidates is not k_ls16's: it is over long before the early chain would be)
  const long cand2 = (long)s_run * next;
  const int lim = (cand2 >= cand_min && (cand2 + 15) / 16 <= wg_max) ? cap : 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    const bool on = b < B && (active == nullptr || active[b] != 0);
    const bool fin = on && run[b] == 0;
    const unsigned long long balf = __ballot(fin);
    if (lane == 0) s_wf[wv] = __popcll(balf);
    __syncthreads();
    int pos = s_bf + __popcll(balf & below);
    for (int w = 0; w < wv; ++w) pos += s_wf[w];
    const bool pick = fin && pos < lim;
    if (pick) tlist[pos] = b;
    const bool other = on && !pick;
    const unsigned long long balo = __ballot(other);
    if (lane == 0) s_wo[wv] = __popcll(balo);
    __syncthreads();
    int po = s_bo + __popcll(balo & below);
    for (int w = 0; w < wv; ++w) po += s_wo[w];
    if (other) llist[po] = b;
    __syncthreads();
    if (tid == 0) {
      int tf = 0, to = 0;
      for (int w = 0; w < 16; ++w) { tf += s_wf[w]; to += s_wo[w]; }
      s_bf += tf;
      s_bo += to;
    }
    __syncthreads();
  }
  if (tid == 0) { *tcount = min(s_bf, lim); *lcount = s_bo; }
}
// `split` (optional): the caller wants to start the backward pass of the trajectories that are done after the FIRST
// round while the later rounds -- shorter work lists that leave part of the chip idle -- are still running: their flags
// are compacted into split->tlist, every other active trajectory into split->llist, and split->ev is recorded behind
// that, in front of the second round.
int gmpc_launch_linesearch(const TrajArgs& a0, const LsWork& w, hipStream_t s, gmpc_ls_eval_fn eval,
                           void* user, const LsSplit* split) {
  TrajArgs a = a0;
  const bool rw = eval == nullptr && gmpc_traj_rw_shape(a);
  if (rw) a.aw = traj_aw(a.n, a.m, a.dyn, &a.cost);
  const bool ls16 = rw && gmpc_ls16_shape(a);
  a.ls_split = ls16 ? gmpc_ls16_split() : 0;
  const bool ls32 = ls16 && gmpc_ls32_split() > 0 && gmpc_ls32_shape(a) && (long)a.B * GMPC_LS_ITEMS >= gmpc_ls32_split();
  a.ls32_split = ls32 ? gmpc_ls32_split() : 0;
  const size_t lds = eval ? 0 : rw ? gmpc_traj_rw_lds(a) : traj_lds(a);
  static bool attr = false;
  if (!attr && !eval) { traj_attr(&k_traj<true>); attr = true; }
  // halvings allowed by trajax' loop: candidate k runs while alpha_0 / 2^k > alpha_min
  int k_max = 0;
  for (float al = a.alpha_0; al > a.alpha_min && k_max < 4096; al *= 0.5f) ++k_max;
  // Size of the first round: one more candidate than the previous search accepted -- or, where a round of 16- or
  // 32-candidate workgroups runs anyway and has room, all GMPC_LS_ITEMS of them: a pass of k_ls32 over the chip holds
  // 8192 candidates (k_ls16: 4096) and takes the same time half empty.  At C3 the previous rule filled it to 7965 and
  // left ~10 trajectories per iteration whose step size had grown by more than four halvings with a THIRD round of
  // their own (a k_traj_rw pass, 0.2 - 0.4 ms for 80 candidates, in most iterations); with the full first round a
  // third round needs 13 halvings (20 of 102,400 searches).  Which candidate is accepted does not change.
  int first_min = 1;
  if (ls16) {
    static const int ncu = []() {
      int v = 256;
      (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0);
      return v;
    }();
    const long all = (long)a.B * GMPC_LS_ITEMS;
    const long pass = all >= a.ls32_split && ls32 ? (long)ncu * 32 : (long)ncu * 16;
    if (all >= a.ls_split && all <= pass) first_min = GMPC_LS_ITEMS;
  }
  // worst case: a first round of first_min candidates, a second of GMPC_LS_NEXT, then GMPC_LS_ITEMS per round (a third
  // round is rare and every round enqueued costs four launches whether it finds work or not: the later rounds take all
  // they can hold; trajax' 15 step sizes are 4 rounds after a one-candidate first round, 3 after a full one)
  int rounds = 0;
  for (int left = k_max, r = 0; left > 0; ++r, ++rounds) left -= r == 0 ? first_min : r == 1 ? GMPC_LS_NEXT : GMPC_LS_ITEMS;
  if (rounds > GMPC_LS_ROUNDS_MAX) return -1;
  hipLaunchKernelGGL(k_ls_init, dim3((a.B + 255) / 256), dim3(256), 0, s, a.B, a.active, a.alpha_0,
                     a.alpha_min, k_max, first_min, a.iters, w.run, w.cnt, w.kfirst, w.prevk, a.alpha, a.U_step,
                     a.obj_step);
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(k_ls_place, dim3(1), dim3(1024), 0, s, a.B, w.cnt, w.kfirst, w.item_b[0], w.item_k[0],
                       w.slot, w.counts + r, w.counts + GMPC_LS_ROUNDS_MAX,
                       w.counts + GMPC_LS_ROUNDS_MAX + 1 + 24 + (r < GMPC_LS_STATS - 24 ? r : GMPC_LS_STATS - 25));
    a.item_b = w.item_b[0]; a.item_k = w.item_k[0]; a.nitems = w.counts + r; a.objc = w.objc;
    const long max_items = (long)a.B * (r == 1 ? GMPC_LS_NEXT : GMPC_LS_ITEMS);
    const int lsgrid = (int)((max_items + GMPC_TB - 1) / GMPC_TB);
    if (eval)
      eval(user, a, (int)max_items, s);
    else if (rw) {
      // short lists: 4 candidates per workgroup; long lists: 16 (each kernel returns on the other's rounds)
      // (three forms, each launch returns at once when the round's count is another form's: 4 candidates per
      // workgroup for short lists, 16 up to one pass over the chip, two groups of 16 beyond)
      gmpc_launch_traj_rw(a, true, lsgrid, lds, s);
      if (ls16) gmpc_launch_ls16(a, max_items, s);
      if (ls32) gmpc_launch_ls32(a, max_items, a.ls32_split, s);
    } else
      hipLaunchKernelGGL(k_traj<true>, dim3((unsigned)lsgrid), dim3(GMPC_TRAJ_THREADS), lds, s, a);
    LsDecideArgs d;
    d.n = a.n; d.m = a.m; d.T = a.T; d.Lh = a.dyn.L - 1; d.k_max = k_max;
    d.next = r == 0 ? GMPC_LS_NEXT : GMPC_LS_ITEMS;      // size of round r + 1
    d.alpha_0 = a.alpha_0;
    d.slot = w.slot; d.cnt = w.cnt; d.kfirst = w.kfirst; d.prevk = w.prevk; d.run = w.run;
    d.objc = w.objc; d.Xc = a.Xc; d.Uc = a.Uc; d.maskc = a.maskc;
    d.X = a.X; d.U = a.Uio; d.masks = a.masks;
    d.obj = a.obj; d.obj_step = a.obj_step; d.U_step = a.U_step; d.alpha = a.alpha;
    d.stats = w.counts + GMPC_LS_ROUNDS_MAX + 1;
    hipLaunchKernelGGL(k_ls_decide, dim3(a.B), dim3(GMPC_THREADS), 0, s, d);
    if (r == 0 && split != nullptr) {
      hipLaunchKernelGGL(k_ls_split, dim3(1), dim3(1024), 0, s, a.B, a.active, w.run, split->cap, GMPC_LS_NEXT,
                         a.ls_split, split->wg_max, split->tlist, split->tcount, split->llist, split->lcount);
      if (hipEventRecord(split->ev, s) != hipSuccess) return -2;
    }
  }
  return 0;
}
// true when the later rounds of this shape's line search run on k_ls16 (16 candidates per workgroup, one per CU)
bool gmpc_ls_rounds_on_ls16(const TrajArgs& a0) {
  TrajArgs a = a0;
  return gmpc_traj_rw_shape(a) && gmpc_ls16_shape(a);
}
void gmpc_launch_masks(int B, int n, int m, int T, const MlpDesc& dyn, const float* X,
                       const float* U, uint32_t* masks, hipStream_t s) {
  const int NS = B * T;
  const int aw = traj_aw(n, m, dyn, nullptr);
  hipLaunchKernelGGL(k_masks, dim3((NS + 3) / 4), dim3(GMPC_THREADS), 2 * (size_t)aw * sizeof(float4), s,
                     NS, n, m, T, dyn, X, U, masks, aw);
}
