// One-wave form of the iLQR Riccati sweep (k_riccati_w<N, M>, mode 0 of k_riccati, gmpc_backward.hip) for
// the reference's state / action sizes: one wavefront per trajectory, no workgroup barriers, and the three
// matrix products of a step on the matrix pipe instead of 15 k multiply-adds with two LDS operands each:
//     W = P [A | B]                          (n x (n+m); P symmetric, so P is the "TN" left operand)
//     Z = [A | B]^T W = [[A^T P A, A^T P B], [B^T P A, B^T P B]]      -> T1, H, G_r in one tile
//     Z[:n, :n] += [K; V]^T [V; K]           (V = H + G K / 2: K^T V + V^T K = K^T H + H^T K + K^T G K)
// as v_mfma_f32_32x32x2_f32 with the operands zero-padded to 32 columns and an even row count in LDS (9 + 9 +
// M MFMAs per step); every operand row is a k-step, so nothing is transposed.  The 32 x 32 accumulator tile
// of a lane holds column (lane & 31) and rows (rg & 3) + 8 (rg >> 2) + 4 (lane >> 5).  What stays on the
// vector pipe: the two vector recursions (lambda, p), the M x M Cholesky (lane 0, in registers, as in
// k_riccati) and the substitutions (one right-hand-side column per lane).
// Same recursion as k_riccati (trajax lqr_step / tvlqr with delta = 1e-8, adjoint); the association of the
// products differs (A^T (P A) instead of (A^T P) A, the K terms through V), the results agree to rounding.
// Reference arithmetic: trajax tvlqr as called from policy/optimizers.py:19,41; cost/cost_model.py:20-31.
#include "gmpc_device.h"
#include <cstdlib>
#include <cstring>

typedef float f32x16_w __attribute__((ext_vector_type(16)));

// The workgroup is ONE wave: its LDS instructions execute in program order, so a value written by one lane is
// there for the lane that reads it later -- what is needed between two phases is only that hipcc keeps them in
// order.  __syncthreads() also drains the vector-memory counter (the next step's prefetch, the K / k / gradient
// stores of this one: a round trip to the Infinity Cache or HBM at every one of the step's 12 phase boundaries).
#ifdef GMPC_RICCATI_W_BARRIERS
#define RW_SYNC() __syncthreads()
#else
// wave_barrier alone is a scheduling barrier the optimiser models as touching no memory: the wavefront-scope fences
// around it are what tells the compiler that LDS stores before it are visible to the loads after it (they emit no
// instruction at this scope -- in particular no s_waitcnt vmcnt --, the hardware keeps one wave's LDS accesses in order)
#define RW_SYNC()                                             \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)
#endif

template <int N_, int M_>
__global__ __launch_bounds__(64) void k_riccati_w(RiccatiArgs a) {
  constexpr int n = N_, m = M_, nm = n + m, LD = 32;
  constexpr int NR = (n + 1) & ~1;              // operand rows, padded to whole k-steps (2 rows each)
  constexpr int KP = NR / 2;                    // k-steps of the n-row products
  static_assert(nm <= 32 && m <= 8 && n >= m, "one 32 x 32 tile");
  __shared__ float Xs[NR * LD], Ps[NR * LD], Ws[NR * LD], Ss[n * LD];
  __shared__ float KVs[2 * m * LD], VKs[2 * m * LD], Hm[m * LD], HGK[m * LD], Kk[m * LD];
  __shared__ float Gr[m * m], Gp[m * m], G[m * m], Lc[m * m];
  __shared__ float pv[LD], lam[LD], lamn[LD], Ap[LD], dv[LD], qv[LD], uv[8], rv[8], hv[8];
  const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x, T = a.T;
  if (a.active != nullptr && a.active[b] == 0) return;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]);
  const float al = GMPC_ALPHA;
  const float delta = 1e-8f;
  const int ng = a.ng > 0 ? a.ng : n;

  for (int e = lane; e < NR * LD; e += 64) { Xs[e] = 0.f; Ps[e] = 0.f; Ws[e] = 0.f; }
  for (int e = lane; e < 2 * m * LD; e += 64) { KVs[e] = 0.f; VKs[e] = 0.f; }
  RW_SYNC();
  for (int e = lane; e < n * n; e += 64) Ps[(e / n) * LD + e % n] = a.QT[(size_t)b * n * n + e];
  if (lane < n) {
    const float q = a.qT[(size_t)b * n + lane];
    pv[lane] = q;
    lam[lane] = q;
    if (a.adj) a.adj[((size_t)b * (T + 1) + T) * n + lane] = q;
  }
  float gn2 = 0.f;

  // the next step's [A | B], x - goal and u are requested at the top of a step and installed at its end
  constexpr int PFN = (n * nm + 63) / 64;
  float pf_ab[PFN];
  float pf_d = 0.f, pf_u = 0.f;
  auto prefetch = [&](int tp) {
    const size_t btp = (size_t)b * T + tp;
#pragma unroll
    for (int r = 0; r < PFN; ++r) {
      const int e = lane + r * 64;
      pf_ab[r] = e < n * nm ? a.AB[btp * n * nm + e] : 0.f;
    }
    if (lane < n)
      pf_d = lane < ng ? a.X[((size_t)b * (T + 1) + tp) * n + lane] - a.goal[((size_t)b * (T + 1) + tp) * ng + lane]
                       : 0.f;
    if (lane < m) pf_u = a.U[btp * m + lane];
  };
  auto commit = [&]() {
#pragma unroll
    for (int r = 0; r < PFN; ++r) {
      const int e = lane + r * 64;
      if (e < n * nm) Xs[(e / nm) * LD + e % nm] = pf_ab[r];
    }
    if (lane < n) dv[lane] = pf_d;
    if (lane < m) uv[lane] = pf_u;
  };
  prefetch(T - 1);
  commit();
  RW_SYNC();

#ifdef GMPC_RICCATI_STAMPS
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
#define RS_(i) { __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long t_ = __builtin_readcyclecounter(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define RS_(i)
#endif
  for (int t = T - 1; t >= 0; --t) {
    const size_t bt = (size_t)b * T + t;
    if (t > 0) prefetch(t - 1);
    // ---- stage-cost scalars; q_t, r_t; the two vector recursions through [A | B]^T
    float dd = 0.f, uu = 0.f;
#pragma unroll
    for (int i = 0; i < n; ++i) dd = fmaf(dv[i], dv[i], dd);
#pragma unroll
    for (int j = 0; j < m; ++j) uu = fmaf(uv[j], uv[j], uu);
    const float s = sqrtf(dd + al * al), su = sqrtf(uu + al * al);
    const float is = 1.f / s, is3 = is * is * is, isu = 1.f / su, isu3 = isu * isu * isu;
    if (lane < nm) {
      float vl = 0.f, vp = 0.f;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        const float x = Xs[k * LD + lane];
        vl = fmaf(x, lam[k], vl);
        vp = fmaf(x, pv[k], vp);
      }
      if (lane < n) {
        const float q = w1 * dv[lane] * is;
        qv[lane] = q;
        lamn[lane] = q + vl;                      // lam_t = q_t + A^T lam
        Ap[lane] = vp;                            // A^T p
      } else {
        const int j = lane - n;
        const float r = w0 * uv[j] * isu;
        rv[j] = r;
        const float g = r + vl;                   // g_t = r_t + B^T lam
        gn2 = fmaf(g, g, gn2);
        if (a.grad) a.grad[bt * m + j] = g;
        hv[j] = r + vp;                           // h = r_t + B^T p
      }
    }
    RS_(0)
    // ---- W = P [A | B]
    f32x16_w acc;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[(2 * kk + half) * LD + l31], Xs[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < NR) Ws[row * LD + l31] = acc[rg];          // (rows n .. NR - 1 come out zero: P's padding)
    }
    RS_(1)
    RW_SYNC();
    // ---- Z = [A | B]^T W
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[(2 * kk + half) * LD + l31], Ws[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
    // rows n .. n + m - 1: H (columns < n) and G_r (columns n ..)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row >= n && row < nm) {
        if (l31 < n) Hm[(row - n) * LD + l31] = acc[rg];
        else if (l31 < nm) Gr[(row - n) * m + l31 - n] = acc[rg];
      }
    }
    RS_(2)
    RW_SYNC();
    // ---- G = sym(R + G_r), Cholesky of G + delta I, [K k] = -(G + delta I)^-1 [H h]
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      const float Rij = w0 * ((i == j ? isu : 0.f) - uv[i] * uv[j] * isu3);
      Gp[lane] = Rij + Gr[lane];
    }
    RW_SYNC();
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      G[lane] = (Gp[lane] + Gp[j * m + i]) * 0.5f;
    }
    RW_SYNC();
    RS_(3)
    if (lane == 0) {
      float Lr[m][m];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        float sdiag = G[j * m + j] + delta;
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= Lr[j][k] * Lr[j][k];
        const float di = 1.0f / sqrtf(sdiag);      // the diagonal is kept as its reciprocal (k_riccati)
        Lr[j][j] = di;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          float v = G[i * m + j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= Lr[i][k] * Lr[j][k];
          Lr[i][j] = v * di;
        }
      }
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lc[i * m + j] = Lr[i][j];
    }
    RS_(4)
    RW_SYNC();
    if (lane <= n) {
      const int c = lane;
      float Lr[m][m], y[m];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lr[i][j] = Lc[i * m + j];
#pragma unroll
      for (int i = 0; i < m; ++i) {
        float v = c < n ? Hm[i * LD + c] : hv[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= Lr[i][k] * y[k];
        y[i] = v * Lr[i][i];
      }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) {
        float v = y[i];
#pragma unroll
        for (int k = i + 1; k < m; ++k) v -= Lr[k][i] * y[k];
        y[i] = v * Lr[i][i];
      }
#pragma unroll
      for (int i = 0; i < m; ++i) Kk[i * LD + c] = -y[i];      // column n: k_t
    }
    RS_(5)
    RW_SYNC();
    // ---- outputs K_t, k_t; H + G K; the stacked operands [K; V], [V; K]
    for (int e = lane; e < m * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float kij = Kk[i * LD + j];
      if (a.K) a.K[bt * m * n + e] = kij;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) v = fmaf(G[i * m + k], Kk[k * LD + j], v);
      const float h = Hm[i * LD + j];
      HGK[i * LD + j] = h + v;
      const float vv = fmaf(0.5f, v, h);
      KVs[i * LD + j] = kij; KVs[(m + i) * LD + j] = vv;
      VKs[i * LD + j] = vv;  VKs[(m + i) * LD + j] = kij;
    }
    if (a.k && lane < m) a.k[bt * m + lane] = Kk[lane * LD + n];
    RS_(6)
    RW_SYNC();
    // ---- S = A^T P A + K^T V + V^T K  (the accumulator still holds Z), P = Q_t + sym(S)
#pragma unroll
    for (int kk = 0; kk < m; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(KVs[(2 * kk + half) * LD + l31], VKs[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < n) Ss[row * LD + l31] = acc[rg];
    }
    // p = q + A^T p + (H + G K)^T k + K^T h
    float pn = 0.f;
    if (lane < n) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) {
        v1 = fmaf(HGK[k * LD + lane], Kk[k * LD + n], v1);
        v2 = fmaf(Kk[k * LD + lane], hv[k], v2);
      }
      pn = ((qv[lane] + Ap[lane]) + v1) + v2;
    }
    RS_(7)
    RW_SYNC();
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float Qij = w1 * ((i == j && i < ng ? is : 0.f) - dv[i] * dv[j] * is3);
      Ps[i * LD + j] = Qij + (Ss[i * LD + j] + Ss[j * LD + i]) * 0.5f;
    }
    if (lane < n) {
      pv[lane] = pn;
      const float ln = lamn[lane];
      lam[lane] = ln;
      if (a.adj) a.adj[((size_t)b * (T + 1) + t) * n + lane] = ln;
    }
    RS_(8)
    RW_SYNC();                               // (dv, uv, Xs of this step are dead)
    if (t > 0) commit();
    RW_SYNC();
    RS_(9)
  }
#ifdef GMPC_RICCATI_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    printf("k_riccati_w cycles per step: scalars+vec %llu | W=P[AB] %llu | Z %llu | G sym %llu | chol %llu | solve %llu | outputs %llu | S mfma+p %llu | P update %llu | commit %llu\n",
           st_[0] / T, st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T, st_[5] / T, st_[6] / T, st_[7] / T, st_[8] / T, st_[9] / T);
#endif

  if (a.cont != nullptr) {
    float un2 = 0.f;
    for (int e = lane; e < T * m; e += 64) {
      const float u = a.U[(size_t)b * T * m + e];
      un2 = fmaf(u, u, un2);
    }
    gn2 = wave_sum(gn2);
    un2 = wave_sum(un2);
    if (lane == 0) {
      float gn = sqrtf(gn2);
      if (isnan(gn)) gn = INFINITY;
      const float aobj = fabsf(a.obj[b]) + 1.0f;
      const float un = sqrtf(un2) + 1.0f;
      const bool progressing = (a.obj_step[b] > a.opts.obj_step_threshold * aobj) &&
                               (a.U_step[b] > a.opts.inputs_step_threshold * un);
      const bool potential = (gn > a.opts.grad_norm_threshold) &&
                             (gn > a.opts.relative_grad_norm_threshold * aobj);
      const bool go = (a.iters[b] < a.opts.maxiter) && progressing && potential &&
                      (a.alpha[b] > a.opts.alpha_min);
      a.cont[b] = go ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Two waves per trajectory (k_riccati_w2): the sweep is one dependent chain per step --
//   W = P [A|B] -> Z = [A|B]^T W -> G -> Cholesky -> [K k] -> S = Z_xx + [K;V]^T [V;K] -> P
// -- and a single wave also runs everything else of the step in that chain's program order: the stage-cost
// scalars (a 17-term sum, two square roots, the reciprocals), q_t, r_t, the adjoint recursion lambda_t = q_t +
// A_t^T lambda_{t+1} with the control gradient, A_t^T p and h = r_t + B_t^T p, the next step's operands coming in
// from global memory (stamps: 3.0 k + 1.0 k of the 13.9 k cycles of a step).  None of that needs this step's P.
// Wave 1 (the helper) does it one step AHEAD into double-buffered LDS; wave 0 (the chain) finds [A|B]_t, x - g, u,
// the scalars and q_t ready at the top of step t and h, A^T p before its solve.  Two workgroup barriers per step
// (LDS counter only: the helper's prefetch and the chain's K stores stay in flight across them).
// Same arithmetic, operation for operation, as k_riccati_w: the results are bit-identical.
// ------------------------------------------------------------------------------------------------
#define RW2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int N_, int M_>
__global__ __launch_bounds__(128) void k_riccati_w2(RiccatiArgs a) {
  constexpr int n = N_, m = M_, nm = n + m, LD = 32;
  constexpr int NR = (n + 1) & ~1;
  constexpr int KP = NR / 2;
  static_assert(nm <= 32 && m <= 8 && n >= m, "one 32 x 32 tile");
  __shared__ float Xs[2][NR * LD];                       // [A | B] of step t in buffer t & 1 (helper -> chain)
  __shared__ float Ps[NR * LD], Ws[NR * LD], Ss[n * LD];
  __shared__ float KVs[2 * m * LD], VKs[2 * m * LD], Hm[m * LD], HGK[m * LD], Kk[m * LD];
  __shared__ float Gr[m * m], Gp[m * m], G[m * m], Lc[m * m];
  __shared__ float dvb[2][LD], qvb[2][LD], uvb[2][8], rvb[2][8], scal[2][4];   // helper -> chain, buffer t & 1
  __shared__ float pv[LD], lam[LD], Ap[LD], hv[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x, T = a.T;
  if (a.active != nullptr && a.active[b] == 0) return;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]);
  const float al = GMPC_ALPHA;
  const float delta = 1e-8f;
  const int ng = a.ng > 0 ? a.ng : n;

  for (int e = tid; e < 2 * NR * LD; e += 128) (&Xs[0][0])[e] = 0.f;
  for (int e = tid; e < NR * LD; e += 128) { Ps[e] = 0.f; Ws[e] = 0.f; }
  for (int e = tid; e < 2 * m * LD; e += 128) { KVs[e] = 0.f; VKs[e] = 0.f; }
  RW2_BARRIER();

  if (wave == 1) {
    // ================= helper =================
    constexpr int PFN = (n * nm + 63) / 64;
    float pf_ab[PFN];
    float pf_d = 0.f, pf_u = 0.f;
    float gn2 = 0.f;
    auto prefetch = [&](int tp) {
      const size_t btp = (size_t)b * T + tp;
#pragma unroll
      for (int r = 0; r < PFN; ++r) {
        const int e = lane + r * 64;
        pf_ab[r] = e < n * nm ? a.AB[btp * n * nm + e] : 0.f;
      }
      if (lane < n)
        pf_d = lane < ng ? a.X[((size_t)b * (T + 1) + tp) * n + lane] - a.goal[((size_t)b * (T + 1) + tp) * ng + lane]
                         : 0.f;
      if (lane < m) pf_u = a.U[btp * m + lane];
    };
    // everything of step tp that needs neither P nor p: operands into buffer tp & 1, scalars, q, r, the adjoint
    auto prepare = [&](int tp) {
      const int bf = tp & 1;
      float* X_ = Xs[bf];
#pragma unroll
      for (int r = 0; r < PFN; ++r) {
        const int e = lane + r * 64;
        if (e < n * nm) X_[(e / nm) * LD + e % nm] = pf_ab[r];
      }
      if (lane < n) dvb[bf][lane] = pf_d;
      if (lane < m) uvb[bf][lane] = pf_u;
      RW_SYNC();
      if (tp > 0) prefetch(tp - 1);
      const float* dv = dvb[bf];
      const float* uv = uvb[bf];
      float dd = 0.f, uu = 0.f;
#pragma unroll
      for (int i = 0; i < n; ++i) dd = fmaf(dv[i], dv[i], dd);
#pragma unroll
      for (int j = 0; j < m; ++j) uu = fmaf(uv[j], uv[j], uu);
      const float s = sqrtf(dd + al * al), su = sqrtf(uu + al * al);
      const float is = 1.f / s, is3 = is * is * is, isu = 1.f / su, isu3 = isu * isu * isu;
      if (lane == 0) { scal[bf][0] = is; scal[bf][1] = is3; scal[bf][2] = isu; scal[bf][3] = isu3; }
      const size_t bt = (size_t)b * T + tp;
      float ln = 0.f;
      if (lane < nm) {
        float vl = 0.f;
#pragma unroll
        for (int k = 0; k < n; ++k) vl = fmaf(X_[k * LD + lane], lam[k], vl);
        if (lane < n) {
          const float q = w1 * dv[lane] * is;
          qvb[bf][lane] = q;
          ln = q + vl;                               // lam_t = q_t + A^T lam
        } else {
          const int j = lane - n;
          const float r = w0 * uv[j] * isu;
          rvb[bf][j] = r;
          const float g = r + vl;                    // g_t = r_t + B^T lam
          gn2 = fmaf(g, g, gn2);
          if (a.grad) a.grad[bt * m + j] = g;
        }
      }
      RW_SYNC();                                     // (every lane has read lam)
      if (lane < n) {
        lam[lane] = ln;
        if (a.adj) a.adj[((size_t)b * (T + 1) + tp) * n + lane] = ln;
      }
      RW_SYNC();
    };
    if (lane < n) {
      const float q = a.qT[(size_t)b * n + lane];
      pv[lane] = q;
      lam[lane] = q;
      if (a.adj) a.adj[((size_t)b * (T + 1) + T) * n + lane] = q;
    }
    RW_SYNC();
    prefetch(T - 1);
    prepare(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      RW2_BARRIER();                                 // S_t: p_{t+1} is in pv
      const int bf = t & 1;
      if (lane < nm) {
        float vp = 0.f;
#pragma unroll
        for (int k = 0; k < n; ++k) vp = fmaf(Xs[bf][k * LD + lane], pv[k], vp);
        if (lane < n) Ap[lane] = vp;                 // A^T p
        else hv[lane - n] = rvb[bf][lane - n] + vp;  // h = r_t + B^T p
      }
      RW2_BARRIER();                                 // V_t: h, A^T p are there for the chain's solve
      if (t > 0) prepare(t - 1);
    }
    if (a.cont != nullptr) {
      float un2 = 0.f;
      for (int e = lane; e < T * m; e += 64) {
        const float u = a.U[(size_t)b * T * m + e];
        un2 = fmaf(u, u, un2);
      }
      gn2 = wave_sum(gn2);
      un2 = wave_sum(un2);
      if (lane == 0) {
        float gn = sqrtf(gn2);
        if (isnan(gn)) gn = INFINITY;
        const float aobj = fabsf(a.obj[b]) + 1.0f;
        const float un = sqrtf(un2) + 1.0f;
        const bool progressing = (a.obj_step[b] > a.opts.obj_step_threshold * aobj) &&
                                 (a.U_step[b] > a.opts.inputs_step_threshold * un);
        const bool potential = (gn > a.opts.grad_norm_threshold) &&
                               (gn > a.opts.relative_grad_norm_threshold * aobj);
        const bool go = (a.iters[b] < a.opts.maxiter) && progressing && potential &&
                        (a.alpha[b] > a.opts.alpha_min);
        a.cont[b] = go ? 1 : 0;
      }
    }
    return;
  }

  // ================= the chain =================
  for (int e = lane; e < n * n; e += 64) Ps[(e / n) * LD + e % n] = a.QT[(size_t)b * n * n + e];
  for (int t = T - 1; t >= 0; --t) {
    const size_t bt = (size_t)b * T + t;
    const int bf = t & 1;
    RW2_BARRIER();                                   // S_t: [A | B]_t, x - g, u, scalars, q_t are in buffer bf
    const float* X_ = Xs[bf];
    const float* dv = dvb[bf];
    const float* uv = uvb[bf];
    const float is = scal[bf][0], is3 = scal[bf][1], isu = scal[bf][2], isu3 = scal[bf][3];
    // ---- W = P [A | B]
    f32x16_w acc;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[(2 * kk + half) * LD + l31], X_[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < NR) Ws[row * LD + l31] = acc[rg];
    }
    RW_SYNC();
    // ---- Z = [A | B]^T W
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X_[(2 * kk + half) * LD + l31], Ws[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row >= n && row < nm) {
        if (l31 < n) Hm[(row - n) * LD + l31] = acc[rg];
        else if (l31 < nm) Gr[(row - n) * m + l31 - n] = acc[rg];
      }
    }
    RW_SYNC();
    // ---- G = sym(R + G_r), Cholesky of G + delta I
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      const float Rij = w0 * ((i == j ? isu : 0.f) - uv[i] * uv[j] * isu3);
      Gp[lane] = Rij + Gr[lane];
    }
    RW_SYNC();
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      G[lane] = (Gp[lane] + Gp[j * m + i]) * 0.5f;
    }
    RW_SYNC();
    if (lane == 0) {
      float Lr[m][m];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        float sdiag = G[j * m + j] + delta;
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= Lr[j][k] * Lr[j][k];
        const float di = 1.0f / sqrtf(sdiag);
        Lr[j][j] = di;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          float v = G[i * m + j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= Lr[i][k] * Lr[j][k];
          Lr[i][j] = v * di;
        }
      }
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lc[i * m + j] = Lr[i][j];
    }
    RW2_BARRIER();                                   // V_t: h and A^T p have arrived
    // ---- [K k] = -(G + delta I)^-1 [H h]
    if (lane <= n) {
      const int c = lane;
      float Lr[m][m], y[m];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lr[i][j] = Lc[i * m + j];
#pragma unroll
      for (int i = 0; i < m; ++i) {
        float v = c < n ? Hm[i * LD + c] : hv[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= Lr[i][k] * y[k];
        y[i] = v * Lr[i][i];
      }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) {
        float v = y[i];
#pragma unroll
        for (int k = i + 1; k < m; ++k) v -= Lr[k][i] * y[k];
        y[i] = v * Lr[i][i];
      }
#pragma unroll
      for (int i = 0; i < m; ++i) Kk[i * LD + c] = -y[i];
    }
    RW_SYNC();
    // ---- outputs K_t, k_t; H + G K; the stacked operands [K; V], [V; K]
    for (int e = lane; e < m * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float kij = Kk[i * LD + j];
      if (a.K) a.K[bt * m * n + e] = kij;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) v = fmaf(G[i * m + k], Kk[k * LD + j], v);
      const float h = Hm[i * LD + j];
      HGK[i * LD + j] = h + v;
      const float vv = fmaf(0.5f, v, h);
      KVs[i * LD + j] = kij; KVs[(m + i) * LD + j] = vv;
      VKs[i * LD + j] = vv;  VKs[(m + i) * LD + j] = kij;
    }
    if (a.k && lane < m) a.k[bt * m + lane] = Kk[lane * LD + n];
    RW_SYNC();
    // ---- S = A^T P A + K^T V + V^T K, P = Q_t + sym(S)
#pragma unroll
    for (int kk = 0; kk < m; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(KVs[(2 * kk + half) * LD + l31], VKs[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < n) Ss[row * LD + l31] = acc[rg];
    }
    // p = q + A^T p + (H + G K)^T k + K^T h
    float pn = 0.f;
    if (lane < n) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) {
        v1 = fmaf(HGK[k * LD + lane], Kk[k * LD + n], v1);
        v2 = fmaf(Kk[k * LD + lane], hv[k], v2);
      }
      pn = ((qvb[bf][lane] + Ap[lane]) + v1) + v2;
    }
    RW_SYNC();
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float Qij = w1 * ((i == j && i < ng ? is : 0.f) - dv[i] * dv[j] * is3);
      Ps[i * LD + j] = Qij + (Ss[i * LD + j] + Ss[j * LD + i]) * 0.5f;
    }
    if (lane < n) pv[lane] = pn;
    RW_SYNC();
  }
}


// ------------------------------------------------------------------------------------------------
// The bilevel Hessian solve (mode 1) in the two-wave form: k_riccati_w2h.  Same sweep with
//   * no regulariser (delta = 0) and G possibly indefinite (make_psd = False): the m x m systems are solved by
//     Gaussian elimination with partial pivoting, as k_riccati<N, M> does in mode 1 -- the factorisation on lane 0 in
//     registers, then one right-hand-side column per lane with the recorded pivots and multipliers (the same
//     operations in the same order as the serial elimination of k_riccati, which spent 0.9 ms of a 1.4 ms
//     gmpc_bilevel_grad on one lane walking m (n + 1) columns through LDS);
//   * linear terms q~ = 0, r~_t = -Bvec_t with Bvec_t = B_t^T mu_{t+1}, mu_T = lx_T, mu_t = lx_t + A_t^T mu_{t+1}
//     (policy/optimizers.py:78-83: the gradient of the upper loss with respect to the controls).  The helper wave
//     runs that adjoint recursion in the same backward sweep -- it is the recursion it runs for lambda in mode 0
//     with lx_t in the place of q_t -- and writes Bvec out: k_bvec (a launch of its own, 0.14 ms) is folded in;
//   * after the sweep, the forward tangent roll dU_t = k_t + K_t dX_t, dX_{t+1} = A_t dX_t + B_t dU_t on wave 0, the
//     next step's operands requested one step ahead.
// Reference: policy/optimizers.py:61-71, 86-105 (dense hessian + solve), restated as the structured solve of
// oracle/gan_mpc_oracle.py:hessian_solve.
// ------------------------------------------------------------------------------------------------
template <int N_, int M_>
__global__ __launch_bounds__(128) void k_riccati_w2h(RiccatiArgs a, const float* lx, float* bvec_out) {
  constexpr int n = N_, m = M_, nm = n + m, LD = 32;
  constexpr int NR = (n + 1) & ~1;
  constexpr int KP = NR / 2;
  static_assert(nm <= 32 && m <= 8 && n >= m, "one 32 x 32 tile");
  __shared__ float Xs[2][NR * LD];                       // [A | B] of step t in buffer t & 1 (helper -> chain)
  __shared__ float Ps[NR * LD], Ws[NR * LD], Ss[n * LD];
  __shared__ float KVs[2 * m * LD], VKs[2 * m * LD], Hm[m * LD], HGK[m * LD], Kk[m * LD];
  __shared__ float Gr[m * m], Gp[m * m], G[m * m], Lc[m * m];
  __shared__ int pivs[8];
  __shared__ float dvb[2][LD], uvb[2][8], rvb[2][8], scal[2][4];   // helper -> chain, buffer t & 1
  __shared__ float pv[LD], lam[LD], Ap[LD], hv[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x, T = a.T;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]);
  const float al = GMPC_ALPHA;
  const int ng = a.ng > 0 ? a.ng : n;

  for (int e = tid; e < 2 * NR * LD; e += 128) (&Xs[0][0])[e] = 0.f;
  for (int e = tid; e < NR * LD; e += 128) { Ps[e] = 0.f; Ws[e] = 0.f; }
  for (int e = tid; e < 2 * m * LD; e += 128) { KVs[e] = 0.f; VKs[e] = 0.f; }
  RW2_BARRIER();

  if (wave == 1) {
    // ================= helper =================
    constexpr int PFN = (n * nm + 63) / 64;
    float pf_ab[PFN];
    float pf_d = 0.f, pf_u = 0.f, pf_lx = 0.f;
    auto prefetch = [&](int tp) {
      const size_t btp = (size_t)b * T + tp;
#pragma unroll
      for (int r = 0; r < PFN; ++r) {
        const int e = lane + r * 64;
        pf_ab[r] = e < n * nm ? a.AB[btp * n * nm + e] : 0.f;
      }
      if (lane < n) {
        pf_d = lane < ng ? a.X[((size_t)b * (T + 1) + tp) * n + lane] - a.goal[((size_t)b * (T + 1) + tp) * ng + lane]
                         : 0.f;
        pf_lx = lx[((size_t)b * (T + 1) + tp) * n + lane];
      }
      if (lane < m) pf_u = a.U[btp * m + lane];
    };
    // everything of step tp that needs neither P nor p: operands into buffer tp & 1, scalars, the loss adjoint
    auto prepare = [&](int tp) {
      const int bf = tp & 1;
      float* X_ = Xs[bf];
#pragma unroll
      for (int r = 0; r < PFN; ++r) {
        const int e = lane + r * 64;
        if (e < n * nm) X_[(e / nm) * LD + e % nm] = pf_ab[r];
      }
      if (lane < n) dvb[bf][lane] = pf_d;
      if (lane < m) uvb[bf][lane] = pf_u;
      const float lxt = pf_lx;
      RW_SYNC();
      if (tp > 0) prefetch(tp - 1);
      const float* dv = dvb[bf];
      const float* uv = uvb[bf];
      float dd = 0.f, uu = 0.f;
#pragma unroll
      for (int i = 0; i < n; ++i) dd = fmaf(dv[i], dv[i], dd);
#pragma unroll
      for (int j = 0; j < m; ++j) uu = fmaf(uv[j], uv[j], uu);
      const float s = sqrtf(dd + al * al), su = sqrtf(uu + al * al);
      const float is = 1.f / s, is3 = is * is * is, isu = 1.f / su, isu3 = isu * isu * isu;
      if (lane == 0) { scal[bf][0] = is; scal[bf][1] = is3; scal[bf][2] = isu; scal[bf][3] = isu3; }
      const size_t bt = (size_t)b * T + tp;
      float ln = 0.f;
      if (lane < nm) {
        float vl = 0.f;
#pragma unroll
        for (int k = 0; k < n; ++k) vl = fmaf(X_[k * LD + lane], lam[k], vl);
        if (lane < n) {
          ln = lxt + vl;                             // mu_t = lx_t + A^T mu
        } else {
          const int j = lane - n;
          bvec_out[bt * m + j] = vl;                 // Bvec_t = B^T mu
          rvb[bf][j] = -vl;                          // the Riccati sweep's linear term r~_t
        }
      }
      RW_SYNC();                                     // (every lane has read mu)
      if (lane < n) lam[lane] = ln;
      RW_SYNC();
    };
    if (lane < n) {
      pv[lane] = 0.f;
      lam[lane] = lx[((size_t)b * (T + 1) + T) * n + lane];
    }
    RW_SYNC();
    prefetch(T - 1);
    prepare(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      RW2_BARRIER();                                 // S_t: p_{t+1} is in pv
      const int bf = t & 1;
      if (lane < nm) {
        float vp = 0.f;
#pragma unroll
        for (int k = 0; k < n; ++k) vp = fmaf(Xs[bf][k * LD + lane], pv[k], vp);
        if (lane < n) Ap[lane] = vp;                 // A^T p
        else hv[lane - n] = rvb[bf][lane - n] + vp;  // h = -Bvec_t + B^T p
      }
      RW2_BARRIER();                                 // V_t: h, A^T p are there for the chain's solve
      if (t > 0) prepare(t - 1);
    }
    RW2_BARRIER();                                   // F: the sweep is over (the chain rolls the tangent forward)
    return;
  }

  // ================= the chain =================
  for (int e = lane; e < n * n; e += 64) Ps[(e / n) * LD + e % n] = a.QT[(size_t)b * n * n + e];
  for (int t = T - 1; t >= 0; --t) {
    const size_t bt = (size_t)b * T + t;
    const int bf = t & 1;
    RW2_BARRIER();                                   // S_t: [A | B]_t, x - g, u, scalars are in buffer bf
    const float* X_ = Xs[bf];
    const float* dv = dvb[bf];
    const float* uv = uvb[bf];
    const float is = scal[bf][0], is3 = scal[bf][1], isu = scal[bf][2], isu3 = scal[bf][3];
    // ---- W = P [A | B]
    f32x16_w acc;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[(2 * kk + half) * LD + l31], X_[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < NR) Ws[row * LD + l31] = acc[rg];
    }
    RW_SYNC();
    // ---- Z = [A | B]^T W
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X_[(2 * kk + half) * LD + l31], Ws[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row >= n && row < nm) {
        if (l31 < n) Hm[(row - n) * LD + l31] = acc[rg];
        else if (l31 < nm) Gr[(row - n) * m + l31 - n] = acc[rg];
      }
    }
    RW_SYNC();
    // ---- G = sym(R + G_r); LU with partial pivoting (jax.scipy.linalg.solve) on lane 0, in registers
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      const float Rij = w0 * ((i == j ? isu : 0.f) - uv[i] * uv[j] * isu3);
      Gp[lane] = Rij + Gr[lane];
    }
    RW_SYNC();
    if (lane < m * m) {
      const int i = lane / m, j = lane - i * m;
      G[lane] = (Gp[lane] + Gp[j * m + i]) * 0.5f;
    }
    RW_SYNC();
    if (lane == 0) {
      float Lr[m][m];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) Lr[i][j] = G[i * m + j];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        int piv = j;
        float best = fabsf(Lr[j][j]);
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          const float v = fabsf(Lr[i][j]);
          if (v > best) { best = v; piv = i; }
        }
        pivs[j] = piv;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          if (piv == i) {
            // (columns >= j only: the multipliers already stored in columns < j belong to the row POSITIONS, because
            // the right-hand-side columns below apply swap j and elimination j in turn, like the serial elimination)
#pragma unroll
            for (int c = j; c < m; ++c) { const float t_ = Lr[j][c]; Lr[j][c] = Lr[i][c]; Lr[i][c] = t_; }
          }
        }
        const float d = Lr[j][j];
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          const float f = Lr[i][j] / d;
#pragma unroll
          for (int c = j; c < m; ++c) Lr[i][c] -= f * Lr[j][c];
          Lr[i][j] = f;                              // the multiplier, for the right-hand-side columns
        }
      }
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) Lc[i * m + j] = Lr[i][j];
    }
    RW2_BARRIER();                                   // V_t: h and A^T p have arrived
    // ---- [K k] = -G^-1 [H h]: one column per lane -- row swaps, elimination with the multipliers, back substitution
    if (lane <= n) {
      const int c = lane;
      float Lr[m][m], y[m];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) Lr[i][j] = Lc[i * m + j];
#pragma unroll
      for (int i = 0; i < m; ++i) y[i] = c < n ? Hm[i * LD + c] : hv[i];
#pragma unroll
      for (int j = 0; j < m; ++j) {
        const int piv = pivs[j];
#pragma unroll
        for (int i = j + 1; i < m; ++i)
          if (piv == i) { const float t_ = y[j]; y[j] = y[i]; y[i] = t_; }
#pragma unroll
        for (int i = j + 1; i < m; ++i) y[i] -= Lr[i][j] * y[j];
      }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) {
        float v = y[i];
#pragma unroll
        for (int k = i + 1; k < m; ++k) v -= Lr[i][k] * y[k];
        y[i] = v / Lr[i][i];
      }
#pragma unroll
      for (int i = 0; i < m; ++i) Kk[i * LD + c] = -y[i];
    }
    RW_SYNC();
    // ---- outputs K_t, k_t; H + G K; the stacked operands [K; V], [V; K]
    for (int e = lane; e < m * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float kij = Kk[i * LD + j];
      a.K[bt * m * n + e] = kij;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) v = fmaf(G[i * m + k], Kk[k * LD + j], v);
      const float h = Hm[i * LD + j];
      HGK[i * LD + j] = h + v;
      const float vv = fmaf(0.5f, v, h);
      KVs[i * LD + j] = kij; KVs[(m + i) * LD + j] = vv;
      VKs[i * LD + j] = vv;  VKs[(m + i) * LD + j] = kij;
    }
    if (lane < m) a.k[bt * m + lane] = Kk[lane * LD + n];
    RW_SYNC();
    // ---- S = A^T P A + K^T V + V^T K, P = Q_t + sym(S)
#pragma unroll
    for (int kk = 0; kk < m; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(KVs[(2 * kk + half) * LD + l31], VKs[(2 * kk + half) * LD + l31], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < n) Ss[row * LD + l31] = acc[rg];
    }
    // p = A^T p + (H + G K)^T k + K^T h   (q~ = 0)
    float pn = 0.f;
    if (lane < n) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int k = 0; k < m; ++k) {
        v1 = fmaf(HGK[k * LD + lane], Kk[k * LD + n], v1);
        v2 = fmaf(Kk[k * LD + lane], hv[k], v2);
      }
      pn = (Ap[lane] + v1) + v2;
    }
    RW_SYNC();
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const float Qij = w1 * ((i == j && i < ng ? is : 0.f) - dv[i] * dv[j] * is3);
      Ps[i * LD + j] = Qij + (Ss[i * LD + j] + Ss[j * LD + i]) * 0.5f;
    }
    if (lane < n) pv[lane] = pn;
    RW_SYNC();
  }
  RW2_BARRIER();                                     // F: (the helper has left the LDS buffers alone since V_0)
  // ================= forward tangent roll (this wave; the gains of step t come back from global memory -- this
  // wave's own stores, drained first) =================
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  constexpr int PFA = (n * nm + 63) / 64, PFK = (m * n + 63) / 64;
  float fa[PFA], fk[PFK], fk0 = 0.f;
  auto fetch = [&](int tp) {
    const size_t btp = (size_t)b * T + tp;
#pragma unroll
    for (int r = 0; r < PFA; ++r) {
      const int e = lane + r * 64;
      fa[r] = e < n * nm ? a.AB[btp * n * nm + e] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < PFK; ++r) {
      const int e = lane + r * 64;
      fk[r] = e < m * n ? a.K[btp * m * n + e] : 0.f;
    }
    if (lane < m) fk0 = a.k[btp * m + lane];
  };
  float* const Xr = Xs[0];                           // [A | B]_t rows; the gains K_t in Ws; dX in pv, dU in hv
  if (lane < n) { pv[lane] = 0.f; a.dX[(size_t)b * (T + 1) * n + lane] = 0.f; }
  fetch(0);
  for (int t = 0; t < T; ++t) {
    const size_t bt = (size_t)b * T + t;
#pragma unroll
    for (int r = 0; r < PFA; ++r) {
      const int e = lane + r * 64;
      if (e < n * nm) Xr[(e / nm) * LD + e % nm] = fa[r];
    }
#pragma unroll
    for (int r = 0; r < PFK; ++r) {
      const int e = lane + r * 64;
      if (e < m * n) Ws[(e / n) * LD + e % n] = fk[r];
    }
    const float k0 = fk0;
    RW_SYNC();
    if (t + 1 < T) fetch(t + 1);
    if (lane < m) {
      float v = k0;
#pragma unroll
      for (int i = 0; i < n; ++i) v = fmaf(Ws[lane * LD + i], pv[i], v);
      hv[lane] = v;
      a.Hout[bt * m + lane] = v;
    }
    RW_SYNC();
    float xn = 0.f;
    if (lane < n) {
#pragma unroll
      for (int k = 0; k < n; ++k) xn = fmaf(Xr[lane * LD + k], pv[k], xn);
#pragma unroll
      for (int k = 0; k < m; ++k) xn = fmaf(Xr[lane * LD + n + k], hv[k], xn);
    }
    RW_SYNC();
    if (lane < n) {
      pv[lane] = xn;
      a.dX[((size_t)b * (T + 1) + t + 1) * n + lane] = xn;
    }
    RW_SYNC();
  }
}

// the shapes the one-wave form is instantiated for (mode 0 only); GMPC_RICCATI=valu keeps k_riccati
bool gmpc_riccati_w_shape(const RiccatiArgs& a) {
  const char* e = getenv("GMPC_RICCATI");
  if (e != nullptr && strcmp(e, "valu") == 0) return false;
  return a.mode == 0 && a.Phi == nullptr && a.n == 17 && a.m == 6;
}
// the Hessian solve (mode 1) with the loss adjoint folded in: same shapes, no curvature term (MLP dynamics)
bool gmpc_riccati_w2h_shape(const RiccatiArgs& a) {
  const char* e = getenv("GMPC_RICCATI");
  if (e != nullptr && strcmp(e, "valu") == 0) return false;
  return a.mode == 1 && a.Phi == nullptr && a.active == nullptr && a.n == 17 && a.m == 6 && (a.ng == 0 || a.ng == a.n);
}
void gmpc_launch_riccati_w2h(const RiccatiArgs& a, const float* lx, float* bvec_out, hipStream_t s) {
  hipLaunchKernelGGL((k_riccati_w2h<17, 6>), dim3(a.B), dim3(128), 0, s, a, lx, bvec_out);
}
void gmpc_launch_riccati_w(const RiccatiArgs& a, hipStream_t s) {
  const char* e = getenv("GMPC_RICCATI_W");          // "1": the one-wave form (A/B timing; read per call: tests)
  if (e != nullptr && e[0] == '1') hipLaunchKernelGGL((k_riccati_w<17, 6>), dim3(a.B), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_riccati_w2<17, 6>), dim3(a.B), dim3(128), 0, s, a);
}
