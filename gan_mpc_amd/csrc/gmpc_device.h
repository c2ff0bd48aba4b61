// Device-side building blocks shared by the gfx950 kernels of libgan_mpc_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>
#include <type_traits>

#include "../../include/gan_mpc_amd.h"

#define GMPC_MW 8          // 32-bit mask words per hidden layer (hidden width <= 256)
#define GMPC_THREADS 256   // workgroup size of the MLP kernels (4 wavefronts)
#define GMPC_TB 4          // trajectories per workgroup in the sequential kernels
#define GMPC_ALPHA 1e-2f   // smoothing constant of the stage cost (reference cost_model.py:22)

// A relu MLP as the kernels see it.  W[l] is the flax kernel (in,out) row-major, WT[l] its
// transpose (out,in) row-major (built once per gmpc_set_params), b[l] the bias.
struct MlpDesc {
  int L;
  int dims[GMPC_MAX_LAYERS + 1];
  const float* W[GMPC_MAX_LAYERS];
  const float* WT[GMPC_MAX_LAYERS];
  const float* b[GMPC_MAX_LAYERS];
};

// Sum over the 64 lanes, result in every lane.  Data-parallel-primitive moves instead of
// __shfl_xor: the shuffle compiles to ds_bpermute_b32, a round trip through the LDS crossbar (~150
// cycles each, six per sum), DPP row operations are plain VALU modifiers.  Steps: within quads, within
// rows of 16 (mirrors), then row_bcast15 / row_bcast31 carry the row sums into lane 63, which is
// broadcast with v_readlane.
__device__ __forceinline__ float wave_sum(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) -> float {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value,
                                                      decltype(row_mask)::value, 0xF, false));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});    // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});    // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{});   // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{});   // row_mirror
  v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});   // row_bcast15 -> rows 1, 3
  v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});   // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& a) {
  acc.x = fmaf(w, a.x, acc.x);
  acc.y = fmaf(w, a.y, acc.y);
  acc.z = fmaf(w, a.z, acc.z);
  acc.w = fmaf(w, a.w, acc.w);
}

// out[j][r] += sum_k W[k][j] * act[k][r] for one output neuron j per thread and 4*R4 rows r.
// W is (K,N) row-major in global memory: lanes read consecutive j (coalesced); act is an LDS
// image [K][R4] of float4, read as a wave-wide broadcast.  k runs in increasing order, so the
// result is a k-ordered fmaf chain per (j, r).
template <int R4>
__device__ __forceinline__ void dense_rows(const float* __restrict__ W, int K, int N, int j,
                                           const float4* act, float4 (&acc)[R4]) {
  if (j >= N) return;
  // UNR weight loads are issued before the FMAs that use them: the loop is bound by the L2 latency
  // of the weight stream, so the number of loads in flight per thread is what matters.
  constexpr int UNR = R4 <= 2 ? 16 : 8;
  const float* wp = W + j;
  int k = 0;
  for (; k + UNR <= K; k += UNR) {
    float w[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) w[u] = wp[(size_t)(k + u) * N];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int q = 0; q < R4; ++q) fma4(acc[q], w[u], act[(k + u) * R4 + q]);
  }
  for (; k + 4 <= K; k += 4) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = wp[(size_t)(k + u) * N];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < R4; ++q) fma4(acc[q], w[u], act[(k + u) * R4 + q]);
  }
  for (; k < K; ++k) {
    const float w = wp[(size_t)k * N];
#pragma unroll
    for (int q = 0; q < R4; ++q) fma4(acc[q], w, act[k * R4 + q]);
  }
}

// dense_rows with the weight matrix resident in LDS (bank-conflict free: lanes read consecutive j).
template <int R4>
__device__ __forceinline__ void dense_rows_lds(const float* W, int K, int N, int j, const float4* act,
                                               float4 (&acc)[R4]) {
  if (j >= N) return;
  const float* wp = W + j;
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const float w = wp[k * N];
#pragma unroll
    for (int q = 0; q < R4; ++q) fma4(acc[q], w, act[k * R4 + q]);
  }
}

// Same product for a narrow output (N << blockDim): the K range is split over NS = blockDim/N
// thread groups, partial sums meet in LDS and are added in segment order (deterministic).
// On return (after the trailing barrier) part[j*R4 + q] holds the sums; `part` needs
// NS*N*R4 float4 of LDS and must not alias `act`.  All threads of the block must call it.
template <int R4>
__device__ __forceinline__ void dense_small(const float* __restrict__ W, int K, int N,
                                            const float4* act, float4* part) {
  const int tid = threadIdx.x;
  int NS = blockDim.x / N;
  if (NS > K) NS = K;
  const int KS = (K + NS - 1) / NS;
  const int j = tid % N, seg = tid / N;
  if (seg < NS) {
    float4 acc[R4];
#pragma unroll
    for (int q = 0; q < R4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k0 = seg * KS;
    const int k1 = min(K, k0 + KS);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {   // 8 weight loads in flight per thread
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(k + u) * N + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < R4; ++q) fma4(acc[q], w[u], act[(k + u) * R4 + q]);
    }
    for (; k < k1; ++k) {
      const float w = W[(size_t)k * N + j];
#pragma unroll
      for (int q = 0; q < R4; ++q) fma4(acc[q], w, act[k * R4 + q]);
    }
#pragma unroll
    for (int q = 0; q < R4; ++q) part[(seg * N + j) * R4 + q] = acc[q];
  }
  __syncthreads();
  const int NE = N * R4;
  for (int e = tid; e < NE; e += blockDim.x) {
    float4 s = part[e];
    for (int sg = 1; sg < NS; ++sg) {
      const float4 p = part[sg * NE + e];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    part[e] = s;
  }
  __syncthreads();
}

// f4get / f4set: ONLY with a compile-time c (fully unrolled loops) on register values.  For a
// run-time component of an LDS float4 index the float view instead (see the note in k_traj).
__device__ __forceinline__ float f4get(const float4& v, int c) {
  return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}
__device__ __forceinline__ void f4set(float4& v, int c, float x) {
  if (c == 0) v.x = x; else if (c == 1) v.y = x; else if (c == 2) v.z = x; else v.w = x;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- kernel argument blocks (shared by the kernel units and gmpc_api.hip) ----
struct TrajArgs {
  int B, n, m, T;
  MlpDesc dyn, cost;
  const float* mpc_w;   // raw weights [3]
  const float* goal;    // [B][T+1][n]
  // plain rollout: x0, U in; X, costs, masks out
  const float* x0;      // [B][n]
  const float* U;       // [B][T][m]
  float* X;             // [B][T+1][n]
  float* costs;         // [B][T+1] or null
  float* obj;           // [B]
  uint32_t* masks;      // [B][T][Lh][GMPC_MW]
  // line search (LS): X,U,masks,obj are the current iterate and are updated in place on accept
  float* Uio;           // [B][T][m]
  const float* Kg;      // [B][T][m][n]
  const float* kg;      // [B][T][m]
  float* Xc;            // candidate buffers
  float* Uc;
  uint32_t* maskc;
  const int* active;    // [B] or null
  float* alpha;         // [B] out
  float* obj_step;      // [B] out
  float* U_step;        // [B] out
  int* iters;           // [B] in/out
  float alpha_0, alpha_min;
  int aw, pw;           // LDS sizing (float4 counts): activation buffers, partial-sum buffer
  int sw0, swl;         // floats of LDS holding W_0 (first layer) / W_L (output layer); 0: not staged
  // line search: the work list of this round; slot = one (trajectory, halving count) candidate,
  // candidate i writes Xc / Uc / maskc / objc at index i
  const int* item_b; const int* item_k; const int* nitems;
  float* objc;
  // work lists of at least ls_split items are k_ls16's, shorter ones k_traj_rw's (0: no split, see
  // gmpc_ls16.hip); both kernels are launched and read the round's count
  int ls_split;
  // ... and lists of at least ls32_split items k_ls32's (two groups of 16 per workgroup, gmpc_ls32.hip; 0: never)
  int ls32_split;
};

#define GMPC_LS_ITEMS 8   // candidates per trajectory held at once by the line search

// device workspace of the round-based line search (gmpc_traj.hip)
struct LsWork {
  int* item_b[2]; int* item_k[2];   // ping-pong work lists [maxB * GMPC_LS_ITEMS]
  int* first; int* cnt; int* kfirst;   // [maxB] this round's candidates of trajectory b: cnt halvings
                                       // kfirst .. kfirst+cnt-1; candidate j is item slot[b*8 + j]
  int* slot;                           // [maxB * GMPC_LS_ITEMS]
  int* prevk;                       // [maxB] halving count accepted by the previous line search
  int* counts;                      // [GMPC_LS_ROUNDS_MAX + 1] items per round, then GMPC_LS_STATS counters
                                    // since the solve began (gmpc_linesearch_stats): [0..15] line searches
                                    // that accepted halving k, [16] exhausted ones, [24 + r] items of round r
  int* run;                         // [maxB]
  float* objc;                      // [maxB * GMPC_LS_ITEMS]
};
// gmpc_launch_linesearch's optional hand-over after the first round (see there)
struct LsSplit {
  int* tlist; int* tcount;          // [cap] trajectories of the early chain, in index order; their number
  int* llist; int* lcount;          // [maxB] every other active trajectory, in index order; their number
  int cap;                          // most trajectories the early chain takes
  int wg_max;                       // second rounds with more 16-candidate workgroups than this keep the list empty
  hipEvent_t ev;
};
#define GMPC_LS_ROUNDS_MAX 40
#define GMPC_LS_STATS 64

struct RiccatiArgs {
  int B, n, m, T, mode;
  int ng;                // columns of `goal` (0: n); the staging cost sees xc[:ng]
  const float* X; const float* U; const float* goal; const float* mpc_w;
  const float* AB; const float* QT; const float* qT;
  const int* active;
  float* K; float* k; float* grad; float* adj;   // mode 0 outputs (may be null)
  // iLQR continuation test (mode 0, optional: cont != null)
  int* cont; const int* iters; const float* obj; const float* alpha; const float* obj_step;
  const float* U_step;
  gmpc_ilqr_opts opts;
  // mode 1
  const float* Bvec;   // [B][T][m]
  const float* Phi;    // [B][T][n+m][n+m] or null: lam_{t+1} . d^2 f / d(x,u)^2 (smooth dynamics, gmpc_dynl.hip)
  float* Hout;         // [B][T][m]
  float* dX;           // [B][T+1][n]
};

struct CriticDesc {
  int n, F, T1;              // input size, lstm features, sequence length T+1
  const float* Wcat;         // [(n+F)][4F]
  const float* WcatT;        // [4F][(n+F)]
  const float* b;            // [4F]
  MlpDesc head;              // dims[0] = F ... dims[L] = 1
};

// one weight-gradient problem  C[M][N] = sum_r A[r][:M]^T B[r][:N],  colsum[N] = sum_{r<cs_rows} B[r]
struct WgProb {
  int rows, M, N, lda, ldb, cs_rows;
  const float* A; const float* B;
  float* C; float* colsum;            // colsum may be null
  // filled by the launcher: tiling and the problem's slices of the partial-sum buffer
  int mstrips, ngroups, nchunks, rpc, item0;
  int cchunks, crpc, cs_block0;
  long part_off, cs_part_off;
  int red_block0, cs_red_block0;
};
#define GMPC_WG_MAX 8
struct WgBatch { int np; int gemm_blocks; WgProb p[GMPC_WG_MAX]; };

// expert sequence model (gmpc_expert.hip)
struct ExpertArgs {
  int B, n, m, T, hist, F;       // F == 0: the MLP variant (first = Dense(n -> h) + relu)
  const float* Wcat;             // LSTM: [(n+F)][4F];  MLP: first.W [n][h]
  const float* bcat;             // LSTM: [4F];         MLP: first.b [h]
  MlpDesc hx, hu;                // heads: dims[0] = F (or h) ... dims[L] = n / m
  const float* history;          // [B][hist+1][n]
  float* goal;                   // [B][T+1][n]
  float* U;                      // [B][T][m]
  int hw;                        // set by the launcher: LDS activations per head (>= every head width)
};

// batched "TN" GEMM of the large-state path (gmpc_large.hip)
struct BgemmArgs {
  int batch, M, N, K;
  const float* X; long sx; int ldx;     // X[b]: K x M row-major (leading dim ldx), batch stride sx
  const float* Y; long sy; int ldy;     // Y[b]: K x N
  float* C; long sc; int ldc;           // C[b]: M x N
  float alpha, beta;
  const int* active;                    // [batch] or null
  // optional second K-segment accumulated into the same product: C += X2^T Y2 (K2 rows)
  const float* X2 = nullptr; long sx2 = 0; int ldx2 = 0;
  const float* Y2 = nullptr; long sy2 = 0; int ldy2 = 0;
  int K2 = 0;
  // ... and a third one (k_bgemm_tn_lds only)
  const float* X3 = nullptr; long sx3 = 0; int ldx3 = 0;
  const float* Y3 = nullptr; long sy3 = 0; int ldy3 = 0;
  int K3 = 0;
  int upper_only = 0;   // square symmetric result: skip the blocks that lie entirely below the diagonal
  // optional epilogue extras (k_bgemm_tn_lds only)
  const float* E = nullptr; long se = 0; int lde = 0; int En = 0;   // C[row][col] += E[b][row][col], col < En
  const uint32_t* rowmask = nullptr; long srm = 0;                  // [b] bit words: row r of C is zeroed when bit r is clear
};

// workspace of the large-state backward pass (gmpc_large.hip)
// LSTM dynamics variant (gmpc_dynl.hip)
struct DynlDesc {
  int nx, F, m;          // x size, cell features, controls; state size N = nx + 2F
  const float* Wx;       // [(nx + m)][4F]
  const float* Wh;       // [F][4F]
  const float* b;        // [4F]
  MlpDesc tail;          // F -> hidden... -> nx
};

struct DynlTrajArgs {
  int B, T;
  DynlDesc d;
  MlpDesc cost;          // cost MLP on the full xc (N inputs)
  const float* mpc_w;
  const float* goal;     // [B][T+1][nx]
  // rollout
  const float* x0;       // [B][N]
  const float* U;        // [B][T][m]
  float* X;              // [B][T+1][N]
  float* costs;          // [B][T+1] or null
  float* obj;            // [B]
  // line search: work list of candidates, nominal iterate, gains, outputs per item
  const int* item_b; const int* item_k; const int* nitems;
  const float* Xn; const float* Un;       // nominal X [B][T+1][N], U [B][T][m]
  const float* Kg; const float* kg;       // [B][T][m][N], [B][T][m]
  float* Xc; float* Uc; float* objc;      // [items][T+1][N], [items][T][m], [items]
  float alpha_0;
  int width;             // LDS activation width: max(N, tail / cost widths)
};

struct BigWork {
  int ng = 0;            // columns of `goal` (0: n)
  int n, m, T;
  float *ABt, *P, *PAB, *T1, *HG, *KV, *VK, *pvec, *lam, *sbuf, *gn2;
  // low-rank form of the Jacobians (MLP dynamics whose last hidden width h < n / 2; see gmpc_large.hip):
  // A_t = I + W_L^T Vx_t^T, B_t = W_L^T Vu_t^T with V_t^T [h][n+m] per trajectory
  int h = 0;             // 0: dense form
  float *Vt = nullptr, *W1b = nullptr, *W2b = nullptr, *Sa = nullptr, *Sb = nullptr;
  float* Phi = nullptr;  // [B][n+m][n+m] curvature of one step (LSTM dynamics, bilevel solve)
  // one-step-ahead Jacobians (gmpc_big_backward): the step's [A | B] (dense form) or its factor V^T (low-rank form)
  // does not depend on P, so step t - 1's is produced on a side stream while step t's products run; second copies
  // of the buffers (step t uses copy t & 1; copy 0 is ABt / Vt), S of the low-rank form off the factor's scratch
  float *ABt2 = nullptr, *Vt2 = nullptr, *Sm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_start = nullptr, ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
};

// zero-padded weight copies read by the MFMA Jacobian chain (gmpc_linearize_mfma.hip)
struct LinPad {
  int NT, NTF, NGF;                     // column tiles: hidden GEMMs / input GEMM per group, groups
  const float* WLP;                     // [dims[Lh]+2][n]            rows >= dims[Lh] are zero
  const float* WTP[GMPC_MAX_LAYERS];    // l>=1: [dims[l+1]+2][32*NT]; l==0: [dims[1]+2][32*NTF]
  unsigned long long* dbg;              // diagnostic: per-segment cycle sums (null in production)
};

// ---- fp32 matrix-core building block -----------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- MFMA -> vector read fence ------------------------------------------------------------------
// gfx950 does not interlock a vector instruction that reads (or overwrites) the LAST registers of the destination of
// an MFMA still in flight: registers 14 / 15 of v_mfma_f32_32x32x2_f32 return their OLD value for 18 wait states,
// 2 / 3 of v_mfma_f32_16x16x4_f32 for 10, register 3 of v_mfma_f32_4x4x1_16b_f32 for 4 (the earlier registers read
// fresh at any distance; measured with tests/repro/mfma_wait_states.hip).  hipcc (ROCm 7.2) pads for that along the
// LAYOUT order of the basic blocks only: where a conditional branch skips a block lying between the MFMA and its
// first reader (`if (stamps) {...}`, `if (beta != 0) {...}`), the taken path comes out short -- the cause of the
// 3.8e-2 error of k_linearize_regs<2, 32> with the scheduling hints (tests/repro/README.md section 2).  Wherever
// an accumulator is first read in a block a branch can reach, the kernels put this fence between the two: every
// MFMA that writes `acc` is ordered before it, every reader after it, 18 wait states in between (volatile asm
// statements keep their order).  tests/repro/check_mfma_hazards.py walks the ISA of every kernel for what is left.
// AG: the accumulators live in the accumulation half of the register file ("a" registers: what hipcc picks for
// the one-wave-per-SIMD kernels); false: in "v" registers (the 256-register, two-waves-per-SIMD kernels).  The
// wrong choice costs a copy of every register in and out, not correctness.
template <bool AG, typename T>
__device__ __forceinline__ void mfma_tie(T& acc) {
  if constexpr (AG) asm volatile("" : "+a"(acc));
  else asm volatile("" : "+v"(acc));
}
template <bool AG, typename... A>
__device__ __forceinline__ void mfma_fence(A&... acc) {
  (mfma_tie<AG>(acc), ...);
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1");
  (mfma_tie<AG>(acc), ...);
}
template <bool AG, int NT, typename... A>
__device__ __forceinline__ void mfma_fence_tiles(f32x16 (&acc)[NT], A&... more) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) mfma_tie<AG>(acc[nt]);
  (mfma_tie<AG>(more), ...);
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1");
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) mfma_tie<AG>(acc[nt]);
  (mfma_tie<AG>(more), ...);
}

// One GEMM of the chain for one 32-row tile: acc[nt] += A[32 x Kp] * B[Kp x 32*NTT].
// bp0 points at this lane's element of B row `half` (row stride NP floats, any address space);
// afn(k0) returns this lane's A element of k-step k0 (k = k0 + half).  B/A of k-step k0+4 are
// requested before the MFMAs of k-step k0 issue (three register sets, no copies), so two k-steps
// of matrix work (2*NTT*64 cycles) cover the load latency even at one wave per SIMD.
// Reads run up to 3 k-steps past Kp: the padded operands provide those rows/columns.
template <int NTT, typename AF>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ bp0, int NP, int Kp, AF afn,
                                          f32x16 (&acc)[NTT]) {
  float b0[NTT], b1[NTT], b2[NTT];
  float a0, a1, a2;
  const float* bp = bp0;
#pragma unroll
  for (int nt = 0; nt < NTT; ++nt) b0[nt] = bp[nt * 32];
  a0 = afn(0);
  bp += 2 * NP;
#pragma unroll
  for (int nt = 0; nt < NTT; ++nt) b1[nt] = bp[nt * 32];
  a1 = afn(2);
  int k0 = 0;
  // sched_barrier(0) pins "request k-step k0+4, then run k-step k0": without it hipcc sinks the loads
  // below the MFMAs and the prefetch distance shrinks to one k-step.
  for (; k0 + 6 <= Kp; k0 += 6) {
    bp += 2 * NP;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) b2[nt] = bp[nt * 32];
    a2 = afn(k0 + 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[nt], acc[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bp += 2 * NP;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) b0[nt] = bp[nt * 32];
    a0 = afn(k0 + 6);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[nt], acc[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bp += 2 * NP;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) b1[nt] = bp[nt * 32];
    a1 = afn(k0 + 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2[nt], acc[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (k0 < Kp) {
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[nt], acc[nt], 0, 0, 0);
    if (k0 + 2 < Kp) {
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[nt], acc[nt], 0, 0, 0);
    }
  }
}

// A-operand source for the ring-buffered GEMMs below.  load() only ISSUES the reads of k-step k0 and
// returns the raw words; fin() turns them into the MFMA operand when the k-step is consumed, two or
// more k-steps later.  Splitting the two keeps the dependent VALU work (the relu select) away from
// the read, so no k-step waits for its own LDS round trip.
struct ARaw { float v; uint32_t w; };

// Same GEMM with B in the lane-interleaved layout Wi[k][32 lanes][8]: the (up to 8) column-tile
// values one lane needs for a k row are 32 contiguous bytes, fetched by two dwordx4 loads instead of
// NTT dword loads (measured 81 -> 73 cycles per MFMA at one wave per SIMD).  bp0 points at this
// lane's float4 pair of row `half`; the row stride is 64 float4.
template <int NTT, typename AL, typename AFIN>
__device__ __forceinline__ void gemm_tile_x4(const float4* __restrict__ bp0, int Kp, AL aload, AFIN afin,
                                             f32x16 (&acc)[NTT]) {
  static_assert(NTT <= 8, "at most 8 column tiles");
  float4 b0[2], b1[2], b2[2];
  ARaw a0, a1, a2;
  const float4* bp = bp0;
  auto ld = [&](float4 (&b)[2]) {
    b[0] = bp[0];
    if (NTT > 4) b[1] = bp[1];
  };
  auto mf = [&](const ARaw& ar, int k0, const float4 (&b)[2]) {
    const float a = afin(ar, k0);
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      const float4& q = b[nt >> 2];
      const float bv = (nt & 3) == 0 ? q.x : (nt & 3) == 1 ? q.y : (nt & 3) == 2 ? q.z : q.w;
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[nt], 0, 0, 0);
    }
  };
  ld(b0); a0 = aload(0); bp += 128;
  ld(b1); a1 = aload(2);
  int k0 = 0;
  for (; k0 + 6 <= Kp; k0 += 6) {
    // Each third of the body requests the operands of k-step +4 and runs the MFMAs of the current
    // one.  The sched_group_barrier sequence spreads the non-MFMA instructions (2 vector-memory
    // reads, 2 LDS reads, the relu select and the pointer bumps) BETWEEN the MFMAs: issued as a block
    // they do not overlap the matrix pipe at one wave per SIMD (measured 84 vs 64 cycles per MFMA).
#define GMPC_INTERLEAVE()                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < NTT; ++i_) {                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */            \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  /* <=1 VMEM read */     \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  /* <=1 LDS read */      \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  /* <=2 VALU */          \
      __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);  /* <=2 SALU */          \
    }
    bp += 128; ld(b2); a2 = aload(k0 + 4);
    mf(a0, k0, b0);
    GMPC_INTERLEAVE()
    bp += 128; ld(b0); a0 = aload(k0 + 6);
    mf(a1, k0 + 2, b1);
    GMPC_INTERLEAVE()
    bp += 128; ld(b1); a1 = aload(k0 + 8);
    mf(a2, k0 + 4, b2);
    GMPC_INTERLEAVE()
#undef GMPC_INTERLEAVE
  }
  if (k0 < Kp) {
    mf(a0, k0, b0);
    if (k0 + 2 < Kp) mf(a1, k0 + 2, b1);
  }
}

// Single-column-tile GEMM (the input layer: one MFMA per k-step, so operand latency is exposed):
// six register sets, operands requested five k-steps (320 MFMA cycles) ahead.  Runs whole groups of
// six k-steps: k-steps past Kp must contribute zero (afin returns 0 there; B rows are zero-padded).
template <typename AL, typename AFIN, typename BF>
__device__ __forceinline__ void gemm_tile_1(int Kp, AL aload, AFIN afin, BF bfn, f32x16& acc) {
  ARaw a[6];
  float b[6];
#pragma unroll
  for (int u = 0; u < 5; ++u) { a[u] = aload(2 * u); b[u] = bfn(2 * u); }
  for (int k0 = 0; k0 < Kp; k0 += 12) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      a[(u + 5) % 6] = aload(k0 + 2 * (u + 5));
      b[(u + 5) % 6] = bfn(k0 + 2 * (u + 5));
      __builtin_amdgcn_sched_barrier(0);   // keep the requests five k-steps ahead of their use
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afin(a[u], k0 + 2 * u), b[u], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// v_mfma_f32_4x4x1_16B_f32 with a broadcast A operand (used by the register-weight trajectory kernels
// and the LSTM kernels): B = one weight per lane (64 output columns per wave), A = act[k][slot] for
// 4 "slots", broadcast to all 16 blocks with cbsz = 4 / abid = k & 15 -- one VGPR holds 16 consecutive
// k of a [k][4 slots] float array read as 64 consecutive floats.  D register i of a lane = its column
// for slot i.  Exact fp32 FMAs at 8 cycles per k and wave.
// ------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// compile-time loop: abid is an immediate of the MFMA, so k has to be a constant expression
template <int... Is, typename F>
__device__ __forceinline__ void rw_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void rw_static_for(F&& f) {
  rw_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// d += act[k][.] (x) w for one k: ar holds rows 16 r .. 16 r + 15 of the activations
template <int K>
__device__ __forceinline__ void rw_mfma(f32x4_t& d, float ar, float w) {
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, w, d, 4, K & 15, 0);
}
