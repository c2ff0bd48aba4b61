// Register-weight MFMA form of the trajectory kernels (rollout + cost, line-search candidates) for the
// reference's default dynamics network: three hidden layers of 200 (dynamics/nn.py:27-34, yaml
// num_layers 4 / num_hidden_units 200).  gmpc_traj.hip holds the general form, the line-search
// bookkeeping kernels and the launchers that choose between the two.
//
// Reference arithmetic as in gmpc_traj.hip: dynamics/nn.py:27-34, cost/cost_model.py:20-42, cost/nn.py:23-29,
// trajax rollout / evaluate / ddp_rollout as called from policy/optimizers.py:19,26-29,55.
#include "gmpc_traj_layers.h"
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// Register-weight MFMA form of the dynamics network (k_traj<LS, H, NHL, K0Q>, 256 threads).
//
// The general form above re-reads every 200 x 200 weight matrix from L2 at each of the T steps
// (160 KB per layer per step at ~20 B/clk per CU: 8k cycles, the rollout's floor).  Here the 4 waves
// of the workgroup -- one per SIMD, 512 registers each -- keep the matrices in their registers for the
// whole horizon and multiply with v_mfma_f32_4x4x1_16B_f32, the one MFMA shape that runs at full
// rate with a 4-wide operand: the 4 trajectories (slots) of the workgroup.
//   B operand:  this lane's weight W[k][neuron 64 wave + lane]       (a register, loaded once)
//   A operand:  act[k][slot i], broadcast to all 16 blocks with cbsz = 4 / abid = k & 15: one VGPR
//               holds 16 consecutive k (lane 4 b + i = act[16 r + b][slot i]), i.e. the [k][slot] float4
//               layout of the activations read as 64 consecutive floats -- 13 LDS reads per 200 x 200
//               layer instead of one per 4 MFMAs (with one wave per SIMD nothing hides an instruction
//               between two 8-cycle MFMAs: every LDS read there cost ~10 cycles)
//   D:          register i of a lane = its neuron for slot i -> relu ballot i IS the mask word pair of
//               slot i; bias, relu and one float4 store give the next layer's activations
// ------------------------------------------------------------------------------------------------
#define GMPC_RW_THREADS 256
// weight rows of the last hidden layer kept in LDS instead of registers: two 200 x 200 layers are 400 registers,
// which with the loop's working set do not fit 512; the 128- and 64-wide instantiations (round 3) hold everything
__host__ __device__ constexpr int rw_xk(int KH, int NHL) { return KH * NHL > 320 ? 64 : 0; }
// row stride of the transposed W_L copy ([n][KHP])
__host__ __device__ constexpr int rw_khp(int KH) { return KH + 4; }
// K range of the output layer taken by wave w: [rw_kb(KH, w), rw_kb(KH, w + 1)), multiples of 4
__host__ __device__ constexpr int rw_kb(int KH, int w) {
  return KH == 200 ? (w == 0 ? 0 : w == 1 ? 52 : w == 2 ? 104 : w == 3 ? 152 : 200) : (KH / 4) * w;
}

// weight rows of a layer that live in LDS: 8 q + {2, 3, 6, 7} for q < QX
__host__ __device__ constexpr bool rw_row_in_lds(int k, int QX) { return (k >> 3) < QX && (k & 2) != 0; }
// register index of weight row k (rows in LDS are skipped)
__host__ __device__ constexpr int rw_reg_index(int k, int QX) {
  int r = 0;
  for (int j = 0; j < k; ++j) r += rw_row_in_lds(j, QX) ? 0 : 1;
  return r;
}

// epilogue of a hidden layer (the bias is already in the accumulator): relu bits -> mask words
// 2 wave, 2 wave + 1 of the 4 slots, relu, row `nn` of hout ([k] float4)
__device__ __forceinline__ void rw_hidden_epilogue(f32x4_t d, int nn, int H, float4* hout, uint32_t* mbase,
                                                   size_t mstride, unsigned wbits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool valid = nn < H;
  const bool on0 = d[0] > 0.f && valid, on1 = d[1] > 0.f && valid;
  const bool on2 = d[2] > 0.f && valid, on3 = d[3] > 0.f && valid;
  const unsigned long long w0 = __ballot(on0), w1 = __ballot(on1), w2 = __ballot(on2), w3 = __ballot(on3);
  if (mbase != nullptr) {
    const unsigned long long word = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
    if (lane < 4 && ((wbits >> lane) & 1u)) {
      uint32_t* mp = mbase + (size_t)lane * mstride + 2 * wave;
      mp[0] = (uint32_t)word;
      mp[1] = (uint32_t)(word >> 32);
    }
  }
  if (valid) hout[nn] = make_float4(on0 ? d[0] : 0.f, on1 ? d[1] : 0.f, on2 ? d[2] : 0.f, on3 ? d[3] : 0.f);
}

// one H x H hidden layer: hin ([k] float4 of the 4 slots) -> accumulator of this lane's neuron.
// XK weight rows live in LDS (two full layers, 400 registers, plus the loop's working set do not fit
// 512 registers): rows 8 q + {2, 3, 6, 7} for q < XK / 4, one float4 of wx ([XK / 4][256 threads]) each,
// interleaved with register rows so that the reads have slack.
template <int H, int XK>
__device__ __forceinline__ f32x4_t rw_layer(const float* wr, const float4* hin, const float4* wx, float bias) {
  static_assert(H % 8 == 0 && XK % 4 == 0 && 2 * XK <= H, "H");
  const int lane = threadIdx.x & 63;
  const float* hf = reinterpret_cast<const float*>(hin);
  constexpr int NR = (4 * H + 63) / 64;      // VGPRs of activations: 16 k each
  constexpr int QX = XK / 4;
  float ar[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) ar[r] = hf[64 * r + lane];
  constexpr int DX = 3;                       // LDS weight reads in flight: DX groups of 8 MFMAs ahead
  float4 xq[DX];
#pragma unroll
  for (int j = 0; j < DX; ++j)
    xq[j] = j < QX ? wx[j * GMPC_RW_THREADS + threadIdx.x] : make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4_t d0 = {bias, bias, bias, bias}, d1 = {0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);          // all reads above are issued before the first MFMA
  rw_static_for<H / 8>([&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    const float4 x = xq[q % DX];
    if (q < QX && q + DX < QX) xq[q % DX] = wx[(q + DX) * GMPC_RW_THREADS + threadIdx.x];
    rw_static_for<8>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      constexpr int k = 8 * q + e;
      float w;
      if constexpr (rw_row_in_lds(k, QX)) w = e == 2 ? x.x : e == 3 ? x.y : e == 6 ? x.z : x.w;
      else { constexpr int ri = rw_reg_index(k, QX); w = wr[ri]; }
      if constexpr (e & 1) rw_mfma<k>(d1, ar[k >> 4], w);
      else rw_mfma<k>(d0, ar[k >> 4], w);
    });
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // <= 1 LDS read
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA
  });
  return d0 + d1;
}

// output-layer partial of wave W: k in [K0, K1), B operand from the transposed LDS copy of W_L
template <int K0, int K1, int KHP>
__device__ __forceinline__ f32x4_t rw_out_part(const float4* hin, const float* wlT, int n) {
  static_assert(K0 % 4 == 0 && K1 % 4 == 0 && K1 > K0, "float4 reads of the weight row");
  const int lane = threadIdx.x & 63;
  const float* hf = reinterpret_cast<const float*>(hin);
  constexpr int R0 = K0 >> 4, R1 = (K1 - 1) >> 4;
  float ar[R1 - R0 + 1];
#pragma unroll
  for (int r = R0; r <= R1; ++r) ar[r - R0] = hf[64 * r + lane];
  const float4* wrow = reinterpret_cast<const float4*>(wlT + (size_t)(lane < n ? lane : n - 1) * KHP + K0);
  float4 wv[(K1 - K0) / 4];
#pragma unroll
  for (int q = 0; q < (K1 - K0) / 4; ++q) wv[q] = wrow[q];
  f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
  // all operand reads go out before the first MFMA (left to itself hipcc issues one read per 4 MFMAs and
  // waits for each: 13 LDS round trips, 1.3 k cycles of the step)
  __builtin_amdgcn_sched_barrier(0);
  rw_static_for<(K1 - K0) / 4>([&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    constexpr int k = K0 + 4 * q;
    rw_mfma<k + 0>(d0, ar[((k + 0) >> 4) - R0], wv[q].x);
    rw_mfma<k + 1>(d1, ar[((k + 1) >> 4) - R0], wv[q].y);
    rw_mfma<k + 2>(d0, ar[((k + 2) >> 4) - R0], wv[q].z);
    rw_mfma<k + 3>(d1, ar[((k + 3) >> 4) - R0], wv[q].w);
  });
  return d0 + d1;
}

// KH: hidden width, NHL: number of KH x KH hidden layers, K0Q: float4 groups of the layer-0 input
// (n + m <= 4 K0Q), whose weights stay in registers as well.  LS: line-search candidates (slot = one
// (trajectory, step size) item of the round's work list) instead of plain rollouts.
template <bool LS, int KH, int NHL, int K0Q>
__global__ __launch_bounds__(GMPC_RW_THREADS, 1) void k_traj_rw(TrajArgs a) {
  static_assert(KH % 16 == 0 || KH == 200, "output-layer K split in multiples of 4");
  constexpr int XK = rw_xk(KH, NHL), KHP = rw_khp(KH);
  // dynamic LDS: actA | actB (aw float4 each) | part (pw) | ksp (256, unused here) | xcur (32 rows: x, then
  // u, then zeros -- the layer-0 input) | transposed W_L [n][KHP] | LDS weight rows [XK / 4][256] float4
  extern __shared__ __attribute__((aligned(16))) char smem_traj[];
  float4* const actA = reinterpret_cast<float4*>(smem_traj);
  float4* const actB = actA + a.aw;
  float4* const part = actB + a.aw;
  float4* const ksp = part + a.pw;
  float4* const xcur = ksp + GMPC_THREADS;
  float* const wl_s = reinterpret_cast<float*>(xcur + 32);
  float4* const wx_s = reinterpret_cast<float4*>(wl_s + a.swl);
  __shared__ float s_alpha[GMPC_TB];
  __shared__ int s_bi[GMPC_TB], s_in[GMPC_TB];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = a.n, m = a.m, T = a.T;
  // slot c of this block: plain rollout -> trajectory b0 + c; line search -> item b0 + c of the work list
  const int b0 = blockIdx.x * GMPC_TB;
  if (LS) {
    const int cnt = *a.nitems;
    if (b0 >= cnt || (a.ls_split > 0 && cnt >= a.ls_split)) return;   // (long work lists: k_ls16)
    if (tid < GMPC_TB) {
      const int it = min(b0 + tid, cnt - 1);
      s_bi[tid] = a.item_b[it];
      s_in[tid] = (b0 + tid) < cnt;
      float al = a.alpha_0;
      for (int k = a.item_k[it]; k > 0; --k) al *= 0.5f;
      s_alpha[tid] = al;
    }
  } else if (tid < GMPC_TB) {
    s_bi[tid] = min(b0 + tid, a.B - 1);
    s_in[tid] = (b0 + tid) < a.B;
  }
  __syncthreads();
  // (component c of an LDS float4 is addressed as a float when c is a run-time value, see gmpc_traj.hip)
  float* const xf = reinterpret_cast<float*>(xcur);
  float* const aAf = reinterpret_cast<float*>(actA);
  const float* const pf = reinterpret_cast<const float*>(part);
  auto BI = [&](int c) -> int { return s_bi[c]; };
  auto INB = [&](int c) -> bool { return s_in[c] != 0; };
  auto CI = [&](int c) -> size_t { return (size_t)(b0 + c); };
  const int Lh = a.dyn.L - 1;
  const size_t mstride = (size_t)T * Lh * GMPC_MW;   // mask words per trajectory
  // transposed copy of W_L, [n][KHP]: lane `no` of the output layer reads its weight row 4 k at a time
  for (int e = tid; e < n * KHP; e += blockDim.x) {
    const int no = e / KHP, k = e - no * KHP;
    wl_s[e] = k < KH ? a.dyn.W[Lh][(size_t)k * n + no] : 0.f;
  }
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  // ---- weights and biases in registers for the whole horizon
  const int nnA = 64 * wave + lane;                      // this lane's neuron
  float wr[NHL][KH];
  float w0r[4 * K0Q];
  float rbias[NHL + 1];
  {
#pragma unroll
    for (int k = 0; k < 4 * K0Q; ++k)
      w0r[k] = (k < n + m && nnA < KH) ? a.dyn.W[0][(size_t)k * KH + nnA] : 0.f;
    rw_static_for<NHL>([&](auto hc) __attribute__((always_inline)) {
      constexpr int hl = decltype(hc)::value;
      const float* Wl = a.dyn.W[hl + 1];
      // rows 8 q + {2, 3, 6, 7} (q < XK / 4) of the last layer -> LDS, the others -> registers in row order
      constexpr int qx = hl == NHL - 1 ? XK / 4 : 0;
      rw_static_for<KH>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (!rw_row_in_lds(k, qx)) {
          constexpr int ri = rw_reg_index(k, qx);
          wr[hl][ri] = nnA < KH ? Wl[(size_t)k * KH + nnA] : 0.f;
        }
      });
      if constexpr (hl == NHL - 1) {
#pragma unroll 4
        for (int q = 0; q < XK / 4; ++q) {
          const float* wq = Wl + (size_t)(8 * q) * KH + nnA;
          wx_s[q * GMPC_RW_THREADS + tid] = nnA < KH ? make_float4(wq[2 * KH], wq[3 * KH], wq[6 * KH], wq[7 * KH])
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    });
#pragma unroll
    for (int l = 0; l <= NHL; ++l) rbias[l] = nnA < KH ? a.dyn.b[l][nnA] : 0.f;
    for (int e = tid; e < 32; e += blockDim.x) xcur[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  }
  // output-layer bias of the state coordinate thread tid reduces (tid < 4 n)
  const float rbl = tid < 4 * n ? a.dyn.b[Lh][tid >> 2] : 0.f;
  auto put_u = [&](int c, int j, float u) { xf[(n + j) * 4 + c] = u; };   // rows n .. n + m - 1 of the input

  // bit c set: slot c writes its outputs
  unsigned wbits = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) wbits |= (INB(c) ? 1u : 0u) << c;
  // ---- initial state
  for (int i = tid; i < n; i += blockDim.x) {
    const float* xs = LS ? a.X : a.x0;
    const size_t st = LS ? (size_t)(T + 1) * n : (size_t)n;
    float4 v = make_float4(xs[BI(0) * st + i], xs[BI(1) * st + i], xs[BI(2) * st + i], xs[BI(3) * st + i]);
    xcur[i] = v;
    if (!LS) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (INB(c)) a.X[(size_t)BI(c) * (T + 1) * n + i] = f4get(v, c);
    }
  }
  float objacc = 0.f;  // lane 0 of wave c accumulates trajectory c
  __syncthreads();
#ifdef GMPC_TRAJ_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
#define TS_(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define TS_(i)
#endif

  // The operands of the step's controls are loaded one step ahead (every global round trip inside the step
  // is ~1k cycles of a ~8k-cycle step).  Plain rollout: thread (c, j) holds u_t[j] of slot c.  Line search:
  // 8 lanes share one (slot, control) pair of u = U + alpha k + K (x - X_nominal) and hold its gain row and
  // nominal state, 4 elements each (n <= 32).
  const bool rw_pf = GMPC_TB * m <= (int)(blockDim.x >> 3);   // all pairs in one pass
  const int rw_pp = tid >> 3, rw_l8 = tid & 7;
  const int rw_c = rw_pp / m, rw_j = rw_pp - rw_c * m;
  const bool rw_on = rw_pf && rw_pp < GMPC_TB * m;
  float pfK[4] = {0.f, 0.f, 0.f, 0.f}, pfX[4] = {0.f, 0.f, 0.f, 0.f}, pfk = 0.f, pfU = 0.f;
  auto rw_prefetch = [&](int t) {
    if (!rw_on) return;
    const int bc = BI(rw_c);
    const size_t ub = ((size_t)bc * T + t) * m + rw_j;
    if (LS) {
      const float* Kr = a.Kg + ub * n;
      const float* Xo = a.X + ((size_t)bc * (T + 1) + t) * n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = rw_l8 + 8 * e;
        pfK[e] = i < n ? Kr[i] : 0.f;
        pfX[e] = i < n ? Xo[i] : 0.f;
      }
      pfk = a.kg[ub];
      pfU = a.Uio[ub];
    } else if (rw_l8 == 0) {
      pfU = a.U[ub];
    }
  };
  rw_prefetch(0);
  for (int t = 0; t < T; ++t) {
    // ---- controls (the stage costs are evaluated after the horizon, see below)
    if (rw_on) {
      float u = pfU;
      if (LS) {
        float du = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = rw_l8 + 8 * e;
          if (i < n) du = fmaf(pfK[e], xf[i * 4 + rw_c] - pfX[e], du);
        }
        du += __shfl_xor(du, 4);
        du += __shfl_xor(du, 2);
        du += __shfl_xor(du, 1);
        u = pfU + fmaf(s_alpha[rw_c], pfk, du);
      }
      if (rw_l8 == 0) {
        if (LS && ((wbits >> rw_c) & 1u)) a.Uc[(CI(rw_c) * T + t) * m + rw_j] = u;
        put_u(rw_c, rw_j, u);
      }
    } else if (!rw_pf) {
      // wide control vectors: the general loops, operands loaded in place
      if (LS) {
        const int l16 = tid & 15;
        for (int p = tid >> 4; p < GMPC_TB * m; p += blockDim.x >> 4) {
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
            put_u(c, j, u);
          }
        }
      } else {
        for (int p = tid; p < GMPC_TB * m; p += blockDim.x) {
          const int c = p / m, j = p - c * m;
          put_u(c, j, a.U[((size_t)BI(c) * T + t) * m + j]);
        }
      }
    }
    if (t + 1 < T) rw_prefetch(t + 1);
    __syncthreads();
    TS_(0)
    // ---- the dynamics network on the matrix pipe (see rw_layer)
    uint32_t* mb = (LS ? a.maskc : a.masks) + (size_t)b0 * mstride + (size_t)t * Lh * GMPC_MW;
    {
      // layer 0: input rows (x ; u ; 0) of xcur, weights from registers
      float ar[(K0Q + 3) / 4];
#pragma unroll
      for (int r = 0; r < (K0Q + 3) / 4; ++r) ar[r] = xf[64 * r + lane];
      f32x4_t d0 = {rbias[0], rbias[0], rbias[0], rbias[0]}, d1 = {0.f, 0.f, 0.f, 0.f};
      rw_static_for<2 * K0Q>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = 2 * decltype(kc)::value;
        rw_mfma<k>(d0, ar[k >> 4], w0r[k]);
        rw_mfma<k + 1>(d1, ar[(k + 1) >> 4], w0r[k + 1]);
      });
      rw_hidden_epilogue(d0 + d1, nnA, KH, actA, mb, mstride, wbits);
    }
    __syncthreads();
    TS_(2)
    float4* hin = actA;
    float4* hout = actB;
    rw_static_for<NHL>([&](auto hc) __attribute__((always_inline)) {
      constexpr int hl = decltype(hc)::value;
      const f32x4_t d = rw_layer<KH, (hl == NHL - 1 ? XK : 0)>(wr[hl], hin, wx_s, rbias[hl + 1]);
      rw_hidden_epilogue(d, nnA, KH, hout, mb + (hl + 1) * GMPC_MW, mstride, wbits);
      __syncthreads();
      TS_(3 + hl)
      float4* tmp = hin; hin = hout; hout = tmp;
    });
    {
      // output layer: the K range is split over the 4 waves; lane = output coordinate
      f32x4_t d;
      if (wave == 0) d = rw_out_part<rw_kb(KH, 0), rw_kb(KH, 1), KHP>(hin, wl_s, n);
      else if (wave == 1) d = rw_out_part<rw_kb(KH, 1), rw_kb(KH, 2), KHP>(hin, wl_s, n);
      else if (wave == 2) d = rw_out_part<rw_kb(KH, 2), rw_kb(KH, 3), KHP>(hin, wl_s, n);
      else d = rw_out_part<rw_kb(KH, 3), rw_kb(KH, 4), KHP>(hin, wl_s, n);
      if (lane < 32) part[wave * 32 + lane] = make_float4(d[0], d[1], d[2], d[3]);
    }
    __syncthreads();
    if (tid < 4 * n) {
      const int no = tid >> 2, c = tid & 3;
      float sum = pf[no * 4 + c];
#pragma unroll
      for (int w = 1; w < GMPC_RW_THREADS / 64; ++w) sum += pf[(w * 32 + no) * 4 + c];
      const float v = (sum + rbl) + xf[no * 4 + c];
      xf[no * 4 + c] = v;
      if ((wbits >> c) & 1u) {
        float* Xo = LS ? a.Xc : a.X;
        Xo[((LS ? CI(c) : (size_t)BI(c)) * (T + 1) + t + 1) * n + no] = v;
      }
    }
    // the line search's controls of step t + 1 read x_{t+1}; the plain rollout's do not, and its
    // next barrier (after the controls) orders the writes above before layer 0 reads them
    if (LS) __syncthreads();
    TS_(6)
  }
#ifdef GMPC_TRAJ_STAMPS
  if (blockIdx.x == 0 && tid == 0)
    printf("k_traj_rw cycles per step: controls %llu L0 %llu L1 %llu L2 %llu out %llu\n", st_[0] / T, st_[2] / T,
           st_[3] / T, st_[4] / T, st_[6] / T);
#endif
  {
    // ---- stage costs of the register-weight form, after the horizon: wave c re-reads the states and
    // controls of slot c (written by this workgroup, ordered by the barrier), one step per lane, and
    // lane 0 adds the costs in step order -- the same sum the in-loop form builds, without a
    // reduction + two square roots on the critical path of every step
    __syncthreads();
    float* cs = aAf + wave * 64;                 // 64 costs of this wave's slot
    const int c = wave;
    const int bc = BI(c);
    const float al = GMPC_ALPHA;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      float cst = 0.f;
      if (t < T && INB(c)) {
        const float* xr = (LS && t > 0) ? a.Xc + (CI(c) * (T + 1) + t) * n
                                         : a.X + ((size_t)bc * (T + 1) + t) * n;
        const float* ur = LS ? a.Uc + (CI(c) * T + t) * m : a.U + ((size_t)bc * T + t) * m;
        const float* g = a.goal + ((size_t)bc * (T + 1) + t) * n;
        float dd = 0.f, uu = 0.f;
        for (int i = 0; i < n; ++i) {
          const float d = xr[i] - g[i];
          dd = fmaf(d, d, dd);
        }
        for (int j = 0; j < m; ++j) uu = fmaf(ur[j], ur[j], uu);
        cst = w0 * (sqrtf(uu + al * al) - al) + w1 * (sqrtf(dd + al * al) - al);
        if (!LS && a.costs) a.costs[(size_t)bc * (T + 1) + t] = cst;
      }
      cs[lane] = cst;
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the wave's LDS writes have landed
      __builtin_amdgcn_wave_barrier();
      const int cnt = min(64, T - t0);
      for (int e = 0; e < cnt; ++e) objacc += cs[e];
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  // ---- terminal cost w2 * |cost_mlp(x_T)|^2
  {
    float4* in = xcur;
    float4* out = actA;
    const int Lc = a.cost.L - 1;
    for (int l = 0; l < Lc; ++l) {
      hidden_layer(a.cost.W[l], a.cost.b[l], a.cost.dims[l], a.cost.dims[l + 1], in, out, nullptr, 0,
                   0u, nullptr);
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

// ---- host side --------------------------------------------------------------------------------------
// the shapes the register-weight form is instantiated for: three equal hidden layers of width 200 (the reference's
// default), 128 or 64, state and control within one 32-row input block
static int rw_width(const TrajArgs& a) {
  const int Lh = a.dyn.L - 1;
  if (Lh != 3 || a.n > 32 || a.m > 32 || a.n + a.m > 32) return 0;
  const int H = a.dyn.dims[1];
  if (H != 200 && H != 128 && H != 64) return 0;
  for (int l = 1; l <= Lh; ++l)
    if (a.dyn.dims[l] != H) return 0;
  return H;
}
bool gmpc_traj_rw_shape(const TrajArgs& a) {
  static const bool off = getenv("GMPC_TRAJ") != nullptr && strcmp(getenv("GMPC_TRAJ"), "valu") == 0;
  return !off && rw_width(a) != 0;
}

// LDS of one workgroup; sets the sizing fields of `a` (aw is set by the caller: widest layer of both networks)
size_t gmpc_traj_rw_lds(TrajArgs& a) {
  const int H = rw_width(a);
  a.pw = GMPC_RW_THREADS;
  a.sw0 = 0;                                             // W_0 lives in registers
  a.swl = a.n * rw_khp(H);                               // transposed W_L
  return ((size_t)2 * a.aw + a.pw + GMPC_THREADS + 32) * sizeof(float4) +
         ((size_t)a.swl + (size_t)rw_xk(H, 2) * GMPC_RW_THREADS) * sizeof(float);
}

// one workgroup per 4 trajectories (ls = false) / work-list items (ls = true: `grid` covers the largest
// possible work list, the kernel reads the actual count)
template <int KH, int K0Q>
static void launch_rw(const TrajArgs& a, bool ls, int grid, size_t lds, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    // one workgroup per CU (the registers hold the weights): up to 160 KB of LDS
    const void* ks[] = {reinterpret_cast<const void*>(&k_traj_rw<false, KH, 2, K0Q>),
                        reinterpret_cast<const void*>(&k_traj_rw<true, KH, 2, K0Q>)};
    for (const void* k : ks) {
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      (void)hipGetLastError();
    }
    attr = true;
  }
  if (!ls) hipLaunchKernelGGL((k_traj_rw<false, KH, 2, K0Q>), dim3(grid), dim3(GMPC_RW_THREADS), lds, s, a);
  else hipLaunchKernelGGL((k_traj_rw<true, KH, 2, K0Q>), dim3(grid), dim3(GMPC_RW_THREADS), lds, s, a);
}

void gmpc_launch_traj_rw(const TrajArgs& a, bool ls, int grid, size_t lds, hipStream_t s) {
  const bool small = a.n + a.m <= 24;
  switch (rw_width(a)) {
    case 200:
      if (small) launch_rw<200, 6>(a, ls, grid, lds, s);
      else launch_rw<200, 8>(a, ls, grid, lds, s);
      break;
    case 128: launch_rw<128, 8>(a, ls, grid, lds, s); break;       // (one layer-0 form: 32 input rows)
    case 64: launch_rw<64, 8>(a, ls, grid, lds, s); break;
    default: break;
  }
}
