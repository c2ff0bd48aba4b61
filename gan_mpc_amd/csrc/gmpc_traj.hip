// Sequential-in-time trajectory kernels: rollout + cost, and the DDP line search.
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
#include "gmpc_device.h"
#include <cstdlib>
#include <cstring>
#include <utility>

#define GMPC_TRAJ_THREADS_ 512   // workgroup size of k_traj (see GMPC_TRAJ_THREADS)


// One hidden layer for the 4 trajectories of the block: z = act_in . W + b; mask bits; relu.
// mbase points at mask word 0 of (trajectory 0, this step, this layer); trajectory c sits
// c*mstride words further; bit c of wbits enables the mask store of trajectory c.
__device__ __forceinline__ void hidden_layer(const float* W, const float* bias, int K, int N,
                                             const float4* actIn, float4* actOut, uint32_t* mbase,
                                             size_t mstride, unsigned wbits, float4* ksplit = nullptr) {
  // Threads beyond the first 256 (k_traj runs 512) take the second half of the K range of the same
  // neuron j: two waves per SIMD share the issue slots, the halves meet in LDS (`ksplit`).
  const int j = threadIdx.x & (GMPC_THREADS - 1);
  const int ks = threadIdx.x >> 8;                 // 0 or 1 (wave-uniform)
  const bool split = ksplit != nullptr;
  const bool valid = j < N;
  const int Kh = split ? ((K + 1) >> 1) : K;
  const int k0 = ks * Kh;
  const int kn = split ? (ks == 0 ? Kh : K - Kh) : K;
  const float bj = (valid && ks == 0) ? bias[j] : 0.f;
  float4 acc[1] = {make_float4(bj, bj, bj, bj)};
  if (ks == 0 || split) dense_rows<1>(W + (size_t)k0 * N, kn, N, j, actIn + k0, acc);
  if (split) {
    if (ks == 1 && valid) ksplit[j] = acc[0];
    __syncthreads();
    if (ks == 1) return;
    if (valid) {
      const float4 o = ksplit[j];
      acc[0].x += o.x; acc[0].y += o.y; acc[0].z += o.z; acc[0].w += o.w;
    }
  } else if (ks != 0) {
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long b0 = __ballot(valid && acc[0].x > 0.f);
  const unsigned long long b1 = __ballot(valid && acc[0].y > 0.f);
  const unsigned long long b2 = __ballot(valid && acc[0].z > 0.f);
  const unsigned long long b3 = __ballot(valid && acc[0].w > 0.f);
  if (lane == 0 && mbase != nullptr) {
    if (wbits & 1u) { mbase[2 * wave] = (uint32_t)b0; mbase[2 * wave + 1] = (uint32_t)(b0 >> 32); }
    if (wbits & 2u) { mbase[mstride + 2 * wave] = (uint32_t)b1; mbase[mstride + 2 * wave + 1] = (uint32_t)(b1 >> 32); }
    if (wbits & 4u) { mbase[2 * mstride + 2 * wave] = (uint32_t)b2; mbase[2 * mstride + 2 * wave + 1] = (uint32_t)(b2 >> 32); }
    if (wbits & 8u) { mbase[3 * mstride + 2 * wave] = (uint32_t)b3; mbase[3 * mstride + 2 * wave + 1] = (uint32_t)(b3 >> 32); }
  }
  if (valid)
    actOut[j] = make_float4(fmaxf(acc[0].x, 0.f), fmaxf(acc[0].y, 0.f), fmaxf(acc[0].z, 0.f),
                            fmaxf(acc[0].w, 0.f));
}

// Output layer of the trajectory kernels for n <= 32: out[j] = sum_k W[k][j] act[k].  512 threads =
// 32 output slots x 16 K-slices; the two slices of a wave meet by a lane-half exchange, the 8 wave
// partials through LDS (`part`, 8 x 32 float4), summed in wave order by the caller-visible result
// part[j].  (dense_small's generic form put 30 thread groups' partials through LDS and summed them
// one after the other: 6.2k of the 27.6k cycles of a rollout step.)
__device__ __forceinline__ void out_layer32(const float* W, int K, int n, const float4* act, float4* part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = tid & 31, kq = tid >> 5;                 // 16 K-slices
  const int Kq = (K + 15) >> 4;
  const int k0 = kq * Kq, k1 = min(K, k0 + Kq);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n) {
    const float* wp = W + j;
    for (int k = k0; k < k1; ++k) fma4(acc, wp[(size_t)k * n], act[k]);
  }
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
  acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
  if (lane < 32) part[32 + wave * 32 + j] = acc;         // slots 32.. : wave partials
  __syncthreads();
  if (tid < n) {
    float4 s = part[32 + tid];
#pragma unroll
    for (int w = 1; w < GMPC_TRAJ_THREADS_ / 64; ++w) {
      const float4 p = part[32 + w * 32 + tid];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    part[tid] = s;
  }
  __syncthreads();
}

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
#define GMPC_RW_XK 64     // weight rows of the last hidden layer kept in LDS instead of registers
#define GMPC_RW_KHP 204   // row stride of the transposed W_L copy ([n][KHP], KHP = H + 4)

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
template <int K0, int K1>
__device__ __forceinline__ f32x4_t rw_out_part(const float4* hin, const float* wlT, int n) {
  static_assert(K0 % 4 == 0 && K1 % 4 == 0, "float4 reads of the weight row");
  const int lane = threadIdx.x & 63;
  const float* hf = reinterpret_cast<const float*>(hin);
  constexpr int R0 = K0 >> 4, R1 = (K1 - 1) >> 4;
  float ar[R1 - R0 + 1];
#pragma unroll
  for (int r = R0; r <= R1; ++r) ar[r - R0] = hf[64 * r + lane];
  const float4* wrow = reinterpret_cast<const float4*>(wlT + (size_t)(lane < n ? lane : n - 1) * GMPC_RW_KHP + K0);
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

#ifndef GMPC_TRAJ_MINW
#define GMPC_TRAJ_MINW 4
#endif
#define GMPC_TRAJ_THREADS 512   // k_traj: 8 waves, the upper 4 take the second half of every K range

// KH > 0 selects the register-weight MFMA form of the dynamics network (hidden width KH, NHL hidden
// KH x KH layers, 256 threads; see rw_layer above); KH = 0 is the general VALU form (512 threads).
template <bool LS, int KH = 0, int NHL = 0, int K0Q = 0>
__global__ __launch_bounds__(KH ? GMPC_RW_THREADS : GMPC_TRAJ_THREADS, KH ? 1 : GMPC_TRAJ_MINW) void k_traj(TrajArgs a) {
  // dynamic LDS: actA | actB (aw float4 each: max(n+m, widest layer)) | part (pw) | ksp (256) | xcur (n)
  extern __shared__ __attribute__((aligned(16))) char smem_traj[];
  float4* const actA = reinterpret_cast<float4*>(smem_traj);
  float4* const actB = actA + a.aw;
  float4* const part = actB + a.aw;
  float4* const ksp = part + a.pw;
  float4* const xcur = ksp + GMPC_THREADS;
  // the two small weight matrices live in LDS for the whole horizon when they fit (launcher decides)
  // (register-weight form: xcur has 32 rows -- x, then u, then zeros: the layer-0 input)
  float* const w0_s = reinterpret_cast<float*>(xcur + (KH > 0 ? 32 : a.n));
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
  // value: hipcc (ROCm 7.2) lowers `c == 0 ? v.x : ...` on an LDS reference with a lane-varying c
  // into a branch tree that gives lanes with c == 3 the .z address (seen in the ISA and on the GPU).
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
  {
    // (the register-weight form stages W_0 with its row count rounded up to 4: the extra rows are zero)
    const int w0n = a.dyn.dims[0] * a.dyn.dims[1];
    for (int e = tid; e < a.sw0; e += blockDim.x) w0_s[e] = e < w0n ? a.dyn.W[0][e] : 0.f;
  }
  if (KH == 0) {
    for (int e = tid; e < a.swl; e += blockDim.x) wl_s[e] = a.dyn.W[Lh][e];
  } else {
    // transposed copy [n][KHP]: lane `no` of the output layer reads its weight row 4 k at a time
    for (int e = tid; e < n * GMPC_RW_KHP; e += blockDim.x) {
      const int no = e / GMPC_RW_KHP, k = e - no * GMPC_RW_KHP;
      wl_s[e] = k < KH ? a.dyn.W[Lh][(size_t)k * n + no] : 0.f;
    }
  }
  const float* const W0 = a.sw0 ? w0_s : a.dyn.W[0];
  const float* const WL = a.swl ? wl_s : a.dyn.W[Lh];
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  // ---- register-weight form (KH > 0): weights and biases in registers for the whole horizon; the
  // activations use actA / actB ([k] float4), the output-layer partials `part`
  float4* const wx_s = reinterpret_cast<float4*>(wl_s + a.swl);   // [XK / 4][256]: LDS weight rows
  const int nnA = 64 * wave + lane;                      // this lane's neuron
  float wr[NHL > 0 ? NHL : 1][KH > 0 ? KH : 1];
  float w0r[K0Q > 0 ? 4 * K0Q : 1];
  float rbias[NHL + 1];
  if constexpr (KH > 0) {
#pragma unroll
    for (int k = 0; k < 4 * K0Q; ++k)
      w0r[k] = (k < n + m && nnA < KH) ? a.dyn.W[0][(size_t)k * KH + nnA] : 0.f;
    rw_static_for<NHL>([&](auto hc) __attribute__((always_inline)) {
      constexpr int hl = decltype(hc)::value;
      const float* Wl = a.dyn.W[hl + 1];
      // rows 8 q + {2, 3, 6, 7} (q < XK / 4) of the last layer -> LDS, the others -> registers in row order
      constexpr int qx = hl == NHL - 1 ? GMPC_RW_XK / 4 : 0;
      rw_static_for<KH>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (!rw_row_in_lds(k, qx)) {
          constexpr int ri = rw_reg_index(k, qx);
          wr[hl][ri] = nnA < KH ? Wl[(size_t)k * KH + nnA] : 0.f;
        }
      });
      if constexpr (hl == NHL - 1) {
#pragma unroll 4
        for (int q = 0; q < GMPC_RW_XK / 4; ++q) {
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
  const float rbl = (KH > 0 && tid < 4 * n) ? a.dyn.b[Lh][tid >> 2] : 0.f;
  const float* const xs_ = KH > 0 ? xf : aAf;            // x_t and u_t of the four slots, [i][slot]
  const float* const us_ = KH > 0 ? xf + 4 * n : aAf + 4 * n;
  auto put_u = [&](int c, int j, float u) {
    if (KH > 0) {
      xf[(n + j) * 4 + c] = u;           // rows n .. n + m - 1 of the layer-0 input
    } else {
      aAf[(n + j) * 4 + c] = u;
    }
  };

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
    // register-weight form: the operands of the step's controls are loaded one step ahead (every global
    // round trip inside the step is ~1k cycles of a ~9k-cycle step).  Plain rollout: thread (c, j) holds
    // u_t[j] of slot c.  Line search: 8 lanes share one (slot, control) pair of u = U + alpha k +
    // K (x - X_nominal) and hold its gain row and nominal state, 4 elements each (n <= 32).
    const bool rw_pf = KH > 0 && GMPC_TB * m <= (int)(blockDim.x >> 3);   // all pairs in one pass
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
    if (KH > 0) rw_prefetch(0);
    for (int t = 0; t < T; ++t) {
      if constexpr (KH > 0) {
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
      }
      // line search: stop as soon as none of the four candidates can still be accepted (flags of the
      // previous step's cost evaluation; the barriers of that step ordered them)
      if (KH == 0 && LS && t > 0 && (s_live[0] | s_live[1] | s_live[2] | s_live[3]) == 0) { aborted = true; break; }
      // ---- controls and layer-0 input
      if (KH == 0)
        for (int i = tid; i < n; i += blockDim.x) actA[i] = xcur[i];
      if (KH > 0) {
        // (done above)
      } else if (LS) {
        // u = U + alpha k + K (x - X_nominal): 16 lanes share one (slot, control) inner product, so
        // the n gain / state loads of a control are issued together instead of one after another
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
      } else if (tid < GMPC_TB * m) {
        const int c = tid / m, j = tid % m;
        put_u(c, j, a.U[((size_t)BI(c) * T + t) * m + j]);
      }
      if (KH == 0) __syncthreads();
      TS_(0)
      // ---- stage cost of (x_t, u_t): wave c (< 4) handles trajectory c
      if (KH == 0 && wave < GMPC_TB) {
        const int c = wave;
        float dd = 0.f, uu = 0.f;
        const int bc = BI(c);
        const float* g = a.goal + ((size_t)bc * (T + 1) + t) * n;
        if (n <= 64) {
          const float gi = gnext;
          if (lane < n) {
            gnext = g[n + lane];           // row t + 1 (exists: the goal has T + 1 rows)
            const float d = xs_[lane * 4 + c] - gi;
            dd = d * d;
          }
        } else {
          for (int i = lane; i < n; i += 64) {
            const float d = xs_[i * 4 + c] - g[i];
            dd = fmaf(d, d, dd);
          }
        }
        for (int j = lane; j < m; j += 64) {
          const float u = us_[j * 4 + c];
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
      if constexpr (KH > 0) {
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
          const f32x4_t d = rw_layer<KH, (hl == NHL - 1 ? GMPC_RW_XK : 0)>(wr[hl], hin, wx_s, rbias[hl + 1]);
          rw_hidden_epilogue(d, nnA, KH, hout, mb + (hl + 1) * GMPC_MW, mstride, wbits);
          __syncthreads();
          TS_(3 + hl)
          float4* tmp = hin; hin = hout; hout = tmp;
        });
        {
          // output layer: the K range is split over the 4 waves; lane = output coordinate
          static_assert(KH == 200, "K split of the output layer");
          f32x4_t d;
          if (wave == 0) d = rw_out_part<0, 52>(hin, wl_s, n);
          else if (wave == 1) d = rw_out_part<52, 104>(hin, wl_s, n);
          else if (wave == 2) d = rw_out_part<104, 152>(hin, wl_s, n);
          else d = rw_out_part<152, 200>(hin, wl_s, n);
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
        continue;
      }
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
    if constexpr (KH > 0) {
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
                     0u, KH > 0 ? nullptr : ksp);
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
// have stopped at -- commits it, or queues the next 4 halvings.  The first round of a trajectory
// covers the halvings up to the one its previous line search accepted (1 candidate for a
// well-conditioned problem that takes full steps, up to 8 for one that backtracks deeply), so the
// rollouts stay close to the sequential loop's count while the launches drop from up to 15
// dependent rollouts to 1-3 rounds.  Nothing is read back by the host.
// ------------------------------------------------------------------------------------------------
#define GMPC_LS_NEXT 4   // candidates queued per trajectory after a round without an accepted step

__global__ void k_ls_init(int B, const int* active, float alpha_0, float alpha_min, int k_max, int* iters,
                          int* run, int* cnt, int* kfirst, const int* prevk, float* alpha, float* U_step,
                          float* obj_step) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  cnt[b] = 0;
  if (active != nullptr && active[b] == 0) { run[b] = 0; return; }
  iters[b] += 1;
  if (alpha_0 > alpha_min) {
    int R = prevk[b] + 1;
    R = R < 1 ? 1 : R;
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
                                                   int* item_k, int* slot, int* count) {
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
  float alpha_0;
  const int* slot; int* cnt; int* kfirst; int* prevk; int* run;
  const float* objc; const float* Xc; const float* Uc; const uint32_t* maskc;
  float* X; float* U; uint32_t* masks;
  float* obj; float* obj_step; float* U_step; float* alpha;
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
    } else if (k0 + R >= a.k_max) {      // every step size down to alpha_min failed
      a.alpha[b] = halved(a.k_max);
      a.U_step[b] = 0.f;
      a.obj_step[b] = 0.f;
      a.prevk[b] = a.k_max - 1;
      a.run[b] = 0;
      a.cnt[b] = 0;
    } else {                               // queue the next GMPC_LS_NEXT halvings (k_ls_place)
      const int left = a.k_max - (k0 + R);
      a.cnt[b] = left < GMPC_LS_NEXT ? left : GMPC_LS_NEXT;
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
// register-weight form: three equal hidden layers of width 200, state and control within one MFMA block row
static bool traj_rw_shape(const TrajArgs& a) {
  static const bool off = getenv("GMPC_TRAJ") != nullptr && strcmp(getenv("GMPC_TRAJ"), "valu") == 0;
  const int Lh = a.dyn.L - 1;
  if (off || Lh != 3 || a.n > 32 || a.m > 32 || a.n + a.m > 32) return false;
  for (int l = 1; l <= Lh; ++l)
    if (a.dyn.dims[l] != 200) return false;
  return true;
}
static size_t traj_lds(TrajArgs& a, bool rw = false) {
  a.aw = traj_aw(a.n, a.m, a.dyn, &a.cost);
  if (rw) {
    // one workgroup per CU (the registers hold the weights): both small matrices and the rows of the
    // MFMA form go to LDS without the 64 KB consideration below
    a.pw = GMPC_RW_THREADS;
    a.sw0 = 0;                                             // W_0 lives in registers
    a.swl = a.n * GMPC_RW_KHP;                             // transposed W_L
    return ((size_t)2 * a.aw + a.pw + GMPC_THREADS + 32) * sizeof(float4) +
           ((size_t)a.swl + (size_t)GMPC_RW_XK * GMPC_RW_THREADS) * sizeof(float);
  }
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

void gmpc_launch_rollout(const TrajArgs& a0, hipStream_t s) {
  TrajArgs a = a0;
  const bool rw = traj_rw_shape(a);
  const size_t lds = traj_lds(a, rw);
  static bool attr = false;
  if (!attr) { traj_attr(&k_traj<false>); traj_attr(&k_traj<false, 200, 2, 6>); traj_attr(&k_traj<false, 200, 2, 8>); attr = true; }
  const int grid = (a.B + GMPC_TB - 1) / GMPC_TB;
  if (rw && a.n + a.m <= 24)
    hipLaunchKernelGGL((k_traj<false, 200, 2, 6>), dim3(grid), dim3(GMPC_RW_THREADS), lds, s, a);
  else if (rw)
    hipLaunchKernelGGL((k_traj<false, 200, 2, 8>), dim3(grid), dim3(GMPC_RW_THREADS), lds, s, a);
  else
    hipLaunchKernelGGL(k_traj<false>, dim3(grid), dim3(GMPC_TRAJ_THREADS), lds, s, a);
}
int gmpc_launch_linesearch(const TrajArgs& a0, const LsWork& w, hipStream_t s) {
  TrajArgs a = a0;
  const bool rw = traj_rw_shape(a);
  const size_t lds = traj_lds(a, rw);
  static bool attr = false;
  if (!attr) { traj_attr(&k_traj<true>); traj_attr(&k_traj<true, 200, 2, 6>); traj_attr(&k_traj<true, 200, 2, 8>); attr = true; }
  // halvings allowed by trajax' loop: candidate k runs while alpha_0 / 2^k > alpha_min
  int k_max = 0;
  for (float al = a.alpha_0; al > a.alpha_min && k_max < 4096; al *= 0.5f) ++k_max;
  // worst case: a first round of one candidate, then GMPC_LS_NEXT per round
  const int rounds = k_max > 0 ? 1 + (k_max - 1 + GMPC_LS_NEXT - 1) / GMPC_LS_NEXT : 0;
  if (rounds > GMPC_LS_ROUNDS_MAX) return -1;
  hipLaunchKernelGGL(k_ls_init, dim3((a.B + 255) / 256), dim3(256), 0, s, a.B, a.active, a.alpha_0,
                     a.alpha_min, k_max, a.iters, w.run, w.cnt, w.kfirst, w.prevk, a.alpha, a.U_step,
                     a.obj_step);
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(k_ls_place, dim3(1), dim3(1024), 0, s, a.B, w.cnt, w.kfirst, w.item_b[0], w.item_k[0],
                       w.slot, w.counts + r);
    a.item_b = w.item_b[0]; a.item_k = w.item_k[0]; a.nitems = w.counts + r; a.objc = w.objc;
    const long max_items = (long)a.B * (r == 0 ? GMPC_LS_ITEMS : GMPC_LS_NEXT);
    const dim3 lsgrid((unsigned)((max_items + GMPC_TB - 1) / GMPC_TB));
    if (rw && a.n + a.m <= 24)
      hipLaunchKernelGGL((k_traj<true, 200, 2, 6>), lsgrid, dim3(GMPC_RW_THREADS), lds, s, a);
    else if (rw)
      hipLaunchKernelGGL((k_traj<true, 200, 2, 8>), lsgrid, dim3(GMPC_RW_THREADS), lds, s, a);
    else
      hipLaunchKernelGGL(k_traj<true>, lsgrid, dim3(GMPC_TRAJ_THREADS), lds, s, a);
    LsDecideArgs d;
    d.n = a.n; d.m = a.m; d.T = a.T; d.Lh = a.dyn.L - 1; d.k_max = k_max;
    d.alpha_0 = a.alpha_0;
    d.slot = w.slot; d.cnt = w.cnt; d.kfirst = w.kfirst; d.prevk = w.prevk; d.run = w.run;
    d.objc = w.objc; d.Xc = a.Xc; d.Uc = a.Uc; d.maskc = a.maskc;
    d.X = a.X; d.U = a.Uio; d.masks = a.masks;
    d.obj = a.obj; d.obj_step = a.obj_step; d.U_step = a.U_step; d.alpha = a.alpha;
    hipLaunchKernelGGL(k_ls_decide, dim3(a.B), dim3(GMPC_THREADS), 0, s, d);
  }
  return 0;
}
void gmpc_launch_masks(int B, int n, int m, int T, const MlpDesc& dyn, const float* X,
                       const float* U, uint32_t* masks, hipStream_t s) {
  const int NS = B * T;
  const int aw = traj_aw(n, m, dyn, nullptr);
  hipLaunchKernelGGL(k_masks, dim3((NS + 3) / 4), dim3(GMPC_THREADS), 2 * (size_t)aw * sizeof(float4), s,
                     NS, n, m, T, dyn, X, U, masks, aw);
}
