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
// Register-weight MFMA form of the dynamics network (k_traj<LS, H, NHL>, 256 threads).
//
// The general form above re-reads every 200 x 200 weight matrix from L2 at each of the T steps
// (160 KB per layer per step at ~20 B/clk per CU: 8k cycles, the rollout's floor).  Here the 4 waves
// of the workgroup -- one per SIMD, 512 registers each -- keep the hidden matrices in their registers
// for the whole horizon and multiply with v_mfma_f32_4x4x1_16B_f32, the one MFMA shape that runs at
// full rate with 4 columns: the 4 trajectories of the workgroup.
//   lane = (g, p): neuron group g = lane >> 2 (16 blocks of the MFMA), p = lane & 3
//   block g:    D[i][c] += W[k][neuron 64 wave + 4 g + i] * act[k][slot c]       (one k per issue)
//   A operand:  the lane's weight W[k][64 wave + lane]            (register wr[k], loaded once)
//   B operand:  act[k][slot p] from the LDS row of slot p          (float4 = 4 k per read)
//   D:          register i of lane (g, p) = neuron 64 wave + 4 g + i of slot p -> bias, relu bit, relu,
//               one float4 store into row p of the next layer's B operand
// ------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define GMPC_RW_THREADS 256
#define GMPC_RW_HP 204    // LDS row stride (floats) of a hidden activation row: 200 + 4, = 12 mod 32
#define GMPC_RW_ZP 36     // row stride of the layer-0 input rows (n + m <= 32), = 4 mod 32

__device__ __forceinline__ void rw_swap_halves(float& a, float& b) {
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  const v2u_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}

// epilogue of a hidden layer: bias, relu bits -> mask words 2 wave, 2 wave + 1 of the 4 slots, relu,
// row p of hout
__device__ __forceinline__ void rw_hidden_epilogue(f32x4_t d, const float (&bias)[4], int H, float* hout,
                                                   uint32_t* mbase, size_t mstride, unsigned wbits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 2, p = lane & 3;
  float v[4];
  unsigned long long word = 0ull;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = d[i] + bias[i];
    // ballot bit 4 g + c: neuron 4 g + i of slot c; slot p keeps every fourth bit, moved to bit 4 g + i
    const unsigned long long bal = __ballot(v[i] > 0.f);
    word |= ((bal >> p) & 0x1111111111111111ull) << i;
  }
  if (mbase != nullptr && lane < 4 && ((wbits >> lane) & 1u)) {
    uint32_t* mp = mbase + (size_t)lane * mstride + 2 * wave;
    mp[0] = (uint32_t)word;
    mp[1] = (uint32_t)(word >> 32);
  }
  const int k = 64 * wave + 4 * g;
  if (k < H)
    *reinterpret_cast<float4*>(hout + p * GMPC_RW_HP + k) =
        make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
}

// one H x H hidden layer: row p of hin -> accumulator of this lane's block.  The first H - XK weight
// rows come from registers, the last XK from the LDS copy wx ([XK / 4][256 threads] float4): two
// full layers (400 registers) plus the loop's working set do not fit 512 registers -- the compiler
// spilled ~60 weights to scratch and waited for each reload in front of its MFMA
#define GMPC_RW_XK 64
template <int H, int XK>
__device__ __forceinline__ f32x4_t rw_layer(const float (&wr)[H], const float* hin, const float4* wx) {
  static_assert(H % 8 == 0 && XK % 4 == 0, "H");
  const int lane = threadIdx.x & 63;
  const float4* brow = reinterpret_cast<const float4*>(hin + (lane & 3) * GMPC_RW_HP);
  f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f}, d3 = {0.f, 0.f, 0.f, 0.f};
  constexpr int QR = (H - XK) / 4;     // float4 steps with register weights
  float4 b0 = brow[0], b1 = brow[1], b2 = brow[2];
  float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
  if (XK > 0) { x0 = wx[threadIdx.x]; if (XK > 4) x1 = wx[GMPC_RW_THREADS + threadIdx.x]; }
#pragma unroll
  for (int q = 0; q < H / 4; ++q) {
    const float4 b = b0;
    b0 = b1;
    b1 = b2;
    if (q + 3 < H / 4) b2 = brow[q + 3];
    float4 w;
    if (q < QR) {
      w = make_float4(wr[4 * q + 0], wr[4 * q + 1], wr[4 * q + 2], wr[4 * q + 3]);
    } else {
      w = x0;
      x0 = x1;
      if (q - QR + 2 < XK / 4) x1 = wx[(q - QR + 2) * GMPC_RW_THREADS + threadIdx.x];
    }
    d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, b.x, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, b.y, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, b.z, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, b.w, d3, 0, 0, 0);
  }
  return (d0 + d1) + (d2 + d3);
}

#ifndef GMPC_TRAJ_MINW
#define GMPC_TRAJ_MINW 4
#endif
#define GMPC_TRAJ_THREADS 512   // k_traj: 8 waves, the upper 4 take the second half of every K range

// KH > 0 selects the register-weight MFMA form of the dynamics network (hidden width KH, NHL hidden
// KH x KH layers, 256 threads; see rw_layer above); KH = 0 is the general VALU form (512 threads).
template <bool LS, int KH = 0, int NHL = 0>
__global__ __launch_bounds__(KH ? GMPC_RW_THREADS : GMPC_TRAJ_THREADS, KH ? 1 : GMPC_TRAJ_MINW) void k_traj(TrajArgs a) {
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
  for (int e = tid; e < a.sw0; e += blockDim.x) w0_s[e] = a.dyn.W[0][e];
  for (int e = tid; e < a.swl; e += blockDim.x) wl_s[e] = a.dyn.W[Lh][e];
  const float* const W0 = a.sw0 ? w0_s : a.dyn.W[0];
  const float* const WL = a.swl ? wl_s : a.dyn.W[Lh];
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  // ---- register-weight form (KH > 0): LDS rows behind the staged weights, weights and biases in
  // registers for the whole horizon
  float* const zT = wl_s + a.swl;                        // [4][ZP]   layer-0 input rows (x ; u) per slot
  float* const usf = zT + 4 * GMPC_RW_ZP;                // [32][4]   controls of the step, [j][slot]
  float* const hA = usf + 4 * 32;                        // [4][HP]   hidden activation rows, ping
  float* const hB = hA + 4 * GMPC_RW_HP;                 //           pong
  float* const op = hB + 4 * GMPC_RW_HP;                 // [4][32][4] output-layer partials per wave
  float4* const wx_s = reinterpret_cast<float4*>(op + 4 * 32 * 4);   // [XK / 4][256] last XK weight rows
                                                                     // of the last hidden layer
  const int rg = lane >> 2, rp = lane & 3;
  const int nnA = 64 * wave + lane;                      // neuron of this lane's A operand
  float wr[NHL > 0 ? NHL : 1][KH > 0 ? KH : 1];
  float rbias[NHL + 1][4];
  if constexpr (KH > 0) {
#pragma unroll
    for (int hl = 0; hl < NHL; ++hl) {
      const float* Wl = a.dyn.W[hl + 1];
      const int kreg = hl == NHL - 1 ? KH - GMPC_RW_XK : KH;
#pragma unroll
      for (int k = 0; k < kreg; ++k) wr[hl][k] = nnA < KH ? Wl[(size_t)k * KH + nnA] : 0.f;
      if (hl == NHL - 1) {
#pragma unroll 4
        for (int q = 0; q < GMPC_RW_XK / 4; ++q) {
          const float* wq = Wl + (size_t)(KH - GMPC_RW_XK + 4 * q) * KH + nnA;
          wx_s[q * GMPC_RW_THREADS + tid] =
              nnA < KH ? make_float4(wq[0], wq[KH], wq[2 * KH], wq[3 * KH]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int l = 0; l <= NHL; ++l)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ne = 64 * wave + 4 * rg + i;
        rbias[l][i] = ne < KH ? a.dyn.b[l][ne] : 0.f;
      }
    for (int e = tid; e < 4 * GMPC_RW_ZP + 4 * 32; e += blockDim.x) zT[e] = 0.f;
    __syncthreads();
  }
  // output-layer bias of the state coordinate thread tid reduces (tid < 4 n)
  const float rbl = (KH > 0 && tid < 4 * n) ? a.dyn.b[Lh][tid >> 2] : 0.f;
  const float* const xs_ = KH > 0 ? xf : aAf;            // x_t and u_t of the four slots, [i][slot]
  const float* const us_ = KH > 0 ? usf : aAf + 4 * n;
  auto put_u = [&](int c, int j, float u) {
    if (KH > 0) {
      usf[j * 4 + c] = u;
      zT[c * GMPC_RW_ZP + n + j] = u;
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
      if (KH > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) zT[c * GMPC_RW_ZP + i] = f4get(v, c);
      }
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
      if (KH == 0)
        for (int i = tid; i < n; i += blockDim.x) actA[i] = xcur[i];
      if (LS) {
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
          // layer 0: A operand from the LDS copy of W_0 ([k][H]); rows k >= n + m of zT are zero
          const int K0 = n + m;
          const float4* brow = reinterpret_cast<const float4*>(zT + rp * GMPC_RW_ZP);
          const float* wcol = W0 + (nnA < KH ? nnA : 0);
          f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
          for (int q = 0; 4 * q < K0; ++q) {
            const float4 b = brow[q];
            float aw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = 4 * q + e;
              aw[e] = (k < K0 && nnA < KH) ? wcol[(size_t)k * KH] : 0.f;
            }
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[0], b.x, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[1], b.y, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[2], b.z, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[3], b.w, d1, 0, 0, 0);
          }
          rw_hidden_epilogue(d0 + d1, rbias[0], KH, hA, mb, mstride, wbits);
        }
        __syncthreads();
        TS_(2)
        float* hin = hA;
        float* hout = hB;
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) {
          const f32x4_t d = hl == NHL - 1 ? rw_layer<KH, GMPC_RW_XK>(wr[hl], hin, wx_s)
                                          : rw_layer<KH, 0>(wr[hl], hin, wx_s);
          rw_hidden_epilogue(d, rbias[hl + 1], KH, hout, mb + (hl + 1) * GMPC_MW, mstride, wbits);
          __syncthreads();
          TS_(3 + hl)
          float* tmp = hin; hin = hout; hout = tmp;
        }
        {
          // output layer: the K range is split over the 4 waves and the two lane halves; block
          // (ph, og) of the MFMA: outputs 4 og + i, k = KW wave + 2 e + ph
          constexpr int KW = KH / 4;
          const int ph = lane >> 5, og = rg & 7;
          const int no = 4 * og + rp;
          const float* brow = hin + rp * GMPC_RW_HP + KW * wave + ph;
          const float* wcol = WL + (size_t)(KW * wave + ph) * n + (no < n ? no : 0);
          f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < (KW + 1) / 2; ++e) {
            const bool kin = 2 * e + ph < KW;
            const float aw = (no < n && kin) ? wcol[(size_t)2 * e * n] : 0.f;
            const float bv = kin ? brow[2 * e] : 0.f;
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(aw, bv, d, 0, 0, 0);
          }
          float a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3];
          rw_swap_halves(a0, a1);      // a0 + a1: output 4 og + ph, both k phases
          rw_swap_halves(a2, a3);      // a2 + a3: output 4 og + 2 + ph
          op[(wave * 32 + 4 * og + ph) * 4 + rp] = a0 + a1;
          op[(wave * 32 + 4 * og + 2 + ph) * 4 + rp] = a2 + a3;
        }
        __syncthreads();
        if (tid < 4 * n) {
          const int no = tid >> 2, c = tid & 3;
          float sum = op[no * 4 + c];
#pragma unroll
          for (int w = 1; w < GMPC_RW_THREADS / 64; ++w) sum += op[(w * 32 + no) * 4 + c];
          const float v = (sum + rbl) + xf[no * 4 + c];
          xf[no * 4 + c] = v;
          zT[c * GMPC_RW_ZP + no] = v;
          if ((wbits >> c) & 1u) {
            float* Xo = LS ? a.Xc : a.X;
            Xo[((LS ? CI(c) : (size_t)BI(c)) * (T + 1) + t + 1) * n + no] = v;
          }
        }
        __syncthreads();
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
    a.sw0 = a.dyn.dims[0] * a.dyn.dims[1];
    a.swl = a.dyn.dims[a.dyn.L - 1] * a.n;
    return ((size_t)2 * a.aw + a.pw + GMPC_THREADS + a.n) * sizeof(float4) +
           ((size_t)a.sw0 + a.swl + 4 * GMPC_RW_ZP + 4 * 32 + 2 * 4 * GMPC_RW_HP + 4 * 32 * 4 +
            (size_t)GMPC_RW_XK * GMPC_RW_THREADS) * sizeof(float);
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
  if (!attr) { traj_attr(&k_traj<false>); traj_attr(&k_traj<false, 200, 2>); attr = true; }
  const int grid = (a.B + GMPC_TB - 1) / GMPC_TB;
  if (rw)
    hipLaunchKernelGGL((k_traj<false, 200, 2>), dim3(grid), dim3(GMPC_RW_THREADS), lds, s, a);
  else
    hipLaunchKernelGGL(k_traj<false>, dim3(grid), dim3(GMPC_TRAJ_THREADS), lds, s, a);
}
int gmpc_launch_linesearch(const TrajArgs& a0, const LsWork& w, hipStream_t s) {
  TrajArgs a = a0;
  const bool rw = traj_rw_shape(a);
  const size_t lds = traj_lds(a, rw);
  static bool attr = false;
  if (!attr) { traj_attr(&k_traj<true>); traj_attr(&k_traj<true, 200, 2>); attr = true; }
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
    if (rw)
      hipLaunchKernelGGL((k_traj<true, 200, 2>), lsgrid, dim3(GMPC_RW_THREADS), lds, s, a);
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
