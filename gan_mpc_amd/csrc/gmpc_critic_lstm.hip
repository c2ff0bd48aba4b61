// Critic LSTM, second generation (round 3): the small-input form (n <= 32, F = 64) of k_lstm_fwd / k_lstm_bwd with
// ONE workgroup barrier per time step and the LSTM weight gradients accumulated inside the backward sweep.
//
// Reference arithmetic: critic/nn.py:28-38 (flax OptimizedLSTMCell scanned over the sequence, zero carry, gate
// order i, f, g, o), gan/js_policy.py:41-58 (its gradient); the same formulas as gmpc_critic.hip.
//
// Mapping.  A workgroup (4 waves) owns 4 sequences ("slots") for the whole sweep.  Wave w owns the hidden units
// 16 w .. 16 w + 15 and ALL FOUR gates of them: lane l = 16 q + ul is gate column j = 64 q + 16 w + ul (q = i, f, g,
// o).  The gate pre-activations are one chain of v_mfma_f32_4x4x1_16b_f32 with the [h ; x] image of the 4 slots as
// the broadcast A operand (gmpc_device.h: rw_mfma) and the lane's weight column -- 64 + n registers, loaded once --
// as B; D register s = slot s.  A 4 x 4 transpose between the four 16-lane rows and the four registers
// (v_permlane32_swap + v_permlane16_swap, two each) then gives lane (q, ul) the gates i, f, g, o of slot q of its
// unit: the cell update runs INSIDE the wave, and h goes back to the LDS image [unit][slot] for the next step.  The
// image is double-buffered, so the step needs a single barrier (the first-generation kernels needed three).
//
// Backward: the same roles in reverse.  Lane (slot, unit) forms dz = (zi, zf, zg, zo), the transpose hands lane
// (gate, ul) its column's dz for the 4 slots, and the wave multiplies its OWN 64 columns: dh_{t-1} partial sums over
// j (A = the wave-private dz image, B = the transposed recurrent weights, 64 MFMAs), added across the four waves
// through LDS behind the step's barrier.  The weight gradient dW[k][j] += [h_{t-1} ; x_t][k][s] dz[j][s] is the same
// instruction with K = 1: A = four rows k of the transposed image [slot][k], B = the lane's dz of that slot -- 84
// MFMAs per step into 84 accumulator registers that stay in the register file for the whole sequence.  Nothing of
// dz is written to memory; a workgroup leaves one partial [85][256] (21 row groups + the bias row), reduced in
// workgroup order by k_lstm_wreduce (deterministic).
//
// Saved by the forward sweep for the backward one (internal layouts, one float per thread and step, coalesced):
// the activated gates AFTER the transpose (4 planes), c_t, h_t.
#include "gmpc_device.h"
#include <cstdlib>

typedef unsigned v2u_ __attribute__((ext_vector_type(2)));

// rows = groups of 16 lanes.  Register s of row q  <->  register q of row s.
__device__ __forceinline__ void swap32_(float& a, float& b) {
  const v2u_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16_(float& a, float& b) {
  const v2u_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
__device__ __forceinline__ void transpose_rows4(float (&v)[4]) {
  swap32_(v[0], v[2]);      // rows {0,1} of v2 <-> rows {2,3} of v0
  swap32_(v[1], v[3]);
  swap16_(v[0], v[1]);      // odd rows of v0 <-> even rows of v1
  swap16_(v[2], v[3]);
}

// sigmoid, and tanh as 2 sigmoid(2 x) - 1: one code path for the four gate rows of a wave.  exp and the reciprocal
// are the hardware's v_exp_f32 / v_rcp_f32 (1 ulp each; 4 instructions per value instead of ~25 for expf and an
// IEEE division: the sweep is a latency chain and these sit on it five times per step).  GMPC_LSTM2_EXACT_ACT=1 at
// build time restores expf and the division (A/B of the parity figures).
__device__ __forceinline__ float gate_act(float x, float aa, float cc) {
#ifdef GMPC_LSTM2_EXACT_ACT
  return fmaf(aa, 1.0f / (1.0f + expf(-aa * x)), cc);
#else
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * aa * x);
  return fmaf(aa, __builtin_amdgcn_rcpf(1.0f + e), cc);
#endif
}

#define GMPC_LSTM2_ROWS(NX) (((64 + (NX) + 3) / 4) * 4 + 1)    // rows of a workgroup's weight-gradient partial

template <int NX>
__global__ __launch_bounds__(GMPC_THREADS, 2) void k_lstm_fwd2(int Bc, CriticDesc cd, const float* __restrict__ xseq,
                                                               float* __restrict__ G, float* __restrict__ Cst,
                                                               float* __restrict__ Hst, float* __restrict__ hT) {
  constexpr int NXR = (NX + 15) / 16;
  __shared__ __attribute__((aligned(16))) float hbuf[2][256];          // [unit][slot]
  __shared__ __attribute__((aligned(16))) float xbuf[2][NXR * 64];     // [k][slot], rows >= n stay zero
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, ul = l & 15;
  const int u = 16 * w + ul, j = 64 * q + u;
  const int n = cd.n, T1 = cd.T1, s0 = blockIdx.x * 4;
  float wh[64], wx[NX];
#pragma unroll
  for (int k = 0; k < 64; ++k) wh[k] = cd.Wcat[(size_t)(n + k) * 256 + j];
#pragma unroll
  for (int k = 0; k < NX; ++k) wx[k] = k < n ? cd.Wcat[(size_t)k * 256 + j] : 0.f;
  const float bj = cd.b[j];
  const float aa = q == 2 ? 2.f : 1.f, cc = q == 2 ? -1.f : 0.f;
  // x loader: thread e < 4 n owns element (slot e / n, coordinate e % n)
  const bool xon = tid < 4 * n;
  const int xs = xon ? tid / n : 0, xi = xon ? tid - xs * n : 0;
  const float* xptr = xseq + ((size_t)min(s0 + xs, Bc - 1) * T1) * n + xi;
  for (int e = tid; e < NXR * 64; e += GMPC_THREADS) { xbuf[0][e] = 0.f; xbuf[1][e] = 0.f; }
  hbuf[0][tid] = 0.f;
  __syncthreads();
  float xnext = 0.f;
  if (xon) {
    xbuf[0][xi * 4 + xs] = xptr[0];
    if (T1 > 1) xnext = xptr[n];
  }
  float c = 0.f;
  const size_t wbase = (size_t)blockIdx.x * T1;
  __syncthreads();
  for (int t = 0; t < T1; ++t) {
    const int cur = t & 1;
    f32x4_t d0 = {bj, bj, bj, bj}, d1 = {0.f, 0.f, 0.f, 0.f};
    float ax[NXR], ah[4];
#pragma unroll
    for (int r = 0; r < NXR; ++r) ax[r] = xbuf[cur][64 * r + l];
#pragma unroll
    for (int r = 0; r < 4; ++r) ah[r] = hbuf[cur][64 * r + l];
    rw_static_for<NX>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k & 1) rw_mfma<k>(d1, ax[k >> 4], wx[k]);
      else rw_mfma<k>(d0, ax[k >> 4], wx[k]);
    });
    rw_static_for<64>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k & 1) rw_mfma<k>(d1, ah[k >> 4], wh[k]);
      else rw_mfma<k>(d0, ah[k >> 4], wh[k]);
    });
    const f32x4_t d = d0 + d1;
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = gate_act(d[s], aa, cc);
    transpose_rows4(v);                       // lane (q, ul): v = (i, f, g, o) of slot q, unit u
    const size_t row = wbase + t;
#pragma unroll
    for (int s = 0; s < 4; ++s) G[(row * 4 + s) * 256 + tid] = v[s];
    c = fmaf(v[1], c, v[0] * v[2]);
    const float h = v[3] * gate_act(c, 2.f, -1.f);
    Cst[row * 256 + tid] = c;
    Hst[row * 256 + tid] = h;
    hbuf[cur ^ 1][u * 4 + q] = h;
    if (xon && t + 1 < T1) {
      xbuf[cur ^ 1][xi * 4 + xs] = xnext;
      if (t + 2 < T1) xnext = xptr[(size_t)(t + 2) * n];
    }
    if (t == T1 - 1 && s0 + q < Bc) hT[(size_t)(s0 + q) * 64 + u] = h;
    __syncthreads();
  }
}

// WANT_W: accumulate the LSTM weight gradients (partial per workgroup -> Wp); WANT_DX: d loss / d x -> dxseq.
// NG: groups of 4 sequences per workgroup.  With NG = 2 every wave runs the step for two independent groups back to
// back: the transposed weights (64 registers) and the gradient accumulators (84) are shared, one group's LDS round
// trips and transcendentals sit under the other's MFMAs, the workgroup needs ONE wave per SIMD -- it fits beside the
// one-wave Riccati sweep (176 registers) where two 200-register waves did not -- and leaves half the partials.
template <int NX, bool WANT_W, bool WANT_DX, int NG>
__global__ __launch_bounds__(GMPC_THREADS, NG == 1 ? 2 : 1) void k_lstm_bwd2(
    int Bc, CriticDesc cd, const float* __restrict__ xseq, const float* __restrict__ G, const float* __restrict__ Cst,
    const float* __restrict__ Hst, const float* __restrict__ dhT, float* __restrict__ Wp, float* __restrict__ dxseq) {
  constexpr int KG = (64 + NX + 3) / 4;          // row groups of the weight gradient
  constexpr int XW = 128;                        // floats per slot row of the transposed image [h (64) ; x ; 0]
  static_assert(NX <= 32, "x part of the image: one register");
  __shared__ __attribute__((aligned(16))) float4 zbuf[NG][4][64];        // wave-private dz image [column][slot]
  __shared__ __attribute__((aligned(16))) float4 part[NG][2][4][64];     // [buffer][wave][unit] partial dh over the slots
  __shared__ __attribute__((aligned(16))) float4 partx[NG][WANT_DX ? 2 : 1][4][32];
  __shared__ __attribute__((aligned(16))) float xh[NG][WANT_W ? 2 : 1][4][XW];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, ul = l & 15;
  const int u = 16 * w + ul;
  const int n = cd.n, T1 = cd.T1;
  // transposed weights of this wave's 64 columns: B operand of step jj is W[row][column(w, jj)]
  float whT[64], wxT[WANT_DX ? 64 : 1];
#pragma unroll
  for (int jj = 0; jj < 64; ++jj) {
    const int col = 64 * (jj >> 4) + 16 * w + (jj & 15);
    whT[jj] = cd.Wcat[(size_t)(n + l) * 256 + col];
    if (WANT_DX) wxT[jj] = l < n ? cd.Wcat[(size_t)l * 256 + col] : 0.f;
  }
  f32x4_t acc[WANT_W ? KG : 1];
  float db = 0.f;
  if (WANT_W) {
#pragma unroll
    for (int g = 0; g < KG; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool xon = tid < 4 * n;
  const int xs = xon ? tid / n : 0, xi = xon ? tid - xs * n : 0;
  // per group: first sequence, block of the saves, x row of this thread's loader role
  int s0[NG];
  size_t wbase[NG], xrow[NG];
  float hpre[NG], xpre[NG], pg[NG][4], pc[NG], ccur[NG], dc[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int blk = blockIdx.x * NG + gi;
    s0[gi] = blk * 4;
    // (a group past the batch re-reads the last block's saves; its d h_T is zero, so it contributes exact zeros)
    wbase[gi] = (size_t)min(blk, (Bc + 3) / 4 - 1) * T1;
    xrow[gi] = (size_t)min(s0[gi] + xs, Bc - 1) * T1;
    // d loss / d h_T enters through the reduction buffer: wave 0's slice carries it, the others zero
    float* partf = reinterpret_cast<float*>(&part[gi][0][0][0]);
    const bool ok = s0[gi] + q < Bc;
    const float v0 = ok ? dhT[(size_t)(s0[gi] + q) * 64 + u] : 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) partf[(ww * 64 + u) * 4 + q] = ww == 0 ? v0 : 0.f;
    if (WANT_DX && tid < 128) partx[gi][0][tid >> 5][tid & 31] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WANT_W) {
      for (int e = tid; e < 2 * 4 * XW; e += GMPC_THREADS) (&xh[gi][0][0][0])[e] = 0.f;
    }
    hpre[gi] = 0.f; xpre[gi] = 0.f; dc[gi] = 0.f;
  }
  __syncthreads();
  // saved gates / cell states of the step are requested one step ahead
  auto prefetch = [&](int gi, int t) {
    const size_t row = wbase[gi] + t;
#pragma unroll
    for (int s = 0; s < 4; ++s) pg[gi][s] = G[(row * 4 + s) * 256 + tid];
    pc[gi] = t > 0 ? Cst[(row - 1) * 256 + tid] : 0.f;
  };
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    // the image of step t - 1 (h_{t-2}, x_{t-1}) is written during step t from registers loaded during step t + 1
    if (WANT_W) {
      // image of the last step: h_{T1-2}, x_{T1-1}
      xh[gi][0][q][u] = T1 > 1 ? Hst[(wbase[gi] + T1 - 2) * 256 + tid] : 0.f;
      if (xon) xh[gi][0][xs][64 + xi] = xseq[(xrow[gi] + T1 - 1) * n + xi];
      if (T1 > 2) hpre[gi] = Hst[(wbase[gi] + T1 - 3) * 256 + tid];
      if (xon && T1 > 1) xpre[gi] = xseq[(xrow[gi] + T1 - 2) * n + xi];
    }
    ccur[gi] = Cst[(wbase[gi] + T1 - 1) * 256 + tid];
    prefetch(gi, T1 - 1);
  }
  __syncthreads();
  for (int t = T1 - 1; t >= 0; --t) {
    const int cur = (T1 - 1 - t) & 1;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const float* partf = reinterpret_cast<const float*>(&part[gi][0][0][0]);
      const float* partxf = reinterpret_cast<const float*>(&partx[gi][0][0][0]);
      // dh_t: the four waves' partial sums (fixed order)
      const float* pp = partf + cur * 4 * 64 * 4 + u * 4 + q;
      const float dh = ((pp[0] + pp[64 * 4]) + pp[2 * 64 * 4]) + pp[3 * 64 * 4];
      if (WANT_DX && t < T1 - 1 && xon) {
        const float* px = partxf + cur * 4 * 32 * 4 + xi * 4 + xs;
        if (s0[gi] + xs < Bc)
          dxseq[(xrow[gi] + t + 1) * n + xi] = ((px[0] + px[32 * 4]) + px[2 * 32 * 4]) + px[3 * 32 * 4];
      }
      const float ig = pg[gi][0], fg = pg[gi][1], gg = pg[gi][2], og = pg[gi][3];
      const float ct = ccur[gi], cprev = pc[gi];
      ccur[gi] = cprev;
      if (t > 0) prefetch(gi, t - 1);
      float hnew = 0.f, xnew = 0.f;
      if (WANT_W) {
        if (t > 2) hnew = Hst[(wbase[gi] + t - 3) * 256 + tid];
        if (xon && t > 1) xnew = xseq[(xrow[gi] + t - 2) * n + xi];
      }
      const float tc = gate_act(ct, 2.f, -1.f);
      const float d_o = dh * tc;
      float dcv = fmaf(dh * og, 1.f - tc * tc, dc[gi]);
      float z[4];
      z[0] = dcv * gg * ig * (1.f - ig);
      z[1] = dcv * cprev * fg * (1.f - fg);
      z[2] = dcv * ig * (1.f - gg * gg);
      z[3] = d_o * og * (1.f - og);
      dc[gi] = dcv * fg;
      transpose_rows4(z);                       // lane (gate q, ul): z[s] = dz[column][slot s]
      if (WANT_W) db += (z[0] + z[1]) + (z[2] + z[3]);
      zbuf[gi][w][l] = make_float4(z[0], z[1], z[2], z[3]);
      // cross-lane exchange inside ONE wave (its LDS accesses execute in order): wavefront-scope fences keep the
      // compiler from forwarding or reordering across it, no instruction is emitted
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float az[4];
      const float* zf = reinterpret_cast<const float*>(&zbuf[gi][w][0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) az[r] = zf[64 * r + l];
      {
        f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        rw_static_for<64>([&](auto jc) __attribute__((always_inline)) {
          constexpr int jj = decltype(jc)::value;
          if constexpr (jj & 1) rw_mfma<jj>(d1, az[jj >> 4], whT[jj]);
          else rw_mfma<jj>(d0, az[jj >> 4], whT[jj]);
        });
        const f32x4_t d = d0 + d1;
        part[gi][cur ^ 1][w][l] = make_float4(d[0], d[1], d[2], d[3]);
      }
      if constexpr (WANT_DX) {
        f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        rw_static_for<64>([&](auto jc) __attribute__((always_inline)) {
          constexpr int jj = decltype(jc)::value;
          if constexpr (jj & 1) rw_mfma<jj>(d1, az[jj >> 4], wxT[jj]);
          else rw_mfma<jj>(d0, az[jj >> 4], wxT[jj]);
        });
        const f32x4_t d = d0 + d1;
        if (l < 32) partx[gi][cur ^ 1][w][l] = make_float4(d[0], d[1], d[2], d[3]);
      }
      if constexpr (WANT_W) {
        float xr[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          xr[s][0] = xh[gi][cur][s][l];
          xr[s][1] = xh[gi][cur][s][64 + l];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          rw_static_for<KG>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            rw_mfma<g>(acc[g], xr[s][g >> 4], z[s]);
          });
        }
        // image of the next step (t - 1): h_{t-2}, x_{t-1}
        if (t > 0) {
          xh[gi][cur ^ 1][q][u] = hpre[gi];
          if (xon) xh[gi][cur ^ 1][xs][64 + xi] = xpre[gi];
        }
        hpre[gi] = hnew;
        xpre[gi] = xnew;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    if (WANT_DX && xon && s0[gi] + xs < Bc) {
      const int cur = T1 & 1;
      const float* px = reinterpret_cast<const float*>(&partx[gi][0][0][0]) + cur * 4 * 32 * 4 + xi * 4 + xs;
      dxseq[xrow[gi] * n + xi] = ((px[0] + px[32 * 4]) + px[2 * 32 * 4]) + px[3 * 32 * 4];
    }
  }
  if constexpr (WANT_W) {
    float* wp = Wp + (size_t)blockIdx.x * GMPC_LSTM2_ROWS(NX) * 256 + tid;
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) wp[(size_t)(4 * g + i) * 256] = acc[g][i];
    wp[(size_t)(4 * KG) * 256] = db;
  }
}

// gWx[n][256], gWh[64][256], gb[256] = sums over the workgroups' partials, in workgroup order (deterministic).
// One thread per (row, thread column); 8 partial sums in flight per thread, four thread groups share the range.
__global__ __launch_bounds__(256) void k_lstm_wreduce(int nwg, int n, int rows, const float* __restrict__ Wp,
                                                      float* __restrict__ gWx, float* __restrict__ gWh,
                                                      float* __restrict__ gb) {
  __shared__ float sh[4][64];
  const int el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;             // element of the [rows][256] partial
  const int qn = (nwg + 3) / 4;
  const int g0 = seg * qn, g1 = min(nwg, g0 + qn);
  const size_t stride = (size_t)rows * 256;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < rows * 256) {
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += Wp[(size_t)(g + k) * stride + e];
    }
    for (; g < g1; ++g) a[0] += Wp[(size_t)g * stride + e];
  }
  sh[seg][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (seg != 0 || e >= rows * 256) return;
  const float v = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
  const int row = e >> 8, col = e & 255;
  const int w = col >> 6, l = col & 63;
  const int jcol = 64 * (l >> 4) + 16 * w + (l & 15);
  if (row < 64) gWh[row * 256 + jcol] = v;
  else if (row < 64 + n) gWx[(row - 64) * 256 + jcol] = v;
  else if (row == rows - 1) gb[jcol] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// launchers: return false when the shape is not one of the instantiations (the caller falls back to gmpc_critic.hip)
// ---------------------------------------------------------------------------------------------------------------
static int lstm2_nx(const CriticDesc& cd) {
  if (cd.F != 64 || cd.n < 1 || cd.n > 32) return 0;
  return cd.n <= 4 ? 4 : cd.n <= 8 ? 8 : cd.n <= 17 ? 17 : 32;
}

bool gmpc_lstm2_supported(const CriticDesc& cd) { return lstm2_nx(cd) != 0; }

// floats of the weight-gradient partial buffer for Bc sequences
long gmpc_lstm2_wpart_floats(const CriticDesc& cd, int Bc) {
  const int nx = lstm2_nx(cd);
  return nx ? (long)((Bc + 3) / 4) * GMPC_LSTM2_ROWS(nx) * 256 : 0;
}

bool gmpc_launch_lstm_fwd2(int Bc, const CriticDesc& cd, const float* xseq, float* G, float* Cst, float* Hst,
                           float* hT, hipStream_t s) {
  const dim3 grid((Bc + 3) / 4), blk(GMPC_THREADS);
  switch (lstm2_nx(cd)) {
    case 4: hipLaunchKernelGGL(k_lstm_fwd2<4>, grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, hT); return true;
    case 8: hipLaunchKernelGGL(k_lstm_fwd2<8>, grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, hT); return true;
    case 17: hipLaunchKernelGGL(k_lstm_fwd2<17>, grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, hT); return true;
    case 32: hipLaunchKernelGGL(k_lstm_fwd2<32>, grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, hT); return true;
    default: return false;
  }
}

#ifndef GMPC_LSTM2_BWD_NG
#define GMPC_LSTM2_BWD_NG 2     // groups of 4 sequences per workgroup of the backward sweep (see k_lstm_bwd2)
#endif
static int bwd2_groups(int Bc) {
  static const int env = getenv("GMPC_LSTM2_BWD_NG") ? atoi(getenv("GMPC_LSTM2_BWD_NG")) : GMPC_LSTM2_BWD_NG;
  return (env == 2 && Bc > 4) ? 2 : 1;
}

template <int NX>
static void launch_bwd2(int Bc, const CriticDesc& cd, const float* xseq, const float* G, const float* Cst,
                        const float* Hst, const float* dhT, float* Wp, float* dxseq, hipStream_t s) {
  const int ng = bwd2_groups(Bc);
  const dim3 grid((Bc + 4 * ng - 1) / (4 * ng)), blk(GMPC_THREADS);
  // (both wanted is not a combination the API asks for today: two sweeps)
  if (Wp) {
    if (ng == 2)
      hipLaunchKernelGGL((k_lstm_bwd2<NX, true, false, 2>), grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, dhT, Wp, nullptr);
    else
      hipLaunchKernelGGL((k_lstm_bwd2<NX, true, false, 1>), grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, dhT, Wp, nullptr);
  }
  if (dxseq) {
    if (ng == 2)
      hipLaunchKernelGGL((k_lstm_bwd2<NX, false, true, 2>), grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, dhT, nullptr, dxseq);
    else
      hipLaunchKernelGGL((k_lstm_bwd2<NX, false, true, 1>), grid, blk, 0, s, Bc, cd, xseq, G, Cst, Hst, dhT, nullptr, dxseq);
  }
}

// Wp != null: weight gradients -> gWx, gWh, gb (sums over the Bc sequences); dxseq != null: input gradient
bool gmpc_launch_lstm_bwd2(int Bc, const CriticDesc& cd, const float* xseq, const float* G, const float* Cst,
                           const float* Hst, const float* dhT, float* Wp, float* gWx, float* gWh, float* gb,
                           float* dxseq, hipStream_t s) {
  const int nx = lstm2_nx(cd);
  switch (nx) {
    case 4: launch_bwd2<4>(Bc, cd, xseq, G, Cst, Hst, dhT, Wp, dxseq, s); break;
    case 8: launch_bwd2<8>(Bc, cd, xseq, G, Cst, Hst, dhT, Wp, dxseq, s); break;
    case 17: launch_bwd2<17>(Bc, cd, xseq, G, Cst, Hst, dhT, Wp, dxseq, s); break;
    case 32: launch_bwd2<32>(Bc, cd, xseq, G, Cst, Hst, dhT, Wp, dxseq, s); break;
    default: return false;
  }
  if (Wp) {
    const int ng = bwd2_groups(Bc);
    const int rows = GMPC_LSTM2_ROWS(nx), nwg = (Bc + 4 * ng - 1) / (4 * ng);
    hipLaunchKernelGGL(k_lstm_wreduce, dim3((rows * 256 + 63) / 64), dim3(256), 0, s, nwg, cd.n, rows, Wp, gWx, gWh,
                       gb);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Critic head on the matrix cores (critic/nn.py:40-42: (num_layers - 1) x relu Dense, Dense(1); gan/js_policy.py:
// 41-68: the losses).  Forward, loss and backward of R = 4 G sequences per workgroup in one launch.  Thread j is
// neuron j of every layer (widths <= 256); a layer is one chain of v_mfma_f32_4x4x1_16b_f32 with the activations of
// the R rows -- an LDS image [group][k][4 rows] -- as the broadcast A operand of G independent accumulators and the
// lane's weight W[k][j] as B, streamed from L2 (coalesced over j) three 16-row chunks ahead: one weight load feeds G
// MFMAs.  Backward layers run the same routine on the transposed weights with the delta image as A.  The relu masks
// stay in registers (thread j is the same neuron in both directions).  Stored for the weight-gradient GEMMs: every
// layer's input (`acts`) and delta (`dels`) row-major, and for the last layer (one output) the products
// act[row][k] * dscore[row] with dscore in column K: its weight and bias gradients are column sums (`plast`).
// ---------------------------------------------------------------------------------------------------------------
#define GMPC_HEAD2_LD 264     // leading dimension of plast: 256 products + dscore, padded to a multiple of 8
#ifndef GMPC_HEAD_RING
#define GMPC_HEAD_RING 3       // weight chunks of head_layer in flight + 1
#endif

// d[g] += sum_k img[g][k][.] (x) W[k][col] over Kred rows of W (leading dimension ldw).  The weights come through a
// buffer resource that ends with the matrix: rows past Kred (the last 16-row chunk, the chunks requested ahead of
// the end) and the lanes without a column read as zero, so the loads are unconditional -- straight-line code whose
// outstanding-load count hipcc can follow (with guarded global loads it waited for ALL loads before every chunk).
// The whole byte offset sits in the per-lane part of the address, which the range check is documented to cover.
template <int G>
__device__ __forceinline__ void head_layer(const float* __restrict__ W, int ldw, int col, bool colok, int Kred,
                                           const float* img, int lane, f32x4_t (&d)[G]) {
  const int chunks = (Kred + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(W), 0, __builtin_amdgcn_readfirstlane(Kred * ldw * (int)sizeof(float)), 0x00020000);
  const int voff = colok ? col * 4 : 0x7ffffff0;
  const int rowb = ldw * 4;
  // ring of GMPC_HEAD_RING chunks: RING - 1 chunks (16 weight rows each) are on their way while one multiplies
  constexpr int RING = GMPC_HEAD_RING;
  float wq[RING][16], aq[RING][G];
  auto loadw = [&](int c, float (&dst)[16]) {
    const int base = voff + 16 * c * rowb;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      dst[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, colok ? base + kk * rowb : 0x7ffffff0, 0, 0));
  };
  auto loada = [&](int c, float (&dst)[G]) {
    const int cc = c < 16 ? c : 15;          // (chunks requested past the end of the image are never multiplied)
#pragma unroll
    for (int g = 0; g < G; ++g) dst[g] = img[g * 1024 + 64 * cc + lane];
  };
  auto run = [&](const float (&a)[G], const float (&w)[16]) {
    rw_static_for<16>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(kc)::value;
#pragma unroll
      for (int g = 0; g < G; ++g) rw_mfma<kk>(d[g], a[g], w[kk]);
    });
  };
#pragma unroll
  for (int i = 0; i + 1 < RING; ++i) { loadw(i, wq[i]); loada(i, aq[i]); }
  for (int c = 0; c < chunks; c += RING) {
    rw_static_for<RING>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int nx = (i + RING - 1) % RING;
      loadw(c + i + RING - 1, wq[nx]); loada(c + i + RING - 1, aq[nx]);
      __builtin_amdgcn_sched_barrier(0);
      if (c + i < chunks) run(aq[i], wq[i]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
}

template <int G>
__global__ __launch_bounds__(GMPC_THREADS, 2) void k_head2(int Bc, CriticDesc cd, int loss_kind,
                                                           const float* __restrict__ hT,
                                                           const float* __restrict__ label, float* __restrict__ score,
                                                           float* __restrict__ loss, float* __restrict__ acts,
                                                           float* __restrict__ dels, float* __restrict__ plast,
                                                           float* __restrict__ dhT, int act_stride) {
  constexpr int R = 4 * G;
  __shared__ __attribute__((aligned(16))) float img[2][G * 1024];     // [buffer][group][k][4 rows]
  __shared__ float red[4][R];
  __shared__ float dsc[R];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int s0 = blockIdx.x * R;
  const MlpDesc& hd = cd.head;
  const int L = hd.L;
  // layer-0 input: the final LSTM state (F0 = lstm_features values per sequence), rows clamped; the image is read in
  // whole 16-row chunks: rows F0 .. pad16(F0) - 1 are zero
  const int F0 = hd.dims[0], F0p = (F0 + 15) & ~15;
  for (int e = tid; e < G * 4 * F0p; e += GMPC_THREADS) {
    const int g = e / (4 * F0p), r = e - g * 4 * F0p, k = r >> 2, s = r & 3;
    const int row = s0 + 4 * g + s;
    const float v = k < F0 ? hT[(size_t)min(row, Bc - 1) * F0 + k] : 0.f;
    img[0][g * 1024 + k * 4 + s] = v;
    if (row < Bc && k < F0) acts[(size_t)row * act_stride + k] = v;
  }
  __syncthreads();
  unsigned zmask[GMPC_MAX_LAYERS];       // bit 4 g + s: relu open for row (g, s) of this thread's neuron
  int in = 0, aoff = F0;
#pragma unroll
  for (int lay = 0; lay < GMPC_MAX_LAYERS; ++lay) {
    zmask[lay] = 0;
    if (lay < L - 1) {
      const int K = hd.dims[lay], N = hd.dims[lay + 1];
      f32x4_t d[G];
#pragma unroll
      for (int g = 0; g < G; ++g) d[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (64 * w < N) {                                   // wave-uniform: waves without a neuron sit the layer out
        head_layer<G>(hd.W[lay], N, tid, tid < N, K, img[in], l, d);
        if constexpr (G == 4) mfma_fence<false>(d[0], d[1], d[2], d[3]);
        else if constexpr (G == 3) mfma_fence<false>(d[0], d[1], d[2]);
        else mfma_fence<false>(d[0], d[1]);
      }
      const float bj = tid < N ? hd.b[lay][tid] : 0.f;
      unsigned zm = 0;
      float* out = img[in ^ 1];
      const int Npad = (N + 15) & ~15;                    // the next layer reads whole 16-row chunks
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float r[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float z = d[g][s] + bj;
          const bool open = tid < N && z > 0.f;
          zm |= open ? (1u << (4 * g + s)) : 0u;
          r[s] = open ? z : 0.f;
          const int row = s0 + 4 * g + s;
          if (tid < N && row < Bc) acts[(size_t)row * act_stride + aoff + tid] = r[s];
        }
        if (tid < Npad) *reinterpret_cast<float4*>(&out[g * 1024 + tid * 4]) = make_float4(r[0], r[1], r[2], r[3]);
      }
      zmask[lay] = zm;
      aoff += N;
      in ^= 1;
      __syncthreads();
    }
  }
  // last layer: one output.  score[row] = sum_k W[k] act[k][row] + b
  const int KL = hd.dims[L - 1];
  const float wl = tid < KL ? hd.W[L - 1][tid] : 0.f;
  float av[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float4 a4 = tid < KL ? *reinterpret_cast<const float4*>(&img[in][g * 1024 + tid * 4])
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    av[g][0] = a4.x; av[g][1] = a4.y; av[g][2] = a4.z; av[g][3] = a4.w;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float sm = wave_sum(wl * av[g][s]);
      if (l == 0) red[w][4 * g + s] = sm;
    }
  }
  __syncthreads();
  if (tid < R) {
    const int row = s0 + tid;
    const float sc = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + hd.b[L - 1][0];
    float ds, ls;
    if (loss_kind == 0) {
      const float p = sigmoidf_(sc);
      const bool pos = label[min(row, Bc - 1)] > 0.f;
      ls = -logf(pos ? p : 1.f - p);
      ds = pos ? -(1.f - p) : p;
    } else if (loss_kind == 1) {
      const float p = sigmoidf_(sc);
      ls = -logf(p) + logf(1.f - p);
      ds = -1.f;
    } else {
      ls = 0.f;
      ds = 1.f;
    }
    if (row >= Bc) ds = 0.f;
    dsc[tid] = ds;
    if (row < Bc) {
      score[row] = sc;
      loss[row] = ls;
    }
  }
  __syncthreads();
  // delta of the last layer, its gradient products, and the delta handed to the layer below
  int doff = 0;
  for (int lay = 0; lay < L; ++lay) doff += hd.dims[lay + 1];
  doff -= 1;
  {
    float* out = img[in ^ 1];
    const int Kpad = (KL + 15) & ~15;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float r[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = s0 + 4 * g + s;
        const float ds = dsc[4 * g + s];
        if (row < Bc) {
          if (tid < KL) plast[(size_t)row * GMPC_HEAD2_LD + tid] = av[g][s] * ds;
          if (tid == 0) {
            plast[(size_t)row * GMPC_HEAD2_LD + KL] = ds;
            dels[(size_t)row * act_stride + doff] = ds;
          }
        }
        const bool open = L == 1 || ((zmask[(L + GMPC_MAX_LAYERS - 2) % GMPC_MAX_LAYERS] >> (4 * g + s)) & 1u);
        r[s] = (tid < KL && open) ? wl * ds : 0.f;
      }
      if (tid < Kpad) *reinterpret_cast<float4*>(&out[g * 1024 + tid * 4]) = make_float4(r[0], r[1], r[2], r[3]);
    }
    in ^= 1;
  }
  __syncthreads();
  // hidden layers, top down: img[in] holds the delta of layer `lay` ([N_lay] x rows)
#pragma unroll
  for (int lay = GMPC_MAX_LAYERS - 2; lay >= 0; --lay) {
    if (lay < L - 1) {
      const int K = hd.dims[lay], N = hd.dims[lay + 1];
      doff -= N;
      // this layer's delta for the weight gradients
      if (tid < N) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(&img[in][g * 1024 + tid * 4]);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int row = s0 + 4 * g + s;
            if (row < Bc) dels[(size_t)row * act_stride + doff + tid] = vv[s];
          }
        }
      }
      f32x4_t d[G];
#pragma unroll
      for (int g = 0; g < G; ++g) d[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (64 * w < K) {
        head_layer<G>(hd.WT[lay], K, tid, tid < K, N, img[in], l, d);
        if constexpr (G == 4) mfma_fence<false>(d[0], d[1], d[2], d[3]);
        else if constexpr (G == 3) mfma_fence<false>(d[0], d[1], d[2]);
        else mfma_fence<false>(d[0], d[1]);
      }
      if (lay > 0) {
        float* out = img[in ^ 1];
        const int Kpad = (K + 15) & ~15;
        const unsigned zm = zmask[(lay + GMPC_MAX_LAYERS - 1) % GMPC_MAX_LAYERS];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float r[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) r[s] = (tid < K && ((zm >> (4 * g + s)) & 1u)) ? d[g][s] : 0.f;
          if (tid < Kpad) *reinterpret_cast<float4*>(&out[g * 1024 + tid * 4]) = make_float4(r[0], r[1], r[2], r[3]);
        }
        in ^= 1;
        __syncthreads();
      } else if (tid < K) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int row = s0 + 4 * g + s;
            if (row < Bc) dhT[(size_t)row * F0 + tid] = d[g][s];
          }
      }
    }
  }
  if (L == 1 && tid < F0) {
    // no hidden layer: d hT = W_last * dscore
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = s0 + 4 * g + s;
        if (row < Bc) dhT[(size_t)row * F0 + tid] = wl * dsc[4 * g + s];
      }
  }
}

// all transposes of an MLP's kernels in one launch: blockIdx.z = layer, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_mlp_transpose_all(MlpDesc d) {
  __shared__ float tile[32][33];
  const int lay = blockIdx.z;
  const int R = d.dims[lay], C = d.dims[lay + 1];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (r0 >= R || c0 >= C) return;
  const float* in = d.W[lay];
  float* out = const_cast<float*>(d.WT[lay]);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (r0 + r < R && c0 + tx < C) tile[r][tx] = in[(size_t)(r0 + r) * C + c0 + tx];
  __syncthreads();
  for (int cc = ty; cc < 32; cc += 8)
    if (c0 + cc < C && r0 + tx < R) out[(size_t)(c0 + cc) * R + r0 + tx] = tile[tx][cc];
}

void gmpc_launch_mlp_transpose_all(const MlpDesc& d, hipStream_t s) {
  int rmax = 1, cmax = 1;
  for (int l = 0; l < d.L; ++l) {
    rmax = d.dims[l] > rmax ? d.dims[l] : rmax;
    cmax = d.dims[l + 1] > cmax ? d.dims[l + 1] : cmax;
  }
  hipLaunchKernelGGL(k_mlp_transpose_all, dim3((cmax + 31) / 32, (rmax + 31) / 32, d.L), dim3(256), 0, s, d);
}

int gmpc_head2_rows() {
  static const int g = [] {
    const char* e = getenv("GMPC_HEAD_G");
    // rows per workgroup = 4 G.  Measured (C3, 2048 sequences, head 3 x 256, alone): G = 2 0.049 ms, 3 0.056, 4 0.063 --
    // the layer chain is latency-bound per workgroup, more workgroups beat more rows per weight load
    const int v = e ? atoi(e) : 2;
    return v >= 2 && v <= 4 ? v : 2;
  }();
  return 4 * g;
}

void gmpc_launch_head2(int Bc, const CriticDesc& cd, int loss_kind, const float* hT, const float* label, float* score,
                       float* loss, float* acts, float* dels, float* plast, float* dhT, int act_stride,
                       hipStream_t s) {
  const int R = gmpc_head2_rows();
  const dim3 grid((Bc + R - 1) / R), blk(GMPC_THREADS);
  switch (R / 4) {
    case 2: hipLaunchKernelGGL(k_head2<2>, grid, blk, 0, s, Bc, cd, loss_kind, hT, label, score, loss, acts, dels, plast, dhT, act_stride); break;
    case 4: hipLaunchKernelGGL(k_head2<4>, grid, blk, 0, s, Bc, cd, loss_kind, hT, label, score, loss, acts, dels, plast, dhT, act_stride); break;
    default: hipLaunchKernelGGL(k_head2<3>, grid, blk, 0, s, Bc, cd, loss_kind, hT, label, score, loss, acts, dels, plast, dhT, act_stride); break;
  }
}
