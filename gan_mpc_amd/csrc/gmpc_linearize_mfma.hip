// Dynamics Jacobian chain on the matrix cores.
//
// For every sample (b,t) the n rows of  [A_t | B_t] - [I | 0] = W_L^T D_{L-1} W_{L-1}^T ... D_1 W_1^T
// are rows of one big GEMM chain whose operands are the SHARED weight matrices: stacking the rows
// of all B*T samples gives M = B*T*n rows (870,400 at the headline shape) against K = N = hidden
// width.  That is a true dense contraction, so it runs on v_mfma_f32_32x32x2_f32 (exact fp32: the
// result is the same k-ordered fmaf chain as the VALU kernel, MI355X_MICROARCH.md "FP32-input MFMA").
//
// One wavefront owns one 32-row tile for the whole chain:
//   - A operand (rows x k): the previous layer's masked tile, kept in the wave's private LDS slab
//     G[32][SK] (SK odd -> the column read of the A fragment and the row write of the accumulator
//     are both bank-conflict free); the seed tile is built on the fly from W_L and the relu bits.
//   - B operand (k x 32 columns): fragments of zero-padded transposed weights (K+2 rows x 32*NT
//     columns, built once per gmpc_set_params) loaded straight from L2/L1 into registers, one k-step
//     ahead of the MFMAs that consume them.  Padding makes the k-loop free of any conditional.
//   - C: NT accumulators of 16 registers (32 x 32*NT columns), masked with the relu bits of the
//     layer below in the epilogue and written back to G as the next A operand.
// Waves never talk to each other: no workgroup barrier in the kernel.
#include "gmpc_device.h"



// out[o][i] = W[i][o] (W is (in,out) row-major) into a zero-initialised (rows x ldo) buffer
__global__ void k_pad_transpose(int in, int out, const float* W, float* dst, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= in * out) return;
  const int i = e / out, o = e - i * out;
  dst[(size_t)o * ldo + i] = W[e];
}

#define GMPC_LIN_PADROWS 24   // zero rows after the last k row of every padded operand

// hidden layers: dst[o][l31][nt] = W[i = nt*32 + l31][o]  (lane-interleaved, 8 slots per lane)
__global__ void k_pad_transpose_x4(int in, int out, const float* W, float* dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= in * out) return;
  const int i = e / out, o = e - i * out;
  dst[(size_t)o * 256 + (i & 31) * 8 + (i >> 5)] = W[e];
}

template <int NT, int NTF>
__global__ __launch_bounds__(GMPC_THREADS, 1) void k_linearize_mfma(
    int NSamp, int T, int n, int m, MlpDesc dyn, LinPad lp, const uint32_t* masks, const int* active,
    float* AB, int ntiles, int nsmax, int stage_wl, int stage_w1, int samp_mul, int samp_add) {
  // sample s of this launch is sample s*samp_mul + samp_add of the masks / trajectory s_id / T of
  // `active` (the whole batch: mul 1, add 0; one time step t of every trajectory: mul T, add t)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SK = 32 * NT + 1;
  constexpr int NGW = 32 * NTF;                 // columns per group of the input GEMM
  constexpr bool ONEG = NTF <= 3;               // NTF <= 3 always means a single group
  const int NGF = ONEG ? 1 : lp.NGF;
  const int NPF = ONEG ? NGW : NGW * lp.NGF;    // padded row length of W_1^T (compile-time if ONEG)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lh = dyn.L - 1;
  const int nm = n + m;
  const int Rtot = NSamp * n;
  const size_t wave_bytes =
      (size_t)(32 * SK + 8) * sizeof(float) + (size_t)Lh * nsmax * GMPC_MW * sizeof(uint32_t);
  float* G = reinterpret_cast<float*>(smem + wave * wave_bytes);
  uint32_t* mk = reinterpret_cast<uint32_t*>(G + 32 * SK + 8);   // [Lh][nsmax][MW]
  // workgroup-shared copies of the two small operands (when they fit): W_L and the padded W_1^T
  float* sh = reinterpret_cast<float*>(smem + (GMPC_THREADS / 64) * wave_bytes);
  const int wl_floats = (dyn.dims[Lh] + GMPC_LIN_PADROWS) * n;
  const int w1_floats = (dyn.dims[1] + GMPC_LIN_PADROWS) * NPF;   // staged only when NGF == 1
  float* wl_s = sh;
  float* w1_s = sh + (stage_wl ? wl_floats : 0);
  if (stage_wl)
    for (int e = threadIdx.x; e < wl_floats; e += blockDim.x) wl_s[e] = lp.WLP[e];
  if (stage_w1)
    for (int e = threadIdx.x; e < w1_floats; e += blockDim.x) w1_s[e] = lp.WTP[0][e];
  __syncthreads();   // the only workgroup barrier: after it the waves are independent
  const int half = lane >> 5, l31 = lane & 31;
  // diagnostic stamps (GMPC_LIN_STAMPS=1): cycles per segment summed over tiles; slots: 0 staging +
  // prologue, 1 hidden GEMMs, 2 hidden epilogues, 3 input GEMM, 4 output stores
  unsigned long long st[5] = {0, 0, 0, 0, 0}, tprev = 0;
  const bool stamps = lp.dbg != nullptr;
#define GMPC_STAMP(i) if (stamps) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - tprev; tprev = t_; }
  if (stamps) tprev = __builtin_readcyclecounter();

  for (int tile = blockIdx.x * (GMPC_THREADS / 64) + wave; tile < ntiles;
       tile += gridDim.x * (GMPC_THREADS / 64)) {
    const int r0 = tile * 32;                 // B*T*n < 2^31 (checked by the launcher)
    const int s_lo = r0 / n;
    const int rem0 = r0 - s_lo * n;           // output coordinate of the tile's first row
    int rlast = r0 + 31;
    if (rlast >= Rtot) rlast = Rtot - 1;
    const int s_hi = rlast / n;
    const int ns = s_hi - s_lo + 1;
    if (active != nullptr) {
      bool any = false;
      for (int s = s_lo; s <= s_hi; ++s) any |= active[(s * samp_mul + samp_add) / T] != 0;
      if (!any) continue;
    }
    // ---- relu bit words of the tile's samples -> LDS
    for (int e = lane; e < Lh * ns * GMPC_MW; e += 64) {
      const int l = e / (ns * GMPC_MW), rem = e - l * ns * GMPC_MW;
      const int si = rem / GMPC_MW, w = rem - si * GMPC_MW;
      const size_t sid = (size_t)(s_lo + si) * samp_mul + samp_add;
      mk[(l * nsmax + si) * GMPC_MW + w] = masks[(sid * Lh + l) * GMPC_MW + w];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // this lane's A-operand row (row l31 of the tile): its sample and output coordinate
    int sa = (rem0 + l31) / n;
    if (sa > ns - 1) sa = ns - 1;              // rows past the end of the batch: clamp (discarded)
    int ia = rem0 + l31 - sa * n;
    if (ia > n - 1) ia = n - 1;
    // A element of the seed tile: W_L[k][i_row] * relu bit(layer Lh-1, sample of row, k)
    const float* wl_g = lp.WLP + (size_t)half * n + ia;
    const float* wl_l = wl_s + half * n + ia;
    const float* aptr = G + l31 * SK + half;
    // relu bits of this lane's row for the layer whose output is being consumed (set per GEMM)
    const uint32_t* mrow = mk + ((Lh - 1) * nsmax + sa) * GMPC_MW;
    // raw A loads (value + relu word of its k column) and the finishing select
    auto l_seed_g = [&](int k0) -> ARaw {
      const int k = k0 + half;
      return ARaw{wl_g[(size_t)k0 * n], mrow[(k >> 5) & (GMPC_MW - 1)]};
    };
    auto l_seed_l = [&](int k0) -> ARaw {
      const int k = k0 + half;
      return ARaw{wl_l[k0 * n], mrow[(k >> 5) & (GMPC_MW - 1)]};
    };
    // from the slab: the unmasked accumulator of the previous GEMM; its relu bit is applied when the
    // operand is consumed -- in the shadow of the MFMAs, not in the epilogue
    auto l_slab = [&](int k0) -> ARaw {
      const int k = k0 + half;
      return ARaw{aptr[k0], mrow[(k >> 5) & (GMPC_MW - 1)]};
    };
    auto a_fin = [&](const ARaw& r, int k0) -> float {
      return ((r.w >> ((k0 + half) & 31)) & 1u) ? r.v : 0.f;
    };
    // plain float sources for the generic gemm_tile (multi-tile input GEMMs, single hidden layer)
    auto a_seed_g = [&](int k0) -> float { return a_fin(l_seed_g(k0), k0); };
    auto a_seed_l = [&](int k0) -> float { return a_fin(l_seed_l(k0), k0); };
    auto a_slab = [&](int k0) -> float { return a_fin(l_slab(k0), k0); };

    GMPC_STAMP(0)
    f32x16 acc[NT];
    // ================= hidden GEMMs: l = Lh-1 (seeded from W_L) ... 1 =================
    for (int l = Lh - 1; l >= 1; --l) {
      const int K = dyn.dims[l + 1];
      const int Kp = (K + 1) & ~1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[nt][rg] = 0.f;
      // lane-interleaved padded W_l^T: this lane's 8 column-tile slots of row `half`
      const float4* __restrict__ bp0 =
          reinterpret_cast<const float4*>(lp.WTP[l]) + half * 64 + l31 * 2;
      // the A operand of this GEMM carries the relu bits of hidden layer l (its k index)
      mrow = mk + (l * nsmax + sa) * GMPC_MW;
      if (l == Lh - 1) {
        if (stage_wl) gemm_tile_x4<NT>(bp0, Kp, l_seed_l, a_fin, acc);
        else gemm_tile_x4<NT>(bp0, Kp, l_seed_g, a_fin, acc);
      } else {
        gemm_tile_x4<NT>(bp0, Kp, l_slab, a_fin, acc);
      }
      GMPC_STAMP(1)
      // ---- epilogue: the raw accumulator becomes the next A operand (masked when it is read).
      // Columns >= dims[l] are exact zeros (zero-padded B), which provides the zero pad columns the
      // next GEMM's read-ahead and an odd K need.
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
          G[row * SK + nt * 32 + l31] = acc[nt][rg];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      GMPC_STAMP(2)
    }
    // ================= input GEMM (l = 0): N = n + m columns in NGF groups of NTF tiles ===========
    for (int cg = 0; cg < NGF; ++cg) {
      const int K = dyn.dims[1];
      const int Kp = (K + 1) & ~1;
      f32x16 acc0[NTF];
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc0[nt][rg] = 0.f;
      const float* bp_g = lp.WTP[0] + (size_t)half * NPF + cg * NGW + l31;
      const float* bp_l = w1_s + half * NPF + l31;      // staged only when NGF == 1
      mrow = mk + (0 * nsmax + sa) * GMPC_MW;   // relu bits of hidden layer 0 (k index of this GEMM)
      if (Lh == 1) {
        // single hidden layer: the seed tile feeds the input GEMM directly
        if (stage_w1) {
          if (stage_wl) gemm_tile<NTF>(bp_l, NPF, Kp, a_seed_l, acc0);
          else gemm_tile<NTF>(bp_l, NPF, Kp, a_seed_g, acc0);
        } else {
          if (stage_wl) gemm_tile<NTF>(bp_g, NPF, Kp, a_seed_l, acc0);
          else gemm_tile<NTF>(bp_g, NPF, Kp, a_seed_g, acc0);
        }
      } else if (NTF == 1) {
        // one MFMA per k-step: deep operand ring; k-steps past Kp multiply zeros (A clamped to 0,
        // B rows zero-padded)
        auto fin_deep = [&](const ARaw& r, int k0) -> float { return k0 < Kp ? a_fin(r, k0) : 0.f; };
        if (stage_w1) {
          auto b_l = [&](int k0) -> float { return bp_l[k0 * NPF]; };
          gemm_tile_1(Kp, l_slab, fin_deep, b_l, acc0[0]);
        } else {
          auto b_g = [&](int k0) -> float { return bp_g[(size_t)k0 * NPF]; };
          gemm_tile_1(Kp, l_slab, fin_deep, b_g, acc0[0]);
        }
      } else {
        if (stage_w1) gemm_tile<NTF>(bp_l, NPF, Kp, a_slab, acc0);
        else gemm_tile<NTF>(bp_g, NPF, Kp, a_slab, acc0);
      }
      GMPC_STAMP(3)
#pragma unroll
      for (int nt = 0; nt < NTF; ++nt) {
        const int c = cg * NGW + nt * 32 + l31;
        if (c < nm) {
#pragma unroll
          for (int rg = 0; rg < 16; ++rg) {
            const int o = (rg & 3) + 8 * (rg >> 2) + 4 * half;
            const int rr = r0 + o;
            if (rr < Rtot) {
              const int irow = (rem0 + o) % n;   // output coordinate of this accumulator row
              AB[(size_t)rr * nm + c] = acc0[nt][rg] + (c == irow ? 1.0f : 0.0f);
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    GMPC_STAMP(4)
  }
  if (stamps && lane == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&lp.dbg[i], st[i]);
#undef GMPC_STAMP
}

// ---------------------------------------------------------------------------------------------
size_t gmpc_linpad_floats(const gmpc_shape* sh) {
  const int L = sh->dyn_layers, Lh = L - 1;
  int wmax = 1;
  for (int l = 1; l < L; ++l) wmax = sh->dyn_dims[l] > wmax ? sh->dyn_dims[l] : wmax;
  const int tilesF = (sh->n + sh->m + 31) / 32;
  const int NTF = tilesF <= 3 ? tilesF : 8, NGF = (tilesF + NTF - 1) / NTF;
  size_t f = (size_t)(sh->dyn_dims[Lh] + GMPC_LIN_PADROWS) * sh->n;
  f += (size_t)(sh->dyn_dims[1] + GMPC_LIN_PADROWS) * 32 * NTF * NGF;
  for (int l = 1; l < Lh; ++l) f += (size_t)(sh->dyn_dims[l + 1] + GMPC_LIN_PADROWS) * 256;
  return f + 64;
}

// builds the padded copies; `pad` must hold gmpc_linpad_floats() floats
void gmpc_linpad_prepare(const MlpDesc& dyn, int n, int m, float* pad, size_t pad_floats, LinPad* out,
                         hipStream_t s) {
  const int Lh = dyn.L - 1;
  int wmax = 1;
  for (int l = 1; l < dyn.L; ++l) wmax = dyn.dims[l] > wmax ? dyn.dims[l] : wmax;
  out->NT = (wmax + 31) / 32;
  const int tilesF = (n + m + 31) / 32;
  out->NTF = tilesF <= 3 ? tilesF : 8;
  out->NGF = (tilesF + out->NTF - 1) / out->NTF;
  out->dbg = nullptr;
  (void)hipMemsetAsync(pad, 0, pad_floats * sizeof(float), s);
  float* p = pad;
  // W_L: (dims[Lh] x n) row-major already is [k][i]; copy, padded rows stay zero
  out->WLP = p;
  (void)hipMemcpyAsync(p, dyn.W[Lh], (size_t)dyn.dims[Lh] * n * sizeof(float), hipMemcpyDeviceToDevice, s);
  p += (size_t)(dyn.dims[Lh] + GMPC_LIN_PADROWS) * n;
  for (int l = 0; l < Lh; ++l) {
    const int in = dyn.dims[l], outd = dyn.dims[l + 1];
    const int ldo = (l == 0) ? 32 * out->NTF * out->NGF : 256;
    out->WTP[l] = p;
    const int cnt = in * outd;
    if (l == 0)
      hipLaunchKernelGGL(k_pad_transpose, dim3((cnt + 255) / 256), dim3(256), 0, s, in, outd, dyn.W[l],
                         p, ldo);
    else
      hipLaunchKernelGGL(k_pad_transpose_x4, dim3((cnt + 255) / 256), dim3(256), 0, s, in, outd,
                         dyn.W[l], p);
    p += (size_t)(outd + GMPC_LIN_PADROWS) * ldo;
  }
}

template <int NT, int NTF>
static int launch_mfma(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                       const uint32_t* masks, const int* active, float* AB, int samp_mul, int samp_add,
                       hipStream_t s) {
  const long Rtot = (long)NSamp * n;
  if (Rtot >= (1L << 31) - 64) return -1;
  const int ntiles = (int)((Rtot + 31) / 32);
  const int nsmax = 32 / n + 2;
  const int Lh = dyn.L - 1;
  const size_t wave_bytes = (size_t)(32 * (32 * NT + 1) + 8) * sizeof(float) +
                            (size_t)Lh * nsmax * GMPC_MW * sizeof(uint32_t);
  size_t lds = wave_bytes * (GMPC_THREADS / 64);
  const size_t lds_max = 160 * 1024;
  if (lds > lds_max) return -1;
  // stage the two small operands in LDS when they fit (W_1^T first: its GEMM has one MFMA per
  // k-step and is latency-bound from L2)
  const size_t w1_bytes = (size_t)(dyn.dims[1] + GMPC_LIN_PADROWS) * 32 * NTF * lp.NGF * sizeof(float);
  const size_t wl_bytes = (size_t)(dyn.dims[Lh] + GMPC_LIN_PADROWS) * n * sizeof(float);
  int stage_w1 = 0, stage_wl = 0;
  if (lp.NGF == 1 && lds + w1_bytes <= lds_max) { stage_w1 = 1; lds += w1_bytes; }
  if (lds + wl_bytes <= lds_max) { stage_wl = 1; lds += wl_bytes; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linearize_mfma<NT, NTF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr_set = true;
  }
  int grid = (ntiles + 3) / 4;
  if (grid > 256) grid = 256;   // one persistent workgroup per CU (LDS-limited to 1 anyway)
  hipLaunchKernelGGL((k_linearize_mfma<NT, NTF>), dim3(grid), dim3(GMPC_THREADS), lds, s, NSamp, T, n, m,
                     dyn, lp, masks, active, AB, ntiles, nsmax, stage_wl, stage_w1, samp_mul, samp_add);
  return 0;
}

// Jacobians of NSamp samples; sample s is sample s*samp_mul + samp_add of `masks`.
// returns 0 on launch, -1 if the shape does not fit this kernel (caller uses the VALU chain)
int gmpc_launch_linearize_mfma(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                               const uint32_t* masks, const int* active, float* AB, int samp_mul,
                               int samp_add, hipStream_t s) {
#define GMPC_LM(a, b)                                                                              \
  if (lp.NT == a && lp.NTF == b)                                                                   \
    return launch_mfma<a, b>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s);
  GMPC_LM(1, 1) GMPC_LM(2, 1) GMPC_LM(3, 1) GMPC_LM(4, 1) GMPC_LM(5, 1) GMPC_LM(6, 1) GMPC_LM(7, 1)
  GMPC_LM(8, 1) GMPC_LM(1, 2) GMPC_LM(2, 2) GMPC_LM(3, 2) GMPC_LM(4, 2) GMPC_LM(5, 2) GMPC_LM(6, 2)
  GMPC_LM(7, 2) GMPC_LM(8, 2) GMPC_LM(1, 3) GMPC_LM(2, 3) GMPC_LM(3, 3) GMPC_LM(4, 3) GMPC_LM(5, 3)
  GMPC_LM(6, 3) GMPC_LM(7, 3) GMPC_LM(8, 3) GMPC_LM(1, 8) GMPC_LM(2, 8) GMPC_LM(3, 8) GMPC_LM(4, 8)
  GMPC_LM(5, 8) GMPC_LM(6, 8) GMPC_LM(7, 8) GMPC_LM(8, 8)
#undef GMPC_LM
  return -1;
}
