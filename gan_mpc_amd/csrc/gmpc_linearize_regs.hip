// Dynamics Jacobian chain on the matrix cores, register-resident variant (gfx950 only).
//
// Same product as gmpc_linearize_mfma.hip -- [A_t | B_t] - [I | 0] = W_L^T D_{L-1} W_{L-1}^T ... D_1 W_1^T
// for the stacked Jacobian rows of all samples -- but transposed: the stacked rows are the COLUMNS
// of the MFMA tiles (one per lane), the weights are the A operand and the running product S_l
// (hidden x 32 rows) never leaves the register file:
//   S_{L-1}[k][r] = W_L[k][i_r] * relu bit(sample_r, k)                                  (seed)
//   S_{l-1}       = D_{l-1} (W_l S_l)        l = L-1 .. 2       (NT x KS MFMAs per tile and layer)
//   out           = W_1 S_1                                     (KS MFMAs)
// An accumulator tile holds (rows = hidden unit, lanes = stacked row); the next GEMM wants it as B
// operand (lane halves = two consecutive k rows).  v_permlane32_swap_b32 (new on gfx950) exchanges
// the upper half of one register with the lower half of its neighbour, which is exactly that
// re-pairing: 8 swaps turn a 16-register accumulator tile into 16 B operands, so the chain needs no
// LDS round trip, no relu select in the k-loop (the bits are applied once per accumulator) and no
// barrier.  Per k-step a wave issues two 16-byte weight loads (lane-interleaved copies shared with
// the LDS variant) and NT MFMAs.  Measured: 69.5 cycles per 32x32x2 issue for this inner loop against
// 72.9 (micro-benchmark) / 82 (in the kernel) for the LDS-operand variant.
//
// Compile-time shape: NT row tiles (hidden width <= 32 NT) and KS k-steps (hidden width = 2 KS or
// 2 KS - 1), all hidden layers of the same width, n + m <= 32.  Everything else runs the LDS variant.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gmpc_device.h"
#ifndef GMPC_LIN_RD0
#define GMPC_LIN_RD0 3      // k-steps of W_1^T fragments in flight + 1 in the wide form's input GEMM (5 or 8: C4 97 vs 91 ms)
#endif

typedef unsigned v2u __attribute__((ext_vector_type(2)));

#define GMPC_LIN_PADROWS 24
// waves per SIMD: the 200-wide instantiation fits 256 registers, and with two waves per SIMD the
// seeds, epilogues and stores of one wave overlap the other's MFMAs (lqr_backward 1.539 -> 1.507 ms);
// the other shapes keep one wave and its 512 registers
#define GMPC_REGS_OCC(NT, TAIL) (((NT) == 6 && (TAIL) == 8) ? 2 : 1)
#ifndef GMPC_REGS_RING
#define GMPC_REGS_RING 3
#endif

__device__ __forceinline__ void swap_halves(float& a, float& b) {
  // a = (a.lower | b.lower), b = (a.upper | b.upper)
  const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// TAIL = 8: the hidden width is 32 NT + 8 (200 = 6 x 32 + 8).  A seventh 32-row tile would be three
// quarters padding; instead the last 8 rows go through v_mfma_f32_4x4x1_16B_f32.  Its 16 blocks of
// 4 x 4 see S[ks] exactly as it lies in the registers: blocks 0..7 (lanes 0..31) hold k = 2 ks for the
// columns 4 blk .. 4 blk + 3, blocks 8..15 the same columns for k = 2 ks + 1.  With the A operand
// W[row r][k = 2 ks + half] the lower lane half accumulates the even-k partial sum and the upper half the
// odd-k one; the halves are added once per layer in the epilogue.  Two issues per k-step (rows 0..3 and
// 4..7, 8 cycles each), no operand shuffling in the loop (a v_permlane32_swap per k-step measured 32
// cycles of a 490-cycle k-step), the A operands -- 8 weights per k -- come from a small LDS table.
// WIDE: n + m > 32 (the large-state path, one time step per launch).  W_L and W_1^T no longer fit LDS: the
// seed comes straight from the padded global copy of W_L, the input GEMM runs over the column tiles of
// [A | B] four at a time with its weight operands through buffer loads, and the accumulator tiles -- lanes are
// stacked rows here -- are transposed through a wave-private LDS tile so that the rows of AB are written in
// 128-byte segments.
// REST: the same kernel under a second name -- the launch that runs the ragged last round on its own (launch_regs), so
// that a kernel trace lists it on a row of its own
// LIST: the samples are those of the trajectories tlist[0 .. *tcount) (T each, NSamp = capacity x T): the launch that
// runs the chain of a compacted subset of a batch (gmpc_ilqr_solve's early chain) with balanced tiles; rows of AB are
// written at the trajectory's own place.  Its workgroups are EIGHT waves (two per SIMD, 256 registers each: the whole
// register file of a CU), so that `grid` workgroups occupy exactly `grid` CUs and leave every other CU entirely to
// the kernel they run beside (k_ls16 needs a whole CU per workgroup).
// (LIST = 2: the same through ordinary four-wave workgroups -- the chain of "everything else", balanced because the
// tiles are those of the listed trajectories only: with flags instead, a wave's share of the static split is
// whatever its tiles happen to hold)
template <int NT, int KS, int TAIL = 0, bool WIDE = false, bool REST = false, int LIST = 0>
__global__ __launch_bounds__(LIST == 1 ? 2 * GMPC_THREADS : GMPC_THREADS, GMPC_REGS_OCC(NT, TAIL)) void k_linearize_regs(
    int NSamp, int T, int n, int m, MlpDesc dyn, LinPad lp, const uint32_t* masks, const int* active,
    float* AB, int ntiles, int samp_mul, int samp_add, int tile0, const int* tlist = nullptr,
    const int* tcount = nullptr) {
  static_assert(NT <= 8 && 2 * KS <= 32 * NT + TAIL && (TAIL == 0 || TAIL == 8), "shape");
  constexpr bool AG = GMPC_REGS_OCC(NT, TAIL) == 1;      // register half of the accumulators (mfma_fence)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl_s = reinterpret_cast<float*>(smem);          // W_L  [(H + pad)][n]
  int lcount = 0;
  if constexpr (LIST != 0) {
    lcount = *tcount;
    if (lcount <= 0) return;                             // (nothing was handed over this time)
    ntiles = min(ntiles, (int)(((long)lcount * T * n + 31) / 32));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int Lh = dyn.L - 1, nm = n + m;
  const int Rtot = NSamp * n;
  const int wl_floats = WIDE ? 4 * 32 * 33 : (dyn.dims[Lh] + GMPC_LIN_PADROWS) * n;   // WIDE: transpose tiles
  if (!WIDE)
    for (int e = threadIdx.x; e < wl_floats; e += blockDim.x) wl_s[e] = lp.WLP[e];
  // tail rows 32 NT .. 32 NT + 7 of every hidden W_l^T: wt_s[l - 1][k][8]
  float* wt_s = wl_s + wl_floats;
  // the padded W_1^T (2 KS x 32) of the input GEMM: one MFMA per k-step cannot hide an L2 round trip
  // per operand, an LDS read it can
  float* w1_s = wt_s + (TAIL > 0 ? (Lh - 1) * 2 * KS * 8 : 0);
  if (!WIDE)
    for (int e = threadIdx.x; e < 2 * KS * 32; e += blockDim.x) w1_s[e] = lp.WTP[0][e];
  if (TAIL > 0) {
    for (int l = 1; l < Lh; ++l)
      for (int e = threadIdx.x; e < 2 * KS * 8; e += blockDim.x) {
        const int k = e >> 3, r = e & 7;
        // lane-interleaved source: row k of W_l^T, slot of input row 32 NT + r
        wt_s[(l - 1) * 2 * KS * 8 + e] = lp.WTP[l][(size_t)k * 256 + r * 8 + NT];
      }
  }
  __syncthreads();   // the only workgroup barrier
  // diagnostic stamps (GMPC_LIN_STAMPS=1): 0 seed, 1 hidden GEMMs, 2 epilogues, 3 input GEMM, 4 stores
  unsigned long long st[5] = {0, 0, 0, 0, 0}, tprev = 0;
  const bool stamps = lp.dbg != nullptr;
#define GMPC_STAMP(i) if (stamps) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - tprev; tprev = t_; }
  if (stamps) tprev = __builtin_readcyclecounter();

  // (static split: 27,200 tiles over 2,048 waves are 13.28 per wave, 14 rounds.  Handing the tiles out through a
  // global ticket counter -- requested before the input GEMM, read after the stores, so its round trip is off the
  // critical path -- was measured in round 3 and is not kept: 1.121 vs 1.107 ms on the same box.)
  // (tile0: first tile of this launch -- the ragged last round of a long tile list can be a launch of its own, so
  // that a caller's event between the two lets other streams use the wave slots the last round leaves idle)
  constexpr int WPB = (LIST == 1 ? 2 * GMPC_THREADS : GMPC_THREADS) / 64;      // waves per workgroup
  for (int tile = tile0 + blockIdx.x * WPB + wave; tile < ntiles; tile += gridDim.x * WPB) {
    const int r0 = tile * 32;
    int R = r0 + l31;                          // this lane's stacked Jacobian row
    const bool rvalid = R < Rtot;
    if (!rvalid) R = Rtot - 1;                 // clamped reads, no writes
    const int s = R / n, irow = R - s * n;
    size_t sid = (size_t)s * samp_mul + samp_add;
    int Rout = R;
    bool lvalid = rvalid;
    if constexpr (LIST != 0) {
      const int tb = s / T, tq = s - tb * T;
      lvalid = rvalid && tb < lcount;
      if (__ballot(lvalid) == 0ull) continue;
      sid = (size_t)tlist[min(tb, lcount - 1)] * T + tq;
      Rout = (int)sid * n + irow;
    }
    if (active != nullptr) {
      const bool on = active[sid / T] != 0;
      if (__ballot(on && rvalid) == 0ull) continue;
    }
    const uint32_t* mrow = masks + sid * Lh * GMPC_MW;

    // ---- seed: S[ks] = W_L[2 ks + half][irow] * relu bit of hidden layer Lh-1
    float S[16 * NT + TAIL / 2];
    {
      uint32_t mw[NT + 1];
#pragma unroll
      for (int w = 0; w < NT + (TAIL > 0 ? 1 : 0); ++w) mw[w] = mrow[(Lh - 1) * GMPC_MW + w] >> half;
      const float* wl = (WIDE ? lp.WLP : wl_s) + half * n + irow;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float v = wl[(size_t)2 * ks * n];
        // (bit -> all-ones word -> AND: two vector instructions per element.  As a `bit ? v : 0` select hipcc built
        // the 100 lane masks first -- 200 scalar registers, spilled through v_writelane / v_readlane -- and selected
        // afterwards: seven vector-port instructions per element, on the port the fp32 MFMAs issue through)
        // (the extraction is inline assembly because the optimiser turns and(v, sext(bit)) back into that select)
#ifdef GMPC_LIN_SEED_SELECT
        S[ks] = ((mw[(2 * ks) >> 5] >> ((2 * ks) & 31)) & 1u) ? v : 0.f;
#else
        unsigned full, vb = __float_as_uint(v);
        asm("v_bfe_i32 %1, %2, %3, 1\n\tv_and_b32 %0, %0, %1"
            : "+v"(vb), "=&v"(full)
            : "v"(mw[(2 * ks) >> 5]), "n"((2 * ks) & 31));
        S[ks] = __uint_as_float(vb);
#endif
      }
    }

    GMPC_STAMP(0)
    // ---- hidden GEMMs  S_{l-1} = D_{l-1} (W_l S_l),  l = Lh-1 .. 1
    for (int l = Lh - 1; l >= 1; --l) {
      f32x16 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[nt][rg] = 0.f;
      // weight operands through a buffer resource: the k-step offset is an SGPR / immediate, the lane
      // offset one fixed VGPR -- no 64-bit VALU address arithmetic between the MFMAs (measured: the
      // flat-address version spent 40 cycles of a 470-cycle k-step on its loads)
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(lp.WTP[l]), 0, 2 * KS * 256 * (int)sizeof(float), 0x00020000);
      const int woff = (half * 64 + l31 * 2) * 16;
      auto wload = [&](int ks_, int j_) -> float4 {
        const v4u r = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff + 16 * j_, ks_ * 2048, 0);
        return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z),
                           __uint_as_float(r.w));
      };
      constexpr int RD = GMPC_REGS_RING;      // operand ring: RD - 1 k-steps of weight loads in flight
      float4 w[RD][2];
#pragma unroll
      for (int j = 0; j < RD - 1; ++j) { w[j][0] = wload(j, 0); w[j][1] = wload(j, 1); }
      // relu words of the layer this GEMM produces (loaded early, used in the epilogue)
      uint32_t mw[NT + 1];
#pragma unroll
      for (int t_ = 0; t_ < NT; ++t_) mw[t_] = mrow[(l - 1) * GMPC_MW + t_] >> (4 * half);
      // two tail accumulators: rows 32 NT + {0..3} and 32 NT + {4..7}; the A operands are read from LDS
      // one k-step ahead
      f32x4 acct = {0.f, 0.f, 0.f, 0.f}, acct2 = {0.f, 0.f, 0.f, 0.f};
      const float* wtl = wt_s + (l - 1) * 2 * KS * 8 + 8 * half + (lane & 3);
      float wta = TAIL > 0 ? wtl[0] : 0.f, wtb = TAIL > 0 ? wtl[4] : 0.f;
      uint32_t mwt = 0;
      if (TAIL > 0) mwt = mrow[(l - 1) * GMPC_MW + NT] >> half;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + RD - 1 < KS) {
          w[(ks + RD - 1) % RD][0] = wload(ks + RD - 1, 0);
          w[(ks + RD - 1) % RD][1] = wload(ks + RD - 1, 1);
        }
        const float b = S[ks];
        const float4 q0 = w[ks % RD][0], q1 = w[ks % RD][1];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0.x, b, acc[0], 0, 0, 0);
        if (NT > 1) acc[1 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0.y, b, acc[1 % NT], 0, 0, 0);
        if (NT > 2) acc[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0.z, b, acc[2 % NT], 0, 0, 0);
        if (NT > 3) acc[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0.w, b, acc[3 % NT], 0, 0, 0);
        if (NT > 4) acc[4 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1.x, b, acc[4 % NT], 0, 0, 0);
        if (NT > 5) acc[5 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1.y, b, acc[5 % NT], 0, 0, 0);
        if (NT > 6) acc[6 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1.z, b, acc[6 % NT], 0, 0, 0);
        if (NT > 7) acc[7 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1.w, b, acc[7 % NT], 0, 0, 0);
        if (TAIL > 0) {
          const float a03 = wta, a47 = wtb;
          if (ks + 1 < KS) {
            wta = wtl[(2 * ks + 2) * 8];
            wtb = wtl[(2 * ks + 2) * 8 + 4];
          }
          acct = __builtin_amdgcn_mfma_f32_4x4x1f32(a03, b, acct, 0, 0, 0);
          acct2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a47, b, acct2, 0, 0, 0);
        }
        // spread the two weight loads and their address arithmetic BETWEEN the MFMAs of the k-step:
        // issued as a block at the k-step boundary they do not overlap the matrix pipe (one wave per
        // SIMD), see gemm_tile_x4
        // (round 2 kept the hints away from NT = 2, whose parity test failed with them: the reordered k-step ended on
        // the MFMA whose last register the epilogue reads first, 12 wait states later on the path that skips the
        // stamp block -- a stale read, not a scheduling defect; the fences below close it for every instantiation)
        {
#pragma unroll
          for (int i_ = 0; i_ < NT + (TAIL > 0 ? 2 : 0); ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // <= 1 VMEM read
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // <= 1 LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // <= 2 VALU
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   // <= 2 SALU
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      GMPC_STAMP(1)
      // the accumulators are first read behind the (skipped) stamp block: see mfma_fence in gmpc_device.h
      if constexpr (TAIL > 0) mfma_fence_tiles<AG>(acc, acct, acct2);
      else mfma_fence_tiles<AG>(acc);
      // epilogue: relu bits of hidden layer l-1 (rows of acc), then re-pair rows into B operands
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int rg = 0; rg < 16; rg += 2) {
          const int rho = (rg & 3) + 8 * (rg >> 2);              // row of the lower lanes of reg rg
          float a0 = ((mw[nt] >> rho) & 1u) ? acc[nt][rg] : 0.f;
          float a1 = ((mw[nt] >> (rho + 1)) & 1u) ? acc[nt][rg + 1] : 0.f;
          swap_halves(a0, a1);      // a0: rows (rho, rho+1); a1: rows (rho+4, rho+5)
          S[(nt * 32 + rho) / 2] = a0;
          S[(nt * 32 + rho + 4) / 2] = a1;
        }
      }
      if (TAIL > 0) {
        // tail: register j of acct / acct2 is row 32 NT + j / 32 NT + 4 + j of this lane's column, summed
        // over the k of this lane half's parity.  One swap per row pair adds the two parities and leaves
        // row 2 q in the lower and row 2 q + 1 in the upper half -- the B operand layout of S
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = q < 2 ? acct[2 * q] : acct2[2 * q - 4];
          float y = q < 2 ? acct[2 * q + 1] : acct2[2 * q - 3];
          swap_halves(x, y);      // x: (row 2q even-k | row 2q+1 even-k), y: the odd-k partial sums
          S[16 * NT + q] = ((mwt >> (2 * q)) & 1u) ? x + y : 0.f;
        }
      }
      GMPC_STAMP(2)
    }

    if constexpr (WIDE) {
      // ---- input GEMM over the column tiles of [A | B], CTW tiles per pass; rows of W_1^T (2 KS x ld0, zero
      // padded) through a buffer resource: k-step and tile offsets are SGPRs / immediates
      const int ld0 = 32 * lp.NTF * lp.NGF;
      const int tilesF = (nm + 31) / 32;
      const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(lp.WTP[0]), 0, (int)((size_t)(2 * KS + GMPC_LIN_PADROWS) * ld0 * sizeof(float)),
          0x00020000);
      const int voff = (half * ld0 + l31) * 4;
      float* tb = wl_s + wave * (32 * 33);                  // this wave's transpose tile
      // one pass: CT column tiles starting at tile t0 (tiles past the last one see the zero padding of W_1^T
      // and are not stored)
      auto pass = [&](auto ctc, int t0) __attribute__((always_inline)) {
        constexpr int CT = decltype(ctc)::value;
        f32x16 acc0[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int rg = 0; rg < 16; ++rg) acc0[j][rg] = 0.f;
        constexpr int RD0 = GMPC_LIN_RD0;
        float aw[RD0][CT];
        auto wload = [&](int ks_, int j_) -> float {
          const unsigned r = __builtin_amdgcn_raw_buffer_load_b32(w1r, voff, (2 * ks_ * ld0 + (t0 + j_) * 32) * 4, 0);
          return __uint_as_float(r);
        };
#pragma unroll
        for (int q = 0; q < RD0 - 1; ++q)
#pragma unroll
          for (int j = 0; j < CT; ++j) aw[q][j] = wload(q, j);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + RD0 - 1 < KS) {
#pragma unroll
            for (int j = 0; j < CT; ++j) aw[(ks + RD0 - 1) % RD0][j] = wload(ks + RD0 - 1, j);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < CT; ++j)
            acc0[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ks % RD0][j], S[ks], acc0[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        mfma_fence_tiles<AG>(acc0);
        // stores: accumulator tile j = 32 input coordinates (rows of the MFMA) x 32 stacked rows (lanes);
        // through the LDS tile it leaves as 32 stacked rows x 32 consecutive columns
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int c0 = (t0 + j) * 32;
          if (c0 < nm) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
              const int cl = (rg & 3) + 8 * (rg >> 2) + 4 * half;
              tb[l31 * 33 + cl] = acc0[j][rg] + (c0 + cl == irow ? 1.0f : 0.0f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int rr = 2 * i + half;
              const float v = tb[rr * 33 + l31];
              if (r0 + rr < Rtot && c0 + l31 < nm) AB[(size_t)(r0 + rr) * nm + c0 + l31] = v;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      };
      // passes of 4 tiles, the rest in passes of 3 when that wastes less (13 tiles = 4 + 3 + 3 + 3)
      int n4 = tilesF / 4;
      while (n4 > 0 && (tilesF - 4 * n4) % 3 != 0) --n4;
      if ((tilesF - 4 * n4) % 3 != 0) n4 = (tilesF + 3) / 4;      // no exact split: pad the last pass of 4
      int t0 = 0;
      for (int i = 0; i < n4; ++i, t0 += 4) pass(std::integral_constant<int, 4>{}, t0);
      for (; t0 < tilesF; t0 += 3) pass(std::integral_constant<int, 3>{}, t0);
      GMPC_STAMP(3)
    } else {
      // ---- input GEMM  out = W_1 S_1 : rows = input coordinate c (n + m <= 32), one MFMA per k-step
      f32x16 acc0;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) acc0[rg] = 0.f;
      {
        const float* ap = w1_s + half * 32 + l31;
        float a[6];
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] = ap[j * 64];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks + 5 < KS) a[(ks + 5) % 6] = ap[(ks + 5) * 64];
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks % 6], S[ks], acc0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      GMPC_STAMP(3)
      mfma_fence<AG>(acc0);
      if (LIST != 0 ? lvalid : rvalid) {
        float* dst = AB + (size_t)(LIST != 0 ? Rout : R) * nm;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int c = (rg & 3) + 8 * (rg >> 2) + 4 * half;
          if (c < nm) dst[c] = acc0[rg] + (c == irow ? 1.0f : 0.0f);
        }
      }
    }
    GMPC_STAMP(4)
  }
  if (stamps && lane == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&lp.dbg[i], st[i]);
#undef GMPC_STAMP
}

// name of the instantiation launched last (bench.py's roofline.kernel; matches the rocprofv3 kernel trace)
static char g_last_name[64] = "";
const char* gmpc_linearize_regs_last_name() { return g_last_name; }

// mid_event (optional): when the static split leaves a ragged last round -- less than GMPC_LIN_SPLIT_MAX of the wave
// slots busy for a whole tile time (C3: 27,200 tiles over 2,048 slots = 13 full rounds + 576 tiles, 28 %) -- the last
// round is launched as a kernel of its own and the event is recorded between the two: a stream waiting for it (the
// critic chain, bench.py) starts while the last tiles are still running, on the CUs and register halves they leave
// free.  Returns 1 when the event was recorded here, 0 when it was not (one launch), -1 on an unsupported shape.
template <int NT, int KS, int TAIL = 0, bool WIDE = false>
static int launch_regs(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                       const uint32_t* masks, const int* active, float* AB, int samp_mul, int samp_add,
                       hipStream_t s, hipEvent_t mid_event = nullptr) {
  const long Rtot = (long)NSamp * n;
  if (Rtot >= (1L << 31) - 64) return -1;
  const int ntiles = (int)((Rtot + 31) / 32);
  const int Lh = dyn.L - 1;
  size_t lds = WIDE ? (size_t)4 * 32 * 33 * sizeof(float)
                    : (size_t)(dyn.dims[Lh] + GMPC_LIN_PADROWS) * n * sizeof(float);
  if (TAIL > 0) lds += (size_t)(Lh - 1) * 2 * KS * 8 * sizeof(float);
  if (!WIDE) lds += (size_t)2 * KS * 32 * sizeof(float);
  if (lds > 64 * 1024) return -1;
  int grid = (ntiles + 3) / 4;
  constexpr int occ = GMPC_REGS_OCC(NT, TAIL);
  if (grid > 256 * occ) grid = 256 * occ;   // persistent workgroups, occ per CU
  // (measured, round 3: one workgroup per CU -- 256 of the 512 registers of every SIMD left to other kernels --
  // costs 5.5 % alone (1.141 -> 1.204 ms), and the critic's kernels beside it are starved by its back-to-back
  // 64-cycle MFMAs: k_head2 0.05 -> 0.37 ms, k_lstm_bwd2 0.11 -> 0.87 ms.  The chain keeps the chip to itself.)
  // (as rocprofv3's kernel trace prints the instantiation: REST = false, LIST = 0 for the single launch)
  snprintf(g_last_name, sizeof(g_last_name), "k_linearize_regs<%d, %d, %d, %s, false, 0>", NT, KS, TAIL,
           WIDE ? "true" : "false");
  const int slots = grid * 4;
  const int full = ntiles / slots, rest = ntiles - full * slots;
  static const int split_pct = []() {
    const char* e = getenv("GMPC_LIN_SPLIT");      // percent of a round below which the last round is split off; 0: never
    // (default 0 -- measured in round 4 and not kept: with the critic chain gated on the event between the two
    // launches the chain runs 0.04 ms longer (the last round's workgroups restage their LDS tables and share their
    // SIMDs with the critic's first kernel) and the step is 0.006 ms slower, not faster)
    return e != nullptr ? atoi(e) : 0;
  }();
  if (mid_event != nullptr && active == nullptr && full >= 2 && rest > 0 && rest * 100 < slots * split_pct) {
    hipLaunchKernelGGL((k_linearize_regs<NT, KS, TAIL, WIDE>), dim3(grid), dim3(GMPC_THREADS), lds, s, NSamp, T, n, m,
                       dyn, lp, masks, active, AB, full * slots, samp_mul, samp_add, 0);
    if (hipEventRecord(mid_event, s) != hipSuccess) return -1;
    hipLaunchKernelGGL((k_linearize_regs<NT, KS, TAIL, WIDE, !WIDE>), dim3((rest + 3) / 4), dim3(GMPC_THREADS), lds, s,
                       NSamp, T, n, m, dyn, lp, masks, active, AB, ntiles, samp_mul, samp_add, full * slots);
    return 1;
  }
  hipLaunchKernelGGL((k_linearize_regs<NT, KS, TAIL, WIDE>), dim3(grid), dim3(GMPC_THREADS), lds, s, NSamp, T, n, m,
                     dyn, lp, masks, active, AB, ntiles, samp_mul, samp_add, 0);
  return 0;
}

// the chain of the trajectories tlist[0 .. *tcount), at most `cap` of them.  whole_cus > 0: on that many workgroups of
// eight waves = that many whole CUs; 0: on the persistent four-wave workgroups of the ordinary launch
template <int NT, int KS, int TAIL = 0>
static int launch_regs_list(int cap, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp, const uint32_t* masks,
                            const int* tlist, const int* tcount, float* AB, int whole_cus, hipStream_t s) {
  const long Rtot = (long)cap * T * n;
  if (Rtot >= (1L << 31) - 64 || cap < 1) return -1;
  const int ntiles = (int)((Rtot + 31) / 32);
  const int Lh = dyn.L - 1;
  size_t lds = (size_t)(dyn.dims[Lh] + GMPC_LIN_PADROWS) * n * sizeof(float) + (size_t)2 * KS * 32 * sizeof(float);
  if (TAIL > 0) lds += (size_t)(Lh - 1) * 2 * KS * 8 * sizeof(float);
  if (lds > 64 * 1024) return -1;
  static_assert(GMPC_REGS_OCC(NT, TAIL) == 2, "eight waves of 256 registers");
  if (whole_cus > 0) {
    const int grid = std::min(whole_cus, (ntiles + 7) / 8);
    hipLaunchKernelGGL((k_linearize_regs<NT, KS, TAIL, false, false, 1>), dim3(grid), dim3(2 * GMPC_THREADS), lds, s,
                       cap * T, T, n, m, dyn, lp, masks, nullptr, AB, ntiles, 1, 0, 0, tlist, tcount);
  } else {
    const int grid = std::min(256 * GMPC_REGS_OCC(NT, TAIL), (ntiles + 3) / 4);
    hipLaunchKernelGGL((k_linearize_regs<NT, KS, TAIL, false, false, 2>), dim3(grid), dim3(GMPC_THREADS), lds, s,
                       cap * T, T, n, m, dyn, lp, masks, nullptr, AB, ntiles, 1, 0, 0, tlist, tcount);
  }
  return 0;
}
int gmpc_launch_linearize_regs_list(int cap, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                                    const uint32_t* masks, const int* tlist, const int* tcount, float* AB, int whole_cus,
                                    hipStream_t s) {
  const int H = dyn.dims[1];
  // (the 200-wide instantiation only: the one that runs two waves per SIMD, and the only width k_ls16 serves)
  if (H == 200 && lp.NT == 7)
    return launch_regs_list<6, 100, 8>(cap, T, n, m, dyn, lp, masks, tlist, tcount, AB, whole_cus, s);
  return -1;
}

// true when gmpc_launch_linearize_regs serves this (narrow) shape -- the same conditions as its dispatch below
bool gmpc_linearize_regs_covers(int n, int m, const MlpDesc& dyn, const LinPad& lp) {
  const int Lh = dyn.L - 1;
  if (Lh < 2 || n + m > 32 || lp.NTF != 1 || lp.NGF != 1) return false;
  const int H = dyn.dims[1];
  for (int l = 1; l <= Lh; ++l)
    if (dyn.dims[l] != H) return false;
  return (H == 200 && lp.NT == 7) || (H == 128 && lp.NT == 4) || (H == 64 && lp.NT == 2);
}

// returns 0 on launch (1: and mid_event was recorded between its two launches), -1 when the shape is not one this
// variant is compiled for
int gmpc_launch_linearize_regs(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                               const uint32_t* masks, const int* active, float* AB, int samp_mul,
                               int samp_add, hipStream_t s, hipEvent_t mid_event) {
  const int Lh = dyn.L - 1;
  if (Lh < 2) return -1;
  const int H = dyn.dims[1];
  for (int l = 1; l <= Lh; ++l)
    if (dyn.dims[l] != H) return -1;
  if (n + m > 32) {
    // wide inputs (large-state path): the 200-wide instantiation only
    static const bool off = getenv("GMPC_LIN_WIDE") != nullptr && getenv("GMPC_LIN_WIDE")[0] == '0';
    if (off || H != 200 || lp.NT != 7 || 32 * lp.NTF * lp.NGF < ((n + m + 31) / 32 + 3) / 4 * 4 * 32) return -1;
    return launch_regs<6, 100, 8, true>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s) < 0 ? -1 : 0;
  }
  if (lp.NTF != 1 || lp.NGF != 1) return -1;
  if (H == 200 && lp.NT == 7) {
    static const bool no_tail = getenv("GMPC_LIN_NOTAIL") != nullptr;
    if (!no_tail)
      return launch_regs<6, 100, 8>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s, mid_event);
    return launch_regs<7, 100>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s, mid_event);
  }
  if (H == 128 && lp.NT == 4)
    return launch_regs<4, 64>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s, mid_event);
  if (H == 64 && lp.NT == 2)
    return launch_regs<2, 32>(NSamp, T, n, m, dyn, lp, masks, active, AB, samp_mul, samp_add, s, mid_event);
  return -1;
}
