// Line-search candidates, 32 per workgroup: two groups of 16 handled by two TEAMS of four waves, half a time step
// apart (k_ls32<K0S, NOB>, 512 threads): the form for rounds whose work list holds more than one pass of k_ls16.
//
// k_ls16 (gmpc_ls16.hip) keeps the dynamics network's two 200 x 200 matrices in the registers of 4 waves (one per
// SIMD) and multiplies 16 candidates at a time on v_mfma_f32_16x16x4_f32.  A step is a chain of six phases --
// controls C, layer 0 L0, hidden layers H1 and H2, output layer O, state update U -- separated by workgroup barriers;
// only H1 and H2 keep the matrix pipe busy (163 of the 209 MFMAs of a wave and step, 5.6 k of their 6.7 k cycles
// each); C, L0, O and U are latency chains (LDS round trips, a few MFMAs, barriers): 6.1 k of the 19.5 k cycles of a
// step, and one wave per SIMD executes in order, so the pipe idles through them.
//
// Here a workgroup is EIGHT waves, two per SIMD, each with half the register file (256): team A (waves 0-3) holds
// layer 1's matrix and layer 0's, team B (waves 4-7) layer 2's.  A group's step is cut in two halves of equal
// matrix-pipe time, C L0 H1 (team A) and H2 O U (team B), and the two groups X, Y alternate between the teams:
//     half-step   2t            2t+1          2t+2
//     team A      X: C L0 H1    Y: C L0 H1    X(t+1) ...
//     team B      Y(t-1): H2 O U    X: H2 O U     Y: H2 O U
// The SIMD's scheduler interleaves its two waves instruction by instruction: while one team waits in a latency chain
// the other team's MFMAs issue.  No reliance on the compiler interleaving two instruction streams (the one-wave form of
// this kernel, round 4's first attempt: profiles/EXPERIMENTS.md).  Four barriers per half-step (the teams' phase
// boundaries are aligned: C | L0 | H1a | H1b against H2a | H2b | O | U).  Everything a candidate computes is what
// k_ls16 computes, operation for operation (same fragments, same accumulation order): results are bit-identical.
//
// Differences in layout against k_ls16 (LDS has to hold two groups): the A fragments of row block 12 and of output
// block 1 keep their non-zero lanes only, the K-split partials of block 12 their 32 useful lanes; the relu bits are
// bytes (plain stores) packed into words by the state update; the operands of a step's controls (gains, k, U, nominal
// state) are loaded by 16 lanes per candidate in the group's O phase and staged in its U phase.
//
// Reference arithmetic: dynamics/nn.py:27-34, cost/cost_model.py:20-42, cost/nn.py:23-29, trajax
// line_search_ddp / ddp_rollout (u = U + alpha k + K (x - X)) as called from policy/optimizers.py:19.
#include "gmpc_device.h"
#include <cstdlib>
#include <cstring>

typedef unsigned v2u __attribute__((ext_vector_type(2)));

#define LS32_THREADS 512         // two teams of 256
#define LS32_NG 2           // groups per workgroup
#define LS32_C 16           // candidates per group
#define LS32_KH 200         // hidden width
#define LS32_KS 50          // k-steps of a hidden layer
#define LS32_GS 80          // floats between groups of 4 activation rows
#define LS32_ROWS 208       // activation rows (13 blocks)
#define LS32_ACT ((LS32_ROWS / 4) * LS32_GS)
#define LS32_LDS_MAX (159 * 1024)   // (the kernel also holds 512 bytes of static LDS)

__device__ __forceinline__ f32x4_t ls32_mfma(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// float index of activation row k, candidate c
__device__ __forceinline__ int ls32_at(int k, int c) { return (k >> 2) * LS32_GS + (k & 3) * 16 + c; }

// epilogue of row block nb < 12 (the bias is in the accumulator): relu, the next layer's activations, and the relu
// bits of the block in the candidate's mask image of this layer -- 32 bytes per candidate and layer in the layout of
// the mask words that leave (bit r of the image = hidden unit r).  Lane (g, c) holds the four bits of rows
// 16 nb + 4 g .. + 3: v_permlane16_swap hands the nibble of the odd g to the even g beside it, which stores the byte
// 2 nb + g / 2.  (k_ls16 ORs nibbles into the mask word with LDS atomics; here the state update copies finished
// words.)  Vector instructions are what this kernel pays for -- the fp32 MFMAs issue through the same port -- so the
// relu is an integer max on the bit pattern (-0 and every negative: 0) and a bit is min(pattern, 1).
// outl / mbl: this lane's first activation slot and mask byte of the block (see ls32_blk)
__device__ __forceinline__ void ls32_epilogue(f32x4_t d, float* outl, unsigned char* mbl, int lane) {
  unsigned nib = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = max(__float_as_int(d[i]), 0);
    unsigned f;
    asm("v_min_u32 %0, 1, %1" : "=v"(f) : "v"(r));      // (as C the optimiser turns it into a compare and a select)
    nib |= f << i;
    outl[i * 16] = __int_as_float(r);
  }
  const v2u sw = __builtin_amdgcn_permlane16_swap(nib, nib, false, false);    // .y: row g <- row g + 1 (g even)
  if ((lane & 16) == 0) mbl[0] = (unsigned char)(nib | (sw.y << 4));
}
// A lane-dependent index as an opaque vector register: what is added to it in the code below is a compile-time
// constant and becomes the instruction's immediate offset.  Left visible, hipcc folds the wave's (scalar) part of an
// index into the constant first and then needs a v_add per access.
__device__ __forceinline__ int ls32_opq(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// rows 192 + g (.x) and 196 + g (.y) of a hidden layer's output for candidate lane & 15 -- the B fragments of
// k-steps 48 and 49 -- from the K-split partials of row block 12 ([4 waves][4 registers][32 lanes]); their relu bits
// are byte 24 of the candidate's mask image: bit g and bit 4 + g per lane, ORed over the four g with one
// v_permlane16_swap and one v_permlane32_swap (every wave stores the same byte)
__device__ __forceinline__ float2 ls32_tail(const float* p12, const float* bias192, unsigned char* mb, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const float* q = p12 + g * 32 + c;
  const float s0 = ((q[0] + q[128]) + (q[256] + q[384])) + bias192[g];
  const float s1 = ((q[16] + q[144]) + (q[272] + q[400])) + bias192[4 + g];
  const int r0 = max(__float_as_int(s0), 0), r1 = max(__float_as_int(s1), 0);
  const unsigned v = (min((unsigned)r0, 1u) | (min((unsigned)r1, 1u) << 4)) << g;
  const v2u a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const unsigned w = a.x | a.y;                                                // rows g, g ^ 1
  const v2u b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  if (lane < 16) mb[c * 96 + 24] = (unsigned char)(b.x | b.y);
  return make_float2(__int_as_float(r0), __int_as_float(r1));
}

// Global store without a branch: lanes that are not to store get an offset past the end of the buffer resource and the
// hardware drops their write (raw buffer, range-checked).  A phase of k_ls32 must stay ONE basic block -- an
// `if (lane has a role) store` is an exec-mask region that the matrix-pipe instructions of the other group's phase
// cannot be scheduled across.
__device__ __forceinline__ void ls32_store_if(__amdgpu_buffer_rsrc_t rs, unsigned elem, unsigned bits, bool on) {
  __builtin_amdgcn_raw_buffer_store_b32(bits, rs, on ? elem * 4u : 0xFFFFFFF0u, 0, 0);
}

// LDS layout in floats, compile-time per instantiation (every access is then one lane-dependent base register plus an
// immediate offset; with run-time group bases hipcc kept dozens of hoisted addresses live across the horizon and
// spilled): NV8 = rows of output block 1 kept per fragment (n - 16 <= 8), MNX = the largest gain block m n
template <int K0S, int NOB>
struct Ls32Lay {
  static constexpr int NV8 = NOB > 1 ? 8 : 0;
  static constexpr int MNX = K0S == 4 ? 64 : 128;
  // one group: its small buffers, then its two activation buffers -- ONE contiguous region per group, so that a
  // slot of the time loop can hand the two groups to the phases as two __restrict__ pointers (see `slots` in the kernel)
  static constexpr int XCUR = 0;                                  // rows x ; u ; 0 (layer-0 input), 8 groups of 4 rows
  static constexpr int DXS = XCUR + 8 * LS32_GS;                  // x - X_nominal in the layout of xcur
  static constexpr int PART = DXS + 8 * LS32_GS;                  // [4 waves][NOB][4][64] output-layer partials
  static constexpr int P12 = PART + 4 * NOB * 256;                // [2][4 waves][4][32] block-12 partials
  static constexpr int MASK = P12 + 2 * 4 * 4 * 32;               // [16][3 layers][32 bytes] mask image = [16][24] words
  static constexpr int KS = MASK + LS32_C * 24;                   // gains [16][MNX]
  static constexpr int KUS = KS + LS32_C * MNX;                   // k and U, [16][8] each
  static constexpr int DUMMY = KUS + 2 * LS32_C * 8;              // [64] where the stores of lanes without a role land
  static constexpr int ACTA = DUMMY + 64;
  static constexpr int ACTB = ACTA + LS32_ACT;                    // (after the horizon: the stage costs [16][T])
  static constexpr int GSZ = ACTB + LS32_ACT;                     // (group gi starts at gi * GSZ)
  // shared tables (read-only inside the time loop), relative to TB0
  static constexpr int TB0 = LS32_NG * GSZ;
  static constexpr int BIAS = 0;                                  // [3][208] hidden biases, [32] output bias
  static constexpr int WXL = BIAS + 3 * LS32_ROWS + 32;           // [2 layers][52 k-steps][4 g][8]: A fragments of block 12
  static constexpr int WOL = WXL + 2 * 52 * 32;                   // [4 waves][13][64] A fragments of output block 0
  static constexpr int WOL1 = WOL + 4 * 13 * 64;                  // [4 waves][13][4 g][NV8] A fragments of output block 1
  static constexpr int TOTAL = TB0 + WOL1 + 4 * 13 * 4 * NV8;
};

// K0S: k-steps of layer 0 (n + m <= 4 K0S); NOB: 16-row blocks of the output layer (n <= 16 NOB)
template <int K0S, int NOB>
__global__ __launch_bounds__(LS32_THREADS) void k_ls32(TrajArgs a, int min_items) {
  constexpr int NG = LS32_NG;
  extern __shared__ __attribute__((aligned(16))) char smem_ls32[];
  // tt: thread of its team; wave: wave of its team (the row blocks / k-steps it owns are those of k_ls16's wave)
  const int tid = threadIdx.x, tt = tid & 255, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tt >> 6);
  const int team = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int n = a.n, m = a.m, T = a.T, MN = m * n;
  const int NV = NOB > 1 ? n - 16 : 0;                     // valid rows of output block 1 (<= NV8)
  using LY = Ls32Lay<K0S, NOB>;
  constexpr int NV8 = LY::NV8;
  float* const smf = reinterpret_cast<float*>(smem_ls32);
  float* const tabs = smf + LY::TB0;
  float* const bias_s = tabs + LY::BIAS;
  float* const wxl = tabs + LY::WXL;
  float* const wol = tabs + LY::WOL;
  float* const wol1 = tabs + LY::WOL1;
  constexpr int GSZ = LY::GSZ;
  // one group's buffers
  struct Grp {
    float *base, *xcur, *dxs, *actA, *actB, *part, *p12, *Ks, *kUs;
    unsigned char* mask;
  };
  auto grp_at = [&](float* p) -> Grp {
    Grp G;
    G.base = p;
    G.xcur = p + LY::XCUR;
    G.dxs = p + LY::DXS;
    G.part = p + LY::PART;
    G.p12 = p + LY::P12;
    G.mask = reinterpret_cast<unsigned char*>(p + LY::MASK);
    G.Ks = p + LY::KS;
    G.kUs = p + LY::KUS;
    G.actA = p + LY::ACTA;
    G.actB = p + LY::ACTB;
    return G;
  };
  auto gbase = [&](int gi) -> float* { return smf + gi * GSZ; };
  auto grp = [&](int gi) -> Grp { return grp_at(gbase(gi)); };
  __shared__ float s_alpha[NG][LS32_C], s_obj[NG][LS32_C];
  __shared__ int s_bi[NG][LS32_C], s_in[NG][LS32_C];

  const int cnt = *a.nitems;
  if (cnt < min_items) return;                  // shorter work lists: k_ls16 / k_traj_rw (their launches return here)
  const int b0 = blockIdx.x * (NG * LS32_C);
  if (b0 >= cnt) return;
  if (tid < NG * LS32_C) {
    const int it = min(b0 + tid, cnt - 1);
    s_bi[tid >> 4][tid & 15] = a.item_b[it];
    s_in[tid >> 4][tid & 15] = (b0 + tid) < cnt;
    float al = a.alpha_0;
    for (int k = a.item_k[it]; k > 0; --k) al *= 0.5f;
    s_alpha[tid >> 4][tid & 15] = al;
  }
  constexpr int Lh = 3;
  const size_t mstride = (size_t)T * Lh * GMPC_MW;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  // ---- weights: registers for the whole horizon.  Team A: layer 1 (row blocks wave, wave + 4, wave + 8; block 12 is
  // split over the waves by k-step, k-steps wave + 4 j, fragments in LDS) and layer 0; team B: layer 2.
  float wr[3][LS32_KS];
  {
    const float* Wl = a.dyn.W[team + 1];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
      for (int ks = 0; ks < LS32_KS; ++ks) wr[r][ks] = Wl[(size_t)(4 * ks + g) * LS32_KH + nn];
    }
  }
  for (int hl = 0; hl < 2; ++hl) {
    const float* Wl = a.dyn.W[hl + 1];
    for (int e = tid; e < 52 * 32; e += LS32_THREADS) {
      const int ks = e >> 5, gg = (e >> 3) & 3, cc = e & 7;
      wxl[hl * 52 * 32 + e] = ks < LS32_KS ? Wl[(size_t)(4 * ks + gg) * LS32_KH + 192 + cc] : 0.f;
    }
  }
  // One register set, two uses: team A keeps layer 0's fragments in it (w0r[r][ks] = shr[r K0S + ks], block 12's
  // shr[3 K0S + q]), team B the operands of the next step's controls in flight between its O and U phases
  float shr[3 * K0S + 2];
#pragma unroll
  for (int e = 0; e < 3 * K0S + 2; ++e) shr[e] = 0.f;
  if (team == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
      for (int ks = 0; ks < K0S; ++ks) {
        const int k = 4 * ks + g;
        shr[r * K0S + ks] = k < n + m ? a.dyn.W[0][(size_t)k * LS32_KH + nn] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 4 * (wave + 4 * q) + g, nn = 192 + c16;
      shr[3 * K0S + q] = (k < n + m && nn < LS32_KH) ? a.dyn.W[0][(size_t)k * LS32_KH + nn] : 0.f;
    }
  }
  // output layer (k-steps 13 wave + j of wave `wave`): fragments in LDS
  for (int e = tid; e < 4 * 13 * 64; e += LS32_THREADS) {
    const int l = e & 63, j = (e >> 6) % 13, wv = (e >> 6) / 13;
    const int ks = 13 * wv + j, no = l & 15;
    wol[e] = (ks < LS32_KS && no < n) ? a.dyn.W[Lh][(size_t)(4 * ks + (l >> 4)) * n + no] : 0.f;
  }
  if constexpr (NOB > 1)
    for (int e = tid; e < 4 * 13 * 4 * NV8; e += LS32_THREADS) {
      const int cc = e % NV8, gg = (e / NV8) & 3, ks = e / (4 * NV8);      // ks = 13 wave + j
      wol1[e] = (ks < LS32_KS && cc < NV) ? a.dyn.W[Lh][(size_t)(4 * ks + gg) * n + 16 + cc] : 0.f;
    }
  for (int e = tid; e < 3 * LS32_ROWS + 32; e += LS32_THREADS) {
    float v = 0.f;
    if (e < 3 * LS32_ROWS) {
      const int l = e / LS32_ROWS, j = e - l * LS32_ROWS;
      if (j < LS32_KH) v = a.dyn.b[l][j];
    } else if (e - 3 * LS32_ROWS < n) {
      v = a.dyn.b[Lh][e - 3 * LS32_ROWS];
    }
    bias_s[e] = v;
  }
  // every per-group buffer starts at zero (xcur rows >= n + m, activation rows 192.., mask words, dxs = x_0 - X_0)
  for (int e = tid; e < NG * GSZ; e += LS32_THREADS) smf[e] = 0.f;
  __syncthreads();
  auto BI = [&](int gi, int c) -> int { return s_bi[gi][c]; };
  auto INB = [&](int gi, int c) -> bool { return s_in[gi][c] != 0; };
  // (candidate index; the per-step global accesses below index with 32-bit offsets from the uniform buffer pointers --
  // gmpc_ls32_shape checks that the buffers are that small -- so that an address costs one register, not two)
  auto CI = [&](int gi, int c) -> unsigned { return (unsigned)(b0 + gi * LS32_C + c); };
  // ---- initial state: the nominal trajectory's x_0
  for (int gi = 0; gi < NG; ++gi) {
    float* xc = grp(gi).xcur;
    for (int e = tid; e < LS32_C * n; e += LS32_THREADS) {
      const int c = e / n, i = e - c * n;
      xc[ls32_at(i, c)] = a.X[(size_t)BI(gi, c) * (T + 1) * n + i];
    }
  }

  // ---- per-thread roles (thread tt of a team has the role of k_ls16's thread tt).  With 256 registers per wave and
  // 170 of them weights, a role's lane-dependent values (indices, addresses, table look-ups) are NOT kept over the
  // horizon: every phase derives what it needs from an opaque copy of the thread index (LS32_LANE), a handful of
  // integer instructions per phase -- left to itself hipcc hoists some seventy such values out of the time loop and
  // spills them.  Kept: the (candidate, control) pair of the controls phase (a division by m).
  constexpr int PE = 2 * K0S;
  const int cc = min((tt >> 1) / m, LS32_C - 1), cj = (tt >> 1) - ((tt >> 1) / m) * m;
  constexpr int KQ = K0S == 4 ? 4 : 8;          // m n <= 16 KQ
  static_assert(KQ + 4 <= 3 * K0S + 2, "the operand registers share layer 0's");
#define LS32_LANE()                                                                                   \
  int tq_ = tt;                                                                                       \
  asm volatile("" : "+v"(tq_));                                                                       \
  const int tq = tq_, lane = tq & 63, g = lane >> 4, c16 = lane & 15;                                 \
  (void)g; (void)c16

  // operands of the controls of step t of one group (team B: requested in the group's O phase, staged in its U
  // phase): 16 lanes per candidate (candidate tq >> 4, elements (tq & 15) + 16 q of its gain block); the nominal
  // state's coordinates of the state update's roles (see phaseU)
  auto prefetch = [&](int gi, int t, int tq) __attribute__((always_inline)) {
    const int kc = tq >> 4, kl = tq & 15, lane = tq & 63, g = lane >> 4, c16 = lane & 15;
    const int no1 = 4 * g + wave, q2 = tq >> 4;
    const unsigned bt = (unsigned)BI(gi, kc) * (unsigned)T + (unsigned)t;
#pragma unroll
    for (int q = 0; q < KQ; ++q) shr[q] = a.Kg[bt * (unsigned)MN + (unsigned)min(kl + 16 * q, MN - 1)];
    const unsigned ku = bt * (unsigned)m + (unsigned)min(kl, m - 1);
    shr[KQ] = a.kg[ku];
    shr[KQ + 1] = a.Uio[ku];
    const unsigned xt = ((unsigned)BI(gi, c16) * (unsigned)(T + 1) + (unsigned)t) * (unsigned)n;
    shr[KQ + 2] = a.X[xt + (unsigned)(no1 < n ? no1 : 0)];
    if (NOB > 1) shr[KQ + 3] = a.X[xt + (unsigned)(16 + q2 < n ? 16 + q2 : 0)];
  };
  // (gain blocks at stride MNX = 16 KQ, k and U at stride 8: every lane has a slot of its own and nothing is predicated
  // but the k / U pair of lanes 8 .. 15)
  auto stage = [&](const Grp& G, int tq) __attribute__((always_inline)) {
    const int kc = tq >> 4, kl = tq & 15;
    float* kd = G.Ks + kc * LY::MNX + kl;
#pragma unroll
    for (int q = 0; q < KQ; ++q) kd[16 * q] = shr[q];
    if (kl < 8) {
      G.kUs[kc * 8 + kl] = shr[KQ];
      G.kUs[LS32_C * 8 + kc * 8 + kl] = shr[KQ + 1];
    }
  };
  // step 0's operands of both groups
  if (team == 1) {
    for (int gi = 0; gi < NG; ++gi) {
      prefetch(gi, 0, tt);
      stage(grp(gi), tt);
    }
  }
  __syncthreads();

  const unsigned wm0 = wave == 0 ? 0xFFFFFFFFu : 0u, wm1 = wave == 1 ? 0xFFFFFFFFu : 0u, wm3 = wave == 3 ? 0xFFFFFFFFu : 0u;
  auto bsel = [](unsigned mask, float a, float b) -> float {
    return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
  };
  // candidate outputs through buffer resources (ls32_store_if); the sizes are bounded by gmpc_ls32_shape
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(a.Uc, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(a.Xc, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(a.maskc, 0, 0x7FFFFFF0, 0x00020000);
  // (LDS-only barrier: the global stores of a phase are not read inside the horizon)
#define LS32_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef GMPC_TRAJ_STAMPS
  // diagnostic build: cycles per half-step spent in each of the four segments (work) and at its barrier (wait)
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
#define TS_(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define TS_(i)
#endif
#define LS32_SEG_END(i) do { TS_(2 * (i)) LS32_BAR(); TS_(2 * (i) + 1) } while (0)
  // ================= the six phases of a group's step (gb: the group's region; tb: the shared tables) =================
  auto phaseC = [&](int gi, float* gb, int t) __attribute__((always_inline)) {
    const Grp G = grp_at(gb);
    LS32_LANE();
    const int cp = tq >> 1, chalf = tq & 1;
    const bool con = cp < LS32_C * m;
    // (lanes without a (candidate, control) pair run the same instructions on pair 0's operands and store nowhere)
    const float* kcb = G.Ks + (con ? cc * LY::MNX + cj * n : 0) + chalf;
    const float* dcb = G.dxs + chalf * 16 + cc;
    float du = 0.f;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      // row i = chalf + 2 e of dxs: group e / 2, row (e & 1) 2 + chalf of the group
      const float dx = dcb[(e >> 1) * LS32_GS + (e & 1) * 32];
      du = fmaf(kcb[2 * e], chalf + 2 * e < n ? dx : 0.f, du);
    }
    du += __shfl_xor(du, 1);
    const int cq = cc * 8 + cj;
    const float u = G.kUs[LS32_C * 8 + cq] + fmaf(s_alpha[gi][cc], G.kUs[cq], du);
    if (con && chalf == 0) {
      ls32_store_if(rsU, (CI(gi, cc) * (unsigned)T + (unsigned)t) * (unsigned)m + (unsigned)cj, __float_as_uint(u),
                    b0 + gi * LS32_C + cc < cnt);
      G.xcur[ls32_at(n + cj, cc)] = u;
    }
  };
  auto phaseL0 = [&](float* gb, const float* tb) __attribute__((always_inline)) {
    const Grp G = grp_at(gb);
    LS32_LANE();
    const float* bias_s = tb + LY::BIAS;
    f32x4_t d[3], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_s + 16 * (wave + 4 * r) + 4 * g);
      d[r] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
    }
    float bf[K0S];
#pragma unroll
    for (int ks = 0; ks < K0S; ++ks) bf[ks] = G.xcur[ks * LS32_GS + lane];
    const float* xw = G.xcur + ls32_opq(wave * LS32_GS + lane);
    const float bx0 = xw[0], bx1 = xw[4 * LS32_GS];
#pragma unroll
    for (int ks = 0; ks < K0S; ++ks)
#pragma unroll
      for (int r = 0; r < 3; ++r) d[r] = ls32_mfma(shr[r * K0S + ks], bf[ks], d[r]);
    dx = ls32_mfma(shr[3 * K0S], bx0, dx);
    dx = ls32_mfma(shr[3 * K0S + 1], bx1, dx);
    {
      const int ob = ls32_opq((4 * wave + g) * LS32_GS + c16), mo = ls32_opq(c16 * 96 + 2 * wave + (g >> 1));
#pragma unroll
      for (int r = 0; r < 3; ++r) ls32_epilogue(d[r], G.actA + ob + 16 * r * LS32_GS, G.mask + mo + 8 * r, lane);
    }
    if (lane < 32) {
      float* pl = G.p12 + ls32_opq(wave * 128 + lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) pl[i * 32] = dx[i];
    }
  };
  // hidden layer hl (0: actA -> actB, partials p12[0] -> p12[1]; 1: actB -> actA, p12[1] -> p12[0]) with the weights
  // of this wave's team; one barrier in the middle (the other team's phase boundary)
  auto phaseH = [&](auto hlc, float* gb, const float* tb) __attribute__((always_inline)) {
    constexpr int hl = decltype(hlc)::value;
    const Grp G = grp_at(gb);
    LS32_LANE();
    const float* bias_s = tb + LY::BIAS;
    const float* wxl = tb + LY::WXL;
    const float* hin = hl == 0 ? G.actA : G.actB;
    float* hout = hl == 0 ? G.actB : G.actA;
    const float2 tail = ls32_tail(G.p12 + (hl & 1) * 512, bias_s + hl * LS32_ROWS + 192, G.mask + hl * 32, lane);
    f32x4_t d[3], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_s + (hl + 1) * LS32_ROWS + 16 * (wave + 4 * r) + 4 * g);
      d[r] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
    }
    // chunks of 4 k-steps, operands of chunk j + 1 read while chunk j multiplies: 4 B fragments, and the A / B
    // fragments of this wave's block-12 k-step 4 j + wave
    const float* wx = wxl + hl * 52 * 32 + ls32_opq(wave * 32 + g * 8 + (c16 & 7));
    const float* hx = hin + ls32_opq(wave * LS32_GS + lane);
    float bq[2][4], ax[2], bx[2];
    auto load_chunk = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ks = 4 * j + e;
        if (ks < 48) bq[j & 1][e] = hin[ks * LS32_GS + lane];
      }
      // (lanes c16 >= 8 read the fragment of lane c16 - 8: rows 200 .. 207 of the product come out as copies of rows
      // 192 .. 199 and are never stored)
      ax[j & 1] = wx[4 * j * 32];
      if (j < 12) bx[j & 1] = hx[4 * j * LS32_GS];
    };
    load_chunk(std::integral_constant<int, 0>{});
    rw_static_for<13>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < 13) load_chunk(std::integral_constant<int, j + 1>{});
      if constexpr (j == 12) {
        bq[0][0] = tail.x;
        bq[0][1] = tail.y;
        bx[0] = __uint_as_float((__float_as_uint(tail.x) & wm0) | (__float_as_uint(tail.y) & wm1));
      }
      rw_static_for<4>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        constexpr int ks = 4 * j + e;
        if constexpr (ks < LS32_KS) {
          d[0] = ls32_mfma(wr[0][ks], bq[j & 1][e], d[0]);
          d[1] = ls32_mfma(wr[1][ks], bq[j & 1][e], d[1]);
          d[2] = ls32_mfma(wr[2][ks], bq[j & 1][e], d[2]);
        }
        if constexpr (e == 1) dx = ls32_mfma(ax[j & 1], bx[j & 1], dx);
      });
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                 // the next chunk's LDS reads
      __builtin_amdgcn_sched_group_barrier(0x008, j < 12 ? 13 : 7, 0);   // this chunk's MFMAs
      if constexpr (j == 6) LS32_SEG_END(hl == 0 ? 2 : 0);
    });
    {
      const int ob = ls32_opq((4 * wave + g) * LS32_GS + c16), mo = ls32_opq(c16 * 96 + 2 * wave + (g >> 1));
#pragma unroll
      for (int r = 0; r < 3; ++r)
        ls32_epilogue(d[r], hout + ob + 16 * r * LS32_GS, G.mask + (hl + 1) * 32 + mo + 8 * r, lane);
    }
    if (lane < 32) {
      float* pl = G.p12 + ((hl + 1) & 1) * 512 + ls32_opq(wave * 128 + lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) pl[i * 32] = dx[i];
    }
  };
  // output layer: k-steps 13 wave .. 13 wave + 12, partial sums through LDS
  auto phaseO = [&](int gi, float* gb, const float* tb, int t) __attribute__((always_inline)) {
    const Grp G = grp_at(gb);
    LS32_LANE();
    const float* bias_s = tb + LY::BIAS;
    const float* wol = tb + LY::WOL;
    const float* wol1 = tb + LY::WOL1;
    const float* hin = G.actA;
    const float2 tail = ls32_tail(G.p12, bias_s + 2 * LS32_ROWS + 192, G.mask + 2 * 32, lane);
    f32x4_t d[NOB];
#pragma unroll
    for (int blk = 0; blk < NOB; ++blk) d[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bf[13], wo[NOB][13];
    const float* hl_ = hin + ls32_opq(13 * wave * LS32_GS + lane);                   // (52 groups)
    const float* wl_ = wol + ls32_opq(wave * 13 * 64 + lane);
    const float* w1_ = wol1 + ls32_opq((wave * 13 * 4 + g) * NV8 + (c16 & 7));
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      bf[j] = hl_[j * LS32_GS];
      wo[0][j] = wl_[j * 64];
      if (NOB > 1) wo[NOB - 1][j] = w1_[j * 4 * NV8];                               // (rows >= 24: copies, never read)
    }
    bf[9] = bsel(wm3, tail.x, bf[9]);                                             // k-steps 48, 49
    bf[10] = bsel(wm3, tail.y, bf[10]);
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
      for (int blk = 0; blk < NOB; ++blk) d[blk] = ls32_mfma(wo[blk][j], bf[j], d[blk]);
    float* pl = G.part + ls32_opq(wave * NOB * 256 + lane);
#pragma unroll
    for (int blk = 0; blk < NOB; ++blk)
#pragma unroll
      for (int i = 0; i < 4; ++i) pl[(blk * 4 + i) * 64] = d[blk][i];
  };
  // x_{t+1} = x_t + b_L + the four partials (thread = one (coordinate, candidate) of block 0, some also of block 1);
  // this step's mask words leave (two of the candidate block's 384 finished words per thread); the next step's
  // operands are staged
  auto phaseU = [&](int gi, float* gb, const float* tb, int t) __attribute__((always_inline)) {
    const Grp G = grp_at(gb);
    LS32_LANE();
    const float* part = G.part;
    const unsigned ci0 = (unsigned)(b0 + gi * LS32_C);                 // first candidate of the group
    const unsigned left = (unsigned)max(cnt - (int)ci0, 0);           // candidates of the group that exist
    // thread (wave i, lane (g, c)) owns coordinate 4 g + i of candidate c (output block 0); threads < 16 (n - 16) also
    // own coordinate 16 + tq / 16 of candidate tq & 15 (output block 1)
    const int no1 = 4 * g + wave, q2 = tq >> 4;
    const bool on1 = no1 < n, on2 = NOB > 1 && 16 + q2 < n;
    const int x1 = ls32_at(no1, c16), x2 = ls32_at(on2 ? 16 + q2 : 0, c16);
    const float bo1 = tb[LY::BIAS + 3 * LS32_ROWS + no1];                            // (0 past n)
    // (coordinates past n: the same sums on in-range addresses, stored nowhere)
    const float v1 = (((part[tq] + part[NOB * 256 + tq]) + (part[2 * NOB * 256 + tq] + part[3 * NOB * 256 + tq])) + bo1) + G.xcur[x1];
    const unsigned xo = ((ci0 + (unsigned)c16) * (unsigned)(T + 1) + (unsigned)(t + 1)) * (unsigned)n;
    const bool in16 = (unsigned)c16 < left;
    if (on1) {
      G.xcur[x1] = v1;
      G.dxs[x1] = v1 - shr[KQ + 2];
    }
    ls32_store_if(rsX, xo + (unsigned)no1, __float_as_uint(v1), in16 && on1);
    if constexpr (NOB > 1) {
      const int pi2 = 256 + (q2 & 3) * 64 + 16 * ((q2 >> 2) & 3) + c16;
      const float bo2 = tb[LY::BIAS + 3 * LS32_ROWS + 16 + q2];
      const float v2 = (((part[pi2] + part[NOB * 256 + pi2]) + (part[2 * NOB * 256 + pi2] + part[3 * NOB * 256 + pi2])) + bo2) + G.xcur[x2];
      if (on2) {
        G.xcur[x2] = v2;
        G.dxs[x2] = v2 - shr[KQ + 3];
      }
      ls32_store_if(rsX, xo + (unsigned)(16 + q2), __float_as_uint(v2), in16 && on2);
    }
    stage(G, tq);
    {
      // words tq and tq + 256 of the [16][24] block: word w of candidate c = image word c 24 + w
      const unsigned* img = reinterpret_cast<const unsigned*>(G.mask);
      const unsigned ms = (unsigned)mstride, tw = ci0 * ms + (unsigned)t * 24u;
      const unsigned mc1 = (unsigned)tq / 24u, mc2 = (unsigned)(tq + 256) / 24u;      // (mc2 > 15: past the block)
      ls32_store_if(rsM, tw + mc1 * (ms - 24u) + (unsigned)tq, img[tq], mc1 < left);   // (mc1 <= 10)
      ls32_store_if(rsM, tw + mc2 * (ms - 24u) + (unsigned)(tq + 256), img[(tq + 256) & 511],
                    mc2 < (unsigned)LS32_C && mc2 < left);
    }
  };

  using H1_ = std::integral_constant<int, 0>;
  using H2_ = std::integral_constant<int, 1>;
  // ================= the horizon: 2 T + 1 half-steps =================
  // half-step hs: team A takes group hs & 1 through C, L0, H1 of step hs >> 1; team B takes the OTHER group through H2,
  // O, U of step (hs - 1) >> 1 (the half-step before, team A had that group).  Four barriers per half-step and team.
#ifdef GMPC_TRAJ_STAMPS
  tp_ = __builtin_readcyclecounter();
  const unsigned long long k0_ = tp_;
#endif
  // (one loop per team: each is then register-allocated on its own -- in a common loop the register set the teams
  // share, `shr`, became a loop-carried value that the compiler copied, behind a wait for every outstanding memory
  // operation, at the top of every half-step)
  if (team == 0) {
    for (int hs = 0; hs < 2 * T; ++hs) {
      const int gi = hs & 1, t = hs >> 1;
      float* gb = gbase(gi);
      // (the latency chains run at raised priority: measured on two waves of one SIMD, a wave issuing MFMAs back to back
      // otherwise keeps the other wave's LDS and MFMA instructions waiting; plain vector instructions of the other wave
      // get about one slot per MFMA even so -- tests/repro/fp32_mfma_valu_port.hip -- which is why the chains are
      // kept short in vector instructions)
      __builtin_amdgcn_s_setprio(3);
      phaseC(gi, gb, t);
      LS32_SEG_END(0);
      phaseL0(gb, tabs);
      __builtin_amdgcn_s_setprio(0);
      LS32_SEG_END(1);
      phaseH(H1_{}, gb, tabs);
      LS32_SEG_END(3);
    }
    LS32_BAR(); LS32_BAR(); LS32_BAR(); LS32_BAR();
  } else {
    LS32_BAR(); LS32_BAR(); LS32_BAR(); LS32_BAR();
    for (int hs = 1; hs <= 2 * T; ++hs) {
      const int gi = (hs & 1) ^ 1, t = (hs - 1) >> 1;
      float* gb = gbase(gi);
      // the operands of the group's next step are requested here and staged by the state update, a hidden layer later
      prefetch(gi, min(t + 1, T - 1), tt);      // (the last step requests its own operands again: unused)
      phaseH(H2_{}, gb, tabs);
      __builtin_amdgcn_s_setprio(3);
      LS32_SEG_END(1);
      phaseO(gi, gb, tabs, t);
      LS32_SEG_END(2);
      phaseU(gi, gb, tabs, t);
      __builtin_amdgcn_s_setprio(0);
      LS32_SEG_END(3);
    }
  }
#ifdef GMPC_TRAJ_STAMPS
  if (blockIdx.x == 0 && tt == 0)
    printf("k_ls32 team %d cycles per half-step: seg0 %llu+%llu seg1 %llu+%llu seg2 %llu+%llu seg3 %llu+%llu | loop %llu\n", team,
           st_[0] / (2 * T), st_[1] / (2 * T), st_[2] / (2 * T), st_[3] / (2 * T), st_[4] / (2 * T), st_[5] / (2 * T),
           st_[6] / (2 * T), st_[7] / (2 * T), __builtin_readcyclecounter() - k0_);
#endif
  __syncthreads();

  // ---- stage costs (team = group): 4 lanes per (candidate, step) pair, 64 pairs per sweep; summed per candidate in
  // step order
  {
    const int gi = team;
    const Grp G = grp(gi);
    float* cst = G.actB;
    const float al = GMPC_ALPHA;
    const int q = tt & 3;
    for (int p = tt >> 2; p < LS32_C * T; p += 64) {
      const int c = p / T, t = p - c * T;
      const int bc = BI(gi, c);
      const size_t ci = INB(gi, c) ? (size_t)CI(gi, c) : 0;  // (unused candidates read item 0's rows: in bounds, discarded)
      const float* xr = t > 0 ? a.Xc + (ci * (T + 1) + t) * n : a.X + (size_t)bc * (T + 1) * n;
      const float* ur = a.Uc + (ci * T + t) * m;
      const float* gl = a.goal + ((size_t)bc * (T + 1) + t) * n;
      float xv[8], gv[8], uv[2];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = min(q + 4 * e, n - 1);
        xv[e] = xr[i];
        gv[e] = gl[i];
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) uv[e] = ur[min(q + 4 * e, m - 1)];
      float dd = 0.f, uu = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dx = q + 4 * e < n ? xv[e] - gv[e] : 0.f;
        dd = fmaf(dx, dx, dd);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float u = q + 4 * e < m ? uv[e] : 0.f;
        uu = fmaf(u, u, uu);
      }
      dd += __shfl_xor(dd, 1); dd += __shfl_xor(dd, 2);
      uu += __shfl_xor(uu, 1); uu += __shfl_xor(uu, 2);
      if (q == 0) cst[p] = INB(gi, c) ? w0 * (sqrtf(uu + al * al) - al) + w1 * (sqrtf(dd + al * al) - al) : 0.f;
    }
  }
  __syncthreads();
  if (tid < NG * LS32_C) {
    const float* cst = smf + (tid >> 4) * GSZ + LY::ACTB + (tid & 15) * T;      // group's actB
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += cst[t];
    s_obj[tid >> 4][tid & 15] = acc;
  }
  __syncthreads();            // (the stage costs have been read: actB is a layer buffer again)
  // ---- terminal cost w2 |cost_mlp(x_T)|^2 on the matrix pipe as well (team = group): activations [k][16] in actA /
  // actB, weight fragments straight from global memory (row blocks nb = wave, wave + 4, ..)
  {
    const int gi = team;
    const Grp G = grp(gi);
    float* in = G.actA;
    float* out = G.actB;
    for (int e = tt; e < LS32_C * ((n + 3) & ~3); e += 256) {
      const int i = e >> 4, c = e & 15;
      in[ls32_at(i, c)] = i < n ? G.xcur[ls32_at(i, c)] : 0.f;
    }
    __syncthreads();
    const int Lc = a.cost.L - 1;
    for (int l = 0; l <= Lc; ++l) {
      const int fi = a.cost.dims[l], fo = a.cost.dims[l + 1];
      const float* W = a.cost.W[l];
      const float* bv = a.cost.b[l];
      const int nks = (fi + 3) >> 2;
      for (int nb = wave; 16 * nb < fo; nb += 4) {
        const int col = 16 * nb + c16;
        const bool colok = col < fo;
        f32x4_t acc;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = 16 * nb + 4 * g + i < fo ? bv[16 * nb + 4 * g + i] : 0.f;
        const float* wp = W + (colok ? col : 0);
        for (int k0 = 0; k0 < nks; k0 += 8) {        // 8 fragments in flight (k-steps past the last: zero weights)
          float wv[8], bq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ks = min(k0 + e, nks - 1), k = 4 * ks + g;
            const float w = wp[(size_t)min(k, fi - 1) * fo];
            wv[e] = (k0 + e < nks && k < fi && colok) ? w : 0.f;
            bq[e] = in[ks * LS32_GS + lane];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = ls32_mfma(wv[e], bq[e], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = l < Lc ? fmaxf(acc[i], 0.f) : acc[i];
          if (16 * nb + 4 * g + i >= fo) v = 0.f;
          out[(4 * nb + g) * LS32_GS + i * 16 + c16] = v;
        }
      }
      __syncthreads();
      float* tmp = in; in = out; out = tmp;
    }
    if (tt < LS32_C && INB(gi, tt)) {
      const int fo = a.cost.dims[Lc + 1];
      float yy = 0.f;
      for (int r = 0; r < fo; ++r) {
        const float y = in[ls32_at(r, tt)];
        yy = fmaf(y, y, yy);
      }
      a.objc[CI(gi, tt)] = s_obj[gi][tt] + w2 * yy;
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------
bool gmpc_ls16_shape(const TrajArgs& a);

static size_t ls32_lds(int n, int m) {
  const int k0s = (n + m + 3) / 4;
  const int fl = n > 16 ? Ls32Lay<6, 2>::TOTAL : k0s <= 4 ? Ls32Lay<4, 1>::TOTAL : Ls32Lay<6, 1>::TOTAL;
  return (size_t)fl * sizeof(float);
}
// shapes k_ls32 is instantiated for: k_ls16's with at most 8 rows in output block 1, a gain block that fits its LDS
// slot, stage costs that fit an activation buffer and candidate buffers a 32-bit index reaches
bool gmpc_ls32_shape(const TrajArgs& a) {
  const char* e = getenv("GMPC_LS");        // read per call: the tests switch forms inside one process
  if (e != nullptr && (strcmp(e, "rw") == 0 || strcmp(e, "ls16") == 0)) return false;
  if (!gmpc_ls16_shape(a) || a.dyn.dims[1] != LS32_KH) return false;      // (the 200-wide form only)
  const int k0s = (a.n + a.m + 3) / 4;
  const long items = (long)a.B * GMPC_LS_ITEMS;
  return a.n <= 24 && a.m * a.n <= (k0s <= 4 && a.n <= 16 ? 64 : 128) && ls32_lds(a.n, a.m) <= LS32_LDS_MAX &&
         LS32_C * a.T <= LS32_ACT && items * (a.T + 1) * a.n < (1L << 31) && items * a.T * 3 * GMPC_MW < (1L << 31);
}
// work lists of at least this many candidates are k_ls32's: more than one pass of k_ls16 over the 256 CUs (4096
// candidates).  Measured in round 4 at C3 (profiles/EXPERIMENTS.md): the 8192 candidates of a first round take one pass
// of this kernel 0.79 ms against 0.92 ms for two passes of k_ls16; a list of 4096 or fewer is one k_ls16 pass (0.46 ms)
// and would be half a chip of this kernel for 0.79 ms.  GMPC_LS32_SPLIT overrides the threshold (0: never; the tests
// set 1 to send every round here).
int gmpc_ls32_split() {
  const char* e = getenv("GMPC_LS32_SPLIT");
  return e != nullptr ? atoi(e) : 4097;
}

template <int K0S, int NOB>
static void ls32_launch(const TrajArgs& a, int grid, int min_items, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ls32<K0S, NOB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LS32_LDS_MAX);
    (void)hipGetLastError();
    attr = true;
  }
  hipLaunchKernelGGL((k_ls32<K0S, NOB>), dim3(grid), dim3(LS32_THREADS), ls32_lds(a.n, a.m), s, a, min_items);
}

// one workgroup per 32 work-list items; `max_items` bounds the list (the kernel reads the actual count)
void gmpc_launch_ls32(const TrajArgs& a, long max_items, int min_items, hipStream_t s) {
  const int per = LS32_NG * LS32_C;
  const int grid = (int)((max_items + per - 1) / per);
  const int k0s = (a.n + a.m + 3) / 4;
  if (a.n <= 16) {
    if (k0s <= 4) ls32_launch<4, 1>(a, grid, min_items, s);
    else ls32_launch<6, 1>(a, grid, min_items, s);
  } else {
    ls32_launch<6, 2>(a, grid, min_items, s);
  }
}
