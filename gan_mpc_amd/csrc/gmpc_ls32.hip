// Line-search candidates, 32 per workgroup in two groups of 16 that run half a time step apart
// (k_ls32<K0S, NOB>, 256 threads): the form for rounds whose work list holds more than one pass of k_ls16.
//
// k_ls16 (gmpc_ls16.hip) keeps the dynamics network's two 200 x 200 matrices in the registers of 4 waves (one per
// SIMD) and multiplies 16 candidates at a time on v_mfma_f32_16x16x4_f32.  A step is a chain of six phases --
// controls C, layer 0 L0, hidden layers H1 and H2, output layer O, state update U -- separated by workgroup barriers;
// only H1 and H2 keep the matrix pipe busy (163 of the 209 MFMAs of a wave and step, 5.6 k of their 6.7 k cycles
// each); C, L0, O and U are latency chains (LDS round trips, a few MFMAs, barriers): 6.1 k of the 19.5 k cycles of a
// step.  At 1024 trajectories the first round of a line search holds up to 8192 candidates: two full passes of k_ls16
// over the chip.
//
// Here a workgroup owns TWO groups of 16 candidates, X and Y, with every per-group buffer twice in LDS and the
// weights once in the registers, and runs Y four phases behind X:
//     slot     1        2        3        4        5        6
//     X        C(t)     L0(t)    H1(t)    H2(t)    O(t)     U(t)
//     Y        O(t-1)   U(t-1)   C(t)     L0(t)    H1(t)    H2(t)
// One barrier per slot: six per step for 32 candidates instead of twelve.  In slots 3 .. 6 the latency chain of one
// group sits in the same basic block as the other group's matrix-pipe phase, so its LDS round trips and vector work
// issue between MFMAs instead of in front of an idle pipe (one wave per SIMD executes in order: the overlap has to
// be in the instruction stream); slots 1 and 2 pair two latency chains.  Everything a candidate computes is what
// k_ls16 computes, operation for operation (same fragments, same accumulation order): results are bit-identical.
// The 8192 candidates of a round are ONE pass over the 256 CUs.
//
// Differences in layout against k_ls16 (LDS has to hold two groups): the A fragments of row block 12 and of output
// block 1 keep their non-zero lanes only, the K-split partials of block 12 their 32 useful lanes; the operands of a
// step's controls (gains, k, U, nominal state) are loaded by 16 lanes per candidate one slot before the state update
// that stages them, one register set shared by the two groups; the per-step global pointers are recomputed.
//
// Reference arithmetic: dynamics/nn.py:27-34, cost/cost_model.py:20-42, cost/nn.py:23-29, trajax
// line_search_ddp / ddp_rollout (u = U + alpha k + K (x - X)) as called from policy/optimizers.py:19.
#include "gmpc_device.h"
#include <cstdlib>
#include <cstring>

#define LS32_THREADS 256
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
// bits of rows 16 nb + 4 g + i as ONE BYTE of the candidate's mask image of this layer (mb: 64 bytes per candidate and
// layer, byte 4 nb + g).  k_ls16 ORs the nibble into the mask word with an LDS atomic; an atomic is an ordered memory
// operation to the instruction scheduler -- nothing of the other group's phase could be moved across it -- so here
// every nibble has a byte of its own (plain stores, written anew every step: no clearing) and the state update packs
// eight of them into the word it writes out.
__device__ __forceinline__ void ls32_epilogue(f32x4_t d, int nb, float* out, unsigned char* mb) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  unsigned nib = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = d[i] > 0.f;
    nib |= on ? (1u << i) : 0u;
    out[(4 * nb + g) * LS32_GS + i * 16 + c] = on ? d[i] : 0.f;
  }
  mb[c * 192 + 4 * nb + g] = (unsigned char)nib;
}

// rows 192 + g (.x) and 196 + g (.y) of a hidden layer's output for candidate lane & 15 -- the B fragments of
// k-steps 48 and 49 -- from the K-split partials of row block 12 ([4 waves][4 registers][32 lanes]); their relu bits go
// into bytes 48 + g and 52 + g of the candidate's mask image (every wave stores the same values)
__device__ __forceinline__ float2 ls32_tail(const float* p12, const float* bias192, unsigned char* mb) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const float* q = p12 + g * 32 + c;
  const float s0 = ((q[0] + q[128]) + (q[256] + q[384])) + bias192[g];
  const float s1 = ((q[16] + q[144]) + (q[272] + q[400])) + bias192[4 + g];
  mb[c * 192 + 48 + g] = s0 > 0.f ? 1 : 0;
  mb[c * 192 + 52 + g] = s1 > 0.f ? 1 : 0;
  return make_float2(fmaxf(s0, 0.f), fmaxf(s1, 0.f));
}

// Global store without a branch: lanes that are not to store get an offset past the end of the buffer resource and the
// hardware drops their write (raw buffer, range-checked).  A phase of k_ls32 must stay ONE basic block -- an
// `if (lane has a role) store` is an exec-mask region that the matrix-pipe instructions of the other group's phase
// cannot be scheduled across.
__device__ __forceinline__ void ls32_store_if(__amdgpu_buffer_rsrc_t rs, unsigned elem, unsigned bits, bool on) {
  __builtin_amdgcn_raw_buffer_store_b32(bits, rs, on ? elem * 4u : 0xFFFFFFF0u, 0, 0);
}

// LDS store without a branch: the element index of a lane without a role is redirected to the group's dummy slot --
// as ARITHMETIC on the index (a `cond ? p : q` store is turned back into two exec-masked stores by hipcc, and an
// exec-mask change orders every vector and matrix instruction of the block around it)
__device__ __forceinline__ int ls32_pick(bool on, int idx, int other) { return other + ((idx - other) & -(int)on); }

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
  static constexpr int MASK = P12 + 2 * 4 * 4 * 32;               // [16][3 layers][64] mask bytes (ls32_epilogue)
  static constexpr int KS = MASK + LS32_C * 48;                   // gains [16][m n] (+ slack: clamped tail reads)
  static constexpr int KUS = KS + LS32_C * MNX + 32;              // k and U, [16][8] each
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
__global__ __launch_bounds__(LS32_THREADS, 1) void k_ls32(TrajArgs a, int min_items) {
  constexpr int NG = LS32_NG;
  extern __shared__ __attribute__((aligned(16))) char smem_ls32[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
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
  const int dmy = LY::DUMMY + lane;          // (float index of this lane's dummy slot inside a group's region)
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
  // The group's base offset is made an opaque scalar at every use inside the time loop: the lane-dependent part of an
  // address is then ONE loop-invariant register shared by the two groups and the group's offset is added where the
  // address is used.  With the offsets visible hipcc hoisted a second, group-1 copy of every address out of the loop
  // (~170 registers) and spilled them.
  auto gbase = [&](int gi) -> float* {
    int go = gi * GSZ;
    asm volatile("" : "+s"(go));
    return smf + go;
  };
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

  // ---- weights: registers for the whole horizon (row blocks wave, wave + 4, wave + 8; block 12 is split over the
  // waves by k-step, k-steps wave + 4 j)
  float wr[2][3][LS32_KS];
  float w0r[3][K0S], w0x[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
    for (int ks = 0; ks < K0S; ++ks) {
      const int k = 4 * ks + g;
      w0r[r][ks] = k < n + m ? a.dyn.W[0][(size_t)k * LS32_KH + nn] : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = 4 * (wave + 4 * q) + g, nn = 192 + c16;
    w0x[q] = (k < n + m && nn < LS32_KH) ? a.dyn.W[0][(size_t)k * LS32_KH + nn] : 0.f;
  }
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const float* Wl = a.dyn.W[hl + 1];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
      for (int ks = 0; ks < LS32_KS; ++ks) wr[hl][r][ks] = Wl[(size_t)(4 * ks + g) * LS32_KH + nn];
    }
    for (int e = tid; e < 52 * 32; e += LS32_THREADS) {
      const int ks = e >> 5, gg = (e >> 3) & 3, cc = e & 7;
      wxl[hl * 52 * 32 + e] = ks < LS32_KS ? Wl[(size_t)(4 * ks + gg) * LS32_KH + 192 + cc] : 0.f;
    }
  }
  // (see k_ls16: the weight registers the allocator has to keep in the accumulation file are named here, so that
  // the MFMAs read them there instead of through a copy)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int ks = 0; ks < LS32_KS; ++ks) asm volatile("" : "+a"(wr[0][r][ks]));
#pragma unroll
  for (int ks = 0; ks < LS32_KS; ++ks) asm volatile("" : "+a"(wr[1][0][ks]));
  // output layer (k-steps 13 wave + j of wave `wave`): fragments in LDS
  for (int e = tid; e < 4 * 13 * 64; e += LS32_THREADS) {
    const int l = e & 63, j = (e >> 6) % 13, wv = (e >> 6) / 13;
    const int ks = 13 * wv + j, no = l & 15;
    wol[e] = (ks < LS32_KS && no < n) ? a.dyn.W[Lh][(size_t)(4 * ks + (l >> 4)) * n + no] : 0.f;
  }
  if (NOB > 1)
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

  // ---- per-thread roles, fixed for the horizon
  // controls u = U + alpha k + K (x - X_nominal): two lanes per (candidate, control) pair (elements i = half + 2 e)
  constexpr int PE = 2 * K0S;
  const int cp = tid >> 1, chalf = tid & 1;
  const int cc = min(cp / m, LS32_C - 1), cj = cp - (cp / m) * m;
  const bool con = cp < LS32_C * m;
  float calpha[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) calpha[gi] = con ? s_alpha[gi][cc] : 0.f;
  // operand loads of a step: 16 lanes per candidate (candidate tid >> 4, elements (tid & 15) + 16 q of its gain block)
  constexpr int KQ = K0S == 4 ? 4 : 8;          // m n <= 16 KQ
  const int kc = tid >> 4, kl = tid & 15;
  // state update: thread (wave i, lane (g, c)) owns coordinate 4 g + i of candidate c (output block 0); threads
  // < 16 (n - 16) also own coordinate 16 + tid / 16 of candidate tid & 15 (output block 1)
  const int no1 = 4 * g + wave;
  const bool on1 = no1 < n;
  const int x1 = ls32_at(no1, c16);
  const float bo1 = on1 ? bias_s[3 * LS32_ROWS + no1] : 0.f;
  const int q2 = tid >> 4;
  const bool on2 = NOB > 1 && 16 + q2 < n;
  const int x2 = ls32_at(on2 ? 16 + q2 : 0, c16);
  const int pi2 = 256 + (q2 & 3) * 64 + 16 * ((q2 >> 2) & 3) + c16;
  const float bo2 = on2 ? bias_s[3 * LS32_ROWS + 16 + q2] : 0.f;
  int bik[NG], bi16[NG];
  bool in16[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) { bik[gi] = BI(gi, kc); bi16[gi] = BI(gi, c16); in16[gi] = INB(gi, c16); }

  // operands of the controls of step t for ONE group at a time (register set shared by the groups: requested in the
  // group's O slot, staged in its U slot)
  float pfK[KQ], pfk = 0.f, pfU = 0.f, pfX1 = 0.f, pfX2 = 0.f;
  auto prefetch = [&](int gi, int t) {
    const unsigned bt = (unsigned)bik[gi] * (unsigned)T + (unsigned)t;
#pragma unroll
    for (int q = 0; q < KQ; ++q) pfK[q] = a.Kg[bt * (unsigned)MN + (unsigned)min(kl + 16 * q, MN - 1)];
    const unsigned ku = bt * (unsigned)m + (unsigned)min(kl, m - 1);
    pfk = a.kg[ku];
    pfU = a.Uio[ku];
    const unsigned xt = ((unsigned)bi16[gi] * (unsigned)(T + 1) + (unsigned)t) * (unsigned)n;
    pfX1 = a.X[xt + (unsigned)(on1 ? no1 : 0)];
    if (NOB > 1) pfX2 = a.X[xt + (unsigned)(on2 ? 16 + q2 : 0)];
  };
  auto stage = [&](const Grp& G) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) G.base[ls32_pick(kl + 16 * q < MN, LY::KS + kc * MN + kl + 16 * q, dmy)] = pfK[q];
    G.base[ls32_pick(kl < m, LY::KUS + kc * m + kl, dmy)] = pfk;
    G.base[ls32_pick(kl < m, LY::KUS + LS32_C * 8 + kc * m + kl, dmy)] = pfU;
  };
  // step 0's operands of both groups
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    prefetch(gi, 0);
    stage(grp(gi));
  }
  __syncthreads();

  // wave-uniform selects as bit masks (a `wave == k ? a : b` on a uniform condition is compiled to a scalar BRANCH,
  // which would cut the phase's basic block in two)
  const unsigned wm0 = wave == 0 ? 0xFFFFFFFFu : 0u, wm1 = wave == 1 ? 0xFFFFFFFFu : 0u, wm3 = wave == 3 ? 0xFFFFFFFFu : 0u;
  auto bsel = [](unsigned mask, float a, float b) -> float {
    return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
  };
  // candidate outputs through buffer resources (ls32_store_if); the sizes are bounded by gmpc_ls32_shape
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(a.Uc, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(a.Xc, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(a.maskc, 0, 0x7FFFFFF0, 0x00020000);
  // ================= the six phases of a group's step: straight-line code, no exec-mask regions =================
  // (gb: the group's region; tb: the shared tables -- handed down from a slot's __restrict__ parameters)
  auto phaseC = [&](auto gic, float* gb, int t) __attribute__((always_inline)) {
    constexpr int gi = decltype(gic)::value;
    const Grp G = grp_at(gb);
    // (lanes without a (candidate, control) pair run the same instructions on pair 0's operands and store nowhere)
    const float* kcb = G.Ks + (con ? cc * MN + cj * n : 0) + chalf;
    const float* dcb = G.dxs + chalf * 16 + cc;
    float du = 0.f;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      // row i = chalf + 2 e of dxs: group e / 2, row (e & 1) 2 + chalf of the group
      const float dx = dcb[(e >> 1) * LS32_GS + (e & 1) * 32];
      du = fmaf(kcb[2 * e], chalf + 2 * e < n ? dx : 0.f, du);
    }
    du += __shfl_xor(du, 1);
    const int cq = con ? cp : 0;
    const float u = G.kUs[LS32_C * 8 + cq] + fmaf(calpha[gi], G.kUs[cq], du);
    const bool st = con && chalf == 0;
    ls32_store_if(rsU, (CI(gi, cc) * (unsigned)T + (unsigned)t) * (unsigned)m + (unsigned)cj, __float_as_uint(u),
                  st && INB(gi, cc));
    G.base[ls32_pick(st, LY::XCUR + ls32_at(n + cj, cc), dmy)] = u;
  };
  auto phaseL0 = [&](float* gb, const float* tb) __attribute__((always_inline)) {
    const Grp G = grp_at(gb);
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
    const float bx0 = G.xcur[wave * LS32_GS + lane], bx1 = G.xcur[(wave + 4) * LS32_GS + lane];
#pragma unroll
    for (int ks = 0; ks < K0S; ++ks)
#pragma unroll
      for (int r = 0; r < 3; ++r) d[r] = ls32_mfma(w0r[r][ks], bf[ks], d[r]);
    dx = ls32_mfma(w0x[0], bx0, dx);
    dx = ls32_mfma(w0x[1], bx1, dx);
#pragma unroll
    for (int r = 0; r < 3; ++r) ls32_epilogue(d[r], wave + 4 * r, G.actA, G.mask);
#pragma unroll
    for (int i = 0; i < 4; ++i) G.base[ls32_pick(lane < 32, LY::P12 + (wave * 4 + i) * 32 + lane, dmy)] = dx[i];
  };
  // hidden layer hl (0: actA -> actB, partials p12[0] -> p12[1]; 1: actB -> actA, p12[1] -> p12[0])
  auto phaseH = [&](auto hlc, float* gb, const float* tb) __attribute__((always_inline)) {
    constexpr int hl = decltype(hlc)::value;
    const Grp G = grp_at(gb);
    const float* bias_s = tb + LY::BIAS;
    const float* wxl = tb + LY::WXL;
    const float* hin = hl == 0 ? G.actA : G.actB;
    float* hout = hl == 0 ? G.actB : G.actA;
    const float2 tail = ls32_tail(G.p12 + (hl & 1) * 512, bias_s + hl * LS32_ROWS + 192, G.mask + hl * 64);
    f32x4_t d[3], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_s + (hl + 1) * LS32_ROWS + 16 * (wave + 4 * r) + 4 * g);
      d[r] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
    }
    // chunks of 4 k-steps, operands of chunk j + 1 read while chunk j multiplies: 4 B fragments, and the A / B
    // fragments of this wave's block-12 k-step 4 j + wave
    const float* wx = wxl + (hl * 52 + wave) * 32 + g * 8 + (c16 & 7);
    const float* hx = hin + wave * LS32_GS + lane;
    float bq[2][4], ax[2], bx[2];
    auto load_chunk = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ks = 4 * j + e;
        if (ks < 48) bq[j & 1][e] = hin[ks * LS32_GS + lane];
      }
      // (lanes c16 >= 8 read the fragment of lane c16 - 8: rows 200 .. 207 of the product come out as copies of rows
      // 192 .. 199 and are never stored -- a select here would put a vector instruction between every load and its MFMA)
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
          d[0] = ls32_mfma(wr[hl][0][ks], bq[j & 1][e], d[0]);
          d[1] = ls32_mfma(wr[hl][1][ks], bq[j & 1][e], d[1]);
          d[2] = ls32_mfma(wr[hl][2][ks], bq[j & 1][e], d[2]);
        }
        if constexpr (e == 1) dx = ls32_mfma(ax[j & 1], bx[j & 1], dx);
      });
      // between two MFMAs of the chunk: at most one LDS read, one LDS write / atomic, one buffer store and three
      // vector instructions -- the next chunk's six fragment reads and whatever the OTHER group's phase in this
      // basic block has ready (its LDS round trips then issue in the shadow of the matrix pipe)
#pragma unroll
      for (int i_ = 0; i_ < (j < 12 ? 13 : 7); ++i_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
    });
#pragma unroll
    for (int r = 0; r < 3; ++r) ls32_epilogue(d[r], wave + 4 * r, hout, G.mask + (hl + 1) * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      G.base[ls32_pick(lane < 32, LY::P12 + ((hl + 1) & 1) * 512 + (wave * 4 + i) * 32 + lane, dmy)] = dx[i];
  };
  // output layer: k-steps 13 wave .. 13 wave + 12, partial sums through LDS; requests the operands of step t + 1
  auto phaseO = [&](auto gic, float* gb, const float* tb, int t) __attribute__((always_inline)) {
    constexpr int gi = decltype(gic)::value;
    const Grp G = grp_at(gb);
    const float* bias_s = tb + LY::BIAS;
    const float* wol = tb + LY::WOL;
    const float* wol1 = tb + LY::WOL1;
    prefetch(gi, min(t + 1, T - 1));           // (the last step requests its own operands again: unused)
    const float* hin = G.actA;
    const float2 tail = ls32_tail(G.p12, bias_s + 2 * LS32_ROWS + 192, G.mask + 2 * 64);
    f32x4_t d[NOB];
#pragma unroll
    for (int blk = 0; blk < NOB; ++blk) d[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bf[13], wo[NOB][13];
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      bf[j] = hin[(13 * wave + j) * LS32_GS + lane];                              // (52 groups)
      wo[0][j] = wol[(wave * 13 + j) * 64 + lane];
      if (NOB > 1) {
        wo[NOB - 1][j] = wol1[((wave * 13 + j) * 4 + g) * NV8 + (c16 & 7)];     // (rows >= 24: copies, never read)
      }
    }
    bf[9] = bsel(wm3, tail.x, bf[9]);                                             // k-steps 48, 49
    bf[10] = bsel(wm3, tail.y, bf[10]);
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
      for (int blk = 0; blk < NOB; ++blk) d[blk] = ls32_mfma(wo[blk][j], bf[j], d[blk]);
#pragma unroll
    for (int blk = 0; blk < NOB; ++blk)
#pragma unroll
      for (int i = 0; i < 4; ++i) G.part[((wave * NOB + blk) * 4 + i) * 64 + lane] = d[blk][i];
  };
  // x_{t+1} = x_t + b_L + the four partials (thread = one (coordinate, candidate) of block 0, some also of block 1);
  // this step's mask words leave, the LDS copy is cleared for the next step; the next step's operands are staged
  auto phaseU = [&](auto gic, float* gb, int t) __attribute__((always_inline)) {
    constexpr int gi = decltype(gic)::value;
    const Grp G = grp_at(gb);
    const float* part = G.part;
    // (coordinates past n: the same sums on in-range addresses, stored nowhere)
    const float v1 = (((part[tid] + part[NOB * 256 + tid]) + (part[2 * NOB * 256 + tid] + part[3 * NOB * 256 + tid])) + bo1) + G.xcur[x1];
    float v2 = 0.f;
    if constexpr (NOB > 1)
      v2 = (((part[pi2] + part[NOB * 256 + pi2]) + (part[2 * NOB * 256 + pi2] + part[3 * NOB * 256 + pi2])) + bo2) + G.xcur[x2];
    // this step's mask words (thread: words tid and tid + 256 of the [16][24] block): eight nibble bytes -- or the
    // eight bit bytes of rows 192 .. 199 -- packed
    const bool hi = tid < LS32_C * 24 - 256;
    auto mword = [&](int widx) -> unsigned {
      const int c = widx / 24, w = widx - c * 24, q = w & 7;
      const uint2 by = *reinterpret_cast<const uint2*>(G.mask + c * 192 + (w >> 3) * 64 + (q < 7 ? q : 0) * 8);
      // nibble bytes: (b & 0xF) << 4 j; bit bytes (q == 6): (b & 1) << j
      const unsigned lo = by.x, hi_ = by.y;
      const unsigned nibs = (lo & 0xF) | ((lo >> 4) & 0xF0) | ((lo >> 8) & 0xF00) | ((lo >> 12) & 0xF000) |
                            ((hi_ & 0xF) << 16) | (((hi_ >> 8) & 0xF) << 20) | (((hi_ >> 16) & 0xF) << 24) |
                            (((hi_ >> 24) & 0xF) << 28);
      const unsigned bits = (lo & 1) | ((lo >> 7) & 2) | ((lo >> 14) & 4) | ((lo >> 21) & 8) | ((hi_ & 1) << 4) |
                            (((hi_ >> 8) & 1) << 5) | (((hi_ >> 16) & 1) << 6) | (((hi_ >> 24) & 1) << 7);
      return q < 6 ? nibs : q == 6 ? bits : 0u;
    };
    const unsigned m1 = mword(tid), m2 = mword(hi ? tid + 256 : tid);
    G.base[ls32_pick(on1, LY::XCUR + x1, dmy)] = v1;
    G.base[ls32_pick(on1, LY::DXS + x1, dmy)] = v1 - pfX1;
    if constexpr (NOB > 1) {
      G.base[ls32_pick(on2, LY::XCUR + x2, dmy)] = v2;
      G.base[ls32_pick(on2, LY::DXS + x2, dmy)] = v2 - pfX2;
    }
    stage(G);
    const unsigned xo = (CI(gi, c16) * (unsigned)(T + 1) + (unsigned)(t + 1)) * (unsigned)n;
    ls32_store_if(rsX, xo + (unsigned)no1, __float_as_uint(v1), in16[gi] && on1);
    if constexpr (NOB > 1) ls32_store_if(rsX, xo + (unsigned)(16 + q2), __float_as_uint(v2), in16[gi] && on2);
    {
      const int mc1 = tid / 24, mc2 = min((tid + 256) / 24, LS32_C - 1);
      const unsigned ms = (unsigned)mstride, tw = (unsigned)t * 24u;
      ls32_store_if(rsM, CI(gi, mc1) * ms + tw + (unsigned)(tid - mc1 * 24), m1, INB(gi, mc1));
      ls32_store_if(rsM, CI(gi, mc2) * ms + tw + (unsigned)(tid + 256 - mc2 * 24), m2, hi && INB(gi, mc2));
    }
  };

  // (a slot is one scheduling region: without the fences the scheduler sees slots 4 .. 6 -- one basic block, barriers
  // do not end one -- as a single region of 1,700 instructions and the group solver takes minutes)
#define LS32_SLOT_END() do { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); } while (0)
  using X_ = std::integral_constant<int, 0>;
  using Y_ = std::integral_constant<int, 1>;
  using H1_ = std::integral_constant<int, 0>;
  using H2_ = std::integral_constant<int, 1>;
  // ================= the horizon: group Y four phases behind group X =================
  // A slot = one phase of each group in ONE basic block.  The two groups and the tables come in as __restrict__
  // parameters: inlined, that becomes scoped no-alias information on every LDS access of the slot -- without it the
  // opaque group offsets leave the compiler unable to tell group X's stores from group Y's loads, and the latency
  // chain of one group is pinned behind the epilogue stores of the other's matrix-pipe phase.
  auto slot1 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb, int t)
      __attribute__((always_inline)) { phaseC(X_{}, gx, t); phaseO(Y_{}, gy, tb, t - 1); };
  auto slot2 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb, int t)
      __attribute__((always_inline)) { phaseL0(gx, tb); phaseU(Y_{}, gy, t - 1); };
  auto slot3 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb, int t)
      __attribute__((always_inline)) { phaseH(H1_{}, gx, tb); phaseC(Y_{}, gy, t); };
  auto slot4 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb)
      __attribute__((always_inline)) { phaseH(H2_{}, gx, tb); phaseL0(gy, tb); };
  auto slot5 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb, int t)
      __attribute__((always_inline)) { phaseH(H1_{}, gy, tb); phaseO(X_{}, gx, tb, t); };
  auto slot6 = [&](float* __restrict__ gx, float* __restrict__ gy, const float* __restrict__ tb, int t)
      __attribute__((always_inline)) { phaseH(H2_{}, gy, tb); phaseU(X_{}, gx, t); };
  for (int t = 0; t < T; ++t) {
    if (t > 0) {
      slot1(gbase(0), gbase(1), tabs, t);
    } else {
      phaseC(X_{}, gbase(0), t);
    }
    LS32_SLOT_END();
    if (t > 0) {
      slot2(gbase(0), gbase(1), tabs, t);
    } else {
      phaseL0(gbase(0), tabs);
    }
    LS32_SLOT_END();
    slot3(gbase(0), gbase(1), tabs, t);
    LS32_SLOT_END();
    slot4(gbase(0), gbase(1), tabs);
    LS32_SLOT_END();
    slot5(gbase(0), gbase(1), tabs, t);
    LS32_SLOT_END();
    slot6(gbase(0), gbase(1), tabs, t);
    LS32_SLOT_END();
  }
  phaseO(Y_{}, gbase(1), tabs, T - 1);
  __syncthreads();
  phaseU(Y_{}, gbase(1), T - 1);
  __syncthreads();

  // ---- stage costs: 4 lanes per (candidate, step) pair, 64 pairs per sweep; summed per candidate in step order
  for (int gi = 0; gi < NG; ++gi) {
    const Grp G = grp(gi);
    float* cst = G.actB;
    const float al = GMPC_ALPHA;
    const int q = tid & 3;
    for (int p = tid >> 2; p < LS32_C * T; p += LS32_THREADS / 4) {
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
  // ---- terminal cost w2 |cost_mlp(x_T)|^2 on the matrix pipe as well: activations [k][16] in actA / actB, weight
  // fragments straight from global memory (row blocks nb = wave, wave + 4, ..)
  for (int gi = 0; gi < NG; ++gi) {
    const Grp G = grp(gi);
    float* in = G.actA;
    float* out = G.actB;
    for (int e = tid; e < LS32_C * ((n + 3) & ~3); e += LS32_THREADS) {
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
    if (tid < LS32_C && INB(gi, tid)) {
      const int fo = a.cost.dims[Lc + 1];
      float yy = 0.f;
      for (int r = 0; r < fo; ++r) {
        const float y = in[ls32_at(r, tid)];
        yy = fmaf(y, y, yy);
      }
      a.objc[CI(gi, tid)] = s_obj[gi][tid] + w2 * yy;
    }
    __syncthreads();
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
  if (!gmpc_ls16_shape(a)) return false;
  const int k0s = (a.n + a.m + 3) / 4;
  const long items = (long)a.B * GMPC_LS_ITEMS;
  return a.n <= 24 && a.m * a.n <= (k0s <= 4 && a.n <= 16 ? 64 : 128) && ls32_lds(a.n, a.m) <= LS32_LDS_MAX &&
         LS32_C * a.T <= LS32_ACT && items * (a.T + 1) * a.n < (1L << 31) && items * a.T * 3 * GMPC_MW < (1L << 31);
}
// work lists of at least this many candidates are k_ls32's.  OFF unless GMPC_LS32_SPLIT is set (the tests set it to
// 1): measured in round 4 (profiles/EXPERIMENTS.md), the 8192 candidates of a first round take one pass of this kernel
// 0.93 ms in its plain two-group form -- what two passes of k_ls16 take -- and 1.2 ms in the present, branch-free form
// (hipcc spills 54 registers of the 512 into scratch inside the time loop), so k_ls16 keeps every long work list.
int gmpc_ls32_split() {
  const char* e = getenv("GMPC_LS32_SPLIT");
  return e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
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
