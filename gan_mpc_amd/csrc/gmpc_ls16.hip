// Line-search candidates, 16 per workgroup (k_ls16<KH, K0S, NOB>, 256 threads): the throughput form of
// k_traj_rw<true> for rounds whose work list is long.
//
// k_traj_rw keeps the dynamics network's matrices in the registers of 4 waves and feeds the matrix pipe a
// 4-wide operand (4 candidates per workgroup): at 1024 trajectories a round of the line search holds 4096 to
// 8192 candidates, i.e. 4 to 8 passes over the 256 CUs.  Here the 16 candidates of a workgroup are the
// 16-column operand of v_mfma_f32_16x16x4_f32 and the weights are the 16-row operand:
//   A operand:  W[k = 4 ks + (lane >> 4)][neuron 16 nb + (lane & 15)], one register per (row block nb, k-step
//               ks).  Row blocks 0..11 of a 200-wide layer are dealt to the waves 0,1,2,3,0,1,.. and stay in
//               registers for the whole horizon (3 blocks x 50 k-steps x 2 hidden layers = 300 registers per
//               lane, half of them pinned to the accumulation file); block 12 (neurons 192..199) is split over
//               the waves by k-step, its A fragments come from LDS, its four partial results are summed by
//               whoever consumes rows 192..199 (ls16_tail) -- 163 MFMAs per wave and layer instead of 200 on
//               one wave and 150 on the others.  Output-layer fragments (k-steps split over the waves, partial
//               sums through LDS) come from LDS as well.
//   B operand:  act[k = 4 ks + (lane >> 4)][candidate lane & 15]: with the activations stored [k][16] the
//               fragment of k-step ks is 64 consecutive floats (groups of 4 rows are 80 floats apart so that
//               the epilogue's stores of rows 4 g + i are conflict-free as well); read one chunk of 4 k-steps
//               ahead of the MFMAs that use it
//   D:          register i of lane (g, c) = neuron 16 nb + 4 g + i of candidate c -> bias (accumulator
//               init), relu, 4 LDS stores; the relu bits go into the candidate's mask words in LDS with one
//               ds_or_b32 per row block, the 24 words of a candidate and step leave with coalesced stores
// The MFMA work per candidate is that of the 4x4x1 form (which wastes nothing either: 16 blocks x 4 neurons x
// 4 candidates); what changes is that the per-step chain -- controls, barriers, epilogues, the state update --
// is shared by 16 candidates instead of 4: 1.22 k instead of 2.0 k cycles per candidate-step, and the 4096
// candidates of a round are ONE pass over the chip.  Short work lists (the tail rounds, small batches) stay
// on k_traj_rw: the launcher starts both kernels and each returns at once when the round's count is on the
// other's side of TrajArgs::ls_split.
//
// Reference arithmetic: dynamics/nn.py:27-34, cost/cost_model.py:20-42, cost/nn.py:23-29, trajax
// line_search_ddp / ddp_rollout (u = U + alpha k + K (x - X)) as called from policy/optimizers.py:19.
#include "gmpc_device.h"
#include <cstdlib>
#include <cstring>

#define LS16_THREADS 256
#define LS16_C 16          // candidates per workgroup
#define LS16_KH 200        // hidden width
#define LS16_KS 50         // k-steps of a hidden layer
#define LS16_NB 13         // row blocks of a hidden layer
#define LS16_GS 80         // floats between groups of 4 activation rows
#define LS16_ROWS 208      // activation rows (13 blocks)
#define LS16_LDS_MAX (150 * 1024)

__device__ __forceinline__ f32x4_t ls16_mfma(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// float index of activation row k, candidate c
__device__ __forceinline__ int ls16_at(int k, int c) { return (k >> 2) * LS16_GS + (k & 3) * 16 + c; }

// epilogue of row block nb < 12 (the bias is in the accumulator): relu, the next layer's activations, the relu
// bits of rows 16 nb + 4 g + i OR-ed into the candidate's mask word (mw: word 0 of this layer, 24 words per
// candidate)
__device__ __forceinline__ void ls16_epilogue(f32x4_t d, int nb, float* out, unsigned* mw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  unsigned nib = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = d[i] > 0.f;
    nib |= on ? (1u << i) : 0u;
    out[(4 * nb + g) * LS16_GS + i * 16 + c] = on ? d[i] : 0.f;
  }
  atomicOr(mw + c * 24 + (nb >> 1), nib << (16 * (nb & 1) + 4 * g));
}

// rows 192 + g (.x) and 196 + g (.y) of a hidden layer's output for candidate lane & 15 -- the B fragments of
// k-steps 48 and 49 -- from the K-split partials of row block 12 ([4 waves][4][64]); wave 0 also records their
// relu bits (mask word 6 of the layer)
__device__ __forceinline__ float2 ls16_tail(const float* p12, const float* bias192, unsigned* mw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const float* q = p12 + g * 64 + c;
  const float s0 = ((q[0] + q[256]) + (q[512] + q[768])) + bias192[g];
  const float s1 = ((q[16] + q[272]) + (q[528] + q[784])) + bias192[4 + g];
  if (threadIdx.x < 64) atomicOr(mw + c * 24 + 6, ((s0 > 0.f ? 1u : 0u) << g) | ((s1 > 0.f ? 1u : 0u) << (4 + g)));
  return make_float2(fmaxf(s0, 0.f), fmaxf(s1, 0.f));
}

#ifdef GMPC_LS_ABORT_STATS
// experiment: how early could a rejected candidate have been dropped (costs are >= 0, so a candidate whose running
// cost has reached the objective to beat is rejected)?  [0] candidates, [1] sum of T, [2] sum over candidates of
// the first step whose running cost reaches the objective (T + 1: never), [3] the same with the workgroup's
// maximum for each of its candidates
__device__ unsigned long long g_ls_abort_stats[4];
extern "C" int gmpc_debug_abort_stats(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_abort_stats), sizeof(g_ls_abort_stats));
}
#endif

// KH: hidden width -- 200 (the form described above: 12 full row blocks + the K-split block 12), or 128 / 64 (round 4:
// 8 / 4 full row blocks dealt to the four waves, no partial block, 32 / 16 k-steps per layer; everything else is the
// same code); K0S: k-steps of layer 0 (n + m <= 4 K0S); NOB: 16-row blocks of the output layer (n <= 16 NOB)
template <int KH, int K0S, int NOB>
__global__ __launch_bounds__(LS16_THREADS, 1) void k_ls16(TrajArgs a) {
  static_assert(KH == 200 || KH == 128 || KH == 64, "hidden width");
  constexpr bool TAILB = KH == 200;                         // rows 192 .. 199: the K-split block 12
  constexpr int NBW = KH == 200 ? 3 : KH / 64;              // full row blocks per wave (wave, wave + 4, ..)
  constexpr int KS = KH / 4;                                // k-steps of a hidden layer
  constexpr int NCH = (KS + 3) / 4;                         // chunks of 4 k-steps
  constexpr int KSW = (KS + 3) / 4;                         // k-steps of the output layer per wave
  constexpr int ACT = (LS16_ROWS / 4) * LS16_GS;            // floats of one activation buffer
  extern __shared__ __attribute__((aligned(16))) char smem_ls16[];
  float* const xcur = reinterpret_cast<float*>(smem_ls16);  // rows x ; u ; 0 (layer-0 input), 8 groups
  float* const actA = xcur + 8 * LS16_GS;
  float* const actB = actA + ACT;
  float* const wxl = actB + ACT;                            // [2 layers][52 k-steps][64] A fragments of block 12
  float* const part = wxl + 2 * 52 * 64;                    // [4 waves][NOB][4][64] output-layer partials
  float* const wol = part + 4 * NOB * 256;                  // [4 waves][NOB][13][64] A fragments of the output layer
  float* const part12 = wol + 4 * NOB * 13 * 64;            // [2][4 waves][4][64] block-12 partials
  float* const bias_s = part12 + 2 * 1024;                  // [3][208] hidden biases, [32] output bias
  float* const cst = bias_s + 3 * LS16_ROWS + 32;           // [16][T] stage costs
  unsigned* const mask_s = reinterpret_cast<unsigned*>(cst + LS16_C * a.T);   // [16][3 layers][8] mask words
  // operands of the step's controls, staged one step ahead: gains [16][m][n] (+ slack for the clamped tail
  // reads), k and U [16][m] each, x - X_nominal in the layout of xcur
  float* const Ks = reinterpret_cast<float*>(mask_s + LS16_C * 24);
  float* const kUs = Ks + LS16_C * a.m * a.n + 32;
  float* const dxs = kUs + 2 * LS16_C * 8;
  __shared__ float s_alpha[LS16_C], s_obj[LS16_C];
  __shared__ int s_bi[LS16_C], s_in[LS16_C];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
  const int n = a.n, m = a.m, T = a.T;
#ifdef GMPC_TRAJ_STAMPS
  const unsigned long long k0_ = __builtin_readcyclecounter(), w0_ = wall_clock64();
#endif
  const int cnt = *a.nitems;
  if (cnt < a.ls_split) return;                 // short work list: k_traj_rw's round
  if (a.ls32_split > 0 && cnt >= a.ls32_split) return;      // more than one pass over the chip: k_ls32's round
  const int b0 = blockIdx.x * LS16_C;
  if (b0 >= cnt) return;
  if (tid < LS16_C) {
    const int it = min(b0 + tid, cnt - 1);
    s_bi[tid] = a.item_b[it];
    s_in[tid] = (b0 + tid) < cnt;
    float al = a.alpha_0;
    for (int k = a.item_k[it]; k > 0; --k) al *= 0.5f;
    s_alpha[tid] = al;
  }
  constexpr int Lh = 3;
  const size_t mstride = (size_t)T * Lh * GMPC_MW;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]), w2 = sigmoidf_(a.mpc_w[2]);

  // ---- weights: registers for the whole horizon (row blocks wave, wave + 4, wave + 8; block 12 is split
  // over the waves by k-step, k-steps wave + 4 j)
  float wr[2][NBW][KS];
  float w0r[NBW][K0S], w0x[2] = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < NBW; ++r) {
    const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
    for (int ks = 0; ks < K0S; ++ks) {
      const int k = 4 * ks + g;
      w0r[r][ks] = k < n + m ? a.dyn.W[0][(size_t)k * KH + nn] : 0.f;
    }
  }
  if constexpr (TAILB) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 4 * (wave + 4 * q) + g, nn = 192 + c16;
      w0x[q] = (k < n + m && nn < KH) ? a.dyn.W[0][(size_t)k * KH + nn] : 0.f;
    }
  }
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const float* Wl = a.dyn.W[hl + 1];
#pragma unroll
    for (int r = 0; r < NBW; ++r) {
      const int nn = 16 * (wave + 4 * r) + c16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wr[hl][r][ks] = Wl[(size_t)(4 * ks + g) * KH + nn];
    }
    if constexpr (TAILB)
      for (int e = tid; e < 52 * 64; e += LS16_THREADS) {
        const int ks = e >> 6, l = e & 63, nn = 192 + (l & 15);
        wxl[hl * 52 * 64 + e] = (ks < KS && nn < KH) ? Wl[(size_t)(4 * ks + (l >> 4)) * KH + nn] : 0.f;
      }
  }
  // half of the 320 weight registers has to live in the accumulation file: say which half, so that the MFMAs
  // read them there (left to itself the allocator parks them there and copies each one back into the
  // architectural file in front of its MFMA: +17 cycles per MFMA for the whole layer).  After all the loads
  // have been issued: the statements below keep their order and each one waits for its operand.
  if constexpr (TAILB) {
#pragma unroll
    for (int r = 0; r < NBW; ++r)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(wr[0][r][ks]));
  }
  // output layer (k-steps KSW wave + j of wave `wave`): fragments in LDS, the registers are taken
  for (int e = tid; e < 4 * NOB * KSW * 64; e += LS16_THREADS) {
    const int l = e & 63, j = (e >> 6) % KSW, wb = (e >> 6) / KSW;      // wb = wave * NOB + blk
    const int ks = KSW * (wb / NOB) + j, no = 16 * (wb % NOB) + (l & 15);
    wol[e] = (ks < KS && no < n) ? a.dyn.W[Lh][(size_t)(4 * ks + (l >> 4)) * n + no] : 0.f;
  }
  for (int e = tid; e < 3 * LS16_ROWS + 32; e += LS16_THREADS) {
    float v = 0.f;
    if (e < 3 * LS16_ROWS) {
      const int l = e / LS16_ROWS, j = e - l * LS16_ROWS;
      if (j < KH) v = a.dyn.b[l][j];
    } else if (e - 3 * LS16_ROWS < n) {
      v = a.dyn.b[Lh][e - 3 * LS16_ROWS];
    }
    bias_s[e] = v;
  }
  // xcur rows >= n + m, activation rows 192.. (never stored: rows 192..199 are rebuilt from the partials,
  // 200..207 are padding) and the mask words start at zero
  for (int e = tid; e < 8 * LS16_GS + 2 * ACT; e += LS16_THREADS) xcur[e] = 0.f;
  for (int e = tid; e < LS16_C * 24; e += LS16_THREADS) mask_s[e] = 0u;
  for (int e = tid; e < LS16_C * a.m * a.n + 32 + 2 * LS16_C * 8 + 8 * LS16_GS; e += LS16_THREADS) Ks[e] = 0.f;
  __syncthreads();
  auto BI = [&](int c) -> int { return s_bi[c]; };
  auto INB = [&](int c) -> bool { return s_in[c] != 0; };
  auto CI = [&](int c) -> size_t { return (size_t)(b0 + c); };
  // ---- initial state: the nominal trajectory's x_0
  for (int e = tid; e < LS16_C * n; e += LS16_THREADS) {
    const int c = e / n, i = e - c * n;
    xcur[ls16_at(i, c)] = a.X[(size_t)BI(c) * (T + 1) * n + i];
  }

  // ---- per-thread roles, fixed for the horizon
  // controls u = U + alpha k + K (x - X_nominal): two lanes per (candidate, control) pair (elements i = half
  // + 2 e); the operands come from LDS, where the workgroup stages them one step ahead with coalesced loads
  // (the [m][n] gain block of a trajectory and step is contiguous)
  constexpr int PE = 2 * K0S;
  constexpr int KQ = K0S == 4 ? 4 : 8;          // 16 m n <= 256 KQ
  const int MN = m * n;
  const int cp = tid >> 1, chalf = tid & 1;
  const int cc = cp / m, cj = cp - cc * m;
  const bool con = cp < LS16_C * m;
  float* pUc = (con && chalf == 0 && INB(cc)) ? a.Uc + CI(cc) * T * m + cj : nullptr;
  const float calpha = con ? s_alpha[cc] : 0.f;
  const float* const kcb = Ks + (con ? cc * MN + cj * n : 0) + chalf;
  const float* const dcb = dxs + chalf * 16 + (con ? cc : 0);
  // element tid + 256 q of the gain blocks, pair tid of k and U: offsets from the step's base
  unsigned koff[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int e = tid + LS16_THREADS * q, c = min(e / MN, LS16_C - 1);
    koff[q] = e < LS16_C * MN ? (unsigned)BI(c) * (unsigned)(T * MN) + (unsigned)(e - c * MN) : 0u;
  }
  const bool kuon = tid < LS16_C * m;
  const unsigned kuoff = kuon ? (unsigned)BI(tid / m) * (unsigned)(T * m) + (unsigned)(tid % m) : 0u;
  // state update: thread (wave i, lane (g, c)) owns coordinate 4 g + i of candidate c (output block 0);
  // threads < 16 (n - 16) also own coordinate 16 + tid / 16 of candidate tid & 15 (output block 1)
  const int no1 = 4 * g + wave;
  const bool on1 = no1 < n;
  float* pX1 = (on1 && INB(c16)) ? a.Xc + (CI(c16) * (T + 1) + 1) * n + no1 : nullptr;
  const int x1 = ls16_at(no1, c16);
  const float bo1 = on1 ? bias_s[3 * LS16_ROWS + no1] : 0.f;
  const int q2 = tid >> 4;
  const bool on2 = NOB > 1 && 16 + q2 < n;
  float* pX2 = (on2 && INB(c16)) ? a.Xc + (CI(c16) * (T + 1) + 1) * n + 16 + q2 : nullptr;
  const int x2 = ls16_at(on2 ? 16 + q2 : 0, c16);
  const int pi2 = 256 + (q2 & 3) * 64 + 16 * ((q2 >> 2) & 3) + c16;
  const float bo2 = on2 ? bias_s[3 * LS16_ROWS + 16 + q2] : 0.f;
  const unsigned xo1 = on1 ? (unsigned)BI(c16) * (unsigned)((T + 1) * n) + no1 : 0u;
  const unsigned xo2 = on2 ? (unsigned)BI(c16) * (unsigned)((T + 1) * n) + 16 + q2 : 0u;
  // mask rows: words tid and tid + 256 of the [16][24] block
  const int mc1 = tid / 24, mc2 = (tid + 256) / 24;
  uint32_t* pM1 = INB(mc1) ? a.maskc + CI(mc1) * mstride + (tid - mc1 * 24) : nullptr;
  uint32_t* pM2 = (tid < LS16_C * 24 - 256 && INB(mc2)) ? a.maskc + CI(mc2) * mstride + (tid + 256 - mc2 * 24)
                                                          : nullptr;

  float pfK[KQ], pfk = 0.f, pfU = 0.f, pfX1 = 0.f, pfX2 = 0.f;
  // loads of step t's operands (uniform base + per-thread offset)
  auto prefetch = [&](int t) {
    const float* Kt = a.Kg + (size_t)t * MN;
#pragma unroll
    for (int q = 0; q < KQ; ++q) pfK[q] = Kt[koff[q]];
    pfk = a.kg[(size_t)t * m + kuoff];
    pfU = a.Uio[(size_t)t * m + kuoff];
    pfX1 = a.X[(size_t)t * n + xo1];
    if (NOB > 1) pfX2 = a.X[(size_t)t * n + xo2];
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      if (tid + LS16_THREADS * q < LS16_C * MN) Ks[tid + LS16_THREADS * q] = pfK[q];
    if (kuon) { kUs[tid] = pfk; kUs[LS16_C * 8 + tid] = pfU; }
  };
  prefetch(0);
  stage();                                      // (dxs = x_0 - X_0 = 0: zeroed above)
  __syncthreads();                              // xcur
#ifdef GMPC_TRAJ_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
#define TS_(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define TS_(i)
#endif

  for (int t = 0; t < T; ++t) {
    // ---- controls
    if (con) {
      float du = 0.f;
#pragma unroll
      for (int e = 0; e < PE; ++e) {
        // row i = chalf + 2 e of dxs: group e / 2, row (e & 1) 2 + chalf of the group
        const float dx = dcb[(e >> 1) * LS16_GS + (e & 1) * 32];
        du = fmaf(kcb[2 * e], chalf + 2 * e < n ? dx : 0.f, du);
      }
      du += __shfl_xor(du, 1);
      const float u = kUs[LS16_C * 8 + cp] + fmaf(calpha, kUs[cp], du);
      if (chalf == 0) {
        if (pUc != nullptr) { *pUc = u; pUc += m; }
        xcur[ls16_at(n + cj, cc)] = u;
      }
    }
    __syncthreads();
    TS_(0)
    if (t + 1 < T) prefetch(t + 1);             // in flight while the network runs
    // ---- layer 0
    {
      f32x4_t d[NBW], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < NBW; ++r) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + 16 * (wave + 4 * r) + 4 * g);
        d[r] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
      }
      float bf[K0S];
#pragma unroll
      for (int ks = 0; ks < K0S; ++ks) bf[ks] = xcur[ks * LS16_GS + lane];
      const float bx0 = xcur[wave * LS16_GS + lane], bx1 = xcur[(wave + 4) * LS16_GS + lane];
#pragma unroll
      for (int ks = 0; ks < K0S; ++ks)
#pragma unroll
        for (int r = 0; r < NBW; ++r) d[r] = ls16_mfma(w0r[r][ks], bf[ks], d[r]);
      if constexpr (TAILB) {
        dx = ls16_mfma(w0x[0], bx0, dx);
        dx = ls16_mfma(w0x[1], bx1, dx);
      }
#pragma unroll
      for (int r = 0; r < NBW; ++r) ls16_epilogue(d[r], wave + 4 * r, actA, mask_s);
      if constexpr (TAILB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part12[(wave * 4 + i) * 64 + lane] = dx[i];
      }
    }
    __syncthreads();
    TS_(1)
    // ---- hidden layers 1, 2
    float* hin = actA;
    float* hout = actB;
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      float2 tail = make_float2(0.f, 0.f);
      if constexpr (TAILB) tail = ls16_tail(part12 + (hl & 1) * 1024, bias_s + hl * LS16_ROWS + 192, mask_s + hl * 8);
      f32x4_t d[NBW], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < NBW; ++r) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + (hl + 1) * LS16_ROWS + 16 * (wave + 4 * r) + 4 * g);
        d[r] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
      }
      // chunks of 4 k-steps, operands of chunk j + 1 read while chunk j multiplies: 4 B fragments, and (KH = 200) the
      // A / B fragments of this wave's block-12 k-step 4 j + wave
      const float* wx = wxl + hl * 52 * 64 + wave * 64 + lane;
      const float* hx = hin + wave * LS16_GS + lane;
      float bq[2][4], ax[2] = {0.f, 0.f}, bx[2] = {0.f, 0.f};
      auto load_chunk = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ks = 4 * j + e;
          if (ks < (TAILB ? 48 : KS)) bq[j & 1][e] = hin[ks * LS16_GS + lane];
        }
        if constexpr (TAILB) {
          ax[j & 1] = wx[4 * j * 64];
          if (j < 12) bx[j & 1] = hx[4 * j * LS16_GS];
        }
      };
      load_chunk(std::integral_constant<int, 0>{});
      rw_static_for<NCH>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j + 1 < NCH) load_chunk(std::integral_constant<int, j + 1>{});
        if constexpr (TAILB && j == 12) {
          bq[0][0] = tail.x;
          bq[0][1] = tail.y;
          bx[0] = wave == 0 ? tail.x : wave == 1 ? tail.y : 0.f;
        }
        rw_static_for<4>([&](auto ec) __attribute__((always_inline)) {
          constexpr int e = decltype(ec)::value;
          constexpr int ks = 4 * j + e;
          if constexpr (ks < KS) {
#pragma unroll
            for (int r = 0; r < NBW; ++r) d[r] = ls16_mfma(wr[hl][r][ks], bq[j & 1][e], d[r]);
          }
          if constexpr (TAILB && e == 1) dx = ls16_mfma(ax[j & 1], bx[j & 1], dx);
        });
        __builtin_amdgcn_sched_group_barrier(0x100, TAILB ? 6 : 4, 0);                               // the next chunk's LDS reads
        __builtin_amdgcn_sched_group_barrier(0x008, TAILB ? (j < 12 ? 13 : 7) : 4 * NBW, 0);         // this chunk's MFMAs
      });
#pragma unroll
      for (int r = 0; r < NBW; ++r) ls16_epilogue(d[r], wave + 4 * r, hout, mask_s + (hl + 1) * 8);
      if constexpr (TAILB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part12[((hl + 1) & 1) * 1024 + (wave * 4 + i) * 64 + lane] = dx[i];
      }
      __syncthreads();
      TS_(2 + hl)
      float* tmp = hin; hin = hout; hout = tmp;
    }
    // ---- output layer: k-steps 13 wave .. 13 wave + 12, partial sums through LDS
    {
      float2 tail = make_float2(0.f, 0.f);
      if constexpr (TAILB) tail = ls16_tail(part12, bias_s + 2 * LS16_ROWS + 192, mask_s + 2 * 8);
      f32x4_t d[NOB];
#pragma unroll
      for (int blk = 0; blk < NOB; ++blk) d[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      float bf[KSW], wo[NOB][KSW];
#pragma unroll
      for (int j = 0; j < KSW; ++j) {
        bf[j] = hin[(KSW * wave + j) * LS16_GS + lane];                             // (KH = 200: 52 groups)
#pragma unroll
        for (int blk = 0; blk < NOB; ++blk) wo[blk][j] = wol[((wave * NOB + blk) * KSW + j) * 64 + lane];
      }
      if constexpr (TAILB) {
        if (wave == 3) { bf[9] = tail.x; bf[10] = tail.y; }                         // k-steps 48, 49
      }
#pragma unroll
      for (int j = 0; j < KSW; ++j)
#pragma unroll
        for (int blk = 0; blk < NOB; ++blk) d[blk] = ls16_mfma(wo[blk][j], bf[j], d[blk]);
#pragma unroll
      for (int blk = 0; blk < NOB; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[((wave * NOB + blk) * 4 + i) * 64 + lane] = d[blk][i];
    }
    __syncthreads();
    TS_(4)
    {
      // x_{t+1} = x_t + b_L + the four partials (thread = one (coordinate, candidate) of block 0, some also of
      // block 1); this step's mask words leave, the LDS copy is cleared for the next step
      float v1 = 0.f, v2 = 0.f;
      if (on1) v1 = (((part[tid] + part[NOB * 256 + tid]) + (part[2 * NOB * 256 + tid] + part[3 * NOB * 256 + tid])) + bo1) + xcur[x1];
      if (on2) v2 = (((part[pi2] + part[NOB * 256 + pi2]) + (part[2 * NOB * 256 + pi2] + part[3 * NOB * 256 + pi2])) + bo2) + xcur[x2];
      const unsigned m1 = mask_s[tid], m2 = tid < LS16_C * 24 - 256 ? mask_s[tid + 256] : 0u;
      if (on1) { xcur[x1] = v1; dxs[x1] = v1 - pfX1; }
      if (on2) { xcur[x2] = v2; dxs[x2] = v2 - pfX2; }
      stage();
      if (pX1 != nullptr) { *pX1 = v1; pX1 += n; }
      if (pX2 != nullptr) { *pX2 = v2; pX2 += n; }
      if (pM1 != nullptr) { *pM1 = m1; pM1 += 24; }
      if (pM2 != nullptr) { *pM2 = m2; pM2 += 24; }
      mask_s[tid] = 0u;
      if (tid < LS16_C * 24 - 256) mask_s[tid + 256] = 0u;
    }
    __syncthreads();
    TS_(5)
  }
#ifdef GMPC_TRAJ_STAMPS
  const unsigned long long k1_ = __builtin_readcyclecounter(), w1_ = wall_clock64();
  if (blockIdx.x == 0 && tid == 0)
    printf("k_ls16 cycles per step: controls %llu L0 %llu L1 %llu L2 %llu out %llu reduce %llu | setup+loop cycles %llu wall(100MHz) %llu\n", st_[0] / T,
           st_[1] / T, st_[2] / T, st_[3] / T, st_[4] / T, st_[5] / T, k1_ - k0_, w1_ - w0_);
#endif
  // ---- stage costs: 4 lanes per (candidate, step) pair, 64 pairs per sweep; summed per candidate in step order
  {
    const float al = GMPC_ALPHA;
    const int q = tid & 3;
    for (int p = tid >> 2; p < LS16_C * T; p += LS16_THREADS / 4) {
      const int c = p / T, t = p - c * T;
      const int bc = BI(c);
      const size_t ci = INB(c) ? CI(c) : 0;          // (unused candidates read item 0's rows: in bounds, discarded)
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
      if (q == 0) cst[p] = INB(c) ? w0 * (sqrtf(uu + al * al) - al) + w1 * (sqrtf(dd + al * al) - al) : 0.f;
    }
    __syncthreads();
    if (tid < LS16_C) {
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += cst[tid * T + t];
      s_obj[tid] = acc;
#ifdef GMPC_LS_ABORT_STATS
      {
        const float oo = a.obj[BI(tid)];
        float run = 0.f;
        int tx = T + 1;
        for (int t = 0; t < T; ++t) { run += cst[tid * T + t]; if (run >= oo && tx > T) tx = t + 1; }
        if (!INB(tid)) tx = 0;
        int mx = tx;
        for (int o = 8; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if (INB(tid)) {
          atomicAdd(&g_ls_abort_stats[0], 1ull);
          atomicAdd(&g_ls_abort_stats[1], (unsigned long long)T);
          atomicAdd(&g_ls_abort_stats[2], (unsigned long long)min(tx, T));
          atomicAdd(&g_ls_abort_stats[3], (unsigned long long)min(mx, T));
        }
      }
#endif
    }
  }
  // ---- terminal cost w2 |cost_mlp(x_T)|^2 on the matrix pipe as well: activations [k][16] in actA / actB,
  // weight fragments straight from global memory (row blocks nb = wave, wave + 4, ..)
  {
    float* in = actA;
    float* out = actB;
    for (int e = tid; e < LS16_C * ((n + 3) & ~3); e += LS16_THREADS) {
      const int i = e >> 4, c = e & 15;
      in[ls16_at(i, c)] = i < n ? xcur[ls16_at(i, c)] : 0.f;
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
            bq[e] = in[ks * LS16_GS + lane];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = ls16_mfma(wv[e], bq[e], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = l < Lc ? fmaxf(acc[i], 0.f) : acc[i];
          if (16 * nb + 4 * g + i >= fo) v = 0.f;
          out[(4 * nb + g) * LS16_GS + i * 16 + c16] = v;
        }
      }
      __syncthreads();
      float* tmp = in; in = out; out = tmp;
    }
    if (tid < LS16_C && INB(tid)) {
      const int fo = a.cost.dims[Lc + 1];
      float yy = 0.f;
      for (int r = 0; r < fo; ++r) {
        const float y = in[ls16_at(r, tid)];
        yy = fmaf(y, y, yy);
      }
      a.objc[CI(tid)] = s_obj[tid] + w2 * yy;
    }
  }
#ifdef GMPC_TRAJ_STAMPS
  if (blockIdx.x == 0 && tid == 0)
    printf("k_ls16 whole kernel: cycles %llu wall(100MHz) %llu\n", __builtin_readcyclecounter() - k0_, wall_clock64() - w0_);
#endif
}

// ---- host side --------------------------------------------------------------------------------------
bool gmpc_traj_rw_shape(const TrajArgs& a);
static size_t ls16_lds(int nob, int T);

// shapes k_ls16 is instantiated for: the register-weight shapes with n + m <= 24, m <= 8 and a cost network
// whose layers fit the activation buffers
bool gmpc_ls16_shape(const TrajArgs& a) {
  const char* e = getenv("GMPC_LS");        // read per call: the tests switch forms inside one process
  const bool off = e != nullptr && strcmp(e, "rw") == 0;
  // (three hidden layers of 200, 128 or 64 -- the widths of the register-weight rollouts)
  const int kh = a.dyn.dims[1];
  if (off || !gmpc_traj_rw_shape(a) || (kh != 200 && kh != 128 && kh != 64) || a.n + a.m > 24 || a.m > 8 || a.n > 32 ||
      ls16_lds(a.n > 16 ? 2 : 1, a.T) > LS16_LDS_MAX)
    return false;
  for (int l = 0; l <= a.cost.L; ++l)
    if (a.cost.dims[l] > LS16_ROWS) return false;
  return true;
}
// work lists shorter than this stay on k_traj_rw (4 candidates per workgroup fill the chip sooner)
int gmpc_ls16_split() {
  const char* e = getenv("GMPC_LS16_SPLIT");
  return e != nullptr && atoi(e) > 0 ? atoi(e) : 1537;
}

static size_t ls16_lds(int nob, int T) {
  const size_t fl = 8 * LS16_GS + 2 * (LS16_ROWS / 4) * LS16_GS + 2 * 52 * 64 + 4 * nob * 256 +
                    4 * nob * 13 * 64 + 2 * 1024 + 3 * LS16_ROWS + 32 + (size_t)LS16_C * T + LS16_C * 24 +
                    LS16_C * 128 + 32 + 2 * LS16_C * 8 + 8 * LS16_GS;
  return fl * sizeof(float);
}
template <int KH, int K0S, int NOB>
static void ls16_launch(const TrajArgs& a, int grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ls16<KH, K0S, NOB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LS16_LDS_MAX);
    (void)hipGetLastError();
    attr = true;
  }
  hipLaunchKernelGGL((k_ls16<KH, K0S, NOB>), dim3(grid), dim3(LS16_THREADS), ls16_lds(NOB, a.T), s, a);
}
template <int KH>
static void ls16_launch_kh(const TrajArgs& a, int grid, hipStream_t s) {
  const int k0s = (a.n + a.m + 3) / 4;
  if (a.n <= 16) {
    if (k0s <= 4) ls16_launch<KH, 4, 1>(a, grid, s);
    else ls16_launch<KH, 6, 1>(a, grid, s);
  } else {
    ls16_launch<KH, 6, 2>(a, grid, s);
  }
}

// one workgroup per 16 work-list items; `max_items` bounds the list (the kernel reads the actual count)
void gmpc_launch_ls16(const TrajArgs& a, long max_items, hipStream_t s) {
  const int grid = (int)((max_items + LS16_C - 1) / LS16_C);
  switch (a.dyn.dims[1]) {
    case 200: ls16_launch_kh<200>(a, grid, s); break;
    case 128: ls16_launch_kh<128>(a, grid, s); break;
    default: ls16_launch_kh<64>(a, grid, s); break;
  }
}
