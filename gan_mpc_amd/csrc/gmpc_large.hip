// Large-state path (n > 64: BASELINE configs C4 Humanoid n=376, C5 synthetic n=1024).
//
// The value matrix P of the Riccati recursion no longer fits a CU's LDS (565 KB at n=376, 4.2 MB at
// n=1024), and the per-step LQR blocks cannot all be materialised either (AB is 15 GB
// per GPU at C4, 460 GB at C5).  The backward pass therefore runs STEP-MAJOR: for t = T-1 .. 0, for
// the whole batch at once,
//     [A_t | B_t]  <- MFMA Jacobian chain on the samples (b, t)          (k_linearize_mfma, strided)
//     [PA | PB] = P [A | B]                 batched fp32-MFMA GEMMs      (k_bgemm_tn_lds / k_bgemm_tn)
//     [H | Gr]  = B^T [PA | PB]                                          (k_bgemm_tn)
//     gains K_t k_t, adjoint, value vector, V = H + G K / 2               (k_big_step)
//     T1 = A^T (PA) + K^T V + V^T K   one GEMM over two K-segments, only the blocks that touch the
//                                     upper triangle                     (k_bgemm_tn_lds)
//     P <- Q_t + T1 (upper triangle, mirrored)                           (k_big_pupdate)
// K^T V + V^T K = K^T H + H^T K + K^T G K are the cross terms of trajax' lqr_step value update; written
// as [K; V]^T [V; K] the product is symmetric term by term, so -- like A^T P A with a symmetric P -- its
// two triangles differ only by rounding and the lower one need not be computed (half of the second
// n^3 product of every step).  trajax symmetrises the sum explicitly; the mirror gives the same
// exactly symmetric P up to that rounding.
// Every product is written as  C = sum_k X[k][:]^T Y[k][:]  ("TN") with row-major operands, so row k
// of X / Y IS the MFMA A / B operand of k-step k and all global reads are coalesced; P's symmetry
// turns P A into that form (X = P).  Reference arithmetic: trajax lqr_step / tvlqr / adjoint.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gmpc_device.h"

// ------------------------------------------------------------------------------------------------
// C[b] = alpha * sum_{k<K} X[b][k][0:M]^T (x) Y[b][k][0:N]  (+ beta * C[b]);  one wave per
// 32 x 32*NTW strip of one batch element.  Y is read up to 6 rows past K and up to 32*NTW-1 columns
// past N (values discarded / multiplied by zero): the caller pads its buffers.
// ------------------------------------------------------------------------------------------------
template <int NTW>
__global__ __launch_bounds__(GMPC_THREADS) void k_bgemm_tn(BgemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int mstrips = (a.M + 31) >> 5, ngroups = (a.N + 32 * NTW - 1) / (32 * NTW);
  const long total = (long)a.batch * mstrips * ngroups;
  const long item = (long)blockIdx.x * (GMPC_THREADS / 64) + wave;
  if (item >= total) return;
  const int b = (int)(item / (mstrips * ngroups));
  const int rem = (int)(item - (long)b * mstrips * ngroups);
  const int mi = rem / ngroups, ng = rem - mi * ngroups;
  if (a.active != nullptr && a.active[b] == 0) return;
  const float* X = a.X + (size_t)b * a.sx;
  const float* Y = a.Y + (size_t)b * a.sy;
  float* C = a.C + (size_t)b * a.sc;
  const int K = a.K, Kp = K & ~1;
  const int acol = mi * 32 + l31;
  const bool aok = acol < a.M;
  const float* ap = X + (aok ? acol : a.M - 1);
  const int ldx = a.ldx;
  auto afn = [&](int k0) -> float {
    const int r = k0 + half;
    const float v = ap[(size_t)min(r, K - 1) * ldx];
    return (aok && r < K) ? v : 0.f;
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[nt][rg] = 0.f;
  const float* bp0 = Y + (size_t)half * a.ldy + ng * 32 * NTW + l31;
  if (Kp > 0) gemm_tile<NTW>(bp0, a.ldy, Kp, afn, acc);
  if (K & 1) {
    // odd K: the last k-step pairs row K-1 with a zero row.  Row K of Y belongs to somebody else
    // (the next batch element or time step) and may hold NaN, which 0 * x would let through.
    const float av = half == 0 ? afn(K - 1) : 0.f;
    const float* yr = Y + (size_t)(K - 1) * a.ldy + ng * 32 * NTW + l31;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const float bv = half == 0 ? yr[nt * 32] : 0.f;
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = ng * 32 * NTW + nt * 32 + l31;
    if (col < a.N) {
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int row = mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
        if (row < a.M) {
          float* cp = C + (size_t)row * a.ldc + col;
          float v = a.alpha * acc[nt][rg];
          if (a.beta != 0.f) v = fmaf(a.beta, *cp, v);
          *cp = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming form of the thin products (k_bthin<WIDE_X>): one operand has at most 32 columns per strip, the
// other one is wide and comes from HBM exactly once per strip (PB = P B and [H | G_r] = B^T [PA | PB] of the
// large-state pass: 290 / 302 MB per time step at the C4 shard).  k_bgemm_tn reads the wide operand one dword
// per lane and k-step with two k-steps in flight: ~1 KB per wave on its way at a time, 1.1-1.7 TB/s.  Here a
// wave owns 128 consecutive columns of the wide operand as FOUR interleaved MFMA tiles -- tile j = columns
// 128 g + 4 i + j, i = 0..31 -- so a lane's 16-byte load of row 2 ks + half IS the operand of the four tiles
// (the matrix instruction does not care which column sits in which tile row as long as the epilogue knows),
// and BT_RD k-steps are in flight (BT_RD KB per wave; 6 or 10 measure the same).  The rows need not be 16-byte aligned (n + m = 393 at
// C4): the loads carry 4-byte alignment, which the memory pipeline of gfx950 serves in its unaligned mode.
// The wide operand is read up to 127 columns past its width in its last column group (clamped to the row's
// last 16 bytes: never out of the matrix), rows past K are not read (clamped, the thin operand is zero there).
// ------------------------------------------------------------------------------------------------
#define BT_RD 10
struct __attribute__((packed, aligned(4))) bt_f4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) bt_f2 { float x, y; };

// NTJ = 2: 64 columns per wave, 8-byte loads (twice the waves: fills the chip when batch * width / 128 does not)
// NS = 2: both 32-column strips of a thin operand of 33..64 columns in one wave (the wide operand is read once)
template <bool WIDE_X, int NTJ, int RD, int NS>
__global__ __launch_bounds__(GMPC_THREADS) void k_bthin(BgemmArgs a) {
  constexpr int GW = 32 * NTJ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int Wd = WIDE_X ? a.M : a.N, Th = WIDE_X ? a.N : a.M;       // wide / thin extents
  const int groups = (Wd + GW - 1) / GW, strips = (Th + 32 * NS - 1) / (32 * NS);
  const long total = (long)a.batch * groups * strips;
  const long item = (long)blockIdx.x * (GMPC_THREADS / 64) + wave;
  if (item >= total) return;
  const int b = (int)(item / (groups * strips));
  const int rem = (int)(item - (long)b * groups * strips);
  const int g = rem / strips, st = rem - g * strips;
  if (a.active != nullptr && a.active[b] == 0) return;
  const float* Wp = (WIDE_X ? a.X + (size_t)b * a.sx : a.Y + (size_t)b * a.sy);
  const float* Tp = (WIDE_X ? a.Y + (size_t)b * a.sy : a.X + (size_t)b * a.sx);
  const int ldw = WIDE_X ? a.ldx : a.ldy, ldt = WIDE_X ? a.ldy : a.ldx;
  const int K = a.K;
  // this lane's NTJ wide columns and its thin column(s)
  const int wc = GW * g + NTJ * l31;
  // lanes past the width read the row's last 16 bytes instead (never out of the matrix); when the width is not
  // a multiple of 4 one lane straddles the edge and finds its columns `sh` places further up in that load
  const int wcl = min(wc, max(Wd - NTJ, 0));
  const int sh = wc - wcl;
  const bool ragged = (Wd & (NTJ - 1)) != 0;        // (uniform)
  const int tc0 = 32 * NS * st + l31;
  const float* wrow = Wp + wcl;
  const float* trow[NS];
  bool tok[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    tok[q] = tc0 + 32 * q < Th;
    trow[q] = Tp + (tok[q] ? tc0 + 32 * q : 0);
  }
  f32x16 acc[NS][NTJ];
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) acc[q][j][rg] = 0.f;
  struct Tq { float v[NS]; };
  bt_f4 wq[RD];
  Tq tq[RD];
  const int KS = (K + 1) >> 1;
  auto issue = [&](int ks, bt_f4& wv, Tq& tv) {
    const int r = min(2 * ks + half, K - 1);
    if (NTJ == 4) {
      wv = *reinterpret_cast<const bt_f4*>(wrow + (size_t)r * ldw);
    } else {
      const bt_f2 q = *reinterpret_cast<const bt_f2*>(wrow + (size_t)r * ldw);
      wv.x = q.x; wv.y = q.y;
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) tv.v[q] = trow[q][(size_t)r * ldt];
  };
  auto mult = [&](int ks, bt_f4 wv, const Tq& tv) {
    const bool rok = 2 * ks + half < K;
    if (ragged) {
      const bt_f4 q = wv;
      if (NTJ == 4) {
        wv.x = sh == 0 ? q.x : sh == 1 ? q.y : sh == 2 ? q.z : q.w;
        wv.y = sh == 0 ? q.y : sh == 1 ? q.z : q.w;
        wv.z = sh == 0 ? q.z : q.w;
      } else {
        wv.x = sh == 0 ? q.x : q.y;
      }
    }
    const float wj[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const float t = (tok[q] && rok) ? tv.v[q] : 0.f;
#pragma unroll
      for (int j = 0; j < NTJ; ++j)
        acc[q][j] = WIDE_X ? __builtin_amdgcn_mfma_f32_32x32x2f32(wj[j], t, acc[q][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x2f32(t, wj[j], acc[q][j], 0, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < RD; ++i) issue(min(i, KS - 1), wq[i], tq[i]);
  int ks = 0;
  for (; ks + RD <= KS; ks += RD) {
#pragma unroll
    for (int i = 0; i < RD; ++i) {
      const bt_f4 wv = wq[i];
      const Tq tv = tq[i];
      issue(min(ks + RD + i, KS - 1), wq[i], tq[i]);      // (past the end: the last k-step again, not used)
      __builtin_amdgcn_sched_barrier(0);
      mult(ks + i, wv, tv);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < RD; ++i)
    if (ks + i < KS) mult(ks + i, wq[i], tq[i]);
  // epilogue: accumulator row i of tile j <-> wide column GW g + NTJ i + j (WIDE_X: a row of C), column l31 <->
  // thin column (WIDE_X) / wide columns GW g + NTJ l31 + j (a run of NTJ floats of row i of C otherwise)
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int tc = tc0 + 32 * q;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int i = (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (WIDE_X) {
#pragma unroll
        for (int j = 0; j < NTJ; ++j) {
          const int row = GW * g + NTJ * i + j;
          if (row < a.M && tok[q]) {
            float* cp = a.C + (size_t)b * a.sc + (size_t)row * a.ldc + tc;
            float v = a.alpha * acc[q][j][rg];
            if (a.beta != 0.f) v = fmaf(a.beta, *cp, v);
            *cp = v;
          }
        }
      } else {
        const int row = 32 * NS * st + 32 * q + i;
        if (row < a.M) {
          float* cp = a.C + (size_t)b * a.sc + (size_t)row * a.ldc + wc;
          if (wc + NTJ <= a.N) {             // the lane's NTJ columns as one store (a row of C is contiguous)
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NTJ; ++j) v[j] = a.alpha * acc[q][j][rg];
            if (NTJ == 4) {
              bt_f4* c4 = reinterpret_cast<bt_f4*>(cp);
              if (a.beta != 0.f) {
                const bt_f4 o = *c4;
                v[0] = fmaf(a.beta, o.x, v[0]); v[1] = fmaf(a.beta, o.y, v[1]);
                v[2] = fmaf(a.beta, o.z, v[2]); v[3] = fmaf(a.beta, o.w, v[3]);
              }
              *c4 = bt_f4{v[0], v[1], v[2], v[3]};
            } else {
              bt_f2* c2 = reinterpret_cast<bt_f2*>(cp);
              if (a.beta != 0.f) {
                const bt_f2 o = *c2;
                v[0] = fmaf(a.beta, o.x, v[0]); v[1] = fmaf(a.beta, o.y, v[1]);
              }
              *c2 = bt_f2{v[0], v[1]};
            }
          } else {
#pragma unroll
            for (int j = 0; j < NTJ; ++j) {
              if (wc + j < a.N) {
                float v = a.alpha * acc[q][j][rg];
                if (a.beta != 0.f) v = fmaf(a.beta, cp[j], v);
                cp[j] = v;
              }
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled variant for products whose M and N are both large: a workgroup of 4 waves (2 x 2) owns a
// (64*WMT) x (64*WNT) block of C[b]; KC rows of X and Y at a time are staged through LDS (double
// buffered, zero-filled past the matrix edges, so nothing is read out of bounds and a NaN in a
// neighbouring matrix cannot leak in), each wave runs WMT x WNT MFMA tiles per k-step from it.
// Against one-wave strips this cuts the L2 traffic per output ~3x, which is what bounded them.
// A second K-segment (X2, Y2, K2) is accumulated into the same tile.
// ------------------------------------------------------------------------------------------------
// VEC: operands whose leading dimensions, sizes and base addresses are multiples of 4 floats are staged
// with 16-byte loads and LDS writes (a quarter of the staging instructions).
template <int WMT, int WNT, int KC, bool VEC = false>
__global__ __launch_bounds__(GMPC_THREADS, 2) void k_bgemm_tn_lds(BgemmArgs a) {
  constexpr int BM = 64 * WMT, BN = 64 * WNT;
  constexpr int VW = VEC ? 4 : 1;
  constexpr int LX = KC * BM / GMPC_THREADS / VW, LY = KC * BN / GMPC_THREADS / VW;
  static_assert(!VEC || ((KC * BM) % (4 * GMPC_THREADS) == 0 && (KC * BN) % (4 * GMPC_THREADS) == 0), "vec staging");
  typedef typename std::conditional<VEC, float4, float>::type stage_t;
  __shared__ __attribute__((aligned(16))) float Xs[2][KC][BM];
  __shared__ __attribute__((aligned(16))) float Ys[2][KC][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int mb = (a.M + BM - 1) / BM, nb = (a.N + BN - 1) / BN;
  // upper-only outputs: the grid holds the LIVE blocks only (row block mi keeps column blocks ni >= mi BM / BN: the
  // ones with an element on or above the diagonal), `a.upper_only` = their number per batch element.  (With the dead
  // blocks in the grid as workgroups that return at once, a 128 x 256 tiling of T1 took the time of the full product.)
  const int lb = a.upper_only ? a.upper_only : mb * nb;
  const long total = (long)a.batch * lb;
  // consecutive workgroup ids go round the 8 XCDs: give every XCD one contiguous range of blocks,
  // so the blocks sharing a batch element's X / Y panels meet in the same L2
  const long per = (total + 7) / 8;
  const long item = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((long)(blockIdx.x >> 3) >= per || item >= total) return;
  const int b = (int)(item / lb);
  int rem = (int)(item - (long)b * lb);
  int mi, ni;
  if (a.upper_only) {
    mi = 0;
    int live = nb;                                   // live blocks of row block mi
    while (rem >= live) { rem -= live; ++mi; live = nb - (mi * BM) / BN; }
    ni = (mi * BM) / BN + rem;
  } else {
    mi = rem / nb;
    ni = rem - mi * nb;
  }
  if (a.active != nullptr && a.active[b] == 0) return;
  const int m0 = mi * BM, n0 = ni * BN;
  // ... and in a block on the diagonal the wave whose 32 WMT x 32 WNT corner lies below it (one of four in a
  // square block) only helps with the staging: no MFMAs, no stores
  const bool dead = a.upper_only && m0 + wm * WMT * 32 > n0 + (wn * WNT + WNT) * 32 - 1;
  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) acc[i][j][rg] = 0.f;
  stage_t rxx[2][LX], ryy[2][LY];   // chunk c travels in set c & 1: two chunks of loads in flight
  const int c1 = (a.K + KC - 1) / KC, c2 = (a.K2 + KC - 1) / KC, c3 = (a.K3 + KC - 1) / KC, nc = c1 + c2 + c3;
  auto zero = []() { stage_t z; memset(&z, 0, sizeof(z)); return z; };
  // Full chunks (all KC rows inside K) are staged through buffer resources: the chunk's row offset is an
  // SGPR, each thread's element offset a loop-invariant VGPR, and a column past the edge is an offset
  // past the resource (the load returns 0) -- no address arithmetic or predicates between the MFMAs
  // (they were ~50 VALU instructions per 32 MFMAs; with two waves per SIMD each costs matrix-pipe time).
  constexpr unsigned OOB = 0x7ff00000u;     // beyond any operand of one batch element (< 2 GB each)
  const __amdgpu_buffer_rsrc_t rX1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.X + (size_t)b * a.sx), 0, (int)(((size_t)(a.K - 1) * a.ldx + a.M) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rY1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Y + (size_t)b * a.sy), 0, (int)(((size_t)(a.K - 1) * a.ldy + a.N) * 4), 0x00020000);
  const bool seg2 = a.K2 > 0;
  const __amdgpu_buffer_rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(seg2 ? a.X2 + (size_t)b * a.sx2 : a.X), 0,
      seg2 ? (int)(((size_t)(a.K2 - 1) * a.ldx2 + a.M) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(seg2 ? a.Y2 + (size_t)b * a.sy2 : a.Y), 0,
      seg2 ? (int)(((size_t)(a.K2 - 1) * a.ldy2 + a.N) * 4) : 0, 0x00020000);
  const bool seg3 = a.K3 > 0;
  const __amdgpu_buffer_rsrc_t rX3 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(seg3 ? a.X3 + (size_t)b * a.sx3 : a.X), 0,
      seg3 ? (int)(((size_t)(a.K3 - 1) * a.ldx3 + a.M) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY3 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(seg3 ? a.Y3 + (size_t)b * a.sy3 : a.Y), 0,
      seg3 ? (int)(((size_t)(a.K3 - 1) * a.ldy3 + a.N) * 4) : 0, 0x00020000);
  unsigned vx1[LX], vx2[LX], vx3[LX], vy1[LY], vy2[LY], vy3[LY];
#pragma unroll
  for (int j = 0; j < LX; ++j) {
    const int e = (tid + GMPC_THREADS * j) * VW, r = e / BM, c = e % BM;
    const bool ok = m0 + c < a.M;
    vx1[j] = ok ? (unsigned)(((size_t)r * a.ldx + m0 + c) * 4) : OOB;
    vx2[j] = ok ? (unsigned)(((size_t)r * a.ldx2 + m0 + c) * 4) : OOB;
    vx3[j] = ok ? (unsigned)(((size_t)r * a.ldx3 + m0 + c) * 4) : OOB;
  }
#pragma unroll
  for (int j = 0; j < LY; ++j) {
    const int e = (tid + GMPC_THREADS * j) * VW, r = e / BN, c = e % BN;
    const bool ok = n0 + c < a.N;
    vy1[j] = ok ? (unsigned)(((size_t)r * a.ldy + n0 + c) * 4) : OOB;
    vy2[j] = ok ? (unsigned)(((size_t)r * a.ldy2 + n0 + c) * 4) : OOB;
    vy3[j] = ok ? (unsigned)(((size_t)r * a.ldy3 + n0 + c) * 4) : OOB;
  }
  auto bload = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff) -> stage_t {
    stage_t out;
    if constexpr (VEC) {
      typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
      const v4u_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
      memcpy(&out, &q, sizeof(out));
    } else {
      const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
      memcpy(&out, &q, sizeof(out));
    }
    return out;
  };
  auto issue = [&](int ci, stage_t (&rx)[LX], stage_t (&ry)[LY]) {
    const int sg = ci < c1 ? 0 : ci < c1 + c2 ? 1 : 2;       // the K-segment of this chunk
    const int K = sg == 0 ? a.K : sg == 1 ? a.K2 : a.K3;
    const int k0 = (sg == 0 ? ci : sg == 1 ? ci - c1 : ci - c1 - c2) * KC;
    const int ldx = sg == 0 ? a.ldx : sg == 1 ? a.ldx2 : a.ldx3, ldy = sg == 0 ? a.ldy : sg == 1 ? a.ldy2 : a.ldy3;
    // (a chunk that runs past row K - 1 needs no other path: those rows lie beyond the buffer resources, whose
    // loads return 0)
    if ((size_t)(K + KC) * (ldx > ldy ? ldx : ldy) * 4 < OOB) {
      // A full chunk lies inside the resource whatever the range check looks at, so its row offset may ride in
      // the scalar offset (no VALU address arithmetic between the MFMAs).  The chunk that runs past row K - 1
      // RELIES on the range check: there the whole byte offset goes into the per-lane part, which the check is
      // documented to cover -- the scalar offset is not (it is covered on gfx950, which is how the first version
      // of this path passed its tests; nothing here depends on that any more).
      const bool tail = k0 + KC > K;         // wave-uniform
      const unsigned ox = (unsigned)k0 * (unsigned)ldx * 4u, oy = (unsigned)k0 * (unsigned)ldy * 4u;
      const unsigned sx_ = tail ? 0u : ox, sy_ = tail ? 0u : oy, ax = tail ? ox : 0u, ay = tail ? oy : 0u;
      if (sg == 0) {
#pragma unroll
        for (int j = 0; j < LX; ++j) rx[j] = bload(rX1, vx1[j] + ax, sx_);
#pragma unroll
        for (int j = 0; j < LY; ++j) ry[j] = bload(rY1, vy1[j] + ay, sy_);
      } else if (sg == 1) {
#pragma unroll
        for (int j = 0; j < LX; ++j) rx[j] = bload(rX2, vx2[j] + ax, sx_);
#pragma unroll
        for (int j = 0; j < LY; ++j) ry[j] = bload(rY2, vy2[j] + ay, sy_);
      } else {
#pragma unroll
        for (int j = 0; j < LX; ++j) rx[j] = bload(rX3, vx3[j] + ax, sx_);
#pragma unroll
        for (int j = 0; j < LY; ++j) ry[j] = bload(rY3, vy3[j] + ay, sy_);
      }
      return;
    }
    const float* X = sg == 0 ? a.X + (size_t)b * a.sx : sg == 1 ? a.X2 + (size_t)b * a.sx2 : a.X3 + (size_t)b * a.sx3;
    const float* Y = sg == 0 ? a.Y + (size_t)b * a.sy : sg == 1 ? a.Y2 + (size_t)b * a.sy2 : a.Y3 + (size_t)b * a.sy3;
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const int e = (tid + GMPC_THREADS * j) * VW, r = e / BM, c = e % BM;
      const bool ok = (k0 + r < K) && (m0 + c < a.M);     // VEC: M % 4 == 0, so a group is in or out
      rx[j] = ok ? *reinterpret_cast<const stage_t*>(X + (size_t)(k0 + r) * ldx + m0 + c) : zero();
    }
#pragma unroll
    for (int j = 0; j < LY; ++j) {
      const int e = (tid + GMPC_THREADS * j) * VW, r = e / BN, c = e % BN;
      const bool ok = (k0 + r < K) && (n0 + c < a.N);
      ry[j] = ok ? *reinterpret_cast<const stage_t*>(Y + (size_t)(k0 + r) * ldy + n0 + c) : zero();
    }
  };
  auto stage = [&](int buf, const stage_t (&rx)[LX], const stage_t (&ry)[LY]) {
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const int e = (tid + GMPC_THREADS * j) * VW;
      *reinterpret_cast<stage_t*>(&Xs[buf][e / BM][e % BM]) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < LY; ++j) {
      const int e = (tid + GMPC_THREADS * j) * VW;
      *reinterpret_cast<stage_t*>(&Ys[buf][e / BN][e % BN]) = ry[j];
    }
  };
  // the loads of chunk c + 2 are issued while chunk c multiplies and chunk c + 1 waits in its registers for
  // the LDS buffer (with one chunk in flight the stage at the end of a chunk waited for loads issued 1.5 k
  // matrix cycles earlier: P and [A | B] come from HBM at the large shapes)
  issue(0, rxx[0], ryy[0]);
  stage(0, rxx[0], ryy[0]);
  if (nc > 1) issue(1, rxx[1], ryy[1]);
  __syncthreads();
  // (the dead wave's chunk is a separate copy: a branch around the MFMAs inside the live one would split the basic
  // block in which the compiler interleaves them with the loads and the LDS writes -- PA at the C4 shard 0.534 ->
  // 0.564 ms)
  auto chunk = [&](int ci, auto par, auto deadc) __attribute__((always_inline)) {
    constexpr int p = decltype(par)::value;            // ci & 1
    constexpr bool DEAD = decltype(deadc)::value;
    const int buf = p;
    if (ci + 2 < nc) issue(ci + 2, rxx[p], ryy[p]);
    if constexpr (!DEAD)
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      float av[WMT], bv[WNT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) av[i] = Xs[buf][kk + half][(wm * WMT + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < WNT; ++j) bv[j] = Ys[buf][kk + half][(wn * WNT + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (ci + 1 < nc) stage(buf ^ 1, rxx[p ^ 1], ryy[p ^ 1]);
    __syncthreads();
  };
  if (dead) {
    for (int ci = 0; ci < nc; ci += 2) {
      chunk(ci, std::integral_constant<int, 0>{}, std::true_type{});
      if (ci + 1 < nc) chunk(ci + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    return;
  }
  for (int ci = 0; ci < nc; ci += 2) {
    chunk(ci, std::integral_constant<int, 0>{}, std::false_type{});
    if (ci + 1 < nc) chunk(ci + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }
  float* C = a.C + (size_t)b * a.sc;
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + (wn * WNT + j) * 32 + l31;
      if (col < a.N) {
        // the addend of the whole tile is requested before the first store (E may alias C as far as the
        // compiler knows: interleaved, every load would wait behind the stores before it).  (All tiles of the
        // wave at once -- one memory round trip per block instead of four -- measured slower: C5 2.059 vs 2.025 s.)
        float ev[16];
        const bool has_e = a.E != nullptr && col < a.En;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = m0 + (wm * WMT + i) * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
          ev[rg] = (has_e && row < a.M) ? a.E[(size_t)b * a.se + (size_t)row * a.lde + col] : 0.f;
        }
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int row = m0 + (wm * WMT + i) * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
          if (row < a.M) {
            float* cp = C + (size_t)row * a.ldc + col;
            float v = a.alpha * acc[i][j][rg];
            if (a.beta != 0.f) v = fmaf(a.beta, *cp, v);
            v += ev[rg];
            if (a.rowmask != nullptr && !((a.rowmask[(size_t)b * a.srm + (row >> 5)] >> (row & 31)) & 1u)) v = 0.f;
            *cp = v;
          }
        }
      }
    }
}

// rows of X and Y per LDS stage: 16 with 16-byte staging (half the barriers per MFMA; C5 5.95 -> 5.85 s),
// 8 with dword staging (16 there doubles the staging instructions: C4 115 -> 120 ms)
#ifndef GMPC_BG_KC_VEC
#define GMPC_BG_KC_VEC 16
#endif
#ifndef GMPC_BG_KC_VEC22            // stage depth of the 128 x 128 blocks with 16-byte staging (C5: 2.025 s with 8, 2.044 with 16, 2.161 with 32)
#define GMPC_BG_KC_VEC22 8
#endif
#ifndef GMPC_BG_KC
#define GMPC_BG_KC 8
#endif
template <int WMT, int WNT>
static void launch_lds(const BgemmArgs& a0, hipStream_t s) {
  constexpr int BM = 64 * WMT, BN = 64 * WNT;
  BgemmArgs a = a0;
  const int mbk = (a.M + BM - 1) / BM, nbk = (a.N + BN - 1) / BN;
  long lb = (long)mbk * nbk;
  if (a.upper_only) {                 // live blocks per batch element (see the kernel)
    lb = 0;
    for (int mi = 0; mi < mbk; ++mi) lb += nbk - (mi * BM) / BN > 0 ? nbk - (mi * BM) / BN : 0;
    a.upper_only = (int)lb;
  }
  const long total = (long)a.batch * lb;
  const long per = (total + 7) / 8;
  // 16-byte staging needs columns in groups of four (M, N multiples of 4); the ROWS need not start on 16 bytes:
  // buffer_load_dwordx4 takes any 4-byte-aligned address (ld = n + m = 393 at C4) -- GMPC_BG_VEC_ALIGNED=1 asks for
  // 16-byte rows as the first version did
  static const bool want_al = [] { const char* e = getenv("GMPC_BG_VEC_ALIGNED"); return e != nullptr && e[0] == '1'; }();
  auto al4 = [](const void* p_, long st, int ld) {
    return p_ == nullptr || !want_al || (((uintptr_t)p_ & 15) == 0 && (st & 3) == 0 && (ld & 3) == 0);
  };
  // (192-wide blocks stage 16 bytes per thread with 16-row stages only: 8 rows x 192 columns are 1.5 loads per thread)
  const bool vec = (a.M & 3) == 0 && (a.N & 3) == 0 && al4(a.X, a.sx, a.ldx) &&
                   al4(a.Y, a.sy, a.ldy) && al4(a.K2 ? a.X2 : nullptr, a.sx2, a.ldx2) &&
                   al4(a.K2 ? a.Y2 : nullptr, a.sy2, a.ldy2) && al4(a.K3 ? a.X3 : nullptr, a.sx3, a.ldx3) &&
                   al4(a.K3 ? a.Y3 : nullptr, a.sy3, a.ldy3);
  if (vec)
    hipLaunchKernelGGL((k_bgemm_tn_lds<WMT, WNT, WNT == 2 ? GMPC_BG_KC_VEC22 : GMPC_BG_KC_VEC, true>), dim3((unsigned)(per * 8)),
                       dim3(GMPC_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL((k_bgemm_tn_lds<WMT, WNT, GMPC_BG_KC>), dim3((unsigned)(per * 8)), dim3(GMPC_THREADS), 0,
                       s, a);
}

void gmpc_launch_bgemm_tn(const BgemmArgs& a, hipStream_t s) {
  // a thin product whose wide operand is worth streaming (k_bthin)
  {
    static const bool off = [] { const char* e = getenv("GMPC_BTHIN"); return e != nullptr && e[0] == '0'; }();
    const bool widex = a.N <= 64 && a.M >= 128, widey = a.M <= 64 && a.N >= 128;
    if (!off && (widex || widey) && a.K >= 2 * BT_RD && a.E == nullptr && a.rowmask == nullptr && a.K2 == 0 &&
        a.K3 == 0 && !a.upper_only) {
      const int Wd = widex ? a.M : a.N, Th = widex ? a.N : a.M;
      // 128 columns per wave (16-byte loads) when that still gives every SIMD a few waves, else 64 (8-byte loads,
      // twice the waves: PB at the C4 shard is 1536 waves of 128 columns -- 1.5 per SIMD, 0.084 ms -- or 3072 of
      // 64, 0.073 ms); a thin operand of 33..64 columns goes through one wave as two strips
      static const int ntj_env = [] { const char* e = getenv("GMPC_BTHIN_NTJ"); return e ? atoi(e) : 0; }();
      static const int ns_env = [] { const char* e = getenv("GMPC_BTHIN_NS"); return e ? atoi(e) : 0; }();
      const int ns = ns_env == 1 ? 1 : Th > 32 ? 2 : 1;
      const long waves4 = (long)a.batch * ((Wd + 127) / 128) * ((Th + 32 * ns - 1) / (32 * ns));
      // (two strips x four tiles are 268 registers, one wave per SIMD: 0.387 ms against 0.360 with two tiles for
      // the [64 x 1088] x K = 200 products of C5)
      const int ntj = ntj_env == 2 || ntj_env == 4 ? ntj_env : (waves4 < 4096 || ns == 2) ? 2 : 4;
      const long total = (long)a.batch * ((Wd + 32 * ntj - 1) / (32 * ntj)) * ((Th + 32 * ns - 1) / (32 * ns));
      const dim3 grid((unsigned)((total + 3) / 4)), blk(GMPC_THREADS);
#define BT_LAUNCH(WX, NJ, S) hipLaunchKernelGGL((k_bthin<WX, NJ, BT_RD, S>), grid, blk, 0, s, a)
#define BT_LAUNCH2(WX, NJ) do { if (ns == 2) BT_LAUNCH(WX, NJ, 2); else BT_LAUNCH(WX, NJ, 1); } while (0)
      if (widex) { if (ntj == 2) BT_LAUNCH2(true, 2); else BT_LAUNCH2(true, 4); }
      else       { if (ntj == 2) BT_LAUNCH2(false, 2); else BT_LAUNCH2(false, 4); }
#undef BT_LAUNCH2
#undef BT_LAUNCH
      return;
    }
  }
  // (the epilogue extras and the second K-segment exist in the LDS-staged kernel only)
  if ((a.M > 32 && a.N > 64) || a.E != nullptr || a.rowmask != nullptr || a.K2 > 0 || a.K3 > 0) {
    // column blocks of 128 / 192 / 256: the one that pads N least (ties: the widest)
    // (upper-only outputs: the area of the blocks that are not skipped -- narrow blocks follow the diagonal)
    int best = 2;
    long waste = -1;
    for (int w = 2; w <= 4; ++w) {
      const int bn = 64 * w, nbk = (a.N + bn - 1) / bn;
      long padded = (long)nbk * bn;
      if (a.upper_only) {
        padded = 0;
        for (int mi = 0; mi * 128 < a.M; ++mi)
          for (int ni = 0; ni < nbk; ++ni)
            if (!(mi * 128 > ni * bn + bn - 1)) padded += bn;
      }
      if (waste < 0 || padded <= waste) { waste = padded; best = w; }
    }
    static const int w_env = [] { const char* e = getenv("GMPC_BG_UPPER_W"); return e ? atoi(e) : 0; }();
    if (a.upper_only && w_env >= 2 && w_env <= 4) best = w_env;      // (A/B timing of the block width)
    switch (best) {
      case 2: launch_lds<2, 2>(a, s); break;
      case 3: launch_lds<2, 3>(a, s); break;
      default: launch_lds<2, 4>(a, s); break;
    }
    return;
  }
  // a thin product: one wave per strip (the second K-segment is not supported here)
  const int tiles = (a.N + 31) / 32;
  const int ntw = tiles >= 8 && tiles % 8 == 0 ? 8 : tiles >= 6 && tiles % 6 == 0 ? 6
                  : tiles >= 4 ? 4 : tiles >= 2 ? 2 : 1;
  const int mstrips = (a.M + 31) / 32, ngroups = (a.N + 32 * ntw - 1) / (32 * ntw);
  const long total = (long)a.batch * mstrips * ngroups;
  const dim3 grid((unsigned)((total + 3) / 4)), blk(GMPC_THREADS);
  switch (ntw) {
    case 8: hipLaunchKernelGGL(k_bgemm_tn<8>, grid, blk, 0, s, a); break;
    case 6: hipLaunchKernelGGL(k_bgemm_tn<6>, grid, blk, 0, s, a); break;
    case 4: hipLaunchKernelGGL(k_bgemm_tn<4>, grid, blk, 0, s, a); break;
    case 2: hipLaunchKernelGGL(k_bgemm_tn<2>, grid, blk, 0, s, a); break;
    default: hipLaunchKernelGGL(k_bgemm_tn<1>, grid, blk, 0, s, a); break;
  }
}

// ------------------------------------------------------------------------------------------------
// per-trajectory pieces of one backward step (everything that is not an n^3 product)
// ------------------------------------------------------------------------------------------------
struct BigStepArgs {
  int B, n, m, T, t;
  int mode;              // 0: iLQR step (trajax lqr_step, Cholesky of G + 1e-8 I)
                         // 1: bilevel Hessian solve (oracle hessian_solve: no regulariser, LU with
                         //    partial pivoting, linear term -Bvec_t, loss adjoint mu in `lam`)
  const float* lx;       // mode 1: [B][T+1][n] d loss / d X
  float* Bvec;           // mode 1: [B][T][m]   out: B_t^T mu_{t+1}
  const float* X; const float* U; const float* goal; const float* mpc_w;
  int ng;                // columns of `goal` (0: n)
  const float* ABt;      // [B][n][n+m]   Jacobians of step t (dense form)
  const float* Vt;       // low-rank form (non-null): [B][h][n+m] with A = I + WL^T Vx^T, B = WL^T Vu^T
  const float* WL;       //   the output layer's kernel [h][n] (shared)
  int h;
  const float* HG;       // [B][m][n+m]   [B^T P A | B^T P B]
  float* KV;             // [B][2m][n]    out: rows 0..m-1 = K_t, rows m..2m-1 = V = H + G K / 2
  float* VK;             // [B][2m][n]    out: rows 0..m-1 = V,   rows m..2m-1 = K_t
  float* pvec; float* lam;   // [B][n]    value vector / adjoint, updated in place
  float* sbuf;           // [B]           out: sqrt(|x-g|^2 + alpha^2) of this step (for Q_t)
  float* gn2;            // [B]           running sum of squared control gradients
  const int* active;
  float* K; float* k; float* grad; float* adj;   // [B][T][m][n], [B][T][m], [B][T][m], [B][T+1][n]
  int solve_valu;        // mode 0: the gain solve on the vector pipe (the form before big_solve_mfma; GMPC_BIG_SOLVE=valu)
};

// 4-row blocks of the matrix-pipe gain solve (big_solve_mfma) for m controls: the instantiated size that holds m
static int big_solve_blocks(int m) { return m <= 8 ? 2 : m <= 20 ? 5 : m <= 32 ? 8 : 16; }
static size_t big_step_lds(int n, int m, int h) {
  const size_t MP = (size_t)((m + 7) & ~7);     // solve columns and the blocked solve's copies are padded to 8
  // the solve's work area: one column per thread + the padded copies of the vector form, or the operand
  // fragments + the padded factor of the matrix-pipe form
  const size_t MB = big_solve_blocks(m), KCH = (4 * MB + 15) / 16;
  const size_t valu = MP * GMPC_THREADS + ((m & 7) ? 3 : 1) * MP * MP + 4, mfma = 3 * MB * KCH * 64 + 16 * MB * MB + 4;
  return ((size_t)2 * m * m + 5 * (size_t)n + 7 * (size_t)m + 16 + 2 * (size_t)h + 2 * GMPC_THREADS +
          (valu > mfma ? valu : mfma)) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// [K_t] = -(G + delta I)^-1 H, V = H + G K / 2 of k_big_step (mode 0) with the matrix pipe doing the
// multiply-subtracts.  One column of H per lane as before; the column lives in REGISTERS (y[4 MB]) and is the B
// operand of v_mfma_f32_4x4x1_16B_f32: d[i] += A[i][k] * y[k] for the 4 rows of a block and the lane's own column,
// A = 16 consecutive k of (-L), (-L^T) or G for the block's 4 rows in one VGPR ([k][4 rows] fragments built once
// per trajectory in LDS, broadcast with cbsz / abid as in the trajectory kernels).  What stays on the vector
// pipe is the 4 x 4 triangle on the diagonal of every block (6 multiply-subtracts and 4 divisions per block and
// sweep).  Same operations as the vector form (exact fp32 FMAs, divisions by the diagonal); the multiply-subtracts
// of a row are summed in two interleaved chains (even / odd k) instead of one.
// At the C5 shard (m = 64, n = 1024) the vector form spent 2.3 of k_big_step's 2.8 ms here (0.4 LDS reads per
// multiply-subtract); this form: 2 300 MFMAs of 8 cycles per 64 columns.
// ------------------------------------------------------------------------------------------------
template <int MB>
__device__ __forceinline__ void big_solve_mfma(int n, int m, int nm, const float* L, const float* G,
                                               const float* __restrict__ HG, float* work, float* __restrict__ Kt,
                                               float* __restrict__ KV, float* __restrict__ VK) {
  constexpr int MP = 4 * MB, KCH = (MP + 15) / 16;
  float* const AsF = work;                       // [MB][KCH][16 k][4 rows]: -L below the block's diagonal block
  float* const AsB = AsF + MB * KCH * 64;        // -L^T right of it
  float* const AsG = AsB + MB * KCH * 64;        // G
  float* const Lp = AsG + MB * KCH * 64;         // [MP][MP] L padded with an identity block
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < MB * KCH * 64; e += GMPC_THREADS) {
    const int l = e & 63, kc = (e >> 6) % KCH, ib = (e >> 6) / KCH;
    const int row = 4 * ib + (l & 3), k = 16 * kc + (l >> 2);
    const bool in = row < m && k < m;
    AsF[e] = (in && k < 4 * ib) ? -L[row * m + k] : 0.f;
    AsB[e] = (in && k >= 4 * ib + 4) ? -L[k * m + row] : 0.f;
    AsG[e] = in ? G[row * m + k] : 0.f;
  }
  for (int e = tid; e < MP * MP; e += GMPC_THREADS) {
    const int i = e / MP, k = e - i * MP;
    Lp[e] = (i < m && k < m) ? L[i * m + k] : (i == k ? 1.f : 0.f);
  }
  __syncthreads();
  const int m_in = m;
  for (int c0 = 0; c0 + 64 * wave < n; c0 += GMPC_THREADS) {          // (uniform per wave: the MFMAs need all lanes)
    // (m made opaque per iteration: the ~200 uniform "row < m" tests below are compared where they are used
    // instead of being hoisted out of the loop into -- and spilled from -- the scalar registers)
    int m = m_in;
    asm volatile("" : "+s"(m));
    const int c = c0 + tid;
    const bool cok = c < n;
    const float* Hc = HG + (cok ? c : n - 1);
    // (rows through walking pointers: 64 loop-invariant row offsets would be hoisted into -- and spilled from --
    // the scalar registers)
    float y[MP];
    {
      const float* hp = Hc;
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        y[i] = i < m ? *hp : 0.f;
        if (i + 1 < m) hp += nm;
      }
    }
    // ---- L y = H: block ib needs y[0 .. 4 ib - 1]
    rw_static_for<MB>([&](auto ibc) __attribute__((always_inline)) {
      constexpr int ib = decltype(ibc)::value;
      f32x4_t d0 = {y[4 * ib], y[4 * ib + 1], y[4 * ib + 2], y[4 * ib + 3]}, d1 = {0.f, 0.f, 0.f, 0.f};
      rw_static_for<(4 * ib + 15) / 16>([&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = decltype(kcc)::value;
        const float ar = AsF[(ib * KCH + kc) * 64 + lane];
        rw_static_for<16>([&](auto kkc) __attribute__((always_inline)) {
          constexpr int k = 16 * kc + decltype(kkc)::value;
          if constexpr (k < 4 * ib) {
            if constexpr (k & 1) rw_mfma<k>(d1, ar, y[k]);
            else rw_mfma<k>(d0, ar, y[k]);
          }
        });
      });
      const float* Ld = Lp + (4 * ib) * MP + 4 * ib;
      float v0 = d0[0] + d1[0], v1 = d0[1] + d1[1], v2 = d0[2] + d1[2], v3 = d0[3] + d1[3];
      v0 = v0 / Ld[0];
      v1 = (v1 - Ld[MP] * v0) / Ld[MP + 1];
      v2 = ((v2 - Ld[2 * MP] * v0) - Ld[2 * MP + 1] * v1) / Ld[2 * MP + 2];
      v3 = (((v3 - Ld[3 * MP] * v0) - Ld[3 * MP + 1] * v1) - Ld[3 * MP + 2] * v2) / Ld[3 * MP + 3];
      y[4 * ib] = v0; y[4 * ib + 1] = v1; y[4 * ib + 2] = v2; y[4 * ib + 3] = v3;
    });
    // ---- L^T x = y: block ib needs x[4 ib + 4 ..]
    rw_static_for<MB>([&](auto ibr) __attribute__((always_inline)) {
      constexpr int ib = MB - 1 - decltype(ibr)::value;
      f32x4_t d0 = {y[4 * ib], y[4 * ib + 1], y[4 * ib + 2], y[4 * ib + 3]}, d1 = {0.f, 0.f, 0.f, 0.f};
      constexpr int kc0 = (4 * ib + 4) / 16;
      rw_static_for<KCH - kc0>([&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = kc0 + decltype(kcc)::value;
        const float ar = AsB[(ib * KCH + kc) * 64 + lane];
        rw_static_for<16>([&](auto kkc) __attribute__((always_inline)) {
          constexpr int k = 16 * kc + decltype(kkc)::value;
          if constexpr (k >= 4 * ib + 4 && k < MP) {
            if constexpr (k & 1) rw_mfma<k>(d1, ar, y[k]);
            else rw_mfma<k>(d0, ar, y[k]);
          }
        });
      });
      const float* Ld = Lp + (4 * ib) * MP + 4 * ib;      // U[r][q] = L[q][r]
      float v0 = d0[0] + d1[0], v1 = d0[1] + d1[1], v2 = d0[2] + d1[2], v3 = d0[3] + d1[3];
      v3 = v3 / Ld[3 * MP + 3];
      v2 = (v2 - Ld[3 * MP + 2] * v3) / Ld[2 * MP + 2];
      v1 = ((v1 - Ld[2 * MP + 1] * v2) - Ld[3 * MP + 1] * v3) / Ld[MP + 1];
      v0 = (((v0 - Ld[MP] * v1) - Ld[2 * MP] * v2) - Ld[3 * MP] * v3) / Ld[0];
      y[4 * ib] = v0; y[4 * ib + 1] = v1; y[4 * ib + 2] = v2; y[4 * ib + 3] = v3;
    });
#pragma unroll
    for (int i = 0; i < MP; ++i) y[i] = -y[i];            // K's column
    // ---- V = H + G K / 2, outputs
    const float* hp = Hc;
    const size_t co = cok ? c : 0, mn = (size_t)m * n;
    float* kp = Kt + co;
    float* kvp = KV + co;
    float* vkp = VK + co;
    rw_static_for<MB>([&](auto ibc) __attribute__((always_inline)) {
      constexpr int ib = decltype(ibc)::value;
      f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      float hr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hr[r] = 4 * ib + r < m ? *hp : 0.f;
        if (4 * ib + r + 1 < m) hp += nm;
      }
      rw_static_for<KCH>([&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = decltype(kcc)::value;
        const float ar = AsG[(ib * KCH + kc) * 64 + lane];
        rw_static_for<16>([&](auto kkc) __attribute__((always_inline)) {
          constexpr int k = 16 * kc + decltype(kkc)::value;
          if constexpr (k < MP) {
            if constexpr (k & 1) rw_mfma<k>(d1, ar, y[k]);
            else rw_mfma<k>(d0, ar, y[k]);
          }
        });
      });
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * ib + r;
        if (i < m && cok) {
          const float kic = y[i];
          const float vic = fmaf(0.5f, d0[r] + d1[r], hr[r]);
          *kp = kic;
          *kvp = kic;
          kvp[mn] = vic;
          *vkp = vic;
          vkp[mn] = kic;
        }
        kp += n; kvp += n; vkp += n;
      }
    });
  }
}

#ifdef GMPC_BIGSTEP_STAMPS
#define BS_STAMP(i) { __syncthreads(); if (threadIdx.x == 0) bs_t[i] = __builtin_readcyclecounter(); }
#else
#define BS_STAMP(i)
#endif
__global__ __launch_bounds__(GMPC_THREADS) void k_big_step(BigStepArgs a) {
#ifdef GMPC_BIGSTEP_STAMPS
  unsigned long long bs_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  BS_STAMP(0)
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = a.n, m = a.m, T = a.T, t = a.t, nm = n + m;
  const int tid = threadIdx.x, b = blockIdx.x;
  if (a.active != nullptr && a.active[b] == 0) return;
  float* G = reinterpret_cast<float*>(smem);     // m x m  (symmetrised R + B^T P B)
  float* L = G + m * m;                          // m x m  Cholesky factor / work copy
  float* pv = L + m * m;                         // n
  float* lv = pv + n;                            // n
  float* dv = lv + n;                            // n   x - goal
  float* qv = dv + n;                            // n
  float* pa = qv + n;                            // n   A^T p
  float* uv = pa + n;                            // m
  float* rv = uv + m;                            // m
  float* hv = rv + m;                            // m
  float* kv = hv + m;                            // m
  float* gk = kv + m;                            // m   G k + h
  float* gsq = gk + m;                           // m
  int* perm = reinterpret_cast<int*>(gsq + m);   // m   row permutation of the LU (mode 1)
  float* red = gsq + 2 * m;                      // 16
  float* yl = red + 16;                          // h   W_L lam   (low-rank form)
  float* yp = yl + a.h;                          // h   W_L p
  float* part = yp + a.h;                        // 2 x 256 partial sums
  float* ycol = part + 2 * GMPC_THREADS;         // m x 256: one solve column per thread
  const float* AB = a.ABt + (size_t)b * n * nm;
  const float* HG = a.HG + (size_t)b * m * nm;
  const size_t bt = (size_t)b * T + t;
  const float w0 = sigmoidf_(a.mpc_w[0]), w1 = sigmoidf_(a.mpc_w[1]);
  const float al = GMPC_ALPHA;
  float dd = 0.f, uu = 0.f;
  const int ng = a.ng > 0 ? a.ng : n;      // the staging cost sees xc[:ng]
  for (int i = tid; i < n; i += blockDim.x) {
    const size_t xi = ((size_t)b * (T + 1) + t) * n + i;
    const float d = i < ng ? a.X[xi] - a.goal[((size_t)b * (T + 1) + t) * ng + i] : 0.f;
    dv[i] = d;
    dd = fmaf(d, d, dd);
    pv[i] = a.pvec[(size_t)b * n + i];
    lv[i] = a.lam[(size_t)b * n + i];
  }
  for (int j = tid; j < m; j += blockDim.x) {
    const float u = a.U[bt * m + j];
    uv[j] = u;
    uu = fmaf(u, u, uu);
  }
  dd = wave_sum(dd); uu = wave_sum(uu);
  if ((tid & 63) == 0) { red[tid >> 6] = dd; red[4 + (tid >> 6)] = uu; }
  __syncthreads();
  dd = (red[0] + red[1]) + (red[2] + red[3]);
  uu = (red[4] + red[5]) + (red[6] + red[7]);
  const float s = sqrtf(dd + al * al), su = sqrtf(uu + al * al);
  const float isu = 1.f / su, isu3 = 1.f / (su * su * su);
  if (tid == 0) a.sbuf[b] = s;
  // linear terms of the two vector recursions: mode 0 the cost gradient (q_t, r_t) for both the
  // adjoint lambda and the value vector p; mode 1 (d loss/d x_t, 0) for the loss adjoint and (0, -Bvec_t)
  // for p
  const bool m1 = a.mode == 1;
  for (int i = tid; i < n; i += blockDim.x)
    qv[i] = m1 ? a.lx[((size_t)b * (T + 1) + t) * n + i] : w1 * dv[i] / s;
  for (int j = tid; j < m; j += blockDim.x) rv[j] = m1 ? 0.f : w0 * uv[j] / su;
  __syncthreads();
  BS_STAMP(1)
  const bool lowrank = a.Vt != nullptr;
  const float* Vt = lowrank ? a.Vt + (size_t)b * a.h * nm : nullptr;
  if (lowrank) {
    // y = W_L v for v = lam, p: one wave per row of W_L (coalesced along the row), four rows and four 64-wide
    // slices at a time so that 16 loads are in flight (one load per iteration left every one of the 16 k
    // loads of a wave's rows exposed: 0.68 M of the kernel's 2.8 M cycles at n = 1024, h = 200)
    const int wave = tid >> 6, ln = tid & 63;
    constexpr int NW = GMPC_THREADS / 64, RW = 4;       // rows per wave and pass: 16 loads in flight
    for (int k = wave; k < a.h; k += RW * NW) {
      const float* wr[RW];
      bool ok[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        ok[r] = k + r * NW < a.h;
        wr[r] = a.WL + (size_t)(ok[r] ? k + r * NW : k) * n;
      }
      float sl[RW], sp[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) { sl[r] = 0.f; sp[r] = 0.f; }
      for (int i0 = ln; i0 < n; i0 += 256) {
        float wv[RW][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(i0 + 64 * q, n - 1);
#pragma unroll
          for (int r = 0; r < RW; ++r) wv[r][q] = wr[r][i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + 64 * q;
          const float lvi = i < n ? lv[i] : 0.f, pvi = i < n ? pv[i] : 0.f;
#pragma unroll
          for (int r = 0; r < RW; ++r) { sl[r] = fmaf(wv[r][q], lvi, sl[r]); sp[r] = fmaf(wv[r][q], pvi, sp[r]); }
        }
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const float a_ = wave_sum(sl[r]), b_ = wave_sum(sp[r]);
        if (ln == 0 && ok[r]) { yl[k + r * NW] = a_; yp[k + r * NW] = b_; }
      }
    }
    __syncthreads();
  }
  BS_STAMP(2)
  // g_t = r + B^T lam ; h = r + B^T p : thread (rp, j) sums rows rp, rp + RP, ... of column j of B
  // (low-rank form: B^T v = Vu (W_L v), rows of V^T instead of rows of B)
  {
    const int MC = m <= 32 ? 32 : 64, RP = GMPC_THREADS / MC;
    const int rp = tid / MC, j = tid - rp * MC;
    float g = 0.f, h = 0.f;
    if (j < m) {
      // (8 loads in flight; the sums keep the order of the one-load loop)
      const float* Mj = (lowrank ? Vt : AB) + n + j;
      const float* vL = lowrank ? yl : lv;
      const float* vP = lowrank ? yp : pv;
      const int rows = lowrank ? a.h : n;
      int i = rp;
      for (; i + 7 * RP < rows; i += 8 * RP) {
        float e[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) e[r] = Mj[(size_t)(i + r * RP) * nm];
#pragma unroll
        for (int r = 0; r < 8; ++r) { g = fmaf(e[r], vL[i + r * RP], g); h = fmaf(e[r], vP[i + r * RP], h); }
      }
      for (; i < rows; i += RP) {
        const float e = Mj[(size_t)i * nm];
        g = fmaf(e, vL[i], g);
        h = fmaf(e, vP[i], h);
      }
    }
    part[tid] = g;
    part[GMPC_THREADS + tid] = h;
    __syncthreads();
    if (tid < m) {
      float gs = 0.f, hs = 0.f;
      for (int r = 0; r < RP; ++r) { gs += part[r * MC + tid]; hs += part[GMPC_THREADS + r * MC + tid]; }
      gs = rv[tid] + gs;
      if (m1) {
        a.Bvec[bt * m + tid] = gs;          // B_t^T mu_{t+1}
        hv[tid] = hs - gs;                  // h = -Bvec_t + B^T p
      } else {
        hv[tid] = rv[tid] + hs;
        a.grad[bt * m + tid] = gs;
      }
      gsq[tid] = gs * gs;
    }
  }
  BS_STAMP(3)
  // lam_t = q + A^T lam ; pa = A^T p: column c of A is read coalesced across threads, NQ columns of a thread and RI
  // rows at a time -- 16 loads in flight.  NQ follows n (four columns per thread at n = 376 made 2.7 loads per useful
  // one: the clamped duplicates of columns past n); the sums run over the rows in the same order whatever RI is.
  {
    const float* M = lowrank ? Vt : AB;
    const float* vL = lowrank ? yl : lv;
    const float* vP = lowrank ? yp : pv;
    const int rows = lowrank ? a.h : n;
    auto matvec = [&](auto nqc, auto ric) __attribute__((always_inline)) {
      constexpr int NQ = decltype(nqc)::value, RI = decltype(ric)::value;
      for (int c0 = tid; c0 < n; c0 += NQ * (int)blockDim.x) {
        float vl[NQ], vp[NQ];
        int cq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { vl[q] = 0.f; vp[q] = 0.f; cq[q] = min(c0 + q * (int)blockDim.x, n - 1); }
        int i = 0;
        for (; i + RI <= rows; i += RI) {
          float e[RI][NQ], lr[RI], pr[RI];
#pragma unroll
          for (int r = 0; r < RI; ++r) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) e[r][q] = M[(size_t)(i + r) * nm + cq[q]];
            lr[r] = vL[i + r];
            pr[r] = vP[i + r];
          }
#pragma unroll
          for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int q = 0; q < NQ; ++q) { vl[q] = fmaf(e[r][q], lr[r], vl[q]); vp[q] = fmaf(e[r][q], pr[r], vp[q]); }
        }
        for (; i < rows; ++i) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const float ev = M[(size_t)i * nm + cq[q]];
            vl[q] = fmaf(ev, vL[i], vl[q]); vp[q] = fmaf(ev, vP[i], vp[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c = c0 + q * (int)blockDim.x;
          if (c >= n) continue;
          if (lowrank) { vl[q] += lv[c]; vp[q] += pv[c]; }      // A^T v = v + Vx (W_L v)
          pa[c] = vp[q];
          const float ln = qv[c] + vl[q];
          a.lam[(size_t)b * n + c] = ln;
          if (!m1) a.adj[((size_t)b * (T + 1) + t) * n + c] = ln;
        }
      }
    };
    const int nq = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    if (nq <= 1) matvec(std::integral_constant<int, 1>{}, std::integral_constant<int, 16>{});
    else if (nq == 2) matvec(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
    else matvec(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
  }
  BS_STAMP(4)
  // G = sym(R + B^T P B)
  for (int e = tid; e < m * m; e += blockDim.x) {
    const int i = e / m, j = e - i * m;
    const float Rij = w0 * ((i == j ? isu : 0.f) - uv[i] * uv[j] * isu3);
    L[e] = Rij + HG[(size_t)i * nm + n + j];
  }
  __syncthreads();
  if (tid == 0 && !m1) {
    float sg = 0.f;
    for (int j = 0; j < m; ++j) sg += gsq[j];
    a.gn2[b] += sg;
  }
  for (int e = tid; e < m * m; e += blockDim.x) {
    const int i = e / m, j = e - i * m;
    G[e] = (L[e] + L[j * m + i]) * 0.5f;
  }
  __syncthreads();
  float* Kt = a.K + bt * m * n;
  float* KV = a.KV + (size_t)b * 2 * m * n;
  float* VK = a.VK + (size_t)b * 2 * m * n;
  float* y = ycol + tid;
  if (!m1) {
    BS_STAMP(5)
    // Cholesky of G + 1e-8 I in L (lower), column by column; NaN on a non-positive pivot
    for (int e = tid; e < m * m; e += blockDim.x) L[e] = G[e] + ((e / m) == (e % m) ? 1e-8f : 0.f);
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      if (tid == 0) L[j * m + j] = sqrtf(L[j * m + j]);
      __syncthreads();
      const float d = L[j * m + j];
      for (int i = j + 1 + tid; i < m; i += blockDim.x) L[i * m + j] /= d;
      __syncthreads();
      // trailing update of the lower triangle: L[i][k] -= L[i][j] L[k][j], j < k <= i; thread (tid / 16, tid % 16)
      // walks rows and columns in steps of 16 (no integer division by the shrinking size in the loop)
      for (int i = j + 1 + (tid >> 4); i < m; i += GMPC_THREADS / 16) {
        const float lij = L[i * m + j];
        for (int k = j + 1 + (tid & 15); k <= i; k += 16) L[i * m + k] -= lij * L[k * m + j];
      }
      __syncthreads();
    }
    BS_STAMP(6)
    // [K k] = -(G + delta I)^-1 [H h]
    if (!a.solve_valu && m <= 64) {
      // the right-hand side h: wave 0 across its lanes (see the vector form below); the n columns of H: one
      // per lane, the multiply-subtracts on the matrix pipe (big_solve_mfma)
      if (tid < 64) {
        float v = tid < m ? hv[tid] : 0.f, yv = 0.f;
        for (int k = 0; k < m; ++k) {
          const float yk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)) / L[k * m + k];
          if (tid == k) yv = yk;
          if (tid > k && tid < m) v -= L[tid * m + k] * yk;
        }
        for (int k = m - 1; k >= 0; --k) {
          const float xk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yv), k)) / L[k * m + k];
          if (tid == k) yv = xk;
          if (tid < k) yv -= L[k * m + tid] * xk;
        }
        if (tid < m) { kv[tid] = -yv; a.k[bt * m + tid] = -yv; }
      }
      if (m <= 8) big_solve_mfma<2>(n, m, nm, L, G, HG, ycol, Kt, KV, VK);
      else if (m <= 20) big_solve_mfma<5>(n, m, nm, L, G, HG, ycol, Kt, KV, VK);
      else if (m <= 32) big_solve_mfma<8>(n, m, nm, L, G, HG, ycol, Kt, KV, VK);
      else big_solve_mfma<16>(n, m, nm, L, G, HG, ycol, Kt, KV, VK);
    } else
    // the vector form: one right-hand-side column per thread (column n is h); the thread keeps its column in
    // LDS (ycol[i][tid])
    {
      // blocks of 8 rows share the loads of the solved part of the column and read their rows of L (L^T in
      // the backward sweep) and G 16 bytes at a time -- 0.4 LDS reads per multiply-subtract instead of 2 (the
      // scalar form spent 9.4 ms per time step of the C5 shard in this loop).  m is padded to a multiple of
      // 8 with an identity block (copies Lb / Ltb / Gb with row stride MP; for m % 8 == 0 L and G are used in
      // place and only the transpose is built)
      const int MP = (m + 7) & ~7;
      float* xtra = reinterpret_cast<float*>(
          (reinterpret_cast<uintptr_t>(ycol + (size_t)MP * GMPC_THREADS) + 15) & ~(uintptr_t)15);
      float* Ltb = xtra;                                    // Ltb[i][k] = L[k][i]
      float* Lb = (m & 7) ? xtra + MP * MP : L;
      float* Gb = (m & 7) ? xtra + 2 * MP * MP : G;
      for (int e = tid; e < MP * MP; e += blockDim.x) {
        const int i = e / MP, k = e - i * MP;
        const bool in = i < m && k < m;
        Ltb[e] = in ? L[k * m + i] : (i == k ? 1.f : 0.f);
        if (m & 7) {
          Lb[e] = in ? L[i * m + k] : (i == k ? 1.f : 0.f);
          Gb[e] = in ? G[i * m + k] : 0.f;
        }
      }
      __syncthreads();
      // the right-hand side h (column n): one more column would be one more sweep of the loop below for a
      // single thread (n = 1024: a fifth sweep as long as the other four) -- wave 0 solves it across its lanes
      // instead, lane i = row i, the pivot's value handed round with v_readlane
      const bool hwave = m <= 64;
      if (hwave && tid < 64) {
        float v = tid < m ? hv[tid] : 0.f, yv = 0.f;
        for (int k = 0; k < m; ++k) {
          const float yk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)) / L[k * m + k];
          if (tid == k) yv = yk;
          if (tid > k && tid < m) v -= L[tid * m + k] * yk;
        }
        for (int k = m - 1; k >= 0; --k) {
          const float xk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yv), k)) / L[k * m + k];
          if (tid == k) yv = xk;
          if (tid < k) yv -= L[k * m + tid] * xk;
        }
        if (tid < m) { kv[tid] = -yv; a.k[bt * m + tid] = -yv; }
      }
      for (int c = tid; c < n + (hwave ? 0 : 1); c += blockDim.x) {
        for (int i0 = 0; i0 < MP; i0 += 8) {
          float acc[8], yb[8];
#pragma unroll
          for (int r = 0; r < 8; ++r)
            acc[r] = i0 + r < m ? (c < n ? HG[(size_t)(i0 + r) * nm + c] : hv[i0 + r]) : 0.f;
          for (int k = 0; k < i0; k += 4) {
            const float y0 = y[(k + 0) * GMPC_THREADS], y1 = y[(k + 1) * GMPC_THREADS];
            const float y2 = y[(k + 2) * GMPC_THREADS], y3 = y[(k + 3) * GMPC_THREADS];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float4 l4 = *reinterpret_cast<const float4*>(&Lb[(i0 + r) * MP + k]);
              acc[r] -= l4.x * y0; acc[r] -= l4.y * y1; acc[r] -= l4.z * y2; acc[r] -= l4.w * y3;
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float v = acc[r];
#pragma unroll
            for (int q = 0; q < r; ++q) v -= Lb[(i0 + r) * MP + i0 + q] * yb[q];
            yb[r] = v / Lb[(i0 + r) * MP + i0 + r];
            y[(i0 + r) * GMPC_THREADS] = yb[r];
          }
        }
        for (int i0 = MP - 8; i0 >= 0; i0 -= 8) {
          float acc[8], xb[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[r] = y[(i0 + r) * GMPC_THREADS];
          for (int k = i0 + 8; k < MP; k += 4) {
            const float y0 = y[(k + 0) * GMPC_THREADS], y1 = y[(k + 1) * GMPC_THREADS];
            const float y2 = y[(k + 2) * GMPC_THREADS], y3 = y[(k + 3) * GMPC_THREADS];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float4 l4 = *reinterpret_cast<const float4*>(&Ltb[(i0 + r) * MP + k]);
              acc[r] -= l4.x * y0; acc[r] -= l4.y * y1; acc[r] -= l4.z * y2; acc[r] -= l4.w * y3;
            }
          }
#pragma unroll
          for (int r = 7; r >= 0; --r) {
            float v = acc[r];
#pragma unroll
            for (int q = r + 1; q < 8; ++q) v -= Ltb[(i0 + r) * MP + i0 + q] * xb[q];
            xb[r] = v / Lb[(i0 + r) * MP + i0 + r];
            y[(i0 + r) * GMPC_THREADS] = xb[r];
          }
        }
        if (c < n) {
          for (int i0 = 0; i0 < MP; i0 += 8) {
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;
            for (int k = 0; k < MP; k += 4) {
              const float y0 = -y[(k + 0) * GMPC_THREADS], y1 = -y[(k + 1) * GMPC_THREADS];
              const float y2 = -y[(k + 2) * GMPC_THREADS], y3 = -y[(k + 3) * GMPC_THREADS];
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const float4 g4 = *reinterpret_cast<const float4*>(&Gb[(i0 + r) * MP + k]);
                acc[r] = fmaf(g4.x, y0, acc[r]); acc[r] = fmaf(g4.y, y1, acc[r]);
                acc[r] = fmaf(g4.z, y2, acc[r]); acc[r] = fmaf(g4.w, y3, acc[r]);
              }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int i = i0 + r;
              if (i >= m) continue;
              const float kic = -y[i * GMPC_THREADS];
              const float vic = fmaf(0.5f, acc[r], HG[(size_t)i * nm + c]);
              Kt[(size_t)i * n + c] = kic;
              KV[(size_t)i * n + c] = kic;
              KV[(size_t)(m + i) * n + c] = vic;
              VK[(size_t)i * n + c] = vic;
              VK[(size_t)(m + i) * n + c] = kic;
            }
          }
        } else {
          for (int i = 0; i < m; ++i) { kv[i] = -y[i * GMPC_THREADS]; a.k[bt * m + i] = -y[i * GMPC_THREADS]; }
        }
      }
    }
  } else {
    // LU with partial pivoting (jax.scipy.linalg.solve as the reference calls it), in place in L:
    // unit-lower multipliers below the diagonal, U on and above; perm = row order
    for (int e = tid; e < m * m; e += blockDim.x) L[e] = G[e];
    for (int i = tid; i < m; i += blockDim.x) perm[i] = i;
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      if (tid == 0) {
        int piv = j;
        float best = fabsf(L[j * m + j]);
        for (int i = j + 1; i < m; ++i)
          if (fabsf(L[i * m + j]) > best) { best = fabsf(L[i * m + j]); piv = i; }
        red[0] = (float)piv;
        if (piv != j) { const int tp = perm[j]; perm[j] = perm[piv]; perm[piv] = tp; }
      }
      __syncthreads();
      const int piv = (int)red[0];
      if (piv != j)
        for (int c = tid; c < m; c += blockDim.x) {
          const float tv_ = L[j * m + c]; L[j * m + c] = L[piv * m + c]; L[piv * m + c] = tv_;
        }
      __syncthreads();
      const float d = L[j * m + j];
      for (int i = j + 1 + tid; i < m; i += blockDim.x) L[i * m + j] /= d;
      __syncthreads();
      const int rem = m - j - 1;
      for (int e = tid; e < rem * rem; e += blockDim.x) {
        const int i = j + 1 + e / rem, k = j + 1 + e % rem;
        L[i * m + k] -= L[i * m + j] * L[j * m + k];
      }
      __syncthreads();
    }
    for (int c = tid; c <= n; c += blockDim.x) {
      for (int i = 0; i < m; ++i) {
        const int pi = perm[i];
        float v = c < n ? HG[(size_t)pi * nm + c] : hv[pi];
        for (int k = 0; k < i; ++k) v -= L[i * m + k] * y[k * GMPC_THREADS];
        y[i * GMPC_THREADS] = v;
      }
      for (int i = m - 1; i >= 0; --i) {
        float v = y[i * GMPC_THREADS];
        for (int k = i + 1; k < m; ++k) v -= L[i * m + k] * y[k * GMPC_THREADS];
        y[i * GMPC_THREADS] = v / L[i * m + i];
      }
      if (c < n) {
        // K column, V = H + G K / 2 and the stacked operands [K; V], [V; K] of the cross-term product
        // (the column is still in LDS: no global re-reads)
        for (int i = 0; i < m; ++i) {
          float v = 0.f;
          for (int k = 0; k < m; ++k) v = fmaf(G[i * m + k], -y[k * GMPC_THREADS], v);
          const float kic = -y[i * GMPC_THREADS];
          const float vic = fmaf(0.5f, v, HG[(size_t)i * nm + c]);
          Kt[(size_t)i * n + c] = kic;
          KV[(size_t)i * n + c] = kic;
          KV[(size_t)(m + i) * n + c] = vic;
          VK[(size_t)i * n + c] = vic;
          VK[(size_t)(m + i) * n + c] = kic;
        }
      } else {
        for (int i = 0; i < m; ++i) { kv[i] = -y[i * GMPC_THREADS]; a.k[bt * m + i] = -y[i * GMPC_THREADS]; }
      }
    }
  }
  BS_STAMP(7)
  __syncthreads();
  for (int i = tid; i < m; i += blockDim.x) {
    float v = 0.f;
    for (int k = 0; k < m; ++k) v = fmaf(G[i * m + k], kv[k], v);
    gk[i] = v + hv[i];
  }
  __syncthreads();
  // p = q + A^T p + H^T k + K^T (G k + h)      [= q + A^T p + (H+GK)^T k + K^T h, G symmetric]
  for (int c = tid; c < n; c += blockDim.x) {
    float v1 = 0.f, v2 = 0.f;
    // (eight rows at a time -- 16 loads in flight -- measured slower: 99 k vs 86 k cycles at the C5 shard)
    for (int i = 0; i < m; ++i) {
      v1 = fmaf(HG[(size_t)i * nm + c], kv[i], v1);
      v2 = fmaf(Kt[(size_t)i * n + c], gk[i], v2);   // own column: written by this thread above
    }
    a.pvec[(size_t)b * n + c] = (((m1 ? 0.f : qv[c]) + pa[c]) + v1) + v2;
  }
#ifdef GMPC_BIGSTEP_STAMPS
  BS_STAMP(8)
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.t == a.T - 2)
    printf("k_big_step cycles: load %llu | y=WL v %llu | g,h %llu | lam,pa %llu | G %llu | chol %llu | solve %llu | gk,p %llu | total %llu\n",
           bs_t[1] - bs_t[0], bs_t[2] - bs_t[1], bs_t[3] - bs_t[2], bs_t[4] - bs_t[3], bs_t[5] - bs_t[4], bs_t[6] - bs_t[5],
           bs_t[7] - bs_t[6], bs_t[8] - bs_t[7], bs_t[8] - bs_t[0]);
#endif
}

// P = Q_t + T1 on the upper triangle, mirrored into the lower one: tile (I, J) with I <= J (64 x 64) is read
// row-wise once and written twice (the transposed copy through LDS), so every global access is
// coalesced and P comes out exactly symmetric.  T1's strictly lower blocks are never read.
#define GMPC_PU_TILE 64
__global__ __launch_bounds__(GMPC_THREADS) void k_big_pupdate(int n, int ng, int T, int t, const float* X,
                                                              const float* goal, const float* mpc_w,
                                                              const float* sbuf, const float* T1,
                                                              const int* active, float* P) {
  constexpr int TS = GMPC_PU_TILE;
  __shared__ float tA[TS][TS + 1], dI[TS], dJ[TS];
  // blockIdx.x enumerates the tile pairs I <= J only (row I holds nt - I of them): a workgroup that returns at once
  // still waits for its dispatch slot behind the ones before it
  const int nt_ = (n + TS - 1) / TS, b = blockIdx.z;
  int I = 0, J = blockIdx.x;
  while (J >= nt_ - I) { J -= nt_ - I; ++I; }
  J += I;
  if (active != nullptr && active[b] == 0) return;
  const int tx = threadIdx.x & (TS - 1), ty = threadIdx.x / TS;      // 64 columns x 4 rows per sweep
  const size_t o = (size_t)b * n * n;
  const size_t xb = ((size_t)b * (T + 1) + t) * n, gb = ((size_t)b * (T + 1) + t) * ng;
  if (threadIdx.x < TS) {
    const int i = I * TS + tx;
    dI[tx] = i < ng ? X[xb + i] - goal[gb + i] : 0.f;
  } else if (threadIdx.x < 2 * TS) {
    const int j = J * TS + tx;
    dJ[tx] = j < ng ? X[xb + j] - goal[gb + j] : 0.f;
  }
  __syncthreads();
  const float w1 = sigmoidf_(mpc_w[1]);
  const float s = sbuf[b];
  const float is = 1.f / s, is3 = 1.f / (s * s * s);
  constexpr int RS = GMPC_THREADS / TS;
  float tv[TS / RS];
  // all the tile's reads first (16 per thread in flight), then the stores
#pragma unroll
  for (int q = 0; q < TS / RS; ++q) {
    const int r = ty + RS * q, i = I * TS + r, j = J * TS + tx;
    // diagonal tiles: take the element of the upper triangle for both (i, j) and (j, i)
    const size_t src = i <= j ? (size_t)i * n + j : (size_t)j * n + i;
    tv[q] = (i < n && j < n) ? T1[o + src] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < TS / RS; ++q) {
    const int r = ty + RS * q, i = I * TS + r, j = J * TS + tx;
    float v = 0.f;
    if (i < n && j < n) {
      const float di = dI[r], dj = dJ[tx];
      v = w1 * ((i == j && i < ng ? is : 0.f) - di * dj * is3) + tv[q];
      P[o + (size_t)i * n + j] = v;
    }
    tA[r][tx] = v;
  }
  __syncthreads();
  if (I != J) {
#pragma unroll
    for (int q = 0; q < TS / RS; ++q) {
      const int r = ty + RS * q, i = J * TS + r, j = I * TS + tx;     // transposed tile
      if (i < n && j < n) P[o + (size_t)i * n + j] = tA[tx][r];
    }
  }
}

__global__ void k_big_init(int B, int n, int T, const float* QT, const float* qT, const int* active,
                           float* P, float* pvec, float* lam, float* adj, float* gn2, const float* lx) {
  const int b = blockIdx.y;
  if (active != nullptr && active[b] == 0) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * n) P[(size_t)b * n * n + e] = QT[(size_t)b * n * n + e];
  if (e < n) {
    if (lx == nullptr) {
      const float q = qT[(size_t)b * n + e];
      pvec[(size_t)b * n + e] = q;
      lam[(size_t)b * n + e] = q;
      adj[((size_t)b * (T + 1) + T) * n + e] = q;
    } else {          // bilevel solve: value vector 0, loss adjoint d loss / d x_T
      pvec[(size_t)b * n + e] = 0.f;
      lam[(size_t)b * n + e] = lx[((size_t)b * (T + 1) + T) * n + e];
    }
  }
  if (e == 0) gn2[b] = 0.f;
}

// trajax continuation test (same rule as the tail of k_riccati)
__global__ void k_big_cont(int B, int T, int m, const float* U, const float* gn2, const int* iters,
                           const float* obj, const float* alpha, const float* obj_step,
                           const float* U_step, gmpc_ilqr_opts opts, const int* active, int* cont) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (active != nullptr && active[b] == 0) return;
  float un2 = 0.f;
  for (int e = 0; e < T * m; ++e) { const float u = U[(size_t)b * T * m + e]; un2 = fmaf(u, u, un2); }
  float gn = sqrtf(gn2[b]);
  if (isnan(gn)) gn = INFINITY;
  const float aobj = fabsf(obj[b]) + 1.0f;
  const float un = sqrtf(un2) + 1.0f;
  const bool progressing = (obj_step[b] > opts.obj_step_threshold * aobj) &&
                           (U_step[b] > opts.inputs_step_threshold * un);
  const bool potential = (gn > opts.grad_norm_threshold) && (gn > opts.relative_grad_norm_threshold * aobj);
  cont[b] = ((iters[b] < opts.maxiter) && progressing && potential && (alpha[b] > opts.alpha_min)) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// host driver of one backward pass
// ------------------------------------------------------------------------------------------------
int gmpc_launch_linearize_regs(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                               const uint32_t* masks, const int* active, float* AB, int samp_mul,
                               int samp_add, hipStream_t s, hipEvent_t mid_event = nullptr);
int gmpc_launch_linearize_mfma(int NSamp, int T, int n, int m, const MlpDesc& dyn, const LinPad& lp,
                               const uint32_t* masks, const int* active, float* AB, int samp_mul,
                               int samp_add, hipStream_t s);

// LSTM dynamics variant (gmpc_dynl.hip): the per-step Jacobians come from its kernel instead of the chain
void gmpc_launch_dynl_jac(int, int, int, int, const DynlDesc&, const float*, const float*, const int*, float*,
                          hipStream_t);
void gmpc_launch_dynl_curv(int, int, int, int, const DynlDesc&, const float*, const float*, const float*, const int*,
                           float*, hipStream_t);
void gmpc_launch_add_phi(int, int, int, const float*, float*, float*, hipStream_t);

static void big_lowrank_factors(const BigWork& w, int B, const MlpDesc& dyn, const uint32_t* masks, int t,
                                const int* active, hipStream_t s);
// out[b][c][r] = in[b][r][c], 64 x 64 tiles through LDS
__global__ __launch_bounds__(GMPC_THREADS) void k_btranspose(int R, int C, const float* in, float* out,
                                                             const int* active) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  if (active != nullptr && active[b] == 0) return;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + (size_t)b * R * C;
  float* dst = out + (size_t)b * R * C;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + ty + 4 * q, c = c0 + tx;
    tile[ty + 4 * q][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = c0 + ty + 4 * q, r = r0 + tx;
    if (r < R && c < C) dst[(size_t)c * R + r] = tile[tx][ty + 4 * q];
  }
}
__global__ void k_add_identity(int n, int ld, const int* active, float* M) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (active != nullptr && active[b] == 0)) return;
  M[(size_t)b * n * ld + (size_t)i * ld + i] += 1.f;
}

int gmpc_big_backward(const BigWork& w, int B, const MlpDesc& dyn, const LinPad& lp,
                      const uint32_t* masks, const float* X, const float* U, const float* goal,
                      const float* mpc_w, const float* QT, const float* qT, const int* active, float* K,
                      float* k, float* grad, float* adj, const float* lx, float* Bvec, hipStream_t s,
                      const DynlDesc* dl, const float* lam_sol) {
  // lam_sol (bilevel solve of the LSTM dynamics only): the adjoints of the rollout objective at the solution;
  // the step's curvature Phi = lam_{t+1} . d^2 f joins R, M^T (through [H | G_r]) and Q (through T1)
  const bool curv = dl != nullptr && lx != nullptr && lam_sol != nullptr && w.Phi != nullptr;
  // lx != null: the bilevel Hessian solve (k_big_step mode 1); grad / adj are not written then
  const int n = w.n, m = w.m, T = w.T, nm = n + m;
  const dim3 ge((n * n + 255) / 256, B);
  hipLaunchKernelGGL(k_big_init, ge, dim3(256), 0, s, B, n, T, QT, qT, active, w.P, w.pvec, w.lam, adj,
                     w.gn2, lx);
  const bool lowrank = w.h > 0 && dl == nullptr;
  const int h = lowrank ? w.h : 0;
  const size_t lds = big_step_lds(n, m, h);
  if (lds > 159 * 1024) return -2;     // one workgroup per CU may take (almost) all of the 160 KB
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_step),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipGetLastError();
    attr = true;
  }
  const long snn = (long)n * n, snm = (long)n * nm, smn = (long)m * n, smnm = (long)m * nm;
  auto gemm = [&](int M, int N, int Kk, const float* Xp, long sx, int ldx, const float* Yp, long sy, int ldy,
                  float* Cp, long sc, int ldc) {
    BgemmArgs g;
    g.batch = B; g.M = M; g.N = N; g.K = Kk;
    g.X = Xp; g.sx = sx; g.ldx = ldx; g.Y = Yp; g.sy = sy; g.ldy = ldy; g.C = Cp; g.sc = sc; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.active = active;
    return g;
  };
  const int nt = (n + GMPC_PU_TILE - 1) / GMPC_PU_TILE;
  // (read per pass, not once: the tests compare the two forms of the gain solve inside one process)
  const char* sv_env = getenv("GMPC_BIG_SOLVE");
  const bool solve_valu = sv_env != nullptr && strcmp(sv_env, "valu") == 0;
  // One step ahead on a side stream: the Jacobians of a step ([A | B], or the factor V^T of the low-rank form) do
  // not depend on P.  Step t's are produced into copy t & 1 of their buffer while step t + 1's products run on the
  // caller's stream -- the Jacobian chain / the factor GEMMs fill the matrix pipe under k_big_step, k_big_pupdate,
  // the thin products and the transposes, which leave it idle.  ev_ready[i]: copy i is written; ev_free[i]: the point of
  // the caller's stream behind which copy i may be overwritten.
  const bool pipe = w.side != nullptr && dl == nullptr && (lowrank ? w.Vt2 != nullptr : w.ABt2 != nullptr);
  float* const ABc[2] = {w.ABt, pipe && !lowrank ? w.ABt2 : w.ABt};
  float* const Vtc[2] = {w.Vt, pipe && lowrank ? w.Vt2 : w.Vt};
  int rc_side = 0;
  auto jacobians = [&](int t, hipStream_t st) {
    if (lowrank) {
      BigWork wv = w;
      wv.Vt = Vtc[t & 1];
      big_lowrank_factors(wv, B, dyn, masks, t, active, st);
    } else if (gmpc_launch_linearize_regs(B, T, n, m, dyn, lp, masks, active, ABc[t & 1], T, t, st) != 0 &&
               gmpc_launch_linearize_mfma(B, T, n, m, dyn, lp, masks, active, ABc[t & 1], T, t, st) != 0) {
      rc_side = -1;
    }
  };
  // event / wait failures: a missed wait would be a silent race on the Jacobian copies, so they end the pass; on any
  // error after the fork the caller's stream first joins the side stream (nothing is left in flight on it)
  auto ev = [&](hipError_t e) { if (e != hipSuccess) rc_side = -1; };
  auto bail = [&]() -> int {
    if (pipe) (void)hipStreamSynchronize(w.side);
    (void)hipGetLastError();
    return -1;
  };
  if (pipe) {
    ev(hipEventRecord(w.ev_start, s));                   // the side stream starts behind the caller's earlier work
    ev(hipStreamWaitEvent(w.side, w.ev_start, 0));
    jacobians(T - 1, w.side);
    ev(hipEventRecord(w.ev_ready[(T - 1) & 1], w.side));
  }
  for (int t = T - 1; t >= 0; --t) {
    const float* A = ABc[t & 1];
    const float* Bm = ABc[t & 1] + n;
    const float* Vt_t = Vtc[t & 1];
    const long shn = (long)h * n, shnm = (long)h * nm;
    const float* WLT = lowrank ? dyn.WT[dyn.L - 1] : nullptr;     // [n][h]: W_L^T, the TN left operand of W_L (.)
    // step t - 1's Jacobians start when step t reaches its stretch of kernels that leave the matrix pipe idle (the
    // thin products, k_big_step, ...): started at the top of the step they only shared the pipe with the step's
    // first big GEMM, both at half speed (kernel trace: linearize 0.66 ms beside PA 1.04 ms, then 0.36 ms of thin
    // products and k_big_step alone)
    auto start_next = [&]() {
      if (!pipe || t == 0) return;
      ev(hipEventRecord(w.ev_free[(t - 1) & 1], s));        // (the copy's last reader was step t + 1; this point is later)
      ev(hipStreamWaitEvent(w.side, w.ev_free[(t - 1) & 1], 0));
      jacobians(t - 1, w.side);
      ev(hipEventRecord(w.ev_ready[(t - 1) & 1], w.side));
    };
    if (pipe) ev(hipStreamWaitEvent(s, w.ev_ready[t & 1], 0));
    else if (dl == nullptr) jacobians(t, s);
    if (rc_side != 0) return bail();
    if (lowrank) {
      // the factors V_t^T (above), then the n^3 products through them (see big_lowrank_factors)
      // Y = W_L P, S = W_L P W_L^T = W_L Y^T, then with Z = Y + S Vx^T / 2:
      //     A^T P A = P + Vx Z + Z^T Vx^T            (T1 below: two K-segments of h rows, a third for K, V)
      //     [H | Gr] = B^T P [A | B] = Vu ([Y | 0] + S V^T) = Vu (2 [Z | S Vu^T / 2] - [Y | 0])
      // 2 h n^2 + 4 h^2 n + 2.2 n^2 h flops instead of the 7.1 n^2 h of [PA | PB] = P [A | B], W_L [PA | PB],
      // A^T (PA) written out (C5: 1.05 instead of 1.49 Gflop per trajectory and step)
      gmpc_launch_bgemm_tn(gemm(h, n, n, WLT, 0, h, w.P, snn, n, w.W1b, shn, n), s);            // Y = W_L P
      float* Yt = w.PAB;                                                                        // [n][h]
      hipLaunchKernelGGL(k_btranspose, dim3((n + 63) / 64, (h + 63) / 64, B), dim3(GMPC_THREADS), 0, s, h, n,
                         w.W1b, Yt, active);
      float* S = pipe ? w.Sm : w.Sa;                          // [h][h] (Sa / Sb are the factor products' scratch)
      const long shh = (long)h * h;
      gmpc_launch_bgemm_tn(gemm(h, h, n, WLT, 0, h, Yt, shn, h, S, shh, h), s);                 // S = W_L Y^T
      // W2b = S V^T / 2 + [Y | 0] = [Z | S Vu^T / 2]   (S symmetric up to rounding: S^T V^T is the TN form)
      BgemmArgs g2 = gemm(h, n, h, S, shh, h, Vt_t, shnm, nm, w.W2b, shnm, nm);
      g2.alpha = 0.5f; g2.E = w.W1b; g2.se = shn; g2.lde = n; g2.En = n;
      gmpc_launch_bgemm_tn(g2, s);
      BgemmArgs g3 = gemm(h, m, h, S, shh, h, Vt_t + n, shnm, nm, w.W2b + n, shnm, nm);
      g3.alpha = 0.5f;
      gmpc_launch_bgemm_tn(g3, s);
      BgemmArgs g4 = gemm(m, nm, h, Vt_t + n, shnm, nm, w.W2b, shnm, nm, w.HG, smnm, nm);       // 2 Vu W2b
      g4.alpha = 2.f;
      gmpc_launch_bgemm_tn(g4, s);
      BgemmArgs g5 = gemm(m, n, h, Vt_t + n, shnm, nm, w.W1b, shn, n, w.HG, smnm, nm);          // - Vu [Y | 0]
      g5.alpha = -1.f; g5.beta = 1.f;
      gmpc_launch_bgemm_tn(g5, s);
      start_next();
    } else {
    if (dl) gmpc_launch_dynl_jac(B, T, 1, t, *dl, X, U, active, w.ABt, s);
    // [PA | PB] = P [A | B]   (P symmetric, so P = P^T is the "TN" left operand)
    gmpc_launch_bgemm_tn(gemm(n, n, n, w.P, snn, n, A, snm, nm, w.PAB, snm, nm), s);
    gmpc_launch_bgemm_tn(gemm(n, m, n, w.P, snn, n, Bm, snm, nm, w.PAB + n, snm, nm), s);
    // [H | Gr] = B^T [PA | PB]
    gmpc_launch_bgemm_tn(gemm(m, nm, n, Bm, snm, nm, w.PAB, snm, nm, w.HG, smnm, nm), s);
    // (behind the two thin products: started behind PA the chain stretched [H | G_r] from 0.09 to 0.6 ms -- C4 98.8 /
    // 100.3 / 98.0 ms for a start behind PA / PB / [H | G_r], 99.6 without the side stream)
    start_next();
    }
    if (curv) {
      gmpc_launch_dynl_curv(B, T, 1, t, *dl, X, U, lam_sol, active, w.Phi, s);
      gmpc_launch_add_phi(B, n, m, w.Phi, w.HG, nullptr, s);
    }
    BigStepArgs a;
    a.Vt = lowrank ? Vt_t : nullptr; a.WL = lowrank ? dyn.W[dyn.L - 1] : nullptr; a.h = h;
    a.B = B; a.n = n; a.m = m; a.T = T; a.t = t;
    a.mode = lx != nullptr ? 1 : 0; a.lx = lx; a.Bvec = Bvec;
    a.X = X; a.U = U; a.goal = goal; a.ng = w.ng; a.mpc_w = mpc_w; a.ABt = A; a.HG = w.HG; a.KV = w.KV; a.VK = w.VK;
    a.pvec = w.pvec; a.lam = w.lam; a.sbuf = w.sbuf; a.gn2 = w.gn2; a.active = active;
    a.K = K; a.k = k; a.grad = grad; a.adj = adj;
    a.solve_valu = solve_valu ? 1 : 0;
    hipLaunchKernelGGL(k_big_step, dim3(B), dim3(GMPC_THREADS), lds, s, a);
    // T1 = A^T (PA) + [K; V]^T [V; K], upper blocks only   (low-rank form: P + Vx Z + Z^T Vx^T)
    BgemmArgs g = lowrank ? gemm(n, n, h, Vt_t, shnm, nm, w.W2b, shnm, nm, w.T1, snn, n)
                          : gemm(n, n, n, A, snm, nm, w.PAB, snm, nm, w.T1, snn, n);
    g.X2 = w.KV; g.sx2 = 2 * smn; g.ldx2 = n;
    g.Y2 = w.VK; g.sy2 = 2 * smn; g.ldy2 = n; g.K2 = 2 * m;
    if (lowrank) {
      g.E = w.P; g.se = snn; g.lde = n; g.En = n;
      g.X3 = w.W2b; g.sx3 = shnm; g.ldx3 = nm; g.Y3 = Vt_t; g.sy3 = shnm; g.ldy3 = nm; g.K3 = h;
    }
    static const bool full_t1 = getenv("GMPC_BIG_FULL_T1") != nullptr;   // A/B timing only
    g.upper_only = full_t1 ? 0 : 1;
    gmpc_launch_bgemm_tn(g, s);
    if (curv) gmpc_launch_add_phi(B, n, m, w.Phi, nullptr, w.T1, s);
    hipLaunchKernelGGL(k_big_pupdate, dim3(nt * (nt + 1) / 2, 1, B), dim3(GMPC_THREADS), 0, s, n, w.ng > 0 ? w.ng : n, T, t, X,
                       goal, mpc_w,
                       w.sbuf, w.T1, active, w.P);
  }
  if (lowrank) {
    // keep the documented content of the step buffer: [A_0 | B_0] = [I | 0] + W_L^T V_0^T of the last step
    // processed (gmpc_debug_buffer 5, the `lqr` slot of the host mirror) -- one thin-K GEMM per pass
    const long shnm = (long)h * nm;
    gmpc_launch_bgemm_tn(gemm(n, nm, h, dyn.W[dyn.L - 1], 0, n, w.Vt, shnm, nm, w.ABt, snm, nm), s);
    hipLaunchKernelGGL(k_add_identity, dim3((n + 255) / 256, B), dim3(256), 0, s, n, nm, active, w.ABt);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// forward tangent roll of the bilevel solve: dU_t = k_t + K_t dX_t ; dX_{t+1} = A_t dX_t + B_t dU_t
// (oracle hessian_solve, second loop).  One workgroup per trajectory and time step; [A|B] of the step
// comes from the same strided Jacobian chain as in the backward pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GMPC_THREADS) void k_big_fwd(int n, int m, int T, int t, const float* ABt,
                                                          const float* K, const float* k, float* Hout,
                                                          float* dX, const float* Vtb, const float* WL, int h) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dx = reinterpret_cast<float*>(smem);   // n
  float* du = dx + n;                           // m
  float* zv = du + m;                           // h   V^T [dx; du]   (low-rank form)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nm = n + m;
  const size_t bt = (size_t)b * T + t;
  for (int i = tid; i < n; i += blockDim.x) dx[i] = t == 0 ? 0.f : dX[((size_t)b * (T + 1) + t) * n + i];
  __syncthreads();
  for (int j = wave; j < m; j += GMPC_THREADS / 64) {
    const float* Kr = K + (bt * m + j) * n;
    float v = 0.f;
    for (int i = lane; i < n; i += 64) v = fmaf(Kr[i], dx[i], v);
    v = wave_sum(v);
    if (lane == 0) { const float u = k[bt * m + j] + v; du[j] = u; Hout[bt * m + j] = u; }
  }
  if (t == 0)
    for (int i = tid; i < n; i += blockDim.x) dX[(size_t)b * (T + 1) * n + i] = 0.f;
  __syncthreads();
  if (Vtb != nullptr) {
    // dX' = dx + W_L^T (V^T [dx; du])
    const float* Vt = Vtb + (size_t)b * h * nm;
    for (int kk = wave; kk < h; kk += GMPC_THREADS / 64) {
      const float* row = Vt + (size_t)kk * nm;
      float v = 0.f;
      for (int c = lane; c < nm; c += 64) v = fmaf(row[c], c < n ? dx[c] : du[c - n], v);
      v = wave_sum(v);
      if (lane == 0) zv[kk] = v;
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
      float v = dx[i];
      for (int kk = 0; kk < h; ++kk) v = fmaf(WL[(size_t)kk * n + i], zv[kk], v);
      dX[((size_t)b * (T + 1) + t + 1) * n + i] = v;
    }
    return;
  }
  const float* AB = ABt + (size_t)b * n * nm;
  for (int i = wave; i < n; i += GMPC_THREADS / 64) {
    const float* row = AB + (size_t)i * nm;
    float v = 0.f;
    for (int c = lane; c < nm; c += 64) v = fmaf(row[c], c < n ? dx[c] : du[c - n], v);
    v = wave_sum(v);
    if (lane == 0) dX[((size_t)b * (T + 1) + t + 1) * n + i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Low-rank form of the per-step Jacobians.  For the relu MLP  J = W_L^T D_{L-2} W_{L-2}^T ... D_0 W_0^T, so
//   A_t = I + W_L^T Vx_t^T,   B_t = W_L^T Vu_t^T,   V_t^T = D_{L-2} W_{L-2}^T ... D_0 W_0^T   [h][n+m],
// with h the last hidden width and W_L^T shared by every sample.  When h < n / 2 (C5: 200 vs 1024) the n^3
// products of the Riccati step go through the factors:
//   W1 = W_L P                    [h][n]      2 h n^2
//   [PA | PB] = [P | 0] + W1^T V^T            2 n h (n+m)
//   W2 = W_L [PA | PB]            [h][n+m]    2 h n (n+m)
//   [H | G_r] = Vu W2             (thin)
//   T1 = PA + Vx W2 + K^T W                   2 n^2 h + 2 n^2 (2m)
// 8 n^2 h instead of 4 n^3 flops (2.56x fewer at C5), and V^T itself costs 2 h (sum h_l h_{l+1} + h (n+m))
// instead of the n-row chain.  V^T is built by masked products: S^T_{L-3}[k][j] = d_{L-3}[k] W_{L-2}[k][j]
// d_{L-2}[j] elementwise, S^T_{l-1} = rowmask_{l-1}(W_l S^T_l) and V^T = S_0 W_0^T as batched GEMMs.
// ------------------------------------------------------------------------------------------------
__global__ void k_mask_scale(int K, int N, const float* W, const uint32_t* maskK, const uint32_t* maskN,
                             long smask, float* out) {
  // out[b][k][j] = W[k][j] if bit k of maskK[b] (optional) and bit j of maskN[b] (optional) are set, else 0
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * N) return;
  const int kk = (int)(e / N), j = (int)(e - (long)kk * N);
  bool on = true;
  if (maskK) on = on && ((maskK[(size_t)b * smask + (kk >> 5)] >> (kk & 31)) & 1u);
  if (maskN) on = on && ((maskN[(size_t)b * smask + (j >> 5)] >> (j & 31)) & 1u);
  out[(size_t)b * K * N + e] = on ? W[e] : 0.f;
}

// V_t^T for the B samples of step t into w.Vt
static void big_lowrank_factors(const BigWork& w, int B, const MlpDesc& dyn, const uint32_t* masks, int t,
                                const int* active, hipStream_t s) {
  const int L = dyn.L, T = w.T, nm = w.n + w.m, h = w.h, Lh = L - 1;
  const long smask = (long)T * Lh * GMPC_MW;
  auto mk = [&](int l) { return masks + ((size_t)t * Lh + l) * GMPC_MW; };   // + b * smask inside the kernels
  auto gemm = [&](int M, int N, int K, const float* X, long sx, int ldx, const float* Y, long sy, int ldy, float* C,
                  long sc, int ldc) {
    BgemmArgs g;
    g.batch = B; g.M = M; g.N = N; g.K = K;
    g.X = X; g.sx = sx; g.ldx = ldx; g.Y = Y; g.sy = sy; g.ldy = ldy; g.C = C; g.sc = sc; g.ldc = ldc;
    g.alpha = 1.f; g.beta = 0.f; g.active = active;
    return g;
  };
  if (L == 2) {          // one hidden layer: V^T = D_0 W_0^T
    const long cnt = (long)h * nm;
    hipLaunchKernelGGL(k_mask_scale, dim3((unsigned)((cnt + 255) / 256), B), dim3(256), 0, s, h, nm, dyn.WT[0], mk(0),
                       nullptr, smask, w.Vt);
    return;
  }
  // S^T_{L-3} [dims[L-2]][h]
  float* cur = w.Sa;
  float* nxt = w.Sb;
  {
    const int K = dyn.dims[L - 2];
    const long cnt = (long)K * h;
    hipLaunchKernelGGL(k_mask_scale, dim3((unsigned)((cnt + 255) / 256), B), dim3(256), 0, s, K, h, dyn.W[L - 2],
                       mk(L - 3), mk(L - 2), smask, cur);
  }
  for (int l = L - 3; l >= 1; --l) {
    // S^T_{l-1} [dims[l]][h] = rowmask_{l-1}( W_l S^T_l ):  X = WT[l] ([dims[l+1]][dims[l]], shared), Y = S^T_l
    BgemmArgs g = gemm(dyn.dims[l], h, dyn.dims[l + 1], dyn.WT[l], 0, dyn.dims[l], cur, (long)dyn.dims[l + 1] * h, h,
                       nxt, (long)dyn.dims[l] * h, h);
    g.rowmask = mk(l - 1); g.srm = smask;
    gmpc_launch_bgemm_tn(g, s);
    float* sw = cur; cur = nxt; nxt = sw;
  }
  // V^T [h][n+m] = S_0 W_0^T:  X = S^T_0 ([dims[1]][h]), Y = WT[0] ([dims[1]][n+m], shared)
  gmpc_launch_bgemm_tn(gemm(h, nm, dyn.dims[1], cur, (long)dyn.dims[1] * h, h, dyn.WT[0], 0, nm, w.Vt, (long)h * nm,
                            nm), s);
}

int gmpc_big_forward_tangent(const BigWork& w, int B, const MlpDesc& dyn, const LinPad& lp,
                             const uint32_t* masks, const float* K, const float* k, float* Hout, float* dX,
                             hipStream_t s, const DynlDesc* dl, const float* X, const float* U) {
  const int n = w.n, m = w.m, T = w.T;
  const bool lowrank = w.h > 0 && dl == nullptr;
  for (int t = 0; t < T; ++t) {
    if (lowrank) {
      big_lowrank_factors(w, B, dyn, masks, t, nullptr, s);
    } else if (dl) {
      gmpc_launch_dynl_jac(B, T, 1, t, *dl, X, U, nullptr, w.ABt, s);
    } else if (gmpc_launch_linearize_regs(B, T, n, m, dyn, lp, masks, nullptr, w.ABt, T, t, s) != 0 &&
               gmpc_launch_linearize_mfma(B, T, n, m, dyn, lp, masks, nullptr, w.ABt, T, t, s) != 0) return -1;
    hipLaunchKernelGGL(k_big_fwd, dim3(B), dim3(GMPC_THREADS), (size_t)(n + m + (lowrank ? w.h : 0)) * sizeof(float),
                       s, n, m, T, t, w.ABt, K, k, Hout, dX, lowrank ? w.Vt : nullptr,
                       lowrank ? dyn.W[dyn.L - 1] : nullptr, lowrank ? w.h : 0);
  }
  return 0;
}

void gmpc_launch_big_cont(int B, int T, int m, const float* U, const float* gn2, const int* iters,
                          const float* obj, const float* alpha, const float* obj_step,
                          const float* U_step, const gmpc_ilqr_opts& opts, const int* active, int* cont,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_big_cont, dim3((B + 63) / 64), dim3(64), 0, s, B, T, m, U, gn2, iters, obj, alpha,
                     obj_step, U_step, opts, active, cont);
}
