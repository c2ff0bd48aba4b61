// Large-state path (n > 64: BASELINE configs C4 Humanoid n=376, C5 synthetic n=1024).
//
// The value matrix P of the Riccati recursion no longer fits a CU's LDS (565 KB at n=376, 4.2 MB at
// n=1024), and the per-step LQR blocks cannot all be materialised either (AB is 15 GB
// per GPU at C4, 460 GB at C5).  The backward pass therefore runs STEP-MAJOR: for t = T-1 .. 0, for
// the whole batch at once,
//     [A_t | B_t]  <- MFMA Jacobian chain on the samples (b, t)          (k_linearize_mfma, strided)
//     P A, P B, A^T(PA), B^T(PA), B^T(PB)  <- batched fp32-MFMA GEMMs    (k_bgemm_tn)
//     gains, adjoint, value-vector update per trajectory                 (k_big_step)
//     (H+GK)^T K + K^T H   <- batched GEMMs (K = m)                      (k_bgemm_tn)
//     P <- sym(Q + sym(A^T P A) + ...)                                   (k_big_pupdate)
// Every product is written as  C = sum_k X[k][:]^T Y[k][:]  ("TN") with row-major operands, so row k
// of X / Y IS the MFMA A / B operand of k-step k and all global reads are coalesced; P's symmetry
// turns P A into that form (X = P).  Reference arithmetic: trajax lqr_step / tvlqr / adjoint.
#include <cstdlib>

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
  const int K = a.K, Kp = (K + 1) & ~1;
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
  gemm_tile<NTW>(bp0, a.ldy, Kp, afn, acc);
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

void gmpc_launch_bgemm_tn(const BgemmArgs& a, hipStream_t s) {
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
