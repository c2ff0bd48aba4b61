// Critic (discriminator) kernels: LSTM(F) over the T+1 rows of each sequence, final h -> relu head ->
// score; BCE / generator loss; BPTT; weight gradients as row-sum GEMMs; clip + Adam.  First generation of the LSTM
// kernels (any input width; the n <= 32 shapes run gmpc_critic_lstm.hip since round 3), the weight-gradient GEMMs,
// the optimiser kernels.
//
// Reference arithmetic: critic/nn.py:28-42 (flax OptimizedLSTMCell scanned over the sequence, zero
// carry, gate order i,f,g,o), gan/js_policy.py:41-68 (losses), gan/runner.py:51-63 (optimiser).
//
// Layout: a workgroup of 256 threads owns SB = 4*R4 sequences for the whole sweep.  Thread j
// produces gate pre-activation j (4F = 256 with the reference's F = 64) for all SB sequences from
// the concatenated kernel Wcat = [Wx; Wh] ((n+F) x 4F, exactly the flat critic layout), the cell
// update runs as thread (unit, wave).  Saved per (sequence, step): activated gates, c_t, h_{t-1}.
#include "gmpc_device.h"
#include <cstdlib>


// NXR > 0: the input size n is known at compile time and thread j keeps column j of [Wx; Wh]
// (n + F floats) in registers for the whole sequence -- the weights are read from memory once per
// workgroup instead of once per time step.  NXR == 0: run-time n, weights streamed from L2 per step.
template <int R4, int NXR>
__global__ __launch_bounds__(GMPC_THREADS) void k_lstm_fwd(int Bc, CriticDesc cd, const float* xseq,
                                                           float* gates, float* cs, float* hp,
                                                           float* hT, int stage_w, const float* xproj) {
  constexpr int SB = 4 * R4;
  constexpr int KC = NXR > 0 ? NXR + 64 : 1;
  float wreg[KC];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* act = reinterpret_cast<float4*>(smem);            // [(n+F)][R4]
  float4* gbuf = act + (cd.n + cd.F) * R4;                  // [4F][R4]
  float* wlds = reinterpret_cast<float*>(gbuf + 4 * cd.F * R4);   // [(n+F)][4F] when staged
  const int tid = threadIdx.x;
  const int n = cd.n, F = cd.F, T1 = cd.T1, G4 = 4 * F, K = n + F;
  const int s0 = blockIdx.x * SB;
  // the whole [Wx; Wh] block (83 KB at n = 17) lives in LDS for all T+1 steps: every step then
  // reads its weights at LDS latency instead of L2 latency
  if (stage_w)
    for (int e = tid; e < K * G4; e += blockDim.x) wlds[e] = cd.Wcat[e];
  const int u = tid % F, grp = tid / F;       // cell-update role (F*4 == blockDim)
  constexpr int SPT = SB / 4;                 // sequences per thread in the cell update
  float c[SPT];
#pragma unroll
  for (int e = 0; e < SPT; ++e) c[e] = 0.f;
  float* actf = reinterpret_cast<float*>(act);
  float* gbf = reinterpret_cast<float*>(gbuf);
  for (int e = tid; e < F * SB; e += blockDim.x) actf[n * SB + e] = 0.f;   // h_{-1} = 0
  const float bj = tid < G4 ? cd.b[tid] : 0.f;
  if (NXR > 0) {
#pragma unroll
    for (int k = 0; k < KC; ++k) wreg[k] = tid < G4 ? cd.Wcat[(size_t)k * G4 + tid] : 0.f;
  }
  // x_t of the next step is requested one step ahead when one element per thread covers it
  const bool xpf = n * SB <= (int)blockDim.x;
  const int xsb = tid / (n > 0 ? n : 1), xi = tid - xsb * n;
  const bool xon = xpf && tid < n * SB;
  const float* xptr = xseq + ((size_t)min(s0 + (xon ? xsb : 0), Bc - 1) * T1) * n + (xon ? xi : 0);
  float xnext = xon ? xptr[0] : 0.f;
  for (int t = 0; t < T1; ++t) {
    if (xpf) {
      if (xon) {
        actf[xi * SB + xsb] = xnext;
        if (t + 1 < T1) xnext = xptr[(size_t)(t + 1) * n];
      }
    } else {
      for (int e = tid; e < n * SB; e += blockDim.x) {
        const int sb = e / n, i = e - sb * n;
        const int s = min(s0 + sb, Bc - 1);
        actf[i * SB + sb] = xseq[((size_t)s * T1 + t) * n + i];
      }
    }
    __syncthreads();
    // save h_{t-1}
    for (int e = tid; e < F * SB; e += blockDim.x) {
      const int sb = e / F, k = e - sb * F;
      if (s0 + sb < Bc) hp[((size_t)(s0 + sb) * T1 + t) * F + k] = actf[(n + k) * SB + sb];
    }
    float4 acc[R4];
#pragma unroll
    for (int q = 0; q < R4; ++q) acc[q] = make_float4(bj, bj, bj, bj);
    if (NXR == 0 && xproj != nullptr && tid < G4) {
      // wide inputs: x_t Wx was formed by a GEMM beforehand (cd.n == 0 here, Wcat = Wh)
#pragma unroll
      for (int q = 0; q < R4; ++q) {
        float v[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int sq = min(s0 + q * 4 + cc, Bc - 1);
          v[cc] = xproj[((size_t)sq * T1 + t) * G4 + tid];
        }
        acc[q] = make_float4(bj + v[0], bj + v[1], bj + v[2], bj + v[3]);
      }
    }
    if constexpr (NXR > 0 && R4 == 1) {
      // 4 sequences: the [x ; h] image IS the broadcast A operand of v_mfma_f32_4x4x1 (gmpc_device.h),
      // this thread's weight column the B operand -- 6 LDS reads and 81 MFMAs per step instead of 81
      // broadcast ds_read_b128 and 324 FMAs per thread
      constexpr int NR = (4 * KC + 63) / 64;
      const int ln = tid & 63;
      float ar[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) ar[r] = actf[64 * r + ln];
      f32x4_t d0 = {bj, bj, bj, bj}, d1 = {0.f, 0.f, 0.f, 0.f};
      rw_static_for<KC>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k & 1) rw_mfma<k>(d1, ar[k >> 4], wreg[k]);
        else rw_mfma<k>(d0, ar[k >> 4], wreg[k]);
      });
      const f32x4_t d = d0 + d1;
      acc[0] = make_float4(d[0], d[1], d[2], d[3]);
    } else if constexpr (NXR > 0) {
#pragma unroll
      for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int q = 0; q < R4; ++q) fma4(acc[q], wreg[k], act[k * R4 + q]);
    } else if (stage_w) {
      dense_rows_lds<R4>(wlds, K, G4, tid, act, acc);
    } else {
      dense_rows<R4>(cd.Wcat, K, G4, tid, act, acc);
    }
    if (tid < G4) {
      const bool is_g = (tid >= 2 * F) && (tid < 3 * F);
#pragma unroll
      for (int q = 0; q < R4; ++q) {
        float4 v = acc[q];
        if (is_g) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        else { v.x = sigmoidf_(v.x); v.y = sigmoidf_(v.y); v.z = sigmoidf_(v.z); v.w = sigmoidf_(v.w); }
        gbuf[tid * R4 + q] = v;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int sb = q * 4 + cc;
          if (s0 + sb < Bc) gates[((size_t)(s0 + sb) * T1 + t) * G4 + tid] = f4get(v, cc);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < SPT; ++e) {
      const int sb = grp * SPT + e;
      const float ig = gbf[(0 * F + u) * SB + sb], fg = gbf[(1 * F + u) * SB + sb];
      const float gg = gbf[(2 * F + u) * SB + sb], og = gbf[(3 * F + u) * SB + sb];
      c[e] = fg * c[e] + ig * gg;
      const float h = og * tanhf(c[e]);
      actf[(n + u) * SB + sb] = h;
      if (s0 + sb < Bc) {
        cs[((size_t)(s0 + sb) * T1 + t) * F + u] = c[e];
        if (t == T1 - 1) hT[(size_t)(s0 + sb) * F + u] = h;
      }
    }
    __syncthreads();
  }
}

// (The head -- forward, loss, backward -- is k_head2 in gmpc_critic_lstm.hip for every shape since round 3.)

template <int R4, int NXR>
__global__ __launch_bounds__(GMPC_THREADS) void k_lstm_bwd(int Bc, CriticDesc cd, const float* gates,
                                                           const float* cs, const float* dhT,
                                                           float* dz, float* dxseq, int stage_w) {
  constexpr int SB = 4 * R4;
  constexpr int SPT = SB / 4;
  // register-resident [Wx; Wh]^T (NXR > 0): thread (j, seg) keeps its K-segment of row j
  constexpr int KC = NXR > 0 ? NXR + 64 : 1;                 // outputs of the product: [dx ; dh]
  constexpr int NSEG = NXR > 0 ? GMPC_THREADS / KC : 1;      // K = 4F = 256 split in NSEG segments
  constexpr int KSG = NXR > 0 ? (256 + NSEG - 1) / NSEG : 1;
  float wt[KSG];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* dzb = reinterpret_cast<float4*>(smem);            // [4F][R4]
  float4* part = dzb + (GMPC_THREADS + 128) * R4;           // dense_small scratch / result
  float* wlds = reinterpret_cast<float*>(part + (GMPC_THREADS + 128) * R4);   // [4F][(n+F)] when staged
  const int tid = threadIdx.x;
  const int n = cd.n, F = cd.F, T1 = cd.T1, G4 = 4 * F, K = n + F;
  const int s0 = blockIdx.x * SB;
  const int u = tid % F, grp = tid / F;
  float* dzf = reinterpret_cast<float*>(dzb);
  if (stage_w)
    for (int e = tid; e < K * G4; e += blockDim.x) wlds[e] = cd.WcatT[e];
  const int jr = NXR > 0 ? tid % KC : 0, segr = NXR > 0 ? tid / KC : 0;
  // MFMA form (R4 == 1): this lane's 128 transposed weights WcatT[128 (wave >> 1) + j][64 (wave & 1) + lane]
  float wtm[(NXR > 0 && R4 == 1) ? 128 : 1];
  if (NXR > 0 && R4 == 1) {
    const int k = ((tid >> 6) & 1) * 64 + (tid & 63);
#pragma unroll
    for (int j = 0; j < 128; ++j)
      wtm[j] = k < KC ? cd.WcatT[(size_t)(((tid >> 6) >> 1) * 128 + j) * KC + k] : 0.f;
  } else if (NXR > 0) {
#pragma unroll
    for (int kk = 0; kk < KSG; ++kk) {
      const int k = segr * KSG + kk;
      wt[kk] = (segr < NSEG && k < G4) ? cd.WcatT[(size_t)k * KC + jr] : 0.f;
    }
    // rows G4 .. G4+KSG of the dz image are read (times a zero weight) by the last segment
    for (int e = tid; e < KSG * R4; e += blockDim.x) dzb[G4 * R4 + e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float dh[SPT], dc[SPT];
#pragma unroll
  for (int e = 0; e < SPT; ++e) {
    const int s = min(s0 + grp * SPT + e, Bc - 1);
    dh[e] = dhT[(size_t)s * F + u];
    dc[e] = 0.f;
  }
  // the saved gates and cell states of step t - 1 are requested while step t is processed: the sweep
  // never waits for a global round trip (the cell state of t - 1 is also step t's c_prev)
  float pg[SPT][4], pc[SPT];
  auto pf_load = [&](int t) {
#pragma unroll
    for (int e = 0; e < SPT; ++e) {
      const int s = min(s0 + grp * SPT + e, Bc - 1);
      const size_t gb = ((size_t)s * T1 + t) * G4;
      pg[e][0] = gates[gb + u]; pg[e][1] = gates[gb + F + u];
      pg[e][2] = gates[gb + 2 * F + u]; pg[e][3] = gates[gb + 3 * F + u];
      pc[e] = t > 0 ? cs[((size_t)s * T1 + t - 1) * F + u] : 0.f;
    }
  };
  float ccur[SPT];
#pragma unroll
  for (int e = 0; e < SPT; ++e)
    ccur[e] = cs[((size_t)min(s0 + grp * SPT + e, Bc - 1) * T1 + T1 - 1) * F + u];
  pf_load(T1 - 1);
  for (int t = T1 - 1; t >= 0; --t) {
    float gq[SPT][4], cq[SPT];
#pragma unroll
    for (int e = 0; e < SPT; ++e) {
      gq[e][0] = pg[e][0]; gq[e][1] = pg[e][1]; gq[e][2] = pg[e][2]; gq[e][3] = pg[e][3];
      cq[e] = pc[e];
    }
    if (t > 0) pf_load(t - 1);
#pragma unroll
    for (int e = 0; e < SPT; ++e) {
      const int sb = grp * SPT + e;
      const int s = min(s0 + sb, Bc - 1);
      const size_t gb = ((size_t)s * T1 + t) * G4;
      const float ig = gq[e][0], fg = gq[e][1], gg = gq[e][2], og = gq[e][3];
      const float ct = ccur[e];
      const float cprev = cq[e];
      ccur[e] = cprev;
      const float tc = tanhf(ct);
      const float d_o = dh[e] * tc;
      dc[e] = dc[e] + dh[e] * og * (1.f - tc * tc);
      const float di = dc[e] * gg, df_ = dc[e] * cprev, dg = dc[e] * ig;
      const float zi = di * ig * (1.f - ig), zf = df_ * fg * (1.f - fg), zg = dg * (1.f - gg * gg),
                  zo = d_o * og * (1.f - og);
      dzf[(0 * F + u) * SB + sb] = zi;
      dzf[(1 * F + u) * SB + sb] = zf;
      dzf[(2 * F + u) * SB + sb] = zg;
      dzf[(3 * F + u) * SB + sb] = zo;
      if (s0 + sb < Bc && dz != nullptr) {
        float* d = dz + gb;
        d[u] = zi; d[F + u] = zf; d[2 * F + u] = zg; d[3 * F + u] = zo;
      }
      dc[e] = dc[e] * fg;
    }
    __syncthreads();
    // [dx ; dh_prev][k][sb] = sum_j WcatT[j][k] dz[j][sb]
    if constexpr (NXR > 0 && R4 == 1) {
      // 4 sequences: dz IS the broadcast A operand ([j][4 slots]), the transposed weights the B operand:
      // output column k = 64 (wave & 1) + lane, the 256 gate rows split over the wave pairs
      constexpr int JH = 128;                       // gate rows per wave pair
      const int ln = tid & 63, wv = tid >> 6;
      const float* dzh = dzf + (wv >> 1) * JH * 4;
      float ar[JH / 16];
#pragma unroll
      for (int r = 0; r < JH / 16; ++r) ar[r] = dzh[64 * r + ln];
      f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      rw_static_for<JH>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j & 1) rw_mfma<j>(d1, ar[j >> 4], wtm[j]);
        else rw_mfma<j>(d0, ar[j >> 4], wtm[j]);
      });
      const f32x4_t d = d0 + d1;
      const int k = (wv & 1) * 64 + ln;
      if (k < KC) part[(wv >> 1) * KC + k] = make_float4(d[0], d[1], d[2], d[3]);
      __syncthreads();
      for (int e = tid; e < KC; e += blockDim.x) {
        float4 sm = part[e];
        const float4 pp = part[KC + e];
        sm.x += pp.x; sm.y += pp.y; sm.z += pp.z; sm.w += pp.w;
        part[e] = sm;
      }
      __syncthreads();
    } else if constexpr (NXR > 0) {
      if (segr < NSEG) {
        float4 acc[R4];
#pragma unroll
        for (int q = 0; q < R4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* dseg = dzb + segr * KSG * R4;
#pragma unroll
        for (int kk = 0; kk < KSG; ++kk)
#pragma unroll
          for (int q = 0; q < R4; ++q) fma4(acc[q], wt[kk], dseg[kk * R4 + q]);
#pragma unroll
        for (int q = 0; q < R4; ++q) part[(segr * KC + jr) * R4 + q] = acc[q];
      }
      __syncthreads();
      for (int e = tid; e < KC * R4; e += blockDim.x) {
        float4 sm = part[e];
#pragma unroll
        for (int sg = 1; sg < NSEG; ++sg) {
          const float4 pp = part[sg * KC * R4 + e];
          sm.x += pp.x; sm.y += pp.y; sm.z += pp.z; sm.w += pp.w;
        }
        part[e] = sm;
      }
      __syncthreads();
    } else {
      dense_small<R4>(stage_w ? wlds : cd.WcatT, G4, K, dzb, part);
    }
    const float* pf = reinterpret_cast<const float*>(part);
#pragma unroll
    for (int e = 0; e < SPT; ++e) dh[e] = pf[(n + u) * SB + grp * SPT + e];
    if (dxseq != nullptr) {
      for (int e = tid; e < n * SB; e += blockDim.x) {
        const int sb = e / n, i = e - sb * n;
        if (s0 + sb < Bc) dxseq[((size_t)(s0 + sb) * T1 + t) * n + i] = pf[i * SB + sb];
      }
    }
    __syncthreads();
  }
}

// C[M][N] (+ colsum) partials: Cp[split][M][N] = sum over a row chunk of A[r][:M]^T B[r][:N].
// 64x64 tile per workgroup, 4x4 micro-tile per thread, 16 rows per LDS stage.
__global__ __launch_bounds__(GMPC_THREADS) void k_wgrad(int rows, int M, int N, const float* A,
                                                        int lda, const float* Bm, int ldb,
                                                        int rows_per_split, float* Cp,
                                                        float* colsum_p, int cs_rows) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x;
  const int tm = blockIdx.x * 64, tn = blockIdx.y * 64, sp = blockIdx.z;
  const int r0 = sp * rows_per_split, r1 = min(rows, r0 + rows_per_split);
  const int ty = tid / 16, tx = tid % 16;   // micro-tile rows ty*4.., cols tx*4..
  float acc[4][4] = {};
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_cs = (colsum_p != nullptr) && (blockIdx.x == 0);
  for (int rb = r0; rb < r1; rb += 16) {
    for (int e = tid; e < 16 * 64; e += blockDim.x) {
      const int rr = e / 64, cidx = e % 64;
      const int r = rb + rr;
      As[rr][cidx] = (r < r1 && tm + cidx < M) ? A[(size_t)r * lda + tm + cidx] : 0.f;
      Bs[rr][cidx] = (r < r1 && tn + cidx < N) ? Bm[(size_t)r * ldb + tn + cidx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[rr][ty * 4 + i]; bv[i] = Bs[rr][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
      if (do_cs && ty == 0 && rb + rr < cs_rows) {
#pragma unroll
        for (int j = 0; j < 4; ++j) csum[j] += bv[j];
      }
    }
    __syncthreads();
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const int mm = tm + ty * 4 + i, nn = tn + tx * 4 + j;
      if (mm < M && nn < N) Cp[((size_t)sp * M + mm) * N + nn] = acc[i][j];
    }
  if (do_cs && ty == 0)
    for (int j = 0; j < 4; ++j) {
      const int nn = tn + tx * 4 + j;
      if (nn < N) colsum_p[(size_t)sp * N + nn] = csum[j];
    }
}

// out[e] = sum_sp part[sp][e], deterministic: 64 elements per block, the splits are shared by 4
// thread groups, each keeping 8 independent partial sums (so 8 loads are in flight per thread);
// the fixed combination order makes the result run-to-run identical.
__global__ __launch_bounds__(256) void k_reduce_splits(int count, int nsplit, const float* part,
                                                       float* out) {
  __shared__ float sh[4][64];
  const int el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int q = (nsplit + 3) / 4;
  const int s0 = seg * q, s1 = min(nsplit, s0 + q);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < count) {
    int sp = s0;
    for (; sp + 8 <= s1; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += part[(size_t)(sp + u) * count + e];
    }
    for (; sp < s1; ++sp) a[0] += part[(size_t)sp * count + e];
  }
  sh[seg][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (seg == 0 && e < count) out[e] = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
}

// Single-block sum of `count` floats (fixed order: thread-strided partials, then tree in LDS).
__global__ __launch_bounds__(1024) void k_sum(int count, const float* v, float* out, int square) {
  __shared__ float sh[1024];
  float s = 0.f;
  for (int e = threadIdx.x; e < count; e += blockDim.x) {
    const float x = v[e];
    s += square ? x * x : x;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ void k_transpose(int R, int C, const float* in, float* out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * C) return;
  const int r = e / C, c = e - r * C;
  out[(size_t)c * R + r] = in[e];
}

// two-stage sum of squares of (grad * scale): partials per block, then k_sum
__global__ __launch_bounds__(GMPC_THREADS) void k_sqsum_part(long count, const float* g, float scale,
                                                             float* part) {
  __shared__ float sh[GMPC_THREADS];
  float s = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (long)gridDim.x * blockDim.x) {
    const float x = g[e] * scale;
    s = fmaf(x, x, s);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = GMPC_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// optax clip_by_global_norm + adam (gan/runner.py:58): g <- g*scale; if !(norm < max_norm)
// g <- g / norm * max_norm; m,v update; p += -lr * mhat / (sqrt(vhat) + eps)
// `sqpart`: the 256 partial sums of k_sqsum_part; every block adds them up itself -- the tree of k_sum over the same
// values, so the norm has the bits it had when a k_sum launch stood between the two kernels (one launch and one
// dependent kernel boundary less on the tail of every step)
__global__ __launch_bounds__(256) void k_adam(long count, float* p, const float* g, float* m, float* v, float scale,
                                              const float* sqpart, float max_norm, float lr, float b1, float b2,
                                              float omb1, float omb2, float eps, float bc1, float bc2) {
  __shared__ float sh[256];
  sh[threadIdx.x] = sqpart[threadIdx.x];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const float gn = sqrtf(sh[0]);
  float x = g[e] * scale;
  if (!(gn < max_norm)) x = x / gn * max_norm;
  const float mn = b1 * m[e] + omb1 * x;
  const float vn = b2 * v[e] + omb2 * x * x;
  m[e] = mn;
  v[e] = vn;
  const float mh = mn / bc1, vh = vn / bc2;
  p[e] = p[e] + (-lr * mh / (sqrtf(vh) + eps));
}

// Polyak blend (norm/cost_trainer.py:88-92): out = f * prev + (1 - f) * cur
__global__ void k_polyak(long count, const float* prev, const float* cur, float f, float omf,
                         float* out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) out[e] = f * prev[e] + omf * cur[e];
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// 32 x 32 tiles through LDS: both the read and the write are coalesced (large activations)
__global__ __launch_bounds__(256) void k_transpose_tiled(int R, int C, const float* in, float* out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8)
    if (r0 + r < R && c0 + tx < C) tile[r][tx] = in[(size_t)(r0 + r) * C + c0 + tx];
  __syncthreads();
  for (int cc = ty; cc < 32; cc += 8)
    if (c0 + cc < C && r0 + tx < R) out[(size_t)(c0 + cc) * R + r0 + tx] = tile[tx][cc];
}

void gmpc_launch_transpose(int R, int C, const float* in, float* out, hipStream_t s) {
  if ((long)R * C > (1L << 16)) {
    hipLaunchKernelGGL(k_transpose_tiled, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, R, C, in,
                       out);
    return;
  }
  const int cnt = R * C;
  hipLaunchKernelGGL(k_transpose, dim3((cnt + 255) / 256), dim3(256), 0, s, R, C, in, out);
}


// ---------------------------------------------------------------------------------------------
// Any lstm_features F <= 128 (the reference makes it a yaml integer, critic/nn.py:11, config/gan_hyperparameters.yaml:
// 60-65; its expert model uses 128): the kernels above and gmpc_critic_lstm.hip are built around 4 F = 256 = one
// workgroup.  Here the 4 F gate columns and the F cell units are strided over the 256 threads; weights come from L2
// (coalesced over the gate column / the input row), expf / tanhf activations.  4 sequences per workgroup, one float4
// per (row, 4 sequences) in LDS.  Same saves as k_lstm_fwd (activated gates, c_t, h_{t-1}) so that the weight-gradient
// GEMMs, the head and the wide-input path (cd.n == 0, Wcat = Wh, x Wx from xproj) are shared.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GMPC_THREADS) void k_lstm_fwd_g(int Bc, CriticDesc cd, const float* xseq, float* gates,
                                                             float* cs, float* hp, float* hT, const float* xproj) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  const int tid = threadIdx.x;
  const int n = cd.n, F = cd.F, T1 = cd.T1, G4 = 4 * F, K = n + F;
  float4* act = reinterpret_cast<float4*>(smem_g);          // [n + F]: x_t, then h_{t-1}
  float4* gbuf = act + K;                                   // [4 F] activated gates
  float4* cst = gbuf + G4;                                  // [F] cell state
  float* actf = reinterpret_cast<float*>(act);
  const int s0 = blockIdx.x * 4;
  int sq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) sq[c] = min(s0 + c, Bc - 1);
  for (int e = tid; e < F; e += GMPC_THREADS) {
    act[n + e] = make_float4(0.f, 0.f, 0.f, 0.f);
    cst[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int t = 0; t < T1; ++t) {
    for (int e = tid; e < n * 4; e += GMPC_THREADS) {
      const int i = e >> 2, c = e & 3;
      actf[e] = xseq[((size_t)sq[c] * T1 + t) * n + i];
    }
    __syncthreads();
    for (int e = tid; e < F * 4; e += GMPC_THREADS) {       // save h_{t-1}
      const int k = e >> 2, c = e & 3;
      if (s0 + c < Bc) hp[((size_t)(s0 + c) * T1 + t) * F + k] = actf[(n + k) * 4 + c];
    }
    for (int j = tid; j < G4; j += GMPC_THREADS) {
      const float bj = cd.b[j];
      float4 acc[1] = {make_float4(bj, bj, bj, bj)};
      if (xproj != nullptr) {
        acc[0].x += xproj[((size_t)sq[0] * T1 + t) * G4 + j];
        acc[0].y += xproj[((size_t)sq[1] * T1 + t) * G4 + j];
        acc[0].z += xproj[((size_t)sq[2] * T1 + t) * G4 + j];
        acc[0].w += xproj[((size_t)sq[3] * T1 + t) * G4 + j];
      }
      dense_rows<1>(cd.Wcat, K, G4, j, act, acc);
      float4 v = acc[0];
      if (j / F == 2) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
      else { v.x = sigmoidf_(v.x); v.y = sigmoidf_(v.y); v.z = sigmoidf_(v.z); v.w = sigmoidf_(v.w); }
      gbuf[j] = v;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (s0 + c < Bc) gates[((size_t)(s0 + c) * T1 + t) * G4 + j] = f4get(v, c);
    }
    __syncthreads();
    for (int u = tid; u < F; u += GMPC_THREADS) {
      const float4 ig = gbuf[u], fg = gbuf[F + u], gg = gbuf[2 * F + u], og = gbuf[3 * F + u];
      float4 c = cst[u], h;
      c.x = fg.x * c.x + ig.x * gg.x; c.y = fg.y * c.y + ig.y * gg.y;
      c.z = fg.z * c.z + ig.z * gg.z; c.w = fg.w * c.w + ig.w * gg.w;
      h.x = og.x * tanhf(c.x); h.y = og.y * tanhf(c.y); h.z = og.z * tanhf(c.z); h.w = og.w * tanhf(c.w);
      cst[u] = c;
      act[n + u] = h;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (s0 + q < Bc) {
          cs[((size_t)(s0 + q) * T1 + t) * F + u] = f4get(c, q);
          if (t == T1 - 1) hT[(size_t)(s0 + q) * F + u] = f4get(h, q);
        }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(GMPC_THREADS) void k_lstm_bwd_g(int Bc, CriticDesc cd, const float* gates, const float* cs,
                                                             const float* dhT, float* dz, float* dxseq) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  const int tid = threadIdx.x;
  const int n = cd.n, F = cd.F, T1 = cd.T1, G4 = 4 * F, K = n + F;
  float4* dzb = reinterpret_cast<float4*>(smem_g);          // [4 F]
  float4* dhb = dzb + G4;                                   // [F] dh_t
  float4* dcb = dhb + F;                                    // [F] dc
  const int s0 = blockIdx.x * 4;
  int sq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) sq[c] = min(s0 + c, Bc - 1);
  for (int u = tid; u < F; u += GMPC_THREADS) {
    dhb[u] = make_float4(dhT[(size_t)sq[0] * F + u], dhT[(size_t)sq[1] * F + u], dhT[(size_t)sq[2] * F + u],
                         dhT[(size_t)sq[3] * F + u]);
    dcb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (int t = T1 - 1; t >= 0; --t) {
    for (int u = tid; u < F; u += GMPC_THREADS) {
      float zi[4], zf[4], zg[4], zo[4], dcn[4];
      const float4 dh4 = dhb[u], dc4 = dcb[u];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t gb = ((size_t)sq[c] * T1 + t) * G4;
        const float ig = gates[gb + u], fg = gates[gb + F + u], gg = gates[gb + 2 * F + u], og = gates[gb + 3 * F + u];
        const float ct = cs[((size_t)sq[c] * T1 + t) * F + u];
        const float cprev = t > 0 ? cs[((size_t)sq[c] * T1 + t - 1) * F + u] : 0.f;
        const float tc = tanhf(ct);
        const float dh = f4get(dh4, c);
        const float d_o = dh * tc;
        const float dc = f4get(dc4, c) + dh * og * (1.f - tc * tc);
        const float di = dc * gg, df_ = dc * cprev, dg = dc * ig;
        zi[c] = di * ig * (1.f - ig); zf[c] = df_ * fg * (1.f - fg); zg[c] = dg * (1.f - gg * gg);
        zo[c] = d_o * og * (1.f - og);
        dcn[c] = dc * fg;
        if (s0 + c < Bc && dz != nullptr) {
          float* d = dz + gb;
          d[u] = zi[c]; d[F + u] = zf[c]; d[2 * F + u] = zg[c]; d[3 * F + u] = zo[c];
        }
      }
      dzb[u] = make_float4(zi[0], zi[1], zi[2], zi[3]);
      dzb[F + u] = make_float4(zf[0], zf[1], zf[2], zf[3]);
      dzb[2 * F + u] = make_float4(zg[0], zg[1], zg[2], zg[3]);
      dzb[3 * F + u] = make_float4(zo[0], zo[1], zo[2], zo[3]);
      dcb[u] = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
    }
    __syncthreads();
    // [dx ; dh_{t-1}][k] = sum_j WcatT[j][k] dz[j]   (WcatT: [4 F][n + F], coalesced over k)
    for (int k = tid; k < K; k += GMPC_THREADS) {
      float4 acc[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
      dense_rows<1>(cd.WcatT, G4, K, k, dzb, acc);
      if (k < n) {
        if (dxseq != nullptr) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (s0 + c < Bc) dxseq[((size_t)(s0 + c) * T1 + t) * n + k] = f4get(acc[0], c);
        }
      } else {
        dhb[k - n] = acc[0];
      }
    }
    __syncthreads();
  }
}

#define GMPC_CR4 1   // 4 sequences per workgroup

void gmpc_launch_lstm_fwd(int Bc, const CriticDesc& cd, const float* xseq, float* gates, float* cs,
                          float* hp, float* hT, const float* xproj, hipStream_t s) {
  constexpr int R4 = GMPC_CR4;
  const int grid = (Bc + 4 * R4 - 1) / (4 * R4);
  if (cd.F != 64) {       // the strided form for other feature counts
    hipLaunchKernelGGL(k_lstm_fwd_g, dim3((Bc + 3) / 4), dim3(GMPC_THREADS),
                       ((size_t)cd.n + 6 * (size_t)cd.F) * sizeof(float4), s, Bc, cd, xseq, gates, cs, hp, hT, xproj);
    return;
  }
  size_t lds = ((size_t)(cd.n + cd.F) + 4 * cd.F) * R4 * sizeof(float4);
  const size_t wbytes = (size_t)(cd.n + cd.F) * 4 * cd.F * sizeof(float);
  // measured on MI355X: staging [Wx;Wh] in LDS does not pay (0.25 -> 0.28 ms): the step is bound by
  // instruction issue (transcendentals + FMAs at one wave per SIMD), not by the L2 weight stream
  const int stage_w = 0 * (lds + wbytes <= 150 * 1024);
  if (stage_w) lds += wbytes;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lstm_fwd<R4, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (cd.n == 17 && cd.F == 64)
    hipLaunchKernelGGL((k_lstm_fwd<R4, 17>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, xseq, gates,
                       cs, hp, hT, 0, nullptr);
  else if (cd.n == 3 && cd.F == 64)
    hipLaunchKernelGGL((k_lstm_fwd<R4, 3>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, xseq, gates,
                       cs, hp, hT, 0, nullptr);
  else
    hipLaunchKernelGGL((k_lstm_fwd<R4, 0>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, xseq, gates,
                       cs, hp, hT, stage_w, xproj);
}

void gmpc_launch_lstm_bwd(int Bc, const CriticDesc& cd, const float* gates, const float* cs,
                          const float* dhT, float* dz, float* dxseq, hipStream_t s) {
  constexpr int R4 = GMPC_CR4;
  const int grid = (Bc + 4 * R4 - 1) / (4 * R4);
  if (cd.F != 64) {
    hipLaunchKernelGGL(k_lstm_bwd_g, dim3((Bc + 3) / 4), dim3(GMPC_THREADS), (size_t)6 * cd.F * sizeof(float4), s, Bc,
                       cd, gates, cs, dhT, dz, dxseq);
    return;
  }
  size_t lds = 2 * (size_t)(GMPC_THREADS + 128) * R4 * sizeof(float4);
  const size_t wbytes = (size_t)(cd.n + cd.F) * 4 * cd.F * sizeof(float);
  const int stage_w = 0 * (lds + wbytes <= 150 * 1024);   // no gain measured (see k_lstm_fwd)
  if (stage_w) lds += wbytes;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lstm_bwd<R4, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (cd.n == 17 && cd.F == 64)
    hipLaunchKernelGGL((k_lstm_bwd<R4, 17>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, gates, cs,
                       dhT, dz, dxseq, 0);
  else if (cd.n == 3 && cd.F == 64)
    hipLaunchKernelGGL((k_lstm_bwd<R4, 3>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, gates, cs,
                       dhT, dz, dxseq, 0);
  else
    hipLaunchKernelGGL((k_lstm_bwd<R4, 0>), dim3(grid), dim3(GMPC_THREADS), lds, s, Bc, cd, gates, cs,
                       dhT, dz, dxseq, stage_w);
}

// C[M][N] = sum_r A[r][:M]^T B[r][:N]; colsum[N] = sum_{r < cs_rows} B[r][:N] (optional).
// part must hold nsplit*(M*N + N) floats.
bool gmpc_launch_wgrad_mfma(int rows, int M, int N, const float* A, int lda, const float* Bm, int ldb,
                            float* C, float* colsum, int cs_rows, float* part, long part_floats,
                            hipStream_t s);
__global__ void k_colsum(int cs_rows, int N, const float* Bm, int ldb, int rows_per_chunk, float* part);

// part holds part_floats floats (at least max_split*(M*N + N)); mfma_ok: B has >= 8 zero pad rows
void gmpc_launch_wgrad(int rows, int M, int N, const float* A, int lda, const float* Bm, int ldb,
                       float* C, float* colsum, int cs_rows, float* part, int max_split,
                       hipStream_t s, long part_floats, bool mfma_ok) {
  if (mfma_ok && gmpc_launch_wgrad_mfma(rows, M, N, A, lda, Bm, ldb, C, colsum, cs_rows, part,
                                        part_floats, s))
    return;
  // narrow N (e.g. the critic head's last layer, N = 1): compute C^T = sum_r B_r^T A_r instead;
  // C^T (N x M) has the same memory image as C when N == 1, otherwise it is transposed afterwards
  if (mfma_ok && N == 1 && M % 32 == 0 &&
      gmpc_launch_wgrad_mfma(rows, 1, M, Bm, ldb, A, lda, C, nullptr, 0, part, part_floats, s)) {
    if (colsum) {
      const float* Bc_ = Bm;
      int cchunks = (cs_rows + 63) / 64;
      if (cchunks > 2048) cchunks = 2048;
      const int crpc = (cs_rows + cchunks - 1) / cchunks;
      cchunks = (cs_rows + crpc - 1) / crpc;
      hipLaunchKernelGGL(k_colsum, dim3(cchunks), dim3(GMPC_THREADS), 0, s, cs_rows, N, Bc_, ldb, crpc,
                         part);
      hipLaunchKernelGGL(k_reduce_splits, dim3((N + 63) / 64), dim3(256), 0, s, N, cchunks, part,
                         colsum);
    }
    return;
  }
  int nsplit = (rows + 511) / 512;
  if (nsplit > max_split) nsplit = max_split;
  // the partial sums must fit the scratch buffer (wide layers: n + m = 1088 inputs at C5)
  const long per_split = (long)M * N + (colsum ? N : 0);
  if (part_floats > 0 && (long)nsplit * per_split > part_floats) nsplit = (int)(part_floats / per_split);
  if (nsplit < 1) nsplit = 1;
  int rps = (rows + nsplit - 1) / nsplit;
  rps = (rps + 15) / 16 * 16;
  nsplit = (rows + rps - 1) / rps;
  float* cpart = part;
  float* cspart = colsum ? part + (size_t)nsplit * M * N : nullptr;
  hipLaunchKernelGGL(k_wgrad, dim3((M + 63) / 64, (N + 63) / 64, nsplit), dim3(GMPC_THREADS), 0, s,
                     rows, M, N, A, lda, Bm, ldb, rps, cpart, cspart, cs_rows);
  hipLaunchKernelGGL(k_reduce_splits, dim3((M * N + 63) / 64), dim3(256), 0, s, M * N, nsplit, cpart,
                     C);
  if (colsum)
    hipLaunchKernelGGL(k_reduce_splits, dim3((N + 63) / 64), dim3(256), 0, s, N, nsplit, cspart,
                       colsum);
}

void gmpc_launch_sum(int count, const float* v, float* out, int square, hipStream_t s) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, count, v, out, square);
}

void gmpc_launch_adam(long count, float* p, const float* g, float* m, float* v, float scale,
                      int step, double lr, double max_norm, double b1, double b2, double eps,
                      float* scratch /* >= 257 floats */, hipStream_t s) {
  const int nb = 256;
  hipLaunchKernelGGL(k_sqsum_part, dim3(nb), dim3(GMPC_THREADS), 0, s, count, g, scale, scratch + 1);
  const float bc1 = (float)(1.0 - pow(b1, (double)step)), bc2 = (float)(1.0 - pow(b2, (double)step));
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, count, p, g, m, v,
                     scale, scratch + 1, (float)max_norm, (float)lr, (float)b1, (float)b2,
                     (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, bc1, bc2);
}

void gmpc_launch_polyak(long count, const float* prev, const float* cur, double f, float* out,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_polyak, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, count, prev, cur,
                     (float)f, (float)(1.0 - f), out);
}

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM on the matrix cores: C[M][N] = sum_r A[r][:M]^T B[r][:N] with both operands
// read straight from global memory in their natural row-major layout (the MFMA A operand of k-step
// r is row r of A, the B operand row r of B: both coalesced).  One wavefront owns a 32 x 32*NTW
// strip of C over one chunk of rows; partial strips are summed in chunk order (deterministic).
// Requires N % (32*NTW) == 0 and B padded with >= 8 zero rows; A is clamped (it may be a caller's
// buffer).  v_mfma_f32_32x32x2_f32 = k-ordered exact fp32 fmaf chain.
// ---------------------------------------------------------------------------------------------
template <int NTW>
__global__ __launch_bounds__(GMPC_THREADS) void k_wgrad_mfma(int rows, int M, int N, const float* A,
                                                             int lda, const float* Bm, int ldb,
                                                             int rows_per_chunk, float* Cp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int mstrips = (M + 31) >> 5, ngroups = N / (32 * NTW);
  const int nchunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
  const int total = mstrips * ngroups * nchunks;
  const int item = blockIdx.x * (GMPC_THREADS / 64) + wave;
  if (item >= total) return;
  const int chunk = item / (mstrips * ngroups);
  const int rem = item - chunk * mstrips * ngroups;
  const int mi = rem / ngroups, ng = rem - mi * ngroups;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  const int Kp = (r1 - r0 + 1) & ~1;
  const int acol = mi * 32 + l31;
  const bool aok = acol < M;
  const float* ap = A + (aok ? acol : M - 1);
  auto afn = [&](int k0) -> float {
    const int r = r0 + k0 + half;
    const float v = ap[(size_t)min(r, rows - 1) * lda];
    return (aok && r < r1) ? v : 0.f;
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[nt][rg] = 0.f;
  const float* bp0 = Bm + (size_t)(r0 + half) * ldb + ng * 32 * NTW + l31;
  gemm_tile<NTW>(bp0, ldb, Kp, afn, acc);
  float* cp = Cp + (size_t)chunk * M * N;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = ng * 32 * NTW + nt * 32 + l31;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < M) cp[(size_t)row * N + col] = acc[nt][rg];
    }
  }
}

// column sums of the first cs_rows rows of B: partial[chunk][j]
__global__ __launch_bounds__(GMPC_THREADS) void k_colsum(int cs_rows, int N, const float* Bm, int ldb,
                                                         int rows_per_chunk, float* part) {
  const int chunk = blockIdx.x;
  const int r0 = chunk * rows_per_chunk, r1 = min(cs_rows, r0 + rows_per_chunk);
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += Bm[(size_t)(r + u) * ldb + j];
    }
    for (; r < r1; ++r) a[0] += Bm[(size_t)r * ldb + j];
    part[(size_t)chunk * N + j] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
}

// ---------------------------------------------------------------------------------------------
// Batched weight gradients: the five or six row-sum GEMMs of one optimiser step (LSTM input and
// recurrent kernels, head layers) differ only in their operands, and each is far too small to fill
// the chip (a few hundred wave-tiles): issued one by one they cost a launch, a tail and a reduction
// launch each.  Here ONE launch covers the wave-tiles of all problems plus their bias column sums
// (the trailing workgroups), and ONE launch reduces all the partial sums, in the same fixed chunk
// order as before (deterministic).  Problems need N % 256 == 0.
// ---------------------------------------------------------------------------------------------
// C(32 x 256) += A^T B over Kp rows for one wave: row r of A / B is the MFMA A / B operand of k-step r / 2.
// B goes through 16-byte loads: lane l31 holds columns 4 l31 .. 4 l31 + 3 of each 128-column half, so tile j
// of the accumulator is columns 128 (j >> 2) + 4 l31 + (j & 3) (the store undoes the permutation); the
// operands of D k-steps are in flight.  Measured (LSTM + head problems of the headline step, batch kernel +
// reductions): 8 tiles / ring 3 / dword loads 0.227 ms; 8 tiles, ring 8 or 12, one wave per SIMD 0.232;
// 8 tiles, ring 4, two waves 0.183; 4 tiles, ring 4, four waves 0.172 -- the rows stream from HBM and it
// is occupancy, not ring depth, that hides their latency.
template <int D, int NTW, typename AF>
__device__ __forceinline__ void wgrad_tile_x4(const float* __restrict__ bp0, int ldb, int Kp, AF afn,
                                              f32x16 (&acc)[NTW]) {
  static_assert(NTW == 4 || NTW == 8, "one or two 16-byte loads per lane and k-step");
  float4 b[D][NTW / 4];
  float a[D];
  const int nks = Kp >> 1;
  auto load = [&](int slot, int ks) {
    const float4* bp = reinterpret_cast<const float4*>(bp0 + (size_t)2 * ks * ldb);
    b[slot][0] = bp[0];
    if (NTW > 4) b[slot][NTW / 4 - 1] = bp[32];
    a[slot] = afn(2 * ks);
  };
#pragma unroll
  for (int j = 0; j < D - 1; ++j)
    if (j < nks) load(j, j);
  for (int k0 = 0; k0 < nks; k0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int ks = k0 + u;
      if (ks + D - 1 < nks) load((u + D - 1) % D, ks + D - 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ks < nks) {
        const float av = a[u];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][0].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][0].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][0].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][0].w, acc[3], 0, 0, 0);
        if (NTW > 4) {
          acc[NTW - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][NTW / 4 - 1].x, acc[NTW - 4], 0, 0, 0);
          acc[NTW - 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][NTW / 4 - 1].y, acc[NTW - 3], 0, 0, 0);
          acc[NTW - 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][NTW / 4 - 1].z, acc[NTW - 2], 0, 0, 0);
          acc[NTW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][NTW / 4 - 1].w, acc[NTW - 1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

#ifndef GMPC_WG_RING
#define GMPC_WG_RING 4     // k-steps of operands in flight per wave
#endif
#ifndef GMPC_WG_OCC
#define GMPC_WG_OCC 4      // waves per SIMD: the rows stream from HBM, occupancy hides what the ring does not
#endif
#ifndef GMPC_WG_NTW
#define GMPC_WG_NTW 4      // 32-column tiles per wave item (64 accumulator registers)
#endif
__global__ __launch_bounds__(GMPC_THREADS, GMPC_WG_OCC) void k_wgrad_batch(WgBatch bt, float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= bt.gemm_blocks) {
    // bias column sums: one workgroup per (problem, row chunk)
    const int cb = blockIdx.x - bt.gemm_blocks;
    int pi = -1;
    for (int i = 0; i < bt.np; ++i)
      if (cb >= bt.p[i].cs_block0 && cb < bt.p[i].cs_block0 + bt.p[i].cchunks) pi = i;
    if (pi < 0) return;
    const WgProb& q = bt.p[pi];
    const int chunk = cb - q.cs_block0;
    const int r0 = chunk * q.crpc, r1 = min(q.cs_rows, r0 + q.crpc);
    float* out = part + q.cs_part_off + (size_t)chunk * q.N;
    for (int j = threadIdx.x; j < q.N; j += blockDim.x) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int r = r0;
      for (; r + 8 <= r1; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += q.B[(size_t)(r + u) * q.ldb + j];
      }
      for (; r < r1; ++r) a[0] += q.B[(size_t)r * q.ldb + j];
      out[j] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    return;
  }
  constexpr int NTW = GMPC_WG_NTW;
  const int item = blockIdx.x * (GMPC_THREADS / 64) + wave;
  int pi = 0;
  for (int i = 1; i < bt.np; ++i)
    if (item >= bt.p[i].item0) pi = i;
  const WgProb& q = bt.p[pi];
  const int local = item - q.item0;
  if (local >= q.mstrips * q.ngroups * q.nchunks) return;
  const int half = lane >> 5, l31 = lane & 31;
  const int chunk = local / (q.mstrips * q.ngroups);
  const int rem = local - chunk * q.mstrips * q.ngroups;
  const int mi = rem / q.ngroups, ng = rem - mi * q.ngroups;
  const int rows = q.rows, M = q.M, N = q.N;
  const int r0 = chunk * q.rpc;
  const int r1 = min(rows, r0 + q.rpc);
  const int Kp = (r1 - r0 + 1) & ~1;
  const int acol = mi * 32 + l31;
  const bool aok = acol < M;
  const float* ap = q.A + (aok ? acol : M - 1);
  const int lda = q.lda;
  auto afn = [&](int k0) -> float {
    const int r = r0 + k0 + half;
    const float v = ap[(size_t)min(r, rows - 1) * lda];
    return (aok && r < r1) ? v : 0.f;
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[nt][rg] = 0.f;
  // (rows past the chunk meet a zero A operand; the B reads stay inside the array: the last chunk's
  // odd tail row is clamped by the row pointer below)
  const float* bp0 = q.B + (size_t)(r0 + half) * q.ldb + ng * 32 * NTW + 4 * l31;
  if (r0 + Kp > rows) {      // wave-uniform
    // the chunk's last k-step would read row `rows`: run it from a clamped pointer
    wgrad_tile_x4<GMPC_WG_RING, NTW>(bp0, q.ldb, Kp - 2, afn, acc);
    const float* bl = q.B + (size_t)min(r0 + Kp - 2 + half, rows - 1) * q.ldb + ng * 32 * NTW + 4 * l31;
    auto afl = [&](int k0) -> float { return afn(k0 + Kp - 2); };
    wgrad_tile_x4<1, NTW>(bl, q.ldb, 2, afl, acc);
  } else {
    wgrad_tile_x4<GMPC_WG_RING, NTW>(bp0, q.ldb, Kp, afn, acc);
  }
  // tiles 4 j .. 4 j + 3 of a lane are four consecutive columns: one 16-byte store per accumulator row (N % 128 == 0
  // and the partial buffer's slices are multiples of 4 floats, so the address is aligned)
  float* cp = part + q.part_off + (size_t)chunk * M * N;
#pragma unroll
  for (int j = 0; j < NTW / 4; ++j) {
    const int col = ng * 32 * NTW + j * 128 + 4 * l31;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int row = mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
      if (row < M)
        *reinterpret_cast<float4*>(cp + (size_t)row * N + col) =
            make_float4(acc[4 * j][rg], acc[4 * j + 1][rg], acc[4 * j + 2][rg], acc[4 * j + 3][rg]);
    }
  }
}

// all reductions of a batch: blocks [red_block0, ...) of problem i sum its nchunks partial C's,
// blocks [cs_red_block0, ...) its column-sum partials; same arithmetic as k_reduce_splits
__global__ __launch_bounds__(256) void k_reduce_batch(WgBatch bt, const float* part) {
  __shared__ float sh[4][64];
  int pi = 0, kind = 0;
  for (int i = 0; i < bt.np; ++i) {
    const int b_ = (int)blockIdx.x;
    if (b_ >= bt.p[i].red_block0 && b_ < bt.p[i].cs_red_block0) { pi = i; kind = 0; }
    if (bt.p[i].colsum != nullptr && b_ >= bt.p[i].cs_red_block0 &&
        b_ < bt.p[i].cs_red_block0 + (bt.p[i].N + 63) / 64) { pi = i; kind = 1; }
  }
  const WgProb& q = bt.p[pi];
  const int count = kind == 0 ? q.M * q.N : q.N;
  const int nsplit = kind == 0 ? q.nchunks : q.cchunks;
  const float* src = part + (kind == 0 ? q.part_off : q.cs_part_off);
  float* out = kind == 0 ? q.C : q.colsum;
  const int blk = blockIdx.x - (kind == 0 ? q.red_block0 : q.cs_red_block0);
  const int el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = blk * 64 + el;
  const int qn = (nsplit + 3) / 4;
  const int s0 = seg * qn, s1 = min(nsplit, s0 + qn);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < count) {
    int sp = s0;
    for (; sp + 8 <= s1; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(sp + u) * count + e];
    }
    for (; sp < s1; ++sp) a[0] += src[(size_t)sp * count + e];
  }
  sh[seg][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (seg == 0 && e < count) out[e] = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
}

// colsum[N] = sum_{r < rows} B[r][:N] on its own (two launches; `part` holds the chunk sums)
void gmpc_launch_colsum(int rows, int N, const float* Bm, int ldb, float* colsum, float* part, hipStream_t s) {
  int cchunks = (rows + 63) / 64;
  if (cchunks > 2048) cchunks = 2048;
  if (cchunks < 1) cchunks = 1;
  const int crpc = (rows + cchunks - 1) / cchunks;
  cchunks = (rows + crpc - 1) / crpc;
  hipLaunchKernelGGL(k_colsum, dim3(cchunks), dim3(GMPC_THREADS), 0, s, rows, N, Bm, ldb, crpc, part);
  hipLaunchKernelGGL(k_reduce_splits, dim3((N + 63) / 64), dim3(256), 0, s, N, cchunks, part, colsum);
}

// returns false when a problem does not qualify (the caller then issues them one by one)
bool gmpc_launch_wgrad_batch(WgProb* probs, int np, float* part, long part_floats, hipStream_t s) {
  if (np < 1 || np > GMPC_WG_MAX) return false;
  WgBatch bt;
  bt.np = np;
  for (int i = 0; i < np; ++i) {
    WgProb& q = probs[i];
    if (q.M == 0 && q.colsum != nullptr) {      // column sums only (no GEMM part)
      q.mstrips = 0;
      q.ngroups = 0;
      continue;
    }
    if (q.N % (32 * GMPC_WG_NTW) != 0 || q.rows < 64 || q.ldb % 4 != 0 || (reinterpret_cast<uintptr_t>(q.B) & 15) != 0)
      return false;
    q.mstrips = (q.M + 31) / 32;
    q.ngroups = q.N / (32 * GMPC_WG_NTW);
  }
  // one chunk length (rows per wave-tile) for all problems, so that every wave does the same amount
  // of work: the shortest one whose partial sums fit the buffer and that needs <= 4096 wave-tiles
  static const int rpcs[] = {64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096, 8192, 1 << 30};
  static const long max_items = getenv("GMPC_WG_ITEMS") ? atol(getenv("GMPC_WG_ITEMS")) : 4096;
  int rpc = 0;
  for (int cand : rpcs) {
    long need = 0, items = 0;
    for (int i = 0; i < np; ++i) {
      const WgProb& q = probs[i];
      const long nch = (q.rows + cand - 1) / cand;
      need += nch * q.M * q.N + (q.colsum ? 1024L * q.N : 0);
      items += nch * q.mstrips * q.ngroups;
    }
    if (need <= part_floats && items <= max_items) { rpc = cand; break; }
  }
  if (rpc == 0) return false;
  int item = 0, cs_blocks = 0, red_blocks = 0;
  long off = 0;
  for (int i = 0; i < np; ++i) {
    WgProb& q = probs[i];
    q.rpc = rpc;
    q.nchunks = (q.rows + rpc - 1) / rpc;
    q.item0 = item;
    item += q.mstrips * q.ngroups * q.nchunks;
    q.part_off = off;
    off += (long)q.nchunks * q.M * q.N;
    if (q.colsum != nullptr) {
      int cchunks = (q.cs_rows + 63) / 64;
      if (cchunks > 1024) cchunks = 1024;
      if (cchunks < 1) cchunks = 1;
      q.crpc = (q.cs_rows + cchunks - 1) / cchunks;
      q.cchunks = (q.cs_rows + q.crpc - 1) / q.crpc;
      q.cs_block0 = cs_blocks;
      cs_blocks += q.cchunks;
      q.cs_part_off = off;
      off += (long)q.cchunks * q.N;
    } else {
      q.cchunks = 0; q.crpc = 0; q.cs_block0 = cs_blocks; q.cs_part_off = off;
    }
    off = (off + 3) & ~3L;               // every slice starts on a 16-byte boundary (float4 stores of the partials)
    q.red_block0 = red_blocks;
    red_blocks += (q.M * q.N + 63) / 64;
    q.cs_red_block0 = red_blocks;
    if (q.colsum != nullptr) red_blocks += (q.N + 63) / 64;
    bt.p[i] = q;
  }
  if (off > part_floats) return false;
  bt.gemm_blocks = (item + 3) / 4;
  hipLaunchKernelGGL(k_wgrad_batch, dim3(bt.gemm_blocks + cs_blocks), dim3(GMPC_THREADS), 0, s, bt, part);
  hipLaunchKernelGGL(k_reduce_batch, dim3(red_blocks), dim3(256), 0, s, bt, part);
  return true;
}

// MFMA path of gmpc_launch_wgrad; returns false when the shape does not qualify.
bool gmpc_launch_wgrad_mfma(int rows, int M, int N, const float* A, int lda, const float* Bm, int ldb,
                            float* C, float* colsum, int cs_rows, float* part, long part_floats,
                            hipStream_t s) {
  if (N % 32 != 0 || rows < 64) return false;
  const int nt_all = N / 32;
  const int ntw = (nt_all % 8 == 0) ? 8 : (nt_all % 4 == 0) ? 4 : (nt_all % 2 == 0) ? 2 : 1;
  const int mstrips = (M + 31) / 32, ngroups = nt_all / ntw;
  int nchunks = 1024 / (mstrips * ngroups);
  if (nchunks < 1) nchunks = 1;
  int rpc = (rows + nchunks - 1) / nchunks;
  if (rpc < 64) rpc = 64;
  rpc = (rpc + 1) & ~1;
  nchunks = (rows + rpc - 1) / rpc;
  while ((long)nchunks * M * N > part_floats && rpc < rows) {
    rpc *= 2;
    nchunks = (rows + rpc - 1) / rpc;
  }
  if ((long)nchunks * M * N > part_floats) return false;
  const int total = mstrips * ngroups * nchunks;
  const dim3 grid((total + 3) / 4), blk(GMPC_THREADS);
  switch (ntw) {
    case 8: hipLaunchKernelGGL(k_wgrad_mfma<8>, grid, blk, 0, s, rows, M, N, A, lda, Bm, ldb, rpc, part); break;
    case 4: hipLaunchKernelGGL(k_wgrad_mfma<4>, grid, blk, 0, s, rows, M, N, A, lda, Bm, ldb, rpc, part); break;
    case 2: hipLaunchKernelGGL(k_wgrad_mfma<2>, grid, blk, 0, s, rows, M, N, A, lda, Bm, ldb, rpc, part); break;
    default: hipLaunchKernelGGL(k_wgrad_mfma<1>, grid, blk, 0, s, rows, M, N, A, lda, Bm, ldb, rpc, part); break;
  }
  hipLaunchKernelGGL(k_reduce_splits, dim3((M * N + 63) / 64), dim3(256), 0, s, M * N, nchunks, part, C);
  if (colsum) {
    int cchunks = (cs_rows + 63) / 64;
    if (cchunks > 2048) cchunks = 2048;
    if (cchunks < 1) cchunks = 1;
    const int crpc = (cs_rows + cchunks - 1) / cchunks;
    cchunks = (cs_rows + crpc - 1) / crpc;
    hipLaunchKernelGGL(k_colsum, dim3(cchunks), dim3(GMPC_THREADS), 0, s, cs_rows, N, Bm, ldb, crpc, part);
    hipLaunchKernelGGL(k_reduce_splits, dim3((N + 63) / 64), dim3(256), 0, s, N, cchunks, part, colsum);
  }
  return true;
}
