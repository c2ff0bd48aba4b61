// Layer routines shared by the trajectory kernels (gmpc_traj.hip: general VALU form, line search
// bookkeeping; gmpc_traj_rw.hip: register-weight MFMA form).  4 trajectories ("slots") per workgroup,
// activations in LDS as float4 (one component per slot).
#pragma once
#include "gmpc_device.h"

#define GMPC_TRAJ_THREADS_ 512   // workgroup size of the general k_traj

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

