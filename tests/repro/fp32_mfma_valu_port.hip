// Two waves on one SIMD of gfx950: what the second wave can issue while the first streams fp32 MFMAs (round 4).
//
//   hipcc --offload-arch=gfx950 -O2 tests/repro/fp32_mfma_valu_port.hip -o /tmp/port && /tmp/port
//
// One workgroup of 512 threads: waves i and i + 4 land on the same SIMD (checked with s_getreg HW_ID: SIMD ids
// 0 2 1 3 0 2 1 3).  Waves 0-3 ("A") issue v_mfma_f32_16x16x4_f32 back to back on four independent accumulators;
// waves 4-7 ("B") run one of: a dependent v_fma chain, eight independent v_fma chains, a dependent chain of LDS round
// trips, or an MFMA -> LDS write -> LDS read chain -- each alone, beside A, and beside A at s_setprio 3.
// Measured (cycles for 1000 iterations; profiles/r04_fp32_mfma_valu_port.txt):
//     A alone                        136 k   (4000 MFMAs: 34 cycles each on that box's clock)
//     A with 4 dependent v_fma behind every MFMA, one wave: 232 k (58 per group = 34 + 4 x 6: nothing overlaps)
//     B dependent fma chain alone    116 k   beside A: 252 k (= 136 k + 116 k: no progress while A runs)   prio 3: 242 k
//     B independent fmas alone        72 k   beside A: 208 k (A 136 k)                                    prio 3: 195 k (A 147 k)
//     B LDS chain alone              284 k   beside A: 418 k                                              prio 3: 361 k
//     B MFMA + LDS chain alone        60 k   beside A: 160 k (A 160 k)                                    prio 3:  64 k (A 166 k)
// Reading: the fp32 MFMAs and the vector ALU instructions go through the same port (AMD quotes the same 157.3
// TFLOP/s for fp32 matrix and fp32 vector on MI355X), a wave that always has an MFMA ready keeps the other wave's
// vector instructions out almost entirely, and s_setprio helps the other wave's MFMA and LDS instructions but gives its
// plain vector instructions about one slot per MFMA of the first wave.  For the kernels of this library: every vector
// instruction inside an fp32-MFMA loop is paid for in matrix-pipe time, and a second wave per SIMD hides LDS / memory
// latency only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { A_MFMA = 1, B_CHAIN = 2, B_LDS = 4, B_MFMA_LDS = 8, B_PRIO = 16, A_MFMA_VALU = 32, B_INDEP = 64 };

__global__ __launch_bounds__(512) void k(unsigned long long* out, int mode, int iters, float* sink) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 512] = 1.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if (wave < 4) {
    if (mode & A_MFMA_VALU) {
      f32x4 d0 = {0, 0, 0, 0}, d1 = d0;
      float a = lane, b = lane * 0.5f, x = lane, y = 1.0001f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
          x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
          x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.5f);
        }
      }
      r = d0[0] + d1[1] + x;
    } else if (mode & A_MFMA) {
      f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
      float a = lane, b = lane * 0.5f;
      for (int i = 0; i < iters; ++i) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
      }
      r = d0[0] + d1[1] + d2[2] + d3[3];
    }
  } else {
    if (mode & B_PRIO) __builtin_amdgcn_s_setprio(3);
    if (mode & B_CHAIN) {
      float x = lane, y = 1.0001f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fmaf(x, y, 0.5f);
      }
      r = x;
    }
    if (mode & B_INDEP) {
      float x[8], y = 1.0001f;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = lane + e;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], y, 0.5f);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) r += x[e];
    }
    if (mode & B_LDS) {
      int idx = lane;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) idx = (int)lds[idx & 511] & 511;
      }
      r = idx;
    }
    if (mode & B_MFMA_LDS) {
      f32x4 d0 = {0, 0, 0, 0};
      float a = lane, b = lane * 0.5f;
      for (int i = 0; i < iters; ++i) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
        lds[1024 + threadIdx.x] = d0[0];
        b = lds[1024 + (threadIdx.x ^ 1)];
      }
      r = d0[0];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}

int main() {
  unsigned long long* d;
  float* sk;
  if (hipMalloc(&d, 64) != hipSuccess || hipMalloc(&sk, 4) != hipSuccess) return 1;
  struct { int mode; const char* what; } runs[] = {
      {A_MFMA, "A: MFMA stream alone"},
      {A_MFMA_VALU, "A: MFMA + 4 dependent v_fma, one wave"},
      {B_CHAIN, "B: dependent v_fma chain alone"},
      {A_MFMA | B_CHAIN, "   beside A"},
      {A_MFMA | B_CHAIN | B_PRIO, "   beside A, s_setprio 3"},
      {B_INDEP, "B: 8 independent v_fma chains alone"},
      {A_MFMA | B_INDEP, "   beside A"},
      {A_MFMA | B_INDEP | B_PRIO, "   beside A, s_setprio 3"},
      {B_LDS, "B: LDS round-trip chain alone"},
      {A_MFMA | B_LDS, "   beside A"},
      {A_MFMA | B_LDS | B_PRIO, "   beside A, s_setprio 3"},
      {B_MFMA_LDS, "B: MFMA -> LDS write -> LDS read chain alone"},
      {A_MFMA | B_MFMA_LDS, "   beside A"},
      {A_MFMA | B_MFMA_LDS | B_PRIO, "   beside A, s_setprio 3"},
  };
  for (auto& rn : runs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, rn.mode, 1000, sk);
      if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    unsigned long long h[8];
    if (hipMemcpy(h, d, 64, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("%-48s A %8llu  B %8llu cycles / 1000 iterations\n", rn.what, h[0], h[4]);
  }
  return 0;
}
