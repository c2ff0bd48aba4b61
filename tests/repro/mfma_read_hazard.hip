// Reduced case for tests/repro/README.md section 2 (hipcc of ROCm 7.2, gfx950).
// Registers 14 / 15 of v_mfma_f32_32x32x2_f32's destination are not interlocked: a vector read needs 18 wait
// states after the MFMA (mfma_wait_states.hip).  hipcc pads for that along the LAYOUT order of the blocks: here a
// wave-uniform branch skips the block between the MFMA and the read of acc1[15], and the taken path gets 8
// (`s_and_b64; s_cbranch_vccnz; s_nop 5; v_accvgpr_read_b32 v19, a15`).  The two launches run the same arithmetic;
// they differ only in whether the block is executed, so any difference is the stale read.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hz tests/repro/mfma_read_hazard.hip && /tmp/hz     (exit code 1 = stale read)
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o /tmp/hz.s tests/repro/mfma_read_hazard.hip &&
//   python tests/repro/check_mfma_hazards.py /tmp/hz.s                                          (flags the path)
// (Whether the run shows it depends on which register hipcc reads first: a read of registers 0..13 stalls until the
// MFMA is through and hides the defect; this form reads a15 first.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void k(const float* a, const float* b, float* out, unsigned long long* dbg, int n) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  unsigned long long t0 = 0;
  float av = a[threadIdx.x], bv = b[threadIdx.x];
#pragma unroll 1
  for (int it = 0; it < n; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (dbg != nullptr) {            // wave-uniform; false in the run under test: the branch skips this block
      const unsigned long long t = __builtin_readcyclecounter();
      dbg[0] += t - t0;
      t0 = t;
    }
    const float x = acc1[15], y = acc1[0];      // readers of the last MFMA's destination
    av = av * 0.5f + x * 1e-3f;
    bv = bv * 0.5f + y * 1e-3f;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + 3.f * acc1[i];
  out[threadIdx.x] = s + av + bv;
}

int main() {
  float ha[64], hb[64], ref[64], got[64], *a, *b, *o;
  unsigned long long* dbg;
  for (int i = 0; i < 64; ++i) { ha[i] = 0.01f * (i + 1); hb[i] = 0.02f * (64 - i); }
  (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&o, 256); (void)hipMalloc(&dbg, 8);
  (void)hipMemset(dbg, 0, 8);
  (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice);
  (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, dbg, 8);                          // block executed: >= 18 wait states
  (void)hipMemcpy(ref, o, 256, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, (unsigned long long*)nullptr, 8);   // block skipped
  (void)hipMemcpy(got, o, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += got[i] != ref[i];
  printf("lanes whose result changes when the branch skips the block: %d of 64\n", bad);
  return bad ? 1 : 0;
}
