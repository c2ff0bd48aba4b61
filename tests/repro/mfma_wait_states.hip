// How many wait states does gfx950 need between an fp32 MFMA and a vector read of its destination, register by
// register?  Hand-written sequence per (instruction, register R, distance W):
//     MFMA; MFMA (same accumulator); s_nop W-1; v_accvgpr_read a[R] (early); 32 wait states; v_accvgpr_read a[R] (late)
// and the two reads are compared.  Measured on MI355X (ROCm 7.2), see tests/repro/README.md section 2:
//     v_mfma_f32_32x32x2_f32 (16 passes): registers 14, 15 stale up to W = 17, every other register fresh at W = 0
//     v_mfma_f32_16x16x4_f32 ( 8 passes): registers  2,  3 stale up to W = 9
//     v_mfma_f32_4x4x1_16b_f32 (2 passes): register   3    stale up to W = 3
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o /tmp/ws tests/repro/mfma_wait_states.hip && /tmp/ws
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>

#define ZERO4(b) "v_accvgpr_write_b32 a" #b "0, 0\n\t"
#define ZERO16                                                                                                \
  "v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\t"   \
  "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"   \
  "v_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\t" \
  "v_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\t"
#define CLOB16 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"

#define KERNEL(NAME, MFMA)                                                                                    \
  template <int W, int R>                                                                                     \
  __global__ __launch_bounds__(64) void NAME(const float* a, const float* b, float* early, float* late) {     \
    const float av = a[threadIdx.x], bv = b[threadIdx.x];                                                     \
    float e, l;                                                                                               \
    asm volatile(ZERO16 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" MFMA "\n\t" MFMA "\n\t"                           \
                 ".if %c4 > 0\n\ts_nop %c4 - 1\n\t.endif\n\t"                                                 \
                 "v_accvgpr_read_b32 %0, a%c5\n\t"                                                            \
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                               \
                 "v_accvgpr_read_b32 %1, a%c5\n\ts_nop 1\n\t"                                                 \
                 : "=&v"(e), "=&v"(l) : "v"(av), "v"(bv), "i"(W), "i"(R) : CLOB16);                           \
    early[threadIdx.x] = e;                                                                                   \
    late[threadIdx.x] = l;                                                                                    \
  }
KERNEL(k32, "v_mfma_f32_32x32x2_f32 a[0:15], %2, %3, a[0:15]")
KERNEL(k16, "v_mfma_f32_16x16x4_f32 a[0:3], %2, %3, a[0:3]")
KERNEL(k4, "v_mfma_f32_4x4x1_16b_f32 a[0:3], %2, %3, a[0:3]")

static float *g_a, *g_b, *g_e, *g_l;
template <typename K>
static bool stale(K kern) {
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, g_a, g_b, g_e, g_l);
  float he[64], hl[64];
  (void)hipMemcpy(he, g_e, 256, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hl, g_l, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i)
    if (he[i] != hl[i]) return true;
  return false;
}
#define ROW(K, NAME, R) row<R>(NAME, [](auto w) { return stale(K<decltype(w)::value, R>); })
template <int R, typename F>
static void row(const char* name, F f) {
  printf("%-26s register %2d: stale at wait states", name, R);
  auto go = [&](auto... w) { ((f(w) ? (void)printf(" %d", decltype(w)::value) : (void)0), ...); };
  [&]<int... W>(std::integer_sequence<int, W...>) { go(std::integral_constant<int, W>{}...); }(std::make_integer_sequence<int, 21>{});
  printf("\n");
}

int main() {
  float ha[64], hb[64];
  for (int i = 0; i < 64; ++i) { ha[i] = (float)(i % 5 + 1); hb[i] = (float)((i * 7) % 4 + 1); }
  (void)hipMalloc(&g_a, 256); (void)hipMalloc(&g_b, 256); (void)hipMalloc(&g_e, 256); (void)hipMalloc(&g_l, 256);
  (void)hipMemcpy(g_a, ha, 256, hipMemcpyHostToDevice);
  (void)hipMemcpy(g_b, hb, 256, hipMemcpyHostToDevice);
  ROW(k32, "v_mfma_f32_32x32x2_f32", 0); ROW(k32, "v_mfma_f32_32x32x2_f32", 7); ROW(k32, "v_mfma_f32_32x32x2_f32", 12);
  ROW(k32, "v_mfma_f32_32x32x2_f32", 13); ROW(k32, "v_mfma_f32_32x32x2_f32", 14); ROW(k32, "v_mfma_f32_32x32x2_f32", 15);
  ROW(k16, "v_mfma_f32_16x16x4_f32", 0); ROW(k16, "v_mfma_f32_16x16x4_f32", 1); ROW(k16, "v_mfma_f32_16x16x4_f32", 2);
  ROW(k16, "v_mfma_f32_16x16x4_f32", 3);
  ROW(k4, "v_mfma_f32_4x4x1_16b_f32", 0); ROW(k4, "v_mfma_f32_4x4x1_16b_f32", 1); ROW(k4, "v_mfma_f32_4x4x1_16b_f32", 2);
  ROW(k4, "v_mfma_f32_4x4x1_16b_f32", 3);
  return 0;
}
