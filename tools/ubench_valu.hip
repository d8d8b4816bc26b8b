#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS(X) \
  X(0, "v_mul_f32 %0, %0, %1", x, fa) \
  X(1, "v_mul_f32_e64 %0, %0, %1", x, fa) \
  X(2, "v_add_f32 %0, %0, %1", x, fa) \
  X(3, "v_fmac_f32 %0, %1, %1", x, fa) \
  X(4, "v_min_f32 %0, %0, %1", x, fa) \
  X(5, "v_and_b32 %0, %0, %1", u, v1) \
  X(6, "v_add_u32 %0, %0, %1", u, v1) \
  X(7, "v_mov_b32 %0, %1", u, v1) \
  X(8, "v_min_u32 %0, %0, %1", u, v1) \
  X(9, "v_max_f32 %0, %0, %1", x, fa) \
  X(10, "v_sub_f32 %0, %0, %1", x, fa) \
  X(11, "v_fma_f32 %0, %0, %1, %1", x, fa) \
  X(12, "v_or_b32 %0, %0, %1", u, v1) \
  X(13, "v_lshlrev_b32 %0, 1, %0", u, v1) \
  X(14, "v_mul_f64 %0, %0, %1", d, da) \
  X(15, "v_add_f64 %0, %0, %1", d, da) \
  X(16, "v_fma_f64 %0, %0, %1, %1", d, da) \
  X(17, "v_mul_lo_u32 %0, %0, %1", u, v1)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, uint32_t c1, long long* cyc) {
  float x[8]; uint32_t u[8]; double d[8];
  uint32_t v1 = c1 + threadIdx.x;
  float fa = a + threadIdx.x * 1e-9f; double da = fa;
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7 + i; d[i] = x[i]; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(N, S, VAR, OPD) if (KIND == N) asm volatile(S : "+v"(VAR[i]) : "v"(OPD));
        OPS(X)
#undef X
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i] + (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  long long* cyc; hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 77u, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 77u, cyc); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  double winstr = (double)blocks * 4 * iters * 32;
  printf("%-28s ms=%.3f  ns/wave-instr/SIMD = %.3f   wave0 memtime ticks/instr = %.2f\n", name, ms, ms * 1e6 * 1024 / winstr, (double)hc / (iters * 32));
  hipFree(out);
}
int main() {
#define X(N, S, VAR, OPD) run<N>(S, 4096);
  OPS(X)
}
