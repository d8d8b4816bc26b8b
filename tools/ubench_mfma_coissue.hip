// Does the f32 matrix pipe run BESIDE the vector pipe for the sweep's instruction mix?  (round-3)
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_coissue.hip -o build/ubench/mfma && build/ubench/mfma
//
// The MFMA form of the render kernel's sweep would, per block of 32 tile pixels x 32 list entries, issue five
// v_mfma_f32_32x32x2_f32 (three chained for the ellipse form, two chained for the depth estimate: 64 cycles each on the
// SIMD's matrix pipe) and then ~96 vector instructions on the 2 x 16 accumulator values of each lane (per value:
// v_min_f32, v_bfi_b32, 3 x v_med3_i32, v_max_i32).  This program times exactly that loop body in three variants --
// matrix only, vector only, both (the vector part works on the PREVIOUS block's accumulators, as a software-pipelined
// kernel would) -- at K = 1, 2, 4, 5 waves per SIMD, and reports cycles per block per SIMD (2.4 GHz nominal).
// If "both" ~ max(matrix, vector) the matrix pipe is free issue capacity; if ~ their sum, it is not.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// F16 = the same loop with TWO v_mfma_f32_32x32x16_f16 per block instead (K = 16 holds the six monomials times a
// hi / lo split of the coefficients: one instruction per output matrix), f32 accumulation
template <int MODE, bool F16 = false>   // 1 = matrix only, 2 = vector only, 3 = both
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, uint32_t c1) {
  f16v accq, accd, prevq, prevd;
  for (int i = 0; i < 16; ++i) { accq[i] = 0.0f; accd[i] = 0.0f; prevq[i] = threadIdx.x * 0.01f + i; prevd[i] = 1.0f + i * 0.001f; }
  int32_t k1 = 0, k2 = 0, k3 = 0, k4 = 0;
  const float A0 = a + threadIdx.x * 1e-3f, B0 = 1.0f + threadIdx.x * 1e-4f;
  uint32_t field = c1;
  h8v HA, HB;
  for (int i = 0; i < 8; ++i) { HA[i] = (_Float16)(0.5f + 0.01f * i + threadIdx.x * 1e-3f); HB[i] = (_Float16)(1.0f + i); }
#define MFMA5(Q, D)                                                                              \
  if (F16) {                                                                                     \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(Q) : "v"(HA), "v"(HB));        \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(D) : "v"(HB), "v"(HA));        \
  } else {                                                                                       \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(Q) : "v"(A0), "v"(B0));           \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(D) : "v"(B0), "v"(A0));           \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(Q) : "v"(A0), "v"(A0));           \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(D) : "v"(B0), "v"(B0));           \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(Q) : "v"(B0), "v"(A0));           \
  }
#define VALU96(Q, D)                                                                             \
  _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
    int32_t sel, key;                                                                            \
    asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(sel) : "v"(Q[v]));                            \
    asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(key) : "s"(0xFFFFF000u), "v"(D[v]), "v"(field)); \
    asm volatile("v_and_b32 %0, %0, %1" : "+v"(key) : "v"(sel));                                 \
    asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(k4) : "v"(k3), "v"(key));                    \
    asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(k3) : "v"(k2), "v"(key));                    \
    asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(k2) : "v"(k1), "v"(key));                    \
    asm volatile("v_max_i32 %0, %0, %1" : "+v"(k1) : "v"(key));                                  \
  }
  // two accumulator sets: while the matrix pipe fills one, the vector part works on the other (no copies)
  for (int it = 0; it < iters; it += 2) {
    if (MODE & 1) { MFMA5(accq, accd) }
    if (MODE & 2) { VALU96(prevq, prevd) }
    field += 1;
    if (MODE & 1) { MFMA5(prevq, prevd) }
    if (MODE & 2) { VALU96(accq, accd) }
    field += 1;
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += accq[i] + accd[i] + prevq[i] + prevd[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + (float)(k1 ^ k2 ^ k3 ^ k4);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.gcnArchName, cus);
  float* out;
  hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float));
  const int iters = 2000;
  const char* names[8] = {"", "matrix only (5 MFMA 32x32x2 f32)", "vector only (112 VALU)", "both (f32 MFMA)",
                          "", "matrix only (2 MFMA 32x32x16 f16)", "", "both (f16 MFMA)"};
  for (int mode : {1, 2, 3, 5, 7}) {
    printf("%-36s", names[mode]);
    for (int K : {1, 2, 4, 5}) {
      const dim3 grid(cus * 4 * K), block(64);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 0.5f, 7u);
        if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, 0.5f, 7u);
        if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters, 0.5f, 7u);
        if (mode == 5) hipLaunchKernelGGL((k<1, true>), grid, block, 0, 0, out, iters, 0.5f, 7u);
        if (mode == 7) hipLaunchKernelGGL((k<3, true>), grid, block, 0, 0, out, iters, 0.5f, 7u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      // cycles per block per SIMD: wall x 2.4 GHz / (K waves x iters)
      printf("  K=%d: %7.1f cyc/block/SIMD (%.3f ms)", K, best * 1e-3 * 2.4e9 / ((double)K * iters), best);
    }
    printf("\n");
  }
  return 0;
}
