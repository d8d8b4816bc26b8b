// Operand / result layout of v_mfma_f32_32x32x16_{f16,bf16} on gfx950, checked against a CPU product.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_layout.hip -o build/ubench/probe_mfma && build/ubench/probe_mfma
// Hypothesis (what the render kernel's MFMA sweep is written for), D = A (32 x 16) * B (16 x 32) + C:
//   A: lane l, slot s (0..7)  ->  A[i = l % 32][k = 8 * (l / 32) + s]
//   B: lane l, slot s         ->  B[k = 8 * (l / 32) + s][j = l % 32]
//   D: lane l, register v     ->  D[i = 8 * (v / 4) + 4 * (l / 32) + v % 4][j = l % 32]
// Also reports whether the sum of 16 exact products is accumulated like a float32 chain or wider (one large
// cancelling pair plus a small term).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__host__ __device__ inline float fa(int i, int k) { return (float)((i + 3 * k) % 7 - 3); }
__host__ __device__ inline float fb(int k, int j) { return (float)((2 * k + j) % 5 - 2); }

template <bool BF>
__global__ void probe(float* out, float big) {
  const int l = threadIdx.x, half = l / 32;
  f16v acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  if (BF) {
    b8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (__bf16)fa(l % 32, 8 * half + s); b[s] = (__bf16)fb(8 * half + s, l % 32); }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else {
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)fa(l % 32, 8 * half + s); b[s] = (_Float16)fb(8 * half + s, l % 32); }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int v = 0; v < 16; ++v) out[l * 16 + v] = acc[v];
  // accumulation width: k = 0: big * 1, k = 1: -big * 1, k = 9 (other half): small * 1 -> exact answer = small
  f16v acc2;
  for (int v = 0; v < 16; ++v) acc2[v] = 0.0f;
  if (!BF) {
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)0.0f; b[s] = (_Float16)0.0f; }
    if (half == 0) { a[0] = (_Float16)big; a[1] = (_Float16)(-big); b[0] = (_Float16)big; b[1] = (_Float16)big; }
    else { a[1] = (_Float16)0.001f; b[1] = (_Float16)1.0f; }
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
  }
  if (l == 0) out[64 * 16] = acc2[0];
}

int main() {
  float* d;
  (void)hipMalloc(&d, (64 * 16 + 4) * sizeof(float));
  float h[64 * 16 + 4];
  for (int bf = 0; bf < 2; ++bf) {
    if (bf) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, d, 60000.0f);
    else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, d, 60000.0f);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * (l / 32) + v % 4, j = l % 32;
        float want = 0.0f;
        for (int k = 0; k < 16; ++k) want += fa(i, k) * fb(k, j);
        if (h[l * 16 + v] != want) { if (bad < 5) printf("  lane %d reg %d: got %g want %g\n", l, v, h[l * 16 + v], want); ++bad; }
      }
    printf("%s: %d of 1024 elements differ from the hypothesised layout\n", bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x16_f16", bad);
    if (!bf) printf("  accumulation probe: 60000^2 - 60000^2 + 0.001 -> %g (0.001 = exact; 0 = the small term was lost to a float32 partial sum)\n", h[64 * 16]);
  }
  return 0;
}
