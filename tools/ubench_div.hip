// Bit-identity check (MI355X): three IEEE fp64 divisions by a common denominator, as hipcc expands them, against the
// shared-reciprocal form used by pixel_ray (srh_device.h).  Prints the number of mismatching results.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_div.hip -o build/ubench/divcheck && build/ubench/divcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ void div3_shared(const double v[3], double len, double d[3]) {
  const double r = __builtin_amdgcn_rcp(len);
  const double e0 = __builtin_fma(-len, r, 1.0);
  const double r1 = __builtin_fma(r, e0, r);
  const double e1 = __builtin_fma(-len, r1, 1.0);
  const double r2 = __builtin_fma(r1, e1, r1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double q = v[i] * r2;
    const double rem = __builtin_fma(-len, q, v[i]);
    d[i] = __builtin_fma(rem, r2, q);
  }
}
__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void k(unsigned long long* bad, int iters, int mode) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long nb = 0;
  for (int it = 0; it < iters; ++it) {
    double v[3], len;
    if (mode == 0) {          // camera-like magnitudes
      for (int i = 0; i < 3; ++i) v[i] = ((double)(int64_t)rng(s)) * (1.0 / 9.2e18) * 3.0;
      len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    } else {                  // wide exponent range, incl. zeros and cancellation-sized values
      for (int i = 0; i < 3; ++i) {
        const double m = ((double)(int64_t)rng(s)) * (1.0 / 9.2e18);
        const int ex = (int)(rng(s) % 400) - 200;
        v[i] = (rng(s) % 17 == 0) ? 0.0 : ldexp(m, ex);
      }
      len = ldexp(1.0 + (double)(rng(s) % 1000003) / 1000003.0, (int)(rng(s) % 200) - 100);
    }
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) a[i] = v[i] / len;
    div3_shared(v, len, b);
    for (int i = 0; i < 3; ++i) nb += (__double_as_longlong(a[i]) != __double_as_longlong(b[i]));
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  unsigned long long* bad; hipMalloc(&bad, 8);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, 400, mode);
    unsigned long long h; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %llu mismatches in %.3g divisions\n", mode, h, 4096.0 * 256 * 400 * 3);
  }
}
