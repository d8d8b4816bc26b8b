// What does a kernel boundary do to the XCD L2s?  R reads a 16 MB buffer, every workgroup its own 8 KB chunk, so each
// XCD (workgroups are dealt round-robin) reads a fixed 2 MB eighth that fits its 4 MB L2.  Timed: R after R (is the
// L2 kept across launches?), R after a kernel that rewrote the buffer from the SAME XCDs, from OTHER XCDs, or wrote
// an unrelated buffer; with vector or scalar loads in R.  Sums are checked against the value last written.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_l2.hip -o build/ub/ubench_l2 && build/ub/ubench_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kChunk = 2048;   // floats per workgroup chunk (8 KB)

__global__ __launch_bounds__(256) void k_read_vec(const float* __restrict__ x, float* __restrict__ out) {
  const float4* p = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * kChunk);
  float4 a = p[threadIdx.x], b = p[256 + threadIdx.x];
  float s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
  for (int m = 32; m; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}
// the same bytes through the scalar cache: every wave walks its quarter of the chunk with wave-uniform 16-byte loads
__global__ __launch_bounds__(256) void k_read_scalar(const float* __restrict__ x, float* __restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float4* p = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * kChunk + wave * (kChunk / 4));
  float s = 0.0f;
#pragma unroll 4
  for (int i = 0; i < kChunk / 16; ++i) { const float4 v = p[i]; s += (v.x + v.y) + (v.z + v.w); }
  if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int FLAVOUR>   // 0 plain, 1 nontemporal, 2 sc1, 3 sc0 sc1
__global__ __launch_bounds__(256) void k_write(float* __restrict__ x, float val, int shift, int nblocks) {
  const int c = (blockIdx.x + shift) % nblocks;
  vf4* p = reinterpret_cast<vf4*>(x + (size_t)c * kChunk);
  const vf4 v = {val, val, val, val};
  for (int k = 0; k < 2; ++k) {
    vf4* q = p + k * 256 + threadIdx.x;
    if (FLAVOUR == 0) *q = v;
    else if (FLAVOUR == 1) __builtin_nontemporal_store(v, q);
    else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
  }
}
static void launch_write(int flavour, hipStream_t st, float* x, float val, int shift, int nblocks) {
  switch (flavour) {
    case 0: hipLaunchKernelGGL(k_write<0>, dim3(nblocks), dim3(256), 0, st, x, val, shift, nblocks); break;
    case 1: hipLaunchKernelGGL(k_write<1>, dim3(nblocks), dim3(256), 0, st, x, val, shift, nblocks); break;
    case 2: hipLaunchKernelGGL(k_write<2>, dim3(nblocks), dim3(256), 0, st, x, val, shift, nblocks); break;
    default: hipLaunchKernelGGL(k_write<3>, dim3(nblocks), dim3(256), 0, st, x, val, shift, nblocks); break;
  }
}

int main() {
  const int nblocks = 2048;
  const size_t n = (size_t)nblocks * kChunk;
  float *x, *y, *out;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&out, nblocks * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> h(nblocks);
  const char* names[] = {"R after R (nothing rewritten)", "R after W from the same XCDs", "R after W from other XCDs",
                         "R after W of an unrelated buffer", "second R after W from other XCDs"};
  const char* flav[] = {"plain", "nt", "sc1", "sc0 sc1"};
  for (int scalar = 0; scalar < 2; ++scalar) {
    for (int flavour = 0; flavour < 4; ++flavour) {
      for (int var = 0; var < 5; ++var) {
        if (flavour > 0 && (var == 0 || var == 3)) continue;
        float total = 0.0f; int bad = 0; const int reps = 50;
        float val = 1.0f;
        launch_write(0, st, x, val, 0, nblocks);
        for (int r = 0; r < reps + 5; ++r) {
          if (var == 1 || var == 2 || var == 4) { val += 1.0f; launch_write(flavour, st, x, val, var == 1 ? 8 : 3, nblocks); }
          if (var == 3) launch_write(flavour, st, y, val, 1, nblocks);
          if (var == 0 || var == 4) {                                                             // the "previous R"
            if (scalar) hipLaunchKernelGGL(k_read_scalar, dim3(nblocks), dim3(256), 0, st, x, out);
            else hipLaunchKernelGGL(k_read_vec, dim3(nblocks), dim3(256), 0, st, x, out);
          }
          CK(hipMemsetAsync(out, 0, nblocks * 4, st));
          CK(hipEventRecord(e0, st));
          if (scalar) hipLaunchKernelGGL(k_read_scalar, dim3(nblocks), dim3(256), 0, st, x, out);
          else hipLaunchKernelGGL(k_read_vec, dim3(nblocks), dim3(256), 0, st, x, out);
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 5) total += ms;
          CK(hipMemcpy(h.data(), out, nblocks * 4, hipMemcpyDeviceToHost));
          for (int b = 0; b < nblocks; ++b) if (h[b] != val * kChunk) ++bad;
        }
        printf("%-7s %-8s %-36s %7.2f us per R (%5.0f GB/s)  wrong sums: %d\n", scalar ? "scalar" : "vector", flav[flavour],
               names[var], 1e3 * total / reps, 16.777 / (total / reps), bad);
      }
    }
  }
  return 0;
}
