// Scattered returning 32-bit atomics, as the one-pass binner issues them (6 per thread on 16384 counters): the rate when
// every workgroup may hit any counter, and when a workgroup only hits the eighth of the counters "of its XCD"
// (workgroups are dealt to the XCDs round-robin) -- would XCD-private counter replicas be cheaper?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_atomics.hip -o build/ub/ubench_atomics && build/ub/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int RETURNING, int LOCAL>
__global__ __launch_bounds__(256) void k_atomics(unsigned* __restrict__ counters, unsigned ncounters, int per_thread, unsigned* __restrict__ sink) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = t * 2654435761u + 12345u, acc = 0;
  const unsigned xcd = blockIdx.x & 7u, per = ncounters / 8u;
  for (int i = 0; i < per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned idx = LOCAL ? xcd * per + (h >> 8) % per : (h >> 8) % ncounters;
    if (RETURNING) acc += atomicAdd(&counters[idx], 1u);
    else atomicAdd(&counters[idx], 1u);
  }
  if (acc == 0xffffffffu) *sink = acc;
}

__global__ __launch_bounds__(256) void k_atomics_stride(unsigned* __restrict__ counters, unsigned ncounters, unsigned stride, int per_thread,
                                                       unsigned* __restrict__ sink) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = t * 2654435761u + 12345u, acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    acc += atomicAdd(&counters[(size_t)((h >> 8) % ncounters) * stride], 1u);
  }
  if (acc == 0xffffffffu) *sink = acc;
}

int main() {
  const unsigned ncounters = 16384;
  unsigned *counters, *sink;
  CK(hipMalloc(&counters, ncounters * 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int threads = 100000, per_thread = 6, blocks = (threads + 255) / 256;
  for (int variant = 0; variant < 4; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipMemset(counters, 0, ncounters * 4));
      CK(hipEventRecord(e0));
      switch (variant) {
        case 0: hipLaunchKernelGGL((k_atomics<1, 0>), dim3(blocks), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        case 1: hipLaunchKernelGGL((k_atomics<1, 1>), dim3(blocks), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        case 2: hipLaunchKernelGGL((k_atomics<0, 0>), dim3(blocks), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        default: hipLaunchKernelGGL((k_atomics<0, 1>), dim3(blocks), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3 && ms < best) best = ms;
    }
    const char* names[] = {"returning, any counter", "returning, own XCD's eighth", "no return, any counter", "no return, own XCD's eighth"};
    printf("%-30s %d threads x %d atomics on %u counters: %7.2f us  (%.1f G atomics/s)\n", names[variant], threads, per_thread,
           ncounters, 1e3 * best, threads * (double)per_thread / (best * 1e-3) / 1e9);
  }
  // the same with ten times the threads (throughput rather than launch latency)
  for (int variant = 0; variant < 4; ++variant) {
    float best = 1e9f;
    const int T = 1000000, B = (T + 255) / 256;
    for (int rep = 0; rep < 10; ++rep) {
      CK(hipMemset(counters, 0, ncounters * 4));
      CK(hipEventRecord(e0));
      switch (variant) {
        case 0: hipLaunchKernelGGL((k_atomics<1, 0>), dim3(B), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        case 1: hipLaunchKernelGGL((k_atomics<1, 1>), dim3(B), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        case 2: hipLaunchKernelGGL((k_atomics<0, 0>), dim3(B), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
        default: hipLaunchKernelGGL((k_atomics<0, 1>), dim3(B), dim3(256), 0, 0, counters, ncounters, per_thread, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
    }
    const char* names[] = {"returning, any counter", "returning, own XCD's eighth", "no return, any counter", "no return, own XCD's eighth"};
    printf("%-30s %d threads x %d atomics on %u counters: %7.2f us  (%.1f G atomics/s)\n", names[variant], 1000000, per_thread,
           ncounters, 1e3 * best, 1000000 * (double)per_thread / (best * 1e-3) / 1e9);
  }
  // counters spread out: one per `stride` words (does the rate depend on how many counters share a cache line?)
  for (unsigned stride : {1u, 2u, 4u, 8u, 16u, 32u}) {
    unsigned* spread;
    CK(hipMalloc(&spread, (size_t)ncounters * stride * 4));
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipMemset(spread, 0, (size_t)ncounters * stride * 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_atomics_stride, dim3(blocks), dim3(256), 0, 0, spread, ncounters, stride, per_thread, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3 && ms < best) best = ms;
    }
    printf("returning, any counter, one counter per %2u words: %7.2f us  (%.1f G atomics/s)\n", stride, 1e3 * best,
           threads * (double)per_thread / (best * 1e-3) / 1e9);
    CK(hipFree(spread));
  }
  return 0;
}
