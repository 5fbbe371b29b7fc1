// LDS pointer-chase latency on gfx950: one wavefront per workgroup, every lane follows a uint16 table for N dependent
// steps, with more and more of K1L's walker chain added.  Prints ns per dependent step.
//   hipcc --offload-arch=gfx950 -O3 -o lds_chase lds_chase.hip && ./lds_chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V>
__global__ void __launch_bounds__(256) k_chase(const unsigned short* __restrict__ init, int n_entries, int steps, int mask, int start,
                                              int H, int* out) {
  extern __shared__ unsigned short tab[];
  for (int i = threadIdx.x; i < n_entries; i += blockDim.x) tab[i] = init[i];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lanebase = (threadIdx.x % 52) * 900;  // 52 "instances", each its own 900-entry region
  int cur = 0, h = 0, acc = 0;
  unsigned char* cnt = reinterpret_cast<unsigned char*>(tab + 52 * 900) + (threadIdx.x % 52) * 64;
  for (int s = 0; s < steps; ++s) {
    if (V == 0) { cur = tab[lanebase + cur] & 511; }                                   // pure chase (+ and)
    if (V == 1) { const int a = (s * 7 + threadIdx.x) & 1; cur = tab[lanebase + cur + a] & mask; }   // + action
    if (V >= 2) {                                                                       // + episode-end select
      const int a = (s * 7 + threadIdx.x) & 1;
      const int w = tab[lanebase + cur + a];
      const int nxt = w & mask;
      ++h;
      const bool term = h >= H;
      cur = term ? start : nxt;
      h = term ? 0 : h;
      if (V >= 3) { const int c = cnt[nxt & 63] + 1; cnt[nxt & 63] = (unsigned char)c; acc += c >> 8; }   // + 8-bit RMW
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = cur + acc;
}
int main() {
  const int n = 52 * 900 + 52 * 64, steps = 200000;
  std::vector<unsigned short> h(n);
  unsigned x = 12345;
  for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 8) % 448) * 2); }
  unsigned short* d; int* o;
  (void)hipMalloc(&d, n * 2); (void)hipMalloc(&o, 256 * 64 * 4);
  (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
#define RUN(V, threads)                                                                                              \
  {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chase<V>), hipFuncAttributeMaxDynamicSharedMemorySize, n * 2); \
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);                                             \
    hipLaunchKernelGGL(k_chase<V>, dim3(256), dim3(threads), n * 2, 0, d, n, 1000, 1023, 0, 30, o);                  \
    (void)hipEventRecord(a);                                                                                         \
    hipLaunchKernelGGL(k_chase<V>, dim3(256), dim3(threads), n * 2, 0, d, n, steps, 1023, 0, 30, o);                 \
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);                                                           \
    float ms; (void)hipEventElapsedTime(&ms, a, b);                                                                  \
    printf("variant %d (%d threads): %.1f ns per step\n", V, threads, ms * 1e6 / steps);                             \
  }
  RUN(0, 64) RUN(1, 64) RUN(2, 64) RUN(3, 64) RUN(3, 256)
  return 0;
}
