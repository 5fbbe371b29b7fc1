// Cost of one workgroup barrier round (s_barrier) on gfx950, by workgroup size: a loop of N barriers with a few
// scalar instructions between them, one workgroup per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(int n, int* out) {
  int acc = 0;
  for (int i = 0; i < n; ++i) {
    acc += i ^ (acc >> 3);
    __syncthreads();
  }
  if (acc == 0x7fffffff) out[0] = acc;
}
__global__ void k_nobar(int n, int* out) {
  int acc = 0;
  for (int i = 0; i < n; ++i) {
    acc += i ^ (acc >> 3);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0" ::: "memory");
  }
  if (acc == 0x7fffffff) out[0] = acc;
}
int main() {
  int* d;
  hipMalloc(&d, 4);
  const int n = 200000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {64, 128, 256, 512, 1024}) {
    for (int variant = 0; variant < 2; ++variant) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, n, d);
        else hipLaunchKernelGGL(k_nobar, dim3(256), dim3(threads), 0, 0, n, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("threads %4d %s: %.1f ns per iteration\n", threads, variant == 0 ? "barrier" : "no barrier", best * 1e6 / n);
    }
  }
  return 0;
}
