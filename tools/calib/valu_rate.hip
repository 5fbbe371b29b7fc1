// How many cycles does a wave64 float32 VALU instruction occupy a SIMD on gfx950?  The VI kernels' `valu_issue` roofline
// depends on it.  W wavefronts per SIMD (one workgroup of 4 W wavefronts per CU, 256 workgroups) run a long stream of
// independent v_mul_f32 / v_add_f32 (ILP = 8 chains per lane, -ffp-contract=off keeps them separate); the kernel time
// gives wave-instructions per second, and with the measured clock (s_memrealtime vs clock64) cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k(float* out, int iters, float a, float b) {
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = __fadd_rn(__fmul_rn(x[i], a), b);   // 2 VALU instructions per chain and iteration
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  int cus = 0, khz = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  float* out;
  hipMalloc(&out, sizeof(float) * cus * 2048);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200000;
  constexpr int ILP = 8;
  for (int w : {1, 2, 3, 4, 8}) {
    const int threads = 64 * 4 * w > 1024 ? 1024 : 64 * 4 * w;
    const int blocks_per_cu = (64 * 4 * w) / threads;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<ILP>, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, iters, 1.0000001f, 1e-9f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * ILP * 2 * w;   // wave-instructions issued on one SIMD
    const double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("waves/SIMD %d: %.3f ms, %.3f ns per wave64 f32 instruction per SIMD = %.2f cycles at %.2f GHz (nominal)\n", w, ms,
           ns_per_inst, ns_per_inst * khz * 1e-6, khz * 1e-6);
  }
  return 0;
}
