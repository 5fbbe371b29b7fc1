// Where do the wavefronts of a workgroup land?  Prints, for 512-thread workgroups at one and two workgroups per CU
// (LDS-limited, like K1P), the SIMD of every wavefront (HW_REG_HW_ID bits [5:4]) of the first few workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_placement wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = hw;
  // stay resident for a while so that workgroups really share CUs
  unsigned acc = 0;
  for (int i = 0; i < spin; ++i) { acc += i ^ (acc >> 3); __syncthreads(); }
  if (acc == 0x7fffffff) smem[0] = 1;
}
int main() {
  unsigned* d;
  const int nwg = 1024;
  hipMalloc(&d, nwg * 8 * 4);
  for (int per_cu : {1, 2}) {
    const size_t lds = per_cu == 1 ? 150 * 1024 : 78 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipMemset(d, 0xff, nwg * 8 * 4);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 8);
    hipMemcpy(h.data(), d, nwg * 8 * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0};  // pattern histogram: number of workgroups whose SIMD pattern is 0,1,2,3,0,1,2,3
    int regular = 0;
    for (int g = 0; g < nwg; ++g) {
      bool reg = true;
      for (int w = 0; w < 8; ++w) reg = reg && (((h[g * 8 + w] >> 4) & 3) == (unsigned)(w & 3));
      regular += reg;
    }
    printf("per_cu=%d: %d of %d workgroups have wave w on SIMD w%%4\n", per_cu, regular, nwg);
    for (int g = 0; g < 6; ++g) {
      printf("  wg %d (cu %u se %u):", g, (h[g * 8] >> 8) & 15, (h[g * 8] >> 13) & 7);
      for (int w = 0; w < 8; ++w) printf(" %u", (h[g * 8 + w] >> 4) & 3);
      printf("\n");
    }
    (void)hist;
  }
  return 0;
}
