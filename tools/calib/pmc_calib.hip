// Calibration of the FETCH_SIZE / WRITE_SIZE counters for the access pattern of K5S: one dword per lane, 256-byte rows.
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip ; rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./pmc_calib
// Reads (k_read_dword) and writes (k_write_dword) 8 GiB each, far beyond the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read_dword(const float* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_write_dword(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}
int main() {
  const size_t n = (size_t)2 << 30;  // 2 Gi floats = 8 GiB
  float *p, *o;
  if (hipMalloc(&p, n * 4) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  hipMemset(p, 0, n * 4);
  hipLaunchKernelGGL(k_write_dword, dim3(256 * 16), dim3(256), 0, 0, p, n);
  hipLaunchKernelGGL(k_read_dword, dim3(256 * 16), dim3(256), 0, 0, p, n, o);
  hipDeviceSynchronize();
  printf("bytes %zu\n", n * 4);
  return 0;
}
