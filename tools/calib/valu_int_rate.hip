// Cycles a wave64 INTEGER VALU instruction occupies a SIMD on gfx950, for the instructions of K1E's walk (cmdp_k1e.h):
// v_bfe_u32, v_lshl_or_b32, v_and_b32, v_and_or_b32, v_alignbit_b32, against v_fma-class float32 (tools/calib/valu_rate.hip:
// 2.2-2.5 cycles at 4-8 wavefronts per SIMD).  W wavefronts per SIMD run a long stream of independent instructions
// (ILP = 8 chains per lane); kernel time -> wave-instructions per second -> cycles per instruction at the nominal clock.
//   hipcc --offload-arch=gfx950 -O3 -o valu_int_rate valu_int_rate.hip && ./valu_int_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int ILP>
__global__ void k(uint32_t* out, int iters, uint32_t a, uint32_t b) {
  uint32_t x[ILP];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 y[ILP];
  const f2 yb = {__uint_as_float(a), __uint_as_float(b)};
#pragma unroll
  for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 2654435761u + i; y[i] = (f2){(float)i, (float)threadIdx.x}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      if (OP == 1) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(b));
      if (OP == 2) asm volatile("v_bfe_u32 %0, %0, %1, 17" : "+v"(x[i]) : "v"(b));
      if (OP == 3) asm volatile("v_alignbit_b32 %0, %1, %0, 2" : "+v"(x[i]) : "v"(a));
      if (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (OP == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      if (OP == 6) { float f = __uint_as_float(x[i]); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(__uint_as_float(a)), "v"(__uint_as_float(b))); x[i] = __float_as_uint(f); }
      if (OP == 7) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(yb));   // two float32 per lane and instruction
      if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(yb));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i] + __float_as_uint(y[i].x) + __float_as_uint(y[i].y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* out, int cus, int khz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 100000;
  constexpr int ILP = 8;
  for (int w : {1, 2, 4, 8}) {
    const int threads = 64 * 4 * w > 1024 ? 1024 : 64 * 4 * w;
    const int blocks_per_cu = (64 * 4 * w) / threads;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<OP, ILP>), dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, iters, 0xfffffff7u, 5u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * ILP * w;
    const double ns = ms * 1e6 / insts_per_simd;
    printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles per wave64 instruction per SIMD at %.2f GHz (nominal)\n", name, w, ms,
           ns * khz * 1e-6, khz * 1e-6);
  }
}

int main() {
  int cus = 0, khz = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  uint32_t* out;
  hipMalloc(&out, sizeof(uint32_t) * cus * 2048);
  run<6>("v_fma_f32", out, cus, khz);
  run<0>("v_and_b32", out, cus, khz);
  run<7>("v_or_b32", out, cus, khz);
  run<5>("v_add_u32", out, cus, khz);
  run<1>("v_lshl_or_b32", out, cus, khz);
  run<2>("v_bfe_u32", out, cus, khz);
  run<3>("v_alignbit_b32", out, cus, khz);
  run<4>("v_and_or_b32", out, cus, khz);
  run<8>("v_pk_mul_f32", out, cus, khz);
  run<9>("v_pk_add_f32", out, cus, khz);
  return 0;
}
