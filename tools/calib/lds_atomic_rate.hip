// What the LDS pipe and the step of K1E's walk (cmdp_k1e.h) cost on gfx950, measured in isolation.
//  (1) cycles a wave64 ds_read_b32 / ds_add_u32 / ds_add_rtn_u32 occupies a CU's LDS pipe, bank-conflict-free (lane i -> bank
//      i & 31), 16 wavefronts per CU, EPL independent operations in flight per wavefront;
//  (2) the walk's step itself -- v_bfe_u32 . v_lshl_or_b32 . v_and_or_b32 . ds_add_rtn_u32 . v_alignbit_b32, EPL chains per lane,
//      16 wavefronts of 1024 threads per CU on a 128-KB table -- as cycles per wave-transition per SIMD: the rate
//      k_rollout_epi's fast loop could reach if nothing else ran.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) uint32_t* lds_u32;

template <int OP, int EPL>
__global__ void __launch_bounds__(1024) k_lds(uint32_t* out, int iters) {
  extern __shared__ uint32_t tab[];
  for (int k = threadIdx.x; k < 32768; k += 1024) tab[k] = (uint32_t)(k * 2654435761u) & 0xff80u;
  __syncthreads();
  const uint32_t lane_base = 4u * (threadIdx.x & 31u);
  uint32_t a[EPL], r[EPL];
#pragma unroll
  for (int c = 0; c < EPL; ++c) { a[c] = lane_base + 128u * ((threadIdx.x * 7u + c * 131u) & 1023u); r[c] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < EPL; ++c) {
      if (OP == 0) r[c] += *(volatile lds_u32)(uintptr_t)a[c];
      if (OP == 1) __hip_atomic_fetch_add((lds_u32)(uintptr_t)a[c], 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 2) r[c] += __hip_atomic_fetch_add((lds_u32)(uintptr_t)a[c], 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < EPL; ++c) s += r[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the walk's step: every chain follows its successor words through the table (dependent), EPL chains per lane
template <int EPL, int FORM>
__global__ void __launch_bounds__(1024) k_walk(uint32_t* out, int iters, uint32_t ash) {
  extern __shared__ uint32_t tab[];
  for (int k = threadIdx.x; k < 32768; k += 1024) tab[k] = ((uint32_t)(k * 2654435761u) >> 9) & 0xff83u;   // successor word | code
  __syncthreads();
  const uint32_t x0 = 4u * (threadIdx.x & 31u), x1 = x0 | (1u << ash);
  uint32_t w[EPL], bits[EPL], cw[EPL];
#pragma unroll
  for (int c = 0; c < EPL; ++c) { w[c] = (threadIdx.x * 640u + c * 4736u) & 0xff80u; bits[c] = threadIdx.x * 2654435761u + c; cw[c] = 0; }
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < 32; j += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint32_t ra[EPL];
#pragma unroll
        for (int c = 0; c < EPL; ++c) {
          if (FORM == 0) {   // v_bfe_u32 . v_lshl_or_b32 . v_and_or_b32
            uint32_t a, t;
            asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(a) : "v"(bits[c]), "s"(j + u));
            asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "s"(ash), "v"(x0));
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ra[c]) : "v"(w[c]), "s"(0xff80u), "v"(t));
          } else if (FORM == 2) {   // v_add_co_u32 (shifts the next action bit into VCC) . v_cndmask_b32 (lane base of that action) . v_and_or_b32:
                                    // two VOP2 instructions instead of two VOP3 ones (bits consumed from the top: bit-reversed words)
            uint32_t t;
            asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc" : "+v"(bits[c]), "=v"(t) : "v"(x0), "v"(x1) : "vcc");
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ra[c]) : "v"(w[c]), "s"(0xff80u), "v"(t));
          } else {           // what the compiler may also pick: bfe . and (literal) . lshl . or3
            const uint32_t a = (bits[c] >> (j + u)) & 1u;
            ra[c] = (w[c] & 0xff80u) | ((a << ash) | x0);
          }
        }
#pragma unroll
        for (int c = 0; c < EPL; ++c)
          w[c] = __hip_atomic_fetch_add((lds_u32)(uintptr_t)ra[c], 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int c = 0; c < EPL; ++c) cw[c] = __builtin_amdgcn_alignbit(w[c], cw[c], 2u);
      }
    }
#pragma unroll
    for (int c = 0; c < EPL; ++c) bits[c] = bits[c] * 1664525u + 1013904223u + cw[c];
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < EPL; ++c) s += w[c] + cw[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static hipEvent_t e0, e1;
template <typename F> float timed(F f) {
  for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int cus = 0, khz = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t* out; hipMalloc(&out, sizeof(uint32_t) * cus * 1024);
  const size_t lds = 131072 + 128;
  hipFuncSetAttribute((const void*)k_lds<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_lds<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_lds<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_walk<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const double ghz = khz * 1e-6;
  const int it1 = 20000;
  const char* names[3] = {"ds_read_b32", "ds_add_u32", "ds_add_rtn_u32"};
  float ms[3];
  ms[0] = timed([&] { hipLaunchKernelGGL((k_lds<0, 4>), dim3(cus), dim3(1024), lds, 0, out, it1); });
  ms[1] = timed([&] { hipLaunchKernelGGL((k_lds<1, 4>), dim3(cus), dim3(1024), lds, 0, out, it1); });
  ms[2] = timed([&] { hipLaunchKernelGGL((k_lds<2, 4>), dim3(cus), dim3(1024), lds, 0, out, it1); });
  for (int i = 0; i < 3; ++i)   // wave-instructions per CU = iters * 4 * 16 wavefronts
    printf("%-16s 16 waves/CU, 4 in flight per wave: %.3f ms, %.2f cycles of the CU's LDS pipe per wave64 instruction at %.2f GHz\n",
           names[i], ms[i], ms[i] * 1e-3 * ghz * 1e9 / ((double)it1 * 4 * 16), ghz);
  const int it2 = 2000;
  auto report = [&](const char* name, float t, int epl) {   // wave-transitions per SIMD = iters * 32 * epl * 4 wavefronts
    const double wt = (double)it2 * 32 * epl * 4;
    printf("%-44s %.3f ms, %.2f cycles per wave-transition per SIMD (x 1/4 per CU LDS pipe: %.2f)\n", name, t,
           t * 1e-3 * ghz * 1e9 / wt, t * 1e-3 * ghz * 1e9 / (wt * 4));
  };
  report("walk step, 4 chains, bfe.lshl_or.and_or", timed([&] { hipLaunchKernelGGL((k_walk<4, 0>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 4);
  report("walk step, 4 chains, compiler's choice", timed([&] { hipLaunchKernelGGL((k_walk<4, 1>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 4);
  report("walk step, 4 chains, add_co.cndmask.and_or", timed([&] { hipLaunchKernelGGL((k_walk<4, 2>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 4);
  report("walk step, 2 chains, bfe.lshl_or.and_or", timed([&] { hipLaunchKernelGGL((k_walk<2, 0>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 2);
  report("walk step, 6 chains, bfe.lshl_or.and_or", timed([&] { hipLaunchKernelGGL((k_walk<6, 0>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 6);
  report("walk step, 8 chains, bfe.lshl_or.and_or", timed([&] { hipLaunchKernelGGL((k_walk<8, 0>), dim3(cus), dim3(1024), lds, 0, out, it2, 16u); }), 8);
  return 0;
}
