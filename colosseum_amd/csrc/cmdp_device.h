// cmdp_device.h -- device-side primitives shared by the kernels of libcmdp (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMDP_WAVE 64

// ---------------------------------------------------------------------------------------------------
// Philox-4x32-10.  Counter (n_lo, n_hi, domain, 0), key = the instance's 64-bit key.
// domain 0, n = transition index      : w0,w1 = 53-bit transition uniform (stochastic rows only)
// domain 1, n = reset index           : w0,w1 = start-state uniform (several start states only)
// domain 2, random-policy actions (build-defined stream; round 4: every bit of a block is used when A allows it)
//           A in {2, 4, 16, 256} (lg = log2 A divides 32): PACKED -- a block holds apb = 128 / lg actions;
//           transition n -> block n / apb, k = n % apb, a = (word[k / (32 / lg)] >> (lg * (k % (32 / lg)))) & (A - 1)
//           (A = 2: 128 one-bit actions per block, least significant bit of word 0 first);
//           any other A: block n >> 2, word (n & 3), a = (word * A) >> 32 (four actions per block).
// domain 3, n = transition index, c3 = draw counter : Beta reward of the transition (philox_beta below)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&w)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // written as 64-bit products so that each becomes ONE v_mad_u64_u32 instead of a mul_hi + mul_lo pair (the
    // quarter-rate integer multiplies are what a Philox block costs)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}

// log2 A when the random-policy stream is packed (A in {2, 4, 16, 256}), else 0
__host__ __device__ __forceinline__ int philox_act_lg(int A) { return A == 2 ? 1 : A == 4 ? 2 : A == 16 ? 4 : A == 256 ? 8 : 0; }

// The four random-policy actions of transitions 4 g .. 4 g + 3 (g = the absolute group of four: every producer of the
// chain kernels works in these groups).  Packed stream: the four sit side by side in one word of block g / (32 / lg).
__device__ __forceinline__ void philox_act4(unsigned long long g, uint2 key, int A, int lg, uint32_t (&act)[4]) {
  uint32_t w[4];
  if (lg) {
    const int gpb_sh = lg == 1 ? 5 : lg == 2 ? 4 : lg == 4 ? 3 : 2;   // groups per block = 32 / lg
    const unsigned long long q = g >> gpb_sh;
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), 2u, 0u, key.x, key.y, w);
    const int gi = (int)(g & ((1ull << gpb_sh) - 1));   // group inside the block
    const int wi = gi >> (gpb_sh - 2);                  // its word: a word holds 32 / lg actions = (32 / lg) / 4 groups
    const uint32_t word = wi == 0 ? w[0] : wi == 1 ? w[1] : wi == 2 ? w[2] : w[3];
    const int off = (gi & ((1 << (gpb_sh - 2)) - 1)) * 4 * lg;
    const uint32_t m = (uint32_t)A - 1u;
#pragma unroll
    for (int j = 0; j < 4; ++j) act[j] = (word >> (off + j * lg)) & m;
  } else {
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 2u, 0u, key.x, key.y, w);
#pragma unroll
    for (int j = 0; j < 4; ++j) act[j] = (uint32_t)(((uint64_t)w[j] * (uint64_t)A) >> 32);
  }
}

__device__ __forceinline__ double u53(uint32_t w0, uint32_t w1) {
  return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------------------------------
// MT19937, one word at a time (identical output sequence to the block "twist" form): CPython's
// random.Random stream of one NextStateSampler (reference colosseum/mdp/utils/custom_samplers.py:52).
// State: 624 words + position, position starts at 0 after seeding.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_next_word(uint32_t* __restrict__ mt, int& pos) {
  const int p = pos;
  const int p1 = (p == 623) ? 0 : p + 1;
  const int pm = (p >= 227) ? p - 227 : p + 397;
  const uint32_t y = (mt[p] & 0x80000000u) | (mt[p1] & 0x7fffffffu);
  uint32_t v = mt[pm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  mt[p] = v;
  pos = p1;
  v ^= v >> 11;
  v ^= (v << 7) & 0x9d2c5680u;
  v ^= (v << 15) & 0xefc60000u;
  v ^= v >> 18;
  return v;
}

__device__ __forceinline__ double mt_random(uint32_t* __restrict__ mt, int32_t* __restrict__ pos_ptr) {
  int pos = *pos_ptr;
  const uint32_t a = mt_next_word(mt, pos) >> 5;
  const uint32_t b = mt_next_word(mt, pos) >> 6;
  *pos_ptr = pos;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// The same draw with the loads of BOTH words issued before either twist: word k + 1 reads mt[p1], mt[p1 + 1] and
// mt[p1 +- 397 / 227], none of which is the one entry (mt[p]) word k writes, so the six loads are independent -- one memory
// round trip instead of two for a stream that lives in HBM (the agent kernels: one lane per instance, every access a
// dependent round trip).
__device__ __forceinline__ double mt_random_pair(uint32_t* __restrict__ mt, int32_t* __restrict__ pos_ptr) {
  const int p = *pos_ptr;
  const int p1 = (p == 623) ? 0 : p + 1;
  const int p2 = (p1 == 623) ? 0 : p1 + 1;
  const int pm = (p >= 227) ? p - 227 : p + 397;
  const int pm1 = (p1 >= 227) ? p1 - 227 : p1 + 397;
  const uint32_t m0 = mt[p], m1 = mt[p1], m2 = mt[p2], f0 = mt[pm], f1 = mt[pm1];
  const uint32_t y0 = (m0 & 0x80000000u) | (m1 & 0x7fffffffu);
  uint32_t v0 = f0 ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u);
  // word k + 1's "far" entry is the NEW value of mt[p] exactly when pm1 == p (p1 - 227 == p never holds; p1 + 397 == p never
  // holds either), so f1 is always the old value read above
  const uint32_t y1 = (m1 & 0x80000000u) | (m2 & 0x7fffffffu);
  uint32_t v1 = f1 ^ (y1 >> 1) ^ ((y1 & 1u) ? 0x9908b0dfu : 0u);
  mt[p] = v0;
  mt[p1] = v1;
  *pos_ptr = p2;
  v0 ^= v0 >> 11; v0 ^= (v0 << 7) & 0x9d2c5680u; v0 ^= (v0 << 15) & 0xefc60000u; v0 ^= v0 >> 18;
  v1 ^= v1 >> 11; v1 ^= (v1 << 7) & 0x9d2c5680u; v1 ^= (v1 << 15) & 0xefc60000u; v1 ^= v1 >> 18;
  return ((double)(v0 >> 5) * 67108864.0 + (double)(v1 >> 6)) * (1.0 / 9007199254740992.0);
}

// random.Random(seed) seeding == init_by_array([seed])
__device__ inline void mt_seed_python_int(uint32_t* __restrict__ mt, uint32_t seed) {
  mt[0] = 19650218u;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int i = 1;
  for (int k = 624; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + seed;  // + j with j == 0 (key length 1)
    if (++i >= 624) { mt[0] = mt[623]; i = 1; }
  }
  for (int k = 623; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    if (++i >= 624) { mt[0] = mt[623]; i = 1; }
  }
  mt[0] = 0x80000000u;
}

// `random.choices` index: bisect_right(cum, u*total, 0, n-1) == #{ i < n-1 : cum[i] <= u*total }.
// Loads are independent (linear count), n is small (<= 1 + lazy + (A-1)*k successors).
__device__ __forceinline__ int choose_index(const double* __restrict__ cum, int n, double u01) {
  const double x = u01 * (cum[n - 1] + 0.0);
  int idx = 0;
  for (int i = 0; i < n - 1; ++i) idx += (cum[i] <= x) ? 1 : 0;
  return idx;
}

// wave-wide reductions (64 lanes)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// max over the wave with DPP moves (VALU only: no LDS traffic, no lgkmcnt waits); the result is valid in LANE 63.
// quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8, row_bcast:15, row_bcast:31; lanes a move does not reach
// keep their own value (old = src), which is harmless for an idempotent operation.
__device__ __forceinline__ float wave_max_lane63(float v) {
  int x = __float_as_int(v);
#define CMDP_DPP_MAX(ctrl) \
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false))))
  CMDP_DPP_MAX(0xb1);
  CMDP_DPP_MAX(0x4e);
  CMDP_DPP_MAX(0x124);
  CMDP_DPP_MAX(0x128);
  CMDP_DPP_MAX(0x142);
  CMDP_DPP_MAX(0x143);
#undef CMDP_DPP_MAX
  return __int_as_float(x);
}
// The same reduction with the DPP modifier on the max itself (one VALU instruction per step plus the two wait states a
// DPP read of a just-written VGPR needs) and no NaN canonicalisation: for values that are never NaN.
__device__ __forceinline__ float wave_max_lane63_nn(float v) {
#define CMDP_DPP_MAX_NN(ctrl) \
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
  CMDP_DPP_MAX_NN("quad_perm:[1,0,3,2]");
  CMDP_DPP_MAX_NN("quad_perm:[2,3,0,1]");
  CMDP_DPP_MAX_NN("row_ror:4");
  CMDP_DPP_MAX_NN("row_ror:8");
  CMDP_DPP_MAX_NN("row_bcast:15");
  CMDP_DPP_MAX_NN("row_bcast:31");
#undef CMDP_DPP_MAX_NN
  return v;
}
// max / max3 / max(a, |b|) of values that are never NaN: the bare instructions (fmaxf adds a canonicalising
// v_max_f32 x, x per operand that may be a signalling NaN)
__device__ __forceinline__ float fmax_nn(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fmax3_nn(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fmax_abs_nn(float a, float b) {  // max(a, |b|)
  float r;
  asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------
// Beta(a, b) reward of transition n in CMDP_RNG_PHILOX mode (build-defined; the reference-exact path is the
// host sampler of colosseum_amd/mdp/reward_sampler.py).  a == 1 or b == 1: inverse CDF of the first uniform of block
// (n, domain 3, draw 0).  Otherwise X = Ga / (Ga + Gb) with Marsaglia-Tsang gamma variates;
// every Philox block (n, domain 3, draw k) supplies the two uniforms of one Box-Muller normal and, in its second
// half, the acceptance uniform; shapes below one use Gamma(shape + 1) * U^(1/shape).  The CPU oracle runs the
// same recipe with libm, so the two agree to rounding of log/sqrt/cos/pow/log1p/expm1/exp (tests compare with rtol 1e-12).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double philox_gamma(double shape, unsigned long long n, uint2 key, uint32_t& draw,
                                               uint32_t domain = 3u) {
  double boost = 1.0;
  uint32_t w[4];
  if (shape < 1.0) {
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), domain, draw++, key.x, key.y, w);
    boost = pow(1.0 - u53(w[0], w[1]), 1.0 / shape);
    shape += 1.0;
  }
  const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (int attempt = 0; attempt < 64; ++attempt) {
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), domain, draw++, key.x, key.y, w);
    const double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    const double z = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586476925286766559 * u2);
    double v = 1.0 + c * z;
    if (v <= 0.0) continue;
    v = v * v * v;
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), domain, draw++, key.x, key.y, w);
    const double u3 = u53(w[0], w[1]);
    if (log(u3) < 0.5 * z * z + d - d * v + d * log(v)) return boost * d * v;
  }
  return boost * d;  // unreachable in practice (acceptance > 95 % per attempt)
}

__device__ __forceinline__ double philox_beta(double a, double b, unsigned long long n, uint2 key, int gammas_only) {
  // one shape equal to one (nearly every triple of the reference's MDP families: Beta(1, b) away from the goal, Beta(a, 1)
  // at it): the inverse CDF, x = 1 - (1 - u)^(1/b) resp. (1 - u)^(1/a), from ONE uniform -- an exact sampler at a tenth of
  // the instructions of two rejection-sampled gammas (the agent kernels run one lane per instance: the sampler WAS their
  // step time)
  if (!gammas_only && (a == 1.0 || b == 1.0)) {
    uint32_t w[4];
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 3u, 0u, key.x, key.y, w);
    const double l = log1p(-u53(w[0], w[1]));  // log(1 - u), u in [0, 1)
    return a == 1.0 ? -expm1(l / b) : exp(l / a);
  }
  uint32_t draw = 0;
  const double ga = philox_gamma(a, n, key, draw);
  const double gb = philox_gamma(b, n, key, draw);
  return ga / (ga + gb);
}
