// cmdp_k1u.h -- K1U k_rollout_tmpl_stream + k_trace_hist: the shared-table rollout K1T with the visit counts taken OUT of
// the chain's workgroup.
//
// K1T is (chains resident per CU) / (latency of a chain's dependent LDS reads), and what caps the residency at 128 chains
// per CU is no longer the successor table (shared) but the 8-bit visit-count deltas: 930 of the 1 236 bytes an instance
// occupies at config C2.  A batch of 65 536 instances is 256 per CU, so K1T walks them in two rounds.  HBM meanwhile idles
// at 6 % of its bandwidth.  K1U streams the chain's 16-bit trace (arrival row under the action taken | reward code; the
// very entries K1T's counting wavefronts read from the LDS ring) to HBM instead -- 2 bytes per transition -- and a second
// kernel histograms it afterwards: an instance then needs its swap bits and the rings only (272 B at C2), ALL 256
// instances of a CU are resident in one round, and every chain sees a quarter of a workgroup's waves instead of an eighth.
//
// k_rollout_tmpl_stream -- roles of the sixteen wavefronts of a 1024-thread workgroup (wave w sits on SIMD w mod 4;
// quarter q = w & 3 owns instances 64 q .. 64 q + 63 of the group):
//   0..3    CHAIN of quarter q (raised priority)      -- K1T's chain, unchanged: bit-equal states, trace, episode logic
//   4..7    REWARDS of quarter q                      -- K1T's reward adder (sequential float64 sum in transition order)
//   8..11   DRAIN of quarter q: the chunk the chain traced one iteration ago, LDS ring -> HBM in 16-byte pieces of 8
//           transitions, layout trace[piece][instance] so that the 64 lanes of a store write 1 KB contiguously
//   12..15  PRODUCER of quarter q: the Philox action bytes of the chunk after the chain's
// k_trace_hist -- 64 instances per workgroup, lane = instance: reads the pieces back (1 KB coalesced per wave load), adds
// into 16-bit LDS counters (two per dword: the two actions of a state) with no-return LDS atomics, and flushes them into
// visits_sa / visits_s (+ the resets of the start state) with coalesced read-modify-writes.  A launch is cut into segments
// of at most 32 768 transitions so that no 16-bit counter can wrap.
// Results: the same counters, reward sums, states and Philox counters as K1T / K1P / K1 and the oracle, bit for bit.
#pragma once

#define K1U_THREADS 1024
#define K1U_FIXED (K1L_NRV * 8 + 16)   // rv2[K1L_NRV] f64 + pad
#define K1U_SEG 32736                  // transitions per segment (multiple of 24, < 65 536: 16-bit histogram counters)
// Trace pieces of 16 bytes: 8 entries of 16 bits (arrival row byte offset | reward code, as in the LDS ring), or -- PACK10,
// batches with at most 1024 rows -- 12 arrival ROW INDICES of 10 bits, three per dword: the histogram only needs the row,
// and it is bound by the bytes it reads (config C2: 2.6 GB instead of 3.9 GB per launch)
#define K1U_EPP(pack10) ((pack10) ? 12 : 8)
#define K1H_MLP 4                      // trace pieces in flight per lane

struct K1uPlan {
  int32_t G;             // instances per workgroup (<= 256)
  int32_t rows;          // S * A (A == 2)
  int32_t tmpl_bytes;
  int32_t mask_bytes;    // swap bits per instance in HBM (multiple of 4)
  int32_t slot_bytes;    // LDS bytes per instance: the swap bits, odd dword stride
  int32_t ch;            // transitions per ring chunk (multiple of 8; of 24 with pack10)
  int32_t pack10;        // trace pieces hold 12 ten-bit row indices instead of 8 sixteen-bit entries
  int32_t n_codes, code_shift;
  const uint16_t* tmpl;
  const uint8_t* swap_bits;
  const double* rvals;
  uint4* trace;          // [pieces of the segment][B] 8 trace entries each (one of two buffers when the histogram overlaps)
  int32_t* seg_resets;   // [B] episode resets of the segment (visits of the start state the histogram adds)
};

__host__ __device__ inline size_t k1u_lds_bytes(const K1uPlan& p, int g) {
  return (size_t)K1U_FIXED + (size_t)p.tmpl_bytes + (size_t)g * (size_t)(p.slot_bytes + 2 * K1P_ACT_STRIDE(p.ch) + 2 * K1P_TR_STRIDE(p.ch));
}
__host__ __device__ inline int k1h_stride_dwords(int S) { return S | 1; }   // one dword per state (two 16-bit counters), odd stride
__host__ __device__ inline size_t k1h_lds_bytes(int S, int G) { return (size_t)G * 4 * (size_t)k1h_stride_dwords(S); }

template <bool PACK10>
__global__ void __launch_bounds__(K1U_THREADS) k_rollout_tmpl_stream(EnvTables t, K1uPlan p, int64_t n_steps,
                                                                    double* __restrict__ reward_sum,
                                                                    int32_t* __restrict__ last_obs, int accumulate) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g0 = blockIdx.x * p.G;
  const int nb = min(p.G, t.B - g0);
  double* rv2 = reinterpret_cast<double*>(smem);
  unsigned char* tmpl = smem + K1U_FIXED;
  const int CH = p.ch;
  const int AS = K1P_ACT_STRIDE(CH), TS = K1P_TR_STRIDE(CH);
  unsigned char* ring = tmpl + p.tmpl_bytes;              // [2][G] action bytes, stride AS
  unsigned char* trace = ring + 2 * p.G * AS;             // [2][G] uint16 trace entries, stride TS bytes
  unsigned char* slots = trace + 2 * p.G * TS;            // [G] swap bits
  constexpr int A = 2;
  const int H = t.H;
  const int64_t so0 = t.state_off[g0];
  const int S = (int)(t.state_off[g0 + 1] - so0);
  const int rows = S * A;
  for (int i = tid; i < p.n_codes; i += K1U_THREADS) rv2[i] = p.rvals[i] * t.rscale - t.rmin;
  for (int i = tid; i < p.tmpl_bytes / 4; i += K1U_THREADS)
    reinterpret_cast<uint32_t*>(tmpl)[i] = (2 * i < rows) ? reinterpret_cast<const uint32_t*>(p.tmpl)[i] : 0u;
  const int mask_dwords = p.mask_bytes / 4;
  for (int j = tid; j < nb * mask_dwords; j += K1U_THREADS) {
    const int slot = j / mask_dwords, off = j - slot * mask_dwords;
    reinterpret_cast<uint32_t*>(slots + (size_t)slot * p.slot_bytes)[off] =
        reinterpret_cast<const uint32_t*>(p.swap_bits + (size_t)(g0 + slot) * p.mask_bytes)[off];
  }
  const int role = wave >> 2, quarter = wave & 3;
  const int li = quarter * 64 + lane;
  const bool owner = li < nb;
  const int b = g0 + (owner ? li : 0);
  unsigned char* base = slots + (size_t)(owner ? li : 0) * p.slot_bytes;
  typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
  typedef const __attribute__((address_space(3))) uint8_t* lds_u8;
  const lds_u32 tmpl_l = (lds_u32)(__attribute__((address_space(3))) unsigned char*)tmpl;
  const lds_u8 swp_l = (lds_u8)(__attribute__((address_space(3))) unsigned char*)base;
  const int smask_u = (1 << p.code_shift) - 1;
  const int32_t start_k = t.start_state[t.start_off[b]] * A * 2;
  int32_t cur = t.cur[b] * A * 2, h = t.hstep[b];
  int32_t n_resets = 0;
  const bool episodic = H > 0;
  const bool uniform_h = episodic && role == 0 && __all(!owner || h == __builtin_amdgcn_readfirstlane(h));
  double sum = (accumulate && role == 1 && owner && reward_sum) ? reward_sum[b] : 0.0;
  const uint2 my_key = t.philox_key[b];
  const unsigned long long my_ntr = t.n_trans[b];
  __syncthreads();

  auto produce = [&](int buf, int64_t first, int len) {   // every Philox block of the window, one producer per quarter
    if (!owner) return;
    const unsigned long long n0 = my_ntr + (unsigned long long)first;
    const int rel0 = (int)(n0 & 3ull);
    const unsigned long long q0 = n0 >> 2;
    const int nblk = (rel0 + len + 3) >> 2;
    unsigned char* dst = ring + ((size_t)buf * p.G + li) * AS;
    for (int qi = 0; qi < nblk; ++qi) {
      const unsigned long long q = q0 + (unsigned long long)qi;
      uint32_t act[4];
      philox_act4(q, my_key, 2, 1, act);   // A = 2: the packed stream, 128 one-bit actions per block
      const int pos0 = 4 * qi - rel0;
      if (rel0 == 0 && pos0 + 4 <= len) {
        *reinterpret_cast<uint32_t*>(dst + pos0) = act[0] | (act[1] << 8) | (act[2] << 16) | (act[3] << 24);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (pos0 + j >= 0 && pos0 + j < len) dst[pos0 + j] = (unsigned char)act[j];
      }
    }
  };

  auto chain_body = [&](int cb, int len) {   // K1T's chain (cmdp_k1t.h), verbatim in what it computes
    if (!owner || len <= 0) return;
    const unsigned char* acts = ring + ((size_t)cb * p.G + li) * AS;
    uint16_t* tr = reinterpret_cast<uint16_t*>(trace + ((size_t)cb * p.G + li) * TS);
    int hs = uniform_h ? __builtin_amdgcn_readfirstlane(h) : 0, nres_s = 0;
    int pend = 0;
    auto step = [&](int s, int a, auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      const uint32_t pair = tmpl_l[cur >> 2];
      const uint32_t mb = swp_l[cur >> 5];
      const uint32_t sh = ((mb >> ((cur >> 2) & 7)) << 4) + (uint32_t)(a << 4);
      const int nxt = (int)__builtin_amdgcn_ubfe(pair, sh, (uint32_t)p.code_shift);
      const int word = (int)((pair >> (sh & 16u)) & 0xffffu);
      tr[s - 1] = (uint16_t)pend;
      pend = word + 2 * a;
      if (MODE == 0) {
        cur = nxt;
      } else if (MODE == 1) {
        ++hs;
        const bool term = hs >= H;
        cur = term ? start_k : nxt;
        hs = term ? 0 : hs;
        nres_s += term ? 1 : 0;
      } else {
        ++h;
        const bool term = episodic && h >= H;
        cur = term ? start_k : nxt;
        h = term ? 0 : h;
        n_resets += term ? 1 : 0;
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    int s0 = 0;
    for (; s0 + 8 <= len; s0 += 8) {
      const uint32_t a_lo = *reinterpret_cast<const uint32_t*>(acts + s0);
      const uint32_t a_hi = *reinterpret_cast<const uint32_t*>(acts + s0 + 4);
#define K1U_ACT(j) (int)((((j) < 4 ? a_lo : a_hi) >> (8 * ((j) & 3))) & 0xffu)
      if (uniform_h && hs + 8 < H) {
        step(s0 + 0, K1U_ACT(0), T0{}); step(s0 + 1, K1U_ACT(1), T0{}); step(s0 + 2, K1U_ACT(2), T0{});
        step(s0 + 3, K1U_ACT(3), T0{}); step(s0 + 4, K1U_ACT(4), T0{}); step(s0 + 5, K1U_ACT(5), T0{});
        step(s0 + 6, K1U_ACT(6), T0{}); step(s0 + 7, K1U_ACT(7), T0{});
        hs += 8;
      } else if (uniform_h && H >= 8) {
        const int jstar = H - hs - 1;  // exactly one episode ends inside this group, after transition jstar
#define K1U_STEP_R(j)                         \
  step(s0 + (j), K1U_ACT(j), T0{});           \
  if (jstar == (j)) { cur = start_k; ++nres_s; }
        K1U_STEP_R(0) K1U_STEP_R(1) K1U_STEP_R(2) K1U_STEP_R(3) K1U_STEP_R(4) K1U_STEP_R(5) K1U_STEP_R(6) K1U_STEP_R(7)
#undef K1U_STEP_R
        hs = 7 - jstar;
      } else if (uniform_h) {
        step(s0 + 0, K1U_ACT(0), T1{}); step(s0 + 1, K1U_ACT(1), T1{}); step(s0 + 2, K1U_ACT(2), T1{});
        step(s0 + 3, K1U_ACT(3), T1{}); step(s0 + 4, K1U_ACT(4), T1{}); step(s0 + 5, K1U_ACT(5), T1{});
        step(s0 + 6, K1U_ACT(6), T1{}); step(s0 + 7, K1U_ACT(7), T1{});
      } else {
        step(s0 + 0, K1U_ACT(0), T2{}); step(s0 + 1, K1U_ACT(1), T2{}); step(s0 + 2, K1U_ACT(2), T2{});
        step(s0 + 3, K1U_ACT(3), T2{}); step(s0 + 4, K1U_ACT(4), T2{}); step(s0 + 5, K1U_ACT(5), T2{});
        step(s0 + 6, K1U_ACT(6), T2{}); step(s0 + 7, K1U_ACT(7), T2{});
      }
#undef K1U_ACT
    }
    for (; s0 < len; ++s0) {
      const int a = acts[s0];
      if (uniform_h) step(s0, a, T1{}); else step(s0, a, T2{});
    }
    tr[len - 1] = (uint16_t)pend;
    if (uniform_h) { h = hs; n_resets += nres_s; }
  };

  // drain: the chunk traced into buffer `tb` (transitions [first, first + plen) of the segment) goes to HBM
  auto drain_body = [&](int tb, int64_t first, int plen) {
    if (!owner || plen <= 0) return;
    const uint32_t* trb = reinterpret_cast<const uint32_t*>(trace + ((size_t)tb * p.G + li) * TS);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int EPP = K1U_EPP(PACK10);
    u32x4* dst = reinterpret_cast<u32x4*>(p.trace) + (size_t)(first / EPP) * (size_t)t.B + (size_t)b;
    const int npiece = (plen + EPP - 1) / EPP;   // a ragged last piece carries stale entries past plen: the histogram stops at n_steps
    const uint32_t sm = (uint32_t)smask_u;
    for (int j = 0; j < npiece; ++j) {
      u32x4 v;
      if (PACK10) {
        uint32_t e[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) e[k] = trb[6 * j + k];
        auto row = [&](int k) { return ((e[k >> 1] >> (16 * (k & 1))) & sm) >> 1; };
        v.x = row(0) | (row(1) << 10) | (row(2) << 20);
        v.y = row(3) | (row(4) << 10) | (row(5) << 20);
        v.z = row(6) | (row(7) << 10) | (row(8) << 20);
        v.w = row(9) | (row(10) << 10) | (row(11) << 20);
      } else {
        v.x = trb[4 * j]; v.y = trb[4 * j + 1]; v.z = trb[4 * j + 2]; v.w = trb[4 * j + 3];
      }
      __builtin_nontemporal_store(v, &dst[(size_t)j * t.B]);   // written once, read once by the histogram: no reuse
    }
  };

  auto rewards_body = [&](int tb, int plen) {
    if (!owner || plen <= 0) return;
    const unsigned char* trb = trace + ((size_t)tb * p.G + li) * TS;
    int s0 = 0;
    for (; s0 + 4 <= plen; s0 += 4) {
      const uint32_t e0 = *reinterpret_cast<const uint32_t*>(trb + 2 * s0);
      const uint32_t e1 = *reinterpret_cast<const uint32_t*>(trb + 2 * s0 + 4);
      const double r0 = rv2[(e0 & 0xffffu) >> p.code_shift], r1 = rv2[e0 >> (16 + p.code_shift)];
      const double r2 = rv2[(e1 & 0xffffu) >> p.code_shift], r3 = rv2[e1 >> (16 + p.code_shift)];
      sum += r0; sum += r1; sum += r2; sum += r3;
    }
    for (; s0 < plen; ++s0) sum += rv2[reinterpret_cast<const uint16_t*>(trb)[s0] >> p.code_shift];
  };

  if (role == 0) __builtin_amdgcn_s_setprio(3);
  if (n_steps > 0 && role == 3) produce(0, 0, (int)min((int64_t)CH, n_steps));
  __syncthreads();
  int cb = 0, len = 0, plen = 0;
  int64_t left = n_steps, done = 0;   // done: transitions before the chunk the bookkeepers take this iteration
  for (;;) {
    done += plen;
    plen = len;
    len = (int)min((int64_t)CH, left);
    left -= len;
    // chunk `cb` is walked while the bookkeepers take the chunk before it (buffer cb ^ 1) and the producers fill the one after
    if (role == 0) chain_body(cb, len);
    else if (role == 1) rewards_body(cb ^ 1, plen);
    else if (role == 2) drain_body(cb ^ 1, done, plen);
    else if (left > 0) produce(cb ^ 1, n_steps - left, (int)min((int64_t)CH, left));
    __syncthreads();
    cb ^= 1;
    if (len == 0) break;
  }
  if (role == 0 && owner) {
    cur /= 2 * A;
    t.cur[b] = cur;
    t.hstep[b] = h;
    t.n_trans[b] = my_ntr + (unsigned long long)n_steps;
    t.n_reset[b] += (unsigned long long)n_resets;
    p.seg_resets[b] = n_resets;
    if (last_obs) last_obs[b] = cur;
  }
  if (role == 1 && owner && reward_sum) reward_sum[b] = sum;
}

// Histogram of a segment's trace: arrival row r = (entry & smask) >> 1 of every transition of the group's G instances.
// G = 64: lane = instance (119 KB of counters at C2: the kernel has the CU to itself).  G = 32: two lanes per instance on
// alternating pieces and 60 KB of counters, so that a workgroup fits NEXT TO a resident k_rollout_tmpl_stream workgroup
// (74 KB at chunk 32) and the histogram of one step runs under the chain of the next (second stream).
template <int G, int THREADS, bool PACK10>
__global__ void __launch_bounds__(THREADS) k_trace_hist(EnvTables t, const uint4* __restrict__ trace,
                                                       const int32_t* __restrict__ seg_resets, int64_t n_steps,
                                                       int code_shift) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
  __shared__ int32_t start_of[G], resets_of[G];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = THREADS / 64, SUB = 64 / G;
  const int g0 = blockIdx.x * G;
  const int nb = min(G, t.B - g0);
  const int64_t so0 = t.state_off[g0];
  const int S = (int)(t.state_off[g0 + 1] - so0);
  const int stride = k1h_stride_dwords(S);
  for (int i = tid; i < G * stride; i += THREADS) cnt[i] = 0u;
  if (tid < nb) {
    start_of[tid] = t.start_state[t.start_off[g0 + tid]];
    resets_of[tid] = seg_resets[g0 + tid];
  }
  __syncthreads();
  const uint32_t smask = (1u << code_shift) - 1u;
  constexpr int EPP = K1U_EPP(PACK10);
  const int64_t npiece = (n_steps + EPP - 1) / EPP;
  const int inst = lane % G, sub = lane / G;
  if (inst < nb) {
    uint32_t* mine = cnt + (size_t)inst * stride;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* src = reinterpret_cast<const u32x4*>(trace) + (size_t)g0 + inst;
    auto bump_row = [&](uint32_t r) { atomicAdd(&mine[r >> 1], 1u << (16 * (r & 1))); };
    auto add = [&](uint32_t w, int64_t tpos) {   // the two (sixteen-bit) or three (ten-bit) entries of a dword
      if (PACK10) {
        if (tpos < n_steps) bump_row(w & 1023u);
        if (tpos + 1 < n_steps) bump_row((w >> 10) & 1023u);
        if (tpos + 2 < n_steps) bump_row((w >> 20) & 1023u);
      } else {
        if (tpos < n_steps) bump_row((w & smask) >> 1);
        if (tpos + 1 < n_steps) bump_row(((w >> 16) & smask) >> 1);
      }
    };
    // K1H_MLP pieces in flight per lane: the loads ARE the kernel (up to 1 KB per wave load; without several loads
    // outstanding per wave the launch is bound by HBM latency, not bandwidth)
    constexpr int STEP = NW * SUB;
    for (int64_t pc = wave * SUB + sub; pc < npiece; pc += (int64_t)K1H_MLP * STEP) {
      u32x4 v[K1H_MLP];
#pragma unroll
      for (int k = 0; k < K1H_MLP; ++k) {
        const int64_t q = pc + (int64_t)k * STEP;
        v[k] = q < npiece ? __builtin_nontemporal_load(&src[(size_t)q * t.B]) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < K1H_MLP; ++k) {
        const int64_t q = pc + (int64_t)k * STEP;
        if (q < npiece) {
          constexpr int EPD = EPP / 4;   // entries per dword
          const int64_t t0 = q * EPP;
          add(v[k].x, t0); add(v[k].y, t0 + EPD); add(v[k].z, t0 + 2 * EPD); add(v[k].w, t0 + 3 * EPD);
        }
      }
    }
  }
  __syncthreads();
  // flush: dword j of instance i = counts of rows (2 j, 2 j + 1) = the two actions of state j
  const int total = nb * S;
  for (int k = tid; k < total; k += THREADS) {
    const int i = k / S, j = k - i * S;
    const uint32_t c = cnt[(size_t)i * stride + j];
    const uint32_t c0 = c & 0xffffu, c1 = c >> 16;
    int32_t add_s = (int32_t)(c0 + c1);
    if (j == start_of[i]) add_s += resets_of[i];
    if (add_s) t.visits_s[so0 + k] += add_s;
    if (c) {
      int2* sa = reinterpret_cast<int2*>(t.visits_sa + (so0 + k) * 2);
      int2 v = *sa;
      v.x += (int32_t)c0;
      v.y += (int32_t)c1;
      *sa = v;
    }
  }
}
