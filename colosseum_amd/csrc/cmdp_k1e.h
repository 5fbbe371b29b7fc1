// cmdp_k1e.h -- K1E k_rollout_epi + k_reward_scan: the EPISODE-PARALLEL random-policy rollout (round 4; config C2's kernel).
//
// Every rollout kernel before this one (K1 .. K1U) walks ONE dependent chain per instance, so a launch is
// (resident chains) x (transitions) x (latency of a chain's dependent LDS read) and the chip idles behind that latency
// (K1U: VALU issue 46 %, LDS pipe 36 %, HBM 16 %).  Under the random policy the episodes of an episodic instance are
// INDEPENDENT of each other:
//   * reset() returns to a start state that depends on nothing the episode did (reference colosseum/mdp/base.py:1268-1277;
//     eligible batches have one start state),
//   * an episode lasts exactly H steps (`h >= H`, base.py:1310-1317),
//   * the action of transition n is a pure function of (instance key, n): the counter-based Philox stream of domain 2
//     (reference RandomActor, colosseum/agent/actors/random.py:34-47, is a seeded stream as well).
// So transition n0 + e H + j of an instance can be walked without its predecessors: lane = (instance, episode), a chain is
// H steps long, and a launch of 65 536 x 30 000 transitions is 65.5 M independent chains of 30 instead of 65 536 of 30 000.
// The kernel is then bound by instruction and LDS THROUGHPUT, not by latency.
//
// k_rollout_epi -- one 1024-thread workgroup owns NI = 32 instances for the whole segment:
//   LDS   per instance a PRIVATE table of S x 2 dwords {successor word : 16 | visit count : 16} in a slot of 2^k bytes, so
//         that the address of the next read is ONE v_and_or_b32 of the word just read: (word & MASK) | (action << 2 | slot).
//         The successor word is the byte offset of the successor's row pair inside the slot with the reward code in its two
//         low bits.  The count of the ARRIVAL row under the action taken (base.py:1302-1303) is a no-return ds_add_u32 of
//         0x10000 on the dword at that same address pattern.  States are stored at s ^ (i & 15) and actions at a ^ (i >> 4)
//         (i = the instance's slot): the 32 lanes of an LDS lane group are 32 DIFFERENT instances -- no two lanes of a group
//         ever add to the same dword -- and while they all sit at the same logical row (the first steps of an episode) they
//         hit 32 different banks.
//   lanes lane = (instance l & 31, half l >> 5); pass p gives every instance its episodes 64 p .. 64 p + 63: wavefront w,
//         half s, chain c in {0, 1} walks episode 64 p + 4 w + 2 s + c -- two independent chains per lane, interleaved.
//   bits  the action bits of a pass (64 H per instance) come from a ring of Philox blocks in LDS that all wavefronts fill
//         for pass p + 1 before they walk pass p (one barrier per pass); a chain fetches its 32 bits with one funnel shift.
//         Every bit of every block is used (cmdp_device.h: the packed domain-2 stream) -- 0.8 VALU instructions per
//         transition instead of the 25 a whole block per four one-bit actions cost K1U.
//   step  v_bfe (action bit) . v_lshl_or (| slot) . v_and_or (address) . ds_read_u16 . v_and_or . ds_add_u32 . v_alignbit
//         (reward code into the episode's code word): 5 VALU + 2 LDS instructions, none of them waiting on another chain.
//   out   the reward codes of an episode (2 bits per step) go to HBM, 8 bytes per (episode, 32-step chunk), layout
//         codes[episode][chunk][instance]; the counts are flushed into visits_sa / visits_s by coalesced read-modify-writes
//         (+ the resets of the start state); state, in-episode time and the Philox counters are advanced as if the
//         transitions had been taken one by one.
// k_reward_scan -- lane = instance: the float64 reward sum in TRANSITION ORDER from the code words (sum += value[code], one
//         add per transition, sequential: bit-equal to the oracle's and every other kernel's sum).
// Results: visit counts, final states, in-episode times, Philox counters and reward sums bit-equal to K1 / K1T / K1U and the
// CPU oracle (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py, tools/stress_k1t.py k1e, tools/fuzz_parity.py).
#pragma once

#define K1E_THREADS 1024
#define K1E_NW (K1E_THREADS / 64)
#define K1E_NI 32                          // instances per workgroup: the 32 lanes of an LDS lane group
#define K1E_EPL 4                          // chains (episodes) per lane
#define K1E_EPP (K1E_NW * 2 * K1E_EPL)     // episodes of one instance per pass (128)
#define K1E_SEG 61440                      // transitions per segment (< 65 536: the 16-bit counts in the table dwords)
#define K1E_DUMMY 16                       // bytes behind the slots: the self-looping dummy row idle chains walk

struct K1ePlan {
  int32_t S, H;
  int32_t slot_bytes;        // power of two >= roundup16(S) * 8
  int32_t ring_blocks;       // power of two: Philox blocks of an instance's action-bit ring
  int32_t n_codes;           // <= 4 distinct reward values (2-bit codes)
  int32_t nch;               // 32-step chunks per episode, ceil(H / 32)
  int32_t n_pass;            // passes of this segment (uniform over the workgroups)
  const uint32_t* etab;      // [B][S] one dword per state: successor words of action 0 (low) and 1 (high), s' * 8 | code
  const double* rvals;       // [n_codes]
  uint2* codes;              // [episode][chunk][B]
  int32_t* seg_h0;           // [B] in-episode time at the start of the segment (k_reward_scan decodes the episodes with it)
};

__host__ __device__ inline size_t k1e_lds_bytes(const K1ePlan& p) {
  return (size_t)K1E_NI * (size_t)p.slot_bytes + K1E_DUMMY + (size_t)K1E_NI * (size_t)p.ring_blocks * 16 + 8 * K1E_NI;
}
__host__ __device__ inline int64_t k1e_max_episodes(int64_t n_steps, int H) { return (n_steps + 2 * (int64_t)H - 2) / H; }

typedef __attribute__((address_space(3))) uint32_t* k1e_lds_u32;
typedef const __attribute__((address_space(3))) uint16_t* k1e_lds_cu16;
typedef const __attribute__((address_space(3))) uint32_t* k1e_lds_cu32;

__global__ void __launch_bounds__(K1E_THREADS) k_rollout_epi(EnvTables t, K1ePlan p, int64_t n_steps,
                                                            int32_t* __restrict__ last_obs) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int inst = lane & (K1E_NI - 1), sub = lane >> 5;
  const int g0 = blockIdx.x * K1E_NI;
  const int nb = min(K1E_NI, t.B - g0);
  const bool owner = inst < nb;
  const int b = g0 + (owner ? inst : 0);
  const int S = p.S, H = p.H;
  const uint32_t slot = (uint32_t)p.slot_bytes;
  const uint32_t MASK = slot - 8u;                       // the state field of a successor word (byte offset of the row pair)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // 0: no static LDS here
  const uint32_t dummy = lds0 + K1E_NI * slot;           // aligned like a slot: (word & MASK) | x works on it too
  const uint32_t RD = (uint32_t)p.ring_blocks * 4u;      // ring dwords per instance
  uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
  uint32_t* ring = reinterpret_cast<uint32_t*>(smem + (size_t)K1E_NI * slot + K1E_DUMMY);
  int32_t* meta = reinterpret_cast<int32_t*>(ring + (size_t)K1E_NI * RD);   // [NI] episode resets of the segment, [NI] start states
  const uint32_t swz = (uint32_t)inst & 15u, fa = (uint32_t)inst >> 4;
  const uint32_t ibase = lds0 + (uint32_t)inst * slot;

  // ---- stage: private tables {successor word | count 0}, states at s ^ swz, actions at a ^ fa, words re-based likewise ----
  const int64_t so0 = t.state_off[g0];
  for (int k = tid; k < nb * S; k += K1E_THREADS) {
    const int i = k / S, s = k - i * S;
    const uint32_t pair = p.etab[(size_t)(g0 + i) * S + s];
    const uint32_t sw = (uint32_t)i & 15u, f = (uint32_t)i >> 4;
    const uint32_t w0 = (pair & 0xffffu) ^ (sw << 3), w1 = (pair >> 16) ^ (sw << 3);
    uint32_t* row = tab + (size_t)i * (slot / 4) + (((uint32_t)s ^ sw) << 1);
    row[f] = w0;
    row[f ^ 1u] = w1;
  }
  if (tid < K1E_DUMMY / 4) tab[(size_t)K1E_NI * (slot / 4) + tid] = 0u;
  if (tid < nb) {
    meta[tid] = (int32_t)(((int64_t)t.hstep[g0 + tid] + n_steps) / H);
    meta[K1E_NI + tid] = t.start_state[t.start_off[g0 + tid]];
  }

  const uint2 key = t.philox_key[b];
  const unsigned long long ntr = t.n_trans[b];
  const int h0 = t.hstep[b];
  const int32_t cur0 = t.cur[b];
  const int32_t start = t.start_state[t.start_off[b]];
  const uint32_t ring_i = (uint32_t)inst * RD;   // dword index of the instance's ring

  // last Philox block (index) needed by the episodes of passes 0 .. q; the blocks of pass q are (lastblk(q-1), lastblk(q)]
  auto lastblk = [&](int q) -> long long {
    const int64_t e_end = (int64_t)(q + 1) * K1E_EPP;               // first episode behind pass q
    int64_t end = e_end * H - h0;                                    // its first transition
    if (end > n_steps) end = n_steps;
    return (long long)((ntr + (unsigned long long)end - 1ull) >> 7);   // n_steps >= 1
  };
  auto produce = [&](int q) {
    if (!owner) return;
    const long long lo = q == 0 ? (long long)(ntr >> 7) : lastblk(q - 1) + 1, hi = lastblk(q);
    for (long long blk = lo + (tid >> 5); blk <= hi; blk += K1E_THREADS / 32) {
      uint32_t w[4];
      philox4x32_10((uint32_t)blk, (uint32_t)((unsigned long long)blk >> 32), 2u, 0u, key.x, key.y, w);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v;
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
      *reinterpret_cast<u32x4*>(ring + ring_i + (((uint32_t)blk & ((uint32_t)p.ring_blocks - 1u)) << 2)) = v;
    }
  };
  // the 32 action bits from absolute transition `a0` on (already in the slot's action polarity)
  auto fetch_bits = [&](unsigned long long a0) -> uint32_t {
    const uint32_t di = (uint32_t)(a0 >> 5) & (RD - 1u);
    const uint32_t d0 = ring[ring_i + di], d1 = ring[ring_i + ((di + 1u) & (RD - 1u))];
    const uint32_t bits = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)a0 & 31u);
    return fa ? ~bits : bits;
  };

  for (int pass = 0; pass < p.n_pass; ++pass) {
    // the action bits of this pass (the ring holds ONE pass: fill, barrier, walk, barrier -- the fill is one Philox block
    // per thread, the same for every wavefront)
    produce(pass);
    __syncthreads();
    // ---- the lane's chains of this pass ----
    int64_t first[K1E_EPL];
    int hs[K1E_EPL], len[K1E_EPL];
    bool valid[K1E_EPL], full[K1E_EPL];
    uint32_t w[K1E_EPL], x0[K1E_EPL];
    const int64_t e0 = (int64_t)pass * K1E_EPP + (int64_t)((wave * 2 + sub) * K1E_EPL);
#pragma unroll
    for (int c = 0; c < K1E_EPL; ++c) {
      const int64_t e = e0 + c;
      first[c] = e == 0 ? 0 : e * H - h0;
      valid[c] = owner && first[c] < n_steps;
      hs[c] = e == 0 ? h0 : 0;
      const int64_t room = n_steps - first[c];
      len[c] = valid[c] ? (int)min((int64_t)(H - hs[c]), room) : 0;
      full[c] = valid[c] && len[c] == H;
      const uint32_t s_from = (uint32_t)(e == 0 ? cur0 : start);
      // a chain that is not walked at full length in the fast loop idles on the dummy row (word 0 -> itself)
      w[c] = full[c] ? ((s_from ^ swz) << 3) : 0u;
      x0[c] = full[c] ? ibase : dummy;
    }
    bool any_full = false;
#pragma unroll
    for (int c = 0; c < K1E_EPL; ++c) any_full |= full[c];

    // ---- fast loop: full episodes, uniform trip counts; lanes without any full chain are masked off as a whole ----
    if (any_full) {
      for (int ch = 0; ch < p.nch; ++ch) {
        const int L = min(32, H - 32 * ch);
        uint32_t bits[K1E_EPL], clo[K1E_EPL], chi[K1E_EPL];
#pragma unroll
        for (int c = 0; c < K1E_EPL; ++c) {
          bits[c] = fetch_bits(ntr + (unsigned long long)(first[c] + 32 * ch));
          clo[c] = 0u; chi[c] = 0u;
        }
        // one step of all the lane's chains: the reads of all chains are issued before the first is waited for
        auto steps = [&](int j, uint32_t (&cw)[K1E_EPL]) {
          uint32_t x[K1E_EPL], ra[K1E_EPL];
#pragma unroll
          for (int c = 0; c < K1E_EPL; ++c) {
            const uint32_t a = __builtin_amdgcn_ubfe(bits[c], (uint32_t)j, 1u);
            x[c] = (a << 2) | x0[c];
            ra[c] = (w[c] & MASK) | x[c];
          }
#pragma unroll
          for (int c = 0; c < K1E_EPL; ++c) w[c] = (uint32_t)*(k1e_lds_cu16)(uintptr_t)ra[c];
#pragma unroll
          for (int c = 0; c < K1E_EPL; ++c) {
            const uint32_t ca = (w[c] & MASK) | x[c];   // arrival row under the action taken (base.py:1302-1303)
            (void)__hip_atomic_fetch_add((k1e_lds_u32)(uintptr_t)ca, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            cw[c] = __builtin_amdgcn_alignbit(w[c], cw[c], 2u);
          }
        };
        const int L0 = min(L, 16);
#pragma unroll 2
        for (int j = 0; j < L0; ++j) steps(j, clo);
#pragma unroll 2
        for (int j = 16; j < L; ++j) steps(j, chi);
#pragma unroll
        for (int c = 0; c < K1E_EPL; ++c) {
          if (full[c]) {
            const uint32_t lo = L0 < 16 ? clo[c] >> (32 - 2 * L0) : clo[c];
            const uint32_t hi = L > 16 ? (L < 32 ? chi[c] >> (32 - 2 * (L - 16)) : chi[c]) : 0u;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 v;
            v.x = lo; v.y = hi;
            __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p.codes) + ((size_t)(e0 + c) * p.nch + ch) * (size_t)t.B + b);
          }
        }
      }
    }
    // ---- slow loop: the partial episodes at the two ends of the segment (per-lane lengths) ----
#pragma unroll
    for (int c = 0; c < K1E_EPL; ++c) {
      if (valid[c] && !full[c]) {
        uint32_t ws = ((uint32_t)((e0 + c) == 0 ? cur0 : start) ^ swz) << 3;
        for (int j0 = 0; j0 < len[c]; j0 += 32) {
          const int L = min(32, len[c] - j0);
          const uint32_t bits = fetch_bits(ntr + (unsigned long long)(first[c] + j0));
          uint32_t lo = 0u, hi = 0u;
          for (int j = 0; j < L; ++j) {
            const uint32_t a = (bits >> j) & 1u;
            const uint32_t x = (a << 2) | ibase;
            const uint32_t ra = (ws & MASK) | x;
            ws = (uint32_t)*(k1e_lds_cu16)(uintptr_t)ra;
            const uint32_t ca = (ws & MASK) | x;
            (void)__hip_atomic_fetch_add((k1e_lds_u32)(uintptr_t)ca, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (j < 16) lo |= (ws & 3u) << (2 * j); else hi |= (ws & 3u) << (2 * (j - 16));
          }
          p.codes[((size_t)(e0 + c) * p.nch + (j0 >> 5)) * (size_t)t.B + b] = make_uint2(lo, hi);
        }
        w[c] = ws;
      }
    }
    // ---- the chain that takes the segment's last transition leaves the instance's state behind ----
#pragma unroll
    for (int c = 0; c < K1E_EPL; ++c) {
      if (valid[c] && first[c] + len[c] == n_steps) {
        const int hend = hs[c] + len[c];
        const bool term = hend >= H;
        const int32_t cur = term ? start : (int32_t)(((w[c] & MASK) >> 3) ^ swz);
        t.cur[b] = cur;
        t.hstep[b] = term ? 0 : hend;
        if (last_obs) last_obs[b] = cur;
      }
    }
    __syncthreads();
  }

  // ---- flush: counts of the slots into visits_sa / visits_s (+ the resets of the start state) ----
  if (wave == 0 && sub == 0 && owner) {
    t.n_trans[b] = ntr + (unsigned long long)n_steps;
    t.n_reset[b] += (unsigned long long)(((int64_t)h0 + n_steps) / H);
    p.seg_h0[b] = h0;
  }
  for (int k = tid; k < nb * S; k += K1E_THREADS) {
    const int i = k / S, s = k - i * S;
    const uint32_t sw = (uint32_t)i & 15u, f = (uint32_t)i >> 4;
    const uint32_t* row = tab + (size_t)i * (slot / 4) + (((uint32_t)s ^ sw) << 1);
    const uint32_t c0 = row[f] >> 16, c1 = row[f ^ 1u] >> 16;
    int32_t add_s = (int32_t)(c0 + c1);
    if (s == meta[K1E_NI + i]) add_s += meta[i];
    if (add_s) t.visits_s[so0 + k] += add_s;
    if (c0 | c1) {
      int2* sa = reinterpret_cast<int2*>(t.visits_sa + (so0 + k) * 2);
      int2 v = *sa;
      v.x += (int32_t)c0;
      v.y += (int32_t)c1;
      *sa = v;
    }
  }
}

// The float64 reward sums of a segment, in transition order: lane = instance, one add per transition (sequential by
// definition of the sum: float64 addition does not associate and the oracle adds reward by reward).  An episode's code word
// holds the 2-bit reward codes of its steps, step j of a 32-step chunk at bits 2 j of the 64-bit word.
#define K1R_THREADS 64
#define K1R_PF 4   // code words in flight per lane
__global__ void __launch_bounds__(K1R_THREADS) k_reward_scan(EnvTables t, K1ePlan p, int64_t n_steps,
                                                            double* __restrict__ reward_sum, int accumulate) {
  __shared__ double rv[4];
  const int b = blockIdx.x * K1R_THREADS + threadIdx.x;
  if (threadIdx.x < 4) rv[threadIdx.x] = (int)threadIdx.x < p.n_codes ? p.rvals[threadIdx.x] * t.rscale - t.rmin : 0.0;
  __syncthreads();
  if (b >= t.B) return;
  const int H = p.H, h0 = p.seg_h0[b], nch = p.nch;
  double sum = accumulate ? reward_sum[b] : 0.0;
  const int64_t E = ((int64_t)h0 + n_steps + H - 1) / H;
  const int64_t W = E * nch;   // code words of this instance, in order
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2* src = reinterpret_cast<const u32x2*>(p.codes) + b;
  auto word_len = [&](int64_t wi) -> int {   // steps the code word wi holds
    const int64_t e = wi / nch;
    const int ch = (int)(wi - e * nch);
    const int64_t first = e == 0 ? 0 : e * H - h0;
    const int64_t len = min((int64_t)(H - (e == 0 ? h0 : 0)), n_steps - first);
    return (int)min((int64_t)32, len - 32 * ch);
  };
  auto add_word = [&](u32x2 c, int L) {
    const int L0 = min(L, 16);
#pragma unroll 4
    for (int j = 0; j < L0; ++j) sum += rv[(c.x >> (2 * j)) & 3u];
#pragma unroll 4
    for (int j = 16; j < L; ++j) sum += rv[(c.y >> (2 * (j - 16))) & 3u];
  };
  u32x2 q[K1R_PF];
#pragma unroll
  for (int k = 0; k < K1R_PF; ++k) q[k] = k < W ? __builtin_nontemporal_load(&src[(size_t)k * t.B]) : u32x2{0u, 0u};
  for (int64_t w0 = 0; w0 < W; w0 += K1R_PF) {
    u32x2 cur[K1R_PF];
#pragma unroll
    for (int k = 0; k < K1R_PF; ++k) {
      cur[k] = q[k];
      const int64_t nx = w0 + K1R_PF + k;
      q[k] = nx < W ? __builtin_nontemporal_load(&src[(size_t)nx * t.B]) : u32x2{0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < K1R_PF; ++k)
      if (w0 + k < W) add_word(cur[k], word_len(w0 + k));
  }
  reward_sum[b] = sum;
}
