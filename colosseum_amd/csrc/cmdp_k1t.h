// cmdp_k1t.h -- K1T k_rollout_tmpl: the wavefront-pipeline rollout K1P with ONE successor table per workgroup.
//
// A Colosseum benchmark batch is "the same MDP, many seeds".  For the families whose structure does not depend on the
// seed (`does_seed_change_MDP_structure()` False: DeepSea, config C2) the seed only permutes the ACTIONS of every state
// (`randomize_actions`, mdp/base.py:505-518): instance i's packed successor words of state s are the template's words of
// s, possibly swapped (A = 2).  K1P nevertheless keeps a private copy of the table per instance (1 860 B at C2) next to
// the 930 B of count deltas, and a CU's LDS holds 52 instances: the rollout is (resident chains) / (latency of a chain's
// dependent LDS read), so the table copies cost two thirds of the throughput.
//
// K1T keeps the template ONCE per workgroup and one swap BIT per state and instance (59 B at C2).  A transition reads the
// state's two template words (one ds_read_b32) and the byte holding its swap bit -- two independent reads, one LDS
// latency -- and selects word[a ^ bit]: the very word instance i's private table holds at (s, a), so trace, counters,
// rewards and episode logic are K1P's, unchanged, and so are the results (bit-equal to K1P, K1 and the oracle).  128
// instances per CU instead of 52: config C2 runs two rounds of workgroups instead of five.
//
// Roles of the sixteen wavefronts of a 1024-thread workgroup (wavefront w sits on SIMD w mod 4; instance = 64 * half + lane):
//   0, 1    CHAIN of half 0 / 1 (raised priority: they share their SIMD with a reward adder and two producers)
//   2, 3    COUNTS of half 0 / 1
//   4, 5    REWARDS of half 0 / 1
//   6..13   PRODUCERS: 6 + 2 j + half -> Philox blocks j, j + 4, ... of its half's chunk (as K1P's four, per half)
//   14, 15  idle
// Eligibility (cmdp_create): the K1P conditions, A = 2, and every instance's words a per-state permutation of instance
// 0's.  Other batches keep K1P.
#pragma once

#define K1T_THREADS 1024
#define K1T_NPROD 4              // producers per half
#define K1T_OVF 14               // wrap events an instance can record between two flushes (K1P: 30)
#define K1T_FIXED (K1L_NRV * 8 + 128 * 4 + 16)   // rv2[K1L_NRV] f64, resets[128] i32, near[2][2] i32

struct TmplPlan {
  int32_t G;             // instances per workgroup (<= 128)
  int32_t rows;          // S * A of every instance (A == 2)
  int32_t tmpl_bytes;    // LDS bytes of the template (rows * 2 rounded up to 16)
  int32_t slot_bytes;    // LDS bytes per instance slot: swap bits | count deltas | overflow list
  int32_t mask_bytes;    // swap bits per instance in HBM and LDS (multiple of 4)
  int32_t off_cnt, off_ovf;
  int32_t ch;            // transitions per ring chunk (multiple of 8)
  int32_t n_codes;
  int32_t code_shift;
  int32_t debug;         // timing experiments only (CMDP_K1T_DEBUG): 1 no counts, 2 no rewards, 4 no producers, 8 no chain
  const uint16_t* tmpl;      // [rows] successor row base as byte offset (2 A s') | reward code << code_shift
  const uint8_t* swap_bits;  // [B][mask_bytes] bit s: instance's actions of state s are the template's, swapped
  const double* rvals;       // [n_codes]
};

__host__ __device__ inline size_t k1t_lds_bytes(const TmplPlan& p, int g) {
  return (size_t)K1T_FIXED + (size_t)p.tmpl_bytes +
         (size_t)g * (size_t)(p.slot_bytes + 2 * K1P_ACT_STRIDE(p.ch) + 2 * K1P_TR_STRIDE(p.ch));
}

__global__ void __launch_bounds__(K1T_THREADS) k_rollout_tmpl(EnvTables t, TmplPlan p, int64_t n_steps,
                                                             double* __restrict__ reward_sum,
                                                             int32_t* __restrict__ last_obs) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // scalar: the role branches are uniform
  const int g0 = blockIdx.x * p.G;
  const int nb = min(p.G, t.B - g0);
  double* rv2 = reinterpret_cast<double*>(smem);
  int32_t* resets = reinterpret_cast<int32_t*>(smem + K1L_NRV * 8);   // [128]
  int32_t* near = resets + 128;                                        // [chunk parity][half]: some counter may wrap next chunk
  unsigned char* tmpl = smem + K1T_FIXED;
  const int CH = p.ch;
  const int AS = K1P_ACT_STRIDE(CH), TS = K1P_TR_STRIDE(CH);
  unsigned char* ring = tmpl + p.tmpl_bytes;              // [2][G] action bytes, stride AS
  unsigned char* trace = ring + 2 * p.G * AS;             // [2][G] uint16 trace entries, stride TS bytes
  unsigned char* slots = trace + 2 * p.G * TS;
  constexpr int A = 2;
  const int H = t.H;
  const int64_t so0 = t.state_off[g0];
  const int S = (int)(t.state_off[g0 + 1] - so0);
  const int rows = S * A;
  const int64_t row00 = so0 * A;
  const int total_rows = nb * rows, total_states = nb * S;
  const int cnt_dwords = (rows + 4 + 3) / 4;
  // ---- stage: reward values, the template (once), every instance's swap bits; count deltas zeroed ------------------
  for (int i = tid; i < p.n_codes; i += K1T_THREADS) rv2[i] = p.rvals[i] * t.rscale - t.rmin;
  for (int i = tid; i < p.tmpl_bytes / 4; i += K1T_THREADS)
    reinterpret_cast<uint32_t*>(tmpl)[i] = (2 * i < rows) ? reinterpret_cast<const uint32_t*>(p.tmpl)[i] : 0u;
  const int mask_dwords = p.mask_bytes / 4;
  for (int j = tid; j < nb * mask_dwords; j += K1T_THREADS) {
    const int slot = j / mask_dwords, off = j - slot * mask_dwords;
    reinterpret_cast<uint32_t*>(slots + (size_t)slot * p.slot_bytes)[off] =
        reinterpret_cast<const uint32_t*>(p.swap_bits + (size_t)(g0 + slot) * p.mask_bytes)[off];
  }
  for (int j = tid; j < nb * cnt_dwords; j += K1T_THREADS) {
    const int slot = j / cnt_dwords, off = j - slot * cnt_dwords;
    reinterpret_cast<uint32_t*>(slots + (size_t)slot * p.slot_bytes + p.off_cnt)[off] = 0u;
  }
  // role and instance of this lane
  const int half = wave < 6 ? (wave & 1) : ((wave - 6) & 1);
  const int pidx = (wave >= 6 && wave < 6 + 2 * K1T_NPROD) ? (wave - 6) >> 1 : -1;
  const int li = half * 64 + lane;
  const bool owner = li < nb;  // this lane's instance exists
  const int b = g0 + (owner ? li : 0);
  unsigned char* base = slots + (size_t)(owner ? li : 0) * p.slot_bytes;
  // LDS-address-space views of the template and of this lane's swap bits: `ds_read v, addr offset:imm`, no generic-pointer
  // arithmetic on the chain
  typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
  typedef const __attribute__((address_space(3))) uint8_t* lds_u8;
  const lds_u32 tmpl_l = (lds_u32)(__attribute__((address_space(3))) unsigned char*)tmpl;
  const lds_u8 swp_l = (lds_u8)(__attribute__((address_space(3))) unsigned char*)base;
  uint8_t* c8 = base + p.off_cnt;
  uint16_t* ovf = reinterpret_cast<uint16_t*>(base + p.off_ovf);
  const int smask = (1 << p.code_shift) - 1;
  // chain state: `cur` is the byte offset of the current state's row base (2 A s = 4 s)
  const int32_t start_k = t.start_state[t.start_off[b]] * A * 2;
  int32_t cur = t.cur[b] * A * 2, h = t.hstep[b];
  int32_t n_resets = 0, n_resets_total = 0;
  const bool episodic = H > 0;
  const bool uniform_h = episodic && wave < 2 && __all(!owner || h == __builtin_amdgcn_readfirstlane(h));
  int n_ovf = 0, maxc = 0;   // counts state: overflow entries, largest counter value since the last flush
  double sum = 0.0;  // rewards state
  const uint2 my_key = t.philox_key[b];
  const unsigned long long my_ntr = t.n_trans[b];
  __syncthreads();

  // actions of transitions [first, first + len) of this lane's instance, Philox blocks pidx, pidx + K1T_NPROD, ...
  auto produce = [&](int buf, int64_t first, int len) {
    if (!owner) return;
    const unsigned long long n0 = my_ntr + (unsigned long long)first;
    const int rel0 = (int)(n0 & 3ull);  // the window starts inside a block when the counter is not a multiple of 4
    const unsigned long long q0 = n0 >> 2;
    const int nblk = (rel0 + len + 3) >> 2;
    unsigned char* dst = ring + ((size_t)buf * p.G + li) * AS;
    for (int qi = pidx; qi < nblk; qi += K1T_NPROD) {
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

  // chain: walks chunk `cb` of `len` transitions
  auto chain_body = [&](int cb, int len) {
    if (!owner || len <= 0) return;
    const unsigned char* acts = ring + ((size_t)cb * p.G + li) * AS;
    uint16_t* tr = reinterpret_cast<uint16_t*>(trace + ((size_t)cb * p.G + li) * TS);
    int hs = uniform_h ? __builtin_amdgcn_readfirstlane(h) : 0, nres_s = 0;
    int pend = 0;
    // flavours as in K1P: T0 no episode logic, T1 scalar episode test, T2 per-lane
    auto step = [&](int s, int a, auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      // the two reads of the dependency chain, independent of each other: the state's two template words and the
      // byte with its swap bit (bit s of the instance's mask: s = cur / 4)
      const uint32_t pair = tmpl_l[cur >> 2];
      const uint32_t mb = swp_l[cur >> 5];
      // word index k = a ^ bit, as a shift 16 k: (mb >> pos) has the bit at position 0 and garbage above it; shifted left
      // by 4 and added to 16 a, bit 4 of the sum is a ^ bit (bits 0-3 are zero) and v_bfe_u32 reads only the low five bits
      // of its offset operand -- three dependent instructions from the reads to the next state
      const uint32_t sh = ((mb >> ((cur >> 2) & 7)) << 4) + (uint32_t)(a << 4);
      const int nxt = (int)__builtin_amdgcn_ubfe(pair, sh, (uint32_t)p.code_shift);
      const int word = (int)((pair >> (sh & 16u)) & 0xffffu);   // == instance's own table word at (s, a); off the chain
      tr[s - 1] = (uint16_t)pend;   // the previous transition's trace entry queues behind this transition's reads
      pend = word + 2 * a;          // arrival row under the action taken (base.py:1302-1303) | reward code
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
        const bool term = episodic && h >= H;  // episodic termination followed at once by reset()
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
#define K1T_ACT(j) (int)((((j) < 4 ? a_lo : a_hi) >> (8 * ((j) & 3))) & 0xffu)
      if (uniform_h && hs + 8 < H) {
        step(s0 + 0, K1T_ACT(0), T0{}); step(s0 + 1, K1T_ACT(1), T0{}); step(s0 + 2, K1T_ACT(2), T0{});
        step(s0 + 3, K1T_ACT(3), T0{}); step(s0 + 4, K1T_ACT(4), T0{}); step(s0 + 5, K1T_ACT(5), T0{});
        step(s0 + 6, K1T_ACT(6), T0{}); step(s0 + 7, K1T_ACT(7), T0{});
        hs += 8;
      } else if (uniform_h && H >= 8) {
        const int jstar = H - hs - 1;  // exactly one episode ends inside this group, after transition jstar
#define K1T_STEP_R(j)                         \
  step(s0 + (j), K1T_ACT(j), T0{});           \
  if (jstar == (j)) { cur = start_k; ++nres_s; }
        K1T_STEP_R(0) K1T_STEP_R(1) K1T_STEP_R(2) K1T_STEP_R(3) K1T_STEP_R(4) K1T_STEP_R(5) K1T_STEP_R(6) K1T_STEP_R(7)
#undef K1T_STEP_R
        hs = 7 - jstar;
      } else if (uniform_h) {
        step(s0 + 0, K1T_ACT(0), T1{}); step(s0 + 1, K1T_ACT(1), T1{}); step(s0 + 2, K1T_ACT(2), T1{});
        step(s0 + 3, K1T_ACT(3), T1{}); step(s0 + 4, K1T_ACT(4), T1{}); step(s0 + 5, K1T_ACT(5), T1{});
        step(s0 + 6, K1T_ACT(6), T1{}); step(s0 + 7, K1T_ACT(7), T1{});
      } else {
        step(s0 + 0, K1T_ACT(0), T2{}); step(s0 + 1, K1T_ACT(1), T2{}); step(s0 + 2, K1T_ACT(2), T2{});
        step(s0 + 3, K1T_ACT(3), T2{}); step(s0 + 4, K1T_ACT(4), T2{}); step(s0 + 5, K1T_ACT(5), T2{});
        step(s0 + 6, K1T_ACT(6), T2{}); step(s0 + 7, K1T_ACT(7), T2{});
      }
#undef K1T_ACT
    }
    for (; s0 < len; ++s0) {  // ragged tail of the launch's last chunk
      const int a = acts[s0];
      if (uniform_h) step(s0, a, T1{}); else step(s0, a, T2{});
    }
    tr[len - 1] = (uint16_t)pend;
    if (uniform_h) { h = hs; n_resets += nres_s; }
  };
  // counts: the chunk the chain traced into buffer `tb`
  auto counts_body = [&](int tb, int plen) {
    if (!owner || plen <= 0) return;
    const unsigned char* trb = trace + ((size_t)tb * p.G + li) * TS;
    auto count2 = [&](int x0, int x1) {   // two transitions per LDS round trip, as K1P
      const int r0 = c8[x0], r1 = c8[x1];
      const int c0 = r0 + 1;
      const int c1 = (x1 == x0 ? (c0 & 255) : r1) + 1;
      ovf[n_ovf] = (uint16_t)x0;
      n_ovf += c0 >> 8;
      ovf[n_ovf] = (uint16_t)x1;
      n_ovf += c1 >> 8;
      c8[x0] = (uint8_t)c0;
      c8[x1] = (uint8_t)c1;
      maxc = max(maxc, max(c0, c1));
    };
    auto count1 = [&](int x0) {
      const int c0 = (int)c8[x0] + 1;
      ovf[n_ovf] = (uint16_t)x0;
      n_ovf += c0 >> 8;
      c8[x0] = (uint8_t)c0;
      maxc = max(maxc, c0);
    };
    int s0 = 0;
    for (; s0 + 4 <= plen; s0 += 4) {
      const uint32_t e0 = *reinterpret_cast<const uint32_t*>(trb + 2 * s0);
      const uint32_t e1 = *reinterpret_cast<const uint32_t*>(trb + 2 * s0 + 4);
      count2((int)((e0 & (uint32_t)smask) >> 1), (int)(((e0 >> 16) & (uint32_t)smask) >> 1));
      count2((int)((e1 & (uint32_t)smask) >> 1), (int)(((e1 >> 16) & (uint32_t)smask) >> 1));
    }
    for (; s0 < plen; ++s0) count1((int)((reinterpret_cast<const uint16_t*>(trb)[s0] & smask) >> 1));
  };
  // rewards: values of the traced reward codes in transition order (bit-equal to the sequential sum)
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

  if (wave < 2) __builtin_amdgcn_s_setprio(3);   // the chains are the critical path of the workgroup
  if (n_steps > 0 && pidx >= 0) produce(0, 0, (int)min((int64_t)CH, n_steps));
  __syncthreads();
  // Pipeline iteration as K1P: the chains walk chunk `cb` while the bookkeepers take the chunk before it and the
  // producers fill the one after; every wave runs its own compact loop around the one barrier per chunk.
  //
  // When to flush the 8-bit count deltas.  Counters start at zero after a flush, so they wrap at most once per 256
  // transitions of their instance and the overflow list (K1T_OVF entries) is safe for 256 K1T_OVF transitions: K1P
  // flushes at that period.  A flush is an HBM read-modify-write of every counter of the group -- at C2 a quarter of
  // the launch -- and almost always for nothing: no counter is anywhere near 256.  The counts wavefronts therefore keep
  // the largest value any of their counters has had since the last flush and publish, per chunk, whether one could wrap
  // during the next chunk (>= 256 - ch); the period rule only applies once that is the case.  Until then nothing has
  // wrapped (the list is empty), afterwards the period bound holds as before.  C2: two flushes per round instead of nine.
  int since_flush = 0, cb = 0, len = 0, plen = 0;
  int64_t left = n_steps;
  bool last = false, flush_now = false;
  auto begin_iter = [&]() {
    plen = len;
    len = (int)min((int64_t)CH, left);
    left -= len;
  };
  auto end_iter = [&]() {   // after the chunk's barrier: every wave reads the same two flags (double-buffered by chunk parity)
    since_flush += plen;
    last = len == 0;  // the bookkeepers have just drained the final chunk
    const bool near_wrap = (near[2 * cb] | near[2 * cb + 1]) != 0;
    flush_now = last || (near_wrap && since_flush + CH > 256 * K1T_OVF);
    cb ^= 1;
  };
  LdsPlan fp{};   // what flush_counts reads of a plan
  fp.slot_bytes = p.slot_bytes;
  fp.off_cnt = p.off_cnt;
  for (;;) {
    if (wave < 2) {
      do {
        begin_iter();
        if (!(p.debug & 8)) chain_body(cb, len);
        __syncthreads();
        end_iter();
      } while (!flush_now);
      if (owner) {
        resets[li] = n_resets;
        n_resets_total += n_resets;
        n_resets = 0;
      }
    } else if (wave < 4) {
      do {
        begin_iter();
        if (!(p.debug & 1)) counts_body(cb ^ 1, plen);
        const bool mine = __any(owner && maxc >= 256 - CH);
        if (lane == 0) near[2 * cb + half] = mine ? 1 : 0;
        __syncthreads();
        end_iter();
      } while (!flush_now);
      maxc = 0;
    } else if (wave < 6) {
      do {
        begin_iter();
        if (!(p.debug & 2)) rewards_body(cb ^ 1, plen);
        __syncthreads();
        end_iter();
      } while (!flush_now);
    } else if (pidx >= 0) {
      do {
        begin_iter();
        if (left > 0 && !(p.debug & 4)) produce(cb ^ 1, n_steps - left, (int)min((int64_t)CH, left));
        __syncthreads();
        end_iter();
      } while (!flush_now);
    } else {
      do {
        begin_iter();
        __syncthreads();
        end_iter();
      } while (!flush_now);
    }
    __syncthreads();   // the chains' reset counts are in LDS
    flush_counts<true, true, K1T_THREADS>(t.visits_sa + row00, total_rows, rows, A, slots, fp, resets, nullptr, tid);
    flush_counts<false, true, K1T_THREADS>(t.visits_s + so0, total_states, S, A, slots, fp, resets, t.start_state + t.start_off[g0], tid);
    __syncthreads();
    if (wave >= 2 && wave < 4 && owner) {  // every recorded wrap is worth 256 visits
      for (int e = 0; e < n_ovf; ++e) {
        const int r = ovf[e];
        t.visits_sa[row00 + (int64_t)li * rows + r] += 256;
        t.visits_s[so0 + (int64_t)li * S + r / A] += 256;
      }
      n_ovf = 0;
    }
    if (last) break;
    for (int j = tid; j < nb * cnt_dwords; j += K1T_THREADS) {
      const int slot = j / cnt_dwords, off = j - slot * cnt_dwords;
      reinterpret_cast<uint32_t*>(slots + (size_t)slot * p.slot_bytes + p.off_cnt)[off] = 0u;
    }
    since_flush = 0;
    __syncthreads();
  }
  if (wave < 2 && owner) {
    cur /= 2 * A;
    // (the addresses of the instance's scalars are formed HERE, from an index the optimiser cannot trace back to the loads at
    // the top: kept from there they were three register pairs spilled across the whole walk)
    int bo = b;
    asm volatile("" : "+v"(bo));
    t.cur[bo] = cur;
    t.hstep[bo] = h;
    t.n_trans[bo] = my_ntr + (unsigned long long)n_steps;
    t.n_reset[bo] += (unsigned long long)n_resets_total;
    if (last_obs) last_obs[bo] = cur;  // the state after the last transition (the start state after a termination)
  }
  if (wave >= 4 && wave < 6 && owner && reward_sum) reward_sum[b] = sum;
}
