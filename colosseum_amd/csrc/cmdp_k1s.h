// cmdp_k1s.h -- K1S k_rollout_stoch: the random-policy rollout of batches with STOCHASTIC dynamics, tables resident in LDS.
//
// K1 (lane per instance, sampler tables in HBM) pays several dependent HBM / L2 round trips per transition -- row
// descriptor, the row's cumulative probabilities, the successor -- and ~640 bytes of sector traffic for a 44-byte
// transition; a batch of a few thousand instances is latency-bound at 0.8-4e9 transitions/s.  The sampler tables of the
// generated families are highly redundant, though:
//   * the cumulative-probability vector of a row takes one of a handful of values in the whole batch (it is a function of
//     p_lazy, p_rand, the number of actions and the shape of the nominal outcome lists): PATTERNS, shared by the batch;
//   * the rows of a state draw their successors from the same few states (a cell's neighbours and itself): a per-state
//     SUCCESSOR SET of U <= 16 states, and every row entry is a 4-bit index into it -- one 64-bit word per row;
//   * (round 3) those words, paired with the row's pattern, take few distinct values in the whole batch as well (FrozenLake
//     20x20: ~100 ROW SHAPES, MiniGrid: 12, DeepSea: 3): a shared dictionary of shapes, and ONE BYTE per row naming its
//     shape (two when the batch has more than 256).
// With that, an instance is 2 bytes per row (shape, visit-count delta) + 2 U bytes per state (FrozenLake 20x20: 7 KB
// instead of 18 KB in round 2 and 67 KB of float64 / int32 tables) and G instances fit a CU's LDS.  One workgroup runs G
// instances: wavefronts 0 .. nw-1 are WALKERS, each walks gw = G / nw of the instances entirely on chip, a TEAM of lanes
// per instance (round 2 had ONE walker wavefront: the LDS pipeline was 2 % busy and the workgroup was bound by that one
// wavefront's instruction issue); the other wavefronts produce the random-policy action bytes and the 53-bit transition
// uniforms of the NEXT chunk (Philox domains 2 and 0, the same streams as K1 and the CPU oracle) into double-buffered LDS rings.
//
// Per transition the walker does: action + uniform from the ring; the row's shape byte, then the shape's word + pattern id (one 16-byte read);
// the pattern's 16 padded cumulative values (independent ds_reads) and the count #{k < n-1 : cum_k <= u * total} ==
// bisect_right of `random.choices` (custom_samplers.py:59-72, identical arithmetic to choose_index); the 4-bit code;
// the successor from the state's set; the arrival row's 8-bit visit counter (overflow list as in K1L); the reward code
// (per successor state or per row) and the sequential float64 reward sum.  Results are bit-equal to K1 and the oracle.
#pragma once

#define K1S_THREADS 512      // up to four walker wavefronts + four producers
#define K1S_MAXE 16          // entries per row (4-bit codes in a 64-bit word)
#define K1S_OVF 30           // wrap events of the 8-bit counters an instance can record between two flushes
#define K1S_MAXSTART 8
#define K1S_PAT_STRIDE 17    // doubles per pattern: 16 padded cumulative values + the total

struct K1sPlan {
  int32_t G;             // instances per workgroup (<= 64)
  int32_t S, rows;       // states / rows of every instance (equal over the batch)
  int32_t U;             // successor-set slots per state
  int32_t n_pat;         // patterns
  int32_t n_codes;       // distinct reward values
  int32_t reward_mode;   // 0: code per successor state, 1: code per row (carried by the row's shape)
  int32_t rc_packed;     // mode 0: the successor-set entries hold state | code << 12 (no per-instance code table)
  int32_t ch;            // transitions per ring chunk
  int32_t team;          // lanes of a walker wavefront per instance (power of two <= 16: 16 / team entries per lane)
  int32_t nw, gw;        // walker wavefronts, instances per walker wavefront (nw * gw >= G, gw * team <= 64)
  int32_t n_shapes, shape_bytes;   // row shapes of the batch; bytes per row naming its shape (1 or 2)
  int32_t slot_bytes;    // LDS bytes per instance
  int32_t off_cnt, off_ovf, off_sets, off_rc, off_start;   // byte offsets inside a slot (the shape ids start the slot)
  const void* shape;                // [R] uint8 / uint16 shape of the row
  const uint4* dict;                // [n_shapes] {word lo, word hi, pattern, 0}: the 4-bit successor-set indices of the row's entries
  const uint16_t* sets;             // [NS][U] successor sets
  const uint8_t* rcode;             // [NS] or [R] reward codes
  const double* patterns;           // [n_pat][K1S_PAT_STRIDE]
  const double* rvals;              // [n_codes]
};

__host__ __device__ inline size_t k1s_fixed_bytes(int n_pat, int n_shapes) {
  // patterns, reward values (after the range rescale), per-instance keys / counters, the shape dictionary
  return (size_t)n_pat * K1S_PAT_STRIDE * 8 + 256 * 8 + 64 * 8 + 64 * 8 + 64 * 8 + (size_t)n_shapes * 16;
}
__host__ __device__ inline size_t k1s_ring_bytes(int ch) { return (size_t)2 * (8 * ch + ch); }  // per instance: uniforms + actions, 2 buffers

__global__ void __launch_bounds__(K1S_THREADS) k_rollout_stoch(EnvTables t, K1sPlan p, int64_t n_steps,
                                                              double* __restrict__ reward_sum, int32_t* __restrict__ last_obs) {
  extern __shared__ __align__(16) unsigned char k1s_smem[];
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * p.G;
  const int nb = min(p.G, t.B - g0);
  const int A = t.A, H = t.H, S = p.S, rows = p.rows, U = p.U, CH = p.ch;
  uint4* dict = reinterpret_cast<uint4*>(k1s_smem);                               // [n_shapes], 16-byte aligned
  double* pats = reinterpret_cast<double*>(dict + p.n_shapes);
  double* rv2 = pats + (size_t)p.n_pat * K1S_PAT_STRIDE;                         // [256]
  uint2* keys = reinterpret_cast<uint2*>(rv2 + 256);                              // [64]
  unsigned long long* ntr = reinterpret_cast<unsigned long long*>(keys + 64);     // [64] transition counters at launch
  unsigned long long* nrs = ntr + 64;                                             // [64] reset counters at launch
  unsigned char* ring_u = reinterpret_cast<unsigned char*>(nrs + 64);             // [2][G][CH] doubles
  unsigned char* ring_a = ring_u + (size_t)2 * p.G * CH * 8;                      // [2][G][CH] bytes
  unsigned char* slots = ring_a + (((size_t)2 * p.G * CH + 15) & ~(size_t)15);
  // ---- stage: shared tables, then every instance's slot (instances may differ in their state counts: a slot is sized
  //      for the largest, S = p.S) ------------------------------------------------------------------------------------
  for (int i = tid; i < p.n_pat * K1S_PAT_STRIDE; i += K1S_THREADS) pats[i] = p.patterns[i];
  for (int i = tid; i < p.n_codes; i += K1S_THREADS) rv2[i] = p.rvals[i] * t.rscale - t.rmin;   // r * (max - min) - min, once per value
  for (int i = tid; i < p.n_shapes; i += K1S_THREADS) dict[i] = p.dict[i];
  const int SBY = p.shape_bytes;
  if (tid < nb) { keys[tid] = t.philox_key[g0 + tid]; ntr[tid] = t.n_trans[g0 + tid]; nrs[tid] = t.n_reset[g0 + tid]; }
  for (int i = tid; i < nb * rows; i += K1S_THREADS) {
    const int slot = i / rows, r = i - slot * rows;
    const int64_t so = t.state_off[g0 + slot];
    unsigned char* sb = slots + (size_t)slot * p.slot_bytes;
    sb[p.off_cnt + r] = 0;
    if (r < (int)(t.state_off[g0 + slot + 1] - so) * A) {
      if (SBY == 1) sb[r] = reinterpret_cast<const uint8_t*>(p.shape)[so * A + r];
      else reinterpret_cast<uint16_t*>(sb)[r] = reinterpret_cast<const uint16_t*>(p.shape)[so * A + r];
    }
  }
  for (int i = tid; i < nb * S; i += K1S_THREADS) {
    const int slot = i / S, s = i - slot * S;
    const int64_t so = t.state_off[g0 + slot];
    if (s < (int)(t.state_off[g0 + slot + 1] - so)) {
      unsigned char* sb = slots + (size_t)slot * p.slot_bytes;
      for (int u = 0; u < U; ++u) reinterpret_cast<uint16_t*>(sb + p.off_sets)[s * U + u] = p.sets[(so + s) * U + u];
      if (p.reward_mode == 0 && !p.rc_packed) sb[p.off_rc + s] = p.rcode[so + s];
    }
  }
  if (tid < nb) {  // start sampler of the instance: states and accumulated probabilities
    unsigned char* sb = slots + (size_t)tid * p.slot_bytes + p.off_start;
    const int64_t lo = t.start_off[g0 + tid];
    const int ns = (int)(t.start_off[g0 + tid + 1] - lo);
    reinterpret_cast<int32_t*>(sb)[0] = ns;
    for (int k = 0; k < ns; ++k) {
      reinterpret_cast<int32_t*>(sb)[1 + k] = t.start_state[lo + k];
      reinterpret_cast<double*>(sb + 48)[k] = t.start_cum[lo + k];
    }
    for (int k = 0; k < K1S_MAXSTART; ++k) reinterpret_cast<int32_t*>(sb + 48 + 8 * K1S_MAXSTART)[k] = 0;  // resets per start state
  }
  // A walker wavefront walks its instances in TEAMS of T lanes: the lanes of a team hold the same state and split the one part
  // of a transition that is wide -- the comparison of u * total with the row's (padded) 16 cumulative probabilities:
  // every lane compares 16 / T of them, one wave-wide ballot per entry collects the results and the popcount of the
  // team's bits is bisect_right's index (the north star's "wavefront CDF lookup").  Lane 0 of a team owns its instance's
  // counters; teams beyond the group's instances shadow instance 0 without writing.
  const int T = p.team;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = p.nw;
  const int team = (tid & 63) / T, sub = (tid & 63) - team * T;
  const int myslot = wave * p.gw + team;
  const bool walker = wave < NW && team < p.gw && myslot < nb;
  const bool writer = walker && sub == 0;
  const int wslot = walker ? myslot : 0;
  const int b = g0 + wslot;
  int32_t cur = t.cur[b], h = t.hstep[b];
  int32_t last_s = t.last_start[b], prev_s = t.prev_start[b];
  unsigned long long nr = t.n_reset[b];
  double sum = 0.0;
  unsigned char* base = slots + (size_t)wslot * p.slot_bytes;
  const uint8_t* shp8 = base;
  const uint16_t* shp16 = reinterpret_cast<const uint16_t*>(base);
  uint8_t* c8 = base + p.off_cnt;
  uint16_t* ovf = reinterpret_cast<uint16_t*>(base + p.off_ovf);
  const uint16_t* sets = reinterpret_cast<const uint16_t*>(base + p.off_sets);
  const uint8_t* rc = base + p.off_rc;
  const int32_t* st_n = reinterpret_cast<const int32_t*>(base + p.off_start);
  const double* st_cum = reinterpret_cast<const double*>(base + p.off_start + 48);
  int32_t* st_res = reinterpret_cast<int32_t*>(base + p.off_start + 48 + 8 * K1S_MAXSTART);
  int n_ovf = 0;
  __syncthreads();

  // producers: action byte and transition uniform of transitions [first, first + len) of every instance
  auto produce = [&](int buf, int64_t first, int len) {
    const int ptid = tid - 64 * NW;
    for (int item = ptid; item < nb * len; item += K1S_THREADS - 64 * NW) {
      const int slot = item / len, j = item - slot * len;
      const unsigned long long n = ntr[slot] + (unsigned long long)(first + j);
      const uint2 key = keys[slot];
      uint32_t w[4];
      philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 0u, 0u, key.x, key.y, w);
      reinterpret_cast<double*>(ring_u)[((size_t)buf * p.G + slot) * CH + j] = u53(w[0], w[1]);
      ring_a[((size_t)buf * p.G + slot) * CH + j] = (unsigned char)philox_action(n, key, A);
    }
  };

  auto flush = [&]() {  // all threads: count deltas of the group's instances into the HBM counters
    for (int i = tid; i < nb * rows; i += K1S_THREADS) {
      const int slot = i / rows, r = i - slot * rows;
      unsigned char* sb = slots + (size_t)slot * p.slot_bytes;
      const int d = sb[p.off_cnt + r];
      if (d) {
        sb[p.off_cnt + r] = 0;
        const int64_t so = t.state_off[g0 + slot];
        atomicAdd(t.visits_sa + so * A + r, d);
        atomicAdd(t.visits_s + so + r / A, d);
      }
    }
  };

  int64_t done = 0;
  int buf = 0;
  if (n_steps > 0 && wave >= NW) produce(0, 0, (int)min((int64_t)CH, n_steps));
  __syncthreads();
  int since_flush = 0;
  const bool episodic = H > 0;
  while (done < n_steps) {
    const int len = (int)min((int64_t)CH, n_steps - done);
    if (wave >= NW) {
      const int64_t nfirst = done + len;
      if (nfirst < n_steps) produce(buf ^ 1, nfirst, (int)min((int64_t)CH, n_steps - nfirst));
    } else {   // a walker wavefront, every lane (the ballots need the whole wave in step)
      const double* us = reinterpret_cast<const double*>(ring_u) + ((size_t)buf * p.G + wslot) * CH;
      const unsigned char* as = ring_a + ((size_t)buf * p.G + wslot) * CH;
      // Software-pipelined by hand: per transition the STATE depends on three LDS round trips in a row (shape byte ->
      // dictionary entry -> successor-set entry).  Everything else is taken off that chain: the action byte and the uniform
      // of transition s + 1 are fetched a transition ahead, and the bookkeeping of transition s - 1 (8-bit visit counter,
      // reward value, float64 sum in transition order) is issued right behind transition s's first read, so that its
      // round trips ride in that read's shadow (LDS returns a wavefront's reads in order).
      // The loop exists in SPECIALISED forms -- team size, single pattern (the cumulative values of the lane's entries and
      // the total then live in registers for the whole chunk), one shape byte, where the reward code comes from -- all
      // compile-time: the generic form below them carries a dozen uniform branches and a full LDS wait behind each, which
      // made a transition cost ~440 ns whatever the team size.  TT = 0 / OP = -1 / RM = -1: read from the plan at run time.
      auto walk = [&](auto tt_tag, auto op_tag, auto rm_tag) {
        constexpr int TT = decltype(tt_tag)::value, OPc = decltype(op_tag)::value, RMc = decltype(rm_tag)::value;
        constexpr int EC = K1S_MAXE / (TT ? TT : K1S_MAXE);   // entries per lane, compile time (1 when TT is not given)
        const int Tn = TT ? TT : T;
        const int En = K1S_MAXE / Tn;
        const bool one_pat = OPc < 0 ? (p.n_pat == 1) : (OPc != 0);
        const int rmode = RMc < 0 ? (p.rc_packed ? 0 : (p.reward_mode == 0 ? 1 : 2)) : RMc;   // 0 packed, 1 per state, 2 per row
        const int sby = (TT && OPc >= 0) ? 1 : SBY;   // the specialised forms are instantiated for one shape byte
        const unsigned long long tmask = (Tn == 64) ? ~0ull : ((1ull << Tn) - 1ull);
        const int tshift = team * Tn;
        double total0 = 0.0, cum0[K1S_MAXE];
        if (one_pat) {
          total0 = pats[16] + 0.0;
          if (TT) {
#pragma unroll
            for (int e = 0; e < EC; ++e) cum0[e] = pats[sub * En + e];
          }
        }
        int a_n = as[0];
        double u_n = us[0];
        int prev_arow = -1;
        double rv_prev = 0.0;
        for (int s = 0; s < len; ++s) {
          const int a = a_n;
          const double u = u_n;
          const int row = cur * A + a;
          const uint32_t sh = sby == 1 ? (uint32_t)shp8[row] : (uint32_t)shp16[row];
          if (s + 1 < len) { a_n = as[s + 1]; u_n = us[s + 1]; }
          if (writer && prev_arow >= 0) {
            const int c1 = (int)c8[prev_arow] + 1;
            ovf[n_ovf] = (uint16_t)prev_arow;                 // kept only on a wrap
            n_ovf += c1 >> 8;
            c8[prev_arow] = (uint8_t)c1;
          }
          sum += rv_prev;
          const uint4 de = dict[sh];
          const unsigned long long w = (unsigned long long)de.x | ((unsigned long long)de.y << 32);
          const double* pc = one_pat ? pats : pats + (size_t)de.z * K1S_PAT_STRIDE;
          const double x = u * (one_pat ? total0 : (pc[16] + 0.0));
          int idx = 0;
          if (Tn == 1) {
#pragma unroll
            for (int k = 0; k < K1S_MAXE; ++k) idx += ((TT && one_pat ? cum0[k] : pc[k]) <= x) ? 1 : 0;   // padded with +inf beyond n - 1
          } else if (TT) {
#pragma unroll
            for (int e = 0; e < EC; ++e) {
              const unsigned long long bal = __ballot((one_pat ? cum0[e] : pc[sub * En + e]) <= x);
              idx += __popcll((bal >> tshift) & tmask);
            }
          } else {
            for (int e = 0; e < En; ++e) {
              const unsigned long long bal = __ballot(pc[sub * En + e] <= x);
              idx += __popcll((bal >> tshift) & tmask);
            }
          }
          const int code = (int)((w >> (4 * idx)) & 15ull);
          const int se = sets[cur * U + code];
          const int nxt = rmode == 0 ? (se & 0xfff) : se;
          const int rcode = rmode == 0 ? (se >> 12) : (rmode == 1 ? (int)rc[nxt] : (int)de.w);
          rv_prev = rv2[rcode];                               // consumed a transition later
          prev_arow = nxt * A + a;                            // arrival node under the action taken (base.py:1302-1303)
          ++h;
          cur = nxt;
          if (episodic && h >= H) {                           // episodic termination followed at once by reset()
            h = 0;
            int k = 0;
            const int ns = st_n[0];
            if (ns > 1) {
              uint32_t ww[4];
              const uint2 key = keys[wslot];
              philox4x32_10((uint32_t)nr, (uint32_t)(nr >> 32), 1u, 0u, key.x, key.y, ww);
              const double xs = u53(ww[0], ww[1]) * (st_cum[ns - 1] + 0.0);
              for (int i = 0; i < ns - 1; ++i) k += (st_cum[i] <= xs) ? 1 : 0;
            }
            ++nr;
            cur = st_n[1 + k];
            if (writer) st_res[k] += 1;
            prev_s = last_s;
            last_s = cur;
          }
        }
        if (writer && prev_arow >= 0) {                       // the last transition's bookkeeping
          const int c1 = (int)c8[prev_arow] + 1;
          ovf[n_ovf] = (uint16_t)prev_arow;
          n_ovf += c1 >> 8;
          c8[prev_arow] = (uint8_t)c1;
        }
        sum += rv_prev;
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I8 = std::integral_constant<int, 8>;
      using I16 = std::integral_constant<int, 16>;
      using IR = std::integral_constant<int, -1>;
      const int rm = p.rc_packed ? 0 : (p.reward_mode == 0 ? 1 : 2);
      if (p.n_pat == 1 && SBY == 1 && (T == 16 || T == 8 || T == 1)) {
        if (T == 16) { if (rm == 0) walk(I16{}, I1{}, I0{}); else if (rm == 1) walk(I16{}, I1{}, I1{}); else walk(I16{}, I1{}, I2{}); }
        else if (T == 8) { if (rm == 0) walk(I8{}, I1{}, I0{}); else if (rm == 1) walk(I8{}, I1{}, I1{}); else walk(I8{}, I1{}, I2{}); }
        else { if (rm == 0) walk(I1{}, I1{}, I0{}); else if (rm == 1) walk(I1{}, I1{}, I1{}); else walk(I1{}, I1{}, I2{}); }
      } else {
        walk(I0{}, IR{}, IR{});
      }
    }
    done += len;
    since_flush += len;
    buf ^= 1;
    const bool flush_now = (since_flush + CH > 256 * K1S_OVF) || done >= n_steps;   // an 8-bit counter wraps at most once per 256 transitions
    __syncthreads();
    if (flush_now) {
      if (writer) {  // wraps first: each is worth 256 visits
        const int64_t so = t.state_off[b];
        for (int i = 0; i < n_ovf; ++i) {
          atomicAdd(t.visits_sa + so * A + ovf[i], 256);
          atomicAdd(t.visits_s + so + ovf[i] / A, 256);
        }
        n_ovf = 0;
      }
      flush();
      since_flush = 0;
      __syncthreads();
    }
  }
  if (writer) {
    const int64_t so = t.state_off[b];
    const int ns = st_n[0];
    for (int k = 0; k < ns; ++k)                            // resets bump the start state's visit count (env_reset)
      if (st_res[k]) atomicAdd(t.visits_s + so + st_n[1 + k], st_res[k]);
    t.last_start[b] = last_s;
    t.prev_start[b] = prev_s;
    t.cur[b] = cur;
    t.hstep[b] = h;
    t.n_trans[b] = ntr[wslot] + (unsigned long long)n_steps;
    t.n_reset[b] = nr;
    if (reward_sum) reward_sum[b] = sum;
    if (last_obs) last_obs[b] = cur;
  }
}
