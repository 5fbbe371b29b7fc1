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
// The kernel is then bound by THROUGHPUT, not by latency -- of two pipes at once (tools/calib/lds_atomic_rate.hip,
// tools/calib/valu_int_rate.hip; profiles/r04_lds_atomic_rate.txt): a wave64 ds_add_rtn_u32 occupies a CU's LDS pipe for 4.3
// cycles (a ds_read_b32 for 1.1), the step's integer VOP3 instructions a SIMD for 4.4-4.8 cycles each; the step alone, in a
// loop that does nothing else, runs at 22.4 cycles per wave-transition per SIMD (5.6 per CU).
//
// k_rollout_epi -- ONE 1024-thread workgroup per CU, each owning groups of NI = 32 instances one after the other:
//   LDS   every instance of the group has a PRIVATE table of S x 2 dwords {visit count : 16 | successor word : 16}, and the 32
//         tables are INTERLEAVED dword by dword: row (s, a) of instance i lives at byte a * 2^ASH + s * 128 + 4 i.  The 32
//         lanes of an LDS lane group are the 32 instances, so lane i only ever touches bank i: every access of the walk is
//         bank-conflict-free whatever the states are (a slot per instance measured 76 % of the LDS cycles as conflicts: 32
//         random dwords over 32 banks collide 3.5-fold), and no two lanes ever add to the same dword.  The successor word
//         is s' << 7 | reward code: the address of the next row is ONE v_and_or_b32, (word & 0xff80) | (action << ASH | 4 i).
//   step  v_add_co (next action bit -> VCC) . v_cndmask (the lane's base in that action's image) . v_and_or (address) .
//         ds_add_rtn_u32 . v_alignbit (reward code into the episode's code word): the returning atomic IS the table read -- it
//         counts the row the chain LEAVES and hands back its successor word: 4 VALU + 1 LDS instruction per transition,
//         spelled out in inline asm (left alone the compiler picks v_bfe . v_and . v_lshlrev . v_or3, five with the
//         v_alignbit; v_bfe . v_lshl_or . v_and_or is four but two of them VOP3 where v_add_co . v_cndmask are VOP2: 25.2 /
//         22.4 / 20.9 cycles per wave-transition per SIMD in isolation).  An idle chain (no episode left for it) walks from
//         state 0 under action 0 and adds ZERO.
//   lanes lane = (instance l & 31, half l >> 5).  Wavefront w owns a CONTIGUOUS range of the segment's episodes and walks it
//         eight at a time (a round): half s, chain c in {0 .. 3} walks episode 8 (R w + r) + 4 s + c -- four independent
//         chains per lane, their atomics issued together.
//   bits  the action bits of a round come from the wavefront's OWN ring of Philox blocks in LDS (interleaved like the
//         tables): lanes 0-31 make one block of their instance, lanes 32-63 the next; consecutive rounds continue in the
//         stream, so every block is made once.  A chain fetches its 32 bits with one funnel shift.  Every bit of every block
//         is used (cmdp_device.h: the packed domain-2 stream): 0.8 VALU instructions per transition instead of the 25 a whole
//         block per four one-bit actions cost K1U.  LDS operations of one wavefront are performed in order, so the walk of a
//         group needs NO workgroup barrier (a ring shared by the workgroup needed two per 128 episodes: 0.04 ms of 0.6).
//   next  the table image of the workgroup's NEXT group is loaded into registers during the last round of the walk (and the
//         instance scalars under the flush): the staging's HBM round trips run under the walk.
//   flush the table dwords hold DEPARTURE counts; they are added to a launch-spanning departure image in HBM (layout of the
//         LDS image) by returnless 64-bit atomics -- no read-modify-write round trip, the workgroup goes on at once.  The
//         reference counts the ARRIVAL state under the action taken (base.py:1302-1303): a linear function of the departure
//         counts, formed by k_epi_fold when somebody needs the counters.
//   out   the reward codes of an episode (2 bits per step) go to HBM, 8 bytes per (episode, 32-step chunk), layout
//         codes[episode][chunk][instance], with the number of steps per code next to them; state, in-episode time and the
//         Philox counters are advanced as if the transitions had been taken one by one.
// k_reward_scan -- lane = instance: the float64 reward sum in TRANSITION ORDER from the code words (bit-equal to the
//         oracle's and every other kernel's sequential sum).  No LDS at all and <= 128 VGPRs: k_rollout_epi holds every byte
//         of the CU's LDS and 4 x 96 VGPRs per SIMD, and a wavefront of the scan still fits beside it.
// Results: visit counts, final states, in-episode times, Philox counters and reward sums bit-equal to K1 / K1T / K1U and the
// CPU oracle (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py, tools/stress_k1t.py k1e, tools/fuzz_parity.py).
#pragma once

#define K1E_THREADS 1024
#define K1E_NW (K1E_THREADS / 64)
#define K1E_NI 32                          // instances per workgroup: the 32 lanes of an LDS lane group, one bank each
#define K1E_EPL 4                          // chains (episodes) per lane
#define K1E_EPP (K1E_NW * 2 * K1E_EPL)     // episodes of one instance per round of the whole workgroup (128)
#define K1E_SEG 61440                      // transitions per segment (< 65 536: the 16-bit counts in the table dwords)
#define K1E_SMASK 0xff80u                  // the state field of a successor word (s' << 7; reward code in bits 1:0)

struct K1ePlan {
  int32_t S, H;
  int32_t ash;               // log2 of the action stride in bytes: 7 + log2(SP), SP = states padded to a power of two
  int32_t ring_blocks;       // power of two: Philox blocks of an instance in ONE WAVEFRONT's action-bit ring (the blocks of the eight
                             // episodes the wavefront walks at a time, plus one)
  int32_t n_codes;           // <= 4 distinct reward values (2-bit codes)
  int32_t nch;               // 32-step chunks per episode, ceil(H / 32)
  int32_t n_pass;            // rounds per wavefront in this segment: ceil(episodes / 128)
  int32_t gdw;               // dwords of a group's table image in HBM: S * 32 rounded up to whole rounds of 1024 (the workgroup's
                             // loads and returnless adds need no bounds test; the padding is zero and stays zero)
  int32_t debug;             // CMDP_K1E_DEBUG (timing experiments, results INVALID): 1 no walk, 2 no flush, 4 no Philox, 8 no code stores
  const uint32_t* etab;      // [group of 32 instances][gdw >= S * 32]: one dword per state, successor words of action 0 (low half)
                             // and 1 (high half), s' << 7 | code; interleaved by instance like the LDS image
  const double* rvals;       // [n_codes]
  uint2* codes;              // [episode][chunk][B] 2-bit reward codes of the chunk's steps, step j at bits 2 j
  uint32_t* cnts;            // [episode][chunk][B] steps of the chunk with reward code 1 | code 2 << 11 | code 3 << 22
  int32_t* seg_h0;           // [B] in-episode time at the start of the segment (k_reward_scan decodes the episodes with it)
  int2* dep;                 // [group][gdw] DEPARTURE counts (action 0, action 1) accumulated over launches, interleaved like
                             // the LDS image; k_epi_fold turns them into the reference's arrival counts when they are needed
  int32_t* dep_res;          // [B] episode resets not yet added to visits_s of the start state
};

// LDS: tables (2 actions x SP states x 32 instances x 4 B) | one ring per wavefront (ring_blocks x 4 dwords x 32 instances)
__host__ __device__ inline size_t k1e_tab_bytes(const K1ePlan& p) { return (size_t)2 << p.ash; }
__host__ __device__ inline size_t k1e_ring_bytes(const K1ePlan& p) { return (size_t)K1E_NW * (size_t)p.ring_blocks * 512; }
__host__ __device__ inline size_t k1e_lds_bytes(const K1ePlan& p) { return k1e_tab_bytes(p) + k1e_ring_bytes(p); }
__host__ __device__ inline size_t k1e_fold_lds_bytes(const K1ePlan& p) { return (size_t)2 * K1E_NI * (size_t)p.S * 4; }
__host__ __device__ inline int64_t k1e_max_episodes(int64_t n_steps, int H) { return (n_steps + 2 * (int64_t)H - 2) / H; }

// steps of a code word (2-bit fields, unused fields zero) with code 1, 2, 3: n1 | n2 << 11 | n3 << 22 (11-bit fields: the
// packed words of a 16-word tile add up without carries into the neighbouring field)
// (at most three reward codes -- DeepSea: no field is 3, so the two bit planes ARE the counts of codes 1 and 2)
__device__ __forceinline__ uint32_t k1e_code_counts3(uint32_t lo, uint32_t hi) {
  const uint32_t n1 = __popc(lo & 0x55555555u) + __popc(hi & 0x55555555u);
  const uint32_t n2 = __popc(lo & 0xaaaaaaaau) + __popc(hi & 0xaaaaaaaau);
  return n1 | (n2 << 11);
}
__device__ __forceinline__ uint32_t k1e_code_counts(uint32_t lo, uint32_t hi) {
  const uint32_t l0 = lo & 0x55555555u, l1 = (lo >> 1) & 0x55555555u;
  const uint32_t h0 = hi & 0x55555555u, h1 = (hi >> 1) & 0x55555555u;
  const uint32_t n3 = __popc(l0 & l1) + __popc(h0 & h1);
  const uint32_t n1 = __popc(l0 & ~l1) + __popc(h0 & ~h1);
  const uint32_t n2 = __popc(~l0 & l1) + __popc(~h0 & h1);
  return n1 | (n2 << 11) | (n3 << 22);
}

typedef __attribute__((address_space(3))) uint32_t* k1e_lds_u32;

// what a workgroup needs of a group of 32 instances before it can walk it: the table image (<= 16 dwords per thread) and the
// lane's instance scalars.  A workgroup owns SEVERAL groups, one after the other, and loads the next group's set during the
// last pass of the walk of the current one: the HBM round trips of the staging run under the walk
// (the table image itself travels in plain local arrays of sixteen, indexed by constants only: scalars after SROA -- a struct
// that is copied as a whole ends up in scratch, a 16-wide vector value is spilled as a whole)
struct K1eGroupRegs {
  uint32_t key_x, key_y, ntr_lo, ntr_hi;
  uint32_t hcs;   // in-episode time << 18 | current state << 9 | start state (S <= 512, H < 2^14: one register instead of three)
};

// FEW: at most three reward codes (the count word is then four population counts; computing both forms and selecting -- what
// a run-time switch compiles to -- cost 1.5 VALU instructions per transition)
template <bool FEW>
__global__ void __launch_bounds__(K1E_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) k_rollout_epi(EnvTables t, K1ePlan p, int n_steps,
                                                            int32_t* __restrict__ last_obs) {
  extern __shared__ __align__(16) unsigned char smem[];
  // (the reward scan of the previous launch shares the SIMDs with this kernel: the walk takes the issue slots first, the scan
  // -- one latency-bound wavefront per SIMD with a whole launch to finish in -- what is left.  Not when a workgroup has a
  // single group to walk: the launch is then shorter than the scan, and the scan is what the step waits for)
  if ((t.B + K1E_NI - 1) / K1E_NI > (int)gridDim.x) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int inst = lane & (K1E_NI - 1), sub = lane >> 5;
  const int H = p.H;
  const int n_groups = (t.B + K1E_NI - 1) / K1E_NI;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // 0: no static LDS here
  const uint32_t tab_bytes = 2u << p.ash;
  const uint32_t a_words = (1u << p.ash) >> 2;           // dwords between the two actions' images
  uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
  const uint32_t RD = (uint32_t)p.ring_blocks * 4u;      // ring dwords per instance
  uint32_t* ring = reinterpret_cast<uint32_t*>(smem + tab_bytes) + (size_t)wave * RD * K1E_NI;   // this wavefront's ring
  const uint32_t lbase = lds0 + 4u * (uint32_t)inst;     // the lane's bank
  const uint32_t lbase1 = lbase | (1u << p.ash);         // ... in the image of action 1
  const int nch = p.nch;
  const uint32_t ash = (uint32_t)p.ash;
  const int nj = p.gdw / K1E_THREADS;                     // rounds of the workgroup over a group's image (<= 16)

  // the loads of group g (k = s * 32 + i of the interleaved HBM image: coalesced; S <= 512: at most 16 dwords per thread)
  // (the empty asm statements pin the address arithmetic to this place: hoisted out of the pass loop -- the group is invariant
  // there -- sixteen 64-bit addresses would sit in registers under the whole walk)
  auto fetch_table = [&](int g, uint32_t (&pair)[16]) {
    const uint32_t* src = p.etab + (size_t)g * (size_t)p.gdw;
    uint32_t tid_o = (uint32_t)tid;
    asm volatile("" : "+s"(src), "+v"(tid_o));
    // (uniform base + 32-bit lane offset: no 64-bit address per load)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      pair[j] = j < nj ? __builtin_nontemporal_load(&src[tid_o + (uint32_t)(j * K1E_THREADS)]) : 0u;
  };
  auto fetch_scalars = [&](int g, K1eGroupRegs& r) {
    int bb = min(g * K1E_NI + inst, t.B - 1);   // (lanes past the batch repeat its last instance; they own nothing)
    asm volatile("" : "+v"(bb));
    const uint2 kk = t.philox_key[bb];
    const unsigned long long nn = t.n_trans[bb];
    r.key_x = kk.x; r.key_y = kk.y;
    r.ntr_lo = (uint32_t)nn; r.ntr_hi = (uint32_t)(nn >> 32);
    r.hcs = ((uint32_t)t.hstep[bb] << 18) | ((uint32_t)t.cur[bb] << 9) | (uint32_t)t.start_state[t.start_off[bb]];
  };

  K1eGroupRegs cur_regs;
  uint32_t cur_pair[16];
  fetch_table(blockIdx.x, cur_pair);
  fetch_scalars(blockIdx.x, cur_regs);

  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int g0 = g * K1E_NI;
    const int nb = min(K1E_NI, t.B - g0);
    const bool owner = inst < nb;
    const int b = g0 + (owner ? inst : 0);
    const bool more = g + (int)gridDim.x < n_groups;
    // ---- stage: {count 0 | successor word} from the registers the previous group's last pass (or the prologue) filled ----
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t k = (uint32_t)tid + (uint32_t)(j * K1E_THREADS);
      if (j < nj) {   // (gdw <= 2^ash / 4: the padding of the image lands in the padding of the LDS table)
        tab[k] = cur_pair[j] & 0xffffu;
        tab[a_words + k] = cur_pair[j] >> 16;
      }
    }
    __syncthreads();   // the image is complete before any wavefront walks it
    const uint2 key = make_uint2(cur_regs.key_x, cur_regs.key_y);
    const unsigned long long ntr = (unsigned long long)cur_regs.ntr_lo | ((unsigned long long)cur_regs.ntr_hi << 32);
    const uint32_t ntr_lo = (uint32_t)ntr;
    const int h0 = (int)(cur_regs.hcs >> 18);
    const int32_t cur0 = (int32_t)((cur_regs.hcs >> 9) & 511u);
    const int32_t start = (int32_t)(cur_regs.hcs & 511u);
    const unsigned long long blk0 = ntr >> 7;               // first Philox block of the segment
    const int off0 = (int)(ntr & 127ull);

    // The action bits.  A wavefront walks a CONTIGUOUS range of the segment's episodes, eight at a time (a round), and makes
    // the Philox blocks of a round itself, into its own ring: lanes 0-31 one block of their instance, lanes 32-63 the next.
    // Consecutive rounds continue in the stream, so every block is made once (but for the two at the ends of a wavefront's
    // range) -- and the walk needs NO workgroup barrier: LDS operations of one wavefront are performed in order.
    // ring dword d of the instance lives at ring[d * 32 + inst]: the stores and the chains' fetches stay in bank i
    int next_blk = 0;   // first block (relative to blk0) of this wavefront's range not made yet
    auto produce = [&](int e_lo) {   // e_lo: the round's first episode (wave-uniform)
      if (!owner) return;
      const int tlo = e_lo == 0 ? 0 : e_lo * H - h0;
      const int thi = min(n_steps, (e_lo + 2 * K1E_EPL) * H - h0);
      if (tlo >= thi) return;
      const int blo = (off0 + tlo) >> 7, bhi = (off0 + thi - 1) >> 7;
      int sub_o = sub;
      asm volatile("" : "+v"(sub_o));   // (as in fetch_group: nothing of this loop's 64-bit arithmetic is worth a register pair across the walk)
      for (int rb = max(blo, next_blk) + sub_o; rb <= bhi; rb += 2) {
        const unsigned long long blk = blk0 + (unsigned long long)rb;
        uint32_t w[4];
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 2u, 0u, key.x, key.y, w);
        const uint32_t d0 = ((uint32_t)blk & ((uint32_t)p.ring_blocks - 1u)) << 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[(size_t)(d0 + j) * K1E_NI + inst] = w[j];
      }
      next_blk = bhi + 1;
    };
    // the 32 action bits from transition `rel` of the segment on
    auto fetch_bits = [&](int rel) -> uint32_t {
      const uint32_t a0 = ntr_lo + (uint32_t)rel;
      const uint32_t di = (a0 >> 5) & (RD - 1u);
      const uint32_t d0 = ring[di * K1E_NI + inst], d1 = ring[((di + 1u) & (RD - 1u)) * K1E_NI + inst];
      return __builtin_amdgcn_alignbit(d1, d0, a0 & 31u);
    };

    uint32_t next_pair[16];
    const int R = p.n_pass;   // rounds per wavefront: wavefront w owns episodes [8 R w, 8 R (w + 1)) of the segment
    for (int pass = 0; pass < R; ++pass) {
      const int e_lo = (wave * R + pass) * 2 * K1E_EPL;
      if (!(p.debug & 4)) produce(e_lo);
      // the next group's table image and scalars: in flight under the whole of this wavefront's last round
      if (pass == R - 1 && more) fetch_table(g + (int)gridDim.x, next_pair);
      __builtin_amdgcn_wave_barrier();
      // ---- the lane's chains of this round (everything relative to the segment fits an int: n_steps <= K1E_SEG) ----
      // (first transition, length and kind of chain c are recomputed where they are needed, not kept in registers across
      // the walk: the next group's table image waits in registers under the last pass)
      const int e0 = e_lo + sub * K1E_EPL;
      auto first_of = [&](int c) -> int { return e0 + c == 0 ? 0 : (e0 + c) * H - h0; };
      auto len_of = [&](int c) -> int {
        const int f = first_of(c);
        return owner && f < n_steps ? min(e0 + c == 0 ? H - h0 : H, n_steps - f) : 0;
      };
      auto full_of = [&](int c) -> bool { return len_of(c) == H; };
      auto valid_of = [&](int c) -> bool { return owner && first_of(c) < n_steps; };
      uint32_t w[K1E_EPL], addv[K1E_EPL];
#pragma unroll
      for (int c = 0; c < K1E_EPL; ++c) {
        // a chain that is not walked at full length in the fast loop idles: it follows action 0 from state 0 and ADDS ZERO
        w[c] = full_of(c) ? ((uint32_t)(e0 + c == 0 ? cur0 : start) << 7) : 0u;
        addv[c] = full_of(c) ? 0x10000u : 0u;
      }
      bool any_full = false;
#pragma unroll
      for (int c = 0; c < K1E_EPL; ++c) any_full |= full_of(c);

      // ---- fast loop: full episodes, uniform trip counts; lanes without any full chain are masked off as a whole ----
      if (any_full && !(p.debug & 1)) {
        for (int ch = 0; ch < nch; ++ch) {
          const int L = __builtin_amdgcn_readfirstlane(min(32, H - 32 * ch));
          uint32_t bits[K1E_EPL], clo[K1E_EPL], chi[K1E_EPL];
#pragma unroll
          for (int c = 0; c < K1E_EPL; ++c) {
            const uint32_t fb = fetch_bits(first_of(c) + 32 * ch);
            bits[c] = full_of(c) ? __builtin_bitreverse32(fb) : 0u;
            clo[c] = 0u; chi[c] = 0u;
          }
          // one step of all the lane's chains: the atomics of all chains are issued before the first is waited for
          auto steps = [&](uint32_t (&cw)[K1E_EPL]) {
            uint32_t ra[K1E_EPL];
#pragma unroll
            for (int c = 0; c < K1E_EPL; ++c) {
              // (spelled out: left to itself the compiler may pick v_bfe . v_and (literal) . v_lshlrev . v_or3 -- five VALU
              // instructions per transition with the code word's v_alignbit instead of four)
              // (the action bits are consumed from the TOP of the bit-reversed window: v_add_co_u32 shifts the next one into VCC,
              // v_cndmask_b32 picks the lane's base in that action's image -- two VOP2 instructions where v_bfe_u32 .
              // v_lshl_or_b32 are two VOP3 ones: 20.9 instead of 22.4 cycles per wave-transition in isolation,
              // tools/calib/lds_atomic_rate.hip)
              uint32_t ax;
              asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc" : "+v"(bits[c]), "=v"(ax) : "v"(lbase), "v"(lbase1) : "vcc");
              asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ra[c]) : "v"(w[c]), "s"(K1E_SMASK), "v"(ax));
            }
#pragma unroll
            for (int c = 0; c < K1E_EPL; ++c)   // counts the row the chain leaves, returns {old count | successor word}
              w[c] = __hip_atomic_fetch_add((k1e_lds_u32)(uintptr_t)ra[c], addv[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int c = 0; c < K1E_EPL; ++c) cw[c] = __builtin_amdgcn_alignbit(w[c], cw[c], 2u);
          };
          const int L0 = __builtin_amdgcn_readfirstlane(min(L, 16));   // (wave-uniform by construction: keep the loop control scalar)
          {
            int j = 0;
            for (; j + 1 < L0; j += 2) { steps(clo); steps(clo); }
            if (j < L0) steps(clo);
            j = 16;
            for (; j + 1 < L; j += 2) { steps(chi); steps(chi); }
            if (j < L) steps(chi);
          }
          if (!(p.debug & 8)) {
            uint32_t wi = ((uint32_t)e0 * (uint32_t)nch + (uint32_t)ch) * (uint32_t)t.B + (uint32_t)b;
#pragma unroll
            for (int c = 0; c < K1E_EPL; ++c) {
              if (full_of(c)) {
                const uint32_t lo = L0 < 16 ? clo[c] >> (32 - 2 * L0) : clo[c];
                const uint32_t hi = L > 16 ? (L < 32 ? chi[c] >> (32 - 2 * (L - 16)) : chi[c]) : 0u;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 v;
                v.x = lo; v.y = hi;
                __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p.codes) + wi);
                __builtin_nontemporal_store(FEW ? k1e_code_counts3(lo, hi) : k1e_code_counts(lo, hi), p.cnts + wi);
              }
              wi += (uint32_t)nch * (uint32_t)t.B;
            }
          }
        }
      }
      // ---- slow loop: the partial episodes at the two ends of the segment (per-lane lengths) ----
#pragma unroll
      for (int c = 0; c < K1E_EPL; ++c) {
        if (valid_of(c) && !full_of(c)) {
          const int len_c = len_of(c), first_c = first_of(c);
          uint32_t ws = (uint32_t)((e0 + c) == 0 ? cur0 : start) << 7;
          for (int j0 = 0; j0 < len_c; j0 += 32) {
            const int L = min(32, len_c - j0);
            const uint32_t bits = fetch_bits(first_c + j0);
            uint32_t lo = 0u, hi = 0u;
            for (int j = 0; j < L; ++j) {
              const uint32_t a = (bits >> j) & 1u;
              const uint32_t ra = (ws & K1E_SMASK) | ((a << ash) | lbase);
              ws = __hip_atomic_fetch_add((k1e_lds_u32)(uintptr_t)ra, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (j < 16) lo |= (ws & 3u) << (2 * j); else hi |= (ws & 3u) << (2 * (j - 16));
            }
            const uint32_t wi = ((uint32_t)(e0 + c) * (uint32_t)nch + (uint32_t)(j0 >> 5)) * (uint32_t)t.B + (uint32_t)b;
            p.codes[wi] = make_uint2(lo, hi);
            p.cnts[wi] = k1e_code_counts(lo, hi);
          }
          w[c] = ws;
        }
      }
      // ---- the chain that takes the segment's last transition leaves the instance's state behind ----
#pragma unroll
      for (int c = 0; c < K1E_EPL; ++c) {
        if (valid_of(c) && first_of(c) + len_of(c) == n_steps) {
          const int hend = ((e0 + c) == 0 ? h0 : 0) + len_of(c);
          const bool term = hend >= H;
          const int32_t cur = term ? start : (int32_t)((w[c] & K1E_SMASK) >> 7);
          t.cur[b] = cur;
          t.hstep[b] = term ? 0 : hend;
          if (last_obs) last_obs[b] = cur;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();   // every wavefront has walked its range: the counts are final

    if (wave == 0 && sub == 0 && owner) {
      const unsigned long long resets = (unsigned long long)(((int64_t)h0 + n_steps) / H);
      t.n_trans[b] = ntr + (unsigned long long)n_steps;
      t.n_reset[b] += resets;
      p.seg_h0[b] = h0;
      if (!(p.debug & 2)) p.dep_res[b] += (int32_t)resets;   // episode resets: they count the start state (k_epi_fold)
    }
    // ---- flush.  The table dwords hold DEPARTURE counts (row (s, a) left); they are added to the launch-spanning departure
    // image in HBM, which has the layout of the LDS image: a straight coalesced, conflict-free pass.  The reference's ARRIVAL
    // counts (base.py:1302-1303) are a linear function of them, formed by k_epi_fold when somebody needs the counters.
    // No read-modify-write round trip: the two counts of an entry are added by ONE returnless 64-bit atomic -- action 0 in
    // the low half, action 1 in the high half; the host's overflow guard keeps every count below 2^31, so the low half never
    // carries into the high one -- executed in L2 behind the workgroup's back while it walks its next group ----
    if (!(p.debug & 2)) {
      typedef __attribute__((address_space(1))) unsigned long long* k1e_glb_u64;
      unsigned long long* dep = reinterpret_cast<unsigned long long*>(p.dep) + (size_t)g * (size_t)p.gdw;
      uint32_t tid_o = (uint32_t)tid;
      asm volatile("" : "+s"(dep), "+v"(tid_o));   // (as in fetch_group: no hoisted 64-bit offsets)
      const k1e_glb_u64 gdep = (k1e_glb_u64)(uintptr_t)dep;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t k = tid_o + (uint32_t)(j * K1E_THREADS);
        if (j < nj) {
          const uint32_t c0 = tab[k] >> 16, c1 = tab[a_words + k] >> 16;
          if (c0 | c1)
            __hip_atomic_fetch_add(gdep + k, (unsigned long long)c0 | ((unsigned long long)c1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!more) break;
    // (the next group's instance scalars: five more registers than the walk has to spare -- this group's are dead now, and the
    // round trip runs under the flush's LDS reads and the barrier)
    fetch_scalars(g + (int)gridDim.x, cur_regs);
    __syncthreads();   // the counts are read: the next group's image may overwrite them
#pragma unroll
    for (int j = 0; j < 16; ++j) cur_pair[j] = next_pair[j];
  }
}

// Departure counts -> the reference's visit counters.  BaseMDP.step counts the ARRIVAL node under the action taken
// (colosseum/mdp/base.py:1302-1303): visits_sa[s'][a] += sum of the departures of the rows (s, a) whose successor is s',
// visits_s[s'] += both actions' arrivals (+ the resets, which count the start state: base.py:1268-1277).  One workgroup per
// group of 32 instances; arrival images [action][instance][state] in LDS filled by atomics, then coalesced read-modify-writes
// (a wavefront per instance); the departure image is cleared.  Counters saturate at INT32_MAX and raise *overflow.
__global__ void __launch_bounds__(K1E_THREADS) k_epi_fold(EnvTables t, K1ePlan p, int32_t* __restrict__ overflow) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* arr0 = reinterpret_cast<uint32_t*>(smem);
  const int S = p.S;
  uint32_t* arr1 = arr0 + (size_t)K1E_NI * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = blockIdx.x * K1E_NI;
  const int nb = min(K1E_NI, t.B - g0);
  for (int k = tid; k < 2 * K1E_NI * S; k += K1E_THREADS) arr0[k] = 0u;
  __syncthreads();
  int2* dep = p.dep + (size_t)blockIdx.x * (size_t)p.gdw;
  const uint32_t* et = p.etab + (size_t)blockIdx.x * (size_t)p.gdw;
  for (int k = tid; k < S * K1E_NI; k += K1E_THREADS) {   // k = s * 32 + i
    const int2 c = dep[k];
    if (c.x | c.y) {
      const uint32_t pair = et[k];
      const int i = k & 31;
      if (c.x) atomicAdd(&arr0[i * S + (int)((pair & K1E_SMASK) >> 7)], (uint32_t)c.x);
      if (c.y) atomicAdd(&arr1[i * S + (int)(((pair >> 16) & K1E_SMASK) >> 7)], (uint32_t)c.y);
      dep[k] = make_int2(0, 0);
    }
  }
  __syncthreads();
  const int64_t so0 = t.state_off[g0];
  bool ovf = false;
  auto sat_add = [&](int32_t old, uint32_t add) -> int32_t {
    const int64_t v = (int64_t)old + (int64_t)add;
    if (v > 0x7fffffffLL) { ovf = true; return 0x7fffffff; }
    return (int32_t)v;
  };
  for (int i = wave; i < nb; i += K1E_NW) {
    const int bi = g0 + i;
    const int32_t s_start = t.start_state[t.start_off[bi]];
    const uint32_t n_res = (uint32_t)p.dep_res[bi];
    const int64_t gs0 = so0 + (int64_t)i * S;
    for (int s = lane; s < S; s += 64) {
      const uint32_t c0 = arr0[i * S + s], c1 = arr1[i * S + s];
      const uint32_t add_s = c0 + c1 + (s == s_start ? n_res : 0u);
      if (add_s) t.visits_s[gs0 + s] = sat_add(t.visits_s[gs0 + s], add_s);
      if (c0 | c1) {
        int2* sa = reinterpret_cast<int2*>(t.visits_sa + (gs0 + s) * 2);
        int2 v = *sa;
        v.x = sat_add(v.x, c0);
        v.y = sat_add(v.y, c1);
        *sa = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) p.dep_res[bi] = 0;
  }
  if (ovf) atomicOr(overflow, 1);
}

// The float64 reward sums of a segment, in TRANSITION ORDER: lane = instance.  The sum is sequential by definition (float64
// addition does not associate and the oracle adds reward by reward), but almost all of it can be done EXACTLY in integers:
// while the running sum S stays inside one binade [2^k, 2^(k+1)) its spacing is u = 2^(k-52), S = m u with an integer m, and
// for a reward v >= 0 the IEEE sum is fl(S + v) = (m + q) u with q = round-to-nearest(v / u) -- an integer that depends on v
// and k only -- unless v / u lies exactly half way between two integers (ties-to-even then looks at m).  Integer addition
// associates, so an episode chunk with n_c steps of reward code c advances m by sum_c n_c q_c, whatever the order: valid
// as long as no code present is a tie case in this binade and m stays below 2^53 (q >= 0: every partial sum then stayed in
// the binade too).  Otherwise -- a binade is crossed (~15 times per 30 000 transitions), S is zero or subnormal, a reward is
// negative or a tie case -- the chunk is added step by step in float64, exactly as the oracle does, and (k, m, q) are
// re-derived from the result.  An episode's code word holds the 2-bit reward codes of its steps, step j of a 32-step
// chunk at bits 2 j of the 64-bit word (unused fields zero).
// element `i` (run-time, per lane) of a register array of 16: a select tree instead of an LDS round trip -- the reward scan
// keeps NO tile in LDS, so that its wavefronts fit beside k_rollout_epi's workgroups (whose tables take the CU's LDS)
typedef uint32_t k1r_u32x16 __attribute__((ext_vector_type(16)));   // (a vector value, not an array: it cannot end up in scratch)
__device__ __forceinline__ uint32_t k1r_sel16(const k1r_u32x16 a, int i) {
  const bool i0 = (i & 1) != 0, i1 = (i & 2) != 0, i2 = (i & 4) != 0, i3 = (i & 8) != 0;
  const uint32_t b0 = i0 ? a[1] : a[0], b1 = i0 ? a[3] : a[2], b2 = i0 ? a[5] : a[4], b3 = i0 ? a[7] : a[6];
  const uint32_t b4 = i0 ? a[9] : a[8], b5 = i0 ? a[11] : a[10], b6 = i0 ? a[13] : a[12], b7 = i0 ? a[15] : a[14];
  const uint32_t c0 = i1 ? b1 : b0, c1 = i1 ? b3 : b2, c2 = i1 ? b5 : b4, c3 = i1 ? b7 : b6;
  const uint32_t d0 = i2 ? c1 : c0, d1 = i2 ? c3 : c2;
  return i3 ? d1 : d0;
}

#define K1R_THREADS 64
#define K1R_T 16   // code words per tile: the next tile's loads are in flight while a tile is summed
// (<= 128 VGPRs and 32 B of LDS: a wavefront of the scan fits on every SIMD beside k_rollout_epi's workgroup -- 4 x 96 VGPRs
// per SIMD, 147 840 B of LDS per CU -- so that the scan of one segment runs under the walk of the next)
__global__ void __launch_bounds__(K1R_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) k_reward_scan(EnvTables t, K1ePlan p, int64_t n_steps,
                                                            double* __restrict__ reward_sum, int accumulate) {
  const int lane = threadIdx.x;
  const int b = min(blockIdx.x * K1R_THREADS + lane, t.B - 1);   // (lanes past the batch repeat its last instance, unstored)
  const bool mine = blockIdx.x * K1R_THREADS + lane < t.B;
  double rv[4];
  bool bulk_allowed = true;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    rv[c] = c < p.n_codes ? p.rvals[c] * t.rscale - t.rmin : 0.0;
    bulk_allowed = bulk_allowed && rv[c] >= 0.0 && rv[c] < 1.0e300;
  }
  const double rv0 = rv[0], rv1 = rv[1], rv2 = rv[2], rv3 = rv[3];
  // per code: mantissa with the hidden bit and biased exponent of the reward (zero / subnormal rewards: eb = 0)
  unsigned long long mv[4];
  int eb[4];
  const unsigned long long M52 = (1ull << 52) - 1ull, B52 = 1ull << 52;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long vb = (unsigned long long)__double_as_longlong(rv[c]);
    eb[c] = (int)(vb >> 52) & 0x7ff;
    mv[c] = (vb & M52) | B52;
  }
  const int H = p.H, h0 = p.seg_h0[b], nch = p.nch;
  double S = accumulate ? reward_sum[b] : 0.0;
  const int64_t E = ((int64_t)h0 + n_steps + H - 1) / H;
  const int W = (int)(E * nch);   // code words of this instance, in order (a segment: < 2^31)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2* csrc = reinterpret_cast<const u32x2*>(p.codes) + b;
  const uint32_t* nsrc = p.cnts + b;
  // integer form of the sum, valid while `kvalid`: S = m * 2^(kb - 1075) with 2^52 <= m < 2^53 (kb = biased exponent);
  // q[c] = round(rv[c] / spacing) as (low, high) halves; `tie`: codes whose presence forces the float64 path (a tie case in
  // this binade, a subnormal reward, a reward at or above 2^(k+1))
  bool kvalid = false;
  int kb = 0;
  unsigned long long m = 0ull;
  uint32_t qlo[4] = {0u, 0u, 0u, 0u}, qhi[4] = {0u, 0u, 0u, 0u};
  uint32_t tie = 0u;
  auto rebase = [&]() {
    kvalid = false;
    if (!bulk_allowed || !(S >= 2.2250738585072014e-308) || !(S < 1.0e300)) return;
    const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
    kb = (int)(sb >> 52);
    m = (sb & M52) | B52;
    tie = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned long long qq = 0ull;
      if (rv[c] != 0.0) {
        const int d = kb - eb[c];
        if (eb[c] == 0 || d < 0) tie |= 1u << c;   // subnormal reward, or v >= 2^(k+1): the sum would leave the binade
        else if (d == 0) qq = mv[c];
        else if (d < 55) {
          const unsigned long long half = 1ull << (d - 1), rem = mv[c] & ((1ull << d) - 1ull);
          qq = (mv[c] >> d) + (rem > half ? 1ull : 0ull);
          if (rem == half) tie |= 1u << c;
        }
      }
      qlo[c] = (uint32_t)qq;
      qhi[c] = (uint32_t)(qq >> 32);
    }
    kvalid = true;
  };
  rebase();
  // m + the advance of `steps` (<= 512) transitions with packed counts nw; true = the whole of it stays in the integer form
  auto advance = [&](uint32_t nw, uint32_t steps, unsigned long long& m2) -> bool {
    const uint32_t n1 = nw & 0x7ffu, n2 = (nw >> 11) & 0x7ffu, n3 = nw >> 22, n0 = steps - n1 - n2 - n3;
    const uint32_t present = (n0 ? 1u : 0u) | (n1 ? 2u : 0u) | (n2 ? 4u : 0u) | (n3 ? 8u : 0u);
    unsigned long long lo64 = (unsigned long long)n0 * qlo[0];
    lo64 += (unsigned long long)n1 * qlo[1];
    lo64 += (unsigned long long)n2 * qlo[2];
    lo64 += (unsigned long long)n3 * qlo[3];
    const uint32_t hi32 = n0 * qhi[0] + n1 * qhi[1] + n2 * qhi[2] + n3 * qhi[3];   // <= 512 * 2^22
    m2 = m + lo64 + ((unsigned long long)hi32 << 32);
    return kvalid && (present & tie) == 0u && m2 < (1ull << 53);
  };
  auto compose = [&]() { return __longlong_as_double((long long)(((unsigned long long)kb << 52) | (m & M52))); };

  // episode / chunk bookkeeping of the next word to be laid out
  int ch = 0;
  int e_len = (int)min((int64_t)(H - h0), n_steps);   // steps of the current episode inside this segment
  int ep_left = e_len;                                 // ... not yet covered by earlier chunks
  int64_t tot_left = n_steps;                          // steps from the current episode's first one on

  // two tiles of count words in flight (rnA, rnB alternate): one tile's sums cover about a third of an HBM round trip
  uint32_t rnA[K1R_T], rnB[K1R_T];
#pragma unroll
  for (int k = 0; k < K1R_T; ++k) {
    rnA[k] = k < W ? __builtin_nontemporal_load(&nsrc[(size_t)k * t.B]) : 0u;
    rnB[k] = K1R_T + k < W ? __builtin_nontemporal_load(&nsrc[(size_t)(K1R_T + k) * t.B]) : 0u;
  }
  auto tile = [&](uint32_t (&rn)[K1R_T], const int w0) {
    // prefix sums of the tile's packed counts and steps (0 steps: a chunk a partial episode does not reach, or past the end)
    k1r_u32x16 P, SP;
    // interior tiles of single-chunk episodes (all lanes): every word holds H steps
    const bool interior = nch == 1 && __all(w0 >= 1 && w0 + K1R_T <= W - 1);
    if (interior) {
      uint32_t acc = 0u;
#pragma unroll
      for (int k = 0; k < K1R_T; ++k) { acc += rn[k]; P[k] = acc; SP[k] = (uint32_t)((k + 1) * H); }
      tot_left -= (int64_t)K1R_T * H;   // sixteen full episodes on: the next one may be the segment's last, partial one
      e_len = (int)min((int64_t)H, tot_left);
      ep_left = e_len;
    } else {
      uint32_t acc = 0u, sacc = 0u;
#pragma unroll
      for (int k = 0; k < K1R_T; ++k) {
        const int L = (w0 + k < W) ? max(0, min(32, ep_left)) : 0;
        acc += L > 0 ? rn[k] : 0u;
        sacc += (uint32_t)L;
        P[k] = acc; SP[k] = sacc;
        ep_left -= 32;
        if (++ch == nch) {
          ch = 0;
          tot_left -= e_len;
          e_len = (int)min((int64_t)H, tot_left);
          ep_left = e_len;
        }
      }
    }
    const uint32_t Pt = P[K1R_T - 1], St = SP[K1R_T - 1];
    unsigned long long m2;
    const bool whole = advance(Pt, St, m2);
    // the count words of the tile after the next (the code words of a word are fetched only when that word has to be added
    // step by step)
    {
      const uint32_t* nn = nsrc + (size_t)(w0 + 2 * K1R_T) * t.B;
#pragma unroll
      for (int k = 0; k < K1R_T; ++k) rn[k] = (w0 + 2 * K1R_T + k < W) ? __builtin_nontemporal_load(&nn[(size_t)k * t.B]) : 0u;
    }
    if (whole) { m = m2; return; }
    // ---- this lane: the first word that does not advance (binary search over the prefix sums: advancing is monotone --
    // q >= 0, and a forcing code stays present), everything before it in one piece, that word in float64 step by step
    // exactly as the oracle adds, then the rest of the tile again ----
    uint32_t base_n = 0u, base_s = 0u;
    int from = 0;
    while (from < K1R_T) {
      if (advance(Pt - base_n, St - base_s, m2)) { m = m2; break; }
      int lo = from, hi = K1R_T - 1;
#pragma unroll
      for (int it = 0; it < 4; ++it) {   // 2^4 = K1R_T
        const int mid = (lo + hi) >> 1;
        const bool ok = advance(k1r_sel16(P, mid) - base_n, k1r_sel16(SP, mid) - base_s, m2);
        if (lo < hi) { if (ok) lo = mid + 1; else hi = mid; }
      }
      const int k = lo;
      const uint32_t Sk1 = k > 0 ? k1r_sel16(SP, k - 1) : 0u;
      if (k > from && advance(k1r_sel16(P, k - 1) - base_n, Sk1 - base_s, m2)) m = m2;
      const uint32_t Sk = k1r_sel16(SP, k);
      const uint32_t L = Sk - Sk1;
      if (L) {
        const u32x2 c = csrc[(size_t)(w0 + k) * t.B];
        if (kvalid) S = compose();
        // (step by step exactly as the oracle adds, but for the steps whose reward is +0.0 -- x + 0.0 == x, the sum is never
        // -0.0 -- which are skipped: the scaled minimum reward is 0, half of DeepSea's steps.  The reward of a code by a
        // select on four registers: the scan owns no LDS at all, k_rollout_epi may hold every byte of the CU's.  One wavefront
        // per SIMD issues an instruction every ~8 cycles, so this path is priced by its instruction count)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const uint32_t cw = half ? c.y : c.x;
          const int Lh = (int)L - 16 * half;
          if (Lh > 0) {
            const uint32_t vm = Lh >= 16 ? 0x55555555u : ((1u << (2 * Lh)) - 1u) & 0x55555555u;
            const uint32_t b0 = cw & vm, b1 = (cw >> 1) & vm;
            uint32_t nz = 0u;
            if (rv0 != 0.0) nz |= vm & ~(b0 | b1);
            if (rv1 != 0.0) nz |= b0 & ~b1;
            if (rv2 != 0.0) nz |= b1 & ~b0;
            if (rv3 != 0.0) nz |= b0 & b1;
            while (nz) {
              const uint32_t pos = (uint32_t)__builtin_ctz(nz);
              nz &= nz - 1u;
              const uint32_t cd = cw >> pos;
              const bool c0 = (cd & 1u) != 0u, c1 = (cd & 2u) != 0u;
              const double lo2 = c0 ? rv1 : rv0, hi2 = c0 ? rv3 : rv2;
              S += c1 ? hi2 : lo2;
            }
          }
        }
        rebase();
      }
      base_n = k1r_sel16(P, k);
      base_s = Sk;
      from = k + 1;
    }
  };
  for (int w0 = 0; w0 < W; w0 += 2 * K1R_T) {
    tile(rnA, w0);
    if (w0 + K1R_T < W) tile(rnB, w0 + K1R_T);
  }
  if (kvalid) S = compose();
  if (mine) reward_sum[b] = S;
}
