// cmdp_agent.h -- batched tabular Q-learning with UCB exploration, fused with the interaction kernel
// (SURVEY.md section 8 f1).  Restates the reference's episodic agent
//   colosseum/agent/agents/episodic/q_learning.py:19-104   (QValuesModel.step_update, Jin et al. 2018)
//   colosseum/agent/actors/Q_values_actor.py:67-88         (greedy action, uniform tie-break from RandomState(seed))
// one lane per (environment instance, agent) pair.  The update mirrors numpy's scalar arithmetic OPERATION BY
// OPERATION, including its dtype promotion (NEP 50, numpy >= 2): which sub-expressions are float32 and which
// float64 depends on whether `alpha` is the Python float `min_at` or the numpy float64 (H+1)/(H+t) -- both
// regimes are reproduced, so Q tables and action streams are bit-equal to the reference's (golden G7).
#pragma once
#include "cmdp_kernels.h"

struct QlArgs {
  int32_t H, ucb;                 // ucb: 0 hoeffding, 1 bernstein
  double c1, c2, min_at, H3;      // H3 = H**3 (exact)
  const double* i_log;            // [B] np.log(S * A * optimization_horizon / p)
  const double* sqrtH7SA;         // [B] np.sqrt(H**7 * S * A)
  const int64_t* q_off;           // [B] = H * state_off[b] * A
  const int64_t* v_off;           // [B] = (H + 1) * state_off[b]
  int32_t* N;                     // [sum H*S*A] starts at 1
  float* Q;                       // starts at H
  float* V;                       // [sum (H+1)*S] starts at 0
  float* mu; float* sigma; float* beta;
  uint32_t* mt;                   // [B][624] numpy RandomState(seed) of the actor
  int32_t* mt_pos;
};

// numpy legacy RandomState(seed): init_genrand
__global__ void k_mt_seed_numpy(uint32_t* __restrict__ mt, int32_t* __restrict__ pos, const uint32_t* __restrict__ seeds,
                                int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* m = mt + (int64_t)i * 624;
  m[0] = seeds[i];
  for (int k = 1; k < 624; ++k) m[k] = 1812433253u * (m[k - 1] ^ (m[k - 1] >> 30)) + (uint32_t)k;
  pos[i] = 0;
}

__global__ void k_fill_f32(float* __restrict__ p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_i32(int32_t* __restrict__ p, int32_t v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// RC: reference-exact reward caches (cmdp_reward_cache.h) -- a lane parks after a transition whose reward block is
// missing; the relaunch (`resume`) completes the saved step first and continues with the steps the lane still owes.
template <int UCB, bool RC>
__global__ void __launch_bounds__(256) k_qlearn_episodic(EnvTables t, QlArgs q, int64_t n_steps,
                                                         const uint8_t* __restrict__ train_mask,
                                                         int8_t* __restrict__ act_trace, double* __restrict__ cum_reward,
                                                         RewardCache rc, int resume, double* __restrict__ cum_host) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  int64_t step0 = 0;
  bool pending = false, parked = false;
  if (RC && resume) {
    const long long left = rc.left[b];
    if (left == 0) return;
    step0 = n_steps - left;
    pending = rc.pend_e[b] >= 0;
  }
  const int64_t soff = t.state_off[b], ebase = t.entry_base[b];
  const int S = (int)(t.state_off[b + 1] - soff);
  const int A = t.A, H = q.H;
  const uint2 key = t.philox_key ? t.philox_key[b] : make_uint2(0, 0);
  int32_t cur = t.cur[b], h = t.hstep[b];
  unsigned long long nt = t.n_trans[b], nr = t.n_reset[b];
  float* Q = q.Q + q.q_off[b];
  int32_t* N = q.N + q.q_off[b];
  float* MU = q.mu + q.q_off[b];
  float* SG = q.sigma + q.q_off[b];
  float* BT = q.beta + q.q_off[b];
  float* V = q.V + q.v_off[b];
  uint32_t* mt = q.mt + (int64_t)b * 624;
  int32_t* mtp = q.mt_pos + b;
  const double il = q.i_log[b], s7 = q.sqrtH7SA[b];
  const bool train = train_mask ? train_mask[b] != 0 : true;
  // `self._cumulative_reward += new_ts.reward` (agent_mdp_interaction.py:291) continues across launches
  double sum = cum_reward[b];
  int64_t step = step0;
  for (; step < n_steps; ++step) {
    // One lane walks one instance and every table lives in HBM / L2, so a step costs its DEPENDENT memory round trips
    // (~1 us each; 9 of them in the straightforward order).  Where the action count allows (A <= 4) the loads are issued as
    // early as their addresses are known: the row descriptors of ALL actions next to the Q row (before the action is
    // chosen), the update's table entries right after the action (before the transition), both MT19937 words of the draw
    // together, and the row's maximum after the update from registers -- 6 round trips.  Same values, same order of
    // arithmetic.
    constexpr int AM = 4;
    const bool fastA = A <= AM;
    float qv[AM];
    RowDesc rd[AM];
    int action = 0;
    int32_t s_t, time, obs;
    double reward;
    int ty;
    if (RC && pending) {  // the step this lane parked in: transition committed, reward and update outstanding
      s_t = rc.pend_prev[b];
      action = rc.pend_act[b];
      time = h - 1;
      ty = (h >= H) ? 2 : 1;
      obs = (ty == 2) ? -1 : cur;
      if (fastA) {
#pragma unroll
        for (int a = 0; a < AM; ++a) qv[a] = (a < A) ? Q[((int64_t)time * S + s_t) * A + a] : -INFINITY;
      }
    } else {
      // ---- QValuesActor.select_action: greedy with uniform tie-break --------------------------------------
      const float* qrow = Q + ((int64_t)h * S + cur) * A;
      if (fastA) {
        const RowDesc* rp = t.row + (soff + cur) * A;
#pragma unroll
        for (int a = 0; a < AM; ++a) {
          qv[a] = (a < A) ? qrow[a] : -INFINITY;
          if (a < A) rd[a] = rp[a];
        }
      }
      float qmax;
      int n_tie = 0;
      if (fastA) {   // fully unrolled over registers (a dynamic index would send the arrays to scratch)
        qmax = qv[0];
#pragma unroll
        for (int a = 1; a < AM; ++a) qmax = fmaxf(qmax, qv[a]);
#pragma unroll
        for (int a = 0; a < AM; ++a) n_tie += (a < A && qv[a] == qmax) ? 1 : 0;
      } else {
        qmax = qrow[0];
        for (int a = 1; a < A; ++a) qmax = fmaxf(qmax, qrow[a]);
        for (int a = 0; a < A; ++a) n_tie += (qrow[a] == qmax) ? 1 : 0;
      }
      int pick = 0;
      if (n_tie > 1) {  // RandomState.choice(ties) == ties[randint(0, n)]: masked rejection on 32-bit draws
        const uint32_t mx = (uint32_t)(n_tie - 1);
        uint32_t mask = mx;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int pos = *mtp;
        uint32_t v;
        do { v = mt_next_word(mt, pos) & mask; } while (v > mx);
        *mtp = pos;
        pick = (int)v;
      }
      if (fastA) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < AM; ++a)
          if (a < A && qv[a] == qmax) { if (k == pick) action = a; ++k; }
      } else {
        for (int a = 0, k = 0; a < A; ++a)
          if (qrow[a] == qmax) { if (k == pick) action = a; ++k; }
      }
      if (act_trace) act_trace[step * t.B + b] = (int8_t)action;
      s_t = cur;
      time = h;
    }
    // the update's table entries: their addresses are known now, their loads ride under the transition's round trips
    const int64_t idx = ((int64_t)time * S + s_t) * A + action;
    int32_t n_pre = 0;
    float mu_pre = 0.0f, sg_pre = 0.0f, bt_pre = 0.0f;
    if (train) {
      n_pre = N[idx];
      if (UCB == 1) { mu_pre = MU[idx]; sg_pre = SG[idx]; bt_pre = BT[idx]; }
    }
    // ---- BaseMDP.step -------------------------------------------------------------------------------------------
    const bool was_pending = RC && pending;
    RowDesc dsel;
    if (fastA && !was_pending) {
      dsel = rd[0];
#pragma unroll
      for (int a = 1; a < AM; ++a)
        if (a == action) dsel = rd[a];
    }
    if (RC) {
      double rraw = 0.0;
      int64_t e;
      if (pending) { e = rc.pend_e[b]; pending = false; }
      else if (fastA) { ty = env_transition_desc<true>(t, soff, ebase, key, cur, h, nt, action, obs, rraw, e, dsel); ++nt; }
      else ty = env_transition(t, soff, ebase, key, cur, h, nt, action, obs, rraw, e);
      if (!rc_fetch(t.sp_rkind, rc, e, rraw)) {
        rc_park(rc, b, e, s_t, action);
        parked = true;
        break;
      }
      reward = rraw * t.rscale - t.rmin;
    } else if (fastA) {
      const unsigned long long n0 = nt;
      double rraw;
      int64_t e;
      ty = env_transition_desc<true>(t, soff, ebase, key, cur, h, n0, action, obs, rraw, e, dsel);
      ++nt;
      if (t.sp_rkind && t.sp_rkind[e] == 1) rraw = philox_beta(t.sp_rp0[e], t.sp_rp1[e], n0, key, t.beta_gammas);  // throughput mode only
      reward = rraw * t.rscale - t.rmin;
    } else {
      ty = env_step(t, soff, ebase, key, cur, h, nt, action, obs, reward);
    }
    sum += reward;
    // ---- QValuesModel.step_update ---------------------------------------------------------------------------------
    if (train) {
      // s_tp1 = -1 at termination indexes the LAST state of row time+1 == H, which is never written (0)
      const float vnext = V[(int64_t)(time + 1) * S + (ty == 2 ? S - 1 : obs)];
      const int32_t tN = n_pre + 1;
      N[idx] = tN;
      const double x = (double)(H + 1) / (double)(H + tN);
      const bool alpha_py = !(x > q.min_at);  // max(min_at, x) returns the Python float unless x is larger
      const double alpha = alpha_py ? q.min_at : x;
      float qold;
      if (fastA) {
        qold = qv[0];
#pragma unroll
        for (int a = 1; a < AM; ++a)
          if (a == action) qold = qv[a];
      } else {
        qold = Q[idx];
      }
      const float rv = __fadd_rn((float)reward, vnext);  // python float + np.float32 -> float32
      float qnew;
      if (UCB == 0) {
        const double b_t = q.c1 * sqrt(q.H3 * il / (double)tN);  // np.float64
        const double first = alpha_py ? (double)__fmul_rn((float)alpha, qold) : alpha * (double)qold;
        qnew = (float)(first + (1.0 - alpha) * ((double)rv + b_t));
      } else {
        const float mu_new = __fadd_rn(mu_pre, vnext);
        const float sg_new = __fadd_rn(sg_pre, __fmul_rn(vnext, vnext));
        MU[idx] = mu_new;
        SG[idx] = sg_new;
        const float old_beta = bt_pre;
        const float dm = __fsub_rn(sg_new, mu_new);
        const float hdm2 = __fmul_rn((float)H, __fmul_rn(dm, dm));
        const int32_t t2 = (int32_t)((uint32_t)tN * (uint32_t)tN);  // np.int32 ** 2 wraps like numpy
        const double inner = (double)hdm2 / (double)t2 + (double)H;  // float32 / int32 -> float64
        const double a1 = sqrt(inner * il);
        const double a2 = s7 * il / (double)tN;
        const double cand1 = q.c1 * (a1 + a2);
        const double cand2 = q.c2 * sqrt(q.H3 * il / (double)tN);
        const float nb = (float)((cand2 < cand1) ? cand2 : cand1);
        BT[idx] = nb;
        if (alpha_py) {  // python-float alpha: everything stays float32
          const float om = (float)(1.0 - alpha);
          const float b_t = __fdiv_rn(__fdiv_rn(__fsub_rn(nb, __fmul_rn(om, old_beta)), 2.0f), (float)alpha);
          qnew = __fadd_rn(__fmul_rn((float)alpha, qold), __fmul_rn(om, __fadd_rn(rv, b_t)));
        } else {  // numpy-float64 alpha
          const double b_t = (((double)nb - (1.0 - alpha) * (double)old_beta) / 2.0) / alpha;
          qnew = (float)(alpha * (double)qold + (1.0 - alpha) * ((double)rv + b_t));
        }
      }
      Q[idx] = qnew;
      float m2;
      if (fastA) {   // the row from registers, the updated entry replaced
        m2 = -INFINITY;
#pragma unroll
        for (int a = 0; a < AM; ++a) m2 = fmaxf(m2, (a == action) ? qnew : qv[a]);
      } else {
        const float* qr = Q + ((int64_t)time * S + s_t) * A;
        m2 = qr[0];
        for (int a = 1; a < A; ++a) m2 = fmaxf(m2, qr[a]);
      }
      V[(int64_t)time * S + s_t] = (m2 < (float)H) ? m2 : (float)H;  // min(H, Q[time, s_t].max())
    }
    if (ty == 2) {
      cur = env_reset(t, b, soff, key, nr);
      h = 0;
    }
  }
  t.cur[b] = cur;
  t.hstep[b] = h;
  t.n_trans[b] = nt;
  t.n_reset[b] = nr;
  cum_reward[b] = sum;
  if (cum_host) cum_host[b] = sum;   // page-locked host memory: the logged loop reads the sum without a copy
  if (RC) {
    rc.left[b] = parked ? (long long)(n_steps - step) : 0;
    if (!parked) rc.pend_e[b] = -1;
  }
}


// argmax_3d (reference colosseum/dynamic_programming/utils.py:28-39): one-hot greedy policy of Q[layers >= H][S][A]
// with ties broken by `np.random.seed(42); np.random.choice(ties)` -- one numpy MT19937(42) stream per table, rows in
// (h, s) order, a draw only where there is a tie (rejection sampling under the smallest covering bit mask).
//
// One WAVE per instance.  The maxima and tie counts of 64 rows are computed lane-parallel from coalesced row loads;
// MT19937 is advanced a whole 624-word block at a time by the wave (three dependency-free segments) and tempered into
// LDS; only the consumption of the stream is sequential, and it runs on wave-uniform values: the tied rows of the
// chunk come from a ballot, their tie counts and the stream words from v_readlane of a 64-word register window.
struct Mt42Init {
  uint32_t w[624];
  constexpr Mt42Init() : w{} {
    w[0] = 42u;
    for (int k = 1; k < 624; ++k) w[k] = 1812433253u * (w[k - 1] ^ (w[k - 1] >> 30)) + (uint32_t)k;
  }
};
__constant__ const Mt42Init kMt42 = Mt42Init();

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
  const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// next block of 624 outputs: mt[i] <- twist(mt[i], mt[i+1], mt[i+397 mod 624]) in sequential order.  Passes of 64
// consecutive words are dependency-free: word i needs the OLD mt[i+1] (same pass, read before written, or a later
// pass), the OLD mt[i+397] for i < 227 (a later pass) and the NEW mt[i-227] for i >= 227 (an earlier pass); i = 623
// wraps to the new mt[0].  One wave, LDS operations in program order.
__device__ __forceinline__ void mt_block(uint32_t* mt, uint32_t* out, int lane) {
  for (int base = 0; base < 624; base += 64) {
    const int i = base + lane;
    const bool on = i < 624;
    uint32_t v = 0;
    if (on) v = mt_twist(mt[i], mt[i == 623 ? 0 : i + 1], mt[i < 227 ? i + 397 : i - 227]);
    __builtin_amdgcn_wave_barrier();
    if (on) mt[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < 624; i += 64) {
    uint32_t v = mt[i];
    v ^= v >> 11;
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= v >> 18;
    out[i] = v;
  }
  __builtin_amdgcn_wave_barrier();
}

// Tie-breaks of one chunk of 64 rows (lane = row, n_tie = number of maximal entries of the row).  Sequentially every
// tied row draws words until (word & mask) <= n_tie - 1.  For a RUN of consecutive tied rows with the same n_tie the
// acceptance of a word does not depend on the row, so the k-th row of the run gets the k-th accepted word after the
// run's start and the stream continues just past the last word used: a run costs one wave step per 64-word window
// instead of one scalar loop iteration per draw.
__device__ __forceinline__ int tie_break_chunk(int n_tie, uint32_t* mt, uint32_t* out, uint32_t* tmp, int& pos, int lane) {
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long me = 1ull << lane;
  int pick = 0;
  unsigned long long need = __ballot(n_tie > 1);
  while (need) {
    const int i = __ffsll((long long)need) - 1;
    const int n = __builtin_amdgcn_readlane(n_tie, i);
    const unsigned long long same = __ballot(n_tie == n) & need;
    const unsigned long long diff = need & ~same;
    const unsigned long long run = diff ? (same & ((1ull << (__ffsll((long long)diff) - 1)) - 1ull)) : same;
    need &= ~run;
    const uint32_t mx = (uint32_t)n - 1u;
    uint32_t mask = mx;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    unsigned long long remaining = run;
    while (remaining) {
      if (pos == 624) {
        mt_block(mt, out, lane);
        pos = 0;
      }
      const uint32_t w = out[pos + lane] & mask;
      const unsigned long long ok = __ballot(pos + lane < 624 && w <= mx);
      const int nacc = __popcll(ok), R = __popcll(remaining);
      const int arank = __popcll(ok & lt);
      if (ok & me) tmp[arank] = w;
      __builtin_amdgcn_wave_barrier();
      const int rrank = __popcll(remaining & lt);
      const bool mine = (remaining & me) != 0;
      if (mine && rrank < nacc) pick = (int)tmp[rrank];
      __builtin_amdgcn_wave_barrier();
      if (R <= nacc) {
        const unsigned long long last = __ballot((ok & me) != 0 && arank == R - 1);
        pos += __ffsll((long long)last);  // lane index of the R-th accepted word + 1
        remaining = 0;
      } else {
        pos = (pos + 64 < 624) ? pos + 64 : 624;
        remaining = __ballot(mine && rrank >= nacc);
      }
    }
  }
  return pick;
}

template <typename QT>
__global__ void __launch_bounds__(64) k_greedy_policy_episodic(int B, int A, int H, int q_layers,
                                                              const int64_t* __restrict__ state_off,
                                                              const QT* __restrict__ Q, float* __restrict__ pi) {
  __shared__ uint32_t mt[624];
  __shared__ uint32_t out[624 + 64];  // padded: a 64-word window read may start at any position < 624
  __shared__ uint32_t tmp[64];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const int64_t soff = state_off[b];
  const int S = (int)(state_off[b + 1] - soff);
  const QT* q = Q + (int64_t)q_layers * soff * A;
  float* p = pi + (int64_t)H * soff * A;
  for (int i = lane; i < 624; i += 64) mt[i] = kMt42.w[i];
  out[624 + lane] = 0u;
  __builtin_amdgcn_wave_barrier();
  int pos = 624;  // wave-uniform: stream position inside the current block (624 = exhausted)
  const int64_t nrows = (int64_t)H * S;
  // rows live in registers (A <= 8: all loads of a row in flight together) and the NEXT chunk is fetched before the
  // current one is walked, so the single wave never waits for HBM inside the sequential part
  constexpr int AR = 8;
  if (A <= AR) {
    QT cur[AR], nxt[AR];
    auto fetch = [&](int64_t r, QT (&dst)[AR]) {
#pragma unroll
      for (int a = 0; a < AR; ++a) dst[a] = (r < nrows && a < A) ? q[r * A + a] : (QT)0;
    };
    fetch(lane, cur);
    for (int64_t r0 = 0; r0 < nrows; r0 += 64) {
      const int64_t r = r0 + lane;
      const bool valid = r < nrows;
      fetch(r + 64, nxt);
      QT m = cur[0];
#pragma unroll
      for (int a = 1; a < AR; ++a) m = (a < A && cur[a] > m) ? cur[a] : m;
      int n_tie = 0;
#pragma unroll
      for (int a = 0; a < AR; ++a) n_tie += (a < A && cur[a] == m) ? 1 : 0;
      if (!valid) n_tie = 0;
      const int pick = tie_break_chunk(n_tie, mt, out, tmp, pos, lane);
      if (valid) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < AR; ++a) {
          if (a < A) {
            const bool tie = cur[a] == m;
            p[r * A + a] = (tie && k == pick) ? 1.0f : 0.0f;
            k += tie ? 1 : 0;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < AR; ++a) cur[a] = nxt[a];
    }
    return;
  }
  for (int64_t r0 = 0; r0 < nrows; r0 += 64) {
    const int64_t r = r0 + lane;
    const bool valid = r < nrows;
    const QT* row = q + r * A;
    int n_tie = 0;
    QT m = 0;
    if (valid) {
      m = row[0];
      for (int a = 1; a < A; ++a) m = (row[a] > m) ? row[a] : m;
      for (int a = 0; a < A; ++a) n_tie += (row[a] == m) ? 1 : 0;
    }
    const int pick = tie_break_chunk(n_tie, mt, out, tmp, pos, lane);
    if (valid) {
      for (int a = 0, k = 0; a < A; ++a) {
        const bool tie = row[a] == m;
        p[r * A + a] = (tie && k == pick) ? 1.0f : 0.0f;
        k += tie ? 1 : 0;
      }
    }
  }
}

// V[0, :] of every instance (row 0 of its [H+1][S_b] block) packed contiguously for one device-to-host copy
// `V0` may be page-locked host memory (the logged loop: no copy kernel per row); `snap` (nullable, host memory as well)
// receives last_start / prev_start / hstep of every instance, [3][B]
__global__ void __launch_bounds__(256) k_gather_v0(int B, int H, const int64_t* __restrict__ state_off,
                                                   const float* __restrict__ V, float* __restrict__ V0,
                                                   const int32_t* __restrict__ last_start, const int32_t* __restrict__ prev_start,
                                                   const int32_t* __restrict__ hstep, int32_t* __restrict__ snap) {
  const int b = blockIdx.x;
  const int64_t so = state_off[b];
  const int S = (int)(state_off[b + 1] - so);
  for (int s = threadIdx.x; s < S; s += blockDim.x) V0[so + s] = V[(int64_t)(H + 1) * so + s];
  if (snap && threadIdx.x == 0) {
    snap[b] = last_start[b];
    snap[B + b] = prev_start[b];
    snap[2 * B + b] = hstep[b];
  }
}


// ---------------------------------------------------------------------------------------------------
// Continuous (average-reward) optimistic Q-learning, reference
//   colosseum/agent/agents/infinite_horizon/q_learning.py:47-112 (_QValuesModel.step_update, Wei et al. 2020)
// with the same greedy actor.  Same discipline as above: numpy's scalar arithmetic, dtype promotion included.
// ---------------------------------------------------------------------------------------------------
struct QlcArgs {
  double min_at;                  // python float (or int 0): `min_at if min_at > 0.009 else 0`
  double four_span;               // 4 * span_approx (python float product)
  double log_term;                // np.log(2 * optimization_horizon / confidence)
  const double* Hh;               // [B] np.float64 horizon approximation h_weight * get_H(...)
  const double* gamma;            // [B] 1 - 1 / H
  const int64_t* q_off;           // [B] = state_off[b] * A
  int32_t* N;                     // [rows] starts at 0
  // `np.zeros(..., np.float32) + self.H` with self.H a numpy float64 scalar is a FLOAT64 array under NEP 50
  // (numpy >= 2, the semantics the goldens were produced with): the tables and every operation are float64.
  double* Q;                      // starts at H
  double* Qmain;                  // starts at H
  double* V;                      // [states] starts at H
  uint32_t* mt;
  int32_t* mt_pos;
};

template <bool RC>
__global__ void __launch_bounds__(256) k_qlearn_continuous(EnvTables t, QlcArgs q, int64_t n_steps,
                                                           const uint8_t* __restrict__ train_mask,
                                                           int8_t* __restrict__ act_trace, double* __restrict__ cum_reward,
                                                           RewardCache rc, int resume, double* __restrict__ cum_host) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  int64_t step0 = 0;
  bool pending = false, parked = false;
  if (RC && resume) {
    const long long left = rc.left[b];
    if (left == 0) return;
    step0 = n_steps - left;
    pending = rc.pend_e[b] >= 0;
  }
  const int64_t soff = t.state_off[b], ebase = t.entry_base[b];
  const int A = t.A;
  const uint2 key = t.philox_key ? t.philox_key[b] : make_uint2(0, 0);
  int32_t cur = t.cur[b], h = t.hstep[b];
  unsigned long long nt = t.n_trans[b];
  double* Q = q.Q + q.q_off[b];
  double* QM = q.Qmain + q.q_off[b];
  int32_t* N = q.N + q.q_off[b];
  double* V = q.V + soff;
  uint32_t* mt = q.mt + (int64_t)b * 624;
  int32_t* mtp = q.mt_pos + b;
  const double Hh = q.Hh[b], gamma = q.gamma[b];
  const bool train = train_mask ? train_mask[b] != 0 : true;
  double sum = cum_reward[b];
  int64_t step = step0;
  for (; step < n_steps; ++step) {
    // loads issued as early as their addresses are known (see k_qlearn_episodic): row descriptors of all actions next to
    // the Q row, the visit count right after the action, both MT19937 words of the draw together
    constexpr int AM = 4;
    const bool fastA = A <= AM;
    double qv[AM];
    RowDesc rd[AM];
    int action = 0;
    int32_t s_t, obs;
    double reward;
    if (RC && pending) {  // the step this lane parked in: transition committed, reward and update outstanding
      s_t = rc.pend_prev[b];
      action = rc.pend_act[b];
      obs = cur;
      if (fastA) {
#pragma unroll
        for (int a = 0; a < AM; ++a) qv[a] = (a < A) ? Q[(int64_t)s_t * A + a] : -INFINITY;
      }
    } else {
      const double* qrow = Q + (int64_t)cur * A;
      if (fastA) {
        const RowDesc* rp = t.row + (soff + cur) * A;
#pragma unroll
        for (int a = 0; a < AM; ++a) {
          qv[a] = (a < A) ? qrow[a] : -INFINITY;
          if (a < A) rd[a] = rp[a];
        }
      }
      double qmax;
      int n_tie = 0;
      if (fastA) {
        qmax = qv[0];
#pragma unroll
        for (int a = 1; a < AM; ++a) qmax = fmax(qmax, qv[a]);
#pragma unroll
        for (int a = 0; a < AM; ++a) n_tie += (a < A && qv[a] == qmax) ? 1 : 0;
      } else {
        qmax = qrow[0];
        for (int a = 1; a < A; ++a) qmax = fmax(qmax, qrow[a]);
        for (int a = 0; a < A; ++a) n_tie += (qrow[a] == qmax) ? 1 : 0;
      }
      int pick = 0;
      if (n_tie > 1) {
        const uint32_t mx = (uint32_t)(n_tie - 1);
        uint32_t mask = mx;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int pos = *mtp;
        uint32_t v;
        do { v = mt_next_word(mt, pos) & mask; } while (v > mx);
        *mtp = pos;
        pick = (int)v;
      }
      if (fastA) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < AM; ++a)
          if (a < A && qv[a] == qmax) { if (k == pick) action = a; ++k; }
      } else {
        for (int a = 0, k = 0; a < A; ++a)
          if (qrow[a] == qmax) { if (k == pick) action = a; ++k; }
      }
      if (act_trace) act_trace[step * t.B + b] = (int8_t)action;
      s_t = cur;
    }
    const int64_t idx = (int64_t)s_t * A + action;
    const int32_t n_pre = train ? N[idx] : 0;   // its load rides under the transition's round trips
    const bool was_pending = RC && pending;
    RowDesc dsel;
    if (fastA && !was_pending) {
      dsel = rd[0];
#pragma unroll
      for (int a = 1; a < AM; ++a)
        if (a == action) dsel = rd[a];
    }
    if (RC) {
      double rraw = 0.0;
      int64_t e;
      if (pending) { e = rc.pend_e[b]; pending = false; }
      else if (fastA) { env_transition_desc<true>(t, soff, ebase, key, cur, h, nt, action, obs, rraw, e, dsel); ++nt; }
      else env_transition(t, soff, ebase, key, cur, h, nt, action, obs, rraw, e);  // continuous: never terminates
      if (!rc_fetch(t.sp_rkind, rc, e, rraw)) {
        rc_park(rc, b, e, s_t, action);
        parked = true;
        break;
      }
      reward = rraw * t.rscale - t.rmin;
    } else if (fastA) {
      const unsigned long long n0 = nt;
      double rraw;
      int64_t e;
      env_transition_desc<true>(t, soff, ebase, key, cur, h, n0, action, obs, rraw, e, dsel);
      ++nt;
      if (t.sp_rkind && t.sp_rkind[e] == 1) rraw = philox_beta(t.sp_rp0[e], t.sp_rp1[e], n0, key, t.beta_gammas);  // throughput mode only
      reward = rraw * t.rscale - t.rmin;
    } else {
      env_step(t, soff, ebase, key, cur, h, nt, action, obs, reward);  // continuous: never terminates
    }
    sum += reward;
    if (train) {
      const int32_t n = n_pre + 1;
      N[idx] = n;
      const double x = (Hh + 1.0) / (Hh + (double)n);
      const double alpha = (x > q.min_at) ? x : q.min_at;  // max(min_at, x)
      const double b_t = q.four_span * sqrt(Hh / (double)n * q.log_term);
      const double target = (reward + gamma * V[obs]) + b_t;
      double qold;
      if (fastA) {
        qold = qv[0];
#pragma unroll
        for (int a = 1; a < AM; ++a)
          if (a == action) qold = qv[a];
      } else {
        qold = Q[idx];
      }
      const double qm = (1.0 - alpha) * qold + alpha * target;
      QM[idx] = qm;
      Q[idx] = (qm < qold) ? qm : qold;  // min(Q, Q_main)
      const double* qr = Q + (int64_t)obs * A;
      double m2 = qr[0];
      for (int a = 1; a < A; ++a) m2 = fmax(m2, qr[a]);
      V[obs] = m2;  // self.V[s_tp1] = self.Q[s_tp1].max()
    }
  }
  t.cur[b] = cur;
  t.hstep[b] = h;
  t.n_trans[b] = nt;
  cum_reward[b] = sum;
  if (cum_host) cum_host[b] = sum;
  if (RC) {
    rc.left[b] = parked ? (long long)(n_steps - step) : 0;
    if (!parked) rc.pend_e[b] = -1;
  }
}
