// cmdp_kernels.h -- HIP kernels of libcmdp (gfx950 / CDNA4).  Included once by cmdp.hip.
//
// K1  k_reset / k_step / k_rollout : agent<->MDP interaction, one lane per instance, tables in HBM/L2
// K2  k_dp_block<...>              : Jacobi sweeps (discounted VI / PE, diameter), one workgroup per
//                                    (instance[, target]); V ping-pong and, when it fits, the CSR in LDS
// K3  k_dp_wave_gs<...>            : Gauss-Seidel sweeps, one wavefront per (instance[, target]), V in LDS
// K4  k_episodic                   : backward induction, one workgroup per instance
// K6  k_value_norm                 : value-norm reduction
//
// No MFMA anywhere: rows have <= ~8 non-zeros; everything is gather / scan / max-reduce.
// Float32 DP arithmetic uses __fmul_rn/__fadd_rn (never contracted into FMA): the reference's
// accumulations round the product and the sum separately.
#pragma once
#include "cmdp_device.h"

struct RowDesc {       // 16 B, one per (instance, state, action)
  int32_t first;       // first entry of the row, relative to the instance's entry base
  int32_t n;           // number of successors (1 = deterministic shortcut, no draw)
  int32_t next_if_det; // successor when n == 1
  int32_t mt_slot;     // MT19937 slot of the row's sampler (MT_COMPAT, n > 1), else -1
};

struct EnvTables {
  int32_t B, A, H, rng_mode;
  double rscale, rmin;               // reward = r * rscale - rmin
  const int64_t* state_off;          // [B+1]
  const int64_t* entry_base;         // [B]
  const RowDesc* row;                // [R]
  const int32_t* sp_next;            // [E]
  const double* sp_cum;              // [E]
  const double* sp_reward;           // [E]
  const int64_t* start_off;          // [B+1]
  const int32_t* start_state;        // [NS]
  const double* start_cum;           // [NS]
  const int32_t* start_slot;         // [B]  MT slot of the start sampler or -1
  const uint2* philox_key;           // [B]
  uint32_t* mt;                      // [n_slots][624]
  int32_t* mt_pos;                   // [n_slots]
  // dynamic state
  int32_t* cur;                      // [B]
  int32_t* hstep;                    // [B]
  uint8_t* need_reset;               // [B]
  unsigned long long* n_trans;       // [B]
  unsigned long long* n_reset;       // [B]
  int32_t* visits_s;                 // [NSTATES]
  int32_t* visits_sa;                // [R]
};

// ---------------------------------------------------------------------------------------------------
__global__ void k_mt_seed(uint32_t* __restrict__ mt, int32_t* __restrict__ mt_pos, const int32_t* __restrict__ seeds,
                          int64_t n_slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  mt_seed_python_int(mt + i * 624, (uint32_t)seeds[i]);
  mt_pos[i] = 0;
}

// BaseMDP.reset (reference colosseum/mdp/base.py:1268-1277) for one instance
__device__ __forceinline__ int32_t env_reset(const EnvTables& t, int b, int64_t soff, uint2 key,
                                             unsigned long long& n_reset) {
  const int64_t lo = t.start_off[b];
  const int n = (int)(t.start_off[b + 1] - lo);
  int idx = 0;
  if (n > 1) {
    double u;
    if (t.rng_mode == 0) {
      const int slot = t.start_slot[b];
      u = mt_random(t.mt + (int64_t)slot * 624, t.mt_pos + slot);
    } else {
      uint32_t w[4];
      philox4x32_10((uint32_t)n_reset, (uint32_t)(n_reset >> 32), 1u, 0u, key.x, key.y, w);
      u = u53(w[0], w[1]);
    }
    idx = choose_index(t.start_cum + lo, n, u);
  }
  n_reset++;
  const int32_t s = t.start_state[lo + idx];
  t.visits_s[soff + s] += 1;
  return s;
}

// BaseMDP.step (reference colosseum/mdp/base.py:1293-1317) for one instance.
// Returns the step type (1 MID, 2 LAST); `action` < 0 requests the Philox random-policy action.
__device__ __forceinline__ int env_step(const EnvTables& t, int64_t soff, int64_t ebase, uint2 key, int32_t& cur,
                                        int32_t& h, unsigned long long& n_trans, int action, int32_t& obs,
                                        double& reward) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  if (t.rng_mode == 1 || action < 0) {
    philox4x32_10((uint32_t)n_trans, (uint32_t)(n_trans >> 32), 0u, 0u, key.x, key.y, w);
  }
  if (action < 0) action = (int)(((uint64_t)w[2] * (uint64_t)t.A) >> 32);
  n_trans++;
  h += 1;
  const int64_t r = (soff + cur) * t.A + action;
  const RowDesc d = t.row[r];
  int64_t e = ebase + d.first;
  int32_t nxt = d.next_if_det;
  if (d.n > 1) {  // NextStateSampler.sample (custom_samplers.py:59-72)
    double u;
    if (t.rng_mode == 0) u = mt_random(t.mt + (int64_t)d.mt_slot * 624, t.mt_pos + d.mt_slot);
    else u = u53(w[0], w[1]);
    e += choose_index(t.sp_cum + e, d.n, u);
    nxt = t.sp_next[e];
  }
  // visit counts on the arrival node with the action taken at the departure node (base.py:1302-1303)
  t.visits_s[soff + nxt] += 1;
  t.visits_sa[(soff + nxt) * t.A + action] += 1;
  reward = t.sp_reward[e] * t.rscale - t.rmin;  // `r * (max - min) - min`, base.py:1205-1207
  cur = nxt;
  if (t.H > 0 && h >= t.H) {
    obs = -1;
    return 2;
  }
  obs = nxt;
  return 1;
}

__global__ void k_reset(EnvTables t, const uint8_t* __restrict__ mask, int32_t* __restrict__ obs_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  if (mask && !mask[b]) return;
  unsigned long long nr = t.n_reset[b];
  const int32_t s = env_reset(t, b, t.state_off[b], t.philox_key ? t.philox_key[b] : make_uint2(0, 0), nr);
  t.n_reset[b] = nr;
  t.cur[b] = s;
  t.hstep[b] = 0;
  t.need_reset[b] = 0;
  if (obs_out) obs_out[b] = s;
}

__global__ void k_any_needs_reset(const uint8_t* __restrict__ need_reset, int B, int32_t* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && need_reset[b]) atomicOr(flag, 1);
}

__global__ void k_check_actions(const int32_t* __restrict__ actions, int B, int A, int32_t* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && (actions[b] < 0 || actions[b] >= A)) atomicOr(flag, 2);
}

__global__ void k_step(EnvTables t, const int32_t* __restrict__ actions, int auto_reset, int32_t* __restrict__ obs,
                       double* __restrict__ reward, uint8_t* __restrict__ step_type) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  const int64_t soff = t.state_off[b];
  if (t.need_reset[b]) {  // only reachable with auto_reset (the host pre-checks otherwise): step() == reset()
    unsigned long long nr = t.n_reset[b];
    const int32_t s = env_reset(t, b, soff, t.philox_key ? t.philox_key[b] : make_uint2(0, 0), nr);
    t.n_reset[b] = nr;
    t.cur[b] = s;
    t.hstep[b] = 0;
    t.need_reset[b] = 0;
    obs[b] = s;
    reward[b] = 0.0;
    step_type[b] = 0;
    return;
  }
  int32_t cur = t.cur[b], h = t.hstep[b], o;
  unsigned long long nt = t.n_trans[b];
  double r;
  const int ty = env_step(t, soff, t.entry_base[b], t.philox_key ? t.philox_key[b] : make_uint2(0, 0), cur, h, nt,
                          actions[b], o, r);
  t.cur[b] = cur;
  t.hstep[b] = h;
  t.n_trans[b] = nt;
  t.need_reset[b] = (ty == 2);
  obs[b] = o;
  reward[b] = r;
  step_type[b] = (uint8_t)ty;
}

// The env side of MDPLoop.run's loop (reference colosseum/experiment/agent_mdp_interaction.py:238-298),
// fused: n_steps transitions per instance, every termination followed at once by reset().
// POLICY 0: Philox random action; 1: actions[t][B] (int8).
template <int POLICY, bool TRACE>
__global__ void __launch_bounds__(256) k_rollout(EnvTables t, const int8_t* __restrict__ actions, int64_t n_steps,
                                                 double* __restrict__ reward_sum, int32_t* __restrict__ last_obs,
                                                 int32_t* __restrict__ tr_obs, double* __restrict__ tr_rew,
                                                 uint8_t* __restrict__ tr_type) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  const int64_t soff = t.state_off[b], ebase = t.entry_base[b];
  const uint2 key = t.philox_key ? t.philox_key[b] : make_uint2(0, 0);
  int32_t cur = t.cur[b], h = t.hstep[b], obs = cur;
  unsigned long long nt = t.n_trans[b], nr = t.n_reset[b];
  double sum = 0.0;
  for (int64_t s = 0; s < n_steps; ++s) {
    const int a = (POLICY == 1) ? (int)actions[s * t.B + b] : -1;
    double r;
    const int ty = env_step(t, soff, ebase, key, cur, h, nt, a, obs, r);
    sum += r;
    if (TRACE) {
      if (tr_obs) tr_obs[s * t.B + b] = obs;
      if (tr_rew) tr_rew[s * t.B + b] = r;
      if (tr_type) tr_type[s * t.B + b] = (uint8_t)ty;
    }
    if (ty == 2) {
      cur = env_reset(t, b, soff, key, nr);
      h = 0;
      obs = cur;
    }
  }
  t.cur[b] = cur;
  t.hstep[b] = h;
  t.n_trans[b] = nt;
  t.n_reset[b] = nr;
  if (reward_sum) reward_sum[b] = sum;
  if (last_obs) last_obs[b] = obs;
}

// ===================================================================================================
// Dynamic programming
// ===================================================================================================
struct DpTables {
  int32_t B, A;
  const int64_t* state_off;  // [B+1]
  const int64_t* csr_ptr;    // [R+1] global offsets
  const int32_t* csr_col;
  const float* csr_val;
  const float* R;            // [R] (already the override when one was given)
  const float* pi;           // [R] or null
  const int64_t* unit_off;   // DIAM: [B+1] prefix of targets (== state_off); else null
  float gamma;
  double eps;
  double max_abs;            // <= 0: off
  int64_t max_sweeps;
  float* Q;                  // [R]        (null for DIAM)
  float* V;                  // [NSTATES]  (null for DIAM)
  int64_t* sweeps;           // [units] or null
  float* per_target;         // DIAM: [NSTATES] = -min V
  int32_t* status;           // [units] 0 ok, -5 max sweeps, -7 max value
};

enum { DP_VI = 0, DP_PE = 1 };

// instance-relative int32 view of the global int64 row-pointer array
struct GPtr {
  const int64_t* p;
  int64_t z;
  __device__ __forceinline__ int operator[](int i) const { return (int)(p[i] - z); }
};

// unit -> (instance, target) for the diameter launches
__device__ __forceinline__ void unit_to_instance(const DpTables& t, int64_t unit, int& b, int& target) {
  if (!t.unit_off) { b = (int)unit; target = -1; return; }
  int lo = 0, hi = t.B;  // largest b with unit_off[b] <= unit
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.unit_off[mid] <= unit) lo = mid; else hi = mid;
  }
  b = lo;
  target = (int)(unit - t.unit_off[lo]);
}

// One Bellman backup of state s from value vector Vin.
//   JACOBI VI : q = R + gamma * sum_k val_k * V[col_k]        (infinite_horizon.py:154)
//   otherwise : q = R + sum_k (gamma * val_k) * V[col_k]       (infinite_horizon.py:134,179,200)
// DIAM: target rows are the absorbing row {target: 1} with R = 0, all other rows have R = -1
//       (hardness/measures/diameter.py:85-90).
template <int MODE, bool JACOBI, bool DIAM, bool WRITE_Q, typename PtrT, typename ColT, typename ValT, typename RT>
__device__ __forceinline__ float backup_state(int s, int A, int target, PtrT ptr, ColT col, ValT val, RT Rv,
                                              const float* __restrict__ pi, const float* __restrict__ Vin, float gamma,
                                              float* __restrict__ Qout) {
  float v = 0.0f;
  for (int a = 0; a < A; ++a) {
    const int r = s * A + a;
    float acc = 0.0f;
    float rew;
    if (DIAM && s == target) {
      const float c = (MODE == DP_VI && JACOBI) ? 1.0f : __fmul_rn(gamma, 1.0f);
      acc = __fadd_rn(acc, __fmul_rn(c, Vin[target]));
      rew = 0.0f;
    } else {
      const int lo = ptr[r], hi = ptr[r + 1];
      for (int k = lo; k < hi; ++k) {
        const float c = (MODE == DP_VI && JACOBI) ? val[k] : __fmul_rn(gamma, val[k]);
        acc = __fadd_rn(acc, __fmul_rn(c, Vin[col[k]]));
      }
      rew = DIAM ? -1.0f : Rv[r];
    }
    const float q = (MODE == DP_VI && JACOBI) ? __fadd_rn(rew, __fmul_rn(gamma, acc)) : __fadd_rn(rew, acc);
    if (WRITE_Q) Qout[r] = q;
    if (MODE == DP_VI) {
      v = (a == 0) ? q : fmaxf(v, q);
    } else {
      const float qp = __fmul_rn(q, pi[r]);
      v = (a == 0) ? qp : __fadd_rn(v, qp);
    }
  }
  return v;
}

// K2: Jacobi sweeps, one workgroup per unit.  Dynamic LDS: V ping-pong [2][S], reduction scratch, and
// (CSR_LDS) ptr[S*A+1] (instance-relative int32), col, val, R.
template <int MODE, bool DIAM, bool CSR_LDS>
__global__ void __launch_bounds__(256) k_dp_block(DpTables t) {
  extern __shared__ __align__(16) unsigned char smem[];
  int b, target;
  unit_to_instance(t, blockIdx.x, b, target);
  const int A = t.A;
  const int64_t soff = t.state_off[b];
  const int S = (int)(t.state_off[b + 1] - soff);
  const int rows = S * A;
  const int64_t row0 = soff * A;
  const int64_t nz0 = t.csr_ptr[row0];
  const int nnz = (int)(t.csr_ptr[row0 + rows] - nz0);
  const int tid = threadIdx.x, nth = blockDim.x, wave = tid >> 6, lane = tid & 63, nwaves = nth >> 6;

  float* Va = reinterpret_cast<float*>(smem);
  float* Vb = Va + S;
  float* red = Vb + S;                 // [2][nwaves][2]
  unsigned char* p = reinterpret_cast<unsigned char*>(red + 4 * nwaves);
  int32_t* lptr = reinterpret_cast<int32_t*>(p);
  int32_t* lcol = lptr + (rows + 1);
  float* lval = reinterpret_cast<float*>(lcol + nnz);
  float* lR = lval + nnz;
  if (CSR_LDS) {
    for (int i = tid; i <= rows; i += nth) lptr[i] = (int32_t)(t.csr_ptr[row0 + i] - nz0);
    for (int i = tid; i < nnz; i += nth) { lcol[i] = t.csr_col[nz0 + i]; lval[i] = t.csr_val[nz0 + i]; }
    if (!DIAM) for (int i = tid; i < rows; i += nth) lR[i] = t.R[row0 + i];
  }
  for (int i = tid; i < S; i += nth) { Va[i] = 0.0f; Vb[i] = 0.0f; }
  __syncthreads();

  const float* pi = (MODE == DP_PE) ? t.pi + row0 : nullptr;
  float* Vold = Va;
  float* Vnew = Vb;
  int64_t it = 0;
  int status = -5;
  while (it < t.max_sweeps) {
    ++it;
    float dmax = 0.0f, vabs = 0.0f;
    for (int s = tid; s < S; s += nth) {
      float v;
      if (CSR_LDS) {
        v = backup_state<MODE, true, DIAM, false>(s, A, target, lptr, lcol, lval, lR, pi, Vold, t.gamma, nullptr);
      } else {
        GPtr gp{t.csr_ptr + row0, nz0};
        v = backup_state<MODE, true, DIAM, false>(s, A, target, gp, t.csr_col + nz0, t.csr_val + nz0, t.R + row0, pi,
                                                  Vold, t.gamma, nullptr);
      }
      Vnew[s] = v;
      dmax = fmaxf(dmax, fabsf(Vold[s] - v));
      vabs = fmaxf(vabs, fabsf(v));
    }
    dmax = wave_max(dmax);
    vabs = wave_max(vabs);
    float* rbuf = red + (it & 1) * 2 * nwaves;
    if (lane == 0) { rbuf[2 * wave] = dmax; rbuf[2 * wave + 1] = vabs; }
    __syncthreads();
    float diff = 0.0f, vmax = 0.0f;
    for (int w = 0; w < nwaves; ++w) { diff = fmaxf(diff, rbuf[2 * w]); vmax = fmaxf(vmax, rbuf[2 * w + 1]); }
    float* tmp = Vold; Vold = Vnew; Vnew = tmp;  // Vold = newest values, Vnew = the vector they were computed from
    if (t.max_abs > 0.0 && (double)vmax > t.max_abs) { status = -7; break; }
    if ((double)diff < t.eps) { status = 0; break; }
  }
  // outputs
  const int64_t unit = blockIdx.x;
  if (tid == 0) {
    t.status[unit] = status;
    if (t.sweeps) t.sweeps[unit] = it;
  }
  if (DIAM) {
    float mn = 3.0e38f;
    for (int s = tid; s < S; s += nth) mn = fminf(mn, Vold[s]);
    mn = wave_min(mn);
    __syncthreads();
    if (lane == 0) red[wave] = mn;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
      for (int w = 1; w < nwaves; ++w) m = fminf(m, red[w]);
      t.per_target[soff + target] = -m;
    }
  } else {
    // Q of the last sweep is a function of the vector that sweep read (now in Vnew); recomputing it
    // repeats the identical arithmetic.
    for (int s = tid; s < S; s += nth) {
      t.V[soff + s] = Vold[s];
      if (it > 0) {
        if (CSR_LDS) {
          backup_state<MODE, true, false, true>(s, A, -1, lptr, lcol, lval, lR, pi, Vnew, t.gamma, t.Q + row0);
        } else {
          GPtr gp{t.csr_ptr + row0, nz0};
          backup_state<MODE, true, false, true>(s, A, -1, gp, t.csr_col + nz0, t.csr_val + nz0, t.R + row0, pi, Vnew,
                                                t.gamma, t.Q + row0);
        }
      }
    }
  }
}

// K3: Gauss-Seidel sweeps (numba paths of the reference), one wavefront per unit, V in LDS, states in
// order.  Lane a < A backs up action a of the current state; later states see the updated V.
template <int MODE, bool DIAM>
__global__ void __launch_bounds__(64) k_dp_wave_gs(DpTables t) {
  extern __shared__ __align__(16) unsigned char smem[];
  int b, target;
  unit_to_instance(t, blockIdx.x, b, target);
  const int A = t.A;
  const int64_t soff = t.state_off[b];
  const int S = (int)(t.state_off[b + 1] - soff);
  const int64_t row0 = soff * A;
  const int lane = threadIdx.x;
  float* V = reinterpret_cast<float*>(smem);
  for (int i = lane; i < S; i += 64) V[i] = 0.0f;
  __syncthreads();
  const int64_t* ptr = t.csr_ptr + row0;
  const int32_t* col = t.csr_col;
  const float* val = t.csr_val;
  const bool active = lane < A;
  int64_t it = 0;
  int status = -5;
  while (it < t.max_sweeps) {
    ++it;
    float dmax = 0.0f;
    bool too_big = false;
    for (int s = 0; s < S; ++s) {
      float q = -3.0e38f, qp = 0.0f;
      if (active) {
        const int r = s * A + lane;
        float acc = 0.0f, rew;
        if (DIAM && s == target) {
          acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(t.gamma, 1.0f), V[target]));
          rew = 0.0f;
        } else {
          const int64_t lo = ptr[r], hi = ptr[r + 1];
          for (int64_t k = lo; k < hi; ++k) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(t.gamma, val[k]), V[col[k]]));
          rew = DIAM ? -1.0f : t.R[row0 + r];
        }
        q = __fadd_rn(rew, acc);
        if (!DIAM) t.Q[row0 + r] = q;
        if (MODE == DP_PE) qp = __fmul_rn(q, t.pi[row0 + r]);
      }
      float v;
      if (MODE == DP_VI) {
        v = wave_max(q);
      } else {
        v = __shfl(qp, 0, 64);
        for (int a = 1; a < A; ++a) v = __fadd_rn(v, __shfl(qp, a, 64));
      }
      __syncthreads();  // every lane has read V for this state before it changes
      if (lane == 0) {
        const float old = V[s];
        V[s] = v;
        dmax = fmaxf(dmax, fabsf(old - v));
        if (t.max_abs > 0.0 && fabs((double)v) > t.max_abs) too_big = true;
      }
      __syncthreads();
    }
    const float diff = __shfl(dmax, 0, 64);
    if (__shfl((int)too_big, 0, 64)) { status = -7; break; }
    if ((double)diff < t.eps) { status = 0; break; }
  }
  const int64_t unit = blockIdx.x;
  if (lane == 0) {
    t.status[unit] = status;
    if (t.sweeps) t.sweeps[unit] = it;
  }
  if (DIAM) {
    float mn = 3.0e38f;
    for (int i = lane; i < S; i += 64) mn = fminf(mn, V[i]);
    mn = wave_min(mn);
    if (lane == 0) t.per_target[soff + target] = -mn;
  } else {
    for (int i = lane; i < S; i += 64) t.V[soff + i] = V[i];
  }
}

// K4: finite-horizon backward induction (reference colosseum/dynamic_programming/finite_horizon.py:11-42),
// one workgroup per instance; V[h+1] is staged in LDS, Q[h]/V[h] go straight to HBM.
template <int MODE>
__global__ void __launch_bounds__(256) k_episodic(DpTables t, int H, float* __restrict__ Qout, float* __restrict__ Vout) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x;
  const int A = t.A;
  const int64_t soff = t.state_off[b];
  const int S = (int)(t.state_off[b + 1] - soff);
  const int64_t row0 = soff * A;
  const int64_t nz0 = t.csr_ptr[row0];
  const int tid = threadIdx.x, nth = blockDim.x;
  float* Va = reinterpret_cast<float*>(smem);
  float* Vb = Va + S;
  float* Q = Qout + (int64_t)(H + 1) * row0;
  float* V = Vout + (int64_t)(H + 1) * soff;
  for (int i = tid; i < S; i += nth) { Va[i] = 0.0f; V[(int64_t)H * S + i] = 0.0f; }
  for (int i = tid; i < S * A; i += nth) Q[(int64_t)H * S * A + i] = 0.0f;
  __syncthreads();
  GPtr gp{t.csr_ptr + row0, nz0};
  float* Vnext = Va;
  float* Vcur = Vb;
  for (int h = H - 1; h >= 0; --h) {
    const float* pi = (MODE == DP_PE) ? t.pi + (int64_t)H * row0 + (int64_t)h * S * A : nullptr;
    float* Qh = Q + (int64_t)h * S * A;
    for (int s = tid; s < S; s += nth) {
      // gamma == 1: `Q[h, s] = R[s] + T[s] @ V[h + 1]`; products 1*val are exact
      const float v = backup_state<MODE, true, false, true>(s, A, -1, gp, t.csr_col + nz0, t.csr_val + nz0, t.R + row0,
                                                            pi, Vnext, 1.0f, Qh);
      Vcur[s] = v;
      V[(int64_t)h * S + s] = v;
    }
    __syncthreads();
    float* tmp = Vnext; Vnext = Vcur; Vcur = tmp;
  }
}

// K6: calculate_norm_discounted (reference colosseum/hardness/measures/value_norm.py:83-87):
//   Ev[j,a] = sum_k T[j,a,k] V[k];  out = max_{i,a} sqrt( sum_j T[i,a,j] (V[j] - Ev[j,a])^2 )
__global__ void __launch_bounds__(256) k_value_norm(DpTables t, const float* __restrict__ V, float* __restrict__ Ev,
                                                    float* __restrict__ out) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int A = t.A;
  const int64_t soff = t.state_off[b];
  const int S = (int)(t.state_off[b + 1] - soff);
  const int rows = S * A;
  const int64_t row0 = soff * A;
  const int tid = threadIdx.x, nth = blockDim.x;
  const float* Vb = V + soff;
  for (int r = tid; r < rows; r += nth) {
    float acc = 0.0f;
    for (int64_t k = t.csr_ptr[row0 + r]; k < t.csr_ptr[row0 + r + 1]; ++k)
      acc = __fadd_rn(acc, __fmul_rn(t.csr_val[k], Vb[t.csr_col[k]]));
    Ev[row0 + r] = acc;
  }
  __syncthreads();  // Ev of this instance was written by this workgroup only
  float best = 0.0f;
  for (int r = tid; r < rows; r += nth) {
    const int a = r % A;
    float acc = 0.0f;
    for (int64_t k = t.csr_ptr[row0 + r]; k < t.csr_ptr[row0 + r + 1]; ++k) {
      const int j = t.csr_col[k];
      const float d = __fsub_rn(Vb[j], Ev[row0 + (int64_t)j * A + a]);
      acc = __fadd_rn(acc, __fmul_rn(t.csr_val[k], __fmul_rn(d, d)));
    }
    best = fmaxf(best, __fsqrt_rn(acc));
  }
  best = wave_max(best);
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int w = 1; w < (nth >> 6); ++w) m = fmaxf(m, red[w]);
    out[b] = m;
  }
}
