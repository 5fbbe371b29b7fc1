/* cmdp.h -- C ABI of libcmdp.so, the MI355X-native batched tabular-MDP engine.
 *
 * The reference (MichelangeloConserva/Colosseum, pure Python + numba) has no FFI of its own; the
 * drop-in boundary is its Python call surface.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference tree).  The library is loaded with
 * ctypes.CDLL by colosseum_amd/_lib.py; INTEGRATION.md shows the binding a reference maintainer
 * would add.
 *
 * Conventions: plain C types only; every function returns 0 (CMDP_OK) or a negative error code and
 * never throws; cmdp_last_error() gives the message of the calling thread's last failure; host
 * buffers are borrowed for the duration of the call; a cmdp_t owns its device memory and one HIP
 * stream, is bound to the device current at cmdp_create time, and is not thread-safe; calls are
 * synchronous on return unless stated otherwise.
 *
 * Batch layout: B independent MDP instances (environment parameterisation x seed), instance b having
 * S_b states and the common A actions.  state_off[b] = sum_{i<b} S_i.  "row" r of instance b is
 * (state_off[b] + s) * A + a for the action a an agent passes to step().  All per-state arrays are
 * the instances' arrays concatenated ([state_off[B]]), all per-row arrays likewise ([state_off[B]*A]).
 */
#ifndef CMDP_H
#define CMDP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMDP_ABI_VERSION 1

enum {
  CMDP_OK = 0,
  CMDP_ERR_INVALID = -1,     /* bad argument (message says which)                                  */
  CMDP_ERR_HIP = -2,         /* a HIP runtime call failed                                           */
  CMDP_ERR_NEEDS_RESET = -3, /* step() on an instance whose episode ended: the reference's
                                `assert not self.necessary_reset`, colosseum/mdp/base.py:1290       */
  CMDP_ERR_UNSUPPORTED = -4, /* feature outside the built scope                                     */
  CMDP_ERR_MAX_ITER = -5,    /* DynamicProgrammingMaxIterationExceeded,
                                colosseum/dynamic_programming/utils.py:8-9                          */
  CMDP_ERR_NO_DEVICE = -6,   /* no HIP device visible                                               */
  CMDP_ERR_MAX_VALUE = -7,   /* |V| exceeded max_abs_value: the reference returns None,
                                colosseum/dynamic_programming/infinite_horizon.py:136-138           */
  CMDP_ERR_OVERFLOW = -8     /* a device visit counter (int32) could wrap during this call: refused before anything is
                                stepped -- read the counters out (cmdp_visits widens them to int64) and
                                cmdp_reset_visits first; the reference's Python ints do not wrap      */
};

/* Random-number discipline of the transition sampler. */
enum {
  /* One CPython MT19937 stream per stochastic (s,a) row and per stochastic start sampler, seeded
     like random.Random(seed) and consumed one random() per sample: reproduces the reference's
     NextStateSampler (colosseum/mdp/utils/custom_samplers.py:49-72) draw for draw. */
  CMDP_RNG_MT_COMPAT = 0,
  /* Philox-4x32-10 keyed by the instance (counter-based, nothing stored per sampler):
     counter (n, domain): domain 0, n = transition index -> 53-bit transition uniform (stochastic rows);
     domain 1, n = reset index -> start-state uniform; domain 2 -> random-policy actions (the reference's
     RandomActor, colosseum/agent/actors/random.py:34-47, draws numpy blocks of 50 000; this is the
     throughput stream): for A in {2, 4, 16, 256} the PACKED form -- block n / apb holds apb = 128 / log2(A)
     actions, action k = n % apb in bits [lg (k % apw), +lg) of word k / apw, apw = 32 / lg (A = 2: 128
     one-bit actions per block, least significant bit of word 0 first); for any other A, block n >> 2,
     word (n & 3), a = (word * A) >> 32. */
  CMDP_RNG_PHILOX = 1
};

/* Action source of cmdp_rollout. */
enum {
  CMDP_POLICY_RANDOM = 0,       /* uniform over A from the instance's Philox stream (config C2)      */
  CMDP_POLICY_HOST_ACTIONS = 1, /* policy_arg = const int8_t actions[n_steps][B]                      */
  CMDP_POLICY_GREEDY_Q = 2      /* policy_arg = const float Q: per instance [S_b][A] (continuous) or
                                   [H][S_b][A] at H*state_off[b]*A (episodic, row = in-episode time);
                                   the action is the first maximiser of the current row            */
};

/* Sweep scheme of the discounted solvers. */
enum {
  CMDP_SCHEME_AUTO = 0,   /* the reference's own size/density rule,
                             colosseum/dynamic_programming/infinite_horizon.py:25-36,53-64          */
  CMDP_SCHEME_JACOBI = 1, /* _discounted_value_iteration_sparse / _discounted_policy_evaluation_sparse */
  CMDP_SCHEME_GAUSS_SEIDEL = 2 /* _discounted_value_iteration / _discounted_policy_evaluation (numba) */
};

/* Device table layout of the transition sampler. */
enum {
  CMDP_LAYOUT_CSR = 0,   /* per-row successor lists (compact; the default)                           */
  CMDP_LAYOUT_DENSE = 1  /* per-instance dense float32 P[s,a,:] rows streamed from HBM, wavefront
                            prefix-sum CDF lookup: next = min{ j : sum_{i<=j} P[s,a,i] > u*total } in
                            STATE order (not the sampler's creation order), so stochastic rows draw
                            different -- equally distributed -- successors than CMDP_LAYOUT_CSR for the
                            same u.  Philox mode, cmdp_rollout only, needs both halves of the description
                            and S_max*A*4*round_up(S_max,256) bytes per instance.                      */
};

/* cmdp_desc.flags */
enum {
  /* Accept entries with sp_rkind != 0 (stochastic reward distributions) and report sp_reward -- which then holds
     the distribution MEAN -- as the reward of such transitions.  For callers that sample those rewards themselves
     (the reference-exact host sampler of colosseum_amd/mdp/reward_sampler.py) or only need expectations. */
  CMDP_FLAG_REWARD_MEANS = 1,
  /* REFERENCE-EXACT stochastic rewards for a whole batch: BaseMDP.sample_reward (colosseum/mdp/base.py:1187-1207) keeps a
     FIFO cache of 5000 samples per visited (node, action, next_node) triple, drawn from the MDP's single numpy stream
     `self._rng` when the triple is first needed and whenever its cache runs dry, so the draw order follows the
     trajectory.  With this flag the device serves rewards from such caches in HBM; an instance that needs a block it does
     not have parks (its transition committed, the step unfinished), the host draws the block from that instance's own
     stream with numpy's legacy Beta sampler (cmdp_set_reward_streams positions the streams; csrc/cmdp_reward_cache.h)
     and the kernel is relaunched -- inside cmdp_step, cmdp_rollout, cmdp_qlearning_run and cmdp_qlearning_run_logged.
     Needs sp_rp0 / sp_rp1 and the CSR layout; the transition streams are whatever rng_mode says (CMDP_RNG_MT_COMPAT for
     the reference's).  Rollouts run on the lane-per-instance kernel; cmdp_rollout_async is synchronous in this mode. */
  CMDP_FLAG_REWARD_CACHE = 2,
  /* CMDP_RNG_PHILOX with Beta rewards sampled on the device: EVERY Beta(a, b) as Ga / (Ga + Gb) with two Marsaglia-Tsang
     gammas -- the recipe of rounds 1-2.  Without the flag a Beta with a == 1 or b == 1 (nearly every triple of the
     reference's MDP families) is drawn by inverse CDF from one uniform: the same distribution from a tenth of the
     instructions, other numbers.  The flag exists so that earlier runs can be reproduced (csrc/cmdp_device.h philox_beta). */
  CMDP_FLAG_BETA_GAMMAS = 4
};

typedef struct cmdp cmdp_t;

/* Model description handed to cmdp_create.  Either half may be absent (all its pointers NULL):
 * the sampler half is what reset/step/rollout/visits need, the CSR half what the DP entry points
 * need.  Replaces the tables the reference builds in BaseMDP.instantiate_MDP
 * (colosseum/mdp/base.py:463-503) and get_transition_matrix_and_rewards
 * (colosseum/mdp/utils/mdp_creation.py:41-95). */
typedef struct cmdp_desc {
  int32_t n_instances;       /* B >= 1                                                               */
  int32_t n_actions;         /* A >= 1                                                               */
  int32_t horizon;           /* H > 0: episodic (EpisodicMDP.H); 0: continuous                       */
  int32_t rng_mode;          /* CMDP_RNG_*                                                           */
  int32_t layout;            /* CMDP_LAYOUT_*                                                        */
  int32_t flags;             /* CMDP_FLAG_*                                                          */
  double reward_min;         /* BaseMDP.rewards_range, applied as r*(max-min) - min                  */
  double reward_max;         /*   (sic, colosseum/mdp/base.py:1205-1207)                             */
  const int64_t* state_off;  /* [B+1]                                                                */

  /* --- sampler half: NextStateSampler tables in creation order, duplicates kept ---------------- */
  const int64_t* sp_ptr;     /* [R+1] entry offsets, R = state_off[B]*A                              */
  const int32_t* sp_next;    /* [E]   successor state index, local to the instance                   */
  const double*  sp_cum;     /* [E]   itertools.accumulate(probs) within the row                     */
  const double*  sp_reward;  /* [E]   reward of (s,a,s'): the deterministic value (loc)              */
  const uint8_t* sp_rkind;   /* [E]   0 deterministic, 1 Beta(sp_rp0, sp_rp1); NULL = all 0.  Beta entries
                                      need CMDP_FLAG_REWARD_MEANS (report the mean) or CMDP_RNG_PHILOX with
                                      sp_rp0/sp_rp1 (sampled on the device, Philox domain 3)            */
  const double*  sp_rp0;     /* [E]   Beta a (ignored for deterministic entries); may be NULL           */
  const double*  sp_rp1;     /* [E]   Beta b                                                            */
  const int32_t* sp_seed;    /* [R]   seed given to the row's NextStateSampler (MT_COMPAT)           */
  const int64_t* start_off;  /* [B+1]                                                                */
  const int32_t* start_state;/* [NS]  starting states, local index                                   */
  const double*  start_cum;  /* [NS]  accumulated starting probabilities                             */
  const int32_t* start_seed; /* [B]   seed of the start sampler (MT_COMPAT; ignored if one start)    */
  const uint64_t* philox_key;/* [B]   per-instance Philox key (PHILOX)                               */

  /* --- DP half: the non-zeros of the reference's dense float32 T[S,A,S], ascending column ------- */
  const int64_t* csr_ptr;    /* [R+1]                                                                */
  const int32_t* csr_col;    /* [N]   local successor index                                          */
  const float*   csr_val;    /* [N]                                                                  */
  const float*   R;          /* [R]   reward matrix R[s,a]                                           */
} cmdp_desc;

/* ---- library / device -------------------------------------------------------------------------- */
int cmdp_version(void);
/* SHA-256 (hex) over the sources this binary was compiled from (csrc files by name order, then this header), stamped by
   build() through -DCMDP_BUILD_ID; colosseum_amd/_lib.py refuses a library whose id differs from the tree's hash, so a
   stale prebuilt .so cannot pass for the sources next to it.  No reference counterpart. */
const char* cmdp_build_id(void);
const char* cmdp_last_error(void);
int cmdp_device_count(void);
int cmdp_set_device(int device);

/* ---- life cycle ------------------------------------------------------------------------------------ */
/* BaseMDP.__init__/instantiate_MDP (colosseum/mdp/base.py:327-503) for B instances at once. */
int cmdp_create(cmdp_t** out, const cmdp_desc* desc);
int cmdp_destroy(cmdp_t* h);
/* Handle's HIP stream (a hipStream_t) so that a caller can time it with HIP events. */
void* cmdp_stream(cmdp_t* h);

/* CMDP_FLAG_REWARD_CACHE: hands over `BaseMDP._rng` (colosseum/mdp/base.py:408) of every instance as
   numpy.random.RandomState.get_state() leaves it after the MDP's construction: mt_key [B][624], mt_pos [B] (0..624),
   has_gauss [B] and cached_gaussian [B] (either may be NULL = 0).  The library continues these streams whenever it
   fills a reward cache.  May be called again to reposition the streams (e.g. a new run on a fresh copy). */
int cmdp_set_reward_streams(cmdp_t* h, const uint32_t* mt_key, const int32_t* mt_pos, const int32_t* has_gauss,
                            const double* cached_gaussian);
/* `RandomState.beta(a, b, n)` on a caller-held stream state (updated in place): numpy's LEGACY sampler
   (numpy/random/src/legacy/legacy-distributions.c: Johnk for a, b <= 1, else Ga / (Ga + Gb) with Marsaglia-Tsang /
   Ahrens-Dieter gammas on the polar Gaussian) -- what scipy.stats.beta(a, b).rvs(n, random_state=rs) draws in
   BaseMDP.sample_reward.  Host only (libm); exists so that the CPU test suite can hold the sampler the reward caches are
   filled with against numpy itself, draw for draw. */
int cmdp_legacy_beta(uint32_t* mt_key, int32_t* mt_pos, int32_t* has_gauss, double* cached_gaussian, double a, double b,
                     int64_t n, double* out);

/* ---- interaction: BaseMDP.reset / BaseMDP.step -------------------------------------------------- */
/* BaseMDP.reset (colosseum/mdp/base.py:1268-1277) on every instance with mask[b] != 0 (all when mask
   is NULL): h = 0, start state sampled, its state-visit count bumped.  obs_out[b] = start state index
   (untouched where masked out); may be NULL. */
int cmdp_reset(cmdp_t* h, const uint8_t* mask, int32_t* obs_out);

/* BaseMDP.step(action, auto_reset) (colosseum/mdp/base.py:1279-1317) on every instance.
   obs[b] = new state index, or -1 when the episode ended (h >= H); step_type: 0 FIRST (an auto-reset
   happened instead of a transition; reward 0), 1 MID, 2 LAST.  With auto_reset == 0 an instance that
   needs a reset makes the call fail with CMDP_ERR_NEEDS_RESET and nothing is modified. */
int cmdp_step(cmdp_t* h, const int32_t* actions, int auto_reset,
              int32_t* obs, double* reward, uint8_t* step_type);

/* The body of MDPLoop.run's loop, env side, fused on the device
   (colosseum/experiment/agent_mdp_interaction.py:238-298): n_steps transitions per instance, each
   episodic termination followed at once by reset() (which is not counted as a step).  Instances must
   have been reset.  last_obs[B], reward_sum[B] (sum of the n_steps rewards) and the three trace
   arrays ([n_steps][B], obs = -1 on termination) may each be NULL. */
int cmdp_rollout(cmdp_t* h, int policy, const void* policy_arg, int64_t n_steps,
                 int32_t* last_obs, double* reward_sum,
                 int32_t* trace_obs, double* trace_reward, uint8_t* trace_type);
/* Same launch without the final synchronisation or any copy-back (for back-to-back timing). */
int cmdp_rollout_async(cmdp_t* h, int policy, int64_t n_steps);
int cmdp_synchronize(cmdp_t* h);

/* Measurements taken inside the library (for bench.py's roofline objects; no reference counterpart).
   CMDP_STAT_DP_KERNEL_MS: HIP-event time, on the handle's stream, of the sweep kernel of the last
   cmdp_vi_discounted / cmdp_pe_discounted (kernel only: no upload, no result copy).
   CMDP_STAT_DP_KERNEL: which kernel that was -- 1 K2 (workgroup, CSR in LDS/HBM), 2 K2R, 5 K2U, 7 K2W, 6 K3 (Gauss-Seidel). */
enum { CMDP_STAT_DP_KERNEL_MS = 1, CMDP_STAT_DP_KERNEL = 2,
       CMDP_STAT_REWARD_FILLS = 3,   /* CMDP_FLAG_REWARD_CACHE: blocks of 5000 samples drawn so far                */
       CMDP_STAT_REWARD_ROUNDS = 4,  /* ... and park / fill / relaunch rounds                                        */
       CMDP_STAT_ROLLOUT_KERNEL_MS = 5, /* K1U: HIP-event time of k_rollout_tmpl_stream in the last launch (last segment) */
       CMDP_STAT_HIST_KERNEL_MS = 6,    /* K1U: ... and of its k_trace_hist (on the second stream when overlapped)          */
       CMDP_STAT_CHAIN_FAST_INSTANCES = 7 /* instances (evaluated or masked out) the last average-reward call did NOT hand to K9:
                                             those K9F solved (irreducible chain, fill-reducing elimination order)          */,
       CMDP_STAT_REWARD_FILL_MS = 8,   /* CMDP_FLAG_REWARD_CACHE: host wall time spent drawing blocks (all host threads together
                                          count once: the time the calling thread waited for the draws)                       */
       CMDP_STAT_REWARD_ROUND_MS = 9   /* ... and wall time of the park rounds as a whole: copy of the park list, draws, install,
                                          up to the relaunch                                                                  */,
       CMDP_STAT_DIAMETER_CLUSTER_LAUNCHES = 10,  /* diameter solves of this handle that ran on K5C (clusters of workgroups per
                                                     group of 64 targets, k_diam_cluster)                                     */
       CMDP_STAT_DIAMETER_CLUSTER_FALLBACKS = 11  /* ... and K5C launches given up because a cluster's workgroups were not all
                                                     resident within the time limit (the targets were then solved by K5S)      */ };
int cmdp_stat(cmdp_t* h, int which, double* out);
/* Latency floor of the LDS-resident rollout kernels, measured on the current device: one wavefront per CU follows
   per-lane uint16 tables in LDS for n_steps dependent reads.  CMDP_CALIB_LDS_READ: the bare dependent ds_read_u16
   (read, mask, address); CMDP_CALIB_LDS_CHAIN: the minimal dependency chain of one deterministic transition (action
   bit, successor read, mask, in-episode step, episode-end select); CMDP_CALIB_LDS_CHAIN_SHARED: the same for the
   shared-table kernel K1T (the state's word pair and its swap bit: two independent reads, the word selected by a bit-field
   extract).  ns_per_step = launch time / n_steps. */
enum { CMDP_CALIB_LDS_READ = 0, CMDP_CALIB_LDS_CHAIN = 1, CMDP_CALIB_LDS_CHAIN_SHARED = 2 };
int cmdp_calibrate(int what, int64_t n_steps, double* ns_per_step);

/* Tuning knobs (never change results).  CMDP_OPT_ROLLOUT_KERNEL: 0 = automatic, 1 = lane-per-instance
   kernel with the tables in HBM, 2 = LDS-resident kernel (fails with CMDP_ERR_UNSUPPORTED when the batch is
   not eligible: deterministic dynamics, one start state, equal state counts <= 65535, <= 256 distinct reward
   values, at least 8 instances per 160 KiB of LDS), 3 = LDS-resident kernel for STOCHASTIC dynamics K1S (Philox mode;
   the sampler tables compressed into shared cumulative-probability patterns and per-state successor sets: <= 16
   entries per row, <= 16 distinct successors per state, <= 64 patterns, rewards a function of the successor or of
   the row, at least 4 instances per 160 KiB of LDS), 4 = the shared-table pipeline K1T (a batch eligible for 2 with two
   actions whose instances are, state by state, the first instance's rows or their swap -- the seeds of a family whose
   structure does not depend on the seed: the workgroup keeps one successor table and a swap bit per state and
   instance, 128 instances per CU at config C2 instead of 52; taken automatically when eligible, 2 keeps K1L / K1P),
   5 = K1U, K1T's chain with the 16-bit trace streamed to HBM and histogrammed by a second kernel instead of 8-bit count
   deltas in LDS: up to 256 instances per CU resident at once (taken automatically when that saves a round of workgroups
   over K1T -- config C2: one round instead of two; 4 keeps K1T),
   6 = K1E, the EPISODE-PARALLEL rollout (a batch eligible for 2 that is episodic, has two actions, at most four distinct
   reward values and at most 512 states per instance): lane = (instance, episode) -- under the random policy the episodes
   of an instance are independent (fixed horizon, one start state: colosseum/mdp/base.py:1268-1277,1310-1317; the action
   of transition n is a function of (key, n)), so a launch is walked as H-step chains, 4 096 of them per CU, against
   private {successor | count} tables in LDS; the float64 reward sums are added in transition order by a second kernel
   from 2-bit reward codes.  Taken automatically when eligible (launches of >= 64 transitions); 5 / 4 / 2 keep the
   chain kernels.
   For 3: the automatic choice takes it while the batch is small enough that
   the HBM-table kernel's rate, which grows with the batch, stays below it -- FrozenLake 20x20: up to ~35 000 instances).
   CMDP_OPT_DP_KERNEL (Jacobi sweeps): 0 = automatic, 1 = workgroup kernel with the CSR in LDS or HBM,
   2 = register-resident CSR kernel (CMDP_ERR_UNSUPPORTED when no compiled shape fits: A in 2..4, <= 8
   non-zeros per row, <= 1024 states), 3 = (cmdp_diameter only) the 64-targets-per-workgroup kernel K5S that is
   otherwise taken when the value vector of an instance does not fit LDS (4 = the same with its generic CSR walker
   instead of the fixed-width-row variant; 6 = its tiled form K5T, which gathers the value rows of a cluster of states
   into LDS first: half the HBM traffic, same bits, not faster -- on request only), 5 = the distinct-successor form K2U of the register-resident kernel (taken
   by default when the A rows of a state share their successors: <= 8 distinct columns per state, rows in ascending
   column order; 2 keeps the per-row form K2R), 7 = K2W, K2U's tables with ONE wavefront per instance and no barrier per
   sweep (taken by default for 257..448 states, <= 5 distinct successors per state, <= 4 non-zeros per row; 5 keeps K2U).
   All forms return identical bits.
   CMDP_OPT_CHAIN_EXACT_ORDER (cmdp_average_reward / cmdp_qlearning_average_reward): 1 = every float64 sum of the GTH
   elimination in the reference's index order and the states eliminated in the reference's order (bit-equal to cmdp_gth
   and the oracle, one serial chain per sum);
   0 (default) = wave butterfly sums and -- for irreducible chains -- a fill-reducing elimination order computed once per
   instance from the MDP's transition graph, several independent pivots per step (kernel K9F): deterministic, within
   ~1e-13 relative of the former (GTH is subtraction-free: any elimination order gives the stationary distribution to
   rounding), 5-8 x faster on chains with hundreds of states.
   CMDP_OPT_DIAMETER_WORKSPACE_MB: HBM the value arrays of K5S may take per launch (default 24576; 512 bytes per
   state per group of 64 targets; more groups in flight = more of the GPU busy).
   CMDP_OPT_DIAMETER_RELABEL_MIN_STATES: K5S stores the rows of instances with at least this many states (default 8192)
   in a locality order of the states (breadth-first clusters of the transition graph) so that the value rows a chunk of
   states gathers were fetched by the chunks before it; row contents and entry order are unchanged, results bit-equal.
   A value above every instance's size keeps the caller's state order.
   CMDP_OPT_MIXING_PATH (cmdp_mixing_time): 0 = automatic (matrix powers when an instance has more than 1024 states or
   a float64 row does not fit LDS), 1 = matrix powers, 2 = one sparse step at a time with the row in LDS.
   CMDP_OPT_LDS_GROUPS_PER_CU: 1 or 2 workgroups of the fused-walker kernel K1L per CU (default: whichever needs
   fewer rounds of workgroups for the batch).  Setting it re-plans the handle onto K1L with the
   balanced number of instances per workgroup for that many groups (the pipeline kernel K1P always runs one workgroup
   per CU); CMDP_ERR_INVALID when the batch is not eligible for the LDS-resident kernels or two groups do not fit. */
enum { CMDP_OPT_ROLLOUT_KERNEL = 1, CMDP_OPT_DP_KERNEL = 2, CMDP_OPT_LDS_GROUPS_PER_CU = 3,
       CMDP_OPT_DIAMETER_WORKSPACE_MB = 4, CMDP_OPT_CHAIN_EXACT_ORDER = 5, CMDP_OPT_MIXING_PATH = 6,
       CMDP_OPT_DIAMETER_RELABEL_MIN_STATES = 7 };
int cmdp_set_option(cmdp_t* h, int option, int64_t value);

/* What the LDS-resident random-policy rollout of this handle is (introspection for benchmarks and tests; no
   reference counterpart): plan[0] = 1 if the batch is eligible for it, plan[1] = 1 for the wavefront-pipeline kernel
   K1P (k_rollout_pipe), 0 for the fused walker K1L (k_rollout_lds), 2 for the stochastic-dynamics kernel K1S
   (k_rollout_stoch), 3 for the shared-table pipeline K1T (k_rollout_tmpl), 4 for its streamed-trace form K1U
   (k_rollout_tmpl_stream + k_trace_hist), plan[2] = instances per workgroup,
   plan[3] = transitions per chunk.  The kernel flavour and chunk length are chosen when the handle is created (fewest
   rounds of workgroups x measured time per transition). */
int cmdp_lds_plan(cmdp_t* h, int32_t plan[4]);

/* BaseMDP.get_visitation_counts / reset_visitation_counts (colosseum/mdp/base.py:1357-1382).
   state_counts [state_off[B]], sa_counts [state_off[B]*A]; either may be NULL. */
int cmdp_visits(cmdp_t* h, int64_t* state_counts, int64_t* sa_counts);
int cmdp_reset_visits(cmdp_t* h);
/* Restores the counters (e.g. of an earlier run that is being continued: the reference's MDPs are pickled with their
   counts).  Either array may be NULL (left as it is); every value must fit the device's int32 counters.  The device
   counters are int32: the library keeps an upper bound of what any of them can hold (largest restored value + two per
   transition taken since -- an arrival and, at an episode end, the reset) and refuses a call that could carry one past
   2^31 - 1 with CMDP_ERR_OVERFLOW, before anything is stepped. */
int cmdp_set_visits(cmdp_t* h, const int64_t* state_counts, const int64_t* sa_counts);
/* Current state index, in-episode step and needs-reset flag of every instance (BaseMDP.cur_node,
   .h, .necessary_reset); each may be NULL. */
int cmdp_state(cmdp_t* h, int32_t* cur, int32_t* hstep, uint8_t* needs_reset);
/* BaseMDP.last_starting_node as a state index: the state the latest reset() of every instance sampled [B], and
   (nullable) the one the reset before it sampled -- MDPLoop evaluates a log row BEFORE the reset that follows a
   terminating step, while the fused kernels reset at once. */
int cmdp_last_start(cmdp_t* h, int32_t* last_start, int32_t* previous_start);

/* ---- dynamic programming ---------------------------------------------------------------------------- */
/* discounted_value_iteration (colosseum/dynamic_programming/infinite_horizon.py:14-44,121-164).
   R_override ([R] or NULL) replaces the reward matrix (e.g. -R for the worst policy).  Q [R], V
   [state_off[B]], sweeps [B] (sweeps executed, including the converging one); sweeps may be NULL.
   max_abs_value <= 0 disables the bound. */
int cmdp_vi_discounted(cmdp_t* h, float gamma, double epsilon, int scheme, int64_t max_sweeps,
                       double max_abs_value, const float* R_override,
                       float* Q, float* V, int64_t* sweeps);
/* discounted_policy_evaluation (infinite_horizon.py:47-64,167-205); pi [R] float32. */
int cmdp_pe_discounted(cmdp_t* h, const float* pi, float gamma, double epsilon, int scheme,
                       int64_t max_sweeps, const float* R_override,
                       float* Q, float* V, int64_t* sweeps);
/* episodic_value_iteration / episodic_policy_evaluation
   (colosseum/dynamic_programming/finite_horizon.py:11-42).  Q of instance b is [H+1][S_b][A] at
   offset (H+1)*state_off[b]*A, V is [H+1][S_b] at (H+1)*state_off[b]; pi is [H][S_b][A] at
   H*state_off[b]*A. */
int cmdp_vi_episodic(cmdp_t* h, int H, const float* R_override, float* Q, float* V);
int cmdp_pe_episodic(cmdp_t* h, int H, const float* pi, const float* R_override, float* Q, float* V);

/* ---- hardness measures --------------------------------------------------------------------------- */
/* get_diameter, continuous setting (colosseum/hardness/measures/diameter.py:76-106): for every
   target state es the optimal expected hitting time by value iteration with es absorbing, R = -1,
   gamma = 1; diameter[b] = max over targets and start states.  per_target [state_off[B]] may be
   NULL.  scheme as in cmdp_vi_discounted (AUTO = the rule applied to the instance's T). */
int cmdp_diameter(cmdp_t* h, double epsilon, int scheme, int64_t max_sweeps,
                  float* per_target, float* diameter);
/* `_get_sparse_diameter` (colosseum/hardness/measures/diameter.py:382-420): the variant a single-core reference runs
   for continuous MDPs above 1000 states (dispatch :35-39) -- float64 hitting times, targets in index order, and the
   running-maximum early exit `diff < 0.05 and max - 1 < diameter so far`, which makes the value depend on that order.
   The device solves every target to diff < epsilon and logs (diff, max) per sweep; the host replays the reference's
   sequential loop on the logs, so the committed values are the reference's.  diameter [B] float64; running_max
   [state_off[B]] (may be NULL) = the running maximum after each target.  Continuous handles only. */
int cmdp_diameter_sparse_f64(cmdp_t* h, double epsilon, int64_t max_sweeps, double* running_max, double* diameter);
/* The per-target hitting-time solves of cmdp_diameter for the targets [target_lo, target_hi) of the flat state
   space [0, state_off[B]) only (Jacobi scheme, kernel K5S; any instance size): per_target[i] belongs to target
   target_lo + i.  This is how config C5 (one MDP with ~50 000 states) is split over GPUs: every rank takes a
   contiguous range of targets and the diameter is the maximum over all ranks
   (colosseum/hardness/measures/diameter.py:108-124 does the same with a process pool). */
int cmdp_diameter_range(cmdp_t* h, double epsilon, int64_t max_sweeps, int64_t target_lo, int64_t target_hi,
                        float* per_target);
/* get_diameter, episodic setting (colosseum/hardness/measures/diameter.py:193-234,285-318) on the
   time-augmented transition array of get_episodic_transition_matrix_and_rewards
   (colosseum/mdp/utils/mdp_creation.py:98-128), which is never materialised: H, the starting states
   (start_off [B+1], start_state [NS], start_prob [NS] as float32) and the CSR define it.  Every target
   runs to diff < epsilon; the reference's running-maximum early exit (order dependent) is not applied.
   per_target [state_off[B]] may be NULL. */
int cmdp_diameter_episodic(cmdp_t* h, int H, const int64_t* start_off, const int32_t* start_state,
                           const float* start_prob, double epsilon, int64_t max_sweeps,
                           float* per_target, float* diameter);
/* calculate_norm_discounted (colosseum/hardness/measures/value_norm.py:83-87). V [state_off[B]]. */
int cmdp_value_norm(cmdp_t* h, const float* V, float* out);

/* ---- agents on the device --------------------------------------------------------- */
/* Batched tabular Q-learning with UCB exploration for episodic MDPs: one agent per environment instance of `env`,
   reproducing colosseum/agent/agents/episodic/q_learning.py (QLearningEpisodic: QValuesModel.step_update + the
   greedy QValuesActor with its RandomState(seed) tie-break) bit for bit.  seeds [B]; ucb_type 0 = hoeffding,
   1 = bernstein (c_2 required). */
typedef struct cmdp_agent cmdp_agent_t;
int cmdp_qlearning_create(cmdp_agent_t** out, cmdp_t* env, const int32_t* seeds, int64_t optimization_horizon,
                          double p, double c_1, double c_2, double min_at, int ucb_type);
int cmdp_qlearning_destroy(cmdp_agent_t* a);
/* The continuous-setting agent, colosseum/agent/agents/infinite_horizon/q_learning.py (QLearningContinuous: optimistic
   Q-learning for the average-reward setting) on a continuous environment handle; get_span_approx / get_H are the
   reference defaults.  The returned handle is used with the same cmdp_qlearning_run / _tables / _destroy. */
int cmdp_qlearning_continuous_create(cmdp_agent_t** out, cmdp_t* env, const int32_t* seeds,
                                     int64_t optimization_horizon, double min_at, double confidence,
                                     double span_approx_weight, double h_weight);
/* MDPLoop.run's loop with the agent in it (colosseum/experiment/agent_mdp_interaction.py:238-298): per step
   select_action -> BaseMDP.step -> step_update -> reset() at the end of an episode.  train_mask [B] (NULL = all
   ones): instances with 0 act but do not update, as after MDPLoop froze their training (:284-288).
   cumulative_reward [B] receives `MDPLoop._cumulative_reward`, the running float64 sum of ALL rewards since the agent
   was created (the additions continue across calls in transition order).  actions_trace [n_steps][B] may be NULL. */
int cmdp_qlearning_run(cmdp_agent_t* a, int64_t n_steps, const uint8_t* train_mask, int8_t* actions_trace,
                       double* cumulative_reward);
/* MDPLoop.run as ONE call (colosseum/experiment/agent_mdp_interaction.py:179-302 with its indicator code :304-578 and
   colosseum/experiment/indicators.py:29-45): the interaction of cmdp_qlearning_run, and at every logging step the
   evaluation of the agents' greedy policies on the device (episodic: V[0] by policy evaluation; continuous: average reward
   of the policy's chain from the current state) followed by the reference's 18 performance indicators, computed on the
   host in C++ with the reference's numpy scalar types (csrc/cmdp_tracker.h), the `_is_policy_optimal` freeze of training
   and the wall-clock limit -- one stream synchronisation per logging step, no Python in between.

   desc: n_steps = T, log_every (<= 0: only the final row), n_check = n_log_intervals_to_check_for_agent_optimality,
   max_time (seconds of training for the batch; the reference gives every instance its own process and clock: here all
   instances still training are frozen at the first logging step after max_time - 0.5 s); base_val / base_kind [B][3]: the
   optimal, worst and random average reward of every instance as Python / numpy scalars (kind 0 = Python float, 1 =
   np.float32, 2 = np.float64); episodic
   handles also need horizon, opt0 / worst0 (V*[0] and V_worst[0], flat [state_off[B]]) and the start distribution as
   start_pos / start_prob [B][kmax] (flat state positions in state-index order, probability 0 for padding).
   Outputs for the n_logs = len(range(log_every, T, log_every)) + 1 rows: steps [n_logs]; values / kinds
   [n_logs][CMDP_LOG_COLUMNS][B] -- the columns are the indicator names in sorted order without "steps" (kind 1 =
   np.float32, 2 = np.float64; the values are rounded to 5 decimals in their type, as the logger receives them);
   last_training_step [B] (-1: the time limit was not hit); is_training [B] after the run (these two may be NULL). */
#define CMDP_LOG_COLUMNS 17
typedef struct cmdp_loop_desc {
  int64_t n_steps, log_every;
  int32_t n_check, horizon, kmax, reserved;
  double max_time;
  const double* base_val;
  const int32_t* base_kind;
  const float* opt0;
  const float* worst0;
  const int64_t* start_pos;
  const double* start_prob;
} cmdp_loop_desc;
int cmdp_qlearning_run_logged(cmdp_agent_t* a, const cmdp_loop_desc* desc, int64_t n_logs, int64_t* steps, double* values,
                              uint8_t* kinds, int64_t* last_training_step, uint8_t* is_training);
/* The indicator code of cmdp_qlearning_run_logged alone, on caller-supplied inputs (host only, no device): per row the
   step, whether it lies inside the loop (the final row does not run the freeze check), the steps since the previous row,
   cumulative rewards [n_logs][B], and -- episodic: V0 [n_logs][state_off[B]] of the policies and the logged episode's
   start state [n_logs][B] (instance-relative); continuous: avg / avg_kind [n_logs][B] (kind != 0: np.float32).
   is_training [n_logs][B] receives the flags after every row.  For tests against the reference's indicator code. */
int cmdp_tracker_replay(const cmdp_loop_desc* desc, int32_t B, int32_t episodic, const int64_t* state_off, int64_t n_logs,
                        const int64_t* log_steps, const uint8_t* in_loop, const int64_t* n_since, const double* cum_reward,
                        const float* V0, const int64_t* start_state, const double* avg, const int32_t* avg_kind,
                        double* values, uint8_t* kinds, uint8_t* is_training);
/* V[0, :] of episodic_policy_evaluation for the agents' current greedy policies
   (BaseAgent.current_optimal_stochastic_policy = argmax_3d(Q), ties by RandomState(42)): what
   MDPLoop._compute_episodic_regret needs.  The environment handle must carry the DP half.  V0 [state_off[B]]. */
int cmdp_qlearning_evaluate(cmdp_agent_t* a, float* V0);
/* argmax_2d (colosseum/dynamic_programming/utils.py:12-25) of the continuous agents' Q tables: one-hot float32
   policies pi [state_off[B]*A] (BaseAgent.current_optimal_stochastic_policy). */
int cmdp_qlearning_policy(cmdp_agent_t* a, float* pi);
/* BaseMDP-side `get_average_reward(T, R, policy, [(state, 1.0)])` (colosseum/mdp/utils/markov_chain.py:12-31, the
   regret of every continuous-setting log row, agent_mdp_interaction.py:518-532) for the agents' current greedy
   policies from the environments' current states, all on the device: recurrent classes in networkx's
   attracting_components order, GTH elimination of the class taken, numpy's summation order.  avg[b] is the value;
   kind[b] = 1 when the reference's result is a numpy float32 (one recurrent class smaller than the chain), else 0
   (float64).  mask [B] selects the instances to evaluate (NULL = all; the others' outputs are left untouched). */
int cmdp_qlearning_average_reward(cmdp_agent_t* a, const uint8_t* mask, double* avg, int32_t* kind);
/* argmax_3d (colosseum/dynamic_programming/utils.py:28-39) of host tables Q [per instance q_layers*S_b*A, q_layers
   >= H]: one-hot float32 policy of the first H layers, pi [per instance H*S_b*A]. */
int cmdp_greedy_policy_episodic(cmdp_t* h, int H, int q_layers, const float* Q, float* pi);
/* Q [B instances concatenated: H*S_b*A float32 each for the episodic agent; S_b*A FLOAT64 each -- pass a double* --
   for the continuous agent, whose tables are float64], N likewise (int32); either may be NULL. */
int cmdp_qlearning_tables(cmdp_agent_t* a, float* Q, int32_t* N);

/* ---- non-tabular observations --------------------------------------------------------------------------------- */
/* EmissionMap.all_observations (colosseum/emission_maps/base.py:56-83) of every instance: float32 feature vectors of
   length F, per instance [S_b][F] at state_off[b]*F, or -- time_indexed, episodic handles -- [H][S_b][F] at
   H*state_off[b]*F.  The table is copied to the device. */
int cmdp_set_observation_table(cmdp_t* h, const float* table, int32_t F, int time_indexed);
/* EmissionMap.get_observation (emission_maps/base.py:110-141) for the current state of every instance: obs [B][F];
   all zeros for an episodic instance that has reached its horizon.  noise_scale > 0 adds scale * N(0,1) per element
   from the instance's Philox stream (throughput mode, CMDP_RNG_PHILOX only; the reference-exact GaussianUncorrelated
   stream is numpy's and stays on the host, colosseum_amd/emission_maps.py). */
int cmdp_observe(cmdp_t* h, double noise_scale, float* obs);
/* The same with any of the reference's four noise classes (the files of colosseum/noises/) in throughput mode: Gaussian
   (scale), Gaussian correlated (y = L z, chol = the lower Cholesky factor L [F][F] of the covariance -- the reference
   draws it from a Wishart distribution once per MDP), Student-t (df) and the multivariate Student-t (chol = factor of
   the shape matrix, df).  Philox mode only; distribution-exact, not stream-exact (the reference-exact streams are
   numpy's / scipy's and stay on the host, colosseum_amd/emission_maps.py: CompatNoise). */
enum { CMDP_NOISE_NONE = 0, CMDP_NOISE_GAUSSIAN = 1, CMDP_NOISE_GAUSSIAN_CORRELATED = 2, CMDP_NOISE_STUDENT_T = 3,
       CMDP_NOISE_STUDENT_T_CORRELATED = 4 };
int cmdp_observe_noise(cmdp_t* h, int kind, double scale, double df, const float* chol, float* obs);

/* ---- Markov chains ------------------------------------------------------------------------------------ */
/* BUILD-DEFINED (the reference has no mixing time): t_mix[b] = smallest t >= 1 with
   max_s TV(P^t(s, .), stationary) <= threshold for the chain P[s, j] = sum_a pi[s, a] T[s, a, j] of instance b, rows
   normalised to sum to one in float64 (float32 probabilities sum to 1 only within ~1e-7, which would leak 1e-2 of mass
   over the 1e5 steps a slow chain needs); pi [state_off[B]*A] float32, NULL = the uniform policy; stationary
   [state_off[B]] float64, e.g. from cmdp_gth on the same normalised chain;
   float64 on the device; -1 when max_steps is reached first (periodic chains never get there).  tv_at [B] (may be
   NULL) receives the total variation at t_mix (or at the last t evaluated).  Parity unpinned by construction.
   Two paths (CMDP_OPT_MIXING_PATH).  Stepping: the S rows of X_t = P^t advance one sparse step at a time, row in LDS
   (S <= ~20 000).  Matrix powers, for large chains (config C5, S = 50 272): X is a dense S x S float64 matrix in HBM
   (20 GB at C5); since max_s TV is non-increasing in t, A_k = P^(2^k) is squared (rocBLAS dgemm) until 2^K is mixed,
   a binary search multiplies the stored powers back in, one dgemm per bit, and the last 2^r steps -- r = the lowest
   power still held when HBM ran out of S x S buffers -- are sparse steps that gather from the row in global memory.
   The two paths round differently (1e-16 relative per product) and return the same t unless the total variation
   crosses the threshold within that distance. */
int cmdp_mixing_time(cmdp_t* h, const float* pi, const double* stationary, double threshold, int64_t max_steps,
                     int64_t* t_mix, double* tv_at);
/* get_average_reward (colosseum/mdp/utils/markov_chain.py:12-31) of deterministic stationary policies: actions
   [state_off[B]] gives the action of every state of every (continuous, horizon 0) instance, start_states [B] the
   instance-relative state the chain starts from.  Outputs as cmdp_qlearning_average_reward; n_classes [B] (may be
   NULL) receives the number of recurrent classes of each chain.  CMDP_ERR_UNSUPPORTED when an instance exceeds the
   kernel's LDS budget (80 bytes per state + 4 per successor). */
int cmdp_average_reward(cmdp_t* h, const int32_t* actions, const int32_t* start_states, const uint8_t* mask, double* avg,
                        int32_t* kind, int32_t* n_classes);
/* _gth_solve_numba (colosseum/mdp/utils/markov_chain.py:139-166): stationary distributions of `count` chains with a
   single recurrent class each, float64 GTH elimination on the current device.  Chain m is the dims[m] x dims[m]
   row-major matrix at mats + sum_{i<m} dims[i]^2 (not modified); its distribution goes to
   out + sum_{i<m} dims[i]. */
int cmdp_gth(int count, const int32_t* dims, const double* mats, double* out);

#ifdef __cplusplus
}
#endif
#endif /* CMDP_H */
