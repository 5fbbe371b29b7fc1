// cmdp.hip -- C ABI (include/cmdp.h) over the HIP kernels of cmdp_kernels.h.  gfx950 only.
#include "../../include/cmdp.h"

#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <limits>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

extern char** environ;

#include "cmdp_kernels.h"
#include "cmdp_tracker.h"
#include "cmdp_k1s.h"
#include "cmdp_k1t.h"
#include "cmdp_k1u.h"
#include "cmdp_k1e.h"
#include "cmdp_agent.h"
#include "cmdp_chain.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(CMDP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

constexpr int kLdsBudget = 160 * 1024;  // bytes of LDS one workgroup may claim on gfx950
constexpr int kDpBlock = 256;

// Device buffer that only grows: hipFree (and often hipMalloc) synchronises the WHOLE device, which would serialise
// handles that run concurrently on their own streams (benchmark runner: one host thread per device batch), so steady-state
// calls must not allocate.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;    // logical size of the current contents
  size_t cap = 0;  // allocated elements
  hipError_t alloc(size_t count) {
    if (count <= cap) {
      n = count;
      return hipSuccess;
    }
    release();
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) { n = count; cap = count; }
    return e;
  }
  hipError_t upload(const T* src, size_t count, hipStream_t s) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
  }
  hipError_t zero(hipStream_t s) { return n ? hipMemsetAsync(p, 0, n * sizeof(T), s) : hipSuccess; }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  ~DevBuf() { release(); }
};

// Page-locked host memory for the per-log read-backs of the logged loop.
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  int alloc(size_t n) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(T) * std::max<size_t>(n, 1), 0));
    return CMDP_OK;
  }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

}  // namespace

struct cmdp {
  int device = 0;
  hipStream_t stream = nullptr;
  int B = 0, A = 0, H = 0, rng_mode = 0, layout = 0;
  double rmin = 0, rmax = 1;
  int64_t n_states = 0, n_rows = 0, n_entries = 0, n_csr = 0, n_slots = 0;
  bool has_env = false, has_dp = false;
  bool sample_beta = false;  // Beta rewards drawn on the device (CMDP_RNG_PHILOX without CMDP_FLAG_REWARD_MEANS)
  bool beta_gammas = false;  // CMDP_FLAG_BETA_GAMMAS
  // CMDP_FLAG_REWARD_CACHE: the reference's per-triple caches of 5000 samples from the MDP's own numpy stream
  // (cmdp_reward_cache.h): blocks in HBM, drawn on the host whenever an instance parks
  bool reward_cache = false, rc_streams_set = false;
  std::vector<uint8_t> h_rkind;
  std::vector<double> h_rp0, h_rp1;
  std::vector<int32_t> h_canon;
  std::vector<cmdp_rc::NumpyStream> rc_streams;   // [B] BaseMDP._rng of every instance, where construction left it
  std::vector<double*> rc_blk_h;                  // [E] host mirror of the block pointers
  std::vector<double*> rc_chunks;                 // pool of blocks: chunks of rc_chunk_blocks blocks each
  size_t rc_chunk_blocks = 0, rc_next_block = 0;  // blocks handed out so far
  int rc_cap = 0;                                 // blocks one install pass can stage
  int64_t rc_fills = 0, rc_rounds = 0;            // CMDP_STAT_REWARD_FILLS / _ROUNDS
  double rc_fill_ms = 0.0, rc_round_ms = 0.0;     // CMDP_STAT_REWARD_FILL_MS / _ROUND_MS
  double* rc_stage_h = nullptr;                   // pinned [rc_cap][5000]
  double** rc_dst_h = nullptr;                    // pinned [rc_cap]
  int32_t* rc_ent_h = nullptr;                    // pinned [rc_cap]
  int32_t* rc_list_h = nullptr;                   // pinned [B]
  int32_t* rc_pend_h = nullptr;                   // pinned [B]
  DevBuf<int32_t> d_rc_canon, d_rc_pos, d_rc_pend_e, d_rc_pend_prev, d_rc_pend_act, d_rc_park_count, d_rc_park_list, d_rc_ent;
  DevBuf<double*> d_rc_blk, d_rc_dst;
  DevBuf<long long> d_rc_left;
  DevBuf<double> d_rc_stage;
  RewardCache rcache() {
    RewardCache c{};
    c.canon = d_rc_canon.p; c.blk = d_rc_blk.p; c.pos = d_rc_pos.p; c.pend_e = d_rc_pend_e.p; c.pend_prev = d_rc_pend_prev.p;
    c.pend_act = d_rc_pend_act.p; c.park_count = d_rc_park_count.p; c.park_list = d_rc_park_list.p; c.left = d_rc_left.p;
    return c;
  }
  std::vector<int64_t> state_off;  // host copy
  std::vector<int64_t> csr_nnz;    // per instance
  int max_S = 0;
  int64_t max_inst_nnz = 0;
  int max_row_nnz = 0;
  int max_state_unique = 0;  // distinct successor columns of a state over its A rows (0: not computed / rows unsorted)
  bool known_reset = false;  // every instance is known to be past reset() (cmdp_rollout_async checks once, cmdp_step clears)
  hipEvent_t ev_dp0 = nullptr, ev_dp1 = nullptr;  // around the sweep kernel of the last discounted solve (cmdp_stat)
  hipEvent_t ev_row[2] = {nullptr, nullptr};      // logged loop: policy + state snapshot taken | evaluation of the row complete
  DevBuf<int32_t> d_cur_snap;                     // logged loop: current states at the row (the solve runs beside the next interval)
  int last_dp_kernel = 0;     // CMDP_STAT_DP_KERNEL: 1 K2, 2 K2R, 5 K2U, 6 K3 (Gauss-Seidel)
  float *zc_Q = nullptr, *zc_V = nullptr;   // discounted(): device aliases of page-locked result arrays (run_sweeps)
  int64_t* zc_sweeps = nullptr;
  bool zc_used = false;
  PinnedBuf<int32_t> pin_status;   // ... and the per-instance status words of such a solve (read after the stream has drained)
  size_t pin_status_n = 0;
  int dp_kernel = 0;  // 0 auto, 1 LDS/global-CSR workgroup kernel, 2 register-resident kernel K2R, 5 its distinct-successor form K2U (3, 4: diameter only)

  DevBuf<int64_t> d_state_off, d_entry_base, d_start_off, d_csr_ptr;
  DevBuf<RowDesc> d_row;
  DevBuf<int32_t> d_sp_next, d_start_state, d_start_slot, d_mt_pos, d_cur, d_h, d_visits_s, d_visits_sa, d_csr_col,
      d_flag, d_status, d_i32_scratch, d_last_start, d_prev_start;
  DevBuf<double> d_sp_cum, d_sp_reward, d_start_cum, d_f64_scratch, d_sp_rp0, d_sp_rp1;
  DevBuf<uint8_t> d_sp_rkind;
  DevBuf<uint2> d_key;
  DevBuf<uint32_t> d_mt;
  DevBuf<uint8_t> d_need_reset, d_u8_scratch;
  DevBuf<unsigned long long> d_ntrans, d_nreset;
  DevBuf<float> d_csr_val, d_R, d_Rov, d_pi, d_Q, d_V, d_per_target, d_Ev, d_out;
  DevBuf<int64_t> d_sweeps;
  DevBuf<int8_t> d_actions8;
  DevBuf<int32_t> d_tr_obs, d_last_obs;
  DevBuf<double> d_tr_rew, d_rsum;
  DevBuf<uint8_t> d_tr_type, d_mask;
  // LDS-resident rollout of stochastic-dynamics batches (K1S)
  bool k1s_ok = false;
  K1sPlan k1s{};
  size_t k1s_bytes = 0;
  DevBuf<uint4> d_k1s_dict;
  DevBuf<uint8_t> d_k1s_pat, d_k1s_rc;
  DevBuf<uint16_t> d_k1s_sets, d_k1s_shape16;
  DevBuf<double> d_k1s_patterns, d_k1s_rvals;
  // LDS-resident rollout (K1L)
  bool lds_ok = false;
  int lds_G1 = 0, lds_G2 = 0;  // LDS capacity in instances per workgroup at one / two workgroups per CU
  int cus = 256;
  int rollout_kernel = 0;  // CMDP_OPT_ROLLOUT_KERNEL
  LdsPlan lds_plan{};
  size_t lds_bytes = 0;
  bool tmpl_auto = false;
  bool tmpl_ok = false;      // K1T: one shared successor table per workgroup + per-instance action-swap bits
  TmplPlan tmpl_plan{};
  size_t tmpl_lds = 0;
  DevBuf<uint16_t> d_tmpl_words;
  DevBuf<uint8_t> d_swap_bits;
  // K1U: K1T with the trace streamed to HBM and histogrammed by a second kernel (all instances of a CU resident at once)
  bool k1u_ok = false, k1u_auto = false;
  K1uPlan k1u{};
  size_t k1u_lds = 0;
  DevBuf<uint4> d_k1u_trace[2];
  DevBuf<int32_t> d_k1u_resets[2];
  // the histogram of segment k runs on a second stream under the chain kernel of segment k + 1 (two trace buffers)
  bool k1u_overlap = false, k1u_overlap_fits = false, k1u_pending = false;
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_trace[2] = {nullptr, nullptr}, ev_hist[2] = {nullptr, nullptr};
  bool ev_hist_used[2] = {false, false}, k1u_last_overlap = false;
  int64_t k1u_seq = 0;
  hipEvent_t ev_k1u[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // CMDP_STAT_ROLLOUT_KERNEL_MS / _HIST_KERNEL_MS of the last segment
  // K1E: the episode-parallel rollout (cmdp_k1e.h): lane = (instance, episode), private {successor | count} tables in LDS
  bool k1e_ok = false;
  K1ePlan k1e{};
  size_t k1e_lds = 0;
  DevBuf<uint32_t> d_etab;
  DevBuf<uint2> d_k1e_codes[2];    // two sets: the reward scan of one segment runs (second stream) under the walk of the next
  DevBuf<uint32_t> d_k1e_cnts[2];
  DevBuf<int32_t> d_k1e_h0b;       // second seg_h0 buffer
  hipEvent_t ev_k1e_walk[2] = {nullptr, nullptr}, ev_k1e_scan[2] = {nullptr, nullptr};
  bool ev_k1e_scan_used[2] = {false, false}, k1e_scan_pending = false;
  int64_t k1e_seq = 0;
  DevBuf<int2> d_k1e_dep;          // departure counts of the K1E launches since the last fold (k_epi_fold)
  DevBuf<int32_t> d_k1e_dep_res, d_vis_ovf;
  int64_t vis_bound = 0;           // upper bound of every device visit counter (int32): CMDP_ERR_OVERFLOW guard
  bool k1e_pending = false;        // d_k1e_dep holds counts the visit counters do not have yet
  int64_t k1e_pending_steps = 0;   // transitions per instance since the last fold (the departure image is int32)
  DevBuf<int32_t> d_k1e_h0;
  DevBuf<float> d_gp_q, d_gp_p;  // cmdp_greedy_policy_episodic workspace
  // K5S workspace (large-instance diameter)
  DevBuf<float> d_dl_v, d_ell_val;
  DevBuf<int32_t> d_ell_col, d_ell_newof;
  int ell_K = 0;
  int64_t relabel_min_states = 8192;  // CMDP_OPT_DIAMETER_RELABEL_MIN_STATES
  bool ell_relabelled = false;  // the fixed-width rows are stored in relabel_states' order (d_ell_newof: original -> new label)
  // K5T cluster tables (large-instance diameter with LDS tiles)
  DevBuf<int32_t> d_tl_c0, d_tl_ncl, d_tl_n, d_tl_R, d_tl_rows, d_tl_lcol;
  DevBuf<float> d_tl_val;
  int tile_K = 0;          // K the tables were built for (0: not built)
  double tile_rows_per_state = 0.0;  // tile rows gathered per state and sweep (1 + halo/cluster)
  DevBuf<int32_t> d_dl_inst, d_dl_t0, d_dl_cnt;
  DevBuf<int64_t> d_dl_voff;
  DevBuf<float> d_k5c_red;          // K5C: the clusters' partial reductions
  DevBuf<unsigned int> d_k5c_bar;   // K5C: barrier counters + error flag
  int64_t k5c_launches = 0, k5c_timeouts = 0;
  int k5c_skip = 0, k5c_backoff = 0;   // after a give-up K5C is skipped for `k5c_backoff` calls (8, 16, ... 1024), then tried again
  bool k5c_agent_scope = false;     // K5C: a cluster was found spread over XCDs once -- agent-scope barriers from then on
  size_t dl_ws_bytes = (size_t)24 << 30;  // value arrays of the target groups in flight per launch
  // observation tables (k_emit)
  DevBuf<float> d_obs_table, d_obs_out, d_obs_chol;
  DevBuf<unsigned long long> d_n_obs;
  int obs_F = 0, obs_time_indexed = 0;
  // cmdp_average_reward workspace (K9)
  int mixing_path = 0;       // CMDP_OPT_MIXING_PATH: 0 auto, 1 matrix powers, 2 stepping
  bool chain_exact = false;  // CMDP_OPT_CHAIN_EXACT_ORDER
  DevBuf<double> d_ch_work, d_ch_avg;
  DevBuf<int64_t> d_ch_off;
  DevBuf<int32_t> d_ch_kind, d_ch_ncls, d_ch_act, d_ch_start, d_ch_idx;
  DevBuf<uint8_t> d_ch_mask;
  // K9F: fill-reducing elimination plan of every instance (build_chain_plan), built at the first average-reward call
  bool chain_plan_built = false, chain_plan_any = false;
  DevBuf<int32_t> d_cf_rank, d_cf_cptr, d_cf_nrounds, d_cf_rptr, d_cf_piv;
  DevBuf<int64_t> d_cf_cbase, d_cf_rbase;
  DevBuf<uint16_t> d_cf_cand;
  DevBuf<uint8_t> d_cf_slow;
  DevBuf<float> d_dense;  // CMDP_LAYOUT_DENSE: [R][dense_spad]
  int dense_spad = 0;
  DevBuf<uint16_t> d_next16;
  DevBuf<uint8_t> d_rcode;
  DevBuf<double> d_rvals;

  EnvTables env() {
    EnvTables t{};
    t.B = B; t.A = A; t.H = H; t.rng_mode = rng_mode;
    t.rscale = rmax - rmin; t.rmin = rmin; t.beta_gammas = beta_gammas ? 1 : 0;
    t.state_off = d_state_off.p; t.entry_base = d_entry_base.p; t.row = d_row.p;
    t.sp_next = d_sp_next.p; t.sp_cum = d_sp_cum.p; t.sp_reward = d_sp_reward.p;
    t.sp_rkind = d_sp_rkind.p; t.sp_rp0 = d_sp_rp0.p; t.sp_rp1 = d_sp_rp1.p;
    t.start_off = d_start_off.p; t.start_state = d_start_state.p; t.start_cum = d_start_cum.p;
    t.start_slot = d_start_slot.p; t.philox_key = d_key.p; t.mt = d_mt.p; t.mt_pos = d_mt_pos.p;
    t.cur = d_cur.p; t.hstep = d_h.p; t.need_reset = d_need_reset.p; t.n_trans = d_ntrans.p; t.n_reset = d_nreset.p;
    t.visits_s = d_visits_s.p; t.visits_sa = d_visits_sa.p; t.last_start = d_last_start.p; t.prev_start = d_prev_start.p;
    return t;
  }
};

namespace {

// the main stream waits for the histogram the second stream still owes (K1U): before anything reads or writes the counters
int k1u_join(cmdp_t* h) {
  if (h->k1u_pending) {
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_hist[(h->k1u_seq + 1) & 1], 0));
    h->k1u_pending = false;
  }
  return CMDP_OK;
}

// ... the main stream waits for the reward scan K1E still owes on the second stream: before anything reads the reward sums
int k1e_scan_join(cmdp_t* h) {
  if (h->k1e_scan_pending) {
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_k1e_scan[(h->k1e_seq + 1) & 1], 0));
    h->k1e_scan_pending = false;
  }
  return CMDP_OK;
}

// ... and the departure counts the episode-parallel rollout K1E has accumulated are turned into the visit counters
int k1e_fold(cmdp_t* h);
int visits_join(cmdp_t* h) {
  if (int rc = k1u_join(h)) return rc;
  if (int rc = k1e_scan_join(h)) return rc;
  return k1e_fold(h);
}

// The device visit counters are int32.  No counter can grow by more than two per transition (the arrival and, when the
// episode ends there, the reset), so `vis_bound` bounds all of them; a call that could carry one past 2^31 - 1 is refused.
int visits_room(cmdp_t* h, int64_t n_transitions) {
  if (h->vis_bound + 2 * n_transitions > 0x7fffffffLL)
    return fail(CMDP_ERR_OVERFLOW, "a visit counter (int32 on the device) could wrap: up to %lld counted since the last "
                "cmdp_reset_visits / cmdp_set_visits, %lld more transitions asked for -- read the counters (cmdp_visits) and reset them",
                (long long)h->vis_bound, (long long)n_transitions);
  h->vis_bound += 2 * n_transitions;
  return CMDP_OK;
}

int bind(cmdp_t* h, bool join = true) {
  if (!h) return fail(CMDP_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (join) return visits_join(h);
  return CMDP_OK;
}

inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }

// dynamic LDS of a K1L / K1P workgroup of g instances
size_t k1l_lds_bytes(const LdsPlan& p, int g) {
  const int rings = p.pipe ? 2 * K1P_ACT_STRIDE(p.ch) + 2 * K1P_TR_STRIDE(p.ch) : 2 * p.ch;
  return (size_t)K1L_FIXED + (size_t)g * (size_t)(p.slot_bytes + rings);
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return CMDP_OK;
}

int k1e_fold(cmdp_t* h) {
  if (!h->k1e_pending) return CMDP_OK;
  K1ePlan e = h->k1e;
  e.dep = h->d_k1e_dep.p;
  e.dep_res = h->d_k1e_dep_res.p;
  const size_t lds = k1e_fold_lds_bytes(e);
  if (int rc = set_lds(k_epi_fold, lds)) return rc;
  hipLaunchKernelGGL(k_epi_fold, dim3(grid_for(h->B, K1E_NI)), dim3(K1E_THREADS), lds, h->stream, h->env(), e, h->d_vis_ovf.p);
  HIP_TRY(hipGetLastError());
  h->k1e_pending = false;
  h->k1e_pending_steps = 0;
  return CMDP_OK;
}


}  // namespace

// Host side of K1S (cmdp_k1s.h): compresses the sampler tables of a batch with stochastic dynamics into shared
// cumulative-probability patterns, per-state successor sets and 4-bit entry codes, and sizes the LDS plan.  Leaves
// h->k1s_ok false (the batch then takes K1) whenever a limit of the format is exceeded.
static int build_k1s(cmdp_t* h, const cmdp_desc* d) {
  const int B = h->B, A = h->A;
  if (h->rng_mode != CMDP_RNG_PHILOX || h->sample_beta || h->reward_cache || h->layout != CMDP_LAYOUT_CSR) return CMDP_OK;
  const int64_t S0 = h->max_S;   // slots are sized for the largest instance
  if (S0 * A >= 65536 || S0 < 1) return CMDP_OK;
  if (d->sp_rkind)
    for (int64_t e = 0; e < h->n_entries; ++e)
      if (d->sp_rkind[e] != 0) return CMDP_OK;   // reward means of Beta entries: K1 reports them
  const int64_t R = h->n_rows, NS = h->n_states;
  std::vector<uint16_t> shape((size_t)R, 0);
  std::vector<uint4> dict;
  std::map<std::tuple<int, unsigned long long, int>, int> shape_of;
  std::vector<unsigned long long> words((size_t)R, 0);
  std::vector<uint8_t> pat_ids((size_t)R, 0);
  std::vector<double> patterns;
  std::map<std::vector<uint64_t>, int> pat_of;
  std::map<uint64_t, int> code_of;
  std::vector<double> rvals;
  std::vector<std::vector<int32_t>> sets((size_t)NS);
  // reward value of every entry -> code
  std::vector<int> ecode((size_t)h->n_entries);
  for (int64_t e = 0; e < h->n_entries; ++e) {
    uint64_t bits;
    std::memcpy(&bits, &d->sp_reward[e], sizeof bits);
    auto it = code_of.find(bits);
    if (it == code_of.end()) {
      if (rvals.size() == 256) return CMDP_OK;
      it = code_of.emplace(bits, (int)rvals.size()).first;
      rvals.push_back(d->sp_reward[e]);
    }
    ecode[(size_t)e] = it->second;
  }
  int U = 1;
  bool by_state = true, by_row = true;      // is the reward a function of the successor state alone / of the row alone?
  std::vector<int> rc_state((size_t)NS, -1), rc_row((size_t)R, -1);
  for (int b = 0; b < B; ++b) {
    const int64_t so = h->state_off[b], Sb = h->state_off[b + 1] - so;
    for (int64_t s = 0; s < Sb; ++s) {
      auto& set = sets[(size_t)(so + s)];
      for (int a = 0; a < A; ++a) {
        const int64_t r = (so + s) * A + a;
        const int64_t lo = d->sp_ptr[r], hi = d->sp_ptr[r + 1];
        const int n = (int)(hi - lo);
        if (n < 1 || n > K1S_MAXE) return CMDP_OK;
        std::vector<uint64_t> key((size_t)n);
        std::memcpy(key.data(), d->sp_cum + lo, sizeof(double) * (size_t)n);
        auto it = pat_of.find(key);
        if (it == pat_of.end()) {
          if (pat_of.size() == 64) return CMDP_OK;
          it = pat_of.emplace(key, (int)pat_of.size()).first;
          for (int k = 0; k < K1S_MAXE; ++k)
            patterns.push_back(k < n - 1 ? d->sp_cum[lo + k] : std::numeric_limits<double>::infinity());
          patterns.push_back(d->sp_cum[hi - 1]);
        }
        const int pat_id = it->second;
        unsigned long long word = 0;
        for (int k = 0; k < n; ++k) {
          const int32_t nx = d->sp_next[lo + k];
          int idx = -1;
          for (size_t j = 0; j < set.size(); ++j)
            if (set[j] == nx) { idx = (int)j; break; }
          if (idx < 0) {
            if (set.size() == 16) return CMDP_OK;
            idx = (int)set.size();
            set.push_back(nx);
          }
          word |= (unsigned long long)idx << (4 * k);
          const int c = ecode[(size_t)(lo + k)];
          if (rc_state[(size_t)(so + nx)] < 0) rc_state[(size_t)(so + nx)] = c;
          else if (rc_state[(size_t)(so + nx)] != c) by_state = false;
          if (rc_row[(size_t)r] < 0) rc_row[(size_t)r] = c;
          else if (rc_row[(size_t)r] != c) by_row = false;
        }
        words[(size_t)r] = word;
        pat_ids[(size_t)r] = (uint8_t)pat_id;
      }
      U = std::max(U, (int)set.size());
    }
    if (d->start_off[b + 1] - d->start_off[b] > K1S_MAXSTART) return CMDP_OK;
  }
  if (!by_state && !by_row) return CMDP_OK;
  // row shapes: (pattern, word) -- and, when the reward is a function of the row rather than of the successor state, the
  // row's reward code, so that the shape's dictionary entry carries it
  for (int64_t r = 0; r < R; ++r) {
    const int rcd = by_state ? 0 : std::max(0, rc_row[(size_t)r]);
    const auto key = std::make_tuple((int)pat_ids[(size_t)r], words[(size_t)r], rcd);
    auto sh = shape_of.find(key);
    if (sh == shape_of.end()) {
      if (dict.size() == 65535) return CMDP_OK;
      sh = shape_of.emplace(key, (int)dict.size()).first;
      dict.push_back(make_uint4((uint32_t)words[(size_t)r], (uint32_t)(words[(size_t)r] >> 32), (uint32_t)pat_ids[(size_t)r], (uint32_t)rcd));
    }
    shape[(size_t)r] = (uint16_t)sh->second;
  }
  K1sPlan p{};
  p.S = (int)S0; p.rows = (int)S0 * A; p.U = U; p.n_pat = (int)pat_of.size(); p.n_codes = (int)rvals.size();
  p.reward_mode = by_state ? 0 : 1;
  p.ch = 32;
  p.n_shapes = (int)dict.size();
  p.shape_bytes = p.n_shapes <= 256 ? 1 : 2;
  auto up8 = [](int x) { return (x + 7) & ~7; };
  p.off_cnt = up8(p.rows * p.shape_bytes);
  p.off_ovf = up8(p.off_cnt + p.rows);
  p.off_sets = up8(p.off_ovf + 2 * (K1S_OVF + 2));
  // reward code of the arrival state in the top four bits of its successor-set entries (no separate look-up on the walk)
  // when both fit sixteen bits; per-row codes travel in the shape's dictionary entry: no per-instance code table then
  p.rc_packed = (by_state && S0 <= 4096 && rvals.size() <= 16) ? 1 : 0;
  p.off_rc = up8(p.off_sets + 2 * p.S * U);
  p.off_start = up8(p.off_rc + ((by_state && !p.rc_packed) ? p.S : 0));
  p.slot_bytes = up8(p.off_start + 48 + 8 * K1S_MAXSTART + 4 * K1S_MAXSTART);
  const size_t fixed = k1s_fixed_bytes(p.n_pat, p.n_shapes) + 64;
  const size_t per = (size_t)p.slot_bytes + k1s_ring_bytes(p.ch);
  if (fixed + 4 * per > (size_t)kLdsBudget) return CMDP_OK;   // fewer than four instances per CU: not worth it
  const int cap = (int)std::min<size_t>(64, ((size_t)kLdsBudget - fixed) / per);
  // the fewest instances per workgroup that keep the number of rounds (as for K1L)
  const int64_t wgs = (B + cap - 1) / cap, rounds = (wgs + h->cus - 1) / h->cus;
  p.G = (int)std::min<int64_t>(cap, std::max<int64_t>(1, (B + rounds * h->cus - 1) / (rounds * h->cus)));
  // walker wavefronts and lanes per instance in them.  Round 2 (ONE walker wavefront; FrozenLake-20 / MiniGrid-8 /
  // DeepSea-20 with p_rand, G = 8 / 11 / 22): teams of 8 (two entries per lane, two ballots) +19 %; teams of 4 (four
  // ballots) -3 %; of 2 -31 % against a lane per instance counting its 16 entries itself -- so teams only where a lane
  // gets at most two entries.  With up to four walker wavefronts a wavefront has a quarter of the instances and its teams
  // are larger.  CMDP_K1S_NW / CMDP_K1S_TEAM override (tuning aids, read per handle).
  p.nw = p.G >= 4 ? 4 : (p.G >= 2 ? 2 : 1);
  if (const char* e = std::getenv("CMDP_K1S_NW")) p.nw = std::max(1, std::min(4, std::atoi(e)));
  p.nw = std::min(p.nw, p.G);
  p.gw = (p.G + p.nw - 1) / p.nw;
  p.team = p.gw <= 4 ? 16 : (p.gw <= 8 ? 8 : 1);
  if (const char* e = std::getenv("CMDP_K1S_TEAM")) {
    const int tm = std::atoi(e);
    if ((tm == 1 || tm == 2 || tm == 4 || tm == 8 || tm == 16) && tm * p.gw <= 64) p.team = tm;
  }
  hipStream_t st = h->stream;
  std::vector<uint16_t> sets_flat((size_t)NS * U, 0);
  for (int b = 0; b < B; ++b) {
    const int64_t so = h->state_off[b];
    for (int64_t s = so; s < h->state_off[b + 1]; ++s)
      for (size_t j = 0; j < sets[(size_t)s].size(); ++j) {
        const int32_t nx = sets[(size_t)s][j];
        const int code = p.rc_packed ? std::max(0, rc_state[(size_t)(so + nx)]) : 0;
        sets_flat[(size_t)s * U + j] = (uint16_t)(nx | (code << 12));
      }
  }
  std::vector<uint8_t> rc(by_state ? (size_t)NS : (size_t)R, 0);
  for (size_t i = 0; i < rc.size(); ++i) rc[i] = (uint8_t)std::max(0, by_state ? rc_state[i] : rc_row[i]);
  if (p.shape_bytes == 1) {
    std::vector<uint8_t> s8((size_t)R);
    for (int64_t r = 0; r < R; ++r) s8[(size_t)r] = (uint8_t)shape[(size_t)r];
    HIP_TRY(h->d_k1s_pat.upload(s8.data(), s8.size(), st));
    HIP_TRY(hipStreamSynchronize(st));
    p.shape = h->d_k1s_pat.p;
  } else {
    HIP_TRY(h->d_k1s_shape16.upload(shape.data(), shape.size(), st));
    p.shape = h->d_k1s_shape16.p;
  }
  HIP_TRY(h->d_k1s_dict.upload(dict.data(), dict.size(), st));
  HIP_TRY(h->d_k1s_sets.upload(sets_flat.data(), sets_flat.size(), st));
  HIP_TRY(h->d_k1s_rc.upload(rc.data(), rc.size(), st));
  HIP_TRY(h->d_k1s_patterns.upload(patterns.data(), patterns.size(), st));
  HIP_TRY(h->d_k1s_rvals.upload(rvals.data(), rvals.size(), st));
  HIP_TRY(hipStreamSynchronize(st));
  p.dict = h->d_k1s_dict.p; p.sets = h->d_k1s_sets.p; p.rcode = h->d_k1s_rc.p;
  p.patterns = h->d_k1s_patterns.p; p.rvals = h->d_k1s_rvals.p;
  h->k1s = p;
  h->k1s_bytes = fixed + (size_t)p.G * per + 16;
  h->k1s_ok = true;
  return CMDP_OK;
}

extern "C" {

int cmdp_version(void) { return CMDP_ABI_VERSION; }
#ifndef CMDP_BUILD_ID
#define CMDP_BUILD_ID "unstamped"
#endif
static const char k_build_id[] = "CMDP_BUILD_ID=" CMDP_BUILD_ID;  /* the tag lets build() read the stamp from the file */
const char* cmdp_build_id(void) { return k_build_id + 14; }

const char* cmdp_last_error(void) { return g_err.c_str(); }

int cmdp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cmdp_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  // CMDP_SYNC_MODE = spin | yield | block: how host threads wait in hipStreamSynchronize (tuning aid for hosts with a CPU
  // quota: several threads driving batches concurrently each spin on a core by default)
  if (const char* e = std::getenv("CMDP_SYNC_MODE")) {
    const unsigned f = !std::strcmp(e, "block") ? hipDeviceScheduleBlockingSync
                     : !std::strcmp(e, "yield") ? hipDeviceScheduleYield
                     : !std::strcmp(e, "spin") ? hipDeviceScheduleSpin : hipDeviceScheduleAuto;
    HIP_TRY(hipSetDeviceFlags(f));
  }
  return CMDP_OK;
}

void* cmdp_stream(cmdp_t* h) { return h ? (void*)h->stream : nullptr; }

int cmdp_destroy(cmdp_t* h) {
  if (h && h->ev_dp0) {
    (void)hipEventDestroy(h->ev_dp0);
    (void)hipEventDestroy(h->ev_dp1);
  }
  if (!h) return CMDP_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  for (int i = 0; i < 2; ++i) {
    if (h->ev_k1e_walk[i]) (void)hipEventDestroy(h->ev_k1e_walk[i]);
    if (h->ev_k1e_scan[i]) (void)hipEventDestroy(h->ev_k1e_scan[i]);
  }
  if (h->aux_stream) {
    (void)hipStreamSynchronize(h->aux_stream);
    (void)hipStreamDestroy(h->aux_stream);
  }
  for (int i = 0; i < 2; ++i) {
    if (h->ev_trace[i]) (void)hipEventDestroy(h->ev_trace[i]);
    if (h->ev_hist[i]) (void)hipEventDestroy(h->ev_hist[i]);
  }
  for (int i = 0; i < 5; ++i)
    if (h->ev_k1u[i]) (void)hipEventDestroy(h->ev_k1u[i]);
  for (int i = 0; i < 2; ++i)
    if (h->ev_row[i]) (void)hipEventDestroy(h->ev_row[i]);
  for (double* c : h->rc_chunks) (void)hipFree(c);
  if (h->rc_stage_h) (void)hipHostFree(h->rc_stage_h);
  if (h->rc_dst_h) (void)hipHostFree(h->rc_dst_h);
  if (h->rc_ent_h) (void)hipHostFree(h->rc_ent_h);
  if (h->rc_list_h) (void)hipHostFree(h->rc_list_h);
  if (h->rc_pend_h) (void)hipHostFree(h->rc_pend_h);
  delete h;
  return CMDP_OK;
}

int cmdp_create(cmdp_t** out, const cmdp_desc* d) {
  if (!out || !d) return fail(CMDP_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->n_instances < 1 || d->n_actions < 1 || d->n_actions > 64 || d->horizon < 0)
    return fail(CMDP_ERR_INVALID, "n_instances/n_actions/horizon out of range (1 <= A <= 64)");
  if (d->rng_mode != CMDP_RNG_MT_COMPAT && d->rng_mode != CMDP_RNG_PHILOX) return fail(CMDP_ERR_INVALID, "rng_mode");
  if (d->layout != CMDP_LAYOUT_CSR && d->layout != CMDP_LAYOUT_DENSE) return fail(CMDP_ERR_INVALID, "layout");
  if (d->layout == CMDP_LAYOUT_DENSE && (d->rng_mode != CMDP_RNG_PHILOX || !d->sp_ptr || !d->csr_ptr))
    return fail(CMDP_ERR_INVALID, "CMDP_LAYOUT_DENSE needs CMDP_RNG_PHILOX and both halves of the description "
                                  "(the float32 rows come from the DP half, rewards and starts from the sampler half)");
  if (!d->state_off) return fail(CMDP_ERR_INVALID, "state_off is required");
  const bool has_env = d->sp_ptr != nullptr;
  const bool has_dp = d->csr_ptr != nullptr;
  if (!has_env && !has_dp) return fail(CMDP_ERR_INVALID, "neither the sampler half nor the DP half is present");
  if (has_env && (!d->sp_next || !d->sp_cum || !d->sp_reward || !d->start_off || !d->start_state || !d->start_cum))
    return fail(CMDP_ERR_INVALID, "sampler half is incomplete");
  if (has_env && d->rng_mode == CMDP_RNG_MT_COMPAT && (!d->sp_seed || !d->start_seed))
    return fail(CMDP_ERR_INVALID, "MT_COMPAT needs sp_seed and start_seed");
  if (has_dp && (!d->csr_col || !d->csr_val || !d->R)) return fail(CMDP_ERR_INVALID, "DP half is incomplete");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CMDP_ERR_NO_DEVICE, "no HIP device visible");

  const int B = d->n_instances, A = d->n_actions;
  if (d->state_off[0] != 0) return fail(CMDP_ERR_INVALID, "state_off[0] != 0");
  int max_S = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t S = d->state_off[b + 1] - d->state_off[b];
    if (S < 1 || S > (1 << 28)) return fail(CMDP_ERR_INVALID, "instance %d has %lld states", b, (long long)S);
    max_S = std::max<int>(max_S, (int)S);
  }
  const int64_t NS = d->state_off[B], R = NS * A;

  cmdp_t* h = new cmdp;
  struct Guard {
    cmdp_t* h;
    ~Guard() { if (h) cmdp_destroy(h); }
  } guard{h};
  HIP_TRY(hipGetDevice(&h->device));
  if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->cus < 1) h->cus = 256;
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  hipStream_t st = h->stream;
  h->B = B; h->A = A; h->H = d->horizon; h->rng_mode = d->rng_mode; h->layout = d->layout;
  h->rmin = d->reward_min; h->rmax = d->reward_max;
  h->n_states = NS; h->n_rows = R; h->max_S = max_S;
  h->has_env = has_env; h->has_dp = has_dp;
  h->state_off.assign(d->state_off, d->state_off + B + 1);
  HIP_TRY(h->d_state_off.upload(d->state_off, B + 1, st));
  HIP_TRY(h->d_flag.alloc(1));

  if (has_env) {
    const int64_t E = d->sp_ptr[R];
    h->n_entries = E;
    if (d->sp_ptr[0] != 0) return fail(CMDP_ERR_INVALID, "sp_ptr[0] != 0");
    bool any_beta = false;
    if (d->sp_rkind)
      for (int64_t e = 0; e < E; ++e) {
        if (d->sp_rkind[e] > 1) return fail(CMDP_ERR_UNSUPPORTED, "unknown reward distribution kind at entry %lld", (long long)e);
        any_beta |= d->sp_rkind[e] == 1;
      }
    if ((d->flags & CMDP_FLAG_REWARD_MEANS) && (d->flags & CMDP_FLAG_REWARD_CACHE))
      return fail(CMDP_ERR_INVALID, "CMDP_FLAG_REWARD_MEANS and CMDP_FLAG_REWARD_CACHE exclude each other");
    const bool reward_cache = any_beta && (d->flags & CMDP_FLAG_REWARD_CACHE);
    const bool sample_beta = any_beta && !(d->flags & (CMDP_FLAG_REWARD_MEANS | CMDP_FLAG_REWARD_CACHE));
    h->sample_beta = sample_beta;
    h->beta_gammas = (d->flags & CMDP_FLAG_BETA_GAMMAS) != 0;
    h->reward_cache = reward_cache;
    if (reward_cache) {
      if (!d->sp_rp0 || !d->sp_rp1 || d->layout != CMDP_LAYOUT_CSR)
        return fail(CMDP_ERR_INVALID, "CMDP_FLAG_REWARD_CACHE needs sp_rp0 / sp_rp1 and the CSR layout");
      if (E > 0x7fffffffLL) return fail(CMDP_ERR_UNSUPPORTED, "CMDP_FLAG_REWARD_CACHE: more than 2^31 entries");
      for (int64_t e = 0; e < E; ++e)
        if (d->sp_rkind[e] == 1 && !(d->sp_rp0[e] > 0.0 && d->sp_rp1[e] > 0.0))
          return fail(CMDP_ERR_INVALID, "Beta parameters must be positive (entry %lld)", (long long)e);
      h->h_rkind.assign(d->sp_rkind, d->sp_rkind + E);
      h->h_rp0.assign(d->sp_rp0, d->sp_rp0 + E);
      h->h_rp1.assign(d->sp_rp1, d->sp_rp1 + E);
      // the reference keys its caches by (node, action, next_node): entries of a row that name the same successor
      // (p_rand adds repeated successors) share one cache -- represented by the first of them
      h->h_canon.resize((size_t)E);
      for (int64_t r = 0; r < R; ++r) {
        const int64_t lo = d->sp_ptr[r], hi = d->sp_ptr[r + 1];
        for (int64_t e = lo; e < hi; ++e) {
          int64_t c = e;
          for (int64_t f = lo; f < e; ++f)
            if (d->sp_next[f] == d->sp_next[e]) { c = f; break; }
          h->h_canon[(size_t)e] = (int32_t)c;
        }
      }
      HIP_TRY(h->d_sp_rkind.upload(d->sp_rkind, E, st));
      HIP_TRY(h->d_rc_canon.upload(h->h_canon.data(), E, st));
      HIP_TRY(h->d_rc_blk.alloc(E)); HIP_TRY(h->d_rc_blk.zero(st));
      HIP_TRY(h->d_rc_pos.alloc(E)); HIP_TRY(h->d_rc_pos.zero(st));
      HIP_TRY(h->d_rc_pend_e.alloc(B)); HIP_TRY(hipMemsetAsync(h->d_rc_pend_e.p, 0xff, sizeof(int32_t) * B, st));
      HIP_TRY(h->d_rc_pend_prev.alloc(B)); HIP_TRY(h->d_rc_pend_prev.zero(st));
      HIP_TRY(h->d_rc_pend_act.alloc(B)); HIP_TRY(h->d_rc_pend_act.zero(st));
      HIP_TRY(h->d_rc_park_count.alloc(1)); HIP_TRY(h->d_rc_park_count.zero(st));
      HIP_TRY(h->d_rc_park_list.alloc(B));
      HIP_TRY(h->d_rc_left.alloc(B)); HIP_TRY(h->d_rc_left.zero(st));
      h->rc_blk_h.assign((size_t)E, nullptr);
      h->rc_cap = std::min(B, 1024);
      h->rc_chunk_blocks = (size_t)std::max(256, std::min(B * 8, 4096));  // 10 .. 164 MB per chunk
      HIP_TRY(h->d_rc_stage.alloc((size_t)h->rc_cap * CMDP_RC_BLOCK));
      HIP_TRY(h->d_rc_dst.alloc(h->rc_cap));
      HIP_TRY(h->d_rc_ent.alloc(h->rc_cap));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rc_stage_h), sizeof(double) * (size_t)h->rc_cap * CMDP_RC_BLOCK, 0));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rc_dst_h), sizeof(double*) * (size_t)h->rc_cap, 0));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rc_ent_h), sizeof(int32_t) * (size_t)h->rc_cap, 0));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rc_list_h), sizeof(int32_t) * (size_t)B, 0));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rc_pend_h), sizeof(int32_t) * (size_t)B, 0));
    }
    if (sample_beta) {
      if (d->rng_mode != CMDP_RNG_PHILOX || !d->sp_rp0 || !d->sp_rp1)
        return fail(CMDP_ERR_UNSUPPORTED, "Beta rewards are sampled on the device only in CMDP_RNG_PHILOX mode with sp_rp0/"
                                          "sp_rp1; the reference-exact stream is host side (CMDP_FLAG_REWARD_MEANS)");
      for (int64_t e = 0; e < E; ++e)
        if (d->sp_rkind[e] == 1 && !(d->sp_rp0[e] > 0.0 && d->sp_rp1[e] > 0.0))
          return fail(CMDP_ERR_INVALID, "Beta parameters must be positive (entry %lld)", (long long)e);
      HIP_TRY(h->d_sp_rkind.upload(d->sp_rkind, E, st));
      HIP_TRY(h->d_sp_rp0.upload(d->sp_rp0, E, st));
      HIP_TRY(h->d_sp_rp1.upload(d->sp_rp1, E, st));
    }
    // row descriptors, entry bases, MT slots -- validated on the host so that no kernel can index out of range
    std::vector<RowDesc> rows((size_t)R);
    std::vector<int64_t> ebase((size_t)B);
    std::vector<int32_t> seeds;
    for (int b = 0; b < B; ++b) {
      const int64_t s0 = d->state_off[b], S = d->state_off[b + 1] - s0;
      const int64_t r0 = s0 * A, r1 = (s0 + S) * A;
      ebase[b] = d->sp_ptr[r0];
      for (int64_t r = r0; r < r1; ++r) {
        const int64_t lo = d->sp_ptr[r], n = d->sp_ptr[r + 1] - lo;
        if (n < 1 || n > 4096 || lo - ebase[b] > 0x7fffffffLL)
          return fail(CMDP_ERR_INVALID, "row %lld has %lld successors", (long long)r, (long long)n);
        for (int64_t e = lo; e < lo + n; ++e) {
          if (d->sp_next[e] < 0 || d->sp_next[e] >= S)
            return fail(CMDP_ERR_INVALID, "successor index out of range at entry %lld", (long long)e);
          if (e > lo && d->sp_cum[e] < d->sp_cum[e - 1])
            return fail(CMDP_ERR_INVALID, "sp_cum not non-decreasing at entry %lld", (long long)e);
        }
        RowDesc rd;
        rd.first = (int32_t)(lo - ebase[b]);
        rd.n = (int32_t)n;
        rd.next_if_det = d->sp_next[lo];
        rd.reward_if_det = d->sp_reward[lo];
        rd.pad = 0.0;
        rd.mt_slot = -1;
        if (n > 1 && d->rng_mode == CMDP_RNG_MT_COMPAT) {
          rd.mt_slot = (int32_t)seeds.size();
          seeds.push_back(d->sp_seed[r]);
        }
        rows[(size_t)r] = rd;
      }
    }
    std::vector<int32_t> start_slot((size_t)B, -1);
    if (d->start_off[0] != 0) return fail(CMDP_ERR_INVALID, "start_off[0] != 0");
    for (int b = 0; b < B; ++b) {
      const int64_t lo = d->start_off[b], n = d->start_off[b + 1] - lo;
      const int64_t S = d->state_off[b + 1] - d->state_off[b];
      if (n < 1) return fail(CMDP_ERR_INVALID, "instance %d has no starting state", b);
      for (int64_t i = lo; i < lo + n; ++i)
        if (d->start_state[i] < 0 || d->start_state[i] >= S)
          return fail(CMDP_ERR_INVALID, "starting state out of range (instance %d)", b);
      if (n > 1 && d->rng_mode == CMDP_RNG_MT_COMPAT) {
        start_slot[b] = (int32_t)seeds.size();
        seeds.push_back(d->start_seed[b]);
      }
    }
    if (seeds.size() > 0x7fffffffULL / 2) return fail(CMDP_ERR_INVALID, "too many MT19937 sampler streams");
    h->n_slots = (int64_t)seeds.size();
    const int64_t NSt = d->start_off[B];
    HIP_TRY(h->d_row.upload(rows.data(), rows.size(), st));
    HIP_TRY(h->d_entry_base.upload(ebase.data(), ebase.size(), st));
    HIP_TRY(h->d_sp_next.upload(d->sp_next, E, st));
    HIP_TRY(h->d_sp_cum.upload(d->sp_cum, E, st));
    HIP_TRY(h->d_sp_reward.upload(d->sp_reward, E, st));
    HIP_TRY(h->d_start_off.upload(d->start_off, B + 1, st));
    HIP_TRY(h->d_start_state.upload(d->start_state, NSt, st));
    HIP_TRY(h->d_start_cum.upload(d->start_cum, NSt, st));
    HIP_TRY(h->d_start_slot.upload(start_slot.data(), B, st));
    std::vector<uint2> keys((size_t)B);
    for (int b = 0; b < B; ++b) {
      const uint64_t k = d->philox_key ? d->philox_key[b] : 0;
      keys[b] = make_uint2((uint32_t)k, (uint32_t)(k >> 32));
    }
    HIP_TRY(h->d_key.upload(keys.data(), B, st));
    HIP_TRY(h->d_cur.alloc(B));
    HIP_TRY(h->d_cur.zero(st));
    HIP_TRY(h->d_last_start.alloc(B));
    HIP_TRY(h->d_last_start.zero(st));
    HIP_TRY(h->d_prev_start.alloc(B));
    HIP_TRY(h->d_prev_start.zero(st));
    HIP_TRY(h->d_h.alloc(B));
    HIP_TRY(h->d_h.zero(st));
    HIP_TRY(h->d_need_reset.alloc(B));
    HIP_TRY(hipMemsetAsync(h->d_need_reset.p, 1, B, st));  // BaseMDP starts with a reset pending
    HIP_TRY(h->d_ntrans.alloc(B));
    HIP_TRY(h->d_ntrans.zero(st));
    HIP_TRY(h->d_nreset.alloc(B));
    HIP_TRY(h->d_nreset.zero(st));
    HIP_TRY(h->d_visits_s.alloc(NS));
    HIP_TRY(h->d_visits_s.zero(st));
    HIP_TRY(h->d_visits_sa.alloc(R));
    HIP_TRY(h->d_visits_sa.zero(st));
    // ---- eligibility of the LDS-resident rollout kernel --------------------------------------------------
    {
      bool ok = max_S <= 65535 && h->n_slots == 0 && !sample_beta && !reward_cache;
      for (int b = 0; ok && b < B; ++b) ok = (d->state_off[b + 1] - d->state_off[b]) == max_S;  // uniform S
      for (int64_t r = 0; ok && r < R; ++r) ok = rows[(size_t)r].n == 1;
      for (int b = 0; ok && b < B; ++b) ok = (d->start_off[b + 1] - d->start_off[b]) == 1;
      std::vector<double> vals;
      std::unordered_map<uint64_t, int> code_of;
      std::vector<uint8_t> codes;
      std::vector<uint16_t> next16;
      if (ok) {
        codes.resize((size_t)R);
        next16.resize((size_t)R);
        for (int64_t r = 0; ok && r < R; ++r) {
          const double v = rows[(size_t)r].reward_if_det;
          uint64_t bits;
          std::memcpy(&bits, &v, sizeof bits);
          auto it = code_of.find(bits);
          if (it == code_of.end()) {
            if (vals.size() == 256) { ok = false; break; }
            it = code_of.emplace(bits, (int)vals.size()).first;
            vals.push_back(v);
          }
          codes[(size_t)r] = (uint8_t)it->second;
          next16[(size_t)r] = (uint16_t)rows[(size_t)r].next_if_det;
        }
      }
      if (ok) {
        const int rows_max = max_S * A;
        LdsPlan p{};
        p.rows_max = rows_max;
        // reward code in the upper bits of the successor word when both fit 16 bits: one table and one LDS read less
        // (the successor is stored as its row base, successor * A)
        int bits_s = 1, bits_c = 0;
        while ((1 << bits_s) < rows_max) ++bits_s;
        while ((1 << bits_c) < (int)vals.size()) ++bits_c;
        p.code_shift = (bits_s + bits_c <= 16) ? bits_s : 0;
        const bool pipe_ok = bits_s + 1 + bits_c <= 16;  // K1P stores the row base as a byte offset: one more bit
        p.off_rcode = (rows_max * 2 + 3) & ~3;
        if (p.code_shift) {  // packed: successor words, 8-bit count deltas (+ the walker's dummy counter), overflow list
          p.off_cnt = p.off_rcode;
          p.off_ovf = p.off_cnt + ((rows_max + 4 + 3) & ~3);
          p.slot_bytes = p.off_ovf + ((2 * K1L_OVF + 3) & ~3);
        } else {
          p.off_cnt = p.off_rcode + ((rows_max + 3) & ~3);
          p.off_ovf = 0;
          p.slot_bytes = p.off_cnt + (((rows_max + 1) / 2) * 4) + 4;  // + the walker's dummy count dword
        }
        const int fixed = K1L_FIXED;
        // The walk is bound by the latency of one transition times the number of "rounds" of workgroups the batch
        // needs (instances resident per CU are limited by LDS capacity).  Choose the action-ring chunk length and the
        // workgroups per CU (two overlap one group's staging / flush with the other's walk, one holds a few more
        // instances) that need the fewest rounds; ties go to the longer chunk (fewer barriers), then to two per CU.
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        if (cus < 1) cus = 256;
        h->cus = cus;
        double best_cost = -1.0;
        int best_cap = 0;  // LDS capacity (instances per workgroup) of the chosen candidate
        p.ch = 256;
        p.G = 0;
        const int force_ch = std::getenv("CMDP_K1L_CH") ? std::atoi(std::getenv("CMDP_K1L_CH")) : 0;      // tuning aids
        const int force_pipe = std::getenv("CMDP_K1L_PIPE") ? std::atoi(std::getenv("CMDP_K1L_PIPE")) : -1;
        struct Cand { int pipe, ch; };
        // Packed tables can also run as the wavefront pipeline K1P: ~0.62x the time per transition plus one barrier
        // per chunk (measured at C2: 53 / 56 / 62 ns per transition at ch = 64 / 32 / 16 against K1L's 82), for
        // 6 ch + 16 bytes of rings per instance instead of 2 ch.
        const Cand cands[] = {{0, 256}, {0, 128}, {0, 112}, {0, 64}, {1, 64}, {1, 32}, {1, 16}};
        for (const Cand& c : cands) {
          if (force_ch && c.ch != force_ch) continue;
          if (c.pipe && !pipe_ok) continue;
          if (force_pipe >= 0 && c.pipe != force_pipe) continue;
          const int pi = p.slot_bytes + (c.pipe ? 2 * K1P_ACT_STRIDE(c.ch) + 2 * K1P_TR_STRIDE(c.ch) : 2 * c.ch);
          const int fx = fixed;
          const int g1 = std::min<int>(64, (kLdsBudget - fx) / pi);
          // K1P: one workgroup per CU.  Two (26 + 26 instances at C2) need <= 128 VGPRs to be co-resident at all (it has
          // 145: the "1.8x slower" of the first trial was simply one resident group at a time); forced to 128 it spills,
          // and two resident groups gain nothing -- they run in lockstep, so their flushes coincide, and a CU holds
          // the same 52 instances either way.
          const int g2 = c.pipe ? 0 : std::min<int>(64, (kLdsBudget / 2 - fx) / pi);
          for (int per_cu : {2, 1}) {
            const int g = per_cu == 2 ? g2 : g1;
            if (g < (per_cu == 2 ? 12 : 8)) continue;
            const int64_t wgs = (B + g - 1) / g, slots_n = (int64_t)cus * per_cu;
            const int64_t rounds = (wgs + slots_n - 1) / slots_n;
            const double per_step = c.pipe ? 0.62 * (1.0 + 3.5 / c.ch) : 1.0 + 2.0 / c.ch;
            const double cost = (double)rounds * per_step;
            if (best_cost < 0 || cost < best_cost) {
              best_cost = cost;
              best_cap = g;
              p.ch = c.ch;
              p.pipe = c.pipe;
              // the fewest instances per workgroup that still need `rounds` rounds: evens out the last round and keeps
              // LDS bank conflicts down (53 instead of 52 lanes per walker measured +1.2 % at C2)
              p.G = (int)std::min<int64_t>(g, std::max<int64_t>(1, (B + rounds * slots_n - 1) / (rounds * slots_n)));
              h->lds_G1 = g1;
              h->lds_G2 = g2;
            }
          }
        }
        if (p.pipe) p.code_shift = bits_s + 1;
        if (p.code_shift)
          for (int64_t r = 0; r < R; ++r)
            next16[(size_t)r] = (uint16_t)((next16[(size_t)r] * A * (p.pipe ? 2 : 1)) | (codes[(size_t)r] << p.code_shift));
        p.n_codes = (int)vals.size();
        // K1T (cmdp_k1t.h): when every instance's packed words are, state by state, instance 0's words or their swap
        // (A = 2; seeds of a family whose structure does not depend on the seed only permute the actions), the workgroup
        // keeps ONE table and a swap bit per state and instance: 2-3 x the instances per CU.
        static const int k1t_env = std::getenv("CMDP_K1T") ? std::atoi(std::getenv("CMDP_K1T")) : -1;   // tuning aid: 0 off, 1 on
        // (K1T packs its own words -- successor row base as a byte offset | reward code above it, K1P's format -- whichever
        // of K1L / K1P the plan above chose: small instances plan onto two K1L workgroups per CU, and K1T still beats that)
        if (best_cap >= 8 && pipe_ok && A == 2 && k1t_env != 0) {
          const int S = max_S, rws = S * 2;
          const int cs_t = bits_s + 1;
          std::vector<uint16_t> w16((size_t)R);
          for (int64_t r = 0; r < R; ++r)
            w16[(size_t)r] = (uint16_t)((rows[(size_t)r].next_if_det * A * 2) | (codes[(size_t)r] << cs_t));
          TmplPlan q{};
          q.rows = rws;
          q.tmpl_bytes = (rws * 2 + 15) & ~15;
          q.mask_bytes = ((S + 7) / 8 + 3) & ~3;
          q.off_cnt = q.mask_bytes;
          q.off_ovf = q.off_cnt + ((rws + 4 + 3) & ~3);
          // two spare entries: the counts wavefront stores ovf[n_ovf] unconditionally before it knows whether a counter
          // wrapped (branch-free), so with the list full the store must still land inside the instance's own slot
          q.slot_bytes = q.off_ovf + ((2 * (K1T_OVF + 2) + 3) & ~3);
          if (((q.slot_bytes / 4) & 1) == 0) q.slot_bytes += 4;   // odd dword stride: the lanes' slots start on different banks
          q.n_codes = p.n_codes;
          q.code_shift = cs_t;
          if (const char* de = std::getenv("CMDP_K1T_DEBUG")) {   // timing experiments: stages switched off
            q.debug = std::atoi(de);
            if (q.debug) std::fprintf(stderr, "libcmdp: CMDP_K1T_DEBUG=%d switches stages of k_rollout_tmpl off -- results are INVALID (timing experiments only)\n", q.debug);
          }
          std::vector<uint8_t> bits((size_t)B * q.mask_bytes, 0);
          bool same = true;
          const uint16_t* T = w16.data();
          for (int b = 0; same && b < B; ++b) {
            const uint16_t* W = w16.data() + (size_t)b * rws;
            uint8_t* mb = bits.data() + (size_t)b * q.mask_bytes;
            for (int s2 = 0; s2 < S; ++s2) {
              const uint16_t w0 = W[2 * s2], w1 = W[2 * s2 + 1], t0 = T[2 * s2], t1 = T[2 * s2 + 1];
              if (w0 == t0 && w1 == t1) continue;
              if (w0 == t1 && w1 == t0) { mb[s2 >> 3] |= (uint8_t)(1u << (s2 & 7)); continue; }
              same = false;
              break;
            }
          }
          if (same) {
            // chunk length and instances per workgroup: fewest rounds x time per transition (K1T's chain carries ~4 more
            // dependent instructions than K1P's: ~1.3 x its time per transition), as for K1L / K1P above
            double best_t = -1.0;
            for (int ch : {64, 32, 16}) {
              if (force_ch && ch != force_ch) continue;
              q.ch = ch;
              const int per = q.slot_bytes + 2 * K1P_ACT_STRIDE(ch) + 2 * K1P_TR_STRIDE(ch);
              const int cap = std::min<int>(128, (kLdsBudget - K1T_FIXED - q.tmpl_bytes) / per);
              if (cap < 16) continue;
              const int64_t wgs = (B + cap - 1) / cap, rounds = (wgs + cus - 1) / cus;
              const double cost = (double)rounds * 1.3 * 0.62 * (1.0 + 3.5 / ch);
              if (best_t < 0 || cost < best_t) {
                best_t = cost;
                h->tmpl_plan = q;
                h->tmpl_plan.G = (int)std::min<int64_t>(cap, std::max<int64_t>(1, (B + rounds * cus - 1) / (rounds * cus)));
                if (const char* ge = std::getenv("CMDP_K1T_G"))   // tests: instances per workgroup (read per handle)
                  h->tmpl_plan.G = std::max(1, std::min(cap, std::atoi(ge)));
              }
            }
            if (best_t > 0) {
              std::vector<uint16_t> tw(T, T + rws);
              tw.resize((size_t)h->tmpl_plan.tmpl_bytes / 2, 0);
              HIP_TRY(h->d_tmpl_words.upload(tw.data(), tw.size(), st));
              HIP_TRY(h->d_swap_bits.upload(bits.data(), bits.size(), st));
              HIP_TRY(hipStreamSynchronize(st));
              h->tmpl_plan.tmpl = h->d_tmpl_words.p;
              h->tmpl_plan.swap_bits = h->d_swap_bits.p;
              h->tmpl_lds = k1t_lds_bytes(h->tmpl_plan, h->tmpl_plan.G);
              h->tmpl_ok = true;                                   // eligible: CMDP_OPT_ROLLOUT_KERNEL 4 may force it
              h->tmpl_auto = best_t < best_cost || k1t_env == 1;   // and the automatic choice when it needs fewer rounds x time
              // K1U: the same chain with the visit counts histogrammed from an HBM trace -- an instance keeps only its swap
              // bits and the rings in LDS, up to 256 instances per workgroup.  Taken automatically when that saves a round
              // of workgroups over K1T (config C2: one round of 256 instead of two of 128); otherwise the histogram pass
              // is pure overhead and K1T stays.
              {
                K1uPlan u{};
                u.rows = q.rows; u.tmpl_bytes = q.tmpl_bytes; u.mask_bytes = q.mask_bytes;
                u.slot_bytes = q.mask_bytes + ((((q.mask_bytes / 4) & 1) == 0) ? 4 : 0);
                u.n_codes = q.n_codes; u.code_shift = q.code_shift;
                u.pack10 = u.rows <= 1024 ? 1 : 0;
                if (const char* pe = std::getenv("CMDP_K1U_PACK10")) u.pack10 = (std::atoi(pe) != 0 && u.rows <= 1024) ? 1 : 0;
                // chunk length: one barrier per chunk (measured at C2: 2.20 ms per launch at 32 transitions, 2.06 ms at 64)
                u.ch = u.pack10 ? 72 : 64;
                if (const char* ce = std::getenv("CMDP_K1U_CH")) u.ch = std::max(8, std::min(240, std::atoi(ce)));
                u.ch = u.pack10 ? std::max(24, u.ch / 24 * 24) : std::max(8, u.ch & ~7);
                const int per = u.slot_bytes + 2 * K1P_ACT_STRIDE(u.ch) + 2 * K1P_TR_STRIDE(u.ch);
                const int cap = std::min<int>(256, (kLdsBudget - K1U_FIXED - u.tmpl_bytes) / per);
                if (cap >= 64 && k1h_lds_bytes(S, 64) <= (size_t)kLdsBudget) {
                  const int64_t wgs = (B + cap - 1) / cap, rounds_u = (wgs + cus - 1) / cus;
                  u.G = (int)std::min<int64_t>(cap, std::max<int64_t>(1, (B + rounds_u * cus - 1) / (rounds_u * cus)));
                  if (const char* ge = std::getenv("CMDP_K1U_G")) u.G = std::max(1, std::min(cap, std::atoi(ge)));
                  const int64_t wgs_t = (B + h->tmpl_plan.G - 1) / h->tmpl_plan.G, rounds_t = (wgs_t + cus - 1) / cus;
                  u.tmpl = h->d_tmpl_words.p; u.swap_bits = h->d_swap_bits.p;
                  h->k1u = u;
                  h->k1u_lds = k1u_lds_bytes(u, u.G);
                  h->k1u_ok = true;
                  h->k1u_auto = h->tmpl_auto && rounds_u < rounds_t;
                  // the histogram of one launch under the chain of the next (second stream) was measured at C2 and LOST:
                  // co-resident, the chain kernel slows from 2.2 to 3.0 ms (the histogram's LDS atomics sit in the same
                  // in-order LDS pipeline as the chain's dependent reads) -- 3.25 ms per step against 3.00 one after the
                  // other.  Kept behind CMDP_K1U_OVERLAP=1 for batches where both workgroups fit one CU.
                  h->k1u_overlap = false;
                  h->k1u_overlap_fits = h->k1u_lds + k1h_lds_bytes(S, 32) + 1024 <= (size_t)kLdsBudget;
                  if (const char* ue = std::getenv("CMDP_K1U")) h->k1u_auto = std::atoi(ue) != 0;
                }
              }
            }
          }
        }
        // K1E (cmdp_k1e.h): episodic batches with two actions and at most four distinct rewards walk their EPISODES in
        // parallel (private {successor word | count} tables of 32 instances per workgroup): throughput- instead of
        // latency-bound, and the tables need not be action-permuted copies of one MDP.
        static const int k1e_env = std::getenv("CMDP_K1E") ? std::atoi(std::getenv("CMDP_K1E")) : -1;   // tuning aid: 0 off
        if (A == 2 && h->H > 0 && h->H < (1 << 14) && vals.size() <= 4 && k1e_env != 0) {
          K1ePlan e{};
          e.S = max_S;
          e.H = h->H;
          e.n_codes = (int)vals.size();
          e.nch = (h->H + 31) / 32;
          // a wavefront's ring: the blocks the eight episodes of a round can touch (cmdp_k1e.h)
          const int64_t round_bits = (int64_t)2 * K1E_EPL * h->H;
          int rb = 2;
          while (rb < (round_bits + 126) / 128 + 1) rb <<= 1;
          e.ring_blocks = rb;
          e.ash = 12;   // (at least 32 padded states: an action's image then holds whole rounds of the workgroup's 1024 lanes)
          while ((1 << (e.ash - 7)) < max_S) ++e.ash;   // action stride: states padded to a power of two, 128 B per state
          if (const char* de = std::getenv("CMDP_K1E_DEBUG")) {   // timing experiments: phases switched off
            e.debug = std::atoi(de);
            if (e.debug) std::fprintf(stderr, "libcmdp: CMDP_K1E_DEBUG=%d switches phases of k_rollout_epi off -- results are INVALID (timing experiments only)\n", e.debug);
          }
          e.gdw = (int32_t)((((int64_t)max_S * K1E_NI + K1E_THREADS - 1) / K1E_THREADS) * K1E_THREADS);   // a group's image: whole rounds of the workgroup's loads
          const size_t k1e_behind = k1e_lds_bytes(e);
          const bool k1e_fits = k1e_behind <= (size_t)kLdsBudget;
          if (std::getenv("CMDP_K1E_VERBOSE"))
            std::fprintf(stderr, "libcmdp: K1E plan S %d H %d codes %d ash %d ring_blocks %d gdw %d tables+rings %zu B fits %d\n", e.S, e.H, e.n_codes,
                         e.ash, e.ring_blocks, e.gdw, k1e_behind, (int)k1e_fits);
          if (max_S <= 512 && k1e_fits) {
            // interleaved by instance like the LDS image: [group of 32][state][instance in group]
            const int64_t groups = ((int64_t)B + K1E_NI - 1) / K1E_NI;
            std::vector<uint32_t> et((size_t)groups * (size_t)e.gdw, 0u);
            for (int64_t b2 = 0; b2 < B; ++b2)
              for (int s2 = 0; s2 < max_S; ++s2) {
                const int64_t sidx = b2 * max_S + s2;
                const uint32_t w0 = ((uint32_t)rows[(size_t)(2 * sidx)].next_if_det << 7) | codes[(size_t)(2 * sidx)];
                const uint32_t w1 = ((uint32_t)rows[(size_t)(2 * sidx + 1)].next_if_det << 7) | codes[(size_t)(2 * sidx + 1)];
                et[(size_t)(b2 / K1E_NI) * (size_t)e.gdw + (size_t)s2 * K1E_NI + (size_t)(b2 % K1E_NI)] = w0 | (w1 << 16);
              }
            HIP_TRY(h->d_etab.upload(et.data(), et.size(), st));
            HIP_TRY(hipStreamSynchronize(st));
            e.etab = h->d_etab.p;
            h->k1e = e;
            h->k1e_lds = k1e_lds_bytes(e);
            h->k1e_ok = true;
          }
        }
        if (best_cap >= 8) {
          // 16 bytes of slack in front of and behind both element arrays: the staging loads are 16-byte wide
          // from the aligned-down address of a group's first element
          next16.insert(next16.begin(), 8, 0);
          next16.insert(next16.end(), 8, 0);
          codes.insert(codes.begin(), 16, 0);
          codes.insert(codes.end(), 16, 0);
          HIP_TRY(h->d_next16.upload(next16.data(), next16.size(), st));
          HIP_TRY(h->d_rcode.upload(codes.data(), codes.size(), st));
          HIP_TRY(h->d_rvals.upload(vals.data(), vals.size(), st));
          p.next16 = h->d_next16.p + 8; p.rcode = h->d_rcode.p + 16; p.rvals = h->d_rvals.p;
          h->lds_plan = p;
          h->lds_bytes = k1l_lds_bytes(p, p.G);
          h->lds_ok = true;
          h->tmpl_plan.rvals = p.rvals;
          h->k1u.rvals = p.rvals;
          h->k1e.rvals = p.rvals;
          HIP_TRY(hipStreamSynchronize(st));  // staging vectors die with this scope
        }
      }
    }
    if (!h->lds_ok) h->k1e_ok = false;   // (its reward values are uploaded with the K1L tables)
    if (!h->lds_ok)
      if (int rc = build_k1s(h, d)) return rc;
    if (h->n_slots) {
      DevBuf<int32_t> d_seeds;
      HIP_TRY(d_seeds.upload(seeds.data(), seeds.size(), st));
      HIP_TRY(h->d_mt.alloc((size_t)h->n_slots * 624));
      HIP_TRY(h->d_mt_pos.alloc(h->n_slots));
      hipLaunchKernelGGL(k_mt_seed, dim3(grid_for(h->n_slots, 64)), dim3(64), 0, st, h->d_mt.p, h->d_mt_pos.p,
                         d_seeds.p, h->n_slots);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(st));  // d_seeds is released at scope exit
    }
  }

  if (has_dp) {
    if (d->csr_ptr[0] != 0) return fail(CMDP_ERR_INVALID, "csr_ptr[0] != 0");
    const int64_t N = d->csr_ptr[R];
    h->n_csr = N;
    h->csr_nnz.resize(B);
    for (int b = 0; b < B; ++b) {
      const int64_t s0 = d->state_off[b], S = d->state_off[b + 1] - s0;
      const int64_t r0 = s0 * A, r1 = (s0 + S) * A;
      h->csr_nnz[b] = d->csr_ptr[r1] - d->csr_ptr[r0];
      if (h->csr_nnz[b] > 0x7fffffffLL) return fail(CMDP_ERR_INVALID, "instance %d has too many non-zeros", b);
      h->max_inst_nnz = std::max(h->max_inst_nnz, h->csr_nnz[b]);
      for (int64_t r = r0; r < r1; ++r) {
        if (d->csr_ptr[r + 1] < d->csr_ptr[r]) return fail(CMDP_ERR_INVALID, "csr_ptr decreasing at row %lld", (long long)r);
        h->max_row_nnz = std::max<int>(h->max_row_nnz, (int)std::min<int64_t>(d->csr_ptr[r + 1] - d->csr_ptr[r], 1 << 30));
        for (int64_t k = d->csr_ptr[r]; k < d->csr_ptr[r + 1]; ++k)
          if (d->csr_col[k] < 0 || d->csr_col[k] >= S)
            return fail(CMDP_ERR_INVALID, "csr_col out of range at %lld", (long long)k);
      }
    }
    // K2U (k_dp_regu): distinct successor columns per STATE over its A rows, and whether every row lists its columns
    // in strictly ascending order (the dense-over-the-distinct-set sum is the ascending-column sum)
    h->max_state_unique = 0;
    if (A <= 4 && h->max_row_nnz <= 8 && max_S <= 1024) {
      bool sorted = true;
      int32_t cols[32];
      for (int64_t s = 0; s < d->state_off[B] && sorted; ++s) {
        int n = 0;
        for (int a = 0; a < A; ++a) {
          const int64_t r = s * A + a;
          for (int64_t k = d->csr_ptr[r]; k < d->csr_ptr[r + 1]; ++k) {
            if (k > d->csr_ptr[r] && d->csr_col[k] <= d->csr_col[k - 1]) sorted = false;
            cols[n++] = d->csr_col[k];
          }
        }
        std::sort(cols, cols + n);
        const int u = (int)(std::unique(cols, cols + n) - cols);
        h->max_state_unique = std::max(h->max_state_unique, u);
      }
      if (!sorted) h->max_state_unique = 0;
    }
    HIP_TRY(h->d_csr_ptr.upload(d->csr_ptr, R + 1, st));
    HIP_TRY(h->d_csr_col.upload(d->csr_col, N, st));
    HIP_TRY(h->d_csr_val.upload(d->csr_val, N, st));
    HIP_TRY(h->d_R.upload(d->R, R, st));
  }
  if (d->layout == CMDP_LAYOUT_DENSE) {
    // exact float64 prefix sums need every probability to be a multiple of 2^-52 after scaling: p >= 2^-28
    for (int64_t k = 0; k < h->n_csr; ++k)
      if (!(d->csr_val[k] >= 3.7252902984619141e-09f) || d->csr_val[k] > 1.0f)
        return fail(CMDP_ERR_INVALID, "dense layout needs probabilities in [2^-28, 1] (entry %lld)", (long long)k);
    h->dense_spad = ((max_S + 255) / 256) * 256;
    {
      const int nv = h->dense_spad / 256;
      const int allowed[] = {1, 2, 3, 4, 6, 8, 12, 16};
      int pick = 0;
      for (int v : allowed) if (v >= nv) { pick = v; break; }
      if (!pick) return fail(CMDP_ERR_UNSUPPORTED, "dense layout supports at most 4096 states per instance");
      h->dense_spad = pick * 256;
    }
    const size_t n = (size_t)R * h->dense_spad;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (n * sizeof(float) > free_b)
      return fail(CMDP_ERR_INVALID, "dense layout needs %zu MiB, %zu MiB free", n * sizeof(float) >> 20, free_b >> 20);
    HIP_TRY(h->d_dense.alloc(n));
    HIP_TRY(h->d_dense.zero(st));
    hipLaunchKernelGGL(k_dense_fill, dim3(grid_for(R, 256)), dim3(256), 0, st, h->d_dense.p, h->dense_spad,
                       h->d_csr_ptr.p, h->d_csr_col.p, h->d_csr_val.p, R);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(st));  // host staging vectors go out of scope
  guard.h = nullptr;
  *out = h;
  return CMDP_OK;
}

// ---- interaction --------------------------------------------------------------------------------------
int cmdp_reset(cmdp_t* h, const uint8_t* mask, int32_t* obs_out) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  if (int rc = visits_room(h, 1)) return rc;
  hipStream_t st = h->stream;
  uint8_t* dmask = nullptr;
  if (mask) {
    HIP_TRY(h->d_mask.upload(mask, h->B, st));
    dmask = h->d_mask.p;
  }
  if (obs_out && h->d_last_obs.n < (size_t)h->B) HIP_TRY(h->d_last_obs.alloc(h->B));
  if (obs_out && mask) HIP_TRY(hipMemcpyAsync(h->d_last_obs.p, obs_out, sizeof(int32_t) * h->B, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_reset, dim3(grid_for(h->B, 256)), dim3(256), 0, st, h->env(), dmask,
                     obs_out ? h->d_last_obs.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (obs_out) HIP_TRY(hipMemcpyAsync(obs_out, h->d_last_obs.p, sizeof(int32_t) * h->B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (!mask) h->known_reset = true;
  return CMDP_OK;
}

}  // extern "C"
// ---- reference-exact reward caches: the park / fill / relaunch loop (cmdp_reward_cache.h) ------------------------------
// `launch(resume)` enqueues the interaction kernel of the call on the handle's stream (resume 0: every lane starts the
// call's steps; 1: only lanes that parked continue).  Returns when no instance is parked any more; the stream is idle then.
template <typename F>
static int rc_drive(cmdp_t* h, F&& launch) {
  hipStream_t st = h->stream;
  const int B = h->B;
  // refused before anything is stepped: a lane that parks has already committed its transition
  if (!h->rc_streams_set)
    return fail(CMDP_ERR_INVALID, "CMDP_FLAG_REWARD_CACHE: cmdp_set_reward_streams has not been called on this handle");
  HIP_TRY(h->d_rc_park_count.zero(st));
  if (int rc = launch(0)) return rc;
  for (;;) {
    HIP_TRY(hipMemcpyAsync(h->rc_list_h, h->d_rc_park_count.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int count = h->rc_list_h[0];
    if (count == 0) return CMDP_OK;
    const auto t_round = std::chrono::steady_clock::now();
    if (count < 0 || count > B) return fail(CMDP_ERR_HIP, "reward cache: corrupt park count %d", count);
    if (!h->rc_streams_set)
      return fail(CMDP_ERR_INVALID, "CMDP_FLAG_REWARD_CACHE: a Beta reward is needed but cmdp_set_reward_streams was not called");
    HIP_TRY(hipMemcpyAsync(h->rc_list_h, h->d_rc_park_list.p, sizeof(int32_t) * count, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h->rc_pend_h, h->d_rc_pend_e.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->rc_rounds++;
    for (int j0 = 0; j0 < count; j0 += h->rc_cap) {
      const int n = std::min(h->rc_cap, count - j0);
      // one task per parked instance: 5000 draws of its triple's distribution from the instance's own numpy stream
      const std::function<void(int)> fill = [&](int j) {
        const int b = h->rc_list_h[j0 + j];
        const int32_t c = h->h_canon[(size_t)h->rc_pend_h[b]];
        const double pa = h->h_rp0[(size_t)c], pb = h->h_rp1[(size_t)c];
        cmdp_rc::NumpyStream& rs = h->rc_streams[(size_t)b];
        double* out = h->rc_stage_h + (size_t)j * CMDP_RC_BLOCK;
        for (int k = 0; k < CMDP_RC_BLOCK; ++k) out[k] = rs.beta(pa, pb);
      };
      const auto t_fill = std::chrono::steady_clock::now();
      cmdp_rc::Pool::get().parallel_for(n, fill);
      h->rc_fill_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fill).count();
      for (int j = 0; j < n; ++j) {
        const int b = h->rc_list_h[j0 + j];
        const int32_t c = h->h_canon[(size_t)h->rc_pend_h[b]];
        double* blk = h->rc_blk_h[(size_t)c];
        if (!blk) {  // first fill of the triple: a block of its own from the pool (a refill overwrites it in place)
          const size_t chunk = h->rc_next_block / h->rc_chunk_blocks, slot = h->rc_next_block % h->rc_chunk_blocks;
          if (chunk == h->rc_chunks.size()) {
            double* cp = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cp), sizeof(double) * h->rc_chunk_blocks * CMDP_RC_BLOCK));
            h->rc_chunks.push_back(cp);
          }
          blk = h->rc_chunks[chunk] + slot * CMDP_RC_BLOCK;
          h->rc_next_block++;
          h->rc_blk_h[(size_t)c] = blk;
        }
        h->rc_dst_h[j] = blk;
        h->rc_ent_h[j] = c;
      }
      h->rc_fills += n;
      HIP_TRY(hipMemcpyAsync(h->d_rc_stage.p, h->rc_stage_h, sizeof(double) * (size_t)n * CMDP_RC_BLOCK, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(h->d_rc_dst.p, h->rc_dst_h, sizeof(double*) * (size_t)n, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(h->d_rc_ent.p, h->rc_ent_h, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_rc_install, dim3(n), dim3(256), 0, st, n, h->d_rc_stage.p, h->d_rc_dst.p, h->d_rc_ent.p, h->rcache());
      HIP_TRY(hipGetLastError());
      if (j0 + n < count) HIP_TRY(hipStreamSynchronize(st));  // the pinned staging area is reused by the next slice
    }
    if (int rc = launch(1)) return rc;
    h->rc_round_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count();
  }
}

extern "C" {
int cmdp_set_reward_streams(cmdp_t* h, const uint32_t* mt_key, const int32_t* mt_pos, const int32_t* has_gauss,
                            const double* cached_gaussian) {
  if (!h || !mt_key || !mt_pos) return fail(CMDP_ERR_INVALID, "null argument");
  const int B = h->B;
  for (int b = 0; b < B; ++b)
    if (mt_pos[b] < 0 || mt_pos[b] > 624) return fail(CMDP_ERR_INVALID, "instance %d: MT19937 position %d outside [0, 624]", b, mt_pos[b]);
  h->rc_streams.resize((size_t)B);
  for (int b = 0; b < B; ++b) {
    cmdp_rc::NumpyStream& rs = h->rc_streams[(size_t)b];
    std::memcpy(rs.key, mt_key + (size_t)b * 624, sizeof rs.key);
    rs.pos = mt_pos[b];
    rs.out_valid = false;
    rs.has_gauss = has_gauss ? has_gauss[b] : 0;
    rs.gauss = cached_gaussian ? cached_gaussian[b] : 0.0;
  }
  if (h->rc_streams_set && h->reward_cache) {
    // repositioned streams = a new run on a fresh copy of every MDP (BaseMDP.sample_reward starts with empty caches,
    // colosseum/mdp/base.py:1187-1207): the blocks installed on the device, their positions and any parked step are dropped;
    // the pool's chunks are handed out again from the start
    if (int rc = bind(h)) return rc;
    hipStream_t st = h->stream;
    HIP_TRY(h->d_rc_blk.zero(st));
    HIP_TRY(h->d_rc_pos.zero(st));
    HIP_TRY(hipMemsetAsync(h->d_rc_pend_e.p, 0xff, sizeof(int32_t) * (size_t)B, st));
    HIP_TRY(h->d_rc_left.zero(st));
    HIP_TRY(h->d_rc_park_count.zero(st));
    HIP_TRY(hipStreamSynchronize(st));
    std::fill(h->rc_blk_h.begin(), h->rc_blk_h.end(), nullptr);
    h->rc_next_block = 0;
  }
  h->rc_streams_set = true;
  return CMDP_OK;
}

int cmdp_legacy_beta(uint32_t* mt_key, int32_t* mt_pos, int32_t* has_gauss, double* cached_gaussian, double a, double b,
                     int64_t n, double* out) {
  if (!mt_key || !mt_pos || !has_gauss || !cached_gaussian || !out || n < 0) return fail(CMDP_ERR_INVALID, "null argument");
  if (!(a > 0.0 && b > 0.0)) return fail(CMDP_ERR_INVALID, "Beta parameters must be positive");
  if (*mt_pos < 0 || *mt_pos > 624) return fail(CMDP_ERR_INVALID, "MT19937 position outside [0, 624]");
  cmdp_rc::NumpyStream rs;
  std::memcpy(rs.key, mt_key, sizeof rs.key);
  rs.pos = *mt_pos; rs.has_gauss = *has_gauss; rs.gauss = *cached_gaussian;
  // several blocks in parallel would not be the reference's stream: one stream, sequential draws
  for (int64_t i = 0; i < n; ++i) out[i] = rs.beta(a, b);
  std::memcpy(mt_key, rs.key, sizeof rs.key);
  *mt_pos = rs.pos; *has_gauss = rs.has_gauss; *cached_gaussian = rs.gauss;
  return CMDP_OK;
}

static int any_needs_reset(cmdp_t* h, bool* any) {
  hipStream_t st = h->stream;
  HIP_TRY(h->d_flag.zero(st));
  hipLaunchKernelGGL(k_any_needs_reset, dim3(grid_for(h->B, 256)), dim3(256), 0, st, h->d_need_reset.p, h->B,
                     h->d_flag.p);
  HIP_TRY(hipGetLastError());
  int32_t f = 0;
  HIP_TRY(hipMemcpyAsync(&f, h->d_flag.p, sizeof f, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *any = f != 0;
  return CMDP_OK;
}

int cmdp_step(cmdp_t* h, const int32_t* actions, int auto_reset, int32_t* obs, double* reward, uint8_t* step_type) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  if (!actions || !obs || !reward || !step_type) return fail(CMDP_ERR_INVALID, "null argument");
  if (h->layout == CMDP_LAYOUT_DENSE) return fail(CMDP_ERR_UNSUPPORTED, "dense layout: use cmdp_rollout");
  if (int rc = visits_room(h, 1)) return rc;
  hipStream_t st = h->stream;
  const int B = h->B;
  if (h->d_i32_scratch.n < (size_t)2 * B) HIP_TRY(h->d_i32_scratch.alloc((size_t)2 * B));
  if (h->d_f64_scratch.n < (size_t)B) HIP_TRY(h->d_f64_scratch.alloc(B));
  if (h->d_u8_scratch.n < (size_t)B) HIP_TRY(h->d_u8_scratch.alloc(B));
  int32_t* d_act = h->d_i32_scratch.p;
  int32_t* d_obs = h->d_i32_scratch.p + B;
  HIP_TRY(hipMemcpyAsync(d_act, actions, sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
  HIP_TRY(h->d_flag.zero(st));
  if (!auto_reset)
    hipLaunchKernelGGL(k_any_needs_reset, dim3(grid_for(B, 256)), dim3(256), 0, st, h->d_need_reset.p, B, h->d_flag.p);
  hipLaunchKernelGGL(k_check_actions, dim3(grid_for(B, 256)), dim3(256), 0, st, d_act, B, h->A, h->d_flag.p);
  HIP_TRY(hipGetLastError());
  int32_t f = 0;
  HIP_TRY(hipMemcpyAsync(&f, h->d_flag.p, sizeof f, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (f & 2) return fail(CMDP_ERR_INVALID, "action out of range [0, %d)", h->A);
  if (f & 1) return fail(CMDP_ERR_NEEDS_RESET, "step() on an instance that needs reset()");
  h->known_reset = false;  // a step may end an episode (LAST): the async rollout re-checks before its next launch
  if (h->reward_cache) {
    if (int rc = rc_drive(h, [&](int resume) -> int {
          hipLaunchKernelGGL(k_step<true>, dim3(grid_for(B, 256)), dim3(256), 0, st, h->env(), d_act, auto_reset, d_obs,
                             h->d_f64_scratch.p, h->d_u8_scratch.p, h->rcache(), resume);
          HIP_TRY(hipGetLastError());
          return CMDP_OK;
        }))
      return rc;
  } else {
    hipLaunchKernelGGL(k_step<false>, dim3(grid_for(B, 256)), dim3(256), 0, st, h->env(), d_act, auto_reset, d_obs,
                       h->d_f64_scratch.p, h->d_u8_scratch.p, RewardCache{}, 0);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(obs, d_obs, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(reward, h->d_f64_scratch.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(step_type, h->d_u8_scratch.p, B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

static int launch_rollout(cmdp_t* h, int policy, const int8_t* d_actions, int64_t n_steps, double* d_rsum,
                          int32_t* d_last, int32_t* d_tobs, double* d_trew, uint8_t* d_ttype, const float* d_q = nullptr,
                          int resume = 0) {
  hipStream_t st = h->stream;
  const dim3 grid(grid_for(h->B, 256)), block(256);
  const bool trace = d_tobs || d_trew || d_ttype;
  EnvTables t = h->env();
  // K1E accumulates DEPARTURE counts over its launches (cmdp_k1e.h); every other kernel updates the visit counters itself,
  // so the image is folded into them first
  const bool take_k1e = !h->reward_cache && policy == CMDP_POLICY_RANDOM && h->layout != CMDP_LAYOUT_DENSE && h->lds_ok && !trace &&
                        h->k1e_ok && n_steps > 0 && (h->rollout_kernel == 6 || (h->rollout_kernel == 0 && n_steps >= 64));
  if (!take_k1e) { if (int rc = k1e_fold(h)) return rc; }
  if (h->reward_cache) {  // reference-exact reward caches: the lane-per-instance kernel with the park protocol
    const RewardCache rc = h->rcache();
#define ROLL_RC(P, TR) \
  hipLaunchKernelGGL((k_rollout<P, TR, false, true>), grid, block, 0, st, t, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype, d_q, rc, resume)
    if (policy == CMDP_POLICY_RANDOM) { if (trace) ROLL_RC(0, true); else ROLL_RC(0, false); }
    else if (policy == CMDP_POLICY_HOST_ACTIONS) { if (trace) ROLL_RC(1, true); else ROLL_RC(1, false); }
    else { if (trace) ROLL_RC(2, true); else ROLL_RC(2, false); }
#undef ROLL_RC
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  if (policy == CMDP_POLICY_GREEDY_Q) {
    if (h->layout == CMDP_LAYOUT_DENSE) return fail(CMDP_ERR_UNSUPPORTED, "CMDP_POLICY_GREEDY_Q runs on the CSR layout");
    if (trace) hipLaunchKernelGGL((k_rollout<2, true, true>), grid, block, 0, st, t, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype, d_q);
    else hipLaunchKernelGGL((k_rollout<2, false, true>), grid, block, 0, st, t, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype, d_q);
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  if (h->layout == CMDP_LAYOUT_DENSE) {
    if (trace) return fail(CMDP_ERR_UNSUPPORTED, "the dense-layout rollout does not record traces");
    DenseArgs dn{h->d_dense.p, h->dense_spad};
    // two instances per wavefront (software-pipelined: one row in flight while the other is scanned) when the row fits
    // the registers twice; CMDP_K1D_NI = 1 / 2 overrides (tuning aid)
    static const int ni_env = std::getenv("CMDP_K1D_NI") ? std::atoi(std::getenv("CMDP_K1D_NI")) : 0;
    const int nv = h->dense_spad / 256;
    const int ni = ni_env ? ni_env : (nv <= 4 ? 2 : 1);
    const dim3 dgrid(grid_for(h->B, 4 * ni));
#define DENSE_LAUNCH(P, NV, BT, NI) \
  hipLaunchKernelGGL((k_rollout_dense<P, NV, BT, NI>), dgrid, block, 0, st, t, dn, d_actions, n_steps, d_rsum, d_last)
#define DENSE_CASE(NV, NI)                                                                        \
  if (nv == NV && ni == NI) {                                                                     \
    if (policy == CMDP_POLICY_RANDOM) { if (h->sample_beta) DENSE_LAUNCH(0, NV, true, NI); else DENSE_LAUNCH(0, NV, false, NI); } \
    else { if (h->sample_beta) DENSE_LAUNCH(1, NV, true, NI); else DENSE_LAUNCH(1, NV, false, NI); }      \
  } else
    DENSE_CASE(1, 1) DENSE_CASE(2, 1) DENSE_CASE(3, 1) DENSE_CASE(4, 1) DENSE_CASE(6, 1) DENSE_CASE(8, 1) DENSE_CASE(12, 1)
    DENSE_CASE(16, 1) DENSE_CASE(1, 2) DENSE_CASE(2, 2) DENSE_CASE(3, 2) DENSE_CASE(4, 2)
    { return fail(CMDP_ERR_UNSUPPORTED, "dense layout: no kernel for a row stride of %d floats, %d instances per wavefront", h->dense_spad, ni); }
#undef DENSE_CASE
#undef DENSE_LAUNCH
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  const bool lds_eligible = h->lds_ok && policy == CMDP_POLICY_RANDOM && !trace;
  if (h->rollout_kernel == 2 && !lds_eligible)
    return fail(CMDP_ERR_UNSUPPORTED, "LDS-resident rollout needs deterministic dynamics, one start state, <= 65535 "
                                      "states, <= 256 distinct rewards, the random policy and no trace");
  // the LDS kernel pays a fixed staging + flush cost per launch: worth it from a few dozen transitions on
  if (h->rollout_kernel == 4 && !(lds_eligible && h->tmpl_ok))
    return fail(CMDP_ERR_UNSUPPORTED, "the shared-table rollout K1T needs a batch eligible for K1P with two actions whose instances "
                                      "are per-state action permutations of the first one, the random policy and no trace");
  if (h->rollout_kernel == 5 && !(lds_eligible && h->k1u_ok))
    return fail(CMDP_ERR_UNSUPPORTED, "the streamed-trace rollout K1U needs a batch eligible for the shared-table rollout K1T (CMDP_OPT_ROLLOUT_KERNEL 4) "
                                      "and room for 64 instances per workgroup");
  if (h->rollout_kernel == 6 && !(lds_eligible && h->k1e_ok))
    return fail(CMDP_ERR_UNSUPPORTED, "the episode-parallel rollout K1E needs a batch eligible for the LDS-resident kernels (CMDP_OPT_ROLLOUT_KERNEL 2) "
                                      "that is episodic, has two actions, at most four distinct reward values, at most 512 states per instance and a horizon whose action bits for 128 episodes fit LDS");
  if (take_k1e) {
    // K1E: per segment of <= K1E_SEG transitions (16-bit counts in the table dwords; the code buffer) the walk kernel, then
    // the reward scan over the code words it left in HBM
    if (int rc = k1u_join(h)) return rc;
    if (!h->ev_k1u[0])
      for (int i = 0; i < 5; ++i) HIP_TRY(hipEventCreateWithFlags(&h->ev_k1u[i], hipEventDisableSystemFence));   // (timestamps only: no host-visibility cache flush per launch)
    K1ePlan e = h->k1e;
    if (!h->d_k1e_dep.p) {
      const size_t nd = (size_t)grid_for(h->B, K1E_NI) * (size_t)e.gdw;
      HIP_TRY(h->d_k1e_dep.alloc(nd));
      HIP_TRY(h->d_k1e_dep.zero(st));
      HIP_TRY(h->d_k1e_dep_res.alloc(h->B));
      HIP_TRY(h->d_k1e_dep_res.zero(st));
      HIP_TRY(h->d_vis_ovf.alloc(1));
      HIP_TRY(h->d_vis_ovf.zero(st));
    }
    if (h->k1e_pending_steps + n_steps > 0x7fff0000LL) { if (int rc = k1e_fold(h)) return rc; }   // the departure image is int32
    e.dep = h->d_k1e_dep.p;
    e.dep_res = h->d_k1e_dep_res.p;
    // The reward scan of a segment runs on a second stream under the walk of the next segment / launch (two sets of code
    // buffers): it is one wavefront per SIMD of sequential sums, the walk fills the rest of the chip (CMDP_K1E_OVERLAP=0: one
    // stream).  cmdp_rollout / cmdp_synchronize / every call that reads the sums waits for it.
    static const int ov_env = std::getenv("CMDP_K1E_OVERLAP") ? std::atoi(std::getenv("CMDP_K1E_OVERLAP")) : 1;
    const bool ov = ov_env != 0;
    if (ov && !h->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    if (ov && !h->ev_k1e_walk[0])
      for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_k1e_walk[i], hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_k1e_scan[i], hipEventDisableTiming | hipEventDisableSystemFence));
      }
    if (!ov) { if (int rc = k1e_scan_join(h)) return rc; }
    // segment length: the code words of a segment (12 bytes per episode chunk and instance, two sets) stay within ~1.5 GB
    const int64_t budget_words = std::max<int64_t>(4, (int64_t)((768ll << 20) / (12 * (int64_t)h->B * e.nch)));
    const int64_t seg = std::max<int64_t>(e.H, std::min<int64_t>(K1E_SEG, (budget_words - 2) * e.H));
    const int64_t epi_cap = k1e_max_episodes(std::min<int64_t>(n_steps, seg), e.H);
    const size_t need = (size_t)epi_cap * (size_t)e.nch * (size_t)h->B;
    if (h->d_k1e_h0.n < (size_t)h->B) HIP_TRY(h->d_k1e_h0.alloc(h->B));
    if (h->d_k1e_h0b.n < (size_t)h->B) HIP_TRY(h->d_k1e_h0b.alloc(h->B));
    if (int rc = set_lds(k_rollout_epi<true>, h->k1e_lds)) return rc;
    if (int rc = set_lds(k_rollout_epi<false>, h->k1e_lds)) return rc;
    for (int64_t s0 = 0; s0 < n_steps; s0 += seg) {
      const int64_t n = std::min<int64_t>(seg, n_steps - s0);
      const int i = ov ? (int)(h->k1e_seq & 1) : 0;
      if (h->d_k1e_codes[i].n < need || h->d_k1e_cnts[i].n < need) {
        if (h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));   // a scan may still read the buffers
        if (h->d_k1e_codes[i].alloc(need) != hipSuccess || h->d_k1e_cnts[i].alloc(need) != hipSuccess) {
          // no room for the code words (12 bytes per episode and instance): the chain kernels need no workspace -- this
          // handle takes them from now on (possible before the first segment only: the buffers never shrink)
          (void)hipGetLastError();
          if (s0 > 0) return fail(CMDP_ERR_HIP, "K1E: out of device memory for the code words of a later segment");
          h->d_k1e_codes[i].release();
          h->d_k1e_cnts[i].release();
          h->k1e_ok = false;
          if (h->rollout_kernel == 6) return fail(CMDP_ERR_HIP, "K1E: out of device memory for %zu code words", need);
          return launch_rollout(h, policy, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype, d_q, resume);
        }
      }
      e.codes = h->d_k1e_codes[i].p;
      e.cnts = h->d_k1e_cnts[i].p;
      e.seg_h0 = i ? h->d_k1e_h0b.p : h->d_k1e_h0.p;
      e.n_pass = (int)((k1e_max_episodes(n, e.H) + K1E_EPP - 1) / K1E_EPP);
      if (ov && h->ev_k1e_scan_used[i]) HIP_TRY(hipStreamWaitEvent(st, h->ev_k1e_scan[i], 0));   // its last scan has read this set
      const bool last = s0 + seg >= n_steps;
      if (last) HIP_TRY(hipEventRecord(h->ev_k1u[0], st));
      // one workgroup per CU (the tables take the CU's LDS), each walking its groups one after the other with the next
      // group's table image in flight under the walk; CMDP_K1E_GRID = workgroups (timing experiments; any value is correct)
      int k1e_grid = std::min(grid_for(h->B, K1E_NI), h->cus);
      if (const char* gs = std::getenv("CMDP_K1E_GRID")) k1e_grid = std::max(1, std::min(grid_for(h->B, K1E_NI), std::atoi(gs)));
      if (e.n_codes <= 3) hipLaunchKernelGGL(k_rollout_epi<true>, dim3(k1e_grid), dim3(K1E_THREADS), h->k1e_lds, st, t, e, (int)n, d_last);
      else hipLaunchKernelGGL(k_rollout_epi<false>, dim3(k1e_grid), dim3(K1E_THREADS), h->k1e_lds, st, t, e, (int)n, d_last);
      if (last) HIP_TRY(hipEventRecord(h->ev_k1u[1], st));
      if (ov) {
        HIP_TRY(hipEventRecord(h->ev_k1e_walk[i], st));
        HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_k1e_walk[i], 0));
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[3], h->aux_stream));
        hipLaunchKernelGGL(k_reward_scan, dim3(grid_for(h->B, K1R_THREADS)), dim3(K1R_THREADS), 0, h->aux_stream, t, e, n, d_rsum, s0 > 0 ? 1 : 0);
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[4], h->aux_stream));
        HIP_TRY(hipEventRecord(h->ev_k1e_scan[i], h->aux_stream));
        h->ev_k1e_scan_used[i] = true;
        h->k1e_scan_pending = true;
        h->k1e_seq++;
      } else {
        hipLaunchKernelGGL(k_reward_scan, dim3(grid_for(h->B, K1R_THREADS)), dim3(K1R_THREADS), 0, st, t, e, n, d_rsum, s0 > 0 ? 1 : 0);
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[2], st));
      }
    }
    h->k1u_last_overlap = ov;
    h->k1e_pending = true;
    h->k1e_pending_steps += n_steps;
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  const bool take_k1u = lds_eligible && h->k1u_ok && (h->rollout_kernel == 5 || (h->rollout_kernel == 0 && h->k1u_auto && n_steps >= 64));
  if (!take_k1u) {
    if (int rc = k1u_join(h)) return rc;   // every other kernel updates the visit counters itself
  } else {
    // K1U: per segment of <= K1U_SEG transitions the chain kernel (trace -> HBM), then the histogram of that trace -- on a
    // second stream, under the chain kernel of the next segment / launch (two trace buffers), unless switched off
    static const int ov_env = std::getenv("CMDP_K1U_OVERLAP") ? std::atoi(std::getenv("CMDP_K1U_OVERLAP")) : -1;
    const bool ov = (ov_env < 0 ? h->k1u_overlap : ov_env != 0) && h->k1u_overlap_fits;
    if (!h->ev_k1u[0])
      for (int i = 0; i < 5; ++i) HIP_TRY(hipEventCreateWithFlags(&h->ev_k1u[i], hipEventDisableSystemFence));   // (timestamps only: no host-visibility cache flush per launch)
    if (ov && !h->aux_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_trace[i], hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_hist[i], hipEventDisableTiming | hipEventDisableSystemFence));
      }
    }
    if (!ov) { if (int rc = k1u_join(h)) return rc; }
    const int64_t seg_max = std::min<int64_t>(n_steps, K1U_SEG);
    K1uPlan u = h->k1u;
    const int epp = K1U_EPP(u.pack10);
    const size_t need = (size_t)((seg_max + epp - 1) / epp) * (size_t)h->B;
    const size_t hist_lds = k1h_lds_bytes(h->max_S, ov ? 32 : 64);
#define K1U_SET_LDS(P10)                                                                             \
    {                                                                                                \
      if (int rc = set_lds(k_rollout_tmpl_stream<P10>, h->k1u_lds)) return rc;                       \
      if (ov) { if (int rc = set_lds(k_trace_hist<32, 512, P10>, hist_lds)) return rc; }             \
      else { if (int rc = set_lds(k_trace_hist<64, 1024, P10>, hist_lds)) return rc; }               \
    }
    if (u.pack10) K1U_SET_LDS(true) else K1U_SET_LDS(false)
#undef K1U_SET_LDS
    for (int64_t s0 = 0; s0 < n_steps; s0 += K1U_SEG) {
      const int64_t n = std::min<int64_t>(K1U_SEG, n_steps - s0);
      const int i = ov ? (int)(h->k1u_seq & 1) : 0;
      if (h->d_k1u_trace[i].n < need) {
        if (ov && h->aux_stream) HIP_TRY(hipStreamSynchronize(h->aux_stream));   // the buffer may still be read
        if (h->d_k1u_trace[i].alloc(need) != hipSuccess) {
          // no room for the trace (16 bytes per 8-12 transitions and instance): K1T counts in LDS and needs none -- this
          // handle takes it from now on (possible before the first segment only: the buffer never shrinks)
          (void)hipGetLastError();
          if (s0 > 0) return fail(CMDP_ERR_HIP, "K1U: out of device memory for the trace of a later segment");
          h->d_k1u_trace[i].release();
          h->k1u_ok = false;
          if (h->rollout_kernel == 5) return fail(CMDP_ERR_HIP, "K1U: out of device memory for a trace of %zu pieces", need);
          return launch_rollout(h, policy, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype, d_q, resume);
        }
      }
      if (h->d_k1u_resets[i].n < (size_t)h->B) HIP_TRY(h->d_k1u_resets[i].alloc(h->B));
      u.trace = h->d_k1u_trace[i].p;
      u.seg_resets = h->d_k1u_resets[i].p;
      if (ov && h->ev_hist_used[i]) HIP_TRY(hipStreamWaitEvent(st, h->ev_hist[i], 0));   // its last histogram has read the buffer
      const bool last = s0 + K1U_SEG >= n_steps;
      if (last) HIP_TRY(hipEventRecord(h->ev_k1u[0], st));
      const dim3 rgrid(grid_for(h->B, u.G)), rblock(K1U_THREADS);
      if (u.pack10) hipLaunchKernelGGL(k_rollout_tmpl_stream<true>, rgrid, rblock, h->k1u_lds, st, t, u, n, d_rsum, d_last, s0 > 0 ? 1 : 0);
      else hipLaunchKernelGGL(k_rollout_tmpl_stream<false>, rgrid, rblock, h->k1u_lds, st, t, u, n, d_rsum, d_last, s0 > 0 ? 1 : 0);
      if (last) HIP_TRY(hipEventRecord(h->ev_k1u[1], st));
      if (ov) {
        HIP_TRY(hipEventRecord(h->ev_trace[i], st));
        HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_trace[i], 0));
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[3], h->aux_stream));
        const dim3 hgrid(grid_for(h->B, 32)), hblock(512);
        if (u.pack10) hipLaunchKernelGGL((k_trace_hist<32, 512, true>), hgrid, hblock, hist_lds, h->aux_stream, t, u.trace, u.seg_resets, n, u.code_shift);
        else hipLaunchKernelGGL((k_trace_hist<32, 512, false>), hgrid, hblock, hist_lds, h->aux_stream, t, u.trace, u.seg_resets, n, u.code_shift);
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[4], h->aux_stream));
        HIP_TRY(hipEventRecord(h->ev_hist[i], h->aux_stream));
        h->ev_hist_used[i] = true;
        h->k1u_pending = true;
        h->k1u_seq++;
      } else {
        const dim3 hgrid(grid_for(h->B, 64)), hblock(1024);
        if (u.pack10) hipLaunchKernelGGL((k_trace_hist<64, 1024, true>), hgrid, hblock, hist_lds, st, t, u.trace, u.seg_resets, n, u.code_shift);
        else hipLaunchKernelGGL((k_trace_hist<64, 1024, false>), hgrid, hblock, hist_lds, st, t, u.trace, u.seg_resets, n, u.code_shift);
        if (last) HIP_TRY(hipEventRecord(h->ev_k1u[2], st));
      }
    }
    h->k1u_last_overlap = ov;
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  if (lds_eligible && h->tmpl_ok && (h->rollout_kernel == 4 || (h->rollout_kernel == 0 && h->tmpl_auto && n_steps >= 64))) {
    if (int rc = set_lds(k_rollout_tmpl, h->tmpl_lds)) return rc;
    hipLaunchKernelGGL(k_rollout_tmpl, dim3(grid_for(h->B, h->tmpl_plan.G)), dim3(K1T_THREADS), h->tmpl_lds, st, t, h->tmpl_plan,
                       n_steps, d_rsum, d_last);
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  if (lds_eligible && (h->rollout_kernel == 2 || (h->rollout_kernel == 0 && n_steps >= 64))) {
    const dim3 lgrid(grid_for(h->B, h->lds_plan.G)), lblock(K1L_THREADS);
    if (h->lds_plan.pipe) {
      if (int rc = set_lds(k_rollout_pipe, h->lds_bytes)) return rc;
      hipLaunchKernelGGL(k_rollout_pipe, lgrid, dim3(K1P_THREADS), h->lds_bytes, st, t, h->lds_plan, n_steps, d_rsum, d_last);
    } else if (h->lds_plan.code_shift) {
      if (int rc = set_lds(k_rollout_lds<true>, h->lds_bytes)) return rc;
      hipLaunchKernelGGL(k_rollout_lds<true>, lgrid, lblock, h->lds_bytes, st, t, h->lds_plan, n_steps, d_rsum, d_last);
    } else {
      if (int rc = set_lds(k_rollout_lds<false>, h->lds_bytes)) return rc;
      hipLaunchKernelGGL(k_rollout_lds<false>, lgrid, lblock, h->lds_bytes, st, t, h->lds_plan, n_steps, d_rsum, d_last);
    }
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
  const bool k1s_eligible = h->k1s_ok && policy == CMDP_POLICY_RANDOM && !trace;
  if (h->rollout_kernel == 3 && !k1s_eligible)
    return fail(CMDP_ERR_UNSUPPORTED, "the LDS-resident stochastic rollout K1S needs Philox mode, the random policy, no trace, equal "
                                      "state counts, <= 16 entries per row and <= 16 distinct successors per state, <= 64 "
                                      "cumulative-probability patterns, deterministic rewards that depend on the successor or on "
                                      "the row alone, and room for four instances in LDS");
  // K1S runs G instances per CU at a time however large the batch is (LDS capacity), K1's rate grows with the batch (more
  // wavefronts cover its HBM latency) until bandwidth caps it: K1 ~ B x 2.5e5 / (1 + B / 40 000) transitions/s (measured,
  // profiles/r02_k1_vs_k1s_batch.json).  Round 3 (row-shape dictionary: 2-3 x the instances per CU; four walker
  // wavefronts; specialised, hand-pipelined walker): K1S ~ CUs x G / 320-470 ns -- FrozenLake-20 1.44e10 at 4 096 instances
  // and 1.85e10 at 131 072 (K1: 0.8e9 / 6.6e9), so K1S stays ahead at every batch size measured
  // (profiles/r03_k1s_walkers.txt); the model is kept for batches where few instances fit a CU.
  const double k1_rate = (double)h->B * 2.5e5 / (1.0 + (double)h->B / 4.0e4);
  const double k1s_rate = (double)h->cus * (double)h->k1s.G / 450e-9;
  const bool k1s_pays = k1_rate < 1.1 * k1s_rate;
  if (k1s_eligible && (h->rollout_kernel == 3 || (h->rollout_kernel == 0 && n_steps >= 64 && k1s_pays))) {
    if (int rc = set_lds(k_rollout_stoch, h->k1s_bytes)) return rc;
    hipLaunchKernelGGL(k_rollout_stoch, dim3(grid_for(h->B, h->k1s.G)), dim3(K1S_THREADS), h->k1s_bytes, st, t, h->k1s, n_steps,
                       d_rsum, d_last);
    HIP_TRY(hipGetLastError());
    return CMDP_OK;
  }
#define ROLL(P, TR, BT) \
  hipLaunchKernelGGL((k_rollout<P, TR, BT>), grid, block, 0, st, t, d_actions, n_steps, d_rsum, d_last, d_tobs, d_trew, d_ttype)
  const bool bt = h->sample_beta;
  if (policy == CMDP_POLICY_RANDOM) {
    if (trace) { if (bt) ROLL(0, true, true); else ROLL(0, true, false); }
    else { if (bt) ROLL(0, false, true); else ROLL(0, false, false); }
  } else {
    if (trace) { if (bt) ROLL(1, true, true); else ROLL(1, true, false); }
    else { if (bt) ROLL(1, false, true); else ROLL(1, false, false); }
  }
#undef ROLL
  HIP_TRY(hipGetLastError());
  return CMDP_OK;
}

int cmdp_rollout(cmdp_t* h, int policy, const void* policy_arg, int64_t n_steps, int32_t* last_obs, double* reward_sum,
                 int32_t* trace_obs, double* trace_reward, uint8_t* trace_type) {
  if (int rc = bind(h, false)) return rc;   // (launch_rollout joins what the kernel it takes needs joined)
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  if (n_steps < 0) return fail(CMDP_ERR_INVALID, "n_steps < 0");
  if (policy != CMDP_POLICY_RANDOM && policy != CMDP_POLICY_HOST_ACTIONS && policy != CMDP_POLICY_GREEDY_Q)
    return fail(CMDP_ERR_INVALID, "policy");
  if (policy != CMDP_POLICY_RANDOM && !policy_arg && n_steps > 0) return fail(CMDP_ERR_INVALID, "policy_arg missing");
  if (int rc = visits_room(h, n_steps)) return rc;
  bool any = false;
  if (int rc = any_needs_reset(h, &any)) return rc;
  if (any) return fail(CMDP_ERR_NEEDS_RESET, "rollout() on an instance that needs reset()");
  h->known_reset = true;  // the fused loop resets at once after every terminating step
  hipStream_t st = h->stream;
  const int B = h->B;
  const size_t NB = (size_t)n_steps * B;
  const int8_t* d_act = nullptr;
  if (policy == CMDP_POLICY_HOST_ACTIONS) {
    const int8_t* a = static_cast<const int8_t*>(policy_arg);
    for (size_t i = 0; i < NB; ++i)
      if (a[i] < 0 || a[i] >= h->A) return fail(CMDP_ERR_INVALID, "action out of range [0, %d)", h->A);
    HIP_TRY(h->d_actions8.upload(a, NB, st));
    d_act = h->d_actions8.p;
  }
  const float* d_q = nullptr;
  if (policy == CMDP_POLICY_GREEDY_Q) {
    HIP_TRY(h->d_gp_q.upload(static_cast<const float*>(policy_arg), (size_t)(h->H > 0 ? h->H : 1) * h->n_rows, st));
    d_q = h->d_gp_q.p;
  }
  if (h->d_rsum.n < (size_t)B) HIP_TRY(h->d_rsum.alloc(B));
  if (h->d_last_obs.n < (size_t)B) HIP_TRY(h->d_last_obs.alloc(B));
  if (trace_obs && h->d_tr_obs.n < NB) HIP_TRY(h->d_tr_obs.alloc(NB));
  if (trace_reward && h->d_tr_rew.n < NB) HIP_TRY(h->d_tr_rew.alloc(NB));
  if (trace_type && h->d_tr_type.n < NB) HIP_TRY(h->d_tr_type.alloc(NB));
  auto launch = [&](int resume) -> int {
    return launch_rollout(h, policy, d_act, n_steps, h->d_rsum.p, h->d_last_obs.p, trace_obs ? h->d_tr_obs.p : nullptr,
                          trace_reward ? h->d_tr_rew.p : nullptr, trace_type ? h->d_tr_type.p : nullptr, d_q, resume);
  };
  if (h->reward_cache) { if (int rc = rc_drive(h, launch)) return rc; }
  else if (int rc = launch(0)) return rc;
  if (int rc = k1e_scan_join(h)) return rc;   // (K1E: the sums of the last segment come from the second stream)
  if (last_obs) HIP_TRY(hipMemcpyAsync(last_obs, h->d_last_obs.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  if (reward_sum) HIP_TRY(hipMemcpyAsync(reward_sum, h->d_rsum.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
  if (trace_obs) HIP_TRY(hipMemcpyAsync(trace_obs, h->d_tr_obs.p, sizeof(int32_t) * NB, hipMemcpyDeviceToHost, st));
  if (trace_reward) HIP_TRY(hipMemcpyAsync(trace_reward, h->d_tr_rew.p, sizeof(double) * NB, hipMemcpyDeviceToHost, st));
  if (trace_type) HIP_TRY(hipMemcpyAsync(trace_type, h->d_tr_type.p, NB, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_rollout_async(cmdp_t* h, int policy, int64_t n_steps) {
  if (int rc = bind(h, false)) return rc;   // K1U's histogram of the previous launch may still run (launch_rollout joins otherwise)
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  if (policy != CMDP_POLICY_RANDOM) return fail(CMDP_ERR_INVALID, "rollout_async supports CMDP_POLICY_RANDOM only");
  if (n_steps < 0) return fail(CMDP_ERR_INVALID, "n_steps < 0");
  if (int rc = visits_room(h, n_steps)) return rc;
  if (!h->known_reset) {  // one 4-byte read-back on the first call after create / cmdp_step, none afterwards
    bool any = false;
    if (int rc = any_needs_reset(h, &any)) return rc;
    if (any) return fail(CMDP_ERR_NEEDS_RESET, "rollout_async() on an instance that needs reset()");
    h->known_reset = true;
  }
  if (h->d_rsum.n < (size_t)h->B) HIP_TRY(h->d_rsum.alloc(h->B));
  if (h->d_last_obs.n < (size_t)h->B) HIP_TRY(h->d_last_obs.alloc(h->B));
  if (h->reward_cache)  // the park / fill / relaunch loop needs the host: synchronous in this mode
    return rc_drive(h, [&](int resume) -> int {
      return launch_rollout(h, policy, nullptr, n_steps, h->d_rsum.p, h->d_last_obs.p, nullptr, nullptr, nullptr, nullptr, resume);
    });
  return launch_rollout(h, policy, nullptr, n_steps, h->d_rsum.p, h->d_last_obs.p, nullptr, nullptr, nullptr);
}

int cmdp_set_option(cmdp_t* h, int option, int64_t value) {
  if (!h) return fail(CMDP_ERR_INVALID, "null handle");
  if (option == CMDP_OPT_ROLLOUT_KERNEL && value >= 0 && value <= 6) {
    h->rollout_kernel = (int)value;
    return CMDP_OK;
  }
  if (option == CMDP_OPT_LDS_GROUPS_PER_CU && (value == 1 || value == 2)) {
    if (!h->lds_ok) return fail(CMDP_ERR_INVALID, "the batch is not eligible for the LDS-resident rollout kernels");
    if (h->lds_plan.pipe)
      return fail(CMDP_ERR_UNSUPPORTED, "this handle runs the pipeline kernel K1P, one workgroup per CU by construction "
                  "(its successor table is encoded for K1P at cmdp_create: set CMDP_K1L_PIPE=0 before creating the handle to use K1L)");
    const int cap = value == 1 ? h->lds_G1 : h->lds_G2;
    if (cap < 1) return fail(CMDP_ERR_INVALID, "no room for %lld workgroups per CU", (long long)value);
    // the fewest instances per workgroup that keep the round count at this many groups per CU (as cmdp_create does)
    const int64_t slots = (int64_t)h->cus * value, wgs = (h->B + cap - 1) / cap, rounds = (wgs + slots - 1) / slots;
    const int g = (int)std::min<int64_t>(cap, std::max<int64_t>(1, (h->B + rounds * slots - 1) / (rounds * slots)));
    h->lds_plan.G = g;
    h->lds_bytes = k1l_lds_bytes(h->lds_plan, g);
    return CMDP_OK;
  }
  if (option == CMDP_OPT_DP_KERNEL && value >= 0 && value <= 7) {
    h->dp_kernel = (int)value;
    return CMDP_OK;
  }
  if (option == CMDP_OPT_CHAIN_EXACT_ORDER && (value == 0 || value == 1)) {
    h->chain_exact = value == 1;
    return CMDP_OK;
  }
  if (option == CMDP_OPT_MIXING_PATH && value >= 0 && value <= 2) {
    h->mixing_path = (int)value;
    return CMDP_OK;
  }
  if (option == CMDP_OPT_DIAMETER_WORKSPACE_MB && value >= 1) {
    h->dl_ws_bytes = (size_t)value << 20;
    return CMDP_OK;
  }
  if (option == CMDP_OPT_DIAMETER_RELABEL_MIN_STATES && value >= 0) {
    h->relabel_min_states = value;
    h->ell_K = 0;  // the fixed-width rows are rebuilt by the next K5S launch
    return CMDP_OK;
  }
  return fail(CMDP_ERR_INVALID, "unknown option %d / value %lld", option, (long long)value);
}

int cmdp_lds_plan(cmdp_t* h, int32_t plan[4]) {
  if (!h || !plan) return fail(CMDP_ERR_INVALID, "bad argument");
  plan[0] = (h->lds_ok || h->k1s_ok) ? 1 : 0;
  const bool k1e = h->lds_ok && h->k1e_ok && (h->rollout_kernel == 6 || h->rollout_kernel == 0);
  const bool k1u = !k1e && h->lds_ok && h->k1u_ok && (h->rollout_kernel == 5 || (h->rollout_kernel == 0 && h->k1u_auto));
  const bool k1t = !k1e && !k1u && h->lds_ok && h->tmpl_ok && (h->rollout_kernel == 4 || (h->rollout_kernel == 0 && h->tmpl_auto));   // what a launch takes
  plan[1] = k1e ? 5 : k1u ? 4 : k1t ? 3 : (h->lds_ok ? h->lds_plan.pipe : (h->k1s_ok ? 2 : 0));
  plan[2] = k1e ? K1E_NI : k1u ? h->k1u.G : k1t ? h->tmpl_plan.G : (h->lds_ok ? h->lds_plan.G : (h->k1s_ok ? h->k1s.G : 0));
  plan[3] = k1e ? K1E_EPP : k1u ? h->k1u.ch : k1t ? h->tmpl_plan.ch : (h->lds_ok ? h->lds_plan.ch : (h->k1s_ok ? h->k1s.ch : 0));
  return CMDP_OK;
}

int cmdp_synchronize(cmdp_t* h) {
  if (int rc = bind(h, false)) return rc;   // (the departure image of K1E stays as it is: nothing here reads the counters)
  if (int rc = k1u_join(h)) return rc;
  if (int rc = k1e_scan_join(h)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CMDP_OK;
}

int cmdp_stat(cmdp_t* h, int which, double* out) {
  if (int rc = bind(h)) return rc;
  if (!out) return fail(CMDP_ERR_INVALID, "null output");
  if (which == CMDP_STAT_DP_KERNEL_MS) {
    if (!h->ev_dp0) return fail(CMDP_ERR_INVALID, "no discounted solve has run on this handle");
    HIP_TRY(hipEventSynchronize(h->ev_dp1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_dp0, h->ev_dp1));
    *out = ms;
    return CMDP_OK;
  }
  if (which == CMDP_STAT_DP_KERNEL) {
    *out = h->last_dp_kernel;
    return CMDP_OK;
  }
  if (which == CMDP_STAT_ROLLOUT_KERNEL_MS || which == CMDP_STAT_HIST_KERNEL_MS) {
    if (!h->ev_k1u[0]) return fail(CMDP_ERR_INVALID, "no streamed-trace rollout (K1U) has run on this handle");
    const bool hist = which == CMDP_STAT_HIST_KERNEL_MS;
    hipEvent_t e0 = hist ? (h->k1u_last_overlap ? h->ev_k1u[3] : h->ev_k1u[1]) : h->ev_k1u[0];
    hipEvent_t e1 = hist ? (h->k1u_last_overlap ? h->ev_k1u[4] : h->ev_k1u[2]) : h->ev_k1u[1];
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *out = ms;
    return CMDP_OK;
  }
  if (which == CMDP_STAT_CHAIN_FAST_INSTANCES) {
    if (!h->chain_plan_any) { *out = 0.0; return CMDP_OK; }
    std::vector<uint8_t> slow((size_t)h->B);
    HIP_TRY(hipMemcpyAsync(slow.data(), h->d_cf_slow.p, (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int n = 0;
    for (uint8_t x : slow) n += x == 0;
    *out = (double)n;
    return CMDP_OK;
  }
  if (which == CMDP_STAT_REWARD_FILLS || which == CMDP_STAT_REWARD_ROUNDS) {
    *out = (double)(which == CMDP_STAT_REWARD_FILLS ? h->rc_fills : h->rc_rounds);
    return CMDP_OK;
  }
  if (which == CMDP_STAT_DIAMETER_CLUSTER_LAUNCHES || which == CMDP_STAT_DIAMETER_CLUSTER_FALLBACKS) {
    *out = (double)(which == CMDP_STAT_DIAMETER_CLUSTER_LAUNCHES ? h->k5c_launches : h->k5c_timeouts);
    return CMDP_OK;
  }
  if (which == CMDP_STAT_REWARD_FILL_MS || which == CMDP_STAT_REWARD_ROUND_MS) {
    *out = which == CMDP_STAT_REWARD_FILL_MS ? h->rc_fill_ms : h->rc_round_ms;
    return CMDP_OK;
  }
  return fail(CMDP_ERR_INVALID, "unknown statistic %d", which);
}

int cmdp_calibrate(int what, int64_t n_steps, double* ns_per_step) {
  if (!ns_per_step || n_steps < 1 || n_steps > 10000000) return fail(CMDP_ERR_INVALID, "bad argument");
  if (what != CMDP_CALIB_LDS_READ && what != CMDP_CALIB_LDS_CHAIN && what != CMDP_CALIB_LDS_CHAIN_SHARED)
    return fail(CMDP_ERR_INVALID, "unknown calibration %d", what);
  int dev = 0, cus = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  DevBuf<int32_t> sink;
  HIP_TRY(sink.alloc((size_t)cus * 64));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const size_t lds = 64 * 1024;
  for (int rep = 0; rep < 2; ++rep) {  // the first launch warms the code object up
    const int n = rep == 0 ? 1000 : (int)n_steps;
    HIP_TRY(hipEventRecord(e0, nullptr));
    if (what == CMDP_CALIB_LDS_READ) hipLaunchKernelGGL(k_calib_lds_chain<0>, dim3(cus), dim3(64), lds, nullptr, n, 30, sink.p);
    else if (what == CMDP_CALIB_LDS_CHAIN) hipLaunchKernelGGL(k_calib_lds_chain<1>, dim3(cus), dim3(64), lds, nullptr, n, 30, sink.p);
    else hipLaunchKernelGGL(k_calib_lds_chain<2>, dim3(cus), dim3(64), lds, nullptr, n, 30, sink.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
  }
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ns_per_step = (double)ms * 1e6 / (double)n_steps;
  return CMDP_OK;
}

int cmdp_visits(cmdp_t* h, int64_t* state_counts, int64_t* sa_counts) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  hipStream_t st = h->stream;
  std::vector<int32_t> tmp;
  if (state_counts) {
    tmp.resize((size_t)h->n_states);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->d_visits_s.p, sizeof(int32_t) * tmp.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < tmp.size(); ++i) state_counts[i] = tmp[i];
  }
  if (sa_counts) {
    tmp.resize((size_t)h->n_rows);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->d_visits_sa.p, sizeof(int32_t) * tmp.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < tmp.size(); ++i) sa_counts[i] = tmp[i];
  }
  return CMDP_OK;
}

int cmdp_reset_visits(cmdp_t* h) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  HIP_TRY(h->d_visits_s.zero(h->stream));
  HIP_TRY(h->d_visits_sa.zero(h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->vis_bound = 0;
  return CMDP_OK;
}

int cmdp_set_visits(cmdp_t* h, const int64_t* state_counts, const int64_t* sa_counts) {
  if (int rc = bind(h)) return rc;   // (folds what K1E still holds: the restored values replace everything)
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  hipStream_t st = h->stream;
  int64_t top = 0;
  std::vector<int32_t> tmp;
  for (int which = 0; which < 2; ++which) {
    const int64_t* src = which ? sa_counts : state_counts;
    if (!src) continue;
    const size_t n = (size_t)(which ? h->n_rows : h->n_states);
    tmp.resize(n);
    for (size_t i = 0; i < n; ++i) {
      if (src[i] < 0 || src[i] > 0x7fffffffLL) return fail(CMDP_ERR_INVALID, "visit count %lld at index %zu does not fit the device's int32 counters", (long long)src[i], i);
      tmp[i] = (int32_t)src[i];
      top = std::max(top, src[i]);
    }
    HIP_TRY(hipMemcpyAsync(which ? h->d_visits_sa.p : h->d_visits_s.p, tmp.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // a counter left as it was keeps its old bound
  h->vis_bound = (state_counts && sa_counts) ? top : std::max(h->vis_bound, top);
  return CMDP_OK;
}

int cmdp_last_start(cmdp_t* h, int32_t* last_start, int32_t* previous_start) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env || !last_start) return fail(CMDP_ERR_INVALID, "bad argument");
  HIP_TRY(hipMemcpyAsync(last_start, h->d_last_start.p, sizeof(int32_t) * h->B, hipMemcpyDeviceToHost, h->stream));
  if (previous_start)
    HIP_TRY(hipMemcpyAsync(previous_start, h->d_prev_start.p, sizeof(int32_t) * h->B, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CMDP_OK;
}

int cmdp_state(cmdp_t* h, int32_t* cur, int32_t* hstep, uint8_t* needs_reset) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  hipStream_t st = h->stream;
  if (cur) HIP_TRY(hipMemcpyAsync(cur, h->d_cur.p, sizeof(int32_t) * h->B, hipMemcpyDeviceToHost, st));
  if (hstep) HIP_TRY(hipMemcpyAsync(hstep, h->d_h.p, sizeof(int32_t) * h->B, hipMemcpyDeviceToHost, st));
  if (needs_reset) HIP_TRY(hipMemcpyAsync(needs_reset, h->d_need_reset.p, h->B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

}  // extern "C"

// ---- dynamic programming ---------------------------------------------------------------------------------
namespace {

// AUTO rules of the reference dispatchers (infinite_horizon.py:28-36 and :60-64), per instance.
int vi_rule(int64_t S, int A, int64_t nnz) {
  const double size = (double)S * A * (double)S;
  return (size > 300.0 * 3 * 300 && (double)nnz / size < 0.2) ? CMDP_SCHEME_JACOBI : CMDP_SCHEME_GAUSS_SEIDEL;
}
int pe_rule(int64_t S, int A, int64_t nnz) {
  const double size = (double)S * A * (double)S;
  return (S > 200 && (double)nnz / size < 0.2) ? CMDP_SCHEME_JACOBI : CMDP_SCHEME_GAUSS_SEIDEL;
}

// One scheme per launch: all instances of a batch must agree under AUTO (they do when they come from
// one parameterisation); otherwise the caller picks the scheme explicitly or splits the batch.
int resolve_scheme(cmdp_t* h, int scheme, bool pe, bool diam, int* out) {
  if (scheme == CMDP_SCHEME_JACOBI || scheme == CMDP_SCHEME_GAUSS_SEIDEL) { *out = scheme; return CMDP_OK; }
  if (scheme != CMDP_SCHEME_AUTO) return fail(CMDP_ERR_INVALID, "scheme");
  int chosen = 0;
  for (int b = 0; b < h->B; ++b) {
    const int64_t S = h->state_off[b + 1] - h->state_off[b];
    int64_t nnz = h->csr_nnz[b];
    if (diam) nnz = std::max<int64_t>(nnz, 1);  // T_es differs from T only by the rows of the target
    const int s = pe ? pe_rule(S, h->A, nnz) : vi_rule(S, h->A, nnz);
    if (chosen == 0) chosen = s;
    else if (chosen != s)
      return fail(CMDP_ERR_INVALID, "CMDP_SCHEME_AUTO selects different schemes inside this batch; pass the scheme "
                                    "explicitly or split the batch");
  }
  *out = chosen;
  return CMDP_OK;
}

// Launches the sweep kernels for `units` work items (instances, or (instance,target) pairs).
int run_sweeps(cmdp_t* h, int mode, bool diam, int scheme, DpTables t, int64_t units) {
  hipStream_t st = h->stream;
  if (units > 0x7fffffffLL) return fail(CMDP_ERR_INVALID, "too many work items");
  const size_t v_bytes = sizeof(float) * (size_t)h->max_S;
  const DpTables t_dev = t;
  h->zc_used = false;
  if (scheme == CMDP_SCHEME_JACOBI && !diam && h->dp_kernel != 1) {
    // The register-resident kernels touch Q, V and the sweep counts exactly once, when an instance has converged: if the
    // caller's result arrays are page-locked (BatchedMDP.dp_buffers), the kernels store into them directly -- the results of
    // the instances that converge early cross PCIe under the sweeps of the rest, and no copy follows the kernel
    if (h->zc_V) {
      t.Q = h->zc_Q; t.V = h->zc_V;
      if (h->zc_sweeps) t.sweeps = h->zc_sweeps;
      if (h->pin_status.p && h->pin_status_n >= (size_t)units) t.status = h->pin_status.p;
    }
    // register-resident CSR (K2R) when the shapes fit one of the compiled instantiations
    const int A = h->A, K = h->max_row_nnz <= 4 ? 4 : (h->max_row_nnz <= 8 ? 8 : 0);
    const int spt = h->max_S <= 256 ? 1 : (h->max_S <= 512 ? 2 : (h->max_S <= 1024 ? 4 : 0));
    const size_t lds = 2 * sizeof(float) * 256 * (size_t)std::max(spt, 1) + sizeof(float) * 16;  // Va, Vb at fixed offsets
    bool launched = true;
    const dim3 grid((unsigned)units), block(256);
    // K2U when the states' rows share their successors: U gathers instead of A x K (dp_kernel 5 forces it, 2 forbids it)
    const int U = h->max_state_unique == 0 ? 0 : (h->max_state_unique <= 5 ? 5 : (h->max_state_unique <= 8 ? 8 : 0));
    const bool want_u = U > 0 && K > 0 && spt > 0 && spt * U <= 20 && h->dp_kernel != 2 &&
                        (h->dp_kernel == 5 || h->dp_kernel == 7 || 2 * U <= A * K);
    if (h->dp_kernel == 5 && !want_u)
      return fail(CMDP_ERR_UNSUPPORTED, "no distinct-successor instantiation (A=%d, %d distinct successors per state, %d states)",
                  A, h->max_state_unique, h->max_S);
    // K2W (one wavefront per instance, 5..7 states per lane): the batches the reference's scheme rule sends to Jacobi
    // sweeps start at ~260 states, and up to 448 the whole instance fits a wavefront's registers.  Option 7 forces it,
    // 5 keeps K2U.
    const int sptw = (h->max_S + 63) / 64;
    const bool want_w = want_u && U == 5 && K == 4 && sptw <= 7 && (h->dp_kernel == 7 || (h->dp_kernel == 0 && sptw >= 5));
    if (h->dp_kernel == 7 && !want_w)
      return fail(CMDP_ERR_UNSUPPORTED, "no one-wavefront instantiation (A=%d, %d distinct successors per state, %d non-zeros/row, %d states)",
                  A, h->max_state_unique, h->max_row_nnz, h->max_S);
    if (want_w) {
      bool done = true;
      const int st_w = sptw <= 5 ? 5 : sptw;
      const size_t ldsw = 2 * sizeof(float) * 64 * (size_t)st_w;
#define REGW_CASE(AT, ST)                                                                                        \
  if (A == AT && st_w == ST) {                                                                                   \
    if (mode == DP_VI) hipLaunchKernelGGL((k_dp_regw<DP_VI, AT, 5, 4, ST>), grid, dim3(64), ldsw, st, t);          \
    else hipLaunchKernelGGL((k_dp_regw<DP_PE, AT, 5, 4, ST>), grid, dim3(64), ldsw, st, t);                        \
  } else
      // (only instantiations that keep their tables in registers: with four actions, seven states per lane -- and six under
      // policy evaluation, which also holds the policy's rows -- spill to scratch; those batches take K2U below)
      REGW_CASE(2, 5) REGW_CASE(2, 6) REGW_CASE(2, 7) REGW_CASE(3, 5) REGW_CASE(3, 6) REGW_CASE(3, 7)
      REGW_CASE(4, 5)
      if (A == 4 && st_w == 6 && mode == DP_VI) hipLaunchKernelGGL((k_dp_regw<DP_VI, 4, 5, 4, 6>), grid, dim3(64), ldsw, st, t);
      else
      { done = false; }
#undef REGW_CASE
      if (done) {
        HIP_TRY(hipGetLastError());
        h->last_dp_kernel = 7;
        h->zc_used = h->zc_V != nullptr;
        return CMDP_OK;
      }
      if (h->dp_kernel == 7) return fail(CMDP_ERR_UNSUPPORTED, "no one-wavefront instantiation for A=%d, %d states per lane%s", A, st_w,
                                         mode == DP_PE ? " (policy evaluation)" : "");
    }
    if (want_u) {
      bool done = true;
#define REGU_CASE(AT, UT, KT, ST)                                                                             \
  if (A == AT && U == UT && K == KT && spt == ST) {                                                           \
    if (mode == DP_VI) hipLaunchKernelGGL((k_dp_regu<DP_VI, AT, UT, KT, ST>), grid, block, lds, st, t);        \
    else hipLaunchKernelGGL((k_dp_regu<DP_PE, AT, UT, KT, ST>), grid, block, lds, st, t);                      \
  } else
      REGU_CASE(2, 5, 4, 1) REGU_CASE(2, 5, 4, 2) REGU_CASE(2, 5, 8, 1) REGU_CASE(2, 5, 8, 2)
      REGU_CASE(3, 5, 4, 1) REGU_CASE(3, 5, 4, 2) REGU_CASE(3, 5, 8, 1) REGU_CASE(3, 5, 8, 2)
      REGU_CASE(4, 5, 4, 1) REGU_CASE(4, 5, 4, 2) REGU_CASE(4, 5, 8, 1) REGU_CASE(4, 5, 8, 2)
      REGU_CASE(2, 5, 4, 4) REGU_CASE(3, 5, 4, 4) REGU_CASE(4, 5, 4, 4) REGU_CASE(4, 5, 8, 4)
      REGU_CASE(2, 5, 8, 4) REGU_CASE(3, 5, 8, 4)
      REGU_CASE(3, 8, 8, 1) REGU_CASE(3, 8, 8, 2) REGU_CASE(4, 8, 4, 1) REGU_CASE(4, 8, 4, 2)
      REGU_CASE(4, 8, 8, 1) REGU_CASE(4, 8, 8, 2)
      { done = false; }
#undef REGU_CASE
      if (done) {
        HIP_TRY(hipGetLastError());
        h->last_dp_kernel = 5;
        h->zc_used = h->zc_V != nullptr;
        return CMDP_OK;
      }
      if (h->dp_kernel == 5)
        return fail(CMDP_ERR_UNSUPPORTED, "no distinct-successor instantiation for A=%d, U=%d, %d non-zeros/row, %d states", A, U, h->max_row_nnz, h->max_S);
    }
#define REG_CASE(AT, KT, ST)                                                                                  \
  if (A == AT && K == KT && spt == ST) {                                                                      \
    if (mode == DP_VI) hipLaunchKernelGGL((k_dp_reg<DP_VI, AT, KT, ST>), grid, block, lds, st, t);             \
    else hipLaunchKernelGGL((k_dp_reg<DP_PE, AT, KT, ST>), grid, block, lds, st, t);                           \
  } else
    REG_CASE(2, 4, 1) REG_CASE(2, 4, 2) REG_CASE(2, 4, 4)
    REG_CASE(3, 4, 1) REG_CASE(3, 4, 2) REG_CASE(3, 4, 4)
    REG_CASE(4, 4, 1) REG_CASE(4, 4, 2) REG_CASE(4, 4, 4)
    REG_CASE(2, 8, 1) REG_CASE(2, 8, 2)
    REG_CASE(3, 8, 1) REG_CASE(3, 8, 2)
    REG_CASE(4, 8, 1) REG_CASE(4, 8, 2)
    { launched = false; }
#undef REG_CASE
    if (launched) {
      HIP_TRY(hipGetLastError());
      h->last_dp_kernel = 2;
      h->zc_used = h->zc_V != nullptr;
      return CMDP_OK;
    }
    if (h->dp_kernel == 2) return fail(CMDP_ERR_UNSUPPORTED, "no register-resident instantiation for A=%d, %d non-zeros/row, %d states", A, h->max_row_nnz, h->max_S);
  }
  t = t_dev;   // the workgroup / wavefront kernels below keep their working values in the device arrays
  if (scheme == CMDP_SCHEME_JACOBI) {
    const size_t base = 2 * v_bytes + sizeof(float) * 4 * (kDpBlock / 64);
    const size_t csr = sizeof(int32_t) * ((size_t)h->max_S * h->A + 1) + 8 * (size_t)h->max_inst_nnz +
                       sizeof(float) * (size_t)h->max_S * h->A;
    if (base > (size_t)kLdsBudget)
      return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states does not fit the LDS-resident sweep (2*4*S > 160 KiB)", h->max_S);
    // CSR in LDS when two workgroups still fit on a CU; otherwise it is streamed from L2/HBM every sweep
    const bool csr_lds = base + csr <= (size_t)kLdsBudget / 2;
    const size_t lds = csr_lds ? base + csr : base;
    const dim3 grid((unsigned)units), block(kDpBlock);
#define LAUNCH_BLOCK(MODE, DIAM)                                                                         \
  do {                                                                                                   \
    if (csr_lds) {                                                                                       \
      if (int rc = set_lds(k_dp_block<MODE, DIAM, true>, lds)) return rc;                                \
      hipLaunchKernelGGL((k_dp_block<MODE, DIAM, true>), grid, block, lds, st, t);                        \
    } else {                                                                                             \
      if (int rc = set_lds(k_dp_block<MODE, DIAM, false>, lds)) return rc;                               \
      hipLaunchKernelGGL((k_dp_block<MODE, DIAM, false>), grid, block, lds, st, t);                       \
    }                                                                                                    \
  } while (0)
    if (diam) LAUNCH_BLOCK(DP_VI, true);
    else if (mode == DP_VI) LAUNCH_BLOCK(DP_VI, false);
    else LAUNCH_BLOCK(DP_PE, false);
#undef LAUNCH_BLOCK
    h->last_dp_kernel = 1;
  } else {
    if (v_bytes > (size_t)kLdsBudget)
      return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states does not fit the LDS-resident sweep", h->max_S);
    const dim3 grid((unsigned)units), block(64);
#define LAUNCH_WAVE(MODE, DIAM)                                                       \
  do {                                                                                \
    if (int rc = set_lds(k_dp_wave_gs<MODE, DIAM>, v_bytes)) return rc;               \
    hipLaunchKernelGGL((k_dp_wave_gs<MODE, DIAM>), grid, block, v_bytes, st, t);       \
  } while (0)
    if (diam) LAUNCH_WAVE(DP_VI, true);
    else if (mode == DP_VI) LAUNCH_WAVE(DP_VI, false);
    else LAUNCH_WAVE(DP_PE, false);
#undef LAUNCH_WAVE
    h->last_dp_kernel = 6;
  }
  HIP_TRY(hipGetLastError());
  return CMDP_OK;
}

int check_status(cmdp_t* h, int64_t units) {
  std::vector<int32_t> status((size_t)units);
  HIP_TRY(hipMemcpyAsync(status.data(), h->d_status.p, sizeof(int32_t) * units, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int64_t u = 0; u < units; ++u) {
    if (status[u] == CMDP_ERR_MAX_ITER) return fail(CMDP_ERR_MAX_ITER, "work item %lld did not converge within max_sweeps", (long long)u);
    if (status[u] == CMDP_ERR_MAX_VALUE) return fail(CMDP_ERR_MAX_VALUE, "work item %lld exceeded max_abs_value", (long long)u);
  }
  return CMDP_OK;
}

int discounted(cmdp_t* h, int mode, const float* pi, float gamma, double eps, int scheme, int64_t max_sweeps,
               double max_abs, const float* R_override, float* Q, float* V, int64_t* sweeps) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!Q || !V) return fail(CMDP_ERR_INVALID, "null output");
  if (mode == DP_PE && !pi) return fail(CMDP_ERR_INVALID, "pi is required");
  if (max_sweeps < 1) return fail(CMDP_ERR_INVALID, "max_sweeps < 1");
  int sch = 0;
  if (int rc = resolve_scheme(h, scheme, mode == DP_PE, false, &sch)) return rc;
  hipStream_t st = h->stream;
  const int64_t R = h->n_rows, NS = h->n_states;
  if (R_override) HIP_TRY(h->d_Rov.upload(R_override, R, st));
  if (pi) HIP_TRY(h->d_pi.upload(pi, R, st));
  if (h->d_Q.n < (size_t)R) HIP_TRY(h->d_Q.alloc(R));
  if (h->d_V.n < (size_t)NS) HIP_TRY(h->d_V.alloc(NS));
  if (h->d_sweeps.n < (size_t)h->B) HIP_TRY(h->d_sweeps.alloc(h->B));
  if (h->d_status.n < (size_t)h->B) HIP_TRY(h->d_status.alloc(h->B));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = R_override ? h->d_Rov.p : h->d_R.p; t.pi = pi ? h->d_pi.p : nullptr;
  t.unit_off = nullptr; t.gamma = gamma; t.eps = eps; t.max_abs = max_abs; t.max_sweeps = max_sweeps;
  t.Q = h->d_Q.p; t.V = h->d_V.p; t.sweeps = h->d_sweeps.p; t.per_target = nullptr; t.status = h->d_status.p;
  if (!h->ev_dp0) {
    HIP_TRY(hipEventCreate(&h->ev_dp0));
    HIP_TRY(hipEventCreate(&h->ev_dp1));
  }
  // device-visible aliases of page-locked result arrays (null for pageable memory): see run_sweeps
  auto alias = [](void* p) -> void* {
    hipPointerAttribute_t a{};
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer) return a.devicePointer;
    (void)hipGetLastError();   // pageable memory is reported as an error by some runtime versions
    return nullptr;
  };
  static const bool zc_env = !(std::getenv("CMDP_DP_ZERO_COPY") && std::atoi(std::getenv("CMDP_DP_ZERO_COPY")) == 0);
  h->zc_Q = zc_env ? static_cast<float*>(alias(Q)) : nullptr;
  h->zc_V = (zc_env && h->zc_Q) ? static_cast<float*>(alias(V)) : nullptr;
  h->zc_sweeps = (h->zc_V && sweeps) ? static_cast<int64_t*>(alias(sweeps)) : nullptr;
  if (!h->zc_V) h->zc_Q = nullptr;
  if (h->zc_V && h->pin_status_n < (size_t)h->B) {
    if (h->pin_status.p) { (void)hipHostFree(h->pin_status.p); h->pin_status.p = nullptr; }
    if (int rc = h->pin_status.alloc((size_t)h->B)) return rc;
    h->pin_status_n = (size_t)h->B;
  }
  HIP_TRY(hipEventRecord(h->ev_dp0, st));
  const int rc_run = run_sweeps(h, mode, false, sch, t, h->B);
  const bool zc = h->zc_used, zc_sw = zc && h->zc_sweeps;
  h->zc_Q = h->zc_V = nullptr; h->zc_sweeps = nullptr; h->zc_used = false;
  if (rc_run) return rc_run;
  HIP_TRY(hipEventRecord(h->ev_dp1, st));
  if (!zc) {
    HIP_TRY(hipMemcpyAsync(Q, h->d_Q.p, sizeof(float) * R, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(V, h->d_V.p, sizeof(float) * NS, hipMemcpyDeviceToHost, st));
  }
  if (sweeps && !zc_sw) HIP_TRY(hipMemcpyAsync(sweeps, h->d_sweeps.p, sizeof(int64_t) * h->B, hipMemcpyDeviceToHost, st));
  if (zc) {   // the kernels wrote the status words into page-locked memory as well: nothing left to copy
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t u = 0; u < h->B; ++u) {
      if (h->pin_status.p[u] == CMDP_ERR_MAX_ITER) return fail(CMDP_ERR_MAX_ITER, "work item %lld did not converge within max_sweeps", (long long)u);
      if (h->pin_status.p[u] == CMDP_ERR_MAX_VALUE) return fail(CMDP_ERR_MAX_VALUE, "work item %lld exceeded max_abs_value", (long long)u);
    }
    return CMDP_OK;
  }
  return check_status(h, h->B);
}

int episodic(cmdp_t* h, int mode, int H, const float* pi, const float* R_override, float* Q, float* V) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!Q || !V || H < 1) return fail(CMDP_ERR_INVALID, "bad argument");
  if (mode == DP_PE && !pi) return fail(CMDP_ERR_INVALID, "pi is required");
  hipStream_t st = h->stream;
  const int64_t R = h->n_rows, NS = h->n_states;
  const size_t lds = 2 * sizeof(float) * (size_t)h->max_S;
  if (lds > (size_t)kLdsBudget) return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states does not fit LDS", h->max_S);
  if (R_override) HIP_TRY(h->d_Rov.upload(R_override, R, st));
  if (pi) HIP_TRY(h->d_pi.upload(pi, (size_t)H * R, st));
  const size_t nq = (size_t)(H + 1) * R, nv = (size_t)(H + 1) * NS;
  if (h->d_Q.n < nq) HIP_TRY(h->d_Q.alloc(nq));
  if (h->d_V.n < nv) HIP_TRY(h->d_V.alloc(nv));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = R_override ? h->d_Rov.p : h->d_R.p; t.pi = pi ? h->d_pi.p : nullptr;
  const dim3 grid(h->B), block(kDpBlock);
  if (mode == DP_VI) {
    if (int rc = set_lds(k_episodic<DP_VI>, lds)) return rc;
    hipLaunchKernelGGL((k_episodic<DP_VI>), grid, block, lds, st, t, H, h->d_Q.p, h->d_V.p);
  } else {
    if (int rc = set_lds(k_episodic<DP_PE>, lds)) return rc;
    hipLaunchKernelGGL((k_episodic<DP_PE>), grid, block, lds, st, t, H, h->d_Q.p, h->d_V.p);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(Q, h->d_Q.p, sizeof(float) * nq, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(V, h->d_V.p, sizeof(float) * nv, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

}  // namespace

extern "C" {

int cmdp_vi_discounted(cmdp_t* h, float gamma, double epsilon, int scheme, int64_t max_sweeps, double max_abs_value,
                       const float* R_override, float* Q, float* V, int64_t* sweeps) {
  return discounted(h, DP_VI, nullptr, gamma, epsilon, scheme, max_sweeps, max_abs_value, R_override, Q, V, sweeps);
}

int cmdp_pe_discounted(cmdp_t* h, const float* pi, float gamma, double epsilon, int scheme, int64_t max_sweeps,
                       const float* R_override, float* Q, float* V, int64_t* sweeps) {
  return discounted(h, DP_PE, pi, gamma, epsilon, scheme, max_sweeps, 0.0, R_override, Q, V, sweeps);
}

int cmdp_vi_episodic(cmdp_t* h, int H, const float* R_override, float* Q, float* V) {
  return episodic(h, DP_VI, H, nullptr, R_override, Q, V);
}

int cmdp_pe_episodic(cmdp_t* h, int H, const float* pi, const float* R_override, float* Q, float* V) {
  return episodic(h, DP_PE, H, pi, R_override, Q, V);
}

// K5S reads a state's value row and its successors' rows once per sweep and target group; with the builder's state
// numbering (depth-first over the grid) a state's successors sit thousands of rows away and every one of them is a fresh
// HBM fetch (PMC: 3.1 x the algorithmic reads at C5).  relabel_states orders the states of every large instance in
// breadth-first clusters of the undirected transition graph (grow a cluster from a seed until it has `cluster` states,
// seed the next one from the frontier it left) so that the rows a chunk of states gathers were fetched by the chunks just
// before it, and build_ell_relabelled stores the fixed-width rows in that order: row n = the row of original state
// orig_of[n], entries in their original order, columns translated.  Sums and stopping rule see the same numbers in the
// same order -- results are bit-equal -- only the memory the gathers touch moves.
constexpr int kK5sCluster = 80;          // states per cluster (8 wavefronts x 10 states = what a workgroup walks at a time at C5)

static void relabel_states(int S, int A, const int64_t* ptr, const int32_t* col, int cluster, std::vector<int32_t>& order) {
  // undirected adjacency (CSR) of the instance's transition graph
  std::vector<int32_t> deg((size_t)S + 1, 0);
  auto each_edge = [&](auto&& f) {
    for (int s = 0; s < S; ++s)
      for (int64_t k = ptr[(int64_t)s * A]; k < ptr[(int64_t)(s + 1) * A]; ++k)
        if (col[k] != s) f(s, col[k]);
  };
  each_edge([&](int s, int w) { ++deg[(size_t)s + 1]; ++deg[(size_t)w + 1]; });
  std::vector<int64_t> off((size_t)S + 1, 0);
  for (int s = 0; s < S; ++s) off[(size_t)s + 1] = off[(size_t)s] + deg[(size_t)s + 1];
  std::vector<int32_t> adj((size_t)off[(size_t)S]);
  std::vector<int64_t> fill(off.begin(), off.end() - 1);
  each_edge([&](int s, int w) { adj[(size_t)fill[(size_t)s]++] = w; adj[(size_t)fill[(size_t)w]++] = s; });
  order.clear();
  order.reserve((size_t)S);
  std::vector<char> seen((size_t)S, 0), placed((size_t)S, 0);
  std::vector<int32_t> pending, queue;
  size_t pending_pos = 0;
  int next_unplaced = 0;
  while ((int)order.size() < S) {
    int seed = -1;
    while (pending_pos < pending.size()) {
      const int c = pending[pending_pos++];
      if (!placed[(size_t)c] && !seen[(size_t)c]) { seed = c; break; }
    }
    if (seed < 0) {
      while (placed[(size_t)next_unplaced]) ++next_unplaced;
      seed = next_unplaced;
    }
    queue.clear();
    queue.push_back(seed);
    seen[(size_t)seed] = 1;
    size_t head = 0;
    int cnt = 0;
    while (head < queue.size() && cnt < cluster) {
      const int v = queue[head++];
      order.push_back(v);
      placed[(size_t)v] = 1;
      ++cnt;
      for (int64_t k = off[(size_t)v]; k < off[(size_t)v + 1]; ++k) {
        const int w = adj[(size_t)k];
        if (!seen[(size_t)w]) { seen[(size_t)w] = 1; queue.push_back(w); }
      }
    }
    for (; head < queue.size(); ++head) {  // the frontier left over seeds the next clusters
      seen[(size_t)queue[head]] = 0;
      pending.push_back(queue[head]);
    }
  }
}

static int build_ell_relabelled(cmdp_t* h, int K, int cluster) {
  hipStream_t st = h->stream;
  const int A = h->A;
  const int64_t NR = h->n_rows, NS = h->n_states;
  std::vector<int64_t> ptr((size_t)NR + 1);
  std::vector<int32_t> col((size_t)h->n_csr);
  std::vector<float> val((size_t)h->n_csr);
  HIP_TRY(hipMemcpyAsync(ptr.data(), h->d_csr_ptr.p, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(col.data(), h->d_csr_col.p, sizeof(int32_t) * col.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(val.data(), h->d_csr_val.p, sizeof(float) * val.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const size_t rows_p = (size_t)NR + 64 / K + 1;  // tail padding, as k_build_ell
  std::vector<int32_t> ecol(rows_p * K, 0), new_of((size_t)NS);
  std::vector<float> eval_(rows_p * K, 0.0f);
  std::vector<int32_t> order;
  for (int b = 0; b < h->B; ++b) {
    const int64_t so = h->state_off[b];
    const int S = (int)(h->state_off[b + 1] - so);
    // the instance's rows start at ptr[so * A]; relabel_states wants pointers relative to the instance's first row
    if (S >= h->relabel_min_states) {   // default 8192: below, the 2 x S x 256 B value arrays of a group sit in L2 anyway
      relabel_states(S, A, ptr.data() + so * A, col.data(), cluster, order);
    } else {
      order.resize((size_t)S);
      for (int s = 0; s < S; ++s) order[(size_t)s] = s;
    }
    for (int n = 0; n < S; ++n) new_of[(size_t)(so + order[(size_t)n])] = n;
    for (int n = 0; n < S; ++n) {
      for (int a = 0; a < A; ++a) {
        const int64_t ro = (so + order[(size_t)n]) * A + a, rn = (so + n) * A + a;
        const int64_t lo = ptr[(size_t)ro], hi = ptr[(size_t)ro + 1];
        if (hi - lo > K) return fail(CMDP_ERR_INVALID, "row %lld has more than %d non-zeros", (long long)ro, K);
        const int32_t c0 = hi > lo ? new_of[(size_t)(so + col[(size_t)lo])] : 0;
        for (int k = 0; k < K; ++k) {
          const bool in = lo + k < hi;
          ecol[(size_t)rn * K + k] = in ? new_of[(size_t)(so + col[(size_t)(lo + k)])] : c0;
          eval_[(size_t)rn * K + k] = in ? val[(size_t)(lo + k)] : 0.0f;
        }
      }
    }
  }
  HIP_TRY(h->d_ell_col.upload(ecol.data(), ecol.size(), st));
  HIP_TRY(h->d_ell_val.upload(eval_.data(), eval_.size(), st));
  HIP_TRY(h->d_ell_newof.upload(new_of.data(), new_of.size(), st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

constexpr int kK5tRmax = 48;  // tile rows per cluster: 6 wavefronts x 48 rows x 256 B = 72 KiB of LDS, two workgroups per CU
constexpr int kK5tNw = 6;

// Host side of K5T: cuts every instance into clusters of <= K5T_C states whose rows (own + distinct outside successors)
// fit a tile, and writes the per-cluster row lists and the rows' entries re-indexed to tile positions (original column
// order kept).  Breadth-first region growing over the undirected transition graph keeps the halo small on grid worlds.
static int build_tiles(cmdp_t* h, int K) {
  hipStream_t st = h->stream;
  const int A = h->A, AK = A * K;
  const int64_t NR = h->n_rows;
  std::vector<int64_t> ptr((size_t)NR + 1);
  std::vector<int32_t> col((size_t)h->n_csr);
  std::vector<float> val((size_t)h->n_csr);
  HIP_TRY(hipMemcpyAsync(ptr.data(), h->d_csr_ptr.p, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(col.data(), h->d_csr_col.p, sizeof(int32_t) * col.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(val.data(), h->d_csr_val.p, sizeof(float) * val.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<int32_t> c0v, nclv, cl_n, cl_R, rows, lcol;
  std::vector<float> lval;
  int64_t total_rows = 0, total_states = 0;
  for (int b = 0; b < h->B; ++b) {
    const int64_t so = h->state_off[b];
    const int S = (int)(h->state_off[b + 1] - so);
    // distinct successors of every state (directed: what a sweep gathers), and undirected neighbours for the growth
    std::vector<std::vector<int32_t>> succ((size_t)S), nbr((size_t)S);
    for (int s = 0; s < S; ++s) {
      auto& v = succ[(size_t)s];
      for (int64_t r = (so + s) * A; r < (so + s + 1) * A; ++r)
        for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) v.push_back(col[k]);
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    for (int s = 0; s < S; ++s)
      for (int32_t w : succ[(size_t)s])
        if (w != s) { nbr[(size_t)s].push_back(w); nbr[(size_t)w].push_back(s); }
    std::vector<char> assigned((size_t)S, 0);
    std::vector<int32_t> mark((size_t)S, -1);   // cluster id while the state is in the cluster or its halo
    std::vector<char> inside((size_t)S, 0);
    std::vector<int32_t> local((size_t)S, -1);
    c0v.push_back((int32_t)cl_n.size());
    int next_seed = 0;
    std::vector<int32_t> seeds;                 // frontier left over by finished clusters: grow next to them
    size_t seed_pos = 0;
    int n_clusters = 0;
    while (true) {
      int seed = -1;
      while (seed_pos < seeds.size()) {
        const int c = seeds[seed_pos++];
        if (!assigned[(size_t)c]) { seed = c; break; }
      }
      if (seed < 0) {
        while (next_seed < S && assigned[(size_t)next_seed]) ++next_seed;
        if (next_seed == S) break;
        seed = next_seed;
      }
      const int cid = n_clusters++;
      std::vector<int32_t> members, halo, queue;
      int n_halo = 0;
      auto try_add = [&](int v) -> bool {
        // rows the tile would hold with v inside: members + 1, halo - (v was halo) + (new outside successors of v)
        int add = 0;
        for (int32_t w : succ[(size_t)v])
          if (w != v && !(inside[(size_t)w]) && mark[(size_t)w] != cid) ++add;
        const int was_halo = (mark[(size_t)v] == cid && !inside[(size_t)v]) ? 1 : 0;
        if ((int)members.size() + 1 + n_halo - was_halo + add > kK5tRmax) return false;
        if (was_halo) --n_halo;
        inside[(size_t)v] = 1;
        mark[(size_t)v] = cid;
        members.push_back(v);
        for (int32_t w : succ[(size_t)v])
          if (w != v && !inside[(size_t)w] && mark[(size_t)w] != cid) { mark[(size_t)w] = cid; halo.push_back(w); ++n_halo; }
        return true;
      };
      // Greedy growth: of the states adjacent to the cluster, take the one with the most neighbours already inside
      // (ties: first found) -- compact blobs with a small halo that also fill the gaps between earlier clusters.
      // `local` doubles as the score of a frontier state while the cluster grows (-1: not on the frontier).
      std::vector<int32_t> frontier{seed};
      local[(size_t)seed] = 0;
      while (!frontier.empty() && (int)members.size() < K5T_C) {
        size_t best = 0;
        for (size_t i = 1; i < frontier.size(); ++i)
          if (local[(size_t)frontier[i]] > local[(size_t)frontier[best]]) best = i;
        const int v = frontier[best];
        frontier[best] = frontier.back();
        frontier.pop_back();
        if (!try_add(v)) { queue.push_back(v); continue; }   // does not fit the tile: left for a later cluster
        assigned[(size_t)v] = 1;
        queue.push_back(v);
        for (int32_t w : nbr[(size_t)v]) {
          if (assigned[(size_t)w]) continue;
          if (local[(size_t)w] < 0) { local[(size_t)w] = 0; frontier.push_back(w); }
          if (local[(size_t)w] >= 0) ++local[(size_t)w];
        }
      }
      for (int32_t w : frontier) queue.push_back(w);
      for (size_t i = 0; i < queue.size(); ++i) {   // reset the scores; unassigned leftovers seed later clusters
        local[(size_t)queue[i]] = -1;
        if (!assigned[(size_t)queue[i]]) seeds.push_back(queue[i]);
      }
      if (members.empty()) {  // a single state whose own successors exceed a tile: K5T cannot take this instance
        return fail(CMDP_ERR_UNSUPPORTED, "state %d of instance %d has more than %d distinct successors: no tile holds its row",
                    seed, b, kK5tRmax - 1);
      }
      // tile rows: members, then the halo states that are still outside (a halo state may have joined later)
      std::vector<int32_t> trow(members);
      for (int32_t w : halo)
        if (!inside[(size_t)w]) trow.push_back(w);
      for (size_t i = 0; i < trow.size(); ++i) local[(size_t)trow[i]] = (int32_t)i;
      const int R = (int)((trow.size() + 3) / 4 * 4);
      cl_n.push_back((int32_t)members.size());
      cl_R.push_back(R);
      total_rows += (int64_t)trow.size();
      total_states += (int64_t)members.size();
      for (int i = 0; i < kK5tRmax; ++i) rows.push_back(i < (int)trow.size() ? trow[(size_t)i] : trow[0]);
      for (int u = 0; u < K5T_C; ++u)
        for (int a = 0; a < A; ++a) {
          int64_t lo = 0, hi = 0;
          if (u < (int)members.size()) {
            const int64_t r = (so + members[(size_t)u]) * A + a;
            lo = ptr[r]; hi = ptr[r + 1];
          }
          for (int k = 0; k < K; ++k) {
            const bool in = lo + k < hi;
            lcol.push_back(in ? local[(size_t)col[lo + k]] : (hi > lo ? local[(size_t)col[lo]] : 0));
            lval.push_back(in ? val[lo + k] : 0.0f);
          }
        }
      for (int32_t v : members) inside[(size_t)v] = 0;
      for (size_t i = 0; i < trow.size(); ++i) local[(size_t)trow[i]] = -1;
    }
    nclv.push_back(n_clusters);
  }
  for (int i = 0; i < 64; ++i) { lcol.push_back(0); lval.push_back(0.0f); }  // a 64-entry load past the last cluster
  (void)AK;
  HIP_TRY(h->d_tl_c0.upload(c0v.data(), c0v.size(), st));
  HIP_TRY(h->d_tl_ncl.upload(nclv.data(), nclv.size(), st));
  HIP_TRY(h->d_tl_n.upload(cl_n.data(), cl_n.size(), st));
  HIP_TRY(h->d_tl_R.upload(cl_R.data(), cl_R.size(), st));
  HIP_TRY(h->d_tl_rows.upload(rows.data(), rows.size(), st));
  HIP_TRY(h->d_tl_lcol.upload(lcol.data(), lcol.size(), st));
  HIP_TRY(h->d_tl_val.upload(lval.data(), lval.size(), st));
  HIP_TRY(hipStreamSynchronize(st));
  h->tile_K = K;
  h->tile_rows_per_state = total_states ? (double)total_rows / (double)total_states : 0.0;
  if (std::getenv("CMDP_K5T_DEBUG"))
    std::fprintf(stderr, "[K5T] %zu clusters, %.1f states and %.1f tile rows per cluster (%.3f rows gathered per state)\n",
                 cl_n.size(), (double)total_states / cl_n.size(), (double)total_rows / cl_n.size(), h->tile_rows_per_state);
  return CMDP_OK;
}

// fixed-width rows of the K5S family (k_build_ell / build_ell_relabelled), built once per handle and width
static int ensure_ell(cmdp_t* h, int K) {
  hipStream_t st = h->stream;
  if (h->ell_K != K) {
    // CMDP_K5S_CLUSTER: tuning aid -- states per breadth-first cluster of the locality order, 0 = keep the caller's order
    static const int cluster_env = std::getenv("CMDP_K5S_CLUSTER") ? std::atoi(std::getenv("CMDP_K5S_CLUSTER")) : -1;
    const int cluster = cluster_env >= 0 ? cluster_env : kK5sCluster;
    if (cluster > 0 && h->max_S >= h->relabel_min_states) {
      if (int rc = build_ell_relabelled(h, K, cluster)) return rc;
      h->ell_relabelled = true;
    } else {
      const size_t rows_p = (size_t)h->n_rows + 64 / K + 1;
      HIP_TRY(h->d_ell_col.alloc(rows_p * K));
      HIP_TRY(h->d_ell_val.alloc(rows_p * K));
      hipLaunchKernelGGL(k_build_ell, dim3(grid_for((int64_t)rows_p, 256)), dim3(256), 0, st, h->n_rows, K,
                         h->d_csr_ptr.p, h->d_csr_col.p, h->d_csr_val.p, h->d_ell_col.p, h->d_ell_val.p);
      HIP_TRY(hipGetLastError());
      h->ell_relabelled = false;
    }
    h->ell_K = K;
  }
  return CMDP_OK;
}

// K5S driver: targets [unit_lo, unit_hi) of the flat state space in groups of 64 consecutive targets of one instance;
// as many groups per launch as the value-array workspace allows.
static int diameter_lanes(cmdp_t* h, DpTables t, int64_t unit_lo, int64_t unit_hi) {
  hipStream_t st = h->stream;
  std::vector<int32_t> inst, t0, cnt;
  std::vector<int64_t> vfl;  // floats per group
  const int K_ell = h->max_row_nnz <= 2 ? 2 : (h->max_row_nnz <= 4 ? 4 : (h->max_row_nnz <= 8 ? 8 : 0));
  const bool ell_ok = K_ell && h->dp_kernel != 4 && h->dp_kernel != 6 && h->A >= 2 && h->A <= 4 && h->A * K_ell <= 32;
  // (two targets per lane -- value rows of 128 floats, the row walk paid once per 128 targets -- measured 2.47 s against
  // 1.79 s at C5: the wider rows halve every group's window in L2; not kept)
  // K5C: clusters of workgroups per group (k_diam_cluster) for instances large enough for the value rows to overflow the L2s;
  // CMDP_K5C = 0 switches it off, = CL (8 | 16 | 32) chooses the cluster size (tuning aid)
  const int k5c_env = std::getenv("CMDP_K5C") ? std::atoi(std::getenv("CMDP_K5C")) : -1;   // read per call: the tests switch it
  const int CLs = k5c_env > 0 ? k5c_env : 16;
  // (a give-up costs every workgroup its 2-second spin: on a GPU that other streams / ranks keep busy the persistent launch is
  // not tried again at once -- the back-off doubles with every consecutive give-up; CMDP_K5C > 0 overrides it)
  const bool backing_off = h->k5c_skip > 0 && k5c_env <= 0;
  if (backing_off) h->k5c_skip--;
  const bool use_cluster = ell_ok && k5c_env != 0 && h->cus % (8 * CLs) == 0 && h->max_S >= h->relabel_min_states && !backing_off;
  const int64_t GW = 64;  // targets per group
  for (int b = 0; b < h->B; ++b) {
    const int64_t so = h->state_off[b], S = h->state_off[b + 1] - so;
    const int64_t lo = std::max<int64_t>(unit_lo, so) - so, hi = std::min<int64_t>(unit_hi, so + S) - so;
    for (int64_t x = lo; x < hi; x += GW) {
      inst.push_back(b);
      t0.push_back((int32_t)x);
      cnt.push_back((int32_t)std::min<int64_t>(GW, hi - x));
      vfl.push_back(2 * S * GW);
    }
  }
  const size_t G = inst.size();
  if (use_cluster && G > 0) {
    if (int rc = ensure_ell(h, K_ell)) return rc;
    const int n_clusters = h->cus / CLs;
    DiamClusterArgs ca{};
    ca.n_groups = (int)G; ca.n_clusters = n_clusters; ca.vstride = 2 * (int64_t)h->max_S * 64;
    const size_t vfloats = (size_t)n_clusters * (size_t)ca.vstride;
    if (h->d_dl_v.n < vfloats) {
      if (hipError_t e = h->d_dl_v.alloc(vfloats); e != hipSuccess)
        return fail(CMDP_ERR_HIP, "K5C workspace of %zu bytes: %s", vfloats * sizeof(float), hipGetErrorString(e));
    }
    HIP_TRY(h->d_k5c_red.alloc((size_t)n_clusters * 2 * CLs * 2 * 64));
    HIP_TRY(h->d_k5c_bar.alloc((size_t)n_clusters + 1 + (size_t)n_clusters * CLs));
    HIP_TRY(h->d_dl_inst.upload(inst.data(), G, st));
    HIP_TRY(h->d_dl_t0.upload(t0.data(), G, st));
    HIP_TRY(h->d_dl_cnt.upload(cnt.data(), G, st));
    ca.cred = h->d_k5c_red.p; ca.cbar = h->d_k5c_bar.p; ca.err = reinterpret_cast<int*>(h->d_k5c_bar.p + n_clusters);
    ca.xcc = ca.err + 1;
    // 2 s of the 100 MHz wall clock; CMDP_K5C_TIMEOUT_TICKS overrides (the test-suite sets 1 to drive the fall-back to K5S)
    ca.timeout_ticks = std::getenv("CMDP_K5C_TIMEOUT_TICKS") ? std::atoll(std::getenv("CMDP_K5C_TIMEOUT_TICKS")) : 200000000LL;
    DiamLanesArgs g{h->d_dl_inst.p, h->d_dl_t0.p, h->d_dl_cnt.p, nullptr, h->d_dl_v.p};
    const int32_t* new_of = h->ell_relabelled ? h->d_ell_newof.p : nullptr;
    const int A = h->A, K = K_ell;
    const unsigned grid = (unsigned)(n_clusters * CLs);
    // first with the XCD-scope barriers (the members verify that they share an XCD); a cluster spread over XCDs makes the
    // launch end with err = 2 and it is repeated with agent-scope barriers; CMDP_K5C_SCOPE = agent skips the first form
    static const bool agent_env = std::getenv("CMDP_K5C_SCOPE") && !std::strcmp(std::getenv("CMDP_K5C_SCOPE"), "agent");
    for (int pass = (agent_env || h->k5c_agent_scope) ? 1 : 0; pass < 2; ++pass) {
      HIP_TRY(h->d_k5c_bar.zero(st));   // counters, error flag, XCC ids
      bool launched = true;
#define K5C_CASE(CLT, AT, KT)                                                                                       \
  if (CLs == CLT && A == AT && K == KT) {                                                                           \
    if (pass == 0) hipLaunchKernelGGL((k_diam_cluster<CLT, AT, KT, true>), dim3(grid), dim3(1024), 0, st, t, g, ca, h->d_ell_col.p, h->d_ell_val.p, new_of); \
    else hipLaunchKernelGGL((k_diam_cluster<CLT, AT, KT, false>), dim3(grid), dim3(1024), 0, st, t, g, ca, h->d_ell_col.p, h->d_ell_val.p, new_of); \
  } else
#define K5C_SHAPES(CLT) K5C_CASE(CLT, 2, 2) K5C_CASE(CLT, 2, 4) K5C_CASE(CLT, 2, 8) K5C_CASE(CLT, 3, 2) K5C_CASE(CLT, 3, 4) \
                        K5C_CASE(CLT, 3, 8) K5C_CASE(CLT, 4, 2) K5C_CASE(CLT, 4, 4) K5C_CASE(CLT, 4, 8)
      K5C_SHAPES(8) K5C_SHAPES(16) K5C_SHAPES(32) { launched = false; }
#undef K5C_SHAPES
#undef K5C_CASE
      if (!launched) break;
      HIP_TRY(hipGetLastError());
      int err = 0;
      HIP_TRY(hipMemcpyAsync(&err, ca.err, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (!err) { h->k5c_launches++; h->k5c_backoff = 0; return CMDP_OK; }
      if (err == 2 && pass == 0) { h->k5c_agent_scope = true; continue; }   // this device does not deal workgroups as assumed
      h->k5c_timeouts++;   // a cluster's workgroups were not all resident: the groups are solved again, one workgroup each
      h->k5c_backoff = std::min(1024, std::max(8, 2 * h->k5c_backoff));
      h->k5c_skip = h->k5c_backoff;
      break;
    }
  }
  // the workspace is also bounded by what the device has free right now (other handles / ranks sharing the GPU):
  // 80 % of the free bytes plus what this handle already holds for the purpose; fewer groups per launch, same results
  size_t ws_cap = h->dl_ws_bytes;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      ws_cap = std::min(ws_cap, h->d_dl_v.n * sizeof(float) + free_b / 5 * 4);
  }
  size_t g0 = 0;
  while (g0 < G) {
    size_t g1 = g0, floats = 0;
    std::vector<int64_t> voff;
    while (g1 < G && (g1 == g0 || (floats + (size_t)vfl[g1]) * sizeof(float) <= ws_cap)) {
      voff.push_back((int64_t)floats);
      floats += (size_t)vfl[g1];
      ++g1;
    }
    const size_t n = g1 - g0;
    if (h->d_dl_v.n < floats) {
      if (hipError_t e = h->d_dl_v.alloc(floats); e != hipSuccess)
        return fail(CMDP_ERR_HIP, "K5S workspace of %zu bytes: %s", floats * sizeof(float), hipGetErrorString(e));
    }
    HIP_TRY(h->d_dl_inst.upload(inst.data() + g0, n, st));
    HIP_TRY(h->d_dl_t0.upload(t0.data() + g0, n, st));
    HIP_TRY(h->d_dl_cnt.upload(cnt.data() + g0, n, st));
    HIP_TRY(h->d_dl_voff.upload(voff.data(), n, st));
    DiamLanesArgs g{h->d_dl_inst.p, h->d_dl_t0.p, h->d_dl_cnt.p, h->d_dl_voff.p, h->d_dl_v.p};
    // fixed-width-row kernel when a compiled (A, K) shape fits; option value 4 keeps the generic CSR walker
    const int A = h->A, K = h->max_row_nnz <= 2 ? 2 : (h->max_row_nnz <= 4 ? 4 : (h->max_row_nnz <= 8 ? 8 : 0));
    bool ell = false;
    // K5T (value rows gathered into LDS tiles per cluster of states): on request only (option 6).  At C5 it halves the
    // HBM traffic of K5S and is bit-equal, but runs 2.3 s against 2.05 s -- see DESIGN.md.
    bool tiles = false;
    if (K && A >= 2 && A <= 4 && A * K <= 32 && h->dp_kernel == 6) {
      if (h->tile_K != K) {
        const int rc = build_tiles(h, K);
        if (rc != CMDP_OK && (rc != CMDP_ERR_UNSUPPORTED || h->dp_kernel == 6)) return rc;
      }
      // worth it only while the halo stays small (two rows gathered per state would equal K5S's traffic at best)
      tiles = h->tile_K == K && (h->dp_kernel == 6 || h->tile_rows_per_state <= 2.0);
    }
    if (tiles) {
      TileArgs ta{h->d_tl_c0.p, h->d_tl_ncl.p, h->d_tl_n.p, h->d_tl_R.p, h->d_tl_rows.p, h->d_tl_lcol.p, h->d_tl_val.p};
      const size_t lds = sizeof(float) * 64 * (size_t)kK5tRmax * kK5tNw;
      bool launched = true;
#define TILE_CASE(AT, KT)                                                                                       \
  if (A == AT && K == KT) {                                                                                     \
    if (int rc = set_lds(k_diam_tiles<kK5tNw, AT, KT, kK5tRmax>, lds)) return rc;                               \
    hipLaunchKernelGGL((k_diam_tiles<kK5tNw, AT, KT, kK5tRmax>), dim3((unsigned)n), dim3(kK5tNw * 64), lds, st, t, g, ta); \
  } else
      TILE_CASE(2, 2) TILE_CASE(2, 4) TILE_CASE(2, 8) TILE_CASE(3, 2) TILE_CASE(3, 4) TILE_CASE(3, 8) TILE_CASE(4, 2)
      TILE_CASE(4, 4) TILE_CASE(4, 8) { launched = false; }
#undef TILE_CASE
      if (!launched) tiles = false;
    }
    if (tiles) {
      ell = true;  // handled
    } else if (K && h->dp_kernel != 4 && A >= 2 && A <= 4 && A * K <= 32) {
      if (int rc = ensure_ell(h, K)) return rc;
      const int32_t* new_of = h->ell_relabelled ? h->d_ell_newof.p : nullptr;
      ell = true;
      // wavefronts per group: 8 fill the chip when there are at least two groups per CU; with fewer groups than CUs (a
      // rank's share of C5 on an 8-GPU node: 98 groups) the launch lasts as long as ONE group, so each group gets 16
      static const int k5s_env = std::getenv("CMDP_K5S_NW") ? std::atoi(std::getenv("CMDP_K5S_NW")) : 0;  // tuning aid
      // (and 16 with the locality order: half as many groups share an L2, so a row is still there when the next chunk
      // wants it -- C5 1.83 -> 1.75 s)
      const int k5s_nw = k5s_env ? k5s_env : (((int64_t)n <= (int64_t)h->cus || h->ell_relabelled) ? 16 : 8);
#define ELL_CASE(AT, KT)                                                                                          \
  if (A == AT && K == KT) {                                                                                       \
    if (k5s_nw == 16)                                                                                             \
      hipLaunchKernelGGL((k_diam_lanes_ell<16, AT, KT>), dim3((unsigned)n), dim3(1024), 0, st, t, g, \
                         h->d_ell_col.p, h->d_ell_val.p, new_of);                                                 \
    else if (k5s_nw == 4)                                                                                         \
      hipLaunchKernelGGL((k_diam_lanes_ell<4, AT, KT>), dim3((unsigned)n), dim3(256), 0, st, t, g, \
                         h->d_ell_col.p, h->d_ell_val.p, new_of);                                                 \
    else                                                                                                          \
      hipLaunchKernelGGL((k_diam_lanes_ell<8, AT, KT>), dim3((unsigned)n), dim3(512), 0, st, t, g, \
                         h->d_ell_col.p, h->d_ell_val.p, new_of);                                                 \
  } else
      ELL_CASE(2, 2) ELL_CASE(2, 4) ELL_CASE(2, 8) ELL_CASE(3, 2) ELL_CASE(3, 4) ELL_CASE(3, 8) ELL_CASE(4, 2)
      ELL_CASE(4, 4) ELL_CASE(4, 8) { ell = false; }
#undef ELL_CASE
    }
    if (!ell) hipLaunchKernelGGL(k_diam_lanes<8>, dim3((unsigned)n), dim3(512), 0, st, t, g);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // the upload staging vectors die at the end of this iteration
    g0 = g1;
  }
  return CMDP_OK;
}

int cmdp_diameter(cmdp_t* h, double epsilon, int scheme, int64_t max_sweeps, float* per_target, float* diameter) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!diameter) return fail(CMDP_ERR_INVALID, "null output");
  if (max_sweeps < 1) return fail(CMDP_ERR_INVALID, "max_sweeps < 1");
  int sch = 0;
  if (int rc = resolve_scheme(h, scheme, false, true, &sch)) return rc;
  hipStream_t st = h->stream;
  const int64_t NS = h->n_states;
  if (h->d_per_target.n < (size_t)NS) HIP_TRY(h->d_per_target.alloc(NS));
  if (h->d_status.n < (size_t)NS) HIP_TRY(h->d_status.alloc(NS));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = h->d_R.p; t.pi = nullptr; t.unit_off = h->d_state_off.p;
  t.gamma = 1.0f; t.eps = epsilon; t.max_abs = 0.0; t.max_sweeps = max_sweeps;
  t.Q = nullptr; t.V = nullptr; t.sweeps = nullptr; t.per_target = h->d_per_target.p; t.status = h->d_status.p;
  const size_t v_need = 2 * sizeof(float) * (size_t)h->max_S + sizeof(float) * 4 * (kDpBlock / 64);
  const bool lanes = sch == CMDP_SCHEME_JACOBI && (h->dp_kernel == 3 || h->dp_kernel == 4 || h->dp_kernel == 6 || v_need > (size_t)kLdsBudget);
  if (lanes) {
    if (int rc = diameter_lanes(h, t, 0, NS)) return rc;
  } else if (int rc = run_sweeps(h, DP_VI, true, sch, t, NS)) return rc;
  std::vector<float> per((size_t)NS);
  HIP_TRY(hipMemcpyAsync(per.data(), h->d_per_target.p, sizeof(float) * NS, hipMemcpyDeviceToHost, st));
  if (int rc = check_status(h, NS)) return rc;
  for (int b = 0; b < h->B; ++b) {
    float dmax = 0.0f;  // `diameter = 0` then max(...), diameter.py:99-105
    for (int64_t s = h->state_off[b]; s < h->state_off[b + 1]; ++s) dmax = std::max(dmax, per[(size_t)s]);
    diameter[b] = dmax;
  }
  if (per_target) std::memcpy(per_target, per.data(), sizeof(float) * NS);
  return CMDP_OK;
}

int cmdp_diameter_sparse_f64(cmdp_t* h, double epsilon, int64_t max_sweeps, double* running_max, double* diameter) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!diameter) return fail(CMDP_ERR_INVALID, "null output");
  if (max_sweeps < 1) return fail(CMDP_ERR_INVALID, "max_sweeps < 1");
  if (h->H != 0) return fail(CMDP_ERR_INVALID, "the sparse float64 diameter is the continuous setting's (horizon 0)");
  hipStream_t st = h->stream;
  const int64_t NS = h->n_states;
  if (h->d_status.n < (size_t)NS) HIP_TRY(h->d_status.alloc(NS));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.eps = epsilon; t.max_sweeps = max_sweeps; t.status = h->d_status.p;
  constexpr int kLogCap = 2048;  // sweeps between diff < 0.05 and diff < eps that can be logged per target
  std::vector<int32_t> inst, t0, cnt;
  std::vector<int64_t> vdoubles;
  for (int b = 0; b < h->B; ++b) {
    const int64_t S = h->state_off[b + 1] - h->state_off[b];
    for (int64_t x = 0; x < S; x += 64) {
      inst.push_back(b);
      t0.push_back((int32_t)x);
      cnt.push_back((int32_t)std::min<int64_t>(64, S - x));
      vdoubles.push_back(2 * S * 64);
    }
  }
  const size_t G = inst.size();
  size_t ws_cap = h->dl_ws_bytes;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ws_cap = std::min(ws_cap, free_b / 5 * 3);
  }
  std::vector<double> log_host;
  std::vector<int32_t> logn_host;
  std::vector<double> run((size_t)NS, 0.0);
  std::vector<double> D((size_t)h->B, -std::numeric_limits<double>::infinity());
  DevBuf<double> d_v, d_log;
  DevBuf<int32_t> d_logn, d_inst, d_t0, d_cnt;
  DevBuf<int64_t> d_voff;
  size_t g0 = 0;
  while (g0 < G) {
    size_t g1 = g0, doubles = 0;
    std::vector<int64_t> voff;
    while (g1 < G && (g1 == g0 || (doubles + (size_t)vdoubles[g1]) * sizeof(double) + (g1 - g0 + 1) * 64 * kLogCap * 16 <= ws_cap)) {
      voff.push_back((int64_t)doubles);
      doubles += (size_t)vdoubles[g1];
      ++g1;
    }
    const size_t n = g1 - g0;
    if (d_v.n < doubles) HIP_TRY(d_v.alloc(doubles));
    if (d_log.n < n * 64 * kLogCap * 2) HIP_TRY(d_log.alloc(n * 64 * kLogCap * 2));
    if (d_logn.n < n * 64) HIP_TRY(d_logn.alloc(n * 64));
    HIP_TRY(d_inst.upload(inst.data() + g0, n, st));
    HIP_TRY(d_t0.upload(t0.data() + g0, n, st));
    HIP_TRY(d_cnt.upload(cnt.data() + g0, n, st));
    HIP_TRY(d_voff.upload(voff.data(), n, st));
    DiamF64Args g{d_inst.p, d_t0.p, d_cnt.p, d_voff.p, d_v.p, d_log.p, d_logn.p, kLogCap};
    hipLaunchKernelGGL(k_diam_lanes_f64<8>, dim3((unsigned)n), dim3(512), 0, st, t, g);
    HIP_TRY(hipGetLastError());
    log_host.resize(n * 64 * kLogCap * 2);
    logn_host.resize(n * 64);
    HIP_TRY(hipMemcpyAsync(logn_host.data(), d_logn.p, sizeof(int32_t) * n * 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(log_host.data(), d_log.p, sizeof(double) * log_host.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // the reference's loop over the targets, in its order: the first logged sweep at which it would have stopped
    for (size_t gi = 0; gi < n; ++gi) {
      const int b = inst[g0 + gi];
      for (int l = 0; l < cnt[g0 + gi]; ++l) {
        const int32_t nl = logn_host[gi * 64 + l];
        if (nl < 0) return fail(CMDP_ERR_MAX_ITER, "target %d of instance %d did not converge within max_sweeps", t0[g0 + gi] + l, b);
        if (nl > kLogCap)
          return fail(CMDP_ERR_UNSUPPORTED, "target %d of instance %d needs more than %d sweeps between diff < 0.05 and diff < eps",
                      t0[g0 + gi] + l, b, kLogCap);
        const double* lg = log_host.data() + (gi * 64 + l) * (size_t)kLogCap * 2;
        double mx = lg[2 * (nl - 1) + 1];
        for (int32_t j = 0; j < nl; ++j) {
          const double diff = lg[2 * j], m = lg[2 * j + 1];
          if (diff < epsilon || (diff < 0.05 && m - 1 < D[(size_t)b])) { mx = m; break; }
        }
        D[(size_t)b] = std::max(D[(size_t)b], mx);
        run[(size_t)(h->state_off[b] + t0[g0 + gi] + l)] = D[(size_t)b];
      }
    }
    g0 = g1;
  }
  for (int b = 0; b < h->B; ++b) diameter[b] = D[(size_t)b];
  if (running_max) std::memcpy(running_max, run.data(), sizeof(double) * (size_t)NS);
  return CMDP_OK;
}

int cmdp_diameter_range(cmdp_t* h, double epsilon, int64_t max_sweeps, int64_t target_lo, int64_t target_hi,
                        float* per_target) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!per_target) return fail(CMDP_ERR_INVALID, "null output");
  if (max_sweeps < 1) return fail(CMDP_ERR_INVALID, "max_sweeps < 1");
  const int64_t NS = h->n_states;
  if (target_lo < 0 || target_hi > NS || target_lo > target_hi) return fail(CMDP_ERR_INVALID, "target range outside [0, %lld]", (long long)NS);
  if (target_lo == target_hi) return CMDP_OK;
  hipStream_t st = h->stream;
  if (h->d_per_target.n < (size_t)NS) HIP_TRY(h->d_per_target.alloc(NS));
  if (h->d_status.n < (size_t)NS) HIP_TRY(h->d_status.alloc(NS));
  HIP_TRY(hipMemsetAsync(h->d_status.p, 0, sizeof(int32_t) * NS, st));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = h->d_R.p; t.pi = nullptr; t.unit_off = h->d_state_off.p;
  t.gamma = 1.0f; t.eps = epsilon; t.max_abs = 0.0; t.max_sweeps = max_sweeps;
  t.per_target = h->d_per_target.p; t.status = h->d_status.p;
  if (int rc = diameter_lanes(h, t, target_lo, target_hi)) return rc;
  HIP_TRY(hipMemcpyAsync(per_target, h->d_per_target.p + target_lo, sizeof(float) * (target_hi - target_lo),
                         hipMemcpyDeviceToHost, st));
  return check_status(h, NS);
}

int cmdp_diameter_episodic(cmdp_t* h, int H, const int64_t* start_off, const int32_t* start_state,
                           const float* start_prob, double epsilon, int64_t max_sweeps, float* per_target,
                           float* diameter) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!diameter || !start_off || !start_state || !start_prob) return fail(CMDP_ERR_INVALID, "null argument");
  if (H < 2 || max_sweeps < 1) return fail(CMDP_ERR_INVALID, "H < 2 or max_sweeps < 1");
  const int B = h->B, A = h->A;
  const int64_t NS = h->n_states;
  const size_t lds = sizeof(float) * (size_t)H * h->max_S;
  if (lds > (size_t)kLdsBudget - 1024)
    return fail(CMDP_ERR_UNSUPPORTED, "H*S = %d*%d floats do not fit the LDS-resident episodic sweep", H, h->max_S);
  if (start_off[0] != 0) return fail(CMDP_ERR_INVALID, "start_off[0] != 0");
  // rows of T_epi that are filled (mdp_creation.py:118-125): layer 0 = starting states, layer h = states with
  // incoming mass in layer h-1, for h <= H-2; layer H-1 is the return to the starting states
  std::vector<int64_t> ptr((size_t)h->n_rows + 1);
  std::vector<int32_t> col((size_t)h->n_csr);
  hipStream_t st = h->stream;
  HIP_TRY(hipMemcpyAsync(ptr.data(), h->d_csr_ptr.p, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(col.data(), h->d_csr_col.p, sizeof(int32_t) * col.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<uint8_t> reach((size_t)H * NS, 0);
  for (int b = 0; b < B; ++b) {
    const int64_t so = h->state_off[b], S = h->state_off[b + 1] - so;
    uint8_t* rb = reach.data() + (size_t)H * so;
    if (start_off[b + 1] <= start_off[b]) return fail(CMDP_ERR_INVALID, "instance %d has no starting state", b);
    for (int64_t i = start_off[b]; i < start_off[b + 1]; ++i) {
      if (start_state[i] < 0 || start_state[i] >= S) return fail(CMDP_ERR_INVALID, "starting state out of range");
      rb[start_state[i]] = 1;
    }
    for (int hh = 1; hh <= H - 2; ++hh)
      for (int64_t s = 0; s < S; ++s)
        if (rb[(size_t)(hh - 1) * S + s])
          for (int a = 0; a < A; ++a) {
            const int64_t r = (so + s) * A + a;
            for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) rb[(size_t)hh * S + col[k]] = 1;
          }
  }
  DevBuf<uint8_t> d_reach;
  DevBuf<int64_t> d_soff;
  DevBuf<int32_t> d_sstate;
  DevBuf<float> d_sprob;
  HIP_TRY(d_reach.upload(reach.data(), reach.size(), st));
  HIP_TRY(d_soff.upload(start_off, B + 1, st));
  HIP_TRY(d_sstate.upload(start_state, start_off[B], st));
  HIP_TRY(d_sprob.upload(start_prob, start_off[B], st));
  if (h->d_per_target.n < (size_t)NS) HIP_TRY(h->d_per_target.alloc(NS));
  if (h->d_status.n < (size_t)NS) HIP_TRY(h->d_status.alloc(NS));
  DpTables t{};
  t.B = B; t.A = A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = h->d_R.p; t.unit_off = h->d_state_off.p; t.eps = epsilon; t.max_sweeps = max_sweeps;
  t.per_target = h->d_per_target.p; t.status = h->d_status.p;
  EpiDiamArgs e{H, d_soff.p, d_sstate.p, d_sprob.p, d_reach.p};
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_diam_episodic), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_diam_episodic, dim3((unsigned)NS), dim3(256), lds, st, t, e);
  HIP_TRY(hipGetLastError());
  std::vector<float> per((size_t)NS);
  HIP_TRY(hipMemcpyAsync(per.data(), h->d_per_target.p, sizeof(float) * NS, hipMemcpyDeviceToHost, st));
  if (int rc = check_status(h, NS)) return rc;  // synchronises; the upload buffers above die after this
  for (int b = 0; b < B; ++b) {
    float dmax = -INFINITY;  // `diameter = -np.inf`, diameter.py:203
    for (int64_t s = h->state_off[b]; s < h->state_off[b + 1]; ++s) dmax = std::max(dmax, per[(size_t)s]);
    diameter[b] = dmax;
  }
  if (per_target) std::memcpy(per_target, per.data(), sizeof(float) * NS);
  return CMDP_OK;
}

// ---- device agents -----------------------------------------------------------------------------------------
}  // extern "C"

struct cmdp_agent {
  cmdp_t* env = nullptr;
  bool continuous = false;
  QlArgs args{};
  QlcArgs cargs{};
  DevBuf<double> d_Hh, d_gamma, d_Qc, d_Qmainc, d_Vc;  // continuous agent: float64 tables
  DevBuf<double> d_ilog, d_s7;
  DevBuf<int64_t> d_qoff, d_voff;
  DevBuf<int32_t> d_N, d_mtpos;
  DevBuf<float> d_Q, d_V, d_mu, d_sigma, d_beta;
  DevBuf<uint32_t> d_mt;
  DevBuf<int8_t> d_act;
  DevBuf<double> d_rsum;   // MDPLoop._cumulative_reward per instance
  DevBuf<uint8_t> d_mask;
  DevBuf<float> d_pi;      // greedy policy [H][S][A]
  DevBuf<float> d_v0;      // packed V[0, :] of the evaluated greedy policies
  int64_t n_q = 0, n_v = 0;
};

extern "C" {

int cmdp_qlearning_create(cmdp_agent_t** out, cmdp_t* env, const int32_t* seeds, int64_t optimization_horizon, double p,
                          double c_1, double c_2, double min_at, int ucb_type) {
  if (!out || !env || !seeds) return fail(CMDP_ERR_INVALID, "null argument");
  *out = nullptr;
  if (int rc = bind(env)) return rc;
  if (!env->has_env || env->H < 1) return fail(CMDP_ERR_INVALID, "the episodic Q-learning agent needs an episodic environment handle");
  if (env->layout != CMDP_LAYOUT_CSR) return fail(CMDP_ERR_UNSUPPORTED, "agents run on the CSR layout");
  if (!(p > 0 && p < 1) || !(c_1 > 0) || !(min_at >= 0 && min_at < 0.99) || (ucb_type != 0 && ucb_type != 1) ||
      (ucb_type == 1 && !(c_2 > 0)) || optimization_horizon < 1)
    return fail(CMDP_ERR_INVALID, "hyper-parameters out of range (0<p<1, c_1>0, 0<=min_at<0.99, bernstein needs c_2>0)");
  const int B = env->B, A = env->A, H = env->H;
  cmdp_agent_t* a = new cmdp_agent;
  struct Guard { cmdp_agent_t* a; ~Guard() { delete a; } } guard{a};
  a->env = env;
  hipStream_t st = env->stream;
  std::vector<double> ilog((size_t)B), s7((size_t)B);
  std::vector<int64_t> qoff((size_t)B), voff((size_t)B);
  for (int b = 0; b < B; ++b) {
    const int64_t S = env->state_off[b + 1] - env->state_off[b];
    // self.i = np.log(n_states * n_actions * optimization_horizon / p): integer product, one division, one log
    ilog[b] = std::log((double)(S * A * optimization_horizon) / p);
    s7[b] = std::sqrt(std::pow((double)H, 7) * (double)S * (double)A);
    qoff[b] = (int64_t)H * env->state_off[b] * A;
    voff[b] = (int64_t)(H + 1) * env->state_off[b];
  }
  a->n_q = (int64_t)H * env->n_states * A;
  a->n_v = (int64_t)(H + 1) * env->n_states;
  HIP_TRY(a->d_ilog.upload(ilog.data(), B, st));
  HIP_TRY(a->d_s7.upload(s7.data(), B, st));
  HIP_TRY(a->d_qoff.upload(qoff.data(), B, st));
  HIP_TRY(a->d_voff.upload(voff.data(), B, st));
  HIP_TRY(a->d_N.alloc(a->n_q));
  HIP_TRY(a->d_Q.alloc(a->n_q));
  HIP_TRY(a->d_V.alloc(a->n_v));
  HIP_TRY(a->d_V.zero(st));
  HIP_TRY(a->d_mu.alloc(a->n_q)); HIP_TRY(a->d_mu.zero(st));
  HIP_TRY(a->d_sigma.alloc(a->n_q)); HIP_TRY(a->d_sigma.zero(st));
  HIP_TRY(a->d_beta.alloc(a->n_q)); HIP_TRY(a->d_beta.zero(st));
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(a->n_q, 256)), dim3(256), 0, st, a->d_N.p, 1, a->n_q);
  hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(a->n_q, 256)), dim3(256), 0, st, a->d_Q.p, (float)H, a->n_q);
  HIP_TRY(a->d_mt.alloc((size_t)B * 624));
  HIP_TRY(a->d_mtpos.alloc(B));
  HIP_TRY(a->d_rsum.alloc(B));
  HIP_TRY(a->d_rsum.zero(st));
  std::vector<uint32_t> useeds((size_t)B);
  for (int b = 0; b < B; ++b) useeds[b] = (uint32_t)seeds[b];
  DevBuf<uint32_t> d_seeds;
  HIP_TRY(d_seeds.upload(useeds.data(), B, st));
  hipLaunchKernelGGL(k_mt_seed_numpy, dim3(grid_for(B, 64)), dim3(64), 0, st, a->d_mt.p, a->d_mtpos.p, d_seeds.p, B);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  QlArgs& q = a->args;
  q.H = H; q.ucb = ucb_type; q.c1 = c_1; q.c2 = c_2; q.min_at = min_at; q.H3 = (double)H * H * H;
  q.i_log = a->d_ilog.p; q.sqrtH7SA = a->d_s7.p; q.q_off = a->d_qoff.p; q.v_off = a->d_voff.p;
  q.N = a->d_N.p; q.Q = a->d_Q.p; q.V = a->d_V.p; q.mu = a->d_mu.p; q.sigma = a->d_sigma.p; q.beta = a->d_beta.p;
  q.mt = a->d_mt.p; q.mt_pos = a->d_mtpos.p;
  guard.a = nullptr;
  *out = a;
  return CMDP_OK;
}

int cmdp_qlearning_continuous_create(cmdp_agent_t** out, cmdp_t* env, const int32_t* seeds, int64_t optimization_horizon,
                                     double min_at, double confidence, double span_approx_weight, double h_weight) {
  if (!out || !env || !seeds) return fail(CMDP_ERR_INVALID, "null argument");
  *out = nullptr;
  if (int rc = bind(env)) return rc;
  if (!env->has_env || env->H != 0) return fail(CMDP_ERR_INVALID, "the continuous Q-learning agent needs a continuous environment handle");
  if (env->layout != CMDP_LAYOUT_CSR) return fail(CMDP_ERR_UNSUPPORTED, "agents run on the CSR layout");
  if (!(min_at >= 0 && min_at < 0.99) || !(confidence > 0 && confidence < 1) || !(span_approx_weight > 0) || !(h_weight > 0) ||
      optimization_horizon < 1)
    return fail(CMDP_ERR_INVALID, "hyper-parameters out of range");
  const int B = env->B, A = env->A;
  cmdp_agent_t* a = new cmdp_agent;
  struct Guard { cmdp_agent_t* a; ~Guard() { delete a; } } guard{a};
  a->env = env;
  a->continuous = true;
  hipStream_t st = env->stream;
  const double T = (double)optimization_horizon;
  std::vector<double> Hh((size_t)B), gm((size_t)B);
  std::vector<int64_t> qoff((size_t)B);
  for (int b = 0; b < B; ++b) {
    const double S = (double)(env->state_off[b + 1] - env->state_off[b]);
    // get_H (q_learning.py:19-44): min(sqrt(span * T / S / A), (T / S / A / log(4 T / confidence)) ** 0.333)
    const double h1 = std::sqrt(span_approx_weight * T / S / A);
    const double h2 = std::pow(T / S / A / std::log(4 * T / confidence), 0.333);
    Hh[b] = h_weight * std::min(h1, h2);
    gm[b] = 1 - 1 / Hh[b];
    qoff[b] = env->state_off[b] * A;
  }
  a->n_q = env->n_rows;
  a->n_v = env->n_states;
  HIP_TRY(a->d_Hh.upload(Hh.data(), B, st));
  HIP_TRY(a->d_gamma.upload(gm.data(), B, st));
  HIP_TRY(a->d_qoff.upload(qoff.data(), B, st));
  HIP_TRY(a->d_N.alloc(a->n_q)); HIP_TRY(a->d_N.zero(st));
  HIP_TRY(a->d_Qc.alloc(a->n_q));
  HIP_TRY(a->d_Qmainc.alloc(a->n_q));
  HIP_TRY(a->d_Vc.alloc(a->n_v));
  // Q, Q_main, V start at H: `np.zeros(float32) + np.float64` is float64 under NEP 50
  std::vector<double> q0((size_t)a->n_q), v0((size_t)a->n_v);
  for (int b = 0; b < B; ++b) {
    for (int64_t r = env->state_off[b] * A; r < env->state_off[b + 1] * A; ++r) q0[(size_t)r] = Hh[b];
    for (int64_t s2 = env->state_off[b]; s2 < env->state_off[b + 1]; ++s2) v0[(size_t)s2] = Hh[b];
  }
  HIP_TRY(hipMemcpyAsync(a->d_Qc.p, q0.data(), sizeof(double) * a->n_q, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(a->d_Qmainc.p, q0.data(), sizeof(double) * a->n_q, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(a->d_Vc.p, v0.data(), sizeof(double) * a->n_v, hipMemcpyHostToDevice, st));
  HIP_TRY(a->d_mt.alloc((size_t)B * 624));
  HIP_TRY(a->d_mtpos.alloc(B));
  HIP_TRY(a->d_rsum.alloc(B));
  HIP_TRY(a->d_rsum.zero(st));
  std::vector<uint32_t> useeds((size_t)B);
  for (int b = 0; b < B; ++b) useeds[b] = (uint32_t)seeds[b];
  DevBuf<uint32_t> d_seeds;
  HIP_TRY(d_seeds.upload(useeds.data(), B, st));
  hipLaunchKernelGGL(k_mt_seed_numpy, dim3(grid_for(B, 64)), dim3(64), 0, st, a->d_mt.p, a->d_mtpos.p, d_seeds.p, B);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  QlcArgs& q = a->cargs;
  q.min_at = min_at > 0.009 ? min_at : 0.0;
  q.four_span = 4 * span_approx_weight;
  q.log_term = std::log(2 * T / confidence);
  q.Hh = a->d_Hh.p; q.gamma = a->d_gamma.p; q.q_off = a->d_qoff.p;
  q.N = a->d_N.p; q.Q = a->d_Qc.p; q.Qmain = a->d_Qmainc.p; q.V = a->d_Vc.p; q.mt = a->d_mt.p; q.mt_pos = a->d_mtpos.p;
  guard.a = nullptr;
  *out = a;
  return CMDP_OK;
}

int cmdp_qlearning_policy(cmdp_agent_t* a, float* pi) {
  if (!a || !pi) return fail(CMDP_ERR_INVALID, "null argument");
  cmdp_t* h = a->env;
  if (int rc = bind(h)) return rc;
  if (!a->continuous) return fail(CMDP_ERR_INVALID, "cmdp_qlearning_policy is for the continuous agent; use cmdp_qlearning_evaluate");
  hipStream_t st = h->stream;
  if (a->d_pi.n < (size_t)a->n_q) HIP_TRY(a->d_pi.alloc(a->n_q));
  hipLaunchKernelGGL(k_greedy_policy_episodic<double>, dim3(h->B), dim3(64), 0, st, h->B, h->A, 1, 1,
                     h->d_state_off.p, a->d_Qc.p, a->d_pi.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pi, a->d_pi.p, sizeof(float) * a->n_q, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

// K9F's host side (cmdp_chain.h): per instance a minimum-degree elimination order of the MDP's transition graph (union over
// the actions, symmetrised), every pivot's candidate list in the filled graph, and a schedule of rounds of up to 16 pivots
// that are pairwise non-adjacent and share at most one candidate.  Instances without a plan (a pivot with more than 64
// candidates, more than 65 535 states) keep K9.
static int build_chain_plan(cmdp_t* h) {
  h->chain_plan_built = true;
  const int B = h->B, A = h->A;
  hipStream_t st = h->stream;
  std::vector<int64_t> ptr((size_t)h->n_rows + 1);
  std::vector<int32_t> col((size_t)h->n_csr);
  std::vector<float> val((size_t)h->n_csr);
  HIP_TRY(hipMemcpyAsync(ptr.data(), h->d_csr_ptr.p, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(col.data(), h->d_csr_col.p, sizeof(int32_t) * col.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(val.data(), h->d_csr_val.p, sizeof(float) * val.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  constexpr int W = 16, NARROW = K9F_MAXC;
  std::vector<int32_t> rank((size_t)h->n_states, -1), cptr((size_t)h->n_states + B, 0), nrounds((size_t)B, -1), rptr, piv((size_t)h->n_states, 0);
  std::vector<int64_t> cbase((size_t)B, 0), rbase((size_t)B, 0);
  std::vector<uint16_t> cand;
  std::vector<std::vector<int>> plans;
  for (int b = 0; b < B; ++b) {
    const int64_t so = h->state_off[b];
    const int S = (int)(h->state_off[b + 1] - so);
    cbase[(size_t)b] = (int64_t)cand.size();
    rbase[(size_t)b] = (int64_t)rptr.size();
    if (S < 2 || S > 65535) continue;
    const int NWD = (S + 63) / 64;
    std::vector<uint64_t> g((size_t)S * NWD, 0);   // adjacency bitsets of the elimination graph
    auto setbit = [&](int u, int v) { g[(size_t)u * NWD + (v >> 6)] |= 1ull << (v & 63); };
    for (int s2 = 0; s2 < S; ++s2)
      for (int a = 0; a < A; ++a) {
        const int64_t r = (so + s2) * A + a;
        for (int64_t k = ptr[(size_t)r]; k < ptr[(size_t)r + 1]; ++k)
          if (val[(size_t)k] > 0.0f && col[(size_t)k] != s2) { setbit(s2, col[(size_t)k]); setbit(col[(size_t)k], s2); }
      }
    std::vector<int> degree((size_t)S, 0), order((size_t)S), rk((size_t)S, -1);
    std::vector<char> alive((size_t)S, 1);
    for (int u = 0; u < S; ++u)
      for (int w = 0; w < NWD; ++w) degree[(size_t)u] += __builtin_popcountll(g[(size_t)u * NWD + w]);
    std::vector<std::vector<int>> cands((size_t)S);
    bool ok = true;
    for (int step = 0; step < S; ++step) {
      int v = -1;
      for (int u = 0; u < S; ++u)
        if (alive[(size_t)u] && (v < 0 || degree[(size_t)u] < degree[(size_t)v])) v = u;   // ties: the smallest state
      order[(size_t)step] = v;
      rk[(size_t)v] = step;
      alive[(size_t)v] = 0;
      std::vector<int>& nb = cands[(size_t)step];
      for (int w = 0; w < NWD; ++w) {
        uint64_t x = g[(size_t)v * NWD + w];
        while (x) { nb.push_back(64 * w + __builtin_ctzll(x)); x &= x - 1; }
      }
      if (step < S - 1 && (int)nb.size() > NARROW) { ok = false; break; }
      for (int u : nb) {   // the neighbours become a clique; v leaves the graph
        uint64_t* gu = &g[(size_t)u * NWD];
        const uint64_t* gv = &g[(size_t)v * NWD];
        int d = 0;
        for (int w = 0; w < NWD; ++w) { gu[w] |= gv[w]; }
        gu[u >> 6] &= ~(1ull << (u & 63));
        gu[v >> 6] &= ~(1ull << (v & 63));
        for (int w = 0; w < NWD; ++w) d += __builtin_popcountll(gu[w]);
        degree[(size_t)u] = d;
      }
    }
    if (!ok) continue;
    // candidate lists as positions, ascending; bitsets of them for the conflict test of the schedule
    std::vector<uint64_t> cb((size_t)S * NWD, 0);
    for (int i = 0; i < S; ++i) {
      std::vector<int>& nb = cands[(size_t)i];
      for (int& x : nb) x = rk[(size_t)x];
      std::sort(nb.begin(), nb.end());
      for (int x : nb) cb[(size_t)i * NWD + (x >> 6)] |= 1ull << (x & 63);
    }
    // rounds: pivots whose lower-ranked neighbours are all done, pairwise sharing at most one candidate
    std::vector<int> pending((size_t)S, 0);
    for (int i = 0; i < S; ++i)
      for (int x : cands[(size_t)i]) pending[(size_t)x]++;
    std::vector<char> done((size_t)S, 0);
    std::vector<int> ready;
    for (int i = 0; i < S - 1; ++i)
      if (!pending[(size_t)i]) ready.push_back(i);
    int remaining = S - 1, nr = 0;
    int32_t* pv = piv.data() + so;
    int filled = 0;
    while (remaining > 0) {
      std::sort(ready.begin(), ready.end());
      std::vector<int> rnd, rest;
      for (int pidx : ready) {
        bool fits = (int)rnd.size() < W;
        for (size_t q = 0; fits && q < rnd.size(); ++q) {
          int common = 0;
          for (int w = 0; w < NWD; ++w) common += __builtin_popcountll(cb[(size_t)pidx * NWD + w] & cb[(size_t)rnd[q] * NWD + w]);
          fits = common <= 1;
        }
        if (fits) rnd.push_back(pidx); else rest.push_back(pidx);
      }
      if (rnd.empty()) { ok = false; break; }
      rptr.push_back(filled);
      for (int pidx : rnd) {
        pv[filled++] = pidx;
        --remaining;
        for (int x : cands[(size_t)pidx])
          if (--pending[(size_t)x] == 0 && x < S - 1) rest.push_back(x);
      }
      ready.swap(rest);
      ++nr;
    }
    if (!ok) { rptr.resize((size_t)rbase[(size_t)b]); continue; }
    rptr.push_back(filled);
    nrounds[(size_t)b] = nr;
    for (int s2 = 0; s2 < S; ++s2) rank[(size_t)(so + s2)] = rk[(size_t)s2];
    int32_t* cp = cptr.data() + so + b;
    cp[0] = 0;
    for (int i = 0; i < S; ++i) {
      for (int x : cands[(size_t)i]) cand.push_back((uint16_t)x);
      cp[i + 1] = cp[i] + (int)cands[(size_t)i].size();
    }
    h->chain_plan_any = true;
  }
  if (!h->chain_plan_any) return CMDP_OK;
  if (cand.empty()) cand.push_back(0);
  if (rptr.empty()) rptr.push_back(0);
  HIP_TRY(h->d_cf_rank.upload(rank.data(), rank.size(), st));
  HIP_TRY(h->d_cf_cptr.upload(cptr.data(), cptr.size(), st));
  HIP_TRY(h->d_cf_nrounds.upload(nrounds.data(), nrounds.size(), st));
  HIP_TRY(h->d_cf_rptr.upload(rptr.data(), rptr.size(), st));
  HIP_TRY(h->d_cf_piv.upload(piv.data(), piv.size(), st));
  HIP_TRY(h->d_cf_cbase.upload(cbase.data(), cbase.size(), st));
  HIP_TRY(h->d_cf_rbase.upload(rbase.data(), rbase.size(), st));
  HIP_TRY(h->d_cf_cand.upload(cand.data(), cand.size(), st));
  HIP_TRY(h->d_cf_slow.alloc(B));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

// K9 launch shared by cmdp_average_reward / cmdp_qlearning_average_reward: policy either as device one-hot rows
// (`d_pi`) or device actions (`d_act`); start states on the device.
// `mask_on_device`: `mask` is already a device pointer (the logged loop keeps its need-mask there)
// `copy_back` false: `avg` / `kind` (when given) are DEVICE-ACCESSIBLE buffers (page-locked host memory in the logged loop) the
// kernels write directly -- no copy kernel per row; nothing is synchronised.
static int chain_launch(cmdp_t* h, const float* d_pi, const int32_t* d_act, const int32_t* d_start, const uint8_t* mask,
                        double* avg, int32_t* kind, int32_t* n_classes, bool mask_on_device = false, bool copy_back = true,
                        hipStream_t on_stream = nullptr) {
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "the handle was created without the DP half (CSR transition matrices)");
  if (h->H != 0) return fail(CMDP_ERR_INVALID, "average rewards are defined for continuous instances (horizon 0)");
  const size_t lds = chain_lds_bytes(h->max_S, h->max_row_nnz);
  if (lds > (size_t)kLdsBudget)
    return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states (max %d successors per row) exceeds the LDS budget of K9",
                h->max_S, h->max_row_nnz);
  hipStream_t st = on_stream ? on_stream : h->stream;   // the logged loop runs the solve beside the agents' next interval
  const int B = h->B;
  if (h->d_ch_off.n < (size_t)B + 1) {
    std::vector<int64_t> off(B + 1, 0);
    for (int b = 0; b < B; ++b) {
      const int64_t S = h->state_off[b + 1] - h->state_off[b];
      off[b + 1] = off[b] + S * S;
    }
    HIP_TRY(h->d_ch_off.upload(off.data(), off.size(), st));
    HIP_TRY(hipStreamSynchronize(st));  // `off` is a local
    HIP_TRY(h->d_ch_work.alloc((size_t)off[B]));
    HIP_TRY(h->d_ch_idx.alloc((size_t)off[B]));
    HIP_TRY(h->d_ch_avg.alloc(B));
    HIP_TRY(h->d_ch_kind.alloc(B));
    HIP_TRY(h->d_ch_ncls.alloc(B));
  }
  const uint8_t* dmask = nullptr;
  if (mask && mask_on_device) {
    dmask = mask;
  } else if (mask) {
    HIP_TRY(h->d_ch_mask.upload(mask, B, st));
    dmask = h->d_ch_mask.p;
  }
  ChainArgs c{};
  c.B = B; c.A = h->A; c.max_deg = h->max_row_nnz;
  c.state_off = h->d_state_off.p; c.csr_ptr = h->d_csr_ptr.p; c.csr_col = h->d_csr_col.p; c.csr_val = h->d_csr_val.p;
  c.R = h->d_R.p; c.pi = d_pi; c.act = d_act; c.start = d_start; c.mask = dmask;
  c.work_off = h->d_ch_off.p; c.work = h->d_ch_work.p; c.work_idx = h->d_ch_idx.p;
  c.avg = (!copy_back && avg) ? avg : h->d_ch_avg.p;
  c.kind = (!copy_back && kind) ? kind : h->d_ch_kind.p;
  c.n_classes = h->d_ch_ncls.p;
  if (lds > 64 * 1024)
  {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_average_reward<16, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_average_reward<16, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  static const bool chain_debug = std::getenv("CMDP_CHAIN_DEBUG") != nullptr;
  DevBuf<long long> d_dbg;
  if (chain_debug) {
    HIP_TRY(d_dbg.alloc((size_t)B * 8));
    HIP_TRY(d_dbg.zero(st));
    c.dbg = d_dbg.p;
  }
  // K9F first (irreducible chains, fill-reducing elimination order, rounds of independent pivots); it flags the instances
  // it leaves to K9.  The reference's summation order (CMDP_OPT_CHAIN_EXACT_ORDER) keeps K9 alone.
  static const int fast_env = std::getenv("CMDP_CHAIN_FAST") ? std::atoi(std::getenv("CMDP_CHAIN_FAST")) : 1;
  if (!h->chain_exact && fast_env) {
    if (!h->chain_plan_built)
      if (int rc = build_chain_plan(h)) return rc;
    const size_t flds = chain_fast_lds_bytes(h->max_S, h->max_row_nnz, 16);
    if (h->chain_plan_any && flds <= (size_t)kLdsBudget) {
      ChainFast f{};
      f.rank = h->d_cf_rank.p; f.cptr = h->d_cf_cptr.p; f.cbase = h->d_cf_cbase.p; f.cand = h->d_cf_cand.p;
      f.nrounds = h->d_cf_nrounds.p; f.rbase = h->d_cf_rbase.p; f.rptr = h->d_cf_rptr.p; f.piv = h->d_cf_piv.p;
      f.slow = h->d_cf_slow.p;
      if (int rc = set_lds(k_chain_fast<16>, flds)) return rc;
      hipLaunchKernelGGL((k_chain_fast<16>), dim3(B), dim3(1024), flds, st, c, f);
      c.mask = h->d_cf_slow.p;
    }
  }
  if (h->chain_exact) hipLaunchKernelGGL((k_chain_average_reward<16, true>), dim3(B), dim3(1024), lds, st, c);
  else hipLaunchKernelGGL((k_chain_average_reward<16, false>), dim3(B), dim3(1024), lds, st, c);
  if (chain_debug) {
    std::vector<long long> t((size_t)B * 8);
    HIP_TRY(hipMemcpyAsync(t.data(), d_dbg.p, sizeof(long long) * t.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < 7; ++k) ph[k] = std::max(ph[k], (double)(t[(size_t)b * 8 + k + 1] - t[(size_t)b * 8 + k]) / 100.0);
    std::fprintf(stderr, "[K9 us, max over instances] A %.0f  B(tarjan) %.0f  C %.0f  D %.0f  E(gth) %.0f  F(backsub) %.0f  sum %.0f\n",
                 ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
  }
  HIP_TRY(hipGetLastError());
  if (!copy_back) return CMDP_OK;  // the caller reads d_ch_avg / d_ch_kind itself
  HIP_TRY(hipMemcpyAsync(avg, h->d_ch_avg.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(kind, h->d_ch_kind.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  if (n_classes) HIP_TRY(hipMemcpyAsync(n_classes, h->d_ch_ncls.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_average_reward(cmdp_t* h, const int32_t* actions, const int32_t* start_states, const uint8_t* mask, double* avg,
                        int32_t* kind, int32_t* n_classes) {
  if (int rc = bind(h)) return rc;
  if (!actions || !start_states || !avg || !kind) return fail(CMDP_ERR_INVALID, "null argument");
  for (int b = 0; b < h->B; ++b) {
    const int64_t S = h->state_off[b + 1] - h->state_off[b];
    if (start_states[b] < 0 || start_states[b] >= S) return fail(CMDP_ERR_INVALID, "start state of instance %d out of range", b);
    if (mask && !mask[b]) continue;
    for (int64_t s = h->state_off[b]; s < h->state_off[b + 1]; ++s)
      if (actions[s] < 0 || actions[s] >= h->A) return fail(CMDP_ERR_INVALID, "action of state %lld out of range", (long long)s);
  }
  hipStream_t st = h->stream;
  HIP_TRY(h->d_ch_act.upload(actions, (size_t)h->n_states, st));
  HIP_TRY(h->d_ch_start.upload(start_states, (size_t)h->B, st));
  return chain_launch(h, nullptr, h->d_ch_act.p, h->d_ch_start.p, mask, avg, kind, n_classes);
}

int cmdp_qlearning_average_reward(cmdp_agent_t* a, const uint8_t* mask, double* avg, int32_t* kind) {
  if (!a || !avg || !kind) return fail(CMDP_ERR_INVALID, "null argument");
  cmdp_t* h = a->env;
  if (int rc = bind(h)) return rc;
  if (!a->continuous) return fail(CMDP_ERR_INVALID, "cmdp_qlearning_average_reward is for the continuous agent");
  hipStream_t st = h->stream;
  if (a->d_pi.n < (size_t)a->n_q) HIP_TRY(a->d_pi.alloc(a->n_q));
  hipLaunchKernelGGL(k_greedy_policy_episodic<double>, dim3(h->B), dim3(64), 0, st, h->B, h->A, 1, 1,
                     h->d_state_off.p, a->d_Qc.p, a->d_pi.p);
  HIP_TRY(hipGetLastError());
  return chain_launch(h, a->d_pi.p, nullptr, h->d_cur.p, mask, avg, kind, nullptr);
}

int cmdp_set_observation_table(cmdp_t* h, const float* table, int32_t F, int time_indexed) {
  if (int rc = bind(h)) return rc;
  if (!h->has_env) return fail(CMDP_ERR_INVALID, "handle was created without the sampler half");
  if (!table || F < 1) return fail(CMDP_ERR_INVALID, "null table or F < 1");
  if (time_indexed && h->H < 1) return fail(CMDP_ERR_INVALID, "a time-indexed table needs an episodic handle");
  hipStream_t st = h->stream;
  const size_t n = (size_t)(time_indexed ? h->H : 1) * (size_t)h->n_states * (size_t)F;
  HIP_TRY(h->d_obs_table.upload(table, n, st));
  HIP_TRY(h->d_obs_out.alloc((size_t)h->B * F));
  HIP_TRY(h->d_n_obs.alloc((size_t)h->B));
  HIP_TRY(h->d_n_obs.zero(st));
  HIP_TRY(hipStreamSynchronize(st));
  h->obs_F = F;
  h->obs_time_indexed = time_indexed ? 1 : 0;
  return CMDP_OK;
}

int cmdp_observe_noise(cmdp_t* h, int kind, double scale, double df, const float* chol, float* obs) {
  if (int rc = bind(h)) return rc;
  if (!obs) return fail(CMDP_ERR_INVALID, "null output");
  if (h->obs_F < 1) return fail(CMDP_ERR_INVALID, "no observation table: call cmdp_set_observation_table first");
  if (kind < CMDP_NOISE_NONE || kind > CMDP_NOISE_STUDENT_T_CORRELATED) return fail(CMDP_ERR_INVALID, "unknown noise kind %d", kind);
  const bool noisy = kind >= CMDP_NOISE_GAUSSIAN_CORRELATED || (kind == CMDP_NOISE_GAUSSIAN && scale > 0.0);
  if (noisy && h->rng_mode != CMDP_RNG_PHILOX)
    return fail(CMDP_ERR_UNSUPPORTED, "device noise needs CMDP_RNG_PHILOX (the reference-exact noise stream is host side)");
  const bool correlated = kind == CMDP_NOISE_GAUSSIAN_CORRELATED || kind == CMDP_NOISE_STUDENT_T_CORRELATED;
  if (correlated && !chol) return fail(CMDP_ERR_INVALID, "correlated noise needs the Cholesky factor");
  if ((kind == CMDP_NOISE_STUDENT_T || kind == CMDP_NOISE_STUDENT_T_CORRELATED) && !(df > 0.0))
    return fail(CMDP_ERR_INVALID, "Student-t noise needs df > 0");
  const size_t lds = correlated ? sizeof(double) * (size_t)h->obs_F : 0;
  if (lds > (size_t)kLdsBudget - 1024) return fail(CMDP_ERR_UNSUPPORTED, "observations of %d elements: the normals do not fit LDS", h->obs_F);
  hipStream_t st = h->stream;
  if (correlated) HIP_TRY(h->d_obs_chol.upload(chol, (size_t)h->obs_F * h->obs_F, st));
  EmitArgs e{};
  e.B = h->B; e.F = h->obs_F; e.H = h->H; e.time_indexed = h->obs_time_indexed;
  e.state_off = h->d_state_off.p; e.table = h->d_obs_table.p; e.cur = h->d_cur.p; e.hstep = h->d_h.p;
  e.key = h->d_key.p; e.n_obs = h->d_n_obs.p; e.scale = scale; e.kind = kind; e.df = df;
  e.chol = correlated ? h->d_obs_chol.p : nullptr; e.out = h->d_obs_out.p;
  if (int rc = set_lds(k_emit, lds)) return rc;
  hipLaunchKernelGGL(k_emit, dim3(h->B), dim3(256), lds, st, e);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(obs, h->d_obs_out.p, sizeof(float) * (size_t)h->B * h->obs_F, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_observe(cmdp_t* h, double noise_scale, float* obs) {
  return cmdp_observe_noise(h, noise_scale > 0.0 ? CMDP_NOISE_GAUSSIAN : CMDP_NOISE_NONE, noise_scale, 0.0, nullptr, obs);
}

}  // extern "C"

// Mixing time of one instance by matrix powers (see cmdp.h).  d(t) = max_s TV(P^t(s, .), pi) is non-increasing in t, so:
// square A_k = P^(2^k) until d(2^K) <= threshold (then t_mix lies in (2^(K-1), 2^K]), walk back down multiplying the
// stored powers in (binary search on t, one dgemm per bit), and finish the last 2^r steps -- r = the lowest power still
// held when HBM ran out of S x S buffers -- with sparse steps.
static int mixing_time_dense(cmdp_t* h, int b, const int64_t* d_cptr, const int32_t* d_crow, const double* d_cval,
                             const double* d_stat, double threshold, int64_t max_steps, int64_t* t_mix, double* tv_at) {
  hipStream_t st = h->stream;
  const int64_t so = h->state_off[b];
  const int S = (int)(h->state_off[b + 1] - so);
  const size_t bytes = sizeof(double) * (size_t)S * (size_t)S;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  int cap = (int)std::min<size_t>(24, (free_b / 10 * 9) / bytes);
  if (const char* e = std::getenv("CMDP_MIX_MAX_BUFFERS")) cap = std::min(cap, std::atoi(e));  // tests: force the sparse tail
  if (cap < 3)
    return fail(CMDP_ERR_UNSUPPORTED, "mixing time of %d states needs three %zu-byte matrices, %zu bytes are free", S, bytes, free_b);
  std::vector<DevBuf<double>> pool((size_t)cap);
  std::vector<int> free_list;
  for (int i = 0; i < cap; ++i) free_list.push_back(i);
  auto take = [&](int* idx) -> int {
    *idx = free_list.back();
    free_list.pop_back();
    if (!pool[(size_t)*idx].p) {
      if (hipError_t e = pool[(size_t)*idx].alloc((size_t)S * S); e != hipSuccess)
        return fail(CMDP_ERR_HIP, "mixing-time matrix of %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    return CMDP_OK;
  };
  DevBuf<unsigned long long> d_tv;
  HIP_TRY(d_tv.alloc(1));
  MixDense m{S, so, d_cptr, d_crow, d_cval, d_stat, d_tv.p};
  auto tv_of = [&](const double* X, double* out) -> int {
    HIP_TRY(d_tv.zero(st));
    hipLaunchKernelGGL(k_mixd_tv, dim3((unsigned)S), dim3(256), 0, st, m, X);
    HIP_TRY(hipGetLastError());
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, d_tv.p, sizeof bits, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(out, &bits, sizeof(double));
    return CMDP_OK;
  };
  rocblas_handle blas = nullptr;
  if (rocblas_create_handle(&blas) != rocblas_status_success) return fail(CMDP_ERR_HIP, "rocblas_create_handle failed");
  struct BlasGuard { rocblas_handle h; ~BlasGuard() { rocblas_destroy_handle(h); } } guard{blas};
  if (rocblas_set_stream(blas, st) != rocblas_status_success) return fail(CMDP_ERR_HIP, "rocblas_set_stream failed");
  // row-major Z = X * Y  <=>  column-major Z^T = Y^T * X^T
  auto matmul = [&](const double* X, const double* Y, double* Z) -> int {
    const double one = 1.0, zero = 0.0;
    if (rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_none, S, S, S, &one, Y, S, X, S, &zero, Z, S) !=
        rocblas_status_success)
      return fail(CMDP_ERR_HIP, "rocblas_dgemm failed");
    return CMDP_OK;
  };
  *t_mix = -1;
  if (tv_at) *tv_at = 0.0;
  std::vector<int> power;  // power[i] = pool index of A_(r+i)
  int r = 0, idx = 0;
  if (int rc = take(&idx)) return rc;
  HIP_TRY(hipMemsetAsync(pool[(size_t)idx].p, 0, bytes, st));
  hipLaunchKernelGGL(k_mixd_build, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, m, pool[(size_t)idx].p);
  HIP_TRY(hipGetLastError());
  power.push_back(idx);
  double d = 0.0;
  if (int rc = tv_of(pool[(size_t)idx].p, &d)) return rc;
  if (tv_at) *tv_at = d;
  if (d <= threshold) {
    *t_mix = 1;
    return CMDP_OK;
  }
  // ---- doubling: A_(K) = A_(K-1)^2 until mixed ------------------------------------------------------------------
  int K = 0;  // the newest power is A_K, t = 2^K, not mixed
  while (true) {
    if (((int64_t)1 << K) >= max_steps) return CMDP_OK;  // not mixed at 2^K >= max_steps: -1
    if (free_list.empty()) {  // drop the lowest power: the final stretch of sparse steps doubles
      free_list.push_back(power.front());
      power.erase(power.begin());
      ++r;
    }
    if (int rc = take(&idx)) return rc;
    const double* prev = pool[(size_t)power.back()].p;
    if (int rc = matmul(prev, prev, pool[(size_t)idx].p)) return rc;
    power.push_back(idx);
    ++K;
    if (int rc = tv_of(pool[(size_t)idx].p, &d)) return rc;
    if (std::getenv("CMDP_MIX_DEBUG")) std::fprintf(stderr, "[mixing] t = 2^%d: max TV %.6g (%d powers held from 2^%d)\n", K, d, (int)power.size(), r);
    if (d <= threshold) break;
    if (tv_at) *tv_at = d;
  }
  // ---- binary search on t: cur = P^lo (not mixed), lo + 2^k' mixed for the current k' ----------------------------------
  int64_t lo = (int64_t)1 << (K - 1);
  int cand = power.back();              // A_K: mixed, its buffer is scratch from here on
  int cur = power[power.size() - 2];    // A_(K-1)
  double d_hi = d;                      // TV at the smallest t known to be mixed
  for (int k = K - 2; k >= r; --k) {
    const double* Ak = pool[(size_t)power[(size_t)(k - r)]].p;
    if (int rc = matmul(pool[(size_t)cur].p, Ak, pool[(size_t)cand].p)) return rc;
    if (int rc = tv_of(pool[(size_t)cand].p, &d)) return rc;
    if (d > threshold) {
      std::swap(cur, cand);  // A_(K-1)'s buffer becomes scratch: no lower step multiplies by it again
      lo += (int64_t)1 << k;
      if (tv_at) *tv_at = d;
    } else {
      d_hi = d;
    }
  }
  // ---- the last 2^r steps one at a time ---------------------------------------------------------------------------------
  const int64_t last = (int64_t)1 << r;
  for (int64_t i = 1; i <= last; ++i) {
    if (i == last) {  // lo + 2^r is known to be mixed
      d = d_hi;
    } else {
      HIP_TRY(d_tv.zero(st));
      hipLaunchKernelGGL(k_mixd_step, dim3((unsigned)S), dim3(256), 0, st, m, pool[(size_t)cur].p, pool[(size_t)cand].p);
      HIP_TRY(hipGetLastError());
      unsigned long long bits = 0;
      HIP_TRY(hipMemcpyAsync(&bits, d_tv.p, sizeof bits, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      std::memcpy(&d, &bits, sizeof(double));
      std::swap(cur, cand);
    }
    if (d <= threshold) {
      if (lo + i <= max_steps) {
        *t_mix = lo + i;
        if (tv_at) *tv_at = d;
      }
      return CMDP_OK;
    }
    if (tv_at) *tv_at = d;
  }
  return fail(CMDP_ERR_HIP, "mixing-time search lost its bracket (total variation is not monotone?)");
}

extern "C" {

int cmdp_mixing_time(cmdp_t* h, const float* pi, const double* stationary, double threshold, int64_t max_steps,
                     int64_t* t_mix, double* tv_at) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!stationary || !t_mix) return fail(CMDP_ERR_INVALID, "null argument");
  if (!(threshold > 0.0) || max_steps < 1) return fail(CMDP_ERR_INVALID, "threshold <= 0 or max_steps < 1");
  const bool fits_lds = sizeof(double) * (size_t)h->max_S <= (size_t)kLdsBudget - 1024;
  if (h->mixing_path == 2 && !fits_lds)
    return fail(CMDP_ERR_UNSUPPORTED, "a row of X (%d states, float64) does not fit LDS: the stepping path cannot run", h->max_S);
  const bool dense_path = h->mixing_path == 1 || (h->mixing_path == 0 && (!fits_lds || h->max_S > 1024));
  hipStream_t st = h->stream;
  const int B = h->B, A = h->A;
  const int64_t NS = h->n_states, NR = h->n_rows;
  std::vector<int64_t> ptr((size_t)NR + 1);
  std::vector<int32_t> col((size_t)h->n_csr);
  std::vector<float> val((size_t)h->n_csr);
  HIP_TRY(hipMemcpyAsync(ptr.data(), h->d_csr_ptr.p, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(col.data(), h->d_csr_col.p, sizeof(int32_t) * col.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(val.data(), h->d_csr_val.p, sizeof(float) * val.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // P[s, j] = sum_a pi[s, a] * T[s, a, j] (float64, actions in order), then its CSC with predecessors in index order
  std::vector<int64_t> cptr((size_t)NS + 1, 0), xoff((size_t)B + 1, 0);
  std::vector<int32_t> crow;
  std::vector<double> cval;
  {
    std::vector<std::vector<std::pair<int32_t, double>>> incoming((size_t)NS);
    std::vector<double> rowacc;
    std::vector<int32_t> touched;
    for (int b = 0; b < B; ++b) {
      const int64_t so = h->state_off[b], S = h->state_off[b + 1] - so;
      xoff[b + 1] = xoff[b] + S * S;
      rowacc.assign((size_t)S, 0.0);
      for (int64_t s = 0; s < S; ++s) {
        touched.clear();
        for (int a = 0; a < A; ++a) {
          const int64_t r = (so + s) * A + a;
          const double w = pi ? (double)pi[r] : 1.0 / A;
          if (w == 0.0) continue;
          for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) {
            if (rowacc[col[k]] == 0.0) touched.push_back(col[k]);
            rowacc[col[k]] += w * (double)val[k];
          }
        }
        // the float32 probabilities of a row sum to 1 only within ~1e-7: without this, X_t would lose or gain that much
        // mass per step (1e-2 over the 1e5 steps a slow chain needs).  The chain is DEFINED with normalised rows.
        double rowsum = 0.0;
        for (int32_t j : touched) rowsum += rowacc[j];
        for (int32_t j : touched) {
          if (rowacc[j] != 0.0) incoming[(size_t)(so + j)].push_back({(int32_t)s, rowacc[j] / rowsum});
          rowacc[j] = 0.0;
        }
      }
    }
    for (int64_t j = 0; j < NS; ++j) {
      cptr[j + 1] = cptr[j] + (int64_t)incoming[j].size();
      for (auto& e : incoming[j]) { crow.push_back(e.first); cval.push_back(e.second); }  // s ascending by construction
    }
  }
  constexpr int CH = 256;
  DevBuf<int64_t> d_cptr, d_xoff;
  DevBuf<int32_t> d_crow;
  DevBuf<double> d_cval, d_stat, d_X, d_Xn;
  DevBuf<unsigned long long> d_dl;
  HIP_TRY(d_cptr.upload(cptr.data(), cptr.size(), st));
  HIP_TRY(d_xoff.upload(xoff.data(), xoff.size(), st));
  HIP_TRY(d_crow.upload(crow.data(), crow.size(), st));
  HIP_TRY(d_cval.upload(cval.data(), cval.size(), st));
  HIP_TRY(d_stat.upload(stationary, (size_t)NS, st));
  if (dense_path) {
    for (int b = 0; b < B; ++b)
      if (int rc = mixing_time_dense(h, b, d_cptr.p, d_crow.p, d_cval.p, d_stat.p, threshold, max_steps, t_mix + b,
                                     tv_at ? tv_at + b : nullptr))
        return rc;
    return CMDP_OK;
  }
  HIP_TRY(d_X.alloc((size_t)xoff[B]));
  HIP_TRY(d_Xn.alloc((size_t)xoff[B]));
  HIP_TRY(d_dl.alloc((size_t)B * CH));
  MixArgs m{};
  m.B = B; m.state_off = h->d_state_off.p; m.x_off = d_xoff.p; m.csc_ptr = d_cptr.p; m.csc_row = d_crow.p;
  m.csc_val = d_cval.p; m.stationary = d_stat.p; m.X = d_X.p; m.Xnew = d_Xn.p; m.dlist = d_dl.p; m.chunk = CH;
  hipLaunchKernelGGL(k_mix_init, dim3((unsigned)NS), dim3(256), 0, st, m);
  HIP_TRY(hipGetLastError());
  const size_t lds = sizeof(double) * (size_t)h->max_S;
  if (int rc = set_lds(k_mix_step, lds)) return rc;
  std::vector<unsigned long long> dl((size_t)B * CH);
  std::vector<char> found((size_t)B, 0);
  for (int b = 0; b < B; ++b) { t_mix[b] = -1; if (tv_at) tv_at[b] = 0.0; }
  int remaining = B;
  for (int64_t t0 = 0; t0 < max_steps && remaining > 0; t0 += CH) {
    const int n = (int)std::min<int64_t>(CH, max_steps - t0);
    HIP_TRY(hipMemsetAsync(d_dl.p, 0, sizeof(unsigned long long) * dl.size(), st));
    for (int k = 0; k < n; ++k) {
      m.k = k;
      hipLaunchKernelGGL(k_mix_step, dim3((unsigned)NS), dim3(256), lds, st, m);
      std::swap(m.X, m.Xnew);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dl.data(), d_dl.p, sizeof(unsigned long long) * dl.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      if (found[b]) continue;
      for (int k = 0; k < n; ++k) {
        double tv;
        std::memcpy(&tv, &dl[(size_t)b * CH + k], sizeof(double));
        if (tv <= threshold) {
          t_mix[b] = t0 + k + 1;  // X after (t0 + k + 1) steps
          if (tv_at) tv_at[b] = tv;
          found[b] = 1;
          --remaining;
          break;
        }
        if (tv_at) tv_at[b] = tv;
      }
    }
  }
  return CMDP_OK;
}

int cmdp_qlearning_destroy(cmdp_agent_t* a) {
  if (!a) return CMDP_OK;
  if (a->env) { (void)hipSetDevice(a->env->device); (void)hipStreamSynchronize(a->env->stream); }
  delete a;
  return CMDP_OK;
}

// the interaction kernel of `n_steps` steps on the handle's stream (no synchronisation, no copies)
static int ql_launch(cmdp_agent_t* a, int64_t n_steps, const uint8_t* dmask, int8_t* d_actions, int resume, double* cum_host) {
  cmdp_t* h = a->env;
  hipStream_t st = h->stream;
  const dim3 grid(grid_for(h->B, 256)), block(256);
  const RewardCache rc = h->rcache();
#define QL_LAUNCH(K, ARGS) hipLaunchKernelGGL(K, grid, block, 0, st, h->env(), ARGS, n_steps, dmask, d_actions, a->d_rsum.p, rc, resume, cum_host)
  if (h->reward_cache) {
    if (a->continuous) QL_LAUNCH(k_qlearn_continuous<true>, a->cargs);
    else if (a->args.ucb == 0) QL_LAUNCH((k_qlearn_episodic<0, true>), a->args);
    else QL_LAUNCH((k_qlearn_episodic<1, true>), a->args);
  } else {
    if (a->continuous) QL_LAUNCH(k_qlearn_continuous<false>, a->cargs);
    else if (a->args.ucb == 0) QL_LAUNCH((k_qlearn_episodic<0, false>), a->args);
    else QL_LAUNCH((k_qlearn_episodic<1, false>), a->args);
  }
#undef QL_LAUNCH
  HIP_TRY(hipGetLastError());
  return CMDP_OK;
}
// With reference-exact reward caches the call returns with the stream idle (instances park, the host fills their blocks,
// the kernel is relaunched); otherwise nothing is synchronised.
// `cum_host` (nullable): page-locked host array that receives the running reward sums at the end of the launch.
static int ql_enqueue_run(cmdp_agent_t* a, int64_t n_steps, const uint8_t* dmask, int8_t* d_actions, double* cum_host = nullptr) {
  if (int rc = visits_room(a->env, n_steps)) return rc;
  if (a->env->reward_cache)
    return rc_drive(a->env, [&](int resume) -> int { return ql_launch(a, n_steps, dmask, d_actions, resume, cum_host); });
  return ql_launch(a, n_steps, dmask, d_actions, 0, cum_host);
}

int cmdp_qlearning_run(cmdp_agent_t* a, int64_t n_steps, const uint8_t* train_mask, int8_t* actions_trace,
                       double* reward_sum) {
  if (!a) return fail(CMDP_ERR_INVALID, "null agent");
  cmdp_t* h = a->env;
  if (int rc = bind(h)) return rc;
  if (n_steps < 0) return fail(CMDP_ERR_INVALID, "n_steps < 0");
  bool any = false;
  if (int rc = any_needs_reset(h, &any)) return rc;
  if (any) return fail(CMDP_ERR_NEEDS_RESET, "the environment needs reset() before the agent can run");
  hipStream_t st = h->stream;
  const size_t NB = (size_t)n_steps * h->B;
  if (actions_trace && a->d_act.n < NB) HIP_TRY(a->d_act.alloc(NB));
  const uint8_t* dmask = nullptr;
  if (train_mask) {
    HIP_TRY(a->d_mask.upload(train_mask, h->B, st));
    dmask = a->d_mask.p;
  }
  if (int rc = ql_enqueue_run(a, n_steps, dmask, actions_trace ? a->d_act.p : nullptr)) return rc;
  if (actions_trace) HIP_TRY(hipMemcpyAsync(actions_trace, a->d_act.p, NB, hipMemcpyDeviceToHost, st));
  if (reward_sum) HIP_TRY(hipMemcpyAsync(reward_sum, a->d_rsum.p, sizeof(double) * h->B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

// greedy policy of the agents' Q tables -> episodic policy evaluation -> V[0, :] packed into a->d_v0 (no synchronisation)
static int ql_enqueue_evaluate(cmdp_agent_t* a, float* v0_out = nullptr, int32_t* snap = nullptr) {
  cmdp_t* h = a->env;
  if (a->continuous) return fail(CMDP_ERR_INVALID, "cmdp_qlearning_evaluate is for the episodic agent; use cmdp_qlearning_policy");
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "the environment handle was created without the DP half");
  hipStream_t st = h->stream;
  const int H = h->H;
  const size_t lds = 2 * sizeof(float) * (size_t)h->max_S;
  if (lds > (size_t)kLdsBudget) return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states does not fit LDS", h->max_S);
  if (a->d_pi.n < (size_t)a->n_q) HIP_TRY(a->d_pi.alloc(a->n_q));
  hipLaunchKernelGGL(k_greedy_policy_episodic<float>, dim3(h->B), dim3(64), 0, st, h->B, h->A, H, H,
                     h->d_state_off.p, a->d_Q.p, a->d_pi.p);
  const size_t nq = (size_t)(H + 1) * h->n_rows, nv = (size_t)(H + 1) * h->n_states;
  if (h->d_Q.n < nq) HIP_TRY(h->d_Q.alloc(nq));
  if (h->d_V.n < nv) HIP_TRY(h->d_V.alloc(nv));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = h->d_R.p; t.pi = a->d_pi.p;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_episodic<DP_PE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_episodic<DP_PE>), dim3(h->B), dim3(kDpBlock), lds, st, t, H, h->d_Q.p, h->d_V.p);
  HIP_TRY(hipGetLastError());
  // V[0, :] of instance b sits at (H+1)*state_off[b]: pack
  if (a->d_v0.n < (size_t)h->n_states) HIP_TRY(a->d_v0.alloc(h->n_states));
  hipLaunchKernelGGL(k_gather_v0, dim3(h->B), dim3(256), 0, st, h->B, H, h->d_state_off.p, h->d_V.p, v0_out ? v0_out : a->d_v0.p,
                     h->d_last_start.p, h->d_prev_start.p, h->d_h.p, snap);
  HIP_TRY(hipGetLastError());
  return CMDP_OK;
}

int cmdp_qlearning_evaluate(cmdp_agent_t* a, float* V0) {
  if (!a || !V0) return fail(CMDP_ERR_INVALID, "null argument");
  cmdp_t* h = a->env;
  if (int rc = bind(h)) return rc;
  if (int rc = ql_enqueue_evaluate(a)) return rc;
  HIP_TRY(hipMemcpyAsync(V0, a->d_v0.p, sizeof(float) * (size_t)h->n_states, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CMDP_OK;
}

int cmdp_qlearning_run_logged(cmdp_agent_t* a, const cmdp_loop_desc* d, int64_t n_logs, int64_t* steps, double* values,
                              uint8_t* kinds, int64_t* last_training_step, uint8_t* is_training) {
  using namespace cmdp_tracker;
  if (!a || !d || !steps || !values || !kinds) return fail(CMDP_ERR_INVALID, "null argument");
  cmdp_t* h = a->env;
  if (int rc = bind(h)) return rc;
  const int B = h->B;
  const int64_t T = d->n_steps, log_every = d->log_every;
  if (T < 1) return fail(CMDP_ERR_INVALID, "n_steps < 1");
  std::vector<int64_t> log_ts;
  if (log_every > 0)
    for (int64_t t = log_every; t < T; t += log_every) log_ts.push_back(t);
  if (n_logs != (int64_t)log_ts.size() + 1)
    return fail(CMDP_ERR_INVALID, "n_logs must be %lld for %lld steps logged every %lld", (long long)log_ts.size() + 1, (long long)T,
                (long long)log_every);
  if (!d->base_val || !d->base_kind) return fail(CMDP_ERR_INVALID, "baseline average rewards missing");
  const bool episodic = !a->continuous;
  if (episodic && (!d->opt0 || !d->worst0 || !d->start_pos || !d->start_prob || d->kmax < 1))
    return fail(CMDP_ERR_INVALID, "episodic baselines missing");
  if (!episodic)
    for (int b = 0; b < B; ++b)
      if (!(d->base_val[3 * b] - d->base_val[3 * b + 1] > 0.0002))   // agent_mdp_interaction.py:379-382
        return fail(CMDP_ERR_INVALID, "instance %d: optimal and worst average reward are closer than 0.0002", b);
  // refused before anything is stepped or reset: a caller that falls back to another loop must find the agent untouched
  if (!episodic && log_every > 0 && chain_lds_bytes(h->max_S, h->max_row_nnz) > (size_t)kLdsBudget)
    return fail(CMDP_ERR_UNSUPPORTED, "instance with %d states (max %d successors per row) exceeds the LDS budget of K9",
                h->max_S, h->max_row_nnz);
  if (!episodic && !h->has_dp) return fail(CMDP_ERR_INVALID, "the handle was created without the DP half (CSR transition matrices)");
  hipStream_t st = h->stream;
  const int64_t NS = h->n_states;
  Tracker tr;
  tr.init(B, d->n_check, d->base_val, d->base_kind);
  EpisodicInputs ein{h->H, d->opt0, d->worst0, d->start_pos, d->start_prob, d->kmax};
  // `cum` twice: the next interval's kernel may already be writing its sums while the host reads this row's
  PinnedBuf<double> cum[2], avg;
  PinnedBuf<float> v0;
  PinnedBuf<int32_t> snap, akind;
  PinnedBuf<uint8_t> mask, need;
  if (int rc = cum[0].alloc(B)) return rc;
  if (int rc = cum[1].alloc(B)) return rc;
  if (int rc = avg.alloc(B)) return rc;
  if (int rc = v0.alloc((size_t)NS)) return rc;
  if (int rc = snap.alloc((size_t)3 * B)) return rc;
  if (int rc = akind.alloc(B)) return rc;
  if (int rc = mask.alloc(B)) return rc;
  if (int rc = need.alloc(B)) return rc;
  if (a->d_mask.n < (size_t)B) HIP_TRY(a->d_mask.alloc(B));
  if (h->d_ch_mask.n < (size_t)B) HIP_TRY(h->d_ch_mask.alloc(B));
  std::vector<int64_t> start_abs((size_t)B);
  for (int b = 0; b < B; ++b) { mask.p[b] = 1; cum[0].p[b] = cum[1].p[b] = 0.0; if (last_training_step) last_training_step[b] = -1; }
  HIP_TRY(hipMemcpyAsync(a->d_mask.p, mask.p, B, hipMemcpyHostToDevice, st));
  // MDPLoop.run: visitation counts cleared, environment reset (agent_mdp_interaction.py:219-224)
  if (int rc = cmdp_reset_visits(h)) return rc;
  if (int rc = cmdp_reset(h, nullptr, nullptr)) return rc;
  const auto t_start = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };

  // The rows of the run.  Row i: `n_run` steps whose reward sums the row logs (the reference reads `_cumulative_reward` at
  // step t BEFORE adding that step's reward), then -- inside the loop -- step t itself, whose update the logged policy
  // already contains; then the evaluation of the agents' greedy policies.  log_every == 1 leaves no step between two rows:
  // the sum through step t-1 is then what the previous row's single step left.
  struct Row { int64_t t, n_run, n_since; bool in_loop; };
  std::vector<Row> plan;
  {
    int64_t done = 0, n_since = 0;
    for (int64_t tl : log_ts) {
      if (tl - done > 0) n_since += tl - done;
      plan.push_back(Row{tl, tl - done, n_since, true});
      done = tl + 1;
      n_since = 1;
    }
    if (T - done > 0) n_since += T - done;
    plan.push_back(Row{T - 1, T - done, n_since, false});
  }
  const size_t n_rows = plan.size();

  // Two things overlap with the evaluation of row i and with the host's work on it: the agents' NEXT interval (same stream,
  // enqueued before the host waits) and, for the continuous agent, the stationary-distribution solve itself (second stream,
  // on a snapshot of the policy and of the current states).  Both need the training mask of the next interval before row
  // i's result is known.  The mask changes in two ways only: (a) an instance freezes -- `after_log` requires the last
  // n_check normalised regrets, this row's included, to be ~0 and t > 0.2 T, so a row whose n_check - 1 predecessors are not
  // all ~0 cannot freeze anything, and the host knows that BEFORE the row; (b) the time limit -- rows closer than a few
  // seconds to it are not run ahead.  Rows that could change the mask are processed in order, as before: the results are
  // the same either way (tests/test_gpu_mdploop.py holds both against the step-by-step loop).
  // CMDP_LOGGED_PIPELINE = 0 | 1 decides; unset: on, except under rocprofv3 (its tool library in LD_PRELOAD or its ROCPROF*
  // variables in the environment), whose queue interception faults on a stream that never drains
  static const bool pipeline_env = [] {
    if (const char* e = std::getenv("CMDP_LOGGED_PIPELINE")) return std::atoi(e) != 0;
    const char* pre = std::getenv("LD_PRELOAD");
    if (pre && std::strstr(pre, "rocprof")) return false;
    for (char** ev = environ; ev && *ev; ++ev)
      if (!std::strncmp(*ev, "ROCPROF", 7)) return false;
    return true;
  }();
  static const bool block_env = std::getenv("CMDP_SYNC_MODE") && !std::strcmp(std::getenv("CMDP_SYNC_MODE"), "block");
  // CMDP_LOGGED_DRAIN_EVERY = n: the stream is drained completely every n rows (0: never).  Only for runs under rocprofv3,
  // whose queue interception faulted in hipLaunchKernel's argument copy once a stream stayed busy long enough for the
  // runtime's kernel-argument pool to wrap (ROCm 7.2; the same run outside the profiler is fine).
  static const int drain_env = std::getenv("CMDP_LOGGED_DRAIN_EVERY") ? std::atoi(std::getenv("CMDP_LOGGED_DRAIN_EVERY")) : 0;
  for (int i = 0; i < 2; ++i)
    if (!h->ev_row[i]) HIP_TRY(hipEventCreateWithFlags(&h->ev_row[i], hipEventDisableTiming | (block_env ? hipEventBlockingSync : 0)));
  hipStream_t sx = st;
  if (!episodic && pipeline_env) {
    if (!h->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    sx = h->aux_stream;
    if (h->d_cur_snap.n < (size_t)B) HIP_TRY(h->d_cur_snap.alloc(B));
  }
  const double atol = episodic ? 1e-4 : 1e-5;
  // rows within reach of the time limit run in order: "within reach" follows the longest row seen so far (a park round of
  // the reward caches, a throttled host, sixteen batches sharing the GPU can make one row take seconds)
  double longest_row = 0.0, last_row_done = elapsed();
  bool limit_passed_while_ahead = false;
  auto mask_may_change = [&](const Row& r) -> bool {
    if (limit_passed_while_ahead || d->max_time - elapsed() < std::max(5.5, 3.0 * longest_row)) return true;
    for (int b = 0; b < B; ++b) {
      const Instance& x = tr.inst[(size_t)b];
      if (!episodic && !x.training && !x.cached) return true;   // the cached evaluation is taken at this row: `need` changes
      if (x.training && may_freeze(x, tr.n_check, r.t, T, atol)) return true;
    }
    return false;
  };

  auto enqueue_interval = [&](size_t i) -> int {
    const Row& r = plan[i];
    double* c = cum[i & 1].p;
    if (r.n_run > 0) {   // the kernel leaves the sums in `c` (page-locked) itself
      if (int rc = ql_enqueue_run(a, r.n_run, a->d_mask.p, nullptr, c)) return rc;
    } else {
      HIP_TRY(hipMemcpyAsync(c, a->d_rsum.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    }
    if (r.in_loop)
      if (int rc = ql_enqueue_run(a, 1, a->d_mask.p, nullptr)) return rc;
    return CMDP_OK;
  };

  // evaluation of the agents' current greedy policies; everything a row reads comes back WITHOUT copy kernels: the kernels
  // write into page-locked host memory directly (V[0, :], the start states and in-episode times, the average rewards and
  // their kinds, the reward sums) and read the evaluation mask from it
  auto enqueue_eval = [&]() -> int {
    if (episodic) {
      if (int rc = ql_enqueue_evaluate(a, v0.p, snap.p)) return rc;
    } else {
      continuous_need(tr, need.p);
      bool any = false;
      for (int b = 0; b < B; ++b) any = any || need.p[b];
      if (any) {
        if (a->d_pi.n < (size_t)a->n_q) HIP_TRY(a->d_pi.alloc(a->n_q));
        hipLaunchKernelGGL(k_greedy_policy_episodic<double>, dim3(B), dim3(64), 0, st, B, h->A, 1, 1, h->d_state_off.p,
                           a->d_Qc.p, a->d_pi.p);
        HIP_TRY(hipGetLastError());
        const int32_t* start = h->d_cur.p;
        if (sx != st) {
          HIP_TRY(hipMemcpyAsync(h->d_cur_snap.p, h->d_cur.p, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st));
          start = h->d_cur_snap.p;
          HIP_TRY(hipEventRecord(h->ev_row[0], st));
          HIP_TRY(hipStreamWaitEvent(sx, h->ev_row[0], 0));
        }
        if (int rc = chain_launch(h, a->d_pi.p, nullptr, start, need.p, avg.p, akind.p, nullptr, true, false, sx)) return rc;
        HIP_TRY(hipEventRecord(h->ev_row[1], sx));
        return CMDP_OK;
      }
    }
    HIP_TRY(hipEventRecord(h->ev_row[1], st));
    return CMDP_OK;
  };

  // CMDP_LOGGED_DEBUG=1: where the host thread's time went (stderr, one line per call)
  static const bool debug_env = std::getenv("CMDP_LOGGED_DEBUG") != nullptr;
  double t_enq = 0.0, t_wait = 0.0, t_host = 0.0;
  int64_t n_ahead = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a0, std::chrono::steady_clock::time_point a1) { return std::chrono::duration<double>(a1 - a0).count(); };
  if (int rc = enqueue_interval(0)) return rc;
  for (size_t i = 0; i < n_rows; ++i) {
    const Row& r = plan[i];
    const auto c0 = now();
    if (int rc = enqueue_eval()) return rc;
    const bool ahead = pipeline_env && i + 1 < n_rows && !(drain_env > 0 && (i + 1) % (size_t)drain_env == 0) && !mask_may_change(r);
    if (ahead)
      if (int rc = enqueue_interval(i + 1)) return rc;
    n_ahead += ahead;
    const auto c1 = now();
    HIP_TRY(hipEventSynchronize(h->ev_row[1]));
    const auto c2 = now();
    {
      const double e = elapsed();
      longest_row = std::max(longest_row, e - last_row_done);
      last_row_done = e;
    }
    t_enq += secs(c0, c1);
    t_wait += secs(c1, c2);
    const double sps = (double)r.t / std::max(elapsed(), 1e-9);
    double* val = values + i * N_COLUMNS * B;
    uint8_t* knd = kinds + i * N_COLUMNS * B;
    const double* cm = cum[i & 1].p;
    steps[i] = r.t;
    if (episodic) {
      // the reference logs step t before the reset that follows a termination: if step t ended an episode (in-episode
      // time back at 0), its `last_starting_node` is still the start of the episode that ended
      for (int b = 0; b < B; ++b)   // snap: last_start | prev_start | hstep
        start_abs[(size_t)b] = h->state_off[b] + ((snap.p[2 * B + b] == 0 && r.in_loop) ? snap.p[B + b] : snap.p[b]);
      episodic_update(tr, ein, r.t, T, v0.p, start_abs.data(), cm, r.n_since, r.in_loop, sps, val, knd);
    } else {
      continuous_update(tr, r.t, T, need.p, avg.p, akind.p, cm, r.n_since, r.in_loop, sps, val, knd);
    }
    if (!r.in_loop) { t_host += secs(c2, now()); break; }
    bool changed = false;
    // `_limit_exceeded` (agent_mdp_interaction.py:172-177) for the batch.  Should the limit pass on a row whose successor is
    // already running (a row far longer than any before it), the freeze is recorded at the next row -- which then runs in order
    bool out_of_time = d->max_time - elapsed() < 0.5;
    if (out_of_time && ahead) { limit_passed_while_ahead = true; out_of_time = false; }
    for (int b = 0; b < B; ++b) {
      if (out_of_time && tr.inst[(size_t)b].training) {
        tr.inst[(size_t)b].training = false;
        if (last_training_step) last_training_step[b] = r.t;
      }
      const uint8_t m = tr.inst[(size_t)b].training ? 1 : 0;
      changed = changed || m != mask.p[b];
      mask.p[b] = m;
    }
    if (changed && ahead)   // cannot happen (see mask_may_change); a wrong row must not be returned silently
      return fail(CMDP_ERR_HIP, "logged loop: the training mask changed at step %lld although the next interval was already running",
                  (long long)r.t);
    if (changed) HIP_TRY(hipMemcpyAsync(a->d_mask.p, mask.p, B, hipMemcpyHostToDevice, st));
    const auto c3 = now();
    t_host += secs(c2, c3);
    if (!ahead) {
      if (drain_env > 0 && (i + 1) % (size_t)drain_env == 0) HIP_TRY(hipStreamSynchronize(st));
      if (int rc = enqueue_interval(i + 1)) return rc;
      t_enq += secs(c3, now());
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (debug_env)
    std::fprintf(stderr, "[logged loop] %d instances, %zu rows (%lld with the next interval started early): enqueue %.2f s, waiting for the device %.2f s, "
                 "host row work %.2f s\n", B, n_rows, (long long)n_ahead, t_enq, t_wait, t_host);
  if (is_training)
    for (int b = 0; b < B; ++b) is_training[b] = tr.inst[(size_t)b].training ? 1 : 0;
  h->known_reset = true;
  return CMDP_OK;
}

// The indicator code alone, on inputs given by the caller (no device involved): what cmdp_qlearning_run_logged does with
// the values it reads back at every logging step.  Exists so that the CPU test suite can hold the C++ tracker against the
// reference's own indicator code (golden G15).
int cmdp_tracker_replay(const cmdp_loop_desc* d, int32_t B, int32_t episodic, const int64_t* state_off, int64_t n_logs,
                        const int64_t* log_steps, const uint8_t* in_loop, const int64_t* n_since, const double* cum_reward,
                        const float* V0, const int64_t* start_state, const double* avg, const int32_t* avg_kind,
                        double* values, uint8_t* kinds, uint8_t* is_training) {
  using namespace cmdp_tracker;
  if (!d || !log_steps || !in_loop || !n_since || !cum_reward || !values || !kinds) return fail(CMDP_ERR_INVALID, "null argument");
  Tracker tr;
  tr.init(B, d->n_check, d->base_val, d->base_kind);
  std::vector<uint8_t> need((size_t)B);
  std::vector<int64_t> start_abs((size_t)B);
  const int64_t NS = episodic ? state_off[B] : 0;
  EpisodicInputs ein{d->horizon, d->opt0, d->worst0, d->start_pos, d->start_prob, d->kmax};
  for (int64_t i = 0; i < n_logs; ++i) {
    double* val = values + (size_t)i * N_COLUMNS * B;
    uint8_t* knd = kinds + (size_t)i * N_COLUMNS * B;
    if (episodic) {
      for (int b = 0; b < B; ++b) start_abs[(size_t)b] = state_off[b] + start_state[(size_t)i * B + b];
      episodic_update(tr, ein, log_steps[i], d->n_steps, V0 + (size_t)i * NS, start_abs.data(), cum_reward + (size_t)i * B,
                      n_since[i], in_loop[i] != 0, 0.0, val, knd);
    } else {
      continuous_need(tr, need.data());
      continuous_update(tr, log_steps[i], d->n_steps, need.data(), avg + (size_t)i * B, avg_kind + (size_t)i * B,
                        cum_reward + (size_t)i * B, n_since[i], in_loop[i] != 0, 0.0, val, knd);
    }
    if (is_training)
      for (int b = 0; b < B; ++b) is_training[(size_t)i * B + b] = tr.inst[(size_t)b].training ? 1 : 0;
  }
  return CMDP_OK;
}

int cmdp_greedy_policy_episodic(cmdp_t* h, int H, int q_layers, const float* Q, float* pi) {
  if (int rc = bind(h)) return rc;
  if (!Q || !pi || H < 1 || q_layers < H) return fail(CMDP_ERR_INVALID, "bad argument");
  hipStream_t st = h->stream;
  DevBuf<float>&d_q = h->d_gp_q, &d_p = h->d_gp_p;
  HIP_TRY(d_q.upload(Q, (size_t)q_layers * h->n_rows, st));
  HIP_TRY(d_p.alloc((size_t)H * h->n_rows));
  hipLaunchKernelGGL(k_greedy_policy_episodic<float>, dim3(h->B), dim3(64), 0, st, h->B, h->A, H, q_layers,
                     h->d_state_off.p, d_q.p, d_p.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pi, d_p.p, sizeof(float) * (size_t)H * h->n_rows, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_qlearning_tables(cmdp_agent_t* a, float* Q, int32_t* N) {
  if (!a) return fail(CMDP_ERR_INVALID, "null agent");
  if (int rc = bind(a->env)) return rc;
  hipStream_t st = a->env->stream;
  if (Q && a->continuous)  // the continuous agent's tables are float64: Q is read as `double*` here
    HIP_TRY(hipMemcpyAsync(Q, a->d_Qc.p, sizeof(double) * a->n_q, hipMemcpyDeviceToHost, st));
  else if (Q) HIP_TRY(hipMemcpyAsync(Q, a->d_Q.p, sizeof(float) * a->n_q, hipMemcpyDeviceToHost, st));
  if (N) HIP_TRY(hipMemcpyAsync(N, a->d_N.p, sizeof(int32_t) * a->n_q, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_gth(int count, const int32_t* dims, const double* mats, double* out) {
  if (count < 0 || (count > 0 && (!dims || !mats || !out))) return fail(CMDP_ERR_INVALID, "bad argument");
  if (count == 0) return CMDP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CMDP_ERR_NO_DEVICE, "no HIP device visible");
  std::vector<int64_t> moff((size_t)count), xoff((size_t)count);
  int64_t mt = 0, xt = 0;
  for (int m = 0; m < count; ++m) {
    if (dims[m] < 1 || dims[m] > 46340) return fail(CMDP_ERR_INVALID, "chain %d has dimension %d", m, dims[m]);
    moff[m] = mt; xoff[m] = xt;
    mt += (int64_t)dims[m] * dims[m];
    xt += dims[m];
  }
  // Workspace that is reused across calls (no per-call hipMalloc/hipFree: both synchronise the device) and deliberately
  // leaked at exit; calls are serialised by a mutex.  One workspace per device.
  struct Ws { DevBuf<int64_t> moff, xoff; DevBuf<int32_t> dims; DevBuf<double> mats, x; hipStream_t st = nullptr; };
  static std::mutex mu;
  static std::map<int, Ws*>* all = new std::map<int, Ws*>;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  Ws*& ws = (*all)[dev];
  if (!ws) {
    ws = new Ws;
    HIP_TRY(hipStreamCreateWithFlags(&ws->st, hipStreamNonBlocking));
  }
  DevBuf<int64_t>&d_moff = ws->moff, &d_xoff = ws->xoff;
  DevBuf<int32_t>& d_dims = ws->dims;
  DevBuf<double>&d_mats = ws->mats, &d_x = ws->x;
  hipStream_t st = ws->st;
  HIP_TRY(d_moff.upload(moff.data(), count, st));
  HIP_TRY(d_xoff.upload(xoff.data(), count, st));
  HIP_TRY(d_dims.upload(dims, count, st));
  HIP_TRY(d_mats.upload(mats, (size_t)mt, st));
  HIP_TRY(d_x.alloc((size_t)xt));
  hipLaunchKernelGGL(k_gth, dim3(count), dim3(256), 0, st, d_moff.p, d_dims.p, d_xoff.p, d_mats.p, d_x.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_x.p, sizeof(double) * (size_t)xt, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

int cmdp_value_norm(cmdp_t* h, const float* V, float* out) {
  if (int rc = bind(h)) return rc;
  if (!h->has_dp) return fail(CMDP_ERR_INVALID, "handle was created without the DP half");
  if (!V || !out) return fail(CMDP_ERR_INVALID, "null argument");
  hipStream_t st = h->stream;
  HIP_TRY(h->d_V.upload(V, h->n_states, st));
  if (h->d_Ev.n < (size_t)h->n_rows) HIP_TRY(h->d_Ev.alloc(h->n_rows));
  if (h->d_out.n < (size_t)h->B) HIP_TRY(h->d_out.alloc(h->B));
  DpTables t{};
  t.B = h->B; t.A = h->A; t.state_off = h->d_state_off.p; t.csr_ptr = h->d_csr_ptr.p; t.csr_col = h->d_csr_col.p;
  t.csr_val = h->d_csr_val.p; t.R = h->d_R.p;
  hipLaunchKernelGGL(k_value_norm, dim3(h->B), dim3(256), 0, st, t, h->d_V.p, h->d_Ev.p, h->d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, h->d_out.p, sizeof(float) * h->B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CMDP_OK;
}

}  // extern "C"
