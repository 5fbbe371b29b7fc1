// cmdp_tracker.h -- host side of the logged interaction loop: the 18 performance indicators of MDPLoop for the B
// instances of a batch (reference colosseum/experiment/agent_mdp_interaction.py:304-578, colosseum/experiment/
// indicators.py:29-45), in C++ so that a logging step costs microseconds of host time (cmdp_qlearning_run_logged drives a
// whole run from one C call; with the Python tracker the C4 benchmark spent its time between kernels).
//
// The reference computes these numbers with Python / numpy SCALARS, and under NEP 50 the type of every intermediate
// (Python float = "weak", np.float32, np.float64) decides the rounding of every operation.  `Num` is one such scalar with
// its kind; `bin_op` promotes and rounds like numpy does.  The Python twin of this file is
// colosseum_amd/experiment/vector_tracker.py (same rules on arrays); both are pinned by golden G15, which holds what the
// reference's own indicator code makes of synthetic inputs -- values, numpy types and the training freeze.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

namespace cmdp_tracker {

enum { WEAK = 0, F32 = 1, F64 = 2 };

struct Num {
  double v = 0.0;
  int k = WEAK;
};

inline double r32(double x) { return (double)(float)x; }

// a <op> b with numpy's promotion: the result is float32 when no operand is float64 and at least one is float32 (a weak
// operand is first converted to float32), else float64 / weak.  A float32 result computed in double and rounded once is
// the correctly rounded float32 result for + - * / (53 >= 2 * 24 + 2).
inline Num bin_op(Num a, Num b, char op) {
  const int k = std::max(a.k, b.k);
  double x = a.v, y = b.v;
  if (k == F32) {
    if (a.k == WEAK) x = r32(x);
    if (b.k == WEAK) y = r32(y);
  }
  double r = op == '+' ? x + y : op == '-' ? x - y : op == '*' ? x * y : x / y;
  if (k == F32) r = r32(r);
  return Num{r, k};
}
inline Num operator+(Num a, Num b) { return bin_op(a, b, '+'); }
inline Num operator-(Num a, Num b) { return bin_op(a, b, '-'); }
inline Num operator*(Num a, Num b) { return bin_op(a, b, '*'); }
inline Num operator/(Num a, Num b) { return bin_op(a, b, '/'); }
inline Num weak(double x) { return Num{x, WEAK}; }

// np.round(x, 5) in the scalar's own type; a Python float comes back as np.float64
inline Num round5(Num a) {
  if (a.k == F32) {
    const double m = std::nearbyint(r32(a.v * 1e5));
    return Num{r32(m / 1e5), F32};
  }
  return Num{std::nearbyint(a.v * 1e5) / 1e5, F64};
}

// np.isclose(0, y, atol=atol) (rtol 1e-5) evaluated in float32 or float64
inline bool isclose_zero_y(double y, bool f32, double atol) {
  if (f32) {
    const float yy = (float)y;
    const float lhs = std::fabs(0.0f - yy);
    const float rhs = (float)atol + (float)((float)1e-5 * std::fabs(yy));
    return (lhs <= rhs && std::isfinite(yy)) || yy == 0.0f;
  }
  return (std::fabs(0.0 - y) <= atol + 1e-5 * std::fabs(y) && std::isfinite(y)) || y == 0.0;
}
// np.isclose(x, 0): |x| <= 1e-8 (+ rtol * 0), in x's type
inline bool isclose_x_zero(double x, bool f32) {
  if (f32) return std::fabs((float)x) <= (float)1e-8;
  return std::fabs(x) <= 1e-8;
}

// order of the logged columns (the sorted indicator names without "steps"; steps_per_second is wall clock)
enum Column {
  C_CUM_EXPECTED_REWARD = 0, C_CUM_REGRET, C_CUM_REWARD, C_NORM_CUM_EXPECTED_REWARD, C_NORM_CUM_REGRET, C_NORM_CUM_REWARD,
  C_OPT_CUM_EXPECTED_REWARD, C_OPT_NORM_CUM_EXPECTED_REWARD, C_RAND_CUM_EXPECTED_REWARD, C_RAND_CUM_REGRET,
  C_RAND_NORM_CUM_EXPECTED_REWARD, C_RAND_NORM_CUM_REGRET, C_STEPS_PER_SECOND, C_WORST_CUM_EXPECTED_REWARD, C_WORST_CUM_REGRET,
  C_WORST_NORM_CUM_EXPECTED_REWARD, C_WORST_NORM_CUM_REGRET, N_COLUMNS
};

struct Instance {
  Num opt, worst, rand;                 // average rewards of the three baseline policies
  Num span, regret_random, norm_regret_random, regret_worst, norm_regret_worst;
  Num cum_regret, norm_cum_regret, cum_expected_reward;
  bool training = true;
  std::deque<Num> ring;                 // latest normalised regrets
  // continuous setting: the first evaluation after the freeze is kept
  bool cached = false;
  Num c_r, c_nr, c_avg;
};

class Tracker {
 public:
  int B = 0, n_check = 10;
  std::vector<Instance> inst;

  void init(int B_, int n_check_, const double* base_val, const int32_t* base_kind) {
    B = B_;
    n_check = n_check_;
    inst.assign((size_t)B, Instance{});
    for (int b = 0; b < B; ++b) {
      Instance& x = inst[(size_t)b];
      x.opt = Num{base_val[3 * b + 0], base_kind[3 * b + 0]};
      x.worst = Num{base_val[3 * b + 1], base_kind[3 * b + 1]};
      x.rand = Num{base_val[3 * b + 2], base_kind[3 * b + 2]};
      x.span = x.opt - x.worst;
      x.regret_random = x.opt - x.rand;
      x.norm_regret_random = x.regret_random / x.span;
      x.regret_worst = x.span;
      x.norm_regret_worst = x.regret_worst / x.span;
      x.cum_regret = x.norm_cum_regret = x.cum_expected_reward = weak(0.0);
    }
  }

  // _accumulate_and_log for instance b: writes the rounded row (N_COLUMNS values + kinds)
  void accumulate_and_log(int b, int64_t t, Num regret, Num nregret, Num agent_avg, double cum_reward, int64_t n_since,
                          double steps_per_second, double* out_val, uint8_t* out_kind) {
    Instance& x = inst[(size_t)b];
    const Num t1 = weak((double)(t + 1)), tt = weak((double)t), ns = weak((double)n_since);
    x.cum_regret = x.cum_regret + regret * ns;
    x.norm_cum_regret = x.norm_cum_regret + nregret * ns;
    x.cum_expected_reward = x.cum_expected_reward + agent_avg * ns;
    const Num cr = weak(cum_reward);
    const Num rnd = x.rand * t1, wst = x.worst * t1, opt = x.opt * t1;
    auto norm = [&](Num c) { return (c - tt * x.worst) / x.span; };
    Num cols[N_COLUMNS];
    cols[C_CUM_REGRET] = x.cum_regret;
    cols[C_CUM_REWARD] = cr;
    cols[C_CUM_EXPECTED_REWARD] = x.cum_expected_reward;
    cols[C_NORM_CUM_REGRET] = x.norm_cum_regret;
    cols[C_NORM_CUM_REWARD] = norm(cr);
    cols[C_NORM_CUM_EXPECTED_REWARD] = norm(x.cum_expected_reward);
    cols[C_RAND_CUM_REGRET] = x.regret_random * t1;
    cols[C_RAND_CUM_EXPECTED_REWARD] = rnd;
    cols[C_RAND_NORM_CUM_REGRET] = x.norm_regret_random * t1;
    cols[C_RAND_NORM_CUM_EXPECTED_REWARD] = norm(rnd);
    cols[C_WORST_CUM_REGRET] = x.regret_worst * t1;
    cols[C_WORST_CUM_EXPECTED_REWARD] = wst;
    cols[C_WORST_NORM_CUM_REGRET] = x.norm_regret_worst * t1;
    cols[C_WORST_NORM_CUM_EXPECTED_REWARD] = norm(wst);
    cols[C_OPT_CUM_EXPECTED_REWARD] = opt;
    cols[C_OPT_NORM_CUM_EXPECTED_REWARD] = norm(opt);
    cols[C_STEPS_PER_SECOND] = weak(steps_per_second);
    for (int c = 0; c < N_COLUMNS; ++c) {
      const Num r = round5(cols[c]);
      out_val[c] = r.v;
      out_kind[c] = (uint8_t)r.k;
    }
  }

  // agent_mdp_interaction.py:265-288: ring of the latest normalised regrets and the optimality freeze
  void after_log(int b, int64_t t, int64_t T, Num nregret, double atol) {
    Instance& x = inst[(size_t)b];
    x.ring.push_back(nregret);
    if ((int)x.ring.size() > n_check) x.ring.pop_front();
    if ((int)x.ring.size() == n_check && (double)t > 0.2 * (double)T && x.training) {
      bool all32 = true;
      for (const Num& r : x.ring) all32 = all32 && r.k == F32;
      bool close = true;
      for (const Num& r : x.ring) close = close && isclose_zero_y(r.v, all32, atol);
      if (close && isclose_x_zero(nregret.v, nregret.k == F32)) x.training = false;
    }
  }
};

// Could the row at step t freeze this (training) instance?  `after_log` needs t > 0.2 T and the LAST n_check normalised regrets,
// the row's own included, close to zero: so not unless the latest n_check - 1 entries already are.  Conservative in the one
// thing that is not known before the row (whether all n_check entries are float32, which selects the comparison): an entry
// counts as close if it is under either.  Used by the logged loop to decide whether the next interval may start before the
// row's result is known.
inline bool may_freeze(const Instance& x, int n_check, int64_t t, int64_t T, double atol) {
  if (!((double)t > 0.2 * (double)T)) return false;
  const int need = n_check - 1;
  if ((int)x.ring.size() < need) return false;
  for (int k = 0; k < need; ++k) {
    const Num& r = x.ring[x.ring.size() - 1 - (size_t)k];
    if (!isclose_zero_y(r.v, true, atol) && !isclose_zero_y(r.v, false, atol)) return false;
  }
  return true;
}

// Episodic regrets (agent_mdp_interaction.py:534-578, indicators.py:29-45).
struct EpisodicInputs {
  int H = 0;
  const float* opt0 = nullptr;     // V*[0], flat
  const float* worst0 = nullptr;   // V_worst[0], flat
  const int64_t* start_pos = nullptr;   // [B][kmax] flat positions of the start states in state-index order (padding repeats one)
  const double* start_prob = nullptr;   // [B][kmax] probabilities (0 for padding)
  int kmax = 0;
};

inline void episodic_update(Tracker& tr, const EpisodicInputs& in, int64_t t, int64_t T, const float* V0, const int64_t* start_abs,
                            const double* cum_reward, int64_t n_since, bool in_loop, double steps_per_second,
                            double* out_val /*[N_COLUMNS][B]*/, uint8_t* out_kind) {
  const float Hf = (float)in.H;
  for (int b = 0; b < tr.B; ++b) {
    const int64_t s = start_abs[b];
    const float Rs = std::max(in.opt0[s] - V0[s], 0.0f);
    const float minimal = in.opt0[s] - in.worst0[s];
    const float regret = Rs / Hf;
    const float nr = tr.inst[(size_t)b].training ? (regret / minimal) * Hf : Rs / minimal;  // cached form once frozen (:562-566)
    double epi = 0.0;
    for (int j = 0; j < in.kmax; ++j) epi = epi + (double)V0[in.start_pos[(size_t)b * in.kmax + j]] * in.start_prob[(size_t)b * in.kmax + j];
    const Num agent_avg = Num{epi, F64} / weak((double)in.H);
    double val[N_COLUMNS];
    uint8_t kind[N_COLUMNS];
    tr.accumulate_and_log(b, t, Num{(double)regret, F32}, Num{(double)nr, F32}, agent_avg, cum_reward[b], n_since, steps_per_second,
                          val, kind);
    for (int c = 0; c < N_COLUMNS; ++c) {
      out_val[(size_t)c * tr.B + b] = val[c];
      out_kind[(size_t)c * tr.B + b] = kind[c];
    }
    if (in_loop) tr.after_log(b, t, T, Num{(double)nr, F32}, 1e-4);
  }
}

// Continuous regrets (agent_mdp_interaction.py:510-532).  avg / avg_kind: average reward of the current greedy policy for
// the instances with need[b] != 0 (need = training or not yet cached -- `continuous_need` below).
inline void continuous_need(const Tracker& tr, uint8_t* need) {
  for (int b = 0; b < tr.B; ++b) need[b] = (tr.inst[(size_t)b].training || !tr.inst[(size_t)b].cached) ? 1 : 0;
}

inline void continuous_update(Tracker& tr, int64_t t, int64_t T, const uint8_t* need, const double* avg, const int32_t* avg_kind,
                              const double* cum_reward, int64_t n_since, bool in_loop, double steps_per_second, double* out_val,
                              uint8_t* out_kind) {
  for (int b = 0; b < tr.B; ++b) {
    Instance& x = tr.inst[(size_t)b];
    Num a = x.c_avg, r = x.c_r, nr = x.c_nr;
    if (need[b]) {
      a = Num{avg[b], avg_kind[b] ? F32 : F64};
      r = x.opt - a;
      const bool close = r.k == F32 ? std::fabs((float)r.v) <= (float)1e-3 : std::fabs(r.v) <= 1e-3;  // np.isclose(r, 0.0, atol=1e-3)
      if (close) r = weak(0.0);
      if (r.v < 0) r = weak(0.0);
      nr = r / x.span;
    }
    if (!x.training && !x.cached) {  // first evaluation after the freeze is kept for the rest of the run
      x.c_r = r; x.c_nr = nr; x.c_avg = a;
      x.cached = true;
    }
    double val[N_COLUMNS];
    uint8_t kind[N_COLUMNS];
    tr.accumulate_and_log(b, t, r, nr, a, cum_reward[b], n_since, steps_per_second, val, kind);
    for (int c = 0; c < N_COLUMNS; ++c) {
      out_val[(size_t)c * tr.B + b] = val[c];
      out_kind[(size_t)c * tr.B + b] = kind[c];
    }
    if (in_loop) tr.after_log(b, t, T, nr, 1e-5);
  }
}

}  // namespace cmdp_tracker
