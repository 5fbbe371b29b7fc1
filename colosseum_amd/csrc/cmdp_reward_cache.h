// cmdp_reward_cache.h -- reference-exact stochastic rewards for BATCHES (CMDP_FLAG_REWARD_CACHE).
//
// The reference keeps, per visited (node, action, next_node) triple, a FIFO cache of 5000 samples drawn from the MDP's
// single numpy stream `self._rng` at the moment the triple is first needed and again whenever its cache runs dry
// (colosseum/mdp/base.py:1187-1207: `get_reward_distribution(...).rvs(5000, random_state=self._rng).tolist()`, `pop(0)`).
// Which triple draws next depends on the trajectory, and the number of MT19937 words a block consumes depends on the
// Beta parameters (rejection sampling), so the stream is sequential per instance by construction.
//
// Split used here: the DEVICE walks the trajectory and serves rewards from per-triple blocks in HBM; when an instance
// needs a block that does not exist or is used up it PARKS -- the transition is committed (successor, visit counts,
// sampler stream), the (entry, previous state, action) of the unfinished step is saved, the instance appends itself to a
// park list and its lane stops.  The HOST then draws the 5000 samples of every parked instance from that instance's own
// numpy stream (this file: MT19937 + numpy's LEGACY distributions, the code path of `RandomState.beta`, one task per
// instance on a process-wide thread pool), installs the blocks and relaunches; a resumed lane first completes its saved
// step.  Instances never wait for each other's streams, and the order of draws inside one instance is the reference's.
//
// The legacy samplers follow numpy/random/src/legacy/legacy-distributions.c (frozen since numpy 1.17):
// legacy_gauss (polar method with the cached second variate), legacy_standard_exponential (-log(1 - U)),
// legacy_standard_gamma (shape 1: exponential; shape < 1: Ahrens-Dieter rejection; shape > 1: Marsaglia-Tsang on
// legacy_gauss), legacy_beta (Johnk for a, b <= 1, otherwise Ga / (Ga + Gb)).  They use libm log/exp/pow/sqrt, which is
// why they run on the host (device libm differs in the last ulp).  tests/test_reward_cache.py holds them against
// numpy.random.RandomState.beta draw for draw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define CMDP_RC_BLOCK 5000  // samples per cache fill (base.py:1199)

// ---- device side -------------------------------------------------------------------------------------------------
struct RewardCache {
  const int32_t* canon;   // [E] entry -> first entry of its row with the same successor (the triple's representative), global index
  double** blk;           // [E] block of the representative entry (nullptr: never filled)
  int32_t* pos;           // [E] next sample of the block (CMDP_RC_BLOCK: used up)
  int32_t* pend_e;        // [B] global entry of the unfinished step, -1: none
  int32_t* pend_prev;     // [B] state the unfinished step left
  int32_t* pend_act;      // [B] its action
  int32_t* park_count;    // [1]
  int32_t* park_list;     // [B] instances parked by the current launch
  long long* left;        // [B] steps the instance still owes to the current call
};

// reward of entry e: true = `rraw` holds it (deterministic entries: their loc; Beta entries: the next cached sample)
__device__ __forceinline__ bool rc_fetch(const uint8_t* __restrict__ rkind, const RewardCache& rc, int64_t e, double& rraw) {
  if (rkind[e] != 1) return true;  // deterministic(loc).rvs draws nothing and returns loc (utils/miscellanea.py:259-270)
  const int64_t c = rc.canon[e];
  const double* p = rc.blk[c];
  const int32_t k = rc.pos[c];
  if (p == nullptr || k >= CMDP_RC_BLOCK) return false;
  rraw = p[k];
  rc.pos[c] = k + 1;
  return true;
}

__device__ __forceinline__ void rc_park(const RewardCache& rc, int b, int64_t e, int32_t prev, int action) {
  rc.pend_e[b] = (int32_t)e;
  rc.pend_prev[b] = prev;
  rc.pend_act[b] = action;
  const int i = atomicAdd(rc.park_count, 1);
  rc.park_list[i] = b;
}

// installs `n` freshly drawn blocks: stage[j][0..5000) -> dst[j], block pointer and position of entry ent[j]; resets the
// park counter for the relaunch
__global__ void __launch_bounds__(256) k_rc_install(int n, const double* __restrict__ stage, double* const* __restrict__ dst,
                                                    const int32_t* __restrict__ ent, RewardCache rc) {
  const int j = blockIdx.x;
  if (j >= n) return;
  double* d = dst[j];
  const double* s = stage + (size_t)j * CMDP_RC_BLOCK;
  for (int i = threadIdx.x; i < CMDP_RC_BLOCK; i += blockDim.x) d[i] = s[i];
  if (threadIdx.x == 0) {
    rc.blk[ent[j]] = d;
    rc.pos[ent[j]] = 0;
    if (j == 0) *rc.park_count = 0;
  }
}

// ---- host side: numpy's legacy RandomState stream ---------------------------------------------------------------------
namespace cmdp_rc {

// The twist and the tempering of a whole 624-word block, written so that the compiler vectorises them (both loops carry
// no dependence shorter than 227 words); the AVX2 clone is chosen once per process.  One block serves ~95 Beta draws, and
// the word-at-a-time form (twist amortised + tempering per word) was a third of a draw's cost.
#if !defined(__HIP_DEVICE_COMPILE__)
#define CMDP_MT_BLOCK_BODY                                                                               \
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;                             \
  for (int i = 0; i < 624 - 397; ++i) {                                                                  \
    const uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);                                          \
    key[i] = key[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);                                          \
  }                                                                                                      \
  for (int i = 624 - 397; i < 623; ++i) {                                                                \
    const uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);                                          \
    key[i] = key[i - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);                                          \
  }                                                                                                      \
  {                                                                                                      \
    const uint32_t y = (key[623] & UPPER) | (key[0] & LOWER);                                            \
    key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);                                            \
  }
#define CMDP_MT_TEMPER_BODY                                                                              \
  for (int i = 0; i < 624; ++i) {                                                                        \
    uint32_t y = key[i];                                                                                 \
    y ^= y >> 11;                                                                                        \
    y ^= (y << 7) & 0x9d2c5680u;                                                                         \
    y ^= (y << 15) & 0xefc60000u;                                                                        \
    y ^= y >> 18;                                                                                        \
    out[i] = y;                                                                                          \
  }
inline void mt_twist_generic(uint32_t* __restrict__ key) { CMDP_MT_BLOCK_BODY }
inline void mt_temper_generic(const uint32_t* __restrict__ key, uint32_t* __restrict__ out) { CMDP_MT_TEMPER_BODY }
__attribute__((target("avx2"))) inline void mt_twist_avx2(uint32_t* __restrict__ key) { CMDP_MT_BLOCK_BODY }
__attribute__((target("avx2"))) inline void mt_temper_avx2(const uint32_t* __restrict__ key, uint32_t* __restrict__ out) { CMDP_MT_TEMPER_BODY }
#undef CMDP_MT_BLOCK_BODY
#undef CMDP_MT_TEMPER_BODY
inline bool mt_have_avx2() {
  static const bool v = __builtin_cpu_supports("avx2");
  return v;
}
#endif

struct NumpyStream {  // `RandomState.get_state()`: MT19937 key + position, and the cached Gaussian of legacy_gauss
  uint32_t key[624];
  int pos = 624;
  int has_gauss = 0;
  double gauss = 0.0;
  uint32_t out[624];       // the block's words tempered (derived from `key`; valid when `out_valid`)
  bool out_valid = false;

  void temper() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (mt_have_avx2()) mt_temper_avx2(key, out); else mt_temper_generic(key, out);
#endif
    out_valid = true;
  }
  void gen() {  // mt19937_gen (numpy/random/src/mt19937/mt19937.c): the next 624 words
#if !defined(__HIP_DEVICE_COMPILE__)
    if (mt_have_avx2()) mt_twist_avx2(key); else mt_twist_generic(key);
#endif
    pos = 0;
    temper();
  }
  inline uint32_t next_u32() {
    if (pos == 624) gen();
    else if (!out_valid) temper();   // a stream handed over mid-block
    return out[pos++];
  }
  inline double next_double() {  // mt19937_next_double
    const int32_t a = (int32_t)(next_u32() >> 5), b = (int32_t)(next_u32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  double legacy_gauss() {
    if (has_gauss) {
      const double temp = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return temp;
    }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
  inline double standard_exponential() { return -std::log(1.0 - next_double()); }
  double standard_gamma(double shape) {
    if (shape == 1.0) return standard_exponential();
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
      for (;;) {
        const double U = next_double();
        const double V = standard_exponential();
        if (U <= 1.0 - shape) {
          const double X = std::pow(U, 1. / shape);
          if (X <= V) return X;
        } else {
          const double Y = -std::log((1 - U) / shape);
          const double X = std::pow(1.0 - shape + shape * Y, 1. / shape);
          if (X <= (V + Y)) return X;
        }
      }
    }
    const double b = shape - 1. / 3.;
    const double c = 1. / std::sqrt(9 * b);
    for (;;) {
      double X, V;
      do {
        X = legacy_gauss();
        V = 1.0 + c * X;
      } while (V <= 0.0);
      V = V * V * V;
      const double U = next_double();
      if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
      if (std::log(U) < 0.5 * X * X + b * (1. - V + std::log(V))) return b * V;
    }
  }
  double beta(double a, double b) {
    if (a <= 1.0 && b <= 1.0) {  // Johnk's algorithm
      for (;;) {
        const double U = next_double();
        const double V = next_double();
        const double X = std::pow(U, 1.0 / a);
        const double Y = std::pow(V, 1.0 / b);
        if ((X + Y) <= 1.0) {
          if (X + Y > 0) return X / (X + Y);
          double logX = std::log(U) / a;
          double logY = std::log(V) / b;
          const double logM = logX > logY ? logX : logY;
          logX -= logM;
          logY -= logM;
          return std::exp(logX - std::log(std::exp(logX) + std::exp(logY)));
        }
      }
    }
    const double Ga = standard_gamma(a);
    const double Gb = standard_gamma(b);
    return Ga / (Ga + Gb);
  }
};

// Process-wide worker pool for the block fills (one task per parked instance).  Created on first use -- after any
// fork()ed model build of the caller; CMDP_HOST_THREADS overrides the worker count (default: hardware threads, <= 32).
class Pool {
 public:
  static Pool& get() {
    // never destroyed: the detached workers wait on the condition variable for the life of the process, and destroying a
    // condition variable that has waiters blocks (glibc) -- the process would hang in its exit handlers
    static Pool* p = new Pool;
    return *p;
  }
  // runs fn(i) for i in [0, n) on the workers and the calling thread; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || workers_.empty()) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    Job job{&fn, n};
    {
      std::lock_guard<std::mutex> lk(m_);
      jobs_.push_back(&job);
    }
    cv_.notify_all();
    run(job);
    std::unique_lock<std::mutex> lk(m_);
    for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
      if (*it == &job) { jobs_.erase(it); break; }
    done_cv_.wait(lk, [&] { return job.done.load() == n && job.active.load() == 0; });
  }

 private:
  struct Job {
    const std::function<void(int)>* fn;
    int n;
    std::atomic<int> next{0}, done{0}, active{0};
    Job(const std::function<void(int)>* f, int n_) : fn(f), n(n_) {}
  };
  void run(Job& j) {
    j.active.fetch_add(1);
    for (;;) {
      const int i = j.next.fetch_add(1);
      if (i >= j.n) break;
      (*j.fn)(i);
      j.done.fetch_add(1);
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      j.active.fetch_sub(1);
    }
    done_cv_.notify_all();
  }
  Pool() {
    int n = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("CMDP_HOST_THREADS")) n = std::atoi(e);
    n = n < 1 ? 1 : (n > 32 ? 32 : n);
    for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this] { loop(); });
    for (auto& t : workers_) t.detach();  // process-lifetime workers
  }
  void loop() {
    for (;;) {
      Job* j = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] {
          for (Job* c : jobs_)
            if (c->next.load() < c->n) { j = c; return true; }
          return false;
        });
        j->active.fetch_add(1);  // registered under the lock: the owner cannot retire the job before this worker leaves it
      }
      for (;;) {
        const int i = j->next.fetch_add(1);
        if (i >= j->n) break;
        (*j->fn)(i);
        j->done.fetch_add(1);
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        j->active.fetch_sub(1);
      }
      done_cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  std::vector<Job*> jobs_;
  std::vector<std::thread> workers_;
};

}  // namespace cmdp_rc
