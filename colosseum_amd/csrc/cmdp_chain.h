// cmdp_chain.h -- K9: long-run average reward of a deterministic stationary policy from a given state, one
// workgroup per instance (gfx950 / CDNA4 only).
//
// Reference: colosseum/mdp/utils/markov_chain.py:12-31 (`get_average_reward`), :64-136 (`get_stationary_distribution`)
// and :139-166 (`_gth_solve_numba`), as called at every logging step of the continuous-setting loop
// (colosseum/experiment/agent_mdp_interaction.py:518-532).  For a one-hot policy the chain is
// tps[s, :] = min(1, T[s, a(s), :]), rewards ars[s] = R[s, a(s)].  The reference then
//   1. lists the recurrent classes in networkx's `attracting_components` order;
//   2. one class smaller than the chain  -> float32 distribution, zero outside the class          (kind F32)
//      one class == the whole chain      -> float64 distribution                                   (kind F64)
//      several classes                   -> float64, the FIRST class (list order) the start state reaches, weight 1;
//   3. solves the class by GTH elimination in float64 and returns (ars * sd).sum() (numpy's pairwise summation).
//
// Phases of the kernel (LDS holds the adjacency and all index arrays; the dense class matrix is a float64 work
// copy in HBM/L2):
//   A  actions, out-degrees (wave scan), adjacency lists in column order                       -- all threads
//   B  networkx's non-recursive Tarjan/Nuutila search, verbatim (sources in node order, neighbours in adjacency
//      order), giving every state its component in EMISSION order                               -- one lane, LDS only
//   C  components with an edge leaving them are transient; the others are the recurrent classes in list order;
//      with several classes, forward reachability from the start state by frontier relaxation  -- all threads
//   D  members of the chosen class in ascending state order (ballot compaction), dense matrix  -- all threads
//   E  GTH: per pivot i the non-zeros of row i (wave 0) and column i (wave 1) are compacted in index order, the
//      scale is their index-ordered sum (v_readlane chain: the reference's order without a serial memory walk),
//      the rank-1 update touches |col| x |row| entries only -- adding the skipped zero products is exact, so the
//      result is bit-identical to the dense elimination
//   F  back-substitution, normalisation (index-ordered sums over the non-zero lanes), then the reward sum in
//      numpy's pairwise order (float32 products for kind F32).
#pragma once
#include "cmdp_device.h"

struct ChainArgs {
  int32_t B, A, max_deg;
  const int64_t* state_off;  // [B+1]
  const int64_t* csr_ptr;    // [R+1] global offsets, row = state * A + action
  const int32_t* csr_col;    // instance-relative state indices, ascending inside a row
  const float* csr_val;
  const float* R;            // [R]
  const float* pi;           // [R] one-hot rows; used when act == null
  const int32_t* act;        // [NSTATES] action of every state, or null
  const int32_t* start;      // [B] instance-relative start state
  const uint8_t* mask;       // [B] instances to evaluate, or null = all
  const int64_t* work_off;   // [B+1] prefix of S_b^2
  double* work;              // class matrices
  int32_t* work_idx;         // same offsets: row indices of the packed elimination columns
  double* avg;               // [B]
  int32_t* kind;             // [B] 0 = float64 result, 1 = float32 result
  int32_t* n_classes;        // [B] number of recurrent classes of the chain
  long long* dbg;            // [B][8] wall-clock stamps (100 MHz) at the phase boundaries, or null (tuning aid)
};

enum { CHAIN_F64 = 0, CHAIN_F32 = 1 };

__host__ __device__ inline size_t chain_lds_bytes(int S, int max_deg) {
  return sizeof(double) * 4 * (size_t)S + sizeof(int) * (12 * (size_t)S + 1 + (size_t)S * max_deg);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// acc + v[t0] + v[t1] + ... over the set lanes of `m` in lane order (wave-uniform result)
__device__ __forceinline__ double ordered_add(double acc, double v, unsigned long long m) {
  while (m) {
    const int t = __ffsll((long long)m) - 1;
    m &= m - 1;
    acc = __dadd_rn(acc, readlane_f64(v, t));
  }
  return acc;
}

// EXACT: acc + (set lanes of v in lane order) -- the reference's summation order, ~28 cycles per term on one chain.
// otherwise: acc + butterfly sum of the wave (fixed shuffle pattern: deterministic, but not the reference's order;
// relative difference ~1e-16 per sum) -- ~100 cycles per 64 terms.
template <bool EXACT>
__device__ __forceinline__ double lanes_add(double acc, double v, unsigned long long m) {
  if (EXACT) return ordered_add(acc, v, m);
  if (!m) return acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = __dadd_rn(v, __shfl_xor(v, o, 64));
  return __dadd_rn(acc, v);
}

// numpy's pairwise_sum (numpy/_core/src/umath/loops_utils.h.src): < 8 plain, <= 128 eight accumulators, else halves
template <typename T, int D>
__device__ T np_pairwise(const T* a, int n) {
  if (n < 8) {
    T r = (T)-0.0;
    for (int i = 0; i < n; ++i) r = r + a[i];
    return r;
  }
  if (D == 0 || n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise<T, (D > 0 ? D - 1 : 0)>(a, n2) + np_pairwise<T, (D > 0 ? D - 1 : 0)>(a + n2, n - n2);
}

template <int NW, bool EXACT>
__global__ void __launch_bounds__(NW * 64) k_chain_average_reward(ChainArgs c) {
  extern __shared__ unsigned char chain_smem[];
  __shared__ int s_i[8];
  __shared__ double s_scale;
  constexpr int NT = NW * 64;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (c.mask && !c.mask[b]) return;
  const int64_t soff = c.state_off[b];
  const int S = (int)(c.state_off[b + 1] - soff);
  const int A = c.A;
  const int64_t row0 = soff * A;
  double* xs = reinterpret_cast<double*>(chain_smem);
  double* rowv = xs + S;
  double* colv = rowv + S;
  double* ev = colv + S;
  int* act = reinterpret_cast<int*>(ev + S);
  int* adjp = act + S;  // [S+1]
  int* pre = adjp + S + 1;
  int* low = pre + S;
  int* it = low + S;
  int* comp = it + S;
  int* queue = comp + S;
  int* sccq = queue + S;
  int* members = sccq + S;
  int* pos = members + S;
  int* rowk = pos + S;
  int* colj = rowk + S;
  int* adj = colj + S;  // [<= S * max_deg]
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 0] = (long long)wall_clock64();
  // ---- A: actions, degrees, adjacency ------------------------------------------------------------------------------
  for (int s = tid; s < S; s += NT) {
    int a = 0;
    if (c.act) {
      a = c.act[soff + s];
    } else {
      const float* p = c.pi + row0 + (int64_t)s * A;
      for (int k = 0; k < A; ++k) a = (p[k] == 1.0f) ? k : a;
    }
    act[s] = a;
    const int64_t r = row0 + (int64_t)s * A + a;
    int d = 0;
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k) d += (c.csr_val[k] > 0.0f) ? 1 : 0;
    pre[s] = d;
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the degrees
    int carry = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      int v = (s < S) ? pre[s] : 0;
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
      }
      if (s < S) adjp[s + 1] = carry + v;
      carry += __shfl(v, 63, 64);
    }
    if (lane == 0) adjp[0] = 0;
  }
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const int64_t r = row0 + (int64_t)s * A + act[s];
    int o = adjp[s];
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k)
      if (c.csr_val[k] > 0.0f) adj[o++] = c.csr_col[k];
    pre[s] = 0;  // 0 = not visited (networkx numbers the preorder from 1)
    comp[s] = -1;
    it[s] = adjp[s];
  }
  __syncthreads();

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 1] = (long long)wall_clock64();
  // ---- B: strongly connected components in networkx's emission order ----------------------------------------------
  if (tid == 0) {
    int cnt = 0, ncomp = 0, sq = 0;
    for (int src = 0; src < S; ++src) {
      if (comp[src] >= 0) continue;
      int qn = 0;
      queue[qn++] = src;
      while (qn) {
        const int v = queue[qn - 1];
        int pv = pre[v];
        if (pv == 0) { pv = ++cnt; pre[v] = pv; }
        bool done = true;
        int k = it[v];
        const int kend = adjp[v + 1];
        while (k < kend) {
          const int w = adj[k++];
          if (pre[w] == 0) { queue[qn++] = w; done = false; break; }
        }
        it[v] = k;
        if (!done) continue;
        int lw = pv;
        for (int e = adjp[v]; e < kend; ++e) {
          const int w = adj[e];
          if (comp[w] < 0) {
            const int pw = pre[w];
            const int cand = (pw > pv) ? low[w] : pw;
            lw = cand < lw ? cand : lw;
          }
        }
        low[v] = lw;
        --qn;
        if (lw == pv) {
          comp[v] = ncomp;
          while (sq && pre[sccq[sq - 1]] > pv) comp[sccq[--sq]] = ncomp;
          ++ncomp;
        } else {
          sccq[sq++] = v;
        }
      }
    }
    s_i[0] = ncomp;
  }
  __syncthreads();
  const int ncomp = s_i[0];

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 2] = (long long)wall_clock64();
  // ---- C: recurrent classes (no edge leaves them), the class taken -------------------------------------------------
  int* leak = it;     // [ncomp]
  int* reach = queue; // [S]
  int* rcomp = sccq;  // [ncomp]
  for (int i = tid; i < S; i += NT) { leak[i] = 0; reach[i] = 0; rcomp[i] = 0; }
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const int cs = comp[s];
    for (int e = adjp[s]; e < adjp[s + 1]; ++e)
      if (comp[adj[e]] != cs) leak[cs] = 1;
  }
  __syncthreads();
  if (tid == 0) {
    int n_attr = 0, first = -1;
    for (int q = 0; q < ncomp; ++q)
      if (!leak[q]) { if (!n_attr) first = q; ++n_attr; }
    s_i[1] = n_attr;
    s_i[2] = first;
    reach[c.start[b]] = 1;
  }
  __syncthreads();
  const int n_attr = s_i[1];
  int chosen = s_i[2];
  if (n_attr > 1) {
    for (;;) {
      __syncthreads();
      if (tid == 0) s_i[3] = 0;
      __syncthreads();
      bool ch = false;
      for (int s = tid; s < S; s += NT) {
        if (!reach[s]) continue;
        for (int e = adjp[s]; e < adjp[s + 1]; ++e) {
          const int w = adj[e];
          if (!reach[w]) { reach[w] = 1; ch = true; }
        }
      }
      if (ch) s_i[3] = 1;
      __syncthreads();
      if (!s_i[3]) break;
    }
    for (int s = tid; s < S; s += NT)
      if (reach[s]) rcomp[comp[s]] = 1;
    __syncthreads();
    if (tid == 0) {
      int pick = -1;
      for (int q = 0; q < ncomp && pick < 0; ++q)
        if (!leak[q] && rcomp[q]) pick = q;
      s_i[2] = pick;
    }
    __syncthreads();
    chosen = s_i[2];
  }

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 3] = (long long)wall_clock64();
  // ---- D: members in ascending state order, dense matrix ----------------------------------------------------------
  if (wave == 0) {
    int m = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const bool in = s < S && chosen >= 0 && comp[s] == chosen;
      const unsigned long long bm = __ballot(in);
      if (in) {
        const int p = m + __popcll(bm & lt);
        members[p] = s;
        pos[s] = p;
      } else if (s < S) {
        pos[s] = -1;
      }
      m += __popcll(bm);
    }
    if (lane == 0) s_i[4] = m;
  }
  __syncthreads();
  const int m = s_i[4];
  double* a = c.work + c.work_off[b];
  int32_t* aj = c.work_idx + c.work_off[b];
  int* lcnt = pre;  // [m] entries of every packed column (the search arrays are free by now)
  for (int64_t e = tid; e < (int64_t)m * m; e += NT) a[e] = 0.0;
  __syncthreads();
  for (int i = tid; i < m; i += NT) {
    const int s = members[i];
    const int64_t r = row0 + (int64_t)s * A + act[s];
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k) {
      const float v = c.csr_val[k];
      if (v > 0.0f) a[(int64_t)i * m + pos[c.csr_col[k]]] = (double)fminf(1.0f, v);
    }
  }
  __syncthreads();

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 4] = (long long)wall_clock64();
  // ---- E: GTH elimination -------------------------------------------------------------------------------------------
  int n_eff = m;
  for (int i = 0; i < m - 1; ++i) {
    // both scans fetch up to PF chunks (64 * PF entries) before looking at any of them: one memory round trip per
    // pivot instead of one per chunk
    constexpr int PF = 8;
    if (wave == 0) {  // row i, columns k > i: ordered compaction + ordered sum
      int cntr = 0;
      double sc = 0.0;
      for (int k0 = i + 1; k0 < m; k0 += 64 * PF) {
        double v[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int k = k0 + 64 * u + lane;
          v[u] = (k < m) ? a[(int64_t)i * m + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (k0 + 64 * u >= m) break;
          const unsigned long long bm = __ballot(v[u] != 0.0);
          if (v[u] != 0.0) {
            const int p = cntr + __popcll(bm & lt);
            rowk[p] = k0 + 64 * u + lane;
            rowv[p] = v[u];
          }
          cntr += __popcll(bm);
          sc = lanes_add<EXACT>(sc, v[u], bm);
        }
      }
      if (lane == 0) { s_i[5] = cntr; s_scale = sc; }
    }
    if (wave == 1 % NW) {  // column i, rows j > i
      int cntc = 0;
      for (int j0 = i + 1; j0 < m; j0 += 64 * PF) {
        double v[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int j = j0 + 64 * u + lane;
          v[u] = (j < m) ? a[(int64_t)j * m + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (j0 + 64 * u >= m) break;
          const unsigned long long bm = __ballot(v[u] != 0.0);
          if (v[u] != 0.0) {
            const int p = cntc + __popcll(bm & lt);
            colj[p] = j0 + 64 * u + lane;
            colv[p] = v[u];
          }
          cntc += __popcll(bm);
        }
      }
      if (lane == 0) s_i[6] = cntc;
    }
    __syncthreads();
    const double sc = s_scale;
    if (sc <= 0.0) { n_eff = i + 1; break; }
    const int nrow = s_i[5], ncol = s_i[6];
    // a[j, i] /= scale.  The scaled column is what back-substitution needs; it is stored PACKED (value, row index) in
    // the part of row i right of the diagonal, which is dead from here on (its entries live in rowv for the update):
    // back-substitution then streams contiguous lists instead of walking strided columns
    for (int pj = tid; pj < ncol; pj += NT) {
      const double l = colv[pj] / sc;
      colv[pj] = l;
      a[(int64_t)i * m + i + 1 + pj] = l;
      aj[(int64_t)i * m + i + 1 + pj] = colj[pj];
    }
    if (tid == 0) lcnt[i] = ncol;
    __syncthreads();
    // rank-1 update of the |col| x |row| non-zero block, (j, k) pairs flattened over the whole workgroup; UB
    // independent read-modify-writes per thread are read first and written afterwards (one memory round trip per
    // batch, not one per element)
    constexpr int UB = 4;
    const int total = ncol * nrow;
    for (int e0 = tid; e0 < total; e0 += NT * UB) {
      double v[UB], add[UB];
      int64_t at[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = e0 + u * NT;
        if (e < total) {
          const int pj = e / nrow, pk = e - pj * nrow;
          at[u] = (int64_t)colj[pj] * m + rowk[pk];
          v[u] = a[at[u]];
          add[u] = __dmul_rn(colv[pj], rowv[pk]);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (e0 + u * NT < total) a[at[u]] = __dadd_rn(v[u], add[u]);
    }
    __syncthreads();
  }

  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 5] = (long long)wall_clock64();
  // ---- F: back-substitution, normalisation, reward sum -------------------------------------------------------------
  if (wave == 0 && m > 0) {
    for (int i = lane; i < m; i += 64) xs[i] = 0.0;
    if (lane == 0) xs[n_eff - 1] = 1.0;
    constexpr int PFB = 4;  // 64-entry chunks of a packed column held in registers; the NEXT pivot's are in flight
    double cv[PFB], nv[PFB];
    int cj[PFB], nj[PFB];
    auto load_col = [&](int i, double (&lv)[PFB], int (&lj)[PFB]) {
      const int64_t base = (int64_t)i * m + i + 1;
      const int cnt = lcnt[i];
#pragma unroll
      for (int u = 0; u < PFB; ++u) {
        const int e = 64 * u + lane;
        lv[u] = (e < cnt) ? a[base + e] : 0.0;
        lj[u] = (e < cnt) ? aj[base + e] : 0;
      }
    };
    if (n_eff >= 2) load_col(n_eff - 2, cv, cj);
    for (int i = n_eff - 2; i >= 0; --i) {
      if (i > 0) load_col(i - 1, nv, nj);
      const int cnt = lcnt[i];
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < PFB; ++u) {
        if (64 * u >= cnt) break;
        const double pr = (cv[u] != 0.0) ? __dmul_rn(xs[cj[u]], cv[u]) : 0.0;
        acc = lanes_add<EXACT>(acc, pr, __ballot(pr != 0.0));
      }
      for (int e0 = 64 * PFB; e0 < cnt; e0 += 64) {  // longer columns: the rest straight from memory
        const int e = e0 + lane;
        const int64_t base = (int64_t)i * m + i + 1;
        const double l = (e < cnt) ? a[base + e] : 0.0;
        const int j = (e < cnt) ? aj[base + e] : 0;
        const double pr = (l != 0.0) ? __dmul_rn(xs[j], l) : 0.0;
        acc = lanes_add<EXACT>(acc, pr, __ballot(pr != 0.0));
      }
      if (lane == 0) xs[i] = acc;
#pragma unroll
      for (int u = 0; u < PFB; ++u) { cv[u] = nv[u]; cj[u] = nj[u]; }
    }
    double tot = 0.0;
    for (int i0 = 0; i0 < n_eff; i0 += 64) {
      const int i = i0 + lane;
      const double v = (i < n_eff) ? xs[i] : 0.0;
      tot = lanes_add<EXACT>(tot, v, __ballot(v != 0.0));
    }
    for (int i = lane; i < n_eff; i += 64) xs[i] = xs[i] / tot;
  }
  __syncthreads();
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 6] = (long long)wall_clock64();
  const bool f32 = (n_attr == 1 && m < S);
  float* ev32 = reinterpret_cast<float*>(ev);
  for (int s = tid; s < S; s += NT) {
    const float ar = c.R[row0 + (int64_t)s * A + act[s]];
    const int p = pos[s];
    if (f32) ev32[s] = __fmul_rn(ar, p >= 0 ? (float)xs[p] : 0.0f);
    else ev[s] = __dmul_rn((double)ar, p >= 0 ? xs[p] : 0.0);
  }
  __syncthreads();
  if (tid == 0) {
    double out;
    if (m == 0) out = 0.0;
    else if (f32) out = (double)__fadd_rn(0.0f, np_pairwise<float, 12>(ev32, S));
    else out = __dadd_rn(0.0, np_pairwise<double, 12>(ev, S));
    c.avg[b] = out;
    c.kind[b] = f32 ? CHAIN_F32 : CHAIN_F64;
    if (c.n_classes) c.n_classes[b] = n_attr;
    if (c.dbg) { c.dbg[(int64_t)b * 8 + 7] = (long long)wall_clock64(); }
  }
}


// ===================================================================================================
// K9F k_chain_fast: the average reward of IRREDUCIBLE policy chains by GTH elimination in a FILL-REDUCING ORDER, several
// independent pivots per round (round 3).
//
// K9 above spends its time in 783 dependent pivot steps (S = 784: 4.1 us each -- two scans of a dense row / column of the
// work matrix in L2, three barriers) after a one-lane Tarjan search (1.2 ms), 5.1 ms per log row, and the continuous
// MiniGrid batches of the benchmark (config C4) are bound by it.  GTH censors one state at a time and ANY order gives the
// same stationary distribution up to rounding (the SURVEY's tolerance for this row is 1e-6 on the average reward; K9's
// default already replaces the reference's index-ordered sums by wave butterflies).  The policy's chain is a sub-graph of
// the MDP's transition graph (union over the actions), which is fixed per instance, so the host computes ONCE per instance
// (cmdp.hip: build_chain_plan):
//   * a minimum-degree elimination order of that graph (symmetrised) -- S = 784: 10 332 candidate entries in the filled
//     graph instead of ~75 000 in state order, at most 59 per pivot;
//   * per pivot its CANDIDATE list (the higher-ranked neighbours in the filled graph: a superset of the non-zeros of its
//     row and column under any policy), so a pivot's scan is one gather of <= 64 values instead of a walk over the row;
//   * a schedule of ROUNDS: up to 16 pivots per round that are pairwise non-adjacent in the filled graph and share at most
//     one candidate -- their rank-1 updates then touch disjoint off-diagonal entries (the diagonal is never read), so a
//     round's pivots are scanned, scaled and applied concurrently with no atomics and a result that does not depend on
//     the interleaving.  S = 784: 197 rounds instead of 783 pivot steps.
// The kernel first checks that the chain is irreducible (forward and backward reachability from state 0 by frontier
// relaxation over the adjacency in LDS, all threads; an irreducible chain is ONE recurrent class == the whole chain, the
// reference's float64 branch) -- otherwise, or without a plan, it flags the instance for K9 (which then runs unchanged).
// One wavefront per pivot of a round: gather the candidates' row / column values, ballot-compact the non-zeros, butterfly
// sum = the scale; scale the column and store it packed in the dead part of the pivot's row (as K9 does); one barrier;
// all threads apply the round's updates; one barrier.  Back-substitution walks the rounds in reverse, one wavefront per
// pivot.  Not bit-equal to K9 (other elimination order): CMDP_OPT_CHAIN_EXACT_ORDER = 1 keeps K9 alone.
// ===================================================================================================
struct ChainFast {
  const int32_t* rank;      // [NSTATES] elimination position of every state (instance-relative positions)
  const int32_t* cptr;      // [NSTATES + B] per instance S_b + 1 offsets into its candidate block, at state_off[b] + b
  const int64_t* cbase;     // [B] start of the instance's candidate block
  const uint16_t* cand;     // candidate positions (> the pivot's), ascending
  const int32_t* nrounds;   // [B] rounds of the instance, -1 = no plan
  const int64_t* rbase;     // [B] start of the instance's round offsets (nrounds + 1 entries)
  const int32_t* rptr;      // round -> first pivot slot
  const int32_t* piv;       // [NSTATES] pivots in schedule order, at state_off[b]
  uint8_t* slow;            // [B] out: 1 = evaluate this instance with K9
};

#define K9F_KC 2                      // candidate chunks of 64 a pivot's wavefront handles
#define K9F_MAXC (64 * K9F_KC)        // candidates per pivot the plan may hold (more: the instance keeps K9)
__host__ __device__ inline size_t chain_fast_lds_bytes(int S, int max_deg, int nw) {
  return sizeof(double) * (2 * (size_t)S + 2 * (size_t)nw * K9F_MAXC + nw) +
         sizeof(int) * ((size_t)7 * S + 2 + (size_t)S * max_deg + 2 * (size_t)nw * K9F_MAXC + 2 * nw + 16);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) k_chain_fast(ChainArgs c, ChainFast f) {
  extern __shared__ unsigned char chain_smem[];
  __shared__ int s_f[4];
  constexpr int NT = NW * 64;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (c.mask && !c.mask[b]) { if (tid == 0) f.slow[b] = 0; return; }
  const int R = f.nrounds[b];
  if (R < 0) { if (tid == 0) f.slow[b] = 1; return; }
  const int64_t soff = c.state_off[b];
  const int S = (int)(c.state_off[b + 1] - soff);
  const int A = c.A;
  const int64_t row0 = soff * A;
  double* xs = reinterpret_cast<double*>(chain_smem);
  double* ev = xs + S;
  constexpr int MC = K9F_MAXC;
  double* wrowv = ev + S;                 // [NW][MC]
  double* wcolv = wrowv + NW * MC;        // [NW][MC]
  double* wscale = wcolv + NW * MC;       // [NW]
  int* act = reinterpret_cast<int*>(wscale + NW);
  int* adjp = act + S;                    // [S + 1]
  int* fwd = adjp + S + 1;
  int* bwd = fwd + S;
  int* rk = bwd + S;                      // rank of every state
  int* lcnt = rk + S;                     // entries of every packed column
  int* deg = lcnt + S;
  int* wrowk = deg + S;                   // [NW][MC]
  int* wcolj = wrowk + NW * MC;           // [NW][MC]
  int* wcnt = wcolj + NW * MC;            // [NW][2]
  int* wpre = wcnt + 2 * NW;              // [16] prefix of the round's update counts
  int* adj = wpre + 16;                   // [<= S * max_deg]
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 0] = (long long)wall_clock64();
  // ---- A: actions, out-degrees, adjacency (as K9) --------------------------------------------------------------------
  for (int s = tid; s < S; s += NT) {
    int a = 0;
    if (c.act) {
      a = c.act[soff + s];
    } else {
      const float* p = c.pi + row0 + (int64_t)s * A;
      for (int k = 0; k < A; ++k) a = (p[k] == 1.0f) ? k : a;
    }
    act[s] = a;
    const int64_t r = row0 + (int64_t)s * A + a;
    int d = 0;
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k) d += (c.csr_val[k] > 0.0f) ? 1 : 0;
    deg[s] = d;
    rk[s] = f.rank[soff + s];
    fwd[s] = (s == 0) ? 1 : 0;
    bwd[s] = (s == 0) ? 1 : 0;
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the degrees
    int carry = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      int v = (s < S) ? deg[s] : 0;
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
      }
      if (s < S) adjp[s + 1] = carry + v;
      carry += __shfl(v, 63, 64);
    }
    if (lane == 0) adjp[0] = 0;
  }
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const int64_t r = row0 + (int64_t)s * A + act[s];
    int o = adjp[s];
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k)
      if (c.csr_val[k] > 0.0f) adj[o++] = c.csr_col[k];
  }
  __syncthreads();
  // ---- R: irreducible?  every state reachable from state 0 and state 0 reachable from every state -----------------------
  for (;;) {
    if (tid == 0) s_f[0] = 0;
    __syncthreads();
    bool ch = false;
    for (int s = tid; s < S; s += NT) {
      const bool fs = fwd[s] != 0;
      bool bs = bwd[s] != 0;
      for (int e = adjp[s]; e < adjp[s + 1]; ++e) {
        const int w = adj[e];
        if (fs && !fwd[w]) { fwd[w] = 1; ch = true; }
        if (!bs && bwd[w]) { bs = true; bwd[s] = 1; ch = true; }
      }
    }
    if (ch) s_f[0] = 1;
    __syncthreads();
    const int again = s_f[0];
    __syncthreads();
    if (!again) break;
  }
  if (tid == 0) s_f[1] = 1;
  __syncthreads();
  for (int s = tid; s < S; s += NT)
    if (!fwd[s] || !bwd[s]) s_f[1] = 0;
  __syncthreads();
  if (!s_f[1]) { if (tid == 0) f.slow[b] = 1; return; }   // several classes or transient states: K9's job
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 1] = c.dbg[(int64_t)b * 8 + 2] = c.dbg[(int64_t)b * 8 + 3] = (long long)wall_clock64();
  // ---- D: the chain as a dense matrix in ELIMINATION positions ------------------------------------------------------
  const int m = S;
  double* a = c.work + c.work_off[b];
  int32_t* aj = c.work_idx + c.work_off[b];
  // only the entries of the FILLED graph are ever read (a pivot's row and column at its candidates; every update lands on a
  // pair of candidates of the same pivot, which the fill makes neighbours), so only those are cleared -- 2 x 10 332 stores
  // at S = 784 instead of the 614 656 of the whole matrix (4.9 MB per instance and log row, which also swept the L2)
  {
    const int32_t* cptr0 = f.cptr + soff + b;
    const uint16_t* cand0 = f.cand + f.cbase[b];
    for (int i = wave; i < m - 1; i += NW) {
      const int c0 = cptr0[i], cnt = cptr0[i + 1] - c0;
      for (int l = lane; l < cnt; l += 64) {
        const int kk = (int)cand0[c0 + l];
        a[(int64_t)i * m + kk] = 0.0;
        a[(int64_t)kk * m + i] = 0.0;
      }
    }
  }
  __syncthreads();
  for (int s = tid; s < S; s += NT) {
    const int64_t r = row0 + (int64_t)s * A + act[s];
    const int64_t base = (int64_t)rk[s] * m;
    for (int64_t k = c.csr_ptr[r]; k < c.csr_ptr[r + 1]; ++k) {
      const float v = c.csr_val[k];
      if (v > 0.0f) a[base + rk[c.csr_col[k]]] = (double)fminf(1.0f, v);
    }
  }
  if (tid == 0) s_f[2] = 0;
  __syncthreads();
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 4] = (long long)wall_clock64();
  // ---- E: GTH elimination, a round of independent pivots at a time ------------------------------------------------------
  const int32_t* cptr = f.cptr + soff + b;
  const uint16_t* cand = f.cand + f.cbase[b];
  const int32_t* rptr = f.rptr + f.rbase[b];
  const int32_t* piv = f.piv + soff;
  double* myrowv = wrowv + wave * MC;
  double* mycolv = wcolv + wave * MC;
  int* myrowk = wrowk + wave * MC;
  int* mycolj = wcolj + wave * MC;
  for (int r = 0; r < R; ++r) {
    const int p0 = rptr[r], np = rptr[r + 1] - p0;
    if (wave < np) {
      const int i = piv[p0 + wave];
      const int c0 = cptr[i], cnt = cptr[i + 1] - c0;
      // the candidates' row and column values (all gathers in flight together), non-zeros compacted in candidate order
      int kk[K9F_KC];
      double rv[K9F_KC], cv[K9F_KC];
#pragma unroll
      for (int u = 0; u < K9F_KC; ++u) {
        const int l = 64 * u + lane;
        kk[u] = (l < cnt) ? (int)cand[c0 + l] : 0;
        rv[u] = (l < cnt) ? a[(int64_t)i * m + kk[u]] : 0.0;
        cv[u] = (l < cnt) ? a[(int64_t)kk[u] * m + i] : 0.0;
      }
      int nrow = 0, ncol = 0;
      double sc = 0.0;
#pragma unroll
      for (int u = 0; u < K9F_KC; ++u) {
        if (64 * u >= cnt) break;
        const unsigned long long bmr = __ballot(rv[u] != 0.0), bmc = __ballot(cv[u] != 0.0);
        if (rv[u] != 0.0) {
          const int q = nrow + __popcll(bmr & lt);
          myrowk[q] = kk[u];
          myrowv[q] = rv[u];
        }
        if (cv[u] != 0.0) {
          const int q = ncol + __popcll(bmc & lt);
          mycolj[q] = kk[u];
          mycolv[q] = cv[u];
        }
        sc = lanes_add<false>(sc, rv[u], bmr);
        nrow += __popcll(bmr);
        ncol += __popcll(bmc);
      }
      __builtin_amdgcn_wave_barrier();
      if (sc <= 0.0) {
        if (lane == 0) s_f[2] = 1;   // cannot happen on an irreducible chain; K9 takes the instance
      } else {
        // a[j, i] /= scale; the scaled column is what back-substitution needs: stored packed (value, position) in the
        // part of row i right of the diagonal, dead from here on
        for (int l = lane; l < ncol; l += 64) {
          const double lv = mycolv[l] / sc;
          mycolv[l] = lv;
          a[(int64_t)i * m + i + 1 + l] = lv;
          aj[(int64_t)i * m + i + 1 + l] = mycolj[l];
        }
      }
      if (lane == 0) {
        lcnt[i] = ncol;
        wcnt[2 * wave] = nrow;
        wcnt[2 * wave + 1] = ncol;
      }
    }
    __syncthreads();
    if (s_f[2]) break;
    if (tid == 0) {
      int acc = 0;
      for (int w = 0; w < np; ++w) { wpre[w] = acc; acc += wcnt[2 * w] * wcnt[2 * w + 1]; }
      wpre[np] = acc;
    }
    __syncthreads();
    const int total = wpre[np];
    for (int e = tid; e < total; e += NT) {
      int w = 0;
      while (w + 1 < np && wpre[w + 1] <= e) ++w;
      const int le = e - wpre[w], nrow = wcnt[2 * w];
      const int pj = le / nrow, pk = le - pj * nrow;
      const int64_t at = (int64_t)wcolj[w * MC + pj] * m + wrowk[w * MC + pk];
      a[at] = __dadd_rn(a[at], __dmul_rn(wcolv[w * MC + pj], wrowv[w * MC + pk]));
    }
    __syncthreads();
  }
  if (s_f[2]) { if (tid == 0) f.slow[b] = 1; return; }
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 5] = (long long)wall_clock64();
  // ---- F: back-substitution, the rounds in reverse ---------------------------------------------------------------------
  for (int i = tid; i < m; i += NT) xs[i] = 0.0;
  __syncthreads();
  if (tid == 0) xs[m - 1] = 1.0;   // the one position that is never eliminated
  __syncthreads();
  for (int r = R - 1; r >= 0; --r) {
    const int p0 = rptr[r], np = rptr[r + 1] - p0;
    if (wave < np) {
      const int i = piv[p0 + wave];
      const int cnt = lcnt[i];
      const int64_t base = (int64_t)i * m + i + 1;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < K9F_KC; ++u) {
        if (64 * u >= cnt) break;
        const int l = 64 * u + lane;
        const double lv = (l < cnt) ? a[base + l] : 0.0;
        const int j = (l < cnt) ? aj[base + l] : 0;
        const double pr = (l < cnt) ? __dmul_rn(xs[j], lv) : 0.0;
        acc = lanes_add<false>(acc, pr, ~0ull);
      }
      if (lane == 0) xs[i] = acc;
    }
    __syncthreads();
  }
  if (wave == 0) {
    double tot = 0.0;
    for (int i0 = 0; i0 < m; i0 += 64) {
      const int i = i0 + lane;
      const double v = (i < m) ? xs[i] : 0.0;
      tot = lanes_add<false>(tot, v, ~0ull);
    }
    if (lane == 0) wscale[0] = tot;
  }
  __syncthreads();
  const double tot = wscale[0];
  if (c.dbg && tid == 0) c.dbg[(int64_t)b * 8 + 6] = (long long)wall_clock64();
  for (int s = tid; s < S; s += NT) {
    const float ar = c.R[row0 + (int64_t)s * A + act[s]];
    ev[s] = __dmul_rn((double)ar, xs[rk[s]] / tot);
  }
  __syncthreads();
  if (tid == 0) {
    c.avg[b] = __dadd_rn(0.0, np_pairwise<double, 12>(ev, S));
    c.kind[b] = CHAIN_F64;   // one recurrent class == the whole chain: the reference's float64 branch
    if (c.n_classes) c.n_classes[b] = 1;
    f.slow[b] = 0;
    if (c.dbg) { c.dbg[(int64_t)b * 8 + 7] = (long long)wall_clock64(); }
  }
}

// ===================================================================================================
// K10: mixing time of the chain of a stationary policy (BUILD-DEFINED: the reference has no mixing time; SURVEY
// section 8 f2 defines it as the smallest t with max_s TV(P^t(s, .), pi) <= threshold, threshold 1/4).
// X_t[s, :] = distribution after t steps from state s (X_0 = I), float64, one workgroup per (instance, start state):
// the row of X_t sits in LDS, column j gathers its predecessors through the CSC of P (index order), the total
// variation to the stationary distribution is block-reduced and max-reduced over the start states with an integer
// atomicMax on the bit pattern (non-negative doubles order like their bits).
// ===================================================================================================
struct MixArgs {
  int32_t B;
  const int64_t* state_off;  // [B+1]
  const int64_t* x_off;      // [B+1] prefix of S_b^2
  const int64_t* csc_ptr;    // [NSTATES+1] global offsets, column = flat state index
  const int32_t* csc_row;    // instance-relative predecessor
  const double* csc_val;
  const double* stationary;  // [NSTATES]
  double* X;
  double* Xnew;
  unsigned long long* dlist; // [B][chunk] bit patterns of max_s TV
  int32_t chunk, k;
};

__global__ void __launch_bounds__(256) k_mix_init(MixArgs m) {
  int lo = 0, hi = m.B;
  const int64_t unit = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (m.state_off[mid] <= unit) lo = mid; else hi = mid;
  }
  const int b = lo;
  const int S = (int)(m.state_off[b + 1] - m.state_off[b]);
  const int s = (int)(unit - m.state_off[b]);
  double* x = m.X + m.x_off[b] + (int64_t)s * S;
  for (int j = threadIdx.x; j < S; j += blockDim.x) x[j] = (j == s) ? 1.0 : 0.0;
}

__global__ void __launch_bounds__(256) k_mix_step(MixArgs m) {
  extern __shared__ double mix_xs[];
  __shared__ double red[4];
  int lo = 0, hi = m.B;
  const int64_t unit = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (m.state_off[mid] <= unit) lo = mid; else hi = mid;
  }
  const int b = lo;
  const int64_t soff = m.state_off[b];
  const int S = (int)(m.state_off[b + 1] - soff);
  const int s = (int)(unit - soff);
  const double* x = m.X + m.x_off[b] + (int64_t)s * S;
  double* xn = m.Xnew + m.x_off[b] + (int64_t)s * S;
  for (int i = threadIdx.x; i < S; i += 256) mix_xs[i] = x[i];
  __syncthreads();
  double part = 0.0;
  for (int j = threadIdx.x; j < S; j += 256) {
    double acc = 0.0;
    for (int64_t k = m.csc_ptr[soff + j]; k < m.csc_ptr[soff + j + 1]; ++k)
      acc = __dadd_rn(acc, __dmul_rn(mix_xs[m.csc_row[k]], m.csc_val[k]));
    xn[j] = acc;
    part += fabs(acc - m.stationary[soff + j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tv = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);
    atomicMax(m.dlist + (int64_t)b * m.chunk + m.k, (unsigned long long)__double_as_longlong(tv));
  }
}


// ---------------------------------------------------------------------------------------------------
// Mixing time of LARGE chains (a float64 row of X does not fit LDS, or S > 1024): X_t = P^t as a dense S x S
// float64 matrix, advanced by repeated squaring (rocBLAS dgemm, the one plain library GEMM of this library) and, for
// the last stretch, by sparse steps that gather from the row in global memory.  cmdp_mixing_time describes the search.
// ---------------------------------------------------------------------------------------------------
struct MixDense {
  int32_t S;
  int64_t soff;              // first flat state of the instance
  const int64_t* csc_ptr;    // global offsets, column = flat state
  const int32_t* csc_row;    // instance-relative predecessor
  const double* csc_val;
  const double* stationary;  // flat
  unsigned long long* out;   // bit pattern of max_s TV (atomicMax)
};

// P as a dense row-major matrix from its CSC (the buffer has been zeroed); one thread per column.
__global__ void __launch_bounds__(256) k_mixd_build(MixDense m, double* __restrict__ P) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m.S) return;
  for (int64_t k = m.csc_ptr[m.soff + j]; k < m.csc_ptr[m.soff + j + 1]; ++k)
    P[(int64_t)m.csc_row[k] * m.S + j] = m.csc_val[k];
}

__device__ __forceinline__ void mixd_reduce_tv(double part, unsigned long long* out) {
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tv = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);
    atomicMax(out, (unsigned long long)__double_as_longlong(tv));
  }
}

// max_s TV(X[s, :], stationary): one workgroup per row.
__global__ void __launch_bounds__(256) k_mixd_tv(MixDense m, const double* __restrict__ X) {
  const double* x = X + (int64_t)blockIdx.x * m.S;
  double part = 0.0;
  for (int j = threadIdx.x; j < m.S; j += 256) part += fabs(x[j] - m.stationary[m.soff + j]);
  mixd_reduce_tv(part, m.out);
}

// One step X_{t+1}[s, :] = X_t[s, :] P for every row s (one workgroup per row; the gathers hit the row in L2), same
// index-order float64 accumulation as k_mix_step, and the TV of the new row.
__global__ void __launch_bounds__(256) k_mixd_step(MixDense m, const double* __restrict__ X, double* __restrict__ Xn) {
  const double* x = X + (int64_t)blockIdx.x * m.S;
  double* xn = Xn + (int64_t)blockIdx.x * m.S;
  double part = 0.0;
  for (int j = threadIdx.x; j < m.S; j += 256) {
    double acc = 0.0;
    for (int64_t k = m.csc_ptr[m.soff + j]; k < m.csc_ptr[m.soff + j + 1]; ++k)
      acc = __dadd_rn(acc, __dmul_rn(x[m.csc_row[k]], m.csc_val[k]));
    xn[j] = acc;
    part += fabs(acc - m.stationary[m.soff + j]);
  }
  mixd_reduce_tv(part, m.out);
}


// ===================================================================================================
// k_emit: non-tabular observations of all instances (reference colosseum/emission_maps/base.py:56-141): row
// [h][state] of the instance's float32 feature table (zeros once an episodic instance has reached its horizon),
// plus scale * N(0, 1) noise in throughput mode: Philox domain 4, counter (observation number of the instance,
// element pair), Box-Muller in float64 on 32-bit uniforms, two normals per block half.  One workgroup per instance.
// ===================================================================================================
struct EmitArgs {
  int32_t B, F, H, time_indexed;
  const int64_t* state_off;
  const float* table;          // per instance [H or 1][S_b][F] at (time_indexed ? H : 1) * state_off[b] * F
  const int32_t* cur;
  const int32_t* hstep;
  const uint2* key;
  unsigned long long* n_obs;   // [B] observations emitted so far (noise counter)
  double scale;                // Gaussian uncorrelated: standard deviation; <= 0 with kind 1: no noise
  int32_t kind;                // 0 none, 1 Gaussian uncorrelated, 2 Gaussian correlated, 3 Student-t uncorrelated, 4 Student-t correlated
  double df;                   // Student-t degrees of freedom
  const float* chol;           // [F][F] lower Cholesky factor of the covariance / shape matrix (kinds 2, 4)
  float* out;                  // [B][F]
};

// Noise kinds of the throughput mode (the reference-exact streams are numpy's and stay on the host): standard normals z_j
// from Philox domain 4 (Box-Muller, counter = (observation number, element pair)); Student-t: z / sqrt(chi2(df) / df) with
// chi2(df) = 2 Gamma(df / 2) (Marsaglia-Tsang on Philox domain 5: one per element when uncorrelated, one per observation
// when correlated -- the multivariate t); correlated kinds multiply by the lower Cholesky factor L of the covariance /
// shape matrix, y = L z (float64 accumulation in index order), with z staged in LDS.
__global__ void __launch_bounds__(256) k_emit(EmitArgs e) {
  extern __shared__ double emit_z[];
  const int b = blockIdx.x;
  const int64_t so = e.state_off[b];
  const int S = (int)(e.state_off[b + 1] - so);
  const int h = e.hstep[b];
  const bool ended = e.H > 0 && h >= e.H;
  const int layer = e.time_indexed ? (h < e.H ? h : 0) : 0;
  const float* row = e.table + ((int64_t)(e.time_indexed ? e.H : 1) * so + (int64_t)layer * S + e.cur[b]) * e.F;
  const unsigned long long n = e.n_obs[b];
  const uint2 key = e.key ? e.key[b] : make_uint2(0, 0);
  const bool noisy = !ended && (e.kind >= 2 || (e.kind == 1 && e.scale > 0.0));
  const bool correlated = e.kind == 2 || e.kind == 4;
  auto normal = [&](int j) -> double {
    uint32_t w[4];
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 4u, (uint32_t)(j >> 1), key.x, key.y, w);
    const uint32_t a = (j & 1) ? w[2] : w[0], c = (j & 1) ? w[3] : w[1];
    const double u1 = ((double)a + 1.0) * (1.0 / 4294967296.0), u2 = (double)c * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
  };
  if (noisy && correlated) {
    for (int j = threadIdx.x; j < e.F; j += blockDim.x) emit_z[j] = normal(j);
    __syncthreads();
  }
  double t_scale = 1.0;
  if (noisy && e.kind == 4) {  // one chi-square per observation, the same for every element
    uint32_t draw = 0;
    t_scale = 1.0 / sqrt(2.0 * philox_gamma(0.5 * e.df, n, key, draw, 5u) / e.df);
  }
  for (int j = threadIdx.x; j < e.F; j += blockDim.x) {
    float v = ended ? 0.0f : row[j];
    if (noisy) {
      double y;
      if (correlated) {
        y = 0.0;
        const float* lrow = e.chol + (int64_t)j * e.F;
        for (int k = 0; k <= j; ++k) y += (double)lrow[k] * emit_z[k];
        y *= t_scale;
      } else if (e.kind == 3) {
        uint32_t draw = (uint32_t)j << 8;
        y = normal(j) / sqrt(2.0 * philox_gamma(0.5 * e.df, n, key, draw, 5u) / e.df);
      } else {
        y = e.scale * normal(j);
      }
      v = v + (float)y;
    }
    e.out[(int64_t)b * e.F + j] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0 && noisy) e.n_obs[b] = n + 1;
}
