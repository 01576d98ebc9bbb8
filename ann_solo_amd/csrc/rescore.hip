// rescore.hip -- batched (shifted) dot-product rescoring.
// Replaces get_best_match (/root/reference/src/ann_solo/spectrum_match.pyx:28-108)
// and SpectrumMatcher::dot (/root/reference/src/ann_solo/SpectrumMatch.cpp:8-133).
//
// Pass 1 (rescore_score_kernel): one workgroup per query, one 64-lane wavefront
// per (query, candidate) pair. Query peaks sit in LDS for the whole workgroup;
// each wave stages its candidate's peaks in LDS, lanes = query peaks run the
// window search for every shift (binary search instead of the reference's
// running cursor -- same cursor value because both peak lists ascend), matches
// go to an LDS list keyed (product desc, generation order asc), a wave-level
// bitonic sort orders them and a scalar-unit greedy loop assigns them one to one.
// Only the double score leaves the kernel.
// Pass 2 (rescore_argmax_kernel): per query first-strict-maximum (cpp:118-129).
// Pass 3 (rescore_matches_kernel): the winning pair is re-run once per query to
// emit its peak_matches in greedy order.
//
// Arithmetic mirrors the reference: window tests in double on float->double
// promoted m/z (cpp:42,53); product = (float)(mult * (double)q_int * (double)c_int)
// (cpp:81); score = double sum of those floats in sorted order (cpp:104).
#include <cstdlib>

#include "common.hpp"

namespace asl {

constexpr int RS_WAVES = 4;
constexpr int RS_MAXP = 256;   // peaks per spectrum the kernels accept
constexpr int RS_MCAP = 512;   // generated peak matches per pair the kernels accept

enum { RS_STATUS_OK = 0, RS_STATUS_PEAKS = 1, RS_STATUS_MATCHES = 2 };

struct WaveLds {
  float c_mz[RS_MAXP];
  float c_int[RS_MAXP];
  unsigned long long keys[RS_MCAP];
  uint32_t pay[RS_MCAP];
  uint8_t c_chg[RS_MAXP];
  int counter;
  int pad[3];
};

struct QueryLds {
  float mz[RS_MAXP];
  float inten[RS_MAXP];
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sort the wave's match list (product desc, generation order asc) and assign greedily
// (SpectrumMatch.cpp:92-111). Lists of <= 64 matches are sorted in registers with
// cross-lane shuffles; longer ones in LDS. Returns the score in every lane.
template <bool EMIT>
__device__ double resolve_matches(int lane, WaveLds &W, uint32_t *out_pairs, int out_cap,
                                  int *out_count, int *status) {
  int M = W.counter;
  M = __builtin_amdgcn_readfirstlane(M);
  if (M > RS_MCAP) {
    if (lane == 0) atomicOr(status, RS_STATUS_MATCHES);
    M = RS_MCAP;
  }
  if (M == 0) return 0.0;
  unsigned long long rkey = 0ull;
  uint32_t rpay = 0;
  const bool small = M <= 64;
  if (small) {
    if (lane < M) {
      rkey = W.keys[lane];
      rpay = W.pay[lane];
    }
    if (M > 1) {
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const unsigned long long ok = __shfl_xor(rkey, j, 64);
          const uint32_t op = __shfl_xor(rpay, j, 64);
          const bool keep_max = (((lane & j) == 0) == ((lane & k) == 0));
          const bool take = ((ok > rkey) == keep_max) && (ok != rkey);
          rkey = take ? ok : rkey;
          rpay = take ? op : rpay;
        }
      }
    }
  } else {
    int P = 2;
    while (P < M) P <<= 1;
    for (int i = M + lane; i < P; i += 64) W.keys[i] = 0ull;
    wave_sync();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < (P >> 1); t += 64) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const unsigned long long a = W.keys[i], b = W.keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            W.keys[i] = b;
            W.keys[l] = a;
            const uint32_t pa = W.pay[i];
            W.pay[i] = W.pay[l];
            W.pay[l] = pa;
          }
        }
        wave_sync();
      }
    }
  }

  // greedy one-to-one assignment on the scalar unit
  unsigned long long qu0 = 0, qu1 = 0, qu2 = 0, qu3 = 0, cu0 = 0, cu1 = 0, cu2 = 0, cu3 = 0;
  double score = 0.0;
  int nm = 0;
  for (int t = 0; t < M; ++t) {
    uint32_t pl, pb;
    if (small) {
      pl = __builtin_amdgcn_readlane(rpay, t);
      pb = __builtin_amdgcn_readlane((uint32_t)(rkey >> 32), t);
    } else {
      pl = __builtin_amdgcn_readfirstlane(W.pay[t]);
      pb = __builtin_amdgcn_readfirstlane((uint32_t)(W.keys[t] >> 32));
    }
    const uint32_t qi = pl >> 16, ci = pl & 0xffffu;
    const unsigned long long qbit = 1ull << (qi & 63), cbit = 1ull << (ci & 63);
    const uint32_t qw = qi >> 6, cw = ci >> 6;
    const unsigned long long qword = qw == 0 ? qu0 : qw == 1 ? qu1 : qw == 2 ? qu2 : qu3;
    const unsigned long long cword = cw == 0 ? cu0 : cw == 1 ? cu1 : cw == 2 ? cu2 : cu3;
    if (!(qword & qbit) && !(cword & cbit)) {
      score += (double)__uint_as_float(pb);
      if (qw == 0) qu0 |= qbit; else if (qw == 1) qu1 |= qbit; else if (qw == 2) qu2 |= qbit; else qu3 |= qbit;
      if (cw == 0) cu0 |= cbit; else if (cw == 1) cu1 |= cbit; else if (cw == 2) cu2 |= cbit; else cu3 |= cbit;
      if (EMIT) {
        if (lane == 0 && nm < out_cap) {
          out_pairs[2 * nm] = qi;
          out_pairs[2 * nm + 1] = ci;
        }
      }
      ++nm;
    }
  }
  if (EMIT && out_count && lane == 0) *out_count = nm;
  return score;
}

// Wave-cooperative SpectrumMatcher::dot for one pair. All 64 lanes call it with
// identical arguments; returns the score in every lane. EMIT: lane 0 writes the
// greedy (query_peak, candidate_peak) pairs.
template <bool EMIT>
__device__ double dot_pair_wave(int lane, const QueryLds &Q, int qn, double q_pmz,
                                const DevPeaks &L, int row, double tol, int allow_shift,
                                WaveLds &W, uint32_t *out_pairs, int out_cap, int *out_count,
                                int *status) {
  const int co = L.offsets[row];
  int cn = L.offsets[row + 1] - co;
  if (EMIT && out_count && lane == 0) *out_count = 0;
  if (cn > RS_MAXP) {
    if (lane == 0) atomicOr(status, RS_STATUS_PEAKS);
    cn = RS_MAXP;
  }
  if (cn <= 0 || qn <= 0) return 0.0;
  for (int i = lane; i < cn; i += 64) {
    W.c_mz[i] = L.mz[co + i];
    W.c_int[i] = L.intensity[co + i];
    W.c_chg[i] = L.charge ? L.charge[co + i] : (uint8_t)0;
  }
  if (lane == 0) W.counter = 0;
  const int c_charge = L.precursor_charge[row];
  const double pmd = (q_pmz - L.precursor_mz[row]) * (double)(unsigned)c_charge;  // cpp:18
  const int S = (allow_shift && fabs(pmd) >= tol) ? c_charge + 1 : 1;            // cpp:20
  wave_sync();

  for (int qb = 0; qb < qn; qb += 64) {
    const int qi = qb + lane;
    if (qi < qn) {
      const double qm = (double)Q.mz[qi];
      const float q_int = Q.inten[qi];
      for (int s = 0; s < S; ++s) {
        const double md = s ? pmd / (double)s : 0.0;  // cpp:26-31
        // cursor = first j where NOT (qm - tol > c_mz[j] + md), capped at cn-1 (cpp:39-46)
        int lo = 0, hi = cn;
        const double lim = qm - tol;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (lim > (double)W.c_mz[mid] + md)
            lo = mid + 1;
          else
            hi = mid;
        }
        int j = min(lo, cn - 1);
        // cpp:49-55
        while (j < cn && fabs(qm - ((double)W.c_mz[j] + md)) <= tol) {
          const int chg = W.c_chg[j];
          double mult = 0.0;
          if (s == 0 || chg == s)
            mult = 1.0;
          else if (chg == 0)
            mult = 2.0 / 3.0;
          if (mult > 0.0) {
            const float prod = (float)(mult * (double)q_int * (double)W.c_int[j]);  // cpp:81
            const int slot = atomicAdd(&W.counter, 1);
            if (slot < RS_MCAP) {
              const uint32_t gen = (uint32_t)((qi * S + s) * cn + j);
              W.keys[slot] = ((unsigned long long)__float_as_uint(prod) << 32) |
                             (unsigned long long)(0xFFFFFFFFu - gen);
              W.pay[slot] = ((uint32_t)qi << 16) | (uint32_t)j;
            }
          }
          ++j;
        }
      }
    }
  }
  wave_sync();
  return resolve_matches<EMIT>(lane, W, out_pairs, out_cap, out_count, status);
}

__device__ __forceinline__ void load_query(int tid, int nthreads, const DevPeaks &Qs, int q,
                                           QueryLds &Q, int &qn, int *status) {
  const int qo = Qs.offsets[q];
  qn = Qs.offsets[q + 1] - qo;
  if (qn > RS_MAXP) {
    if (tid == 0) atomicOr(status, RS_STATUS_PEAKS);
    qn = RS_MAXP;
  }
  for (int i = tid; i < qn; i += nthreads) {
    Q.mz[i] = Qs.mz[qo + i];
    Q.inten[i] = Qs.intensity[qo + i];
  }
}

// Candidate addressing: CSR (cand_offsets != null) or fixed stride.
struct CandView {
  const int64_t *rows64;
  const int32_t *rows32;
  const int32_t *offsets;
  int32_t stride;
  __device__ __forceinline__ void range(int q, long long &c0, long long &c1) const {
    if (offsets) {
      c0 = offsets[q];
      c1 = offsets[q + 1];
    } else {
      c0 = (long long)q * stride;
      c1 = c0 + stride;
    }
  }
  __device__ __forceinline__ long long row(long long c) const {
    return rows64 ? rows64[c] : (long long)rows32[c];
  }
};

__global__ __launch_bounds__(64 * RS_WAVES) void rescore_score_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, double tol, int allow_shift,
    double *__restrict__ pair_score, int *status) {
  __shared__ QueryLds Q;
  __shared__ WaveLds W[RS_WAVES];
  const int q = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long c0, c1;
  cv.range(q, c0, c1);
  if (c0 >= c1) return;
  int qn;
  load_query(threadIdx.x, blockDim.x, Qs, q, Q, qn, status);
  __syncthreads();
  const double q_pmz = Qs.precursor_mz[q];
  const long long step = (long long)RS_WAVES * gridDim.y;
  for (long long c = c0 + (long long)blockIdx.y * RS_WAVES + wave; c < c1; c += step) {
    const long long row = cv.row(c);
    double s = -1.0;
    if (row >= 0 && row < L.n)
      s = dot_pair_wave<false>(lane, Q, qn, q_pmz, L, (int)row, tol, allow_shift, W[wave],
                               nullptr, 0, nullptr, status);
    if (lane == 0) pair_score[c] = s;
  }
}

// ---------------------------------------------------------------------------------
// Pass 1, fast path. The binary-search formulation above spends its time in dependent
// LDS round trips (6 per shift per query peak). Here the QUERY is hashed once per
// workgroup instead: every query peak is entered into an LDS hash table under the
// m/z bins of width 2*tol that its window [mz-tol, mz+tol] touches; a candidate peak at
// shifted position x can only match peaks filed under floor(x / (2*tol)). Lanes are the
// candidate's peaks (loaded straight from HBM, coalesced), each does one independent
// probe per shift, and the reference's exact window/cursor predicate is applied to the
// few hits:  match(i,j,s)  <=>  |q_i - x_j| <= tol  and  (not(q_i - tol > x_j) or j = n_c-1)
// and every peak between the cursor and j passes the window test (SpectrumMatch.cpp:39-55).
// A wave owns a contiguous slice of the candidate list: 64 candidates' metadata are
// gathered at once, the next candidate's peaks are prefetched while one is scored.
constexpr int RS_HT = 512;       // hash slots
constexpr int RS_HQ_MAX = 100;   // query peaks the hash path accepts (<= 3 bins each)
constexpr int RS_EMPTY = (int)0x80000000;

struct HashLds {
  int bin[RS_HT];
  int peak[RS_HT];
};

__device__ __forceinline__ uint32_t hbin(int b) {
  return ((uint32_t)b * 2654435761u) >> 23;  // 9 bits
}

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ long long rl_ll(long long v, int l) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(unsigned long long)v, l);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)((unsigned long long)v >> 32), l);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double rl_d(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, l);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

constexpr int RS_PF = 4;   // candidates staged per burst (4 x 64 peaks = WaveLds::c_mz/c_int/c_chg)  // candidates whose peaks are in flight ahead of the scoring

// One (query, candidate) pair on the hash path; lanes are the candidate's peaks.
__device__ __forceinline__ double score_candidate(int lane, const QueryLds &Q, const HashLds &H,
                                                  WaveLds &Wv, int slot, int cn, int c_charge,
                                                  double c_pmz, double q_pmz, double tol,
                                                  double inv_w, int allow_shift, int *status) {
  if (lane == 0) Wv.counter = 0;
  const double pmd = (q_pmz - c_pmz) * (double)(unsigned)c_charge;    // cpp:18
  const int S = (allow_shift && fabs(pmd) >= tol) ? c_charge + 1 : 1;  // cpp:20
  const float *s_mz = Wv.c_mz + slot * 64;   // this candidate's staged peaks (cn <= 64)
  const float a_mz = lane < cn ? s_mz[lane] : 0.0f;
  const float a_int = lane < cn ? Wv.c_int[slot * 64 + lane] : 0.0f;
  const int a_chg = lane < cn ? Wv.c_chg[slot * 64 + lane] : 0;
  // mass_diff[s] = pmd / s (cpp:26-31): lane s does the (expensive, exact) fp64 division
  // once, every lane then reads the quotient it needs with v_readlane
  const double md_lane = (lane > 0 && lane < S) ? pmd / (double)lane : 0.0;
  const float inv_w_f = (float)inv_w;
  {
    const int j = lane;
    const bool act = j < cn;
    const float cm = a_mz, ci = a_int;
    const int cc = a_chg;
    for (int s = 0; s < S; ++s) {      // wave-uniform trip count
      const double md = rl_d(md_lane, s);
      if (act) {
        double mult = 0.0;
        if (s == 0 || cc == s)
          mult = 1.0;
        else if (cc == 0)
          mult = 2.0 / 3.0;
        if (mult == 0.0) continue;  // this peak cannot pair under shift s (cpp:58-75)
        // bin of the shifted peak in fp32 (the query was filed with a margin that covers
        // the fp32 rounding); the exact fp64 window test runs on the rare hits only
        const int b = (int)floorf((cm + (float)md) * inv_w_f);
        uint32_t h = hbin(b);
        for (;;) {
          const int eb = H.bin[h];
          if (eb == RS_EMPTY) break;
          if (eb == b) {
            const int i = H.peak[h];
            const double x = (double)cm + md;
            const double qm = (double)Q.mz[i];
            const double lim = qm - tol;
            if (fabs(qm - x) <= tol && (!(lim > x) || j == cn - 1)) {
              // the reference walks from its cursor: every peak between the cursor and j
              // must pass the window test too (differs only on fp boundaries)
              bool run = true;
              for (int jj = j; jj > 0; --jj) {
                const double xp = (double)s_mz[jj - 1] + md;
                if (lim > xp) break;
                if (!(fabs(qm - xp) <= tol)) {
                  run = false;
                  break;
                }
              }
              if (run) {
                const float prod = (float)(mult * (double)Q.inten[i] * (double)ci);
                const int mslot = atomicAdd(&Wv.counter, 1);
                if (mslot < RS_MCAP) {
                  const uint32_t gen = (uint32_t)((i * S + s) * cn + j);
                  Wv.keys[mslot] = ((unsigned long long)__float_as_uint(prod) << 32) |
                                   (unsigned long long)(0xFFFFFFFFu - gen);
                  Wv.pay[mslot] = ((uint32_t)i << 16) | (uint32_t)j;
                }
              }
            }
          }
          h = (h + 1) & (RS_HT - 1);
        }
      }
    }
  }
  wave_sync();
  return resolve_matches<false>(lane, Wv, nullptr, 0, nullptr, status);
}

__global__ __launch_bounds__(64 * RS_WAVES) void rescore_score_v2_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, double tol, int allow_shift,
    double *__restrict__ pair_score, int *status) {
  __shared__ QueryLds Q;
  __shared__ HashLds H;
  __shared__ WaveLds W[RS_WAVES];
  const int q = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  long long c0, c1;
  cv.range(q, c0, c1);
  if (c0 >= c1) return;
  int qn;
  load_query(tid, blockDim.x, Qs, q, Q, qn, status);
  for (int i = tid; i < RS_HT; i += blockDim.x) H.bin[i] = RS_EMPTY;
  __syncthreads();
  // fp32 evaluation of a probe bin (m/z <= ~2600): error bound in bin units; the query
  // peaks are filed with that margin on both sides (still <= 3 bins per peak)
  const double margin = 1e-3 + (tol > 0.0 ? (0.5 / tol) * (3.75e-4 + 2600.0 * 1.2e-7) : 1.0);
  if (!(tol > 0.0) || qn > RS_HQ_MAX || margin > 0.45) {  // uniform: binary-search formulation
    const double q_pmz0 = Qs.precursor_mz[q];
    const long long step = (long long)RS_WAVES * gridDim.y;
    for (long long c = c0 + (long long)blockIdx.y * RS_WAVES + wave; c < c1; c += step) {
      const long long row = cv.row(c);
      double sc = -1.0;
      if (row >= 0 && row < L.n)
        sc = dot_pair_wave<false>(lane, Q, qn, q_pmz0, L, (int)row, tol, allow_shift, W[wave],
                                  nullptr, 0, nullptr, status);
      if (lane == 0) pair_score[c] = sc;
    }
    return;
  }
  const double inv_w = 1.0 / (2.0 * tol);
  if (tid < qn) {
    const double qm = (double)Q.mz[tid];
    const int blo = (int)floor((qm - tol) * inv_w - margin);
    const int bhi = (int)floor((qm + tol) * inv_w + margin);
    for (int b = blo; b <= bhi; ++b) {
      uint32_t h = hbin(b);
      for (;;) {
        const int old = atomicCAS(&H.bin[h], RS_EMPTY, b);
        if (old == RS_EMPTY) {
          H.peak[h] = tid;
          break;
        }
        h = (h + 1) & (RS_HT - 1);
      }
    }
  }
  __syncthreads();
  const double q_pmz = Qs.precursor_mz[q];
  WaveLds &Wv = W[wave];

  // contiguous slice of the candidate list for this wave
  const long long n = c1 - c0;
  const long long parts = (long long)RS_WAVES * gridDim.y;
  const long long per = (n + parts - 1) / parts;
  const long long wb = c0 + ((long long)blockIdx.y * RS_WAVES + wave) * per;
  const long long we = wb + per < c1 ? wb + per : c1;

  for (long long base = wb; base < we; base += 64) {
    const long long c = base + lane;
    long long row = -1;
    if (c < we) row = cv.row(c);
    const bool okr = row >= 0 && row < L.n;
    int m_co = 0, m_cn = 0, m_chg = 0;
    double m_pmz = 0.0;
    if (okr) {
      m_co = L.offsets[row];
      m_cn = L.offsets[row + 1] - m_co;
      m_chg = L.precursor_charge[row];
      m_pmz = L.precursor_mz[row];
    }
    double my_score = -1.0;
    const int cnt = (int)((we - base) < 64 ? (we - base) : 64);
    // Bursts of RS_PF candidates: all their peak loads are issued back to back (straight
    // line, so hipcc keeps them in flight together), parked in this wave's LDS staging
    // area, and the candidates are then scored from LDS with no global load in the way.
    for (int g0 = 0; g0 < cnt; g0 += RS_PF) {
      float fm[RS_PF], fi[RS_PF];
      int fc[RS_PF];
#pragma unroll
      for (int u = 0; u < RS_PF; ++u) {
        const int l = g0 + u < cnt ? g0 + u : cnt - 1;
        const int co = rl_i(m_co, l), cn = rl_i(m_cn, l);
        const bool ld = lane < cn && g0 + u < cnt;
        fm[u] = ld ? L.mz[co + lane] : 0.0f;
        fi[u] = ld ? L.intensity[co + lane] : 0.0f;
        fc[u] = (ld && L.charge) ? L.charge[co + lane] : 0;
      }
#pragma unroll
      for (int u = 0; u < RS_PF; ++u) {
        Wv.c_mz[u * 64 + lane] = fm[u];
        Wv.c_int[u * 64 + lane] = fi[u];
        Wv.c_chg[u * 64 + lane] = (uint8_t)fc[u];
      }
      wave_sync();
#pragma unroll
      for (int u = 0; u < RS_PF; ++u) {
        const int l = g0 + u;
        if (l < cnt && rl_i((int)okr, l)) {   // wave-uniform
          const int cn = rl_i(m_cn, l);
          const int c_charge = rl_i(m_chg, l);
          const double c_pmz = rl_d(m_pmz, l);
          double score = 0.0;
          if (cn > 64) {   // rare: more peaks than a staging slot holds -> after the bursts
            score = -2.0;
          } else if (cn > 0 && qn > 0) {
            score = score_candidate(lane, Q, H, Wv, u, cn, c_charge, c_pmz, q_pmz, tol, inv_w,
                                    allow_shift, status);
          }
          if (lane == l) my_score = score;
        }
      }
      wave_sync();
    }
    // candidates with more than 64 peaks: binary-search formulation (uses the whole staging area)
    unsigned long long big = __ballot(my_score == -2.0);
    while (big) {
      const int l = __builtin_ctzll(big);
      big &= big - 1;
      const double sc = dot_pair_wave<false>(lane, Q, qn, q_pmz, L, (int)rl_ll(row, l), tol,
                                             allow_shift, Wv, nullptr, 0, nullptr, status);
      if (lane == l) my_score = sc;
    }
    if (c < we) pair_score[c] = my_score;
  }
}

// tie_by_row = 0: first position wins ties (get_best_match on a caller-ordered list);
// tie_by_row = 1: lowest library row wins ties (the reference's lists ascend in row,
// spectral_library.py:451, so this is the same rule for an unordered ANN list).
__global__ __launch_bounds__(64) void rescore_argmax_kernel(
    CandView cv, int nq, const double *__restrict__ pair_score, int tie_by_row,
    int32_t *__restrict__ best_cand, long long *__restrict__ best_slot,
    double *__restrict__ best_score, int32_t *__restrict__ n_valid) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  long long c0, c1;
  cv.range(q, c0, c1);
  double bs = -1.0;
  long long bkey = 0x7fffffffffffffffll, bpos = -1;
  int cnt = 0;
  for (long long c = c0 + lane; c < c1; c += 64) {
    const double s = pair_score[c];
    if (s < 0.0) continue;
    ++cnt;
    const long long key = tie_by_row ? cv.row(c) : c;
    if (s > bs || (s == bs && key < bkey)) {
      bs = s;
      bkey = key;
      bpos = c;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double os = __shfl_xor(bs, off);
    const long long ok = __shfl_xor(bkey, off);
    const long long op = __shfl_xor(bpos, off);
    cnt += __shfl_xor(cnt, off);
    if (os > bs || (os == bs && ok < bkey)) {
      bs = os;
      bkey = ok;
      bpos = op;
    }
  }
  if (lane == 0) {
    if (best_cand) best_cand[q] = bpos >= 0 ? (int32_t)(bpos - c0) : -1;
    if (best_slot) best_slot[q] = bpos;
    if (best_score) best_score[q] = bpos >= 0 ? bs : 0.0;
    if (n_valid) n_valid[q] = cnt;
  }
}

__global__ __launch_bounds__(64 * RS_WAVES) void rescore_matches_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, int nq, const long long *__restrict__ best_slot,
    double tol, int allow_shift, int32_t *__restrict__ pm_count,
    uint32_t *__restrict__ pm_pairs, int pm_stride, int32_t *__restrict__ best_row,
    int *status) {
  __shared__ QueryLds Q[RS_WAVES];
  __shared__ WaveLds W[RS_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * RS_WAVES + wave;
  if (q >= nq) return;
  const long long slot = best_slot[q];
  const long long row = slot >= 0 ? cv.row(slot) : -1;
  if (best_row && lane == 0) best_row[q] = (int32_t)row;
  if (row < 0) {
    if (pm_count && lane == 0) pm_count[q] = 0;
    return;
  }
  if (!pm_count && !pm_pairs) return;
  int qn;
  load_query(lane, 64, Qs, q, Q[wave], qn, status);
  wave_sync();
  int cnt_tmp = 0;
  __shared__ int s_cnt[RS_WAVES];
  dot_pair_wave<true>(lane, Q[wave], qn, Qs.precursor_mz[q], L, (int)row, tol, allow_shift,
                      W[wave], pm_pairs ? pm_pairs + (size_t)q * pm_stride * 2 : nullptr,
                      pm_pairs ? pm_stride : 0, &s_cnt[wave], status);
  wave_sync();
  cnt_tmp = s_cnt[wave];
  if (pm_count && lane == 0) pm_count[q] = cnt_tmp;
}

// Host driver shared by asl_rescore_batch and asl_search_batch. All pointers are
// device pointers. pair_score scratch must hold one double per candidate slot.
int rescore_device(const DevPeaks &Q, const DevPeaks &L, const int64_t *rows64,
                   const int32_t *rows32, const int32_t *cand_offsets, int32_t stride,
                   int64_t total_slots, double tol, int allow_shift, int tie_by_row,
                   double *pair_score, long long *best_slot, int32_t *best_cand,
                   int32_t *best_row, double *best_score, int32_t *n_valid,
                   int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride, int *status) {
  const int nq = Q.n;
  if (nq == 0) return ASL_OK;
  CandView cv{rows64, rows32, cand_offsets, stride};
  HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), stream()));
  {
    ProfScope ps("rescore");
    // split long candidate lists over blockIdx.y when there are few queries
    int64_t avg = total_slots / (nq > 0 ? nq : 1);
    int ysplit = 1;
    if (nq < 2048 && avg > 4096) ysplit = (int)std::min<int64_t>(64, cdiv(avg, 4096));
    // the v2 kernel falls back to the binary-search formulation per query when tol <= 0 or
    // the query has more than RS_HQ_MAX peaks
    static const bool force_v1 = getenv("ASL_RESCORE_V1") != nullptr;  // A/B knob
    if (!force_v1)
      hipLaunchKernelGGL(rescore_score_v2_kernel, dim3(nq, ysplit), dim3(64 * RS_WAVES), 0,
                         stream(), Q, L, cv, tol, allow_shift, pair_score, status);
    else
      hipLaunchKernelGGL(rescore_score_kernel, dim3(nq, ysplit), dim3(64 * RS_WAVES), 0,
                         stream(), Q, L, cv, tol, allow_shift, pair_score, status);
    ASL_CHECK_LAUNCH();
    hipLaunchKernelGGL(rescore_argmax_kernel, dim3(nq), dim3(64), 0, stream(), cv, nq,
                       pair_score, tie_by_row, best_cand, best_slot, best_score, n_valid);
    ASL_CHECK_LAUNCH();
  }
  {
    ProfScope ps("rescore_matches");
    hipLaunchKernelGGL(rescore_matches_kernel, dim3((unsigned)cdiv(nq, RS_WAVES)),
                       dim3(64 * RS_WAVES), 0, stream(), Q, L, cv, nq, best_slot, tol,
                       allow_shift, pm_count, pm_pairs, pm_stride, best_row, status);
    ASL_CHECK_LAUNCH();
  }
  return ASL_OK;
}

int rescore_check_status(const int *status_dev) {
  int st = 0;
  HIP_TRY(hipMemcpyAsync(&st, status_dev, sizeof(int), hipMemcpyDeviceToHost, stream()));
  ASL_TRY(sync_stream());
  if (st & RS_STATUS_PEAKS)
    return fail(ASL_ERR_CAPACITY, "rescore: a spectrum has more than %d peaks", RS_MAXP);
  if (st & RS_STATUS_MATCHES)
    return fail(ASL_ERR_CAPACITY, "rescore: a pair generated more than %d peak matches", RS_MCAP);
  return ASL_OK;
}

}  // namespace asl

using namespace asl;

extern "C" int asl_rescore_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                                 const int64_t *cand_rows, const int32_t *cand_offsets,
                                 double tol, int allow_shift, int32_t *best_cand,
                                 double *best_score, int32_t *pm_count, uint32_t *pm_pairs,
                                 int32_t pm_stride) {
  clear_error();
  if (!queries || !library) return fail(ASL_ERR_INVALID, "rescore_batch: null spectra");
  const int nq = queries->n;
  if (nq == 0) return ASL_OK;
  if (!cand_offsets) return fail(ASL_ERR_INVALID, "rescore_batch: null cand_offsets");
  if (pm_pairs && pm_stride <= 0) return fail(ASL_ERR_INVALID, "rescore_batch: pm_stride");
  ASL_TRY(ensure_device());
  PeaksStage Q, L;
  ASL_TRY(Q.init(queries));
  ASL_TRY(L.init(library));
  In<int32_t> off;
  ASL_TRY(off.init(cand_offsets, (size_t)nq + 1));
  int32_t total = 0;
  if (is_device_ptr(cand_offsets)) {
    HIP_TRY(hipMemcpyAsync(&total, cand_offsets + nq, sizeof(int32_t), hipMemcpyDeviceToHost,
                           stream()));
    ASL_TRY(sync_stream());
  } else {
    total = cand_offsets[nq];
  }
  if (total < 0) return fail(ASL_ERR_INVALID, "rescore_batch: negative cand_offsets");
  if (total > 0 && !cand_rows) return fail(ASL_ERR_INVALID, "rescore_batch: null cand_rows");
  In<int64_t> rows;
  ASL_TRY(rows.init(cand_rows, (size_t)total));
  Out<int32_t> o_best, o_cnt;
  Out<double> o_score;
  Out<uint32_t> o_pairs;
  ASL_TRY(o_best.init(best_cand, nq));
  ASL_TRY(o_score.init(best_score, nq));
  ASL_TRY(o_cnt.init(pm_count, nq));
  ASL_TRY(o_pairs.init(pm_pairs, (size_t)nq * (pm_pairs ? pm_stride : 0) * 2));
  DevBuf<double> pair_score;
  DevBuf<long long> best_slot;
  DevBuf<int> status;
  ASL_TRY(pair_score.reserve((size_t)std::max(total, 1)));
  ASL_TRY(best_slot.reserve(nq));
  ASL_TRY(status.reserve(1));
  ASL_TRY(rescore_device(Q.dev, L.dev, rows.d, nullptr, off.d, 0, total, tol, allow_shift, 0,
                         pair_score.p, best_slot.p, o_best.d, nullptr, o_score.d, nullptr,
                         o_cnt.d, o_pairs.d, pm_stride, status.p));
  ASL_TRY(o_best.finish());
  ASL_TRY(o_score.finish());
  ASL_TRY(o_cnt.finish());
  ASL_TRY(o_pairs.finish());
  return rescore_check_status(status.p);  // synchronises
}
