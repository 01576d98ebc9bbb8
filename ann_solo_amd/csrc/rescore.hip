// rescore.hip -- batched (shifted) dot-product rescoring.
// Replaces get_best_match (/root/reference/src/ann_solo/spectrum_match.pyx:28-108)
// and SpectrumMatcher::dot (/root/reference/src/ann_solo/SpectrumMatch.cpp:8-133).
//
// Pass 1a (rescore_flat_kernel): one workgroup per query. The query's peaks are hashed once
// into LDS (m/z bins of width 2*tol + a bitmap of occupied bins); the candidate list is
// filtered (precursor window, rescore.hpp) and compacted in LDS; a wave sets 32 candidates up
// one lane each and walks their peaks (m/z and fragment charge) as ONE stream, a peak per lane,
// testing the bitmap for every shift; the (peak, shift) items whose bin is marked are queued
// and probed -- hash walk, exact window / cursor predicate -- a full row of lanes at a time;
// a match is accumulated where it is found (LDS atomics: masks of matched peaks, exponent
// range, exact fp64 sum). Candidates with a doubly matched peak are marked for
// Pass 1b (rescore_score_v2_kernel, the pair kernel): two candidates per wave, one per
// half-wave, matches into per-half LDS lists keyed (product desc, generation order asc),
// resolved by a conflict-free fast path or a bitonic sort + scalar greedy loop. What neither
// takes (> 64 peaks or matches, > 100 query peaks, tol <= 0) goes to
// Pass 1c (rescore_score_kernel): one wave per (query, candidate) pair, lanes = query
// peaks, cursor by binary search (same cursor value as the reference's running cursor
// because both peak lists ascend).
// Pass 2 (rescore_argmax_kernel): per query first-strict-maximum (cpp:118-129).
// Pass 3 (rescore_matches_kernel): the winning pair is re-run once per query to
// emit its peak_matches in greedy order.
//
// Arithmetic mirrors the reference: window tests in double on float->double
// promoted m/z (cpp:42,53); product = (float)(mult * (double)q_int * (double)c_int)
// (cpp:81); score = double sum of those floats in sorted order (cpp:104).
#include <cstdlib>

#include "common.hpp"
#include "rescore.hpp"

namespace asl {

constexpr int RS_WAVES = 4;
#ifndef RS_OCC
#define RS_OCC 7   // waves per SIMD the hash kernel is built for (A/B: -DRS_OCC=6)
#endif
constexpr int RS_MAXP = 256;   // peaks per spectrum the kernels accept
constexpr int RS_MCAP = 512;   // generated peak matches per pair the kernels accept

enum { RS_STATUS_OK = 0, RS_STATUS_PEAKS = 1, RS_STATUS_MATCHES = 2 };

template <int MAXP_, int MCAP_>
struct WaveLdsT {
  static constexpr int MAXP = MAXP_, MCAP = MCAP_;
  float c_mz[MAXP_];
  float c_int[MAXP_];
  unsigned long long keys[MCAP_];
  uint32_t pay[MCAP_];
  uint8_t c_chg[MAXP_];
  uint8_t own_q[RS_MAXP];   // resolve fast path: last lane that claimed a query / candidate peak
  uint8_t own_c[RS_MAXP];
  int counter;
  int pad[3];
};
typedef WaveLdsT<RS_MAXP, RS_MCAP> WaveLds;

template <int MAXP_>
struct QueryLdsT {
  static constexpr int MAXP = MAXP_;
  float mz[MAXP_];
  float inten[MAXP_];
};
typedef QueryLdsT<RS_MAXP> QueryLds;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ uint32_t rs_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
}

// max over the wave, in every lane (row butterflies on DPP, rows joined on the scalar unit)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, rs_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
  v = max(v, rs_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
  v = max(v, rs_dpp<0x141>(v));   // row_half_mirror
  v = max(v, rs_dpp<0x140>(v));   // row_mirror
  const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

template <int CTRL>
__device__ __forceinline__ double rs_dpp_d(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = rs_dpp<CTRL>((uint32_t)u), hi = rs_dpp<CTRL>((uint32_t)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Sum over the wave in a fixed tree order; callers use it only where the sum is exact.
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += rs_dpp_d<0xB1>(v);
  v += rs_dpp_d<0x4E>(v);
  v += rs_dpp_d<0x141>(v);
  v += rs_dpp_d<0x140>(v);
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  double r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t l = __builtin_amdgcn_readlane(lo, 16 * i), h = __builtin_amdgcn_readlane(hi, 16 * i);
    r[i] = __longlong_as_double((long long)(((unsigned long long)h << 32) | l));
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// Sort the wave's match list (product desc, generation order asc) and assign greedily
// (SpectrumMatch.cpp:92-111). Lists of <= 64 matches are sorted in registers with
// cross-lane shuffles; longer ones in LDS. Returns the score in every lane.
template <bool EMIT, class WL>
__device__ double resolve_matches(int lane, WL &W, uint32_t *out_pairs, int out_cap,
                                  int *out_count, int *status, int kbase = 0, int m_in = -1,
                                  int mcap = RS_MCAP) {
  int M = m_in >= 0 ? m_in : W.counter;
  M = __builtin_amdgcn_readfirstlane(M);
  if (M > mcap) {
    if (lane == 0) atomicOr(status, RS_STATUS_MATCHES);
    M = mcap;
  }
  if (M == 0) return 0.0;
  unsigned long long *wkeys = W.keys + kbase;
  uint32_t *wpay = W.pay + kbase;
  unsigned long long rkey = 0ull;
  uint32_t rpay = 0;
  const bool small = M <= 64;
  if (small) {
    if (lane < M) {
      rkey = wkeys[lane];
      rpay = wpay[lane];
    }
    if (!EMIT && M > 1) {
      // Fast path: if no query peak and no candidate peak occurs twice, the greedy pass
      // accepts every match and the score is the plain sum of the products. The products
      // are fp32; when their exponents span <= 23 binades every partial sum of <= 64 of
      // them is exactly representable in fp64, so the sum does not depend on the order
      // and equals the reference's sorted accumulation bit for bit.
      const bool inl = lane < M;
      const uint32_t qi = (rpay >> 16) & (RS_MAXP - 1), ci = rpay & (RS_MAXP - 1);
      if (inl) {
        W.own_q[qi] = (uint8_t)lane;
        W.own_c[ci] = (uint8_t)lane;
      }
      wave_sync();
      const bool mine = !inl || (W.own_q[qi] == (uint8_t)lane && W.own_c[ci] == (uint8_t)lane);
      const uint32_t e = inl ? max((uint32_t)(rkey >> 55) & 0xffu, 1u) : 0u;
      const uint32_t emax = wave_max_u32(e);
      const bool exact = !inl || e + 23u >= emax;
      if (!__ballot(!(mine && exact)))
        return wave_sum_f64(inl ? (double)__uint_as_float((uint32_t)(rkey >> 32)) : 0.0);
    }
    if (M > 1) {
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
        if ((k >> 1) >= M) break;   // sorted runs of nextpow2(M) lanes suffice (wave-uniform)
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
    for (int i = M + lane; i < P; i += 64) wkeys[i] = 0ull;
    wave_sync();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < (P >> 1); t += 64) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const unsigned long long a = wkeys[i], b = wkeys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            wkeys[i] = b;
            wkeys[l] = a;
            const uint32_t pa = wpay[i];
            wpay[i] = wpay[l];
            wpay[l] = pa;
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
      pl = __builtin_amdgcn_readfirstlane(wpay[t]);
      pb = __builtin_amdgcn_readfirstlane((uint32_t)(wkeys[t] >> 32));
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
// DEFER (the small-LDS instantiation of the matches kernel): a pair that does not fit the
// structures -- more candidate peaks or generated matches than they hold -- is not an error, the
// call returns -1 and the pair is left to the full-size instantiation.
template <bool EMIT, bool DEFER = false, class QL = QueryLds, class WL = WaveLds>
__device__ double dot_pair_wave(int lane, const QL &Q, int qn, double q_pmz,
                                const DevPeaks &L, int row, double tol, int allow_shift,
                                WL &W, uint32_t *out_pairs, int out_cap, int *out_count,
                                int *status) {
  const int co = L.offsets[row];
  int cn = L.offsets[row + 1] - co;
  if (EMIT && out_count && lane == 0) *out_count = 0;
  if (cn > WL::MAXP) {
    if (DEFER) return -1.0;
    if (lane == 0) atomicOr(status, RS_STATUS_PEAKS);
    cn = WL::MAXP;
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
            if (slot < WL::MCAP) {
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
  if (DEFER && W.counter > WL::MCAP) return -1.0;     // (wave-uniform: the counter is final)
  return resolve_matches<EMIT>(lane, W, out_pairs, out_cap, out_count, status, 0, -1, WL::MCAP);
}

template <class QL>
__device__ __forceinline__ void load_query(int tid, int nthreads, const DevPeaks &Qs, int q,
                                           QL &Q, int &qn, int *status) {
  const int qo = Qs.offsets[q];
  qn = Qs.offsets[q + 1] - qo;
  if (qn > QL::MAXP) {
    if (tid == 0) atomicOr(status, RS_STATUS_PEAKS);
    qn = QL::MAXP;
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
  PrecFilter flt;
  // fixed-stride rows whose length the producer wrote (the scans' post-filter, common.hpp:
  // ScanPostFilter): counts[q] >= 0: the row holds that many hits, ALREADY filtered by the
  // precursor window; -1: the row holds `stride` unfiltered hits (filter here, as without counts)
  const int32_t *counts = nullptr;
  __device__ __forceinline__ bool prefiltered(int q) const { return counts != nullptr && counts[q] >= 0; }
  // row of slot c if it is a candidate of the query (in range, passes the filter), else -1
  __device__ __forceinline__ long long cand(long long c, double q_pmz, int n_lib) const {
    const long long r = row(c);
    return (r >= 0 && r < n_lib && filter_pass(flt, q_pmz, r)) ? r : -1;
  }
  __device__ __forceinline__ void range(int q, long long &c0, long long &c1) const {
    if (offsets) {
      c0 = offsets[q];
      c1 = offsets[q + 1];
    } else {
      c0 = (long long)q * stride;
      int len = stride;
      if (counts) {
        const int c = counts[q];
        if (c >= 0) len = c < stride ? c : stride;
      }
      c1 = c0 + len;
    }
  }
  __device__ __forceinline__ long long row(long long c) const {
    return rows64 ? rows64[c] : (long long)rows32[c];
  }
};

constexpr int RS_BS_GROUP = 16;      // queries per workgroup of the binary-search kernel
__global__ __launch_bounds__(64 * RS_WAVES) void rescore_score_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, double tol, int allow_shift,
    double *__restrict__ pair_score, const int *__restrict__ q_defer, int *status, int group) {
  __shared__ QueryLds Q;
  __shared__ WaveLds W[RS_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // A workgroup takes `group` (RS_BS_GROUP; 1 for small batches) consecutive queries: as the third launch of the rescoring the
  // kernel normally finds nothing marked -- one coalesced read of the group's flags says so (a
  // workgroup per query, 46 KB of LDS each, placed only to return, cost 0.02 ms of a step); when
  // every query is marked (tol <= 0, long queries) nq / 16 workgroups still fill the chip.
  // deferred mode: only the pairs the fast kernels marked RS_DEFER_BS (-3)
  __shared__ unsigned long long s_todo;
  const int q0 = blockIdx.x * group;
  if (wave == 0) {
    const int qt = q0 + lane;
    const unsigned long long m = __ballot(lane < group && qt < Qs.n && (!q_defer || (q_defer[qt] & 2)));
    if (lane == 0) s_todo = m;
  }
  __syncthreads();
  {
   unsigned long long todo = s_todo;          // workgroup-uniform
   while (todo) {
    const int q = q0 + __builtin_ctzll(todo);
    todo &= todo - 1ull;
    long long c0, c1;
    cv.range(q, c0, c1);
    if (c0 >= c1) continue;
    int qn;
    __syncthreads();              // the previous query's peaks are no longer read
    load_query(threadIdx.x, blockDim.x, Qs, q, Q, qn, status);
    __syncthreads();
    const double q_pmz = Qs.precursor_mz[q];
    const long long step = (long long)RS_WAVES * gridDim.y;
    for (long long c = c0 + (long long)blockIdx.y * RS_WAVES + wave; c < c1; c += step) {
      if (q_defer && pair_score[c] != -3.0) continue;
      const long long row = cv.cand(c, q_pmz, L.n);
      double s = -1.0;
      if (row >= 0)
        s = dot_pair_wave<false>(lane, Q, qn, q_pmz, L, (int)row, tol, allow_shift, W[wave],
                                 nullptr, 0, nullptr, status);
      if (lane == 0) pair_score[c] = s;
    }
   }
  }
}

// ---------------------------------------------------------------------------------
// Pass 1, fast path. The binary-search formulation above spends its time in dependent
// LDS round trips (6 per shift per query peak). Here the QUERY is hashed once per
// workgroup instead: every query peak is entered into an LDS hash table under the
// m/z bins of width 2*tol that its window [mz-tol, mz+tol] touches; a candidate peak at
// shifted position x can only match peaks filed under floor(x / (2*tol)). Lanes are the
// candidate's peaks, each does one independent probe per shift (a one-word Bloom-style
// bitmap rejects most probes), and the reference's exact window/cursor predicate is
// applied to the few hits:
//   match(i,j,s) <=> |q_i - x_j| <= tol and (not(q_i - tol > x_j) or j = n_c-1)
// and every peak between the cursor and j passes the window test (SpectrumMatch.cpp:39-55).
//
// Work distribution: the valid slots of the candidate list are compacted in LDS and dealt
// out evenly to the waves; a wave gathers 64 candidates' metadata at once, stages the
// peaks of RS_PF candidates per burst in LDS and scores them two at a time (one candidate
// per half-wave: an average library spectrum has ~27 peaks).
//
// The kernel only handles what fits its small LDS budget (4 workgroups per CU):
// candidates with <= 64 peaks and <= RS_HC generated matches, queries with <= RS_HQ_MAX
// peaks, tol > 0. Everything else is marked RS_DEFER and scored by the binary-search
// kernel in a second launch that skips queries with nothing deferred.
constexpr int RS_HT = 512;            // hash slots
constexpr int RS_HQ_MAX = 100;        // query peaks the hash path accepts (<= 3 bins each)
constexpr int RS_EMPTY = (int)0x80000000;
constexpr int RS_BM_BITS = 1 << 14;   // bin filter: <= 300 bits set of 16 384
constexpr int RS_SUPER = 1024;        // candidate slots compacted at a time
constexpr int RS_PF = 2;              // candidates staged per burst
constexpr int RS_HC = 64;             // matches per candidate resolved in this kernel
constexpr double RS_DEFER = -2.0;     // pair_score marker: left to the pair kernel (second launch)
constexpr double RS_DEFER_BS = -3.0;  // pair_score marker: left to the binary-search kernel (third launch)
enum { RS_QD_PAIR = 1, RS_QD_BS = 2 };   // q_defer bits: the query has slots marked RS_DEFER / RS_DEFER_BS;
                                         // bits 2.. count the RS_DEFER slots (work split of the second launch)
constexpr int RS_DEF_Y = 8;             // blocks per query the second launch may use

struct QueryLds2 {   // hash path: at most RS_HQ_MAX query peaks
  float mz[128];
  float inten[128];
};

struct HashLds {
  int bin[RS_HT];
  uint32_t bm[RS_BM_BITS / 32];
  uint8_t peak[RS_HT];
};

struct PairLds {   // per wave
  float c_mz[RS_PF * 64];
  float c_int[RS_PF * 64];
  unsigned long long keys[2 * RS_HC];   // half A, half B
  uint32_t pay[2 * RS_HC];
  uint8_t c_chg[RS_PF * 64];
  uint8_t own_q[RS_MAXP];
  uint8_t own_c[RS_MAXP];
  double mdt[64];   // mass_diff[s] of half A (0..31) and half B (32..63)
  int counter;
  int pad[3];
};

// bit of a bin in the query's bin filter: the bin's low bits (bins 16 384 apart = 650 Da at the
// default tolerance share a bit; as dense as a multiplicative hash, without the quarter-rate multiply)
__device__ __forceinline__ uint32_t hbit(int b) { return (uint32_t)b & (uint32_t)(RS_BM_BITS - 1); }
__device__ __forceinline__ bool bm_test(const HashLds &H, int b) {
  const uint32_t bit = hbit(b);
  return (H.bm[bit >> 5] >> (bit & 31)) & 1u;
}

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

// Conflict-free resolve of BOTH half-wave match lists in one pass (each <= 32 matches):
// lanes 0-31 hold A's matches, lanes 32-63 B's; same test and same exact-sum argument as the
// fast path of resolve_matches, evaluated per half (row reductions on DPP, the two rows of a
// half joined on the scalar unit). ok bit 0 / 1: A / B resolved here (else the caller runs
// the general path for that half).
__device__ __forceinline__ int resolve_two_fast(int lane, PairLds &W, int MA, int MB,
                                                double &sA, double &sB) {
  const int half = lane >> 5, hl = lane & 31;
  const int M = half ? MB : MA;
  const bool inl = hl < M;
  unsigned long long key = 0ull;
  uint32_t pay = 0;
  if (inl) {
    key = W.keys[half * RS_HC + hl];
    pay = W.pay[half * RS_HC + hl];
  }
  // query peaks < 128 (RS_HQ_MAX) and candidate peaks < 64 on this path: one table per half
  const uint32_t qi = half * 128 + ((pay >> 16) & 127), ci = half * 128 + (pay & 127);
  if (inl) {
    W.own_q[qi] = (uint8_t)lane;
    W.own_c[ci] = (uint8_t)lane;
  }
  wave_sync();
  const bool mine = !inl || (W.own_q[qi] == (uint8_t)lane && W.own_c[ci] == (uint8_t)lane);
  uint32_t e = inl ? max((uint32_t)(key >> 55) & 0xffu, 1u) : 0u;
  uint32_t v = e;
  v = max(v, rs_dpp<0xB1>(v));
  v = max(v, rs_dpp<0x4E>(v));
  v = max(v, rs_dpp<0x141>(v));
  v = max(v, rs_dpp<0x140>(v));
  const uint32_t eA = max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
  const uint32_t eB = max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
  const bool exact = !inl || e + 23u >= (half ? eB : eA);
  const unsigned long long bad = __ballot(!(mine && exact));
  double x = inl ? (double)__uint_as_float((uint32_t)(key >> 32)) : 0.0;
  x += rs_dpp_d<0xB1>(x);
  x += rs_dpp_d<0x4E>(x);
  x += rs_dpp_d<0x141>(x);
  x += rs_dpp_d<0x140>(x);
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  double r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t l = __builtin_amdgcn_readlane(lo, 16 * i), h = __builtin_amdgcn_readlane(hi, 16 * i);
    r[i] = __longlong_as_double((long long)(((unsigned long long)h << 32) | l));
  }
  sA = r[0] + r[1];
  sB = r[2] + r[3];
  return ((bad & 0xffffffffull) ? 0 : 1) | ((bad >> 32) ? 0 : 2);
}

// General resolve of one half's match list, out of line: it is the rare path, and inlined its
// loop-invariant lane predicates (bitonic directions) get hoisted into scalar registers that
// the hot loops of the kernel then spill around.
__device__ __attribute__((noinline)) double resolve_half_slow(int lane, PairLds &W, int *status,
                                                              int kbase, int M) {
  return resolve_matches<false>(lane, W, nullptr, 0, nullptr, status, kbase, M, RS_HC);
}

// Two candidates per wave: lanes 0-31 score candidate A, lanes 32-63 candidate B (peaks in
// passes of 32; cn = 0 leaves a half idle). The probing instruction stream -- the bulk of
// this instruction-bound kernel -- is shared by both. Per-half match lists live in the two
// halves of the wave's key buffer. A score of RS_DEFER means the half overflowed RS_HC.
__device__ __forceinline__ void score_two(int lane, const QueryLds2 &Q, const HashLds &H,
                                          PairLds &Wv, int slotA, int slotB, int cnA, int cnB,
                                          int chgA, int chgB, double pmzA, double pmzB,
                                          double q_pmz, double tol, double inv_w, int allow_shift,
                                          int *status, double &scoreA, double &scoreB) {
  const int half = lane >> 5, hl = lane & 31;
  if (lane == 0) {
    Wv.pad[0] = 0;
    Wv.pad[1] = 0;
  }
  const double pmdA = (q_pmz - pmzA) * (double)(unsigned)chgA;          // cpp:18
  const double pmdB = (q_pmz - pmzB) * (double)(unsigned)chgB;
  const int SA = (allow_shift && fabs(pmdA) >= tol) ? chgA + 1 : 1;      // cpp:20
  const int SB = (allow_shift && fabs(pmdB) >= tol) ? chgB + 1 : 1;
  const int slot = half ? slotB : slotA, cn = half ? cnB : cnA, S = half ? SB : SA;
  const double pmd = half ? pmdB : pmdA;
  const float *s_mz = Wv.c_mz + slot * 64;
  const float *s_in = Wv.c_int + slot * 64;
  const uint8_t *s_ch = Wv.c_chg + slot * 64;
  // mass_diff[s] = pmd / s (cpp:26-31): lane (32*half + s) performs its half's (expensive,
  // exact) fp64 division once and parks the quotient in LDS for the other lanes
  const int Smax = SA > SB ? SA : SB;
  {
    // x / 1 and x / 2 are exact scalings: precursor charges <= 2 (S <= 3) never need the fp64
    // division sequence (~40 VALU instructions every lane would execute per pair); the branch is
    // wave-uniform. Identical bits either way.
    double mdv = 0.0;
    if (Smax <= 3) {
      if (hl > 0 && hl < S) mdv = hl == 1 ? pmd : pmd * 0.5;
    } else if (hl > 0 && hl < S) {
      mdv = hl == 1 ? pmd : hl == 2 ? pmd * 0.5 : hl == 4 ? pmd * 0.25 : pmd / (double)hl;
    }
    Wv.mdt[lane] = mdv;
  }
  const float inv_w_f = (float)inv_w;
  const int cmax = cnA > cnB ? cnA : cnB;
  unsigned long long *keys = Wv.keys + half * RS_HC;
  uint32_t *pay = Wv.pay + half * RS_HC;
  wave_sync();
  for (int jb = 0; jb < cmax; jb += 32) {   // wave-uniform
    const int j = jb + hl;
    const bool act = j < cn;
    const float cm = act ? s_mz[j] : 0.0f, ci = act ? s_in[j] : 0.0f;
    const int cc = act ? s_ch[j] : 0;
    // shifts this peak takes part in (cpp:58-75): every s < S for an unannotated peak, else
    // s = 0 and s = its fragment charge
    const uint32_t smask = !act ? 0u : cc == 0 ? (1u << S) - 1u : (1u | (cc < S ? 1u << cc : 0u));
    for (int s = 0; s < Smax; ++s) {        // wave-uniform
      const double md = Wv.mdt[half * 32 + s];
      // bin of the shifted peak in fp32 (the query was filed with a margin that covers the
      // fp32 rounding); bitmap reject first, the exact fp64 window test on the rare hits
      const bool can = (smask >> s) & 1u;
      const int b = (int)floorf((cm + (float)md) * inv_w_f);
      const bool maybe = can && bm_test(H, b);
      if (!__ballot(maybe)) continue;         // wave-uniform
      if (maybe) {
        const double mult = (s == 0 || cc == s) ? 1.0 : 2.0 / 3.0;
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
                const float prod = (float)(mult * (double)Q.inten[i] * (double)ci);   // cpp:81
                const int mslot = atomicAdd(&Wv.pad[half], 1);
                if (mslot < RS_HC) {
                  const uint32_t gen = (uint32_t)((i * S + s) * cn + j);
                  keys[mslot] = ((unsigned long long)__float_as_uint(prod) << 32) |
                                (unsigned long long)(0xFFFFFFFFu - gen);
                  pay[mslot] = ((uint32_t)i << 16) | (uint32_t)j;
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
  const int MA = __builtin_amdgcn_readfirstlane(Wv.pad[0]);
  const int MB = __builtin_amdgcn_readfirstlane(Wv.pad[1]);
  if (MA == 0 && MB == 0) {   // wave-uniform: nothing matched
    scoreA = scoreB = 0.0;
    return;
  }
  int ok = 0;
  if (MA <= 32 && MB <= 32) ok = resolve_two_fast(lane, Wv, MA, MB, scoreA, scoreB);
  if (!(ok & 1))
    scoreA = MA > RS_HC ? RS_DEFER
             : MA     ? resolve_half_slow(lane, Wv, status, 0, MA)
                      : 0.0;
  if (!(ok & 2))
    scoreB = MB > RS_HC ? RS_DEFER
             : MB     ? resolve_half_slow(lane, Wv, status, RS_HC, MB)
                      : 0.0;
}

// FORM: the two shapes the search path calls with -- fixed-stride neighbour lists (1: int32
// rows, 2: int64 rows), packed row records for the precursor filter, annotated library peaks --
// are compiled with those facts folded in (each open "is this pointer null" question is a
// wave-uniform predicate held in scalar registers across the hot loops); 0 = any shape.
// DEF: second-launch mode behind rescore_flat_kernel -- only the slots that kernel marked
// RS_DEFER, only for queries whose q_defer has RS_QD_PAIR; what this kernel cannot resolve
// either goes on to the binary-search kernel (RS_DEFER_BS / RS_QD_BS).
template <int FORM, bool DEF = false>
__global__ __launch_bounds__(64 * RS_WAVES, RS_OCC) void rescore_score_v2_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, double tol, int allow_shift,
    double *__restrict__ pair_score, int *__restrict__ q_defer, int *status) {
  // DEF: a query's deferred slots are few (a handful) or nearly all of its candidates (two
  // query peaks closer than the tolerance make every candidate with a peak there a conflict):
  // as many of the launch's blocks per query take part as there is work for, the others leave
  int ny = (int)gridDim.y;
  if (DEF) {
    const int qd = q_defer[blockIdx.x];
    if (!(qd & RS_QD_PAIR)) return;
    const int want = ((qd >> 2) + 2 * RS_WAVES - 1) / (2 * RS_WAVES);     // one pair step per wave
    ny = want < ny ? want : ny;
    if ((int)blockIdx.y >= ny) return;
  }
  cv.flt.wcol = nullptr;      // this kernel wants filter column and metadata from ONE gather
  if (FORM != 0) {
    cv.offsets = nullptr;
    cv.flt.lib_pmz = nullptr;
    cv.flt.valid = nullptr;
    __builtin_assume(cv.flt.meta != nullptr);
    __builtin_assume(L.charge != nullptr);
    if (FORM == 1) {
      cv.rows64 = nullptr;
    } else {
      __builtin_assume(cv.rows64 != nullptr);
    }
  }
  __shared__ QueryLds2 Q;
  __shared__ HashLds H;
  __shared__ PairLds W[RS_WAVES];
  __shared__ uint16_t s_list[RS_SUPER];
  __shared__ int s_nv, s_defer;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  long long c0, c1;
  cv.range(q, c0, c1);
  if (c0 >= c1) return;
  const int qo = Qs.offsets[q];
  const int qn = Qs.offsets[q + 1] - qo;
  const double q_pmz = Qs.precursor_mz[q];
  // fp32 evaluation of a probe bin (m/z <= ~2600): error bound in bin units; the query
  // peaks are filed with that margin on both sides (still <= 3 bins per peak)
  const double margin = 1e-3 + (tol > 0.0 ? (0.5 / tol) * (3.75e-4 + 2600.0 * 1.2e-7) : 1.0);
  if (!(tol > 0.0) || qn > RS_HQ_MAX || margin > 0.45) {   // uniform: whole query deferred
    for (long long c = c0 + (long long)blockIdx.y * blockDim.x + tid; c < c1;
         c += (long long)blockDim.x * gridDim.y) {
      pair_score[c] = cv.cand(c, q_pmz, L.n) >= 0 ? RS_DEFER_BS : -1.0;
    }
    if (tid == 0) atomicOr(&q_defer[q], RS_QD_BS);
    return;
  }
  for (int i = tid; i < qn; i += blockDim.x) {   // qn <= RS_HQ_MAX here
    Q.mz[i] = Qs.mz[qo + i];
    Q.inten[i] = Qs.intensity[qo + i];
  }
  for (int i = tid; i < RS_HT; i += blockDim.x) H.bin[i] = RS_EMPTY;
  for (int i = tid; i < RS_BM_BITS / 32; i += blockDim.x) H.bm[i] = 0u;
  if (tid == 0) s_defer = 0;
  __syncthreads();
  const double inv_w = 1.0 / (2.0 * tol);
  if (tid < qn) {
    const double qm = (double)Q.mz[tid];
    const int blo = (int)floor((qm - tol) * inv_w - margin);
    const int bhi = (int)floor((qm + tol) * inv_w + margin);
    for (int b = blo; b <= bhi; ++b) {
      const uint32_t bit = hbit(b);
      atomicOr(&H.bm[bit >> 5], 1u << (bit & 31));
      uint32_t h = hbin(b);
      for (;;) {
        const int old = atomicCAS(&H.bin[h], RS_EMPTY, b);
        if (old == RS_EMPTY) {
          H.peak[h] = (uint8_t)tid;
          break;
        }
        h = (h + 1) & (RS_HT - 1);
      }
    }
  }
  __syncthreads();
  PairLds &Wv = W[wave];
  const int parts = RS_WAVES, part = wave;

  for (long long sb = c0; sb < c1; sb += RS_SUPER) {
    // ---- compact the valid slots of this super-chunk (order is irrelevant: scores are
    // written back to their slots)
    const int sn = (int)((c1 - sb) < RS_SUPER ? (c1 - sb) : RS_SUPER);
    if (tid == 0) s_nv = 0;
    __syncthreads();
    for (int i0 = 0; i0 < sn; i0 += blockDim.x) {
      const int i = i0 + tid;
      bool ok = false;
      if (i < sn) {
        if (DEF) {
          ok = pair_score[sb + i] == RS_DEFER;     // a valid candidate: the first launch checked
        } else {
          ok = cv.cand(sb + i, q_pmz, L.n) >= 0;
          if (!ok && blockIdx.y == 0) pair_score[sb + i] = -1.0;
        }
        // several blocks per query: each takes the slots of every ny-th group of 32 (the order
        // inside s_list depends on the waves' timing, so blocks must not split it by position)
        ok = ok && (ny == 1 || ((i >> 5) % ny) == (int)blockIdx.y);
      }
      const unsigned long long bal = __ballot(ok);
      int wbase = 0;
      if (lane == 0 && bal) wbase = atomicAdd(&s_nv, __popcll(bal));
      wbase = rl_i(wbase, 0);
      if (ok) s_list[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    const int nv = s_nv;
    const int per = (nv + parts - 1) / parts;
    const int wb = part * per;
    const int we = wb + per < nv ? wb + per : nv;

    for (int base = wb; base < we; base += 64) {
      const bool okr = base + lane < we;
      const int slot = okr ? (int)s_list[base + lane] : 0;   // candidate slot = sb + slot
      int m_co = 0, m_cn = 0, m_chg = 0;
      double m_pmz = 0.0;
      if (okr) {
        const long long row = cv.row(sb + slot);
        if (cv.flt.meta) {     // one 32-byte sector per candidate
          const RowMeta *mr = meta_row(cv.flt, row);
          const uint4 a = *reinterpret_cast<const uint4 *>(mr);
          m_co = (int)a.x;
          m_cn = (int)a.y;
          m_chg = (int)a.z;
          m_pmz = mr->pmz64;
        } else {
          m_co = L.offsets[row];
          m_cn = L.offsets[row + 1] - m_co;
          m_chg = L.precursor_charge[row];
          m_pmz = L.precursor_mz[row];
        }
      }
      double my_score = 0.0;
      const int cnt = we - base < 64 ? we - base : 64;
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
        for (int u = 0; u < RS_PF; u += 2) {
          const int lA = g0 + u, lB = g0 + u + 1;
          const bool vA = lA < cnt, vB = lB < cnt;   // wave-uniform
          const int cnA = vA ? rl_i(m_cn, lA) : 0, cnB = vB ? rl_i(m_cn, lB) : 0;
          const int chA = vA ? rl_i(m_chg, lA) : 0, chB = vB ? rl_i(m_chg, lB) : 0;
          const bool defA = cnA > 64 || chA >= 31, defB = cnB > 64 || chB >= 31;
          const bool runA = vA && !defA && cnA > 0 && qn > 0;
          const bool runB = vB && !defB && cnB > 0 && qn > 0;
          double sA = 0.0, sB = 0.0;
          if (runA || runB) {
            score_two(lane, Q, H, Wv, u, u + 1, runA ? cnA : 0, runB ? cnB : 0, chA, chB,
                      rl_d(m_pmz, vA ? lA : 0), rl_d(m_pmz, vB ? lB : 0), q_pmz, tol, inv_w,
                      allow_shift, status, sA, sB);
          }
          if (defA || sA == RS_DEFER) sA = RS_DEFER_BS;
          if (defB || sB == RS_DEFER) sB = RS_DEFER_BS;
          if (vA && lane == lA) my_score = sA;
          if (vB && lane == lB) my_score = sB;
        }
        wave_sync();
      }
      if (okr) pair_score[sb + slot] = my_score;
      if (__ballot(okr && my_score == RS_DEFER_BS) && lane == 0) s_defer = 1;
    }
    __syncthreads();   // s_list is rebuilt for the next super-chunk
  }
  if (tid == 0 && s_defer) atomicOr(&q_defer[q], RS_QD_BS);
}

// ---------------------------------------------------------------------------------
// Pass 1, FLAT formulation (the first launch of the search path). The pair kernel above spends
// two thirds of its instructions outside the probing loop: per pair of candidates it stages
// peaks through LDS, sets up shifts and tables with every lane, and resolves two match lists
// with DPP trees. Here a wave takes RF_NC candidates at a time and
//   * SETS THEM UP ONE LANE PER CANDIDATE (row record, shift count, mass-difference table,
//     zeroed accumulators, a prefix sum of the peak counts),
//   * walks their peaks as ONE STREAM, one candidate peak per lane whatever candidate it
//     belongs to (an owner table maps stream positions to candidates; peaks are loaded straight
//     from HBM a step ahead, no LDS staging), probing the query's bin bitmap / hash per shift
//     exactly like the pair kernel,
//   * and ACCUMULATES a match where it is found: atomicOr into the candidate's masks of matched
//     query / candidate peaks (a bit that was already set = a doubly matched peak), atomic
//     min / max of the product's exponent, and an fp64 atomic add of the product.
// If no peak was matched twice the reference's greedy pass accepts every match, and if the
// products span <= 23 binades the fp64 sum of <= 64 fp32 products is exact in ANY order
// (24 + 23 + 6 = 53 bits): the accumulated sum IS the reference's sorted sum, bit for bit.
// Anything else -- a doubly matched peak, a wider span, more shifts than the table holds -- is
// marked RS_DEFER for the pair kernel (second launch, ~2 % of the candidates); what that
// kernel does not take either (> 64 peaks, > 64 matches, query > 100 peaks, tol <= 0) goes to
// the binary-search kernel as RS_DEFER_BS.
constexpr int RF_NC = 32;        // candidates per wave chunk (set-up lanes)
constexpr int RF_PMAX = 1024;    // candidate peaks per wave chunk (owner table)
constexpr int RF_SMAX = 5;       // shifts the mass-difference table holds (precursor charge <= 4)
constexpr int RF_MQ = 128;       // (peak, shift) items queued per wave before a drain

struct FlatLds {   // per wave
  double sum[RF_NC];
  double pmd[RF_NC];
  float mdf[RF_NC][RF_SMAX - 1];   // (float)(pmd / s), s = 1..4: the probe bins
  uint32_t qmask[RF_NC][4];        // matched query peaks (<= 128 on this path)
  uint32_t cmask[RF_NC][2];        // matched candidate peaks (<= 64)
  uint32_t emax[RF_NC];            // largest exponent byte of a product | bit 31: doubly matched peak
  uint32_t emin[RF_NC];
  int base[RF_NC];                 // first peak in the library arrays
  uint32_t rec[RF_NC];             // peaks | shifts << 8 | precursor charge << 16
  uint16_t pref[RF_NC];            // first stream position
  uint8_t owner[RF_PMAX];
  uint16_t mq[RF_MQ];              // queued (peak | shift << 10) items whose bin is marked
};

constexpr int RF_OCC = 6;      // waves per SIMD: 80 VGPRs (7 / 8: 72 / 64 VGPRs and more spills: +2 % / +12 %, profiles/r03_rescore_ab.txt)
template <int FORM>
__global__ __launch_bounds__(64 * RS_WAVES, RF_OCC) void rescore_flat_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, double tol, int allow_shift,
    double *__restrict__ pair_score, int *__restrict__ q_defer, int *status) {
  if (FORM != 0) {
    cv.offsets = nullptr;
    cv.flt.lib_pmz = nullptr;
    cv.flt.valid = nullptr;
    __builtin_assume(cv.flt.meta != nullptr);
    __builtin_assume(cv.flt.wcol != nullptr);
    __builtin_assume(L.charge != nullptr);
    __builtin_assume(L.records != nullptr);
    if (FORM == 1) {
      cv.rows64 = nullptr;
    } else {
      __builtin_assume(cv.rows64 != nullptr);
    }
  }
  __shared__ QueryLds2 Q;
  __shared__ HashLds H;
  __shared__ FlatLds W[RS_WAVES];
  __shared__ uint16_t s_list[RS_SUPER];
  __shared__ int s_nv, s_defer, s_ndef;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  long long c0, c1;
  cv.range(q, c0, c1);
  if (c0 >= c1) return;
  const int qo = Qs.offsets[q];
  const int qn = Qs.offsets[q + 1] - qo;
  const double q_pmz = Qs.precursor_mz[q];
  // the row was filtered where it was made (block-uniform): every slot in range is a candidate
  const bool pre = cv.prefiltered(q);
  auto is_cand = [&](long long c) -> bool {
    if (pre) {
      const long long r = cv.row(c);
      return r >= 0 && r < L.n;
    }
    return cv.cand(c, q_pmz, L.n) >= 0;
  };
  // (same bin filter as the pair kernel: see there)
  const double margin = 1e-3 + (tol > 0.0 ? (0.5 / tol) * (3.75e-4 + 2600.0 * 1.2e-7) : 1.0);
  if (!(tol > 0.0) || qn > RS_HQ_MAX || margin > 0.45) {   // uniform: whole query deferred
    for (long long c = c0 + (long long)blockIdx.y * blockDim.x + tid; c < c1;
         c += (long long)blockDim.x * gridDim.y) {
      pair_score[c] = is_cand(c) ? RS_DEFER_BS : -1.0;
    }
    if (tid == 0) atomicOr(&q_defer[q], RS_QD_BS);
    return;
  }
  for (int i = tid; i < qn; i += blockDim.x) {
    Q.mz[i] = Qs.mz[qo + i];
    Q.inten[i] = Qs.intensity[qo + i];
  }
  for (int i = tid; i < RS_HT; i += blockDim.x) H.bin[i] = RS_EMPTY;
  for (int i = tid; i < RS_BM_BITS / 32; i += blockDim.x) H.bm[i] = 0u;
  if (tid == 0) s_defer = s_ndef = 0;
  __syncthreads();
  const double inv_w = 1.0 / (2.0 * tol);
  if (tid < qn) {
    const double qm = (double)Q.mz[tid];
    const int blo = (int)floor((qm - tol) * inv_w - margin);
    const int bhi = (int)floor((qm + tol) * inv_w + margin);
    for (int b = blo; b <= bhi; ++b) {
      const uint32_t bit = hbit(b);
      atomicOr(&H.bm[bit >> 5], 1u << (bit & 31));
      uint32_t h = hbin(b);
      for (;;) {
        const int old = atomicCAS(&H.bin[h], RS_EMPTY, b);
        if (old == RS_EMPTY) {
          H.peak[h] = (uint8_t)tid;
          break;
        }
        h = (h + 1) & (RS_HT - 1);
      }
    }
  }
  __syncthreads();
  FlatLds &Wv = W[wave];
  const float inv_w_f = (float)inv_w;
  const int parts = RS_WAVES, part = wave, ny = (int)gridDim.y;
  const int t = lane & 31, hi = lane >> 5;

  for (long long sb = c0; sb < c1; sb += RS_SUPER) {
    // ---- compact the valid slots of this super-chunk
    const int sn = (int)((c1 - sb) < RS_SUPER ? (c1 - sb) : RS_SUPER);
    if (tid == 0) s_nv = 0;
    __syncthreads();
    for (int i0 = 0; i0 < sn; i0 += blockDim.x) {
      const int i = i0 + tid;
      bool ok = false;
      if (i < sn) {
        ok = is_cand(sb + i);
        if (!ok && blockIdx.y == 0) pair_score[sb + i] = -1.0;
        ok = ok && (ny == 1 || ((i >> 5) % ny) == (int)blockIdx.y);   // (see the pair kernel)
      }
      const unsigned long long bal = __ballot(ok);
      int wbase = 0;
      if (lane == 0 && bal) wbase = atomicAdd(&s_nv, __popcll(bal));
      wbase = rl_i(wbase, 0);
      if (ok) s_list[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    const int nv = s_nv;
    const int per = (nv + parts - 1) / parts;
    const int wb = part * per;
    const int we = wb + per < nv ? wb + per : nv;

    for (int base = wb; base < we;) {
      // ---- set-up: lanes t and t + 32 both hold candidate t (they share the owner fill)
      const bool okr = base + t < we;
      const int slot = okr ? (int)s_list[base + t] : 0;
      int m_co = 0, m_cn = 0, m_chg = 0;
      double m_pmz = 0.0;
      if (okr) {
        const long long row = cv.row(sb + slot);
        if (cv.flt.meta) {     // one 32-byte sector per candidate
          const RowMeta *mr = meta_row(cv.flt, row);
          const uint4 a = *reinterpret_cast<const uint4 *>(mr);
          m_co = (int)a.x;
          m_cn = (int)a.y;
          m_chg = (int)a.z;
          m_pmz = mr->pmz64;
          if (L.records) m_co = (int)mr->rec4;     // base of the peak record, 4-byte units
        } else {
          m_co = L.offsets[row];
          m_cn = L.offsets[row + 1] - m_co;
          m_chg = L.precursor_charge[row];
          m_pmz = L.precursor_mz[row];
        }
      }
      const double pmd = (q_pmz - m_pmz) * (double)(unsigned)m_chg;            // cpp:18
      const int S = (allow_shift && fabs(pmd) >= tol) ? m_chg + 1 : 1;          // cpp:20
      const bool def_bs = okr && (m_cn > 64 || m_chg >= 31);
      const bool def_pair = okr && !def_bs && S > RF_SMAX;
      const int cn_eff = (okr && !def_bs && !def_pair && qn > 0) ? m_cn : 0;
      // prefix sums over the 32 candidates: both halves of the wave hold the same values, so the
      // 64-lane DPP scan (common.hpp) is the 32-lane one plus, in the upper half, the total
      int incl = (int)wave_incl_scan((uint32_t)cn_eff);
      incl -= hi ? rl_i(incl, 31) : 0;
      // the chunk ends where the owner table is full (the prefix sums ascend)
      const uint32_t fit = (uint32_t)__ballot(okr && incl <= RF_PMAX);
      const int ntake = __popc(fit);          // >= 1: a candidate has <= 64 peaks here
      const bool take = t < ntake;
      const int excl = incl - cn_eff;
      const int P = rl_i(incl, ntake - 1);
      int cmaxw = take ? cn_eff : 0, Smaxw = (take && cn_eff > 0) ? S : 0;
      cmaxw = rl_i(wave_max_to_lane63(cmaxw), 63);     // (both >= 0)
      Smaxw = rl_i(wave_max_to_lane63(Smaxw), 63);
      if (take && hi == 0) {
        Wv.base[t] = m_co;
        Wv.rec[t] = (uint32_t)cn_eff | ((uint32_t)S << 8) | ((uint32_t)m_chg << 16);
        Wv.pref[t] = (uint16_t)excl;
        Wv.pmd[t] = pmd;
        Wv.sum[t] = 0.0;
        Wv.qmask[t][0] = Wv.qmask[t][1] = Wv.qmask[t][2] = Wv.qmask[t][3] = 0u;
        Wv.cmask[t][0] = Wv.cmask[t][1] = 0u;
        Wv.emax[t] = 0u;
        Wv.emin[t] = 255u;
        // mass_diff[s] = pmd / s (cpp:26-31); x / 1, x / 2, x / 4 are exact scalings
        Wv.mdf[t][0] = (float)pmd;
        Wv.mdf[t][1] = (float)(pmd * 0.5);
        Wv.mdf[t][3] = (float)(pmd * 0.25);
      }
      if (Smaxw > 3) {          // wave-uniform: the fp64 division only where a charge-3 candidate is
        if (take && hi == 0) Wv.mdf[t][2] = (float)(pmd / 3.0);
      }
      for (int j0 = 0; j0 < cmaxw; j0 += 2) {
        const int j = j0 + hi;
        if (take && j < cn_eff) Wv.owner[excl + j] = (uint8_t)t;
      }
      wave_sync();

      // ---- the chunk's peaks as one stream, loads one step ahead. The stream needs a peak's m/z
      // and fragment charge only; intensities are read by the drain, for the peaks that hit
      float n_cm = 0.0f;
      int n_cc = 0, n_tt = 0;
      uint32_t n_rec = 0;
      auto fetch = [&](int p0) {
        const int p = p0 + lane;
        const bool act = p < P;
        n_tt = act ? (int)Wv.owner[p] : 0;
        n_rec = act ? Wv.rec[n_tt] : 0u;
        const int n_j = p - (int)Wv.pref[n_tt];
        const int n_co = Wv.base[n_tt];
        if (L.records && cv.flt.meta) {      // [mz x cn][charge x cn][intensity x cn], one record
          const float *rf = reinterpret_cast<const float *>(L.records) + (uint32_t)n_co;
          const int cnr = (int)(n_rec & 0xffu);
          n_cm = act ? rf[n_j] : 0.0f;
          n_cc = act ? (int)reinterpret_cast<const uint8_t *>(rf)[4 * cnr + n_j] : 0;
        } else {
          n_cm = act ? L.mz[n_co + n_j] : 0.0f;
          n_cc = (act && L.charge) ? (int)L.charge[n_co + n_j] : 0;
        }
      };
      // ---- queued (peak, shift) items, a row at a time: everything about the item is looked up
      // again (the peak's values come from the lines the stream has just read)
      int mq_n = 0;
      auto drain = [&]() {
        wave_sync();
        for (int e0 = 0; e0 < mq_n; e0 += 64) {
          const bool on = e0 + lane < mq_n;
          const uint32_t ent = on ? (uint32_t)Wv.mq[e0 + lane] : 0u;
          const int s = (int)(ent >> 10), p = (int)(ent & 1023u);
          const int tt = (int)Wv.owner[p];
          const int j = p - (int)Wv.pref[tt], co = Wv.base[tt];
          const int cn = (int)(Wv.rec[tt] & 0xffu);
          float cm = 0.0f, ci = 0.0f;
          int cc = 0;
          if (on) {
            if (L.records && cv.flt.meta) {
              const float *rf = reinterpret_cast<const float *>(L.records) + (uint32_t)co;
              cm = rf[j];
              ci = rf[rec_int0(cn) + j];
              cc = (int)reinterpret_cast<const uint8_t *>(rf)[4 * cn + j];
            } else {
              cm = L.mz[co + j];
              ci = L.intensity[co + j];
              cc = L.charge ? (int)L.charge[co + j] : 0;
            }
          }
          if (on) {
            const float mdv = s ? Wv.mdf[tt][s - 1] : 0.0f;
            const int b = (int)floorf((cm + mdv) * inv_w_f);
            const double pm = Wv.pmd[tt];
            const double md = s == 0 ? 0.0 : s == 1 ? pm : s == 2 ? pm * 0.5 : s == 4 ? pm * 0.25 : pm / (double)s;
            const double mult = (s == 0 || cc == s) ? 1.0 : 2.0 / 3.0;
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
                    const float *pm_ = (L.records && cv.flt.meta)
                                           ? reinterpret_cast<const float *>(L.records) + (uint32_t)co
                                           : L.mz + co;
                    const double xp = (double)pm_[jj - 1] + md;
                    if (lim > xp) break;
                    if (!(fabs(qm - xp) <= tol)) {
                      run = false;
                      break;
                    }
                  }
                  if (run) {
                    const float prod = (float)(mult * (double)Q.inten[i] * (double)ci);   // cpp:81
                    const uint32_t e = max((__float_as_uint(prod) >> 23) & 0xffu, 1u);
                    const uint32_t qb = 1u << (i & 31), cb = 1u << (j & 31);
                    const uint32_t oq = atomicOr(&Wv.qmask[tt][i >> 5], qb);
                    const uint32_t oc = atomicOr(&Wv.cmask[tt][j >> 5], cb);
                    atomicMax(&Wv.emax[tt], ((oq & qb) || (oc & cb)) ? (0x80000000u | e) : e);
                    atomicMin(&Wv.emin[tt], e);
                    atomicAdd(&Wv.sum[tt], (double)prod);
                  }
                }
              }
              h = (h + 1) & (RS_HT - 1);
            }
          }
        }
        mq_n = 0;
        wave_sync();
      };
      if (P > 0) fetch(0);
      for (int p0 = 0; p0 < P; p0 += 64) {
        const float cm = n_cm;
        const int cc = n_cc, tt = n_tt;
        const int Sc = (int)((n_rec >> 8) & 0xffu);
        const bool act = p0 + lane < P;
        if (p0 + 64 < P) fetch(p0 + 64);
        // shifts this peak takes part in (cpp:58-75): every s < S for an unannotated peak, else
        // s = 0 and s = its fragment charge. A (peak, shift) whose bin is marked goes to the
        // wave's queue; the probing proper runs over full rows of queued items (drain) instead
        // of inside this loop with the one or two lanes that hit.
        const uint32_t smask = !act ? 0u : cc == 0 ? (1u << Sc) - 1u : (1u | (cc < Sc ? 1u << cc : 0u));
        auto probe = [&](int s, float mdv) {
          const bool can = (smask >> s) & 1u;
          const int b = (int)floorf((cm + mdv) * inv_w_f);
          const bool maybe = can && bm_test(H, b);
          const unsigned long long mm = __ballot(maybe);
          if (!mm) return;                        // wave-uniform
          const int c = __popcll(mm);
          if (mq_n + c > RF_MQ) drain();
          if (maybe)
            Wv.mq[mq_n + __popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)((p0 + lane) | (s << 10));
          mq_n += c;
        };
        probe(0, 0.0f);
        for (int s = 1; s < Smaxw; ++s) probe(s, Wv.mdf[tt][s - 1]);        // wave-uniform bound
      }
      drain();
      wave_sync();
      // ---- scores of the chunk
      if (take && hi == 0) {
        const uint32_t em = Wv.emax[t], en = Wv.emin[t];
        double sc = Wv.sum[t];
        if ((em >> 31) || ((em & 0xffu) > en + 23u)) sc = RS_DEFER;     // sort + greedy pass needed
        if (def_pair) sc = RS_DEFER;
        if (def_bs) sc = RS_DEFER_BS;
        pair_score[sb + slot] = sc;
        if (sc == RS_DEFER) {
          atomicOr(&s_defer, RS_QD_PAIR);
          atomicAdd(&s_ndef, 1);
        }
        if (sc == RS_DEFER_BS) atomicOr(&s_defer, RS_QD_BS);
      }
      base += ntake;
      wave_sync();     // the chunk's tables are rebuilt
    }
    __syncthreads();   // s_list is rebuilt for the next super-chunk
  }
  if (tid == 0 && s_defer) {      // (several blocks per query when gridDim.y > 1: or + add)
    atomicOr(&q_defer[q], s_defer);
    if (s_ndef) atomicAdd(&q_defer[q], s_ndef << 2);
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

// One wave per query, and a wave is a chain of dependent memory round trips (winner's slot -> its
// row -> its peaks) around little arithmetic: the kernel lives on the number of waves in flight,
// i.e. on the LDS a wave needs. SMALL: structures for spectra of <= 128 peaks and <= 128
// generated matches (4 KB per wave instead of 11: 32 waves per CU instead of 12); a query beyond
// them is marked in m_defer and done by the full-size instantiation, which runs second and only
// looks at marked queries.
constexpr int RS_SMALL_P = 128, RS_SMALL_M = 128;
template <bool SMALL, class QL, class WL>
__device__ __forceinline__ void matches_one(int q, int lane, QL &Qw, WL &Ww, int *s_cnt_w, const DevPeaks &Qs,
                                            const DevPeaks &L, const CandView &cv,
                                            const long long *__restrict__ best_slot, double tol, int allow_shift,
                                            int32_t *__restrict__ pm_count, uint32_t *__restrict__ pm_pairs,
                                            int pm_stride, int32_t *__restrict__ best_row, int *status,
                                            int *__restrict__ m_defer) {
  const long long slot = best_slot[q];
  const long long row = slot >= 0 ? cv.row(slot) : -1;
  if (best_row && lane == 0) best_row[q] = (int32_t)row;
  if (row < 0) {
    if (pm_count && lane == 0) pm_count[q] = 0;
    if (pm_pairs)   // rows are fully defined: zero beyond the matches (callers may pass raw memory)
      for (int t = lane; t < 2 * pm_stride; t += 64) pm_pairs[(size_t)q * pm_stride * 2 + t] = 0u;
    return;
  }
  if (!pm_count && !pm_pairs) return;
  if (SMALL && Qs.offsets[q + 1] - Qs.offsets[q] > RS_SMALL_P) {
    if (lane == 0) m_defer[q] = 1;
    return;
  }
  int qn;
  load_query(lane, 64, Qs, q, Qw, qn, status);
  wave_sync();
  int cnt_tmp = 0;
  const double sc = dot_pair_wave<true, SMALL>(lane, Qw, qn, Qs.precursor_mz[q], L, (int)row, tol, allow_shift,
                                               Ww, pm_pairs ? pm_pairs + (size_t)q * pm_stride * 2 : nullptr,
                                               pm_pairs ? pm_stride : 0, s_cnt_w, status);
  if (SMALL && sc < 0.0) {        // (wave-uniform) does not fit: the second launch does this query
    if (lane == 0) m_defer[q] = 1;
    return;
  }
  wave_sync();
  cnt_tmp = *s_cnt_w;
  if (pm_count && lane == 0) pm_count[q] = cnt_tmp;
  if (pm_pairs)
    for (int t = 2 * (cnt_tmp < pm_stride ? cnt_tmp : pm_stride) + lane; t < 2 * pm_stride; t += 64)
      pm_pairs[(size_t)q * pm_stride * 2 + t] = 0u;
}

// SMALL: a wave per query. Full size: a wave per RS_MF_GROUP queries, which reads their flags at once
// and does the marked ones (normally none: a 44 KB workgroup per four queries -- one per 64 now --, placed only to
// return, took 0.04 ms of a step).
constexpr int RS_MF_GROUP = 16;
template <bool SMALL>
__global__ __launch_bounds__(64 * RS_WAVES) void rescore_matches_kernel(
    DevPeaks Qs, DevPeaks L, CandView cv, int nq, const long long *__restrict__ best_slot,
    double tol, int allow_shift, int32_t *__restrict__ pm_count,
    uint32_t *__restrict__ pm_pairs, int pm_stride, int32_t *__restrict__ best_row,
    int *status, int *__restrict__ m_defer) {
  typedef QueryLdsT<SMALL ? RS_SMALL_P : RS_MAXP> QL;
  typedef WaveLdsT<SMALL ? RS_SMALL_P : RS_MAXP, SMALL ? RS_SMALL_M : RS_MCAP> WL;
  __shared__ QL Q[RS_WAVES];
  __shared__ WL W[RS_WAVES];
  __shared__ int s_cnt[RS_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (SMALL) {
    const int q = blockIdx.x * RS_WAVES + wave;
    if (q >= nq) return;
    if (lane == 0) m_defer[q] = 0;
    matches_one<true>(q, lane, Q[wave], W[wave], &s_cnt[wave], Qs, L, cv, best_slot, tol, allow_shift, pm_count,
                      pm_pairs, pm_stride, best_row, status, m_defer);
    return;
  }
  // a wave takes RS_MF_GROUP consecutive queries: one read of their flags; if every query is marked
  // (spectra of more than 128 peaks) the grid still holds nq / 16 workgroups
  const int base = (blockIdx.x * RS_WAVES + wave) * RS_MF_GROUP;
  unsigned long long todo = __ballot(lane < RS_MF_GROUP && base + lane < nq && m_defer[base + lane] != 0);
  while (todo) {          // wave-uniform
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    matches_one<false>(base + l, lane, Q[wave], W[wave], &s_cnt[wave], Qs, L, cv, best_slot, tol, allow_shift,
                       pm_count, pm_pairs, pm_stride, best_row, status, m_defer);
    wave_sync();
  }
}

// Host driver shared by asl_rescore_batch and asl_search_batch. All pointers are
// device pointers. pair_score scratch must hold one double per candidate slot.
int rescore_device(const DevPeaks &Q, const DevPeaks &L, const int64_t *rows64,
                   const int32_t *rows32, const int32_t *cand_offsets, int32_t stride,
                   int64_t total_slots, double tol, int allow_shift, int tie_by_row,
                   double *pair_score, long long *best_slot, int32_t *best_cand,
                   int32_t *best_row, double *best_score, int32_t *n_valid,
                   int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride, int *status,
                   const PrecFilter &filter, bool clear_status, RescoreScratch *scratch,
                   const int32_t *row_counts) {
  const int nq = Q.n;
  if (nq == 0) return ASL_OK;
  if (!scratch) return fail(ASL_ERR_INVALID, "rescore: no scratch (internal)");
  DevBuf<int> &q_defer = scratch->q_defer, &m_defer = scratch->m_defer;
  CandView cv{rows64, rows32, cand_offsets, stride, filter};
  cv.counts = cand_offsets ? nullptr : row_counts;
  if (clear_status) HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), stream()));
  {
    ProfScope ps("rescore");
    // split long candidate lists over blockIdx.y when there are few queries
    int64_t avg = total_slots / (nq > 0 ? nq : 1);
    int ysplit = 1;
    if (nq < 2048 && avg > 4096) ysplit = (int)std::min<int64_t>(64, cdiv(avg, 4096));
    {
      // hash kernel, then the binary-search kernel on whatever it deferred (its blocks
      // return at once for queries with nothing deferred)
      ASL_TRY(q_defer.reserve((size_t)nq));
      HIP_TRY(hipMemsetAsync(q_defer.p, 0, sizeof(int) * (size_t)nq, stream()));
      const bool shaped = !cand_offsets && filter.meta && filter.wcol && L.charge && L.records &&
                          (rows64 || rows32);
      // 1. flat kernel; 2. pair kernel on what it marked RS_DEFER; 3. binary-search kernel on
      // RS_DEFER_BS (blocks of 2 / 3 return at once for queries without such slots)
      auto flat = !shaped ? rescore_flat_kernel<0> : rows64 ? rescore_flat_kernel<2> : rescore_flat_kernel<1>;
      hipLaunchKernelGGL(flat, dim3(nq, ysplit), dim3(64 * RS_WAVES), 0, stream(), Q, L, cv, tol,
                         allow_shift, pair_score, q_defer.p, status);
      ASL_CHECK_LAUNCH();
      auto kern = !shaped ? rescore_score_v2_kernel<0, true>
                  : rows64 ? rescore_score_v2_kernel<2, true>
                           : rescore_score_v2_kernel<1, true>;
      hipLaunchKernelGGL(kern, dim3(nq, std::max(ysplit, RS_DEF_Y)), dim3(64 * RS_WAVES), 0, stream(), Q,
                         L, cv, tol, allow_shift, pair_score, q_defer.p, status);
      ASL_CHECK_LAUNCH();
      // (small batches keep a workgroup per query: when every query is marked -- tol <= 0, queries
      // of more than 100 peaks -- they need all the parallelism there is)
      const int bs_group = nq >= 4096 ? RS_BS_GROUP : 1;
      hipLaunchKernelGGL(rescore_score_kernel, dim3((unsigned)cdiv(nq, bs_group), ysplit), dim3(64 * RS_WAVES), 0,
                         stream(), Q, L, cv, tol, allow_shift, pair_score,
                         (const int *)q_defer.p, status, bs_group);
    }
    ASL_CHECK_LAUNCH();
    hipLaunchKernelGGL(rescore_argmax_kernel, dim3(nq), dim3(64), 0, stream(), cv, nq,
                       pair_score, tie_by_row, best_cand, best_slot, best_score, n_valid);
    ASL_CHECK_LAUNCH();
  }
  {
    ProfScope ps("rescore_matches");
    ASL_TRY(m_defer.reserve((size_t)nq));
    hipLaunchKernelGGL(rescore_matches_kernel<true>, dim3((unsigned)cdiv(nq, RS_WAVES)),
                       dim3(64 * RS_WAVES), 0, stream(), Q, L, cv, nq, best_slot, tol,
                       allow_shift, pm_count, pm_pairs, pm_stride, best_row, status, m_defer.p);
    ASL_CHECK_LAUNCH();
    hipLaunchKernelGGL(rescore_matches_kernel<false>, dim3((unsigned)cdiv(nq, RS_MF_GROUP * RS_WAVES)),
                       dim3(64 * RS_WAVES), 0, stream(), Q, L, cv, nq, best_slot, tol,
                       allow_shift, pm_count, pm_pairs, pm_stride, best_row, status, m_defer.p);
    ASL_CHECK_LAUNCH();
  }
  return ASL_OK;
}

int rescore_status_error(int st) {
  if (st & RS_STATUS_PEAKS)
    return fail(ASL_ERR_CAPACITY, "rescore: a spectrum has more than %d peaks", RS_MAXP);
  if (st & RS_STATUS_MATCHES)
    return fail(ASL_ERR_CAPACITY, "rescore: a pair generated more than %d peak matches", RS_MCAP);
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
  RescoreScratch scratch;      // lives until rescore_check_status below has synchronised
  ASL_TRY(rescore_device(Q.dev, L.dev, rows.d, nullptr, off.d, 0, total, tol, allow_shift, 0,
                         pair_score.p, best_slot.p, o_best.d, nullptr, o_score.d, nullptr,
                         o_cnt.d, o_pairs.d, pm_stride, status.p, PrecFilter(), true, &scratch));
  ASL_TRY(o_best.finish());
  ASL_TRY(o_score.finish());
  ASL_TRY(o_cnt.finish());
  ASL_TRY(o_pairs.finish());
  return rescore_check_status(status.p);  // synchronises
}
