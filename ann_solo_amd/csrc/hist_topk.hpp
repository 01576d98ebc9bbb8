// hist_topk.hpp -- exact top-k of a long candidate stream for one workgroup of NT = 256 or
// 512 threads
// WITHOUT sorting while streaming (used by pq_scan_v3.hip and flat_scan.hip).
//
//   * hist[512]: counts of the appended candidates per score bucket (monotone linear
//     bucketing of the fp32 score over [-0.25, 1)),
//   * bstar: the highest bucket with at least k appended candidates at or above it.
// A candidate whose bucket is below bstar can never be among the k best (k candidates with
// strictly larger scores exist): it is not stored; when the key buffer runs full it is
// compacted by the same test (one prefix scan, no sort). Keys carry a storage SLOT; ids
// are resolved and the survivors sorted exactly once, in finish(), under
// (score desc, id asc). If compaction cannot free the buffer (thousands of bit-identical
// scores) the workgroup switches to StreamTopK's exact sort-and-truncate flushes.
//
// Calling discipline (all 256 threads, uniform control flow):
//   init();  per round: { b = begin_round(); ... offer() per candidate ...; end_round(n_appended_by_my_wave); }
//   finish(D, I64, I32);
#pragma once
#include "topk.hpp"

namespace asl {

constexpr int HT_NT = 256;
constexpr int HT_NB = 512;
// bucket range [-0.25, 1): hashed spectra are non-negative unit vectors, so inner products (and
// their PQ approximations, up to the quantisation error) live in [0, 1]; scores outside clamp to
// the end buckets, which only costs resolution there, never exactness
constexpr float HT_LO = -0.25f, HT_SCALE = HT_NB / 1.25f;

__device__ __forceinline__ int score_bucket(float s) {
  float t = (s - HT_LO) * HT_SCALE;
  t = fminf(fmaxf(t, 0.0f), (float)(HT_NB - 1));
  return (int)t;
}

// smallest score whose bucket is >= b (1 <= b < HT_NB): score_bucket is monotone, so
// score_bucket(s) >= b  <=>  s >= bucket_floor(b) for every non-NaN s -- one compare instead of
// the bucket arithmetic where only the verdict is needed
__device__ __forceinline__ float bucket_floor(int b) {
  float s = (float)b / HT_SCALE + HT_LO;
  while (score_bucket(s) >= b) s = nextafterf(s, -INFINITY);
  while (score_bucket(s) < b) s = nextafterf(s, INFINITY);
  return s;
}

// exclusive prefix sum of one int per thread over a workgroup of NW waves; `part` = NW LDS
// ints reserved for this call site. Total in `total`.
// (wave_incl_scan: common.hpp)
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int *part, int tid, int &total) {
  const int lane = tid & 63, wave = tid >> 6;
  const int incl = (int)wave_incl_scan((uint32_t)v);
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int t = part[w];
    if (w < wave) base += t;
    total += t;
  }
  return base + incl - v;
}
__device__ __forceinline__ int block_excl_scan256(int v, int *part, int tid, int &total) {
  return block_excl_scan<4>(v, part, tid, total);
}

// LDS footprint: CAP*8 (keys) + 256 (control) + HT_NB*4 (histogram)
template <int CAP, int ROUND_VECS, int NT = HT_NT>
struct HistTopK {
  static constexpr int NW = NT / 64;
  enum { C_FILL = 0, C_BSTAR = 1, C_WCNT = 2 /* 2*NW <= 16 */, C_PART_B = 18 /* NW <= 8 */,
         C_PART_C = 26 /* NW <= 8 */, C_USER = 34 /* 4 ints for the caller */ };
  static constexpr int PER = CAP / NT;
  u64 *keys;
  u64 *thr_p;
  int *ctl;
  int *hist;
  const int32_t *slot_ids;
  StreamTopK<NT, CAP> tk;
  int k, tid, lane, wave;
  int fill, parity, round_no;
  bool sort_mode;
  bool out_keys = false;   // rows of packed (score, ~id) keys in I64 instead of (D, ids)
  // per-round snapshot
  int bstar;
  uint32_t thr_hi;
  float thr_f;             // bucket_floor(bstar), -inf while bstar == 0

  static constexpr size_t lds_bytes() { return (size_t)CAP * 8 + 256 + (size_t)HT_NB * 4; }

  // `base` must be 16-byte aligned; returns the first byte after the structure's LDS
  __device__ __forceinline__ char *init(char *base, int k_, const int32_t *slot_ids_, int tid_) {
    keys = reinterpret_cast<u64 *>(base);
    thr_p = keys + CAP;
    ctl = reinterpret_cast<int *>(thr_p + 1);
    hist = reinterpret_cast<int *>(thr_p + 32);
    slot_ids = slot_ids_;
    k = k_;
    tid = tid_;
    lane = tid & 63;
    wave = tid >> 6;
    fill = parity = round_no = 0;
    sort_mode = false;
    tk.init(keys, ctl, thr_p, CAP, k, tid);
    for (int i = tid; i < HT_NB; i += NT) hist[i] = 0;
    if (tid == 0) ctl[C_BSTAR] = 0;
    __syncthreads();
    bstar = 0;
    thr_hi = 0;
    thr_f = -INFINITY;
    return reinterpret_cast<char *>(hist + HT_NB);
  }

  // The threshold snapshot only changes inside end_round (compaction / update / flush), where
  // every thread refreshes it after the closing barrier: nothing to read per round.
  __device__ __forceinline__ void begin_round() {}
  __device__ __forceinline__ void refresh_threshold() {
    thr_hi = (uint32_t)(*thr_p >> 32);
    bstar = ctl[C_BSTAR];
    thr_f = bstar > 0 ? bucket_floor(bstar) : -INFINITY;
  }

  // one candidate per lane; returns true if it was kept (count kept lanes per wave with
  // __ballot and pass the sum to end_round)
  // counted: the candidate is in the histogram already (cold start below)
  __device__ __forceinline__ bool offer(bool valid, float score, uint32_t slot, bool counted = false) {
    const uint32_t ob = f2ord(score);
    const int b = score_bucket(score);
    const bool pass = sort_mode ? (ob >= thr_hi) : (b >= bstar);
    if (valid && pass) {
      const int s = atomicAdd(&ctl[C_FILL], 1);
      keys[s] = ((u64)ob << 32) | (u64)slot;
      if (!sort_mode && !counted) atomicAdd(&hist[b], 1);
      return true;
    }
    return false;
  }

  // Cold start of a stream whose first round holds several times k candidates: count ALL of
  // them in the histogram first (cold_count, every wave, no barrier), then fix the threshold
  // bucket from the counts (cold_threshold, all threads) and store only the candidates at or
  // above it, with offer(..., counted = true) -- instead of appending everything 64 per wave at
  // a time and compacting on the way. Candidates below the bucket leave the histogram again, so
  // it counts exactly what is stored. If what is left does not fit the buffer (a crowded
  // threshold bucket), the free-running appends below run into a free_sync, whose compaction
  // cannot free a slot, and the top-k falls back to exact sort-and-truncate flushes.
  __device__ __forceinline__ void cold_count(bool valid, float score) {
    if (valid) atomicAdd(&hist[score_bucket(score)], 1);
  }
  __device__ __forceinline__ void cold_threshold() {
    __syncthreads();
    update_bstar();
    refresh_threshold();
    for (int i = tid; i < bstar; i += NT) hist[i] = 0;
    __syncthreads();
  }

  // the test offer() applies, against the current snapshot (scores are never NaN)
  __device__ __forceinline__ bool passes(float score) const {
    return sort_mode ? (f2ord(score) >= thr_hi) : (score >= thr_f);
  }

  __device__ __forceinline__ void update_bstar() {
    constexpr int BPT = HT_NB / NT;  // buckets per thread (2 or 1), highest buckets in thread 0
    int h[BPT], s = 0;
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      h[u] = hist[HT_NB - 1 - (tid * BPT + u)];
      s += h[u];
    }
    int tot;
    int above = block_excl_scan<NW>(s, ctl + C_PART_B, tid, tot);
    if (above < k && above + s >= k) {
      int b = HT_NB - 1 - tid * BPT;
#pragma unroll
      for (int u = 0; u < BPT; ++u) {
        above += h[u];
        if (above >= k) break;
        --b;
      }
      ctl[C_BSTAR] = b;
    }
    __syncthreads();
  }

  // ---- free-running appends -----------------------------------------------------------------
  // The waves of a workgroup append on their own, a row of candidates at a time, without a
  // barrier: free_append reserves the slots of the wave with one atomic and fails -- writing
  // nothing -- when the buffer cannot take them. The failing wave then asks for a free_sync
  // (caller's flag + barrier: every wave must join), which compacts (or, with ties, sorts
  // and truncates), moves the threshold and zeroes the free tail; a failed reservation leaves
  // its slots below CAP empty (0), which every consumer skips. C_FILL may run past CAP.
  __device__ __forceinline__ bool free_append(bool p, float score, uint32_t slot, bool counted = false) {
    const unsigned long long m = __ballot(p);
    const int c = __popcll(m);
    int base = 0;
    if (lane == 0) base = atomicAdd(&ctl[C_FILL], c);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + c > CAP) return false;
    if (p) {
      const int b = score_bucket(score);
      keys[base + __popcll(m & ((1ull << lane) - 1ull))] = ((u64)f2ord(score) << 32) | (u64)slot;
      if (!sort_mode && !counted) atomicAdd(&hist[b], 1);
    }
    return true;
  }
  // The same in two steps, for a wave that offers several rows at once: one reservation of c
  // slots (c <= 64, wave-uniform; -1: the buffer cannot take them -- nothing was written, go
  // through free_append and the sync it asks for), then the rows' candidates one after the
  // other into base, base + popcount(row 0), ...
  __device__ __forceinline__ int free_reserve(int c) {
    int base = 0;
    if (lane == 0) base = atomicAdd(&ctl[C_FILL], c);
    base = __builtin_amdgcn_readfirstlane(base);
    return base + c > CAP ? -1 : base;
  }
  __device__ __forceinline__ void free_write(bool p, unsigned long long m, float score, uint32_t slot,
                                             int base) {
    if (p) {
      keys[base + __popcll(m & ((1ull << lane) - 1ull))] = ((u64)f2ord(score) << 32) | (u64)slot;
      if (!sort_mode) atomicAdd(&hist[score_bucket(score)], 1);
    }
  }
  // all threads, after a barrier that every wave reached
  __device__ __forceinline__ void free_sync() {
    const int cf = ctl[C_FILL];
    __syncthreads();
    fill = cf < CAP ? cf : CAP;
    if (tid == 0) ctl[C_FILL] = fill;
    __syncthreads();
    if (!sort_mode) {
      fill = compact();
      if (fill > CAP - ROUND_VECS) {   // ties defeat the buckets: exact flushes from now on
        sort_mode = true;
        tk.slot_ids = slot_ids;
        tk.conv_from = 0;
      }
    }
    if (sort_mode && fill > CAP - ROUND_VECS) fill = tk.flush(tid);
    for (int i = fill + tid; i < CAP; i += NT) keys[i] = 0ull;
    __syncthreads();
    refresh_threshold();
  }
  // before finish(): the fill level as the appends left it (no reservation is pending)
  __device__ __forceinline__ void free_done() {
    __syncthreads();
    const int cf = ctl[C_FILL];
    fill = cf < CAP ? cf : CAP;
    __syncthreads();
    if (tid == 0) ctl[C_FILL] = fill;
    __syncthreads();
  }

  // drop every buffered key whose bucket is below bstar; returns the new fill
  __device__ __forceinline__ int compact() {
    update_bstar();
    const int bs = ctl[C_BSTAR];
    u64 kk[PER];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * NT;
      kk[u] = i < fill ? keys[i] : 0ull;
      if (kk[u] != 0ull && score_bucket(ord2f((uint32_t)(kk[u] >> 32))) < bs) kk[u] = 0ull;
      cnt += kk[u] != 0ull;
    }
    int tot;
    int pos = block_excl_scan<NW>(cnt, ctl + C_PART_C, tid, tot);  // barrier inside: all loaded
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (kk[u] != 0ull) keys[pos++] = kk[u];
    if (tid == 0) ctl[C_FILL] = tot;
    __syncthreads();
    return tot;
  }

  // `appended` = candidates kept by THIS wave in the round (wave-uniform). The fill level is
  // tracked identically in every thread from parity-double-buffered per-wave counts, so the
  // compaction decision cannot race with a faster wave's next round.
  __device__ __forceinline__ void end_round(int appended) {
    if (lane == 0) ctl[C_WCNT + parity * NW + wave] = appended;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) fill += ctl[C_WCNT + parity * NW + w];
    if (fill > CAP - ROUND_VECS) {
      if (!sort_mode) {
        fill = compact();
        if (fill > CAP - ROUND_VECS) {   // ties defeat the buckets: exact flushes from now on
          sort_mode = true;
          tk.slot_ids = slot_ids;
          tk.conv_from = 0;
        }
      }
      if (sort_mode && fill > CAP - ROUND_VECS) fill = tk.flush(tid);
      refresh_threshold();
    } else if (!sort_mode && (round_no & 63) == 63) {   // rarely: compactions raise the threshold anyway (every 4th round cost 3 % of the scan)
      update_bstar();
      refresh_threshold();
    }
    parity ^= 1;
    ++round_no;
  }

  // Exact top-k as an UNORDERED SET: no sort. After the compaction every key above the
  // threshold bucket is in (fewer than k of them, by the definition of bstar); the keys of
  // the threshold bucket are ranked among themselves by pairwise counting (ids resolved
  // first, so ties break by id exactly as in the sorted order) and the best k - n_above of
  // them fill the rest of the row. Rows hold min(k, candidates) hits in an unspecified
  // order, then -FLT_MAX / -1 padding. `scratch`: CAP keys of LDS that are dead by now.
  // pf (ScanPostFilter, common.hpp; pf->idpay != nullptr): the precursor-window post-filter applied
  // to the k hits right here -- the ids AND their window values come from pf->idpay[slot] (one 8-byte
  // gather instead of slot_ids' 4), only the passing hits are written (I32, unordered, compacted at
  // the front of the row through an LDS counter) and pf->count[q] says how many. The SELECTION of the
  // k hits is untouched (filter after top-k). Needs CAP more bytes of scratch behind the CAP keys.
  __device__ __forceinline__ void finish_set(float *D, int64_t *I64, int32_t *I32, u64 *scratch,
                                             const ScanPostFilter *pf = nullptr, int q = 0) {
    __syncthreads();
    const bool filt = pf != nullptr && pf->idpay != nullptr && !out_keys && I32 != nullptr;
    if (sort_mode) {               // exact flushes were in use: the sorted row is a valid set
      tk.emit_keys = out_keys;
      tk.finish(out_keys ? nullptr : D, I64, out_keys ? nullptr : I32, tid);
      if (filt && tid == 0) pf->count[q] = -1;       // the k unfiltered hits: the rescoring filters this row
      return;
    }
    double f_q = 0.0;
    uint8_t *sflag = reinterpret_cast<uint8_t *>(scratch + CAP);
    if (filt) {
      f_q = pf->q_pmz[q];
      if (tid == 0) ctl[C_USER] = 0;                 // (update_bstar's barriers order it before the first append)
    }
    // the last compaction, fused: keys below the threshold bucket are dropped in registers (no
    // write-back, no second pass over the buffer)
    update_bstar();
    const int bs = ctl[C_BSTAR];
    u64 kk[PER];
    int32_t idv[PER];
    bool okf[PER];       // (filt) the hit passes the window
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * NT;
      kk[u] = i < fill ? keys[i] : 0ull;
      if (kk[u] != 0ull && score_bucket(ord2f((uint32_t)(kk[u] >> 32))) < bs) kk[u] = 0ull;
      okf[u] = true;
    }
    if (filt) {
      int2 ip[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u)   // all gathers of a thread in flight together
        ip[u] = kk[u] != 0ull ? pf->idpay[(uint32_t)kk[u]] : make_int2(0, 0);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        idv[u] = ip[u].x;
        okf[u] = ip[u].x >= 0 && precursor_ok(f_q, __int_as_float(ip[u].y), pf->charge, pf->tol, pf->mode);
      }
    } else {
#pragma unroll
      for (int u = 0; u < PER; ++u)   // all gathers of a thread in flight together
        idv[u] = (kk[u] != 0ull && slot_ids) ? slot_ids[(uint32_t)kk[u]] : 0;
    }
    int na = 0, nb = 0;
    bool above[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (kk[u] != 0ull) {
        if (slot_ids)
          kk[u] = (kk[u] & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (uint32_t)idv[u]);
        above[u] = score_bucket(ord2f((uint32_t)(kk[u] >> 32))) > bs;
        na += above[u];
        nb += !above[u];
      } else {
        above[u] = false;
      }
    }
    int tot;
    const int pre = block_excl_scan<NW>(na | (nb << 16), ctl + C_PART_C, tid, tot);
    const int n_above = tot & 0xffff, n_bound = tot >> 16;
    int pa = pre & 0xffff, pb = pre >> 16;
    auto emit = [&](int pos, u64 key, bool pass) {
      if (filt) {               // survivors only, wherever the counter puts them (a set)
        if (pass) I32[atomicAdd(&ctl[C_USER], 1)] = (int32_t)key_id(key);
        return;
      }
      if (out_keys) {
        I64[pos] = (int64_t)key;
        return;
      }
      if (D) D[pos] = key_score(key);
      if (I64) I64[pos] = (int64_t)key_id(key);
      if (I32) I32[pos] = (int32_t)key_id(key);
    };
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (kk[u] == 0ull) continue;
      if (above[u]) {
        emit(pa++, kk[u], okf[u]);
      } else {
        if (filt) sflag[pb] = (uint8_t)okf[u];
        scratch[pb++] = kk[u];
      }
    }
    __syncthreads();
    const int r = k - n_above;          // > 0
    const int take = n_bound < r ? n_bound : r;
    if (n_bound <= r) {                  // block-uniform: the whole threshold bucket is in
      for (int i = tid; i < n_bound; i += NT) emit(n_above + i, scratch[i], !filt || sflag[i] != 0);
    } else if (n_bound <= NT) {          // block-uniform
      // rank = number of larger keys, counted by ALL threads: the bucket's keys padded to a power
      // of two n_pad, the NT / n_pad groups of n_pad threads each count over their share of the
      // keys (a wave reads one key at a time: LDS broadcasts), partial counts through the dead
      // histogram. (One thread per key walking the whole bucket left most of the workgroup idle:
      // 0.43 ms of a 131 072-row shard scan.)
      int n_pad = 64;
      while (n_pad < n_bound) n_pad <<= 1;
      const int P = NT / n_pad, i = tid & (n_pad - 1), p = tid / n_pad;
      const int seg = (n_bound + P - 1) / P, j0 = p * seg, j1 = min(n_bound, j0 + seg);
      const u64 mine = i < n_bound ? scratch[i] : ~0ull;
      int c = 0;
      for (int j = j0; j < j1; ++j) c += scratch[j] > mine;
      hist[p * n_pad + i] = c;
      __syncthreads();
      if (tid < n_bound) {
        int rank = 0;
        for (int pp = 0; pp < P; ++pp) rank += hist[pp * n_pad + tid];
        if (rank < take) emit(n_above + rank, scratch[tid], !filt || sflag[tid] != 0);
      }
    } else {
      for (int i = tid; i < n_bound; i += NT) {
        const u64 key = scratch[i];
        int rank = 0;
        for (int j = 0; j < n_bound; ++j) rank += scratch[j] > key;
        if (rank < take) emit(n_above + rank, key, !filt || sflag[i] != 0);
      }
    }
    if (filt) {                 // the row's length; nothing is padded behind it
      __syncthreads();
      if (tid == 0) pf->count[q] = ctl[C_USER];
      return;
    }
    for (int i = n_above + take + tid; i < k; i += NT) {
      if (out_keys) {
        I64[i] = 0;
        continue;
      }
      if (D) D[i] = -3.402823466e+38f;
      if (I64) I64[i] = -1;
      if (I32) I32[i] = -1;
    }
  }

  __device__ __forceinline__ void finish(float *D, int64_t *I64, int32_t *I32) {
    __syncthreads();
    if (!sort_mode) {
      fill = compact();          // typically leaves k .. k + one bucket's population
      tk.slot_ids = slot_ids;    // every surviving key still carries its storage slot
      tk.conv_from = 0;
    }
    tk.emit_keys = out_keys;
    tk.finish(out_keys ? nullptr : D, I64, out_keys ? nullptr : I32, tid);
  }
};

}  // namespace asl
