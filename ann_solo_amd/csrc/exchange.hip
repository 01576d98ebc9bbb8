// exchange.hip -- device side of the TWO-PHASE EXACT top-k exchange of the list-sharded search
// (no reference counterpart: /root/reference/src/ann_solo/spectral_library.py:494 uses device 0
// only; this serves SURVEY.md 8(e) / the north star's "RCCL exchange of per-shard top-k").
//
// Every shard holds, per query, its local top-k as a row of packed 64-bit keys (order-preserving
// score bits << 32 | ~id: larger key = better hit, keys are unique across shards because ids
// are). Shipping all world * k keys of a query to its owner moves 8 * world * k bytes; the owner
// keeps k of them. Instead:
//
//   phase 1  every shard sends the head of its row: up to kp - 1 of its best keys (all keys at or
//            above a score-bucket floor chosen so that at most kp - 1 qualify -- no sorting) and,
//            in the row's last slot, T = its best UNSENT key (0: nothing was held back). kp =
//            ceil(2 k / world): about 2 k keys per query on the wire.
//   bound    the owner merges the heads: B = the k-th best key it has seen (0 if it has seen
//            fewer than k). Every unsent key of shard s is <= T_s. If T_s < B nothing shard s
//            held back can be among the k best of the union; otherwise the owner asks that
//            shard for its keys above B.
//   phase 2  the bounds travel back (8 bytes per (query, shard)), the shards answer with the
//            held-back keys above the bound -- usually none -- compacted into one fixed-size
//            buffer per destination, and the owner merges again where it asked.
//
// The result is the exact top-k of the union: a key that is never shipped is below a bound B
// that k shipped keys reach. If a destination's phase-2 buffer overflows, a flag is raised and
// the caller repeats the batch with the full exchange (ann_solo_amd/distributed.py).
//
//   phase 3  (round 5) the shards may scan with a SHARD-SIDE k_s < k (512 of 1024 at 8 ranks): a
//            shard sees an eighth of a query's candidates, its threshold rises late, and the
//            appends of a k-deep row cost ~1 ms per step (profiles/r04_shard_scan_probe.txt).
//            A row that is full (k_s keys) may have dropped keys; all of them are below M = the
//            row's smallest key, which travels in the head next to T. After phase 2 the owner
//            holds the exact top k of the union of the k_s-rows and B' = its k-th best key (0
//            if fewer). If M_s < B' shard s dropped nothing that matters; otherwise -- a fraction
//            of a percent of the queries -- the owner sends (B', M_s), the shard scans that query
//            again with the full k and answers with its keys strictly between the two, and the
//            owner merges once more. Still exact: a key that never travels is below a bound that
//            k travelled keys reach.
#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"

namespace asl {

constexpr u64 XK_NONE = ~0ull;      // bound meaning "send nothing"

// ---- phase 1: head of a row ---------------------------------------------------------------
// K [nrows, k] (0 = empty, any order) -> head [nrows, kp]: slots 0 .. nkeep-1 the kept keys (0
// padded), slot kp-1 = T (the best key held back); with_min (phase 3): slot kp-2 = M, the row's
// smallest key when the row is full (k keys: the scan may have dropped keys, all below M), else 0;
// nkeep = kp - 1 - with_min. floor_out [nrows]: the row's bucket floor.
// The kept set = all keys whose score bucket (hist_topk.hpp: 512 buckets over [-0.25, 1)) is at
// or above the lowest bucket floor that admits at most kp - 1 keys. The held-back keys are not
// copied anywhere: they are the keys of K below the floor, and phase 2 (keys_extras_kernel) reads
// them from K itself -- only for the rows an owner asks about. (Until the second half of round 4
// the split also wrote them out as a second [nrows, k] array: the kernel is HBM-bound, and that
// was 1.07 of its 2.4 GB per 131 072 rows.)
// One WAVE per row, no barrier: PER keys per lane in registers, a 512-bucket histogram of the
// wave in LDS (ds_add), cumulative counts from the top by a DPP scan of eight buckets per lane,
// ballots for the compaction. (The first version -- a 256-thread workgroup per row with three
// barriers -- took 13.6 us per row: 1.78 ms for the 131 072 rows of a rank at 8 GPUs.)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t xk_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t xk_scan(uint32_t x) {      // inclusive, 64 lanes
  uint32_t t = x + xk_dpp<0x111, 0xf, 0xf>(x);
  t += xk_dpp<0x112, 0xf, 0xf>(x);
  t += xk_dpp<0x113, 0xf, 0xf>(x);
  t += xk_dpp<0x114, 0xf, 0xe>(t);
  t += xk_dpp<0x118, 0xf, 0xc>(t);
  t += xk_dpp<0x142, 0xa, 0xf>(t);
  t += xk_dpp<0x143, 0xc, 0xf>(t);
  return t;
}

constexpr int XS_WAVES = 4;
template <int PER>
__global__ __launch_bounds__(64 * XS_WAVES) void keys_split_kernel(const u64 *__restrict__ K, int64_t nrows,
                                                                   int k, int kp, int with_min,
                                                                   u64 *__restrict__ head,
                                                                   int32_t *__restrict__ floor_out) {
  __shared__ int s_hist[XS_WAVES][HT_NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * XS_WAVES + wave;
  if (row >= nrows) return;
  int *hist = s_hist[wave];
  u64 kk[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = lane + u * 64;
    kk[u] = i < k ? K[row * k + i] : 0ull;
  }
#pragma unroll
  for (int u = 0; u < HT_NB / 64; ++u) hist[lane + u * 64] = 0;
  __builtin_amdgcn_wave_barrier();      // the zeroes are in LDS before any lane's atomic
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (kk[u]) atomicAdd(&hist[score_bucket(key_score(kk[u]))], 1);
  __builtin_amdgcn_wave_barrier();      // ... and every atomic before the counts are read back
  // lane L owns buckets 511 - 8 L .. 504 - 8 L (descending); `above` = keys in higher buckets
  int h[8], mine = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    h[u] = hist[HT_NB - 1 - (lane * 8 + u)];
    mine += h[u];
  }
  int cum = (int)xk_scan((uint32_t)mine) - mine;
  const int nkeep = kp - 1 - (with_min ? 1 : 0);
  const int cap = nkeep;
  int fl = HT_NB;                       // the lowest bucket of mine that still admits <= cap keys
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    cum += h[u];
    if (cum <= cap) fl = HT_NB - 1 - (lane * 8 + u);
  }
  // the wave's floor = the lowest such bucket of the LAST lane whose first bucket still fits
  // (cumulative counts only grow): lanes are ordered from the top bucket down
  const unsigned long long okm = __ballot(fl < HT_NB);
  int floor_b = HT_NB;
  if (okm) {
    const int last = 63 - __builtin_clzll(okm);      // ok lanes form a prefix 0 .. last
    floor_b = __builtin_amdgcn_readlane(fl, last);
  }
  int na = 0, nv = 0;
  u64 best = 0ull, least = XK_NONE;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const bool valid = kk[u] != 0ull;
    const bool a = valid && score_bucket(key_score(kk[u])) >= floor_b;
    const unsigned long long ma = __ballot(a);
    if (a) head[row * kp + na + __popcll(ma & below)] = kk[u];
    if (valid && !a) best = kk[u] > best ? kk[u] : best;
    if (valid) least = kk[u] < least ? kk[u] : least;
    na += __popcll(ma);
    nv += __popcll(__ballot(valid));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const u64 o = (u64)__shfl_xor((unsigned long long)best, off);
    best = o > best ? o : best;
  }
  if (with_min) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const u64 o = (u64)__shfl_xor((unsigned long long)least, off);
      least = o < least ? o : least;
    }
  }
  for (int i = na + lane; i < nkeep; i += 64) head[row * kp + i] = 0ull;
  if (lane == 0) {
    if (with_min) head[row * kp + kp - 2] = nv >= k ? least : 0ull;
    head[row * kp + kp - 1] = best;
    floor_out[row] = floor_b;
  }
}

// ---- owner: merge of the heads (+ phase-2 / phase-3 answers) ---------------------------------
// heads [S, nq, kp]: kin key slots per head, then (mslot >= 0) M and (tslot >= 0) T; answers:
// xbuf [XS, nq + xcap] as keys_extras_kernel / keys_rescan_kernel fill it (per source nq header
// words count << 32 | start, then xcap payload slots).
// Round 1 (bounds != null): out_keys [nq, k] = the best k keys seen (set, 0 padded);
//   bounds [S, nq] = B if shard s must answer (T_s > B) else XK_NONE; need[q] = any shard asked.
// Round 2 (I != null): I [nq, k] = ids of the exact top-k (set, -1 padded), D optional scores;
//   queries with need[q] == 0 only convert prev_keys [nq, k]. With req != null (phase 3 armed):
//   fin_keys [nq, k] = the keys behind I, req [S, nq, 2] = (B', M_s) where shard s must scan the
//   query again with the full k (M_s > B' = the k-th best key of the result, 0 if fewer than k)
//   else (XK_NONE, 0), need3[q] = any such shard, *n3 += the number of requests.
struct MergeArgs {
  const u64 *heads;
  int S, nq, kp, kin, tslot, mslot, k;
  const u64 *xbuf;
  int XS;
  long long xcap;
  const u64 *prev_keys;
  int32_t *need;
  u64 *out_keys, *bounds;
  int64_t *I;
  float *D;
  int sorted;
  u64 *fin_keys, *req;
  int32_t *need3;
  unsigned int *n3;
};

template <int CAP>
__global__ __launch_bounds__(HT_NT) void keys_merge_kernel(const MergeArgs a) {
  using TopK = HistTopK<CAP, HT_NT * 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ u64 s_min;
  __shared__ int s_cnt, s_any;
  const int tid = threadIdx.x, q = blockIdx.x;
  const int S = a.S, nq = a.nq, kp = a.kp, k = a.k;
  const u64 *heads = a.heads;
  const bool round2 = a.I != nullptr;
  // B = the k-th best key of a row of <= k keys (0 when it holds fewer): every thread gets it
  auto row_bound = [&](const u64 *row) -> u64 {
    if (tid == 0) {
      s_min = XK_NONE;
      s_cnt = 0;
      s_any = 0;
    }
    __syncthreads();
    u64 m = XK_NONE;
    int c = 0;
    for (int i = tid; i < k; i += HT_NT) {
      const u64 key = row[i];
      if (key) {
        m = key < m ? key : m;
        ++c;
      }
    }
    atomicMin(&s_min, m);
    atomicAdd(&s_cnt, c);
    __syncthreads();
    return s_cnt >= k ? s_min : 0ull;
  };
  // phase 3: which shards may have dropped a key that belongs to the result?
  auto third = [&](const u64 *row) {
    const u64 B = row_bound(row);
    for (int s = tid; s < S; s += HT_NT) {
      const u64 M = heads[((size_t)s * nq + q) * kp + a.mslot];
      const bool ask = M != 0ull && M > B;
      a.req[((size_t)s * nq + q) * 2] = ask ? B : XK_NONE;
      a.req[((size_t)s * nq + q) * 2 + 1] = ask ? M : 0ull;
      if (ask) {
        s_any = 1;
        atomicAdd(a.n3, 1u);
      }
    }
    if (a.fin_keys)
      for (int i = tid; i < k; i += HT_NT) a.fin_keys[(size_t)q * k + i] = row[i];
    __syncthreads();
    if (tid == 0) a.need3[q] = s_any;
  };
  if (round2 && a.need && a.need[q] == 0 && !a.sorted) {   // nothing was asked for: the earlier set stands
    const u64 *prev = a.prev_keys + (size_t)q * k;
    if (a.req) third(prev);
    for (int i = tid; i < k; i += HT_NT) {
      const u64 key = prev[i];
      a.I[(size_t)q * k + i] = key ? (int64_t)key_id(key) : -1;
      if (a.D) a.D[(size_t)q * k + i] = key ? key_score(key) : -3.402823466e+38f;
    }
    return;
  }
  TopK top;
  top.init(smem, k, nullptr, tid);
  top.out_keys = true;
  auto stream = [&](auto &&key_at, int total) {   // total: a multiple of nothing in particular
    for (int base = 0; base < total; base += HT_NT * 2) {
      top.begin_round();
      int appended = 0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int v = base + u * HT_NT + tid;
        const u64 key = v < total ? key_at(v) : 0ull;
        const bool take = top.offer(key != 0ull, key ? key_score(key) : 0.0f, (uint32_t)key);
        appended += __popcll(__ballot(take));
      }
      top.end_round(appended);
    }
  };
  // heads: the M / T slots skipped. When all S * kin slots fit the key buffer
  // (the two-phase exchange's own shape: ceil(2k / S) keys from each of S shards) they go straight
  // into it, counted in the histogram -- no rounds, no per-key reservations; empty slots stay
  // (every consumer of the buffer skips zeros). Otherwise 64-entry chunks of the S lists
  // interleaved through the streaming offers, as topk_merge_hist_kernel does.
  const int kin = a.kin;
  const bool answers = round2 && a.xbuf && (!a.need || a.need[q] != 0);
  if (S * kin <= CAP) {
    const int total = S * kin;
    for (int i = tid; i < total; i += HT_NT) {
      const int s = i / kin, j = i - s * kin;
      const u64 key = heads[((size_t)s * nq + q) * kp + j];
      top.keys[i] = key;
      if (key) atomicAdd(&top.hist[score_bucket(key_score(key))], 1);
    }
    __syncthreads();
    top.fill = total;
    if (tid == 0) top.ctl[TopK::C_FILL] = total;
    __syncthreads();
    if (answers) {      // answers follow through the streaming offers: they need a round's worth of room
      top.begin_round();
      top.end_round(0);
    }
  } else {
    const int kc = (kin + 63) >> 6;
    stream([&](int v) -> u64 {
      const int c = v >> 6, s = c % S, j = (c / S) * 64 + (v & 63);
      return j < kin ? heads[((size_t)s * nq + q) * kp + j] : 0ull;
    }, kc * S * 64);
  }
  if (answers) {
    for (int s = 0; s < a.XS; ++s) {                // block-uniform
      const u64 *src = a.xbuf + (size_t)s * ((size_t)nq + (size_t)a.xcap);
      const u64 h = src[q];
      const int cnt = (int)(h >> 32);
      const u64 *pay = src + nq + (size_t)(uint32_t)h;
      if (cnt > 0) stream([&](int v) -> u64 { return v < cnt ? pay[v] : 0ull; }, cnt);
    }
  }
  u64 *row = (round2 ? reinterpret_cast<u64 *>(a.I) : a.out_keys) + (size_t)q * k;
  if (round2 && a.sorted)      // rows under (score desc, id asc), as the unsharded index returns them
    top.finish(nullptr, reinterpret_cast<int64_t *>(row), nullptr);
  else
    top.finish_set(nullptr, reinterpret_cast<int64_t *>(row), nullptr,
                   reinterpret_cast<u64 *>(smem + TopK::lds_bytes()));
  __syncthreads();
  __threadfence_block();
  if (round2) {                                  // keys -> ids in place (every thread its own slots)
    if (a.req) {
      third(row);
      __syncthreads();
    }
    for (int i = tid; i < k; i += HT_NT) {
      const u64 key = row[i];
      if (a.D) a.D[(size_t)q * k + i] = key ? key_score(key) : -3.402823466e+38f;
      a.I[(size_t)q * k + i] = key ? (int64_t)key_id(key) : -1;
    }
    return;
  }
  // round 1: B = the k-th best key seen (0 when fewer than k were), then the question to every shard
  const u64 B = row_bound(row);
  for (int s = tid; s < S; s += HT_NT) {
    const u64 T = heads[((size_t)s * nq + q) * kp + a.tslot];
    const bool ask = T > B;                      // T == 0: nothing held back
    a.bounds[(size_t)s * nq + q] = ask ? B : XK_NONE;
    if (ask) s_any = 1;
  }
  __syncthreads();
  if (tid == 0) a.need[q] = s_any;
}

// ---- phase 2 / phase 3 on the shard: answers into one buffer per destination ------------------
// xbuf [W, nq + xcap]: per destination nq header words (count << 32 | start) followed by xcap
// payload slots; cursor [W] (zeroed by the caller) hands out the payload; *overflow is raised when
// a destination's payload is full. kk[PER]: this thread's keys of the row, 0 = not wanted.
template <int PER>
__device__ __forceinline__ void answer_row(u64 (&kk)[PER], int c, int dst, int q, int nq, long long xcap,
                                           u64 *__restrict__ xbuf, unsigned int *__restrict__ cursor,
                                           int32_t *__restrict__ overflow, int *part, unsigned int *s_start) {
  const int tid = threadIdx.x;
  u64 *hdr = xbuf + (size_t)dst * (nq + xcap), *pay = hdr + nq;
  int tot;
  int pos = block_excl_scan<4>(c, part, tid, tot);
  if (tid == 0) {
    unsigned int st = tot ? atomicAdd(&cursor[dst], (unsigned int)tot) : 0u;
    if ((long long)st + tot > xcap) {            // no room: the caller falls back to the full exchange
      *overflow = 1;
      st = 0xFFFFFFFFu;
    }
    *s_start = st;
    hdr[q] = st == 0xFFFFFFFFu ? 0ull : (((u64)tot << 32) | st);
  }
  __syncthreads();
  const unsigned int st = *s_start;
  if (st == 0xFFFFFFFFu) return;
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (kk[u]) pay[st + pos++] = kk[u];
}

// phase 2: the held-back keys (below the row's bucket floor) above the owner's bound. Rows are
// destination-major: row = dst * nq + q.
__global__ __launch_bounds__(256) void keys_extras_kernel(const u64 *__restrict__ K, int k,
                                                          const int32_t *__restrict__ floor_in,
                                                          const u64 *__restrict__ bounds, int nq,
                                                          long long xcap, u64 *__restrict__ xbuf,
                                                          unsigned int *__restrict__ cursor,
                                                          int32_t *__restrict__ overflow) {
  __shared__ int part[8];
  __shared__ unsigned int s_start;
  const int tid = threadIdx.x;
  const size_t row = blockIdx.x;
  const int dst = (int)(row / nq), q = (int)(row % nq);
  const u64 B = bounds[row];
  if (B == XK_NONE) {                            // block-uniform
    if (tid == 0) xbuf[(size_t)dst * (nq + xcap) + q] = 0ull;
    return;
  }
  constexpr int PER = 8;
  const int floor_b = floor_in[row];
  u64 kk[PER];
  int c = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * 256;
    kk[u] = i < k ? K[row * k + i] : 0ull;
    // held back = below the row's bucket floor (the split's own test); wanted = above the bound
    if (kk[u] <= B || score_bucket(key_score(kk[u])) >= floor_b) kk[u] = 0ull;
    c += kk[u] != 0ull;
  }
  answer_row<PER>(kk, c, dst, q, nq, xcap, xbuf, cursor, overflow, part, &s_start);
}

// phase 3: K3 [n3, k] = the FULL-k rows of the queries an owner asked about, rowidx [n3] = their
// destination-major row (dst * nq + q), req [W * nq, 2] = (B', M) as the owners sent them. The
// answer of a row = its keys strictly between B' and M (everything at or above M travelled in
// phases 1-2). xbuf must be zeroed by the caller (rows nobody asked about keep an empty header).
__global__ __launch_bounds__(256) void keys_rescan_kernel(const u64 *__restrict__ K3, int k,
                                                          const int64_t *__restrict__ rowidx,
                                                          const u64 *__restrict__ req, int nq,
                                                          long long xcap, u64 *__restrict__ xbuf,
                                                          unsigned int *__restrict__ cursor,
                                                          int32_t *__restrict__ overflow) {
  __shared__ int part[8];
  __shared__ unsigned int s_start;
  const int tid = threadIdx.x;
  const size_t r = blockIdx.x;
  const int64_t row = rowidx[r];
  const int dst = (int)(row / nq), q = (int)(row % nq);
  const u64 B = req[2 * row], M = req[2 * row + 1];
  constexpr int PER = 8;
  u64 kk[PER];
  int c = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * 256;
    kk[u] = i < k ? K3[r * k + i] : 0ull;
    if (B == XK_NONE || kk[u] <= B || kk[u] >= M) kk[u] = 0ull;
    c += kk[u] != 0ull;
  }
  answer_row<PER>(kk, c, dst, q, nq, xcap, xbuf, cursor, overflow, part, &s_start);
}

// rows of req [nrows, 2] that carry a request, compacted (ascending order is NOT guaranteed; the
// answers are addressed through headers, so any order gives the same result): rowidx, *count
__global__ void req_rows_kernel(const u64 *__restrict__ req, int64_t nrows, int64_t *__restrict__ rowidx,
                                unsigned int *__restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ask = i < nrows && req[2 * i] != XK_NONE;
  const unsigned long long m = __ballot(ask);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  unsigned int base = 0;
  if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(m));
  base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
  if (ask) rowidx[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

int keys_split(const u64 *K, int64_t nrows, int k, int kp, u64 *head, int32_t *floor_out, int with_min) {
  if (nrows <= 0) return ASL_OK;
  const dim3 grid((unsigned)cdiv(nrows, XS_WAVES)), block(64 * XS_WAVES);
  if (k <= 256)
    hipLaunchKernelGGL(keys_split_kernel<4>, grid, block, 0, stream(), K, nrows, k, kp, with_min, head, floor_out);
  else if (k <= 1024)
    hipLaunchKernelGGL(keys_split_kernel<16>, grid, block, 0, stream(), K, nrows, k, kp, with_min, head, floor_out);
  else
    hipLaunchKernelGGL(keys_split_kernel<32>, grid, block, 0, stream(), K, nrows, k, kp, with_min, head, floor_out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

static int launch_merge(const MergeArgs &a) {
  if (a.nq <= 0) return ASL_OK;
  const size_t lds = HistTopK<2048, HT_NT * 2>::lds_bytes() + (size_t)2048 * 8;
  hipLaunchKernelGGL((keys_merge_kernel<2048>), dim3(a.nq), dim3(HT_NT), lds, stream(), a);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// heads [S, nq, kp] with nkeep = kp - 1 - with_min key slots, then M (with_min), then T
int keys_merge(const u64 *heads, int S, int nq, int kp, int k, const u64 *xbuf, long long xcap,
               const u64 *prev_keys, int32_t *need, u64 *out_keys, u64 *bounds, int64_t *I, float *D,
               int sorted, int with_min, u64 *fin_keys, u64 *req, int32_t *need3, unsigned int *n3) {
  MergeArgs a;
  a.heads = heads;
  a.S = S;
  a.nq = nq;
  a.kp = kp;
  a.kin = kp - 1 - (with_min ? 1 : 0);
  a.tslot = kp - 1;
  a.mslot = with_min ? kp - 2 : -1;
  a.k = k;
  a.xbuf = xbuf;
  a.XS = S;
  a.xcap = xcap;
  a.prev_keys = prev_keys;
  a.need = need;
  a.out_keys = out_keys;
  a.bounds = bounds;
  a.I = I;
  a.D = D;
  a.sorted = sorted;
  a.fin_keys = with_min ? fin_keys : nullptr;
  a.req = with_min ? req : nullptr;
  a.need3 = need3;
  a.n3 = n3;
  return launch_merge(a);
}

// phase 3 on the owner: fin_keys [nq, k] (the result of phases 1-2) + the answers xbuf
// [W, nq + xcap] of the W shards -> I / D [nq, k]; queries with need3[q] == 0 keep their set
int keys_merge3(const u64 *fin_keys, int W, int nq, int k, const u64 *xbuf, long long xcap,
                const int32_t *need3, int64_t *I, float *D, int sorted) {
  MergeArgs a;
  a.heads = fin_keys;
  a.S = 1;
  a.nq = nq;
  a.kp = k;
  a.kin = k;
  a.tslot = a.mslot = -1;
  a.k = k;
  a.xbuf = xbuf;
  a.XS = W;
  a.xcap = xcap;
  a.prev_keys = fin_keys;
  a.need = const_cast<int32_t *>(need3);
  a.out_keys = a.bounds = nullptr;
  a.I = I;
  a.D = D;
  a.sorted = sorted;
  a.fin_keys = a.req = nullptr;
  a.need3 = nullptr;
  a.n3 = nullptr;
  return launch_merge(a);
}

int keys_extras(const u64 *K, const int32_t *floor_in, int64_t nrows, int k, const u64 *bounds, int nq,
                long long xcap, u64 *xbuf, unsigned int *cursor, int32_t *overflow) {
  if (nrows <= 0) return ASL_OK;
  hipLaunchKernelGGL(keys_extras_kernel, dim3((unsigned)nrows), dim3(256), 0, stream(), K, k, floor_in, bounds,
                     nq, xcap, xbuf, cursor, overflow);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int keys_rescan(const u64 *K3, int64_t n3, int k, const int64_t *rowidx, const u64 *req, int nq, long long xcap,
                u64 *xbuf, unsigned int *cursor, int32_t *overflow) {
  if (n3 <= 0) return ASL_OK;
  hipLaunchKernelGGL(keys_rescan_kernel, dim3((unsigned)n3), dim3(256), 0, stream(), K3, k, rowidx, req, nq,
                     xcap, xbuf, cursor, overflow);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int req_rows(const u64 *req, int64_t nrows, int64_t *rowidx, unsigned int *count) {
  if (nrows <= 0) return ASL_OK;
  hipLaunchKernelGGL(req_rows_kernel, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, stream(), req, nrows,
                     rowidx, count);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl

using namespace asl;

extern "C" {

// see include/annsolo_mi.h
int asl_keys_split(int64_t nrows, int32_t k, int32_t kp, int32_t with_min, const int64_t *K, int64_t *head,
                   int32_t *floor_out) {
  clear_error();
  const int nkeep = kp - 1 - (with_min ? 1 : 0);
  if (nrows < 0 || k < 1 || k > 2048 || nkeep < 1 || nkeep > k || !K || !head || !floor_out)
    return fail(ASL_ERR_INVALID, "keys_split: need 1 <= k <= 2048, 1 <= key slots <= k and non-null device arrays");
  ASL_TRY(ensure_device());
  return keys_split(reinterpret_cast<const u64 *>(K), nrows, k, kp, reinterpret_cast<u64 *>(head), floor_out,
                    with_min ? 1 : 0);
}

int asl_keys_merge_heads(int32_t S, int32_t nq, int32_t kp, int32_t with_min, int32_t k, const int64_t *heads,
                         int64_t *out_keys, int64_t *bounds, int32_t *need) {
  clear_error();
  if (S < 1 || nq < 0 || kp - (with_min ? 1 : 0) < 2 || k < 1 || k + 768 > 2048 || !heads || !out_keys || !bounds || !need)
    return fail(ASL_ERR_INVALID, "keys_merge_heads: need S >= 1, at least one key slot, 1 <= k <= 1280 and non-null device arrays");
  ASL_TRY(ensure_device());
  return keys_merge(reinterpret_cast<const u64 *>(heads), S, nq, kp, k, nullptr, 0, nullptr, need,
                    reinterpret_cast<u64 *>(out_keys), reinterpret_cast<u64 *>(bounds), nullptr, nullptr, 0,
                    with_min ? 1 : 0, nullptr, nullptr, nullptr, nullptr);
}

int asl_keys_extras(int32_t W, int32_t nq, int32_t k, const int64_t *K, const int32_t *floor_in,
                    const int64_t *bounds, int64_t xcap, int64_t *xbuf, int32_t *cursor, int32_t *overflow) {
  clear_error();
  if (W < 1 || nq < 0 || k < 1 || k > 2048 || xcap < 0 || xcap >= 0xFFFFFFFFLL || !K || !floor_in || !bounds || !xbuf ||
      !cursor || !overflow)
    return fail(ASL_ERR_INVALID, "keys_extras: bad argument");
  ASL_TRY(ensure_device());
  return keys_extras(reinterpret_cast<const u64 *>(K), floor_in, (int64_t)W * nq, k,
                     reinterpret_cast<const u64 *>(bounds), nq, xcap, reinterpret_cast<u64 *>(xbuf),
                     reinterpret_cast<unsigned int *>(cursor), overflow);      // nothing waits here
}

int asl_keys_merge_final(int32_t S, int32_t nq, int32_t kp, int32_t with_min, int32_t k, const int64_t *heads,
                         const int64_t *xbuf, int64_t xcap, const int64_t *prev_keys, const int32_t *need,
                         float *D, int64_t *I, int64_t *fin_keys, int64_t *req, int32_t *need3, int32_t *n3) {
  clear_error();
  if (S < 1 || nq < 0 || kp - (with_min ? 1 : 0) < 2 || k < 1 || k + 768 > 2048 || !heads || !prev_keys || !I)
    return fail(ASL_ERR_INVALID, "keys_merge_final: bad argument");
  if (with_min && (!fin_keys || !req || !need3 || !n3))
    return fail(ASL_ERR_INVALID, "keys_merge_final: phase 3 needs fin_keys, req, need3 and n3");
  ASL_TRY(ensure_device());
  return keys_merge(reinterpret_cast<const u64 *>(heads), S, nq, kp, k, reinterpret_cast<const u64 *>(xbuf),
                    xbuf ? xcap : 0, reinterpret_cast<const u64 *>(prev_keys), const_cast<int32_t *>(need),
                    nullptr, nullptr, I, D, 0, with_min ? 1 : 0, reinterpret_cast<u64 *>(fin_keys),
                    reinterpret_cast<u64 *>(req), need3, reinterpret_cast<unsigned int *>(n3));
}

int asl_keys_rescan(int32_t W, int32_t nq, int32_t k, int64_t n3, const int64_t *K3, const int64_t *rowidx,
                    const int64_t *req, int64_t xcap, int64_t *xbuf, int32_t *cursor, int32_t *overflow) {
  clear_error();
  if (W < 1 || nq < 0 || k < 1 || k > 2048 || n3 < 0 || xcap < 0 || xcap >= 0xFFFFFFFFLL || !req || !xbuf || !cursor ||
      !overflow || (n3 > 0 && (!K3 || !rowidx)))
    return fail(ASL_ERR_INVALID, "keys_rescan: bad argument");
  ASL_TRY(ensure_device());
  return keys_rescan(reinterpret_cast<const u64 *>(K3), n3, k, rowidx, reinterpret_cast<const u64 *>(req), nq, xcap,
                     reinterpret_cast<u64 *>(xbuf), reinterpret_cast<unsigned int *>(cursor), overflow);
}

int asl_keys_merge3(int32_t W, int32_t nq, int32_t k, const int64_t *fin_keys, const int64_t *xbuf, int64_t xcap,
                    const int32_t *need3, float *D, int64_t *I) {
  clear_error();
  if (W < 1 || nq < 0 || k < 1 || k + 768 > 2048 || !fin_keys || !xbuf || !need3 || !I)
    return fail(ASL_ERR_INVALID, "keys_merge3: bad argument");
  ASL_TRY(ensure_device());
  return keys_merge3(reinterpret_cast<const u64 *>(fin_keys), W, nq, k, reinterpret_cast<const u64 *>(xbuf), xcap,
                     need3, I, D, 0);
}

}  // extern "C"
