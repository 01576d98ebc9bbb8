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
//   shard-side k_s < k (round 5). A shard sees 1 / world of a query's candidates, its top-k
//   threshold rises late, and the appends of a k-deep row cost the scan ~0.5-0.9 ms per step at
//   eight ranks (profiles/r05_sim_rank.txt). So the shards scan with k_s (asl_shard_k: 512 of
//   1024 at 8 ranks). A FULL row of k_s keys may have dropped keys, all of them below M = the
//   row's smallest key. Nothing changes for the owner: every key outside a head -- held back or
//   dropped -- is <= T (a full row always holds something back, and M <= T), so "ask shard s iff
//   T_s > B" still covers them. The SHARD completes its answer: where the bound it is sent lies
//   below M (rescan_list_kernel; ~1 % of its rows) it scans that query again with the full k --
//   a launch of fixed size gated by a device-side count, no host round trip -- and answers from
//   that row instead: every key above B that is not in the head. Same collectives, same exact
//   result: the top k of the union of the shards' FULL rows.
#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"

namespace asl {

constexpr u64 XK_NONE = ~0ull;      // bound meaning "send nothing"

// ---- phase 1: head of a row ---------------------------------------------------------------
// K [nrows, k] (0 = empty, any order) -> head [nrows, kp]: slots 0 .. kp-2 the kept keys (0
// padded), slot kp-1 = T (the best key held back); floor_out [nrows]: the row's bucket floor;
// rowmin_out [nrows] (may be null): M = the row's smallest key if the row is FULL (k keys: the
// scan may have dropped keys, all of them below M), else 0.
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
                                                                   int k, int kp, u64 *__restrict__ head,
                                                                   int32_t *__restrict__ floor_out,
                                                                   u64 *__restrict__ rowmin_out) {
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
  const int cap = kp - 1;
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
  if (rowmin_out) {                     // wave-uniform
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const u64 o = (u64)__shfl_xor((unsigned long long)least, off);
      least = o < least ? o : least;
    }
  }
  for (int i = na + lane; i < kp - 1; i += 64) head[row * kp + i] = 0ull;
  if (lane == 0) {
    head[row * kp + kp - 1] = best;
    floor_out[row] = floor_b;
    if (rowmin_out) rowmin_out[row] = nv >= k ? least : 0ull;
  }
}

// ---- owner: merge of the heads (+ phase-2 answers) ------------------------------------------
// heads [S, nq, kp]; extras (round 2): xbuf [S, nq + xcap] as keys_extras_kernel fills it (per
// source nq header words count << 32 | start, then xcap payload slots).
// Round 1 (bounds != null): out_keys [nq, k] = the best k keys seen (set, 0 padded);
//   bounds [S, nq] = B if shard s must answer (T_s > B) else XK_NONE; need[q] = any shard asked.
// Round 2 (I != null): I [nq, k] = ids of the exact top-k (set, -1 padded), D optional scores;
//   queries with need[q] == 0 only convert prev_keys [nq, k].
template <int CAP>
__global__ __launch_bounds__(HT_NT) void keys_merge_kernel(
    const u64 *__restrict__ heads, int S, int nq, int kp, int k, const u64 *__restrict__ xbuf,
    long long xcap, const u64 *__restrict__ prev_keys,
    int32_t *__restrict__ need, u64 *__restrict__ out_keys, u64 *__restrict__ bounds,
    int64_t *__restrict__ I, float *__restrict__ D, int sorted) {
  using TopK = HistTopK<CAP, HT_NT * 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ u64 s_min;
  __shared__ int s_cnt, s_any;
  const int tid = threadIdx.x, q = blockIdx.x;
  const bool round2 = I != nullptr;
  if (round2 && need && need[q] == 0 && !sorted) {   // nothing was asked for: the round-1 set stands
    for (int i = tid; i < k; i += HT_NT) {
      const u64 key = prev_keys[(size_t)q * k + i];
      I[(size_t)q * k + i] = key ? (int64_t)key_id(key) : -1;
      if (D) D[(size_t)q * k + i] = key ? key_score(key) : -3.402823466e+38f;
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
  // heads: slot kp-1 (the held-back key) skipped. When all S * (kp - 1) slots fit the key buffer
  // (the two-phase exchange's own shape: ceil(2k / S) keys from each of S shards) they go straight
  // into it, counted in the histogram -- no rounds, no per-key reservations; empty slots stay
  // (every consumer of the buffer skips zeros). Otherwise 64-entry chunks of the S lists
  // interleaved through the streaming offers, as topk_merge_hist_kernel does.
  const int kin = kp - 1;
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
    if (round2 && xbuf) {      // answers follow through the streaming offers: they need a round's worth of room
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
  if (round2 && xbuf) {
    for (int s = 0; s < S; ++s) {                // block-uniform
      const u64 *src = xbuf + (size_t)s * ((size_t)nq + (size_t)xcap);
      const u64 h = src[q];
      const int cnt = (int)(h >> 32);
      const u64 *pay = src + nq + (size_t)(uint32_t)h;
      if (cnt > 0) stream([&](int v) -> u64 { return v < cnt ? pay[v] : 0ull; }, cnt);
    }
  }
  u64 *row = (round2 ? reinterpret_cast<u64 *>(I) : out_keys) + (size_t)q * k;
  if (round2 && sorted)      // rows under (score desc, id asc), as the unsharded index returns them
    top.finish(nullptr, reinterpret_cast<int64_t *>(row), nullptr);
  else
    top.finish_set(nullptr, reinterpret_cast<int64_t *>(row), nullptr,
                   reinterpret_cast<u64 *>(smem + TopK::lds_bytes()));
  __syncthreads();
  if (round2) {                                  // keys -> ids in place (every thread its own slots)
    __threadfence_block();
    for (int i = tid; i < k; i += HT_NT) {
      const u64 key = row[i];
      if (D) D[(size_t)q * k + i] = key ? key_score(key) : -3.402823466e+38f;
      I[(size_t)q * k + i] = key ? (int64_t)key_id(key) : -1;
    }
    return;
  }
  // round 1: B = the k-th best key seen (0 when fewer than k were), then the question to every shard
  if (tid == 0) {
    s_min = XK_NONE;
    s_cnt = 0;
    s_any = 0;
  }
  __syncthreads();
  __threadfence_block();
  {
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
  }
  __syncthreads();
  const u64 B = s_cnt >= k ? s_min : 0ull;
  for (int s = tid; s < S; s += HT_NT) {
    const u64 T = heads[((size_t)s * nq + q) * kp + kp - 1];
    const bool ask = T > B;                      // T == 0: nothing held back
    bounds[(size_t)s * nq + q] = ask ? B : XK_NONE;
    if (ask) s_any = 1;
  }
  __syncthreads();
  if (tid == 0) need[q] = s_any;
}

// ---- phase 2 on the shard: the keys outside the head above the owner's bound -----------------
// Which rows need a second scan with the full k? Those whose bound lies below the smallest key
// of a FULL row: a dropped key could be above the bound. rows destination-major. rowlist [R]
// (zero-initialised by the caller: slots past the count stay valid row numbers) receives the
// rows, rmap [nrows] their slot or -1, *count the number (the GATE of the scan launch that
// follows: a device-side count, nothing returns to the host); more than R: *overflow = 1 (the
// caller repeats the batch with k-deep rows) and the row keeps rmap = -1.
__global__ void rescan_list_kernel(const u64 *__restrict__ bounds, const u64 *__restrict__ rowmin,
                                   int64_t nrows, int R, int64_t *__restrict__ rowlist,
                                   int32_t *__restrict__ rmap, int *__restrict__ count,
                                   int32_t *__restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool need = false;
  if (i < nrows) {
    const u64 B = bounds[i], M = rowmin[i];
    need = B != XK_NONE && M != 0ull && M > B;
  }
  const unsigned long long m = __ballot(need);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (m && lane == 0) base = atomicAdd(count, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (i < nrows) {
    int slot = -1;
    if (need) {
      slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < R) {
        rowlist[slot] = i;
      } else {
        slot = -1;
        *overflow = 1;
      }
    }
    rmap[i] = slot;
  }
}

// rows are destination-major: row = dst * nq + q. xbuf [W, nq + xcap]: per destination nq
// header words (count << 32 | start) followed by xcap payload slots; cursor [W] (zeroed by the
// caller) hands out the payload; *overflow is raised when a destination's payload is full.
// rmap != null: a row with rmap[row] >= 0 answers from K3 [*, k3], its second scan with the
// full k (every key above the bound that is not in the head: held back or dropped the first
// time), the others from K [*, k] as before.
__global__ __launch_bounds__(256) void keys_extras_kernel(const u64 *__restrict__ K, int k,
                                                          const int32_t *__restrict__ floor_in,
                                                          const u64 *__restrict__ bounds, int nq,
                                                          long long xcap, u64 *__restrict__ xbuf,
                                                          unsigned int *__restrict__ cursor,
                                                          int32_t *__restrict__ overflow,
                                                          const int32_t *__restrict__ rmap,
                                                          const u64 *__restrict__ K3, int k3) {
  __shared__ int part[8];
  __shared__ unsigned int s_start;
  const int tid = threadIdx.x;
  const size_t row = blockIdx.x;
  const int dst = (int)(row / nq), q = (int)(row % nq);
  u64 *hdr = xbuf + (size_t)dst * (nq + xcap), *pay = hdr + nq;
  const u64 B = bounds[row];
  if (B == XK_NONE) {                            // block-uniform
    if (tid == 0) hdr[q] = 0ull;
    return;
  }
  constexpr int PER = 8;
  const int floor_b = floor_in[row];
  const int slot = rmap ? rmap[row] : -1;        // block-uniform
  const u64 *src = slot >= 0 ? K3 + (size_t)slot * k3 : K + row * k;
  const int width = slot >= 0 ? k3 : k;
  u64 kk[PER];
  int c = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * 256;
    kk[u] = i < width ? src[i] : 0ull;
    // outside the head = below the row's bucket floor (the split's own test); wanted = above the bound
    if (kk[u] <= B || score_bucket(key_score(kk[u])) >= floor_b) kk[u] = 0ull;
    c += kk[u] != 0ull;
  }
  int tot;
  int pos = block_excl_scan<4>(c, part, tid, tot);
  if (tid == 0) {
    unsigned int st = tot ? atomicAdd(&cursor[dst], (unsigned int)tot) : 0u;
    if ((long long)st + tot > xcap) {            // no room: the caller falls back to the full exchange
      *overflow = 1;
      st = 0xFFFFFFFFu;
    }
    s_start = st;
    hdr[q] = st == 0xFFFFFFFFu ? 0ull : (((u64)tot << 32) | st);
  }
  __syncthreads();
  const unsigned int st = s_start;
  if (st == 0xFFFFFFFFu) return;
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (kk[u]) pay[st + pos++] = kk[u];
}

int keys_split(const u64 *K, int64_t nrows, int k, int kp, u64 *head, int32_t *floor_out, u64 *rowmin_out) {
  if (nrows <= 0) return ASL_OK;
  const dim3 grid((unsigned)cdiv(nrows, XS_WAVES)), block(64 * XS_WAVES);
  if (k <= 256)
    hipLaunchKernelGGL(keys_split_kernel<4>, grid, block, 0, stream(), K, nrows, k, kp, head, floor_out, rowmin_out);
  else if (k <= 1024)
    hipLaunchKernelGGL(keys_split_kernel<16>, grid, block, 0, stream(), K, nrows, k, kp, head, floor_out, rowmin_out);
  else
    hipLaunchKernelGGL(keys_split_kernel<32>, grid, block, 0, stream(), K, nrows, k, kp, head, floor_out, rowmin_out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int keys_merge(const u64 *heads, int S, int nq, int kp, int k, const u64 *xbuf, long long xcap,
               const u64 *prev_keys, int32_t *need, u64 *out_keys, u64 *bounds, int64_t *I, float *D,
               int sorted) {
  if (nq <= 0) return ASL_OK;
  const size_t lds = HistTopK<2048, HT_NT * 2>::lds_bytes() + (size_t)2048 * 8;
  hipLaunchKernelGGL((keys_merge_kernel<2048>), dim3(nq), dim3(HT_NT), lds, stream(), heads, S, nq, kp,
                     k, xbuf, xcap, prev_keys, need, out_keys, bounds, I, D, sorted);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int keys_extras(const u64 *K, const int32_t *floor_in, int64_t nrows, int k, const u64 *bounds, int nq,
                long long xcap, u64 *xbuf, unsigned int *cursor, int32_t *overflow, const int32_t *rmap,
                const u64 *K3, int k3) {
  if (nrows <= 0) return ASL_OK;
  hipLaunchKernelGGL(keys_extras_kernel, dim3((unsigned)nrows), dim3(256), 0, stream(), K, k, floor_in, bounds,
                     nq, xcap, xbuf, cursor, overflow, rmap, K3, k3);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int rescan_list(const u64 *bounds, const u64 *rowmin, int64_t nrows, int R, int64_t *rowlist, int32_t *rmap,
                int *count, int32_t *overflow) {
  if (nrows <= 0) return ASL_OK;
  hipLaunchKernelGGL(rescan_list_kernel, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, stream(), bounds, rowmin,
                     nrows, R, rowlist, rmap, count, overflow);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl

using namespace asl;

extern "C" {

// see include/annsolo_mi.h
int asl_keys_split(int64_t nrows, int32_t k, int32_t kp, const int64_t *K, int64_t *head, int32_t *floor_out,
                   int64_t *rowmin_out) {
  clear_error();
  if (nrows < 0 || k < 1 || k > 2048 || kp < 2 || kp > k + 1 || !K || !head || !floor_out)
    return fail(ASL_ERR_INVALID, "keys_split: need 1 <= k <= 2048, 2 <= kp <= k + 1 and non-null device arrays");
  ASL_TRY(ensure_device());
  return keys_split(reinterpret_cast<const u64 *>(K), nrows, k, kp, reinterpret_cast<u64 *>(head), floor_out,
                    reinterpret_cast<u64 *>(rowmin_out));
}

int asl_keys_merge_heads(int32_t S, int32_t nq, int32_t kp, int32_t k, const int64_t *heads,
                         int64_t *out_keys, int64_t *bounds, int32_t *need) {
  clear_error();
  if (S < 1 || nq < 0 || kp < 2 || k < 1 || k + 768 > 2048 || !heads || !out_keys || !bounds || !need)
    return fail(ASL_ERR_INVALID, "keys_merge_heads: need S >= 1, kp >= 2, 1 <= k <= 1280 and non-null device arrays");
  ASL_TRY(ensure_device());
  return keys_merge(reinterpret_cast<const u64 *>(heads), S, nq, kp, k, nullptr, 0, nullptr, need,
                    reinterpret_cast<u64 *>(out_keys), reinterpret_cast<u64 *>(bounds), nullptr, nullptr, 0);
}

int asl_keys_rescan_list(int64_t nrows, const int64_t *bounds, const int64_t *rowmin, int32_t R, int64_t *rowlist,
                         int32_t *rmap, int32_t *count, int32_t *overflow) {
  clear_error();
  if (nrows < 0 || R < 1 || !bounds || !rowmin || !rowlist || !rmap || !count || !overflow)
    return fail(ASL_ERR_INVALID, "keys_rescan_list: bad argument");
  ASL_TRY(ensure_device());
  return rescan_list(reinterpret_cast<const u64 *>(bounds), reinterpret_cast<const u64 *>(rowmin), nrows, R, rowlist,
                     rmap, count, overflow);
}

int asl_keys_extras(int32_t W, int32_t nq, int32_t k, const int64_t *K, const int32_t *floor_in,
                    const int64_t *bounds, int64_t xcap, int64_t *xbuf, int32_t *cursor, int32_t *overflow,
                    const int32_t *rmap, const int64_t *K3, int32_t k3) {
  clear_error();
  if (W < 1 || nq < 0 || k < 1 || k > 2048 || xcap < 0 || xcap >= 0xFFFFFFFFLL || !K || !floor_in || !bounds || !xbuf ||
      !cursor || !overflow || (rmap && (!K3 || k3 < k || k3 > 2048)))
    return fail(ASL_ERR_INVALID, "keys_extras: bad argument");
  ASL_TRY(ensure_device());
  return keys_extras(reinterpret_cast<const u64 *>(K), floor_in, (int64_t)W * nq, k,
                     reinterpret_cast<const u64 *>(bounds), nq, xcap, reinterpret_cast<u64 *>(xbuf),
                     reinterpret_cast<unsigned int *>(cursor), overflow, rmap,
                     reinterpret_cast<const u64 *>(K3), k3);      // nothing waits here
}

int asl_keys_merge_final(int32_t S, int32_t nq, int32_t kp, int32_t k, const int64_t *heads,
                         const int64_t *xbuf, int64_t xcap, const int64_t *prev_keys, const int32_t *need,
                         float *D, int64_t *I) {
  clear_error();
  if (S < 1 || nq < 0 || kp < 2 || k < 1 || k + 768 > 2048 || !heads || !prev_keys || !I)
    return fail(ASL_ERR_INVALID, "keys_merge_final: bad argument");
  ASL_TRY(ensure_device());
  return keys_merge(reinterpret_cast<const u64 *>(heads), S, nq, kp, k, reinterpret_cast<const u64 *>(xbuf),
                    xbuf ? xcap : 0, reinterpret_cast<const u64 *>(prev_keys), const_cast<int32_t *>(need),
                    nullptr, nullptr, I, D, 0);
}

}  // extern "C"
