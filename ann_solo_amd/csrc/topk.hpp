// topk.hpp -- exact streaming top-k for one workgroup (wave64, 256 threads).
//
// Hits are 64-bit keys: (order-preserving bits of the fp32 score) << 32 | (~id),
// so "larger key" == (score desc, id asc) -- the total order the oracle uses, which
// makes results independent of scan order, of sharding and of the flush schedule.
// A workgroup keeps CAP keys in LDS: every round each thread may append one key
// that beats the current k-th best; when fewer than NT free slots remain the
// buffer is bitonic-sorted, truncated to k and the threshold raised. Expected
// flushes for n streamed hits: ~log_{CAP/k}(n/k).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace asl {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, uint32_t id) {
  return ((u64)f2ord(score) << 32) | (u64)(0xFFFFFFFFu - id);
}
__device__ __forceinline__ float key_score(u64 k) { return ord2f((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_id(u64 k) { return 0xFFFFFFFFu - (uint32_t)k; }

// Sort n (power of two) keys in LDS, descending, with NT threads.
template <int NT>
__device__ __forceinline__ void bitonic_sort_desc(u64 *buf, int n, int tid) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n >> 1); t += NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const u64 a = buf[i], b = buf[l];
        const bool desc = (i & k) == 0;
        if (desc ? (a < b) : (a > b)) {
          buf[i] = b;
          buf[l] = a;
        }
      }
      __syncthreads();
    }
  }
}

// Compile-time-sized variant: all compare-exchanges of a step are loaded before any is
// resolved (CAP/(2*NT) independent LDS round trips in flight instead of one), which is
// what makes the flush cheap -- the run-time-sized loop above is latency-bound.
template <int NT, int CAP>
__device__ __forceinline__ void bitonic_sort_desc_ct(u64 *buf, int tid) {
  constexpr int PER = CAP / (2 * NT);
  static_assert(PER >= 1, "CAP must be at least 2*NT");
  // the stage loops must stay rolled: fully unrolled they are ~78 copies of the body and
  // thrash the instruction cache (measured 10x slower)
#pragma nounroll
  for (int k = 2; k <= CAP; k <<= 1) {
#pragma nounroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      u64 a[PER], b[PER];
      int ia[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int t = tid + u * NT;
        ia[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        a[u] = buf[ia[u]];
        b[u] = buf[ia[u] | j];
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const bool desc = (ia[u] & k) == 0;
        if (desc ? (a[u] < b[u]) : (a[u] > b[u])) {
          buf[ia[u]] = b[u];
          buf[ia[u] | j] = a[u];
        }
      }
      __syncthreads();
    }
  }
}

// Register-blocked bitonic sort (descending) of n = NT*P keys: every thread owns P keys
// in VGPRs (virtual positions tid*P + r). Compare-exchanges with stride < P are pure
// register work; strides >= P swap whole register blocks with thread tid ^ (stride/P)
// through a conflict-free transposed LDS image ([r][tid]), and only partner distances
// >= 64 threads (3 sub-steps of 4096) need a workgroup barrier -- inside a wave the
// LDS pipe already orders the write before the read. ~10x fewer instructions and ~25x
// fewer barriers than the one-compare-per-thread loop above.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}

template <int NT, int P>
__device__ __forceinline__ void block_sort_desc(u64 *buf, int tid, int n_out) {
  u64 x[P];
#pragma unroll
  for (int r = 0; r < P; ++r) x[r] = buf[r * NT + tid];   // any initial arrangement will do
  __syncthreads();
  // stages whose pairs never leave the thread
#pragma unroll
  for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int r = 0; r < P; ++r) {
        if ((r & j) == 0) {
          const int l = r | j;
          const bool desc = (k < P) ? ((r & k) == 0) : ((tid & 1) == 0);
          const bool lt = x[r] < x[l];
          const bool sw = (lt == desc);
          const u64 a = sw ? x[l] : x[r], b = sw ? x[r] : x[l];
          x[r] = a;
          x[l] = b;
        }
      }
    }
  }
#pragma nounroll
  for (int k = 2 * P; k <= NT * P; k <<= 1) {
    const bool desc = (tid & (k / P)) == 0;
#pragma nounroll
    for (int j = k >> 1; j >= P; j >>= 1) {
      const int jj = j / P;
      const bool keep_max = (((tid & jj) == 0) == desc);
      u64 y[P];
      if (jj >= 64) {          // partner in another wave: through the LDS image
#pragma unroll
        for (int r = 0; r < P; ++r) buf[r * NT + tid] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < P; ++r) y[r] = buf[r * NT + (tid ^ jj)];
        __syncthreads();
      } else if (jj == 1) {    // partner inside the wave: cross-lane moves, no LDS memory
#pragma unroll
        for (int r = 0; r < P; ++r) y[r] = dpp_u64<0xB1>(x[r]);   // quad_perm [1,0,3,2]
      } else if (jj == 2) {
#pragma unroll
        for (int r = 0; r < P; ++r) y[r] = dpp_u64<0x4E>(x[r]);   // quad_perm [2,3,0,1]
      } else {
#pragma unroll
        for (int r = 0; r < P; ++r) y[r] = __shfl_xor(x[r], jj, 64);
      }
#pragma unroll
      for (int r = 0; r < P; ++r) {
        const bool gt = y[r] > x[r];            // branch-free: equal keys are interchangeable
        x[r] = (gt == keep_max) ? y[r] : x[r];
      }
    }
#pragma unroll
    for (int j = P >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int r = 0; r < P; ++r) {
        if ((r & j) == 0) {
          const int l = r | j;
          const bool lt = x[r] < x[l];
          const bool sw = (lt == desc);
          const u64 a = sw ? x[l] : x[r], b = sw ? x[r] : x[l];
          x[r] = a;
          x[l] = b;
        }
      }
    }
  }
  __syncthreads();
  // natural order write-back of the leading n_out keys (position tid*P + r)
  if (tid * P < n_out) {
#pragma unroll
    for (int r = 0; r < P; ++r) buf[tid * P + r] = x[r];
  }
  __syncthreads();
}

template <int NT, int CAP = 0>
struct StreamTopK {
  u64 *buf;   // [cap] LDS
  int *ctl;   // LDS: ctl[0] = fill
  u64 *thr;   // LDS: current k-th best (0 = none yet)
  int cap, k;
  bool force_rt = false;  // measurement knob: use the run-time-sized sort
  // Deferred id resolution (the tiled PQ scan): keys appended since the last flush carry a
  // storage SLOT in their low word; the flush turns them into (score, ~id) keys by
  // gathering slot_ids[slot], so the hot loop never waits on an id load.
  const int32_t *slot_ids = nullptr;
  int conv_from = 0;
  bool emit_keys = false;   // finish(): I receives the packed keys (0 = empty) instead of ids

  __device__ __forceinline__ void init(u64 *b, int *c, u64 *t, int cap_, int k_, int tid) {
    buf = b;
    ctl = c;
    thr = t;
    cap = cap_;
    k = k_;
    for (int i = tid; i < cap; i += NT) buf[i] = 0ull;
    if (tid == 0) {
      ctl[0] = 0;
      *thr = 0ull;
    }
    __syncthreads();
  }

  // Called by ALL threads of the workgroup each round (key == 0: nothing to offer).
  // Barrier discipline: the caller's round is [compute key] -> push(); push() holds
  // one barrier before the appends and one after, so `fill` is stable when tested.
  __device__ __forceinline__ void push(u64 key, int tid) {
    __syncthreads();
    if (key > *thr) {
      const int s = atomicAdd(&ctl[0], 1);
      buf[s] = key;
    }
    __syncthreads();
    if (ctl[0] > cap - NT) flush(tid);
  }

  // Returns the fill level after the flush (identical in every thread).
  __device__ __forceinline__ int flush(int tid) {
    const int f = ctl[0];
    __syncthreads();
    if (slot_ids) {
      if constexpr (CAP > 0) {  // all gathers of a thread in flight together
        constexpr int PERC = CAP / NT;
        u64 kk[PERC];
        int32_t idv[PERC];
#pragma unroll
        for (int u = 0; u < PERC; ++u) {
          const int i = conv_from + tid + u * NT;
          kk[u] = i < f ? buf[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < PERC; ++u) {
          const int i = conv_from + tid + u * NT;
          idv[u] = (i < f && kk[u] != 0ull) ? slot_ids[(uint32_t)kk[u]] : -1;   // 0: an empty slot
        }
#pragma unroll
        for (int u = 0; u < PERC; ++u) {
          const int i = conv_from + tid + u * NT;
          if (i < f)
            buf[i] = idv[u] >= 0 ? ((kk[u] & 0xFFFFFFFF00000000ull) |
                                    (u64)(0xFFFFFFFFu - (uint32_t)idv[u])) : 0ull;
        }
      } else {
        for (int i = conv_from + tid; i < f; i += NT) {
          const u64 key = buf[i];
          const int32_t id = key != 0ull ? slot_ids[(uint32_t)key] : -1;
          buf[i] = id >= 0 ? ((key & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (uint32_t)id)) : 0ull;
        }
      }
    }
    for (int i = f + tid; i < cap; i += NT) buf[i] = 0ull;
    __syncthreads();
    if (CAP > 0 && !force_rt)
      block_sort_desc<NT, (CAP > 0 ? CAP / NT : 2)>(buf, tid, k);
    else
      bitonic_sort_desc<NT>(buf, cap, tid);
    const int nf = f < k ? f : k;
    conv_from = nf;
    if (tid == 0) {
      ctl[0] = nf;
      *thr = (nf == k) ? buf[k - 1] : 0ull;
    }
    __syncthreads();
    return nf;
  }

  // Final sort + write-out: D/I rows of length k (missing: -FLT_MAX / -1).
  __device__ __forceinline__ void finish(float *D, int64_t *I, int32_t *I32, int tid) {
    __syncthreads();
    flush(tid);
    const int f = ctl[0];
    for (int i = tid; i < k; i += NT) {
      const u64 key = buf[i];
      const bool ok = i < f;
      if (D) D[i] = ok ? key_score(key) : -3.402823466e+38f;
      if (I) I[i] = emit_keys ? (ok ? (int64_t)key : 0) : (ok ? (int64_t)key_id(key) : -1);
      if (I32) I32[i] = ok ? (int32_t)key_id(key) : -1;
    }
  }
};

// LDS capacity (keys) used for a given k.
inline int topk_cap_for(int k) {
  int cap = 1024;
  while (cap < 4 * k) cap <<= 1;
  return cap;
}

}  // namespace asl
