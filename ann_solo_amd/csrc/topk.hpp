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

template <int NT>
struct StreamTopK {
  u64 *buf;   // [cap] LDS
  int *ctl;   // LDS: ctl[0] = fill
  u64 *thr;   // LDS: current k-th best (0 = none yet)
  int cap, k;

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
    for (int i = f + tid; i < cap; i += NT) buf[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc<NT>(buf, cap, tid);
    const int nf = f < k ? f : k;
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
      if (I) I[i] = ok ? (int64_t)key_id(key) : -1;
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
