// ivf_kernels.hip -- device kernels of the hand-written IVF index (replaces the
// FAISS IndexFlatIP / IndexIVFFlat calls at
// /root/reference/src/ann_solo/spectral_library.py:167-181,443-445 and adds the
// IVF-PQ the north star asks for).
//
//   row_select_kernel exact top-k of a SHORT score row held in registers (the coarse
//                     top-nprobe of nlist): one histogram pass, one small sort
//   row_topk_kernel   exact streaming top-k of one score row per workgroup (IndexFlatIP,
//                     long rows, and IVF-Flat through a probed-list bitmap mask)
//   pq_scan_kernel    per query: per-query ADC look-up table built in LDS, the
//                     packed PQ codes of the probed lists streamed from HBM
//                     (32 B/vector, coalesced 16-B loads), fused exact top-k
//   topk_merge_hist_kernel / topk_merge_kernel  merge of per-shard partial top-k lists
//                     ((score, id) pairs or packed 64-bit keys), histogram-threshold selection
//   + small helpers (bitmap, argmax, residual/encode, gathers, L2 assignment)
#include <algorithm>

#include "common.hpp"
#include "ivf_kernels.hpp"
#include "hist_topk.hpp"
#include "topk.hpp"

namespace asl {

// ------------------------------------------------------------------ row top-k
// scores: [rows, ld]; hit id of column c is ids ? ids[c] : id_base + c.
// Mask (IVF-Flat): column c is visible to row r iff bit vlist[c] of bitmap[r] is set.
__global__ __launch_bounds__(TK_NT) void row_topk_kernel(
    const float *__restrict__ scores, int64_t ld, int n, int k, int cap,
    const int32_t *__restrict__ ids, int32_t id_base, const int32_t *__restrict__ vlist,
    const uint32_t *__restrict__ bitmap, int bitmap_words, float *__restrict__ D,
    int64_t *__restrict__ I64, int32_t *__restrict__ I32, int64_t out_ld,
    const u64 *__restrict__ upper_in, u64 *__restrict__ upper_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);
  u64 *thr = buf + cap;
  int *ctl = reinterpret_cast<int *>(thr + 1);
  const int tid = threadIdx.x;
  const int r = blockIdx.x;
  StreamTopK<TK_NT> tk;
  tk.init(buf, ctl, thr, cap, k, tid);
  const float *row = scores + (size_t)r * ld;
  const uint32_t *bm = bitmap ? bitmap + (size_t)r * bitmap_words : nullptr;
  // bounded pass (k beyond the LDS top-k, index.hip: search_large_k): only keys strictly below
  // the row's bound take part -- the keys at or above it were emitted by the earlier passes
  const u64 ub = upper_in ? upper_in[r] : ~0ull;
  for (int base = 0; base < n; base += TK_NT) {
    const int c = base + tid;
    u64 key = 0ull;
    if (c < n) {
      bool vis = true;
      if (bm) {
        const int l = vlist[c];
        vis = (bm[l >> 5] >> (l & 31)) & 1u;
      }
      if (vis) key = make_key(row[c], (uint32_t)(ids ? ids[c] : id_base + c));
      if (key >= ub) key = 0ull;
    }
    tk.push(key, tid);
  }
  tk.finish(D ? D + (size_t)r * out_ld : nullptr, I64 ? I64 + (size_t)r * out_ld : nullptr,
            I32 ? I32 + (size_t)r * out_ld : nullptr, tid);
  // the next pass's bound: the smallest key of a FULL row (0 = the row is exhausted: nothing passes)
  if (upper_out && tid == 0) upper_out[r] = ctl[0] >= k ? buf[k - 1] : 0ull;
}

// Short rows (n <= 4096, k <= 256; the coarse quantiser's top-nprobe of nlist): the whole row
// sits in registers, one histogram pass over [row min, row max] finds the bucket of the k-th
// score, the <= RS_SEL keys at or above it are compacted and sorted once. Rows whose
// threshold bucket is overfull (mass ties: an all-zero query scores every centroid 0)
// take the streaming path below inside the same launch.
constexpr int SEL_VPT = 16, SEL_CAP = 512, SEL_NB = 512;

__global__ __launch_bounds__(TK_NT) void row_select_kernel(
    const float *__restrict__ scores, int64_t ld, int n, int k, int cap, float *__restrict__ D,
    int64_t *__restrict__ I64, int32_t *__restrict__ I32, int64_t out_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);              // max(SEL_CAP, cap) keys
  const int nkeys = cap > SEL_CAP ? cap : SEL_CAP;
  u64 *thr = buf + nkeys;
  int *ctl = reinterpret_cast<int *>(thr + 1);            // 4 ints (StreamTopK) + 8 for scans
  int *hist = ctl + 12;                                   // SEL_NB
  float *red = reinterpret_cast<float *>(hist + SEL_NB);  // 8 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = blockIdx.x;
  const float *row = scores + (size_t)r * ld;
  float v[SEL_VPT];
  float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
#pragma unroll
  for (int u = 0; u < SEL_VPT; ++u) {
    const int c = tid + u * TK_NT;
    v[u] = c < n ? row[c] : 0.0f;
    if (c < n) {
      lo = fminf(lo, v[u]);
      hi = fmaxf(hi, v[u]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if (lane == 0) {
    red[wave] = lo;
    red[4 + wave] = hi;
  }
  for (int i = tid; i < SEL_NB; i += TK_NT) hist[i] = 0;
  __syncthreads();
  lo = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  hi = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float scale = hi > lo ? (float)(SEL_NB - 1) / (hi - lo) : 0.0f;   // monotone in the score
  int bk[SEL_VPT];
#pragma unroll
  for (int u = 0; u < SEL_VPT; ++u) {
    const int c = tid + u * TK_NT;
    float t = (v[u] - lo) * scale;
    t = fminf(fmaxf(t, 0.0f), (float)(SEL_NB - 1));
    bk[u] = (int)t;
    if (c < n) atomicAdd(&hist[bk[u]], 1);
  }
  __syncthreads();
  // bstar = highest bucket with >= k values at or above it (thread t owns the two buckets
  // SEL_NB-1-2t, SEL_NB-2-2t)
  {
    const int h0 = hist[SEL_NB - 1 - 2 * tid], h1 = hist[SEL_NB - 2 - 2 * tid];
    int tot;
    const int above = block_excl_scan256(h0 + h1, ctl + 4, tid, tot);
    if (tid == 0) ctl[3] = 0;                 // fewer than k values in the row: keep everything
    __syncthreads();
    if (above < k && above + h0 + h1 >= k) ctl[3] = (above + h0 >= k) ? SEL_NB - 1 - 2 * tid : SEL_NB - 2 - 2 * tid;
    __syncthreads();
  }
  const int bstar = ctl[3];
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < SEL_VPT; ++u) cnt += (tid + u * TK_NT < n) && bk[u] >= bstar;
  int total;
  int pos = block_excl_scan256(cnt, ctl + 8, tid, total);
  if (total <= SEL_CAP) {    // block-uniform
    for (int i = total + tid; i < SEL_CAP; i += TK_NT) buf[i] = 0ull;
#pragma unroll
    for (int u = 0; u < SEL_VPT; ++u) {
      const int c = tid + u * TK_NT;
      if (c < n && bk[u] >= bstar) buf[pos++] = make_key(v[u], (uint32_t)c);
    }
    __syncthreads();
    if (total <= TK_NT)      // block-uniform; the usual case (k = nprobe <= 128): half the sort
      block_sort_desc<TK_NT, 1>(buf, tid, TK_NT);
    else
      block_sort_desc<TK_NT, SEL_CAP / TK_NT>(buf, tid, SEL_CAP);
    const int f = total < k ? total : k;
    for (int i = tid; i < k; i += TK_NT) {
      const u64 key = buf[i];
      const bool ok = i < f;
      if (D) D[(size_t)r * out_ld + i] = ok ? key_score(key) : -3.402823466e+38f;
      if (I64) I64[(size_t)r * out_ld + i] = ok ? (int64_t)key_id(key) : -1;
      if (I32) I32[(size_t)r * out_ld + i] = ok ? (int32_t)key_id(key) : -1;
    }
    return;
  }
  __syncthreads();
  StreamTopK<TK_NT> tk;      // mass ties at the threshold: exact streaming selection
  tk.init(buf, ctl, thr, cap, k, tid);
  for (int base = 0; base < n; base += TK_NT) {   // re-read the row: keeps v[] out of scratch
    const int c = base + tid;
    u64 key = 0ull;
    if (c < n) {
      const float x = row[c];
      const float t = fminf(fmaxf((x - lo) * scale, 0.0f), (float)(SEL_NB - 1));
      if ((int)t >= bstar) key = make_key(x, (uint32_t)c);
    }
    tk.push(key, tid);
  }
  tk.finish(D ? D + (size_t)r * out_ld : nullptr, I64 ? I64 + (size_t)r * out_ld : nullptr,
            I32 ? I32 + (size_t)r * out_ld : nullptr, tid);
}

int row_topk(const float *scores, int64_t ld, int rows, int n, int k, const int32_t *ids,
             int32_t id_base, const int32_t *vlist, const uint32_t *bitmap, int bitmap_words,
             float *D, int64_t *I64, int32_t *I32, int64_t out_ld, const uint64_t *upper_in,
             uint64_t *upper_out) {
  if (rows <= 0) return ASL_OK;
  if (k <= 0 || k > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "top-k: k=%d outside 1..%d", k, TK_MAX_K);
  const int cap = topk_cap_for(k);
  if (n <= SEL_VPT * TK_NT && k <= 256 && !ids && id_base == 0 && !bitmap && !upper_in && !upper_out) {
    const size_t lds_sel = (size_t)std::max(cap, SEL_CAP) * 8 + 8 + 12 * 4 + SEL_NB * 4 + 8 * 4;
    hipLaunchKernelGGL(row_select_kernel, dim3(rows), dim3(TK_NT), lds_sel, stream(), scores, ld,
                       n, k, cap, D, I64, I32, out_ld);
    ASL_CHECK_LAUNCH();
    return ASL_OK;
  }
  const size_t lds = (size_t)cap * 8 + 16;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)row_topk_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(row_topk_kernel, dim3(rows), dim3(TK_NT), lds, stream(), scores, ld, n,
                     k, cap, ids, id_base, vlist, bitmap, bitmap_words, D, I64, I32, out_ld,
                     reinterpret_cast<const u64 *>(upper_in), reinterpret_cast<u64 *>(upper_out));
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ merge
// Ds/Is: [S, nq, k] partial lists -> [nq, k].
__global__ __launch_bounds__(TK_NT) void topk_merge_kernel(
    const float *__restrict__ Ds, const int64_t *__restrict__ Is, int S, int nq, int k,
    int cap, float *__restrict__ D, int64_t *__restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);
  u64 *thr = buf + cap;
  int *ctl = reinterpret_cast<int *>(thr + 1);
  const int tid = threadIdx.x, q = blockIdx.x;
  StreamTopK<TK_NT> tk;
  tk.init(buf, ctl, thr, cap, k, tid);
  const int total = S * k;
  for (int base = 0; base < total; base += TK_NT) {
    const int t = base + tid;
    u64 key = 0ull;
    if (t < total) {
      const int s = t / k, i = t - s * k;
      const size_t o = ((size_t)s * nq + q) * k + i;
      const int64_t id = Is[o];
      if (id >= 0) key = make_key(Ds[o], (uint32_t)id);
    }
    tk.push(key, tid);
  }
  tk.finish(D + (size_t)q * k, I + (size_t)q * k, nullptr, tid);
}

// Histogram-threshold variant (hist_topk.hpp): no sort while streaming, the survivors are
// sorted once. The S lists are visited in interleaved 64-entry chunks, so that (for the
// usual sorted partial lists) the best entries of every list come first and the
// threshold bucket rises after the first rounds; reads stay coalesced (256 B per wave).
// KEYS: Is holds packed (ord(score) << 32 | ~id) keys (0 = empty), Ds is unused.
template <int CAP, int PT, bool KEYS = false>
__global__ __launch_bounds__(HT_NT) void topk_merge_hist_kernel(
    const float *__restrict__ Ds, const int64_t *__restrict__ Is, int S, int nq, int k,
    float *__restrict__ D, int64_t *__restrict__ I, int unordered) {
  using TopK = HistTopK<CAP, HT_NT * PT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, q = blockIdx.x;
  TopK top;
  top.init(smem, k, nullptr, tid);
  const int kc = (k + 63) >> 6;          // 64-entry chunks per list
  const int total = kc * S * 64;
  for (int base = 0; base < total; base += HT_NT * PT) {
    top.begin_round();
    int appended = 0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const int v = base + u * HT_NT + tid;
      const int c = v >> 6, s = c % S, j = (c / S) * 64 + (v & 63);
      bool valid = v < total && j < k;
      float score = 0.0f;
      int64_t id = -1;
      uint32_t slot = 0;
      if (valid) {
        const size_t o = ((size_t)s * nq + q) * k + j;
        if (KEYS) {
          const u64 key = (u64)Is[o];
          valid = key != 0ull;
          score = key_score(key);
          slot = (uint32_t)key;
        } else {
          score = Ds[o];
          id = Is[o];
          valid = id >= 0;
          slot = 0xFFFFFFFFu - (uint32_t)id;
        }
      }
      const bool take = top.offer(valid, score, slot);
      appended += __popcll(__ballot(take));
    }
    top.end_round(appended);
  }
  if (unordered)   // exact top-k as a set: no sort (scratch = CAP keys behind the structure's LDS)
    top.finish_set(D ? D + (size_t)q * k : nullptr, I + (size_t)q * k, nullptr,
                   reinterpret_cast<u64 *>(smem + TopK::lds_bytes()));
  else
    top.finish(D ? D + (size_t)q * k : nullptr, I + (size_t)q * k, nullptr);
}

// merge of packed-key lists [S, nq, k] (asl_index_set_unordered mode 2)
int topk_merge_keys(const int64_t *Ks, int S, int nq, int k, float *D, int64_t *I, int unordered) {
  if (nq <= 0) return ASL_OK;
  if (k <= 0 || k + 256 + 512 > 2048)
    return fail(ASL_ERR_CAPACITY, "merge_keys: k=%d outside 1..1280", k);
  const size_t lds = HistTopK<2048, HT_NT * 2>::lds_bytes() + (unordered ? (size_t)2048 * 8 : 0);
  hipLaunchKernelGGL((topk_merge_hist_kernel<2048, 2, true>), dim3(nq), dim3(HT_NT), lds, stream(),
                     (const float *)nullptr, Ks, S, nq, k, D, I, unordered);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int topk_merge(const float *Ds, const int64_t *Is, int S, int nq, int k, float *D, int64_t *I) {
  if (nq <= 0) return ASL_OK;
  if (k <= 0 || k > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "merge: k=%d outside 1..%d", k, TK_MAX_K);
  if ((int64_t)S * k >= 2048 && k + 256 + 1024 <= 4096) {
    // CAP 2048 keeps the final sort at 8 keys per thread (~100 VGPRs, 4+ workgroups per CU);
    // CAP 4096 needs 16 (240 VGPRs, one workgroup per CU)
    if (k + 256 + 512 <= 2048)
      hipLaunchKernelGGL((topk_merge_hist_kernel<2048, 2>), dim3(nq), dim3(HT_NT),
                         (HistTopK<2048, HT_NT * 2>::lds_bytes()), stream(), Ds, Is, S, nq, k, D, I, 0);
    else
      hipLaunchKernelGGL((topk_merge_hist_kernel<4096, 4>), dim3(nq), dim3(HT_NT),
                         (HistTopK<4096, HT_NT * 4>::lds_bytes()), stream(), Ds, Is, S, nq, k, D, I, 0);
    ASL_CHECK_LAUNCH();
    return ASL_OK;
  }
  const int cap = topk_cap_for(k);
  const size_t lds = (size_t)cap * 8 + 16;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)topk_merge_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(TK_NT), lds, stream(), Ds, Is, S, nq, k,
                     cap, D, I);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ bitmap of probed lists
__global__ void probe_bitmap_kernel(const int32_t *__restrict__ coarse_I, int nq, int nprobe,
                                    uint32_t *__restrict__ bitmap, int words) {
  const int q = blockIdx.x;
  for (int p = threadIdx.x; p < nprobe; p += blockDim.x) {
    const int l = coarse_I[(size_t)q * nprobe + p];
    if (l >= 0) atomicOr(&bitmap[(size_t)q * words + (l >> 5)], 1u << (l & 31));
  }
}

int probe_bitmap(const int32_t *coarse_I, int nq, int nprobe, uint32_t *bitmap, int words) {
  HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)nq * words * sizeof(uint32_t), stream()));
  hipLaunchKernelGGL(probe_bitmap_kernel, dim3(nq), dim3(128), 0, stream(), coarse_I, nq,
                     nprobe, bitmap, words);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ row arg-max (ties: lowest col)
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *__restrict__ scores,
                                                         int64_t ld, int n,
                                                         int32_t *__restrict__ out) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float *row = scores + (size_t)r * ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < n; c += 256) {
    const float v = row[c];
    if (v > bv || (v == bv && c < bi)) {
      bv = v;
      bi = c;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    s_v[tid >> 6] = bv;
    s_i[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
        bv = s_v[w];
        bi = s_i[w];
      }
    out[r] = bi == 0x7fffffff ? 0 : bi;
  }
}

int row_argmax(const float *scores, int64_t ld, int rows, int n, int32_t *out) {
  if (rows <= 0) return ASL_OK;
  hipLaunchKernelGGL(row_argmax_kernel, dim3(rows), dim3(256), 0, stream(), scores, ld, n, out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ gathers
__global__ void gather_rows_f32_kernel(const float *__restrict__ src, int64_t ld_src,
                                       const int64_t *__restrict__ rows, int64_t n, int d,
                                       float *__restrict__ dst, int64_t ld_dst) {
  const int64_t i = block_linear();
  if (i >= n) return;
  const int64_t r = rows ? rows[i] : i;
  for (int j = threadIdx.x; j < d; j += blockDim.x)
    dst[i * ld_dst + j] = src[r * ld_src + j];
}
int gather_rows_f32(const float *src, int64_t ld_src, const int64_t *rows, int64_t n, int d,
                    float *dst, int64_t ld_dst) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(gather_rows_f32_kernel, grid_2d(n), dim3(256), 0, stream(), src,
                     ld_src, rows, n, d, dst, ld_dst);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

__global__ void gather_rows_u8_kernel(const uint8_t *__restrict__ src,
                                      const int32_t *__restrict__ rows, int64_t n, int m,
                                      uint8_t *__restrict__ dst) {
  const int64_t t = block_linear() * blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const int64_t i = t / m;
  const int j = (int)(t - i * m);
  dst[t] = src[(int64_t)rows[i] * m + j];
}
int gather_rows_u8(const uint8_t *src, const int32_t *rows, int64_t n, int m, uint8_t *dst) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(gather_rows_u8_kernel, grid_2d(cdiv(n * m, 256)), dim3(256), 0,
                     stream(), src, rows, n, m, dst);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ k-means update
// One workgroup per cluster: members (ascending point order) are summed in fp32 in
// that order, then divided by the count -- the oracle's loop, bit for bit.
__global__ __launch_bounds__(256) void centroid_update_kernel(
    const float *__restrict__ x, int64_t ld, int d, const int32_t *__restrict__ order,
    const int32_t *__restrict__ offsets, float *__restrict__ centroids) {
  const int c = blockIdx.x;
  const int b = offsets[c], e = offsets[c + 1];
  if (e == b) return;  // empty: left for the host-side split
  const float cnt = (float)(e - b);
  for (int j = threadIdx.x; j < d; j += 256) {
    float s = 0.0f;
    for (int t = b; t < e; ++t) s += x[(size_t)order[t] * ld + j];
    centroids[(size_t)c * d + j] = s / cnt;
  }
}
int centroid_update(const float *x, int64_t ld, int d, int k, const int32_t *order,
                    const int32_t *offsets, float *centroids) {
  hipLaunchKernelGGL(centroid_update_kernel, dim3(k), dim3(256), 0, stream(), x, ld, d, order,
                     offsets, centroids);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// Spherical k-means (FAISS cp.spherical for inner-product indexes): rows with a non-zero
// norm are rescaled to unit L2 length. Canonical chain: ascending fmaf from +0, sqrtf, IEEE
// divide (the encoder's norm) -- one lane walks the chain, the block divides.
__global__ __launch_bounds__(256) void renorm_rows_kernel(float *__restrict__ c, int d) {
  __shared__ float s_nrm;
  float *row = c + (size_t)blockIdx.x * d;
  if (threadIdx.x == 0) {
    float acc = 0.0f;
    for (int j = 0; j < d; ++j) acc = __builtin_fmaf(row[j], row[j], acc);
    s_nrm = acc > 0.0f ? __builtin_sqrtf(acc) : 0.0f;
  }
  __syncthreads();
  const float nrm = s_nrm;
  if (nrm > 0.0f)
    for (int j = threadIdx.x; j < d; j += 256) row[j] = row[j] / nrm;
}
int renorm_rows(float *c, int k, int d) {
  if (k <= 0) return ASL_OK;
  hipLaunchKernelGGL(renorm_rows_kernel, dim3(k), dim3(256), 0, stream(), c, d);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ L2 assignment (PQ)
// x: [n, ld] (sub-vector view, dsub <= 32), cb: [ksub, dsub]. Chain
// acc = fmaf(x-c, x-c, acc) ascending, arg-min with lowest-index ties.
__global__ __launch_bounds__(256) void l2_assign_kernel(const float *__restrict__ x,
                                                        int64_t ld, int64_t n, int dsub,
                                                        const float *__restrict__ cb, int ksub,
                                                        int32_t *__restrict__ assign) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_cb = reinterpret_cast<float *>(smem);
  for (int i = threadIdx.x; i < ksub * dsub; i += 256) s_cb[i] = cb[i];
  __syncthreads();
  const int64_t i = block_linear() * 256 + threadIdx.x;
  if (i >= n) return;
  float bs = INFINITY;
  int best = 0;
  if (dsub <= PQ_MAX_DSUB) {
    float xv[PQ_MAX_DSUB];
#pragma unroll
    for (int t = 0; t < PQ_MAX_DSUB; ++t) xv[t] = t < dsub ? x[i * ld + t] : 0.0f;
    for (int c = 0; c < ksub; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int t = 0; t < PQ_MAX_DSUB; ++t)
        if (t < dsub) {
          const float df = xv[t] - s_cb[c * dsub + t];
          acc = __builtin_fmaf(df, df, acc);
        }
      if (acc < bs) {
        bs = acc;
        best = c;
      }
    }
  } else {  // wide sub-vectors: operands re-read through L1 (slow path, same arithmetic)
    for (int c = 0; c < ksub; ++c) {
      float acc = 0.0f;
      for (int t = 0; t < dsub; ++t) {
        const float df = x[i * ld + t] - s_cb[c * dsub + t];
        acc = __builtin_fmaf(df, df, acc);
      }
      if (acc < bs) {
        bs = acc;
        best = c;
      }
    }
  }
  assign[i] = best;
}
int l2_assign(const float *x, int64_t ld, int64_t n, int dsub, const float *cb, int ksub,
              int32_t *assign) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(l2_assign_kernel, grid_2d(cdiv(n, 256)), dim3(256),
                     (size_t)ksub * dsub * 4, stream(), x, ld, n, dsub, cb, ksub, assign);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// residual r = x - centroid[assign] (fp32 subtract), in place or to dst
__global__ void residual_kernel(const float *__restrict__ x, const int32_t *__restrict__ assign,
                                const float *__restrict__ centroids, int64_t n, int d,
                                float *__restrict__ dst) {
  const int64_t i = block_linear();
  if (i >= n) return;
  const float *c = centroids + (size_t)assign[i] * d;
  for (int j = threadIdx.x; j < d; j += blockDim.x)
    dst[i * d + j] = x[i * d + j] - c[j];
}
int residual(const float *x, const int32_t *assign, const float *centroids, int64_t n, int d,
             float *dst) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(residual_kernel, grid_2d(n), dim3(256), 0, stream(), x, assign,
                     centroids, n, d, dst);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// PQ encode: grid (vector tiles, m). code = argmin_c L2(residual_sub, cb[m][c]).
__global__ __launch_bounds__(256) void pq_encode_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ assign,
    const float *__restrict__ centroids, const float *__restrict__ codebooks, int64_t n, int d,
    int m, int ksub, int dsub, uint8_t *__restrict__ codes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_cb = reinterpret_cast<float *>(smem);
  const int mi = blockIdx.y;
  const float *cb = codebooks + (size_t)mi * ksub * dsub;
  for (int i = threadIdx.x; i < ksub * dsub; i += 256) s_cb[i] = cb[i];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *xi = x + (size_t)i * d + (size_t)mi * dsub;
  const float *ci = centroids + (size_t)assign[i] * d + (size_t)mi * dsub;
  float bs = INFINITY;
  int best = 0;
  if (dsub <= PQ_MAX_DSUB) {
    float rv[PQ_MAX_DSUB];
#pragma unroll
    for (int t = 0; t < PQ_MAX_DSUB; ++t) rv[t] = t < dsub ? xi[t] - ci[t] : 0.0f;
    for (int c = 0; c < ksub; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int t = 0; t < PQ_MAX_DSUB; ++t)
        if (t < dsub) {
          const float df = rv[t] - s_cb[c * dsub + t];
          acc = __builtin_fmaf(df, df, acc);
        }
      if (acc < bs) {
        bs = acc;
        best = c;
      }
    }
  } else {
    for (int c = 0; c < ksub; ++c) {
      float acc = 0.0f;
      for (int t = 0; t < dsub; ++t) {
        const float df = (xi[t] - ci[t]) - s_cb[c * dsub + t];
        acc = __builtin_fmaf(df, df, acc);
      }
      if (acc < bs) {
        bs = acc;
        best = c;
      }
    }
  }
  codes[(size_t)i * m + mi] = (uint8_t)best;
}
int pq_encode(const float *x, const int32_t *assign, const float *centroids,
              const float *codebooks, int64_t n, int d, int m, int ksub, int dsub,
              uint8_t *codes) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(pq_encode_kernel, dim3((unsigned)cdiv(n, 256), m), dim3(256),
                     (size_t)ksub * dsub * 4, stream(), x, assign, centroids, codebooks, n, d,
                     m, ksub, dsub, codes);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ------------------------------------------------------------------ PQ LUT + scan
// LUT[m][c] = ascending-t fmaf chain of q[m*dsub+t] * cb[m][c][t]; thread c owns code c.
__device__ __forceinline__ void build_lut_lds(const float *__restrict__ xq_row, int d,
                                              const float *__restrict__ codebooks, int m,
                                              int ksub, int dsub, float *s_q, float *s_lut,
                                              int tid) {
  for (int i = tid; i < d; i += TK_NT) s_q[i] = xq_row[i];
  __syncthreads();
  for (int c = tid; c < ksub; c += TK_NT) {
    for (int mi = 0; mi < m; ++mi) {
      const float *cb = codebooks + ((size_t)mi * ksub + c) * dsub;
      const float *qs = s_q + mi * dsub;
      float acc = 0.0f;
      for (int t = 0; t < dsub; ++t) acc = __builtin_fmaf(qs[t], cb[t], acc);
      s_lut[mi * ksub + c] = acc;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(TK_NT) void pq_lut_kernel(const float *__restrict__ xq, int d,
                                                       const float *__restrict__ codebooks,
                                                       int m, int ksub, int dsub,
                                                       float *__restrict__ lut_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_lut = reinterpret_cast<float *>(smem);
  float *s_q = s_lut + m * ksub;
  const int q = blockIdx.x;
  build_lut_lds(xq + (size_t)q * d, d, codebooks, m, ksub, dsub, s_q, s_lut, threadIdx.x);
  for (int i = threadIdx.x; i < m * ksub; i += TK_NT) lut_out[(size_t)q * m * ksub + i] = s_lut[i];
}
int pq_lut(const float *xq, int nq, int d, const float *codebooks, int m, int ksub, int dsub,
           float *lut_out) {
  if (nq <= 0) return ASL_OK;
  const size_t lds = ((size_t)m * ksub + d) * 4;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)pq_lut_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pq_lut_kernel, dim3(nq), dim3(TK_NT), lds, stream(), xq, d, codebooks, m,
                     ksub, dsub, lut_out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// Canonical ADC sum: p_j = sum_t v[j+16t]; 16->1 mirror tree
// (j,15-j)(j,7-j)(j,3-j)(0,1); + coarse.
template <int M>
__device__ __forceinline__ float adc_score(const float *__restrict__ s_lut, int ksub,
                                           const uint32_t *__restrict__ cw, float coarse) {
  float v[M];
#pragma unroll
  for (int w = 0; w < M / 4; ++w) {
    const uint32_t word = cw[w];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int mi = w * 4 + b;
      v[mi] = s_lut[mi * ksub + ((word >> (8 * b)) & 0xffu)];
    }
  }
  float p[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < M) {
      float a = v[j];
#pragma unroll
      for (int t = j + 16; t < M; t += 16) a = a + v[t];
      p[j] = a;
    } else {
      p[j] = 0.0f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = p[j] + p[15 - j];
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = p[j] + p[7 - j];
#pragma unroll
  for (int j = 0; j < 2; ++j) p[j] = p[j] + p[3 - j];
  return coarse + (p[0] + p[1]);
}

// One workgroup per query: LUT in LDS, stream the probed lists, fused top-k.
template <int M>
__global__ __launch_bounds__(TK_NT) void pq_scan_kernel(
    const float *__restrict__ xq, int d, const float *__restrict__ codebooks, int ksub,
    int dsub, const float *__restrict__ coarse_D, const int32_t *__restrict__ coarse_I,
    int nprobe, const int32_t *__restrict__ list_offsets, const int32_t *__restrict__ ids,
    const uint8_t *__restrict__ codes, int k, int cap, float *__restrict__ D,
    int64_t *__restrict__ I64, int32_t *__restrict__ I32, int64_t out_ld,
    const u64 *__restrict__ upper_in, u64 *__restrict__ upper_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);
  u64 *thr = buf + cap;
  int *ctl = reinterpret_cast<int *>(thr + 1);       // 2 ints
  float *s_lut = reinterpret_cast<float *>(thr + 2);
  float *s_q = s_lut + M * ksub;
  const int tid = threadIdx.x, q = blockIdx.x;
  build_lut_lds(xq + (size_t)q * d, d, codebooks, M, ksub, dsub, s_q, s_lut, tid);
  StreamTopK<TK_NT> tk;
  tk.init(buf, ctl, thr, cap, k, tid);
  const u64 ub = upper_in ? upper_in[q] : ~0ull;      // bounded pass, see row_topk_kernel
  for (int p = 0; p < nprobe; ++p) {
    const int l = coarse_I[(size_t)q * nprobe + p];
    if (l < 0) continue;  // uniform
    const float coarse = coarse_D[(size_t)q * nprobe + p];
    const int start = list_offsets[l], len = list_offsets[l + 1] - start;
    for (int base = 0; base < len; base += TK_NT) {
      const int i = base + tid;
      u64 key = 0ull;
      if (i < len) {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(codes + (size_t)(start + i) * M);
        key = make_key(adc_score<M>(s_lut, ksub, cw, coarse), (uint32_t)ids[start + i]);
        if (key >= ub) key = 0ull;
      }
      tk.push(key, tid);
    }
  }
  tk.finish(D ? D + (size_t)q * out_ld : nullptr, I64 ? I64 + (size_t)q * out_ld : nullptr,
            I32 ? I32 + (size_t)q * out_ld : nullptr, tid);
  if (upper_out && tid == 0) upper_out[q] = ctl[0] >= k ? buf[k - 1] : 0ull;
}

template <int M>
static int launch_pq_scan(const float *xq, int nq, int d, const float *codebooks, int ksub,
                          int dsub, const float *coarse_D, const int32_t *coarse_I, int nprobe,
                          const int32_t *list_offsets, const int32_t *ids, const uint8_t *codes,
                          int k, float *D, int64_t *I64, int32_t *I32, int64_t out_ld,
                          const uint64_t *upper_in, uint64_t *upper_out) {
  const int cap = topk_cap_for(k);
  const size_t lds = (size_t)cap * 8 + 16 + ((size_t)M * ksub + d) * 4;
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "pq scan: k=%d / m=%d do not fit LDS", k, M);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)pq_scan_kernel<M>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pq_scan_kernel<M>, dim3(nq), dim3(TK_NT), lds, stream(), xq, d, codebooks,
                     ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, cap,
                     D, I64, I32, out_ld > 0 ? out_ld : (int64_t)k, reinterpret_cast<const u64 *>(upper_in),
                     reinterpret_cast<u64 *>(upper_out));
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int pq_scan(const float *xq, int nq, int d, const float *codebooks, int m, int ksub, int dsub,
            const float *coarse_D, const int32_t *coarse_I, int nprobe,
            const int32_t *list_offsets, const int32_t *ids, const uint8_t *codes, int k,
            float *D, int64_t *I64, int32_t *I32, int64_t out_ld, const uint64_t *upper_in,
            uint64_t *upper_out) {
  if (nq <= 0) return ASL_OK;
  if (k <= 0 || k > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "pq scan: k=%d outside 1..%d", k, TK_MAX_K);
  switch (m) {
    case 4: return launch_pq_scan<4>(xq, nq, d, codebooks, ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, D, I64, I32, out_ld, upper_in, upper_out);
    case 8: return launch_pq_scan<8>(xq, nq, d, codebooks, ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, D, I64, I32, out_ld, upper_in, upper_out);
    case 16: return launch_pq_scan<16>(xq, nq, d, codebooks, ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, D, I64, I32, out_ld, upper_in, upper_out);
    case 32: return launch_pq_scan<32>(xq, nq, d, codebooks, ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, D, I64, I32, out_ld, upper_in, upper_out);
    case 64: return launch_pq_scan<64>(xq, nq, d, codebooks, ksub, dsub, coarse_D, coarse_I, nprobe, list_offsets, ids, codes, k, D, I64, I32, out_ld, upper_in, upper_out);
    default: return fail(ASL_ERR_INVALID, "pq scan: pq_m must be one of 4,8,16,32,64 (got %d)", m);
  }
}

// sum over queries of probed list lengths (algorithmic work of a scan launch)
__global__ void scanned_count_kernel(const int32_t *__restrict__ coarse_I, int64_t n,
                                     const int32_t *__restrict__ list_offsets,
                                     unsigned long long *__restrict__ out) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = coarse_I[i];
    if (l >= 0) acc += (unsigned long long)(list_offsets[l + 1] - list_offsets[l]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
int scanned_count(const int32_t *coarse_I, int64_t n, const int32_t *list_offsets,
                  unsigned long long *out_dev) {
  // accumulates into out_dev (the caller owns its zeroing)
  hipLaunchKernelGGL(scanned_count_kernel, dim3(256), dim3(256), 0, stream(), coarse_I, n,
                     list_offsets, out_dev);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl
