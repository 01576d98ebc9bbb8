// flat_bm_scan.hip -- exact IVF-Flat, BLOCK-MAJOR: the same per-dimension postings as
// flat_scan.hip's flat_inv_scan (same arithmetic, same bits), walked in the opposite order.
//
// flat_inv_scan gives a workgroup one QUERY and lets it walk the ~147 blocks of its probed
// lists: every (query, block) pair is two dependent rounds of scattered ~100-byte reads (segment
// table, then postings) that nobody else is reading at that moment -- 19 GB of 32-byte sectors
// per 16 384-query launch and long latency chains. But a batch probes every block ~880 times
// (16 384 x 128 probes / 4096 lists). Here a workgroup takes one BLOCK and a chunk of the queries
// that probe its list: the block's segment table sits in LDS, its ~124 KB of postings stay hot
// in L2 while the chunk streams through, and the only per-pair global read is the query's own
// list of non-zeros (one coalesced 384-byte load).
//
// A workgroup that owns a block cannot own a query's top-k, so the selection is split:
//   1. tau_q := the k-th best score among the query's P0 best-ranked lists (the query-major
//      kernel on those lists only; -inf if they hold fewer than k vectors). tau_q is a lower
//      bound of the final k-th best score, whatever the other lists hold.
//   2. block-major pass over ALL probed lists: every vector with score >= tau_q is appended to
//      the query's slab in global memory as a 64-bit key (score bits << 32 | ~id).
//   3. per query: sort the slab, keep the k best -- (score desc, id asc), the oracle's order.
// The emitted set contains the exact top-k by construction; a slab that overflows (BM_CAPQ)
// raises a flag and the caller re-runs the batch with the query-major kernel.
//
// Scores: per vector the ascending-dimension fp32 fmaf chain over the dimensions where both
// factors are non-zero -- exactly flat_inv_scan's (and the dense chain's, the GEMM's, the
// oracle's) bits.
#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"

namespace asl {

constexpr int BM_NW = 8, BM_NT = 64 * BM_NW;
constexpr int BM_QS = 64;       // non-zeros of a query the sparse rows hold (else: fall back)
constexpr int BM_QCH = 64;      // queries per work item (8 per wave)
constexpr int BM_U = 8;         // dimensions whose postings are loaded together

// ---- probe inversion: list -> the queries that probe it
__global__ void bm_count_kernel(const int32_t *__restrict__ coarse_I, int64_t n, int nlist,
                                int32_t *__restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int l = coarse_I[i];
  if (l >= 0 && l < nlist) atomicAdd(&cnt[l], 1);
}

// one workgroup: exclusive scans of the per-list query counts and of the per-list work items
// (blocks x query chunks); also resets the fill cursors
__global__ __launch_bounds__(1024) void bm_scan_kernel(const int32_t *__restrict__ cnt,
                                                       const int32_t *__restrict__ blk_offsets,
                                                       int nlist, int32_t *__restrict__ lq_off,
                                                       int32_t *__restrict__ item_off,
                                                       int32_t *__restrict__ cursor) {
  __shared__ int part[16];
  __shared__ int carry[2];
  const int tid = threadIdx.x;
  if (tid == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (int l0 = 0; l0 < nlist; l0 += 1024) {
    const int l = l0 + tid;
    const int c = l < nlist ? cnt[l] : 0;
    const int nb = l < nlist ? blk_offsets[l + 1] - blk_offsets[l] : 0;
    const int items = c > 0 ? nb * ((c + BM_QCH - 1) / BM_QCH) : 0;
    int tot_c, tot_i;
    const int pc = block_excl_scan<16>(c, part, tid, tot_c);
    __syncthreads();
    const int pi = block_excl_scan<16>(items, part, tid, tot_i);
    if (l < nlist) {
      lq_off[l] = carry[0] + pc;
      item_off[l] = carry[1] + pi;
      cursor[l] = 0;
    }
    __syncthreads();
    if (tid == 0) {
      carry[0] += tot_c;
      carry[1] += tot_i;
    }
    __syncthreads();
  }
  if (tid == 0) {
    lq_off[nlist] = carry[0];
    item_off[nlist] = carry[1];
  }
}

__global__ void bm_fill_kernel(const int32_t *__restrict__ coarse_I, int64_t n, int nprobe,
                               int nlist, const int32_t *__restrict__ lq_off,
                               int32_t *__restrict__ cursor, int32_t *__restrict__ lq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int l = coarse_I[i];
  if (l < 0 || l >= nlist) return;
  const int p = atomicAdd(&cursor[l], 1);
  lq[lq_off[l] + p] = (int32_t)(i / nprobe);
}

// ---- the queries' non-zero components, ascending: [nq][BM_QS] (dim u16, value f32) + count
__global__ __launch_bounds__(256) void bm_sparsify_kernel(const float *__restrict__ xq, int nq,
                                                          int d, uint16_t *__restrict__ qdim,
                                                          float *__restrict__ qval,
                                                          int32_t *__restrict__ qcnt,
                                                          int *__restrict__ status) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  const float *row = xq + (size_t)q * d;
  int base = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const float x = j < d ? row[j] : 0.0f;
    const unsigned long long m = __ballot(x != 0.0f);
    if (x != 0.0f) {
      const int t = base + __popcll(m & ((1ull << lane) - 1ull));
      if (t < BM_QS) {
        qdim[(size_t)q * BM_QS + t] = (uint16_t)j;
        qval[(size_t)q * BM_QS + t] = x;
      }
    }
    base += __popcll(m);
  }
  if (lane == 0) {
    qcnt[q] = base < BM_QS ? base : BM_QS;
    if (base > BM_QS) atomicOr(status, 1);   // a denser query: the caller falls back
  }
}

// first P0 columns of the probe lists (the lists are sorted by descending coarse score)
__global__ void bm_head_kernel(const int32_t *__restrict__ coarse_I, int nq, int nprobe, int p0,
                               int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nq * p0) return;
  const int q = (int)(i / p0), p = (int)(i - (int64_t)q * p0);
  out[i] = coarse_I[(size_t)q * nprobe + p];
}

// tau_q from the sorted top-k of the head lists
__global__ void bm_tau_kernel(const float *__restrict__ D0, const int32_t *__restrict__ I0, int nq,
                              int k, float *__restrict__ tau, int32_t *__restrict__ emit_cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  tau[q] = I0[(size_t)q * k + (k - 1)] >= 0 ? D0[(size_t)q * k + (k - 1)] : -INFINITY;
  emit_cnt[q] = 0;
}

// ---- the block-major pass
__global__ __launch_bounds__(BM_NT, 4) void flat_bm_kernel(
    int d, int nlist, const int32_t *__restrict__ list_offsets,
    const int32_t *__restrict__ blk_offsets, const uint32_t *__restrict__ seg_start,
    const uint32_t *__restrict__ seg_data, const int32_t *__restrict__ ids,
    const int32_t *__restrict__ lq_off, const int32_t *__restrict__ lq,
    const int32_t *__restrict__ item_off, const uint16_t *__restrict__ qdim,
    const float *__restrict__ qval, const int32_t *__restrict__ qcnt,
    const float *__restrict__ tau, int32_t *__restrict__ emit_cnt, u64 *__restrict__ slab,
    int capq, int *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem);                  // [BM_NW][FI_BLK]
  uint32_t *s_row = reinterpret_cast<uint32_t *>(s_acc + BM_NW * FI_BLK);   // [d + 1]
  __shared__ int s_item[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total_items = item_off[nlist];
  float *acc = s_acc + wave * FI_BLK;

  for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
    // item -> (list, block inside the list, query chunk): upper bound in item_off
    if (tid == 0) {
      int lo = 0, hi = nlist;            // last l with item_off[l] <= item
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_off[mid] <= item) lo = mid; else hi = mid;
      }
      const int l = lo;
      const int nb_l = blk_offsets[l + 1] - blk_offsets[l];
      const int r = item - item_off[l];
      s_item[0] = l;
      s_item[1] = r % nb_l;              // block of the list (consecutive items share a chunk)
      s_item[2] = r / nb_l;              // query chunk
    }
    __syncthreads();
    const int l = s_item[0], jb = s_item[1], chunk = s_item[2];
    const uint32_t blk = (uint32_t)(blk_offsets[l] + jb);
    const int pos_l = list_offsets[l];
    const int len_l = list_offsets[l + 1] - pos_l;
    const int pos0 = pos_l + jb * FI_BLK;
    const int nb = min(FI_BLK, len_l - jb * FI_BLK);
    const int q_lo = lq_off[l] + chunk * BM_QCH;
    const int q_hi = min(lq_off[l + 1], q_lo + BM_QCH);
    const uint32_t *erow = seg_start + (size_t)blk * d;
    for (int i = tid; i <= d; i += BM_NT) s_row[i] = erow[i];
    __syncthreads();

    for (int qi = q_lo + wave; qi < q_hi; qi += BM_NW) {
      const int q = lq[qi];
      const int K = qcnt[q];
      const float tq = tau[q];
      for (int o = lane; o < nb; o += 64) acc[o] = 0.0f;
      uint2 e = make_uint2(0u, 0u);
      float qv = 0.0f;
      if (lane < K) {
        const uint32_t dim = qdim[(size_t)q * BM_QS + lane];
        const uint32_t w0 = s_row[dim], w1 = s_row[dim + 1];
        e = make_uint2(w0, (2u * (w1 - w0)) / 3u);      // first word, postings
        qv = qval[(size_t)q * BM_QS + lane];
      }
      for (int j0 = 0; j0 < K; j0 += BM_U) {
        uint32_t st[BM_U], cn[BM_U];
        float qj[BM_U];
        uint32_t cmax = 0;
#pragma unroll
        for (int u = 0; u < BM_U; ++u) {
          const int j = j0 + u < K ? j0 + u : K - 1;
          st[u] = (uint32_t)__builtin_amdgcn_readlane((int)e.x, j);
          cn[u] = j0 + u < K ? (uint32_t)__builtin_amdgcn_readlane((int)e.y, j) : 0u;
          qj[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), j));
          cmax = cn[u] > cmax ? cn[u] : cmax;
        }
        if (cmax == 0) continue;
        uint32_t loc[BM_U];
        float val[BM_U];
#pragma unroll
        for (int u = 0; u < BM_U; ++u) {
          const bool on = (uint32_t)lane < cn[u];
          val[u] = on ? reinterpret_cast<const float *>(seg_data)[st[u] + lane] : 0.0f;
          loc[u] = on ? (uint32_t)reinterpret_cast<const uint16_t *>(seg_data)
                            [2 * (size_t)(st[u] + cn[u]) + lane]
                      : 0u;
        }
        // ascending dimensions, one after the other: a vector occurs at most once per dimension,
        // and the steps of one wave reach LDS in program order -> the canonical chain
#pragma unroll
        for (int u = 0; u < BM_U; ++u) {
          if ((uint32_t)lane < cn[u]) acc[loc[u]] = __builtin_fmaf(qj[u], val[u], acc[loc[u]]);
          if (cn[u] > 64u)      // wave-uniform
            for (uint32_t o = 64u + lane; o < cn[u]; o += 64) {
              const uint32_t lo2 =
                  reinterpret_cast<const uint16_t *>(seg_data)[2 * (size_t)(st[u] + cn[u]) + o];
              acc[lo2] = __builtin_fmaf(qj[u], reinterpret_cast<const float *>(seg_data)[st[u] + o],
                                        acc[lo2]);
            }
        }
      }
      // emission: everything that reaches tau_q (zero scores included, as in a dense scan)
      int cnt = 0;
      for (int v = lane; v - lane < nb; v += 64)
        cnt += __popcll(__ballot(v < nb && acc[v] >= tq));
      if (cnt > 0) {              // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&emit_cnt[q], cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + cnt > capq) {
          if (lane == 0) atomicOr(status, 2);
        } else {
          for (int v = lane; v - lane < nb; v += 64) {
            const bool pass = v < nb && acc[v] >= tq;
            const unsigned long long m = __ballot(pass);
            if (pass) {
              const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
              slab[(size_t)q * capq + slot] = make_key(acc[v], (uint32_t)ids[pos0 + v]);
            }
            base += __popcll(m);
          }
        }
      }
    }
    __syncthreads();     // s_row / s_item are rewritten by the next item
  }
}

// ---- per query: the k best of its slab, sorted (score desc, id asc)
__global__ __launch_bounds__(BM_NT) void bm_select_kernel(const u64 *__restrict__ slab,
                                                          const int32_t *__restrict__ emit_cnt,
                                                          int capq, int k, float *__restrict__ D,
                                                          int64_t *__restrict__ I64,
                                                          int32_t *__restrict__ I32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);
  const int q = blockIdx.x, tid = threadIdx.x;
  const int n = min(emit_cnt[q], capq);
  // sort size by what the query actually emitted (typically 2-3 k keys)
  const int N = n <= BM_NT * 4 ? BM_NT * 4 : (n <= BM_NT * 8 ? BM_NT * 8 : BM_NT * 16);
  for (int i = tid; i < N; i += BM_NT) buf[i] = i < n ? slab[(size_t)q * capq + i] : 0ull;
  __syncthreads();
  if (N == BM_NT * 4)
    block_sort_desc<BM_NT, 4>(buf, tid, k);
  else if (N == BM_NT * 8)
    block_sort_desc<BM_NT, 8>(buf, tid, k);
  else
    block_sort_desc<BM_NT, 16>(buf, tid, k);
  for (int i = tid; i < k; i += BM_NT) {
    const u64 key = i < N ? buf[i] : 0ull;
    const bool have = i < n && key != 0ull;
    const size_t o = (size_t)q * k + i;
    if (D) D[o] = have ? key_score(key) : -3.402823466e+38f;
    if (I64) I64[o] = have ? (int64_t)key_id(key) : -1;
    if (I32) I32[o] = have ? (int32_t)key_id(key) : -1;
  }
}

struct BmScratch {
  DevBuf<int32_t> cnt, lq_off, item_off, cursor, lq, qcnt, head, I0, emit_cnt;
  DevBuf<uint16_t> qdim;
  DevBuf<float> qval, D0, tau;
  DevBuf<u64> slab;
  DevBuf<int> status;
};

bool flat_bm_supported(int d, int k, int nprobe) {
  return flat_inv_supported(d, k, nprobe) && d < 65535 && k <= 2048;
}

// Returns ASL_OK and *fell_back = 1 when the batch must be redone by the query-major kernel
// (a query with more than BM_QS non-zeros, or a slab overflow); nothing has been written then.
int flat_bm_scan(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe, int nlist,
                 const int32_t *list_offsets, const int32_t *blk_offsets,
                 const uint32_t *seg_start, const uint32_t *seg_data, const int32_t *ids, int k,
                 float *D, int64_t *I64, int32_t *I32, int *fell_back) {
  *fell_back = 0;
  if (nq <= 0) return ASL_OK;
  static BmScratch &S = *new BmScratch();   // process lifetime (one device per process)
  const int p0 = std::min(nprobe, 16);
  const int capq = 8192;
  const int64_t np = (int64_t)nq * nprobe;
  ASL_TRY(S.cnt.reserve((size_t)nlist));
  ASL_TRY(S.lq_off.reserve((size_t)nlist + 1));
  ASL_TRY(S.item_off.reserve((size_t)nlist + 1));
  ASL_TRY(S.cursor.reserve((size_t)nlist));
  ASL_TRY(S.lq.reserve((size_t)np));
  ASL_TRY(S.qcnt.reserve((size_t)nq));
  ASL_TRY(S.qdim.reserve((size_t)nq * BM_QS));
  ASL_TRY(S.qval.reserve((size_t)nq * BM_QS));
  ASL_TRY(S.head.reserve((size_t)nq * p0));
  ASL_TRY(S.D0.reserve((size_t)nq * k));
  ASL_TRY(S.I0.reserve((size_t)nq * k));
  ASL_TRY(S.tau.reserve((size_t)nq));
  ASL_TRY(S.emit_cnt.reserve((size_t)nq));
  ASL_TRY(S.slab.reserve((size_t)nq * capq));
  ASL_TRY(S.status.reserve(1));
  hipStream_t st = stream();
  HIP_TRY(hipMemsetAsync(S.status.p, 0, sizeof(int), st));
  HIP_TRY(hipMemsetAsync(S.cnt.p, 0, sizeof(int32_t) * (size_t)nlist, st));
  // 1. the queries' sparse rows and the probe inversion
  hipLaunchKernelGGL(bm_sparsify_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, st, xq, nq, d,
                     S.qdim.p, S.qval.p, S.qcnt.p, S.status.p);
  ASL_CHECK_LAUNCH();
  hipLaunchKernelGGL(bm_count_kernel, dim3((unsigned)cdiv(np, 256)), dim3(256), 0, st, coarse_I, np,
                     nlist, S.cnt.p);
  ASL_CHECK_LAUNCH();
  hipLaunchKernelGGL(bm_scan_kernel, dim3(1), dim3(1024), 0, st, S.cnt.p, blk_offsets, nlist,
                     S.lq_off.p, S.item_off.p, S.cursor.p);
  ASL_CHECK_LAUNCH();
  hipLaunchKernelGGL(bm_fill_kernel, dim3((unsigned)cdiv(np, 256)), dim3(256), 0, st, coarse_I, np,
                     nprobe, nlist, S.lq_off.p, S.cursor.p, S.lq.p);
  ASL_CHECK_LAUNCH();
  // 2. tau from the head lists (query-major kernel, sorted rows)
  hipLaunchKernelGGL(bm_head_kernel, dim3((unsigned)cdiv((int64_t)nq * p0, 256)), dim3(256), 0, st,
                     coarse_I, nq, nprobe, p0, S.head.p);
  ASL_CHECK_LAUNCH();
  ASL_TRY(flat_inv_scan(xq, nq, d, S.head.p, p0, list_offsets, blk_offsets, seg_start, seg_data, ids,
                        k, S.D0.p, nullptr, S.I0.p, 0));
  hipLaunchKernelGGL(bm_tau_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, st, S.D0.p, S.I0.p,
                     nq, k, S.tau.p, S.emit_cnt.p);
  ASL_CHECK_LAUNCH();
  // 3. block-major pass
  const size_t lds = (size_t)BM_NW * FI_BLK * 4 + (size_t)(d + 1) * 4 + 16;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)flat_bm_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(flat_bm_kernel, dim3(256 * 4), dim3(BM_NT), lds, st, d, nlist, list_offsets,
                     blk_offsets, seg_start, seg_data, ids, S.lq_off.p, S.lq.p, S.item_off.p,
                     S.qdim.p, S.qval.p, S.qcnt.p, S.tau.p, S.emit_cnt.p, S.slab.p, capq, S.status.p);
  ASL_CHECK_LAUNCH();
  int h_status = 0;
  HIP_TRY(hipMemcpyAsync(&h_status, S.status.p, sizeof(int), hipMemcpyDeviceToHost, st));
  ASL_TRY(sync_stream());
  if (h_status) {
    *fell_back = 1;
    return ASL_OK;
  }
  // 4. per-query selection
  const size_t lds_sel = (size_t)capq * 8;
  if (capq != BM_NT * 16) return fail(ASL_ERR_INVALID, "flat_bm: slab capacity must be %d", BM_NT * 16);
  HIP_TRY(hipFuncSetAttribute((const void *)bm_select_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sel));
  hipLaunchKernelGGL(bm_select_kernel, dim3(nq), dim3(BM_NT), lds_sel, st, S.slab.p,
                     S.emit_cnt.p, capq, k, D, I64, I32);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl
