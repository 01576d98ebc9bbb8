// refine.hip -- exact re-rank of an IVF-PQ short-list (FAISS IndexRefineFlat's role): the k'
// ids the ADC scan returns are rescored with the exact fp32 inner product against the stored
// vectors and the k best survive, (score desc, id asc).
//
// Stored vectors: hashed spectra have <= ~50 non-zeros of 800, so a row is kept sparse at a
// fixed stride of RF_STRIDE (dimension u16, value f32) entries, ascending dimension, add order
// (row = global id): 384 B per vector instead of 3 200. Score = acc = fmaf(q[dim], val, acc)
// over the row's entries -- the ascending-dimension chain restricted to the stored non-zeros,
// bit-identical to the dense chain, to IVF-Flat's scores and to the oracle (orc_refine).
#include "common.hpp"
#include "ivf_kernels.hpp"
#include "topk.hpp"

namespace asl {

constexpr int RF_STRIDE = 64;   // entries per stored row
constexpr int RF_NT = 256;

// dense rows [n, d] -> sparse rows appended at row0
__global__ __launch_bounds__(256) void refine_rows_kernel(const float *__restrict__ x, int64_t n,
                                                          int d, int64_t row0,
                                                          uint16_t *__restrict__ r_dim,
                                                          float *__restrict__ r_val,
                                                          uint8_t *__restrict__ r_cnt,
                                                          int *__restrict__ status) {
  const int64_t i = block_linear() * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float *row = x + (size_t)i * d;
  uint16_t *od = r_dim + (size_t)(row0 + i) * RF_STRIDE;
  float *ov = r_val + (size_t)(row0 + i) * RF_STRIDE;
  od[lane] = 0;
  ov[lane] = 0.0f;
  int base = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const float v = j < d ? row[j] : 0.0f;
    const unsigned long long m = __ballot(v != 0.0f);
    if (v != 0.0f) {
      const int t = base + __popcll(m & ((1ull << lane) - 1ull));
      if (t < RF_STRIDE) {
        od[t] = (uint16_t)j;
        ov[t] = v;
      }
    }
    base += __popcll(m);
  }
  if (lane == 0) {
    r_cnt[row0 + i] = (uint8_t)(base < RF_STRIDE ? base : RF_STRIDE);
    if (base > RF_STRIDE) atomicOr(status, 1);
  }
}

// one workgroup per query: dense query in LDS, a thread per candidate, then one sort
template <int P>
__global__ __launch_bounds__(RF_NT) void refine_kernel(
    const float *__restrict__ xq, int d, const int32_t *__restrict__ I_in,
    const int64_t *__restrict__ I_in64, int kp, const uint16_t *__restrict__ r_dim, const float *__restrict__ r_val,
    const uint8_t *__restrict__ r_cnt, int64_t n_rows, int k, float *__restrict__ D,
    int64_t *__restrict__ I64, int32_t *__restrict__ I32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *buf = reinterpret_cast<u64 *>(smem);                    // [RF_NT * P]
  float *s_q = reinterpret_cast<float *>(buf + RF_NT * P);     // [d]
  const int q = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < d; i += RF_NT) s_q[i] = xq[(size_t)q * d + i];
  __syncthreads();
  for (int c = tid; c < RF_NT * P; c += RF_NT) {
    u64 key = 0ull;
    const int64_t id = c >= kp ? -1
                       : (I_in ? (int64_t)I_in[(size_t)q * kp + c] : I_in64[(size_t)q * kp + c]);
    if (id >= 0 && id < n_rows) {
      const int cnt = r_cnt[id];
      const uint4 *pd = reinterpret_cast<const uint4 *>(r_dim + (size_t)id * RF_STRIDE);
      const float4 *pv = reinterpret_cast<const float4 *>(r_val + (size_t)id * RF_STRIDE);
      float acc = 0.0f;
      for (int t0 = 0; t0 < cnt; t0 += 8) {          // 8 entries: one 16-B load of dims, two of values
        const uint4 dd = pd[t0 >> 3];
        const float4 v0 = pv[t0 >> 2], v1 = pv[(t0 >> 2) + 1];
        const uint32_t dw[4] = {dd.x, dd.y, dd.z, dd.w};
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (t0 + u < cnt) {
            const uint32_t dim = (dw[u >> 1] >> (16 * (u & 1))) & 0xffffu;
            acc = __builtin_fmaf(s_q[dim], vv[u], acc);
          }
      }
      key = make_key(acc, (uint32_t)id);
    }
    buf[c] = key;
  }
  __syncthreads();
  block_sort_desc<RF_NT, P>(buf, tid, k);
  for (int i = tid; i < k; i += RF_NT) {
    const u64 key = i < RF_NT * P ? buf[i] : 0ull;
    const bool have = key != 0ull;
    const size_t o = (size_t)q * k + i;
    if (D) D[o] = have ? key_score(key) : -3.402823466e+38f;
    if (I64) I64[o] = have ? (int64_t)key_id(key) : -1;
    if (I32) I32[o] = have ? (int32_t)key_id(key) : -1;
  }
}

int refine_append_rows(const float *x, int64_t n, int d, int64_t row0, uint16_t *r_dim, float *r_val,
                       uint8_t *r_cnt, int *status) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(refine_rows_kernel, grid_2d(cdiv(n, 4)), dim3(256), 0, stream(), x, n, d,
                     row0, r_dim, r_val, r_cnt, status);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int refine_stride() { return RF_STRIDE; }

template <int P>
static int launch_refine(const float *xq, int nq, int d, const int32_t *I_in,
                         const int64_t *I_in64, int kp, const uint16_t *r_dim, const float *r_val, const uint8_t *r_cnt,
                         int64_t n_rows, int k, float *D, int64_t *I64, int32_t *I32) {
  const size_t lds = (size_t)RF_NT * P * 8 + (size_t)((d + 3) & ~3) * 4;
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "refine: d=%d does not fit LDS", d);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)refine_kernel<P>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(refine_kernel<P>, dim3(nq), dim3(RF_NT), lds, stream(), xq, d, I_in, I_in64, kp, r_dim,
                     r_val, r_cnt, n_rows, k, D, I64, I32);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// I_in / I_in64 [nq, kp] ids (one of the two; any order, -1 = empty) -> the k best by exact
// score, sorted
int refine_topk(const float *xq, int nq, int d, const int32_t *I_in, const int64_t *I_in64, int kp,
                const uint16_t *r_dim, const float *r_val, const uint8_t *r_cnt, int64_t n_rows, int k, float *D,
                int64_t *I64, int32_t *I32) {
  if (nq <= 0) return ASL_OK;
  if (k > kp) return fail(ASL_ERR_INVALID, "refine: k=%d exceeds the short-list %d", k, kp);
  if (kp <= RF_NT * 2) return launch_refine<2>(xq, nq, d, I_in, I_in64, kp, r_dim, r_val, r_cnt, n_rows, k, D, I64, I32);
  if (kp <= RF_NT * 4) return launch_refine<4>(xq, nq, d, I_in, I_in64, kp, r_dim, r_val, r_cnt, n_rows, k, D, I64, I32);
  if (kp <= RF_NT * 8) return launch_refine<8>(xq, nq, d, I_in, I_in64, kp, r_dim, r_val, r_cnt, n_rows, k, D, I64, I32);
  return fail(ASL_ERR_CAPACITY, "refine: short-list %d exceeds %d", kp, RF_NT * 8);
}

}  // namespace asl
