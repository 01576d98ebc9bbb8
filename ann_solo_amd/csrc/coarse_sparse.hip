// coarse_sparse.hip -- coarse quantiser scores for SPARSE queries: S[q, l] = <x_q, c_l> for every
// inverted list l (FAISS IndexFlatIP inside IndexIVF*.search, call site
// /root/reference/src/ann_solo/spectral_library.py:443-444), without the dense GEMM.
//
// A hashed spectrum has <= ~50 non-zero components of d = 800 (one per peak), so 15/16 of the
// dense product's multiply-adds have a zero factor. The inner product is canonically the
// ascending-k fp32 fmaf chain from +0; fmaf(0, c, acc) == acc for finite c, so
// the chain over the query's NON-ZERO components in ascending k has the same bits -- as the
// MFMA GEMM, as the oracle.
//
// Formulation (what sank the first sparse attempt -- one lane per (query, list) with per-lane
// loads of the (dimension, value) entries -- is avoided on both sides):
//   * a workgroup owns a TILE OF 32 LISTS, staged once in LDS as tile[dim][32] (128 bytes per
//     dimension, 100 KB for d = 800) from the transposed centroid copy Ct[dim][nlist], and
//     streams the whole query batch through it;
//   * a wave takes 4 queries at a time, one per 16-lane ROW; lane j of a row owns lists 2j and
//     2j + 1 of the tile (one ds_read_b64 per step, two accumulators);
//   * a query's entries live in the REGISTERS of its row: lane j holds entries j, j + 16, j +
//     32, j + 48, and step k takes (dimension, value) of entry k by DPP row_share -- a
//     broadcast inside the row that costs one v_mov (the dimension, for the LDS address) and
//     nothing for the value (a DPP operand of the v_fmac). Entries beyond a query's count are
//     (dimension 0, value +0): fmaf(+0, c, acc) leaves acc unchanged, bit for bit.
// Per step and wave: 1 v_mov_dpp + 1 address add + 1 ds_read_b64 + 2 v_fmac_dpp for 4 queries x
// 32 lists -- 16 x fewer multiply-adds than the GEMM, and LDS-bound (~0.1 ms for 16 384 queries x
// 4 096 lists, against 0.86 ms for the dense MFMA GEMM at 125 TFLOP/s).
// Queries with more than 64 non-zeros (none among processed spectra: max_peaks_used = 50) are
// walked entry by entry from the dense row instead.
#include "common.hpp"
#include "ivf_kernels.hpp"

namespace asl {

constexpr int CS_TL = 32;          // lists per tile
constexpr int CS_NW = 16;          // waves per workgroup (one workgroup per CU: the tile fills LDS)
constexpr int CS_CAP = 64;         // entries a query keeps in registers

// ---- the non-zero components of every query row, ascending: ent[q][CS_CAP] = (dim, value bits),
// cnt[q] = their number (or -1 - count when there are more than CS_CAP). One wave per row.
__global__ __launch_bounds__(256) void list_nonzeros_kernel(const float *__restrict__ xq, int nq, int d,
                                                            uint2 *__restrict__ ent,
                                                            int32_t *__restrict__ cnt,
                                                            int *__restrict__ n_over) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  const float *row = xq + (size_t)q * d;
  uint2 *out = ent + (size_t)q * CS_CAP;
  int base = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const float x = j < d ? row[j] : 0.0f;
    const unsigned long long m = __ballot(x != 0.0f);
    if (x != 0.0f) {
      const int t = base + __popcll(m & ((1ull << lane) - 1ull));
      if (t < CS_CAP) out[t] = make_uint2((uint32_t)j, __float_as_uint(x));
    }
    base += __popcll(m);
  }
  for (int t = base + lane; t < CS_CAP; t += 64) out[t] = make_uint2(0u, 0u);     // (dim 0, +0.0)
  if (lane == 0) {
    cnt[q] = base <= CS_CAP ? base : -1 - base;
    if (base > CS_CAP) atomicAdd(n_over, 1);     // too many such rows: the dense GEMM takes the batch
  }
}

template <int CTRL>
__device__ __forceinline__ uint32_t cs_row_share(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// one step: entry K (0..15) of the register set (ed, ev) against the tile
#define CS_STEP(K, ED, EV)                                                                     \
  {                                                                                            \
    const uint32_t dm = cs_row_share<0x150 + (K)>(ED);                                         \
    const float vv = __uint_as_float(cs_row_share<0x150 + (K)>(EV));                           \
    const float2 c = *reinterpret_cast<const float2 *>(tile_lane + (size_t)dm * (CS_TL * 4));  \
    acc0 = __builtin_fmaf(vv, c.x, acc0);                                                      \
    acc1 = __builtin_fmaf(vv, c.y, acc1);                                                      \
  }
#define CS_BLOCK16(ED, EV)                                                                    \
  CS_STEP(0, ED, EV) CS_STEP(1, ED, EV) CS_STEP(2, ED, EV) CS_STEP(3, ED, EV) CS_STEP(4, ED, EV)   \
  CS_STEP(5, ED, EV) CS_STEP(6, ED, EV) CS_STEP(7, ED, EV) CS_STEP(8, ED, EV) CS_STEP(9, ED, EV)   \
  CS_STEP(10, ED, EV) CS_STEP(11, ED, EV) CS_STEP(12, ED, EV) CS_STEP(13, ED, EV)                  \
  CS_STEP(14, ED, EV) CS_STEP(15, ED, EV)

// grid (tiles, query parts); scores[q][l] row-major with leading dimension ld
__global__ __launch_bounds__(64 * CS_NW) void coarse_sparse_kernel(
    const float *__restrict__ xq, int nq, int d, const uint2 *__restrict__ ent,
    const int32_t *__restrict__ cnt, const float *__restrict__ Ct, int nlist,
    float *__restrict__ scores, int ld, const int *__restrict__ n_over, int over_max) {
  if (*n_over > over_max) return;       // a dense batch: gemm_nt_f32 (gated the other way) scores it
  extern __shared__ __attribute__((aligned(16))) char cs_smem[];
  float *tile = reinterpret_cast<float *>(cs_smem);       // [d][CS_TL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * CS_TL;
  // ---- the tile: row `dim` = Ct[dim][l0 .. l0 + 32) (zero beyond nlist)
  for (int i = tid; i < d * CS_TL; i += 64 * CS_NW) {
    const int dim = i >> 5, l = i & 31;
    tile[i] = l0 + l < nlist ? Ct[(size_t)dim * nlist + l0 + l] : 0.0f;
  }
  __syncthreads();
  const int row = lane >> 4, j = lane & 15;
  const char *tile_lane = reinterpret_cast<const char *>(tile) + j * 8;
  const int groups = (nq + 3) >> 2;
  const int gper = (groups + gridDim.y - 1) / gridDim.y;
  const int g_lo = blockIdx.y * gper, g_hi = min(groups, g_lo + gper);
  for (int g = g_lo + wave; g < g_hi; g += CS_NW) {
    const int q = g * 4 + row;
    const bool live = q < nq;
    const int c = live ? cnt[q] : 0;
    uint32_t ed[4] = {0u, 0u, 0u, 0u}, ev[4] = {0u, 0u, 0u, 0u};
    if (live) {
      const uint2 *e = ent + (size_t)q * CS_CAP + j;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint2 x = e[16 * u];
        ed[u] = x.x;
        ev[u] = x.y;
      }
    }
    float acc0 = 0.0f, acc1 = 0.0f;
    int cm = c < 0 ? 0 : c;                    // entries to walk (wave maximum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cm = max(cm, __shfl_xor(cm, o, 64));
    cm = __builtin_amdgcn_readfirstlane(cm);
    if (cm > 0) { CS_BLOCK16(ed[0], ev[0]) }
    if (cm > 16) { CS_BLOCK16(ed[1], ev[1]) }
    if (cm > 32) { CS_BLOCK16(ed[2], ev[2]) }
    if (cm > 48) { CS_BLOCK16(ed[3], ev[3]) }
    if (__ballot(c < 0)) {          // wave-uniform: a row with more than CS_CAP non-zeros -> its dense row
      if (c < 0) {
        acc0 = acc1 = 0.0f;
        const float *xr = xq + (size_t)q * d;
        for (int k = 0; k < d; ++k) {
          const float vv = xr[k];
          if (vv != 0.0f) {
            const float2 cc = *reinterpret_cast<const float2 *>(tile_lane + (size_t)k * (CS_TL * 4));
            acc0 = __builtin_fmaf(vv, cc.x, acc0);
            acc1 = __builtin_fmaf(vv, cc.y, acc1);
          }
        }
      }
    }
    if (live) {
      const int l = l0 + 2 * j;
      float *o = scores + (size_t)q * ld + l;
      if (l + 1 < nlist && (ld & 1) == 0) {
        *reinterpret_cast<float2 *>(o) = make_float2(acc0, acc1);
      } else {
        if (l < nlist) o[0] = acc0;
        if (l + 1 < nlist) o[1] = acc1;
      }
    }
  }
}

bool coarse_sparse_supported(int d, int nlist) {
  return d >= 1 && (size_t)d * CS_TL * 4 <= 150 * 1024 && nlist >= 1;
}

// out[dim][nlist] = in[nlist][dim]
__global__ void transpose_f32_kernel(const float *__restrict__ in, int rows, int cols,
                                     float *__restrict__ out) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
  for (int y = y0; y < 32; y += 8)
    if (by + y < rows && bx + x < cols) t[y][x] = in[(size_t)(by + y) * cols + bx + x];
  __syncthreads();
  for (int y = y0; y < 32; y += 8)
    if (bx + y < cols && by + x < rows) out[(size_t)(bx + y) * rows + by + x] = t[x][y];
}

int transpose_f32(const float *in, int rows, int cols, float *out) {
  if (rows <= 0 || cols <= 0) return ASL_OK;
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)),
                     dim3(256), 0, stream(), in, rows, cols, out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// scores[nq][ld] (first nlist columns) = xq[nq][d] . centroids^T, centroids given TRANSPOSED
// (Ct[d][nlist]); ent / cnt: scratch of nq * CS_CAP uint2 and nq ints; n_over: one int, left
// holding the number of rows with more than CS_CAP non-zeros. With more than over_max such rows
// the kernel leaves the batch alone (the caller's gated dense GEMM then runs instead): the
// decision is taken on the device, nothing waits.
int coarse_sparse(const float *xq, int nq, int d, const float *Ct, int nlist, uint2 *ent,
                  int32_t *cnt, int *n_over, int over_max, float *scores, int ld) {
  if (nq <= 0) return ASL_OK;
  HIP_TRY(hipMemsetAsync(n_over, 0, sizeof(int), stream()));
  hipLaunchKernelGGL(list_nonzeros_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, stream(), xq,
                     nq, d, ent, cnt, n_over);
  ASL_CHECK_LAUNCH();
  const int tiles = (int)cdiv(nlist, CS_TL);
  // one workgroup per CU (the tile fills LDS): cover the chip about twice over
  int parts = std::max(1, std::min((512 + tiles - 1) / tiles, (nq + 4 * CS_NW - 1) / (4 * CS_NW)));
  const size_t lds = (size_t)d * CS_TL * 4;
  HIP_TRY(hipFuncSetAttribute((const void *)coarse_sparse_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(coarse_sparse_kernel, dim3(tiles, parts), dim3(64 * CS_NW), lds, stream(), xq, nq, d,
                     ent, cnt, Ct, nlist, scores, ld, n_over, over_max);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int coarse_sparse_cap() { return CS_CAP; }

}  // namespace asl
