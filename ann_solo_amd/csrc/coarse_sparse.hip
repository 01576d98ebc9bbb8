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
// Per step and wave: 1 v_add_u32_dpp (address) + 1 ds_read_b64 + 2 v_fmac_f32_dpp for 4 queries
// x 32 lists, eight reads in flight -- 16 x fewer multiply-adds than the GEMM. The kernel is
// LDS-bound (13.4 GB of tile reads per 16 384 queries x 4 096 lists = 85 us at 256 B/clk/CU):
// 0.23 ms measured against 0.86 ms for the dense MFMA GEMM at 125 TFLOP/s.
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
// The dimension is stored as the byte offset of its tile row (dim * 128).
__global__ __launch_bounds__(256) void list_nonzeros_kernel(const float *__restrict__ xq, int nq, int d,
                                                            int64_t ldq, uint2 *__restrict__ ent,
                                                            int32_t *__restrict__ cnt,
                                                            int *__restrict__ n_over) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  const float *row = xq + (size_t)q * ldq;
  uint2 *out = ent + (size_t)q * CS_CAP;
  int base = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const float x = j < d ? row[j] : 0.0f;
    const unsigned long long m = __ballot(x != 0.0f);
    if (x != 0.0f) {
      const int t = base + __popcll(m & ((1ull << lane) - 1ull));
      if (t < CS_CAP) out[t] = make_uint2((uint32_t)j * (CS_TL * 4), __float_as_uint(x));   // byte offset of the tile row
    }
    base += __popcll(m);
  }
  for (int t = base + lane; t < CS_CAP; t += 64) out[t] = make_uint2(0u, 0u);     // (dim 0, +0.0)
  if (lane == 0) {
    cnt[q] = base <= CS_CAP ? base : -1 - base;
    if (base > CS_CAP) atomicAdd(n_over, 1);     // too many such rows: the dense GEMM takes the batch
  }
}

typedef float cs_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) cs_f2 cs_lds_f2;

// one step: entry K (0..15) of the register set (ED = dimension * 128, the byte offset of the
// tile row; EV = value bits). Three VALU instructions: the address add and the two multiply-adds
// take the row's lane K as a DPP operand (row_newbcast = row_share), no separate broadcast moves.
// (The DPP source registers are compiler-allocated and the hazard recogniser does not look inside
// inline asm: a VALU write of ED / EV -- e.g. the loop-carried copy at the back-edge -- needs two
// wait states before a DPP read. CS_NOP2 in front of the first address add and the first
// multiply-add of every block of eight provides them whatever the schedule; later steps of a
// block are separated from any earlier write by those instructions.)
#define CS_NOP2 "s_nop 1\n\t"
#define CS_ADDR_(PRE, K, ED, I)                                                          \
  asm(PRE "v_add_u32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf"     \
      : "=v"(ad[I])                                                                      \
      : "v"(ED), "v"(lane_off));
#define CS_ADDR(K, ED, I) CS_ADDR_("", K, ED, I)
#define CS_FMA(K, EV, I) CS_FMA_("", K, EV, I)
#define CS_FMA_(PRE, K, EV, I)                                                           \
  asm(PRE "v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf"    \
      : "+v"(acc0)                                                                   \
      : "v"(EV), "v"(cv[I].x));                                                      \
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf"    \
      : "+v"(acc1)                                                                   \
      : "v"(EV), "v"(cv[I].y));
// eight steps: all addresses, all LDS reads (in flight together), then the multiply-adds in
// entry order (the canonical chain)
#define CS_BLOCK8(K0, K1, K2, K3, K4, K5, K6, K7, ED, EV)                                       \
  {                                                                                             \
    uint32_t ad[8];                                                                             \
    cs_f2 cv[8];                                                                                \
    CS_ADDR_(CS_NOP2, K0, ED, 0) CS_ADDR(K1, ED, 1) CS_ADDR(K2, ED, 2) CS_ADDR(K3, ED, 3)                 \
    CS_ADDR(K4, ED, 4) CS_ADDR(K5, ED, 5) CS_ADDR(K6, ED, 6) CS_ADDR(K7, ED, 7)                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) cv[i_] = *(const cs_lds_f2 *)(uintptr_t)ad[i_]; \
    CS_FMA_(CS_NOP2, K0, EV, 0) CS_FMA(K1, EV, 1) CS_FMA(K2, EV, 2) CS_FMA(K3, EV, 3)                     \
    CS_FMA(K4, EV, 4) CS_FMA(K5, EV, 5) CS_FMA(K6, EV, 6) CS_FMA(K7, EV, 7)                     \
  }
#define CS_BLOCK8A(ED, EV) CS_BLOCK8(0, 1, 2, 3, 4, 5, 6, 7, ED, EV)
#define CS_BLOCK8B(ED, EV) CS_BLOCK8(8, 9, 10, 11, 12, 13, 14, 15, ED, EV)

// grid (tiles, query parts); scores[q][l] row-major with leading dimension ld
__global__ __launch_bounds__(64 * CS_NW) void coarse_sparse_kernel(
    const float *__restrict__ xq, int nq, int d, int64_t ldq, const uint2 *__restrict__ ent,
    const int32_t *__restrict__ cnt, const float *__restrict__ Ct, int nlist,
    float *__restrict__ scores, int ld, const int *__restrict__ n_over, int over_max) {
  if (*n_over > over_max) return;       // a dense batch: gemm_nt_f32 (gated the other way) scores it
  extern __shared__ __attribute__((aligned(16))) char cs_smem[];
  float *tile = reinterpret_cast<float *>(cs_smem);       // [d][CS_TL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * CS_TL;
  // ---- the tile: row `dim` = Ct[dim][l0 .. l0 + 32) (zero beyond nlist)
  if (l0 + CS_TL <= nlist && (nlist & 3) == 0) {     // 16-byte loads: 8 lanes per dimension
    for (int i = tid; i < d * (CS_TL / 4); i += 64 * CS_NW) {
      const int dim = i >> 3, l4 = (i & 7) * 4;
      reinterpret_cast<float4 *>(tile)[i] =
          *reinterpret_cast<const float4 *>(Ct + (size_t)dim * nlist + l0 + l4);
    }
  } else {
    for (int i = tid; i < d * CS_TL; i += 64 * CS_NW) {
      const int dim = i >> 5, l = i & 31;
      tile[i] = l0 + l < nlist ? Ct[(size_t)dim * nlist + l0 + l] : 0.0f;
    }
  }
  __syncthreads();
  const int row = lane >> 4, j = lane & 15;
  const char *tile_lane = reinterpret_cast<const char *>(tile) + j * 8;
  // (LDS byte address of the lane's two lists in row 0 of the tile)
  const uint32_t lane_off =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)cs_smem + (uint32_t)(j * 8);
  const int groups = (nq + 3) >> 2;
  const int gper = (groups + gridDim.y - 1) / gridDim.y;
  const int g_lo = blockIdx.y * gper, g_hi = min(groups, g_lo + gper);
  // a group's entries are requested one group ahead: the 64 steps of a group take ~0.3 us, an
  // entry load from L2 / HBM 1 - 2 us
  uint32_t n_ed[4] = {0u, 0u, 0u, 0u}, n_ev[4] = {0u, 0u, 0u, 0u};
  int n_c = 0;
  auto fetch = [&](int g) {
    const int q = g * 4 + row;
    if (g < g_hi && q < nq) {
      n_c = cnt[q];
      const uint2 *e = ent + (size_t)q * CS_CAP + j;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint2 x = e[16 * u];
        n_ed[u] = x.x;
        n_ev[u] = x.y;
      }
    } else {
      n_c = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) n_ed[u] = n_ev[u] = 0u;
    }
  };
  fetch(g_lo + wave);
  for (int g = g_lo + wave; g < g_hi; g += CS_NW) {
    const int q = g * 4 + row;
    const bool live = q < nq;
    const int c = n_c;
    const uint32_t ed[4] = {n_ed[0], n_ed[1], n_ed[2], n_ed[3]};
    const uint32_t ev[4] = {n_ev[0], n_ev[1], n_ev[2], n_ev[3]};
    fetch(g + CS_NW);
    float acc0 = 0.0f, acc1 = 0.0f;
    int cm = c < 0 ? 0 : c;                    // entries to walk (wave maximum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cm = max(cm, __shfl_xor(cm, o, 64));
    cm = __builtin_amdgcn_readfirstlane(cm);
    if (cm > 0) { CS_BLOCK8A(ed[0], ev[0]) }
    if (cm > 8) { CS_BLOCK8B(ed[0], ev[0]) }
    if (cm > 16) { CS_BLOCK8A(ed[1], ev[1]) }
    if (cm > 24) { CS_BLOCK8B(ed[1], ev[1]) }
    if (cm > 32) { CS_BLOCK8A(ed[2], ev[2]) }
    if (cm > 40) { CS_BLOCK8B(ed[2], ev[2]) }
    if (cm > 48) { CS_BLOCK8A(ed[3], ev[3]) }
    if (cm > 56) { CS_BLOCK8B(ed[3], ev[3]) }
    if (__ballot(c < 0)) {          // wave-uniform: a row with more than CS_CAP non-zeros -> its dense row
      if (c < 0) {
        acc0 = acc1 = 0.0f;
        const float *xr = xq + (size_t)q * ldq;
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

int list_nonzeros(const float *xq, int nq, int d, int64_t ldq, uint2 *ent, int32_t *cnt, int *n_over) {
  if (nq <= 0) return ASL_OK;
  hipLaunchKernelGGL(list_nonzeros_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, stream(), xq,
                     nq, d, ldq > 0 ? ldq : (int64_t)d, ent, cnt, n_over);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
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
                  int32_t *cnt, int *n_over, int over_max, float *scores, int ld, int64_t ldq) {
  if (ldq <= 0) ldq = d;
  if (nq <= 0) return ASL_OK;
  HIP_TRY(hipMemsetAsync(n_over, 0, sizeof(int), stream()));
  hipLaunchKernelGGL(list_nonzeros_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, stream(), xq,
                     nq, d, ldq, ent, cnt, n_over);
  ASL_CHECK_LAUNCH();
  const int tiles = (int)cdiv(nlist, CS_TL);
  // one workgroup per CU (the tile fills LDS): cover the chip about twice over
  int parts = std::max(1, std::min((512 + tiles - 1) / tiles, (nq + 4 * CS_NW - 1) / (4 * CS_NW)));
  const size_t lds = (size_t)d * CS_TL * 4;
  HIP_TRY(hipFuncSetAttribute((const void *)coarse_sparse_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(coarse_sparse_kernel, dim3(tiles, parts), dim3(64 * CS_NW), lds, stream(), xq, nq, d,
                     ldq, ent, cnt, Ct, nlist, scores, ld, n_over, over_max);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int coarse_sparse_cap() { return CS_CAP; }

}  // namespace asl
