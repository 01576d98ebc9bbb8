// pq_scan_v2.hip -- IVF-PQ asymmetric-distance scan, m = 32 sub-quantisers of 8 bits,
// laid out for gfx950's LDS (the dominant kernel of the hot path; replaces the list scan
// inside FAISS' IndexIVF*.search, call site
// /root/reference/src/ann_solo/spectral_library.py:443-444).
//
// Why not "one lane per vector": 32 random LUT reads per vector hit random LDS banks
// (~3.5-way conflicts on ds_read_b32). Here lanes are SUB-QUANTISERS instead:
//
//   * LUT image in LDS: lut[c*32 + m] (row stride 128 B) -> bank == m. A 16-lane row
//     reads 16 different m for one code position, the neighbouring row the other 16:
//     every ds_read_b32 is bank-conflict-free BY CONSTRUCTION, whatever the codes are.
//   * codes are stored in 64-vector tiles (2 KiB): chunk[rho][m][16 B], the 16 bytes
//     being sub-quantiser m's codes of the row's 16 vectors, permuted b -> b ^ (m & 15).
//     Lane (rho, j) loads the chunks of m = j and m = j + 16 (two coalesced 16-B loads).
//   * register r of lane j then holds p_j(vector r ^ j); four DPP butterflies
//     (row_mirror, row_half_mirror, quad [3,2,1,0], quad [1,0,3,2]) reduce the 16x16
//     block with 15 v_add_dpp and NO selects, leaving vector j's sum in lane j -- the
//     canonical mirror tree of DESIGN.md, bit-identical to the oracle.
//   * ids are fetched only for lanes whose score clears the current k-th best.
//
// One workgroup (4 waves) per query; the probed lists are flattened into a tile
// table (LDS, 512 entries per chunk) so waves stay balanced; each round a wave takes
// V2_T tiles, appends survivors to the shared StreamTopK buffer with one barrier
// per 512 vectors. The per-query LUT is built in LDS with a rotated sub-quantiser
// order so its 64 KiB/query of stores are conflict-free too.
#include "common.hpp"
#include "ivf_kernels.hpp"
#include "pq_tile.hpp"
#include "topk.hpp"

namespace asl {

constexpr int V2_NT = 256;
constexpr int V2_T = 4;                      // tiles per wave per round
constexpr int V2_ROUND_TILES = 4 * V2_T;     // tiles per workgroup round
constexpr int V2_ROUND_VECS = V2_ROUND_TILES * 64;
constexpr int V2_CHUNK = 512;                // tile-table entries per chunk
constexpr int V2_M = 32, V2_KSUB = 256;

template <int CAP>
__global__ __launch_bounds__(V2_NT) void pq_scan_v2_kernel(
    const float *__restrict__ xq, int d, const float *__restrict__ codebooks, int dsub,
    const float *__restrict__ coarse_D, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ list_offsets, const int32_t *__restrict__ tile_offsets,
    const uint8_t *__restrict__ codes_tiled, const int32_t *__restrict__ ids_tiled, int k,
    float *__restrict__ D, int64_t *__restrict__ I64, int32_t *__restrict__ I32, int dbg) {
  constexpr int cap = CAP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *keys = reinterpret_cast<u64 *>(smem);
  u64 *thr_p = keys + cap;
  int *ctl = reinterpret_cast<int *>(thr_p + 1);  // ctl[0] = fill; ctl[2..9] = per-wave round counts
  int *s_wcnt = ctl + 2;                          // [2 parities][4 waves]
  float *s_lut = reinterpret_cast<float *>(thr_p + 8);  // 64-B control block
  TileEnt *table = reinterpret_cast<TileEnt *>(s_lut + V2_KSUB * V2_M);
  float *s_q = reinterpret_cast<float *>(table);  // aliases the table during the LUT build
  int *s_scan = reinterpret_cast<int *>(table);   // and the prefix scan scratch

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
  long long t_lut_g = 0;

  // ---- per-query LUT, rotated sub-quantiser order (conflict-free LDS stores)
  // (dbg bits are measurement knobs: 1 no appends, 2 no LUT build, 4 no ADC, 8 no code loads)
  if (!(dbg & 2)) {
    const long long tl0 = clock64();
    const float *xq_row = xq + (size_t)q * d;
    for (int i = tid; i < d; i += V2_NT) s_q[i] = xq_row[i];
    __syncthreads();
    const int c = tid;
    for (int s = 0; s < V2_M; ++s) {
      const int m = (s + lane) & (V2_M - 1);
      const float *cb = codebooks + ((size_t)m * V2_KSUB + c) * dsub;
      const float *qs = s_q + m * dsub;
      float acc = 0.0f;
      for (int t = 0; t < dsub; ++t) acc = __builtin_fmaf(qs[t], cb[t], acc);
      s_lut[c * V2_M + m] = acc;
    }
    __syncthreads();
    t_lut_g = clock64() - tl0;
  }

  // ---- my probe (thread p < nprobe), exclusive scan of tile counts
  int my_len = 0, my_tile0 = 0, my_nt = 0;
  float my_coarse = 0.0f;
  if (tid < nprobe) {
    const int l = coarse_I[(size_t)q * nprobe + tid];
    if (l >= 0) {
      my_len = list_offsets[l + 1] - list_offsets[l];
      my_tile0 = tile_offsets[l];
      my_nt = (my_len + 63) >> 6;
      my_coarse = coarse_D[(size_t)q * nprobe + tid];
    }
  }
  int incl = my_nt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int wave_base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = s_scan[w];
    if (w < wave) wave_base += t;
    total += t;
  }
  const int my_pre = wave_base + incl - my_nt;
  __syncthreads();

  StreamTopK<V2_NT, CAP> tk;
  tk.init(keys, ctl, thr_p, cap, k, tid);
  tk.force_rt = (dbg & 16) != 0;
  tk.slot_ids = ids_tiled;

  const char *lut_bytes = reinterpret_cast<const char *>(s_lut);
  const int rho = lane >> 4, j = lane & 15;
  const int ma = (rho & 1) ? j + 16 : j, mb = ma ^ 16;
  const uint32_t offA = (uint32_t)ma * 4u, offB = (uint32_t)mb * 4u;
  const uint32_t chunkA = (uint32_t)(rho * 512 + ma * 16), chunkB = (uint32_t)(rho * 512 + mb * 16);

  int fill = 0, parity = 0, n_flush = 0, n_app = 0;
  long long t_flush = 0;
  const long long t_start = clock64();
  float dbg_acc = 0.0f;
  for (int c0 = 0; c0 < total; c0 += V2_CHUNK) {
    // tile table of this chunk: every probe writes the entries of its own tiles
    {
      const int lo = max(my_pre, c0), hi = min(my_pre + my_nt, c0 + V2_CHUNK);
      for (int t = lo; t < hi; ++t) {
        const int local = t - my_pre;
        TileEnt e;
        e.tile = (uint32_t)(my_tile0 + local);
        e.coarse = my_coarse;
        e.nvalid = min(64, my_len - local * 64);
        e.pad = 0;
        table[t - c0] = e;
      }
    }
    __syncthreads();
    const int nent = min(V2_CHUNK, total - c0);
    const int nrounds = (nent + V2_ROUND_TILES - 1) / V2_ROUND_TILES;
    uint4 A[V2_T], B[V2_T], A2[V2_T], B2[V2_T];
    TileEnt ent[V2_T], ent2[V2_T];
    auto fetch = [&](int rr, uint4 *a, uint4 *b, TileEnt *e) {
#pragma unroll
      for (int u = 0; u < V2_T; ++u) {
        const int i = rr * V2_ROUND_TILES + wave * V2_T + u;
        e[u] = table[i < nent ? i : 0];
        if (i >= nent) e[u].nvalid = 0;
        const uint8_t *base = codes_tiled + (size_t)e[u].tile * 2048;
        if (!(dbg & 8)) {
          a[u] = *reinterpret_cast<const uint4 *>(base + chunkA);
          b[u] = *reinterpret_cast<const uint4 *>(base + chunkB);
        } else {
          a[u] = make_uint4(lane, rr, u, 1);
          b[u] = make_uint4(rr, lane, 2, u);
        }
      }
    };
    fetch(0, A, B, ent);
    for (int rr = 0; rr < nrounds; ++rr, parity ^= 1) {
      if (rr + 1 < nrounds) fetch(rr + 1, A2, B2, ent2);   // prefetch across the barrier
      const uint32_t thr_hi = (uint32_t)(*thr_p >> 32);
      int appended = 0;
#pragma unroll
      for (int u = 0; u < V2_T; ++u) {
        if (ent[u].nvalid > 0) {  // wave-uniform
          float score;
          if (!(dbg & 4))
            score = ent[u].coarse + tile_adc(lut_bytes, A[u], B[u], offA, offB);
          else
            score = __uint_as_float((A[u].x ^ B[u].y ^ A[u].z ^ B[u].w) & 0x3fffffffu);
          const uint32_t ob = f2ord(score);
          bool take = false;
          if (dbg & 1) {
            dbg_acc += score;
          } else if (lane < ent[u].nvalid && ob >= thr_hi) {
            // score >= current k-th best score: keep (score, storage slot); the flush
            // resolves ids and the exact (score desc, id asc) order
            const int s = atomicAdd(&ctl[0], 1);
            keys[s] = ((u64)ob << 32) | (u64)(ent[u].tile * 64u + (uint32_t)lane);
            take = true;
          }
          appended += __popcll(__ballot(take));
        }
      }
      // `fill` is tracked identically in every thread from per-wave counts that live in
      // a parity-double-buffered LDS slot: the flush decision cannot race with the next
      // round's appends of a faster wave.
      if (lane == 0) s_wcnt[parity * 4 + wave] = appended;
      __syncthreads();
      fill += s_wcnt[parity * 4] + s_wcnt[parity * 4 + 1] + s_wcnt[parity * 4 + 2] +
              s_wcnt[parity * 4 + 3];
      if (fill > cap - V2_ROUND_VECS) {
        const long long t0 = clock64();
        fill = tk.flush(tid);
        t_flush += clock64() - t0;
        ++n_flush;
      }
      n_app += appended;
#pragma unroll
      for (int u = 0; u < V2_T; ++u) {
        A[u] = A2[u];
        B[u] = B2[u];
        ent[u] = ent2[u];
      }
    }
    __syncthreads();
  }
  if ((dbg & 1) && dbg_acc == 12345.678f) keys[0] = 1;  // keeps the scores live
  const long long t_fin0 = clock64();
  tk.finish(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
            I32 ? I32 + (size_t)q * k : nullptr, tid);
  const long long t_end = clock64();
  if ((dbg & 32) && D) {  // measurement knob: report flushes / this wave's appends / tiles
    __syncthreads();
    if (tid == 0) {
      D[(size_t)q * k] = (float)n_flush;
      D[(size_t)q * k + 1] = (float)n_app;
      D[(size_t)q * k + 2] = (float)total;
      D[(size_t)q * k + 3] = (float)(t_end - t_start);
      D[(size_t)q * k + 4] = (float)t_flush;
      D[(size_t)q * k + 5] = (float)(t_end - t_fin0);
      D[(size_t)q * k + 6] = (float)t_lut_g;
    }
  }
}

// list-ordered codes [n,32] -> 64-vector tiles (see file header); dst_slot[i] = tile*64 + v
__global__ void tile_codes_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ ids,
                                  const int32_t *__restrict__ dst_slot, int64_t n,
                                  uint8_t *__restrict__ codes_tiled,
                                  int32_t *__restrict__ ids_tiled) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * V2_M) return;
  const int64_t i = t >> 5;
  const int m = (int)(t & 31);
  const int32_t slot = dst_slot[i];
  const int64_t tile = slot >> 6;
  const int v = slot & 63, rho = v >> 4, b = (v & 15) ^ (m & 15);
  codes_tiled[tile * 2048 + rho * 512 + m * 16 + b] = codes[t];
  if (m == 0) ids_tiled[tile * 64 + v] = ids[i];
}

int tile_codes(const uint8_t *codes, const int32_t *ids, const int32_t *dst_slot, int64_t n,
               int64_t ntiles, uint8_t *codes_tiled, int32_t *ids_tiled) {
  HIP_TRY(hipMemsetAsync(codes_tiled, 0, (size_t)ntiles * 2048, stream()));
  HIP_TRY(hipMemsetAsync(ids_tiled, 0xff, (size_t)ntiles * 64 * 4, stream()));
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(tile_codes_kernel, dim3((unsigned)cdiv(n * V2_M, 256)), dim3(256), 0,
                     stream(), codes, ids, dst_slot, n, codes_tiled, ids_tiled);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

bool pq_scan_v2_supported(int m, int ksub, int k, int nprobe) {
  return m == V2_M && ksub == V2_KSUB && nprobe <= V2_NT && k >= 1 && k <= TK_MAX_K;
}

template <int CAP>
static int launch_v2(const float *xq, int nq, int d, const float *codebooks, int dsub,
                     const float *coarse_D, const int32_t *coarse_I, int nprobe,
                     const int32_t *list_offsets, const int32_t *tile_offsets,
                     const uint8_t *codes_tiled, const int32_t *ids_tiled, int k, float *D,
                     int64_t *I64, int32_t *I32, int dbg) {
  size_t table_bytes = std::max((size_t)V2_CHUNK * sizeof(TileEnt), (size_t)d * 4);
  const size_t lds = (size_t)CAP * 8 + 64 + (size_t)V2_KSUB * V2_M * 4 + table_bytes;
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "pq scan: k=%d does not fit LDS", k);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)pq_scan_v2_kernel<CAP>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pq_scan_v2_kernel<CAP>, dim3(nq), dim3(V2_NT), lds, stream(), xq, d,
                     codebooks, dsub, coarse_D, coarse_I, nprobe, list_offsets, tile_offsets,
                     codes_tiled, ids_tiled, k, D, I64, I32, dbg);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int pq_scan_v2(const float *xq, int nq, int d, const float *codebooks, int dsub,
               const float *coarse_D, const int32_t *coarse_I, int nprobe,
               const int32_t *list_offsets, const int32_t *tile_offsets,
               const uint8_t *codes_tiled, const int32_t *ids_tiled, int k, float *D,
               int64_t *I64, int32_t *I32, int dbg) {
  if (nq <= 0) return ASL_OK;
#define V2_ARGS xq, nq, d, codebooks, dsub, coarse_D, coarse_I, nprobe, list_offsets, tile_offsets, \
                codes_tiled, ids_tiled, k, D, I64, I32, dbg
  // capacity rule: after a flush k keys stay; leave >= k slots of headroom on top of one
  // round's worst case so flushes stay logarithmic in the number of scanned vectors
  if (2 * k + V2_ROUND_VECS <= 2048) return launch_v2<2048>(V2_ARGS);
  if (2 * k + V2_ROUND_VECS <= 4096) return launch_v2<4096>(V2_ARGS);
  return launch_v2<8192>(V2_ARGS);
#undef V2_ARGS
}

}  // namespace asl
