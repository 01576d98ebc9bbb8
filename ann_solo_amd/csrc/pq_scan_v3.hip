// pq_scan_v3.hip -- IVF-PQ asymmetric-distance scan, m = 32 sub-quantisers of 8 bits, laid out
// for gfx950's LDS: the dominant kernel of the hot path (replaces the list scan inside FAISS'
// IndexIVF*.search, call site /root/reference/src/ann_solo/spectral_library.py:443-444).
//
// Layout (helpers in pq_tile.hpp). "One lane per vector" makes 32 random LUT reads per vector
// hit random LDS banks (~3.5-way conflicts on ds_read_b32); here lanes are SUB-QUANTISERS:
//   * LUT image in LDS: lut[c*32 + m] (row stride 128 B) -> bank == m. A 16-lane row reads 16
//     different m for one code position, the neighbouring row the other 16: every ds_read_b32
//     is bank-conflict-free BY CONSTRUCTION, whatever the codes are.
//   * codes are stored in 64-vector tiles (2 KiB): chunk[rho][m][16 B], the 16 bytes being
//     sub-quantiser m's codes of the row's 16 vectors, permuted b -> b ^ (m & 15). Lane
//     (rho, j) loads the chunks of m = j and m = j + 16 (two coalesced 16-B loads).
//   * register r of lane j then holds p_j(vector r ^ j); four DPP butterflies (row_mirror,
//     row_half_mirror, quad [3,2,1,0], quad [1,0,3,2]) reduce the 16x16 block with 15
//     v_add_dpp and NO selects, leaving vector j's sum in lane j -- the canonical mirror tree
//     of DESIGN.md, bit-identical to the oracle.
//   * ids are fetched only for the survivors of the top-k.
//
// Top-k without sorting while streaming (hist_topk.hpp). Only the k-th best SCORE is needed
// while scanning, so the kernel keeps
//   * hist[512]: counts of the appended candidates per score bucket (monotone linear
//     bucketing of the fp32 score over [-0.25, 1), i.e. 0.0024 per bucket),
//   * bstar: the highest bucket with at least k appended candidates at or above it.
// A candidate whose bucket is below bstar can never be among the k best (k candidates
// with strictly larger scores exist), so it is dropped without being stored; when the key
// buffer runs full it is COMPACTED by the same test (one prefix scan, no sort). The k
// best are sorted exactly once, at the end, after the storage slots of the survivors
// have been turned into ids -- (score desc, id asc), identical to the oracle.
// If compaction cannot free the buffer (thousands of bit-identical scores) the kernel
// switches, for that query, to exact sort-and-truncate flushes.
//
// LDS: LUT 32 KB + key buffer (2048 or 4096 keys) + histogram + tile table => three (two)
// workgroups of 8 waves per CU.
#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"
#include "pq_tile.hpp"

namespace asl {

constexpr int V3_CHUNK = 256;   // tile-table entries per chunk
// Tuning constants of the code stream (scripts/build_variant.sh rewrites them for same-box A/Bs;
// profiles/r06_pq_beyond_llc_*: measured both where the codes sit in the Infinity Cache and where
// they come from DRAM).
constexpr int V3_DEPTH = 1;     // rounds of tiles in flight ahead of the one being scored
constexpr int V3_NT = 0;        // 1: non-temporal loads of the codes (streamed once per query)
constexpr int V3_WAVES_PER_SIMD = 6;   // occupancy the 2048-key instantiation is compiled for (3 workgroups per CU)

template <int NT_>
__device__ __forceinline__ uint4 load_codes16(const uint8_t *p) {
  if (NT_) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(w));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const uint4 *>(p);
}

// LDS tile-table entry, 8 bytes: tile index (26 bits: ids are int32, so an index holds fewer
// than 2^25 + nlist tiles) | nvalid - 1 (6 bits), coarse term. Half the size of TileEnt: 256
// entries per chunk in the same 2 KB, i.e. half as many pipeline restarts (7.05 -> 6.84 ms).
struct TileEnt8 {
  uint32_t tile_nv;
  float coarse;
};

// NW = waves per workgroup. LDS (LUT 32 KB + keys 16 KB + ...) allows three workgroups per CU
// whatever their size, so 8 waves per workgroup (one tile per wave and round) double the
// waves that share one LUT and one key buffer: 24 waves per CU at 80 VGPRs instead of 12 at
// 157 -- measured 8.28 -> 7.46 ms at the bench config (with ONE round of prefetch: at this
// occupancy the second prefetch stage only costs registers).
template <int CAP, int T, int NW, int DEPTH, bool WIDE>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (CAP <= 2048 ? V3_WAVES_PER_SIMD : 4) : (CAP <= 2048 ? 3 : 1))) void pq_scan_v3_kernel(
    const float *__restrict__ xq, int d, const float *__restrict__ codebooks, int dsub,
    const float *__restrict__ coarse_D, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ list_offsets, const int32_t *__restrict__ tile_offsets,
    const uint8_t *__restrict__ codes_tiled, const int32_t *__restrict__ ids_tiled, int k,
    float *__restrict__ D, int64_t *__restrict__ I64, int32_t *__restrict__ I32, int set_mode,
    const uint2 *__restrict__ ent, const int32_t *__restrict__ ent_cnt, const int *__restrict__ gate,
    const ScanPostFilter pf) {
  // gate: a device-side row count -- workgroups past it leave at once (a launch of fixed size over
  // a list whose length only the device knows: the shard-side rescans of exchange.hip)
  if (gate && (int)blockIdx.x >= *gate) return;
  static_assert(DEPTH >= 1 && DEPTH <= 3, "rounds of prefetch");
  constexpr int NT = 64 * NW, ROUND_TILES = NW * T, ROUND_VECS = ROUND_TILES * 64;
  using TopK = HistTopK<CAP, ROUND_VECS, NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_lut = reinterpret_cast<float *>(smem + TopK::lds_bytes());
  TileEnt8 *table = reinterpret_cast<TileEnt8 *>(s_lut + PQT_KSUB * PQT_M);
  float *s_q = reinterpret_cast<float *>(smem);  // aliases the key buffer during the LUT build

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
  // ---- my probe (thread p < nprobe; WIDE: probes p and p + NT -- nprobe up to 2 x NT = 1024, the
  // reference's clamp, spectral_library.py:77-81): its dependent gathers are issued before the
  // table build and complete under it
  constexpr int PP = WIDE ? 2 : 1;
  int my_len[PP], my_tile0[PP], my_nt[PP], my_pre[PP];
  float my_coarse[PP];
#pragma unroll
  for (int pp = 0; pp < PP; ++pp) {
    my_len[pp] = 0, my_tile0[pp] = 0, my_nt[pp] = 0, my_coarse[pp] = 0.0f;
    const int p = tid + pp * NT;
    if (p < nprobe) {
      const int l = coarse_I[(size_t)q * nprobe + p];
      if (l >= 0) {
        my_len[pp] = list_offsets[l + 1] - list_offsets[l];
        my_tile0[pp] = tile_offsets[l];
        my_nt[pp] = (my_len[pp] + 63) >> 6;
        my_coarse[pp] = coarse_D[(size_t)q * nprobe + p];
      }
    }
  }
  build_lut_cbt<NT>(xq + (size_t)q * d, d, codebooks, dsub, s_q, s_lut, reinterpret_cast<uint8_t *>(table),
                    tid, ent ? ent + (size_t)q * 64 : nullptr,
                    ent ? (xq ? ent_cnt[q] : max(ent_cnt[q], 0)) : -1);  // codebooks = cbT[m][t][c]; the tile table is
                                                                        // not live yet; entry lists only: a row with
                                                                        // more than 64 non-zeros is searched as all-zero

  // ---- exclusive scan of the probes' tile counts (probe order: p, then p + NT)
  int total = 0;
  int *scan_part = reinterpret_cast<int *>(table);   // table is not live yet
#pragma unroll
  for (int pp = 0; pp < PP; ++pp) {
    int part_total;
    my_pre[pp] = total + block_excl_scan<NW>(my_nt[pp], scan_part, tid, part_total);
    total += part_total;
    __syncthreads();
  }

  TopK top;   // init zeroes the keys (which aliased s_q)
  top.init(smem, k, ids_tiled, tid);
  top.out_keys = set_mode == 2 && I64 != nullptr;

  const char *lut_bytes = reinterpret_cast<const char *>(s_lut);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int rho = lane >> 4, j = lane & 15;
  const int ma = (rho & 1) ? j + 16 : j, mb = ma ^ 16;
  const uint32_t offA = (uint32_t)ma * 4u, offB = (uint32_t)mb * 4u;
  const uint32_t chunkA = (uint32_t)(rho * 512 + ma * 16), chunkB = (uint32_t)(rho * 512 + mb * 16);

  for (int c0 = 0; c0 < total; c0 += V3_CHUNK) {
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
      const int lo = max(my_pre[pp], c0), hi = min(my_pre[pp] + my_nt[pp], c0 + V3_CHUNK);
      for (int t = lo; t < hi; ++t) {
        const int local = t - my_pre[pp];
        TileEnt8 e;
        e.tile_nv = (uint32_t)(my_tile0[pp] + local) | ((uint32_t)(min(64, my_len[pp] - local * 64) - 1) << 26);
        e.coarse = my_coarse[pp];
        table[t - c0] = e;
      }
    }
    __syncthreads();
    const int nent = min(V3_CHUNK, total - c0);
    const int nrounds = (nent + ROUND_TILES - 1) / ROUND_TILES;
    // Register pipeline, DEPTH rounds deep (the codes of rounds rr + 1 .. rr + DEPTH are in flight
    // while round rr is scored). DEPTH + 1 register sets, rotated by full unrolling: every index
    // below is a compile-time constant.
    constexpr int NS = DEPTH + 1;
    uint4 A[NS][T], B[NS][T];
    TileEnt e[NS][T];
    auto fetch = [&](int rr, uint4 *a, uint4 *b, TileEnt *en) {
#pragma unroll
      for (int u = 0; u < T; ++u) {
        // the entry is the same for the whole wave: keep it in scalar registers
        const int i = rr * ROUND_TILES + wave_u * T + u;
        const TileEnt8 t = table[i < nent ? i : 0];
        const uint32_t tn = __builtin_amdgcn_readfirstlane(t.tile_nv);
        en[u].tile = tn & 0x3ffffffu;
        en[u].coarse = __builtin_bit_cast(
            float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.coarse)));
        en[u].nvalid = i < nent ? (int)(tn >> 26) + 1 : 0;
        const uint8_t *base = codes_tiled + (size_t)en[u].tile * 2048;
        a[u] = load_codes16<V3_NT>(base + chunkA);
        b[u] = load_codes16<V3_NT>(base + chunkB);
      }
    };
    auto process = [&](const uint4 *a, const uint4 *b, const TileEnt *en) {
      top.begin_round();
      int appended = 0;
#pragma unroll
      for (int u = 0; u < T; ++u) {
        if (en[u].nvalid > 0) {  // wave-uniform
          const float score = en[u].coarse + tile_adc(lut_bytes, a[u], b[u], offA, offB);
          const bool take = top.offer(lane < en[u].nvalid, score,
                                      en[u].tile * 64u + (uint32_t)lane);
          appended += __popcll(__ballot(take));
        }
      }
      top.end_round(appended);
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
      if (s < nrounds) fetch(s, A[s], B[s], e[s]);
    for (int rr = 0; rr < nrounds; rr += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int r = rr + s;
        if (r < nrounds) {          // block-uniform
          if (r + DEPTH < nrounds) fetch(r + DEPTH, A[(s + DEPTH) % NS], B[(s + DEPTH) % NS], e[(s + DEPTH) % NS]);
          process(A[s], B[s], e[s]);
        }
      }
    }
    __syncthreads();
  }
  if (set_mode && CAP * 9 <= PQT_KSUB * PQT_M * 4)   // unordered exact top-k; the LUT is dead: scratch
    top.finish_set(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
                   I32 ? I32 + (size_t)q * k : nullptr, reinterpret_cast<u64 *>(s_lut), &pf, q);
  else if (set_mode && CAP * 8 <= PQT_KSUB * PQT_M * 4)
    top.finish_set(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
                   I32 ? I32 + (size_t)q * k : nullptr, reinterpret_cast<u64 *>(s_lut));
  else
    top.finish(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
               I32 ? I32 + (size_t)q * k : nullptr);
}

template <int CAP, int T, int NW, int DEPTH, bool WIDE>
static int launch_v3(const float *xq, int nq, int d, const float *codebooks, int dsub,
                     const float *coarse_D, const int32_t *coarse_I, int nprobe,
                     const int32_t *list_offsets, const int32_t *tile_offsets,
                     const uint8_t *codes_tiled, const int32_t *ids_tiled, int k, float *D,
                     int64_t *I64, int32_t *I32, int set_mode, const uint2 *ent,
                     const int32_t *ent_cnt, const int *gate, const ScanPostFilter &pf) {
  if ((size_t)d * 4 > (size_t)CAP * 8 || dsub > 64 ||
      (size_t)d * 2 + 8 > (size_t)V3_CHUNK * sizeof(TileEnt8) || d != PQT_M * dsub)
    return fail(ASL_ERR_CAPACITY, "pq scan: d=%d too large for the LDS staging", d);
  const size_t lds = HistTopK<CAP, NW * T * 64, 64 * NW>::lds_bytes() + (size_t)PQT_KSUB * PQT_M * 4 +
                     (size_t)V3_CHUNK * sizeof(TileEnt8);
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "pq scan: k=%d does not fit LDS", k);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)pq_scan_v3_kernel<CAP, T, NW, DEPTH, WIDE>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((pq_scan_v3_kernel<CAP, T, NW, DEPTH, WIDE>), dim3(nq), dim3(64 * NW), lds, stream(), xq, d,
                     codebooks, dsub, coarse_D, coarse_I, nprobe, list_offsets, tile_offsets,
                     codes_tiled, ids_tiled, k, D, I64, I32, set_mode, ent, ent_cnt, gate, pf);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

bool pq_scan_tiled_supported(int m, int ksub, int k, int nprobe) {
  return m == PQT_M && ksub == PQT_KSUB && nprobe <= 1024 && k >= 1 && k <= TK_MAX_K;
}

// k <= 1280: 2048-key buffer, three workgroups per CU; larger k: 4096 keys, two per CU
int pq_scan_v3(const float *xq, int nq, int d, const float *codebooks, int dsub,
               const float *coarse_D, const int32_t *coarse_I, int nprobe,
               const int32_t *list_offsets, const int32_t *tile_offsets,
               const uint8_t *codes_tiled, const int32_t *ids_tiled, int k, float *D,
               int64_t *I64, int32_t *I32, int set_mode, const uint2 *ent, const int32_t *ent_cnt,
               const int *gate, const ScanPostFilter *post) {
  if (nq <= 0) return ASL_OK;
  // the post-filter needs the set-mode finish of the 2048-key instantiation (its scratch behind the keys)
  ScanPostFilter pf;
  if (post && post->idpay) {
    if (!(set_mode == 1 && I32 && k + 256 + 512 <= 2048))
      return fail(ASL_ERR_STATE, "pq scan: a post-filter needs set-mode int32 rows and k <= 1280");
    pf = *post;
  }
#define V3_ARGS xq, nq, d, codebooks, dsub, coarse_D, coarse_I, nprobe, list_offsets, tile_offsets, \
                codes_tiled, ids_tiled, k, D, I64, I32, set_mode, ent, ent_cnt, gate, pf
  if (nprobe > 512) {         // two probes per thread (the one-probe form keeps its registers)
    if (k + 256 + 512 <= 2048) return launch_v3<2048, 1, 8, V3_DEPTH, true>(V3_ARGS);
    return launch_v3<4096, 1, 8, V3_DEPTH, true>(V3_ARGS);
  }
  if (k + 256 + 512 <= 2048) return launch_v3<2048, 1, 8, V3_DEPTH, false>(V3_ARGS);
  return launch_v3<4096, 1, 8, V3_DEPTH, false>(V3_ARGS);
#undef V3_ARGS
}

// list-ordered codes [n,32] -> 64-vector tiles (see file header); dst_slot[i] = tile*64 + v
__global__ void tile_codes_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ ids,
                                  const int32_t *__restrict__ dst_slot, int64_t n,
                                  uint8_t *__restrict__ codes_tiled,
                                  int32_t *__restrict__ ids_tiled) {
  const int64_t t = block_linear() * blockDim.x + threadIdx.x;
  if (t >= n * PQT_M) return;
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
  hipLaunchKernelGGL(tile_codes_kernel, grid_2d(cdiv(n * PQT_M, 256)), dim3(256), 0,
                     stream(), codes, ids, dst_slot, n, codes_tiled, ids_tiled);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl
