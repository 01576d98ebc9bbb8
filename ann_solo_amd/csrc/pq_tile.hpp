// pq_tile.hpp -- shared device helpers of the tiled IVF-PQ scan kernels (m = 32, 8 bits):
// DPP butterflies and the 64-vector tile ADC (tile layout: pq_scan_v3.hip / DESIGN.md 5).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace asl {

constexpr int PQT_M = 32, PQT_KSUB = 256;

struct TileEnt {
  uint32_t tile;   // global tile index into codes_tiled / ids_tiled
  float coarse;    // q . centroid of the tile's list
  int32_t nvalid;  // vectors in the tile (64 except a list's last tile)
  int32_t pad;
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float lut_at(const char *lut_bytes, uint32_t word, int byte_idx,
                                        uint32_t lane_off) {
  uint32_t c = (word >> (8 * byte_idx)) & 0xffu;
  // keep the byte a value of its own: otherwise byte 0 is rewritten to (word << 7) & 0x7f80 and
  // costs three VALU instructions (shift, and, add) instead of two (and / bfe, lshl_add)
  asm volatile("" : "+v"(c));
  return *reinterpret_cast<const float *>(lut_bytes + ((c << 7) + lane_off));
}

typedef float pqt_f2 __attribute__((ext_vector_type(2)));

// 64 ADC sums of one tile: lane l returns the sum of vector l (without the coarse term).
__device__ __forceinline__ float tile_adc(const char *lut_bytes, const uint4 A, const uint4 B,
                                          uint32_t offA, uint32_t offB) {
  float v[16];
  const uint32_t a[4] = {A.x, A.y, A.z, A.w};
  const uint32_t b[4] = {B.x, B.y, B.z, B.w};
  // the 16 first-level sums as 8 packed fp32 adds (v_pk_add_f32: two IEEE adds per instruction,
  // same bits); the DPP butterflies below have no packed form
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    pqt_f2 x, y;
    x.x = lut_at(lut_bytes, a[r >> 2], r & 3, offA);
    x.y = lut_at(lut_bytes, a[(r + 1) >> 2], (r + 1) & 3, offA);
    y.x = lut_at(lut_bytes, b[r >> 2], r & 3, offB);
    y.y = lut_at(lut_bytes, b[(r + 1) >> 2], (r + 1) & 3, offB);
    const pqt_f2 z = x + y;
    v[r] = z.x;
    v[r + 1] = z.y;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = v[r] + dpp_mov<0x140>(v[r ^ 15]);   // row_mirror
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] + dpp_mov<0x141>(v[r ^ 7]);    // row_half_mirror
#pragma unroll
  for (int r = 0; r < 2; ++r) v[r] = v[r] + dpp_mov<0x1B>(v[r ^ 3]);     // quad_perm [3,2,1,0]
  return v[0] + dpp_mov<0xB1>(v[1]);                                      // quad_perm [1,0,3,2]
}

// Per-query LUT image lut[c*32 + m]. Thread c owns code word c; the codebooks are read
// from the TRANSPOSED copy cbT[m][t][c] so that a wave's load is one contiguous 256-B
// line (the [m][c][t] original makes every lane touch its own cache line: measured 4x
// slower). The image's bank is the sub-quantiser (that is what makes the scan's reads
// conflict-free), so a wave that stores ONE sub-quantiser's 64 entries hits one bank 64 times:
// until round 4 that was ~13 such stores per thread, ~6 k LDS cycles per table -- nothing next to
// a 16 384-query scan, a third of the per-(query, shard) fixed cost of a sharded one. Now a
// thread keeps its 16 (32) finished sums in registers and stores them as 16-byte groups, the
// lanes of one instruction rotated over the groups (lane l stores group (i + l) & 3 in
// instruction i): 4 (8) stores per thread, 16 banks busy at a time. The table needs no zeroing
// pass either: every entry is written.
//
// Zero query components are SKIPPED: fmaf(0, cb, acc) == acc bit for bit (finite
// codebooks), so the chain over the non-zero components in ascending t is the canonical
// ascending-t chain. A hashed spectrum has <= 50 non-zeros of 800, which cuts the
// codebook traffic of a table from 819 KB to ~50 KB (the table build was the largest
// per-query fixed cost of a sharded search). The non-zeros of all sub-vectors are listed
// once (wave 0) and walked flat, U codebook loads in flight per thread: the build is
// latency-bound otherwise. s_q: staging of the query; s_nz: 8 + 2*d bytes of scratch.
//
// ENTRY LISTS (ent_row != nullptr, 0 <= ent_cnt <= 64): the query's non-zero components come
// ready-made, ascending, as (dimension * 128, value bits) -- list_nonzeros_kernel of
// coarse_sparse.hip. Wave 0 then needs one coalesced 512-byte load instead of the 3.2 KB row,
// its staging barrier and the serial listing: ~4 of the ~12 us a table costs, which is paid once
// per (query, shard) in a sharded search.
template <int NT = 256>
__device__ __forceinline__ void build_lut_cbt(const float *__restrict__ xq_row, int d,
                                              const float *__restrict__ cbT, int dsub,
                                              float *s_q, float *s_lut, uint8_t *s_nz, int tid,
                                              const uint2 *__restrict__ ent_row = nullptr,
                                              int ent_cnt = -1) {
  // s_nz: [0..4) K (int), [4..8) K16 = entries of the sub-quantisers 0..15, then K
  // sub-quantiser indices at +8 and K offsets t at +8+d (entry lists: t at +72, values at +136)
  const bool fast = ent_row != nullptr && ent_cnt >= 0;      // block-uniform
  uint8_t *nz_m = s_nz + 8, *nz_t = s_nz + 8 + (fast ? 64 : d);
  float *nz_v = reinterpret_cast<float *>(s_nz + 136);
  if (!fast) {
    for (int i = tid; i < d; i += NT) s_q[i] = xq_row[i];
    __syncthreads();
  }
  if (fast) {
    if (tid < 64) {   // wave 0: one entry per lane (ascending dimension = ascending (m, t))
      const bool live = tid < ent_cnt;
      const uint2 e = live ? ent_row[tid] : make_uint2(0u, 0u);
      const int dim = (int)(e.x >> 7);
      const int m = dim / dsub;
      nz_m[tid] = (uint8_t)m;
      nz_t[tid] = (uint8_t)(dim - m * dsub);
      nz_v[tid] = __uint_as_float(e.y);
      const unsigned long long lo = __ballot(live && m < 16);
      if (tid == 0) {
        reinterpret_cast<int *>(s_nz)[0] = ent_cnt;
        reinterpret_cast<int *>(s_nz)[1] = __popcll(lo);
      }
    }
  } else if (tid < 64) {   // wave 0: lane m lists the non-zero components of sub-vector m, in order
    const int m = tid;
    // per lane a bit mask of its non-zero components, read 8 at a time (independent LDS reads
    // in flight; one read at a time made this serial listing half of the table build)
    unsigned long long nzmask = 0ull;     // dsub <= 64 (checked by the launcher)
    if (m < PQT_M)
      for (int t0 = 0; t0 < dsub; t0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t0 + u < dsub ? s_q[m * dsub + t0 + u] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (v[u] != 0.0f) nzmask |= 1ull << (t0 + u);
      }
    const int cnt = __popcll(nzmask);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (tid >= o) incl += v;
    }
    int pos = incl - cnt;
    while (nzmask) {
      const int t = __builtin_ctzll(nzmask);
      nzmask &= nzmask - 1ull;
      nz_m[pos] = (uint8_t)m;
      nz_t[pos] = (uint8_t)t;
      ++pos;
    }
    if (tid == 63) reinterpret_cast<int *>(s_nz)[0] = incl;
    if (tid == 16) reinterpret_cast<int *>(s_nz)[1] = incl - cnt;
  }
  __syncthreads();
  // 256 threads: thread c walks all K entries. 512 threads: the two halves of the workgroup
  // split the sub-quantisers (0..15 | 16..31), each thread still owns one code word.
  const int Kall = reinterpret_cast<const int *>(s_nz)[0];
  const int K16 = reinterpret_cast<const int *>(s_nz)[1];
  const int c = tid & (PQT_KSUB - 1);
  // wave-uniform values, kept in scalar registers (the register file below is indexed by them)
  const int kbeg = __builtin_amdgcn_readfirstlane((NT > PQT_KSUB && tid >= PQT_KSUB) ? K16 : 0);
  const int K = __builtin_amdgcn_readfirstlane((NT > PQT_KSUB && tid < PQT_KSUB) ? K16 : Kall);
  constexpr int U = 8;   // codebook loads in flight per thread
  constexpr int MS = NT > PQT_KSUB ? PQT_M / 2 : PQT_M;   // sub-quantisers per thread
  const int mbase = __builtin_amdgcn_readfirstlane((NT > PQT_KSUB && tid >= PQT_KSUB) ? PQT_M / 2 : 0);
  typedef float lut_regs __attribute__((ext_vector_type(MS)));
  lut_regs r = 0.0f;     // indexed by a scalar below (v_movrel / gpr-index mode, no scratch)
  float acc = 0.0f;
  int cur = mbase;       // "the run of sub-quantiser mbase, nothing summed yet"
  for (int k0 = kbeg; k0 < K; k0 += U) {
    float cbv[U], qv[U];
    int mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u < K ? k0 + u : K - 1;
      mm[u] = nz_m[k];
      const int e = mm[u] * dsub + nz_t[k];
      qv[u] = fast ? nz_v[k] : s_q[e];
      cbv[u] = cbT[(size_t)e * PQT_KSUB + c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u < K) {               // block-uniform
        const int m_u = __builtin_amdgcn_readfirstlane(mm[u]);
        if (m_u != cur) {             // block-uniform: a sub-quantiser's run ends
          asm volatile("; a run ends" ::: "memory");   // keeps this a (scalar) branch: as a select it is 16 v_cndmask per entry
          r[cur - mbase] = acc;
          acc = 0.0f;
          cur = m_u;
        }
        acc = __builtin_fmaf(qv[u], cbv[u], acc);
      }
    }
  }
  r[cur - mbase] = acc;
  // groups of four sub-quantisers, 16 bytes each; lane l stores group (i + l) mod NG in
  // instruction i
  constexpr int NG = MS / 4;
  float4 *row = reinterpret_cast<float4 *>(s_lut + c * PQT_M + mbase);
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int g = (i + tid) & (NG - 1);
    float4 v = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
    for (int h = 1; h < NG; ++h)
      if (g == h) v = make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
    row[g] = v;
  }
  __syncthreads();
}

}  // namespace asl
