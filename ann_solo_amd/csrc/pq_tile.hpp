// pq_tile.hpp -- shared device helpers of the tiled IVF-PQ scan kernels (m = 32, 8 bits):
// DPP butterflies and the 64-vector tile ADC (see pq_scan_v2.hip for the layout).
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
  const uint32_t c = (word >> (8 * byte_idx)) & 0xffu;
  return *reinterpret_cast<const float *>(lut_bytes + ((c << 7) + lane_off));
}

// 64 ADC sums of one tile: lane l returns the sum of vector l (without the coarse term).
__device__ __forceinline__ float tile_adc(const char *lut_bytes, const uint4 A, const uint4 B,
                                          uint32_t offA, uint32_t offB) {
  float v[16];
  const uint32_t a[4] = {A.x, A.y, A.z, A.w};
  const uint32_t b[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
  for (int r = 0; r < 16; ++r)
    v[r] = lut_at(lut_bytes, a[r >> 2], r & 3, offA) + lut_at(lut_bytes, b[r >> 2], r & 3, offB);
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
// slower). The 32 stores of a wave all hit bank m (64-way conflict, 64 LDS cycles each) --
// cheaper than any scheme that scatters the codebook reads. s_q: staging of the query.
__device__ __forceinline__ void build_lut_cbt(const float *__restrict__ xq_row, int d,
                                              const float *__restrict__ cbT, int dsub,
                                              float *s_q, float *s_lut, int tid) {
  for (int i = tid; i < d; i += 256) s_q[i] = xq_row[i];
  __syncthreads();
  const int c = tid;
  for (int m = 0; m < PQT_M; ++m) {
    const float *cb = cbT + (size_t)m * dsub * PQT_KSUB + c;
    const float *qs = s_q + m * dsub;
    float acc = 0.0f;
    for (int t = 0; t < dsub; ++t) acc = __builtin_fmaf(qs[t], cb[(size_t)t * PQT_KSUB], acc);
    s_lut[c * PQT_M + m] = acc;
  }
  __syncthreads();
}

}  // namespace asl
