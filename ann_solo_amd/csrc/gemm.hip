// gemm.hip -- C[M,N] = A[M,K] . B[N,K]^T in exact fp32 on the MFMA pipe.
//
// Used for the coarse quantiser (queries x centroids, FAISS IndexFlatIP inside
// IndexIVFFlat, /root/reference/src/ann_solo/spectral_library.py:167-176), k-means
// assignment and exact brute-force search. v_mfma_f32_32x32x2_f32 accumulates as a
// k-ordered fmaf chain (one rounding per product, no wider accumulator), so every
// C element is bit-identical to the oracle's `acc = fmaf(a[k], b[k], acc)` loop:
// probe lists and id sets match the CPU restatement exactly, not approximately.
//
// Tiling (wave64): 128x128 block tile, BK=16, 4 waves as 2x2, each wave 64x64 =
// 2x2 MFMA tiles of 32x32 (64 accumulator registers). Operands are staged through
// LDS k-major ([k][row], row stride 132 floats) so an MFMA fragment read is one
// conflict-free ds_read_b32 per operand; global loads are float4 along K, double
// buffered through registers (issue tile t+1, compute tile t, then store).
// blockIdx is remapped so that each XCD (private L2) owns a contiguous band of
// row tiles.
#include "common.hpp"

namespace asl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = 132;

struct GemmTileRegs {
  float4 a[2], b[2];
};

template <bool VEC4>
__device__ __forceinline__ float4 load4(const float *__restrict__ p, int row, int nrows,
                                        int ld, int k, int K) {
  const int r = row < nrows ? row : nrows - 1;
  const float *src = p + (size_t)r * ld + k;
  float4 v;
  if (VEC4) {
    if (k + 3 < K) {
      v = *reinterpret_cast<const float4 *>(src);
    } else {
      v.x = k < K ? src[0] : 0.0f;
      v.y = k + 1 < K ? src[1] : 0.0f;
      v.z = k + 2 < K ? src[2] : 0.0f;
      v.w = 0.0f;
    }
  } else {
    v.x = k < K ? src[0] : 0.0f;
    v.y = k + 1 < K ? src[1] : 0.0f;
    v.z = k + 2 < K ? src[2] : 0.0f;
    v.w = k + 3 < K ? src[3] : 0.0f;
  }
  return v;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
    int N, int K, int lda, int ldb, int ldc, int nbx, int nby, const int *__restrict__ gate,
    int gate_max) {
  // gated launch (coarse quantiser): the sparse kernel scores the batch unless more than
  // gate_max query rows are dense -- decided on the device, no host round trip
  if (gate && *gate <= gate_max) return;
  __shared__ float As[2][GBK][GLD];
  __shared__ float Bs[2][GBK][GLD];
  // XCD-aware remap: consecutive workgroup ids round-robin over the 8 XCDs; give
  // each XCD a contiguous chunk of the tile grid (bijective for any grid size).
  const int nwg = nbx * nby;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, idx = orig >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int by = logical / nbx, bx = logical - by * nbx;
  const int m0 = by * GBM, n0 = bx * GBN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lkq = (tid & 3) * 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  GemmTileRegs t;
  auto gload = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      t.a[h] = load4<VEC4>(A, m0 + lrow + 64 * h, M, lda, k0 + lkq, K);
      t.b[h] = load4<VEC4>(B, n0 + lrow + 64 * h, N, ldb, k0 + lkq, K);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = lrow + 64 * h;
      As[buf][lkq + 0][r] = t.a[h].x;
      As[buf][lkq + 1][r] = t.a[h].y;
      As[buf][lkq + 2][r] = t.a[h].z;
      As[buf][lkq + 3][r] = t.a[h].w;
      Bs[buf][lkq + 0][r] = t.b[h].x;
      Bs[buf][lkq + 1][r] = t.b[h].y;
      Bs[buf][lkq + 2][r] = t.b[h].z;
      Bs[buf][lkq + 3][r] = t.b[h].w;
    }
  };

  const int nk = (K + GBK - 1) / GBK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GBK);
#pragma unroll
    for (int kk = 0; kk < GBK / 2; ++kk) {
      const int k = 2 * kk + fk;
      const float a0 = As[buf][k][wm * 64 + frow];
      const float a1 = As[buf][k][wm * 64 + 32 + frow];
      const float b0 = Bs[buf][k][wn * 64 + frow];
      const float b1 = Bs[buf][k][wn * 64 + 32 + frow];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) C[(size_t)row * ldc + col] = acc[i][j][r];
      }
    }
}

// C = A . B^T, all device pointers, row-major with leading dimensions.
int gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K, int lda,
                int ldb, int ldc, const int *gate, int gate_max) {
  if (M <= 0 || N <= 0) return ASL_OK;
  if (K <= 0) return fail(ASL_ERR_INVALID, "gemm: K must be positive");
  const int nbx = (int)cdiv(N, GBN), nby = (int)cdiv(M, GBM);
  const bool vec4 = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A) % 16 == 0) &&
                    (((uintptr_t)B) % 16 == 0);
  dim3 grid((unsigned)(nbx * nby));
  if (vec4)
    hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, grid, dim3(256), 0, stream(), A, B, C, M, N,
                       K, lda, ldb, ldc, nbx, nby, gate, gate_max);
  else
    hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, grid, dim3(256), 0, stream(), A, B, C, M, N,
                       K, lda, ldb, ldc, nbx, nby, gate, gate_max);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl
