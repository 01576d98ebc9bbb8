// flat_scan.hip -- exact IVF-Flat list scan over per-dimension POSTINGS inside every inverted
// list (replaces the scan inside FAISS IndexIVFFlat.search,
// /root/reference/src/ann_solo/spectral_library.py:443-444).
//
// A hashed spectrum vector has at most ~50 non-zeros out of 800 (one per peak). The score of a
// stored vector is acc = fmaf(q[dim], val, acc) over its non-zeros in ascending dimension --
// the ascending-index fp32 chain restricted to the stored non-zeros, which is bit-identical to
// the dense chain (a zero factor leaves the accumulator unchanged), to the fp32-MFMA GEMM and
// to the oracle. Top-k: hist_topk.hpp (no sorting while streaming).
#include <cstdlib>

#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"
#include "pq_tile.hpp"

namespace asl {

// non-zeros per vector (one wave per vector) and their maximum: decides at build time whether
// the postings layout pays
// nnz_max[1] is raised when a non-zero component is NOT a fixed-point value (0 < x < 1 on the
// grid of 2^-22): such an index keeps float postings
__global__ void count_nnz_kernel(const float *__restrict__ vecs, int d, int64_t n,
                                 int32_t *__restrict__ nnz, int32_t *__restrict__ nnz_max) {
  const int64_t v = block_linear() * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (v >= n) return;
  int c = 0;
  bool off_grid = false;
  for (int j = lane; j < d; j += 64) {
    const float x = vecs[v * d + j];
    c += x != 0.0f;
    off_grid |= x != 0.0f && !fx22_on_grid(x);
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if (__ballot(off_grid) && lane == 0) atomicMax(nnz_max + 1, 1);
  if (lane == 0) {
    nnz[v] = c;
    atomicMax(nnz_max, c);
  }
}

// add() of an IVF-Flat index in fixed-point storage mode: components in [0, 1) go to the nearest
// multiple of 2^-22 (ties to even; the largest is 1 - 2^-22); anything else is stored as given
// (and keeps the whole index on float postings). The oracle's orc_quantize_fx22 is the same rule.
__global__ void quantize_fx22_kernel(float *__restrict__ x, int64_t n) {
  const int64_t i = block_linear() * blockDim.x + threadIdx.x;
  if (i < n) x[i] = fx22_round(x[i]);
}

int quantize_fx22(float *x, int64_t n) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(quantize_fx22_kernel, grid_2d(cdiv(n, 256)), dim3(256), 0, stream(), x, n);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int count_nnz(const float *vecs, int d, int64_t n, int32_t *nnz, int32_t *nnz_max_dev) {
  HIP_TRY(hipMemsetAsync(nnz_max_dev, 0, 2 * sizeof(int32_t), stream()));
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(count_nnz_kernel, grid_2d(cdiv(n, 4)), dim3(256), 0, stream(), vecs, d,
                     n, nnz, nnz_max_dev);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ---------------------------------------------------------------------------------------
// Dimension-major ("postings") IVF-Flat. Streaming EVERY stored non-zero of the probed lists
// (~27 per vector; the sparse-tile scan of round 1) reads 16x more than needed: a query has
// <= ~50 non-zero components of 800 and only stored entries in those dimensions can
// contribute. So every inverted list is cut into blocks
// of FI_BLK vectors and each block keeps, per dimension, the postings (local vector index
// u16, value f32) of the vectors that are non-zero there -- an inverted file inside the
// inverted file. Storage: one segment per (block, dimension), its c values followed by its c
// local indices (6c bytes), and one table word per (block, dimension): the segment's start in
// 64-byte units from the block's base (high 16 bits) and c (low 16 bits). Every (query, block)
// pair touches its own lines (L2 hit rate < 10 %), so segments are PLACED BY LINE: one that
// fits a 128-byte line never straddles two, a longer one starts on a line boundary -- a
// dimension of a 512-vector list (17 postings, 102 bytes) costs one line instead of 1.8, for
// ~25 % padding. A wave takes one (query, block): it zeroes the block's accumulators in LDS
// and walks the query's non-zero dimensions in ASCENDING order, acc[loc] = fmaf(q_d, val,
// acc[loc]) over that dimension's postings (a vector occurs at most once per dimension, so
// the lanes of one step never collide -- lanes past a row's count repeat its last posting and
// store the same bits, see the row loop -- and steps of one wave reach LDS in program order).
// Per vector this is the ascending-dimension fp32 chain restricted to the dimensions where
// both factors are non-zero -- bit-identical to the dense chain, the MFMA GEMM and the oracle
// (a zero factor leaves the accumulator unchanged) -- at 1/16 of the tile kernel's traffic
// (50 of 800 dimensions) and 1/4 of its lane-steps. Then the wave offers its accumulators to
// the workgroup's histogram top-k, vectors with score 0 included (they are candidates of the
// dense scan too). The eight waves of a workgroup run their blocks independently (blocks are
// handed out by an LDS counter, a block's candidates are appended with one atomic, or one per
// row when they are many) and meet only when the key buffer is full: see "Offers" below.
// FI_CHUNK: blocks of a query listed at a time (a query of the bench probes ~150: one chunk, so the
// waves of a workgroup wait for each other once per query)
constexpr int FI_NW = 8, FI_NT = 64 * FI_NW, FI_CHUNK = 256, FI_U = 8;   // FI_U: rows in flight per wave
constexpr int FI_ROWS = (FI_BLK + 63) / 64;      // rows of accumulators in a block
static_assert(64 % FI_U == 0, "a batch of dimensions must not straddle the 64 lanes of a chunk");

// (wave_incl_scan -- the seven-DPP-add prefix sum of hist_topk.hpp -- runs twice per block in the
// per-block bookkeeping: the kernel is VALU-bound, profiles/r04_ivfflat_np112_pmc_summary.txt:
// SQ_ACTIVE_INST_VALU 89 % of the SIMD cycles)

struct FiUnit {
  uint32_t blk;   // block index into the per-dimension table
  int32_t pos0;   // list-order position of the block's first vector
};                // (+ the block's vector count in a 16-bit array: 10 bytes per block of a chunk)

// FI_CAP: key buffer of the top-k (2048: k <= 1280, three workgroups per CU; 4096: k <= 3328, two)
//
// FX = true: the FIXED-POINT layout (round 4). Every stored component lies in (0, 1) on the grid
// of 2^-22 (asl_index_set_flat_storage: add() rounds them there), so a posting is ONE 32-bit word
// -- value numerator M (22 bits) << 10 | local vector index (10 bits) -- instead of a float and
// a 16-bit index. Segments are whole 128-byte lines of 32 postings, the tail of the last line
// repeating the last posting (a lane that applies a repeat reads the same accumulator, computes
// the same sum and stores the same bits -- the rule the row loop already relies on), so a
// segment needs no count, and the table shrinks from a 4-byte word to ONE BYTE per (block,
// dimension): the segment's number of lines. Its start is the running sum of the bytes before
// it, which the wave forms itself: lane L loads the 16 bytes of dimensions 16 L .. 16 L + 15
// (one 16-byte load covers the whole row of a block, 7 lines for d = 800 instead of the ~18 of
// 25 a query's dimensions hit in the 4-byte table), byte sums with v_sad_u8, a wave prefix
// scan, and the lane that owns a query dimension pulls its 16-dimension group's prefix and words
// (ds_bpermute) and adds the bytes in front of its own. The chain is unchanged --
// acc = fmaf(q_d, x_d, acc) over ascending d with x_d = M 2^-22 (the 2^-22 is folded into the
// query value: a power of two, exact) -- so ids and score bits equal the oracle's over the same
// stored vectors. Measured on the bench library (scripts/flat_layout_model.py): 13 135 -> 8 980
// lines per query at nprobe 112, and the row loop loses one of its two loads.
template <int FI_CAP, bool FX, bool WIDE>
__global__ __launch_bounds__(FI_NT, FI_CAP <= 2048 ? 6 : 4) void flat_inv_scan_kernel(
    const float *__restrict__ xq, int d, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ list_offsets, const int32_t *__restrict__ blk_offsets,
    const uint32_t *__restrict__ blk_base, const uint32_t *__restrict__ seg_tab,
    const char *__restrict__ seg_bytes, const int32_t *__restrict__ ids, int k,
    float *__restrict__ D, int64_t *__restrict__ I64, int32_t *__restrict__ I32, int set_mode,
    const uint2 *__restrict__ ent, const int32_t *__restrict__ ent_cnt, int tab_stride,
    const int *__restrict__ gate, const ScanPostFilter pf) {
  if (gate && (int)blockIdx.x >= *gate) return;      // device-side row count (see pq_scan_v3_kernel)
  constexpr float FX_SCALE = FX ? 1.0f / 4194304.0f : 1.0f;      // 2^-22, folded into the query values
  using TopK = HistTopK<FI_CAP, FI_NT, FI_NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem + TopK::lds_bytes());      // [FI_NW][FI_BLK]
  float *s_nzv = s_acc + FI_NW * FI_BLK;                                   // [d]
  FiUnit *table = reinterpret_cast<FiUnit *>(s_nzv + ((d + 3) & ~3));      // [FI_CHUNK]
  int *s_misc = reinterpret_cast<int *>(table + FI_CHUNK);                 // [16]
  uint16_t *s_nzd = reinterpret_cast<uint16_t *>(s_misc + 16);             // [d rounded up to 8]
  uint16_t *s_nbv = s_nzd + ((d + 7) & ~7);                                // [FI_CHUNK] vectors per block
  volatile int *s_flag = s_misc + 9;      // a wave asks for a sync of the top-k
  int *s_done = s_misc + 10;              // waves that finished their blocks, summed over the chunks
  int *s_next = s_misc + 11;              // next block of the chunk to hand out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;

  // the query's non-zero components, ascending: from the ready entry list (list_nonzeros: 512
  // bytes) or listed here from the dense row (staged through the accumulator area)
  float *s_q = s_acc;
  int ecnt = ent ? ent_cnt[q] : -1;                // block-uniform; < 0: more than 64 non-zeros
  if (!xq && ecnt < 0) ecnt = 0;                   // (entry lists only: such a row is searched as all-zero)
  const bool fast = ecnt >= 0;
  if (!fast)
    for (int i = tid; i < d; i += FI_NT) s_q[i] = xq[(size_t)q * d + i];
  // a thread per probe (WIDE: probes tid and tid + FI_NT -- nprobe up to 1024, the reference's clamp,
  // spectral_library.py:77-81)
  constexpr int PP = WIDE ? 2 : 1;
  int my_len[PP], my_pos[PP], my_b0[PP], my_nb[PP], my_pre[PP];
#pragma unroll
  for (int pp = 0; pp < PP; ++pp) {
    my_len[pp] = 0, my_pos[pp] = 0, my_b0[pp] = 0, my_nb[pp] = 0;
    const int p = tid + pp * FI_NT;
    if (p < nprobe) {
      const int l = coarse_I[(size_t)q * nprobe + p];
      if (l >= 0) {
        my_pos[pp] = list_offsets[l];
        my_len[pp] = list_offsets[l + 1] - my_pos[pp];
        my_b0[pp] = blk_offsets[l];
        my_nb[pp] = blk_offsets[l + 1] - my_b0[pp];
      }
    }
  }
  int total = 0;
#pragma unroll
  for (int pp = 0; pp < PP; ++pp) {
    int part_total;
    if (pp) __syncthreads();      // the partial sums of the first scan have been read
    my_pre[pp] = total + block_excl_scan<FI_NW>(my_nb[pp], s_misc, tid, part_total);   // barrier inside: s_q complete
    total += part_total;
  }
  if (wave == 0 && fast) {
    if (lane < ecnt) {
      const uint2 e = ent[(size_t)q * 64 + lane];
      s_nzd[lane] = (uint16_t)(e.x >> 7);
      s_nzv[lane] = __uint_as_float(e.y) * FX_SCALE;
    }
    if (lane == 0) {
      s_misc[8] = ecnt;
      s_misc[9] = 0;      // sync flag
      s_misc[10] = 0;     // finished waves
    }
  } else if (wave == 0) {
    int base = 0;
    for (int j0 = 0; j0 < d; j0 += 64) {
      const int j = j0 + lane;
      const float x = j < d ? s_q[j] : 0.0f;
      const unsigned long long m = __ballot(x != 0.0f);
      if (x != 0.0f) {
        const int t = base + __popcll(m & ((1ull << lane) - 1ull));
        s_nzd[t] = (uint16_t)j;
        s_nzv[t] = x * FX_SCALE;
      }
      base += __popcll(m);
    }
    if (lane == 0) {
      s_misc[8] = base;
      s_misc[9] = 0;      // sync flag
      s_misc[10] = 0;     // finished waves
    }
  }
  __syncthreads();
  const int K = s_misc[8];
  TopK top;
  top.init(smem, k, ids, tid);
  top.out_keys = set_mode == 2 && I64 != nullptr;     // rows of packed keys (sharded exchange)
  float *acc = s_acc + wave * FI_BLK;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  int chunks_done = 0;
  auto sync = [&]() {             // raise the flag, meet the other waves, compact
    if (lane == 0) *s_flag = 1;
    __syncthreads();
    if (tid == 0) *s_flag = 0;
    top.free_sync();
  };
  auto sync_wanted = [&]() -> bool { return __builtin_amdgcn_readfirstlane(*s_flag) != 0; };
  for (int c0 = 0; c0 < total; c0 += FI_CHUNK) {
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
      const int lo = max(my_pre[pp], c0), hi = min(my_pre[pp] + my_nb[pp], c0 + FI_CHUNK);
      for (int t = lo; t < hi; ++t) {
        const int j = t - my_pre[pp];
        FiUnit u;
        u.blk = (uint32_t)(my_b0[pp] + j);
        u.pos0 = my_pos[pp] + j * FI_BLK;
        table[t - c0] = u;
        s_nbv[t - c0] = (uint16_t)min(FI_BLK, my_len[pp] - j * FI_BLK);
      }
    }
    __syncthreads();
    const int nent = min(FI_CHUNK, total - c0);
    // The waves take blocks on their own: the first FI_NW in wave order (the cold start below
    // needs every wave), then whichever is next (probe order: the lists closest to the query
    // first, they raise the threshold soonest).
    if (tid == 0) *s_next = FI_NW;
    __syncthreads();
    bool first = true;
    for (int i = wave;; first = false) {
      int nb = 0, pos0 = 0;
      if (i < nent) {          // wave-uniform
        const FiUnit u = table[i];
        nb = __builtin_amdgcn_readfirstlane((int)s_nbv[i]);
        pos0 = __builtin_amdgcn_readfirstlane(u.pos0);
        const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.blk);
        const char *bptr = seg_bytes + (size_t)blk_base[blk] * (FX ? 128 : 64);     // (a scalar load: blk is wave-uniform)
        for (int o = lane; o < nb; o += 64) acc[o] = 0.0f;
        const uint32_t *erow = seg_tab + (size_t)blk * d;
        // FX: the block's byte table (lines per dimension), 16 dimensions per lane, and the
        // lines in front of each group of 16
        uint4 tb = make_uint4(0u, 0u, 0u, 0u);
        uint32_t tpre = 0;
        if constexpr (FX) {
          const uint8_t *trow = reinterpret_cast<const uint8_t *>(seg_tab) + (size_t)blk * tab_stride;
          if (lane * 16 < tab_stride) tb = *reinterpret_cast<const uint4 *>(trow + lane * 16);
          const uint32_t mine = __builtin_amdgcn_sad_u8(tb.x, 0u, 0u) + __builtin_amdgcn_sad_u8(tb.y, 0u, 0u) +
                                __builtin_amdgcn_sad_u8(tb.z, 0u, 0u) + __builtin_amdgcn_sad_u8(tb.w, 0u, 0u);
          tpre = wave_incl_scan(mine) - mine;
        }
        for (int kk0 = 0; kk0 < K; kk0 += 64) {
          const int kk = kk0 + lane;
          uint2 e = make_uint2(0u, 0u);     // first byte (from the block's base), postings (FX: padded to whole lines)
          float qv = 0.0f;
          if constexpr (FX) {
            const int dim = kk < K ? (int)s_nzd[kk] : 0;
            const int src = (dim >> 4) << 2, b = dim & 15, wi = b >> 2, sh = (b & 3) * 8;
            const uint32_t gp = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tpre);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.x);
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.y);
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.z);
            const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.w);
            const uint32_t ws = wi == 0 ? w0 : wi == 1 ? w1 : wi == 2 ? w2 : w3;
            uint32_t before = __builtin_amdgcn_sad_u8(ws & ((1u << sh) - 1u), 0u, 0u);
            before += wi > 0 ? __builtin_amdgcn_sad_u8(w0, 0u, 0u) : 0u;
            before += wi > 1 ? __builtin_amdgcn_sad_u8(w1, 0u, 0u) : 0u;
            before += wi > 2 ? __builtin_amdgcn_sad_u8(w2, 0u, 0u) : 0u;
            if (kk < K) {
              e = make_uint2((gp + before) * 128u, ((ws >> sh) & 0xffu) * 32u);
              qv = s_nzv[kk];
            }
          } else if (kk < K) {
            const uint32_t w = erow[s_nzd[kk]];
            e = make_uint2((w >> 16) * 64u, w & 0xffffu);
            qv = s_nzv[kk];
          }
          // The unit of work is a ROW: up to 64 postings of one dimension. A dimension of a
          // 512-vector list has ~17 postings (one row), but a fragment bin that half the library
          // shares has hundreds -- a third of the work of a query that holds it -- so rows, not
          // dimensions, go through the pipeline. Lane j owns dimension j of the chunk (start e.x,
          // postings e.y, query value qv); rows are numbered in dimension order, and lane r
          // looks its row up: the dimension whose rows include r (binary search over the
          // prefix sums, ds_bpermute pulls), then start, count and query value of that
          // dimension. More than 64 rows in a chunk (dense data): further passes.
          const uint32_t rows_j = (e.y + 63u) >> 6;
          const uint32_t incl = wave_incl_scan(rows_j);
          const uint32_t pre = incl - rows_j;      // first row of my dimension
          const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          for (uint32_t r0 = 0; r0 < R; r0 += 64) {
            // my row; a lane past the last row takes the last row again with a zero query value
            // (fmaf(0, val, acc) == acc: the accumulators are never -0), so every row the
            // pipeline sees is a real one
            const bool live = r0 + lane < R;
            const uint32_t x = min(r0 + (uint32_t)lane, R - 1u);
            int lo = 0, hi = 63;                   // the last dimension with pre <= x
#pragma unroll
            for (int it = 0; it < 6; ++it) {
              const int mid = (lo + hi + 1) >> 1;
              const uint32_t pm = (uint32_t)__builtin_amdgcn_ds_bpermute(mid << 2, (int)pre);
              if (pm <= x) lo = mid; else hi = mid - 1;
            }
            const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)pre);
            const uint32_t stj = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)e.x);
            const uint32_t cnj = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)e.y);
            const float rq0 = __builtin_bit_cast(
                float, __builtin_amdgcn_ds_bpermute(lo << 2, __builtin_bit_cast(int, qv)));
            const float rq = live ? rq0 : 0.0f;
            const uint32_t t = x - pj;             // row inside the dimension
            const uint32_t rc1 = min(64u, cnj - 64u * t) - 1u;    // last lane of my row with a posting of its own
            const uint32_t rvo = stj + 256u * t;                  // its values (FX: its posting words) ...
            const uint32_t rlo = stj + 4u * cnj + 128u * t;       // ... and local indices
            const uint32_t rfx = rvo | (rc1 >> 5);                // FX: offset | (two lines wide)
            // Software pipeline over the rows, FI_U deep: row r + FI_U is requested as soon as
            // row r has been applied, so FI_U - 1 rows are always in flight (loads return in
            // order: the wait before applying r is "all but the 2 (FI_U - 1) youngest"). Loads
            // AND updates are unconditional: a lane without a posting of its own repeats the
            // row's last one -- same line, no traffic; it computes the same sum from the same
            // accumulator and stores the same bits -- so the row loop has neither an execution
            // mask nor a branch (4 of its ~20 instructions), nothing hides a load from the wait
            // counters, and the loop runs to a multiple of FI_U without a tail case. Rows of
            // one dimension touch different vectors; rows of different dimensions are applied in
            // dimension order: the canonical chain.
            // (FX: a row is one or two whole lines of posting words; rc1 is 31 or 63 -- a power of
            // two minus one, so the clamp is a mask -- and the lanes past a single line repeat it)
            uint32_t loc[FI_U];
            float qj[FI_U], val[FI_U];
#define FI_FETCH(u, r)                                                                            \
  {                                                                                               \
    qj[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rq), (r))); \
    if constexpr (FX) {                                                                           \
      /* one descriptor word: the row's byte offset (a multiple of 128) | 1 when it is two lines \
         wide; the offset goes into the scalar base, the width into the mask of my word index */  \
      const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)rfx, (r));                     \
      const char *row_ = bptr + (s_ & ~1u);                                                       \
      loc[u] = *reinterpret_cast<const uint32_t *>(row_ + (lane4 & ((s_ & 1u) ? 255u : 127u)));   \
    } else {                                                                                      \
      const uint32_t vo_ = (uint32_t)__builtin_amdgcn_readlane((int)rvo, (r));                    \
      const uint32_t c1_ = (uint32_t)__builtin_amdgcn_readlane((int)rc1, (r));                    \
      const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readlane((int)rlo, (r));                    \
      const uint32_t l_ = min((uint32_t)lane, c1_);                                               \
      val[u] = *reinterpret_cast<const float *>(bptr + (vo_ + 4u * l_));                          \
      loc[u] = *reinterpret_cast<const uint16_t *>(bptr + (lo_ + 2u * l_));                       \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);   /* the request stays here: FI_U - 1 rows in flight */     \
  }
#define FI_APPLY(u)                                                     \
  {                                                                     \
    if constexpr (FX) {                                                 \
      const uint32_t l_ = loc[u] & 1023u;                               \
      acc[l_] = __builtin_fmaf(qj[u], (float)(loc[u] >> 10), acc[l_]);  \
    } else {                                                            \
      acc[loc[u]] = __builtin_fmaf(qj[u], val[u], acc[loc[u]]);         \
    }                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                  \
  }
            const int n = (int)min(64u, R - r0);
            const int n_up = (n + FI_U - 1) & ~(FI_U - 1);
#pragma unroll
            for (int u = 0; u < FI_U; ++u) FI_FETCH(u, u)
            for (int j0 = FI_U; j0 < n_up; j0 += FI_U) {
#pragma unroll
              for (int u = 0; u < FI_U; ++u) {
                FI_APPLY(u)
                FI_FETCH(u, j0 + u)
              }
            }
#pragma unroll
            for (int u = 0; u < FI_U; ++u) FI_APPLY(u)
#undef FI_FETCH
#undef FI_APPLY
          }
        }
      }
      // the first blocks of a query: every accumulator of eight blocks would pass (4 k candidates
      // against a buffer of 2 k) -- the threshold is fixed from a histogram of all of them first
      const bool cold = c0 == 0 && first;     // the same for every wave
      if (cold) {
        for (int r = 0; r * 64 < nb; ++r) {
          const int v = r * 64 + lane;
          top.cold_count(v < nb, v < nb ? acc[v] : 0.0f);
        }
        top.cold_threshold();
      }
      // Offers: a row of accumulators at a time, straight into the key buffer (free_append: one
      // atomic per row that holds a candidate, no barrier). When the buffer is full the wave
      // raises the sync flag and waits; every wave looks at the flag between blocks (and in
      // the wait at the end of the chunk) and joins; then the row is tried again against the
      // new threshold.
      // A block past the first few of a query holds a handful of candidates: the wave reads
      // all its accumulators at once (13 rows, one LDS latency instead of 13), counts what
      // passes and, if that fits one row of slots, reserves it with ONE atomic instead of one
      // per row.
      bool offered = false;
      if (!cold && !top.sort_mode) {
        float sc[FI_ROWS];
#pragma unroll
        for (int r = 0; r < FI_ROWS; ++r) {
          const int v = r * 64 + lane;
          sc[r] = v < nb ? acc[v] : 0.0f;
        }
        int c = 0;
#pragma unroll
        for (int r = 0; r < FI_ROWS; ++r)
          c += __popcll(__ballot(r * 64 + lane < nb && top.passes(sc[r])));
        if (c == 0) {
          offered = true;
        } else if (c <= 64) {
          int base = top.free_reserve(c);
          if (base >= 0) {
#pragma unroll
            for (int r = 0; r < FI_ROWS; ++r) {
              const int v = r * 64 + lane;
              const bool p = v < nb && top.passes(sc[r]);
              const unsigned long long m = __ballot(p);
              if (m) {                                                     // wave-uniform
                top.free_write(p, m, sc[r], (uint32_t)(pos0 + v), base);
                base += __popcll(m);
              }
            }
            offered = true;
          }
        }
      }
      for (int r = 0; !offered && r * 64 < nb; ++r) {
        const int v = r * 64 + lane;
        const float score = v < nb ? acc[v] : 0.0f;
        for (;;) {
          const bool p = v < nb && top.passes(score);
          if (!__ballot(p)) break;                                        // wave-uniform
          if (top.free_append(p, score, (uint32_t)(pos0 + v), cold)) break;
          sync();
        }
      }
      if (sync_wanted()) sync();
      if (i >= nent) break;
      if (lane == 0) i = atomicAdd(s_next, 1);
      i = __builtin_amdgcn_readfirstlane(i);
      if (i >= nent) break;
    }
    // end of the chunk: wait for the other waves, joining the syncs they ask for
    if (lane == 0) atomicAdd(s_done, 1);
    ++chunks_done;
    for (;;) {
      if (sync_wanted()) {
        sync();
        continue;
      }
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >=
          FI_NW * chunks_done)
        break;
      __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
  }
  top.free_done();
  if (set_mode && (size_t)FI_CAP * 9 <= (size_t)FI_NW * FI_BLK * 4)   // unordered exact top-k; the accumulators are dead: scratch
    top.finish_set(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
                   I32 ? I32 + (size_t)q * k : nullptr, reinterpret_cast<u64 *>(s_acc), &pf, q);
  else if (set_mode && (size_t)FI_CAP * 8 <= (size_t)FI_NW * FI_BLK * 4)
    top.finish_set(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
                   I32 ? I32 + (size_t)q * k : nullptr, reinterpret_cast<u64 *>(s_acc));
  else
    top.finish(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
               I32 ? I32 + (size_t)q * k : nullptr);
}

bool flat_inv_supported(int d, int k, int nprobe) {
  return d <= 4096 && nprobe <= 2 * FI_NT && k >= 1 && k + FI_NT + 256 <= 4096;
}

template <int FI_CAP, bool FX, bool WIDE>
static int launch_flat_inv(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                           const int32_t *list_offsets, const int32_t *blk_offsets,
                           const uint32_t *blk_base, const uint32_t *seg_tab,
                           const char *seg_bytes, const int32_t *ids, int k, float *D,
                           int64_t *I64, int32_t *I32, int set_mode, const uint2 *ent,
                           const int32_t *ent_cnt, int tab_stride, const int *gate, const ScanPostFilter &pf) {
  using TopK = HistTopK<FI_CAP, FI_NT, FI_NT>;
  const size_t lds = TopK::lds_bytes() + (size_t)FI_NW * FI_BLK * 4 + (size_t)((d + 3) & ~3) * 4 +
                     (size_t)FI_CHUNK * (sizeof(FiUnit) + 2) + 64 + (size_t)((d + 7) & ~7) * 2;
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "flat scan: d=%d does not fit LDS", d);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)flat_inv_scan_kernel<FI_CAP, FX, WIDE>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((flat_inv_scan_kernel<FI_CAP, FX, WIDE>), dim3(nq), dim3(FI_NT), lds, stream(), xq, d,
                     coarse_I, nprobe, list_offsets, blk_offsets, blk_base, seg_tab, seg_bytes, ids,
                     k, D, I64, I32, set_mode, ent, ent_cnt, tab_stride, gate, pf);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// layout: 1 = float postings behind a 4-byte table word (seg_tab u32 [nblocks, d], blk_base in
// 64-byte units); 2 = fixed-point posting words behind a byte table (seg_tab = u8 [nblocks,
// tab_stride], blk_base in 128-byte lines)
int flat_inv_scan(int layout, const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                  const int32_t *list_offsets, const int32_t *blk_offsets,
                  const uint32_t *blk_base, const void *seg_tab, int tab_stride, const char *seg_bytes,
                  const int32_t *ids, int k, float *D, int64_t *I64, int32_t *I32, int set_mode,
                  const uint2 *ent, const int32_t *ent_cnt, const int *gate, const ScanPostFilter *post) {
  if (nq <= 0) return ASL_OK;
  const uint32_t *tab = reinterpret_cast<const uint32_t *>(seg_tab);
  const bool small = k + FI_NT + 256 <= 2048;
  ScanPostFilter pf;       // the set-mode finish of the 2048-key instantiation only (scratch behind the keys)
  if (post && post->idpay) {
    if (!(set_mode == 1 && I32 && small))
      return fail(ASL_ERR_STATE, "postings scan: a post-filter needs set-mode int32 rows and k <= 1280");
    pf = *post;
  }
#define FI_LAUNCH(CAP, FX)                                                                                 \
  do {                                                                                                     \
    if (nprobe > FI_NT)       /* two probes per thread (the one-probe form keeps its registers) */          \
      return launch_flat_inv<CAP, FX, true>(xq, nq, d, coarse_I, nprobe, list_offsets, blk_offsets, blk_base, \
                                            tab, seg_bytes, ids, k, D, I64, I32, set_mode, ent, ent_cnt,      \
                                            tab_stride, gate, pf);                                            \
    return launch_flat_inv<CAP, FX, false>(xq, nq, d, coarse_I, nprobe, list_offsets, blk_offsets, blk_base, \
                                           tab, seg_bytes, ids, k, D, I64, I32, set_mode, ent, ent_cnt,      \
                                           tab_stride, gate, pf);                                            \
  } while (0)
  if (layout == 2) {
    if (small) FI_LAUNCH(2048, true);
    FI_LAUNCH(4096, true);
  }
  if (small) FI_LAUNCH(2048, false);
  FI_LAUNCH(4096, false);
#undef FI_LAUNCH
}

// ---- algorithmic work of a postings scan (bench.py: the roofline of this kernel). Per (query,
// probed block, non-zero query dimension) the scan needs the 4-byte table word and the
// dimension's postings (6 bytes each): out[0] += 4 + 6 c. out[1] counts what that costs in
// 128-byte lines: the lines every non-empty segment spans plus the distinct lines of the
// block's table row that hold a wanted word. One workgroup per query, outside any timed region.
__global__ __launch_bounds__(256) void flat_inv_work_kernel(
    const float *__restrict__ xq, int d, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ blk_offsets, const uint32_t *__restrict__ seg_tab,
    unsigned long long *__restrict__ out) {
  extern __shared__ int s_work[];          // [d] non-zero dimensions, then 1 counter
  int *s_dim = s_work, *s_n = s_work + d;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) *s_n = 0;
  __syncthreads();
  for (int j = tid; j < d; j += 256)
    if (xq[(size_t)q * d + j] != 0.0f) s_dim[atomicAdd(s_n, 1)] = j;
  __syncthreads();
  const int K = *s_n;
  unsigned long long bytes = 0, lines = 0;
  for (int p = 0; p < nprobe; ++p) {
    const int l = coarse_I[(size_t)q * nprobe + p];
    if (l < 0) continue;
    for (int b = blk_offsets[l]; b < blk_offsets[l + 1]; ++b) {
      const uint32_t *row = seg_tab + (size_t)b * d;
      for (int t = tid; t < K; t += 256) {
        const uint32_t w = row[s_dim[t]];
        const uint32_t c = w & 0xffffu, st = (w >> 16) * 64u;
        bytes += 4ull + 6ull * c;
        if (c) lines += ((st + 6u * c - 1u) >> 7) - (st >> 7) + 1u;
      }
      // distinct table lines: one wave walks the query's dimensions in line order
      if (tid < 64) {
        for (int ln0 = 0; ln0 * 32 < d; ln0 += 64) {      // 32 words per 128-byte line
          const int ln = ln0 + lane;
          bool hit = false;
          for (int t = 0; t < K; ++t) hit |= (s_dim[t] >> 5) == ln;
          lines += __popcll(__ballot(hit)) * (lane == 0);
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bytes += __shfl_xor(bytes, off);
    lines += __shfl_xor(lines, off);
  }
  if (lane == 0) {
    atomicAdd(&out[0], bytes);
    atomicAdd(&out[1], lines);
  }
}

int flat_inv_work(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                  const int32_t *blk_offsets, const uint32_t *seg_tab, unsigned long long *out_dev) {
  if (nq <= 0) return ASL_OK;
  hipLaunchKernelGGL(flat_inv_work_kernel, dim3(nq), dim3(256), (size_t)(d + 1) * 4, stream(), xq, d,
                     coarse_I, nprobe, blk_offsets, seg_tab, out_dev);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// Placement of a block's segments (counts cnt[0..d) -> table words tab[0..d)); returns the
// block's size in 64-byte units, or 0 with *ok = false when a start does not fit 16 bits.
uint32_t inv_place_block(const uint32_t *cnt, int d, uint32_t *tab, bool *ok) {
  uint32_t pos = 0;     // 64-byte units; even = on a 128-byte line
  for (int j = 0; j < d; j++) {
    const uint32_t c = cnt[j];
    if (c == 0) {
      tab[j] = 0u;
      continue;
    }
    const uint32_t units = (6u * c + 63u) / 64u;
    if (units >= 2u && (pos & 1u)) ++pos;     // a full line or more: start on a line
    if (pos > 0xffffu || c > 0xffffu) {
      *ok = false;
      return 0;
    }
    tab[j] = (pos << 16) | c;
    pos += units;
  }
  return pos;
}

// ---- building the postings from dense rows. pos_blk / pos_loc: block and local index of
// the list-ordered position i (add-order row order[i]).
__global__ void inv_count_kernel(const float *__restrict__ vecs, int d,
                                 const int32_t *__restrict__ order,
                                 const int32_t *__restrict__ pos_blk, int64_t n,
                                 uint32_t *__restrict__ cnt) {
  const int64_t i = block_linear() * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float *row = vecs + (size_t)order[i] * d;
  uint32_t *c = cnt + (size_t)pos_blk[i] * d;
  for (int j = lane; j < d; j += 64)
    if (row[j] != 0.0f) atomicAdd(&c[j], 1u);
}

__global__ void inv_fill_kernel(const float *__restrict__ vecs, int d,
                                const int32_t *__restrict__ order,
                                const int32_t *__restrict__ pos_blk,
                                const uint16_t *__restrict__ pos_loc, int64_t n,
                                const uint32_t *__restrict__ blk_base,
                                const uint32_t *__restrict__ seg_tab,
                                uint32_t *__restrict__ cursor, char *__restrict__ seg_bytes) {
  const int64_t i = block_linear() * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float *row = vecs + (size_t)order[i] * d;
  const size_t b = (size_t)pos_blk[i] * d;
  char *bptr = seg_bytes + (size_t)blk_base[pos_blk[i]] * 64;
  const uint16_t loc = pos_loc[i];
  for (int j = lane; j < d; j += 64) {
    const float x = row[j];
    if (x != 0.0f) {
      const uint32_t w = seg_tab[b + j];
      const size_t st = (size_t)(w >> 16) * 64;
      const uint32_t c = w & 0xffffu;
      const uint32_t p = atomicAdd(&cursor[b + j], 1u);     // any order inside a segment
      *reinterpret_cast<float *>(bptr + st + 4 * (size_t)p) = x;
      *reinterpret_cast<uint16_t *>(bptr + st + 4 * (size_t)c + 2 * (size_t)p) = loc;
    }
  }
}

// ---- canonical order inside every segment. inv_fill_kernel leaves the postings of a segment
// in the order its atomics happened to run; a vector occurs at most once per dimension, so ANY
// order gives the same accumulators. This pass (one wave per segment) makes the layout
// deterministic -- postings sorted by local index -- and, for segments longer than a row, deals
// each row of 64 to its first and second half (the LDS services lanes {0-31} and {32-63} of a
// ds instruction separately) so that the banks (local index mod 32) inside a half are distinct
// where the row allows it. Measured (profiles/r03_flat_scan_notes.txt): bank conflicts are NOT
// what this kernel waits for -- a layout without any conflict runs at the same speed -- the
// sorted order is worth 1 %; it is kept for the reproducible file image.
constexpr int IO_WAVES = 4;
__global__ __launch_bounds__(64 * IO_WAVES) void inv_order_kernel(
    int64_t nseg, int d, const uint32_t *__restrict__ blk_base,
    const uint32_t *__restrict__ seg_tab, char *__restrict__ seg_bytes) {
  __shared__ float s_val[IO_WAVES][2][FI_BLK];
  __shared__ uint16_t s_loc[IO_WAVES][2][FI_BLK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t seg = (int64_t)blockIdx.x * IO_WAVES + wave;
  if (seg >= nseg) return;
  const uint32_t w = seg_tab[seg];
  const int c = (int)(w & 0xffffu);
  if (c <= 1 || c > FI_BLK) return;
  char *ptr = seg_bytes + (size_t)blk_base[seg / d] * 64 + (size_t)(w >> 16) * 64;
  float *gv = reinterpret_cast<float *>(ptr);
  uint16_t *gl = reinterpret_cast<uint16_t *>(ptr + 4 * (size_t)c);
  float *v0 = s_val[wave][0], *v1 = s_val[wave][1];
  uint16_t *l0 = s_loc[wave][0], *l1 = s_loc[wave][1];
  for (int i = lane; i < c; i += 64) {
    v0[i] = gv[i];
    l0[i] = gl[i];
  }
  __builtin_amdgcn_wave_barrier();
  // rank sort by local index (unique inside a segment)
  for (int i = lane; i < c; i += 64) {
    const uint16_t me = l0[i];
    int r = 0;
    for (int t = 0; t < c; ++t) r += l0[t] < me;
    v1[r] = v0[i];
    l1[r] = me;
  }
  __builtin_amdgcn_wave_barrier();
  // greedy deal, serial per row (c is ~30; a shared fragment bin has a few hundred)
  if (lane == 0) {
    for (int r0 = 0; r0 < c; r0 += 64) {
      const int n = min(64, c - r0);
      const int cap0 = min(n, 32), cap1 = n - cap0;      // lanes 0-31 | lanes 32-63 of the row
      uint32_t used0 = 0, used1 = 0;
      int n0 = 0, n1 = 0;
      for (int t = 0; t < n; ++t) {
        const uint32_t bit = 1u << (l1[r0 + t] & 31);
        int g;
        if (!(used0 & bit) && n0 < cap0) g = 0;
        else if (!(used1 & bit) && n1 < cap1) g = 1;
        else g = (n0 < cap0 && (n1 >= cap1 || n0 - cap0 <= n1 - cap1)) ? 0 : 1;
        int dst;
        if (g == 0) {
          used0 |= bit;
          dst = r0 + n0++;
        } else {
          used1 |= bit;
          dst = r0 + cap0 + n1++;
        }
        v0[dst] = v1[r0 + t];
        l0[dst] = l1[r0 + t];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < c; i += 64) {
    gv[i] = v0[i];
    gl[i] = l0[i];
  }
}

int inv_order(int64_t nblocks, int d, const uint32_t *blk_base, const uint32_t *seg_tab,
              char *seg_bytes) {
  const int64_t nseg = nblocks * d;
  if (nseg <= 0) return ASL_OK;
  hipLaunchKernelGGL(inv_order_kernel, dim3((unsigned)cdiv(nseg, IO_WAVES)), dim3(64 * IO_WAVES), 0,
                     stream(), nseg, d, blk_base, seg_tab, seg_bytes);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int inv_count(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk, int64_t n,
              uint32_t *cnt) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(inv_count_kernel, grid_2d(cdiv(n, 4)), dim3(256), 0, stream(), vecs, d,
                     order, pos_blk, n, cnt);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int inv_fill(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk,
             const uint16_t *pos_loc, int64_t n, const uint32_t *blk_base,
             const uint32_t *seg_tab, uint32_t *cursor, char *seg_bytes) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(inv_fill_kernel, grid_2d(cdiv(n, 4)), dim3(256), 0, stream(), vecs, d,
                     order, pos_blk, pos_loc, n, blk_base, seg_tab, cursor, seg_bytes);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// ---- the fixed-point layout (flat_inv_scan_kernel<.., true>): work accounting and builders.
// Algorithmic bytes per (query, probed block, non-zero query dimension): the table byte + 4 bytes
// per posting (cnt16: the real number of postings of every cell, kept for this accounting only);
// lines: the block's table row (tab_stride / 128, read whole) + the lines of the wanted segments.
__global__ __launch_bounds__(256) void flat_fx_work_kernel(
    const float *__restrict__ xq, int d, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ blk_offsets, const uint8_t *__restrict__ tab8, int tab_stride,
    const uint16_t *__restrict__ cnt16, unsigned long long *__restrict__ out) {
  extern __shared__ int s_work[];          // [d] non-zero dimensions, then 1 counter
  int *s_dim = s_work, *s_n = s_work + d;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) *s_n = 0;
  __syncthreads();
  for (int j = tid; j < d; j += 256)
    if (xq[(size_t)q * d + j] != 0.0f) s_dim[atomicAdd(s_n, 1)] = j;
  __syncthreads();
  const int K = *s_n;
  unsigned long long bytes = 0, lines = 0;
  for (int p = 0; p < nprobe; ++p) {
    const int l = coarse_I[(size_t)q * nprobe + p];
    if (l < 0) continue;
    for (int b = blk_offsets[l]; b < blk_offsets[l + 1]; ++b) {
      for (int t = tid; t < K; t += 256) {
        bytes += 1ull + 4ull * cnt16[(size_t)b * d + s_dim[t]];
        lines += tab8[(size_t)b * tab_stride + s_dim[t]];
      }
      if (tid == 0) lines += (unsigned long long)((d + 127) / 128);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bytes += __shfl_xor(bytes, off);
    lines += __shfl_xor(lines, off);
  }
  if (lane == 0) {
    atomicAdd(&out[0], bytes);
    atomicAdd(&out[1], lines);
  }
}

int flat_fx_work(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                 const int32_t *blk_offsets, const uint8_t *tab8, int tab_stride,
                 const uint16_t *cnt16, unsigned long long *out_dev) {
  if (nq <= 0) return ASL_OK;
  hipLaunchKernelGGL(flat_fx_work_kernel, dim3(nq), dim3(256), (size_t)(d + 1) * 4, stream(), xq, d,
                     coarse_I, nprobe, blk_offsets, tab8, tab_stride, cnt16, out_dev);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// posting words into their segments, in the order the atomics run (fx_order_kernel sorts them).
// seg_line[cell]: first line of the cell's segment, counted from the start of the data.
__global__ void fx_fill_kernel(const float *__restrict__ vecs, int d,
                               const int32_t *__restrict__ order,
                               const int32_t *__restrict__ pos_blk,
                               const uint16_t *__restrict__ pos_loc, int64_t n,
                               const uint32_t *__restrict__ seg_line,
                               uint32_t *__restrict__ cursor, uint32_t *__restrict__ words) {
  const int64_t i = block_linear() * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float *row = vecs + (size_t)order[i] * d;
  const size_t b = (size_t)pos_blk[i] * d;
  const uint32_t loc = pos_loc[i];
  for (int j = lane; j < d; j += 64) {
    const float x = row[j];
    if (x != 0.0f) {
      const uint32_t p = atomicAdd(&cursor[b + j], 1u);
      words[(size_t)seg_line[b + j] * 32 + p] = ((uint32_t)(x * 4194304.0f) << 10) | loc;   // exact: x is on the grid
    }
  }
}

// canonical image of every segment: postings ascending in the local index (unique inside a
// segment), the rest of the last line filled with repeats of the last posting. One wave per cell.
__global__ __launch_bounds__(64 * IO_WAVES) void fx_order_kernel(
    int64_t ncell, const uint32_t *__restrict__ seg_line, const uint32_t *__restrict__ count,
    uint32_t *__restrict__ words) {
  __shared__ uint32_t s_w[IO_WAVES][2][FI_BLK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t cell = (int64_t)blockIdx.x * IO_WAVES + wave;
  if (cell >= ncell) return;
  const int c = (int)count[cell];
  if (c <= 0 || c > FI_BLK) return;
  uint32_t *g = words + (size_t)seg_line[cell] * 32;
  uint32_t *a = s_w[wave][0], *b = s_w[wave][1];
  for (int i = lane; i < c; i += 64) a[i] = g[i];
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < c; i += 64) {
    const uint32_t me = a[i] & 1023u;
    int r = 0;
    for (int t = 0; t < c; ++t) r += (a[t] & 1023u) < me;
    b[r] = a[i];
  }
  __builtin_amdgcn_wave_barrier();
  const int padded = (c + 31) & ~31;
  for (int i = lane; i < padded; i += 64) g[i] = b[min(i, c - 1)];
}

int fx_fill(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk,
            const uint16_t *pos_loc, int64_t n, const uint32_t *seg_line, uint32_t *cursor,
            uint32_t *words) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(fx_fill_kernel, grid_2d(cdiv(n, 4)), dim3(256), 0, stream(), vecs, d, order,
                     pos_blk, pos_loc, n, seg_line, cursor, words);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int fx_order(int64_t ncell, const uint32_t *seg_line, const uint32_t *count, uint32_t *words) {
  if (ncell <= 0) return ASL_OK;
  hipLaunchKernelGGL(fx_order_kernel, dim3((unsigned)cdiv(ncell, IO_WAVES)), dim3(64 * IO_WAVES), 0,
                     stream(), ncell, seg_line, count, words);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

}  // namespace asl
