// flat_scan_q.hip -- the IVF-Flat postings scan with FOUR BLOCK STREAMS PER WAVE (round 5; the
// scan inside FAISS IndexIVFFlat.search, /root/reference/src/ann_solo/spectral_library.py:443-444).
//
// flat_inv_scan_kernel (flat_scan.hip) gives a wave one (query, block) at a time; the unit of
// work is a row of 64 lanes for the postings of ONE (block, dimension) cell, and a cell of a
// ~440-vector block holds ~14 postings: 23 % of the lanes work, the kernel issues 133 k VALU
// wave-instructions per query (10 per row, ~470 of bookkeeping per block) and sits at 89 % of the
// SIMD cycles (profiles/r04_ivfflat_np112_pmc_summary.txt). Here a wave takes up to FOUR blocks
// at once, one per QUARTER of its lanes:
//
//   * different blocks never share an accumulator, so the four quarters of a row never collide
//     and no ordering hazard exists between them; inside a quarter the rows run in ascending
//     dimension, as before: per vector still the ascending-dimension fp32 fmaf chain over the
//     stored non-zeros -- ids and score bits equal the oracle's, flat_inv_scan_kernel's and the
//     dense GEMM's;
//   * a quarter-row is 16 posting words = half a 128-byte line of the fixed-point layout; the
//     table byte of a cell now also says whether the second half of its last line holds repeats
//     only (bit 7), so a ~14-posting cell costs ONE quarter-row at 88 % of its lanes;
//   * the rows of a quarter are not looked up by binary search any more: the lane that owns a
//     query dimension WRITES the descriptors of its cell's quarter-rows (byte offset, query
//     value: 8 bytes) into a per-quarter list in LDS, and the row loop reads one descriptor per
//     quarter per row -- one LDS broadcast read instead of four v_readlane, one add for the
//     address, five VALU for the update.
//
// Budget per group of four blocks (~2 450 postings): table + look-up 4 x ~45, descriptor lists
// 4 x ~25, ~65 rows x 6, offers 4 x ~60 VALU: ~1 000 against ~3 960 for four blocks before.
#include <cstdlib>

#include "common.hpp"
#include "hist_topk.hpp"
#include "ivf_kernels.hpp"

namespace asl {

constexpr int FQ_NW = 4, FQ_NT = 64 * FQ_NW, FQ_G = 4, FQ_CHUNK = 256;
constexpr int FQ_ROWS = 96;                     // descriptor rows per quarter and window
constexpr int FQ_D = 8;                         // rows per batch of the software pipeline (two batches in flight)
constexpr int FQ_ACC = 2176;                    // accumulators per wave (>= 2 x FI_BLK): ~4 average blocks
constexpr int FQ_DSTRIDE = (FQ_ROWS + 2 * FQ_D) * 8 + 16;   // bytes between the quarters' lists: two batches of idle
                                                           // rows behind the window (the pipeline reads ahead), +16: other LDS banks
constexpr int FQ_ROWS_BLK = (FI_BLK + 63) / 64;
static_assert(FQ_ROWS % (2 * FQ_D) == 0 && FQ_ACC >= FI_BLK, "");

struct FqUnit {
  uint32_t blk;   // block index into the per-dimension table
  int32_t pos0;   // list-order position of the block's first vector
};

// phase timers of the measurement build (ASL_FLAT_Q=2): wave-cycles per phase, summed over all waves
__device__ unsigned long long g_fq_prof[16];
#define FQ_T(i)                                                   \
  if constexpr (PROF) {                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();   \
    prof[i] += t_ - t_last;                                       \
    t_last = t_;                                                  \
  }

template <int CAP, bool PROF>
__global__ __launch_bounds__(FQ_NT, 2) void flat_q_scan_kernel(
    const float *__restrict__ xq, int d, const int32_t *__restrict__ coarse_I, int nprobe,
    const int32_t *__restrict__ list_offsets, const int32_t *__restrict__ blk_offsets,
    const uint32_t *__restrict__ blk_base, const uint8_t *__restrict__ tab8, int tab_stride,
    const char *__restrict__ seg_bytes, const int32_t *__restrict__ ids, int k,
    float *__restrict__ D, int64_t *__restrict__ I64, int32_t *__restrict__ I32, int set_mode,
    const uint2 *__restrict__ ent, const int32_t *__restrict__ ent_cnt, const int *__restrict__ gate) {
  if (gate && (int)blockIdx.x >= *gate) return;
  unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = PROF ? __builtin_readcyclecounter() : 0ull;
  constexpr float FX_SCALE = 1.0f / 4194304.0f;      // 2^-22, folded into the query values
  using TopK = HistTopK<CAP, FQ_NT, FQ_NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_acc = reinterpret_cast<float *>(smem + TopK::lds_bytes());      // [FQ_NW][FQ_ACC]
  char *s_desc = reinterpret_cast<char *>(s_acc + FQ_NW * FQ_ACC);         // [FQ_NW][FQ_G] lists of FQ_DSTRIDE bytes
  float *s_nzv = reinterpret_cast<float *>(s_desc + FQ_NW * FQ_G * FQ_DSTRIDE);   // [d]
  FqUnit *table = reinterpret_cast<FqUnit *>(s_nzv + ((d + 3) & ~3));      // [FQ_CHUNK]
  int *s_misc = reinterpret_cast<int *>(table + FQ_CHUNK);                 // [16]
  uint16_t *s_nzd = reinterpret_cast<uint16_t *>(s_misc + 16);             // [d rounded up to 8]
  uint16_t *s_nbv = s_nzd + ((d + 7) & ~7);                                // [FQ_CHUNK] vectors per block
  volatile int *s_flag = s_misc + 9;      // a wave asks for a sync of the top-k
  int *s_done = s_misc + 10;              // waves that finished their blocks, summed over the chunks
  int *s_next = s_misc + 11;              // next block of the chunk to hand out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;

  // ---- the query's non-zero components, ascending (as flat_inv_scan_kernel)
  float *s_q = s_acc;
  const int ecnt = ent ? ent_cnt[q] : -1;          // block-uniform; < 0: more than 64 non-zeros
  const bool fast = ecnt >= 0;
  if (!fast)
    for (int i = tid; i < d; i += FQ_NT) s_q[i] = xq[(size_t)q * d + i];
  int my_len = 0, my_pos = 0, my_b0 = 0, my_nb = 0;
  if (tid < nprobe) {
    const int l = coarse_I[(size_t)q * nprobe + tid];
    if (l >= 0) {
      my_pos = list_offsets[l];
      my_len = list_offsets[l + 1] - my_pos;
      my_b0 = blk_offsets[l];
      my_nb = blk_offsets[l + 1] - my_b0;
    }
  }
  int total;
  const int my_pre = block_excl_scan<FQ_NW>(my_nb, s_misc, tid, total);   // barrier inside: s_q complete
  if (wave == 0 && fast) {
    if (lane < ecnt) {
      const uint2 e = ent[(size_t)q * 64 + lane];
      s_nzd[lane] = (uint16_t)(e.x >> 7);
      s_nzv[lane] = __uint_as_float(e.y) * FX_SCALE;
    }
    if (lane == 0) {
      s_misc[8] = ecnt;
      s_misc[9] = 0;
      s_misc[10] = 0;
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
      s_misc[9] = 0;
      s_misc[10] = 0;
    }
  }
  __syncthreads();
  const int K = s_misc[8];
  TopK top;
  top.init(smem, k, ids, tid);
  top.out_keys = set_mode == 2 && I64 != nullptr;     // rows of packed keys (sharded exchange)
  float *acc = s_acc + wave * FQ_ACC;
  const int quarter = lane >> 4;                  // my block of the group = my quarter of the lanes
  const uint32_t lane16x4 = (uint32_t)(lane & 15) * 4u;
  char *dlist = s_desc + (size_t)(wave * FQ_G + quarter) * FQ_DSTRIDE;       // my quarter's descriptor list
  int chunks_done = 0;
  auto sync = [&]() {             // raise the flag, meet the other waves, compact
    unsigned long long t0_ = 0;
    if constexpr (PROF) t0_ = __builtin_readcyclecounter();
    if (lane == 0) *s_flag = 1;
    __syncthreads();
    if (tid == 0) *s_flag = 0;
    top.free_sync();
    if constexpr (PROF) {
      const unsigned long long t1_ = __builtin_readcyclecounter();
      prof[8] += t1_ - t0_;
      t_last += t1_ - t0_;        // (not charged to the phase the sync interrupted)
    }
  };
  auto sync_wanted = [&]() -> bool { return __builtin_amdgcn_readfirstlane(*s_flag) != 0; };

  FQ_T(0)
  for (int c0 = 0; c0 < total; c0 += FQ_CHUNK) {
    {
      const int lo = max(my_pre, c0), hi = min(my_pre + my_nb, c0 + FQ_CHUNK);
      for (int t = lo; t < hi; ++t) {
        const int j = t - my_pre;
        FqUnit u;
        u.blk = (uint32_t)(my_b0 + j);
        u.pos0 = my_pos + j * FI_BLK;
        table[t - c0] = u;
        s_nbv[t - c0] = (uint16_t)min(FI_BLK, my_len - j * FI_BLK);
      }
    }
    if (tid == 0) *s_next = 0;
    __syncthreads();
    const int nent = min(FQ_CHUNK, total - c0);
    int pend = -1;              // a block this wave took that did not fit its accumulators any more
    bool first = true;
    for (;; first = false) {
      // ---- take up to four blocks (probe order: the lists closest to the query first)
      int g_nb[FQ_G], g_pos0[FQ_G], g_acc[FQ_G];
      uint32_t g_blk[FQ_G];
      int ng = 0, used = 0;
      bool exhausted = false;
      // towards the end of the chunk the groups shrink (4, 2, 1 blocks), so that the waves finish together
      const int left = nent - __builtin_amdgcn_readfirstlane(*(volatile int *)s_next);
      const int gmax = left > 4 * FQ_NW ? 4 : left > 2 * FQ_NW ? 2 : 1;
#pragma unroll
      for (int g = 0; g < FQ_G; ++g) {
        g_nb[g] = 0;
        g_pos0[g] = 0;
        g_acc[g] = 0;
        g_blk[g] = 0;
        if (!exhausted && ng == g && g < gmax) {            // wave-uniform
          int i = pend;
          if (i < 0) {
            if (lane == 0) i = atomicAdd(s_next, 1);
            i = __builtin_amdgcn_readfirstlane(i);
          }
          pend = -1;
          if (i >= nent) {
            exhausted = true;
          } else {
            const int nb = __builtin_amdgcn_readfirstlane((int)s_nbv[i]);
            if (used + nb > FQ_ACC) {
              pend = i;
              exhausted = true;                 // (for this group)
            } else {
              const FqUnit u = table[i];
              g_nb[g] = nb;
              g_pos0[g] = __builtin_amdgcn_readfirstlane(u.pos0);
              g_blk[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.blk);
              g_acc[g] = used;
              used += nb;
              ++ng;
            }
          }
        }
      }
      const bool cold = c0 == 0 && first;     // the same for every wave: every wave runs its first group
      FQ_T(1)
      if (ng == 0 && !cold) break;
      for (int o = lane; o < used; o += 64) acc[o] = 0.0f;
      FQ_T(2)
      // my quarter's block
      const uint32_t my_accb = (uint32_t)((quarter == 0 ? g_acc[0] : quarter == 1 ? g_acc[1] : quarter == 2 ? g_acc[2] : g_acc[3]) * 4 +
                                          wave * FQ_ACC * 4) +
                               (uint32_t)(reinterpret_cast<char *>(s_acc) - smem);      // byte offset inside smem
      const bool q_on = quarter < ng;
      for (int kk0 = 0; kk0 < K && ng > 0; kk0 += 64) {
        const int kk = kk0 + lane;
        const bool have = kk < K;
        const int dim = have ? (int)s_nzd[kk] : 0;
        const float qv = have ? s_nzv[kk] : 0.0f;
        const int src = (dim >> 4) << 2, b = dim & 15, wi = b >> 2, sh = (b & 3) * 8;
        // per block: first byte of my dimension's cell and its quarter-rows
        uint32_t c_off[FQ_G], c_nq[FQ_G], c_pre[FQ_G], c_tot[FQ_G];
        int rmax = 0;
        // the four table rows first (one 16-byte load per lane and block: four round trips in flight
        // instead of one after the other -- at two waves per SIMD nobody else hides them)
        uint4 tbg[FQ_G];
        uint32_t bbase[FQ_G];
#pragma unroll
        for (int g = 0; g < FQ_G; ++g) {
          const uint8_t *trow = tab8 + (size_t)g_blk[g] * tab_stride;     // (g >= ng: block 0's row, unused)
          tbg[g] = make_uint4(0u, 0u, 0u, 0u);
          if (lane * 16 < tab_stride) tbg[g] = *reinterpret_cast<const uint4 *>(trow + lane * 16);
          bbase[g] = blk_base[g_blk[g]];
        }
#pragma unroll
        for (int g = 0; g < FQ_G; ++g) {
          c_off[g] = 0u;
          c_nq[g] = 0u;
          c_pre[g] = 0u;
          c_tot[g] = 0u;
          if (g < ng) {                         // wave-uniform
            const uint4 tb = tbg[g];
            const uint32_t m7 = 0x7f7f7f7fu;
            const uint32_t mine = __builtin_amdgcn_sad_u8(tb.x & m7, 0u, 0u) + __builtin_amdgcn_sad_u8(tb.y & m7, 0u, 0u) +
                                  __builtin_amdgcn_sad_u8(tb.z & m7, 0u, 0u) + __builtin_amdgcn_sad_u8(tb.w & m7, 0u, 0u);
            const uint32_t tpre = wave_incl_scan(mine) - mine;
            const uint32_t gp = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tpre);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.x);
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.y);
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.z);
            const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tb.w);
            const uint32_t ws = wi == 0 ? w0 : wi == 1 ? w1 : wi == 2 ? w2 : w3;
            uint32_t before = __builtin_amdgcn_sad_u8(ws & m7 & ((1u << sh) - 1u), 0u, 0u);
            before += wi > 0 ? __builtin_amdgcn_sad_u8(w0 & m7, 0u, 0u) : 0u;
            before += wi > 1 ? __builtin_amdgcn_sad_u8(w1 & m7, 0u, 0u) : 0u;
            before += wi > 2 ? __builtin_amdgcn_sad_u8(w2 & m7, 0u, 0u) : 0u;
            const uint32_t byte = (ws >> sh) & 0xffu, nl = byte & 0x7fu;
            if (have && nl) {
              c_off[g] = (bbase[g] + gp + before) * 128u;
              c_nq[g] = 2u * nl - (byte >> 7);
            }
            const uint32_t incl = wave_incl_scan(c_nq[g]);
            c_pre[g] = incl - c_nq[g];
            c_tot[g] = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            rmax = max(rmax, (int)c_tot[g]);
          }
        }
        FQ_T(3)
        // ---- windows of FQ_ROWS rows (one, unless a block holds very long cells)
        for (int w0r = 0; w0r < rmax; w0r += FQ_ROWS) {
          const int nrows = min(FQ_ROWS, rmax - w0r);
          const int npad = (nrows + 2 * FQ_D - 1) & ~(2 * FQ_D - 1);      // whole double batches
#pragma unroll
          for (int g = 0; g < FQ_G; ++g) {
            char *lst = s_desc + (size_t)(wave * FQ_G + g) * FQ_DSTRIDE;
            // rows past the block's last one (and idle quarters): a valid address, query value 0
            const uint32_t idle_off = g < ng ? bbase[g] * 128u : 0u;
            // (a quarter whose rows go on in the next window still needs valid read-ahead rows behind this one)
            const int tail0 = min(FQ_ROWS, max(0, (int)c_tot[g] - w0r));
            for (int r = tail0 + lane; r < npad + 2 * FQ_D; r += 64) *reinterpret_cast<uint2 *>(lst + r * 8) = make_uint2(idle_off, 0u);
            if (g < ng) {
              // cells of one or two quarter-rows: written by their own lane
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int r = (int)c_pre[g] + t - w0r;
                if ((uint32_t)t < c_nq[g] && (unsigned)r < (unsigned)FQ_ROWS)
                  *reinterpret_cast<uint2 *>(lst + r * 8) = make_uint2(c_off[g] + 64u * t, __float_as_uint(qv));
              }
              // longer cells (a fragment bin that half of the library shares): the whole wave writes
              unsigned long long m = __ballot(c_nq[g] > 2u);
              while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1ull;
                const int pj = __builtin_amdgcn_readlane((int)c_pre[g], j) - w0r;
                const int nj = __builtin_amdgcn_readlane((int)c_nq[g], j);
                const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)c_off[g], j);
                const uint32_t qj = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qv), j);
                for (int t = 2 + lane; t < nj; t += 64) {
                  const int r = pj + t;
                  if ((unsigned)r < (unsigned)FQ_ROWS) *reinterpret_cast<uint2 *>(lst + r * 8) = make_uint2(oj + 64u * t, qj);
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();      // the lists are complete before the row loop reads them
          FQ_T(4)
          // ---- the row loop: per row one descriptor per quarter, 16 posting words per quarter.
          // Two batches of FQ_D rows in flight: the posting loads of batch b + 1 are issued before
          // batch b is applied, the descriptors of batch b + 2 are read meanwhile.
          uint32_t w[2][FQ_D];
          float qvr[2][FQ_D];
          uint2 nd[FQ_D];
  /* (the sched_barriers pin the requests where they are written: left alone the compiler sinks  \
     the posting loads of the next batch below the updates of this one, and every batch then waits \
     out a full memory round trip: 10.8 ms instead of 4.2 in the first build) */                  \
#define FQ_READ_DESC(r0)                                                                        \
  {                                                                                             \
    _Pragma("unroll") for (int u = 0; u < FQ_D; ++u)                                            \
        nd[u] = *reinterpret_cast<const uint2 *>(dlist + ((r0) + u) * 8);                       \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }
#define FQ_ISSUE(p)                                                                             \
  {                                                                                             \
    _Pragma("unroll") for (int u = 0; u < FQ_D; ++u) {                                          \
      qvr[p][u] = __uint_as_float(nd[u].y);                                                     \
      w[p][u] = *reinterpret_cast<const uint32_t *>(seg_bytes + (size_t)(nd[u].x + lane16x4));  \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }
  /* a quarter without a block stays out of the updates (its rows would read-modify-write    \
     accumulators another quarter owns); a quarter past ITS block's last row repeats that block's \
     first line with a zero query value: fmaf(0, val, acc) == acc, nobody else touches that block */ \
#define FQ_APPLY(p)                                                                             \
  if (q_on) {                                                                                   \
    _Pragma("unroll") for (int u = 0; u < FQ_D; ++u) {                                          \
      float *a_ = reinterpret_cast<float *>(smem + my_accb + (w[p][u] & 1023u) * 4u);           \
      *a_ = __builtin_fmaf(qvr[p][u], (float)(w[p][u] >> 10), *a_);                             \
      __builtin_amdgcn_sched_barrier(0);                                                        \
    }                                                                                           \
  }
          // (branch-free: npad is a whole number of double batches and two batches of idle rows lie
          // behind it, so every request below is valid -- with a branch around a request the compiler's
          // wait counters fall back to "everything outstanding" and the pipeline is gone)
          FQ_READ_DESC(0)
          FQ_ISSUE(0)
          FQ_READ_DESC(FQ_D)
          for (int r0 = 0; r0 < npad; r0 += 2 * FQ_D) {
            // a wave that found the key buffer full waits for everybody: look at the flag every 16
            // rows (a group of four blocks runs for tens of microseconds -- polling only between
            // groups left the other three waves waiting that long, several times per query)
            if (sync_wanted()) sync();
            FQ_ISSUE(1)                         // rows r0 + D ..: in flight under the updates of r0 ..
            FQ_READ_DESC(r0 + 2 * FQ_D)
            FQ_APPLY(0)
            FQ_ISSUE(0)                         // rows r0 + 2 D .. (idle rows after the last double batch: never applied)
            FQ_READ_DESC(r0 + 3 * FQ_D)
            FQ_APPLY(1)
          }
#undef FQ_READ_DESC
#undef FQ_ISSUE
#undef FQ_APPLY
          __builtin_amdgcn_wave_barrier();      // the lists are free again
          FQ_T(5)
        }
      }
      // ---- offers, block by block (as flat_inv_scan_kernel: histogram cold start for a query's first
      // blocks, then free-running appends; a full key buffer brings the waves together)
      if (cold) {
#pragma unroll
        for (int g = 0; g < FQ_G; ++g)
          for (int r = 0; r * 64 < g_nb[g]; ++r) {
            const int v = r * 64 + lane;
            top.cold_count(v < g_nb[g], v < g_nb[g] ? acc[g_acc[g] + v] : 0.0f);
          }
        top.cold_threshold();
        FQ_T(6)
      }
#pragma unroll
      for (int g = 0; g < FQ_G; ++g) {
        const int nb = g_nb[g], pos0 = g_pos0[g];
        const float *ag = acc + g_acc[g];
        if (nb == 0) continue;                  // wave-uniform
        bool offered = false;
        if (!cold && !top.sort_mode) {
          float sc[FQ_ROWS_BLK];
#pragma unroll
          for (int r = 0; r < FQ_ROWS_BLK; ++r) {
            const int v = r * 64 + lane;
            sc[r] = v < nb ? ag[v] : 0.0f;
          }
          int c = 0;
#pragma unroll
          for (int r = 0; r < FQ_ROWS_BLK; ++r)
            c += __popcll(__ballot(r * 64 + lane < nb && top.passes(sc[r])));
          if (c == 0) {
            offered = true;
          } else if (c <= 64) {
            int base = top.free_reserve(c);
            if (base >= 0) {
#pragma unroll
              for (int r = 0; r < FQ_ROWS_BLK; ++r) {
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
          const float score = v < nb ? ag[v] : 0.0f;
          for (;;) {
            const bool p = v < nb && top.passes(score);
            if (!__ballot(p)) break;                                        // wave-uniform
            if (top.free_append(p, score, (uint32_t)(pos0 + v), cold)) break;
            sync();
          }
        }
        if (sync_wanted()) sync();
      }
      if (sync_wanted()) sync();
      FQ_T(7)
      if (ng == 0) break;           // (a cold first group without blocks)
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
          FQ_NW * chunks_done)
        break;
      __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    FQ_T(9)
  }
  top.free_done();
  if (set_mode && (size_t)CAP * 8 <= (size_t)FQ_NW * FQ_ACC * 4)   // unordered exact top-k; the accumulators are dead: scratch
    top.finish_set(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
                   I32 ? I32 + (size_t)q * k : nullptr, reinterpret_cast<u64 *>(s_acc));
  else
    top.finish(D ? D + (size_t)q * k : nullptr, I64 ? I64 + (size_t)q * k : nullptr,
               I32 ? I32 + (size_t)q * k : nullptr);
  FQ_T(10)
  if constexpr (PROF) {
    if (lane == 0)
      for (int i = 0; i < 11; ++i) atomicAdd(&g_fq_prof[i], prof[i]);
  }
}

bool flat_q_supported(int d, int k, int nprobe) {
  return d <= 1024 && nprobe <= FQ_NT && k >= 1 && k + FQ_NT + 256 <= 2048;
}

// the fixed-point layout only (seg_tab = the byte table with the half-line flag, blk_base in lines)
int flat_q_scan(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                const int32_t *list_offsets, const int32_t *blk_offsets, const uint32_t *blk_base,
                const uint8_t *tab8, int tab_stride, const char *seg_bytes, const int32_t *ids, int k,
                float *D, int64_t *I64, int32_t *I32, int set_mode, const uint2 *ent,
                const int32_t *ent_cnt, const int *gate) {
  if (nq <= 0) return ASL_OK;
  using TopK = HistTopK<2048, FQ_NT, FQ_NT>;
  const size_t lds = TopK::lds_bytes() + (size_t)FQ_NW * FQ_ACC * 4 + (size_t)FQ_NW * FQ_G * FQ_DSTRIDE +
                     (size_t)((d + 3) & ~3) * 4 + (size_t)FQ_CHUNK * (sizeof(FqUnit) + 2) + 64 +
                     (size_t)((d + 7) & ~7) * 2;
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "flat scan: d=%d does not fit LDS", d);
  static const bool prof = [] {
    const char *e = getenv("ASL_FLAT_Q");
    return e && e[0] == '2';
  }();
  if (prof) {
    HIP_TRY(hipFuncSetAttribute((const void *)flat_q_scan_kernel<2048, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((flat_q_scan_kernel<2048, true>), dim3(nq), dim3(FQ_NT), lds, stream(), xq, d, coarse_I, nprobe,
                       list_offsets, blk_offsets, blk_base, tab8, tab_stride, seg_bytes, ids, k, D, I64, I32, set_mode,
                       ent, ent_cnt, gate);
    ASL_CHECK_LAUNCH();
    return ASL_OK;
  }
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)flat_q_scan_kernel<2048, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((flat_q_scan_kernel<2048, false>), dim3(nq), dim3(FQ_NT), lds, stream(), xq, d, coarse_I, nprobe,
                     list_offsets, blk_offsets, blk_base, tab8, tab_stride, seg_bytes, ids, k, D, I64, I32, set_mode,
                     ent, ent_cnt, gate);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// measurement build: wave-cycles per phase since the last call (reads and clears the counters)
int flat_q_prof(unsigned long long *out16) {
  HIP_TRY(hipStreamSynchronize(stream()));
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fq_prof), sizeof(unsigned long long) * 16));
  unsigned long long z[16] = {0};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fq_prof), z, sizeof z));
  return ASL_OK;
}

}  // namespace asl

// measurement hook of the ASL_FLAT_Q=2 build (scripts/flat_q_ab.py); not part of include/annsolo_mi.h
extern "C" int asl_debug_flat_q_prof(unsigned long long *out16) { return asl::flat_q_prof(out16); }
