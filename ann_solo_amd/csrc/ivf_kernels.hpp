// ivf_kernels.hpp -- host-callable wrappers of the IVF device kernels (all pointers
// are device pointers; everything is enqueued on asl::stream()).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace asl {

struct ScanPostFilter;            // common.hpp

constexpr int TK_NT = 256;        // threads per top-k workgroup
constexpr int TK_MAX_K = 2048;    // largest k / nprobe the LDS top-k supports
constexpr int TK_MAX_K_PASSES = 16384;   // largest k of a search: beyond TK_MAX_K in bounded passes of the generic kernels
constexpr int PQ_MAX_DSUB = 32;   // register fast path of the PQ L2 kernels

// gate != nullptr: the kernel returns at once when *gate <= gate_max (device-side choice
// between this GEMM and the sparse coarse quantiser)
int gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K, int lda,
                int ldb, int ldc, const int *gate = nullptr, int gate_max = 0);
// coarse quantiser for sparse queries (coarse_sparse.hip)
bool coarse_sparse_supported(int d, int nlist);
int coarse_sparse_cap();
// ent[q][64] = (dimension * 128, value bits) of the non-zero components of row q, ascending;
// cnt[q] = their number or -1 - count beyond 64; *n_over += rows beyond 64
int list_nonzeros(const float *xq, int nq, int d, int64_t ldq, uint2 *ent, int32_t *cnt, int *n_over);
int transpose_f32(const float *in, int rows, int cols, float *out);
int coarse_sparse(const float *xq, int nq, int d, const float *Ct, int nlist, uint2 *ent,
                  int32_t *cnt, int *n_over, int over_max, float *scores, int ld, int64_t ldq = 0);
int row_topk(const float *scores, int64_t ld, int rows, int n, int k, const int32_t *ids,
             int32_t id_base, const int32_t *vlist, const uint32_t *bitmap, int bitmap_words,
             float *D, int64_t *I64, int32_t *I32, int64_t out_ld,
             // bounded pass (k beyond TK_MAX_K in passes of <= TK_MAX_K): only keys below upper_in[row]
             // take part; upper_out[row] = the smallest key written by a full row, else 0
             const uint64_t *upper_in = nullptr, uint64_t *upper_out = nullptr);
int topk_merge(const float *Ds, const int64_t *Is, int S, int nq, int k, float *D, int64_t *I);
int topk_merge_keys(const int64_t *Ks, int S, int nq, int k, float *D, int64_t *I, int unordered);
int probe_bitmap(const int32_t *coarse_I, int nq, int nprobe, uint32_t *bitmap, int words);
int row_argmax(const float *scores, int64_t ld, int rows, int n, int32_t *out);
int gather_rows_f32(const float *src, int64_t ld_src, const int64_t *rows, int64_t n, int d,
                    float *dst, int64_t ld_dst);
int gather_rows_u8(const uint8_t *src, const int32_t *rows, int64_t n, int m, uint8_t *dst);
int centroid_update(const float *x, int64_t ld, int d, int k, const int32_t *order,
                    const int32_t *offsets, float *centroids);
int renorm_rows(float *c, int k, int d);
int l2_assign(const float *x, int64_t ld, int64_t n, int dsub, const float *cb, int ksub,
              int32_t *assign);
int residual(const float *x, const int32_t *assign, const float *centroids, int64_t n, int d,
             float *dst);
int pq_encode(const float *x, const int32_t *assign, const float *centroids,
              const float *codebooks, int64_t n, int d, int m, int ksub, int dsub,
              uint8_t *codes);
int pq_lut(const float *xq, int nq, int d, const float *codebooks, int m, int ksub, int dsub,
           float *lut_out);
int pq_scan(const float *xq, int nq, int d, const float *codebooks, int m, int ksub, int dsub,
            const float *coarse_D, const int32_t *coarse_I, int nprobe,
            const int32_t *list_offsets, const int32_t *ids, const uint8_t *codes, int k,
            float *D, int64_t *I64, int32_t *I32, int64_t out_ld = 0,
            const uint64_t *upper_in = nullptr, uint64_t *upper_out = nullptr);
// tiled IVF-PQ scan (pq_scan_v3.hip): m = 32 sub-quantisers of 8 bits
bool pq_scan_tiled_supported(int m, int ksub, int k, int nprobe);
int pq_scan_v3(const float *xq, int nq, int d, const float *codebooks, int dsub,
               const float *coarse_D, const int32_t *coarse_I, int nprobe,
               const int32_t *list_offsets, const int32_t *tile_offsets,
               const uint8_t *codes_tiled, const int32_t *ids_tiled, int k, float *D,
               int64_t *I64, int32_t *I32, int set_mode, const uint2 *ent = nullptr,
               const int32_t *ent_cnt = nullptr,    // ent / ent_cnt: list_nonzeros (64 entries per query)
               const int *gate = nullptr,
               const ScanPostFilter *post = nullptr);   // (common.hpp) set-mode int32 rows, k <= 1280 only          // device-side row count: workgroups past it return at once
int tile_codes(const uint8_t *codes, const int32_t *ids, const int32_t *dst_slot, int64_t n,
               int64_t ntiles, uint8_t *codes_tiled, int32_t *ids_tiled);
// dimension-major IVF-Flat (flat_scan.hip): blocks of FI_BLK vectors with per-dimension postings
// FI_BLK measured (scan ms at nprobe 128): 512 7.69 | 640 7.14 | 768 6.81 | 832 6.71 | 864 6.70 (the LDS limit of three workgroups per CU) | 1024: two workgroups per CU
constexpr int FI_BLK = 832;
bool flat_inv_supported(int d, int k, int nprobe);
// layout 1: float postings (seg_tab u32 [nblocks, d]); 2: fixed-point posting words behind a byte
// table (seg_tab u8 [nblocks, tab_stride]); see flat_scan.hip
int flat_inv_scan(int layout, const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                  const int32_t *list_offsets, const int32_t *blk_offsets,
                  const uint32_t *blk_base, const void *seg_tab, int tab_stride, const char *seg_bytes,
                  const int32_t *ids, int k, float *D, int64_t *I64, int32_t *I32, int set_mode,
                  const uint2 *ent, const int32_t *ent_cnt, const int *gate = nullptr,
                  const ScanPostFilter *post = nullptr);   // set-mode int32 rows, k <= 1280 only
int flat_inv_work(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                  const int32_t *blk_offsets, const uint32_t *seg_tab, unsigned long long *out_dev);
uint32_t inv_place_block(const uint32_t *cnt, int d, uint32_t *tab, bool *ok);
// exact re-rank of a short-list against sparse stored rows (refine.hip)
int refine_stride();
int refine_append_rows(const float *x, int64_t n, int d, int64_t row0, uint16_t *r_dim, float *r_val,
                       uint8_t *r_cnt, int *status);
int refine_topk(const float *xq, int nq, int d, const int32_t *I_in, const int64_t *I_in64, int kp,
                const uint16_t *r_dim, const float *r_val, const uint8_t *r_cnt, int64_t n_rows, int k, float *D,
                int64_t *I64, int32_t *I32);
int inv_count(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk, int64_t n,
              uint32_t *cnt);
int inv_fill(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk,
             const uint16_t *pos_loc, int64_t n, const uint32_t *blk_base,
             const uint32_t *seg_tab, uint32_t *cursor, char *seg_bytes);
// fixed-point storage (22 fractional bits) of IVF-Flat components in [0, 1)
__host__ __device__ inline float fx22_round(float x) {
  if (!(x >= 0.0f && x < 1.0f)) return x;
  const float m = rintf(x * 4194304.0f);          // a power of two: exact; ties to even
  return (m < 4194303.0f ? m : 4194303.0f) * (1.0f / 4194304.0f);
}
__host__ __device__ inline bool fx22_on_grid(float x) {
  return x > 0.0f && x < 1.0f && x * 4194304.0f == rintf(x * 4194304.0f);
}
int quantize_fx22(float *x, int64_t n);
// the exact top-k exchange of a sharded search (exchange.hip)
int keys_split(const unsigned long long *K, int64_t nrows, int k, int kp, unsigned long long *head,
               int32_t *floor_out, unsigned long long *rowmin_out = nullptr);
int keys_merge(const unsigned long long *heads, int S, int nq, int kp, int k, const unsigned long long *xbuf,
               long long xcap, const unsigned long long *prev_keys, int32_t *need,
               unsigned long long *out_keys, unsigned long long *bounds, int64_t *I, float *D, int sorted);
int keys_extras(const unsigned long long *K, const int32_t *floor_in, int64_t nrows, int k,
                const unsigned long long *bounds, int nq, long long xcap, unsigned long long *xbuf,
                unsigned int *cursor, int32_t *overflow, const int32_t *rmap = nullptr,
                const unsigned long long *K3 = nullptr, int k3 = 0);
int rescan_list(const unsigned long long *bounds, const unsigned long long *rowmin, int64_t nrows, int R,
                int64_t *rowlist, int32_t *rmap, int *count, int32_t *overflow);
int flat_fx_work(const float *xq, int nq, int d, const int32_t *coarse_I, int nprobe,
                 const int32_t *blk_offsets, const uint8_t *tab8, int tab_stride,
                 const uint16_t *cnt16, unsigned long long *out_dev);
int fx_fill(const float *vecs, int d, const int32_t *order, const int32_t *pos_blk,
            const uint16_t *pos_loc, int64_t n, const uint32_t *seg_line, uint32_t *cursor,
            uint32_t *words);
int fx_order(int64_t ncell, const uint32_t *seg_line, const uint32_t *count, uint32_t *words);
int inv_order(int64_t nblocks, int d, const uint32_t *blk_base, const uint32_t *seg_tab,
              char *seg_bytes);
int count_nnz(const float *vecs, int d, int64_t n, int32_t *nnz, int32_t *nnz_max_dev);
int scanned_count(const int32_t *coarse_I, int64_t n, const int32_t *list_offsets,
                  unsigned long long *out_dev);

}  // namespace asl
