// index.hip -- the ANN index object behind the FAISS-shaped C ABI
// (replaces IndexFlatIP / IndexIVFFlat + train/add/search/write_index/read_index used at
// /root/reference/src/ann_solo/spectral_library.py:73-87,167-181,191,443-445,487-497;
// IVF-PQ is the north star's addition). Host orchestration only -- every flop runs
// in the kernels of gemm.hip / ivf_kernels.hip.
//
// Training restates FAISS' Clustering (Lloyd iterations, assignment by the
// quantiser's metric, mean update, empty-cluster split with eps = 1/1024, at most
// 256 points per centroid; SPHERICAL -- centroids L2-renormalised every iteration -- for
// the inner-product coarse quantiser, as FAISS' IndexIVF sets cp.spherical for
// METRIC_INNER_PRODUCT) with a library-local RNG; it is deterministic and
// bit-identical to oracle/asl_oracle.c:orc_kmeans for the same seed.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.hpp"
#include "ivf_kernels.hpp"

namespace asl {

// ------------------------------------------------------------------ RNG (same as the oracle's)
static inline uint64_t sm64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void rand_perm(int64_t n, uint64_t seed, std::vector<int64_t> &perm) {
  uint64_t s = seed;
  perm.resize((size_t)n);
  for (int64_t i = 0; i < n; i++) perm[(size_t)i] = i;
  for (int64_t i = 0; i + 1 < n; i++) {
    int64_t j = i + (int64_t)(sm64(&s) % (uint64_t)(n - i));
    std::swap(perm[(size_t)i], perm[(size_t)j]);
  }
}

template <class T>
static int dev_append(DevBuf<T> &buf, size_t old_n, const T *src_dev, size_t n) {
  if (old_n + n > buf.cap) {
    DevBuf<T> nb;
    size_t want = std::max(old_n + n, buf.cap + buf.cap / 2);
    ASL_TRY(nb.reserve(want));
    if (old_n)
      HIP_TRY(hipMemcpyAsync(nb.p, buf.p, old_n * sizeof(T), hipMemcpyDeviceToDevice, stream()));
    ASL_TRY(sync_stream());
    buf = std::move(nb);
  }
  if (n)
    HIP_TRY(hipMemcpyAsync(buf.p + old_n, src_dev, n * sizeof(T), hipMemcpyDeviceToDevice, stream()));
  return ASL_OK;
}

constexpr size_t SCORE_CHUNK_BYTES = (size_t)1 << 30;
constexpr int FLAT_KEYS_SLACK = 768;      // packed-key rows of the postings scan: k + 768 <= its 2048-key buffer

}  // namespace asl

using namespace asl;

struct asl_index {
  int d = 0, nlist = 0, kind = 0, pq_m = 0, pq_bits = 8, ksub = 0, dsub = 0;
  int niter = 25;
  bool trained = false;
  int64_t ntotal = 0;   // global vectors added
  int64_t n_store = 0;  // vectors stored here
  int shard_rank = 0, shard_world = 1;
  DevBuf<float> centroids, codebooks;
  DevBuf<float> codebooks_t;  // [m][dsub][ksub] copy for the tiled scan's LUT build
  bool cbt_ready = false;
  // sparse coarse quantiser (coarse_sparse.hip): transposed centroids [d][nlist] + per-batch scratch
  DevBuf<float> centroids_t;
  bool cent_t_ready = false;
  DevBuf<float> kmeans_ct;       // transposed centroids of the running k-means iteration
  DevBuf<uint2> scan_ent;        // the scan's own entry lists (the coarse stage of the NEXT batch
  DevBuf<int32_t> scan_cnt;      // overwrites cs_ent on the other stream of the pipeline)
  DevBuf<int> scan_over;
  DevBuf<uint2> cs_ent;
  DevBuf<int32_t> cs_cnt;
  DevBuf<int> cs_over;
  // add-order storage
  DevBuf<float> vecs;        // FLAT, IVFFLAT
  DevBuf<int32_t> vlist;     // IVF kinds: inverted list of each stored vector
  DevBuf<int32_t> vids;      // global id of each stored vector (only when sharded)
  bool has_vids = false;
  DevBuf<uint8_t> codes_add; // IVFPQ
  // list-order storage (IVFPQ scan layout)
  DevBuf<uint8_t> codes;
  DevBuf<int32_t> ids, list_offsets;
  std::vector<int32_t> h_list_offsets;
  // 64-vector tiles for the tiled scan, pq_scan_v3.hip (m = 32)
  DevBuf<uint8_t> codes_tiled;
  DevBuf<int32_t> ids_tiled, tile_offsets;
  // post-filter in the scan's finish (common.hpp: ScanPostFilter): (id, window value) per storage slot
  // -- per tile slot (IVF-PQ) or per list position (IVF-Flat) --, built from pay_src on demand
  DevBuf<int2> idpay;
  int64_t n_tile_slots = 0;
  const float *pay_src = nullptr;
  int64_t pay_n = 0;
  bool idpay_ready = false;
  IndexPostFilter post;          // for the next search only
  bool post_set = false, post_applied = false;
  bool has_tiles = false;
  // dimension-major postings for flat_inv_scan (IVF-Flat): blocks of FI_BLK vectors
  DevBuf<int32_t> blk_offsets;   // [nlist + 1] first block of each list
  DevBuf<uint32_t> blk_base;     // [nblocks] start of the block's postings, 64-byte units
  DevBuf<uint32_t> inv_tab;      // [nblocks * d] (start from the block's base in 64-byte units) << 16 | postings
  DevBuf<char> inv_data;         // segments: c values (f32) then c local vector indices (u16), placed by 128-byte line
  bool has_inv = false;
  // the fixed-point layout (inv_layout 2): blk_base in 128-byte lines, one byte per (block,
  // dimension) = lines of the segment, posting words (numerator << 10 | local index) in inv_data
  int flat_storage = ASL_FLAT_F32;   // ASL_FLAT_F32 (default): components as given; ASL_FLAT_FX22: add() rounds
                                     // components in [0, 1) to 22 fractional bits
  int inv_layout = 0;            // what build_lists found the data fit for: 0 none, 1 float postings, 2 fixed-point words
  int tab_stride = 0;            // bytes per block of inv_tab8 (d rounded up to a 128-byte line)
  DevBuf<uint8_t> inv_tab8;
  DevBuf<uint16_t> inv_cnt16;    // postings per (block, dimension): work accounting only (asl_index_postings_work)
  int scan_variant = 0;  // 0 = the layout-specific scan when the shape allows; 1 = the generic kernels
  int unordered = 0;  // 1: search rows = exact top-k as a set, unspecified order (no final sort); 2: rows of packed keys
  bool lists_dirty = true;
  // asl_index_search_sharded: did every rank's shard answer asl_index_supports_keys with 1 for this
  // (k, nprobe, world)? -1 = not agreed yet (reset whenever the lists are rebuilt)
  int agreed_k = -1, agreed_np = -1, agreed_world = -1, agreed_val = -1;
  // exact re-rank of the IVF-PQ short-list (refine.hip): sparse copies of the added vectors,
  // add order = global id; kept whole on every shard
  int refine_k = 0;            // 0 = off; else the short-list size k' (> k) that is re-ranked
  bool refine_rows = false;    // rows are being stored on add()
  bool refine_bad = false;     // a vector had more non-zeros than a row holds
  int64_t r_n = 0;
  DevBuf<uint16_t> r_dim;
  DevBuf<float> r_val;
  DevBuf<uint8_t> r_cnt;
  DevBuf<int32_t> ws_short;
  // scratch
  DevBuf<float> ws_scores, coarse_D, ws_x;
  DevBuf<int32_t> coarse_I, ws_assign;
  DevBuf<uint32_t> bitmap;
};

namespace asl {

// Assign rows of x (device, [n, ld]) to the centroid with the largest inner product
// (ties: lowest index) -> assign_dev[n].
static int assign_ip(asl_index *ix, const float *x, int64_t ld, int64_t n, const float *cent,
                     int k, int d, int32_t *assign_dev) {
  if (n <= 0) return ASL_OK;
  int64_t rows = (int64_t)std::max<size_t>(1, SCORE_CHUNK_BYTES / ((size_t)k * 4));
  rows = std::min<int64_t>(rows, n);
  ASL_TRY(ix->ws_scores.reserve((size_t)rows * k));
  // hashed spectra are sparse: the scores come from the sparse kernel (same bits as the GEMM,
  // 1/16 of its multiply-adds); a chunk with many dense rows is left to the GEMM by the
  // device-side gate (coarse_sparse.hip). The centroids change every iteration: transposed here.
  const bool sparse = ix->scan_variant == 0 && coarse_sparse_supported(d, k);
  if (sparse) {
    ASL_TRY(ix->kmeans_ct.reserve((size_t)k * d));
    ASL_TRY(transpose_f32(cent, k, d, ix->kmeans_ct.p));
    ASL_TRY(ix->cs_ent.reserve((size_t)rows * coarse_sparse_cap()));
    ASL_TRY(ix->cs_cnt.reserve((size_t)rows));
    ASL_TRY(ix->cs_over.reserve(1));
  }
  for (int64_t r0 = 0; r0 < n; r0 += rows) {
    const int m = (int)std::min<int64_t>(rows, n - r0);
    const int over_max = m / 64;
    if (sparse)
      ASL_TRY(coarse_sparse(x + (size_t)r0 * ld, m, d, ix->kmeans_ct.p, k, ix->cs_ent.p, ix->cs_cnt.p,
                            ix->cs_over.p, over_max, ix->ws_scores.p, k, ld));
    ASL_TRY(gemm_nt_f32(x + (size_t)r0 * ld, cent, ix->ws_scores.p, m, k, d, (int)ld, d, k,
                        sparse ? ix->cs_over.p : nullptr, over_max));
    ASL_TRY(row_argmax(ix->ws_scores.p, k, m, k, assign_dev + r0));
  }
  return ASL_OK;
}

// Lloyd k-means on the device; see file header. x: device [n, ld].
static int kmeans_device(asl_index *ix, const float *x, int64_t n, int64_t ld, int d, int k,
                         int niter, uint64_t seed, bool l2, int max_ppc, float *cent_dev) {
  if (n < 1) return fail(ASL_ERR_INVALID, "train: no training vectors");
  int64_t nt = n;
  const float *xt = x;
  int64_t ldt = ld;
  DevBuf<float> sub;
  DevBuf<int64_t> rows_dev;
  std::vector<int64_t> perm;
  if (max_ppc > 0 && n > (int64_t)k * max_ppc) {
    nt = (int64_t)k * max_ppc;
    rand_perm(n, seed, perm);
    ASL_TRY(rows_dev.upload(perm.data(), (size_t)nt));
    ASL_TRY(sub.reserve((size_t)nt * d));
    ASL_TRY(gather_rows_f32(x, ld, rows_dev.p, nt, d, sub.p, d));
    xt = sub.p;
    ldt = d;
  }
  {  // init: k distinct random training points
    rand_perm(nt, seed + 1, perm);
    std::vector<int64_t> pick((size_t)k);
    for (int c = 0; c < k; c++) pick[(size_t)c] = perm[(size_t)(c % nt)];
    ASL_TRY(rows_dev.upload(pick.data(), (size_t)k));
    ASL_TRY(gather_rows_f32(xt, ldt, rows_dev.p, k, d, cent_dev, d));
  }
  // FAISS IndexIVF: cp.spherical = true for METRIC_INNER_PRODUCT -> post_process_centroids
  // renormalises after the initialisation and after every iteration (oracle: orc_kmeans)
  const bool spherical = !l2;
  if (spherical) ASL_TRY(renorm_rows(cent_dev, k, d));
  DevBuf<int32_t> assign, order, offsets;
  ASL_TRY(assign.reserve((size_t)nt));
  ASL_TRY(order.reserve((size_t)nt));
  ASL_TRY(offsets.reserve((size_t)k + 1));
  std::vector<int32_t> h_assign((size_t)nt), h_order((size_t)nt), h_off((size_t)k + 1);
  std::vector<float> hassign((size_t)k), h_cent;
  uint64_t rng = seed + 2;
  for (int it = 0; it < niter; it++) {
    if (l2)
      ASL_TRY(l2_assign(xt, ldt, nt, d, cent_dev, k, assign.p));
    else
      ASL_TRY(assign_ip(ix, xt, ldt, nt, cent_dev, k, d, assign.p));
    ASL_TRY(assign.download(h_assign.data(), (size_t)nt));
    ASL_TRY(sync_stream());
    // counting sort by cluster (stable: ascending point order inside a cluster)
    std::fill(h_off.begin(), h_off.end(), 0);
    for (int64_t i = 0; i < nt; i++) h_off[(size_t)h_assign[(size_t)i] + 1]++;
    for (int c = 0; c < k; c++) {
      hassign[(size_t)c] = (float)h_off[(size_t)c + 1];
      h_off[(size_t)c + 1] += h_off[(size_t)c];
    }
    {
      std::vector<int32_t> cur(h_off.begin(), h_off.end() - 1);
      for (int64_t i = 0; i < nt; i++) h_order[(size_t)cur[(size_t)h_assign[(size_t)i]]++] = (int32_t)i;
    }
    ASL_TRY(order.upload(h_order.data(), (size_t)nt));
    ASL_TRY(offsets.upload(h_off.data(), (size_t)k + 1));
    ASL_TRY(centroid_update(xt, ldt, d, k, order.p, offsets.p, cent_dev));
    bool any_empty = false;
    for (int c = 0; c < k; c++) any_empty |= hassign[(size_t)c] == 0.0f;
    if (any_empty) {  // FAISS split_clusters, on the host (rare)
      h_cent.resize((size_t)k * d);
      HIP_TRY(hipMemcpyAsync(h_cent.data(), cent_dev, h_cent.size() * 4, hipMemcpyDeviceToHost, stream()));
      ASL_TRY(sync_stream());
      const float eps = 1.0f / 1024.0f;
      for (int ci = 0; ci < k; ci++) {
        if (hassign[(size_t)ci] != 0.0f) continue;
        int cj = 0;
        for (int guard = 0; guard < 64 * k + 64; guard++, cj = (cj + 1) % k) {
          float p = (hassign[(size_t)cj] - 1.0f) / (float)(nt - k);
          float r = (float)(sm64(&rng) >> 40) * (1.0f / 16777216.0f);
          if (r < p) break;
        }
        float *a = h_cent.data() + (size_t)ci * d, *b = h_cent.data() + (size_t)cj * d;
        memcpy(a, b, sizeof(float) * (size_t)d);
        for (int j = 0; j < d; j++) {
          if (j % 2 == 0) {
            a[j] *= 1 + eps;
            b[j] *= 1 - eps;
          } else {
            a[j] *= 1 - eps;
            b[j] *= 1 + eps;
          }
        }
        hassign[(size_t)ci] = floorf(hassign[(size_t)cj] / 2);
        hassign[(size_t)cj] -= hassign[(size_t)ci];
      }
      HIP_TRY(hipMemcpyAsync(cent_dev, h_cent.data(), h_cent.size() * 4, hipMemcpyHostToDevice, stream()));
      ASL_TRY(sync_stream());
    }
    if (spherical) ASL_TRY(renorm_rows(cent_dev, k, d));
  }
  ASL_TRY(sync_stream());
  return ASL_OK;
}

static int pq_train_device(asl_index *ix, const float *x, int64_t n, uint64_t seed) {
  const int d = ix->d, m = ix->pq_m, ksub = ix->ksub, dsub = ix->dsub;
  int64_t nt = n;
  const int64_t cap = (int64_t)ksub * 256;
  DevBuf<float> xt;
  DevBuf<int64_t> rows_dev;
  if (nt > cap) {
    nt = cap;
    std::vector<int64_t> perm;
    rand_perm(n, seed, perm);
    ASL_TRY(rows_dev.upload(perm.data(), (size_t)nt));
  }
  ASL_TRY(xt.reserve((size_t)nt * d));
  ASL_TRY(gather_rows_f32(x, d, nt < n ? rows_dev.p : nullptr, nt, d, xt.p, d));
  DevBuf<int32_t> assign;
  ASL_TRY(assign.reserve((size_t)nt));
  ASL_TRY(assign_ip(ix, xt.p, d, nt, ix->centroids.p, ix->nlist, d, assign.p));
  ASL_TRY(residual(xt.p, assign.p, ix->centroids.p, nt, d, xt.p));
  ASL_TRY(ix->codebooks.reserve((size_t)m * ksub * dsub));
  for (int mi = 0; mi < m; mi++)
    ASL_TRY(kmeans_device(ix, xt.p + (size_t)mi * dsub, nt, d, dsub, ksub, ix->niter,
                          seed + 16 + (uint64_t)mi, true, 0,
                          ix->codebooks.p + (size_t)mi * ksub * dsub));
  return ASL_OK;
}

// list-ordered copy of the PQ codes (the scan layout) from the add-order master
static int build_lists(asl_index *ix) {
  if (!ix->lists_dirty) return ASL_OK;
  ix->agreed_val = -1;
  ix->idpay_ready = false;
  const int64_t n = ix->n_store;
  std::vector<int32_t> h_vlist((size_t)n), h_order((size_t)n), h_ids;
  ix->h_list_offsets.assign((size_t)ix->nlist + 1, 0);
  if (n) {
    ASL_TRY(ix->vlist.download(h_vlist.data(), (size_t)n));
    ASL_TRY(sync_stream());
  }
  auto &off = ix->h_list_offsets;
  for (int64_t i = 0; i < n; i++) off[(size_t)h_vlist[(size_t)i] + 1]++;
  for (int l = 0; l < ix->nlist; l++) off[(size_t)l + 1] += off[(size_t)l];
  {
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    for (int64_t i = 0; i < n; i++) h_order[(size_t)cur[(size_t)h_vlist[(size_t)i]]++] = (int32_t)i;
  }
  ASL_TRY(ix->list_offsets.upload(off.data(), off.size()));
  DevBuf<int32_t> order;
  ASL_TRY(order.upload(h_order.data(), (size_t)n));
  if (ix->kind == ASL_INDEX_IVFPQ) {
    ASL_TRY(ix->codes.reserve((size_t)std::max<int64_t>(n, 1) * ix->pq_m));
    ASL_TRY(gather_rows_u8(ix->codes_add.p, order.p, n, ix->pq_m, ix->codes.p));
  }
  if (ix->has_vids) {
    h_ids.resize((size_t)n);
    std::vector<int32_t> h_vids((size_t)n);
    if (n) {
      ASL_TRY(ix->vids.download(h_vids.data(), (size_t)n));
      ASL_TRY(sync_stream());
    }
    for (int64_t i = 0; i < n; i++) h_ids[(size_t)i] = h_vids[(size_t)h_order[(size_t)i]];
    ASL_TRY(ix->ids.upload(h_ids.data(), (size_t)n));
  } else {
    ASL_TRY(ix->ids.upload(h_order.data(), (size_t)n));
  }
  ix->has_tiles = false;
  if (ix->kind == ASL_INDEX_IVFPQ && ix->pq_m == 32 && ix->ksub == 256) {
    std::vector<int32_t> tile_off((size_t)ix->nlist + 1, 0), dst_slot((size_t)n);
    for (int l = 0; l < ix->nlist; l++)
      tile_off[(size_t)l + 1] = tile_off[(size_t)l] + (off[(size_t)l + 1] - off[(size_t)l] + 63) / 64;
    for (int l = 0; l < ix->nlist; l++)
      for (int32_t i = off[(size_t)l]; i < off[(size_t)l + 1]; i++)
        dst_slot[(size_t)i] = tile_off[(size_t)l] * 64 + (i - off[(size_t)l]);
    const int64_t ntiles = std::max<int64_t>(tile_off[(size_t)ix->nlist], 1);
    DevBuf<int32_t> slot_dev;
    ASL_TRY(slot_dev.upload(dst_slot.data(), (size_t)n));
    ASL_TRY(ix->tile_offsets.upload(tile_off.data(), tile_off.size()));
    ASL_TRY(ix->codes_tiled.reserve((size_t)ntiles * 2048));
    ASL_TRY(ix->ids_tiled.reserve((size_t)ntiles * 64));
    ix->n_tile_slots = ntiles * 64;
    ASL_TRY(tile_codes(ix->codes.p, ix->ids.p, slot_dev.p, n, ntiles, ix->codes_tiled.p, ix->ids_tiled.p));
    ASL_TRY(sync_stream());
    ix->has_tiles = true;
  }
  ix->has_inv = false;
  if (ix->kind == ASL_INDEX_IVFFLAT && n > 0 && ix->d <= 65535) {
    DevBuf<int32_t> nnz, nnz_max;
    ASL_TRY(nnz.reserve((size_t)n));
    ASL_TRY(nnz_max.reserve(2));
    ASL_TRY(count_nnz(ix->vecs.p, ix->d, n, nnz.p, nnz_max.p));
    int32_t h_nm[2] = {0, 0};
    ASL_TRY(nnz_max.download(h_nm, 2));
    ASL_TRY(sync_stream());
    const int32_t h_max = h_nm[0];
    // every non-zero on the 2^-22 grid inside (0, 1): posting words (flat_scan.hip, FX)
    const bool fixed_point = h_nm[1] == 0 && ix->d <= 1024 && FI_BLK <= 1024;
    // dimension-major postings (the default IVF-Flat scan)
    if (h_max > 0 && (size_t)h_max * 8 < (size_t)ix->d) {
      std::vector<int32_t> blk_off((size_t)ix->nlist + 1, 0), pos_blk((size_t)n);
      std::vector<uint16_t> pos_loc((size_t)n);
      for (int l = 0; l < ix->nlist; l++)
        blk_off[(size_t)l + 1] =
            blk_off[(size_t)l] + (off[(size_t)l + 1] - off[(size_t)l] + FI_BLK - 1) / FI_BLK;
      for (int l = 0; l < ix->nlist; l++)
        for (int32_t i = off[(size_t)l]; i < off[(size_t)l + 1]; i++) {
          const int32_t r = i - off[(size_t)l];
          pos_blk[(size_t)i] = blk_off[(size_t)l] + r / FI_BLK;
          pos_loc[(size_t)i] = (uint16_t)(r % FI_BLK);
        }
      const size_t nblk = (size_t)std::max<int32_t>(blk_off[(size_t)ix->nlist], 1);
      const size_t ncell = nblk * (size_t)ix->d;
      DevBuf<int32_t> pos_blk_dev;
      DevBuf<uint16_t> pos_loc_dev;
      DevBuf<uint32_t> cnt_dev;
      ASL_TRY(pos_blk_dev.upload(pos_blk.data(), (size_t)n));
      ASL_TRY(pos_loc_dev.upload(pos_loc.data(), (size_t)n));
      ASL_TRY(cnt_dev.reserve(ncell));
      HIP_TRY(hipMemsetAsync(cnt_dev.p, 0, ncell * 4, stream()));
      ASL_TRY(inv_count(ix->vecs.p, ix->d, order.p, pos_blk_dev.p, n, cnt_dev.p));
      std::vector<uint32_t> h_cnt(ncell);
      ASL_TRY(cnt_dev.download(h_cnt.data(), ncell));
      ASL_TRY(sync_stream());
      ix->inv_layout = 0;
      if (fixed_point) {
        // whole lines of 32 posting words per (block, dimension); the table byte is the line count
        const int stride = (ix->d + 127) & ~127;
        std::vector<uint8_t> h_tab8(nblk * (size_t)stride, 0);
        std::vector<uint16_t> h_c16(ncell);
        std::vector<uint32_t> h_line(ncell), h_base(nblk);
        uint64_t run = 0;     // 128-byte lines
        for (size_t b = 0; b < nblk; b++) {
          h_base[b] = (uint32_t)run;
          for (int j = 0; j < ix->d; j++) {
            const uint32_t c = h_cnt[b * (size_t)ix->d + j];       // <= FI_BLK: at most 26 lines
            h_line[b * (size_t)ix->d + j] = (uint32_t)run;
            h_c16[b * (size_t)ix->d + j] = (uint16_t)c;
            h_tab8[b * (size_t)stride + j] = (uint8_t)((c + 31) / 32);
            run += (c + 31) / 32;
          }
        }
        if (run < (1ull << 32)) {
          const size_t bytes = (size_t)std::max<uint64_t>(run, 1) * 128 + 512;
          DevBuf<uint32_t> line_dev;
          ASL_TRY(line_dev.upload(h_line.data(), ncell));
          ASL_TRY(ix->blk_offsets.upload(blk_off.data(), blk_off.size()));
          ASL_TRY(ix->blk_base.upload(h_base.data(), nblk));
          ASL_TRY(ix->inv_tab8.upload(h_tab8.data(), h_tab8.size()));
          ASL_TRY(ix->inv_cnt16.upload(h_c16.data(), ncell));
          ASL_TRY(ix->inv_data.reserve(bytes));
          HIP_TRY(hipMemsetAsync(ix->inv_data.p, 0, bytes, stream()));
          HIP_TRY(hipMemsetAsync(cnt_dev.p, 0, ncell * 4, stream()));
          ASL_TRY(fx_fill(ix->vecs.p, ix->d, order.p, pos_blk_dev.p, pos_loc_dev.p, n, line_dev.p,
                          cnt_dev.p, reinterpret_cast<uint32_t *>(ix->inv_data.p)));
          ASL_TRY(fx_order((int64_t)ncell, line_dev.p, cnt_dev.p, reinterpret_cast<uint32_t *>(ix->inv_data.p)));
          ASL_TRY(sync_stream());
          ix->tab_stride = stride;
          ix->has_inv = true;
          ix->inv_layout = 2;
        }
      }
      if (!ix->has_inv) {
      // segments placed block by block (flat_scan.hip: inv_place_block)
      std::vector<uint32_t> h_tab(ncell), h_base(nblk);
      uint64_t run = 0;     // 64-byte units
      bool ok = true;
      for (size_t b = 0; b < nblk && ok; b++) {
        h_base[b] = (uint32_t)run;
        run += inv_place_block(h_cnt.data() + b * (size_t)ix->d, ix->d, h_tab.data() + b * (size_t)ix->d, &ok);
        run += run & 1ull;      // every block starts on a 128-byte line
      }
      if (ok && run < (1ull << 32)) {     // 32-bit unit offsets (256 GB of postings)
        // (+ 512: the scan reads a full wave-width from the start of an empty segment)
        const size_t bytes = (size_t)std::max<uint64_t>(run, 2) * 64 + 512;
        ASL_TRY(ix->blk_offsets.upload(blk_off.data(), blk_off.size()));
        ASL_TRY(ix->blk_base.upload(h_base.data(), nblk));
        ASL_TRY(ix->inv_tab.upload(h_tab.data(), ncell));
        ASL_TRY(ix->inv_data.reserve(bytes));
        HIP_TRY(hipMemsetAsync(ix->inv_data.p, 0, bytes, stream()));
        HIP_TRY(hipMemsetAsync(cnt_dev.p, 0, ncell * 4, stream()));
        ASL_TRY(inv_fill(ix->vecs.p, ix->d, order.p, pos_blk_dev.p, pos_loc_dev.p, n,
                         ix->blk_base.p, ix->inv_tab.p, cnt_dev.p, ix->inv_data.p));
        ASL_TRY(inv_order((int64_t)nblk, ix->d, ix->blk_base.p, ix->inv_tab.p, ix->inv_data.p));
        ASL_TRY(sync_stream());
        ix->has_inv = true;
        ix->inv_layout = 1;
      }
      }
    }
  }
  ASL_TRY(sync_stream());
  ix->lists_dirty = false;
  return ASL_OK;
}

// transposed centroid copy for the sparse coarse quantiser (rebuilt after train / set_trained)
static int coarse_transposed(asl_index *ix) {
  if (ix->cent_t_ready) return ASL_OK;
  ASL_TRY(ix->centroids_t.reserve((size_t)ix->nlist * ix->d));
  ASL_TRY(transpose_f32(ix->centroids.p, ix->nlist, ix->d, ix->centroids_t.p));
  ix->cent_t_ready = true;
  return ASL_OK;
}

// coarse quantiser: top-nprobe centroids by inner product -> ix->coarse_D / coarse_I
// ent_out / cnt_out (caller buffers [nq * 64] / [nq], may be null): the queries' entry lists, which
// the sparse coarse kernel lists anyway, for the scan that follows (*have_ent says whether they were
// produced: only the sparse formulation makes them)
static int coarse_search(asl_index *ix, const float *xq, int nq, int nprobe,
                         float *out_D = nullptr, int32_t *out_I = nullptr, uint2 *ent_out = nullptr,
                         int32_t *cnt_out = nullptr, bool *have_ent = nullptr) {
  const int nlist = ix->nlist, d = ix->d;
  if (have_ent) *have_ent = false;
  if (!out_D) {
    ASL_TRY(ix->coarse_D.reserve((size_t)nq * nprobe));
    ASL_TRY(ix->coarse_I.reserve((size_t)nq * nprobe));
    out_D = ix->coarse_D.p;
    out_I = ix->coarse_I.p;
  }
  int rows = (int)std::min<int64_t>(nq, std::max<int64_t>(1, (int64_t)(SCORE_CHUNK_BYTES / ((size_t)nlist * 4))));
  ASL_TRY(ix->ws_scores.reserve((size_t)rows * nlist));
  // Hashed spectra are sparse (<= ~50 of 800 components): the scores come from the sparse
  // kernel, bit-identical to the GEMM. Both are enqueued; a device-side count of dense rows
  // (more than 64 non-zeros) decides which of the two does the work (the other returns at once).
  const bool sparse = ix->scan_variant == 0 && coarse_sparse_supported(d, nlist);
  if (sparse) {
    ASL_TRY(coarse_transposed(ix));
    ASL_TRY(ix->cs_ent.reserve((size_t)rows * coarse_sparse_cap()));
    ASL_TRY(ix->cs_cnt.reserve((size_t)rows));
    ASL_TRY(ix->cs_over.reserve(1));
  }
  for (int r0 = 0; r0 < nq; r0 += rows) {
    const int m = std::min(rows, nq - r0);
    {
      ProfScope ps("coarse_gemm");
      const int over_max = m / 64;
      if (sparse)
        ASL_TRY(coarse_sparse(xq + (size_t)r0 * d, m, d, ix->centroids_t.p, nlist,
                              ent_out ? ent_out + (size_t)r0 * 64 : ix->cs_ent.p,
                              cnt_out ? cnt_out + (size_t)r0 : ix->cs_cnt.p,
                              ix->cs_over.p, over_max, ix->ws_scores.p, nlist));
      ASL_TRY(gemm_nt_f32(xq + (size_t)r0 * d, ix->centroids.p, ix->ws_scores.p, m, nlist, d, d, d, nlist,
                          sparse ? ix->cs_over.p : nullptr, over_max));
    }
    {
      ProfScope ps("coarse_select");
      ASL_TRY(row_topk(ix->ws_scores.p, nlist, m, nlist, nprobe, nullptr, 0, nullptr, nullptr, 0,
                       out_D + (size_t)r0 * nprobe, nullptr, out_I + (size_t)r0 * nprobe, nprobe));
    }
  }
  // (own buffers: valid for the whole batch only when it was one chunk)
  if (have_ent) *have_ent = sparse && coarse_sparse_cap() == 64 && ((ent_out && cnt_out) || nq <= rows);
  return ASL_OK;
}

// Search with all-device arguments. Exactly one of I64 / I32 may be non-null (or both).
// (id, window value) per storage slot for the post-filter of the scans' finish
__global__ void make_idpay_kernel(const int32_t *__restrict__ slot_ids, int64_t nslots,
                                  const float *__restrict__ payload, int64_t n, int2 *__restrict__ out) {
  const int64_t i = block_linear() * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  const int32_t id = slot_ids[i];
  const float v = (id >= 0 && id < n) ? payload[id] : __builtin_nanf("");
  out[i] = make_int2(id, __float_as_int(v));
}

// the scan's post-filter for this call, or an empty one (and post_applied = false)
static int take_post_filter(asl_index *ix, bool usable, const int32_t *slot_ids, int64_t nslots,
                            ScanPostFilter &pf) {
  pf = ScanPostFilter();
  ix->post_applied = false;
  if (!ix->post_set) return ASL_OK;
  ix->post_set = false;
  const IndexPostFilter &p = ix->post;
  if (!usable || !p.payload || !p.q_pmz || !p.count || p.n != ix->ntotal || ix->has_vids) return ASL_OK;
  if (!ix->idpay_ready || ix->pay_src != p.payload || ix->pay_n != p.n) {
    ASL_TRY(ix->idpay.reserve((size_t)std::max<int64_t>(nslots, 1)));
    if (nslots > 0) {
      hipLaunchKernelGGL(make_idpay_kernel, grid_2d(cdiv(nslots, 256)), dim3(256), 0, stream(), slot_ids, nslots,
                         p.payload, p.n, ix->idpay.p);
      ASL_CHECK_LAUNCH();
    }
    ix->pay_src = p.payload;
    ix->pay_n = p.n;
    ix->idpay_ready = true;
  }
  pf.idpay = ix->idpay.p;
  pf.q_pmz = p.q_pmz;
  pf.count = p.count;
  pf.tol = p.tol;
  pf.mode = p.mode;
  pf.charge = p.charge;
  ix->post_applied = true;
  return ASL_OK;
}

// k > TK_MAX_K: ceil(k / TK_MAX_K) bounded passes of the generic kernels. Every hit has a unique
// 64-bit key (score, ~id); a pass keeps the TK_MAX_K best keys strictly below the row's bound = the
// smallest key the pass before it wrote (0 once a row is exhausted), and writes them behind the
// earlier ones: the rows are the exact (score desc, id asc) top-k, -1 padded, as for small k.
// IndexFlatIP / IVF-Flat: the scores of a row chunk are computed once (GEMM) and selected from
// ceil(k / 2048) times; IVF-PQ: the ADC scan of the generic kernel is repeated per pass.
static int index_search_large_k(asl_index *ix, int nq, const float *xq, int k, int nprobe, float *D,
                                int64_t *I64, int32_t *I32, const float *pre_D, const int32_t *pre_I) {
  const int d = ix->d;
  const int64_t n = ix->n_store;
  DevBuf<uint64_t> upper;
  ASL_TRY(upper.reserve((size_t)nq));
  auto out_at = [&](auto *base, int64_t r0, int c0) { return base ? base + (size_t)r0 * k + c0 : nullptr; };
  if (ix->kind == ASL_INDEX_FLAT || ix->kind == ASL_INDEX_IVFFLAT) {
    const bool ivf = ix->kind == ASL_INDEX_IVFFLAT;
    int words = 0;
    if (ivf) {
      nprobe = std::max(1, std::min(nprobe, ix->nlist));
      if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "search: nprobe=%d > %d", nprobe, TK_MAX_K);
      if (!pre_I) ASL_TRY(coarse_search(ix, xq, nq, nprobe));
      const int32_t *cI = pre_I ? pre_I : ix->coarse_I.p;
      ASL_TRY(build_lists(ix));
      words = (ix->nlist + 31) / 32;
      ASL_TRY(ix->bitmap.reserve((size_t)nq * words));
      ASL_TRY(probe_bitmap(cI, nq, nprobe, ix->bitmap.p, words));
      if (n > 0 && prof_counts())
        if (unsigned long long *acc = prof_scanned_dev())
          ASL_TRY(scanned_count(cI, (int64_t)nq * nprobe, ix->list_offsets.p, acc));
    }
    const int64_t ncol = std::max<int64_t>(n, 1);
    int rows = (int)std::min<int64_t>(nq, std::max<int64_t>(1, (int64_t)(SCORE_CHUNK_BYTES / ((size_t)ncol * 4))));
    ASL_TRY(ix->ws_scores.reserve((size_t)rows * ncol));
    for (int r0 = 0; r0 < nq; r0 += rows) {
      const int m = std::min(rows, nq - r0);
      ProfScope ps("scan");
      if (n > 0)
        ASL_TRY(gemm_nt_f32(xq + (size_t)r0 * d, ix->vecs.p, ix->ws_scores.p, m, (int)n, d, d, d, (int)n));
      for (int c0 = 0; c0 < k; c0 += TK_MAX_K) {
        const int kp = std::min<int>(TK_MAX_K, k - c0);
        ASL_TRY(row_topk(ix->ws_scores.p, n, m, (int)n, kp, ix->has_vids ? ix->vids.p : nullptr, 0,
                         ivf ? ix->vlist.p : nullptr, ivf ? ix->bitmap.p + (size_t)r0 * words : nullptr, words,
                         out_at(D, r0, c0), out_at(I64, r0, c0), out_at(I32, r0, c0), k,
                         c0 ? upper.p + r0 : nullptr, upper.p + r0));
      }
    }
    return sync_stream();       // `upper` is freed on return
  }
  nprobe = std::max(1, std::min(nprobe, ix->nlist));
  if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "search: nprobe=%d > %d", nprobe, TK_MAX_K);
  ASL_TRY(build_lists(ix));
  if (!pre_D) ASL_TRY(coarse_search(ix, xq, nq, nprobe));
  const float *cD = pre_D ? pre_D : ix->coarse_D.p;
  const int32_t *cI = pre_D ? pre_I : ix->coarse_I.p;
  {
    ProfScope ps("scan");
    for (int c0 = 0; c0 < k; c0 += TK_MAX_K) {
      const int kp = std::min<int>(TK_MAX_K, k - c0);
      ASL_TRY(pq_scan(xq, nq, d, ix->codebooks.p, ix->pq_m, ix->ksub, ix->dsub, cD, cI, nprobe, ix->list_offsets.p,
                      ix->ids.p, ix->codes.p, kp, out_at(D, 0, c0), out_at(I64, 0, c0), out_at(I32, 0, c0), k,
                      c0 ? upper.p : nullptr, upper.p));
    }
  }
  if (prof_counts())
    if (unsigned long long *acc = prof_scanned_dev())
      ASL_TRY(scanned_count(cI, (int64_t)nq * nprobe, ix->list_offsets.p, acc));
  return sync_stream();
}

int index_search_device(asl_index *ix, int nq, const float *xq, int k, int nprobe, float *D,
                        int64_t *I64, int32_t *I32, const float *pre_D = nullptr,
                        const int32_t *pre_I = nullptr, bool set_mode = false, const int *gate = nullptr,
                        const uint2 *pre_ent = nullptr, const int32_t *pre_cnt = nullptr) {
  if (nq <= 0) return ASL_OK;
  // pre_ent / pre_cnt: the queries as ENTRY LISTS (list_nonzeros / encode_entries_device); xq may
  // then be null -- only the layout-specific scans read their queries in that form, and a row
  // whose count is negative (more than 64 non-zeros) is searched as an all-zero query: the
  // caller watches the producer's n_over
  if (pre_ent && (!pre_cnt || !pre_I || (!xq && !pre_D) || ix->kind == ASL_INDEX_FLAT))
    return fail(ASL_ERR_STATE, "entry-list search: needs the counts, the caller's probe lists (with their scores when "
                               "no dense rows are given) and an IVF index");
  // (with dense rows given as well, entry lists are a hint: a scan that does not read them ignores them)
  if (!xq && !pre_ent) return fail(ASL_ERR_INVALID, "search: null queries");
  // gate: a device-side count -- only the first *gate rows are searched (layout-specific scans only)
  if (gate && (!pre_I || ix->kind == ASL_INDEX_FLAT))
    return fail(ASL_ERR_STATE, "gated search: needs the caller's probe lists and an IVF index");
  if (!ix->trained) return fail(ASL_ERR_STATE, "search: index is not trained");
  if (k <= 0 || k > TK_MAX_K_PASSES) return fail(ASL_ERR_CAPACITY, "search: k=%d outside 1..%d", k, TK_MAX_K_PASSES);
  if (k > TK_MAX_K) {
    // beyond the LDS top-k (the reference's CPU path has no bound on --num_candidates, config.py:188-192;
    // its notebooks evaluate 5 000+ neighbours): ordered dense searches only, in bounded passes
    // (rows asked for as an unordered set are served ordered: a valid answer)
    if (gate || (pre_ent && !xq) || ix->unordered == 2)
      return fail(ASL_ERR_STATE, "search: k=%d > %d is served from dense queries only (no packed keys, "
                                 "entry-list-only queries or gates)", k, TK_MAX_K);
    ix->post_set = ix->post_applied = false;
    return index_search_large_k(ix, nq, xq, k, nprobe, D, I64, I32, pre_D, pre_I);
  }
  const int d = ix->d;
  const int64_t n = ix->n_store;
  if (ix->kind == ASL_INDEX_FLAT || ix->kind == ASL_INDEX_IVFFLAT) {
    const bool ivf = ix->kind == ASL_INDEX_IVFFLAT;
    if (ix->unordered == 2 && !ivf) return fail(ASL_ERR_STATE, "packed-key rows need an IVF index");
    int words = 0;
    if (ivf) {
      nprobe = std::max(1, std::min(nprobe, ix->nlist));
      if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "search: nprobe=%d > %d", nprobe, TK_MAX_K);
      bool own_ent = false;
      if (!pre_I) {
        ASL_TRY(coarse_search(ix, xq, nq, nprobe, nullptr, nullptr, nullptr, nullptr, &own_ent));
      }
      // search_preassigned: the caller's probe lists are read where they lie (device memory that
      // stays valid until the scan has run: the pipeline's per-parity buffers, a caller's tensor on
      // this stream); the coarse scores are unused
      const int32_t *cI = pre_I ? pre_I : ix->coarse_I.p;
      ASL_TRY(build_lists(ix));
      // variant 0: dimension-major postings; 1 (or an unsupported shape): dense GEMM + masked top-k
      const bool use_inv = ix->has_inv && ix->scan_variant == 0 && flat_inv_supported(d, k, nprobe);
      if (ix->unordered == 2 && !(use_inv && I64 && k + FLAT_KEYS_SLACK <= TK_MAX_K))
        return fail(ASL_ERR_STATE, "packed-key rows need the postings scan of IVF-Flat (sparse vectors, k <= 1280) and an int64 output");
      if (gate && !use_inv) return fail(ASL_ERR_STATE, "gated search: needs the postings scan of IVF-Flat");
      if (pre_ent && !use_inv && !xq) return fail(ASL_ERR_STATE, "entry-list search: needs the postings scan of IVF-Flat");
      if (use_inv) {
        const uint2 *q_ent = pre_ent;
        const int32_t *q_cnt = pre_cnt;
        if (!pre_ent && own_ent) {           // the coarse stage of this very call listed them (same stream)
          q_ent = ix->cs_ent.p;
          q_cnt = ix->cs_cnt.p;
        } else if (!pre_ent) {
          ASL_TRY(ix->scan_ent.reserve((size_t)nq * 64));
          ASL_TRY(ix->scan_cnt.reserve((size_t)nq));
          ASL_TRY(ix->scan_over.reserve(1));
          ASL_TRY(list_nonzeros(xq, nq, d, d, ix->scan_ent.p, ix->scan_cnt.p, ix->scan_over.p));
          q_ent = ix->scan_ent.p;
          q_cnt = ix->scan_cnt.p;
        }
        const int mode_ = ix->unordered ? ix->unordered : (set_mode ? 1 : 0);
        ScanPostFilter pf;
        ASL_TRY(take_post_filter(ix, mode_ == 1 && I32 && !I64 && !D && !gate && k + FLAT_KEYS_SLACK <= 2048, ix->ids.p, n, pf));
        {
          ProfScope ps("scan");     // the scan kernel itself
          const bool fx = ix->inv_layout == 2;
          ASL_TRY(flat_inv_scan(ix->inv_layout, xq, nq, d, cI, nprobe, ix->list_offsets.p,
                                ix->blk_offsets.p, ix->blk_base.p,
                                fx ? (const void *)ix->inv_tab8.p : (const void *)ix->inv_tab.p,
                                ix->tab_stride, ix->inv_data.p, ix->ids.p, k, D, I64, I32,
                                mode_, q_ent, q_cnt, gate, &pf));
        }
        if (prof_counts() && !gate) {
          // vectors scored by this launch, summed on the device (nothing waits inside a step)
          if (unsigned long long *acc = prof_scanned_dev())
            ASL_TRY(scanned_count(cI, (int64_t)nq * nprobe, ix->list_offsets.p, acc));
        }
        return ASL_OK;
      }
      words = (ix->nlist + 31) / 32;
      ASL_TRY(ix->bitmap.reserve((size_t)nq * words));
      ASL_TRY(probe_bitmap(cI, nq, nprobe, ix->bitmap.p, words));
      if (n > 0 && prof_counts()) {
        // algorithmic work: vectors in probed lists, summed on the device (nothing waits inside a step)
        if (unsigned long long *acc = prof_scanned_dev())
          ASL_TRY(scanned_count(cI, (int64_t)nq * nprobe, ix->list_offsets.p, acc));
      }
    }
    ix->post_set = ix->post_applied = false;      // the generic kernels take no post-filter
    const int64_t ncol = std::max<int64_t>(n, 1);
    int rows = (int)std::min<int64_t>(nq, std::max<int64_t>(1, (int64_t)(SCORE_CHUNK_BYTES / ((size_t)ncol * 4))));
    ASL_TRY(ix->ws_scores.reserve((size_t)rows * ncol));
    for (int r0 = 0; r0 < nq; r0 += rows) {
      const int m = std::min(rows, nq - r0);
      ProfScope ps("scan");
      if (n > 0)
        ASL_TRY(gemm_nt_f32(xq + (size_t)r0 * d, ix->vecs.p, ix->ws_scores.p, m, (int)n, d, d, d, (int)n));
      ASL_TRY(row_topk(ix->ws_scores.p, n, m, (int)n, k, ix->has_vids ? ix->vids.p : nullptr, 0,
                       ivf ? ix->vlist.p : nullptr, ivf ? ix->bitmap.p + (size_t)r0 * words : nullptr,
                       words, D ? D + (size_t)r0 * k : nullptr, I64 ? I64 + (size_t)r0 * k : nullptr,
                       I32 ? I32 + (size_t)r0 * k : nullptr, k));
    }
    return ASL_OK;
  }
  // IVF-PQ
  nprobe = std::max(1, std::min(nprobe, ix->nlist));
  if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "search: nprobe=%d > %d", nprobe, TK_MAX_K);
  ASL_TRY(build_lists(ix));
  bool own_ent = false;
  if (!pre_D) ASL_TRY(coarse_search(ix, xq, nq, nprobe, nullptr, nullptr, nullptr, nullptr, &own_ent));
  // search_preassigned: the caller's probe lists, read where they lie (see the IVF-Flat branch)
  const float *cD = pre_D ? pre_D : ix->coarse_D.p;
  const int32_t *cI = pre_D ? pre_I : ix->coarse_I.p;
  // exact re-rank: the ADC scan returns k' > k candidates as a set, refine.hip keeps the k best
  const bool refine = ix->refine_k > k && ix->refine_rows && ix->unordered == 0;
  if (gate && refine) return fail(ASL_ERR_STATE, "gated search: not with the exact re-rank");
  if (pre_ent && refine && !xq) return fail(ASL_ERR_STATE, "entry-list search: not with the exact re-rank (it reads the dense queries)");
  float *fin_D = D;
  int64_t *fin_I64 = I64;
  int32_t *fin_I32 = I32;
  const int k_out = k;
  if (refine) {
    if (ix->refine_bad)
      return fail(ASL_ERR_CAPACITY, "search: refine is unavailable, a stored vector has more than %d non-zeros",
                  refine_stride());
    if (ix->r_n != ix->ntotal)
      return fail(ASL_ERR_STATE, "search: refine rows cover %lld of %lld vectors (enable refine before add)",
                  (long long)ix->r_n, (long long)ix->ntotal);
    k = std::min(ix->refine_k, (int)TK_MAX_K);
    ASL_TRY(ix->ws_short.reserve((size_t)nq * k));
    D = nullptr;
    I64 = nullptr;
    I32 = ix->ws_short.p;
    set_mode = true;
  }
  {
    const bool tiled = ix->has_tiles && ix->scan_variant == 0 &&
                       pq_scan_tiled_supported(ix->pq_m, ix->ksub, k, nprobe);
    if (ix->unordered == 2 && !(tiled && I64))
      return fail(ASL_ERR_STATE, "packed-key rows need the tiled IVF-PQ scan (m = 32, 8 bits) and an int64 output");
    if (gate && !tiled) return fail(ASL_ERR_STATE, "gated search: needs the tiled IVF-PQ scan");
    if (pre_ent && !tiled && !xq) return fail(ASL_ERR_STATE, "entry-list search: needs the tiled IVF-PQ scan (m = 32, 8 bits)");
    if (tiled) {
      if (!ix->cbt_ready) {
        const size_t ncb = (size_t)ix->pq_m * ix->ksub * ix->dsub;
        std::vector<float> h((size_t)ncb), ht((size_t)ncb);
        ASL_TRY(ix->codebooks.download(h.data(), ncb));
        ASL_TRY(sync_stream());
        for (int m = 0; m < ix->pq_m; m++)
          for (int c = 0; c < ix->ksub; c++)
            for (int t = 0; t < ix->dsub; t++)
              ht[((size_t)m * ix->dsub + t) * ix->ksub + c] = h[((size_t)m * ix->ksub + c) * ix->dsub + t];
        ASL_TRY(ix->codebooks_t.upload(ht.data(), ncb));
        ASL_TRY(sync_stream());
        ix->cbt_ready = true;
      }
      // the queries' non-zero components as ready lists: the table build of every workgroup
      // starts from 512 bytes instead of listing a 3.2 KB row (17 us per 16 384 queries here,
      // ~4 us saved per (query, shard) workgroup)
      const uint2 *q_ent = pre_ent;
      const int32_t *q_cnt = pre_cnt;
      if (!pre_ent && own_ent) {             // the coarse stage of this very call listed them (same stream)
        q_ent = ix->cs_ent.p;
        q_cnt = ix->cs_cnt.p;
      } else if (!pre_ent) {
        ASL_TRY(ix->scan_ent.reserve((size_t)nq * 64));
        ASL_TRY(ix->scan_cnt.reserve((size_t)nq));
        ASL_TRY(ix->scan_over.reserve(1));
        ASL_TRY(list_nonzeros(xq, nq, d, d, ix->scan_ent.p, ix->scan_cnt.p, ix->scan_over.p));
        q_ent = ix->scan_ent.p;
        q_cnt = ix->scan_cnt.p;
      }
      const int mode_ = ix->unordered ? ix->unordered : (set_mode ? 1 : 0);
      ScanPostFilter pf;
      ASL_TRY(take_post_filter(ix, mode_ == 1 && I32 && !I64 && !D && !gate && !refine && k + 768 <= 2048,
                               ix->ids_tiled.p, ix->n_tile_slots, pf));
      ProfScope ps("scan");     // the scan kernel itself
      ASL_TRY(pq_scan_v3(xq, nq, d, ix->codebooks_t.p, ix->dsub, cD, cI,
                         nprobe, ix->list_offsets.p, ix->tile_offsets.p, ix->codes_tiled.p,
                         ix->ids_tiled.p, k, D, I64, I32, mode_, q_ent, q_cnt, gate, &pf));
    } else {
      ix->post_set = ix->post_applied = false;      // the generic kernel takes no post-filter
      ProfScope ps("scan");
      ASL_TRY(pq_scan(xq, nq, d, ix->codebooks.p, ix->pq_m, ix->ksub, ix->dsub, cD,
                      cI, nprobe, ix->list_offsets.p, ix->ids.p, ix->codes.p, k, D,
                      I64, I32));
    }
  }
  if (prof_counts() && !gate) {
    // vectors scored by this launch, summed on the device (nothing waits inside a step)
    if (unsigned long long *acc = prof_scanned_dev())
      ASL_TRY(scanned_count(cI, (int64_t)nq * nprobe, ix->list_offsets.p, acc));
  }
  if (refine) {
    ProfScope ps("refine");
    ASL_TRY(refine_topk(xq, nq, d, ix->ws_short.p, nullptr, k, ix->r_dim.p, ix->r_val.p, ix->r_cnt.p, ix->r_n,
                        k_out, fin_D, fin_I64, fin_I32));
  }
  return ASL_OK;
}

int index_dim(const asl_index *ix) { return ix->d; }
void index_set_post_filter(asl_index *ix, const IndexPostFilter &p) {
  ix->post = p;
  ix->post_set = true;
  ix->post_applied = false;
}
bool index_post_filter_applied(asl_index *ix) {
  const bool a = ix->post_applied;
  ix->post_set = ix->post_applied = false;
  return a;
}

// The two halves of an IVF search for the two-stream pipeline (search.hip): the coarse
// quantiser into caller-owned buffers, then index_search_device with those as pre_D / pre_I.
int index_nprobe(const asl_index *ix, int nprobe) {
  return ix->kind == ASL_INDEX_FLAT ? 0 : std::max(1, std::min(nprobe, ix->nlist));
}
int index_prepare(asl_index *ix) {   // everything that may allocate or synchronise, up front
  if (!ix->trained) return fail(ASL_ERR_STATE, "search: index is not trained");
  if (ix->kind == ASL_INDEX_FLAT) return ASL_OK;
  if (coarse_sparse_supported(ix->d, ix->nlist)) ASL_TRY(coarse_transposed(ix));
  return build_lists(ix);
}
int index_coarse_device(asl_index *ix, int nq, const float *xq, int nprobe, float *out_D,
                        int32_t *out_I, uint2 *ent_out, int32_t *cnt_out, bool *have_ent) {
  return coarse_search(ix, xq, nq, nprobe, out_D, out_I, ent_out, cnt_out, have_ent);
}
int index_agreed_keys(const asl_index *ix, int k, int np, int world) {
  return (ix->agreed_val >= 0 && ix->agreed_k == k && ix->agreed_np == np && ix->agreed_world == world) ? ix->agreed_val : -1;
}
void index_set_agreed_keys(asl_index *ix, int k, int np, int world, int v) {
  ix->agreed_k = k;
  ix->agreed_np = np;
  ix->agreed_world = world;
  ix->agreed_val = v;
}
int index_shard_world(const asl_index *ix, int *rank) {
  if (rank) *rank = ix->shard_rank;
  return ix->kind == ASL_INDEX_FLAT ? 0 : ix->shard_world;
}

}  // namespace asl

extern "C" {

asl_index_t *asl_index_create(int32_t d, int32_t nlist, int32_t kind, int32_t pq_m,
                              int32_t pq_bits) {
  clear_error();
  if (d <= 0 || kind < ASL_INDEX_FLAT || kind > ASL_INDEX_IVFPQ) {
    fail(ASL_ERR_INVALID, "index_create: bad d/kind");
    return nullptr;
  }
  if (kind != ASL_INDEX_FLAT && nlist <= 0) {
    fail(ASL_ERR_INVALID, "index_create: nlist must be positive");
    return nullptr;
  }
  if (kind == ASL_INDEX_IVFPQ) {
    if (pq_bits <= 0) pq_bits = 8;
    if (pq_bits > 8 || pq_m <= 0 || d % pq_m != 0 ||
        !(pq_m == 4 || pq_m == 8 || pq_m == 16 || pq_m == 32 || pq_m == 64)) {
      fail(ASL_ERR_INVALID, "index_create: pq_m must be 4/8/16/32/64 and divide d; pq_bits <= 8");
      return nullptr;
    }
  }
  if (ensure_device() != ASL_OK) return nullptr;
  asl_index *ix = new asl_index();
  ix->d = d;
  ix->nlist = kind == ASL_INDEX_FLAT ? 0 : nlist;
  ix->kind = kind;
  if (kind == ASL_INDEX_IVFPQ) {
    ix->pq_m = pq_m;
    ix->pq_bits = pq_bits;
    ix->ksub = 1 << pq_bits;
    ix->dsub = d / pq_m;
  }
  ix->trained = kind == ASL_INDEX_FLAT;
  return ix;
}

void asl_index_free(asl_index_t *ix) { delete ix; }

int asl_index_set_unordered(asl_index_t *ix, int32_t unordered) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "set_unordered: null index");
  if (unordered < 0 || unordered > 2) return fail(ASL_ERR_INVALID, "set_unordered: mode must be 0, 1 or 2");
  ix->unordered = unordered;
  return ASL_OK;
}

int asl_index_set_flat_storage(asl_index_t *ix, int32_t mode) {
  clear_error();
  if (!ix || ix->kind != ASL_INDEX_IVFFLAT) return fail(ASL_ERR_INVALID, "set_flat_storage: an IVF-Flat index is required");
  if (mode != ASL_FLAT_FX22 && mode != ASL_FLAT_F32) return fail(ASL_ERR_INVALID, "set_flat_storage: ASL_FLAT_FX22 or ASL_FLAT_F32");
  if (ix->ntotal > 0 && mode != ix->flat_storage)
    return fail(ASL_ERR_STATE, "set_flat_storage: set before add() (stored components are rounded as they arrive)");
  ix->flat_storage = mode;
  ix->agreed_val = -1;      // what asl_index_supports_keys answers may change: the ranks agree again
  return ASL_OK;
}

int asl_index_get_flat_storage(const asl_index_t *ix) { return ix ? ix->flat_storage : ASL_FLAT_F32; }

int asl_index_flat_layout(asl_index_t *ix) {
  clear_error();
  if (!ix || ix->kind != ASL_INDEX_IVFFLAT) return fail(ASL_ERR_INVALID, "flat_layout: an IVF-Flat index is required");
  if (ix->trained && ix->n_store > 0) {
    ASL_TRY(ensure_device());
    ASL_TRY(build_lists(ix));
  }
  return ix->has_inv ? ix->inv_layout : 0;
}

int asl_index_set_scan_variant(asl_index_t *ix, int32_t variant) {
  clear_error();
  if (!ix || variant < 0 || variant > 1)
    return fail(ASL_ERR_INVALID, "set_scan_variant: 0 (layout-specific scan) or 1 (generic kernels)");
  ix->scan_variant = variant;
  ix->agreed_val = -1;      // asl_index_supports_keys depends on the variant: the ranks agree again
  return ASL_OK;
}

// 1 when asl_index_search_preassigned can emit packed 64-bit keys (unordered mode 2) for this
// index at (k, nprobe): the predicate index_search_device applies, for callers that must
// choose the exchange format up front (ann_solo_amd/distributed.py).
// IVF-Flat: the answer depends on the vectors THIS handle stores (has_inv: an empty or dense shard
// has no postings), so a stale layout is rebuilt first -- the value is then what a search meets --
// and sharded drivers agree on it across ranks before they pick the exchange format.
int asl_index_supports_keys(asl_index_t *ix, int32_t k, int32_t nprobe) {
  if (!ix) return 0;
  nprobe = std::max(1, std::min(nprobe, ix->nlist));
  if (ix->kind == ASL_INDEX_IVFFLAT) {    // the postings scan's set finish (flat_scan.hip)
    if (ix->lists_dirty && ix->trained && (ensure_device() != ASL_OK || build_lists(ix) != ASL_OK)) return 0;
    return ix->scan_variant == 0 && ix->has_inv && flat_inv_supported(ix->d, k, nprobe) &&
           k + 768 <= TK_MAX_K;
  }
  if (ix->kind != ASL_INDEX_IVFPQ) return 0;
  return ix->scan_variant == 0 && pq_scan_tiled_supported(ix->pq_m, ix->ksub, k, nprobe) &&
         k + 768 <= TK_MAX_K;
}

// the shards' own k (exchange.hip "shard-side k_s < k"; profiles/r05_sim_rank.txt: k / 2 at 8 ranks takes
// 0.3 (IVF-PQ) / 0.9 ms (IVF-Flat) off the shard scan, 0.25 % / 0.04 % of a shard's rows are scanned a second time)
int32_t asl_shard_k(int32_t k, int32_t world) {
  if (k < 1 || world < 4) return k;
  const int raw = world >= 8 ? (k + 1) / 2 : (5 * k + 7) / 8;
  const int ks = std::min(k, (raw + 63) / 64 * 64);
  const int head = std::min(k, (2 * k + world - 1) / world);
  return ks > head ? ks : k;
}

int asl_index_set_niter(asl_index_t *ix, int32_t niter) {
  if (!ix || niter < 0) return fail(ASL_ERR_INVALID, "set_niter");
  ix->niter = niter;
  return ASL_OK;
}

int asl_index_train(asl_index_t *ix, int64_t n, const float *x, uint64_t seed) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "train: null index");
  ASL_TRY(ensure_device());
  if (ix->kind == ASL_INDEX_FLAT) return ASL_OK;
  if (n <= 0 || !x) return fail(ASL_ERR_INVALID, "train: no data");
  if (n < ix->nlist) return fail(ASL_ERR_INVALID, "train: %lld vectors < nlist %d", (long long)n, ix->nlist);
  In<float> dx;
  ASL_TRY(dx.init(x, (size_t)n * ix->d));
  ASL_TRY(ix->centroids.reserve((size_t)ix->nlist * ix->d));
  ASL_TRY(kmeans_device(ix, dx.d, n, ix->d, ix->d, ix->nlist, ix->niter, seed, false, 256, ix->centroids.p));
  if (ix->kind == ASL_INDEX_IVFPQ) {
    if (n < ix->ksub) return fail(ASL_ERR_INVALID, "train: %lld vectors < 2^pq_bits", (long long)n);
    ASL_TRY(pq_train_device(ix, dx.d, n, seed + 7));
  }
  ASL_TRY(sync_stream());
  ix->trained = true;
  ix->cbt_ready = false;
  ix->cent_t_ready = false;
  return ASL_OK;
}

int asl_index_set_trained(asl_index_t *ix, const float *centroids, const float *codebooks) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "set_trained: null index");
  ASL_TRY(ensure_device());
  if (ix->kind != ASL_INDEX_FLAT) {
    if (!centroids) return fail(ASL_ERR_INVALID, "set_trained: centroids required");
    ASL_TRY(ix->centroids.upload(centroids, (size_t)ix->nlist * ix->d));
  }
  if (ix->kind == ASL_INDEX_IVFPQ) {
    if (!codebooks) return fail(ASL_ERR_INVALID, "set_trained: codebooks required");
    ASL_TRY(ix->codebooks.upload(codebooks, (size_t)ix->pq_m * ix->ksub * ix->dsub));
  }
  ASL_TRY(sync_stream());
  ix->trained = true;
  ix->cbt_ready = false;
  ix->cent_t_ready = false;
  return ASL_OK;
}

static int index_add_impl(asl_index_t *ix, int64_t n, const float *x, const int32_t *lists);

int asl_index_add(asl_index_t *ix, int64_t n, const float *x) { return index_add_impl(ix, n, x, nullptr); }

// add() with the inverted list of every vector supplied by the caller instead of computed by the
// coarse quantiser: what re-creating an index from its stored inverted lists needs (an imported
// FAISS file keeps FAISS' own assignments, bit for bit whatever its BLAS rounded).
int asl_index_add_preassigned(asl_index_t *ix, int64_t n, const float *x, const int32_t *lists) {
  if (!lists) return fail(ASL_ERR_INVALID, "add_preassigned: null list assignment");
  if (!ix || ix->kind == ASL_INDEX_FLAT) return fail(ASL_ERR_INVALID, "add_preassigned: an IVF index is required");
  return index_add_impl(ix, n, x, lists);
}

static int index_add_impl(asl_index_t *ix, int64_t n, const float *x, const int32_t *lists) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "add: null index");
  if (!ix->trained) return fail(ASL_ERR_STATE, "add: index is not trained");
  if (ix->shard_world > 1) return fail(ASL_ERR_STATE, "add: index is already sharded");
  if (n <= 0) return ASL_OK;
  if (!x) return fail(ASL_ERR_INVALID, "add: null data");
  if (ix->ntotal + n > 0x7fffffffLL) return fail(ASL_ERR_CAPACITY, "add: more than 2^31-1 vectors");
  ASL_TRY(ensure_device());
  In<float> dx;
  ASL_TRY(dx.init(x, (size_t)n * ix->d));
  if (ix->kind != ASL_INDEX_FLAT) {
    ASL_TRY(ix->ws_assign.reserve((size_t)n));
    if (lists) {
      std::vector<int32_t> h((size_t)n);
      HIP_TRY(hipMemcpy(h.data(), lists, (size_t)n * 4, hipMemcpyDefault));
      for (int64_t i = 0; i < n; i++)
        if (h[(size_t)i] < 0 || h[(size_t)i] >= ix->nlist)
          return fail(ASL_ERR_INVALID, "add_preassigned: list %d of vector %lld outside 0..%d", h[(size_t)i],
                      (long long)i, ix->nlist - 1);
      ASL_TRY(ix->ws_assign.upload(h.data(), (size_t)n));
      ASL_TRY(sync_stream());
    } else {
      ASL_TRY(assign_ip(ix, dx.d, ix->d, n, ix->centroids.p, ix->nlist, ix->d, ix->ws_assign.p));
    }
    ASL_TRY(dev_append(ix->vlist, (size_t)ix->n_store, ix->ws_assign.p, (size_t)n));
  }
  if (ix->kind == ASL_INDEX_IVFPQ) {
    DevBuf<uint8_t> codes;
    ASL_TRY(codes.reserve((size_t)n * ix->pq_m));
    ASL_TRY(pq_encode(dx.d, ix->ws_assign.p, ix->centroids.p, ix->codebooks.p, n, ix->d, ix->pq_m,
                      ix->ksub, ix->dsub, codes.p));
    ASL_TRY(dev_append(ix->codes_add, (size_t)ix->n_store * ix->pq_m, codes.p, (size_t)n * ix->pq_m));
    ASL_TRY(sync_stream());
    if (ix->refine_rows) {
      const size_t S = (size_t)refine_stride(), have = (size_t)ix->r_n, want = have + (size_t)n;
      auto grow = [&](auto &buf, size_t per) -> int {
        using T = typename std::remove_reference<decltype(*buf.p)>::type;
        if (want * per <= buf.cap) return ASL_OK;
        DevBuf<T> nb;
        ASL_TRY(nb.reserve(std::max(want * per, buf.cap + buf.cap / 2)));
        if (have) HIP_TRY(hipMemcpyAsync(nb.p, buf.p, have * per * sizeof(T), hipMemcpyDeviceToDevice, stream()));
        ASL_TRY(sync_stream());
        buf = std::move(nb);
        return ASL_OK;
      };
      ASL_TRY(grow(ix->r_dim, S));
      ASL_TRY(grow(ix->r_val, S));
      ASL_TRY(grow(ix->r_cnt, 1));
      DevBuf<int> st;
      ASL_TRY(st.reserve(1));
      HIP_TRY(hipMemsetAsync(st.p, 0, sizeof(int), stream()));
      ASL_TRY(refine_append_rows(dx.d, n, ix->d, ix->r_n, ix->r_dim.p, ix->r_val.p, ix->r_cnt.p, st.p));
      int h = 0;
      HIP_TRY(hipMemcpyAsync(&h, st.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
      ASL_TRY(sync_stream());
      if (h) ix->refine_bad = true;
      ix->r_n += n;
    }
  } else {
    ASL_TRY(dev_append(ix->vecs, (size_t)ix->n_store * ix->d, dx.d, (size_t)n * ix->d));
    if (ix->kind == ASL_INDEX_IVFFLAT && ix->flat_storage == ASL_FLAT_FX22)      // stored components: 22-bit fixed point
      ASL_TRY(quantize_fx22(ix->vecs.p + (size_t)ix->n_store * ix->d, n * ix->d));
  }
  ASL_TRY(sync_stream());
  ix->n_store += n;
  ix->ntotal += n;
  ix->lists_dirty = true;
  ix->agreed_val = -1;
  return ASL_OK;
}

int asl_index_set_refine(asl_index_t *ix, int32_t kprime) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "set_refine: null index");
  if (ix->kind != ASL_INDEX_IVFPQ) return fail(ASL_ERR_INVALID, "set_refine: an IVF-PQ index is required");
  if (kprime < 0 || kprime > TK_MAX_K) return fail(ASL_ERR_INVALID, "set_refine: k' must be in 0..%d", TK_MAX_K);
  if (kprime > 0 && !ix->refine_rows) {
    if (ix->ntotal > 0)
      return fail(ASL_ERR_STATE, "set_refine: enable before add() (the exact rows are stored as vectors arrive)");
    ix->refine_rows = true;
  }
  ix->refine_k = kprime;
  ix->agreed_val = -1;
  return ASL_OK;
}

int asl_index_get_refine(const asl_index_t *ix) { return ix ? ix->refine_k : 0; }

}  // extern "C"

namespace asl {
// sharded.hip: k' when the exact re-rank is usable, else 0; and the re-rank of merged device rows
int index_refine_k(const asl_index *ix) {
  return (ix->refine_rows && !ix->refine_bad && ix->r_n == ix->ntotal) ? ix->refine_k : 0;
}
int index_swap_unordered(asl_index *ix, int mode, int *prev) {
  if (prev) *prev = ix->unordered;
  ix->unordered = mode;
  return ASL_OK;
}
int index_refine_device(asl_index *ix, int nq, const float *xq, int kp, const int64_t *I_in, int k,
                        float *D, int64_t *I) {
  return refine_topk(xq, nq, ix->d, nullptr, I_in, kp, ix->r_dim.p, ix->r_val.p, ix->r_cnt.p, ix->r_n, k, D, I,
                     nullptr);
}
}  // namespace asl

extern "C" {

int asl_index_refine(asl_index_t *ix, int32_t nq, const float *xq, int32_t kp, const int64_t *I_in,
                     int32_t k, float *D, int64_t *I) {
  clear_error();
  if (!ix || !xq || !I_in || !I) return fail(ASL_ERR_INVALID, "refine: null argument");
  if (!ix->refine_rows || ix->r_n != ix->ntotal)
    return fail(ASL_ERR_STATE, "refine: the index stores no exact rows (asl_index_set_refine before add)");
  if (ix->refine_bad) return fail(ASL_ERR_CAPACITY, "refine: a stored vector has more than %d non-zeros", refine_stride());
  if (nq <= 0) return ASL_OK;
  if (k <= 0 || k > kp || kp > TK_MAX_K) return fail(ASL_ERR_INVALID, "refine: need 0 < k <= k' <= %d", TK_MAX_K);
  ASL_TRY(ensure_device());
  In<float> dq;
  In<int64_t> dI;
  Out<float> oD;
  Out<int64_t> oI;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(dI.init(I_in, (size_t)nq * kp));
  ASL_TRY(oD.init(D, (size_t)nq * k));
  ASL_TRY(oI.init(I, (size_t)nq * k));
  ASL_TRY(refine_topk(dq.d, nq, ix->d, nullptr, dI.d, kp, ix->r_dim.p, ix->r_val.p, ix->r_cnt.p, ix->r_n,
                      k, oD.d, oI.d, nullptr));
  ASL_TRY(oD.finish());
  ASL_TRY(oI.finish());
  if (oD.to_host() || oI.to_host() || dq.own.p || dI.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

int asl_index_reset(asl_index_t *ix) {
  if (!ix) return fail(ASL_ERR_INVALID, "reset: null index");
  ix->vecs.release();
  ix->vlist.release();
  ix->vids.release();
  ix->codes_add.release();
  ix->codes.release();
  ix->ids.release();
  ix->ws_scores.release();
  ix->ntotal = 0;
  ix->n_store = 0;
  ix->r_n = 0;
  ix->refine_bad = false;
  ix->has_vids = false;
  ix->shard_rank = 0;
  ix->shard_world = 1;
  ix->lists_dirty = true;
  return ASL_OK;
}

int64_t asl_index_ntotal(const asl_index_t *ix) { return ix ? ix->ntotal : 0; }
int asl_index_is_trained(const asl_index_t *ix) { return ix && ix->trained; }

int asl_index_info(const asl_index_t *ix, asl_index_info_t *info) {
  if (!ix || !info) return fail(ASL_ERR_INVALID, "info: null");
  info->d = ix->d;
  info->nlist = ix->nlist;
  info->kind = ix->kind;
  info->pq_m = ix->pq_m;
  info->pq_ksub = ix->ksub;
  info->pq_dsub = ix->dsub;
  info->ntotal = ix->ntotal;
  info->nlocal = ix->n_store;
  info->trained = ix->trained;
  info->shard_rank = ix->shard_rank;
  info->shard_world = ix->shard_world;
  return ASL_OK;
}

int asl_index_get_centroids(const asl_index_t *ix, float *out) {
  clear_error();
  if (!ix || !out || ix->kind == ASL_INDEX_FLAT || !ix->trained)
    return fail(ASL_ERR_STATE, "get_centroids: not available");
  HIP_TRY(hipMemcpyAsync(out, ix->centroids.p, (size_t)ix->nlist * ix->d * 4, hipMemcpyDefault, stream()));
  return sync_stream();
}

int asl_index_get_codebooks(const asl_index_t *ix, float *out) {
  clear_error();
  if (!ix || !out || ix->kind != ASL_INDEX_IVFPQ || !ix->trained)
    return fail(ASL_ERR_STATE, "get_codebooks: not available");
  HIP_TRY(hipMemcpyAsync(out, ix->codebooks.p, (size_t)ix->pq_m * ix->ksub * ix->dsub * 4, hipMemcpyDefault, stream()));
  return sync_stream();
}

int asl_index_get_lists(const asl_index_t *cix, int32_t *list_offsets, int32_t *ids,
                        uint8_t *codes, float *vecs) {
  clear_error();
  asl_index *ix = const_cast<asl_index *>(cix);
  if (!ix || ix->kind == ASL_INDEX_FLAT) return fail(ASL_ERR_STATE, "get_lists: IVF index required");
  ASL_TRY(build_lists(ix));
  const int64_t n = ix->n_store;
  if (list_offsets)
    HIP_TRY(hipMemcpyAsync(list_offsets, ix->list_offsets.p, ((size_t)ix->nlist + 1) * 4, hipMemcpyDefault, stream()));
  if (ids && n) HIP_TRY(hipMemcpyAsync(ids, ix->ids.p, (size_t)n * 4, hipMemcpyDefault, stream()));
  if (codes && n) {
    if (ix->kind != ASL_INDEX_IVFPQ) return fail(ASL_ERR_STATE, "get_lists: no PQ codes in this index");
    HIP_TRY(hipMemcpyAsync(codes, ix->codes.p, (size_t)n * ix->pq_m, hipMemcpyDefault, stream()));
  }
  if (vecs && n) {
    if (ix->kind != ASL_INDEX_IVFFLAT) return fail(ASL_ERR_STATE, "get_lists: no flat vectors in this index");
    // list order = stable sort of add order by list: reuse ids when unsharded
    std::vector<int32_t> h_vlist((size_t)n);
    ASL_TRY(ix->vlist.download(h_vlist.data(), (size_t)n));
    ASL_TRY(sync_stream());
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return h_vlist[(size_t)a] < h_vlist[(size_t)b]; });
    DevBuf<int64_t> od;
    DevBuf<float> tmp;
    ASL_TRY(od.upload(order.data(), (size_t)n));
    ASL_TRY(tmp.reserve((size_t)n * ix->d));
    ASL_TRY(gather_rows_f32(ix->vecs.p, ix->d, od.p, n, ix->d, tmp.p, ix->d));
    HIP_TRY(hipMemcpyAsync(vecs, tmp.p, (size_t)n * ix->d * 4, hipMemcpyDefault, stream()));
    ASL_TRY(sync_stream());
  }
  return sync_stream();
}

static void lpt_owner(const std::vector<int64_t> &sizes, int world, std::vector<int32_t> &owner) {
  const int nlist = (int)sizes.size();
  std::vector<int> order((size_t)nlist);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[(size_t)a] > sizes[(size_t)b]; });
  std::vector<int64_t> load((size_t)world, 0);
  owner.assign((size_t)nlist, 0);
  for (int l : order) {
    int best = 0;
    for (int r = 1; r < world; r++)
      if (load[(size_t)r] < load[(size_t)best]) best = r;
    owner[(size_t)l] = best;
    load[(size_t)best] += sizes[(size_t)l];
  }
}

static int list_sizes(asl_index *ix, std::vector<int32_t> &h_vlist, std::vector<int64_t> &sizes) {
  const int64_t n = ix->n_store;
  h_vlist.resize((size_t)n);
  if (n) {
    ASL_TRY(ix->vlist.download(h_vlist.data(), (size_t)n));
    ASL_TRY(sync_stream());
  }
  sizes.assign((size_t)ix->nlist, 0);
  for (int64_t i = 0; i < n; i++) sizes[(size_t)h_vlist[(size_t)i]]++;
  return ASL_OK;
}

int asl_lpt_owner(int32_t nlist, const int64_t *sizes, int32_t world, int32_t *owner_out) {
  clear_error();
  if (nlist < 0 || world <= 0 || (nlist && (!sizes || !owner_out)))
    return fail(ASL_ERR_INVALID, "lpt_owner: bad arguments");
  std::vector<int64_t> sz(sizes, sizes + nlist);
  std::vector<int32_t> owner;
  lpt_owner(sz, world, owner);
  if (nlist) memcpy(owner_out, owner.data(), owner.size() * 4);
  return ASL_OK;
}

// Ownership balances the EXPECTED SCAN LOAD, not the stored vectors: a list is probed
// roughly in proportion to its population (dense regions attract queries as well as library
// spectra), so its expected contribution to a query's scan is ~ size^2. Measured on the
// 2.1M-spectrum bench library, 8 shards: max/mean scanned vectors 1.094 with weights = size,
// 1.039 with size^2 (stored vectors then differ by +-5 %).
static void shard_owner(const std::vector<int64_t> &sizes, int world, std::vector<int32_t> &owner) {
  std::vector<int64_t> w(sizes.size());
  for (size_t i = 0; i < sizes.size(); i++) w[i] = sizes[i] * sizes[i];
  lpt_owner(w, world, owner);
}

int asl_index_shard_map(const asl_index_t *cix, int32_t world, int32_t *owner_out) {
  clear_error();
  asl_index *ix = const_cast<asl_index *>(cix);
  if (!ix || ix->kind == ASL_INDEX_FLAT || world <= 0 || !owner_out)
    return fail(ASL_ERR_INVALID, "shard_map: IVF index and world > 0 required");
  if (ix->shard_world > 1) return fail(ASL_ERR_STATE, "shard_map: call before sharding");
  std::vector<int32_t> h_vlist, owner;
  std::vector<int64_t> sizes;
  ASL_TRY(list_sizes(ix, h_vlist, sizes));
  shard_owner(sizes, world, owner);
  memcpy(owner_out, owner.data(), owner.size() * 4);
  return ASL_OK;
}

int asl_index_shard(asl_index_t *ix, int32_t rank, int32_t world) {
  clear_error();
  if (!ix || ix->kind == ASL_INDEX_FLAT) return fail(ASL_ERR_INVALID, "shard: IVF index required");
  if (world <= 0 || rank < 0 || rank >= world) return fail(ASL_ERR_INVALID, "shard: bad rank/world");
  if (ix->shard_world > 1) return fail(ASL_ERR_STATE, "shard: already sharded");
  if (world == 1) return ASL_OK;
  std::vector<int32_t> h_vlist, owner;
  std::vector<int64_t> sizes;
  ASL_TRY(list_sizes(ix, h_vlist, sizes));
  shard_owner(sizes, world, owner);
  const int64_t n = ix->n_store;
  std::vector<int64_t> keep;
  std::vector<int32_t> keep32, new_vlist;
  for (int64_t i = 0; i < n; i++)
    if (owner[(size_t)h_vlist[(size_t)i]] == rank) {
      keep.push_back(i);
      keep32.push_back((int32_t)i);
      new_vlist.push_back(h_vlist[(size_t)i]);
    }
  const int64_t nk = (int64_t)keep.size();
  if (ix->kind == ASL_INDEX_IVFFLAT) {
    DevBuf<int64_t> kd;
    DevBuf<float> nv;
    ASL_TRY(kd.upload(keep.data(), (size_t)nk));
    ASL_TRY(nv.reserve((size_t)std::max<int64_t>(nk, 1) * ix->d));
    ASL_TRY(gather_rows_f32(ix->vecs.p, ix->d, kd.p, nk, ix->d, nv.p, ix->d));
    ASL_TRY(sync_stream());
    ix->vecs = std::move(nv);
  } else {
    DevBuf<int32_t> kd;
    DevBuf<uint8_t> nc;
    ASL_TRY(kd.upload(keep32.data(), (size_t)nk));
    ASL_TRY(nc.reserve((size_t)std::max<int64_t>(nk, 1) * ix->pq_m));
    ASL_TRY(gather_rows_u8(ix->codes_add.p, kd.p, nk, ix->pq_m, nc.p));
    ASL_TRY(sync_stream());
    ix->codes_add = std::move(nc);
  }
  ASL_TRY(ix->vlist.upload(new_vlist.data(), (size_t)nk));
  ASL_TRY(ix->vids.upload(keep32.data(), (size_t)nk));
  ASL_TRY(sync_stream());
  ix->has_vids = true;
  ix->n_store = nk;
  ix->shard_rank = rank;
  ix->shard_world = world;
  ix->lists_dirty = true;
  return ASL_OK;
}

int asl_index_coarse(asl_index_t *ix, int32_t nq, const float *xq, int32_t nprobe,
                     float *coarse_D, int32_t *coarse_I) {
  clear_error();
  if (!ix || ix->kind == ASL_INDEX_FLAT || !ix->trained)
    return fail(ASL_ERR_STATE, "coarse: trained IVF index required");
  if (nq <= 0) return ASL_OK;
  nprobe = std::max(1, std::min(nprobe, ix->nlist));
  if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "coarse: nprobe=%d > %d", nprobe, TK_MAX_K);
  In<float> dq;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(coarse_search(ix, dq.d, nq, nprobe));
  if (coarse_D)
    HIP_TRY(hipMemcpyAsync(coarse_D, ix->coarse_D.p, (size_t)nq * nprobe * 4, hipMemcpyDefault, stream()));
  if (coarse_I)
    HIP_TRY(hipMemcpyAsync(coarse_I, ix->coarse_I.p, (size_t)nq * nprobe * 4, hipMemcpyDefault, stream()));
  return sync_stream();
}

int asl_index_postings_work(asl_index_t *ix, int32_t nq, const float *xq, int32_t nprobe,
                            int64_t *bytes, int64_t *lines) {
  clear_error();
  if (!ix || ix->kind != ASL_INDEX_IVFFLAT || !ix->trained)
    return fail(ASL_ERR_STATE, "postings_work: trained IVF-Flat index required");
  if (bytes) *bytes = 0;
  if (lines) *lines = 0;
  if (nq <= 0) return ASL_OK;
  ASL_TRY(ensure_device());
  nprobe = std::max(1, std::min(nprobe, ix->nlist));
  if (nprobe > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "postings_work: nprobe=%d > %d", nprobe, TK_MAX_K);
  ASL_TRY(build_lists(ix));
  if (!ix->has_inv) return fail(ASL_ERR_STATE, "postings_work: the index holds no postings (dense vectors)");
  In<float> dq;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(coarse_search(ix, dq.d, nq, nprobe));
  DevBuf<unsigned long long> acc;
  ASL_TRY(acc.reserve(2));
  HIP_TRY(hipMemsetAsync(acc.p, 0, 16, stream()));
  if (ix->inv_layout == 2)
    ASL_TRY(flat_fx_work(dq.d, nq, ix->d, ix->coarse_I.p, nprobe, ix->blk_offsets.p, ix->inv_tab8.p,
                         ix->tab_stride, ix->inv_cnt16.p, acc.p));
  else
    ASL_TRY(flat_inv_work(dq.d, nq, ix->d, ix->coarse_I.p, nprobe, ix->blk_offsets.p, ix->inv_tab.p, acc.p));
  unsigned long long h[2] = {0, 0};
  ASL_TRY(acc.download(h, 2));
  ASL_TRY(sync_stream());
  if (bytes) *bytes = (int64_t)h[0];
  if (lines) *lines = (int64_t)h[1];
  return ASL_OK;
}

int asl_index_pq_lut(asl_index_t *ix, int32_t nq, const float *xq, float *lut) {
  clear_error();
  if (!ix || ix->kind != ASL_INDEX_IVFPQ || !ix->trained)
    return fail(ASL_ERR_STATE, "pq_lut: trained IVF-PQ index required");
  if (nq <= 0) return ASL_OK;
  In<float> dq;
  Out<float> dl;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(dl.init(lut, (size_t)nq * ix->pq_m * ix->ksub));
  ASL_TRY(pq_lut(dq.d, nq, ix->d, ix->codebooks.p, ix->pq_m, ix->ksub, ix->dsub, dl.d));
  ASL_TRY(dl.finish());
  return sync_stream();
}

int asl_index_search(asl_index_t *ix, int32_t nq, const float *xq, int32_t k, int32_t nprobe,
                     float *D, int64_t *I) {
  clear_error();
  if (!ix) return fail(ASL_ERR_INVALID, "search: null index");
  if (nq <= 0) return ASL_OK;
  if (!xq || !I) return fail(ASL_ERR_INVALID, "search: null xq / I");
  ASL_TRY(ensure_device());
  In<float> dq;
  Out<float> dD;
  Out<int64_t> dI;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(dD.init(D, (size_t)nq * k));
  ASL_TRY(dI.init(I, (size_t)nq * k));
  ASL_TRY(index_search_device(ix, nq, dq.d, k, nprobe, dD.d, dI.d, nullptr));
  ASL_TRY(dD.finish());
  ASL_TRY(dI.finish());
  if (dD.to_host() || dI.to_host() || dq.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

int asl_index_search_preassigned(asl_index_t *ix, int32_t nq, const float *xq, int32_t k,
                                 int32_t nprobe, const float *coarse_D,
                                 const int32_t *coarse_I, float *D, int64_t *I) {
  clear_error();
  if (!ix || ix->kind == ASL_INDEX_FLAT)
    return fail(ASL_ERR_INVALID, "search_preassigned: IVF index required");
  if (nq <= 0) return ASL_OK;
  if (!xq || !I || !coarse_D || !coarse_I) return fail(ASL_ERR_INVALID, "search_preassigned: null argument");
  if (nprobe < 1 || nprobe > ix->nlist) return fail(ASL_ERR_INVALID, "search_preassigned: nprobe outside 1..nlist");
  ASL_TRY(ensure_device());
  In<float> dq, dcD;
  In<int32_t> dcI;
  Out<float> dD;
  Out<int64_t> dI;
  ASL_TRY(dq.init(xq, (size_t)nq * ix->d));
  ASL_TRY(dcD.init(coarse_D, (size_t)nq * nprobe));
  ASL_TRY(dcI.init(coarse_I, (size_t)nq * nprobe));
  ASL_TRY(dD.init(D, (size_t)nq * k));
  ASL_TRY(dI.init(I, (size_t)nq * k));
  ASL_TRY(index_search_device(ix, nq, dq.d, k, nprobe, dD.d, dI.d, nullptr, dcD.d, dcI.d));
  ASL_TRY(dD.finish());
  ASL_TRY(dI.finish());
  if (dD.to_host() || dI.to_host() || dq.own.p || dcD.own.p || dcI.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

// search_preassigned over a list whose length only the device knows: a launch for `cap` rows
// of which the first *count (device memory) are searched; the other rows of D / I are left
// untouched. Device pointers only; never waits.
int asl_index_search_gated(asl_index_t *ix, int32_t cap, const float *xq, int32_t k, int32_t nprobe,
                           const float *coarse_D, const int32_t *coarse_I, float *D, int64_t *I,
                           const int32_t *count) {
  clear_error();
  if (!ix || ix->kind == ASL_INDEX_FLAT) return fail(ASL_ERR_INVALID, "search_gated: IVF index required");
  if (cap <= 0) return ASL_OK;
  if (!xq || !I || !coarse_D || !coarse_I || !count) return fail(ASL_ERR_INVALID, "search_gated: null argument");
  if (nprobe < 1 || nprobe > ix->nlist) return fail(ASL_ERR_INVALID, "search_gated: nprobe outside 1..nlist");
  ASL_TRY(ensure_device());
  if (!is_device_ptr(xq) || !is_device_ptr(I) || !is_device_ptr(coarse_D) || !is_device_ptr(coarse_I) ||
      !is_device_ptr(count) || (D && !is_device_ptr(D)))
    return fail(ASL_ERR_INVALID, "search_gated: device pointers only");
  return index_search_device(ix, cap, xq, k, nprobe, D, I, nullptr, coarse_D, coarse_I, false,
                             reinterpret_cast<const int *>(count));
}

// search_preassigned with the queries as ENTRY LISTS (asl_encode_entries_batch: entries [nq][64]
// word pairs, counts [nq]) instead of dense rows: what the layout-specific scans read anyway -- 512
// bytes per query instead of 3.2 KB, and no listing pass. Results are those of
// asl_index_search_preassigned on the dense rows, bit for bit. A row whose count is negative (more
// than 64 non-zeros) is searched as an all-zero query: check the n_over the encoder reported. `count`
// (may be null): a device int -- only the first *count rows are searched, the launch covers nq
// (asl_index_search_gated). Device pointers only; never waits.
int asl_index_search_entries(asl_index_t *ix, int32_t nq, const uint32_t *entries, const int32_t *counts,
                             int32_t k, int32_t nprobe, const float *coarse_D, const int32_t *coarse_I,
                             float *D, int64_t *I, const int32_t *count) {
  clear_error();
  if (!ix || ix->kind == ASL_INDEX_FLAT) return fail(ASL_ERR_INVALID, "search_entries: IVF index required");
  if (nq <= 0) return ASL_OK;
  if (!entries || !counts || !I || !coarse_D || !coarse_I) return fail(ASL_ERR_INVALID, "search_entries: null argument");
  if (nprobe < 1 || nprobe > ix->nlist) return fail(ASL_ERR_INVALID, "search_entries: nprobe outside 1..nlist");
  ASL_TRY(ensure_device());
  if (!is_device_ptr(entries) || !is_device_ptr(counts) || !is_device_ptr(I) || !is_device_ptr(coarse_D) ||
      !is_device_ptr(coarse_I) || (count && !is_device_ptr(count)) || (D && !is_device_ptr(D)))
    return fail(ASL_ERR_INVALID, "search_entries: device pointers only");
  return index_search_device(ix, nq, nullptr, k, nprobe, D, I, nullptr, coarse_D, coarse_I, false,
                             reinterpret_cast<const int *>(count), reinterpret_cast<const uint2 *>(entries), counts);
}

int asl_topk_merge(int32_t S, int32_t nq, int32_t k, const float *Ds, const int64_t *Is,
                   float *D, int64_t *I) {
  clear_error();
  if (S <= 0 || nq < 0 || k <= 0 || !Ds || !Is || !D || !I) return fail(ASL_ERR_INVALID, "topk_merge: bad arguments");
  if (nq == 0) return ASL_OK;
  ASL_TRY(ensure_device());
  In<float> dDs;
  In<int64_t> dIs;
  Out<float> dD;
  Out<int64_t> dI;
  ASL_TRY(dDs.init(Ds, (size_t)S * nq * k));
  ASL_TRY(dIs.init(Is, (size_t)S * nq * k));
  ASL_TRY(dD.init(D, (size_t)nq * k));
  ASL_TRY(dI.init(I, (size_t)nq * k));
  ASL_TRY(topk_merge(dDs.d, dIs.d, S, nq, k, dD.d, dI.d));
  ASL_TRY(dD.finish());
  ASL_TRY(dI.finish());
  if (dD.to_host() || dI.to_host() || dDs.own.p || dIs.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

int asl_topk_merge_keys(int32_t S, int32_t nq, int32_t k, const int64_t *Ks, float *D, int64_t *I,
                        int32_t unordered) {
  clear_error();
  if (S <= 0 || nq < 0 || k <= 0 || !Ks || !I) return fail(ASL_ERR_INVALID, "topk_merge_keys: bad arguments");
  if (nq == 0) return ASL_OK;
  ASL_TRY(ensure_device());
  In<int64_t> dKs;
  Out<float> dD;
  Out<int64_t> dI;
  ASL_TRY(dKs.init(Ks, (size_t)S * nq * k));
  ASL_TRY(dD.init(D, (size_t)nq * k));
  ASL_TRY(dI.init(I, (size_t)nq * k));
  ASL_TRY(topk_merge_keys(dKs.d, S, nq, k, dD.d, dI.d, unordered ? 1 : 0));
  ASL_TRY(dD.finish());
  ASL_TRY(dI.finish());
  if (dD.to_host() || dI.to_host() || dKs.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

// ---------------------------------------------------------------- persistence
// '<base>_<hash7>_<charge>.idxann' stays the file name (spectral_library.py:98-108);
// the payload is this library's own little-endian format, not FAISS'.
struct IdxHeader {
  char magic[8];
  int32_t version, d, nlist, kind, pq_m, pq_bits, niter, trained;
  int64_t ntotal, n_store;
  int32_t shard_rank, shard_world, has_vids, pad;
};

int asl_index_save(const asl_index_t *ix, const char *path) {
  clear_error();
  if (!ix || !path) return fail(ASL_ERR_INVALID, "save: null");
  FILE *f = fopen(path, "wb");
  if (!f) return fail(ASL_ERR_IO, "save: cannot open %s", path);
  IdxHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "ASLIDX01", 8);
  h.version = 2;      // 2: the storage field of an IVF-Flat header is authoritative (see asl_index_load)
  h.d = ix->d;
  h.nlist = ix->nlist;
  h.kind = ix->kind;
  h.pq_m = ix->pq_m;
  h.pq_bits = ix->pq_bits;
  h.niter = ix->niter;
  h.trained = ix->trained;
  h.ntotal = ix->ntotal;
  h.n_store = ix->n_store;
  h.shard_rank = ix->shard_rank;
  h.shard_world = ix->shard_world;
  h.has_vids = ix->has_vids;
  h.pad = ix->refine_rows ? (1 | (ix->refine_k << 1)) : 0;   // IVF-PQ: exact rows follow the payload
  if (ix->kind == ASL_INDEX_IVFFLAT) h.pad = ix->flat_storage;   // IVF-Flat: component storage mode
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  auto dump = [&](const void *dev, size_t bytes) {
    if (!ok || bytes == 0) return;
    std::vector<char> tmp(bytes);
    if (hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      ok = false;
      return;
    }
    ok = fwrite(tmp.data(), 1, bytes, f) == bytes;
  };
  (void)hipStreamSynchronize(stream());
  if (ix->trained && ix->kind != ASL_INDEX_FLAT) dump(ix->centroids.p, (size_t)ix->nlist * ix->d * 4);
  if (ix->trained && ix->kind == ASL_INDEX_IVFPQ) dump(ix->codebooks.p, (size_t)ix->pq_m * ix->ksub * ix->dsub * 4);
  const size_t n = (size_t)ix->n_store;
  if (ix->kind != ASL_INDEX_FLAT) dump(ix->vlist.p, n * 4);
  if (ix->has_vids) dump(ix->vids.p, n * 4);
  if (ix->kind == ASL_INDEX_IVFPQ)
    dump(ix->codes_add.p, n * ix->pq_m);
  else
    dump(ix->vecs.p, n * ix->d * 4);
  if (ix->refine_rows) {
    const size_t rn = (size_t)ix->r_n, S = (size_t)refine_stride();
    dump(ix->r_cnt.p, rn);
    dump(ix->r_dim.p, rn * S * 2);
    dump(ix->r_val.p, rn * S * 4);
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(ASL_ERR_IO, "save: write to %s failed", path);
  return ASL_OK;
}

asl_index_t *asl_index_load(const char *path) {
  clear_error();
  if (!path) {
    fail(ASL_ERR_INVALID, "load: null path");
    return nullptr;
  }
  if (ensure_device() != ASL_OK) return nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) {
    fail(ASL_ERR_IO, "load: cannot open %s", path);
    return nullptr;
  }
  IdxHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "ASLIDX01", 8) != 0) {
    fclose(f);
    fail(ASL_ERR_IO, "load: %s is not an annsolo_mi index", path);
    return nullptr;
  }
  {  // never trust a file: every count below sizes a host vector or a device allocation
    const char *bad = nullptr;
    const bool ivf = h.kind == ASL_INDEX_IVFFLAT || h.kind == ASL_INDEX_IVFPQ;
    if (h.version != 1 && h.version != 2) bad = "unsupported version";
    else if (h.kind < ASL_INDEX_FLAT || h.kind > ASL_INDEX_IVFPQ) bad = "unknown index kind";
    else if (h.d <= 0 || h.d > (1 << 20)) bad = "bad dimension";
    else if (ivf && (h.nlist <= 0 || h.nlist > (1 << 24))) bad = "bad nlist";
    else if (h.kind == ASL_INDEX_IVFPQ &&
             (!(h.pq_m == 4 || h.pq_m == 8 || h.pq_m == 16 || h.pq_m == 32 || h.pq_m == 64) ||
              h.d % h.pq_m != 0 || h.pq_bits < 1 || h.pq_bits > 8)) bad = "bad product quantiser";
    else if (h.n_store < 0 || h.ntotal < h.n_store || h.ntotal >= ((int64_t)1 << 31)) bad = "bad vector counts";
    else if (h.niter < 0 || (h.trained != 0 && h.trained != 1) || (h.has_vids != 0 && h.has_vids != 1)) bad = "bad flags";
    else if (h.shard_world < 1 || h.shard_rank < 0 || h.shard_rank >= h.shard_world) bad = "bad shard fields";
    else if (!h.trained && h.n_store > 0 && ivf) bad = "vectors in an untrained index";
    else if (h.kind == ASL_INDEX_IVFFLAT ? (h.pad != ASL_FLAT_FX22 && h.pad != ASL_FLAT_F32)
                                         : (h.pad < 0 || ((h.pad & 1) && h.kind != ASL_INDEX_IVFPQ) || (h.pad >> 1) > TK_MAX_K))
      bad = "bad refine / storage fields";
    if (!bad) {  // the payload must be exactly what the header announces
      const uint64_t ksub = h.kind == ASL_INDEX_IVFPQ ? (1ull << h.pq_bits) : 0;
      uint64_t want = sizeof h;
      if (h.trained && ivf) want += (uint64_t)h.nlist * h.d * 4;
      if (h.trained && h.kind == ASL_INDEX_IVFPQ) want += (uint64_t)h.pq_m * ksub * (uint64_t)(h.d / h.pq_m) * 4;
      if (ivf) want += (uint64_t)h.n_store * 4;
      if (h.has_vids) want += (uint64_t)h.n_store * 4;
      want += h.kind == ASL_INDEX_IVFPQ ? (uint64_t)h.n_store * h.pq_m : (uint64_t)h.n_store * h.d * 4;
      if (h.kind == ASL_INDEX_IVFPQ && (h.pad & 1)) want += (uint64_t)h.ntotal * (1 + (uint64_t)refine_stride() * 6);
      const long here = ftell(f);
      if (fseek(f, 0, SEEK_END) != 0 || (uint64_t)ftell(f) != want) bad = "file size does not match the header";
      fseek(f, here, SEEK_SET);
    }
    if (bad) {
      fclose(f);
      fail(ASL_ERR_IO, "load: %s: %s", path, bad);
      return nullptr;
    }
  }
  asl_index *ix = new asl_index();
  ix->d = h.d;
  ix->nlist = h.nlist;
  ix->kind = h.kind;
  ix->pq_m = h.pq_m;
  ix->pq_bits = h.pq_bits;
  ix->niter = h.niter;
  ix->trained = h.trained;
  ix->ntotal = h.ntotal;
  ix->n_store = h.n_store;
  ix->shard_rank = h.shard_rank;
  ix->shard_world = h.shard_world;
  ix->has_vids = h.has_vids;
  if (ix->kind == ASL_INDEX_IVFPQ) {
    ix->ksub = 1 << ix->pq_bits;
    ix->dsub = ix->d / ix->pq_m;
  }
  bool ok = true;
  auto slurp = [&](auto &buf, size_t count) {
    using T = typename std::remove_reference<decltype(*buf.p)>::type;
    if (!ok || count == 0) return;
    std::vector<T> tmp(count);
    if (fread(tmp.data(), sizeof(T), count, f) != count) {
      ok = false;
      return;
    }
    if (buf.reserve(count) != ASL_OK ||
        hipMemcpy(buf.p, tmp.data(), count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      ok = false;
  };
  if (ix->trained && ix->kind != ASL_INDEX_FLAT) slurp(ix->centroids, (size_t)ix->nlist * ix->d);
  if (ix->trained && ix->kind == ASL_INDEX_IVFPQ) slurp(ix->codebooks, (size_t)ix->pq_m * ix->ksub * ix->dsub);
  const size_t n = (size_t)ix->n_store;
  if (ix->kind != ASL_INDEX_FLAT) slurp(ix->vlist, n);
  if (ix->has_vids) slurp(ix->vids, n);
  if (ix->kind == ASL_INDEX_IVFPQ)
    slurp(ix->codes_add, n * ix->pq_m);
  else
    slurp(ix->vecs, n * ix->d);
  if (ix->kind == ASL_INDEX_IVFFLAT) {
    ix->flat_storage = h.pad;
    // version 1: the field was 0 both for fixed-point files and for the unrounded float32 files
    // of the builds before the storage modes existed -- the components say which
    if (ok && h.version == 1 && h.pad == ASL_FLAT_FX22 && n > 0) {
      DevBuf<int32_t> nnz, nnz_max;
      int32_t h_nm[2] = {0, 0};
      ok = nnz.reserve(n) == ASL_OK && nnz_max.reserve(2) == ASL_OK &&
           count_nnz(ix->vecs.p, ix->d, (int64_t)n, nnz.p, nnz_max.p) == ASL_OK &&
           nnz_max.download(h_nm, 2) == ASL_OK && sync_stream() == ASL_OK;
      if (ok && h_nm[1] != 0) ix->flat_storage = ASL_FLAT_F32;
    }
  }
  if (ix->kind == ASL_INDEX_IVFPQ && (h.pad & 1)) {
    const size_t rn = (size_t)ix->ntotal, S = (size_t)refine_stride();
    ix->refine_rows = true;
    ix->refine_k = h.pad >> 1;
    ix->r_n = ix->ntotal;
    if (rn) {
      slurp(ix->r_cnt, rn);
      slurp(ix->r_dim, rn * S);
      slurp(ix->r_val, rn * S);
    }
  }
  fclose(f);
  if (ok && ix->n_store > 0 && (ix->kind != ASL_INDEX_FLAT || ix->has_vids)) {
    // list assignments / global ids index host and device arrays later: range-check them now
    std::vector<int32_t> tmp((size_t)ix->n_store);
    if (ix->kind != ASL_INDEX_FLAT) {
      ok = hipMemcpy(tmp.data(), ix->vlist.p, tmp.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
      for (size_t i = 0; ok && i < tmp.size(); i++) ok = tmp[i] >= 0 && tmp[i] < ix->nlist;
    }
    if (ok && ix->has_vids) {
      ok = hipMemcpy(tmp.data(), ix->vids.p, tmp.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
      for (size_t i = 0; ok && i < tmp.size(); i++) ok = tmp[i] >= 0 && (int64_t)tmp[i] < ix->ntotal;
    }
    if (!ok) {
      delete ix;
      fail(ASL_ERR_IO, "load: %s holds list or id entries out of range", path);
      return nullptr;
    }
  }
  if (!ok) {
    delete ix;
    fail(ASL_ERR_IO, "load: %s is truncated or unreadable", path);
    return nullptr;
  }
  ix->lists_dirty = true;
  return ix;
}

}  // extern "C"
