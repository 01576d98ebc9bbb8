/*
 * annsolo_mi.h -- C ABI of libannsolo_mi.so, the MI355X (gfx950) implementation of
 * ANN-SoLo's open-modification search hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference). INTEGRATION.md shows the ctypes stubs a maintainer of the
 * reference would add at those seams.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary.
 *   - return value: 0 on success, a negative ASL_ERR_* code on failure;
 *     asl_last_error() returns a thread-local message for the last failure.
 *   - every array argument may be a HOST pointer or a DEVICE (HIP) pointer; the
 *     library detects which (hipPointerGetAttributes) and stages host buffers.
 *     Buffers stay owned by the caller. Handles are owned by the library.
 *   - all work is issued on the stream set by asl_set_stream() (default: the
 *     null stream). Calls that return results into host memory synchronise that
 *     stream before returning; calls whose outputs are device pointers do not.
 *   - one host thread per handle at a time (the reference calls FAISS from one
 *     thread and serialises index loads, spectral_library.py:44,483).
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns ASL_ERR_NO_DEVICE.
 */
#ifndef ANNSOLO_MI_H
#define ANNSOLO_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASL_OK 0
#define ASL_ERR_INVALID (-1)    /* bad argument */
#define ASL_ERR_NO_DEVICE (-2)  /* no HIP device / runtime error */
#define ASL_ERR_STATE (-3)      /* e.g. search on an untrained index */
#define ASL_ERR_CAPACITY (-4)   /* a compiled-in capacity was exceeded (message says which) */
#define ASL_ERR_IO (-5)
#define ASL_ERR_HIP (-6)

#define ASL_INDEX_FLAT 0      /* IndexFlatIP: exact inner product            */
#define ASL_INDEX_IVFFLAT 1   /* IndexIVFFlat(IndexFlatIP, d, nlist, IP)     */
#define ASL_INDEX_IVFPQ 2     /* IVF + product quantiser, by-residual IP ADC */

#define ASL_TOL_DA 0
#define ASL_TOL_PPM 1

const char *asl_last_error(void);
const char *asl_version(void);
/* number of usable HIP devices (faiss.get_num_gpus(), spectral_library.py:73) */
int asl_get_num_gpus(void);
int asl_set_device(int device);
/* hipStream_t to issue work on (NULL = default stream). */
int asl_set_stream(void *hip_stream);
/* Waits for all work issued by the library, including batches of the pipeline below, and
 * reports any error those batches deferred. */
int asl_synchronize(void);
/* Software pipeline for asl_search_batch over internal streams (mode 0 = off, the default).
 * When on, a call with use_ann = 1 whose arrays ALL live on the device and whose
 * queries->n_peaks is set returns without waiting. Mode 1 (= 2): the encoder and the coarse
 * quantiser of a batch run on one stream, list scan + rescoring on a second, so the MFMA-bound
 * front of batch i+1 executes under the scan of batch i. Mode 3 moves filter + rescoring to a
 * third stream (measured no faster on MI355X: both compete for the vector ALUs). Contract in this mode: inputs must stay untouched and outputs are valid only after
 * asl_synchronize() (or a device-wide synchronisation); capacity errors the kernels flag are
 * reported by that call instead. Any other entry point first waits for the batches in flight.
 * Results are bit-identical to the synchronous path. */
int asl_set_pipeline(int on);
/* Where asl_search_batch applies the precursor-window post-filter of the neighbour lists
 * (/root/reference/src/ann_solo/spectral_library.py:417-429 AND :441-446 -- after the top-k in both
 * places): 1 (default) = inside the list scan's finish whenever the lists are consumed as a set (no
 * knn_I requested, layout-specific scan, k <= 1280): ids and window values arrive in one gather, only
 * the passing hits are written, the rescoring walks short rows; 0 = inside the rescoring kernel's
 * compaction (the only place until round 6; still used wherever the scan cannot take the filter).
 * Results are bit-identical. Returns the previous setting; environment ASL_SCAN_POSTFILTER=0 sets
 * the initial value. */
int asl_set_scan_postfilter(int on);

/* ------------------------------------------------------------------ encoder
 * Replaces spectrum_to_vector / get_dim / hash_idx, src/ann_solo/spectrum.py:122-214
 * (callers spectral_library.py:158-161,438-440). Batch form: spectrum i owns
 * peaks [offsets[i], offsets[i+1]). out is row-major [n, hash_len], fully
 * overwritten. min_bound is get_dim()'s start_dim (10.96 for 11/2010/0.04). */
int asl_get_dim(double min_mz, double max_mz, double bin_size, int64_t *n_bins,
                double *start_dim, double *end_dim);
int32_t asl_hash_idx(int64_t bin_idx, int32_t hash_len, uint32_t seed);
int asl_encode_batch(const float *mz, const float *intensity, const int32_t *offsets,
                     int32_t n, double min_bound, double bin_size, int32_t hash_len,
                     uint32_t seed, int norm, float *out);
/* The same vectors as ENTRY LISTS -- the non-zero components only, the form the IVF scans read
 * their queries in (asl_index_search_entries; a hashed spectrum has <= ~50 non-zeros of 800):
 * entries [n][64] pairs of 32-bit words (dimension * 128, the bits of the fp32 value), ascending
 * dimension, unused pairs zero; counts [n] = the number of non-zero components, or -1 - count
 * when a vector has more than 64 (that query needs the dense form); n_over (may be null): a
 * device int the call ADDS the number of such rows to. The values are those asl_encode_batch
 * stores, bit for bit. entries / counts / n_over: device memory; n_peaks = offsets[n] when the
 * caller knows it (>= 0: nothing is read back, the call never waits), else -1. */
int asl_encode_entries_batch(const float *mz, const float *intensity, const int32_t *offsets,
                             int32_t n, int32_t n_peaks, double min_bound, double bin_size,
                             int32_t hash_len, uint32_t seed, int norm, uint32_t *entries,
                             int32_t *counts, int32_t *n_over);

/* ------------------------------------------------------------------ ANN index
 * Replaces the FAISS objects used at spectral_library.py:73-87,167-181,191,
 * 443-445,487-497: IndexFlatIP, IndexIVFFlat(quantizer, d, nlist, IP), train, add,
 * search, nprobe, reset, write_index/read_index. Row ids are implicit 0..ntotal-1
 * in add order. search(): rows sorted by (score desc, id asc); missing results
 * are I = -1, D = -FLT_MAX. */
typedef struct asl_index asl_index_t;

asl_index_t *asl_index_create(int32_t d, int32_t nlist, int32_t kind, int32_t pq_m,
                              int32_t pq_bits);
void asl_index_free(asl_index_t *idx);
int asl_index_train(asl_index_t *idx, int64_t n, const float *x, uint64_t seed);
int asl_index_add(asl_index_t *idx, int64_t n, const float *x);
/* add() with every vector's inverted list given by the caller (lists[n], host or device) instead
 * of computed by the coarse quantiser: re-creates an index from stored inverted lists, e.g. from
 * a FAISS .idxann file (faiss_compat.read_index_faiss), keeping the file's own assignments. */
int asl_index_add_preassigned(asl_index_t *idx, int64_t n, const float *x, const int32_t *lists);
int asl_index_search(asl_index_t *idx, int32_t nq, const float *xq, int32_t k,
                     int32_t nprobe, float *D, int64_t *I);
int asl_index_reset(asl_index_t *idx);
int64_t asl_index_ntotal(const asl_index_t *idx);
int asl_index_is_trained(const asl_index_t *idx);
int asl_index_save(const asl_index_t *idx, const char *path);
asl_index_t *asl_index_load(const char *path);
/* k-means iterations (FAISS ClusteringParameters.niter, default 25) */
int asl_index_set_niter(asl_index_t *idx, int32_t niter);

/* Exact re-rank of the IVF-PQ short-list (FAISS IndexRefineFlat's role). kprime > 0, set BEFORE
 * add(): the index also keeps every added vector as a sparse fp32 row (<= 64 non-zeros, 384 B),
 * and asl_index_search / asl_search_batch with k < kprime let the ADC scan return kprime
 * candidates, rescore them with the exact inner product and return the k best, (score desc, id
 * asc) -- for the vectors it reaches, IVF-Flat's scores. kprime = 0 switches the re-rank off (the
 * rows stay). asl_index_refine re-ranks a short-list obtained elsewhere (e.g. merged shard rows).
 * In the unordered modes (asl_index_set_unordered: per-shard rows of a sharded search) the scan
 * returns its ADC candidates UN-refined: a sharded driver asks every shard for kprime hits, merges
 * them by ADC score into the kprime best of the whole index -- the unsharded short-list -- and
 * re-ranks that with asl_index_refine (every rank keeps the exact rows of ALL vectors), so 1 and
 * N GPUs return the same rows. asl_index_search_sharded does exactly this. */
int asl_index_set_refine(asl_index_t *idx, int32_t kprime);
/* kprime of the index (0: re-rank off) -- what a sharded driver needs to size the per-shard rows. */
int asl_index_get_refine(const asl_index_t *idx);
int asl_index_refine(asl_index_t *idx, int32_t nq, const float *xq, int32_t kprime,
                     const int64_t *I_in /* [nq,kprime], -1 = empty */, int32_t k, float *D, int64_t *I);

/* IVF-Flat component storage. ASL_FLAT_F32 (the default since round 5) keeps every component as
 * given: the float32 vectors FAISS' CPU IndexIVFFlat stores (spectral_library.py:174-181) -- ids
 * and scores are those of an index over the unquantised vectors. ASL_FLAT_FX22 (opt-in; no FAISS
 * counterpart on the CPU, the reference's GPU clone stores float16, spectral_library.py:490-497):
 * add() rounds every
 * component in [0, 1) to the nearest multiple of 2^-22 (ties to even, at most 1 - 2^-22;
 * |dx| <= 1.2e-7) and stores anything else as given. When ALL stored non-zeros are such values --
 * unit-norm hashed spectra always are -- the inverted lists are kept as 4-byte postings
 * (22-bit numerator | 10-bit local index) in whole 128-byte lines behind a one-byte-per-dimension
 * table (csrc/flat_scan.hip) instead of 6-byte postings behind 4-byte table words. Scores are
 * the canonical ascending-dimension fp32 fmaf chain over the STORED components either way, so
 * both layouts (and the generic kernels of asl_index_set_scan_variant) return identical ids and
 * score bits for the same stored vectors. Set
 * before the first add(); saved with the index (file version 2; a version-1 file that says
 * fixed point but holds off-grid components -- written before the modes existed -- loads as float32). asl_index_flat_layout: 0 = dense rows only (no
 * postings: vectors too dense), 1 = float postings, 2 = fixed-point postings. */
#define ASL_FLAT_FX22 0
#define ASL_FLAT_F32 1
int asl_index_set_flat_storage(asl_index_t *idx, int32_t mode);
int asl_index_get_flat_storage(const asl_index_t *idx);
int asl_index_flat_layout(asl_index_t *idx);

/* Scan kernel selection (identical results either way; the switch exists for A/B measurements
 * and for the parity tests): 0 = the layout-specific kernels (IVF-PQ: the tiled
 * sub-quantiser-per-lane scan with the histogram top-k when m = 32, 8 bits, nprobe <= 256;
 * IVF-Flat: the per-dimension postings scan), 1 = the generic kernels (IVF-PQ: lane-per-vector
 * scan; IVF-Flat: dense GEMM + masked top-k). Any other value is ASL_ERR_INVALID. */
int asl_index_set_scan_variant(asl_index_t *idx, int32_t variant);

/* Introspection, used by the parity tests and by multi-GPU sharding. Sizes via
 * asl_index_info; every pointer may be NULL to skip. Lists are stored in list
 * order: ids/codes/vecs of list l are [list_offsets[l], list_offsets[l+1]). */
typedef struct {
  int32_t d, nlist, kind, pq_m, pq_ksub, pq_dsub;
  int64_t ntotal;      /* vectors added through this handle (global count)      */
  int64_t nlocal;      /* vectors stored on this shard (== ntotal if unsharded) */
  int32_t trained, shard_rank, shard_world;
} asl_index_info_t;
int asl_index_info(const asl_index_t *idx, asl_index_info_t *info);
int asl_index_get_centroids(const asl_index_t *idx, float *centroids /* [nlist,d] */);
int asl_index_get_codebooks(const asl_index_t *idx, float *cb /* [m,ksub,dsub] */);
/* Install externally trained quantisers (FAISS: quantizer.add(centroids) / pq copy). */
int asl_index_set_trained(asl_index_t *idx, const float *centroids, const float *codebooks);
int asl_index_get_lists(const asl_index_t *idx, int32_t *list_offsets /* [nlist+1] */,
                        int32_t *ids /* [nlocal] */, uint8_t *codes /* [nlocal,m] PQ */,
                        float *vecs /* [nlocal,d] FLAT */);
/* Keep only the inverted lists owned by `rank` of `world` (greedy heaviest-first balancing
 * of the expected scan load, weight = list size squared; identical on every rank, see
 * asl_lpt_owner). Must be called after add(). search() then
 * returns this shard's partial top-k; combine with asl_topk_merge. */
/* Unordered result rows: with mode 1 subsequent IVF-PQ searches return the exact top-k of
 * every query as a SET -- same ids and scores, unspecified order inside the row, padding last
 * -- and skip the final sort. For consumers that re-order anyway (asl_topk_merge of shard
 * results). Mode 2 additionally packs every hit into one 64-bit key in the int64 output
 * (order-preserving score bits << 32 | ~id, 0 = empty; D is not written): 8 instead of 12
 * bytes per hit on the wire of a sharded search; merge with asl_topk_merge_keys. 0 = sorted. */
int asl_index_set_unordered(asl_index_t *idx, int32_t mode);
/* Merge of S packed-key lists Ks[S, nq, k] (mode 2 above) -> D[nq,k] (may be NULL), I[nq,k];
 * k <= 1280. unordered != 0: the rows hold the exact top-k as a set (no final sort), for
 * consumers such as asl_rescore_knn that do not depend on the order. */
int asl_topk_merge_keys(int32_t S, int32_t nq, int32_t k, const int64_t *Ks, float *D, int64_t *I,
                        int32_t unordered);
int asl_index_shard(asl_index_t *idx, int32_t rank, int32_t world);
/* The sharded search itself, for hosts that bind the C ABI directly: every rank
 * of `rccl_comm` (an ncclComm_t the caller created; one rank per GPU) calls this with ITS nq
 * queries -- the same nq everywhere, device pointers only -- after asl_index_shard(idx, rank,
 * world) on an index that was filled identically on every rank. All-gather of the queries (as
 * entry lists, 516 bytes per query, whenever the packed-key scans run; a batch holding a query with
 * more than 64 non-zero components is repeated with dense rows) and of
 * the probe lists, scan of the local inverted lists for all world x nq queries, grouped
 * send/recv of the per-shard top-k (all-to-all), merge: D / I [nq, k] equal the unsharded
 * index's rows for these queries. Enqueued on the library's stream; RCCL is resolved from the
 * process at first use (never linked). */
int asl_index_search_sharded(asl_index_t *idx, void *rccl_comm, int32_t nq, const float *xq,
                             int32_t k, int32_t nprobe, float *D, int64_t *I);
/* The same with the exchange's three sizes given: keys a head carries (<= 0: ceil(2k / world)), the
 * shards' own k (<= 0: asl_shard_k(k, world)), answer slots per query and destination (< 0:
 * max(8, k / 16)). The result does not depend on them (a buffer that runs full sends the batch
 * down the full-row exchange); tests set them to run every step of the exchange on a communicator
 * of any size -- at world 1 the default head is the whole row. */
int asl_index_search_sharded_ex(asl_index_t *idx, void *rccl_comm, int32_t nq, const float *xq,
                                int32_t k, int32_t nprobe, float *D, int64_t *I, int32_t head_keys,
                                int32_t shard_keys, int64_t extras_per_query);
/* 1 if asl_index_search_preassigned can emit packed keys (unordered mode 2) for this index at
 * (k, nprobe) -- IVF-PQ: the tiled scan (m = 32, 8-bit codes, automatic scan variant, nprobe
 * within the tiled kernel's limit, k + 768 <= 2048); IVF-Flat: the postings scan (sparse stored
 * vectors, k <= 1280) --, else 0 (exchange (D, I) rows then). Builds the scan layout if it is
 * out of date, so the answer is the one a search would meet. For IVF-Flat it depends on the
 * vectors THIS shard holds (an empty or dense shard scans dense rows): a sharded driver must
 * agree on it across ranks (MIN) before choosing the exchange format, as distributed.py and
 * asl_index_search_sharded do. */
int asl_index_supports_keys(asl_index_t *idx, int32_t k, int32_t nprobe);
/* The exact top-k exchange of a sharded search (csrc/exchange.hip; device pointers only; what
 * ann_solo_amd/distributed.py and asl_index_search_sharded run between their collectives; no
 * reference counterpart: spectral_library.py:494 uses device 0 only). Rows of packed keys as
 * asl_index_set_unordered mode 2 emits them (0 = empty, any order).
 *   asl_keys_split: K [nrows, k_s] -> head [nrows, kp] (slots 0 .. kp-2: the row's best keys, all
 *     those at or above a score-bucket floor that admits at most kp - 1; slot kp-1: T, the best
 *     key held back, 0 if none), floor [nrows] (that bucket floor: the keys held back are the
 *     keys of K below it, which stay where they are) and -- rowmin != NULL, for shards that scan
 *     with k_s < k -- rowmin [nrows]: the row's smallest key if the row is FULL (the scan may have
 *     dropped keys, all below it), else 0.
 *   asl_keys_merge_heads: heads [S, nq, kp] of the S shards -> out_keys [nq, k] (the best k keys
 *     seen, a set), bounds [S, nq] (B = the k-th best key seen if shard s must send what it holds
 *     above B -- its T beats B --, else ~0: send nothing), need [nq] (some shard was asked).
 *   asl_keys_rescan_list: on the shard (k_s < k), after the bounds arrived: the rows whose bound
 *     lies below rowmin -- a dropped key may be above the bound -- -> rowlist [R] (int64, ZEROED BY
 *     THE CALLER; the first *count slots), rmap [nrows] (slot or -1), *count += their number
 *     (device memory: the gate of asl_index_search_gated, which scans exactly those rows again
 *     with the full k); more than R rows: *overflow = 1.
 *   asl_keys_extras: on the shard, rows destination-major (row = dst * nq + q): K [W * nq, k_s]
 *     and floor [W * nq] as asl_keys_split saw / wrote them, bounds [W * nq] -> xbuf
 *     [W, nq + xcap]: per destination nq header words (count << 32 | start) then the payload =
 *     the row's keys outside the head above the bound; rmap != NULL: a row with rmap[row] >= 0
 *     answers from K3 [rmap[row], k3], its second scan. cursor [W] int32, ZEROED BY THE CALLER
 *     (the payload cursors; the call returns without waiting); *overflow = 1 when a
 *     destination's xcap slots do not suffice (the caller must then repeat the batch with the
 *     full exchange of k-deep rows). *overflow is never cleared here.
 *   asl_keys_merge_final: heads + the xbuf [S, nq + xcap] received (NULL: none) + out_keys/need of
 *     asl_keys_merge_heads -> I [nq, k] ids (a set, -1 padded) and D (may be NULL): the exact
 *     top-k of the union of the shards' FULL rows. k <= 1280. */
int asl_keys_split(int64_t nrows, int32_t k, int32_t kp, const int64_t *K, int64_t *head, int32_t *floor,
                   int64_t *rowmin);
int asl_keys_merge_heads(int32_t S, int32_t nq, int32_t kp, int32_t k, const int64_t *heads,
                         int64_t *out_keys, int64_t *bounds, int32_t *need);
int asl_keys_rescan_list(int64_t nrows, const int64_t *bounds, const int64_t *rowmin, int32_t R, int64_t *rowlist,
                         int32_t *rmap, int32_t *count, int32_t *overflow);
int asl_keys_extras(int32_t W, int32_t nq, int32_t k, const int64_t *K, const int32_t *floor,
                    const int64_t *bounds, int64_t xcap, int64_t *xbuf, int32_t *cursor, int32_t *overflow,
                    const int32_t *rmap, const int64_t *K3, int32_t k3);
int asl_keys_merge_final(int32_t S, int32_t nq, int32_t kp, int32_t k, const int64_t *heads,
                         const int64_t *xbuf, int64_t xcap, const int64_t *prev_keys, const int32_t *need,
                         float *D, int64_t *I);
/* The shards' own k for a final k at `world` ranks: k / 2 from 8 ranks on, 5 k / 8 from 4
 * (rounded up to 64); k itself below 4 ranks or when that is not more than the head's
 * ceil(2 k / world) key slots (pure host integer code). */
int32_t asl_shard_k(int32_t k, int32_t world);
/* asl_index_search_preassigned for a list whose length only the device knows (device pointers
 * only, never waits): a launch for `cap` rows of which the first *count are searched; the other
 * rows of D / I stay untouched. Layout-specific scans only (what asl_index_supports_keys says). */
int asl_index_search_gated(asl_index_t *idx, int32_t cap, const float *xq, int32_t k, int32_t nprobe,
                           const float *coarse_D, const int32_t *coarse_I, float *D, int64_t *I,
                           const int32_t *count);
/* asl_index_search_preassigned with the queries as entry lists (asl_encode_entries_batch) instead
 * of dense rows: 512 bytes per query instead of 3.2 KB and no listing pass -- a sharded search
 * encodes the other ranks' queries straight into this form. Results equal the dense call's, bit
 * for bit. A row with a negative count is searched as an all-zero query (watch the encoder's
 * n_over). count (may be null): as in asl_index_search_gated. Device pointers only, never waits;
 * layout-specific scans only (what asl_index_supports_keys says), not with the exact re-rank. */
int asl_index_search_entries(asl_index_t *idx, int32_t nq, const uint32_t *entries, const int32_t *counts,
                             int32_t k, int32_t nprobe, const float *coarse_D, const int32_t *coarse_I,
                             float *D, int64_t *I, const int32_t *count);
/* list -> owner rank map of the balancing above, for inspection. */
int asl_index_shard_map(const asl_index_t *idx, int32_t world, int32_t *owner /* [nlist] */);

/* The list -> rank balancing rule by itself (pure host integer code, no device):
 * lists sorted by size descending (ties: lower list id first), each given to the
 * currently least-loaded rank (ties: lower rank). */
int asl_lpt_owner(int32_t nlist, const int64_t *sizes, int32_t world, int32_t *owner);

/* IndexIVF::search_preassigned: like asl_index_search but with the coarse quantiser's
 * answer supplied by the caller (coarse_D / coarse_I [nq, nprobe], as asl_index_coarse
 * emits; -1 entries are skipped). Used by the sharded search: every rank quantises only
 * its own slice of the batch, the probe lists are all-gathered, and each rank scans its
 * own inverted lists for the whole batch. IVF-PQ and IVF-Flat (which ignores coarse_D: its
 * scores do not contain the coarse term). */
int asl_index_search_preassigned(asl_index_t *idx, int32_t nq, const float *xq, int32_t k,
                                 int32_t nprobe, const float *coarse_D,
                                 const int32_t *coarse_I, float *D, int64_t *I);

/* Merge S partial results [S, nq, k] into [nq, k] under (score desc, id asc). */
int asl_topk_merge(int32_t S, int32_t nq, int32_t k, const float *Ds, const int64_t *Is,
                   float *D, int64_t *I);

/* Algorithmic work of the IVF-Flat postings scan for these queries at this nprobe (measurement;
 * no reference counterpart): *bytes = sum over (query, probed block, non-zero query dimension)
 * of the 4-byte table word + 6 bytes per posting (fixed-point layout: 1-byte table entry + 4
 * bytes per posting); *lines = the 128-byte lines the scan touches for them (segments are placed
 * by line; fixed-point layout: the block's whole table row + the lines of the wanted segments).
 * What bench.py prices the scan kernel's roofline with. */
int asl_index_postings_work(asl_index_t *idx, int32_t nq, const float *xq, int32_t nprobe,
                            int64_t *bytes, int64_t *lines);

/* Exposed stages of the IVF search (parity tests; each mirrors one oracle function). */
int asl_index_coarse(asl_index_t *idx, int32_t nq, const float *xq, int32_t nprobe,
                     float *coarse_D /* [nq,nprobe] */, int32_t *coarse_I /* [nq,nprobe] */);
int asl_index_pq_lut(asl_index_t *idx, int32_t nq, const float *xq,
                     float *lut /* [nq,m,ksub] */);

/* ------------------------------------------------------------------ rescoring
 * Replaces get_best_match (src/ann_solo/spectrum_match.pyx:28-108) and
 * SpectrumMatcher::dot (src/ann_solo/SpectrumMatch.cpp:8-133), batched over
 * queries. Spectra are packed SoA; peaks ascending in m/z; `charge` is the
 * per-peak fragment-charge annotation (0 = none, pyx:74-79; ignored for queries). */
typedef struct {
  int32_t n;
  const int32_t *offsets;          /* [n+1] */
  const float *mz;                 /* [offsets[n]] */
  const float *intensity;          /* [offsets[n]] */
  const uint8_t *charge;           /* [offsets[n]] or NULL (all 0) */
  const double *precursor_mz;      /* [n] */
  const int32_t *precursor_charge; /* [n] */
  int64_t n_peaks;                 /* offsets[n] when the caller knows it, else 0: device-resident
                                      offsets are then read back (one stream synchronisation) */
} asl_peaks_t;

/* Candidates of query q are library rows cand_rows[cand_offsets[q] .. cand_offsets[q+1]).
 * Entries < 0 are skipped. Outputs (each may be NULL):
 *   best_cand[q]  position inside the query's candidate list (first strict maximum
 *                 wins, SpectrumMatch.cpp:118), -1 if the list has no valid entry
 *   best_score[q] SpectrumMatcher::dot score (double)
 *   pm_count[q]   number of matched peak pairs of the winner
 *   pm_pairs      [nq, pm_stride, 2] (query_peak, candidate_peak) in greedy order, zero beyond
 *                 pm_count[q] (asl_search_batch / asl_rescore_knn write every slot) */
int asl_rescore_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                      const int64_t *cand_rows, const int32_t *cand_offsets,
                      double fragment_mz_tolerance, int allow_shift, int32_t *best_cand,
                      double *best_score, int32_t *pm_count, uint32_t *pm_pairs,
                      int32_t pm_stride);

/* ------------------------------------------------------------------ peak preprocessing
 * Replaces process_spectrum (src/ann_solo/spectrum.py:57-119: spectrum_utils set_mz_range,
 * round(resolution,'sum'), remove_precursor_peak(tol,'Da',2), filter_intensity,
 * scale_intensity, L2 norm, and the validity checks of :13-36), batched. Raw peaks ascending
 * in m/z, <= 4096 per spectrum. Outputs are padded to max_peaks per spectrum: out_mz /
 * out_intensity / out_src [n, max_peaks] (out_src = index of the kept peak inside its raw
 * spectrum -- after rounding: of the most intense of the merged peaks, whose annotation
 * survives --, may be NULL), out_count[n], out_valid[n] (is_valid).
 * scaling: 0 none, 1 rank, 2 root. */
typedef struct {
  double min_mz, max_mz;              /* config.min_mz / max_mz (inclusive) */
  int32_t remove_precursor;           /* config.remove_precursor */
  double remove_precursor_tolerance;  /* config.remove_precursor_tolerance (Da) */
  double min_intensity;               /* config.min_intensity (relative to the base peak) */
  int32_t max_peaks;                  /* config.max_peaks_used(_library), <= 256 */
  int32_t scaling;
  int32_t min_peaks;                  /* config.min_peaks */
  double min_mz_range;                /* config.min_mz_range */
  int32_t round_mz;                   /* config.resolution is not None (spectrum.py:84) */
  int32_t resolution;                 /* config.resolution: decimals of MsmsSpectrum.round */
} asl_process_params_t;
int asl_process_batch(const asl_peaks_t *raw, const asl_process_params_t *params,
                      float *out_mz, float *out_intensity, int32_t *out_src,
                      int32_t *out_count, uint8_t *out_valid);

/* ------------------------------------------------------------------ SSM features
 * Replaces the per-SSM SpectrumSimilarityCalculator calls of _compute_ssm_features
 *   /root/reference/src/ann_solo/utils.py:344-456
 *   /root/reference/src/ann_solo/spectrum_similarity.py:13-730
 * For every query i: the similarity features of (query i, library row lib_rows[i]) over
 * the peak matches pm_pairs[i, :pm_count[i]] that asl_search_batch / asl_rescore_* emit.
 * features[i, ASL_SSM_NFEAT], in the order of the reference's feature dictionary
 * (utils.py:309-342; `*_top` = restricted to the `top` = 5 most intense library peaks):
 *   0 cosine, 1 cosine_top, 2 n_matched_peaks, 3 frac_n_peaks_query, 4 frac_n_peaks_lib,
 *   5 frac_n_peaks_lib_top, 6 frac_int_query, 7 frac_int_lib, 8 frac_int_lib_top,
 *   9 mse_mz, 10 mse_mz_top, 11 mse_int, 12 mse_int_top, 13 contrast_angle,
 *   14 contrast_angle_top, 15 hypergeometric_score(min_mz, max_mz, bin_size), 16 kendalltau,
 *   17 ms_for_id_v1, 18 ms_for_id_v2, 19 entropy_unweighted, 20 entropy_weighted,
 *   21 scribe_fragment_acc, 22 scribe_fragment_acc_top, 23 manhattan, 24 euclidean,
 *   25 chebyshev, 26 pearsonr, 27 pearsonr_top, 28 spearmanr, 29 spearmanr_top,
 *   30 braycurtis, 31 canberra, 32 ruzicka.
 * Rows with lib_rows[i] < 0 are NaN (the reference skips SSMs without a match). */
#define ASL_SSM_NFEAT 33
int asl_ssm_features_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                           const int32_t *lib_rows /* [nq] */,
                           const uint32_t *pm_pairs /* [nq, pm_stride, 2] */,
                           const int32_t *pm_count /* [nq] */, int32_t pm_stride,
                           double min_mz, double max_mz, double bin_size, int32_t top,
                           double *features /* [nq, ASL_SSM_NFEAT] */);

/* features[:, 0] alone -- the cosine over the peak matches (spectrum_similarity.py:81-106), the
 * cascade's default search-engine score (utils.py:407) -- with the same bits as column 0 of
 * asl_ssm_features_batch; rows with lib_rows[i] < 0 are NaN. */
int asl_ssm_cosine_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                         const int32_t *lib_rows /* [nq] */,
                         const uint32_t *pm_pairs /* [nq, pm_stride, 2] */,
                         const int32_t *pm_count /* [nq] */, int32_t pm_stride,
                         double *cosine /* [nq] */);

/* ------------------------------------------------------------------ hot path
 * One batch of same-charge queries through
 *   SpectralLibrary._search_batch / _get_library_candidates
 *   (src/ann_solo/spectral_library.py:328-455)
 * entirely on the device: encode -> index.search(k) -> precursor-window
 * post-filter (:417-429,:441-446) -> best match per query (:356-365).
 * A library handle keeps the packed peak store resident in HBM. */
typedef struct asl_library asl_library_t;
/* lib_pmz_f32: spec_info's float32 precursor m/z column (reader.py:184-191), may be
 * NULL (then (float)precursor_mz). valid: per-spectrum is_valid flag or NULL. */
asl_library_t *asl_library_create(const asl_peaks_t *library, const float *lib_pmz_f32,
                                  const uint8_t *valid);
void asl_library_free(asl_library_t *lib);
int64_t asl_library_size(const asl_library_t *lib);

typedef struct {
  double min_bound, bin_size; /* encoder grid (get_dim) */
  uint32_t hash_seed;         /* 42 */
  int32_t k;                  /* config.num_candidates */
  int32_t nprobe;             /* config.num_probe */
  int32_t charge;             /* precursor charge of this batch */
  double precursor_tol;       /* config.precursor_tolerance_mass(_open) */
  int32_t precursor_mode;     /* ASL_TOL_DA / ASL_TOL_PPM */
  double fragment_mz_tolerance;
  int32_t allow_shift;        /* config.allow_peak_shifts */
  int32_t use_ann;            /* 1: ANN top-k AND window (open+ann); 0: window only (std / bf) */
} asl_search_params_t;

/* Outputs, each [nq] (NULL to skip): best_row = library row of the best match
 * (-1: no candidate), best_score, n_cand = candidates that reached rescoring,
 * pm_count / pm_pairs as in asl_rescore_batch; knn_I [nq,k] = raw ANN ids. */
int asl_search_batch(asl_library_t *lib, asl_index_t *idx, const asl_peaks_t *queries,
                     const asl_search_params_t *params, int32_t *best_row,
                     double *best_score, int32_t *n_cand, int32_t *pm_count,
                     uint32_t *pm_pairs, int32_t pm_stride, int64_t *knn_I);

/* Same pipeline from the post-filter on, for candidates retrieved elsewhere (the
 * multi-GPU path: per-shard top-k lists are exchanged and merged first). knn_I is
 * [nq, k] int64 library rows, -1 padded, as asl_index_search / asl_topk_merge emit. */
int asl_rescore_knn(asl_library_t *lib, const asl_peaks_t *queries,
                    const asl_search_params_t *params, const int64_t *knn_I,
                    int32_t *best_row, double *best_score, int32_t *n_cand,
                    int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride);

/* Precursor-window candidate generation alone (spectral_library.py:417-429):
 * CSR lists of library rows (ascending) whose precursor passes the window. Two-call
 * protocol: first with cand_rows == NULL to get cand_offsets[nq+1], then with a
 * buffer of cand_offsets[nq] entries. */
int asl_window_candidates(asl_library_t *lib, int32_t nq, const double *query_pmz,
                          int32_t charge, double tol, int32_t mode, int32_t *cand_offsets,
                          int64_t *cand_rows);

/* ------------------------------------------------------------------ profiling
 * HIP-event timing of the individual stages of the last asl_search_batch /
 * asl_index_search calls on the library's stream. Names: "encode","coarse_gemm",
 * "coarse_select","scan","filter","rescore","rescore_matches".
 * on = 1: every stage (two events around each: ~16 per batch, which keep the stages of a
 * pipelined batch from being dispatched back to back -- measured 0.3 ms of a 7.9 ms step);
 * on = 2: the list scan kernel only (the dominant kernel, two events per batch; the
 * scanned-vector counter, which costs a small kernel per scan, is not fed either); 0: off. */
int asl_profile_enable(int on);
int asl_profile_reset(void);
/* Accumulated milliseconds and launch count for a stage since the last reset. */
int asl_profile_get(const char *stage, double *total_ms, int64_t *launches);
/* Algorithmic work of the scan kernels since the last reset: sum over queries of
 * probed-list lengths (vectors scored). */
int64_t asl_profile_scanned_vectors(void);

#ifdef __cplusplus
}
#endif
#endif /* ANNSOLO_MI_H */
