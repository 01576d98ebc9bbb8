/*
 * asl_oracle.h -- CPU ORACLE for the ANN-SoLo open-modification search hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so. The product path (ann_solo_amd/ + libannsolo_mi.so) never links,
 * imports or calls anything in this directory and fails loudly without the HIP
 * library.
 *
 * It is a plain-C restatement of the reference algorithm, function by function:
 *   encoder      /root/reference/src/ann_solo/spectrum.py:122-214
 *   rescoring    /root/reference/src/ann_solo/SpectrumMatch.cpp:8-133
 *                /root/reference/src/ann_solo/spectrum_match.pyx:28-108
 *   candidates   /root/reference/src/ann_solo/spectral_library.py:372-455
 *   ANN index    FAISS (un-vendored, un-pinned: src/setup.py:99) call sites
 *                spectral_library.py:167-181,443-445 -- restated from the
 *                published IVF-Flat / IVF-PQ definitions.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   rescoring : pinned against the reference's own compiled SpectrumMatch.cpp
 *               (oracle/_ref, built from the sources where they lie) and the
 *               reference test constant 0.44582117
 *               (src/tests/spectrum_similarity_test.py:339-341,443).
 *   encoder   : pinned against golden vectors produced by shim-importing the
 *               reference spectrum.py (tests/golden/make_golden.py).
 *   ANN       : PARITY UNPINNED -- FAISS is absent and the reference has no
 *               test at this boundary; the restatement is checked against
 *               exact brute force (recall) only.
 */
#ifndef ASL_ORACLE_H
#define ASL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Packed spectra, SoA. Peaks of spectrum i are [offsets[i], offsets[i+1]),
 * ascending m/z (the MsmsSpectrum invariant the reference relies on). */
typedef struct {
  int32_t n;
  const int32_t *offsets;         /* [n+1] */
  const float *mz;                /* [offsets[n]] */
  const float *intensity;         /* [offsets[n]] */
  const uint8_t *charge;          /* [offsets[n]] fragment-charge annotation, 0 = none */
  const double *precursor_mz;     /* [n] */
  const int32_t *precursor_charge;/* [n] */
} orc_peaks_t;

/* ---- encoder: spectrum.py:122-214 ------------------------------------- */
uint32_t orc_murmur3_32(const uint8_t *key, int len, uint32_t seed);
int32_t orc_hash_idx(int64_t bin_idx, int32_t hash_len, uint32_t seed);
void orc_get_dim(double min_mz, double max_mz, double bin_size, int64_t *n_bins,
                 double *start_dim, double *end_dim);
double orc_npy_floor_divide(double a, double b);
int64_t orc_bin_idx(float mz, double min_bound, double bin_size);
void orc_encode(const float *mz, const float *intensity, int32_t n_peaks,
                double min_bound, double bin_size, int32_t hash_len, uint32_t seed,
                int norm, float *out /* [hash_len], overwritten */);
void orc_encode_batch(const float *mz, const float *intensity, const int32_t *offsets,
                      int32_t n, double min_bound, double bin_size, int32_t hash_len,
                      uint32_t seed, int norm, float *out /* [n,hash_len] */);

/* ---- process_spectrum: spectrum.py:57-119 (+ spectrum_utils 0.3.x semantics; PARITY
 * UNPINNED: spectrum_utils is un-vendored and the reference has no test for it).
 * Input peaks ascending in m/z. Outputs up to max_peaks kept peaks (ascending m/z) with
 * their source index; returns 1 if the spectrum is valid, 0 otherwise (outputs then
 * undefined, *n_out = 0). scaling: 0 none, 1 rank, 2 root (sqrt). */
typedef struct {
  double min_mz, max_mz;
  int32_t remove_precursor;
  double remove_precursor_tolerance;
  double min_intensity;
  int32_t max_peaks;
  int32_t scaling;
  int32_t min_peaks;
  double min_mz_range;
  int32_t round_mz;   /* config.resolution is not None (spectrum.py:84) */
  int32_t resolution; /* decimals of MsmsSpectrum.round */
} orc_process_params_t;
float orc_round_mz(float mz, int32_t decimals);
int orc_process_spectrum(const float *mz, const float *intensity, int32_t n, double precursor_mz,
                         int32_t precursor_charge, const orc_process_params_t *p,
                         float *out_mz, float *out_int, int32_t *out_src, int32_t *n_out);

/* ---- SSM similarity features: spectrum_similarity.py:13-730 (asl_oracle_sim.c) ---- */
#define ORC_SIM_NFEAT 33
void orc_ssm_features(const float *q_mz, const float *q_int, int32_t nq, const float *l_mz,
                      const float *l_int, int32_t nl, const uint32_t *pm /* [npm,2] */,
                      int32_t npm, double min_mz, double max_mz, double bin_size, int32_t top,
                      double *out /* [ORC_SIM_NFEAT] */);

/* ---- rescoring: SpectrumMatch.cpp:8-133 -------------------------------- */
/* One (query,candidate) pair. matches_out (may be NULL) receives (q_idx,c_idx)
 * pairs in greedy order; returns the score. n_matches_out may be NULL. */
double orc_dot_pair(const float *q_mz, const float *q_int, int32_t q_n, double q_pmz,
                    const float *c_mz, const float *c_int, const uint8_t *c_chg,
                    int32_t c_n, double c_pmz, int32_t c_charge, double tol,
                    int allow_shift, uint32_t *matches_out, int32_t *n_matches_out);
/* get_best_match over an explicit candidate row list; first strict maximum wins
 * (SpectrumMatch.cpp:118). Returns index INTO cand_rows, -1 if n_cand==0. */
int32_t orc_best_match(const orc_peaks_t *queries, int32_t qi, const orc_peaks_t *library,
                       const int64_t *cand_rows, int32_t n_cand, double tol,
                       int allow_shift, double *score_out, uint32_t *matches_out,
                       int32_t *n_matches_out);

/* ---- inner product / brute force (IndexFlatIP) ------------------------- */
float orc_ip(const float *a, const float *b, int32_t d);
void orc_flat_search(const float *xb, int64_t nb, const float *xq, int32_t nq, int32_t d,
                     int32_t k, float *D, int64_t *I);

/* ---- k-means / IVF / PQ (restated FAISS definitions) ------------------- */
#define ORC_METRIC_IP 0
#define ORC_METRIC_L2 1
void orc_refine(const float *xb, const float *xq, int32_t nq, int32_t d, const int64_t *I_in,
                int32_t kp, int32_t k, float *D, int64_t *I);
void orc_rand_perm(int64_t n, uint64_t seed, int64_t *perm);
void orc_kmeans(const float *x, int64_t n, int32_t d, int32_t k, int32_t niter,
                uint64_t seed, int metric, int32_t max_points_per_centroid,
                float *centroids /* [k,d] */);
void orc_assign(const float *x, int64_t n, int32_t d, const float *centroids, int32_t k,
                int metric, int32_t *assign);
void orc_pq_train(const float *x, int64_t n, int32_t d, const float *centroids,
                  int32_t nlist, int32_t m, int32_t ksub, int32_t niter, uint64_t seed,
                  float *codebooks /* [m,ksub,d/m] */);
void orc_pq_encode(const float *x, int64_t n, int32_t d, const float *centroids,
                   const int32_t *assign, const float *codebooks, int32_t m,
                   int32_t ksub, uint8_t *codes /* [n,m] */);
void orc_coarse(const float *xq, int32_t nq, int32_t d, const float *centroids,
                int32_t nlist, int32_t nprobe, float *coarse_D /* [nq,nprobe] */,
                int32_t *coarse_I /* [nq,nprobe] */);
/* list_offsets[nlist+1]; ids[ntotal] (row ids in list order, ascending in list);
 * vecs[ntotal,d] / codes[ntotal,m] in the same list order. */
void orc_ivfflat_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                        int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                        const float *vecs, int32_t k, int32_t nprobe, float *D, int64_t *I);
/* fixed-point storage of IVF-Flat components (the rule add() applies in ASL_FLAT_FX22 mode) */
float orc_fx22(float x);
void orc_quantize_fx22(float *x, int64_t n);
/* the same search over a sparse (CSR, list order, dimensions ascending) copy of the stored vectors */
typedef struct {
  const int64_t *indptr; /* [ntotal+1] */
  const uint16_t *dims;  /* [nnz] */
  const float *vals;     /* [nnz] */
} orc_csr_t;
int64_t orc_csr_count(const float *x, int64_t n, int32_t d);
void orc_csr_fill(const float *x, int64_t n, int32_t d, int64_t *indptr, uint16_t *dims,
                  float *vals);
void orc_ivfflat_csr_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                            int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                            const int64_t *indptr, const uint16_t *dims, const float *vals,
                            int32_t k, int32_t nprobe, float *D, int64_t *I);
void orc_pq_lut(const float *xq, int32_t d, const float *codebooks, int32_t m,
                int32_t ksub, float *lut /* [m,ksub] */);
float orc_adc(const float *lut, int32_t m, int32_t ksub, const uint8_t *code, float coarse);
void orc_ivfpq_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                      int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                      const uint8_t *codes, const float *codebooks, int32_t m,
                      int32_t ksub, int32_t k, int32_t nprobe, float *D, int64_t *I);
/* Merge S per-shard top-k lists [S,nq,k] into [nq,k] under (score desc, id asc). */
void orc_topk_merge(const float *Ds, const int64_t *Is, int32_t S, int32_t nq, int32_t k,
                    float *D, int64_t *I);

/* ---- precursor window: spectral_library.py:417-429 --------------------- */
#define ORC_TOL_DA 0
#define ORC_TOL_PPM 1
int orc_precursor_ok(double q_mz, float lib_mz, int32_t charge, double tol, int mode);

/* ---- whole hot path for one batch (CPU baseline leg) ------------------- */
/* encode -> IVF(-PQ|-Flat) top-k -> precursor post-filter -> best match.
 * lib_pmz_f32[library->n] is spec_info's float32 precursor m/z column.
 * kind: 0 flat (payload = vecs), 1 pq (payload = codes), 2 flat over a sparse copy (payload =
 * const orc_csr_t *). Outputs per query:
 * best_row (library row, -1 if no candidate), best_score, n_cand. pm_* may be
 * NULL; otherwise pm_pairs has capacity pm_stride pairs per query. */
void orc_search_batch(const orc_peaks_t *queries, const orc_peaks_t *library,
                      const float *lib_pmz_f32, int32_t charge, double min_bound,
                      double bin_size, int32_t d, uint32_t seed, int kind,
                      const float *centroids, int32_t nlist, const int32_t *list_offsets,
                      const int32_t *ids, const void *payload, const float *codebooks,
                      int32_t m, int32_t ksub, int32_t k, int32_t nprobe,
                      double prec_tol, int prec_mode, double frag_tol, int allow_shift,
                      int32_t *best_row, double *best_score, int32_t *n_cand,
                      int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride,
                      int64_t *knn_I /* [nq,k] or NULL */, int32_t nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
