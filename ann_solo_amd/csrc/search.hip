// search.hip -- one batch of the open-modification hot path, device resident:
// SpectralLibrary._search_batch / _get_library_candidates
// (/root/reference/src/ann_solo/spectral_library.py:328-455).
//
//   encode (spectrum.py:166-214)  ->  index.search(k) (:443-444)
//   -> precursor-window post-filter (:417-429 AND :441-446)  ->  best match (:356-365)
//
// The reference builds two dense nq x N boolean masks; here the ANN ids are
// post-filtered inside the rescoring kernel's candidate compaction (rescore.hpp:
// PrecFilter) and the window-only modes
// (cascade level 'std', --mode bf) binary-search a precursor-sorted copy of the
// library, so nothing is O(nq*N).
#include <algorithm>

#include "common.hpp"
#include "ivf_kernels.hpp"
#include "rescore.hpp"

namespace asl {
int encode_device(const float *mz, const float *inten, const int32_t *offsets, int32_t n,
                  double min_bound, double bin_size, int32_t hash_len, uint32_t seed,
                  int norm, float *out);
int index_search_device(asl_index *ix, int nq, const float *xq, int k, int nprobe, float *D,
                        int64_t *I64, int32_t *I32, const float *pre_D, const int32_t *pre_I,
                        bool set_mode, const int *gate = nullptr, const uint2 *pre_ent = nullptr,
                        const int32_t *pre_cnt = nullptr);
int index_dim(const asl_index *ix);
void index_set_post_filter(asl_index *ix, const IndexPostFilter &p);
bool index_post_filter_applied(asl_index *ix);
int index_nprobe(const asl_index *ix, int nprobe);
int index_prepare(asl_index *ix);
int index_coarse_device(asl_index *ix, int nq, const float *xq, int nprobe, float *out_D,
                        int32_t *out_I, uint2 *ent_out = nullptr, int32_t *cnt_out = nullptr,
                        bool *have_ent = nullptr);
// Window [lo,hi) of each query inside the precursor-sorted library.
__global__ void window_range_kernel(const double *__restrict__ q_pmz, int nq,
                                    const float *__restrict__ sorted_pmz, int n, int charge,
                                    double tol, int mode, int32_t *__restrict__ lo_out,
                                    int32_t *__restrict__ cnt_out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const double qm = q_pmz[q];
  int a = 0, b = n;  // p0 = first element with (double)l >= qm
  while (a < b) {
    const int mid = (a + b) >> 1;
    if ((double)sorted_pmz[mid] < qm) a = mid + 1; else b = mid;
  }
  const int p0 = a;
  a = 0; b = p0;     // left side: first index whose value passes
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (precursor_ok(qm, sorted_pmz[mid], charge, tol, mode)) b = mid; else a = mid + 1;
  }
  const int lo = a;
  a = p0; b = n;     // right side: first index whose value fails
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (precursor_ok(qm, sorted_pmz[mid], charge, tol, mode)) a = mid + 1; else b = mid;
  }
  lo_out[q] = lo;
  cnt_out[q] = a - lo;
}

__global__ void window_fill_kernel(const int32_t *__restrict__ lo, const int32_t *__restrict__ offsets,
                                   const int32_t *__restrict__ sorted_row,
                                   const uint8_t *__restrict__ valid, int32_t *__restrict__ cand) {
  const int q = blockIdx.x;
  const int b = offsets[q], n = offsets[q + 1] - b, l = lo[q];
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int32_t row = sorted_row[l + t];
    cand[b + t] = (!valid || valid[row]) ? row : -1;
  }
}

}  // namespace asl

using namespace asl;

struct asl_library {
  int64_t n = 0;
  DevBuf<int32_t> offsets, pcharge;
  DevBuf<float> mz, intensity, pmz32;
  DevBuf<uint8_t> charge, valid;
  DevBuf<double> pmz;
  // One fixed-size SLOT per row: [RowMeta 32 B][mz x n][charge x n][intensity x n]. The rescoring
  // kernels gather the row record and find the row's first ~24 m/z values in the SAME 128-byte line;
  // the address of a row's peaks is row * slot + 32, not a second, dependent look-up (round 5; until
  // then a 32-byte record array and a separately packed record per row: 4.1 lines and two hops per
  // candidate instead of 3 lines and one).
  DevBuf<uint8_t> records;   // n * slot bytes (DevPeaks::records)
  uint32_t slot = 0;         // bytes per row (a multiple of 128)
  DevBuf<float> wcol;     // window column alone, NaN for invalid spectra
  bool has_valid = false;
  DevPeaks dev;
  // precursor-sorted view (window search)
  DevBuf<float> sorted_pmz;
  DevBuf<int32_t> sorted_row;
  // scratch
  DevBuf<float> qvec;
  DevBuf<int32_t> knn, cand, lo, cnt, woff;
  // buffers that cross the two streams of the pipeline, by batch parity
  DevBuf<float> p_qvec[2], p_cD[2];
  DevBuf<int32_t> p_cI[2], p_knn[2], p_cnt[2];
  DevBuf<int32_t> p_rows[2], rows_len;   // lengths of the neighbour rows when the scan applied the precursor filter
  DevBuf<uint2> p_ent[2];          // the batch's entry lists: listed by the coarse stage, read by the scan
  bool p_have_ent[2] = {false, false};
  DevBuf<double> pair_score;
  DevBuf<long long> best_slot;
  DevBuf<int> status;
  RescoreScratch rs_scratch;       // per-query flags between the rescoring launches of THIS handle's stream
};

// one wave per spectrum: its row record and its peaks from the three arrays into its slot
__global__ void pack_records_kernel(const int32_t *__restrict__ offsets, const float *__restrict__ mz,
                                    const float *__restrict__ inten, const uint8_t *__restrict__ chg,
                                    const RowMeta *__restrict__ meta, int64_t n, uint32_t slot,
                                    uint8_t *__restrict__ rec) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const int co = offsets[r], cn = offsets[r + 1] - co;
  uint8_t *s0 = rec + (size_t)r * slot;
  if (lane < 8) reinterpret_cast<uint32_t *>(s0)[lane] = reinterpret_cast<const uint32_t *>(meta + r)[lane];
  uint8_t *b = rec + (size_t)meta[r].rec4 * 4;      // = s0 + 32
  float *f = reinterpret_cast<float *>(b);
  for (int j = lane; j < cn; j += 64) {
    f[j] = mz[co + j];
    b[4 * (size_t)cn + j] = chg[co + j];
    f[rec_int0(cn) + j] = inten[co + j];
  }
}

static int pack_peak_records(const int32_t *offsets, const float *mz, const float *inten,
                             const uint8_t *chg, const RowMeta *meta, int64_t n, uint32_t slot, uint8_t *rec) {
  if (n <= 0) return ASL_OK;
  hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream(), offsets,
                     mz, inten, chg, meta, n, slot, rec);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

// The precursor window applied inside the scan's finish (ScanPostFilter, common.hpp) whenever the
// neighbour rows are consumed as a set; ASL_SCAN_POSTFILTER=0 keeps it in the rescoring (A/B runs).
static int &scan_postfilter_flag() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("ASL_SCAN_POSTFILTER");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on;
}
static bool scan_postfilter_on() { return scan_postfilter_flag() == 1; }
static void offer_post_filter(asl_library *L, asl_index *idx, const DevPeaks &Q, const asl_search_params_t *P,
                              int32_t *row_len) {
  IndexPostFilter pf;
  pf.payload = L->wcol.p;
  pf.n = L->n;
  pf.q_pmz = Q.precursor_mz;
  pf.count = row_len;
  pf.tol = P->precursor_tol;
  pf.mode = P->precursor_mode;
  pf.charge = P->charge;
  index_set_post_filter(idx, pf);
}

// the precursor filter / row records of a library handle
static void library_filter(const asl_library *L, PrecFilter &flt) {
  flt.meta = reinterpret_cast<const RowMeta *>(L->records.p);
  flt.meta_stride = L->slot;
  flt.wcol = L->wcol.p;
}

extern "C" {

int asl_set_scan_postfilter(int on) {
  clear_error();
  int &f = scan_postfilter_flag();
  const int prev = f;
  f = on ? 1 : 0;
  return prev;
}

asl_library_t *asl_library_create(const asl_peaks_t *p, const float *lib_pmz_f32,
                                  const uint8_t *valid) {
  clear_error();
  if (!p || p->n < 0) {
    fail(ASL_ERR_INVALID, "library_create: null peaks");
    return nullptr;
  }
  if (ensure_device() != ASL_OK) return nullptr;
  PeaksStage st;
  if (st.init(p) != ASL_OK) return nullptr;
  asl_library *L = new asl_library();
  L->n = p->n;
  const size_t n = (size_t)p->n, np = (size_t)st.dev.n_peaks;
  bool ok = true;
  auto up = [&](auto &buf, const auto *src, size_t cnt) {
    if (ok && cnt && buf.upload(src, cnt) != ASL_OK) ok = false;
  };
  up(L->offsets, st.dev.offsets, n + 1);
  up(L->mz, st.dev.mz, np);
  up(L->intensity, st.dev.intensity, np);
  if (st.dev.charge) {
    up(L->charge, st.dev.charge, np);
  } else if (np) {
    ok = ok && L->charge.reserve(np) == ASL_OK &&
         hipMemsetAsync(L->charge.p, 0, np, stream()) == hipSuccess;
  }
  up(L->pmz, st.dev.precursor_mz, n);
  up(L->pcharge, st.dev.precursor_charge, n);
  // float32 precursor column + precursor-sorted view (host side: one-time, O(n log n))
  std::vector<double> h_pmz(n);
  std::vector<float> h_pmz32(n);
  if (ok && n) {
    ok = hipMemcpyAsync(h_pmz.data(), L->pmz.p, n * 8, hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         sync_stream() == ASL_OK;
    if (lib_pmz_f32) {
      ok = ok && hipMemcpy(h_pmz32.data(), lib_pmz_f32, n * 4, hipMemcpyDefault) == hipSuccess;
    } else {
      for (size_t i = 0; i < n; i++) h_pmz32[i] = (float)h_pmz[i];
    }
  }
  up(L->pmz32, h_pmz32.data(), n);
  if (valid) {
    up(L->valid, valid, n);
    L->has_valid = true;
  }
  if (ok && n) {   // packed rows: invalid spectra get a NaN window column (never a candidate)
    std::vector<int32_t> h_off(n + 1), h_chg(n);
    std::vector<uint8_t> h_valid(n, 1);
    ok = hipMemcpyAsync(h_off.data(), L->offsets.p, (n + 1) * 4, hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         hipMemcpyAsync(h_chg.data(), L->pcharge.p, n * 4, hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         sync_stream() == ASL_OK;
    if (ok && valid) ok = hipMemcpy(h_valid.data(), valid, n, hipMemcpyDefault) == hipSuccess;
    std::vector<RowMeta> hm(n);
    std::vector<float> h_wcol(n);
    int max_cn = 0;
    for (size_t i = 0; i < n; i++) max_cn = std::max(max_cn, h_off[i + 1] - h_off[i]);
    // slot = row record + the largest packed peak record, rounded up to whole 128-byte lines
    const uint64_t slot = (sizeof(RowMeta) + asl::rec_bytes((uint64_t)max_cn) + 127) & ~127ull;
    for (size_t i = 0; ok && i < n; i++) {
      hm[i].off = h_off[i];
      hm[i].cn = h_off[i + 1] - h_off[i];
      hm[i].charge = h_chg[i];
      hm[i].pmz32 = h_valid[i] ? h_pmz32[i] : __builtin_nanf("");
      hm[i].pmz64 = h_pmz[i];
      hm[i].rec4 = (uint32_t)(((uint64_t)i * slot + sizeof(RowMeta)) >> 2);
      hm[i].pad = 0u;
      h_wcol[i] = hm[i].pmz32;
    }
    if ((uint64_t)n * slot >= (1ull << 34)) {     // rec4 is 32 bits of 4-byte units
      ok = false;
      fail(ASL_ERR_CAPACITY, "library_create: more than 16 GiB of row slots (%llu rows x %llu bytes) in one partition",
           (unsigned long long)n, (unsigned long long)slot);
    }
    L->slot = (uint32_t)slot;
    up(L->wcol, h_wcol.data(), n);
    if (ok) {
      DevBuf<RowMeta> meta_tmp;
      ok = meta_tmp.upload(hm.data(), n) == ASL_OK && L->records.reserve((size_t)n * slot + 16) == ASL_OK &&
           hipMemsetAsync(L->records.p, 0, (size_t)n * slot + 16, stream()) == hipSuccess &&
           pack_peak_records(L->offsets.p, L->mz.p, L->intensity.p, L->charge.p, meta_tmp.p, (int64_t)n,
                             L->slot, L->records.p) == ASL_OK &&
           sync_stream() == ASL_OK;
    }
  }
  if (ok && n) {
    std::vector<int32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(),
                     [&](int32_t a, int32_t b) { return h_pmz32[(size_t)a] < h_pmz32[(size_t)b]; });
    std::vector<float> sp(n);
    for (size_t i = 0; i < n; i++) sp[i] = h_pmz32[(size_t)order[i]];
    up(L->sorted_pmz, sp.data(), n);
    up(L->sorted_row, order.data(), n);
  }
  if (ok) ok = sync_stream() == ASL_OK;
  if (!ok) {
    delete L;
    if (!*asl_last_error()) fail(ASL_ERR_HIP, "library_create: device upload failed");
    return nullptr;
  }
  L->dev.n = (int32_t)p->n;
  L->dev.n_peaks = (int64_t)np;
  L->dev.offsets = L->offsets.p;
  L->dev.mz = L->mz.p;
  L->dev.intensity = L->intensity.p;
  L->dev.charge = L->charge.p;
  L->dev.precursor_mz = L->pmz.p;
  L->dev.precursor_charge = L->pcharge.p;
  L->dev.records = L->records.p;
  return L;
}

void asl_library_free(asl_library_t *L) { delete L; }
int64_t asl_library_size(const asl_library_t *L) { return L ? L->n : 0; }

// CSR window candidates on the device: fills L->woff ([nq+1]) and L->cand; total -> *total.
static int window_candidates_device(asl_library *L, int nq, const double *q_pmz_dev, int charge,
                                    double tol, int mode, int64_t *total) {
  ASL_TRY(L->lo.reserve((size_t)nq));
  ASL_TRY(L->cnt.reserve((size_t)nq));
  ASL_TRY(L->woff.reserve((size_t)nq + 1));
  hipLaunchKernelGGL(window_range_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, stream(),
                     q_pmz_dev, nq, L->sorted_pmz.p, (int)L->n, charge, tol, mode, L->lo.p, L->cnt.p);
  ASL_CHECK_LAUNCH();
  std::vector<int32_t> h_cnt((size_t)nq), h_off((size_t)nq + 1, 0);
  ASL_TRY(L->cnt.download(h_cnt.data(), (size_t)nq));
  ASL_TRY(sync_stream());
  int64_t acc = 0;
  for (int q = 0; q < nq; q++) {
    h_off[(size_t)q] = (int32_t)acc;
    acc += h_cnt[(size_t)q];
    if (acc > 0x7fffffffLL)
      return fail(ASL_ERR_CAPACITY, "window: more than 2^31-1 candidate pairs in one batch; "
                                    "use a smaller batch_size for brute-force open search");
  }
  h_off[(size_t)nq] = (int32_t)acc;
  *total = acc;
  ASL_TRY(L->woff.upload(h_off.data(), (size_t)nq + 1));
  ASL_TRY(L->cand.reserve((size_t)std::max<int64_t>(acc, 1)));
  hipLaunchKernelGGL(window_fill_kernel, dim3(nq), dim3(256), 0, stream(), L->lo.p, L->woff.p,
                     L->sorted_row.p, L->has_valid ? L->valid.p : nullptr, L->cand.p);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int asl_window_candidates(asl_library_t *L, int32_t nq, const double *query_pmz, int32_t charge,
                          double tol, int32_t mode, int32_t *cand_offsets, int64_t *cand_rows) {
  clear_error();
  if (!L || nq < 0 || !cand_offsets) return fail(ASL_ERR_INVALID, "window_candidates: bad arguments");
  if (nq == 0) {
    cand_offsets[0] = 0;
    return ASL_OK;
  }
  In<double> dq;
  ASL_TRY(dq.init(query_pmz, (size_t)nq));
  int64_t total = 0;
  ASL_TRY(window_candidates_device(L, nq, dq.d, charge, tol, mode, &total));
  std::vector<int32_t> h_off((size_t)nq + 1), h_cand((size_t)total);
  ASL_TRY(L->woff.download(h_off.data(), (size_t)nq + 1));
  if (total) ASL_TRY(L->cand.download(h_cand.data(), (size_t)total));
  ASL_TRY(sync_stream());
  // compact invalid rows, ascending row order inside each list (spectral_library.py:451)
  std::vector<int32_t> out_off((size_t)nq + 1, 0);
  std::vector<int64_t> rows;
  rows.reserve((size_t)total);
  for (int q = 0; q < nq; q++) {
    const size_t b = rows.size();
    for (int32_t t = h_off[(size_t)q]; t < h_off[(size_t)q + 1]; t++)
      if (h_cand[(size_t)t] >= 0) rows.push_back(h_cand[(size_t)t]);
    std::sort(rows.begin() + (long)b, rows.end());
    out_off[(size_t)q + 1] = (int32_t)rows.size();
  }
  HIP_TRY(hipMemcpy(cand_offsets, out_off.data(), ((size_t)nq + 1) * 4, hipMemcpyDefault));
  if (cand_rows && !rows.empty())
    HIP_TRY(hipMemcpy(cand_rows, rows.data(), rows.size() * 8, hipMemcpyDefault));
  return ASL_OK;
}

int asl_rescore_knn(asl_library_t *L, const asl_peaks_t *queries, const asl_search_params_t *P,
                    const int64_t *knn_I, int32_t *best_row, double *best_score,
                    int32_t *n_cand, int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride) {
  clear_error();
  if (!L || !queries || !P || !knn_I) return fail(ASL_ERR_INVALID, "rescore_knn: null argument");
  const int nq = queries->n, k = P->k;
  if (nq == 0) return ASL_OK;
  if (k <= 0) return fail(ASL_ERR_INVALID, "rescore_knn: k must be positive");
  if (pm_pairs && pm_stride <= 0) return fail(ASL_ERR_INVALID, "rescore_knn: pm_stride");
  ASL_TRY(ensure_device());
  PeaksStage Q;
  ASL_TRY(Q.init(queries));
  In<int64_t> knn;
  ASL_TRY(knn.init(knn_I, (size_t)nq * k));
  Out<int32_t> o_row, o_ncand, o_cnt;
  Out<double> o_score;
  Out<uint32_t> o_pairs;
  ASL_TRY(o_row.init(best_row, nq));
  ASL_TRY(o_score.init(best_score, nq));
  ASL_TRY(o_ncand.init(n_cand, nq));
  ASL_TRY(o_cnt.init(pm_count, nq));
  ASL_TRY(o_pairs.init(pm_pairs, (size_t)nq * (pm_pairs ? pm_stride : 0) * 2));
  ASL_TRY(L->best_slot.reserve((size_t)nq));
  ASL_TRY(L->status.reserve(1));
  ASL_TRY(L->pair_score.reserve((size_t)nq * k));
  // the precursor filter runs inside the rescoring kernel's compaction stage
  PrecFilter flt;
  flt.lib_pmz = L->pmz32.p;
  flt.valid = L->has_valid ? L->valid.p : nullptr;
  library_filter(L, flt);
  flt.tol = P->precursor_tol;
  flt.mode = P->precursor_mode;
  flt.charge = P->charge;
  ASL_TRY(rescore_device(Q.dev, L->dev, knn.d, nullptr, nullptr, k, (int64_t)nq * k,
                         P->fragment_mz_tolerance, P->allow_shift, 1, L->pair_score.p,
                         L->best_slot.p, nullptr, o_row.d, o_score.d, o_ncand.d, o_cnt.d,
                         o_pairs.d, pm_stride, L->status.p, flt, true, &L->rs_scratch));
  ASL_TRY(o_row.finish());
  ASL_TRY(o_score.finish());
  ASL_TRY(o_ncand.finish());
  ASL_TRY(o_cnt.finish());
  ASL_TRY(o_pairs.finish());
  return rescore_check_status(L->status.p);
}

// asl_search_batch in pipeline mode (asl_set_pipeline): nothing here waits for the device.
//   stream A: encode -> coarse GEMM -> coarse select          (MFMA-bound, ~1.1 ms of a 16 384 batch)
//   stream B: list scan                                       (fabric / VALU bound, ~6.3 ms)
//   stream C (three-stream mode; else B): filter + rescoring -> peak matches (VALU bound, ~2.2 ms)
// so the front of the next batch (and in three-stream mode the rescoring of the previous one)
// runs under the scan of this one.
// Buffers written by one stream and read by the next (hashed queries + probe lists: A -> B;
// neighbour ids: B -> C) exist twice; the producer re-uses a pair only after the consumer of
// the batch that read it has finished (ev_scan / ev_resc). Everything else is touched by one
// stream only. Errors the kernels flag are sticky and reported by the next call that drains
// (asl_synchronize or any other entry point).
static int search_batch_pipelined(asl_library *L, asl_index *idx, const asl_peaks_t *queries,
                                  const asl_search_params_t *P, int32_t *best_row,
                                  double *best_score, int32_t *n_cand, int32_t *pm_count,
                                  uint32_t *pm_pairs, int32_t pm_stride, int64_t *knn_I) {
  Pipeline &pp = pipeline();
  struct InCall {
    Pipeline &p;
    explicit InCall(Pipeline &q) : p(q) { p.in_call = true; }
    ~InCall() { p.in_call = false; }
  } guard(pp);
  const int nq = queries->n, k = P->k, d = index_dim(idx);
  const int nprobe = index_nprobe(idx, P->nprobe);
  PeaksStage Q;
  ASL_TRY(Q.init(queries));   // device pointers + known peak count: no copy, no wait
  const int par = pp.parity;
  // allocations first (growing a buffer synchronises the device: only ever on the first batches)
  ASL_TRY(index_prepare(idx));
  ASL_TRY(L->p_qvec[par].reserve((size_t)nq * d));
  ASL_TRY(L->p_cD[par].reserve((size_t)nq * nprobe));
  ASL_TRY(L->p_cI[par].reserve((size_t)nq * nprobe));
  ASL_TRY(L->p_knn[par].reserve((size_t)nq * k));
  ASL_TRY(L->p_ent[par].reserve((size_t)nq * 64));
  ASL_TRY(L->p_cnt[par].reserve((size_t)nq));
  ASL_TRY(L->p_rows[par].reserve((size_t)nq));
  ASL_TRY(L->pair_score.reserve((size_t)nq * k));
  ASL_TRY(L->best_slot.reserve((size_t)nq));
  pp.parity ^= 1;
  // the caller's stream produced the inputs (and owns the output memory) up to here
  HIP_TRY(hipEventRecord(pp.ev_in, stream()));
  HIP_TRY(hipStreamWaitEvent(pp.A, pp.ev_in, 0));
  HIP_TRY(hipStreamWaitEvent(pp.B, pp.ev_in, 0));
  HIP_TRY(hipStreamWaitEvent(pp.C, pp.ev_in, 0));
  pp.inflight = true;
  bool rows_filtered = false;
  {
    StreamScope on_a(pp.A);
    // The buffers of this parity were last read by the scan of batch i-2: the front of batch i
    // starts when that scan ends, i.e. it runs under the RESCORING of batch i-2. Measured
    // (profiles/r02_pipeline_ab.txt): holding it back until the batch has finished, so that it
    // runs under the next scan instead, is worse -- the scan loses 1.3 ms to a 0.9 ms GEMM
    // beside it (9.88 ms per step), the rescoring only 0.6 ms (9.35 ms).
    if (pp.scan_recorded[par]) HIP_TRY(hipStreamWaitEvent(pp.A, pp.ev_scan[par], 0));
    ASL_TRY(encode_device(Q.dev.mz, Q.dev.intensity, Q.dev.offsets, nq, P->min_bound, P->bin_size,
                          d, P->hash_seed, 1, L->p_qvec[par].p));
    ASL_TRY(index_coarse_device(idx, nq, L->p_qvec[par].p, nprobe, L->p_cD[par].p, L->p_cI[par].p,
                                L->p_ent[par].p, L->p_cnt[par].p, &L->p_have_ent[par]));
    HIP_TRY(hipEventRecord(pp.ev_front[par], pp.A));
  }
  {
    StreamScope on_b(pp.B);
    HIP_TRY(hipStreamWaitEvent(pp.B, pp.ev_front[par], 0));
    if (pp.resc_recorded[par]) HIP_TRY(hipStreamWaitEvent(pp.B, pp.ev_resc[par], 0));
    // (the entry lists of the coarse stage, when it made them: the scan does not list the rows again)
    if (knn_I == nullptr && scan_postfilter_on()) offer_post_filter(L, idx, Q.dev, P, L->p_rows[par].p);
    const int rc_scan = index_search_device(idx, nq, L->p_qvec[par].p, k, nprobe, nullptr, knn_I,
                                            L->p_knn[par].p, L->p_cD[par].p, L->p_cI[par].p,
                                            knn_I == nullptr, nullptr, L->p_have_ent[par] ? L->p_ent[par].p : nullptr,
                                            L->p_have_ent[par] ? L->p_cnt[par].p : nullptr);
    rows_filtered = index_post_filter_applied(idx);
    ASL_TRY(rc_scan);
    HIP_TRY(hipEventRecord(pp.ev_scan[par], pp.B));
    pp.scan_recorded[par] = true;
  }
  {
    // measured (profiles/r02_pipeline_ab.txt): scan and rescoring are both VALU-limited, so a
    // third stream only makes them share the CUs -- the default keeps rescoring behind its scan
    hipStream_t sc = pp.streams == 3 ? pp.C : pp.B;
    StreamScope on_c(sc);
    HIP_TRY(hipStreamWaitEvent(sc, pp.ev_scan[par], 0));
    PrecFilter flt;
    flt.lib_pmz = L->pmz32.p;
    flt.valid = L->has_valid ? L->valid.p : nullptr;
    library_filter(L, flt);
    flt.tol = P->precursor_tol;
    flt.mode = P->precursor_mode;
    flt.charge = P->charge;
    ASL_TRY(rescore_device(Q.dev, L->dev, nullptr, L->p_knn[par].p, nullptr, k, (int64_t)nq * k,
                           P->fragment_mz_tolerance, P->allow_shift, 1, L->pair_score.p,
                           L->best_slot.p, nullptr, best_row, best_score, n_cand, pm_count,
                           pm_pairs, pm_stride, pp.status, flt, /*clear_status=*/false, &L->rs_scratch,
                           rows_filtered ? L->p_rows[par].p : nullptr));
    HIP_TRY(hipEventRecord(pp.ev_resc[par], sc));
    pp.resc_recorded[par] = true;
  }
  return ASL_OK;
}

int asl_search_batch(asl_library_t *L, asl_index_t *idx, const asl_peaks_t *queries,
                     const asl_search_params_t *P, int32_t *best_row, double *best_score,
                     int32_t *n_cand, int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride,
                     int64_t *knn_I) {
  clear_error();
  if (!L || !queries || !P) return fail(ASL_ERR_INVALID, "search_batch: null argument");
  const int nq = queries->n;
  if (nq == 0) return ASL_OK;
  if (pm_pairs && pm_stride <= 0) return fail(ASL_ERR_INVALID, "search_batch: pm_stride");
  if (P->use_ann && !idx) return fail(ASL_ERR_INVALID, "search_batch: use_ann needs an index");
  if (P->use_ann && P->k <= 0) return fail(ASL_ERR_INVALID, "search_batch: k must be positive");
  {
    // pipeline mode applies to ANN batches whose arguments all live on the device (nothing to
    // stage, nothing to copy back) and whose peak count the caller supplied; anything else takes
    // the synchronous path below, after the batches in flight have drained
    Pipeline &pp = pipeline();
    auto dev_or_null = [](const void *p) { return !p || is_device_ptr(p); };
    // (an exhaustive ASL_INDEX_FLAT index has no coarse stage to overlap: synchronous path)
    if (pp.on && P->use_ann && index_nprobe(idx, P->nprobe) > 0 && queries->n_peaks > 0 &&
        peaks_on_device(queries) &&
        is_device_ptr(best_row) && is_device_ptr(best_score) && dev_or_null(n_cand) &&
        dev_or_null(pm_count) && dev_or_null(pm_pairs) && dev_or_null(knn_I)) {
      pp.in_call = true;                    // do not drain: this call joins the pipeline
      const int rc = ensure_device();
      pp.in_call = false;
      ASL_TRY(rc);
      return search_batch_pipelined(L, idx, queries, P, best_row, best_score, n_cand, pm_count,
                                    pm_pairs, pm_stride, knn_I);
    }
  }
  ASL_TRY(ensure_device());
  PeaksStage Q;
  ASL_TRY(Q.init(queries));
  Out<int32_t> o_row, o_ncand, o_cnt;
  Out<double> o_score;
  Out<uint32_t> o_pairs;
  Out<int64_t> o_knn;
  ASL_TRY(o_row.init(best_row, nq));
  ASL_TRY(o_score.init(best_score, nq));
  ASL_TRY(o_ncand.init(n_cand, nq));
  ASL_TRY(o_cnt.init(pm_count, nq));
  ASL_TRY(o_pairs.init(pm_pairs, (size_t)nq * (pm_pairs ? pm_stride : 0) * 2));
  ASL_TRY(L->best_slot.reserve((size_t)nq));
  ASL_TRY(L->status.reserve(1));
  if (P->use_ann) {
    const int d = index_dim(idx), k = P->k;
    if (k <= 0) return fail(ASL_ERR_INVALID, "search_batch: k must be positive");
    ASL_TRY(o_knn.init(knn_I, (size_t)nq * k));
    ASL_TRY(L->qvec.reserve((size_t)nq * d));
    ASL_TRY(L->knn.reserve((size_t)nq * k));
    ASL_TRY(L->pair_score.reserve((size_t)nq * k));
    ASL_TRY(encode_device(Q.dev.mz, Q.dev.intensity, Q.dev.offsets, nq, P->min_bound, P->bin_size,
                          d, P->hash_seed, 1, L->qvec.p));
    // the candidates are consumed as a set (filter + best match): no final sort unless the
    // caller asked for the ordered neighbour list
    ASL_TRY(L->rows_len.reserve((size_t)nq));
    if (knn_I == nullptr && scan_postfilter_on()) offer_post_filter(L, idx, Q.dev, P, L->rows_len.p);
    const int rc_scan = index_search_device(idx, nq, L->qvec.p, k, P->nprobe, nullptr, o_knn.d, L->knn.p,
                                            nullptr, nullptr, knn_I == nullptr);
    const bool rows_filtered = index_post_filter_applied(idx);
    ASL_TRY(rc_scan);
    PrecFilter flt;
    flt.lib_pmz = L->pmz32.p;
    flt.valid = L->has_valid ? L->valid.p : nullptr;
    library_filter(L, flt);
    flt.tol = P->precursor_tol;
    flt.mode = P->precursor_mode;
    flt.charge = P->charge;
    ASL_TRY(rescore_device(Q.dev, L->dev, nullptr, L->knn.p, nullptr, k, (int64_t)nq * k,
                           P->fragment_mz_tolerance, P->allow_shift, 1, L->pair_score.p,
                           L->best_slot.p, nullptr, o_row.d, o_score.d, o_ncand.d, o_cnt.d,
                           o_pairs.d, pm_stride, L->status.p, flt, true, &L->rs_scratch,
                           rows_filtered ? L->rows_len.p : nullptr));
  } else {
    int64_t total = 0;
    {
      ProfScope ps("filter");
      ASL_TRY(window_candidates_device(L, nq, Q.dev.precursor_mz, P->charge, P->precursor_tol,
                                       P->precursor_mode, &total));
    }
    ASL_TRY(L->pair_score.reserve((size_t)std::max<int64_t>(total, 1)));
    PrecFilter rows_only;       // packed row records for the kernels, no second filtering
    library_filter(L, rows_only);
    rows_only.wcol = nullptr;
    rows_only.pass_all = true;
    ASL_TRY(rescore_device(Q.dev, L->dev, nullptr, L->cand.p, L->woff.p, 0, total,
                           P->fragment_mz_tolerance, P->allow_shift, 1, L->pair_score.p,
                           L->best_slot.p, nullptr, o_row.d, o_score.d, o_ncand.d, o_cnt.d,
                           o_pairs.d, pm_stride, L->status.p, rows_only, true, &L->rs_scratch));
  }
  ASL_TRY(o_row.finish());
  ASL_TRY(o_score.finish());
  ASL_TRY(o_ncand.finish());
  ASL_TRY(o_cnt.finish());
  ASL_TRY(o_pairs.finish());
  ASL_TRY(o_knn.finish());
  return rescore_check_status(L->status.p);  // synchronises the stream
}

}  // extern "C"
