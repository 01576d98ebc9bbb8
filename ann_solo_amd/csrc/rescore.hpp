// rescore.hpp -- interface of the rescoring driver (rescore.hip) shared with search.hip.
#pragma once
#include "common.hpp"

namespace asl {

// Precursor-window post-filter of the candidate lists, evaluated inside the rescoring
// kernel's compaction stage (spectral_library.py:417-429 AND :441-446): a candidate row
// passes if it is valid and within the window of the query's precursor m/z.
// lib_pmz == nullptr switches the filter off.
// Everything the rescoring kernel needs to know about a library row, in one 32-byte sector
// (asl_library: the head of the row's slot, see PrecFilter::meta_stride):
// the candidate filter and the per-candidate metadata cost one random gather each instead of
// one per field (offsets x2, charge, precursor m/z, window column, validity).
struct __attribute__((aligned(32))) RowMeta {
  int32_t off;      // first peak of the spectrum
  int32_t cn;       // number of peaks
  int32_t charge;   // precursor charge
  float pmz32;      // spec_info's float32 precursor m/z column; NaN for invalid spectra
  double pmz64;     // precursor m/z
  uint32_t rec4;    // the spectrum's packed peak record (DevPeaks::records), in 4-byte units
  uint32_t pad;
};

struct PrecFilter {
  const float *lib_pmz = nullptr;   // spec_info's float32 precursor m/z column
  const uint8_t *valid = nullptr;   // is_valid flags (nullptr: all valid)
  double tol = 0.0;
  int mode = ASL_TOL_DA;
  int charge = 0;
  const RowMeta *meta = nullptr;    // packed rows (asl_library): replaces lib_pmz / valid
  // bytes from one row's record to the next: sizeof(RowMeta) for a plain array; the library keeps
  // every row record at the head of its row's fixed-size SLOT, right in front of the packed peaks
  // (one gather brings the record AND the first peaks; the peaks' address needs no second hop)
  uint32_t meta_stride = sizeof(RowMeta);
  // the window column alone, NaN for invalid spectra (4 bytes per row: the flat kernel filters
  // 16.7 M slots per batch on it and touches the 32-byte records of the survivors only)
  const float *wcol = nullptr;
  // window-only modes: the candidate rows ARE the window's hits; `meta` is passed for the packed
  // rows (one gather per candidate, one peak record) and nothing is filtered
  bool pass_all = false;
};

__device__ __forceinline__ const RowMeta *meta_row(const PrecFilter &f, long long row) {
  return reinterpret_cast<const RowMeta *>(reinterpret_cast<const char *>(f.meta) + (size_t)row * f.meta_stride);
}

// precursor_ok (spectral_library.py:421-427): common.hpp -- the scans' finish applies it too

__device__ __forceinline__ bool filter_pass(const PrecFilter &f, double q_pmz, long long row) {
  if (f.pass_all) return true;
  if (f.wcol) return precursor_ok(q_pmz, f.wcol[row], f.charge, f.tol, f.mode);
  if (f.meta) return precursor_ok(q_pmz, meta_row(f, row)->pmz32, f.charge, f.tol, f.mode);
  if (!f.lib_pmz) return true;
  if (f.valid && !f.valid[row]) return false;
  return precursor_ok(q_pmz, f.lib_pmz[row], f.charge, f.tol, f.mode);
}

// Per-query flags that pass from one rescoring launch to the next (which queries the flat kernel
// left to the binary-search kernel, which winners the small matches kernel left to the full-size
// one). Owned by whoever owns the stream the launches go to -- a library handle (its batches are
// issued in order on its own stream) or a one-shot call -- so that two handles pipelined on
// different streams never share, or re-allocate under each other, a buffer in flight.
struct RescoreScratch {
  DevBuf<int> q_defer, m_defer;
};

// Host driver shared by asl_rescore_batch, asl_search_batch and asl_rescore_knn. All pointers
// are device pointers. pair_score scratch must hold one double per candidate slot.
int rescore_device(const DevPeaks &Q, const DevPeaks &L, const int64_t *rows64,
                   const int32_t *rows32, const int32_t *cand_offsets, int32_t stride,
                   int64_t total_slots, double tol, int allow_shift, int tie_by_row,
                   double *pair_score, long long *best_slot, int32_t *best_cand,
                   int32_t *best_row, double *best_score, int32_t *n_valid,
                   int32_t *pm_count, uint32_t *pm_pairs, int32_t pm_stride, int *status,
                   const PrecFilter &filter = PrecFilter(), bool clear_status = true,
                   RescoreScratch *scratch = nullptr,
                   // fixed-stride rows: their lengths as the scans' post-filter wrote them (-1: unfiltered row)
                   const int32_t *row_counts = nullptr);
int rescore_check_status(const int *status_dev);   // reads the flags back: synchronises
int rescore_status_error(int status_bits);         // ASL_OK or the error the flags stand for

}  // namespace asl
