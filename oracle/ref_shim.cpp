// ref_shim.cpp -- extern "C" driver around the REFERENCE's own scorer.
//
// TEST INFRASTRUCTURE. This file is ours; it is compiled together with
// /root/reference/src/ann_solo/SpectrumMatch.cpp *where it lies* (never copied)
// into oracle/_ref/libref_spectrummatch.so by oracle/Makefile. It replaces the
// Cython wrapper spectrum_match.pyx:28-108, keeping every buffer alive for the
// duration of the call (the wrapper's use-after-free, SURVEY.md 9.1, is thereby
// avoided: this is the *intended* semantics of SpectrumMatcher::dot).
#include <cstddef>
#include <cstdint>
#include <vector>

#include "SpectrumMatch.h"

extern "C" {

// Candidates are rows `cand_rows[0..n_cand)` of a packed SoA library.
// Returns the index into cand_rows of the best match (SpectrumMatch.cpp:118-129).
int32_t ref_best_match(double q_pmz, int32_t q_charge, int32_t q_n, const float *q_mz,
                       const float *q_int, const int32_t *lib_offsets, const float *lib_mz,
                       const float *lib_int, const uint8_t *lib_chg, const double *lib_pmz,
                       const int32_t *lib_pcharge, const int64_t *cand_rows, int32_t n_cand,
                       double tol, int allow_shift, double *score_out, uint32_t *matches_out,
                       int32_t *n_matches_out) {
  std::vector<uint8_t> qchg(q_n > 0 ? q_n : 1, 0);
  ann_solo::Spectrum query(q_pmz, (unsigned)q_charge, (unsigned)q_n, const_cast<float *>(q_mz),
                           const_cast<float *>(q_int), qchg.data());
  std::vector<ann_solo::Spectrum *> cands;
  for (int32_t c = 0; c < n_cand; c++) {
    int64_t r = cand_rows[c];
    int32_t o = lib_offsets[r], n = lib_offsets[r + 1] - o;
    cands.push_back(new ann_solo::Spectrum(lib_pmz[r], (unsigned)lib_pcharge[r], (unsigned)n,
                                           const_cast<float *>(lib_mz + o),
                                           const_cast<float *>(lib_int + o),
                                           const_cast<uint8_t *>(lib_chg + o)));
  }
  ann_solo::SpectrumMatcher matcher;
  ann_solo::SpectrumSpectrumMatch *res = matcher.dot(&query, cands, tol, allow_shift != 0);
  int32_t best = -1;
  if (res) {
    best = (int32_t)res->getCandidateIndex();
    if (score_out) *score_out = res->getScore();
    auto *pm = res->getPeakMatches();
    if (n_matches_out) *n_matches_out = (int32_t)pm->size();
    if (matches_out)
      for (size_t i = 0; i < pm->size(); i++) {
        matches_out[2 * i] = (*pm)[i].first;
        matches_out[2 * i + 1] = (*pm)[i].second;
      }
    delete res;
  }
  for (auto *c : cands) delete c;
  return best;
}
}
