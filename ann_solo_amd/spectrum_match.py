"""(Shifted) dot-product rescoring -- host-side mirror of the reference's
``ann_solo/spectrum_match.pyx`` (``get_best_match`` :28-108) on top of the HIP
kernels behind ``asl_rescore_batch``.
"""
import numpy as np

from . import _lib
from .packed import PackedSpectra


def get_best_match(query, candidates, fragment_mz_tolerance, allow_shift):
    """Drop-in for ``spectrum_match.get_best_match``: returns
    ``(best candidate object, score, [(query_peak, candidate_peak), ...])``."""
    if len(candidates) == 0:
        raise ValueError('get_best_match needs at least one candidate '
                         '(the reference guards this at spectral_library.py:359)')
    q = PackedSpectra.from_spectra([query])
    lib = PackedSpectra.from_spectra(candidates)
    offsets = np.array([0, len(candidates)], np.int32)
    rows = np.arange(len(candidates), dtype=np.int64)
    best, score, counts, pairs = rescore_batch(q, lib, rows, offsets, fragment_mz_tolerance,
                                               allow_shift)
    n = int(counts[0])
    return (candidates[int(best[0])], float(score[0]),
            [(int(a), int(b)) for a, b in pairs[0, :n]])


def rescore_batch(queries: PackedSpectra, library: PackedSpectra, cand_rows, cand_offsets,
                  fragment_mz_tolerance, allow_shift, pm_stride=None):
    """Batched ``get_best_match``: candidates of query q are
    ``cand_rows[cand_offsets[q]:cand_offsets[q+1]]`` (library rows). Returns numpy
    ``(best_cand[nq], best_score[nq], pm_count[nq], pm_pairs[nq, pm_stride, 2])``."""
    nq = queries.n
    if pm_stride is None:
        cnt = np.diff(np.asarray(queries.offsets.cpu()))
        pm_stride = int(cnt.max()) if nq else 1
    cand_rows = np.ascontiguousarray(cand_rows, np.int64) if isinstance(
        cand_rows, (list, np.ndarray)) else cand_rows
    cand_offsets = np.ascontiguousarray(cand_offsets, np.int32) if isinstance(
        cand_offsets, (list, np.ndarray)) else cand_offsets
    best = np.empty(nq, np.int32)
    score = np.empty(nq, np.float64)
    count = np.empty(nq, np.int32)
    pairs = np.zeros((nq, pm_stride, 2), np.uint32)
    qs, ls = _lib.peaks_struct(queries), _lib.peaks_struct(library)
    _lib.check(_lib.lib().asl_rescore_batch(
        qs, ls, _lib.ptr(cand_rows), _lib.ptr(cand_offsets), float(fragment_mz_tolerance),
        int(bool(allow_shift)), _lib.ptr(best), _lib.ptr(score), _lib.ptr(count),
        _lib.ptr(pairs), pm_stride))
    return best, score, count, pairs
