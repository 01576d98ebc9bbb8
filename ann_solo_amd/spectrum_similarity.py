"""Batched SSM similarity features on the device: host mirror of the reference's
``SpectrumSimilarityCalculator`` (/root/reference/src/ann_solo/spectrum_similarity.py:13-730)
and of the similarity part of ``_compute_ssm_features``
(/root/reference/src/ann_solo/utils.py:276-457), over ``asl_ssm_features_batch``.

The reference evaluates ~35 NumPy/SciPy calls per spectrum-spectrum match in a Python
loop; here one kernel launch fills the whole ``[n_ssm, 33]`` feature matrix from the packed
peak store and the peak matches the rescoring kernel already emitted.
"""
import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from .packed import PackedSpectra

#: column order of ``asl_ssm_features_batch`` = key order of the reference's feature
#: dictionary (utils.py:309-342)
FEATURE_NAMES = [
    'cosine', 'cosine_top5', 'n_matched_peaks', 'frac_n_peaks_query', 'frac_n_peaks_lib',
    'frac_n_peaks_lib_top5', 'frac_int_query', 'frac_int_lib', 'frac_int_lib_top5', 'mse_mz',
    'mse_mz_top5', 'mse_int', 'mse_int_top5', 'contrast_angle', 'contrast_angle_top5',
    'hypergeometric_score', 'kendalltau', 'ms_for_id_v1', 'ms_for_id_v2', 'entropy_unweighted',
    'entropy_weighted', 'scribe_fragment_acc', 'scribe_fragment_acc_top5', 'manhattan',
    'euclidean', 'chebyshev', 'pearsonr', 'pearsonr_top5', 'spearmanr', 'spearmanr_top5',
    'braycurtis', 'canberra', 'ruzicka']
N_FEATURES = len(FEATURE_NAMES)


def ssm_features(queries: PackedSpectra, library: PackedSpectra, lib_rows, pm_pairs, pm_count,
                 min_mz: float = 11, max_mz: float = 2010, bin_size: float = 0.04, top: int = 5):
    """Feature matrix ``[nq, 33]`` (float64) of the SSMs (query i, library row ``lib_rows[i]``)
    with peak matches ``pm_pairs[i, :pm_count[i]]``; rows without a match are NaN. Arrays may
    be numpy (host) or torch (device); the result lives where ``lib_rows`` lives."""
    nq = queries.n
    if hasattr(lib_rows, 'data_ptr'):
        import torch
        lib_rows = lib_rows.to(torch.int32).contiguous()
        pm_pairs, pm_count = pm_pairs.contiguous(), pm_count.to(torch.int32).contiguous()
        out = torch.empty((nq, N_FEATURES), dtype=torch.float64, device=lib_rows.device)
    else:
        lib_rows = np.ascontiguousarray(lib_rows, np.int32)
        pm_pairs = np.ascontiguousarray(pm_pairs, np.uint32)
        pm_count = np.ascontiguousarray(pm_count, np.int32)
        out = np.empty((nq, N_FEATURES), np.float64)
    if nq == 0:
        return out
    stride = int(pm_pairs.shape[1])
    _lib.check(_lib.lib().asl_ssm_features_batch(
        C.byref(_lib.peaks_struct(queries)), C.byref(_lib.peaks_struct(library)),
        _lib.ptr(lib_rows), _lib.ptr(pm_pairs), _lib.ptr(pm_count), stride, float(min_mz),
        float(max_mz), float(bin_size), int(top), _lib.ptr(out)))
    return out


def ssm_cosine(queries: PackedSpectra, library: PackedSpectra, lib_rows, pm_pairs, pm_count):
    """Column 0 of ``ssm_features`` alone (``SpectrumSimilarityCalculator.cosine``,
    spectrum_similarity.py:81-106): the cascade's default search-engine score. ``[nq]``
    float64, same bits as ``ssm_features(...)[:, 0]``; lives where ``lib_rows`` lives."""
    nq = queries.n
    if hasattr(lib_rows, 'data_ptr'):
        import torch
        lib_rows = lib_rows.to(torch.int32).contiguous()
        pm_pairs, pm_count = pm_pairs.contiguous(), pm_count.to(torch.int32).contiguous()
        out = torch.empty(nq, dtype=torch.float64, device=lib_rows.device)
    else:
        lib_rows = np.ascontiguousarray(lib_rows, np.int32)
        pm_pairs = np.ascontiguousarray(pm_pairs, np.uint32)
        pm_count = np.ascontiguousarray(pm_count, np.int32)
        out = np.empty(nq, np.float64)
    if nq == 0:
        return out
    _lib.check(_lib.lib().asl_ssm_cosine_batch(
        C.byref(_lib.peaks_struct(queries)), C.byref(_lib.peaks_struct(library)),
        _lib.ptr(lib_rows), _lib.ptr(pm_pairs), _lib.ptr(pm_count), int(pm_pairs.shape[1]),
        _lib.ptr(out)))
    return out


def compute_ssm_features(queries: PackedSpectra, library: PackedSpectra, result,
                         config=None) -> Dict[str, np.ndarray]:
    """``_compute_ssm_features`` (utils.py:276-457) for one batch: ``result`` is the
    ``BatchResult`` of ``SpectralLibrary._search_batch`` over ``queries`` against the charge
    partition ``library``. SSMs without peak matches are skipped (utils.py:345-346). The
    sequence-derived columns (``sequence``, ``sequence_len``, ``is_target``) stay with the
    caller, who owns the library metadata."""
    cfg = config
    g = (lambda k, d: getattr(cfg, k, d) if cfg is not None else d)
    to_np = lambda a: a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)
    feats = to_np(ssm_features(queries, library, result.best_row, result.pm_pairs,
                               result.pm_count, g('min_mz', 11), g('max_mz', 2010),
                               g('bin_size', 0.04)))
    rows, cnt = to_np(result.best_row), to_np(result.pm_count)
    keep = np.nonzero((rows >= 0) & (cnt > 0))[0]
    q_pmz = to_np(queries.precursor_mz)[keep]
    l_pmz = to_np(library.precursor_mz)[rows[keep]]
    z = to_np(queries.precursor_charge)[keep]
    out = {'index': keep,
           'precursor_charge_2': (z <= 2).astype(np.int64),        # utils.py:353-372
           'precursor_charge_3': (z == 3).astype(np.int64),
           'precursor_charge_4': (z == 4).astype(np.int64),
           'precursor_charge_5': (z >= 5).astype(np.int64),
           'query_prec_mz': q_pmz, 'lib_prec_mz': l_pmz,
           # spectrum_utils.utils.mass_diff(mz1, mz2, mode_is_da)
           'mz_diff_ppm': (q_pmz - l_pmz) / l_pmz * 10 ** 6,
           'abs_mz_diff_ppm': np.abs((q_pmz - l_pmz) / l_pmz * 10 ** 6),
           'mz_diff_da': q_pmz - l_pmz, 'abs_mz_diff_da': np.abs(q_pmz - l_pmz)}
    for f, name in enumerate(FEATURE_NAMES):
        out[name] = feats[keep, f]
    out['n_matched_peaks'] = out['n_matched_peaks'].astype(np.int64)
    return out


class SpectrumSimilarityCalculator:
    """Single-SSM view with the reference's method names (spectrum_similarity.py:13-700);
    every value comes from one ``asl_ssm_features_batch`` call. ``top`` selects the
    ``*_top5`` columns utils.py requests; the top-peak variants it never requests
    (kendalltau, ms_for_id_v1, hypergeometric_score, n_matched_peaks) are not provided."""
    _COL = {'cosine': (0, 1), 'n_matched_peaks': (2, None), 'frac_n_peaks_query': (3, None),
            'frac_n_peaks_library': (4, 5), 'frac_intensity_query': (6, None),
            'frac_intensity_library': (7, 8), 'spectral_contrast_angle': (13, 14),
            'kendalltau': (16, None), 'ms_for_id_v1': (17, None), 'ms_for_id_v2': (18, None),
            'scribe_fragment_acc': (21, 22), 'manhattan': (23, None), 'euclidean': (24, None),
            'chebyshev': (25, None), 'pearsonr': (26, 27), 'spearmanr': (28, 29),
            'braycurtis': (30, None), 'canberra': (31, None), 'ruzicka': (32, None)}

    def __init__(self, ssm, top: Optional[int] = None):
        self._ssm, self._top = ssm, top
        self._cache = {}

    def _features(self, min_mz=11, max_mz=2010, bin_size=0.04):
        key = (min_mz, max_mz, bin_size)
        if key not in self._cache:
            q = PackedSpectra.from_spectra([self._ssm.query_spectrum])
            lib = PackedSpectra.from_spectra([self._ssm.library_spectrum])
            pm = np.asarray(self._ssm.peak_matches, np.uint32).reshape(-1, 2)
            pairs = np.zeros((1, max(1, len(pm)), 2), np.uint32)
            pairs[0, :len(pm)] = pm
            self._cache[key] = ssm_features(q, lib, np.zeros(1, np.int32), pairs,
                                            np.array([len(pm)], np.int32), min_mz, max_mz,
                                            bin_size, self._top or 5)[0]
        return self._cache[key]

    def _get(self, name):
        full, top = self._COL[name]
        if self._top is not None:
            if top is None:
                raise NotImplementedError(f'{name} is not defined when filtering by the top '
                                          'intensity library peaks')
            return float(self._features()[top])
        return float(self._features()[full])

    def __getattr__(self, name):
        if name in SpectrumSimilarityCalculator._COL:
            return lambda: (int(self._get(name)) if name == 'n_matched_peaks'
                            else self._get(name))
        raise AttributeError(name)

    def mean_squared_error(self, axis: str) -> float:
        if axis not in ('mz', 'intensity'):
            raise ValueError('Unknown axis specified')
        col = (9 if axis == 'mz' else 11) + (1 if self._top is not None else 0)
        return float(self._features()[col])

    def hypergeometric_score(self, min_mz: float, max_mz: float, fragment_mz_tol: float) -> float:
        if self._top is not None:
            raise NotImplementedError('hypergeometric_score over the top peaks is not batched')
        return float(self._features(min_mz, max_mz, fragment_mz_tol)[15])

    def entropy(self, weighted: bool = False) -> float:
        if self._top is not None:
            raise NotImplementedError('The spectral entropy is not defined when filtering by '
                                      'the top intensity library peaks')
        return float(self._features()[20 if weighted else 19])
