"""Shared helpers of the similarity-feature tests (CPU oracle and GPU kernel)."""
import numpy as np

# (method, args as written in the reference test, top) -> feature column
COLUMN = {
    ('cosine', '', False): 0, ('cosine', '', True): 1, ('n_matched_peaks', '', False): 2,
    ('frac_n_peaks_query', '', False): 3, ('frac_n_peaks_library', '', False): 4,
    ('frac_n_peaks_library', '', True): 5, ('frac_intensity_query', '', False): 6,
    ('frac_intensity_library', '', False): 7, ('frac_intensity_library', '', True): 8,
    ('mean_squared_error', "'mz'", False): 9, ('mean_squared_error', "'mz'", True): 10,
    ('mean_squared_error', "'intensity'", False): 11,
    ('mean_squared_error', "'intensity'", True): 12,
    ('spectral_contrast_angle', '', False): 13, ('spectral_contrast_angle', '', True): 14,
    ('hypergeometric_score', '', False): 15, ('kendalltau', '', False): 16,
    ('ms_for_id_v1', '', False): 17, ('ms_for_id_v2', '', False): 18,
    ('entropy', 'False', False): 19, ('entropy', 'True', False): 20,
    ('entropy', 'weighted=False', False): 19, ('entropy', 'weighted=True', False): 20,
    ('entropy', '', False): 19,
    ('scribe_fragment_acc', '', False): 21, ('scribe_fragment_acc', '', True): 22,
    ('manhattan', '', False): 23, ('euclidean', '', False): 24, ('chebyshev', '', False): 25,
    ('pearsonr', '', False): 26, ('pearsonr', '', True): 27, ('spearmanr', '', False): 28,
    ('spearmanr', '', True): 29, ('braycurtis', '', False): 30, ('canberra', '', False): 31,
    ('ruzicka', '', False): 32}


def kat_case(kat, name):
    return (kat[f'{name}_q_mz'], kat[f'{name}_q_intensity'], kat[f'{name}_l_mz'],
            kat[f'{name}_l_intensity'], kat[f'{name}_peak_matches'].astype(np.uint32))


def check_features(got, want, tag, rel=1e-5, abs_=5e-6):
    """Feature vectors agree to `rel`/`abs_`; the contrast angle inherits arccos' sensitivity
    near cosine = 1 (d angle = (2/pi) sqrt(2 d cos))."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    for f in range(len(want)):
        a, b = got[f], want[f]
        if np.isinf(b) or np.isnan(b):
            assert (np.isnan(a) and np.isnan(b)) or a == b, (tag, f, a, b)
            continue
        tol = abs_ + rel * abs(b)
        if f in (13, 14) and want[f - 13] > 0.999:
            tol = 1e-3
        assert abs(a - b) <= tol, (tag, f, a, b)
