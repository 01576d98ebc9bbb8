"""The cascade's FDR gate without semi-supervised learning (``--model none``): target-decoy
q-values on the cosine, per precursor-mass-difference group for an open search.

Mirrors ``utils.score_ssms(ssms, fdr, None, grouped)`` (/root/reference/src/ann_solo/utils.py:
69-201 with ``model is None``, :139-142) as a COLUMNAR scorer for ``SpectralLibrary``: it reads
and writes the columns of an ``SSMTable``; no per-SSM objects, no feature table (the 33
features of :104 are only consumed by the rf / svm models, which stay out of scope).

  * ``ssm_groups``  -- ``utils._get_ssm_groups`` (:204-274): the reference's own code, so the
    golden vectors in tests/golden/fdr_groups_golden.npz come from running it.
  * ``tdc_qvalues`` -- the q-value estimate mokapot applies there
    (``LinearPsmDataset.assign_confidence(scores, desc=True)`` -> ``mokapot.qvalues.tdc``).
    mokapot is a third-party dependency that is not vendored in /root/reference and not
    installed here (the reference's environment.yml names it without a pin): its published
    algorithm is restated and pinned by the constants of the reference's own test
    (src/tests/utils_test.py:60-80, committed as tests/golden/fdr_kat.json).
"""
from typing import Optional

import numpy as np

__all__ = ['tdc_qvalues', 'ssm_groups', 'CosineTDC']


def tdc_qvalues(scores, is_target, desc: bool = True) -> np.ndarray:
    """q-values by target-decoy competition (mokapot ``qvalues.tdc``): walking the matches from
    best to worst, FDR = (#decoys + 1) / #targets (1 where there is no target yet); matches of
    equal score form one step that takes the FDR at its END; q = the smallest FDR at this or
    any worse step, never above 1. Returned in input order, for targets and decoys alike."""
    scores = np.asarray(scores, np.float64)
    target = np.asarray(is_target, bool)
    if scores.shape != target.shape or scores.ndim != 1:
        raise ValueError('scores and is_target: 1-D arrays of one length')
    if np.isnan(scores).any():
        raise ValueError('scores must not hold NaN')
    n = len(scores)
    if n == 0:
        return np.zeros(0, np.float64)
    order = np.argsort(-scores if desc else scores, kind='stable')
    s, t = scores[order], target[order]
    cum_t = np.cumsum(t)
    cum_d = np.cumsum(~t)
    fdr = np.ones(n, np.float64)
    np.divide(cum_d + 1, cum_t, out=fdr, where=cum_t != 0)
    # last position of every run of equal scores carries the run's FDR
    last = np.nonzero(np.append(s[1:] != s[:-1], True))[0]
    run_fdr = np.minimum(fdr[last], 1.0)
    run_q = np.minimum.accumulate(run_fdr[::-1])[::-1]          # min over this and worse runs
    run_of = np.searchsorted(last, np.arange(n), side='left')
    q = np.empty(n, np.float64)
    q[order] = run_q[run_of]
    return q


def ssm_groups(mass_diffs, min_group_size: int) -> np.ndarray:
    """Group labels from the precursor mass differences (``utils._get_ssm_groups``, :204-274):
    per nominal mass difference a 100-bin histogram over [nominal - 0.5, nominal + 0.5], its
    peaks (``scipy.signal.find_peaks`` with prominences, as the reference), every SSM to the
    NEAREST peak (left bin edge) among those whose bases enclose it; groups smaller than
    ``min_group_size`` and unassigned SSMs -> -1.

    Two quirks of the reference's loop are kept: a run is closed when the nominal mass changes
    OR at the very last SSM (which then joins the run it closes only when it has the same
    nominal mass), so an SSM that is last in mass order and alone in its interval is never
    assigned."""
    import scipy.signal
    md = np.asarray(mass_diffs, np.float64)
    n = len(md)
    groups = -np.ones(n, np.int32)
    if n == 0:
        return groups
    order = np.argsort(md)
    nominal = np.rint(md[order])                 # round(): half to even, like np.float64.__round__
    starts = np.nonzero(np.append(True, nominal[1:] != nominal[:-1]))[0]
    ends = np.append(starts[1:], n)
    group = 0
    for a, b in zip(starts, ends):
        if b == n and b - a == 1 and n > 1:      # last SSM opening its own interval: skipped
            continue
        if n == 1:                               # a single SSM: the loop never closes a run
            continue
        members = order[a:b]
        g = nominal[a]
        bins = np.linspace(g - 0.5, g + 0.5, 101)
        hist, _ = np.histogram(md[members], bins=bins)
        peaks, prom = scipy.signal.find_peaks(hist, prominence=(None, None))
        if len(peaks):
            x = md[members][:, None]
            inside = (bins[prom['left_bases']][None, :] < x) & (x < bins[prom['right_bases']][None, :])
            dist = np.where(inside, np.abs(bins[peaks][None, :] - x), np.inf)
            best = np.argmin(dist, axis=1)       # first of equal distances, like the strict <
            ok = np.isfinite(dist[np.arange(len(members)), best])
            groups[members[ok]] = group + best[ok].astype(np.int32)
        group += len(peaks)
    labels, counts = np.unique(groups, return_counts=True)
    small = labels[counts < min_group_size]
    groups[np.isin(groups, small)] = -1
    return groups


class CosineTDC:
    """``score_ssms=`` for ``SpectralLibrary`` when the reference would run with ``--model
    none``: q-values from the cosine by target-decoy competition, grouped by mass difference
    for the open level (utils.py:118, spectral_library.py:319-326). Targets get their cosine as
    ``search_engine_score`` and a q-value; decoys keep NaN for both (mokapot's confidence table
    lists targets only, utils.py:186-200), so the ``q < fdr`` gate drops them."""
    columnar = True

    def __init__(self, min_group_size: int = 100):
        self.min_group_size = int(min_group_size)
        self.n_groups: Optional[int] = None      # of the last grouped call (utils.py:127-131)

    def __call__(self, table, mode: str):
        n = len(table)
        target = ~table.is_decoy()
        cos = np.asarray(table.score, np.float64)
        if mode == 'open':
            groups = ssm_groups(table.mass_diffs(), self.min_group_size)
            self.n_groups = len(np.unique(groups))
        else:
            groups = np.zeros(n, np.int32)
        q = np.full(n, np.nan)
        for g in np.unique(groups):
            m = groups == g
            q[m] = tdc_qvalues(cos[m], target[m])
        table.q = np.where(target, q, np.nan)
        table.score = np.where(target, cos, np.nan)
        return None
