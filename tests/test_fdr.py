"""The cascade's FDR gate without a learned model (tests/fdr_gate.py: test infrastructure,
the stand-in for ``utils.score_ssms(model=None)`` the cascade tests inject) against the reference:
group labels against ``utils._get_ssm_groups`` run on seeded mass differences, q-values against
the constants of the reference's own test (tests/golden/make_golden.py gen_fdr)."""
import json
import os

import numpy as np
import pytest

import fdr_gate as fdr
from ann_solo_amd.spectral_library import SSMTable

HERE = os.path.dirname(os.path.abspath(__file__))


def test_qvalues_reference_kat():
    kat = json.load(open(os.path.join(HERE, 'golden', 'fdr_kat.json')))
    cos, decoy = np.asarray(kat['cosine']), np.asarray(kat['is_decoy'])
    q = fdr.tdc_qvalues(cos, ~decoy)
    got = np.where(decoy, np.nan, q)
    np.testing.assert_array_equal(got, np.asarray(kat['q_expected'], np.float64))   # exact, as the reference asserts


def _tdc_loops(scores, target):
    """mokapot's published procedure, literally: cumulative counts over the sorted matches, runs
    of equal score take the FDR of their last member, running minimum from the worst run up."""
    order = np.argsort(-scores, kind='stable')
    s, t = scores[order], target[order]
    fdrs, nt, nd = [], 0, 0
    for x in t:
        nt, nd = nt + bool(x), nd + (not x)
        fdrs.append((nd + 1) / nt if nt else 1.0)
    q = np.ones(len(s))
    min_q, i = 1.0, len(s)
    while i > 0:
        j = i
        while j > 0 and s[j - 1] == s[i - 1]:
            j -= 1
        min_q = min(min_q, fdrs[i - 1])
        q[j:i] = min_q
        i = j
    out = np.empty(len(s))
    out[order] = q
    return out


@pytest.mark.parametrize('seed', range(6))
def test_qvalues_against_loops(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 13, 500, 4000]))
    target = rng.random(n) < rng.choice([0.5, 0.8, 1.0])
    scores = np.where(target, rng.beta(5, 2, n), rng.beta(2, 5, n))
    if seed % 2:
        scores = np.round(scores, 2)            # many ties
    got = fdr.tdc_qvalues(scores, target)
    np.testing.assert_array_equal(got, _tdc_loops(scores, target))
    assert (got <= 1).all() and (got > 0).all()
    o = np.argsort(-scores, kind='stable')
    assert (np.diff(got[o]) >= 0).all()         # q never improves towards worse scores


def test_qvalues_edges():
    assert fdr.tdc_qvalues([], []).shape == (0,)
    np.testing.assert_array_equal(fdr.tdc_qvalues([0.5, 0.4], [False, False]), [1.0, 1.0])
    np.testing.assert_array_equal(fdr.tdc_qvalues([0.9, 0.8, 0.7], [True, True, True]), [1 / 3] * 3)
    with pytest.raises(ValueError):
        fdr.tdc_qvalues([0.1, np.nan], [True, True])
    with pytest.raises(ValueError):
        fdr.tdc_qvalues([0.1], [True, False])


def test_groups_reference_golden():
    g = np.load(os.path.join(HERE, 'golden', 'fdr_groups_golden.npz'))
    n_groups = 0
    for c in range(int(g['n_cases'])):
        got = fdr.ssm_groups(g[f'md_{c}'], int(g[f'mgs_{c}']))
        np.testing.assert_array_equal(got, g[f'groups_{c}'], err_msg=f'case {c}')
        n_groups += len(np.unique(got))
    assert n_groups > 100                       # the fixtures do exercise real groups


class _Meta:
    def __init__(self, **cols):
        self.cols = cols

    def column(self, name, rows):
        return self.cols[name][rows]


def _table(n, rng, decoy_frac=0.3, columns=False):
    lib = dict(precursor_mz=rng.uniform(400, 900, 50), is_decoy=rng.random(50) < decoy_frac)
    qry = dict(precursor_mz=rng.uniform(400, 900, n), precursor_charge=np.full(n, 2))
    if columns:
        lmeta, qmeta = {2: _Meta(**lib)}, {2: _Meta(**qry)}
    else:
        lmeta = {2: [dict(identifier=i, peptide='P', precursor_mz=float(lib['precursor_mz'][i]),
                          **({'is_decoy': True} if lib['is_decoy'][i] else {})) for i in range(50)]}
        qmeta = {2: [dict(identifier=i, index=i, precursor_mz=float(qry['precursor_mz'][i]),
                          precursor_charge=2) for i in range(n)]}
    t = SSMTable(qmeta, lmeta)
    t.charge = np.full(n, 2, np.int32)
    t.qrow = np.arange(n, dtype=np.int64)
    t.lib_row = rng.integers(0, 50, n).astype(np.int32)
    t.score = rng.random(n)
    t.q = np.full(n, np.nan)
    t.batch = np.zeros(n, np.int32)
    t.pos = np.arange(n, dtype=np.int32)
    return t, lib, qry


@pytest.mark.parametrize('columns', [False, True])
def test_scorer_on_table(columns):
    rng = np.random.default_rng(5)
    t, lib, qry = _table(600, rng, columns=columns)
    decoy = lib['is_decoy'][t.lib_row]
    np.testing.assert_array_equal(t.is_decoy(), decoy)
    md = (qry['precursor_mz'] - lib['precursor_mz'][t.lib_row]) * 2
    np.testing.assert_array_equal(t.mass_diffs(), md)
    cos = t.score.copy()
    scorer = fdr.CosineTDC(min_group_size=20)
    assert scorer.columnar and scorer(t, 'std') is None
    want = np.where(decoy, np.nan, fdr.tdc_qvalues(cos, ~decoy))
    np.testing.assert_array_equal(t.q, want)
    np.testing.assert_array_equal(t.score, np.where(decoy, np.nan, cos))
    # open level: one competition per mass-difference group
    t.score = cos.copy()
    scorer(t, 'open')
    groups = fdr.ssm_groups(md, 20)
    assert scorer.n_groups == len(np.unique(groups))
    for gid in np.unique(groups):
        m = groups == gid
        np.testing.assert_array_equal(t.q[m], np.where(decoy[m], np.nan, fdr.tdc_qvalues(cos[m], ~decoy[m])))
