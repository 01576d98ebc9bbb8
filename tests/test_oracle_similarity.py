"""CPU checks of the oracle's SSM similarity features (oracle/asl_oracle_sim.c; reference
spectrum_similarity.py:13-730 as called by utils.py:344-456):
  * the constants the reference's own tests hold (src/tests/spectrum_similarity_test.py,
    copied as data into tests/golden/similarity_expected.json) on the reference's fixtures,
  * tests/golden/ssm_features_golden.npz = the reference module run on seeded SSMs.
The reference sums float32 arrays (NumPy pairwise order); the restatement carries doubles,
hence 1e-5 (north star tolerance for scores)."""
import json
import os

import numpy as np
import pytest

from sim_common import COLUMN, check_features, kat_case

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_test_constants(O):
    exp = json.load(open(os.path.join(HERE, 'golden', 'similarity_expected.json')))
    kat = np.load(os.path.join(HERE, 'golden', 'similarity_kat.npz'))
    checked = 0
    cache = {}
    for e in exp:
        top = e['fixture'].endswith('_top')
        name = e['fixture'][:-4] if top else e['fixture']
        col = COLUMN.get((e['method'], e['args'].replace('"', "'") if e['method'] != 'hypergeometric_score'
                          else '', top))
        if col is None:
            continue                      # variant that utils._compute_ssm_features never requests
        hg = e['method'] == 'hypergeometric_score'
        key = (name, hg)
        if key not in cache:
            q_mz, q_int, l_mz, l_int, pm = kat_case(kat, name)
            args = (101, 1500, 0.1) if hg else (11, 2010, 0.04)   # params of test_hypergeometric_score
            cache[key] = O.ssm_features(q_mz, q_int, l_mz, l_int, pm, *args)
        got = cache[key][col]
        if np.isinf(e['value']):
            assert got == e['value'], e
        else:
            assert got == pytest.approx(e['value'], rel=1e-5, abs=2e-6), e
        checked += 1
    assert checked >= 60


def test_golden_features(O):
    g = np.load(os.path.join(HERE, 'golden', 'ssm_features_golden.npz'))
    qo, lo, po = g['q_offsets'], g['l_offsets'], g['pm_offsets']
    for c in range(len(g['features'])):
        got = O.ssm_features(g['q_mz'][qo[c]:qo[c + 1]], g['q_intensity'][qo[c]:qo[c + 1]],
                             g['l_mz'][lo[c]:lo[c + 1]], g['l_intensity'][lo[c]:lo[c + 1]],
                             g['pm_pairs'][po[c]:po[c + 1]])
        check_features(got, g['features'][c], f'case {c}')


def test_degenerate_inputs(O):
    """No matches (the reference skips such SSMs, the calculator still defines the values),
    a single match, fewer library peaks than `top`."""
    mz = np.linspace(100, 1000, 12).astype(np.float32)
    it = (np.arange(12) + 1).astype(np.float32)
    it /= np.linalg.norm(it)
    f = O.ssm_features(mz, it, mz, it, np.zeros((0, 2), np.uint32))
    assert f[0] == 0 and f[2] == 0 and np.isinf(f[9]) and np.isinf(f[23]) and f[30] == 1.0
    assert f[16] == 0 and f[26] == 0 and np.isinf(f[31])
    f = O.ssm_features(mz, it, mz, it, np.array([[3, 3]], np.uint32))
    assert f[2] == 1 and f[16] == 0.0 and f[0] == pytest.approx(float(it[3]) ** 2, rel=1e-6)
    f = O.ssm_features(mz[:4], it[:4], mz[:4], it[:4], np.array([[0, 0], [1, 1]], np.uint32))
    assert f[5] == pytest.approx(0.5) and f[4] == pytest.approx(0.5)     # top >= n_library: all peaks
