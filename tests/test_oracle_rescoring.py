"""Oracle vs the reference's own SpectrumMatcher::dot (tests/golden/rescoring_golden.npz,
made by oracle/_ref = /root/reference/src/ann_solo/SpectrumMatch.cpp compiled in place)
and vs the constants of the reference's own test-suite."""
import numpy as np
import pytest


def _spectra(O, g):
    L = O.Spectra(g['lib_offsets'], g['lib_mz'], g['lib_intensity'], g['lib_charge'],
                  g['lib_pmz'], g['lib_pcharge'])
    Q = O.Spectra(g['q_offsets'], g['q_mz'], g['q_intensity'], None, g['q_pmz'], g['q_pcharge'])
    return Q, L


def _products(Q, qi, L, row, pairs):
    qm, qi_, _ = Q.peaks(qi)
    lm, li, _ = L.peaks(row)
    return [float(np.float32(qi_[a]) * np.float32(li[b])) for a, b in pairs]


def test_known_answer_from_reference_tests(O, golden):
    """src/tests/spectrum_similarity_test.py:339-341,443: partial_match pair."""
    k = golden('similarity_kat.npz')
    for name, n_match in (('all_match', 12), ('no_match', 0), ('partial_match', 8)):
        q = (k[f'{name}_q_mz'], k[f'{name}_q_intensity'])
        l = (k[f'{name}_l_mz'], k[f'{name}_l_intensity'])
        for shift in (False, True):
            s, m = O.dot_pair(q[0], q[1], float(k[f'{name}_q_pmz']), l[0], l[1],
                              np.zeros(len(l[0]), np.uint8), float(k[f'{name}_l_pmz']),
                              int(k[f'{name}_l_charge']), 0.02, shift)
            want = k[f'{name}_peak_matches']
            if name == 'all_match':
                # the fixture drops its two zero-intensity peaks from the hand-written list
                assert len(m) >= n_match
            else:
                assert sorted(map(tuple, m.tolist())) == sorted(map(tuple, want.tolist()))
            # cosine() (spectrum_similarity.py:95-106) over the returned matches
            cos = float(np.dot(q[1][m[:, 0]], l[1][m[:, 1]])) if len(m) else 0.0
            assert cos == pytest.approx(float(k[f'{name}_cosine']), abs=1e-6)
    assert float(k['partial_match_cosine']) == pytest.approx(0.44582117, abs=1e-7)


def test_oracle_equals_reference_dot(O, golden):
    g = golden('rescoring_golden.npz')
    Q, L = _spectra(O, g)
    n_order_diff = 0
    for c in range(len(g['case_query'])):
        qi = int(g['case_query'][c])
        cand = g['cand_rows'][g['cand_offsets'][c]:g['cand_offsets'][c + 1]]
        b, s, m = O.best_match(Q, qi, L, cand, float(g['case_tol'][c]), bool(g['case_shift'][c]))
        assert b == int(g['case_best'][c])
        assert s == pytest.approx(float(g['case_score'][c]), abs=1e-12)
        want = g['pm_pairs'][g['pm_offsets'][c]:g['pm_offsets'][c + 1]]
        assert sorted(map(tuple, m.tolist())) == sorted(map(tuple, want.tolist()))
        if m.tolist() != want.tolist():
            # std::sort is unstable (SpectrumMatch.cpp:92): order may differ only
            # inside runs of equal products
            n_order_diff += 1
            row = int(cand[b])
            pa = _products(Q, qi, L, row, m.tolist())
            pb = _products(Q, qi, L, row, want.tolist())
            assert np.allclose(pa, pb, rtol=0, atol=1e-7) or sorted(pa) == sorted(pb)
    assert n_order_diff < len(g['case_query'])


@pytest.mark.skipif(not __import__('os').path.exists(
    __import__('os').path.join(__import__('os').path.dirname(__file__), '..', 'oracle', '_ref',
                               'libref_spectrummatch.so')), reason='oracle/_ref not built')
def test_oracle_equals_live_reference_build(O):
    """Fresh random cases straight against oracle/_ref (when it is present)."""
    from ann_solo_amd import synthetic
    lib, aux = synthetic.make_library(500, seed=3, device='cpu')
    q, truth = synthetic.make_queries(lib, aux, 64, seed=4)
    L, Q = O.Spectra(*lib.numpy()), O.Spectra(*q.numpy())
    rng = np.random.default_rng(1)
    for qi in range(Q.n):
        cand = np.unique(np.concatenate([rng.integers(0, 500, 30),
                                         [int(truth['source_row'][qi])]]))
        for shift in (True, False):
            b1, s1, m1 = O.best_match(Q, qi, L, cand, 0.02, shift)
            b2, s2, m2 = O.ref_best_match(Q, qi, L, cand, 0.02, shift)
            assert b1 == b2 and abs(s1 - s2) < 1e-12
            assert sorted(map(tuple, m1.tolist())) == sorted(map(tuple, m2.tolist()))


def test_first_candidate_wins_ties(O, golden):
    g = golden('rescoring_golden.npz')
    Q, L = _spectra(O, g)
    n0 = 600                         # rows 600,601 duplicate row 0; 602 duplicates row 1
    b, s, _ = O.best_match(Q, 0, L, np.array([n0, 0, n0 + 1]), 0.02, True)
    assert b == 0
    b2, s2, _ = O.best_match(Q, 0, L, np.array([0, n0, n0 + 1]), 0.02, True)
    assert b2 == 0 and s2 == s


def test_empty_candidates(O, golden):
    g = golden('rescoring_golden.npz')
    Q, L = _spectra(O, g)
    b, s, m = O.best_match(Q, 0, L, np.zeros(0, np.int64), 0.02, True)
    assert b == -1 and len(m) == 0
