"""GPU parity of the batched SSM similarity features (asl_ssm_features_batch) against the
oracle restatement, the goldens generated from the reference module, and the constants of
the reference's own tests (tests/golden/similarity_expected.json)."""
import json
import os

import numpy as np
import pytest

from sim_common import COLUMN, check_features, kat_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _pack(offsets, mz, inten):
    from ann_solo_amd.packed import PackedSpectra
    n = len(offsets) - 1
    return PackedSpectra.from_numpy(offsets, mz, inten, None, np.full(n, 500.0), np.full(n, 2))


def test_features_match_goldens_and_oracle(O):
    from ann_solo_amd import spectrum_similarity as sim
    g = np.load(os.path.join(HERE, 'golden', 'ssm_features_golden.npz'))
    qo, lo, po = g['q_offsets'], g['l_offsets'], g['pm_offsets']
    n = len(qo) - 1
    cnt = np.diff(po).astype(np.int32)
    pairs = np.zeros((n, max(1, cnt.max()), 2), np.uint32)
    for c in range(n):
        pairs[c, :cnt[c]] = g['pm_pairs'][po[c]:po[c + 1]]
    Q, L = _pack(qo, g['q_mz'], g['q_intensity']), _pack(lo, g['l_mz'], g['l_intensity'])
    F = sim.ssm_features(Q, L, np.arange(n, dtype=np.int32), pairs, cnt)
    assert F.shape == (n, sim.N_FEATURES)
    for c in range(n):
        check_features(F[c], g['features'][c], f'golden {c}')
        want = O.ssm_features(g['q_mz'][qo[c]:qo[c + 1]], g['q_intensity'][qo[c]:qo[c + 1]],
                              g['l_mz'][lo[c]:lo[c + 1]], g['l_intensity'][lo[c]:lo[c + 1]],
                              g['pm_pairs'][po[c]:po[c + 1]])
        check_features(F[c], want, f'oracle {c}', rel=1e-9, abs_=1e-11)
    # device-resident inputs and outputs give the same bits
    import torch
    Fd = sim.ssm_features(Q.to('cuda'), L.to('cuda'), torch.arange(n, device='cuda'),
                          torch.from_numpy(pairs.view(np.int32)).cuda(), torch.from_numpy(cnt).cuda())
    assert np.array_equal(Fd.cpu().numpy().view(np.uint64), F.view(np.uint64))


def test_cosine_kernel_has_the_bits_of_feature_column_0():
    """asl_ssm_cosine_batch (the cascade's default score, spectrum_similarity.py:81-106) == column
    0 of the feature kernel bit for bit, on the goldens and on a searched batch, host and device
    arrays; rows without a match are NaN; the reference test constant 0.44582117 comes out."""
    import torch
    from ann_solo_amd import spectrum_similarity as sim
    g = np.load(os.path.join(HERE, 'golden', 'ssm_features_golden.npz'))
    qo, lo, po = g['q_offsets'], g['l_offsets'], g['pm_offsets']
    n = len(qo) - 1
    cnt = np.diff(po).astype(np.int32)
    pairs = np.zeros((n, max(1, cnt.max()), 2), np.uint32)
    for c in range(n):
        pairs[c, :cnt[c]] = g['pm_pairs'][po[c]:po[c + 1]]
    Q, L = _pack(qo, g['q_mz'], g['q_intensity']), _pack(lo, g['l_mz'], g['l_intensity'])
    rows = np.arange(n, dtype=np.int32)
    rows[::7] = -1
    F = sim.ssm_features(Q, L, rows, pairs, cnt)
    c0 = sim.ssm_cosine(Q, L, rows, pairs, cnt)
    assert np.array_equal(c0.view(np.uint64), F[:, 0].view(np.uint64))
    assert np.isnan(c0[::7]).all() and np.isfinite(c0[rows >= 0]).all()
    cd = sim.ssm_cosine(Q.to('cuda'), L.to('cuda'), torch.from_numpy(rows).cuda(),
                        torch.from_numpy(pairs.view(np.int32)).cuda(), torch.from_numpy(cnt).cuda())
    assert np.array_equal(cd.cpu().numpy().view(np.uint64), c0.view(np.uint64))
    kat = np.load(os.path.join(HERE, 'golden', 'similarity_kat.npz'))
    q_mz, q_int, l_mz, l_int, pm = kat_case(kat, 'partial_match')
    one = sim.ssm_cosine(_pack([0, len(q_mz)], q_mz, q_int), _pack([0, len(l_mz)], l_mz, l_int),
                         np.zeros(1, np.int32), np.asarray(pm, np.uint32)[None], np.array([len(pm)], np.int32))
    assert abs(one[0] - 0.44582117) < 1e-7
    with pytest.raises(Exception):
        bad = pairs.copy()
        c = int(np.nonzero(cnt > 0)[0][0])       # a pair that is actually read
        bad[c, 0, 0] = 10 ** 6
        sim.ssm_cosine(Q, L, np.arange(n, dtype=np.int32), bad, cnt)


def test_reference_test_constants_through_the_calculator_mirror():
    from ann_solo_amd.spectrum_similarity import SpectrumSimilarityCalculator
    exp = json.load(open(os.path.join(HERE, 'golden', 'similarity_expected.json')))
    kat = np.load(os.path.join(HERE, 'golden', 'similarity_kat.npz'))

    class Spec:
        def __init__(self, mz, inten):
            self.mz, self.intensity, self.precursor_mz, self.precursor_charge = mz, inten, 500.0, 2

    class SSM:
        pass
    calc = {}
    for name in ('all_match', 'no_match', 'partial_match'):
        q_mz, q_int, l_mz, l_int, pm = kat_case(kat, name)
        ssm = SSM()
        ssm.query_spectrum, ssm.library_spectrum, ssm.peak_matches = Spec(q_mz, q_int), Spec(l_mz, l_int), pm
        calc[name] = SpectrumSimilarityCalculator(ssm)
        calc[name + '_top'] = SpectrumSimilarityCalculator(ssm, 5)
    checked = 0
    for e in exp:
        top = e['fixture'].endswith('_top')
        if COLUMN.get((e['method'], e['args'].replace('"', "'") if e['method'] != 'hypergeometric_score'
                       else '', top)) is None:
            continue
        c = calc[e['fixture']]
        if e['method'] == 'hypergeometric_score':
            got = c.hypergeometric_score(min_mz=101, max_mz=1500, fragment_mz_tol=0.1)
        elif e['method'] == 'mean_squared_error':
            got = c.mean_squared_error(e['args'].strip('"\''))
        elif e['method'] == 'entropy':
            got = c.entropy('True' in e['args'])
        else:
            got = getattr(c, e['method'])()
        if np.isinf(e['value']):
            assert got == e['value'], e
        else:
            assert got == pytest.approx(e['value'], rel=1e-5, abs=2e-6), e
        checked += 1
    assert checked >= 60
    with pytest.raises(NotImplementedError):
        calc['all_match_top'].manhattan()
    with pytest.raises(ValueError):
        calc['all_match'].mean_squared_error('unknown')


def test_features_of_a_searched_batch(O):
    """End to end: search a batch, then the features of every best match equal the oracle's on
    the same (query, library spectrum, peak matches); the cosine column is the reported cosine
    (spectrum_similarity.py:81-106)."""
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    from ann_solo_amd.spectrum_similarity import compute_ssm_features, FEATURE_NAMES
    lib, aux = synthetic.make_library(3000, seed=31, device='cpu', charges=(2,), charge_p=(1.0,))
    cfg = Config.open_search(mode='bf', precursor_tolerance_mass_open=300, precursor_tolerance_mode_open='Da')
    sl = SpectralLibrary(lib, config=cfg)
    q, _ = synthetic.make_queries(lib, aux, 200, seed=32, charge=2)
    res = sl._search_batch(q, 2, 'open')
    part = sl.partitions[2].spectra
    feats = compute_ssm_features(q.to('cuda'), part, res, cfg)
    keep = feats['index']
    assert len(keep) > 100 and (res.pm_count[keep] > 0).all()
    qo, qmz, qit, *_ = q.numpy()
    lo, lmz, lit, *_ = part.numpy()
    for j, i in enumerate(keep[:60]):
        r = res.best_row[i]
        pm = res.peak_matches(i)
        want = O.ssm_features(qmz[qo[i]:qo[i + 1]], qit[qo[i]:qo[i + 1]], lmz[lo[r]:lo[r + 1]],
                              lit[lo[r]:lo[r + 1]], pm, cfg.min_mz, cfg.max_mz, cfg.bin_size)
        got = np.array([feats[nm][j] for nm in FEATURE_NAMES], np.float64)
        check_features(got, want, f'query {i}', rel=1e-9, abs_=1e-11)
        cos = float((qit[qo[i]:qo[i + 1]][pm[:, 0]].astype(np.float64) *
                     lit[lo[r]:lo[r + 1]][pm[:, 1]].astype(np.float64)).sum())
        assert feats['cosine'][j] == pytest.approx(cos, rel=1e-12)
    assert feats['n_matched_peaks'].dtype == np.int64
    assert np.allclose(feats['mz_diff_da'], feats['query_prec_mz'] - feats['lib_prec_mz'])
    assert ((feats['precursor_charge_2'] == 1).all())


def test_invalid_peak_match_index_is_an_error():
    from ann_solo_amd import _lib
    from ann_solo_amd import spectrum_similarity as sim
    mz = np.linspace(100, 900, 12).astype(np.float32)
    it = np.full(12, 12 ** -0.5, np.float32)
    P = _pack(np.array([0, 12]), mz, it)
    pairs = np.array([[[0, 40]]], np.uint32)
    with pytest.raises(_lib.AnnSoloMiError):
        sim.ssm_features(P, P, np.zeros(1, np.int32), pairs, np.ones(1, np.int32))
    # no library row -> NaN row, not an error
    F = sim.ssm_features(P, P, np.full(1, -1, np.int32), pairs, np.ones(1, np.int32))
    assert np.isnan(F).all()


def test_long_spectra_take_the_256_peak_instantiation(O):
    """A batch with a spectrum of more than 128 peaks is redone by the wider kernel; more than
    256 peaks is a capacity error."""
    from ann_solo_amd import _lib
    from ann_solo_amd import spectrum_similarity as sim
    rng = np.random.default_rng(3)
    sizes = [40, 200, 130, 12]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    mz = np.concatenate([np.sort(rng.uniform(100, 1900, n)) for n in sizes]).astype(np.float32)
    it = np.concatenate([(lambda v: v / np.linalg.norm(v))(rng.lognormal(0, 1, n)) for n in sizes]).astype(np.float32)
    P = _pack(offs, mz, it)
    pairs = np.zeros((4, 12, 2), np.uint32)
    cnt = np.array([10, 12, 5, 3], np.int32)
    for i, n in enumerate(sizes):
        sel = np.sort(rng.choice(n, cnt[i], replace=False))
        pairs[i, :cnt[i], 0] = sel
        pairs[i, :cnt[i], 1] = sel
    F = sim.ssm_features(P, P, np.arange(4, dtype=np.int32), pairs, cnt)
    for i in range(4):
        sl = slice(offs[i], offs[i + 1])
        want = O.ssm_features(mz[sl], it[sl], mz[sl], it[sl], pairs[i, :cnt[i]])
        check_features(F[i], want, f'ssm {i}', rel=1e-9, abs_=1e-11)
    big = _pack(np.array([0, 300]), np.sort(rng.uniform(100, 1900, 300)).astype(np.float32),
                np.full(300, 300 ** -0.5, np.float32))
    with pytest.raises(_lib.AnnSoloMiError, match='peaks'):
        sim.ssm_features(big, big, np.zeros(1, np.int32), pairs[:1], cnt[:1])
