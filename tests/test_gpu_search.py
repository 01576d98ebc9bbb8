"""GPU parity of the whole hot path (asl_search_batch through the SpectralLibrary mirror)
against the oracle's orc_search_batch on the same seeded inputs: identical ANN id sets,
identical winning library rows, identical peak matches, identical double scores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def world(O):
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(6000, seed=41, device='cpu')
    cfg = Config.open_search(num_list=16, num_probe=6, num_candidates=256, index='ivfpq', kmeans_niter=5,
                 precursor_tolerance_mass=20, precursor_tolerance_mode='ppm',
                 precursor_tolerance_mass_open=300, precursor_tolerance_mode_open='Da')
    sl = SpectralLibrary(lib, config=cfg)
    return lib, aux, sl


def _oracle_partition(O, sl, z):
    part = sl.partitions[z]
    L = O.Spectra(*part.spectra.to('cpu').numpy())
    idx = part.index
    cen = idx.centroids()
    off, ids, payload = idx.lists()
    info = idx.info()
    cb = idx.codebooks() if info.kind == 2 else None
    ivf = O.HostIVF.__new__(O.HostIVF)
    ivf.centroids, ivf.nlist, ivf.d = cen, info.nlist, info.d
    ivf.list_offsets, ivf.ids, ivf.payload, ivf.codebooks = off, ids, payload, cb
    ivf.kind = 1 if cb is not None else 0
    return L, part.precursor_mz, ivf


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_open_search_batch_matches_oracle(O, world, index):
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux, sl = world
    if index == 'ivfflat':
        cfg = Config.open_search(**{**sl.config.__dict__, 'index': 'ivfflat'})
        sl = SpectralLibrary(lib, config=cfg)
    for z in (2, 3):
        q, truth = synthetic.make_queries(lib, aux, 300, seed=50 + z, charge=z)
        res = sl._search_batch(q, z, 'open', want_knn=True)
        L, pmz32, ivf = _oracle_partition(O, sl, z)
        Q = O.Spectra(*q.numpy())
        ref = O.search_batch(Q, L, pmz32, z, ivf, 256, 6, 300, 'Da', 0.02, True,
                             pm_stride=res.pm_pairs.shape[1], want_knn=True)
        assert np.array_equal(res.knn, ref['knn_I'])                  # identical candidate ids
        assert np.array_equal(res.n_candidates, ref['n_cand'])
        assert np.array_equal(res.best_row, ref['best_row'])
        assert np.array_equal(res.best_score, ref['best_score'])      # same doubles
        assert np.array_equal(res.pm_count, ref['pm_count'])
        for i in range(q.n):
            n = res.pm_count[i]
            assert np.array_equal(res.pm_pairs[i, :n], ref['pm_pairs'][i, :n])
        # sanity: unmodified queries find their source spectrum
        part_rows = np.nonzero(lib.precursor_charge.numpy() == z)[0]
        src = truth['source_row'].numpy()
        unmod = ~truth['is_modified'].numpy()
        hit = part_rows[np.clip(res.best_row, 0, None)] == src
        assert hit[unmod].mean() > 0.9


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_open_search_with_more_than_2048_candidates(O, world, index):
    """num_candidates = 3 000 through the fused batch call (the index searches in bounded passes, the
    rescoring walks 3 000-wide rows), pipelined and not: neighbours, winners, scores and peak matches
    equal the oracle's."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux, sl0 = world
    cfg = Config.open_search(**{**sl0.config.__dict__, 'index': index, 'num_candidates': 3000, 'num_probe': 12})
    sl = SpectralLibrary(lib, config=cfg)
    z = 2
    q, truth = synthetic.make_queries(lib, aux, 150, seed=91, charge=z)
    res = sl._search_batch(q, z, 'open', want_knn=True)
    L, pmz32, ivf = _oracle_partition(O, sl, z)
    ref = O.search_batch(O.Spectra(*q.numpy()), L, pmz32, z, ivf, 3000, 12, 300, 'Da', 0.02, True,
                         pm_stride=res.pm_pairs.shape[1], want_knn=True)
    assert np.array_equal(res.knn, ref['knn_I'])
    assert np.array_equal(res.n_candidates, ref['n_cand'])
    assert np.array_equal(res.best_row, ref['best_row'])
    assert np.array_equal(res.best_score, ref['best_score'])
    assert np.array_equal(res.pm_count, ref['pm_count'])
    sl.set_pipeline(True)
    a = sl._search_batch(q.to('cuda:0'), z, 'open', device_out=True)
    b = sl._search_batch(q.to('cuda:0'), z, 'open', device_out=True)
    sl.synchronize()
    sl.set_pipeline(False)
    for r in (a, b):
        assert np.array_equal(r.best_row.cpu().numpy(), ref['best_row'])
        assert np.array_equal(r.best_score.cpu().numpy(), ref['best_score'])
    sl.shutdown()


def test_std_and_bruteforce_modes_match_oracle(O, world):
    """Cascade level 'std' (20 ppm window, no ANN) and --mode bf open search: candidates =
    the whole precursor window (spectral_library.py:417-429)."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux, sl = world
    z = 2
    q, truth = synthetic.make_queries(lib, aux, 200, seed=60, charge=z)
    L, pmz32, _ = _oracle_partition(O, sl, z)
    Q = O.Spectra(*q.numpy())
    bf = SpectralLibrary(lib, config=Config.open_search(**{**sl.config.__dict__, 'mode': 'bf'}))
    for engine, mode, tol, tmode in ((sl, 'std', 20, 'ppm'), (bf, 'open', 300, 'Da')):
        res = engine._search_batch(q, z, mode)
        lists = engine._get_library_candidates(q, z, mode)
        for i in range(q.n):
            want = np.array([r for r in range(L.n)
                             if O.precursor_ok(Q.precursor_mz[i], pmz32[r], z, tol, tmode)])
            assert np.array_equal(lists[i], want)
            assert res.n_candidates[i] == len(want)
            b, s, m = O.best_match(Q, i, L, want, 0.02, True)
            if b < 0:
                assert res.best_row[i] == -1
                continue
            assert res.best_row[i] == want[b] and res.best_score[i] == s
            assert np.array_equal(res.peak_matches(i), m)


def test_device_resident_io_and_small_charge_fallback(O, world):
    """Queries and outputs as device tensors; a charge with fewer than num_list spectra
    has no ANN index and falls back to the window search (spectral_library.py:102-104)."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux, sl = world
    z = 2
    q, _ = synthetic.make_queries(lib, aux, 128, seed=70, charge=z)
    a = sl._search_batch(q, z, 'open')
    b = sl._search_batch(q.to('cuda'), z, 'open', device_out=True)
    torch.cuda.synchronize()
    assert np.array_equal(a.best_row, b.best_row.cpu().numpy())
    assert np.array_equal(a.best_score, b.best_score.cpu().numpy())
    small = SpectralLibrary(lib, config=Config.open_search(**{**sl.config.__dict__, 'num_list': 1000}))
    assert 4 not in small._ann_filenames and 2 in small._ann_filenames
    q4, _ = synthetic.make_queries(lib, aux, 32, seed=71, charge=4)
    r4 = small._search_batch(q4, 4, 'open')
    assert (r4.best_row >= 0).any()
    assert small._search_batch(q4, 7, 'open') is None      # charge absent from the library


def test_index_cache_files(tmp_path, world):
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    import os
    lib, aux, sl = world
    cfg = Config.open_search(**sl.config.__dict__)
    a = SpectralLibrary(lib, config=cfg, index_dir=str(tmp_path), basename='lib')
    h7 = a._get_index_hash()[:7]        # (== the reference's five-key hash for its own IVF-Flat setup)
    assert (cfg.index == 'ivfflat') == (a._get_index_hash() == a._get_hyperparameter_hash())
    files = sorted(os.listdir(tmp_path))
    assert files == [f'lib_{h7}_{z}.idxmi' for z in (2, 3, 4)]
    for p in a.partitions.values():
        p.index = None                     # force a reload from the cache files
    from ann_solo_amd import synthetic
    q, _ = synthetic.make_queries(lib, aux, 64, seed=80, charge=3)
    r1 = a._search_batch(q, 3, 'open')
    r0 = sl._search_batch(q, 3, 'open')
    assert np.array_equal(r1.best_row, r0.best_row)


def test_search_driver_end_to_end_to_mztab(tmp_path, monkeypatch):
    """SpectralLibrary.search (cascade std -> open) over packed queries: unmodified queries are
    identified by the standard search, modified ones only by the open search; the columns the
    reference's mzTab writer prints (tests/mztab_check.py) carry the device results."""
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    import mztab_check as M
    lib, aux = synthetic.make_library(4000, seed=71, device='cpu', charges=(2,), charge_p=(1.0,))
    q, truth = synthetic.make_queries(lib, aux, 300, seed=72, charge=2, open_range=300.0)
    cfg = Config.open_search(num_list=32, num_probe=32, num_candidates=1024, index='ivfpq', kmeans_niter=5,
                 batch_size=128, query_filename='/data/q.mgf', spectral_library_filename='/data/l.splib')
    sl = SpectralLibrary(lib, config=cfg)
    qmeta = {2: [dict(identifier=f'scan={i}', index=i, retention_time=0.5 * i, precursor_charge=2,
                      precursor_mz=float(q.precursor_mz[i])) for i in range(q.n)]}
    lmeta = {2: [dict(identifier=int(r), peptide=f'PEPTIDE{r}K', precursor_mz=float(p), is_decoy=False)
                 for r, p in enumerate(sl.partitions[2].precursor_mz)]}
    seen = []

    def scorer(ssms, mode):          # stands for utils.score_ssms: accept confident matches only
        seen.append((mode, len(ssms)))
        for s in ssms:
            s.q = 0.001 if s.search_engine_score > 0.5 else 0.5
        return ssms
    ids = sl.search({2: q}, qmeta, lmeta, score_ssms=scorer)
    by = {s.query_identifier: s for s in ids}
    src, mod = truth['source_row'].numpy(), truth['is_modified'].numpy()
    std = sl._search_batch(q, 2, 'std')
    opn = sl._search_batch(q, 2, 'open')
    assert seen[0][0] == 'std' and seen[1][0] == 'open' and seen[1][1] < q.n
    n_std = n_open = 0
    for i in range(q.n):
        s = by.get(f'scan={i}')
        if s is None:
            assert opn.best_row[i] < 0
            continue
        if std.best_row[i] >= 0 and s.library_identifier == std.best_row[i] and not mod[i]:
            n_std += 1                                           # kept from level 1
        else:
            assert s.library_identifier == opn.best_row[i]       # level 2 result
            n_open += 1
    assert n_std > 50 and n_open > 50
    rows = [M.record_fields(s_) for s_ in sorted(ids, key=lambda s_: M.natural_key(s_.query_identifier))]
    assert len(rows) == len(ids)
    nums = [int(r['PSM_ID'].split('=')[1]) for r in rows]
    assert nums == sorted(nums)                                  # natural order of the identifiers
    for r in rows[:20]:
        s = by[r['PSM_ID']]
        assert r['sequence'] == s.sequence == f'PEPTIDE{s.library_identifier}K'
        assert float(r['search_engine_score[1]']) == s.search_engine_score
        assert r['opt_ms_run[1]_cv_MS:1003062_spectrum_index'] == str(s.library_identifier)
        assert r['spectra_ref'] == f'ms_run[1]:index={s.query_index}' and r['charge'] == '2'
    sl.shutdown()


def test_cascade_with_the_cosine_tdc_gate():
    """An injected columnar scorer (tests/fdr_gate.py: the reference's ``--model none`` gate,
    target-decoy q-values on the cosine, per mass-difference group at the open level) drives both
    cascade levels. Checked against the gate applied by hand to the same levels' SSMs."""
    import torch
    import fdr_gate as fdr
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(6000, seed=81, device='cpu', charges=(2,), charge_p=(1.0,))
    q, truth = synthetic.make_queries(lib, aux, 900, seed=82, charge=2, open_range=300.0)
    cfg = Config.open_search(num_list=32, num_probe=32, num_candidates=1024, index='ivfpq', kmeans_niter=5,
                 batch_size=256, model='none', fdr=0.05, fdr_min_group_size=10)
    sl = SpectralLibrary(lib, config=cfg, score_ssms=fdr.CosineTDC(cfg.fdr_min_group_size))
    rng = np.random.default_rng(7)
    decoy = rng.random(lib.n) < 0.5                    # half the library flagged as decoys,
    decoy[truth['source_row'].numpy()] = False         # none of them a query's true source
    qmeta = {2: [dict(identifier=f'scan={i}', index=i, retention_time=0.0, precursor_charge=2,
                      precursor_mz=float(q.precursor_mz[i])) for i in range(q.n)]}
    lmeta = {2: [dict(identifier=int(r), peptide=f'PEPTIDE{r}K', precursor_mz=float(p), is_decoy=bool(decoy[r]))
                 for r, p in enumerate(sl.partitions[2].precursor_mz)]}
    ids = sl.search({2: q}, qmeta, lmeta)
    # by hand: level 1 ungated, then the gate; level 2 on the rest
    def ungated(table, mode):
        return None
    ungated.columnar = True
    everything = {2: np.arange(q.n)}
    t1 = sl._search_cascade({2: q}, qmeta, lmeta, everything, 'std', score_ssms=ungated)
    d1 = decoy[t1.lib_row]
    q1 = np.where(d1, np.nan, fdr.tdc_qvalues(t1.score, ~d1))
    keep1 = q1 < cfg.fdr
    assert 50 < keep1.sum() < len(t1)
    rest = np.setdiff1d(np.arange(q.n), t1.qrow[keep1])
    t2 = sl._search_cascade({2: q}, qmeta, lmeta, {2: rest}, 'open', score_ssms=ungated)
    d2 = decoy[t2.lib_row]
    md = (np.asarray(q.precursor_mz)[t2.qrow] - sl.partitions[2].precursor_mz[t2.lib_row].astype(np.float64)) * 2
    g2 = fdr.ssm_groups(md, cfg.fdr_min_group_size)
    assert len(np.unique(g2)) >= 2
    q2 = np.full(len(t2), np.nan)
    for g in np.unique(g2):
        m = g2 == g
        q2[m] = np.where(d2[m], np.nan, fdr.tdc_qvalues(t2.score[m], ~d2[m]))
    np.testing.assert_array_equal(ids.qrow, np.concatenate([t1.qrow[keep1], t2.qrow]))
    np.testing.assert_array_equal(ids.lib_row, np.concatenate([t1.lib_row[keep1], t2.lib_row]))
    np.testing.assert_array_equal(ids.q, np.concatenate([q1[keep1], q2]))
    assert np.isnan(ids.score[decoy[ids.lib_row]]).all() and not np.isnan(ids.score[~decoy[ids.lib_row]]).any()
    rec = ids[0]
    assert rec.q == ids.q[0] and rec.is_decoy == bool(decoy[ids.lib_row[0]])
    sl.shutdown()


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_window_filter_inside_the_scan_equals_the_filter_inside_the_rescoring(O, index):
    """`asl_set_scan_postfilter`: the precursor window applied in the list scan's finish (rows of passing hits
    + their lengths) against the window applied in the rescoring's compaction (the only place until round 6) --
    winners, scores, candidate counts and peak matches must be identical, and equal the oracle's, for Da and
    ppm windows, a library with 400 copies of one spectrum (mass ties at the k-th score: those rows leave the
    scan unfiltered, length -1), windows that pass nothing and windows that pass everything; pipelined too."""
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd.packed import PackedSpectra
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib0, aux = synthetic.make_library(5000, seed=71, device='cpu', charges=(2,), charge_p=(1.0,))
    o, mz, it, chg, pmz, pz = lib0.numpy()
    # rows 0 .. 399 become copies of row 0 with precursors spread over +-300 Da
    a, b = int(o[0]), int(o[1])
    n0 = b - a
    rng = np.random.default_rng(5)
    offs = [0]
    MZ, IT, CH, PM = [], [], [], []
    for r in range(lib0.n):
        s, e = (a, b) if r < 400 else (int(o[r]), int(o[r + 1]))
        MZ.append(mz[s:e]); IT.append(it[s:e]); CH.append(chg[s:e])
        PM.append(pmz[0] + rng.uniform(-300, 300) if r < 400 else pmz[r])
        offs.append(offs[-1] + (e - s))
    lib = PackedSpectra.from_numpy(np.asarray(offs, np.int32), np.concatenate(MZ), np.concatenate(IT),
                                   np.concatenate(CH), np.asarray(PM), pz)
    q, _ = synthetic.make_queries(lib0, aux, 200, seed=72, charge=2)
    qo, qmz, qit, qchg, qpmz, qpz = q.numpy()
    q = PackedSpectra.from_numpy(qo, qmz, qit, qchg, qpmz, qpz)
    L = _lib.lib()
    for tol, mode in ((250.0, 'Da'), (0.01, 'Da'), (1e9, 'Da'), (2e5, 'ppm'), (10.0, 'ppm')):
        cfg = Config.open_search(num_list=16, num_probe=8, num_candidates=1024, index=index, kmeans_niter=4,
                                 precursor_tolerance_mass_open=tol, precursor_tolerance_mode_open=mode)
        sl = SpectralLibrary(lib, config=cfg)
        res = {}
        try:
            for on in (0, 1):
                L.asl_set_scan_postfilter(on)
                res[on] = sl._search_batch(q, 2, 'open')
                sl.set_pipeline(True)
                a_ = sl._search_batch(q.to('cuda:0'), 2, 'open', device_out=True)
                b_ = sl._search_batch(q.to('cuda:0'), 2, 'open', device_out=True)
                sl.synchronize()
                sl.set_pipeline(False)
                for r in (a_, b_):
                    assert np.array_equal(r.best_row.cpu().numpy(), res[on].best_row)
                    assert np.array_equal(r.best_score.cpu().numpy(), res[on].best_score)
                    assert np.array_equal(r.n_candidates.cpu().numpy(), res[on].n_candidates)
        finally:
            L.asl_set_scan_postfilter(1)
        for f in ('best_row', 'best_score', 'n_candidates', 'pm_count'):
            assert np.array_equal(getattr(res[0], f), getattr(res[1], f)), (tol, mode, f)
        assert np.array_equal(res[0].pm_pairs, res[1].pm_pairs)
        Lo, pmz32, ivf = _oracle_partition(O, sl, 2)
        ref = O.search_batch(O.Spectra(*q.numpy()), Lo, pmz32, 2, ivf, 1024, 8, tol, mode, 0.02, True,
                             pm_stride=res[1].pm_pairs.shape[1])
        assert np.array_equal(res[1].best_row, ref['best_row']) and np.array_equal(res[1].best_score, ref['best_score'])
        assert np.array_equal(res[1].n_candidates, ref['n_cand'])
        if tol == 0.01:
            assert res[1].n_candidates.max() <= 4 and (res[1].n_candidates == 0).any()
        if tol == 1e9:
            assert res[1].n_candidates.max() == 1024
        sl.shutdown()
