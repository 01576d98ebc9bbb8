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
    cfg = Config(num_list=16, num_probe=6, num_candidates=256, index='ivfpq', kmeans_niter=5,
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
        cfg = Config(**{**sl.config.__dict__, 'index': 'ivfflat'})
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
    bf = SpectralLibrary(lib, config=Config(**{**sl.config.__dict__, 'mode': 'bf'}))
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
    small = SpectralLibrary(lib, config=Config(**{**sl.config.__dict__, 'num_list': 1000}))
    assert 4 not in small._ann_filenames and 2 in small._ann_filenames
    q4, _ = synthetic.make_queries(lib, aux, 32, seed=71, charge=4)
    r4 = small._search_batch(q4, 4, 'open')
    assert (r4.best_row >= 0).any()
    assert small._search_batch(q4, 7, 'open') is None      # charge absent from the library


def test_index_cache_files(tmp_path, world):
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    import os
    lib, aux, sl = world
    cfg = Config(**sl.config.__dict__)
    a = SpectralLibrary(lib, config=cfg, index_dir=str(tmp_path), basename='lib')
    h7 = a._get_hyperparameter_hash()[:7]
    files = sorted(os.listdir(tmp_path))
    assert files == [f'lib_{h7}_{z}.idxann' for z in (2, 3, 4)]
    for p in a.partitions.values():
        p.index = None                     # force a reload from the cache files
    from ann_solo_amd import synthetic
    q, _ = synthetic.make_queries(lib, aux, 64, seed=80, charge=3)
    r1 = a._search_batch(q, 3, 'open')
    r0 = sl._search_batch(q, 3, 'open')
    assert np.array_equal(r1.best_row, r0.best_row)
