"""Block-major exact IVF-Flat (scan variant 3, flat_bm_scan.hip) against the query-major postings
scan (variant 0, itself pinned to the oracle in test_gpu_index.py): ids AND scores bit for bit,
including the cases that take the fall-back or leave rows short."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def world():
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(300000, seed=41, device=dev, charges=(2,), charge_p=(1.0,))
    cfg = Config(num_list=1024, num_probe=64, num_candidates=1024, index='ivfflat', kmeans_niter=4,
                 precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da')
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    q, _ = synthetic.make_queries(lib, aux, 4096, seed=42, open_range=500.0, charge=2)
    yield sl, q
    sl.shutdown()


def _both(idx, vec, k):
    idx.set_scan_variant(0)
    D0, I0 = idx.search(vec, k)
    idx.set_scan_variant(3)
    D3, I3 = idx.search(vec, k)
    idx.set_scan_variant(0)
    return D0, I0, D3, I3


@pytest.mark.parametrize('nprobe,k', [(64, 1024), (128, 1024), (8, 300), (16, 2048), (1, 50), (200, 1)])
def test_block_major_equals_query_major(world, nprobe, k):
    sl, q = world
    idx = sl._get_ann_index(2)
    vec = sl._encode(q)
    idx.nprobe = nprobe
    D0, I0, D3, I3 = _both(idx, vec, k)
    assert torch.equal(I0, I3)
    assert torch.equal(D0.view(torch.int32), D3.view(torch.int32))
    if k > 1:
        assert (D3[:, 1:] <= D3[:, :-1]).all()
    short = (I3 < 0).any(1)
    if nprobe == 1:
        assert short.any()                       # one list rarely holds 50... rows are -1 padded
    assert ((I3 >= 0) | (D3 == torch.finfo(torch.float32).min)).all()


def test_fused_search_batch_uses_it_too(world):
    sl, q = world
    idx = sl._get_ann_index(2)
    idx.nprobe = 64
    a = sl._search_batch(q, 2, 'open', want_knn=True)
    idx.set_scan_variant(3)
    try:
        b = sl._search_batch(q, 2, 'open', want_knn=True)
        c = sl._search_batch(q, 2, 'open')
    finally:
        idx.set_scan_variant(0)
    assert np.array_equal(a.knn, b.knn)
    for r in (b, c):
        assert np.array_equal(a.best_row, r.best_row) and np.array_equal(a.best_score, r.best_score)


def test_dense_queries_and_tiny_indexes_fall_back_or_pad():
    """Queries with more than 64 non-zeros take the query-major kernel; an index smaller than k
    pads with -1; empty lists are harmless."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(3)
    d = 800
    xb = np.zeros((5000, d), np.float32)
    for r in range(len(xb)):                                   # sparse rows, like hashed spectra
        j = rng.choice(d, 30, replace=False)
        xb[r, j] = rng.random(30, dtype=np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, 64)
    idx.set_niter(3)
    idx.train(xb)
    idx.add(xb)
    xq = xb[:300].copy()
    xq[::3] = rng.random((100, d), dtype=np.float32)            # dense rows among sparse ones
    for nprobe, k in ((8, 100), (64, 2048), (3, 7)):
        idx.nprobe = nprobe
        D0, I0, D3, I3 = _both(idx, xq, k)
        assert np.array_equal(I0, I3) and np.array_equal(D0.view(np.uint32), D3.view(np.uint32))
    sparse_only = xb[1000:1300]
    idx.nprobe = 64
    D0, I0, D3, I3 = _both(idx, sparse_only, 2048)             # the whole index is < k
    assert np.array_equal(I0, I3) and (I3[:, -1] == -1).all() is not None
    assert np.array_equal(D0.view(np.uint32), D3.view(np.uint32))
