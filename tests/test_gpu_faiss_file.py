"""FAISS' on-disk IndexIVFFlat layout (faiss/impl/index_write.cpp; the reference's .idxann cache,
spectral_library.py:181): export, import with FAISS' own centroids and list assignments kept, a
hand-assembled byte-level fixture, rejection of anything else, and the engine picking up an
existing reference cache. FAISS itself is absent here: the layout is restated from its source."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sparse_rows(rng, n, d, nnz):
    x = np.zeros((n, d), np.float32)
    for r in range(n):
        x[r, rng.choice(d, nnz, replace=False)] = rng.random(nnz, dtype=np.float32) + 0.1
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_round_trip_keeps_centroids_lists_and_results(tmp_path):
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(1)
    xb, xq = _sparse_rows(rng, 5000, 800, 30), _sparse_rows(rng, 200, 800, 30)
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 64)
    idx.set_niter(3)
    idx.train(xb)
    idx.add(xb)
    idx.nprobe = 9
    p = str(tmp_path / 'lib_abc1234_2.idxann')
    faiss.write_index_faiss(idx, p)
    raw = open(p, 'rb').read()
    assert raw[:4] == b'IwFl' and raw[4 + 33 + 16:4 + 33 + 20] == b'IxFI' and b'ilar' in raw
    assert struct.unpack_from('<iq', raw, 4) == (800, 5000)
    back = faiss.read_index_faiss(p)
    assert back.nprobe == 9 and back.ntotal == 5000
    assert np.array_equal(back.centroids().view(np.uint32), idx.centroids().view(np.uint32))
    for a, b in zip(idx.lists(), back.lists()):
        assert np.array_equal(a, b)
    D0, I0 = idx.search(xq, 300)
    D1, I1 = back.search(xq, 300)
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))


def _hand_made(path, d, cen, lists, sparse=False, metric=0, quant=b'IxFI', ids_bad=False,
               list_oob=False, size_huge=False, dup_ids=False):
    """A file assembled field by field from the layout, independently of write_index_faiss:
    ``lists``: {list: (ids, vectors)}; FAISS keeps whatever assignment it was given."""
    nlist = len(cen)
    ntotal = sum(len(v[0]) for v in lists.values())
    hdr = lambda dd, n, m: struct.pack('<i', dd) + struct.pack('<q', n) + struct.pack('<qq', 1 << 20, 1 << 20) + \
        struct.pack('<B', 1) + struct.pack('<i', m)
    out = b'IwFl' + hdr(d, ntotal, metric) + struct.pack('<Q', nlist) + struct.pack('<Q', 5)
    out += quant + hdr(d, nlist, metric) + struct.pack('<Q', nlist * d) + cen.astype('<f4').tobytes()
    out += struct.pack('<B', 0) + struct.pack('<Q', 0)
    out += b'ilar' + struct.pack('<QQ', nlist, 4 * d)
    sizes = [len(lists[l][0]) if l in lists else 0 for l in range(nlist)]
    if sparse:
        pairs = [(l, s) for l, s in enumerate(sizes) if s]
        if list_oob:
            pairs[-1] = (nlist + 3, pairs[-1][1])
        if size_huge:
            pairs[0] = (pairs[0][0], (1 << 63) + 5)
        out += b'sprs' + struct.pack('<Q', 2 * len(pairs)) + b''.join(struct.pack('<QQ', *p) for p in pairs)
    else:
        out += b'full' + struct.pack('<Q', nlist) + b''.join(struct.pack('<Q', s) for s in sizes)
    for l in range(nlist):
        if l in lists:
            ids, vec = lists[l]
            ids = np.asarray(ids, '<i8')
            if ids_bad:
                ids = ids + 1
            if dup_ids and len(ids) > 1:
                ids = ids.copy()
                ids[1] = ids[0]
            out += np.asarray(vec, '<f4').tobytes() + ids.tobytes()
    open(path, 'wb').write(out)


def test_hand_assembled_files_and_rejections(tmp_path):
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(2)
    d, nlist = 16, 6
    cen = _sparse_rows(rng, nlist, d, 4)
    x_file = _sparse_rows(rng, 40, d, 5)
    # an assignment FAISS might have made (NOT the arg-max one everywhere), ids scattered over lists
    assign = rng.integers(0, nlist, 40)
    assign[assign == 4] = 1                                   # list 4 stays empty
    lists = {int(l): (np.nonzero(assign == l)[0], x_file[assign == l]) for l in np.unique(assign)}
    from oracle import oracle_py as O
    for sparse, storage in ((False, 'fx22'), (True, 'fx22'), (False, 'fp32')):
        p = str(tmp_path / f'h{int(sparse)}.idxann')
        _hand_made(p, d, cen, lists, sparse=sparse)
        idx = faiss.read_index_faiss(p, storage=storage)
        # (the default storage of IVF-Flat keeps components in [0, 1) on the 2^-22 grid)
        x = x_file if storage == 'fp32' else O.quantize_fx22(x_file)
        off, ids, vecs = idx.lists()
        assert idx.ntotal == 40 and idx.nprobe == 5
        for l in range(nlist):
            got = ids[off[l]:off[l + 1]]
            want = lists[l][0] if l in lists else np.zeros(0, np.int64)
            assert np.array_equal(np.sort(got), np.sort(want))          # the file's assignment is kept
            assert np.array_equal(vecs[off[l]:off[l + 1]][np.argsort(got)], x[np.sort(want)])
        idx.nprobe = nlist
        D, I = idx.search(x[:5], 3)
        assert (I[:, 0] == np.arange(5)).all()                              # every vector finds itself
    bad = str(tmp_path / 'bad.idxann')
    for kw in (dict(metric=1), dict(quant=b'IxHN'), dict(ids_bad=True)):
        _hand_made(bad, d, cen, lists, **kw)
        with pytest.raises(ValueError):
            faiss.read_index_faiss(bad)
    good = open(str(tmp_path / 'h0.idxann'), 'rb').read()
    for blob in (good[:100], b'IwPQ' + good[4:], good[:-8]):
        open(bad, 'wb').write(blob)
        with pytest.raises(ValueError):
            faiss.read_index_faiss(bad)


def test_engine_imports_an_existing_reference_cache(tmp_path):
    """A library directory that already holds the reference's <library>_<hash7>_<z>.idxann: the
    engine loads it (no training: the file's centroids are the index's), answers like the index
    that wrote it, leaves the .idxann untouched and keeps its own container next to it."""
    from ann_solo_amd import faiss_compat as faiss, synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary, INDEX_EXT
    lib, aux = synthetic.make_library(6000, seed=61, device='cpu', charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, 200, seed=62, charge=2, open_range=300.0)
    cfg = Config.open_search(num_list=32, num_probe=8, num_candidates=256, index='ivfflat', kmeans_niter=7, seed=99)
    first = SpectralLibrary(lib, config=cfg)                       # "FAISS": some trainer, some seed
    want = first._search_batch(q, 2, 'open', want_knn=True)
    cen = first._get_ann_index(2).centroids()
    ref_cfg = Config.open_search(num_list=32, num_probe=8, num_candidates=256, index='ivfflat')
    probe = SpectralLibrary.__new__(SpectralLibrary)
    probe.config = ref_cfg
    name = f'lib_{probe._get_hyperparameter_hash()[:7]}_2.idxann'
    faiss.write_index_faiss(first._get_ann_index(2), str(tmp_path / name))
    stamp = os.path.getmtime(tmp_path / name)
    eng = SpectralLibrary(lib, config=ref_cfg, index_dir=str(tmp_path), basename='lib')
    assert eng.partitions[2].index is None                         # nothing was trained
    got = eng._search_batch(q, 2, 'open', want_knn=True)
    assert np.array_equal(eng._get_ann_index(2).centroids().view(np.uint32), cen.view(np.uint32))
    assert np.array_equal(got.knn, want.knn) and np.array_equal(got.best_row, want.best_row)
    assert os.path.getmtime(tmp_path / name) == stamp
    assert any(f.endswith(INDEX_EXT) for f in os.listdir(tmp_path))
    # import switched off, or a damaged file: a fresh index is built instead
    own = SpectralLibrary(lib, config=ref_cfg, index_dir=str(tmp_path / 'x'), basename='lib',
                          import_faiss_cache=False) if os.makedirs(tmp_path / 'x', exist_ok=True) is None else None
    assert own.partitions[2].index is not None
    os.makedirs(tmp_path / 'y')
    open(tmp_path / 'y' / name, 'wb').write(b'IwFl' + b'\\0' * 50)
    dmg = SpectralLibrary(lib, config=ref_cfg, index_dir=str(tmp_path / 'y'), basename='lib')
    r = dmg._search_batch(q, 2, 'open')
    assert (r.best_row >= 0).any() and dmg._get_ann_index(2).ntotal == lib.n


@pytest.mark.parametrize('name', ['', '_sprs'])
@pytest.mark.parametrize('storage', ['fx22', 'fp32'])
def test_writer_reproduces_the_independent_byte_fixture(tmp_path, name, storage):
    """The fixture of tests/golden/make_faiss_fixture.py (FAISS' writer restated field by field,
    nothing of this repository imported) through the device and back: import, search, export --
    the exported file must be the fixture's bytes (its components are multiples of 2^-12, so the
    fixed-point storage keeps them bit for bit as well)."""
    from ann_solo_amd import faiss_compat as faiss
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    src = os.path.join(g, f'faiss_ivfflat_kat{name}.idxann')
    want = np.load(os.path.join(g, 'faiss_ivfflat_kat.npz'))
    pre = 'sprs_' if name else ''
    idx = faiss.read_index_faiss(src, storage=storage)
    assert idx.ntotal == len(want[pre + 'x']) and idx.nprobe == int(want[pre + 'nprobe'])
    x = want[pre + 'x']
    off, ids, vecs = idx.lists()
    stored = np.empty_like(x)
    stored[ids] = vecs
    assert np.array_equal(stored.view(np.uint32), x.view(np.uint32))  # the file's vectors, bit for bit
    assert np.array_equal(want[pre + 'lists'][ids], np.repeat(np.arange(int(want['nlist'])), np.diff(off)))
    idx.nprobe = int(want['nlist'])
    D, I = idx.search(x[:8], 5)
    exact = (x[:8].astype(np.float64) @ x.astype(np.float64).T)
    assert np.allclose(np.sort(exact, 1)[:, ::-1][:, :5], D, rtol=0, atol=1e-6)
    idx.nprobe = int(want[pre + 'nprobe'])
    out = str(tmp_path / 'back.idxann')
    faiss.write_index_faiss(idx, out)
    assert open(out, 'rb').read() == open(src, 'rb').read()
