"""Damaged FAISS index files are rejected with ValueError before anything reaches the device
(ann_solo_amd/faiss_compat.read_index_faiss; the engine's rebuild fallback in ``_get_ann_index``
catches exactly that) -- ADVICE r2: a corrupt 'sprs' table, absurd sizes, duplicated ids."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_faiss_file import _hand_made, _sparse_rows    # noqa: E402


@pytest.mark.parametrize('kw', [dict(sparse=True, list_oob=True), dict(sparse=True, size_huge=True),
                                dict(dup_ids=True), dict(sparse=True, dup_ids=True),
                                dict(ids_bad=True), dict(metric=1)])
def test_corrupt_tables_raise_value_error(tmp_path, kw):
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(5)
    d, nlist = 16, 6
    cen = _sparse_rows(rng, nlist, d, 4)
    x = _sparse_rows(rng, 40, d, 5)
    assign = rng.integers(0, nlist, 40)
    assign[assign == 4] = 1
    lists = {int(l): (np.nonzero(assign == l)[0], x[assign == l]) for l in np.unique(assign)}
    p = str(tmp_path / 'bad.idxann')
    _hand_made(p, d, cen, lists, **kw)
    with pytest.raises(ValueError):
        faiss.read_index_faiss(p)


def test_absurd_header_counts_raise_value_error(tmp_path):
    import struct
    from ann_solo_amd import faiss_compat as faiss
    p = str(tmp_path / 'hdr.idxann')
    hdr = struct.pack('<iqqqBi', 16, 1 << 40, 1 << 20, 1 << 20, 1, 0)
    open(p, 'wb').write(b'IwFl' + hdr + struct.pack('<QQ', 6, 5) + b'IxFI' + b'\0' * 64)
    with pytest.raises(ValueError):
        faiss.read_index_faiss(p)


@pytest.mark.parametrize('name', ['', '_sprs'])
def test_reader_parses_the_independent_byte_fixture(name):
    """tests/golden/faiss_ivfflat_kat*.idxann were written by tests/golden/make_faiss_fixture.py,
    which restates FAISS' writer (faiss/impl/index_write.cpp) field by field and imports nothing
    of this repository: the parser must recover exactly the content that script put in."""
    from ann_solo_amd import faiss_compat as faiss
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    want = np.load(os.path.join(g, 'faiss_ivfflat_kat.npz'))
    f = faiss.parse_index_faiss(os.path.join(g, f'faiss_ivfflat_kat{name}.idxann'))
    pre = 'sprs_' if name else ''
    assert f['d'] == int(want['d']) and f['nlist'] == int(want['nlist'])
    assert f['nprobe'] == int(want[pre + 'nprobe']) and f['ntotal'] == len(want[pre + 'x'])
    assert np.array_equal(f['centroids'].view(np.uint32), want['centroids'].view(np.uint32))
    assert np.array_equal(f['x'].view(np.uint32), want[pre + 'x'].view(np.uint32))
    assert np.array_equal(f['lists'], want[pre + 'lists'])
