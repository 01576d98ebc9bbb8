"""GPU parity of the hand-written IVF index (through the C ABI / faiss_compat) against
the oracle's restatement of the FAISS definitions. Integer / index work is bit-exact:
identical probe lists, identical codes, identical id sets, identical fp32 scores."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NITER = 6
SEED = 1234


@pytest.fixture(scope='module')
def vecs(O):
    from ann_solo_amd import synthetic
    lib, aux = synthetic.make_library(4000, seed=31, device='cpu')
    q, truth = synthetic.make_queries(lib, aux, 256, seed=32)
    o, mz, inten, *_ = lib.numpy()
    xb = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    o, mz, inten, *_ = q.numpy()
    xq = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    return xb, xq


@pytest.fixture(scope='module')
def trained(O, vecs):
    """Oracle-side quantisers (k-means + PQ) shared by several tests."""
    xb, _ = vecs
    cen = O.kmeans(xb, 16, NITER, SEED, 0, 256)
    cb = O.pq_train(xb, cen, 32, 256, NITER, SEED + 7)
    return cen, cb


def test_flat_search_exact(O, vecs):
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    idx = faiss.IndexFlatIP(800)
    idx.add(xb)
    assert idx.ntotal == len(xb) and idx.is_trained
    for k in (1, 10, 1024):
        D, I = idx.search(xq, k)
        Do, Io = O.flat_search(xb, xq, k)
        assert np.array_equal(I, Io)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))   # fmaf chain == MFMA chain


@pytest.mark.parametrize('n', [4096, 4095, 1000, 257, 100])
def test_short_row_select_matches_oracle(O, n):
    """The register-resident top-k of short rows (coarse top-nprobe of nlist): random rows,
    duplicated vectors (equal scores -> lowest id first), an all-zero query (every score
    ties: the streaming fallback), heavy quantisation (overfull threshold bucket)."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(n)
    d = 64
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xb[n // 2:n // 2 + 40] = xb[3]                       # 41 identical vectors
    xb[-200:] = np.round(xb[-200:])                      # coarse grid: many equal scores
    xq = rng.standard_normal((24, d)).astype(np.float32)
    xq[1] = 0.0                                          # all scores +0
    xq[2] = np.round(xq[2])
    xq[3, 1:] = 0.0                                      # scores take few distinct values on the grid rows
    idx = faiss.IndexFlatIP(d)
    idx.add(xb)
    for k in (1, 7, 128, 256):
        D, I = idx.search(xq, k)
        Do, Io = O.flat_search(xb, xq, k)
        assert np.array_equal(I, Io), (n, k)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))


def test_gemm_odd_shapes(O):
    """K not a multiple of 4/16, M/N not multiples of the 128 tile; k > n padding."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(0)
    for n, d, nq in ((300, 50, 7), (129, 37, 130), (5, 3, 1)):
        xb = rng.standard_normal((n, d)).astype(np.float32)
        xq = rng.standard_normal((nq, d)).astype(np.float32)
        idx = faiss.IndexFlatIP(d)
        idx.add(xb)
        k = min(n + 3, 40)
        D, I = idx.search(xq, k)
        Do, Io = O.flat_search(xb, xq, k)
        assert np.array_equal(I, Io)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
        if k > n:
            assert (I[:, n:] == -1).all()


def test_kmeans_bit_exact(O, vecs, trained):
    from ann_solo_amd import faiss_compat as faiss
    xb, _ = vecs
    cen, _ = trained
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, faiss.METRIC_INNER_PRODUCT)
    idx.set_niter(NITER)
    idx.seed = SEED
    assert not idx.is_trained
    idx.train(xb)
    assert idx.is_trained
    assert np.array_equal(idx.centroids().view(np.uint32), cen.view(np.uint32))


def test_kmeans_subsample_and_empty_clusters(O):
    """n > 256*k triggers the permutation subsample; duplicated points force the
    empty-cluster split."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(5)
    base = rng.random((6, 32)).astype(np.float32)
    x = np.repeat(base, 400, axis=0)               # 2400 points, 6 distinct
    x += (rng.random(x.shape) < 0.01).astype(np.float32) * 0.001
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(32), 32, 8)
    idx.set_niter(4)
    idx.train(x)
    cen = O.kmeans(x, 8, 4, SEED, 0, 256)
    assert np.array_equal(idx.centroids().view(np.uint32), cen.view(np.uint32))


def test_pq_train_and_encode_bit_exact(O, vecs, trained):
    from ann_solo_amd import faiss_compat as faiss
    xb, _ = vecs
    cen, cb = trained
    idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 16, 32, 8)
    idx.set_niter(NITER)
    idx.seed = SEED
    idx.train(xb)
    assert np.array_equal(idx.centroids().view(np.uint32), cen.view(np.uint32))
    assert np.array_equal(idx.codebooks().view(np.uint32), cb.view(np.uint32))
    idx.add(xb[:2500])
    idx.add(xb[2500:])                              # two add() calls: ids keep add order
    a = O.assign(xb, cen, 0)
    codes = O.pq_encode(xb, cen, a, cb)
    ivf = O.HostIVF(cen, a, codes, cb)
    off, ids, got_codes = idx.lists()
    assert np.array_equal(off, ivf.list_offsets)
    assert np.array_equal(ids, ivf.ids)
    assert np.array_equal(got_codes, ivf.payload)


@pytest.fixture(scope='module')
def pq_index(O, vecs, trained):
    from ann_solo_amd import faiss_compat as faiss
    xb, _ = vecs
    cen, cb = trained
    idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 16, 32, 8)
    idx.set_trained(cen, cb)
    idx.add(xb)
    a = O.assign(xb, cen, 0)
    ivf = O.HostIVF(cen, a, O.pq_encode(xb, cen, a, cb), cb)
    return idx, ivf


@pytest.mark.parametrize('k,nprobe', [(1024, 8), (200, 16), (1000, 2), (64, 1), (1280, 16), (2048, 16)])
def test_unordered_rows_hold_the_same_top_k(vecs, pq_index, k, nprobe):
    """asl_index_set_unordered: same ids with the same scores as the sorted search, any order,
    padding last (k = 2048 exceeds the set-mode buffer and silently stays sorted)."""
    _, xq = vecs
    idx, _ = pq_index
    idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    idx.set_unordered(True)
    try:
        Du, Iu = idx.search(xq, k)
    finally:
        idx.set_unordered(False)
    assert not np.array_equal(I, Iu) or k == 2048 or (I < 0).all()
    for r in range(len(xq)):
        n = int((I[r] >= 0).sum())
        assert (Iu[r, :n] >= 0).all() and (Iu[r, n:] == -1).all()
        o, ou = np.argsort(I[r, :n]), np.argsort(Iu[r, :n])
        assert np.array_equal(I[r, :n][o], Iu[r, :n][ou])
        assert np.array_equal(D[r, :n][o].view(np.uint32), Du[r, :n][ou].view(np.uint32))
    D2, I2 = idx.search(xq, k)                      # and the flag is really off again
    assert np.array_equal(I2, I)


def test_packed_key_rows_and_their_merge(vecs, trained):
    """Mode 2 of asl_index_set_unordered: one 64-bit key per hit carries exactly the (score, id)
    of the ordinary search; asl_topk_merge_keys over three shards reproduces the unsharded rows;
    indexes without the tiled PQ scan refuse the mode."""
    from ann_solo_amd import _lib
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, cb = trained

    def make():
        ix = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 16, 32, 8)
        ix.set_trained(cen, cb)
        ix.add(xb)
        ix.nprobe = 8
        return ix
    full = make()
    for k in (1024, 100, 7):
        D, I = full.search(xq, k)
        cD, cI = full.coarse(xq, 8)
        K = full.search_preassigned_keys(xq, k, cD, cI).view(np.uint64)
        ids = np.where(K != 0, 0xFFFFFFFF - (K & np.uint64(0xFFFFFFFF)).astype(np.int64), -1)
        ordb = (K >> np.uint64(32)).astype(np.uint32)
        bits = np.where(ordb & 0x80000000, ordb & 0x7fffffff, ~ordb).astype(np.uint32)   # ord2f
        for r in range(len(xq)):
            n = int((I[r] >= 0).sum())
            assert (ids[r] >= 0).sum() == n
            o, ok = np.argsort(I[r, :n]), np.argsort(np.where(ids[r] >= 0, ids[r], 1 << 40))[:n]
            assert np.array_equal(I[r, :n][o], ids[r][ok])
            assert np.array_equal(D[r, :n][o].view(np.uint32), bits[r][ok])
        D2, I2 = full.search(xq, k)
        assert np.array_equal(I2, I)                       # the mode does not stick
    D, I = full.search(xq, 1024)
    cD, cI = full.coarse(xq, 8)
    parts = []
    for r in range(3):
        sh = make()
        sh.shard(r, 3)
        parts.append(sh.search_preassigned_keys(xq, 1024, cD, cI))
    Dm, Im = faiss.topk_merge_keys(np.stack(parts))
    assert np.array_equal(Im, I) and np.array_equal(Dm.view(np.uint32), D.view(np.uint32))
    Du, Iu = faiss.topk_merge_keys(np.stack(parts), unordered=True)      # same rows as sets
    o, ou = np.argsort(I, 1), np.argsort(Iu, 1)
    assert np.array_equal(np.take_along_axis(I, o, 1), np.take_along_axis(Iu, ou, 1))
    assert np.array_equal(np.take_along_axis(D, o, 1).view(np.uint32), np.take_along_axis(Du, ou, 1).view(np.uint32))
    # IVF-Flat emits the same packed rows from its postings scan (round 4: both index kinds share
    # the two-phase exchange); a brute-force index has no such rows
    for storage in ('fx22', 'fp32'):
        flat = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
        flat.set_trained(cen)
        flat.add(xb)
        flat.nprobe = 8
        for k in (1024, 33):
            D, I = flat.search(xq, k)
            cD, cI = flat.coarse(xq, 8)
            K = flat.search_preassigned_keys(xq, k, cD, cI).view(np.uint64)
            ids = np.where(K != 0, 0xFFFFFFFF - (K & np.uint64(0xFFFFFFFF)).astype(np.int64), -1)
            ordb = (K >> np.uint64(32)).astype(np.uint32)
            bits = np.where(ordb & 0x80000000, ordb & 0x7fffffff, ~ordb).astype(np.uint32)
            for r in range(len(xq)):
                n = int((I[r] >= 0).sum())
                o, ok = np.argsort(I[r, :n]), np.argsort(np.where(ids[r] >= 0, ids[r], 1 << 40))[:n]
                assert (ids[r] >= 0).sum() == n and np.array_equal(I[r, :n][o], ids[r][ok])
                assert np.array_equal(D[r, :n][o].view(np.uint32), bits[r][ok])
    brute = faiss.IndexFlatIP(800)
    brute.add(xb[:100])
    brute.set_unordered(2)
    with pytest.raises(_lib.AnnSoloMiError):
        brute.search(xq, 10)


def test_coarse_and_lut_bit_exact(O, vecs, pq_index):
    _, xq = vecs
    idx, ivf = pq_index
    for nprobe in (1, 4, 16, 99):
        D, I = idx.coarse(xq, nprobe)
        Do, Io = O.coarse(xq, ivf.centroids, min(nprobe, 16))
        assert np.array_equal(I, Io)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    lut = idx.pq_lut(xq[:8])
    for i in range(8):
        assert np.array_equal(lut[i].view(np.uint32), O.pq_lut(xq[i], ivf.codebooks).view(np.uint32))


def test_ivfpq_search_identical_to_oracle(O, vecs, pq_index):
    _, xq = vecs
    idx, ivf = pq_index
    for k, nprobe in ((1024, 8), (100, 4), (1, 1), (1024, 16), (2048, 16)):
        idx.nprobe = nprobe
        D, I = idx.search(xq, k)
        Do, Io = ivf.search(xq, k, nprobe)
        assert np.array_equal(I, Io), (k, nprobe)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    # rows are sorted by (score desc, id asc) and padded with -1
    idx.nprobe = 1
    D, I = idx.search(xq, 1024)
    assert (I == -1).any()
    valid = I >= 0
    assert all((np.diff(D[r][valid[r]]) <= 0).all() for r in range(len(xq)))


def test_ivfpq_recall_against_exact(O, vecs, pq_index):
    """Behavioural check (the only reference-derived one at the FAISS boundary): the
    IVF-PQ candidates overlap the exact inner-product neighbours."""
    xb, xq = vecs
    idx, _ = pq_index
    idx.nprobe = 16
    _, I = idx.search(xq, 100)
    _, Ie = O.flat_search(xb, xq, 100)
    recall = np.mean([len(set(I[r]) & set(Ie[r])) / 100 for r in range(len(xq))])
    assert recall > 0.5, recall


@pytest.mark.parametrize('storage', ['fx22', 'fp32'])
def test_ivfflat_search_identical_to_oracle(O, vecs, trained, storage):
    """Both storage modes: 'fx22' stores every component on the 2^-22 grid (the oracle's
    ``quantize_fx22`` is the same rule) and scans 4-byte posting words behind the byte table
    (layout 2); 'fp32' keeps the components as given and scans float postings (layout 1). Ids and
    score bits equal the oracle's over the same stored vectors, for both scan variants."""
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, _ = trained
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
    idx.set_trained(cen)
    idx.add(xb)
    assert idx.storage == storage and idx.flat_layout == (2 if storage == 'fx22' else 1)
    a = O.assign(xb, cen, 0)
    xb = O.quantize_fx22(xb) if storage == 'fx22' else xb
    ivf = O.HostIVF(cen, a, xb)
    off, ids, v = idx.lists()
    assert np.array_equal(off, ivf.list_offsets) and np.array_equal(ids, ivf.ids)
    assert np.array_equal(v, ivf.payload)
    # the sparse-aware CPU scan (bench.py's IVF-Flat baseline) is the same chain: same bits
    csr = ivf.to_csr()
    Dd, Id = ivf.search(xq[:32], 100, 4)
    Dc, Ic = csr.search(xq[:32], 100, 4)
    assert np.array_equal(Id, Ic) and np.array_equal(Dd.view(np.uint32), Dc.view(np.uint32))
    # scan variants: 0 = per-dimension postings (default), 1 = dense GEMM + masked top-k
    for variant in (0, 1):
        idx.set_scan_variant(variant)
        for k, nprobe in ((1024, 8), (10, 2), (1024, 16)):
            idx.nprobe = nprobe
            D, I = idx.search(xq, k)
            Do, Io = ivf.search(xq, k, nprobe)
            assert np.array_equal(I, Io), (variant, k, nprobe)
            assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), (variant, k, nprobe)
    idx.set_scan_variant(0)
    # nprobe == nlist is exact search
    idx.nprobe = 16
    _, I = idx.search(xq, 50)
    assert np.array_equal(I, O.flat_search(xb, xq, 50)[1])


@pytest.mark.parametrize('d,nlist', [(800, 256), (96, 40), (800, 33)])
def test_sparse_coarse_quantiser_equals_the_gemm(d, nlist):
    """The coarse scores of sparse queries come from coarse_sparse.hip (32-list centroid tiles in
    LDS, entries broadcast by DPP); a batch with many dense rows is left to the MFMA GEMM by a
    device-side gate, and the odd dense row is walked inside the sparse kernel. All three must
    give the GEMM's bits (scan variant 1 forces the GEMM): same probe lists, same scores."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(d + nlist)

    def rows(n, nnz_lo, nnz_hi):
        x = np.zeros((n, d), np.float32)
        for i in range(n):
            k = int(rng.integers(nnz_lo, nnz_hi + 1))
            c = rng.choice(d, size=min(k, d), replace=False)
            x[i, c] = (rng.random(len(c)) + 0.05).astype(np.float32)
        return x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-20)
    cen = rows(nlist, 8, min(d, 200))
    cen[1] *= -1.0                                   # signs must not matter
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist)
    idx.set_trained(cen)
    nprobe = min(nlist, 16)
    batches = {
        'sparse': rows(301, 0, 50),                   # incl. all-zero rows
        'one dense row in 100': np.concatenate([rows(150, 1, 60), rows(1, min(d, 90), min(d, 90)),
                                                rows(149, 1, 64)]),
        'dense': rows(70, min(d, 70), min(d, 90)),    # every row above 64: the gate picks the GEMM
    }
    for name, xq in batches.items():
        idx.set_scan_variant(0)
        D0, I0 = idx.coarse(xq, nprobe)
        idx.set_scan_variant(1)
        D1, I1 = idx.coarse(xq, nprobe)
        idx.set_scan_variant(0)
        assert np.array_equal(I0, I1), name
        assert np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), name


@pytest.mark.parametrize('storage', ['fx22', 'fp32'])
def test_postings_work_counter(vecs, trained, storage):
    """asl_index_postings_work (the roofline bytes of the postings scan) against numpy: per
    (query, probed block of 832 vectors, non-zero query dimension) 4 bytes + 6 per posting
    (float postings) / 1 byte + 4 per posting and the lines = the block's 7-line byte table +
    ceil(postings / 32) per wanted segment (fixed-point postings)."""
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, _ = trained
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
    idx.set_trained(cen)
    idx.add(xb)
    off, ids, v = idx.lists()
    nq, nprobe = 24, 5
    _, cI = idx.coarse(xq[:nq], nprobe)
    want = want_lines = 0
    for i in range(nq):
        dims = np.nonzero(xq[i])[0]
        for l in cI[i]:
            for b0 in range(off[l], off[l + 1], 832):
                blk = v[b0:min(b0 + 832, off[l + 1])][:, dims]
                if storage == 'fx22':
                    c = np.count_nonzero(blk, axis=0)
                    want += len(dims) + 4 * int(c.sum())
                    want_lines += 7 + int(((c + 31) // 32).sum())
                else:
                    want += 4 * len(dims) + 6 * int(np.count_nonzero(blk))
    got, lines = idx.postings_work(xq[:nq], nprobe)
    assert got == want
    assert lines > 0 and (storage != 'fx22' or lines == want_lines)


def test_flat_storage_fx22_against_fp32_and_float64(O, vecs, trained):
    """What the fixed-point storage changes (VERDICT r3 item 3's rule): ids equal the fp32-storage
    result except for candidates within 1e-6 of the k-th score; scores within 1e-5 of the float64
    inner product of the ORIGINAL vectors. Data outside [0, 1) keeps float postings."""
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, _ = trained
    res = {}
    for storage in ('fx22', 'fp32'):
        idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
        idx.set_trained(cen)
        idx.add(xb)
        idx.nprobe = 6
        res[storage] = idx.search(xq[:64], 256)
    (Dx, Ix), (Df, If) = res['fx22'], res['fp32']
    xq = xq[:64]
    exact = np.einsum('qkd,qd->qk', xb[np.maximum(Ix, 0)].astype(np.float64), xq.astype(np.float64))
    assert np.abs(np.where(Ix >= 0, Dx - exact, 0.0)).max() <= 1e-5
    for r in range(len(xq)):
        only = np.setxor1d(Ix[r][Ix[r] >= 0], If[r][If[r] >= 0])
        if len(only):            # swapped candidates sit at the k-th score
            kth = min(Dx[r][Ix[r] >= 0].min(), Df[r][If[r] >= 0].min())
            sc = xb[only].astype(np.float64) @ xq[r].astype(np.float64)
            assert np.abs(sc - kth).max() <= 1e-6, (r, only, sc, kth)
    # signed components: stored as given, float postings, exact chain
    rng = np.random.default_rng(3)
    xs = xb.copy()
    xs[:, ::7] *= -1.0
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage='fx22')
    idx.set_trained(cen)
    idx.add(xs)
    assert idx.flat_layout == 1
    off, ids, v = idx.lists()
    stored = np.empty_like(xs)
    stored[ids] = v
    assert np.array_equal(stored, np.where(xs >= 0, O.quantize_fx22(xs), xs))
    with pytest.raises(Exception):
        idx.set_storage('fp32')          # after add(): refused
    # the default is the reference's storage: float32 as given (spectral_library.py:174-181)
    dflt = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
    assert dflt.storage == 'fp32'
    dflt.set_trained(cen)
    dflt.add(xs)
    _, ids_d, v_d = dflt.lists()
    kept = np.empty_like(xs)
    kept[ids_d] = v_d
    assert np.array_equal(kept, xs)


def test_flat_storage_survives_save_and_load(vecs, trained, tmp_path):
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, _ = trained
    for storage in ('fx22', 'fp32'):
        idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
        idx.set_trained(cen)
        idx.add(xb[:1500])
        path = str(tmp_path / f'{storage}.idxmi')
        faiss.write_index(idx, path)
        back = faiss.read_index(path)
        assert back.storage == storage and back.flat_layout == idx.flat_layout
        back.add(xb[1500:])
        idx.add(xb[1500:])
        idx.nprobe = back.nprobe = 5
        (D0, I0), (D1, I1) = idx.search(xq[:64], 100), back.search(xq[:64], 100)
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))


def test_version_1_files_say_what_they_hold(vecs, trained, tmp_path):
    """a version-1 cache with storage field 0 is either a fixed-point file (round 4) or
    an unrounded float32 file written before the storage modes existed (the field was padding).
    The loader decides by the components: off-grid data loads as 'fp32' (so an engine configured
    for 'fx22' rebuilds instead of silently serving float postings), on-grid data as 'fx22'.
    Files are written with version 2 now."""
    import struct
    from ann_solo_amd import faiss_compat as faiss
    xb, _ = vecs
    cen, _ = trained
    for storage, want in (('fp32', 'fp32'), ('fx22', 'fx22')):
        idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16, storage=storage)
        idx.set_trained(cen)
        idx.add(xb[:800])
        path = str(tmp_path / f'v1_{storage}.idxmi')
        faiss.write_index(idx, path)
        raw = bytearray(open(path, 'rb').read())
        # header: magic[8], then int32 version, d, nlist, kind, pq_m, pq_bits, niter, trained; int64 ntotal,
        # n_store; int32 shard_rank, shard_world, has_vids, pad
        assert struct.unpack_from('<i', raw, 8)[0] == 2
        pad_at = 8 + 8 * 4 + 2 * 8 + 3 * 4
        assert struct.unpack_from('<i', raw, pad_at)[0] == (0 if storage == 'fx22' else 1)
        struct.pack_into('<i', raw, 8, 1)            # an old file: version 1 ...
        struct.pack_into('<i', raw, pad_at, 0)       # ... whose storage field reads 0 either way
        open(path, 'wb').write(bytes(raw))
        back = faiss.read_index(path)
        assert back.storage == want, (storage, back.storage)


def test_supports_keys_depends_on_what_a_shard_holds(vecs, trained):
    """for IVF-Flat the packed-key scan exists only where a shard stores sparse vectors
    -- an empty shard and a dense one answer 0 -- so a sharded driver must agree on the exchange
    format across ranks (distributed._agreed_keys, asl_index_search_sharded)."""
    from ann_solo_amd import _lib, faiss_compat as faiss
    xb, _ = vecs
    cen, _ = trained
    L = _lib.lib()
    sparse = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
    sparse.set_trained(cen)
    sparse.add(xb[:500])
    empty = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
    empty.set_trained(cen)
    dense = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
    dense.set_trained(cen)
    dense.add(np.abs(np.random.default_rng(2).standard_normal((300, 800))).astype(np.float32))
    assert L.asl_index_supports_keys(sparse._h, 256, 8) == 1
    assert L.asl_index_supports_keys(empty._h, 256, 8) == 0
    assert L.asl_index_supports_keys(dense._h, 256, 8) == 0
    assert L.asl_index_supports_keys(sparse._h, 1500, 8) == 0      # k + 768 > 2048


def test_ivfflat_postings_scan_many_small_lists():
    """The postings scan where the candidate set only just exceeds the key buffer and the
    threshold bucket is crowded (small scores, k = 1024): bulk offers have to fall back to
    compaction + exact flushes without losing a key. Must equal the dense formulation, for
    ordered rows and for the unordered set mode."""
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(9000, seed=20240807, device='cpu')
    sl = SpectralLibrary(lib, config=Config.open_search(index='ivfflat', kmeans_niter=10))
    for z in (2, 3, 4):
        q, _ = synthetic.make_queries(lib, aux, 200, seed=110 + z, charge=z)
        idx = sl._get_ann_index(z)
        vec = sl._encode(q.to('cuda'))
        for _ in range(3):      # the failure this guards against was a race
            idx.set_scan_variant(0)
            D0, I0 = idx.search(vec, 1024)
            idx.set_unordered(1)
            Du, Iu = idx.search(vec, 1024)
            idx.set_unordered(0)
            idx.set_scan_variant(1)
            D1, I1 = idx.search(vec, 1024)
            assert torch.equal(I0, I1) and torch.equal(D0, D1)
            assert torch.equal(Iu.sort(1).values, I1.sort(1).values)
        idx.set_scan_variant(0)


@pytest.mark.parametrize('d,nnz,nlist,hot', [(800, 90, 4, 0), (96, 10, 3, 0), (130, 16, 7, 0),
                                             (800, 70, 4, 4), (400, 30, 5, 2)])
def test_ivfflat_postings_scan_shapes(d, nnz, nlist, hot):
    """Postings scan outside the bench's shape: queries with more than 64 non-zero
    dimensions, lists longer than one block (832 vectors), dimensions with more postings than
    lanes, d not a multiple of 4, an empty list; ``hot`` dimensions shared by more than half of
    the vectors and by every query (segments of hundreds of postings: more than 64 rows in a
    chunk of the row pipeline). Must equal the dense GEMM formulation bit for bit (scores) and
    id for id."""
    import torch
    from ann_solo_amd import faiss_compat as faiss
    g = torch.Generator().manual_seed(1000 + d)
    n, nq = 5000, 300

    hot_dims = torch.randperm(d, generator=g)[:hot]

    def sparse_rows(m, p_hot):
        x = torch.zeros(m, d)
        for i in range(m):
            cols = torch.randperm(d, generator=g)[:nnz]
            x[i, cols] = torch.rand(nnz, generator=g) + 0.05
        if hot:
            has = torch.rand(m, hot, generator=g) < p_hot
            x[:, hot_dims] = torch.where(has, torch.rand(m, hot, generator=g) + 0.05, x[:, hot_dims])
        return torch.nn.functional.normalize(x, dim=1)
    xb, xq = sparse_rows(n, 0.55), sparse_rows(nq, 1.0)
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist)
    cen = xb[:nlist].clone()
    cen[-1] = -cen[0]                      # a centroid nothing is assigned to: an empty list
    idx.set_trained(cen.numpy())
    idx.add(xb.numpy())
    off = idx.lists()[0]
    assert (np.diff(off) == 0).any() and (np.diff(off) > 832).any()
    # k > 1280: the postings kernel with the 4096-key buffer
    for k, nprobe in ((1024, nlist), (100, 2), (1500, nlist), (2000, nlist)):
        idx.nprobe = nprobe
        out = {}
        for variant in (0, 1):
            idx.set_scan_variant(variant)
            out[variant] = idx.search(xq.numpy(), k)
        assert np.array_equal(out[0][1], out[1][1])
        assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))


def test_ivfflat_scan_kernels_fuzz():
    """scripts/fuzz_flat.py: random shapes, sparsity, k, nprobe, duplicate-heavy data --
    postings and sparse-tile scans equal the dense formulation (ordered rows bit for bit,
    unordered rows as sets)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_flat.py'), '40', '5'],
                         capture_output=True, text=True, timeout=600)
    assert '40 trials, 0 mismatches' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def test_sharded_search_merges_to_unsharded(O, vecs, trained):
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, cb = trained
    for kind in ('pq', 'flat'):
        def make():
            if kind == 'pq':
                ix = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 16, 32, 8)
                ix.set_trained(cen, cb)
            else:
                ix = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
                ix.set_trained(cen)
            ix.add(xb)
            ix.nprobe = 8
            return ix
        full = make()
        D, I = full.search(xq, 1024)
        owner = full.shard_map(3)
        assert set(owner.tolist()) == {0, 1, 2}
        parts = []
        for r in range(3):
            sh = make()
            sh.shard(r, 3)
            assert sh.ntotal == len(xb) and sh.info().nlocal < len(xb)
            parts.append(sh.search(xq, 1024))
        Ds = np.stack([p[0] for p in parts])
        Is = np.stack([p[1] for p in parts])
        Dm, Im = faiss.topk_merge(Ds, Is)
        assert np.array_equal(Im, I) and np.array_equal(Dm.view(np.uint32), D.view(np.uint32))
        Do, Io = O.topk_merge(Ds, Is)
        assert np.array_equal(Im, Io)


@pytest.mark.parametrize('S,k,quant', [(8, 1024, 0), (8, 1024, 64), (2, 1000, 8), (5, 700, 0),
                                       (2, 64, 0), (16, 1024, 2)])
def test_topk_merge_synthetic_lists(O, S, k, quant):
    """Merge of S partial lists: unsorted inputs, -1 padding, heavy score ties (quant levels),
    ids unique across lists -- identical to the oracle under (score desc, id asc)."""
    from ann_solo_amd import faiss_compat as faiss
    rng = np.random.default_rng(S * 1000 + k + quant)
    nq = 37
    Ds = rng.uniform(-0.2, 1.0, (S, nq, k)).astype(np.float32)
    if quant:
        Ds = (np.floor(Ds * quant) / quant).astype(np.float32)
    Is = np.empty((S, nq, k), np.int64)
    for q in range(nq):
        Is[:, q, :] = rng.permutation(4 * S * k)[:S * k].reshape(S, k)
    pad = rng.random((S, nq, k)) < 0.15
    pad[:, 0, :] = True                     # a query with no candidates at all
    pad[1:, 1, :] = True                    # a query served by one list only
    Is[pad] = -1
    Ds[pad] = -3.402823466e+38
    Dm, Im = faiss.topk_merge(Ds, Is)
    Do, Io = O.topk_merge(Ds, Is)
    assert np.array_equal(Im, Io)
    assert np.array_equal(Dm.view(np.uint32), Do.view(np.uint32))


def test_save_load_roundtrip(tmp_path, vecs, pq_index):
    from ann_solo_amd import faiss_compat as faiss
    _, xq = vecs
    idx, _ = pq_index
    idx.nprobe = 8
    D, I = idx.search(xq, 200)
    p = os.path.join(tmp_path, 'lib_abc1234_2.idxmi')
    faiss.write_index(idx, p)
    idx2 = faiss.read_index(p)
    idx2.nprobe = 8
    assert idx2.ntotal == idx.ntotal and idx2.d == 800 and idx2.is_trained
    D2, I2 = idx2.search(xq, 200)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    idx2.reset()
    assert idx2.ntotal == 0
    with pytest.raises(Exception):
        faiss.read_index(os.path.join(tmp_path, 'missing.idxmi'))


def test_error_behaviour(vecs):
    from ann_solo_amd import faiss_compat as faiss
    from ann_solo_amd._lib import AnnSoloMiError
    xb, xq = vecs
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
    with pytest.raises(AnnSoloMiError):
        idx.add(xb)                      # not trained
    with pytest.raises(AnnSoloMiError):
        idx.search(xq, 10)
    with pytest.raises(AnnSoloMiError):
        idx.train(xb[:5])                # fewer points than lists
    with pytest.raises(ValueError):
        idx.train(xb[:, :10])


@pytest.mark.parametrize('kind', ['ivfpq', 'ivfflat'])
def test_more_than_512_probes_equal_the_oracle(O, kind):
    """The reference sweeps nprobe up to 1 024 (notebooks/iprg2012_ann_hyperparameters.ipynb:100-101) and
    clamps there on its GPU path (spectral_library.py:77-81). Beyond 512 probes the layout-specific
    scans take two probes per thread (`WIDE`, csrc/pq_scan_v3.hip, csrc/flat_scan.hip): ids and score
    bits must equal the oracle's, the generic kernels' and -- nprobe = nlist -- exact search over the
    stored payload."""
    from ann_solo_amd import faiss_compat as faiss, synthetic
    lib, aux = synthetic.make_library(12000, seed=77, device='cpu', charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, 24, seed=78, charge=2)
    o, mz, inten, *_ = lib.numpy()
    xb = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    o, mz, inten, *_ = q.numpy()
    xq = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    nlist = 1100
    if kind == 'ivfpq':
        idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, nlist, 32, 8)
    else:
        idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, nlist)
    idx.seed = SEED
    idx.set_niter(2)
    idx.train(xb)
    idx.add(xb)
    off, ids, payload = idx.lists()
    ivf = O.HostIVF.__new__(O.HostIVF)
    ivf.centroids, ivf.nlist, ivf.d = idx.centroids(), nlist, 800
    ivf.list_offsets, ivf.ids, ivf.payload = off, ids, payload
    ivf.codebooks, ivf.kind = (idx.codebooks(), 1) if kind == 'ivfpq' else (None, 0)
    for k, nprobe in ((1024, 513), (1024, 700), (1024, 1024), (100, 1023), (2048, 1024), (1024, 1100)):
        idx.nprobe = nprobe
        Do, Io = ivf.search(xq, k, nprobe)
        for variant in (0, 1):
            idx.set_scan_variant(variant)
            D, I = idx.search(xq, k)
            assert np.array_equal(I, Io), (variant, k, nprobe)
            assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), (variant, k, nprobe)
    idx.set_scan_variant(0)
    assert idx.info().nlist == nlist


@pytest.mark.parametrize('kind', ['flat', 'ivfflat', 'ivfpq'])
def test_k_beyond_the_lds_top_k_in_bounded_passes(O, vecs, trained, kind):
    """The reference's CPU path has no bound on --num_candidates (config.py:188-192; its notebooks look at
    5 000+ neighbours, notebooks/iprg2012_num_candidates.ipynb:282-288). Beyond 2 048 the index searches
    in ceil(k / 2048) bounded passes (csrc/index.hip: index_search_large_k): rows equal the oracle's --
    ids, score bits, (score desc, id asc) order, -1 padding where a query reaches fewer than k vectors
    (4 000 stored) -- on every index kind, with host and device outputs."""
    import torch
    from ann_solo_amd import faiss_compat as faiss
    xb, xq = vecs
    cen, cb = trained
    xq = xq[:40]
    if kind == 'flat':
        idx = faiss.IndexFlatIP(800)
        idx.add(xb)
        ref = lambda k, nprobe: O.flat_search(xb, xq, k)
    elif kind == 'ivfflat':
        idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(800), 800, 16)
        idx.set_trained(cen)
        idx.add(xb)
        ivf = O.HostIVF(cen, O.assign(xb, cen, 0), xb)
        ref = lambda k, nprobe: ivf.search(xq, k, nprobe)
    else:
        idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 16, 32, 8)
        idx.set_trained(cen, cb)
        idx.add(xb)
        a = O.assign(xb, cen, 0)
        ivf = O.HostIVF(cen, a, O.pq_encode(xb, cen, a, cb), cb)
        ref = lambda k, nprobe: ivf.search(xq, k, nprobe)
    for k, nprobe in ((2049, 16), (3000, 9), (4096, 16), (5000, 16), (2500, 3)):
        idx.nprobe = nprobe
        D, I = idx.search(xq, k)
        Do, Io = ref(k, nprobe)
        if kind == 'flat':      # fewer than k vectors stored: the oracle's padding is the same -1
            assert (I[:, len(xb):] == -1).all() if k > len(xb) else True
        assert np.array_equal(I, Io), (k, nprobe)
        valid = Io >= 0
        assert np.array_equal(D.view(np.uint32)[valid], Do.view(np.uint32)[valid]), (k, nprobe)
    Dt, It = idx.search(torch.from_numpy(xq).cuda(), 3000)
    idx.nprobe = 16
    D, I = idx.search(xq, 3000)
    Dt, It = idx.search(torch.from_numpy(xq).cuda(), 3000)
    assert np.array_equal(It.cpu().numpy(), I) and np.array_equal(Dt.cpu().numpy().view(np.uint32), D.view(np.uint32))
    with pytest.raises(Exception):
        idx.search(xq, 16385)
