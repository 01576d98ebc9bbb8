"""The oracle's restatement of the FAISS definitions (PARITY UNPINNED at this boundary:
FAISS is absent and the reference holds no test here) checked against first principles:
exact brute force in float64, decomposition identities, sharding invariance."""
import numpy as np
import pytest


@pytest.fixture(scope='module')
def vecs(O):
    from ann_solo_amd import synthetic
    lib, aux = synthetic.make_library(1200, seed=61, device='cpu')
    q, _ = synthetic.make_queries(lib, aux, 40, seed=62)
    o, mz, inten, *_ = lib.numpy()
    xb = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    o, mz, inten, *_ = q.numpy()
    return xb, O.encode_batch(mz, inten, o, 10.96, 0.04, 800)


def test_flat_search_is_exact_inner_product(O, vecs):
    xb, xq = vecs
    D, I = O.flat_search(xb, xq, 20)
    ref = xq.astype(np.float64) @ xb.astype(np.float64).T
    for r in range(len(xq)):
        np.testing.assert_allclose(D[r], ref[r, I[r]], rtol=0, atol=2e-6)
        assert (np.diff(D[r]) <= 0).all()
        kth = D[r, -1]
        assert (ref[r] > kth + 1e-5).sum() <= 20
    # ties -> ascending id
    D2, I2 = O.flat_search(np.repeat(xb[:3], 2, axis=0), xq[:4], 6)
    for r in range(4):
        for a, b in zip(range(5), range(1, 6)):
            if D2[r, a] == D2[r, b]:
                assert I2[r, a] < I2[r, b]


def test_ivfflat_full_probe_equals_flat_and_padding(O, vecs):
    xb, xq = vecs
    cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
    a = O.assign(xb, cen, 0)
    ivf = O.HostIVF(cen, a, xb)
    D, I = ivf.search(xq, 30, 8)
    Df, If = O.flat_search(xb, xq, 30)
    assert np.array_equal(I, If) and np.array_equal(D, Df)
    D1, I1 = ivf.search(xq, 1024, 1)
    assert (I1 == -1).any() and (D1[I1 == -1] == -np.finfo(np.float32).max).all()
    # every returned id belongs to a probed list
    cD, cI = O.coarse(xq, cen, 1)
    for r in range(len(xq)):
        assert set(a[I1[r][I1[r] >= 0]]) <= {cI[r, 0]}


def test_adc_is_inner_product_with_the_reconstruction(O, vecs):
    xb, xq = vecs
    cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
    a = O.assign(xb, cen, 0)
    cb = O.pq_train(xb, cen, 8, 16, 4, 1241)          # m=8, 16 codewords, dsub=100
    codes = O.pq_encode(xb, cen, a, cb)
    assert codes.max() < 16
    for qi in range(3):
        lut = O.pq_lut(xq[qi], cb)
        for i in (0, 17, 400):
            recon = cen[a[i]] + np.concatenate([cb[m, codes[i, m]] for m in range(8)])
            want = float(xq[qi].astype(np.float64) @ recon.astype(np.float64))
            got = O.adc(lut, codes[i], O.ip(xq[qi], cen[a[i]]))
            assert abs(got - want) < 1e-5
    # encode picks the nearest codeword of the residual
    r = xb[5] - cen[a[5]]
    for m in range(8):
        d2 = ((r[m * 100:(m + 1) * 100][None] - cb[m]) ** 2).sum(1)
        assert codes[5, m] == int(np.argmin(d2)) or np.isclose(d2[codes[5, m]], d2.min(), rtol=1e-5)


def test_sharding_and_merge_invariance(O, vecs):
    xb, xq = vecs
    cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
    a = O.assign(xb, cen, 0)
    cb = O.pq_train(xb, cen, 8, 16, 4, 1241)
    codes = O.pq_encode(xb, cen, a, cb)
    full = O.HostIVF(cen, a, codes, cb)
    D, I = full.search(xq, 50, 4)
    parts = []
    for s in range(3):
        sub = O.HostIVF.__new__(O.HostIVF)
        sub.centroids, sub.nlist, sub.d, sub.codebooks, sub.kind = cen, 8, 800, cb, 1
        keep = (a[full.ids] % 3) == s
        sub.ids, sub.payload = full.ids[keep].copy(), full.payload[keep].copy()
        cnt = np.bincount(a[sub.ids], minlength=8)
        sub.list_offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        parts.append(sub.search(xq, 50, 4))
    Dm, Im = O.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_kmeans_is_deterministic_and_finite(O, vecs):
    xb, _ = vecs
    c1 = O.kmeans(xb, 16, 5, 7, 0, 256)
    c2 = O.kmeans(xb, 16, 5, 7, 0, 256)
    assert np.array_equal(c1, c2) and np.isfinite(c1).all()
    assert not np.array_equal(c1, O.kmeans(xb, 16, 5, 8, 0, 256))
    p = O.rand_perm(1000, 3)
    assert sorted(p.tolist()) == list(range(1000))


def test_ip_kmeans_is_spherical_l2_is_not(O, vecs):
    """FAISS' IndexIVF sets cp.spherical = true for METRIC_INNER_PRODUCT (the reference's index,
    spectral_library.py:174-178): centroids are L2-renormalised after every iteration. The PQ
    sub-quantisers train with L2 and stay plain means."""
    xb, _ = vecs
    c = O.kmeans(xb, 16, 5, 7, 0, 256)                       # metric 0 = inner product
    assert np.allclose(np.linalg.norm(c.astype(np.float64), axis=1), 1.0, atol=2e-7)
    # a centroid is the renormalised mean of its cluster (last iteration's assignment)
    prev = O.kmeans(xb, 16, 4, 7, 0, 256)
    a = np.argmax(xb @ prev.T, axis=1)
    for j in range(16):
        m = xb[a == j].astype(np.float64).mean(0)
        if (a == j).any():
            assert np.allclose(c[j], m / np.linalg.norm(m), atol=1e-5)
    sub = np.ascontiguousarray(xb[:, :25])
    c2 = O.kmeans(sub, 16, 5, 7, 1, 0)                       # metric 1 = L2: means, not unit vectors
    assert np.abs(np.linalg.norm(c2, axis=1) - 1).max() > 1e-3


def test_precursor_window_formulae(O):
    """spectral_library.py:421-427 in float64 on (float64 query, float32 library)."""
    rng = np.random.default_rng(0)
    for _ in range(2000):
        q = rng.uniform(300, 1500)
        l = np.float32(q + rng.normal(0, 0.02))
        z = int(rng.integers(1, 5))
        assert O.precursor_ok(q, l, z, 0.05, 'Da') == bool(abs(q - np.float64(l)) * z <= 0.05)
        assert O.precursor_ok(q, l, z, 20, 'ppm') == bool(
            abs(q - np.float64(l)) / np.float64(l) * 10 ** 6 <= 20)
