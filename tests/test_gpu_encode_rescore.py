"""GPU parity: HIP encoder and rescoring kernels (through the C ABI) vs the oracle
and vs the committed golden vectors of the reference."""
import numpy as np
import torch
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def data():
    from ann_solo_amd import synthetic
    lib, aux = synthetic.make_library(3000, seed=21, device='cpu')
    q, truth = synthetic.make_queries(lib, aux, 300, seed=22)
    return lib, q, truth


def test_encoder_bit_exact_vs_oracle(O, data):
    from ann_solo_amd import spectrum
    lib, q, _ = data
    for pack in (lib, q):
        o, mz, inten, *_ = pack.numpy()
        for bin_size, hl, norm in ((0.04, 800, True), (0.04, 800, False), (0.05, 64, True),
                                   (1.0005, 400, True)):
            mb = O.get_dim(11, 2010, bin_size)[1]
            got = spectrum.spectra_to_vectors(mz, inten, o, 11, 2010, bin_size, hl, norm)
            want = O.encode_batch(mz, inten, o, mb, bin_size, hl, 42, norm)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_encoder_vs_reference_golden(O, golden):
    from ann_solo_amd import spectrum
    g = golden('encoder_golden.npz')
    got = spectrum.spectra_to_vectors(g['mz'], g['intensity'], g['offsets'], 11, 2010, 0.04, 800,
                                      False)
    assert np.array_equal(got, g['vec_nonorm'])          # bit-exact bins and fp32 adds
    got = spectrum.spectra_to_vectors(g['mz'], g['intensity'], g['offsets'], 11, 2010, 0.04, 800)
    np.testing.assert_allclose(got, g['vec'], rtol=4e-7, atol=0)   # BLAS-order norm: few ulp
    got = spectrum.spectra_to_vectors(g['mz'], g['intensity'], g['offsets'], 11, 2010, 0.05, 64)
    np.testing.assert_allclose(got, g['vec_h64'], rtol=4e-7, atol=0)
    assert [spectrum.hash_idx(int(b), 800) for b in g['bins']] == g['hashes'].tolist()
    for args, want in zip(g['dim_args'], g['dims']):
        assert spectrum.get_dim(*args) == (int(want[0]), want[1], want[2])


def test_encoder_device_pointers_and_empty(O, data):
    import torch
    from ann_solo_amd import spectrum
    lib, q, _ = data
    qd = q.to('cuda')
    out = torch.empty((q.n, 800), dtype=torch.float32, device='cuda')
    spectrum.spectra_to_vectors(qd.mz, qd.intensity, qd.offsets, 11, 2010, 0.04, 800, True, out)
    torch.cuda.synchronize()
    o, mz, inten, *_ = q.numpy()
    want = O.encode_batch(mz, inten, o, 10.96, 0.04, 800)
    assert np.array_equal(out.cpu().numpy(), want)
    # ragged: empty spectrum in the middle, single spectrum, zero spectra
    mz2 = np.array([100.0, 200.0, 300.5], np.float32)
    in2 = np.array([1.0, 2.0, 3.0], np.float32)
    off2 = np.array([0, 2, 2, 3], np.int32)
    got = spectrum.spectra_to_vectors(mz2, in2, off2, 11, 2010, 0.04, 800, False)
    want = O.encode_batch(mz2, in2, off2, 10.96, 0.04, 800, 42, False)
    assert np.array_equal(got, want) and not got[1].any()
    assert spectrum.spectra_to_vectors(mz2[:0], in2[:0], np.zeros(1, np.int32), 11, 2010, 0.04,
                                       800).shape == (0, 800)


def test_encoder_colliding_bins_and_entry_lists(O):
    """Peaks that share a hash bin are added in peak order (the kernel applies them in rounds by
    their rank within the bin): spectra built to collide -- runs of equal m/z, up to 200 peaks, ties
    across the 64-peak chunks -- equal the oracle's serial loop bit for bit; and
    asl_encode_entries_batch emits exactly the non-zero components of those rows, ascending, as
    (dimension * 128, value bits), the count negative beyond 64 entries."""
    import torch
    from ann_solo_amd import _lib, spectrum
    rng = np.random.default_rng(12)
    mzs, ins, off = [], [], [0]
    for npk in [0, 1, 2, 50, 63, 64, 65, 70, 128, 129, 200] + list(rng.integers(1, 90, 60)):
        # ~3 peaks per distinct m/z (collisions); the long ones mostly distinct (more than 64 non-zeros)
        base = rng.uniform(100, 1900, max(1, npk if npk > 100 else npk // 3 + 1)).astype(np.float32)
        m = base[rng.integers(0, len(base), npk)]
        # intensities of very different magnitude: the order of the additions shows in the last bit
        i = (rng.random(npk) * 10.0 ** rng.integers(-6, 4, npk)).astype(np.float32)
        mzs.append(m)
        ins.append(i)
        off.append(off[-1] + npk)
    # bins around and beyond the limits of the encoder's short routes (decimal strings of up to nine
    # digits in 32-bit words; quotients below 2^31 by division + fused remainder): 10^9 - 1 | 10^9,
    # 2^31 - 1 | 2^31, far beyond, below the first bin (negative indices), exactly on bin edges
    edge = np.array([10.96 + 0.04 * 999999999.0, 10.96 + 0.04 * 1000000000.0, 10.96 + 0.04 * 2147483647.0,
                     10.96 + 0.04 * 2147483648.0, 5e7, 1e8, 1e12, 1e15, 0.0, 5.0, 10.96, 10.959999, 11.0,
                     10.96 + 0.04 * 7, 10.96 + 0.04 * 12345], np.float64).astype(np.float32)
    mzs.append(edge)
    ins.append(np.arange(1, len(edge) + 1, dtype=np.float32))
    off.append(off[-1] + len(edge))
    mz, inten, off = np.concatenate(mzs).astype(np.float32), np.concatenate(ins).astype(np.float32), np.asarray(off, np.int32)
    n = len(off) - 1
    for norm in (True, False):
        want = O.encode_batch(mz, inten, off, 10.96, 0.04, 800, 42, norm)
        got = spectrum.spectra_to_vectors(mz, inten, off, 11, 2010, 0.04, 800, norm)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ent = torch.full((n, 64, 2), -1, dtype=torch.int32, device='cuda')
        cnt = torch.full((n,), -99, dtype=torch.int32, device='cuda')
        over = torch.zeros(1, dtype=torch.int32, device='cuda')
        dm, di, do = (torch.from_numpy(a).cuda() for a in (mz, inten, off))
        _lib.check(_lib.lib().asl_encode_entries_batch(_lib.ptr(dm), _lib.ptr(di), _lib.ptr(do), n, len(mz), 10.96,
                                                       0.04, 800, 42, int(norm), _lib.ptr(ent), _lib.ptr(cnt),
                                                       _lib.ptr(over)))
        ent, cnt = ent.cpu().numpy(), cnt.cpu().numpy()
        n_over = 0
        for q in range(n):
            nz = np.nonzero(want[q])[0]
            if len(nz) > 64:
                assert cnt[q] == -1 - len(nz)
                n_over += 1
                nz = nz[:64]
            else:
                assert cnt[q] == len(nz)
            assert np.array_equal(ent[q, :len(nz), 0], nz * 128)
            assert np.array_equal(ent[q, :len(nz), 1], want[q, nz].view(np.int32))
            assert not ent[q, len(nz):].any()
        assert n_over >= 2 and int(over.item()) == n_over
        # host pointers for the peaks, the peak count read back by the library
        ent2 = torch.empty((n, 64, 2), dtype=torch.int32, device='cuda')
        cnt2 = torch.empty((n,), dtype=torch.int32, device='cuda')
        _lib.check(_lib.lib().asl_encode_entries_batch(_lib.ptr(mz), _lib.ptr(inten), _lib.ptr(off), n, -1, 10.96,
                                                       0.04, 800, 42, int(norm), _lib.ptr(ent2), _lib.ptr(cnt2), None))
        assert np.array_equal(ent2.cpu().numpy(), ent) and np.array_equal(cnt2.cpu().numpy(), cnt)


def _check_rescoring(O, Q, L, cand, off, tol, shift, res):
    best, score, count, pairs = res
    for qi in range(Q.n):
        c = cand[off[qi]:off[qi + 1]]
        b, s, m = O.best_match(Q, qi, L, c, tol, shift)
        assert best[qi] == b, (qi, best[qi], b)
        if b < 0:
            continue
        assert score[qi] == s                              # same doubles, same order: exact
        assert count[qi] == len(m)
        assert pairs[qi, :len(m)].tolist() == m.tolist()   # same deterministic tie rule


def test_rescoring_vs_oracle(O, data):
    from ann_solo_amd import spectrum_match
    lib, q, truth = data
    L, Q = O.Spectra(*lib.numpy()), O.Spectra(*q.numpy())
    rng = np.random.default_rng(0)
    cands, off = [], [0]
    for qi in range(q.n):
        n = int(rng.integers(0, 60))
        c = np.unique(np.concatenate([rng.integers(0, lib.n, n),
                                      [int(truth['source_row'][qi])] if qi % 5 else []]))
        cands.append(c.astype(np.int64))
        off.append(off[-1] + len(c))
    cand = np.concatenate(cands)
    off = np.array(off, np.int32)
    for tol, shift in ((0.02, True), (0.02, False), (0.05, True)):
        res = spectrum_match.rescore_batch(q, lib, cand, off, tol, shift)
        _check_rescoring(O, Q, L, cand, off, tol, shift, res)


def test_rescoring_few_queries_long_lists(O, data):
    """Few queries with very long candidate lists: the kernels run several blocks per query
    (each block takes every n-th group of 32 slots; results must not depend on how the blocks
    order their compacted lists) -- and a query whose two strongest peaks are closer than the
    tolerance, which makes nearly every candidate a doubly-matched one (pair kernel, up to 8
    blocks per query in the second launch)."""
    from ann_solo_amd import spectrum_match, synthetic
    from ann_solo_amd.packed import PackedSpectra
    lib, aux = synthetic.make_library(9000, seed=23, device='cpu')
    q, truth = synthetic.make_queries(lib, aux, 8, seed=24)
    rng = np.random.default_rng(3)
    nq = 3
    o, mz, it, chg, pmz, pz = q.select(torch.arange(nq)).numpy()
    mz, it = mz.copy(), it.copy()
    # query 0: plant a twin 0.01 Da above its most intense peak
    a, b = o[0], o[1]
    top = a + int(np.argmax(it[a:b]))
    other = a + int(np.argmin(it[a:b]))
    mz[other] = mz[top] + np.float32(0.01)
    it[other] = it[top] * np.float32(0.9)
    order = np.argsort(mz[a:b], kind='stable')
    mz[a:b], it[a:b], chg[a:b] = mz[a:b][order], it[a:b][order], chg[a:b][order]
    qq = PackedSpectra.from_numpy(o, mz, it, chg, pmz, pz)
    L, Q = O.Spectra(*lib.numpy()), O.Spectra(*qq.numpy())
    cands = [np.sort(rng.choice(lib.n, size=8500, replace=False)).astype(np.int64)    # > 4096: ysplit
             for _ in range(nq)]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int32)
    cand = np.concatenate(cands)
    for tol, shift in ((0.02, True), (0.02, False)):
        res = spectrum_match.rescore_batch(qq, lib, cand, off, tol, shift)
        _check_rescoring(O, Q, L, cand, off, tol, shift, res)


def test_peak_matches_of_long_spectra_and_many_matches(O):
    """The winner's peak matches are emitted by a kernel built for spectra of <= 128 peaks and <= 128
    generated matches (small LDS: many waves in flight); anything beyond is left to the full-size
    instantiation that runs behind it. Queries and library spectra of 20 .. 250 peaks on a coarse
    m/z grid with a wide tolerance (many candidates per window, hundreds of generated matches):
    winners, scores, counts and the match pairs equal the oracle's."""
    from ann_solo_amd import spectrum_match
    from ann_solo_amd.packed import PackedSpectra
    rng = np.random.default_rng(77)

    def spectra(n, sizes, charge):
        offs, mzs, its, chs = [0], [], [], []
        for i in range(n):
            m = int(sizes[i % len(sizes)])
            g = np.sort(rng.choice(np.arange(200, 1800), size=m, replace=False)).astype(np.float32)
            mzs.append(g + rng.normal(0, 0.004, m).astype(np.float32))
            its.append(rng.random(m).astype(np.float32) + np.float32(0.01))
            chs.append(rng.integers(0, 3, m).astype(np.uint8))
            offs.append(offs[-1] + m)
        pmz = rng.uniform(400, 900, n)
        pz = np.full(n, charge, np.int32)
        order = [np.argsort(x, kind='stable') for x in mzs]
        mzs = [x[o_] for x, o_ in zip(mzs, order)]
        its = [x[o_] for x, o_ in zip(its, order)]
        chs = [x[o_] for x, o_ in zip(chs, order)]
        return PackedSpectra.from_numpy(np.asarray(offs, np.int32), np.concatenate(mzs), np.concatenate(its),
                                        np.concatenate(chs), pmz, pz)
    lib = spectra(60, [20, 50, 100, 128, 129, 200, 250], 2)
    qq = spectra(24, [30, 127, 128, 129, 250, 60], 2)
    L, Q = O.Spectra(*lib.numpy()), O.Spectra(*qq.numpy())
    cands = [np.sort(rng.choice(lib.n, size=40, replace=False)).astype(np.int64) for _ in range(qq.n)]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int32)
    cand = np.concatenate(cands)
    for tol, shift in ((0.02, True), (0.4, True), (0.4, False)):
        res = spectrum_match.rescore_batch(qq, lib, cand, off, tol, shift)
        _check_rescoring(O, Q, L, cand, off, tol, shift, res)
        assert res[2].max() > 40        # long match lists were produced


def test_rescoring_many_hits_per_chunk(O):
    """A library of copies and near-copies of one spectrum, queried with it: every chunk of 32
    candidates holds > 1 000 (peak, shift) items whose bin is marked, so the per-wave queue of the
    flat kernel (128 items) drains many times inside the peak stream; copies with a peak moved by
    half a tolerance give doubly matched peaks (second launch) beside the plain ones."""
    from ann_solo_amd import spectrum_match, synthetic
    from ann_solo_amd.packed import PackedSpectra
    lib, aux = synthetic.make_library(400, seed=31, device='cpu')
    o, mz, it, chg, pmz, pz = lib.numpy()
    a, b = int(o[7]), int(o[8])
    n = b - a
    rng = np.random.default_rng(5)
    copies = 300
    O2 = np.arange(copies + 1, dtype=np.int32) * n
    MZ, IT, CH = np.tile(mz[a:b], copies), np.tile(it[a:b], copies), np.tile(chg[a:b], copies)
    for c in range(0, copies, 3):           # every third copy: one peak moved next to its neighbour
        j = int(rng.integers(1, n))
        MZ[c * n + j] = MZ[c * n + j - 1] + np.float32(0.008)
        seg = slice(c * n, (c + 1) * n)
        order = np.argsort(MZ[seg], kind='stable')
        MZ[seg], IT[seg], CH[seg] = MZ[seg][order], IT[seg][order], CH[seg][order]
    PM = np.full(copies, pmz[7]) + rng.normal(0, 30.0, copies)      # open-search mass differences: shifts on
    PZ = np.full(copies, pz[7], np.int32)
    L2 = PackedSpectra.from_numpy(O2, MZ, IT, CH, PM, PZ)
    q = lib.select(torch.tensor([7, 7, 8]))
    L, Q = O.Spectra(*L2.numpy()), O.Spectra(*q.numpy())
    cand = np.concatenate([np.arange(copies), np.arange(copies)[::-1], np.arange(0, copies, 2)]).astype(np.int64)
    off = np.array([0, copies, 2 * copies, 2 * copies + len(range(0, copies, 2))], np.int32)
    for tol, shift in ((0.02, True), (0.02, False), (0.05, True)):
        res = spectrum_match.rescore_batch(q, L2, cand, off, tol, shift)
        _check_rescoring(O, Q, L, cand, off, tol, shift, res)


def test_rescoring_vs_reference_golden(O, golden):
    """Every case of tests/golden/rescoring_golden.npz (outputs of the reference's own
    SpectrumMatch.cpp): best index, score within 1e-12, peak matches as a set."""
    from ann_solo_amd import spectrum_match
    from ann_solo_amd.packed import PackedSpectra
    g = golden('rescoring_golden.npz')
    lib = PackedSpectra.from_numpy(g['lib_offsets'], g['lib_mz'], g['lib_intensity'],
                                   g['lib_charge'], g['lib_pmz'], g['lib_pcharge'])
    n_case = len(g['case_query'])
    for tol, shift in ((0.02, 1), (0.02, 0), (0.05, 1)):
        sel = np.nonzero((g['case_tol'] == tol) & (g['case_shift'] == shift))[0]
        qrows = g['case_query'][sel]
        qo = g['q_offsets']
        cnt = (qo[qrows + 1] - qo[qrows])
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        idx = np.concatenate([np.arange(qo[r], qo[r + 1]) for r in qrows])
        q = PackedSpectra.from_numpy(off, g['q_mz'][idx], g['q_intensity'][idx], None,
                                     g['q_pmz'][qrows], g['q_pcharge'][qrows])
        coff = np.concatenate([[0], np.cumsum(g['cand_offsets'][sel + 1] - g['cand_offsets'][sel])])
        cand = np.concatenate([g['cand_rows'][g['cand_offsets'][c]:g['cand_offsets'][c + 1]]
                               for c in sel])
        best, score, count, pairs = spectrum_match.rescore_batch(q, lib, cand,
                                                                 coff.astype(np.int32), tol, shift)
        for i, c in enumerate(sel):
            assert best[i] == g['case_best'][c]
            assert abs(score[i] - g['case_score'][c]) <= 1e-12
            want = g['pm_pairs'][g['pm_offsets'][c]:g['pm_offsets'][c + 1]]
            assert sorted(map(tuple, pairs[i, :count[i]].tolist())) == sorted(map(tuple, want.tolist()))
    assert n_case > 0


def test_get_best_match_dropin(O, golden):
    """Reference-style call with spectrum objects, incl. the reference test-suite's
    partial_match pair (spectrum_similarity_test.py:339-341,443)."""
    from ann_solo_amd import spectrum_match
    k = golden('similarity_kat.npz')

    class Spec:
        def __init__(self, mz, inten, pmz, z):
            self.mz, self.intensity, self.precursor_mz, self.precursor_charge = mz, inten, pmz, z
            self.annotation = [None] * len(mz)
    q = Spec(k['partial_match_q_mz'], k['partial_match_q_intensity'], 453.75, 2)
    c0 = Spec(k['all_match_l_mz'], k['all_match_l_intensity'], 453.75, 2)
    c1 = Spec(k['partial_match_l_mz'], k['partial_match_l_intensity'], 453.75, 2)
    cand, score, pm = spectrum_match.get_best_match(q, [c0, c1], 0.02, True)
    assert cand is c1
    assert sorted(pm) == sorted(map(tuple, k['partial_match_peak_matches'].tolist()))
    assert score == pytest.approx(0.44582117, abs=1e-7)
    with pytest.raises(ValueError):
        spectrum_match.get_best_match(q, [], 0.02, True)
