"""GPU edge cases of the hot path: the rescoring kernel's deferred (slow) cases, degenerate
queries, and capacity errors -- each against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spectra(rng, n, sizes, z=2, dense=False, exact=False):
    from ann_solo_amd.packed import PackedSpectra
    offs, mzs, its, chg, pmz = [0], [], [], [], []
    for s in range(n):
        k = int(sizes[s]) if exact else int(rng.choice(sizes))
        if dense:      # many peaks inside a narrow window: long match lists at a wide tolerance
            mz = np.sort(rng.uniform(400, 1000, k)).astype(np.float32)
        else:
            mz = np.sort(rng.uniform(100, 1900, k)).astype(np.float32)
        it = rng.lognormal(0, 1, k).astype(np.float32)
        it /= max(np.linalg.norm(it), 1e-30)
        mzs.append(mz)
        its.append(it.astype(np.float32))
        chg.append(rng.integers(0, z + 1, k).astype(np.uint8))
        offs.append(offs[-1] + k)
        pmz.append(float(rng.uniform(400, 900)))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if offs[-1] else np.zeros(0, dt)
    return PackedSpectra.from_numpy(np.array(offs), cat(mzs, np.float32), cat(its, np.float32),
                                    cat(chg, np.uint8), np.array(pmz), np.full(n, z))


def _check(O, q, lib, cand, off, tol, shift):
    from ann_solo_amd import spectrum_match
    best, score, count, pairs = spectrum_match.rescore_batch(q, lib, cand, off, tol, shift)
    L, Q = O.Spectra(*lib.numpy()), O.Spectra(*q.numpy())
    n_match = 0
    for qi in range(q.n):
        c = cand[off[qi]:off[qi + 1]]
        pos = np.nonzero((c >= 0) & (c < lib.n))[0]       # the oracle takes valid rows only
        b, s, m = O.best_match(Q, qi, L, c[pos], tol, shift)
        b = int(pos[b]) if b >= 0 else -1
        assert best[qi] == b, (qi, best[qi], b)
        if b >= 0:
            assert score[qi] == s and count[qi] == len(m)
            assert pairs[qi, :len(m)].tolist() == m.tolist()
            n_match = max(n_match, len(m))
    return n_match


@pytest.mark.parametrize('tol,shift', [(0.02, True), (0.5, True), (0.5, False), (0.0, True)])
def test_deferred_rescoring_cases_match_oracle(O, tol, shift):
    """Candidates with > 64 peaks, queries with > 100 peaks, > 128 generated matches per pair and
    tol = 0 all leave the hash kernel for the binary-search kernel; empty spectra on either side."""
    rng = np.random.default_rng(5)
    lib = _spectra(rng, 120, [0, 1, 20, 50, 64, 65, 90, 150, 256], dense=tol > 0.1)
    q = _spectra(rng, 40, [0, 1, 30, 50, 100, 101, 180, 256], dense=tol > 0.1)
    cands, off = [], [0]
    for qi in range(q.n):
        c = np.sort(rng.choice(lib.n, int(rng.integers(0, 60)), replace=False)).astype(np.int64)
        if qi % 7 == 0:
            c = np.concatenate([c, [-1, lib.n + 5]])       # ignored slots
        cands.append(c)
        off.append(off[-1] + len(c))
    n_match = _check(O, q, lib, np.concatenate(cands), np.array(off, np.int32), tol, shift)
    if tol == 0.5:
        assert n_match > 64          # the long-match-list path was exercised


def _grouped_case(O, nq, marked, tol, shift, seed):
    """nq queries of 30 peaks except the MARKED ones (101 .. 200 peaks: left to the binary-search
    kernel; beyond 128 peaks also to the full-size matches kernel), 3 .. 8 candidates each."""
    rng = np.random.default_rng(seed)
    lib = _spectra(rng, 300, [20, 50, 64, 65, 90, 129, 150], dense=tol > 0.1)
    sizes = np.full(nq, 30)
    marked = np.asarray(sorted(set(int(m) for m in marked if 0 <= m < nq)), np.int64)
    sizes[marked] = rng.choice([101, 128, 129, 200], len(marked))
    q = _spectra(rng, nq, sizes, dense=tol > 0.1, exact=True)
    n_c = rng.integers(3, 9, nq)
    off = np.concatenate([[0], np.cumsum(n_c)]).astype(np.int32)
    cand = rng.integers(0, lib.n, int(off[-1])).astype(np.int64)
    _check(O, q, lib, cand, off, tol, shift)


@pytest.mark.parametrize('nq', [4095, 4096, 4097])
def test_grouped_deferred_rescoring_every_query_marked(O, nq):
    """From 4 096 queries on the binary-search kernel takes RS_BS_GROUP = 16 queries per workgroup
    and the full-size matches kernel 16 per wave (``rescore_device``, csrc/rescore.hip). tol = 0
    marks EVERY query for the binary-search kernel; every 3rd query has more than 100 peaks
    (half of those more than 128: full-size matches). Around the switch: 4 095 (one workgroup
    per query), 4 096 and 4 097 (grouped; a last group of one). Against ``SpectrumMatch.cpp:8-133``
    as the oracle restates it, query by query."""
    _grouped_case(O, nq, range(0, nq, 3), 0.0, True, 900 + nq)


@pytest.mark.parametrize('tol,shift', [(0.02, True), (0.5, False)])
def test_grouped_deferred_rescoring_sparse_marks(O, tol, shift):
    """1 query in 50 marked, plus marks at the group boundaries (15 / 16 / 17, 31 / 32, the first
    and the last query of the batch, the last -- partial -- group of 4 100 = 256 groups + 4 queries),
    two adjacent groups fully marked and one group with its last query alone."""
    nq = 4100
    marked = list(range(7, nq, 50)) + [0, 15, 16, 17, 31, 32, 63, 64, 4079, 4080, 4095, 4096, 4097, 4099] + \
        list(range(1024, 1056)) + [2063]
    _grouped_case(O, nq, marked, tol, shift, 77)


@pytest.mark.parametrize('z', [3, 4, 5, 7, 31])
def test_high_precursor_charges_match_oracle(O, z):
    """Shifted matching with many shifts (S = charge + 1): the flat kernel's mass-difference
    table holds five (charge <= 4; 3 and 5-shift tables use the exact fp64 division), charge 5+
    goes to the pair kernel, charge >= 31 to the binary-search kernel -- all equal the oracle."""
    rng = np.random.default_rng(40 + z)
    lib = _spectra(rng, 150, [12, 27, 40, 64], z=z)
    q = _spectra(rng, 60, [20, 50, 80], z=z)
    cands, off = [], [0]
    for qi in range(q.n):
        c = np.sort(rng.choice(lib.n, int(rng.integers(1, 80)), replace=False)).astype(np.int64)
        cands.append(c)
        off.append(off[-1] + len(c))
    for tol in (0.02, 0.3):
        _check(O, q, lib, np.concatenate(cands), np.array(off, np.int32), tol, True)


def test_more_than_256_peaks_is_a_capacity_error():
    from ann_solo_amd import _lib, spectrum_match
    rng = np.random.default_rng(6)
    lib = _spectra(rng, 4, [300])
    q = _spectra(rng, 2, [40])
    with pytest.raises(_lib.AnnSoloMiError, match='peaks'):
        spectrum_match.rescore_batch(q, lib, np.array([0, 1, 2, 3], np.int64),
                                     np.array([0, 2, 4], np.int32), 0.02, True)


def test_calls_under_a_side_stream_are_refused():
    """The library issues on the null stream: under a non-default current PyTorch stream its
    inputs could still be in flight, so every device tensor handed over is refused there
    (``_lib._require_default_stream``) -- and accepted again once the default stream is current."""
    import torch
    from ann_solo_amd import _lib, spectrum
    rng = np.random.default_rng(8)
    q = _spectra(rng, 8, [30]).to('cuda:0')
    out = torch.zeros((8, 800), dtype=torch.float32, device='cuda:0')
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with pytest.raises(_lib.AnnSoloMiError, match='default HIP stream'):
            spectrum.spectra_to_vectors(q.mz, q.intensity, q.offsets, 11, 2010, 0.04, 800, out=out)
    side.synchronize()
    v = spectrum.spectra_to_vectors(q.mz, q.intensity, q.offsets, 11, 2010, 0.04, 800, out=out)
    assert v.shape == (8, 800) and bool(torch.isfinite(v).all()) and float(v.abs().sum()) > 0


def test_degenerate_queries_through_the_fused_search(O):
    """Empty query spectrum (zero vector: every coarse score ties), a query whose window holds
    no library spectrum, duplicate library spectra (tie -> lowest row), k larger than the
    partition, nprobe = nlist."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.packed import PackedSpectra
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(1500, seed=61, device='cpu', charges=(2,), charge_p=(1.0,))
    o, mz, it, chg, pmz, pz = lib.numpy()
    # duplicate spectrum 7 as the last row
    dup = slice(o[7], o[8])
    lib = PackedSpectra.from_numpy(np.concatenate([o, [o[-1] + o[8] - o[7]]]),
                                   np.concatenate([mz, mz[dup]]), np.concatenate([it, it[dup]]),
                                   np.concatenate([chg, chg[dup]]), np.concatenate([pmz, [pmz[7]]]),
                                   np.concatenate([pz, [2]]))
    q, _ = synthetic.make_queries(lib, aux, 48, seed=62, charge=2)
    qo, qmz, qit, qchg, qpmz, qpz = q.numpy()
    # query 0 := exact copy of library spectrum 7; query 1 := empty; query 2 := far precursor
    parts = [(mz[dup], it[dup], chg[dup])] + [(qmz[:0], qit[:0], qchg[:0])] + \
            [(qmz[qo[i]:qo[i + 1]], qit[qo[i]:qo[i + 1]], qchg[qo[i]:qo[i + 1]]) for i in range(2, q.n)]
    qpmz = qpmz.copy()
    qpmz[0], qpmz[2] = pmz[7], 5000.0
    q = PackedSpectra.from_numpy(np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]),
                                 np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                                 np.concatenate([p[2] for p in parts]), qpmz, qpz)
    for index in ('ivfpq', 'ivfflat'):
        cfg = Config.open_search(num_list=16, num_probe=16, num_candidates=2048, index=index, kmeans_niter=4,
                     precursor_tolerance_mass_open=300, precursor_tolerance_mode_open='Da')
        sl = SpectralLibrary(lib, config=cfg)
        res = sl._search_batch(q, 2, 'open', want_knn=True)
        assert res.best_row[0] == 7 and res.best_score[0] == pytest.approx(1.0, abs=1e-6)
        assert res.best_row[1] == -1 or res.best_score[1] == 0.0     # empty query: nothing matches
        assert res.best_row[2] == -1 and res.n_candidates[2] == 0    # empty precursor window
        # every query against the oracle on the returned neighbour lists
        L, Q = O.Spectra(*lib.numpy()), O.Spectra(*q.numpy())
        pmz32 = lib.precursor_mz.numpy().astype(np.float32)
        for i in range(q.n):
            knn = res.knn[i]
            cand = np.sort(np.array([r for r in knn if r >= 0 and O.precursor_ok(
                q.precursor_mz[i].item(), pmz32[r], 2, 300, 'Da')], np.int64))
            assert res.n_candidates[i] == len(cand)
            b, s, _ = O.best_match(Q, i, L, cand, 0.02, True)
            if b < 0:
                assert res.best_row[i] == -1
            else:
                assert res.best_row[i] == cand[b] and res.best_score[i] == s
        sl.shutdown()
