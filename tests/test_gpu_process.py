"""GPU parity of the batched process_spectrum kernel against the oracle's restatement
(PARITY UNPINNED towards spectrum_utils itself) plus semantic checks of the
reference's documented behaviour (spectrum.py:57-119, config.py:71-121)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _raw_pack(rng, n_spec, sizes):
    from ann_solo_amd.packed import PackedSpectra
    offs, mzs, its, pmz, pz = [0], [], [], [], []
    for s in range(n_spec):
        n = int(rng.choice(sizes))
        mz = np.sort(rng.uniform(5, 2100, n)).astype(np.float32)
        it = rng.lognormal(0, 1.5, n).astype(np.float32)
        if s % 5 == 0 and n > 4:
            it[rng.integers(0, n, 3)] = it[0]          # intensity ties
        mzs.append(mz)
        its.append(it)
        offs.append(offs[-1] + n)
        pmz.append(float(rng.uniform(300, 1200)))
        pz.append(int(rng.integers(1, 5)))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.float32)
    chg = rng.integers(0, 4, offs[-1]).astype(np.uint8)
    return PackedSpectra.from_numpy(np.array(offs), cat(mzs), cat(its), chg, np.array(pmz),
                                    np.array(pz))


@pytest.mark.parametrize('cfg', [
    dict(),                                                             # reference defaults
    dict(scaling='sqrt', max_peaks_used=30),
    dict(remove_precursor=True, remove_precursor_tolerance=1.5, min_intensity=0.05),
    dict(scaling=None, min_peaks=3, min_mz_range=10.0, max_peaks_used=150),
    dict(resolution=0),                                                 # spectrum.py:84-85
    dict(resolution=1, remove_precursor=True, remove_precursor_tolerance=1.5, scaling='sqrt'),
    dict(resolution=-1, scaling=None, max_peaks_used=80),
    dict(resolution=3),
])
def test_process_matches_oracle(O, cfg):
    from types import SimpleNamespace
    from ann_solo_amd import spectrum
    rng = np.random.default_rng(7)
    raw = _raw_pack(rng, 160, [0, 3, 9, 12, 40, 120, 300, 700, 1500, 3000])
    out, valid = spectrum.process_spectra(raw, False, SimpleNamespace(**cfg))
    o, mz, it, chg, pmz, pz = raw.numpy()
    oo, omz, oit, ochg, _, _ = out.to('cpu').numpy()
    valid = valid.cpu().numpy()
    n_valid = 0
    for s in range(raw.n):
        sl = slice(o[s], o[s + 1])
        ok, rm, ri, rs = O.process_spectrum(
            mz[sl], it[sl], pmz[s], pz[s], 11, 2010, cfg.get('remove_precursor', False),
            cfg.get('remove_precursor_tolerance', 0.0), cfg.get('min_intensity', 0.01),
            cfg.get('max_peaks_used', 50), cfg.get('scaling', 'rank'), cfg.get('min_peaks', 10),
            cfg.get('min_mz_range', 250.0), cfg.get('resolution'))
        assert bool(valid[s]) == ok, s
        got = slice(oo[s], oo[s + 1])
        if not ok:
            assert oo[s + 1] == oo[s]
            continue
        n_valid += 1
        assert np.array_equal(omz[got], rm)
        assert np.array_equal(oit[got].view(np.uint32), ri.view(np.uint32))
        assert np.array_equal(ochg[got], chg[sl][rs])        # annotations follow their peaks
    assert 20 < n_valid < raw.n


def test_process_semantics():
    """Documented behaviour: m/z window inclusive, <= max_peaks most intense peaks strictly above
    1 % of the base peak, rank scaling (base peak -> max_rank), unit L2 norm, >= 10 peaks over
    >= 250 m/z or invalid."""
    from ann_solo_amd import spectrum
    from ann_solo_amd.packed import PackedSpectra
    mz = np.concatenate([[10.9, 11.0], np.linspace(100, 1900, 70), [2010.0, 2010.1]]).astype(np.float32)
    it = np.concatenate([[500, 400], np.arange(1, 71), [300, 999]]).astype(np.float32)
    it[5] = 0.0001                                   # below 1 % of the base peak (500 in range)
    raw = PackedSpectra.from_numpy([0, len(mz), len(mz) + 4], np.concatenate([mz, mz[:4]]),
                                   np.concatenate([it, it[:4]]), None, [700.0, 700.0], [2, 2])
    out, valid = spectrum.process_spectra(raw, False)
    assert valid.tolist() == [True, False]
    o, omz, oit, *_ = out.to('cpu').numpy()
    assert o.tolist() == [0, 50, 50]
    assert omz[0] == np.float32(11.0) and omz[-1] == np.float32(2010.0)   # inclusive bounds kept
    assert 10.9 not in omz and np.float32(2010.1) not in omz
    assert abs(np.linalg.norm(oit) - 1) < 1e-6
    base = np.argmax(oit)
    assert omz[base] == np.float32(11.0)             # intensity 400 is the in-range base peak
    ranks = np.round(oit / oit.min()).astype(int)    # rank scaling: integers 1..50 before the norm
    assert sorted(ranks.tolist()) == list(range(1, 51))
