"""CPU checks of the oracle's process_spectrum restatement (oracle/asl_oracle.c,
reference spectrum.py:57-119 over spectrum_utils 0.3.x -- un-vendored, PARITY UNPINNED):
against an independent numpy formulation written the way spectrum_utils' array code reads
(boolean masks, argsort), and against the behaviour the reference documents."""
import numpy as np
import pytest


def _numpy_process(mz, it, pmz, pz, min_mz=11, max_mz=2010, remove_precursor=False, rp_tol=0.0,
                   min_intensity=0.01, max_peaks=50, scaling='rank', min_peaks=10,
                   min_mz_range=250.0, resolution=None):
    def valid(m):                                            # spectrum.py:13-36
        return len(m) >= min_peaks and len(m) > 0 and float(m[-1] - m[0]) >= min_mz_range
    idx = np.arange(len(mz))
    keep = (mz.astype(np.float64) >= min_mz) & (mz.astype(np.float64) <= max_mz)
    mz, it, idx = mz[keep], it[keep], idx[keep]
    if not valid(mz):
        return False, None, None, None
    if resolution is not None:                               # spectrum.py:84-85 round(d, 'sum')
        rmz = np.round(mz.astype(np.float64), resolution).astype(np.float32)
        uniq, first, inv = np.unique(rmz, return_index=True, return_inverse=True)
        it_sum = np.zeros(len(uniq), np.float32)
        best = first.copy()
        for j in range(len(rmz)):                            # sequential float32 sum, first argmax
            g = inv[j]
            it_sum[g] = np.float32(it_sum[g] + it[j])
            if it[j] > it[best[g]]:
                best[g] = j
        mz, it, idx = uniq, it_sum, idx[best]
        if not valid(mz):
            return False, None, None, None
    if remove_precursor:
        neutral = (pmz - 1.0072766) * pz
        rm = np.array([(neutral + iso) / c + 1.0072766 for c in range(pz, 0, -1)
                       for iso in range(3)])
        far = (np.abs(mz.astype(np.float64)[:, None] - rm[None, :]) > rp_tol).all(1)
        mz, it, idx = mz[far], it[far], idx[far]
        if not valid(mz):
            return False, None, None, None
    order = np.argsort(it, kind='stable')[::-1]              # descending; ties: later peak first
    top = order[:max_peaks]
    top = top[it[top].astype(np.float64) > min_intensity * float(it.max())]
    sel = np.sort(top)
    if not valid(mz[sel]):
        return False, None, None, None
    if scaling == 'rank':
        val = np.zeros(len(mz), np.float32)
        val[top] = max_peaks - np.arange(len(top), dtype=np.float32)
        val = val[sel]
    elif scaling == 'sqrt':
        val = np.sqrt(it[sel])
    else:
        val = it[sel].copy()
    acc = np.float32(0)
    for v in val:                                            # canonical ascending fp32 chain
        acc = np.float32(np.float64(v) * np.float64(v) + np.float64(acc))
    return True, mz[sel], val / np.sqrt(acc), idx[sel]


@pytest.mark.parametrize('cfg', [
    dict(),
    dict(scaling='sqrt', max_peaks=30),
    dict(remove_precursor=True, rp_tol=1.5, min_intensity=0.05),
    dict(scaling=None, min_peaks=3, min_mz_range=10.0, max_peaks=150),
    dict(resolution=0),
    dict(resolution=1, remove_precursor=True, rp_tol=1.5, scaling='sqrt'),
    dict(resolution=-1, scaling=None, max_peaks=80),
    dict(resolution=3),
])
def test_oracle_process_matches_numpy_formulation(O, cfg):
    rng = np.random.default_rng(11)
    n_valid = 0
    for s in range(120):
        n = int(rng.choice([0, 3, 9, 12, 40, 120, 300, 700]))
        mz = np.sort(rng.uniform(5, 2100, n)).astype(np.float32)
        it = rng.lognormal(0, 1.5, n).astype(np.float32)
        if s % 4 == 0 and n > 4:
            it[rng.integers(0, n, 3)] = it[0]
        pmz, pz = float(rng.uniform(300, 1200)), int(rng.integers(1, 5))
        ok, om, oi, src = O.process_spectrum(
            mz, it, pmz, pz, 11, 2010, cfg.get('remove_precursor', False), cfg.get('rp_tol', 0.0),
            cfg.get('min_intensity', 0.01), cfg.get('max_peaks', 50), cfg.get('scaling', 'rank'),
            cfg.get('min_peaks', 10), cfg.get('min_mz_range', 250.0), cfg.get('resolution'))
        ok2, m2, i2, s2 = _numpy_process(mz, it, pmz, pz, **cfg)
        assert ok == ok2, s
        if not ok:
            continue
        n_valid += 1
        assert np.array_equal(om, m2) and np.array_equal(src, s2)
        # the fmaf chain rounds once per step, the numpy emulation twice at most: 1 ulp
        assert np.allclose(oi, i2, rtol=3e-7, atol=0)
    assert n_valid > 20


def test_oracle_process_documented_behaviour(O):
    mz = np.concatenate([[10.9, 11.0], np.linspace(100, 1900, 70), [2010.0, 2010.1]]).astype(np.float32)
    it = np.concatenate([[500, 400], np.arange(1, 71), [300, 999]]).astype(np.float32)
    it[5] = 0.0001
    ok, om, oi, src = O.process_spectrum(mz, it, 700.0, 2)
    assert ok and len(om) == 50
    assert om[0] == np.float32(11.0) and om[-1] == np.float32(2010.0)      # inclusive window
    assert abs(np.linalg.norm(oi) - 1) < 1e-6
    assert om[np.argmax(oi)] == np.float32(11.0)                          # in-range base peak
    ranks = np.round(oi / oi.min()).astype(int)
    assert sorted(ranks.tolist()) == list(range(1, 51))                   # rank scaling
    assert (np.diff(src) > 0).all()                                       # m/z order preserved
    # too few peaks / too narrow a range -> invalid
    assert not O.process_spectrum(mz[:6], it[:6], 700.0, 2)[0]
    # resolution: peaks that round to the same m/z are merged, intensities summed, the
    # annotation (source index) of the most intense one survives (spectrum.py:84-85)
    mz3 = np.sort(np.concatenate([np.linspace(100, 1900, 30), [500.2, 500.4, 499.6]])).astype(np.float32)
    it3 = np.ones(len(mz3), np.float32)
    it3[np.searchsorted(mz3, np.float32(500.2))] = 7.0
    ok, om, oi, src = O.process_spectrum(mz3, it3, 700.0, 2, scaling=None, resolution=0)
    assert ok and (om == np.round(om)).all() and len(np.unique(om)) == len(om)
    at = int(np.nonzero(om == 500.0)[0][0])
    assert src[at] == np.searchsorted(mz3, np.float32(500.2))
    assert np.isclose(oi[at] / oi.min(), 9.0)                             # 1 + 7 + 1
    narrow = np.linspace(500, 600, 40).astype(np.float32)
    assert not O.process_spectrum(narrow, np.ones(40, np.float32), 700.0, 2)[0]
    # precursor removal: peaks within tol of (M + iso)/c + proton for c = z..1, iso = 0..2
    pm = 500.0
    mz2 = np.sort(np.concatenate([np.linspace(100, 1900, 40),
                                  [pm, pm + 0.5, (pm - 1.0072766) * 2 + 1.0072766]])).astype(np.float32)
    ok, om, _, _ = O.process_spectrum(mz2, np.ones(len(mz2), np.float32), pm, 2, 11, 2010, True, 0.1)
    assert ok and len(om) == 40
