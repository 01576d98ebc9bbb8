"""Randomised parity run of the front of the path: batched process_spectrum (all options, incl.
resolution rounding and precursor-peak removal) and the feature-hashing encoder against the
oracle, on random raw spectra (empty, tiny, > 64 peaks, intensity ties, duplicate and out-of-range
m/z).   python scripts/fuzz_pre.py [seconds] [seed]       (test infrastructure: uses oracle/)"""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import spectrum
from ann_solo_amd.packed import PackedSpectra
from oracle import oracle_py as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
t_end = time.time() + budget
trials = bad = n_spec = 0
while time.time() < t_end:
    cfg = dict(min_mz=float(rng.choice([11, 50, 200])), max_mz=float(rng.choice([2010, 1500, 900])),
               resolution=[None, None, 0, 1, 2, 3, -1][int(rng.integers(0, 7))],
               remove_precursor=bool(rng.random() < 0.4),
               remove_precursor_tolerance=float(rng.choice([0.0, 0.05, 1.5])),
               min_intensity=float(rng.choice([0.0, 0.01, 0.05, 0.5])),
               max_peaks_used=int(rng.choice([5, 30, 50, 150, 256])),
               scaling=[None, 'rank', 'sqrt', 'root'][int(rng.integers(0, 4))],
               min_peaks=int(rng.choice([1, 3, 10, 25])), min_mz_range=float(rng.choice([0.0, 10.0, 250.0])))
    offs, mzs, its, pmz, pz = [0], [], [], [], []
    for s in range(int(rng.integers(1, 200))):
        n = int(rng.choice([0, 1, 2, 9, 12, 40, 63, 64, 65, 120, 300, 900, 3000]))
        lo, hi = (5, 2100) if rng.random() < 0.7 else (400, 460)
        mz = np.sort(rng.uniform(lo, hi, n)).astype(np.float32)
        if n > 6 and rng.random() < 0.3:
            mz[rng.integers(1, n, 3)] = mz[0]                       # duplicate m/z
            mz = np.sort(mz)
        it = rng.lognormal(0, 1.5, n).astype(np.float32)
        if n > 4 and rng.random() < 0.3:
            it[rng.integers(0, n, 4)] = it[0]                        # intensity ties
        mzs.append(mz), its.append(it)
        offs.append(offs[-1] + n)
        pmz.append(float(rng.uniform(300, 1200)))
        pz.append(int(rng.integers(1, 5)))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.float32)
    chg = rng.integers(0, 4, offs[-1]).astype(np.uint8)
    raw = PackedSpectra.from_numpy(np.array(offs), cat(mzs), cat(its), chg, np.array(pmz), np.array(pz))
    out, valid = spectrum.process_spectra(raw, False, SimpleNamespace(**cfg))
    o, mz, it, chg, pmz, pz = raw.numpy()
    oo, omz, oit, ochg, _, _ = out.to('cpu').numpy()
    valid = valid.cpu().numpy()
    ok = True
    for s in range(raw.n):
        sl = slice(o[s], o[s + 1])
        g, rm, ri, rs = O.process_spectrum(mz[sl], it[sl], pmz[s], pz[s], cfg['min_mz'], cfg['max_mz'],
                                           cfg['remove_precursor'], cfg['remove_precursor_tolerance'],
                                           cfg['min_intensity'], cfg['max_peaks_used'], cfg['scaling'],
                                           cfg['min_peaks'], cfg['min_mz_range'], cfg['resolution'])
        got = slice(oo[s], oo[s + 1])
        ok &= bool(valid[s]) == g
        if g:
            ok &= np.array_equal(omz[got], rm) and np.array_equal(oit[got].view(np.uint32), ri.view(np.uint32))
            ok &= np.array_equal(ochg[got], chg[sl][rs])
        else:
            ok &= oo[s + 1] == oo[s]
    # encoder on whatever survived, random grid
    hash_len = int(rng.choice([64, 400, 800, 1000, 4096]))
    bin_size = float(rng.choice([0.02, 0.04, 0.05, 1.0005]))
    if out.n and int(oo[-1]) > 0:
        vec = spectrum.spectra_to_vectors(out.mz, out.intensity, out.offsets, cfg['min_mz'], cfg['max_mz'],
                                          bin_size, hash_len, True)
        _, min_bound, _ = spectrum.get_dim(cfg['min_mz'], cfg['max_mz'], bin_size)
        want = O.encode_batch(omz, oit, oo.astype(np.int32), min_bound, bin_size, hash_len)
        v = vec.cpu().numpy() if hasattr(vec, 'cpu') else np.asarray(vec)
        nz = np.diff(oo) > 0
        ok &= np.array_equal(v[nz].view(np.uint32), want[nz].view(np.uint32))
    trials += 1
    n_spec += raw.n
    if not ok:
        bad += 1
        print('MISMATCH', cfg, 'hash_len', hash_len, 'bin', bin_size, flush=True)
print(f'{trials} trials ({n_spec} spectra), {bad} mismatches')
