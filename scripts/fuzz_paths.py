"""Randomised end-to-end parity run: the whole device hot path (encode -> IVF retrieve -> precursor
post-filter -> (shifted) dot best match) against the CPU oracle over random configurations --
index kind and PQ shape, nlist / nprobe / k, tolerance mode and width, peak shifts on or off,
exact re-rank on or off, the two-stream pipeline on or off, standard and open searches.
Neighbour ids, winners, scores, candidate counts and peak matches must be IDENTICAL.

  python scripts/fuzz_paths.py [seconds] [seed]        (test infrastructure: uses oracle/)
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from oracle import oracle_py as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t_end = time.time() + budget
trials = bad = 0
dev = torch.device('cuda', 0)
while time.time() < t_end and trials < int(os.environ.get('FUZZ_MAX', 10 ** 9)):
    n = int(rng.choice([1500, 4000, 12000, 40000]))
    z = int(rng.choice([2, 3]))
    nlist = int(rng.choice([8, 16, 64, 200, 200, 1100]))       # 1100: more than 512 probes (two probes per thread)
    nlist = min(nlist, n // 40)
    nprobe = int(rng.integers(1, nlist + 1))
    k = int(rng.choice([1, 17, 128, 512, 1024, 2048, 1024, 2049, 3000, 5000]))   # > 2048: bounded passes
    index = str(rng.choice(['ivfpq', 'ivfflat']))
    pq_m = int(rng.choice([8, 16, 32]))
    pq_bits = int(rng.choice([6, 8])) if pq_m != 32 or rng.random() < 0.3 else 8
    tol_mode = str(rng.choice(['Da', 'ppm']))
    tol = float(rng.choice([0.05, 5.0, 300.0, 500.0])) if tol_mode == 'Da' else float(rng.choice([10.0, 2e4, 2e5]))
    shifts = bool(rng.random() < 0.8)
    frag = float(rng.choice([0.02, 0.05, 0.3]))
    refine = int(rng.choice([0, 0, 2 * k])) if index == 'ivfpq' and 2 * k <= 2048 and k > 1 else 0
    pipe = bool(rng.random() < 0.5)
    nq = int(rng.choice([1, 33, 257, 700]))
    s_train, s_lib, s_q = (int(rng.integers(1, 1 << 30)) for _ in range(3))
    n_iter = int(rng.integers(1, 5))
    only = os.environ.get('FUZZ_ONLY')
    if only is not None and trials not in [int(x) for x in only.split(',')]:
        trials += 1
        continue
    if os.environ.get('FUZZ_PIPE') is not None:
        pipe = bool(int(os.environ['FUZZ_PIPE']))
    cfg = Config.open_search(num_list=nlist, num_probe=nprobe, num_candidates=k, index=index, pq_m=pq_m,
                 pq_bits=pq_bits, kmeans_niter=n_iter, seed=s_train,
                 precursor_tolerance_mass_open=tol, precursor_tolerance_mode_open=tol_mode,
                 fragment_mz_tolerance=frag, allow_peak_shifts=shifts, refine_k=refine or None)
    desc = dict(n=n, z=z, nlist=nlist, nprobe=nprobe, k=k, index=index, pq_m=pq_m, pq_bits=pq_bits,
                tol=(tol, tol_mode), shifts=shifts, frag=frag, refine=refine, pipe=pipe, nq=nq)
    if os.environ.get('FUZZ_VERBOSE'):
        print('trial', trials, desc, flush=True)
    lib, aux = synthetic.make_library(n, seed=s_lib, device='cpu', charges=(z,),
                                      charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, nq, seed=s_q, charge=z,
                                  open_range=500.0)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    part = sl.partitions[z]
    idx = sl._get_ann_index(z)
    info = idx.info()
    off, ids, payload = idx.lists()
    ivf = O.HostIVF.__new__(O.HostIVF)
    ivf.centroids, ivf.nlist, ivf.d = idx.centroids(), info.nlist, info.d
    ivf.list_offsets, ivf.ids, ivf.payload = off, ids, payload
    ivf.codebooks = idx.codebooks() if info.kind == 2 else None
    ivf.kind = 1 if info.kind == 2 else 0
    Q, L = O.Spectra(*q.numpy()), O.Spectra(*part.spectra.to('cpu').numpy())
    stride = q.max_peaks()
    ok = True
    try:
        if refine:      # the oracle's short-list, re-ranked exactly, then its rescoring
            xb = sl._encode(part.spectra).cpu().numpy()
            xq = sl._encode(q.to(dev)).cpu().numpy()
            _, I_short = ivf.search(xq, refine, nprobe)
            _, knn = O.refine(xb, xq, I_short, k)
            want = dict(knn_I=knn)
        else:
            want = O.search_batch(Q, L, part.precursor_mz, z, ivf, k, nprobe, tol, tol_mode, frag, shifts,
                                  pm_stride=stride, want_knn=True)
        sl.set_pipeline(pipe)
        got = sl._search_batch(q.to(dev), z, 'open', want_knn=True, device_out=True)
        got2 = sl._search_batch(q.to(dev), z, 'open', device_out=True)
        std = sl._search_batch(q, z, 'std')
        sl.synchronize()
        sl.set_pipeline(False)
        knn = got.knn.cpu().numpy()
        ok &= np.array_equal(knn, want['knn_I'])
        if not refine:
            for r in (got, got2):
                ok &= np.array_equal(r.best_row.cpu().numpy(), want['best_row'])
                ok &= np.array_equal(r.best_score.cpu().numpy(), want['best_score'])
                ok &= np.array_equal(r.pm_count.cpu().numpy(), want['pm_count'])
            ok &= np.array_equal(got.n_candidates.cpu().numpy(), want['n_cand'])
            pm = got.pm_pairs.cpu().numpy().view(np.uint32)
            ok &= np.array_equal(pm, want['pm_pairs'][:, :pm.shape[1]])
        else:
            ok &= np.array_equal(got.best_row.cpu().numpy(), got2.best_row.cpu().numpy())
        # standard search of a few queries: window candidates + best match
        lp = part.precursor_mz.astype(np.float64)
        for i in range(0, nq, max(1, nq // 5)):
            cand = np.nonzero(np.abs(Q.precursor_mz[i] - lp) / lp * 1e6 <= 20.0)[0].astype(np.int64)
            b, s, m = O.best_match(Q, i, L, cand, frag, shifts) if len(cand) else (-1, 0.0, None)
            ok &= std.n_candidates[i] == len(cand)
            ok &= std.best_row[i] == (cand[b] if b >= 0 else -1)
            if b >= 0:
                ok &= std.best_score[i] == s and np.array_equal(std.peak_matches(i), m)
    except Exception as e:      # a configuration the library refuses must say so, not crash
        ok = False
        print('EXCEPTION', type(e).__name__, str(e)[:200])
    trials += 1
    if not ok:
        bad += 1
        print('MISMATCH', desc, flush=True)
    sl.shutdown()
    del sl
print(f'{trials} trials, {bad} mismatches')
