#!/usr/bin/env python3
"""The reference's ANN hyper-parameter grid (notebooks/iprg2012_ann_hyperparameters.ipynb:100-122:
num_list in {64, 256, 1024, 4096, 16384} x num_probe in {1, 8, 32, 64, 128, 256, 512, 1024},
num_list > num_probe; search speed against identifications) on the bench library, with the
reference's index type (IVF-Flat, float32 storage): per point the whole hot path's query spectra/s
(16 384-query batches, open +-500 Da, shifted dot) next to what the candidates are worth -- recall@k
against exact inner-product search, hit@k of the source spectrum and the fraction of queries whose
best match IS the source spectrum -- on the HARD synthetic queries (calibrated to the reference's
exact-search hit@1024 of 75 %) and on the default ones.

  python scripts/hyperparameter_grid.py [--library-size N] [--nlists 64,256,...] > grid.json
(num_probe 1024 -- the reference's clamp, spectral_library.py:77-81 -- runs on the two-probes-per-thread
form of the scans since round 6: the grid's two such points, num_list 4096 and 16384.)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--library-size', type=int, default=2_100_000)
    ap.add_argument('--queries', type=int, default=1024, help='queries of the recall / hit sample')
    ap.add_argument('--batch', type=int, default=16384)
    ap.add_argument('--k', type=int, default=1024)
    ap.add_argument('--nlists', default='64,256,1024,4096,16384')
    ap.add_argument('--nprobes', default='1,8,32,64,128,256,512,1024')
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    import torch
    from dataclasses import replace
    from ann_solo_amd import synthetic
    from ann_solo_amd import faiss_compat as faiss
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(args.library_size, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
    base = Config.open_search(num_list=4096, num_probe=128, num_candidates=args.k, index='ivfflat', kmeans_niter=25, mode='ann',
                  precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da', batch_size=args.batch)
    sets = {}
    for name, hard in (('hard', synthetic.HARD_DEFAULT), ('default', 0.0)):
        q, truth = synthetic.make_queries(lib, aux, args.batch, seed=43 if hard else 42, open_range=500.0, charge=2,
                                          hard=hard)
        sets[name] = (q.contiguous(), truth)
    sl0 = SpectralLibrary(lib, config=replace(base, mode='bf'), device=dev)
    exact = faiss.IndexFlatIP(base.hash_len)
    exact.add(sl0._encode(sl0.partitions[2].spectra))
    nr = args.queries
    Ie = {}
    for name, (q, truth) in sets.items():
        _, Ie[name] = exact.search(sl0._encode(q.select(torch.arange(nr, device=dev))), args.k)
    del exact
    sl0.shutdown()
    out = {'library_size': args.library_size, 'k': args.k, 'batch': args.batch, 'index': 'ivfflat (float32 storage)',
           'sample_queries': nr, 'hard': synthetic.hard_levers(synthetic.HARD_DEFAULT), 'rows': []}
    for nlist in [int(x) for x in args.nlists.split(',')]:
        t0 = time.time()
        sl = SpectralLibrary(lib, config=replace(base, num_list=nlist), device=dev)
        idx = sl._get_ann_index(2)
        torch.cuda.synchronize()
        t_build = time.time() - t0
        for nprobe in [int(x) for x in args.nprobes.split(',')]:
            if nprobe >= nlist:
                continue
            sl._num_probe = nprobe
            idx.nprobe = nprobe
            row = {'num_list': nlist, 'num_probe': nprobe, 'index_build_s': round(t_build, 1)}
            for name, (q, truth) in sets.items():
                qs = q.select(torch.arange(nr, device=dev))
                r = sl._search_batch(qs, 2, 'open', want_knn=True, device_out=True)
                src, mod = truth['source_row'][:nr], truth['is_modified'][:nr]
                hit = (r.knn == src.unsqueeze(1)).any(1)
                rec = sum(int(torch.isin(r.knn[i][r.knn[i] >= 0], Ie[name][i]).sum()) for i in range(nr)) / float(nr * args.k)
                top1 = r.best_row.to(torch.int64) == src
                sl._search_batch(q, 2, 'open', device_out=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    sl._search_batch(q, 2, 'open', device_out=True)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / args.steps * 1e3
                row[name] = {'spectra_per_s': round(q.n / ms * 1e3, 1), 'ms_per_batch': round(ms, 3),
                             'recall_at_k_vs_exact_ip': round(rec, 4),
                             'hit_at_k_source': round(float(hit.float().mean()), 4),
                             'hit_at_k_modified_only': round(float(hit[mod].float().mean()), 4),
                             'best_match_is_source': round(float(top1.float().mean()), 4),
                             'best_match_is_source_modified_only': round(float(top1[mod].float().mean()), 4)}
            out['rows'].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
        sl.shutdown()
        del sl, idx
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
