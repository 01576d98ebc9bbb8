#!/usr/bin/env python3
"""Recall of the bench index geometry against exact inner product, swept over nprobe and over the
size k' of the IVF-PQ short-list that an exact re-rank would cut back to k (SURVEY.md 8d:
recall@k = |ANN_k & Exact_k| / k; "fixed recall" = recall@1024 >= 0.95 x IVF-Flat(nlist, nprobe)).

IVF-Flat and IVF-PQ share the coarse quantiser (same centroids installed in both), so at equal
nprobe they look at the same inverted lists: IVF-Flat's result is the exact top-k inside those
lists, i.e. the ceiling a re-ranked IVF-PQ short-list can reach.

  python scripts/recall_sweep.py [--library-size N] [--queries Q] > gpurun_out/recall_sweep.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--library-size', type=int, default=2_100_000)
    ap.add_argument('--queries', type=int, default=2048)
    ap.add_argument('--nlist', type=int, default=4096)
    ap.add_argument('--k', type=int, default=1024)
    ap.add_argument('--niter', type=int, default=25)
    ap.add_argument('--nprobes', default='32,64,128,192,256')
    ap.add_argument('--shortlists', default='1024,1280,1536,2048')
    args = ap.parse_args()
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd import faiss_compat as faiss
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(args.library_size, seed=20240807, device=dev, charges=(2,),
                                      charge_p=(1.0,))
    cfg = Config.open_search(num_list=args.nlist, num_probe=128, num_candidates=args.k, index='ivfpq',
                 kmeans_niter=args.niter, precursor_tolerance_mass_open=500.0,
                 precursor_tolerance_mode_open='Da')
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    part = sl.partitions[2]
    pq = sl._get_ann_index(2)
    q, truth = synthetic.make_queries(lib, aux, args.queries, seed=42, open_range=500.0, charge=2)
    xq = sl._encode(q)
    xb = sl._encode(part.spectra)
    exact = faiss.IndexFlatIP(cfg.hash_len)
    exact.add(xb)
    _, Ie = exact.search(xq, args.k)
    del exact
    flat = faiss.IndexIVFFlat(faiss.IndexFlatIP(cfg.hash_len), cfg.hash_len, args.nlist)
    flat.set_trained(pq.centroids())
    flat.add(xb)
    del xb
    src = truth['source_row'].to(dev)

    def overlap(A, B):           # mean |A_i & B_i| / |B_i valid|
        tot = num = 0
        for i in range(A.shape[0]):
            b = B[i][B[i] >= 0]
            num += int(torch.isin(A[i][A[i] >= 0], b).sum())
            tot += int(b.numel())
        return num / max(tot, 1)

    out = {'library_size': args.library_size, 'queries': args.queries, 'nlist': args.nlist,
           'k': args.k, 'rows': []}
    for nprobe in [int(x) for x in args.nprobes.split(',')]:
        flat.nprobe = nprobe
        pq.nprobe = nprobe
        _, If = flat.search(xq, args.k)
        row = {'nprobe': nprobe,
               'ivfflat_recall_vs_exact': overlap(If, Ie),
               'ivfflat_hit_source': float((If == src.unsqueeze(1)).any(1).float().mean())}
        for kp in [int(x) for x in args.shortlists.split(',')]:
            _, Ip = pq.search(xq, kp)
            r = {'ivfpq_shortlist_recall_vs_ivfflat_topk': overlap(Ip, If),
                 'ivfpq_shortlist_recall_vs_exact_topk': overlap(Ip, Ie),
                 'hit_source': float((Ip == src.unsqueeze(1)).any(1).float().mean())}
            if kp == args.k:
                r['ivfpq_recall_vs_exact'] = overlap(Ip, Ie)
            row[f'shortlist_{kp}'] = r
        out['rows'].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
