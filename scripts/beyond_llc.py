#!/usr/bin/env python3
"""The IVF-PQ scan BEYOND the Infinity Cache: the same kernel over libraries whose
tiled codes no longer fit the 256 MiB L3.

An index is trained on the first 2.1 M-spectrum synthetic library (as FAISS trains an IndexIVF on a
sub-sample: at most 256 x nlist points) and grown by `add()` with further 2.1 M-spectrum libraries of
other seeds (distinct spectra, real list-length distribution, real codes -- no code generator).
After 1, 2, 4, 8, 16 ... chunks the scan is timed at a fixed number of queries: HIP events around the
kernel on its stream (asl_profile), algorithmic bytes = scanned vectors x 36 B.

  python3 scripts/beyond_llc.py --chunks 16 --queries 8192            # one JSON line per size
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 scripts/beyond_llc.py --only 16
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', type=int, default=16)
    ap.add_argument('--chunk-size', type=int, default=2_100_000)
    ap.add_argument('--queries', type=int, default=8192)
    ap.add_argument('--nlist', type=int, default=4096)
    ap.add_argument('--nprobe', type=int, default=128)
    ap.add_argument('--k', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--only', type=int, default=0, help='time only at this many chunks (profiler runs)')
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    import torch
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd import faiss_compat as faiss
    from ann_solo_amd.spectrum import spectra_to_vectors
    dev = torch.device('cuda', 0)
    L = _lib.lib()

    def encode(sp):
        out = torch.empty((sp.n, 800), dtype=torch.float32, device=dev)
        spectra_to_vectors(sp.mz, sp.intensity, sp.offsets, 11, 2010, 0.04, 800, True, out)
        return out

    t0 = time.time()
    lib0, aux0 = synthetic.make_library(args.chunk_size, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib0, aux0, args.queries, seed=42, open_range=500.0, charge=2)
    xq = encode(q)
    idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, args.nlist, 32, 8)
    idx.seed = 1234
    idx.set_niter(25)
    x0 = encode(lib0)
    idx.train(x0)
    idx.add(x0)
    del x0, lib0, aux0
    idx.nprobe = args.nprobe
    D = torch.empty((args.queries, args.k), dtype=torch.float32, device=dev)
    I = torch.empty((args.queries, args.k), dtype=torch.int64, device=dev)
    sizes = set()
    c = 1
    while c <= args.chunks:
        sizes.add(c)
        c *= 2
    sizes.add(args.chunks)
    if args.only:
        sizes = {args.only}
    have = 1
    print(f'[beyond_llc] first chunk + training {time.time() - t0:.1f}s', file=sys.stderr, flush=True)
    while True:
        if have in sizes:
            idx.search(xq, args.k, D, I)            # builds the lists, warms up
            torch.cuda.synchronize()
            L.asl_profile_reset()
            L.asl_profile_enable(1)
            for _ in range(args.reps):
                idx.search(xq, args.k, D, I)
            torch.cuda.synchronize()
            L.asl_profile_enable(0)
            ms, n = C.c_double(), C.c_int64()
            L.asl_profile_get(b'scan', C.byref(ms), C.byref(n))
            scanned = L.asl_profile_scanned_vectors() / max(n.value, 1)
            avg = ms.value / max(n.value, 1)
            info = idx.info()
            ntiles = (scanned / args.queries) / 64
            rec = {'tag': args.tag, 'chunks': have, 'ntotal': int(info.ntotal),
                   'codes_mb_untiled': round(info.ntotal * 32 / 1e6, 1),
                   'queries': args.queries, 'nprobe': args.nprobe, 'k': args.k,
                   'scan_ms': round(avg, 3), 'launches': n.value,
                   'vectors_per_query': round(scanned / args.queries, 1),
                   'algorithmic_gb': round(scanned * 36 / 1e9, 2),
                   'achieved_gbs_36B': round(scanned * 36 / (avg * 1e-3) / 1e9, 1),
                   'frac_of_8tbs': round(scanned * 36 / (avg * 1e-3) / 8e12, 4),
                   'frac_of_6.29tbs': round(scanned * 36 / (avg * 1e-3) / 6.29e12, 4),
                   'ns_per_query_tile': round(avg * 1e6 / (ntiles * args.queries), 3),
                   'min_D': float(D.min()), 'valid': int((I >= 0).sum())}
            print(json.dumps(rec), flush=True)
        if have >= max(sizes):
            break
        t1 = time.time()
        lib_i, _ = synthetic.make_library(args.chunk_size, seed=7000 + have, device=dev, charges=(2,), charge_p=(1.0,))
        x = encode(lib_i)
        del lib_i
        idx.add(x)
        del x
        have += 1
        print(f'[beyond_llc] chunk {have} added in {time.time() - t1:.1f}s', file=sys.stderr, flush=True)


if __name__ == '__main__':
    main()
