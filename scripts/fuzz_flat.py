"""Randomised cross-check of the IVF-Flat scan kernels: postings (variant 0) and sparse tiles
(variant 2) against the dense GEMM formulation (variant 1) on random sparse data, ordered rows
bit for bit and unordered rows as sets.   python scripts/fuzz_flat.py [trials] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import faiss_compat as faiss

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for t in range(trials):
    d = int(rng.integers(64, 1000))
    nnz = int(rng.integers(1, max(2, d // 9)))
    n = int(rng.integers(200, 20000))
    nlist = int(rng.integers(1, 64))
    nprobe = int(rng.integers(1, nlist + 1))
    k = int(rng.integers(1, 2049))
    nq = int(rng.integers(1, 700))
    dup = rng.random() < 0.3          # many identical vectors: ties everywhere
    signed = rng.random() < 0.3       # negative components too (exact cancellations, zero scores)

    def rows(m):
        x = np.zeros((m, d), np.float32)
        for i in range(m):
            c = rng.choice(d, size=min(nnz, d), replace=False)
            x[i, c] = rng.random(len(c)).astype(np.float32) + 0.05
            if signed:
                x[i, c] *= rng.choice(np.float32([-1.0, 1.0]), size=len(c))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x
    xb = rows(n if not dup else max(50, n // 40))
    if dup:
        xb = xb[rng.integers(0, len(xb), size=n)]
    xq = rows(nq)
    idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist)
    idx.set_trained(xb[rng.choice(n, nlist, replace=False)].copy())
    idx.add(xb)
    idx.nprobe = nprobe
    out = {}
    for v in (1, 0):
        idx.set_scan_variant(v)
        out[v] = idx.search(xq, k)
    idx.set_scan_variant(0)
    idx.set_unordered(1)
    Du, Iu = idx.search(xq, k)
    idx.set_unordered(0)
    ok = True
    for v in (0,):
        ok &= np.array_equal(out[v][1], out[1][1]) and np.array_equal(out[v][0].view(np.uint32), out[1][0].view(np.uint32))
    ok &= np.array_equal(np.sort(Iu, 1), np.sort(out[1][1], 1))
    if not ok:
        bad += 1
        print('MISMATCH', dict(d=d, nnz=nnz, n=n, nlist=nlist, nprobe=nprobe, k=k, nq=nq, dup=dup))
print(f'{trials} trials, {bad} mismatches')
