"""CPU-only study (no GPU minutes): which 32-byte-per-vector code reaches the fixed-recall
criterion of SURVEY.md 8(d) -- recall@k >= 0.95 x IVF-Flat(nlist, nprobe) -- on the bench's
synthetic library? VERDICT r2 item 4.

    python scripts/pq_variants.py [library_size=200000] [queries=1024] [out.json]

Scaled-down model of configs[2] (2.1 M vectors, nlist 4096, nprobe 128, k 1024): the same
generator, N = 200 000, nlist = 512 and nprobe = 16 (same nprobe / nlist = 1/32 and about the same
list length, 390 vs 512), and the depth of the neighbour list scaled with the library: k = 100
(k / scanned vectors = 1.6 % as at full size; k = 1024 of 6 250 scanned vectors would be a far
easier question than 1024 of 73 000). The IVF-PQ baseline of the product (m = 32 x 8 bit on
residuals) is one of the variants, so the model can be checked against the full-size measurement
(ratio 0.63, profiles/r02_recall_sweep.json).

Everything is numpy / scipy.sparse float32 with float64 k-means accumulators; trainers are plain
Lloyd (k random points, 10 iterations; spherical for the inner-product coarse quantiser as in the
product). This is an evaluation script: recall figures, not bit-exact parity.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'profiles', 'r03_pq_variants.json')
NLIST, NPROBE, D = 512, 16, 800
KS = (100, 1024)
rng = np.random.default_rng(1234)


def log(*a):
    print(f'[{time.time() - T0:7.1f}s]', *a, flush=True)


def kmeans(x, k, niter, spherical=False, seed=0):
    """Lloyd on dense float32 rows (L2) or, ``spherical``, on sparse rows by inner product."""
    r = np.random.default_rng(seed)
    n = x.shape[0]
    c = x[r.choice(n, k, replace=False)]
    c = np.asarray(c.todense(), np.float32) if sp.issparse(c) else c.astype(np.float32).copy()
    if spherical:
        c /= np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-20)
    for _ in range(niter):
        a = assign(x, c, spherical)
        if sp.issparse(x):
            onehot = sp.csr_matrix((np.ones(n, np.float32), (a, np.arange(n))), shape=(k, n))
            s = np.asarray((onehot @ x).todense(), np.float64)
        else:
            s = np.zeros((k, x.shape[1]), np.float64)
            np.add.at(s, a, x)
        cnt = np.bincount(a, minlength=k)
        live = cnt > 0
        c[live] = (s[live] / cnt[live, None]).astype(np.float32)
        if (~live).any():                       # empty cluster: re-seed on a random point
            p = x[r.choice(n, int((~live).sum()), replace=False)]
            c[~live] = np.asarray(p.todense(), np.float32) if sp.issparse(p) else p
        if spherical:
            c /= np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-20)
    return c


def assign(x, c, ip=False, chunk=32768):
    out = np.empty(x.shape[0], np.int32)
    cn = None if ip else (c.astype(np.float64) ** 2).sum(1).astype(np.float32)
    for a in range(0, x.shape[0], chunk):
        s = np.asarray(x[a:a + chunk] @ c.T)
        out[a:a + chunk] = np.argmax(s if ip else 2 * s - cn[None, :], axis=1)
    return out


class PQ:
    """m sub-quantisers of 2^bits centroids over ``d / m`` consecutive dimensions."""

    def __init__(self, m, bits, train, seed=7, niter=10):
        self.m, self.ksub, self.dsub = m, 1 << bits, D // m
        assert D % m == 0
        self.cb = np.stack([kmeans(train[:, j * self.dsub:(j + 1) * self.dsub], self.ksub, niter,
                                   seed=seed + j) for j in range(m)])        # [m, ksub, dsub]

    def encode(self, x, chunk=65536):
        codes = np.empty((x.shape[0], self.m), np.int32)
        for j in range(self.m):
            codes[:, j] = assign(np.ascontiguousarray(x[:, j * self.dsub:(j + 1) * self.dsub]),
                                 self.cb[j], chunk=chunk)
        return codes

    def lut(self, q):                     # [m, ksub] inner products of the query's sub-vectors
        return np.einsum('mkd,md->mk', self.cb, q.reshape(self.m, self.dsub))

    def bytes(self):
        return self.m * int(np.log2(self.ksub)) / 8


def main():
    global T0
    T0 = time.time()
    import torch
    from ann_solo_amd import synthetic
    from oracle import oracle_py as O
    lib, aux = synthetic.make_library(N, seed=20240807, device='cpu', charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, NQ, seed=42, charge=2)
    lo, lmz, lit, *_ = lib.numpy()
    qo, qmz, qit, *_ = q.numpy()
    X = O.encode_batch(lmz, lit, lo, 10.96, 0.04, D)
    Q = O.encode_batch(qmz, qit, qo, 10.96, 0.04, D)
    Xs = sp.csr_matrix(X)
    log(f'library {X.shape}, {Xs.nnz / N:.1f} non-zeros per vector; queries {Q.shape}')

    cen = kmeans(Xs, NLIST, 10, spherical=True, seed=1234)
    a = assign(Xs, cen, ip=True)
    order = np.argsort(a, kind='stable')
    off = np.concatenate([[0], np.cumsum(np.bincount(a, minlength=NLIST))])
    probes = np.argsort(-(Q @ cen.T), axis=1, kind='stable')[:, :NPROBE]
    log(f'coarse quantiser: {NLIST} lists, {np.bincount(a, minlength=NLIST).mean():.0f} vectors per list')

    # exact neighbours (brute force) and IVF-Flat (exact inside the probed lists)
    S = np.asarray(Xs @ Q.T)                                   # [N, NQ]
    kmax = max(KS)
    exact = np.argsort(-S, axis=0, kind='stable')[:kmax].T     # [NQ, kmax]
    cands, cand_exact = [], []
    for i in range(NQ):
        c = np.concatenate([order[off[l]:off[l + 1]] for l in probes[i]])
        cands.append(c)
        cand_exact.append(S[c, i])
    del S
    scanned = float(np.mean([len(c) for c in cands]))
    log(f'{scanned:.0f} vectors scanned per query')

    def recall(topk_ids, k):
        return float(np.mean([len(np.intersect1d(topk_ids[i][:k], exact[i][:k])) / k for i in range(NQ)]))

    def topk_by(scores_per_query, k):
        out = []
        for i in range(NQ):
            s = scores_per_query[i]
            kk = min(k, len(s))
            sel = np.argpartition(-s, kk - 1)[:kk]
            out.append(cands[i][sel[np.argsort(-s[sel], kind='stable')]])
        return out

    flat = {k: recall(topk_by(cand_exact, k), k) for k in KS}
    log('IVF-Flat recall@k', flat)
    results = {'model': dict(N=N, nlist=NLIST, nprobe=NPROBE, queries=NQ, scanned_per_query=scanned,
                             note='k = 100 at N = 200 000 stands for k = 1024 at N = 2.1 M (same k / '
                                  'scanned vectors); k = 1024 is printed for reference'),
               'ivfflat_recall': {str(k): flat[k] for k in KS}, 'criterion': 0.95, 'variants': {}}

    def evaluate(name, adc_scores, bytes_per_vector, rerank=((4, ), (8, ))):
        """adc_scores[i]: approximate scores of cands[i]. Reports recall@k, its ratio to IVF-Flat,
        and the same after an exact re-rank of the k' = 2k / 4k / 8k best approximate hits."""
        r = {'bytes_per_vector': bytes_per_vector}
        for k in KS:
            top = topk_by(adc_scores, k)
            rec = recall(top, k)
            r[f'recall@{k}'] = round(rec, 4)
            r[f'ratio@{k}'] = round(rec / flat[k], 4)
            for f in (2, 4, 8):
                kp = f * k
                short = topk_by(adc_scores, kp)
                rr = []
                for i in range(NQ):
                    pos = {v: j for j, v in enumerate(cands[i])}
                    idx = np.fromiter((pos[v] for v in short[i]), np.int64, len(short[i]))
                    ex = cand_exact[i][idx]
                    rr.append(short[i][np.argsort(-ex, kind='stable')[:k]])
                rec2 = recall(rr, k)
                r[f'rerank{f}x_ratio@{k}'] = round(rec2 / flat[k], 4)
        results['variants'][name] = r
        log(name, json.dumps(r))
        with open(OUT, 'w') as f:
            json.dump(results, f, indent=1)

    train_rows = rng.choice(N, min(N, 65536), replace=False)
    resid = X - cen[a]
    coarse_ip = Q @ cen.T                                      # [NQ, NLIST]
    list_of = a

    def adc_residual(pq, codes, renorm=None):
        out = []
        ar = np.arange(pq.m)
        for i in range(NQ):
            lut = pq.lut(Q[i])
            c = cands[i]
            s = lut[ar[None, :], codes[c]].sum(1)
            if renorm is not None:
                s = s * renorm[c]
            out.append((coarse_ip[i, list_of[c]] + s).astype(np.float32))
        return out

    def adc_raw(pq, codes):
        out = []
        ar = np.arange(pq.m)
        for i in range(NQ):
            lut = pq.lut(Q[i])
            out.append(lut[ar[None, :], codes[cands[i]]].sum(1).astype(np.float32))
        return out

    # 1. the product's IVF-PQ: m = 32 x 8 bit on residuals (by_residual = True)
    for m, bits in ((32, 8), (40, 6), (50, 5), (25, 10), (80, 4), (100, 4)):
        pq = PQ(m, bits, resid[train_rows])
        evaluate(f'pq_residual_m{m}x{bits}', adc_residual(pq, pq.encode(resid)), pq.bytes())
    # 2. by_residual off: the raw hashed vectors are quantised
    for m, bits in ((32, 8), (50, 5)):
        pq = PQ(m, bits, X[train_rows])
        evaluate(f'pq_raw_m{m}x{bits}', adc_raw(pq, pq.encode(X)), pq.bytes())
    # 3. m = 32 x 8 bit on L2-renormalised residuals (+ 1 byte for the norm, quantised to 8 bits)
    nrm = np.linalg.norm(resid, axis=1)
    unit = resid / np.maximum(nrm[:, None], 1e-20)
    nq8 = np.round(nrm / nrm.max() * 255) / 255 * nrm.max()
    pq = PQ(32, 8, unit[train_rows])
    evaluate('pq_unit_residual_m32x8_plus_norm_byte', adc_residual(pq, pq.encode(unit), renorm=nq8.astype(np.float32)), 33)
    # 4. rank-order code: the hashed dimensions of the 25 most intense entries, most intense first,
    #    10 bits each (+ 6 bits count): the value of position j is implied by the rank scaling,
    #    (R - j) / norm -- an exact partial inner product over the vector's top entries
    for keep in (25, 16):
        out = []
        rows_i, cols_i, vals_i = [], [], []
        indptr, indices, data = Xs.indptr, Xs.indices, Xs.data
        for v in range(N):
            s, e = indptr[v], indptr[v + 1]
            d_, x_ = indices[s:e], data[s:e]
            o = np.argsort(-x_, kind='stable')[:keep]
            n_ = e - s
            # implied values: rank scaling gives (R - j) for the j-th most intense of n peaks,
            # R = 50 for the library; hash collisions (summed peaks) are approximated the same way
            imp = (50.0 - np.arange(len(o))).astype(np.float32)
            full = (50.0 - np.arange(n_)).astype(np.float32)
            imp /= np.sqrt((full ** 2).sum())
            rows_i.append(np.full(len(o), v)), cols_i.append(d_[o]), vals_i.append(imp)
        T = sp.csr_matrix((np.concatenate(vals_i), (np.concatenate(rows_i), np.concatenate(cols_i))), shape=(N, D))
        for i in range(NQ):
            out.append(np.asarray(T[cands[i]] @ Q[i]).ravel().astype(np.float32))
        evaluate(f'rank_order_top{keep}_10bit_dims', out, (keep * 10 + 6) / 8)
    # 5. truncated sparse code with explicit values: 16 entries x (10-bit dimension + 6-bit value)
    out = []
    rows_i, cols_i, vals_i = [], [], []
    for v in range(N):
        s, e = Xs.indptr[v], Xs.indptr[v + 1]
        d_, x_ = Xs.indices[s:e], Xs.data[s:e]
        o = np.argsort(-x_, kind='stable')[:16]
        rows_i.append(np.full(len(o), v)), cols_i.append(d_[o])
        vals_i.append(np.round(x_[o] * 63 / 0.5).clip(0, 63) * 0.5 / 63)
    T = sp.csr_matrix((np.concatenate(vals_i).astype(np.float32), (np.concatenate(rows_i), np.concatenate(cols_i))),
                      shape=(N, D))
    for i in range(NQ):
        out.append(np.asarray(T[cands[i]] @ Q[i]).ravel().astype(np.float32))
    evaluate('sparse_top16_10bit_dim_6bit_value', out, 32)
    log('done ->', OUT)


if __name__ == '__main__':
    main()
