"""Oracle-backed compute backend for ann_solo_amd.distributed (tests only): lets the
multi-GPU host logic (list ownership, exchange, merge) run on CPU under gloo."""
import numpy as np
import torch

from oracle import oracle_py as O


class OracleShardBackend:
    def __init__(self, lib_np, lib_pmz32, centroids, assign, payload, codebooks, rank, world,
                 charge, k, nprobe, prec_tol, prec_mode, frag_tol, allow_shift, owner_fn):
        self.L = O.Spectra(*lib_np)
        self.pmz32 = lib_pmz32
        self.charge, self.k, self.nprobe = charge, k, nprobe
        self.prec_tol, self.prec_mode = prec_tol, prec_mode
        self.frag_tol, self.allow_shift = frag_tol, allow_shift
        nlist = len(centroids)
        sizes = np.bincount(assign, minlength=nlist)
        owner = owner_fn(sizes, world)
        self.owner = owner
        full = O.HostIVF(centroids, assign, payload, codebooks)
        # keep only owned lists: zero-length for the others (ids stay global)
        keep = owner[assign[full.ids]] == rank
        self.ivf = O.HostIVF.__new__(O.HostIVF)
        self.ivf.centroids, self.ivf.nlist, self.ivf.d = full.centroids, full.nlist, full.d
        self.ivf.codebooks, self.ivf.kind = full.codebooks, full.kind
        self.ivf.ids = np.ascontiguousarray(full.ids[keep])
        self.ivf.payload = np.ascontiguousarray(full.payload[keep])
        cnt = np.bincount(assign[self.ivf.ids], minlength=nlist)
        self.ivf.list_offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        self.full = full

    def encode(self, queries):
        o, mz, inten, *_ = queries.numpy()
        return torch.from_numpy(O.encode_batch(mz, inten, o, 10.96, 0.04, self.ivf.d))

    def shard_search(self, vectors):
        D, I = self.ivf.search(vectors.numpy(), self.k, self.nprobe)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(self, Ds, Is):
        D, I = O.topk_merge(Ds.numpy(), Is.numpy())
        return torch.from_numpy(D), torch.from_numpy(I)

    def rescore_knn(self, queries, knn, device_out=False):
        Q = O.Spectra(*queries.numpy())
        knn = knn.numpy()
        best_row = np.full(Q.n, -1, np.int32)
        best_score = np.zeros(Q.n)
        for i in range(Q.n):
            cand = np.sort(np.array([r for r in knn[i] if r >= 0 and O.precursor_ok(
                Q.precursor_mz[i], self.pmz32[r], self.charge, self.prec_tol, self.prec_mode)],
                np.int64))
            b, s, _ = O.best_match(Q, i, self.L, cand, self.frag_tol, self.allow_shift)
            if b >= 0:
                best_row[i], best_score[i] = cand[b], s
        return dict(best_row=best_row, best_score=best_score, knn=knn)
