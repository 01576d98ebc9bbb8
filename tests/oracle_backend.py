"""Oracle-backed compute backend for ann_solo_amd.distributed (tests only): lets the
multi-GPU host logic (list ownership, exchange, merge) run on CPU under gloo."""
import numpy as np
import torch

from oracle import oracle_py as O

try:                                   # the workers of the gloo tests put tests/ itself on sys.path
    import exchange_ref as X
except ImportError:
    from tests import exchange_ref as X


class OracleShardBackend:
    def __init__(self, lib_np, lib_pmz32, centroids, assign, payload, codebooks, rank, world,
                 charge, k, nprobe, prec_tol, prec_mode, frag_tol, allow_shift, owner_fn):
        self.L = O.Spectra(*lib_np)
        self.pmz32 = lib_pmz32
        self.charge, self.k, self.nprobe = charge, k, nprobe
        self.prec_tol, self.prec_mode = prec_tol, prec_mode
        self.frag_tol, self.allow_shift = frag_tol, allow_shift
        self.window_tol = {'std': (20.0, 'ppm'), 'open': (prec_tol, prec_mode)}
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

    # ---- the queries as entry lists (csrc/encode.hip: the non-zero components, ascending, as
    # (dimension * 128, value bits), 64 per query), restated with numpy
    MAX_ENTRIES = 64

    def encode_entries(self, queries):
        from ann_solo_amd.distributed import EntryQueries
        v = self.encode(queries).numpy()
        n = v.shape[0]
        ent = np.zeros((n, self.MAX_ENTRIES, 2), np.int32)
        cnt = np.zeros(n, np.int32)
        for q in range(n):
            nz = np.nonzero(v[q])[0]
            cnt[q] = len(nz) if len(nz) <= self.MAX_ENTRIES else -1 - len(nz)
            nz = nz[:self.MAX_ENTRIES]
            ent[q, :len(nz), 0] = nz * 128
            ent[q, :len(nz), 1] = v[q, nz].view(np.int32)
        return EntryQueries(torch.from_numpy(ent), torch.from_numpy(cnt))

    def _dense(self, vectors):
        from ann_solo_amd.distributed import EntryQueries
        if not isinstance(vectors, EntryQueries):
            return vectors.numpy()
        ent, cnt = vectors.entries.numpy(), vectors.counts.numpy()
        v = np.zeros((len(cnt), self.ivf.d), np.float32)
        for q in range(len(cnt)):
            assert cnt[q] >= 0
            v[q, ent[q, :cnt[q], 0] // 128] = ent[q, :cnt[q], 1].view(np.float32)
        return v

    def shard_search(self, vectors):
        D, I = self.ivf.search(vectors.numpy(), self.k, self.nprobe)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(self, Ds, Is):
        D, I = O.topk_merge(Ds.numpy(), Is.numpy())
        return torch.from_numpy(D), torch.from_numpy(I)

    # ---- packed keys + the two-phase exchange (tests/exchange_ref.py restates csrc/exchange.hip);
    # ``keys = False`` makes the driver take the (D, I) exchange instead
    keys = True

    @property
    def supports_keys(self):
        return self.keys

    def shard_search_keys(self, vectors, coarse_D=None, coarse_I=None, k=None):
        if vectors.shape[0] == 0:
            return torch.zeros((0, k or self.k), dtype=torch.int64)
        D, I = self.ivf.search(self._dense(vectors), k or self.k, self.nprobe)
        return torch.from_numpy(X.pack_keys(D, I))

    def merge_keys(self, Ks):
        S, n, k = Ks.shape
        K = Ks.numpy().view(np.uint64)
        I = np.full((n, k), -1, np.int64)
        for q in range(n):
            best = X._topk_set(K[:, q, :].reshape(-1), k)
            I[q, :len(best)] = X.key_id(best)
        return None, torch.from_numpy(I)

    def keys_split(self, K, kp, want_rowmin=False):
        head, floors = X.keys_split(K.numpy(), kp)
        if want_rowmin:
            return torch.from_numpy(head), torch.from_numpy(floors), torch.from_numpy(X.row_min(K.numpy()))
        return torch.from_numpy(head), torch.from_numpy(floors)

    def keys_merge_heads(self, heads, k):
        out, bounds, need = X.keys_merge_heads(heads.numpy(), k)
        return torch.from_numpy(out), torch.from_numpy(bounds), torch.from_numpy(need)

    rescan_capacity = None          # tests: a tiny capacity forces the overflow fallback

    def keys_extras(self, K, floors, bounds, world, xcap, overflow, rescan=None):
        """``rescan`` = (rowmin, vectors, coarse_D, coarse_I, k): rows whose bound lies below the
        smallest key of a full row are searched again with the full k and answer from that row."""
        rmap = K3 = None
        if rescan is not None:
            rowmin, xv, cD, cI, k_full = rescan
            R = self.rescan_capacity or max(64, K.shape[0] // 16)
            rowlist, rmap, cnt, ov = X.rescan_list(bounds.numpy(), rowmin.numpy(), R)
            overflow[0] = max(int(overflow[0]), ov)
            overflow[1] += cnt
            n3 = min(cnt, R)
            K3 = np.zeros((R, k_full), np.int64)
            if n3:
                sel = torch.from_numpy(rowlist[:n3])
                K3[:n3] = self.shard_search_keys(xv.index_select(0, sel), None, None, k=k_full).numpy()
        xbuf, ov = X.keys_extras(K.numpy(), floors.numpy(), bounds.numpy(), world, xcap, rmap, K3)
        overflow[0] = max(int(overflow[0]), ov)
        return torch.from_numpy(xbuf)

    def keys_merge_final(self, heads, xbuf, out_keys, need, k):
        return torch.from_numpy(X.keys_merge_final(heads.numpy(), None if xbuf is None else xbuf.numpy(),
                                                   out_keys.numpy(), need.numpy(), k))

    def new_flag(self):
        """[0]: a buffer ran full somewhere; [1]: rows this shard searched a second time."""
        return torch.zeros(2, dtype=torch.int32)

    def _window_ok(self, q_pmz, tol, mode):
        """spectral_library.py:421-427 in float64 over the float32 library column."""
        l = self.pmz32.astype(np.float64)
        if mode == 'Da':
            return np.abs(q_pmz - l) * self.charge <= tol
        return np.abs(q_pmz - l) / l * 10 ** 6 <= tol

    def _best(self, Q, cands, pm_stride):
        n = Q.n
        stride = pm_stride or max(1, int(np.diff(Q.offsets).max()))
        out = dict(best_row=np.full(n, -1, np.int32), best_score=np.zeros(n),
                   n_candidates=np.zeros(n, np.int32), pm_count=np.zeros(n, np.int32),
                   pm_pairs=np.zeros((n, stride, 2), np.uint32))
        for i, cand in enumerate(cands):
            out['n_candidates'][i] = len(cand)
            if len(cand) == 0:
                continue
            b, s, m = O.best_match(Q, i, self.L, cand, self.frag_tol, self.allow_shift)
            if b >= 0:
                out['best_row'][i], out['best_score'][i] = cand[b], s
                out['pm_count'][i] = len(m)
                out['pm_pairs'][i, :len(m)] = m
        return out

    def rescore_knn(self, queries, knn, device_out=False, pm_stride=None):
        Q = O.Spectra(*queries.numpy())
        knn = knn.numpy()
        cands = []
        for i in range(Q.n):
            ok = self._window_ok(Q.precursor_mz[i], self.prec_tol, self.prec_mode)
            r = knn[i][knn[i] >= 0]
            cands.append(np.sort(r[ok[r]]).astype(np.int64))
        out = self._best(Q, cands, pm_stride)
        out['knn'] = knn
        return out

    def window_search(self, queries, mode, pm_stride=None, tol=None):
        """Window-only search (cascade level 1 / charges without an index)."""
        Q = O.Spectra(*queries.numpy())
        tol_val, tol_mode = tol if tol is not None else self.window_tol[mode]
        cands = [np.nonzero(self._window_ok(Q.precursor_mz[i], tol_val, tol_mode))[0]
                 .astype(np.int64) for i in range(Q.n)]
        return self._best(Q, cands, pm_stride)


# ---------------------------------------------------------------------------- cascade on CPU
from types import SimpleNamespace                                    # noqa: E402

from ann_solo_amd.spectral_library import BatchResult, Config, SpectralLibrary   # noqa: E402


def oracle_cosines(q, lib_spectra, rows, pm_pairs, pm_count):
    """Stand-in for the device ``ssm_cosine`` (the cosine over the peak matches,
    spectrum_similarity.py:81-106)."""
    qo, _, qi, *_ = q.numpy()
    lo, _, li, *_ = lib_spectra.numpy()
    out = np.full(q.n, np.nan)
    for i in range(q.n):
        r = int(rows[i])
        if r < 0:
            continue
        pm = np.asarray(pm_pairs[i][:int(pm_count[i])]).astype(np.int64)
        out[i] = float(np.sum(qi[qo[i] + pm[:, 0]].astype(np.float64) *
                              li[lo[r] + pm[:, 1]].astype(np.float64)))
    return out


class OracleSpectralLibrary(SpectralLibrary):
    """``SpectralLibrary`` whose device calls are answered by the oracle: the cascade driver,
    the batching and the multi-rank dispatch are the product's own code."""

    def __init__(self, parts, config, k, nprobe, world=1, rank=0, group=None):
        # parts: {charge: dict(lib_np, pmz32, centroids, assign, payload, codebooks) or window-only}
        self.config = config
        self.device = torch.device('cpu')
        self._num_candidates, self._num_probe = k, nprobe
        self._ann_filenames = {z: 'oracle' for z, p in parts.items() if p.get('centroids') is not None}
        self._dist = None if world == 1 else SimpleNamespace(group=group, world=world, rank=rank,
                                                             backends={})
        self.partitions, self._full, self._shard = {}, {}, {}
        from ann_solo_amd.distributed import lpt_owner
        from ann_solo_amd.packed import PackedSpectra
        for z, p in parts.items():
            self.partitions[z] = SimpleNamespace(spectra=PackedSpectra.from_numpy(*p['lib_np']),
                                                 ids=np.arange(len(p['pmz32'])))
            if z in self._ann_filenames:
                mk = lambda r, w: OracleShardBackend(
                    p['lib_np'], p['pmz32'], p['centroids'], p['assign'], p['payload'],
                    p.get('codebooks'), r, w, z, k, nprobe, config.precursor_tolerance_mass_open,
                    config.precursor_tolerance_mode_open, config.fragment_mz_tolerance,
                    config.allow_peak_shifts, lpt_owner)
                self._full[z] = mk(0, 1)
                self._shard[z] = mk(rank, world)
            else:                       # window-only partition: a backend without an index
                be = OracleShardBackend.__new__(OracleShardBackend)
                be.L, be.pmz32, be.charge = O.Spectra(*p['lib_np']), p['pmz32'], z
                be.frag_tol, be.allow_shift = config.fragment_mz_tolerance, config.allow_peak_shifts
                be.window_tol = {'std': (20.0, 'ppm'), 'open': (config.precursor_tolerance_mass_open,
                                                                config.precursor_tolerance_mode_open)}
                self._full[z] = self._shard[z] = be
        for be in list(self._full.values()) + list(self._shard.values()):
            be.window_tol['std'] = (config.precursor_tolerance_mass, config.precursor_tolerance_mode)

    def _shard_backend(self, charge, mode):
        return self._shard[charge]

    def _search_batch_local(self, queries, charge, mode, want_knn=False, device_out=False,
                            pm_stride=None):
        if charge not in self.partitions:
            return None
        be = self._full[charge]
        if self._uses_ann(charge, mode):
            _, I = be.full.search(be.encode(queries).numpy(), self._num_candidates, self._num_probe)
            r = be.rescore_knn(queries, torch.from_numpy(I), pm_stride=pm_stride)
        else:
            r = be.window_search(queries, mode, pm_stride=pm_stride)
        return BatchResult(r['best_row'], r['best_score'], r['n_candidates'], r['pm_count'],
                           r['pm_pairs'], r.get('knn'))
