"""Multi-GPU search: the spectral library's IVF index sharded by inverted list over the
ranks of one node, one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The reference has no distributed code at all; this is the north star's
addition. Per batch and charge partition:

  1. the PEAKS of every rank's slice of the query batch are all-gathered as fixed-width rows
     (~0.5 KB per query; the hashed vector would be 3.2 KB) and every rank hashes all
     queries itself (HIP encoder);
  2. every rank searches its own inverted lists for ALL queries: coarse quantiser is
     replicated (identical probe lists everywhere, bit-exact fp32 MFMA chain), the
     PQ/flat scan touches only locally owned lists -> per-shard top-k;
  3. the exchange (round 4: two phases, exact; csrc/exchange.hip has the argument). Hits are
     packed 8-byte keys (order-preserving score bits << 32 | ~id). Phase 1: an all-to-all of the
     HEAD of every per-shard row -- about 2 k / world of its best keys plus its best held-back
     key T -- so that rank r receives ``world`` heads for each query of its own slice (~2 k keys
     per query instead of world * k). The owner merges them, takes the k-th best key it has seen
     as a bound B and asks exactly those shards whose T beats B for what they hold above B
     (all-to-all of 8 bytes per (query, shard), then an all-to-all of one small fixed-size
     buffer per pair of ranks: usually empty). xGMI is point-to-point, so an all-to-all maps to
     direct peer copies. The scan runs in a few query chunks and the collectives of one chunk
     are in flight on RCCL's stream while the next chunk is scanned;
  4. merge under (score desc, id asc) -- identical to the unsharded result by construction
     (a key that is never shipped lies below a bound that k shipped keys reach) -- then precursor
     post-filter + shifted-dot rescoring data-parallel over the rank's own queries (the packed
     peak store is replicated: ~1 GB of 288). If a phase-2 buffer overflows (a flag, checked
     once per batch) the batch is repeated with the full world * k exchange;
  5. (round 5) from four ranks on the shards scan with a shard-side k_s < k (``shard_k``: 512 of
     1024 at eight ranks -- a shard sees an eighth of a query's candidates, and the appends of a
     k-deep row cost its scan 0.5-0.9 ms per step). A full k_s-row may have dropped keys, all of
     them below its smallest key M. Nothing changes for the owners (a full row always holds keys
     back, and M <= T: "ask iff T > B" covers dropped keys too); the SHARD completes its answer:
     where the bound B it is sent lies below M (~1 % of its rows) it scans that query again with
     the full k -- a launch of fixed size gated by a device-side count, no host round trip, no
     extra collective -- and answers from that row. The result stays the exact top k of the
     union of the shards' full rows.

List ownership is the greedy heaviest-first balancing of ``asl_lpt_owner`` over the
expected scan load of each list (size squared: populous lists are also probed more often);
identical on every rank, no communication. The compute backend is injectable so
that the host logic (ownership, exchange, merge order) is covered by world_size-2
``gloo`` tests on CPU; the product backend is the HIP library and nothing else.

Shard degree. ``world = degree x replicas``: the lists are sharded over the ``degree``
consecutive ranks of a *shard group* (``make_shard_groups``) and every group holds a full
copy of the library and serves its own queries -- no traffic between groups. ``degree =
world`` is the fully list-sharded layout above; ``degree = 1`` is the replicas-only
fallback of SURVEY.md 8(e) for libraries that fit one GPU (no collective on the data path
at all). ``pick_shard_degree`` returns the smallest power-of-two degree whose per-GPU share
of the index fits a memory budget: per-(query, shard) costs (LUT build, top-k finish) are
paid ``degree`` times, so the smallest degree that fits is the fastest one.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .packed import PackedSpectra


def lpt_owner(list_sizes, world: int) -> np.ndarray:
    """Owner rank of every inverted list (host integer code inside libannsolo_mi)."""
    sizes = np.ascontiguousarray(list_sizes, np.int64)
    owner = np.empty(len(sizes), np.int32)
    _lib.check(_lib.lib().asl_lpt_owner(len(sizes), _lib.ptr(sizes), int(world),
                                        _lib.ptr(owner)))
    return owner


def pick_shard_degree(index_bytes: int, replicated_bytes: int, world: int,
                      budget_bytes: int = 240 << 30) -> int:
    """Smallest power-of-two shard degree (dividing ``world``) such that index_bytes/degree +
    replicated_bytes (peak store, centroids, codebooks, work buffers) fits ``budget_bytes``
    of one GPU's 288 GB. Returns ``world`` if nothing smaller fits."""
    d = 1
    while d < world:
        if world % d == 0 and index_bytes / d + replicated_bytes <= budget_bytes:
            return d
        d *= 2
    return world


def make_shard_groups(degree: int, backend: Optional[str] = None, world_group=None):
    """Split the job into ``world // degree`` shard groups of ``degree`` consecutive ranks
    (consecutive = same xGMI neighbourhood, and the same node first when a job spans nodes).
    Collective over ALL ranks (every rank creates every group, as ``dist.new_group`` requires).
    ``backend``: the backend of the new groups when it is not the default group's (a job whose
    default group is gloo -- the control plane -- and whose data plane is RCCL: 'nccl');
    ``world_group``: the all-ranks group of that backend, returned when degree == world (None =
    the default group). Returns (group of this rank, rank inside the group, index of the group)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if degree < 1 or world % degree:
        raise ValueError(f'shard degree {degree} does not divide the world size {world}')
    if degree == world:
        return world_group, rank, 0
    mine = None
    for g in range(world // degree):
        ranks = list(range(g * degree, (g + 1) * degree))
        grp = dist.new_group(ranks, backend=backend) if backend else dist.new_group(ranks)
        if rank in ranks:
            mine = grp
    return mine, rank % degree, rank // degree


class EntryQueries:
    """A batch of hashed queries as ENTRY LISTS -- the non-zero components only (entries [n, 64, 2]
    int32 = (dimension * 128, value bits), counts [n] int32; ``asl_encode_entries_batch``): what the
    shard scans read, 512 bytes per query instead of a 3.2 KB dense row. Behaves like the dense
    tensor where the driver touches it (rows, device, ``index_select``)."""

    def __init__(self, entries: torch.Tensor, counts: torch.Tensor):
        self.entries, self.counts = entries, counts

    @property
    def shape(self):
        return (self.counts.shape[0],)

    @property
    def device(self):
        return self.counts.device

    def index_select(self, dim: int, rows: torch.Tensor) -> 'EntryQueries':
        assert dim == 0
        return EntryQueries(self.entries.index_select(0, rows), self.counts.index_select(0, rows))


class HipShardBackend:
    """Product backend: one charge partition of a ``SpectralLibrary`` whose ANN index has
    been sharded with ``index.shard(rank, world)``."""

    def __init__(self, spectral_library, charge: int, mode: str = 'open'):
        self.sl = spectral_library
        self.charge = charge
        self.mode = mode
        self.part = spectral_library.partitions[charge]
        # charges too rare for an ANN index (and brute-force mode) only ever run window searches
        self.index = (spectral_library._get_ann_index(charge)
                      if charge in spectral_library._ann_filenames else None)
        self.device = spectral_library.device
        self.k = spectral_library._num_candidates
        # exact re-rank of the IVF-PQ short-list (Config.refine_k): every shard returns its k'
        # best ADC hits UN-refined, the merge yields the k' best of the whole index -- the
        # short-list one GPU would re-rank -- and the query's owner re-ranks it against the exact
        # rows every rank keeps: 1 and N GPUs hand the same k candidates to the rescoring
        rk = self.index.refine_k if self.index is not None else 0
        self.k_scan = rk if rk > self.k else self.k

    @property
    def index_epoch(self):
        """Changes whenever the answer of ``supports_keys`` may: the index handle's own counter (add,
        train, scan variant, storage, re-rank, sharding) and the probe count of the engine."""
        return (getattr(self.index, 'epoch', 0), int(self.sl._num_probe)) if self.index is not None else (0, 0)

    def encode(self, queries: PackedSpectra) -> torch.Tensor:
        return self.sl._encode(queries.to(self.device))

    MAX_ENTRIES = 64      # non-zero components an entry list holds (csrc/encode.hip: QE_CAP)

    def encode_entries(self, queries: PackedSpectra) -> EntryQueries:
        """The hashed vectors of ``queries`` as entry lists (same values as ``encode``, bit for bit);
        only for spectra of at most ``MAX_ENTRIES`` peaks -- a vector has no more non-zeros than
        its spectrum has peaks -- which the driver checks on the agreed peak-row width."""
        from .spectrum import get_dim, HASH_SEED
        cfg = self.sl.config
        q = queries.to(self.device)
        n = q.n
        ent = torch.empty((n, self.MAX_ENTRIES, 2), dtype=torch.int32, device=self.device)
        cnt = torch.empty((n,), dtype=torch.int32, device=self.device)
        _, start_dim, _ = get_dim(cfg.min_mz, cfg.max_mz, cfg.bin_size)
        _lib.check(_lib.lib().asl_encode_entries_batch(
            _lib.ptr(q.mz), _lib.ptr(q.intensity), _lib.ptr(q.offsets.to(torch.int32)), n, int(q.mz.numel()),
            float(start_dim), float(cfg.bin_size), int(cfg.hash_len), int(HASH_SEED), 1,
            _lib.ptr(ent), _lib.ptr(cnt), None))
        return EntryQueries(ent, cnt)

    def shard_search(self, vectors: torch.Tensor):
        self.index.nprobe = self.sl._num_probe
        return self.index.search(vectors, self.k)

    # coarse quantiser on the rank's own slice only (probe lists are identical on every
    # rank, so they are all-gathered instead of being recomputed world times)
    supports_preassigned = True

    def coarse(self, vectors: torch.Tensor):
        return self.index.coarse(vectors, self.sl._num_probe)

    def shard_search_preassigned(self, vectors, coarse_D, coarse_I):
        # per-shard rows go straight into the merge, which orders them: skip the shard's sort
        self.index.set_unordered(True)
        try:
            return self.index.search_preassigned(vectors, self.k_scan, coarse_D, coarse_I)
        finally:
            self.index.set_unordered(False)

    # packed 8-byte hits for the exchange (instead of 4-byte score + 8-byte id)
    @property
    def supports_keys(self):
        """Packed-key rows come from the tiled IVF-PQ scan (index.hip: m = 32, 8-bit codes,
        automatic scan variant, nprobe within the tiled kernel's limit, k + 768 <= 2048) and from
        the postings scan of IVF-Flat -- which THIS shard has only if it stores sparse vectors;
        every other configuration exchanges (D, I) rows. The value is local: the driver agrees
        on it across the ranks (``sharded_search_batch``)."""
        return bool(_lib.lib().asl_index_supports_keys(self.index._h, int(self.k_scan),
                                                       int(self.sl._num_probe)))

    def shard_search_keys(self, vectors, coarse_D, coarse_I, k=None):
        k = int(k or self.k_scan)
        if vectors.shape[0] == 0:
            return torch.zeros((0, k), dtype=torch.int64, device=vectors.device)
        if isinstance(vectors, EntryQueries):
            return self.index.search_entries_keys(vectors.entries, vectors.counts, k, coarse_D, coarse_I)
        return self.index.search_preassigned_keys(vectors, k, coarse_D, coarse_I)

    def merge_keys(self, Ks: torch.Tensor):
        from . import faiss_compat
        return faiss_compat.topk_merge_keys(Ks, unordered=True)   # rescoring consumes a set

    # ---- the two-phase exchange (csrc/exchange.hip); every tensor lives on the device
    def keys_split(self, K: torch.Tensor, kp: int, want_rowmin: bool = False):
        """-> head [rows, kp] and the rows' bucket floors [rows]: the keys held back are the keys of
        ``K`` below its row's floor -- ``keys_extras`` reads them from ``K`` itself. ``want_rowmin``:
        also M [rows], the smallest key of every FULL row (0 otherwise): shards that scan with
        k_s < k."""
        rows, k = K.shape
        head = torch.empty((rows, kp), dtype=torch.int64, device=K.device)
        floor = torch.empty((rows,), dtype=torch.int32, device=K.device)
        rowmin = torch.empty((rows,), dtype=torch.int64, device=K.device) if want_rowmin else None
        _lib.check(_lib.lib().asl_keys_split(rows, k, kp, _lib.ptr(K), _lib.ptr(head), _lib.ptr(floor),
                                             _lib.ptr(rowmin)))
        return (head, floor, rowmin) if want_rowmin else (head, floor)

    def keys_merge_heads(self, heads: torch.Tensor, k: int):
        S, n, kp = heads.shape
        dev = heads.device
        out = torch.empty((n, k), dtype=torch.int64, device=dev)
        bounds = torch.empty((S, n), dtype=torch.int64, device=dev)
        need = torch.empty((n,), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().asl_keys_merge_heads(S, n, kp, k, _lib.ptr(heads), _lib.ptr(out),
                                                   _lib.ptr(bounds), _lib.ptr(need)))
        return out, bounds, need

    rescan_capacity = None      # rows a piece may scan a second time (default max(64, rows / 16))

    def keys_extras(self, K: torch.Tensor, floor: torch.Tensor, bounds: torch.Tensor, world: int, xcap: int,
                    overflow: torch.Tensor, rescan=None):
        """``rescan`` = (rowmin, vectors, coarse_D, coarse_I, k) for shards that scanned with
        k_s < k: the rows whose bound lies below the smallest key of a full row are listed on the
        device, searched again with the full k by a launch of fixed size that the device-side count
        gates, and answer from that row. Nothing here waits for the device."""
        rows, k = K.shape
        n = rows // world
        dev = K.device
        L = _lib.lib()
        xbuf = torch.empty((world, n + xcap), dtype=torch.int64, device=dev)
        # the payload cursors are the caller's (a tensor on the stream): nothing waits in here, so
        # the collectives of this piece really travel under the next piece's scan
        cursor = torch.zeros(world, dtype=torch.int32, device=dev)
        rmap = K3 = None
        k3 = 0
        if rescan is not None:
            rowmin, xv, cD, cI, k3 = rescan
            R = int(self.rescan_capacity or max(64, rows // 16))
            rowlist = torch.zeros(R, dtype=torch.int64, device=dev)
            rmap = torch.empty(rows, dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.asl_keys_rescan_list(rows, _lib.ptr(bounds), _lib.ptr(rowmin), R, _lib.ptr(rowlist),
                                              _lib.ptr(rmap), _lib.ptr(cnt), _lib.ptr(overflow)))
            overflow[1:2] += cnt                              # (statistics: rows scanned a second time)
            x3 = xv.index_select(0, rowlist)                  # slots past the count repeat row 0: never scanned
            cD3, cI3 = cD.index_select(0, rowlist), cI.index_select(0, rowlist)
            if isinstance(x3, EntryQueries):
                K3 = self.index.search_entries_keys(x3.entries, x3.counts, int(k3), cD3, cI3, gate=cnt)
            else:
                K3 = torch.empty((R, k3), dtype=torch.int64, device=dev)
                self.index.set_unordered(2)
                try:
                    _lib.check(L.asl_index_search_gated(self.index._h, R, _lib.ptr(x3), int(k3), int(cI3.shape[1]),
                                                        _lib.ptr(cD3), _lib.ptr(cI3), None, _lib.ptr(K3),
                                                        _lib.ptr(cnt)))
                finally:
                    self.index.set_unordered(0)
        _lib.check(L.asl_keys_extras(world, n, k, _lib.ptr(K), _lib.ptr(floor), _lib.ptr(bounds),
                                     int(xcap), _lib.ptr(xbuf), _lib.ptr(cursor), _lib.ptr(overflow),
                                     _lib.ptr(rmap), _lib.ptr(K3), int(k3)))
        return xbuf

    def keys_merge_final(self, heads: torch.Tensor, xbuf: Optional[torch.Tensor], out_keys, need, k: int):
        S, n, kp = heads.shape
        I = torch.empty((n, k), dtype=torch.int64, device=heads.device)
        xcap = 0 if xbuf is None else xbuf.shape[1] - n
        _lib.check(_lib.lib().asl_keys_merge_final(S, n, kp, k, _lib.ptr(heads), _lib.ptr(xbuf), int(xcap),
                                                   _lib.ptr(out_keys), _lib.ptr(need), None, _lib.ptr(I)))
        return I

    def new_flag(self):
        """[0]: a buffer ran full somewhere; [1]: rows this shard scanned a second time."""
        return torch.zeros(2, dtype=torch.int32, device=self.device)

    def refine(self, vectors: torch.Tensor, knn: torch.Tensor):
        """Merged k' short-list of the own queries -> the k best by exact inner product."""
        if self.k_scan == self.k:
            return knn
        return self.index.refine(vectors, knn, self.k)[1]

    def merge(self, Ds: torch.Tensor, Is: torch.Tensor):
        from . import faiss_compat
        return faiss_compat.topk_merge(Ds, Is)

    def window_search(self, queries: PackedSpectra, mode: str, pm_stride=None):
        """Precursor-window search of the rank's own queries against the replicated peak store
        (cascade level 1, and charges without an ANN index): no exchange at all."""
        return self.sl._search_batch_local(queries, self.charge, mode, device_out=True,
                                           pm_stride=pm_stride)

    def rescore_knn(self, queries: PackedSpectra, knn: torch.Tensor, device_out=False,
                    pm_stride=None):
        from .spectral_library import BatchResult
        from .spectrum import get_dim, HASH_SEED
        sl, cfg = self.sl, self.sl.config
        tol_val, tol_mode = sl._tolerance(self.mode)
        q = queries.to(self.device).contiguous()
        nq = q.n
        stride = pm_stride or q.max_peaks()
        if device_out:
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.device)
            best_row, best_score = mk((nq,), torch.int32), mk((nq,), torch.float64)
            n_cand, pm_count = mk((nq,), torch.int32), mk((nq,), torch.int32)
            pm_pairs = torch.empty((nq, stride, 2), dtype=torch.int32, device=self.device)
        else:
            best_row, best_score = np.empty(nq, np.int32), np.empty(nq, np.float64)
            n_cand, pm_count = np.empty(nq, np.int32), np.empty(nq, np.int32)
            pm_pairs = np.empty((nq, stride, 2), np.uint32)
        _, min_bound, _ = get_dim(cfg.min_mz, cfg.max_mz, cfg.bin_size)
        P = _lib.AslSearchParams(min_bound, cfg.bin_size, HASH_SEED, self.k, sl._num_probe,
                                 self.charge, float(tol_val), 0 if tol_mode == 'Da' else 1,
                                 cfg.fragment_mz_tolerance, int(cfg.allow_peak_shifts), 1)
        knn = knn.contiguous()
        _lib.check(_lib.lib().asl_rescore_knn(
            self.part.handle, C.byref(_lib.peaks_struct(q)), C.byref(P), _lib.ptr(knn),
            _lib.ptr(best_row), _lib.ptr(best_score), _lib.ptr(n_cand), _lib.ptr(pm_count),
            _lib.ptr(pm_pairs), stride))
        return BatchResult(best_row, best_score, n_cand, pm_count, pm_pairs, knn)


# The collectives below have two forms: on RCCL the tensors travel as they are (all_to_all_single /
# all_gather_into_tensor on device memory, asynchronous handles); on gloo (CPU tests, two ranks
# sharing one GPU) host copies are all-gathered and sliced. gloo implements the direct forms for
# HOST tensors too, so the CPU tests can run the RCCL branches -- shapes, views, handles, the
# four-piece pipeline -- by setting this flag (tests/test_distributed_cpu.py, kinds '*_direct').
FORCE_DIRECT_COLLECTIVES = False


def _direct(group=None) -> bool:
    return FORCE_DIRECT_COLLECTIVES or dist.get_backend(group) == 'nccl'


def _comm_tensor(x: torch.Tensor, group=None) -> torch.Tensor:
    """RCCL moves device memory only: a host tensor handed to a collective (results a caller
    produced with numpy, a query batch read from a file) goes to this process's GPU first."""
    if x.device.type == 'cpu' and dist.get_backend(group) == 'nccl':
        return x.cuda()
    return x


def exchange_partials(D: torch.Tensor, I: torch.Tensor, world: int, group=None,
                      async_op: bool = False):
    """[world*n, k] per-shard results for n queries of every rank (rank-major) ->
    [world, n, k] partial lists of THIS rank's n queries. all-to-all on RCCL; on backends
    without all-to-all (gloo) an all-gather followed by a slice.

    With ``async_op`` the RCCL collectives are only enqueued (on RCCL's own stream, after
    the work already queued on the current stream); call ``wait()`` on the returned handles
    before the outputs are consumed. The inputs are kept alive by the returned tuple."""
    nq_all, k = D.shape
    n = nq_all // world
    rank = dist.get_rank(group)
    if _direct(group):
        Do, Io = torch.empty_like(D), torch.empty_like(I)
        w1 = dist.all_to_all_single(Do, D, group=group, async_op=async_op)
        w2 = dist.all_to_all_single(Io, I, group=group, async_op=async_op)
        out = (Do.view(world, n, k), Io.view(world, n, k))
        return out + ([w1, w2], (D, I)) if async_op else out
    dev = D.device
    Dc, Ic = D.cpu(), I.cpu()          # gloo: collectives on host tensors
    Dg = [torch.empty_like(Dc) for _ in range(world)]
    Ig = [torch.empty_like(Ic) for _ in range(world)]
    dist.all_gather(Dg, Dc, group=group)
    dist.all_gather(Ig, Ic, group=group)
    sl = slice(rank * n, (rank + 1) * n)
    out = (torch.stack([d[sl] for d in Dg]).to(dev), torch.stack([i[sl] for i in Ig]).to(dev))
    return out + ([], None) if async_op else out


class CommLog:
    """Bytes this rank hands to every collective of the sharded search, by name (bench.py prints
    it as the ``comm`` block; ``sharded_search_batch(..., comm=CommLog())``)."""

    def __init__(self):
        self.calls = {}

    def add(self, name: str, t: torch.Tensor, world: int, kind: str):
        b = t.numel() * t.element_size()
        # all-to-all: (world - 1) / world of the buffer leaves the rank; all-gather: the buffer goes
        # to every other rank
        out = b * (world - 1) // world if kind == 'all_to_all' else b * (world - 1)
        c = self.calls.setdefault(name, {'kind': kind, 'calls': 0, 'buffer_bytes': 0, 'bytes_out': 0})
        c['calls'] += 1
        c['buffer_bytes'] += b
        c['bytes_out'] += out

    def summary(self, steps: int = 1):
        out = {}
        for name, c in self.calls.items():
            out[name] = {'kind': c['kind'], 'calls_per_step': c['calls'] / steps,
                         'bytes_out_per_rank_per_step': c['bytes_out'] // max(steps, 1),
                         'buffer_bytes_per_call': c['buffer_bytes'] // max(c['calls'], 1)}
        out['total_bytes_out_per_rank_per_step'] = sum(c['bytes_out'] for c in self.calls.values()) // max(steps, 1)
        return out


def _all_to_all(x: torch.Tensor, world: int, group=None, comm: Optional[CommLog] = None, name: str = ''):
    """x [world * m, ...] (block r goes to rank r) -> ([world, m, ...] the blocks the ranks sent
    here, work handles). RCCL: one all_to_all_single, enqueued asynchronously; gloo (CPU tests):
    all-gather of host copies and a slice."""
    m = x.shape[0] // world
    rank = dist.get_rank(group)
    if comm is not None:
        comm.add(name, x, world, 'all_to_all')
    if _direct(group):
        x = _comm_tensor(x, group).contiguous()
        out = torch.empty_like(x)
        w = dist.all_to_all_single(out, x, group=group, async_op=True)
        return out.view((world, m) + tuple(x.shape[1:])), [w], x
    xc = x.cpu().contiguous()
    parts = [torch.empty_like(xc) for _ in range(world)]
    dist.all_gather(parts, xc, group=group)
    sl = slice(rank * m, (rank + 1) * m)
    return torch.stack([p_[sl] for p_ in parts]).to(x.device), [], None


def exchange_keys(K: torch.Tensor, world: int, group=None):
    """Packed-key variant of ``exchange_partials``: ONE all-to-all of 8 bytes per hit.
    Returns ([world, n, k] keys of this rank's n queries, work handles, keep-alive)."""
    nq_all, k = K.shape
    n = nq_all // world
    rank = dist.get_rank(group)
    if _direct(group):
        Ko = torch.empty_like(K)
        w = dist.all_to_all_single(Ko, K, group=group, async_op=True)
        return Ko.view(world, n, k), [w], K
    dev = K.device
    Kc = K.cpu()
    Kg = [torch.empty_like(Kc) for _ in range(world)]
    dist.all_gather(Kg, Kc, group=group)
    sl = slice(rank * n, (rank + 1) * n)
    return torch.stack([g[sl] for g in Kg]).to(dev), [], None


def _all_gather_rows(x: torch.Tensor, world: int, group=None, async_op: bool = False):
    """Concatenate every rank's [n, ...] tensor along dim 0 (rank order). With ``async_op``
    returns (tensor, work-or-None); ``work.wait()`` before the tensor is read."""
    if _direct(group):
        x = _comm_tensor(x, group).contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype,
                          device=x.device)
        w = dist.all_gather_into_tensor(out, x, group=group, async_op=async_op)
        return (out, w) if async_op else out
    parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(world)]
    dist.all_gather(parts, x.cpu().contiguous(), group=group)
    out = torch.cat(parts).to(x.device)
    return (out, None) if async_op else out


PEAK_ROW_ALIGN = 16      # fixed row width of the peak exchange = max peaks rounded up to this


def _peak_row_width(queries: PackedSpectra, world: int, group=None,
                    agreed: Optional[int] = None) -> int:
    """Row width of the peak exchange: the widest spectrum of ANY rank's slice (at least 50:
    the reference's ``max_peaks_used``, spectrum.py:97-99), rounded up to PEAK_ROW_ALIGN. Every
    rank must arrive at the same number or ``all_gather_into_tensor`` runs with mismatched
    shapes, so it comes either from ``agreed`` -- a bound the caller guarantees to be identical
    on every rank (the full batch's widest spectrum in ``sharded_cascade_batch``,
    ``config.max_peaks_used`` for processed queries; a wider local spectrum is an error) -- or
    from an all-reduce(MAX) of the local widths."""
    local = int(queries.max_peaks())
    if agreed is not None:
        if local > agreed:
            raise ValueError(f'peak exchange: a local spectrum has {local} peaks, more than the '
                             f'agreed width {agreed}')
        m = int(agreed)
    elif world > 1:
        t = torch.tensor([local], dtype=torch.int64)
        if _direct(group):
            t = _comm_tensor(t, group)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        m = int(t[0])
    else:
        m = local
    m = max(m, 50)
    return -(-m // PEAK_ROW_ALIGN) * PEAK_ROW_ALIGN


def _all_gather_peaks(queries: PackedSpectra, world: int, group=None, agreed: Optional[int] = None,
                      comm: Optional['CommLog'] = None):
    """All-gather of the local queries' peaks as fixed-width rows [n, 2 W + 1] of 4-byte words
    (W m/z values, W intensities, the peak count). Returns (gathered [world * n, 2 W + 1], work)."""
    W = _peak_row_width(queries, world, group, agreed)
    dev = queries.device
    n = queries.n
    off = queries.offsets.to(torch.int64)
    cnt = (off[1:] - off[:-1])
    slot = torch.arange(W, device=dev).unsqueeze(0)
    have = slot < cnt.unsqueeze(1)
    pos = (off[:-1].unsqueeze(1) + slot).clamp_(max=max(int(queries.mz.numel()) - 1, 0))
    buf = torch.zeros((n, 2 * W + 1), dtype=torch.float32, device=dev)
    if queries.mz.numel():
        buf[:, :W] = torch.where(have, queries.mz[pos], buf[:, :W])
        buf[:, W:2 * W] = torch.where(have, queries.intensity[pos], buf[:, W:2 * W])
    buf[:, 2 * W] = cnt.to(torch.int32).view(torch.float32)          # bit pattern, not a value
    if comm is not None:
        comm.add('query_peaks_all_gather', buf, world, 'all_gather')
    return _all_gather_rows(buf, world, group, async_op=True)


def _unpack_peaks(rows: torch.Tensor, like: PackedSpectra) -> PackedSpectra:
    """Fixed-width peak rows -> PackedSpectra (peaks only: what the encoder reads)."""
    W = (rows.shape[1] - 1) // 2
    n = rows.shape[0]
    dev = rows.device
    cnt = rows[:, 2 * W].contiguous().view(torch.int32).to(torch.int64)
    have = torch.arange(W, device=dev).unsqueeze(0) < cnt.unsqueeze(1)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(cnt, 0)
    return PackedSpectra(offsets.to(torch.int32), rows[:, :W][have], rows[:, W:2 * W][have],
                         torch.zeros(int(offsets[-1]), dtype=torch.uint8, device=dev),
                         torch.zeros(n, dtype=torch.float64, device=dev),
                         torch.zeros(n, dtype=torch.int32, device=dev))


def _concat_results(parts):
    """Concatenate per-chunk results (``BatchResult`` of the HIP backend or the dictionaries of
    a test backend) along the query axis; peak-match tables are padded to the widest chunk."""
    if len(parts) == 1:
        return parts[0]
    cat = lambda xs: torch.cat(xs) if torch.is_tensor(xs[0]) else np.concatenate(xs)
    first = parts[0]
    if isinstance(first, dict):
        return {k: cat([p[k] for p in parts]) for k in first}
    stride = max(p.pm_pairs.shape[1] for p in parts)

    def pad(pp):
        if pp.shape[1] == stride:
            return pp
        if torch.is_tensor(pp):
            out = torch.zeros((pp.shape[0], stride, 2), dtype=pp.dtype, device=pp.device)
        else:
            out = np.zeros((pp.shape[0], stride, 2), pp.dtype)
        out[:, :pp.shape[1]] = pp
        return out
    return type(first)(cat([p.best_row for p in parts]), cat([p.best_score for p in parts]),
                       cat([p.n_candidates for p in parts]), cat([p.pm_count for p in parts]),
                       cat([pad(p.pm_pairs) for p in parts]),
                       None if first.knn is None else cat([p.knn for p in parts]))


def head_width(k: int, world: int, head_keys: Optional[int] = None) -> int:
    """Row width kp of the phase-1 exchange: ``min(k, ceil(2 k / world))`` key slots (or
    ``head_keys``) + the slot of the best held-back key."""
    keys = head_keys if head_keys is not None else -(-2 * k // max(world, 1))
    return max(1, min(int(keys), k)) + 1


def shard_k(k: int, world: int) -> int:
    """The shards' own k (``asl_shard_k``; completed by second scans on the shard, module docstring 5): k / 2 from 8 ranks on, 5 k / 8 from
    4, rounded up to 64; k below 4 ranks or when that is not more than a head's key slots."""
    return int(_lib.lib().asl_shard_k(int(k), int(world)))


def _agreed_keys(backend, world: int, group, k_scan: int) -> bool:
    """Can EVERY rank's shard emit packed keys? For IVF-Flat that depends on the vectors a shard
    holds (an empty or dense shard scans dense rows), and ranks that disagreed would run
    collectives of different shapes: all-reduce(MIN), once per (backend, group, k) -- the answer
    only changes with the index."""
    local = bool(getattr(backend, 'supports_keys', False))
    if world == 1:
        return local
    cache = backend.__dict__.setdefault('_keys_agreed', {}) if hasattr(backend, '__dict__') else {}
    # the group object itself is part of the key (held, so its id cannot be recycled); index_epoch
    # changes with every call that changes the index or how it is scanned -- the same on every rank
    key = (group, world, k_scan, getattr(backend, 'index_epoch', 0))
    if key not in cache:
        t = torch.tensor([int(local)], dtype=torch.int32)
        if _direct(group):
            t = _comm_tensor(t, group)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        cache[key] = bool(int(t.item()))
    return cache[key]


def sharded_search_batch(backend, queries_local: PackedSpectra, group=None, device_out=False,
                         chunks: Optional[int] = None, _force_exchange: bool = False,
                         pm_stride: Optional[int] = None, check_sizes: bool = False,
                         peak_width: Optional[int] = None, two_phase: Optional[bool] = None,
                         head_keys: Optional[int] = None, extras_per_query: Optional[int] = None,
                         comm: Optional['CommLog'] = None, stats: Optional[dict] = None,
                         shard_keys: Optional[int] = None, entry_lists: Optional[bool] = None):
    """One batch: ``queries_local`` is this rank's equally sized slice of the global
    batch. Returns the BatchResult of the local slice (library rows are global).
    ``entry_lists=False`` keeps the other ranks' queries as dense rows (default: entry lists
    whenever the packed-key scans run and no spectrum has more than 64 peaks).
    ``peak_width``: a bound on the peaks per query that is IDENTICAL on every rank (e.g.
    ``config.max_peaks_used`` for processed queries); without it the ranks agree on the row
    width of the peak exchange by an all-reduce.

    The shard scan runs in ``chunks`` pieces (the same sub-slice of every rank's queries per
    piece) so that the collectives of one piece travel over xGMI while the next piece is being
    scanned: every piece walks through the stages head exchange -> merge + bounds -> held-back
    keys -> final merge + rescoring, one stage per scan that is issued behind it.

    ``two_phase`` (default: whenever every rank's backend emits packed keys): the exact
    exchange of csrc/exchange.hip; ``head_keys`` overrides the keys per head (default
    ``ceil(2 k / world)``), ``extras_per_query`` the capacity of the phase-2 buffers (slots per
    query and pair of ranks, default ``max(8, k // 16)``). With heads as wide as the rows (two
    ranks) the rows travel whole. ``shard_keys``: the shards' own k (default ``shard_k(k,
    world)``; below k the shards scan again, with the full k, the rows whose bound asks for it).
    A full answer buffer -- or more second scans than a piece has room for -- repeats the batch
    with the full exchange of k-deep rows (``stats['fallback']``). ``comm``: a ``CommLog`` that
    receives the bytes of every collective; ``stats`` also gets ``shard_k`` and
    ``third_phase_queries`` (the (query, owner) rows this shard scanned a second time)."""
    world = dist.get_world_size(group)
    if queries_local.n == 0:
        # every rank must bring the same, non-zero number of queries (the collectives below are
        # fixed-shape); callers with ragged batches pad -- see sharded_cascade_batch
        raise ValueError('sharded_search_batch: empty local slice (pad ragged batches)')
    if getattr(backend, 'device', None) is not None:
        queries_local = queries_local.to(backend.device)     # (the peaks travel from device memory)
    if check_sizes and world > 1:
        t = torch.tensor([queries_local.n, -queries_local.n], dtype=torch.int64)
        if _direct(group):
            t = _comm_tensor(t, group)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        if int(t[0]) != -int(t[1]):
            raise ValueError('sharded_search_batch: ranks hold slices of different sizes')
    # one peak-match table width for every chunk of the slice
    kw = {'pm_stride': pm_stride or queries_local.max_peaks()}
    vec = backend.encode(queries_local)
    if world == 1 and not _force_exchange:   # (_force_exchange: tests drive the collectives at world 1)
        D, I = backend.shard_search(vec)
        return backend.rescore_knn(queries_local, I, device_out, **kw)
    n_local = vec.shape[0]
    # every rank needs the hashed vectors of ALL queries for its shard scan. A query is <= ~50
    # peaks (8 B each) but 800 floats once hashed, so the PEAKS travel (fixed-width rows,
    # ~0.5 KB per query instead of 3.2 KB) while the coarse quantiser runs on the own slice, and
    # every rank hashes the foreign queries itself (encode: ~6 us per 1000 queries)
    packed, w_vec = _all_gather_peaks(queries_local, world, group, peak_width, comm)
    co = backend.coarse(vec) if getattr(backend, 'supports_preassigned', False) else None
    if co is not None:
        if comm is not None:
            comm.add('probe_lists_all_gather', co[0], world, 'all_gather')
            comm.add('probe_lists_all_gather', co[1], world, 'all_gather')
        cD, cI = _all_gather_rows(co[0], world, group), _all_gather_rows(co[1], world, group)
    k_scan = int(getattr(backend, 'k_scan', getattr(backend, 'k', 0)))
    # the exchange format is a COLLECTIVE choice (see _agreed_keys)
    use_keys = _agreed_keys(backend, world, group, k_scan) and \
        (co is not None or not getattr(backend, 'supports_preassigned', False))
    if w_vec is not None:
        w_vec.wait()
    rank = dist.get_rank(group)
    # The packed-key scans read their queries as ENTRY LISTS (the non-zero components), so the other
    # ranks' queries are hashed straight into that form -- no 3.2 KB dense row per query written,
    # listed and gathered per piece. A vector has no more non-zeros than its spectrum has peaks: the
    # (agreed) row width of the peak exchange decides, identically on every rank.
    use_entries = (entry_lists is not False and use_keys and hasattr(backend, 'encode_entries')
                   and (packed.shape[1] - 1) // 2 <= getattr(backend, 'MAX_ENTRIES', 0)
                   and k_scan == int(getattr(backend, 'k', k_scan)))
    if use_entries:
        allvec = backend.encode_entries(_unpack_peaks(packed, queries_local))
    else:
        allvec = backend.encode(_unpack_peaks(packed, queries_local))
        allvec[rank * n_local:(rank + 1) * n_local] = vec      # (same bits; skips nothing, documents intent)
    if stats is not None:
        stats['query_form'] = 'entry lists' if use_entries else 'dense rows'
    if chunks is None:
        chunks = 4 if _direct(group) else 2
    chunks = max(1, min(chunks, n_local))
    bounds = [(n_local * c) // chunks for c in range(chunks + 1)]
    rank_base = torch.arange(world, device=vec.device).unsqueeze(1) * n_local
    if two_phase is None:
        two_phase = use_keys and hasattr(backend, 'keys_split')
    two_phase = bool(two_phase and use_keys)
    kp = head_width(k_scan, world, head_keys) if two_phase else 0
    if two_phase and kp - 1 >= k_scan:       # (two ranks: a head of ceil(2k / 2) keys IS the row) the
        two_phase, kp = False, 0             # rows travel whole: no split, no bound, one merge
    # the shards' own k: rows of k_row < k_scan keys, completed by second scans where a bound asks
    k_row = k_scan
    if two_phase:
        ks = int(shard_keys) if shard_keys is not None else shard_k(k_scan, world)
        if kp - 1 < ks < k_scan:
            k_row = ks
    second = two_phase                       # heads hold something back
    flag = backend.new_flag() if second else None

    def rescore(lo, hi, knn):
        if hasattr(backend, 'refine'):       # exact re-rank of the merged short-list (k' -> k)
            knn = backend.refine(vec[lo:hi], knn)
        sub = queries_local if (lo, hi) == (0, n_local) else queries_local.select(
            torch.arange(lo, hi, device=queries_local.device))
        return backend.rescore_knn(sub, knn, device_out, **kw)

    def scan(lo, hi):
        """shard scan of rows [lo, hi) of every rank's slice, rank-major -> (rows, its inputs)"""
        if chunks == 1:
            xv, pre = allvec, (cD, cI) if co is not None else None
        else:
            rows = (rank_base + torch.arange(lo, hi, device=vec.device).unsqueeze(0)).reshape(-1)
            xv = allvec.index_select(0, rows)
            pre = (cD.index_select(0, rows), cI.index_select(0, rows)) if co is not None else None
        if use_keys:
            pre_ = pre if pre is not None else (None, None)
            if two_phase and k_row < k_scan:
                return backend.shard_search_keys(xv, *pre_, k=k_row), (xv,) + tuple(pre_)
            return backend.shard_search_keys(xv, *pre_), None
        return (backend.shard_search_preassigned(xv, *pre) if pre is not None
                else backend.shard_search(xv)), None

    def wait(works):
        for w in works:
            w.wait()

    # every piece is a little state machine; `step` advances it by one stage and returns True
    # when its result has been appended
    def piece(lo, hi, out, inputs):
        st = {'stage': 0}

        def step():
            n = hi - lo
            if st['stage'] == 0:
                if not two_phase:             # the full rows travel (one collective)
                    if use_keys:
                        st['x'] = _all_to_all(out, world, group, comm, 'topk_rows_all_to_all')
                    else:
                        if comm is not None:
                            comm.add('topk_rows_all_to_all', out[0], world, 'all_to_all')
                            comm.add('topk_rows_all_to_all', out[1], world, 'all_to_all')
                        st['x'] = exchange_partials(out[0], out[1], world, group, async_op=True)
                    st['stage'] = 10
                    return False
                if inputs is not None:        # rows of k_row < k keys: keep M and the scan's inputs
                    head, floor, rowmin = backend.keys_split(out, kp, True)
                    st['rescan'] = (rowmin,) + inputs + (k_scan,)
                else:
                    head, floor = backend.keys_split(out, kp)
                    st['rescan'] = None
                st['rest'] = (out, floor)          # the held-back keys stay in the scan's rows
                st['x'] = _all_to_all(head, world, group, comm, 'heads_all_to_all')
                st['stage'] = 1
                return False
            if st['stage'] == 10:             # full exchange: merge, rescore
                if use_keys:
                    wait(st['x'][1])
                    knn = backend.merge_keys(st['x'][0].contiguous())[1]
                else:
                    wait(st['x'][2])
                    knn = backend.merge(st['x'][0].contiguous(), st['x'][1].contiguous())[1]
                results.append((lo, rescore(lo, hi, knn)))
                return True
            if st['stage'] == 1:              # heads are here: merge, bound, questions to the shards
                wait(st['x'][1])
                st['heads'] = st['x'][0].contiguous()
                st['keys'], bnd, st['need'] = backend.keys_merge_heads(st['heads'], k_scan)
                if not second:
                    knn = backend.keys_merge_final(st['heads'], None, st['keys'], st['need'], k_scan)
                    results.append((lo, rescore(lo, hi, knn)))
                    return True
                st['x'] = _all_to_all(bnd.reshape(world * n), world, group, comm, 'bounds_all_to_all')
                st['stage'] = 2
                return False
            if st['stage'] == 2:              # the owners' bounds are here: what lies above them outside the heads
                wait(st['x'][1])
                xcap = n * (extras_per_query if extras_per_query is not None else max(8, k_scan // 16))
                bnd_in = st['x'][0].reshape(world * n).contiguous()
                if st['rescan'] is not None:
                    xbuf = backend.keys_extras(st['rest'][0], st['rest'][1], bnd_in, world, xcap, flag,
                                               rescan=st['rescan'])
                else:
                    xbuf = backend.keys_extras(st['rest'][0], st['rest'][1], bnd_in, world, xcap, flag)
                st['rest'] = st['rescan'] = None
                st['x'] = _all_to_all(xbuf.reshape(world * (n + xcap)), world, group, comm,
                                      'held_back_keys_all_to_all')
                st['xcap'] = xcap
                st['stage'] = 3
                return False
            wait(st['x'][1])                  # stage 3: final merge, rescoring
            xr = st['x'][0].reshape(world, n + st['xcap']).contiguous()
            knn = backend.keys_merge_final(st['heads'], xr, st['keys'], st['need'], k_scan)
            results.append((lo, rescore(lo, hi, knn)))
            return True
        return step

    def run():
        pending = []
        for c in range(chunks):
            lo, hi = bounds[c], bounds[c + 1]
            if hi == lo:
                continue
            out, inputs = scan(lo, hi)
            for p_ in list(pending):          # older pieces: one stage each, behind this scan
                if p_():
                    pending.remove(p_)
            new = piece(lo, hi, out, inputs)
            new()                             # split + first collective, right behind its scan
            pending.append(new)
        while pending:
            for p_ in list(pending):
                if p_():
                    pending.remove(p_)

    results = []
    run()
    fallback = False
    rescans = 0
    if second:
        # one pair of ints per batch: did any buffer run full anywhere (answers of phase 2, second
        # scans of a piece), and how many rows did the shards scan a second time?
        f = flag if _direct(group) else flag.cpu()
        full = f[:1].clone()
        dist.all_reduce(full, op=dist.ReduceOp.MAX, group=group)
        full, mine = torch.cat([full, f[1:2]]).tolist()         # the batch's one host round trip
        rescans = int(mine) if k_row < k_scan else 0
        if int(full):
            fallback = True
            if stats is not None:
                stats['fallback'] = stats.get('fallback', 0) + 1
            two_phase = second = False
            results = []
            run()
    if stats is not None:
        stats['two_phase'] = bool(kp) and not fallback
        stats['exchange_used'] = ('full rows (fallback)' if fallback else
                                  'two-phase, shard-side k + second scans' if kp and k_row < k_scan
                                  else 'two-phase' if kp else 'full rows')
        stats['head_width'] = kp
        stats['shard_k'] = k_row if kp else k_scan
        stats['third_phase_queries'] = stats.get('third_phase_queries', 0) + rescans
    results.sort(key=lambda t: t[0])
    return _concat_results([r for _, r in results])


# ---------------------------------------------------------------------------- cascade batches
_RESULT_FIELDS = ('best_row', 'best_score', 'n_candidates', 'pm_count', 'pm_pairs')


def _result_field(res, name):
    v = res[name] if isinstance(res, dict) else getattr(res, name)
    if torch.is_tensor(v):
        return v
    v = np.ascontiguousarray(v)
    return torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v)   # no u32 collectives


def sharded_cascade_batch(backend, queries: PackedSpectra, mode: str, use_ann: bool, group=None):
    """One batch of one cascade level on ``world`` ranks (configs[4]: the second, open pass of
    the cascade over the list-sharded library; reference: ``_search_cascade`` ->
    ``_search_batch``, /root/reference/src/ann_solo/spectral_library.py:301-317,328-370).

    Every rank holds the SAME ``queries`` (the whole batch -- each process reads the query
    file, as the reference's single process does) and answers for its own contiguous slice of
    ``ceil(n / world)`` rows; a ragged tail is padded by repeating the last query, and the
    padding rows are dropped after the gather:

      * ``use_ann`` (open search on a charge with an ANN index): ``sharded_search_batch`` --
        the list-sharded scan + exchange + merge, then rescoring of the own slice;
      * otherwise (standard search, brute-force mode, charges too rare for an index): the
        precursor-window search is data-parallel over the queries against the replicated
        peak store -- no data-path collective.

    The per-query results of all slices are all-gathered (about 0.5 KB per query), so every
    rank continues the cascade (FDR filter, remaining queries) on identical data, exactly as a
    single process would. Returns a ``BatchResult`` (numpy) of ``queries.n`` rows."""
    from .spectral_library import BatchResult
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = queries.n
    if n == 0:
        raise ValueError('sharded_cascade_batch: empty batch')
    n_local = -(-n // world)
    rows = torch.arange(rank * n_local, (rank + 1) * n_local).clamp_(max=n - 1)
    q_loc = queries.select(rows.to(queries.device))
    stride = queries.max_peaks()
    if use_ann:
        # (every rank holds the whole batch: its widest spectrum is the same number everywhere)
        res = sharded_search_batch(backend, q_loc, group=group, device_out=True, pm_stride=stride,
                                   peak_width=stride)
    else:
        res = backend.window_search(q_loc, mode, pm_stride=stride)
    out = {}
    for name in _RESULT_FIELDS:
        t = _result_field(res, name)
        if name == 'pm_pairs' and t.shape[1] != stride:      # a backend that ignored pm_stride
            p = torch.zeros((t.shape[0], stride, 2), dtype=t.dtype, device=t.device)
            p[:, :min(stride, t.shape[1])] = t[:, :stride]
            t = p
        out[name] = _all_gather_rows(t, world, group)[:n].cpu().numpy()
    pm = out['pm_pairs']
    return BatchResult(out['best_row'], out['best_score'], out['n_candidates'], out['pm_count'],
                       pm.view(np.uint32) if pm.dtype == np.int32 else pm, None)
