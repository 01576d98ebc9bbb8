"""FAISS-shaped facade over the hand-written gfx950 IVF index (``asl_index_*``).

Covers exactly the FAISS surface the reference touches
(/root/reference/src/ann_solo/spectral_library.py:73-87,167-181,191,443-445,487-497):
``IndexFlatIP``, ``IndexIVFFlat``, ``METRIC_INNER_PRODUCT``, ``train``, ``add``,
``search``, ``nprobe``, ``reset``, ``write_index``/``read_index``, ``get_num_gpus``,
``StandardGpuResources``, ``GpuClonerOptions``, ``index_cpu_to_gpu``/``setNumProbes`` --
plus ``IndexIVFPQ`` (the north star's addition). ``import ann_solo_amd.faiss_compat as
faiss`` is the whole change at those call sites. Indexes always live on the GPU.
"""
import ctypes as C
import os
import tempfile

import numpy as np

from . import _lib

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
_KIND_FLAT, _KIND_IVFFLAT, _KIND_IVFPQ = 0, 1, 2
DEFAULT_SEED = 1234          # FAISS ClusteringParameters.seed


def get_num_gpus() -> int:
    return _lib.lib().asl_get_num_gpus()


class StandardGpuResources:
    """Placeholder: device memory is managed inside libannsolo_mi."""


class GpuClonerOptions:
    useFloat16 = False


def _as_f32(x, d):
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
    elif x.dtype.is_floating_point and str(x.dtype) != 'torch.float32':
        x = x.float()
    if x.ndim != 2 or x.shape[1] != d:
        raise ValueError(f'expected an [n, {d}] float32 matrix')
    return x if isinstance(x, np.ndarray) else x.contiguous()


class Index:
    def __init__(self, handle, d):
        if not handle:
            _lib.check(-1 if not _lib.lib().asl_last_error() else -3)
        self._h = C.c_void_p(handle)
        self.d = d
        self.nprobe = 1
        self.seed = DEFAULT_SEED
        # bumped by every call that can change what a search over this handle does (contents, layout,
        # scan variant, storage, sharding): SPMD drivers make these calls on every rank alike, so the
        # counter is the same everywhere and keys what the ranks agreed on (distributed._agreed_keys)
        self.epoch = 0

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.lib().asl_index_free(h)
            except Exception:
                pass

    @property
    def ntotal(self) -> int:
        return _lib.lib().asl_index_ntotal(self._h)

    @property
    def is_trained(self) -> bool:
        return bool(_lib.lib().asl_index_is_trained(self._h))

    def info(self):
        i = _lib.AslIndexInfo()
        _lib.check(_lib.lib().asl_index_info(self._h, C.byref(i)))
        return i

    def set_niter(self, niter: int):
        _lib.check(_lib.lib().asl_index_set_niter(self._h, int(niter)))

    def set_scan_variant(self, variant: int):
        self.epoch += 1
        _lib.check(_lib.lib().asl_index_set_scan_variant(self._h, int(variant)))

    def train(self, x):
        self.epoch += 1
        x = _as_f32(x, self.d)
        _lib.check(_lib.lib().asl_index_train(self._h, x.shape[0], _lib.ptr(x), self.seed))

    def add(self, x):
        self.epoch += 1
        x = _as_f32(x, self.d)
        _lib.check(_lib.lib().asl_index_add(self._h, x.shape[0], _lib.ptr(x)))

    def search(self, x, k, D=None, I=None):
        """(D float32[nq,k], I int64[nq,k]); rows by (score desc, id asc), -1 padded.
        numpy in -> numpy out; pass torch device tensors for ``x``/``D``/``I`` to stay in HBM."""
        x = _as_f32(x, self.d)
        nq = x.shape[0]
        if D is None and I is None:
            if isinstance(x, np.ndarray):
                D = np.empty((nq, k), np.float32)
                I = np.empty((nq, k), np.int64)
            else:
                import torch
                D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
                I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        _lib.check(_lib.lib().asl_index_search(self._h, nq, _lib.ptr(x), int(k), int(self.nprobe),
                                               _lib.ptr(D), _lib.ptr(I)))
        return D, I

    def reset(self):
        self.epoch += 1
        _lib.check(_lib.lib().asl_index_reset(self._h))

    def setNumProbes(self, nprobe: int):      # GpuIndexIVF spelling (spectral_library.py:495)
        self.nprobe = int(nprobe)

    # ---- introspection / sharding (not part of FAISS) ----
    def centroids(self):
        i = self.info()
        out = np.empty((i.nlist, i.d), np.float32)
        _lib.check(_lib.lib().asl_index_get_centroids(self._h, _lib.ptr(out)))
        return out

    def codebooks(self):
        i = self.info()
        out = np.empty((i.pq_m, i.pq_ksub, i.pq_dsub), np.float32)
        _lib.check(_lib.lib().asl_index_get_codebooks(self._h, _lib.ptr(out)))
        return out

    def set_trained(self, centroids, codebooks=None):
        self.epoch += 1
        c = np.ascontiguousarray(centroids, np.float32)
        cb = None if codebooks is None else np.ascontiguousarray(codebooks, np.float32)
        _lib.check(_lib.lib().asl_index_set_trained(self._h, _lib.ptr(c), _lib.ptr(cb)))

    def lists(self):
        """(list_offsets[nlist+1], ids[nlocal], payload) in inverted-list order."""
        i = self.info()
        off = np.empty(i.nlist + 1, np.int32)
        ids = np.empty(i.nlocal, np.int32)
        codes = vecs = None
        if i.kind == _KIND_IVFPQ:
            codes = np.empty((i.nlocal, i.pq_m), np.uint8)
        else:
            vecs = np.empty((i.nlocal, i.d), np.float32)
        _lib.check(_lib.lib().asl_index_get_lists(self._h, _lib.ptr(off), _lib.ptr(ids),
                                                  _lib.ptr(codes), _lib.ptr(vecs)))
        return off, ids, codes if codes is not None else vecs

    def search_preassigned(self, x, k, coarse_D, coarse_I):
        """IndexIVF.search_preassigned: scan with caller-supplied probe lists
        (``coarse_D``/``coarse_I`` [nq, nprobe] as ``coarse()`` returns them)."""
        x = _as_f32(x, self.d)
        nq, nprobe = coarse_I.shape
        if isinstance(x, np.ndarray):
            D = np.empty((nq, k), np.float32)
            I = np.empty((nq, k), np.int64)
        else:
            import torch
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        _lib.check(_lib.lib().asl_index_search_preassigned(
            self._h, nq, _lib.ptr(x), int(k), int(nprobe), _lib.ptr(coarse_D),
            _lib.ptr(coarse_I), _lib.ptr(D), _lib.ptr(I)))
        return D, I

    def coarse(self, x, nprobe):
        x = _as_f32(x, self.d)
        nprobe = min(nprobe, self.info().nlist)
        if not isinstance(x, np.ndarray):
            import torch
            D = torch.empty((x.shape[0], nprobe), dtype=torch.float32, device=x.device)
            I = torch.empty((x.shape[0], nprobe), dtype=torch.int32, device=x.device)
            _lib.check(_lib.lib().asl_index_coarse(self._h, x.shape[0], _lib.ptr(x), nprobe,
                                                   _lib.ptr(D), _lib.ptr(I)))
            return D, I
        D = np.empty((x.shape[0], nprobe), np.float32)
        I = np.empty((x.shape[0], nprobe), np.int32)
        _lib.check(_lib.lib().asl_index_coarse(self._h, x.shape[0], _lib.ptr(x), nprobe,
                                               _lib.ptr(D), _lib.ptr(I)))
        return D, I

    def pq_lut(self, x):
        x = _as_f32(x, self.d)
        i = self.info()
        lut = np.empty((x.shape[0], i.pq_m, i.pq_ksub), np.float32)
        _lib.check(_lib.lib().asl_index_pq_lut(self._h, x.shape[0], _lib.ptr(x), _lib.ptr(lut)))
        return lut

    def postings_work(self, x, nprobe=None):
        """(algorithmic bytes, 128-byte lines) the IVF-Flat postings scan needs for these queries
        (``asl_index_postings_work``: measurement, for the kernel's roofline)."""
        x = _as_f32(x, self.d)
        b, l = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().asl_index_postings_work(self._h, x.shape[0], _lib.ptr(x),
                                                      int(nprobe or self.nprobe), C.byref(b), C.byref(l)))
        return b.value, l.value

    def set_storage(self, storage: str):
        """IVF-Flat component storage, before the first ``add``: 'fx22' | 'fp32'."""
        self.epoch += 1
        mode = {'fx22': 0, 'fp32': 1}[storage]
        _lib.check(_lib.lib().asl_index_set_flat_storage(self._h, mode))

    @property
    def storage(self) -> str:
        return ('fx22', 'fp32')[_lib.lib().asl_index_get_flat_storage(self._h)]

    @property
    def flat_layout(self) -> int:
        """IVF-Flat scan layout the stored data allowed: 0 dense rows only, 1 float postings,
        2 fixed-point postings (builds the lists if they are stale)."""
        r = _lib.lib().asl_index_flat_layout(self._h)
        if r < 0:
            _lib.check(r)
        return r

    def set_refine(self, kprime: int):
        """IVF-PQ: re-rank the ``kprime`` best ADC candidates with the exact inner product and
        return the k best (FAISS ``IndexRefineFlat``); call before ``add``. 0 switches it off."""
        self.epoch += 1
        _lib.check(_lib.lib().asl_index_set_refine(self._h, int(kprime)))

    @property
    def refine_k(self) -> int:
        """k' of the exact re-rank (0: off) -- also for an index read from a file."""
        return int(_lib.lib().asl_index_get_refine(self._h))

    def refine(self, x, I_in, k):
        """Exact re-rank of a short-list obtained elsewhere: (D [nq,k], I [nq,k])."""
        x = _as_f32(x, self.d)
        nq, kp = I_in.shape
        if isinstance(x, np.ndarray):
            I_in = np.ascontiguousarray(I_in, np.int64)
            D, I = np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)
        else:
            import torch
            I_in = I_in.to(torch.int64).contiguous()
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        _lib.check(_lib.lib().asl_index_refine(self._h, nq, _lib.ptr(x), int(kp), _lib.ptr(I_in),
                                               int(k), _lib.ptr(D), _lib.ptr(I)))
        return D, I

    def set_unordered(self, mode=True):
        """Result rows as exact top-k SETS (unspecified order, no final sort). ``mode`` 2 packs
        every hit into one 64-bit key in the id output (see ``search_preassigned_keys``)."""
        _lib.check(_lib.lib().asl_index_set_unordered(self._h, int(mode)))

    def search_preassigned_keys(self, x, k, coarse_D, coarse_I):
        """Like ``search_preassigned`` but returns ONE int64 array [nq, k] of packed hits
        (order-preserving score bits << 32 | ~id, 0 = empty, unspecified order): 8 bytes per
        hit for the exchange of a sharded search; merge with ``topk_merge_keys``."""
        x = _as_f32(x, self.d)
        nq, nprobe = coarse_I.shape
        if isinstance(x, np.ndarray):
            K = np.empty((nq, k), np.int64)
        else:
            import torch
            K = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        self.set_unordered(2)
        try:
            _lib.check(_lib.lib().asl_index_search_preassigned(
                self._h, nq, _lib.ptr(x), int(k), int(nprobe), _lib.ptr(coarse_D),
                _lib.ptr(coarse_I), None, _lib.ptr(K)))
        finally:
            self.set_unordered(0)
        return K

    def search_entries_keys(self, entries, counts, k, coarse_D, coarse_I, gate=None):
        """``search_preassigned_keys`` with the queries as ENTRY LISTS (``encode_entries``: entries
        [nq, 64, 2] int32 = (dimension * 128, value bits), counts [nq] int32) instead of dense rows
        -- device tensors only; the rows equal the dense call's bit for bit. ``gate``: a device int,
        only the first ``gate[0]`` rows are searched (the others are left as they are)."""
        import torch
        nq, nprobe = coarse_I.shape
        K = torch.empty((nq, k), dtype=torch.int64, device=entries.device)
        self.set_unordered(2)
        try:
            _lib.check(_lib.lib().asl_index_search_entries(
                self._h, nq, _lib.ptr(entries), _lib.ptr(counts), int(k), int(nprobe), _lib.ptr(coarse_D),
                _lib.ptr(coarse_I), None, _lib.ptr(K), _lib.ptr(gate)))
        finally:
            self.set_unordered(0)
        return K

    def shard(self, rank: int, world: int):
        self.epoch += 1
        _lib.check(_lib.lib().asl_index_shard(self._h, int(rank), int(world)))

    def shard_map(self, world: int):
        owner = np.empty(self.info().nlist, np.int32)
        _lib.check(_lib.lib().asl_index_shard_map(self._h, int(world), _lib.ptr(owner)))
        return owner


class IndexFlatIP(Index):
    def __init__(self, d):
        super().__init__(_lib.lib().asl_index_create(d, 0, _KIND_FLAT, 0, 0), d)


class IndexIVFFlat(Index):
    """``storage``: 'fp32' (default: the components as given -- what FAISS' CPU ``IndexIVFFlat``
    of the reference stores, spectral_library.py:174-181) or 'fx22' (opt-in: ``add`` stores every
    component in [0, 1) as the nearest multiple of 2^-22, which lets the inverted lists hold
    4-byte postings); ``asl_index_set_flat_storage``. Not a FAISS argument (the GPU clone the
    reference makes stores float16, spectral_library.py:490-497)."""

    def __init__(self, quantizer, d, nlist, metric=METRIC_INNER_PRODUCT, storage='fp32'):
        if metric != METRIC_INNER_PRODUCT:
            raise ValueError('only METRIC_INNER_PRODUCT is implemented (the reference uses no other)')
        self.quantizer = quantizer
        self.nlist = nlist
        super().__init__(_lib.lib().asl_index_create(d, nlist, _KIND_IVFFLAT, 0, 0), d)
        self.set_storage(storage)


class IndexIVFPQ(Index):
    def __init__(self, quantizer, d, nlist, m, nbits=8, metric=METRIC_INNER_PRODUCT):
        if metric != METRIC_INNER_PRODUCT:
            raise ValueError('only METRIC_INNER_PRODUCT is implemented')
        self.quantizer = quantizer
        self.nlist = nlist
        super().__init__(_lib.lib().asl_index_create(d, nlist, _KIND_IVFPQ, m, nbits), d)


def write_index(index: Index, path: str):
    # private name + rename: concurrent ranks write the same cache file, readers never see a
    # truncated one
    path = str(path)
    fd, tmp = tempfile.mkstemp(dir=os.path.dirname(os.path.abspath(path)) or '.',
                               prefix=os.path.basename(path) + '.tmp')   # unique across hosts sharing the directory
    os.close(fd)
    try:
        _lib.check(_lib.lib().asl_index_save(index._h, tmp.encode()))
        os.chmod(tmp, 0o644)     # mkstemp creates 0600
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def read_index(path: str) -> Index:
    h = _lib.lib().asl_index_load(str(path).encode())
    if not h:
        _lib.check(-5)
    idx = Index.__new__(Index)
    Index.__init__(idx, h, 0)
    idx.d = idx.info().d
    return idx


# ---------------------------------------------------------------------------- FAISS files
# FAISS' own on-disk layout of an IndexIVFFlat over an IndexFlatIP / IndexFlatL2 quantiser, as
# faiss/impl/index_write.cpp (v1.5 .. 1.8) writes it -- the files the reference caches as
# <library>_<hash7>_<charge>.idxann (spectral_library.py:181). Little endian:
#   'IwFl' | header | nlist u64 | nprobe u64 | quantiser index | direct map | inverted lists
#   header    = d i32 | ntotal i64 | 1<<20 i64 | 1<<20 i64 | is_trained u8 | metric i32 (0 IP, 1 L2)
#   quantiser = 'IxFI' ('IxF2') | header | n_floats u64 | centroids f32[nlist * d]
#   direct map= type u8 (0 none) | n u64 | i64[n]
#   lists     = 'ilar' | nlist u64 | code_size u64 (= 4 d) | 'full' | n u64 | sizes u64[nlist]
#               (or 'sprs' | n u64 | (list, size) u64 pairs) | per non-empty list: codes, ids i64
# FAISS itself is not installed here, so the layout is restated from its published source and
# checked by round trips and a byte-level fixture only (tests/test_gpu_faiss_file.py).
def _fourcc(s):
    return s.encode('ascii')


def write_index_faiss(index: Index, path: str):
    """Write an IVF-Flat index in FAISS' format (``faiss.read_index`` layout). Row ids are the
    add-order ids, as FAISS assigns them."""
    import struct
    i = index.info()
    if i.kind != _KIND_IVFFLAT or i.shard_world != 1:
        raise ValueError('write_index_faiss: an unsharded IVF-Flat index is required')
    cen = index.centroids()
    off, ids, vecs = index.lists()
    hdr = lambda d, n: struct.pack('<iqqqBi', d, n, 1 << 20, 1 << 20, 1, METRIC_INNER_PRODUCT)
    with open(path, 'wb') as f:
        f.write(_fourcc('IwFl') + hdr(i.d, i.ntotal) + struct.pack('<QQ', i.nlist, max(1, int(index.nprobe))))
        f.write(_fourcc('IxFI') + hdr(i.d, i.nlist) + struct.pack('<Q', cen.size))
        f.write(np.ascontiguousarray(cen, '<f4').tobytes())
        f.write(struct.pack('<BQ', 0, 0))                                   # no direct map
        sizes = np.diff(off).astype('<u8')
        f.write(_fourcc('ilar') + struct.pack('<QQ', i.nlist, 4 * i.d))
        if int((sizes > 0).sum()) > i.nlist // 2:
            f.write(_fourcc('full') + struct.pack('<Q', i.nlist) + sizes.tobytes())
        else:
            nz = np.nonzero(sizes)[0]
            pairs = np.stack([nz.astype('<u8'), sizes[nz]], 1)
            f.write(_fourcc('sprs') + struct.pack('<Q', pairs.size) + pairs.tobytes())
        for l in range(i.nlist):
            a, b = int(off[l]), int(off[l + 1])
            if b > a:
                f.write(np.ascontiguousarray(vecs[a:b], '<f4').tobytes())
                f.write(ids[a:b].astype('<i8').tobytes())


def parse_index_faiss(path: str) -> dict:
    """The content of a FAISS IVF-Flat / inner-product file as host arrays (no device involved):
    ``d, nlist, nprobe, ntotal, centroids [nlist, d], x [ntotal, d]`` (vector of id i in row i),
    ``lists [ntotal]`` (the inverted list FAISS filed every vector under). Raises ValueError on
    anything that is not such a file, damaged or not."""
    import struct
    with open(path, 'rb') as f:
        buf = f.read()
    pos = [0]

    def take(fmt):
        v = struct.unpack_from('<' + fmt, buf, pos[0])
        pos[0] += struct.calcsize('<' + fmt)
        return v if len(v) > 1 else v[0]

    def cc():
        v = buf[pos[0]:pos[0] + 4]
        pos[0] += 4
        return v

    def arr(dt, count):
        n = np.dtype(dt).itemsize * count
        if pos[0] + n > len(buf):
            raise ValueError(f'{path}: truncated')
        a = np.frombuffer(buf, dt, count, pos[0])
        pos[0] += n
        return a
    try:
        if cc() != b'IwFl':
            raise ValueError(f'{path}: not a FAISS IndexIVFFlat file')
        d, ntotal, _, _, trained, metric = take('iqqqBi')
        nlist, nprobe = take('QQ')
        if ntotal < 0 or ntotal * max(d, 0) * 4 > len(buf) or nlist * max(d, 0) * 4 > len(buf):
            raise ValueError(f'{path}: header promises more data than the file holds')
        q4 = cc()
        if q4 not in (b'IxFI', b'IxF2', b'IxFl') or metric != METRIC_INNER_PRODUCT:
            raise ValueError(f'{path}: only inner-product IVF-Flat over a flat quantiser is supported')
        qd, qn, _, _, _, qmetric = take('iqqqBi')
        nfl = take('Q')
        if qd != d or qn != nlist or nfl != nlist * d or not trained or d <= 0 or nlist <= 0:
            raise ValueError(f'{path}: inconsistent quantiser')
        cen = arr('<f4', nlist * d).reshape(nlist, d)
        dm_type, dm_n = take('B'), take('Q')
        arr('<i8', dm_n)
        if dm_type == 2:                                # hashtable: vector of (id, offset) pairs
            arr('<i8', 2 * take('Q'))
        if cc() != b'ilar':
            raise ValueError(f'{path}: inverted lists are not an ArrayInvertedLists')
        il_nlist, code_size = take('QQ')
        if il_nlist != nlist or code_size != 4 * d:
            raise ValueError(f'{path}: code size {code_size} is not {4 * d} (fp32 IVF-Flat)')
        lt = cc()
        sizes = np.zeros(nlist, np.int64)
        if lt == b'full':
            full = arr('<u8', take('Q'))
            if len(full) != nlist or (len(full) and full.max() > ntotal):
                raise ValueError(f'{path}: list size table out of range')
            sizes[:] = full.astype(np.int64)
        elif lt == b'sprs':
            npr = take('Q')
            if npr % 2:
                raise ValueError(f'{path}: odd sparse list table')
            pr = arr('<u8', npr).reshape(-1, 2)
            if len(pr) and (pr[:, 0].max() >= nlist or pr[:, 1].max() > ntotal):
                raise ValueError(f'{path}: sparse list table out of range')
            sizes[pr[:, 0].astype(np.int64)] = pr[:, 1].astype(np.int64)
        else:
            raise ValueError(f'{path}: unknown list layout {lt!r}')
        if int(sizes.sum()) != ntotal:
            raise ValueError(f'{path}: list sizes do not add up to ntotal')
        x = np.empty((ntotal, d), np.float32)
        lists = np.empty(ntotal, np.int32)
        seen = np.zeros(ntotal, bool)
        for l in range(nlist):
            n = int(sizes[l])
            if n == 0:
                continue
            codes = arr('<f4', n * d).reshape(n, d)
            ids = arr('<i8', n)
            if (ids.min() < 0 or ids.max() >= ntotal or seen[ids].any() or
                    len(np.unique(ids)) != n):
                raise ValueError(f'{path}: ids are not a permutation of 0..ntotal-1')
            seen[ids] = True
            x[ids] = codes
            lists[ids] = l
    except struct.error as e:
        raise ValueError(f'{path}: truncated ({e})') from None
    except (IndexError, OverflowError, MemoryError) as e:      # a damaged header: sizes, counts
        raise ValueError(f'{path}: corrupt ({type(e).__name__}: {e})') from None
    return {'d': int(d), 'nlist': int(nlist), 'nprobe': int(nprobe), 'ntotal': int(ntotal),
            'centroids': cen, 'x': x, 'lists': lists}


def read_index_faiss(path: str, storage: str = 'fp32') -> 'IndexIVFFlat':
    """Load a FAISS IVF-Flat / inner-product file (e.g. a ``.idxann`` the reference cached):
    FAISS' centroids and FAISS' own list assignments are kept; ids must be 0..ntotal-1 (what
    ``index.add`` gives and the reference relies on). Raises ValueError on anything else."""
    f = parse_index_faiss(path)
    d, nlist, nprobe, ntotal, cen, x, lists = (f['d'], f['nlist'], f['nprobe'], f['ntotal'],
                                               f['centroids'], f['x'], f['lists'])
    idx = IndexIVFFlat(IndexFlatIP(d), d, int(nlist), METRIC_INNER_PRODUCT, storage=storage)
    idx.set_trained(cen)
    if ntotal:
        _lib.check(_lib.lib().asl_index_add_preassigned(idx._h, int(ntotal), _lib.ptr(x), _lib.ptr(lists)))
    idx.nprobe = int(nprobe)
    return idx


def index_cpu_to_gpu(res, device, index, co=None):
    """Indexes of this library are GPU-resident already; kept for call-site parity
    (spectral_library.py:494)."""
    return index


def topk_merge_keys(Ks, unordered: bool = False):
    """Merge per-shard packed-key rows [S,nq,k] (``search_preassigned_keys``) -> (D [nq,k],
    I [nq,k]), sorted unless ``unordered`` (exact top-k sets, no final sort)."""
    S, nq, k = Ks.shape
    if isinstance(Ks, np.ndarray):
        Ks = np.ascontiguousarray(Ks, np.int64)
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
    else:
        import torch
        Ks = Ks.contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=Ks.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=Ks.device)
    _lib.check(_lib.lib().asl_topk_merge_keys(S, nq, k, _lib.ptr(Ks), _lib.ptr(D), _lib.ptr(I),
                                              int(bool(unordered))))
    return D, I


def topk_merge(Ds, Is):
    """Merge per-shard results [S,nq,k] -> [nq,k] (numpy or torch device tensors)."""
    S, nq, k = Ds.shape
    if isinstance(Ds, np.ndarray):
        Ds = np.ascontiguousarray(Ds, np.float32)
        Is = np.ascontiguousarray(Is, np.int64)
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
    else:
        import torch
        Ds, Is = Ds.contiguous(), Is.contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=Ds.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=Ds.device)
    _lib.check(_lib.lib().asl_topk_merge(S, nq, k, _lib.ptr(Ds), _lib.ptr(Is), _lib.ptr(D),
                                         _lib.ptr(I)))
    return D, I
