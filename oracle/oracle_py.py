"""ctypes binding of the CPU oracle (oracle/liboracle.so, oracle/_ref/*.so).

TEST INFRASTRUCTURE. Import this only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. Nothing under ann_solo_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)


class OrcPeaks(C.Structure):
    _fields_ = [('n', C.c_int32), ('offsets', c_i32p), ('mz', c_f32p),
                ('intensity', c_f32p), ('charge', c_u8p), ('precursor_mz', c_f64p),
                ('precursor_charge', c_i32p)]


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'asl_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, os.path.join(_HERE, 'liboracle.so')])
    ref = os.path.join(_HERE, '_ref', 'libref_spectrummatch.so')
    if os.path.exists('/root/reference/src/ann_solo/SpectrumMatch.cpp') and (
            force or not os.path.exists(ref)):
        subprocess.check_call(['make', '-C', _HERE, 'ref'])


def _sanitized():
    """ASL_ORACLE_SANITIZE=1: the ASan + UBSan builds of `make -C oracle sanitize` (oracle/_san/) stand in
    for liboracle.so and _ref/ (scripts/sanitize_cpu.sh; the interpreter must run with libasan preloaded)."""
    return os.environ.get('ASL_ORACLE_SANITIZE') == '1'


def lib():
    global _LIB
    if _LIB is None:
        if _sanitized():
            _LIB = C.CDLL(os.path.join(_HERE, '_san', 'liboracle.so'))
        else:
            build()
            _LIB = C.CDLL(os.path.join(_HERE, 'liboracle.so'))
        L = _LIB
        L.orc_murmur3_32.restype = C.c_uint32
        L.orc_murmur3_32.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        L.orc_hash_idx.restype = C.c_int32
        L.orc_hash_idx.argtypes = [C.c_int64, C.c_int32, C.c_uint32]
        L.orc_npy_floor_divide.restype = C.c_double
        L.orc_npy_floor_divide.argtypes = [C.c_double, C.c_double]
        L.orc_bin_idx.restype = C.c_int64
        L.orc_bin_idx.argtypes = [C.c_float, C.c_double, C.c_double]
        L.orc_dot_pair.restype = C.c_double
        L.orc_ip.restype = C.c_float
        L.orc_adc.restype = C.c_float
        L.orc_best_match.restype = C.c_int32
        L.orc_precursor_ok.restype = C.c_int
        L.orc_precursor_ok.argtypes = [C.c_double, C.c_float, C.c_int32, C.c_double, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_csr_count.restype = C.c_int64
        L.orc_fx22.restype = C.c_float
        L.orc_fx22.argtypes = [C.c_float]
    return _LIB


def ref_lib():
    """The reference's own SpectrumMatch.cpp (None if it was never built)."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, '_san' if _sanitized() else '_ref', 'libref_spectrummatch.so')
        if not os.path.exists(p) and not _sanitized():
            build()
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        _REF.ref_best_match.restype = C.c_int32
    return _REF


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Spectra:
    """Host-side packed spectra (numpy), the oracle's input format."""

    def __init__(self, offsets, mz, intensity, charge, precursor_mz, precursor_charge):
        self.offsets = _c(offsets, np.int32)
        self.mz = _c(mz, np.float32)
        self.intensity = _c(intensity, np.float32)
        self.charge = (_c(charge, np.uint8) if charge is not None
                       else np.zeros(len(self.mz), np.uint8))
        self.precursor_mz = _c(precursor_mz, np.float64)
        self.precursor_charge = _c(precursor_charge, np.int32)
        self.n = len(self.offsets) - 1

    def struct(self):
        return OrcPeaks(self.n, _p(self.offsets, c_i32p), _p(self.mz, c_f32p),
                        _p(self.intensity, c_f32p), _p(self.charge, c_u8p),
                        _p(self.precursor_mz, c_f64p), _p(self.precursor_charge, c_i32p))

    def peaks(self, i):
        o, e = self.offsets[i], self.offsets[i + 1]
        return self.mz[o:e], self.intensity[o:e], self.charge[o:e]


# ---------------------------------------------------------------- encoder
def murmur3_32(data: bytes, seed: int) -> int:
    return lib().orc_murmur3_32(data, len(data), seed & 0xffffffff)


def hash_idx(bin_idx: int, hash_len: int, seed: int = 42) -> int:
    return lib().orc_hash_idx(int(bin_idx), int(hash_len), seed)


def get_dim(min_mz, max_mz, bin_size):
    n = C.c_int64()
    s = C.c_double()
    e = C.c_double()
    lib().orc_get_dim(C.c_double(min_mz), C.c_double(max_mz), C.c_double(bin_size),
                      C.byref(n), C.byref(s), C.byref(e))
    return n.value, s.value, e.value


def bin_idx(mz, min_bound, bin_size):
    return lib().orc_bin_idx(np.float32(mz), min_bound, bin_size)


def encode_batch(mz, intensity, offsets, min_bound, bin_size, hash_len, seed=42, norm=True):
    mz = _c(mz, np.float32)
    intensity = _c(intensity, np.float32)
    offsets = _c(offsets, np.int32)
    n = len(offsets) - 1
    out = np.empty((n, hash_len), np.float32)
    lib().orc_encode_batch(_p(mz, c_f32p), _p(intensity, c_f32p), _p(offsets, c_i32p),
                           C.c_int32(n), C.c_double(min_bound), C.c_double(bin_size),
                           C.c_int32(hash_len), C.c_uint32(seed), C.c_int(bool(norm)),
                           _p(out, c_f32p))
    return out


class OrcProcessParams(C.Structure):
    _fields_ = [('min_mz', C.c_double), ('max_mz', C.c_double), ('remove_precursor', C.c_int32),
                ('remove_precursor_tolerance', C.c_double), ('min_intensity', C.c_double),
                ('max_peaks', C.c_int32), ('scaling', C.c_int32), ('min_peaks', C.c_int32),
                ('min_mz_range', C.c_double), ('round_mz', C.c_int32), ('resolution', C.c_int32)]


def process_spectrum(mz, intensity, precursor_mz, precursor_charge, min_mz=11, max_mz=2010,
                     remove_precursor=False, remove_precursor_tolerance=0.0, min_intensity=0.01,
                     max_peaks=50, scaling='rank', min_peaks=10, min_mz_range=250.0,
                     resolution=None):
    mz, intensity = _c(mz, np.float32), _c(intensity, np.float32)
    P = OrcProcessParams(min_mz, max_mz, int(remove_precursor), remove_precursor_tolerance,
                         min_intensity, max_peaks, {'rank': 1, 'root': 2, 'sqrt': 2, None: 0}[scaling],
                         min_peaks, min_mz_range, int(resolution is not None),
                         int(resolution or 0))
    om, oi = np.zeros(max_peaks, np.float32), np.zeros(max_peaks, np.float32)
    src = np.zeros(max_peaks, np.int32)
    n = C.c_int32()
    L = lib()
    L.orc_process_spectrum.restype = C.c_int
    ok = L.orc_process_spectrum(_p(mz, c_f32p), _p(intensity, c_f32p), C.c_int32(len(mz)),
                                C.c_double(precursor_mz), C.c_int32(precursor_charge), C.byref(P),
                                _p(om, c_f32p), _p(oi, c_f32p), _p(src, c_i32p), C.byref(n))
    return bool(ok), om[:n.value].copy(), oi[:n.value].copy(), src[:n.value].copy()


# ---------------------------------------------------------------- similarity features
SIM_NFEAT = 33


def ssm_features(q_mz, q_int, l_mz, l_int, peak_matches, min_mz=11, max_mz=2010, bin_size=0.04,
                 top=5):
    q_mz, q_int = _c(q_mz, np.float32), _c(q_int, np.float32)
    l_mz, l_int = _c(l_mz, np.float32), _c(l_int, np.float32)
    pm = _c(np.asarray(peak_matches).reshape(-1, 2), np.uint32)
    out = np.zeros(SIM_NFEAT, np.float64)
    L = lib()
    L.orc_ssm_features.restype = None
    L.orc_ssm_features(_p(q_mz, c_f32p), _p(q_int, c_f32p), C.c_int32(len(q_mz)),
                       _p(l_mz, c_f32p), _p(l_int, c_f32p), C.c_int32(len(l_mz)),
                       _p(pm, c_u32p), C.c_int32(len(pm)), C.c_double(min_mz), C.c_double(max_mz),
                       C.c_double(bin_size), C.c_int32(top), _p(out, c_f64p))
    return out


# ---------------------------------------------------------------- rescoring
def dot_pair(q_mz, q_int, q_pmz, c_mz, c_int, c_chg, c_pmz, c_charge, tol, allow_shift):
    q_mz, q_int = _c(q_mz, np.float32), _c(q_int, np.float32)
    c_mz, c_int, c_chg = _c(c_mz, np.float32), _c(c_int, np.float32), _c(c_chg, np.uint8)
    cap = max(1, min(len(q_mz), len(c_mz)))
    m = np.zeros((cap, 2), np.uint32)
    nm = C.c_int32()
    s = lib().orc_dot_pair(_p(q_mz, c_f32p), _p(q_int, c_f32p), C.c_int32(len(q_mz)),
                           C.c_double(q_pmz), _p(c_mz, c_f32p), _p(c_int, c_f32p),
                           _p(c_chg, c_u8p), C.c_int32(len(c_mz)), C.c_double(c_pmz),
                           C.c_int32(c_charge), C.c_double(tol), C.c_int(bool(allow_shift)),
                           _p(m, c_u32p), C.byref(nm))
    return s, m[:nm.value].copy()


def best_match(queries: Spectra, qi, library: Spectra, cand_rows, tol, allow_shift):
    cand = _c(cand_rows, np.int64)
    qn = queries.offsets[qi + 1] - queries.offsets[qi]
    m = np.zeros((max(1, qn), 2), np.uint32)
    nm = C.c_int32()
    sc = C.c_double()
    qs, ls = queries.struct(), library.struct()
    b = lib().orc_best_match(C.byref(qs), C.c_int32(qi), C.byref(ls), _p(cand, c_i64p),
                             C.c_int32(len(cand)), C.c_double(tol),
                             C.c_int(bool(allow_shift)), C.byref(sc), _p(m, c_u32p),
                             C.byref(nm))
    return b, sc.value, m[:nm.value].copy()


def ref_best_match(queries: Spectra, qi, library: Spectra, cand_rows, tol, allow_shift):
    """Same call answered by the reference's own compiled SpectrumMatch.cpp."""
    R = ref_lib()
    if R is None:
        raise RuntimeError('oracle/_ref/libref_spectrummatch.so not built')
    cand = _c(cand_rows, np.int64)
    o, e = queries.offsets[qi], queries.offsets[qi + 1]
    q_mz, q_int = queries.mz[o:e].copy(), queries.intensity[o:e].copy()
    m = np.zeros((max(1, e - o), 2), np.uint32)
    nm = C.c_int32()
    sc = C.c_double()
    b = R.ref_best_match(C.c_double(queries.precursor_mz[qi]),
                         C.c_int32(queries.precursor_charge[qi]), C.c_int32(e - o),
                         _p(q_mz, c_f32p), _p(q_int, c_f32p), _p(library.offsets, c_i32p),
                         _p(library.mz, c_f32p), _p(library.intensity, c_f32p),
                         _p(library.charge, c_u8p), _p(library.precursor_mz, c_f64p),
                         _p(library.precursor_charge, c_i32p), _p(cand, c_i64p),
                         C.c_int32(len(cand)), C.c_double(tol), C.c_int(bool(allow_shift)),
                         C.byref(sc), _p(m, c_u32p), C.byref(nm))
    return b, sc.value, m[:nm.value].copy()


# ---------------------------------------------------------------- ANN
def ip(a, b):
    a, b = _c(a, np.float32), _c(b, np.float32)
    return lib().orc_ip(_p(a, c_f32p), _p(b, c_f32p), C.c_int32(len(a)))


def flat_search(xb, xq, k):
    xb, xq = _c(xb, np.float32), _c(xq, np.float32)
    nq, d = xq.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().orc_flat_search(_p(xb, c_f32p), C.c_int64(len(xb)), _p(xq, c_f32p), C.c_int32(nq),
                          C.c_int32(d), C.c_int32(k), _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def rand_perm(n, seed):
    p = np.empty(n, np.int64)
    lib().orc_rand_perm(C.c_int64(n), C.c_uint64(seed), _p(p, c_i64p))
    return p


def kmeans(x, k, niter=25, seed=1234, metric=0, max_ppc=256):
    x = _c(x, np.float32)
    n, d = x.shape
    cen = np.empty((k, d), np.float32)
    lib().orc_kmeans(_p(x, c_f32p), C.c_int64(n), C.c_int32(d), C.c_int32(k),
                     C.c_int32(niter), C.c_uint64(seed), C.c_int(metric),
                     C.c_int32(max_ppc), _p(cen, c_f32p))
    return cen


def assign(x, centroids, metric=0):
    x, centroids = _c(x, np.float32), _c(centroids, np.float32)
    a = np.empty(len(x), np.int32)
    lib().orc_assign(_p(x, c_f32p), C.c_int64(len(x)), C.c_int32(x.shape[1]),
                     _p(centroids, c_f32p), C.c_int32(len(centroids)), C.c_int(metric),
                     _p(a, c_i32p))
    return a


def pq_train(x, centroids, m, ksub=256, niter=25, seed=1234):
    x, centroids = _c(x, np.float32), _c(centroids, np.float32)
    n, d = x.shape
    cb = np.empty((m, ksub, d // m), np.float32)
    lib().orc_pq_train(_p(x, c_f32p), C.c_int64(n), C.c_int32(d), _p(centroids, c_f32p),
                       C.c_int32(len(centroids)), C.c_int32(m), C.c_int32(ksub),
                       C.c_int32(niter), C.c_uint64(seed), _p(cb, c_f32p))
    return cb


def pq_encode(x, centroids, assign_, codebooks):
    x, centroids = _c(x, np.float32), _c(centroids, np.float32)
    assign_, codebooks = _c(assign_, np.int32), _c(codebooks, np.float32)
    m, ksub, _ = codebooks.shape
    codes = np.empty((len(x), m), np.uint8)
    lib().orc_pq_encode(_p(x, c_f32p), C.c_int64(len(x)), C.c_int32(x.shape[1]),
                        _p(centroids, c_f32p), _p(assign_, c_i32p), _p(codebooks, c_f32p),
                        C.c_int32(m), C.c_int32(ksub), _p(codes, c_u8p))
    return codes


def pq_lut(xq, codebooks):
    xq, codebooks = _c(xq, np.float32), _c(codebooks, np.float32)
    m, ksub, _ = codebooks.shape
    lut = np.empty((m, ksub), np.float32)
    lib().orc_pq_lut(_p(xq, c_f32p), C.c_int32(len(xq)), _p(codebooks, c_f32p), C.c_int32(m),
                     C.c_int32(ksub), _p(lut, c_f32p))
    return lut


def adc(lut, code, coarse):
    lut, code = _c(lut, np.float32), _c(code, np.uint8)
    return lib().orc_adc(_p(lut, c_f32p), C.c_int32(lut.shape[0]), C.c_int32(lut.shape[1]),
                         _p(code, c_u8p), C.c_float(coarse))


def coarse(xq, centroids, nprobe):
    xq, centroids = _c(xq, np.float32), _c(centroids, np.float32)
    nq, d = xq.shape
    D = np.empty((nq, nprobe), np.float32)
    I = np.empty((nq, nprobe), np.int32)
    lib().orc_coarse(_p(xq, c_f32p), C.c_int32(nq), C.c_int32(d), _p(centroids, c_f32p),
                     C.c_int32(len(centroids)), C.c_int32(nprobe), _p(D, c_f32p),
                     _p(I, c_i32p))
    return D, I


class OrcCsr(C.Structure):
    _fields_ = [('indptr', C.c_void_p), ('dims', C.c_void_p), ('vals', C.c_void_p)]


def quantize_fx22(x):
    """The fixed-point storage rule of IVF-Flat (``asl_index_set_flat_storage``, FX22): a copy of
    ``x`` with every component in [0, 1) on the grid of 2^-22."""
    out = np.array(x, np.float32, copy=True, order='C')
    lib().orc_quantize_fx22(_p(out, c_f32p), C.c_int64(out.size))
    return out


class HostIVF:
    """Inverted lists laid out for the oracle (list order, ascending id in list)."""

    def __init__(self, centroids, assign_, payload, codebooks=None):
        self.centroids = _c(centroids, np.float32)
        self.nlist, self.d = self.centroids.shape
        assign_ = np.asarray(assign_)
        order = np.argsort(assign_, kind='stable').astype(np.int32)
        self.ids = order
        counts = np.bincount(assign_, minlength=self.nlist)
        self.list_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.payload = np.ascontiguousarray(payload[order])
        self.codebooks = None if codebooks is None else _c(codebooks, np.float32)
        self.kind = 0 if codebooks is None else 1

    def to_csr(self):
        """Sparse copy of an IVF-Flat payload (CSR in list order): ``kind`` becomes 2 and
        ``search`` / ``search_batch`` run the sparse-aware scan -- same chain, same bits."""
        assert self.kind == 0
        x = _c(self.payload, np.float32)
        n, d = x.shape
        nnz = int(lib().orc_csr_count(_p(x, c_f32p), C.c_int64(n), C.c_int32(d)))
        out = HostIVF.__new__(HostIVF)
        out.__dict__.update(self.__dict__)
        out.indptr = np.empty(n + 1, np.int64)
        out.dims = np.empty(max(nnz, 1), np.uint16)
        out.vals = np.empty(max(nnz, 1), np.float32)
        lib().orc_csr_fill(_p(x, c_f32p), C.c_int64(n), C.c_int32(d), _p(out.indptr, c_i64p),
                           out.dims.ctypes.data_as(C.c_void_p), _p(out.vals, c_f32p))
        out.payload = None
        out.kind = 2
        out._csr = OrcCsr(out.indptr.ctypes.data, out.dims.ctypes.data, out.vals.ctypes.data)
        return out

    def search(self, xq, k, nprobe):
        xq = _c(xq, np.float32)
        nq = len(xq)
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        if self.kind == 2:
            lib().orc_ivfflat_csr_search(
                _p(xq, c_f32p), C.c_int32(nq), C.c_int32(self.d), _p(self.centroids, c_f32p),
                C.c_int32(self.nlist), _p(self.list_offsets, c_i32p), _p(self.ids, c_i32p),
                _p(self.indptr, c_i64p), self.dims.ctypes.data_as(C.c_void_p),
                _p(self.vals, c_f32p), C.c_int32(k), C.c_int32(nprobe), _p(D, c_f32p),
                _p(I, c_i64p))
        elif self.kind == 0:
            lib().orc_ivfflat_search(
                _p(xq, c_f32p), C.c_int32(nq), C.c_int32(self.d), _p(self.centroids, c_f32p),
                C.c_int32(self.nlist), _p(self.list_offsets, c_i32p), _p(self.ids, c_i32p),
                _p(self.payload, c_f32p), C.c_int32(k), C.c_int32(nprobe), _p(D, c_f32p),
                _p(I, c_i64p))
        else:
            m, ksub, _ = self.codebooks.shape
            lib().orc_ivfpq_search(
                _p(xq, c_f32p), C.c_int32(nq), C.c_int32(self.d), _p(self.centroids, c_f32p),
                C.c_int32(self.nlist), _p(self.list_offsets, c_i32p), _p(self.ids, c_i32p),
                _p(self.payload, c_u8p), _p(self.codebooks, c_f32p), C.c_int32(m),
                C.c_int32(ksub), C.c_int32(k), C.c_int32(nprobe), _p(D, c_f32p),
                _p(I, c_i64p))
        return D, I


def refine(xb, xq, I_in, k):
    """Exact re-rank of a short-list (FAISS IndexRefineFlat): (D [nq,k], I [nq,k])."""
    xb, xq, I_in = _c(xb, np.float32), _c(xq, np.float32), _c(I_in, np.int64)
    nq, kp = I_in.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().orc_refine(_p(xb, c_f32p), _p(xq, c_f32p), C.c_int32(nq), C.c_int32(xb.shape[1]),
                     _p(I_in, c_i64p), C.c_int32(kp), C.c_int32(k), _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def topk_merge(Ds, Is):
    Ds, Is = _c(Ds, np.float32), _c(Is, np.int64)
    S, nq, k = Ds.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().orc_topk_merge(_p(Ds, c_f32p), _p(Is, c_i64p), C.c_int32(S), C.c_int32(nq),
                         C.c_int32(k), _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def precursor_ok(q_mz, lib_mz, charge, tol, mode):
    return bool(lib().orc_precursor_ok(float(q_mz), np.float32(lib_mz), int(charge),
                                       float(tol), 0 if mode == 'Da' else 1))


def search_batch(queries: Spectra, library: Spectra, lib_pmz_f32, charge, ivf: HostIVF, k,
                 nprobe, prec_tol, prec_mode, frag_tol, allow_shift, min_bound=10.96,
                 bin_size=0.04, seed=42, pm_stride=64, nthreads=0, want_knn=False):
    nq = queries.n
    lib_pmz_f32 = _c(lib_pmz_f32, np.float32)
    best_row = np.empty(nq, np.int32)
    best_score = np.empty(nq, np.float64)
    n_cand = np.empty(nq, np.int32)
    pm_count = np.empty(nq, np.int32)
    pm_pairs = np.zeros((nq, pm_stride, 2), np.uint32)
    knn = np.empty((nq, k), np.int64) if want_knn else None
    if ivf.kind == 1:
        m, ksub, _ = ivf.codebooks.shape
    else:
        m, ksub = 0, 0
    qs, ls = queries.struct(), library.struct()
    lib().orc_search_batch(
        C.byref(qs), C.byref(ls), _p(lib_pmz_f32, c_f32p), C.c_int32(charge),
        C.c_double(min_bound), C.c_double(bin_size), C.c_int32(ivf.d), C.c_uint32(seed),
        C.c_int(ivf.kind), _p(ivf.centroids, c_f32p), C.c_int32(ivf.nlist),
        _p(ivf.list_offsets, c_i32p), _p(ivf.ids, c_i32p),
C.cast(C.pointer(ivf._csr), C.c_void_p) if ivf.kind == 2 else ivf.payload.ctypes.data_as(C.c_void_p),
        _p(ivf.codebooks, c_f32p) if ivf.codebooks is not None else None, C.c_int32(m),
        C.c_int32(ksub), C.c_int32(k), C.c_int32(nprobe), C.c_double(prec_tol),
        C.c_int(0 if prec_mode == 'Da' else 1), C.c_double(frag_tol),
        C.c_int(bool(allow_shift)), _p(best_row, c_i32p), _p(best_score, c_f64p),
        _p(n_cand, c_i32p), _p(pm_count, c_i32p), _p(pm_pairs, c_u32p), C.c_int32(pm_stride),
        _p(knn, c_i64p) if knn is not None else None, C.c_int32(nthreads))
    return dict(best_row=best_row, best_score=best_score, n_cand=n_cand, pm_count=pm_count,
                pm_pairs=pm_pairs, knn_I=knn)


def max_threads():
    return lib().orc_max_threads()
