"""The options that reach the hot path.

``Config`` carries the reference's flags under the reference's names
(/root/reference/src/ann_solo/config.py:62-216) plus the ADDITIVE flags of this implementation
(``index``, ``pq_m``, ``pq_bits``, ``refine_k``, ``kmeans_niter``, ``ann_seed``, ``num_gpus``,
``flat_storage``). Defaults are the reference's: ``precursor_tolerance_mass_open`` /
``_mode_open`` are ``None`` (cascade off, config.py:151-156) and ``allow_peak_shifts`` is False
(a ``store_true`` flag, config.py:157), as a parsed reference configuration without those flags
has them. The one deviation: ``precursor_tolerance_mass`` / ``precursor_tolerance_mode`` /
``fragment_mz_tolerance`` are REQUIRED by the reference's parser (config.py:137-150) and default
here to 20 ppm / 0.02 Da, the values of the reference's notebooks. ``Config.open_search(**kw)``
is the hand-built configuration the tests and ``bench.py`` use: the notebooks' open search
(+-300 Da second cascade level, notebooks/iprg2012_ann_hyperparameters.ipynb:100-101) with peak
shifts on. ``tests/ref_config.py`` holds the reference's own defaults; ``Config.from_reference``
takes every value from the parsed reference object.

Capacity limits of the device kernels are checked at construction (``ValueError``), not deep
inside a search: ``max_peaks_used`` / ``max_peaks_used_library`` <= 256 (the reference has no
limit; its default is 50), ``num_probe`` <= 2048, ``num_candidates`` <= 16 384 (beyond 2 048 the
search runs ceil(k / 2048) bounded passes of the generic kernels: exact, slower).
``Config.from_reference(obj)`` snapshots any configuration object -- in particular the
reference's own ``ann_solo.config.config`` singleton, whose ``__getattr__`` answers unknown
options with ``KeyError`` (config.py:285-291), not ``AttributeError`` -- so the engine can be
constructed exactly as ``ann_solo.py:78-79`` does. ``add_arguments(parser)`` is the additive
patch for the reference's ``Config.__init__``."""
from dataclasses import dataclass, fields
from typing import Optional

_MISSING = object()


@dataclass
class Config:
    # --- the reference's flags (config.py:62-216), same names; defaults: see the module docstring
    resolution: Optional[int] = None
    min_mz: int = 11
    max_mz: int = 2010
    remove_precursor: bool = False
    remove_precursor_tolerance: float = 0
    min_intensity: float = 0.01
    min_peaks: int = 10
    min_mz_range: float = 250
    max_peaks_used: int = 50
    max_peaks_used_library: int = 50
    scaling: Optional[str] = 'rank'
    fdr: float = 0.01
    fdr_min_group_size: int = 100
    spectral_library_filename: str = ''
    query_filename: str = ''
    out_filename: str = ''
    bin_size: float = 0.04
    hash_len: int = 800
    num_candidates: int = 1024
    batch_size: int = 16384
    num_list: int = 256
    num_probe: int = 128
    mode: str = 'ann'                       # 'ann' | 'bf'
    precursor_tolerance_mass: float = 20.0
    precursor_tolerance_mode: str = 'ppm'   # 'Da' | 'ppm'
    precursor_tolerance_mass_open: Optional[float] = None
    precursor_tolerance_mode_open: Optional[str] = None
    fragment_mz_tolerance: float = 0.02
    allow_peak_shifts: bool = False
    no_gpu: bool = False
    # SSM scoring (config.py:158-164): carried for the caller's scorer. The engine itself has no
    # FDR model (out of scope): ``score_ssms=`` is injected; without one every SSM is
    # accepted with its cosine as the score and q = 0.
    model: Optional[str] = None
    # --- additive flags of this implementation
    index: str = 'ivfflat'                  # 'ivfflat' (the reference's index type) | 'ivfpq'
    pq_m: int = 32
    pq_bits: int = 8
    refine_k: Optional[int] = None          # IVF-PQ: exact re-rank of the refine_k best ADC candidates
    kmeans_niter: int = 25                  # FAISS' ClusteringParameters.niter default
    seed: int = 1234                        # flag --ann_seed; FAISS' ClusteringParameters.seed default
    num_gpus: int = 0                       # > 1: list-shard the ANN indexes over that many ranks
    flat_storage: str = 'fp32'              # IVF-Flat component storage: 'fp32' (as given: the reference's
                                            # CPU index) | 'fx22' (opt-in: 22-bit fixed point for
                                            # components in [0, 1), 4-byte postings, |dx| <= 1.2e-7)

    MAX_PEAKS = 256      # peaks per spectrum the preprocessing / rescoring kernels hold (csrc/process.hip)
    MAX_TOPK = 2048      # largest nprobe / single-pass k of the LDS top-k (csrc/ivf_kernels.hpp: TK_MAX_K)
    MAX_CANDIDATES = 16384   # largest num_candidates: beyond MAX_TOPK in bounded passes (TK_MAX_K_PASSES)

    def __post_init__(self):
        for name in ('max_peaks_used', 'max_peaks_used_library'):
            v = getattr(self, name)
            if v is not None and int(v) > self.MAX_PEAKS:
                raise ValueError(f'{name} = {v}: the device kernels hold at most {self.MAX_PEAKS} '
                                 'peaks per spectrum (the reference has no limit; its default is 50)')
        for name, lim in (('num_candidates', self.MAX_CANDIDATES), ('num_probe', self.MAX_TOPK)):
            v = getattr(self, name)
            if v is not None and int(v) > lim:
                raise ValueError(f'{name} = {v}: the device top-k holds at most {lim} entries')
        if self.flat_storage not in ('fx22', 'fp32'):
            raise ValueError(f"flat_storage = {self.flat_storage!r}: 'fx22' or 'fp32'")

    def __getitem__(self, k):
        return getattr(self, k)

    @property
    def ann_seed(self) -> int:              # the flag's name on the command line
        return self.seed

    @classmethod
    def open_search(cls, **kw) -> 'Config':
        """A hand-built configuration with the open-modification search switched on the way the
        reference's notebooks run it: second cascade level +-300 Da, shifted dot product."""
        base = dict(precursor_tolerance_mass_open=300.0, precursor_tolerance_mode_open='Da',
                    allow_peak_shifts=True)
        base.update(kw)
        return cls(**base)

    @classmethod
    def from_reference(cls, obj, **overrides) -> 'Config':
        """Snapshot of ``obj`` (a ``Config``, the reference's ``config`` singleton, an
        ``argparse.Namespace``, a mapping ...): every field it answers is taken, every field it
        does not know (``KeyError`` / ``AttributeError`` / ``RuntimeError`` of an unparsed
        reference config is NOT swallowed) keeps the default above."""
        if isinstance(obj, cls) and not overrides:
            return obj
        kw = {}
        for f in fields(cls):
            v = _lookup(obj, 'ann_seed' if f.name == 'seed' else f.name)
            if v is _MISSING and f.name == 'seed':
                v = _lookup(obj, 'seed')
            if v is not _MISSING:
                kw[f.name] = v
        kw.update(overrides)
        if kw.get('index') is None:
            kw.pop('index', None)
        return cls(**kw)


def _lookup(obj, name):
    if obj is None:
        return _MISSING
    if isinstance(obj, dict):
        return obj.get(name, _MISSING)
    try:
        return getattr(obj, name)
    except (AttributeError, KeyError):
        return _MISSING


def add_arguments(parser) -> None:
    """The additive command-line flags, for the reference's parser: one call at the end of
    ``ann_solo.config.Config.__init__`` (config.py:268, before ``self._namespace = None`` at :270):

        from ann_solo_amd.config import add_arguments; add_arguments(self._parser)

    No existing flag changes name, default or meaning."""
    d = Config()
    parser.add_argument('--index', default=d.index, type=str, choices=['ivfflat', 'ivfpq'],
                        help='ANN index type: inverted file with exact inner products (the '
                             "reference's index) or with product-quantised codes "
                             '(default: %(default)s)')
    parser.add_argument('--pq_m', default=d.pq_m, type=int,
                        help='IVF-PQ: number of sub-quantisers; must divide hash_len '
                             '(default: %(default)s)')
    parser.add_argument('--pq_bits', default=d.pq_bits, type=int,
                        help='IVF-PQ: bits per sub-quantiser code (default: %(default)s)')
    parser.add_argument('--refine_k', default=d.refine_k, type=int,
                        help='IVF-PQ: re-rank this many ADC candidates with exact inner products '
                             '(default: no re-ranking)')
    parser.add_argument('--kmeans_niter', default=d.kmeans_niter, type=int,
                        help='k-means iterations when training the ANN index '
                             '(default: %(default)s)')
    parser.add_argument('--ann_seed', default=d.seed, type=int,
                        help='random seed of the ANN index trainer (default: %(default)s)')
    parser.add_argument('--flat_storage', default=d.flat_storage, type=str, choices=['fx22', 'fp32'],
                        help='IVF-Flat: store vector components as float32 (what FAISS stores) or, '
                             'opt-in, those in [0, 1) as 22-bit fixed point (4-byte postings, '
                             '|dx| <= 1.2e-7) (default: %(default)s)')
    parser.add_argument('--num_gpus', default=d.num_gpus, type=int,
                        help='shard the ANN index by inverted list over this many GPUs; the job '
                             'runs one process per GPU (torchrun --nproc-per-node N) and N must '
                             'equal this value; 0 or 1: every process searches the whole index '
                             '(default: %(default)s)')
