"""ann_solo_amd -- MI355X-native hot path of ANN-SoLo's open-modification search.

encode (feature hashing) -> IVF-Flat / IVF-PQ candidate retrieval -> precursor
post-filter -> (shifted) dot-product rescoring, behind the reference's own seams
(``spectrum_to_vector``, the FAISS-style ``index.search``, ``get_best_match``).
All compute runs in hand-written HIP kernels inside ``libannsolo_mi.so`` (C ABI in
include/annsolo_mi.h); there is no CPU fallback.
"""
__version__ = '0.1.0'
