"""Byte-level test vector of FAISS' on-disk IndexIVFFlat format, written field by field in the
order FAISS' own writer uses -- WITHOUT importing anything of this repository (numpy + struct
only), so that the reader / writer in ann_solo_amd/faiss_compat.py are checked against bytes
they did not produce.

FAISS is not installed in the build container and the reference pins no version
(/root/reference/src/setup.py:99), so no file written by FAISS itself exists here; this script
restates the published writer, faiss/impl/index_write.cpp (identical for these classes from
v1.5.0 to v1.8.0), function by function:

  write_index(const Index*)            IndexIVFFlat branch: fourcc "IwFl", write_ivf_header,
                                       write_InvertedLists
  write_index_header                   d (int), ntotal (idx_t = int64), two dummies 1 << 20
                                       (idx_t), is_trained (bool, 1 byte), metric_type (int:
                                       METRIC_INNER_PRODUCT = 0, METRIC_L2 = 1; metric_arg only
                                       for metric_type > 1)
  write_ivf_header                     write_index_header, nlist (size_t), nprobe (size_t),
                                       write_index(quantizer), write_direct_map
  write_index, IndexFlat branch        fourcc "IxFI" (IndexFlatIP; "IxF2" = IndexFlatL2, "IxFl"
                                       = IndexFlat), write_index_header, WRITEVECTOR(xb): count
                                       of floats (size_t) + the floats (>= 1.7: WRITEXBVECTOR
                                       of the byte codes: count of 4-byte units -- the same bytes)
  write_direct_map                     type (char: 0 = NoMap), WRITEVECTOR(array): size_t 0
  write_InvertedLists                  ArrayInvertedLists: fourcc "ilar", nlist (size_t),
                                       code_size (size_t = 4 d), then the list sizes: more than
                                       nlist / 2 non-empty lists -> fourcc "full" + WRITEVECTOR
                                       of nlist sizes (size_t); else fourcc "sprs" + WRITEVECTOR
                                       of (list, size) pairs; then per NON-EMPTY list its codes
                                       (n * code_size bytes) followed by its ids (n * idx_t)
  (faiss/impl/io_macros.h: WRITE1 = the raw object, WRITEVECTOR = size_t count + raw data; all
  little endian on x86-64.)

Run from the repository root:  python tests/golden/make_faiss_fixture.py
Writes tests/golden/faiss_ivfflat_kat.idxann (full size table), faiss_ivfflat_kat_sprs.idxann
(sparse size table) and faiss_ivfflat_kat.npz (the content both must parse to)."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FOURCC = lambda s: s.encode('ascii')            # faiss::fourcc packs the 4 characters little endian


def index_header(d, ntotal, metric):
    return (struct.pack('<i', d) + struct.pack('<q', ntotal) + struct.pack('<q', 1 << 20) +
            struct.pack('<q', 1 << 20) + struct.pack('<?', True) + struct.pack('<i', metric))


def ivfflat_bytes(d, centroids, list_ids, list_vecs, nprobe):
    nlist = len(centroids)
    ntotal = sum(len(i) for i in list_ids)
    out = FOURCC('IwFl')
    out += index_header(d, ntotal, 0)                              # write_ivf_header ...
    out += struct.pack('<Q', nlist) + struct.pack('<Q', nprobe)
    out += FOURCC('IxFI') + index_header(d, nlist, 0)              # ... the quantiser (IndexFlatIP)
    out += struct.pack('<Q', nlist * d) + np.asarray(centroids, '<f4').tobytes()
    out += struct.pack('<b', 0) + struct.pack('<Q', 0)             # write_direct_map: NoMap, empty array
    out += FOURCC('ilar') + struct.pack('<Q', nlist) + struct.pack('<Q', 4 * d)
    sizes = [len(i) for i in list_ids]
    n_non0 = sum(1 for s in sizes if s > 0)
    if n_non0 > nlist // 2:
        out += FOURCC('full') + struct.pack('<Q', nlist) + b''.join(struct.pack('<Q', s) for s in sizes)
    else:
        pairs = [v for l, s in enumerate(sizes) if s > 0 for v in (l, s)]
        out += FOURCC('sprs') + struct.pack('<Q', len(pairs)) + b''.join(struct.pack('<Q', v) for v in pairs)
    for ids, vecs in zip(list_ids, list_vecs):
        if len(ids):
            out += np.asarray(vecs, '<f4').tobytes() + np.asarray(ids, '<i8').tobytes()
    return out


def main():
    rng = np.random.default_rng(20241003)
    d, nlist, n = 24, 10, 57

    def rows(m, nnz):
        x = np.zeros((m, d), np.float32)
        for r in range(m):
            c = rng.choice(d, nnz, replace=False)
            # multiples of 2^-12: exactly representable, and on the 2^-22 grid of the fixed-point
            # IVF-Flat storage, so the vectors survive either storage mode bit for bit
            x[r, c] = rng.integers(1, 2048, nnz).astype(np.float32) / np.float32(4096)
        return x
    centroids = rows(nlist, 6)
    x = rows(n, 7)
    assign = rng.integers(0, nlist, n)
    assign[assign == 3] = 7                                        # an empty list in the middle
    order = rng.permutation(n)                                     # ids inside a list in FAISS' add order need
    list_ids = [np.sort(np.nonzero(assign == l)[0]) for l in range(nlist)]   # not be sorted; here they are
    list_vecs = [x[i] for i in list_ids]
    full = ivfflat_bytes(d, centroids, list_ids, list_vecs, nprobe=4)
    # the same content with most lists empty (FAISS then writes the sparse table)
    few = [l for l in range(nlist) if len(list_ids[l])][:3]
    keep = np.concatenate([list_ids[l] for l in few])
    renum = {int(old): new for new, old in enumerate(np.sort(keep))}
    s_ids = [np.asarray([renum[int(i)] for i in list_ids[l]], np.int64) if l in few else np.zeros(0, np.int64)
             for l in range(nlist)]
    s_vecs = [x[list_ids[l]] if l in few else np.zeros((0, d), np.float32) for l in range(nlist)]
    sprs = ivfflat_bytes(d, centroids, s_ids, s_vecs, nprobe=2)
    open(os.path.join(HERE, 'faiss_ivfflat_kat.idxann'), 'wb').write(full)
    open(os.path.join(HERE, 'faiss_ivfflat_kat_sprs.idxann'), 'wb').write(sprs)
    s_x = x[np.sort(keep)]
    s_assign = np.empty(len(keep), np.int32)
    for l in few:
        s_assign[s_ids[l]] = l
    np.savez(os.path.join(HERE, 'faiss_ivfflat_kat.npz'), d=d, nlist=nlist, nprobe=4, centroids=centroids,
             x=x, lists=assign.astype(np.int32), sprs_x=s_x, sprs_lists=s_assign, sprs_nprobe=2)
    print(len(full), len(sprs), 'bytes')


if __name__ == '__main__':
    main()
