"""does the PLACEMENT of the packed peak records move the rescoring stage?
The rescoring kernels gather ~330 candidates x ~270 B per query from a 0.58 GB record array that
lies in spec_info (= row) order, i.e. scattered over the whole array. Modes (argv[1]):

  base        records in row order, queries in batch order (the product until round 4)
  store       records in inverted-list order of the partition's index (asl_library_set_record_order)
  query       row order, but the batch's queries sorted by their first probed list
  both        list-order records AND probe-sorted queries
  random      records in a random order (control: is row order already 'local'?)

(The modes that move records need ``asl_library_set_record_order``, which existed only for this
experiment -- commits 0b5a6c3..eb62a9e -- and went when the records became fixed-size row slots;
on the current tree only ``base`` and ``query`` run.) Prints the stage times (HIP events around the stages of a synchronous step, asl_profile) of
steps over ONE batch; run the same command under `rocprofv3 --pmc FETCH_SIZE` for the traffic of
rescore_flat_kernel.   python scripts/rescore_locality.py MODE [ivfflat|ivfpq] [nprobe] [steps]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import _lib, synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

mode = sys.argv[1] if len(sys.argv) > 1 else 'base'
index = sys.argv[2] if len(sys.argv) > 2 else 'ivfflat'
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 112
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=nprobe, num_candidates=1024, index=index, pq_m=32, kmeans_niter=25,
             mode='ann', precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da',
             batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
part = sl.partitions[2]
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
L = _lib.lib()
ref = sl._search_batch(q, 2, 'open', device_out=True)
if mode in ('store', 'both', 'random') and not hasattr(L, 'asl_library_set_record_order'):
    sys.exit('this build has no asl_library_set_record_order (see the module docstring)')
if mode in ('store', 'both'):
    _, ids, _ = idx.lists()                      # library rows in inverted-list order
    order = np.ascontiguousarray(ids, np.int32)
    _lib.check(L.asl_library_set_record_order(part.handle, _lib.ptr(order)))
elif mode == 'random':
    order = np.random.default_rng(1).permutation(lib.n).astype(np.int32)
    _lib.check(L.asl_library_set_record_order(part.handle, _lib.ptr(order)))
perm = None
if mode in ('query', 'both'):
    cD, cI = idx.coarse(sl._encode(q), nprobe)
    perm = torch.argsort(cI[:, 0].to(torch.int64) * 4096 + cI[:, 1].to(torch.int64), stable=True)
    q_run = q.select(perm).contiguous()
else:
    q_run = q
got = sl._search_batch(q_run, 2, 'open', device_out=True)
br, bs = got.best_row, got.best_score
if perm is not None:
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(len(perm), device=dev)
    br, bs = br[inv], bs[inv]
same = bool(torch.equal(br, ref.best_row) and torch.equal(bs, ref.best_score))
for pipelined in (False, True):
    sl.set_pipeline(pipelined)
    for _ in range(2):
        sl._search_batch(q_run, 2, 'open', device_out=True)
    sl.synchronize()
    L.asl_profile_reset()
    L.asl_profile_enable(1)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        sl._search_batch(q_run, 2, 'open', device_out=True)
    sl.synchronize()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    L.asl_profile_enable(0)
    out = {}
    for name in ('scan', 'filter', 'rescore', 'rescore_matches'):
        ms, n = C.c_double(), C.c_int64()
        L.asl_profile_get(name.encode(), C.byref(ms), C.byref(n))
        out[name] = round(ms.value / max(n.value, 1), 4)
    print(f'{mode:7s} {index} nprobe {nprobe} {"pipelined" if pipelined else "serial   "}: step {el:.3f} ms  ' +
          '  '.join(f'{k_} {v}' for k_, v in out.items()) + f'  identical results: {same}', flush=True)
sl.set_pipeline(False)
sl.shutdown()
