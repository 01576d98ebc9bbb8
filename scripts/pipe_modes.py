"""Measurement helper: step time of the whole hot path in the three pipeline modes (0 = one
stream, 1 = two streams, 3 = three: rescoring on its own stream) for IVF-Flat at the fixed-recall
point and for IVF-PQ.   python scripts/pipe_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic, _lib
from ann_solo_amd.spectral_library import Config, SpectralLibrary
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
for index, nprobe in (('ivfflat', 112), ('ivfpq', 128)):
    cfg = Config.open_search(num_list=4096, num_probe=nprobe, num_candidates=1024, index=index, pq_m=32, kmeans_niter=25, mode='ann',
                 precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da', batch_size=16384, seed=1234)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
    sl._get_ann_index(2)
    for mode in (0, 1, 3, 1, 3):
        sl.synchronize()
        _lib.check(_lib.lib().asl_set_pipeline(mode)); sl._pipeline_on = bool(mode)
        for _ in range(3): sl._search_batch(q, 2, 'open', device_out=True)
        sl.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12): sl._search_batch(q, 2, 'open', device_out=True)
        sl.synchronize(); torch.cuda.synchronize()
        print(index, 'pipeline mode', mode, '%.3f ms/step' % ((time.perf_counter() - t0) / 12 * 1e3), flush=True)
    _lib.check(_lib.lib().asl_set_pipeline(0)); sl._pipeline_on = False
    sl.shutdown()
