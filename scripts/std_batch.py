"""Measurement helper: one standard-search batch (precursor window only, cascade level 1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', kmeans_niter=2, batch_size=16384)
sl = SpectralLibrary(lib, config=cfg, device=dev)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
for mode in ('std', 'std', 'std'):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sl._search_batch(q, 2, mode, device_out=True)
    torch.cuda.synchronize(); print(mode, 'batch ms', round((time.perf_counter() - t0) * 1e3, 2), 'identified', int((r.best_row >= 0).sum()), 'mean candidates', float(r.n_candidates.float().mean()))
