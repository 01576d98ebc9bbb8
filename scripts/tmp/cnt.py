import os, sys, ctypes as C
sys.path.insert(0, '.')
import torch
from ann_solo_amd import synthetic, _lib
from ann_solo_amd.spectral_library import Config, SpectralLibrary
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', pq_m=32, kmeans_niter=5,
             precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da', batch_size=16384)
sl = SpectralLibrary(lib, config=cfg, device=dev)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
r = sl._search_batch(q, 2, 'open', device_out=True)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
_lib.lib().asl_dbg_counts(out)
print('scored', out[0], 'deferred', out[1], 'conflict', out[2], 'span', out[3], 'with matches', out[4], 'defer_bs', out[5])
