import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
lib, aux = synthetic.make_library(60000, seed=3, device='cuda', charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=256, num_probe=64, num_candidates=1024, index='ivfpq', kmeans_niter=4)
sl = SpectralLibrary(lib, config=cfg)
idx = sl._get_ann_index(2); idx.nprobe = 64
q, _ = synthetic.make_queries(lib, aux, 2000, seed=4, charge=2)
vec = sl._encode(q)
for k in (1024, 100, 1):
    D, I = idx.search(vec, k)
    idx.set_scan_variant(3)
    D8, I8 = idx.search(vec, k)
    idx.set_unordered(True)
    Du, Iu = idx.search(vec, k)
    idx.set_unordered(False); idx.set_scan_variant(0)
    assert torch.equal(I, I8) and torch.equal(D, D8), k
    assert torch.equal(I.sort(1)[0], Iu.sort(1)[0]), k
print('4-wave variant identical to the default')
