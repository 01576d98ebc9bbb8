"""Measurement helper: fingerprint of the IVF-Flat results at bench scale (2.1 M library, nlist 4096,
nprobe 128, k 1024, 16 384 queries) -- two builds of the library must print the same line:
  ASL_LIB_PATH=build_ab/lib_x.so python scripts/flat_hash.py"""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch, numpy as np
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index='ivfflat', kmeans_niter=10, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
idx.nprobe = 128
D, I = idx.search(vec, 1024)
D, I = torch.as_tensor(D).cpu().numpy(), torch.as_tensor(I).cpu().numpy()
print('hash', hashlib.sha1(I.tobytes()).hexdigest()[:16], hashlib.sha1(D.tobytes()).hexdigest()[:16], I[0, :4], D[0, :4])
