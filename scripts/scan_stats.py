"""Measurement helper: flush / append statistics of the PQ scan kernel (dbg bit 32)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib, aux = synthetic.make_library(n, device='cuda', charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=nlist, num_probe=128, num_candidates=1024, index='ivfpq', kmeans_niter=10)
sl = SpectralLibrary(lib, config=cfg)
q, _ = synthetic.make_queries(lib, aux, 2048, charge=2)
idx = sl._get_ann_index(2)
idx.nprobe = 128
vec = sl._encode(q.to('cuda'))
idx.set_scan_variant(32 << 8)
D, I = idx.search(vec, 1024)
torch.cuda.synchronize()
D = D.cpu()
print('flushes/query mean %.2f max %d | wave0 appends/query mean %.0f | tiles/query mean %.0f'
      % (D[:, 0].mean(), D[:, 0].max(), D[:, 1].mean(), D[:, 2].mean()))
print('cycles(100MHz ticks?)/query: total %.0f | mid flushes %.0f | finish %.0f | lut build %.0f'
      % (D[:, 3].mean(), D[:, 4].mean(), D[:, 5].mean(), D[:, 6].mean()))
