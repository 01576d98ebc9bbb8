"""Measurement helper: does the ORDER of the queries inside a scan launch matter (workgroups are
dispatched in query order; the last ones decide when the launch ends)? Times the list scan of one
batch in the given order, by descending scanned vectors (longest first), ascending, and shuffled.
  python scripts/lpt_probe.py [ivfpq|ivfflat] [batch]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

index = sys.argv[1] if len(sys.argv) > 1 else 'ivfpq'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index=index, pq_m=32, kmeans_niter=25,
                         mode='ann', precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da',
                         batch_size=batch, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, batch, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
cD, cI = idx.coarse(vec, 128)
off = torch.from_numpy(idx.lists()[0].astype(np.int64)).to(dev)
sizes = off[1:] - off[:-1]
work = sizes[cI.long()].sum(1)
print(f'{index} batch {batch}: scanned vectors per query min {int(work.min())} median {int(work.median())} max {int(work.max())}')
idx.set_unordered(True)


def scan_ms(order, reps=5):
    v, d_, i_ = vec[order].contiguous(), cD[order].contiguous(), cI[order].contiguous()
    idx.search_preassigned(v, 1024, d_, i_)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        idx.search_preassigned(v, 1024, d_, i_)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


g = torch.Generator(device='cpu').manual_seed(1)
orders = {'as generated': torch.arange(batch, device=dev),
          'longest first': torch.argsort(work, descending=True),
          'shortest first': torch.argsort(work),
          'shuffled': torch.randperm(batch, generator=g).to(dev)}
for rnd in range(2):
    for name, o in orders.items():
        print(f'  {name:15s} {scan_ms(o):7.3f} ms')
