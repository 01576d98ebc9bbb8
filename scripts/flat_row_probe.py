"""Measurement helper: per-block fixed cost against per-row cost of the IVF-Flat postings scan.
The same queries and probe lists, scanned with all / half / a quarter of each query's non-zero
components kept (rows per block fall in proportion, blocks per query do not).
  python scripts/flat_row_probe.py [nprobe]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import HipShardBackend

nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 112
n, batch = 2_100_000, 16384
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(n, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=nprobe, num_candidates=1024, index='ivfflat', kmeans_niter=25, mode='ann',
             precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da', batch_size=batch, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, batch, seed=42, open_range=500.0, charge=2)
be = HipShardBackend(sl, 2, 'open')
vec = be.encode(q)
cD, cI = be.coarse(vec)
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(vec.shape, device=dev, generator=g)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for keep in (1.0, 0.5, 0.25):
    x = torch.where(u < keep, vec, torch.zeros_like(vec)).contiguous()
    nnz = float((x != 0).sum(1).float().mean())
    for npr in (nprobe, nprobe // 2):
        d_, i_ = cD[:, :npr].contiguous(), cI[:, :npr].contiguous()
        t = timed(lambda: idx.search_preassigned_keys(x, 1024, d_, i_))
        print(f'non-zeros per query {nnz:5.1f}, {npr:3d} lists: {t:.3f} ms', flush=True)
