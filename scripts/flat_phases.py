"""Measurement helper: per-wave phase durations of flat_inv_scan_kernel (IVF-Flat postings scan).
Needs a library whose flat_scan.hip was compiled with -DFI_PHASES=1 (results are replaced by the
timers):   scripts/build_variant.sh scripts/tmp/phases.so "-DFI_PHASES=1" && \
           ASL_LIB_PATH=scripts/tmp/phases.so python scripts/flat_phases.py [nprobe]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=4096, num_probe=nprobe, num_candidates=1024, index='ivfflat',
             kmeans_niter=10, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
idx.nprobe = nprobe
idx.search(vec, 1024)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
D, _ = idx.search(vec, 1024)
ev1.record()
torch.cuda.synchronize()
print('search call: %.2f ms' % ev0.elapsed_time(ev1))
ph = torch.as_tensor(D)[:, :64].double().reshape(-1, 8, 8) / 100.0      # us, [query, wave, phase]
names = ['prologue', 'chunk table + end barrier', 'zero + table fetch', 'rows',
         'cold start + offers', 'syncs (asked for inside the offers or joined between blocks)', 'end-of-chunk wait', 'finish']
m = ph.mean((0, 1))
for n, v in zip(names, m):
    print(f'{n:32s} {v:8.1f} us per wave per query')
print(f'{"sum":32s} {m.sum():8.1f} us  (wave 0: {ph[:, 0].sum(1).mean():.1f}, wave 7: {ph[:, 7].sum(1).mean():.1f})')
