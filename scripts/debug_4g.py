import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
from ann_solo_amd import _lib, synthetic
from ann_solo_amd import faiss_compat as faiss
from ann_solo_amd.spectrum import spectra_to_vectors
dev = torch.device('cuda', 0)
def encode(sp):
    out = torch.empty((sp.n, 800), dtype=torch.float32, device=dev)
    spectra_to_vectors(sp.mz, sp.intensity, sp.offsets, 11, 2010, 0.04, 800, True, out)
    return out
lib0, aux0 = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
q, _ = synthetic.make_queries(lib0, aux0, 256, seed=42, open_range=500.0, charge=2)
xq = encode(q)
idx = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, 4096, 32, 8); idx.seed = 1234; idx.set_niter(10)
x0 = encode(lib0); idx.train(x0); idx.add(x0); del x0
idx.nprobe = 128
have = 1
for target in (62, 64):
    while have < target:
        lib_i, _ = synthetic.make_library(2_100_000, seed=7000 + have, device=dev, charges=(2,), charge_p=(1.0,))
        x = encode(lib_i); del lib_i; idx.add(x); del x; have += 1
    res = {}
    for variant in (0, 1):
        idx.set_scan_variant(variant)
        D = torch.full((256, 1024), -7.0, dtype=torch.float32, device=dev)
        I = torch.full((256, 1024), -7, dtype=torch.int64, device=dev)
        t = time.time()
        idx.search(xq, 1024, D, I)
        torch.cuda.synchronize()
        res[variant] = (D.cpu().numpy(), I.cpu().numpy())
        d = res[variant][0]
        print(target, 'variant', variant, 'sec', round(time.time() - t, 3), 'nan rows', int(np.isnan(d).any(1).sum()), 'nan', int(np.isnan(d).sum()),
              'untouched', int((d == -7.0).sum()), 'min', np.nanmin(d), 'max', np.nanmax(d), 'max id', res[variant][1].max(), flush=True)
    print(target, 'ids equal', np.array_equal(res[0][1], res[1][1]), 'scores equal', np.array_equal(res[0][0], res[1][0]), flush=True)
    d0 = res[0][0]
    bad = np.nonzero(np.isnan(d0).any(1))[0][:5]
    for b in bad:
        print('row', b, 'first nan at', int(np.argmax(np.isnan(d0[b]))), 'ids there', res[0][1][b][np.isnan(d0[b])][:8], flush=True)
