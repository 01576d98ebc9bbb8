"""Stand-alone timing of the short-row selection (coarse top-nprobe of nlist): IndexFlatIP over
nlist tiny vectors, so that a search is a small GEMM + row_select_kernel. Prints ms per search for
a few (n, k); run under ASL_LIB_PATH=<other build> for an A/B on the same box."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from ann_solo_amd import faiss_compat as faiss

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
for n, k in ((4096, 128), (4096, 112), (4096, 256), (4096, 32), (1024, 128), (4096, 1)):
    d = 16
    idx = faiss.IndexFlatIP(d)
    idx.add(rng.standard_normal((n, d)).astype(np.float32))
    xq = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)).cuda()
    D = torch.empty((nq, k), dtype=torch.float32, device='cuda')
    I = torch.empty((nq, k), dtype=torch.int64, device='cuda')
    for _ in range(3):
        idx.search(xq, k, D, I)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx.search(xq, k, D, I)
    e1.record()
    torch.cuda.synchronize()
    print(f'n={n} k={k} nq={nq}: {e0.elapsed_time(e1) / 20:.4f} ms per search (GEMM + select)', flush=True)
