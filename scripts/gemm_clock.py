#!/usr/bin/env python3
"""Measurement helper: the coarse-quantiser GEMM in a tight loop while the engine clock is sampled
(rocm-smi), to separate MFMA-pipe efficiency from clock throttling: fp32-MFMA peak = 256 CUs x
256 flop/clk x sclk.   python scripts/gemm_clock.py [seconds]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import faiss_compat as faiss

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device('cuda', 0)
nq, nlist, d = 16384, 4096, 800
torch.manual_seed(0)
idx = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist)
idx.set_trained(torch.randn(nlist, d).numpy())
xq = torch.randn(nq, d, device=dev)
idx.coarse(xq, 128)
torch.cuda.synchronize()
samples = []
stop = False


def sample():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r'sclk clock level:.*?\((\d+)Mhz\)', out)
            if m:
                samples.append(int(m.group(1)))
        except Exception:
            pass
        time.sleep(0.05)


def idle_clock():
    out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
    m = re.search(r'sclk clock level:.*?\((\d+)Mhz\)', out)
    return int(m.group(1)) if m else None


idle = idle_clock()
th = threading.Thread(target=sample)
th.start()
from ann_solo_amd import _lib
L = _lib.lib()
import ctypes as C
L.asl_profile_enable(1)
L.asl_profile_reset()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        idx.coarse(xq, 128)
    torch.cuda.synchronize()
    n += 20
el = time.perf_counter() - t0
stop = True
th.join()
ms, cnt = C.c_double(), C.c_int64()
L.asl_profile_get(b'coarse_gemm', C.byref(ms), C.byref(cnt))
L.asl_profile_enable(0)
flop = 2.0 * nq * nlist * d
per = ms.value / max(cnt.value, 1)
tf = flop / (per * 1e-3) / 1e12
print(f'idle sclk {idle} MHz; {cnt.value} GEMMs, {per:.3f} ms each = {tf:.1f} TFLOP/s')
if samples:
    s = sorted(samples)
    med = s[len(s) // 2]
    print(f'sclk under load: min {s[0]} median {med} max {s[-1]} MHz ({len(s)} samples); '
          f'fp32-MFMA peak at the median clock = {256 * 256 * med * 1e6 / 1e12:.1f} TFLOP/s '
          f'-> MFMA-pipe efficiency {tf / (256 * 256 * med * 1e6 / 1e12):.2f}')
else:
    print('rocm-smi gave no clock samples')
