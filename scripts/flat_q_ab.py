"""Same-box timing of the IVF-Flat postings scan with one block per wave (ASL_FLAT_Q=0, flat_scan.hip)
against four block streams per wave (ASL_FLAT_Q=1; 2 = with phase timers): the round-5 experiment of
profiles/r05_flat_scan_notes.txt. The second kernel lives in commit eb62a9e only (csrc/flat_scan_q.hip;
measured 2.3 x slower and removed), so on the current tree all three runs time the same kernel.
python scripts/flat_q_ab.py [nprobe] [storage] [steps]"""
import ctypes as C
import os
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import numpy as np
    import torch
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    nprobe, storage, steps, out = int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
    cfg = Config.open_search(num_list=4096, num_probe=nprobe, num_candidates=1024, index='ivfflat', kmeans_niter=25, mode='ann',
                 precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da', batch_size=16384,
                 seed=1234, flat_storage=storage)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    idx = sl._get_ann_index(2)
    q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
    L = _lib.lib()
    vec = sl._encode(q)
    idx.nprobe = nprobe
    D, I = idx.search(vec, 1024)
    torch.save((D.cpu(), I.cpu()), out)
    for pipelined in (False, True):
        sl.set_pipeline(pipelined)
        for _ in range(2):
            sl._search_batch(q, 2, 'open', device_out=True)
        sl.synchronize()
        L.asl_profile_reset()
        L.asl_profile_enable(2 if pipelined else 1)
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sl._search_batch(q, 2, 'open', device_out=True)
        sl.synchronize()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps * 1e3
        L.asl_profile_enable(0)
        ms, n = C.c_double(), C.c_int64()
        L.asl_profile_get(b'scan', C.byref(ms), C.byref(n))
        print(f'ASL_FLAT_Q={os.environ.get("ASL_FLAT_Q", "(default: 1)")} storage {storage} layout {idx.flat_layout} nprobe {nprobe} '
              f'{"pipelined" if pipelined else "serial   "}: step {el:.3f} ms, scan {ms.value / max(n.value, 1):.4f} ms', flush=True)
    sl.set_pipeline(False)
    if os.environ.get('ASL_FLAT_Q') == '2' and hasattr(L, 'asl_debug_flat_q_prof'):       # the measurement build's phase timers
        import numpy as np
        buf = np.zeros(16, np.uint64)
        L.asl_debug_flat_q_prof(buf.ctypes.data_as(C.c_void_p))        # (clear)
        idx.search(vec, 1024)
        L.asl_debug_flat_q_prof(buf.ctypes.data_as(C.c_void_p))
        names = ['prologue', 'take', 'zero', 'table+lookup', 'emit', 'rows', 'cold', 'offers', 'sync', 'end-wait', 'finish']
        tot = float(buf[:11].sum())
        print('wave-cycles per phase (one launch): ' + '  '.join(f'{n_} {int(v)/1e9:.2f}G ({100*int(v)/tot:.0f}%)' for n_, v in zip(names, buf)), flush=True)
    sl.shutdown()
    sys.exit(0)

nprobe = sys.argv[1] if len(sys.argv) > 1 else '112'
storage = sys.argv[2] if len(sys.argv) > 2 else 'fx22'
steps = sys.argv[3] if len(sys.argv) > 3 else '10'
outs = []
for flag in ('0', '1', '2'):
    out = f'/tmp/flat_q_ab_{flag}.pt'
    outs.append(out)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child', nprobe, storage, steps, out],
                   env=dict(os.environ, ASL_FLAT_Q=flag), check=True)
import torch
(D0, I0), (D1, I1) = torch.load(outs[0]), torch.load(outs[1])
assert torch.equal(torch.load(outs[2])[1], I1)
print('ids equal:', bool(torch.equal(I0, I1)), ' score bits equal:', bool(torch.equal(D0.view(torch.int32), D1.view(torch.int32))),
      ' rows differing:', int((I0 != I1).any(1).sum()))
