"""Runs scripts/lds_atomic_bench.hip (see there): ms and ns per wave-instruction for the four
accumulator updates. Build first (hipcc line in the .hip file)."""
import ctypes as C
import os
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tmp', 'ldsbench.so')
L = C.CDLL(so)
L.run.restype = C.c_float
L.run.argtypes = [C.c_int, C.c_int, C.c_int]
blocks, iters = 256 * 3 * 4, 2000
for mode, name in enumerate(['read-fma-write', 'ds_add_f32', 'ds_add_u32', 'ds_add_u64']):
    ms = L.run(mode, iters, blocks)
    n = blocks * 8 * iters * 8           # wave-instructions (8 waves x 8 updates per iteration)
    print(f'{name:16s} {ms:8.2f} ms  {n / (ms * 1e-3) / 256 / 1e6:8.1f} M wave-instructions/s per CU '
          f'= one per {256 * 2.4e9 * ms * 1e-3 / n:6.1f} cycles per CU')
