"""Line / row counts of IVF-Flat posting layouts on the bench library (model, no kernel runs).

    python scripts/flat_layout_model.py [library_size=2100000] [queries=2048] [nprobe=112]

Builds the bench's library and coarse quantiser (through the product's IVF-PQ index: same list
membership as IVF-Flat), takes the probe lists of the bench's queries and counts, per layout,
the 128-byte lines and the wave-rows one scan launch would touch:

  A  the round-3 layout: blocks of 832 vectors, 6-byte postings (f32 value + u16 index) in
     segments placed by 64-byte units behind a 4-byte table word per (block, dimension)
  F  the same with 4-byte postings
  G  fixed slots: one 128-byte line of 32 four-byte postings per (block, dimension), no table;
     dimensions with more postings continue in overflow rows listed in a one-line directory
     per block; lists are cut evenly into blocks of at most B vectors
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_100_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NPROBE = int(sys.argv[3]) if len(sys.argv) > 3 else 112
NLIST = 4096 if N >= 1_000_000 else max(16, N // 512)
D = 800


def main():
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(N, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
    cfg = Config.open_search(num_list=NLIST, num_probe=NPROBE, num_candidates=1024, index='ivfpq', pq_m=32,
                 kmeans_niter=25, mode='ann', precursor_tolerance_mass_open=500.0,
                 precursor_tolerance_mode_open='Da', batch_size=16384, seed=1234)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    part = sl.partitions[2]
    idx = sl._get_ann_index(2)
    q_all, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
    q = q_all.select(torch.arange(NQ, device=dev))
    off, ids, _ = idx.lists()
    off = torch.from_numpy(off).to(dev).long()
    ids = torch.from_numpy(ids).to(dev).long()
    llen = off[1:] - off[:-1]
    print(f'lists: {NLIST}, length mean {llen.float().mean():.0f} min {int(llen.min())} '
          f'p10 {int(llen.float().quantile(0.1))} p50 {int(llen.float().median())} '
          f'p90 {int(llen.float().quantile(0.9))} max {int(llen.max())}')
    xq = sl._encode(q)
    _, probes = idx.coarse(xq, NPROBE)
    probes = probes.long()
    qnz = (xq != 0).float()                                   # [nq, d]
    print(f'query non-zeros mean {qnz.sum(1).mean():.1f}')
    # library non-zeros in list order
    vec = sl._encode(part.spectra)                            # [N, d]
    list_of_pos = torch.repeat_interleave(torch.arange(NLIST, device=dev), llen)
    rank_in_list = torch.arange(N, device=dev) - off[list_of_pos]

    def layout(bmax, even):
        """block id of every list-order position, blocks per list"""
        if even:
            nb = (llen + bmax - 1) // bmax
            nb = torch.clamp(nb, min=0)
            bsz = torch.where(nb > 0, (llen + torch.clamp(nb, min=1) - 1) // torch.clamp(nb, min=1), llen)
        else:
            nb = (llen + bmax - 1) // bmax
            bsz = torch.full_like(llen, bmax)
        boff = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(nb, 0)])
        blk = boff[list_of_pos] + rank_in_list // torch.clamp(bsz[list_of_pos], min=1)
        return blk, boff, int(boff[-1])

    def counts(blk, nblk):
        c = torch.zeros(nblk * D, dtype=torch.int32, device=dev)
        chunk = 262144
        for a in range(0, N, chunk):
            rows = ids[a:a + chunk]
            nz = (vec[rows] != 0).nonzero()
            key = blk[a + nz[:, 0]] * D + nz[:, 1]
            c.index_add_(0, key, torch.ones_like(key, dtype=torch.int32))
        return c.view(nblk, D)

    def per_query(M, boff, nblk):
        """sum over (query, probed block, query dimension) of M[block, dim]"""
        P = torch.zeros(NQ, nblk + 1, device=dev)
        nb = boff[1:] - boff[:-1]
        for j in range(int(nb.max())):
            b = boff[probes] + j
            ok = j < nb[probes]
            P.scatter_(1, torch.where(ok, b, torch.full_like(b, nblk)), 1.0)
        P = P[:, :nblk]
        visits = float(P.sum())
        tot = float(((qnz @ M.float().t()) * P).sum())
        return tot, visits

    def report(name, c, boff, nblk, lines_fn, rows_fn, per_visit_lines, table=False):
        lines, visits = per_query(lines_fn(c), boff, nblk)
        rows, _ = per_query(rows_fn(c), boff, nblk)
        post, _ = per_query(c, boff, nblk)
        tl = 0.0
        if table:      # distinct 128-byte lines (32 words) of a block's table row holding a wanted word
            tl = float((qnz.view(NQ, D // 32, 32).sum(2) > 0).float().sum(1).mean()) * visits
        lines_all = lines + tl + per_visit_lines * visits
        print(f'{name:34s} blocks {nblk:6d} visits/q {visits / NQ:6.1f} postings/q {post / NQ:9.0f} '
              f'lines/q {lines_all / NQ:8.0f} (seg {lines / NQ:7.0f} table {tl / NQ:6.0f} fixed {per_visit_lines * visits / NQ:5.0f}) '
              f'rows/q {rows / NQ:7.0f} lines/visit {lines_all / visits:6.1f}')
        return lines_all / NQ

    blk, boff, nblk = layout(832, False)
    c = counts(blk, nblk)
    print(f'postings per (block, dim): mean {c.float().mean():.1f}; dims > 32: {float((c > 32).float().mean()) * 100:.2f} %')

    def seg_lines(bytes_per):
        def f(c):
            b = c * bytes_per
            return torch.where(c == 0, torch.zeros_like(c), torch.where(b <= 128, torch.ones_like(c), (b + 127) // 128))
        return f
    rows64 = lambda c: (c + 63) // 64
    base = report('A 6-B postings, table, 832', c, boff, nblk, seg_lines(6), rows64, 0, table=True)
    report('F 4-B postings, table, 832', c, boff, nblk, seg_lines(4), rows64, 0, table=True)
    for bmax in (448, 512, 576, 640, 704, 832):
        blk, boff, nblk = layout(bmax, True)
        c = counts(blk, nblk)
        ovf = (c > 32)
        nov = ovf.sum(1)
        g_lines = lambda c: 1 + (torch.clamp(c - 32, min=0) + 31) // 32
        g_lines64 = lambda c: 1 + 2 * ((torch.clamp(c - 32, min=0) + 63) // 64)
        g_rows = lambda c: 1 + (torch.clamp(c - 32, min=0) + 31) // 32
        v = report(f'G slots, B<={bmax}, 32-wide overflow', c, boff, nblk, g_lines, g_rows, 1)
        print(f'    overflow dims per block: mean {nov.float().mean():.1f} p99 {int(nov.float().quantile(0.99))} '
              f'max {int(nov.max())}; blocks with > 31: {int((nov > 31).sum())}; lines vs A: {v / base:.3f}')
        report(f'G slots, B<={bmax}, 64-wide overflow', c, boff, nblk, g_lines64,
               lambda c: 1 + (torch.clamp(c - 32, min=0) + 63) // 64, 1)


if __name__ == '__main__' and not (len(sys.argv) > 4 and sys.argv[4] == 'pair'):
    main()


def pairing_model():
    """Passes and rows per query when two blocks whose sizes sum to <= 832 share a wave pass
    (half-wave each; an unpaired block keeps 64-lane rows of two lines): two-pointer pairing of
    the size-sorted probed blocks."""
    import numpy as np
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(N, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
    cfg = Config.open_search(num_list=NLIST, num_probe=NPROBE, num_candidates=1024, index='ivfpq', pq_m=32,
                 kmeans_niter=25, mode='ann', precursor_tolerance_mass_open=500.0,
                 precursor_tolerance_mode_open='Da', batch_size=16384, seed=1234)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    part = sl.partitions[2]
    idx = sl._get_ann_index(2)
    q_all, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
    nq = 256
    q = q_all.select(torch.arange(nq, device=dev))
    off, ids, _ = idx.lists()
    llen = np.diff(off)
    xq = sl._encode(q)
    _, probes = idx.coarse(xq, NPROBE)
    probes = probes.cpu().numpy()
    qnz = (xq != 0).cpu().numpy()
    vec = sl._encode(part.spectra)
    ids_t = torch.from_numpy(ids).to(dev).long()
    # blocks of 832 in list order, per-(block, dim) line counts
    nb = (llen + 831) // 832
    boff = np.concatenate([[0], np.cumsum(nb)])
    bsize, bstart = [], []
    for l in range(NLIST):
        for j in range(nb[l]):
            bstart.append(off[l] + j * 832)
            bsize.append(min(832, llen[l] - j * 832))
    bsize = np.array(bsize)
    nblk = len(bsize)
    blk_of_pos = np.repeat(np.arange(nblk), bsize)
    c = torch.zeros(nblk * D, dtype=torch.int32, device=dev)
    blk_t = torch.from_numpy(blk_of_pos).to(dev)
    for a in range(0, N, 262144):
        nz = (vec[ids_t[a:a + 262144]] != 0).nonzero()
        key = blk_t[a + nz[:, 0]] * D + nz[:, 1]
        c.index_add_(0, key, torch.ones_like(key, dtype=torch.int32))
    nl = ((c.view(nblk, D) + 31) // 32).cpu().numpy()
    tot = dict(passes=0, rows_pair=0, rows_now=0, visits=0, lines=0)
    for i in range(nq):
        blks = np.concatenate([np.arange(boff[l], boff[l + 1]) for l in probes[i]])
        dims = np.nonzero(qnz[i])[0]
        order = blks[np.argsort(bsize[blks], kind='stable')]
        lo, hi = 0, len(order) - 1
        tot['visits'] += len(order)
        while lo <= hi:
            A = order[hi]
            nlA = nl[A][dims]
            tot['rows_now'] += int(((nlA + 1) // 2).sum())
            tot['lines'] += int(nlA.sum())
            if lo < hi and bsize[order[lo]] + bsize[A] <= 832:
                B = order[lo]
                nlB = nl[B][dims]
                tot['rows_now'] += int(((nlB + 1) // 2).sum())
                tot['lines'] += int(nlB.sum())
                tot['rows_pair'] += int(np.maximum(nlA, nlB).sum())
                lo += 1
            else:
                tot['rows_pair'] += int(((nlA + 1) // 2).sum())
            hi -= 1
            tot['passes'] += 1
    print({k: v / nq for k, v in tot.items()})


if __name__ == '__main__' and len(sys.argv) > 4 and sys.argv[4] == 'pair':
    pairing_model()
