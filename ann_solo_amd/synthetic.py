"""Seeded synthetic spectral libraries and query batches.

Modelled on the reference's own test generator
(/root/reference/src/tests/query_reader_test.py:73-99: b/y ladders from
monoisotopic residue masses). Spectra come out *already processed* the way
``process_spectrum`` (spectrum.py:57-119) leaves them: m/z within [11, 2010],
<= 50 most intense peaks, rank-scaled, L2-normalised, ascending m/z, >= 10 peaks
spanning >= 250 m/z.

Written against torch so the same code makes small CPU fixtures for the parity
tests and the 2.1 M-spectrum library directly in HBM for bench.py.
"""
import math
from typing import Dict, Tuple

import torch

from .packed import PackedSpectra

PROTON = 1.007276466
WATER = 18.0105647
# G A S P V T C(+57) L I N D Q K E M H F R Y W  (monoisotopic residue masses)
RES_MASS = [57.02146, 71.03711, 87.03203, 97.05276, 99.06841, 101.04768, 160.03065,
            113.08406, 113.08406, 114.04293, 115.02694, 128.05858, 128.09496, 129.04259,
            131.04049, 137.05891, 147.06841, 156.10111, 163.06333, 186.07931]
IDX_K, IDX_R = 12, 17
MAX_LEN = 30
MIN_LEN = 7
MAX_FRAG_Z = 3
PTM_MASSES = [15.9949, 57.0215, 27.9949, 0.9840, -17.0265, 79.9663]

MIN_MZ, MAX_MZ = 11.0, 2010.0
MAX_PEAKS = 50
MIN_PEAKS = 10
MIN_MZ_RANGE = 250.0
MIN_INTENSITY = 0.01


def _process_padded(mz: torch.Tensor, inten: torch.Tensor, ann: torch.Tensor, extra=()):
    """process_spectrum on padded [n,P] arrays (intensity 0 = absent peak).

    Returns (count[n], mz[n,50] f32, intensity[n,50] f32, ann[n,50], extras..., valid[n]);
    rows are ascending in m/z with the absent slots at the end.
    """
    n, P = mz.shape
    dev = mz.device
    ok = (inten > 0) & (mz >= MIN_MZ) & (mz <= MAX_MZ)
    inten = torch.where(ok, inten, torch.zeros_like(inten))
    mx = inten.max(dim=1, keepdim=True).values
    inten = torch.where(inten > MIN_INTENSITY * mx, inten, torch.zeros_like(inten))
    if P < MAX_PEAKS:
        pad = MAX_PEAKS - P
        mz = torch.cat([mz, torch.zeros(n, pad, dtype=mz.dtype, device=dev)], 1)
        inten = torch.cat([inten, torch.zeros(n, pad, dtype=inten.dtype, device=dev)], 1)
        ann = torch.cat([ann, torch.zeros(n, pad, dtype=ann.dtype, device=dev)], 1)
        extra = tuple(torch.cat([e, torch.zeros(n, pad, dtype=e.dtype, device=dev)], 1)
                      for e in extra)
    top_i, top_idx = torch.topk(inten, MAX_PEAKS, dim=1)          # descending intensity
    present = top_i > 0
    cnt = present.sum(1)
    # rank scaling: base peak -> max_rank, next -> max_rank-1, ...
    rank_int = (MAX_PEAKS - torch.arange(MAX_PEAKS, device=dev)).to(torch.float32)
    scaled = torch.where(present, rank_int.expand(n, -1), torch.zeros((), device=dev))
    scaled = scaled / torch.sqrt((scaled * scaled).sum(1, keepdim=True)).clamp_min(1e-30)
    mz_k = torch.gather(mz, 1, top_idx).to(torch.float32)
    mz_key = torch.where(present, mz_k, torch.full_like(mz_k, float('inf')))
    order = torch.argsort(mz_key, dim=1, stable=True)
    mz_s = torch.gather(mz_k, 1, order)
    in_s = torch.gather(scaled, 1, order)
    raw_s = torch.gather(top_i, 1, order)
    ann_s = torch.gather(torch.gather(ann, 1, top_idx), 1, order)
    ex_s = tuple(torch.gather(torch.gather(e, 1, top_idx), 1, order) for e in extra)
    first = mz_s[:, 0]
    last = torch.gather(mz_s, 1, (cnt - 1).clamp_min(0).unsqueeze(1)).squeeze(1)
    valid = (cnt >= MIN_PEAKS) & ((last - first) >= MIN_MZ_RANGE)
    return (cnt, mz_s, in_s, ann_s, raw_s) + ex_s + (valid,)


def _pack(cnt, cols, rows_mask):
    """Pack padded [n,50] columns of the selected rows into flat arrays."""
    cnt = cnt[rows_mask]
    n = cnt.numel()
    dev = cnt.device
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(cnt, 0)
    slot = torch.arange(MAX_PEAKS, device=dev).expand(n, -1) < cnt.unsqueeze(1)
    flat = tuple(c[rows_mask][slot] for c in cols)
    return offsets.to(torch.int32), flat


def make_library(n: int, seed: int = 20240807, device='cpu', charges=(2, 3, 4),
                 charge_p=(0.55, 0.35, 0.10), chunk: int = 1 << 17,
                 annotate_frac: float = 0.8) -> Tuple[PackedSpectra, Dict[str, torch.Tensor]]:
    """``n`` processed library spectra (invalid draws are redrawn, so exactly ``n``)."""
    dev = torch.device(device)
    res_mass = torch.tensor(RES_MASS, dtype=torch.float64, device=dev)
    parts = []
    made = 0
    ci = 0
    while made < n:
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1000003 + ci)
        ci += 1
        m = min(chunk, max(1024, int((n - made) * 1.05) + 16))
        length = torch.randint(MIN_LEN, MAX_LEN + 1, (m,), generator=g, device=dev)
        res = torch.randint(0, 20, (m, MAX_LEN), generator=g, device=dev)
        kr = torch.where(torch.rand(m, generator=g, device=dev) < 0.5, IDX_K, IDX_R)
        res.scatter_(1, (length - 1).unsqueeze(1), kr.unsqueeze(1))
        pos = torch.arange(MAX_LEN, device=dev)
        mass = res_mass[res] * (pos < length.unsqueeze(1))
        zsel = torch.multinomial(torch.tensor(charge_p, dtype=torch.float32, device=dev), m,
                                 replacement=True, generator=g)
        z = torch.tensor(charges, device=dev)[zsel]
        pre = torch.cumsum(mass, 1)                                   # [m,30]
        total = pre[:, -1]
        i = torch.arange(1, MAX_LEN, device=dev)                      # fragment index 1..29
        frag_ok = i.unsqueeze(0) < length.unsqueeze(1)                # [m,29]
        b_neutral = pre[:, :MAX_LEN - 1]
        y_idx = (length.unsqueeze(1) - i.unsqueeze(0) - 1).clamp_min(0)
        y_neutral = total.unsqueeze(1) - torch.gather(pre, 1, y_idx) + WATER
        neutral = torch.stack([b_neutral, y_neutral], 1)              # [m,2,29]
        c = torch.arange(1, MAX_FRAG_Z + 1, device=dev).to(torch.float64)
        mz = (neutral.unsqueeze(-1) + c * PROTON) / c                 # [m,2,29,3]
        c_ok = c.unsqueeze(0) <= (z.unsqueeze(1) - 1).clamp_min(1)    # [m,3]
        ok = frag_ok.unsqueeze(1).unsqueeze(-1) & c_ok.unsqueeze(1).unsqueeze(1)
        ok = ok.expand(-1, 2, -1, -1)
        raw = torch.exp(torch.randn(mz.shape, generator=g, device=dev))
        raw = raw * torch.tensor([1.0, 0.35, 0.15], device=dev)
        raw = raw * torch.tensor([0.7, 1.0], device=dev).view(1, 2, 1, 1)
        raw = torch.where(ok, raw, torch.zeros((), device=dev)).to(torch.float32)
        P = 2 * (MAX_LEN - 1) * MAX_FRAG_Z
        fc = c.to(torch.uint8).view(1, 1, 1, -1).expand(m, 2, MAX_LEN - 1, -1)
        annot = torch.where(torch.rand(mz.shape, generator=g, device=dev) < annotate_frac, fc,
                            torch.zeros((), dtype=torch.uint8, device=dev))
        itype = torch.tensor([0, 1], dtype=torch.uint8, device=dev).view(1, 2, 1, 1).expand_as(fc)
        iidx = i.to(torch.uint8).view(1, 1, -1, 1).expand_as(fc)
        cnt, mz_s, in_s, ann_s, raw_s, it_s, ii_s, fc_s, valid = _process_padded(
            mz.reshape(m, P), raw.reshape(m, P), annot.reshape(m, P),
            extra=(itype.reshape(m, P), iidx.reshape(m, P), fc.reshape(m, P)))
        keep = valid.clone()
        over = int(keep.sum()) - (n - made)
        if over > 0:                       # drop surplus valid rows from the tail
            idx = torch.nonzero(keep).squeeze(1)
            keep[idx[-over:]] = False
        offsets, flat = _pack(cnt, (mz_s, in_s, ann_s, raw_s, it_s, ii_s, fc_s), keep)
        pmz = ((total + WATER + z * PROTON) / z)[keep]
        parts.append((offsets, flat, pmz, z[keep].to(torch.int32), length[keep].to(torch.int32)))
        made += int(keep.sum())
    offs = [parts[0][0]]
    base = int(parts[0][0][-1])
    for p in parts[1:]:
        offs.append(p[0][1:] + base)
        base += int(p[0][-1])
    cat = lambda k: torch.cat([p[1][k] for p in parts])
    lib = PackedSpectra(torch.cat(offs).to(torch.int32), cat(0), cat(1), cat(2),
                        torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts]))
    aux = dict(raw_intensity=cat(3), ion_type=cat(4), ion_idx=cat(5), frag_charge=cat(6),
               pep_len=torch.cat([p[4] for p in parts]))
    return lib, aux


# ``hard`` mode, calibrated on the bench library (2.1 M spectra, seed 20240807; scripts/tune_hard.py,
# profiles/r05_hard_calibration.txt) against the reference's one behavioural anchor at this boundary:
# with EXACT inner-product search the true match of a MODIFIED spectrum is inside the top 1024 for
# 75.1 % of the iPRG2012 SSMs (notebooks/iprg2012_num_candidates.ipynb:282-288). The default queries
# give 97.8 % -- far cleaner than real spectra of modified peptides.
HARD_DEFAULT = 0.21


def hard_levers(h: float) -> Dict[str, float]:
    """The knobs of ``make_queries(hard=h)``, h in [0, 1] (0 = the default generator): dropped
    peaks, intensity noise (rank changes), fragment m/z jitter (0.04-wide hash bins), noise peaks
    and their level, and co-fragmentation (the peaks of a second, random library spectrum)."""
    h = float(h)
    return dict(drop=0.10 + 0.45 * h, int_sigma=0.3 + 0.9 * h, jitter=0.005 + 0.010 * h,
                extra_noise=int(round(40 * h)), noise_level=0.15 + 0.45 * h, chimera=0.7 * h)


def make_queries(lib: PackedSpectra, aux: Dict[str, torch.Tensor], nq: int, seed: int = 42,
                 mod_frac: float = 0.5, open_range: float = 500.0,
                 charge: int = None, hard: float = 0.0) -> Tuple[PackedSpectra, Dict[str, torch.Tensor]]:
    """Query spectra re-drawn from library peptides: half unmodified (5 ppm precursor
    jitter), half with one PTM on a random residue; 0.005 Da fragment jitter, 10 % of
    the peaks dropped, 10 noise peaks. ``hard`` > 0 (``hard_levers``): noisier spectra --
    more dropped peaks, stronger intensity noise, larger jitter, more and stronger noise
    peaks, a co-fragmented second spectrum; ``hard`` = 0 draws exactly the default queries."""
    lv = hard_levers(hard)
    dev = lib.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if charge is None:
        pool = torch.arange(lib.n, device=dev)
    else:
        pool = torch.nonzero(lib.precursor_charge == charge).squeeze(1)
    out_parts = []
    truth_src, truth_mod, truth_dm = [], [], []
    chunk = 1 << 16
    for c0 in range(0, nq, chunk):
        m = min(chunk, nq - c0)
        # oversample: a few percent of the draws fail the validity check
        mm = int(m * 1.08) + 8
        src = pool[torch.randint(0, pool.numel(), (mm,), generator=g, device=dev)]
        off = lib.offsets.to(torch.int64)
        cnt = (off[src + 1] - off[src])
        slot = torch.arange(MAX_PEAKS, device=dev).expand(mm, -1)
        have = slot < cnt.unsqueeze(1)
        pos = (off[src].unsqueeze(1) + slot).clamp_max(lib.mz.numel() - 1)
        mz = torch.where(have, lib.mz[pos].to(torch.float64), torch.zeros((), dtype=torch.float64, device=dev))
        raw = torch.where(have, aux['raw_intensity'][pos], torch.zeros((), device=dev))
        it, ii, fc = aux['ion_type'][pos], aux['ion_idx'][pos].to(torch.int64), aux['frag_charge'][pos]
        z = lib.precursor_charge[src].to(torch.float64)
        plen = aux['pep_len'][src].to(torch.int64)
        is_mod = torch.rand(mm, generator=g, device=dev) < mod_frac
        kind = torch.randint(0, len(PTM_MASSES) + 1, (mm,), generator=g, device=dev)
        table = torch.tensor(PTM_MASSES + [0.0], dtype=torch.float64, device=dev)
        uni = (torch.rand(mm, generator=g, device=dev, dtype=torch.float64) * 2 - 1) * open_range
        dm = torch.where(kind == len(PTM_MASSES), uni, table[kind])
        dm = torch.where(is_mod, dm, torch.zeros_like(dm))
        ppos = (torch.rand(mm, generator=g, device=dev) * plen).to(torch.int64).clamp_max(plen - 1)
        # b_i carries the residue if i > pos ; y_i if i >= len - pos
        shifted = torch.where(it == 0, ii > ppos.unsqueeze(1), ii >= (plen - ppos).unsqueeze(1))
        mz = mz + torch.where(shifted, dm.unsqueeze(1) / fc.clamp_min(1).to(torch.float64),
                              torch.zeros((), dtype=torch.float64, device=dev))
        mz = mz + lv['jitter'] * torch.randn(mz.shape, generator=g, device=dev, dtype=torch.float64)
        raw = raw * torch.exp(lv['int_sigma'] * torch.randn(raw.shape, generator=g, device=dev))
        raw = torch.where(torch.rand(raw.shape, generator=g, device=dev) < lv['drop'],
                          torch.zeros((), device=dev), raw)
        nmz = 100 + 1400 * torch.rand(mm, 10, generator=g, device=dev, dtype=torch.float64)
        med = raw.max(1, keepdim=True).values * lv['noise_level']
        nin = med * torch.exp(0.5 * torch.randn(mm, 10, generator=g, device=dev))
        mz = torch.cat([mz, nmz], 1)
        raw = torch.cat([raw, nin], 1)
        hard_draws = hard > 0           # (extra draws come last in the chunk: hard = 0 is the default stream)
        pmz = lib.precursor_mz[src] + dm / z
        ppm = 5e-6 * torch.randn(mm, generator=g, device=dev, dtype=torch.float64)
        pmz = pmz * (1 + ppm)
        if hard_draws:
            ne = lv['extra_noise']
            if ne > 0:
                emz = 100 + 1400 * torch.rand(mm, ne, generator=g, device=dev, dtype=torch.float64)
                ein = med * torch.exp(0.5 * torch.randn(mm, ne, generator=g, device=dev))
                mz, raw = torch.cat([mz, emz], 1), torch.cat([raw, ein], 1)
            if lv['chimera'] > 0:       # co-fragmentation: a second precursor's fragments in the same scan
                oth = pool[torch.randint(0, pool.numel(), (mm,), generator=g, device=dev)]
                ocnt = off[oth + 1] - off[oth]
                ohave = slot < ocnt.unsqueeze(1)
                opos = (off[oth].unsqueeze(1) + slot).clamp_max(lib.mz.numel() - 1)
                omz = torch.where(ohave, lib.mz[opos].to(torch.float64), torch.zeros((), dtype=torch.float64, device=dev))
                oraw = torch.where(ohave, aux['raw_intensity'][opos], torch.zeros((), device=dev))
                scale = lv['chimera'] * raw[:, :MAX_PEAKS].max(1, keepdim=True).values / oraw.max(1, keepdim=True).values.clamp_min(1e-30)
                oraw = oraw * scale * torch.exp(0.5 * torch.randn(oraw.shape, generator=g, device=dev))
                mz, raw = torch.cat([mz, omz], 1), torch.cat([raw, oraw], 1)
        ann = torch.zeros(mz.shape, dtype=torch.uint8, device=dev)
        cnt2, mz_s, in_s, ann_s, raw_s, valid = _process_padded(mz, raw, ann)
        keep = valid.clone()
        over = int(keep.sum()) - m
        if over > 0:
            idx = torch.nonzero(keep).squeeze(1)
            keep[idx[-over:]] = False
        if int(keep.sum()) < m:
            raise RuntimeError('synthetic query generator: too many invalid draws')
        offsets, flat = _pack(cnt2, (mz_s, in_s, ann_s), keep)
        out_parts.append((offsets, flat, pmz[keep], lib.precursor_charge[src][keep]))
        truth_src.append(src[keep])
        truth_mod.append(is_mod[keep])
        truth_dm.append(dm[keep])
    offs = [out_parts[0][0]]
    base = int(out_parts[0][0][-1])
    for p in out_parts[1:]:
        offs.append(p[0][1:] + base)
        base += int(p[0][-1])
    cat = lambda k: torch.cat([p[1][k] for p in out_parts])
    q = PackedSpectra(torch.cat(offs).to(torch.int32), cat(0), cat(1), cat(2),
                      torch.cat([p[2] for p in out_parts]), torch.cat([p[3] for p in out_parts]))
    truth = dict(source_row=torch.cat(truth_src), is_modified=torch.cat(truth_mod),
                 delta_mass=torch.cat(truth_dm))
    return q, truth
