#!/usr/bin/env python3
"""Benchmark of the open-modification search hot path on MI355X.

One "step" = one batch of 32 768 same-charge query spectra through the whole hot
path, device resident: encode (feature hashing) -> IVF-PQ retrieve (coarse GEMM on
MFMA, ADC scan, top-1024) -> precursor-window post-filter -> (shifted) dot-product
best match. Workload = BASELINE.json configs[2]: a MassIVE-KB-scale synthetic
library (2.1 M processed spectra in one precursor-charge partition), IVF-PQ m=32,
nlist=4096, nprobe=128, k=1024, open window +-500 Da, fragment tolerance 0.02 Da.

  python bench.py --gpus N --steps K --warmup W
  (N>1: the same line -- without a launcher's WORLD_SIZE in the environment the process starts the N
  ranks itself as children of `python -m torch.distributed.run`; launched BY torch.distributed.run,
  as the driver does, every rank runs main() directly)

With N>1 the IVF lists are sharded over the ranks (ann_solo_amd/distributed.py) and
every rank contributes its own 32 768-query slice per step (weak scaling).
Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how roofline.achieved
(algorithmic bytes of the PQ scan / HIP-event kernel time) and cpu_baseline (the
oracle on the host cores, bounded sample) are defined.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0        # what a float4 copy kernel reaches of it (same guide, "HBM": 79 %)
LLC_BYTES = 256 << 20        # Infinity Cache (L3), die level
BYTES_PER_SCANNED_VECTOR = 36  # 32-B PQ code + 4-B id
BYTES_PER_CODE = 32            # the codes-only variant SURVEY.md 8(d) asks to report too


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The contract is ONE JSON line on stdout. Libraries under a rank write there too (Gloo announces its
# connections on stdout, once per group): protect_stdout() points file descriptor 1 at stderr for the
# rest of the process and keeps the real stdout for emit_line() alone.
_REAL_STDOUT = None


def protect_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)


def emit_line(line):
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(line + '\n')
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--library-size', type=int, default=2_100_000)
    ap.add_argument('--batch', type=int, default=32768,
                    help='query spectra per step and GPU (32 768: the throughput no longer grows beyond it -- '
                         '16 384: 2.19 M, 32 768: 2.29 M, 65 536: 2.29 M spectra/s, profiles/r05_batch_size.txt)')
    ap.add_argument('--nlist', type=int, default=4096)
    ap.add_argument('--nprobe', type=int, default=128)
    ap.add_argument('--k', type=int, default=1024)
    ap.add_argument('--index', default='ivfpq', choices=['ivfpq', 'ivfflat'])
    ap.add_argument('--pq-m', type=int, default=32)
    ap.add_argument('--niter', type=int, default=25)
    ap.add_argument('--open-da', type=float, default=500.0)
    ap.add_argument('--scan-variant', type=int, default=0,
                    help='0 = the layout-specific kernels (default), 1 = the generic ones (asl_index_set_scan_variant)')
    ap.add_argument('--recall-queries', type=int, default=2048)
    ap.add_argument('--refine-k', type=int, default=0,
                    help="IVF-PQ: exact re-rank of the k' best ADC candidates (asl_index_set_refine); "
                         '0 = off (the default workload)')
    ap.add_argument('--no-recall-hard', action='store_true',
                    help='skip the `recall_hard` block (recall / hit@k / the fixed-recall operating point on the '
                         'HARD synthetic queries, calibrated to the reference\'s exact-search hit@1024 of 75 %%)')
    ap.add_argument('--no-fixed-recall', action='store_true',
                    help='skip the IVF-Flat measurement at the fixed-recall operating point '
                         '(default run, N = 1, IVF-PQ: same library, same coarse quantiser)')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='run the stages of consecutive batches strictly one after the other '
                         '(default: two-stream software pipeline, asl_set_pipeline)')
    ap.add_argument('--workload', default='batch', choices=['batch', 'cascade'],
                    help="'batch' (default): BASELINE configs[2]/[3], one 32 768-query open-search "
                         "batch per step; 'cascade': configs[4], standard search -> FDR gate -> "
                         "open search of the unidentified remainder over the same library")
    ap.add_argument('--cascade-batches', type=int, default=2,
                    help='cascade workload: query batches per GPU in one pass')
    ap.add_argument('--accept-cosine', type=float, default=0.7,
                    help='cascade workload: stand-in for the mokapot/FDR gate between the levels '
                         '(utils.score_ssms is out of scope): accept an SSM when its cosine >= this')
    ap.add_argument('--shard-degree', type=int, default=0,
                    help='N > 1: ranks per shard group (lists sharded inside a group, groups are '
                         'replicas); 0 = N, the fully list-sharded layout of the north star; '
                         '1 = replicas only')
    ap.add_argument('--exchange', default='two-phase', choices=['two-phase', 'full'],
                    help="N > 1: 'two-phase' (default) = heads of ~2k/N keys per (query, shard), a "
                         "bound, then the held-back keys above it (exact; csrc/exchange.hip); 'full' "
                         "= every shard's whole top-k row")
    ap.add_argument('--head-keys', type=int, default=None,
                    help='N > 1, two-phase exchange: keys per head instead of ceil(2k / N) (diagnostic: '
                         'small heads force the bound / held-back-keys round at any N)')
    ap.add_argument('--extras-per-query', type=int, default=None,
                    help='N > 1, two-phase exchange: phase-2 buffer slots per query and pair of ranks '
                         '(default max(8, k / 16); diagnostic: 0 forces the fallback to the full exchange)')
    ap.add_argument('--shard-keys', type=int, default=None,
                    help='N > 1: the shards\' own k (default asl_shard_k(k, N): k / 2 from 8 ranks on, 5 k / 8 '
                         'from 4, k below); below k a shard scans again, with the full k, the rows whose bound '
                         'lies under the smallest key of a full row (csrc/exchange.hip; exact)')
    ap.add_argument('--sharded-watchdog-seconds', type=float, default=900.0,
                    help='N > 1, list-sharded layout: once the replicas-only layout has been measured, a '
                         'watchdog prints that measurement as the run\'s line (marked as the fallback) and ends the '
                         'process with code 0 if the sharded path has not finished within this many seconds; '
                         '0 = no watchdog (exceptions still fall back)')
    ap.add_argument('--preflight-seconds', type=float, default=120.0,
                    help='N > 1: every collective of the sharded path is first run at a tiny size '
                         'under a watchdog that ends the process (exit code 17, a diagnostic line on '
                         'stderr) if it has not finished within this many seconds; 0 = skip')
    ap.add_argument('--no-reference-geometry', action='store_true',
                    help='skip the two legs at the reference\'s own defaults (JSON field `reference_geometry`: '
                         'configs[1] = a 9 k-spectrum library, and the 2.1 M library, both IVF-Flat nlist 256 / '
                         'nprobe 128, src/ann_solo/config.py:188-211)')
    ap.add_argument('--beyond-llc-chunks', type=int, default=16,
                    help='default N = 1 IVF-PQ run: the same scan kernel over an index grown to this many '
                         'library-sized chunks (16 x 2.1 M = 33.6 M vectors: 1.08 GB of codes, four times the 256 MiB '
                         'Infinity Cache) -- `roofline.beyond_llc`, the figure that IS served by HBM; 0 = skip')
    ap.add_argument('--no-cascade', action='store_true',
                    help='skip the configs[4] cascade pass of the default N = 1 run (JSON field `cascade`)')
    ap.add_argument('--ring', type=int, default=4,
                    help='distinct query batches the timed loop cycles through (step i searches batch '
                         'i mod ring; every leg that is not `value` uses batch 0)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0,
                    help='target core-seconds of the CPU baseline sample (0 = skip)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks. Nothing here
        # has touched the GPU (torch is not even imported yet) and the process is not replaced:
        # the ranks are CHILDREN, rank 0's JSON line passes through, their exit code is ours.
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world
    protect_stdout()
    if os.environ.get('ASL_BENCH_LAUNCH_CHECK') == '1':
        sys.exit(launch_check(args, world, rank))

    import numpy as np
    import torch
    import torch.distributed as dist
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    from ann_solo_amd import faiss_compat as faiss
    from ann_solo_amd.distributed import (CommLog, HipShardBackend, make_shard_groups,
                                          sharded_search_batch)

    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no CPU fallback exists)')
    # ASL_BENCH_BACKEND=gloo runs every rank on GPU 0 with host-side collectives: a
    # functional check of the sharded path on a 1-GPU box, never a performance number
    backend = os.environ.get('ASL_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    _lib.check(_lib.lib().asl_set_device(dev_index))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # CONTROL plane = gloo on host tensors, always (barriers, the max-over-ranks clock, parity
        # flags): it does not depend on RCCL / xGMI coming up. The DATA plane -- the collectives of the
        # list-sharded search -- is its own RCCL group, created only once the replicas layout has
        # been measured and that line is held in reserve (data_plane() below): no multi-GPU
        # hardware was available to any round, so nothing RCCL does may stand between the start
        # of a run and its first complete result.
        dist.init_process_group('gloo')
        if backend != 'nccl' and os.environ.get('ASL_DIRECT_COLLECTIVES') == '1':
            # the collectives in their RCCL form (all_to_all_single / all_gather_into_tensor on the
            # device tensors as they are, asynchronous handles, four pieces per batch) over gloo:
            # what tests/test_gpu_distributed.py can run of the multi-GPU path on a 1-GPU box
            from ann_solo_amd import distributed as _dist_mod
            _dist_mod.FORCE_DIRECT_COLLECTIVES = True

    data = {'ready': False, 'world_group': None}

    def data_plane():
        """The RCCL group of all ranks (None = the default gloo group under ASL_BENCH_BACKEND=gloo),
        created and preflighted on first use."""
        if not data['ready']:
            _FALLBACK['stage'] = 'creating the RCCL group of the data plane'
            if backend == 'nccl':
                try:
                    data['world_group'] = dist.new_group(list(range(world)), backend='nccl', device_id=dev)
                except TypeError:       # an older new_group without device_id
                    data['world_group'] = dist.new_group(list(range(world)), backend='nccl')
            if args.preflight_seconds > 0:
                _FALLBACK['stage'] = 'preflight of the collectives'
                preflight(args, world, rank, dev, backend, data['world_group'])
            data['ready'] = True
        return data['world_group']

    t_build = time.time()
    charge = 2
    lib, aux = synthetic.make_library(args.library_size, seed=20240807, device=dev,
                                      charges=(charge,), charge_p=(1.0,))
    cfg = Config.open_search(num_list=args.nlist, num_probe=args.nprobe, num_candidates=args.k,
                 index=args.index, pq_m=args.pq_m, kmeans_niter=args.niter, mode='ann',
                 precursor_tolerance_mass_open=args.open_da, precursor_tolerance_mode_open='Da',
                 batch_size=args.batch, seed=1234, refine_k=args.refine_k or None)
    sl = SpectralLibrary(lib, config=cfg, device=dev)
    part = sl.partitions[charge]
    idx = sl._get_ann_index(charge)
    idx.set_scan_variant(args.scan_variant)
    torch.cuda.synchronize()
    if rank == 0:
        log(f'[bench] library {lib.n} spectra, {lib.mz.numel()} peaks; index {args.index} '
            f'nlist={args.nlist} built in {time.time() - t_build:.1f}s')

    if args.workload == 'cascade':
        run_cascade(args, world, rank, dev, backend, sl, lib, aux, charge, cfg,
                    data_plane() if world > 1 else None)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- queries: ring x world x batch, identical on every rank; ring element j = queries
    # [j * world * batch, (j + 1) * world * batch), of which each rank owns one slice. Element 0 is
    # the batch of every earlier round (same seed, same draws: the generator works in chunks of 65 536
    # whose first world * batch queries do not depend on how many follow... they do when world * batch
    # * ring crosses a chunk, so element 0 is drawn on its own and the rest from another seed)
    ring = max(1, args.ring)
    q_all, truth = synthetic.make_queries(lib, aux, world * args.batch, seed=42,
                                          open_range=args.open_da, charge=charge)
    sl_rows = torch.arange(rank * args.batch, (rank + 1) * args.batch, device=dev)
    q = q_all.select(sl_rows).contiguous()
    src_local = truth['source_row'][sl_rows]
    mod_local = truth['is_modified'][sl_rows]
    q_ring = [q]
    if ring > 1:
        q_more, _ = synthetic.make_queries(lib, aux, (ring - 1) * world * args.batch, seed=4242,
                                           open_range=args.open_da, charge=charge)
        for j in range(ring - 1):
            rows_j = torch.arange((j * world + rank) * args.batch, (j * world + rank + 1) * args.batch, device=dev)
            q_ring.append(q_more.select(rows_j).contiguous())
        del q_more
    ring_pos = [0]

    def next_batch():
        b = q_ring[ring_pos[0] % ring]
        ring_pos[0] += 1
        return b

    # ---- recall@k vs exact inner product (outside the timed region; unsharded index)
    recall = None
    recall_hard, hard_ctx = None, None
    recall_ctx = None
    if rank == 0 and args.recall_queries > 0:
        nr = min(args.recall_queries, q.n)
        qs = q.select(torch.arange(nr, device=dev))
        r = sl._search_batch(qs, charge, 'open', want_knn=True, device_out=True)
        vec = sl._encode(part.spectra)
        flat = faiss.IndexFlatIP(cfg.hash_len)
        flat.add(vec)
        del vec
        _, Ie = flat.search(sl._encode(qs), args.k)
        # the same on HARD queries (synthetic.make_queries(hard=HARD_DEFAULT): calibrated so that exact
        # search finds the source of ~75 % of the modified queries, the reference's iPRG2012 figure)
        hard_ctx = None
        if not args.no_recall_hard:
            qh, truth_h = synthetic.make_queries(lib, aux, 2 * args.batch, seed=43, open_range=args.open_da,
                                                 charge=charge, hard=synthetic.HARD_DEFAULT)
            qhs = qh.select(torch.arange(nr, device=dev))
            _, Ie_h = flat.search(sl._encode(qhs), args.k)
            hard_ctx = (qh, qhs, Ie_h, truth_h['source_row'][:nr], truth_h['is_modified'][:nr])
        del flat
        knn = r.knn

        def overlap(A):         # sum over queries of |A_i & Exact_i| (sets, on the device)
            return sum(int(torch.isin(A[i][A[i] >= 0], Ie[i]).sum()) for i in range(nr))
        src = src_local[:nr]
        hit = (knn == src.unsqueeze(1)).any(1)
        mod = mod_local[:nr]
        top1 = (r.best_row.to(torch.int64) == src)
        rec = overlap(knn) / float(nr * args.k)
        ehit = (Ie == src.unsqueeze(1)).any(1)        # the notebook's metric, on EXACT inner-product search
        recall = {'queries': nr, 'k': args.k,
                  'recall_at_k_vs_exact_ip': rec,
                  'hit_at_k_source_spectrum': float(hit.float().mean()),
                  'hit_at_k_modified_only': float(hit[mod].float().mean()) if mod.any() else None,
                  'top1_is_source_spectrum': float(top1.float().mean()),
                  # the reference's one behavioural number at this boundary: with EXACT inner-product
                  # search the true match is inside the top 1024 for 75.1 % of the modified SSMs of
                  # iPRG2012 (notebooks/iprg2012_num_candidates.ipynb:282-288). The same quantity on
                  # this synthetic library (source spectrum of a query inside the exact top-k):
                  'exact_hit_at_k_source': float(ehit.float().mean()),
                  'exact_hit_at_k_modified_only': float(ehit[mod].float().mean()) if mod.any() else None,
                  'reference_exact_hit_at_1024_modified_iprg2012': 0.751,
                  'generator_note': 'synthetic queries are one library spectrum each with 10 % of the peaks '
                                    'dropped, 10 noise peaks, 5 mDa fragment jitter and (half of them) one '
                                    'mass shift on the ions that contain the modified residue (SURVEY.md 8d): '
                                    'cleaner than real spectra of modified peptides, so the synthetic hit@k '
                                    'is an upper bound on what the iPRG2012 figure measures, not a reproduction'}
        if args.index == 'ivfpq':
            # SURVEY.md 8(d) operating point: IVF-Flat over the SAME coarse quantiser and nprobe =
            # exact scores inside the probed lists, the ceiling of this index geometry
            fl = faiss.IndexIVFFlat(faiss.IndexFlatIP(cfg.hash_len), cfg.hash_len, args.nlist)
            fl.set_trained(idx.centroids())
            vec = sl._encode(part.spectra)
            fl.add(vec)
            del vec
            fl.nprobe = args.nprobe
            _, If = fl.search(sl._encode(qs), args.k)
            If_h = fl.search(sl._encode(hard_ctx[1]), args.k)[1] if hard_ctx else None
            del fl
            rf = overlap(If) / float(nr * args.k)
            recall.update({'ivfflat_same_geometry_recall_at_k': rf,
                           'ratio_to_ivfflat': rec / rf if rf > 0 else None,
                           'ivfflat_hit_at_k_source_spectrum':
                               float((If == src.unsqueeze(1)).any(1).float().mean()),
                           'fixed_recall_criterion': 'recall@k >= 0.95 x IVF-Flat(nlist, nprobe) '
                                                     '(SURVEY.md 8d)',
                           'meets_criterion': bool(rf > 0 and rec / rf >= 0.95)})
        recall_ctx = (qs, Ie, nr) if args.index == 'ivfpq' else None
        if hard_ctx:
            qh, qhs, Ie_h, src_h, mod_h = hard_ctx
            rh = sl._search_batch(qhs, charge, 'open', want_knn=True, device_out=True)

            def ov_h(A):
                return sum(int(torch.isin(A[i][A[i] >= 0], Ie_h[i]).sum()) for i in range(nr)) / float(nr * args.k)

            def hits(A):
                h_ = (A == src_h.unsqueeze(1)).any(1)
                return {'hit_at_k_source_spectrum': float(h_.float().mean()),
                        'hit_at_k_modified_only': float(h_[mod_h].float().mean()) if mod_h.any() else None}
            recall_hard = {
                'data': f'synthetic.make_queries(hard={synthetic.HARD_DEFAULT}): ' + json.dumps(synthetic.hard_levers(synthetic.HARD_DEFAULT)),
                'calibration': 'exact inner-product hit@1024 of MODIFIED queries tuned to the reference\'s 75.1 % on '
                               'iPRG2012 (notebooks/iprg2012_num_candidates.ipynb:282-288); scripts/tune_hard.py, '
                               'profiles/r05_hard_calibration.txt',
                'queries': nr, 'k': args.k,
                'exact': dict(hits(Ie_h), reference_exact_hit_at_1024_modified_iprg2012=0.751),
                args.index: dict(hits(rh.knn), recall_at_k_vs_exact_ip=ov_h(rh.knn), nprobe=args.nprobe,
                                 top1_is_source_spectrum=float((rh.best_row.to(torch.int64) == src_h).float().mean())),
            }
            if args.index == 'ivfpq' and If_h is not None:
                rfh = ov_h(If_h)
                recall_hard['ivfflat_same_geometry'] = dict(hits(If_h), recall_at_k_vs_exact_ip=rfh, nprobe=args.nprobe)
                recall_hard[args.index]['ratio_to_ivfflat'] = recall_hard[args.index]['recall_at_k_vs_exact_ip'] / rfh if rfh > 0 else None
            del rh
        torch.cuda.empty_cache()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """barrier, `steps` calls, barrier; MAX over ranks of the elapsed seconds"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)          # control plane: gloo, host tensor
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    def unsharded_step():
        return sl._search_batch(next_batch(), charge, 'open', device_out=True)

    # ---- shard for N > 1
    shard_check, alt = None, None
    degree = 1
    if world > 1:
        degree = args.shard_degree or world
        ns = min(256, q.n)
        ref = sl._search_batch(q.select(torch.arange(ns, device=dev)), charge, 'open',
                               device_out=True)          # unsharded result of my first queries
        if degree > 1:
            # the other layout the library supports, for the record (not `value`): every
            # rank a full replica serving its own slice, no collective on the data path --
            # pipelined as a replica would run (the sharded step below is not)
            sl.set_pipeline(not args.no_pipeline)
            for _ in range(2):
                unsharded_step()
            sl.synchronize()
            n_alt = max(3, min(args.steps, 10))
            el, _ = timed(unsharded_step, n_alt)
            sl.synchronize()
            sl.set_pipeline(False)
            ring_pos[0] = 0
            alt = {'replicas_only': {'value': round(world * args.batch * n_alt / el, 2),
                                     'ms_per_step': round(el / n_alt * 1e3, 3),
                                     'steps': n_alt, 'pipelined': not args.no_pipeline}}
            # insurance for the first multi-GPU lease (no round ever had one): from here on a complete
            # line exists. If the list-sharded path then raises or stops inside a collective, rank 0
            # prints THIS line -- the replicas layout, marked as such -- instead of losing the run
            # (`sharded_path_failed` says where and why); see _arm_fallback.
            _arm_fallback(rank, args.sharded_watchdog_seconds, {
                'metric': 'query spectra/sec + recall@k vs brute-force, open-mod search on MassIVE-KB',
                'value': alt['replicas_only']['value'], 'unit': 'query spectra/s',
                'n_gpus': world, 'steps': n_alt, 'warmup': 2, 'ms_per_step': alt['replicas_only']['ms_per_step'],
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'batch_per_step': args.batch,
                'config': {'workload': f'configs[2] on {world} REPLICAS (every rank the whole {args.library_size}-spectrum '
                                       f'library, {args.index} nlist={args.nlist} nprobe={args.nprobe} k={args.k}, its own '
                                       f'{args.batch}-query slice per step, no data-path collective): the FALLBACK line -- '
                                       f'the list-sharded path of configs[3] did not finish',
                           'library_size': args.library_size, 'batch_per_gpu': args.batch,
                           'global_batch': world * args.batch, 'index': args.index, 'nlist': args.nlist,
                           'nprobe': args.nprobe, 'k': args.k, 'parallelism': f'replicas x{world}'},
                'roofline': None, 'cpu_baseline': None})
        if degree > 1:
            world_group = data_plane()          # RCCL comes up here, behind the fallback line
            _FALLBACK['stage'] = 'sharding the index'
            group, shard_rank, _ = make_shard_groups(degree, backend='nccl' if backend == 'nccl' else None,
                                                     world_group=world_group)
            idx.shard(shard_rank, degree)
            shard_backend = HipShardBackend(sl, charge, 'open')

            # the row width of the peak all-gather must be the same number on every rank: agreed
            # once, outside the timed loop (every step then runs without that all-reduce)
            wmax = torch.tensor([max(int(q.max_peaks()), int(cfg.max_peaks_used))], dtype=torch.int64)
            dist.all_reduce(wmax, op=dist.ReduceOp.MAX)          # control plane
            peak_width = int(wmax.item())

            comm_log, xstats = CommLog(), {}

            def step():
                return sharded_search_batch(shard_backend, next_batch(), group=group, device_out=True,
                                            peak_width=peak_width, comm=comm_log, stats=xstats,
                                            two_phase=None if args.exchange == 'two-phase' else False,
                                            head_keys=args.head_keys, extras_per_query=args.extras_per_query,
                                            shard_keys=args.shard_keys)
        else:
            step = unsharded_step
        ring_pos[0] = 0
        _FALLBACK['stage'] = 'first sharded batch (parity check against the unsharded result)'
        if degree > 1 and os.environ.get('ASL_BENCH_INJECT') == 'raise':      # tests/test_gpu_distributed.py
            raise RuntimeError('injected failure of the sharded path')
        if degree > 1 and os.environ.get('ASL_BENCH_INJECT') == 'hang':
            time.sleep(1e6)
        got = step()                     # (ring element 0 = q)
        same = bool(torch.equal(got.best_row[:ns], ref.best_row) and
                    torch.equal(got.best_score[:ns], ref.best_score))
        flag = torch.tensor([int(same)])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)              # control plane
        shard_check = {'queries_per_rank': ns, 'sharded_equals_unsharded': bool(flag.item())}
    else:
        step = unsharded_step

    comm = None
    L = _lib.lib()
    # consecutive steps are independent batches: the encoder + coarse quantiser of step i+1 run
    # on one stream under the list scan of step i on another (bit-identical results)
    pipelined = not args.no_pipeline and degree == 1
    sl.set_pipeline(pipelined)
    _FALLBACK['stage'] = 'warm-up steps'
    for _ in range(args.warmup):
        step()
    sl.synchronize()
    ring_pos[0] = 0
    if world > 1 and degree > 1:
        comm_log.calls.clear()
        xstats.clear()
    # inside the timed region only the dominant kernel -- the list scan -- is bracketed by HIP
    # events on the stream it is launched on (two per step: `roofline.achieved`); the other stages
    # are timed in a second pass of the same steps right after it (events around every stage keep
    # a pipelined step's kernels from being dispatched back to back: 0.3 ms of 7.9 in round 4)
    L.asl_profile_enable(2)
    L.asl_profile_reset()
    _FALLBACK['stage'] = 'timed steps'
    elapsed, res = timed(step, args.steps)
    _FALLBACK['stage'] = 'after the timed steps (collectives alone, second pass, legs)'
    sl.synchronize()            # reports any error a pipelined batch deferred
    if world > 1 and degree > 1:
        # bytes every collective of a step moved (counted inside the timed steps) and what each
        # costs on its own at its real per-chunk size (outside the timed region, after it)
        comm = comm_log.summary(args.steps)
        comm['exchange'] = xstats.get('exchange_used') or ('two-phase' if xstats.get('two_phase') else 'full rows')
        comm['head_width'] = xstats.get('head_width')
        comm['shard_k'] = xstats.get('shard_k')
        comm['fallbacks_to_full_exchange'] = xstats.get('fallback', 0)
        # (query, owner) rows the shards scanned a second time with the full k, per step, summed
        # over the ranks of the job (the "third phase" of VERDICT r4, done on the shard inside phase 2)
        t3 = torch.tensor([xstats.get('third_phase_queries', 0)], dtype=torch.int64)
        dist.all_reduce(t3, op=dist.ReduceOp.SUM)               # control plane
        comm['third_phase_queries'] = round(int(t3[0]) / max(args.steps, 1), 2)
        comm['third_phase_share_of_rows'] = round(int(t3[0]) / max(args.steps, 1) / (degree * degree * args.batch), 6)
        comm['collective_ms_alone'] = time_collectives(comm_log, group, degree, dev, backend, args.steps)
    L.asl_profile_enable(0)
    scan_timed = {}
    ms, n = C.c_double(), C.c_int64()
    L.asl_profile_get(b'scan', C.byref(ms), C.byref(n))
    scan_timed = {'ms_total': ms.value, 'launches': n.value}
    # second pass over the same batches, every stage bracketed and the scanned vectors counted on
    # the device (not part of `value`; the count is a property of the batch, the same in both passes)
    L.asl_profile_reset()
    L.asl_profile_enable(1)
    ring_pos[0] = 0                      # the same batches as the timed pass
    timed(step, args.steps)
    sl.synchronize()
    L.asl_profile_enable(0)
    scanned = L.asl_profile_scanned_vectors()
    # the same step at the reference's own default --batch_size (config.py:195-198: 16 384): the
    # first 16 384 queries of every batch of the ring, twice the steps (N = 1 only; not `value`)
    ref_batch = None
    REF_BATCH = 16384
    if world == 1 and pipelined and args.batch > REF_BATCH:
        halves = [b.select(torch.arange(REF_BATCH, device=dev)).contiguous() for b in q_ring]
        pos = [0]

        def half_step():
            b = halves[pos[0] % len(halves)]
            pos[0] += 1
            return sl._search_batch(b, charge, 'open', device_out=True)
        for _ in range(max(args.warmup, 2)):
            half_step()
        sl.synchronize()
        n_half = 2 * args.steps
        el_half, _ = timed(half_step, n_half)
        sl.synchronize()
        ref_batch = {'batch': REF_BATCH, 'value': round(REF_BATCH * n_half / el_half, 2), 'unit': 'query spectra/s',
                     'ms_per_step': round(el_half / n_half * 1e3, 3), 'steps': n_half,
                     'note': 'the reference\'s default --batch_size (src/ann_solo/config.py:195-198); a 16 384-query '
                             'launch is 21.3 rounds of the chip\'s 768 resident workgroups: the partial last round and '
                             'the spread of the workgroups\' durations cost ~3 % of the scan, the rescoring launches\' '
                             'tails as much again (profiles/r05_batch_size.txt)'}
        del halves
    sl.set_pipeline(False)
    # batch 0 once more, synchronously: the result the parity / post-path legs below refer to
    q_keep, q_ring[:] = q_ring[:], [q]
    ring_pos[0] = 0
    res = step()
    sl.synchronize()
    q_ring[:] = q_keep

    stages = {}
    for name in ('encode', 'coarse_gemm', 'coarse_select', 'scan', 'refine', 'filter', 'rescore',
                 'rescore_matches'):
        ms, n = C.c_double(), C.c_int64()
        L.asl_profile_get(name.encode(), C.byref(ms), C.byref(n))
        stages[name] = {'ms_total': ms.value, 'launches': n.value}
    stages_pass_scan = dict(stages['scan'])
    if scan_timed['launches'] > 0:
        stages['scan'] = scan_timed          # the roofline's figure: the timed region's own
    # the dominant kernel on its own (outside the timed region): in the pipelined step it shares
    # the chip with the rescoring of the previous batch, which `roofline` (timed region) includes
    scan_alone_ms = None
    if pipelined and world == 1:
        L.asl_profile_reset()
        L.asl_profile_enable(1)
        ring_pos[0] = 0
        for _ in range(ring if args.steps % ring == 0 else 3):       # the same mix of batches as the timed pass
            step()
        sl.synchronize()
        L.asl_profile_enable(0)
        ms_a, n_a = C.c_double(), C.c_int64()
        L.asl_profile_get(b'scan', C.byref(ms_a), C.byref(n_a))
        if n_a.value > 0:
            scan_alone_ms = ms_a.value / n_a.value

    # ---- the same metric at FIXED RECALL: IVF-Flat over the same library and the
    # same coarse quantiser, at the smallest nprobe (steps of 8) whose recall@k is still >= 0.95 x
    # IVF-Flat(nlist, nprobe) -- the whole hot path timed like `value`, after it, in the same run
    fixed_recall = None
    if (world == 1 and rank == 0 and args.index == 'ivfpq' and recall_ctx is not None
            and not args.no_fixed_recall and not args.refine_k):
        from dataclasses import replace
        qs, Ie, nr = recall_ctx
        sl_f = SpectralLibrary(lib, config=replace(cfg, index='ivfflat', refine_k=None), device=dev)
        idx_f = sl_f._get_ann_index(charge)
        same_q = bool(np.array_equal(idx_f.centroids(), idx.centroids()))
        vq = sl_f._encode(qs)

        def recall_at(nprobe):
            idx_f.nprobe = nprobe
            _, If = idx_f.search(vq, args.k)
            return sum(int(torch.isin(If[i][If[i] >= 0], Ie[i]).sum()) for i in range(nr)) / float(nr * args.k)
        ref_rec = recall_at(args.nprobe)
        best, best_rec = args.nprobe, ref_rec
        for nprobe in range(args.nprobe - 8, args.nprobe // 2, -8):
            r_ = recall_at(nprobe)
            if r_ < 0.95 * ref_rec:
                break
            best, best_rec = nprobe, r_
        sl_f._num_probe = best

        def flat_step():
            return sl_f._search_batch(next_batch(), charge, 'open', device_out=True)
        sl_f.set_pipeline(pipelined)
        for _ in range(args.warmup):
            flat_step()
        sl_f.synchronize()
        ring_pos[0] = 0
        L.asl_profile_enable(2)
        L.asl_profile_reset()
        el_f, _ = timed(flat_step, args.steps)
        sl_f.synchronize()
        L.asl_profile_enable(0)
        sl_f.set_pipeline(False)
        ms_f, n_f = C.c_double(), C.c_int64()
        L.asl_profile_get(b'scan', C.byref(ms_f), C.byref(n_f))
        res_f = sl_f._search_batch(q, charge, 'open', device_out=True)     # batch 0: what the parity leg checks
        sl_f.synchronize()
        alone_f = None
        if pipelined:           # the scan kernel on its own, as for the headline kernel below
            L.asl_profile_reset()
            L.asl_profile_enable(1)
            ring_pos[0] = 0
            for _ in range(ring if args.steps % ring == 0 else 3):
                flat_step()
            sl_f.synchronize()
            L.asl_profile_enable(0)
            ms_a, n_a = C.c_double(), C.c_int64()
            L.asl_profile_get(b'scan', C.byref(ms_a), C.byref(n_a))
            if n_a.value > 0:
                alone_f = ms_a.value / n_a.value
        fixed_recall = {'index': 'ivfflat', 'nlist': args.nlist, 'nprobe': best,
                        # the reference's storage: float32 components as given (spectral_library.py:174-181):
                        # ids / scores below are those of an index over the UNQUANTISED hashed vectors
                        'storage': idx_f.storage,
                        'same_coarse_quantiser_as_the_ivfpq_index': same_q,
                        'recall_at_k_vs_exact_ip': best_rec,
                        'criterion': f'>= 0.95 x {ref_rec:.4f} (IVF-Flat, nprobe {args.nprobe})',
                        'value': round(args.batch * args.steps / el_f, 2), 'unit': 'query spectra/s',
                        'ms_per_step': round(el_f / args.steps * 1e3, 3), 'steps': args.steps,
                        'scan_ms_per_step': round(ms_f.value / max(args.steps, 1), 3),
                        'pipelined': pipelined,
                        'roofline': postings_roofline(sl_f, idx_f, q, best, ms_f.value / max(n_f.value, 1), args)}
        if alone_f:
            rf_ = fixed_recall['roofline']
            avg_f = ms_f.value / max(n_f.value, 1)
            rf_['kernel_alone'] = {'avg_launch_ms': round(alone_f, 4),
                                   'frac': round(rf_['frac'] * avg_f / alone_f, 5),
                                   'frac_by_lines': round(rf_['frac_by_lines'] * avg_f / alone_f, 5),
                                   'note': '3 launches after the timed region, stages strictly one after the other'}
        if args.cpu_seconds > 0:
            # the SAME index (this leg's 25-iteration quantiser, nprobe = best) against the oracle:
            # neighbour id sets, winners and scores of a sample of the batch
            from argparse import Namespace
            a_f = Namespace(**{**vars(args), 'nprobe': best, 'cpu_seconds': min(args.cpu_seconds, 8.0)})
            cb = cpu_baseline(a_f, sl_f, sl_f.partitions[charge], idx_f, q, res_f, charge, cfg, faiss_leg=False)
            fixed_recall['parity_vs_gpu'] = cb['parity_vs_gpu']
            fixed_recall['cpu_baseline'] = {k_: cb[k_] for k_ in ('value', 'unit', 'cores', 'kind', 'sample',
                                                                     'single_core_value', 'parallel_efficiency',
                                                                     'dense_definition_check')}
        if hard_ctx and recall_hard is not None:
            # the same question on the HARD queries: the smallest nprobe (steps of 8) that keeps
            # recall@k >= 0.95 x IVF-Flat(nprobe) there, and the whole hot path timed at that point
            qh, qhs, Ie_h, src_h, mod_h = hard_ctx
            vqh = sl_f._encode(qhs)

            def recall_h(nprobe):
                idx_f.nprobe = nprobe
                _, A = idx_f.search(vqh, args.k)
                return sum(int(torch.isin(A[i][A[i] >= 0], Ie_h[i]).sum()) for i in range(nr)) / float(nr * args.k), A
            ref_h, _ = recall_h(args.nprobe)
            best_h, best_rec_h, A_best = args.nprobe, ref_h, None
            for nprobe in range(args.nprobe - 8, args.nprobe // 2, -8):
                r_, A_ = recall_h(nprobe)
                if r_ < 0.95 * ref_h:
                    break
                best_h, best_rec_h, A_best = nprobe, r_, A_
            if A_best is None:
                _, A_best = recall_h(best_h)
            hb = [qh.select(torch.arange(j * args.batch, (j + 1) * args.batch, device=dev)).contiguous() for j in range(2)]
            hp = [0]
            sl_f._num_probe = best_h

            def hard_step():
                b_ = hb[hp[0] % 2]
                hp[0] += 1
                return sl_f._search_batch(b_, charge, 'open', device_out=True)
            sl_f.set_pipeline(pipelined)
            for _ in range(args.warmup):
                hard_step()
            sl_f.synchronize()
            hp[0] = 0
            n_h = max(2, min(args.steps, 10))
            el_h, _ = timed(hard_step, n_h)
            sl_f.synchronize()
            sl_f.set_pipeline(False)
            hb_hit = (A_best == src_h.unsqueeze(1)).any(1)
            recall_hard['fixed_recall'] = {
                'index': 'ivfflat', 'storage': idx_f.storage, 'nlist': args.nlist, 'nprobe': best_h,
                'recall_at_k_vs_exact_ip': best_rec_h,
                'criterion': f'>= 0.95 x {ref_h:.4f} (IVF-Flat, nprobe {args.nprobe}, hard queries)',
                'hit_at_k_source_spectrum': float(hb_hit.float().mean()),
                'hit_at_k_modified_only': float(hb_hit[mod_h].float().mean()) if mod_h.any() else None,
                'value': round(args.batch * n_h / el_h, 2), 'unit': 'query spectra/s',
                'ms_per_step': round(el_h / n_h * 1e3, 3), 'steps': n_h, 'distinct_batches': 2}
            del vqh, hb
        sl_f.shutdown()
        del sl_f, idx_f
        torch.cuda.empty_cache()
        # the opt-in fixed-point storage at the same operating point, for the record (never `value`)
        sl_x = SpectralLibrary(lib, config=replace(cfg, index='ivfflat', refine_k=None, flat_storage='fx22'),
                               device=dev)
        idx_x = sl_x._get_ann_index(charge)
        sl_x._num_probe = best

        def fx_step():
            return sl_x._search_batch(next_batch(), charge, 'open', device_out=True)
        sl_x.set_pipeline(pipelined)
        for _ in range(args.warmup):
            fx_step()
        sl_x.synchronize()
        ring_pos[0] = 0
        L.asl_profile_enable(2)
        L.asl_profile_reset()
        n_x = max(2, min(args.steps, 10))
        el_x, _ = timed(fx_step, n_x)
        sl_x.synchronize()
        L.asl_profile_enable(0)
        sl_x.set_pipeline(False)
        res_x = sl_x._search_batch(q, charge, 'open', device_out=True)
        sl_x.synchronize()
        ms_x, c_x = C.c_double(), C.c_int64()
        L.asl_profile_get(b'scan', C.byref(ms_x), C.byref(c_x))
        fixed_recall['alt_storage'] = {
            'storage': idx_x.storage, 'value': round(args.batch * n_x / el_x, 2),
            'ms_per_step': round(el_x / n_x * 1e3, 3), 'steps': n_x,
            'scan_ms_per_step': round(ms_x.value / max(c_x.value, 1), 3),
            'winners_equal_the_fp32_index': int((res_x.best_row == res_f.best_row).sum()),
            'queries': int(res_x.best_row.numel()),
            'note': "opt-in: components rounded to 2^-22 (|dx| <= 1.2e-7), 4-byte postings; ids differ from the "
                    "float32 index only within 1e-6 of the k-th score (tests/test_gpu_fullscale.py bounds it "
                    "at this size)"}
        sl_x.shutdown()
        del sl_x, idx_x
        torch.cuda.empty_cache()


    # ---- the reference's own default geometry (config.py:202-211: num_list 256, num_probe 128, IVF-Flat,
    # k 1024): configs[1] (an iPRG2012-sized library, ~9 k spectra) and the 2.1 M library
    ref_geometry = None
    if world == 1 and rank == 0 and not args.no_reference_geometry and args.index == 'ivfpq' and not args.refine_k:
        from argparse import Namespace
        from dataclasses import replace
        ref_geometry = {}

        def leg(name, lib_x, aux_x, open_da, steps, cpu_s, what):
            cfg_x = replace(cfg, index='ivfflat', num_list=256, num_probe=128, num_candidates=1024, refine_k=None,
                            precursor_tolerance_mass_open=open_da)
            t0 = time.time()
            sl_x = SpectralLibrary(lib_x, config=cfg_x, device=dev)
            idx_x = sl_x._get_ann_index(charge)
            t_build = time.time() - t0
            nb = max(2, min(ring, 4))
            qx, _ = synthetic.make_queries(lib_x, aux_x, nb * args.batch, seed=42, open_range=open_da, charge=charge)
            bx = [qx.select(torch.arange(j * args.batch, (j + 1) * args.batch, device=dev)).contiguous() for j in range(nb)]
            pos = [0]

            def step_x():
                b_ = bx[pos[0] % nb]
                pos[0] += 1
                return sl_x._search_batch(b_, charge, 'open', device_out=True)
            sl_x.set_pipeline(pipelined)
            for _ in range(2):
                step_x()
            sl_x.synchronize()
            pos[0] = 0
            L.asl_profile_enable(2)
            L.asl_profile_reset()
            el_x, _ = timed(step_x, steps)
            sl_x.synchronize()
            L.asl_profile_enable(0)
            sl_x.set_pipeline(False)
            ms_x, c_x = C.c_double(), C.c_int64()
            L.asl_profile_get(b'scan', C.byref(ms_x), C.byref(c_x))
            res_x = sl_x._search_batch(bx[0], charge, 'open', device_out=True)
            sl_x.synchronize()
            out_x = {'workload': what, 'library_size': int(lib_x.n), 'index': 'ivfflat', 'storage': idx_x.storage,
                     'nlist': 256, 'nprobe': 128, 'k': 1024, 'open_da': open_da, 'batch': args.batch,
                     'value': round(args.batch * steps / el_x, 2), 'unit': 'query spectra/s',
                     'ms_per_step': round(el_x / steps * 1e3, 3), 'steps': steps, 'distinct_batches': nb,
                     'scan_ms_per_step': round(ms_x.value / max(c_x.value, 1), 3),
                     'index_build_s': round(t_build, 2),
                     'identified': int((res_x.best_row >= 0).sum())}
            if cpu_s > 0:
                a_x = Namespace(**{**vars(args), 'nprobe': 128, 'k': 1024, 'nlist': 256, 'open_da': open_da,
                                   'cpu_seconds': cpu_s})
                cb = cpu_baseline(a_x, sl_x, sl_x.partitions[charge], idx_x, bx[0], res_x, charge, cfg_x, faiss_leg=False)
                out_x['parity_vs_gpu'] = cb['parity_vs_gpu']
                out_x['cpu_baseline'] = {k_: cb[k_] for k_ in ('value', 'unit', 'cores', 'kind', 'sample',
                                                               'single_core_value', 'parallel_efficiency')}
            sl_x.shutdown()
            del sl_x, idx_x, qx, bx
            torch.cuda.empty_cache()
            ref_geometry[name] = out_x
        lib9, aux9 = synthetic.make_library(9000, seed=20120701, device=dev, charges=(charge,), charge_p=(1.0,))
        leg('configs_1_iprg2012_sized', lib9, aux9, 300.0, max(4, min(args.steps, 20)), min(args.cpu_seconds, 4.0),
            'configs[1]: an iPRG2012-sized synthetic library (9 000 spectra), IVF-Flat at the reference defaults '
            '(nlist 256, nprobe 128, k 1024), open +-300 Da (the notebooks\' window), shifted dot')
        del lib9, aux9
        leg('massivekb_sized_reference_defaults', lib, aux, args.open_da, max(2, min(args.steps, 5)),
            min(args.cpu_seconds, 6.0),
            f'the {args.library_size}-spectrum library at the reference defaults (IVF-Flat nlist 256, nprobe 128: a '
            f'query scans half of the library), open +-{args.open_da:g} Da, shifted dot')

    if rank == 0:
        total_queries = world * args.batch * args.steps
        scan = stages['scan']
        roofline = None
        if scan['launches'] > 0 and scan['ms_total'] > 0:
            avg_ms = scan['ms_total'] / scan['launches']
            traffic, traffic_src = pmc_traffic(args, world)
            if args.index == 'ivfpq':
                per_vec = BYTES_PER_SCANNED_VECTOR
                bytes_per_launch = scanned / scan['launches'] * per_vec
                achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
                # which memory level serves the stream: the tiled codes + ids of this library (every probed
                # tile is read by ~ batch x nprobe / nlist queries of the launch)
                info_ = idx.info()
                resident = int(info_.ntotal) * (BYTES_PER_SCANNED_VECTOR + 1)      # + ~3 % tile padding
                cached = resident < LLC_BYTES
                roofline = {'bound': 'hbm', 'kernel': 'pq_scan_v3_kernel',
                            'served_by': (f'Infinity Cache (L3, 256 MiB): the {resident / 1e6:.0f} MB of codes + ids of '
                                          f'this library stay resident in it, so this stream runs at the XCD<->fabric '
                                          f'ceiling, ABOVE what DRAM delivers ({HBM_COPY_GBS / 1e3:.2f} TB/s copy rate); '
                                          f'`frac` is still quoted against the {HBM_PEAK_GBS / 1e3:.0f} TB/s HBM peak as the '
                                          f'contract asks -- the HBM-served figure is roofline.beyond_llc') if cached else
                                         f'HBM ({resident / 1e6:.0f} MB of codes + ids exceed the 256 MiB Infinity Cache)',
                            'resident_mb': round(resident / 1e6, 1),
                            'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': round(achieved / HBM_PEAK_GBS, 5),
                            'frac_of_measured_hbm_copy': round(achieved / HBM_COPY_GBS, 5),
                            'bytes_per_vector': per_vec,
                            'traffic': traffic, 'traffic_source': traffic_src,
                            'avg_launch_ms': round(avg_ms, 4),
                            'algorithmic_bytes_per_launch': int(bytes_per_launch)}
            else:
                roofline = postings_roofline(sl, idx, q, args.nprobe, avg_ms, args if world == 1 else None)
            roofline['vectors_scanned_per_query'] = round(scanned / args.steps / (degree * args.batch), 1)
            if scan_alone_ms:
                roofline['kernel_alone'] = {
                    'avg_launch_ms': round(scan_alone_ms, 4),
                    'frac': round(roofline['frac'] * avg_ms / scan_alone_ms, 5),
                    'note': '3 launches after the timed region with the stages of consecutive batches '
                            'strictly one after the other (no rescoring kernel beside the scan)'}
            if args.index == 'ivfpq':
                # the ids are read for survivors only: SURVEY.md 8(d)'s "ids implicit" variant
                codes = achieved * BYTES_PER_CODE / BYTES_PER_SCANNED_VECTOR
                roofline['achieved_codes_only'] = round(codes, 2)
                roofline['frac_codes_only'] = round(codes / HBM_PEAK_GBS, 5)
                if traffic:
                    roofline['achieved_measured_traffic'] = round(traffic / (avg_ms * 1e-3) / 1e9, 2)
                    roofline['frac_measured_traffic'] = round(
                        traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if (roofline is not None and world == 1 and args.index == 'ivfpq' and args.beyond_llc_chunks > 1
                and not args.refine_k and args.scan_variant == 0):
            roofline['beyond_llc'] = beyond_llc_leg(args, sl, part, idx, q, dev)
        # post-path step of the same batch, outside the timed region: the 33 SSM similarity
        # features of every best match (utils._compute_ssm_features), one kernel launch
        from ann_solo_amd.spectrum_similarity import ssm_features
        feats = ssm_features(q, part.spectra, res.best_row, res.pm_pairs, res.pm_count)      # (first call: code load, buffers)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        feats = ssm_features(q, part.spectra, res.best_row, res.pm_pairs, res.pm_count)
        torch.cuda.synchronize()
        feat_ms = (time.perf_counter() - t1) * 1e3
        n_ssm = int((res.pm_count > 0).sum())
        del feats
        cpu, ctx = None, None
        if world == 1 and args.cpu_seconds > 0:
            ctx = oracle_context(sl, part, idx)
            cpu = cpu_baseline(args, sl, part, idx, q, res, charge, cfg, ctx=ctx)
        stages_gbs = stage_rates(args, sl, part, q, charge, stages, lib) if world == 1 else None
        # configs[4] in the same run: one pass of the two-level cascade over
        # cascade_batches x batch queries through the same engine and index, with oracle parity
        cascade = None
        if world == 1 and not args.no_cascade:
            from argparse import Namespace
            a_c = Namespace(**{**vars(args), 'steps': max(1, min(args.steps, 10)), 'warmup': max(1, min(args.warmup, 2))})
            c = cascade_pass(a_c, 1, 0, dev, backend, sl, lib, aux, charge, cfg,
                             parity_seconds=min(args.cpu_seconds, 8.0), oracle_ctx=ctx)
            cascade = {'workload': c['config']['workload'], 'value': c['value'], 'unit': c['unit'],
                       'queries_per_pass': c['identifications']['queries'], 'ms_per_pass': c['ms_per_step'],
                       'passes': c['steps'], 'levels': c['levels'],
                       'identifications': c['identifications'],
                       'parity_vs_oracle': c.get('parity_vs_oracle'),
                       'device_stages_ms_per_pass': c['device_stages_ms_per_step']}
        out = {
            'metric': 'query spectra/sec + recall@k vs brute-force, open-mod search on MassIVE-KB',
            'value': round(total_queries / elapsed, 2),
            'unit': 'query spectra/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            # the metric is "spectra/s + recall@k" and the north star asks for it "at fixed recall@k":
            # the companions of `value`, at the top level (details in `recall`, `fixed_recall`,
            # `at_reference_batch_size`)
            'batch_per_step': args.batch,
            'recall_at_k': None if recall is None else round(recall['recall_at_k_vs_exact_ip'], 5),
            'hit_at_k': None if recall is None else round(recall['hit_at_k_source_spectrum'], 5),
            'meets_fixed_recall_criterion': None if recall is None else recall.get('meets_criterion'),
            'value_at_fixed_recall': None if fixed_recall is None else fixed_recall['value'],
            'recall_at_k_at_fixed_recall': None if fixed_recall is None else fixed_recall.get('recall_at_k_vs_exact_ip'),
            'value_at_reference_batch': None if ref_batch is None else ref_batch['value'],
            'reference_batch': None if ref_batch is None else ref_batch['batch'],
            'config': {'workload': f'configs[2]: MassIVE-KB-scale synthetic library '
                                   f'({args.library_size} spectra, one charge-{charge} partition), '
                                   f'{args.index} m={args.pq_m} nlist={args.nlist} nprobe={args.nprobe} '
                                   f'k={args.k}{f" (exact re-rank of {args.refine_k})" if args.refine_k else ""}, open +-{args.open_da:g} Da, shifted dot, '
                                   f'fragment tol 0.02 Da',
                       'library_size': args.library_size, 'batch_per_gpu': args.batch,
                       'reference_default_batch_size': 16384,
                       'distinct_batches_in_the_timed_loop': ring,
                       'global_batch': world * args.batch, 'index': args.index,
                       'nlist': args.nlist, 'nprobe': args.nprobe, 'k': args.k,
                       'refine_k': args.refine_k or None,
                       'parallelism': 'single' if world == 1 else
                       (f'ivf-list-shard x{degree}' if degree == world else
                        f'replicas x{world}' if degree == 1 else
                        f'ivf-list-shard x{degree} in {world // degree} replica groups')},
            'pipeline': {'streams': 2 if pipelined else 1,
                         'note': 'stage times overlap across consecutive steps when true: their '
                                 'sum exceeds ms_per_step',
                         'stage_timing': 'scan: HIP events inside the timed region (the roofline figure); '
                                         'the other stages: a second pass of the same steps right after it '
                                         'with events around every stage (not part of value; its scan: '
                                         f"{stages_pass_scan['ms_total'] / max(stages_pass_scan['launches'], 1):.3f} ms)"},
            'recall': recall,
            'recall_hard': recall_hard,
            # which number answers "query spectra/sec at fixed recall@k" (north star / SURVEY 8d)
            'metric_note': (None if args.index != 'ivfpq' or recall is None else
                            'value is configs[2] as BASELINE.json names it (IVF-PQ m=32): its recall@k is '
                            f"{recall.get('ratio_to_ivfflat', 0) or 0:.2f} of IVF-Flat(nlist, nprobe), "
                            + ('which meets' if recall.get('meets_criterion') else 'which does NOT meet')
                            + ' the fixed-recall criterion (>= 0.95); the throughput AT FIXED RECALL is '
                            'fixed_recall.value (IVF-Flat over the same quantiser, measured in this run '
                            'with its own roofline and oracle parity). Data sets: `value`, `recall`, `fixed_recall`, '
                            '`cascade`, `reference_geometry` use the DEFAULT synthetic queries (clean: exact search '
                            'finds the source of 98 % of the modified ones); `recall_hard` (and its '
                            '`fixed_recall`) uses the HARD queries calibrated to the reference\'s 75 % '
                            '(iPRG2012, exact search, k = 1024)'),
            'fixed_recall': fixed_recall,
            'reference_geometry': ref_geometry,
            'cascade': cascade,
            'shard_check': shard_check,
            'comm': comm,
            'alt_layouts': alt,
            # `coarse_stage_incl_wait`: events around the coarse quantiser's launches on the pipeline's
            # LOW-priority stream -- mostly time spent queued behind the other stream's scan (the
            # kernels themselves: ~0.25 + 0.19 ms per 16 384 queries)
            'stages_ms_per_step': {('coarse_stage_incl_wait' if k == 'coarse_gemm' and pipelined else k):
                                   round(v['ms_total'] / args.steps, 3) for k, v in stages.items()},
            'stages_gbs': stages_gbs,
            'post_path': {'ssm_features_ms_per_batch': round(feat_ms, 3), 'ssms': n_ssm},
            'at_reference_batch_size': ref_batch,
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        _disarm_fallback()
        emit_line(json.dumps(out))
    _disarm_fallback()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class _LibraryMeta:
    """``library_meta[charge]`` for a synthetic library: row r on demand instead of millions of
    dictionaries (the writer reads identifier / peptide / precursor_mz / is_decoy)."""

    def __init__(self, ids, pmz):
        self.ids, self.pmz = ids, pmz

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, r):
        return {'identifier': int(self.ids[r]), 'peptide': f'SYNTH{int(self.ids[r])}K',
                'precursor_mz': float(self.pmz[r]), 'is_decoy': False}


def run_cascade(args, world, rank, dev, backend, sl, lib, aux, charge, cfg, data_group=None):
    out = cascade_pass(args, world, rank, dev, backend, sl, lib, aux, charge, cfg, data_group,
                       parity_seconds=min(args.cpu_seconds, 10.0) if world == 1 else 0.0)
    if out is not None:
        emit_line(json.dumps(out))


def cascade_pass(args, world, rank, dev, backend, sl, lib, aux, charge, cfg, data_group=None, parity_seconds=0.0,
                 oracle_ctx=None, index_name=None, nprobe=None):
    """BASELINE configs[4]: the reference's two-level cascade (spectral_library.py:237-259) --
    standard search (20 ppm window, no ANN) of every query, a gate standing for the mokapot FDR
    filter, then the open search (ANN + +-open Da window, shifted dot) of the unidentified
    remainder -- through ``SpectralLibrary.search`` (results as a columnar SSM table; the
    reference-shaped SSM records are materialised after the timed pass and reported). With N > 1 the
    library's IVF lists are sharded over the ranks (``enable_sharding``): level 1 is data-parallel
    over the queries, level 2 is the list-sharded search. One "step" = one pass over
    ``cascade_batches`` x ``batch`` queries per GPU. Returns the JSON object (rank 0; None elsewhere);
    ``parity_seconds`` > 0 adds ``parity_vs_oracle``: a sample of the pass's queries through the
    oracle's two levels (``cascade_parity``)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from ann_solo_amd import synthetic
    index_name = index_name or args.index
    nprobe = nprobe or args.nprobe
    part = sl.partitions[charge]
    sl.pipeline_cascade = not args.no_pipeline      # open-search batches through the two-stream pipeline
    nq = world * args.cascade_batches * args.batch
    q, truth = synthetic.make_queries(lib, aux, nq, seed=42, open_range=args.open_da, charge=charge)
    q = q.contiguous()
    pmz = q.precursor_mz.cpu().numpy()
    qmeta = {charge: [dict(identifier=f'scan={i}', index=i, retention_time=0.0,
                           precursor_charge=charge, precursor_mz=float(pmz[i])) for i in range(nq)]}
    lmeta = {charge: _LibraryMeta(part.ids, part.precursor_mz)}
    thr = args.accept_cosine

    def gate(table, mode):          # stands for utils.score_ssms (mokapot, out of scope)
        table.q[:] = np.where(table.score >= thr, 0.0, 1.0)
    gate.columnar = True            # works on the SSM table: no per-SSM Python objects

    def sub(n):
        rows = torch.arange(n, device=dev)
        return {charge: q.select(rows)}, {charge: qmeta[charge][:n]}

    def key(t):                     # identifications as arrays, ordered by query
        o = np.argsort(t.qrow, kind='stable')
        return t.qrow[o], t.lib_row[o], t.score[o], t.q[o]
    check = None
    if world > 1:                   # the same queries through one GPU, before the index is sharded
        ns = min(2 * args.batch // 8 + 37, nq)          # ragged against batch and world
        qs, qm = sub(ns)
        ref = key(sl.search(qs, qm, lmeta, score_ssms=gate))
        sl.enable_sharding(data_group)       # the data plane's group (RCCL; None = the default gloo group)
        got = key(sl.search(qs, qm, lmeta, score_ssms=gate))
        same = all(np.array_equal(a, b) for a, b in zip(ref, got))
        flag = torch.tensor([int(same)])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # control plane: gloo, host tensor
        check = {'queries': ns, 'identifications': int(len(ref[0])),
                 'sharded_cascade_equals_unsharded': bool(flag.item())}
    for _ in range(args.warmup):
        qs, qm = sub(min(nq, 2 * args.batch))
        sl.search(qs, qm, lmeta, score_ssms=gate)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    L = _lib_handle()
    L.asl_profile_enable(0)         # no stage events inside the timed passes (they cost ~4 % of a pass)
    sl.level_seconds = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = sl.search({charge: q}, qmeta, lmeta, score_ssms=gate)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # the device stages of ONE more pass, events around every stage (not part of `value`)
    level_seconds, sl.level_seconds = sl.level_seconds, {}
    L.asl_profile_reset()
    L.asl_profile_enable(1)
    sl.search({charge: q}, qmeta, lmeta, score_ssms=gate)
    barrier()
    L.asl_profile_enable(0)
    sl.level_seconds = level_seconds
    if rank != 0:
        return None
    stages = {}
    for name in ('encode', 'coarse_gemm', 'coarse_select', 'scan', 'filter', 'rescore',
                 'rescore_matches'):
        ms, n = C.c_double(), C.c_int64()
        L.asl_profile_get(name.encode(), C.byref(ms), C.byref(n))
        stages[name] = round(ms.value, 3)
    src = truth['source_row'].cpu().numpy()
    correct = int((ids.lib_row == src[ids.qrow]).sum())
    t0 = time.perf_counter()
    n_obj = len(ids.materialize())          # the reference-shaped SSM records, outside the timed pass
    t_obj = time.perf_counter() - t0
    lv = {}
    for mode, (sec, n_in, n_out) in sl.level_seconds.items():
        lv[mode] = {'queries_in_per_step': n_in // args.steps, 'ssms_out_per_step': n_out // args.steps,
                    'seconds_per_step': round(sec / args.steps, 4),
                    'spectra_per_s': round(n_in / sec, 1) if sec > 0 else None}
    out = {
        'metric': 'query spectra/sec + recall@k vs brute-force, open-mod search on MassIVE-KB',
        'value': round(nq * args.steps / el, 2), 'unit': 'query spectra/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(el / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'configs[4]: cascade over a MassIVE-KB-scale synthetic library '
                               f'({args.library_size} spectra): standard search 20 ppm -> gate '
                               f'(cosine >= {thr:g}, stands for the mokapot FDR filter) -> open search '
                               f'+-{args.open_da:g} Da of the remainder, {index_name} nlist={args.nlist} '
                               f'nprobe={nprobe} k={args.k}, shifted dot; results as a columnar SSM table',
                   'queries_per_gpu': args.cascade_batches * args.batch, 'batch': args.batch,
                   'parallelism': 'single' if world == 1 else
                   f'level 1 query-parallel x{world}, level 2 ivf-list-shard x{world}'},
        'levels': lv,
        'identifications': {'total': len(ids), 'source_spectrum_identified': correct,
                            'queries': nq,
                            'ssm_records_materialised_after_the_pass': n_obj,
                            'materialise_seconds': round(t_obj, 3)},
        'cascade_check': check,
        'device_stages_ms_per_step': stages,
    }
    if parity_seconds > 0 and world == 1:
        out['parity_vs_oracle'] = cascade_parity(args, sl, part, charge, cfg, q, ids, thr, parity_seconds,
                                                 oracle_ctx, nprobe)
    return out


def stage_rates(args, sl, part, q, charge, stages, lib):
    """GB/s of the two stages SURVEY.md 8(d) lists without a roofline claim. Encoder: the peaks
    read (8 B each) + the hashed vectors written (4 B x hash_len per query). Rescoring: per
    candidate that reaches it (it passed the precursor window) its packed peak record (9 B per
    peak: m/z, intensity, fragment charge) + the 32-byte row record; the candidates are counted
    exactly, outside the timed region: the neighbour ids of this batch through the same window
    test, whose count must equal the kernel's own ``n_candidates``."""
    import torch
    r = sl._search_batch(q, charge, 'open', want_knn=True, device_out=True)
    knn = r.knn
    ok = knn >= 0
    rows = knn.clamp(min=0)
    lp = torch.as_tensor(part.precursor_mz, device=knn.device).to(torch.float32).to(torch.float64)
    qp = q.precursor_mz.to(knn.device).to(torch.float64).unsqueeze(1)
    ok &= (qp - lp[rows]).abs() * charge <= float(args.open_da)
    valid = torch.as_tensor(part.valid, device=knn.device) if getattr(part, 'valid', None) is not None else None
    if valid is not None:
        ok &= valid[rows].bool()
    off = part.spectra.offsets.to(knn.device).to(torch.int64)
    npk = (off[1:] - off[:-1])[rows]
    n_cand = int(ok.sum())
    cand_peaks = int((npk * ok).sum())
    same = int(r.n_candidates.to(torch.int64).sum()) == n_cand
    enc_ms = stages['encode']['ms_total'] / max(stages['encode']['launches'], 1)
    res_ms = (stages['rescore']['ms_total'] + stages['filter']['ms_total']) / max(stages['rescore']['launches'], 1)
    enc_b = int(q.mz.numel()) * 8 + q.n * sl.config.hash_len * 4
    res_b = cand_peaks * 9 + n_cand * 32
    return {'encode': {'bytes_per_batch': enc_b, 'ms': round(enc_ms, 4),
                       'GBps': round(enc_b / (enc_ms * 1e-3) / 1e9, 2) if enc_ms > 0 else None,
                       'bytes': '8 B per query peak + 4 B x hash_len per vector written'},
            'rescore': {'bytes_per_batch': res_b, 'ms': round(res_ms, 4),
                        'GBps': round(res_b / (res_ms * 1e-3) / 1e9, 2) if res_ms > 0 else None,
                        'candidates': n_cand, 'candidate_peaks': cand_peaks,
                        'bytes_per_candidate': round(res_b / max(n_cand, 1), 1),
                        'bytes': '9 B per candidate peak (packed record) + 32-B row record per candidate that '
                                 'passed the precursor window',
                        'candidate_count_equals_the_kernels': same,
                        'note': 'stage time inside the pipelined step (filter + rescoring launches); '
                                'latency / gather bound: no roofline claim (SURVEY.md 8d)'}}


def oracle_context(sl, part, idx):
    """Host copies the oracle legs share: the library's peaks and the index as the oracle reads it
    (inverted lists from the device, in list order). IVF-Flat: the stored vectors as sparse rows
    (``HostIVF.to_csr``) -- the sparse-aware CPU scan, same chain, same bits as the dense one."""
    from oracle import oracle_py as O
    Lh = O.Spectra(*part.spectra.to('cpu').numpy())
    off, ids, payload = idx.lists()
    info = idx.info()
    ivf = O.HostIVF.__new__(O.HostIVF)
    ivf.centroids, ivf.nlist, ivf.d = idx.centroids(), info.nlist, info.d
    ivf.list_offsets, ivf.ids, ivf.payload = off, ids, payload
    ivf.codebooks = idx.codebooks() if info.kind == 2 else None
    ivf.kind = 1 if info.kind == 2 else 0
    dense = None
    if ivf.kind == 0:
        dense = ivf
        ivf = ivf.to_csr()
    return {'O': O, 'Lh': Lh, 'ivf': ivf, 'ivf_dense': dense}


def cascade_parity(args, sl, part, charge, cfg, q, ids, thr, seconds, ctx, nprobe):
    """A sample of the cascade pass's queries through the ORACLE's two levels: standard search
    (every library spectrum inside the standard window, best shifted dot, cosine over its peak
    matches), the gate (cosine >= thr), then for the rest the oracle's open search over the same
    index. Compared with what the pass filed for those queries: library row, score, q."""
    import numpy as np
    import torch
    t0 = time.time()
    ctx = ctx or oracle_context(sl, part, sl._get_ann_index(charge))
    O, Lh, ivf = ctx['O'], ctx['Lh'], ctx['ivf']
    lib_pmz32 = np.asarray(part.precursor_mz, np.float32)
    lib_pmz = lib_pmz32.astype(np.float64)
    order = np.argsort(lib_pmz, kind='stable')
    sorted_pmz = lib_pmz[order]
    tol, mode = cfg.precursor_tolerance_mass, cfg.precursor_tolerance_mode
    nq = q.n
    # bounded sample, spread over the pass: ~1 ms (std) + ~4 ms / cores (open) per query
    n_s = int(max(64, min(nq, seconds * 150)))
    rows = np.unique(np.linspace(0, nq - 1, n_s).astype(np.int64))
    qs = q.select(torch.as_tensor(rows, device=q.device)).to('cpu')
    Q = O.Spectra(*qs.numpy())
    qpmz = np.asarray(qs.precursor_mz, np.float64)
    qo, qmz, qit = Q.offsets, Q.mz, Q.intensity

    def cosine(i, row, pm):
        a, b = Lh.offsets[row], Lh.offsets[row + 1]
        return float(O.ssm_features(qmz[qo[i]:qo[i + 1]], qit[qo[i]:qo[i + 1]], Lh.mz[a:b],
                                    Lh.intensity[a:b], pm)[0])
    want = {}
    rest = []
    for i in range(len(rows)):
        if mode == 'ppm':        # |q - l| / l * 1e6 <= tol (spectral_library.py:421-427): a slightly wider
            lo = np.searchsorted(sorted_pmz, qpmz[i] / (1 + tol * 1.001e-6) - 1e-9)       # range, then the formula
            hi = np.searchsorted(sorted_pmz, qpmz[i] / (1 - tol * 1.001e-6) + 1e-9)
            c = order[lo:hi]
            c = c[np.abs(qpmz[i] - lib_pmz[c]) / lib_pmz[c] * 10 ** 6 <= tol]
        else:
            lo = np.searchsorted(sorted_pmz, qpmz[i] - tol / charge - 1e-6)
            hi = np.searchsorted(sorted_pmz, qpmz[i] + tol / charge + 1e-6)
            c = order[lo:hi]
            c = c[np.abs(qpmz[i] - lib_pmz[c]) * charge <= tol]
        c = np.sort(c).astype(np.int64)
        if len(c):
            b, sc, pm = O.best_match(Q, i, Lh, c, cfg.fragment_mz_tolerance, cfg.allow_peak_shifts)
            if b >= 0:
                cs = cosine(i, int(c[b]), pm)
                if cs >= thr:
                    want[int(rows[i])] = (int(c[b]), cs, 0.0)
                    continue
        rest.append(i)
    if rest:                     # level 2: the oracle's open search of the remainder, all host threads
        sub = qs.select(torch.as_tensor(np.asarray(rest)))
        Qr = O.Spectra(*sub.numpy())
        r = O.search_batch(Qr, Lh, lib_pmz32, charge, ivf, args.k, nprobe, args.open_da, 'Da',
                           cfg.fragment_mz_tolerance, cfg.allow_peak_shifts, pm_stride=64)
        for j, i in enumerate(rest):
            row = int(r['best_row'][j])
            if row >= 0:
                cs = cosine(i, row, r['pm_pairs'][j][:int(r['pm_count'][j])])
                want[int(rows[i])] = (row, cs, 0.0 if cs >= thr else 1.0)
    got = {}
    sel = np.isin(ids.qrow, rows)
    for qr, lr, sc, qq in zip(ids.qrow[sel], ids.lib_row[sel], ids.score[sel], ids.q[sel]):
        got[int(qr)] = (int(lr), float(sc), float(qq))
    rows_equal = sum(1 for k_ in want if k_ in got and got[k_][0] == want[k_][0] and got[k_][2] == want[k_][2])
    dmax = max([abs(got[k_][1] - want[k_][1]) for k_ in want if k_ in got] or [0.0])
    return {'queries': int(len(rows)), 'kept_at_level_1': int(len(rows) - len(rest)),
            'open_search_of_the_rest': int(len(rest)),
            'same_queries_identified': bool(set(want) == set(got)),
            'library_row_and_q_equal': int(rows_equal), 'identified_by_oracle': int(len(want)),
            'cosine_max_abs_diff': float(dmax),
            'all_equal': bool(set(want) == set(got) and rows_equal == len(want) and dmax <= 1e-9),
            'seconds': round(time.time() - t0, 1)}


def beyond_llc_leg(args, sl, part, idx, q, dev, n_queries=8192, reps=3):
    """The dominant kernel where HBM serves it: an IVF-PQ index with the bench index's
    own coarse quantiser and codebooks (FAISS, too, trains an IndexIVF on a sub-sample: at most 256 x
    nlist points), grown by add() to `--beyond-llc-chunks` libraries of the bench's size -- chunk 0 is
    the bench library, the others are the same generator under other seeds: distinct spectra, real
    codes, the real list-length distribution -- until codes + ids are a multiple of the 256 MiB Infinity
    Cache. The same `pq_scan_v3_kernel`, same nprobe and k, 8 192 queries per launch (a launch is
    ~4x the bytes of the bench's own); HIP events around the kernel on its stream; algorithmic bytes
    = scanned vectors x 36 B. Not part of `value`."""
    import torch
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd import faiss_compat as faiss
    L = _lib.lib()
    t0 = time.time()
    big = faiss.IndexIVFPQ(faiss.IndexFlatIP(800), 800, args.nlist, args.pq_m, 8)
    big.set_trained(idx.centroids(), idx.codebooks())
    x = sl._encode(part.spectra)
    big.add(x)
    del x
    for c in range(1, args.beyond_llc_chunks):
        lib_c, _ = synthetic.make_library(args.library_size, seed=7000 + c, device=dev, charges=(2,), charge_p=(1.0,))
        x = sl._encode(lib_c)
        del lib_c
        big.add(x)
        del x
    big.nprobe = args.nprobe
    nq = min(n_queries, q.n)
    xq = sl._encode(q.select(torch.arange(nq, device=dev)).contiguous())
    D = torch.empty((nq, args.k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, args.k), dtype=torch.int64, device=dev)
    big.search(xq, args.k, D, I)            # builds the lists, loads the code
    torch.cuda.synchronize()
    t_build = time.time() - t0
    L.asl_profile_reset()
    L.asl_profile_enable(1)
    for _ in range(reps):
        big.search(xq, args.k, D, I)
    torch.cuda.synchronize()
    L.asl_profile_enable(0)
    ms, n = C.c_double(), C.c_int64()
    L.asl_profile_get(b'scan', C.byref(ms), C.byref(n))
    launches = max(n.value, 1)
    scanned = L.asl_profile_scanned_vectors() / launches
    avg = ms.value / launches
    ntotal = int(big.info().ntotal)
    ok = bool(torch.isfinite(D).all()) and int(I.min()) >= 0 and int(I.max()) < ntotal
    del big, D, I, xq
    torch.cuda.empty_cache()
    achieved = scanned * BYTES_PER_SCANNED_VECTOR / (avg * 1e-3) / 1e9
    resident = ntotal * (BYTES_PER_SCANNED_VECTOR + 1)
    return {'kernel': 'pq_scan_v3_kernel', 'bound': 'hbm',
            'served_by': f'HBM: {resident / 1e6:.0f} MB of codes + ids = {resident / LLC_BYTES:.1f} x the 256 MiB Infinity '
                         f'Cache (L2 hit rate 0.6 %, profiles/r06_pq_beyond_llc_pmc_summary.txt)',
            'library_vectors': ntotal, 'chunks': args.beyond_llc_chunks, 'resident_mb': round(resident / 1e6, 1),
            'queries_per_launch': nq, 'launches': launches, 'avg_launch_ms': round(avg, 3),
            'vectors_scanned_per_query': round(scanned / nq, 1),
            'algorithmic_bytes_per_launch': int(scanned * BYTES_PER_SCANNED_VECTOR),
            'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 5),
            'frac_of_measured_hbm_copy': round(achieved / HBM_COPY_GBS, 5),
            'achieved_codes_only': round(achieved * BYTES_PER_CODE / BYTES_PER_SCANNED_VECTOR, 2),
            'results_sane': ok, 'build_seconds': round(t_build, 1),
            'note': 'index = the bench index\'s quantiser + codebooks, grown by add() with further synthetic libraries '
                    '(other seeds); same kernel, nprobe and k. Design choices re-measured in this regime '
                    '(prefetch depth 2 / 3, non-temporal loads, 2 instead of 3 workgroups per CU: all slower, '
                    'profiles/r06_pq_beyond_llc_variants.txt)'}


_FALLBACK = {'line': None, 'rank': 0, 'timer': None, 'stage': 'sharding the index'}


def _arm_fallback(rank, seconds, line):
    """Keep a complete result line (the replicas layout, already measured) in reserve for the rest of an
    N > 1 run. A watchdog thread prints it and leaves through os._exit(0) -- never by replacing the
    process -- when `seconds` pass; main()'s exception handler prints it when the sharded path raises."""
    import threading
    _FALLBACK['line'], _FALLBACK['rank'] = line, rank
    if seconds and seconds > 0:
        def bark():
            _emit_fallback(f'no result after {seconds:g} s (watchdog); last stage: {_FALLBACK["stage"]}')
            os._exit(0)
        t = threading.Timer(seconds, bark)
        t.daemon = True
        t.start()
        _FALLBACK['timer'] = t


def _disarm_fallback():
    if _FALLBACK['timer'] is not None:
        _FALLBACK['timer'].cancel()
    _FALLBACK['line'] = _FALLBACK['timer'] = None


def _emit_fallback(why):
    line = _FALLBACK['line']
    if line is None:
        return False
    _FALLBACK['line'] = None
    log(f'[bench] rank {_FALLBACK["rank"]}: list-sharded path failed: {why} -- falling back to the replicas line')
    if _FALLBACK['rank'] == 0:
        line = dict(line, sharded_path_failed=why)
        emit_line(json.dumps(line))
    return True


def self_launch(n):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a
    child process (the launch line the driver itself uses for N > 1) on a free local port and return
    its exit code. Called before any import of torch: the launcher never initialises a GPU."""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this pool
    env.pop('MASTER_PORT', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    log(f'[bench] --gpus {n} without a launcher: starting {n} ranks as child processes (port {port})')
    child = subprocess.Popen(cmd, env=env)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        try:
            return child.wait(timeout=30)
        except subprocess.TimeoutExpired:
            child.kill()
            return child.wait()


def launch_check(args, world, rank):
    """ASL_BENCH_LAUNCH_CHECK=1: the ranks only prove that they were started and can talk (rendezvous,
    one all-reduce over gloo on host tensors, no GPU, no library); rank 0 prints one JSON line. What the
    CPU-box test of the launcher runs (tests/test_cabi_and_host.py)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        dist.init_process_group('gloo')
        x = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(x)
        total = int(x[0])
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1
    if rank == 0:
        emit_line(json.dumps({'launch_check': True, 'n_gpus': world, 'rank_sum': total,
                              'steps': args.steps, 'warmup': args.warmup}))
    return 0 if total == world * (world + 1) // 2 else 19


def preflight(args, world, rank, dev, backend, group=None):
    """Every collective the sharded path uses, once, at a tiny size, under a watchdog: a job whose
    RCCL / xGMI set-up hangs ends here with a diagnostic line and exit code 17 instead of running
    into the driver's limit with nothing to read. The watchdog is a thread that leaves through
    ``os._exit`` -- the process is never replaced by another program."""
    import threading
    import torch
    import torch.distributed as dist
    state = {'at': 'start'}

    def bark():
        sys.stderr.write(f'[bench] PREFLIGHT TIMEOUT on rank {rank}/{world}: collective '
                         f'"{state["at"]}" did not finish within {args.preflight_seconds:g} s '
                         f'(backend {backend}); MASTER_ADDR={os.environ.get("MASTER_ADDR")} '
                         f'LOCAL_RANK={os.environ.get("LOCAL_RANK")} '
                         f'HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}\n')
        sys.stderr.flush()
        if _emit_fallback(f'preflight: collective "{state["at"]}" did not finish within {args.preflight_seconds:g} s'):
            os._exit(0)          # the replicas line is out: the run has its result
        os._exit(17)
    dog = threading.Timer(args.preflight_seconds, bark)
    dog.daemon = True
    dog.start()
    t0 = time.perf_counter()
    cdev = dev if backend == 'nccl' else torch.device('cpu')
    took = {}

    def run(name, fn):
        state['at'] = name
        t = time.perf_counter()
        fn()
        if backend == 'nccl':
            torch.cuda.synchronize()
        took[name] = round((time.perf_counter() - t) * 1e3, 2)
    x = torch.full((world * 8,), rank, dtype=torch.int64, device=cdev)
    run('barrier', lambda: dist.barrier(group=group))
    run('all_reduce(max)', lambda: dist.all_reduce(x.clone(), op=dist.ReduceOp.MAX, group=group))
    if backend == 'nccl':
        out = torch.empty(world * x.numel(), dtype=torch.int64, device=cdev)
        run('all_gather_into_tensor', lambda: dist.all_gather_into_tensor(out, x, group=group))
        y = torch.empty_like(x)
        run('all_to_all_single', lambda: dist.all_to_all_single(y, x, group=group))
        ok = bool((y.view(world, 8) == torch.arange(world, device=cdev).unsqueeze(1)).all())
    else:
        parts = [torch.empty_like(x) for _ in range(world)]
        run('all_gather', lambda: dist.all_gather(parts, x, group=group))
        ok = all(int(p_[0]) == r for r, p_ in enumerate(parts))
    dog.cancel()
    if not ok:
        sys.stderr.write(f'[bench] PREFLIGHT: rank {rank} received wrong data from a collective\n')
        if _emit_fallback('preflight: a collective delivered wrong data'):
            os._exit(0)
        os._exit(18)
    if rank == 0:
        log(f'[bench] preflight ok in {time.perf_counter() - t0:.2f}s: {took} ms (first call of each)')


def time_collectives(comm_log, group, world, dev, backend, steps, reps=5):
    """Every collective of the step on its own, at the buffer size the step used per call: ms per
    call (max over ranks, synchronised) and the rate at which a rank's outgoing bytes left."""
    import torch
    import torch.distributed as dist
    out = {}
    for name, c in comm_log.calls.items():
        nbytes = c['buffer_bytes'] // max(c['calls'], 1)
        n = max(world, (nbytes // 8) // world * world)
        cdev = dev if backend == 'nccl' else torch.device('cpu')
        x = torch.zeros(n, dtype=torch.int64, device=cdev)
        if c['kind'] == 'all_to_all' and backend == 'nccl':
            y = torch.empty_like(x)
            fn = lambda: dist.all_to_all_single(y, x, group=group)
        elif backend == 'nccl':
            y = torch.empty(n * world, dtype=torch.int64, device=cdev)
            fn = lambda: dist.all_gather_into_tensor(y, x, group=group)
        else:
            parts = [torch.empty_like(x) for _ in range(world)]
            fn = lambda: dist.all_gather(parts, x, group=group)
        fn()
        if backend == 'nccl':
            torch.cuda.synchronize()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        if backend == 'nccl':
            torch.cuda.synchronize()
        el = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=cdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX, group=group)
        ms = float(el.item()) * 1e3
        leaving = n * 8 * (world - 1) // world if c['kind'] == 'all_to_all' else n * 8 * (world - 1)
        out[name] = {'buffer_bytes': n * 8, 'ms_per_call': round(ms, 4),
                     'GBps_out_per_rank': round(leaving / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                     'calls_per_step': c['calls'] / max(steps, 1)}
    return out


def _lib_handle():
    from ann_solo_amd import _lib
    return _lib.lib()


PMC_TRAFFIC_FILE = 'profiles/r06_pmc_traffic.json'
PMC_TRAFFIC_FILE_FLAT = 'profiles/r06_ivfflat_np112_pmc_traffic.json'     # IVF-Flat (float32 postings, the default storage), nprobe 112
# the sources that define each scan kernel: a committed PMC pass is quoted only while their hash is
# the one recorded in it (scripts/profile_round.sh writes `kernel_source_sha1`)
KERNEL_SOURCES = {'pq': ('pq_scan_v3.hip', 'pq_tile.hpp', 'hist_topk.hpp', 'topk.hpp', 'common.hpp'),
                  'flat': ('flat_scan.hip', 'hist_topk.hpp', 'topk.hpp', 'common.hpp')}


def kernel_source_sha1(which):
    import hashlib
    h = hashlib.sha1()
    for name in KERNEL_SOURCES[which]:
        with open(os.path.join(ROOT, 'ann_solo_amd', 'csrc', name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()


def _load_traffic(rel, which):
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f'{rel} missing'
    try:
        with open(path) as f:
            d = json.load(f)
        want = d.get('kernel_source_sha1')
        have = kernel_source_sha1(which)
        if want != have:
            return None, (f'{rel} was taken on other kernel sources (sha1 {str(want)[:10]} there, {have[:10]} here): '
                          f'not quoted')
        return int(d['scan_hbm_bytes_per_launch']), have
    except Exception as e:
        return None, f'{rel} unreadable: {e}'


def pmc_traffic(args, world):
    """(bytes, source): HBM-side bytes per scan launch from the COMMITTED PMC pass of this build
    (rocprofv3 --pmc FETCH_SIZE in its own run, scripts/pmc.sh; FETCH_SIZE is in KiB and reads 1/2
    of a wide coalesced stream on gfx950, MI355X_MICROARCH.md) -- a constant read from a file, not
    measured by this run (counters cannot be collected from inside the process); the JSON says so
    in `traffic_source`. Only for the exact default workload; else (None, reason)."""
    default = (world == 1 and args.library_size == 2_100_000 and args.nlist == 4096 and
               args.nprobe == 128 and args.k == 1024 and args.batch == 32768 and
               args.index == 'ivfpq' and args.scan_variant == 0 and args.niter == 25)
    if not default:
        return None, 'not the default workload: no committed PMC pass applies'
    traffic, how = _load_traffic(PMC_TRAFFIC_FILE, 'pq')
    if traffic is None:
        return None, how
    return traffic, (f"{PMC_TRAFFIC_FILE} (committed rocprofv3 --pmc FETCH_SIZE pass of these kernel sources, sha1 "
                     f"{how[:10]}; x2 gfx950 correction; not measured in this run)")


def flat_pmc_traffic(args, nprobe):
    """Fabric bytes per launch of the postings scan from the committed PMC pass (same rule as
    ``pmc_traffic``: only for the workload the pass was taken on)."""
    ok = (args.library_size == 2_100_000 and args.nlist == 4096 and nprobe == 112 and
          args.k == 1024 and args.batch == 32768 and args.niter == 25)
    if not ok:
        return None, 'no committed PMC pass for this workload'
    traffic, how = _load_traffic(PMC_TRAFFIC_FILE_FLAT, 'flat')
    if traffic is None:
        return None, how
    return traffic, (f"{PMC_TRAFFIC_FILE_FLAT} (committed rocprofv3 --pmc FETCH_SIZE pass of these kernel sources, sha1 "
                     f"{how[:10]}; x2: one 128-byte request per line tallied at 64 B, which agrees with the line "
                     f"count; not measured in this run)")


def postings_roofline(sl, idx, q, nprobe, avg_ms, args=None):
    """Roofline of ``flat_inv_scan_kernel`` for this batch: algorithmic bytes = sum over (query,
    probed block, non-zero query dimension) of the 4-byte table word + 6 bytes per posting
    (u16 local index + f32 value), counted on the device by ``asl_index_postings_work`` outside
    the timed region; ``lines`` = the 128-byte lines those bytes occupy (what a cold scan has to
    move). HBM-bound: every (query, block) pair touches its own lines."""
    b, l = idx.postings_work(sl._encode(q), nprobe)
    layout = idx.flat_layout
    traffic, src = flat_pmc_traffic(args, nprobe) if args is not None and layout == 1 else (None, None)
    achieved = b / (avg_ms * 1e-3) / 1e9
    by_line = l * 128 / (avg_ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': 'flat_inv_scan_kernel', 'achieved': round(achieved, 2),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5),
            'layout': {1: 'float postings (6 B) behind 4-byte table words', 2: 'fixed-point posting words (4 B) '
                       'in whole 128-byte lines behind a one-byte-per-dimension table'}.get(layout),
            'bytes': ('1-B table entry + 4 B per posting' if layout == 2 else '4-B table word + 6 B per posting') +
                     ', per (probed block, non-zero query dimension)',
            'algorithmic_bytes_per_launch': int(b), 'avg_launch_ms': round(avg_ms, 4),
            'lines_128B_per_launch': int(l), 'achieved_by_lines': round(by_line, 2),
            'frac_by_lines': round(by_line / HBM_PEAK_GBS, 5), 'traffic': traffic,
            'traffic_source': src,
            'frac_measured_traffic': (round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                      if traffic else None)}


def faiss_cpu_leg(args, sl, part, idx, q, charge, cores, n_queries, knn_gpu=None):
    """when ``import faiss`` succeeds on this box, time the reference's own
    retrieval library on the host cores -- ``IndexIVFFlat`` / ``IndexIVFPQ`` (inner product) over
    the same hashed vectors, with THIS index's coarse centroids as the quantiser (so nothing but
    the PQ codebooks is trained), same nlist / nprobe / k, ``omp_set_num_threads(cores)``.
    Retrieval only (FAISS has no rescoring). For IVF-Flat the ids FAISS returns are compared with
    the GPU's for the same queries (``knn_gpu``: rows of the same index): same quantiser, same
    lists, exact inner products on both sides, so the sets differ only where scores tie within
    float rounding at the k-th place (FAISS sums in SIMD-lane order, the GPU in ascending
    dimension) -- ``knn_id_overlap_vs_gpu`` and the largest |difference| of the sorted score rows
    say by how much. Returns a dict, or the string 'unavailable'."""
    try:
        import faiss as F
    except Exception:
        return 'unavailable'
    import numpy as np
    import torch
    try:
        info = idx.info()
        d, nlist = info.d, info.nlist
        F.omp_set_num_threads(int(cores))
        quantizer = F.IndexFlatIP(d)
        quantizer.add(np.ascontiguousarray(idx.centroids(), np.float32))
        if info.kind == 2:
            fx = F.IndexIVFPQ(quantizer, d, nlist, info.pq_m, 8, F.METRIC_INNER_PRODUCT)
        else:
            fx = F.IndexIVFFlat(quantizer, d, nlist, F.METRIC_INNER_PRODUCT)
        t0 = time.perf_counter()
        n = part.spectra.n
        step = 262144
        if not fx.is_trained:          # PQ codebooks: FAISS' own trainer on <= 256 points per code
            rows = torch.randperm(n)[:min(n, 65536)].sort().values
            fx.train(sl._encode(part.spectra.select(rows)).cpu().numpy())
        for a in range(0, n, step):    # encode on the GPU in slices, add on the host
            rows = torch.arange(a, min(n, a + step))
            fx.add(sl._encode(part.spectra.select(rows)).cpu().numpy())
        t_build = time.perf_counter() - t0
        fx.nprobe = int(args.nprobe)
        nq = int(min(q.n, max(64, n_queries)))
        xq = sl._encode(q.select(torch.arange(nq, device=q.device))).cpu().numpy()
        fx.search(xq[:min(nq, 64)], args.k)                      # warm-up
        t0 = time.perf_counter()
        Df, I = fx.search(xq, args.k)
        dt = time.perf_counter() - t0
        out = {'value': round(nq / dt, 2), 'unit': 'query spectra/s (retrieval only)',
               'index': 'IndexIVFPQ' if info.kind == 2 else 'IndexIVFFlat', 'threads': int(cores),
               'queries': nq, 'build_s': round(t_build, 1), 'version': getattr(F, '__version__', '?')}
        if info.kind != 2 and knn_gpu is not None:
            g = np.asarray(knn_gpu)[:nq]
            m = min(len(g), len(I))
            inter = sum(len(np.intersect1d(g[i][g[i] >= 0], I[i][I[i] >= 0])) for i in range(m))
            total = sum(int((g[i] >= 0).sum()) for i in range(m))
            idx.nprobe = int(args.nprobe)
            Dg, _ = idx.search(xq[:m], args.k)
            out.update({'knn_id_overlap_vs_gpu': round(inter / max(total, 1), 6), 'queries_compared': m,
                        'max_abs_score_diff_sorted_rows': float(np.abs(np.asarray(Dg) - Df[:m]).max()),
                        'tie_rule': 'GPU / oracle: (score desc, id asc) on fp32 scores of the ascending-'
                                    'dimension fmaf chain over the stored (2^-22 grid) components; FAISS: '
                                    'its own summation order on the float32 vectors'})
        return out
    except Exception as e:             # a FAISS build without these classes, out of memory, ...
        return f'importable, leg failed: {type(e).__name__}: {e}'


def host_cores():
    """How many host cores this process may really use: the affinity mask, cut down to the cgroup
    CPU quota when there is one (a container that sees 128 CPUs but is throttled to a quota of a
    few runs 128 OpenMP threads at a fraction of their speed -- round 4's "11x on 128 threads")."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota|max> <period>"
            a, b = f.read().split()
            if a != 'max':
                quota = float(a) / float(b)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f1, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f2:
                qv, pv = float(f1.read()), float(f2.read())
                if qv > 0:
                    quota = qv / pv
        except (OSError, ValueError):
            pass
    physical = None
    try:
        seen = set()
        with open('/proc/cpuinfo') as f:
            phys = core = None
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        physical = len(seen) or None
    except OSError:
        pass
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {'affinity_cpus': aff, 'cgroup_cpu_quota': quota, 'physical_cores_in_cpuinfo': physical,
            'usable': usable}


def cpu_baseline(args, sl, part, idx, q, res, charge, cfg, faiss_leg=True, ctx=None):
    """The oracle (plain-C port of the reference path, oracle/) on the host cores of this
    box, on a bounded sample of the SAME batch against the SAME index; doubles as a
    full-size parity check of the GPU results. IVF-Flat: the oracle scans a SPARSE copy of the
    stored vectors (per-list CSR, the same ascending fmaf chain: what a sparse-aware CPU
    implementation does), and a few queries also go through its dense rows -- the definition --
    to show both give the same bits."""
    import numpy as np
    import torch
    t0 = time.time()
    ctx = ctx or oracle_context(sl, part, idx)
    O, Lh, ivf = ctx['O'], ctx['Lh'], ctx['ivf']
    host = host_cores()
    cores = max(1, min(host['usable'], O.max_threads()))

    def run(n, threads, rows=None):
        rows = torch.arange(n, device=q.device) if rows is None else rows
        qs = q.select(rows).to('cpu')
        Q = O.Spectra(*qs.numpy())
        t = time.perf_counter()
        r = O.search_batch(Q, Lh, part.precursor_mz, charge, ivf, args.k, args.nprobe,
                           args.open_da, 'Da', cfg.fragment_mz_tolerance, cfg.allow_peak_shifts,
                           pm_stride=64, nthreads=threads, want_knn=True)   # (`ivf`: the binding at call time)
        return time.perf_counter() - t, r
    n1 = min(q.n, 64)
    t1, r1 = run(n1, 1)                       # single core, also calibrates the sample size
    dense_check = None
    if ctx.get('ivf_dense') is not None:      # the dense rows (the definition) on the same queries
        sparse_ivf, ivf = ivf, ctx['ivf_dense']
        td, rd = run(n1, 1)
        ivf = sparse_ivf
        dense_check = {'queries': n1, 'ms_per_query_dense_rows': round(td / n1 * 1e3, 2),
                       'ms_per_query_sparse_rows': round(t1 / n1 * 1e3, 2),
                       'knn_ids_and_winners_equal': bool(np.array_equal(rd['knn_I'], r1['knn_I']) and
                                                         np.array_equal(rd['best_row'], r1['best_row']) and
                                                         np.array_equal(rd['best_score'], r1['best_score']))}
    per_q = t1 / n1
    n_all = int(min(q.n, max(cores, args.cpu_seconds / max(per_q, 1e-6))))
    t_all, r_all = run(n_all, cores)
    if n_all > n1:      # the single-core rate on a sample spread over the SAME queries (queries differ in cost),
        rows1 = torch.arange(n1, device=q.device) * (n_all // n1)      # the faster of two runs (the first one pages the index in)
        t1 = min(run(n1, 1, rows1)[0], run(n1, 1, rows1)[0])
        per_q = t1 / n1
    best_row = res.best_row[:n_all].cpu().numpy()
    best_score = res.best_score[:n_all].cpu().numpy()
    # candidate id sets at bench size: the GPU's k nearest ids of the same queries (sorted rows)
    g = sl._search_batch(q.select(torch.arange(n_all, device=q.device)), charge, 'open',
                         want_knn=True)
    knn_gpu, knn_cpu = np.sort(g.knn, axis=1), np.sort(r_all['knn_I'], axis=1)
    rows_equal = (knn_gpu == knn_cpu).all(axis=1)
    parity = {'queries': n_all,
              'knn_id_sets_equal': bool(rows_equal.all()),
              'knn_rows_differing': int((~rows_equal).sum()),
              'knn_ids_compared': int(knn_gpu.size),
              'best_row_equal': bool(np.array_equal(best_row, r_all['best_row'])),
              'best_score_max_abs_diff': float(np.abs(best_score - r_all['best_score']).max())}
    faiss_note = faiss_cpu_leg(args, sl, part, idx, q, charge, cores, n_all, g.knn) if faiss_leg else None
    log(f'[bench] cpu baseline: {n_all} queries on {cores} threads in {t_all:.2f}s, '
        f'single core {per_q * 1e3:.2f} ms/query (setup {time.time() - t0:.1f}s)')
    eff = (n_all / t_all) / (cores / per_q) if per_q > 0 else None
    return {'value': round(n_all / t_all, 2), 'unit': 'query spectra/s', 'cores': cores,
            'host': host,
            # value / (cores x the single-thread rate of the same build on the same box)
            'parallel_efficiency': round(eff, 3) if eff else None,
            'kind': 'port', 'faiss': faiss_note,
            'sample': f'{n_all} queries of the same batch, same index, all {cores} host threads '
                      f'(OpenMP over queries)' + ('; IVF-Flat lists as sparse rows (CSR), same fmaf chain'
                                                  if dense_check else ''),
            'dense_definition_check': dense_check,
            'single_core_value': round(1.0 / per_q, 2),
            'single_core_sample': f'{n1} queries spread over the same sample, 1 thread',
            'parity_vs_gpu': parity}


if __name__ == '__main__':
    try:
        main()
    except BaseException as e:          # (SystemExit included: sys.exit('...') inside the sharded path)
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        if _emit_fallback(f'{type(e).__name__}: {e} (stage: {_FALLBACK["stage"]})'):
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(0)                 # the other ranks may sit in a collective: do not wait for teardown
        raise
