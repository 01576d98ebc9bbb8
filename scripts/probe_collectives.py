"""What the collectives of the sharded search can be exercised with on a ONE-GPU box: two ranks on
GPU 0 over (a) gloo with device tensors in the RCCL form (all_to_all_single /
all_gather_into_tensor), (b) RCCL itself (normally refused: two ranks on one device).
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/probe_collectives.py gloo|nccl"""
import sys
import torch
import torch.distributed as dist

kind = sys.argv[1] if len(sys.argv) > 1 else 'gloo'
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
if kind == 'nccl':
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group('gloo')
r, w = dist.get_rank(), dist.get_world_size()
x = (torch.arange(4 * w, dtype=torch.int64) + 100 * r).to(dev)
for name, fn in (('all_to_all_single', lambda: dist.all_to_all_single(torch.empty_like(x), x, async_op=True)),
                 ('all_gather_into_tensor', lambda: dist.all_gather_into_tensor(
                     torch.empty(4 * w * w, dtype=torch.int64, device=dev), x, async_op=True)),
                 ('all_reduce', lambda: dist.all_reduce(x.clone(), op=dist.ReduceOp.MAX, async_op=True))):
    try:
        fn().wait()
        torch.cuda.synchronize()
        print(kind, r, name, 'ok', flush=True)
    except Exception as e:      # noqa: BLE001 -- a probe: report and go on
        print(kind, r, name, 'FAILED', repr(e)[:160], flush=True)
dist.destroy_process_group()
