"""Probe: can two RCCL ranks share ONE GPU (so that the N>1 sharded path can be exercised on a 1-GPU box)?
Run:  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/probes/rccl_dup.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
x = torch.full((1024,), float(rank + 1), device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
ok = float(x[0].item()) == sum(range(1, world + 1))
y = torch.full((4,), float(rank), device=dev)
if rank == 0:
    dist.send(y, 1)
else:
    dist.recv(y, 0)
torch.cuda.synchronize()
print("rank %d: all_reduce ok=%s p2p got %s" % (rank, ok, y.tolist()), flush=True)
dist.destroy_process_group()
