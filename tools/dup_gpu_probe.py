"""Can two RCCL ranks share the one GPU of this box?  (decides whether the sharded CG can be
run for real here)"""
import os, sys
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl")
t = torch.ones(4, device="cuda") * (rank + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "allreduce ok", t.tolist(), flush=True)
a = torch.zeros(3, device="cuda"); b = torch.full((3,), float(rank), device="cuda")
ops = [dist.P2POp(dist.isend, b, 1 - rank), dist.P2POp(dist.irecv, a, 1 - rank)]
for r in dist.batch_isend_irecv(ops): r.wait()
torch.cuda.synchronize()
print("rank", rank, "sendrecv ok", a.tolist(), flush=True)
dist.destroy_process_group()
