"""development probe: does RCCL run two ranks on ONE device here?  (python -m torch.distributed.run --nproc-per-node 2 ...)"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
t = torch.tensor([rank + 1], device="cuda", dtype=torch.int32)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rank", rank, "allreduce max", int(t.item()), flush=True)
dist.destroy_process_group()
