"""When does RCCL write its version banner to stdout?  (development probe)"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
mode = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
print("A after init", flush=True)
t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("B after all_reduce", flush=True)
dist.barrier(); dist.destroy_process_group()
print("C after destroy", flush=True)
if mode == "dup":
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
elif mode == "exit":
    os._exit(0)
