"""bench.py in a process that has initialised RCCL first (as every rank of a multi-GPU run has): do the streams
RCCL creates change what the context's streams share?"""
import os, sys, runpy
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
