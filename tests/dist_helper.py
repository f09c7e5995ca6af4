"""Rank body of the 2-process gloo test (kept free of package imports so that the spawned
interpreter can import it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    secs = 0.5 if rank == 0 else 0.8          # the slower rank sets the job time
    v = bench.job_throughput(256, secs, world)
    seeds = bench.stream_seed(rank, world)
    out.put((rank, v, seeds))
    dist.destroy_process_group()


