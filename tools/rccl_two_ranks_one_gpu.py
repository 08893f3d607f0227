"""Probe: does RCCL accept two ranks on ONE device?  (The pool leases one GPU per call, so the N > 1 step has never met a second RCCL rank:
VERDICT round 5, missing #1.)  Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rccl_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", device_id=dev)
        x = torch.full((1 << 20,), float(rank + 1), device=dev)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(f"rank {rank}/{world}: all_reduce over two ranks on one device -> {x[0].item()} (expected {world * (world + 1) / 2})", flush=True)
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001 -- the probe's answer IS the exception
        print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:600]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
