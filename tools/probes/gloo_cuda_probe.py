#!/usr/bin/env python
"""Probe: do gloo's all_gather_into_tensor / all_reduce accept device tensors (two ranks sharing one GPU)?"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    x = torch.full((4, 3), float(rank + 1), device=dev)
    out = torch.empty(world * 4, 3, device=dev)
    try:
        dist.all_gather_into_tensor(out, x)
        print(rank, "all_gather_into_tensor ok", out[:, 0].tolist(), flush=True)
    except Exception as e:  # noqa: BLE001
        print(rank, "all_gather_into_tensor FAILED", type(e).__name__, str(e)[:120], flush=True)
    try:
        dist.all_reduce(x)
        print(rank, "all_reduce ok", x[0, 0].item(), flush=True)
    except Exception as e:  # noqa: BLE001
        print(rank, "all_reduce FAILED", type(e).__name__, str(e)[:120], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2, 29611), nprocs=2)
