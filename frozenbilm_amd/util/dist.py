"""Process-group helpers with the reference's names (util/dist.py).  One process per GPU; backend "nccl" is RCCL on
ROCm (xGMI between the 8 GPUs of a node), "gloo" on CPU-only hosts (tests).  The SLURM discovery branch of the
reference is cluster specific and not part of the path."""
from __future__ import annotations

import os
import pickle

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def reduce_dict(input_dict, average=True):
    """all-reduce the (scalar) values of a dict, sorted by key (util/dist.py:89-113)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float() for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world
        return {k: v for k, v in zip(names, values)}


def all_gather(data):
    """gather arbitrary picklable data from every rank (util/dist.py:27-86): sizes first, then padded byte tensors."""
    world = get_world_size()
    if world == 1:
        return [data]
    backend = dist.get_backend()
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    payload = torch.frombuffer(bytearray(pickle.dumps(data)), dtype=torch.uint8).to(device)
    size = torch.tensor([payload.numel()], device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    padded = torch.zeros(cap, dtype=torch.uint8, device=device)
    padded[: payload.numel()] = payload
    bufs = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, padded)
    return [pickle.loads(b[:n].cpu().numpy().tobytes()) for b, n in zip(bufs, sizes)]


def init_distributed_mode(args):
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK (util/dist.py:201-238, non-SLURM branch)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        args.distributed = False
        return
    args.distributed = True
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(args.gpu)
    args.dist_backend = backend
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
