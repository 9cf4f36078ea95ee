"""Batch sharding of the sampler across the GPUs of one node.

Samples are independent through all denoising steps (GroupNorm and attention are per-sample;
SURVEY.md 8e), so the only communication is ONE all-gather of the final latents over RCCL/xGMI
(16 KiB per sample).  One process per GPU (`torch.distributed.run`); weights are replicated.
The global noise is always drawn on CPU from the seed for the WHOLE batch and sliced per
rank, so results do not depend on the GPU count."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous, balanced [start, end) rows of rank (first total % world ranks get one more)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def global_noise(total, shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((total,) + tuple(shape), generator=g)


def gather_rows(local, total, rank, world):
    """All-gather variable-size row shards back into the global [total, ...] tensor on every rank."""
    if world == 1:
        return local
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * pad: r * pad + sizes[r]] for r in range(world)], 0)


def sample_sharded(denoise_fn, total, shape, seed, rank, world, device):
    """denoise_fn(local_noise [b, *shape] on `device`) -> local latents.  Returns the global
    [total, *shape] result on every rank."""
    s, e = shard_range(total, rank, world)
    local = global_noise(total, shape, seed)[s:e].to(device)
    return gather_rows(denoise_fn(local), total, rank, world)


def barrier():
    if dist.is_initialized():
        dist.barrier()
