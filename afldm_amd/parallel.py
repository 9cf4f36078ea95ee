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
    if os.environ.get("AFLDM_SAME_GPU") == "1":
        # test rig only: every rank of a single-GPU box drives device 0 (exercises the N > 1 code path - per-rank capture,
        # barriers, the gather - where only one GPU exists; RCCL refuses two ranks on one device, so the backend is gloo)
        local = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("AFLDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")     # "nccl" is RCCL on ROCm
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


def all_gather_into(out, local):
    """dist.all_gather_into_tensor(out, local); under the gloo backend (test rigs: CPU ranks, or several ranks on one GPU)
    device tensors are staged through host memory, which gloo's all-gather needs."""
    if dist.get_backend() == "gloo" and local.is_cuda:
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous())
        out.copy_(host)
        return
    dist.all_gather_into_tensor(out, local)


def gather_rows(local, total, rank, world):
    """All-gather variable-size row shards back into the global [total, ...] tensor on every rank."""
    if world == 1:
        return local
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        all_gather_into(out, local.contiguous())
        return out
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather_into(out, buf)
    return torch.cat([out[r * pad: r * pad + sizes[r]] for r in range(world)], 0)


def sample_sharded(denoise_fn, total, shape, seed, rank, world, device):
    """denoise_fn(local_noise [b, *shape] on `device`) -> local latents.  Returns the global
    [total, *shape] result on every rank."""
    s, e = shard_range(total, rank, world)
    local = global_noise(total, shape, seed)[s:e].to(device)
    return gather_rows(denoise_fn(local), total, rank, world)


def interleaved(n, rank, world):
    """Indices of `n` independent work items (the harness's shift offsets) taken by `rank`: rank, rank + world, ..."""
    return list(range(rank, n, world))


def gather_indexed(world, *dicts):
    """Each rank holds {index: value} dicts for ITS items (harness: frames, errors); returns the dicts merged over all
    ranks, on every rank (one all_gather_object: host-side objects - the frames are CPU tensors)."""
    if world == 1:
        return dicts
    gathered = [None] * world
    dist.all_gather_object(gathered, dicts)
    return tuple({k: v for part in gathered for k, v in part[j].items()} for j in range(len(dicts)))


def run_interleaved(n, rank, world, fn):
    """fn(indices of this rank) -> tuple of {index: value} dicts; returns the merged dicts (every rank)."""
    return gather_indexed(world, *fn(interleaved(n, rank, world)))


def rccl_record(device, local_rank, payload=None):
    """What bench.py prints about the process group so that a multi-GPU line proves N ranks ran: world size, backend,
    every rank's device, and the time of the sampler's ONE collective (all-gather of the final latents) on its own."""
    import time
    if not dist.is_initialized():
        return None
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = torch.device(device).type == "cuda"
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(),
          "device": torch.cuda.get_device_name(device) if on_gpu else "cpu",
          "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", "")) if on_gpu else ""}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    rec = {"world_size": world, "backend": dist.get_backend(), "ranks": ranks,
           "distinct_devices": len({(r["local_rank"], r["uuid"]) for r in ranks})}
    if payload is not None:
        out = torch.empty((world * payload.shape[0],) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        all_gather_into(out, payload)
        sync = torch.cuda.synchronize if on_gpu else (lambda: None)
        sync()
        dist.barrier()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            all_gather_into(out, payload)
        sync()
        dt = (time.perf_counter() - t0) / reps
        # every rank's block must be that rank's payload: compare block checksums with the owners' own
        mine = payload.double().sum().reshape(1)
        gl = dist.get_backend() == "gloo" and mine.is_cuda
        mine_x = mine.cpu() if gl else mine
        sums = [torch.empty_like(mine_x) for _ in range(world)]
        dist.all_gather(sums, mine_x)
        sums = [t.to(mine.device) for t in sums]
        n = payload.shape[0]
        ok = all(bool(torch.equal(out[r * n:(r + 1) * n].double().sum().reshape(1), sums[r])) for r in range(world))
        rec.update(all_gather_us=round(1e6 * dt, 1), all_gather_bytes=out.numel() * out.element_size(),
                   all_gather_verified=ok)
    return rec


def barrier():
    if dist.is_initialized():
        dist.barrier()
