"""Multi-GPU plumbing: independent ensembles are sharded across ranks (one
process per GPU); there is NO data-path collective -- the only exchange is one
small reduction of per-rank statistics per measurement batch (RCCL over xGMI on
GPUs, gloo on CPU).  SURVEY.md 8(e)."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n_units, rank, world):
    """Contiguous block partition of n_units ensembles: [begin, end) of `rank`.
    Block sizes differ by at most one; every unit belongs to exactly one rank."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_process_group(backend=None):
    """Rendezvous from the torchrun environment (MASTER_ADDR/PORT, RANK, ...)."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def reduce_stats(elapsed_s, units_done, contact_iters, max_residual, failed, device=None):
    """One collective round: MAX of elapsed/residual/failure, SUM of work.
    Returns the job-wide (elapsed_s, units, contact_iters, residual, failed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s, units_done, contact_iters, max_residual, bool(failed)
    kw = {"device": device} if device is not None else {}
    mx = torch.tensor([elapsed_s, max_residual, 1.0 if failed else 0.0], dtype=torch.float64, **kw)
    sm = torch.tensor([float(units_done), float(contact_iters)], dtype=torch.float64, **kw)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    mx, sm = mx.cpu(), sm.cpu()
    return float(mx[0]), int(round(float(sm[0]))), float(sm[1]), float(mx[1]), bool(mx[2] > 0)
