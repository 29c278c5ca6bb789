"""CPU, world_size 2 over gloo: the N>1 plumbing (shard partition + the one
statistics reduction).  The per-rank work is a stand-in (a counter): the
solve itself needs a GPU and is covered by the -m gpu tests."""
import os
import socket

import pytest

from eggshell_amd import dist as egs_dist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = egs_dist.shard_range(n, r, world)
                assert 0 <= b <= e <= n
                seen.extend(range(b, e))
            assert seen == list(range(n))
            sizes = [egs_dist.shard_range(n, r, world)[1] - egs_dist.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    r, w, _ = egs_dist.init_process_group("gloo")
    b, e = egs_dist.shard_range(1024, r, w)            # C4: 1024 independent ensembles
    units = e - b
    out = egs_dist.reduce_stats(elapsed_s=1.0 + 0.5 * r, units_done=units * 3, contact_iters=units * 256.0 * 50,
                                max_residual=0.1 * (r + 1), failed=(r == 1 and False))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_two_rank_stats_reduction_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        elapsed, units, citers, resid, failed = res[r]
        assert elapsed == 1.5                       # MAX over ranks
        assert units == 1024 * 3                    # SUM of per-rank units
        assert citers == 1024 * 256.0 * 50
        assert abs(resid - 0.2) < 1e-15 and failed is False
