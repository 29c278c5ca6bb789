"""CPU, world_size 2 over gloo: the N>1 plumbing (shard partition + the one
statistics reduction).  Each rank picks its BASELINE config 4 shard with the very
function bench.py uses (bench.c4_shard_seeds) and builds the shard's ensembles; the
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
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from eggshell_amd import scenes
    r, w, _ = egs_dist.init_process_group("gloo")
    seeds = bench.c4_shard_seeds(r, w)                 # C4: ensembles 0..1023, seed = global ensemble index
    units = len(seeds)
    # the shard's first and last ensemble, built as bench.run_piles builds them
    nx, ny, nz, sweeps, prec, dt = bench.WORKLOADS["c4"]
    ends = [scenes.box_stack(nx, ny, nz, jitter=1e-3, seed=sd) for sd in (seeds[0], seeds[-1])]
    contacts = sum(e["kind"].shape[0] for e in ends) // 2
    out = egs_dist.reduce_stats(elapsed_s=1.0 + 0.5 * r, units_done=units * 3, contact_iters=units * float(contacts) * sweeps,
                                max_residual=0.1 * (r + 1), failed=(r == 1 and False))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, (out, seeds, float(ends[0]["p"][0, 0]))))


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
    assert sorted(res[0][1] + res[1][1]) == list(range(1024))   # the shards are a partition of ensembles 0..1023
    assert res[0][2] != res[1][2]                                # different seeds: different ensembles
    for r in range(2):
        elapsed, units, citers, resid, failed = res[r][0]
        assert elapsed == 1.5                       # MAX over ranks
        assert units == 1024 * 3                    # SUM of per-rank units = 1024 ensembles x steps
        assert citers == 1024 * 256.0 * 50
        assert abs(resid - 0.2) < 1e-15 and failed is False


def _run_bench(args, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts two ranks itself (gloo here, no GPU:
    --dry-run stops before the first GPU call), rank 0 prints ONE line, and the reduction saw both shards."""
    pytest.importorskip("torch")
    r, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo", "--legs", "none",
                           "--workload", "c4", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = lines[0]
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["units_reduced"] == 1024 * 3             # SUM over both ranks: every ensemble exactly once
    assert out["elapsed_max_s"] == 1.25                 # MAX over ranks
    assert out["rank0_seeds"] == [0, 511]
    # the default (C3, weak scaling): every rank its own --batch piles
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--dist-backend", "gloo", "--legs", "none", "--batch", "5", "--dry-run"])
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["units_reduced"] == 2 * 5 * 2 and lines[0]["scaling"] == "weak"


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies takes the job down with a non-zero exit code (here: no GPU and no --dry-run, so every rank
    fails at Context creation -- the product path never falls back to the CPU)."""
    pytest.importorskip("torch")
    r, lines = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-backend", "gloo", "--legs", "none", "--cpu-seconds", "0"])
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the ranks would run")
    assert r.returncode != 0 and not lines


def test_bench_under_a_launcher_checks_the_world_size():
    r, lines = _run_bench(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
