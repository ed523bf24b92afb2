"""The N > 1 path on CPU: world_size-2 gloo process group, slab split + all-gather, with the
per-slab compute supplied by the oracle (there is no GPU here)."""
import os
import socket

import numpy as np
import pytest

from tools.torchrun_sharded import slab_bounds


def test_slab_bounds_cover_the_grid_once():
    for n_grid in (0, 1, 7, 8, 9, 1000, 1_000_001):
        for world in (1, 2, 3, 8):
            seen = []
            pers = set()
            for rank in range(world):
                b, e, per = slab_bounds(n_grid, world, rank)
                assert 0 <= b <= e <= n_grid and e - b <= per
                seen.extend(range(b, e)) if n_grid < 2000 else seen.append((b, e))
                pers.add(per)
            assert len(pers) == 1
            if n_grid < 2000:
                assert seen == list(range(n_grid))
            else:
                assert seen[0][0] == 0 and seen[-1][1] == n_grid
                assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    with pytest.raises(ValueError):
        slab_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_grid, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import scan_oracle as so
    from tools.torchrun_sharded import sharded_scan
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    t = np.sort(rng.uniform(0, 300, 300))
    dy = rng.uniform(0.05, 0.2, 300)
    y = np.sin(2 * np.pi * t / 9.0) + dy * rng.standard_normal(300)
    f0, delta = 0.0007, 0.0011
    freq = f0 + delta * np.arange(n_grid)
    calls = []

    def compute(begin, count):
        calls.append((begin, count))
        part = so.gls_power(t, y, dy, freq[begin:begin + count], delta, f0, sums="exact")
        return torch.from_numpy(part)

    full = sharded_scan(compute, n_grid).numpy()
    # the period-sweep wrapper (PDM / StringLength shape) with the oracle as the per-slab scan
    from tools.torchrun_sharded import sharded_periods
    periods = np.linspace(1.0, 40.0, n_grid)
    theta = sharded_periods(lambda p, dev: so.pdm_scan(t, y, p, 5, 2), periods)
    np.save(os.path.join(out_dir, f"theta{rank}.npy"), theta)
    if rank == 0:
        np.save(os.path.join(out_dir, "theta_want.npy"), so.pdm_scan(t, y, periods, 5, 2))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full)
    np.save(os.path.join(out_dir, f"calls{rank}.npy"), np.array(calls))
    if rank == 0:
        np.save(os.path.join(out_dir, "want.npy"),
                so.gls_power(t, y, dy, freq, delta, f0, sums="exact"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_grid", [101, 64])
def test_two_rank_gloo_allgather_reassembles_the_spectrum(tmp_path, n_grid):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_grid, str(tmp_path)), nprocs=world, join=True)
    want = np.load(tmp_path / "want.npy")
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert got.shape == (n_grid,)
        np.testing.assert_allclose(got, want, rtol=1e-12)      # every rank holds the whole array
        b, e, _ = slab_bounds(n_grid, world, rank)
        assert np.load(tmp_path / f"calls{rank}.npy").tolist() == [[b, e - b]]
        assert np.array_equal(np.load(tmp_path / f"theta{rank}.npy"), np.load(tmp_path / "theta_want.npy"))


def _fallback_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    backend, err = bench.init_dist(torch, dist, rank)           # no GPU here: the RCCL group cannot be built
    coll = bench.Coll(torch, dist, backend)
    send = torch.full((5,), float(rank + 1), dtype=torch.float64)
    full = torch.empty(5 * world, dtype=torch.float64)
    assert coll.gather_into(full, send, async_op=True) is None
    worst = coll.max(float(rank))
    every = coll.all_scalars(10.0 + rank)
    dist.barrier()
    np.save(os.path.join(out_dir, f"fb{rank}.npy"), full.numpy())
    with open(os.path.join(out_dir, f"fb{rank}.txt"), "w") as f:
        f.write(f"{backend}|{err}|{worst}|{every}|{coll.describe()}")
    dist.destroy_process_group()


def test_bench_falls_back_to_gloo_loudly_when_the_rccl_group_cannot_be_built(tmp_path):
    """the first real multi-GPU run must not come back empty.  bench.init_dist() tries RCCL and,
    when that fails (here: no GPU at all), re-initialises the process group on gloo; the collectives bench.py uses
    (max over ranks, all-gather of the slabs, per-rank scalars) then go through host memory and the line records why."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(tmp_path / f"fb{rank}.npy")
        assert got.tolist() == [1.0] * 5 + [2.0] * 5
        backend, err, worst, every, what = open(tmp_path / f"fb{rank}.txt").read().split("|")
        assert backend == "gloo" and err != "None" and float(worst) == 1.0 and every == "[10.0, 11.0]"
        assert what.startswith("FALLBACK")
