"""Multi-process (gloo, world_size 2, CPU) test of the slice sharding + single gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from miccai24_immoco_amd.shard import gather_images, shard_range, solve_sharded


def test_shard_range_partitions():
    for n in (1, 2, 7, 64, 512):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _fake_solve(i):
    # deterministic "image" of slice i
    g = torch.Generator().manual_seed(100 + i)
    return torch.complex(torch.randn(6, 8, generator=g), torch.randn(6, 8, generator=g))


def _worker(rank, world, port, n_slices, dst, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if n_slices < world:
            # fewer slices than ranks: EVERY rank refuses up front (nobody is left waiting in the collective)
            try:
                solve_sharded(n_slices, _fake_solve, dst=dst)
                q.put((rank, False))
            except ValueError:
                q.put((rank, True))
            return
        out = solve_sharded(n_slices, _fake_solve, dst=dst)
        ok = True
        if dst is None or rank == dst:
            ref = torch.stack([_fake_solve(i) for i in range(n_slices)])
            ok = out is not None and out.shape == ref.shape and torch.equal(out, ref)
        else:
            ok = out is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_slices,dst", [(4, None), (5, None), (5, 0), (1, None)])
def test_sharded_solve_and_gather_world2(n_slices, dst):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, dst, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_gather_single_process_passthrough():
    x = torch.stack([_fake_solve(i) for i in range(3)])
    assert gather_images(x, 3) is x
